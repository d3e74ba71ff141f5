"""Order of LDS scratch writes (ds_write_b32 / ds_write2*_b32) and read-backs (ds_read_b128 / ds_read2*_b64) in the
epilogue of every k_conv_mfma_o / k_conv_mfma_p instantiation: the in-wave transposition is correct only if each
round's reads follow all of that round's writes in program order (the LDS serves one wave's accesses in order)."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
labs = [(k, l.split(":")[0]) for k, l in enumerate(src) if re.match(r"^_Z1\dk_conv_mfma_[op]I\w+:", l)]
for i, lab in labs:
    j = [k for k in range(i, len(src)) if ".amdhsa_kernel" in src[k] or src[k].startswith(".Lfunc_end")][0]
    body = src[i:j]
    last_mfma = max(k for k, l in enumerate(body) if "v_mfma" in l)
    seq = []
    for l in body[last_mfma:]:
        t = l.strip()
        if t.startswith("ds_write") or t.startswith("ds_store"):
            seq.append("W")
        elif t.startswith("ds_read") or t.startswith("ds_load"):
            seq.append("R")
        elif re.match(r"^\.LBB", t):
            seq.append("|")
    s = "".join(seq)
    # run-length encode
    rle = re.sub(r"(.)\1*", lambda m: f"{m.group(1)}{len(m.group(0))} " if m.group(1) != "|" else "| ", s)
    print(lab[:48], rle)
