"""Static instruction audit of the k_conv_mfma_o instantiations: vector-ALU instructions before / inside / after the
MFMA loop, VGPRs, scratch.  usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only openglottal_amd/csrc/og_api.hip -o /tmp/api.s
       python tools/isa_audit.py /tmp/api.s"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
labs = [l.split(":")[0] for l in src if re.match(r"^_Z13k_conv_mfma_oI\w+:", l)]
for lab in labs:
    i = [k for k, l in enumerate(src) if l.startswith(lab + ":")][0]
    j = [k for k, l in enumerate(src) if k > i and ".amdhsa_kernel" in l][0]
    blocks = [["entry", []]]
    for l in src[i:j]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append([m.group(1), []]); continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        blocks[-1][1].append(t)
    pre = True; n_pre = n_in = n_mfma = n_post = 0
    for _, ops in blocks:
        v = sum(1 for o in ops if o.startswith("v_") and not o.startswith("v_mfma")); mf = sum(1 for o in ops if o.startswith("v_mfma"))
        if mf:
            pre = False; n_in += v; n_mfma += mf
        elif pre:
            n_pre += v
        else:
            n_post += v
    meta = " ".join(l.strip().replace(".amdhsa_", "") for l in src[j:j + 80] if "next_free_vgpr" in l or "private_segment_fixed_size" in l)
    t = re.search(r"oILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", lab).groups()
    print(f"k_conv_mfma_o<NT={t[0]},MODE={t[1]},TH={t[2]},OCC={t[3]},FIRST={t[4]}>: VALU before loop {n_pre:4d}, in MFMA blocks {n_in:3d} ({n_mfma} MFMA), after (all paths) {n_post:5d}; {meta}")
