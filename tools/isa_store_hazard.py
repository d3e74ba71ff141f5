"""gfx950 VMEM-store data hazard scan of a hipcc -S listing: a buffer_store_dwordx3/x4 whose data VGPRs are written by the
VERY NEXT instruction (labels skipped, i.e. following the fall-through path).  GFX9 documents "VMEM store of more than
64 bits followed by a VALU write of the write-data VGPRs: 1 wait state"; hipcc does not insert it when the store uses an
SGPR soffset, and on MI355X the store then writes garbage in one 16-lane beat (profiles/r02_epilogue_fence_audit.md).
usage: python tools/isa_store_hazard.py og_api.s   -> per (opcode, overwritten dwords) counts; exit code 1 if any."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def scan(path):
    ins, kern = [], None
    for l in open(path).read().split("\n"):
        l = l.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        if not l or l.startswith(";") or (l.startswith(".") and not l.startswith(".LBB")):
            continue
        ins.append((kern, l))
    hits = []
    for i, (k, t) in enumerate(ins):
        if t.startswith("buffer_store_dwordx4") or t.startswith("buffer_store_dwordx3"):
            data = regs(t.split()[1])
            j = i + 1
            while j < len(ins) and ins[j][1].startswith(".LBB"):
                j += 1
            if j >= len(ins):
                continue
            u = ins[j][1]
            if u.startswith("v_") and not u.startswith("v_cmp") and not u.startswith("v_mfma"):
                ov = tuple(sorted(data.index(r) for r in regs(u.split()[1]) if r in data))
                if ov:
                    hits.append((k, t, u, ov))
    return hits


if __name__ == "__main__":
    hits = scan(sys.argv[1])
    hist = {}
    for k, t, u, ov in hits:
        hist.setdefault((u.split()[0], ov), []).append(k)
    for (op, ov), ks in sorted(hist.items(), key=lambda x: -len(x[1])):
        print(f"  {op:22s} overwrites store-data dwords {ov}: {len(ks):3d} x, e.g. in {ks[0][:56]}")
    print("stores whose data VGPRs are overwritten by the next instruction:", len(hits))
    sys.exit(1 if hits else 0)
