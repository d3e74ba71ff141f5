"""CPU-side check of k_conv_wino_w's LDS addressing (no GPU): (1) the raw-halo image the LDS-DMA pieces build and the
addresses the lanes read back name the same (pixel, channel) everywhere, for WB = 1 and 2; (2) every ds_read_b128 of the
raw image, of the U ring and of the exchange buffer is bank-conflict-free under the guide's lane groups
(MI355X_MICROARCH.md, LDS: a b128 read is serviced in four groups of 16 lanes, 64 banks of 4 bytes).
usage: python tools/wino_w_layout_check.py"""
import itertools
import sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def conflicts(addrs):
    """addrs: 64 byte addresses of a ds_read_b128 -> worst number of distinct 16-B slots sharing a bank quad in a lane group"""
    worst = 1
    for g in GROUPS:
        seen = {}
        for l in g:
            a = addrs[l]
            seen.setdefault((a // 16) % 16, set()).add(a // 16)
        worst = max(worst, max(len(v) for v in seen.values()))
    return worst


def check(WB):
    TH, RP = 8 * WB, 18
    RAW_PIX = (TH + 2) * RP
    RAW_PIECES = RAW_PIX * 8
    RAW_IT = (RAW_PIECES + 255) // 256
    GSTRIDE = RAW_PIX * 32
    # (1) DMA side: piece q = it * 256 + tid lands at LDS byte q * 16 and carries (hy, hx, channel quad) of the 32-channel stage
    lds = {}
    for q in range(RAW_IT * 256):
        g, rem = divmod(q, 2 * RAW_PIX)
        P, hp = rem >> 1, rem & 1
        hy, r18 = divmod(P, RP)
        hx = 2 * (r18 - 9) + 1 if r18 >= 9 else 2 * r18
        lg = hp ^ ((hy >> 2) & 1)
        if q < RAW_PIECES:
            lds[q * 16] = (hy, hx, g * 8 + lg * 4)   # first channel of the 16-byte piece
    worst = 1
    for w in range(4):       # wave = row i of the 4x4 position grid
        ra = 0 if w == 0 else 2 if w == 2 else 1
        rb = 2 if w in (0, 1) else 1 if w == 2 else 3
        for wb, j, r, pc in itertools.product(range(WB), range(4), (ra, rb), range(4)):
            addrs = []
            for lane in range(64):
                li, lh = lane & 31, lane >> 5
                wr, wc = li & 3, 2 * (li >> 3) + ((li >> 2) & 1)
                P0 = (2 * wr + r) * RP + (pc & 1) * 9 + wc + (pc >> 1)
                base = P0 * 32 + ((lh ^ (((2 * wr + r) >> 2) & 1)) << 4)
                a = base + j * GSTRIDE + wb * (8 * RP * 32)
                want = (8 * wb + 2 * wr + r, 2 * wc + pc, j * 8 + lh * 4)
                assert lds.get(a) == want, (WB, w, wb, j, r, pc, lane, lds.get(a), want)
                addrs.append(a)
            worst = max(worst, conflicts(addrs))
    print(f"WB={WB}: raw image consistent over {RAW_PIECES} pieces ({RAW_IT} DMA instructions per wave and stage); "
          f"worst raw-read conflict {worst}-way")
    return worst


def check_u():
    worst = 1
    for cc in range(4):
        addrs = []
        for lane in range(64):
            n, lh = lane & 31, lane >> 5
            addrs.append(n * 32 + ((lh ^ ((n >> 3) & 1)) << 4) + cc * 1024)
        worst = max(worst, conflicts(addrs))
    print(f"U fragment reads: worst conflict {worst}-way")
    return worst


if __name__ == "__main__":
    bad = max(check(1), check(2), check_u())
    xw = conflicts([lane * 16 for lane in range(64)])
    print(f"exchange buffer (lane-linear 16 B): {xw}-way")
    sys.exit(0 if bad == 1 and xw == 1 else 1)
