"""CPU: the library's own launch decisions, walked without a GPU (og_unet_plan: a host-only handle, every launch recorded
instead of issued).

Round 3 lost a GPU run to an abort inside the C-ABI while a kernel that exchanges accumulators through a fixed workspace was being
brought up (DESIGN, "the 22:20 abort"): a launch whose tiles do not fit the workspace, the arrival counters or grid.z writes out of
bounds on the device, and the HIP runtime answers a GPU fault with abort() -- nothing the library can turn into an error code after
the fact.  So the bounds are checked BEFORE anything runs: for every micro-batch size 1..64 (and a few larger), every layer of the
full-width net and of nets with other widths / depths / frame shapes, every forced scheduling option -- each launch's grid, LDS,
workspace bytes and counter slots against the library's limits.
"""
import ctypes as C

import pytest

from openglottal_amd._lib import lib

FULL = (32, 64, 128, 256)


def plan(feats, B, H, W, lanes=1, options=""):
    l = lib()
    f = (C.c_int * len(feats))(*feats)
    buf = C.create_string_buffer(1 << 16)
    arena = C.c_longlong(0)
    n = l.og_unet_plan(f, len(feats), B, H, W, lanes, options.encode(), buf, len(buf), C.byref(arena))
    assert n > 0, (n, l.og_last_error())
    recs = []
    for line in buf.value.decode().strip().split("\n"):
        k, gx, gy, gz, blk, lds, ws, cnt = line.rsplit("|", 7)
        recs.append(dict(kernel=k, grid=(int(gx), int(gy), int(gz)), block=int(blk), lds=int(lds), ws=int(ws), cnt=int(cnt)))
    assert len(recs) == n
    return recs, arena.value


def check(recs, what):
    l = lib()
    ws_max, cnt_max, g_max, lds_max = (l.og_workspace_limit(i) for i in range(4))
    for r in recs:
        gx, gy, gz = r["grid"]
        assert gx >= 1 and gy >= 1 and gz >= 1, (what, r)
        assert gy <= g_max and gz <= g_max and gx < 2 ** 31, (what, r)
        assert r["block"] == 256 and r["lds"] <= lds_max, (what, r)
        assert r["ws"] <= ws_max and r["cnt"] <= cnt_max, (what, r)


OPTION_SETS = ["", "wino_w=0", "wino_w=0,wino_ps=0", "wino_w=0,wino_ps=2", "wino_w=0,wino_ps=3", "wino_w=0,wino_ps=4", "wino_w=2", "wino_w=3",
               "wino_w=4", "wino=0", "wino=0,splitk=1", "wino=0,splitk=1,splitk_nt1=0", "wino=0,splitk=1,splitk_fused=0", "convt_w=0",
               "wino=0,conv_impl=1", "wino=0,conv_impl=3", "precision=1", "precision=1,splitk=1", "fuse_head=0", "wino_first=0"]


@pytest.mark.parametrize("options", OPTION_SETS)
def test_every_micro_batch_size_of_the_full_width_net_stays_inside_the_workspace(options):
    """B = 1 .. 64 (every ragged tail a 64-frame chunk can leave) and the larger chunks bench.py / the tests use, 1 and 3 lanes."""
    for B in list(range(1, 65)) + [96, 128, 200, 256]:
        for lanes in (1, 3):
            recs, arena = plan(FULL, B, 256, 256, lanes, options)
            check(recs, (options, B, lanes))
            assert arena <= 288 * 2 ** 30


@pytest.mark.parametrize("feats,shape", [((32, 64), (128, 256)), ((32, 64), (96, 160)), ((64, 128), (48, 64)), ((40, 80), (64, 64)),
                                         ((32, 64, 128), (64, 32)), ((4, 8, 16, 32), (256, 256)), ((32, 64, 128, 256), (512, 512)),
                                         ((16, 32, 64, 128, 256), (256, 256)), ((96, 192), (64, 128))])
def test_other_nets_and_frame_shapes(feats, shape):
    H, W = shape
    for options in ("", "wino_w=0", "wino_w=0,wino_ps=4", "wino_w=2", "wino_w=3", "wino_w=4", "wino=0,splitk=1"):
        for B in (1, 2, 3, 5, 11, 16, 33, 64):
            recs, _ = plan(feats, B, H, W, 1, options)
            check(recs, (feats, shape, options, B))


def test_the_plan_is_the_chain_the_product_runs():
    """Shape of the record list: one first layer, 17 3x3 convs and 4 transposed convs of the full-width net, the per-frame count
    reduction behind the fused head; the one-frame chain takes the wave-split kernels, the 64-frame chain k_conv_wino."""
    r1, _ = plan(FULL, 1, 256, 256)
    k1 = [r["kernel"] for r in r1]
    assert len(k1) == 23 and k1[0].startswith("k_conv_first") and k1[-1] == "k_sum_counts", k1
    assert sum("k_conv_wino_w" in k for k in k1) == 17 and sum("k_convt_w" in k for k in k1) >= 3, k1
    assert sum("k_conv_wino_wp" in k for k in k1) >= 4 and all(r["ws"] > 0 and r["cnt"] > 0 for r in r1 if "k_conv_wino_wp" in r["kernel"])
    r3, _ = plan(FULL, 1, 256, 256, lanes=3)
    assert not any("k_conv_wino_wp" in r["kernel"] for r in r3)          # several lanes in flight: the unsplit kernel (occupancy)
    r64, _ = plan(FULL, 64, 256, 256, lanes=2)
    k64 = [r["kernel"] for r in r64]
    assert sum(k == "k_conv_wino<NT>" for k in k64) == 17 and k64[-1] == "k_sum_counts", k64      # (the recorder keeps the launch site's text)
    assert all(r["ws"] == 0 for r in r64)


def test_bad_arguments_are_error_codes():
    l = lib()
    buf = C.create_string_buffer(64)
    f = (C.c_int * 4)(*FULL)
    assert l.og_unet_plan(f, 4, 1, 250, 256, 1, b"", buf, len(buf), None) < 0          # H not a multiple of 16
    assert l.og_unet_plan(f, 4, 1, 256, 256, 1, b"nonsense=1", buf, len(buf), None) < 0
    assert l.og_unet_plan(f, 4, 1, 256, 256, 1, b"", buf, len(buf), None) < 0           # buffer too small for 23 records
    assert l.og_unet_plan(f, 4, 0, 256, 256, 1, b"", buf, len(buf), None) < 0
