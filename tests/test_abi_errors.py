"""CPU tests of the C-ABI's argument checking (no GPU call is made: create/set_tensor are host-only)."""
import ctypes as C

import numpy as np

from openglottal_amd import _lib
from openglottal_amd._lib import lib, ptr


def err():
    return lib().og_last_error().decode()


def test_unet_create_argument_checks():
    l = lib()
    f = (C.c_int * 4)(32, 64, 128, 256)
    assert not l.og_unet_create(None, 4, 1, 1) and "features" in err()
    assert not l.og_unet_create(f, 0, 1, 1)
    assert not l.og_unet_create(f, 4, 3, 1) and "in_ch=1" in err()
    bad = (C.c_int * 3)(8, 12, 20)
    assert not l.og_unet_create(bad, 3, 1, 1) and "double" in err()
    h = l.og_unet_create(f, 4, 1, 1)
    assert h
    l.og_unet_destroy(h)
    l.og_unet_destroy(None)  # harmless


def test_set_tensor_shape_key_dtype_checks_and_call_order():
    l = lib()
    f = (C.c_int * 2)(4, 8)
    h = l.og_unet_create(f, 2, 1, 1)
    w = np.zeros((4, 1, 3, 3), np.float32)
    shp = (C.c_int64 * 4)(4, 1, 3, 3)
    assert l.og_unet_set_tensor(h, b"downs.0.net.0.weight", ptr(w), shp, 4, _lib.OG_DTYPE_F32) == 0
    bad_shp = (C.c_int64 * 4)(4, 1, 5, 5)
    assert l.og_unet_set_tensor(h, b"downs.0.net.0.weight", ptr(w), bad_shp, 4, _lib.OG_DTYPE_F32) == -1 and "size mismatch" in err()
    assert l.og_unet_set_tensor(h, b"no.such.key", ptr(w), shp, 4, _lib.OG_DTYPE_F32) == -1 and "unexpected key" in err()
    assert l.og_unet_set_tensor(h, b"downs.0.net.0.weight", ptr(w), shp, 4, _lib.OG_DTYPE_I64) == -1 and "float32" in err()
    nbt = np.array([7], np.int64)
    assert l.og_unet_set_tensor(h, b"downs.0.net.1.num_batches_tracked", ptr(nbt), (C.c_int64 * 1)(0), 0, _lib.OG_DTYPE_I64) == 0
    assert l.og_unet_set_tensor(h, b"bogus.num_batches_tracked", ptr(nbt), (C.c_int64 * 1)(0), 0, _lib.OG_DTYPE_I64) == -1
    assert l.og_unet_set_tensor(None, b"x", ptr(w), shp, 4, 0) == -1
    # nothing may run before finalize
    x = np.zeros((1, 1, 16, 16), np.float32)
    assert l.og_unet_forward_f32(h, ptr(x), 1, 16, 16, ptr(x)) == -2 and "finalize" in err()
    g = np.zeros((1, 16, 16), np.uint8)
    assert l.og_unet_segment_u8(h, ptr(g), 1, 16, 16, 0.5, None, None, None, None) == -2
    assert l.og_unet_sync(h) == -2
    assert l.og_unet_set_chunk(h, 0) == -1 and l.og_unet_set_chunk(h, 8) == 0
    assert l.og_unet_set_option(h, b"nonsense", 1) == -1 and l.og_unet_set_option(h, b"splitk", 0) == 0
    l.og_unet_destroy(h)


def test_yolo_host_side_checks():
    l = lib()
    assert not l.og_yolo_create(0)
    h = l.og_yolo_create(1)
    w = np.zeros((16, 3, 3, 3), np.float32)
    shp = (C.c_int64 * 4)(16, 3, 3, 3)
    assert l.og_yolo_set_tensor(h, b"model.0.conv.weight", ptr(w), shp, 4, _lib.OG_DTYPE_F32) == 0
    assert l.og_yolo_set_tensor(h, b"backbone.0.weight", ptr(w), shp, 4, _lib.OG_DTYPE_F32) == -1
    assert l.og_yolo_num_anchors(h, 256, 256) == 1344 and l.og_yolo_num_anchors(h, 250, 256) < 0
    g = np.zeros((1, 256, 256, 3), np.uint8)
    best = np.zeros((1, 5), np.float32)
    assert l.og_yolo_detect_u8(h, ptr(g), 1, 256, 256, 0.25, ptr(best), None) == -2   # not finalized
    l.og_yolo_destroy(h)
