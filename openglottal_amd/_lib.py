"""ctypes binding of ``libopenglottal_hip.so`` (include/openglottal_hip.h).

cffi is not installed in the build image, so the thin C-ABI layer
the north-star asks for is bound with ctypes; the declarations below are the
exact prototypes of the header.  There is NO fallback: if the shared library is
missing or a call fails, an ``OpenGlottalHipError`` is raised.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPENGLOTTAL_HIP_LIB: load another build of the same C-ABI (A/B builds, an installed copy)
LIB_PATH = os.environ.get("OPENGLOTTAL_HIP_LIB") or os.path.join(_HERE, "libopenglottal_hip.so")

OG_DTYPE_F32, OG_DTYPE_I64 = 0, 1


class OpenGlottalHipError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); the complete export list of openglottal_hip.h
PROTOTYPES = {
    "og_last_error": (C.c_char_p, []),
    "og_version": (C.c_char_p, []),
    "og_device_count": (C.c_int, []),
    "og_init": (C.c_int, [C.c_int]),
    "og_malloc": (C.c_void_p, [C.c_size_t]),
    "og_free": (C.c_int, [C.c_void_p]),
    "og_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "og_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "og_unet_create": (C.c_void_p, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]),
    "og_unet_destroy": (None, [C.c_void_p]),
    "og_unet_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "og_unet_finalize": (C.c_int, [C.c_void_p]),
    "og_unet_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "og_unet_segment_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_unet_stream_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_unet_stream_frames_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_unet_segment_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_unet_segment_crops_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_float, C.c_void_p]),
    "og_unet_segment_crops_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "og_canvas_letterbox_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p]),
    "og_canvas_letterbox_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_int, C.c_void_p]),
    "og_mask_stats_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "og_mask_area_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "og_bgr2gray_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "og_bgr2gray_host": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "og_unet_sync": (C.c_int, [C.c_void_p]),
    "og_unet_stream": (C.c_void_p, [C.c_void_p]),
    "og_unet_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "og_unet_set_chunk": (C.c_int, [C.c_void_p, C.c_int]),
    "og_unet_set_graphs": (C.c_int, [C.c_void_p, C.c_int]),
    "og_unet_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "og_timer_start": (C.c_int, [C.c_void_p]),
    "og_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "og_unet_get_activation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "og_unet_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                  C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "og_unet_clock_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_int)]),
    "og_unet_clock_probe_raw": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "og_unet_flops_per_frame": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "og_unet_plan": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t,
                               C.POINTER(C.c_longlong)]),
    "og_workspace_limit": (C.c_longlong, [C.c_int]),
    "og_yolo_create": (C.c_void_p, [C.c_int]),
    "og_yolo_destroy": (None, [C.c_void_p]),
    "og_yolo_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "og_yolo_finalize": (C.c_int, [C.c_void_p]),
    "og_yolo_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "og_yolo_num_anchors": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "og_yolo_detect_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "og_yolo_detect_u8_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]),
    "og_yolo_detect_u8_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "og_yolo_detect_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "og_yolo_sync": (C.c_int, [C.c_void_p]),
    "og_yolo_get_activation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
}


def lib() -> C.CDLL:
    """Load the HIP library once; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OpenGlottalHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or openglottal_amd/csrc/build.sh.  There is no CPU fallback."
            )
        # PyTorch-ROCm bundles its own HIP/HSA runtime.  If our library (linked against the system ROCm) initialises HIP
        # first, torch.cuda later reports "No HIP GPUs are available"; loaded in the other order, both share torch's copy.
        # So pull torch in first whenever it is installed (the reference depends on it anyway).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise OpenGlottalHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError if a symbol is missing: also loud
            fn.restype = res
            fn.argtypes = args
        ver = (l.og_version() or b"").decode()
        if "store_nop=1" not in ver and os.environ.get("OPENGLOTTAL_HIP_ALLOW_AUDIT_BUILD") != "1":
            # an audit build (-DOG_STORE_NOP=0, tools/epilogue_fence_audit.sh) reproduces a hardware store hazard and computes
            # wrong lanes: never let OPENGLOTTAL_HIP_LIB point the product at one
            raise OpenGlottalHipError(f"{LIB_PATH} reports {ver!r}: not a production build (store_nop=1 missing); refusing to load it")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().og_last_error()
        raise OpenGlottalHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(x) -> C.c_void_p | None:
    """Raw pointer of a numpy array, a torch tensor (host or device), an int, or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)
