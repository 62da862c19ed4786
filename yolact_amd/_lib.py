"""ctypes loader for libyolact_post.so (the C ABI declared in include/yolact_post.h).

The library is the only compute path of this package.  If it is missing or fails to load, every
entry point raises - there is no eager/PyTorch/CPU fallback.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (YOLACT_POST_LIB: experiment builds of the same sources - tools/ - under another name; never a different backend)
LIB_PATH = os.environ.get("YOLACT_POST_LIB") or os.path.join(HERE, "libyolact_post.so")

YP_NMS_CROSS_CLASS = 0
YP_NMS_PER_CLASS = 1
YP_RESIZE_THEN_THRESHOLD = 0
YP_THRESHOLD_THEN_RESIZE = 1
YP_NUM_KERNELS = 3
YP_STATIC_SCRATCH_BYTES = 64 << 20
KERNEL_NAMES = ("score_filter", "nms", "mask_assemble")

# every symbol include/yolact_post.h declares
EXPORTS = ("yp_create", "yp_destroy", "yp_run", "yp_sync", "yp_last_survivors", "yp_profile_enable",
           "yp_profile_read", "yp_scratch_bytes", "yp_last_error", "yp_version", "yp_make_priors", "yp_run_static",
           "yp_masks_to_rle", "yp_rle_string", "yp_rle_strings", "yp_assemble_masks", "yp_assemble_masks_backward",
           "yp_assemble_masks_backward_workspace", "yp_create_bounded", "yp_overflowed", "yp_masks_to_rle_sized",
           "yp_kept_create", "yp_kept_destroy", "yp_kept_reset", "yp_run_kept")


class YpConfig(ctypes.Structure):
    """struct yp_config (include/yolact_post.h)."""
    _fields_ = [
        ("num_anchors", ctypes.c_int32), ("num_classes", ctypes.c_int32), ("mask_dim", ctypes.c_int32),
        ("proto_h", ctypes.c_int32), ("proto_w", ctypes.c_int32), ("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32),
        ("conf_thresh", ctypes.c_float), ("nms_thresh", ctypes.c_float),
        ("nms_mode", ctypes.c_int32), ("top_k", ctypes.c_int32), ("max_dets", ctypes.c_int32),
        ("use_cropped_mask", ctypes.c_int32), ("zero_pad_masks", ctypes.c_int32), ("mask_threshold", ctypes.c_float),
        ("max_batch", ctypes.c_int32), ("device_id", ctypes.c_int32),
    ]


class YolactPostError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"libyolact_post error {code}: {text}")
        self.code = code


_lib = None


def load():
    """dlopen the in-tree library and declare its prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m yolact_amd.build` (hipcc, gfx950). "
            "yolact_amd has no fallback path.")
    L = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    L.yp_create.restype = ctypes.c_int
    L.yp_create.argtypes = [ctypes.POINTER(YpConfig), ctypes.POINTER(vp)]
    L.yp_destroy.restype = ctypes.c_int
    L.yp_destroy.argtypes = [vp]
    L.yp_run.restype = ctypes.c_int
    L.yp_run.argtypes = [vp, ctypes.c_int32] + [vp] * 5 + [vp] * 7 + [vp]
    L.yp_run_static.restype = ctypes.c_int
    L.yp_run_static.argtypes = [ctypes.POINTER(YpConfig), ctypes.c_int32] + [vp] * 5 + [vp] * 7 + [vp]
    L.yp_kept_create.restype = ctypes.c_int
    L.yp_kept_create.argtypes = [vp, ctypes.POINTER(vp)]
    L.yp_kept_destroy.restype = ctypes.c_int
    L.yp_kept_destroy.argtypes = [vp]
    L.yp_kept_reset.restype = ctypes.c_int
    L.yp_kept_reset.argtypes = [vp]
    L.yp_run_kept.restype = ctypes.c_int
    L.yp_run_kept.argtypes = [vp, vp, ctypes.c_int32] + [vp] * 5 + [vp] * 7 + [vp]
    L.yp_sync.restype = ctypes.c_int
    L.yp_sync.argtypes = [vp]
    L.yp_last_survivors.restype = ctypes.c_int
    L.yp_last_survivors.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
    L.yp_profile_enable.restype = ctypes.c_int
    L.yp_profile_enable.argtypes = [vp, ctypes.c_int32]
    L.yp_profile_read.restype = ctypes.c_int
    L.yp_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.yp_scratch_bytes.restype = ctypes.c_int64
    L.yp_scratch_bytes.argtypes = [vp]
    L.yp_last_error.restype = ctypes.c_char_p
    L.yp_last_error.argtypes = []
    L.yp_make_priors.restype = ctypes.c_int64
    L.yp_make_priors.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                 ctypes.POINTER(ctypes.c_double), ctypes.c_int32, ctypes.POINTER(ctypes.c_double), vp,
                                 ctypes.c_int64, ctypes.c_int32, vp]
    L.yp_masks_to_rle.restype = ctypes.c_int
    L.yp_masks_to_rle.argtypes = [ctypes.c_int32] * 5 + [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_int32,
                                  vp, vp, vp, ctypes.c_int32, vp]
    L.yp_masks_to_rle_sized.restype = ctypes.c_int
    L.yp_masks_to_rle_sized.argtypes = [ctypes.c_int32] * 5 + [vp, vp, vp, ctypes.c_float, ctypes.c_int32, vp, vp, ctypes.c_int32, vp]
    L.yp_assemble_masks.restype = ctypes.c_int
    L.yp_assemble_masks.argtypes = [ctypes.c_int32] * 8 + [vp] * 6 + [vp]
    L.yp_assemble_masks_backward_workspace.restype = ctypes.c_int64
    L.yp_assemble_masks_backward_workspace.argtypes = [ctypes.c_int32] * 3
    L.yp_assemble_masks_backward.restype = ctypes.c_int
    L.yp_assemble_masks_backward.argtypes = [ctypes.c_int32] * 8 + [vp] * 9 + [vp]
    L.yp_rle_strings.restype = ctypes.c_int64
    L.yp_rle_strings.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int64, vp, ctypes.c_int64, vp]
    L.yp_rle_string.restype = ctypes.c_int
    L.yp_rle_string.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]
    L.yp_create_bounded.restype = ctypes.c_int
    L.yp_create_bounded.argtypes = [vp, ctypes.c_int32, vp]
    L.yp_overflowed.restype = ctypes.c_int
    L.yp_overflowed.argtypes = [vp]
    L.yp_version.restype = ctypes.c_int
    L.yp_version.argtypes = []
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise YolactPostError(rc, load().yp_last_error().decode("utf-8", "replace"))
