"""ctypes binding of oracle/liboracle_detect.so (TEST INFRASTRUCTURE, see detect_oracle.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_detect.so")
_lib = None

NMS_MODES = {"cc": 0, "per_class": 1}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "detect_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_detect.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        L.yo_detect.restype = ctypes.c_int
        L.yo_detect.argtypes = [ctypes.c_int] * 8 + [ctypes.c_float] * 2 + [ctypes.c_int] * 4 + [fp] * 5 + [fp] * 4 + [ip] * 3 + [ctypes.c_int]
        L.yo_exp.restype = ctypes.c_float
        L.yo_exp.argtypes = [ctypes.c_float]
        L.yo_sigmoid.restype = ctypes.c_float
        L.yo_sigmoid.argtypes = [ctypes.c_float]
        L.yo_iou.restype = ctypes.c_float
        L.yo_iou.argtypes = [fp, fp]
        L.yo_decode.restype = None
        L.yo_decode.argtypes = [fp, fp, fp]
        L.yo_softmax_row.restype = None
        L.yo_softmax_row.argtypes = [fp, ctypes.c_int, fp]
        L.yo_loss_masks.restype = None
        L.yo_loss_masks.argtypes = [ctypes.c_int] * 7 + [fp] * 4
        L.yo_loss_masks_backward.restype = None
        L.yo_loss_masks_backward.argtypes = [ctypes.c_int] * 4 + [fp] * 6
        L.yo_cpu_ok.restype = ctypes.c_int
        if not L.yo_cpu_ok():
            raise RuntimeError("oracle library needs a CPU with FMA")
        _lib = L
    return _lib


def _p(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def detect(net_outs, img_shape, conf_thresh=0.05, nms_thresh=0.5, top_k=15, nms_mode="cc", max_dets=None,
           use_cropped_mask=True, return_masks=True, nthreads=1):
    """Same contract as oracle.detect_oracle.detect but compact rows: [B,max_dets,...]."""
    L = lib()
    f32 = lambda k: np.ascontiguousarray(net_outs[k], dtype=np.float32)
    off, cls, coef, priors, proto = f32("pred_offset"), f32("pred_cls"), f32("pred_mask_coef"), f32("priors"), f32("proto_out")
    B, A, C = cls.shape
    K = coef.shape[2]
    ph, pw = proto.shape[1], proto.shape[2]
    if max_dets is None:
        max_dets = top_k if nms_mode == "cc" else 100
    H, W = int(img_shape[1]), int(img_shape[2])
    boxes = np.zeros((B, max_dets, 4), np.float32)
    classes = np.zeros((B, max_dets), np.float32)
    scores = np.zeros((B, max_dets), np.float32)
    masks = np.zeros((B, max_dets, ph, pw), np.float32) if return_masks else None
    num = np.zeros((B,), np.int32)
    aidx = np.full((B, max_dets), -1, np.int32)
    nsurv = np.zeros((B,), np.int32)
    rc = L.yo_detect(B, A, C, K, ph, pw, H, W, conf_thresh, nms_thresh, NMS_MODES[nms_mode], top_k, max_dets,
                     1 if use_cropped_mask else 0, _p(off), _p(cls), _p(coef), _p(priors), _p(proto), _p(boxes),
                     _p(classes), _p(scores), _p(masks) if masks is not None else None, _p(num, ctypes.c_int32),
                     _p(aidx, ctypes.c_int32), _p(nsurv, ctypes.c_int32), nthreads)
    if rc != 0:
        raise RuntimeError(f"yo_detect failed: {rc}")
    return {"detection_boxes": boxes, "detection_classes": classes, "detection_scores": scores,
            "detection_masks": masks, "num_detections": num, "detection_anchor_indices": aidx,
            "num_survivors": nsurv}


def exp(x: float) -> float:
    return float(lib().yo_exp(ctypes.c_float(x)))


def loss_masks(proto, pos_coef, pos_box, img_h, img_w, use_cropped_mask=True):
    """loss/loss_yolact.py:239-253 for one image: proto [ph,pw,K], pos_coef [n,K], pos_box [n,4] pixels yxyx
    -> [n,ph,pw] float32."""
    L = lib()
    proto = np.ascontiguousarray(proto, np.float32); pos_coef = np.ascontiguousarray(pos_coef, np.float32)
    pos_box = np.ascontiguousarray(pos_box, np.float32)
    ph, pw, K = proto.shape
    n = pos_coef.shape[0]
    out = np.zeros((n, ph, pw), np.float32)
    L.yo_loss_masks(n, K, ph, pw, int(img_h), int(img_w), 1 if use_cropped_mask else 0, _p(proto), _p(pos_coef), _p(pos_box), _p(out))
    return out


def loss_masks_backward(proto, pos_coef, masks, grad_masks):
    """Gradients of loss/loss_yolact.py:239-253 for one image, in the HIP kernels' summation order:
    proto [ph,pw,K], pos_coef [n,K], masks / grad_masks [n,ph,pw] -> (grad_proto [ph,pw,K], grad_coef [n,K])."""
    L = lib()
    proto = np.ascontiguousarray(proto, np.float32); pos_coef = np.ascontiguousarray(pos_coef, np.float32)
    masks = np.ascontiguousarray(masks, np.float32); grad_masks = np.ascontiguousarray(grad_masks, np.float32)
    ph, pw, K = proto.shape
    n = pos_coef.shape[0]
    gp = np.zeros((ph, pw, K), np.float32)
    gc = np.zeros((n, K), np.float32)
    L.yo_loss_masks_backward(n, K, ph, pw, _p(proto), _p(pos_coef), _p(masks), _p(grad_masks), _p(gp), _p(gc))
    return gp, gc
