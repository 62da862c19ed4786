"""Drop-in `Detect` for anshkumar/yolact, backed by libyolact_post.so (HIP, gfx950).

Mirrors the reference class (``detection.py:6-117``): same constructor signature, same
``__call__(net_outs, img_shape, trad_nms=False, use_cropped_mask=True)``, same five result keys,
shapes and dtypes (``detection.py:116``), plus ``detection_anchor_indices``.  Tensors are
torch-ROCm CUDA tensors; torch is plumbing only (device memory, current stream) - all arithmetic
happens in the HIP kernels behind the C ABI of ``include/yolact_post.h``.  There is no CPU or
eager fallback: without the library or without a GPU the calls raise.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib

NMS_MODES = {"cc": _lib.YP_NMS_CROSS_CLASS, "per_class": _lib.YP_NMS_PER_CLASS}


@dataclass(frozen=True)
class PostConfig:
    num_anchors: int
    num_classes: int          # incl. background
    proto_h: int
    proto_w: int
    img_h: int
    img_w: int
    conf_thresh: float = 0.05
    nms_thresh: float = 0.5
    nms_mode: str = "cc"
    top_k: int = 15
    max_dets: int = 300
    use_cropped_mask: bool = True
    zero_pad_masks: bool = False
    mask_threshold: float = -1.0
    max_batch: int = 1
    device_id: int = 0
    mask_dim: int = 32


class PostProcessor:
    """One ``yp_handle``: the device scratch and launch plan for a fixed shape on one GPU.

    ``run`` is asynchronous on the current torch stream and allocates nothing when ``out`` is
    given.  One PostProcessor serves one stream at a time.
    """

    def __init__(self, cfg: PostConfig, keys_per_anchor: int | None = None):
        """``keys_per_anchor``: size the candidate lists for that many keys per anchor of a work item instead of
        the worst case (yp_create_bounded); ``overflowed()`` then tells whether a batch lost candidates."""
        self.cfg = cfg
        self._L = _lib.load()
        c = _lib.YpConfig(
            num_anchors=cfg.num_anchors, num_classes=cfg.num_classes, mask_dim=cfg.mask_dim,
            proto_h=cfg.proto_h, proto_w=cfg.proto_w, img_h=cfg.img_h, img_w=cfg.img_w,
            conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh, nms_mode=NMS_MODES[cfg.nms_mode],
            top_k=cfg.top_k, max_dets=cfg.max_dets, use_cropped_mask=int(cfg.use_cropped_mask),
            zero_pad_masks=int(cfg.zero_pad_masks), mask_threshold=cfg.mask_threshold,
            max_batch=cfg.max_batch, device_id=cfg.device_id)
        h = ctypes.c_void_p()
        if keys_per_anchor is None:
            _lib.check(self._L.yp_create(ctypes.byref(c), ctypes.byref(h)))
        else:
            _lib.check(self._L.yp_create_bounded(ctypes.byref(c), int(keys_per_anchor), ctypes.byref(h)))
        self._h = h
        self.device = torch.device("cuda", cfg.device_id)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.yp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers -------------------------------------------------------------------------
    def alloc_outputs(self, batch: int, masks: bool = True, bitmaps: bool = False):
        c, dev = self.cfg, self.device
        out = {
            "detection_boxes": torch.empty((batch, c.max_dets, 4), dtype=torch.float32, device=dev),
            "detection_classes": torch.empty((batch, c.max_dets), dtype=torch.float32, device=dev),
            "detection_scores": torch.empty((batch, c.max_dets), dtype=torch.float32, device=dev),
            "num_detections": torch.empty((batch,), dtype=torch.int32, device=dev),
            "detection_anchor_indices": torch.empty((batch, c.max_dets), dtype=torch.int32, device=dev),
        }
        if masks:
            out["detection_masks"] = torch.empty((batch, c.max_dets, c.proto_h, c.proto_w), dtype=torch.float32, device=dev)
        if bitmaps:
            out["detection_bitmaps"] = torch.empty((batch, c.max_dets, c.proto_h, c.proto_w), dtype=torch.uint8, device=dev)
        return OutputSet(out)

    def alloc_kept(self, batch: int, bitmaps: bool = False):
        """An output set the library keeps track of (yp_kept_create / yp_run_kept): a dict like alloc_outputs',
        plus the record of what the last run wrote into its masks.  `run(..., out=that dict)` then clears only the
        previous run's crop boxes instead of refilling every row; same results.  Always pass the same dict, with the
        same batch size, and do not write into its mask tensors yourself (``kept.reset()`` after you did)."""
        out = KeptOutputs(self.alloc_outputs(batch, masks=True, bitmaps=bitmaps))
        k = ctypes.c_void_p()
        _lib.check(self._L.yp_kept_create(self._h, ctypes.byref(k)))
        out._bind(self._L, k)
        return out

    def _check(self, t: torch.Tensor, shape, name: str, dtype=torch.float32):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device != self.device:
            raise ValueError(f"{name} must be a CUDA tensor on {self.device}")
        if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous {dtype} of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")
        return t.data_ptr()

    def run(self, pred_offset, pred_cls, pred_mask_coef, priors, proto, out=None):
        c = self.cfg
        B = int(pred_cls.shape[0])
        fresh = out is None
        if fresh:
            out = self.alloc_outputs(B, masks=proto is not None, bitmaps=c.mask_threshold >= 0 and proto is not None)
        A, C, K = c.num_anchors, c.num_classes, c.mask_dim
        p_off = self._check(pred_offset, (B, A, 4), "pred_offset")
        p_cls = self._check(pred_cls, (B, A, C), "pred_cls")
        p_coef = self._check(pred_mask_coef, (B, A, K), "pred_mask_coef")
        p_pri = self._check(priors, (A, 4), "priors")
        md = c.max_dets
        masks = out.get("detection_masks")
        p_proto = self._check(proto, (B, c.proto_h, c.proto_w, K), "proto_out") if masks is not None else None
        # An output set of ours that has been checked before, with the same tensors in it, is not checked again: at
        # batch 1 the seven checks cost more host time than the kernels take on the GPU.
        cache = getattr(out, "_checked", None)
        if cache is not None and cache[0] == B and all(out.get(k) is t for k, t in zip(_OUT_KEYS, cache[1])):
            p_boxes, p_clsout, p_scores, p_num, p_aidx, p_masks, p_bits = cache[2]
        else:
            bitmaps = out.get("detection_bitmaps")
            p_boxes = self._check(out["detection_boxes"], (B, md, 4), "detection_boxes")
            p_clsout = self._check(out["detection_classes"], (B, md), "detection_classes")
            p_scores = self._check(out["detection_scores"], (B, md), "detection_scores")
            p_num = self._check(out["num_detections"], (B,), "num_detections", torch.int32)
            aidx = out.get("detection_anchor_indices")
            p_aidx = self._check(aidx, (B, md), "detection_anchor_indices", torch.int32) if aidx is not None else None
            p_masks = self._check(masks, (B, md, c.proto_h, c.proto_w), "detection_masks") if masks is not None else None
            p_bits = self._check(bitmaps, (B, md, c.proto_h, c.proto_w), "detection_bitmaps", torch.uint8) if bitmaps is not None else None
            if not fresh and isinstance(out, OutputSet):
                out._checked = (B, tuple(out.get(k) for k in _OUT_KEYS), (p_boxes, p_clsout, p_scores, p_num, p_aidx, p_masks, p_bits))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if isinstance(out, KeptOutputs):
            _lib.check(self._L.yp_run_kept(self._h, out._k, B, p_off, p_cls, p_coef, p_pri, p_proto, p_boxes, p_clsout,
                                           p_scores, p_masks, p_bits, p_num, p_aidx, ctypes.c_void_p(stream)))
        else:
            _lib.check(self._L.yp_run(self._h, B, p_off, p_cls, p_coef, p_pri, p_proto, p_boxes, p_clsout, p_scores,
                                      p_masks, p_bits, p_num, p_aidx, ctypes.c_void_p(stream)))
        return out

    def sync(self):
        _lib.check(self._L.yp_sync(self._h))

    def overflowed(self) -> bool:
        """Synchronises; True if a candidate list overflowed since the last call (bounded lists only)."""
        rc = int(self._L.yp_overflowed(self._h))
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    def last_survivors(self, batch: int) -> np.ndarray:
        buf = (ctypes.c_int32 * batch)()
        _lib.check(self._L.yp_last_survivors(self._h, buf, batch))
        return np.frombuffer(buf, dtype=np.int32).copy()

    def profile(self, on: bool = True):
        _lib.check(self._L.yp_profile_enable(self._h, int(on)))

    def kernel_times_ms(self) -> dict:
        buf = (ctypes.c_float * _lib.YP_NUM_KERNELS)()
        _lib.check(self._L.yp_profile_read(self._h, buf))
        return dict(zip(_lib.KERNEL_NAMES, [float(x) for x in buf]))

    @property
    def scratch_bytes(self) -> int:
        return int(self._L.yp_scratch_bytes(self._h))


_OUT_KEYS = ("detection_boxes", "detection_classes", "detection_scores", "num_detections", "detection_anchor_indices",
             "detection_masks", "detection_bitmaps")


class OutputSet(dict):
    """The output tensors of PostProcessor.alloc_outputs: a dict that remembers having been checked (run())."""
    _checked = None


class KeptOutputs(OutputSet):
    """The output tensors of PostProcessor.alloc_kept together with the library's record of their masks' contents."""
    _L = None
    _k = None

    def _bind(self, L, k):
        self._L, self._k = L, k

    def reset(self):
        """Forget the record (after writing into the mask tensors yourself): the next run refills them."""
        _lib.check(self._L.yp_kept_reset(self._k))

    def close(self):
        if self._k is not None and self._k.value:
            self._L.yp_kept_destroy(self._k)
            self._k = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _as_device_tensor(x, device):
    if isinstance(x, torch.Tensor):
        if x.dtype is torch.float32 and x.device == device:      # (the usual case: no dispatcher round trip)
            return x if x.is_contiguous() else x.contiguous()
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


class Detect(object):
    """Same interface as the reference ``Detect`` (detection.py:6).

    By default (``strict_reference=True``) it reproduces the reference as shipped, quirks
    included: constructor ``max_output_size`` / ``per_class_max_output_size`` are ignored and
    replaced by 300 / 100 (detection.py:21-22), NMS is the cross-class Fast NMS with
    ``iou_threshold=0.5, top_k=15`` that ``__call__`` invokes at detection.py:83 (``nms_thresh`` is
    validated but not forwarded, exactly as there), and every output has 300 zero-padded rows.

    ``strict_reference=False`` unlocks the knobs the reference hard-codes: ``nms_mode``
    ("cc" | "per_class"), ``top_k``, ``max_dets`` rows (compact, un-padded masks unless
    ``zero_pad_masks``), ``nms_thresh`` forwarded as the IoU threshold, ``mask_threshold`` for an
    extra uint8 ``detection_bitmaps`` output (the callers' ``mask > 0.5``, train.py:497).

    ``reuse_outputs=True`` (either mode; opt-in because it changes the lifetime of the results, not their
    values): the 300 zero-padded rows cost 23 MB of zeros per image and call, of which at most ``top_k`` rows can
    ever hold anything else - and inside those, a few per cent of the pixels.  With this option the object keeps
    ONE set of output tensors per input shape (the last four shapes) whose contents the library tracks (``yp_run_kept``): the first call
    zero-fills it, every later call clears exactly the crop boxes the call before wrote and writes its own.  The
    returned tensors are that set and are overwritten by the next call with the same shapes - clone what must
    outlive it, and do not write into the masks.
    """

    def __init__(self, num_classes, max_output_size, per_class_max_output_size, conf_thresh, nms_thresh, *,
                 strict_reference: bool = True, nms_mode: str = "cc", top_k: int = 15, max_dets: int | None = None,
                 zero_pad_masks: bool | None = None, mask_threshold: float | None = None, device=None,
                 reuse_outputs: bool = False):
        self.num_classes = num_classes
        self.nms_thresh = nms_thresh
        if nms_thresh <= 0:
            raise ValueError('nms_threshold must be non negative.')       # detection.py:17-18
        self.conf_thresh = conf_thresh
        self.use_fast_nms = False
        self.strict_reference = strict_reference
        if strict_reference:
            self.max_output_size = 300                                      # detection.py:21
            self.per_class_max_output_size = 100                            # detection.py:22
            self.nms_mode, self.top_k, self.iou_thresh = "cc", 15, 0.5     # detection.py:83,181
            self.rows, self.zero_pad_masks = 300, True
            self.mask_threshold = -1.0
        else:
            if nms_mode not in NMS_MODES:
                raise ValueError(f"nms_mode must be one of {sorted(NMS_MODES)}")
            self.max_output_size = max_output_size
            self.per_class_max_output_size = per_class_max_output_size
            self.nms_mode, self.top_k, self.iou_thresh = nms_mode, top_k, nms_thresh
            self.rows = max_dets if max_dets is not None else max_output_size
            self.zero_pad_masks = bool(zero_pad_masks) if zero_pad_masks is not None else False
            self.mask_threshold = -1.0 if mask_threshold is None else float(mask_threshold)
        self.device = torch.device(device) if device is not None else None
        self.reuse_outputs = bool(reuse_outputs)
        self._pp = {}
        self._outs = {}       # reuse_outputs: (processor key, batch) -> the kept output set

    def _processor(self, key, build):
        pp = self._pp.get(key)
        if pp is None:
            pp = build()
            self._pp[key] = pp
        return pp

    def __call__(self, net_outs, img_shape, trad_nms=False, use_cropped_mask=True):
        if trad_nms:
            # detection.py:85 -> tf.image.non_max_suppression: TF's C++ kernel, out of scope
            raise NotImplementedError("trad_nms=True (tf.image.non_max_suppression) is not part of this path")
        if not torch.cuda.is_available():
            raise RuntimeError("yolact_amd.Detect needs a HIP device; there is no CPU fallback")
        dev = self.device
        if dev is None:
            t0 = net_outs['pred_cls']
            dev = t0.device if isinstance(t0, torch.Tensor) and t0.is_cuda else torch.device("cuda", torch.cuda.current_device())
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        box_p = _as_device_tensor(net_outs['pred_offset'], dev)            # detection.py:44-48
        class_p = _as_device_tensor(net_outs['pred_cls'], dev)
        coef_p = _as_device_tensor(net_outs['pred_mask_coef'], dev)
        anchors = _as_device_tensor(net_outs['priors'], dev)
        proto_p = _as_device_tensor(net_outs['proto_out'], dev)
        B, A, C = class_p.shape
        proto_h, proto_w = int(proto_p.shape[1]), int(proto_p.shape[2])    # detection.py:50-51
        shp = img_shape.tolist() if hasattr(img_shape, "tolist") else list(img_shape)
        H, W = int(shp[1]), int(shp[2])
        key = (dev.index, A, C, proto_h, proto_w, H, W, bool(use_cropped_mask))
        pp = self._pp.get(key)
        if pp is None or pp.cfg.max_batch < B:
            if pp is not None:
                pp.close()
            self._outs = {k: v for k, v in self._outs.items() if k[0] != key}
            # (the library restores the caller's current device in every entry point, destroy included)
            pp = PostProcessor(PostConfig(
                num_anchors=A, num_classes=C, proto_h=proto_h, proto_w=proto_w, img_h=H, img_w=W,
                conf_thresh=float(self.conf_thresh), nms_thresh=float(self.iou_thresh), nms_mode=self.nms_mode,
                top_k=int(self.top_k), max_dets=int(self.rows), use_cropped_mask=bool(use_cropped_mask),
                zero_pad_masks=self.zero_pad_masks, mask_threshold=self.mask_threshold,
                max_batch=int(B), device_id=dev.index))
            self._pp[key] = pp
        with torch.cuda.device(dev):
            kept = None
            if self.reuse_outputs:
                kept = self._outs.get((key, B))
                if kept is None:          # (its first run zero-fills every row, the padding rows included)
                    while len(self._outs) >= 4:                         # a set is a whole batch of masks: keep a few
                        self._outs.pop(next(iter(self._outs))).close()
                    kept = pp.alloc_kept(B, bitmaps=self.mask_threshold >= 0)
                    self._outs[(key, B)] = kept
            out = pp.run(box_p, class_p, coef_p, anchors, proto_p, out=kept)
        result = {'detection_boxes': out['detection_boxes'], 'detection_classes': out['detection_classes'],
                  'detection_scores': out['detection_scores'], 'detection_masks': out['detection_masks'],
                  'num_detections': out['num_detections'],
                  'detection_anchor_indices': out['detection_anchor_indices']}
        if 'detection_bitmaps' in out:
            result['detection_bitmaps'] = out['detection_bitmaps']
        return result
