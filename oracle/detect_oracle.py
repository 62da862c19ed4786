"""CPU oracle for YOLACT post-processing (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The shipped path (``yolact_amd``) never does.

This is an op-for-op numpy restatement of the reference's ``Detect`` layer
(``/root/reference/detection.py``) and of the helpers it calls in
``/root/reference/utils/utils.py``.  Every function names the reference lines it
follows.  The restatement is serial over images exactly like the reference
(``detection.py:71``).

PARITY UNPINNED: TensorFlow is not installable in the build container (ordinary
``ModuleNotFoundError``) and the reference ships no golden
vectors, known-answer tests or fixtures for this path.  The TF
op semantics restated here (documented TF behaviour, not captured outputs):

* ``tf.nn.softmax``      -> exp(x - max) * (1 / sum)      (detection.py:56; Eigen's form)
* ``tf.boolean_mask``    -> ascending index order kept    (detection.py:73-76)
* ``tf.math.top_k``      -> descending, ties: lower index first (detection.py:186)
* ``tf.argmax``          -> first maximum                 (detection.py:184)
* ``tf.linalg.band_part``-> strict upper triangle via (0,-1) - (0,0) (detection.py:193)
* ``tf.clip_by_value``   -> min(max(t, lo), hi)

``dtype`` selects the arithmetic type: ``np.float32`` restates the reference,
``np.float64`` is used by the tests to compute decision margins.

Besides the reference's five output tensors the oracle returns
``detection_anchor_indices`` (original anchor index of every kept detection;
-1 padded), which the reference never exposes but BASELINE.json grades.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# TF op semantics
# --------------------------------------------------------------------------


def softmax_lastaxis(x: np.ndarray) -> np.ndarray:
    """tf.nn.softmax(x, axis=-1) as called at detection.py:56.  TF's CPU kernel (Eigen,
    SoftmaxEigenImpl) computes exp(x - max) and multiplies by the RECIPROCAL of the sum
    (SURVEY.md 8c): one rounding more than a true division, which is exactly the 1-ulp class of
    difference that reorders saturated near-ties - so the stand-in for TF's arithmetic uses TF's form."""
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    one = x.dtype.type(1.0)
    return e * (one / np.sum(e, axis=-1, keepdims=True, dtype=x.dtype))


def top_k_indices(scores: np.ndarray, k: int) -> np.ndarray:
    """Indices tf.math.top_k(scores, k) returns: descending, ties -> lower index."""
    order = np.argsort(-scores, kind="stable")
    return order[:k]


def clip_by_value(t, lo, hi):
    return np.minimum(np.maximum(t, lo), hi)


# --------------------------------------------------------------------------
# utils/utils.py helpers
# --------------------------------------------------------------------------


def _area(boxes: np.ndarray) -> np.ndarray:
    """utils/utils.py:7-21."""
    y_min, x_min, y_max, x_max = (boxes[:, i] for i in range(4))
    return (y_max - y_min) * (x_max - x_min)


def _intersection(b1: np.ndarray, b2: np.ndarray) -> np.ndarray:
    """utils/utils.py:23-48.  [n,4],[m,4] -> [n,m]."""
    dt = b1.dtype
    y_min1, x_min1, y_max1, x_max1 = (b1[:, i : i + 1] for i in range(4))
    y_min2, x_min2, y_max2, x_max2 = (b2[:, i : i + 1] for i in range(4))
    all_pairs_min_ymax = np.minimum(y_max1, y_max2.T)
    all_pairs_max_ymin = np.maximum(y_min1, y_min2.T)
    ih = np.maximum(dt.type(0.0), all_pairs_min_ymax - all_pairs_max_ymin)
    all_pairs_min_xmax = np.minimum(x_max1, x_max2.T)
    all_pairs_max_xmin = np.maximum(x_min1, x_min2.T)
    iw = np.maximum(dt.type(0.0), all_pairs_min_xmax - all_pairs_max_xmin)
    return ih * iw


def _iou(b1: np.ndarray, b2: np.ndarray) -> np.ndarray:
    """utils/utils.py:50-70."""
    inter = _intersection(b1, b2)
    a1 = _area(b1)
    a2 = _area(b2)
    unions = (a1[:, None] + a2[None, :]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        q = inter / unions
    return np.where(inter == 0.0, np.zeros_like(inter), q)


def sanitize_coordinates(_x1, _x2, img_size, normalized, padding=0.0):
    """utils/utils.py:89-104."""
    dt = _x1.dtype
    img_size = dt.type(img_size)
    padding = dt.type(padding)
    if normalized:
        _x1 = _x1 * img_size
        _x2 = _x2 * img_size
    x1 = np.minimum(_x1, _x2)
    x2 = np.maximum(_x1, _x2)
    x1 = clip_by_value(x1 - padding, dt.type(0.0), img_size)
    x2 = clip_by_value(x2 + padding, dt.type(0.0), img_size)
    return x1, x2


def crop_bounds(boxes: np.ndarray, mask_h: int, mask_w: int):
    """The four per-box edges utils.crop computes at utils/utils.py:116-117.

    The call there passes ``(…, img_size, padding, normalized)`` positionally
    into ``sanitize_coordinates(_x1, _x2, img_size, normalized, padding)``, so
    inside ``normalized = 1`` (truthy: coordinates are scaled by the mask size)
    and ``padding = True``, which arithmetic turns into 1.0.
    Returns (x1, x2, y1, y2) in proto pixels.
    """
    x1, x2 = sanitize_coordinates(boxes[:, 1], boxes[:, 3], float(mask_w), normalized=True, padding=1.0)
    y1, y2 = sanitize_coordinates(boxes[:, 0], boxes[:, 2], float(mask_h), normalized=True, padding=1.0)
    return x1, x2, y1, y2


def crop(mask_p: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """utils/utils.py:108-133.  mask_p [h,w,n], boxes [n,4] normalised yxyx."""
    dt = mask_p.dtype
    h, w, _ = mask_p.shape
    x1, x2, y1, y2 = crop_bounds(boxes, h, w)
    rows = np.arange(h, dtype=dt).reshape(-1, 1, 1)
    cols = np.arange(w, dtype=dt).reshape(1, -1, 1)
    mask_left = cols >= x1.reshape(1, 1, -1)
    mask_right = cols < x2.reshape(1, 1, -1)
    mask_bottom = rows >= y1.reshape(1, 1, -1)
    mask_top = rows < y2.reshape(1, 1, -1)
    crop_mask = (mask_left & mask_right) & (mask_bottom & mask_top)
    return mask_p * crop_mask.astype(dt)


def _decode(box_p: np.ndarray, priors: np.ndarray) -> np.ndarray:
    """utils/utils.py:173-213 (include_variances=True)."""
    dt = box_p.dtype
    v0, v1 = dt.type(0.1), dt.type(0.2)
    two = dt.type(2)
    ph = priors[:, 2] - priors[:, 0]
    pw = priors[:, 3] - priors[:, 1]
    pri = np.stack([priors[:, 1] + (pw / two), priors[:, 0] + (ph / two), pw, ph], axis=-1)
    b_x_y = pri[:, :2] + box_p[:, :2] * pri[:, 2:] * v0
    b_w_h = pri[:, 2:] * np.exp(box_p[:, 2:] * v1)
    boxes = np.concatenate([b_x_y, b_w_h], axis=-1)
    boxes = np.concatenate([boxes[:, :2] - boxes[:, 2:] / two, boxes[:, 2:] / two + boxes[:, :2]], axis=-1)
    return np.stack([boxes[:, 1], boxes[:, 0], boxes[:, 3], boxes[:, 2]], axis=-1)


# --------------------------------------------------------------------------
# detection.py
# --------------------------------------------------------------------------


def _sanitize(boxes: np.ndarray, width, height) -> np.ndarray:
    """Detect._sanitize, detection.py:119-131."""
    x1, x2 = sanitize_coordinates(boxes[:, 1], boxes[:, 3], float(width), normalized=False)
    y1, y2 = sanitize_coordinates(boxes[:, 0], boxes[:, 2], float(height), normalized=False)
    return np.stack((y1, x1, y2, x2), axis=1)


def cc_fast_nms(boxes, masks, scores, iou_threshold=0.5, top_k=15):
    """Detect._cc_fast_nms, detection.py:181-209.

    Returns (boxes, masks, classes, scores, idx_out) where idx_out indexes the
    conf-filtered survivor list (extra, for anchor bookkeeping).
    """
    dt = boxes.dtype
    classes = np.argmax(scores, axis=-1) + 1
    scores = np.max(scores, axis=-1)
    k = min(top_k, scores.shape[0])
    idx = top_k_indices(scores, k)
    boxes_idx = boxes[idx]
    iou = _iou(boxes_idx, boxes_idx)
    iou = np.triu(iou, 0) - np.diag(np.diag(iou))  # band_part(0,-1) - band_part(0,0)
    iou_max = np.max(iou, axis=0)
    idx_out = idx[iou_max <= dt.type(iou_threshold)]
    classes = classes[idx_out].astype(dt)
    return boxes[idx_out], masks[idx_out], classes, scores[idx_out], idx_out


def per_class_fast_nms(boxes, masks, scores, conf_thresh, iou_threshold=0.5, top_k=200, max_num_detection=100):
    """Intent of the dead ``Detect._fast_nms`` (detection.py:211-250), its three
    defects repaired: ``self.conf_threshold`` -> conf_thresh,
    ``self.max_num_detection`` -> argument, per-class rank-2 IoU instead of the
    rank-3 call, ``k = min(top_k, S)``.  Class ids are returned 1-based so that
    0 keeps meaning "empty slot" in detection_classes (detection.py:66); the
    dead code's ``tf.range(num_classes)`` (L235) would be 0-based.

    SPEC-DERIVED, PARITY UNPINNED (the reference never runs this code).
    """
    dt = boxes.dtype
    S, cfg = scores.shape
    k = min(top_k, S)
    sc_t = scores.T  # [cfg, S]                                     L212-216
    kept_scores, kept_cls, kept_idx = [], [], []
    for c in range(cfg):
        idx = top_k_indices(sc_t[c], k)  # L217
        s_sorted = sc_t[c][idx]
        b = boxes[idx]  # L219
        iou = _iou(b, b)  # L221 (per class)
        iou = np.triu(iou, 0) - np.diag(np.diag(iou))  # L223-225
        iou_max = np.max(iou, axis=0)  # L228: max over higher-scoring boxes
        keep = (iou_max <= dt.type(iou_threshold)) & (s_sorted > dt.type(conf_thresh))  # L229-234
        kept_scores.append(s_sorted[keep])
        kept_cls.append(np.full(int(keep.sum()), c + 1, dtype=np.int64))
        kept_idx.append(idx[keep])
    f_scores = np.concatenate(kept_scores) if kept_scores else np.zeros((0,), dt)  # class-major, L234-239
    f_cls = np.concatenate(kept_cls) if kept_cls else np.zeros((0,), np.int64)
    f_idx = np.concatenate(kept_idx) if kept_idx else np.zeros((0,), np.int64)
    n = min(max_num_detection, f_scores.shape[0])  # L242
    sel = top_k_indices(f_scores, n)  # L243
    idx_out = f_idx[sel]
    return boxes[idx_out], masks[idx_out], f_cls[sel].astype(dt), f_scores[sel], idx_out


def detect(
    net_outs: dict,
    img_shape,
    conf_thresh: float = 0.05,
    nms_thresh: float = 0.5,
    top_k: int = 15,
    nms_mode: str = "cc",
    max_output_size: int = 300,
    max_dets: int | None = None,
    use_cropped_mask: bool = True,
    dtype=np.float32,
    return_masks: bool = True,
):
    """Detect.__call__, detection.py:24-117.

    ``nms_mode='cc'`` is the live path (``_cc_fast_nms`` called at L83 with its
    defaults iou_threshold=0.5, top_k=15; ``nms_thresh`` is NOT forwarded by the
    reference, the argument here is the iou threshold so that other values can be
    exercised).  ``max_dets`` (default: no truncation) keeps the first rows of
    the score-descending result, which is how a caller slices
    ``[0][:det_num]`` (train.py:492-500).  ``img_shape`` is ``[B,H,W,3]``.
    """
    dt = np.dtype(dtype)
    box_p = np.asarray(net_outs["pred_offset"], dtype=dt)
    class_p = np.asarray(net_outs["pred_cls"], dtype=dt)
    coef_p = np.asarray(net_outs["pred_mask_coef"], dtype=dt)
    anchors = np.asarray(net_outs["priors"], dtype=dt)
    proto_p = np.asarray(net_outs["proto_out"], dtype=dt)

    proto_h, proto_w = proto_p.shape[1], proto_p.shape[2]
    H, W = int(img_shape[1]), int(img_shape[2])

    class_p = softmax_lastaxis(class_p)  # L56
    class_p = class_p[:, :, 1:]  # L59
    class_p_max = np.max(class_p, axis=-1)  # L62
    batch_size = class_p_max.shape[0]
    rows = max_output_size

    detection_boxes = np.zeros((batch_size, rows, 4), dt)  # L65-69
    detection_classes = np.zeros((batch_size, rows), dt)
    detection_scores = np.zeros((batch_size, rows), dt)
    detection_masks = np.zeros((batch_size, rows, proto_h, proto_w), dt) if return_masks else None
    num_detections = np.zeros((batch_size,), np.int32)
    anchor_indices = np.full((batch_size, rows), -1, np.int32)
    num_survivors = np.zeros((batch_size,), np.int32)

    for b in range(batch_size):  # L71
        m = class_p_max[b] > dt.type(conf_thresh)  # L73-76 (strict >)
        surv = np.nonzero(m)[0]
        num_survivors[b] = surv.shape[0]
        class_thre = class_p[b][m]
        coef_thre = coef_p[b][m]
        raw_boxes = box_p[b][m]
        raw_anchors = anchors[m]
        boxes = _decode(raw_boxes, raw_anchors)  # L79
        if class_thre.size > 0:  # L81
            if nms_mode == "cc":
                boxes, coef_thre, class_ids, scores, idx_out = cc_fast_nms(
                    boxes, coef_thre, class_thre, iou_threshold=nms_thresh, top_k=top_k)  # L83
            elif nms_mode == "per_class":
                boxes, coef_thre, class_ids, scores, idx_out = per_class_fast_nms(
                    boxes, coef_thre, class_thre, conf_thresh, iou_threshold=nms_thresh, top_k=top_k,
                    max_num_detection=(max_dets if max_dets is not None else rows))
            else:
                raise ValueError(nms_mode)
            if max_dets is not None:
                boxes, coef_thre, class_ids, scores, idx_out = (
                    boxes[:max_dets], coef_thre[:max_dets], class_ids[:max_dets], scores[:max_dets], idx_out[:max_dets])
            n = boxes.shape[0]  # L87
            if n > rows:
                raise ValueError("more detections than output rows")

            boxes = _sanitize(boxes, width=W, height=H)  # L92
            boxes = np.stack([boxes[:, 0] / dt.type(H), boxes[:, 1] / dt.type(W),
                              boxes[:, 2] / dt.type(H), boxes[:, 3] / dt.type(W)], axis=-1)  # L93-98
            if return_masks:
                masks = proto_p[b] @ coef_thre.T  # L89
                masks = 1.0 / (1.0 + np.exp(-masks))  # L90 tf.sigmoid
                masks = masks.astype(dt)
                if use_cropped_mask:
                    masks = crop(masks, boxes)  # L102-103
                masks = clip_by_value(masks, dt.type(0.0), dt.type(1.0))  # L105
                masks = np.transpose(masks, (2, 0, 1))  # L107
                detection_masks[b, :n] = masks  # L113
            detection_boxes[b, :n] = boxes  # L110
            detection_classes[b, :n] = class_ids  # L111
            detection_scores[b, :n] = scores  # L112
            num_detections[b] = n  # L114
            anchor_indices[b, :n] = surv[idx_out]

    return {
        "detection_boxes": detection_boxes,
        "detection_classes": detection_classes,
        "detection_scores": detection_scores,
        "detection_masks": detection_masks,
        "num_detections": num_detections,
        "detection_anchor_indices": anchor_indices,
        "num_survivors": num_survivors,
    }


def loss_mask_assembly(proto: np.ndarray, pos_coef: np.ndarray, pos_prior_box: np.ndarray, img_h, img_w,
                       use_cropped_mask: bool = True) -> np.ndarray:
    """The forward mask assembly of the training loss, loss/loss_yolact.py:239-253, for one image:
    proto [ph,pw,K], pos_coef [n,K], pos_prior_box [n,4] yxyx pixels -> [n,ph,pw] (L273's transpose)."""
    dt = proto.dtype
    mask_p = proto @ pos_coef.T                                                   # L239-240
    mask_p = (1.0 / (1.0 + np.exp(-mask_p))).astype(dt)                           # L241
    if use_cropped_mask:
        b = pos_prior_box
        boxes = np.stack([b[:, 0] / dt.type(img_h), b[:, 1] / dt.type(img_w),
                          b[:, 2] / dt.type(img_h), b[:, 3] / dt.type(img_w)], axis=-1)   # L247-252
        mask_p = crop(mask_p, boxes)                                              # L253
    return np.transpose(mask_p, (2, 0, 1))


def loss_mask_assembly_backward(proto: np.ndarray, pos_coef: np.ndarray, masks: np.ndarray, grad_masks: np.ndarray):
    """Gradients of loss_mask_assembly w.r.t. proto and pos_coef given dL/d mask_p [n,ph,pw].  The reference
    leaves this to TensorFlow's autodiff of loss/loss_yolact.py:239-253: the crop passes the gradient inside
    the box only, SigmoidGrad = dy*y*(1-y), matmul: dA = dC B, dB = A^T dC.  With the cropped forward output
    `masks` (0 outside the box) that is dz = g*m*(1-m).  Plain matrix products in the arrays' dtype (pass
    float64 for a tolerance check; the summation order of the HIP kernels is restated in detect_oracle.c)."""
    dz = grad_masks * masks * (1.0 - masks)                                        # [n,ph,pw]
    grad_proto = np.einsum('nyx,nk->yxk', dz, pos_coef)
    grad_coef = np.einsum('nyx,yxk->nk', dz, proto)
    return grad_proto, grad_coef


# --------------------------------------------------------------------------
# data/anchor.py (produces `priors`)
# --------------------------------------------------------------------------


def generate_anchors(img_h, img_w, feature_map_size, aspect_ratio, scale) -> np.ndarray:
    """Anchor._generate_anchors + get_anchors, data/anchor.py:23-68.

    Python floats (float64) for [cx,cy,w,h], cast to float32 (L52-53), then the
    yxyx conversion in float32 (L59-66).
    """
    from itertools import product
    from math import sqrt

    prior_boxes = []
    for idx, f_size in enumerate(feature_map_size):
        for j, i in product(range(int(f_size[0])), range(int(f_size[1]))):
            x = (i + 0.5) / f_size[1] * img_w
            y = (j + 0.5) / f_size[0] * img_h
            for ars in aspect_ratio:
                a = sqrt(ars)
                w = scale[idx] * a
                h = scale[idx] / a
                prior_boxes += [x, y, w, h]
    out = np.asarray(prior_boxes, dtype=np.float64).reshape(-1, 4).astype(np.float32)
    w = out[:, 2]
    h = out[:, 3]
    two = np.float32(2)
    return np.stack([out[:, 1] - (h / two), out[:, 0] - (w / two), out[:, 1] + (h / two), out[:, 0] + (w / two)],
                    axis=-1).astype(np.float32)
