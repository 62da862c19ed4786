"""CPU oracle for the consumer step after Detect (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  It restates what the reference's callers do with Detect's outputs:

* ``coco_eval.py:131-137`` / ``infer.py:31-33``: ``cv2.resize(mask, (w, h))`` on the float mask, then
  ``> 0.5``, then ``pycocotools.mask.encode`` of the Fortran-ordered bitmap;
* ``train.py:497-506``: ``mask > 0.5`` first, ``astype("uint8")``, then ``cv2.resize`` of the 0/1 bytes;
* ``coco_eval.py:91-98`` ``binary_mask_to_rle``: the uncompressed run lengths (column-major, a leading
  0 when the first pixel is set) - the one piece of this step the reference states itself;
* ``coco_eval.py:123-155`` / ``utils/coco_tools.py:682-737``: the annotation dicts.

PARITY UNPINNED: neither ``cv2`` nor ``pycocotools`` is importable here and the reference holds no
fixture of their outputs.  The two third-party algorithms are restated from their published sources:

* OpenCV ``cv::resize`` with ``INTER_LINEAR`` (imgproc/src/resize.cpp, 3.4/4.x; no version is pinned
  by the reference): half-pixel centres ``fx = (float)((dx + 0.5) * scale_x - 0.5)`` with
  ``scale_x = 1. / ((double)dst_w / src_w)``, ``sx = cvFloor(fx)``, left/right border clamps that
  zero ``fx``, row indices clipped (``fy`` kept) in the vertical pass, horizontal pass first.
  float32: ``S[sx]*a0 + S[sx+1]*a1`` then ``R0*b0 + R1*b1``, plain multiplies and adds (an FMA-contracting
  build of OpenCV can differ in the last bit).  uint8: 11-bit fixed point coefficients
  (``cvRound(c * 2048)``), ``uchar((((b0 * (R0 >> 4)) >> 16) + ((b1 * (R1 >> 4)) >> 16) + 2) >> 2)``.
* pycocotools ``rleEncode`` / ``rleToString`` (common/maskApi.c): counts of alternating runs starting
  with zeros; the string is a LEB128-like code of the counts, from the third count on as differences
  to the count two places before.
"""
from __future__ import annotations

from itertools import groupby

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _linear_coeffs(src: int, dst: int):
    """Per destination index: (source index, float weights (1-f, f), first index of the right border)."""
    scale = 1.0 / (float(dst) / float(src))                       # double, as cv::resize computes it
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)             # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                             # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _no_area_substitution(h, w, out_h, out_w):
    """cv::resize: `interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2` -> INTER_AREA
    (imgproc/src/resize.cpp).  Its float 2x2 average sums in a SIMD-width dependent order; not restated - the
    product rejects this size pair too."""
    if h == 2 * out_h and w == 2 * out_w:
        raise ValueError("exact 2x decimation in both axes: cv::resize uses INTER_AREA there (not restated)")


def resize_linear_f32(src: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(src, (out_w, out_h)) for a float32 single-channel image, INTER_LINEAR."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src.shape
    if (h, w) == (out_h, out_w):
        return src.copy()                                         # cv::resize copies when the sizes agree
    _no_area_substitution(h, w, out_h, out_w)
    one = np.float32(1.0)
    sx, fx = _linear_coeffs(w, out_w)
    left = sx < 0
    fx = np.where(left, np.float32(0), fx); sx = np.where(left, 0, sx)
    right = sx >= w - 1                                           # dx >= xmax: D[dx] = S[sx] * ONE
    fx = np.where(right, np.float32(0), fx); sx = np.where(right, w - 1, sx)
    a0, a1 = (one - fx).astype(np.float32), fx.astype(np.float32)
    sx1 = np.minimum(sx + 1, w - 1)
    rows = np.where(right[None, :], src[:, sx] * one, src[:, sx] * a0[None, :] + src[:, sx1] * a1[None, :]).astype(np.float32)
    sy, fy = _linear_coeffs(h, out_h)                             # vertical: weights kept, row indices clipped
    b0, b1 = (one - fy).astype(np.float32), fy.astype(np.float32)
    r0 = np.clip(sy, 0, h - 1)
    r1 = np.clip(sy + 1, 0, h - 1)
    return (rows[r0] * b0[:, None] + rows[r1] * b1[:, None]).astype(np.float32)


def resize_linear_u8(src: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(src, (out_w, out_h)) for a uint8 single-channel image, INTER_LINEAR (fixed point)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    if (h, w) == (out_h, out_w):
        return src.copy()
    _no_area_substitution(h, w, out_h, out_w)
    one = np.float32(1.0)
    scale = np.float32(INTER_RESIZE_COEF_SCALE)
    sx, fx = _linear_coeffs(w, out_w)
    left = sx < 0
    fx = np.where(left, np.float32(0), fx); sx = np.where(left, 0, sx)
    right = sx >= w - 1
    fx = np.where(right, np.float32(0), fx); sx = np.where(right, w - 1, sx)
    ia0 = np.rint((one - fx).astype(np.float32) * scale).astype(np.int64)   # saturate_cast<short>(c * 2048): cvRound
    ia1 = np.rint(fx.astype(np.float32) * scale).astype(np.int64)
    sx1 = np.minimum(sx + 1, w - 1)
    s = src.astype(np.int64)
    rows = np.where(right[None, :], s[:, sx] * INTER_RESIZE_COEF_SCALE, s[:, sx] * ia0[None, :] + s[:, sx1] * ia1[None, :])
    sy, fy = _linear_coeffs(h, out_h)
    ib0 = np.rint((one - fy).astype(np.float32) * scale).astype(np.int64)
    ib1 = np.rint(fy.astype(np.float32) * scale).astype(np.int64)
    r0 = np.clip(sy, 0, h - 1)
    r1 = np.clip(sy + 1, 0, h - 1)
    v = (((ib0[:, None] * (rows[r0] >> 4)) >> 16) + ((ib1[:, None] * (rows[r1] >> 4)) >> 16) + 2) >> 2
    return v.astype(np.uint8)


def binary_mask_to_rle(binary_mask: np.ndarray):
    """coco_eval.py:91-98, as written there (itertools.groupby over the column-major ravel)."""
    rle = {'counts': [], 'size': list(binary_mask.shape)}
    counts = rle.get('counts')
    for i, (value, elements) in enumerate(groupby(binary_mask.ravel(order='F'))):
        if i == 0 and value == 1:
            counts.append(0)
        counts.append(len(list(elements)))
    return rle


def rle_counts(mask: np.ndarray) -> np.ndarray:
    """pycocotools rleEncode: the same counts as binary_mask_to_rle, vectorised."""
    v = np.asarray(mask).ravel(order='F').astype(np.uint8) != 0
    n = v.shape[0]
    prev = np.concatenate([[False], v[:-1]])
    b = np.nonzero(v != prev)[0]                                   # positions where the value changes (virtual v[-1] = 0)
    return np.diff(np.concatenate([[0], b, [n]])).astype(np.uint32)


def rle_to_string(counts) -> str:
    """pycocotools rleToString (maskApi.c)."""
    out = []
    cnts = [int(c) for c in counts]
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            c = x & 0x1f
            x >>= 5                                                # arithmetic shift (python ints: floor)
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def rle_from_string(s: str):
    """pycocotools rleFrString: inverse of rle_to_string."""
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1; k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def rle_decode(counts, h: int, w: int) -> np.ndarray:
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + int(c)] = 1
        pos += int(c); val ^= 1
    return flat.reshape((h, w), order='F')


RESIZE_THEN_THRESHOLD = 0     # coco_eval.py:134-135, infer.py:32-33
THRESHOLD_THEN_RESIZE = 1     # train.py:497,504-505


def mask_to_bitmap(mask: np.ndarray, out_h: int, out_w: int, threshold: float = 0.5, order: int = RESIZE_THEN_THRESHOLD):
    if order == RESIZE_THEN_THRESHOLD:
        return (resize_linear_f32(mask, out_h, out_w) > np.float32(threshold)).astype(np.uint8)
    return resize_linear_u8((np.asarray(mask, np.float32) > np.float32(threshold)).astype(np.uint8), out_h, out_w)


def masks_to_rle(det_masks: np.ndarray, num_det: np.ndarray, out_h: int, out_w: int, threshold: float = 0.5,
                 order: int = RESIZE_THEN_THRESHOLD):
    """Per image, per kept detection: (bitmap uint8 [out_h,out_w], counts uint32[])."""
    res = []
    for b in range(det_masks.shape[0]):
        img = []
        for n in range(int(num_det[b])):
            bm = mask_to_bitmap(det_masks[b, n], out_h, out_w, threshold, order)
            img.append((bm, rle_counts(bm)))
        res.append(img)
    return res


def export_detections(result: dict, image_ids, orig_hw, threshold: float = 0.5, score_thresh: float = 0.05):
    """coco_eval.py:123-155: the annotation list for a batch of Detect results (numpy arrays)."""
    annotations = []
    for b, image_id in enumerate(image_ids):
        _h, _w = orig_hw[b]
        scores = result['detection_scores'][b]
        det_num = int(np.count_nonzero(scores > score_thresh))                        # L123
        det_boxes = result['detection_boxes'][b][:det_num] * np.array([_h, _w, _h, _w])   # L124-125 (float64 product)
        det_masks = result['detection_masks'][b][:det_num]
        det_scores = scores[:det_num]
        det_classes = result['detection_classes'][b][:det_num]
        for i in range(det_num):
            bm = mask_to_bitmap(det_masks[i], _h, _w, threshold, RESIZE_THEN_THRESHOLD)   # L134-135
            rle = {'size': [int(_h), int(_w)], 'counts': rle_to_string(rle_counts(bm))}   # L136-137
            _y1, _x1, _y2, _x2 = (int(v) for v in det_boxes[i].astype(int))            # L139-140
            annotations.append({"image_id": image_id, "category_id": int(det_classes[i]), "segmentation": rle,
                                "score": float(det_scores[i])})                        # L150-155: these four keys
    return annotations
