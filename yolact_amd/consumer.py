"""The consumer step after ``Detect`` on the device (SURVEY.md 8f, F2 + F3).

What the reference's callers do on the CPU with every kept detection - ``cv2.resize`` of the
``[ph,pw]`` mask to the image size, ``> 0.5``, ``pycocotools.mask.encode``, and the annotation dicts
they build from it (``coco_eval.py:123-155``, ``infer.py:24-33``, ``train.py:492-515``) - with the
resize, the threshold and the run-length counting done by ``yp_masks_to_rle`` (``csrc/rle.hip``).
torch is plumbing (device memory, current stream); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

ORDERS = {"resize_then_threshold": _lib.YP_RESIZE_THEN_THRESHOLD,     # coco_eval.py:134-135, infer.py:32-33
          "threshold_then_resize": _lib.YP_THRESHOLD_THEN_RESIZE}     # train.py:497,504-505


_CAP_HINT = {}     # (max_dets, ph, pw, widest output) -> run-list capacity that the last overflowing call needed


def masks_to_rle(det_masks: torch.Tensor, num_det: torch.Tensor, out_hw, threshold: float = 0.5,
                 order: str = "resize_then_threshold", bitmaps: bool = False, rle_cap: int | None = None):
    """Resize + binarise + run-length encode the first ``num_det[b]`` masks of every image.

    det_masks [B,max_dets,ph,pw] float32 and num_det [B] int32 are CUDA tensors (yp_run's outputs).
    ``out_hw`` is one (height, width) for the whole batch, or a sequence of B such pairs - every image's own
    original size, as ``coco_eval.py:112-137`` resizes them - which still is ONE call (no bitmaps then).
    Returns a dict: ``counts`` [B,max_dets,cap] int32 (bit pattern of uint32: uncompressed COCO
    counts, column-major, zeros first), ``len`` [B,max_dets] int32, ``bitmaps`` [B,max_dets,out_h,out_w]
    uint8 or None, ``size`` (h, w) or the list of them.  If a detection has more runs than ``rle_cap`` the call is
    repeated once with room for the longest list (that needs one synchronisation; pass a generous ``rle_cap`` to
    avoid it).
    """
    if not det_masks.is_cuda or det_masks.dtype != torch.float32 or det_masks.dim() != 4 or not det_masks.is_contiguous():
        raise ValueError("det_masks must be a contiguous float32 CUDA tensor [B,max_dets,ph,pw]")
    B, md, ph, pw = det_masks.shape
    dev = det_masks.device
    if num_det.device != dev or num_det.dtype != torch.int32 or tuple(num_det.shape) != (B,) or not num_det.is_contiguous():
        raise ValueError("num_det must be a contiguous int32 tensor [B] on the masks' device")
    per_image = len(out_hw) > 0 and hasattr(out_hw[0], "__len__")
    L = _lib.load()
    if per_image:
        if len(out_hw) != B:
            raise ValueError("out_hw must hold one (height, width) per image")
        if bitmaps:
            raise ValueError("bitmaps are not written with per-image sizes")
        sizes = np.ascontiguousarray([[int(h), int(w)] for h, w in out_hw], dtype=np.int32).reshape(B, 2)
        full = int((sizes[:, 0].astype(np.int64) * sizes[:, 1]).max()) + 1 if B else 1
        widest = int(sizes[:, 1].max()) if B else 1
        size_out = [(int(h), int(w)) for h, w in sizes]
    else:
        out_h, out_w = int(out_hw[0]), int(out_hw[1])
        full, widest, size_out = out_h * out_w + 1, out_w, (out_h, out_w)
    # (without an explicit rle_cap: start from what the last call with masks and outputs of this shape needed, so that a
    #  loop over batches repeats the encoding at most once)
    hint_key = (md, ph, pw, widest)
    cap = min(full, int(rle_cap) if rle_cap is not None else max(8 * widest + 64, _CAP_HINT.get(hint_key, 0)))
    bm = torch.empty((B, md, out_h, out_w), dtype=torch.uint8, device=dev) if bitmaps else None
    while True:
        counts = torch.empty((B, md, cap), dtype=torch.int32, device=dev)
        lens = torch.empty((B, md), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if per_image:
            _lib.check(L.yp_masks_to_rle_sized(dev.index, B, md, ph, pw, det_masks.data_ptr(), num_det.data_ptr(), sizes.ctypes.data,
                                               float(threshold), ORDERS[order], counts.data_ptr(), lens.data_ptr(), cap,
                                               ctypes.c_void_p(stream)))
        else:
            _lib.check(L.yp_masks_to_rle(dev.index, B, md, ph, pw, det_masks.data_ptr(), num_det.data_ptr(), out_h, out_w,
                                         float(threshold), ORDERS[order], bm.data_ptr() if bm is not None else None,
                                         counts.data_ptr(), lens.data_ptr(), cap, ctypes.c_void_p(stream)))
        need = int(lens.max().item()) if B * md > 0 else 0
        if need <= cap:
            return {"counts": counts, "len": lens, "bitmaps": bm, "size": size_out}
        cap = min(full, need + need // 4)                    # some room for the next batch's longest list
        if cap <= 8 * (8 * widest + 64) and B * md * cap * 4 <= (1 << 30):   # (one noise mask must not size every later call)
            _CAP_HINT[hint_key] = cap


def rle_string(counts: np.ndarray) -> str:
    """pycocotools' compressed ``counts`` string of one list of run lengths (host; yp_rle_string)."""
    c = np.ascontiguousarray(counts).astype(np.uint32, copy=False)
    n = int(c.shape[0])
    cap = 7 * n + 1                                  # a count takes at most 7 characters
    buf = ctypes.create_string_buffer(cap)
    rc = _lib.load().yp_rle_string(c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, buf, cap)
    if rc < 0:
        _lib.check(rc)
    return buf.value.decode("ascii")


def rle_strings(counts: np.ndarray, lens: np.ndarray, starts: np.ndarray | None = None):
    """The compressed ``counts`` strings of many lists at once (host; yp_rle_strings).  ``counts`` is either
    [n, stride] (list i = the first lens[i] counts of row i) or, with ``starts``, a flat array in which list i
    begins at starts[i].  Returns a list of n str."""
    c = np.ascontiguousarray(counts).astype(np.uint32, copy=False)
    ln = np.ascontiguousarray(lens, dtype=np.int32)
    n = int(ln.shape[0])
    if starts is None:
        starts = np.arange(n, dtype=np.int64) * int(c.shape[1])
    st = np.ascontiguousarray(starts, dtype=np.int64)
    cap = 7 * int(ln.sum()) + n + 1                    # a count takes at most 7 characters
    buf = ctypes.create_string_buffer(cap)
    offs = np.zeros(n + 1, np.int64)
    rc = _lib.load().yp_rle_strings(c.ctypes.data, st.ctypes.data, ln.ctypes.data, n, int(c.size), ctypes.addressof(buf), cap,
                                    offs.ctypes.data)
    if rc < 0:
        _lib.check(int(rc))
    raw = buf.raw
    return [raw[offs[i]:offs[i + 1] - 1].decode("ascii") for i in range(n)]


def rle_dicts(rle: dict):
    """Per image, per detection: ``{'size': [h, w], 'counts': str}`` - what ``mask.encode`` returns after
    the ``.decode('ascii')`` of coco_eval.py:137.  The counts in use are packed on the device, copied to the host
    once, and encoded by one call for the whole batch."""
    lens_d = rle["len"]
    B, md = lens_d.shape
    lens = lens_d.cpu().numpy()
    width = max(int(lens.max()) if lens.size else 0, 1)
    c2 = rle["counts"][:, :, :width].reshape(B * md, width)
    used = torch.arange(width, device=c2.device)[None, :] < lens_d.reshape(-1, 1)
    flat = c2[used].cpu().numpy().view(np.uint32)                       # the lists back to back, in (b, n) order
    ln = lens.reshape(-1)
    starts = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]) if ln.size else np.zeros(0, np.int64)
    size = rle["size"]
    sizes = size if isinstance(size, list) else [size] * B
    strs = rle_strings(flat, ln, starts)
    return [[{"size": [sizes[b][0], sizes[b][1]], "counts": strs[b * md + n]} for n in range(md) if lens[b, n] > 0] for b in range(B)]


def export_detections(result: dict, image_ids, orig_hw, score_thresh: float = 0.05, threshold: float = 0.5,
                      return_boxes: bool = False):
    """The annotation list ``coco_eval.py:123-155`` builds from a Detect result (and
    ``utils/coco_tools.py:728-737`` for the evaluator): per image ``det_num = count_nonzero(scores >
    0.05)`` rows, and per row EXACTLY the four keys of L150-155 - ``image_id``, ``category_id = int(class)``,
    ``segmentation`` = the RLE of the mask resized to the image's own (h, w) and thresholded, ``score =
    float(score)`` - so that ``json.dump`` of the list is what L158-159 writes.  ``orig_hw[b]`` = (h, w) of
    image b before it was resized for the network.  All images go through ONE resize + RLE call, whatever
    their sizes.  ``return_boxes=True`` returns ``(annotations, boxes)`` with the ``[y1, x1, y2, x2]`` ints
    of L139-140 per annotation (the reference only draws them)."""
    masks, scores = result["detection_masks"], result["detection_scores"]
    B = masks.shape[0]
    det_num = (scores > score_thresh).sum(dim=1).to(torch.int32).contiguous()          # L123
    n_host = det_num.cpu().numpy()
    boxes = result["detection_boxes"].cpu().numpy()
    classes = result["detection_classes"].cpu().numpy()
    scores_h = scores.cpu().numpy()
    sizes = [(int(orig_hw[b][0]), int(orig_hw[b][1])) for b in range(B)]
    enc = rle_dicts(masks_to_rle(masks, det_num, sizes, threshold=threshold))           # L131-137, every image at its own size
    annotations, box_pixels = [], []
    for b, image_id in enumerate(image_ids):
        _h, _w = sizes[b]
        det_boxes = boxes[b][:n_host[b]] * np.array([_h, _w, _h, _w])                   # L124-125
        for i in range(int(n_host[b])):
            _y1, _x1, _y2, _x2 = (int(v) for v in det_boxes[i].astype(int))             # L139-140
            annotations.append({"image_id": image_id, "category_id": int(classes[b][i]),      # L150-155
                                "segmentation": enc[b][i], "score": float(scores_h[b][i])})
            box_pixels.append([_y1, _x1, _y2, _x2])
    return (annotations, box_pixels) if return_boxes else annotations


def assemble_loss_masks(proto: torch.Tensor, pos_coef: torch.Tensor, pos_prior_box: torch.Tensor, num_pos: torch.Tensor,
                        img_hw, use_cropped_mask: bool = True, out: torch.Tensor | None = None) -> torch.Tensor:
    """Forward mask assembly of the training loss (``loss/loss_yolact.py:239-253``) for a whole batch:
    proto [B,ph,pw,32], pos_coef [B,rows,32], pos_prior_box [B,rows,4] yxyx pixels, num_pos [B] int32 ->
    [B,rows,ph,pw] float32 (rows >= num_pos[b] are zeros).  No gradient is recorded here: LossMaskAssembly
    below is the differentiable form."""
    B, ph, pw, K = proto.shape
    rows = pos_coef.shape[1]
    dev = proto.device
    for name, t, shape, dt in (("proto", proto, (B, ph, pw, 32), torch.float32), ("pos_coef", pos_coef, (B, rows, 32), torch.float32),
                               ("pos_prior_box", pos_prior_box, (B, rows, 4), torch.float32), ("num_pos", num_pos, (B,), torch.int32)):
        if not t.is_cuda or t.device != dev or t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous {dt} CUDA tensor of shape {shape}")
    if out is None:
        out = torch.empty((B, rows, ph, pw), dtype=torch.float32, device=dev)
    ws = torch.empty((B, rows, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.load().yp_assemble_masks(dev.index, B, rows, ph, pw, int(img_hw[0]), int(img_hw[1]), int(use_cropped_mask),
                                             proto.data_ptr(), pos_coef.data_ptr(), pos_prior_box.data_ptr(), num_pos.data_ptr(),
                                             ws.data_ptr(), out.data_ptr(), ctypes.c_void_p(stream)))
    return out


def assemble_loss_masks_backward(proto: torch.Tensor, pos_coef: torch.Tensor, pos_prior_box: torch.Tensor, num_pos: torch.Tensor,
                                 masks: torch.Tensor, grad_masks: torch.Tensor, img_hw, use_cropped_mask: bool = True):
    """Gradients of assemble_loss_masks w.r.t. proto and pos_coef given dL/dmasks (what TensorFlow's autodiff
    derives from ``loss/loss_yolact.py:239-253``): returns (grad_proto [B,ph,pw,32], grad_coef [B,rows,32]).
    ``masks`` must be assemble_loss_masks' output for the same inputs.  Deterministic (fixed summation order)."""
    B, ph, pw, K = proto.shape
    rows = pos_coef.shape[1]
    dev = proto.device
    for name, t, shape, dt in (("proto", proto, (B, ph, pw, 32), torch.float32), ("pos_coef", pos_coef, (B, rows, 32), torch.float32),
                               ("pos_prior_box", pos_prior_box, (B, rows, 4), torch.float32), ("num_pos", num_pos, (B,), torch.int32),
                               ("masks", masks, (B, rows, ph, pw), torch.float32), ("grad_masks", grad_masks, (B, rows, ph, pw), torch.float32)):
        if not t.is_cuda or t.device != dev or t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous {dt} CUDA tensor of shape {shape}")
    grad_proto = torch.empty((B, ph, pw, 32), dtype=torch.float32, device=dev)
    grad_coef = torch.empty((B, rows, 32), dtype=torch.float32, device=dev)
    L = _lib.load()
    ws = torch.empty((int(L.yp_assemble_masks_backward_workspace(B, rows, ph)) // 4,), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.yp_assemble_masks_backward(dev.index, B, rows, ph, pw, int(img_hw[0]), int(img_hw[1]), int(use_cropped_mask),
                                                      proto.data_ptr(), pos_coef.data_ptr(), pos_prior_box.data_ptr(), num_pos.data_ptr(),
                                                      masks.data_ptr(), grad_masks.data_ptr(), ws.data_ptr(), grad_proto.data_ptr(),
                                                      grad_coef.data_ptr(), ctypes.c_void_p(stream)))
    return grad_proto, grad_coef


class LossMaskAssembly(torch.autograd.Function):
    """Differentiable mask assembly of the training loss: ``LossMaskAssembly.apply(proto, pos_coef, pos_prior_box,
    num_pos, img_h, img_w, use_cropped_mask)`` -> masks [B,rows,ph,pw]; gradients flow to proto and pos_coef
    (the boxes are constants of the crop, as in the reference: utils.crop multiplies by a 0/1 mask)."""

    @staticmethod
    def forward(ctx, proto, pos_coef, pos_prior_box, num_pos, img_h, img_w, use_cropped_mask=True):
        proto, pos_coef, pos_prior_box = proto.contiguous(), pos_coef.contiguous(), pos_prior_box.contiguous()
        masks = assemble_loss_masks(proto, pos_coef, pos_prior_box, num_pos, (img_h, img_w), use_cropped_mask)
        ctx.save_for_backward(proto, pos_coef, pos_prior_box, num_pos, masks)
        ctx.cfg = (int(img_h), int(img_w), bool(use_cropped_mask))
        return masks

    @staticmethod
    def backward(ctx, grad_masks):
        proto, pos_coef, pos_prior_box, num_pos, masks = ctx.saved_tensors
        gp, gc = assemble_loss_masks_backward(proto, pos_coef, pos_prior_box, num_pos, masks, grad_masks.contiguous(),
                                              ctx.cfg[:2], ctx.cfg[2])
        return gp, gc, None, None, None, None, None
