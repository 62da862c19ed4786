"""Decision margins of a Detect run in float64 (TEST INFRASTRUCTURE, NOT PRODUCT CODE; only tests/ and tools/ import it).

SURVEY.md 8(d), "deterministic configs": an input on which every decision of the path has margin - the confidence
compare (detection.py:73, :233), the IoU compare (:202, :229), the top_k / max_dets cuts (:186, :217, :242) and the
order of the kept rows, the float pixel compares of utils.crop (utils/utils.py:125-130) - gives the same kept indices,
classes and order under ANY float32 arithmetic whose errors stay below those margins, TensorFlow's included.  This
module measures the margins with the numpy oracle's own functions run in float64."""
from __future__ import annotations

import numpy as np

from . import detect_oracle as do


def image_margins(net_outs: dict, img_shape, b: int, conf_thresh=0.05, nms_thresh=0.5, top_k=200, nms_mode="per_class", max_dets=100):
    """Minimum margins of image b: dict with conf (every confidence compare; decides num_survivors), conf_relevant (the
    same where it can change the output rows), iou, cut (score gap at a top_k / max_dets cut), order (smallest gap
    between neighbouring kept scores) and crop_edge (distance of a crop edge inside the mask from an integer)."""
    f8 = np.float64
    cls = np.asarray(net_outs["pred_cls"][b], f8)
    prob = do.softmax_lastaxis(cls[None])[0][:, 1:]
    pmax = prob.max(-1)
    surv = np.nonzero(pmax > conf_thresh)[0]
    out = {"conf": np.inf, "conf_relevant": np.inf, "iou": np.inf, "cut": np.inf, "order": np.inf, "crop_edge": np.inf,
           "survivors": int(surv.size)}
    if nms_mode == "cc":
        out["conf"] = float(np.abs(pmax - conf_thresh).min())
    else:
        out["conf"] = float(np.abs(prob - conf_thresh).min())          # every (anchor, class) is a decision (detection.py:233)
    if surv.size == 0:
        out["conf_relevant"] = out["conf"]
        return out
    boxes = do._decode(np.asarray(net_outs["pred_offset"][b], f8)[surv], np.asarray(net_outs["priors"], f8)[surv])
    sp = prob[surv]

    def nms_margins(scores, bx):
        """scores [n] of one candidate set: its top_k cut and the IoU decisions among the k boxes; returns the kept ones."""
        k = min(top_k, scores.shape[0])
        idx = do.top_k_indices(scores, k)
        s = scores[idx]
        if scores.shape[0] > k:
            rest = np.delete(scores, idx)
            out["cut"] = min(out["cut"], float(s[-1] - rest.max()))
        iou = do._iou(bx[idx], bx[idx])
        iou = np.triu(iou, 0) - np.diag(np.diag(iou))
        iou_max = iou.max(0)
        if k > 1:
            out["iou"] = min(out["iou"], float(np.abs(iou_max[1:] - nms_thresh).min()))
        return idx, s, iou_max <= nms_thresh

    if nms_mode == "cc":
        idx, s, keep = nms_margins(sp.max(-1), boxes)
        ks, kb = s[keep], boxes[idx[keep]]
    else:
        ks, kb = [], []
        for c in range(sp.shape[1]):
            cand = np.nonzero(sp[:, c] > conf_thresh)[0]                 # (entries <= conf never survive L233 and never suppress a survivor)
            if cand.size == 0:
                continue
            idx, s, keep = nms_margins(sp[cand, c], boxes[cand])
            ks.append(s[keep]); kb.append(boxes[cand][idx[keep]])
        ks = np.concatenate(ks) if ks else np.zeros(0)
        kb = np.concatenate(kb) if kb else np.zeros((0, 4))
        order = np.argsort(-ks, kind="stable")
        ks, kb = ks[order], kb[order]
    n = min(max_dets, ks.shape[0])
    # A candidate at the confidence threshold is the lowest-scoring one of its image / class: it suppresses nobody ranked
    # above it and reaches the output only if fewer than max_dets rows outscore it.  So the confidence compare can change
    # the OUTPUT ROWS only on an image whose list is not full.
    out["conf_relevant"] = out["conf"] if ks.shape[0] < max_dets else np.inf
    if ks.shape[0] > n:
        out["cut"] = min(out["cut"], float(ks[n - 1] - ks[n]))
    if n > 1:
        out["order"] = float(np.min(ks[:n - 1] - ks[1:n]))
    H, W = float(img_shape[1]), float(img_shape[2])
    ph, pw = net_outs["proto_out"].shape[1:3]
    nb = do._sanitize(kb[:n], width=W, height=H) / np.array([H, W, H, W])
    x1, x2, y1, y2 = do.crop_bounds(nb, ph, pw)
    for e, size in ((x1, pw), (x2, pw), (y1, ph), (y2, ph)):
        e = np.asarray(e, f8)
        inner = e[(e > 0) & (e < size)]                                  # an edge clipped to 0 or to the size is that integer exactly
        if inner.size:
            out["crop_edge"] = min(out["crop_edge"], float(np.abs(inner - np.rint(inner)).min()))
    return out


def batch_margins(net_outs, img_shape, **kw):
    B = net_outs["pred_cls"].shape[0]
    per = [image_margins(net_outs, img_shape, b, **kw) for b in range(B)]
    keys = ("conf", "conf_relevant", "iou", "cut", "order", "crop_edge")
    return {"min": {k: min(p[k] for p in per) for k in keys}, "per_image": per}
