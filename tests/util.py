"""Helpers shared by the parity tests."""
from __future__ import annotations

import numpy as np


def to_numpy(d):
    out = {}
    for k, v in d.items():
        out[k] = v.detach().cpu().numpy() if hasattr(v, "detach") else v
    return out


def compare(got, ref, *, exact_values: bool, atol=1e-5, masks=True, label=""):
    """got: product outputs (numpy), ref: oracle outputs.  Rows beyond num_det are ignored for masks."""
    B = ref["num_detections"].shape[0]
    rows = min(got["detection_scores"].shape[1], ref["detection_scores"].shape[1])
    assert np.array_equal(got["num_detections"], ref["num_detections"]), (label, got["num_detections"], ref["num_detections"])
    assert np.array_equal(got["detection_anchor_indices"][:, :rows], ref["detection_anchor_indices"][:, :rows]), label + " kept anchor indices"
    assert np.array_equal(got["detection_classes"][:, :rows], ref["detection_classes"][:, :rows]), label + " classes"
    for key in ("detection_scores", "detection_boxes"):
        g, r = got[key][:, :rows], ref[key][:, :rows]
        if exact_values:
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), f"{label} {key} not bit-exact, max diff {np.abs(g - r).max()}"
        else:
            assert np.abs(g - r).max() <= atol, f"{label} {key} max diff {np.abs(g - r).max()}"
    if masks and ref.get("detection_masks") is not None:
        for b in range(B):
            n = int(ref["num_detections"][b])
            g, r = got["detection_masks"][b, :n], ref["detection_masks"][b, :n]
            if exact_values:
                assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), f"{label} masks image {b} not bit-exact, max diff {np.abs(g - r).max()}"
            else:
                assert np.abs(g - r).max() <= atol, f"{label} masks image {b} max diff {np.abs(g - r).max()}"


def kept_index_agreement(got, ref):
    """The north_star's "kept-index agreement" between two arithmetics, three ways:
    per_detection  fraction of output rows whose (anchor index, class) are identical
    kept_set       fraction of images that keep the same set of (anchor, class) pairs
    exact_order    fraction of images identical row by row (1-ulp score near-ties reorder rows)"""
    B = ref["num_detections"].shape[0]
    rows_same = rows_tot = sets = order = 0
    for b in range(B):
        n, m = int(ref["num_detections"][b]), int(got["num_detections"][b])
        a = list(zip(got["detection_anchor_indices"][b, :m].tolist(), got["detection_classes"][b, :m].tolist()))
        d = list(zip(ref["detection_anchor_indices"][b, :n].tolist(), ref["detection_classes"][b, :n].tolist()))
        rows_same += sum(x == y for x, y in zip(a, d))
        rows_tot += max(n, m)
        sets += 1 if set(a) == set(d) else 0
        order += 1 if a == d else 0
    return {"per_detection": rows_same / max(1, rows_tot), "kept_set": sets / max(1, B), "exact_order": order / max(1, B)}
