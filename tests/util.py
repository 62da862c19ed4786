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
