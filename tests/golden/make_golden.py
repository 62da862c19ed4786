"""Generates the committed golden fixtures from the numpy oracle (oracle/detect_oracle.py).

The reference cannot run here (no TensorFlow: SURVEY.md 8c), so these vectors pin the ORACLE,
not the reference: "parity unpinned".  They exist so that (a) the C oracle and the numpy oracle
are held to the same answers on any machine, (b) the GPU path is checked against committed data
as well as against a live oracle.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import detect_oracle as po  # noqa: E402
from yolact_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

TOY_CASES = [
    # name, seed, nms_mode, top_k, max_dets, conf
    ("toy_cc15", 3, "cc", 15, 15, 0.05),
    ("toy_cc200", 3, "cc", 200, 100, 0.05),
    ("toy_pc200", 3, "per_class", 200, 100, 0.05),
    ("toy_cc15_s7", 7, "cc", 15, 15, 0.3),
    ("toy_pc50_s11", 11, "per_class", 50, 20, 0.1),
]


def main():
    for name, seed, mode, top_k, max_dets, conf in TOY_CASES:
        net, shp = synth.make_toy(seed)
        ref = po.detect(net, shp, conf_thresh=conf, top_k=top_k, nms_mode=mode, max_output_size=max_dets,
                        max_dets=max_dets)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), seed=seed, nms_mode=mode, top_k=top_k, max_dets=max_dets, conf=conf,
            img_shape=shp, **{"in_" + k: v for k, v in net.items()},
            **{"out_" + k: v for k, v in ref.items()})
        print(name, ref["num_detections"], ref["num_survivors"])
    # one full-size image of BASELINE config 2 per mode: inputs are regenerated from the seed
    # (yolact_amd.synth), only the small outputs are stored (not the 19 M-float masks).
    cfg = synth.CONFIGS[2]
    net, shp = synth.make_batch(cfg, batch=1)
    out = {}
    for mode, top_k, max_dets in (("cc", 15, 15), ("cc", 200, 100), ("per_class", 200, 100)):
        ref = po.detect(net, shp, conf_thresh=cfg.conf_thresh, top_k=top_k, nms_mode=mode, max_output_size=max_dets,
                        max_dets=max_dets)
        tag = f"{mode}_{top_k}"
        for k in ("detection_boxes", "detection_classes", "detection_scores", "num_detections",
                  "detection_anchor_indices", "num_survivors"):
            out[f"{tag}_{k}"] = ref[k]
        m = ref["detection_masks"][0]
        out[f"{tag}_mask_sums"] = m.reshape(m.shape[0], -1).sum(axis=1, dtype=np.float64)
        out[f"{tag}_mask_nonzero"] = (m.reshape(m.shape[0], -1) > 0).sum(axis=1)
        print("config2", tag, ref["num_detections"], ref["num_survivors"])
    np.savez_compressed(os.path.join(HERE, "config2_image0.npz"), **out)


if __name__ == "__main__":
    main()
