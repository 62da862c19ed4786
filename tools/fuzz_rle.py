"""Randomised fuzz of yp_masks_to_rle on the GPU against the consumer oracle (test infrastructure): random mask and
output sizes, sparse / box-limited / dense masks, both threshold orders, thresholds incl. 0 and negative.
usage: python tools/fuzz_rle.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_consumer import _check, _blobs
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
checked = 0
for i in range(cases):
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    oh = int(rng.integers(1, 5 * h + 40)); ow = int(rng.integers(1, 5 * w + 40))
    if rng.random() < 0.1:
        oh, ow = h, w                                            # cv::resize's copy
    B, N = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    kind = rng.integers(0, 4)
    if kind == 0:
        m = _blobs(rng, B, N, h, w)
    elif kind == 1:
        m = rng.random((B, N, h, w)).astype(np.float32)
    elif kind == 2:
        m = (rng.random((B, N, h, w)) * (rng.random((B, N, h, w)) < 0.03)).astype(np.float32)     # isolated pixels
    else:
        m = np.zeros((B, N, h, w), np.float32)
    nd = rng.integers(0, N + 1, B)
    order = str(rng.choice(["resize_then_threshold", "threshold_then_resize"]))
    thr = float(rng.choice([0.5, 0.5, 0.0, 0.3, -0.1]))
    try:
        _check(m, nd, (oh, ow), order, threshold=thr, rle_cap=oh * ow + 1)
        checked += int(nd.sum())
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {i}: {h}x{w} -> {oh}x{ow} kind={kind} order={order} thr={thr}: {str(e)[:160]}")
print(f"{cases} cases, {checked} detections compared, {bad} mismatches")
sys.exit(1 if bad else 0)
