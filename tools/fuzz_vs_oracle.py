"""Randomised parity fuzz on the GPU: random small shapes / class counts / thresholds / NMS settings, every output
compared bit for bit with the C oracle (test infrastructure).  usage: python tools/fuzz_vs_oracle.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolact_amd import PostConfig, PostProcessor, synth
from oracle import c_oracle
from tests.util import compare, to_numpy
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for i in range(cases):
    C = int(rng.choice([2, 3, 5, 9, 17, 33, 81, 82, 130, 256]))
    f0 = int(rng.integers(3, 40)); f1 = int(rng.integers(1, 8))
    proto = (int(rng.integers(5, 40)), int(rng.integers(5, 40)))
    B = int(rng.integers(1, 6))
    p_fg = float(rng.choice([0.02, 0.1, 0.5, 0.95]))
    net, shp = synth.make_toy(int(rng.integers(0, 10**6)), batch=B, fmaps=((f0, f0), (f1, f1)), num_classes=C, proto_hw=proto, p_fg=p_fg)
    if rng.random() < 0.3:                       # scale / shift the logits: near-threshold and saturated rows
        net["pred_cls"] = (net["pred_cls"] * float(rng.choice([0.1, 3.0])) + float(rng.choice([0.0, 50.0, -90.0]))).astype(np.float32)
    conf = float(rng.choice([0.01, 0.05, 0.2, 0.5]))
    nms = float(rng.choice([0.3, 0.5, 0.7]))
    mode = str(rng.choice(["cc", "per_class"]))
    top_k = int(rng.choice([1, 5, 15, 64, 200, 256]))
    max_dets = int(rng.choice([1, 7, 40, 100, 256])) if mode == "per_class" else int(rng.choice([1, 7, 40, 100, 300]))
    crop = bool(rng.random() < 0.8)
    A = net["priors"].shape[0]
    ph, pw = net["proto_out"].shape[1:3]
    pp = PostProcessor(PostConfig(A, C, ph, pw, int(shp[1]), int(shp[2]), conf_thresh=conf, nms_thresh=nms, nms_mode=mode, top_k=top_k,
                                  max_dets=max_dets, use_cropped_mask=crop, max_batch=B))
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in net.items()}
    out = pp.run(dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
    pp.sync()
    got = to_numpy(out)
    got["num_survivors"] = pp.last_survivors(B)
    kw = dict(conf_thresh=conf, nms_thresh=nms, top_k=top_k, nms_mode=mode, max_dets=max_dets, use_cropped_mask=crop, nthreads=8)
    ref = c_oracle.detect(net, shp, **kw)
    try:
        assert np.array_equal(got["num_survivors"], ref["num_survivors"]), "num_survivors"
        compare(got, ref, exact_values=True, label=f"case{i}")
        # the same handle on a tracked output set (yp_run_kept): this batch, another one of the same shape, this one
        # again - every run must equal the oracle although nothing refills the masks after the first
        net2, _ = synth.make_toy(int(rng.integers(0, 10**6)), batch=B, fmaps=((f0, f0), (f1, f1)), num_classes=C, proto_hw=proto,
                                 p_fg=float(rng.choice([0.02, 0.5, 0.95])))
        dev2 = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in net2.items()}
        ref2 = c_oracle.detect(net2, shp, **kw)
        kept = pp.alloc_kept(B)
        for v in kept.values():
            v.fill_(3)
        for d, r, tag in ((dev, ref, "a"), (dev2, ref2, "b"), (dev, ref, "a2")):
            pp.run(d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"], out=kept)
            pp.sync()
            gk = to_numpy(kept)
            compare(gk, r, exact_values=True, label=f"case{i}/kept-{tag}")
            lim = min(max_dets, 256) if mode == "per_class" else min(max_dets, top_k, 256)
            for b in range(B):                      # the rows behind the detections, as far as a detection can land
                assert not gk["detection_masks"][b, int(r["num_detections"][b]):lim].any(), f"case{i}/kept-{tag}: stale mask rows, image {b}"
        kept.close()
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {i}: C={C} A={A} B={B} conf={conf} nms={nms} mode={mode} top_k={top_k} max_dets={max_dets} crop={crop} p_fg={p_fg}: {str(e)[:200]}")
    pp.close()
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
