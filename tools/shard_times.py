"""Single-GPU step times of the shard shapes BASELINE configs 3 and 5 produce on 1/2/4/8 GPUs, and the strong-scaling
curve they PREDICT (not measured: one GPU here; the 8-GPU runs are the driver's).  A rank's step on G GPUs is
yp_run on 64/G images; with no data-path collective the job's step is the slowest rank's, so
    predicted images/s on G GPUs = 64 / t(64/G),   efficiency = t(64) / (G * t(64/G)).
usage: python tools/shard_times.py > profiles/r03/shard_times.json"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import PostConfig, PostProcessor, synth

def step_ms(cfg, net, nb, mode, steps=30, warmup=5):
    A, C = net["pred_cls"].shape[1:]; ph, pw = net["proto_out"].shape[1:3]
    pp = PostProcessor(PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh,
                                  nms_mode=mode, top_k=cfg.top_k, max_dets=cfg.max_dets, max_batch=nb))
    sets = []
    for _ in range(3):
        d = {k: torch.from_numpy(v if k == "priors" else v[:nb]).cuda() for k, v in net.items()}
        sets.append((d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"]))
    outs = [pp.alloc_outputs(nb) for _ in range(3)]
    for i in range(warmup):
        pp.run(*sets[i % 3], out=outs[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pp.run(*sets[i % 3], out=outs[i % 3])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the gather of this shard: outputs to pinned host memory
    host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in outs[0].items()}
    t1 = time.perf_counter()
    for k, hv in host.items():
        hv.copy_(outs[0][k], non_blocking=True)
    torch.cuda.synchronize()
    g = (time.perf_counter() - t1) * 1e3
    pp.close()
    return ms, g

res = {"note": "PREDICTED strong scaling from single-GPU shard times (one MI355X; rotating input and output sets); not a multi-GPU measurement",
       "configs": {}}
for cid in (3, 5):
    cfg = synth.CONFIGS[cid]
    net, shp = synth.make_batch(cfg, batch=64)
    for mode in ("per_class", "cc"):
        rows = {}
        for G in (1, 2, 4, 8):
            ms, g = step_ms(cfg, net, 64 // G, mode)
            rows[G] = {"images_per_gpu": 64 // G, "ms_per_step": round(ms, 4), "gather_ms_pinned_d2h": round(g, 3)}
        t1 = rows[1]["ms_per_step"]
        for G, r in rows.items():
            r["predicted_images_per_s"] = round(64 / (r["ms_per_step"] * 1e-3), 1)
            r["predicted_efficiency"] = round(t1 / (G * r["ms_per_step"]), 3)
        res["configs"][f"config{cid}:{mode}"] = rows
    del net
print(json.dumps(res, indent=1))
