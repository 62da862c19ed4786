"""Diagnostic: config 2 with one handle / one stream against two handles on two streams (batches alternate)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import PostConfig, PostProcessor, synth
cfg = synth.CONFIGS[2]; B = cfg.batch
net, shp = synth.make_batch(cfg, batch=B)
A, C = net["pred_cls"].shape[1:]; ph, pw = net["proto_out"].shape[1:3]
d = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
def mk():
    return PostProcessor(PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode="per_class",
                                    top_k=cfg.top_k, max_dets=cfg.max_dets, max_batch=B))
for nstreams in (1, 2, 3):
    pps = [mk() for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [[pp.alloc_outputs(B) for _ in range(3)] for pp in pps]
    def step(i):
        s = i % nstreams
        with torch.cuda.stream(streams[s]):
            pps[s].run(d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"], out=outs[s][(i // nstreams) % 3])
    for i in range(30): step(i)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{nstreams} stream(s): {t / K * 1e3:.4f} ms per step, {B * K / t:,.0f} images/s")
