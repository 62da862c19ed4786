"""Diagnostic: config 2 with one handle / one stream against two or three handles on as many streams (batches alternate).
usage: python tools/two_streams_probe.py [kept]   (kept: library-tracked output sets, yp_run_kept; two input sets of different images)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import PostConfig, PostProcessor, synth
cfg = synth.CONFIGS[2]; B = cfg.batch
net, shp = synth.make_batch(cfg, batch=B)
A, C = net["pred_cls"].shape[1:]; ph, pw = net["proto_out"].shape[1:3]
d = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
kept = len(sys.argv) > 1 and sys.argv[1] == "kept"
net2, _ = synth.make_batch(cfg, batch=B, first_image=4096)
d2 = {k: torch.from_numpy(v).cuda() for k, v in net2.items()}
def mk():
    return PostProcessor(PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode="per_class",
                                    top_k=cfg.top_k, max_dets=cfg.max_dets, max_batch=B))
for nstreams in (1, 2, 3):
    pps = [mk() for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [[pp.alloc_kept(B) if kept else pp.alloc_outputs(B) for _ in range(3)] for pp in pps]
    def step(i):
        s = i % nstreams
        with torch.cuda.stream(streams[s]):
            x = d2 if (i // (3 * nstreams)) & 1 else d
            pps[s].run(x["pred_offset"], x["pred_cls"], x["pred_mask_coef"], x["priors"], x["proto_out"], out=outs[s][(i // nstreams) % 3])
    for i in range(30): step(i)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{nstreams} stream(s): {t / K * 1e3:.4f} ms per step, {B * K / t:,.0f} images/s")
