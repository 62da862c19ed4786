"""How long does one Detect() call take on the host at batch 1 (the reference's inference call, infer.py:5-11), next to
the 21 us its kernels take on the GPU?  Prints the mean wall clock per call with and without a synchronisation per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import Detect, synth
cfg = synth.CONFIGS[1]
net, shp = synth.make_batch(cfg, batch=1, mode="planted")
dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
for name, kw in (("Detect()", {}), ("Detect(reuse_outputs=True)", {"reuse_outputs": True})):
    d = Detect(cfg.num_classes, 300, 100, cfg.conf_thresh, 0.5, **kw)
    for _ in range(50):
        d(dev, shp)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        d(dev, shp)
    t_issue = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        d(dev, shp)
        torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / n * 1e6
    print(f"{name}: host time per call {t_issue:.1f} us, throughput period {t_all:.1f} us, call + synchronise {t_sync:.1f} us")

# The same three launches captured once in a HIP graph (yp_run on fixed input and output tensors) and replayed
from yolact_amd import PostConfig, PostProcessor
A, C = net["pred_cls"].shape[1:]
pp = PostProcessor(PostConfig(A, C, 138, 138, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode="cc", top_k=15, max_dets=300,
                              zero_pad_masks=True, max_batch=1))
out = pp.alloc_outputs(1)
args = (dev["pred_offset"], dev["pred_cls"], dev["pred_mask_coef"], dev["priors"], dev["proto_out"])
pp.run(*args, out=out)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    pp.run(*args, out=out)
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
t_issue = (time.perf_counter() - t0) / n * 1e6
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n * 1e6
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
    torch.cuda.synchronize()
t_sync = (time.perf_counter() - t0) / n * 1e6
print(f"yp_run in a HIP graph, replayed: host time per replay {t_issue:.1f} us, throughput period {t_all:.1f} us, replay + synchronise {t_sync:.1f} us")
