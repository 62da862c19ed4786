"""Diagnostic (needs a library built with -DYP_STAMPS): per-phase cycle counts of the NMS kernels.
usage: python -m yolact_amd.build -DYP_STAMPS && python tools/nms_stamps.py [per_class|cc] [config] [kept]
(kept: library-tracked output sets, i.e. no zero-fill workgroups beside the NMS)"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import PostConfig, PostProcessor, synth, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "per_class"
cfg = synth.CONFIGS[int(sys.argv[2]) if len(sys.argv) > 2 else 2]
net, shp = synth.make_batch(cfg)
B, A, C = net["pred_cls"].shape
ph, pw = net["proto_out"].shape[1:3]
pp = PostProcessor(PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode=mode, top_k=cfg.top_k,
                              max_dets=cfg.max_dets, max_batch=B))
d = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
kept = len(sys.argv) > 3 and sys.argv[3] == "kept"
outs = [pp.alloc_kept(B) if kept else pp.alloc_outputs(B) for _ in range(3)]     # rotating output sets, as bench.py's headline does
for i in range(9):
    pp.run(d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"], out=outs[i % 3])
pp.sync()
L = _lib.load()
buf = np.zeros((3, 4096, 16), np.uint64)
L.yp_debug_stamps(pp._h, buf.ctypes.data_as(ctypes.c_void_p))
t0 = min(int(buf[k][buf[k][:, 15] == 1][:, 0].min()) for k in range(2) if (buf[k][:, 15] == 1).any())
for k, name in enumerate(("nms workgroups (cc_nms / class groups)", "final selection (the image's last group)", "spare")):
    t = buf[k][buf[k][:, 15] == 1]
    if not len(t):
        continue
    print(f"{name}: {len(t)} blocks; us since the first workgroup started: median / max per stamp")
    for i in range(0, 15):
        ok = t[:, i] > 0
        if not ok.any():
            continue
        v = (t[ok, i].astype(np.int64) - t0) / 100.0
        print(f"  stamp {i:2d}: {np.median(v):8.2f} {v.max():8.2f}")
if buf[2][0][1] > 0:
    print(f"fill workgroups: first start {(int(buf[2][0][0]) - t0) / 100.0:.2f} us, last end (max over ALL runs' clocks) {(int(buf[2][0][1]) - t0) / 100.0:.2f} us")
