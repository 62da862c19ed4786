"""Diagnostic (needs a library built with -DYP_STAMPS): per-phase cycle counts of the NMS kernels.
usage: python -m yolact_amd.build -DYP_STAMPS && python tools/nms_stamps.py [per_class|cc] [config]"""
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
out = pp.alloc_outputs(B)
for _ in range(5):
    pp.run(d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"], out=out)
pp.sync()
L = _lib.load()
buf = np.zeros((3, 4096, 16), np.uint64)
L.yp_debug_stamps(pp._h, buf.ctypes.data_as(ctypes.c_void_p))
for k, name in enumerate(("segment_nms (cc_nms / group_nms)", "final_select", "spare")):
    t = buf[k][buf[k][:, 15] == 1]
    if not len(t):
        continue
    print(f"{name}: {len(t)} blocks; median cycles from stamp 0 (and delta):")
    prev = 0.0
    for i in range(1, 15):
        ok = t[:, i] > 0
        if not ok.any():
            continue
        v = np.median((t[ok, i] - t[ok, 0]).astype(np.int64))
        print(f"  stamp {i:2d}: {v:9.0f}  (+{v - prev:7.0f})")
        prev = v
