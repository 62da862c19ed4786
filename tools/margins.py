"""Decision margins (float64, oracle/margins.py) of the batches the parity tests call deterministic, and of the headline
config: SURVEY.md 8(d).  usage: python tools/margins.py [images]   (CPU only; a few seconds per image)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import margins
from yolact_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = synth.CONFIGS[2]
out = {}
for name, first, mode in (("planted_soft (images 300...)", 300, "planted_soft"), ("config 2 (images 0...)", 0, None)):
    net, shp = synth.make_batch(cfg, batch=n, first_image=first, mode=mode)
    for nms, top_k, md in (("per_class", 200, 100), ("cc", 15, 15)):
        m = margins.batch_margins(net, shp, conf_thresh=cfg.conf_thresh, nms_thresh=cfg.nms_thresh, top_k=top_k, nms_mode=nms, max_dets=md)
        out[f"{name} {nms}"] = {"min": m["min"], "images": n}
print(json.dumps(out, indent=1))
