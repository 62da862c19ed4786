"""What the host side of a sharded step costs with the float masks gathered, and with the consumers' resize + threshold +
RLE done on the device instead (ShardedDetect(..., rle_hw=...)).  One GPU, one shard: config 3's 64 images."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import synth
from yolact_amd.sharding import ShardedDetect
cfg = synth.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
net, shp = synth.make_batch(cfg)
dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
B = net["pred_cls"].shape[0]
sd = ShardedDetect(cfg.num_classes, 300, 100, cfg.conf_thresh, 0.5, devices=[0], strict_reference=False, nms_mode="per_class",
                   top_k=200, max_dets=100)
out = {"config": int(sys.argv[1]) if len(sys.argv) > 1 else 3, "images": B, "image_hw": [cfg.img_h, cfg.img_w]}
for name, kw in (("float_masks_gathered", {}), ("rle_on_device", {"rle_hw": (cfg.img_h, cfg.img_w)})):
    for _ in range(2):
        r = sd(dev, shp, **kw)
    t0 = time.perf_counter()
    r = sd(dev, shp, **kw)
    wall = (time.perf_counter() - t0) * 1e3
    t = sd.last_timing[0]
    host_bytes = sum(v.numel() * v.element_size() for k, v in r.items() if hasattr(v, "numel"))
    out[name] = {"wall_ms": round(wall, 2), "compute_ms": round(t["compute_ms"], 3), "gather_ms": round(t["gather_ms"], 3),
                 "rle_ms": None if t["rle_ms"] is None else round(t["rle_ms"], 2), "host_tensor_bytes": host_bytes,
                 "rle_string_bytes": sum(len(d["counts"]) for img in r.get("detection_rle", []) for d in img) if "detection_rle" in r else None}
print(json.dumps(out))
