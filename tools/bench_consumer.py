#!/usr/bin/env python3
"""Times the consumer step (resize + binarise + RLE) on Detect's outputs at BASELINE
config 2 (32 images x 100 detections, 138x138 -> 550x550), and - as the CPU comparison - the numpy
restatement of what the reference's callers do per detection (cv2.resize + threshold + encode) on a
bounded sample.  Prints one JSON line.  Not the headline bench (that is bench.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from yolact_amd import PostConfig, PostProcessor, consumer, synth
    cfg = synth.CONFIGS[2]
    B = cfg.batch
    net, shp = synth.make_batch(cfg, batch=B)
    A, C = net["pred_cls"].shape[1:]
    ph, pw = net["proto_out"].shape[1:3]
    pp = PostProcessor(PostConfig(A, C, ph, pw, cfg.img_h, cfg.img_w, conf_thresh=cfg.conf_thresh, nms_mode="per_class",
                                  top_k=cfg.top_k, max_dets=cfg.max_dets, max_batch=B))
    d = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
    out = pp.run(d["pred_offset"], d["pred_cls"], d["pred_mask_coef"], d["priors"], d["proto_out"])
    torch.cuda.synchronize()
    res = {}
    for order in ("resize_then_threshold", "threshold_then_resize"):
        for bitmaps in (False, True):
            cap = cfg.img_h * cfg.img_w + 1 if order == "resize_then_threshold" else 8 * cfg.img_w
            r = consumer.masks_to_rle(out["detection_masks"], out["num_detections"], (cfg.img_h, cfg.img_w), order=order,
                                      bitmaps=bitmaps, rle_cap=20000)
            torch.cuda.synchronize()
            cap = int(r["counts"].shape[2])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                consumer.masks_to_rle(out["detection_masks"], out["num_detections"], (cfg.img_h, cfg.img_w), order=order,
                                      bitmaps=bitmaps, rle_cap=cap)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            ndet = int(out["num_detections"].sum())
            res[f"{order}{'+bitmaps' if bitmaps else ''}"] = {
                "ms_per_batch": round(ms, 4), "detections": ndet, "detections_per_s": round(ndet / ms * 1e3, 1),
                "output_pixels_per_s": round(ndet * cfg.img_h * cfg.img_w / ms * 1e3, 1), "max_runs": int(r["len"].max())}
    # CPU: the numpy restatement of cv2.resize + threshold + rleEncode, per detection, on 16 detections
    from oracle import consumer_oracle as co
    m = out["detection_masks"][0, :16].cpu().numpy()
    t0 = time.perf_counter()
    for i in range(16):
        co.rle_counts(co.mask_to_bitmap(m[i], cfg.img_h, cfg.img_w))
    t = time.perf_counter() - t0
    res["cpu_numpy_restatement"] = {"detections_per_s": round(16 / t, 1), "sample": "16 detections, one core"}
    # F3: the whole export (coco_eval.py:123-155): device resize + RLE, one copy of the counts, one encoder call, dicts
    ids = list(range(B)); hw = [(cfg.img_h, cfg.img_w)] * B
    consumer.export_detections(out, ids, hw)
    t0 = time.perf_counter()
    for _ in range(3):
        ann = consumer.export_detections(out, ids, hw)
    t = (time.perf_counter() - t0) / 3
    res["export_detections"] = {"ms_per_batch": round(t * 1e3, 2), "annotations": len(ann), "note": "wall clock, host included"}
    # F4: the loss's mask assembly, forward and backward, at the same shape (32 images x 100 positives, boxes
    # = the detections' boxes in pixels)
    boxes_px = (out["detection_boxes"] * torch.tensor([cfg.img_h, cfg.img_w, cfg.img_h, cfg.img_w], device="cuda", dtype=torch.float32)).contiguous()
    coef = d["pred_mask_coef"][:, :cfg.max_dets].contiguous()
    npos = out["num_detections"].to(torch.int32).contiguous()
    proto = d["proto_out"].contiguous()
    masks = consumer.assemble_loss_masks(proto, coef, boxes_px, npos, (cfg.img_h, cfg.img_w))
    grad = torch.randn_like(masks)
    torch.cuda.synchronize()

    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    fwd_ms = timed(lambda: consumer.assemble_loss_masks(proto, coef, boxes_px, npos, (cfg.img_h, cfg.img_w), out=masks))
    bwd_ms = timed(lambda: consumer.assemble_loss_masks_backward(proto, coef, boxes_px, npos, masks, grad, (cfg.img_h, cfg.img_w)))
    inside = float((masks != 0).float().mean())

    def torch_ref():
        z = torch.einsum('byxk,bnk->bnyx', proto, coef)
        m = torch.sigmoid(z)
        dz = grad * m * (1 - m)
        return torch.einsum('bnyx,bnk->byxk', dz, coef), torch.einsum('bnyx,byxk->bnk', dz, proto)
    ref_ms = timed(torch_ref, reps=3)
    # the same with boxes of a training set's size: sides uniform in 10-60 % of the image (~12 % of the pixels inside)
    g = torch.Generator().manual_seed(0)
    hh = (0.1 + 0.5 * torch.rand(B, cfg.max_dets, generator=g)) * cfg.img_h
    ww = (0.1 + 0.5 * torch.rand(B, cfg.max_dets, generator=g)) * cfg.img_w
    y1 = torch.rand(B, cfg.max_dets, generator=g) * (cfg.img_h - hh)
    x1 = torch.rand(B, cfg.max_dets, generator=g) * (cfg.img_w - ww)
    big = torch.stack([y1, x1, y1 + hh, x1 + ww], -1).cuda().contiguous()
    full = torch.full((B,), cfg.max_dets, dtype=torch.int32, device="cuda")
    masks_big = consumer.assemble_loss_masks(proto, coef, big, full, (cfg.img_h, cfg.img_w))
    fwd_big = timed(lambda: consumer.assemble_loss_masks(proto, coef, big, full, (cfg.img_h, cfg.img_w), out=masks_big))
    bwd_big = timed(lambda: consumer.assemble_loss_masks_backward(proto, coef, big, full, masks_big, grad, (cfg.img_h, cfg.img_w)))
    res["loss_mask_assembly_large_boxes"] = {"forward_ms": round(fwd_big, 4), "backward_ms": round(bwd_big, 4), "rows": B * cfg.max_dets,
                                             "fraction_of_pixels_inside_boxes": round(float((masks_big != 0).float().mean()), 4)}
    # F2 again on those masks: boxes of a realistic size instead of the synthetic detections' small ones
    r = consumer.masks_to_rle(masks_big, full, (cfg.img_h, cfg.img_w), rle_cap=20000)
    cap_big = int(r["counts"].shape[2])
    rle_big = timed(lambda: consumer.masks_to_rle(masks_big, full, (cfg.img_h, cfg.img_w), rle_cap=cap_big))
    res["resize_then_threshold_large_boxes"] = {"ms_per_batch": round(rle_big, 4), "detections": B * cfg.max_dets,
                                                "detections_per_s": round(B * cfg.max_dets / rle_big * 1e3, 1),
                                                "fraction_of_pixels_inside_boxes": round(float((masks_big != 0).float().mean()), 4),
                                                "max_runs": int(r["len"].max())}
    res["loss_mask_assembly"] = {"forward_ms": round(fwd_ms, 4), "backward_ms": round(bwd_ms, 4), "rows": int(npos.sum()),
                                 "fraction_of_pixels_inside_boxes": round(inside, 4),
                                 "torch_rocblas_forward_plus_backward_ms": round(ref_ms, 4),
                                 "note": "torch leg: dense einsum/sigmoid on the same GPU, no crop, for scale only"}
    print(json.dumps({"workload": "config 2: 32 images x 100 detections, 138x138 -> 550x550", **res}))


if __name__ == "__main__":
    main()
