"""Diagnostic: mask_rle time per batch of 3200 detections (138x138 -> 550x550) against how much of a mask is non-zero."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import consumer
B, N, h, w, H, W = 32, 100, 138, 138, 550, 550
nd = torch.full((B,), N, dtype=torch.int32, device="cuda")
g = torch.Generator().manual_seed(0)
noise = torch.rand(h, w, generator=g).cuda()
for name, side in (("all zero", 0), ("8x8 box", 8), ("24x24 box", 24), ("48x48 box", 48), ("96x96 box", 96), ("full", 138)):
    m = torch.zeros(B, N, h, w, device="cuda")
    if side == 138:
        m[:] = noise
    elif side:
        m[:, :, 20:20 + side, 30:30 + side] = noise[20:20 + side, 30:30 + side]
    r = consumer.masks_to_rle(m, nd, (H, W), rle_cap=40000)
    cap = int(r["counts"].shape[2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        consumer.masks_to_rle(m, nd, (H, W), rle_cap=cap)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / 5:8.3f} ms   max runs {int(r['len'].max())}")
