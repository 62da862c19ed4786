"""Diagnostic: mask_rle with bitmaps for output widths with and without 4-byte aligned rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import consumer
B, N, h, w = 32, 100, 138, 138
nd = torch.full((B,), N, dtype=torch.int32, device="cuda")
g = torch.Generator().manual_seed(0)
m = torch.zeros(B, N, h, w, device="cuda")
m[:, :, 20:44, 30:54] = torch.rand(24, 24, generator=g).cuda()
for (H, W) in ((550, 550), (550, 552), (552, 552), (550, 548)):
    for bm in (False, True):
        r = consumer.masks_to_rle(m, nd, (H, W), rle_cap=4000, bitmaps=bm)
        cap = int(r["counts"].shape[2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            consumer.masks_to_rle(m, nd, (H, W), rle_cap=cap, bitmaps=bm)
        e1.record(); torch.cuda.synchronize()
        print(f"{H}x{W} bitmaps={bm}: {e0.elapsed_time(e1) / 5:7.3f} ms")
