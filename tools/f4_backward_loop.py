"""Diagnostic: runs the loss mask assembly backward 20 times on boxes of 10-60 % of the image side (for rocprofv3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import consumer
B, N, ph, pw, H, W = 32, 100, 138, 138, 550, 550
g = torch.Generator().manual_seed(0)
proto = torch.relu(torch.randn(B, ph, pw, 32, generator=g)).cuda()
coef = (0.3 * torch.tanh(torch.randn(B, N, 32, generator=g))).cuda()
hh = (0.1 + 0.5 * torch.rand(B, N, generator=g)) * H; ww = (0.1 + 0.5 * torch.rand(B, N, generator=g)) * W
y1 = torch.rand(B, N, generator=g) * (H - hh); x1 = torch.rand(B, N, generator=g) * (W - ww)
boxes = torch.stack([y1, x1, y1 + hh, x1 + ww], -1).cuda().contiguous()
n = torch.full((B,), N, dtype=torch.int32, device="cuda")
m = consumer.assemble_loss_masks(proto, coef, boxes, n, (H, W))
grad = torch.randn_like(m)
for _ in range(20):
    consumer.assemble_loss_masks_backward(proto, coef, boxes, n, m, grad, (H, W))
torch.cuda.synchronize()
