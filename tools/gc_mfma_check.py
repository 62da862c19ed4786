"""Experiment harness for the matrix-core grad_coef build (-DYP_EXP_GC_MFMA): runs the loss mask assembly backward of
the library named by YOLACT_POST_LIB (default: the shipped one) on boxes of 10-60 % of the image side, times it with
HIP events and saves grad_coef; with two saved files, prints how far they are apart.

    python tools/gc_mfma_check.py run gpurun_out/gc_default.npy
    YOLACT_POST_LIB=yolact_amd/libyp_gcmfma.so python tools/gc_mfma_check.py run gpurun_out/gc_mfma.npy
    python tools/gc_mfma_check.py cmp gpurun_out/gc_default.npy gpurun_out/gc_mfma.npy
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(out):
    import torch
    from yolact_amd import consumer
    B, N, ph, pw, H, W = 32, 100, 138, 138, 550, 550
    g = torch.Generator().manual_seed(0)
    proto = torch.relu(torch.randn(B, ph, pw, 32, generator=g)).cuda()
    coef = (0.3 * torch.tanh(torch.randn(B, N, 32, generator=g))).cuda()
    hh = (0.1 + 0.5 * torch.rand(B, N, generator=g)) * H
    ww = (0.1 + 0.5 * torch.rand(B, N, generator=g)) * W
    y1 = torch.rand(B, N, generator=g) * (H - hh)
    x1 = torch.rand(B, N, generator=g) * (W - ww)
    boxes = torch.stack([y1, x1, y1 + hh, x1 + ww], -1).cuda().contiguous()
    n = torch.full((B,), N, dtype=torch.int32, device="cuda")
    m = consumer.assemble_loss_masks(proto, coef, boxes, n, (H, W))
    grad = torch.randn(m.shape, generator=g).cuda()
    gp, gc = consumer.assemble_loss_masks_backward(proto, coef, boxes, n, m, grad, (H, W))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        consumer.assemble_loss_masks_backward(proto, coef, boxes, n, m, grad, (H, W))
    e0.record()
    for _ in range(20):
        consumer.assemble_loss_masks_backward(proto, coef, boxes, n, m, grad, (H, W))
    e1.record()
    torch.cuda.synchronize()
    # float64 value of the same sums, for the distance of either build from the exact result
    dz = (grad.double() * m.double()) * (1.0 - m.double())
    exact = torch.einsum("bnp,bpk->bnk", dz.reshape(B, N, -1), proto.double().reshape(B, -1, 32))
    scale = torch.einsum("bnp,bpk->bnk", dz.abs().reshape(B, N, -1), proto.double().abs().reshape(B, -1, 32))
    err = ((gc.double() - exact).abs() / scale.clamp_min(1e-30)).max().item()
    np.save(out, gc.cpu().numpy())
    print(f"lib={os.environ.get('YOLACT_POST_LIB', 'default')} backward_ms={e0.elapsed_time(e1) / 20:.4f} "
          f"max |gc - f64| / sum|terms| = {err:.3e}", flush=True)


def cmp(a, b):
    x, y = np.load(a), np.load(b)
    d = np.abs(x.astype(np.float64) - y)
    print(f"bit-identical={np.array_equal(x, y)} max_abs_diff={d.max():.3e} max_rel_to_max={d.max() / np.abs(x).max():.3e}")


if __name__ == "__main__":
    {"run": run, "cmp": cmp}[sys.argv[1]](*sys.argv[2:])
