import torch, time
for mb in (244, 731, 1258):
    x = torch.empty(mb * 1000 * 1000 // 4, dtype=torch.float32, device="cuda")
    for _ in range(3): x.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): x.zero_()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fill {mb} MB: {ms*1e3:.1f} us  {mb/ms/1e3:.2f} TB/s")
