import torch, time
dev = torch.device("cuda:0")
for mb in (302, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device=dev).normal_(); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("copy %5d MB: %.3f ms -> %.2f TB/s (read+write)" % (mb, ms, 2 * n * 8 / ms / 1e9))
    # read-only reduction
    for _ in range(2): a.sum()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("sum  %5d MB: %.3f ms -> %.2f TB/s (read)" % (mb, ms, n * 8 / ms / 1e9))
