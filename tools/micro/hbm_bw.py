import torch, time
dev = torch.device("cuda")
for mb in (134, 536, 2147):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev).normal_()
    b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 8 * n), ("read(sum)", lambda: a.sum(), 4 * n), ("write(fill)", lambda: b.fill_(1.0), 4 * n),
                             ("axpy", lambda: b.add_(a), 12 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb} MB {name}: {ms*1e3:.1f} us  {bytes_/ms/1e9:.2f} TB/s")
