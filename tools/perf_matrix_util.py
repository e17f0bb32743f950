import torch


_warm = {"done": False}


def warm_gpu(ms=300):
    """keep the GPU busy for a while before timing: short, light runs are otherwise measured at idle clocks (the same small
    kernel was seen at 7, 10 and 18 us in three consecutive sessions)"""
    a = torch.randn(4096, 4096, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t = 0.0
    while t < ms:
        for _ in range(20):
            a = a @ a
            a = a / a.abs().max()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1)


def timeit(fn, reps=20):
    if not _warm["done"]:
        warm_gpu()
        _warm["done"] = True
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return d[len(d) // 2] * 1e3


def jl(shape, dtype):
    t = torch.randn(*reversed(shape), dtype=dtype, device="cuda")
    return t.permute(*reversed(range(len(shape)))) if len(shape) > 1 else t
