import torch


def timeit(fn, reps=20):
    for _ in range(3):
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
