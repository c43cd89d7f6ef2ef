"""Practical HBM ceilings on this box next to the 8 TB/s spec peak: device-wide fill and copy rates at
several sizes, from 16.8 MB (the depth output of one 256-crop forward launch) to 2 GB."""
import torch
dev = torch.device("cuda:0")
def t_us(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for mb in (16.8, 67, 268, 1074, 2147):
    n = int(mb * 1e6 / 4)
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    reps = max(3, int(2000 / mb))
    tf = t_us(lambda: a.fill_(100.0), reps)
    tc = t_us(lambda: b.copy_(a), reps)
    print("%7.1f MB: fill %8.1f us = %.2f TB/s written ; copy %8.1f us = %.2f TB/s (read + written)"
          % (mb, tf, mb / tf, tc, 2 * mb / tc))
    del a, b
