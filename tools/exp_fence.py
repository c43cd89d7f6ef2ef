#!/usr/bin/env python3
"""What the driver's short run (--steps 20 --warmup 5) pays besides the kernels: the timed region of bench.py is
20 steps = 40 launches (~300 us), so a fixed cost of 30 us (the host waking up from hipStreamSynchronize, the first
launch on an idle queue) is 1.5 us per step.  Compares fence variants on the same step:
  sync        torch.cuda.synchronize() only (round 2)
  spin+sync   poll hipStreamQuery until the stream is idle, then synchronize (returns at once)
and prints the host cost of a step's two C-ABI calls."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spherehand_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.lib()
spheres, grad = bench.make_inputs(0, dev)
N, S, J = bench.BATCH, bench.S, bench.J
depth = torch.empty(N, S, S, device=dev)
owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8)
gs = torch.empty(N, J, 4, device=dev)
stream = torch.cuda.Stream(device=dev)
sp, gp, dp, op, ap = (t.data_ptr() for t in (spheres, grad, depth, gs, owner))
sh = stream.cuda_stream
f_fwd, f_bwd = lib.shr_sphere_raster_fwd, lib.shr_sphere_raster_bwd


def step():
    f_fwd(sp, N, J, S, S, dp, ap, sh)
    f_bwd(sp, gp, ap, N, J, S, S, op, sh)


def fence_sync():
    torch.cuda.synchronize(dev)


def fence_spin():
    while not stream.query():
        pass
    torch.cuda.synchronize(dev)


def run(fence, steps, warmup, reps):
    out = []
    for _ in range(reps):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        out.append((time.perf_counter() - t0) / steps * 1e6)
    return out


with torch.cuda.stream(stream):
    for _ in range(200):
        step()
    torch.cuda.synchronize(dev)
    # host cost of enqueuing (queue kept short: sync every 8 steps)
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    host = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize(dev)
    print("host enqueue of one step (two C-ABI calls, deep queue): %.2f us" % host)
    for steps, warm in ((20, 5), (200, 20), (2000, 200)):
        for name, fence in (("sync", fence_sync), ("spin+sync", fence_spin)):
            r = run(fence, steps, warm, 30 if steps <= 200 else 5)
            print("steps %4d  %-9s  us/step: median %.2f  mean %.2f  min %.2f  max %.2f"
                  % (steps, name, statistics.median(r), statistics.mean(r), min(r), max(r)))
    # the same after host-side idling (the driver's process does imports / setup right before)
    for name, fence in (("sync", fence_sync), ("spin+sync", fence_spin)):
        r = []
        for _ in range(10):
            time.sleep(0.05)
            r += run(fence, 20, 5, 1)
        print("steps   20  %-9s after 50 ms idle: median %.2f  min %.2f  max %.2f" % (name, statistics.median(r), min(r), max(r)))
