#!/usr/bin/env python3
"""tools/exp_storewave.hip driver: issue rate of streaming waves beside VALU-busy waves (see the .hip header).
Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/libexpsw.so tools/exp_storewave.hip"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpsw.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.storewave_launch.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
dev = torch.device("cuda:0")
out = torch.empty(256 * 64 * 64 * 4, device=dev)
tbuf = torch.zeros(256 * 16 * 2, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(S, nst, iters, mode, every=0, reps=30):
    for _ in range(5):
        lib.storewave_launch(out.data_ptr(), tbuf.data_ptr(), S, nst, iters, mode, every, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.storewave_launch(out.data_ptr(), tbuf.data_ptr(), S, nst, iters, mode, every, st)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    t = tbuf.cpu().numpy().reshape(256, 16, 2).astype(np.float64)
    d = t[:, :, 1] - t[:, :, 0]
    if mode & 8 or S == 0:
        stream_c, comp_c = float("nan"), d.mean()
    else:
        sw = slice(0, S) if mode & 4 else slice(16 - S, 16)
        cw = slice(S, 16) if mode & 4 else slice(0, 16 - S)
        stream_c, comp_c = d[:, sw].mean(), (d[:, cw].mean() if S < 16 else float("nan"))
    kb = (S if not mode & 8 else 16) * nst
    print("S %2d nst %3d iters %4d mode %2d every %2d | kernel %6.2f us | streaming waves %7.0f cycles (%.1f B/clk/CU) | compute waves %7.0f cycles | %d KB per CU"
          % (S, nst, iters, mode, every, us, stream_c, kb * 1024 / stream_c if stream_c == stream_c else 0, comp_c, kb))


print("-- compute only")
run(0, 0, 200, 0)
print("-- 32 KB per CU by S streaming waves, idle SIMDs")
for S in (1, 2, 4, 8, 16):
    run(S, 32 // S, 0, 0)
print("-- the same beside 200 rounds of FMAs on the other waves")
for mode in (0, 1, 4, 5, 2):
    for S in (1, 2, 4):
        run(S, 32 // S, 200, mode)
print("-- the compute waves issue the stores themselves (2 per wave = 32 KB), every k rounds")
for every in (10, 50, 90):
    run(0, 2, 200, 8, every)
