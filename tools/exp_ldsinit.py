#!/usr/bin/env python3
"""tools/exp_ldsinit.hip driver.  Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/libexpli.so tools/exp_ldsinit.hip"""
import ctypes, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpli.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.ldsinit_launch.argtypes = [vp, ci, ci, ci, ci, vp, vp]
dev = torch.device("cuda:0")
tbuf = torch.zeros(256 * 16 * 2, dtype=torch.int64, device=dev)
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
for mode in (0, 1, 2, 3):
    for nw in (16, 8, 4):
        for kb in (139, 70, 35):
            for _ in range(20):
                lib.ldsinit_launch(tbuf.data_ptr(), nw, kb, mode, 30, sink.data_ptr(), st)
            torch.cuda.synchronize()
            t = tbuf.cpu().numpy().reshape(256, 16, 2).astype(np.float64)
            d = (t[:, 16 - nw:, 1] - t[:, 16 - nw:, 0])
            span = (t[:, 16 - nw:, 1].max(1) - t[:, :, 0].min(1)).mean()
            print("mode %d (%s%s) nw %2d kb %3d: per-wave %6.0f cycles, workgroup span %6.0f -> %.1f B/clk"
                  % (mode, "b64" if mode & 1 else "b128", ", others FMA" if mode & 2 else "", nw, kb, d.mean(), span, kb * 1024 / span))
