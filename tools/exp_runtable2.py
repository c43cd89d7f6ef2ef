#!/usr/bin/env python3
"""Batch-256 forward / step with the run table, over work-list shares (argv: hex share words)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spherehand_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
S, J, n = 128, 41, 256
sph, grad = bench.make_inputs(0, dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    depth = torch.empty(n, S, S, device=dev)
    owner = torch.full((n, S, S), 254, device=dev, dtype=torch.uint8)
    gs = torch.empty(n, J, 4, device=dev)
    p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
    for mode in (0, -1):
        ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, mode)
        for sh in ([0x24344464] + [int(a, 16) for a in sys.argv[1:]] if mode else [0x24344464]):
            ops.set_tuning(ops.TUNE_FWD_SHARES, sh)
            r = []
            for flags in (0, 1):
                f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], flags, s), stream, 400, 5, 5, warm_ms=40.0)

                def step(s):
                    lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], flags, s)
                    lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s)
                st = bench.mean_launch_us(step, stream, 400, 5, 5, warm_ms=40.0)
                r += [f, st]
            print("table %2d shares %08x: full map fwd %.2f step %.2f | touched rows fwd %.2f step %.2f us" % (mode, sh, *r), flush=True)
ops.set_tuning(ops.TUNE_FWD_SHARES, 0x24344464)
ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, -1)
