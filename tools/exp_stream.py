#!/usr/bin/env python3
"""Batch-256 forward (and forward + backward step) for every SHR_TUNE_FWD_STREAM_WAVES setting, with the full owner
map and with SHR_RASTER_OWNER_TOUCHED_ROWS; depth / owner bits checked against the setting 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spherehand_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
N, S, J = 256, 128, 41
sph, grad = bench.make_inputs(0, dev)
stream = torch.cuda.Stream(device=dev)
shares = [int(x, 16) for x in os.environ.get("SHARES", "").split(",") if x]
with torch.cuda.stream(stream):
    ref = {}
    for flags in (0, 1):
        for ns in (0, 1, 2, 3):
            for sh in (shares or [None]):
                lib.shr_set_tuning(14, ns)
                if sh is not None:
                    lib.shr_set_tuning(6, sh)
                depth = torch.empty(N, S, S, device=dev)
                owner = torch.full((N, S, S), 254, device=dev, dtype=torch.uint8)
                gs = torch.empty(N, J, 4, device=dev)
                p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
                f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd_ex(p[0], N, J, S, S, p[1], p[2], flags, s), stream, 200, 5, 20)

                def step(s):
                    lib.shr_sphere_raster_fwd_ex(p[0], N, J, S, S, p[1], p[2], flags, s)
                    lib.shr_sphere_raster_bwd(p[0], p[3], p[2], N, J, S, S, p[4], s)
                st = bench.mean_launch_us(step, stream, 200, 5, 20)
                stream.synchronize()
                key = (flags,)
                if key not in ref:
                    ref[key] = (depth.clone(), owner.clone(), gs.clone())
                ok = torch.equal(depth, ref[key][0]) and torch.equal(owner, ref[key][1]) and torch.equal(gs, ref[key][2])
                print("flags %d stream waves %d shares %s: fwd %.2f us, fwd+bwd %.2f us, identical to setting 0: %s"
                      % (flags, ns, hex(sh) if sh is not None else "default", f, st, ok))
