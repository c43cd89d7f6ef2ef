#!/usr/bin/env python3
"""data_to_model_kernel at config 5's per-GPU crop count for every (parts per crop, waves per workgroup): us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tools.bench_d2m import inputs  # noqa: E402
from spherehand_amd import _lib, ops  # noqa: E402
lib = _lib.lib()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B, S in ((128, 128), (128, 256), (1024, 128)):
        obs, index, cen, rad = inputs(B, S)
        N, J = B * 9, 41
        for parts, band in [(p_, b_) for p_ in (1, 2, 4) for b_ in [int(x) for x in os.environ.get("BANDS", "0").split(",")]]:
            ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, band)
            line = "%5d crops @%d, %d part(s), band %d:" % (N, S, parts, band)
            for waves in [int(x) for x in os.environ.get("WAVES", "4,8,16").split(",")]:
                ops.set_tuning(ops.TUNE_D2M_WAVES, waves)
                ls = torch.empty(N * parts, device="cuda"); gr = torch.empty(N * parts, J, 3, device="cuda")
                a = [t.data_ptr() for t in (obs, index, cen, rad, ls, gr)]
                t = bench.mean_launch_us(lambda s: lib.shr_data_to_model_partial(a[0], a[1], a[2], 3, a[3], N, J, S, S, parts, a[4], a[5], s),
                                         stream, 20, 3, 3)
                line += "  %2d waves %.1f us" % (waves, t)
            print(line)
        ops.set_tuning(ops.TUNE_D2M_WAVES, 0)
        ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, 0)
