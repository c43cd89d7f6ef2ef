"""LDS-cap choice vs batch size (crops per launch)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, ops
lib = _lib.lib()
dev = torch.device("cuda:0")
sp256, g256 = bench.make_inputs(0, dev)
J, S = 41, 128
def timeit(fn, reps=100):
    for _ in range(10): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 2)
st = torch.cuda.current_stream().cuda_stream
for mult in (1, 2, 4, 9, 36):
    N = 256 * mult
    spheres = sp256.repeat(mult, 1, 1).contiguous(); grad = g256.repeat(mult, 1, 1).contiguous()
    depth = torch.empty(N, S, S, device=dev); owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8)
    gs = torch.empty(N, J, 4, device=dev)
    f = lambda: lib.shr_sphere_raster_fwd(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), owner.data_ptr(), st)
    b = lambda: lib.shr_sphere_raster_bwd(spheres.data_ptr(), grad.data_ptr(), owner.data_ptr(), N, J, S, S, gs.data_ptr(), st)
    row = {}
    for kb in (80, 160):
        ops.set_tuning(ops.TUNE_FWD_OWNER_LDS_BYTES, kb * 1024); row["fwd%d" % kb] = timeit(f)
    for kb in (64, 128):
        ops.set_tuning(ops.TUNE_BWD_LDS_BYTES, kb * 1024); row["bwd%d" % kb] = timeit(b)
    print("N=%5d us/launch %s  -> us per 256 crops: %s" % (N, row, {k: round(v / mult, 2) for k, v in row.items()}))
