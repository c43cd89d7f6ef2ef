"""Time the product sphere raster kernels under different LDS caps (rows per region)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, ops
lib = _lib.lib()
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
N, J, S = 256, 41, 128
depth = torch.empty(N, S, S, device=dev)
owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8)
gs = torch.empty(N, J, 4, device=dev)
st = torch.cuda.current_stream().cuda_stream

def timeit(fn, reps=200):
    for _ in range(20): fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 2)

f_noarg = lambda: lib.shr_sphere_raster_fwd(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), None, st)
f_arg = lambda: lib.shr_sphere_raster_fwd(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), owner.data_ptr(), st)
b_arg = lambda: lib.shr_sphere_raster_bwd(spheres.data_ptr(), grad.data_ptr(), owner.data_ptr(), N, J, S, S, gs.data_ptr(), st)
b_noarg = lambda: lib.shr_sphere_raster_bwd(spheres.data_ptr(), grad.data_ptr(), None, N, J, S, S, gs.data_ptr(), st)
for kb in (20, 40, 64, 100, 160):
    ops.set_tuning(ops.TUNE_FWD_LDS_BYTES, kb * 1024)
    print("fwd depth-only  lds cap %3d KB: %s us" % (kb, timeit(f_noarg)))
for kb in (24, 40, 80, 160):
    ops.set_tuning(ops.TUNE_FWD_OWNER_LDS_BYTES, kb * 1024)
    print("fwd depth+owner lds cap %3d KB: %s us" % (kb, timeit(f_arg)))
f_arg(); torch.cuda.synchronize()
for kb in (24, 48, 64, 128):
    ops.set_tuning(ops.TUNE_BWD_LDS_BYTES, kb * 1024)
    print("bwd (owner map) lds cap %3d KB: %s us" % (kb, timeit(b_arg)))
print("bwd (recompute, tile kernel): %s us" % timeit(b_noarg))
ops.set_tuning(ops.TUNE_FORCE_GENERAL, 1)
print("general tile fwd: %s us ; fwd+owner: %s us" % (timeit(f_noarg), timeit(f_arg)))
