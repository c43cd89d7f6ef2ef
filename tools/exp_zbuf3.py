"""fwd: sweep waves per workgroup x LDS cap (rows per region)."""
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
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=300):
    for _ in range(30): fn()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 2)
f_noarg = lambda: lib.shr_sphere_raster_fwd(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), None, st)
f_arg = lambda: lib.shr_sphere_raster_fwd(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), owner.data_ptr(), st)
for nw in (16, 8, 4, 2):
    ops.set_tuning(ops.TUNE_FWD_WAVES, nw)
    r1, r2 = {}, {}
    for kb in (6, 12, 22, 40, 80, 160):
        ops.set_tuning(ops.TUNE_FWD_LDS_BYTES, kb * 1024); r1[kb] = timeit(f_noarg)
        ops.set_tuning(ops.TUNE_FWD_OWNER_LDS_BYTES, kb * 1024); r2[kb] = timeit(f_arg)
    print("waves=%2d depth-only {ldsKB: us} %s" % (nw, r1))
    print("waves=%2d depth+owner         %s" % (nw, r2))
