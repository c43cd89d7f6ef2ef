import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from spherehand_amd import ops, _lib
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
depth, owner = ops.sphere_raster_fwd(spheres, 128, 128, want_argmin=True)
def t_us(fn, reps=400):
    for _ in range(20): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for w in (16, 12, 8, 4):
    ops.set_tuning(ops.TUNE_FWD_WAVES, w)
    print("fwd waves", w, "%.2f us" % t_us(lambda: L.shr_sphere_raster_fwd(spheres.data_ptr(), 256, 41, 128, 128, depth.data_ptr(), owner.data_ptr(), st)))
