"""Sweep the age-group work-list shares of the z-buffer kernels (fwd / bwd) at batch 256."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import ops
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
def t_us(fn, reps=400):
    """device time per call: a captured graph of `reps` launches between two events"""
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
depth, owner = ops.sphere_raster_fwd(spheres, 128, 128, want_argmin=True)
from spherehand_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
gs = torch.empty(256, 41, 4, device=dev)
def pack(a): return a[0] | a[1] << 8 | a[2] << 16 | a[3] << 24
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cands = [(64, 64, 64, 64), (72, 66, 62, 56), (76, 68, 60, 52), (80, 70, 58, 48), (84, 70, 58, 44), (88, 72, 56, 40),
         (80, 72, 64, 40), (84, 74, 64, 34), (88, 76, 64, 28), (90, 78, 66, 22), (76, 70, 66, 44), (72, 70, 68, 46),
         (92, 80, 60, 24), (86, 80, 70, 20), (96, 80, 64, 16), (88, 78, 60, 30), (84, 76, 60, 36), (88, 72, 60, 36),
         (80, 72, 62, 42), (84, 72, 64, 36), (80, 76, 64, 36), (76, 72, 64, 44), (84, 78, 56, 38), (90, 74, 58, 34)]
for c in cands:
    if which == "fwd":
        ops.set_tuning(ops.TUNE_FWD_SHARES, pack(c))
        t = t_us(lambda: L.shr_sphere_raster_fwd(spheres.data_ptr(), 256, 41, 128, 128, depth.data_ptr(), owner.data_ptr(), st))
    else:
        ops.set_tuning(ops.TUNE_BWD_SHARES, pack(c))
        t = t_us(lambda: L.shr_sphere_raster_bwd(spheres.data_ptr(), grad.data_ptr(), owner.data_ptr(), 256, 41, 128, 128, gs.data_ptr(), st))
    print(which, c, "%.2f us" % t)
