"""Sweep the age-group work-list shares of the z-buffer kernels (fwd / bwd) at batch 256."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import _lib
if os.environ.get("SHR_LIB"):          # an experiment build of the library (SHR_HIPCC_EXTRA=-D... python -m spherehand_amd.build)
    _lib.SO_PATH = os.environ["SHR_LIB"]
import bench
from spherehand_amd import ops
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
def t_us(fn, reps=600):
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
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
gs = torch.empty(256, 41, 4, device=dev)
def pack(a): return a[0] | a[1] << 8 | a[2] << 16 | a[3] << 24
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cands = [(64, 64, 64, 64), (72, 68, 60, 56), (56, 60, 68, 72), (48, 56, 72, 80), (60, 56, 72, 68), (76, 60, 64, 56), (80, 64, 60, 52),
         (88, 72, 56, 40), (84, 70, 58, 44), (90, 74, 58, 34), (86, 72, 58, 40), (92, 72, 54, 38), (88, 76, 54, 38),
         (84, 74, 58, 40), (80, 72, 60, 44), (88, 68, 58, 42), (92, 76, 56, 32), (86, 70, 56, 44), (82, 70, 60, 44),
         (88, 72, 60, 36), (84, 72, 56, 44), (90, 70, 56, 40), (86, 74, 56, 40)]
if os.environ.get("CANDS"):   # e.g. CANDS="100:72:50:34,96:72:52:36"
    cands = [tuple(int(v) for v in c.split(":")) for c in os.environ["CANDS"].split(",")]
for c in cands:
    if which == "fwd":
        ops.set_tuning(ops.TUNE_FWD_SHARES, pack(c))
        t = t_us(lambda: L.shr_sphere_raster_fwd(spheres.data_ptr(), 256, 41, 128, 128, depth.data_ptr(), owner.data_ptr(), st))
    else:
        ops.set_tuning(ops.TUNE_BWD_SHARES, pack(c))
        t = t_us(lambda: L.shr_sphere_raster_bwd(spheres.data_ptr(), grad.data_ptr(), owner.data_ptr(), 256, 41, 128, 128, gs.data_ptr(), st))
    print(which, c, "%.2f us" % t)
