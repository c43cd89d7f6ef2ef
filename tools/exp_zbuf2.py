import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpz.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_zfwd_launch.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, ci, vp]
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
names = {0: "full", 1: "no raster", 2: "no writeout", 3: "init only", 4: "no hits", 8: "no atomic", 16: "approx sqrt", 24: "approx sqrt+no atomic"}
for rows in (128,):
    for arg in (None, owner.data_ptr()):
        row = {names[m]: timeit(lambda: lib.exp_zfwd_launch(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), arg, rows, m, st)) for m in names}
        print("rows=%d owner=%s" % (rows, arg is not None), row)
