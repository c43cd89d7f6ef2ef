"""Does a long warm-up (clock ramp) change kernel times?"""
import os, sys, time, subprocess
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
f_arg = lambda: lib.shr_sphere_raster_fwd(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), owner.data_ptr(), st)
b_arg = lambda: lib.shr_sphere_raster_bwd(spheres.data_ptr(), grad.data_ptr(), owner.data_ptr(), N, J, S, S, gs.data_ptr(), st)
def t(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / reps, 2)
print(subprocess.run("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk|fclk' | head -4", shell=True, capture_output=True, text=True).stdout)
for reps in (50, 200, 1000, 5000, 20000, 50000):
    print("fwd reps", reps, t(f_arg, reps), "us")
print(subprocess.run("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk|fclk' | head -4", shell=True, capture_output=True, text=True).stdout)
# big matmul warmup then measure
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 3.0:
    (a @ a); torch.cuda.synchronize()
print("after 3 s of GEMM: fwd", t(f_arg, 2000), "bwd", t(b_arg, 2000))
# host launch cost: time of enqueueing only
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): f_arg()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per launch %.2f us, drain %.2f us/launch" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
