"""Does the forward time depend on where depth / owner map / spheres sit in memory?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
N, S = 256, 128
pool = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
base = (pool.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
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
print("torch-allocated spheres at %x (mod 2MB %x)" % (spheres.data_ptr(), spheres.data_ptr() % (2 << 20)))
MB = 1 << 20
for doff, ooff in ((0, 32 * MB), (0, 17 * MB), (0, 16 * MB + 4096), (0, 16 * MB + 64 * 1024), (4096, 32 * MB), (512, 32 * MB + 512),
                   (0, 20 * MB), (1 * MB, 40 * MB), (0, 16 * MB), (0, 24 * MB), (256, 32 * MB + 1024)):
    d, o = base + doff, base + ooff
    t = t_us(lambda: L.shr_sphere_raster_fwd(spheres.data_ptr(), N, 41, S, S, d, o, st))
    print("depth +%9d  owner +%9d : %.2f us" % (doff, ooff, t))
for trial in range(6):
    junk = torch.empty((trial * 3 + 1) * 12345, device=dev)
    dd = torch.empty(N, S, S, device=dev); oo = torch.empty(N, S, S, dtype=torch.uint8, device=dev)
    t = t_us(lambda: L.shr_sphere_raster_fwd(spheres.data_ptr(), N, 41, S, S, dd.data_ptr(), oo.data_ptr(), st))
    print("torch alloc: depth %x owner %x : %.2f us" % (dd.data_ptr(), oo.data_ptr(), t))
