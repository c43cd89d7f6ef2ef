"""Host cost of one bench step (two C-ABI launches through ctypes) against its device time; graphs of M steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spherehand_amd import _lib
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
L = _lib.lib()
N, J, S = 256, 41, 128
depth = torch.empty(N, S, S, device=dev); owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8); gs = torch.empty(N, J, 4, device=dev)
stream = torch.cuda.Stream()
sp, gp, dp, op, ap = spheres.data_ptr(), grad.data_ptr(), depth.data_ptr(), gs.data_ptr(), owner.data_ptr()
def step(s):
    L.shr_sphere_raster_fwd(sp, N, J, S, S, dp, ap, s); L.shr_sphere_raster_bwd(sp, gp, ap, N, J, S, S, op, s)
with torch.cuda.stream(stream):
    sh = stream.cuda_stream
    for _ in range(200): step(sh)
    torch.cuda.synchronize()
    for K in (200, 2000):
        t0 = time.perf_counter()
        for _ in range(K): step(sh)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("K=%d: host submission %.2f us/step, submission + drain %.2f us/step" % (K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    for M in (1, 4, 10, 20, 50):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(M): step(sh)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        R = 2000 // M
        t0 = time.perf_counter()
        for _ in range(R): g.replay()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("graph of %d steps: %.2f us/step" % (M, (t2 - t0) / (R * M) * 1e6))
