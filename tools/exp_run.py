"""Ablation timing of experimental kernel variants (tools/exp_sphere.hip) on the bench workload."""
import ctypes, os, sys, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
so = os.path.join(ROOT, "tools", "libexp.so")
lib = ctypes.CDLL(so)
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_fwd_launch.argtypes = [vp, ci, ci, ci, ci, vp, ci, ci, ci, vp]
lib.exp_bwd_launch.argtypes = [vp, vp, ci, ci, ci, ci, vp, ci, ci, vp]
lib.exp_fill_launch.argtypes = [vp, ctypes.c_size_t, ci, ci, vp]
lib.exp_sum_launch.argtypes = [vp, ctypes.c_size_t, vp, ci, ci, vp]
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
N, J, S = 256, 41, 128
depth = torch.empty(N, S, S, device=dev)
gs = torch.empty(N, J, 4, device=dev)
scratch = torch.empty(16, device=dev)
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
    return best

print("fill 16.8MB:", {(b, t): round(timeit(lambda: lib.exp_fill_launch(depth.data_ptr(), depth.numel(), b, t, st)), 2)
                       for b, t in [(256, 1024), (1024, 256), (2048, 256), (4096, 256), (16384, 256)]})
print("sum  16.8MB:", {(b, t): round(timeit(lambda: lib.exp_sum_launch(grad.data_ptr(), grad.numel(), scratch.data_ptr(), b, t, st)), 2)
                       for b, t in [(256, 1024), (1024, 256), (2048, 256), (4096, 256)]})
for nw, sl in [(4, 16), (16, 1), (16, 4), (8, 8), (2, 32), (1, 64), (4, 4), (8, 2)]:
    row = {}
    for mode in (0, 1, 2, 3, 4, 5):
        row[mode] = round(timeit(lambda: lib.exp_fwd_launch(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), nw, sl, mode, st)), 2)
    print("fwd nwaves=%d slices=%d modes{0 full,1 nomin,2 noLDS,3 nomin+noLDS,4 nostore,5 nomin+nostore}:" % (nw, sl), row)
for nw in (16, 8, 4):
    row = {}
    for mode in (0, 1, 17, 2, 4):
        row[mode] = round(timeit(lambda: lib.exp_bwd_launch(spheres.data_ptr(), grad.data_ptr(), N, J, S, S, gs.data_ptr(), nw, mode, st)), 2)
    print("bwd nwaves=%d modes{0 full,1 loadonly(cand tiles),17 loadonly(all tiles),2 noreduce,4 ldsatomic}:" % nw, row)
