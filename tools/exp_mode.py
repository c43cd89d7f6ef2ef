"""The forward runs at ~8.05 or ~8.5 us depending on the process.  Does the mode follow the output buffer
(several live allocations timed in one process), the records buffer, or the process?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
N, S = 256, 128
def t_us(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
bufs = []
for trial in range(8):
    bufs.append((torch.empty(N, S, S, device=dev), torch.empty(N, S, S, dtype=torch.uint8, device=dev), spheres.clone()))
    if trial == 3:
        junk = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
for rnd in range(3):
    row = []
    for dd, oo, sp in bufs:
        row.append(t_us(lambda: L.shr_sphere_raster_fwd(sp.data_ptr(), N, 41, S, S, dd.data_ptr(), oo.data_ptr(), st)))
    print("round %d: " % rnd + " ".join("%.2f" % t for t in row))
print("depth  " + " ".join("%x" % (b[0].data_ptr() >> 20 & 0xfff) for b in bufs))
print("owner  " + " ".join("%x" % (b[1].data_ptr() >> 20 & 0xfff) for b in bufs))
# same depth buffer, different owner buffers / records
dd = bufs[2][0]
print("depth 2 with owners 0..7: " + " ".join("%.2f" % t_us(lambda: L.shr_sphere_raster_fwd(bufs[2][2].data_ptr(), N, 41, S, S, dd.data_ptr(), b[1].data_ptr(), st)) for b in bufs))
print("depth 2 with records 0..7: " + " ".join("%.2f" % t_us(lambda: L.shr_sphere_raster_fwd(b[2].data_ptr(), N, 41, S, S, dd.data_ptr(), bufs[2][1].data_ptr(), st)) for b in bufs))
# does the mode follow the HIP stream (hardware queue)?
dd, oo, sp = bufs[0]
for k in range(10):
    stream = torch.cuda.Stream(priority=0 if k < 8 else -1)
    with torch.cuda.stream(stream):
        h = stream.cuda_stream
        t = t_us(lambda: L.shr_sphere_raster_fwd(sp.data_ptr(), N, 41, S, S, dd.data_ptr(), oo.data_ptr(), h))
    print("stream %d (%x): %.2f us" % (k, h, t))
