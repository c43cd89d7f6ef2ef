"""The launch floor of the sphere forward's shape at MORE crops than CUs (shr_selftest_launch_floor: records in, depth +
the touched rows' owner bytes out, no arithmetic): what fraction of the HBM peak a pure mover of the forward's bytes
reaches at 1152 / 9216 crops @128x128 with one / two / four workgroups per CU -- the ceiling of `roofline.large_batch`.
    python tools/exp_floor_large.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from spherehand_amd import _lib
from bench import mean_launch_us, roof
lib = _lib.lib()
dev = torch.device("cuda")
stream = torch.cuda.Stream()
S, J = 128, 41
with torch.cuda.stream(stream):
    for n in (256, 1152, 9216):
        sph = torch.randn(n, J, 4, device=dev)
        depth = torch.empty(n, S, S, device=dev)
        owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        p = [t.data_ptr() for t in (sph, depth, owner)]
        rows = 64
        nbytes = n * (4 * S * S + rows * S + 16 * J)
        for lds in (160 * 1024, 80 * 1024, 40 * 1024):
            t = mean_launch_us(lambda s: _lib.check(lib.shr_selftest_launch_floor(p[0], n, J, S, S, 32, 32 + rows, p[1], p[2], lds, s), "floor"),
                               stream, 40 if n < 9216 else 10, 4, 3, warm_ms=30.0)
            print("floor: %5d crops, %3d KB of LDS per workgroup (%d per CU): %.2f us = %.3f of the peak (%.2f us per 256 crops)" %
                  (n, lds // 1024, 160 * 1024 // lds, t, roof(nbytes, t)["frac"], t * 256 / n), flush=True)
        t = mean_launch_us(lambda _s: depth.fill_(100.0), stream, 40 if n < 9216 else 10, 4, 3)
        print("torch fill of the depth output alone: %.2f us = %.3f of the peak" % (t, roof(n * 4 * S * S, t)["frac"]))
        t = mean_launch_us(lambda _s: depth.copy_(depth2) if False else None, stream, 1, 1, 0) if False else 0
        del sph, depth, owner
