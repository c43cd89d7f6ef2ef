#!/usr/bin/env python3
"""The headline step launched from a C loop (tools/cloop.c), timed with HIP events; run it unprofiled and under
`rocprofv3 --kernel-trace --stats` to compare the tracer's per-kernel averages with the launch-to-launch means when
the host is not what the launches wait for."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spherehand_amd import _lib  # noqa: E402

so = os.path.join(ROOT, "tools", "libcloop.so")
if not os.path.exists(so):
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "cloop.c"), "-ldl"])
_lib.lib()
cl = ctypes.CDLL(so)
vp, ci = ctypes.c_void_p, ctypes.c_int
cl.cloop_run.argtypes = [ctypes.c_char_p, vp, ci, ci, ci, ci, vp, vp, vp, vp, ci, ci, ci, vp]
dev = torch.device("cuda", 0)
S, J, n = 128, 41, 256
sph, grad = bench.make_inputs(0, dev)
stream = torch.cuda.Stream(device=dev)
lib_path = os.path.join(ROOT, "spherehand_amd", "libspherehand_hip.so").encode()
with torch.cuda.stream(stream):
    depth = torch.empty(n, S, S, device=dev)
    owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
    gs = torch.empty(n, J, 4, device=dev)

    def run(steps, what):
        e = cl.cloop_run(lib_path, sph.data_ptr(), n, J, S, S, depth.data_ptr(), owner.data_ptr(), grad.data_ptr(), gs.data_ptr(), 1,
                         steps, what, stream.cuda_stream)
        assert e == 0, e

    run(3000, 3)                      # clocks
    stream.synchronize()
    for name, what in (("step (fwd + bwd)", 3), ("fwd only", 1), ("bwd only", 2)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        run(2000, what)
        e1.record(stream)
        e1.synchronize()
        print("%-18s %.3f us per iteration (2000 iterations from C, HIP events)" % (name, e0.elapsed_time(e1) * 1e3 / 2000), flush=True)
