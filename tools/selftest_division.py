import os, sys, torch
sys.path.insert(0, os.getcwd())
from spherehand_amd import _lib
lib = _lib.lib()
for seed in (1, 2, 3, 12345):
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(lib.shr_selftest_division(seed, 2000, bad.data_ptr(), torch.cuda.current_stream().cuda_stream), "st")
    torch.cuda.synchronize()
    print("seed", seed, "cases", 4096 * 256 * 2000, "mismatches", int(bad.item()))
