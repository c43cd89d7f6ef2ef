import ctypes, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpl.so"))
ci = ctypes.c_int
lib.run.argtypes = [ci, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
torch.zeros(1, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t(which, grid, block, lds, cycles=0, reps=500):
    ms = ctypes.c_float()
    lib.run(which, grid, block, lds, cycles, reps, ctypes.byref(ms), st)
    return ms.value / reps * 1e3
for grid, block, lds in ((256, 1024, 150 * 1024), (256, 1024, 0), (256, 256, 0), (1, 64, 0), (512, 1024, 75 * 1024), (256, 512, 150 * 1024)):
    print("empty  grid %4d block %4d lds %6d: %.2f us/launch" % (grid, block, lds, t(0, grid, block, lds)))
for cyc in (2000, 6000, 12000, 20000):
    print("spin %5d cycles grid 256 block 1024 lds 150K: %.2f us/launch" % (cyc, t(1, 256, 1024, 150 * 1024, cyc)))
