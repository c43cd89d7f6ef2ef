import ctypes, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexps.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.launch_store.argtypes = [vp, vp, ci, ci, ci, ci, vp]
dev = torch.device("cuda:0")
out = torch.empty(256 * 16 * 64 * 64 * 4, device=dev)   # up to 64 stores per wave
tb = torch.zeros(256 * 16 * 4, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for grid in (1, 256):
    for mode in (0, 1):
        for active in (1, 4, 16):
            for nstore in (1, 4, 16, 64):
                for _ in range(3):
                    lib.launch_store(out.data_ptr(), tb.data_ptr(), grid, nstore, active, mode, st)
                torch.cuda.synchronize()
                t = tb.cpu().numpy().reshape(256, 16, 4)[:grid, :active]
                issue = (t[:, :, 1] - t[:, :, 0]).mean(); done = (t[:, :, 2] - t[:, :, 0]).mean()
                byt = (1024 if mode == 0 else 256) * nstore * active
                print("grid %3d %s waves %2d stores/wave %2d: issue %6.0f cyc, retired %6.0f cyc  -> %.1f B/clk/CU issued, %.1f retired"
                      % (grid, "x4" if mode == 0 else "x1", active, nstore, issue, done, byt / max(issue, 1), byt / max(done, 1)))
