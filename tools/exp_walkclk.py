"""Where a wave's scan time goes: cycles from the top of a run to its chunk loop (set-up) and inside the chunk loops,
per wave, from clock64 stamps inside walk_slice (-DSHR_EXP_WALKCLK builds of tools/exp_ztime.hip)."""
import ctypes, os, sys
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
lib = ctypes.CDLL(os.path.join(ROOT, "tools", os.environ.get("EXPLIB", "libexpt.so")))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_zfwd_t_launch.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, ci, vp, vp]
SHARES = int(os.environ.get('SHARES', '0x24344464'), 16)
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
N, J, S = 256, 41, 128
depth = torch.empty(N, S, S, device=dev)
owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8)
tbuf = torch.zeros(N * 129 + N * 64, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(50):
    lib.exp_zfwd_t_launch(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), owner.data_ptr(), 128, SHARES, tbuf.data_ptr(), st)
torch.cuda.synchronize()
t = tbuf.cpu().numpy()
w = t[N * 129:].reshape(N, 16, 4).astype(np.float64)
tt = t[:N * 128].reshape(N, 16, 8).astype(np.float64)
scan = (tt[:, :, 4] - tt[:, :, 7])
print("scan cycles by wave      :", np.round(scan.mean(0)).astype(int).tolist())
print("set-up cycles by wave    :", np.round(w[:, :, 0].mean(0)).astype(int).tolist())
print("chunk-loop cycles by wave:", np.round(w[:, :, 1].mean(0)).astype(int).tolist())
print("runs by wave             :", np.round(w[:, :, 2].mean(0), 2).tolist())
print("pair iterations by wave  :", np.round(w[:, :, 3].mean(0), 2).tolist())
print("per run set-up %.0f cycles, per pair iteration %.0f cycles; runs per crop %.1f, pair iterations per crop %.1f" % (
    w[:, :, 0].sum() / w[:, :, 2].sum(), w[:, :, 1].sum() / w[:, :, 3].sum(), w[:, :, 2].sum() / N, w[:, :, 3].sum() / N))
