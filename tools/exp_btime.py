import ctypes, os, sys
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import ops
lib = ctypes.CDLL(os.path.join(ROOT, "tools", os.environ.get("EXPLIB", "libexpb.so")))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_zbwd_t_launch.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, ci, ci, vp, vp]
SHARES = int(os.environ.get('SHARES', '0x2c3a4654'), 16)
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
N, J, S = 256, 41, 128
depth, owner = ops.sphere_raster_fwd(spheres, S, S, want_argmin=True)
gs = torch.empty(N, J, 4, device=dev)
tbuf = torch.zeros(N * 16 * 8, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
MODE = os.environ.get("MODE", "fwd_bwd")
for _ in range(50):
    if MODE == "fwd_bwd":
        ops.sphere_raster_fwd(spheres, S, S, want_argmin=True)   # as in the bench: forward, then backward
    lib.exp_zbwd_t_launch(spheres.data_ptr(), grad.data_ptr(), owner.data_ptr(), N, J, S, S, gs.data_ptr(), 128, SHARES, tbuf.data_ptr(), st)
torch.cuda.synchronize()
ref = ops.sphere_raster_bwd(spheres, grad, owner)
print("max |diff| vs library:", (gs - ref).abs().max().item())
t = tbuf.cpu().numpy().reshape(N, 16, 8).astype(np.float64)
names = ["T1 staging issued+list", "T2 after barrier (staged)", "T3 walk done", "T4 end"]
for i, nm in enumerate(names, 1):
    d = t[:, :, i] - t[:, :, 0]
    print("%-26s mean %8.0f  min %8.0f  max %8.0f" % (nm, d.mean(), d.min(), d.max()))
d = t[:, :, 5] - t[:, :, 0]
print("%-26s mean %8.0f  min %8.0f  max %8.0f" % ("TS records arrived", d.mean(), d.min(), d.max()))
w = t[:, :, 3] - t[:, :, 2]
print("walk cycles by wave (mean):", np.round(w.mean(0)).astype(int).tolist())
s1 = t[:, :, 1] - t[:, :, 0]
print("staging by wave (mean):", np.round(s1.mean(0)).astype(int).tolist())
b0 = t[:, :, 0].min(1, keepdims=True)
for i, nm in [(0, "T0 wave start"), (5, "TS records in (lead waves)"), (1, "T1 staged"), (2, "T2 after the barrier"), (3, "T3 walk done"), (4, "T4 end")]:
    v = t[:, :, i] - b0
    print("%-28s by wave (mean, since the workgroup's first wave start):" % nm, np.round(np.where(t[:, :, i] > 0, v, 0).mean(0)).astype(int).tolist())
