import ctypes, os, sys
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
lib = ctypes.CDLL(os.path.join(ROOT, "tools", os.environ.get("EXPLIB", "libexpt.so")))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_zfwd_t_launch.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, ci, vp, vp]
SHARES = int(os.environ.get('SHARES', '0x40404040'), 16)
dev = torch.device("cuda:0")
spheres, grad = bench.make_inputs(0, dev)
N, J, S = 256, 41, 128
depth = torch.empty(N, S, S, device=dev)
owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8)
tbuf = torch.zeros(N * 16 * 8 + N, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(50):
    lib.exp_zfwd_t_launch(spheres.data_ptr(), N, J, S, S, depth.data_ptr(), owner.data_ptr(), 128, SHARES, tbuf.data_ptr(), st)
torch.cuda.synchronize()
npass = tbuf.cpu().numpy()[N * 128:]
print('box rows*1000 + columns per crop (first 12):', npass[:12].tolist(), ' mean rows %.1f columns %.1f' % ((npass // 1000).mean(), (npass % 1000).mean()))
t = tbuf.cpu().numpy()[:N * 128].reshape(N, 16, 8).astype(np.float64)
base = t[:, :, 0].min()
print("clock64 ticks are 100 MHz (10 ns) on gfx9 s_memtime? checking span:", (t[:, :, 6].max() - base))
names = ["T1 spheres+list (wave0)", "T2 init done", "T3 after barrier1", "T4 raster done", "T5 after barrier2", "T6 end"]
for i, nm in enumerate(names, 1):
    d = t[:, :, i] - t[:, :, 0]
    print("%-26s mean %8.0f  min %8.0f  max %8.0f (ticks since block start)" % (nm, d.mean(), d.min(), d.max()))
print("block start skew (ticks): max-min over blocks", t[:, 0, 0].max() - t[:, 0, 0].min())
d = t[:, :, 7] - t[:, :, 0]
print("%-26s mean %8.0f  min %8.0f  max %8.0f" % ("T3b background issued", d.mean(), d.min(), d.max()))
print("raster per wave: mean %.0f max-in-block mean %.0f" % ((t[:, :, 4] - t[:, :, 7]).mean(), (t[:, :, 4] - t[:, :, 7]).max(1).mean()))
print("kernel span first start -> last end:", t[:, :, 6].max() - t[:, :, 0].min())
r = (t[:, :, 4] - t[:, :, 7])
print("raster cycles by wave index (mean over crops):", np.round(r.mean(0)).astype(int).tolist())
print("raster cycles by wave index (max over crops): ", np.round(r.max(0)).astype(int).tolist())
print("crop 0 per wave:", r[0].astype(int).tolist())
print("crop 1 per wave:", r[1].astype(int).tolist())
w = (t[:, :, 6] - t[:, :, 5])
print("writeout cycles by wave (mean):", np.round(w.mean(0)).astype(int).tolist())
i = (t[:, :, 2] - t[:, :, 1])
print("init cycles by wave (mean):", np.round(i.mean(0)).astype(int).tolist())
# absolute times inside a workgroup (relative to its earliest wave start)
b0 = t[:, :, 0].min(1, keepdims=True)
for i, nm in [(0, "T0 wave start"), (1, "T1 records requested"), (5, "T1b z-buffer init done"), (2, "T2 pre-barrier work done"), (3, "T3 after barrier1"), (7, "T3b scan starts"), (4, "T4 scan done"), (6, "T6 end")]:
    print("%-26s by wave (mean, since the workgroup's first wave start):" % nm, np.round((t[:, :, i] - b0).mean(0)).astype(int).tolist())
