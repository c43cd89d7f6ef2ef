import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spherehand_amd import ops
def run(sp, H, W):
    return ops.sphere_raster_fwd(torch.from_numpy(np.asarray(sp, np.float32)).cuda(), H, W).cpu().numpy()
print("J=1 nan x:", run([[[np.nan, 0, 0, 5]]], 8, 32)[0, 0, :6])
print("J=2 [ok, nan x]:", run([[[0, 0, 10, 20], [np.nan, 0, 0, 5]]], 8, 32)[0, 0, :6])
print("J=2 [nan x, ok]:", run([[[np.nan, 0, 0, 5], [0, 0, 10, 20]]], 8, 32)[0, 0, :6])
print("J=2 [ok, nan r]:", run([[[0, 0, 10, 20], [0, 0, 0, np.nan]]], 8, 32)[0, 0, :6])
print("J=2 [ok, nan y]:", run([[[0, 0, 10, 20], [0, np.nan, 0, 5]]], 8, 32)[0, 0, :6])
print("J=3 [ok, nan x, far]:", run([[[0, 0, 10, 20], [np.nan, 0, 0, 5], [1000, 0, 0, 5]]], 8, 32)[0, 0, :6])
sp = np.array([[[0, 0, 10, 20], [np.nan, 0, 0, 5]],
               [[0, 0, np.nan, 20], [50, 50, 0, 5]],
               [[np.inf, 0, 0, 20], [0, 0, 5, 30]],
               [[0, 0, 5, np.inf], [0, 0, 5, 30]]], np.float32)
d = run(sp, 16, 32)
print("test arr crop0 nan count", np.isnan(d[0]).sum(), "of", d[0].size, "rows with nan", np.isnan(d[0]).any(1))
d = run(sp[:1], 16, 32); print("crop0 alone H=16:", np.isnan(d[0]).sum())
d = run(sp[:1], 8, 32); print("crop0 alone H=8:", np.isnan(d[0]).sum())
d = run(sp[:1], 16, 64); print("crop0 alone 16x64:", np.isnan(d[0]).sum(), np.isnan(d[0]).any(0))
