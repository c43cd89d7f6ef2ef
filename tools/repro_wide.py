"""Fused render-and-compare on wide, flat images with spheres wider than a wave (ncx > 1)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import _lib
if os.environ.get("SHR_LIB"):
    _lib.SO_PATH = os.environ["SHR_LIB"]
from oracle import oracle
from spherehand_amd import ops
oracle.build()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for seed in range(60):
    rs = np.random.RandomState(seed)
    H, W, N, J = 16, 1024, 3, 41
    scale = 160.0
    sp = np.concatenate([rs.uniform(-scale, scale, (N, J, 2)), rs.uniform(-120, 130, (N, J, 1)),
                         rs.uniform(0.02, 1.0, (N, J, 1)) * rs.choice([2.0, 12.0, 45.0, 300.0])], -1).astype(np.float32)
    tgt = rs.uniform(-50, 100, (N, H, W)).astype(np.float32)
    od, oa = oracle.sphere_raster_fwd(sp, H, W)
    dep, sse, gsp = ops.sphere_raster_mse(dev(sp), dev(tgt))
    og2 = oracle.sphere_raster_bwd(sp, (2 * (od - tgt)).astype(np.float32))
    err = np.abs(gsp.cpu().numpy() - og2).max()
    gd = rs.standard_normal((N, H, W)).astype(np.float32)
    d, a = ops.sphere_raster_fwd(dev(sp), H, W, want_argmin=True)
    gs = ops.sphere_raster_bwd(dev(sp), dev(gd), a).cpu().numpy()
    og = oracle.sphere_raster_bwd(sp, gd)
    e2 = np.abs(gs - og).max()
    if err > 2e-5 * np.abs(og2).max() + 1e-3 or e2 > 1e-5 * np.abs(og).max() + 2e-4 or not np.array_equal(dep.cpu().numpy().view(np.uint32), od.view(np.uint32)):
        bad += 1
        print("seed", seed, "fused grad err %.3g of %.3g; bwd err %.3g of %.3g" % (err, np.abs(og2).max(), e2, np.abs(og).max()))
print("bad", bad, "of 60")
