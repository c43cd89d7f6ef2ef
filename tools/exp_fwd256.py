"""Depth-only forward at config 5's size (1152 crops @256x256): launch shapes (run table on / off, LDS budget)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, ops, hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
dev = torch.device("cuda", 0)
mesh = hand_model.load_mesh()
B, S, J = 128, 256, 41
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0, device=dev)
crit = MutualProjectionLoss(S, mesh).to(dev)
with torch.no_grad():
    _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + 1.0)
n = B * 9
rad = crit.data_to_model_criterion.radiuses.view(-1)
sph = torch.cat([pts.squeeze(-1).reshape(n, J, 3), rad.view(1, J, 1).expand(n, J, 1)], -1).contiguous()
dep = torch.empty(n, S, S, device=dev)
own = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
lib = _lib.lib()
stream = torch.cuda.Stream(device=dev)
ref = None
with torch.cuda.stream(stream):
    for table in (-1, 0):
        for ldsb in (80 * 1024, 160 * 1024, 120 * 1024):
            ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, table)
            ops.set_tuning(ops.TUNE_FWD_LDS_BYTES, ldsb)
            f = lambda s: lib.shr_sphere_raster_fwd_ex(sph.data_ptr(), n, J, S, S, dep.data_ptr(), None, 0, s)
            assert f(stream.cuda_stream) == 0
            stream.synchronize()
            if ref is None: ref = dep.clone()
            assert torch.equal(ref, dep)
            t = bench.mean_launch_us(f, stream, 20, 3, 5)
            print("depth-only  table %2d lds %3d KB: %6.1f us  frac %.3f" % (table, ldsb // 1024, t, n * (4 * S * S + 16 * J) / (t * 1e-6) / 8e12))
    ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, -1); ops.set_tuning(ops.TUNE_FWD_LDS_BYTES, 0)
    for table in (-1, 0):
        for cap in (0, 80 * 1024):
            ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, table)
            ops.set_tuning(ops.TUNE_FWD_OWNER_LDS_BYTES, cap)
            f = lambda s: lib.shr_sphere_raster_fwd_ex(sph.data_ptr(), n, J, S, S, dep.data_ptr(), own.data_ptr(), 1, s)
            assert f(stream.cuda_stream) == 0
            t = bench.mean_launch_us(f, stream, 20, 3, 5)
            print("with owner  table %2d cap %3d KB: %6.1f us" % (table, cap // 1024, t))
