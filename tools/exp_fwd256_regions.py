"""Depth-only / owner forward at config 5's own size (1152 projected crops @256x256: WIDE boxes) with the row regions forced
smaller (SHR_TUNE_FWD_LDS_BYTES / _OWNER_LDS_BYTES size the regions; the box variant then takes half of a CU's LDS): fewer rows of
a wide box fall to the tile code, more workgroups pay a prologue."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
from bench import mean_launch_us, roof
lib = _lib.lib()
dev = torch.device("cuda")
stream = torch.cuda.Stream()
mesh = hand_model.load_mesh()
B5, S5, J = 128, 256, 41
ds = SyntheticMultiviewDataset(mesh, B5, S5, seed=0, device=dev)
crit = MutualProjectionLoss(S5, mesh).to(dev)
n = B5 * 9
with torch.no_grad():
    _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + 1.0)
rad = crit.data_to_model_criterion.radiuses.view(-1)
sph = torch.cat([pts.squeeze(-1).reshape(n, J, 3), rad.view(1, J, 1).expand(n, J, 1)], -1).contiguous()
depth = torch.empty(n, S5, S5, device=dev); owner = torch.empty(n, S5, S5, device=dev, dtype=torch.uint8)
p = [t.data_ptr() for t in (sph, depth, owner)]
ref = None
with torch.cuda.stream(stream):
    for kb in (0, 120, 100, 80, 60, 48):
        ops.set_tuning(ops.TUNE_FWD_LDS_BYTES, kb * 1024); ops.set_tuning(ops.TUNE_FWD_OWNER_LDS_BYTES, kb * 1024)
        f0 = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S5, S5, p[1], None, 0, s), "fwd"), stream, 25, 4, 3, warm_ms=30.0)
        d0 = depth.clone()
        f1 = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S5, S5, p[1], p[2], 1, s), "fwd"), stream, 25, 4, 3, warm_ms=30.0)
        if ref is None: ref = d0
        print("region budget %3d KB: depth-only %.1f us (%.3f of the peak), with owner bytes %.1f us; same depth bits: %s" %
              (kb, f0, roof(n * (4 * S5 * S5 + 16 * J), f0)["frac"], f1, torch.equal(d0, ref) and torch.equal(depth, ref)), flush=True)
    ops.set_tuning(ops.TUNE_FWD_LDS_BYTES, 0); ops.set_tuning(ops.TUNE_FWD_OWNER_LDS_BYTES, 0)
