#!/bin/bash
# LDS row padding (kRowPad, -DSHR_ROW_PAD=<n>) of the zbuf kernels: forward / backward / fused timings per variant.
# Variants are built beforehand into tools/libshr_pad<n>.so (see git log); run on the GPU box from the repo root.
cp spherehand_amd/libspherehand_hip.so /tmp/orig.so
for pad in 8 4 12 16 20 24; do
  if [ $pad = 8 ]; then cp /tmp/orig.so spherehand_amd/libspherehand_hip.so; else cp tools/libshr_pad$pad.so spherehand_amd/libspherehand_hip.so; fi
  echo "== pad $pad"
  timeout 200 python tools/exp_owner_rows.py 2>&1 | grep "^N" | sed 's/full owner: //;s/| owner bytes.*//'
  timeout 200 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from spherehand_amd import _lib, hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
lib = _lib.lib(); dev = torch.device("cuda:0"); mesh = hand_model.load_mesh()
B5, S5, J = 128, 256, 41
ds = SyntheticMultiviewDataset(mesh, B5, S5, seed=0, device=dev)
crit = MutualProjectionLoss(S5, mesh).to(dev)
with torch.no_grad():
    _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + 1.0)
n5 = B5 * 9
obs = ds.dms.to(dev).view(B5 * 3, S5, S5).contiguous()
index = (torch.arange(B5, device=dev, dtype=torch.int32).view(B5, 1, 1) * 3 + torch.arange(3, device=dev, dtype=torch.int32).view(1, 1, 3)).expand(B5, 3, 3).reshape(-1).contiguous()
rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
sph = torch.cat([pts.squeeze(-1).reshape(n5, J, 3), rad.view(1, J, 1).expand(n5, J, 1)], -1).contiguous()
Rm = lib.shr_sphere_raster_mse_regions(S5, S5)
dep = torch.empty(n5, S5, S5, device=dev); sse = torch.empty(n5 * Rm, device=dev); gsp = torch.empty(n5 * Rm, J, 4, device=dev)
m = [t.data_ptr() for t in (sph, obs, index, dep, sse, gsp)]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    t = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_mse(m[0], n5, J, S5, S5, m[1], m[2], m[3], m[4], m[5], s), stream, 20, 3, 3)
print("fused render-and-compare 1152 crops @256: %.1f us (regions %d)" % (t, Rm))
PY
done
cp /tmp/orig.so spherehand_amd/libspherehand_hip.so
