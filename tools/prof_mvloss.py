"""MutualProjectionLoss fwd+bwd at the per-GPU share of BASELINE config 5 (B=128 samples, V=3, S=256 -> 1152 crops)."""
import gc, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S = int(os.environ.get("B", 128)), int(os.environ.get("S", 256))
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda()
crit.cache_points = os.environ.get('CACHE', '0') == '1'    # default: fresh observations every call (what training pays)
# OVERLAP=0 (default HERE): both terms on one stream, so that a tracer's per-kernel durations are those of each kernel
# alone; the module's default (ops.MV_OVERLAP = True) runs the render-and-compare kernel beside the point search
from spherehand_amd import ops
ops.MV_OVERLAP = os.environ.get('OVERLAP', '0') == '1' 
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
joints = (ds.joints.cuda() + torch.randn_like(ds.joints.cuda())).requires_grad_(True)
def step():
    joints.grad = None
    loss, _ = crit(cam, inv, joints, real, True)
    loss.backward()
gc.collect(); gc.freeze()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("MutualProjectionLoss fwd+bwd B=%d S=%d (overlap %d, cache %d): %.1f us wall per step" % (B, S, ops.MV_OVERLAP, crit.cache_points, (time.perf_counter() - t0) / 20 * 1e6))
