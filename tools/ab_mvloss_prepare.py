"""A/B on one box: loss step with the fused prepare call vs projection + compaction (memset) separately."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from spherehand_amd import hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S = 128, 256
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda(); crit.cache_points = False
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
joints = (ds.joints.cuda() + torch.randn_like(ds.joints.cuda())).requires_grad_(True)
def step():
    joints.grad = None
    loss, _ = crit(cam, inv, joints, real, True)
    loss.backward()
orig = crit._point_lists
def separate(observed):
    return ops.d2m_compact(observed), False
def run(tag):
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); print("%s: %.1f us" % (tag, (time.perf_counter() - t0) / 200 * 1e6))
for r in range(3):
    crit._point_lists = orig; run("fused prepare (2 launches)")
    crit._point_lists = separate; run("memset + compaction + projection (3 launches)")
