"""A/B on one box: MutualProjectionLoss step with the render-and-compare kernel on a side stream beside the point search."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return loss
def run(tag):
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): l = step()
    torch.cuda.synchronize(); print("%s: %.1f us  loss %.9g" % (tag, (time.perf_counter() - t0) / 200 * 1e6, float(l.detach())))
gc.collect(); gc.freeze()   # (a full collection of torch's objects is a 40-ms host stall: one run in six read 470 us)
for r in range(3):
    for mode, tag in ((False, "one stream"), (True, "render-and-compare on a side stream beside the point search")):
        ops.MV_OVERLAP = mode; run(tag)
