import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S = 16, 256
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda(); crit.cache_points = False
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
joints = (ds.joints.cuda() + torch.randn_like(ds.joints.cuda())).requires_grad_(True)
def step():
    joints.grad = None
    loss, _ = crit(cam, inv, joints, real, True)
    loss.backward()
for _ in range(3): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or str(e.device_type).endswith("CUDA"):
        print("KERNEL", e.name[:90])
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=25, max_name_column_width=50))
