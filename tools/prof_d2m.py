import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S = 128, int(os.environ.get('S', '128'))
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda()
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
with torch.no_grad():
    _, pts = crit.mutual_projection(cam, inv, ds.joints.cuda() + 1.0)
N = B * 9
obs = real.view(B * 3, S, S).contiguous()
idx = (torch.arange(B, device="cuda", dtype=torch.int32).view(B, 1, 1) * 3 + torch.arange(3, device="cuda", dtype=torch.int32).view(1, 1, 3)).expand(B, 3, 3).reshape(-1).contiguous()
cen = pts.squeeze(-1).reshape(N, 41, 3).contiguous()
rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
if 'TILED' in os.environ:
    ops.set_tuning(ops.TUNE_D2M_TILED, int(os.environ['TILED']))
if 'BAND' in os.environ:
    ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, int(os.environ['BAND']))
for _ in range(int(os.environ.get('REPS', '5'))): ops.data_to_model(obs, cen, rad, want_grad=True, depth_index=idx)
torch.cuda.synchronize()
