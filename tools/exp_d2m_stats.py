"""How far are the observed points from the model in the multiview benchmark data? (fraction needing the full search)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S = 16, 128
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda()
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
joints = ds.joints.cuda() + torch.randn_like(ds.joints.cuda())
with torch.no_grad():
    _, pts = crit.mutual_projection(cam, inv, joints)
N = B * 9
obs = real.unsqueeze(1).expand(B, 3, 3, S, S).reshape(N, S, S)
cen = pts.squeeze(-1).reshape(N, 41, 3)
rad = crit.data_to_model_criterion.radiuses.view(-1)
xs = (torch.arange(S, device="cuda") - S / 2) * 300.0 / S
X = xs.view(1, 1, S).expand(N, S, S); Y = xs.view(1, S, 1).expand(N, S, S)
P = torch.stack([X, Y, obs], -1)                       # [N,S,S,3]
d = (P.unsqueeze(3) - cen.view(N, 1, 1, 41, 3)).norm(dim=-1)
a = (d - rad.view(1, 1, 1, 41)).abs().min(-1).values
fg = obs <= 99
print("foreground fraction %.3f ; a: mean %.2f median %.2f ; fraction > 15 mm: %.3f ; > 8: %.3f ; > 30: %.3f"
      % (fg.float().mean().item(), a[fg].mean().item(), a[fg].median().item(), (a[fg] > 15).float().mean().item(),
         (a[fg] > 8).float().mean().item(), (a[fg] > 30).float().mean().item()))
