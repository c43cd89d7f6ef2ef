#!/usr/bin/env python3
"""How many spheres must a data->model search evaluate per point?  Emulates, in numpy, groups of G tile-sorted (or
pixel-order) foreground points of config 5's crops and counts stage 1 (spheres whose bound is <= 0) and stage 2
(bound below the reach) for: strip bound, x-y box bound, x-y-z box bound."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss

S = int(os.environ.get("S", "256")); B = 6
mesh = hand_model.load_mesh()
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0, device="cuda")
crit = MutualProjectionLoss(S, mesh).cuda()
torch.manual_seed(0)
joints = ds.joints.cuda() + torch.randn(ds.joints.shape, device="cuda") * float(os.environ.get("NOISE", "1.0"))
with torch.no_grad():
    _, pts = crit.mutual_projection(ds.cam.cuda(), ds.inv_cam.cuda(), joints)
cen = pts.squeeze(-1).reshape(B * 9, 41, 3).cpu().numpy()
rad = crit.data_to_model_criterion.radiuses.view(-1).cpu().numpy()
obs = ds.dms.view(B * 3, S, S).numpy()
index = (np.arange(B)[:, None, None] * 3 + np.arange(3)[None, None, :]).repeat(3, 1).reshape(-1)

def order(v, u, mode):
    if mode == "pixel":
        return np.lexsort((u, v))
    th, tw = mode
    ty, tx = v // th, u // tw
    txs = np.where(ty % 2 == 1, (S // tw) - 1 - tx, tx)
    return np.lexsort((u, v, txs, ty))

for mode in ("pixel", (16, 16), (8, 32), (32, 32), (16, 32)):
    for G in (256, 128, 64):
        s1 = {"strip": [], "xy": [], "xyz": []}; s2 = {"strip": [], "xy": [], "xyz": []}
        for n in range(0, B * 9, 2):
            im = obs[index[n]]
            v, u = np.nonzero(im <= 99)
            z = im[v, u]
            o = order(v, u, mode)
            v, u, z = v[o], u[o], z[o]
            x = (u - S / 2) * 300.0 / S; y = (v - S / 2) * 300.0 / S
            c, r = cen[n], rad
            for g0 in range(0, len(v), G):
                sl = slice(g0, g0 + G)
                P = np.stack([x[sl], y[sl], z[sl]], 1)
                d = np.abs(np.linalg.norm(P[:, None, :] - c[None], axis=2) - r[None])      # [pts, J]
                lo, hi = P.min(0), P.max(0)
                dd = np.maximum(np.maximum(lo[None] - c, c - hi[None]), 0)                 # [J,3]
                lbs = {"strip": dd[:, 1] - r, "xy": np.hypot(dd[:, 0], dd[:, 1]) - r, "xyz": np.linalg.norm(dd, axis=1) - r}
                for k, lb in lbs.items():
                    m1 = lb <= 1e-3
                    if not m1.any():
                        m1 = lb <= lb.min()
                    reach = min(d[:, m1].min(1).max(), 50.0)
                    m2 = (~m1) & (lb <= reach)
                    s1[k].append(m1.sum()); s2[k].append(m2.sum())
        print("order %-10s G=%3d  " % (str(mode), G) + "   ".join("%s: %.1f + %.1f" % (k, np.mean(s1[k]), np.mean(s2[k])) for k in s1), flush=True)
