#!/usr/bin/env python3
"""Golden vectors for the render losses, by RUNNING the imported reference (CPU).

    python tests/golden/make_goldens_loss.py

  tests/golden/g5_data_to_model.npz   SURVEY 8c G5: DataToModelLoss alone
  tests/golden/g4_mutual_projection.npz  SURVEY 8c G4: MutualProjection /
                                      MutualProjectionLoss, B=4, V=3, S=64

Reference entry points (file:line in /root/reference):
  mesh/render.py:93-142                DataToModelLoss
  mesh/multiview_utility.py:9-30       MutualTransformation
  mesh/multiview_utility.py:32-77      MutualProjection
  mesh/multiview_utility.py:80-130     MutualProjectionLoss
  mesh/multiview_utility.py:133-167    MultiviewConsistencyLoss
torch.sqrt runs as the IEEE square root (see make_goldens_sphere.ieee_sqrt) for
the rendered depth maps stored bit-exactly; losses/gradients are compared with a
tolerance anyway.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference, load_reference_mesh  # noqa: E402
from make_goldens_sphere import ieee_sqrt  # noqa: E402


def rand_rigid(rs, n, max_deg=30.0):
    """n random rigid 4x4 (rotation <= max_deg about a random axis, translation +-10 mm) and inverses."""
    out, inv = [], []
    for _ in range(n):
        axis = rs.standard_normal(3)
        axis /= np.linalg.norm(axis)
        ang = np.deg2rad(rs.uniform(-max_deg, max_deg))
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rs.uniform(-10, 10, 3)
        out.append(T)
        inv.append(np.linalg.inv(T))
    return np.asarray(out, np.float32), np.asarray(inv, np.float32)


def main():
    import_reference()
    import torch
    from mesh.render import DataToModelLoss, HandBallPrimitiveRender
    from mesh.kinematicsTransformation import HandTransformationMat
    from mesh.multiview_utility import MutualProjection, MutualProjectionLoss, MultiviewConsistencyLoss
    from dataset.joint_angle import JointAngleDataset

    torch.set_num_threads(8)
    mesh = load_reference_mesh()
    fk = HandTransformationMat([b["offset_matrix"].astype(np.float32) for b in mesh["bones"]])
    torch.manual_seed(3)
    ds = JointAngleDataset()

    # ---------------- G5: DataToModelLoss --------------------------------------------
    out = {}
    for tag, (S, N) in {"a": (64, 6), "b": (128, 3)}.items():
        hbr = HandBallPrimitiveRender(mesh["bones"], S, S)
        p_obs = torch.stack([ds[i] for i in range(N)])
        p_est = p_obs + 0.08 * torch.randn_like(p_obs)
        with torch.no_grad(), ieee_sqrt():
            _, dms = hbr(fk(p_obs))                        # "observed" depth, background 100
            joints = hbr.lbs(fk(p_est))[:, :, :3].contiguous()
        joints = joints.clone().requires_grad_(True)
        crit = DataToModelLoss(S, S, mesh)
        loss = crit(dms, joints)
        loss.backward()
        out[tag + "_dms"] = dms.numpy()
        out[tag + "_joints"] = joints.detach().numpy()
        out[tag + "_radii"] = crit.radiuses.view(-1).numpy()
        out[tag + "_loss"] = np.asarray(loss.item(), np.float64)
        out[tag + "_grad_joints"] = joints.grad.numpy()
        print("G5", tag, "loss", loss.item(), "max|grad|", joints.grad.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "g5_data_to_model.npz"), **out)

    # ---------------- G4: MutualProjection(Loss) ------------------------------------
    B, V, S = 4, 3, 64
    rs = np.random.RandomState(11)
    cam, inv_cam = rand_rigid(rs, B * V)
    cam = torch.from_numpy(cam).view(B, V, 4, 4)
    inv_cam = torch.from_numpy(inv_cam).view(B, V, 4, 4)
    hbr = HandBallPrimitiveRender(mesh["bones"], S, S)
    p_true = torch.stack([ds[i] for i in range(B)])
    with torch.no_grad():
        c_true = hbr.lbs(fk(p_true))[:, :, :3]                        # canonical-frame centres [B,41,3]
        # view v sees the hand through inv_cam[v]: x_v = R_inv x + t_inv
        def to_view(c, M):
            return torch.einsum("bvij,bkj->bvki", M[:, :, :3, :3], c) + M[:, :, None, :3, 3]
        joints_true = to_view(c_true, inv_cam)                        # [B,V,41,3]
        p_est = p_true + 0.05 * torch.randn_like(p_true)
        joints_est = to_view(hbr.lbs(fk(p_est))[:, :, :3], inv_cam)
    mp = MutualProjection(S, mesh)
    with torch.no_grad(), ieee_sqrt():
        # observed depth of view j = spheres of the TRUE pose rendered in view j
        obs, _ = mp(cam, inv_cam, joints_true)
        real_dms = torch.stack([obs[:, j, j] for j in range(V)], dim=1).contiguous()   # [B,V,S,S]
        proj_ieee, pts = mp(cam, inv_cam, joints_est)
    out = {"cam": cam.numpy(), "inv_cam": inv_cam.numpy(), "joints": joints_est.numpy(),
           "real_dms": real_dms.numpy(), "proj_ieee": proj_ieee.numpy(), "projected_points": pts.numpy(),
           "radii": mp.radiuses.view(-1).numpy()}
    crit = MutualProjectionLoss(S, mesh)
    for is_mv in (True, False):
        j = joints_est.clone().requires_grad_(True)
        loss, proj = crit(cam, inv_cam, j, real_dms, is_mv)
        loss.backward()
        tag = "mv" if is_mv else "diag"
        out[tag + "_loss"] = np.asarray(loss.item(), np.float64)
        out[tag + "_grad_joints"] = j.grad.numpy()
        print("G4", tag, "loss", loss.item(), "max|grad|", j.grad.abs().max().item())
    mvc = MultiviewConsistencyLoss()
    j = joints_est.clone().requires_grad_(True)
    l = mvc(cam, j)
    l.backward()
    out["mvc_loss"] = np.asarray(l.item(), np.float64)
    out["mvc_grad_joints"] = j.grad.numpy()
    # views that disagree (per-view noise of a few mm): a non-trivial median
    noisy = joints_est + torch.from_numpy(np.random.RandomState(5).normal(0, 3.0, joints_est.shape).astype(np.float32))
    j = noisy.clone().requires_grad_(True)
    l = mvc(cam, j)
    l.backward()
    out["mvc2_joints"] = noisy.numpy()
    out["mvc2_loss"] = np.asarray(l.item(), np.float64)
    out["mvc2_grad_joints"] = j.grad.numpy()
    w = torch.from_numpy(np.random.RandomState(6).uniform(0.1, 1.0, (B, V, 41)).astype(np.float32))
    j = noisy.clone().requires_grad_(True)
    l = mvc(cam, j, w)
    l.backward()
    out["mvc2_hm_weight"] = w.numpy()
    out["mvc2w_loss"] = np.asarray(l.item(), np.float64)
    out["mvc2w_grad_joints"] = j.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g4_mutual_projection.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
