#!/usr/bin/env python3
"""Frozen priors: weight re-export + golden vectors, by RUNNING the imported reference on CPU.

    python tests/golden/make_goldens_priors.py
        -> spherehand_amd/data/pose_denoiser.npz, spherehand_amd/data/pose_vae.npz  (weights: data only)
        -> tests/golden/g9_priors.npz                                             (inputs + reference outputs)

Reference entry points (file:line in /root/reference):
  network/pose_denoiser.py:21-81                  PoseDenoiser (weights mesh/model/pose_denoiser.pth)
  network/pose_vae.py:11-99                       PoseVae      (weights mesh/model/pose_vae.pth)
  network/engine.py:200-206                       the Eval metric: view 0 only, denoiser first
  network/utils_metric.py:7-17                    average_joint_error
  network/create_network_and_criterion.py:27-38   HeatmapEstimationNetwork state-dict keys / shapes
  network/create_network_and_criterion.py:238-243 MultiTaskLoss 'pose_prior' term
The VAE's reparameterisation draws torch.randn_like (pose_vae.py:49); for the vectors that involve it
the draw is replaced by a recorded `eps` (torch.randn_like patched for the duration of the call).
Environment accommodation: the shipped .pth files hold CUDA storages and the reference loads them
without map_location (pose_denoiser.py:42, pose_vae.py:20); there is no GPU here, so torch.load is
given map_location='cpu' while the reference's constructors run.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from _refimport import import_reference, load_reference_mesh  # noqa: E402


def export_state_dict(sd, path):
    arrays = {}
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        arrays[k] = a.astype(np.int64) if a.dtype.kind in "iu" else a.astype(np.float32)
    np.savez_compressed(path, **arrays)


def main():
    import_reference()
    import torch
    import torch.nn as nn
    from network.pose_denoiser import PoseDenoiser
    from network.pose_vae import PoseVae
    import network.utils_metric as um

    torch.manual_seed(0)
    real_load = torch.load
    torch.load = lambda f, *a, **k: real_load(f, *a, **dict(k, map_location="cpu"))
    out = {}
    data_dir = os.path.join(ROOT, "spherehand_amd", "data")

    dn = PoseDenoiser(model_path="mesh/model/pose_denoiser.pth").eval()
    export_state_dict(dn.state_dict(), os.path.join(data_dir, "pose_denoiser.npz"))
    vae = PoseVae(41 * 3, 32, "mesh/model/pose_vae.pth").eval()
    export_state_dict(vae.state_dict(), os.path.join(data_dir, "pose_vae.npz"))

    # ---- denoiser on posed sphere centres (g3's JointAngleDataset poses through FK) + noise ------------
    g3 = np.load(os.path.join(HERE, "g3_batch256.npz"))
    centres = torch.from_numpy(g3["centres"][:24, :, :3]).clone()              # [24,41,3] mm
    centres = centres - centres[:, 5:6]                                         # roughly palm-centred, as the crops are
    est = (centres + torch.randn_like(centres) * 4.0).reshape(8, 3, 41, 3)      # "network output" [B,V,41,3]
    with torch.no_grad():
        den = dn(est[:, 0])
        den_flat = dn(est[:, 1].reshape(8, -1))
    out["dn_in"], out["dn_out"], out["dn_out_flat_view1"] = est.numpy(), den.numpy(), den_flat.numpy()

    # ---- the Eval metric exactly as engine.py:200-206 spells it -----------------------------------------
    gt = torch.randn(8, 3, 36, 3) * 30
    gt0 = gt[:, 0].unsqueeze(dim=1)
    with torch.no_grad():
        est0 = dn(est[:, 0]).unsqueeze(dim=1)
    out["metric_gt"] = gt.numpy()
    out["metric_eval"] = np.asarray(um.average_joint_error(gt_joints=gt0, est_joints=est0), np.float64)
    out["metric_train"] = np.asarray(um.average_joint_error(gt_joints=gt, est_joints=est), np.float64)

    # ---- VAE: deterministic pass and the prior term with a recorded draw --------------------------------
    x = (est / 100.0).reshape(-1, 123)
    with torch.no_grad():
        recon, mu, logvar, lik = vae(x)
    out["vae_x"], out["vae_recon"], out["vae_mu"], out["vae_logvar"] = x.numpy(), recon.numpy(), mu.numpy(), logvar.numpy()
    out["vae_likelihood"] = np.asarray(float(lik), np.float64)
    eps = torch.randn(24, 32)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.clone() if tuple(t.shape) == tuple(eps.shape) else orig(t, *a, **k)
    try:
        xg = (est / 100.0).clone().requires_grad_(True)
        pl = vae.prior_loss(xg)
        pl.backward()
    finally:
        torch.randn_like = orig
    out["vae_eps"] = eps.numpy()
    out["vae_prior_loss"] = np.asarray(float(pl), np.float64)
    out["vae_prior_grad"] = xg.grad.numpy()

    # ---- MultiTaskLoss with the prior switched on (pins the /100 and the 1e-2 weight) --------------------
    orig_cuda = nn.Module.cuda
    nn.Module.cuda = lambda self, *a, **k: self
    try:
        from network.create_network_and_criterion import HeatmapEstimationNetwork, MultiTaskLoss

        class C:
            pass
        C.mesh = load_reference_mesh()
        crit = MultiTaskLoss(False, False, False, False, True, False, False, C, image_size=64)
    finally:
        nn.Module.cuda = orig_cuda
    torch.randn_like = lambda t, *a, **k: eps.clone() if tuple(t.shape) == tuple(eps.shape) else orig(t, *a, **k)
    try:
        terms, _ = crit({"real_xyz": [est]}, real_target=None)
    finally:
        torch.randn_like = orig
    out["mt_pose_prior"] = np.asarray(float(terms["pose_prior"]), np.float64)
    print({k: float(v) for k, v in terms.items()})

    # ---- checkpoint interop: the reference network's state-dict keys and shapes --------------------------
    net = HeatmapEstimationNetwork(16, 0.01, 41, 1)
    sd = net.state_dict()
    out["net_keys"] = np.asarray(["%s:%s:%s" % (k, tuple(v.shape), str(v.dtype).replace("torch.", ""))
                                  for k, v in sd.items()])
    out["net_u_grid"], out["net_v_grid"] = sd["xyz_recover.u_grid"].numpy(), sd["xyz_recover.v_grid"].numpy()
    np.savez_compressed(os.path.join(HERE, "g9_priors.npz"), **out)
    print("done: eval metric %.6f (train-style %.6f), prior loss %.6f" %
          (out["metric_eval"], out["metric_train"], out["vae_prior_loss"]))


if __name__ == "__main__":
    main()
