#!/usr/bin/env python3
"""Golden vectors for the network-side modules and the loss assembly (SURVEY 8c G7,
G8), by RUNNING the imported reference on CPU.

    python tests/golden/make_goldens_network.py  ->  tests/golden/g7_network.npz

Reference entry points (file:line in /root/reference):
  network/hourglass.py:175                     create_hourglass_network
  network/util_modules.py:164-201, :204-240    RecoverXYZCoordinateFromHeatmap, HeatmapVariance
  mesh/render.py:145-206, :210-279             CollisionLoss, BoneLengthLoss, HeatmapRender, Hand3DHeatmapRender
  network/create_network_and_criterion.py:147-263  MultiTaskLoss (prior / temporal off; then, keys mtp_*, with
                                               --temporal and --prior ON over two consecutive batches: the
                                               TemporalSmoothnessLoss of network/util_modules.py:349-381 is stateful)
  network/utils_metric.py:7-17                 average_joint_error
Environment accommodation: mesh/bone_length.py calls .cuda() at import time
(:33); nn.Module.cuda is made a no-op while it is imported (there is no GPU here;
only its three data lists are used).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference, load_reference_mesh  # noqa: E402


def det_fill(module):
    """Deterministic, architecture-independent parameter values."""
    import torch
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            n = p.numel()
            seed = sum(ord(c) for c in name) % 97
            v = torch.sin(torch.arange(n, dtype=torch.float64) * 0.37 + seed) * (0.5 / max(1.0, (n / p.shape[0]) ** 0.5))
            if name.endswith('bn1.weight') or name.endswith('bn2.weight') or name.endswith('bn3.weight') or '.1.weight' in name:
                v = v * 0.2 + 1.0
            p.copy_(v.view_as(p).float())


def main():
    import_reference()
    import torch
    import torch.nn as nn
    orig_cuda = nn.Module.cuda
    nn.Module.cuda = lambda self, *a, **k: self
    try:
        from mesh.render import BoneLengthLoss, CollisionLoss, Hand3DHeatmapRender
        bl = BoneLengthLoss()
    finally:
        nn.Module.cuda = orig_cuda
    from network.hourglass import create_hourglass_network
    from network.util_modules import HeatmapVariance, RecoverXYZCoordinateFromHeatmap
    from network.create_network_and_criterion import MultiTaskLoss
    import network.utils_metric as um

    torch.manual_seed(0)
    torch.set_num_threads(8)
    mesh = load_reference_mesh()
    out = {}

    net = create_hourglass_network(82, 1).eval()
    det_fill(net)
    x = torch.sin(torch.arange(2 * 64 * 64, dtype=torch.float32) * 0.01).view(2, 64, 64)
    with torch.no_grad():
        y, lat = net(x)
    out["hg_out"] = y[0].numpy()
    out["hg_latent"] = lat[0].numpy()
    out["hg_keys"] = np.asarray(["%s:%s" % (k, tuple(v.shape)) for k, v in net.state_dict().items()])

    uv = torch.randn(5, 41, 16, 16)
    d = torch.randn(5, 41, 16, 16)
    out["uv_hms"], out["d_hms"] = uv.numpy(), d.numpy()
    out["xyz"] = RecoverXYZCoordinateFromHeatmap(16, 16, 0.01)(uv, d).numpy()
    out["hm_var"] = HeatmapVariance(16, 16)(uv).numpy()

    g3 = np.load(os.path.join(HERE, "g3_batch256.npz"))
    T = torch.from_numpy(g3["T"][:4])
    rand_f = torch.tensor([0.9, 1.1, 1.0, 0.95])
    hr = Hand3DHeatmapRender(mesh["bones"], 16)
    hms, dms, xyz = hr(T, rand_f)
    out["hm_T"], out["hm_rand_f"] = T.numpy(), rand_f.numpy()
    out["hm_uv"], out["hm_d"], out["hm_xyz"] = hms.numpy(), dms.numpy(), xyz.numpy()
    hms2, dms2, xyz2 = hr(T)
    out["hm_uv_nof"], out["hm_xyz_nof"] = hms2.numpy(), xyz2.numpy()

    joints = torch.from_numpy(g3["centres"][:6, :, :3]).reshape(2, 3, 41, 3).clone()
    joints = joints * torch.tensor([0.8, 0.9, 1.1]).view(1, 1, 1, 3)      # bend the skeleton: non-zero terms
    j = joints.clone().requires_grad_(True)
    col = CollisionLoss()(j)
    col.backward()
    out["geo_joints"] = joints.numpy()
    out["collision"], out["collision_grad"] = np.asarray(col.item()), j.grad.numpy()
    out["collision_pairs"] = np.stack([CollisionLoss().joint_1.numpy(), CollisionLoss().joint_2.numpy()])
    j = joints.clone().requires_grad_(True)
    b = bl(j)
    b.backward()
    out["bone_length"], out["bone_length_grad"] = np.asarray(b.item()), j.grad.numpy()
    out["bone_pairs"] = np.stack([bl.joint_1.numpy(), bl.joint_2.numpy()])
    out["bone_min_max"] = np.stack([bl.min_length.numpy()[0], bl.max_length.numpy()[0]])

    # ---- G7: MultiTaskLoss on a fixed fake result ------------------------------------
    class C:                       # the `constant` argument: only .mesh is read (:161)
        pass
    C.mesh = mesh
    nn.Module.cuda = lambda self, *a, **k: self
    try:
        crit = MultiTaskLoss(True, True, True, False, False, True, True, C, image_size=64)
    finally:
        nn.Module.cuda = orig_cuda
    g4 = np.load(os.path.join(HERE, "g4_mutual_projection.npz"))
    B, V = 4, 3
    result = {
        "real_xyz": [torch.from_numpy(g4["joints"])],
        "real_uv_hms": [torch.randn(B, V, 41, 16, 16) * 0.1],
        "synt_uv_hms": [torch.randn(6, 41, 16, 16) * 0.1],
        "synt_xyz": [torch.randn(6, 41, 3) * 20],
        "batch_synt_fea": [torch.randn(6, 256, 4, 4)],
        "batch_real_fea": [torch.randn(B * V, 256, 4, 4)],
    }
    synt_target = {"uv_hms": torch.rand(6, 41, 16, 16), "d_hms": torch.rand(6, 41, 16, 16),
                   "xyz_pts": torch.randn(6, 41, 4) * 20}
    real_target = {"real_dms": torch.from_numpy(g4["real_dms"]), "camera_poses": torch.from_numpy(g4["cam"]),
                   "inv_camera_poses": torch.from_numpy(g4["inv_cam"])}
    for k, v in result.items():
        out["mt_res_" + k] = v[0].numpy()
    for k, v in synt_target.items():
        out["mt_synt_" + k] = v.numpy()
    for is_mv in (True, False):
        real_target["is_mv"] = is_mv
        terms, proj = crit(result, synt_target=synt_target, real_target=real_target)
        for k, v in terms.items():
            out["mt_%s_%s" % ("mv" if is_mv else "diag", k)] = np.asarray(float(v), np.float64)
        print("G7", is_mv, {k: round(float(v), 5) for k, v in terms.items()})
    out["mt_weights"] = np.asarray(["%s=%r" % kv for kv in sorted(crit.weights.items())])

    # ---- G8: metric ------------------------------------------------------------------------
    gt = torch.randn(5, 3, 36, 3) * 30
    est = torch.randn(5, 3, 41, 3) * 30
    out["metric_gt"], out["metric_est"] = gt.numpy(), est.numpy()
    out["metric"] = np.asarray(um.average_joint_error(gt, est), np.float64)
    # ---- G7b: --temporal and --prior ON (create_network_and_criterion.py:157-158, :238-246) --------------
    # Two consecutive batches through ONE criterion: TemporalSmoothnessLoss keeps the previous batch's last
    # sample (util_modules.py:367-381; first call: B - 1 pairs, later calls: B pairs led by the remembered one).
    # PoseVae's reparameterisation draw (pose_vae.py:49) is recorded and replayed.  Placed after every other draw
    # of this script: the arrays above do not change when this block is added.
    # (the shipped pose_vae.pth holds CUDA storages and pose_vae.py:20 loads it without map_location: no GPU here)
    nn.Module.cuda = lambda self, *a, **k: self
    orig_load = torch.load
    torch.load = lambda f, *a, **k: orig_load(f, *a, **dict(k, map_location="cpu"))
    try:
        crit_tp = MultiTaskLoss(True, True, True, True, True, True, True, C, image_size=64)
    finally:
        nn.Module.cuda = orig_cuda
        torch.load = orig_load
    eps = torch.randn(B * V, 32)
    out["mtp_eps"] = eps.numpy()
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t_, *a, **k: eps.clone() if tuple(t_.shape) == tuple(eps.shape) else orig_randn_like(t_, *a, **k)
    try:
        base = result["real_xyz"][0]
        for call, xyz in enumerate((base, base * 1.02 + torch.tensor([1.5, -2.0, 0.7]))):
            xg = xyz.clone().requires_grad_(True)
            res = dict(result, real_xyz=[xg])
            real_target["is_mv"] = True
            terms, _ = crit_tp(res, synt_target=synt_target, real_target=real_target)
            (terms["temporal_smooth"] + terms["pose_prior"]).backward()
            out["mtp_call%d_real_xyz" % call] = xyz.numpy()
            out["mtp_call%d_grad_temporal_plus_prior" % call] = xg.grad.numpy()
            for k, v in terms.items():
                out["mtp_call%d_%s" % (call, k)] = np.asarray(float(v), np.float64)
            print("G7b call", call, {k: round(float(v), 5) for k, v in terms.items()})
    finally:
        torch.randn_like = orig_randn_like
    np.savez_compressed(os.path.join(HERE, "g7_network.npz"), **out)
    print("done", out["metric"])


if __name__ == "__main__":
    main()
