"""CPU: network-side modules and the loss assembly vs the reference's golden vectors
(tests/golden/g7_network.npz, made by make_goldens_network.py).  Pure-torch modules
run on CPU here; the terms that need the HIP kernels are covered in test_engine_gpu."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, golden

torch = pytest.importorskip("torch")
sys.path.insert(0, GOLDEN)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_hourglass_matches_reference():
    from make_goldens_network import det_fill
    from spherehand_amd.hourglass import create_hourglass_network
    g = golden("g7_network.npz")
    net = create_hourglass_network(82, 1).eval()
    assert sum(p.numel() for p in net.parameters()) == 2308946
    assert ["%s:%s" % (k, tuple(v.shape)) for k, v in net.state_dict().items()] == list(g["hg_keys"])
    det_fill(net)
    x = torch.sin(torch.arange(2 * 64 * 64, dtype=torch.float32) * 0.01).view(2, 64, 64)
    with torch.no_grad():
        y, lat = net(x)
    assert np.abs(y[0].numpy() - g["hg_out"]).max() <= 1e-4 * np.abs(g["hg_out"]).max()
    assert np.abs(lat[0].numpy() - g["hg_latent"]).max() <= 1e-4 * np.abs(g["hg_latent"]).max()
    out2, _ = create_hourglass_network(82, 2)(x)
    assert len(out2) == 2 and out2[1].shape == (2, 82, 16, 16)


def test_soft_argmax_and_variance():
    from spherehand_amd.util_modules import HeatmapVariance, RecoverXYZCoordinateFromHeatmap
    g = golden("g7_network.npz")
    xyz = RecoverXYZCoordinateFromHeatmap(16, 16, 0.01)(t(g["uv_hms"]), t(g["d_hms"]))
    assert np.abs(xyz.numpy() - g["xyz"]).max() <= 1e-4 * np.abs(g["xyz"]).max()
    var = HeatmapVariance(16, 16)(t(g["uv_hms"]))
    assert np.abs(var.numpy() - g["hm_var"]).max() <= 1e-5


def test_heatmap_render():
    from spherehand_amd import hand_model
    from spherehand_amd.render import Hand3DHeatmapRender, HeatmapRender
    g = golden("g7_network.npz")
    hr = Hand3DHeatmapRender(hand_model.load_mesh()["bones"], 16)
    hms, dms, xyz = hr(t(g["hm_T"]), t(g["hm_rand_f"]))
    assert np.abs(hms.numpy() - g["hm_uv"]).max() <= 1e-5
    assert np.abs(xyz.numpy() - g["hm_xyz"]).max() <= 1e-3
    # the 0.05 mask is a threshold: allow the few cells within rounding of it to differ
    mism = (dms.numpy() != 0) != (g["hm_d"] != 0)
    assert mism.mean() < 1e-4
    assert np.abs(dms.numpy() - g["hm_d"])[~mism].max() <= 1e-3
    hms2, _, xyz2 = hr(t(g["hm_T"]))
    assert np.abs(hms2.numpy() - g["hm_uv_nof"]).max() <= 1e-5 and np.abs(xyz2.numpy() - g["hm_xyz_nof"]).max() <= 1e-3
    assert HeatmapRender(16)(torch.zeros(1, 2, 3))[0].shape == (1, 2, 16, 16)


def test_collision_and_bone_length():
    from spherehand_amd.render import BoneLengthLoss, CollisionLoss
    g = golden("g7_network.npz")
    col, bl = CollisionLoss(), BoneLengthLoss()
    assert np.array_equal(np.stack([col.joint_1.numpy(), col.joint_2.numpy()]), g["collision_pairs"])
    assert col.joint_1.numel() == 690                                                 # 330 + 360
    assert np.array_equal(np.stack([bl.joint_1.numpy(), bl.joint_2.numpy()]), g["bone_pairs"])
    assert np.allclose(np.stack([bl.min_length.numpy()[0], bl.max_length.numpy()[0]]), g["bone_min_max"], rtol=1e-6)
    for crit, key in ((col, "collision"), (bl, "bone_length")):
        j = t(g["geo_joints"]).requires_grad_(True)
        v = crit(j)
        v.backward()
        assert abs(v.item() - float(g[key])) <= 1e-5 * max(1.0, abs(float(g[key])))
        assert np.abs(j.grad.numpy() - g[key + "_grad"]).max() <= 1e-5 * max(1e-6, np.abs(g[key + "_grad"]).max())
    assert float(g["bone_length"]) > 0


def test_average_joint_error():
    from spherehand_amd.criterion import average_joint_error
    g = golden("g7_network.npz")
    v = average_joint_error(t(g["metric_gt"]), t(g["metric_est"]))
    assert abs(v.item() - float(g["metric"])) <= 1e-5 * float(g["metric"])


def test_multitask_loss_wiring_cpu_terms():
    """The terms that do not need the GPU kernels: weights, sums over stacks, the
    synthetic branch, heat-map mean, collision, bone length, domain (weight 0)."""
    from spherehand_amd import hand_model
    from spherehand_amd.criterion import MultiTaskLoss
    g = golden("g7_network.npz")
    crit = MultiTaskLoss(True, False, False, False, False, True, True, hand_model.load_mesh(), image_size=64)
    assert sorted("%s=%r" % kv for kv in crit.weights.items()) == list(g["mt_weights"])
    result = {k: [t(g["mt_res_" + k])] for k in
              ("real_xyz", "real_uv_hms", "synt_uv_hms", "synt_xyz", "batch_synt_fea", "batch_real_fea")}
    synt_target = {k: t(g["mt_synt_" + k]) for k in ("uv_hms", "d_hms", "xyz_pts")}
    real_target = {"real_dms": None, "camera_poses": None, "inv_camera_poses": None, "is_mv": True}
    terms, proj = crit(result, synt_target=synt_target, real_target=real_target)
    assert proj == []
    for k in ("synt_uv", "synt_d", "uv_hm_mean", "collision", "bone_length", "domain_loss"):
        ref = float(g["mt_mv_" + k])
        assert abs(float(terms[k]) - ref) <= 1e-5 * max(1.0, abs(ref)), k
    assert "mv_projection" not in terms and "mv_consistency" not in terms


def _tp_inputs(g, conv):
    result = {k: [conv(g["mt_res_" + k])] for k in
              ("real_uv_hms", "synt_uv_hms", "synt_xyz", "batch_synt_fea", "batch_real_fea")}
    synt_target = {k: conv(g["mt_synt_" + k]) for k in ("uv_hms", "d_hms", "xyz_pts")}
    return result, synt_target


def test_multitask_loss_temporal_and_prior_wiring_two_batches():
    """--temporal and --prior ON, two consecutive batches through one criterion (g7 keys mtp_*: the imported
    reference's MultiTaskLoss(True x 7), network/create_network_and_criterion.py:238-246; TemporalSmoothnessLoss keeps
    the previous batch's last sample, network/util_modules.py:367-381; the VAE's draw is the recorded one).  CPU: the
    terms that need no HIP kernel, and the gradient of temporal + prior w.r.t. the joints."""
    from spherehand_amd import hand_model
    from spherehand_amd.criterion import MultiTaskLoss
    from spherehand_amd.pose_vae import default_pose_vae
    g = golden("g7_network.npz")
    crit = MultiTaskLoss(True, False, False, True, default_pose_vae(), True, True, hand_model.load_mesh(), image_size=64)
    result, synt_target = _tp_inputs(g, t)
    eps = t(g["mtp_eps"])
    orig = torch.randn_like
    torch.randn_like = lambda a, *aa, **k: eps.clone() if tuple(a.shape) == tuple(eps.shape) else orig(a, *aa, **k)
    try:
        for call in (0, 1):
            xg = t(g["mtp_call%d_real_xyz" % call]).requires_grad_(True)
            terms, _ = crit(dict(result, real_xyz=[xg]), synt_target=synt_target,
                            real_target={"real_dms": None, "camera_poses": None, "inv_camera_poses": None, "is_mv": True})
            (terms["temporal_smooth"] + terms["pose_prior"]).backward()
            for k in ("synt_uv", "synt_d", "uv_hm_mean", "pose_prior", "temporal_smooth", "collision", "bone_length",
                      "domain_loss"):
                ref = float(g["mtp_call%d_%s" % (call, k)])
                assert abs(float(terms[k]) - ref) <= 1e-5 * max(1.0, abs(ref)), (call, k, float(terms[k]), ref)
            gref = g["mtp_call%d_grad_temporal_plus_prior" % call]
            assert np.abs(xg.grad.numpy() - gref).max() <= 1e-5 * np.abs(gref).max() + 1e-7, call
    finally:
        torch.randn_like = orig
    # the state the second call used: the first batch's last sample
    assert np.array_equal(crit.temporal_smooth_loss.previous_skel.numpy(), g["mtp_call1_real_xyz"][-1])


def test_network_wrapper_shapes_and_scale_division():
    from spherehand_amd.criterion import HeatmapEstimationNetwork
    net = HeatmapEstimationNetwork(16, 0.01, 41, 1).eval()
    with torch.no_grad():
        r = net(real_dms=torch.rand(2, 3, 64, 64), synt_dms=torch.rand(4, 64, 64))
        r_real = net(real_dms=torch.rand(2, 3, 64, 64))
        r_synt = net(synt_dms=torch.rand(4, 64, 64))
    assert r["real_xyz"][0].shape == (2, 3, 41, 3) and r["synt_xyz"][0].shape == (4, 41, 3)
    assert r["real_uv_hms"][0].shape == (2, 3, 41, 16, 16) and r["batch_real_fea"][0].shape == (6, 256, 4, 4)
    assert set(r_real) == {"real_uv_hms", "real_d_hms", "real_xyz"}
    assert set(r_synt) == {"synt_uv_hms", "synt_d_hms", "synt_xyz"}


def test_nyu_shard_roundtrip(tmp_path):
    from spherehand_amd.datasets import create_nyu_dataset, write_nyu_shard
    rs = np.random.RandomState(0)
    for k, n in enumerate((5, 3)):
        cams = np.tile(np.eye(4, dtype=np.float32), (n, 3, 1, 1))
        cams[:, :, :3, 3] = rs.randn(n, 3, 3)
        write_nyu_shard(str(tmp_path / ("mv_data_%d" % k)), rs.rand(n, 3, 64, 64), rs.randn(n, 3, 36, 3), cams)
    ds = create_nyu_dataset(str(tmp_path))
    assert len(ds) == 8
    dms, joints, cam, inv = ds[6]
    assert dms.shape == (3, 64, 64) and joints.shape == (3, 36, 3) and cam.shape == (3, 4, 4)
    assert np.allclose(cam @ inv, np.eye(4), atol=1e-5)
    with pytest.raises(FileNotFoundError):
        create_nyu_dataset(str(tmp_path / "missing"))


def test_resize_crop_batched_equals_per_image_loop():
    """The batched gather reproduces the reference's image-by-image F.interpolate + paste
    (network/util_modules.py:383-424), including its 'v > 1 is not pasted' quirk."""
    from spherehand_amd.util_modules import ResizeCropImage
    m = ResizeCropImage()
    g = torch.Generator().manual_seed(0)
    for (n, h, w) in ((9, 64, 64), (5, 48, 80), (4, 128, 128)):
        dms = torch.rand(n, h, w, generator=g)
        us = torch.rand(n, generator=g) * 0.5 + 0.7          # 0.7 .. 1.2: both branches of u
        vs = torch.rand(n, generator=g) * 0.45 + 0.65        # some > 1: left as ones
        us[0], vs[0] = 1.0, 1.0
        a, b = m(dms, us, vs), m.forward_loop(dms, us, vs)
        assert torch.equal(a, b), (n, h, w)
    scale = torch.rand(75, generator=g) * 0.2 + 0.75          # the training-time range (:99-101 of the wrapper)
    dms = torch.rand(75, 64, 64, generator=g)
    u, v = scale + torch.rand(75, generator=g) * 0.1 - 0.05, scale + torch.rand(75, generator=g) * 0.1 - 0.05
    assert torch.equal(m(dms, u, v), m.forward_loop(dms, u, v))


def test_batched_pose_sampler_has_the_sequential_samplers_distribution():
    """spherehand_amd.joint_angle.sample_poses_batched (what Engine._pose_iter draws per step) against n calls of
    sample_pose (the draw-for-draw port of dataset/joint_angle.py:7-236, pinned by g3): per-parameter means within
    5 standard errors, standard deviations within 6 %, the 5/25/50/75/95 % quantiles within 8 % of the parameter's
    spread (the flex angles are mixtures of five generators: a wrong table or weight moves the quantiles), and the
    support inside the sequential sampler's analytic bounds."""
    from spherehand_amd.joint_angle import sample_poses, sample_poses_batched
    n = 3000
    a = sample_poses(n, seed=11).double()
    b = sample_poses_batched(20000, generator=torch.Generator().manual_seed(5)).double()
    assert b.shape == (20000, 26) and b.dtype == torch.float64 and torch.isfinite(b).all()
    se = (a.var(0) / n + b.var(0) / 20000).sqrt()
    assert ((a.mean(0) - b.mean(0)).abs() <= 5 * se + 1e-9).all(), (a.mean(0) - b.mean(0)) / se
    assert ((a.std(0) / b.std(0) - 1).abs() <= 0.06).all(), a.std(0) / b.std(0)
    q = torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64)
    spread = a.quantile(0.99, 0) - a.quantile(0.01, 0)
    dq = (a.quantile(q, 0) - b.quantile(q, 0)).abs() / spread
    assert (dq <= 0.08).all(), dq.max()
    # palm ranges (joint_angle.py:22-29), thumb (:118-129)
    lo = torch.tensor([-3.14, -3.14, -3.14, -15, -15, -35], dtype=torch.float64)
    hi = torch.tensor([3.14, 0, 3.14, 15, 15, 15], dtype=torch.float64)
    assert (b[:, :6] >= lo - 1e-6).all() and (b[:, :6] <= hi + 1e-6).all()
    assert (b[:, 22].abs() <= 0.5).all() and (b[:, 23] >= -0.25 - 1e-6).all() and (b[:, 23] <= 0.7 + 1e-6).all()
    assert torch.allclose(b[:, 24], 0.25 * b[:, 23]) and (b[:, 25] >= -1.7 - 1e-6).all() and (b[:, 25] <= 0.3 + 1e-6).all()
    # every finger sees every generator family: straight/open flexes are <= 0.25, curled ones reach > 1 rad
    for base in (6, 10, 14, 18):
        assert (b[:, base + 1] < 0).float().mean() > 0.1 and (b[:, base + 1] > 0.9).float().mean() > 0.1
    # same generator, same draws
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    assert torch.equal(sample_poses_batched(48, g1), sample_poses_batched(48, g2))


def test_legacy_shaped_soft_argmax_grids_load():
    """Checkpoints this repository wrote before round 2 hold xyz_recover.u_grid / v_grid as [1,1,1,W] / [1,1,H,1]; the
    reference's (and today's) are full [1,1,H,W] grids: both load strictly, with the same values."""
    from spherehand_amd.util_modules import RecoverXYZCoordinateFromHeatmap
    m = RecoverXYZCoordinateFromHeatmap(16, 16, 0.01)
    sd = m.state_dict()
    legacy = {"u_grid": sd["u_grid"][:, :, :1, :].clone(), "v_grid": sd["v_grid"][:, :, :, :1].clone()}
    m2 = RecoverXYZCoordinateFromHeatmap(16, 16, 0.01)
    m2.u_grid.zero_(); m2.v_grid.zero_()
    m2.load_state_dict(legacy, strict=True)
    assert torch.equal(m2.u_grid, sd["u_grid"]) and torch.equal(m2.v_grid, sd["v_grid"])
    m2.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"u_grid": torch.zeros(1, 1, 3, 3), "v_grid": sd["v_grid"]}, strict=True)
