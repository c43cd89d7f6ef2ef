"""CPU: the frozen priors (PoseDenoiser, PoseVae), the reference's Eval metric semantics and
checkpoint interoperability, against vectors made by the imported reference
(tests/golden/make_goldens_priors.py -> g9_priors.npz)."""
import numpy as np
import pytest

from conftest import golden

torch = pytest.importorskip("torch")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_pose_denoiser_matches_reference():
    """network/pose_denoiser.py:56-73 with the shipped weights; bar 1e-5 relative to the 100-mm scale."""
    from spherehand_amd.pose_denoiser import PoseDenoiser, default_pose_denoiser
    g = golden("g9_priors.npz")
    dn = default_pose_denoiser()
    assert not dn.training and all(not p.requires_grad for p in dn.parameters())
    est = t(g["dn_in"])
    out = dn(est[:, 0])
    assert out.shape == (8, 41, 3)
    assert np.abs(out.numpy() - g["dn_out"]).max() <= 1e-5 * 100
    assert torch.equal(out[:, 11:], est[:, 0, 11:])                     # fingers pass through untouched
    flat = dn(est[:, 1].reshape(8, -1))                                  # [B,123] in -> [B,123] out
    assert flat.shape == (8, 123) and np.abs(flat.numpy() - g["dn_out_flat_view1"]).max() <= 1e-3
    assert list(PoseDenoiser().state_dict()) == list(dn.state_dict())


def test_eval_metric_is_view0_after_the_denoiser():
    """network/engine.py:200-206: gt view 0, pose_denoiser(est view 0), then average_joint_error."""
    from spherehand_amd.criterion import average_joint_error
    from spherehand_amd.pose_denoiser import default_pose_denoiser
    g = golden("g9_priors.npz")
    gt, est = t(g["metric_gt"]), t(g["dn_in"])
    dn = default_pose_denoiser()
    ev = average_joint_error(gt[:, 0].unsqueeze(1), dn(est[:, 0]).unsqueeze(1))
    assert abs(float(ev) - float(g["metric_eval"])) <= 1e-5 * float(g["metric_eval"])
    tr = average_joint_error(gt, est)
    assert abs(float(tr) - float(g["metric_train"])) <= 1e-5 * float(g["metric_train"])
    assert abs(float(g["metric_eval"]) - float(g["metric_train"])) > 1.0    # the two definitions differ


def test_pose_vae_matches_reference():
    from spherehand_amd.criterion import MultiTaskLoss
    from spherehand_amd.pose_vae import default_pose_vae
    g = golden("g9_priors.npz")
    vae = default_pose_vae()
    x = t(g["vae_x"])
    recon, mu, logvar, lik = vae(x)
    for a, k in ((recon, "vae_recon"), (mu, "vae_mu"), (logvar, "vae_logvar")):
        assert np.abs(a.numpy() - g[k]).max() <= 1e-5 * max(1.0, np.abs(g[k]).max()), k
    assert abs(float(lik) - float(g["vae_likelihood"])) <= 1e-5 * abs(float(g["vae_likelihood"]))
    xg = t(g["dn_in"] / np.float32(100.0)).requires_grad_(True)
    pl = vae.prior_loss(xg, eps=t(g["vae_eps"]))
    pl.backward()
    assert abs(float(pl) - float(g["vae_prior_loss"])) <= 1e-5 * abs(float(g["vae_prior_loss"]))
    assert np.abs(xg.grad.numpy() - g["vae_prior_grad"]).max() <= 1e-5 * np.abs(g["vae_prior_grad"]).max() + 1e-6
    # the loss assembly's 'pose_prior' term: /100 and the 1e-2 weight (create_network...:238-243)
    from spherehand_amd import hand_model
    crit = MultiTaskLoss(False, False, False, False, vae, False, False, hand_model.load_mesh(), image_size=64)
    eps = t(g["vae_eps"])
    orig = torch.randn_like
    torch.randn_like = lambda a, *aa, **k: eps.clone() if tuple(a.shape) == tuple(eps.shape) else orig(a, *aa, **k)
    try:
        terms, _ = crit({"real_xyz": [t(g["dn_in"])]}, real_target=None)
    finally:
        torch.randn_like = orig
    assert set(terms) == {"pose_prior"}
    assert abs(float(terms["pose_prior"]) - float(g["mt_pose_prior"])) <= 1e-5 * float(g["mt_pose_prior"])


def test_checkpoints_interoperate_with_the_reference_network():
    """network_state_dict keys, shapes and dtypes equal the reference HeatmapEstimationNetwork's
    (incl. the xyz_recover.u_grid / v_grid buffers, network/util_modules.py:174-181), so a state dict
    saved by either side loads strictly in the other."""
    from spherehand_amd.criterion import HeatmapEstimationNetwork
    g = golden("g9_priors.npz")
    net = HeatmapEstimationNetwork(16, 0.01, 41, 1)
    ours = ["%s:%s:%s" % (k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in net.state_dict().items()]
    assert ours == [str(s) for s in g["net_keys"]]
    assert np.array_equal(net.state_dict()["xyz_recover.u_grid"].numpy(), g["net_u_grid"])
    assert np.array_equal(net.state_dict()["xyz_recover.v_grid"].numpy(), g["net_v_grid"])
    # a state dict with the reference's shapes (as parsed from its key list) loads strictly
    ref_sd = {}
    for s in g["net_keys"]:
        k, shape, dtype = str(s).split(":")
        ref_sd[k] = torch.zeros(eval(shape), dtype=getattr(torch, dtype))
    net.load_state_dict(ref_sd, strict=True)
