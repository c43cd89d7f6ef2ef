"""GPU: the loss assembly with the HIP-backed terms vs the reference's G7 vector, the
synthetic branch, and the engine's train/eval steps on sphere-rendered multiview data."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def opts(model_dir, **kw):
    o = dict(synthesize=True, mv_projection=True, mv_consistency=True, temporal=False, prior=False, collision=True,
             bone_length=True, mode='Train', model_dir=str(model_dir), initial_model=None, restore_from_model=None,
             restore_from_epoch=-1, num_stacks=1, epoch=2, dataset_dir=None, depth_resample=0, lr=1e-3, tag='t',
             image_size=64, log_every=1000, real_batch=4, synt_batch=6, steps_per_epoch=3)
    o.update(kw)
    return SimpleNamespace(**o)


@pytest.mark.parametrize("is_mv", [True, False])
def test_multitask_loss_terms_vs_reference(is_mv):
    from spherehand_amd import hand_model
    from spherehand_amd.criterion import MultiTaskLoss
    g, g4 = golden("g7_network.npz"), golden("g4_mutual_projection.npz")
    crit = MultiTaskLoss(True, True, True, False, False, True, True, hand_model.load_mesh(), image_size=64).cuda()
    result = {k: [dev(g["mt_res_" + k])] for k in
              ("real_xyz", "real_uv_hms", "synt_uv_hms", "synt_xyz", "batch_synt_fea", "batch_real_fea")}
    synt_target = {k: dev(g["mt_synt_" + k]) for k in ("uv_hms", "d_hms", "xyz_pts")}
    real_target = {"real_dms": dev(g4["real_dms"]), "camera_poses": dev(g4["cam"]),
                   "inv_camera_poses": dev(g4["inv_cam"]), "is_mv": is_mv}
    terms, proj = crit(result, synt_target=synt_target, real_target=real_target)
    tag = "mv" if is_mv else "diag"
    for k, v in terms.items():
        ref = float(g["mt_%s_%s" % (tag, k)])
        assert abs(float(v) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(v), ref)
    assert set(terms) == {"synt_uv", "synt_d", "mv_projection", "mv_consistency", "uv_hm_mean", "collision",
                          "bone_length", "domain_loss"}
    assert len(proj) == 1 and proj[0].shape == (4, 3, 3, 64, 64)


def test_multitask_loss_temporal_and_prior_on_vs_reference():
    """Every switch ON (--temporal, --prior included) on the device, two consecutive batches: all ten terms against the
    imported reference (g7 keys mtp_*; network/create_network_and_criterion.py:183-263, util_modules.py:367-381)."""
    from spherehand_amd import hand_model
    from spherehand_amd.criterion import MultiTaskLoss
    from spherehand_amd.pose_vae import default_pose_vae
    g, g4 = golden("g7_network.npz"), golden("g4_mutual_projection.npz")
    crit = MultiTaskLoss(True, True, True, True, default_pose_vae(), True, True, hand_model.load_mesh(), image_size=64).cuda()
    result = {k: [dev(g["mt_res_" + k])] for k in
              ("real_uv_hms", "synt_uv_hms", "synt_xyz", "batch_synt_fea", "batch_real_fea")}
    synt_target = {k: dev(g["mt_synt_" + k]) for k in ("uv_hms", "d_hms", "xyz_pts")}
    real_target = {"real_dms": dev(g4["real_dms"]), "camera_poses": dev(g4["cam"]),
                   "inv_camera_poses": dev(g4["inv_cam"]), "is_mv": True}
    eps = dev(g["mtp_eps"])
    orig = torch.randn_like
    torch.randn_like = lambda a, *aa, **k: eps.clone() if tuple(a.shape) == tuple(eps.shape) else orig(a, *aa, **k)
    try:
        for call in (0, 1):
            xg = dev(g["mtp_call%d_real_xyz" % call]).requires_grad_(True)
            terms, proj = crit(dict(result, real_xyz=[xg]), synt_target=synt_target, real_target=real_target)
            assert set(terms) == {"synt_uv", "synt_d", "mv_projection", "mv_consistency", "uv_hm_mean", "pose_prior",
                                  "temporal_smooth", "collision", "bone_length", "domain_loss"}
            for k, v in terms.items():
                ref = float(g["mtp_call%d_%s" % (call, k)])
                assert abs(float(v) - ref) <= 1e-4 * max(1.0, abs(ref)), (call, k, float(v), ref)
            (terms["temporal_smooth"] + terms["pose_prior"]).backward()
            gref = g["mtp_call%d_grad_temporal_plus_prior" % call]
            assert np.abs(xg.grad.cpu().numpy() - gref).max() <= 1e-4 * np.abs(gref).max() + 1e-6, call
    finally:
        torch.randn_like = orig


def test_hand_synthesizer():
    from spherehand_amd import hand_model
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.util_modules import HandSynthesizer
    syn = HandSynthesizer(hand_model.load_mesh(), 64, 16, 1.0, 0.01).cuda()
    torch.manual_seed(0)
    dms, uv, d, xyz = syn(sample_poses(8, seed=3).cuda())
    assert dms.shape == (8, 64, 64) and uv.shape == (8, 41, 16, 16) and d.shape == uv.shape and xyz.shape == (8, 41, 4)
    assert not dms.requires_grad and torch.isfinite(dms).all()
    fg = (dms < 0.99).float().mean().item()
    assert 0.03 < fg < 0.5                        # a hand covers part of the crop; background = 1.0 (scaled 100)
    assert uv.max().item() <= 1.0 + 1e-6 and uv.max().item() > 0.5
    clean = HandSynthesizer(hand_model.load_mesh(), 64, 16, 1.0, 0.01, add_noise=False, out_heatmap=False).cuda()
    assert clean(sample_poses(2, seed=4).cuda()).shape == (2, 64, 64)


@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("fused", [True, False])
def test_hand_synthesizer_values_against_the_oracle_chain(S, fused):
    """HandSynthesizer.forward (network/util_modules.py:104-122) VALUES, not shapes: the module's outputs equal the chain
    assembled by hand from pieces that are pinned elsewhere -- the random scale and focal jitter the module drew (fused:
    its kernels' counter-based draws, read back from `last_draws`; fused = False: RandScale's three CPU draws and
    torch.rand, replayed under the same seed), the ORACLE's skinning + camera + triangle raster at 640 x 640 + clamp +
    bilinear resize (bit-exact: the fused kernel rasterizes the same arithmetic), x depth_scale, DepthNoise's formula on
    the module's own draws (1e-6), and the heat-map renderer's torch ops (6e-6 / 4e-4, its bars in tools/fuzz.py)."""
    from oracle import oracle
    from spherehand_amd import hand_model, synth_rng
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.util_modules import HandSynthesizer
    oracle.build()
    mesh = hand_model.load_mesh()
    B = 6
    pose = sample_poses(B, seed=5).cuda()
    syn = HandSynthesizer(mesh, S, 16, 1.0, 0.01, add_noise=False).cuda()
    syn.fused = fused
    torch.manual_seed(11)
    depth, uv, dh, xyz = syn(pose)
    if fused:
        d = syn.last_draws
        T = syn.hand_skeleton_transform(pose) * torch.cat([d[0:3].t(), torch.ones(B, 1, device="cuda")], 1).view(B, 1, 4, 1)
        rand_f = d[3].clone()
    else:       # the same draws, by hand: RandScale's three CPU draws, then the focal jitter on the device
        torch.manual_seed(11)
        T = syn.rand_scale(syn.hand_skeleton_transform(pose))
        rand_f = torch.rand(B, device="cuda") * 0.2 + 0.9
    assert ((T[:, :, 0, 0].abs() <= 0.95 + 1e-6).all() and (rand_f >= 0.9).all() and (rand_f < 1.1).all())
    start, bone, wv = hand_model.sparse_skin(mesh)
    verts = oracle.lbs_project(T.cpu().numpy(), start, bone, wv, True, syn.dm_render.camera, rand_f.cpu().numpy())
    faces = np.asarray(mesh["faces"], np.int64)[:, [1, 0, 2]]        # [F,3] into the mesh's own vertex list, winding as
    #                                                                  rasterized (right hand: mesh/render.py:298-300)
    fv = np.ascontiguousarray(verts[:, faces, 0:3], np.float32)                     # [B,F,3,3]
    raw = oracle.tri_raster_fwd(fv, 640, 640)
    ref = oracle.clamp_bilinear(raw, S, S, 100.0) * np.float32(0.01)
    assert np.array_equal(depth.cpu().numpy().view(np.uint32), ref.astype(np.float32).view(np.uint32))
    assert 0.03 < float((depth < 0.99).float().mean()) < 0.5
    # heat-maps and key-points: the module's torch formulation on the same transforms
    with torch.enable_grad():
        uv_t, dh_t, xyz_t = syn.hm_render(T, rand_f, 1.0, 0.01)
    for a, b, tol in ((uv, uv_t, 6e-6), (xyz, xyz_t, 4e-4)):
        assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())
    gate = (uv_t - 0.05).abs() > 1e-5                      # (depth maps are gated by uv_hm > 0.05: skip pixels on the gate)
    assert ((dh - dh_t) * gate).abs().max().item() <= 1e-6 * max(1.0, dh_t.abs().max().item())
    # with the noise: DepthNoise's formula on the draws the module consumes
    noisy = HandSynthesizer(mesh, S, 16, 1.0, 0.01, add_noise=True, out_heatmap=False).cuda()
    noisy.fused = fused
    torch.manual_seed(11)
    out = noisy(pose)
    if fused:       # same seed, same call counter: the same scale / jitter draws, hence the same clean image `depth`
        assert torch.equal(noisy.last_draws.view(torch.int32), syn.last_draws.view(torch.int32))
        keys = noisy.last_draws[4:6].cpu().numpy().view(np.uint32)
        expect = torch.from_numpy(synth_rng.depth_noise(depth.cpu().numpy(), keys, 0.5, 0.05)).cuda()
    else:           # one randn of three planes: x shift, y shift, z noise
        torch.manual_seed(11)
        noisy.rand_scale(noisy.hand_skeleton_transform(pose)); torch.rand(B, device="cuda")
        rn = torch.randn(3, B, S, S, device="cuda")
        u = torch.arange(S, device="cuda").view(1, 1, S); v = torch.arange(S, device="cuda").view(1, S, 1)
        sx = torch.clamp((rn[0] * 0.5 + 0.5).long() + u, 0, S - 1)
        sy = torch.clamp((rn[1] * 0.5 + 0.5).long() + v, 0, S - 1)
        g = torch.gather(depth.reshape(B, S * S), 1, (sy * S + sx).reshape(B, S * S)).view(B, S, S)
        expect = torch.where(g < 1.0, g + rn[2] * 0.05, g)
    assert (out - expect).abs().max().item() <= (2e-5 if fused else 1e-6)      # (fused: hardware log2 near 1, tests/test_synth_gpu.py)
    assert (out != depth).float().mean().item() > 0.02     # (the noise did something)


def test_engine_train_eval_checkpoint(tmp_path):
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.engine import Engine
    mesh = hand_model.load_mesh()
    train = SyntheticMultiviewDataset(mesh, 16, 64, seed=0)
    evald = SyntheticMultiviewDataset(mesh, 8, 64, seed=1)
    d, gt, cam, inv = train[0]
    assert d.shape == (3, 64, 64) and gt.shape == (3, 36, 3) and cam.shape == (3, 4, 4)
    assert 0.03 < (d < 100).float().mean().item() < 0.5
    torch.manual_seed(0)
    eng = Engine(opts(tmp_path), mesh=mesh, real_train_dataset=train, real_eval_dataset=evald)
    assert eng.with_real and eng.with_synt
    summary = eng.train()                          # 2 epochs x 3 steps of _epoch_with_both
    assert np.isfinite(list(summary["loss"].values())).all()
    assert {"mv_projection", "synt_uv", "collision", "bone_length"} <= set(summary["loss"])
    files = os.listdir(eng.model_path)
    assert {"model_-1.pth", "model_0.pth", "model_1.pth", "loss_weights.txt", "log.txt"} <= set(files)
    ev = eng.eval()
    assert np.isfinite(ev["metric"]["avg_joint_error"])
    # restore: weights + optimizer + scheduler position
    eng2 = Engine(opts(tmp_path, restore_from_model=eng.model_name, restore_from_epoch=1), mesh=mesh,
                  real_train_dataset=train, real_eval_dataset=evald)
    assert eng2.starting_epoch == 1
    for a, b in zip(eng.network.parameters(), eng2.network.parameters()):
        assert torch.equal(a, b)
    # weights-only load of a checkpoint path (--initial_model)
    eng3 = Engine(opts(tmp_path, initial_model=os.path.join(eng.model_path, "model_1.pth"), mode="Test"), mesh=mesh,
                  real_train_dataset=train, real_eval_dataset=evald)
    assert eng3.starting_epoch == 0
    assert torch.equal(next(eng3.network.parameters()), next(eng.network.parameters()))


def test_render_loss_fits_a_pose(tmp_path):
    """Training by fitting: descending the multiview render loss w.r.t. the joints
    pulls a perturbed skeleton back onto the observed depth maps."""
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, 4, 64, seed=2)
    crit = MutualProjectionLoss(64, mesh).cuda()
    real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
    truth = ds.joints.cuda()
    joints = (truth + 4.0 * torch.randn_like(truth)).requires_grad_(True)
    opt = torch.optim.Adam([joints], lr=0.5)
    first = None
    for it in range(60):
        opt.zero_grad()
        loss, _ = crit(cam, inv, joints, real, True)
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < 0.5 * first
    assert (joints - truth).norm(dim=-1).mean().item() < 4.0 * 1.7 * 0.8     # closer than the start (E|N(0,4^2 I)| ~ 6.4)


def test_synth_post_kernels_match_the_torch_modules():
    """heat-map painting + back-projection and the depth-noise kernel against the torch modules they replace."""
    import numpy as np
    from spherehand_amd import hand_model, ops
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.render import Hand3DHeatmapRender
    from spherehand_amd.util_modules import DepthNoise
    mesh = hand_model.load_mesh()
    fk = HandTransformationMat([b["offset_matrix"].astype(np.float32) for b in mesh["bones"]]).cuda()
    T = fk(sample_poses(12, seed=3).cuda()).detach()
    rf = torch.rand(12, device="cuda") * 0.2 + 0.9
    hm = Hand3DHeatmapRender(mesh["bones"], 16).cuda()
    for rand_f in (None, rf):
        with torch.no_grad():
            a = hm(T, rand_f, 1.0, 0.01)                      # kernels
        with torch.enable_grad():
            b = hm(T, rand_f, 1.0, 0.01)                      # torch ops
        for x, y, tol in zip(a, b, (2e-6, 1.2e-7, 2e-4)):      # (depth heat-map: values up to 1, one ulp there)
            assert x.shape == y.shape and (x - y).abs().max().item() <= tol
    # depth noise: identical to the torch formula fed the same normal draws
    dm = (torch.rand(5, 64, 64, device="cuda") * 1.4).contiguous()
    g = torch.Generator(device="cuda").manual_seed(7)
    out = ops.depth_noise(dm, 0.5, 0.05, generator=g)
    g.manual_seed(7)
    n3 = torch.randn((3, 5, 64, 64), device="cuda", generator=g)
    u = torch.arange(64, device="cuda").view(1, 1, 64); v = torch.arange(64, device="cuda").view(1, 64, 1)
    sx = torch.clamp((n3[0] * 0.5 + 0.5).long() + u, 0, 63); sy = torch.clamp((n3[1] * 0.5 + 0.5).long() + v, 0, 63)
    noisy = torch.gather(dm.reshape(5, -1), 1, (sy * 64 + sx).reshape(5, -1)).view(5, 64, 64)
    ref = torch.where(noisy < 1.0, noisy + n3[2] * 0.05, noisy)
    assert torch.equal(out, ref)
    with torch.no_grad():
        z = DepthNoise(64, 64).cuda()(dm)
    assert z.shape == dm.shape and ((z - dm).abs() > 0).float().mean().item() > 0.3


def test_pose_denoiser_on_gpu_matches_reference():
    """The shipped denoiser on the device vs the imported reference's outputs (g9): 1e-5 of the 100-mm scale."""
    from spherehand_amd.pose_denoiser import default_pose_denoiser
    g = golden("g9_priors.npz")
    dn = default_pose_denoiser().cuda()
    out = dn(dev(g["dn_in"])[:, 0])
    assert np.abs(out.cpu().numpy() - g["dn_out"]).max() <= 1e-5 * 100
    from spherehand_amd.criterion import average_joint_error
    ev = average_joint_error(dev(g["metric_gt"])[:, 0].unsqueeze(1), out.unsqueeze(1))
    assert abs(float(ev) - float(g["metric_eval"])) <= 1e-5 * float(g["metric_eval"])


def test_run_engine_eval_on_nyu_shards_reports_the_reference_metric(tmp_path, capsys):
    """BASELINE configs[2] end to end without the absent assets: NYU-format shards on disk ->
    `run_engine` (default mode = evaluation, every loss term on incl. the VAE prior) -> the printed
    `avg_joint_error` is the reference's definition (network/engine.py:158-159, :200-206, :262-263): batches of 8
    in order, view 0 only, PoseDenoiser first, mean of the per-batch means."""
    from spherehand_amd import hand_model, run_engine
    from spherehand_amd.criterion import HeatmapEstimationNetwork, average_joint_error
    from spherehand_amd.datasets import SyntheticMultiviewDataset, create_nyu_dataset, write_nyu_shard
    from spherehand_amd.pose_denoiser import default_pose_denoiser
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, 20, 64, seed=5)
    for split, sl in (("train", slice(0, 8)), ("test", slice(0, 20))):
        os.makedirs(tmp_path / "nyu" / split)
        # two shards per split (the reader concatenates mv_data_0, mv_data_1, ...)
        n = (sl.stop - sl.start) // 2
        for k, s in enumerate((slice(sl.start, sl.start + n), slice(sl.start + n, sl.stop))):
            write_nyu_shard(str(tmp_path / "nyu" / split / ("mv_data_%d" % k)), ds.dms[s].numpy(), ds.gt[s].numpy(),
                            ds.cam[s].numpy())
    torch.manual_seed(3)
    net = HeatmapEstimationNetwork(16, 0.01, 41, 1)
    ckpt = str(tmp_path / "self-supervised.pth")
    torch.save({"epoch": 0, "network_state_dict": net.state_dict()}, ckpt)
    summary = run_engine.main(["--dataset_dir", str(tmp_path / "nyu"), "--model_dir", str(tmp_path / "out"),
                               "--initial_model", ckpt])
    assert "avg_joint_error" in capsys.readouterr().out
    # the same number by hand
    net = net.cuda().to(memory_format=torch.channels_last).eval()
    dn = default_pose_denoiser().cuda()
    test = create_nyu_dataset(str(tmp_path / "nyu" / "test"))
    assert len(test) == 20
    per_batch = []
    with torch.no_grad():
        for s in range(0, 20, 8):
            items = [test[i] for i in range(s, min(s + 8, 20))]
            dms = torch.from_numpy(np.stack([it[0] for it in items])).cuda()
            gt = torch.from_numpy(np.stack([it[1] for it in items])).cuda()
            xyz = net(real_dms=dms * 0.01)["real_xyz"][-1]
            per_batch.append(float(average_joint_error(gt[:, 0].unsqueeze(1), dn(xyz[:, 0]).unsqueeze(1))))
    expect = float(np.mean(per_batch))
    got = summary["metric"]["avg_joint_error"]
    assert abs(got - expect) <= 1e-5 * expect, (got, expect)
    assert "pose_prior" in summary["loss"] and "mv_projection" in summary["loss"]
