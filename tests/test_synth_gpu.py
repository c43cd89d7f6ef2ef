"""HandSynthesizer as three launches (network/util_modules.py:104-122): shr_synth_pose_fwd, shr_mesh_render_post_fwd,
shr_heatmap_render_fwd -- the kernels' counter-based random numbers against their numpy restatement
(spherehand_amd/synth_rng.py), the one-graph path against the module-by-module chain on the same draws, the noise's
DISTRIBUTION against the reference's torch formulas (network/util_modules.py:60-84, :110;
mesh/pointTransformation.py:143-145), and capture + replay in a hipGraph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _syn(S, hm=16, one_launch=True, **kw):
    from spherehand_amd import hand_model
    from spherehand_amd.util_modules import HandSynthesizer
    syn = HandSynthesizer(hand_model.load_mesh(), S, hm, 1.0, 0.01, **kw).cuda()
    syn.one_launch = one_launch       # True: ONE kernel where its sizes allow; False: always the three launches
    return syn


def _poses(B, seed):
    from spherehand_amd.joint_angle import sample_poses
    return sample_poses(B, seed=seed).cuda()


def _state(syn):
    seed, ctr = (int(v) for v in syn.rng_state.cpu().tolist()[:2])
    return seed & (2 ** 64 - 1), ctr


@pytest.mark.parametrize("one_launch", [True, False])
def test_draws_equal_the_numpy_restatement_and_the_counter_advances(one_launch):
    from spherehand_amd import synth_rng
    syn = _syn(64, one_launch=one_launch)
    pose = _poses(37, 1)
    torch.manual_seed(123456789012345)
    for call in range(3):
        syn(pose)
        seed, ctr = _state(syn)
        assert seed == 123456789012345 and ctr == call + 1          # the render launch advanced it
        f, keys = synth_rng.sample_draws(seed, call, 37, 0.1)
        d = syn.last_draws.cpu().numpy()
        assert np.array_equal(d[0:4].view(np.uint32), f.view(np.uint32))
        assert np.array_equal(d[4:6].view(np.uint32), keys)
        assert (d[0:3] >= 0.85 - 1e-6).all() and (d[0:3] < 0.95 + 1e-6).all() and (d[3] >= 0.9).all() and (d[3] < 1.1 + 1e-6).all()
    # a changed torch seed restarts the stream; the same seed continues it; reseed() restarts by hand
    torch.manual_seed(7)
    syn(pose)
    assert _state(syn) == (7, 1)
    torch.manual_seed(7)
    syn(pose)
    assert _state(syn) == (7, 2)
    syn.reseed()
    assert _state(syn) == (7, 0)
    # an explicit seed (and call counter) stays whatever torch is seeded with, also when set before the first call
    fresh = _syn(64, one_launch=one_launch)
    fresh.reseed(4242, counter=5)
    torch.manual_seed(1)
    fresh(pose)
    assert _state(fresh) == (4242, 6)
    torch.manual_seed(2)
    fresh(pose)
    assert _state(fresh) == (4242, 7)
    torch.manual_seed(7)
    # seed_offset (Engine: the rank): ranks under one torch seed draw different streams
    syn.seed_offset = 3
    syn(pose)
    assert _state(syn) == (10, 1)


@pytest.mark.parametrize("one_launch", [True, False])
@pytest.mark.parametrize("S,hm", [(64, 16), (128, 32), (32, 8), (256, 16), (128, 12)])
def test_fused_paths_equal_the_module_chain_on_the_same_draws(S, hm, one_launch):
    """Noise off: depth, heat-maps and key-points bit for bit the modules' (FK -> x diag(s) -> DepthRender(rand_f) x
    depth_scale; Hand3DHeatmapRender) on the draws the kernels made.  S = 256 takes the two-launch fallback of the
    render entry (no lattice kernel at that size)."""
    B = 9
    syn = _syn(S, hm, one_launch, add_noise=False)
    pose = _poses(B, 2)
    torch.manual_seed(3)
    depth, uv, dh, xyz = syn(pose)
    d = syn.last_draws
    T = syn.hand_skeleton_transform(pose) * torch.cat([d[0:3].t(), torch.ones(B, 1, device="cuda")], 1).view(B, 1, 4, 1)
    with torch.no_grad():
        depth_m = syn.dm_render(T, d[3].clone()) * syn.depth_scale
        uv_m, dh_m, xyz_m = syn.hm_render(T, d[3].clone(), syn.uv_hm_scale, syn.depth_scale)
    assert torch.equal(depth, depth_m)
    assert torch.equal(uv, uv_m) and torch.equal(dh, dh_m) and torch.equal(xyz, xyz_m)
    assert 0.03 < float((depth < 0.99).float().mean()) < 0.5


def test_one_launch_equals_three_launches():
    """ONE kernel (FK + draws + skinning + raster + scale + noise + heat-maps per workgroup) against the three launches:
    every output and the draws bit for bit, over several calls, and with more crops than the device holds at once (the
    call counter is advanced by the launch's LAST workgroup: a late workgroup must still read this call's counter)."""
    for B, S, hm in ((7, 128, 16), (700, 128, 32), (33, 64, 16)):
        a, b = _syn(S, hm, True), _syn(S, hm, False)
        pose = _poses(B, 9)
        torch.manual_seed(1234)
        for call in range(3):
            oa, ob = a(pose), b(pose)
            assert torch.equal(a.last_draws.view(torch.int32), b.last_draws.view(torch.int32))   # (bits: the keys are not numbers)
            for x, y in zip(oa, ob):
                assert torch.equal(x, y)
            assert _state(a) == _state(b) == (1234, call + 1) and int(a.rng_state[2]) == 0
        d = a.last_draws.cpu().numpy()
        from spherehand_amd import synth_rng
        f, keys = synth_rng.sample_draws(1234, 2, B, 0.1)
        assert np.array_equal(d[0:4].view(np.uint32), f.view(np.uint32)) and np.array_equal(d[4:6].view(np.uint32), keys)


@pytest.mark.parametrize("S", [64, 128, 256])
def test_noised_images_equal_depth_noise_on_the_restated_draws(S):
    """Noise on: every pixel = DepthNoise's formula (network/util_modules.py:60-84) applied to the clean image with the
    shifts and normals synth_rng derives from the sample's keys -- shifts exactly (background and >= 1.0 pixels must be
    bit-equal), the depth noise to 2e-5: the kernels evaluate Box-Muller with the hardware's log2 / sqrt / cos, and v_log_f32
    loses its relative accuracy where log2(u1) -> 0: for u1 within 2^-16 of 1 (one draw in 65 536) the normal -- ~0.004, a
    negligible sample -- comes out up to 1.5e-4 off (tools/fuzz.py: one pixel in ~5 million at 2-8e-6 of scaled depth);
    everything else agrees to 3e-7."""
    from spherehand_amd import synth_rng
    B = 5
    clean, noisy = _syn(S, add_noise=False, out_heatmap=False), _syn(S, add_noise=True, out_heatmap=False)
    pose = _poses(B, 4)
    torch.manual_seed(21)
    c = clean(pose)
    torch.manual_seed(21)
    n = noisy(pose)
    assert torch.equal(clean.last_draws.view(torch.int32), noisy.last_draws.view(torch.int32))
    keys = noisy.last_draws[4:6].cpu().numpy().view(np.uint32)
    expect = synth_rng.depth_noise(c.cpu().numpy(), keys, 0.5, 0.05)
    got = n.cpu().numpy()
    assert np.abs(got - expect).max() <= 2e-5 and (np.abs(got - expect) > 5e-7).mean() < 1e-5
    dx, dy, _ = synth_rng.noise_field(keys, S, S, 0.5)
    v = np.clip(np.arange(S)[None, :, None] + dy, 0, S - 1); u = np.clip(np.arange(S)[None, None, :] + dx, 0, S - 1)
    z = c.cpu().numpy()[np.arange(B)[:, None, None], v, u]
    bgd = z >= 1.0
    assert np.array_equal(got[bgd].view(np.uint32), z[bgd].view(np.uint32))      # shifted, no depth noise
    assert (got != c.cpu().numpy()).mean() > 0.02


def test_noise_distribution_matches_the_reference_formulas():
    """Parity in distribution with network/util_modules.py:60-84: the shifts' frequencies against
    (randn * 0.5 + 0.5).long() (exact probabilities from the normal CDF and a large torch sample), the depth noise
    against N(0, 0.05^2) (moments + Kolmogorov-Smirnov); RandScale / focal jitter against their uniform ranges."""
    import math
    from scipy import stats
    from spherehand_amd import synth_rng
    S, B = 128, 64
    noisy = _syn(S, add_noise=True, out_heatmap=False)
    clean = _syn(S, add_noise=False, out_heatmap=False)
    pose = _poses(B, 8)
    torch.manual_seed(5)
    c = clean(pose).cpu().numpy()
    torch.manual_seed(5)
    n = noisy(pose).cpu().numpy()
    keys = noisy.last_draws[4:6].cpu().numpy().view(np.uint32)
    dx, dy, nz = synth_rng.noise_field(keys, S, S, 0.5)
    N = dx.size
    Phi = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    prob = {-1: Phi(-3.0) - Phi(-5.0), 0: Phi(1.0) - Phi(-3.0), 1: Phi(3.0) - Phi(1.0), 2: Phi(5.0) - Phi(3.0)}
    tref = (torch.randn(4_000_000, generator=torch.Generator().manual_seed(0)) * 0.5 + 0.5).long().numpy()
    for d in (dx, dy):
        assert d.min() >= -1 and d.max() <= 2
        for k, p in prob.items():
            f = float((d == k).mean())
            assert abs(f - p) <= 5.0 * math.sqrt(p * (1 - p) / N) + 1e-5, (k, f, p)          # 5 sigma + the 16-bit table
            assert abs(float((tref == k).mean()) - p) <= 5.0 * math.sqrt(p * (1 - p) / tref.size)   # ... and torch agrees
    # x and y shifts of a pixel are independent draws
    assert abs(np.corrcoef(dx.ravel(), dy.ravel())[0, 1]) < 5.0 / math.sqrt(N)
    # the depth noise the KERNEL added, on interior foreground pixels whose source was not shifted across a silhouette
    v = np.clip(np.arange(S)[None, :, None] + dy, 0, S - 1); u = np.clip(np.arange(S)[None, None, :] + dx, 0, S - 1)
    z = c[np.arange(B)[:, None, None], v, u]
    fg = z < 1.0
    added = ((n - z)[fg] / 0.05).astype(np.float64)
    assert fg.sum() > 50_000
    assert abs(added.mean()) < 5.0 / math.sqrt(added.size) and abs(added.std() - 1.0) < 0.01
    assert stats.kstest(added[:200_000], "norm").pvalue > 1e-3
    assert abs(stats.kurtosis(added)) < 0.05 and abs(np.corrcoef(added[:-1], added[1:])[0, 1]) < 0.01
    # RandScale's factors and the focal jitter over many samples: uniform on their ranges
    f, _ = synth_rng.sample_draws(99, 0, 200_000, 0.1)
    for row, lo, width in ((f[0], 0.85, 0.1), (f[1], 0.85, 0.1), (f[2], 0.85, 0.1), (f[3], 0.9, 0.2)):
        assert stats.kstest((row.astype(np.float64) - lo) / width, "uniform").pvalue > 1e-3
    assert abs(np.corrcoef(f[0], f[1])[0, 1]) < 0.01 and abs(np.corrcoef(f[2], f[3])[0, 1]) < 0.01


def test_capture_and_replay_in_a_hipgraph():
    """The three launches captured once; every replay draws with the next call counter (new noise, new scales) and
    equals the eager module on that counter -- checked against the numpy restatement of the draws."""
    from spherehand_amd import synth_rng
    S, B = 64, 12
    syn = _syn(S)
    pose = _poses(B, 6)
    fresh = [_poses(B, 30 + k) for k in range(3)]          # (sample_poses seeds torch's generator: drawn before the seed below)
    torch.manual_seed(17)
    syn(pose)                                              # seeds the state (a host-to-device copy: not capturable)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        static_pose = pose.clone()
        syn(static_pose)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            outs = syn(static_pose)
        ctr0 = _state(syn)[1]
        results = []
        for k in range(3):
            static_pose.copy_(fresh[k])
            g.replay()
            stream.synchronize()
            assert _state(syn)[1] == ctr0 + k + 1
            results.append([o.clone() for o in outs] + [syn.last_draws.clone()])
    eager = _syn(S)
    for k, (depth, uv, dh, xyz, draws) in enumerate(results):
        f, keys = synth_rng.sample_draws(17, ctr0 + k, B, 0.1)
        assert np.array_equal(draws.cpu().numpy()[0:4].view(np.uint32), f.view(np.uint32))
        eager.reseed(17, device=pose.device, counter=ctr0 + k)        # (an explicit seed: torch's later seeds do not touch it)
        e = eager(fresh[k])
        for a, b in zip((depth, uv, dh, xyz), e):
            assert torch.equal(a, b)
    assert not torch.equal(results[0][0], results[1][0])


def test_two_devices():
    """ the >64 KB dynamic-LDS opt-in of the lattice / band kernels is per device: the second GPU of a process
    must render too.  Skipped on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU")
    from spherehand_amd import hand_model, ops
    from spherehand_amd.render import DepthRender
    mesh = hand_model.load_mesh()
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        dr = DepthRender(mesh, 128).to(dev)
        from spherehand_amd.kinematicsTransformation import HandTransformationMat
        fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
        with torch.no_grad():
            T = fk(_poses(4, 0).to(dev))
            outs.append(dr(T).cpu())
            fv = dr.lbs(T, dr.camera, None)[:, dr.rasterizer.faces, 0:3].reshape(4, -1, 3, 3).contiguous()
            outs.append(ops.tri_raster_fwd(640, 640, fv).cpu())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
