"""GPU: (1) the TWO-STEP data->model path (shr_data_to_model_compact + shr_data_to_model_from_points: every observed
image compacted once into a tile-sorted point list, the lists searched per crop) and (2) the streaming kernel with TILE units (32 x 8 pixels, a search bounded by its points' own x-y box; round 4)
against the same kernel with round 2's units (256 consecutive pixels, strip bounds) -- mesh/render.py:123-142.

Bar.  All of them run ONE implementation of the per-point search (csrc/d2m_search.h); only the grouping of the points and the
bound that prunes the spheres differ, and both bounds are conservative.  The sums are fixed-point integers, so for the
same number of partial results per crop the outputs are BIT-IDENTICAL floats -- at every image shape, sphere count,
launch shape and band size."""
import numpy as np
import pytest

from conftest import bits

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _spheres(rs, n, j=41, spread=70.0):
    sp = np.zeros((n, j, 4), np.float32)
    sp[..., 0:2] = rs.uniform(-spread, spread, (n, j, 2))
    sp[..., 2] = rs.uniform(-60, 60, (n, j))
    sp[..., 3] = rs.uniform(6, 26, (n, j))
    return sp


def _observed(rs, m, H, W, fg=0.3):
    """Background 100 with blobs of foreground of both signs, some exactly 99 / just above it."""
    t = np.full((m, H, W), 100.0, np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(m):
        for _ in range(3):
            cy, cx, r = rs.uniform(0, H), rs.uniform(0, W), rs.uniform(0.1, 0.35) * min(H, W) * (fg / 0.3)
            mask = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
            t[k][mask] = rs.uniform(-60, 60) + rs.normal(0, 8, int(mask.sum()))
    t[0, 0, 0:4] = (99.0, 99.00001, 98.99999, 100.0)
    return t


def _d2m(obs, index, sph, parts):
    """shr_data_to_model_partial on the rasterizer's records in place."""
    from spherehand_amd import _lib, ops
    lib = _lib.lib()
    N, J, _ = sph.shape
    H, W = obs.shape[1:]
    radii = sph[0, :, 3].contiguous()
    loss = torch.empty((N, parts), device="cuda")
    grad = torch.empty((N, parts, J, 3), device="cuda")
    _lib.check(lib.shr_data_to_model_partial(obs.data_ptr(), index.data_ptr(), sph.data_ptr(), 4, radii.data_ptr(), N, J, H, W,
                                             parts, loss.data_ptr(), grad.data_ptr(), ops._stream()), "d2m")
    return loss, grad


@pytest.fixture(autouse=True)
def two_step_at_any_size():
    """The module-level entries take the two-step path only above ops.D2M_TWO_STEP_MIN_PIXELS (where it pays); here it
    is what is under test, at every size."""
    from spherehand_amd import ops
    keep = ops.D2M_TWO_STEP_MIN_PIXELS
    ops.D2M_TWO_STEP_MIN_PIXELS = 0
    yield
    ops.D2M_TWO_STEP_MIN_PIXELS = keep


@pytest.fixture
def tune():
    from spherehand_amd import ops
    yield ops
    ops.set_tuning(ops.TUNE_D2M_TILED, -1)
    ops.set_tuning(ops.TUNE_D2M_WAVES, 0)
    ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, 0)


def _both(ops, od, ixd, spd, parts):
    ops.set_tuning(ops.TUNE_D2M_TILED, 0)
    ref = _d2m(od, ixd, spd, parts)
    ops.set_tuning(ops.TUNE_D2M_TILED, 1)
    return ref, _d2m(od, ixd, spd, parts)


def _two_step(obs, index, sph, parts):
    from spherehand_amd import ops
    ws = ops.d2m_compact(obs)
    N, J, _ = sph.shape
    from spherehand_amd import _lib
    lib = _lib.lib()
    radii = sph[0, :, 3].contiguous()
    loss = torch.empty((N, parts), device="cuda")
    grad = torch.empty((N, parts, J, 3), device="cuda")
    _lib.check(lib.shr_data_to_model_from_points(ws.data_ptr(), obs.shape[0], index.data_ptr(), sph.data_ptr(), 4, radii.data_ptr(),
                                                 N, J, obs.shape[1], obs.shape[2], parts, loss.data_ptr(), grad.data_ptr(),
                                                 ops._stream()), "from_points")
    return loss, grad


@pytest.mark.parametrize("parts", [1, 2, 3, 7])
@pytest.mark.parametrize("W,H", [(64, 64), (128, 128), (256, 256), (96, 72), (320, 200), (256, 128), (32, 64), (8, 8),
                                 (4, 40), (1024, 16), (36, 52), (100, 31), (512, 512)])
def test_two_step_sums_are_bit_identical_to_the_streaming_kernel(W, H, parts, tune):
    ops = tune
    rs = np.random.RandomState(W * 7 + H)
    n, m = 13, 5
    sp = _spheres(rs, n)
    sp[:, :, 3] = sp[0, :, 3]                    # one radius per sphere index (radii[J])
    sp[6, :, 0] = 1e4                            # nothing near: every point clamps at 50
    obs = _observed(rs, m, H, W)
    index = rs.randint(0, m, n).astype(np.int32)
    spd, od, ixd = dev(sp), dev(obs), dev(index)
    assert ops.d2m_points_supported(od)
    ops.set_tuning(ops.TUNE_D2M_TILED, 0)
    rl, rg = _d2m(od, ixd, spd, 1)               # round 2's kernel, one partial per crop
    tl, tg = _two_step(od, ixd, spd, parts)
    if parts == 1:
        assert np.array_equal(bits(tl.cpu().numpy()), bits(rl.cpu().numpy()))
        assert np.array_equal(bits(tg.cpu().numpy()), bits(rg.cpu().numpy()))
    else:                                        # the partial results are floats: their sum is the total to a rounding each
        a, b = tl.double().sum(1).cpu().numpy(), rl.double().view(n).cpu().numpy()
        assert np.abs(a - b).max() <= 3e-7 * np.abs(b).max()
        ga, gb = tg.double().sum(1).cpu().numpy(), rg.double().view(n, 41, 3).cpu().numpy()
        assert np.abs(ga - gb).max() <= 3e-7 * np.abs(gb).max() + 1e-6
    assert float(tl.sum()) > 0
    # the module-level entry takes the two-step path on its own
    l2, g2 = ops.data_to_model(od, spd[:, :, :3].contiguous(), spd[0, :, 3].contiguous(), want_grad=True, depth_index=ixd)
    assert (l2.double() - rl.double().view(n)).abs().max().item() <= 3e-7 * rl.abs().max().item()


@pytest.mark.parametrize("S", [64, 256])
def test_two_step_dense_empty_and_nan(S, tune):
    """Every pixel foreground (full regions), all background (empty lists), half images, NaN / infinite pixels, a NaN
    record: as the streaming kernel, NaN crops included."""
    ops = tune
    rs = np.random.RandomState(S)
    n = 9
    sp = _spheres(rs, n)
    sp[:, :, 3] = sp[0, :, 3]
    sp[7, 5, 2] = np.nan
    obs = rs.uniform(-50, 60, (n, S, S)).astype(np.float32)
    obs[1] = 100.0
    obs[2, : S // 2] = 200.0
    obs[3, :, : S // 2] = 100.0
    obs[4, 3, 9] = np.nan
    obs[5, S - 1, S - 1] = np.inf
    spd, od = dev(sp), dev(obs)
    ix = torch.arange(n, dtype=torch.int32, device="cuda")
    ops.set_tuning(ops.TUNE_D2M_TILED, 0)
    rl, rg = _d2m(od, ix, spd, 1)
    tl, tg = _two_step(od, ix, spd, 1)
    a, b = tl.view(n).cpu().numpy(), rl.view(n).cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(b[[4, 7]]).all()
    ok = ~np.isnan(b)
    assert np.array_equal(bits(a[ok]), bits(b[ok]))
    assert np.array_equal(bits(tg.view(n, 41, 3).cpu().numpy()[ok]), bits(rg.view(n, 41, 3).cpu().numpy()[ok]))
    assert float(tl[1].sum()) == 0.0 and float(tg[1].abs().max()) == 0.0


@pytest.mark.parametrize("J", [1, 7, 64])
def test_two_step_sphere_counts_and_wave_counts(J, tune):
    ops = tune
    rs = np.random.RandomState(J)
    n, S = 6, 64
    sp = _spheres(rs, n, J)
    sp[:, :, 3] = sp[0, :, 3]
    obs = _observed(rs, n, S, S)
    spd, od = dev(sp), dev(obs)
    ix = torch.arange(n, dtype=torch.int32, device="cuda")
    ops.set_tuning(ops.TUNE_D2M_TILED, 0)
    rl, rg = _d2m(od, ix, spd, 1)
    for waves in (4, 8, 16):
        ops.set_tuning(ops.TUNE_D2M_WAVES, waves)
        tl, tg = _two_step(od, ix, spd, 1)
        assert torch.equal(tl, rl) and torch.equal(tg, rg), waves


@pytest.mark.parametrize("is_mv", [True, False])
@pytest.mark.parametrize("S", [64, 128, 256])
def test_mutual_projection_loss_with_either_data_to_model_path(S, is_mv):
    """MutualProjectionLossFused through the two-step path vs the streaming kernel: the same per-point terms -- the
    scalar and d loss / d joints agree to the float conversions of the partial results."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    B = 6
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=2, device="cuda")
    crit = MutualProjectionLoss(S, mesh).cuda()
    out = {}
    try:
        for two in (True, False):
            ops.D2M_TWO_STEP = two
            j = (ds.joints.cuda() + 1.5 * torch.randn(ds.joints.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1))).requires_grad_(True)
            loss, proj = crit(ds.cam.cuda(), ds.inv_cam.cuda(), j, ds.dms.cuda(), is_mv)
            loss.backward()
            out[two] = (float(loss.detach()), proj.detach().clone(), j.grad.clone())
    finally:
        ops.D2M_TWO_STEP = True
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * abs(out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    assert (out[True][2] - out[False][2]).abs().max().item() <= 2e-6 * out[False][2].abs().max().item()


@pytest.mark.parametrize("W,H", [(64, 64), (128, 128), (256, 256), (96, 72), (320, 200), (256, 128), (32, 64), (8, 8),
                                 (4, 40), (1024, 16), (36, 52), (100, 31)])
def test_tile_units_give_bit_identical_sums(W, H, tune):
    parts = 1
    ops = tune
    rs = np.random.RandomState(W * 7 + H)
    n, m = 13, 5
    sp = _spheres(rs, n)
    sp[:, :, 3] = sp[0, :, 3]                    # one radius per sphere index (radii[J])
    sp[6, :, 0] = 1e4                            # nothing near: every point clamps at 50
    obs = _observed(rs, m, H, W)
    index = rs.randint(0, m, n).astype(np.int32)
    spd, od, ixd = dev(sp), dev(obs), dev(index)
    (rl, rg), (tl, tg) = _both(ops, od, ixd, spd, parts)
    assert np.array_equal(bits(tl.cpu().numpy()), bits(rl.cpu().numpy()))
    assert np.array_equal(bits(tg.cpu().numpy()), bits(rg.cpu().numpy()))
    assert float(tl.sum()) > 0


@pytest.mark.parametrize("waves,band", [(4, 1), (4, 3), (8, 2), (16, 5), (16, 40)])
def test_tile_units_any_launch_shape(waves, band, tune):
    """The sums do not depend on waves per workgroup or band size (integer accumulation)."""
    ops = tune
    rs = np.random.RandomState(waves + band)
    n, S = 7, 128
    sp = _spheres(rs, n)
    sp[:, :, 3] = sp[0, :, 3]
    obs = _observed(rs, n, S, S)
    spd, od = dev(sp), dev(obs)
    ix = torch.arange(n, dtype=torch.int32, device="cuda")
    ref_l, ref_g = _d2m(od, ix, spd, 1)
    ops.set_tuning(ops.TUNE_D2M_WAVES, waves)
    ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, band)
    l, g = _d2m(od, ix, spd, 1)
    assert torch.equal(l, ref_l) and torch.equal(g, ref_g)


@pytest.mark.parametrize("S", [64, 256])
def test_tile_units_dense_and_empty_images(S, tune):
    """Every pixel foreground (every unit fills the ring), all background, half images."""
    ops = tune
    rs = np.random.RandomState(S)
    n = 9
    sp = _spheres(rs, n)
    sp[:, :, 3] = sp[0, :, 3]
    obs = rs.uniform(-50, 60, (n, S, S)).astype(np.float32)
    obs[1] = 100.0
    obs[2, : S // 2] = 200.0
    obs[3, :, : S // 2] = 100.0
    spd, od = dev(sp), dev(obs)
    ix = torch.arange(n, dtype=torch.int32, device="cuda")
    (rl, rg), (tl, tg) = _both(ops, od, ix, spd, 1)     # (one partial per crop: with more, the two kernels cut a crop differently)
    assert np.array_equal(bits(tl.cpu().numpy()), bits(rl.cpu().numpy()))
    assert np.array_equal(bits(tg.cpu().numpy()), bits(rg.cpu().numpy()))
    assert float(tl[1].sum()) == 0.0 and float(tg[1].abs().max()) == 0.0


def test_tile_units_nan_and_sphere_counts(tune):
    """A NaN record / NaN or infinite observed value: the crop's loss is NaN (torch.min / clamp propagate it), the
    others are untouched; J = 1, 7, 64 spheres."""
    ops = tune
    rs = np.random.RandomState(3)
    for J in (1, 7, 64):
        n, S = 6, 64
        sp = _spheres(rs, n, J)
        sp[:, :, 3] = sp[0, :, 3]
        obs = _observed(rs, n, S, S)
        if J == 7:
            sp[2, 3, 1] = np.nan
            obs[4, 10, 10] = np.nan
            obs[5, 11, 12] = -np.inf
        spd, od = dev(sp), dev(obs)
        ix = torch.arange(n, dtype=torch.int32, device="cuda")
        (rl, rg), (tl, tg) = _both(ops, od, ix, spd, 1)
        a, b = tl.view(n).cpu().numpy(), rl.view(n).cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b)), (J, a, b)
        ok = ~np.isnan(b)
        assert np.array_equal(bits(a[ok]), bits(b[ok])), J
        assert np.array_equal(bits(tg.view(n, J, 3).cpu().numpy()[ok]), bits(rg.view(n, J, 3).cpu().numpy()[ok])), J
        if J == 7:
            assert np.isnan(b[[2, 4]]).all()


def test_point_list_cache_follows_the_observations():
    """MutualProjectionLoss keeps the compacted point lists while it is handed the same observations (same storage,
    same version): a second call reuses them, an in-place change or another tensor rebuilds them -- and the values
    are those of the uncached module either way."""
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, 4, 64, seed=5, device="cuda")
    cam, inv, dms, joints = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda(), ds.joints.cuda() + 0.7
    crit, ref = MutualProjectionLoss(64, mesh).cuda(), MutualProjectionLoss(64, mesh).cuda()
    ref.cache_points = False
    a, _ = crit(cam, inv, joints, dms, True)
    ws = crit._points[2]
    b, _ = crit(cam, inv, joints * 1.01, dms, True)
    assert crit._points[2] is ws                                  # same observations: the lists were reused
    assert float(b) == float(ref(cam, inv, joints * 1.01, dms, True)[0]) and ref._points is None
    dms[:, 0, 10:20, 10:20] = 35.0                                # in-place change: version counter moves
    c, _ = crit(cam, inv, joints, dms, True)
    assert crit._points[2] is not ws and float(c) == float(ref(cam, inv, joints, dms, True)[0]) and float(c) != float(a)
    ws2 = crit._points[2]
    d, _ = crit(cam, inv, joints, dms.clone(), True)              # another tensor with the same values
    assert crit._points[2] is not ws2 and float(d) == float(c)


@pytest.mark.parametrize("B,V,S", [(2, 3, 64), (5, 3, 128), (1, 2, 32), (700, 1, 8)])
def test_project_and_compact_in_two_launches_equals_the_two_calls(B, V, S):
    """shr_mv_project_compact (the view projection clears the point lists' fill counters, the compaction follows without
    its memset) == shr_mutual_project_fwd + shr_data_to_model_compact: same sphere records, same sums from the lists --
    also into a workspace that is full of stale counters, and with more images than projected records (B*V > B*V*V*J
    never happens with J >= 1, but M > the projection's thread count rounded to workgroups is covered by V = 1, S = 8)."""
    from spherehand_amd import _lib, ops
    lib = _lib.lib()
    rs = np.random.RandomState(B * 100 + S)
    J = 41 if B < 100 else 1
    cam = np.tile(np.eye(4, dtype=np.float32), (B, V, 1, 1))
    cam[..., :3, 3] = rs.uniform(-20, 20, (B, V, 3))
    cam[..., :3, :3] += rs.uniform(-0.05, 0.05, (B, V, 3, 3)).astype(np.float32)
    inv = np.linalg.inv(cam).astype(np.float32)
    joints = rs.uniform(-60, 60, (B, V, J, 3)).astype(np.float32)
    radii = rs.uniform(6, 20, J).astype(np.float32)
    obs = dev(_observed(rs, B * V, S, S))
    camd, invd, jd, rd = dev(cam), dev(inv), dev(joints), dev(radii)
    want_sph = ops.MutualProject.apply(camd, invd, jd, rd).view(B * V * V, J, 4)
    ws = ops.d2m_points_workspace(obs)
    ws.fill_(0x5a)                                                # stale counters and points
    sph = torch.empty_like(want_sph)
    _lib.check(lib.shr_mv_project_compact(camd.data_ptr(), invd.data_ptr(), jd.data_ptr(), rd.data_ptr(), B, V, J, sph.data_ptr(),
                                          obs.data_ptr(), B * V, S, S, ws.data_ptr(), ops._stream()), "shr_mv_project_compact")
    assert np.array_equal(bits(sph.cpu().numpy()), bits(want_sph.cpu().numpy()))
    index = dev((np.arange(B)[:, None, None] * V + np.arange(V)[None, None, :] + np.zeros((1, V, 1), np.int64)).reshape(-1).astype(np.int32))
    N = B * V * V
    out = []
    for w in (ws, ops.d2m_compact(obs)):
        loss = torch.empty((N, 1), device="cuda")
        grad = torch.empty((N, 1, J, 3), device="cuda")
        _lib.check(lib.shr_data_to_model_from_points(w.data_ptr(), B * V, index.data_ptr(), want_sph.data_ptr(), 4, rd.data_ptr(), N, J,
                                                     S, S, 1, loss.data_ptr(), grad.data_ptr(), ops._stream()), "from_points")
        out.append((bits(loss.cpu().numpy()), bits(grad.cpu().numpy())))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    nb = ws.numel() - 4 * B * V
    counts_a = ws[nb:].view(torch.int32).cpu().numpy()
    counts_b = w[nb:].view(torch.int32).cpu().numpy()
    assert np.array_equal(counts_a, counts_b) and int(counts_a.sum()) == int((obs <= 99).sum())


@pytest.mark.parametrize("is_mv", [True, False])
def test_render_and_compare_beside_the_point_search_gives_the_same_bits(is_mv):
    """ops.MV_OVERLAP: the fused render-and-compare kernel on a side stream while the point search runs on the caller's
    -- loss, projected depth and d loss / d joints are bit-identical to the one-stream order, call after call (the
    joins are stream waits; a missing one would show as stale partial results)."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, 24, 128, seed=4, device="cuda")
    crit = MutualProjectionLoss(128, mesh).cuda()
    crit.cache_points = False
    cam, inv, dms = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    gen = torch.Generator("cuda").manual_seed(3)
    keep = ops.MV_OVERLAP
    try:
        for it in range(6):
            base = ds.joints.cuda() + 2.0 * torch.randn(ds.joints.shape, device="cuda", generator=gen)
            out = []
            for mode in (False, True):
                ops.MV_OVERLAP = mode
                j = base.clone().requires_grad_(True)
                loss, proj = crit(cam, inv, j, dms, is_mv)
                loss.backward()
                out.append((loss.detach().clone(), proj.detach().clone(), j.grad.clone()))
            assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2]), it
    finally:
        ops.MV_OVERLAP = keep


def test_loss_on_a_callers_own_stream_with_the_side_stream_beside_it():
    """The caller's current stream is not the default one: the launches follow it (ops._stream), the side stream forks
    from and joins it -- same bits as on the default stream, call after call."""
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, 12, 128, seed=6, device="cuda")
    crit = MutualProjectionLoss(128, mesh).cuda()
    cam, inv, dms = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    base = ds.joints.cuda() + 1.0
    want = []
    for k in range(3):
        j = (base + 0.3 * k).requires_grad_(True)
        loss, proj = crit(cam, inv, j, dms, True)
        loss.backward()
        want.append((loss.detach().clone(), proj.detach().clone(), j.grad.clone()))
    torch.cuda.synchronize()
    own = torch.cuda.Stream()
    own.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(own):
        for k in range(3):
            j = (base + 0.3 * k).requires_grad_(True)
            loss, proj = crit(cam, inv, j, dms, True)
            loss.backward()
            got = (loss.detach(), proj.detach(), j.grad)
            own.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(got, want[k])), k


@pytest.mark.parametrize("S,B", [(64, 5), (128, 12), (256, 4)])
def test_same_view_pairs_compared_alone_give_the_same_results(S, B):
    """is_mv = False (the reference after its first 1500 iterations): ops.SAME_VIEW_SPLIT renders all V*V projections
    with the plain forward kernel and runs the fused render-and-compare kernel on the B*V same-view pairs alone
    (shr_mv_loss_combine with is_mv = 2) -- projection bit-identical, loss and d loss / d joints bit-identical to the
    one-launch wiring (same per-pair values, same order of summation)."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=8, device="cuda")
    crit = MutualProjectionLoss(S, mesh).cuda()
    cam, inv, dms = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    gen = torch.Generator("cuda").manual_seed(5)
    keep = ops.SAME_VIEW_SPLIT
    try:
        for it in range(3):
            base = ds.joints.cuda() + 2.0 * torch.randn(ds.joints.shape, device="cuda", generator=gen)
            out = []
            for mode in (False, True):
                ops.SAME_VIEW_SPLIT = mode
                j = base.clone().requires_grad_(True)
                loss, proj = crit(cam, inv, j, dms, False)
                loss.backward()
                out.append((loss.detach().clone(), proj.detach().clone(), j.grad.clone()))
            assert torch.equal(out[0][1], out[1][1]), it
            assert torch.equal(out[0][2], out[1][2]), it
            assert abs(out[0][0].item() - out[1][0].item()) <= 1e-6 * abs(out[0][0].item()), it
    finally:
        ops.SAME_VIEW_SPLIT = keep


@pytest.mark.parametrize("is_mv", [True, False])
@pytest.mark.parametrize("cache", [True, False])
def test_loss_module_captured_in_a_hipgraph_replays_on_new_inputs(is_mv, cache):
    """MutualProjectionLoss forward + backward captured in ONE hipGraph -- two-step data->model path, render-and-compare
    on the side stream (fork and join captured with it), the same-view split when is_mv is off, the point-list cache on
    and off -- after an eager warm-up that leaves a filled cache behind (the case where a stale hit would keep the
    compaction out of the graph).  Every replay on NEW joints and NEW observations copied into the static buffers must
    equal the eager result of a fresh module on those inputs bit for bit: loss, projections, d loss / d joints."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    B, S = 12, 128
    ds = [SyntheticMultiviewDataset(mesh, B, S, seed=20 + k, device="cuda") for k in range(3)]
    cam, inv = ds[0].cam.cuda(), ds[0].inv_cam.cuda()
    keep = ops.D2M_TWO_STEP_MIN_PIXELS
    ops.D2M_TWO_STEP_MIN_PIXELS = 1 << 18                        # the two-step path at this (small) size
    try:
        crit = MutualProjectionLoss(S, mesh).cuda()
        crit.cache_points = cache
        ref = MutualProjectionLoss(S, mesh).cuda()
        ref.cache_points = False
        static_j = (ds[0].joints.cuda() + 0.5).requires_grad_(True)
        static_dms = ds[0].dms.cuda().clone()
        assert ops.d2m_two_step_pays(static_dms.view(B * 3, S, S))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                             # eager warm-up: fills the cache when it is on
            for _ in range(2):
                loss, _ = crit(cam, inv, static_j, static_dms, is_mv)
                loss.backward()
                static_j.grad = None
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, proj = crit(cam, inv, static_j, static_dms, is_mv)
            loss.backward()
        for k in (1, 2, 0, 1):
            with torch.no_grad():
                static_j.copy_(ds[k].joints.cuda() + 0.25 * k)
                static_dms.copy_(ds[k].dms.cuda())
            g.replay()
            torch.cuda.synchronize()
            j = (ds[k].joints.cuda() + 0.25 * k).requires_grad_(True)
            want_loss, want_proj = ref(cam, inv, j, ds[k].dms.cuda(), is_mv)
            want_loss.backward()
            assert torch.equal(loss, want_loss.detach()), (k, loss.item(), want_loss.item())
            assert torch.equal(proj, want_proj.detach()), k
            assert torch.equal(static_j.grad, j.grad), k
        # and eagerly again afterwards, on changed observations in the same buffer (version counter moved by copy_)
        l2, _ = crit(cam, inv, static_j.detach(), static_dms, is_mv)
        assert torch.equal(l2, ref(cam, inv, static_j.detach(), static_dms.clone(), is_mv)[0])
    finally:
        ops.D2M_TWO_STEP_MIN_PIXELS = keep


@pytest.mark.parametrize("S,B", [(64, 5), (128, 12), (256, 43)])
@pytest.mark.parametrize("is_mv", [True, False])
def test_loss_without_materialised_projections_is_the_same_loss(S, B, is_mv):
    """MutualProjectionLoss.return_projections = False (what Engine's epoch loops set: nobody reads the projected depth
    maps there): the fused kernel writes no depth output, the same-view mode launches no plain forward -- loss and
    d loss / d joints bit-identical to the default, which returns the reference's (loss, projected_dms); sizes below
    and above the two-step / two-stream threshold."""
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=9, device="cuda")
    full, lean = MutualProjectionLoss(S, mesh).cuda(), MutualProjectionLoss(S, mesh).cuda()
    lean.return_projections = False
    cam, inv, dms = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    for k in range(2):
        j1 = (ds.joints.cuda() + 0.4 * (k + 1)).requires_grad_(True)
        j2 = j1.detach().clone().requires_grad_(True)
        l1, p1 = full(cam, inv, j1, dms, is_mv)
        l2, p2 = lean(cam, inv, j2, dms, is_mv)
        assert p2 is None and p1.shape == (B, 3, 3, S, S)
        l1.backward()
        l2.backward()
        if is_mv:
            assert torch.equal(l1, l2)
        else:      # (the default may take the one-launch wiring on a small stack: same per-pair values, another summation order)
            assert abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())
        assert torch.equal(j1.grad, j2.grad)


@pytest.mark.parametrize("S,B", [(64, 8), (128, 16), (256, 48)])
@pytest.mark.parametrize("lean", [False, True])
def test_xcd_aware_launch_order_gives_the_same_bits(S, B, lean):
    """All pairs, B * V a multiple of 8: the fused render-and-compare kernel and the point search are launched in the
    order that puts the V pairs of one observed image on one XCD (shr_sphere_raster_mse_ordered,
    shr_data_to_model_from_points_ordered; multiview_utility._indices) -- against the batch's own order: loss,
    projections and d loss / d joints bit for bit, below and above the two-step threshold."""
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=13, device="cuda")
    ordered, plain = MutualProjectionLoss(S, mesh).cuda(), MutualProjectionLoss(S, mesh).cuda()
    ordered.return_projections = plain.return_projections = not lean
    ordered.xcd_order_min_crops = 0                         # (the module orders stacks of >= 512 crops only)
    cam, inv, dms = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    plain._indices(B, 3, cam.device)
    plain._order = plain._order_target = None               # the batch's own launch order
    for k in range(2):
        j1 = (ds.joints.cuda() + 0.3 * (k + 1)).requires_grad_(True)
        j2 = j1.detach().clone().requires_grad_(True)
        l1, p1 = ordered(cam, inv, j1, dms, True)
        assert ordered._order is not None and sorted(ordered._order.tolist()) == list(range(B * 9))
        l2, p2 = plain(cam, inv, j2, dms, True)
        l1.backward()
        l2.backward()
        assert torch.equal(l1, l2) and torch.equal(j1.grad, j2.grad)
        assert lean or torch.equal(p1, p2)
