"""GPU parity of the multiview projection path vs the reference's golden vectors
(tests/golden/g4_mutual_projection.npz, B=4, V=3, 64x64)."""
import numpy as np
import pytest

from conftest import bits, golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_projected_points_and_depth():
    from spherehand_amd import ops
    from spherehand_amd.multiview_utility import MutualProjection, MutualTransformation
    g = golden("g4_mutual_projection.npz")
    mp = MutualProjection(64, list(g["radii"])).cuda()
    depth, pts = mp(dev(g["cam"]), dev(g["inv_cam"]), dev(g["joints"]))
    assert depth.shape == (4, 3, 3, 64, 64) and pts.shape == (4, 3, 3, 41, 3, 1)
    # 4x4 product + 3x3 transform: association differs from the reference's bmm -> tolerance
    assert np.abs(pts.cpu().numpy() - g["projected_points"]).max() <= 2e-4
    # the rasterizer fed the reference's own projected points is bit-exact
    sph = np.concatenate([g["projected_points"][..., 0], np.broadcast_to(g["radii"][None, None, None, :, None],
                                                                          (4, 3, 3, 41, 1))], -1).astype(np.float32)
    d = ops.sphere_raster_fwd(dev(sph.reshape(36, 41, 4)), 64, 64).cpu().numpy().reshape(4, 3, 3, 64, 64)
    assert np.array_equal(bits(d), bits(g["proj_ieee"]))
    # end to end: only silhouette pixels may flip (centres differ by <= 2e-4 mm)
    dd = depth.cpu().numpy()
    flipped = (dd >= 100) != (g["proj_ieee"] >= 100)
    assert flipped.mean() < 1e-4
    assert np.abs(dd - g["proj_ieee"])[~flipped].max() < 5e-3
    mt = MutualTransformation()(dev(g["cam"]), dev(g["inv_cam"]))
    ref = np.einsum("bjrk,bikc->bijrc", g["inv_cam"], g["cam"])
    assert np.abs(mt.cpu().numpy() - ref).max() < 1e-5


@pytest.mark.parametrize("is_mv", [True, False])
def test_mutual_projection_loss(is_mv):
    """Loss: the depth MSE term sees (100 - depth)^2 ~ 1e4 per flipped silhouette
    pixel, so 1e-4 relative is the natural bar; gradients 1e-3 relative to max."""
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    g = golden("g4_mutual_projection.npz")
    crit = MutualProjectionLoss(64, list(g["radii"])).cuda()
    joints = dev(g["joints"]).requires_grad_(True)
    loss, proj = crit(dev(g["cam"]), dev(g["inv_cam"]), joints, dev(g["real_dms"]), is_mv)
    loss.backward()
    tag = "mv" if is_mv else "diag"
    ref = float(g[tag + "_loss"])
    assert abs(loss.item() - ref) <= 1e-4 * ref
    gr = g[tag + "_grad_joints"]
    assert np.abs(joints.grad.cpu().numpy() - gr).max() <= 1e-3 * np.abs(gr).max()
    assert proj.shape == (4, 3, 3, 64, 64)


def test_multiview_consistency_loss():
    from spherehand_amd.multiview_utility import MultiviewConsistencyLoss
    g = golden("g4_mutual_projection.npz")
    crit = MultiviewConsistencyLoss()
    joints = dev(g["joints"]).requires_grad_(True)
    l = crit(dev(g["cam"]), joints)              # consistent views: the loss is rounding noise
    assert abs(l.item()) < 1e-9 and abs(float(g["mvc_loss"])) < 1e-9
    joints = dev(g["mvc2_joints"]).requires_grad_(True)
    l = crit(dev(g["cam"]), joints)
    l.backward()
    assert abs(l.item() - float(g["mvc2_loss"])) <= 1e-5 * float(g["mvc2_loss"])
    assert np.abs(joints.grad.cpu().numpy() - g["mvc2_grad_joints"]).max() <= 1e-5 * np.abs(g["mvc2_grad_joints"]).max()
    joints = dev(g["mvc2_joints"]).requires_grad_(True)
    l = crit(dev(g["cam"]), joints, dev(g["mvc2_hm_weight"]))
    l.backward()
    assert abs(l.item() - float(g["mvc2w_loss"])) <= 1e-5 * float(g["mvc2w_loss"])
    assert np.abs(joints.grad.cpu().numpy() - g["mvc2w_grad_joints"]).max() <= 1e-5 * np.abs(g["mvc2w_grad_joints"]).max()


def test_mutual_project_gradient_is_transpose():
    """<spheres.xyz, G> differentiated w.r.t. joints equals sum_j R^T G (adjoint test)."""
    from spherehand_amd import ops
    g = golden("g4_mutual_projection.npz")
    cam, inv = dev(g["cam"]), dev(g["inv_cam"])
    joints = dev(g["joints"]).requires_grad_(True)
    radii = dev(g["radii"])
    sph = ops.MutualProject.apply(cam, inv, joints, radii)
    G = torch.randn_like(sph)
    (sph * G).sum().backward()
    M = torch.matmul(inv.unsqueeze(1), cam.unsqueeze(2))[..., :3, :3]           # [B,i,j,3,3]
    ref = torch.einsum("bijrc,bijkr->bikc", M, G[..., :3])
    assert (joints.grad - ref).abs().max().item() < 1e-4
    assert torch.equal(sph[..., 3], radii.view(1, 1, 1, -1).expand(4, 3, 3, 41))


def test_config5_per_gpu_shard_properties(oracle):
    """BASELINE configs[4]: 256x256, batch 1024 over 8 GPUs -> 128 samples x 9 view pairs
    = 1152 crops per GPU.  Full-size run of projection + raster + both losses, checked by
    size-independent properties and an oracle spot check."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjection, MutualProjectionLoss
    mesh = hand_model.load_mesh()
    B, V, S = 128, 3, 256
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=5)
    cam, inv, real = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    truth = ds.joints.cuda()
    mp = MutualProjection(S, mesh).cuda()
    depth, pts = mp(cam, inv, truth)
    assert depth.shape == (B, V, V, S, S)
    # every view's true joints rendered into view j reproduce view j's observation: up to the
    # rounding of the two rigid transforms (<= 1e-3 mm) only pixels on a silhouette or on an
    # occlusion edge between two spheres may differ
    obs = real.unsqueeze(1).expand(B, V, V, S, S)
    assert ((depth - obs).abs() > 0.05).float().mean().item() < 2e-4
    # (the oracle comparison of every crop: test_config5_full_size_against_the_oracle)
    sph = torch.cat([pts.squeeze(-1), mp.radiuses.view(1, 1, 1, -1, 1).expand(B, V, V, -1, 1)], -1)
    # the loss at the truth is (near) its minimum; away from it, larger
    crit = MutualProjectionLoss(S, mesh).cuda()
    j0 = truth.clone().requires_grad_(True)
    l0, _ = crit(cam, inv, j0, real, True)
    l0.backward()
    j1 = (truth + 2.0 * torch.randn_like(truth)).requires_grad_(True)
    l1, _ = crit(cam, inv, j1, real, True)
    l1.backward()
    assert torch.isfinite(l0) and torch.isfinite(l1) and l1.item() > 5 * l0.item()
    assert torch.isfinite(j0.grad).all() and torch.isfinite(j1.grad).all()   # (|x| kink: no zero gradient at the truth)
    # gradient-sum identity of the rasterizer at this size: sum_j dL/dz_j == sum of g over foreground
    g = torch.randn(B * V * V, S, S, device="cuda")
    spheres = sph.view(-1, 41, 4).contiguous()
    d, owner = ops.sphere_raster_fwd(spheres, S, S, want_argmin=True)
    gs = ops.sphere_raster_bwd(spheres, g, owner)
    fg_sum = (g.double() * (d < 100)).sum(dim=(1, 2))
    assert (gs[:, :, 2].double().sum(1) - fg_sum).abs().max().item() < 5e-3


def test_config5_full_size_against_the_oracle(oracle):
    """BASELINE configs[4] at its per-GPU size -- ALL 1152 crops @256x256 -- against the CPU oracle (OpenMP over the
    host's cores: seconds), what mesh/multiview_utility.py:55-130 must match:
      forward          depth bits and owner maps of every crop (bit-exact), with the full owner map and in the autograd
                       pair's touched-rows mode;
      backward         d<g, depth>/d spheres for an N(0,1) upstream (1e-5 of the largest entry + 1e-4: fp32 summation
                       order against the oracle's fp64 sums);
      fused            render-and-compare: projected depth bits, per-crop sum of squares (2e-5) and its gradient
                       against the oracle's backward of 2 (depth - observed);
      data -> model    per-crop loss sums (2e-7: fp32 output rounding) and unit gradients (1e-5 of the largest entry, but for the unit vectors of
                       pixels equidistant from two spheres) through the image index;
      the criterion    MutualProjectionLoss(is_mv=True): loss (1e-4) and d loss / d joints (1e-3 of the largest entry)
                       against the value ASSEMBLED from the oracle's pieces with the reference's weights (9 x MSE +
                       500 x 9 x data->model, :98-105, :129) and pulled back through the view transforms in fp64."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    mesh = hand_model.load_mesh()
    B, V, S, J = 128, 3, 256, 41
    N = B * V * V
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=5)
    cam, inv, real = ds.cam.cuda(), ds.inv_cam.cuda(), ds.dms.cuda()
    joints = (ds.joints.cuda() + 1.5 * torch.randn(ds.joints.shape, device="cuda",
                                                   generator=torch.Generator(device="cuda").manual_seed(3)))
    crit = MutualProjectionLoss(S, mesh).cuda()
    radii = crit.mutual_projection.radiuses.view(-1)
    sph = ops.MutualProject.apply(cam, inv, joints, radii).view(N, J, 4).contiguous()
    sph_h = sph.cpu().numpy()
    # ---- forward: every crop's depth and owner map
    od, oa = oracle.sphere_raster_fwd(sph_h, S, S, want_argmin=True)
    d, a = ops.sphere_raster_fwd(sph, S, S, want_argmin=True)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od)) and np.array_equal(a.cpu().numpy(), oa)
    d2, a2 = ops.sphere_raster_fwd(sph, S, S, want_argmin=True, flags=ops.RASTER_OWNER_TOUCHED_ROWS)
    assert torch.equal(d2, d)
    fg_rows = torch.from_numpy((od < 100).any(2)).cuda()                       # rows with foreground are touched rows
    assert torch.equal(a2[fg_rows], a[fg_rows])
    # ---- backward for a random upstream gradient (from either owner map)
    g = torch.randn(N, S, S, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    og = oracle.sphere_raster_bwd(sph_h, g.cpu().numpy())
    tol = 1e-5 * np.abs(og).max() + 1e-4
    for owner in (a, a2):
        assert np.abs(ops.sphere_raster_bwd(sph, g, owner).cpu().numpy() - og).max() <= tol
    del g, d2, a2
    # ---- fused render-and-compare against the observed images (crop (b,i,j) vs image b*V+j)
    obs = real.view(B * V, S, S).contiguous()
    index = crit._indices(B, V, sph.device)[0]
    obs_h = obs.cpu().numpy()
    idx_h = index.cpu().numpy()
    e_h = od.astype(np.float64) - obs_h[idx_h]
    sse_ref = (e_h ** 2).sum((1, 2))
    gs_ref = oracle.sphere_raster_bwd(sph_h, (2.0 * e_h).astype(np.float32))
    proj, sse, gs = ops.sphere_raster_mse(sph, obs, index)
    assert np.array_equal(bits(proj.cpu().numpy()), bits(od))
    assert np.abs(sse.double().cpu().numpy() - sse_ref).max() <= 2e-5 * sse_ref.max()
    assert np.abs(gs.cpu().numpy() - gs_ref).max() <= 2e-5 * np.abs(gs_ref).max() + 1e-3
    # ---- data -> model through the same index
    cen = sph[..., :3].contiguous()
    exp_h = obs_h[idx_h]                                                        # the reference's 3x expansion (:99)
    dl_ref = oracle.data_to_model_fwd(exp_h, sph_h[..., :3], radii.cpu().numpy())
    dg_ref = oracle.data_to_model_bwd(exp_h, sph_h[..., :3], radii.cpu().numpy()).astype(np.float64) * exp_h.size
    # (the oracle returns d mean / d centres; the kernel the unit gradient of the per-crop sums)
    dl, dg = ops.data_to_model(obs, cen, radii, want_grad=True, depth_index=index)
    # (every point's term is the oracle's arithmetic for its owner -- a correctly rounded root, the host's association --
    # and the sums are integers: the per-crop results agree to the fp32 rounding of the output)
    assert np.abs(dl.double().cpu().numpy() - dl_ref).max() <= 2e-7 * dl_ref.max()
    # (a pixel equidistant from two spheres to within an ulp of the root -- the search compares through v_sqrt_f32 --
    # may be assigned to either: the loss is continuous there, the pixel's UNIT vector moves between two gradient
    # entries.  Ten million foreground pixels hold a few of those: entries off by more than the rounding bar must be
    # rare and off by no more than a couple of unit vectors)
    dd = np.abs(dg.cpu().numpy() - dg_ref)
    off = dd > 1e-5 * np.abs(dg_ref).max() + 1e-6
    assert off.sum() <= 6 * 16 and dd.max() <= 2.5, (off.sum(), dd.max())     # <= 16 such pixels x (2 spheres x 3 components)
    del exp_h, e_h
    # ---- the criterion, assembled from the oracle's pieces
    count = float(N) * S * S
    loss_ref = 9.0 * sse_ref.sum() / count + 500.0 * 9.0 * dl_ref.sum() / count
    gp = (9.0 / count) * gs_ref[..., :3].astype(np.float64) + (500.0 * 9.0 / count) * dg_ref
    M = np.matmul(inv.cpu().numpy().astype(np.float64)[:, None], cam.cpu().numpy().astype(np.float64)[:, :, None])
    gj_ref = np.einsum("bijrc,bijkr->bikc", M[..., :3, :3], gp.reshape(B, V, V, J, 3))
    jr = joints.clone().requires_grad_(True)
    loss, projected = crit(cam, inv, jr, real, True)
    loss.backward()
    assert abs(loss.item() - loss_ref) <= 1e-4 * abs(loss_ref)
    assert np.array_equal(bits(projected.view(N, S, S).cpu().numpy()), bits(od))
    assert np.abs(jr.grad.cpu().numpy() - gj_ref).max() <= 1e-3 * np.abs(gj_ref).max()


def _random_spheres(rs, n, j=41, spread=80.0):
    sp = np.zeros((n, j, 4), np.float32)
    sp[..., 0:2] = rs.uniform(-spread, spread, (n, j, 2))
    sp[..., 2] = rs.uniform(-60, 60, (n, j))
    sp[..., 3] = rs.uniform(6, 26, (n, j))
    return sp


@pytest.fixture
def mse_box(request):
    """The fused kernel's launch shape: -1 the launcher's choice, 1 the box variant (touched-box z-buffer at half of a
    CU's LDS), a byte count: the box variant squeezed into that much LDS (rows beyond it go through the tile code)."""
    from spherehand_amd import ops
    ops.set_tuning(ops.TUNE_MSE_BOX, request.param)
    yield request.param
    ops.set_tuning(ops.TUNE_MSE_BOX, -1)


@pytest.mark.parametrize("mse_box", [-1, 1, 20 * 1024], indirect=True)
@pytest.mark.parametrize("S,H", [(64, 64), (128, 128), (256, 256), (96, 72), (320, 200), (256, 128), (32, 64)])
def test_fused_render_and_compare_equals_composition(S, H, mse_box):
    """shr_sphere_raster_mse == raster fwd + (depth - target)^2 + raster bwd (the unfused chain)."""
    from spherehand_amd import ops
    rs = np.random.RandomState(S + H)
    n, m = 14, 5
    sp = _random_spheres(rs, n)
    sp[3, 7, 0] = np.nan                         # general path: NaN sphere
    sp[5, :, 2] = 150.0                          # general path: no sphere in front of the background
    sp[6, :, 0] = 1e4                            # nothing on screen
    target = np.full((m, H, S), 100.0, np.float32)
    target[:, H // 4: 3 * H // 4, S // 4: 3 * S // 4] = rs.uniform(-50, 50, (m, H // 2, S // 2))
    index = rs.randint(0, m, n).astype(np.int32)
    spd, tgd, ixd = dev(sp), dev(target), dev(index)
    assert ops.sphere_raster_mse_supported(spd, tgd, H, S)
    depth, sse, grad = ops.sphere_raster_mse(spd, tgd, ixd)
    d_ref, owner = ops.sphere_raster_fwd(spd, H, S, want_argmin=True)
    assert np.array_equal(bits(depth.cpu().numpy()), bits(d_ref.cpu().numpy()))          # NaNs included
    e = d_ref - tgd[ixd.long()]
    sse_ref = (e.double() ** 2).sum((1, 2)).cpu().numpy()
    g_ref = np.zeros((n, 41, 4), np.float32)
    fast = [k for k in range(n) if k not in (3, 5)]
    g_ref[fast] = ops.sphere_raster_bwd(spd[fast].contiguous(), (2 * e)[fast].contiguous(), owner[fast].contiguous()).cpu().numpy()
    g_ref[[3, 5]] = ops.sphere_raster_bwd(spd[[3, 5]].contiguous(), (2 * e)[[3, 5]].contiguous()).cpu().numpy()   # owners recomputed
    s_, g_ = sse.cpu().numpy(), grad.cpu().numpy()
    ok = np.isfinite(sse_ref)
    assert np.array_equal(np.isfinite(s_), ok)
    assert np.abs(s_[ok] - sse_ref[ok]).max() <= 2e-5 * np.abs(sse_ref[ok]).max()
    okg = np.isfinite(g_ref)
    assert np.array_equal(np.isfinite(g_), okg)
    assert np.abs(g_[okg] - g_ref[okg]).max() <= 2e-5 * np.abs(g_ref[okg]).max() + 1e-3
    # without the depth output
    none, sse2, grad2 = ops.sphere_raster_mse(spd, tgd, ixd, want_depth=False)
    assert none is None and torch.equal(torch.nan_to_num(sse2), torch.nan_to_num(sse))
    assert torch.equal(torch.nan_to_num(grad2), torch.nan_to_num(grad))


@pytest.mark.parametrize("is_mv", [True, False])
def test_fused_mutual_projection_loss_equals_reference_wiring(is_mv):
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    g = golden("g4_mutual_projection.npz")
    crit = MutualProjectionLoss(64, list(g["radii"])).cuda()
    out = {}
    for fused in (True, False):
        crit.fused = fused
        joints = dev(g["joints"]).requires_grad_(True)
        loss, proj = crit(dev(g["cam"]), dev(g["inv_cam"]), joints, dev(g["real_dms"]), is_mv)
        loss.backward()
        out[fused] = (loss.item(), proj.detach().cpu().numpy(), joints.grad.cpu().numpy())
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * abs(out[False][0])
    assert np.array_equal(bits(out[True][1]), bits(out[False][1]))
    assert np.abs(out[True][2] - out[False][2]).max() <= 2e-5 * np.abs(out[False][2]).max()


def test_data_to_model_indexed_equals_expanded():
    from spherehand_amd import ops
    g = golden("g5_data_to_model.npz")
    dms, joints, radii = dev(g["a_dms"]), dev(g["a_joints"]), dev(g["a_radii"])
    n = joints.shape[0]
    index = torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda")
    a, ga = ops.data_to_model(dms, joints, radii, want_grad=True, depth_index=index)
    b, gb = ops.data_to_model(dms[index.long()].contiguous(), joints, radii, want_grad=True)
    assert torch.equal(a, b) and torch.equal(ga, gb)


@pytest.mark.parametrize("J", [1, 7, 64])
def test_fused_render_and_compare_sphere_counts(J):
    from spherehand_amd import ops
    rs = np.random.RandomState(J)
    sp = _random_spheres(rs, 9, J)
    target = rs.uniform(-40, 100, (9, 64, 64)).astype(np.float32)
    spd, tgd = dev(sp), dev(target)
    depth, sse, grad = ops.sphere_raster_mse(spd, tgd)
    d_ref, owner = ops.sphere_raster_fwd(spd, 64, 64, want_argmin=True)
    assert torch.equal(depth, d_ref)
    e = d_ref - tgd
    g_ref = ops.sphere_raster_bwd(spd, (2 * e).contiguous(), owner)
    assert (sse.double() - (e.double() ** 2).sum((1, 2))).abs().max().item() <= 2e-5 * sse.abs().max().item()
    assert (grad - g_ref).abs().max().item() <= 2e-5 * g_ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("V", [3, 4, 2, 5])
def test_mv_consistency_kernel_matches_the_torch_formulation(V):
    """MultiviewConsistencyLoss (mesh/multiview_utility.py:138-167): the one-launch kernel (value + gradient routed
    through torch.median's selected view) against the torch ops it replaces; the value itself is pinned to the
    reference by g7's 'mv_consistency' term (test_engine_gpu.py)."""
    from spherehand_amd import ops
    from spherehand_amd.datasets import random_rotations
    from spherehand_amd.multiview_utility import MultiviewConsistencyLoss
    g = torch.Generator().manual_seed(V)
    B, J = 7, 41
    cam = torch.eye(4).repeat(B, V, 1, 1)
    cam[:, :, :3, :3] = random_rotations(B * V, 40.0, g).view(B, V, 3, 3)
    cam[:, :, :3, 3] = torch.randn(B, V, 3, generator=g) * 5
    joints = torch.randn(B, V, J, 3, generator=g) * 40
    joints[0, :, 3] = joints[0, 0:1, 3]                     # nearly tied canonical points exercise the rank rule
    cam, joints = cam.cuda(), joints.cuda()

    def torch_loss(cam, joints):
        R = cam[:, :, None, 0:3, 0:3]
        t = cam[:, :, None, 0:3, 3].unsqueeze(-1)
        canonical = torch.matmul(R, joints.unsqueeze(-1)) + t
        med, _ = torch.median(canonical, dim=1)
        return torch.nn.functional.mse_loss(med.unsqueeze(1).expand_as(canonical), canonical)

    a = joints.clone().requires_grad_(True)
    la = MultiviewConsistencyLoss()(cam, a)
    (la * 3.0).backward()
    b = joints.clone().requires_grad_(True)
    lb = torch_loss(cam, b)
    (lb * 3.0).backward()
    assert ops.mv_consistency_supported(cam, a)
    assert abs(la.item() - lb.item()) <= 1e-5 * abs(lb.item())
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * b.grad.abs().max().item() + 1e-7
    # no-grad forward gives the same value
    with torch.no_grad():
        assert MultiviewConsistencyLoss()(cam, joints).item() == la.item()


@pytest.mark.parametrize("ksize", [3, 5])
def test_depth_resample_kernel_matches_the_module_ops(ksize):
    """DepthResample (network/util_modules.py:10-43): drop-out to 1.0 + the fixed Gaussian in one launch, against
    torch.where + the module's nn.Conv2d fed the same uniform draws."""
    from spherehand_amd import ops
    from spherehand_amd.util_modules import DepthResample
    mod = DepthResample(0.95, ksize).cuda()
    dm = (torch.rand(5, 48, 40, device="cuda") * 1.2).contiguous()
    g = torch.Generator(device="cuda").manual_seed(3)
    out = ops.depth_resample(dm, 0.95, ksize, generator=g)
    g.manual_seed(3)
    u = torch.rand((5, 48, 40), device="cuda", generator=g)
    x = torch.where(u > 0.95, torch.ones_like(dm), dm).unsqueeze(1)
    ref = mod.gaussian_filter(x)
    assert out.shape == ref.shape == (5, 1, 48, 40)
    assert (out - ref).abs().max().item() <= 2e-6
    with torch.no_grad():
        z = mod(dm)                                           # the module takes the kernel on CUDA tensors
    assert z.shape == (5, 1, 48, 40) and 0.0 < (z - dm.unsqueeze(1)).abs().max().item() < 1.3


def test_mutual_projection_loss_on_an_empty_batch():
    """B = 0: the entry points return at once, so the wrapper must not hand out uninitialised memory: loss 0, an empty
    projection, a zero gradient."""
    from spherehand_amd import hand_model
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    crit = MutualProjectionLoss(64, hand_model.load_mesh()).cuda()
    cam = torch.zeros(0, 3, 4, 4, device="cuda")
    joints = torch.zeros(0, 3, 41, 3, device="cuda", requires_grad=True)
    loss, proj = crit(cam, cam, joints, torch.zeros(0, 3, 64, 64, device="cuda"), True)
    assert loss.item() == 0.0 and proj.shape == (0, 3, 3, 64, 64)
    loss.backward()
    assert joints.grad is not None and joints.grad.shape == joints.shape
