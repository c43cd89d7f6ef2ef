"""GPU parity: hand-written HIP sphere rasterizer (through the C ABI) vs the CPU
oracle and the reference's golden vectors.  Depth: BIT-EXACT.  Gradients: fp32
summation order differs from the oracle's fp64 accumulation -> tolerance
|err| <= 1e-5 * max|grad| + 1e-4 (stated per test)."""
import hashlib

import numpy as np
import pytest

from conftest import bits, golden, spheres_from

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["zbuf", "general", "zbuf-small-lds", "zbuf-4-waves", "zbuf-bands",
                                        "zbuf-no-run-table"])
def ops(request):
    """Every test runs on the fast LDS z-buffer kernels, on the general tile
    kernels, on the z-buffer kernels squeezed into 16 KB of LDS (many row regions
    per crop, several passes over the touched rows in the backward), with 4-wave forward /
    8-wave backward workgroups, and with a 12-KB forward z-buffer (the rows of a crop's
    touched box beyond it go through the tile code), and with the forward's run table off (whole-crop
    workgroups start their runs from it by default, i.e. under "zbuf")."""
    from spherehand_amd import ops as o
    assert torch.cuda.is_available()
    o.set_tuning(o.TUNE_FORCE_GENERAL, 1 if request.param == "general" else 0)
    o._shr_test_general = request.param == "general"
    small = request.param == "zbuf-small-lds"
    o.set_tuning(o.TUNE_FWD_LDS_BYTES, 16 * 1024 if small else 0)
    o.set_tuning(o.TUNE_FWD_OWNER_LDS_BYTES, 24 * 1024 if small else 0)
    o.set_tuning(o.TUNE_BWD_LDS_BYTES, 16 * 1024 if small else 128 * 1024)
    o.set_tuning(o.TUNE_FWD_WAVES, 4 if request.param == "zbuf-4-waves" else 16)
    o.set_tuning(o.TUNE_BWD_WAVES, 8 if request.param in ("zbuf-4-waves", "zbuf-bands") else 0)
    o.set_tuning(o.TUNE_FWD_ZBUF_BYTES, 12 * 1024 if request.param == "zbuf-bands" else 0)
    # the fused kernel's box variant: with half of a CU's LDS, and squeezed into 28 KB (rows of a box beyond that go
    # through the tile code)
    o.set_tuning(o.TUNE_MSE_BOX, {"zbuf-4-waves": 1, "zbuf-bands": 28 * 1024}.get(request.param, -1))
    o.set_tuning(o.TUNE_FWD_RUN_TABLE, 0 if request.param == "zbuf-no-run-table" else -1)
    yield o
    o._shr_test_general = False
    o.set_tuning(o.TUNE_FWD_RUN_TABLE, -1)
    o.set_tuning(o.TUNE_FWD_ZBUF_BYTES, 0)
    o.set_tuning(o.TUNE_MSE_BOX, -1)
    o.set_tuning(o.TUNE_BWD_WAVES, 0)
    o.set_tuning(o.TUNE_FWD_WAVES, 16)
    o.set_tuning(o.TUNE_FORCE_GENERAL, 0)
    o.set_tuning(o.TUNE_FWD_LDS_BYTES, 0)
    o.set_tuning(o.TUNE_FWD_OWNER_LDS_BYTES, 0)
    o.set_tuning(o.TUNE_BWD_LDS_BYTES, 128 * 1024)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_g1_rest_pose(ops):
    g = golden("g1_rest_pose.npz")
    sp = dev(spheres_from(g["centres"], g["radii"]))
    for S in (64, 128, 256):
        d, a = ops.sphere_raster_fwd(sp, S, S, want_argmin=True)
        ref = g["depth%d_ieee" % S]
        assert np.array_equal(bits(d.cpu().numpy()[0]), bits(ref)), S
        fg = ref < 100
        assert np.array_equal(a.cpu().numpy()[0][fg], g["argmin%d" % S][fg])
        assert np.all(a.cpu().numpy()[0][~fg] == 255)


def test_g3_batch256_full_size(ops, oracle):
    """BASELINE config 2 inputs: 256 crops, 128x128, 41 spheres."""
    g = golden("g3_batch256.npz")
    sp_h = spheres_from(g["centres"], g["radii"])
    d, a = ops.sphere_raster_fwd(dev(sp_h), 128, 128, want_argmin=True)
    d = d.cpu().numpy()
    sha = [hashlib.sha256(d[i].tobytes()).hexdigest() for i in range(256)]
    assert sha == list(g["depth_ieee_sha256"])
    od, oa = oracle.sphere_raster_fwd(sp_h, 128, 128)
    assert np.array_equal(bits(d), bits(od))
    assert np.array_equal(a.cpu().numpy(), oa)
    # without the argmin output: same depth
    d2 = ops.sphere_raster_fwd(dev(sp_h), 128, 128).cpu().numpy()
    assert np.array_equal(bits(d2), bits(d))


def test_g3_backward(ops, oracle):
    g = golden("g3_batch256.npz")
    sp_h = spheres_from(g["centres"], g["radii"])
    gd = np.random.RandomState(int(g["g_seed"])).standard_normal((256, 128, 128)).astype(np.float32)
    _, owner = ops.sphere_raster_fwd(dev(sp_h), 128, 128, want_argmin=True)
    gs = ops.sphere_raster_bwd(dev(sp_h), dev(gd), owner).cpu().numpy()      # saved owner map
    gs_re = ops.sphere_raster_bwd(dev(sp_h), dev(gd)).cpu().numpy()          # owners recomputed
    og = oracle.sphere_raster_bwd(sp_h, gd)
    tol = 1e-5 * np.abs(og).max() + 1e-4
    assert np.abs(gs - og).max() <= tol
    assert np.abs(gs_re - og).max() <= tol
    ref = g["grad_centres"]                       # reference autograd
    assert np.abs(gs[:, :, :3] - ref[:, :, :3]).max() <= tol
    gr = gs[:, :, 3].reshape(8, 32, 41).sum(1)
    assert np.abs(gr - g["grad_radii_chunk32"]).max() <= 1e-5 * np.abs(g["grad_radii_chunk32"]).max() + 1e-3
    # deterministic: bit-identical on a second launch
    gs2 = ops.sphere_raster_bwd(dev(sp_h), dev(gd), owner).cpu().numpy()
    assert np.array_equal(bits(gs), bits(gs2))


def test_ball_render_ragged_sizes(ops):
    """J == 1 is BallRender.forward; odd sizes take the scalar-store path."""
    g = golden("g_ballrender.npz")
    for t in "abcde":
        W, H, N = (int(x) for x in g[t + "_whn"])
        sp = np.concatenate([g[t + "_centres"], g[t + "_radii"][:, None]], 1).reshape(N, 1, 4)
        d = ops.sphere_raster_fwd(dev(sp), H, W).cpu().numpy()
        assert np.array_equal(bits(d), bits(g[t + "_maps_ieee"])), t


@pytest.mark.parametrize("N,J,H,W", [(5, 41, 64, 64), (3, 64, 96, 80), (4, 7, 53, 37), (2, 1, 8, 32),
                                       (1, 3, 1, 1), (6, 2, 130, 258), (2, 41, 256, 256), (3, 5, 24, 100)])
def test_random_spheres_vs_oracle(ops, oracle, N, J, H, W):
    rs = np.random.RandomState(N * 1000 + J)
    sp = np.concatenate([rs.uniform(-160, 160, (N, J, 2)), rs.uniform(-60, 120, (N, J, 1)),
                         rs.uniform(0.05, 45, (N, J, 1))], -1).astype(np.float32)
    d, a = ops.sphere_raster_fwd(dev(sp), H, W, want_argmin=True)
    od, oa = oracle.sphere_raster_fwd(sp, H, W)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    assert np.array_equal(a.cpu().numpy(), oa)
    gd = rs.standard_normal((N, H, W)).astype(np.float32)
    og = oracle.sphere_raster_bwd(sp, gd)
    for owner in (a, None):
        gs = ops.sphere_raster_bwd(dev(sp), dev(gd), owner).cpu().numpy()
        assert np.abs(gs - og).max() <= 1e-5 * np.abs(og).max() + 1e-4


@pytest.mark.parametrize("S", [64, 128, 256])
def test_narrow_boxes_near_the_last_rows(ops, oracle, S):
    """Narrow spheres (many rows per 64-lane chunk) whose boxes end on / just above the image's or a
    row region's last row: the scan's last chunk reaches below the box there -- into rows of other
    spheres, untouched rows and past the region.  Forward bit-exact, backward within the gradient
    tolerance, with and without the saved owner map."""
    rs = np.random.RandomState(S)
    px = 300.0 / S
    N, J = 6, 24
    sp = np.zeros((N, J, 4), np.float32)
    for n in range(N):
        for j in range(J):
            r = rs.uniform(1.2, 9.0) * px                      # boxes 2..18 px wide
            last = rs.choice([S - 1, S - 2, S - 3, S // 2 - 1, S // 2, S // 4 - 1, rs.randint(4, S)])
            yc = (last + rs.uniform(-0.4, 0.4) - S / 2) * px - r       # bottom edge of the disc near row `last`
            sp[n, j] = (rs.uniform(-140, 140), yc, rs.uniform(-40, 60), r)
    sp[:, -1, 2] = -80.0                                        # one sphere in front everywhere it reaches
    d, a = ops.sphere_raster_fwd(dev(sp), S, S, want_argmin=True)
    od, oa = oracle.sphere_raster_fwd(sp, S, S)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    assert np.array_equal(a.cpu().numpy(), oa)
    gd = rs.standard_normal((N, S, S)).astype(np.float32)
    og = oracle.sphere_raster_bwd(sp, gd)
    for owner in (a, None):
        gs = ops.sphere_raster_bwd(dev(sp), dev(gd), owner).cpu().numpy()
        assert np.isfinite(gs).all()
        assert np.abs(gs - og).max() <= 1e-5 * np.abs(og).max() + 1e-4


def test_every_sphere_a_candidate_everywhere(ops, oracle):
    """Spheres bigger than the image, some deeper than the background: the min
    may exceed 100 only where all J spheres hit (reference min over J maps)."""
    sp = np.array([[[0, 0, 450, 300], [10, -5, 500, 350]],
                   [[0, 0, 450, 300], [200, 0, 0, 20]]], np.float32)
    d, a = ops.sphere_raster_fwd(dev(sp), 32, 64, want_argmin=True)
    od, oa = oracle.sphere_raster_fwd(sp, 32, 64)
    assert od[0].min() > 100.0
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    assert np.array_equal(a.cpu().numpy(), oa)


def test_hit_at_exactly_the_background_depth(ops, oracle):
    """torch.min keeps the FIRST index at the minimum (reference mesh/render.py:89): a sphere whose hit lands on
    exactly 100.0 owns the pixel -- and receives its gradient -- only if no lower index holds 100 there, i.e. if
    every lower-index sphere hits the pixel behind the background.  (Found by tools/fuzz.py: the z-buffer used to
    let any such hit win the tie against the background.)"""
    S = 64
    px = lambda i: (i - S / 2) * 300.0 / S                    # pixel-centre coordinates: exact in fp32
    tie = lambda u, v: [px(u), px(v), 102.0, 2.0]             # covers ONE pixel, at depth 102 - sqrt(4) = 100.0
    far = [1e4, 1e4, 0.0, 1.0]                                # off screen
    front = lambda u, v, r=30.0: [px(u), px(v), 20.0, r]      # an ordinary sphere in front
    sp = np.array([
        [tie(40, 40), front(10, 10), far],                    # sphere 0 itself: owns the pixel (general path: z0 > 100)
        [front(10, 10), tie(40, 40), far],                    # sphere 0 misses the pixel: its background comes first
        [[px(40), px(40), 105.0, 3.0], tie(40, 40), front(10, 10)],   # sphere 0 hits it at 102: sphere 1's 100 wins
        [front(10, 10), tie(40, 40), tie(40, 40)],            # two such hits, behind sphere 0's background
        [front(40, 40), tie(40, 40), far],                    # sphere 0 in front there
        [far, tie(40, 40), front(10, 10)],                    # sphere 0 off screen
    ], np.float32)
    d, a = ops.sphere_raster_fwd(dev(sp), S, S, want_argmin=True)
    od, oa = oracle.sphere_raster_fwd(sp, S, S)
    assert od[0, 40, 40] == 100.0 and oa[0, 40, 40] == 0 and oa[1, 40, 40] == 255 and oa[2, 40, 40] == 1
    assert oa[3, 40, 40] == 255 and oa[4, 40, 40] == 0 and oa[5, 40, 40] == 255
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    assert np.array_equal(a.cpu().numpy(), oa)
    gd = np.random.RandomState(3).standard_normal((len(sp), S, S)).astype(np.float32)
    gd[:, 40, 40] = 1000.0                                    # the tie pixel's gradient must go where the reference sends it
    og = oracle.sphere_raster_bwd(sp, gd)
    tol = 1e-5 * np.abs(og).max() + 1e-4
    assert np.abs(ops.sphere_raster_bwd(dev(sp), dev(gd), a).cpu().numpy() - og).max() <= tol
    assert np.abs(ops.sphere_raster_bwd(dev(sp), dev(gd)).cpu().numpy() - og).max() <= tol      # owners recomputed
    tgt = np.full((len(sp), S, S), 100.0, np.float32); tgt[:, 40, 40] = -400.0
    if ops.sphere_raster_mse_supported(dev(sp), dev(tgt), S, S):
        dep, sse, gsp = ops.sphere_raster_mse(dev(sp), dev(tgt))
        assert np.array_equal(bits(dep.cpu().numpy()), bits(od))
        og2 = oracle.sphere_raster_bwd(sp, (2 * (od - tgt)).astype(np.float32))
        assert np.abs(gsp.cpu().numpy() - og2).max() <= 2e-5 * np.abs(og2).max() + 1e-3


def test_launches_with_two_workgroups_per_cu(oracle):
    """768 crops (the golden batch three times): the launcher's own choice is then the box z-buffer at half of a CU's
    LDS in the forward and 8-wave workgroups over the touched rows in the backward (two workgroups per CU).  Same bits
    / same gradients as the 256-crop launch (one whole-crop workgroup per CU) and as the golden hashes."""
    from spherehand_amd import ops
    g = golden("g3_batch256.npz")
    sp_h = spheres_from(g["centres"], g["radii"])
    sp3 = np.concatenate([sp_h, sp_h, sp_h])
    d1, a1 = ops.sphere_raster_fwd(dev(sp_h), 128, 128, want_argmin=True)
    d3, a3 = ops.sphere_raster_fwd(dev(sp3), 128, 128, want_argmin=True)
    d3h = d3.cpu().numpy()
    sha = [hashlib.sha256(d3h[i].tobytes()).hexdigest() for i in range(768)]
    assert sha == list(g["depth_ieee_sha256"]) * 3
    assert np.array_equal(a3.cpu().numpy(), np.concatenate([a1.cpu().numpy()] * 3))
    assert np.array_equal(bits(ops.sphere_raster_fwd(dev(sp3), 128, 128).cpu().numpy()), bits(d3h))     # depth only
    gd = np.random.RandomState(int(g["g_seed"])).standard_normal((256, 128, 128)).astype(np.float32)
    gs1 = ops.sphere_raster_bwd(dev(sp_h), dev(gd), a1).cpu().numpy()
    gs3 = ops.sphere_raster_bwd(dev(sp3), dev(np.concatenate([gd, gd, gd])), a3).cpu().numpy()
    og = oracle.sphere_raster_bwd(sp_h, gd)
    tol = 1e-5 * np.abs(og).max() + 1e-4
    for k in range(3):
        assert np.abs(gs3[256 * k:256 * (k + 1)] - og).max() <= tol
    assert np.abs(gs1 - og).max() <= tol


@pytest.mark.parametrize("S,reps", [(64, 1), (128, 1), (128, 3), (256, 1)])
def test_owner_map_on_touched_rows_only(ops, oracle, S, reps):
    """shr_sphere_raster_fwd_ex(SHR_RASTER_OWNER_TOUCHED_ROWS): what SphereDepthRaster saves for its backward.  The
    owner buffer starts as GARBAGE (random bytes, valid sphere indices among them): depth is complete and bit-exact,
    every row with a foreground pixel carries the oracle's owners, every other byte is either 255 or still the
    garbage it was (rows no sphere touches are not written: 8 of a hand crop's 82 KB), and the backward returns,
    bit for bit, what it returns with the complete owner map -- it never looks at the rows the forward skipped."""
    from spherehand_amd import _lib
    g = golden("g3_batch256.npz")
    sp_h = np.concatenate([spheres_from(g["centres"], g["radii"])] * reps)
    N = sp_h.shape[0]
    sp = dev(sp_h)
    rs = np.random.RandomState(S + reps)
    garbage = rs.randint(0, 256, (N, S, S)).astype(np.uint8)
    owner = dev(garbage)
    depth = torch.empty(N, S, S, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().shr_sphere_raster_fwd_ex(sp.data_ptr(), N, 41, S, S, depth.data_ptr(), owner.data_ptr(),
                                                   ops.RASTER_OWNER_TOUCHED_ROWS, st), "fwd_ex")
    od, oa = oracle.sphere_raster_fwd(sp_h, S, S)
    assert np.array_equal(bits(depth.cpu().numpy()), bits(od))
    a = owner.cpu().numpy()
    fg_row = (oa != 255).any(axis=2)                                    # [N, S]
    assert np.array_equal(a[fg_row], oa[fg_row])
    rest = ~fg_row
    assert np.all((a[rest] == 255) | (a[rest] == garbage[rest]))
    # on the z-buffer kernels the skipped rows really are skipped (the general tile kernels write every byte)
    kept = float((a[rest] == garbage[rest]).mean())
    d_full, a_full = ops.sphere_raster_fwd(sp, S, S, want_argmin=True)
    assert np.array_equal(a_full.cpu().numpy(), oa)
    gd = dev(rs.standard_normal((N, S, S)).astype(np.float32))
    g_part = ops.sphere_raster_bwd(sp, gd, owner).cpu().numpy()
    g_full = ops.sphere_raster_bwd(sp, gd, a_full).cpu().numpy()
    assert np.array_equal(bits(g_part), bits(g_full))
    if not _forced_general(ops):
        assert kept > 0.5, kept
    # and through the autograd Function (which passes the flag)
    spg = sp.clone().requires_grad_(True)
    (ops.SphereDepthRaster.apply(spg, S, S) * gd).sum().backward()
    assert np.array_equal(bits(spg.grad.cpu().numpy()), bits(g_full))


def _forced_general(ops):
    """True under the `general` parametrisation of the ops fixture (probed: a depth-only launch that the tile kernels
    serve writes the same bits either way, so the fixture's own state is read back through a tuning round trip)."""
    return getattr(ops, "_shr_test_general", False)


def test_nan_inf(ops, oracle):
    sp = np.array([[[0, 0, 10, 20], [np.nan, 0, 0, 5]],
                   [[0, 0, np.nan, 20], [50, 50, 0, 5]],
                   [[np.inf, 0, 0, 20], [0, 0, 5, 30]],
                   [[0, 0, 5, np.inf], [0, 0, 5, 30]]], np.float32)
    d = ops.sphere_raster_fwd(dev(sp), 16, 32).cpu().numpy()
    od = oracle.sphere_raster_fwd(sp, 16, 32, want_argmin=False)
    assert np.array_equal(np.isnan(d), np.isnan(od))
    assert np.array_equal(bits(d)[~np.isnan(od)], bits(od)[~np.isnan(od)])


def test_empty_batch_and_errors(ops):
    e = ops.sphere_raster_fwd(torch.empty((0, 41, 4), device="cuda"), 64, 64)
    assert e.shape == (0, 64, 64)
    with pytest.raises(RuntimeError):
        ops.sphere_raster_fwd(torch.zeros(1, 41, 4), 64, 64)            # CPU tensor
    with pytest.raises(RuntimeError):
        ops.sphere_raster_fwd(torch.zeros(1, 4, 41, device="cuda").transpose(1, 2), 64, 64)
    with pytest.raises(RuntimeError):
        ops.sphere_raster_fwd(torch.zeros(1, 65, 4, device="cuda"), 64, 64)  # > SHR_MAX_SPHERES
    with pytest.raises(RuntimeError):
        ops.sphere_raster_fwd(torch.zeros(1, 41, 4, device="cuda", dtype=torch.float64), 64, 64)


def test_backward_properties_full_size(ops):
    """Size-independent properties at BASELINE size (256 x 128 x 128):
    linearity in the upstream gradient, and sum_j dz_j == sum of g over the
    foreground (every foreground pixel routes its g to exactly one sphere)."""
    g = golden("g3_batch256.npz")
    sp = dev(spheres_from(g["centres"], g["radii"]))
    gen = torch.Generator(device="cuda").manual_seed(5)
    g1 = torch.randn(256, 128, 128, device="cuda", generator=gen)
    g2 = torch.randn(256, 128, 128, device="cuda", generator=gen)
    depth, owner = ops.sphere_raster_fwd(sp, 128, 128, want_argmin=True)
    b1, b2 = ops.sphere_raster_bwd(sp, g1, owner), ops.sphere_raster_bwd(sp, g2, owner)
    b12 = ops.sphere_raster_bwd(sp, 2.0 * g1 + g2, owner)
    assert (b12 - (2.0 * b1 + b2)).abs().max().item() <= 1e-5 * b12.abs().max().item() + 1e-3
    assert torch.equal(owner != 255, depth < 100)
    fg_sum = (g1.double() * (depth < 100)).sum(dim=(1, 2))
    assert (b1[:, :, 2].double().sum(1) - fg_sum).abs().max().item() <= 1e-3
    assert depth.max().item() == 100.0


def test_autograd_function(ops):
    g = golden("g1_rest_pose.npz")
    sp = dev(spheres_from(g["centres"], g["radii"])).requires_grad_(True)
    d = ops.SphereDepthRaster.apply(sp, 64, 64)
    w = torch.linspace(-1, 1, 64 * 64, device="cuda").view(1, 64, 64)
    (d * w).sum().backward()
    assert sp.grad.shape == (1, 41, 4)
    assert torch.isfinite(sp.grad).all() and sp.grad.abs().sum() > 0


def test_sqrt_rn_exhaustive():
    """The rasterizer's trimmed square root is correctly rounded on every fp32
    value in [0.01, 1e12] (q > 0.01 is the only range it is called on)."""
    from spherehand_amd import _lib
    lo = int(np.float32(0.0099).view(np.uint32))
    hi = int(np.float32(1e12).view(np.uint32))
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().shr_selftest_sqrt(lo, hi, bad.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "shr_selftest_sqrt")
    assert int(bad.item()) == 0
    assert hi - lo > 350_000_000


@pytest.mark.parametrize("S,wgs", [(64, 3), (128, 5), (256, 2)])
def test_persistent_workgroups_equal_one_workgroup_per_crop(oracle, S, wgs):
    """More crops than workgroups: a workgroup walks crops b, b + G, ... with the next crop's records prefetched.
    Same bits as one workgroup per crop (and as the oracle), including crops that take the general path in the
    middle of a workgroup's sequence (a NaN sphere; all spheres behind the background)."""
    from spherehand_amd import ops
    rs = np.random.RandomState(S)
    n, J = 23, 41
    sp = np.zeros((n, J, 4), np.float32)
    sp[..., 0:2] = rs.uniform(-90, 90, (n, J, 2)); sp[..., 2] = rs.uniform(-40, 40, (n, J)); sp[..., 3] = rs.uniform(8, 24, (n, J))
    sp[4, 7, 0] = np.nan                      # general path
    sp[9, :, 2] = 150.0                       # no sphere in front of the background: general path
    sp[10:12, :, 0] = 1e4                     # nothing on screen
    d_sp = torch.from_numpy(sp).cuda()
    g = torch.from_numpy(rs.standard_normal((n, S, S)).astype(np.float32)).cuda()
    res = {}
    for mode in (0, wgs, -wgs):
        ops.set_tuning(ops.TUNE_PERSISTENT, abs(mode))
        ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, 20 * 1024 if mode < 0 else 0)   # < 0: ... a small forward z-buffer
        ops.set_tuning(ops.TUNE_BWD_WAVES, 8 if mode < 0 else 0)                # ... and 8-wave backward workgroups
        ops.set_tuning(ops.TUNE_MSE_BOX, 1 if mode < 0 else 0)                  # ... and the fused kernel's box variant
        d, o = ops.sphere_raster_fwd(d_sp, S, S, want_argmin=True)
        d2 = ops.sphere_raster_fwd(d_sp, S, S)
        gs = ops.sphere_raster_bwd(d_sp, g, o)
        tgt = torch.full((5, S, S), 100.0, device="cuda"); tgt[:, S // 4:3 * S // 4, S // 4:3 * S // 4] = -5.0
        tidx = (torch.arange(n, device="cuda", dtype=torch.int32) % 5).contiguous()
        fd, fsse, fgrad = ops.sphere_raster_mse(d_sp, tgt, tidx)                       # fused render-and-compare
        res[mode] = [t.cpu().numpy() for t in (d, o, d2, gs, fd, fsse, fgrad)]
    ops.set_tuning(ops.TUNE_PERSISTENT, 1)
    ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, 0)
    ops.set_tuning(ops.TUNE_BWD_WAVES, 0)
    ops.set_tuning(ops.TUNE_MSE_BOX, -1)
    for other in (wgs, -wgs):
        for i, (a, b) in enumerate(zip(res[0], res[other])):
            if i in (3, 5, 6) and other < 0:
                # 8-wave backward / box variant of the fused kernel: another summation order (crop 4 has a NaN sphere)
                fin = np.isfinite(a)
                assert np.array_equal(fin, np.isfinite(b))
                assert np.abs(a[fin] - b[fin]).max() <= 2e-5 * np.abs(a[fin]).max() + 1e-3
            else:
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    ref = oracle.sphere_raster_fwd(sp, S, S, want_argmin=False)
    ok = ~np.isnan(ref)                                  # (a NaN's payload is not part of the contract)
    assert np.array_equal(np.isnan(res[wgs][0]), ~ok)
    assert np.array_equal(res[wgs][0][ok].view(np.uint32), ref[ok].view(np.uint32))


@pytest.mark.parametrize("S,J", [(128, 41), (128, 64), (64, 41), (256, 41), (32, 5)])
def test_run_table_equals_the_arithmetic_run_start(oracle, S, J):
    """Whole-crop forward workgroups start a run on a sphere from an LDS table (built once per sphere by the idle
    waves) instead of recomputing the lane layout, column term, cell and row coordinate per run: same operations,
    so depth and owner map are bit-identical with the table on and off, and equal to the oracle's.  Hand crops, random
    spheres (boxes clipped by the last row: the row-test path), spheres wider than a wave (arithmetic path inside
    the table kernel) and J = 64 (the table does not fit behind a 64-bit 128 x 128 z-buffer: launcher falls back)."""
    from spherehand_amd import ops
    rs = np.random.RandomState(S + J)
    g = golden("g3_batch256.npz")
    hand = spheres_from(g["centres"], g["radii"])[:24]
    if J != 41:
        hand = np.concatenate([hand, hand], 1)[:, :J] if J > 41 else hand[:, :J]
    rnd = np.concatenate([rs.uniform(-160, 160, (24, J, 2)), rs.uniform(-60, 120, (24, J, 1)),
                          rs.uniform(0.05, 45, (24, J, 1))], -1).astype(np.float32)
    wide = rnd.copy()
    wide[:, : J // 2, 3] = rs.uniform(80, 140, (24, J // 2))       # boxes wider than 64 lanes at S >= 128
    sp = np.ascontiguousarray(np.concatenate([hand, rnd, wide], 0).astype(np.float32))
    out = {}
    try:
        for mode in (0, -1):
            ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, mode)
            for flags in (0, 1):
                d, a = ops.sphere_raster_fwd(dev(sp), S, S, want_argmin=True, flags=flags)
                out[mode, flags] = (d.cpu().numpy(), a.cpu().numpy())
            out[mode, "depth"] = ops.sphere_raster_fwd(dev(sp), S, S).cpu().numpy()
    finally:
        ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, -1)
    od, oa = oracle.sphere_raster_fwd(sp, S, S)
    for mode in (0, -1):
        assert np.array_equal(bits(out[mode, 0][0]), bits(od)), mode
        assert np.array_equal(out[mode, 0][1], oa), mode
        assert np.array_equal(bits(out[mode, 1][0]), bits(od)), mode
        assert np.array_equal(bits(out[mode, "depth"]), bits(od)), mode
        fg = od < 100                                        # (touched-rows mode: untouched rows stay unwritten)
        assert np.array_equal(out[mode, 1][1][fg], oa[fg]), mode
