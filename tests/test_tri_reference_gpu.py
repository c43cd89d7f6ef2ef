"""GPU: the REFERENCE's own triangle kernel, executed on the MI355X, against the C oracle and the HIP kernels.

oracle/_ref/libref_tri.so is the device code of /root/reference/mesh/cuda_kernel/depth_rasterization_cuda_kernel.cu
(`kernel` and its `atomicMin`, lines 1-113) compiled for gfx950 where it lies by oracle/Makefile (target ref_tri; built
in the container that has the reference, shipped to the GPU box as a built file).  Its launcher (lines 115-133: a
1000-filled map, one single-thread workgroup per face) is restated by the recipe's C entry because it no longer
compiles against this torch.  Two builds: -ffp-contract=off -- every operation as the source writes it, the build of
record -- and clang's default contraction.

Bar: BIT-EXACT between the reference kernel (as written), the C oracle (oracle_tri_raster_fwd) and
depth_rasterization.forward (tri_raster_kernel), on the hand mesh at the reference's 640 x 640, on the quirk cases and
on random soups.  The contracted build is reported, not asserted equal: fusing a*b+c changes the near-singular 1/z sums
-- which is also why nvcc's default (-fmad=true, a fusion pattern of its own) cannot be reproduced bit for bit here."""
import time

import numpy as np
import pytest

from conftest import bits, golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ref(oracle):
    h = oracle.ref_tri_lib()
    if h is None:
        pytest.skip("oracle/_ref/libref_tri.so not built (needs /root/reference at build time: make -C oracle ref_tri)")
    return h


def run_ref(h, fv, W, H):
    v = dev(fv)
    out = torch.empty((fv.shape[0], H, W), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rc = h.ref_tri_forward(v.data_ptr(), fv.shape[0], fv.shape[1], W, H, out.data_ptr())
    assert rc == 0, rc
    return out.cpu().numpy()


def soup(rs, B, F, W, H, spreads=(3.0, 12.0, 40.0), zr=(-50, 50)):
    c = rs.uniform(-20, [W + 20, H + 20], (B, F, 1, 2))
    spread = rs.choice(spreads, (B, F, 1, 1))
    return np.concatenate([c + rs.normal(0, 1, (B, F, 3, 2)) * spread, rs.uniform(zr[0], zr[1], (B, F, 3, 1))], -1).astype(np.float32)


def test_reference_kernel_on_the_hand_mesh(ref, oracle):
    """4 poses x 3382 faces at 640 x 640 (g2: the reference's skinning and camera produced the vertices)."""
    import depth_rasterization
    fv = golden("g2_mesh.npz")["face_vertices"]
    r = run_ref(ref, fv, 640, 640)
    assert np.array_equal(bits(r), bits(oracle.tri_raster_fwd(fv, 640, 640)))
    assert np.array_equal(bits(r), bits(depth_rasterization.forward(640, 640, dev(fv)).cpu().numpy()))
    assert np.array_equal(bits(r[0]), bits(golden("g2_mesh.npz")["raw640_first"]))
    assert int((r < 1000).sum()) > 200000
    r2 = run_ref(ref, fv[:1], 320, 640)                          # width != height
    assert np.array_equal(bits(r2), bits(oracle.tri_raster_fwd(fv[:1], 320, 640)))


def test_reference_kernel_quirks(ref, oracle):
    """Truncating bounding boxes, extrapolated rows left of the image, degenerate / NaN / behind-the-camera faces."""
    import depth_rasterization
    tri = np.array([
        [[-0.5, -0.7, 5], [-0.2, 3.0, 5], [-0.1, -0.6, 5]],
        [[2, 2, 0], [2, 9, 4], [9, 2, 4]],
        [[5, 5, 3], [5, 9, 3], [5, 7, 3]],
        [[1, 1, 3], [4, 4, 3], [7, 7, 3]],
        [[np.nan, 1, 3], [4, 2, 3], [7, 9, 3]],
        [[3, 12, 2], [12, 3, 2], [3, 3, -2]],
        [[-40, -30, 7], [60, -20, 7], [10, 70, 7]],
        [[-0.7, 7.1, 5], [-3.2, 14.3, 7], [-9.4, 7.6, 6]],
        [[1e9, 2, 3], [4, -1e9, 3], [7, 9, 3]],                  # coordinates beyond int32 after the casts
        [[3, 3, 0], [9, 3, 0], [3, 9, 0]],                       # z = 0 at every vertex: 1/z
    ], np.float32)[None]
    for fv in (tri, tri[:, ::-1], tri[:, :, [1, 0, 2], :], tri[:, :, [2, 1, 0], :]):
        for (W, H) in ((16, 16), (17, 9), (5, 33)):
            r = run_ref(ref, fv, W, H)
            assert np.array_equal(bits(r), bits(oracle.tri_raster_fwd(fv, W, H))), (W, H)
            assert np.array_equal(bits(r), bits(depth_rasterization.forward(W, H, dev(fv)).cpu().numpy())), (W, H)
    assert np.all(run_ref(ref, np.zeros((2, 0, 3, 3), np.float32), 8, 8) == 1000.0)


@pytest.mark.parametrize("seed", range(40))
def test_reference_kernel_on_random_soups(ref, oracle, seed):
    import depth_rasterization
    rs = np.random.RandomState(100 + seed)
    W = int(rs.choice([5, 33, 96, 200, 320])); H = int(rs.choice([7, 40, 72, 160, 256]))
    B = int(rs.randint(1, 4)); F = int(rs.choice([1, 33, 200, 500]))
    fv = soup(rs, B, F, W, H, spreads=(0.7, 3.0, 12.0, 60.0))
    if seed % 3 == 0:
        fv[:, :, :, 0] = np.round(fv[:, :, :, 0])                # vertices on pixel columns, equal x
    if seed % 4 == 1:
        fv[:, ::7, 1] = fv[:, ::7, 0]                            # degenerate faces
    r = run_ref(ref, fv, W, H)
    o = oracle.tri_raster_fwd(fv, W, H)
    assert np.array_equal(bits(r), bits(o)), int((bits(r) != bits(o)).sum())
    assert np.array_equal(bits(r), bits(depth_rasterization.forward(W, H, dev(fv)).cpu().numpy()))


def test_fused_depth_render_against_the_reference_kernel(ref):
    """DepthRasterization's fused kernel (raster + clamp + bilinear resize, only the sampled source pixels) == clamp +
    F.interpolate over the REFERENCE kernel's own 640 x 640 image of the same face vertices (mesh/render.py:284-311;
    g2's `face_vertices` are `verts` gathered by the winding-swapped faces, exactly what the module rasterizes)."""
    import torch.nn.functional as F
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRasterization
    g = golden("g2_mesh.npz")
    raw = torch.from_numpy(run_ref(ref, g["face_vertices"], 640, 640)).cuda()
    for S in (64, 128, 256):
        want = F.interpolate(raw.clamp(max=100.0).unsqueeze(1), size=(S, S), mode="bilinear", align_corners=False).squeeze(1)
        got = DepthRasterization(S, S, hand_model.load_mesh()["faces"]).cuda()(dev(g["verts"]))
        assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
        if S in (64, 128):
            assert torch.equal(got, want)             # weights 1 / 0.5: every product is exact


def test_contraction_changes_the_reference_kernels_output(ref, oracle, capsys):
    """What mul+add fusion does to THIS kernel (clang's pattern on gfx950): reported for DESIGN.md section 3."""
    hc = oracle.ref_tri_lib(contract=True)
    if hc is None:
        pytest.skip("libref_tri_contract.so not built")
    fv = golden("g2_mesh.npz")["face_vertices"]
    a, b = run_ref(ref, fv, 640, 640), run_ref(hc, fv, 640, 640)
    cov = (a < 1000) | (b < 1000)
    d = np.abs(a - b)[cov] / np.maximum(np.abs(a[cov]), 1e-6)
    with capsys.disabled():
        print("\n[reference kernel, contraction on vs as written] covered %d px; coverage differs at %d; bits differ at %d; "
              "relative difference: median %.2e, 99th pct %.2e, > 1e-4 at %d px, max %.3g"
              % (int(cov.sum()), int(((a < 1000) != (b < 1000)).sum()), int((bits(a) != bits(b)).sum()),
                 float(np.median(d)), float(np.percentile(d, 99)), int((d > 1e-4).sum()), float(d.max())))
    # ... and what is left of it where the product uses the image: clamp(max=100) + bilinear 640 -> 128 (mesh/render.py:310-311)
    import torch.nn.functional as F
    ra, rb = (F.interpolate(torch.from_numpy(x).clamp(max=100.0).unsqueeze(1), size=(128, 128), mode="bilinear",
                            align_corners=False).squeeze(1).numpy() for x in (a, b))
    dr = np.abs(ra - rb)
    with capsys.disabled():
        print("[the same two images after clamp(100) + resize to 128 x 128] max |diff| %.3g mm; > 1e-3 mm at %d of %d px; > 0.1 mm at %d"
              % (float(dr.max()), int((dr > 1e-3).sum()), dr.size, int((dr > 0.1).sum())))
    assert int(((a < 1000) != (b < 1000)).sum()) <= 16           # coverage is decided by comparisons the fusion barely touches


def test_reference_kernel_duration_beside_ours(ref, capsys):
    """The benchmark's triangle workload (256 crops x 3382 faces at 640 x 640): the reference's launch shape (one
    single-thread workgroup per face) on the MI355X beside tri_raster_kernel.  Reported, loosely asserted."""
    import depth_rasterization
    fv = np.tile(golden("g2_mesh.npz")["face_vertices"], (64, 1, 1, 1))
    v = dev(fv)
    out = torch.empty((256, 640, 640), dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert ref.ref_tri_forward(v.data_ptr(), 256, fv.shape[1], 640, 640, out.data_ptr()) == 0
        ts.append(time.perf_counter() - t0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    mine = depth_rasterization.forward(640, 640, v)
    e0.record()
    for _ in range(5):
        mine = depth_rasterization.forward(640, 640, v)
    e1.record(); e1.synchronize()
    ours = e0.elapsed_time(e1) / 5
    assert torch.equal(mine, out)
    with capsys.disabled():
        print("\n[256 crops x %d faces @640x640 on this GPU] reference kernel (fill + <<<B*F, 1>>>): %.2f ms; tri_raster_kernel: %.3f ms (%.0f x)"
              % (fv.shape[1], min(ts) * 1e3, ours, min(ts) * 1e3 / ours))
    assert ours < min(ts) * 1e3


def test_reference_kernel_randomised_slice(ref, oracle):
    """A bounded randomised run (the generator of tools/fuzz.py's `tri` family, fixed seed): every case three ways."""
    import depth_rasterization
    rs = np.random.RandomState(2024)
    t0, cases = time.time(), 0
    while cases < 400 and time.time() - t0 < 25.0:
        W = int(rs.choice([5, 16, 33, 96, 200, 320])); H = int(rs.choice([7, 16, 40, 72, 160, 256]))
        B = int(rs.randint(1, 4)); F = int(rs.choice([1, 31, 33, 64, 200, 500]))
        c = rs.uniform(-0.2 * W, 1.2 * W, (B, F, 1, 1)) * np.array([1.0, H / W])
        spread = rs.choice([0.7, 3.0, 12.0, 60.0], (B, F, 1, 1))
        fv = np.concatenate([c + rs.normal(0, 1, (B, F, 3, 2)) * spread, rs.uniform(-50, 50, (B, F, 3, 1))], -1).astype(np.float32)
        if rs.rand() < 0.3:
            fv[:, :, :, 0] = np.round(fv[:, :, :, 0])
        if rs.rand() < 0.2:
            fv[:, ::7, 1] = fv[:, ::7, 0]
        if rs.rand() < 0.1:
            fv[:, ::5, 2, 2] = 0.0                               # a vertex at z = 0
        r = run_ref(ref, fv, W, H)
        assert np.array_equal(bits(r), bits(oracle.tri_raster_fwd(fv, W, H))), (cases, W, H, B, F)
        assert np.array_equal(bits(r), bits(depth_rasterization.forward(W, H, dev(fv)).cpu().numpy())), (cases, W, H, B, F)
        cases += 1
    assert cases >= 50
