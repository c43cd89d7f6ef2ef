"""CPU: the triangle-raster oracle (C) against an INDEPENDENT numpy restatement of
the reference kernel, and the oracle's skinning / camera / resize against the
reference's goldens.  These tests make the two restatements agree bit for bit on
the CPU; the pin to the reference's own kernel is tests/test_tri_reference_gpu.py
(oracle/_ref: its device code compiled for gfx950, run on the MI355X)."""
import numpy as np

from conftest import bits, golden

F32 = np.float32


def _cvt(d):
    """CUDA cvt.rzi.s32.f64: truncate toward zero, NaN -> 0, saturate."""
    d = np.asarray(d, np.float64)
    return np.where(np.isnan(d), 0, np.clip(np.trunc(d), -2147483648.0, 2147483647.0)).astype(np.int64)


def numpy_tri_raster(face_vertices, W, H):
    """Vectorised-per-face numpy version of depth_rasterization_cuda_kernel.cu:18-113
    (written from the kernel text, independently of oracle/spherehand_oracle.c)."""
    B, F = face_vertices.shape[:2]
    out = np.full((B, H, W), 1000.0, F32)
    for b in range(B):
        for fn in range(F):
            f = face_vertices[b, fn].reshape(9).astype(F32)
            if F32(F32(f[7] - f[1]) * F32(f[3] - f[0])) < F32(F32(f[4] - f[1]) * F32(f[6] - f[0])):
                continue
            if f[0] < f[3]:
                i0 = 2 if f[6] < f[0] else 0
                i2 = 2 if f[3] < f[6] else 1
            else:
                i0 = 2 if f[6] < f[3] else 1
                i2 = 2 if f[0] < f[6] else 0
            i1 = [k for k in range(3) if k != i0 and k != i2][-1] if i0 != i2 else [k for k in range(3) if k != i0][-1]
            p = np.stack([f[3 * i0:3 * i0 + 3], f[3 * i1:3 * i1 + 3], f[3 * i2:3 * i2 + 3]]).astype(F32)
            if p[0, 0] == p[2, 0]:
                continue
            with np.errstate(all="ignore"):
                fi = np.array([p[1, 1] - p[2, 1], p[2, 0] - p[1, 0], F32(p[1, 0] * p[2, 1]) - F32(p[2, 0] * p[1, 1]),
                               p[2, 1] - p[0, 1], p[0, 0] - p[2, 0], F32(p[2, 0] * p[0, 1]) - F32(p[0, 0] * p[2, 1]),
                               p[0, 1] - p[1, 1], p[1, 0] - p[0, 0], F32(p[0, 0] * p[1, 1]) - F32(p[1, 0] * p[0, 1])], F32)
                den = F32(F32(F32(p[2, 0] * F32(p[0, 1] - p[1, 1])) + F32(p[0, 0] * F32(p[1, 1] - p[2, 1])))
                          + F32(p[1, 0] * F32(p[2, 1] - p[0, 1])))
                fi = (fi / den).astype(F32)
                xi_min = int(_cvt(np.fmax(np.ceil(p[0, 0]), 0.0)))          # CUDA max/min = fmax/fmin
                xi_max = int(_cvt(np.fmin(float(p[2, 0]), W - 1.0)))
                if xi_max < xi_min:
                    continue
                xs = np.arange(xi_min, xi_max + 1)
                xf = xs.astype(F32)
                left = xf <= p[1, 0]
                d01, d12 = F32(p[1, 0] - p[0, 0]), F32(p[2, 0] - p[1, 0])
                ya = (F32(F32(p[1, 1] - p[0, 1]) / d01) * (xf - p[0, 0]).astype(F32)).astype(F32) + p[0, 1] \
                    if d01 != 0 else np.full(len(xs), p[1, 1], F32)
                yb = (F32(F32(p[2, 1] - p[1, 1]) / d12) * (xf - p[1, 0]).astype(F32)).astype(F32) + p[1, 1] \
                    if d12 != 0 else np.full(len(xs), p[1, 1], F32)
                yi1 = np.where(left, ya, yb).astype(F32)
                yi2 = ((F32(F32(p[2, 1] - p[0, 1]) / F32(p[2, 0] - p[0, 0])) * (xf - p[0, 0]).astype(F32)).astype(F32)
                       + p[0, 1]).astype(F32)
                ymin = _cvt(np.fmax(0.0, np.ceil(np.fmin(yi1, yi2).astype(np.float64))))
                ymax = _cvt(np.fmin(np.fmax(yi1, yi2).astype(np.float64), H - 1.0))
                for c, xi in enumerate(xs):
                    if not (ymin[c] <= ymax[c]):
                        continue
                    ys = np.arange(int(ymin[c]), int(ymax[c]) + 1)
                    yf = ys.astype(F32)
                    w = np.stack([((fi[3 * k] * xf[c]).astype(F32) + (fi[3 * k + 1] * yf).astype(F32)).astype(F32)
                                  + fi[3 * k + 2] for k in range(3)]).astype(F32)
                    w = np.fmin(np.fmax(w, F32(0)), F32(1))
                    ws = ((F32(0) + w[0]).astype(F32) + w[1]).astype(F32) + w[2]
                    w = (w / ws).astype(F32)
                    s = (((w[0] / p[0, 2]).astype(F32) + (w[1] / p[1, 2]).astype(F32)).astype(F32)
                         + (w[2] / p[2, 2]).astype(F32)).astype(F32)
                    zp = (1.0 / s.astype(np.float64)).astype(F32)
                    col = out[b, ys, xi]
                    out[b, ys, xi] = np.where(np.isnan(zp), col, np.minimum(zp, col))
    return out


def test_oracle_tri_vs_independent_numpy(oracle):
    g = golden("g2_mesh.npz")
    # pose 1 (random), every 3rd face, scaled into a 160x160 image: small enough for numpy loops
    fv = g["face_vertices"][1, ::3].copy()
    fv[:, :, :2] *= 0.25
    fv = fv[None]
    a = oracle.tri_raster_fwd(fv, 160, 160)
    b = numpy_tri_raster(fv, 160, 160)
    assert (a < 1000).sum() > 1000
    assert np.array_equal(bits(a), bits(b))


def test_oracle_tri_quirks(oracle):
    """Truncation-toward-zero spans (.cu:69,90), degenerate and NaN faces, z = 0."""
    tri = np.array([
        [[-0.5, -0.7, 5], [-0.2, 3.0, 5], [-0.1, -0.6, 5]],      # entirely left of column 0: xi_max = trunc(-0.1) = 0
        [[2, 2, 0], [2, 9, 4], [9, 2, 4]],                          # a vertex at z = 0: w/0
        [[5, 5, 3], [5, 9, 3], [5, 7, 3]],                          # vertical line: skipped (.cu:54)
        [[1, 1, 3], [4, 4, 3], [7, 7, 3]],                          # collinear: denominator 0 -> NaN, no write
        [[np.nan, 1, 3], [4, 2, 3], [7, 9, 3]],
        [[3, 12, 2], [12, 3, 2], [3, 3, -2]],                       # straddles z = 0
    ], np.float32)[None]
    for order in (slice(None), slice(None, None, -1)):
        fv = tri[:, order]
        a = oracle.tri_raster_fwd(fv, 16, 16)
        b = numpy_tri_raster(fv, 16, 16)
        assert np.array_equal(bits(a), bits(b))
    # winding flipped: what was culled now draws and vice versa -> both restatements still agree
    flipped = tri[:, :, [1, 0, 2], :]
    assert np.array_equal(bits(oracle.tri_raster_fwd(flipped, 16, 16)), bits(numpy_tri_raster(flipped, 16, 16)))


def test_oracle_lbs_project_vs_reference(oracle):
    """Reference LinearBlendSkinning + OthographicalProjection (dense torch) vs the
    oracle's sparse restatement: association of <= 5 non-zero terms differs ->
    tolerance 5e-5 on values up to 640."""
    from spherehand_amd import hand_model
    g = golden("g2_mesh.npz")
    mesh = hand_model.load_mesh()
    start, bone, wv = hand_model.sparse_skin(mesh)
    cam = (320.0, 320.0, 640 / 300, 640 / 300)
    assert np.abs(oracle.lbs_project(g["T"], start, bone, wv, True, None) - g["skinned"]).max() <= 5e-5
    assert np.abs(oracle.lbs_project(g["T"], start, bone, wv, True, cam) - g["verts"]).max() <= 1e-4
    assert np.abs(oracle.lbs_project(g["T"], start, bone, wv, True, cam, g["rand_f"]) - g["verts_rand_f"]).max() <= 1e-4
    faces = np.asarray(mesh["faces"]).copy()
    faces[:, [0, 1]] = faces[:, [1, 0]]
    assert np.array_equal(faces.astype(np.int32), g["faces_swapped"])


def test_oracle_clamp_bilinear_vs_reference(oracle):
    """F.interpolate(bilinear, align_corners=False) after clamp(max=100): the
    reference's torch result on the oracle's 640x640 raster."""
    g = golden("g2_mesh.npz")
    raw = oracle.tri_raster_fwd(g["face_vertices"], 640, 640)
    assert np.array_equal(bits(raw[0]), bits(g["raw640_first"]))
    assert np.array_equal((raw < 1000).reshape(4, -1).sum(1), g["raw640_covered"])
    for S in (64, 128, 256):
        d = oracle.clamp_bilinear(raw, S, S, 100.0)
        ref = g["depth%d" % S]
        assert np.abs(d - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), S
    assert int((g["depth64"][0] < 100).sum()) == 956       # SURVEY appendix A (host-compiled reference kernel)


def test_rest_pose_statistics_of_the_reference_kernel(oracle):
    """What SURVEY 2.3 / App. A measured on the reference kernel body at the rest pose: 1794 of 3382 faces
    culled (.cu:33) and 956 foreground pixels in the 64x64 DepthRender.  (SURVEY also quotes a raw minimum of
    -74 297; that value is the 1/z interpolation next to a zero of w0/z0+w1/z1+w2/z2 on a face straddling z = 0
    (.cu:109) and moves by a factor with the last bit of a vertex -- this oracle gives -7230.78 on the
    reference's own fp32 vertices, the FMA-contracted build -6922.72; it is recorded, not asserted equal.)"""
    g = golden("g2_mesh.npz")
    assert np.abs(g["params"][0]).max() == 0.0
    f = g["face_vertices"][0].reshape(-1, 9)
    culled = (f[:, 7] - f[:, 1]) * (f[:, 3] - f[:, 0]) < (f[:, 4] - f[:, 1]) * (f[:, 6] - f[:, 0])
    assert int(culled.sum()) == 1794 and len(f) == 3382
    assert int((g["depth64"][0] < 100).sum()) == 956
    raw = oracle.tri_raster_fwd(g["face_vertices"][:1], 640, 640)
    assert raw.min() < -1000.0 and raw.min() == g["raw640_min"][0]


def test_tri_fma_sensitivity(oracle):
    """Parity of this kernel is UNPINNED (no nvcc here); this quantifies what that can cost.  nvcc contracts
    mul+add into FMA by default; which products it fuses is not recoverable, so the restatement is compiled a
    second time with contraction allowed everywhere (-ffp-contract=fast -mfma) and compared with the oracle of
    record (-ffp-contract=off) on g2's four poses.  Measured (this test prints the table):
      * 640x640 raster: 12-21 % of the pixels differ in their last bits, NO pixel changes coverage, and
        1.3-3.1 % of the covered pixels move by more than 1e-4 relative -- all of them 1/z interpolations on
        faces that straddle z = 0 (.cu:109), where the value is the reciprocal of a near-cancelling sum;
      * after clamp(max=100) + bilinear resize: at most 5 / 12 / 68 of the 64^2 / 128^2 / 256^2 output pixels
        differ by more than 1e-4 of the image's maximum, every other pixel agrees to <= 1e-4.
    So "<= 1e-4 relative L-inf against the CUDA rasterizer" is attainable everywhere except at a handful of
    near-singular pixels per image whose value the reference itself does not determine beyond its compiler."""
    g = golden("g2_mesh.npz")
    fv = g["face_vertices"]
    a = oracle.tri_raster_fwd(fv, 640, 640)
    b = oracle.tri_raster_fwd(fv, 640, 640, oracle.fma_variant())
    assert np.array_equal(a < 1000, b < 1000)                       # coverage never flips
    for i in range(4):
        cov = a[i] < 1000
        rel = np.abs(a[i] - b[i])[cov] / np.maximum(np.abs(a[i][cov]), 1e-30)
        frac_bits = float((a[i] != b[i]).mean())
        frac_far = float((rel > 1e-4).mean())
        print("pose %d: 640^2 px differing %.1f %%, covered px beyond 1e-4 rel: %.2f %% (max rel %.3g)"
              % (i, 100 * frac_bits, 100 * frac_far, rel.max()))
        assert 0.05 < frac_bits < 0.30 and frac_far < 0.05
    for S, cap in ((64, 8), (128, 16), (256, 96)):
        da, db = oracle.clamp_bilinear(a, S, S, 100.0), oracle.clamp_bilinear(b, S, S, 100.0)
        for i in range(4):
            diff = np.abs(da[i] - db[i])
            far = int((diff > 1e-4 * np.abs(da[i]).max()).sum())
            print("S=%d pose %d: %d px beyond 1e-4 of max, L-inf %.3g" % (S, i, far, diff.max() / np.abs(da[i]).max()))
            assert far <= cap


def test_oracle_fk_vs_reference(oracle):
    from spherehand_amd import hand_model
    g = golden("g3_batch256.npz")
    off, inv = hand_model.offset_matrices(hand_model.load_mesh())
    T = oracle.fk_fwd(g["params"], off, inv)
    assert np.abs(T - g["T"]).max() <= 2e-4 and np.abs(g["T"]).max() > 100
    g1 = golden("g1_rest_pose.npz")
    assert np.abs(oracle.fk_fwd(g1["params"], off, inv) - g1["T"]).max() <= 5e-5
