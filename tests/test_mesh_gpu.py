"""GPU parity of the triangle-mesh path (through the C ABI): triangle rasterizer
bit-exact vs the CPU oracle; skinning / camera bit-exact vs the oracle and within
tolerance of the reference's dense torch result; DepthRender end to end vs the
reference pipeline (tests/golden/g2_mesh.npz, see its generator for what is and
is not the reference's code)."""
import numpy as np
import pytest

from conftest import bits, golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_tri_raster_full_size_vs_oracle(oracle):
    """4 poses x 3382 faces at the reference's 640x640 working resolution."""
    import depth_rasterization
    g = golden("g2_mesh.npz")
    fv = g["face_vertices"]
    d = depth_rasterization.forward(640, 640, dev(fv))
    assert d.shape == (4, 640, 640) and d.dtype == torch.float32 and d.is_cuda
    o = oracle.tri_raster_fwd(fv, 640, 640)
    assert np.array_equal(bits(d.cpu().numpy()), bits(o))
    assert np.array_equal(bits(d.cpu().numpy()[0]), bits(g["raw640_first"]))
    assert d.max().item() == 1000.0
    # order independence: faces reversed -> identical image
    d2 = depth_rasterization.forward(640, 640, dev(fv[:, ::-1]))
    assert torch.equal(d, d2)
    # width != height: output is [B, height, width]
    fv2 = fv[:1].copy()
    d3 = depth_rasterization.forward(320, 640, dev(fv2))
    assert d3.shape == (1, 640, 320)
    assert np.array_equal(bits(d3.cpu().numpy()), bits(oracle.tri_raster_fwd(fv2, 320, 640)))


def test_tri_raster_quirks_and_edge_cases(oracle):
    import depth_rasterization
    tri = np.array([
        [[-0.5, -0.7, 5], [-0.2, 3.0, 5], [-0.1, -0.6, 5]],
        [[2, 2, 0], [2, 9, 4], [9, 2, 4]],
        [[5, 5, 3], [5, 9, 3], [5, 7, 3]],
        [[1, 1, 3], [4, 4, 3], [7, 7, 3]],
        [[np.nan, 1, 3], [4, 2, 3], [7, 9, 3]],
        [[3, 12, 2], [12, 3, 2], [3, 3, -2]],
        [[-40, -30, 7], [60, -20, 7], [10, 70, 7]],            # larger than the image
        [[-0.7, 7.1, 5], [-3.2, 14.3, 7], [-9.4, 7.6, 6]],     # left of the image, x2 in (-1, 0): column 0 by truncation, rows EXTRApolated
    ], np.float32)[None]
    for fv in (tri, tri[:, ::-1], tri[:, :, [1, 0, 2], :]):
        for (W, H) in ((16, 16), (17, 9), (5, 33)):
            d = depth_rasterization.forward(W, H, dev(fv)).cpu().numpy()
            assert np.array_equal(bits(d), bits(oracle.tri_raster_fwd(fv, W, H))), (W, H)
    # no faces: all background; empty batch
    e = depth_rasterization.forward(8, 8, torch.empty(2, 0, 3, 3, device="cuda"))
    assert e.shape == (2, 8, 8) and torch.all(e == 1000.0)
    assert depth_rasterization.forward(8, 8, torch.empty(0, 5, 3, 3, device="cuda")).shape == (0, 8, 8)
    with pytest.raises(RuntimeError):
        depth_rasterization.forward(8, 8, torch.zeros(1, 4, 3, 3))                     # not a CUDA tensor
    with pytest.raises(RuntimeError):
        depth_rasterization.forward(8, 8, torch.zeros(1, 3, 4, 3, device="cuda").transpose(1, 2))  # not contiguous


def test_random_faces_vs_oracle(oracle):
    import depth_rasterization
    rs = np.random.RandomState(4)
    B, F, W, H = 3, 500, 96, 72
    c = rs.uniform(-10, [W + 10, H + 10], (B, F, 1, 2))
    fv = np.concatenate([c + rs.normal(0, 6, (B, F, 3, 2)), rs.uniform(-30, 60, (B, F, 3, 1))], -1).astype(np.float32)
    d = depth_rasterization.forward(W, H, dev(fv)).cpu().numpy()
    assert np.array_equal(bits(d), bits(oracle.tri_raster_fwd(fv, W, H)))


def test_random_large_faces_vs_oracle(oracle):
    """Boxes above 256 pixels (visited alone, 8x8 patches) mixed with small ones (16-pixel groups), depths of
    both signs (signed-min / unsigned-max atomics on the float bits), F not a multiple of the faces per wave."""
    import depth_rasterization
    rs = np.random.RandomState(11)
    B, F, W, H = 2, 333, 200, 160
    c = rs.uniform(-20, [W + 20, H + 20], (B, F, 1, 2))
    spread = rs.choice([3.0, 12.0, 40.0], (B, F, 1, 1))
    fv = np.concatenate([c + rs.normal(0, 1, (B, F, 3, 2)) * spread, rs.uniform(-50, 50, (B, F, 3, 1))], -1).astype(np.float32)
    d = depth_rasterization.forward(W, H, dev(fv)).cpu().numpy()
    assert np.array_equal(bits(d), bits(oracle.tri_raster_fwd(fv, W, H)))


@pytest.mark.parametrize("band", [-1, 0, 8, 13, 4096])
def test_tri_raster_band_kernel_and_atomic_kernel_agree_with_the_oracle(oracle, band):
    """shr_tri_raster_fwd's two kernels -- bands of rows held in LDS (the default for up to 4 crops; `band` rows at
    most when a test says so: faces then straddle many bands, the last band is ragged; 4096 = as many as fit) and the
    global-atomic kernel (band = 0; the default for more crops) --
    against the oracle, bit for bit: random soups with small and large faces and depths of both signs, widths that are
    not a multiple of 4 (scalar stream-out), a single crop (the bands are dealt to several workgroups), a single face,
    the quirk faces, the indexed entry, and the hand mesh at 640x640."""
    import depth_rasterization
    from spherehand_amd import ops
    ops.set_tuning(ops.TUNE_TRI_BAND, band)
    try:
        rs = np.random.RandomState(21)
        for B, F, W, H in ((2, 333, 200, 160), (1, 900, 96, 72), (3, 64, 37, 53), (1, 1, 64, 64), (5, 33, 130, 9)):
            c = rs.uniform(-20, [W + 20, H + 20], (B, F, 1, 2))
            spread = rs.choice([3.0, 12.0, 40.0], (B, F, 1, 1))
            fv = np.concatenate([c + rs.normal(0, 1, (B, F, 3, 2)) * spread, rs.uniform(-50, 50, (B, F, 3, 1))], -1).astype(np.float32)
            d = depth_rasterization.forward(W, H, dev(fv)).cpu().numpy()
            assert np.array_equal(bits(d), bits(oracle.tri_raster_fwd(fv, W, H))), (B, F, W, H)
            # the indexed entry (vertices + faces) on the same triangles
            verts = np.concatenate([fv.reshape(B, F * 3, 3), np.ones((B, F * 3, 1), np.float32)], -1)
            faces = np.arange(F * 3, dtype=np.int32).reshape(F, 3)
            di = ops.tri_raster_indexed_fwd(W, H, dev(verts), dev(faces)).cpu().numpy()
            assert np.array_equal(bits(di), bits(d)), (B, F, W, H)
        tri = np.array([
            [[-0.5, -0.7, 5], [-0.2, 3.0, 5], [-0.1, -0.6, 5]], [[2, 2, 0], [2, 9, 4], [9, 2, 4]], [[5, 5, 3], [5, 9, 3], [5, 7, 3]],
            [[1, 1, 3], [4, 4, 3], [7, 7, 3]], [[np.nan, 1, 3], [4, 2, 3], [7, 9, 3]], [[3, 12, 2], [12, 3, 2], [3, 3, -2]],
            [[-40, -30, 7], [60, -20, 7], [10, 70, 7]], [[-0.7, 7.1, 5], [-3.2, 14.3, 7], [-9.4, 7.6, 6]],
            [[1e9, 3, 2], [2, 1e9, 2], [3, 3, 2]], [[2, -1e9, 2], [9, 1e9, 2], [4, 3, 2]],
        ], np.float32)[None]
        for fv in (tri, tri[:, ::-1], tri[:, :, [1, 0, 2], :]):
            for (W, H) in ((16, 16), (17, 9), (5, 33), (8, 40)):
                d = depth_rasterization.forward(W, H, dev(fv)).cpu().numpy()
                assert np.array_equal(bits(d), bits(oracle.tri_raster_fwd(fv, W, H))), (W, H)
        e = depth_rasterization.forward(8, 8, torch.empty(2, 0, 3, 3, device="cuda"))
        assert e.shape == (2, 8, 8) and torch.all(e == 1000.0)
        g = golden("g2_mesh.npz")
        d = depth_rasterization.forward(640, 640, dev(g["face_vertices"][:2])).cpu().numpy()
        assert np.array_equal(bits(d[0]), bits(g["raw640_first"]))
        assert np.array_equal(bits(d), bits(oracle.tri_raster_fwd(g["face_vertices"][:2], 640, 640)))
    finally:
        ops.set_tuning(ops.TUNE_TRI_BAND, -1)


def test_lbs_project(oracle):
    from spherehand_amd import hand_model, ops
    g = golden("g2_mesh.npz")
    start, bone, wv = hand_model.sparse_skin(hand_model.load_mesh())
    cam = (320.0, 320.0, 640 / 300, 640 / 300)
    args = (dev(g["T"]), dev(start), dev(bone), dev(wv))
    for camera, rand_f, ref in ((None, None, g["skinned"]), (cam, None, g["verts"]), (cam, g["rand_f"], g["verts_rand_f"])):
        out = ops.lbs_project(*args, True, camera, None if rand_f is None else dev(rand_f)).cpu().numpy()
        o = oracle.lbs_project(g["T"], start, bone, wv, True, camera, rand_f)
        assert np.array_equal(bits(out), bits(o))
        assert np.abs(out - ref).max() <= 1e-4          # reference's dense bmm + sum: different association


def test_depth_render_end_to_end():
    """T -> depth [B,S,S]: skinning, camera, raster at 640, clamp, bilinear resize.
    The vertices differ from the reference's by <= 1e-4 px (association), so a few
    silhouette pixels of the 640x640 raster may flip: bar = < 0.1 % of pixels off
    by more than 1e-3 mm."""
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRender
    g = golden("g2_mesh.npz")
    mesh = hand_model.load_mesh()
    faces_before = mesh["faces"].copy()
    for S in (64, 128, 256):
        render = DepthRender(mesh, S).cuda()
        d = render(dev(g["T"])).cpu().numpy()
        ref = g["depth%d" % S]
        assert d.shape == ref.shape
        off = np.abs(d - ref) > 1e-3
        assert off.mean() < 1e-3, (S, off.mean())
    assert np.array_equal(mesh["faces"], faces_before)        # the caller's array is not mutated
    d = DepthRender(mesh, 64).cuda()(dev(g["T"]), dev(g["rand_f"])).cpu().numpy()
    assert (np.abs(d - g["depth64_rand_f"]) > 1e-3).mean() < 1e-3


def test_depth_rasterization_module_paths_agree():
    """Fused indexed raster == explicit gather + depth_rasterization.forward."""
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRasterization, DepthRasterizationFunction
    g = golden("g2_mesh.npz")
    mesh = hand_model.load_mesh()
    r = DepthRasterization(128, 128, mesh["faces"]).cuda()
    verts = dev(g["verts"])
    a = r(verts)
    fv = verts[:, r.faces, 0:3].view(4, r.num_faces, 3, 3)
    assert np.abs(fv.cpu().numpy() - g["face_vertices"]).max() <= 1e-4
    b = torch.nn.functional.interpolate(DepthRasterizationFunction.apply(640, 640, fv).unsqueeze(1), size=(128, 128),
                                        mode="bilinear", align_corners=False).squeeze(1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("S", [64, 128, 256, 96, 40])
def test_fused_depth_render_equals_the_explicit_chain(S):
    """Fused raster+clamp+resize vs the explicit 640x640 raster -> torch clamp -> F.interpolate.
    The sampled raster values are the same numbers; only the 4-term bilinear sum may contract
    differently inside torch's kernel (weights 0.25/0.75 at S=256): 1e-5 relative."""
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRasterization
    g = golden("g2_mesh.npz")
    r = DepthRasterization(S, S, hand_model.load_mesh()["faces"]).cuda()
    verts = dev(g["verts"])
    fused = r(verts)
    r.fused = False
    chain = r(verts)
    assert fused.shape == (4, S, S)
    assert (fused - chain).abs().max().item() <= 1e-5 * max(1.0, chain.abs().max().item())
    if S in (64, 128):
        assert torch.equal(fused, chain)              # weights 1 / 0.5: every product is exact
    assert (fused < 100).float().mean().item() > 0.02


@pytest.mark.parametrize("S", [256, 320, 400, 500, 640, 230])
@pytest.mark.parametrize("B", [1, 5, 70, 300])
def test_band_kernel_with_resize_epilogue_equals_the_tile_kernel(S, B):
    """Sizes without a lattice kernel whose resize samples at least half of the source pixels (S = 256 from 640 ...):
    the triangle band kernel at full resolution with clamp + resize as its stream-out (tri_raster.hip RESIZE) against the
    tile kernel (SHR_TUNE_MESH_BAND 0) -- the same rasterized values through the same bilinear formula: bit for bit, at
    every launch plan (one crop: many workgroups per crop; 300 crops: more crops than CUs)."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.render import DepthRender
    mesh = hand_model.load_mesh()
    dr = DepthRender(mesh, S).cuda()
    fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).cuda()
    with torch.no_grad():
        verts = dr.lbs(fk(sample_poses(B, seed=S).cuda()), dr.camera, None).contiguous()
    faces = dr.rasterizer.faces_i32
    try:
        ops.set_tuning(ops.TUNE_MESH_BAND, 0)
        tile = ops.mesh_depth_fwd(verts, faces, S, 640, 100.0)
    finally:
        ops.set_tuning(ops.TUNE_MESH_BAND, 1)
    band = ops.mesh_depth_fwd(verts, faces, S, 640, 100.0)
    assert torch.equal(band, tile)
    assert 0.02 < (band < 100).float().mean().item() < 0.6


@pytest.mark.parametrize("S", [400, 500, 640])
def test_tile_kernel_at_small_ratios(S):
    """shr_mesh_depth_fwd accepts any S <= 640: ratios below 2 (every source pixel sampled, bilinear weights on
    both neighbours) against the explicit raster -> clamp -> F.interpolate chain; S = 640 is the raster itself."""
    from spherehand_amd import hand_model, ops
    g = golden("g2_mesh.npz")
    faces = torch.from_numpy(hand_model.load_mesh()["faces"].astype(np.int32)).cuda()[:, [0, 2, 1]].contiguous()
    verts = dev(g["verts"])[:2].contiguous()
    tile = ops.mesh_depth_fwd(verts, faces, S, 640, 100.0)
    raw = ops.tri_raster_indexed_fwd(640, 640, verts, faces)
    chain = torch.nn.functional.interpolate(torch.clamp(raw, max=100.0).unsqueeze(1), size=(S, S), mode="bilinear",
                                            align_corners=False).squeeze(1)
    assert (tile - chain).abs().max().item() <= 1e-5 * max(1.0, chain.abs().max().item())
    if S == 640:
        assert torch.equal(tile, torch.clamp(raw, max=100.0))
    assert (tile < 100).float().mean().item() > 0.02


@pytest.mark.parametrize("src,S", [(257, 200), (320, 96), (96, 64)])
def test_tile_kernel_non_dyadic_source_index(src, S):
    """Non-dyadic resize ratios: the source index scale * (d + 0.5) - 0.5 is one FMA in ATen's GPU kernel; formed
    with two roundings it is off by an ulp (3e-5 px at 256) and the bilinear weights with it (tools/fuzz.py, mesh)."""
    from spherehand_amd import hand_model, ops
    g = golden("g2_mesh.npz")
    faces = torch.from_numpy(hand_model.load_mesh()["faces"].astype(np.int32)).cuda()[:, [0, 2, 1]].contiguous()
    verts = dev(g["verts"])[:2].clone()
    verts[:, :, 0:2] *= src / 640.0
    verts = verts.contiguous()
    tile = ops.mesh_depth_fwd(verts, faces, S, src, 100.0)
    raw = ops.tri_raster_indexed_fwd(src, src, verts, faces)
    chain = torch.nn.functional.interpolate(torch.clamp(raw, max=100.0).unsqueeze(1), size=(S, S), mode="bilinear",
                                            align_corners=False).squeeze(1)
    assert (tile - chain).abs().max().item() <= 1e-5 * max(1.0, chain.abs().max().item())
    assert (tile < 100).float().mean().item() > 0.02


def test_fused_depth_render_rand_f_and_batch():
    from spherehand_amd import hand_model
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.render import DepthRender
    mesh = hand_model.load_mesh()
    fk = HandTransformationMat([b["offset_matrix"].astype(np.float32) for b in mesh["bones"]]).cuda()
    T = fk(sample_poses(96, seed=9).cuda())
    rf = torch.rand(96, device="cuda") * 0.2 + 0.9
    for S in (64, 128):
        render = DepthRender(mesh, S).cuda()
        a = render(T, rf)
        render.rasterizer.fused = False
        b = render(T, rf)
        assert torch.equal(a, b)


@pytest.mark.parametrize("S", [64, 128, 256, 200])
def test_fused_depth_render_large_and_odd_faces(S):
    """Synthetic soup the hand mesh never produces: faces wider than 32 sampled columns, faces
    covering the whole image, slivers, faces partly or wholly outside, > 4096 faces (several
    rounds), > 3584 work items per round -- against the explicit 640x640 chain."""
    from spherehand_amd import ops
    rs = np.random.RandomState(S)
    B, NV, F = 3, 6000, 9000
    v = np.zeros((B, NV, 4), np.float32)
    v[..., 0:2] = rs.uniform(-80, 720, (B, NV, 2))
    v[..., 2] = rs.uniform(-60, 140, (B, NV))
    v[..., 3] = 1
    v[:, 1::3, 0:2] = v[:, 0::3, 0:2] + rs.uniform(-14, 14, (B, NV // 3, 2))     # vertex triples: small faces
    v[:, 2::3, 0:2] = v[:, 0::3, 0:2] + rs.uniform(-14, 14, (B, NV // 3, 2))
    t = rs.randint(0, NV // 3, F)
    faces = np.stack([3 * t, 3 * t + 1, 3 * t + 2], 1).astype(np.int32)
    big = rs.rand(F) < 0.03                                                       # arbitrary vertices: huge faces
    faces[big] = rs.randint(0, NV, (int(big.sum()), 3))
    faces[0] = (0, 1, 2); v[:, 0, 0:3] = (-50, -50, 90); v[:, 1, 0:3] = (700, -40, 95); v[:, 2, 0:3] = (300, 720, 99)   # whole image
    faces[1] = (2, 1, 0)                                                                                               # its back face
    faces[2] = (3, 4, 5); v[:, 3, 0:3] = (10, 300, 50); v[:, 4, 0:3] = (630, 300.4, 50); v[:, 5, 0:3] = (320, 301, 50)   # sliver
    vd, fd = dev(v), dev(faces)
    fused = ops.mesh_depth_fwd(vd, fd, S, 640, 100.0)
    raw = ops.tri_raster_indexed_fwd(640, 640, vd, fd)
    chain = torch.nn.functional.interpolate(torch.clamp(raw, max=100.0).unsqueeze(1), size=(S, S), mode="bilinear",
                                            align_corners=False).squeeze(1)
    assert (fused < 100).float().mean().item() > 0.5
    assert (fused - chain).abs().max().item() <= 1e-5 * max(1.0, chain.abs().max().item())
    if S in (64, 128):
        assert torch.equal(fused, chain)


@pytest.mark.parametrize("src,S", [(128, 128), (64, 64), (640, 128)])
def test_fused_depth_render_dense_front_facing_grid(src, S):
    """A height field of 64 x 64 quads (8192 triangles, all front-facing) over the whole image: more than 2048 of a
    round's 4096 faces survive the culls in ONE tile (src == S: every source pixel is sampled), then a second round.
    The round-3 kernel packed the survivor count into the upper 12 bits of a signed work-item scan."""
    from spherehand_amd import ops
    n = 64
    rs = np.random.RandomState(5)
    gx, gy = np.meshgrid(np.linspace(1.0, src - 2.0, n + 1), np.linspace(1.0, src - 2.0, n + 1))
    gx = gx + rs.uniform(-0.2, 0.2, gx.shape)
    gy = gy + rs.uniform(-0.2, 0.2, gy.shape)
    v = np.zeros((2, (n + 1) * (n + 1), 4), np.float32)
    v[..., 0] = gx.reshape(-1); v[..., 1] = gy.reshape(-1); v[..., 3] = 1
    v[0, :, 2] = (30 + 20 * np.sin(gx / 9.0) * np.cos(gy / 7.0)).reshape(-1)
    v[1, :, 2] = rs.uniform(5, 90, (n + 1) * (n + 1))
    idx = lambda r, c: r * (n + 1) + c
    faces = []
    for r in range(n):
        for c in range(n):
            a, b, d, e = idx(r, c), idx(r, c + 1), idx(r + 1, c), idx(r + 1, c + 1)
            faces += [(a, d, b), (b, d, e)]
    faces = np.asarray(faces, np.int32)
    vd, fd = dev(v), dev(faces)
    raw = ops.tri_raster_indexed_fwd(src, src, vd, fd)
    if (raw < 1000).float().mean().item() < 0.5:          # the other winding is the front-facing one
        faces = faces[:, [0, 2, 1]].copy()
        fd = dev(faces)
        raw = ops.tri_raster_indexed_fwd(src, src, vd, fd)
    assert (raw < 1000).float().mean().item() > 0.9
    fused = ops.mesh_depth_fwd(vd, fd, S, src, 100.0)
    chain = torch.nn.functional.interpolate(torch.clamp(raw, max=100.0).unsqueeze(1), size=(S, S), mode="bilinear",
                                            align_corners=False).squeeze(1)
    assert torch.equal(fused, chain)


def test_depth_render_on_the_distinct_vertices_equals_the_full_mesh():
    """DepthRender skins the mesh's 1 721 distinct vertices (the reference stores each face's corners separately:
    10 144 records with byte-identical skin entries per copy) and lets the faces index them: the skinned vertices of
    the copies are bit-identical to the full table's, and so is the image."""
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRasterization, DepthRender, SparseSkinning
    g = golden("g2_mesh.npz")
    mesh = hand_model.load_mesh()
    T, rf = dev(g["T"]), dev(g["rand_f"])
    full = SparseSkinning(mesh).cuda()
    for S in (64, 128, 256):
        dr = DepthRender(mesh, S).cuda()
        assert dr.lbs.num_vertices == 1721 and full.num_vertices == 10144
        for rand_f in (None, rf):
            vu = dr.lbs(T, dr.camera, rand_f)
            va = full(T, dr.camera, rand_f)
            assert torch.equal(vu[:, torch.from_numpy(dr.lbs.vertex_index).cuda()], va)
            want = DepthRasterization(S, S, mesh["faces"]).cuda()(va)
            assert torch.equal(dr(T, rand_f), want)


@pytest.mark.parametrize("S", [128, 64, 32, 256, 200])
def test_mesh_render_one_launch_equals_skinning_then_raster(S):
    """shr_mesh_render_fwd (DepthRender.forward as one call: the lattice kernel skins its crop's vertices into LDS where
    it applies -- S = 128 / 64 / 32 from 640 --, two launches through the workspace otherwise) against shr_lbs_project
    followed by shr_mesh_depth_fwd: the same bits, with and without the random focal factor."""
    from spherehand_amd import hand_model, ops
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.render import DepthRender
    mesh = hand_model.load_mesh()
    fk = HandTransformationMat([b["offset_matrix"].astype(np.float32) for b in mesh["bones"]]).cuda()
    dr = DepthRender(mesh, S).cuda()
    for B in (1, 37):
        T = fk(sample_poses(B, seed=11 + B).cuda()).contiguous()
        rf = torch.rand(B, device="cuda") * 0.3 + 0.85
        for rand_f in (None, rf):
            verts = ops.lbs_project(T, dr.lbs.skin_vertex_start, dr.lbs.skin_bone, dr.lbs.skin_wv, dr.lbs.right_hand,
                                    dr.camera, rand_f)
            want = ops.mesh_depth_fwd(verts, dr.rasterizer.faces_i32, S, 640, 100.0)
            got = ops.mesh_render_fwd(T, dr.lbs.skin_vertex_start, dr.lbs.skin_bone, dr.lbs.skin_wv, dr.lbs.right_hand,
                                      dr.camera, rand_f, dr.rasterizer.faces_i32, S, 640, 100.0)
            assert torch.equal(got, want)
            assert torch.equal(dr(T, rand_f), want)
            assert float((want < 100.0).float().mean()) > 0.02          # a hand is there


def test_shared_reciprocal_divisions_equal_the_plain_ones():
    """The triangle kernels' pixel depth (common.h tri_pixel_depth: three divisions by the weights' sum share one refined
    reciprocal, the three by the corners' z bring theirs from the face's set-up) against the reference's seven plain
    IEEE divisions (.cu:104-110) on 2 x 10^9 pseudo-random (weights, depths) cases on both sides of every guard: not one
    bit of difference."""
    from spherehand_amd import _lib
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    for seed in (7, 20260930):
        _lib.check(_lib.lib().shr_selftest_division(seed, 1000, bad.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "shr_selftest_division")
    assert int(bad.item()) == 0
