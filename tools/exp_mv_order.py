"""Config 5's loss kernels with the crops ORDERED so that the three pairs that share an observed image run on one XCD
one after the other (workgroup w -> XCD w % 8): timing of the fused render-and-compare kernel and of the point search in
the batch's own order against that order (C ABI, HIP events; the partial results land in other slots: timing only)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S, V, J = 128, 256, 3, 41
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda()
lib = _lib.lib()
real, cam, inv = ds.dms.cuda().contiguous(), ds.cam.cuda().contiguous(), ds.inv_cam.cuda().contiguous()
joints = (ds.joints.cuda() + torch.randn_like(ds.joints.cuda())).contiguous()
observed = real.reshape(B * V, S, S).contiguous()
radii = crit.mutual_projection.radiuses.view(-1).cuda().contiguous() if hasattr(crit, "mutual_projection") else None
if radii is None:
    from spherehand_amd.hand_model import radii_of
    import numpy as np
    radii = torch.from_numpy(np.asarray(radii_of(mesh), np.float32)).cuda()
N = B * V * V
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    sh = stream.cuda_stream
    spheres = torch.empty(N, J, 4, device="cuda")
    _lib.check(lib.shr_mutual_project_fwd(cam.data_ptr(), inv.data_ptr(), joints.data_ptr(), radii.data_ptr(), B, V, J, spheres.data_ptr(), sh), "proj")
    ws = ops.d2m_points_workspace(observed)
    _lib.check(lib.shr_data_to_model_compact(observed.data_ptr(), B * V, S, S, ws.data_ptr(), sh), "compact")
    b = torch.arange(B, dtype=torch.int32).view(B, 1, 1); j = torch.arange(V, dtype=torch.int32).view(1, 1, V)
    index = (b * V + j).expand(B, V, V).reshape(-1).contiguous().cuda()           # pair (b,i,j) -> image b*V+j
    ident = torch.arange(N, dtype=torch.int32).cuda()
    w = torch.arange(N)
    x, k = w % 8, w // 8
    p, i = k // 3, k % 3
    t = 8 * p + x
    perm = ((t // V * V + i) * V + t % V).to(torch.int32).cuda()                  # workgroup w -> crop: image t's three pairs in a row on XCD x
    assert sorted(perm.tolist()) == list(range(N))
    R = lib.shr_sphere_raster_mse_regions(S, S)
    depth = torch.empty(N, S, S, device="cuda")
    sse = torch.empty(N * R, device="cuda"); gsp = torch.empty(N * R * J * 4, device="cuda")
    d2m = torch.empty(N, device="cuda"); gd = torch.empty(N * J * 3, device="cuda")
    for name, order in (("batch order", ident), ("by image, per XCD", perm), ("batch order", ident), ("by image, per XCD", perm)):
        tgt_of = index.index_select(0, order.long()).contiguous()
        for want_depth in (True, False):
            mse = lambda s: lib.shr_sphere_raster_mse_indexed(spheres.data_ptr(), order.data_ptr(), N, J, S, S, observed.data_ptr(), index.data_ptr(),
                                                              depth.data_ptr() if want_depth else None, sse.data_ptr(), gsp.data_ptr(), s)
            assert mse(sh) == 0
            print("%-18s render-and-compare (%s): %7.1f us" % (name, "depth written" if want_depth else "no depth", bench.mean_launch_us(mse, stream, 20, 3, 3, warm_ms=30.0)), flush=True)
        pts = lambda s: lib.shr_data_to_model_from_points_indexed(ws.data_ptr(), B * V, tgt_of.data_ptr(), order.data_ptr(), spheres.data_ptr(), 4, radii.data_ptr(),
                                                                  N, J, S, S, 1, d2m.data_ptr(), gd.data_ptr(), s)
        assert pts(sh) == 0
        print("%-18s point search: %7.1f us" % (name, bench.mean_launch_us(pts, stream, 20, 3, 3, warm_ms=30.0)), flush=True)
