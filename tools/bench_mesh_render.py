"""DepthRender's back end through the C ABI (no Python op in the loop): shr_mesh_render_fwd (one launch) against
shr_lbs_project + shr_mesh_depth_fwd (two), HIP events on the launching stream."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model
from spherehand_amd.render import DepthRender
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.joint_angle import sample_poses
mesh = hand_model.load_mesh()
lib = _lib.lib()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).cuda()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B, S in ((256, 128), (256, 64), (48, 64)):
        dr = DepthRender(mesh, S).cuda()
        T = fk(sample_poses(B, seed=1).cuda()).contiguous()
        l = dr.lbs
        NV, F = l.num_vertices, dr.rasterizer.faces_i32.shape[0]
        ws = torch.empty(B, NV, 4, device="cuda"); out = torch.empty(B, S, S, device="cuda"); out2 = torch.empty_like(out)
        cx, cy, fx, fy = dr.camera
        p = lambda t: t.data_ptr()
        one = lambda s: lib.shr_mesh_render_fwd(p(T), B, 17, NV, p(l.skin_vertex_start), p(l.skin_bone), p(l.skin_wv), 1, cx, cy, fx, fy,
                                                None, p(dr.rasterizer.faces_i32), F, 640, S, 100.0, p(ws), p(out), s)
        def two(s):
            lib.shr_lbs_project(p(T), B, 17, NV, p(l.skin_vertex_start), p(l.skin_bone), p(l.skin_wv), 1, 1, cx, cy, fx, fy, None, p(ws), s)
            return lib.shr_mesh_depth_fwd(p(ws), p(dr.rasterizer.faces_i32), B, NV, F, 640, S, 100.0, p(out2), s)
        ras = lambda s: lib.shr_mesh_depth_fwd(p(ws), p(dr.rasterizer.faces_i32), B, NV, F, 640, S, 100.0, p(out2), s)
        assert one(stream.cuda_stream) == 0 and two(stream.cuda_stream) == 0
        stream.synchronize()
        t1 = bench.mean_launch_us(one, stream, 100, 3, 5, warm_ms=20.0)
        t2 = bench.mean_launch_us(two, stream, 100, 3, 5, warm_ms=20.0)
        t3 = bench.mean_launch_us(ras, stream, 100, 3, 5, warm_ms=20.0)
        print("B=%d S=%d: one launch %.1f us | skinning + raster %.1f us (raster alone %.1f) | same bits: %s"
              % (B, S, t1, t2, t3, torch.equal(out, out2)), flush=True)
