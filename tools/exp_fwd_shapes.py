"""Forward (+ owner bytes on the touched rows) and backward at 1152 / 9216 crops @128x128: workgroup shapes
(waves per workgroup x z-buffer bytes -> workgroups per CU).  usage: tools/exp_fwd_shapes.py [n ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, ops, hand_model
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
dev = torch.device("cuda", 0)
S, J = 128, 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
lib = _lib.lib()
stream = torch.cuda.Stream(device=dev)
ns = [int(a) for a in sys.argv[1:]] or [9216, 1152]
with torch.cuda.stream(stream):
    for n in ns:
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev); gs = torch.empty(n, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
        ref = None
        reps = 40 if n <= 1152 else 8
        for waves, zb in ((16, 0), (16, 50000), (12, 0), (12, 50000), (10, 50000), (8, 0), (8, 78000), (8, 50000), (8, 37000), (8, 30000), (6, 37000), (4, 37000)):
            ops.set_tuning(ops.TUNE_FWD_WAVES, waves); ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, zb)
            f = lambda s: lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], 1, s)
            assert f(stream.cuda_stream) == 0
            stream.synchronize()
            if ref is None: ref = depth.clone()
            assert torch.equal(ref, depth)
            t = bench.mean_launch_us(f, stream, reps, 3, 3, warm_ms=30.0)
            print("n %5d fwd waves %2d zbuf %6d: %7.2f us  (%.3f per 256)" % (n, waves, zb, t, t * 256 / n), flush=True)
        ops.set_tuning(ops.TUNE_FWD_WAVES, 16); ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, 0)
        for bw, bl in ((0, 0), (8, 0), (16, 0), (8, 60000), (8, 50000)):
            ops.set_tuning(ops.TUNE_BWD_WAVES, bw); ops.set_tuning(ops.TUNE_BWD_LDS_BYTES, bl if bl else 128 * 1024)
            f = lambda s: lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s)
            assert f(stream.cuda_stream) == 0
            t = bench.mean_launch_us(f, stream, reps, 3, 3, warm_ms=30.0)
            print("n %5d bwd waves %2d lds %6d: %7.2f us  (%.3f per 256)" % (n, bw, bl, t, t * 256 / n), flush=True)
        ops.set_tuning(ops.TUNE_BWD_WAVES, 0); ops.set_tuning(ops.TUNE_BWD_LDS_BYTES, 128 * 1024)
