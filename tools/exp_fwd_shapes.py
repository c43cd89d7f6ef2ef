"""Forward with owner map at 1152 / 9216 crops: workgroup shapes (waves x z-buffer bytes -> workgroups per CU)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); S, J = 128, 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for n in (1152, 9216):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        p = [t.data_ptr() for t in (sph, depth, owner)]
        reps = max(4, 4000 // n)
        for waves, zb in ((16, 0), (16, 60 * 1024), (8, 76 * 1024), (8, 49 * 1024), (8, 36 * 1024), (12, 49 * 1024)):
            ops.set_tuning(ops.TUNE_FWD_WAVES, waves); ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, zb)
            f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], s), stream, reps, 5, 3, warm_ms=30.0)
            print("N=%5d  %2d waves, z-buffer %6d B: %.3f us per 256 crops" % (n, waves, zb, f * 256 / n), flush=True)
        ops.set_tuning(ops.TUNE_FWD_WAVES, 16); ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, 0)
