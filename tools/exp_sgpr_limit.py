"""Does a forward box kernel with 84 SGPRs (the non-power-of-two instantiation) still get two workgroups per CU?
Its launch is timed as is and with its LDS padded past half of the CU's (one workgroup per CU)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); J, n = 41, 4608
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for S in (128, 132):
        hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        p = [t.data_ptr() for t in (sph, depth, owner)]
        for pad in (0, 16 * 1024):
            ops.set_tuning(99, pad)
            f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], s), stream, 10, 5, 3, warm_ms=30.0)
            print("S=%d  LDS pad %5d: %.2f us per 256 crops" % (S, pad, f * 256 / n), flush=True)
        ops.set_tuning(99, 0)
