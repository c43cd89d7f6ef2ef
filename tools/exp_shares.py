"""Work-list shares of the four wave age groups (SHR_TUNE_FWD_SHARES / BWD_SHARES) at 1152 / 9216 crops @128x128."""
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
with torch.cuda.stream(stream):
    for n in (9216, 1152):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev); gs = torch.empty(n, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
        reps = 40 if n <= 1152 else 8
        f = lambda s: lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], 1, s)
        b = lambda s: lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s)
        f(stream.cuda_stream)
        for sh in (0x24344464, 0x40404040, 0x30384850, 0x38404448, 0x48444038, 0x50483830, 0x1c304c68):
            ops.set_tuning(ops.TUNE_FWD_SHARES, sh)
            tf = bench.mean_launch_us(f, stream, reps, 3, 3, warm_ms=30.0)
            print("n %5d fwd shares %08x: %7.2f us" % (n, sh, tf), flush=True)
        ops.set_tuning(ops.TUNE_FWD_SHARES, 0x24344464)
        for sh in (0x2c3a4654, 0x40404040, 0x383e4246, 0x46423e38, 0x20304c64):
            ops.set_tuning(ops.TUNE_BWD_SHARES, sh)
            tb = bench.mean_launch_us(b, stream, reps, 3, 3, warm_ms=30.0)
            print("n %5d bwd shares %08x: %7.2f us" % (n, sh, tb), flush=True)
        ops.set_tuning(ops.TUNE_BWD_SHARES, 0x2c3a4654)
