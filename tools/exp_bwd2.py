"""Backward: 16 waves / full LDS (one workgroup per CU) against 8 waves / half of the LDS (two per CU), us per 256 crops."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import _lib
if os.environ.get("SHR_LIB"):
    _lib.SO_PATH = os.environ["SHR_LIB"]
import bench
from spherehand_amd import hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); S, J = 128, 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for n in [int(v) for v in os.environ.get("NS", "256,512,1152,9216").split(",")]:
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev); gs = torch.empty(n, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
        lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], stream.cuda_stream)
        reps = max(4, 4000 // n)
        row = []
        for waves in ((16,) if os.environ.get("SHR_LIB") else (16, 8, 0)):
            if not os.environ.get("SHR_LIB"): ops.set_tuning(ops.TUNE_BWD_WAVES, waves)
            b = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s), stream, reps, 3, 3)
            row.append(b * 256 / n)
        if not os.environ.get("SHR_LIB"): ops.set_tuning(ops.TUNE_BWD_WAVES, 0)
        print("N=%5d  backward per 256 crops:" % n, " ".join("%.3f" % r for r in row), "(16 waves, 8 waves, default)", flush=True)
