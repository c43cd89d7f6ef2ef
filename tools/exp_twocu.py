"""Does a second resident workgroup per CU pay?  The depth-only forward (32-bit keys, 73 KB of LDS) already fits two per
CU: time it as is and with its LDS padded past 80 KB (one per CU), beside the forward with the owner map (139 KB)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import _lib
if os.environ.get("SHR_LIB"):          # an experiment build of the library
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
        p = [t.data_ptr() for t in (sph, depth, owner)]
        reps = max(4, 4000 // n)
        for persist in [int(v) for v in os.environ.get("PERSIST", "0,1").split(",")]:
            ops.set_tuning(ops.TUNE_PERSISTENT, persist)
            row = []
            for pad in (0, 12 * 1024):
                ops.set_tuning(99, pad)
                f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], None, s), stream, reps, 3, 3)
                row.append(f * 256 / n)
            ops.set_tuning(99, 0)
            o = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], s), stream, reps, 3, 3)
            print("N=%5d persistent=%d  per 256 crops: depth-only two/CU %.3f us, one/CU %.3f us; with owner map (one/CU) %.3f us"
                  % (n, persist, row[0], row[1], o * 256 / n), flush=True)
