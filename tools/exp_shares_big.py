"""Work-list shares of the four wave age groups for the BOX forward (two workgroups per CU), 9216 crops."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); S, J, n = 128, 41, 9216
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
stream = torch.cuda.Stream(device=dev)
pack = lambda a: a[0] | a[1] << 8 | a[2] << 16 | a[3] << 24
with torch.cuda.stream(stream):
    with torch.no_grad():
        sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
    depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
    p = [t.data_ptr() for t in (sph, depth, owner)]
    for c in [(100, 68, 52, 36), (64, 64, 64, 64), (80, 68, 60, 48), (88, 72, 56, 40), (72, 68, 60, 56), (112, 68, 46, 30), (92, 68, 54, 42), (100, 68, 52, 36)]:
        ops.set_tuning(ops.TUNE_FWD_SHARES, pack(c))
        f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], s), stream, 8, 5, 3, warm_ms=30.0)
        print(c, "%.3f us per 256 crops" % (f * 256 / n), flush=True)
