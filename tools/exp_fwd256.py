"""Forward with owner map, 1152 crops @256x256 (config 5's per-GPU share): box z-buffer at half of the LDS (two
workgroups per CU) against the whole 64-row region per workgroup."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); S, J, n = 256, 41, 1152
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    with torch.no_grad():
        sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
    depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
    p = [t.data_ptr() for t in (sph, depth, owner)]
    for zb, tag in ((0, "auto (box, half of the LDS)"), (160 * 1024, "whole region"), (60 * 1024, "box 60 KB"), (48 * 1024, "box 48 KB")):
        ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, zb)
        f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], s), stream, 20, 5, 3, warm_ms=30.0)
        print("%-30s %.1f us" % (tag, f), flush=True)
