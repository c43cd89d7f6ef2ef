"""Sphere rasterizer forward / backward / fused kernels: persistent workgroups (next crop's records prefetched) vs one
workgroup per crop, at 256 / 1152 / 2304 / 9216 crops @128x128 (us per launch, us per 256 crops)."""
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
modes = [int(m) for m in os.environ.get("MODES", "0,1").split(",")]
with torch.cuda.stream(stream):
    for n in (256, 1152, 2304, 9216):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev); gs = torch.empty(n, J, 4, device=dev)
        tgt = torch.full((n, S, S), 100.0, device=dev); tgt[:, 32:96, 32:96] = 0.0
        sse = torch.empty(n, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs, tgt, sse)]
        reps = max(4, 4000 // n)
        for mode in modes:
            ops.set_tuning(ops.TUNE_PERSISTENT, mode)
            f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd(p[0], n, J, S, S, p[1], p[2], s), stream, reps, 3, 3)
            b = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s), stream, reps, 3, 3)
            m = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_mse(p[0], n, J, S, S, p[5], None, p[1], p[6], p[4], s), stream, reps, 3, 3)
            print("N=%5d persistent=%d: fwd %8.2f us (%.3f /256)  bwd %8.2f us (%.3f /256)  fused %8.2f us (%.3f /256)"
                  % (n, mode, f, f * 256 / n, b, b * 256 / n, m, m * 256 / n), flush=True)
        ops.set_tuning(ops.TUNE_PERSISTENT, 1)
