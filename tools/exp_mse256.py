"""Fused render-and-compare, 1152 crops @256x256 and 9216 @128x128: box variant (two workgroups per CU) against the whole-region one."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); J = 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for S, n in ((256, 1152), (128, 1152), (128, 9216)):
        hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        R = lib.shr_sphere_raster_mse_regions(S, S)
        depth = torch.empty(n, S, S, device=dev); tgt = torch.full((n, S, S), 100.0, device=dev); tgt[:, S // 4:3 * S // 4, S // 4:3 * S // 4] = 0.0
        sse = torch.empty(n * R, device=dev); gs = torch.empty(n * R, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, tgt, depth, sse, gs)]
        res = {}
        for box, tag in ((0, "whole region"), (1, "box, half of the LDS"), (-1, "launcher's choice")):
            ops.set_tuning(ops.TUNE_MSE_BOX, box)
            t = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_mse(p[0], n, J, S, S, p[1], None, p[2], p[3], p[4], s), stream, 10, 5, 3, warm_ms=30.0)
            res[box] = (depth.clone(), sse.clone().view(n, R).sum(1), gs.clone().view(n, R, J, 4).sum(1))
            print("S=%d N=%d  %-22s %.1f us" % (S, n, tag, t), flush=True)
        ops.set_tuning(ops.TUNE_MSE_BOX, -1)
        a, b = res[0], res[1]
        print("   depth identical:", bool(torch.equal(a[0], b[0])), " sse rel diff %.2e  grad diff %.2e of %.2e" % (
            float(((a[1] - b[1]).abs() / a[1].abs().clamp_min(1)).max()), float((a[2] - b[2]).abs().max()), float(a[2].abs().max())))
