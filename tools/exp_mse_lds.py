"""Fused render-and-compare, 1152 crops @256x256: the box variant at several LDS sizes per workgroup."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
lib = _lib.lib(); dev = torch.device("cuda:0"); J, S, n = 41, 256, 1152
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
    with torch.no_grad():
        sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
    R = lib.shr_sphere_raster_mse_regions(S, S)
    depth = torch.empty(n, S, S, device=dev); tgt = torch.full((n, S, S), 100.0, device=dev); tgt[:, S // 4:3 * S // 4, S // 4:3 * S // 4] = 0.0
    sse = torch.empty(n * R, device=dev); gs = torch.empty(n * R, J, 4, device=dev)
    p = [t.data_ptr() for t in (sph, tgt, depth, sse, gs)]
    for kb in (0, 48, 64, 72, 80, 96, 112, 128, 160):
        ops.set_tuning(ops.TUNE_MSE_BOX, kb * 1024)
        t = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_mse(p[0], n, J, S, S, p[1], None, p[2], p[3], p[4], s), stream, 10, 5, 3, warm_ms=30.0)
        print("LDS %3d KB (%s): %.1f us" % (kb, "whole region" if kb == 0 else "box", t), flush=True)
