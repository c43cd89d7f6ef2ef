#!/usr/bin/env python3
"""pose -> FK -> key-point skinning -> raster fwd -> bwd -> skinning bwd -> FK bwd, per kernel (HIP events, back to back)
and as one hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spherehand_amd import hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
dev = torch.device("cuda:0"); S, B = 128, 256
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
pose = sample_poses(B, seed=0).to(dev).requires_grad_(True)
g = torch.randn(B, S, S, device=dev)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    def chain():
        pose.grad = None
        d = ops.SphereDepthRaster.apply(hbr.spheres(fk(pose)).contiguous(), S, S)
        d.backward(g)
    for _ in range(3): chain()
    stream.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=stream):
        chain()
    print("chain as one graph: %.2f us" % bench.mean_launch_us(lambda _s: gr.replay(), stream, 100, 3, 10))
    T = fk(pose).detach()
    GT = torch.randn_like(T)
    p2 = pose.detach()
    off, inv = fk.offset, fk.offset_inv
    from spherehand_amd import _lib
    lib = _lib.lib()
    out = torch.empty(B, 17, 4, 4, device=dev); gp = torch.empty(B, 26, device=dev)
    a = [t.data_ptr() for t in (p2, off, inv, out, GT, gp)]
    print("fk_fwd %.2f us, fk_bwd %.2f us" % (
        bench.mean_launch_us(lambda s: lib.shr_fk_fwd(a[0], B, a[1], a[2], a[3], s), stream, 100, 3, 10),
        bench.mean_launch_us(lambda s: lib.shr_fk_bwd(a[0], B, a[1], a[2], a[4], a[5], s), stream, 100, 3, 10)))
