#!/usr/bin/env python3
"""Batch-256 / 1152 / 9216 forward with the full owner map against SHR_RASTER_OWNER_TOUCHED_ROWS (owner bytes of
untouched rows not stored), and the step (forward + backward) in both modes; also the owner bytes actually written."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spherehand_amd import _lib, hand_model  # noqa: E402
from spherehand_amd.joint_angle import sample_poses  # noqa: E402
from spherehand_amd.kinematicsTransformation import HandTransformationMat  # noqa: E402
from spherehand_amd.render import HandBallPrimitiveRender  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
S, J = 128, 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for n in (256, 1152, 9216):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=0 if n == 256 else 7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev)
        owner = torch.full((n, S, S), 254, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev)
        gs = torch.empty(n, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
        reps = {256: 200, 1152: 40, 9216: 8}[n]
        res = {}
        for flags in (0, 1):
            f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], flags, s), stream, reps, 5, 5, warm_ms=40.0)
            b = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s), stream, reps, 5, 5, warm_ms=40.0)

            def step(s):
                lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], flags, s)
                lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s)
            st = bench.mean_launch_us(step, stream, reps, 5, 5, warm_ms=40.0)
            res[flags] = (f, b, st)
        owner.fill_(254)
        lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], 1, stream.cuda_stream)
        stream.synchronize()
        written = float((owner != 254).float().mean())
        print("N %5d  full owner: fwd %.2f bwd %.2f step %.2f us | touched rows only: fwd %.2f bwd %.2f step %.2f us | owner bytes written %.1f %%"
              % (n, *res[0], *res[1], 100 * written))
