#!/usr/bin/env python3
"""Forward (and step) at 256 / 384 / 448 crops with the run table on and off (SHR_TUNE_FWD_RUN_TABLE), alternating
A/B/A/B so that clock drift shows; bit-identity of the two outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spherehand_amd import _lib, hand_model, ops  # noqa: E402
from spherehand_amd.joint_angle import sample_poses  # noqa: E402
from spherehand_amd.kinematicsTransformation import HandTransformationMat  # noqa: E402
from spherehand_amd.render import HandBallPrimitiveRender  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
S, J = int(os.environ.get("S", 128)), 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for n in (256, 384, 448):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=0 if n == 256 else 7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev)
        owner = torch.full((n, S, S), 254, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev)
        gs = torch.empty(n, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
        outs = {}
        for rnd in range(3):
            for mode in (0, -1):
                ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, mode)
                r = []
                for flags in (0, 1):
                    f = bench.mean_launch_us(lambda s: lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], flags, s), stream, 400, 5, 5, warm_ms=40.0)

                    def step(s):
                        lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], flags, s)
                        lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s)
                    st = bench.mean_launch_us(step, stream, 400, 5, 5, warm_ms=40.0)
                    r += [f, st]
                owner.fill_(254)
                lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], 0, stream.cuda_stream)
                stream.synchronize()
                outs[mode] = (depth.clone(), owner.clone())
                print("N %4d round %d table %2d: full map fwd %.2f step %.2f | touched rows fwd %.2f step %.2f us" % (n, rnd, mode, *r), flush=True)
        same = torch.equal(outs[0][0].view(torch.int32), outs[-1][0].view(torch.int32)) and torch.equal(outs[0][1], outs[-1][1])
        print("N %4d bit-identical: %s" % (n, same), flush=True)
ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, -1)
