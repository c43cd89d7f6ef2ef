"""DepthRender at config 5's resolution (S = 256 from 640): module, raster alone, band kernel vs tile kernel.
    python tools/bench_mesh256.py [B S]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from spherehand_amd import hand_model, ops, _lib
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import DepthRender
from bench import mean_launch_us
B, S = (int(v) for v in (sys.argv[1:3] + [64, 256][len(sys.argv) - 1:]))
mesh = hand_model.load_mesh()
dev = torch.device("cuda")
stream = torch.cuda.Stream()
dr = DepthRender(mesh, S).to(dev)
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
lib = _lib.lib()
with torch.cuda.stream(stream), torch.no_grad():
    T = fk(sample_poses(B, seed=1).to(dev))
    verts = dr.lbs(T, dr.camera, None).contiguous()
    out = torch.empty(B, S, S, device=dev)
    nv, nf = verts.shape[1], dr.rasterizer.num_faces
    v = [verts.data_ptr(), dr.rasterizer.faces_i32.data_ptr(), out.data_ptr()]
    res = {}
    for name, mode in (("band kernel + resize epilogue", 1), ("tile kernel", 0)):
        ops.set_tuning(ops.TUNE_MESH_BAND, mode)
        res[name] = mean_launch_us(lambda s: _lib.check(lib.shr_mesh_depth_fwd(v[0], v[1], B, nv, nf, 640, S, 100.0, v[2], s), "mesh"),
                                   stream, 20, 3, 3)
        t_mod = mean_launch_us(lambda _s: dr(T), stream, 20, 3, 3)
        print("B=%d S=%d %s: raster %.1f us, DepthRender module %.1f us" % (B, S, name, res[name], t_mod), flush=True)
    ops.set_tuning(ops.TUNE_MESH_BAND, 1)
