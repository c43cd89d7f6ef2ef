"""depth_rasterization.forward(640, 640): the LDS band kernel against the global-atomic kernel, B = 1 / 48 / 256."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, depth_rasterization
from spherehand_amd import hand_model, ops
from spherehand_amd.render import DepthRender
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.joint_angle import sample_poses
mesh = hand_model.load_mesh()
dev = torch.device("cuda", 0)
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
dr = DepthRender(mesh, 128).to(dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for B in [int(v) for v in os.environ.get("BS", "256,48,1").split(",")]:
        with torch.no_grad():
            verts = dr.lbs(fk(sample_poses(B, seed=1).to(dev)), dr.camera, None)
            fv = verts[:, dr.rasterizer.faces, 0:3].reshape(B, -1, 3, 3).contiguous()
        outs = []
        for band in [int(v) for v in os.environ.get("BANDS", "0,-1,32,16").split(",")]:
            ops.set_tuning(ops.TUNE_TRI_BAND, band)
            outs.append(depth_rasterization.forward(640, 640, fv))
            t = bench.mean_launch_us(lambda _s: depth_rasterization.forward(640, 640, fv), stream, 10, 3, 3)
            print("B=%3d band %3d: %8.1f us  same bits as the atomic kernel: %s" % (B, band, t, torch.equal(outs[0], outs[-1])), flush=True)
        ops.set_tuning(ops.TUNE_TRI_BAND, -1)
