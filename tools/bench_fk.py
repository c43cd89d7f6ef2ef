"""Launch durations of the pose <-> bone transforms <-> sphere records kernels (fk.hip, keypoint_skin.hip) at B poses:
HIP events over back-to-back launches, and the chain pose -> depth -> pose as one hipGraph.  usage: tools/bench_fk.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spherehand_amd import _lib, hand_model  # noqa: E402
from spherehand_amd.joint_angle import sample_poses  # noqa: E402
from spherehand_amd.kinematicsTransformation import HandTransformationMat  # noqa: E402
from spherehand_amd.render import HandBallPrimitiveRender  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], 128, 128).to(dev)
lbs = hbr.lbs
lib = _lib.lib()
p = sample_poses(B, seed=0).to(dev)
T = torch.empty(B, 17, 4, 4, device=dev)
gT = torch.randn(B, 17, 4, 4, device=dev)
sph = torch.empty(B, 41, 4, device=dev)
gs = torch.randn(B, 41, 4, device=dev)
gp = torch.empty(B, 26, device=dev)
stream = torch.cuda.Stream(device=dev)
a = dict(p=p.data_ptr(), o=fk.offset.data_ptr(), i=fk.offset_inv.data_ptr(), T=T.data_ptr(), gT=gT.data_ptr(), s=sph.data_ptr(),
         gs=gs.data_ptr(), gp=gp.data_ptr(), bone=lbs.kp_bone.data_ptr(), wv=lbs.skin_wv.data_ptr(), r=hbr.radiuses.data_ptr(),
         bs=lbs.kp_bone_start.data_ptr(), bp=lbs.kp_bone_points.data_ptr())
runs = {
    "fk_fwd": lambda s: lib.shr_fk_fwd(a["p"], B, a["o"], a["i"], a["T"], s),
    "fk_bwd": lambda s: lib.shr_fk_bwd(a["p"], B, a["o"], a["i"], a["gT"], a["gp"], s),
    "keypoint_spheres_fwd": lambda s: lib.shr_keypoint_spheres_fwd(a["T"], B, 17, 41, a["bone"], a["wv"], a["r"], 1, a["s"], s),
    "keypoint_spheres_bwd": lambda s: lib.shr_keypoint_spheres_bwd(a["gs"], B, 17, 41, a["bs"], a["bp"], a["wv"], 1, a["gT"], s),
    "pose_spheres_fwd": lambda s: lib.shr_pose_spheres_fwd(a["p"], B, a["o"], a["i"], 41, a["bone"], a["wv"], a["r"], 1, a["s"], None, s),
    "pose_spheres_bwd": lambda s: lib.shr_pose_spheres_bwd(a["p"], B, a["o"], a["i"], 41, a["bs"], a["bp"], a["wv"], 1, a["gs"], a["gp"], s),
}
with torch.cuda.stream(stream):
    for name, fn in runs.items():
        assert fn(stream.cuda_stream) == 0
        print("%-22s %7.2f us" % (name, bench.mean_launch_us(fn, stream, 200, 5, 50)))
    pose = p.clone().requires_grad_(True)
    gd = torch.randn(B, 128, 128, device=dev)

    def chain():
        pose.grad = None
        hbr.pose_depth(fk, pose).backward(gd)
    for _ in range(3):
        chain()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        chain()
    print("pose->depth->pose graph %7.2f us" % bench.mean_launch_us(lambda _s: g.replay(), stream, 100, 3, 10))
    print("pose->depth->pose eager %7.2f us" % bench.mean_launch_us(lambda _s: chain(), stream, 50, 5, 30, warm_ms=800.0))
