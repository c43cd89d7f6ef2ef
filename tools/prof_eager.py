import sys, os, cProfile, pstats, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from spherehand_amd import hand_model, ops
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
dev = torch.device("cuda", 0)
mesh = hand_model.load_mesh()
S = 128
fkm = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
pose = sample_poses(256, seed=0).to(dev).requires_grad_(True)
gdepth = torch.randn(256, S, S, device=dev)
def chain():
    pose.grad = None
    depth = ops.SphereDepthRaster.apply(hbr.spheres(fkm(pose)).contiguous(), S, S)
    depth.backward(gdepth)
for rep in range(3):
    for _ in range(50): chain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): chain()
    torch.cuda.synchronize()
    print("eager chain: %.1f us per iteration" % ((time.perf_counter() - t0) / 500 * 1e6), flush=True)
if os.environ.get("PROFILE"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): chain()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
