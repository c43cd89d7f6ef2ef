import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spherehand_amd import hand_model
from spherehand_amd.render import HandBallPrimitiveRender
from spherehand_amd.kinematicsTransformation import HandTransformationMat
g = np.load("tests/golden/g3_batch256.npz")
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype(np.float32) for b in mesh["bones"]]).cuda()
hbr = HandBallPrimitiveRender(mesh["bones"], 128, 128).cuda()
p = torch.from_numpy(g["params"]).cuda().requires_grad_(True)
_, depth = hbr(fk(p))
gd = torch.from_numpy(np.random.RandomState(1).standard_normal((256, 128, 128)).astype(np.float32)).cuda()
(depth * gd).sum().backward()
a, ref = p.grad.cpu().numpy(), g["grad_params"]
err = np.abs(a - ref)
print("max err", err.max(), "max ref", np.abs(ref).max(), "rel l2", np.linalg.norm(a - ref) / np.linalg.norm(ref))
print("err percentiles", np.percentile(err, [50, 90, 99, 99.9, 100]))
print("samples with err > 1:", np.unique(np.argwhere(err > 1.0)[:, 0]))
d = depth.detach().cpu().numpy()[:16]
print("flipped px in first16", ((d >= 100) != (g["depth_first16_ieee"] >= 100)).sum())
# same but feed reference centres directly (raster bwd only) then torch FK autograd
