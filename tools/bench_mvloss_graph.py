"""Config 5's loss (MutualProjectionLoss forward + backward, 1152 crops @256x256 per GPU): eager wall time per step
against the replay of the same step captured in ONE hipGraph (what the GPU needs when the host is out of the way), for
all view pairs / same-view pairs only, point-list cache off (fresh observations).  usage: tools/bench_mvloss_graph.py [B] [S]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spherehand_amd import hand_model  # noqa: E402
from spherehand_amd.datasets import SyntheticMultiviewDataset  # noqa: E402
from spherehand_amd.multiview_utility import MutualProjectionLoss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
mesh = hand_model.load_mesh()
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0, device=dev)
crit = MutualProjectionLoss(S, mesh).to(dev)
crit.cache_points = False
real, cam, inv = ds.dms.to(dev), ds.cam.to(dev), ds.inv_cam.to(dev)
joints = (ds.joints.to(dev) + torch.randn(ds.joints.shape, device=dev)).requires_grad_(True)
stream = torch.cuda.Stream(device=dev)
import gc
gc.collect(); gc.freeze()
with torch.cuda.stream(stream):
    for is_mv in (True, False):
        def step():
            joints.grad = None
            loss, _ = crit(cam, inv, joints, real, is_mv)
            loss.backward()
        for _ in range(5):
            step()
        stream.synchronize()
        t_eager = bench.mean_launch_us(lambda _s: step(), stream, 50, 3, 10)
        g = torch.cuda.CUDAGraph()
        joints.grad = None
        with torch.cuda.graph(g, stream=stream):
            loss, _ = crit(cam, inv, joints, real, is_mv)
            loss.backward()
        t_graph = bench.mean_launch_us(lambda _s: g.replay(), stream, 50, 3, 10)
        print("is_mv=%-5s eager %7.1f us   hipGraph replay %7.1f us" % (is_mv, t_eager, t_graph))
        del g
