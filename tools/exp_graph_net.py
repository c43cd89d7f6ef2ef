"""Reference-sized training step with the hourglass forward + backward replayed as HIP graphs
(torch.cuda.make_graphed_callables) against eager launches."""
import os, sys, time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.engine import Engine
from spherehand_amd.joint_angle import sample_poses
mesh = hand_model.load_mesh()
o = SimpleNamespace(synthesize=True, mv_projection=True, mv_consistency=True, temporal=False, prior=False, collision=True,
                    bone_length=True, mode='Train', model_dir='/tmp/eng', initial_model=None, restore_from_model=None,
                    restore_from_epoch=-1, num_stacks=1, epoch=3, dataset_dir=None, depth_resample=0, lr=1e-3, tag='b',
                    image_size=64, log_every=10**9, real_batch=25, synt_batch=48)
ds = SyntheticMultiviewDataset(mesh, 50, 64, seed=0)
eng = Engine(o, mesh=mesh, real_train_dataset=ds, real_eval_dataset=ds)
eng.network.train()
real = [torch.stack([ds[i][k] for i in range(25)]) for k in range(4)]
pose = sample_poses(48, seed=1)
def T(n=30):
    for _ in range(8): eng.step(real, pose, True, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.step(real, pose, True, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager: %.2f ms, %.2f ms" % (T(), T()))
terms0 = {k: float(v) for k, v in eng.step(real, pose, True, True)[0].items()}
x = torch.rand(123, 64, 64, device="cuda")
hg = eng.network.hg
t0 = time.perf_counter()
eng.network.hg = torch.cuda.make_graphed_callables(hg, (x,))
print("captured in %.1f s" % (time.perf_counter() - t0))
print("graphed hourglass: %.2f ms, %.2f ms" % (T(), T()))
print({k: round(float(v), 4) for k, v in eng.step(real, pose, True, True)[0].items()})
