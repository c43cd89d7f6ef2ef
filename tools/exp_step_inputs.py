"""Training step with the real batch on the host (as a DataLoader hands it over) against device-resident tensors."""
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
dev = torch.device("cuda:0")
o = SimpleNamespace(synthesize=True, mv_projection=True, mv_consistency=True, temporal=False, prior=False, collision=True,
                    bone_length=True, mode='Train', model_dir='/tmp/eng', initial_model=None, restore_from_model=None,
                    restore_from_epoch=-1, num_stacks=1, epoch=3, dataset_dir=None, depth_resample=0, lr=1e-3, tag='b',
                    image_size=64, log_every=10**9, real_batch=25, synt_batch=48)
on_dev = os.environ.get("ON_DEVICE", "0") == "1"
ds = SyntheticMultiviewDataset(mesh, 50, 64, seed=0, device=dev) if on_dev else SyntheticMultiviewDataset(mesh, 50, 64, seed=0)
eng = Engine(o, mesh=mesh, real_train_dataset=ds, real_eval_dataset=ds, device=dev)
eng.network.train()
real = [torch.stack([ds[i][k] for i in range(25)]) for k in range(4)]
pose = sample_poses(48, seed=1)
for _ in range(10): eng.step(real, pose, True, True)
import gc
if os.environ.get("GC") == "off": gc.disable()
if os.environ.get("GC") == "freeze": gc.collect(); gc.freeze()
b = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.step(real, pose, True, True)
    torch.cuda.synchronize(); b.append((time.perf_counter() - t0) / 10 * 1e3)
print("gc=%s real batch on %s: %s ms" % (os.environ.get("GC", "default"), "device" if on_dev else "host", " ".join("%.2f" % t for t in sorted(b))))
