"""Kernel launches and GPU time per section of the reference-sized training step (torch profiler, record_function ranges)."""
import os, sys, collections
from types import SimpleNamespace
import torch
from torch.profiler import profile, ProfilerActivity, record_function
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.engine import Engine, RunningAverage, combine_loss
from spherehand_amd.criterion import average_joint_error
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
losses = RunningAverage(); mets = RunningAverage()
def step():
    dev = eng.env.device
    with record_function("S1_synth"):
        synt_dms, uv_hms, d_hms, xyz = eng.hand_synthesizer(pose.to(dev))
    with record_function("S2_prepare"):
        scaled, orig, gt, cam, inv = eng._prepare_real(real)
    with record_function("S3_zero_grad"):
        eng.optimizer.zero_grad(set_to_none=True)
    with record_function("S4_network"):
        result = eng.ddp_network(synt_dms=synt_dms, real_dms=scaled)
    with record_function("S5_criterion"):
        terms, _ = eng.criterion(result, synt_target={'uv_hms': uv_hms, 'd_hms': d_hms, 'xyz_pts': xyz},
                                 real_target={'real_dms': orig, 'camera_poses': cam, 'inv_camera_poses': inv, 'is_mv': True})
    with record_function("S6_metric"):
        m = {'avg_joint_error': average_joint_error(gt, result['real_xyz'][-1].detach())}
    with record_function("S7_combine"):
        loss = combine_loss(terms)
    with record_function("S8_backward"):
        loss.backward()
    with record_function("S9_adam"):
        eng.optimizer.step()
    with record_function("S10_running_average"):
        losses.append(terms); mets.append(m)
for _ in range(8): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
ev = prof.events()
ranges = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("S") and "_" in e.name and e.name[1].isdigit()]
count = collections.Counter(); gpu = collections.Counter(); ops_in = collections.defaultdict(collections.Counter)
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA or not e.kernels: continue
    for name, a, b in ranges:
        if a <= e.time_range.start <= b and e.name != name and not e.name.startswith("S"):
            # only leaf ops that launched kernels
            count[name] += len(e.kernels); gpu[name] += sum(k.duration for k in e.kernels)
            ops_in[name][e.name] += len(e.kernels)
            break
for name in sorted(count, key=lambda s: int(s[1:s.index("_")])):
    print("%-22s %6.1f launches/step %8.1f us GPU/step   top: %s" % (name, count[name] / 5, gpu[name] / 5,
          ", ".join("%s x%.0f" % (k, v / 5) for k, v in ops_in[name].most_common(6))))
