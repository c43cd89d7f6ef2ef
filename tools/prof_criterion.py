"""Kernel launches and GPU/CPU time of the loss side of a reference-sized step (network output -> losses -> backward)."""
import os, sys, time, collections
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.engine import Engine, combine_loss
from spherehand_amd.joint_angle import sample_poses
from torch.profiler import profile, ProfilerActivity
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
dev = eng.env.device
synt_dms, uv_hms, d_hms, xyz_t = eng.hand_synthesizer(pose.to(dev))
scaled, orig, gt, cam, inv = eng._prepare_real(real)
with torch.no_grad():
    out, lat = eng.network.hg(torch.cat([synt_dms, scaled.reshape(-1, 64, 64)], 0))
def loss_side():
    o_ = [t.detach().clone().requires_grad_(True) for t in out]
    l_ = [t.detach() for t in lat]
    net = eng.network
    n_synt = synt_dms.shape[0]
    res = {}
    so = [x[:n_synt] for x in o_]; uv, d = net._split(so)
    res.update({'synt_uv_hms': uv, 'synt_d_hms': d, 'synt_xyz': [net.xyz_recover.from_output(x) for x in so]})
    ro = [x[n_synt:] for x in o_]; uv, d = net._split(ro)
    ones = torch.ones(75, device=dev)
    res.update(net._real_result(ro, uv, d, ones, ones, 25, 3))
    res['batch_synt_fea'] = [l[:n_synt] for l in l_]; res['batch_real_fea'] = [l[n_synt:] for l in l_]
    terms, _ = eng.criterion(res, synt_target={'uv_hms': uv_hms, 'd_hms': d_hms, 'xyz_pts': xyz_t},
                             real_target={'real_dms': orig, 'camera_poses': cam, 'inv_camera_poses': inv, 'is_mv': True})
    combine_loss(terms).backward()
    return terms
for _ in range(5): loss_side()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): loss_side()
torch.cuda.synchronize(); print("loss side (read-out + criterion + backward to the network output): %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    loss_side(); torch.cuda.synchronize()
ka = prof.key_averages()
n_launch = sum(e.count for e in ka if e.key in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel"))
print("kernel launches:", n_launch)
print(ka.table(sort_by="self_cpu_time_total", row_limit=18, max_name_column_width=50))
