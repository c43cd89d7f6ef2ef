"""Where a reference-sized training step spends its time (B=25 real x 3 views + 48 synthetic, 64x64)."""
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
def T(fn, reps=20):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("full step (synth + CNN fwd/bwd + losses + Adam): %.2f ms" % T(lambda: eng.step(real, pose, True, True)))
print("hand synthesizer (48 poses):                    %.2f ms" % T(lambda: eng.hand_synthesizer(pose.cuda())))
x = torch.rand(123, 64, 64, device="cuda")
def cnn():
    eng.optimizer.zero_grad(set_to_none=True)
    out, _ = eng.network.hg(x); out[0].square().mean().backward()
print("hourglass fwd+bwd on 123 crops:                 %.2f ms" % T(cnn))
scaled, orig, gt, cam, inv = eng._prepare_real(real)
xyz = ds.joints[:25].cuda().requires_grad_(True)
def losses():
    xyz.grad = None
    l, _ = eng.criterion.mv_projection_loss(cam, inv, xyz, orig, True); l.backward()
print("MutualProjectionLoss fwd+bwd (225 crops):       %.2f ms" % T(losses))
# cumulative sections of the step (each timed with a device sync at the end)
dev = eng.env.device
def upto(stage):
    def f():
        synt_dms, uv_hms, d_hms, xyz_t = eng.hand_synthesizer(pose.to(dev))
        if stage == 0: return
        scaled, orig, gt, cam, inv = eng._prepare_real(real)
        if stage == 1: return
        eng.optimizer.zero_grad(set_to_none=True)
        result = eng.ddp_network(synt_dms=synt_dms, real_dms=scaled)
        if stage == 2: return
        terms, _ = eng.criterion(result, synt_target={'uv_hms': uv_hms, 'd_hms': d_hms, 'xyz_pts': xyz_t},
                                 real_target={'real_dms': orig, 'camera_poses': cam, 'inv_camera_poses': inv, 'is_mv': True})
        if stage == 3: return
        from spherehand_amd.engine import combine_loss
        combine_loss(terms).backward()
        if stage == 4: return
        eng.optimizer.step()
    return f
names = ["synthesizer", "+ prepare real batch", "+ network forward (augment, hourglass, xyz)", "+ criterion forward",
         "+ backward", "+ Adam step"]
prev = 0.0
for i, nm in enumerate(names):
    t = T(upto(i))
    print("%-46s %6.2f ms (+%.2f)" % (nm, t, t - prev)); prev = t
from spherehand_amd import ops
for fused in (True, False, True):
    ops.FUSED_GROUP_NORM_RELU = fused
    print("full step, NHWC GroupNorm+ReLU kernels %-5s     %.2f ms ; hourglass fwd+bwd %.2f ms"
          % (fused, T(lambda: eng.step(real, pose, True, True)), T(cnn)))
