"""data_to_model_kernel timings (HIP events, mean of back-to-back launches) at config 5's per-GPU share
(1152 crops @128^2 / @256^2), the reference training step's shape (225 crops @64^2) and batch 256 @128^2,
for each waves-per-crop launch shape.  Inputs: sphere-rendered multiview observations + noised joints."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss

mesh = hand_model.load_mesh()

def kernel_us(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def inputs(B, S):
    ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
    crit = MutualProjectionLoss(S, mesh).cuda()
    real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
    joints = ds.joints.cuda() + torch.randn_like(ds.joints.cuda())
    with torch.no_grad():
        _, pts = crit.mutual_projection(cam, inv, joints)
    N = B * 9
    obs = real.view(B * 3, S, S).contiguous()
    b = torch.arange(B, device="cuda", dtype=torch.int32).view(B, 1, 1)
    j = torch.arange(3, device="cuda", dtype=torch.int32).view(1, 1, 3)
    index = (b * 3 + j).expand(B, 3, 3).reshape(-1).contiguous()
    cen = pts.squeeze(-1).reshape(N, 41, 3).contiguous()
    rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
    return obs, index, cen, rad

if __name__ == "__main__":
    shapes = [(128, 128), (128, 256), (25, 64), (28, 128)]
    bands = [int(b) for b in os.environ.get("BANDS", "0").split(",")]
    for B, S in shapes:
      for band in bands:
        ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, band)
        obs, index, cen, rad = inputs(B, S)
        N = B * 9
        fg = float((obs <= 99).float().mean())
        line = "d2m %4d crops @%dx%d (fg %.1f%%) band %d:" % (N, S, S, 100 * fg, band)
        for waves in [int(w) for w in os.environ.get('WAVES', '4,8,16').split(',')]:
            ops.set_tuning(ops.TUNE_D2M_WAVES, waves)
            t = kernel_us(lambda: ops.data_to_model(obs, cen, rad, True, index))
            t0 = kernel_us(lambda: ops.data_to_model(obs, cen, rad, False, index))
            line += "  waves=%d: %.1f us (loss only %.1f)" % (waves, t, t0)
        ops.set_tuning(ops.TUNE_D2M_WAVES, 0)
        ops.set_tuning(ops.TUNE_D2M_BAND_UNITS, 0)
        t = kernel_us(lambda: ops.data_to_model(obs, cen, rad, True, index))
        line += "  default: %.1f us" % t
        print(line, flush=True)
