#!/usr/bin/env python3
"""Config 5's per-GPU share (1152 crops @256x256, and @128x128): render-and-compare, the pixel-order data->model kernel
and the tile-sorted one at each search width; the whole MutualProjectionLoss forward + backward with either."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spherehand_amd import _lib, hand_model, ops  # noqa: E402
from spherehand_amd.datasets import SyntheticMultiviewDataset  # noqa: E402
from spherehand_amd.multiview_utility import MutualProjectionLoss  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
mesh = hand_model.load_mesh()
J = 41
stream = torch.cuda.Stream(device=dev)
sizes = [int(s) for s in os.environ.get("SIZES", "256,128").split(",")]
B5 = int(os.environ.get("B", "128"))
with torch.cuda.stream(stream):
    for S5 in sizes:
        ds = SyntheticMultiviewDataset(mesh, B5, S5, seed=0, device=dev)
        crit = MutualProjectionLoss(S5, mesh).to(dev)
        real, cam, inv = ds.dms.to(dev), ds.cam.to(dev), ds.inv_cam.to(dev)
        joints = (ds.joints.to(dev) + torch.randn(ds.joints.shape, device=dev)).requires_grad_(True)
        n5 = B5 * 9
        with torch.no_grad():
            _, pts = crit.mutual_projection(cam, inv, joints.detach())
        obs = real.view(B5 * 3, S5, S5).contiguous()
        index = (torch.arange(B5, device=dev, dtype=torch.int32).view(B5, 1, 1) * 3 +
                 torch.arange(3, device=dev, dtype=torch.int32).view(1, 1, 3)).expand(B5, 3, 3).reshape(-1).contiguous()
        cen = pts.squeeze(-1).reshape(n5, J, 3).contiguous()
        rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
        sph = torch.cat([cen, rad.view(1, J, 1).expand(n5, J, 1)], -1).contiguous()
        R = lib.shr_data_to_model_parts(n5, S5, S5)
        ls = torch.empty(n5 * R, device=dev); gr = torch.empty(n5 * R, J, 3, device=dev)
        a = [t.data_ptr() for t in (obs, index, sph, rad, ls, gr)]
        t_d2m = bench.mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_partial(a[0], a[1], a[2], 4, a[3], n5, J, S5, S5, R,
                                                                                        a[4], a[5], s), "d2m"), stream, 20, 3, 3)
        Rm = lib.shr_sphere_raster_mse_regions(S5, S5)
        dep = torch.empty(n5, S5, S5, device=dev); sse = torch.empty(n5 * Rm, device=dev)
        gsp = torch.empty(n5 * Rm, J, 4, device=dev)
        m = [t.data_ptr() for t in (sph, obs, index, dep, sse, gsp)]
        t_mse = bench.mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_mse(m[0], n5, J, S5, S5, m[1], m[2], m[3], m[4],
                                                                                    m[5], s), "mse"), stream, 20, 3, 3)
        print("S=%d  crops=%d  fg=%.3f  mse %.1f us   d2m %.1f us (R=%d)   sum %.1f" %
              (S5, n5, float((obs <= 99).float().mean()), t_mse, t_d2m, R, t_mse + t_d2m), flush=True)
        ws = ops.d2m_compact(obs)
        M = obs.shape[0]
        t_c = bench.mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_compact(obs.data_ptr(), M, S5, S5, ws.data_ptr(), s), "compact"),
                                   stream, 20, 3, 3)
        print("   two-step: compact %d images %.1f us (%.1f MB workspace)" % (M, t_c, ws.numel() / 1e6), flush=True)
        for waves in (0, 8):
            ops.set_tuning(ops.TUNE_D2M_WAVES, waves)
            for parts in (1, 2, 4, 8):
                ls2 = torch.empty(n5 * parts, device=dev); gr2 = torch.empty(n5 * parts, J, 3, device=dev)
                t = bench.mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_from_points(ws.data_ptr(), M, a[1], a[2], 4, a[3], n5, J, S5, S5, parts,
                                                                                                ls2.data_ptr(), gr2.data_ptr(), s), "pts"), stream, 20, 3, 3)
                print("   two-step: search waves=%d parts=%d: %.1f us" % (waves, parts, t), flush=True)
        ops.set_tuning(ops.TUNE_D2M_WAVES, 0)
        for parts in (1, 2):
            ls2 = torch.empty(n5 * parts, device=dev)
            t = bench.mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_from_points(ws.data_ptr(), M, a[1], a[2], 4, a[3], n5, J, S5, S5, parts,
                                                                                            ls2.data_ptr(), None, s), "pts"), stream, 20, 3, 3)
            print("   two-step: search WITHOUT gradient parts=%d: %.1f us" % (parts, t), flush=True)
        for tiled in (1, 0):
            ops.set_tuning(ops.TUNE_D2M_TILED, tiled)
            t = bench.mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_partial(a[0], a[1], a[2], 4, a[3], n5, J, S5, S5, R,
                                                                                        a[4], a[5], s), "d2m"), stream, 20, 3, 3)
            print("   streaming kernel tiled=%d parts=%d: %.1f us" % (tiled, R, t), flush=True)
        ops.set_tuning(ops.TUNE_D2M_TILED, -1)
        for is_mv in (True, False):
            for two in (True, False):
                ops.D2M_TWO_STEP = two

                def mv_step():
                    joints.grad = None
                    loss, _ = crit(cam, inv, joints, real, is_mv)
                    loss.backward()
                t = bench.mean_launch_us(lambda _s: mv_step(), stream, 10, 3, 3)
                print("   MutualProjectionLoss fwd+bwd is_mv=%s two_step_d2m=%s: %.1f us" % (is_mv, two, t), flush=True)
        ops.D2M_TWO_STEP = True
        del ws
        del ds, crit, real, obs, dep, gsp, gr
