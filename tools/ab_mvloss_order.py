import gc, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
B, S = 128, 256
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crits = {"ordered": MutualProjectionLoss(S, mesh).cuda(), "batch order": MutualProjectionLoss(S, mesh).cuda()}
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
joints = (ds.joints.cuda() + torch.randn_like(ds.joints.cuda())).requires_grad_(True)
crits["batch order"]._indices(B, 3, cam.device); crits["batch order"]._order = crits["batch order"]._order_target = None
gc.collect(); gc.freeze()
for cache in (False, True):
    for rnd in range(3):
        for name, crit in crits.items():
            crit.cache_points = cache
            def step():
                joints.grad = None
                loss, _ = crit(cam, inv, joints, real, True)
                loss.backward()
            for _ in range(5): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): step()
            torch.cuda.synchronize()
            print("cache %d %-12s %.1f us" % (cache, name, (time.perf_counter() - t0) / 30 * 1e6), flush=True)
