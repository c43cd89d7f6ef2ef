import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpd.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_d2m_launch.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, vp]
mesh = hand_model.load_mesh()
B, S = 128, 128
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0)
crit = MutualProjectionLoss(S, mesh).cuda()
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
with torch.no_grad():
    _, pts = crit.mutual_projection(cam, inv, ds.joints.cuda() + 1.0)
N = B * 9
obs = real.unsqueeze(1).expand(B, 3, 3, S, S).reshape(N, S, S).contiguous()
cen = pts.squeeze(-1).reshape(N, 41, 3).contiguous()
rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
ls = torch.empty(N, device="cuda"); gr = torch.empty(N, 41, 3, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=30):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 1)
names = {0: "full", 1: "load+scan+compact only", 2: "no stage-2 candidates", 18: "box phase only (no seeds, no candidates)", 4: "no owner reduction", 22: "box phase only, no reduction"}
print({names[m]: timeit(lambda: lib.exp_d2m_launch(obs.data_ptr(), cen.data_ptr(), rad.data_ptr(), N, 41, S, S, ls.data_ptr(), gr.data_ptr(), m, st)) for m in names})
