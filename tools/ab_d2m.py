"""A/B of the two-step data->model kernels (compaction, point search) at config 5's size between the product library and the
variant libraries tools/libspherehand_exp_*.so (SRC=data_to_model python tools/ab_variant.py build:NAME -D...)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from spherehand_amd import _lib, hand_model, ops
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
vp, i = ctypes.c_void_p, ctypes.c_int
libs = {"product": _lib.lib()}
for f in sorted(glob.glob(os.path.join(ROOT, "tools", "libspherehand_exp*.so"))):
    libs[os.path.basename(f)[len("libspherehand_exp"):-3].lstrip("_") or "variant"] = ctypes.CDLL(f)
for l in libs.values():
    l.shr_data_to_model_compact.argtypes = [vp, i, i, i, vp, vp]
    l.shr_data_to_model_from_points.argtypes = [vp, i, vp, vp, i, vp, i, i, i, i, i, vp, vp, vp]
dev = torch.device("cuda", 0)
mesh = hand_model.load_mesh()
stream = torch.cuda.Stream(device=dev)
J = 41
for B5, S5 in ((128, 256), (128, 128)):
    ds = SyntheticMultiviewDataset(mesh, B5, S5, seed=0, device=dev)
    crit = MutualProjectionLoss(S5, mesh).to(dev)
    n5 = B5 * 9
    with torch.no_grad():
        _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + torch.randn(ds.joints.shape, device=dev))
    obs = ds.dms.to(dev).view(B5 * 3, S5, S5).contiguous()
    index = crit._indices(B5, 3, dev)[0]
    cen = pts.squeeze(-1).reshape(n5, J, 3).contiguous()
    rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
    ws = ops.d2m_compact(obs)
    ls = torch.empty(n5, device=dev); gr = torch.empty(n5, J, 3, device=dev)
    ref = None
    with torch.cuda.stream(stream):
        for rnd in range(3):
            for name, l in libs.items():
                c = lambda s: l.shr_data_to_model_compact(obs.data_ptr(), B5 * 3, S5, S5, ws.data_ptr(), s)
                p = lambda s: l.shr_data_to_model_from_points(ws.data_ptr(), B5 * 3, index.data_ptr(), cen.data_ptr(), 3, rad.data_ptr(), n5, J, S5, S5, 1,
                                                              ls.data_ptr(), gr.data_ptr(), s)
                assert c(stream.cuda_stream) == 0 and p(stream.cuda_stream) == 0
                tc = bench.mean_launch_us(c, stream, 20, 3, 3, warm_ms=20.0)
                tp = bench.mean_launch_us(p, stream, 20, 3, 3, warm_ms=20.0)
                stream.synchronize()
                key = (ls.clone(), gr.clone())
                if ref is None: ref = key
                print("%4d crops @%d %-10s compaction %6.2f us  point search %6.2f us  same bits: %s" %
                      (n5, S5, name, tc, tp, torch.equal(key[0], ref[0]) and torch.equal(key[1], ref[1])), flush=True)
