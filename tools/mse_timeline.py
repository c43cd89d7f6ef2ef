"""In-kernel timeline of the fused render-and-compare kernel at config 5's size (1152 crops @256x256, two workgroups per
CU): phases of the steady-state workgroups 512..767 of row region 1 (rows 64..127: the upper half of the hand), s_memtime inside a workgroup.  Needs
tools/libspherehand_tl.so (python tools/headline_timeline.py build)."""
import ctypes, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libspherehand_tl.so"))
vp, i = ctypes.c_void_p, ctypes.c_int
lib.shr_sphere_raster_mse.argtypes = [vp, i, i, i, i, vp, vp, vp, vp, vp, vp]
lib.shr_debug_timeline.argtypes = [i, vp, ctypes.c_size_t]
lib.shr_debug_timeline_rt.argtypes = [i, vp, ctypes.c_size_t]
dev = torch.device("cuda", 0)
mesh = hand_model.load_mesh()
B, S, J = 128, 256, 41
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0, device=dev)
crit = MutualProjectionLoss(S, mesh).to(dev)
with torch.no_grad():
    _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + torch.randn(ds.joints.shape, device=dev))
n = B * 9
obs = ds.dms.to(dev).view(B * 3, S, S).contiguous()
index = crit._indices(B, 3, dev)[0]
rad = crit.data_to_model_criterion.radiuses.view(-1)
sph = torch.cat([pts.squeeze(-1).reshape(n, J, 3), rad.view(1, J, 1).expand(n, J, 1)], -1).contiguous()
dep = torch.empty(n, S, S, device=dev); sse = torch.empty(n * 2, device=dev); gsp = torch.empty(n * 2, J, 4, device=dev)
stream = torch.cuda.Stream(device=dev)
f = lambda s: lib.shr_sphere_raster_mse(sph.data_ptr(), n, J, S, S, obs.data_ptr(), index.data_ptr(), dep.data_ptr(), sse.data_ptr(), gsp.data_ptr(), s)
with torch.cuda.stream(stream):
    assert f(stream.cuda_stream) == 0
    t_us = bench.mean_launch_us(f, stream, 20, 3, 3)
    rows, rts = [], []
    for rep in range(20):
        f(stream.cuda_stream)
        buf = np.zeros(256 * 16 * 8, np.uint64); rt = np.zeros(256 * 16 * 2, np.uint64)
        lib.shr_debug_timeline(2, buf.ctypes.data, buf.nbytes); lib.shr_debug_timeline_rt(2, rt.ctypes.data, rt.nbytes)
        rows.append(buf.reshape(256, 16, 8).astype(np.int64)); rts.append(rt.reshape(256, 16, 2).astype(np.int64))
a, rt = np.stack(rows), np.stack(rts)
e = a[..., 0].min(2)
rel = a - e[:, :, None, None]
med = lambda v: float(np.median(v))
life = rel[..., 7].max(2)
ghz = med(life) / med((rt[..., 1].max(2) - rt[..., 0].min(2)) * 10.0)
names = ["entry", "reaches barrier 1 (prologue work done)", "passes barrier 1", "scan slice done", "passes barrier 2", "convert units done",
         "passes barrier 3", "walk done"]
out = {"kernel_us (HIP events, instrumented build)": round(t_us, 1), "shader_clock_GHz": round(ghz, 3),
       "workgroup_lifetime_us_median": round(med(life) / ghz / 1e3, 2),
       "phase_boundaries_us_after_entry (median workgroup: slowest wave | fastest wave)":
           {names[k]: [round(med(rel[..., k].max(2)) / ghz / 1e3, 2), round(med(rel[..., k].min(2)) / ghz / 1e3, 2)] for k in range(1, 8)},
       "per_wave_us_reaches_barrier_1": [round(med(rel[:, :, w, 1]) / ghz / 1e3, 2) for w in range(16)],
       "per_wave_us_scan_done": [round(med(rel[:, :, w, 3]) / ghz / 1e3, 2) for w in range(16)],
       "per_wave_us_convert_done": [round(med(rel[:, :, w, 5]) / ghz / 1e3, 2) for w in range(16)],
       "per_wave_us_walk_done": [round(med(rel[:, :, w, 7]) / ghz / 1e3, 2) for w in range(16)]}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_mse_timeline.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
