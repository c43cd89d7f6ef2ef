import ctypes, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.render import DepthRender
from spherehand_amd.util_modules import HandSynthesizer
from spherehand_amd.joint_angle import sample_poses
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpm.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.exp_mesh_launch.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
mesh = hand_model.load_mesh()
for B, S in ((256, 128), (48, 64)):
    syn = HandSynthesizer(mesh, S, 16, 1.0, 0.01).cuda()
    T = syn.hand_skeleton_transform(sample_poses(B, seed=1).cuda())
    dr = DepthRender(mesh, S).cuda()
    verts = dr.lbs(T, dr.camera, None).contiguous()
    faces = dr.rasterizer.faces_i32
    out = torch.empty(B, S, S, device="cuda")
    nwg = B * (1 if S == 128 else ((S + 63) // 64) ** 2)
    tb = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device="cuda")
    for _ in range(5):
        lib.exp_mesh_launch(verts.data_ptr(), faces.data_ptr(), B, verts.shape[1], faces.numel() // 3, 640, S, out.data_ptr(), tb.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = dr.rasterizer(verts)
    t = tb.cpu().numpy().reshape(nwg, 16, 8).astype(np.float64)
    print("B=%d S=%d max|diff| %.1e: per wave cycles (mean over WGs, then min/mean/max over waves)" % (B, S, (out - ref).abs().max().item()))
    for i, nm in enumerate(["phase A", "scan+sync", "phase B", "until epilogue", "total"]):
        m = t[:, :, i].mean(0)
        print("  %-15s %8.0f %8.0f %8.0f" % (nm, m.min(), m.mean(), m.max()))
    print("  work items per crop: mean %.0f max %.0f" % (t[:, 0, 5].mean(), t[:, 0, 5].max()))
