"""The literal drop-in: depth_rasterization.forward(640, 640, face_vertices[B,3382,3,3]) -- time and traffic bound."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import depth_rasterization
from spherehand_amd import hand_model, ops
from spherehand_amd.render import DepthRender
from spherehand_amd.util_modules import HandSynthesizer
from spherehand_amd.joint_angle import sample_poses
mesh = hand_model.load_mesh()
def t_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for B in (1, 48, 256):
    syn = HandSynthesizer(mesh, 128, 16, 1.0, 0.01).cuda()
    T = syn.hand_skeleton_transform(sample_poses(B, seed=1).cuda())
    dr = DepthRender(mesh, 128).cuda()
    with torch.no_grad():
        verts = dr.lbs(T, dr.camera, None)
        fv = verts[:, dr.rasterizer.faces, 0:3].reshape(B, -1, 3, 3).contiguous()
    t = t_us(lambda: depth_rasterization.forward(640, 640, fv))
    ref = depth_rasterization.forward(640, 640, fv)
    alt = ops.mesh_depth_fwd(verts.contiguous(), dr.rasterizer.faces_i32, 640, 640, float("inf"))
    print("   tile kernel at ratio 1: equal to the 3-pass kernel: %s (max |diff| %.3g); %.1f us"
          % (torch.equal(ref, alt), (ref - alt).abs().max().item(),
             t_us(lambda: ops.mesh_depth_fwd(verts.contiguous(), dr.rasterizer.faces_i32, 640, 640, float("inf")))))
    out_mb = B * 640 * 640 * 4 / 1e6
    print("B=%d: depth_rasterization.forward(640,640) %.1f us (%.0f crops/s); output %.1f MB -> one write pass at 8 TB/s = %.1f us; torch.full of the output %.1f us"
          % (B, t, B / t * 1e6, out_mb, out_mb / 8.0, t_us(lambda: torch.full((B, 640, 640), 1000.0, device="cuda"))))
