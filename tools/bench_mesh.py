"""Timings of the triangle path: lbs_project, mesh_depth (fused) and the explicit 640^2 chain."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model, ops
from spherehand_amd.render import DepthRender
from spherehand_amd.util_modules import HandSynthesizer
from spherehand_amd.joint_angle import sample_poses
mesh = hand_model.load_mesh()
def t_us(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for B, S in ((256, 128), (256, 64), (48, 64), (64, 256)):
    syn = HandSynthesizer(mesh, S, 16, 1.0, 0.01).cuda()
    T = syn.hand_skeleton_transform(sample_poses(B, seed=1).cuda())
    dr = DepthRender(mesh, S).cuda()
    with torch.no_grad():
        verts = dr.lbs(T, dr.camera, None)
    poses = sample_poses(B, seed=2).cuda()      # (drawn once: the sequential sampler takes 0.3 ms per pose on the host)
    print("B=%d S=%d: DepthRender %.1f us = lbs_project %.1f + mesh_depth %.1f ; HandSynthesizer %.1f us"
          % (B, S, t_us(lambda: dr(T)), t_us(lambda: dr.lbs(T, dr.camera, None)), t_us(lambda: dr.rasterizer(verts)),
             t_us(lambda: syn(poses), 10)))
