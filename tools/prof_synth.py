import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hand_model
from spherehand_amd.util_modules import HandSynthesizer
from spherehand_amd.joint_angle import sample_poses
mesh = hand_model.load_mesh()
syn = HandSynthesizer(mesh, 64, 16, 1.0, 0.01).cuda()
p = sample_poses(48, seed=0).cuda()
for _ in range(5): syn(p)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): syn(p)
torch.cuda.synchronize(); print("HandSynthesizer B=48 S=64: %.1f us wall per call" % ((time.perf_counter() - t0) / 20 * 1e6))
