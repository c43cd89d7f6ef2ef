"""HandSynthesizer: eager, one hipGraph, kernel by kernel (HIP events on the launching stream).
    python tools/bench_synth.py [B S hm]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from spherehand_amd import hand_model
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.util_modules import HandSynthesizer
from bench import mean_launch_us

B, S, hm = (int(v) for v in (sys.argv[1:4] + [256, 128, 16][len(sys.argv) - 1:]))
mesh = hand_model.load_mesh()
dev = torch.device("cuda")
stream = torch.cuda.Stream()
pose = sample_poses(B, seed=0).to(dev)
for fused in (2, 1, 0):
    for noise, heat in ((True, True), (False, True), (True, False)):
        syn = HandSynthesizer(mesh, S, hm, 1.0, 0.01, add_noise=noise, out_heatmap=heat).to(dev)
        syn.fused = fused > 0
        syn.one_launch = fused == 2
        with torch.cuda.stream(stream):
            syn(pose); stream.synchronize()
            eager = sorted(mean_launch_us(lambda _s: syn(pose), stream, 50, 1, 20 if i == 0 else 0, warm_ms=400.0 if i == 0 else 0.0) for i in range(7))
            graph_us = None
            if fused:
                for _ in range(3):
                    syn(pose)
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    syn(pose)
                graph_us = mean_launch_us(lambda _s: g.replay(), stream, 100, 3, 10)
        print("B=%d S=%d hm=%d fused=%d noise=%d heatmaps=%d: eager %.1f us (fastest %.1f, slowest %.1f)%s" %
              (B, S, hm, fused, noise, heat, eager[3], eager[0], eager[-1], "" if graph_us is None else " | one hipGraph %.1f us" % graph_us), flush=True)
