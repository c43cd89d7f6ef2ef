"""The rasterizer kernels at config 5's own size (1152 crops @256x256) launched >= 1000 times each for a rocprofv3
--kernel-trace --stats summary: depth-only forward, forward + owner bytes on the touched rows, backward
(bench.config5_size_kernels with more repetitions)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib, hand_model
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    print(json.dumps(bench.config5_size_kernels(_lib.lib(), _lib, dev, stream, hand_model.load_mesh(), reps=int(os.environ.get("REPS", 125)), batches=8)))
