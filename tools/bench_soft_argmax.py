"""soft-argmax read-out kernels on the step's shape (123 samples, 41 key-points, 16 x 16 maps, channels-last)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spherehand_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
N, J, h = 123, 41, 16
hm = torch.randn(N, 2 * J, h, h, device=dev).contiguous(memory_format=torch.channels_last)
xyz = torch.empty(N, J, 3, device=dev); g = torch.randn(N, J, 3, device=dev); ghm = torch.empty_like(hm)
sn, sc, sh, sw = hm.stride()
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    f = bench.mean_launch_us(lambda s: lib.shr_soft_argmax_fwd(hm.data_ptr(), sn, sc, sw, N, J, h, h, 8.0, 8.0, 16 / 300.0, 16 / 300.0, 100.0, xyz.data_ptr(), s), stream, 50, 5, 5, warm_ms=10.0)
    b = bench.mean_launch_us(lambda s: lib.shr_soft_argmax_bwd(hm.data_ptr(), sn, sc, sw, N, J, h, h, 8.0, 8.0, 16 / 300.0, 16 / 300.0, 100.0, g.data_ptr(), ghm.data_ptr(), s), stream, 50, 5, 5, warm_ms=10.0)
print("soft-argmax forward %.1f us, backward %.1f us" % (f, b))
