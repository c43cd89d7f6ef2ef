"""1x1 convolutions of the hourglass as plain GEMMs on the channels-last view (rocBLAS / hipBLASLt) against MIOpen."""
import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import hourglass
from spherehand_amd.ops import group_norm_relu
def T(fn, reps=10):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def lin1x1(conv, x):
    y = F.linear(x.permute(0, 2, 3, 1), conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)
    return y.permute(0, 3, 1, 2)
def fwd_gemm(self, x):
    y = lin1x1(self.conv1, group_norm_relu(x, self.bn1))
    y = self.conv2(group_norm_relu(y, self.bn2))
    y = lin1x1(self.conv3, group_norm_relu(y, self.bn3))
    return y + (x if self.downsample is None else self.downsample(x))
orig = hourglass.Bottleneck.forward
torch.manual_seed(0)
net = hourglass.create_hourglass_network(82, 1).cuda().to(memory_format=torch.channels_last)
x = torch.rand(123, 1, 64, 64, device="cuda")
def step():
    net.zero_grad(set_to_none=True)
    out, _ = net(x); out[0].square().mean().backward()
for name, f in (("MIOpen 1x1", orig), ("GEMM 1x1", fwd_gemm), ("MIOpen 1x1", orig), ("GEMM 1x1", fwd_gemm)):
    hourglass.Bottleneck.forward = f
    o = net(x)[0][0]
    print("%-11s fwd+bwd %.2f ms   out checksum %.6f  channels_last out: %s" % (name, T(step), o.double().sum().item(), o.is_contiguous(memory_format=torch.channels_last)))
