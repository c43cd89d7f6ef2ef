"""Which part of the synthetic-branch step is not run-to-run deterministic?  Same inputs, same weights,
repeated forward+backward in one process; compare the gradients."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import ops
from spherehand_amd.criterion import HeatmapEstimationNetwork

def grads(net, x, tgt, ztgt):
    net.zero_grad(set_to_none=True)
    r = net(synt_dms=x)
    loss = 1e3 * torch.nn.functional.mse_loss(r['synt_uv_hms'][0], tgt) + 1e-1 * torch.nn.functional.mse_loss(r['synt_xyz'][0][:, :, 2], ztgt)
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters()}, float(loss)

def probe(tag, cl, fused, det, n=12):
    torch.manual_seed(7)
    net = HeatmapEstimationNetwork(16, 0.01, 41, 1).cuda().train()
    if cl:
        net = net.to(memory_format=torch.channels_last)
    ops.FUSED_GROUP_NORM_RELU = fused
    torch.backends.cudnn.deterministic = det
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.rand(n, 64, 64, device='cuda', generator=g)
    tgt = torch.rand(n, 41, 16, 16, device='cuda', generator=g)
    ztgt = torch.randn(n, 41, device='cuda', generator=g) * 20
    a, la = grads(net, x, tgt, ztgt)
    worst = 0.0; wk = None
    for _ in range(4):
        b, lb = grads(net, x, tgt, ztgt)
        for k in a:
            d = (a[k] - b[k]).abs().max().item()
            if d > worst:
                worst, wk = d, k
    gmax = max(v.abs().max().item() for v in a.values())
    print("%-40s gmax %.4g worst %.4g (%s) loss %.6f/%.6f" % (tag, gmax, worst, wk, la, lb), flush=True)

probe("NCHW torch-GN", False, False, False)
probe("NCHW torch-GN deterministic", False, False, True)
probe("NHWC torch-GN", True, False, False)
probe("NHWC fused-GN", True, True, False)
probe("NHWC fused-GN deterministic", True, True, True)
probe("NHWC fused-GN n=18", True, True, False, 18)
probe("NHWC fused-GN n=36", True, True, False, 36)
