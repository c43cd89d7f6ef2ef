"""NHWC GroupNorm+ReLU kernels (network/hourglass.py:28-31 pairs) vs torch's group_norm + relu."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C,H,W,G", [(5, 64, 32, 32, 4), (5, 64, 32, 32, 16), (7, 128, 16, 16, 16), (3, 128, 4, 4, 16),
                                       (4, 256, 8, 8, 16), (2, 256, 16, 16, 8), (3, 32, 5, 7, 8)])
def test_group_norm_relu_matches_torch(N, C, H, W, G):
    from spherehand_amd import ops
    g = torch.Generator().manual_seed(N * C + G)
    x = (torch.randn(N, C, H, W, generator=g) * 3 + 1).cuda().to(memory_format=torch.channels_last).requires_grad_(True)
    gn = torch.nn.GroupNorm(G, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g)); gn.bias.copy_(torch.randn(C, generator=g) * 0.5)
    up = torch.randn(N, C, H, W, generator=g).cuda().to(memory_format=torch.channels_last)
    assert ops.group_norm_relu_supported(x, G)
    y = ops.group_norm_relu(x, gn)
    assert y.is_contiguous(memory_format=torch.channels_last)
    (y * up).sum().backward()
    got = (y.detach(), x.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone())
    x.grad = None; gn.weight.grad = None; gn.bias.grad = None
    xd = x.detach().double().requires_grad_(True)
    gd = torch.nn.GroupNorm(G, C).cuda().double()
    gd.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
    yr = torch.relu(gd(xd))
    (yr * up.double()).sum().backward()
    ref = (yr.detach(), xd.grad, gd.weight.grad, gd.bias.grad)
    for a, b, tol in zip(got, ref, (2e-6, 2e-5, 2e-5, 2e-5)):
        assert (a.double() - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def test_group_norm_relu_falls_back_outside_its_shapes():
    from spherehand_amd import ops
    x = torch.randn(2, 48, 8, 8).cuda()                                    # NCHW, C % 32 != 0
    gn = torch.nn.GroupNorm(4, 48).cuda()
    assert not ops.group_norm_relu_supported(x, 4)
    assert torch.equal(ops.group_norm_relu(x, gn), torch.relu(gn(x)))


def test_hourglass_gpu_matches_cpu_reference_path():
    """The network with the kernels (GPU, channels-last) against itself on torch ops only (CPU)."""
    from spherehand_amd.hourglass import create_hourglass_network
    torch.manual_seed(0)
    net = create_hourglass_network(82, 1)
    x = torch.randn(3, 1, 64, 64)
    ref, _ = net(x)
    out, _ = net.cuda().to(memory_format=torch.channels_last)(x.cuda().to(memory_format=torch.channels_last))
    assert (out[0].cpu() - ref[0]).abs().max().item() <= 2e-4 * ref[0].abs().max().item()


@pytest.mark.parametrize("layout", ["nchw", "nhwc", "nhwc_slice"])
def test_soft_argmax_kernels_match_the_torch_formulation(layout):
    """RecoverXYZCoordinateFromHeatmap: one launch per direction vs the torch ops (fp64 reference)."""
    from spherehand_amd import ops
    from spherehand_amd.util_modules import RecoverXYZCoordinateFromHeatmap
    g = torch.Generator().manual_seed(5)
    N, J, S = 9, 41, 16
    base = torch.randn(N + 3, 2 * J, S, S, generator=g) * 0.6
    base[:, :J] += torch.exp(-((torch.arange(S).view(1, 1, S, 1) - 7.3) ** 2 + (torch.arange(S).view(1, 1, 1, S) - 4.6) ** 2) / 3)
    base = base.cuda()
    if layout != "nchw":
        base = base.to(memory_format=torch.channels_last)
    hm = (base[2:2 + N] if layout == "nhwc_slice" else base[:N]).detach().requires_grad_(True)
    rec = RecoverXYZCoordinateFromHeatmap(S, S, 0.01).cuda()
    assert ops.soft_argmax_supported(hm, J)
    up = torch.randn(N, J, 3, generator=g).cuda()
    xyz = rec.from_output(hm)
    (xyz * up).sum().backward()
    got_x, got_g = xyz.detach(), hm.grad.clone()
    hd = hm.detach().double().requires_grad_(True)
    rd = RecoverXYZCoordinateFromHeatmap(S, S, 0.01).cuda().double()
    ref = rd.forward(hd[:, :J], hd[:, J:])
    (ref * up.double()).sum().backward()
    assert (got_x.double() - ref.detach()).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert (got_g.double() - hd.grad).abs().max().item() <= 2e-5 * max(1.0, hd.grad.abs().max().item())


def test_soft_argmax_sharply_peaked_maps_keep_their_gradient_precision():
    """Logits x 20 spanning hundreds (a nearly one-hot softmax): x - u must not be formed against the absolute
    expectation (an ulp of u times 20 p du was 4e-4 of the largest gradient entry); bar = the torch ops' own fp32
    distance from fp64 on the same inputs, x4, plus 4e-5."""
    from spherehand_amd import ops
    from spherehand_amd.util_modules import RecoverXYZCoordinateFromHeatmap
    g = torch.Generator().manual_seed(11)
    worst = 0.0
    for N, J, S in ((6, 2, 32), (5, 1, 32), (8, 2, 16), (3, 41, 16)):
        hm = (torch.randn(N, 2 * J, S, S, generator=g) * 3.0).cuda().requires_grad_(True)
        assert ops.soft_argmax_supported(hm, J)
        rec = RecoverXYZCoordinateFromHeatmap(S, S, 0.01).cuda()
        up = torch.randn(N, J, 3, generator=g).cuda()
        (rec.from_output(hm) * up).sum().backward()
        h32 = hm.detach().clone().requires_grad_(True)
        (rec.forward(h32[:, :J], h32[:, J:]) * up).sum().backward()
        hd = hm.detach().double().requires_grad_(True)
        (RecoverXYZCoordinateFromHeatmap(S, S, 0.01).cuda().double().forward(hd[:, :J], hd[:, J:]) * up.double()).sum().backward()
        err = (hm.grad.double() - hd.grad).abs().max().item()
        err_t = (h32.grad.double() - hd.grad).abs().max().item()
        assert err <= 4e-5 * max(1.0, hd.grad.abs().max().item()) + 4 * err_t, (N, J, S, err, err_t)
        worst = max(worst, err / max(1.0, hd.grad.abs().max().item()))
    assert worst < 2e-4


@pytest.mark.parametrize("N,Cin,C,H,W,G,k", [(5, 32, 64, 16, 16, 16, 1), (3, 64, 128, 8, 8, 16, 3), (4, 1, 64, 32, 32, 4, 5)])
def test_conv_bias_folded_into_group_norm_relu(N, Cin, C, H, W, G, k):
    """relu(gn(conv(x))) with the convolution's bias folded into the normalisation kernel (forward: x + bias on the
    fly; backward: the bias gradient = per-channel sum of dx, returned by the same kernel) against the plain
    torch composition in fp64: outputs, dx, and the gradients of the conv weight, conv BIAS, gamma and beta."""
    from spherehand_amd import ops
    g = torch.Generator().manual_seed(N + C + k)
    conv = torch.nn.Conv2d(Cin, C, k, padding=k // 2).cuda().to(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(G, C).cuda()
    with torch.no_grad():
        conv.bias.copy_(torch.randn(C, generator=g) * 2)
        gn.weight.copy_(torch.randn(C, generator=g)); gn.bias.copy_(torch.randn(C, generator=g) * 0.5)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().to(memory_format=torch.channels_last).requires_grad_(True)
    up = torch.randn(N, C, H, W, generator=g).cuda().to(memory_format=torch.channels_last)
    y = ops.conv_then_group_norm_relu(x, conv, gn)
    (y * up).sum().backward()
    got = [y.detach(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), gn.weight.grad.clone(),
           gn.bias.grad.clone()]
    cd, gd = torch.nn.Conv2d(Cin, C, k, padding=k // 2).cuda().double(), torch.nn.GroupNorm(G, C).cuda().double()
    cd.load_state_dict({n_: v.double() for n_, v in conv.state_dict().items()})
    gd.load_state_dict({n_: v.double() for n_, v in gn.state_dict().items()})
    xd = x.detach().double().contiguous().requires_grad_(True)
    yr = torch.relu(gd(cd(xd)))
    (yr * up.double()).sum().backward()
    ref = [yr.detach(), xd.grad, cd.weight.grad, cd.bias.grad, gd.weight.grad, gd.bias.grad]
    for a, b, tol, name in zip(got, ref, (5e-6, 5e-5, 5e-5, 5e-5, 5e-5, 5e-5), ("y", "dx", "dW", "dbias", "dgamma", "dbeta")):
        assert (a.double() - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), name


def test_hourglass_gradients_with_and_without_the_fused_kernels():
    """The whole network, backward included: fused GroupNorm+ReLU (+ folded conv biases) vs torch ops only, on the GPU."""
    from spherehand_amd import ops
    from spherehand_amd.hourglass import create_hourglass_network
    torch.manual_seed(1)
    net = create_hourglass_network(82, 1).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(6, 1, 64, 64).cuda().to(memory_format=torch.channels_last)
    grads = []
    for fused in (True, False):
        ops.FUSED_GROUP_NORM_RELU = fused
        net.zero_grad(set_to_none=True)
        out, _ = net(x)
        out[0].square().mean().backward()
        grads.append({k: p.grad.clone() for k, p in net.named_parameters()})
    ops.FUSED_GROUP_NORM_RELU = True
    gmax = max(v.abs().max().item() for v in grads[1].values())
    for k in grads[1]:
        assert (grads[0][k] - grads[1][k]).abs().max().item() <= 2e-4 * gmax + 1e-7, k
    assert any(k.endswith("conv1.bias") for k in grads[0])
