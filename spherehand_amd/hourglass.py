"""Stacked-hourglass heat-map regressor (stays on PyTorch-ROCm / MIOpen: the
north star keeps the CNN out of the hand-written path).

Architecture and parameter names follow the reference's network/hourglass.py
(:7-41 pre-activation GroupNorm bottleneck, :44-85 depth-2 hourglass, :88-173
stem + stacks) so that its checkpoints (`network_state_dict` with keys
`hg.conv1.weight`, `hg.hg.0.hg.1.2.0.bn1.weight`, ...) load unchanged.
1 stack, 82 outputs: 2,308,946 parameters.
"""
import torch.nn as nn
import torch.nn.functional as F

from .ops import conv_then_group_norm_relu, group_norm_relu


class Bottleneck(nn.Module):
    """GN-ReLU-1x1 -> GN-ReLU-3x3 -> GN-ReLU-1x1 (x2 channels) + skip."""
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.GroupNorm(16, inplanes)
        self.conv1 = nn.Conv2d(inplanes, planes, 1)
        self.bn2 = nn.GroupNorm(16, planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1)
        self.bn3 = nn.GroupNorm(16, planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1)
        self.downsample = downsample

    def forward(self, x):
        # NHWC GroupNorm+ReLU kernels on the GPU path; conv1's / conv2's bias rides in the kernel that normalises
        # their output (no bias-add pass, the bias gradient comes back with dx)
        y = group_norm_relu(x, self.bn1)
        y = conv_then_group_norm_relu(y, self.conv1, self.bn2)
        y = conv_then_group_norm_relu(y, self.conv2, self.bn3)
        y = self.conv3(y)
        return y + (x if self.downsample is None else self.downsample(x))


def _chain(block, planes, count):
    return nn.Sequential(*[block(planes * block.expansion, planes) for _ in range(count)])


class Hourglass(nn.Module):
    """Recursive encoder/decoder; level n: skip branch [0], down [1], up [2];
    the innermost level owns the bottleneck chain [3].  Returns (features,
    innermost latent)."""

    def __init__(self, block, num_blocks, planes, depth):
        super().__init__()
        self.depth = depth
        self.hg = nn.ModuleList([
            nn.ModuleList([_chain(block, planes, num_blocks) for _ in range(4 if level == 0 else 3)])
            for level in range(depth)])

    def _level(self, n, x):
        branches = self.hg[n - 1]
        skip = branches[0](x)
        low = branches[1](F.max_pool2d(x, 2, stride=2))
        if n > 1:
            low, latent = self._level(n - 1, low)
        else:
            low = branches[3](low)
            latent = low
        up = F.interpolate(branches[2](low), scale_factor=2, mode='bilinear', align_corners=False)
        return skip + up, latent

    def forward(self, x):
        return self._level(self.depth, x)


class HourglassNet(nn.Module):
    def __init__(self, block, num_stacks, num_blocks, num_outputs):
        super().__init__()
        self.inplanes, self.num_feats, self.num_stacks = 64, 128, num_stacks
        self.conv1 = nn.Conv2d(1, self.inplanes, 5, stride=2, padding=2)
        self.bn1 = nn.GroupNorm(4, self.inplanes)
        self.layer1 = self._stage(block, self.inplanes, 1)
        self.layer2 = self._stage(block, self.inplanes, 1)
        self.layer3 = self._stage(block, self.num_feats, 1)
        ch = self.num_feats * block.expansion
        self.hg = nn.ModuleList([Hourglass(block, num_blocks, self.num_feats, 2) for _ in range(num_stacks)])
        self.res = nn.ModuleList([self._stage(block, self.num_feats, num_blocks) for _ in range(num_stacks)])
        self.fc = nn.ModuleList([nn.Sequential(nn.Conv2d(ch, ch, 1), nn.GroupNorm(16, ch), nn.ReLU(inplace=True))
                                 for _ in range(num_stacks)])
        self.score = nn.ModuleList([nn.Conv2d(ch, num_outputs, 1) for _ in range(num_stacks)])
        self.fc_ = nn.ModuleList([nn.Conv2d(ch, ch, 1) for _ in range(num_stacks - 1)])
        self.score_ = nn.ModuleList([nn.Conv2d(num_outputs, ch, 1) for _ in range(num_stacks - 1)])

    def _stage(self, block, planes, blocks):
        out = planes * block.expansion
        down = nn.Sequential(nn.Conv2d(self.inplanes, out, 1)) if self.inplanes != out else None
        layers = [block(self.inplanes, planes, 1, down)] + [block(out, planes) for _ in range(1, blocks)]
        self.inplanes = out
        return nn.Sequential(*layers)

    def forward(self, x):
        """x [N,S,S] or [N,1,S,S] -> ([scores [N,num_outputs,S/4,S/4]] per stack, [latent] per stack)."""
        if x.dim() == 3:
            x = x.unsqueeze(1)
        x = self.layer1(conv_then_group_norm_relu(x, self.conv1, self.bn1))
        x = self.layer3(self.layer2(F.max_pool2d(x, 2, stride=2)))
        out, latents = [], []
        for i in range(self.num_stacks):
            y, latent = self.hg[i](x)
            y = conv_then_group_norm_relu(self.res[i](y), self.fc[i][0], self.fc[i][1])   # fc = Conv-GN-ReLU (keys unchanged)
            score = self.score[i](y)
            out.append(score)
            latents.append(latent)
            if i < self.num_stacks - 1:
                x = x + self.fc_[i](y) + self.score_[i](score)
        return out, latents


def create_hourglass_network(num_outputs, num_stacks=1):
    return HourglassNet(Bottleneck, num_stacks=num_stacks, num_blocks=1, num_outputs=num_outputs)
