"""Frozen palm re-predictor applied before the evaluation metric -- the reference's
network/pose_denoiser.py (PoseDenoiser :21-81), state-dict compatible
(`input_indices`, `output_indices`, `network.{0,1,3,4,6}.*`).

The reference scores NYU with `pose_denoiser(result['real_xyz'][-1][:, 0])`
(network/engine.py:200-206): the 11 palm spheres (33 coordinates) are re-predicted
from the 30 finger spheres' xyz and the palm spheres' xy.  A 112-256-256-33 MLP on
a handful of samples at eval time: torch ops (rocBLAS), not a kernel of this path.

Weights: ``spherehand_amd/data/pose_denoiser.npz`` is a plain re-export (data only,
tests/golden/make_goldens_priors.py) of the reference's mesh/model/pose_denoiser.pth,
which the reference's Engine loads at construction (network/engine.py:71-73).
"""
import os

import numpy as np
import torch
import torch.nn as nn

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DEFAULT_WEIGHTS = os.path.join(_DATA, "pose_denoiser.npz")

KEY_POINTS = list(range(11))                     # palm spheres: re-predicted (pose_denoiser.py:12)
INPUT_3D_POINTS = list(range(11, 41))            # finger spheres: x, y, z are inputs (:13)
INPUT_2D_POINTS = list(range(11))                # palm spheres: x, y are inputs (:14)
INPUT_INDICES = ([3 * i for i in INPUT_3D_POINTS] + [3 * i + 1 for i in INPUT_3D_POINTS] +
                 [3 * i + 2 for i in INPUT_3D_POINTS] +
                 [3 * i for i in INPUT_2D_POINTS] + [3 * i + 1 for i in INPUT_2D_POINTS])
OUTPUT_INDICES = [3 * p + c for p in KEY_POINTS for c in range(3)]


def load_npz_state_dict(module, path):
    """Load a state dict stored as an .npz of arrays keyed by the state-dict names."""
    with np.load(path) as z:
        sd = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    module.load_state_dict(sd)
    for p in module.parameters():
        p.requires_grad = False
    return module


class PoseDenoiser(nn.Module):
    """forward(fea [B,41,3] or [B,123]) -> same shape with the palm coordinates replaced by the
    MLP's prediction (millimetres in, millimetres out; the MLP works in units of 100 mm).
    `model_path`: an .npz re-export, a reference .pth ({'network_state_dict': ...}), or None for
    random weights.  Loading freezes the parameters, like the reference (:41-44)."""

    def __init__(self, input_indices=INPUT_INDICES, output_indices=OUTPUT_INDICES, model_path=None):
        super().__init__()
        self.input_fea, self.output_fea = len(input_indices), len(output_indices)
        self.scale_factor = 0.01
        self.register_buffer('input_indices', torch.tensor(input_indices).long())
        self.register_buffer('output_indices', torch.tensor(output_indices).long())
        self.network = nn.Sequential(
            nn.Linear(self.input_fea, 256), nn.GroupNorm(16, 256), nn.ReLU(),
            nn.Linear(256, 256), nn.GroupNorm(16, 256), nn.ReLU(),
            nn.Linear(256, self.output_fea))
        self.criterion = nn.MSELoss()
        if model_path is not None:
            if str(model_path).endswith('.npz'):
                load_npz_state_dict(self, model_path)
            else:
                self.load_state_dict(torch.load(model_path, map_location='cpu')['network_state_dict'])
                for p in self.parameters():
                    p.requires_grad = False

    def forward(self, fea):
        is_skel = fea.ndimension() == 3
        if is_skel:
            num_batch, num_joints = fea.shape[0], fea.shape[1]
            fea = fea.reshape(num_batch, -1)
        input_fea = fea[:, self.input_indices] * self.scale_factor
        if self.training:
            input_fea = input_fea + torch.randn_like(input_fea) * 0.1
        output_fea = self.network(input_fea) / self.scale_factor
        denoised = fea.clone()
        denoised[:, self.output_indices] = output_fea
        return denoised.reshape(num_batch, num_joints, 3) if is_skel else denoised

    def loss(self, gt_fea, est_fea):
        n = gt_fea.shape[0]
        return self.criterion(gt_fea.reshape(n, -1)[:, self.output_indices],
                              est_fea.reshape(n, -1)[:, self.output_indices])


def default_pose_denoiser():
    """The denoiser the reference's Engine builds (mesh/model/pose_denoiser.pth), in eval mode."""
    if not os.path.exists(DEFAULT_WEIGHTS):
        raise FileNotFoundError('skipped: asset missing ({})'.format(DEFAULT_WEIGHTS))
    return PoseDenoiser(model_path=DEFAULT_WEIGHTS).eval()
