"""Forward kinematics: 26 pose parameters -> 17 bone transforms [B,17,4,4].

Same classes/signatures as the reference's mesh/kinematicsTransformation.py
(HandTransformationMat :157-177, SkeletonFK :180-207), evaluated batched over
the five fingers instead of module-per-joint.
"""
import numpy as np
import torch
import torch.nn as nn

from .pointTransformation import LinearBlendSkinning, RandScale


def axis_rotation(axis, angles):
    """Rodrigues rotation about `axis` [...,3] by `angles` [...] -> [...,4,4]
    (mesh/kinematicsTransformation.py:29-54)."""
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    c, s = torch.cos(angles), torch.sin(angles)
    i = 1 - c
    o, l = torch.zeros_like(c), torch.ones_like(c)
    rows = [x * x * i + c, x * y * i - z * s, x * z * i + y * s, o,
            x * y * i + z * s, y * y * i + c, y * z * i - x * s, o,
            x * z * i - y * s, y * z * i + x * s, z * z * i + c, o,
            o, o, o, l]
    return torch.stack(rows, dim=-1).view(*c.shape, 4, 4)


class HandTransformationMat(nn.Module):
    """params [B,26] -> T [B,17,4,4].  Bone order: 0 metacarpals, 1 carpals (both
    the palm transform, :153-155), then 3 bones per finger for fingers 0..4 which
    read params [6+4k : 10+4k] = (abduct, flex1, flex2, flex3) (:165-175)."""

    def __init__(self, offset_mats):
        super().__init__()
        off = torch.from_numpy(np.stack([np.asarray(m, np.float32) for m in offset_mats]))
        self.register_buffer('offset', off)                          # [17,4,4]
        self.register_buffer('offset_inv', torch.inverse(off).contiguous())  # :87
        ab = torch.tensor([[0, 0, 1], [0, 0, 1], [0, -1, 0], [0, -1, 0], [0, 0, 1]], dtype=torch.float32)
        self.register_buffer('abduct_axis', ab)                      # :162-164
        self.register_buffer('axes', torch.eye(3))

    def forward(self, parameters):
        # one HIP launch forward, one backward (forward_torch below is ~60 torch ops)
        if not parameters.is_cuda:
            raise RuntimeError("parameters must be a CUDA tensor: forward kinematics runs on the HIP kernel "
                               "(forward_torch() is the explicit torch-op evaluation for host-side fixtures)")
        from . import ops
        return ops.ForwardKinematics.apply(parameters, self.offset, self.offset_inv)

    def forward_torch(self, parameters):
        """The same map written with torch ops (any device, autograd by torch):
        fixture preparation and the independent check of the HIP kernels."""
        p = parameters
        B = p.shape[0]
        ex, ey, ez = self.axes[0], self.axes[1], self.axes[2]
        rot = axis_rotation(ez, p[:, 2]) @ (axis_rotation(ey, p[:, 1]) @ axis_rotation(ex, p[:, 0]))
        trans = torch.eye(4, dtype=p.dtype, device=p.device).repeat(B, 1, 1)
        trans = torch.cat([trans[:, :, :3], torch.cat([p[:, 3:6], p.new_ones(B, 1)], 1).unsqueeze(-1)], dim=2)
        palm = trans @ rot                                           # :148-152
        a = p[:, 6:26].view(B, 5, 4)
        off = self.offset[2:17].view(5, 3, 4, 4)
        inv = self.offset_inv[2:17].view(5, 3, 4, 4)
        local1 = axis_rotation(self.abduct_axis.unsqueeze(0).expand(B, 5, 3), a[:, :, 0]) @ \
            axis_rotation(ex.expand(B, 5, 3), a[:, :, 1])            # :119, :96-103
        local2 = axis_rotation(ex.expand(B, 5, 3), a[:, :, 2])
        local3 = axis_rotation(ex.expand(B, 5, 3), a[:, :, 3])
        g1 = palm.unsqueeze(1) @ ((inv[:, 0] @ local1) @ off[:, 0])   # :108-111
        g2 = g1 @ ((inv[:, 1] @ local2) @ off[:, 1])
        g3 = g2 @ ((inv[:, 2] @ local3) @ off[:, 2])
        fingers = torch.stack([g1, g2, g3], dim=2).view(B, 15, 4, 4)
        return torch.cat([palm.unsqueeze(1), palm.unsqueeze(1), fingers], dim=1)


def keypoint_skinning(mesh):
    """LinearBlendSkinning of the 41 sphere centres, each bound to one bone with
    weight 1 (mesh/render.py:65-77, mesh/kinematicsTransformation.py:189-202)."""
    vertices, weights, indices = [], [], []
    for bone in (mesh['bones'] if isinstance(mesh, dict) else mesh):
        weights.append([])
        indices.append([])
        for pt, _ in bone.get('keypoint', []):
            vertices.append(np.asarray([pt[0], pt[1], pt[2], 1.0], np.float32))
            weights[-1].append(1.0)
            indices[-1].append(len(vertices) - 1)
    return LinearBlendSkinning(np.asarray(vertices, np.float32), weights, indices)


class SkeletonFK(nn.Module):
    """mesh/kinematicsTransformation.py:180-207: pose -> (randomly scaled)
    sphere centres [B,41,4]."""

    def __init__(self, mesh):
        super().__init__()
        self.hand_skeleton_transform = HandTransformationMat(
            [b['offset_matrix'].astype(np.float32) for b in mesh['bones']])
        self.rand_scale = RandScale(0.2)
        self.lbs = keypoint_skinning(mesh)
        self.num_vertices = self.lbs.num_vertices

    def forward(self, para):
        return self.lbs(self.rand_scale(self.hand_skeleton_transform(para)))
