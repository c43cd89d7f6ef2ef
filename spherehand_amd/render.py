"""Differentiable renderers and render losses -- the module-level API of the
reference's mesh/render.py (same class names, constructor arguments, forward
signatures and return values), running on the hand-written HIP kernels.

    BallRender               mesh/render.py:10-53
    HandBallPrimitiveRender  mesh/render.py:56-90
    DataToModelLoss          mesh/render.py:93-142
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .hand_model import radii_of
from .kinematicsTransformation import keypoint_skinning


class BallRender(nn.Module):
    """forward(xyz_centers[N,>=3], radiuses[N]) -> [N,H,W]: one orthographic
    front-surface depth map per sphere, background 100 (mesh/render.py:26-53).
    One launch of the sphere rasterizer with J = 1; differentiable in both
    arguments."""

    def __init__(self, width, height):
        super().__init__()
        self.width = width
        self.height = height

    def forward(self, xyz_centers, radiuses):
        n = xyz_centers.shape[0]
        spheres = torch.cat([xyz_centers[:, 0:3], radiuses.reshape(n, 1)], dim=1).view(n, 1, 4)
        return ops.SphereDepthRaster.apply(spheres.float(), self.height, self.width)


class HandBallPrimitiveRender(nn.Module):
    """forward(T[B,17,4,4]) -> (part_maps[B,41,H,W], depth_maps[B,H,W])
    (mesh/render.py:81-90).  depth_maps comes from the fused min kernel and
    carries the gradient; the 41x larger part_maps, which only the reference's
    viewer reads (mesh/interactive_viewer.py:61), is rendered by a second launch
    and returned detached unless `differentiable_part_maps` is set."""

    def __init__(self, bones, width, height, differentiable_part_maps=False):
        super().__init__()
        self.width = width
        self.height = height
        self.ball_renderer = BallRender(width, height)
        self.lbs = keypoint_skinning(bones)
        self.num_vertices = self.lbs.num_vertices
        radiuses = [r for bone in bones for _, r in bone.get('keypoint', [])]
        self.register_buffer('radiuses', torch.tensor(radiuses).float().unsqueeze(0))
        self.differentiable_part_maps = differentiable_part_maps

    def spheres(self, transformation_mats):
        pts = self.lbs(transformation_mats)                                  # [B,41,4]
        B = pts.shape[0]
        return torch.cat([pts[:, :, 0:3], self.radiuses.expand(B, -1).unsqueeze(-1)], dim=2)

    def forward(self, transformation_mats):
        sph = self.spheres(transformation_mats).contiguous()
        B = sph.shape[0]
        depth_maps = ops.SphereDepthRaster.apply(sph, self.height, self.width)
        flat = sph.view(-1, 4)
        if self.differentiable_part_maps:
            balls = self.ball_renderer(flat[:, 0:3], flat[:, 3])
        else:
            with torch.no_grad():
                balls = self.ball_renderer(flat[:, 0:3], flat[:, 3])
        part_maps = balls.view(B, self.num_vertices, self.height, self.width)
        return part_maps, depth_maps


class DataToModelLoss(nn.Module):
    """forward(dms[N,H,W], joints[N,J,3]) -> scalar: mean over all pixels of
    clamp(min_j | ||(xg,yg,depth) - c_j|| - r_j |, 0, 50) on pixels with depth <=
    99 (mesh/render.py:123-142).  `mesh` is the model dict or a list of radii
    (mesh/render.py:107-117)."""

    def __init__(self, width, height, mesh):
        super().__init__()
        self.width = width
        self.height = height
        radiuses = torch.from_numpy(np.asarray(radii_of(mesh), np.float32))
        self.num_joints = len(radiuses)
        self.register_buffer('radiuses', radiuses.view(1, 1, 1, self.num_joints))

    def forward(self, dms, joints):
        num_batch = dms.shape[0]
        joints = joints.reshape(num_batch, self.num_joints, 3)
        return ops.DataToModel.apply(dms.reshape(num_batch, self.height, self.width).float(), joints.float(),
                                     self.radiuses.view(-1))
