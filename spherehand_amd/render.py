"""Differentiable renderers and render losses -- the module-level API of the
reference's mesh/render.py (same class names, constructor arguments, forward
signatures and return values), running on the hand-written HIP kernels.

    BallRender               mesh/render.py:10-53
    HandBallPrimitiveRender  mesh/render.py:56-90
    DataToModelLoss          mesh/render.py:93-142
    DepthRasterizationFunction / DepthRasterization / DepthRender  mesh/render.py:282-331
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .hand_model import radii_of, sparse_skin
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


class DepthRasterizationFunction(torch.autograd.Function):
    """mesh/render.py:282-287: the extension call + clamp(max=100).  Forward only
    (the reference defines no backward; callers detach the result)."""

    @staticmethod
    def forward(ctx, width, height, face_vertices):
        depth_maps = ops.tri_raster_fwd(width, height, face_vertices.contiguous())
        return torch.clamp(depth_maps, max=100.0)


class DepthRasterization(nn.Module):
    """mesh/render.py:289-312.  forward(vertices[B,NV,>=3]) -> [B,height,width]:
    rasterize at 640x640, clamp, bilinear-downsample.  `np_faces` is NOT modified
    (the reference swaps its columns in place for the right hand, :298-300)."""

    def __init__(self, width, height, np_faces, right_hand=True):
        super().__init__()
        self.width = width
        self.height = height
        faces = np.array(np_faces, dtype=np.int64, copy=True)
        if right_hand:
            faces[:, [0, 1]] = faces[:, [1, 0]]
        self.register_buffer('faces', torch.from_numpy(faces).view(-1))
        self.register_buffer('faces_i32', torch.from_numpy(faces.astype(np.int32)).contiguous())
        self.num_faces = len(faces)

    def forward(self, vertices):
        num_batch = vertices.shape[0]
        if vertices.is_cuda and vertices.shape[-1] == 4 and vertices.dtype == torch.float32:
            # face gather fused into the rasterizer (no [B,F,3,3] intermediate)
            raw = ops.tri_raster_indexed_fwd(640, 640, vertices.contiguous(), self.faces_i32)
            rendered_dm = torch.clamp(raw, max=100.0).unsqueeze(1)
        else:
            face_vertices = vertices[:, self.faces, 0:3].view(num_batch, self.num_faces, 3, 3)
            rendered_dm = DepthRasterizationFunction.apply(640, 640, face_vertices).unsqueeze(1)
        return torch.nn.functional.interpolate(rendered_dm, size=(self.height, self.width), mode='bilinear',
                                               align_corners=False).squeeze(1)


class SparseSkinning(nn.Module):
    """LinearBlendSkinning (+ optional orthographic camera) of the full mesh on the
    HIP kernel: forward(T[B,17,4,4], camera=None, rand_f=None) -> [B,NV,4]."""

    def __init__(self, mesh, right_hand=True):
        super().__init__()
        start, bone, wv = sparse_skin(mesh)
        self.register_buffer('skin_vertex_start', torch.from_numpy(start))
        self.register_buffer('skin_bone', torch.from_numpy(bone))
        self.register_buffer('skin_wv', torch.from_numpy(wv))
        self.right_hand = right_hand
        self.num_vertices = len(start) - 1

    def forward(self, transformation_mats, camera=None, rand_f=None):
        return ops.lbs_project(transformation_mats.contiguous().float(), self.skin_vertex_start, self.skin_bone,
                               self.skin_wv, self.right_hand, camera,
                               None if rand_f is None else rand_f.contiguous().float())


class DepthRender(nn.Module):
    """mesh/render.py:315-331.  forward(T[B,17,4,4], rand_fx[B]=None) -> depth
    [B,S,S] in mm, background 100: skinning + camera (one launch), triangle raster
    with the face gather fused (fill, raster, decode), clamp + bilinear resize."""

    def __init__(self, mesh, image_size):
        super().__init__()
        self.lbs = SparseSkinning(mesh)
        self.camera = (320.0, 320.0, 640 / 300, 640 / 300)             # :325
        self.rasterizer = DepthRasterization(image_size, image_size, mesh['faces'])

    def forward(self, transformation_mats, rand_fx=None):
        skinned_points = self.lbs(transformation_mats, self.camera, rand_fx)
        return self.rasterizer(skinned_points)
