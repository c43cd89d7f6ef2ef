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
from .hand_model import radii_of, sparse_skin, unique_skin
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
        T = transformation_mats
        if T.is_cuda and T.dtype == torch.float32 and self.lbs.single_bone:   # one launch per direction (keypoint_skin.hip)
            if T.dim() == 5:
                T = T.squeeze(2)
            lbs = self.lbs
            return ops.KeypointSpheres.apply(T, lbs.kp_bone, lbs.skin_wv, self.radiuses.view(-1), lbs.kp_bone_start,
                                             lbs.kp_bone_points, lbs.right_hand)
        pts = self.lbs(transformation_mats)                                  # [B,41,4]
        B = pts.shape[0]
        return torch.cat([pts[:, :, 0:3], self.radiuses.expand(B, -1).unsqueeze(-1)], dim=2)

    def pose_spheres(self, hand_transformation_mat, parameters):
        """spheres(hand_transformation_mat(parameters)) without the bone transforms visiting HBM: pose [B,26] -> sphere
        records [B,41,4], one launch per direction (ops.PoseSpheres; the same records and pose gradient, bit for bit,
        as the two modules chained).  `hand_transformation_mat`: the kinematicsTransformation.HandTransformationMat
        whose offset matrices the bones carry."""
        fk, lbs = hand_transformation_mat, self.lbs
        if not (parameters.is_cuda and parameters.dtype == torch.float32 and lbs.single_bone and fk.offset.shape[0] == 17):
            return self.spheres(fk(parameters))
        return ops.PoseSpheres.apply(parameters, fk.offset, fk.offset_inv, lbs.kp_bone, lbs.skin_wv, self.radiuses.view(-1),
                                     lbs.kp_bone_start, lbs.kp_bone_points, lbs.right_hand)

    def pose_depth(self, hand_transformation_mat, parameters):
        """pose [B,26] -> depth maps [B,H,W] (the differentiable output of forward(), without the part maps): the fit
        chain's three launches per direction."""
        fk, lbs = hand_transformation_mat, self.lbs
        if not (parameters.is_cuda and parameters.dtype == torch.float32 and lbs.single_bone and fk.offset.shape[0] == 17
                and parameters.shape[0] > 0):
            return ops.SphereDepthRaster.apply(self.pose_spheres(fk, parameters), self.height, self.width)
        return ops.PoseDepthRaster.apply(parameters, fk.offset, fk.offset_inv, lbs.kp_bone, lbs.skin_wv, self.radiuses.view(-1),
                                         lbs.kp_bone_start, lbs.kp_bone_points, lbs.right_hand, self.height, self.width)

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
    rasterize at 640x640, clamp, bilinear-downsample -- by default as ONE fused kernel that
    only rasterizes the source pixels the resize reads (`fused`).  `np_faces` is NOT
    modified (the reference swaps its columns in place for the right hand, :298-300)."""

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
        self.fused = True     # False: explicit 640x640 raster, then torch clamp + interpolate

    def forward(self, vertices):
        num_batch = vertices.shape[0]
        on_kernel = vertices.is_cuda and vertices.shape[-1] == 4 and vertices.dtype == torch.float32
        if on_kernel and self.fused and self.width == self.height and 2 * self.width <= 641:
            # raster + clamp + resize in one pass over the sampled source pixels only
            return ops.mesh_depth_fwd(vertices.contiguous(), self.faces_i32, self.height, 640, 100.0)
        if on_kernel:
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
    HIP kernel: forward(T[B,17,4,4], camera=None, rand_f=None) -> [B,NV,4].
    distinct=True: only the mesh's DISTINCT vertices (hand_model.unique_skin: 1 721 of 10 144 -- the reference stores
    each face's corners separately) -> [B,NU,4]; `vertex_index` [NV] maps a mesh vertex to its row."""

    def __init__(self, mesh, right_hand=True, distinct=False):
        super().__init__()
        if distinct:
            start, bone, wv, index = unique_skin(mesh)
            self.vertex_index = index
        else:
            start, bone, wv = sparse_skin(mesh)
            self.vertex_index = None
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
        # the skinned vertices never leave this module: the distinct ones are enough, the faces index them (identical
        # face corners, hence identical images; 16 -> 4 us of skinning and a sixth of the vertex traffic per call)
        self.lbs = SparseSkinning(mesh, distinct=True)
        self.camera = (320.0, 320.0, 640 / 300, 640 / 300)             # :325
        self.rasterizer = DepthRasterization(image_size, image_size, self.lbs.vertex_index[np.asarray(mesh['faces'], np.int64)])

    def forward(self, transformation_mats, rand_fx=None):
        ras = self.rasterizer
        T = transformation_mats
        if T.is_cuda and ras.fused and ras.width == ras.height and 2 * ras.width <= 641:
            # skinning + camera + raster + clamp + resize: one launch where the lattice kernel applies
            # (shr_mesh_render_fwd), the two launches below through a workspace otherwise -- the same bits
            return ops.mesh_render_fwd(T.contiguous().float(), self.lbs.skin_vertex_start, self.lbs.skin_bone,
                                       self.lbs.skin_wv, self.lbs.right_hand, self.camera,
                                       None if rand_fx is None else rand_fx.contiguous().float(), ras.faces_i32,
                                       ras.height, 640, 100.0)
        skinned_points = self.lbs(transformation_mats, self.camera, rand_fx)
        return self.rasterizer(skinned_points)


class HeatmapRender(nn.Module):
    """forward(uvd_points[B,J,>=3]) -> (uv_hm[B,J,S,S], scaled_d_hm[B,J,S,S]): a
    Gaussian exp(-0.5*sigma*d^2) per joint and the joint's depth painted where the
    Gaussian exceeds 0.05 (mesh/render.py:226-248)."""

    def __init__(self, hm_size, sigma=1.0):
        super().__init__()
        self.sigma = sigma
        self.height = self.width = hm_size
        grid = torch.arange(hm_size, dtype=torch.float32)
        self.register_buffer('u_grid', grid.view(1, 1, 1, hm_size))
        self.register_buffer('v_grid', grid.view(1, 1, hm_size, 1))

    def forward(self, uvd_points):
        assert uvd_points.ndimension() == 3
        u = uvd_points[:, :, 0, None, None]
        v = uvd_points[:, :, 1, None, None]
        uv_hm = torch.exp(-0.5 * self.sigma * ((self.u_grid - u) ** 2 + (self.v_grid - v) ** 2))
        d = uvd_points[:, :, 2, None, None].expand_as(uv_hm)
        return uv_hm, torch.where(uv_hm > 0.05, d, torch.zeros_like(d))


class Hand3DHeatmapRender(nn.Module):
    """forward(T[B,17,4,4], rand_f=None) -> (uv heat-maps, depth heat-maps, xyz of the
    41 key-points back-projected from the heat-map camera) (mesh/render.py:274-279)."""

    def __init__(self, bones, heatmap_size):
        super().__init__()
        from .pointTransformation import InverseOthographicalProjection, OthographicalProjection
        self.width = self.height = heatmap_size
        self.hm_renderer = HeatmapRender(heatmap_size)
        half, f = heatmap_size / 2, heatmap_size / 300
        self.camera = OthographicalProjection(half, half, f, f)
        self.inv_camera = InverseOthographicalProjection(half, half, f, f)
        self.lbs = keypoint_skinning(bones)
        self.num_vertices = self.lbs.num_vertices
        self._skin_bone_i32 = None

    def forward(self, transformation_mats, rand_f=None, uv_scale=1.0, d_scale=1.0):
        """(uv_scale / d_scale: HandSynthesizer's heat-map scalings, applied in the same launch on the GPU path)"""
        T = transformation_mats
        if T.is_cuda and not torch.is_grad_enabled() and T.dtype == torch.float32:
            # forward-only GPU path: skinning + camera (one launch), then Gaussians + depth painting +
            # back-projection (one launch)
            lbs = self.lbs
            if self._skin_bone_i32 is None or self._skin_bone_i32.device != T.device:
                self._skin_bone_i32 = lbs.skin_bone.to(device=T.device, dtype=torch.int32).contiguous()
                self._inv_k = self.inv_camera.inv_k_mat[0].cpu().tolist()
            cam = self.camera
            uvd = ops.lbs_project(T.contiguous(), lbs.skin_vertex_start, self._skin_bone_i32, lbs.skin_wv, lbs.right_hand,
                                  (cam.cx, cam.cy, cam.fx, cam.fy), None if rand_f is None else rand_f.contiguous().float())
            return ops.heatmap_paint(uvd, self.width, self.hm_renderer.sigma, self._inv_k, uv_scale, d_scale)
        uvd_points = self.camera(self.lbs(T), rand_f)
        hms, dms = self.hm_renderer(uvd_points)
        return hms * uv_scale, dms * d_scale, self.inv_camera(uvd_points)


def _collision_pairs():
    """Every finger sphere against the 11 palm spheres, and against every sphere of
    another finger (mesh/render.py:150-162): 330 + 360 pairs."""
    a, b = [], []
    for p in range(11):
        for q in range(11, 41):
            a.append(p); b.append(q)
    for p in range(11, 41):
        for q in range(p + 1, 41):
            if (p - 11) // 6 != (q - 11) // 6:
                a.append(p); b.append(q)
    return a, b


class CollisionLoss(nn.Module):
    """sum of relu(min_dist^2 - |c_a - c_b|^2) over the pair table (mesh/render.py:168-176)."""

    def __init__(self, min_dist=6):
        super().__init__()
        self.min_sq_dist = min_dist ** 2
        a, b = _collision_pairs()
        self.register_buffer('joint_1', torch.tensor(a).long())
        self.register_buffer('joint_2', torch.tensor(b).long())

    def forward(self, joints):
        joints = joints.reshape(joints.shape[0], -1, 3)
        sq = ((joints[:, self.joint_1] - joints[:, self.joint_2]) ** 2).sum(-1)
        return torch.relu(self.min_sq_dist - sq).sum()


# The 35 sphere pairs whose distance is held to [0.80, 1.05] x its rest length
# (data of mesh/bone_length.py:36-55: 20 palm pairs + 3 per finger).
BONE_PAIRS_1 = [3, 2, 3, 8, 2, 2, 9, 8, 4, 8, 7, 4, 6, 7, 0, 5, 7, 7, 6, 6] + \
    [11 + 6 * f + 2 * k for f in range(5) for k in range(3)]
BONE_PAIRS_2 = [2, 9, 8, 2, 4, 10, 10, 4, 10, 7, 4, 6, 10, 6, 5, 1, 0, 5, 5, 1] + \
    [12 + 6 * f + 2 * k for f in range(5) for k in range(3)]
BONE_REST_LENGTH = [
    25.212656021118164, 18.249488830566406, 27.5742244720459, 38.532264709472656, 25.10819435119629,
    31.173757553100586, 18.329626083374023, 19.15080451965332, 16.209327697753906, 21.52261734008789,
    32.740535736083984, 30.58920669555664, 33.205970764160156, 11.672294616699219, 17.084707260131836,
    17.084720611572266, 16.697546005249023, 23.92103385925293, 20.87999725341797, 22.58038330078125,
    27.55999755859375, 15.471183776855469, 13.214692115783691, 21.748210906982422, 13.021653175354004,
    16.643720626831055, 18.83765983581543, 12.724685668945312, 16.238431930541992, 18.04928970336914,
    11.045844078063965, 11.320968627929688, 30.078536987304688, 16.255985260009766, 19.434825897216797]


class BoneLengthLoss(nn.Module):
    """mean relu(min^2 - d^2) + mean relu(d^2 - max^2) over the 35 pairs (mesh/render.py:196-206)."""

    def __init__(self):
        super().__init__()
        rest = torch.tensor(BONE_REST_LENGTH).float()
        self.register_buffer('joint_1', torch.tensor(BONE_PAIRS_1).long())
        self.register_buffer('joint_2', torch.tensor(BONE_PAIRS_2).long())
        self.register_buffer('max_length', ((rest * 1.05) ** 2).unsqueeze(0))
        self.register_buffer('min_length', ((rest * 0.80) ** 2).unsqueeze(0))

    def forward(self, joints):
        joints = joints.reshape(joints.shape[0], -1, 3)
        sq = ((joints[:, self.joint_1] - joints[:, self.joint_2]) ** 2).sum(-1)
        return torch.relu(self.min_length - sq).mean() + torch.relu(sq - self.max_length).mean()
