"""Skinning and cameras -- same classes and signatures as the reference's
mesh/pointTransformation.py, restructured around sparse skinning.

The reference multiplies every bone with a dense [1,17,NV,4,1] buffer that is
85 % zeros (mesh/pointTransformation.py:25-43).  Here the (bone, vertex) pairs
are kept as a CSR-by-vertex table (hand_model.sparse_skin) and gathered.
"""
import numpy as np
import torch
import torch.nn as nn


class LinearBlendSkinning(nn.Module):
    """mesh/pointTransformation.py:11-46.  forward(T[B,NB,4,4]) -> [B,NV,4]."""

    def __init__(self, vertices, skinning_weights, skinning_vertex_indices, right_hand=True):
        super().__init__()
        assert len(skinning_vertex_indices) == len(skinning_weights), \
            'vertex index and weight should be with the same size'
        vertices = np.asarray(vertices)
        nv = vertices.shape[0]
        vid, bid, wv = [], [], []
        for b, (w, ids) in enumerate(zip(skinning_weights, skinning_vertex_indices)):
            assert len(w) == len(ids), 'vertex index and weight should be the same size'
            if len(ids) == 0:
                continue
            ids = np.asarray(ids, np.int64)
            w = np.asarray(w, np.float64)
            vid.append(ids)
            bid.append(np.full(len(ids), b, np.int64))
            # float32(w * v): the value the reference stores (:31)
            wv.append((w[:, None] * vertices[ids].astype(np.float64)).astype(np.float32))
        vid = np.concatenate(vid) if vid else np.zeros(0, np.int64)
        bid = np.concatenate(bid) if bid else np.zeros(0, np.int64)
        wv = np.concatenate(wv) if wv else np.zeros((0, 4), np.float32)
        order = np.lexsort((bid, vid))
        self.num_vertices = nv
        self.right_hand = right_hand
        start = np.zeros(nv + 1, np.int64)
        np.add.at(start, vid + 1, 1)
        self.register_buffer('skin_vertex', torch.from_numpy(vid[order]))
        self.register_buffer('skin_bone', torch.from_numpy(bid[order]))
        self.register_buffer('skin_wv', torch.from_numpy(np.ascontiguousarray(wv[order])))
        self.register_buffer('skin_vertex_start', torch.from_numpy(np.cumsum(start).astype(np.int32)))
        # one-bone-per-vertex (the 41 key-points: mesh/render.py:65-77) is a pure gather
        self.single_bone = bool(len(vid) == nv and np.array_equal(np.sort(vid), np.arange(nv)))
        if self.single_bone:
            # the key-point kernels' tables (ops.KeypointSpheres): bone of every point, and the points of every bone (CSR)
            kb = bid[order].astype(np.int64)
            nb = len(skinning_weights)
            bstart = np.zeros(nb + 1, np.int64)
            np.add.at(bstart, kb + 1, 1)
            self.register_buffer('kp_bone', torch.from_numpy(kb.astype(np.int32)), persistent=False)
            self.register_buffer('kp_bone_start', torch.from_numpy(np.cumsum(bstart).astype(np.int32)), persistent=False)
            self.register_buffer('kp_bone_points', torch.from_numpy(np.argsort(kb, kind='stable').astype(np.int32)),
                                 persistent=False)
        # x -> -x for the right hand (:44-45) as a resident constant (a new_tensor() per call is a host -> device copy
        # on every step and cannot be captured in a hipGraph)
        self.register_buffer('hand_sign', torch.tensor([-1.0, 1.0, 1.0, 1.0]), persistent=False)

    def forward(self, bone_transformations):
        T = bone_transformations
        if T.dim() == 5:
            T = T.squeeze(2)
        B = T.shape[0]
        per_entry = torch.matmul(T[:, self.skin_bone], self.skin_wv.unsqueeze(0).unsqueeze(-1)).squeeze(-1)
        if self.single_bone:
            out = per_entry                                   # already ordered by vertex
        else:
            out = torch.zeros(B, self.num_vertices, 4, dtype=T.dtype, device=T.device)
            out.index_add_(1, self.skin_vertex, per_entry)
        if self.right_hand:
            out = out * self.hand_sign.to(out.dtype)
        return out


class OthographicalProjection(nn.Module):
    """mesh/pointTransformation.py:69-99: u = x*fx (+ per-sample focal) + cx ..."""

    def __init__(self, cx, cy, fx, fy):
        super().__init__()
        self.cx, self.cy, self.fx, self.fy = cx, cy, fx, fy
        k = torch.eye(4)
        k[0, 0], k[1, 1], k[0, 3], k[1, 3] = fx, fy, cx, cy
        self.register_buffer('k_mat', k.unsqueeze(0).float())

    def forward(self, xyz_points, rand_f=None):
        B, NV = xyz_points.shape[0], xyz_points.shape[1]
        if rand_f is None:
            return torch.matmul(self.k_mat, xyz_points.reshape(-1, 4, 1)).view(B, NV, 4)
        f = rand_f.view(-1, 1)
        p = xyz_points.view(B, -1, 4)
        return torch.stack([p[:, :, 0] * f * self.fx + self.cx, p[:, :, 1] * f * self.fy + self.cy,
                            p[:, :, 2], torch.ones_like(p[:, :, 2])], dim=-1)


class InverseOthographicalProjection(nn.Module):
    """mesh/pointTransformation.py:102-124."""

    def __init__(self, cx, cy, fx, fy):
        super().__init__()
        self.cx, self.cy, self.fx, self.fy = cx, cy, fx, fy
        k = torch.eye(4)
        k[0, 0], k[1, 1], k[0, 3], k[1, 3] = fx, fy, cx, cy
        self.register_buffer('inv_k_mat', torch.inverse(k).unsqueeze(0).float())

    def forward(self, uvd_points):
        B, NV = uvd_points.shape[0], uvd_points.shape[1]
        return torch.matmul(self.inv_k_mat, uvd_points.reshape(-1, 4, 1)).view(B, NV, 4)


class RandScale(nn.Module):
    """mesh/pointTransformation.py:128-148: per-sample anisotropic scale in
    0.90 +- rand_scale/2, left-multiplied onto every bone transform.  The three
    torch.rand(batch) draws come from the CPU generator, as in the reference."""

    def __init__(self, rand_scale):
        super().__init__()
        self.rand_scale = rand_scale

    def forward(self, transform_mats):
        B = transform_mats.shape[0]
        s = [torch.rand(B) * self.rand_scale + 0.90 - self.rand_scale / 2 for _ in range(3)]
        diag = torch.stack(s + [torch.ones(B)], dim=1).to(transform_mats)       # [B,4]
        return transform_mats * diag.view(B, 1, 4, 1)                          # diag(s) @ T
