"""Hand model (41-sphere proxy + triangle mesh + skinning) for the render path.

The arrays live in ``spherehand_amd/data/hand_model.npz``; they are a plain
re-export of the reference's ``mesh/model/preprocessed_hand.pkl`` (data only,
made by tests/golden/make_goldens_sphere.py).  ``load_mesh()`` returns a dict in
the reference's own layout -- ``{'vertices', 'faces', 'bones': [{'name',
'offset_matrix', 'weight_vertexid', 'weight_coeff', 'keypoint': [(xyz, r)]}]}``
-- so the drop-in modules accept either this dict or the reference's unpickled
one (network/constants.py:4-5, mesh/render.py:65-77,318-326).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "hand_model.npz")

NUM_BONES = 17
NUM_SPHERES = 41
NUM_POSE_PARAMS = 26


def load_arrays(path=_DATA):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def load_mesh(path=_DATA):
    """Fresh copy of the model in the reference's dict layout (callers may mutate
    it: mesh/render.py:298-300 swaps face columns in place)."""
    a = load_arrays(path)
    bones = []
    for b in range(len(a["bone_names"])):
        sel = a["skin_bone"] == b
        bone = {
            "name": str(a["bone_names"][b]),
            "offset_matrix": a["offset_matrices"][b].copy(),
            "weight_vertexid": a["skin_vertex"][sel].astype(np.int64),
            "weight_coeff": a["skin_weight"][sel].copy(),
        }
        kp = np.nonzero(a["keypoint_bone"] == b)[0]
        if len(kp):
            bone["keypoint"] = [(a["keypoint_xyz"][i].copy(), float(a["keypoint_radius"][i])) for i in kp]
        bones.append(bone)
    return {"bones": bones, "vertices": a["vertices"].copy(), "faces": a["faces"].copy()}


def sphere_model(mesh_or_bones):
    """(rest centres [J,4] f32 homogeneous, radii [J] f32, bone id [J] i32) in the
    reference's sphere order (mesh/render.py:65-77: bones in order, key-points in
    order within a bone)."""
    bones = mesh_or_bones["bones"] if isinstance(mesh_or_bones, dict) else mesh_or_bones
    c, r, b = [], [], []
    for bi, bone in enumerate(bones):
        for pt, radius in bone.get("keypoint", []):
            c.append(np.asarray([pt[0], pt[1], pt[2], 1.0], np.float32))
            r.append(radius)
            b.append(bi)
    return (np.asarray(c, np.float32).reshape(-1, 4), np.asarray(r, np.float32),
            np.asarray(b, np.int32))


def radii_of(mesh):
    """Radii from a model dict or a plain list (mesh/render.py:107-117)."""
    if isinstance(mesh, dict):
        return sphere_model(mesh)[1]
    if isinstance(mesh, (list, tuple, np.ndarray)):
        return np.asarray(mesh, np.float32).reshape(-1)
    raise TypeError("mesh can only be list or dict")


def offset_matrices(mesh):
    """(offset [17,4,4] f32, offset^-1 [17,4,4] f32).  The reference casts the
    offsets to fp32 and inverts them (mesh/kinematicsTransformation.py:86-87,
    185); the inverse is taken in fp64 here and rounded once."""
    off = np.stack([np.asarray(b["offset_matrix"]).astype(np.float32) for b in mesh["bones"]])
    inv = np.linalg.inv(off.astype(np.float64)).astype(np.float32)
    return off, inv


def sparse_skin(mesh):
    """CSR-by-vertex skinning table for the triangle mesh.

    Returns (vertex_start [NV+1] i32, bone [NS] i32, wv [NS,4] f32) where
    wv = float32(weight * vertex) exactly as the reference builds its dense
    buffer (mesh/pointTransformation.py:26-33), entries of a vertex in ascending
    bone order."""
    V = np.asarray(mesh["vertices"], np.float64)
    NV = V.shape[0]
    vid, bid, wv = [], [], []
    for b, bone in enumerate(mesh["bones"]):
        ids = np.asarray(bone["weight_vertexid"], np.int64)
        w = np.asarray(bone["weight_coeff"], np.float64)
        vid.append(ids)
        bid.append(np.full(len(ids), b, np.int32))
        wv.append((w[:, None] * V[ids]).astype(np.float32))
    vid = np.concatenate(vid)
    bid = np.concatenate(bid)
    wv = np.concatenate(wv)
    order = np.lexsort((bid, vid))
    vid, bid, wv = vid[order], bid[order], wv[order]
    start = np.zeros(NV + 1, np.int32)
    np.add.at(start, vid + 1, 1)
    start = np.cumsum(start).astype(np.int32)
    return start, bid.astype(np.int32), np.ascontiguousarray(wv, np.float32)


def unique_skin(mesh):
    """The skin table of the DISTINCT vertices of the triangle mesh, and where each of the mesh's vertices went.

    The reference's mesh stores every face's three corners as vertices of their own (10 144 records for 1 721 distinct
    points); copies of a point carry byte-identical (bone, weight * vertex) entries, so skinning one of them gives the
    bits of all.  Returns (vertex_start [NU+1] i32, bone [NSU] i32, wv [NSU,4] f32, index [NV] i64) with
    skinned_all[:, v] == skinned_unique[:, index[v]] bit for bit (first-occurrence order)."""
    start, bone, wv = sparse_skin(mesh)
    NV = len(start) - 1
    seen, keep, index = {}, [], np.empty(NV, np.int64)
    for v in range(NV):
        a, b = start[v], start[v + 1]
        key = (bone[a:b].tobytes(), wv[a:b].tobytes())
        u = seen.get(key)
        if u is None:
            u = seen[key] = len(keep)
            keep.append(v)
        index[v] = u
    ustart = np.zeros(len(keep) + 1, np.int32)
    ubone, uwv = [], []
    for u, v in enumerate(keep):
        a, b = start[v], start[v + 1]
        ustart[u + 1] = ustart[u] + (b - a)
        ubone.append(bone[a:b]); uwv.append(wv[a:b])
    return ustart, np.concatenate(ubone).astype(np.int32), np.ascontiguousarray(np.concatenate(uwv), np.float32), index
