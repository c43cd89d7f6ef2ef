"""ctypes/numpy binding of the CPU ORACLE (oracle/libspherehand_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (spherehand_amd/) must never
import this module and has no CPU fallback.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspherehand_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "spherehand_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


# ---- oracle/_ref: the reference's own triangle kernel, compiled for gfx950 (oracle/Makefile, target ref_tri) ----------
REF_CU = "/root/reference/mesh/cuda_kernel/depth_rasterization_cuda_kernel.cu"
_REF_DIR = os.path.join(_HERE, "_ref")


def build_ref():
    """Builds oracle/_ref/libref_tri*.so when the reference is there (the build container); the GPU box gets the
    built files with the snapshot.  Returns the path of the build of record, or None."""
    so = os.path.join(_REF_DIR, "libref_tri.so")
    if os.path.exists(REF_CU):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref_tri"])
    return so if os.path.exists(so) else None


def ref_tri_lib(contract=False):
    """ctypes handle of the reference kernel build (contract=False: -ffp-contract=off, every operation as written;
    True: clang's default contraction), or None when oracle/_ref was not built.  Needs a GPU to call."""
    so = os.path.join(_REF_DIR, "libref_tri_contract.so" if contract else "libref_tri.so")
    if not os.path.exists(so):
        return None
    h = ctypes.CDLL(so)
    h.ref_tri_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    h.ref_tri_forward.restype = ctypes.c_int
    return h


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def num_threads():
    return int(lib().oracle_num_threads())


def ball_render(centres, radii, H, W):
    c = _f32(centres)
    r = _f32(radii).reshape(-1)
    n, stride = c.shape
    out = np.empty((n, H, W), np.float32)
    _chk(lib().oracle_ball_render(_p(c), stride, _p(r), n, H, W, _p(out)), "oracle_ball_render")
    return out


def sphere_raster_fwd(spheres, H, W, want_argmin=True):
    s = _f32(spheres)
    N, J, four = s.shape
    assert four == 4
    depth = np.empty((N, H, W), np.float32)
    arg = np.empty((N, H, W), np.uint8) if want_argmin else None
    _chk(lib().oracle_sphere_raster_fwd(_p(s), N, J, H, W, _p(depth),
                                        _p(arg, _u8p) if want_argmin else None),
         "oracle_sphere_raster_fwd")
    return (depth, arg) if want_argmin else depth


def sphere_raster_bwd(spheres, grad_depth):
    s = _f32(spheres)
    g = _f32(grad_depth)
    N, J, _ = s.shape
    _, H, W = g.shape
    out = np.empty((N, J, 4), np.float32)
    _chk(lib().oracle_sphere_raster_bwd(_p(s), _p(g), N, J, H, W, _p(out)), "oracle_sphere_raster_bwd")
    return out


def data_to_model_fwd(depth, centres, radii):
    d = _f32(depth)
    c = _f32(centres)
    r = _f32(radii).reshape(-1)
    N, H, W = d.shape
    J = c.shape[1]
    sums = np.empty(N, np.float64)
    _chk(lib().oracle_data_to_model_fwd(_p(d), _p(c), _p(r), N, J, H, W, _p(sums, _f64p)),
         "oracle_data_to_model_fwd")
    return sums


def data_to_model_loss(depth, centres, radii):
    d = np.asarray(depth)
    return float(data_to_model_fwd(depth, centres, radii).sum() / d.size)


def data_to_model_bwd(depth, centres, radii):
    d = _f32(depth)
    c = _f32(centres)
    r = _f32(radii).reshape(-1)
    N, H, W = d.shape
    J = c.shape[1]
    out = np.empty((N, J, 3), np.float32)
    _chk(lib().oracle_data_to_model_bwd(_p(d), _p(c), _p(r), N, J, H, W, _p(out)),
         "oracle_data_to_model_bwd")
    return out


def fma_variant():
    """The SAME source compiled with FMA contraction (-ffp-contract=fast -mfma): a sensitivity probe for
    the triangle kernel (what nvcc's default -fmad=true might do to the reference), never the oracle."""
    so = os.path.join(_HERE, "libspherehand_oracle_fma.so")
    src = os.path.join(_HERE, "spherehand_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libspherehand_oracle_fma.so"])
    return ctypes.CDLL(so)


def tri_raster_fwd(face_vertices, W, H, handle=None):
    fv = _f32(face_vertices)
    B, F = fv.shape[0], fv.shape[1]
    out = np.empty((B, H, W), np.float32)
    _chk((handle or lib()).oracle_tri_raster_fwd(_p(fv), B, F, W, H, _p(out)), "oracle_tri_raster_fwd")
    return out


def clamp_bilinear(src, Hd, Wd, clamp_max=100.0):
    s = _f32(src)
    B, Hs, Ws = s.shape
    out = np.empty((B, Hd, Wd), np.float32)
    _chk(lib().oracle_clamp_bilinear(_p(s), B, Hs, Ws, Hd, Wd, ctypes.c_float(clamp_max), _p(out)),
         "oracle_clamp_bilinear")
    return out


def lbs_project(T, skin_vertex_start, skin_bone, skin_wv, right_hand=True, camera=None, rand_f=None):
    """camera = (cx, cy, fx, fy) or None (no projection)."""
    T = _f32(T)
    B, NB = T.shape[0], T.shape[1]
    vs = np.ascontiguousarray(skin_vertex_start, np.int32)
    sb = np.ascontiguousarray(skin_bone, np.int32)
    wv = _f32(skin_wv)
    NV = len(vs) - 1
    out = np.empty((B, NV, 4), np.float32)
    cx, cy, fx, fy = camera if camera is not None else (0.0, 0.0, 1.0, 1.0)
    rf = _f32(rand_f).reshape(-1) if rand_f is not None else None
    _chk(lib().oracle_lbs_project(_p(T), B, NB, NV, _p(vs, _i32p), _p(sb, _i32p), _p(wv),
                                  int(bool(right_hand)), int(camera is not None),
                                  ctypes.c_float(cx), ctypes.c_float(cy), ctypes.c_float(fx),
                                  ctypes.c_float(fy), _p(rf) if rf is not None else None, _p(out)),
         "oracle_lbs_project")
    return out


def fk_fwd(params, offset, offset_inv):
    p = _f32(params)
    B = p.shape[0]
    assert p.shape[1] == 26
    off = _f32(offset)
    inv = _f32(offset_inv)
    out = np.empty((B, 17, 4, 4), np.float32)
    _chk(lib().oracle_fk_fwd(_p(p), B, _p(off), _p(inv), _p(out)), "oracle_fk_fwd")
    return out
