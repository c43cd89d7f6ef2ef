"""ctypes loader for libspherehand_hip.so (the C ABI in include/spherehand_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing
any op raises.  Build it with ``python -m spherehand_amd.build``.
"""
import ctypes
import os

# torch FIRST: the wheel bundles its own ROCm runtime (torch/lib/libamdhip64.so,
# SONAME libamdhip64.so.7).  Loaded after torch, our library's NEEDED
# libamdhip64.so.7 binds to that same runtime; loaded before torch, the process
# would end up with two HIP runtimes (/opt/rocm's and torch's) and kernels
# launched on torch's streams fail with hipErrorNoDevice.
import torch  # noqa: F401  (device memory, streams: the plumbing this library sits on)

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_PKG, "libspherehand_hip.so")
ABI_VERSION = 22

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> argtypes; every symbol include/spherehand_hip.h declares
SIGNATURES = {
    "shr_abi_version": ([], _i),
    "shr_error_string": ([_i], ctypes.c_char_p),
    "shr_device_info": ([ctypes.c_char_p, _i, ctypes.POINTER(_i)], _i),
    "shr_sphere_raster_fwd": ([_vp, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_sphere_raster_fwd_ex": ([_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp], _i),
    "shr_keypoint_spheres_fwd": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "shr_keypoint_spheres_bwd": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "shr_sphere_raster_bwd": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp], _i),
    "shr_set_tuning": ([_i, _i], _i),
    "shr_data_to_model": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_data_to_model_indexed": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_data_to_model_parts": ([_i, _i, _i], _i),
    "shr_data_to_model_partial": ([_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_data_to_model_points_bytes": ([_i, _i, _i], ctypes.c_longlong),
    "shr_data_to_model_compact": ([_vp, _i, _i, _i, _vp, _vp], _i),
    "shr_mv_project_compact": ([_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "shr_data_to_model_from_points": ([_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_data_to_model_from_points_indexed": ([_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_data_to_model_from_points_ordered": ([_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_mv_loss_combine": ([_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "shr_sphere_raster_mse_regions": ([_i, _i], _i),
    "shr_pair_losses": ([_vp, ctypes.c_longlong, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_mv_consistency": ([_vp, _vp, _i, _i, _i, _vp, _vp, _vp], _i),
    "shr_depth_resample": ([_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp], _i),
    "shr_soft_argmax_supported": ([_i, _i, _i], _i),
    "shr_soft_argmax_fwd": ([_vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _i, _i, _i, _i, _f, _f, _f, _f, _f,
                             _vp, _vp], _i),
    "shr_soft_argmax_bwd": ([_vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _i, _i, _i, _i, _f, _f, _f, _f, _f,
                             _vp, _vp, _vp], _i),
    "shr_heatmap_paint": ([_vp, _i, _i, _f, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp], _i),
    "shr_depth_noise": ([_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp], _i),
    "shr_synth_pose_fwd": ([_vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp], _i),
    "shr_mesh_render_post_fwd": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _i, _i, _i, _f, _f, _vp, _f, _f,
                                 _vp, _vp, _vp, _vp, _vp], _i),
    "shr_mesh_render_one_launch": ([_i, _i, _i, _i, _i], _i),
    "shr_hand_synth_one_launch": ([_i, _i, _i, _i, _i, _i, _i], _i),
    "shr_hand_synth_fwd": ([_vp, _i, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _i, _i, _i, _f, _f, _i, _f, _f,
                           _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_heatmap_render_fwd": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _i, _f, _f, _f, _f, _f, _f, _f, _vp, _vp,
                               _vp, _vp], _i),
    "shr_group_norm_relu_supported": ([_i, _i], _i),
    "shr_group_norm_relu_fwd": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp], _i),
    "shr_group_norm_relu_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_sphere_raster_mse": ([_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_sphere_raster_mse_indexed": ([_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_sphere_raster_mse_ordered": ([_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_mutual_project_fwd": ([_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "shr_mutual_project_bwd": ([_vp, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "shr_tri_raster_fwd": ([_vp, _i, _i, _i, _i, _vp, _vp], _i),
    "shr_tri_raster_indexed_fwd": ([_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "shr_mesh_depth_fwd": ([_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp], _i),
    "shr_mesh_render_fwd": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "shr_lbs_project": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp], _i),
    "shr_fk_fwd": ([_vp, _i, _vp, _vp, _vp, _vp], _i),
    "shr_fk_bwd": ([_vp, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "shr_pose_spheres_fwd": ([_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp], _i),
    "shr_pose_spheres_bwd": ([_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp], _i),
    "shr_selftest_sqrt": ([ctypes.c_uint, ctypes.c_uint, _vp, _vp], _i),
    "shr_selftest_division": ([ctypes.c_uint, ctypes.c_uint, _vp, _vp], _i),
    "shr_selftest_launch_floor": ([_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp], _i),
}

_lib = None


class SphereHandLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise SphereHandLibraryError(
            "%s not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m spherehand_amd.build`." % SO_PATH)
    try:
        h = ctypes.CDLL(SO_PATH)
    except OSError as e:  # pragma: no cover
        raise SphereHandLibraryError("cannot load %s: %s" % (SO_PATH, e))
    for name, (args, res) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError:
            raise SphereHandLibraryError("%s does not export %s (stale build?)" % (SO_PATH, name))
        fn.argtypes = args
        fn.restype = res
    v = h.shr_abi_version()
    if v != ABI_VERSION:
        raise SphereHandLibraryError("ABI version %d, expected %d: rebuild" % (v, ABI_VERSION))
    _lib = h
    return h


def check(rc, what):
    if rc != 0:
        msg = lib().shr_error_string(rc).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, rc))
