"""Import shim for the *reference* (melonwan/sphereHand, read-only at /root/reference).

ONLY used by the golden-vector generator scripts in this directory, which run in
the build container.  Nothing here (and nothing under /root/reference) is needed,
read or imported by the tests, bench.py or the product at run time: the GPU box
has no /root/reference.

The reference needs three import-time accommodations on this image
(SURVEY.md section 8c):
  * numpy >= 1.24 removed the ``np.float`` alias the reference still spells
    (mesh/kinematicsTransformation.py:118,132-134,162-163);
  * ``cv2`` is not installed; it is imported by network/engine.py:9 but never
    touched on the render / loss path we generate vectors for;
  * the CUDA extension ``depth_rasterization`` (mesh/cuda_kernel/__init__.py:1)
    cannot be built here (no nvcc, no CUDA headers).  A module object of that
    name is registered so that ``mesh.render`` imports; its ``forward`` raises
    unless a caller explicitly installs a rasterizer (see make_goldens_mesh.py,
    which installs OUR oracle and says so in the fixture's metadata).
"""
import os
import sys
import types

REF = "/root/reference"


def _no_cuda_ext(*_a, **_k):
    raise RuntimeError(
        "reference CUDA extension depth_rasterization is unbuildable in this "
        "container (needs nvcc + CUDA headers)")


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference mount %s not present: goldens can only be "
                         "regenerated in the build container" % REF)
    sys.dont_write_bytecode = True          # the mount is read-only
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    if "depth_rasterization" not in sys.modules:
        dr = types.ModuleType("depth_rasterization")
        dr.forward = _no_cuda_ext
        sys.modules["depth_rasterization"] = dr
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)                            # constants.py uses cwd-relative paths
    return sys.modules["depth_rasterization"]


def load_reference_mesh():
    """A fresh (deep) copy of the reference hand model dict.

    mesh/render.py:298-300 swaps face columns in place on the caller's array, so
    every consumer gets its own unpickled copy."""
    import pickle
    with open(os.path.join(REF, "mesh/model/preprocessed_hand.pkl"), "rb") as f:
        return pickle.load(f)
