"""Top-level drop-in for the reference's CUDA extension module `depth_rasterization`
(mesh/cuda_kernel/setup.py:4-7, imported by mesh/cuda_kernel/__init__.py:1 and
mesh/render.py:6).

    forward(width: int, height: int, vertices: Tensor[B,F,3,3]) -> Tensor[B,height,width]

Same argument order, preconditions (CUDA + contiguous, violation -> RuntimeError:
depth_rasterization_cuda.cpp:11-19), background value (1000.0) and output
ownership (a fresh tensor on the input's device) as the reference; the work is
done by the hand-written HIP kernels behind libspherehand_hip.so, enqueued on
torch's current stream.
"""
from spherehand_amd.ops import tri_raster_fwd as _tri_raster_fwd


def forward(width, height, vertices):
    return _tri_raster_fwd(int(width), int(height), vertices)
