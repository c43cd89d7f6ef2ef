"""spherehand_amd -- MI355X-native depth rasterizers and render-and-fit losses behind
the API surface of melonwan/sphereHand.

Importing the package does not load the HIP library; the first op does, and raises
if `libspherehand_hip.so` is missing (`python -m spherehand_amd.build`).  There is no
CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["ops", "render", "multiview_utility", "kinematicsTransformation", "pointTransformation", "criterion",
           "engine", "hand_model", "joint_angle", "datasets", "util_modules", "hourglass", "build"]
