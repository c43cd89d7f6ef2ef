"""A tiny synthetic NYU-layout directory (shared by tests/golden/make_goldens_nyu.py and
tests/test_nyu_generator_cpu.py): `<root>/<subset>/joint_data.mat` with joint_xyz [3, n, 36, 3] and one
`depth_<cam>_<frame>.png` per camera and frame (480 x 640, depth in mm: high byte in G, low byte in B), drawn
from a seed -- a blob of hand-sized depth values around the projection of joint 32 on a far background."""
import os

import numpy as np


def write_synthetic_nyu(root, subset="train", frames=3, seed=0):
    import scipy.io as sio
    from PIL import Image
    rs = np.random.RandomState(seed)
    d = os.path.join(root, subset)
    os.makedirs(d, exist_ok=True)
    fx, fy, cx, cy = 588.235, 587.084, 320, 240
    joints = np.zeros((3, frames, 36, 3), np.float32)
    for c in range(3):
        for f in range(frames):
            centre = np.array([rs.uniform(-150, 150), rs.uniform(-100, 100), rs.uniform(600, 900)], np.float32)
            joints[c, f] = centre[None] + rs.normal(0, 35, (36, 3)).astype(np.float32)
            joints[c, f, 32] = centre
            depth = np.full((480, 640), 2001, np.int32)
            # the mat stores y up; the generator flips it (joint[:, :, 1] *= -1) before projecting
            u = int(centre[0] * fx / centre[2] + cx)
            v = int(centre[1] * fy / centre[2] + cy)
            r = 70
            vv, uu = np.mgrid[max(0, v - r):min(480, v + r), max(0, u - r):min(640, u + r)]
            blob = (uu - u) ** 2 + (vv - v) ** 2 < r * r
            vals = (centre[2] + 60 * np.sin(uu / 7.0) * np.cos(vv / 5.0) + rs.randint(-20, 20, uu.shape)).astype(np.int32)
            depth[vv[blob], uu[blob]] = vals[blob]
            rgb = np.zeros((480, 640, 3), np.uint8)
            rgb[..., 1] = (depth >> 8) & 255
            rgb[..., 2] = depth & 255
            Image.fromarray(rgb).save(os.path.join(d, "depth_%d_%07d.png" % (c + 1, f + 1)))
    joints_mat = joints.copy()
    joints_mat[..., 1] *= -1            # stored with y up
    sio.savemat(os.path.join(d, "joint_data.mat"), {"joint_xyz": joints_mat})
    return d
