"""NYU hand-pose frames -> multiview orthographic crops + camera poses -> shards: the offline step that feeds
`datasets.NyuShardDataset`, parameterised by the crop size S (the reference is fixed at 64,
dataset/nyu_generator.py:18).

    NyuDatasetGenerator(dataset_dir, subset, image_size=64)     dataset/nyu_generator.py:15-130
    crop_dm(dm, xyz_center, depth_camera, cube_size, img_size)  dataset/utils.py:70-124
    estimate_rigid_transformation(point_1, point_2)             dataset/utils.py:127-145

Pure numpy on the host (data preparation, no GPU work).  The arithmetic follows the reference operation by
operation -- the same float32 / float64 steps, truncating float -> int conversions, later pixels overwriting earlier ones in
row-major order of the source region -- so the crops, joints and poses equal the reference's bit for bit
(tests/test_nyu_generator_cpu.py, against vectors made by running the reference on the same synthetic frames).
Reference quirks kept because the training data depends on them: the y axis of the annotations is flipped
(:30-31), crops are centred on joint 32 (:63), and the rigid transform's translation is stored in ROW 3 of the
4x4 (dataset/utils.py:144) while the losses read column 3 -- so it is effectively ignored downstream.
"""
import os
from typing import NamedTuple

import numpy as np

from .datasets import write_nyu_shard


class CameraIntrinsic(NamedTuple):
    fx: float = 588.235
    fy: float = 587.084
    cx: float = 320
    cy: float = 240


def _project(xyz, cam):
    """pinhole projection of one point (x f / z + c)"""
    return np.array([xyz[0] * cam.fx / xyz[2] + cam.cx, xyz[1] * cam.fy / xyz[2] + cam.cy, xyz[2]], np.float64)


def crop_dm(dm, xyz_center, depth_camera, cube_size, img_size, far_point_value=100.0):
    """Depth frame [h,w] (mm) -> orthographic crop `img_size` of the `cube_size` box around `xyz_center`:
    every frame pixel inside the box is back-projected, centred, and dropped into the crop cell it falls in
    (truncation toward zero); empty cells hold `far_point_value`."""
    assert dm.ndim == 2, 'unknown dimension of depth map, should be 2'
    h, w = dm.shape
    centre = _project(xyz_center, depth_camera)
    assert 0 <= centre[0] < w and 0 <= centre[1] < h
    half = np.asarray([cube_size[0] / 2, cube_size[1] / 2, cube_size[2] / 2], np.float32)
    z_start, z_end = float(xyz_center[2] - cube_size[2] / 2), float(xyz_center[2] + cube_size[2] / 2)
    top_left = _project(xyz_center + half * np.asarray([-1, -1, -1], np.float32), depth_camera)
    bottom_right = _project(xyz_center + half * np.asarray([1, 1, -1], np.float32), depth_camera)
    u0, u1 = int(max(top_left[0], 0)), int(min(bottom_right[0], w))
    v0, v1 = int(max(top_left[1], 0)), int(min(bottom_right[1], h))
    out = np.ones(img_size) * far_point_value
    fx, fy = img_size[0] / cube_size[0], img_size[1] / cube_size[1]
    cx, cy = img_size[0] / 2, img_size[1] / 2
    roi = dm[v0:v1, u0:u1]
    keep = np.logical_and(roi >= z_start, roi < z_end)
    vv, uu = np.nonzero(keep)                                      # row-major: the order later writes win in
    # back-projection in the frame's own fp32 (pixel indices and depth are float32 arrays in the reference,
    # dataset/utils.py:103-107), centring and the orthographic camera in float64
    u = (uu + u0).astype(np.float32)
    v = (vv + v0).astype(np.float32)
    d = roi[keep]
    x = ((u - depth_camera.cx) * d / depth_camera.fx).astype(np.float64) - xyz_center[0]
    y = ((v - depth_camera.cy) * d / depth_camera.fy).astype(np.float64) - xyz_center[1]
    z = d.astype(np.float64) - xyz_center[2]
    cu = (x * fx + cx).astype(np.int32)
    cv = (y * fy + cy).astype(np.int32)
    inside = (cu >= 0) & (cu < img_size[0]) & (cv >= 0) & (cv < img_size[1])
    out[cv[inside], cu[inside]] = z[inside]
    return out


def estimate_rigid_transformation(point_1, point_2):
    """Least-squares rotation + translation taking point_1 [n,3] onto point_2 (Kabsch); the translation goes
    into ROW 3 of the returned 4x4, as in the reference."""
    assert point_1.ndim == 2 and point_1.shape[1] == 3
    assert point_2.ndim == 2 and point_2.shape[1] == 3
    c1, c2 = point_1.mean(axis=0), point_2.mean(axis=0)
    H = np.matmul((point_1 - c1[None]).T, point_2 - c2[None])
    U, _, Vt = np.linalg.svd(H)
    R = np.matmul(Vt.T, U.T)
    if np.linalg.det(R) < 0:
        Vt[2, :] *= -1
        R = np.matmul(Vt.T, U.T)
    t = np.matmul(-R, c1.reshape(3, 1)) + c2.reshape(3, 1)
    T = np.eye(4)
    T[:3, :3] = R
    T[3, :3] = t.reshape(3)
    return T


class NyuDatasetGenerator:
    """`<dataset_dir>/<subset>/{joint_data.mat, depth_<cam>_<frame>.png}` -> shards
    `<dataset_dir>/npy-<S>/<subset>/mv_data_<k>_*` of `num_samples_per_segment` frames each."""

    def __init__(self, dataset_dir, subset, image_size=64):
        import scipy.io as sio
        self.cube_size = (300, 300, 300)
        self.img_size = (image_size, image_size)
        self.src_dir = os.path.join(dataset_dir, subset)
        self.npy_dir = os.path.join(dataset_dir, 'npy-%d' % image_size, subset)
        os.makedirs(self.npy_dir, exist_ok=True)
        annotation = sio.loadmat(os.path.join(self.src_dir, 'joint_data.mat'))
        self.camera_num = 3
        self.joints = [annotation['joint_xyz'][c] for c in range(self.camera_num)]
        for joint in self.joints:
            joint[:, :, 1] *= -1
        self.names = [['depth_{}_{:07d}.png'.format(c + 1, i + 1) for i in range(len(self.joints[c]))]
                      for c in range(self.camera_num)]
        self.depth_camera = CameraIntrinsic(fx=588.235, fy=587.084, cx=320, cy=240)
        self.num_sample = len(self.names[0])

    def load_sample_from_file(self, idx):
        from PIL import Image
        dms, annotations = [], []
        for c in range(self.camera_num):
            _, g, b = Image.open(os.path.join(self.src_dir, self.names[c][idx])).split()
            g, b = np.asarray(g, np.int32), np.asarray(b, np.int32)
            dms.append(((g << 8) | b).astype(np.float32))            # depth in mm: high byte in G, low byte in B
            annotations.append(self.joints[c][idx])
        return dms, annotations

    def crop_sample(self, dms, annotations):
        crops = [crop_dm(dm, a[32], self.depth_camera, self.cube_size, self.img_size) for dm, a in zip(dms, annotations)]
        poses = [a - a[32][None] for a in annotations]
        return np.stack(crops), np.stack(poses)

    def estimate_camera_pose(self, cropped_poses):
        return np.stack([np.eye(4) if c == 0 else estimate_rigid_transformation(cropped_poses[c], cropped_poses[0])
                         for c in range(self.camera_num)])

    def prepare_sample(self, idx):
        dms, annotations = self.load_sample_from_file(idx)
        cropped_dms, cropped_poses = self.crop_sample(dms, annotations)
        return cropped_dms, cropped_poses, self.estimate_camera_pose(cropped_poses)

    def create_npy_dataset_from_indices(self, file_name, indices):
        samples = [self.prepare_sample(i) for i in indices]
        dms = np.stack([s[0] for s in samples]).astype(np.float32)
        joint_poses = np.stack([s[1] for s in samples]).astype(np.float32)
        camera_poses = np.stack([s[2] for s in samples]).astype(np.float32)
        write_nyu_shard(os.path.join(self.npy_dir, file_name), dms, joint_poses, camera_poses)

    def create_npy_dataset(self, num_samples_per_segment):
        for k in range(self.num_sample // num_samples_per_segment + 1):
            start = k * num_samples_per_segment
            end = min(start + num_samples_per_segment, self.num_sample)
            if end > start:
                self.create_npy_dataset_from_indices('mv_data_%d' % k, list(range(start, end)))


def main(argv=None):
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument('--nyu_path', type=str, required=True)
    p.add_argument('--image_size', type=int, default=64)
    args = p.parse_args(argv)
    for subset in ('train', 'test'):
        NyuDatasetGenerator(args.nyu_path, subset, args.image_size).create_npy_dataset(1000)


if __name__ == '__main__':
    main()
