"""CPU: the NYU crop generator (spherehand_amd/nyu_generator.py, parameterised by the crop size) against the
imported reference run on the same synthetic NYU-layout frames (tests/golden/make_goldens_nyu.py):
bit-exact crops, joints and camera poses, and the shards it writes read back through the shard reader."""
import os

import numpy as np

from conftest import golden
from nyu_synth import write_synthetic_nyu


def test_generator_matches_reference_bit_for_bit(tmp_path):
    from spherehand_amd.datasets import create_nyu_dataset
    from spherehand_amd.nyu_generator import NyuDatasetGenerator
    g = golden("g10_nyu_generator.npz")
    root = str(tmp_path)
    write_synthetic_nyu(root, "train", frames=3, seed=0)
    gen = NyuDatasetGenerator(root, "train", image_size=64)
    assert gen.num_sample == 3
    gen.create_npy_dataset(2)                       # two shards: frames 0-1 and 2
    ds = create_nyu_dataset(os.path.join(root, "npy-64", "train"))
    assert len(ds) == 3
    dms = np.stack([ds[i][0] for i in range(3)])
    joints = np.stack([ds[i][1] for i in range(3)])
    cams = np.stack([ds[i][2] for i in range(3)])
    assert np.array_equal(dms.view(np.uint32), g["dms64"].view(np.uint32))
    assert np.array_equal(joints.view(np.uint32), g["joint_poses"].view(np.uint32))
    assert np.array_equal(cams.view(np.uint32), g["camera_poses"].view(np.uint32))
    assert (dms < 100).sum() > 5000 and np.all(joints[:, :, 32] == 0)          # crops centred on joint 32
    assert np.array_equal(cams[:, 0], np.broadcast_to(np.eye(4, dtype=np.float32), (3, 4, 4)))
    assert np.abs(cams[:, 1:, 3, :3]).max() > 1 and np.all(cams[:, :, :3, 3] == 0)   # translation in ROW 3
    # the same frames at 128 x 128
    gen128 = NyuDatasetGenerator(root, "train", image_size=128)
    frames, ann = gen128.load_sample_from_file(1)
    crops, _ = gen128.crop_sample(frames, ann)
    assert crops.shape == (3, 128, 128)
    assert np.array_equal(crops.astype(np.float32).view(np.uint32), g["dms128_frame1"].view(np.uint32))
    assert gen128.npy_dir.endswith(os.path.join("npy-128", "train"))


def test_rigid_transformation_recovers_a_known_motion():
    from spherehand_amd.nyu_generator import estimate_rigid_transformation
    rs = np.random.RandomState(0)
    p = rs.normal(0, 40, (36, 3))
    a = 0.4
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    t = np.array([5.0, -3.0, 12.0])
    T = estimate_rigid_transformation(p, p @ R.T + t)
    assert np.allclose(T[:3, :3], R, atol=1e-9) and np.allclose(T[3, :3], t, atol=1e-9)
    mirrored = p * np.array([1, 1, -1])             # a reflection is not a rigid motion: det must stay +1
    assert np.linalg.det(estimate_rigid_transformation(p, mirrored)[:3, :3]) > 0
