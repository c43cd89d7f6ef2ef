"""Data either side of the render path.

    NyuShardDataset / create_nyu_dataset   reader of the reference's NYU shard format
                                           (dataset/nyu_dataset.py:9-50, written by
                                           dataset/nyu_generator.py:89-119)
    write_nyu_shard                        writer of the same format (tests, tools)
    SyntheticMultiviewDataset              stand-in "real" multiview data rendered from the
                                           sphere model (the NYU set is not shipped)
"""
import os
import pickle

import numpy as np
import torch
import torch.utils.data as data

from .criterion import REAL_KEY_POINTS, SYNT_KEY_POINTS


class NyuShardDataset(data.Dataset):
    """One shard `<path>_{shape.pkl, dms.bat, joint_poses.npy, camera_poses.npy}`:
    dms f32 memmap [N,V,S,S] (mm, background 100), joint_poses [N,V,36,3], camera_poses
    [N,V,4,4]; items are (dms, joints, camera_poses, inverse camera_poses)."""

    def __init__(self, file_path, transform=None):
        super().__init__()
        with open(file_path + '_shape.pkl', 'rb') as f:
            shape_info = pickle.load(f)
        self.dms = np.memmap(file_path + '_dms.bat', dtype='float32', mode='r', shape=tuple(shape_info['dms']))
        self.joint_poses = np.load(file_path + '_joint_poses.npy')
        self.camera_poses = np.load(file_path + '_camera_poses.npy')
        self.inv_camera_poses = np.linalg.inv(self.camera_poses.reshape(-1, 4, 4)).reshape(
            self.camera_poses.shape).astype(self.camera_poses.dtype)
        self.transform = transform

    def __getitem__(self, index):
        item = (np.asarray(self.dms[index]), self.joint_poses[index], self.camera_poses[index],
                self.inv_camera_poses[index])
        return item if self.transform is None else self.transform(*item)

    def __len__(self):
        return self.joint_poses.shape[0]


def create_nyu_dataset(file_dir):
    """ConcatDataset over `mv_data_0, mv_data_1, ...` in each directory."""
    dirs = file_dir if isinstance(file_dir, list) else [file_dir]
    shards = []
    for d in dirs:
        idx = 0
        while os.path.exists(os.path.join(d, 'mv_data_%d_shape.pkl' % idx)):
            shards.append(NyuShardDataset(os.path.join(d, 'mv_data_%d' % idx)))
            idx += 1
    if not shards:
        raise FileNotFoundError('no mv_data_<k>_shape.pkl shard under %s' % dirs)
    return data.ConcatDataset(shards)


def write_nyu_shard(file_path, dms, joint_poses, camera_poses):
    dms = np.ascontiguousarray(dms, np.float32)
    with open(file_path + '_shape.pkl', 'wb') as f:
        pickle.dump({'dms': dms.shape, 'joint_poses': joint_poses.shape, 'camera_poses': camera_poses.shape}, f,
                    protocol=pickle.HIGHEST_PROTOCOL)
    fp = np.memmap(file_path + '_dms.bat', dtype='float32', mode='w+', shape=dms.shape)
    fp[:] = dms[:]
    fp.flush()
    np.save(file_path + '_joint_poses.npy', np.asarray(joint_poses, np.float32))
    np.save(file_path + '_camera_poses.npy', np.asarray(camera_poses, np.float32))


def random_rotations(n, max_deg, generator=None):
    """n rotation matrices about random axes by angles in +-max_deg (Rodrigues)."""
    axis = torch.randn(n, 3, generator=generator)
    axis = axis / axis.norm(dim=1, keepdim=True)
    ang = (torch.rand(n, generator=generator) * 2 - 1) * (max_deg * np.pi / 180)
    K = torch.zeros(n, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -axis[:, 2], axis[:, 1], axis[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -axis[:, 0], -axis[:, 1], axis[:, 0]
    s, c = torch.sin(ang).view(n, 1, 1), torch.cos(ang).view(n, 1, 1)
    return torch.eye(3).unsqueeze(0) + s * K + (1 - c) * (K @ K)


class SyntheticMultiviewDataset(data.Dataset):
    """`num_samples` hands seen from `num_views` cameras (rotations <= 30 degrees about
    the crop centre), rendered with the sphere model on the GPU at construction.
    Items match the NYU shards: (dms [V,S,S] mm/background 100, gt joints [V,36,3],
    camera_poses [V,4,4], inv_camera_poses [V,4,4]); gt joint REAL_KEY_POINTS[k] is the
    sphere centre SYNT_KEY_POINTS[k], so the NYU metric is exact for a perfect fit."""

    def __init__(self, mesh, num_samples, image_size, num_views=3, seed=0, device='cuda'):
        super().__init__()
        from .joint_angle import sample_poses
        from .kinematicsTransformation import HandTransformationMat, keypoint_skinning
        from .multiview_utility import MutualProjection
        g = torch.Generator().manual_seed(seed)
        state = torch.get_rng_state()
        params = sample_poses(num_samples, seed=seed)
        torch.set_rng_state(state)
        R = random_rotations(num_samples * num_views, 30.0, g).view(num_samples, num_views, 3, 3)
        cam = torch.eye(4).repeat(num_samples, num_views, 1, 1)
        cam[:, :, :3, :3] = R                                    # view -> canonical
        inv_cam = torch.eye(4).repeat(num_samples, num_views, 1, 1)
        inv_cam[:, :, :3, :3] = R.transpose(-1, -2)
        dev = torch.device(device)
        fk = HandTransformationMat([b['offset_matrix'].astype(np.float32) for b in mesh['bones']]).to(dev)
        lbs = keypoint_skinning(mesh).to(dev)
        mp = MutualProjection(image_size, mesh).to(dev)
        with torch.no_grad():
            canonical = lbs(fk(params.to(dev)))[:, :, :3]                                   # [N,41,3]
            joints = torch.einsum('nvij,nkj->nvki', inv_cam[:, :, :3, :3].to(dev), canonical).contiguous()  # per view
            # (einsum returns a permuted view: as a leaf of the fitting loop it cost a layout copy in every forward and
            # another in AccumulateGrad -- 9 us of a 300-us loss step)
            dms = []
            for s in range(0, num_samples, 64):
                d, _ = mp(cam[s:s + 64].to(dev), inv_cam[s:s + 64].to(dev), joints[s:s + 64].contiguous())
                dms.append(torch.diagonal(d, dim1=1, dim2=2).permute(0, 3, 1, 2))         # view j in view j
            self.dms = torch.cat(dms).cpu()
        gt = torch.zeros(num_samples, num_views, 36, 3)
        gt[:, :, REAL_KEY_POINTS] = joints[:, :, SYNT_KEY_POINTS].cpu()
        self.gt, self.cam, self.inv_cam, self.joints = gt, cam, inv_cam, joints.cpu()

    def __getitem__(self, i):
        return self.dms[i], self.gt[i], self.cam[i], self.inv_cam[i]

    def __len__(self):
        return self.dms.shape[0]
