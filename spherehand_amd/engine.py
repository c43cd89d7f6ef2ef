"""Train / eval step loops around the render-and-fit losses -- the reference's
network/engine.py (Engine :52-477) re-designed for one process per GPU:

  * `torchrun`-style launch (RANK / LOCAL_RANK / WORLD_SIZE); the hourglass is wrapped
    in DistributedDataParallel over RCCL (backend "nccl" on ROCm; "gloo" on CPU for
    tests) with ONE flat gradient bucket (9.24 MB for 1 stack: latency-bound on xGMI,
    so one all-reduce overlapped with the tail of backward beats many small ones);
    GroupNorm only -> no cross-rank statistics.  The render losses shard with the
    batch and need no collective.
  * metrics/loss terms accumulate ON DEVICE; the host reads them every `log_every`
    steps (the reference syncs per term per step, engine.py:40).
  * checkpoints keep the reference's keys {epoch, network_state_dict,
    optimizer_state_dict} (engine.py:438-460), written by rank 0 without the DDP
    `module.` prefix, so either side's files load in the other.
"""
import gc
import json
import os
import random
import string
import time
from enum import Enum

import torch
import torch.distributed as dist
import torch.utils.data as data

from .criterion import HeatmapEstimationNetwork, MultiTaskLoss, average_joint_error, combine_loss, stack_terms
from .hand_model import load_mesh
from .joint_angle import JointAngleDataset, sample_poses_batched
from .pose_denoiser import default_pose_denoiser
from .pose_vae import default_pose_vae
from .util_modules import DepthResample, HandSynthesizer


class Mode(Enum):
    Train = 1
    Eval = 2


class Constant:
    """network/constants.py:10-34 (values only; the mesh comes from hand_model)."""
    depthmap_size = 64
    heatmap_size = 16
    num_joint = 41
    depth_scale = 1.0 / 100.0
    uv_hm_scale = 1.0

    def __init__(self, mesh=None, depthmap_size=64):
        self.mesh = mesh if mesh is not None else load_mesh()
        self.depthmap_size = depthmap_size
        self.heatmap_size = depthmap_size // 4


class RunningAverage:
    """Mean of per-step dicts of 0-dim tensors, kept on the device as ONE [K] tensor (one launch per step; the
    reference converts every term to a python float every step: a device sync each, network/engine.py:40)."""

    def __init__(self):
        self.num, self.keys, self.total = 0, None, None

    def append(self, terms, stacked=None):
        """`stacked` = criterion.stack_terms(terms) if the caller already has it."""
        if stacked is None:
            from .criterion import stack_terms
            stacked = stack_terms(terms)
        stacked = stacked.detach().float()
        if self.total is None or list(terms) != self.keys:
            if self.total is not None:          # the set of terms changed (e.g. another epoch type): fold by name
                old = dict(zip(self.keys, self.total.unbind(0)))
                keys = list(dict.fromkeys(self.keys + list(terms)))
                new = dict(zip(terms, stacked.unbind(0)))
                zero = stacked.new_zeros(())
                self.total = torch.stack([old.get(k, zero) + new.get(k, zero).to(self.total.device) for k in keys])
                self.keys = keys
            else:
                self.keys, self.total = list(terms), stacked.clone()
        else:
            self.total = self.total + stacked.to(self.total.device)
        self.num += 1

    def means(self):
        if self.total is None:
            return {}
        vals = (self.total.double() / self.num).tolist()      # one device -> host copy
        return dict(zip(self.keys, vals))

    def __str__(self):
        return ' '.join('{}: {:.4f}'.format(k, v) for k, v in self.means().items())


class DistEnv:
    """Process-group bookkeeping: rank/world from the launcher's environment."""

    def __init__(self, device=None):
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if device is None:
            device = torch.device('cuda', self.local_rank) if torch.cuda.is_available() else torch.device('cpu')
        self.device = torch.device(device)
        if self.device.type == 'cuda':
            torch.cuda.set_device(self.device)
        self.owns_group = False
        # SHR_FORCE_DIST=1: a single rank still forms a process group and wraps the network in DDP -- the RCCL code
        # path of a multi-GPU job exercised on the one GPU a test box has (tests/test_ddp_gpu.py)
        self.distributed = self.world > 1 or os.environ.get('SHR_FORCE_DIST') == '1'
        if self.distributed and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            if self.device.type == 'cuda':
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
            self.owns_group = True

    @property
    def is_main(self):
        return self.rank == 0

    def wrap(self, module):
        if not self.distributed:
            return module
        kw = dict(bucket_cap_mb=32, gradient_as_bucket_view=True)   # one bucket: 9.24 MB of fp32 grads
        if self.device.type == 'cuda':
            kw['device_ids'] = [self.device.index]
        return torch.nn.parallel.DistributedDataParallel(module, **kw)

    def mean_scalars(self, values):
        """All-reduce a dict of python floats (log intervals only)."""
        if not self.distributed or not values:
            return values
        keys = sorted(values)
        t = torch.tensor([values[k] for k in keys], dtype=torch.float64, device=self.device)
        dist.all_reduce(t)
        return {k: float(v) / self.world for k, v in zip(keys, t)}

    def close(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()


def _unwrap(m):
    return m.module if hasattr(m, 'module') else m


class Engine:
    """opts: the reference's argparse namespace (network/run_engine.py:10-31) plus the
    optional attributes image_size, log_every, real_batch, synt_batch, steps_per_epoch, num_workers.
    Datasets default to the NYU shards under opts.dataset_dir/{train,test}; any
    Dataset yielding (dms [V,S,S] mm, gt_joints [V,36,3], cam [V,4,4], inv_cam [V,4,4])
    can be passed instead."""

    def __init__(self, opts, mesh=None, real_train_dataset=None, real_eval_dataset=None, device=None,
                 prior_loss=None, pose_denoiser=None):
        self.env = DistEnv(device)
        dev = self.env.device
        if getattr(opts, 'deterministic', False):
            # MIOpen's default backward-weights solvers accumulate with float atomics: the hourglass gradient
            # then varies run to run (up to 3e-3 of its largest entry on MI355X, tools/debug_determinism.py).
            # This project's own kernels are deterministic by construction (no float atomics anywhere).
            torch.backends.cudnn.deterministic = True
        if getattr(opts, 'miopen_find', False) and dev.type == 'cuda':
            # MIOpen's find mode: every convolution shape is timed once and the fastest solver kept, instead of the
            # immediate-mode heuristic.  Reference-sized step on MI355X: 9.3-9.6 -> 8.7-8.8 ms, for ~25 s at start-up
            # (tools/exp_miopen_find.py).  Off by default, like the reference (which never sets cudnn.benchmark).
            torch.backends.cudnn.benchmark = True
        S = getattr(opts, 'image_size', 64)
        self.constant = Constant(mesh, S)
        c = self.constant
        self.network = HeatmapEstimationNetwork(c.heatmap_size, c.depth_scale, c.num_joint, opts.num_stacks).to(dev)
        if dev.type == 'cuda':
            # MIOpen's NHWC kernels: hourglass fwd+bwd on 123 crops 25.0 -> 14.0 ms on MI355X (fp32, same math)
            self.network = self.network.to(memory_format=torch.channels_last)
        self.ddp_network = self.env.wrap(self.network)
        if opts.prior and prior_loss is None:
            prior_loss = default_pose_vae()        # PoseVae(41*3, 32, 'mesh/model/pose_vae.pth'), create_network...:164
        self.criterion = MultiTaskLoss(opts.synthesize, opts.mv_projection, opts.mv_consistency, opts.temporal,
                                       prior_loss if opts.prior else None, opts.collision, opts.bone_length, c,
                                       image_size=S, heatmap_size=c.heatmap_size).to(dev)
        self.hand_synthesizer = None
        if opts.synthesize and dev.type == 'cuda':
            self.hand_synthesizer = HandSynthesizer(c.mesh, S, c.heatmap_size, c.uv_hm_scale, c.depth_scale).to(dev)
            self.hand_synthesizer.seed_offset = self.env.rank      # ranks seeded alike (identical initial weights) still draw their own noise
        # network/engine.py:71-73: the frozen palm re-predictor the Eval metric goes through
        # (loaded on the first evaluation step: a training-only run needs neither the module nor its weight file)
        self._pose_denoiser = pose_denoiser.to(dev).eval() if pose_denoiser is not None else None
        self.depth_sampler = DepthResample(0.95, opts.depth_resample).to(dev) if getattr(opts, 'depth_resample', 0) else None
        self.num_stacks = opts.num_stacks
        self.temporal_smooth = opts.temporal
        self.mode = Mode.Train if opts.mode == 'Train' else Mode.Eval
        self.epoch = opts.epoch
        self.log_every = getattr(opts, 'log_every', 100)
        self.real_batch = getattr(opts, 'real_batch', 25)
        self.synt_batch = getattr(opts, 'synt_batch', 48)
        self.steps_per_epoch = getattr(opts, 'steps_per_epoch', None)
        self.num_workers = int(getattr(opts, 'num_workers', 0) or 0)
        # network/engine.py:100: Adam(lr, weight_decay=1e-5); on the GPU the single-kernel (fused) implementation
        on_gpu = self.env.device.type == 'cuda'
        self.optimizer = torch.optim.Adam(self.network.parameters(), lr=opts.lr, weight_decay=1e-5,
                                          **({'fused': True} if on_gpu else {}))
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=max(1, self.epoch // 3), gamma=0.1)
        self.starting_epoch = 0

        self.model_dir = opts.model_dir
        if getattr(opts, 'restore_from_model', None) is not None:
            self.model_name = opts.restore_from_model
            self.model_path = os.path.join(self.model_dir, self.model_name)
            self.load_model(opts.restore_from_epoch)
        else:
            name = [getattr(opts, 'tag', '') + ''.join(random.choice(string.ascii_letters + string.digits)
                                                       for _ in range(6))]
            if self.env.distributed:
                dist.broadcast_object_list(name, src=0)
            self.model_name = name[0]
            self.model_path = os.path.join(self.model_dir, self.model_name)
        if self.env.is_main:
            os.makedirs(self.model_path, exist_ok=True)
            with open(os.path.join(self.model_path, 'loss_weights.txt'), 'w') as f:
                json.dump(self.criterion.weights, f)
        if getattr(opts, 'initial_model', None) is not None:
            self.load_model(opts.initial_model)
        self.log_file = os.path.join(self.model_path, 'log.txt')

        if real_train_dataset is None and getattr(opts, 'dataset_dir', None):
            from .datasets import create_nyu_dataset
            try:
                real_train_dataset = create_nyu_dataset([os.path.join(opts.dataset_dir, 'train')])
                real_eval_dataset = create_nyu_dataset(os.path.join(opts.dataset_dir, 'test'))
            except FileNotFoundError as e:
                self.log('skipped: asset missing ({})'.format(e))
        self.real_train_dataset, self.real_eval_dataset = real_train_dataset, real_eval_dataset
        self.synt_dataset = JointAngleDataset()
        self.with_synt = bool(opts.synthesize) and self.hand_synthesizer is not None
        self.with_real = any([opts.mv_projection, opts.mv_consistency, opts.temporal, opts.prior, opts.collision,
                              opts.bone_length]) and real_train_dataset is not None

    @property
    def pose_denoiser(self):
        if self._pose_denoiser is None:
            self._pose_denoiser = default_pose_denoiser().to(self.env.device).eval()
        return self._pose_denoiser

    # ------------------------------------------------------------------ utilities
    def log(self, msg):
        if self.env.is_main:
            print(msg, flush=True)
            if hasattr(self, 'log_file'):
                with open(self.log_file, 'a') as f:
                    f.write(msg + '\n')

    def curr_lr(self):
        return self.optimizer.param_groups[0]['lr']

    def _real_loader(self, dataset, batch_size, train):
        sampler = None
        shuffle = train and not self.temporal_smooth
        if self.env.world > 1:
            sampler = data.distributed.DistributedSampler(dataset, self.env.world, self.env.rank, shuffle=shuffle)
            shuffle = False
        # (the reference: num_workers=2, network/engine.py:158-159, :326-327; opts.num_workers, default 0: the shards are
        # memory-mapped and a batch is a slice copy, cheaper than a worker hand-over at these batch sizes)
        return data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler,
                               num_workers=self.num_workers, persistent_workers=self.num_workers > 0, drop_last=train)

    def _pose_iter(self, batch_size):
        """Pose batches of the synthetic branch (the reference: a DataLoader over JointAngleDataset, shuffle=True,
        one worker, network/engine.py:328-329): one vectorised draw per step from a per-rank generator -- the same
        distribution as JointAngleDataset[i], without its ~55 torch.rand(1) calls per pose on the step's host path."""
        g = torch.Generator().manual_seed(1234 + self.env.rank)
        while True:
            yield sample_poses_batched(batch_size, generator=g)

    def _prepare_real(self, batch):
        dev, c = self.env.device, self.constant
        real_dms, gt_joints, cam, inv_cam = (torch.as_tensor(t).to(dev, non_blocking=True).float() for t in batch)
        orig = real_dms                                        # unscaled mm, background 100 (engine.py:336,358)
        scaled = real_dms * c.depth_scale
        if self.depth_sampler is not None:
            V, S = scaled.shape[1], scaled.shape[-1]
            scaled = self.depth_sampler(scaled.view(-1, S, S)).view(-1, V, S, S)
        return scaled, orig, gt_joints, cam, inv_cam

    # ------------------------------------------------------------------ steps
    def step(self, real_batch=None, pose_parameter=None, train=True, is_mv=True):
        """One optimisation (or evaluation) step.  Returns (loss_terms, metrics,
        result, projected_dms); tensors stay on the device.  The metric follows the
        reference: in training the raw network output of every view (engine.py:207-210,
        :369-370), in evaluation VIEW 0 ONLY after the pose denoiser (engine.py:200-206)
        -- that is the "NYU mean 3D joint error"."""
        net = self.ddp_network if train else self.network
        synt_target = real_target = None
        kwargs = {}
        if pose_parameter is not None:
            synt_dms, uv_hms, d_hms, xyz = self.hand_synthesizer(pose_parameter.to(self.env.device))
            if self.depth_sampler is not None and real_batch is not None:
                # only the mixed epoch resamples the synthetic crops (engine.py:352-353; _epoch_with_synt does not)
                synt_dms = self.depth_sampler(synt_dms).squeeze(1)
            kwargs['synt_dms'] = synt_dms
            synt_target = {'uv_hms': uv_hms, 'd_hms': d_hms, 'xyz_pts': xyz}
        gt_joints = None
        if real_batch is not None:
            scaled, orig, gt_joints, cam, inv_cam = self._prepare_real(real_batch)
            kwargs['real_dms'] = scaled
            real_target = {'real_dms': orig, 'camera_poses': cam, 'inv_camera_poses': inv_cam, 'is_mv': is_mv}
        if train:
            self.optimizer.zero_grad(set_to_none=True)
        result = net(**kwargs)
        loss_terms, projected = self.criterion(result, synt_target=synt_target, real_target=real_target)
        metrics = {}
        if gt_joints is not None:
            est = result['real_xyz'][-1].detach()
            if not train:
                with torch.no_grad():
                    metrics['avg_joint_error'] = average_joint_error(
                        gt_joints[:, 0].unsqueeze(1), self.pose_denoiser(est[:, 0]).unsqueeze(1))
            else:
                metrics['avg_joint_error'] = average_joint_error(gt_joints, est)
        stacked = stack_terms(loss_terms)           # one launch: the total loss below and the running averages share it
        self._last_stacked = stacked.detach()
        if train:
            combine_loss(loss_terms, stacked).backward()
            self.optimizer.step()
        return loss_terms, metrics, result, projected

    def _run_epoch(self, mode, epoch, with_real, with_synt):
        train = mode is Mode.Train
        self.network.train(train)
        dataset = self.real_train_dataset if train else self.real_eval_dataset
        losses, metrics_avg = RunningAverage(), RunningAverage()
        t_prev = time.time()
        if with_real:
            bs = self.real_batch if with_synt else 8
            loader = self._real_loader(dataset, bs, train)
            if self.env.world > 1 and hasattr(loader.sampler, 'set_epoch'):
                loader.sampler.set_epoch(epoch)
            real_it = iter(loader)
            n_steps = len(loader)
        else:
            real_it, n_steps = None, 1000 * self.num_stacks
        if self.steps_per_epoch:
            n_steps = min(n_steps, self.steps_per_epoch)
        pose_it = self._pose_iter(self.synt_batch if with_real else 128 // self.num_stacks) if with_synt else None
        # the epoch loops drop the projected depth maps (the reference hands them to its visualiser only): the loss
        # does not materialise them here -- half of the render-and-compare kernel's HBM bytes, and in the same-view
        # mode a whole rasterizer launch per stack (MutualProjectionLoss.return_projections)
        mp = getattr(self.criterion, 'mv_projection_loss', None)
        keep_projections = None if mp is None else mp.return_projections
        if mp is not None:
            mp.return_projections = False
        try:
            return self._epoch_steps(train, epoch, n_steps, real_it, pose_it, with_synt, losses, metrics_avg, t_prev)
        finally:
            if mp is not None:
                mp.return_projections = keep_projections

    def _epoch_steps(self, train, epoch, n_steps, real_it, pose_it, with_synt, losses, metrics_avg, t_prev):
        for it in range(n_steps):
            real = next(real_it) if real_it is not None else None
            pose = next(pose_it) if pose_it is not None else None
            with torch.set_grad_enabled(train):
                terms, metrics, _, _ = self.step(real, pose, train, is_mv=(it < 1500) if with_synt else True)
            losses.append(terms, self._last_stacked)
            if metrics:
                metrics_avg.append(metrics)
            if it % self.log_every == 0:
                self.log('[{}-{}]: metric: {}, loss: {}, lr: {}, time: {:.2f}s'.format(
                    epoch, it, self.env.mean_scalars(metrics_avg.means()), self.env.mean_scalars(losses.means()),
                    self.curr_lr(), time.time() - t_prev))
                t_prev = time.time()
        summary = {'metric': self.env.mean_scalars(metrics_avg.means()), 'loss': self.env.mean_scalars(losses.means())}
        self.log('[epoch: {}]: metric: {}, loss: {}, lr: {}'.format(epoch, summary['metric'], summary['loss'],
                                                                  self.curr_lr()))
        return summary

    def _epoch_with_real(self, mode, epoch):
        return self._run_epoch(mode, epoch, True, False)

    def _epoch_with_synt(self, mode, epoch):
        return self._run_epoch(mode, epoch, False, True)

    def _epoch_with_both(self, mode, epoch):
        return self._run_epoch(mode, epoch, True, True)

    # ------------------------------------------------------------------ checkpoints
    def save_model(self, epoch):
        if not self.env.is_main:
            return
        torch.save({'epoch': epoch, 'network_state_dict': _unwrap(self.network).state_dict(),
                    'optimizer_state_dict': self.optimizer.state_dict()},
                   os.path.join(self.model_path, 'model_{}.pth'.format(epoch)))

    def load_model(self, epoch):
        if isinstance(epoch, int):
            pth_path = os.path.join(self.model_path, 'model_{}.pth'.format(epoch))
        elif isinstance(epoch, str):
            pth_path = epoch
        else:
            raise ValueError
        if not os.path.exists(pth_path):
            raise FileNotFoundError('skipped: asset missing ({})'.format(pth_path))
        check_point = torch.load(pth_path, map_location=self.env.device)
        self.network.load_state_dict(check_point['network_state_dict'])
        if isinstance(epoch, int):
            self.optimizer.load_state_dict(check_point['optimizer_state_dict'])
            self.starting_epoch = check_point['epoch']
            for _ in range(self.starting_epoch):
                self.scheduler.step()

    # ------------------------------------------------------------------ drivers
    def train(self):
        last = None
        # (the modules, data sets and torch itself are here to stay: out of the garbage collector's way -- a full
        # collection walks ~1e6 objects, a 40-ms stall of the launching thread every few hundred steps)
        gc.collect()
        gc.freeze()
        for epoch in range(self.starting_epoch, self.epoch):
            if self.with_real and self.with_synt:
                last = self._epoch_with_both(Mode.Train, epoch)
            elif self.with_synt:
                last = self._epoch_with_synt(Mode.Train, epoch)
            elif self.with_real:
                last = self._epoch_with_real(Mode.Train, epoch)
            else:
                raise RuntimeError('nothing to train on: no real dataset and the synthetic branch is off')
            self.scheduler.step()
            self.save_model(-1)
            self.save_model(epoch)
        return last

    def eval(self):
        if self.real_eval_dataset is None:
            raise RuntimeError('skipped: asset missing (no evaluation dataset)')
        with torch.no_grad():
            return self._epoch_with_real(Mode.Eval, 0)
