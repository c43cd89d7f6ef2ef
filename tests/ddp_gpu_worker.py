"""Worker for tests/test_ddp_gpu.py: `world` processes share cuda:0 (backend gloo -- RCCL refuses
two ranks on one device; the collective is the same flat-bucket all-reduce DDP issues over RCCL)
and each takes its shard of a FIXED global batch through ONE Engine.step with every kernel-backed
render / fit term ON: HandSynthesizer (FK, skinning, fused triangle raster, noise, heat-map paint),
MutualProjectionLoss (view projection, fused render-and-compare, data->model), the consistency
median, collision + bone length.  Rank 0 saves the averaged gradients, the parameters after Adam
and the loss terms."""
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GLOBAL_REAL, GLOBAL_SYNT = 8, 12


def make_opts(model_dir):
    on = os.environ.get('SHR_DDP_TERMS', 'synthesize,mv_projection,mv_consistency,collision,bone_length').split(',')
    return SimpleNamespace(synthesize='synthesize' in on, mv_projection='mv_projection' in on,
                           mv_consistency='mv_consistency' in on, temporal=False, prior=False,
                           collision='collision' in on, bone_length='bone_length' in on, mode='Train', model_dir=model_dir, initial_model=None,
                           restore_from_model=None, restore_from_epoch=-1, num_stacks=1, epoch=3, dataset_dir=None,
                           depth_resample=0, lr=1e-3, tag='ddpgpu', image_size=64, log_every=1,
                           deterministic=True)


class ShardedSynth:
    """The HIP HandSynthesizer run on the WHOLE global pose batch under a fixed seed (identical on every
    rank: same device type, same draws), then this rank's rows."""

    def __init__(self, synth, rank, world):
        from spherehand_amd.joint_angle import sample_poses
        poses = sample_poses(GLOBAL_SYNT, seed=21).cuda()
        torch.manual_seed(99)
        synth.seed_offset = 0          # (Engine gives every rank its own noise stream: here every rank renders the SAME global batch)
        out = synth(poses)
        n = GLOBAL_SYNT // world
        self.out = tuple(o[rank * n:(rank + 1) * n].contiguous() for o in out)

    def __call__(self, pose):
        return self.out


def run(out_path, model_dir):
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.engine import Engine
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    # SHR_DDP_BACKEND=nccl + SHR_FORCE_DIST=1: ONE rank that still forms an RCCL process group and wraps the network in
    # DDP (Engine's DistEnv does both) -- the multi-GPU job's code path on the one GPU a test box has
    backend = os.environ.get('SHR_DDP_BACKEND', 'gloo')
    if world > 1 and backend == 'gloo':
        dist.init_process_group('gloo')
    mesh = hand_model.load_mesh()
    torch.manual_seed(7)                                   # identical initial weights on every rank
    eng = Engine(make_opts(model_dir), mesh=mesh, device='cuda:0')
    assert eng.env.world == world and eng.env.rank == rank
    eng.network.resize_dm = None                           # no random scale augmentation: a fixed global batch
    eng.hand_synthesizer = ShardedSynth(eng.hand_synthesizer, rank, world)
    # CollisionLoss is a SUM over the batch (mesh/render.py:176): DDP's gradient averaging turns a rank's sum
    # into sum/world, so a fixed GLOBAL batch split over `world` ranks needs the weight x world to reproduce the
    # single-process gradient (with the per-rank batch fixed -- weak scaling -- it keeps the reference's scale)
    eng.criterion.weights['collision'] *= world
    ds = SyntheticMultiviewDataset(mesh, GLOBAL_REAL, 64, seed=4)
    n = GLOBAL_REAL // world
    sl = slice(rank * n, (rank + 1) * n)
    real = (ds.dms[sl], ds.gt[sl], ds.cam[sl], ds.inv_cam[sl])
    eng.network.train()
    terms, metrics, _, _ = eng.step(real, torch.zeros(1, 26), train=True, is_mv=True)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().clone() for k, p in eng.network.named_parameters()}
    params = {k: p.detach().cpu().clone() for k, p in eng.network.named_parameters()}
    vals = {k: float(v) for k, v in terms.items()}
    if 'collision' in vals:
        vals['collision'] /= world                              # back to this rank's share of the global sum ...
    means = eng.env.mean_scalars(vals)
    if 'collision' in means:
        means['collision'] *= world                             # ... whose mean x world is the global sum
    metric = eng.env.mean_scalars({k: float(v) for k, v in metrics.items()})
    if eng.env.is_main:
        torch.save({'grads': grads, 'params': params, 'terms': means, 'metric': metric, 'world': eng.env.world,
                    'ddp': type(eng.ddp_network).__name__,
                    'backend': dist.get_backend() if dist.is_initialized() else None}, out_path)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    run(sys.argv[1], sys.argv[2])
