"""Worker for tests/test_ddp_cpu.py: run with torch.distributed.run (gloo, CPU).
Each rank takes its shard of a fixed global synthetic batch, does ONE Engine.step
through DistributedDataParallel and rank 0 saves the averaged gradients and the
updated parameters."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_opts(model_dir):
    return SimpleNamespace(synthesize=True, mv_projection=False, mv_consistency=False, temporal=False, prior=False,
                           collision=False, bone_length=False, mode='Train', model_dir=model_dir, initial_model=None,
                           restore_from_model=None, restore_from_epoch=-1, num_stacks=1, epoch=3, dataset_dir=None,
                           depth_resample=0, lr=1e-3, tag='ddp', image_size=64, log_every=1)


def global_batch(n=8):
    g = torch.Generator().manual_seed(123)
    return (torch.rand(n, 64, 64, generator=g), torch.rand(n, 41, 16, 16, generator=g),
            torch.rand(n, 41, 16, 16, generator=g), torch.randn(n, 41, 4, generator=g) * 20)


class ShardSynth:
    """Stands in for HandSynthesizer (which needs the GPU): returns this rank's rows
    of the fixed global batch, whatever pose vector it is given."""

    def __init__(self, rank, world):
        dms, uv, d, xyz = global_batch()
        sl = slice(rank * len(dms) // world, (rank + 1) * len(dms) // world)
        self.out = (dms[sl], uv[sl], d[sl], xyz[sl])

    def __call__(self, pose):
        return self.out


def run(out_path, model_dir):
    from spherehand_amd.engine import Engine
    from spherehand_amd import hand_model
    torch.manual_seed(7)                                   # identical initial weights on every rank
    eng = Engine(make_opts(model_dir), mesh=hand_model.load_mesh(), device='cpu')
    eng.hand_synthesizer = ShardSynth(eng.env.rank, eng.env.world)
    eng.network.train()
    terms, _, _, _ = eng.step(None, torch.zeros(1, 26), train=True)
    grads = {k: p.grad.clone() for k, p in eng.network.named_parameters()}
    params = {k: p.detach().clone() for k, p in eng.network.named_parameters()}
    means = eng.env.mean_scalars({k: float(v) for k, v in terms.items()})
    if eng.env.is_main:
        torch.save({'grads': grads, 'params': params, 'terms': means, 'world': eng.env.world}, out_path)
        eng.save_model(0)
    eng.env.close()


if __name__ == '__main__':
    run(sys.argv[1], sys.argv[2])
