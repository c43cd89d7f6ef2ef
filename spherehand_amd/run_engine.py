"""CLI with the reference's flags (network/run_engine.py:10-31; the switches keep
their store_false polarity: passing --mv_projection DISABLES the term).

    python -m spherehand_amd.run_engine --mode Train --dataset_dir /data/nyu/npy-64
    torchrun --nproc-per-node 8 -m spherehand_amd.run_engine --mode Train --synthetic_real 2048
"""
import argparse

from .engine import Engine


def build_parser():
    p = argparse.ArgumentParser()
    for flag in ('synthesize', 'mv_projection', 'mv_consistency', 'collision', 'bone_length', 'prior'):
        p.add_argument('--' + flag, default=True, action='store_false')
    p.add_argument('--temporal', default=False, action='store_true')
    p.add_argument('--mode', default='Test', type=str)
    p.add_argument('--model_dir', default='./trained_model', type=str)
    p.add_argument('--initial_model', type=str)
    p.add_argument('--restore_from_model', type=str)
    p.add_argument('--restore_from_epoch', default=-1, type=int)
    p.add_argument('--num_stacks', default=1, type=int)
    p.add_argument('--epoch', default=75, type=int)
    p.add_argument('--dataset_dir', default=None, type=str)
    p.add_argument('--depth_resample', default=0, type=int)
    p.add_argument('--lr', default=1e-3, type=float)
    p.add_argument('--tag', default='', type=str)
    # additions
    p.add_argument('--image_size', default=64, type=int)
    p.add_argument('--synthetic_real', default=0, type=int,
                   help='use N sphere-rendered multiview samples in place of the NYU shards')
    p.add_argument('--steps_per_epoch', default=None, type=int)
    p.add_argument('--log_every', default=100, type=int)
    p.add_argument('--num_workers', default=0, type=int,
                   help='DataLoader workers for the real shards (the reference: 2, network/engine.py:158-159)')
    p.add_argument('--deterministic', default=False, action='store_true',
                   help='deterministic MIOpen solvers (run-to-run reproducible hourglass gradients)')
    p.add_argument('--miopen_find', default=False, action='store_true',
                   help='let MIOpen time every convolution shape once and keep the fastest solver (~25 s at start-up, -8 %% per step)')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    train_ds = eval_ds = None
    if args.synthetic_real:
        from .datasets import SyntheticMultiviewDataset
        from .hand_model import load_mesh
        mesh = load_mesh()
        train_ds = SyntheticMultiviewDataset(mesh, args.synthetic_real, args.image_size, seed=0)
        eval_ds = SyntheticMultiviewDataset(mesh, max(8, args.synthetic_real // 8), args.image_size, seed=1)
    engine = Engine(args, real_train_dataset=train_ds, real_eval_dataset=eval_ds)
    try:
        return engine.train() if args.mode == 'Train' else engine.eval()
    finally:
        engine.env.close()


if __name__ == '__main__':
    main()
