"""CPU, world_size 2, gloo: the engine's data-parallel step (DDP gradient all-reduce
of the hourglass, one bucket) reproduces the single-process global-batch step."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, out, model_dir):
    from conftest import run_torchrun
    env = dict(os.environ, OMP_NUM_THREADS="2")
    run_torchrun(world, [os.path.join(ROOT, "tests", "ddp_worker.py"), out, model_dir], env=env, timeout=600)
    return torch.load(out)


def test_two_rank_step_equals_single_process(tmp_path):
    one = launch(1, str(tmp_path / "w1.pt"), str(tmp_path / "m1"))
    two = launch(2, str(tmp_path / "w2.pt"), str(tmp_path / "m2"))
    assert one["world"] == 1 and two["world"] == 2
    gmax = max(v.abs().max().item() for v in one["grads"].values())
    for k in one["grads"]:
        assert (one["grads"][k] - two["grads"][k]).abs().max().item() <= 1e-5 * gmax + 1e-8, k
    for k in one["params"]:          # after one Adam step
        assert (one["params"][k] - two["params"][k]).abs().max().item() <= 2e-5, k
    for k in one["terms"]:
        assert abs(one["terms"][k] - two["terms"][k]) <= 1e-5 * max(1.0, abs(one["terms"][k])), k
    # rank 0 wrote a checkpoint with the reference's keys and without the DDP prefix
    ckpt_dir = [d for d in os.listdir(tmp_path / "m2")][0]
    ckpt = torch.load(tmp_path / "m2" / ckpt_dir / "model_0.pth")
    assert set(ckpt) == {"epoch", "network_state_dict", "optimizer_state_dict"}
    assert all(not k.startswith("module.") for k in ckpt["network_state_dict"])
    assert "hg.conv1.weight" in ckpt["network_state_dict"]
