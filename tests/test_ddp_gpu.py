"""GPU, world_size 2 on ONE MI355X (gloo; RCCL refuses duplicate devices): the sharded render + fit
step -- every HIP-backed loss term on -- reproduces the single-process global-batch step."""
import pytest

from test_ddp_cpu import free_port  # noqa: F401  (same launcher helpers)
import os
import subprocess
import sys

from conftest import ROOT

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def launch(world, out, model_dir, **extra):
    from conftest import run_torchrun
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    run_torchrun(world, [os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), out, model_dir], env=env, timeout=900)
    return torch.load(out)


def _params_agree(a, b, gmax):
    """After one Adam step (lr 1e-3) the first update is lr * g / (|g| + 1e-8): where the gradient is well above the
    rounding noise of the two runs the weights agree closely; a weight whose gradient IS noise (|g| <~ 1e-7) moves by
    up to lr in a direction the noise decides -- bounded by 2 lr, not comparable further."""
    for k in a["params"]:
        d = (a["params"][k] - b["params"][k]).abs()
        solid = a["grads"][k].abs() > 1e-5 * gmax
        assert d[solid].max().item() <= 2e-5 if solid.any() else True, k
        assert d.max().item() <= 2.1e-3, k


def test_two_rank_render_fit_step_equals_single_process(tmp_path):
    one = launch(1, str(tmp_path / "w1.pt"), str(tmp_path / "m1"))
    two = launch(2, str(tmp_path / "w2.pt"), str(tmp_path / "m2"))
    assert one["world"] == 1 and two["world"] == 2 and two["ddp"] == "DistributedDataParallel"
    assert {"synt_uv", "synt_d", "mv_projection", "mv_consistency", "uv_hm_mean", "collision", "bone_length"} \
        <= set(one["terms"])
    assert one["terms"]["mv_projection"] > 0 and one["terms"]["collision"] >= 0
    for k in one["terms"]:           # loss terms: per-crop results are identical, only the summation order differs
        assert abs(one["terms"][k] - two["terms"][k]) <= 1e-5 * max(1.0, abs(one["terms"][k])), k
    assert abs(one["metric"]["avg_joint_error"] - two["metric"]["avg_joint_error"]) <= 1e-4
    gmax = max(v.abs().max().item() for v in one["grads"].values())
    worst = max((one["grads"][k] - two["grads"][k]).abs().max().item() for k in one["grads"])
    print("gmax %.4g worst grad diff %.4g" % (gmax, worst))
    for k in one["grads"]:           # averaged gradients (MIOpen may pick another algorithm at half the batch)
        assert (one["grads"][k] - two["grads"][k]).abs().max().item() <= 1e-5 * gmax + 1e-8, k
    _params_agree(one, two, gmax)


def test_one_rank_rccl_ddp_step_equals_the_plain_step(tmp_path):
    """RCCL executed on hardware: ONE rank forms an `nccl` process group on the MI355X (Engine's DistEnv:
    init_process_group("nccl", device_id=...)), the network is wrapped in DistributedDataParallel with the flat
    9.24-MB bucket, the name broadcast and the scalar all-reduce run, and one Engine.step's bucket all-reduce goes
    through RCCL (/root/reference network/engine.py:366-376 under DDP).  A one-rank all-reduce is the identity:
    the step equals the unwrapped single-process step."""
    plain = launch(1, str(tmp_path / "p.pt"), str(tmp_path / "mp"))
    rccl = launch(1, str(tmp_path / "r.pt"), str(tmp_path / "mr"), SHR_FORCE_DIST="1", SHR_DDP_BACKEND="nccl")
    assert plain["ddp"] != "DistributedDataParallel" and plain["backend"] is None
    assert rccl["ddp"] == "DistributedDataParallel" and rccl["backend"] == "nccl" and rccl["world"] == 1
    # (two processes: MIOpen's convolution backward is not bit-reproducible from run to run, the bars are the two-rank test's)
    for k in plain["terms"]:
        assert abs(plain["terms"][k] - rccl["terms"][k]) <= 1e-5 * max(1.0, abs(plain["terms"][k])), k
    gmax = max(v.abs().max().item() for v in plain["grads"].values())
    for k in plain["grads"]:
        assert (plain["grads"][k] - rccl["grads"][k]).abs().max().item() <= 1e-5 * gmax + 1e-8, k
    _params_agree(plain, rccl, gmax)


def test_one_rank_rccl_bucket_allreduce():
    """The bare collective of the design on RCCL: init_process_group("nccl", device_id=...) with one rank, all-reduce
    of the flat 2 308 946-float gradient bucket, a barrier -- the calls bench.py --gpus N and
    engine.DistEnv make."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "b = torch.randn(2308946, device=dev); ref = b.clone()\n"
        "dist.all_reduce(b); dist.barrier(); torch.cuda.synchronize()\n"
        "assert torch.equal(b, ref)\n"
        "t = torch.ones(1, device=dev); dist.all_reduce(t); assert int(t.item()) == 1\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n")
    import tempfile
    from conftest import run_torchrun
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    try:
        out = run_torchrun(1, [f.name], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=600, capture=True)
    finally:
        os.unlink(f.name)
    assert "RCCL_OK" in out
