"""CPU: bench.py's timing contract over 2 gloo ranks, and its static helpers."""
import json
import os
import subprocess
import sys

from conftest import ROOT
from test_ddp_cpu import free_port


def test_timed_steps_is_max_over_ranks(tmp_path):
    out = str(tmp_path / "o.json")
    from conftest import run_torchrun
    run_torchrun(2, [os.path.join(ROOT, "tests", "bench_worker.py"), out], env=dict(os.environ, OMP_NUM_THREADS="1"),
                 timeout=300)
    res = json.load(open(out))
    assert [r["calls"] for r in res] == [23, 23]                       # 3 warm-up + exactly 20 timed
    assert res[0]["elapsed"] == res[1]["elapsed"]                       # every rank reports the max
    assert 0.055 <= res[0]["elapsed"] <= 0.5                            # >= the slow rank's 20 x 3 ms


def test_bench_constants_and_traffic_lookup():
    sys.path.insert(0, ROOT)
    import bench
    assert (bench.S, bench.BATCH, bench.J) == (128, 256, 41)
    # 165 808 B per crop forward + backward with the u8 owner map saved
    fwd = 4 * 128 * 128 + 128 * 128 + 16 * 41
    bwd = 4 * 128 * 128 + 128 * 128 + 16 * 41 + 16 * 41
    assert fwd + bwd == 165808
    # the forward must write every byte; the backward may read fewer (it skips untouched rows)
    traffic, src = bench.pmc_traffic("sphere_zbuf_fwd_kernel")
    assert traffic is None or (traffic > 0.9 * 256 * fwd and src.startswith("profiles/"))
    traffic, src = bench.pmc_traffic("sphere_zbuf_bwd_kernel")
    assert traffic is None or (0.3 * 256 * bwd < traffic < 1.5 * 256 * bwd and src.startswith("profiles/"))
