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
    # The committed PMC traffic against the bytes the committed line CLAIMS (touched-rows convention: depth / grad image
    # + the owner bytes of the touched rows + records): the forward has to write all of them and little more; the
    # backward stages the touched rows of the gradient only, so it may move less -- never much more.
    import glob
    import re
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
    traffic_f, src = bench.pmc_traffic("sphere_zbuf_fwd_kernel")
    traffic_b, _ = bench.pmc_traffic("sphere_zbuf_bwd_kernel")
    if traffic_f is None or not lines:
        return
    assert src.startswith("profiles/")
    line = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    own = float(re.search(r"([\d.]+) %", line["roofline"]["owner_map"]).group(1)) / 100.0
    claimed_f = 256 * (4 * 128 * 128 + own * 128 * 128 + 16 * 41)
    claimed_b = 256 * (4 * 128 * 128 + own * 128 * 128 + 32 * 41)
    conv = line["roofline"].get("byte_conventions")
    if conv is not None:                                   # (lines since round 4 carry the byte counts themselves)
        assert abs(conv["touched_rows"]["fwd_bytes"] - claimed_f) <= 1e-3 * claimed_f
        assert abs(conv["touched_rows"]["bwd_bytes"] - claimed_b) <= 1e-3 * claimed_b
    assert 0.97 * claimed_f <= traffic_f <= 1.15 * claimed_f, (traffic_f, claimed_f)
    assert 0.5 * claimed_b <= traffic_b <= 1.15 * claimed_b, (traffic_b, claimed_b)
