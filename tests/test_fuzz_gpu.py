"""GPU: a bounded, seeded slice of tools/fuzz.py -- randomised differential testing of every kernel family against
the CPU oracle (or, for the kernels that replace torch modules, against those modules): shapes, batch sizes, launch
shapes (shr_set_tuning), non-finite records, negative radii, behind-the-background crops ... drawn at random.
The long run on the round's final kernels is kept in profiles/rNN_fuzz_summary.txt; this one holds every push to the
same checker: a FIXED number of cases per family under a fixed seed (the same cases on every box: what passes here
passes at the next run), ZERO mismatches."""
import importlib.util
import os

import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CASES = {"*": 200, "sphere": 100, "mesh": 100, "d2m": 150, "band": 60, "synth": 60}
SEED = 20260929


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("shr_fuzz", os.path.join(ROOT, "tools", "fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_family_of_the_fuzzer_is_clean(fuzz):
    lines = []
    res = fuzz.run(None, CASES, None, SEED, log=lines.append)
    print("\n".join(lines))
    assert set(res) == {name for name, _ in fuzz.FAMILIES} and len(res) == 14
    assert all(n == CASES.get(name, CASES["*"]) for name, (n, _) in res.items()), res
    assert sum(m for _, m in res.values()) == 0, "\n".join(lines)
