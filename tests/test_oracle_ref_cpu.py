"""CPU: the oracle/_ref recipe (oracle/Makefile, target ref_tri) -- the reference's triangle kernel compiled for gfx950
where it lies.  Here: it builds when /root/reference exists, exports the C entry the GPU test calls, leaves no
translation unit (no reference text) behind, and nothing outside tests/ loads it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def test_ref_tri_recipe_builds_and_exports_the_entry(oracle):
    if not os.path.exists(oracle.REF_CU):
        pytest.skip("no /root/reference here: oracle/_ref is built in the container that has it")
    so = oracle.build_ref()
    assert so and os.path.exists(so) and os.path.exists(os.path.join(REF, "libref_tri_contract.so"))
    for name in ("libref_tri.so", "libref_tri_contract.so"):
        syms = subprocess.check_output(["nm", "-D", os.path.join(REF, name)], text=True)
        assert " T ref_tri_forward" in syms
    left = [f for f in os.listdir(REF) if not f.endswith(".so")]
    assert left == [], left                                    # the piped translation unit is removed after the build


def test_ref_tri_is_test_infrastructure_only():
    """Nothing in the product, bench.py or tools/ mentions the reference build; .gitignore keeps it out of history and
    .gpurunignore does not hold it back from the GPU box."""
    hits = []
    for base, _dirs, files in os.walk(ROOT):
        rel = os.path.relpath(base, ROOT)
        if rel.split(os.sep)[0] in (".git", "tests", "oracle", "gpurun_out", "docs", "profiles", ".pytest_cache", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".sh")) and f != "__graft_entry__.py":
                if "ref_tri" in open(os.path.join(base, f), errors="ignore").read():
                    hits.append(os.path.join(rel, f))
    assert hits == [], hits
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read()
    ign = os.path.join(ROOT, ".gpurunignore")
    assert not os.path.exists(ign) or "oracle/_ref" not in open(ign).read()
