import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def spheres_from(centres, radii):
    """[N,J,>=3] centres + [J] radii -> [N,J,4] fp32 (x,y,z,r)."""
    c = np.asarray(centres, np.float32)[..., :3]
    r = np.broadcast_to(np.asarray(radii, np.float32)[None, :, None], c.shape[:2] + (1,))
    return np.ascontiguousarray(np.concatenate([c, r], -1), np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def run_torchrun(world, script_args, env=None, timeout=600, attempts=3, capture=False):
    """`python -m torch.distributed.run --nproc-per-node world <script_args>` on 127.0.0.1 with a fresh port.  A launch
    that exits non-zero is retried (a fresh port each time): the rendezvous of back-to-back launches on a box that has
    just come up fails now and then (the store's port taken between the probe and the bind), which says nothing about
    the code under test -- an assertion inside a worker fails every attempt and still fails the test.  Returns stdout
    (capture=True) or None."""
    import socket
    import subprocess
    import sys
    last = None
    for attempt in range(attempts):
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(script_args)
        r = subprocess.run(cmd, env=env, timeout=timeout, cwd=ROOT, capture_output=True, text=True)
        if r.returncode == 0:
            return r.stdout if capture else None
        last = r
        msg = "torchrun attempt %d failed (rc %d): %s\n%s" % (attempt + 1, r.returncode, " ".join(script_args)[-200:],
                                                            (r.stderr or "")[-3000:])
        print(msg)
        try:      # kept where the GPU box's scratch output is collected, if there is such a place
            d = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(d):
                with open(os.path.join(d, "torchrun_retries.log"), "a") as f:
                    f.write(msg + "\n----\n")
        except OSError:
            pass
    raise AssertionError("torchrun failed %d times; last stderr:\n%s" % (attempts, (last.stderr or "")[-6000:]))
