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


# stderr of a launch that failed before any worker ran test code: the rendezvous, not the code under test
_RENDEZVOUS_ERRORS = ("EADDRINUSE", "Address already in use", "address already in use", "RendezvousConnectionError",
                      "RendezvousTimeoutError", "DistNetworkError", "DistStoreError", "TCPStore", "store timed out",
                      "failed to connect", "Connection refused", "Connection reset")


def run_torchrun(world, script_args, env=None, timeout=600, attempts=3, capture=False):
    """`python -m torch.distributed.run --nproc-per-node world <script_args>` on 127.0.0.1 with a fresh port.  A launch
    is retried (fresh port) ONLY when its stderr names a rendezvous / port-bind error and no worker assertion: the
    store's port can be taken between the probe and the bind on a box that has just come up.  Anything else -- a worker
    assertion, a mismatch, a rank timeout -- fails the test at once.  Retries are reported as a pytest warning.
    Returns stdout (capture=True) or None."""
    import socket
    import subprocess
    import sys
    import warnings
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
            if attempt:
                warnings.warn("torchrun needed %d attempts (rendezvous errors): %s" % (attempt + 1, " ".join(script_args)[-120:]))
            return r.stdout if capture else None
        last = r
        err = r.stderr or ""
        rendezvous = any(k in err for k in _RENDEZVOUS_ERRORS) and "AssertionError" not in err
        msg = "torchrun attempt %d failed (rc %d, %s): %s\n%s" % (attempt + 1, r.returncode,
                                                                  "rendezvous: retrying" if rendezvous else "not retried",
                                                                  " ".join(script_args)[-200:], err[-3000:])
        print(msg)
        try:      # kept where the GPU box's scratch output is collected, if there is such a place
            d = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(d):
                with open(os.path.join(d, "torchrun_retries.log"), "a") as f:
                    f.write(msg + "\n----\n")
        except OSError:
            pass
        if not rendezvous:
            break
    raise AssertionError("torchrun failed (attempt %d of %d); last stderr:\n%s" % (attempt + 1, attempts, (last.stderr or "")[-6000:]))
