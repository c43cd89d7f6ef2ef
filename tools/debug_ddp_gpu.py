"""Bisect tests/test_ddp_gpu.py: which loss term makes the 2-rank gradient differ from the 1-rank one."""
import os, subprocess, sys, tempfile, socket
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

def launch(world, terms, tag):
    d = tempfile.mkdtemp()
    env = dict(os.environ, OMP_NUM_THREADS="2", SHR_DDP_TERMS=terms)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), d + "/o.pt", d + "/m"]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return torch.load(d + "/o.pt")

def diff(a, b):
    gmax = max(v.abs().max().item() for v in a["grads"].values())
    worst = max((a["grads"][k] - b["grads"][k]).abs().max().item() for k in a["grads"])
    return gmax, worst

for terms in sys.argv[1:]:
    a = launch(1, terms, "a"); a2 = launch(1, terms, "a2"); b = launch(2, terms, "b")
    print(terms, "1v1: gmax %.4g worst %.4g | 1v2: gmax %.4g worst %.4g" % (diff(a, a2) + diff(a, b)), flush=True)
