"""Phase ablations of the round-2 data_to_model_kernel: variants of data_to_model.hip built with -DEXP_<V> into
tools/libexpd2m_<V>.so (python tools/exp_d2m2.py build  -- here, then run on the GPU box without arguments)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = os.environ.get("VARIANTS", "BASE,NOSEARCH,STAGE1ONLY,BRUTE,NOATOMIC").split(",")
def build():
    from spherehand_amd import build as b
    for v in VARIANTS:
        out = os.path.join(ROOT, "tools", "libexpd2m_%s.so" % v)
        cmd = [b.HIPCC] + b.FLAGS + ["-DEXP_" + v] + (["-DD2M_DEPTH=" + v[5:]] if v.startswith("DEPTH") else []) + (["-DD2M_WPE=" + v[3:]] if v.startswith("WPE") else []) + [ "-I", os.path.join(ROOT, "include"), "-I", os.path.join(b.PKG, "csrc"),
                                    "-o", out, os.path.join(b.PKG, "csrc", "data_to_model.hip")]
        subprocess.check_call(cmd)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    build(); sys.exit(0)
import torch
from bench_d2m import inputs, kernel_us
vp, ci = ctypes.c_void_p, ctypes.c_int
st = lambda: torch.cuda.current_stream().cuda_stream
for B, S in [(128, int(x)) for x in os.environ.get('SIZES', '128,256').split(',')]:
    obs, index, cen, rad = inputs(B, S)
    N = B * 9
    ls = torch.empty(N, device="cuda"); gr = torch.empty(N, 41, 3, device="cuda")
    for v in VARIANTS:
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpd2m_%s.so" % v))
        f = lib.shr_data_to_model_indexed
        f.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp]
        for grad in (True, False):
            t = kernel_us(lambda: f(obs.data_ptr(), index.data_ptr(), cen.data_ptr(), rad.data_ptr(), N, 41, S, S,
                                    ls.data_ptr(), gr.data_ptr() if grad else None, st()))
            print("%d crops @%d  %-10s grad=%d: %.1f us" % (N, S, v, grad, t), flush=True)
