import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd.hourglass import create_hourglass_network
def T(fn, reps=10):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for cl in (False, True):
        net = create_hourglass_network(82, 1).cuda()
        x = torch.rand(123, 1, 64, 64, device="cuda")
        if cl:
            net = net.to(memory_format=torch.channels_last)
        def step():
            net.zero_grad(set_to_none=True)
            out, _ = net(x); out[0].square().mean().backward()
        print("benchmark=%s channels_last=%s: fwd+bwd %.2f ms" % (bench_mode, cl, T(step)))
        with torch.no_grad():
            print("   fwd only %.2f ms" % T(lambda: net(x)))
