import ctypes, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpsqrt.so"))
lib.exp_sqrt_classify.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(4, dtype=torch.int64, device="cuda")
lo, hi = int(np.float32(0.01).view(np.uint32)), int(np.float32(1e12).view(np.uint32))
lib.exp_sqrt_classify(lo, hi, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("v_sqrt_f32 on [0.01, 1e12): equal %d, one ulp low %d, one ulp high %d, further %d" % tuple(out.tolist()))
