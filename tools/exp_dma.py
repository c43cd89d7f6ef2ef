import ctypes, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpdma.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.run_dma.argtypes = [vp, vp, vp, vp, ci, vp]
g16 = torch.arange(256 * 4, dtype=torch.float32, device="cuda"); g4 = torch.arange(256, dtype=torch.int32, device="cuda") + 1000
for nvalid in (256, 200, 70):
    o16 = torch.zeros(256 * 4, device="cuda"); o4 = torch.zeros(256, dtype=torch.int32, device="cuda")
    lib.run_dma(g16.data_ptr(), g4.data_ptr(), o16.data_ptr(), o4.data_ptr(), nvalid, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ok16 = torch.equal(o16[: nvalid * 4], g16[: nvalid * 4]); ok4 = torch.equal(o4[:nvalid], g4[:nvalid])
    tail = o4[nvalid:].cpu().tolist()[:3]
    print("nvalid %3d: x4 ok %s, dword ok %s, untouched tail (expect -559038737): %s" % (nvalid, ok16, ok4, tail))
