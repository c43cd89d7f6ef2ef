#!/usr/bin/env python3
"""tools/exp_icold.hip driver: cycles per taken branch into a cold / warm instruction line at kernel start, alone and with
another kernel (a 16-MB fill) between launches.  Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/libexpic.so tools/exp_icold.hip"""
import ctypes, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libexpic.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.icold_launch.argtypes = [vp, ci, vp]
dev = torch.device("cuda:0")
tbuf = torch.zeros(256 * 16 * 3, dtype=torch.int64, device=dev)
big = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for nw in (1, 16):
    for between in (False, True):
        for _ in range(10):
            if between:
                big.fill_(1)
            lib.icold_launch(tbuf.data_ptr(), nw, st)
        torch.cuda.synchronize()
        t = tbuf.cpu().numpy().reshape(256, 16, 3)[:, :nw].astype(np.float64)
        cold, warm = (t[:, :, 1] - t[:, :, 0]) / 48, (t[:, :, 2] - t[:, :, 1]) / 48
        print("waves/wg %2d, %s: cold %.0f cycles per line (first wave of a workgroup %.0f, last %.0f), warm %.0f"
              % (nw, "64-MB fill between launches" if between else "back to back", cold.mean(), cold[:, 0].mean(), cold[:, -1].mean(), warm.mean()))
