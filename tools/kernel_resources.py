#!/usr/bin/env python3
"""Register / LDS budgets of the kernels in one csrc/*.hip file, read from the cross-compiled assembly's kernel
descriptors (no GPU).  usage: tools/kernel_resources.py [sphere_raster] [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import build  # noqa: E402

unit = sys.argv[1] if len(sys.argv) > 1 else "sphere_raster"
filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = os.path.join(tempfile.mkdtemp(), unit + ".s")
flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
subprocess.check_call([build.HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I",
                                               os.path.join(build.PKG, "csrc"), "-o", out,
                                               os.path.join(build.PKG, "csrc", unit + ".hip")], stderr=subprocess.DEVNULL)
text = open(out).read()
meta = text[text.index("amdhsa.kernels:"):]
print("asm:", out)
for block in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    if filt not in name:
        continue
    d = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
         for k in ("sgpr_count", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count")}
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-78s sgpr %3d vgpr %3d spills v%d s%d" % (short[-78:], d["sgpr_count"], d["vgpr_count"],
                                                      d["vgpr_spill_count"], d["sgpr_spill_count"]))
