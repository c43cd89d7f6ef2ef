"""Replay a case dumped by tools/fuzz.py (FUZZ_DUMP=...) step by step."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spherehand_amd import _lib
if os.environ.get("SHR_LIB"):
    _lib.SO_PATH = os.environ["SHR_LIB"]
from oracle import oracle
from spherehand_amd import ops
oracle.build()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
z = np.load(sys.argv[1]); sp, gd, tgt, H, W = z["sp"], z["gd"], z["tgt"], int(z["H"]), int(z["W"])
od, oa = oracle.sphere_raster_fwd(sp, H, W)
og2 = oracle.sphere_raster_bwd(sp, (2 * (od - tgt)).astype(np.float32))
def fused(tag):
    dep, sse, gsp = ops.sphere_raster_mse(dev(sp), dev(tgt))
    e = np.abs(gsp.cpu().numpy() - og2)
    print("%-40s fused grad err %.3g of %.3g at %s ; depth exact %s" % (tag, e.max(), np.abs(og2).max(), np.unravel_index(e.argmax(), e.shape),
          np.array_equal(dep.cpu().numpy().view(np.uint32), od.view(np.uint32))))
fused("fresh process")
fused("again")
if os.environ.get("SHR_LIB"): sys.exit(0)
ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, int(z["zb"]))
d, a = ops.sphere_raster_fwd(dev(sp), H, W, want_argmin=True); torch.cuda.synchronize()
fused("after forward (zb=%d)" % int(z["zb"]))
for w in (0, 8, 16):
    ops.set_tuning(ops.TUNE_BWD_WAVES, w)
    ops.sphere_raster_bwd(dev(sp), dev(gd), a); torch.cuda.synchronize()
    fused("after backward (%d waves)" % w)
ops.sphere_raster_bwd(dev(sp), dev(gd), None); torch.cuda.synchronize()
fused("after backward (recomputed owners)")
print("spheres of crop with the error:", sp[np.unravel_index(np.abs(ops.sphere_raster_mse(dev(sp), dev(tgt))[2].cpu().numpy() - og2).argmax(), og2.shape)[0]][:, :].round(1)[:6])
n, j = np.unravel_index(np.abs(ops.sphere_raster_mse(dev(sp), dev(tgt))[2].cpu().numpy() - og2).argmax(), og2.shape)[:2]
ops.set_tuning(ops.TUNE_FWD_ZBUF_BYTES, 0); ops.set_tuning(ops.TUNE_BWD_WAVES, 0)
dep, sse, gsp = ops.sphere_raster_mse(dev(sp), dev(tgt))
d, a = ops.sphere_raster_fwd(dev(sp), H, W, want_argmin=True)
g_img = (2 * (od - tgt)).astype(np.float32)
gs = ops.sphere_raster_bwd(dev(sp), dev(g_img), a).cpu().numpy()
print("crop", n, "sphere", j, sp[n, j])
print("oracle ", og2[n, j]); print("fused  ", gsp.cpu().numpy()[n, j]); print("unfused", gs[n, j])
own = np.argwhere(oa[n] == j); print("owned pixels (oracle):", own.tolist(), " gpu owner map agrees:", np.array_equal(a.cpu().numpy(), oa))
diff = np.abs(gsp.cpu().numpy() - og2); bad = np.argwhere(diff > 2e-5 * np.abs(og2).max() + 1e-3)
print("all (crop, sphere, component) beyond the bar:", bad.tolist())
for (nn, jj) in sorted({(int(b[0]), int(b[1])) for b in bad}):
    print(" sphere", jj, sp[nn, jj], "owned px", np.argwhere(oa[nn] == jj).tolist()[:8], "oracle", og2[nn, jj], "fused", gsp.cpu().numpy()[nn, jj])
