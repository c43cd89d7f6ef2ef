"""Config 5's projections (1152 crops @256x256): per (crop, 64-row region) the touched box of the fused render-and-compare
kernel -- width, rows -- against the cells its z-buffer holds in half of a CU's LDS: how many regions leave rows to the tile
code at the standard row pitch (box width + 8) and at the tight one (round 6), and how many rows."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
S, B = int(os.environ.get("S", 256)), 128
ds = SyntheticMultiviewDataset(mesh, B, S, seed=0, device="cuda")
crit = MutualProjectionLoss(S, mesh).cuda()
with torch.no_grad():
    _, pts = crit.mutual_projection(ds.cam.cuda(), ds.inv_cam.cuda(), ds.joints.cuda() + torch.randn(ds.joints.shape, device="cuda"))
p = pts.squeeze(-1).reshape(B * 9, 41, 3).cpu().numpy()
rad = crit.data_to_model_criterion.radiuses.view(-1).cpu().numpy()
k = S / 300.0
u0 = np.clip(np.ceil((p[:, :, 0] - rad) * k + S / 2 - 1e-3), 0, S - 1); u1 = np.clip(np.floor((p[:, :, 0] + rad) * k + S / 2 + 1e-3), 0, S - 1)
v0 = np.clip(np.ceil((p[:, :, 1] - rad) * k + S / 2 - 1e-3), 0, S - 1); v1 = np.clip(np.floor((p[:, :, 1] + rad) * k + S / 2 + 1e-3), 0, S - 1)
R = int(os.environ.get("ROWS", 64))
zc = int(os.environ.get("ZCELLS", 8504))
tot = over_std = over_tight = rows_std = rows_tight = rows_all = 0
bws, needs = [], []
for r in range(S // R):
    r0, r1 = R * r, R * r + R - 1
    on = (v1 >= r0) & (v0 <= r1) & (u1 >= u0)
    cv0 = np.where(on, np.maximum(v0, r0), 1e9).min(1); cv1 = np.where(on, np.minimum(v1, r1), -1e9).max(1)
    cu0 = np.where(on, u0, 1e9).min(1); cu1 = np.where(on, u1, -1e9).max(1)
    for i in np.nonzero(cv1 >= cv0)[0]:
        a0 = int(cu0[i]) & ~3; b = (int(cu1[i]) | 3) - a0 + 1; n = int(cv1[i] - cv0[i] + 1)
        tot += 1; bws.append(b); needs.append(n); rows_all += n
        p1 = b + 8; p1 = p1 + 8 if (p1 & 31) < 8 else p1
        if n * p1 > zc:
            over_std += 1; rows_std += n - ((zc // p1) & ~7)
            if not any(((b + pad) & 31) != 0 and n * (b + pad) <= zc for pad in (6, 4, 2)):
                over_tight += 1; rows_tight += n - ((zc // p1) & ~7)
print("S=%d, %d-row regions, %d cells: %d regions with a box (of %d); at the standard pitch %d leave %d rows to the tile code (%.1f %% of the box rows); "
      "with the tight pitch %d regions / %d rows (%.1f %%)" % (S, R, zc, tot, (S // R) * p.shape[0], over_std, rows_std, 100.0 * rows_std / rows_all,
                                                        over_tight, rows_tight, 100.0 * rows_tight / rows_all))
print("box width percentiles 10/50/75/90/99/max:", np.percentile(bws, [10, 50, 75, 90, 99]).tolist(), max(bws),
      "| rows 50/90/max:", np.percentile(needs, [50, 90]).tolist(), max(needs))
for need_cells in (8504, 9000, 9500, 10000, 10500, 11000):
    fit = sum(1 for b, n in zip(bws, needs) if n * (b + 2) <= need_cells)
    print("  z-buffer of %5d cells: %.1f %% of the boxes fit at pitch bw + 2" % (need_cells, 100.0 * fit / len(bws)))

# What other z-buffer layouts would need (round 6, docs/EXPERIMENTS.md S5): the rows' own extents (per row the hull of the
# spheres' spans, + 1 pixel of margin, whole 4-pixel pieces, + 2 cells of padding) and two half boxes (the box's rows cut in
# two at the best row, each half with its own columns).
k2 = k * k
tot = over_rows = over_half = 0
cells_box, cells_rows, cells_half = [], [], []
left_rows = left_half = 0
for r in range(S // R):
    r0, r1 = R * r, R * r + R - 1
    vv = np.arange(r0, r1 + 1)[None, None, :]                                   # [1, 1, R]
    yg = (vv - S / 2) / k                                                          # mm (pixel centres; half-pixel conventions are inside the margin)
    dy = yg - p[:, :, 1:2]
    hw = np.sqrt(np.maximum(rad[None, :, None] ** 2 - dy * dy, 0.0)) * k           # half width in pixels, 0 where the row misses the sphere
    hit = (rad[None, :, None] ** 2 - dy * dy) > 0
    cx = p[:, :, 0:1] * k + S / 2
    lo = np.where(hit, np.floor(cx - hw) - 1, 1e9).min(1); hi = np.where(hit, np.ceil(cx + hw) + 1, -1e9).max(1)   # [N, R]
    lo = np.clip(lo, 0, S - 1); hi = np.clip(hi, 0, S - 1)
    for i in range(p.shape[0]):
        rows = np.nonzero(hi[i] >= lo[i])[0]
        if rows.size == 0: continue
        a = lo[i, rows].astype(int) & ~3; b = (hi[i, rows].astype(int) | 3) + 1
        w = b - a                                                                   # per-row widths (multiples of 4)
        n = rows[-1] - rows[0] + 1
        box_w = b.max() - a.min()
        tot += 1
        cells_box.append(n * (box_w + 2)); cells_rows.append(int((w + 2).sum()))
        if cells_rows[-1] > zc: over_rows += 1
        best = min(((np.arange(n) < m) * 0).sum() + (m * ((b[:m].max() - a[:m].min()) + 2) if m else 0) + ((n - m) * ((b[m:].max() - a[m:].min()) + 2) if m < n else 0)
                   for m in range(0, n + 1, 8)) if rows.size == n else n * (box_w + 2)
        cells_half.append(int(best))
        if best > zc: over_half += 1
cb, cr, ch = np.array(cells_box), np.array(cells_rows), np.array(cells_half)
print("cells needed -- one box (pitch bw + 2): %d of %d regions over %d cells; two half boxes: %d; per-row extents: %d" %
      (int((cb > zc).sum()), tot, zc, over_half, over_rows))
print("   per-row / box cell ratio: median %.2f, p90 %.2f; largest per-row need %d cells" % (np.median(cr / cb), np.percentile(cr / cb, 90), cr.max()))
