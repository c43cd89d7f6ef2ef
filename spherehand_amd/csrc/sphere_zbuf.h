// sphere_zbuf.h -- FAST sphere rasterizer: sphere-parallel scan conversion into an
// LDS-resident z-buffer (forward) and an LDS-staged owner walk (backward).
//
// Replaces (reference file:line): mesh/render.py:26-53 BallRender.forward + the
// min over spheres (mesh/render.py:89, mesh/multiview_utility.py:76) and the
// autograd backward of those lines.
//
// Why not pixel-parallel: a 128x128 crop has 16384 pixels but its 41 spheres
// cover only ~6000 bounding-box pixels in total (13-15 % foreground), so testing
// candidates per pixel tile spends >10x more lane-operations than walking each
// sphere's own pixel box (measured on MI355X, batch 256: tile kernel 24 us vs
// 4.6 us for a plain 16.8 MB fill).  Here:
//
//   forward   one workgroup (16 waves) per (crop, row region).  The region's
//             z-buffer lives in LDS as order-preserving integer keys (depth bits
//             made monotonic; with OWNER the sphere index is packed in the low
//             word of a 64-bit key, so ds_min_u64 also yields the first-index
//             owner on exact depth ties).  Wave 0 turns the spheres into a work
//             list of row-packed 64-lane chunks (lanes = spheres: pixel box,
//             packing, prefix sum) while waves 1-7 store the rows no sphere
//             touches straight from registers; the 16 waves then take contiguous
//             slices of the list (lanes = pixels): exact reference arithmetic per
//             pixel, one LDS atomic min per hit.  Finally the touched rows are
//             decoded and streamed out with full-line 16-byte write-through
//             stores -- the only HBM traffic besides the 16*J-byte sphere read.
//   backward  one workgroup per crop.  The rows some sphere touches of grad_depth
//             and of the forward's owner map are staged into LDS (the central half
//             speculatively with the records, the rest once they are in); the
//             waves walk the same chunk list in static slices, accumulate the
//             four partials of the pixels a sphere owns in registers, ONE
//             transposed four-component wave reduction per (wave, sphere) run
//             into the wave's private LDS slot, slots combined in wave order:
//             deterministic (the slot's ds_add_f32 is only ever this wave's).
//   fused     forward + (depth - target)^2 + backward in one kernel for the
//             model->data term of MutualProjectionLoss (sphere_zbuf_mse_kernel).
//
// Exactness: the per-pixel arithmetic is the reference's operation sequence
// (common.h, -ffp-contract=off; sqrt_rn() is a correctly rounded square root).
// Integer-key minima are exact.  The fast forward requires, per crop, all sphere
// parameters finite, |x|,|y|,|r| < 1e6 and at least one sphere with z <= 100
// (then initialising the z-buffer to the background is the reference's min: see
// the kernel); any other crop takes the general tile kernel (sphere_tile.h)
// inside the same workgroup.
#pragma once
#include <type_traits>
#include "sphere_tile.h"

namespace shr {

constexpr int kZWaves = 16;   // 1024 threads
// In-kernel timeline (tools/headline_timeline.py builds the library with -DSHR_TIMELINE and reads shr_tl back): every
// wave of the first 256 workgroups of a launch stamps s_memtime at its phase boundaries -- kernel k (0 forward, 1
// backward), slot 0 = entry, 1 .. 5 = the kernel's own marks.  Nothing of it exists in the product build.
#ifdef SHR_TIMELINE
#ifndef SHR_TL_REGION
#define SHR_TL_REGION 1u
#endif
__device__ unsigned long long shr_tl[3][256 * kZWaves * 8];
// (s_memtime counters are not synchronised between CUs: only differences inside a workgroup mean anything.  Entry and
// end are stamped with s_memrealtime as well, the device-wide 100-MHz counter, for the launch's ramp and tail.)
__device__ unsigned long long shr_tl_rt[3][256 * kZWaves * 2];
#define SHR_TL(k, slot)                                                                              \
  do {                                                                                               \
    /* kernel 2 (fused render-and-compare at config 5's size): steady-state workgroups of row region 1 */ \
    const unsigned tl_bx = blockIdx.x - ((k) == 2 ? 512u : 0u);                                      \
    if ((threadIdx.x & 63) == 0 && tl_bx < 256u && blockIdx.y == ((k) == 2 ? SHR_TL_REGION : 0u)) {  \
      shr_tl[k][(tl_bx * kZWaves + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
      if ((slot) == 0 || (slot) == ((k) == 2 ? 7 : 5))                                               \
        shr_tl_rt[k][(tl_bx * kZWaves + (threadIdx.x >> 6)) * 2 + ((slot) != 0)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                \
  } while (0)
#define SHR_TL_ENTRY(k) SHR_TL(k, 0)
#else
#define SHR_TL(k, slot) do {} while (0)
#define SHR_TL_ENTRY(k) do {} while (0)
#endif
#ifndef SHR_ROW_PAD
#define SHR_ROW_PAD 8
#endif
constexpr int kRowPad = SHR_ROW_PAD;    // LDS row padding (elements): chunk rows start in different banks (no lane writes there)
constexpr int kPadRows = 0;   // rows after the region's last (none: a chunk never overhangs its box)
#ifndef SHR_BG_WAVES
#define SHR_BG_WAVES 7
#endif
#ifndef SHR_BG_WAVES_BOX
#define SHR_BG_WAVES_BOX SHR_BG_WAVES
#endif
constexpr int kBgWavesBox = SHR_BG_WAVES_BOX;   // ... in the BOX kernels (two workgroups per CU: launches that are VALU-issue-bound)
constexpr int kBgWaves = SHR_BG_WAVES;   // forward: waves that store the background rows before the first barrier
// (storing them after the barrier instead, with smaller list shares for those waves, moves the barrier from 4.6 k
// to 3.8 k cycles but the stores then cost the scan conversion more than that: measured 8.9 vs 8.6 us)
// LDS header: spheres [64] float4 | work items [64] int4 | ends [64] int | flags [16] | next crop's spheres [64] float4
constexpr int kOffItems = 1024;
constexpr int kOffEnds = 2048;
constexpr int kOffFlags = 2304;
constexpr int kOffNext = 2304 + 64;       // [64] float4: the NEXT crop's records (persistent workgroups)
#ifndef SHR_HDR_PAD
#define SHR_HDR_PAD 0
#endif
constexpr int kHdrBytes = kOffNext + 1024 + SHR_HDR_PAD;   // (SHR_HDR_PAD: experiment -- where the z-buffers start relative to the LDS banks)
constexpr int kMaxFastWidth = 8192;  // 16-bit fields of the work items

__device__ __forceinline__ uint32_t depth_key(float d) {
  const uint32_t b = __float_as_uint(d);
  return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_depth(uint32_t k) {
  // (top bit set: the float was non-negative, only the sign flips back; clear: every bit does -- an arithmetic shift and one
  // three-input bit operation.  The select form, k ^ (k & 0x80000000 ? 0x80000000 : ~0), compiled to a 64-bit compare +
  // select + xor per cell on the u64 z-buffers; same bits, and no measurable difference: round 6, tools/ab_variant.py)
  return __uint_as_float(k ^ ((uint32_t)(~((int32_t)k >> 31)) | 0x80000000u));
}
// 64-bit cells (depth key << 32 | owner): ds_min_u64 keeps the nearest hit and, among equal depths, the lowest sphere
// index; the background cell is (key(100) << 32 | 0xFFFFFFFF).  One case needs a second look when a cell is decoded:
// a hit at EXACTLY 100.0 beats the background cell, but torch.min keeps the first index at the minimum, and the
// maps of the spheres that miss the pixel hold 100 there as well -- sphere j owns the pixel (and receives its
// gradient) only if every lower-index sphere hits it BEHIND the background.  Such a cell is recognisable (depth key
// of 100 with a sphere index), it practically never occurs (tools/fuzz.py found one in ~40 k crops), and
// tie_owner() then evaluates the lower spheres directly.
__device__ __forceinline__ unsigned long long background_cell() {
  return ((unsigned long long)depth_key(kBackground) << 32) | 0xFFFFFFFFull;
}
__device__ __forceinline__ bool is_background_tie(unsigned long long k) {
  return (uint32_t)(k >> 32) == depth_key(kBackground) && (uint32_t)k < (uint32_t)SHR_MAX_SPHERES;
}
// the owner of a pixel whose cell holds (100.0, j): j if spheres 0 .. j-1 all hit it deeper than 100, else nobody
__device__ __forceinline__ uint32_t tie_owner(const float4 *s_sph, uint32_t j, float xg, float yg) {
  for (uint32_t i = 0; i < j; i++) {
    const float4 s = s_sph[i];
    const float dx = xg - s.x, dy = yg - s.y;
    const float q = (s.w * s.w - dx * dx) - dy * dy;
    if (q <= kHitMin || !(s.z - sqrtf(q) > kBackground)) return SHR_ARGMIN_NONE;
  }
  return j;
}

// Correctly rounded sqrt for a normal, positive, finite fp32 argument (here
// q > 0.01): the hardware estimate (v_sqrt_f32, <= 1 ulp) corrected by the exact
// residuals of its two neighbours -- the same correction hipcc's sqrtf() applies,
// without its denormal-scaling prologue.  Checked against sqrtf() on every fp32
// value in [0.01, 1e12] (tests/test_sphere_raster_gpu.py::test_sqrt_rn_exhaustive).

__device__ __forceinline__ bool sphere_is_tame(const float4 s) {
  return fabsf(s.x) < 1e6f && fabsf(s.y) < 1e6f && fabsf(s.w) < 1e6f;  // false for NaN/Inf
}

// One work item = one sphere's pixel box clipped to the region.  A hit needs
// |fl(xg - x)| <= |r| (sphere_tile.h), i.e. the pixel centre inside [x - |r|, x + |r|] up
// to one rounding.  With xg(u) = (u - half)*300/size  <=>  u = xg*size/300 + half the box is
//   u in [ceil(ulo - eps), floor(uhi + eps)],
// eps = 1e-3 px + 2e-6 relative: >> the fp32 error of the inverse map (<= 1e-5 px
// near the image, 6e-8 relative far away), << a pixel, so the box is the tight
// conservative one.  Non-tame spheres (NaN/Inf/huge) take the whole region.
//
// The box is visited in CHUNKS of 64 lanes packed row-major over its own width:
// pw = min(width, 64) lanes per row, ph = 64 / pw rows per chunk (a 12-px-wide finger
// sphere: 5 rows x 12 = 60 busy lanes; a 19-px palm sphere: 3 x 19 = 57), ncx =
// ceil(width / 64) column segments for boxes wider than a wave.  Fixed 16x4 patches
// left 43 % of the box lanes idle on hand crops (230 patch visits per crop against
// 172 chunks) and paid a row loop on top.
struct Item { int u0, v0, u1, v1, pw, ph, ncx, nchunks; };

__device__ __forceinline__ void axis_box(float c, float ar, float k, float half, float hi_clamp, int lo_lim,
                                         int hi_lim, int &i0, int &i1) {
  const float lo = (c - ar) * k + half, hi = (c + ar) * k + half;
  const float eps = 1e-3f + 2e-6f * (fabsf(lo) + fabsf(hi));
  i0 = max((int)ceilf(fminf(fmaxf(lo - eps, -2.f), hi_clamp)), lo_lim);
  i1 = min((int)floorf(fminf(fmaxf(hi + eps, -2.f), hi_clamp)), hi_lim);
}

// (kx, ky = size / 300 per axis: wave-uniform, computed by the caller before the records arrive)
// SEG2 (kernels for images from 192 pixels on, where a hand's spheres get there): a box 33 .. 64 pixels wide packed
// one row per chunk leaves up to half of a chunk's lanes without a pixel (35 columns: 29 idle lanes; at 256 x 256
// fourteen of the hand's 41 spheres are that wide: 822 chunks per crop where the boxes hold 569 x 64 pixels).  Such a
// box is walked as TWO column segments of ceil(w / 2) columns, each packed like a narrow box (35 columns -> 2 x 18,
// three rows per chunk: 24 chunks instead of 35); chunk c = (row group c / 2, segment c % 2).  The tag rides in the
// walkers' kSphereCost parameter (kSeg2Tag): kernels without it are round 4's code, instruction for instruction --
// the two-segment code in every kernel cost the 128 x 128 launches 2.6-4 % (EXPERIMENTS R2e).
constexpr int kSeg2Tag = 0x100;
template <bool SEG2>
__device__ __forceinline__ Item sphere_item(const float4 s, const Axis &ax, const Axis &ay, float kx, float ky, int W,
                                            int r0, int r1) {
  int u0 = 0, u1 = W - 1, v0 = r0, v1 = r1 - 1;
  if (sphere_is_tame(s)) {
    const float ar = fabsf(s.w);
    axis_box(s.x, ar, kx, ax.half, (float)W + 2.f, 0, W - 1, u0, u1);
    axis_box(s.y, ar, ky, ay.half, (float)r1 + 2.f, r0, r1 - 1, v0, v1);
  }
  Item it;
  it.u0 = u0; it.v0 = v0; it.u1 = u1; it.v1 = v1;
  const int w = u1 - u0 + 1, h = v1 - v0 + 1;
  if (w <= 0 || h <= 0) {
    it.pw = 1; it.ph = 1; it.ncx = 0; it.nchunks = 0;
  } else {
    // Small integer quotients through v_rcp_f32 (1 ulp): floor(64 / pw) -- the exact quotient is an
    // integer or at least 1/64 below the next one, so + 0.01 then truncation is exact; ceil(h / ph) =
    // floor((h + ph - 1 + 0.5) / ph) -- at least 1/128 from an integer, the error stays below that
    // for h < 16384 (taller boxes take the IEEE division).
    const bool seg2 = SEG2 && w > 32 && w <= kWave;
    it.pw = seg2 ? (w + 1) >> 1 : min(w, kWave);
    it.ph = (int)(64.0f * __builtin_amdgcn_rcpf((float)it.pw) + 0.01f);
    it.ncx = w > kWave ? (w + kWave - 1) >> 6 : (seg2 ? 2 : 1);
    const float hh = (float)(h + it.ph - 1) + 0.5f;
    const int ngr = h < 16384 ? (int)(hh * __builtin_amdgcn_rcpf((float)it.ph)) : (int)(hh / (float)it.ph);
    it.nchunks = ngr * it.ncx;
  }
  return it;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rfl(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// The work list is the sequence of chunks, sphere after sphere, on a weight axis where a
// chunk is 2^kChunkShift long and starting a sphere (broadcasts, lane layout, column terms)
// kSphereCost: equal slices of the axis are equal work.
constexpr int kChunkShift = 3;
constexpr int kChunkCost = 1 << kChunkShift;
constexpr int kSphereCostFwd = 20;
constexpr int kSphereCostBwd = 36;   // ... plus four wave reductions when a run on a sphere ends

// Wave 0: build the list in LDS.  s_items[j] = (u0 | v0<<16, v1 | ceil(2^15/pw)<<16, pw | ncx<<8 |
// u1<<16, weight prefix before sphere j); s_ends[j] = prefix after it; a sphere weighs
// kSphereCost + nchunks * kChunkCost (nothing when it touches no pixel).  Returns the total
// weight (valid in every lane of wave 0).  Field widths: the launcher keeps W <= 8192 and
// H <= 32768.
template <bool POW2>
__device__ __forceinline__ float axis_coord_t(const Axis &a, int u);

// (s_run, RUN TABLE kernels only: lane j also leaves what a run on sphere j needs as VECTOR values -- y, z, the row
// coordinate of the box's first row and the row limit -- for one uniform-address ds_read_b128; see build_run_table)
template <int kSphereCost>
__device__ __forceinline__ int build_work_list(const float4 s, bool valid, const Axis &ax, const Axis &ay,
                                               float kx, float ky, int W, int r0, int r1, int4 *s_items,
                                               int *s_ends, int lane, bool *too_big, float4 *s_run = nullptr) {
  const Item it = sphere_item<(kSphereCost & kSeg2Tag) != 0>(s, ax, ay, kx, ky, W, r0, r1);
  const int nchunks = valid ? it.nchunks : 0;
  const int cost = nchunks > 0 ? (kSphereCost & 0xff) + nchunks * kChunkCost : 0;
  // inclusive scan over the 64 lanes: 4 DPP steps inside each row of 16, then the
  // three row totals are added with SGPR broadcasts
  int incl = cost;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8
  const int r0s = rl(incl, 15), r1s = rl(incl, 31), r2s = rl(incl, 47);
  const int row = lane >> 4;
  incl += (row >= 1 ? r0s : 0) + (row >= 2 ? r1s : 0) + (row >= 3 ? r2s : 0);
  // ceil(2^15 / pw) = floor((2^15 + pw - 1 + 1/2) / pw): the quotient is at least 1/128 from an integer and
  // v_rcp_f32's ulp leaves 0.004 of error at this magnitude; the walk derives ph = 64 / pw from it as well
  const int inv15 = (int)((32768.0f + (float)it.pw - 0.5f) * __builtin_amdgcn_rcpf((float)it.pw));
  s_items[lane] = make_int4(it.u0 | (it.v0 << 16), (int)((unsigned)(it.v1 & 0xffff) | ((unsigned)inv15 << 16)),
                            it.pw | (it.ncx << 8) | ((it.u1 & 0xffff) << 16), incl - cost);
  s_ends[lane] = incl;
  if (s_run)
    s_run[lane] = make_float4(s.y, s.z, axis_coord_t<true>(ay, it.v0), axis_coord_t<true>(ay, it.v1) + 0.5f * ay.mul);
  *too_big = __ballot(valid && nchunks > 0 && (it.ncx > 255 || it.u1 > 65535 || it.v1 > 65535 ||
                                               nchunks > (1 << 24))) != 0ull;
  return rl(incl, 63);
}

// Each wave keeps sphere j's record and work item in lane j's registers: a run on a
// sphere starts with a few v_readlane (SGPR results), no LDS round trip.
struct WaveList {
  float4 sph;   // lane j: sphere j
  int4 item;    // lane j: see build_work_list
  int end;      // lane j: weight prefix after sphere j
};

__device__ __forceinline__ WaveList load_wave_list(const float4 *s_sph, const int4 *s_items, const int *s_ends,
                                                   int lane) {
  WaveList w;
  w.sph = s_sph[lane];
  w.item = s_items[lane];
  w.end = s_ends[lane];
  return w;
}

// RUN TABLE (whole-crop workgroups on power-of-two images; round 3).  What the start of a run on a sphere computes per
// LANE -- the lane layout lane -> (lx, ly), the column term r*r - dx*dx, the cell, the row -- depends on (sphere,
// lane) only, and its ~42 VALU instructions hang on a chain of v_readlane -> scalar arithmetic -> branch: measured
// (clock64 stamps inside walk_slice) 750 cycles from the top of a run to its first chunk, 47 runs per
// hand crop, a third of a wave's scan time -- latency, not issue slots.  Four waves therefore compute it ONCE per
// sphere into LDS in front of the first barrier (which ones: see the kernel),
//     tab[j * 64 + lane] = (bits(ca or -1 outside the packing), ly << 24 | ((v0 + ly) * LW + u)),
// with the same operations as walk_slice (bit-identical values).  A run then starts from ONE v_readlane (its
// chunk range, packed per (wave, sphere) by a vector pass at the walk's entry), the lane's table entry and the
// sphere's run record (build_work_list), both requested one run ahead.  Spheres wider than a wave (ncx > 1) keep
// the arithmetic path.
struct RunTab {
  const uint2 *tab;    // [J][64]
  const float4 *run;   // [64]: (y, z, yg(v0), yg(v1) + half a pixel)
};

// The table's entries of the spheres in `mine` (wave-uniform mask), lanes = the chunk's lanes; the sphere's box origin
// and packing come from a lanes = spheres prologue (the same axis_box calls as sphere_item) through v_readlane.
template <bool POW2>
__device__ __forceinline__ void build_run_table(const float4 s, bool valid, const Axis &ax, const Axis &ay, float kx,
                                                float ky, int W, int r0, int r1, int LW, uint2 *tab, int lane,
                                                unsigned long long mine) {
  int u0 = 0, u1 = W - 1, v0 = r0, v1 = r1 - 1;
  if (sphere_is_tame(s)) {
    const float ar = fabsf(s.w);
    axis_box(s.x, ar, kx, ax.half, (float)W + 2.f, 0, W - 1, u0, u1);
    axis_box(s.y, ar, ky, ay.half, (float)r1 + 2.f, r0, r1 - 1, v0, v1);
  }
  const int pw = min(max(u1 - u0 + 1, 1), kWave);
  const int inv15 = (int)((32768.0f + (float)pw - 0.5f) * __builtin_amdgcn_rcpf((float)pw));   // as build_work_list
  const int cell00 = __mul24(v0, LW) + u0;
  const int pwinv = pw | (inv15 << 8);             // (pw <= 64, inv15 <= 32768)
  const float rr = s.w * s.w;
  auto entry = [&](int j) {
    const int pk = rl(pwinv, j), u0j = rl(u0, j), c00 = rl(cell00, j);
    const float sx = readlane_f(s.x, j), rrj = readlane_f(rr, j);
    const int pwj = pk & 0xff, invj = (int)((unsigned)pk >> 8), ph = (invj << 6) >> 15;
    const int ly = __mul24(lane, invj) >> 15;
    const int lx = lane - __mul24(ly, pwj);
    const float dx = axis_coord_t<POW2>(ax, u0j + lx) - sx;
    const float ca = rrj - dx * dx;
    return make_uint2(__float_as_uint(ly < ph ? ca : -1.f), (unsigned)(c00 + __mul24(ly, LW) + lx) | ((unsigned)ly << 24));
  };
  // two spheres per iteration: a single wave issues a DEPENDENT vector instruction every ~8 cycles, and these waves
  // have their SIMD almost to themselves while the others wait for memory
  while (mine) {
    const int ja = __builtin_ctzll(mine);
    mine &= mine - 1;
    if (!mine) { tab[ja * kWave + lane] = entry(ja); break; }
    const int jb = __builtin_ctzll(mine);
    mine &= mine - 1;
    const uint2 ea = entry(ja), eb = entry(jb);
    tab[ja * kWave + lane] = ea;
    tab[jb * kWave + lane] = eb;
  }
}

template <bool POW2, int kSphereCost, bool ROWFREE, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_slice(const WaveList &w, int J, int lo, int hi, int lane, const Axis &ax,
                                           const Axis &ay, int r0, int r1, int LW, Body &&body,
                                           EndSphere &&end_sphere, unsigned long long only = ~0ull);

// The table walk of slice [lo, hi): same chunks, same body calls as walk_slice (the order of the spheres differs only
// where wide ones are deferred; the forward's minima do not depend on it).
template <int kSphereCost, bool ROWFREE, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_slice_table(const WaveList &w, int J, int lo, int hi, int lane, const Axis &ax,
                                                 const Axis &ay, int r0, int r1, int LW, const RunTab rt, Body &&body,
                                                 EndSphere &&end_sphere) {
  // lanes = spheres: this slice's chunk range of every sphere, packed for one v_readlane per run
  const int wstart = w.item.w, wend = w.end, base = wstart + (kSphereCost & 0xff);
  const int cb = lo <= base ? 0 : (lo - base + kChunkCost - 1) >> kChunkShift;
  const int ce = min((wend - base) >> kChunkShift, (hi - base + kChunkCost - 1) >> kChunkShift);
  const int v1 = w.item.y & 0xffff, inv15 = (int)((unsigned)w.item.y >> 16), ph = (inv15 << 6) >> 15;
  const bool active = lane < J && wend > wstart && cb < ce;
  const bool narrow = ((w.item.z >> 8) & 0xff) == 1;
  // (cb, ce < 4096: the launcher keeps the regions of a table kernel below that many rows)
  const int pk = cb | (ce << 12) | (ph << 24) | ((ROWFREE && v1 < r1 - 1) ? (int)0x80000000u : 0);
  unsigned long long m = __ballot(active && narrow);
  const unsigned long long wide = __ballot(active && !narrow);
  if (m) {
    const uint2 *tl = rt.tab + lane;
    int j = __builtin_ctzll(m);
    uint2 t = tl[j * kWave];
    float4 rn = rt.run[j];                       // uniform address: an LDS broadcast
    while (true) {
      m &= m - 1;
      const int jn = m ? __builtin_ctzll(m) : j;
      const int pkj = rl(pk, j);
      const uint2 tn = tl[jn * kWave];           // the next run's entry and record, requested a run ahead
      const float4 rnn = rt.run[jn];
      int c = pkj & 0xfff;
      const int c_end = (pkj >> 12) & 0xfff, phj = (pkj >> 24) & 0x7f;
      const int cph = c * phj, dcell = phj * LW;
      int cell = (int)(t.y & 0xffffffu) + (cph - r0) * LW;
      const float cav = __uint_as_float(t.x);
      // row coordinates are multiples of 300 / S below 2^24: the products and sums are exact, the fused form gives
      // axis_coord_t(ay, v0 + c * ph + ly) to the bit
      float yg = __builtin_fmaf((float)((int)(t.y >> 24) + cph), ay.mul, rn.z);
      const float dyg = (float)phj * ay.mul;
      const float4 s = make_float4(0.f, rn.x, rn.y, 0.f);   // (the bodies of the table kernels use y and z)
      if (ROWFREE && pkj < 0) {
        for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell)
          body(j, s, cell, cell + dcell, 0.f, cav, yg, yg + dyg, true, true, c + 1 < c_end, std::false_type());
      } else {
        const float ylim = rn.w;                 // (a lane outside the packing never hits: its column term is -1)
        for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell) {
          const float ygb = yg + dyg;
          body(j, s, cell, cell + dcell, 0.f, cav, yg, ygb, yg <= ylim, ygb <= ylim, c + 1 < c_end, std::true_type());
        }
      }
      end_sphere(j);
      if (!m) break;
      j = jn; t = tn; rn = rnn;
    }
  }
  if (wide) walk_slice<true, kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere, wide);
}

// The same entry without a table (any power-of-two kernel): the slice's chunk ranges packed by one vector pass, the
// spheres with work iterated over a mask, and a run's broadcasts -- ONE for its chunk range, the item's three words,
// the record -- issued back to back with no branch between them; the per-lane set-up is walk_slice's, operation for
// operation.  walk_slice reads a sphere's weight range, branches, reads its end, branches, reads the item, branches:
// every step waits for a v_readlane to reach the scalar unit and for a branch to refill (~750 cycles from the top
// of a run to its first chunk, 47 runs per hand crop: DESIGN 4.1).
template <int kSphereCost, bool ROWFREE, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_slice_packed(const WaveList &w, int J, int lo, int hi, int lane, const Axis &ax,
                                                  const Axis &ay, int r0, int r1, int LW, Body &&body,
                                                  EndSphere &&end_sphere) {
  const int wstart = w.item.w, wend = w.end, base = wstart + (kSphereCost & 0xff);
  const int cb = lo <= base ? 0 : (lo - base + kChunkCost - 1) >> kChunkShift;
  const int ce = min((wend - base) >> kChunkShift, (hi - base + kChunkCost - 1) >> kChunkShift);
  const int v1l = w.item.y & 0xffff, inv15l = (int)((unsigned)w.item.y >> 16), phl = (inv15l << 6) >> 15;
  const bool active = lane < J && wend > wstart && cb < ce;
  const bool narrow = ((w.item.z >> 8) & 0xff) == 1;
  // (cb, ce < 4096: the caller sends taller regions through walk_slice; pw <= 64)
  const int pk = cb | (ce << 12) | ((w.item.z & 0x7f) << 24) | ((ROWFREE && v1l < r1 - 1) ? (int)0x80000000u : 0);
  // what a run needs of its sphere beyond the record, formed ONCE here with lanes = spheres -- a run broadcasts the
  // result instead of repeating the arithmetic on a uniform value: r * r, the chunk's height in millimetres, the
  // row limit of a box that reaches the region's last row
  const float rrl = w.sph.w * w.sph.w;
  const float dygl = (float)phl * ay.mul;
  const float yliml = axis_coord_t<true>(ay, min(v1l, r1 - 1)) + 0.5f * ay.mul;
  unsigned long long m = __ballot(active && narrow);
  const unsigned long long wide = __ballot(active && !narrow);
  while (m) {
    const int j = __builtin_ctzll(m);
    m &= m - 1;
    const int pkj = rl(pk, j), geom = rl(w.item.x, j), rows = rl(w.item.y, j);
    const float4 s = make_float4(readlane_f(w.sph.x, j), readlane_f(w.sph.y, j), readlane_f(w.sph.z, j), 0.f);
    const float rr = readlane_f(rrl, j);
    int c = pkj & 0xfff;
    const int c_end = (pkj >> 12) & 0xfff, pw = (pkj >> 24) & 0x7f;
    const int u0 = geom & 0xffff, v0 = (int)((unsigned)geom >> 16);
    const int inv15 = (int)((unsigned)rows >> 16), ph = (inv15 << 6) >> 15;
    const int ly = __mul24(lane, inv15) >> 15;
    const int lx = lane - __mul24(ly, pw);
    const bool packed = ly < ph;
    const int u = u0 + lx;
    const float dx = axis_coord_t<true>(ax, u) - s.x;
    const float ca = rr - dx * dx;
    const int v = v0 + c * ph + ly;
    int cell = __mul24(v - r0, LW) + u;
    const int dcell = ph * LW;
    const float cav = packed ? ca : -1.f;
    float yg = axis_coord_t<true>(ay, v);
    const float dyg = readlane_f(dygl, j);
    if (ROWFREE && pkj < 0) {
      for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell)
        body(j, s, cell, cell + dcell, dx, cav, yg, yg + dyg, packed, packed, c + 1 < c_end, std::false_type());
    } else {
      const float ylim = packed ? readlane_f(yliml, j) : -3.0e38f;
      for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell) {
        const float ygb = yg + dyg;
        body(j, s, cell, cell + dcell, dx, cav, yg, ygb, yg <= ylim, ygb <= ylim, c + 1 < c_end, std::true_type());
      }
    }
    end_sphere(j);
  }
  if (wide) walk_slice<true, kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere, wide);
}

// walk_slice_packed for the kernels whose lists hold two-segment boxes (sphere_item<SEG2 = true>): a box of 33 .. 64
// columns is two runs, one per column segment.
template <int kSphereCost, bool ROWFREE, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_slice_packed_seg2(const WaveList &w, int J, int lo, int hi, int lane, const Axis &ax,
                                                  const Axis &ay, int r0, int r1, int LW, Body &&body,
                                                  EndSphere &&end_sphere) {
  const int wstart = w.item.w, wend = w.end, base = wstart + (kSphereCost & 0xff);
  const int cb = lo <= base ? 0 : (lo - base + kChunkCost - 1) >> kChunkShift;
  const int ce = min((wend - base) >> kChunkShift, (hi - base + kChunkCost - 1) >> kChunkShift);
  const int v1l = w.item.y & 0xffff, inv15l = (int)((unsigned)w.item.y >> 16), phl = (inv15l << 6) >> 15;
  const bool active = lane < J && wend > wstart && cb < ce;
  const int ncxl = (w.item.z >> 8) & 0xff;
  const bool seg2l = ncxl == 2 && (w.item.z & 0x7f) <= 32;       // two column segments (sphere_item), walked here as two runs
  const bool narrow = ncxl == 1 || seg2l;
  // (cb, ce < 4096: the caller sends taller regions through walk_slice; pw <= 64; a two-segment box: pw + 64 <= 96)
  const int pk = cb | (ce << 12) | (((w.item.z & 0x7f) + (seg2l ? 64 : 0)) << 24) | ((ROWFREE && v1l < r1 - 1) ? (int)0x80000000u : 0);
  // what a run needs of its sphere beyond the record, formed ONCE here with lanes = spheres -- a run broadcasts the
  // result instead of repeating the arithmetic on a uniform value: r * r, the chunk's height in millimetres, the
  // row limit of a box that reaches the region's last row
  const float rrl = w.sph.w * w.sph.w;
  const float dygl = (float)phl * ay.mul;
  const float yliml = axis_coord_t<true>(ay, min(v1l, r1 - 1)) + 0.5f * ay.mul;
  unsigned long long m = __ballot(active && narrow);
  const unsigned long long wide = __ballot(active && !narrow);
  while (m) {
    const int j = __builtin_ctzll(m);
    m &= m - 1;
    const int pkj = rl(pk, j), geom = rl(w.item.x, j), rows = rl(w.item.y, j);
    const float4 s = make_float4(readlane_f(w.sph.x, j), readlane_f(w.sph.y, j), readlane_f(w.sph.z, j), 0.f);
    const float rr = readlane_f(rrl, j);
    const int cb_j = pkj & 0xfff, ce_j = (pkj >> 12) & 0xfff, pwf = (pkj >> 24) & 0x7f;
    const bool seg2 = pwf > 64;
    const int pw = seg2 ? pwf - 64 : pwf;
    const int u0 = geom & 0xffff, v0 = (int)((unsigned)geom >> 16);
    const int inv15 = (int)((unsigned)rows >> 16), ph = (inv15 << 6) >> 15;
    const int ly = __mul24(lane, inv15) >> 15;
    const int lx = lane - __mul24(ly, pw);
    const bool packed = ly < ph;
    const float dyg = readlane_f(dygl, j);
    const int dcell = ph * LW;
    const float ylim = packed ? readlane_f(yliml, j) : -3.0e38f;
    // one run: the row groups [c, c_end) of the columns u0s .. (the whole box, or one of its two segments; a segment's
    // lanes right of the box -- the second one of an odd width -- never hit)
    auto run = [&](int u0s, int u1s, int c, int c_end) {
      const int u = u0s + lx;
      const float dx = axis_coord_t<true>(ax, u) - s.x;
      const float ca = rr - dx * dx;
      const int v = v0 + c * ph + ly;
      int cell = __mul24(v - r0, LW) + u;
      const float cav = (packed && u <= u1s) ? ca : -1.f;
      float yg = axis_coord_t<true>(ay, v);
      if (ROWFREE && pkj < 0) {
        for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell)
          body(j, s, cell, cell + dcell, dx, cav, yg, yg + dyg, packed, packed, c + 1 < c_end, std::false_type());
      } else {
        for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell) {
          const float ygb = yg + dyg;
          body(j, s, cell, cell + dcell, dx, cav, yg, ygb, yg <= ylim, ygb <= ylim, c + 1 < c_end, std::true_type());
        }
      }
    };
    if (!seg2) {
      run(u0, 0x7fffffff, cb_j, ce_j);
    } else {
      // chunks [cb, ce) of the list interleave the segments (c = 2 g + segment): each segment's row groups are a
      // contiguous range, walked as a run of its own (pairs along the rows, as in a narrow box)
      const int u1 = (int)((unsigned)rl(w.item.z, j) >> 16);
      const int a0 = (cb_j + 1) >> 1, a1 = (ce_j + 1) >> 1;          // segment 0: even c
      const int b0 = cb_j >> 1, b1 = ce_j >> 1;                      // segment 1: odd c
      if (a0 < a1) run(u0, u1, a0, a1);
      if (b0 < b1) run(u0 + pw, u1, b0, b1);
    }
    end_sphere(j);
  }
  if (wide) walk_slice<true, kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere, wide);
}

// Walk the chunks whose weight position lies in [lo, hi) (wave-uniform), sphere by
// sphere.  Per sphere the lane layout (lx, ly) and the column term c = r*r - dx*dx are
// set up once; a chunk then costs its row coordinate, dy*dy and one subtraction before
// `body(j, s, cell_a, cell_b, dx, dy_a, dy_b, q_a, q_b, has_b)` -- two consecutive chunks
// per call (independent dependency chains).  The body gets the pixel's cell = (v - r0) * LW + u
// in the region's LDS arrays, dx = xg - x, ca = r*r - dx*dx (-1 for a lane outside the packing),
// the row coordinate yg, whether the lane HAS a pixel in the chunk (not beyond the last row,
// the packing remainder, or the box's right edge in a column segment) and a compile-time tag:
// false_type when that flag only repeats "inside the packing" (ca already says so).  The body
// forms q = ca - (yg - y)^2 -- the reference's association -- when it needs it (the backward
// only for pixels the sphere owns).
// `end_sphere(j)` closes a run on sphere j.
// (`only`: the spheres to visit -- all of them, or the wide ones a table walk has left)
template <bool POW2, int kSphereCost, bool ROWFREE, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_slice(const WaveList &w, int J, int lo, int hi, int lane, const Axis &ax,
                                           const Axis &ay, int r0, int r1, int LW, Body &&body,
                                           EndSphere &&end_sphere, unsigned long long only) {
  int j = __popcll(__ballot(lane < J && w.end <= lo));   // prefixes are non-decreasing
  while (j < J) {
    const int wstart = rl(w.item.w, j);
    if (wstart >= hi) break;
    if (!((only >> j) & 1ull)) { ++j; continue; }
    const int wend = rl(w.end, j);
    const int base = wstart + (kSphereCost & 0xff);
    // chunk c sits at base + c * kChunkCost and belongs to the slice that holds that position
    int c = lo <= base ? 0 : (lo - base + kChunkCost - 1) >> kChunkShift;
    const int c_end = min((wend - base) >> kChunkShift, (hi - base + kChunkCost - 1) >> kChunkShift);
    if (wend > wstart && c < c_end) {
      const int geom = rl(w.item.x, j), rows = rl(w.item.y, j), cols = rl(w.item.z, j);
      const float4 s = make_float4(readlane_f(w.sph.x, j), readlane_f(w.sph.y, j), readlane_f(w.sph.z, j),
                                   readlane_f(w.sph.w, j));
      const int u0 = geom & 0xffff, v0 = (int)((unsigned)geom >> 16);
      const int v1 = rows & 0xffff, inv15 = (int)((unsigned)rows >> 16), ph = (inv15 << 6) >> 15;   // ph = 64 / pw
      const int v1c = min(v1, r1 - 1);   // (r1 may cut the list's boxes: the forward's z-buffer can end above the region's last row)
      const int pw = cols & 0xff, ncx = (cols >> 8) & 0xff, u1 = (int)((unsigned)cols >> 16);
      const float rr = s.w * s.w;
      // lane -> (lx, ly) inside a chunk: ly = lane / pw = (lane * ceil(2^15 / pw)) >> 15, exact for lane < 64
      const int ly = __mul24(lane, inv15) >> 15;
      const int lx = lane - __mul24(ly, pw);   // (24-bit multiplies: v_mul_lo_u32 is quarter rate)
      const bool packed = ly < ph;
      if (ncx == 1) {
        const int u = u0 + lx;
        const float dx = axis_coord_t<POW2>(ax, u) - s.x;
        const float ca = rr - dx * dx;
        int v = v0 + c * ph + ly;
        int cell = __mul24(v - r0, LW) + u;
        const int dcell = ph * LW;
        const float cav = packed ? ca : -1.f;   // a lane outside the packing never hits
        if (POW2) {
          // power-of-two image: the grid coordinates are multiples of 300 / S below 2^24 --
          // exact in fp32, so the row coordinate is carried by additions
          float yg = axis_coord_t<true>(ay, v);
          const float dyg = (float)ph * ay.mul;
          if (ROWFREE && v1 < r1 - 1) {
            // The box ends inside the region: the rows a last chunk reaches below it are real
            // pixels, outside the (conservative) box -- the hit test fails there and the sphere
            // owns none of them -- so no row test at all.  (ROWFREE: every cell of the region holds
            // valid data.  The backward stages only the touched rows: it keeps the test.)
            for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell)
              body(j, s, cell, cell + dcell, dx, cav, yg, yg + dyg, packed, packed, c + 1 < c_end, std::false_type());
          } else {
            // clipped by the region's last row: "row <= v1" as ONE comparison of coordinates
            // (half a pixel of margin; the limit of a lane outside the packing is -huge)
            const float ylim = packed ? axis_coord_t<true>(ay, v1c) + 0.5f * ay.mul : -3.0e38f;
            for (; c < c_end; c += 2, yg += 2.f * dyg, cell += 2 * dcell) {
              const float ygb = yg + dyg;
              body(j, s, cell, cell + dcell, dx, cav, yg, ygb, yg <= ylim, ygb <= ylim, c + 1 < c_end, std::true_type());
            }
          }
        } else {
          for (; c < c_end; c += 2, v += 2 * ph, cell += 2 * dcell)
            body(j, s, cell, cell + dcell, dx, cav, axis_coord_t<POW2>(ay, v), axis_coord_t<POW2>(ay, v + ph),
                 packed && v <= v1c, packed && v + ph <= v1c, c + 1 < c_end, std::true_type());
        }
      } else if constexpr ((kSphereCost & kSeg2Tag) != 0) {
        // column segments of any packing: chunk = (row group c / ncx, segment c % ncx) -- boxes wider than a wave (pw =
        // 64, ph = 1) and the two-segment boxes of 33 .. 64 columns where the packed walk does not take them
        const int ngr_c = (v1c - v0 + ph) / ph;
        const int c_end_w = min(c_end, ngr_c * ncx);
        for (; c < c_end_w; ++c) {
          const int g = rfl((int)(((float)c + 0.5f) / (float)ncx));
          const int u = u0 + __mul24(c - g * ncx, pw) + lx, v = v0 + __mul24(g, ph) + ly;
          const float dx = axis_coord_t<POW2>(ax, u) - s.x;
          body(j, s, (v - r0) * LW + u, 0, dx, packed ? rr - dx * dx : -1.f, axis_coord_t<POW2>(ay, v), 0.f,
               packed && u <= u1 && v <= v1c, false, false, std::true_type());
        }
      } else {   // a box wider than a wave: pw = 64, ph = 1, chunk = (row c / ncx, segment c % ncx)
        const int c_end_w = min(c_end, (v1c - v0 + 1) * ncx);
        for (; c < c_end_w; ++c) {
          const int g = rfl((int)(((float)c + 0.5f) / (float)ncx));
          const int u = u0 + ((c - g * ncx) << 6) + lane, v = v0 + g;
          const float dx = axis_coord_t<POW2>(ax, u) - s.x;
          body(j, s, (v - r0) * LW + u, 0, dx, rr - dx * dx, axis_coord_t<POW2>(ay, v), 0.f, u <= u1, false, false,
               std::true_type());
        }
      }
      end_sphere(j);
    }
    ++j;
  }
}

// Wave w of nwaves takes the w-th contiguous slice of the total weight.  The SIMD's issue
// arbitration favours its oldest wave, so with equal slices the 16 waves of a workgroup
// finish in a staircase (wave 0-3 first, 12-15 last: measured 6.3 k vs 10.3 k cycles);
// `shares` gives the four age groups (waves 4g..4g+3) their part of the list, one byte per
// group, oldest first, summing to 256 (the launcher normalises).  Other workgroup sizes
// split equally.
template <bool POW2, int kSphereCost, bool ROWFREE, bool TABLE = false, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_my_slice(const WaveList &w, int J, int total, int wave, int nwaves, int shares,
                                              int lane, const Axis &ax, const Axis &ay, int r0, int r1, int LW,
                                              Body &&body, EndSphere &&end_sphere,
                                              const RunTab rt = RunTab{nullptr, nullptr}) {
  wave = rfl(wave);   // everything that steers the loops is wave-uniform: keep it in SGPRs
  total = rfl(total);
  int lo, hi;
  if (nwaves == kZWaves) {
    const int g = wave >> 2, k = wave & 3;
    const int s0 = shares & 255, s1 = (shares >> 8) & 255, s2 = (shares >> 16) & 255, s3 = (shares >> 24) & 255;
    const int sg = g == 0 ? s0 : (g == 1 ? s1 : (g == 2 ? s2 : s3));
    const int before = 4 * ((g > 0 ? s0 : 0) + (g > 1 ? s1 : 0) + (g > 2 ? s2 : 0)) + k * sg;   // of 1024
    lo = (int)(((long long)total * before) >> 10);
    hi = wave == kZWaves - 1 ? total : (int)(((long long)total * (before + sg)) >> 10);
  } else {
    lo = (int)(((long long)wave * total) / nwaves);
    hi = (int)(((long long)(wave + 1) * total) / nwaves);
  }
  if (lo >= hi) return;
  if constexpr (TABLE) {
    static_assert(POW2, "the run table carries exact power-of-two coordinates");
    walk_slice_table<kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, rt, body, end_sphere);
  } else if constexpr (POW2) {
    if (r1 - r0 < 4096) {
      if constexpr ((kSphereCost & kSeg2Tag) != 0) walk_slice_packed_seg2<kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere);
      else walk_slice_packed<kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere);
    }
    else walk_slice<POW2, kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere);
  } else {
    walk_slice<POW2, kSphereCost, ROWFREE>(w, J, lo, hi, lane, ax, ay, r0, r1, LW, body, end_sphere);
  }
}

// image axis coordinate with the power-of-two case resolved at compile time
template <bool POW2>
__device__ __forceinline__ float axis_coord_t(const Axis &a, int u) {
  const float t = (float)u - a.half;
  return POW2 ? t * a.mul : (t * 300.0f) / a.size;
}

// Output stores are written THROUGH at agent scope (`sc1`): the depth map is read next by
// another kernel, possibly on another XCD, so every line has to reach memory before the
// kernel can retire anyway.  Left dirty in the L2 they are flushed by the end-of-kernel
// write-back (plain stores: forward 8.9 us; `nt`: 8.05 or 8.45 us depending on the process;
// `sc1`: 7.8 us in every process, tools/exp_storebits.sh).  Mode digits for the experiment
// builds (-DSHR_STORE_MODE=<depth><owner>, -DSHR_BWD_STORE_MODE=<grad>): 0 plain, 1 nt,
// 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1, 5 sc0, 6 sc1 nt.
#ifndef SHR_STORE_MODE
#define SHR_STORE_MODE 44
#endif
#ifndef SHR_BWD_STORE_MODE
#define SHR_BWD_STORE_MODE 4
#endif
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
// (the s_nop covers the "store of more than 64 bits, then its data registers rewritten"
// hazard the compiler cannot see inside an asm statement)
template <int M> __device__ __forceinline__ void asm_store16(void *p, v4u_t t) {
  if (M == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  if (M == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  if (M == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  if (M == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  if (M == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  if (M == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  if (M == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
template <int M> __device__ __forceinline__ void asm_store4(void *p, uint32_t t) {
  if (M == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(t) : "memory");
  if (M == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(t) : "memory");
  if (M == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
  if (M == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(t) : "memory");
  if (M == 4) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
  if (M == 5) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(t) : "memory");
  if (M == 6) asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void stream_store(float4 *p, const float4 v) {
  v4u_t t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  asm_store16<SHR_STORE_MODE / 10>(p, t);
}
__device__ __forceinline__ void stream_store(uint4 *p, const uint4 v) {
  v4u_t t = {v.x, v.y, v.z, v.w};
  asm_store16<SHR_STORE_MODE % 10>(p, t);
}
__device__ __forceinline__ void stream_store(uchar4 *p, const uchar4 v) {
  const uint32_t t = (uint32_t)v.x | ((uint32_t)v.y << 8) | ((uint32_t)v.z << 16) | ((uint32_t)v.w << 24);
  asm_store4<SHR_STORE_MODE % 10>(p, t);
}

// The touched rows' OWNER bytes (stored after the scan conversion) are the only part of the
// output the backward reads, next, from the same XCD (crop n runs on XCD n % 8 in both
// kernels): left in the L2 with a plain store instead of written through.  The forward alone
// gets 0.18 us slower (dirty lines for the end-of-kernel write-back), forward + backward
// 0.15 us faster (tools/exp_latestore.sh; -DSHR_LATE_STORE_MODE=<depth><owner> to vary).
#ifndef SHR_LATE_STORE_MODE
#define SHR_LATE_STORE_MODE 40
#endif
__device__ __forceinline__ void late_store(float4 *p, const float4 v) {
  v4u_t t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  asm_store16<SHR_LATE_STORE_MODE / 10>(p, t);
}
__device__ __forceinline__ void late_store(uchar4 *p, const uchar4 v) {
  const uint32_t t = (uint32_t)v.x | ((uint32_t)v.y << 8) | ((uint32_t)v.z << 16) | ((uint32_t)v.w << 24);
  asm_store4<SHR_LATE_STORE_MODE % 10>(p, t);
}

template <bool OWNER> struct KeyOf { using type = uint32_t; };
template <> struct KeyOf<true> { using type = unsigned long long; };

// Rows of the region that some sphere's pixel box touches, [cv0, cv1] inclusive (cv1 < cv0:
// none).  Every other row is background whatever the spheres' depths.  Evaluated with lanes =
// spheres by any wave from its own copy of the records: same instructions on the same inputs,
// so every wave of the workgroup gets the same answer without an LDS exchange.  A crop that
// takes the general path reports the whole region (nothing is known to be background).
__device__ __forceinline__ void touched_rows(const float4 s, bool valid, const Axis &ay, float ky, int r0, int r1,
                                             int &cv0, int &cv1) {
  const bool tame = sphere_is_tame(s);
  const unsigned long long bad = __ballot(valid && !(tame && fabsf(s.z) < 1e30f));
  const unsigned long long low = __ballot(valid && s.z <= kBackground);
  int v0 = r0, v1 = r1 - 1;
  if (tame) axis_box(s.y, fabsf(s.w), ky, ay.half, (float)r1 + 2.f, r0, r1 - 1, v0, v1);
  const bool on = valid && v1 >= v0;   // row numbers are < 2^24: exact in fp32
  cv0 = (int)wave_minmax_all<true>(on ? (float)v0 : 1e9f);
  cv1 = (int)wave_minmax_all<false>(on ? (float)v1 : -1e9f);
  if (cv1 < cv0) { cv0 = r1; cv1 = r0 - 1; }
  if (bad != 0ull || low == 0ull) { cv0 = r0; cv1 = r1 - 1; }
}

// The same for rows AND columns: the pixel box [cv0, cv1] x [cu0, cu1] that holds every pixel a sphere
// of the crop can touch inside rows [r0, r1) -- the only part of the region that needs a z-buffer.  The
// per-sphere ranges are sphere_item's (same function, same inputs).  One transposed four-component wave
// minimum (v0, -v1, u0, -u1) instead of four reductions.
__device__ __forceinline__ void touched_box(const float4 s, bool valid, const Axis &ax, const Axis &ay, float kx,
                                            float ky, int W, int r0, int r1, int lane, int &cv0, int &cv1, int &cu0,
                                            int &cu1) {
  const bool tame = sphere_is_tame(s);
  const unsigned long long bad = __ballot(valid && !(tame && fabsf(s.z) < 1e30f));
  const unsigned long long low = __ballot(valid && s.z <= kBackground);
  int v0 = r0, v1 = r1 - 1, u0 = 0, u1 = W - 1;
  if (tame) {
    const float ar = fabsf(s.w);
    axis_box(s.x, ar, kx, ax.half, (float)W + 2.f, 0, W - 1, u0, u1);
    axis_box(s.y, ar, ky, ay.half, (float)r1 + 2.f, r0, r1 - 1, v0, v1);
  }
  const bool on = valid && v1 >= v0 && u1 >= u0;   // (pixel numbers are < 2^24: exact in fp32)
  const float m = wave_min4_transposed(on ? (float)v0 : 1e9f, on ? -(float)v1 : 1e9f, on ? (float)u0 : 1e9f,
                                       on ? -(float)u1 : 1e9f, lane);
  cv0 = (int)readlane_f(m, 12);
  cv1 = -(int)readlane_f(m, 13);
  cu0 = (int)readlane_f(m, 14);
  cu1 = -(int)readlane_f(m, 15);
  if (cv1 < cv0) { cv0 = r1; cv1 = r0 - 1; cu0 = 0; cu1 = -1; }
  if (bad != 0ull || low == 0ull) { cv0 = r0; cv1 = r1 - 1; cu0 = 0; cu1 = W - 1; }
}

// Row pitch (cells) of a z-buffer over a box `bw` columns wide (bw % 4 == 0): kRowPad cells of padding, and never
// within 8 cells of a multiple of 32 -- the rows of a chunk would start in (nearly) the same LDS banks.
__host__ __device__ __forceinline__ int box_pitch(int bw) {
  const int p = bw + kRowPad;
  return (p & 31) < 8 ? p + 8 : p;
}
// the widest pitch any box of a W-pixel row can get
__host__ __device__ __forceinline__ int max_box_pitch(int W) { return ((W + 3) & ~3) + kRowPad + 8; }

// ---------------------------------------------------------------------------
// Forward.  grid = (N or fewer, nregions), block = 64 * nwaves (<= 1024), dynamic LDS = kHdrBytes +
// zcells * sizeof(key).
//
// The z-buffer covers only the TOUCHED BOX of the region -- the rows and columns some sphere's pixel box
// reaches (a hand crop: ~63 x 62 of 128 x 128 pixels) -- at a row pitch that follows the box's width.
// Everything outside it is background whatever the depths: whole background rows are stored straight from
// registers before the first barrier, the background columns of the touched rows are filled in by the
// stream-out.  A box of more than `zcells` cells is rasterized in PASSES over row bands (the work list is
// rebuilt clipped to each band), so that any crop is exact at any LDS budget: the launcher gives a
// workgroup half of the CU's LDS when the launch has at least two workgroups per CU -- two resident
// workgroups overlap each other's prologue (record read, work list, background stores), issue-bound scan
// conversion and stream-out, which one workgroup per CU runs strictly one after the other (5.9 -> 4.2 us
// per 256 crops for the depth-only forward at 9216 crops, tools/exp_twocu.py) -- and the whole of it
// otherwise (one pass for any 128 x 128 box).
//
// Schedule of one workgroup (in-kernel clock64, batch 256 = one crop per CU): the sphere
// read takes ~2 k cycles to arrive and wave 0 needs ~1.5 k more for the work list; the
// other waves fill that time with the z-buffer initialisation and then with the
// BACKGROUND ROWS: rows no sphere's box touches (half of a hand crop) are stored straight
// from registers before the first barrier and never pass through LDS or the decode.
template <bool OWNER, bool VEC4, bool POW2, bool PERSIST, bool BOX, bool TABLE = false, bool SEG2 = false>
__global__ void __launch_bounds__(1024)
sphere_zbuf_fwd_kernel(const float4 *__restrict__ spheres, int N, int J_, int H_, int W_,
                       float *__restrict__ depth, uint8_t *__restrict__ argmin, int rows_per_region_,
                       int w4_shift_flags, int shares, int zcells_, AxisK axk) {
  using Key = typename KeyOf<OWNER>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_sph = reinterpret_cast<float4 *>(smem);
  int4 *s_items = reinterpret_cast<int4 *>(smem + kOffItems);
  int *s_ends = reinterpret_cast<int *>(smem + kOffEnds);
  int *s_flag = reinterpret_cast<int *>(smem + kOffFlags);
  float4 *s_next = reinterpret_cast<float4 *>(smem + kOffNext);
  Key *zbuf = reinterpret_cast<Key *>(smem + kHdrBytes);
  // TABLE (whole-region z-buffer, one workgroup per CU, not persistent): the run table follows the z-buffer, the
  // spheres' run records take the place of the next crop's records (build_run_table)
  static_assert(!TABLE || (POW2 && VEC4 && !BOX && !PERSIST && !SEG2), "run table: whole-crop workgroups on power-of-two images");
  uint2 *s_tab = reinterpret_cast<uint2 *>(smem + kHdrBytes + (size_t)zcells_ * sizeof(Key));
  float4 *s_run = s_next;

  // PERSISTENT workgroups: with more crops than the launch has workgroups (gridDim.x < N) a workgroup takes crops
  // blockIdx.x, blockIdx.x + gridDim.x, ...  The youngest wave requests the NEXT crop's records right after the
  // first barrier and parks them in LDS before the second one, so every crop but a workgroup's first starts with
  // its records at hand (their first read is a 2-2.6 k-cycle round trip in front of everything else).  Worth 8 % of
  // the forward at 36 crops per CU, nothing at 4.5 (the launcher decides: sphere_raster.hip persistent_grid).
  const int crop_step = gridDim.x;
  for (int n = blockIdx.x, crop_it = 0; PERSIST ? n < N : crop_it == 0; n += crop_step, ++crop_it) {
  // Every crop starts from OPAQUE copies of the launch constants and of the thread index: otherwise the compiler
  // hoists each crop-invariant value out of the crop loop and keeps it in a register (forward: 92 instead of 51
  // VGPRs; the fused kernel spilled).
  // (w4_shift_flags: log2(W / 4) or -1 in the low byte (signed), the SHR_RASTER_* flags in the second, the waves per
  // workgroup in the third -- one SGPR less in a kernel whose box variant sits at the occupancy limit of 80)
  int J = J_, H = H_, W = W_, rows_per_region = rows_per_region_, w4_shift = (int)(signed char)(w4_shift_flags & 0xff), tid = threadIdx.x;
  const int flags = (w4_shift_flags >> 8) & 0xff;
  int zcells = zcells_;
  if (PERSIST) {
    asm volatile("" : "+s"(J), "+s"(H), "+s"(W), "+s"(rows_per_region), "+s"(w4_shift), "+s"(zcells));
    asm volatile("" : "+v"(tid));
  }
  const int lane = tid & 63, wave = tid >> 6;
  // BOX (two workgroups per CU): a workgroup in its PROLOGUE -- records, list, background rows: latency and stores, few
  // instructions, and every wave has to arrive -- outranks the co-resident workgroup's scan, whose waves otherwise win the
  // SIMDs' oldest-first arbitration and stretch this phase (in-kernel timeline: the youngest waves reach the first barrier
  // last).  Round 6, tools/ab_variant.py: forward + owner bytes 23.8 -> 22.5 us at 1152 crops, 41.9 -> 39.1 at 2304, depth-only
  // @256 x 256 57.8 -> 55.3; 9216 crops unchanged; the same bits.  (Priority 2 or 3: the same.  The stream-out at raised
  // priority: nothing.  The backward's staging at raised priority: +3 % at 9216 crops -- not taken.)
  if (BOX) __builtin_amdgcn_s_setprio(1);
  SHR_TL_ENTRY(0);
  // (the workgroup size rides in the same launch argument: blockDim.x is a hidden kernel argument that is NOT among
  // the preloaded ones -- reading it put an s_load round trip in front of the records' request)
  const int nwaves = (w4_shift_flags >> 16) & 0xff, nthr = nwaves << 6;
  const int r0 = blockIdx.y * rows_per_region;
  const int r1 = min(H, r0 + rows_per_region);
  const int rh = r1 - r0;
  Axis ax = axis_of(W, axk.mulx), ay = axis_of(H, axk.muly);
  const float kx = axk.kx, ky = axk.ky;   // pixels per millimetre (launch constants: common.h AxisK)

  // the waves that need the crop's records before the first barrier (wave 0: work list; the
  // background waves: touched box) read them from memory, lane j = sphere j; the others take
  // wave 0's LDS copy after the barrier (their requests would only lengthen the memory queue)
  const int wave_s = rfl(wave);
  const bool list_wave = wave_s == 0;
  // TABLE (16 waves): one job per wave in front of the first barrier, the vector arithmetic on the OLD waves (a
  // young wave gets an instruction through every ~12 cycles there, tools/exp_ztime.py: six of them took 3-3.9 k
  // cycles over the table): wave 0 the work list and then a seventh of the run table, waves 1 .. 3 the rest of
  // it, waves 4 .. 7 the z-buffer initialisation, the YOUNG waves 8 .. 15 the background rows -- without LDS
  // traffic of their own in front of the request their records are there ~0.7 k cycles after the wave's start.
  // (Measured and not kept on this split: the storing waves at raised priority, +0.1 us; the initialisation of the
  // touched rows only, once the records are in, +0.2 us; other work-list shares.)
  const int nbgw = TABLE ? 8 : min(BOX ? kBgWavesBox : kBgWaves, nwaves - 1);
  const int bg_first = TABLE ? 8 : 1;
  const bool bg_wave = nwaves == 1 || (wave_s >= bg_first && wave_s < bg_first + nbgw);
  const bool valid = lane < J;
  const bool pf_wave = wave_s == nwaves - 1 && !list_wave && !bg_wave;
  const bool tab_wave = TABLE && wave_s >= 1 && wave_s <= 3;
  const bool init_wave = TABLE && wave_s >= 4 && wave_s <= 7;
  float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid && (list_wave || bg_wave || tab_wave)) sph = crop_it == 0 ? spheres[(size_t)n * J + lane] : s_next[lane];
  // (the box variant sits at gfx950's occupancy limit of 80 SGPRs: its axis constants live in vector registers;
  // they are not among the preloaded arguments -- touched here, behind the records' request, not in front of it)
  if (BOX) asm volatile("" : "+v"(ax.mul), "+v"(ay.mul), "+v"(ax.half), "+v"(ay.half));

  const Key bg = OWNER ? (Key)background_cell() : (Key)depth_key(kBackground);
  auto init_zbuf = [&](int ncell) {   // background everywhere, by every wave: they are all waiting for the records
    constexpr int per16 = 16 / sizeof(Key);
    const int nvec = ncell / per16;
    if (OWNER) {
      const ulonglong2 v = make_ulonglong2(bg, bg);
      for (int i = tid; i < nvec; i += nthr) reinterpret_cast<ulonglong2 *>(zbuf)[i] = v;
    } else {
      const uint4 v = make_uint4((uint32_t)bg, (uint32_t)bg, (uint32_t)bg, (uint32_t)bg);
      for (int i = tid; i < nvec; i += nthr) reinterpret_cast<uint4 *>(zbuf)[i] = v;
    }
    for (int i = nvec * per16 + tid; i < ncell; i += nthr) zbuf[i] = bg;
  };
  // (the box is not known yet: every cell a one-pass box of this region can use; overlaps the read above.  Tried
  // in round 3 and dropped, tools/exp_ztime.py: the idle waves 8 .. 15 alone -- the young waves' LDS stores take
  // until 3.0-3.8 k cycles and hold the barrier; whole-crop z-buffers on the central half of the rows first and the
  // touched rows outside it by the storing waves -- their stores start later than the initialisation saves)
  if (!TABLE) init_zbuf(BOX ? min(zcells, rh * max_box_pitch(W)) : rh * (W + kRowPad));
  if (init_wave) {   // TABLE: the whole z-buffer by waves 4 .. 7
    constexpr int per16 = 16 / sizeof(Key);
    const ulonglong2 v2 = make_ulonglong2((unsigned long long)bg, (unsigned long long)bg);
    const uint4 v4 = make_uint4((uint32_t)bg, (uint32_t)bg, (uint32_t)bg, (uint32_t)bg);
    for (int i = tid - 256; i < rh * (W + kRowPad) / per16; i += 256) {   // (pitch and key size: whole 16-byte pieces)
      if (OWNER) reinterpret_cast<ulonglong2 *>(zbuf)[i] = v2;
      else reinterpret_cast<uint4 *>(zbuf)[i] = v4;
    }
  }

  // Waves 1..kBgWaves store the background rows while wave 0 builds the list: they are the
  // first to finish the z-buffer initialisation (the SIMD arbitration favours old waves) and
  // a wave issues a wave-wide store every ~55 cycles, so ~6 stores apiece fit in wave 0's
  // shadow.  The other waves learn the touched box after the barrier.

  // The records are waited for HERE by every wave -- the waves that asked for them need them now, the others asked for nothing.
  // Left to the compiler, the wait sits on each path's first use, and along the paths of the waves that never use the loaded
  // value it stays "pending": wherever those registers are written later (in the scan, in the stream-out) the compiler puts a
  // vmcnt(0) -- which at run time waits for the stores issued since: vmcnt counts loads and stores in one in-order queue and
  // the stream stores are asm statements the compiler does not count.  The storing waves stood at the scan's entry until
  // their background rows had reached memory.
#ifndef EXP_NO_RECORD_FENCE
  asm volatile("" : : "v"(sph.x), "v"(sph.y), "v"(sph.z), "v"(sph.w));
#endif
  if (list_wave) {
    s_sph[lane] = sph;
    // general path unless every sphere is tame and at least one has z <= 100: a pixel's
    // minimum can exceed the background only where ALL J spheres hit it, and there the
    // sphere with z <= 100 contributes z - sqrt(q) < 100, so min(100, hits) is exact
    // (a hit at exactly 100.0: background_cell() / tie_owner()).
    const unsigned long long bad = __ballot(valid && !(sphere_is_tame(sph) && fabsf(sph.z) < 1e30f));
    const unsigned long long low = __ballot(valid && sph.z <= kBackground);
    const unsigned long long behind = __ballot(valid && sph.z > kBackground);
    SHR_TL(0, 6);   // (list wave) the crop's records have arrived
    bool too_big;   // excluded by the launcher (W <= kMaxFastWidth, H <= 32768)
    const int total = build_work_list<kSphereCostFwd | (SEG2 ? kSeg2Tag : 0)>(sph, valid, ax, ay, kx, ky, W, r0, r1, s_items, s_ends, lane, &too_big,
                                                      TABLE ? s_run : nullptr);
    SHR_TL(0, 7);   // (list wave) the work list stands
    if (lane == 0) {
      s_flag[0] = (bad != 0ull) || (low == 0ull) || too_big;
      s_flag[1] = total;
      s_flag[12] = behind != 0ull;   // only a sphere centred behind the background can hit at exactly 100.0 (tie_owner)
    }
  }

  if (TABLE && wave_s <= 3) {
    // sphere j belongs to slot j mod 7: slots 0-1 -> wave 1, 2-3 -> wave 2, 4-5 -> wave 3, slot 6 -> the list wave,
    // which has a third of its time in front of the barrier left when the list stands
    const unsigned long long every7 = 0x8102040810204081ull;   // bits 0, 7, 14, ..., 63
    const unsigned long long all = J >= 64 ? ~0ull : ((1ull << J) - 1ull);
    const int sl = wave_s == 0 ? 6 : 2 * (wave_s - 1);
    const unsigned long long mine = ((every7 << sl) | (wave_s == 0 ? 0ull : every7 << (sl + 1))) & all;
    build_run_table<POW2>(sph, valid, ax, ay, kx, ky, W, r0, r1, W + kRowPad, s_tab, lane, mine);
  }

  float *out = depth + (size_t)n * H * W;
  uint8_t *aout = OWNER ? argmin + (size_t)n * H * W : nullptr;

  // The region is streamed in UNITS of 64 consecutive 16-byte chunks (a wave-wide store =
  // whole rows or a row segment); whether a unit lies in background rows is a scalar test.
  const int w4 = W >> 2;
  const int nchunk = rh * w4;
  const int nunits = (nchunk + 63) >> 6;
  float4 *out4 = reinterpret_cast<float4 *>(out + (size_t)r0 * W);
  uchar4 *aout4 = OWNER ? reinterpret_cast<uchar4 *>(aout + (size_t)r0 * W) : nullptr;
  // units [0, ua) and [ub, nunits) lie entirely in background rows, [ua, ub) is touched
  int ua = 0, ub = nunits;
  int4 bx = make_int4(0, 0, 0, 0);
  if (bg_wave && (BOX || VEC4)) {
    int cv0, cv1, cu0 = 0, cu1 = W - 1;
    if (BOX) touched_box(sph, valid, ax, ay, kx, ky, W, r0, r1, lane, cv0, cv1, cu0, cu1);
    else touched_rows(sph, valid, ay, ky, r0, r1, cv0, cv1);
    if (VEC4) {
      if (cv1 < cv0) ua = ub = nunits;
      else { ua = ((cv0 - r0) * w4) >> 6; ub = min(nunits, ((cv1 - r0 + 1) * w4 + 63) >> 6); }
      ua = rfl(ua);
      ub = rfl(ub);
    }
    // (the publishing wave keeps the box; see below)
    bx = make_int4(cv0, cv1, cu0, cu1);
  }
  // ---- background rows: stored by the waves that wait for the work list ---------------
  const float4 bgd = make_float4(kBackground, kBackground, kBackground, kBackground);
  const uchar4 bga = make_uchar4(SHR_ARGMIN_NONE, SHR_ARGMIN_NONE, SHR_ARGMIN_NONE, SHR_ARGMIN_NONE);
  // owner bytes go out 16 at a time when rows allow it (a wave-wide store of 4-byte pieces
  // occupies the write queue like a 16-byte one and carries a quarter of the data)
  const bool own16 = OWNER && (W & 15) == 0 && is_aligned16(argmin);
  // SHR_RASTER_OWNER_TOUCHED_ROWS: the owner bytes of background ROWS stay unwritten -- the backward stages and walks
  // the touched rows only (same touched_rows() on the same records), so a forward whose owner map is only ever handed
  // to shr_sphere_raster_bwd saves a tenth of its stores (8 of a hand crop's 82 KB)
  const bool bg_owner = OWNER && !(flags & SHR_RASTER_OWNER_TOUCHED_ROWS);
  auto store_background = [&](int first, int step) {
    const int nbg = ua + (nunits - ub);
    for (int t = first; t < nbg; t += step) {
      const int u = t < ua ? t : t - ua + ub;
      const int c = (u << 6) + lane;
      if (c < nchunk) {
        stream_store(out4 + c, bgd);
        if (bg_owner && !own16) stream_store(aout4 + c, bga);
      }
    }
    if (bg_owner && own16) {   // 16-pixel pieces: unit u = pieces [16u, 16u + 16)
      const int npiece = nchunk >> 2, pa = ua << 4, pb = ub << 4;
      const int nbgp = min(pa, npiece) + max(npiece - pb, 0);
      const uint4 bg16 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      static_assert(SHR_ARGMIN_NONE == 255, "background owner bytes");
      uint4 *aout16 = reinterpret_cast<uint4 *>(aout + (size_t)r0 * W);
      for (int t = first * 64 + lane; t < nbgp; t += step * 64) {
        const int pc = t < pa ? t : t - pa + pb;
        stream_store(aout16 + pc, bg16);
      }
    }
  };
  if (VEC4 && bg_wave) store_background(nwaves == 1 ? 0 : wave_s - bg_first, nwaves == 1 ? 1 : nbgw);
  if (!BOX) {
    if (VEC4 && wave_s == (nwaves == 1 ? 0 : bg_first) && lane == 0) { s_flag[2] = ua; s_flag[3] = ub; }   // for the other waves
  } else if (wave_s == (nwaves == 1 ? 0 : bg_first) && lane == 0) {
    // Everything the other waves derive from the box, computed ONCE (the stores above drain meanwhile): sixteen
    // waves repeating this scalar arithmetic -- a division among it -- after the barrier cost 2 k cycles per crop.
    const int cv0 = bx.x, cv1 = bx.y;
    const int cu0 = bx.z & ~3;                                          // 16-byte chunks lie inside or outside the box
    const int bw = bx.w >= cu0 ? ((bx.w | 3) - cu0 + 1) : 4;
    const int pitch = box_pitch(bw);
    // the touched part of the region in 16-byte chunks (VEC4) / pixels (otherwise): what the stream-out writes
    const int row_len = VEC4 ? w4 : W;
    const int out_lo = VEC4 ? ua << 6 : 0, out_hi = VEC4 ? min(ub << 6, nchunk) : rh * W;
    // Rows [cv0, pe) of the box fit the z-buffer; a box with more rows than that (possible only at the small
    // budget of two workgroups per CU, or when a test squeezes it) leaves rows [pe, cv1] -- from a tile boundary
    // on -- to the general tile code, which writes whole rows itself: no second pass, no loop around the scan.
    const int split = (cv0 + zcells / pitch) & ~(kTileH - 1);           // (zcells / pitch >= 8: the launcher's budget)
    const bool over = split <= cv1;
    s_flag[2] = out_lo;
    s_flag[3] = over ? min(out_hi, (split - r0) * row_len) : out_hi;
    s_flag[4] = cv0; s_flag[5] = over ? split : cv1 + 1; s_flag[6] = cu0; s_flag[7] = bw;
    s_flag[8] = pitch;
    s_flag[9] = over ? split : r1;                                      // the scan conversion's clip row
    // tile rows: from the split to the end of the touched units (rows below the box inside them are background,
    // which the tile code reproduces), on tile boundaries
    s_flag[10] = over ? split : r1;
    s_flag[11] = over ? min(r1, (r0 + (out_hi + row_len - 1) / row_len + kTileH - 1) & ~(kTileH - 1)) : r1;
  }
  SHR_TL(0, 1);   // this wave's work in front of the first barrier is done (list / table / background rows / init)
  __syncthreads();
  if (BOX) __builtin_amdgcn_s_setprio(0);
  SHR_TL(0, 2);   // past the first barrier: the scan starts
  if (!(list_wave || bg_wave || tab_wave)) sph = s_sph[lane];
  const bool has_next = PERSIST && n + crop_step < N;   // (the launcher keeps a prefetch wave whenever gridDim.x < N)
  float4 sph_next = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pf_wave && has_next && valid) sph_next = spheres[(size_t)(n + crop_step) * J + lane];

  const bool general = s_flag[0] != 0;   // workgroup-uniform: the whole region takes the tile code
  const bool may_tie = rfl(s_flag[12]) != 0;
  int tile_lo = r0, tile_hi = r1;        // rows for the tile code
  if (!general) {
    // (BOX = false: the z-buffer holds the whole region at the image's own pitch -- one workgroup per CU has the LDS
    // for it, and nothing has to be derived from a box)
    int out_lo = 0, out_hi = VEC4 ? nchunk : rh * W;
    if (!BOX && VEC4) { out_lo = rfl(s_flag[2]) << 6; out_hi = min(rfl(s_flag[3]) << 6, nchunk); }
    if (BOX) { out_lo = rfl(s_flag[2]); out_hi = rfl(s_flag[3]); }
    const int p0 = BOX ? rfl(s_flag[4]) : r0, pe = BOX ? rfl(s_flag[5]) : r1;
    const int cu0 = BOX ? rfl(s_flag[6]) : 0, bw = BOX ? rfl(s_flag[7]) : W;
    const int pitch = BOX ? rfl(s_flag[8]) : W + kRowPad, clip = BOX ? rfl(s_flag[9]) : r1;
    tile_lo = BOX ? rfl(s_flag[10]) : r1;
    tile_hi = BOX ? rfl(s_flag[11]) : r1;

    // ---- scan-convert the chunk list ---------------------------------------------------
    // A chunk may reach below its sphere's box (rows that are real pixels, or lie beyond the
    // z-buffer's last row): the hit test is exact for ANY pixel and fails there, nothing is written.
    {
      WaveList wl;
      wl.sph = sph;
      wl.item = s_items[lane];
      wl.end = s_ends[lane];
      Key *zb = zbuf - (p0 * pitch + cu0);   // cell of pixel (v, u) = zb[v * pitch + u]
      walk_my_slice<POW2, kSphereCostFwd | (SEG2 ? kSeg2Tag : 0), true, TABLE>(
          wl, J, s_flag[1], wave, nwaves, shares, lane, ax, ay, 0, clip, pitch,
          [&](int j, const float4 s, int cell_a, int cell_b, float, float ca, float yga, float ygb, bool ok_a,
              bool ok_b, bool has_b, auto row_test) {
            const float dya = yga - s.y, dyb = ygb - s.y;
            float qa = ca - dya * dya, qb = ca - dyb * dyb;
            if (decltype(row_test)::value) { qa = ok_a ? qa : -1.f; qb = ok_b ? qb : -1.f; }
            // (the owner index as an opaque vector value: the compiler then keeps it in the low register of the
            // 64-bit pair across the run instead of re-materialising it from the scalar in front of every atomic)
            unsigned jv = (unsigned)j;
            if (OWNER) asm("" : "+v"(jv));
            auto put = [&](Key *cell, float d) {
              if (OWNER)
                atomicMin(reinterpret_cast<unsigned long long *>(cell),
                          ((unsigned long long)depth_key(d) << 32) | jv);
              else
                atomicMin(reinterpret_cast<unsigned int *>(cell), depth_key(d));
            };
            if (has_b) {  // wave-uniform; branch-free up to the atomics: both roots in flight together
              const bool ha = qa > kHitMin, hb = qb > kHitMin;
              // (the root of a non-hit lane's q may be NaN: never stored)
              const float da = s.z - sqrt_rn(qa), db = s.z - sqrt_rn(qb);
              if (ha) put(zb + cell_a, da);
              if (hb) put(zb + cell_b, db);
            } else if (qa > kHitMin) {
              put(zb + cell_a, s.z - sqrt_rn(qa));
            }
          },
          [](int) {}, RunTab{s_tab, s_run});
    }
    if (pf_wave && has_next) s_next[lane] = sph_next;   // (arrived long ago: the wave's own scan slice lies in between)
    SHR_TL(0, 3);   // this wave's scan slice is done
    __syncthreads();
    SHR_TL(0, 4);   // past the second barrier: the stream-out starts

    // ---- stream the touched rows out: [out_lo, out_hi) of the region's chunks / pixels -------
    const Key *zrow = zbuf - (p0 - r0) * pitch - cu0;   // cell of region pixel (v, x) = zrow[v * pitch + x]
    const unsigned box_h = pe > p0 ? (unsigned)(pe - p0) : 0u;
    if (VEC4) {
      for (int cb = out_lo + (wave_s << 6); cb < out_hi; cb += nwaves << 6) {   // (a scalar loop: wave-wide stores of whole units)
        const int c = cb + lane;
        if (c >= out_hi) continue;
        int v, x;
        if (POW2 || w4_shift >= 0) { v = c >> w4_shift; x = (c & (w4 - 1)) << 2; }
        else { v = c / w4; x = (c - v * w4) << 2; }
        float4 o = bgd;
        uchar4 a = bga;
        if (!BOX || ((unsigned)(v + r0 - p0) < box_h && (unsigned)(x - cu0) < (unsigned)bw)) {
          const Key *cell = zrow + v * pitch + x;
          if (OWNER) {
            const ulonglong2 k01 = reinterpret_cast<const ulonglong2 *>(cell)[0];
            const ulonglong2 k23 = reinterpret_cast<const ulonglong2 *>(cell)[1];
            o = make_float4(key_depth((uint32_t)(k01.x >> 32)), key_depth((uint32_t)(k01.y >> 32)),
                            key_depth((uint32_t)(k23.x >> 32)), key_depth((uint32_t)(k23.y >> 32)));
            a = make_uchar4((uint8_t)k01.x, (uint8_t)k01.y, (uint8_t)k23.x, (uint8_t)k23.y);
            if (may_tie && (is_background_tie(k01.x) || is_background_tie(k01.y) || is_background_tie(k23.x) ||
                            is_background_tie(k23.y))) {   // (practically never)
              const float yg = axis_coord_t<POW2>(ay, v + r0);
              if (is_background_tie(k01.x)) a.x = (uint8_t)tie_owner(s_sph, (uint32_t)k01.x, axis_coord_t<POW2>(ax, x), yg);
              if (is_background_tie(k01.y)) a.y = (uint8_t)tie_owner(s_sph, (uint32_t)k01.y, axis_coord_t<POW2>(ax, x + 1), yg);
              if (is_background_tie(k23.x)) a.z = (uint8_t)tie_owner(s_sph, (uint32_t)k23.x, axis_coord_t<POW2>(ax, x + 2), yg);
              if (is_background_tie(k23.y)) a.w = (uint8_t)tie_owner(s_sph, (uint32_t)k23.y, axis_coord_t<POW2>(ax, x + 3), yg);
            }
          } else {
            const uint4 k = *reinterpret_cast<const uint4 *>(cell);
            o = make_float4(key_depth(k.x), key_depth(k.y), key_depth(k.z), key_depth(k.w));
          }
        }
        if (OWNER) late_store(aout4 + c, a);
        late_store(out4 + c, o);
      }
    } else {
      for (int p = out_lo + tid; p < out_hi; p += nthr) {
        const int v = p / W, u = p - v * W;
        Key k = bg;
        if (!BOX || ((unsigned)(v + r0 - p0) < box_h && (unsigned)(u - cu0) < (unsigned)bw)) k = zrow[v * pitch + u];
        if (OWNER) {
          out[(size_t)(r0 + v) * W + u] = key_depth((uint32_t)((unsigned long long)k >> 32));
          aout[(size_t)(r0 + v) * W + u] =
              (may_tie && is_background_tie((unsigned long long)k))
                  ? (uint8_t)tie_owner(s_sph, (uint32_t)k, axis_coord_t<POW2>(ax, u), axis_coord_t<POW2>(ay, v + r0))
                  : (uint8_t)k;
        } else {
          out[(size_t)(r0 + v) * W + u] = key_depth((uint32_t)k);
        }
      }
    }
  }
  if (tile_lo < tile_hi) {   // workgroup-uniform: a crop on the general path, or the rows its z-buffer could not hold
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int t0 = (tile_lo / kTileH) * tiles_x, t1 = ((tile_hi + kTileH - 1) / kTileH) * tiles_x;
    tile_forward<VEC4, OWNER>(sph, J, H, W, out, aout, tiles_x, t0 + wave, t1, nwaves, lane);
  }
  if (general && pf_wave && has_next) s_next[lane] = sph_next;
  SHR_TL(0, 5);   // end: the touched rows are streamed out
  if (PERSIST && n + crop_step < N) __syncthreads();   // the z-buffer is re-initialised next: every wave's stream-out reads are done
  }  // crops
}

// ---------------------------------------------------------------------------
// Backward with the forward's owner map.  grid = (N or fewer), block = 64 * NW (NW = 16, or 8 when two
// workgroups share a CU), dynamic LDS = kHdrBytes + 16*64*16 (wave x sphere partial sums) + rows * (W +
// kRowPad) * (4 + 1), `rows` = the rows of the crop the staging buffers hold.
//
// Only the TOUCHED ROWS [cv0, cv1] of the crop are staged and walked, from row cv0 on: a crop whose touched rows
// exceed the buffers is done in passes of `rows` rows (the work list is rebuilt clipped to each pass, the
// partial sums carry over).  With at least two workgroups per CU in the launch the launcher gives a workgroup
// half of the CU's LDS (88 rows of a 128-wide crop: one pass for 9 hand crops of 10) and 8 waves, so that two
// resident workgroups overlap one crop's staging (HBM latency) with the other's walk (VALU issue).
constexpr int kPartBytes = kZWaves * SHR_MAX_SPHERES * 16;
constexpr int kStageBatch = 4;   // backward: units (16-byte chunks per lane) requested per wait while staging
constexpr int kSpecUnits = 2;    // ... and units per wave requested before the touched rows are known
// The speculative units land in v[96:105], registers the compiler does not allocate (the kernel is limited to the
// 96 below them): a request whose destination is an asm OUTPUT counts as available at once, and the compiler did
// copy such registers -- before the data was there -- as soon as the path from the request to its wait branched.
constexpr int kBwdVgprs = 96;

template <bool VEC4, bool POW2, bool PERSIST, int NW, bool WHOLE, bool SEG2 = false>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_num_vgpr(kBwdVgprs)))
sphere_zbuf_bwd_kernel(const float4 *__restrict__ spheres, const float *__restrict__ grad_depth,
                       const uint8_t *__restrict__ argmin, int N, int J_, int H_, int W_,
                       float4 *__restrict__ grad_spheres, int rows_, int w4_shift_, int shares, AxisK axk) {
  static_assert(NW == 8 || NW == 16, "waves per workgroup");
  constexpr int NT = 64 * NW, NW_SHIFT = NW == 16 ? 4 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_sph = reinterpret_cast<float4 *>(smem);
  int4 *s_items = reinterpret_cast<int4 *>(smem + kOffItems);
  int *s_ends = reinterpret_cast<int *>(smem + kOffEnds);
  int *s_flag = reinterpret_cast<int *>(smem + kOffFlags);
  float4 *s_next = reinterpret_cast<float4 *>(smem + kOffNext);
  float4 *s_part = reinterpret_cast<float4 *>(smem + kHdrBytes);
  // PERSISTENT workgroups (gridDim.x < N: crops blockIdx.x, blockIdx.x + gridDim.x, ...): the youngest wave requests
  // the next crop's records after the staging barrier and parks them in LDS (see the forward).
  const int crop_step = gridDim.x;
  for (int n = blockIdx.x, crop_it = 0; PERSIST ? n < N : crop_it == 0; n += crop_step, ++crop_it) {
  // Every crop starts from OPAQUE copies of the launch constants and of the thread index: otherwise the compiler
  // hoists each crop-invariant value out of the crop loop and keeps it in a register (forward: 92 instead of 51
  // VGPRs; the fused kernel spilled).
  int J = J_, H = H_, W = W_, rows = rows_, w4_shift = w4_shift_, tid = threadIdx.x;
  if (PERSIST) {
    asm volatile("" : "+s"(J), "+s"(H), "+s"(W), "+s"(rows), "+s"(w4_shift));
    asm volatile("" : "+v"(tid));
  }
  const int lane = tid & 63, wave = tid >> 6;
  SHR_TL_ENTRY(1);
  const int LW = W + kRowPad;
  float *gbuf = reinterpret_cast<float *>(smem + kHdrBytes + kPartBytes);
  uint8_t *obuf = smem + kHdrBytes + kPartBytes + (size_t)(rows + kPadRows) * LW * 4;
  const Axis ax = axis_of(W, axk.mulx), ay = axis_of(H, axk.muly);
  const float kx = axk.kx, ky = axk.ky;   // pixels per millimetre (launch constants: common.h AxisK)
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int wave_s = rfl(wave);
  // The four OLDEST waves (one per SIMD: the arbitration serves them first, their requests head the memory
  // queue) read the crop's records (lane j = sphere j), with an explicit instruction so that the staging requests
  // below can be queued behind it and awaited separately: wave 0 builds the work list, all four derive the
  // touched rows.  A young wave's copy of the records arrived up to 3 k cycles later and held the barrier.
  const bool lead = wave_s < 4;
  int lead_v0 = 0, lead_v1 = -1;   // the touched rows, in the lead waves
  const bool pf_wave = wave_s == NW - 1;
  const float *gin = grad_depth + (size_t)n * H * W;
  const uint8_t *oin = argmin + (size_t)n * H * W;
  const float4 *rec = spheres + (size_t)n * J;
  const bool has_next = PERSIST && n + crop_step < N;
  // (the records land in v[110:113], the next crop's in v[106:109]: registers outside the compiler's, see kBwdVgprs)
  float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
  bool have_sph = false;
  if (crop_it == 0) {
    if (lead)
      asm volatile("global_load_dwordx4 v[110:113], %0, off" : : "v"(rec + min(lane, J - 1)) : "v110", "v111", "v112", "v113", "memory");
  } else if (lead) {
    if (lane < J) sph = s_next[lane];
    have_sph = true;
  }
  for (int i = tid; i < kZWaves * SHR_MAX_SPHERES; i += NT) s_part[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // [wave][sphere]

  // ---- stage grad_depth + owner map ------------------------------------------------------
  // Reading is the bandwidth-bound part (82 KB per 128x128 crop arrive in ~8 k cycles when all CUs read), and
  // only the rows some sphere's box touches are ever looked at (half of a hand crop).  Those rows are known once
  // the records have arrived, so the staging is split: the central half of the crop is requested at once,
  // SPECULATIVELY, by all waves (units = 64 consecutive 16-byte chunks of the crop, unit u belongs to wave u mod
  // NW); what the touched rows need beyond it is requested when the records are in.
  const int w4 = W >> 2;
  const int nchunk = H * w4;
  const int nunits = (nchunk + 63) >> 6;
  const float4 *gin4 = reinterpret_cast<const float4 *>(gin);
  const uchar4 *oin4 = reinterpret_cast<const uchar4 *>(oin);
  int uc0 = nunits >> 2;                                          // the speculative units: the central [uc0, uc1)
  int uc1 = min(nunits - uc0, uc0 + NW * kSpecUnits);
  const int k1 = (max(uc0 - wave_s, 0) + NW - 1) >> NW_SHIFT;     // this wave's first central unit is wave + NW k1
  static_assert(kSpecUnits == 2, "two speculative units per wave: v[96:99] + v104, v[100:103] + v105");
  if (VEC4) {
    // (an absent unit re-reads the records: the count of requests in flight stays fixed)
    const int u0 = wave_s + (k1 << NW_SHIFT), u1 = wave_s + ((k1 + 1) << NW_SHIFT);
    const int c0 = min((u0 << 6) + lane, nchunk - 1), c1 = min((u1 << 6) + lane, nchunk - 1);
    const void *pg0 = u0 < uc1 ? static_cast<const void *>(gin4 + c0) : static_cast<const void *>(rec);
    const void *po0 = u0 < uc1 ? static_cast<const void *>(oin4 + c0) : static_cast<const void *>(rec);
    const void *pg1 = u1 < uc1 ? static_cast<const void *>(gin4 + c1) : static_cast<const void *>(rec);
    const void *po1 = u1 < uc1 ? static_cast<const void *>(oin4 + c1) : static_cast<const void *>(rec);
    asm volatile("global_load_dwordx4 v[96:99], %0, off\n\tglobal_load_dword v104, %1, off\n\t"
                 "global_load_dwordx4 v[100:103], %2, off\n\tglobal_load_dword v105, %3, off"
                 : : "v"(pg0), "v"(po0), "v"(pg1), "v"(po1)
                 : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "memory");
  }
  // owner padding = "nobody": the walk may overhang the image edge
  for (int i = tid; i < rows * kRowPad; i += NT) obuf[(i / kRowPad) * LW + W + (i % kRowPad)] = SHR_ARGMIN_NONE;
  if (lead) {
    if (!have_sph) {
      float4 t;
      if (VEC4)
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v110\n\tv_mov_b32 %1, v111\n\tv_mov_b32 %2, v112\n\tv_mov_b32 %3, v113"
                     : "=v"(t.x), "=v"(t.y), "=v"(t.z), "=v"(t.w) : "n"(2 * kSpecUnits) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v110\n\tv_mov_b32 %1, v111\n\tv_mov_b32 %2, v112\n\tv_mov_b32 %3, v113"
                     : "=v"(t.x), "=v"(t.y), "=v"(t.z), "=v"(t.w) : : "memory");
      if (lane < J) sph = t;
    }
    touched_rows(sph, lane < J, ay, ky, 0, H, lead_v0, lead_v1);
    SHR_TL(1, 6);   // (lead waves) the records have arrived, the touched rows are known
    if (wave_s == 0) {
      s_sph[lane] = sph;
      if (!WHOLE && lane == 0) { s_flag[2] = lead_v0; s_flag[3] = lead_v1; }
    }
  }
  // WHOLE (the buffers hold the whole crop: one workgroup per CU): rows sit at their own index, one pass over
  // [0, H), and only the lead waves -- which know the touched rows -- request what lies outside the speculative
  // units, at once.  Otherwise the rows start at the first touched one, which every wave has to know before it
  // writes a unit into LDS: one more barrier (the others arrive at it straight from their requests).
  int cv0 = 0, cv1 = H - 1;
  if (!WHOLE) {
    __syncthreads();
    cv0 = rfl(s_flag[2]);
    cv1 = rfl(s_flag[3]);
  }

  // Passes over the touched rows (one, unless they exceed the staging buffers).  The first pass's list and
  // staging stand in front of the loop: the registers of the speculative requests must not become loop-carried
  // values (the compiler would copy them at the loop's entry, before the data has landed).
  auto build_list = [&](int r0, int r1) {
    if (wave_s == 0) {
      bool too_big;   // excluded by the launcher (W <= kMaxFastWidth)
      const int total = build_work_list<kSphereCostBwd | (SEG2 ? kSeg2Tag : 0)>(sph, lane < J, ax, ay, kx, ky, W, r0, r1, s_items, s_ends, lane, &too_big);
      if (lane == 0) s_flag[1] = total;
      SHR_TL(1, 7);   // (wave 0) the work list stands
    }
  };
  auto stage = [&](int r0, int r1, int t_lo, int t_hi, auto first_tag) {   // rows [t_lo, t_hi) of [r0, r1) are wanted
    constexpr bool FIRST = decltype(first_tag)::value;
    const int rh = r1 - r0;
    auto put_unit = [&](int u, const v4f g, uint32_t o) {
      const int c = (u << 6) + lane;
      int v, x;
      if (POW2 || w4_shift >= 0) { v = c >> w4_shift; x = (c & (w4 - 1)) << 2; }
      else { v = c / w4; x = (c - v * w4) << 2; }
      if (c < nchunk && (unsigned)(v - r0) < (unsigned)rh) {   // (a unit may straddle the pass's first or last row)
        *reinterpret_cast<v4f *>(gbuf + (v - r0) * LW + x) = g;
        *reinterpret_cast<uint32_t *>(obuf + (v - r0) * LW + x) = o;
      }
    };
    if (VEC4) {
      // the pass's units [ua, ub); those outside the speculative range (first pass only), [ua, min(ub, uc0)) and
      // [max(ua, uc1), ub), are dealt to the waves now
      const int ua = t_hi > t_lo ? (t_lo * w4) >> 6 : 0, ub = t_hi > t_lo ? min(nunits, (t_hi * w4 + 63) >> 6) : 0;
      const int s0 = FIRST ? uc0 : 0, s1 = FIRST ? uc1 : 0;
      const int n_lo = max(0, min(ub, s0) - ua), hi0 = max(ua, s1), n_out = n_lo + max(0, ub - hi0);
      // A batch = kStageBatch units per wave; its requests and their wait are ONE asm statement: the compiler
      // treats an asm result as available at once and may copy it before a separate wait.  A slot without a unit
      // re-reads the records (no control flow around the statement).  The first pass's first batch also lands
      // the speculative units (registers outside the compiler's, kBwdVgprs) and fetches them.
      static_assert(kStageBatch == 4, "the staging statements below name four slots");
      static_assert(kSpecUnits == 2, "... and tie two speculative units");
      constexpr int DW = WHOLE ? 3 : NW - 1;   // the units are dealt to waves 1 .. DW
      auto batch = [&](int t0, auto with_spec_tag) {
        constexpr bool first_batch = decltype(with_spec_tag)::value;
        v4f g2[kStageBatch];
        uint32_t o2[kStageBatch];
        bool ok[kStageBatch];
        const void *pg[kStageBatch], *po[kStageBatch];
        int us[kStageBatch];
#pragma unroll
        for (int b = 0; b < kStageBatch; b++) {
          const int t = t0 + DW * b;
          ok[b] = t < n_out;                                                  // wave-uniform
          const int u = us[b] = t < n_lo ? ua + t : hi0 + (t - n_lo);
          const int c = min(max((u << 6) + lane, 0), nchunk - 1);
          pg[b] = ok[b] ? static_cast<const void *>(gin4 + c) : static_cast<const void *>(rec);
          po[b] = ok[b] ? static_cast<const void *>(oin4 + c) : static_cast<const void *>(rec);
        }
        if (FIRST && first_batch) {
          float4 sa, sb;       // the two speculative units, fetched from their registers once everything has landed
          uint32_t soa, sob;
          if (!ok[0]) {        // no unit for this wave (wave-uniform): only the speculative ones to land
            asm volatile(
                "s_waitcnt vmcnt(0)\n\t"
                "v_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v98\n\tv_mov_b32 %3, v99\n\t"
                "v_mov_b32 %4, v100\n\tv_mov_b32 %5, v101\n\tv_mov_b32 %6, v102\n\tv_mov_b32 %7, v103\n\t"
                "v_mov_b32 %8, v104\n\tv_mov_b32 %9, v105"
                : "=&v"(sa.x), "=&v"(sa.y), "=&v"(sa.z), "=&v"(sa.w), "=&v"(sb.x), "=&v"(sb.y), "=&v"(sb.z),
                  "=&v"(sb.w), "=&v"(soa), "=&v"(sob)
                : : "memory");
          } else
          asm volatile(
              "global_load_dwordx4 %0, %18, off\n\tglobal_load_dword %4, %22, off\n\t"
              "global_load_dwordx4 %1, %19, off\n\tglobal_load_dword %5, %23, off\n\t"
              "global_load_dwordx4 %2, %20, off\n\tglobal_load_dword %6, %24, off\n\t"
              "global_load_dwordx4 %3, %21, off\n\tglobal_load_dword %7, %25, off\n\t"
              "s_waitcnt vmcnt(0)\n\t"      // ... the speculative units have landed too: fetch them
              "v_mov_b32 %8, v96\n\tv_mov_b32 %9, v97\n\tv_mov_b32 %10, v98\n\tv_mov_b32 %11, v99\n\t"
              "v_mov_b32 %12, v100\n\tv_mov_b32 %13, v101\n\tv_mov_b32 %14, v102\n\tv_mov_b32 %15, v103\n\t"
              "v_mov_b32 %16, v104\n\tv_mov_b32 %17, v105"
              : "=&v"(g2[0]), "=&v"(g2[1]), "=&v"(g2[2]), "=&v"(g2[3]), "=&v"(o2[0]), "=&v"(o2[1]), "=&v"(o2[2]),
                "=&v"(o2[3]), "=&v"(sa.x), "=&v"(sa.y), "=&v"(sa.z), "=&v"(sa.w), "=&v"(sb.x), "=&v"(sb.y),
                "=&v"(sb.z), "=&v"(sb.w), "=&v"(soa), "=&v"(sob)
              : "v"(pg[0]), "v"(pg[1]), "v"(pg[2]), "v"(pg[3]), "v"(po[0]), "v"(po[1]), "v"(po[2]), "v"(po[3])
              : "memory");
          const int u0 = wave_s + (k1 << NW_SHIFT), u1 = wave_s + ((k1 + 1) << NW_SHIFT);
          if (u0 < uc1) put_unit(u0, v4f{sa.x, sa.y, sa.z, sa.w}, soa);
          if (u1 < uc1) put_unit(u1, v4f{sb.x, sb.y, sb.z, sb.w}, sob);
        } else {
          asm volatile(
              "global_load_dwordx4 %0, %8, off\n\tglobal_load_dword %4, %12, off\n\t"
              "global_load_dwordx4 %1, %9, off\n\tglobal_load_dword %5, %13, off\n\t"
              "global_load_dwordx4 %2, %10, off\n\tglobal_load_dword %6, %14, off\n\t"
              "global_load_dwordx4 %3, %11, off\n\tglobal_load_dword %7, %15, off\n\t"
              "s_waitcnt vmcnt(0)"
              : "=&v"(g2[0]), "=&v"(g2[1]), "=&v"(g2[2]), "=&v"(g2[3]), "=&v"(o2[0]), "=&v"(o2[1]), "=&v"(o2[2]),
                "=&v"(o2[3])
              : "v"(pg[0]), "v"(pg[1]), "v"(pg[2]), "v"(pg[3]), "v"(po[0]), "v"(po[1]), "v"(po[2]), "v"(po[3])
              : "memory");
        }
#pragma unroll
        for (int b = 0; b < kStageBatch; b++)
          if (ok[b]) put_unit(us[b], g2[b], o2[b]);
      };
      int t0 = (wave_s == 0 || wave_s > DW) ? n_out : wave_s - 1;   // (wave 0 builds the list: its requests would go out last)
      if (FIRST) {
        batch(t0, std::true_type());
        t0 += DW * kStageBatch;
      }
      for (; t0 < n_out; t0 += DW * kStageBatch) batch(t0, std::false_type());
    } else {
      for (int p = tid; p < rh * W; p += NT) {
        const int v = p / W, u = p - v * W;
        gbuf[v * LW + u] = gin[(size_t)(r0 + v) * W + u];
        obuf[v * LW + u] = oin[(size_t)(r0 + v) * W + u];
      }
    }
  };
  if (WHOLE) {
    build_list(0, H);
    stage(0, H, lead_v0, lead_v1 + 1, std::true_type());
  } else if (cv1 >= cv0) {
    build_list(cv0, min(cv0 + rows, cv1 + 1));
    stage(cv0, min(cv0 + rows, cv1 + 1), cv0, min(cv0 + rows, cv1 + 1), std::true_type());
  } else if (VEC4) {   // no touched row at all: the speculative requests still have to land before their registers are reused
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
  }
  if (pf_wave && has_next)   // the next crop's records (its own staging requests have landed: nothing queues behind them)
    asm volatile("global_load_dwordx4 v[106:109], %0, off" : : "v"(spheres + (size_t)(n + crop_step) * J + min(lane, J - 1))
                 : "v106", "v107", "v108", "v109", "memory");
  for (int r0 = cv0; r0 <= cv1;) {
    const int r1 = min(r0 + rows, cv1 + 1), rh = r1 - r0;
    SHR_TL(1, 1);   // this wave's staging (and, wave 0, the list) is done
    __syncthreads();
    SHR_TL(1, 2);   // past the staging barrier: the walk starts

    // Static schedule: wave w walks the w-th contiguous slice of the list (which wave sums which pixels must
    // not depend on timing); at the end of a run on a sphere its register partials are reduced with one
    // transposed wave sum into the wave's private LDS slot.
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int cell_max = rh * LW - 1;
    const WaveList wl = load_wave_list(s_sph, s_items, s_ends, lane);
    walk_my_slice<POW2, kSphereCostBwd | (SEG2 ? kSeg2Tag : 0), false>(
        wl, J, s_flag[1], wave, NW, shares, lane, ax, ay, r0, r1, LW,
        [&](int j, const float4 s, int cell_a, int cell_b, float dx, float ca, float yga, float ygb, bool ok_a,
            bool ok_b, bool has_b, auto) {
          // lanes without a pixel may point past the pass: clamped, and never counted.  Most chunks own nothing: only
          // the owner byte is looked at -- BOTH chunks' bytes requested together --, dy and q are formed for owned
          // pixels (a branch-free form with all four LDS reads in flight measured slower)
          const int ca_ = min(cell_a, cell_max), cb_ = min(cell_b, cell_max);
          const uint8_t oa = obuf[ca_], ob = obuf[cb_];
          auto take = [&](int cell, uint8_t o, float yg, bool ok) {
            if (o == (uint8_t)j && ok) {
              const float dy = yg - s.y, q = ca - dy * dy;
              const float g = gbuf[cell];
              const float w = g * __builtin_amdgcn_rsqf(q);  // g / sqrt(q), ~1e-7 rel.
              a0 = __builtin_fmaf(-w, dx, a0);
              a1 = __builtin_fmaf(-w, dy, a1);
              a2 += g;
              a3 -= w;
            }
          };
          take(ca_, oa, yga, ok_a);
          if (has_b) take(cb_, ob, ygb, ok_b);
        },
        [&](int j) {
          // (the slot belongs to this wave: ds_add_f32 in program order, no read-back to wait for;
          // a crop of several passes visits a sphere once per pass)
          const float t = wave_sum4_transposed(a0, a1, a2, a3, lane);
          if (lane >= 60) atomicAdd(reinterpret_cast<float *>(s_part + wave * SHR_MAX_SPHERES + j) + (lane & 3), t);
          a0 = a1 = a2 = a3 = 0.f;
        });
    r0 += rows;
    if (r0 <= cv1) {
      __syncthreads();  // this pass's walk is done: list and buffers are rewritten
      build_list(r0, min(r0 + rows, cv1 + 1));
      stage(r0, min(r0 + rows, cv1 + 1), r0, min(r0 + rows, cv1 + 1), std::false_type());
    }
  }
  if (pf_wave && has_next) {   // (arrived long ago: the wave's own walk lies in between)
    float4 t;
    asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v106\n\tv_mov_b32 %1, v107\n\tv_mov_b32 %2, v108\n\tv_mov_b32 %3, v109"
                 : "=v"(t.x), "=v"(t.y), "=v"(t.z), "=v"(t.w) : : "memory");
    s_next[lane] = t;
  }
  SHR_TL(1, 3);   // this wave's walk is done
  __syncthreads();
  SHR_TL(1, 4);   // past the closing barrier
  // combine the waves' partials in wave order; d/dr = r * sum(-g/sqrt(q))
  if (tid < J) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < NW; w++) {
      const float4 a = s_part[w * SHR_MAX_SPHERES + tid];
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    t.w = t.w * s_sph[tid].w;
    const v4u_t tt = {__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)};
    asm_store16<SHR_BWD_STORE_MODE>(grad_spheres + (size_t)n * J + tid, tt);
  }
  SHR_TL(1, 5);   // end: gradients stored
  if (has_next) __syncthreads();   // the partials are zeroed and the staging buffers refilled next
  }  // crops
}

// ---------------------------------------------------------------------------
// Fused render-and-compare: depth = raster(spheres), e = depth - target, per-crop
// sum(e*e) and d sum(e*e) / d spheres in ONE kernel.  Replaces, for the model->data
// term of mesh/multiview_utility.py:98-101 / :107-113, the chain rasterize -> MSELoss
// -> MSELoss backward -> rasterizer backward: no owner map, no gradient image and no
// difference image ever reach HBM (read 4 S^2 of target, write 4 S^2 of depth if it is
// wanted; the unfused chain moves >= 50 bytes per pixel).
//
// grid = (N, nregions), block = 1024, dynamic LDS = kHdrBytes + kPartBytes + rows * (W +
// kRowPad) * 8.  The forward's schedule up to the z-buffer, then
//   convert  every pixel: e = d - t, the thread's running sum of e*e; touched rows rewrite
//            their z-buffer cell as (bits(2e) << 32 | owner);
//   walk     the backward's walk, reading gradient and owner from that cell;
// outputs are PER REGION (sse[n * R + region], grad[(n * R + region) * J + j]) and summed
// by the caller (R = 1 up to 128x128).  A crop that needs the general path evaluates its
// tiles like sphere_tile_bwd_kernel with the gradient formed in registers.
constexpr int kSphereCostMse = 28;

// (BOX: the z-buffer covers the touched box only, `zcells` cells, as in the forward -- half of a CU's LDS and, with
// 64 VGPRs, two workgroups per CU; the rows a box has beyond it go through the tile code of the general path, which
// adds to the same partial sums.  Needs a power-of-two image at least 32 wide: tile rows are then whole units.)
template <bool POW2, bool PERSIST, bool BOX, bool SEG2 = false>
__device__ __forceinline__ void
sphere_zbuf_mse_body(const float4 *__restrict__ spheres, int N, int J_, int H_, int W_, const float *__restrict__ target,
                     const int *__restrict__ target_index, float *__restrict__ depth,
                     float *__restrict__ sse_out, float4 *__restrict__ grad_out, int rows_per_region_,
                     int w4_shift_, int shares_fwd, int shares_bwd, int zcells_, AxisK axk,
                     const int *__restrict__ crop_index) {
  using Key = unsigned long long;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_sph = reinterpret_cast<float4 *>(smem);
  int4 *s_items = reinterpret_cast<int4 *>(smem + kOffItems);
  int *s_ends = reinterpret_cast<int *>(smem + kOffEnds);
  int *s_flag = reinterpret_cast<int *>(smem + kOffFlags);
  float4 *s_next = reinterpret_cast<float4 *>(smem + kOffNext);
  float4 *s_part = reinterpret_cast<float4 *>(smem + kHdrBytes);                          // [wave][J]
  Key *zbuf = reinterpret_cast<Key *>(smem + kHdrBytes + (size_t)kZWaves * J_ * sizeof(float4));

  // PERSISTENT workgroups (gridDim.x < N: crops blockIdx.x, blockIdx.x + gridDim.x, ... of this region): the youngest
  // wave requests the next crop's records after the first barrier and parks them in LDS (see the forward).
  const int crop_step = gridDim.x;
  for (int n = blockIdx.x, crop_it = 0; PERSIST ? n < N : crop_it == 0; n += crop_step, ++crop_it) {
  int region = blockIdx.y;
  const int nregions = gridDim.y;
  // Launch order of a crop's row regions (the box variant's grid is crops x regions, crops fastest): from the second quarter
  // on, the first quarter last.  A hand crop's outer regions are mostly rows no sphere touches -- workgroups of ~3 us that
  // only store background and sum the error, against ~25 us for a region with the hand in it: all of them at the END of the
  // launch they fill the slots the last round of long workgroups leaves idle (1152 crops @256 x 256: 149.5 -> 147.4 us; the other
  // rotations 153 / 162 us -- a long workgroup in the tail -- and any order that MIXES short and long ones 162-200 us: the dispatcher
  // hands workgroups to the CUs in turn, not to the first free one).
#ifndef EXP_MSE_NO_ROTATE
  if (BOX && !PERSIST && (nregions & 3) == 0) region = (region + (nregions >> 2)) & (nregions - 1);
#endif
  // Every crop starts from OPAQUE copies of the launch constants and of the thread index: otherwise the compiler
  // hoists each crop-invariant value out of the crop loop and keeps it in a register (forward: 92 instead of 51
  // VGPRs; the fused kernel spilled).
  // (w4_shift_: log2(W / 4) or -1 in the low byte (signed); bit 8: the partial results go to the CROP's slots, not the
  // workgroup's -- shr_sphere_raster_mse_ordered, where crop_index is a permutation that only changes the launch order)
  int J = J_, H = H_, W = W_, rows_per_region = rows_per_region_, w4_shift = (int)(signed char)(w4_shift_ & 0xff), tid = threadIdx.x;
  const bool slot_by_crop = (w4_shift_ & 0x100) != 0;
  int zcells = zcells_;
  if (PERSIST) {
    asm volatile("" : "+s"(J), "+s"(H), "+s"(W), "+s"(rows_per_region), "+s"(w4_shift), "+s"(zcells));
    asm volatile("" : "+v"(tid));
  }
  const int lane = tid & 63, wave = tid >> 6;
  // (BOX: the prologue and the convert pass -- the phases that wait for memory and issue the stores -- outrank the co-resident
  // workgroup's scan / walk, see the forward: 1152 crops @128 x 128 43.3 -> 42.5 us, 9216 crops 302 -> 293, @256 x 256 160.1 -> 159.5)
  if (BOX) __builtin_amdgcn_s_setprio(1);
#ifdef EXP_MSE_STAGGER   // (timing experiment: the second workgroup of every CU starts EXP_MSE_STAGGER x 3.4 us late)
  {
    const unsigned wg = blockIdx.x + blockIdx.y * gridDim.x;
    if (wg >= 256u && wg < 512u)
      for (int k = 0; k < EXP_MSE_STAGGER; k++) __builtin_amdgcn_s_sleep(127);
  }
#endif
  SHR_TL_ENTRY(2);
  const int r0 = region * rows_per_region;
  const int r1 = min(H, r0 + rows_per_region);
  const int rh = r1 - r0;
  const int LW = W + kRowPad;
  Axis ax = axis_of(W, axk.mulx), ay = axis_of(H, axk.muly);
  const float kx = axk.kx, ky = axk.ky;   // pixels per millimetre (launch constants: common.h AxisK)
  const int wave_s = rfl(wave);
  // (ONE wave derives the touched box and publishes it: this kernel stores the background rows in its convert pass -- see
  // there --, so the forward's seven storing waves would each evaluate the box for nothing; EXP_MSE_BG_PROLOGUE brings them back)
#if defined(EXP_MSE_BG_PROLOGUE) || defined(EXP_MSE_SEVEN_BOXES)
  constexpr int kBgW = BOX ? kBgWavesBox : kBgWaves;
#else
  constexpr int kBgW = 1;
#endif
  const bool bg_wave = wave_s >= 1 && wave_s <= kBgW;
  const bool valid = lane < J;
  const bool pf_wave = wave_s == kZWaves - 1;
  const bool has_next = PERSIST && n + crop_step < N;
  // crop_index (shr_sphere_raster_mse_indexed): workgroup n renders crop c = crop_index[n] of the batch -- its records,
  // its observed image, its slot of the depth output -- and reports into slot n of the partial results
#if defined(EXP_MSE_EMPTY) && EXP_MSE_EMPTY == 3   // (timing experiment: the launch alone -- every workgroup returns at once)
  if (tid == 0 && n == 0x7fffffff) sse_out[0] = 0.f;
  return;
#endif
  const int c = crop_index ? crop_index[n] : n;
  float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid && (wave_s == 0 || bg_wave))   // the others: wave 0's LDS copy, later
    sph = crop_it == 0 ? spheres[(size_t)c * J + lane] : s_next[lane];
  // (behind the records' request: the index is a scalar load, the output pointers are not among the preloaded arguments)
  const float *tgt = target + (size_t)(target_index ? target_index[c] : c) * H * W + (size_t)r0 * W;
  float *out = depth ? depth + (size_t)c * H * W + (size_t)r0 * W : nullptr;
  if (BOX) asm volatile("" : "+v"(ax.mul), "+v"(ay.mul), "+v"(ax.half), "+v"(ay.half));   // (SGPR budget, see the forward)
  if (tid < kZWaves * J) s_part[tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // (16 waves x J <= 64 spheres)
  {  // background everywhere (BOX: every cell a box of this region can use)
    const Key bg = (Key)background_cell();
    const int ninit = BOX ? min(zcells, rh * max_box_pitch(W)) : rh * LW;
    const int nvec = ninit >> 1;
    const ulonglong2 v = make_ulonglong2(bg, bg);
    for (int i = tid; i < nvec; i += 1024) reinterpret_cast<ulonglong2 *>(zbuf)[i] = v;
    if (tid == 0 && (ninit & 1)) zbuf[ninit - 1] = bg;
  }

  const int w4 = W >> 2;
  const int nchunk = rh * w4;
  const int nunits = (nchunk + 63) >> 6;
  const float4 *tgt4 = reinterpret_cast<const float4 *>(tgt);
  float4 *out4 = reinterpret_cast<float4 *>(out);
  // The observed image is needed after the scan conversion only (units u = wave + 16 k go to this
  // wave whatever the spheres are): requested now, it arrives under the list building and the scan
  // conversion instead of costing the convert pass one HBM round trip per unit.
  constexpr int kTgtAhead = 4;   // a 128x128 crop / a 64-row region of a 256-wide one: 64 units, four per wave
  float4 tpre[kTgtAhead];
#ifdef EXP_MSE_LATE_TARGET    // (experiment: requested past the first barrier, under the scan, instead of at the kernel's entry)
#elif !defined(EXP_MSE_SKIP_TARGET)   // (timing experiment: the observed image is never read)
#pragma unroll
  for (int k = 0; k < kTgtAhead; k++) tpre[k] = tgt4[min(((wave_s + (k << 4)) << 6) + lane, nchunk - 1)];
#else
#pragma unroll
  for (int k = 0; k < kTgtAhead; k++) tpre[k] = make_float4((float)lane, 1.f, 2.f, (float)k);
#endif
  int ua = 0, ub = nunits;
#ifndef EXP_NO_RECORD_FENCE
  asm volatile("" : : "v"(sph.x), "v"(sph.y), "v"(sph.z), "v"(sph.w));   // (the records: waited for by every wave, see the forward)
#endif
  if (bg_wave) {   // rows no sphere touches: depth = background, stored while wave 0 builds the list
    int cv0, cv1, cu0 = 0, cu1 = W - 1;
    if (BOX) touched_box(sph, valid, ax, ay, kx, ky, W, r0, r1, lane, cv0, cv1, cu0, cu1);
    else touched_rows(sph, valid, ay, ky, r0, r1, cv0, cv1);
    if (cv1 < cv0) ua = ub = nunits;
    else { ua = ((cv0 - r0) * w4) >> 6; ub = min(nunits, ((cv1 - r0 + 1) * w4 + 63) >> 6); }
    ua = rfl(ua);
    ub = rfl(ub);
    if (wave_s == 1 && lane == 0) {
      s_flag[2] = ua; s_flag[3] = ub;
      if (BOX) {   // everything the other waves derive from the box (see the forward)
        cu0 &= ~3;
        const int bw = cu1 >= cu0 ? ((cu1 | 3) - cu0 + 1) : 4;
        int pitch = box_pitch(bw);
        // A box whose rows do not all fit at that pitch would leave its last rows to the tile code -- at 256 x 256 nearly
        // every hand region: 64 rows x 136 cells against 8 504 in half of a CU's LDS, i.e. EIGHT rows of tile code per
        // region, 14-17 us of the kernel's 160 (round 6, EXP_MSE_SKIP_TILE).  A tighter row pitch (even: 16-byte cell pairs;
        // never a multiple of 32 cells) that holds the whole box is taken instead: the padding only spreads a chunk's rows
        // over the banks, no lane writes there, and the waves wait for LDS 1-2 % of their cycles.
        if ((cv1 - cv0 + 1) * pitch > zcells) {
          for (int pad = kRowPad - 2; pad >= 2; pad -= 2)
            if (((bw + pad) & 31) != 0 && (cv1 - cv0 + 1) * (bw + pad) <= zcells) { pitch = bw + pad; break; }
        }
        const int split = (cv0 + zcells / pitch) & ~(kTileH - 1);        // rows [cv0, split) fit the z-buffer
        const bool over = split <= cv1;
        s_flag[4] = cv0; s_flag[5] = over ? split : cv1 + 1; s_flag[6] = cu0; s_flag[7] = bw;
        s_flag[8] = pitch;
        s_flag[9] = over ? split : r1;                                   // the walks' clip row
        // tile rows: from the split to the end of the touched units (whole units: W >= 32 is a power of two)
        s_flag[10] = over ? split : r1;
        s_flag[11] = over ? min(r1, (r0 + ((ub << 6) + w4 - 1) / w4 + kTileH - 1) & ~(kTileH - 1)) : r1;
      }
    }
    // The background rows' depth is NOT stored here, as the forward does it, but by the convert pass below, unit by unit:
    // this kernel has the observed image's pieces in flight from its entry, and a store issued behind them makes every later
    // wait for a piece a wait for that store (one in-order vmcnt queue).  Stored here, the untouched rows cost the kernel
    // 7 us more at 1152 crops @256 x 256 (157.2 -> 150.0), 2.1 of 41.7 at 128 x 128, 18 of 295 at 9216 crops.
#ifdef EXP_MSE_BG_PROLOGUE
    if (out) {
      const float4 bgd = make_float4(kBackground, kBackground, kBackground, kBackground);
      const int nbg = ua + (nunits - ub);
      for (int t = wave_s - 1; t < nbg; t += kBgW) {
        const int u = t < ua ? t : t - ua + ub;
        const int c = (u << 6) + lane;
        if (c < nchunk) stream_store(out4 + c, bgd);
      }
    }
#endif
  }
  if (wave_s == 0) {
    s_sph[lane] = sph;
    const unsigned long long bad = __ballot(valid && !(sphere_is_tame(sph) && fabsf(sph.z) < 1e30f));
    const unsigned long long low = __ballot(valid && sph.z <= kBackground);
    const unsigned long long behind = __ballot(valid && sph.z > kBackground);
    bool too_big;
    const int total = build_work_list<kSphereCostMse | (SEG2 ? kSeg2Tag : 0)>(sph, valid, ax, ay, kx, ky, W, r0, r1, s_items, s_ends, lane, &too_big);
    if (lane == 0) {
      s_flag[0] = (bad != 0ull) || (low == 0ull) || too_big;
      s_flag[1] = total;
      s_flag[12] = behind != 0ull;   // only a sphere centred behind the background can hit at exactly 100.0 (tie_owner)
    }
  }
  SHR_TL(2, 1);   // this wave's work in front of the first barrier is done
  __syncthreads();
  if (BOX) __builtin_amdgcn_s_setprio(0);
#ifdef EXP_MSE_LATE_TARGET
#pragma unroll
  for (int k = 0; k < kTgtAhead; k++) tpre[k] = tgt4[min(((wave_s + (k << 4)) << 6) + lane, nchunk - 1)];
#endif
  SHR_TL(2, 2);   // past the first barrier
#if defined(EXP_MSE_EMPTY) && EXP_MSE_EMPTY == 1   // (timing experiment: the prologue alone -- every workgroup returns past the first barrier)
  if (tid == 0) sse_out[(size_t)n * nregions + region] = tpre[0].x + tpre[1].y + tpre[2].z + tpre[3].w + (float)s_flag[1];
  return;
#endif
  if (!(wave_s == 0 || bg_wave)) sph = s_sph[lane];
  float4 sph_next = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pf_wave && has_next && valid)
    sph_next = spheres[(size_t)(crop_index ? crop_index[n + crop_step] : n + crop_step) * J + lane];
  const bool general = s_flag[0] != 0;
  const bool may_tie = rfl(s_flag[12]) != 0;
  ua = rfl(s_flag[2]);
  ub = rfl(s_flag[3]);
  // the z-buffer's rows [p0, pe) and columns [cu0, cu0 + bw) at `pitch` (BOX = false: the whole region at the image's
  // own), the walks' clip row, and the rows [tile_lo, tile_hi) that go through the tile code
  const int p0 = BOX ? rfl(s_flag[4]) : r0, pe = BOX ? rfl(s_flag[5]) : r1;
  const int cu0 = BOX ? rfl(s_flag[6]) : 0, bw = BOX ? rfl(s_flag[7]) : W;
  const int pitch = BOX ? rfl(s_flag[8]) : LW, clip = BOX ? rfl(s_flag[9]) : r1;
  // (both words read whatever `general` says: as conditional reads they gave the compiler a branch structure with a dead edge
  // from here to the closing reductions, along which it believed the observed image's pieces still in flight -- and made every
  // wave wait for its depth stores there)
  int flag_lo = BOX ? rfl(s_flag[10]) : r1, flag_hi = BOX ? rfl(s_flag[11]) : r1;
  if (BOX) asm volatile("" : "+s"(flag_lo), "+s"(flag_hi));
  const int tile_lo = general ? r0 : flag_lo, tile_hi = general ? r1 : flag_hi;
  Key *zb = zbuf - (p0 * pitch + cu0);   // cell of pixel (v, u) = zb[v * pitch + u]

  float sse = 0.f;
  if (!general) {
    // ---- scan-convert (forward) ------------------------------------------------------------
    WaveList wl;
    wl.sph = sph;
    wl.item = s_items[lane];
    wl.end = s_ends[lane];
    // (EXP_MSE_SKIP_*: timing-only ablations of tools/ab_variant.py -- wrong results, never in the product build)
#ifndef EXP_MSE_SKIP_SCAN
    walk_my_slice<POW2, kSphereCostMse | (SEG2 ? kSeg2Tag : 0), true>(
        wl, J, s_flag[1], wave, kZWaves, shares_fwd, lane, ax, ay, 0, clip, pitch,
        [&](int j, const float4 s, int cell_a, int cell_b, float, float ca, float yga, float ygb, bool ok_a,
            bool ok_b, bool has_b, auto row_test) {
          const float dya = yga - s.y, dyb = ygb - s.y;
          float qa = ca - dya * dya, qb = ca - dyb * dyb;
          if (decltype(row_test)::value) { qa = ok_a ? qa : -1.f; qb = ok_b ? qb : -1.f; }
#ifdef EXP_MSE_SCAN_JUNK   // (timing experiment: EXP_MSE_SCAN_JUNK independent VALU instructions more per chunk pair)
          {
            float junk = qa;
#pragma unroll
            for (int k = 0; k < EXP_MSE_SCAN_JUNK; k++) asm volatile("v_add_f32 %0, %1, %1" : "=v"(junk) : "v"(qb));
            asm volatile("" : : "v"(junk));
          }
#endif
          unsigned jv = (unsigned)j;
          asm("" : "+v"(jv));   // (kept in the low register of the 64-bit pair across the run, see the forward)
          auto put = [&](Key *cell, float d) { atomicMin(cell, ((Key)depth_key(d) << 32) | jv); };
          if (has_b) {
            const bool ha = qa > kHitMin, hb = qb > kHitMin;
            const float da = s.z - sqrt_rn(qa), db = s.z - sqrt_rn(qb);
            if (ha) put(zb + cell_a, da);
            if (hb) put(zb + cell_b, db);
          } else if (qa > kHitMin) {
            put(zb + cell_a, s.z - sqrt_rn(qa));
          }
        },
        [](int) {});
#endif
    if (pf_wave && has_next) s_next[lane] = sph_next;
    SHR_TL(2, 3);   // this wave's scan slice is done
    // (a region no sphere touches -- the top and the bottom quarter of a 256 x 256 hand crop cut into four 64-row
    // regions: the list is empty, every unit of the convert pass is a background unit that never looks at the
    // z-buffer, the walk has nothing to visit: the two barriers around the convert pass order nothing)
    const bool untouched = ua >= ub && tile_lo >= tile_hi;
    if (!untouched) __syncthreads();
    if (BOX) __builtin_amdgcn_s_setprio(1);
    SHR_TL(2, 4);   // past the second barrier: the convert pass starts

    // ---- convert: error, its square, gradient image in place ---------------------------------
    auto convert_unit = [&](int u, const float4 t) {
      const int c = (u << 6) + lane;
      if (c >= nchunk) return;
      int v, x;
      if (POW2 || w4_shift >= 0) { v = c >> w4_shift; x = (c & (w4 - 1)) << 2; }
      else { v = c / w4; x = (c - v * w4) << 2; }
      if (BOX && v + r0 >= tile_lo && v + r0 < tile_hi) return;   // rows left to the tile code (whole units, background ones among them)
      if (u < ua || u >= ub) {   // background rows
        const float e0 = kBackground - t.x, e1 = kBackground - t.y, e2 = kBackground - t.z, e3 = kBackground - t.w;
        sse += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
#if !defined(EXP_MSE_BG_PROLOGUE) && !defined(EXP_MSE_NO_BG_STORE)
        if (out) stream_store(out4 + c, make_float4(kBackground, kBackground, kBackground, kBackground));
#endif
        return;
      }
      if (BOX) {
        if (!((unsigned)(v + r0 - p0) < (unsigned)(pe - p0) && (unsigned)(x - cu0) < (unsigned)bw)) {
          // a touched row beside the box (or a background row inside a touched unit)
          const float e0 = kBackground - t.x, e1 = kBackground - t.y, e2 = kBackground - t.z, e3 = kBackground - t.w;
          sse += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
          if (out) stream_store(out4 + c, make_float4(kBackground, kBackground, kBackground, kBackground));
          return;
        }
      }
      ulonglong2 *cell = reinterpret_cast<ulonglong2 *>(zb + (v + r0) * pitch + x);
      ulonglong2 k01 = cell[0], k23 = cell[1];
      const float4 d = make_float4(key_depth((uint32_t)(k01.x >> 32)), key_depth((uint32_t)(k01.y >> 32)),
                                   key_depth((uint32_t)(k23.x >> 32)), key_depth((uint32_t)(k23.y >> 32)));
      if (out) stream_store(out4 + c, d);
      if (may_tie && (is_background_tie(k01.x) || is_background_tie(k01.y) || is_background_tie(k23.x) ||
                      is_background_tie(k23.y))) {   // a hit at exactly 100.0 (practically never): who owns it (tie_owner)
        const float yg = axis_coord_t<POW2>(ay, v + r0);
        if (is_background_tie(k01.x)) k01.x = (k01.x & ~0xffull) | tie_owner(s_sph, (uint32_t)k01.x, axis_coord_t<POW2>(ax, x), yg);
        if (is_background_tie(k01.y)) k01.y = (k01.y & ~0xffull) | tie_owner(s_sph, (uint32_t)k01.y, axis_coord_t<POW2>(ax, x + 1), yg);
        if (is_background_tie(k23.x)) k23.x = (k23.x & ~0xffull) | tie_owner(s_sph, (uint32_t)k23.x, axis_coord_t<POW2>(ax, x + 2), yg);
        if (is_background_tie(k23.y)) k23.y = (k23.y & ~0xffull) | tie_owner(s_sph, (uint32_t)k23.y, axis_coord_t<POW2>(ax, x + 3), yg);
      }
      const float e0 = d.x - t.x, e1 = d.y - t.y, e2 = d.z - t.z, e3 = d.w - t.w;
      sse += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
      // (the low word stays as it is -- the owner's index, 0xFFFFFFFF on a background cell: the walk compares its low byte)
      k01.x = ((Key)__float_as_uint(2.f * e0) << 32) | (uint32_t)k01.x;
      k01.y = ((Key)__float_as_uint(2.f * e1) << 32) | (uint32_t)k01.y;
      k23.x = ((Key)__float_as_uint(2.f * e2) << 32) | (uint32_t)k23.x;
      k23.y = ((Key)__float_as_uint(2.f * e3) << 32) | (uint32_t)k23.y;
      cell[0] = k01;
      cell[1] = k23;
    };
#ifndef EXP_MSE_SKIP_CONVERT
    // Every requested piece of the observed image is waited for HERE, before the pass issues its first store.  vmcnt counts
    // loads and stores in one in-order queue: left alone, the wait in front of piece k covers the stores of units 0 .. k - 1
    // -- conditional, so the compiler cannot count them -- and becomes vmcnt(0) at the last unit and again at the walk's
    // entry (the pieces' registers are reused): the depth stores' write latency on every wave's critical path, twice.
    // The pieces were requested at the kernel's entry; nothing else is in flight but a background wave's early stores.
#ifndef EXP_MSE_NO_LOAD_FENCE
#pragma unroll
    for (int k = 0; k < kTgtAhead; k++) asm volatile("" : : "v"(tpre[k].x), "v"(tpre[k].y), "v"(tpre[k].z), "v"(tpre[k].w));
#endif
#pragma unroll
    for (int k = 0; k < kTgtAhead; k++)
      if (wave_s + (k << 4) < nunits) convert_unit(wave_s + (k << 4), tpre[k]);
    for (int u = wave_s + (kTgtAhead << 4); u < nunits; u += kZWaves) {
      const int c = (u << 6) + lane;
      float4 t = tgt4[min(c, nchunk - 1)];
#ifndef EXP_MSE_NO_LOAD_FENCE
      asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));   // (as above: no piece is left pending on the paths that do not use it)
#endif
      convert_unit(u, t);
    }
#else
    sse += tpre[0].x + tpre[1].y + tpre[2].z + tpre[3].w;
#endif
    SHR_TL(2, 5);   // this wave's convert units are done
    if (BOX) __builtin_amdgcn_s_setprio(0);
    if (!untouched) __syncthreads();
    SHR_TL(2, 6);   // past the third barrier: the walk starts

    // ---- walk (backward): static slices, per-run DPP sums into the wave's LDS row ------------
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int cell_max = (p0 * pitch + cu0) + (pe - p0) * pitch - 1;   // the z-buffer's last cell, as the walk counts cells
#ifndef EXP_MSE_SKIP_WALK
    walk_my_slice<POW2, kSphereCostMse | (SEG2 ? kSeg2Tag : 0), true>(
        wl, J, s_flag[1], wave, kZWaves, shares_bwd, lane, ax, ay, 0, clip, pitch,
        [&](int j, const float4 s, int cell_a, int cell_b, float dx, float ca, float yga, float ygb, bool ok_a,
            bool ok_b, bool has_b, auto) {
          auto take = [&](int cell, float yg, bool ok) {
            const Key k = zb[min(cell, cell_max)];
            if ((uint8_t)k == (uint8_t)j && ok) {
              const float dy = yg - s.y, q = ca - dy * dy;
              const float g = __uint_as_float((uint32_t)(k >> 32));
              const float w = g * __builtin_amdgcn_rsqf(q);
              a0 = __builtin_fmaf(-w, dx, a0);
              a1 = __builtin_fmaf(-w, dy, a1);
              a2 += g;
              a3 -= w;
            }
          };
          take(cell_a, yga, ok_a);
          if (has_b) take(cell_b, ygb, ok_b);
        },
        [&](int j) {
          // (the slot belongs to this wave: ds_add_f32 in program order, no read-back to wait for;
          // a crop of several row regions visits a sphere once per region)
          const float t = wave_sum4_transposed(a0, a1, a2, a3, lane);
          if (lane >= 60) atomicAdd(reinterpret_cast<float *>(s_part + wave * J + j) + (lane & 3), t);
          a0 = a1 = a2 = a3 = 0.f;
        });
#else
    (void)cell_max; (void)a0; (void)a1; (void)a2; (void)a3;
#endif
  }
#ifndef EXP_MSE_NO_LOAD_FENCE
  if (general) {   // (the pieces nobody used: see the convert pass -- both ways into the code below are now free of pending loads)
#pragma unroll
    for (int k = 0; k < kTgtAhead; k++) asm volatile("" : : "v"(tpre[k].x), "v"(tpre[k].y), "v"(tpre[k].z), "v"(tpre[k].w));
  }
#endif
#ifdef EXP_MSE_SKIP_TILE   // (timing experiment: the rows a box has beyond its z-buffer are dropped instead of going through the tile code)
  if (false) {
#else
  if (tile_lo < tile_hi) {
#endif
    // ---- general path (or the rows beyond the z-buffer): 32x8 tiles, owners and gradient in registers ---
    // (rows_per_region is a multiple of the tile height whenever there are several regions)
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int t0 = (tile_lo / kTileH) * tiles_x, t1 = ((tile_hi + kTileH - 1) / kTileH) * tiles_x;
    float4 *acc = s_part + wave * J;
    const float *tfull = tgt - (size_t)r0 * W;
    float *ofull = out ? out - (size_t)r0 * W : nullptr;
    for (int tile = t0 + wave; tile < t1; tile += kZWaves) {
      const TileGeom g = tile_geom(tile, tiles_x, ax, ay, lane);
      const unsigned long long mask = tile_candidates(sph, valid, g, ax, ay, H, W);
      float best[4], bsq[4];
      int owner[4];
      tile_min<true>(mask, J, sph, g, best, owner, bsq);
      const bool row_ok = g.v < H;
      const size_t base = (size_t)g.v * W + g.u0;
      float gk[4] = {0.f, 0.f, 0.f, 0.f};
      if (row_ok && g.u0 < W) {   // W % 4 == 0 on this path
        const float4 t = *reinterpret_cast<const float4 *>(tfull + base);
        const float e0 = best[0] - t.x, e1 = best[1] - t.y, e2 = best[2] - t.z, e3 = best[3] - t.w;
        sse += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        gk[0] = 2.f * e0; gk[1] = 2.f * e1; gk[2] = 2.f * e2; gk[3] = 2.f * e3;
        if (ofull) *reinterpret_cast<float4 *>(ofull + base) = make_float4(best[0], best[1], best[2], best[3]);
      }
      float px[4], py[4], pz[4], pw[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool in = row_ok && (g.u0 + k < W) && owner[k] != SHR_ARGMIN_NONE;
        if (!in) owner[k] = SHR_ARGMIN_NONE;
        const float w = in ? gk[k] / bsq[k] : 0.f;
        pz[k] = in ? gk[k] : 0.f;
        pw[k] = -w;
        const float4 o = s_sph[in ? owner[k] : 0];
        px[k] = -(w * (g.xg[k] - o.x));
        py[k] = -(w * (g.yg - o.y));
      }
      unsigned long long m = mask;
      while (m) {
        const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
        m &= m - 1;
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const bool mine = owner[k] == j;
          any |= mine;
          sx += mine ? px[k] : 0.f;
          sy += mine ? py[k] : 0.f;
          sz += mine ? pz[k] : 0.f;
          sw += mine ? pw[k] : 0.f;
        }
        if (__ballot(any) == 0) continue;
        // (one transposed four-component wave sum -- 15 instructions against 24 for four separate ones: round 6, the rows a
        // wide box leaves to this code are 4-5 % of config 5's box rows and cost four times a z-buffer row)
        const float t4 = wave_sum4_transposed(sx, sy, sz, sw, lane);
        if (lane >= 60) reinterpret_cast<float *>(acc + j)[lane & 3] += t4;
      }
    }
  }

  if (general && pf_wave && has_next) s_next[lane] = sph_next;
  // ---- reductions: waves in order --------------------------------------------------------------
  sse = wave_sum_lane63(sse);
  SHR_TL(2, 7);   // this wave's walk (and tile code) is done
  // the waves' partial sums: in the next crop's record slots where no next crop exists (not PERSIST: the box variant) -- a place
  // of their own, so ONE barrier closes the walk and publishes them; over the work list otherwise, behind a barrier of its own
  // (round 6: -0.3 to -1.2 % on the fused kernel; the header keeps its size, so no z-buffer moves -- SHR_HDR_PAD 16 .. 192
  // measured on the way: where the z-buffers start relative to the LDS banks changes nothing, +-0.3 %)
  float *s_wsum = PERSIST ? reinterpret_cast<float *>(s_items) : reinterpret_cast<float *>(s_next);
  if (PERSIST) __syncthreads();
  if (lane == 63) s_wsum[wave] = sse;
  __syncthreads();
  const size_t slot = (size_t)(slot_by_crop ? c : n) * nregions + region;
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < kZWaves; w++) t += s_wsum[w];
    sse_out[slot] = t;
  }
  if (tid < J) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < kZWaves; w++) {
      const float4 a = s_part[w * J + tid];
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    t.w = t.w * s_sph[tid].w;
    grad_out[slot * J + tid] = t;
  }
  if (has_next) __syncthreads();   // records, work list, partials and z-buffer are rewritten next
  }  // crops
}

// (argument order of the two wrappers: what a wave needs before it requests its records -- and the observed image's
// index -- comes first, among the 14 argument dwords that are preloaded into SGPRs; the outputs and the axis constants
// follow and are fetched by an s_load whose wait sits behind those requests)
template <bool POW2, bool PERSIST>
__global__ void __launch_bounds__(1024)
sphere_zbuf_mse_kernel(const float4 *__restrict__ spheres, int N, int J, int H, int W, const float *__restrict__ target,
                       const int *__restrict__ target_index, int rows_per_region, int w4_shift, int shares_fwd,
                       int shares_bwd, float *__restrict__ depth, float *__restrict__ sse_out,
                       float4 *__restrict__ grad_out, AxisK axk, const int *__restrict__ crop_index) {
  sphere_zbuf_mse_body<POW2, PERSIST, false>(spheres, N, J, H, W, target, target_index, depth, sse_out, grad_out,
                                             rows_per_region, w4_shift, shares_fwd, shares_bwd, 0, axk, crop_index);
}

// two of these per CU: eight waves per SIMD, i.e. at most 64 VGPRs -- and few enough SGPRs: left alone the kernel takes
// 93 (descriptor count) and the second workgroup does not become resident (242 us against 172 with the cap's 78 and 15
// scalar spills, 1152 crops @256x256)
template <bool POW2, bool SEG2 = false>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
sphere_zbuf_mse_box_kernel(const float4 *__restrict__ spheres, int N, int J, int H, int W, const float *__restrict__ target,
                           const int *__restrict__ target_index, int rows_per_region, int w4_shift, int zcells,
                           int shares_fwd, int shares_bwd, float *__restrict__ depth, float *__restrict__ sse_out,
                           float4 *__restrict__ grad_out, AxisK axk, const int *__restrict__ crop_index) {
  static_assert(POW2, "box variant: power-of-two images");
  sphere_zbuf_mse_body<POW2, false, true, SEG2>(spheres, N, J, H, W, target, target_index, depth, sse_out, grad_out,
                                          rows_per_region, w4_shift, shares_fwd, shares_bwd, zcells, axk, crop_index);
}

}  // namespace shr
