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
//             list of 16x4-pixel patches (lanes = spheres: pixel box, patch
//             count, prefix sum); the 16 waves then take equal contiguous slices
//             of that list (lanes = pixels): exact reference arithmetic per
//             pixel, one LDS atomic min per hit.  Finally the z-buffer is decoded
//             and streamed out with full-line 16-byte stores -- the only HBM
//             traffic besides the 16*J-byte sphere read.
//   backward  one workgroup per crop.  grad_depth and the forward's owner map are
//             staged into LDS with coalesced 16-byte loads; the waves walk the
//             same balanced patch list, accumulate the four partials of the
//             pixels a sphere owns in registers, one DPP wave-sum per (wave,
//             sphere) segment into a private LDS slot, slots combined in wave
//             order: deterministic, no float atomics.
//
// Exactness: the per-pixel arithmetic is the reference's operation sequence
// (common.h, -ffp-contract=off; sqrt_rn() is a correctly rounded square root).
// Integer-key minima are exact.  The fast forward requires, per crop, all sphere
// parameters finite, |x|,|y|,|r| < 1e6 and at least one sphere with z <= 100
// (then initialising the z-buffer to the background is the reference's min: see
// the kernel); any other crop takes the general tile kernel (sphere_tile.h)
// inside the same workgroup.
#pragma once
#include "sphere_tile.h"

namespace shr {

constexpr int kZWaves = 16;   // 1024 threads
constexpr int kPatchW = 16;   // lanes along x
constexpr int kPatchH = 4;    // lanes along y
constexpr int kRowPad = 16;   // LDS row padding (elements): patches may overhang the image edge
constexpr int kPadRows = kPatchH - 1;  // ... and the region's last row
// LDS header: spheres [64] float4 | work items [64] int4 | ends [64] int | flags
constexpr int kOffItems = 1024;
constexpr int kOffEnds = 2048;
constexpr int kOffFlags = 2304;
constexpr int kHdrBytes = 2304 + 16;
constexpr int kMaxFastWidth = 8192;  // 16-bit fields of the work items

__device__ __forceinline__ uint32_t depth_key(float d) {
  const uint32_t b = __float_as_uint(d);
  return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_depth(uint32_t k) {
  return __uint_as_float(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}

// Correctly rounded sqrt for a normal, positive, finite fp32 argument (here
// q > 0.01): the hardware estimate (v_sqrt_f32, <= 1 ulp) corrected by the exact
// residuals of its two neighbours -- the same correction hipcc's sqrtf() applies,
// without its denormal-scaling prologue.  Checked against sqrtf() on every fp32
// value in [0.01, 1e12] (tests/test_sphere_raster_gpu.py::test_sqrt_rn_exhaustive).
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = __uint_as_float(__float_as_uint(s) - 1u);
  const float up = __uint_as_float(__float_as_uint(s) + 1u);
  const float e_dn = __builtin_fmaf(-dn, s, x);
  const float e_up = __builtin_fmaf(-up, s, x);
  float r = (e_dn <= 0.0f) ? dn : s;
  r = (e_up > 0.0f) ? up : r;
  return r;
}

__device__ __forceinline__ bool sphere_is_tame(const float4 s) {
  return fabsf(s.x) < 1e6f && fabsf(s.y) < 1e6f && fabsf(s.w) < 1e6f;  // false for NaN/Inf
}

// One work item = one sphere's pixel box clipped to the region, as a grid of
// 16x4-pixel patches.  A hit needs |fl(xg - x)| <= |r| (sphere_tile.h), i.e. the
// pixel centre inside [x - |r|, x + |r|] up to one rounding.  With
// xg(u) = (u - half)*300/size  <=>  u = xg*size/300 + half the box is
//   u in [ceil(ulo - eps), floor(uhi + eps)],
// eps = 1e-3 px + 2e-6 relative: >> the fp32 error of the inverse map (<= 1e-5 px
// near the image, 6e-8 relative far away), << a pixel, so the box is the tight
// conservative one.  Non-tame spheres (NaN/Inf/huge) take the whole region.
struct Item { int u0, v0, npx, npy; };

__device__ __forceinline__ void axis_box(float c, float ar, float k, float half, float hi_clamp, int lo_lim,
                                         int hi_lim, int &i0, int &i1) {
  const float lo = (c - ar) * k + half, hi = (c + ar) * k + half;
  const float eps = 1e-3f + 2e-6f * (fabsf(lo) + fabsf(hi));
  i0 = max((int)ceilf(fminf(fmaxf(lo - eps, -2.f), hi_clamp)), lo_lim);
  i1 = min((int)floorf(fminf(fmaxf(hi + eps, -2.f), hi_clamp)), hi_lim);
}

__device__ __forceinline__ Item sphere_item(const float4 s, const Axis &ax, const Axis &ay, int W, int r0,
                                            int r1) {
  int u0 = 0, u1 = W - 1, v0 = r0, v1 = r1 - 1;
  if (sphere_is_tame(s)) {
    const float ar = fabsf(s.w);
    axis_box(s.x, ar, ax.size / 300.0f, ax.half, (float)W + 2.f, 0, W - 1, u0, u1);
    axis_box(s.y, ar, ay.size / 300.0f, ay.half, (float)r1 + 2.f, r0, r1 - 1, v0, v1);
  }
  Item it;
  it.u0 = u0;
  it.v0 = v0;
  it.npx = (u1 >= u0) ? ((u1 - u0) / kPatchW + 1) : 0;
  it.npy = (v1 >= v0) ? ((v1 - v0) / kPatchH + 1) : 0;
  return it;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rfl(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// The work list is a sequence of PATCH ROWS (npx patches side by side), sphere after
// sphere, each row weighted by its estimated cost (row overhead + per-patch work) so
// that equal slices of the cumulative weight are equal work.
// Measured on MI355X (in-kernel clock64, one crop per CU): starting a sphere ~1300
// cycles (broadcasts, column terms), a patch row ~350, each patch ~150.
constexpr int kSphereCost = 26;
constexpr int kRowCost = 7;
constexpr int kPatchCost = 3;

// Wave 0: build the list in LDS.  s_items[j] = (u0 | v0<<16, npx | nrows<<16, row
// weight, weight prefix before sphere j); s_ends[j] = prefix after it; a sphere weighs
// kSphereCost + nrows * row weight.  Returns the total weight (valid in every lane of
// wave 0).
__device__ __forceinline__ int build_work_list(const float4 s, bool valid, const Axis &ax, const Axis &ay,
                                               int W, int r0, int r1, int4 *s_items, int *s_ends, int lane,
                                               bool *too_big) {
  const Item it = sphere_item(s, ax, ay, W, r0, r1);
  const int nrows = (valid && it.npx > 0) ? it.npy : 0;
  const int wt = kRowCost + kPatchCost * it.npx;
  const int cost = nrows > 0 ? kSphereCost + nrows * wt : 0;
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
  s_items[lane] = make_int4(it.u0 | (it.v0 << 16), it.npx | (nrows << 16), wt, incl - cost);
  s_ends[lane] = incl;
  // fields are 16-bit; the launcher keeps W <= kMaxFastWidth, H <= 32768
  *too_big = __ballot(valid && (it.u0 > 65535 || it.v0 > 65535 || it.npx > 65535 || nrows > 32767)) != 0ull;
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

template <bool POW2>
__device__ __forceinline__ float axis_coord_t(const Axis &a, int u);

// Walk the patch rows whose weight position lies in [lo, hi) (wave-uniform), sphere by
// sphere.  Per sphere the column terms c = r*r - dx*dx of the first two patch columns
// are hoisted out of the row loop and dy*dy out of the column loop, so a patch costs
// two subtractions and a compare before `body(j, s, ua, ub, v, qa, qb, has_b)` -- two
// side-by-side patches per call, q = (r*r - dx*dx) - dy*dy in the reference's
// association.  `end_sphere(j)` closes a run on sphere j.
template <bool POW2, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_slice(const WaveList &w, int J, int lo, int hi, int lane, const Axis &ax,
                                           const Axis &ay, Body &&body, EndSphere &&end_sphere) {
  const int lx = lane & (kPatchW - 1), ly = lane >> 4;
  int j = __popcll(__ballot(lane < J && w.end <= lo));   // prefixes are non-decreasing
  while (j < J) {
    const int wstart = rl(w.item.w, j);
    if (wstart >= hi) break;
    const int shape = rl(w.item.y, j), wt = rl(w.item.z, j);
    const int npx = shape & 0xffff, nrows = shape >> 16;
    // first row whose position wstart + kSphereCost + r*wt is at or after lo (a few
    // scalar steps; an integer division would cost more than the rows it skips)
    int r = 0, p = wstart + kSphereCost;
    while (p < lo && r < nrows) { ++r; p += wt; }
    if (r < nrows && p < hi) {
      const int geom = rl(w.item.x, j);
      const float4 s = make_float4(readlane_f(w.sph.x, j), readlane_f(w.sph.y, j), readlane_f(w.sph.z, j),
                                   readlane_f(w.sph.w, j));
      const float rr = s.w * s.w;
      const int ua = (geom & 0xffff) + lx, ub = ua + kPatchW;
      const int v0 = (int)((unsigned)geom >> 16) + ly;
      const float dxa = axis_coord_t<POW2>(ax, ua) - s.x, dxb = axis_coord_t<POW2>(ax, ub) - s.x;
      const float ca = rr - dxa * dxa, cb = rr - dxb * dxb;
      for (; r < nrows && p < hi; ++r, p += wt) {
        const int v = v0 + r * kPatchH;
        const float dy = axis_coord_t<POW2>(ay, v) - s.y;
        const float dy2 = dy * dy;
        // the first two columns go to the body together: two independent chains the
        // scheduler can interleave (a wave runs one dependent instruction stream)
        body(j, s, ua, ub, v, ca - dy2, cb - dy2, npx >= 2);
        for (int px = 2; px < npx; px += 2) {
          const int u = ua + px * kPatchW;
          const float dx0 = axis_coord_t<POW2>(ax, u) - s.x, dx1 = axis_coord_t<POW2>(ax, u + kPatchW) - s.x;
          body(j, s, u, u + kPatchW, v, (rr - dx0 * dx0) - dy2, (rr - dx1 * dx1) - dy2, px + 1 < npx);
        }
      }
      end_sphere(j);
    }
    ++j;
  }
}

// wave w of nwaves takes the w-th equal slice of the total weight
template <bool POW2, typename Body, typename EndSphere>
__device__ __forceinline__ void walk_my_slice(const WaveList &w, int J, int total, int wave, int nwaves, int lane,
                                              const Axis &ax, const Axis &ay, Body &&body, EndSphere &&end_sphere) {
  wave = rfl(wave);   // everything that steers the loops is wave-uniform: keep it in SGPRs
  total = rfl(total);
  const int lo = (int)(((long long)wave * total) / nwaves), hi = (int)(((long long)(wave + 1) * total) / nwaves);
  if (lo < hi) walk_slice<POW2>(w, J, lo, hi, lane, ax, ay, body, end_sphere);
}

// image axis coordinate with the power-of-two case resolved at compile time
template <bool POW2>
__device__ __forceinline__ float axis_coord_t(const Axis &a, int u) {
  const float t = (float)u - a.half;
  return POW2 ? t * a.mul : (t * 300.0f) / a.size;
}

template <bool OWNER> struct KeyOf { using type = uint32_t; };
template <> struct KeyOf<true> { using type = unsigned long long; };

// ---------------------------------------------------------------------------
// Forward.  grid = (N, nregions), block = 64 * nwaves (<= 1024), dynamic LDS = kHdrBytes +
// (rows_per_region + kPadRows) * (W + kRowPad) * sizeof(key).
template <bool OWNER, bool VEC4, bool POW2>
__global__ void __launch_bounds__(1024)
sphere_zbuf_fwd_kernel(const float4 *__restrict__ spheres, int J, int H, int W,
                       float *__restrict__ depth, uint8_t *__restrict__ argmin, int rows_per_region,
                       int w4_shift) {
  using Key = typename KeyOf<OWNER>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_sph = reinterpret_cast<float4 *>(smem);
  int4 *s_items = reinterpret_cast<int4 *>(smem + kOffItems);
  int *s_ends = reinterpret_cast<int *>(smem + kOffEnds);
  int *s_flag = reinterpret_cast<int *>(smem + kOffFlags);
  Key *zbuf = reinterpret_cast<Key *>(smem + kHdrBytes);

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwaves = nthr >> 6;
  const int r0 = blockIdx.y * rows_per_region;
  const int r1 = min(H, r0 + rows_per_region);
  const int rh = r1 - r0;
  const int LW = W + kRowPad;
  const Axis ax = make_axis(W), ay = make_axis(H);

  if (wave == 0) {
    const bool valid = lane < J;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) s = spheres[(size_t)n * J + lane];
    s_sph[lane] = s;
    // general path unless every sphere is tame and at least one has z <= 100: a pixel's
    // minimum can exceed the background only where ALL J spheres hit it, and there the
    // sphere with z <= 100 contributes z - sqrt(q) < 100, so min(100, hits) is exact.
    const unsigned long long bad = __ballot(valid && !(sphere_is_tame(s) && fabsf(s.z) < 1e30f));
    const unsigned long long low = __ballot(valid && s.z <= kBackground);
    bool too_big;
    const int total = build_work_list(s, valid, ax, ay, W, r0, r1, s_items, s_ends, lane, &too_big);
    if (lane == 0) {
      s_flag[0] = (bad != 0ull) || (low == 0ull) || too_big;
      s_flag[1] = total;
    }
  }
  {  // background everywhere (pad rows/columns included)
    const Key bg = OWNER ? (Key)(((unsigned long long)depth_key(kBackground) << 32) | SHR_ARGMIN_NONE)
                         : (Key)depth_key(kBackground);
    constexpr int per16 = 16 / sizeof(Key);
    const int ncell = (rh + kPadRows) * LW;
    const int nvec = ncell / per16;
    if (OWNER) {
      const ulonglong2 v = make_ulonglong2(bg, bg);
      for (int i = tid; i < nvec; i += nthr) reinterpret_cast<ulonglong2 *>(zbuf)[i] = v;
    } else {
      const uint4 v = make_uint4((uint32_t)bg, (uint32_t)bg, (uint32_t)bg, (uint32_t)bg);
      for (int i = tid; i < nvec; i += nthr) reinterpret_cast<uint4 *>(zbuf)[i] = v;
    }
    for (int i = nvec * per16 + tid; i < ncell; i += nthr) zbuf[i] = bg;
  }
  __syncthreads();

  float *out = depth + (size_t)n * H * W;
  uint8_t *aout = OWNER ? argmin + (size_t)n * H * W : nullptr;

  if (s_flag[0]) {  // workgroup-uniform: this crop needs the general path
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int t0 = (r0 / kTileH) * tiles_x, t1 = ((r1 + kTileH - 1) / kTileH) * tiles_x;
    const float4 sph = lane < J ? s_sph[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    tile_forward<VEC4, OWNER>(sph, J, H, W, out, aout, tiles_x, t0 + wave, t1, nwaves, lane);
    return;
  }

  // ---- scan-convert the patch list -------------------------------------------------
  // A patch may overhang the box, the image's right edge or the region's last row:
  // the hit test is exact for ANY pixel, overhanging lanes land in LDS padding.
  {
    const WaveList wl = load_wave_list(s_sph, s_items, s_ends, lane);
    walk_my_slice<POW2>(
        wl, J, s_flag[1], wave, nwaves, lane, ax, ay,
        [&](int j, const float4 s, int ua, int ub, int v, float qa, float qb, bool has_b) {
          Key *row = zbuf + (v - r0) * LW;
          auto put = [&](Key *cell, float d) {
            if (OWNER)
              atomicMin(reinterpret_cast<unsigned long long *>(cell),
                        ((unsigned long long)depth_key(d) << 32) | (unsigned)j);
            else
              atomicMin(reinterpret_cast<unsigned int *>(cell), depth_key(d));
          };
          if (has_b) {  // wave-uniform; branch-free up to the atomics: both roots in flight together
            const bool ha = qa > kHitMin, hb = qb > kHitMin;
            // (the root of a non-hit lane's q may be NaN: never stored)
            const float da = s.z - sqrt_rn(qa), db = s.z - sqrt_rn(qb);
            if (ha) put(row + ua, da);
            if (hb) put(row + ub, db);
          } else if (qa > kHitMin) {
            put(row + ua, s.z - sqrt_rn(qa));
          }
        },
        [](int) {});
  }
  __syncthreads();

  // ---- stream the region out ---------------------------------------------------------
  if (VEC4) {
    const int w4 = W >> 2;
    const int nchunk = rh * w4;
    for (int c = tid; c < nchunk; c += nthr) {
      int v, u;
      if (w4_shift >= 0) { v = c >> w4_shift; u = (c & (w4 - 1)) << 2; }
      else { v = c / w4; u = (c - v * w4) << 2; }
      const Key *cell = zbuf + v * LW + u;
      float4 o;
      if (OWNER) {
        const ulonglong2 k01 = reinterpret_cast<const ulonglong2 *>(cell)[0];
        const ulonglong2 k23 = reinterpret_cast<const ulonglong2 *>(cell)[1];
        o = make_float4(key_depth((uint32_t)(k01.x >> 32)), key_depth((uint32_t)(k01.y >> 32)),
                        key_depth((uint32_t)(k23.x >> 32)), key_depth((uint32_t)(k23.y >> 32)));
        *reinterpret_cast<uchar4 *>(aout + (size_t)(r0 + v) * W + u) =
            make_uchar4((uint8_t)k01.x, (uint8_t)k01.y, (uint8_t)k23.x, (uint8_t)k23.y);
      } else {
        const uint4 k = *reinterpret_cast<const uint4 *>(cell);
        o = make_float4(key_depth(k.x), key_depth(k.y), key_depth(k.z), key_depth(k.w));
      }
      *reinterpret_cast<float4 *>(out + (size_t)(r0 + v) * W + u) = o;
    }
  } else {
    for (int p = tid; p < rh * W; p += nthr) {
      const int v = p / W, u = p - v * W;
      const Key k = zbuf[v * LW + u];
      if (OWNER) {
        out[(size_t)(r0 + v) * W + u] = key_depth((uint32_t)((unsigned long long)k >> 32));
        aout[(size_t)(r0 + v) * W + u] = (uint8_t)k;
      } else {
        out[(size_t)(r0 + v) * W + u] = key_depth((uint32_t)k);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Backward with the forward's owner map.  grid = (N), block = 1024, dynamic LDS =
// kHdrBytes + 16*64*16 (wave x sphere partial sums) + (rows + kPadRows) * (W +
// kRowPad) * (4 + 1).
constexpr int kPartBytes = kZWaves * SHR_MAX_SPHERES * 16;

template <bool VEC4, bool POW2>
__global__ void __launch_bounds__(1024)
sphere_zbuf_bwd_kernel(const float4 *__restrict__ spheres, const float *__restrict__ grad_depth,
                       const uint8_t *__restrict__ argmin, int J, int H, int W,
                       float4 *__restrict__ grad_spheres, int rows_per_region, int w4_shift) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_sph = reinterpret_cast<float4 *>(smem);
  int4 *s_items = reinterpret_cast<int4 *>(smem + kOffItems);
  int *s_ends = reinterpret_cast<int *>(smem + kOffEnds);
  int *s_flag = reinterpret_cast<int *>(smem + kOffFlags);
  float4 *s_part = reinterpret_cast<float4 *>(smem + kHdrBytes);
  const int LW = W + kRowPad;
  float *gbuf = reinterpret_cast<float *>(smem + kHdrBytes + kPartBytes);
  uint8_t *obuf = smem + kHdrBytes + kPartBytes + (size_t)(rows_per_region + kPadRows) * LW * 4;

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *gin = grad_depth + (size_t)n * H * W;
  const uint8_t *oin = argmin + (size_t)n * H * W;
  const Axis ax = make_axis(W), ay = make_axis(H);
  float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wave == 0) {
    if (lane < J) sph = spheres[(size_t)n * J + lane];
    s_sph[lane] = sph;
  }
  s_part[tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // 1024 = 16 waves x 64 spheres

  for (int r0 = 0; r0 < H; r0 += rows_per_region) {
    const int r1 = min(H, r0 + rows_per_region), rh = r1 - r0;
    if (r0 > 0) __syncthreads();  // the previous region's walk is done
    if (wave == 0) {
      bool too_big;   // excluded by the launcher (W <= kMaxFastWidth)
      const int total = build_work_list(sph, lane < J, ax, ay, W, r0, r1, s_items, s_ends, lane, &too_big);
      if (lane == 0) s_flag[1] = total;
    }
    // owner padding = "nobody": the walk may overhang the image edge / region end
    for (int i = tid; i < rh * kRowPad; i += 1024) obuf[(i / kRowPad) * LW + W + (i % kRowPad)] = SHR_ARGMIN_NONE;
    for (int i = tid; i < kPadRows * LW; i += 1024) obuf[rh * LW + i] = SHR_ARGMIN_NONE;
    if (VEC4) {
      const int w4 = W >> 2;
      const int nchunk = rh * w4;
      for (int c = tid; c < nchunk; c += 1024) {
        int v, u;
        if (w4_shift >= 0) { v = c >> w4_shift; u = (c & (w4 - 1)) << 2; }
        else { v = c / w4; u = (c - v * w4) << 2; }
        const size_t src = (size_t)(r0 + v) * W + u;
        *reinterpret_cast<float4 *>(gbuf + v * LW + u) = *reinterpret_cast<const float4 *>(gin + src);
        *reinterpret_cast<uchar4 *>(obuf + v * LW + u) = *reinterpret_cast<const uchar4 *>(oin + src);
      }
    } else {
      for (int p = tid; p < rh * W; p += 1024) {
        const int v = p / W, u = p - v * W;
        gbuf[v * LW + u] = gin[(size_t)(r0 + v) * W + u];
        obuf[v * LW + u] = oin[(size_t)(r0 + v) * W + u];
      }
    }
    __syncthreads();

    // Static schedule: wave w walks the w-th of 16 equal-weight contiguous slices of the
    // list (which wave sums which pixels must not depend on timing); at the end of a
    // run on a sphere its register partials are reduced with one DPP wave sum per
    // component into the wave's private LDS slot.
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const WaveList wl = load_wave_list(s_sph, s_items, s_ends, lane);
    walk_my_slice<POW2>(
        wl, J, s_flag[1], wave, kZWaves, lane, ax, ay,
        [&](int j, const float4 s, int ua, int ub, int v, float qa, float qb, bool has_b) {
          const int row = (v - r0) * LW;
          auto take = [&](int u, float q) {
            if (obuf[row + u] == (uint8_t)j) {
              const float g = gbuf[row + u];
              const float dx = axis_coord_t<POW2>(ax, u) - s.x;
              const float dy = axis_coord_t<POW2>(ay, v) - s.y;
              const float w = g * __builtin_amdgcn_rsqf(q);  // g / sqrt(q), ~1e-7 rel.
              a0 = __builtin_fmaf(-w, dx, a0);
              a1 = __builtin_fmaf(-w, dy, a1);
              a2 += g;
              a3 -= w;
            }
          };
          take(ua, qa);
          if (has_b) take(ub, qb);
        },
        [&](int j) {
          const float sx = wave_sum_lane63(a0), sy = wave_sum_lane63(a1);
          const float sz = wave_sum_lane63(a2), sw = wave_sum_lane63(a3);
          if (lane == 63) {
            float4 t = s_part[wave * SHR_MAX_SPHERES + j];
            t.x += sx; t.y += sy; t.z += sz; t.w += sw;
            s_part[wave * SHR_MAX_SPHERES + j] = t;
          }
          a0 = a1 = a2 = a3 = 0.f;
        });
  }
  __syncthreads();
  // combine the waves' partials in wave order; d/dr = r * sum(-g/sqrt(q))
  if (tid < J) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < kZWaves; w++) {
      const float4 a = s_part[w * SHR_MAX_SPHERES + tid];
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    t.w = t.w * s_sph[tid].w;
    grad_spheres[(size_t)n * J + tid] = t;
  }
}

}  // namespace shr
