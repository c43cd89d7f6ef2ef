// data_to_model.hip -- data-to-model loss: observed depth pixels vs the sphere set.
//
// Replaces (reference file:line): mesh/render.py:123-142 DataToModelLoss.forward
// and its autograd backward.  Per pixel p = (xg, yg, depth) with depth <= 99:
//     e = min_j | ||p - c_j||_2 - r_j | ,  clamp(e, 0, 50)
// loss_sum[n] = sum over the crop's pixels (background contributes 0; the
// reference's scalar is sum_n loss_sum[n] / (N*H*W), mesh/render.py:142).
// The loss is a plain sum, so its gradient w.r.t. the centres does not depend on
// the upstream value: the same pass also emits
//     grad_centres[n,j,:] = d loss_sum[n] / d c_j
//                         = sum over pixels owned by j with e <= 50 of
//                           -sign(dist - r_j) * (p - c_j) / dist
// and the caller scales it by upstream / (N*H*W).
//
// Design (round 2; the round-1 kernel pruned per 64 points with wave-wide boxes and seeds and was
// bound by the latency of those reductions and of one-point-per-lane dependent chains):
//   * every WAVE is independent until the final combine.  The crop is cut into UNITS of 256
//     consecutive pixels (one 16-byte load per lane) and BANDS of a few consecutive units, handed out
//     to the waves DYNAMICALLY (an LDS counter; legal because the sums below do not depend on their
//     order): every wave carries an equal share of the hand while the points it searches together
//     stay neighbours.  Two units' loads are in flight per wave, across band boundaries;
//   * a wave keeps the ~15 % foreground pixels of a unit and appends them -- lane order,
//     ballot/mbcnt prefix -- to its own ring in LDS as 8-byte (v << 16 | u, z) entries; whenever
//     256 entries are there (and at the end of the band) it searches them, FOUR neighbouring points
//     per lane: a sphere's record is read once for the 256 points (uniform-address ds_read_b128 =
//     an LDS broadcast, requested one sphere ahead), 12 VALU instructions per (point, sphere), four independent chains per lane;
//   * the points of a search lie in a thin strip of rows [y_lo, y_hi] (first / last ring entry:
//     256 points are ~5 rows of a hand), and | ||p - c|| - r | >= dist_y(c, strip) - r, so with
//     lanes = spheres one ballot gives the spheres whose y extent meets the strip.  They are
//     searched first; the largest of the running minima (one wave maximum per search) then bounds
//     what any other sphere would have to beat, and only spheres whose gap is below it are
//     searched as well -- exact (a pruned sphere can neither win nor tie);
//   * the loss and the gradient are accumulated as FIXED-POINT integers (2^-20 mm / 2^-26 per
//     unit-vector component; 64-bit LDS atomics on one table per workgroup, a lane's four points
//     combined first when they share their owner): integer sums do not depend on their order, so
//     the result is bit-reproducible AND (one workgroup per crop) independent of the launch shape,
//     without any per-owner wave reduction (the round-1 ballot loop over distinct owners was a third of the kernel).
// A non-finite sphere record or depth value sends the search through the exact index-order loop
// (torch.min / clamp propagate NaN).  HBM: reads 4*H*W + 12*J + 4*J bytes per crop.

#include <type_traits>

#include "common.h"
#include "d2m_search.h"

namespace shr {

constexpr int kD2mK = 4;                    // points per lane per search
constexpr int kD2mGroup = 64 * kD2mK;       // entries per full search
constexpr int kD2mCap = 512;                // ring entries per wave (< 256 left over + <= 256 new per unit)
// The ring is stored TRANSPOSED: entry e sits in row e & 3, column (e >> 2) & 127 of a [4][128 + pad] array.  A search
// reads entries head + 4 l + i for lane l: row (head + i) & 3, columns consecutive in l -- one conflict-free
// ds_read_b64 per point, where the linear layout put a lane's four entries 32 bytes from the next lane's (every
// eighth lane on the same banks).  The append writes entries consecutive in pixel order: the four of a dense lane
// go to the four rows at one column, the rows' pitch shifts them by 16 banks each.
constexpr int kD2mTableStride = SHR_MAX_SPHERES * 4 + 2;   // u64 per copy (+ 16 bytes)
constexpr int kD2mRingCols = kD2mCap / 4;
constexpr int kD2mRingPitch = kD2mRingCols + 8;   // uint2 per row (+ 64 bytes: rows start 16 banks apart)
__device__ __forceinline__ int d2m_slot(int e) { return (e & 3) * kD2mRingPitch + ((e >> 2) & (kD2mRingCols - 1)); }

// A wave's position in its sequence of units: band `band` = units band * band_units .. (clipped).  Bands are handed
// out DYNAMICALLY inside the workgroup (an LDS counter): the sums are order-independent integers, so which wave
// takes which band changes nothing in the result, and no wave idles at the final barrier while a sibling still
// has rows of the hand to search.
struct D2mUnitIter {
  int band, unit, unit_end;
  bool done;
  int tx, ty, dty;   // TILED: the unit's tile (column, row) and the direction of the walk
  __device__ void seek(int b, int nbands, int band_units, int units, int parts = 1, int part = 0) {
    band = b;
    done = b * parts + part >= nbands;
    unit = (b * parts + part) * band_units;
    unit_end = min(units, unit + band_units);
  }
  // TILED: band gb = (tile column tx, segment of band_units tile rows), the segments of a column visited downwards in
  // even columns and upwards in odd ones (consecutive bands are neighbours), the tile rows of a segment likewise
  __device__ void seek_tiled(int b, int nbands, int band_units, int nseg, int tiles_y, int parts, int part) {
    band = b;
    const int gb = b * parts + part;
    done = gb >= nbands;
    tx = gb / nseg;
    const int sidx = gb - tx * nseg, seg = (tx & 1) ? nseg - 1 - sidx : sidx;
    const int cnt = min(band_units, tiles_y - seg * band_units);
    unit = 0;
    unit_end = cnt;
    dty = (tx & 1) ? -1 : 1;
    ty = (tx & 1) ? seg * band_units + cnt - 1 : seg * band_units;
  }
};

// TILED (round 4; needs W % 4 == 0 and a 16-byte aligned image): a unit is a TILE of 32 x 8 pixels -- lane l holds
// the four pixels from column 4 (l & 7) of row l >> 3, eight lanes = one full 128-byte line per row -- instead of 256
// consecutive pixels, the units of a band walk down (or up) a tile column, and a search is bounded by its points' own
// x-y bounding box (d2m_search.h BOX2D): a group of 256 points is then two or three neighbouring tiles, ~40 x 25 mm at
// 256 x 256, where the pixel-order groups are strips across the whole hand whose only usable bound is their y extent:
// ~8 instead of ~19 of the 41 spheres evaluated per point -- and those evaluations are the kernel (80 us of VALU issue
// on 1024 SIMDs at config 5's size against 63 us of streaming).  Groups may span two bands (the box is the points'
// own, whatever they are): no remainder search at a band's end, only one at the wave's.  The bound is conservative
// either way: every point keeps its minimum, its first index and its fixed-point terms -- bit-identical sums.
template <bool WANT_GRAD, int WAVES, bool TILED = false>
__global__ void __launch_bounds__(WAVES * 64)
data_to_model_kernel(const float *__restrict__ depth, const int *__restrict__ depth_index,
                     const float *__restrict__ centres, int centre_stride, const float *__restrict__ radii, int J,
                     int H, int W, int band_units, int parts, float *__restrict__ loss_sum,
                     float *__restrict__ grad_centres) {
  __shared__ float4 s_c[SHR_MAX_SPHERES];                 // (cx, cy, cz, r)
  __shared__ int s_odd, s_nan;                            // non-finite sphere table / a NaN loss term
  __shared__ unsigned long long s_loss;                   // fixed-point loss sum
  // fixed-point gradient rows [table][sphere][x,y,z,-]: kD2mTables copies, a lane adds into copy lane & (kD2mTables - 1).
  // Neighbouring lanes hold neighbouring pixels -- mostly one owner -- and a 64-bit LDS atomic is serialised over the
  // lanes that share its ADDRESS (SQ_LDS_ADDR_CONFLICT was 3/4 of the kernel's bank-conflict cycles with one table);
  // the copies start 4 banks apart.  Integer sums: adding the copies at the end changes nothing in the result.
  __shared__ unsigned long long s_acc[WANT_GRAD ? kD2mTables * kD2mTableStride : 1];
  __shared__ uint2 s_q[WAVES][4 * kD2mRingPitch];         // per-wave rings of foreground pixels (transposed: d2m_slot)
  __shared__ int s_next_band;                             // bands are handed out dynamically (see below)
  __shared__ int s_bandq[WAVES][8];                       // per wave: bands its loads have entered, not yet processed

  const int n = blockIdx.x / parts, part = blockIdx.x - n * parts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < J) {
      const float *p = centres + ((size_t)n * J + lane) * centre_stride;
      c = make_float4(p[0], p[1], p[2], radii[lane]);
      s_c[lane] = c;
    }
    const float inf = __builtin_inff();
    const bool bad = !(fabsf(c.x) < inf) || !(fabsf(c.y) < inf) || !(fabsf(c.z) < inf) || !(fabsf(c.w) < inf);
    const bool any = __ballot(bad) != 0ull;
    if (lane == 0) { s_odd = any; s_nan = 0; s_loss = 0ull; s_next_band = WAVES; }
  }
  if (WANT_GRAD)
    for (int i = tid; i < kD2mTables * kD2mTableStride; i += WAVES * 64) s_acc[i] = 0ull;
  __syncthreads();
  const bool table_odd = s_odd != 0;
  // lanes = spheres: this lane's record for the box bounds
  const float4 cj = lane < J ? s_c[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned long long all = J >= 64 ? ~0ull : ((1ull << J) - 1ull);

  const float *dm = depth + (size_t)(depth_index ? depth_index[n] : n) * H * W;
  const Axis ax = make_axis(W), ay = make_axis(H);
  const bool vec4 = (W % 4 == 0) && is_aligned16(dm);
  const int P = H * W;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + 7) >> 3, nseg = (tiles_y + band_units - 1) / band_units;
  const int units = (P + 255) >> 8, nbands = TILED ? tiles_x * nseg : (units + band_units - 1) / band_units;
  const int wshift = (W & (W - 1)) == 0 ? __builtin_ctz(W) : -1;
  uint2 *ring = s_q[wave];
  long long loss_fx = 0;

  // ---- the search over `count` (<= 64 K) ring entries starting at `head` (d2m_search.h, shared with the fused
  // render-and-compare kernel) ----------------------------------------------------------------------
  D2mCtx ctx;
  ctx.s_c = s_c; ctx.cj = cj; ctx.all = all; ctx.table_odd = table_odd; ctx.J = J; ctx.lane = lane;
  ctx.ax = ax; ctx.ay = ay; ctx.s_acc = s_acc; ctx.acc_stride = kD2mTableStride; ctx.s_nan = &s_nan;
  auto search = [&](auto kc, int head, int count) {
    constexpr int K = decltype(kc)::value;
    d2m_search<K, WANT_GRAD, TILED>(ctx, [&](int idx) { return ring[d2m_slot(head + idx)]; }, count, loss_fx);
  };

  // ---- this wave's units: three loads in flight, compact, search ------------------------------
  auto load_unit = [&](const D2mUnitIter &it, float z[4]) {
    z[0] = z[1] = z[2] = z[3] = 100.f;
    if (it.done) return;
    if (TILED) {
      const int v = (it.ty << 3) + (lane >> 3), u = (it.tx << 5) + ((lane & 7) << 2);
      if (v < H && u < W) {
        const float4 t = *reinterpret_cast<const float4 *>(dm + (size_t)v * W + u);
        z[0] = t.x; z[1] = t.y; z[2] = t.z; z[3] = t.w;
      }
      return;
    }
    const int p = it.unit * 256 + lane * 4;
    if (vec4) {
      if (p < P) {
        const float4 t = *reinterpret_cast<const float4 *>(dm + p);
        z[0] = t.x; z[1] = t.y; z[2] = t.z; z[3] = t.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (p + c < P) z[c] = dm[p + c];
    }
  };
  auto seek = [&](D2mUnitIter &it, int b) {
    if (TILED) it.seek_tiled(b, nbands, band_units, nseg, tiles_y, parts, part);
    else it.seek(b, nbands, band_units, units, parts, part);
  };
  auto step = [&](D2mUnitIter &it) {   // to the band's next unit; true at the band's end
    if (TILED) it.ty += it.dty;
    return ++it.unit >= it.unit_end;
  };
  // the LOAD iterator draws the bands (it runs three units ahead and may be up to three bands ahead: the drawn
  // bands wait in a 4-entry queue), the PROCESS iterator follows the same sequence
  int q_put = 0, q_get = 0;
  auto advance_load = [&](D2mUnitIter &it) {
    if (it.done) return;
    if (step(it)) {
      int b = 0;
      if (lane == 0) b = atomicAdd(&s_next_band, 1);
      b = __builtin_amdgcn_readfirstlane(b);
      if (lane == 0) s_bandq[wave][q_put & 7] = b;
      q_put++;
      seek(it, b);
    }
  };
  auto advance_proc = [&](D2mUnitIter &it) {
    if (it.done) return;
    if (step(it)) {
      const int b = __builtin_amdgcn_readfirstlane(s_bandq[wave][q_get & 7]);
      q_get++;
      seek(it, b);
    }
  };
  D2mUnitIter itp, itl;
  seek(itp, wave);
  itl = itp;
#ifndef D2M_DEPTH
#define D2M_DEPTH 2      // units in flight per wave (2: 55 us, 3: 59, 4: 58, 6: 69 for 1152 crops @128^2 -- registers)
#endif
  float zq[D2M_DEPTH][4];
#pragma unroll
  for (int d = 0; d < D2M_DEPTH; d++) { load_unit(itl, zq[d]); advance_load(itl); }
  float (&z0)[4] = zq[0];
  int head = 0, tail = 0;                         // ring positions (monotonic; masked on use)
  while (!itp.done) {
    const int p0 = itp.unit * 256 + lane * 4;
    const int tv = (itp.ty << 3) + (lane >> 3), tu = (itp.tx << 5) + ((lane & 7) << 2);   // TILED: this lane's pixels
    const bool tin = tv < H && tu < W;
    // foreground flags (mesh/render.py:138: background = d > 99), exclusive prefix in pixel order
    bool fg[4];
    int before = 0, total = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      fg[c] = (TILED ? tin : (p0 + c < P)) && !(z0[c] > 99.0f);
      const unsigned long long m = __ballot(fg[c]);
      before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, before));
      total += __builtin_popcountll(m);
    }
    if (total) {
      // `before` = flagged pixels of LOWER lanes over all four components: this lane's entries start there
      // (a lane's four pixels are consecutive: pixel order)
      int pos = tail + before;
      int vc, uc;
      if (TILED) { vc = tv; uc = tu; }
      else if (wshift >= 0) { vc = p0 >> wshift; uc = p0 & (W - 1); }
      else { vc = p0 / W; uc = p0 - vc * W; }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (!TILED) while (uc >= W) { uc -= W; vc++; }      // never taken when W % 4 == 0
        if (fg[c]) {
          ring[d2m_slot(pos)] = make_uint2(((unsigned)vc << 16) | (unsigned)uc, __float_as_uint(z0[c]));
          pos++;
        }
        uc++;
      }
      tail += total;
    }
    // pixel order: a search never spans two bands (its strip bound wants consecutive rows), what a band leaves is
    // searched at its end.  TILED: only full groups here, the rest after the wave's last unit
    const bool band_end = !TILED && itp.unit + 1 >= itp.unit_end;
    if (tail - head >= kD2mGroup || (band_end && tail > head)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // full searches; at the end of a band also what is left
      while (true) {
        const int avail = tail - head;
        const int take = avail >= kD2mGroup ? kD2mGroup : ((band_end && avail > 192) ? avail : 0);
        if (!take) break;
        search(std::integral_constant<int, 4>(), head, take);
        head += take;
      }
      if (band_end && tail > head) {
        if (tail - head > 128) search(std::integral_constant<int, 3>(), head, tail - head);
        else if (tail - head > 64) search(std::integral_constant<int, 2>(), head, tail - head);
        else search(std::integral_constant<int, 1>(), head, tail - head);
        head = tail;
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int d = 0; d + 1 < D2M_DEPTH; d++) zq[d][c] = zq[d + 1][c];
    }
    load_unit(itl, zq[D2M_DEPTH - 1]); advance_load(itl);
    advance_proc(itp);
  }
  if (TILED && tail > head) {   // what the wave's last units left
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (tail - head > 192) search(std::integral_constant<int, 4>(), head, tail - head);
    else if (tail - head > 128) search(std::integral_constant<int, 3>(), head, tail - head);
    else if (tail - head > 64) search(std::integral_constant<int, 2>(), head, tail - head);
    else search(std::integral_constant<int, 1>(), head, tail - head);
  }

  // ---- combine ----------------------------------------------------------------------------------
  if (loss_fx) atomicAdd(&s_loss, (unsigned long long)loss_fx);
  __syncthreads();
  if (tid == 0)
    loss_sum[blockIdx.x] = s_nan ? __builtin_nanf("") : (float)((double)(long long)s_loss * (1.0 / (double)kLossScale));
  if (WANT_GRAD && tid < J * 3) {
    const int j = tid / 3, c = tid - j * 3;
    long long t = 0;
#pragma unroll
    for (int k = 0; k < kD2mTables; k++) t += (long long)s_acc[k * kD2mTableStride + j * 4 + c];
    grad_centres[(size_t)blockIdx.x * J * 3 + tid] = (float)((double)t * (1.0 / (double)kGradScale));
  }
}



// ---------------------------------------------------------------------------------------------------------------
// TWO-STEP data->model (round 4): compact every observed IMAGE once, search the point lists per CROP.
//
// In the multiview loss every observed image is compared with the sphere sets of V crops
// (mesh/multiview_utility.py:99: real_dms expanded V times), and 40 % of the streaming kernel's VALU instructions --
// the resource it is bound by -- stream and compact pixels (SQ counters, round 4): work that depends on the image only.
//   step 1  d2m_compact_kernel: one workgroup (8 waves) per (image, region of 8192 consecutive pixels; measured with
//           16 / 8 / 4 waves: compaction 29.0 / 26.8 / 27.0 us, search 74.6 / 74.7 / 83.6 us at config 5).  Foreground
//           pixels (depth <= 99, mesh/render.py:138) are counted per TILE -- (SB waves) x (column block of 2^cbs
//           pixels), about 16 x 16 pixels -- a prefix over the tiles in boustrophedon order gives every (wave, column
//           block) its range, the region draws its place in the image's list from a counter (one atomic per region:
//           the order of the regions in the list is whatever it comes out as), and each pixel is written there as its
//           8-byte POINT (v << 16 | u, depth).  The list of an image is sorted by tile inside each region.
//   step 2  d2m_points_kernel: one workgroup per (crop, part).  Wave w takes the groups of 256 consecutive points
//           of its part (g = part mod parts) in turn (equal work by construction), requests the next group's
//           points while it searches the current one (d2m_search.h with the group's own bounding box: one or two
//           tiles, except the few groups that straddle two regions), and the fixed-point sums are combined as in the
//           streaming kernel.  Every lane of every group but an image's last holds four points.
// Same per-point code, conservative bounds, integer sums: results bit-identical to data_to_model_kernel's -- and
// independent of the order the regions arrived in.
constexpr int kCompactWaves = 8;
constexpr int kCompactUnits = 4;
constexpr int kRegionPixels = kCompactWaves * kCompactUnits * 256;   // 8192

__global__ void __launch_bounds__(kCompactWaves * 64)
d2m_compact_kernel(const float *__restrict__ depth, int H, int W, int geom, uint2 *__restrict__ points,
                   int *__restrict__ counts) {
  __shared__ int s_wcnt[kCompactWaves * 16], s_wcur[kCompactWaves * 16], s_base;
  const int m = blockIdx.x, region = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbs = geom & 0xff, SB = (geom >> 8) & 0xff, NC = (geom >> 16) & 0xff;
  const float *dm = depth + (size_t)m * H * W;
  const int P = H * W, w4 = W >> 2, nchunk = P >> 2;
  const int wshift = (w4 & (w4 - 1)) == 0 ? __builtin_ctz(w4) : -1;
  // this wave's units: one 16-byte piece per lane (W % 4 == 0: a piece never straddles a row)
  const int chunk0 = ((region * kCompactWaves + wave) * kCompactUnits) << 6;
  float4 t[kCompactUnits];
#pragma unroll
  for (int k = 0; k < kCompactUnits; k++)
    t[k] = reinterpret_cast<const float4 *>(dm)[min(chunk0 + (k << 6) + lane, nchunk - 1)];
  if (lane < 16) { s_wcnt[wave * 16 + lane] = 0; s_wcur[wave * 16 + lane] = 0; }   // (this wave's rows: its own LDS operations are ordered)
  auto unit = [&](int k, bool fg[4], int &v, int &x) {   // (pixels past the image's end do not exist)
    const int c = chunk0 + (k << 6) + lane;
    const bool in = c < nchunk;
    if (wshift >= 0) { v = c >> wshift; x = (c & (w4 - 1)) << 2; }
    else { v = c / w4; x = (c - v * w4) << 2; }
    fg[0] = in && !(t[k].x > 99.0f); fg[1] = in && !(t[k].y > 99.0f);
    fg[2] = in && !(t[k].z > 99.0f); fg[3] = in && !(t[k].w > 99.0f);
  };
#pragma unroll
  for (int k = 0; k < kCompactUnits; k++) {
    bool fg[4];
    int v, x;
    unit(k, fg, v, x);
    const int cnt = (int)fg[0] + (int)fg[1] + (int)fg[2] + (int)fg[3];
    if (cnt) atomicAdd(&s_wcnt[wave * 16 + (x >> cbs)], cnt);
  }
  // All four pieces are waited for HERE.  A piece past the image's end is never looked at (`in &&` above), so along that path
  // the compiler kept the loads "in flight" into the second pass and put vmcnt(3) .. vmcnt(0) in front of whatever reads
  // or reuses their registers there -- between the point stores, which count in the same in-order vmcnt queue: a wave stood
  // with at most three stores under way (SQ counters, round 5: 59 % of this kernel's wave cycles parked at a wait).
#ifndef EXP_COMPACT_NO_FENCE
#pragma unroll
  for (int k = 0; k < kCompactUnits; k++) asm volatile("" : : "v"(t[k].x), "v"(t[k].y), "v"(t[k].z), "v"(t[k].w));
#endif
  __syncthreads();
  // tile prefix: lane = position of a tile in the region's boustrophedon order (at most 64 tiles)
  const int nsb = (kCompactWaves + SB - 1) / SB, ntiles = nsb * NC;
  int tile_total = 0;
  {
    const int sbp = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)NC)), tt = lane - sbp * NC;
    const int tcp = (sbp & 1) ? NC - 1 - tt : tt;
    if (lane < ntiles)
      for (int w = sbp * SB; w < min(kCompactWaves, sbp * SB + SB); w++) tile_total += s_wcnt[w * 16 + tcp];
  }
  int incl = tile_total;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1, 2, 4, 8 inside rows of 16 ...
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
  {
    const int r0s = __builtin_amdgcn_readlane(incl, 15), r1s = __builtin_amdgcn_readlane(incl, 31);
    const int r2s = __builtin_amdgcn_readlane(incl, 47), row = lane >> 4;   // ... then the row totals
    incl += (row >= 1 ? r0s : 0) + (row >= 2 ? r1s : 0) + (row >= 3 ? r2s : 0);
  }
  const int T = __builtin_amdgcn_readlane(incl, 63), excl = incl - tile_total;
  if (T == 0) return;
  if (tid == 0) s_base = atomicAdd(&counts[m], T);     // this region's place in the image's list
  __syncthreads();
  uint2 *out = points + (size_t)m * P + s_base;
  const int my_sb = (int)(((float)wave + 0.5f) * __builtin_amdgcn_rcpf((float)SB));
#pragma unroll
  for (int k = 0; k < kCompactUnits; k++) {
    bool fg[4];
    int v, x;
    unit(k, fg, v, x);
    const int cnt = (int)fg[0] + (int)fg[1] + (int)fg[2] + (int)fg[3], tc = x >> cbs;
    const int pos = my_sb * NC + ((my_sb & 1) ? NC - 1 - tc : tc);
    int p = __builtin_amdgcn_ds_bpermute(pos << 2, excl);
    for (int w = my_sb * SB; w < wave; w++) p += s_wcnt[w * 16 + tc];
    if (cnt) {
      p += atomicAdd(&s_wcur[wave * 16 + tc], cnt);     // rank inside the wave's (tile) range: any order will do
      const unsigned vu = ((unsigned)v << 16) | (unsigned)x;
      const float tz[4] = {t[k].x, t[k].y, t[k].z, t[k].w};
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (fg[c]) { out[p] = make_uint2(vu + (unsigned)c, __float_as_uint(tz[c])); p++; }
    }
  }
}

template <bool WANT_GRAD, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
d2m_points_kernel(const uint2 *__restrict__ points, const int *__restrict__ counts, int P,
                  const int *__restrict__ depth_index, const float *__restrict__ centres, int centre_stride,
                  const float *__restrict__ radii, int J, int H, int W, int parts, float *__restrict__ loss_sum,
                  float *__restrict__ grad_centres, const int *__restrict__ centre_index, int slot_by_centre) {
  constexpr int K = 4, GS = 64 * K;
  __shared__ float4 s_c[SHR_MAX_SPHERES];
  __shared__ int s_odd, s_nan;
  __shared__ unsigned long long s_loss;
  __shared__ unsigned long long s_acc[WANT_GRAD ? kD2mTables * kD2mTableStride : 1];

  const int n = blockIdx.x / parts, part = blockIdx.x - n * parts;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = depth_index ? depth_index[n] : n;
  // (slot_by_centre, shr_data_to_model_from_points_ordered: centre_index is a permutation that only changes the launch
  // order -- the results go to the record set's own slots, not the workgroup's)
  const int out_slot = (slot_by_centre ? centre_index[n] : n) * parts + part;
  const int T = counts[m], G = (T + GS - 1) / GS;
  const uint2 *img = points + (size_t)m * P;
  // group g = points [256 g, 256 g + 256) of the image's list; lane l holds the four points 256 g + 4 l ... (32
  // contiguous, 16-byte aligned bytes per lane: two 16-byte loads; the list's last, partial group point by point)
  static_assert(K == 4, "two 16-byte loads per lane");
  auto load_group = [&](int g, uint2 e[K]) {
    const int i0 = g * GS + K * lane;
    if (g * GS + GS <= T) {
      const uint4 a = reinterpret_cast<const uint4 *>(img + i0)[0], b = reinterpret_cast<const uint4 *>(img + i0)[1];
      e[0] = make_uint2(a.x, a.y); e[1] = make_uint2(a.z, a.w); e[2] = make_uint2(b.x, b.y); e[3] = make_uint2(b.z, b.w);
    } else {
#pragma unroll
      for (int i = 0; i < K; i++) e[i] = img[min(i0 + i, T - 1)];
    }
  };
  // part p owns the groups g = p (mod parts) -- whatever the workgroup size: a part's sum does not depend on the launch
  // shape -- and its waves take them in turn
  const int stride = WAVES * parts;
  int g = part + parts * wave;
  uint2 e[K], en[K];
  if (g < G) load_group(g, e);                    // (in flight while wave 0 reads the records)
  if (wave == 0) {
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < J) {
      // (centre_index: crop n searches with record set centre_index[n] of the batch, shr_data_to_model_from_points_indexed)
      const float *p = centres + ((size_t)(centre_index ? centre_index[n] : n) * J + lane) * centre_stride;
      c = make_float4(p[0], p[1], p[2], radii[lane]);
    }
    s_c[lane] = c;
    const float inf = __builtin_inff();
    const bool bad = lane < J && (!(fabsf(c.x) < inf) || !(fabsf(c.y) < inf) || !(fabsf(c.z) < inf) || !(fabsf(c.w) < inf));
    const bool any = __ballot(bad) != 0ull;
    if (lane == 0) { s_odd = any; s_nan = 0; s_loss = 0ull; }
  }
  if (WANT_GRAD)
    for (int i = tid; i < kD2mTables * kD2mTableStride; i += WAVES * 64) s_acc[i] = 0ull;
  __syncthreads();
  D2mCtx ctx;
  ctx.s_c = s_c; ctx.cj = lane < J ? s_c[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  ctx.all = J >= 64 ? ~0ull : ((1ull << J) - 1ull);
  ctx.table_odd = s_odd != 0; ctx.J = J; ctx.lane = lane; ctx.ax = make_axis(W); ctx.ay = make_axis(H);
  ctx.s_acc = s_acc; ctx.acc_stride = kD2mTableStride; ctx.s_nan = &s_nan;
  long long loss_fx = 0;
  while (g < G) {
    const int gn = g + stride;
    if (gn < G) load_group(gn, en);               // the next group's points travel while this one is searched
#ifndef EXP_NO_SEARCH
    d2m_search_points<K, WANT_GRAD>(ctx, e, min(GS, T - g * GS), loss_fx);
#else
    loss_fx += (long long)e[0].x + (long long)e[3].z;
#endif
#pragma unroll
    for (int i = 0; i < K; i++) e[i] = en[i];
    g = gn;
  }
  if (loss_fx) atomicAdd(&s_loss, (unsigned long long)loss_fx);
  __syncthreads();
  if (tid == 0)
    loss_sum[out_slot] = s_nan ? __builtin_nanf("") : (float)((double)(long long)s_loss * (1.0 / (double)kLossScale));
  if (WANT_GRAD && tid < J * 3) {
    const int j = tid / 3, c = tid - j * 3;
    long long t = 0;
#pragma unroll
    for (int k = 0; k < kD2mTables; k++) t += (long long)s_acc[k * kD2mTableStride + j * 4 + c];
    grad_centres[(size_t)out_slot * J * 3 + tid] = (float)((double)t * (1.0 / (double)kGradScale));
  }
}

}  // namespace shr

namespace {
int g_d2m_waves = 0;   // 0 = by batch size (SHR_TUNE_D2M_WAVES)
int g_d2m_band = 0;    // 0 = by crop size (SHR_TUNE_D2M_BAND_UNITS)
int g_d2m_tiled = -1;  // units = 32 x 8-pixel tiles + box bounds: -1 = from 192 x 192 pixels on, 0 = never, 1 = wherever W % 4 == 0 (SHR_TUNE_D2M_TILED)

template <bool WANT_GRAD, bool TILED>
void launch_d2m_waves(int waves, const float *depth, const int32_t *depth_index, const float *centres,
                      int centre_stride, const float *radii, int N, int J, int H, int W, int band_units, int parts, float *loss_sum,
                      float *grad_centres, hipStream_t s) {
  using namespace shr;
  const dim3 grid((unsigned)(N * parts));
  if (waves >= 16)
    hipLaunchKernelGGL((data_to_model_kernel<WANT_GRAD, 16, TILED>), grid, dim3(1024), 0, s, depth, depth_index, centres, centre_stride,
                       radii, J, H, W, band_units, parts, loss_sum, grad_centres);
  else if (waves >= 8)
    hipLaunchKernelGGL((data_to_model_kernel<WANT_GRAD, 8, TILED>), grid, dim3(512), 0, s, depth, depth_index, centres, centre_stride,
                       radii, J, H, W, band_units, parts, loss_sum, grad_centres);
  else
    hipLaunchKernelGGL((data_to_model_kernel<WANT_GRAD, 4, TILED>), grid, dim3(256), 0, s, depth, depth_index, centres, centre_stride,
                       radii, J, H, W, band_units, parts, loss_sum, grad_centres);
}

int launch_d2m(const float *depth, const int32_t *depth_index, const float *centres, int centre_stride,
               const float *radii, int N, int J, int H, int W, int parts, float *loss_sum, float *grad_centres,
               void *stream) {
  if (N == 0) return SHR_OK;
  if (!depth || !centres || !radii || !loss_sum || N < 0 || J <= 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if ((parts != 1 && parts != 2 && parts != 4) || (centre_stride != 3 && centre_stride != 4)) return SHR_EINVAL;
  // ring entries pack (v, u) into 16 bits each
  if (J > SHR_MAX_SPHERES || (long long)H * W > (1LL << 30) || H > 65535 || W > 65535 ||
      (long long)N * parts > 0x7fffffffLL)
    return SHR_ETOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  // waves per workgroup: enough waves in flight to fill 256 CUs x 4 SIMDs whatever the batch size
  const long long wgs = (long long)N * parts;
  const int waves = g_d2m_waves ? g_d2m_waves : (wgs >= 1024 ? 4 : (wgs >= 384 ? 8 : 16));
  // bands of consecutive units, handed out dynamically: small enough to balance the waves, large enough that the
  // remainder search at the end of every band stays a small share (tools/exp_d2m_parts.py, round 3: 3 units at
  // 128 x 128 with four waves -- 52.5 us against 55-58 with 2 and 57 with 4 for 1152 crops --, 4-8 at 256 x 256)
  const int units = (int)(((long long)H * W + 255) >> 8);
  int band_units = units / (5 * waves * parts);
  if (band_units > 8) band_units = 8;
  // TILED: 32 x 8-pixel units, bands = a quarter of a tile column (2 .. 8 tile rows)
  // (measured, 1152 crops: 146 -> 128-135 us at 256 x 256, 55 -> 61 us at 128 x 128 -- a 32 x 8 tile is half a hand there)
  const bool tiled = (g_d2m_tiled == 1 || (g_d2m_tiled < 0 && (long long)H * W >= 192LL * 192LL)) && (W % 4) == 0 &&
                     (((uintptr_t)depth) & 15u) == 0;
  if (tiled) {
    band_units = ((H + 7) >> 3) / 4;
    if (band_units > 8) band_units = 8;
    if (band_units < 2) band_units = 2;
  }
  if (g_d2m_band) band_units = g_d2m_band;
  if (band_units < 1) band_units = 1;
#define D2M_LAUNCH(G, T) launch_d2m_waves<G, T>(waves, depth, depth_index, centres, centre_stride, radii, N, J, H, W, \
                                                band_units, parts, loss_sum, grad_centres, s)
  if (grad_centres) { if (tiled) D2M_LAUNCH(true, true); else D2M_LAUNCH(true, false); }
  else { if (tiled) D2M_LAUNCH(false, true); else D2M_LAUNCH(false, false); }
#undef D2M_LAUNCH
  return (int)hipGetLastError();
}
}  // namespace

// launch-shape hooks behind shr_set_tuning (results never depend on them: the sums are integers)
int shr::d2m_set_waves(int waves) {
  if (waves != 0 && waves != 4 && waves != 8 && waves != 16) return SHR_EINVAL;
  g_d2m_waves = waves;
  return SHR_OK;
}
int shr::d2m_set_tiled(int on) {
  if (on < -1 || on > 1) return SHR_EINVAL;
  g_d2m_tiled = on;
  return SHR_OK;
}
int shr::d2m_set_band_units(int units) {
  if (units < 0 || units > 4096) return SHR_EINVAL;
  g_d2m_band = units;
  return SHR_OK;
}

extern "C" int shr_data_to_model(const float *depth, const float *centres, const float *radii, int N, int J, int H,
                                 int W, float *loss_sum, float *grad_centres, void *stream) {
  return launch_d2m(depth, nullptr, centres, 3, radii, N, J, H, W, 1, loss_sum, grad_centres, stream);
}

extern "C" int shr_data_to_model_indexed(const float *depth, const int32_t *depth_index, const float *centres,
                                         const float *radii, int N, int J, int H, int W, float *loss_sum,
                                         float *grad_centres, void *stream) {
  if (!depth_index && N > 0) return SHR_EINVAL;
  return launch_d2m(depth, depth_index, centres, 3, radii, N, J, H, W, 1, loss_sum, grad_centres, stream);
}

// Large crops are split over several workgroups (on different CUs): the kernel then writes `parts` partial
// results per crop and the caller adds them (fixed order: deterministic).
extern "C" int shr_data_to_model_parts(int N, int H, int W) {
  (void)N;
  return (long long)H * W >= 192LL * 192LL ? 2 : 1;
}

extern "C" int shr_data_to_model_partial(const float *depth, const int32_t *depth_index, const float *centres,
                                         int centre_stride, const float *radii, int N, int J, int H, int W, int parts,
                                         float *loss_parts, float *grad_parts, void *stream) {
  return launch_d2m(depth, depth_index, centres, centre_stride, radii, N, J, H, W, parts, loss_parts, grad_parts,
                    stream);
}

// ---- two-step data->model: C ABI -----------------------------------------------------------------------------------
namespace {
// Tiles of a region (d2m_compact_kernel): a wave holds 1024 / W rows (a band); SB waves x one column block of 2^cbs
// pixels make a tile of about 16 x 16 pixels; at most 16 column blocks (a wave's counter row) and 64 tiles (one per
// lane of the prefix).  Returns cbs | SB << 8 | NC << 16.
int d2m_tile_geometry(int W) {
  int sb = W / 64;
  if (sb < 1) sb = 1;
  if (sb > shr::kCompactWaves) sb = shr::kCompactWaves;
  const int nsb = (shr::kCompactWaves + sb - 1) / sb;
  int cbs = 4;
  while (((W + (1 << cbs) - 1) >> cbs) > 16 || nsb * ((W + (1 << cbs) - 1) >> cbs) > 64) cbs++;
  return cbs | (sb << 8) | (((W + (1 << cbs) - 1) >> cbs) << 16);
}
bool d2m_points_ok(int H, int W) {
  return H > 0 && W > 0 && (W & 3) == 0 && W <= 16384 && H <= 65535 && (long long)H * W <= (1LL << 28);
}
size_t d2m_counts_offset(int M, int H, int W) { return (size_t)M * H * W * sizeof(uint2); }
}  // namespace

extern "C" long long shr_data_to_model_points_bytes(int M, int H, int W) {
  if (!d2m_points_ok(H, W) || M <= 0) return 0;
  return (long long)d2m_counts_offset(M, H, W) + 4LL * M;
}

namespace shr {
// argument checks of the compaction; *counts = the lists' fill counters inside the workspace
int d2m_compact_check(const float *depth, int M, int H, int W, void *workspace, int **counts) {
  if (!depth || !workspace || M < 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if (!d2m_points_ok(H, W)) return SHR_ETOOLARGE;
  if (((((uintptr_t)depth) | ((uintptr_t)workspace)) & 15u) != 0) return SHR_EINVAL;
  *counts = reinterpret_cast<int *>(static_cast<unsigned char *>(workspace) + d2m_counts_offset(M, H, W));
  return SHR_OK;
}
// the compaction launch alone: the fill counters are zero already (checked arguments)
int d2m_compact_launch(const float *depth, int M, int H, int W, void *workspace, hipStream_t s) {
  const int R = (int)(((long long)H * W + kRegionPixels - 1) / kRegionPixels);
  uint2 *points = static_cast<uint2 *>(workspace);
  int *counts = reinterpret_cast<int *>(static_cast<unsigned char *>(workspace) + d2m_counts_offset(M, H, W));
  hipLaunchKernelGGL(d2m_compact_kernel, dim3((unsigned)M, (unsigned)R), dim3(kCompactWaves * 64), 0, s, depth, H, W,
                     d2m_tile_geometry(W), points, counts);
  return (int)hipGetLastError();
}
}  // namespace shr

extern "C" int shr_data_to_model_compact(const float *depth, int M, int H, int W, void *workspace, void *stream) {
  using namespace shr;
  if (M == 0) return SHR_OK;
  int *counts = nullptr;
  const int rc = d2m_compact_check(depth, M, H, W, workspace, &counts);
  if (rc != SHR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);   // the lists' fill counters
  if (e != hipSuccess) return (int)e;
  return d2m_compact_launch(depth, M, H, W, workspace, s);
}

extern "C" int shr_data_to_model_from_points(const void *workspace, int M, const int32_t *depth_index,
                                             const float *centres, int centre_stride, const float *radii, int N,
                                             int J, int H, int W, int parts, float *loss_parts, float *grad_parts,
                                             void *stream) {
  return shr_data_to_model_from_points_indexed(workspace, M, depth_index, nullptr, centres, centre_stride, radii, N, J, H, W,
                                               parts, loss_parts, grad_parts, stream);
}

static int d2m_from_points_launch(const void *workspace, int M, const int32_t *depth_index, const int32_t *centre_index,
                                  int slot_by_centre, const float *centres, int centre_stride, const float *radii, int N, int J,
                                  int H, int W, int parts, float *loss_parts, float *grad_parts, void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!workspace || !centres || !radii || !loss_parts || N < 0 || M <= 0 || J <= 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if (parts < 1 || parts > 64 || (centre_stride != 3 && centre_stride != 4) || (!depth_index && M != N)) return SHR_EINVAL;
  if (!d2m_points_ok(H, W) || J > SHR_MAX_SPHERES || (long long)N * parts > 0x7fffffffLL) return SHR_ETOOLARGE;
  const uint2 *points = static_cast<const uint2 *>(workspace);
  const int *counts = reinterpret_cast<const int *>(static_cast<const unsigned char *>(workspace) + d2m_counts_offset(M, H, W));
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(N * parts));
  // (measured at 1152 crops @256x256: 8 waves x 1 part 75 us, 4 x 2 76, 4 x 1 88, 16 x 1 see DESIGN)
  const long long wgs = (long long)N * parts;
  const int waves = g_d2m_waves ? g_d2m_waves : (wgs >= 2048 ? 4 : (wgs >= 768 ? 8 : 16));
#define D2P_LAUNCH(G, NW) hipLaunchKernelGGL((d2m_points_kernel<G, NW>), grid, dim3(NW * 64), 0, s, points, counts, H * W, \
                                             depth_index, centres, centre_stride, radii, J, H, W, parts, loss_parts, grad_parts, centre_index, slot_by_centre)
  if (grad_parts) { if (waves >= 16) D2P_LAUNCH(true, 16); else if (waves >= 8) D2P_LAUNCH(true, 8); else D2P_LAUNCH(true, 4); }
  else { if (waves >= 16) D2P_LAUNCH(false, 16); else if (waves >= 8) D2P_LAUNCH(false, 8); else D2P_LAUNCH(false, 4); }
#undef D2P_LAUNCH
  return (int)hipGetLastError();
}

extern "C" int shr_data_to_model_from_points_indexed(const void *workspace, int M, const int32_t *depth_index,
                                                     const int32_t *centre_index, const float *centres, int centre_stride,
                                                     const float *radii, int N, int J, int H, int W, int parts,
                                                     float *loss_parts, float *grad_parts, void *stream) {
  return d2m_from_points_launch(workspace, M, depth_index, centre_index, 0, centres, centre_stride, radii, N, J, H, W, parts,
                                loss_parts, grad_parts, stream);
}

// `order`: a permutation of the N record sets -- workgroup w searches for record set order[w] (against image
// depth_index[w]: the caller permutes that array alike) and writes that SET's slots: shr_data_to_model_from_points'
// results in another launch order (XCD placement: the pairs that read one image's list on one L2)
extern "C" int shr_data_to_model_from_points_ordered(const void *workspace, int M, const int32_t *depth_index,
                                                     const int32_t *order, const float *centres, int centre_stride,
                                                     const float *radii, int N, int J, int H, int W, int parts,
                                                     float *loss_parts, float *grad_parts, void *stream) {
  if (!order || !depth_index) return SHR_EINVAL;
  return d2m_from_points_launch(workspace, M, depth_index, order, 1, centres, centre_stride, radii, N, J, H, W, parts,
                                loss_parts, grad_parts, stream);
}
