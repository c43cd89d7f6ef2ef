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
  __device__ void seek(int b, int nbands, int band_units, int units, int parts = 1, int part = 0) {
    band = b;
    done = b * parts + part >= nbands;
    unit = (b * parts + part) * band_units;
    unit_end = min(units, unit + band_units);
  }
};

template <bool WANT_GRAD, int WAVES>
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
  const int units = (P + 255) >> 8, nbands = (units + band_units - 1) / band_units;
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
    d2m_search<K, WANT_GRAD>(ctx, [&](int idx) { return ring[d2m_slot(head + idx)]; }, count, loss_fx);
  };

  // ---- this wave's units: three loads in flight, compact, search ------------------------------
  auto load_unit = [&](const D2mUnitIter &it, float z[4]) {
    z[0] = z[1] = z[2] = z[3] = 100.f;
    if (it.done) return;
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
  // the LOAD iterator draws the bands (it runs three units ahead and may be up to three bands ahead: the drawn
  // bands wait in a 4-entry queue), the PROCESS iterator follows the same sequence
  int q_put = 0, q_get = 0;
  auto advance_load = [&](D2mUnitIter &it) {
    if (it.done) return;
    if (++it.unit >= it.unit_end) {
      int b = 0;
      if (lane == 0) b = atomicAdd(&s_next_band, 1);
      b = __builtin_amdgcn_readfirstlane(b);
      if (lane == 0) s_bandq[wave][q_put & 7] = b;
      q_put++;
      it.seek(b, nbands, band_units, units, parts, part);
    }
  };
  auto advance_proc = [&](D2mUnitIter &it) {
    if (it.done) return;
    if (++it.unit >= it.unit_end) {
      const int b = __builtin_amdgcn_readfirstlane(s_bandq[wave][q_get & 7]);
      q_get++;
      it.seek(b, nbands, band_units, units, parts, part);
    }
  };
  D2mUnitIter itp, itl;
  itp.seek(wave, nbands, band_units, units, parts, part);
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
    // foreground flags (mesh/render.py:138: background = d > 99), exclusive prefix in pixel order
    bool fg[4];
    int before = 0, total = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      fg[c] = (p0 + c < P) && !(z0[c] > 99.0f);
      const unsigned long long m = __ballot(fg[c]);
      before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, before));
      total += __builtin_popcountll(m);
    }
    if (total) {
      // `before` = flagged pixels of LOWER lanes over all four components: this lane's entries start there
      // (a lane's four pixels are consecutive: pixel order)
      int pos = tail + before;
      int vc, uc;
      if (wshift >= 0) { vc = p0 >> wshift; uc = p0 & (W - 1); }
      else { vc = p0 / W; uc = p0 - vc * W; }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        while (uc >= W) { uc -= W; vc++; }      // never taken when W % 4 == 0
        if (fg[c]) {
          ring[d2m_slot(pos)] = make_uint2(((unsigned)vc << 16) | (unsigned)uc, __float_as_uint(z0[c]));
          pos++;
        }
        uc++;
      }
      tail += total;
    }
    const bool band_end = itp.unit + 1 >= itp.unit_end;
    if (tail - head >= kD2mGroup || (band_end && tail > head)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // full searches; at the end of a band also what is left (a search never spans two bands)
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

}  // namespace shr

namespace {
int g_d2m_waves = 0;   // 0 = by batch size (SHR_TUNE_D2M_WAVES)
int g_d2m_band = 0;    // 0 = by crop size (SHR_TUNE_D2M_BAND_UNITS)

template <bool WANT_GRAD>
void launch_d2m_waves(int waves, const float *depth, const int32_t *depth_index, const float *centres,
                      int centre_stride, const float *radii, int N, int J, int H, int W, int band_units, int parts, float *loss_sum,
                      float *grad_centres, hipStream_t s) {
  using namespace shr;
  const dim3 grid((unsigned)(N * parts));
  if (waves >= 16)
    hipLaunchKernelGGL((data_to_model_kernel<WANT_GRAD, 16>), grid, dim3(1024), 0, s, depth, depth_index, centres, centre_stride,
                       radii, J, H, W, band_units, parts, loss_sum, grad_centres);
  else if (waves >= 8)
    hipLaunchKernelGGL((data_to_model_kernel<WANT_GRAD, 8>), grid, dim3(512), 0, s, depth, depth_index, centres, centre_stride,
                       radii, J, H, W, band_units, parts, loss_sum, grad_centres);
  else
    hipLaunchKernelGGL((data_to_model_kernel<WANT_GRAD, 4>), grid, dim3(256), 0, s, depth, depth_index, centres, centre_stride,
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
  if (g_d2m_band) band_units = g_d2m_band;
  if (band_units < 1) band_units = 1;
  if (grad_centres)
    launch_d2m_waves<true>(waves, depth, depth_index, centres, centre_stride, radii, N, J, H, W, band_units, parts, loss_sum,
                           grad_centres, s);
  else
    launch_d2m_waves<false>(waves, depth, depth_index, centres, centre_stride, radii, N, J, H, W, band_units, parts, loss_sum,
                            grad_centres, s);
  return (int)hipGetLastError();
}
}  // namespace

// launch-shape hooks behind shr_set_tuning (results never depend on them: the sums are integers)
int shr::d2m_set_waves(int waves) {
  if (waves != 0 && waves != 4 && waves != 8 && waves != 16) return SHR_EINVAL;
  g_d2m_waves = waves;
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
