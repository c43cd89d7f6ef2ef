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
// The nearest surface can be ANY sphere (no culling is valid), but only ~15 % of
// the pixels are foreground, so each 4096-pixel chunk is first compacted:
//   1. coalesced 16-byte depth loads, foreground flags, deterministic block scan;
//   2. foreground pixels packed into an LDS queue (lanes fully used from here);
//   3. each wave bounds the J spheres against the bounding box of its 64 queue entries
//      (lanes = spheres) and searches only the candidates that can be nearest (lanes =
//      points) and adds the clamped distance;
//   4. the gradient vectors of a wave's 64 points are summed per owner with DPP wave
//      sums into the wave's private LDS row; rows are combined in wave order:
//      deterministic, no atomics.
// One workgroup per crop.  HBM: reads 4*H*W + 12*J + 4*J bytes per crop.

#include "common.h"

namespace shr {

constexpr int kD2mThreads = 512;   // 8 waves: two workgroups per CU overlap one crop's loads with the other's search
constexpr int kD2mPix = 16;                           // a 4x4-pixel block per thread per chunk (4 x 16-byte loads in flight);
                                                      // 1024 threads = 16384 px = a whole 128x128 crop
constexpr int kD2mQueue = 2048;                       // queue entries (32 KB); denser chunks take extra passes

constexpr int kD2mSeeds = 3;                           // best-first visits before the candidate set is fixed

struct QEntry { float a, b, c; int d; };  // phase 2: (xg, yg, z, -) ; phase 3: (gx, gy, gz, owner)

template <bool WANT_GRAD>
__global__ void __launch_bounds__(kD2mThreads)
data_to_model_kernel(const float *__restrict__ depth, const int *__restrict__ depth_index,
                     const float *__restrict__ centres, const float *__restrict__ radii, int J, int H, int W,
                     float *__restrict__ loss_sum, float *__restrict__ grad_centres) {
  __shared__ float4 s_c[SHR_MAX_SPHERES];     // (cx, cy, cz, r)
  __shared__ int s_wave_cnt[kD2mThreads / 64];
  __shared__ float s_wave_loss[kD2mThreads / 64];
  __shared__ QEntry s_q[kD2mQueue];           // 32 KB
  __shared__ float4 s_part[(kD2mThreads / 64) * SHR_MAX_SPHERES];   // [wave][sphere] gradient partials

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < J) {
    const float *c = centres + ((size_t)n * J + tid) * 3;
    s_c[tid] = make_float4(c[0], c[1], c[2], radii[tid]);
  }
  const float *dm = depth + (size_t)(depth_index ? depth_index[n] : n) * H * W;
  const Axis ax = make_axis(W), ay = make_axis(H);
  const bool row4 = (W % 4 == 0) && is_aligned16(dm);

  float loss = 0.f;
  if (WANT_GRAD) s_part[tid] = make_float4(0.f, 0.f, 0.f, 0.f);   // 1024 = 16 waves x 64 spheres

  // Threads own 4x4-pixel blocks (block id = chunk base + thread id, row-major over the
  // ceil(W/4) x ceil(H/4) block grid): each of a thread's four 16-byte loads is contiguous
  // with its neighbours' in x, and the queue (thread order, row-major inside a block) keeps
  // neighbouring pixels together, so a wave's 64 entries have a tight bounding box (step 3).
  // The 512 blocks of a chunk form a 32 x 16 tile visited in Morton order: 64 consecutive
  // threads = an 8 x 8 group of blocks (32 x 32 px), 4-5 consecutive blocks = a near-square
  // patch, so the points a wave searches together are neighbours in x AND y.
  const int nbx = (W + 3) >> 2, nby = (H + 3) >> 2;
  const int tiles_x = (nbx + 31) >> 5, tiles_y = (nby + 15) >> 4;
  auto even_bits = [](int t) { t &= 0x55; t = (t | (t >> 1)) & 0x33; return (t | (t >> 2)) & 0x0f; };
  static_assert(kD2mThreads == 512, "tile = 32 x 16 blocks");
  const int mx = even_bits(tid) | ((tid >> 8) << 4), my = even_bits(tid >> 1);
  for (int tile = 0; tile < tiles_x * tiles_y; tile++) {
    // ---- 1. load the block (all four loads issued before the first use), flag foreground
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int bx = tile_x * 32 + mx, by = tile_y * 16 + my;
    const bool inside = bx < nbx && by < nby;
    const int u0 = bx * 4, v0 = by * 4;
    float z[kD2mPix];
    int cnt = 0;
    unsigned fgmask = 0;
    if (row4) {
      // unconditional, clamped: the four requests go out back to back (one round trip); what
      // lies outside the image is masked below
      const int byc = min(by, nby - 1), bxc = min(bx, nbx - 1);
      const float4 *p0 = reinterpret_cast<const float4 *>(dm + (size_t)min(byc * 4 + 0, H - 1) * W) + bxc;
      const float4 *p1 = reinterpret_cast<const float4 *>(dm + (size_t)min(byc * 4 + 1, H - 1) * W) + bxc;
      const float4 *p2 = reinterpret_cast<const float4 *>(dm + (size_t)min(byc * 4 + 2, H - 1) * W) + bxc;
      const float4 *p3 = reinterpret_cast<const float4 *>(dm + (size_t)min(byc * 4 + 3, H - 1) * W) + bxc;
      const float4 t0 = *p0, t1 = *p1, t2 = *p2, t3 = *p3;
      z[0] = t0.x; z[1] = t0.y; z[2] = t0.z; z[3] = t0.w;
      z[4] = t1.x; z[5] = t1.y; z[6] = t1.z; z[7] = t1.w;
      z[8] = t2.x; z[9] = t2.y; z[10] = t2.z; z[11] = t2.w;
      z[12] = t3.x; z[13] = t3.y; z[14] = t3.z; z[15] = t3.w;
    } else {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int v = v0 + g;
        float4 t = make_float4(100.f, 100.f, 100.f, 100.f);
        if (inside && v < H) {
          const float *rowp = dm + (size_t)v * W + u0;
          if (u0 + 0 < W) t.x = rowp[0];
          if (u0 + 1 < W) t.y = rowp[1];
          if (u0 + 2 < W) t.z = rowp[2];
          if (u0 + 3 < W) t.w = rowp[3];
        }
        z[4 * g] = t.x; z[4 * g + 1] = t.y; z[4 * g + 2] = t.z; z[4 * g + 3] = t.w;
      }
    }
#pragma unroll
    for (int k = 0; k < kD2mPix; k++) {
      const bool in = inside && (v0 + (k >> 2)) < H && (u0 + (k & 3)) < W;
      const bool fg = in && !(z[k] > 99.0f);  // mesh/render.py:138 background = d > 99
      fgmask |= (unsigned)fg << k;
      cnt += fg;
    }
    // deterministic exclusive scan of cnt over the workgroup (queue order = thread order):
    // 4 DPP steps inside each row of 16 lanes, then the row totals via SGPR broadcasts
    int incl = cnt;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8
    {
      const int r0s = __builtin_amdgcn_readlane(incl, 15), r1s = __builtin_amdgcn_readlane(incl, 31);
      const int r2s = __builtin_amdgcn_readlane(incl, 47);
      const int row = lane >> 4;
      incl += (row >= 1 ? r0s : 0) + (row >= 2 ? r1s : 0) + (row >= 3 ? r2s : 0);
    }
    if (tile > 0) __syncthreads();  // previous chunk's queue fully consumed
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    int offset = incl - cnt, total_all = 0;
    for (int w = 0; w < kD2mThreads / 64; w++) {
      const int c = s_wave_cnt[w];
      if (w < wave) offset += c;
      total_all += c;
    }
    // a chunk with more foreground than the queue holds is consumed in several passes
    for (int q0 = 0; q0 < total_all; q0 += kD2mQueue) {
    const int total = min(kD2mQueue, total_all - q0);
    if (q0 > 0) __syncthreads();
    // ---- 2. compact ---------------------------------------------------------------------
    {
      int slot = offset - q0;
      const float xg0 = axis_coord(ax, u0), xg1 = axis_coord(ax, u0 + 1), xg2 = axis_coord(ax, u0 + 2),
                  xg3 = axis_coord(ax, u0 + 3);
#pragma unroll
      for (int k = 0; k < kD2mPix; k++) {
        if ((fgmask >> k) & 1u) {
          if (slot >= 0 && slot < kD2mQueue) {
            QEntry e;
            e.a = (k & 3) == 0 ? xg0 : ((k & 3) == 1 ? xg1 : ((k & 3) == 2 ? xg2 : xg3));
            e.b = axis_coord(ay, v0 + (k >> 2));
            e.c = z[k];
            e.d = 0;
            s_q[slot] = e;
          }
          ++slot;
        }
      }
    }
    __syncthreads();
    // ---- 3. nearest-surface search per foreground pixel ------------------------------
    // A wave takes 64 consecutive queue entries (neighbouring pixels in scan order).  With
    // lanes = spheres it bounds a_j = | ||p - c_j|| - r_j | from below over the points' bounding
    // box (lb_j), then lanes = points visit the few spheres with the smallest lb_j first, after
    // which the running minima are millimetres (the points lie on the model's surface) and
    // only spheres whose lb_j is below the largest of them remain candidates (10-15 of 41).
    // Bounds carry a rounding slack; a NaN anywhere disables the pruning.
    {
      const float4 cj = lane < J ? s_c[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i0 = wave * 64; i0 < total; i0 += kD2mThreads) {
        const int i = i0 + lane;
        const bool act = i < total;
        QEntry e = s_q[act ? i : i0];
        const float inf = __builtin_inff();
        // the points' bounding box: six wave minima (max = -min(-v)) in two transposed reductions
        const float mxy = wave_min4_transposed(e.a, -e.a, e.b, -e.b, lane), mz = wave_min4_transposed(e.c, -e.c, e.c, -e.c, lane);
        const float xlo = readlane_f(mxy, 12), xhi = -readlane_f(mxy, 13), ylo = readlane_f(mxy, 14), yhi = -readlane_f(mxy, 15);
        const float zlo = readlane_f(mz, 12), zhi = -readlane_f(mz, 13);
        // lanes = spheres: nearest / farthest distance from the centre to the box
        const float nx = fmaxf(fmaxf(xlo - cj.x, cj.x - xhi), 0.f), fx = fmaxf(fabsf(xlo - cj.x), fabsf(xhi - cj.x));
        const float ny = fmaxf(fmaxf(ylo - cj.y, cj.y - yhi), 0.f), fy = fmaxf(fabsf(ylo - cj.y), fabsf(yhi - cj.y));
        const float nz = fmaxf(fmaxf(zlo - cj.z, cj.z - zhi), 0.f), fz = fmaxf(fabsf(zlo - cj.z), fabsf(zhi - cj.z));
        const float dmin = __builtin_amdgcn_sqrtf((nx * nx + ny * ny) + nz * nz);
        const float dmax = __builtin_amdgcn_sqrtf((fx * fx + fy * fy) + fz * fz);
        const float lb = fmaxf(fmaxf(dmin - cj.w, cj.w - dmax), 0.f);
        const bool odd = __ballot((e.a != e.a) || (e.b != e.b) || (e.c != e.c) ||
                                  (lane < J && (lb != lb || cj.x != cj.x || cj.y != cj.y || cj.z != cj.z || cj.w != cj.w))) != 0ull;
        float best = 0.f;
        int bj = 0;
        auto surface_distance = [&](int j) {   // lanes = points, sphere j's record through SGPRs
          const float cx = readlane_f(cj.x, j), cy = readlane_f(cj.y, j), cz = readlane_f(cj.z, j);
          const float cr = readlane_f(cj.w, j);
          const float dx = e.a - cx, dy = e.b - cy, dz = e.c - cz;
          const float dist = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);  // <= 1 ulp: loss is continuous
          return fabsf(dist - cr);
        };
        if (odd) {   // a NaN somewhere: every sphere, in index order (torch.min: NaN wins, ties keep first)
          for (int j = 0; j < J; j++) {
            const float a = surface_distance(j);
            if (j == 0 || ((best == best) && (a < best || a != a))) { best = a; bj = j; }
          }
        } else {
          // Two stages.  (1) Best first: visit the kD2mSeeds unvisited spheres with the smallest
          // lower bounds; the largest of the 64 running minima (`reach`) is then small, the
          // points lie on the model's surface.  (2) No sphere whose lower bound (discounted
          // for rounding) exceeds `reach` can reach, or tie, any point's minimum: the others
          // are visited in a plain loop.
          float rem = lane < J ? lb * 0.99999f - 1e-3f : inf;
          best = inf;
          for (int it = 0; it < kD2mSeeds; it++) {
            const float m = wave_minmax_all<true>(rem);
            if (m == inf) break;
            const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(rem == m)));
            if (lane == j) rem = inf;
            const float a = surface_distance(j);
            if (a < best || (a == best && j < bj)) { best = a; bj = j; }   // ties keep the first index
          }
          // (one wave maximum after the seeds, not one per seed: a seed that could not have improved
          // anything costs one evaluation, a wave reduction costs as much)
          const float reach = wave_minmax_all<false>(act ? best : -inf) * 1.00001f + 1e-3f;
          unsigned long long cand = __ballot(rem <= reach && rem != inf);
          while (cand) {
            const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(cand));
            cand &= cand - 1;
            const float a = surface_distance(j);
            if (a < best || (a == best && j < bj)) { best = a; bj = j; }
          }
        }
        if (act) loss += (best != best) ? best : fminf(fmaxf(best, 0.f), 50.f);   // torch.clamp keeps NaN
        if (WANT_GRAD) {
          float gx = 0.f, gy = 0.f, gz = 0.f;
          int owner = -1;
          if (act) {
            const float4 c = s_c[bj];
            const float dx = e.a - c.x, dy = e.b - c.y, dz = e.c - c.z;
            const float dist = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
            const float t = dist - c.w;
            const float sgn = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
            const bool live = (best <= 50.f) && (dist != 0.f) && (sgn != 0.f);
            const float k = live ? -(sgn / dist) : 0.f;
            gx = k * dx; gy = k * dy; gz = k * dz;
            owner = live ? bj : -1;
          }
          // the 64 neighbouring points have 2-4 distinct owners: one DPP wave sum per owner
          // into this wave's private LDS row (fixed order: deterministic)
          unsigned long long todo = __ballot(owner >= 0);
          while (todo) {
            const int j = __builtin_amdgcn_readlane(owner, __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo)));
            const bool mine = owner == j;
            todo &= ~__ballot(mine);
            // (three sums in one transposed reduction; ds_add_f32 into the wave's own slot)
            const float t = wave_sum4_transposed(mine ? gx : 0.f, mine ? gy : 0.f, mine ? gz : 0.f, 0.f, lane);
            if (lane >= 60 && lane < 63)
              atomicAdd(reinterpret_cast<float *>(s_part + wave * SHR_MAX_SPHERES + j) + (lane & 3), t);
          }
        }
      }
    }
    }  // passes
  }

  // ---- reductions ---------------------------------------------------------------------------
  loss = wave_sum_lane63(loss);
  if (lane == 63) s_wave_loss[wave] = loss;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < kD2mThreads / 64; w++) t += s_wave_loss[w];
    loss_sum[n] = t;
  }
  if (WANT_GRAD && tid < J) {   // combine the waves' partials in wave order
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int w = 0; w < kD2mThreads / 64; w++) {
      const float4 a = s_part[w * SHR_MAX_SPHERES + tid];
      gx += a.x; gy += a.y; gz += a.z;
    }
    float *o = grad_centres + ((size_t)n * J + tid) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
  }
}

}  // namespace shr

namespace {
int launch_d2m(const float *depth, const int32_t *depth_index, const float *centres, const float *radii, int N, int J,
               int H, int W, float *loss_sum, float *grad_centres, void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!depth || !centres || !radii || !loss_sum || N < 0 || J <= 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if (J > SHR_MAX_SPHERES || (long long)H * W > (1LL << 30)) return SHR_ETOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  if (grad_centres)
    hipLaunchKernelGGL(data_to_model_kernel<true>, dim3((unsigned)N), dim3(kD2mThreads), 0, s, depth, depth_index,
                       centres, radii, J, H, W, loss_sum, grad_centres);
  else
    hipLaunchKernelGGL(data_to_model_kernel<false>, dim3((unsigned)N), dim3(kD2mThreads), 0, s, depth, depth_index,
                       centres, radii, J, H, W, loss_sum, grad_centres);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" int shr_data_to_model(const float *depth, const float *centres, const float *radii, int N, int J, int H,
                                 int W, float *loss_sum, float *grad_centres, void *stream) {
  return launch_d2m(depth, nullptr, centres, radii, N, J, H, W, loss_sum, grad_centres, stream);
}

extern "C" int shr_data_to_model_indexed(const float *depth, const int32_t *depth_index, const float *centres,
                                         const float *radii, int N, int J, int H, int W, float *loss_sum,
                                         float *grad_centres, void *stream) {
  if (!depth_index && N > 0) return SHR_EINVAL;
  return launch_d2m(depth, depth_index, centres, radii, N, J, H, W, loss_sum, grad_centres, stream);
}
