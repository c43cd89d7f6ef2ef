#include "../spherehand_amd/csrc/common.h"
namespace shr {
constexpr int kD2mThreads = 1024;
constexpr int kD2mPix = 16;                           // pixels per thread per chunk (4 x 16-byte loads in flight)
constexpr int kD2mChunk = kD2mPix * kD2mThreads;      // 16384 px: a whole 128x128 crop
constexpr int kD2mQueue = 4096;                       // queue entries (64 KB); denser chunks take extra passes

struct QEntry { float a, b, c; int d; };  // phase 2: (xg, yg, z, -) ; phase 3: (gx, gy, gz, owner)

template <bool WANT_GRAD>
__global__ void __launch_bounds__(kD2mThreads)
exp_d2m(const float *__restrict__ depth, const float *__restrict__ centres,
                     const float *__restrict__ radii, int J, int H, int W, float *__restrict__ loss_sum,
                     float *__restrict__ grad_centres, int mode) {
  __shared__ float4 s_c[SHR_MAX_SPHERES];     // (cx, cy, cz, r)
  __shared__ int s_wave_cnt[kD2mThreads / 64];
  __shared__ float s_wave_loss[kD2mThreads / 64];
  __shared__ QEntry s_q[kD2mQueue];           // 64 KB
  __shared__ float4 s_part[(kD2mThreads / 64) * SHR_MAX_SPHERES];   // [wave][sphere] gradient partials

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < J) {
    const float *c = centres + ((size_t)n * J + tid) * 3;
    s_c[tid] = make_float4(c[0], c[1], c[2], radii[tid]);
  }
  const float *dm = depth + (size_t)n * H * W;
  const Axis ax = make_axis(W), ay = make_axis(H);
  const int npix = H * W;
  const bool row4 = (W % 4 == 0) && is_aligned16(dm);

  float loss = 0.f;
  if (WANT_GRAD) s_part[tid] = make_float4(0.f, 0.f, 0.f, 0.f);   // 1024 = 16 waves x 64 spheres

  for (int base = 0; base < npix; base += kD2mChunk) {
    // ---- 1. load 16 consecutive pixels (four 16-byte loads, all issued before the first
    // use), flag foreground.  Consecutive pixels per thread keep the queue in scan order,
    // so a wave's 64 entries are neighbours and their bounding box is tight (step 3).
    float z[kD2mPix];
    int cnt = 0;
    unsigned fgmask = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int p0 = base + kD2mPix * tid + 4 * g;
      float4 t = make_float4(100.f, 100.f, 100.f, 100.f);
      if (row4) {
        if (p0 < npix) t = *reinterpret_cast<const float4 *>(dm + p0);
      } else {
        if (p0 + 0 < npix) t.x = dm[p0 + 0];
        if (p0 + 1 < npix) t.y = dm[p0 + 1];
        if (p0 + 2 < npix) t.z = dm[p0 + 2];
        if (p0 + 3 < npix) t.w = dm[p0 + 3];
      }
      z[4 * g] = t.x; z[4 * g + 1] = t.y; z[4 * g + 2] = t.z; z[4 * g + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < kD2mPix; k++) {
      const int p = base + kD2mPix * tid + k;
      const bool fg = (p < npix) && !(z[k] > 99.0f);  // mesh/render.py:138 background = d > 99
      fgmask |= (unsigned)fg << k;
      cnt += fg;
    }
    // deterministic exclusive scan of cnt over the workgroup (queue order = thread order)
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (base > 0) __syncthreads();  // previous chunk's queue fully consumed
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
      const int pfirst = base + kD2mPix * tid;
      int v = pfirst / W, u = pfirst - v * W;   // one division per thread, then incremental
#pragma unroll
      for (int k = 0; k < kD2mPix; k++, u = (u + 1 == W) ? 0 : u + 1, v += (u == 0)) {
        if ((fgmask >> k) & 1u) {
          if (slot >= 0 && slot < kD2mQueue) {
            QEntry e;
            e.a = axis_coord(ax, u);
            e.b = axis_coord(ay, v);
            e.c = z[k];
            e.d = 0;
            if (!(mode & 8)) s_q[slot] = e;
          }
          ++slot;
        }
      }
    }
    __syncthreads();
    // ---- 3. nearest-surface search per foreground pixel ------------------------------
    // A wave takes 64 consecutive queue entries (neighbouring pixels in scan order).  With
    // lanes = spheres it bounds a_j = | ||p - c_j|| - r_j | over the points' bounding box:
    // a_j in [lb_j, ub_j]; every point's minimum is <= U = min_j ub_j, so a sphere with
    // lb_j > U cannot be nearest for any of the 64 points (strictly, so index ties are
    // unaffected).  Then lanes = points walk the surviving candidates only (typically 4-8
    // of 41).  Bounds carry a rounding slack; a NaN anywhere disables the pruning.
    {
      const float4 cj = lane < J ? s_c[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!(mode & 1)) for (int i0 = wave * 64; i0 < total; i0 += kD2mThreads) {
        const int i = i0 + lane;
        const bool act = i < total;
        QEntry e = s_q[act ? i : i0];
        const float inf = __builtin_inff();
        const float xlo = wave_minmax_all<true>(e.a), xhi = wave_minmax_all<false>(e.a);
        const float ylo = wave_minmax_all<true>(e.b), yhi = wave_minmax_all<false>(e.b);
        const float zlo = wave_minmax_all<true>(e.c), zhi = wave_minmax_all<false>(e.c);
        // lanes = spheres: nearest / farthest distance from the centre to the box
        const float nx = fmaxf(fmaxf(xlo - cj.x, cj.x - xhi), 0.f), fx = fmaxf(fabsf(xlo - cj.x), fabsf(xhi - cj.x));
        const float ny = fmaxf(fmaxf(ylo - cj.y, cj.y - yhi), 0.f), fy = fmaxf(fabsf(ylo - cj.y), fabsf(yhi - cj.y));
        const float nz = fmaxf(fmaxf(zlo - cj.z, cj.z - zhi), 0.f), fz = fmaxf(fabsf(zlo - cj.z), fabsf(zhi - cj.z));
        const float dmin = __builtin_amdgcn_sqrtf((nx * nx + ny * ny) + nz * nz);
        const float dmax = __builtin_amdgcn_sqrtf((fx * fx + fy * fy) + fz * fz);
        const float lb = fmaxf(fmaxf(dmin - cj.w, cj.w - dmax), 0.f);
        const float ub = fmaxf(fabsf(dmin - cj.w), fabsf(dmax - cj.w));
        const float U = wave_minmax_all<true>(lane < J ? ub : inf);
        const bool pruned = (lb * 0.99999f - 1e-3f) > (U * 1.00001f + 1e-3f);   // false on NaN
        unsigned long long cand = __ballot(lane < J && (!pruned || (mode & 2)));
        float best = 0.f;
        int bj = 0;
        bool first = true;
        while (cand) {
          const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(cand));
          cand &= cand - 1;
          const float cx = readlane_f(cj.x, j), cy = readlane_f(cj.y, j), cz = readlane_f(cj.z, j);
          const float cr = readlane_f(cj.w, j);
          const float dx = e.a - cx, dy = e.b - cy, dz = e.c - cz;
          const float dist = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);  // <= 1 ulp: loss is continuous
          const float a = fabsf(dist - cr);
          if (first || ((best == best) && (a < best || a != a))) { best = a; bj = j; }   // torch.min: NaN wins, ties keep first
          first = false;
        }
        if (act) loss += fminf(fmaxf(best, 0.f), 50.f);
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
          unsigned long long todo = (mode & 4) ? 0ull : __ballot(owner >= 0);
          if (mode & 4) loss += gx + gy + gz;
          while (todo) {
            const int j = __builtin_amdgcn_readlane(owner, __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo)));
            const bool mine = owner == j;
            todo &= ~__ballot(mine);
            const float sx = wave_sum_lane63(mine ? gx : 0.f), sy = wave_sum_lane63(mine ? gy : 0.f);
            const float sz = wave_sum_lane63(mine ? gz : 0.f);
            if (lane == 63) {
              float4 t = s_part[wave * SHR_MAX_SPHERES + j];
              t.x += sx; t.y += sy; t.z += sz;
              s_part[wave * SHR_MAX_SPHERES + j] = t;
            }
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

}
extern "C" int exp_d2m_launch(const float *depth, const float *centres, const float *radii, int N, int J, int H, int W,
                              float *loss_sum, float *grad, int mode, void *stream) {
  hipLaunchKernelGGL(shr::exp_d2m<true>, dim3(N), dim3(1024), 0, (hipStream_t)stream, depth, centres, radii, J, H, W, loss_sum, grad, mode);
  return (int)hipGetLastError();
}
