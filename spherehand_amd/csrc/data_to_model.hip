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
//   3. each queue entry searches the J spheres (LDS broadcast reads), adds its
//      clamped distance, leaves (owner, gradient vector) in the queue;
//   4. waves take spheres round-robin and sum their owners' vectors over the
//      queue in lane order + one DPP wave sum: deterministic, no atomics.
// One workgroup per crop.  HBM: reads 4*H*W + 12*J + 4*J bytes per crop.

#include "common.h"

namespace shr {

constexpr int kD2mThreads = 1024;
constexpr int kD2mChunk = 4 * kD2mThreads;  // pixels per chunk
constexpr int kD2mSlots = SHR_MAX_SPHERES / (kD2mThreads / 64);

struct QEntry { float a, b, c; int d; };  // phase 2: (xg, yg, z, -) ; phase 3: (gx, gy, gz, owner)

template <bool WANT_GRAD>
__global__ void __launch_bounds__(kD2mThreads)
data_to_model_kernel(const float *__restrict__ depth, const float *__restrict__ centres,
                     const float *__restrict__ radii, int J, int H, int W, float *__restrict__ loss_sum,
                     float *__restrict__ grad_centres) {
  __shared__ float4 s_c[SHR_MAX_SPHERES];     // (cx, cy, cz, r)
  __shared__ int s_wave_cnt[kD2mThreads / 64];
  __shared__ float s_wave_loss[kD2mThreads / 64];
  __shared__ QEntry s_q[kD2mChunk];           // 64 KB

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
  float acc[kD2mSlots][3];
#pragma unroll
  for (int t = 0; t < kD2mSlots; t++) acc[t][0] = acc[t][1] = acc[t][2] = 0.f;

  for (int base = 0; base < npix; base += kD2mChunk) {
    // ---- 1. load 4 pixels, flag foreground ------------------------------------------
    const int p0 = base + 4 * tid;
    float z[4] = {100.f, 100.f, 100.f, 100.f};
    if (row4) {
      if (p0 < npix) {
        const float4 t = *reinterpret_cast<const float4 *>(dm + p0);
        z[0] = t.x; z[1] = t.y; z[2] = t.z; z[3] = t.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (p0 + k < npix) z[k] = dm[p0 + k];
    }
    int cnt = 0;
    bool fg[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      fg[k] = (p0 + k < npix) && !(z[k] > 99.0f);  // mesh/render.py:138 background = d > 99
      cnt += fg[k];
    }
    // deterministic exclusive scan of cnt over the workgroup
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (base > 0) __syncthreads();  // previous chunk's queue fully consumed
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    int offset = incl - cnt, total = 0;
    for (int w = 0; w < kD2mThreads / 64; w++) {
      const int c = s_wave_cnt[w];
      if (w < wave) offset += c;
      total += c;
    }
    // ---- 2. compact ---------------------------------------------------------------------
    {
      int v = p0 / W, u = p0 - v * W;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (fg[k]) {
          QEntry e;
          e.a = axis_coord(ax, u);
          e.b = axis_coord(ay, v);
          e.c = z[k];
          e.d = 0;
          s_q[offset++] = e;
        }
        if (++u == W) { u = 0; ++v; }
      }
    }
    __syncthreads();
    // ---- 3. nearest-surface search per foreground pixel ------------------------------
    for (int i = tid; i < total; i += kD2mThreads) {
      const QEntry e = s_q[i];
      float best = 0.f;
      int bj = 0;
      for (int j = 0; j < J; j++) {
        const float4 c = s_c[j];
        const float dx = e.a - c.x, dy = e.b - c.y, dz = e.c - c.z;
        const float dist = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);  // <= 1 ulp: loss is continuous
        const float a = fabsf(dist - c.w);
        if (j == 0 || a < best || a != a) {
          if (j == 0 || best == best) { best = a; bj = j; }
        }
      }
      loss += fminf(fmaxf(best, 0.f), 50.f);
      if (WANT_GRAD) {
        const float4 c = s_c[bj];
        const float dx = e.a - c.x, dy = e.b - c.y, dz = e.c - c.z;
        const float dist = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
        const float t = dist - c.w;
        const float sgn = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
        const bool live = (best <= 50.f) && (dist != 0.f) && (sgn != 0.f);
        const float k = live ? -(sgn / dist) : 0.f;
        QEntry g;
        g.a = k * dx; g.b = k * dy; g.c = k * dz; g.d = live ? bj : -1;
        s_q[i] = g;
      }
    }
    // ---- 4. per-sphere sums over the queue ---------------------------------------------
    if (WANT_GRAD) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < kD2mSlots; t++) {
        const int j = wave + t * (kD2mThreads / 64);
        if (j >= J) continue;
        for (int i = lane; i < total; i += 64) {
          const QEntry g = s_q[i];
          const bool mine = g.d == j;
          acc[t][0] += mine ? g.a : 0.f;
          acc[t][1] += mine ? g.b : 0.f;
          acc[t][2] += mine ? g.c : 0.f;
        }
      }
    }
  }

  // ---- reductions ---------------------------------------------------------------------------
  loss = wave_sum_lane63(loss);
  if (lane == 63) s_wave_loss[wave] = loss;
  if (WANT_GRAD) {
#pragma unroll
    for (int t = 0; t < kD2mSlots; t++) {
      const int j = wave + t * (kD2mThreads / 64);
      if (j >= J) continue;
      const float gx = wave_sum_lane63(acc[t][0]);
      const float gy = wave_sum_lane63(acc[t][1]);
      const float gz = wave_sum_lane63(acc[t][2]);
      if (lane == 63) {
        float *o = grad_centres + ((size_t)n * J + j) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < kD2mThreads / 64; w++) t += s_wave_loss[w];
    loss_sum[n] = t;
  }
}

}  // namespace shr

extern "C" int shr_data_to_model(const float *depth, const float *centres, const float *radii, int N, int J, int H,
                                 int W, float *loss_sum, float *grad_centres, void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!depth || !centres || !radii || !loss_sum || N < 0 || J <= 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if (J > SHR_MAX_SPHERES || (long long)H * W > (1LL << 30)) return SHR_ETOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  if (grad_centres)
    hipLaunchKernelGGL(data_to_model_kernel<true>, dim3((unsigned)N), dim3(kD2mThreads), 0, s, depth, centres, radii,
                       J, H, W, loss_sum, grad_centres);
  else
    hipLaunchKernelGGL(data_to_model_kernel<false>, dim3((unsigned)N), dim3(kD2mThreads), 0, s, depth, centres, radii,
                       J, H, W, loss_sum, grad_centres);
  return (int)hipGetLastError();
}
