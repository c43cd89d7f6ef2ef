// group_norm_relu.hip -- y = relu(GroupNorm(x)) on channels-last (NHWC) activations, forward
// and backward, for the hourglass network's pre-activation blocks.
//
// Replaces the pairs `F.relu(self.bnK(x))` of network/hourglass.py:28-31 (35 per forward
// pass).  The convolutions run channels-last on MIOpen; torch's GroupNorm kernels are NCHW,
// so every pair costs two layout copies, the normalisation, the ReLU and, backwards, three
// kernels more (measured on MI355X, 123 crops: 9.5 ms of a 16.6 ms training step).  Here one
// kernel per direction reads and writes NHWC directly.
//
// One workgroup (256 threads) per (sample, block of 32 channels): a thread owns 4 consecutive
// channels (one 16-byte access) of every 32nd pixel, so a wave touches 8 pixels x 128 bytes
// per access.  A 32-channel block holds whole groups (C/G in {4, 8, 16, 32}).  The sample's
// slab (<= 256 KB) is read up to three times; passes two and three hit the L2.  All sums are
// reduced in a fixed order (shuffles inside a wave, waves in order): deterministic.
#include "common.h"

namespace shr {

constexpr int kGnThreads = 256;
constexpr int kGnBlockC = 32;

// sum over the 8 pixel slots of a wave (lanes with equal lane & 7), then over the cpg/4
// neighbouring channel-lanes of a group, then over the 4 waves: every lane ends up with the
// total of ITS group
__device__ __forceinline__ float gn_group_total(float v, int cpg, float (*s_w)[8], int wave, int lane) {
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  if (cpg >= 8) v += __shfl_xor(v, 1);
  if (cpg >= 16) v += __shfl_xor(v, 2);
  if (cpg >= 32) v += __shfl_xor(v, 4);
  __syncthreads();   // s_w free
  if (lane < 8) s_w[wave][lane] = v;
  __syncthreads();
  const int k = lane & 7;
  return ((s_w[0][k] + s_w[1][k]) + s_w[2][k]) + s_w[3][k];
}

// per-CHANNEL totals (no group combine): lane k (< 8) of every wave returns the totals of
// channels 4k .. 4k+3
__device__ __forceinline__ float4 gn_channel_total(float4 v, float4 (*s_w4)[8], int wave, int lane) {
#pragma unroll
  for (int m = 8; m <= 32; m <<= 1) {
    v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
  }
  __syncthreads();
  if (lane < 8) s_w4[wave][lane] = v;
  __syncthreads();
  const int k = lane & 7;
  float4 t = s_w4[0][k];
#pragma unroll
  for (int w = 1; w < 4; w++) { const float4 a = s_w4[w][k]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
  return t;
}

__global__ void __launch_bounds__(kGnThreads)
group_norm_relu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ pre_bias,
                           const float *__restrict__ gamma, const float *__restrict__ beta,
                           int C, int HW, int G, float eps, float *__restrict__ y, float *__restrict__ mean_out,
                           float *__restrict__ rstd_out) {
  __shared__ float s_w[4][8];
  const int n = blockIdx.y, c0 = blockIdx.x * kGnBlockC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = tid & 7, p0 = tid >> 3;          // channel-lane (4 channels), first pixel
  const int cpg = C / G;
  const int c = c0 + 4 * k, g = c / cpg;
  const float4 *xs = reinterpret_cast<const float4 *>(x + (size_t)n * HW * C + c);
  float4 *ys = reinterpret_cast<float4 *>(y + (size_t)n * HW * C + c);
  const int stride4 = C >> 2;                    // float4 per pixel
  const float inv_m = 1.0f / (float)(cpg * HW);
  // the producing convolution's bias, added on the fly (x' = x + pre_bias[c]): the convolution then runs without
  // its bias-add kernel, and the backward returns the bias gradient with dx (no separate reduction)
  const float4 pb = pre_bias ? *reinterpret_cast<const float4 *>(pre_bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);

  float s = 0.f;
  for (int p = p0; p < HW; p += 32) {
    float4 v = xs[(size_t)p * stride4];
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = gn_group_total(s, cpg, s_w, wave, lane) * inv_m;
  float q = 0.f;
  for (int p = p0; p < HW; p += 32) {
    float4 v = xs[(size_t)p * stride4];
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
    q += (a * a + b * b) + (cc * cc + d * d);
  }
  const float var = gn_group_total(q, cpg, s_w, wave, lane) * inv_m;
  const float rstd = 1.0f / __builtin_sqrtf(var + eps);
  if (tid < 8 && (c % cpg) == 0) { mean_out[(size_t)n * G + g] = mean; rstd_out[(size_t)n * G + g] = rstd; }
  const float4 ga = *reinterpret_cast<const float4 *>(gamma + c), be = *reinterpret_cast<const float4 *>(beta + c);
  const float4 sc = make_float4(ga.x * rstd, ga.y * rstd, ga.z * rstd, ga.w * rstd);
  for (int p = p0; p < HW; p += 32) {
    float4 v = xs[(size_t)p * stride4];
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    float4 o;
    o.x = fmaxf((v.x - mean) * sc.x + be.x, 0.f);
    o.y = fmaxf((v.y - mean) * sc.y + be.y, 0.f);
    o.z = fmaxf((v.z - mean) * sc.z + be.z, 0.f);
    o.w = fmaxf((v.w - mean) * sc.w + be.w, 0.f);
    ys[(size_t)p * stride4] = o;
  }
}

// dx = rstd * (dy' * gamma - (s1 + xhat * s2) / M) with dy' = dy where the output was positive,
// s1 = sum_group dy' * gamma, s2 = sum_group dy' * gamma * xhat; per-sample partials of
// dgamma = sum dy' * xhat and dbeta = sum dy' (the caller sums them over the samples).
__global__ void __launch_bounds__(kGnThreads)
group_norm_relu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ pre_bias,
                           const float *__restrict__ dy, const float *__restrict__ gamma,
                           const float *__restrict__ beta, const float *__restrict__ mean_in,
                           const float *__restrict__ rstd_in, int C, int HW, int G, float *__restrict__ dx,
                           float *__restrict__ dgamma_part, float *__restrict__ dbeta_part,
                           float *__restrict__ dpre_part) {
  __shared__ float4 s_w4[4][8];
  const int n = blockIdx.y, c0 = blockIdx.x * kGnBlockC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = tid & 7, p0 = tid >> 3;
  const int cpg = C / G;
  const int c = c0 + 4 * k, g = c / cpg;
  const size_t base = (size_t)n * HW * C + c;
  const float4 *xs = reinterpret_cast<const float4 *>(x + base);
  const float4 *ds = reinterpret_cast<const float4 *>(dy + base);
  float4 *os = reinterpret_cast<float4 *>(dx + base);
  const int stride4 = C >> 2;
  const float mean = mean_in[(size_t)n * G + g], rstd = rstd_in[(size_t)n * G + g];
  const float4 ga = *reinterpret_cast<const float4 *>(gamma + c), be = *reinterpret_cast<const float4 *>(beta + c);
  const float4 pb = pre_bias ? *reinterpret_cast<const float4 *>(pre_bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);

  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = p0; p < HW; p += 32) {
    float4 v = xs[(size_t)p * stride4];
    const float4 d = ds[(size_t)p * stride4];
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    const float h0 = (v.x - mean) * rstd, h1 = (v.y - mean) * rstd, h2 = (v.z - mean) * rstd, h3 = (v.w - mean) * rstd;
    const float d0 = (h0 * ga.x + be.x > 0.f) ? d.x : 0.f, d1 = (h1 * ga.y + be.y > 0.f) ? d.y : 0.f;
    const float d2 = (h2 * ga.z + be.z > 0.f) ? d.z : 0.f, d3 = (h3 * ga.w + be.w > 0.f) ? d.w : 0.f;
    sa.x += d0; sa.y += d1; sa.z += d2; sa.w += d3;
    sb.x += d0 * h0; sb.y += d1 * h1; sb.z += d2 * h2; sb.w += d3 * h3;
  }
  const float4 A = gn_channel_total(sa, s_w4, wave, lane);   // sum dy'          per channel
  const float4 B = gn_channel_total(sb, s_w4, wave, lane);   // sum dy' * xhat   per channel
  if (tid < 8) {
    *reinterpret_cast<float4 *>(dbeta_part + (size_t)n * C + c) = A;
    *reinterpret_cast<float4 *>(dgamma_part + (size_t)n * C + c) = B;
  }
  // group sums from the channel totals (every lane holds its channel-lane's totals)
  float t1 = (ga.x * A.x + ga.y * A.y) + (ga.z * A.z + ga.w * A.w);
  float t2 = (ga.x * B.x + ga.y * B.y) + (ga.z * B.z + ga.w * B.w);
  if (cpg >= 8) { t1 += __shfl_xor(t1, 1); t2 += __shfl_xor(t2, 1); }
  if (cpg >= 16) { t1 += __shfl_xor(t1, 2); t2 += __shfl_xor(t2, 2); }
  if (cpg >= 32) { t1 += __shfl_xor(t1, 4); t2 += __shfl_xor(t2, 4); }
  const float inv_m = 1.0f / (float)(cpg * HW);
  const float m1 = t1 * inv_m, m2 = t2 * inv_m;
  float4 so = make_float4(0.f, 0.f, 0.f, 0.f);     // sum of dx per channel = the producing convolution's bias gradient
  for (int p = p0; p < HW; p += 32) {
    float4 v = xs[(size_t)p * stride4];
    const float4 d = ds[(size_t)p * stride4];
    v.x += pb.x; v.y += pb.y; v.z += pb.z; v.w += pb.w;
    const float h0 = (v.x - mean) * rstd, h1 = (v.y - mean) * rstd, h2 = (v.z - mean) * rstd, h3 = (v.w - mean) * rstd;
    const float d0 = (h0 * ga.x + be.x > 0.f) ? d.x : 0.f, d1 = (h1 * ga.y + be.y > 0.f) ? d.y : 0.f;
    const float d2 = (h2 * ga.z + be.z > 0.f) ? d.z : 0.f, d3 = (h3 * ga.w + be.w > 0.f) ? d.w : 0.f;
    float4 o;
    o.x = rstd * (d0 * ga.x - (m1 + h0 * m2));
    o.y = rstd * (d1 * ga.y - (m1 + h1 * m2));
    o.z = rstd * (d2 * ga.z - (m1 + h2 * m2));
    o.w = rstd * (d3 * ga.w - (m1 + h3 * m2));
    os[(size_t)p * stride4] = o;
    so.x += o.x; so.y += o.y; so.z += o.z; so.w += o.w;
  }
  if (dpre_part) {                                   // (uniform)
    const float4 D = gn_channel_total(so, s_w4, wave, lane);
    if (tid < 8) *reinterpret_cast<float4 *>(dpre_part + (size_t)n * C + c) = D;
  }
}

// dgamma[c] = sum_n dgamma_part[n][c] (same for dbeta).  32 channels x 8 sample-lanes per
// workgroup: lane s sums the samples n = s (mod 8), the eight partial sums are added in order:
// deterministic.
__global__ void __launch_bounds__(256)
group_norm_param_grad_kernel(const float *__restrict__ dg_part, const float *__restrict__ db_part,
                             const float *__restrict__ dp_part, int N, int C,
                             float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dpre) {
  __shared__ float s_a[8][32], s_b[8][32], s_p[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a = 0.f, b = 0.f, q = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int n = sl; n < N; n += 8) {
      a += dg_part[(size_t)n * C + c]; b += db_part[(size_t)n * C + c];
      if (dp_part) q += dp_part[(size_t)n * C + c];
    }
  }
  s_a[sl][cl] = a;
  s_b[sl][cl] = b;
  s_p[sl][cl] = q;
  __syncthreads();
  if (sl == 0 && c < C) {
    float ta = 0.f, tb = 0.f, tp = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) { ta += s_a[k][cl]; tb += s_b[k][cl]; tp += s_p[k][cl]; }
    dgamma[c] = ta;
    dbeta[c] = tb;
    if (dpre) dpre[c] = tp;
  }
}

bool gn_supported(int C, int G) {
  if (C <= 0 || G <= 0 || C % G || C % kGnBlockC) return false;
  const int cpg = C / G;
  return cpg % 4 == 0 && cpg <= kGnBlockC && kGnBlockC % cpg == 0;
}

}  // namespace shr

extern "C" int shr_group_norm_relu_supported(int C, int G) { return shr::gn_supported(C, G) ? 1 : 0; }

extern "C" int shr_group_norm_relu_fwd(const float *x, const float *pre_bias, const float *gamma, const float *beta, int N,
                                       int C, int HW, int G, float eps, float *y, float *mean, float *rstd,
                                       void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || N < 0 || HW <= 0) return SHR_EINVAL;
  if (!gn_supported(C, G) ||
      ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)pre_bias)) & 15u))
    return SHR_EINVAL;
  if (N > 65535) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(group_norm_relu_fwd_kernel, dim3((unsigned)(C / kGnBlockC), (unsigned)N), dim3(kGnThreads), 0,
                     (hipStream_t)stream, x, pre_bias, gamma, beta, C, HW, G, eps, y, mean, rstd);
  return (int)hipGetLastError();
}

extern "C" int shr_group_norm_relu_bwd(const float *x, const float *pre_bias, const float *dy, const float *gamma,
                                       const float *beta, const float *mean, const float *rstd, int N, int C, int HW,
                                       int G, float *dx, float *dgamma_partial, float *dbeta_partial,
                                       float *dpre_partial, float *dgamma, float *dbeta, float *dpre, void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!x || !dy || !gamma || !beta || !mean || !rstd || !dx || !dgamma_partial || !dbeta_partial || N < 0 || HW <= 0)
    return SHR_EINVAL;
  if (!gn_supported(C, G) ||
      ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)dgamma_partial |
         (uintptr_t)dbeta_partial | (uintptr_t)pre_bias | (uintptr_t)dpre_partial)) & 15u))
    return SHR_EINVAL;
  if ((pre_bias != nullptr) != (dpre_partial != nullptr)) return SHR_EINVAL;
  if (N > 65535) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(group_norm_relu_bwd_kernel, dim3((unsigned)(C / kGnBlockC), (unsigned)N), dim3(kGnThreads), 0,
                     (hipStream_t)stream, x, pre_bias, dy, gamma, beta, mean, rstd, C, HW, G, dx, dgamma_partial,
                     dbeta_partial, dpre_partial);
  if (dgamma && dbeta)
    hipLaunchKernelGGL(group_norm_param_grad_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       dgamma_partial, dbeta_partial, dpre_partial, N, C, dgamma, dbeta, dpre);
  return (int)hipGetLastError();
}
