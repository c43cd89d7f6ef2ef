// soft_argmax.hip -- RecoverXYZCoordinateFromHeatmap (network/util_modules.py:164-201), forward and
// backward, on the network's raw output hm[N][2J][h][w] (uv heat-maps = channels 0..J-1, depth
// heat-maps = channels J..2J-1, any channel/pixel strides: NCHW or channels-last):
//   p = softmax(20 * uv) over the map, u = sum p * x, v = sum p * y,
//   d = sum d_hm * relu(uv) / (sum relu(uv) + 1e-5),
//   xyz = ((u - cx) / fx, (v - cy) / fy, d * depth_scale_inv).
// The torch formulation is ~15 elementwise / reduction launches forward and ~30 backward on
// 123 x 41 maps of 16 x 16 pixels; here one workgroup per sample stages its 2J maps in LDS
// (channel-major: the lanes of a wave read consecutive pixels), a wave per key-point reduces
// with DPP, and the backward overwrites the staged values with their gradients and copies
// them out in the input's layout.  Deterministic (fixed reduction order).
#include "common.h"

namespace shr {

// u, v are kept RELATIVE to the pixel (xm, ym) of the largest logit: u = xm + cu.  A sharply peaked map has
// cu ~ 0, and the backward's x - u = (x - xm) - cu keeps its relative precision there (computed against the
// absolute u, an ulp of u -- 1e-6 px -- times 20 p du was up to 4e-4 of the largest gradient entry).
struct SoftArgmaxStats { float m, z, cu, cv, r, d; int xm, ym; };

// wave-wide stats of key-point j from the LDS slab (uvm = its uv map, dm = its depth map, npx pixels)
__device__ __forceinline__ SoftArgmaxStats soft_argmax_stats(const float *uvm, const float *dm, int npx, int w, int lane) {
  float lm = -__builtin_inff();
  int lp = 0;
  for (int p = lane; p < npx; p += 64) {
    const float a = uvm[p];
    if (a > lm) { lm = a; lp = p; }
  }
  const float m = wave_minmax_all<false>(lm);
  // the first lane that holds the maximum names the anchor pixel (any pixel would do: it only centres the sums)
  const unsigned long long holders = __ballot(lm == m);
  const int pm = holders ? __builtin_amdgcn_readlane(lp, __builtin_amdgcn_readfirstlane(__builtin_ctzll(holders))) : 0;
  const int ym = pm / w, xm = pm - ym * w;
  float z = 0.f, su = 0.f, sv = 0.f, r = 0.f, sd = 0.f;
  for (int p = lane; p < npx; p += 64) {
    const float a = uvm[p];
    const float e = __expf(20.0f * (a - m));
    const int y = p / w, x = p - y * w;
    z += e; su += e * (float)(x - xm); sv += e * (float)(y - ym);
    const float rl = fmaxf(a, 0.f);
    r += rl; sd += dm[p] * rl;
  }
  SoftArgmaxStats s;
  s.m = m; s.xm = xm; s.ym = ym;
  s.z = readlane_f(wave_sum_lane63(z), 63);
  s.cu = readlane_f(wave_sum_lane63(su), 63) / s.z;
  s.cv = readlane_f(wave_sum_lane63(sv), 63) / s.z;
  s.r = readlane_f(wave_sum_lane63(r), 63) + 1e-5f;
  s.d = readlane_f(wave_sum_lane63(sd), 63) / s.r;
  return s;
}

template <bool BACKWARD>
__global__ void __launch_bounds__(1024)
soft_argmax_kernel(const float *__restrict__ hm, long long sn, long long sc, long long sp, int J, int h, int w,
                   float cx, float cy, float inv_fx, float inv_fy, float d_scale, float *__restrict__ xyz,
                   const float *__restrict__ grad_xyz, float *__restrict__ grad_hm) {
  extern __shared__ float slab[];                 // [2J][npx]
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwaves = nthr >> 6;   // 16 waves: a sample's 41 key-points in three rounds (4 waves: 31 / 50 us forward / backward for 123 samples, half of the CUs idle)
  const int npx = h * w, C = 2 * J;
  const float *src = hm + (size_t)n * sn;
  const bool nhwc = sc == 1;
  // stage (coalesced along the input's fast axis)
  for (int e = tid; e < C * npx; e += nthr) {
    int c, p;
    if (nhwc) { p = e / C; c = e - p * C; } else { c = e / npx; p = e - c * npx; }
    slab[c * npx + p] = src[(size_t)c * sc + (size_t)p * sp];
  }
  __syncthreads();
  for (int j = wave; j < J; j += nwaves) {
    float *uvm = slab + (size_t)j * npx, *dm = slab + (size_t)(J + j) * npx;
    const SoftArgmaxStats s = soft_argmax_stats(uvm, dm, npx, w, lane);
    if (!BACKWARD) {
      if (lane == 0) {
        float *o = xyz + ((size_t)n * J + j) * 3;
        o[0] = (((float)s.xm - cx) + s.cu) * inv_fx; o[1] = (((float)s.ym - cy) + s.cv) * inv_fy; o[2] = s.d * d_scale;
      }
    } else {
      const float *g = grad_xyz + ((size_t)n * J + j) * 3;
      const float du = g[0] * inv_fx, dv = g[1] * inv_fy, dd = g[2] * d_scale;
      const float inv_z = 1.0f / s.z, inv_r = 1.0f / s.r;
      for (int p = lane; p < npx; p += 64) {
        const float a = uvm[p], dval = dm[p];
        const float pr = __expf(20.0f * (a - s.m)) * inv_z;
        const int y = p / w, x = p - y * w;
        float ga = 20.0f * pr * (du * ((float)(x - s.xm) - s.cu) + dv * ((float)(y - s.ym) - s.cv));
        if (a > 0.f) ga += dd * (dval - s.d) * inv_r;
        uvm[p] = ga;
        dm[p] = dd * fmaxf(a, 0.f) * inv_r;
      }
    }
  }
  if (BACKWARD) {
    __syncthreads();
    float *dst = grad_hm + (size_t)n * sn;
    for (int e = tid; e < C * npx; e += nthr) {
      int c, p;
      if (nhwc) { p = e / C; c = e - p * C; } else { c = e / npx; p = e - c * npx; }
      dst[(size_t)c * sc + (size_t)p * sp] = slab[c * npx + p];
    }
  }
}

}  // namespace shr

extern "C" int shr_soft_argmax_supported(int J, int h, int w) {
  return (J > 0 && h > 0 && w > 0 && (long long)2 * J * h * w * 4 <= 150 * 1024) ? 1 : 0;
}

static int soft_argmax_launch(bool backward, const float *hm, long long sn, long long sc, long long sp, int N, int J,
                              int h, int w, float cx, float cy, float fx, float fy, float d_scale, float *xyz,
                              const float *grad_xyz, float *grad_hm, void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!hm || N < 0 || fx == 0.f || fy == 0.f || !shr_soft_argmax_supported(J, h, w)) return SHR_EINVAL;
  if (backward ? (!grad_xyz || !grad_hm) : !xyz) return SHR_EINVAL;
  const size_t lds = (size_t)2 * J * h * w * 4;
  // (the attribute is per device: one flag per device ordinal and direction)
  static bool done_dev[64][2] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = -1;
  if (d < 0 || !done_dev[d][backward]) {
    const hipError_t e = backward ? hipFuncSetAttribute(reinterpret_cast<const void *>(soft_argmax_kernel<true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                                  : hipFuncSetAttribute(reinterpret_cast<const void *>(soft_argmax_kernel<false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (d >= 0) done_dev[d][backward] = true;
  }
  if (backward)
    hipLaunchKernelGGL(soft_argmax_kernel<true>, dim3((unsigned)N), dim3(1024), lds, (hipStream_t)stream, hm, sn, sc, sp, J,
                       h, w, cx, cy, 1.0f / fx, 1.0f / fy, d_scale, xyz, grad_xyz, grad_hm);
  else
    hipLaunchKernelGGL(soft_argmax_kernel<false>, dim3((unsigned)N), dim3(1024), lds, (hipStream_t)stream, hm, sn, sc, sp, J,
                       h, w, cx, cy, 1.0f / fx, 1.0f / fy, d_scale, xyz, grad_xyz, grad_hm);
  return (int)hipGetLastError();
}

extern "C" int shr_soft_argmax_fwd(const float *hm, long long stride_n, long long stride_c, long long stride_px, int N, int J,
                                   int h, int w, float cx, float cy, float fx, float fy, float depth_scale_inv, float *xyz,
                                   void *stream) {
  return soft_argmax_launch(false, hm, stride_n, stride_c, stride_px, N, J, h, w, cx, cy, fx, fy, depth_scale_inv, xyz, nullptr,
                            nullptr, stream);
}

extern "C" int shr_soft_argmax_bwd(const float *hm, long long stride_n, long long stride_c, long long stride_px, int N, int J,
                                   int h, int w, float cx, float cy, float fx, float fy, float depth_scale_inv,
                                   const float *grad_xyz, float *grad_hm, void *stream) {
  return soft_argmax_launch(true, hm, stride_n, stride_c, stride_px, N, J, h, w, cx, cy, fx, fy, depth_scale_inv, nullptr,
                            grad_xyz, grad_hm, stream);
}
