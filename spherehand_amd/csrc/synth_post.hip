// synth_post.hip -- the pointwise tail of the synthetic branch (network/util_modules.py:104-122):
//
//   heatmap_paint_kernel   HeatmapRender.forward (mesh/render.py:226-248): per key-point a
//                          Gaussian exp(-0.5 * sigma * ((u - uj)^2 + (v - vj)^2)) over the S x S
//                          heat-map grid and the key-point's depth where the Gaussian exceeds
//                          0.05; the xyz back-projection of InverseOthographicalProjection
//                          (mesh/pointTransformation.py:102-124) rides along.
//   depth_noise_kernel     DepthNoise.forward (network/util_modules.py:46-84): per pixel a source
//                          shift (sigma 0.5 px, rounded, clamped to the image) and Gaussian depth
//                          noise (sigma 0.05) on foreground pixels (< 1.0 in scaled depth).  The
//                          three standard-normal draws per pixel come from ONE torch.randn tensor
//                          [3, B, H, W] (same generator stream as three randn_like calls of the
//                          same total size would consume; parity with the reference is in
//                          distribution only, as for any RNG-dependent step).
//
// Both are forward only (the synthetic branch is detached, util_modules.py:122) and replace
// ~35 elementwise launches per step.
#include "common.h"

namespace shr {

__global__ void __launch_bounds__(256)
heatmap_paint_kernel(const float4 *__restrict__ uvd, int S, float sigma, float uv_scale, float d_scale, float a00,
                     float a03, float a11, float a13, float *__restrict__ uv_hm, float *__restrict__ d_hm,
                     float4 *__restrict__ xyz) {
  const int bj = blockIdx.x;                      // b * J + j
  const float4 p = uvd[bj];
  const int npx = S * S;
  float *uo = uv_hm + (size_t)bj * npx, *dout = d_hm + (size_t)bj * npx;
  for (int i = threadIdx.x; i < npx; i += blockDim.x) {
    const int v = i / S, u = i - v * S;
    const float du = (float)u - p.x, dv = (float)v - p.y;
    const float g = __expf(-0.5f * sigma * (du * du + dv * dv));
    uo[i] = g * uv_scale;
    dout[i] = (g > 0.05f ? p.z : 0.f) * d_scale;
  }
  // inverse camera: rows 0 and 1 of K^-1 are (a00, 0, 0, a03) and (0, a11, 0, a13), rows 2, 3 the identity's
  if (threadIdx.x == 0) xyz[bj] = make_float4(a00 * p.x + a03 * p.w, a11 * p.y + a13 * p.w, p.z, p.w);
}

// Hand3DHeatmapRender.forward (mesh/render.py:274-279) in ONE launch: the key-point's skinning + heat-map camera
// (lbs_project_kernel's arithmetic on its CSR entries: common.h lbs_add_entry / lbs_finish -- the same bits as
// shr_lbs_project followed by shr_heatmap_paint), then the paint above.  One workgroup per (sample, key-point); every
// thread forms the point itself (a dozen FMAs on uniform operands: cheaper than an LDS round trip).
__global__ void __launch_bounds__(256)
heatmap_render_kernel(const float *__restrict__ T, int NB, int J, const int *__restrict__ kstart, const int *__restrict__ kbone,
                      const float4 *__restrict__ kwv, int right_hand, float cx, float cy, float fx, float fy,
                      const float *__restrict__ rand_f, int S, float sigma, float uv_scale, float d_scale, float a00,
                      float a03, float a11, float a13, float *__restrict__ uv_hm, float *__restrict__ d_hm,
                      float4 *__restrict__ xyz) {
  const int bj = blockIdx.x;
  const int b = bj / J, j = bj - b * J;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int e = kstart[j]; e < kstart[j + 1]; e++) lbs_add_entry(acc, T + ((size_t)b * NB + kbone[e]) * 16, kwv[e]);
  const float4 p = lbs_finish(acc, right_hand, 1, cx, cy, fx, fy, rand_f != nullptr, rand_f ? rand_f[b] : 0.f);
  const int npx = S * S;
  float *uo = uv_hm + (size_t)bj * npx, *dout = d_hm + (size_t)bj * npx;
  for (int i = threadIdx.x; i < npx; i += blockDim.x) {
    const int v = i / S, u = i - v * S;
    const float du = (float)u - p.x, dv = (float)v - p.y;
    const float g = __expf(-0.5f * sigma * (du * du + dv * dv));
    uo[i] = g * uv_scale;
    dout[i] = (g > 0.05f ? p.z : 0.f) * d_scale;
  }
  if (threadIdx.x == 0) xyz[bj] = make_float4(a00 * p.x + a03 * p.w, a11 * p.y + a13 * p.w, p.z, p.w);
}

__global__ void __launch_bounds__(256)
depth_noise_kernel(const float *__restrict__ dm, const float *__restrict__ normal3, int B, int H, int W, float sigma_xy,
                   float sigma_z, float *__restrict__ out) {
  const size_t n = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((size_t)H * W));
    const int r = (int)(i - (size_t)b * H * W);
    const int v = r / W, u = r - v * W;
    // (randn * sigma + 0.5).long(): truncation toward zero, as torch's float -> int64 cast
    const int sx = min(max((int)(normal3[i] * sigma_xy + 0.5f) + u, 0), W - 1);
    const int sy = min(max((int)(normal3[n + i] * sigma_xy + 0.5f) + v, 0), H - 1);
    const float z = dm[(size_t)b * H * W + (size_t)sy * W + sx];
    out[i] = z < 1.0f ? z + normal3[2 * n + i] * sigma_z : z;
  }
}

}  // namespace shr

extern "C" int shr_heatmap_paint(const float *uvd, int BJ, int S, float sigma, float uv_scale, float d_scale, float a00,
                                 float a03, float a11, float a13, float *uv_hm, float *d_hm, float *xyz, void *stream) {
  using namespace shr;
  if (BJ == 0) return SHR_OK;
  if (!uvd || !uv_hm || !d_hm || !xyz || BJ < 0 || S <= 0) return SHR_EINVAL;
  if ((((uintptr_t)uvd | (uintptr_t)xyz) & 15u) != 0) return SHR_EINVAL;
  hipLaunchKernelGGL(heatmap_paint_kernel, dim3((unsigned)BJ), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(uvd), S, sigma, uv_scale, d_scale, a00, a03, a11, a13, uv_hm, d_hm,
                     reinterpret_cast<float4 *>(xyz));
  return (int)hipGetLastError();
}

extern "C" int shr_heatmap_render_fwd(const float *T, int B, int NB, int J, const int32_t *kp_start, const int32_t *kp_bone,
                                      const float *kp_wv, int right_hand, float cx, float cy, float fx, float fy,
                                      const float *rand_f, int S, float sigma, float uv_scale, float d_scale, float a00,
                                      float a03, float a11, float a13, float *uv_hm, float *d_hm, float *xyz, void *stream) {
  using namespace shr;
  if (B == 0 || J == 0) return SHR_OK;
  if (!T || !kp_start || !kp_bone || !kp_wv || !uv_hm || !d_hm || !xyz || B < 0 || J < 0 || NB <= 0 || S <= 0) return SHR_EINVAL;
  if ((((uintptr_t)kp_wv | (uintptr_t)xyz) & 15u) != 0) return SHR_EINVAL;
  if ((long long)B * J > (1LL << 30)) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(heatmap_render_kernel, dim3((unsigned)(B * J)), dim3(S * S >= 256 ? 256 : 64), 0, (hipStream_t)stream, T, NB, J,
                     kp_start, kp_bone, reinterpret_cast<const float4 *>(kp_wv), right_hand, cx, cy, fx, fy, rand_f, S, sigma,
                     uv_scale, d_scale, a00, a03, a11, a13, uv_hm, d_hm, reinterpret_cast<float4 *>(xyz));
  return (int)hipGetLastError();
}

extern "C" int shr_depth_noise(const float *depth, const float *normal3, int B, int H, int W, float sigma_xy,
                               float sigma_z, float *out, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!depth || !normal3 || !out || B < 0 || H <= 0 || W <= 0 || depth == out) return SHR_EINVAL;
  const size_t n = (size_t)B * H * W;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(depth_noise_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, normal3, B, H, W,
                     sigma_xy, sigma_z, out);
  return (int)hipGetLastError();
}
