// mesh_depth.hip -- fused DepthRender back end: triangle raster + clamp + bilinear resize
// in one pass, rasterizing ONLY the source pixels the resize reads.
//
// Replaces the chain DepthRasterizationFunction.apply(640, 640, ...) -> clamp(max=100) ->
// F.interpolate(size=(S,S), mode='bilinear', align_corners=False) of
// mesh/render.py:284-287, :310-311 (kernel: mesh/cuda_kernel/depth_rasterization_cuda_kernel.cu:18-113).
//
// The reference rasterizes 640x640 (1.64 MB per crop) and then keeps 1 source pixel in
// 25 (S = 128), 4 in 100 (S = 64) or 16 in 25 (S = 256).  Here each workgroup owns a tile
// of TO x TO OUTPUT pixels; every output pixel has 2 x 2 source "slots" (x0/x1 x y0/y1 of
// ATen's bilinear source index), kept in LDS as order-preserving integer keys.  Lanes =
// faces: each lane sets its face up exactly like the reference kernel (tri_raster.hip:
// cull, sort by x, inverse barycentric matrix), finds the output pixels whose slots fall in
// the face's box, and for those source pixels repeats the reference's per-column span test
// and per-pixel arithmetic verbatim, finishing with a native LDS integer min (order
// independent, hence deterministic).  Epilogue: clamp, ATen's bilinear formula, coalesced
// stores of the S x S result.  HBM traffic per crop: the vertices (162 KB, L2-shared by the
// crop's tiles) + 4*S*S written, instead of >= 3 x 1.64 MB.
//
// Zero-weight slots are not rasterized (0 * finite = 0 contributes nothing); the only
// input on which this differs from the reference chain is a raster value of -inf next to a
// sampled pixel (an exactly zero 1/z denominator), where the reference's 0 * -inf is NaN.
#include "common.h"

namespace shr {

__device__ __forceinline__ uint32_t mkey(float d) {
  const uint32_t b = __float_as_uint(d);
  return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float mkey_inv(uint32_t k) {
  return __uint_as_float(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ int m_cvt_rz_sat(float d) {
  if (d != d) return 0;
  if (d >= 2147483648.0f) return 2147483647;
  if (d <= -2147483648.0f) return (int)0x80000000;
  return (int)d;
}

// ATen area_pixel_compute_source_index (align_corners=False) for output index d:
// src = scale*(d+0.5)-0.5 clamped at 0; i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0.
struct Lin { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lin lin_index(int d, float scale, int in_size) {
  float src = scale * ((float)d + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lin r;
  r.i0 = min((int)src, in_size - 1);
  r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}

template <int TO>
__global__ void __launch_bounds__(1024)
mesh_depth_kernel(const float4 *__restrict__ vertices, const int *__restrict__ faces, int NV, int F, int src,
                  int S, float clamp_max, float *__restrict__ depth) {
  __shared__ uint32_t s_z[2 * TO][2 * TO + 1];   // [2*dy + sy][2*dx + sx], +1: bank spread
  const int b = blockIdx.y;
  const int tiles = (S + TO - 1) / TO;
  const int ty0 = (blockIdx.x / tiles) * TO, tx0 = (blockIdx.x % tiles) * TO;
  const int tid = threadIdx.x;
  const float scale = (float)src / (float)S;

  for (int i = tid; i < 2 * TO * (2 * TO + 1); i += blockDim.x) (&s_z[0][0])[i] = 0x447A0000u ^ 0x80000000u;  // 1000.0f
  __syncthreads();

  // source range of the tile's slots (for the face/tile cull)
  const Lin tlx = lin_index(tx0, scale, src), thx = lin_index(min(tx0 + TO, S) - 1, scale, src);
  const Lin tly = lin_index(ty0, scale, src), thy = lin_index(min(ty0 + TO, S) - 1, scale, src);
  const float inv_scale = (float)S / (float)src;

  for (int f = tid; f < F; f += blockDim.x) {
    float fv[9];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float4 v = vertices[(size_t)b * NV + faces[f * 3 + k]];
      fv[3 * k] = v.x; fv[3 * k + 1] = v.y; fv[3 * k + 2] = v.z;
    }
    // ---- reference set-up (.cu:33-69) --------------------------------------------------
    if ((fv[7] - fv[1]) * (fv[3] - fv[0]) < (fv[4] - fv[1]) * (fv[6] - fv[0])) continue;
    int p0, p2;
    if (fv[0] < fv[3]) { p0 = (fv[6] < fv[0]) ? 2 : 0; p2 = (fv[3] < fv[6]) ? 2 : 1; }
    else               { p0 = (fv[6] < fv[3]) ? 2 : 1; p2 = (fv[0] < fv[6]) ? 2 : 0; }
    int p1 = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) if (p0 != k && p2 != k) p1 = k;
    float p[3][3];
    const int order[3] = {p0, p1, p2};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int o = order[a];
        p[a][d] = (o == 0) ? fv[d] : ((o == 1) ? fv[3 + d] : fv[6 + d]);
      }
    if (p[0][0] == p[2][0]) continue;
    const int xi_min = m_cvt_rz_sat(fmaxf(ceilf(p[0][0]), 0.f));
    const int xi_max = m_cvt_rz_sat(fminf(p[2][0], (float)src - 1.f));
    if (xi_min > xi_max || xi_max < tlx.i0 || xi_min > thx.i1) continue;
    const float ylo = fminf(fminf(p[0][1], p[1][1]), p[2][1]), yhi = fmaxf(fmaxf(p[0][1], p[1][1]), p[2][1]);
    const bool wild = !(fabsf(ylo) < 1e9f) || !(fabsf(yhi) < 1e9f);
    const int r_lo = wild ? 0 : max(0, (int)floorf(ylo) - 1);
    const int r_hi = wild ? src - 1 : min(src - 1, max(0, (int)ceilf(yhi) + 1));
    if (r_hi < tly.i0 || r_lo > thy.i1) continue;
    float fi[9];
    fi[0] = p[1][1] - p[2][1]; fi[1] = p[2][0] - p[1][0]; fi[2] = p[1][0] * p[2][1] - p[2][0] * p[1][1];
    fi[3] = p[2][1] - p[0][1]; fi[4] = p[0][0] - p[2][0]; fi[5] = p[2][0] * p[0][1] - p[0][0] * p[2][1];
    fi[6] = p[0][1] - p[1][1]; fi[7] = p[1][0] - p[0][0]; fi[8] = p[0][0] * p[1][1] - p[1][0] * p[0][1];
    const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1])) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
    for (int k = 0; k < 9; k++) fi[k] = fi[k] / den;

    // output pixels whose slots can fall in [xi_min, xi_max] x [r_lo, r_hi]: the inverse of
    // the source-index map, widened by one pixel; membership is re-checked exactly per slot
    const int dx_lo = max(tx0, (int)floorf(((float)xi_min + 0.5f) * inv_scale - 0.5f) - 1);
    const int dx_hi = min(min(tx0 + TO, S) - 1, (int)ceilf(((float)xi_max + 0.5f) * inv_scale - 0.5f) + 1);
    const int dy_lo = max(ty0, (int)floorf(((float)r_lo + 0.5f) * inv_scale - 0.5f) - 1);
    const int dy_hi = min(min(ty0 + TO, S) - 1, (int)ceilf(((float)r_hi + 0.5f) * inv_scale - 0.5f) + 1);

    for (int dx = dx_lo; dx <= dx_hi; dx++) {
      const Lin lx = lin_index(dx, scale, src);
#pragma unroll
      for (int sx = 0; sx < 2; sx++) {
        const int xi = sx ? lx.i1 : lx.i0;
        if ((sx ? lx.l1 : lx.l0) == 0.f || xi < xi_min || xi > xi_max) continue;
        // ---- column span (.cu:72-90) -------------------------------------------------
        const float xf = (float)xi;
        float yi1;
        if (xf <= p[1][0]) {
          if (p[1][0] - p[0][0] != 0.f) yi1 = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (xf - p[0][0]) + p[0][1];
          else yi1 = p[1][1];
        } else {
          if (p[2][0] - p[1][0] != 0.f) yi1 = (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]) * (xf - p[1][0]) + p[1][1];
          else yi1 = p[1][1];
        }
        const float yi2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (xf - p[0][0]) + p[0][1];
        const int yi_min = m_cvt_rz_sat(fmaxf(0.f, ceilf(fminf(yi1, yi2))));
        const int yi_max = m_cvt_rz_sat(fminf(fmaxf(yi1, yi2), (float)src - 1.f));
        for (int dy = dy_lo; dy <= dy_hi; dy++) {
          const Lin ly = lin_index(dy, scale, src);
#pragma unroll
          for (int sy = 0; sy < 2; sy++) {
            const int yi = sy ? ly.i1 : ly.i0;
            if ((sy ? ly.l1 : ly.l0) == 0.f || yi < yi_min || yi > yi_max) continue;
            // ---- pixel (.cu:97-110) ----------------------------------------------------
            const float yf = (float)yi;
            float w[3], w_sum = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              w[k] = (fi[3 * k] * xf + fi[3 * k + 1] * yf) + fi[3 * k + 2];
              w[k] = fminf(fmaxf(w[k], 0.f), 1.f);
              w_sum += w[k];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) w[k] = w[k] / w_sum;
            const float zp = 1.0f / ((w[0] / p[0][2] + w[1] / p[1][2]) + w[2] / p[2][2]);
            if (zp == zp) atomicMin(&s_z[2 * (dy - ty0) + sy][2 * (dx - tx0) + sx], mkey(zp));
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- clamp + bilinear (mesh/render.py:286, :311; ATen upsample_bilinear2d) ---------------
  float *out = depth + (size_t)b * S * S;
  for (int i = tid; i < TO * TO; i += blockDim.x) {
    const int oy = i / TO, ox = i - oy * TO;
    const int y = ty0 + oy, x = tx0 + ox;
    if (y >= S || x >= S) continue;
    const Lin lx = lin_index(x, scale, src), ly = lin_index(y, scale, src);
    float v[2][2];
#pragma unroll
    for (int sy = 0; sy < 2; sy++)
#pragma unroll
      for (int sx = 0; sx < 2; sx++) v[sy][sx] = fminf(mkey_inv(s_z[2 * oy + sy][2 * ox + sx]), clamp_max);
    out[(size_t)y * S + x] = ly.l0 * (lx.l0 * v[0][0] + lx.l1 * v[0][1]) + ly.l1 * (lx.l0 * v[1][0] + lx.l1 * v[1][1]);
  }
}

}  // namespace shr

extern "C" int shr_mesh_depth_fwd(const float *vertices, const int32_t *faces, int B, int NV, int F, int src_size,
                                  int S, float clamp_max, float *depth, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!vertices || (!faces && F > 0) || !depth || B < 0 || NV <= 0 || F < 0 || src_size <= 0 || S <= 0)
    return SHR_EINVAL;
  if (((uintptr_t)vertices & 15u) != 0) return SHR_EINVAL;
  if (B > 65535 || S > 16384 || src_size > (1 << 20) || 2 * S > src_size + 1) return SHR_ETOOLARGE;  // down-sampling only
  hipStream_t s = (hipStream_t)stream;
  const float4 *v4 = reinterpret_cast<const float4 *>(vertices);
  if (S >= 128) {
    const int t = (S + 63) / 64;
    hipLaunchKernelGGL((mesh_depth_kernel<64>), dim3((unsigned)(t * t), (unsigned)B), dim3(1024), 0, s, v4, faces, NV, F,
                       src_size, S, clamp_max, depth);
  } else {
    const int t = (S + 31) / 32;
    hipLaunchKernelGGL((mesh_depth_kernel<32>), dim3((unsigned)(t * t), (unsigned)B), dim3(1024), 0, s, v4, faces, NV, F,
                       src_size, S, clamp_max, depth);
  }
  return (int)hipGetLastError();
}
