// mutual_project.hip -- view-to-view projection of the sphere centres.
//
// Replaces (reference file:line): mesh/multiview_utility.py:13-30
// MutualTransformation (M[b,i,j] = inv_cam[b,j] @ cam[b,i], detached at :68) and
// :62-72 (p = M[:3,:3] @ joint + M[:3,3] for every ordered view pair), emitting
// the rasterizer's sphere records directly:
//     spheres[b,i,j,k] = (p.x, p.y, p.z, radii[k])
// Backward: grad_joints[b,i,k] = sum_j M[b,i,j][:3,:3]^T @ grad_spheres[b,i,j,k].xyz
// (the radii are buffers in the reference and get no gradient).
// Tiny, latency-bound kernels: one thread per output record.

#include "common.h"

namespace shr {

__device__ __forceinline__ void pair_rt(const float *__restrict__ cam, const float *__restrict__ inv_cam, int b, int V,
                                        int i, int j, float R[3][3], float t[3]) {
  const float *A = inv_cam + ((size_t)b * V + j) * 16;  // rows of inv_cam[b,j]
  const float *Bm = cam + ((size_t)b * V + i) * 16;     // cam[b,i]
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) s += A[4 * r + k] * Bm[4 * k + c];
      if (c < 3) R[r][c] = s; else t[r] = s;
    }
  }
}

__global__ void mutual_project_fwd_kernel(const float *__restrict__ cam, const float *__restrict__ inv_cam,
                                          const float *__restrict__ joints, const float *__restrict__ radii, int B,
                                          int V, int J, float4 *__restrict__ spheres, int *__restrict__ zero,
                                          int nzero) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * V * V * J;
  // (shr_mv_project_compact: the fill counters of the point lists the NEXT launch builds -- a memset of their own
  // is a 5-us launch)
  if (idx < nzero) zero[idx] = 0;
  if (idx >= total) return;
  const int k = (int)(idx % J);
  const int j = (int)((idx / J) % V);
  const int i = (int)((idx / ((long long)J * V)) % V);
  const int b = (int)(idx / ((long long)J * V * V));
  float R[3][3], t[3];
  pair_rt(cam, inv_cam, b, V, i, j, R, t);
  const float *p = joints + (((size_t)b * V + i) * J + k) * 3;
  float o[3];
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = ((R[r][0] * p[0] + R[r][1] * p[1]) + R[r][2] * p[2]) + t[r];
  spheres[idx] = make_float4(o[0], o[1], o[2], radii[k]);
}

__global__ void mutual_project_bwd_kernel(const float *__restrict__ cam, const float *__restrict__ inv_cam,
                                          const float4 *__restrict__ grad_spheres, int B, int V, int J,
                                          float *__restrict__ grad_joints) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * V * J;
  if (idx >= total) return;
  const int k = (int)(idx % J);
  const int i = (int)((idx / J) % V);
  const int b = (int)(idx / ((long long)J * V));
  float g[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < V; j++) {
    float R[3][3], t[3];
    pair_rt(cam, inv_cam, b, V, i, j, R, t);
    const float4 gs = grad_spheres[(((size_t)b * V + i) * V + j) * J + k];
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] += (R[0][c] * gs.x + R[1][c] * gs.y) + R[2][c] * gs.z;
  }
  float *o = grad_joints + idx * 3;
  o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
}

// Assembly of MutualProjectionLoss (mesh/multiview_utility.py:98-129) from the partial results of the fused
// render-and-compare kernel and of the data->model kernel, with its whole backward, in one call:
//   is_mv:   loss = 9 * sum(sse) / (B V V H W) + 500 * 9 * sum(d2m) / (B V V H W)          over all V*V pairs
//   else:    loss = 3 * sum_diag(sse) / (B H W)  + 500 * 3 * sum_diag(d2m) / (B H W)       over the V same-view pairs
// and grad_joints[b,i,k] = sum_j R(b,i,j)^T (w_m * d sse / d centre + w_d * d d2m / d centre) -- the sphere
// gradients of the two kernels (R partials each) weighted, added and pulled back through the view transforms
// (constants: detached at :68; the radii are buffers).  The loss is accumulated in fp64 in a fixed order.
// d2m_part / gd2m_part hold one entry per pair (is_mv) or per DIAGONAL pair b*V+i (else); is_mv == 2: the same-view
// pairs only AND sse_part / gsp_part compacted the same way (the caller ran the fused kernel on those pairs alone).
__global__ void __launch_bounds__(256)
mv_loss_combine_kernel(const float *__restrict__ cam, const float *__restrict__ inv_cam,
                       const float *__restrict__ sse_part, const float4 *__restrict__ gsp_part, int Rm,
                       const float *__restrict__ d2m_part, const float *__restrict__ gd2m_part, int Rd, int B, int V,
                       int J, int is_mv, float w_m, float w_d, float *__restrict__ loss_out,
                       float *__restrict__ grad_joints) {
  const long long total = (long long)B * V * J;
  if (blockIdx.x == gridDim.x - 1) {
    // the scalar: one workgroup; every thread a strided fp64 partial over the FLAT partial-result arrays (one element
    // per iteration: independent loads, where a loop over pairs with inner loops over the parts chained 4.5 x 8
    // dependent round trips per thread -- 11 us for config 5's 1152 pairs), then a fixed tree: deterministic
    __shared__ double s_m[256], s_d[256];
    // (32-bit indices: the launcher keeps B V V J below 2^31 and the partial counts small; four loads in flight per
    // thread -- one element per iteration with 64-bit divisions for its pair chained ~20 round trips and ~60 long
    // divisions per thread: 11 us for config 5's 1152 pairs)
    // (is_mv == 2: the same-view pairs only, and sse_part / gsp_part hold just those B*V entries, like the d2m parts)
    const int N = B * V * V, NE = (is_mv == 2 ? B * V : N) * Rm;
    double am = 0.0, ad = 0.0;
    // (eight loads in flight: two rounds of four, added round by round -- the sums are those of the one-round loop)
    for (int e0 = threadIdx.x; e0 < NE; e0 += 2048) {
      float v[2][4];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int e = e0 + 1024 * h + 256 * u;
          bool take = e < NE;
          if (is_mv == 0) {
            const int ec = min(e, NE - 1), n = ec / Rm, j = n % V, i = (n / V) % V;
            take = take && i == j;
          }
          const float x = sse_part[min(e, NE - 1)];   // (always requested -- a load behind a branch waits before the next one is issued)
          v[h][u] = take ? x : 0.f;
        }
      am += ((double)v[0][0] + (double)v[0][1]) + ((double)v[0][2] + (double)v[0][3]);
      if (e0 + 1024 < NE) am += ((double)v[1][0] + (double)v[1][1]) + ((double)v[1][2] + (double)v[1][3]);
    }
    const int DE = (is_mv == 1 ? N : B * V) * Rd;     // d2m entries: every pair, or the same-view pairs only
    for (int e0 = threadIdx.x; e0 < DE; e0 += 2048) {
      float v[2][4];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int e = e0 + 1024 * h + 256 * u;
          const float x = d2m_part[min(e, DE - 1)];
          v[h][u] = e < DE ? x : 0.f;
        }
      ad += ((double)v[0][0] + (double)v[0][1]) + ((double)v[0][2] + (double)v[0][3]);
      if (e0 + 1024 < DE) ad += ((double)v[1][0] + (double)v[1][1]) + ((double)v[1][2] + (double)v[1][3]);
    }
    s_m[threadIdx.x] = am; s_d[threadIdx.x] = ad;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if ((int)threadIdx.x < h) { s_m[threadIdx.x] += s_m[threadIdx.x + h]; s_d[threadIdx.x] += s_d[threadIdx.x + h]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = (float)((double)w_m * s_m[0] + (double)w_d * s_d[0]);
    return;
  }
  // d loss / d joints: VP = 4 (V <= 4) or 8 lanes per (b, i, k), lane j = the pair (b, i, j): the V contributions are
  // gathered side by side and added across the lanes (xor butterfly: (c0 + c1) + (c2 + c3), for V = 3 the serial
  // order (c0 + c1) + c2 to the bit) -- a loop over j chained V rounds of dependent loads per thread (10 us at config 5)
  if (!grad_joints) return;
  const int vp_shift = V <= 4 ? 2 : 3, VP = 1 << vp_shift;
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int idx = t >> vp_shift, j = t & (VP - 1);
  const bool in = idx < (int)total;
  const int k = in ? idx % J : 0;
  const int i = in ? (idx / J) % V : 0;
  const int b = in ? idx / (J * V) : 0;
  float g[3] = {0.f, 0.f, 0.f};
  if (in && j < V && (is_mv == 1 || j == i)) {
    const long long n = ((long long)b * V + i) * V + j;
    const long long e = is_mv == 1 ? n : (long long)b * V + i;
    const long long nm = is_mv == 2 ? e : n;
    // the first four parts of either kind are requested side by side (a loop of unknown length waits for every part
    // before it asks for the next: with config 5's 4 + 2 parts six chained round trips per lane, most of this kernel's
    // 5-8 us); added in the same order
    float4 am[4];
    float ad[4][3];
#pragma unroll
    for (int r = 0; r < 4; r++) am[r] = gsp_part[(nm * Rm + min(r, Rm - 1)) * J + k];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float *a = gd2m_part + ((e * Rd + min(r, Rd - 1)) * J + k) * 3;
      ad[r][0] = a[0]; ad[r][1] = a[1]; ad[r][2] = a[2];
    }
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (r < Rm) { gx += am[r].x; gy += am[r].y; gz += am[r].z; }
    for (int r = 4; r < Rm; r++) {
      const float4 a = gsp_part[(nm * Rm + r) * J + k];
      gx += a.x; gy += a.y; gz += a.z;
    }
    float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (r < Rd) { dx += ad[r][0]; dy += ad[r][1]; dz += ad[r][2]; }
    for (int r = 4; r < Rd; r++) {
      const float *a = gd2m_part + ((e * Rd + r) * J + k) * 3;
      dx += a[0]; dy += a[1]; dz += a[2];
    }
    const float sx = w_m * gx + w_d * dx, sy = w_m * gy + w_d * dy, sz = w_m * gz + w_d * dz;
    float R[3][3], tt[3];
    pair_rt(cam, inv_cam, b, V, i, j, R, tt);
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = (R[0][c] * sx + R[1][c] * sy) + R[2][c] * sz;
  }
  for (int m = 1; m < VP; m <<= 1) {
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] += __shfl_xor(g[c], m);
  }
  if (in && j == 0) {
    float *o = grad_joints + (size_t)idx * 3;
    o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
  }
}

}  // namespace shr

extern "C" int shr_mv_loss_combine(const float *cam, const float *inv_cam, const float *sse_part,
                                   const float *grad_spheres_part, int Rm, const float *d2m_part,
                                   const float *grad_d2m_part, int Rd, int B, int V, int J, int H, int W, int is_mv,
                                   float d2m_weight, float *loss, float *grad_joints, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!cam || !inv_cam || !sse_part || !grad_spheres_part || !d2m_part || !grad_d2m_part || !loss || B < 0 || V <= 0 ||
      J <= 0 || H <= 0 || W <= 0 || Rm <= 0 || Rd <= 0 || is_mv < 0 || is_mv > 2)
    return SHR_EINVAL;
  if (((uintptr_t)grad_spheres_part & 15u) != 0) return SHR_EINVAL;
  if ((long long)B * V * V * J > (1LL << 31) - 256) return SHR_ETOOLARGE;
  // MSELoss means and the reference's x9 / x3 (mesh/multiview_utility.py:100-101, :126-127); DataToModelLoss means
  // over the same pixel counts (mesh/render.py:142) times its x9 / x3 and the caller's 500 (:129)
  const double px = (double)H * W;
  const double wm = is_mv == 1 ? 9.0 / ((double)B * V * V * px) : 3.0 / ((double)B * px);
  const double wd = (double)d2m_weight * wm;
  const long long total = (long long)B * V * J;
  if (V > 8) return SHR_ETOOLARGE;   // (the pairs of a view sit side by side in 4 or 8 lanes)
  if ((long long)B * V * V * (Rm > Rd ? Rm : Rd) * J * 4 > (1LL << 31) - 4096) return SHR_ETOOLARGE;   // 32-bit entry counts and offsets
  const long long lanes = total * (V <= 4 ? 4 : 8);
  if (lanes > (1LL << 31) - 512) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(mv_loss_combine_kernel, dim3((unsigned)((lanes + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream,
                     cam, inv_cam, sse_part, reinterpret_cast<const float4 *>(grad_spheres_part), Rm, d2m_part,
                     grad_d2m_part, Rd, B, V, J, is_mv, (float)wm, (float)wd, loss, grad_joints);
  return (int)hipGetLastError();
}

extern "C" int shr_mutual_project_fwd(const float *cam, const float *inv_cam, const float *joints,
                                      const float *radii, int B, int V, int J, float *spheres, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!cam || !inv_cam || !joints || !radii || !spheres || B < 0 || V <= 0 || J <= 0) return SHR_EINVAL;
  if (((uintptr_t)spheres & 15u) != 0) return SHR_EINVAL;
  const long long total = (long long)B * V * V * J;
  if (total > (1LL << 31) - 256) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(mutual_project_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cam, inv_cam, joints, radii, B, V, J, reinterpret_cast<float4 *>(spheres),
                     (int *)nullptr, 0);
  return (int)hipGetLastError();
}

extern "C" int shr_mv_project_compact(const float *cam, const float *inv_cam, const float *joints, const float *radii,
                                      int B, int V, int J, float *spheres, const float *depth, int M, int H, int W,
                                      void *workspace, void *stream) {
  using namespace shr;
  if (B == 0 || M == 0) {   // nothing to fuse: each step on its own (either returns at once on its empty side)
    const int rc = shr_mutual_project_fwd(cam, inv_cam, joints, radii, B, V, J, spheres, stream);
    return rc != SHR_OK ? rc : shr_data_to_model_compact(depth, M, H, W, workspace, stream);
  }
  if (!cam || !inv_cam || !joints || !radii || !spheres || B < 0 || V <= 0 || J <= 0) return SHR_EINVAL;
  if (((uintptr_t)spheres & 15u) != 0) return SHR_EINVAL;
  const long long total = (long long)B * V * V * J;
  if (total > (1LL << 31) - 256) return SHR_ETOOLARGE;
  int *counts = nullptr;
  const int rc = d2m_compact_check(depth, M, H, W, workspace, &counts);
  if (rc != SHR_OK) return rc;
  const long long threads = total > M ? total : (long long)M;
  hipLaunchKernelGGL(mutual_project_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cam, inv_cam, joints, radii, B, V, J, reinterpret_cast<float4 *>(spheres),
                     counts, M);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return d2m_compact_launch(depth, M, H, W, workspace, (hipStream_t)stream);
}

extern "C" int shr_mutual_project_bwd(const float *cam, const float *inv_cam, const float *grad_spheres, int B, int V,
                                      int J, float *grad_joints, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!cam || !inv_cam || !grad_spheres || !grad_joints || B < 0 || V <= 0 || J <= 0) return SHR_EINVAL;
  if (((uintptr_t)grad_spheres & 15u) != 0) return SHR_EINVAL;
  const long long total = (long long)B * V * J;
  if ((long long)B * V * V * J > (1LL << 31) - 256) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(mutual_project_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cam, inv_cam, reinterpret_cast<const float4 *>(grad_spheres), B, V, J,
                     grad_joints);
  return (int)hipGetLastError();
}
