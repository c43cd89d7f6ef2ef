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
                                          int V, int J, float4 *__restrict__ spheres) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * V * V * J;
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

}  // namespace shr

extern "C" int shr_mutual_project_fwd(const float *cam, const float *inv_cam, const float *joints,
                                      const float *radii, int B, int V, int J, float *spheres, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!cam || !inv_cam || !joints || !radii || !spheres || B < 0 || V <= 0 || J <= 0) return SHR_EINVAL;
  if (((uintptr_t)spheres & 15u) != 0) return SHR_EINVAL;
  const long long total = (long long)B * V * V * J;
  if (total > (1LL << 31) - 256) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(mutual_project_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cam, inv_cam, joints, radii, B, V, J, reinterpret_cast<float4 *>(spheres));
  return (int)hipGetLastError();
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
