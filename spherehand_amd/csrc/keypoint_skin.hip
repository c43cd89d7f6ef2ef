// keypoint_skin.hip -- bone transforms -> the rasterizer's sphere records, and back.
//
// Replaces (reference file:line): the LinearBlendSkinning of the 41 key-points inside
// HandBallPrimitiveRender (mesh/render.py:65-85: every key-point is bound to ONE bone with weight 1, so
// mesh/pointTransformation.py:39-46 reduces to p = T[bone] (w v), x -> -x for the right hand) and the
// torch.cat with the radii that feeds BallRender (mesh/render.py:85-88), plus their autograd backward:
//     spheres[b,j] = (s * p.x, p.y, p.z, radii[j]),   p = T[b, bone[j]] @ wv[j]
//     grad_T[b,nb,r,c] = sum over the key-points j of bone nb of g[b,j,r] * wv[j,c]   (r < 3; s on r = 0)
// In torch this is an index_select, a batched 4x4 matmul, a multiply and a cat forward (and an index_add among
// the backward): ~280 of the 316 us a pose -> depth -> pose round trip took as one hipGraph (round 3) around 14 us
// of rasterizer.  Tiny, latency-bound kernels: one thread per record / per (sample, bone); the backward walks a
// bone's key-points in index order (deterministic).
#include "common.h"

namespace shr {

__global__ void keypoint_spheres_fwd_kernel(const float *__restrict__ T, const int *__restrict__ bone,
                                            const float4 *__restrict__ wv, const float *__restrict__ radii, float sx,
                                            int B, int NB, int J, float4 *__restrict__ spheres) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * J) return;
  const int j = idx % J, b = idx / J;
  const float4 *t = reinterpret_cast<const float4 *>(T + ((size_t)b * NB + bone[j]) * 16);
  const float4 v = wv[j];
  const float4 r0 = t[0], r1 = t[1], r2 = t[2];
  const float x = ((r0.x * v.x + r0.y * v.y) + r0.z * v.z) + r0.w * v.w;
  const float y = ((r1.x * v.x + r1.y * v.y) + r1.z * v.z) + r1.w * v.w;
  const float z = ((r2.x * v.x + r2.y * v.y) + r2.z * v.z) + r2.w * v.w;
  spheres[idx] = make_float4(sx * x, y, z, radii[j]);
}

// bone_start[NB + 1], bone_points[J]: the key-points of each bone (CSR), ascending
__global__ void keypoint_spheres_bwd_kernel(const float4 *__restrict__ grad_spheres, const int *__restrict__ bone_start,
                                            const int *__restrict__ bone_points, const float4 *__restrict__ wv,
                                            float sx, int B, int NB, int J, float *__restrict__ grad_T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * NB) return;
  const int nb = idx % NB, b = idx / NB;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
  for (int k = bone_start[nb]; k < bone_start[nb + 1]; k++) {
    const int j = bone_points[k];
    const float4 g = grad_spheres[(size_t)b * J + j];
    const float4 v = wv[j];
    const float gx = sx * g.x;
    a0.x += gx * v.x; a0.y += gx * v.y; a0.z += gx * v.z; a0.w += gx * v.w;
    a1.x += g.y * v.x; a1.y += g.y * v.y; a1.z += g.y * v.z; a1.w += g.y * v.w;
    a2.x += g.z * v.x; a2.y += g.z * v.y; a2.z += g.z * v.z; a2.w += g.z * v.w;
  }
  float4 *o = reinterpret_cast<float4 *>(grad_T + (size_t)idx * 16);
  o[0] = a0; o[1] = a1; o[2] = a2;
  o[3] = make_float4(0.f, 0.f, 0.f, 0.f);   // (the homogeneous row never reaches a sphere record)
}

}  // namespace shr

extern "C" int shr_keypoint_spheres_fwd(const float *T, int B, int NB, int J, const int32_t *bone, const float *wv,
                                        const float *radii, int right_hand, float *spheres, void *stream) {
  using namespace shr;
  if (B == 0 || J == 0) return SHR_OK;
  if (!T || !bone || !wv || !radii || !spheres || B < 0 || NB <= 0 || J < 0) return SHR_EINVAL;
  if ((((uintptr_t)T | (uintptr_t)wv | (uintptr_t)spheres) & 15u) != 0) return SHR_EINVAL;
  if ((long long)B * J > (1LL << 30)) return SHR_ETOOLARGE;
  const int n = B * J;
  hipLaunchKernelGGL(keypoint_spheres_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, bone,
                     reinterpret_cast<const float4 *>(wv), radii, right_hand ? -1.0f : 1.0f, B, NB, J,
                     reinterpret_cast<float4 *>(spheres));
  return (int)hipGetLastError();
}

extern "C" int shr_keypoint_spheres_bwd(const float *grad_spheres, int B, int NB, int J, const int32_t *bone_start,
                                        const int32_t *bone_points, const float *wv, int right_hand, float *grad_T,
                                        void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!grad_spheres || !bone_start || !bone_points || !wv || !grad_T || B < 0 || NB <= 0 || J < 0) return SHR_EINVAL;
  if ((((uintptr_t)grad_spheres | (uintptr_t)wv | (uintptr_t)grad_T) & 15u) != 0) return SHR_EINVAL;
  if ((long long)B * NB > (1LL << 30) || (long long)B * J > (1LL << 30)) return SHR_ETOOLARGE;
  const int n = B * NB;
  hipLaunchKernelGGL(keypoint_spheres_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(grad_spheres), bone_start, bone_points,
                     reinterpret_cast<const float4 *>(wv), right_hand ? -1.0f : 1.0f, B, NB, J, grad_T);
  return (int)hipGetLastError();
}
