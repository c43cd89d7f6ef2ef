// pair_losses.hip -- the two hinge losses on sphere-centre pairs, with their gradients, one launch:
//   CollisionLoss  (mesh/render.py:145-176): sum over the pair table of relu(min_dist^2 - |c_a - c_b|^2);
//                  the table = every finger sphere against the 11 palm spheres and against every sphere
//                  of another finger (330 + 360 pairs; spheres 11 + 6f .. 16 + 6f belong to finger f);
//   BoneLengthLoss (mesh/render.py:179-206): relu(min_k - d_k^2) and relu(d_k^2 - max_k) over K given
//                  pairs, min / max = (0.80 / 1.05 x rest length)^2.
// One wave per sample, lanes = spheres: lane j walks its partners, so its gradient is a register sum in a
// fixed order (deterministic, no atomics); each pair's loss is counted by its smaller index.  The kernel
// emits per-sample sums and UNIT gradients (d sum / d centres); the caller applies the reference's
// reductions (sum / two means) and weights.  Replaces ~45 indexing / elementwise launches per step.
#include "common.h"

namespace shr {


__global__ void __launch_bounds__(64)
pair_losses_kernel(const float *__restrict__ joints, long long sample_stride, int J, int npalm, int per_finger, float min_sq,
                   const int *__restrict__ bone_a, const int *__restrict__ bone_b, const float *__restrict__ bone_min,
                   const float *__restrict__ bone_max, int K, float *__restrict__ coll_sum, float *__restrict__ bone_lo_sum,
                   float *__restrict__ bone_hi_sum, float *__restrict__ grad_coll, float *__restrict__ grad_bone_lo,
                   float *__restrict__ grad_bone_hi) {
  __shared__ float s_c[SHR_MAX_SPHERES][3];
  const int m = blockIdx.x, j = threadIdx.x;
  const float *c = joints + (size_t)m * sample_stride;
  if (j < J) { s_c[j][0] = c[3 * j]; s_c[j][1] = c[3 * j + 1]; s_c[j][2] = c[3 * j + 2]; }
  __syncthreads();
  const bool live = j < J;
  const float x = live ? s_c[j][0] : 0.f, y = live ? s_c[j][1] : 0.f, z = live ? s_c[j][2] : 0.f;
  // ---- collision ----------------------------------------------------------------------------
  float loss = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
  const int fj = j < npalm ? -1 : (j - npalm) / per_finger;
  for (int q = 0; q < J; q++) {
    const int fq = q < npalm ? -1 : (q - npalm) / per_finger;
    const bool pair = live && q != j && (fj != fq) ;   // palm-finger or two different fingers (palm-palm: fj == fq == -1)
    const float dx = x - s_c[q][0], dy = y - s_c[q][1], dz = z - s_c[q][2];
    const float h = min_sq - ((dx * dx + dy * dy) + dz * dz);
    if (pair && h > 0.f) {
      if (j < q) loss += h;
      gx -= 2.f * dx; gy -= 2.f * dy; gz -= 2.f * dz;
    }
  }
  if (live) {
    float *g = grad_coll + ((size_t)m * J + j) * 3;
    g[0] = gx; g[1] = gy; g[2] = gz;
  }
  const float csum = readlane_f(wave_sum_lane63(loss), 63);
  // ---- bone lengths ---------------------------------------------------------------------------
  float lo = 0.f, hi = 0.f, lx = 0.f, ly = 0.f, lz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
  for (int k = 0; k < K; k++) {
    const int a = bone_a[k], b = bone_b[k];
    const float dx = s_c[a][0] - s_c[b][0], dy = s_c[a][1] - s_c[b][1], dz = s_c[a][2] - s_c[b][2];
    const float sq = (dx * dx + dy * dy) + dz * dz;
    const float under = bone_min[k] - sq, over = sq - bone_max[k];
    const float sgn = (j == a) ? 1.f : ((j == b) ? -1.f : 0.f);     // d sq / d c_j = sgn * 2 d
    if (under > 0.f) { if (j == 0) lo += under; lx -= sgn * 2.f * dx; ly -= sgn * 2.f * dy; lz -= sgn * 2.f * dz; }
    if (over > 0.f) { if (j == 0) hi += over; hx += sgn * 2.f * dx; hy += sgn * 2.f * dy; hz += sgn * 2.f * dz; }
  }
  if (live) {
    float *g = grad_bone_lo + ((size_t)m * J + j) * 3;
    g[0] = lx; g[1] = ly; g[2] = lz;
    g = grad_bone_hi + ((size_t)m * J + j) * 3;
    g[0] = hx; g[1] = hy; g[2] = hz;
  }
  if (j == 0) { coll_sum[m] = csum; bone_lo_sum[m] = lo; bone_hi_sum[m] = hi; }
}

}  // namespace shr

extern "C" int shr_pair_losses(const float *joints, long long sample_stride, int M, int J, int num_palm, int per_finger,
                               float min_dist_sq,
                               const int32_t *bone_a, const int32_t *bone_b, const float *bone_min_sq,
                               const float *bone_max_sq, int K, float *coll_sum, float *bone_lo_sum, float *bone_hi_sum,
                               float *grad_coll, float *grad_bone_lo, float *grad_bone_hi, void *stream) {
  using namespace shr;
  if (M == 0) return SHR_OK;
  if (!joints || !coll_sum || !bone_lo_sum || !bone_hi_sum || !grad_coll || !grad_bone_lo || !grad_bone_hi || M < 0 ||
      sample_stride < (long long)J * 3 || J <= 0 || num_palm < 0 || per_finger <= 0 || K < 0 || (K > 0 && (!bone_a || !bone_b || !bone_min_sq || !bone_max_sq)))
    return SHR_EINVAL;
  if (J > SHR_MAX_SPHERES) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(pair_losses_kernel, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, joints, sample_stride, J, num_palm,
                     per_finger, min_dist_sq, bone_a, bone_b, bone_min_sq, bone_max_sq, K, coll_sum, bone_lo_sum,
                     bone_hi_sum, grad_coll, grad_bone_lo, grad_bone_hi);
  return (int)hipGetLastError();
}
