// mv_consistency.hip -- MultiviewConsistencyLoss (mesh/multiview_utility.py:138-167, hm_weight = None) with its
// gradient, one launch: every view's joints are mapped to the canonical frame (p = R joint + t, t = COLUMN 3
// of the camera pose, as the reference reads it, :151-153), the per-coordinate median over the views is the
// robust average (torch.median: the LOWER median, first index on ties, :156) and the loss is the squared
// distance of every view's point to it (MSELoss over [B,V,J,3], :167).
//
// One wave per sample, lanes = joints; the V <= 8 canonical points of a joint live in registers, the median
// is found by rank counting (no sorting network, any V).  Autograd routes the median's gradient to the view
// it was taken from, so with m = c_{v*}:  d/dc_v = 2 (c_v - m) for v != v*,  d/dc_{v*} = -sum of the others;
// the joints receive R_v^T times that.  The kernel emits the per-sample sum and the UNIT gradient
// (d sum / d joints); the caller applies the mean and its weight.  Replaces ~25 indexing / elementwise
// launches per step (forward + backward).
#include "common.h"

namespace shr {

constexpr int kMaxViews = 8;

__global__ void __launch_bounds__(64)
mv_consistency_kernel(const float *__restrict__ cam, const float *__restrict__ joints, int V, int J,
                      float *__restrict__ loss_sum, float *__restrict__ grad_joints) {
  const int b = blockIdx.x, j = threadIdx.x;
  const bool live = j < J;
  float c[kMaxViews][3];
  bool nan = false;
#pragma unroll
  for (int v = 0; v < kMaxViews; v++) {
    c[v][0] = c[v][1] = c[v][2] = 0.f;
    if (v < V) {
      const float *T = cam + ((size_t)b * V + v) * 16;     // wave-uniform: scalar loads
      const float *p = joints + (((size_t)b * V + v) * J + (live ? j : 0)) * 3;
      const float x = p[0], y = p[1], z = p[2];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        // torch.matmul of a 3x3 with a 3-vector, then + t: ((R0 x + R1 y) + R2 z) + t
        c[v][r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
        nan |= c[v][r] != c[v][r];
      }
    }
  }
  const int kth = (V - 1) >> 1;                              // lower median
  float loss = 0.f;
  float g[kMaxViews][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float m = 0.f;
    int vstar = 0;
#pragma unroll
    for (int v = 0; v < kMaxViews; v++) {
      if (v < V) {
        int rank = 0;
#pragma unroll
        for (int u = 0; u < kMaxViews; u++)
          if (u < V) rank += (c[u][r] < c[v][r]) || (c[u][r] == c[v][r] && u < v);
        if (rank == kth) { m = c[v][r]; vstar = v; }
      }
    }
    if (nan) m = __builtin_nanf("");                          // torch.median propagates NaN
    float others = 0.f;
#pragma unroll
    for (int v = 0; v < kMaxViews; v++) {
      g[v][r] = 0.f;
      if (v < V) {
        const float d = c[v][r] - m;
        loss += d * d;
        g[v][r] = 2.f * d;
        others += (v == vstar) ? 0.f : 2.f * d;
      }
    }
#pragma unroll
    for (int v = 0; v < kMaxViews; v++)
      if (v < V && v == vstar) g[v][r] = -others;
  }
  if (!live) loss = 0.f;
  const float total = readlane_f(wave_sum_lane63(loss), 63);
  if (j == 0) loss_sum[b] = total;
  if (grad_joints && live) {
#pragma unroll
    for (int v = 0; v < kMaxViews; v++) {
      if (v < V) {
        const float *T = cam + ((size_t)b * V + v) * 16;
        float *o = grad_joints + (((size_t)b * V + v) * J + j) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++)                            // R^T g
          o[k] = (T[k] * g[v][0] + T[4 + k] * g[v][1]) + T[8 + k] * g[v][2];
      }
    }
  }
}

// DepthResample.forward (network/util_modules.py:10-43): pixels whose uniform draw exceeds `sample_ratio` become
// 1.0 (the scaled background), then the module's fixed 3x3 / 5x5 Gaussian, zero padding (nn.Conv2d(padding = k/2)).
__global__ void __launch_bounds__(256)
depth_resample_kernel(const float *__restrict__ dm, const float *__restrict__ uniform, int N, int H, int W,
                      float sample_ratio, int ksize, float *__restrict__ out) {
  // (1 2 1 / 2 6 2 / 1 2 1) / 18 and the 5x5 table / 273
  const float k3[9] = {1, 2, 1, 2, 6, 2, 1, 2, 1};
  const float k5[25] = {1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1};
  const float norm = ksize == 5 ? 273.f : 18.f;
  const int half = ksize >> 1;
  const size_t n = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((size_t)H * W));
    const int r = (int)(i - (size_t)b * H * W);
    const int v = r / W, u = r - v * W;
    float acc = 0.f;
    for (int dy = -half; dy <= half; dy++) {
      for (int dx = -half; dx <= half; dx++) {
        const int vv = v + dy, uu = u + dx;
        if (vv < 0 || vv >= H || uu < 0 || uu >= W) continue;
        const size_t q = (size_t)b * H * W + (size_t)vv * W + uu;
        const float x = uniform[q] > sample_ratio ? 1.0f : dm[q];
        const float w = (ksize == 5 ? k5[(dy + 2) * 5 + dx + 2] : k3[(dy + 1) * 3 + dx + 1]) / norm;
        acc += w * x;
      }
    }
    out[i] = acc;
  }
}

}  // namespace shr

extern "C" int shr_mv_consistency(const float *cam, const float *joints, int B, int V, int J, float *loss_sum,
                                  float *grad_joints, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!cam || !joints || !loss_sum || B < 0 || V <= 0 || J <= 0) return SHR_EINVAL;
  if (V > kMaxViews || J > SHR_MAX_SPHERES) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(mv_consistency_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, cam, joints, V, J,
                     loss_sum, grad_joints);
  return (int)hipGetLastError();
}

extern "C" int shr_depth_resample(const float *depth, const float *uniform, int N, int H, int W, float sample_ratio,
                                  int kernel_size, float *out, void *stream) {
  using namespace shr;
  if (N == 0) return SHR_OK;
  if (!depth || !uniform || !out || N < 0 || H <= 0 || W <= 0 || depth == out || (kernel_size != 3 && kernel_size != 5))
    return SHR_EINVAL;
  const size_t n = (size_t)N * H * W;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(depth_resample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, uniform, N, H, W,
                     sample_ratio, kernel_size, out);
  return (int)hipGetLastError();
}
