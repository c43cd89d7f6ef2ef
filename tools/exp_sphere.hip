// Experimental ablation kernels for the sphere rasterizer (NOT part of the product
// library).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC ... -o tools/libexp.so
#include "../spherehand_amd/csrc/sphere_raster.hip"

namespace shr {

// mode bit0: skip tile_min (mask=0); bit1: no LDS staging (direct global load per wave);
// bit2: skip store
__global__ void __launch_bounds__(1024)
exp_fwd(const float4 *__restrict__ spheres, int J, int H, int W, float *__restrict__ depth,
        int tiles_x, int ntiles, int mode) {
  __shared__ float4 s_sph[SHR_MAX_SPHERES];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const bool valid = lane < J;
  float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
  if (mode & 2) {
    if (valid) sph = spheres[(size_t)n * J + lane];
  } else {
    if (threadIdx.x < J) s_sph[threadIdx.x] = spheres[(size_t)n * J + threadIdx.x];
    __syncthreads();
    if (valid) sph = s_sph[lane];
  }
  const Axis ax = make_axis(W), ay = make_axis(H);
  float *out = depth + (size_t)n * H * W;
  for (int tile = blockIdx.y * nwaves + wave; tile < ntiles; tile += nwaves * gridDim.y) {
    const TileGeom g = tile_geom(tile, tiles_x, ax, ay, lane);
    unsigned long long mask = tile_candidates(sph, valid, g, ax, ay, H, W);
    if (mode & 1) mask = 0;
    float best[4], bsq[4];
    int owner[4];
    tile_min<false>(mask, J, sph, g, best, owner, bsq);
    if (g.v >= H) continue;
    const size_t base = (size_t)g.v * W + g.u0;
    if (!(mode & 4) && g.u0 < W)
      *reinterpret_cast<float4 *>(out + base) = make_float4(best[0], best[1], best[2], best[3]);
  }
}

__global__ void exp_fill(float4 *__restrict__ out, size_t n4, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) out[i] = make_float4(v, v, v, v);
}

__global__ void exp_sum(const float4 *__restrict__ in, size_t n4, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i < n4; i += stride) { float4 t = in[i]; s += (t.x + t.y) + (t.z + t.w); }
  s = wave_sum_lane63(s);
  if ((threadIdx.x & 63) == 63 && s == 12345.678f) out[0] = s;
}

// bwd ablation: mode bit0: skip everything after the grad load (mask=0 path but still load);
// bit1: skip wave reductions; bit3: skip the per-pixel partial computation
__global__ void __launch_bounds__(1024)
exp_bwd(const float4 *__restrict__ spheres, const float *__restrict__ grad_depth, int J, int H, int W,
        float4 *__restrict__ grad_spheres, int tiles_x, int ntiles, int mode) {
  __shared__ float4 s_sph[SHR_MAX_SPHERES];
  __shared__ float4 s_acc[16 * SHR_MAX_SPHERES];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  if (threadIdx.x < J) s_sph[threadIdx.x] = spheres[(size_t)n * J + threadIdx.x];
  for (int i = threadIdx.x; i < nwaves * J; i += blockDim.x) s_acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const bool valid = lane < J;
  const float4 sph = valid ? s_sph[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  const Axis ax = make_axis(W), ay = make_axis(H);
  const float *gin = grad_depth + (size_t)n * H * W;
  float4 *acc = s_acc + wave * J;
  float keep = 0.f;
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const TileGeom g = tile_geom(tile, tiles_x, ax, ay, lane);
    unsigned long long mask = tile_candidates(sph, valid, g, ax, ay, H, W);
    if (!(mode & 16) && mask == 0) continue;
    float gk[4] = {0.f, 0.f, 0.f, 0.f};
    const bool row_ok = g.v < H;
    const size_t base = (size_t)g.v * W + g.u0;
    if (row_ok && g.u0 < W) {
      const float4 t = *reinterpret_cast<const float4 *>(gin + base);
      gk[0] = t.x; gk[1] = t.y; gk[2] = t.z; gk[3] = t.w;
    }
    if (mode & 1) { keep += gk[0] + gk[1] + gk[2] + gk[3]; continue; }
    float best[4], bsq[4];
    int owner[4];
    tile_min<true>(mask, J, sph, g, best, owner, bsq);
    float px[4], py[4], pz[4], pw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = row_ok && (g.u0 + k < W) && owner[k] != SHR_ARGMIN_NONE;
      if (!in) owner[k] = SHR_ARGMIN_NONE;
      const float w = in ? gk[k] / bsq[k] : 0.f;
      pz[k] = in ? gk[k] : 0.f;
      pw[k] = -w;
      const float4 o = s_sph[in ? owner[k] : 0];
      px[k] = -(w * (g.xg[k] - o.x));
      py[k] = -(w * (g.yg - o.y));
    }
    if (mode & 2) { keep += px[0] + py[1] + pz[2] + pw[3] + px[3] + py[2]+ pz[1] + pw[0]; continue; }
    unsigned long long m = mask;
    while (m) {
      const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
      m &= m - 1;
      float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool mine = owner[k] == j;
        any |= mine;
        sx += mine ? px[k] : 0.f;
        sy += mine ? py[k] : 0.f;
        sz += mine ? pz[k] : 0.f;
        sw += mine ? pw[k] : 0.f;
      }
      if (__ballot(any) == 0) continue;
      sx = wave_sum_lane63(sx);
      sy = wave_sum_lane63(sy);
      sz = wave_sum_lane63(sz);
      sw = wave_sum_lane63(sw);
      if (lane == 63) {
        if (mode & 4) {
          atomicAdd(&acc[j].x, sx); atomicAdd(&acc[j].y, sy); atomicAdd(&acc[j].z, sz); atomicAdd(&acc[j].w, sw);
        } else {
          float4 a = acc[j];
          a.x += sx; a.y += sy; a.z += sz; a.w += sw;
          acc[j] = a;
        }
      }
    }
  }
  if (keep == 12345.678f) s_acc[0].x = keep;
  __syncthreads();
  if (threadIdx.x < J) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < nwaves; w++) {
      const float4 a = s_acc[w * J + threadIdx.x];
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    }
    t.w = t.w * s_sph[threadIdx.x].w;
    grad_spheres[(size_t)n * J + threadIdx.x] = t;
  }
}
}  // namespace shr

extern "C" int exp_fwd_launch(const float *spheres, int N, int J, int H, int W, float *depth, int nwaves,
                              int slices, int mode, void *stream) {
  const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
  dim3 grid(N, slices), block(64 * nwaves);
  hipLaunchKernelGGL(shr::exp_fwd, grid, block, 0, (hipStream_t)stream, (const float4 *)spheres, J, H, W, depth,
                     tiles_x, tiles_x * tiles_y, mode);
  return (int)hipGetLastError();
}
extern "C" int exp_bwd_launch(const float *spheres, const float *grad, int N, int J, int H, int W, float *gs,
                              int nwaves, int mode, void *stream) {
  const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
  dim3 grid(N, 1), block(64 * nwaves);
  hipLaunchKernelGGL(shr::exp_bwd, grid, block, 0, (hipStream_t)stream, (const float4 *)spheres, grad, J, H, W,
                     (float4 *)gs, tiles_x, tiles_x * tiles_y, mode);
  return (int)hipGetLastError();
}
extern "C" int exp_fill_launch(float *out, size_t n, int blocks, int threads, void *stream) {
  hipLaunchKernelGGL(shr::exp_fill, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (float4 *)out, n / 4, 100.0f);
  return (int)hipGetLastError();
}
extern "C" int exp_sum_launch(const float *in, size_t n, float *out, int blocks, int threads, void *stream) {
  hipLaunchKernelGGL(shr::exp_sum, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (const float4 *)in, n / 4, out);
  return (int)hipGetLastError();
}
