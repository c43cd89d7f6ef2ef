// sphere_tile.h -- GENERAL (any input) sphere rasterizer kernels: pixel-parallel
// 32x8 wave tiles with an exact per-tile candidate cull.  Exact for every fp32
// input including NaN / Inf / spheres behind the background plane; used
//   * as the per-crop fallback of the LDS z-buffer kernels (sphere_zbuf.h) when
//     a crop fails their precondition, and
//   * for the backward when no saved owner map is supplied.
//
// Replaces (reference file:line): mesh/render.py:26-53 BallRender.forward and the
// min over the sphere axis at mesh/render.py:89 / mesh/multiview_utility.py:76.
//
//   1. lanes = spheres: lane j tests sphere j's extent against the wave's tile,
//      __ballot gives the 64-bit candidate mask (exact conservative cull);
//   2. lanes = pixels: each lane owns 4 consecutive pixels of one row; the wave
//      walks the set bits, the candidate's parameters are broadcast with
//      v_readlane (SGPR operands), each lane keeps (min depth, owner);
//   3. forward: one 16-byte store per lane (8 lanes = one 128-B line);
//      backward: per owner sphere a DPP wave sum of the four partials into the
//      wave's private LDS row; rows are combined in wave order at the end ->
//      deterministic, no float atomics.
#pragma once
#include "common.h"

namespace shr {

constexpr float kBackground = 100.0f;  // mesh/render.py:52
constexpr float kHitMin = 0.01f;       // mesh/render.py:41-42

struct TileGeom {
  int x0, y0;        // first pixel of the wave's tile
  float yg;          // this lane's row coordinate
  float xg[4];       // this lane's four column coordinates
  int u0, v;         // this lane's first column / row
};

__device__ __forceinline__ TileGeom tile_geom(int tile, int tiles_x, const Axis &ax, const Axis &ay,
                                              int lane) {
  TileGeom g;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  g.x0 = tx * kTileW;
  g.y0 = ty * kTileH;
  g.u0 = g.x0 + 4 * (lane & 7);
  g.v = g.y0 + (lane >> 3);
  g.yg = axis_coord(ay, g.v);
#pragma unroll
  for (int k = 0; k < 4; k++) g.xg[k] = axis_coord(ax, g.u0 + k);
  return g;
}

// Candidate mask of a tile.  A pixel can only be hit if |fl(xg - x)| <= |r|:
// fl(r*r) - fl(dx*dx) is exact-or-negative once |dx| > |r| (rounding is
// monotonic), and q must exceed 0.01.  fl(xg - x) is monotonic in xg, so the
// tile is culled iff fl(xlo - x) > |r| or fl(xhi - x) < -|r| (same for y): an
// exact test, no slack needed.  A NaN in x, y or r makes q NaN at EVERY pixel
// (a hit, mesh/render.py:41-42), so such a sphere is never culled.
__device__ __forceinline__ unsigned long long tile_candidates(const float4 sph, bool valid,
                                                              const TileGeom &g, const Axis &ax,
                                                              const Axis &ay, int H, int W) {
  const float xlo = axis_coord(ax, g.x0);
  const float xhi = axis_coord(ax, min(g.x0 + kTileW, W) - 1);
  const float ylo = axis_coord(ay, g.y0);
  const float yhi = axis_coord(ay, min(g.y0 + kTileH, H) - 1);
  const float ar = fabsf(sph.w);
  const bool outside = (xlo - sph.x > ar) || (xhi - sph.x < -ar) || (ylo - sph.y > ar) ||
                       (yhi - sph.y < -ar);
  const bool has_nan = (sph.x != sph.x) || (sph.y != sph.y) || (ar != ar);
  return __ballot(valid && (has_nan || !outside));
}

// Running minimum over the candidates for this lane's 4 pixels.
//   best[k]  depth so far;  owner[k]  owning sphere (SHR_ARGMIN_NONE = none);
//   bsq[k]   sqrt(q) of the owner (backward only).
template <bool KEEP_SQ>
__device__ __forceinline__ void tile_min(unsigned long long mask, int J, const float4 sph,
                                         const TileGeom &g, float best[4], int owner[4],
                                         float bsq[4]) {
  // torch.min over the J maps keeps the FIRST index at the minimum, so the maps are merged in index order: a
  // candidate's map is its hit depth or the background (a miss), a culled sphere's is the background at every
  // pixel of the tile -- the lowest culled index c0 takes its turn between the candidates below and above it.
  // (Only then is a hit at exactly 100.0 attributed as the reference does: it owns the pixel -- and gets its
  // gradient -- iff no lower index holds 100 there.)
  const unsigned long long all = J >= 64 ? ~0ull : ((1ull << J) - 1ull);
  const unsigned long long culled = ~mask & all;
  const int c0 = culled ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(culled)) : J;
  bool pending = culled != 0ull;   // wave-uniform
#pragma unroll
  for (int k = 0; k < 4; k++) {
    best[k] = __builtin_inff();
    owner[k] = SHR_ARGMIN_NONE;
    if (KEEP_SQ) bsq[k] = 1.0f;
  }
  auto miss = [&](int k) {
    if (kBackground < best[k]) {   // (a NaN minimum stays)
      best[k] = kBackground;
      owner[k] = SHR_ARGMIN_NONE;
    }
  };
  while (mask) {
    const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask));
    mask &= mask - 1;
    if (pending && j > c0) {
#pragma unroll
      for (int k = 0; k < 4; k++) miss(k);
      pending = false;
    }
    const float sx = readlane_f(sph.x, j), sy = readlane_f(sph.y, j);
    const float sz = readlane_f(sph.z, j), sr = readlane_f(sph.w, j);
    const float rr = sr * sr;
    const float dy = g.yg - sy;
    const float dy2 = dy * dy;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float dx = g.xg[k] - sx;
      const float q = (rr - dx * dx) - dy2;
      const bool hit = !(q <= kHitMin);  // clamp(min)!=min; NaN counts as hit (render.py:41-42)
      if (hit) {
        const float sq = sqrtf(q);
        const float d = sz - sq;
        const bool take = (d < best[k]) || (d != d);  // torch.min: NaN wins, ties keep first
        if (take) {
          best[k] = d;
          owner[k] = j;
          if (KEEP_SQ) bsq[k] = sq;
        }
      } else {
        miss(k);
      }
    }
  }
  if (pending) {
#pragma unroll
    for (int k = 0; k < 4; k++) miss(k);
  }
}

// --------------------------------------------------------------------------
// Forward over the tiles tile_begin, tile_begin+tile_step, ... < tile_end of one
// crop.  `sph` = this lane's sphere (lane < J).
template <bool VEC4, bool WRITE_ARG>
__device__ __forceinline__ void tile_forward(const float4 sph, int J, int H, int W,
                                             float *__restrict__ out, uint8_t *__restrict__ aout,
                                             int tiles_x, int tile_begin, int tile_end, int tile_step,
                                             int lane) {
  const bool valid = lane < J;
  const Axis ax = make_axis(W), ay = make_axis(H);
  for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
    const TileGeom g = tile_geom(tile, tiles_x, ax, ay, lane);
    const unsigned long long mask = tile_candidates(sph, valid, g, ax, ay, H, W);
    float best[4], bsq[4];
    int owner[4];
    tile_min<false>(mask, J, sph, g, best, owner, bsq);
    if (g.v >= H) continue;
    const size_t base = (size_t)g.v * W + g.u0;
    if (VEC4) {
      if (g.u0 < W) {
        *reinterpret_cast<float4 *>(out + base) = make_float4(best[0], best[1], best[2], best[3]);
        if (WRITE_ARG)
          *reinterpret_cast<uchar4 *>(aout + base) =
              make_uchar4((uint8_t)owner[0], (uint8_t)owner[1], (uint8_t)owner[2], (uint8_t)owner[3]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (g.u0 + k < W) {
          out[base + k] = best[k];
          if (WRITE_ARG) aout[base + k] = (uint8_t)owner[k];
        }
    }
  }
}

template <bool VEC4, bool WRITE_ARG>
__global__ void __launch_bounds__(1024)
sphere_tile_fwd_kernel(const float4 *__restrict__ spheres, int J, int H, int W,
                       float *__restrict__ depth, uint8_t *__restrict__ argmin, int tiles_x,
                       int ntiles) {
  __shared__ float4 s_sph[SHR_MAX_SPHERES];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  if (threadIdx.x < J) s_sph[threadIdx.x] = spheres[(size_t)n * J + threadIdx.x];
  __syncthreads();
  const float4 sph = lane < J ? s_sph[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  tile_forward<VEC4, WRITE_ARG>(sph, J, H, W, depth + (size_t)n * H * W,
                                WRITE_ARG ? argmin + (size_t)n * H * W : nullptr, tiles_x,
                                blockIdx.y * nwaves + wave, ntiles, nwaves * gridDim.y, lane);
}

// --------------------------------------------------------------------------
// Backward.  grid = (N, 1): the whole crop is reduced inside one workgroup.
template <bool VEC4>
__global__ void __launch_bounds__(1024)
sphere_tile_bwd_kernel(const float4 *__restrict__ spheres, const float *__restrict__ grad_depth,
                         int J, int H, int W, float4 *__restrict__ grad_spheres, int tiles_x,
                         int ntiles) {
  __shared__ float4 s_sph[SHR_MAX_SPHERES];
  __shared__ float4 s_acc[16 * SHR_MAX_SPHERES];  // [wave][sphere]
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

  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const TileGeom g = tile_geom(tile, tiles_x, ax, ay, lane);
    const unsigned long long mask = tile_candidates(sph, valid, g, ax, ay, H, W);
    if (mask == 0) continue;  // wave-uniform: nothing can be hit in this tile

    // upstream gradient of this lane's pixels, issued before the min loop
    float gk[4] = {0.f, 0.f, 0.f, 0.f};
    const bool row_ok = g.v < H;
    const size_t base = (size_t)g.v * W + g.u0;
    if (VEC4) {
      if (row_ok && g.u0 < W) {
        const float4 t = *reinterpret_cast<const float4 *>(gin + base);
        gk[0] = t.x; gk[1] = t.y; gk[2] = t.z; gk[3] = t.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (row_ok && g.u0 + k < W) gk[k] = gin[base + k];
    }

    float best[4], bsq[4];
    int owner[4];
    tile_min<true>(mask, J, sph, g, best, owner, bsq);

    // per-pixel partials  g * ( -dx/sq, -dy/sq, 1, -r/sq )
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
      if (__ballot(any) == 0) continue;  // candidate owns no pixel of the tile
      sx = wave_sum_lane63(sx);
      sy = wave_sum_lane63(sy);
      sz = wave_sum_lane63(sz);
      sw = wave_sum_lane63(sw);
      if (lane == 63) {
        float4 a = acc[j];
        a.x += sx; a.y += sy; a.z += sz; a.w += sw;
        acc[j] = a;
      }
    }
  }
  __syncthreads();
  // combine the waves' rows in wave order; d/dr = r * sum(-g/sq)
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
