// timestamped copy of sphere_zbuf_fwd_kernel (generated)
#include "../spherehand_amd/csrc/sphere_zbuf.h"
namespace shr {
template <bool OWNER, bool VEC4, bool POW2>
__global__ void __launch_bounds__(1024)
exp_zfwd_t(const float4 *__restrict__ spheres, int J, int H, int W,
                       float *__restrict__ depth, uint8_t *__restrict__ argmin, int rows_per_region,
                       int w4_shift, long long *tbuf) {
  const long long T0 = clock64();
  using Key = typename KeyOf<OWNER>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_sph = reinterpret_cast<float4 *>(smem);
  int4 *s_items = reinterpret_cast<int4 *>(smem + kOffItems);
  int *s_flag = reinterpret_cast<int *>(smem + kOffFlags);
  Key *zbuf = reinterpret_cast<Key *>(smem + kHdrBytes);

  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwaves = nthr >> 6;
  const int r0 = blockIdx.y * rows_per_region;
  const int r1 = min(H, r0 + rows_per_region);
  const int rh = r1 - r0;
  const int LW = W + kRowPad;
  const Axis ax = make_axis(W), ay = make_axis(H);

  if (wave == 0) {
    const bool valid = lane < J;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) s = spheres[(size_t)n * J + lane];
    s_sph[lane] = s;
    // general path unless every sphere is tame and at least one has z <= 100: a pixel's
    // minimum can exceed the background only where ALL J spheres hit it, and there the
    // sphere with z <= 100 contributes z - sqrt(q) < 100, so min(100, hits) is exact.
    const unsigned long long bad = __ballot(valid && !(sphere_is_tame(s) && fabsf(s.z) < 1e30f));
    const unsigned long long low = __ballot(valid && s.z <= kBackground);
    const int total = build_work_list(s, valid, ax, ay, W, r0, r1, s_items, lane);
    if (lane == 0) {
      s_flag[0] = (bad != 0ull) || (low == 0ull);
      s_flag[1] = total;
    }
  }
  const long long T1 = clock64();
  {  // background everywhere (pad rows/columns included)
    const Key bg = OWNER ? (Key)(((unsigned long long)depth_key(kBackground) << 32) | SHR_ARGMIN_NONE)
                         : (Key)depth_key(kBackground);
    constexpr int per16 = 16 / sizeof(Key);
    const int ncell = (rh + kPadRows) * LW;
    const int nvec = ncell / per16;
    if (OWNER) {
      const ulonglong2 v = make_ulonglong2(bg, bg);
      for (int i = tid; i < nvec; i += nthr) reinterpret_cast<ulonglong2 *>(zbuf)[i] = v;
    } else {
      const uint4 v = make_uint4((uint32_t)bg, (uint32_t)bg, (uint32_t)bg, (uint32_t)bg);
      for (int i = tid; i < nvec; i += nthr) reinterpret_cast<uint4 *>(zbuf)[i] = v;
    }
    for (int i = nvec * per16 + tid; i < ncell; i += nthr) zbuf[i] = bg;
  }
  const long long T2 = clock64();
  __syncthreads();
  const long long T3 = clock64();

  float *out = depth + (size_t)n * H * W;
  uint8_t *aout = OWNER ? argmin + (size_t)n * H * W : nullptr;

  if (s_flag[0]) {  // workgroup-uniform: this crop needs the general path
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int t0 = (r0 / kTileH) * tiles_x, t1 = ((r1 + kTileH - 1) / kTileH) * tiles_x;
    const float4 sph = lane < J ? s_sph[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    tile_forward<VEC4, OWNER>(sph, J, H, W, out, aout, tiles_x, t0 + wave, t1, nwaves, lane);
    return;
  }

  // ---- scan-convert the patch list -------------------------------------------------
  // A patch may overhang the box, the image's right edge or the region's last row:
  // the hit test is exact for ANY pixel, overhanging lanes land in LDS padding.
  {
    const int lx = lane & (kPatchW - 1), ly = lane >> 4;
    for_each_patch(s_sph, s_items, J, s_flag[1], wave, nwaves, lane,
                   [&](int j, const float4 s, int pu, int pv, bool, bool) {
                     const int u = pu + lx, v = pv + ly;
                     const float dx = axis_coord_t<POW2>(ax, u) - s.x;
                     const float dy = axis_coord_t<POW2>(ay, v) - s.y;
                     const float q = (s.w * s.w - dx * dx) - dy * dy;
                     if (q > kHitMin) {
                       const float d = s.z - sqrt_rn(q);
                       Key *cell = zbuf + (v - r0) * LW + u;
                       if (OWNER)
                         atomicMin(reinterpret_cast<unsigned long long *>(cell),
                                   ((unsigned long long)depth_key(d) << 32) | (unsigned)j);
                       else
                         atomicMin(reinterpret_cast<unsigned int *>(cell), depth_key(d));
                     }
                   });
  }
  const long long T4 = clock64();
  __syncthreads();
  const long long T5 = clock64();

  // ---- stream the region out ---------------------------------------------------------
  if (VEC4) {
    const int w4 = W >> 2;
    const int nchunk = rh * w4;
    for (int c = tid; c < nchunk; c += nthr) {
      int v, u;
      if (w4_shift >= 0) { v = c >> w4_shift; u = (c & (w4 - 1)) << 2; }
      else { v = c / w4; u = (c - v * w4) << 2; }
      const Key *cell = zbuf + v * LW + u;
      float4 o;
      if (OWNER) {
        const ulonglong2 k01 = reinterpret_cast<const ulonglong2 *>(cell)[0];
        const ulonglong2 k23 = reinterpret_cast<const ulonglong2 *>(cell)[1];
        o = make_float4(key_depth((uint32_t)(k01.x >> 32)), key_depth((uint32_t)(k01.y >> 32)),
                        key_depth((uint32_t)(k23.x >> 32)), key_depth((uint32_t)(k23.y >> 32)));
        *reinterpret_cast<uchar4 *>(aout + (size_t)(r0 + v) * W + u) =
            make_uchar4((uint8_t)k01.x, (uint8_t)k01.y, (uint8_t)k23.x, (uint8_t)k23.y);
      } else {
        const uint4 k = *reinterpret_cast<const uint4 *>(cell);
        o = make_float4(key_depth(k.x), key_depth(k.y), key_depth(k.z), key_depth(k.w));
      }
      *reinterpret_cast<float4 *>(out + (size_t)(r0 + v) * W + u) = o;
    }
  } else {
    for (int p = tid; p < rh * W; p += nthr) {
      const int v = p / W, u = p - v * W;
      const Key k = zbuf[v * LW + u];
      if (OWNER) {
        out[(size_t)(r0 + v) * W + u] = key_depth((uint32_t)((unsigned long long)k >> 32));
        aout[(size_t)(r0 + v) * W + u] = (uint8_t)k;
      } else {
        out[(size_t)(r0 + v) * W + u] = key_depth((uint32_t)k);
      }
    }
  }
  const long long T6 = clock64();
  if (lane == 0) {
    long long *t = tbuf + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 16 + wave) * 8;
    t[0] = T0; t[1] = T1; t[2] = T2; t[3] = T3; t[4] = T4; t[5] = T5; t[6] = T6; t[7] = s_flag[1];
  }
}

}
extern "C" int exp_zfwd_t_launch(const float *spheres, int N, int J, int H, int W, float *depth, unsigned char *argmin,
                               int rows, long long *tbuf, void *stream) {
  using namespace shr;
  const size_t lds = kHdrBytes + (size_t)(rows + kPadRows) * (W + kRowPad) * 8;
  dim3 grid(N, (H + rows - 1) / rows), block(1024);
  int sh = 0; while ((1 << sh) < W / 4) sh++;
  hipFuncSetAttribute((const void *)exp_zfwd_t<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((exp_zfwd_t<true, true, true>), grid, block, lds, (hipStream_t)stream, (const float4 *)spheres, J, H, W, depth, argmin, rows, sh, tbuf);
  return (int)hipGetLastError();
}
