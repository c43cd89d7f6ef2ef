// sphere_raster.hip -- C-ABI entry points of the sphere-set depth rasterizer and
// the dispatch between the fast LDS z-buffer kernels (sphere_zbuf.h) and the
// general tile kernels (sphere_tile.h).
#include <stdlib.h>

#include "sphere_zbuf.h"

using namespace shr;

namespace {

struct Tuning {
  // depth-only forward: the LDS budget the ROW REGIONS are sized for.  A whole 128x128 crop per workgroup takes 73 KB
  // (7.8 vs 9.6 us at 40 KB), two per CU; a 256x256 crop is cut into two 128-row regions (142 KB at the full pitch ->
  // the box variant at half of the LDS, two per CU: 60 us for 1152 crops).  Round 4's 80 KB cut it into four 64-row
  // regions whose whole-region z-buffer + run table took 92 KB: ONE workgroup per CU and 109 us (tools/exp_fwd256.py).
  int fwd_lds_bytes = 160 * 1024;
  int fwd_owner_lds_bytes = 0;          // forward + owner map (64-bit keys); 0 = by batch size (below)
  int bwd_lds_bytes = 128 * 1024;       // backward staging (grad f32 + owner u8)
  int force_general = 0;                // 1: always the tile kernels (tests)
  int fwd_waves = 16;                   // waves per forward workgroup
  int fwd_shares = 0x24344464;          // work-list shares of the four wave age groups, oldest in the low byte (sum 256)
  int bwd_shares = 0x2c3a4654;
  int mse_box = -1;                     // fused kernel's box variant: -1 = by launch size, 0 = never, 1 = always, > 1 = with that much LDS
  int bwd_waves = 0;                    // waves per backward workgroup: 0 = by launch size, 8 or 16
  int fwd_zbuf_bytes = 0;               // forward z-buffer bytes per workgroup; 0 = by launch size (launch_zbuf_fwd_t)
  int persistent = 1;                   // 0: one workgroup per crop; 1: persistent workgroups when N exceeds the device; > 1: that many
  int run_table = -1;                   // whole-crop workgroups start their runs from an LDS table: -1 = when it fits, 0 = never
} g_tune;

constexpr int kMaxLds = 160 * 1024;

// rows per LDS region: H if the crop fits, else a multiple of 8 balancing the regions; 0 = infeasible
int pick_rows(int H, long long row_bytes, long long cap, long long hdr) {
  if (cap > kMaxLds) cap = kMaxLds;
  long long max_rows = (cap - hdr) / row_bytes;
  if (max_rows >= H) return H;
  max_rows &= ~7LL;
  if (max_rows < 8) return 0;
  const int nreg = (int)((H + max_rows - 1) / max_rows);
  int rows = (((H + nreg - 1) / nreg) + 7) & ~7;
  return rows > max_rows ? (int)max_rows : rows;
}

int log2_if_pow2(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) s++;
  return s;
}

// (the per-device bookkeeping -- AttrDone, allow_dynamic_lds, device_cus -- lives in common.h)
template <typename K>
hipError_t allow_big_lds(K kernel, AttrDone *done) { return allow_dynamic_lds(kernel, kMaxLds, done); }

int pick_waves(int ntiles) {
  int w = 16;
  while (w > 1 && w / 2 >= ntiles) w /= 2;
  return w;
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Workgroups along x for a launch over N crops: all of them when they are resident at once anyway; otherwise as many
// as the device holds at a time (CUs x workgroups per CU by LDS), each looping over its crops with the next crop's
// records prefetched (sphere_zbuf.h).  Needs a wave that is neither the list wave nor a background wave.
int num_cus() { return device_cus(); }

int persistent_grid(int N, int regions, size_t lds, int nwaves) {
  if (g_tune.persistent == 0 || nwaves <= kBgWaves + 1) return N;
  const int per_cu = (int)(kMaxLds / (lds ? lds : 1)) > 0 ? (int)(kMaxLds / lds) : 1;
  // two resident workgroups per CU already overlap one crop's prologue with another's scan: persistent workgroups
  // measured slower there (depth-only forward at 9216 crops: 5.5 vs 4.2 us per 256 crops, tools/exp_twocu.py)
  if (per_cu >= 2 && g_tune.persistent == 1) return N;
  long long cap = (long long)num_cus() * per_cu / (regions > 0 ? regions : 1);
  if (cap < 1) cap = 1;
  if (g_tune.persistent > 1) return N > g_tune.persistent ? g_tune.persistent : N;   // tests / experiments: explicit count
  // measured (MI355X, 128x128, tools/exp_persist.py): nothing to win at 4.5 crops per CU (1152), forward -2 % at 9
  // (2304), forward -8 % / backward -3 % at 36 (9216): the hardware already overlaps a workgroup's exit with the next
  // one's dispatch, what is saved is the records' first read
  return N >= 4 * cap ? (int)cap : N;
}

template <bool OWNER, bool VEC4, bool POW2, bool PERSIST, bool BOX, bool TABLE = false, bool SEG2 = false>
int launch_zbuf_fwd_p(const float4 *sp, int N, int J, int H, int W, float *depth, uint8_t *argmin, int rows, size_t lds,
                      int zcells, dim3 grid, int flags, hipStream_t s) {
  static AttrDone attr_done;
  auto k = sphere_zbuf_fwd_kernel<OWNER, VEC4, POW2, PERSIST, BOX, TABLE, SEG2>;
  const hipError_t e = allow_big_lds(k, &attr_done);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k, grid, dim3(64 * g_tune.fwd_waves), lds, s, sp, N, J, H, W, depth, argmin, rows,
                     (log2_if_pow2(W / 4) & 0xff) | (flags << 8) | (g_tune.fwd_waves << 16), g_tune.fwd_shares, zcells,
                     make_axis_k(W, H));
  return (int)hipGetLastError();
}

template <bool OWNER, bool VEC4, bool POW2>
int launch_zbuf_fwd_t(const float4 *sp, int N, int J, int H, int W, float *depth, uint8_t *argmin, int rows,
                      int flags, hipStream_t s) {
  // LDS per workgroup.  `full` holds any touched box of a region in one pass (pick_rows sized the regions for it).
  // With at least two workgroups per CU in the launch a workgroup gets HALF of the CU's LDS instead: two resident
  // workgroups overlap each other's prologue, scan conversion and stream-out (depth-only forward at 9216 crops:
  // 5.9 -> 4.2 us per 256 crops, tools/exp_twocu.py); a hand crop's box (~63 x 62 px of 128 x 128) fits it in one
  // pass, a larger one takes two.
  const size_t key = OWNER ? 8 : 4;
  const int regions = (H + rows - 1) / rows;
  const size_t pitch = (size_t)max_box_pitch(W);
  const size_t full = kHdrBytes + (size_t)rows * pitch * key, least = kHdrBytes + 8 * pitch * key, half = kMaxLds / 2;
  size_t lds = full;
  if (g_tune.fwd_zbuf_bytes > 0) lds = kHdrBytes + (size_t)g_tune.fwd_zbuf_bytes;
  else if (POW2 && full > half && (long long)N * regions >= 2LL * num_cus()) lds = half;   // (the other instantiations take 84 SGPRs: one workgroup per CU whatever the LDS)
  if (lds < least) lds = least;
  if (lds > full) lds = full;
  // A workgroup with the full budget holds its whole region at the image's own pitch: nothing is derived from a box
  // (BOX = false; measured on one box, batch 256: 7.64 us against 8.15 with the box bookkeeping).
  const bool box = lds < full;
  if (!box) lds = kHdrBytes + (size_t)rows * (W + kRowPad) * key;
  const int zcells = (int)((lds - kHdrBytes) / key);
  dim3 grid((unsigned)persistent_grid(N, regions, lds, g_tune.fwd_waves), (unsigned)regions);
  if (box) {
    // images from 192 pixels on: the instantiation that packs boxes 33 .. 64 columns wide as two segments (sphere_zbuf.h SEG2)
    if constexpr (VEC4 && POW2)
      if ((H >= 192 || W >= 192) && (int)grid.x >= N)
        return launch_zbuf_fwd_p<OWNER, true, true, false, true, false, true>(sp, N, J, H, W, depth, argmin, rows, lds, zcells, grid, flags, s);
    return (int)grid.x < N
               ? launch_zbuf_fwd_p<OWNER, VEC4, POW2, true, true>(sp, N, J, H, W, depth, argmin, rows, lds, zcells, grid, flags, s)
               : launch_zbuf_fwd_p<OWNER, VEC4, POW2, false, true>(sp, N, J, H, W, depth, argmin, rows, lds, zcells, grid, flags, s);
  }
  // Whole-region z-buffer: when the run table (J x 64 entries of 8 bytes, sphere_zbuf.h build_run_table) fits behind it
  // and the workgroup has a wave to spare for it, the runs start from LDS
  if constexpr (VEC4 && POW2) {
    const size_t tab = (size_t)J * kWave * sizeof(uint2);
    static_assert(kZWaves == 16 && kBgWaves == 7, "the table kernel's wave roles");
    // (not when the table is what keeps a second workgroup off the CU in a launch that has two per CU: 1152 crops
    // @256x256 in 64-row regions, 109 us with it, 70 without -- the table pays one workgroup's runs, not its neighbour)
    const bool costs_a_neighbour = lds <= half && lds + tab > half && (long long)N * regions >= 2LL * num_cus();
    if (g_tune.run_table != 0 && (g_tune.run_table > 0 || !costs_a_neighbour) && (int)grid.x >= N &&
        g_tune.fwd_waves == kZWaves && lds + tab <= (size_t)kMaxLds &&
        ((size_t)zcells * key) % 16 == 0 && rows < 4096 && (long long)H * (W + kRowPad) < (1LL << 24))
      return launch_zbuf_fwd_p<OWNER, true, true, false, false, true>(sp, N, J, H, W, depth, argmin, rows, lds + tab, zcells, grid,
                                                                      flags, s);
  }
  return (int)grid.x < N
             ? launch_zbuf_fwd_p<OWNER, VEC4, POW2, true, false>(sp, N, J, H, W, depth, argmin, rows, lds, zcells, grid, flags, s)
             : launch_zbuf_fwd_p<OWNER, VEC4, POW2, false, false>(sp, N, J, H, W, depth, argmin, rows, lds, zcells, grid, flags, s);
}

template <bool OWNER, bool VEC4>
int launch_zbuf_fwd(const float4 *sp, int N, int J, int H, int W, float *depth, uint8_t *argmin, int rows,
                    int flags, hipStream_t s) {
  return (is_pow2(W) && is_pow2(H))
             ? launch_zbuf_fwd_t<OWNER, VEC4, true>(sp, N, J, H, W, depth, argmin, rows, flags, s)
             : launch_zbuf_fwd_t<OWNER, VEC4, false>(sp, N, J, H, W, depth, argmin, rows, flags, s);
}

template <bool VEC4, bool POW2, bool PERSIST, int NW, bool WHOLE, bool SEG2 = false>
int launch_zbuf_bwd_p(const float4 *sp, const float *grad, const uint8_t *argmin, int N, int J, int H, int W, float4 *gs,
                      int rows, size_t lds, int gridx, hipStream_t s) {
  static AttrDone attr_done;
  auto k = sphere_zbuf_bwd_kernel<VEC4, POW2, PERSIST, NW, WHOLE, SEG2>;
  const hipError_t e = allow_big_lds(k, &attr_done);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k, dim3((unsigned)gridx), dim3(64 * NW), lds, s, sp, grad, argmin, N, J, H, W, gs, rows,
                     log2_if_pow2(W / 4), g_tune.bwd_shares, make_axis_k(W, H));
  return (int)hipGetLastError();
}

template <bool VEC4, bool POW2>
int launch_zbuf_bwd_t(const float4 *sp, const float *grad, const uint8_t *argmin, int N, int J, int H, int W,
                      float4 *gs, int rows, hipStream_t s) {
  // `rows` = the rows of a crop the staging buffers hold at the full budget (pick_rows).  With at least two
  // workgroups per CU in the launch a workgroup gets half of the CU's LDS and 8 waves instead (two resident
  // workgroups overlap one crop's staging with the other's walk): the touched rows of 9 hand crops of 10 still
  // fit in one pass, the others take two.
  const size_t row_bytes = (size_t)(W + kRowPad) * 5, fixed = kHdrBytes + kPartBytes, half = kMaxLds / 2;
  size_t lds = fixed + (size_t)rows * row_bytes;
  int waves = g_tune.bwd_waves;
  if (waves == 0) waves = (lds > half && (long long)N >= 2LL * num_cus() && half >= fixed + 8 * row_bytes) ? 8 : 16;
  if (waves == 8 && lds > half && half >= fixed + 8 * row_bytes) {
    rows = (int)((half - fixed) / row_bytes) & ~7;
    lds = fixed + (size_t)rows * row_bytes;
  }
  const int gridx = persistent_grid(N, 1, lds, kZWaves);   // (any backward workgroup has a prefetch wave to spare)
  if (waves == 8) {
    if constexpr (VEC4 && POW2)
      if ((H >= 192 || W >= 192) && gridx >= N)   // (two-segment boxes: see the forward)
        return launch_zbuf_bwd_p<true, true, false, 8, false, true>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s);
    return gridx < N ? launch_zbuf_bwd_p<VEC4, POW2, true, 8, false>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s)
                     : launch_zbuf_bwd_p<VEC4, POW2, false, 8, false>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s);
  }
  if (rows >= H)   // the buffers hold the whole crop: rows at their own index, no exchange of the touched rows
    return gridx < N ? launch_zbuf_bwd_p<VEC4, POW2, true, 16, true>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s)
                     : launch_zbuf_bwd_p<VEC4, POW2, false, 16, true>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s);
  return gridx < N ? launch_zbuf_bwd_p<VEC4, POW2, true, 16, false>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s)
                   : launch_zbuf_bwd_p<VEC4, POW2, false, 16, false>(sp, grad, argmin, N, J, H, W, gs, rows, lds, gridx, s);
}

template <bool VEC4>
int launch_zbuf_bwd(const float4 *sp, const float *grad, const uint8_t *argmin, int N, int J, int H, int W,
                    float4 *gs, int rows, hipStream_t s) {
  return (is_pow2(W) && is_pow2(H)) ? launch_zbuf_bwd_t<VEC4, true>(sp, grad, argmin, N, J, H, W, gs, rows, s)
                                    : launch_zbuf_bwd_t<VEC4, false>(sp, grad, argmin, N, J, H, W, gs, rows, s);
}

}  // namespace

extern "C" int shr_set_tuning(int key, int value) {
  switch (key) {
    case SHR_TUNE_FWD_LDS_BYTES: g_tune.fwd_lds_bytes = value > 0 ? value : Tuning().fwd_lds_bytes; return SHR_OK;   // 0: the default
    case SHR_TUNE_FWD_OWNER_LDS_BYTES: g_tune.fwd_owner_lds_bytes = value; return SHR_OK;
    case SHR_TUNE_BWD_LDS_BYTES: g_tune.bwd_lds_bytes = value; return SHR_OK;
    case SHR_TUNE_FORCE_GENERAL: g_tune.force_general = value; return SHR_OK;
    case SHR_TUNE_FWD_WAVES:
      if (value < 1 || value > 16) return SHR_EINVAL;
      g_tune.fwd_waves = value;
      return SHR_OK;
    case SHR_TUNE_MSE_BOX: if (value < -1) return SHR_EINVAL; g_tune.mse_box = value; return SHR_OK;
    case SHR_TUNE_BWD_WAVES: if (value != 0 && value != 8 && value != 16) return SHR_EINVAL; g_tune.bwd_waves = value; return SHR_OK;
    case SHR_TUNE_FWD_ZBUF_BYTES: if (value < 0) return SHR_EINVAL; g_tune.fwd_zbuf_bytes = value; return SHR_OK;
    case SHR_TUNE_PERSISTENT: if (value < 0) return SHR_EINVAL; g_tune.persistent = value; return SHR_OK;
    case SHR_TUNE_FWD_RUN_TABLE: if (value < -1 || value > 1) return SHR_EINVAL; g_tune.run_table = value; return SHR_OK;
    case SHR_TUNE_D2M_TILED: return d2m_set_tiled(value);
    case SHR_TUNE_TRI_BAND: if (value < -1) return SHR_EINVAL; return tri_set_band(value);
    case SHR_TUNE_MESH_BAND: if (value < 0 || value > 1) return SHR_EINVAL; return mesh_set_band(value);
    case SHR_TUNE_D2M_WAVES: return d2m_set_waves(value);
    case SHR_TUNE_D2M_BAND_UNITS: return d2m_set_band_units(value);
    case SHR_TUNE_FWD_SHARES:
    case SHR_TUNE_BWD_SHARES: {
      // four bytes, oldest group first; normalised to sum 256 with every group >= 1
      int b[4], sum = 0;
      for (int g = 0; g < 4; g++) { b[g] = (value >> (8 * g)) & 255; sum += b[g]; }
      if (sum == 0) return SHR_EINVAL;
      int acc = 0, packed = 0;
      for (int g = 0; g < 4; g++) {
        int v = g == 3 ? 256 - acc : (b[g] * 256 + sum / 2) / sum;
        if (v < 1) v = 1;
        if (g < 3 && acc + v > 256 - (3 - g)) v = 256 - (3 - g) - acc;
        if (v > 255) v = 255;
        acc += v;
        packed |= v << (8 * g);
      }
      if (acc != 256) return SHR_EINVAL;
      (key == SHR_TUNE_FWD_SHARES ? g_tune.fwd_shares : g_tune.bwd_shares) = packed;
      return SHR_OK;
    }
    default: return SHR_EINVAL;
  }
}

extern "C" int shr_sphere_raster_fwd(const float *spheres, int N, int J, int H, int W, float *depth,
                                     uint8_t *argmin, void *stream) {
  return shr_sphere_raster_fwd_ex(spheres, N, J, H, W, depth, argmin, 0, stream);
}

extern "C" int shr_sphere_raster_fwd_ex(const float *spheres, int N, int J, int H, int W, float *depth,
                                        uint8_t *argmin, int flags, void *stream) {
  if (N == 0) return SHR_OK;
  if (flags & ~SHR_RASTER_OWNER_TOUCHED_ROWS) return SHR_EINVAL;
  if (!spheres || !depth || N < 0 || J <= 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if (J > SHR_MAX_SPHERES || (long long)H * W > (1LL << 30)) return SHR_ETOOLARGE;
  if (((uintptr_t)spheres & 15u) != 0) return SHR_EINVAL;
  const bool vec4 = (W % 4 == 0) && (((uintptr_t)depth & 15u) == 0) &&
                    (!argmin || ((uintptr_t)argmin & 3u) == 0);
  hipStream_t s = (hipStream_t)stream;
  const float4 *sp = reinterpret_cast<const float4 *>(spheres);

  const long long row_bytes = (long long)max_box_pitch(W) * (argmin ? 8 : 4);   // the widest pitch a touched box can get
  // Measured (MI355X, 128x128): one whole-crop workgroup per CU beats two 80-KB half-crop
  // workgroups at every batch size (N = 256: 8.8 vs 10.7 us; N = 9216: 6.1 vs 6.6 us per 256
  // crops): a half-crop workgroup repeats the prologue and splits the spheres that straddle
  // the cut.
  const int owner_cap = g_tune.fwd_owner_lds_bytes ? g_tune.fwd_owner_lds_bytes : kMaxLds;
  const int rows = g_tune.force_general
                       ? 0
                       : pick_rows(H, row_bytes, argmin ? owner_cap : g_tune.fwd_lds_bytes,
                                   kHdrBytes + kPadRows * row_bytes);
  if (rows > 0 && (H + rows - 1) / rows <= 65535 && W <= kMaxFastWidth && H <= 32768) {
    if (argmin) return vec4 ? launch_zbuf_fwd<true, true>(sp, N, J, H, W, depth, argmin, rows, flags, s)
                            : launch_zbuf_fwd<true, false>(sp, N, J, H, W, depth, argmin, rows, flags, s);
    return vec4 ? launch_zbuf_fwd<false, true>(sp, N, J, H, W, depth, argmin, rows, 0, s)
                : launch_zbuf_fwd<false, false>(sp, N, J, H, W, depth, argmin, rows, 0, s);
  }

  // general tile kernels: 4 waves per workgroup, one 32x8 tile per wave
  const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
  const int ntiles = tiles_x * tiles_y;
  const int nwaves = ntiles >= 4 ? 4 : pick_waves(ntiles);
  int slices = (ntiles + nwaves - 1) / nwaves;
  if (slices > 65535) slices = 65535;
  dim3 grid((unsigned)N, (unsigned)slices), block(64 * nwaves);
#define LAUNCH(V, A) \
  hipLaunchKernelGGL((sphere_tile_fwd_kernel<V, A>), grid, block, 0, s, sp, J, H, W, depth, argmin, tiles_x, ntiles)
  if (vec4) { if (argmin) LAUNCH(true, true); else LAUNCH(true, false); }
  else      { if (argmin) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return (int)hipGetLastError();
}

extern "C" int shr_sphere_raster_bwd(const float *spheres, const float *grad_depth, const uint8_t *argmin,
                                     int N, int J, int H, int W, float *grad_spheres, void *stream) {
  if (N == 0) return SHR_OK;
  if (!spheres || !grad_depth || !grad_spheres || N < 0 || J <= 0 || H <= 0 || W <= 0) return SHR_EINVAL;
  if (J > SHR_MAX_SPHERES || (long long)H * W > (1LL << 30)) return SHR_ETOOLARGE;
  if ((((uintptr_t)spheres | (uintptr_t)grad_spheres) & 15u) != 0) return SHR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float4 *sp = reinterpret_cast<const float4 *>(spheres);
  float4 *gs = reinterpret_cast<float4 *>(grad_spheres);

  if (argmin && !g_tune.force_general && W <= kMaxFastWidth && H <= 32768) {
    const long long row_bytes = (long long)(W + kRowPad) * 5;
    const int rows = pick_rows(H, row_bytes, g_tune.bwd_lds_bytes, kHdrBytes + kPartBytes + kPadRows * row_bytes);
    if (rows > 0) {
      const bool vec4 = (W % 4 == 0) && (((uintptr_t)grad_depth & 15u) == 0) && (((uintptr_t)argmin & 3u) == 0);
      return vec4 ? launch_zbuf_bwd<true>(sp, grad_depth, argmin, N, J, H, W, gs, rows, s)
                  : launch_zbuf_bwd<false>(sp, grad_depth, argmin, N, J, H, W, gs, rows, s);
    }
  }
  // no owner map (or image rows too wide for LDS): recompute the owners, tile kernel
  const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
  const int ntiles = tiles_x * tiles_y;
  const bool vec4 = (W % 4 == 0) && (((uintptr_t)grad_depth & 15u) == 0);
  const int nwaves = pick_waves(ntiles);
  dim3 grid((unsigned)N, 1), block(64 * nwaves);
  if (vec4)
    hipLaunchKernelGGL((sphere_tile_bwd_kernel<true>), grid, block, 0, s, sp, grad_depth, J, H, W, gs, tiles_x, ntiles);
  else
    hipLaunchKernelGGL((sphere_tile_bwd_kernel<false>), grid, block, 0, s, sp, grad_depth, J, H, W, gs, tiles_x, ntiles);
  return (int)hipGetLastError();
}

namespace {
// rows per region of the fused kernel: sized for the most partial sums (SHR_MAX_SPHERES), so that the number of
// regions -- part of the output layout -- depends on the image only
int mse_rows(int H, int W) {
  using namespace shr;
  if (H <= 0 || W <= 0 || W > kMaxFastWidth || H > 32768) return 0;
  return pick_rows(H, (long long)(W + kRowPad) * 8, kMaxLds, kHdrBytes + kPartBytes);
}
}  // namespace

extern "C" int shr_sphere_raster_mse_regions(int H, int W) {
  const int rows = mse_rows(H, W);
  return rows > 0 ? (H + rows - 1) / rows : 0;
}

extern "C" int shr_sphere_raster_mse(const float *spheres, int N, int J, int H, int W, const float *target,
                                     const int32_t *target_index, float *depth, float *sse_partial,
                                     float *grad_spheres_partial, void *stream) {
  return shr_sphere_raster_mse_indexed(spheres, nullptr, N, J, H, W, target, target_index, depth, sse_partial,
                                       grad_spheres_partial, stream);
}

static int sphere_raster_mse_launch(const float *spheres, const int32_t *crop_index, bool slot_by_crop, int N, int J, int H,
                                    int W, const float *target, const int32_t *target_index, float *depth,
                                    float *sse_partial, float *grad_spheres_partial, void *stream) {
  using namespace shr;
  const int by_crop = slot_by_crop ? 0x100 : 0;
  if (N == 0) return SHR_OK;
  if (!spheres || !target || !sse_partial || !grad_spheres_partial || N < 0 || J <= 0 || H <= 0 || W <= 0)
    return SHR_EINVAL;
  if (J > SHR_MAX_SPHERES || (long long)H * W > (1LL << 30)) return SHR_ETOOLARGE;
  if ((W % 4) != 0 || ((((uintptr_t)spheres | (uintptr_t)target | (uintptr_t)depth | (uintptr_t)grad_spheres_partial)) & 15u) != 0)
    return SHR_EINVAL;   // 16-byte rows only: compose shr_sphere_raster_fwd / _bwd otherwise
  const int rows = mse_rows(H, W);
  if (rows <= 0 || (H + rows - 1) / rows > 65535) return SHR_ETOOLARGE;
  hipStream_t s = (hipStream_t)stream;
  const size_t part = (size_t)kZWaves * J * 16;   // [wave][J] partial sums
  const size_t lds = kHdrBytes + part + (size_t)rows * (W + kRowPad) * 8;
  // (persistent workgroups measured no gain for this kernel -- its per-crop chain is longer and the observed image's
  // first read follows an index load -- so it stays at one workgroup per crop and region: PERSIST = false)
  const int regions = (H + rows - 1) / rows;
  dim3 grid((unsigned)N, (unsigned)regions);
  static AttrDone attr_a, attr_b, attr_c;
  // Two workgroups per CU when the launch has them (forward with owner map, 1152 crops @256x256: 157 -> 95 us with the
  // box z-buffer at half of the LDS, tools/exp_fwd256.py): the box variant, 64 VGPRs, zcells cells of z-buffer.
  const size_t half = kMaxLds / 2;
  const bool box_ok = is_pow2(W) && is_pow2(H) && W >= 32 && lds > half && (long long)N * regions >= 2LL * num_cus() &&
                      half >= kHdrBytes + part + 8 * (size_t)max_box_pitch(W) * 8;
  const int box = g_tune.mse_box < 0 ? (box_ok ? 1 : 0) : (g_tune.mse_box && is_pow2(W) && is_pow2(H) && W >= 32);
  if (box) {
    size_t blds = g_tune.mse_box > 1 ? (size_t)g_tune.mse_box : half;
    const size_t least = kHdrBytes + part + 8 * (size_t)max_box_pitch(W) * 8;
    if (blds < least) blds = least;
    if (blds > (size_t)kMaxLds) blds = kMaxLds;
    const int zcells = (int)((blds - kHdrBytes - part) / 8);
    static AttrDone attr_d;
    const bool seg2 = H >= 192 || W >= 192;        // (two-segment boxes: see the forward)
    auto k = seg2 ? sphere_zbuf_mse_box_kernel<true, true> : sphere_zbuf_mse_box_kernel<true, false>;
    const hipError_t e = allow_big_lds(k, seg2 ? &attr_d : &attr_c);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, grid, dim3(1024), blds, s, reinterpret_cast<const float4 *>(spheres), N, J, H, W, target,
                       target_index, rows, (log2_if_pow2(W / 4) & 0xff) | by_crop, zcells, g_tune.fwd_shares, g_tune.bwd_shares, depth,
                       sse_partial, reinterpret_cast<float4 *>(grad_spheres_partial), make_axis_k(W, H), crop_index);
  } else if (is_pow2(W) && is_pow2(H)) {
    auto k = sphere_zbuf_mse_kernel<true, false>;
    const hipError_t e = allow_big_lds(k, &attr_a);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, s, reinterpret_cast<const float4 *>(spheres), N, J, H, W, target,
                       target_index, rows, (log2_if_pow2(W / 4) & 0xff) | by_crop, g_tune.fwd_shares, g_tune.bwd_shares, depth, sse_partial,
                       reinterpret_cast<float4 *>(grad_spheres_partial), make_axis_k(W, H), crop_index);
  } else {
    auto k = sphere_zbuf_mse_kernel<false, false>;
    const hipError_t e = allow_big_lds(k, &attr_b);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, grid, dim3(1024), lds, s, reinterpret_cast<const float4 *>(spheres), N, J, H, W, target,
                       target_index, rows, (log2_if_pow2(W / 4) & 0xff) | by_crop, g_tune.fwd_shares, g_tune.bwd_shares, depth, sse_partial,
                       reinterpret_cast<float4 *>(grad_spheres_partial), make_axis_k(W, H), crop_index);
  }
  return (int)hipGetLastError();
}

#ifdef SHR_TIMELINE
// tools/headline_timeline.py only (never in the product build): the stamps of kernel `which` to host memory
extern "C" int shr_debug_timeline(int which, void *host, size_t bytes) {
  if (which < 0 || which > 2 || bytes > sizeof(shr::shr_tl[0])) return SHR_EINVAL;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(shr::shr_tl), bytes, (size_t)which * sizeof(shr::shr_tl[0]), hipMemcpyDeviceToHost);
}
extern "C" int shr_debug_timeline_rt(int which, void *host, size_t bytes) {
  if (which < 0 || which > 2 || bytes > sizeof(shr::shr_tl_rt[0])) return SHR_EINVAL;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(shr::shr_tl_rt), bytes, (size_t)which * sizeof(shr::shr_tl_rt[0]), hipMemcpyDeviceToHost);
}
extern "C" int shr_debug_timeline_clear(void) {
  static unsigned long long zeros[sizeof(shr::shr_tl) / 8];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(shr::shr_tl), zeros, sizeof(shr::shr_tl), 0, hipMemcpyHostToDevice);
}
#endif

extern "C" int shr_sphere_raster_mse_indexed(const float *spheres, const int32_t *crop_index, int N, int J, int H, int W,
                                             const float *target, const int32_t *target_index, float *depth,
                                             float *sse_partial, float *grad_spheres_partial, void *stream) {
  return sphere_raster_mse_launch(spheres, crop_index, false, N, J, H, W, target, target_index, depth, sse_partial,
                                  grad_spheres_partial, stream);
}

// `order`: a permutation of the N crops -- workgroup w works on crop order[w] and writes that CROP's slots: the results
// of shr_sphere_raster_mse, in another launch order (the caller's XCD placement: workgroups w, w + 8, ... share an L2)
extern "C" int shr_sphere_raster_mse_ordered(const float *spheres, const int32_t *order, int N, int J, int H, int W,
                                             const float *target, const int32_t *target_index, float *depth,
                                             float *sse_partial, float *grad_spheres_partial, void *stream) {
  if (!order) return SHR_EINVAL;
  return sphere_raster_mse_launch(spheres, order, true, N, J, H, W, target, target_index, depth, sse_partial,
                                  grad_spheres_partial, stream);
}
