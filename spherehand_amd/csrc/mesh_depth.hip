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
#include <cstdlib>

#include "fk_rows.h"

namespace shr {

// In-kernel timeline (tools/exp_mesh_phases.py builds this file alone with -DMESH_TL): s_memtime of every wave of the
// first 256 workgroups at the phase boundaries, read back through shr_mesh_debug_timeline.  Not in the product build.
#ifdef MESH_TL
__device__ unsigned long long mesh_tl[256 * 16 * 16];
#define MESH_STAMP(slot)                                                                                      \
  do {                                                                                                        \
    asm volatile("" ::: "memory");   /* (what precedes the stamp in the source is issued in front of it) */      \
    if ((threadIdx.x & 63) == 0 && blockIdx.x * gridDim.y + blockIdx.y < 256 && (gridDim.y == 1 || blockIdx.x == 0)) \
      mesh_tl[((blockIdx.x * gridDim.y + blockIdx.y) * 16 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    asm volatile("" ::: "memory");                                                                            \
  } while (0)
#define MESH_NOTE(slot, value)                                                                                \
  do {                                                                                                        \
    if ((threadIdx.x & 63) == 0 && blockIdx.x * gridDim.y + blockIdx.y < 256 && (gridDim.y == 1 || blockIdx.x == 0)) \
      mesh_tl[((blockIdx.x * gridDim.y + blockIdx.y) * 16 + (threadIdx.x >> 6)) * 16 + (slot)] = (unsigned long long)(value); \
  } while (0)
#else
#define MESH_STAMP(slot) do {} while (0)
#define MESH_NOTE(slot, value) do {} while (0)
#endif

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

// A face set up like the reference kernel (.cu:33-69): culled, vertices sorted by x, pixel
// box.  `live` = front-facing, non-degenerate and its box meets the image.
struct FaceSetup {
  float p[3][3];
  int xi_min, xi_max, r_lo, r_hi;
  bool live;
};

__device__ __forceinline__ FaceSetup face_setup_from(const float (&fv_)[9], int src) {
  // (opaque copies: see tri_raster.hip face_setup -- without them the sort's selects become loads from a scratch array)
  float fv[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { fv[k] = fv_[k]; asm("" : "+v"(fv[k])); }
  FaceSetup s;
  s.live = false;
  if ((fv[7] - fv[1]) * (fv[3] - fv[0]) < (fv[4] - fv[1]) * (fv[6] - fv[0])) return s;
  int p0, p2;
  if (fv[0] < fv[3]) { p0 = (fv[6] < fv[0]) ? 2 : 0; p2 = (fv[3] < fv[6]) ? 2 : 1; }
  else               { p0 = (fv[6] < fv[3]) ? 2 : 1; p2 = (fv[0] < fv[6]) ? 2 : 0; }
  int p1 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) if (p0 != k && p2 != k) p1 = k;
  const int order[3] = {p0, p1, p2};
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int o = order[a];
      s.p[a][d] = (o == 0) ? fv[d] : ((o == 1) ? fv[3 + d] : fv[6 + d]);
    }
  if (s.p[0][0] == s.p[2][0]) return s;
  s.xi_min = m_cvt_rz_sat(fmaxf(ceilf(s.p[0][0]), 0.f));
  s.xi_max = m_cvt_rz_sat(fminf(s.p[2][0], (float)src - 1.f));
  if (s.xi_min > s.xi_max) return s;
  const float ylo = fminf(fminf(s.p[0][1], s.p[1][1]), s.p[2][1]), yhi = fmaxf(fmaxf(s.p[0][1], s.p[1][1]), s.p[2][1]);
  // (a face whose largest x lies in (-1, 0) still reaches column 0 -- the reference truncates x2 towards zero,
  // .cu:69 -- and the span there is an EXTRApolation of the edges: any row)
  const bool wild = !(fabsf(ylo) < 1e9f) || !(fabsf(yhi) < 1e9f) || s.p[2][0] < 0.f;
  // A column's span ends are edge interpolations slope * (x - xa) + ya at an x inside the edge:
  // convex combinations of the vertices' y up to 4 roundings (<= 2.4e-7 * |y|); rows
  // [ceil(min), trunc(max)] (.cu:89-90; a span end in (-1, 0) truncates to row 0).
  const float yeps = 1e-5f * (fabsf(ylo) + fabsf(yhi)) + 1e-4f;
  s.r_lo = wild ? 0 : max(0, (int)ceilf(ylo - yeps));
  s.r_hi = wild ? src - 1 : min(src - 1, max(0, (int)floorf(yhi + yeps)));
  s.live = true;
  return s;
}

__device__ __forceinline__ FaceSetup face_setup_sorted(const float4 *__restrict__ verts, const int *__restrict__ faces,
                                                       int f, int src) {
  float fv[9];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float4 v = verts[faces[f * 3 + k]];
    fv[3 * k] = v.x; fv[3 * k + 1] = v.y; fv[3 * k + 2] = v.z;
  }
  return face_setup_from(fv, src);
}

constexpr int kMeshQueue = 3584;   // work items per round (14 KB next to the 66-KB slot array)
constexpr int kMeshFaces = 4;      // faces per thread and round (one round for the 3382-face hand mesh)
// Everything phase B needs of a face, computed ONCE in phase A (lanes = faces) and parked in LDS: the sorted
// vertices, the inverse barycentric matrix (9 IEEE divisions, .cu:62-66) and the three edge slopes (.cu:75-85; the
// reference divides per column, the quotient depends on the face only), so that a work item -- one sampled column
// of a face, ~3 per face -- starts from six 16-byte LDS reads instead of a vertex gather, the sort and 12 divisions
// (DepthRender B = 256, S = 128: 59 -> 3x us, round 3).  A face beyond the table's capacity keeps the old path
// (its items carry the face index and recompute): any mesh stays exact.
constexpr int kMeshRows = 768;
struct __attribute__((aligned(16))) FaceRow {
  float fi[9];        // inverse barycentric matrix / den
  float s01, s12, s02;   // (y1 - y0) / (x1 - x0), (y2 - y1) / (x2 - x1), (y2 - y0) / (x2 - x0)
  float x0, y0, x1, y1;  // sorted vertices 0 and 1 (the spans' base points)
  float z0, z1, z2;
  int xr;             // xi_min | xi_max << 16
  int yr;             // r_lo | r_hi << 16
  int flags;          // bit 0: x1 - x0 != 0, bit 1: x2 - x1 != 0
  int pad[2];
};
static_assert(sizeof(FaceRow) == 96, "six 16-byte reads per work item");

__device__ __forceinline__ FaceRow face_row(const FaceSetup &fs) {
  const float (&p)[3][3] = fs.p;
  FaceRow r;
  r.fi[0] = p[1][1] - p[2][1]; r.fi[1] = p[2][0] - p[1][0]; r.fi[2] = p[1][0] * p[2][1] - p[2][0] * p[1][1];
  r.fi[3] = p[2][1] - p[0][1]; r.fi[4] = p[0][0] - p[2][0]; r.fi[5] = p[2][0] * p[0][1] - p[0][0] * p[2][1];
  r.fi[6] = p[0][1] - p[1][1]; r.fi[7] = p[1][0] - p[0][0]; r.fi[8] = p[0][0] * p[1][1] - p[1][0] * p[0][1];
  const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1])) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
  for (int k = 0; k < 9; k++) r.fi[k] = r.fi[k] / den;
  const bool d01 = p[1][0] - p[0][0] != 0.f, d12 = p[2][0] - p[1][0] != 0.f;
  r.s01 = d01 ? (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) : 0.f;
  r.s12 = d12 ? (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]) : 0.f;
  r.s02 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]);
  r.x0 = p[0][0]; r.y0 = p[0][1]; r.x1 = p[1][0]; r.y1 = p[1][1];
  r.z0 = p[0][2]; r.z1 = p[1][2]; r.z2 = p[2][2];
  r.xr = fs.xi_min | (fs.xi_max << 16);   // (source sizes up to 32767: the launcher checks)
  r.yr = fs.r_lo | (fs.r_hi << 16);
  r.flags = (d01 ? 1 : 0) | (d12 ? 2 : 0);
  r.pad[0] = r.pad[1] = 0;
  return r;
}

// TO = output pixels per tile side, SL = source slots per output pixel and axis: 1 when the
// resize ratio is an odd integer (the bilinear weights are exactly (1, 0): S = 128 from
// 640), else 2.  Faces are taken 1024 at a time, in two phases:
//   A. lanes = faces: set-up + culls only (back faces, faces outside the tile, faces whose
//      box holds no sampled pixel: three quarters go); a survivor becomes one WORK ITEM per
//      sampled source COLUMN inside its x range (a deterministic block scan assigns the queue
//      slots), so a large face is spread over many lanes instead of stalling one wave (lanes
//      = whole faces: 230 us per 256 crops, bound by each crop's largest face) and the
//      lanes of a wave do not wait for each other's column loops (8-column items: 7.4 k VALU
//      instructions per wave, the union of the 64 lanes' nested loops);
//   B. lanes = work items: the expensive part (9 IEEE divisions per face, 3 per column, 7 per
//      pixel) with every lane busy and bounded work per lane.
// EXACT: the resize ratio R = src / S is an integer -- odd (SL = 1: output d samples source R d + (R - 1) / 2 with weight
// 1) or even (SL = 2: sources R d + R / 2 - 1 and the next one, weights 1/2 each; lin_index's fma is exact in both
// cases) -- and "the output pixels with a sample inside [lo, hi]" is a closed form instead of a loop over candidates.
template <int TO, int SL, bool EXACT>
__global__ void __launch_bounds__(1024)
mesh_depth_kernel(const float4 *__restrict__ vertices, const int *__restrict__ faces, int NV, int F, int src,
                  int S, float clamp_max, float *__restrict__ depth) {
  __shared__ uint32_t s_z[SL * TO][SL * TO + 1];   // [SL*dy + sy][SL*dx + sx], +1: bank spread
  __shared__ int s_queue[kMeshQueue];
  __shared__ int s_wave_cnt[16], s_wave_rows[16], s_next_item;
  __shared__ FaceRow s_rows[kMeshRows];
  const int b = blockIdx.y;
  const int tiles = (S + TO - 1) / TO;
  const int ty0 = (blockIdx.x / tiles) * TO, tx0 = (blockIdx.x % tiles) * TO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float scale = (float)src / (float)S;
  const float4 *verts = vertices + (size_t)b * NV;
  MESH_STAMP(0);

  for (int i = tid; i < SL * TO * (SL * TO + 1); i += blockDim.x) (&s_z[0][0])[i] = 0x447A0000u ^ 0x80000000u;  // 1000.0f

  // source range of the tile's slots (for the face/tile cull)
  const Lin tlx = lin_index(tx0, scale, src), thx = lin_index(min(tx0 + TO, S) - 1, scale, src);
  const Lin tly = lin_index(ty0, scale, src), thy = lin_index(min(ty0 + TO, S) - 1, scale, src);
  const float inv_scale = (float)S / (float)src;
  // output pixels whose slots can fall in a source interval: the inverse of the source-index
  // map, widened by one pixel; membership is re-checked exactly per slot
  auto out_lo = [&](int lo, int t0) { return max(t0, (int)floorf(((float)lo + 0.5f) * inv_scale - 0.5f) - 1); };
  auto out_hi = [&](int hi, int t0) { return min(min(t0 + TO, S) - 1, (int)ceilf(((float)hi + 0.5f) * inv_scale - 0.5f) + 1); };
  // bit sx set: output column dx reads source column (sx ? i1 : i0) inside [lo, hi] with a non-zero weight
  auto column_slots = [&](int dx, int lo, int hi) {
    const Lin l = lin_index(dx, scale, src);
    return ((l.l0 != 0.f && l.i0 >= lo && l.i0 <= hi) ? 1 : 0) | ((SL == 2 && l.l1 != 0.f && l.i1 >= lo && l.i1 <= hi) ? 2 : 0);
  };
  // does output pixel d sample a source index inside [lo, hi] with a non-zero weight?
  auto samples = [&](int d, int lo, int hi) {
    const Lin l = lin_index(d, scale, src);
    return (l.l0 != 0.f && l.i0 >= lo && l.i0 <= hi) || (SL == 2 && l.l1 != 0.f && l.i1 >= lo && l.i1 <= hi);
  };

  // SL == 1 (odd integer ratio R): output index d samples source index R d + h, h = (R - 1) / 2, with weight exactly 1
  // (lin_index's fma is exact there), so "the output pixels whose sample falls in [lo, hi]" is a closed form instead of
  // a loop over candidates: d in [ceil((lo - h) / R), floor((hi - h) / R)].  (x + 0.5) * (1 / R) is at least 0.5 / R
  // from an integer and the product's error below 1e-4 for |x| < 2^15: the floor is exact.
  // (even R, SL = 2: first slot R d + R / 2 - 1; a d counts when EITHER slot lies in [lo, hi] -- the slots' own
  // membership is still tested where they are rasterized)
  const int ratio = src / S, half_r = SL == 1 ? (ratio - 1) >> 1 : (ratio >> 1) - 1;
  const float rcp_r = 1.0f / (float)ratio;
  auto fdiv = [&](int x) { return (int)floorf(((float)x + 0.5f) * rcp_r); };   // floor(x / R), x may be negative
  auto first_out = [&](int lo, int t0) { return max(t0, fdiv(lo - (SL - 1) - half_r + ratio - 1)); };
  auto last_out = [&](int hi, int t0) { return min(min(t0 + TO, S) - 1, fdiv(hi - half_r)); };
  auto exact_lin = [&](int d) { Lin l; l.i0 = ratio * d + half_r; l.i1 = l.i0 + (SL - 1); l.l0 = SL == 1 ? 1.f : 0.5f; l.l1 = SL == 1 ? 0.f : 0.5f; return l; };

  for (int f0 = 0; f0 < F; f0 += 1024 * kMeshFaces) {
    // ---- A. cull, count work items, block scan ---------------------------------------------
    int nk[kMeshFaces], dx0[kMeshFaces];
    unsigned colmask[kMeshFaces];   // bit i: output column dx0 + i holds a sampled source column of the face
    int n = 0, nrow = 0;
#pragma unroll
    for (int k = 0; k < kMeshFaces; k++) {
      const int f = f0 + k * 1024 + tid;
      nk[k] = 0; dx0[k] = 0; colmask[k] = 0u;
      if (f < F) {
        const FaceSetup fs = face_setup_sorted(verts, faces, f, src);
        if (fs.live && !(fs.xi_max < tlx.i0 || fs.xi_min > thx.i1 || fs.r_hi < tly.i0 || fs.r_lo > thy.i1)) {
          if (EXACT) {   // closed forms: every output column / row in the range samples the box
            const int dx_lo = first_out(fs.xi_min, tx0), dx_hi = last_out(fs.xi_max, tx0);
            if (dx_lo <= dx_hi && first_out(fs.r_lo, ty0) <= last_out(fs.r_hi, ty0)) {
              dx0[k] = dx_lo;
              nk[k] = dx_hi - dx_lo + 1;
              colmask[k] = nk[k] >= 32 ? 0xffffffffu : ((1u << nk[k]) - 1u);
            }
          } else {
          const int dx_lo = out_lo(fs.xi_min, tx0), dx_hi = out_hi(fs.xi_max, tx0);
          bool cx = false, cy = false;   // any sampled column AND any sampled row inside the box?
          for (int dx = dx_lo; dx <= dx_hi && !cx; dx++) cx = samples(dx, fs.xi_min, fs.xi_max);
          for (int dy = out_lo(fs.r_lo, ty0); dy <= out_hi(fs.r_hi, ty0) && !cy; dy++) cy = samples(dy, fs.r_lo, fs.r_hi);
          if (cx && cy) {
            dx0[k] = dx_lo;
            for (int dx = dx_lo; dx <= dx_hi; dx++) {
              const bool c = column_slots(dx, fs.xi_min, fs.xi_max) != 0;
              nk[k] += c;
              if (c && dx - dx_lo < 32) colmask[k] |= 1u << (dx - dx_lo);
            }
          }
          }
        }
      }
      n += nk[k];
      nrow += nk[k] > 0;
    }
    // inclusive scans over the workgroup (work items, surviving faces): DPP inside rows of 16, SGPR row totals, LDS
    // wave totals.  Two separate counters: up to 4096 faces x 128 columns of items and up to 4096 surviving faces per
    // round do not fit one packed 32-bit word (a dense front-facing mesh has > 2048 survivors in one tile).
    auto wave_scan = [&](int v) {
      int incl = v;
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
      const int r0s = __builtin_amdgcn_readlane(incl, 15), r1s = __builtin_amdgcn_readlane(incl, 31);
      const int r2s = __builtin_amdgcn_readlane(incl, 47);
      const int row = lane >> 4;
      return incl + (row >= 1 ? r0s : 0) + (row >= 2 ? r1s : 0) + (row >= 3 ? r2s : 0);
    };
    const int incl = wave_scan(n), incl_rows = wave_scan(nrow);
    MESH_STAMP(1);   // this wave's culls and counts are done
    __syncthreads();   // the previous round's queue is consumed, s_wave_cnt is free
    if (lane == 63) { s_wave_cnt[wave] = incl; s_wave_rows[wave] = incl_rows; }
    __syncthreads();
    int off = incl - n, total = 0;
    int row = incl_rows - nrow, total_rows = 0;   // this thread's first row in the face table
    for (int w = 0; w < 16; w++) {
      const int c = s_wave_cnt[w], cr = s_wave_rows[w];
      if (w < wave) { off += c; row += cr; }
      total += c;
      total_rows += cr;
    }
    // ---- the surviving faces' rows: lanes = SURVIVORS (a quarter of the faces: computed where they were found, every
    // wave would run the divisions for each of its four face slots at a quarter of its lanes) -------------------
    int rowk[kMeshFaces];
#pragma unroll
    for (int k = 0; k < kMeshFaces; k++) {
      rowk[k] = -1;
      if (nk[k] == 0) continue;
      if (row < kMeshRows) {
        rowk[k] = row;
        s_rows[row].pad[0] = f0 + k * 1024 + tid;      // the face of this row (the row itself follows)
      }
      row++;
    }
    __syncthreads();
    MESH_STAMP(2);   // the scans are through, the survivors' rows are assigned
    for (int r = tid; r < min(kMeshRows, total_rows); r += blockDim.x)
      s_rows[r] = face_row(face_setup_sorted(verts, faces, s_rows[r].pad[0], src));
    MESH_STAMP(3);   // this wave's face rows stand
    MESH_NOTE(8, total);
    MESH_NOTE(9, total_rows);
    for (int w0 = 0; w0 < total; w0 += kMeshQueue) {
      if (w0 > 0) __syncthreads();
      {
        int slot = off - w0;
#pragma unroll
        for (int k = 0; k < kMeshFaces; k++) {
          if (nk[k] == 0) continue;
          const int f = f0 + k * 1024 + tid;
          // an item = (row in the face table, or face index | bit 31 beyond its capacity) << 7 | output column
          const int tag = rowk[k] >= 0 ? rowk[k] << 7 : (int)(0x80000000u | ((unsigned)f << 7));
          for (unsigned m = colmask[k]; m; m &= m - 1, slot++)
            if (slot >= 0 && slot < kMeshQueue) s_queue[slot] = tag | (dx0[k] + __builtin_ctz(m) - tx0);
          if (nk[k] > __builtin_popcount(colmask[k])) {   // a face wider than 32 output columns: the rest by re-enumeration
            const FaceSetup fs = face_setup_sorted(verts, faces, f, src);
            const int dx_hi = EXACT ? last_out(fs.xi_max, tx0) : out_hi(fs.xi_max, tx0);
            for (int dx = dx0[k] + 32; dx <= dx_hi; dx++)
              if (EXACT || column_slots(dx, fs.xi_min, fs.xi_max)) {
                if (slot >= 0 && slot < kMeshQueue) s_queue[slot] = tag | (dx - tx0);
                slot++;
              }
          }
        }
      }
      if (tid == 0) s_next_item = 0;
      if (w0 == 0) MESH_STAMP(4);   // this wave's queue entries are written (first round)
      __syncthreads();
      if (w0 == 0) MESH_STAMP(5);   // phase B starts (first round)
      // ---- B. rasterize the queued items (order is irrelevant: integer minima) ---------------
      // (64 items at a time from a counter: consecutive items are one face's columns and neighbouring faces -- with a
      // static stride the wave that holds the crop's large faces finished at 33 k cycles against a mean of 26 k)
      const int count = min(kMeshQueue, total - w0);
      for (;;) {
        int q0 = 0;
        if (lane == 0) q0 = atomicAdd(&s_next_item, 64);
        q0 = __builtin_amdgcn_readfirstlane(q0);
        if (q0 >= count) break;
        const int q = q0 + lane;
        if (q >= count) continue;
        const int item = s_queue[q];
        FaceRow r;
        if (item >= 0) r = s_rows[item >> 7];
        else r = face_row(face_setup_sorted(verts, faces, (int)(((unsigned)item & 0x7fffffffu) >> 7), src));
        const int xi_min = r.xr & 0xffff, xi_max = r.xr >> 16;
        const int dy_lo = out_lo(r.yr & 0xffff, ty0), dy_hi = out_hi(r.yr >> 16, ty0);
        {
          const int dx = tx0 + (item & 127);
          Lin lx;
          if (EXACT) lx = exact_lin(dx);
          else lx = lin_index(dx, scale, src);
#pragma unroll
          for (int sx = 0; sx < SL; sx++) {
            const int xi = sx ? lx.i1 : lx.i0;
            if (SL != 1 && ((!EXACT && (sx ? lx.l1 : lx.l0) == 0.f) || xi < xi_min || xi > xi_max)) continue;
            // ---- column span (.cu:72-90; the slopes are the face's) --------------------------
            const float xf = (float)xi;
            float yi1;
            if (xf <= r.x1) yi1 = (r.flags & 1) ? r.s01 * (xf - r.x0) + r.y0 : r.y1;
            else yi1 = (r.flags & 2) ? r.s12 * (xf - r.x1) + r.y1 : r.y1;
            const float yi2 = r.s02 * (xf - r.x0) + r.y0;
            const int yi_min = m_cvt_rz_sat(fmaxf(0.f, ceilf(fminf(yi1, yi2))));
            const int yi_max = m_cvt_rz_sat(fminf(fmaxf(yi1, yi2), (float)src - 1.f));
            // (only the output rows whose slots can fall inside the column's span; SL == 1: exactly the rows that
            // sample it, by the closed form -- the span lies inside the face's rows)
            const int ry_lo = yi_min > yi_max ? 1 : (EXACT ? first_out(yi_min, ty0) : max(dy_lo, out_lo(yi_min, ty0)));
            const int ry_hi = yi_min > yi_max ? 0 : (EXACT ? last_out(yi_max, ty0) : min(dy_hi, out_hi(yi_max, ty0)));
            for (int dy = ry_lo; dy <= ry_hi; dy++) {
              Lin ly;
              if (EXACT) ly = exact_lin(dy);
              else ly = lin_index(dy, scale, src);
#pragma unroll
              for (int sy = 0; sy < SL; sy++) {
                const int yi = sy ? ly.i1 : ly.i0;
                if (SL != 1 && ((!EXACT && (sy ? ly.l1 : ly.l0) == 0.f) || yi < yi_min || yi > yi_max)) continue;
                // ---- pixel (.cu:97-110) ----------------------------------------------------
                const float yf = (float)yi;
                float w[3], w_sum = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                  w[k] = (r.fi[3 * k] * xf + r.fi[3 * k + 1] * yf) + r.fi[3 * k + 2];
                  w[k] = fminf(fmaxf(w[k], 0.f), 1.f);
                  w_sum += w[k];
                }
#pragma unroll
                for (int k = 0; k < 3; k++) w[k] = w[k] / w_sum;
                const float zp = 1.0f / ((w[0] / r.z0 + w[1] / r.z1) + w[2] / r.z2);
                if (zp == zp) atomicMin(&s_z[SL * (dy - ty0) + sy][SL * (dx - tx0) + sx], mkey(zp));
              }
            }
          }
        }
      }
    }
  }
  MESH_STAMP(6);     // this wave found the queue empty
  __syncthreads();

  // ---- clamp + bilinear (mesh/render.py:286, :311; ATen upsample_bilinear2d) ---------------
  float *out = depth + (size_t)b * S * S;
  if (SL == 1 && (S & 3) == 0 && (TO & 3) == 0) {
    // four pixels per thread, one 16-byte write-through store (the map is read next by another kernel: left dirty in
    // the L2 it would be flushed by the end-of-kernel write-back, sphere_zbuf.h)
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    for (int i = tid; i < TO * TO / 4; i += blockDim.x) {
      const int oy = i / (TO / 4), ox = (i - oy * (TO / 4)) * 4;
      const int y = ty0 + oy, x = tx0 + ox;
      if (y >= S || x >= S) continue;
      v4u_t v;
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = __float_as_uint(fminf(mkey_inv(s_z[oy][ox + c]), clamp_max));
      float *dst = out + (size_t)y * S + x;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
  } else
  for (int i = tid; i < TO * TO; i += blockDim.x) {
    const int oy = i / TO, ox = i - oy * TO;
    const int y = ty0 + oy, x = tx0 + ox;
    if (y >= S || x >= S) continue;
    if (SL == 1) {   // weights are exactly (1, 0): ly.l0 * (lx.l0 * v) with both factors 1
      out[(size_t)y * S + x] = fminf(mkey_inv(s_z[oy][ox]), clamp_max);
    } else {
      const Lin lx = lin_index(x, scale, src), ly = lin_index(y, scale, src);
      float v[2][2];
#pragma unroll
      for (int sy = 0; sy < 2; sy++)
#pragma unroll
        for (int sx = 0; sx < 2; sx++) v[sy][sx] = fminf(mkey_inv((&s_z[0][0])[(SL * oy + sy) * (SL * TO + 1) + SL * ox + sx]), clamp_max);
      out[(size_t)y * S + x] = ly.l0 * (lx.l0 * v[0][0] + lx.l1 * v[0][1]) + ly.l1 * (lx.l0 * v[1][0] + lx.l1 * v[1][1]);
    }
  }
  MESH_STAMP(7);
}


// ---------------------------------------------------------------------------------------------------------------------
// The LATTICE kernel (round 5): integer resize ratios whose sampled source pixels form a lattice of at most 128 x 128
// (S = 128 from 640: every fifth row and column; S = 64: the pairs 10 d + 4, 10 d + 5), one workgroup per crop with the
// whole lattice as integer keys in LDS.  It is the triangle band kernel's structure (tri_raster.hip) on the lattice:
//   1  lanes = faces (all of them, four per thread): gather, the reference's culls, "does the box hold a lattice column
//      and a lattice row" -> the numbers of the surviving faces, compacted into one list (half of a hand's 3 382);
//   2  equal shares of the list for the 16 waves, 32 faces at a time: lanes = faces for the set-up (sort, inverse
//      barycentric matrix, slopes: face_row) parked in the wave's own LDS rows; lanes = the batch's lattice COLUMNS for
//      the span of rows (.cu:72-90) and the lattice rows inside it; those pixels queued level by level (level t = t-th
//      lattice row of every column that has one, compacted with a ballot); lanes = queued pixels, 64 at a time, every
//      lane busy, for the seven IEEE divisions of .cu:97-110 and the LDS minimum;
//   3  the tile kernel's epilogue: clamp, ATen's bilinear formula, 16-byte stores.
// mesh_depth_kernel (below this size: its tiles; any non-integer ratio) walks (face, column) items with a row loop per
// lane -- 68 batches x 3-4 iterations of ~140 instructions for a crop's ~4 800 sampled pixels, three lanes in ten busy
// (EXPERIMENTS R4) -- and rebuilds the faces its 768-row table cannot hold; here every face is set up once and the
// division chains run on full waves.  Same pixels, same arithmetic per pixel, integer minima: bit-identical images.
// WAVES x NF: waves per workgroup x faces per batch.  16 x 32 (3.6 KB of scratch per wave) or 12 x 64 (6.7 KB): full
// lanes in the set-up and fuller chunks and drains for a quarter fewer waves to hide latency with.
constexpr int kLatQueue = 128;                                  // queued pixels per wave (4-byte entries)
constexpr int lat_scratch_bytes(int nf) { return nf * (int)sizeof(FaceRow) + kLatQueue * 4 + 64; }   // rows | queue | start marks
constexpr int kLatMax = 128;                                    // lattice rows / columns (7 bits each in a queue entry)

__host__ __device__ inline size_t lattice_lds_bytes(int L, int F, int waves, int nf, int skinned_vertices = 0) {
  return (((size_t)L * (L + 1) * 4 + 15) & ~(size_t)15) + (size_t)((F + 7) & ~7) * 2 + (size_t)waves * lat_scratch_bytes(nf) +
         (size_t)skinned_vertices * 16;
}
// SKIN (shr_mesh_render_fwd: all of DepthRender.forward in one launch): the crop's vertices are skinned and projected by
// the workgroup itself into LDS (lbs_project_kernel's arithmetic, common.h) -- no vertex array in HBM, no second launch,
// and the culls' and the set-up's gathers are LDS reads.
struct LatticeSkin {
  const float *T;            // [B][NB][16]
  const int *vstart, *sbone;
  const float4 *swv;
  const float *rand_f;
  int NB, right_hand;
  float cx, cy, fx, fy;
};

// POST (shr_mesh_render_post_fwd: HandSynthesizer.forward's `* depth_scale` and DepthNoise, network/util_modules.py:112-116,
// :46-84, in the epilogue): the crop's whole lattice is in this workgroup's LDS, so output pixel (v, u) of the noised
// image -- the scaled image's pixel (v + dy, u + dx), clamped to the image, plus depth noise where that is foreground --
// is evaluated straight from the lattice; the draws come from the counter-based generator of common.h, keyed per
// sample (keys[2][B], drawn by shr_synth_pose_fwd) and per pixel.
struct LatticePost {
  float depth_scale;
  const uint32_t *keys;          // [2][B], or nullptr: no noise
  uint32_t t0, t1, t2;           // the shift's cumulative thresholds (common.h noise_shift)
  float sigma_z;
  unsigned long long *state;     // the generator's (seed, call counter, ticket), or nullptr: the counter is advanced here, by
  int B;                         // the launch that consumes the call's last draws -- by its LAST workgroup to finish
  // ONE-LAUNCH synthesizer (shr_hand_synth_fwd): params != nullptr -> the workgroup's first wave runs the crop's forward
  // kinematics + RandScale + draws (fk_rows.h) into LDS instead of reading T; uv_hm != nullptr -> the crop's heat-maps
  // (Hand3DHeatmapRender: key-point skinning + heat-map camera + paint, synth_post.hip's arithmetic) in the epilogue
  const float *params, *offset, *offset_inv;
  float rand_scale, rand_half;
  float *draws;                  // [6][B]
  const int *kp_start, *kp_bone;
  const float4 *kp_wv;
  int J, hm_shift;               // key-points (<= 64); log2 of the heat-map's side
  float hcx, hcy, hfx, hfy, hsigma, uv_scale, d_scale, a00, a03, a11, a13;
  float *uv_hm, *d_hm;
  float4 *xyz;
};

// the same tail for images that are in HBM already (sizes the lattice kernel does not take): same draws, same bits
__global__ void __launch_bounds__(256)
depth_post_kernel(const float *__restrict__ in, int B, int H, int W, LatticePost post, float *__restrict__ out) {
  const int b = blockIdx.y;
  const float *src = in + (size_t)b * H * W;
  const bool noise = post.keys != nullptr;
  const uint32_t key0 = noise ? post.keys[b] : 0u, key1 = noise ? post.keys[post.B + b] : 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    int v = i / W, u = i - v * W;
    if (noise) {
      const NoiseShift sh = noise_shift(key0, (uint32_t)i, post.t0, post.t1, post.t2);
      v = min(max(v + sh.dy, 0), H - 1);
      u = min(max(u + sh.dx, 0), W - 1);
    }
    float z = src[(size_t)v * W + u] * post.depth_scale;
    if (noise && z < 1.0f) z = z + noise_normal(key1, (uint32_t)i) * post.sigma_z;
    out[(size_t)b * H * W + i] = z;
  }
  if (post.state && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) post.state[1] += 1ull;
}

template <int SL, int kLatWaves, int kLatFaces, bool SKIN = false, bool POST = false>
__global__ void __launch_bounds__(kLatWaves * 64)
mesh_lattice_kernel(const float4 *__restrict__ vertices, const int *__restrict__ faces, int NV, int F, int src, int S,
                    float clamp_max, float *__restrict__ depth, LatticeSkin skin, LatticePost post) {
  static_assert(kLatFaces == 32 || kLatFaces == 64, "queue entries: 5 or 6 bits of face");
  constexpr int kFaceBits = kLatFaces == 64 ? 6 : 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char lat_smem[];
  const int L = SL * S, LP = L + 1;
  uint32_t *s_z = reinterpret_cast<uint32_t *>(lat_smem);                                   // [L][L + 1] keys
  const size_t zbytes = ((size_t)L * LP * 4 + 15) & ~(size_t)15;
  uint16_t *s_surv = reinterpret_cast<uint16_t *>(lat_smem + zbytes);        // the surviving faces' numbers
  unsigned char *s_scr = lat_smem + zbytes + (size_t)((F + 7) & ~7) * 2;
  __shared__ int s_nsurv;
  __shared__ Rot s_sc[POST ? kAngles : 1];         // (one-launch synthesizer: the FK wave's sincos table, the sample's draws,
  __shared__ float s_draw[POST ? 8 : 1];           //  the key-points in the heat-map's camera)
  __shared__ float4 s_kp[POST ? 64 : 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool own_pose = POST && post.params != nullptr, paint = POST && post.uv_hm != nullptr;
  float4 *s_verts = reinterpret_cast<float4 *>(s_scr + (size_t)kLatWaves * lat_scratch_bytes(kLatFaces));   // SKIN: [NV]
  const float4 *verts = SKIN ? s_verts : vertices + (size_t)b * NV;
  MESH_STAMP(0);
  if (tid == 0) s_nsurv = 0;
  // SKIN: the corners of this thread's faces of the first round of culls are requested now -- they depend on nothing, and
  // by the time the vertices stand in LDS the culls' only global round trip is over
  constexpr int kPer = kLatWaves >= 16 ? 4 : 5;   // faces per thread and round of culls: one round for the 3382-face hand mesh
  int corner[kPer][3];
  if (SKIN) {
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int f = k * kLatWaves * 64 + tid;
#pragma unroll
      for (int c = 0; c < 3; c++) corner[k][c] = f < F ? faces[f * 3 + c] : 0;
    }
  }
  if (SKIN) {
    // ---- 0. skinning + camera (mesh/render.py:320-329; tri_raster.hip lbs_project_kernel) ----------------------------
    float *s_T = reinterpret_cast<float *>(s_scr);                     // the bones' matrices, in the scratch nobody uses yet
    // A chain of round trips -- matrices, entry ranges, entries -- of which only the sum needs the matrices: a thread's
    // first two vertices' ranges and their first four entries each (a hand vertex has 1 .. 5) are requested before the
    // barrier that the matrices need.
    constexpr int kVerts = 2, kAhead = 4;
    int e0[kVerts], e1[kVerts], bone_a[kVerts][kAhead];
    float4 q_a[kVerts][kAhead];
#pragma unroll
    for (int j = 0; j < kVerts; j++) {
      const int v = tid + j * kLatWaves * 64;
      e0[j] = v < NV ? skin.vstart[v] : 0;
      e1[j] = v < NV ? skin.vstart[v + 1] : 0;
    }
    if (own_pose) {
      if (wave == 0) {      // pose -> diag(s) T straight into the LDS copy (NB = 17: the hand's bones)
        SynthDraws syn;
        syn.state = post.state; syn.rand_scale = post.rand_scale; syn.rand_half = post.rand_half; syn.draws = post.draws; syn.B = post.B;
        pose_transforms_wave<true>(post.params, post.offset, post.offset_inv, b, lane, s_sc, s_draw, reinterpret_cast<float4 *>(s_T), syn);
      }
    } else {
      for (int i = tid; i < skin.NB * 16; i += kLatWaves * 64) s_T[i] = skin.T[(size_t)b * skin.NB * 16 + i];
    }
    const bool has_rand = own_pose || skin.rand_f != nullptr;
    float rf = (!own_pose && has_rand) ? skin.rand_f[b] : 0.f;
#pragma unroll
    for (int j = 0; j < kVerts; j++)
#pragma unroll
      for (int k = 0; k < kAhead; k++) {
        const bool on = e0[j] + k < e1[j];
        bone_a[j][k] = on ? skin.sbone[e0[j] + k] : 0;
        q_a[j][k] = on ? skin.swv[e0[j] + k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    __syncthreads();
    if (own_pose) rf = s_draw[3];
    if (paint && tid < post.J) {       // the crop's key-points in the heat-map's camera (lbs_project_kernel's arithmetic)
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int e = post.kp_start[tid]; e < post.kp_start[tid + 1]; e++) lbs_add_entry(acc, s_T + post.kp_bone[e] * 16, post.kp_wv[e]);
      s_kp[tid] = lbs_finish(acc, skin.right_hand, 1, post.hcx, post.hcy, post.hfx, post.hfy, has_rand, rf);
    }
#pragma unroll
    for (int j = 0; j < kVerts; j++) {
      const int v = tid + j * kLatWaves * 64;
      if (v >= NV) continue;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < kAhead; k++)
        if (e0[j] + k < e1[j]) lbs_add_entry(acc, s_T + bone_a[j][k] * 16, q_a[j][k]);     // (ascending entries: the reference's sum)
      for (int e = e0[j] + kAhead; e < e1[j]; e++) lbs_add_entry(acc, s_T + skin.sbone[e] * 16, skin.swv[e]);
      s_verts[v] = lbs_finish(acc, skin.right_hand, 1, skin.cx, skin.cy, skin.fx, skin.fy, has_rand, rf);
    }
    for (int v = tid + kVerts * kLatWaves * 64; v < NV; v += kLatWaves * 64) {                // (meshes above 2 x 1024 vertices)
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int e = skin.vstart[v]; e < skin.vstart[v + 1]; e++) lbs_add_entry(acc, s_T + skin.sbone[e] * 16, skin.swv[e]);
      s_verts[v] = lbs_finish(acc, skin.right_hand, 1, skin.cx, skin.cy, skin.fx, skin.fy, has_rand, rf);
    }
    // (the barrier in front of phase 1 below orders these writes)
  }
  for (int i = tid; i < L * LP; i += kLatWaves * 64) s_z[i] = 0x447A0000u ^ 0x80000000u;  // 1000.0f

  // lattice index <-> source index.  SL = 1 (odd ratio R): c -> R c + (R - 1) / 2.  SL = 2 (even R): c = 2 d + s ->
  // R d + R / 2 - 1 + s.  first_lat(lo) = the first lattice index whose source index is >= lo, last_lat(hi) = the last
  // one whose source index is <= hi (-1: none); floor(x / R) through a float product as in mesh_depth_kernel.
  const int ratio = src / S, base = SL == 1 ? (ratio - 1) >> 1 : (ratio >> 1) - 1;
  const float rcp_r = 1.0f / (float)ratio;
  auto fdiv = [&](int x) { return (int)floorf(((float)x + 0.5f) * rcp_r); };   // floor(x / R), |x| < 2^15
  auto src_of = [&](int c) { return SL == 1 ? ratio * c + base : ratio * (c >> 1) + base + (c & 1); };
  auto first_lat = [&](int lo) {
    const int a = lo - base;
    if (a <= 0) return 0;
    if (SL == 1) return fdiv(a + ratio - 1);                   // ceil(a / R)
    const int d = fdiv(a), rem = a - d * ratio;
    return rem == 0 ? 2 * d : (rem == 1 ? 2 * d + 1 : 2 * d + 2);
  };
  auto last_lat = [&](int hi) {
    const int a = hi - base;
    if (a < 0) return -1;
    const int d = fdiv(a), rem = a - d * ratio;
    const int c = SL == 1 ? d : (rem >= 1 ? 2 * d + 1 : 2 * d);
    return min(c, L - 1);
  };
  MESH_STAMP(4);   // (SKIN: this wave's vertices are skinned) the lattice is initialised
  __syncthreads();
  MESH_NOTE(8, 0);   // (work items: none -- tools/exp_mesh_phases.py tells the lattice kernel by it)

  // ---- 1. culls; the survivors' numbers ---------------------------------------------------------------------------
  for (int f0 = 0; f0 < F; f0 += kLatWaves * 64 * kPer) {
    bool keep[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int f = f0 + k * kLatWaves * 64 + tid;
      keep[k] = false;
      if (f < F) {
        FaceSetup fs;
        if (SKIN && f0 == 0) {   // (corners requested at the kernel's start, vertices in LDS)
          const float4 a = verts[corner[k][0]], bq = verts[corner[k][1]], c = verts[corner[k][2]];
          const float fv[9] = {a.x, a.y, a.z, bq.x, bq.y, bq.z, c.x, c.y, c.z};
          fs = face_setup_from(fv, src);
        } else {
          fs = face_setup_sorted(verts, faces, f, src);
        }
        keep[k] = fs.live && first_lat(fs.xi_min) <= last_lat(fs.xi_max) && first_lat(fs.r_lo) <= last_lat(fs.r_hi);
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const unsigned long long m = __ballot(keep[k]);
      if (m == 0ull) continue;
      int at = 0;
      if (lane == 0) at = atomicAdd(&s_nsurv, __popcll(m));
      at = __builtin_amdgcn_readfirstlane(at);
      if (keep[k])
        s_surv[at + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
            (uint16_t)(f0 + k * kLatWaves * 64 + tid);
    }
  }
  MESH_STAMP(1);
  __syncthreads();
  MESH_STAMP(2);

  // ---- 2. the survivors, 32 at a time per wave ------------------------------------------------------------------------
  {
    FaceRow *s_face = reinterpret_cast<FaceRow *>(s_scr + (size_t)wave * lat_scratch_bytes(kLatFaces));
    uint32_t *s_queue = reinterpret_cast<uint32_t *>(s_scr + (size_t)wave * lat_scratch_bytes(kLatFaces) + kLatFaces * sizeof(FaceRow));
    unsigned char *s_mark = s_scr + (size_t)wave * lat_scratch_bytes(kLatFaces) + kLatFaces * sizeof(FaceRow) + kLatQueue * 4;
    const int n = s_nsurv;
    MESH_NOTE(9, n);
    // batches of 32 survivors dealt round robin (face numbers cluster: a wave's own contiguous share was all palm or all
    // finger tips, and the waves finished between 21 and 30 us).  Measured on top and not kept: equal strided shares in
    // equal batches (1 630 survivors = 102 per wave = four batches of 26 each, instead of three waves with a fourth full
    // batch): 29.1 -> 30.5 us -- a batch costs nearly the same with 26 faces as with 32; batches drawn from a
    // counter with the next batch's vertices requested a batch ahead (corner numbers in the list instead of face
    // numbers: one round trip) -- 29.9 -> 31.6 us: the kernel is VALU-bound (82 % busy), not waiting for its gathers.
    const int from = wave * kLatFaces;
    for (int at = from; at < n; at += kLatWaves * kLatFaces) {
      if (at == from + kLatWaves * kLatFaces) MESH_STAMP(5);   // this wave's second batch starts
      const int count = min(kLatFaces, n - at);
      int ncol = 0, c_lo = 0;
      FaceRow row;
      if (lane < count) {
        const FaceSetup fs = face_setup_sorted(verts, faces, (int)s_surv[at + lane], src);
        row = face_row(fs);
        c_lo = first_lat(fs.xi_min);
        ncol = max(0, last_lat(fs.xi_max) - c_lo + 1);
      }
      const int fincl = wave_scan_incl(ncol, lane);            // lanes = faces: the face's lattice columns end here
      const int ncols = __builtin_amdgcn_readlane(fincl, 63);
      if (lane < count) {
        row.pad[0] = c_lo - (fincl - ncol);                    // column item k of the face is lattice column k + this
        // the corners' z: their refined reciprocals for the pixels' divisions (common.h tri_pixel_depth) in the words this
        // kernel does not read (xr, yr, pad[1]), flags bit 2: all three tame
        const bool tame = div_tame_z(row.z0) && div_tame_z(row.z1) && div_tame_z(row.z2);
        row.xr = __float_as_int(tame ? div_rcp_refined(row.z0) : 0.f);
        row.yr = __float_as_int(tame ? div_rcp_refined(row.z1) : 0.f);
        row.pad[1] = __float_as_int(tame ? div_rcp_refined(row.z2) : 0.f);
        row.flags |= tame ? 4 : 0;
        s_face[lane] = row;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);                      // this wave's LDS writes (rows and queue are its own)
      __builtin_amdgcn_wave_barrier();
      if (at == from) MESH_STAMP(3);   // this wave's first batch: its faces are set up

      int qn = 0;   // queued pixels (wave-uniform)
      auto drain = [&](int take) {
        if (lane < take) {
          const uint32_t e = s_queue[qn - take + lane];
          const int cx = (int)((e >> kFaceBits) & 127u), cy = (int)(e >> (kFaceBits + 7));
          const float4 *r4 = reinterpret_cast<const float4 *>(&s_face[e & (uint32_t)(kLatFaces - 1)]);
          const float4 f0v = r4[0], f1v = r4[1], f2v = r4[2], zv = r4[4], iv = r4[5];   // . | z0 z1 z2 rz0 | rz1 flags pad0 rz2
          const float fi[9] = {f0v.x, f0v.y, f0v.z, f0v.w, f1v.x, f1v.y, f1v.z, f1v.w, f2v.x};
          // ---- pixel (.cu:97-110) ----------------------------------------------------
          const float xf = (float)src_of(cx), yf = (float)src_of(cy);
          float w[3], w_sum = 0.f;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            w[k] = (fi[3 * k] * xf + fi[3 * k + 1] * yf) + fi[3 * k + 2];
            w[k] = fminf(fmaxf(w[k], 0.f), 1.f);
            w_sum += w[k];
          }
          const float pz[3] = {zv.x, zv.y, zv.z}, rz[3] = {zv.w, iv.x, iv.w};
          const float zp = tri_pixel_depth(w[0], w[1], w[2], w_sum, pz, rz, (__float_as_int(iv.y) & 4) != 0);
          if (zp == zp) atomicMin(&s_z[cy * LP + cx], mkey(zp));
        }
        qn -= take;
      };
      for (int k0 = 0; k0 < ncols; k0 += 64) {
        const int k = k0 + lane;
        const bool colv = k < ncols;
        // the face of column item k: the faces whose first column lies in this chunk mark it (their number + 1, the
        // numbers rise with the position), a max-scan over the lanes spreads the marks, and the face that covers the
        // chunk's first column is one ballot -- a face holds ~3 lattice columns, nearly every run ends inside the chunk
        // (run_of's one-by-one loop over those ends was 1 000 cycles per chunk)
        s_mark[lane] = 0;
        const int start = fincl - ncol;
        if (ncol > 0 && start >= k0 && start < k0 + 64) s_mark[start - k0] = (unsigned char)(lane + 1);
        const int before = __popcll(__ballot(ncol > 0 && start < k0)) ;   // faces (with columns) that start before the chunk
        const int cover = before == 0 ? 0 : (int)(63 - __builtin_clzll(__ballot(ncol > 0 && start < k0)));   // the last of them
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        int mk = (int)s_mark[lane];
        mk = max(mk, __builtin_amdgcn_update_dpp(0, mk, 0x111, 0xF, 0xF, false));   // row_shr:1
        mk = max(mk, __builtin_amdgcn_update_dpp(0, mk, 0x112, 0xF, 0xF, false));   // row_shr:2
        mk = max(mk, __builtin_amdgcn_update_dpp(0, mk, 0x114, 0xF, 0xF, false));   // row_shr:4
        mk = max(mk, __builtin_amdgcn_update_dpp(0, mk, 0x118, 0xF, 0xF, false));   // row_shr:8
        {
          const int m0 = __builtin_amdgcn_readlane(mk, 15), m1 = __builtin_amdgcn_readlane(mk, 31), m2 = __builtin_amdgcn_readlane(mk, 47);
          const int rowi = lane >> 4;
          mk = max(mk, rowi >= 1 ? m0 : 0);
          mk = max(mk, rowi >= 2 ? m1 : 0);
          mk = max(mk, rowi >= 3 ? m2 : 0);
        }
        const int face = mk > 0 ? mk - 1 : cover;
        __builtin_amdgcn_wave_barrier();   // (the marks are rewritten by the next chunk)
        const float4 *r4 = reinterpret_cast<const float4 *>(&s_face[face]);
        const float4 sl = r4[2], pv = r4[3], iv = r4[5];       // . s01 s12 s02 | x0 y0 x1 y1 | yr flags pad0 pad1
        const int flags = __float_as_int(iv.y);
        const int cx = k + __float_as_int(iv.z);
        // ---- column span (.cu:72-90; the slopes are the face's) --------------------------
        const float xf = (float)src_of(cx);
        float yi1;
        if (xf <= pv.z) yi1 = (flags & 1) ? sl.y * (xf - pv.x) + pv.y : pv.w;
        else yi1 = (flags & 2) ? sl.z * (xf - pv.z) + pv.w : pv.w;
        const float yi2 = sl.w * (xf - pv.x) + pv.y;
        const int yi_min = m_cvt_rz_sat(fmaxf(0.f, ceilf(fminf(yi1, yi2))));
        const int yi_max = m_cvt_rz_sat(fminf(fmaxf(yi1, yi2), (float)src - 1.f));
        int cy_lo = 0, cnt = 0;
        if (colv && yi_min <= yi_max) {
          cy_lo = first_lat(yi_min);
          cnt = max(0, last_lat(yi_max) - cy_lo + 1);
        }
        const uint32_t packed = (uint32_t)face | ((uint32_t)cx << kFaceBits) | ((uint32_t)cy_lo << (kFaceBits + 7));
        for (int t = 0;; t++) {
          const bool on = cnt > t;
          const unsigned long long m = __ballot(on);
          if (m == 0ull) break;
          if (on)
            s_queue[qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                packed + ((uint32_t)t << (kFaceBits + 7));
          qn += __popcll(m);
          if (qn >= 64) drain(64);
        }
      }
      if (qn > 0) drain(qn);
      __builtin_amdgcn_s_waitcnt(0xc07f);   // the queue's and the rows' last reads: the next batch rewrites them
      __builtin_amdgcn_wave_barrier();
    }
  }
  MESH_STAMP(6);
  __syncthreads();

  // ---- 3. clamp + bilinear (mesh/render.py:286, :311; ATen upsample_bilinear2d) -----------------------------------
  float *out = depth + (size_t)b * S * S;
  if (POST) {
    // the module's pixel (oy, ox) before the noise: clamp, ATen's bilinear formula, * depth_scale (one fp32 multiply)
    const float scale = (float)src / (float)S;
    auto value = [&](int oy, int ox) -> float {
      if (SL == 1) return fminf(mkey_inv(s_z[oy * LP + ox]), clamp_max) * post.depth_scale;
      const Lin lx = lin_index(ox, scale, src), ly = lin_index(oy, scale, src);
      float v[2][2];
#pragma unroll
      for (int sy = 0; sy < 2; sy++)
#pragma unroll
        for (int sx = 0; sx < 2; sx++) v[sy][sx] = fminf(mkey_inv(s_z[(2 * oy + sy) * LP + 2 * ox + sx]), clamp_max);
      return (ly.l0 * (lx.l0 * v[0][0] + lx.l1 * v[0][1]) + ly.l1 * (lx.l0 * v[1][0] + lx.l1 * v[1][1])) * post.depth_scale;
    };
    // (one launch: the keys are this workgroup's own draws; post.keys then only says "noise on")
    const bool noise = post.keys != nullptr;
    const uint32_t key0 = !noise ? 0u : (own_pose ? __float_as_uint(s_draw[4]) : post.keys[b]);
    const uint32_t key1 = !noise ? 0u : (own_pose ? __float_as_uint(s_draw[5]) : post.keys[post.B + b]);
    auto pixel = [&](int i, int oy, int ox) -> float {
      if (noise) {
        const NoiseShift sh = noise_shift(key0, (uint32_t)i, post.t0, post.t1, post.t2);
        oy = min(max(oy + sh.dy, 0), S - 1);
        ox = min(max(ox + sh.dx, 0), S - 1);
      }
      float z = value(oy, ox);
      if (noise && z < 1.0f) z = z + noise_normal(key1, (uint32_t)i) * post.sigma_z;
      return z;
    };
    if ((S & 3) == 0) {
      typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
      for (int i = tid; i < S * S / 4; i += kLatWaves * 64) {
        const int oy = i / (S / 4), ox = (i - oy * (S / 4)) * 4;
        v4u_t v;
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = __float_as_uint(pixel(4 * i + c, oy, ox + c));
        float *dst = out + (size_t)oy * S + ox;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      }
    } else {
      for (int i = tid; i < S * S; i += kLatWaves * 64) {
        const int oy = i / S;
        out[i] = pixel(i, oy, i - oy * S);
      }
    }
    if (paint) {
      // HeatmapRender.forward (mesh/render.py:226-248) for the crop's J maps, element e = j * hm^2 + pixel: coalesced stores
      const int sh = 2 * post.hm_shift, npx = 1 << sh, hm = 1 << post.hm_shift;
      float *uo = post.uv_hm + (size_t)b * post.J * npx, *dout = post.d_hm + (size_t)b * post.J * npx;
      for (int e = tid; e < post.J * npx; e += kLatWaves * 64) {
        const float4 p = s_kp[e >> sh];
        const int i = e & (npx - 1), v = i >> post.hm_shift, u = i & (hm - 1);
        const float du = (float)u - p.x, dv = (float)v - p.y;
        const float g = __expf(-0.5f * post.hsigma * (du * du + dv * dv));
        uo[e] = g * post.uv_scale;
        dout[e] = (g > 0.05f ? p.z : 0.f) * post.d_scale;
      }
      if (tid < post.J) {
        const float4 p = s_kp[tid];
        post.xyz[(size_t)b * post.J + tid] = make_float4(post.a00 * p.x + post.a03 * p.w, post.a11 * p.y + post.a13 * p.w, p.z, p.w);
      }
    }
    // the call counter advances when the launch's LAST workgroup is through: every workgroup has read it by then
    if (post.state && tid == 0) {
      const unsigned long long done = atomicAdd(&post.state[2], 1ull);
      if (done == (unsigned long long)(gridDim.x - 1)) {
        post.state[2] = 0ull;
        post.state[1] += 1ull;
      }
    }
  } else
  if (SL == 1 && (S & 3) == 0) {
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    for (int i = tid; i < S * S / 4; i += kLatWaves * 64) {
      const int oy = i / (S / 4), ox = (i - oy * (S / 4)) * 4;
      v4u_t v;
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = __float_as_uint(fminf(mkey_inv(s_z[oy * LP + ox + c]), clamp_max));
      float *dst = out + (size_t)oy * S + ox;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
  } else {
    const float scale = (float)src / (float)S;
    for (int i = tid; i < S * S; i += kLatWaves * 64) {
      const int oy = i / S, ox = i - oy * S;
      if (SL == 1) {   // weights are exactly (1, 0)
        out[i] = fminf(mkey_inv(s_z[oy * LP + ox]), clamp_max);
      } else {
        const Lin lx = lin_index(ox, scale, src), ly = lin_index(oy, scale, src);
        float v[2][2];
#pragma unroll
        for (int sy = 0; sy < 2; sy++)
#pragma unroll
          for (int sx = 0; sx < 2; sx++) v[sy][sx] = fminf(mkey_inv(s_z[(2 * oy + sy) * LP + 2 * ox + sx]), clamp_max);
        out[i] = ly.l0 * (lx.l0 * v[0][0] + lx.l1 * v[0][1]) + ly.l1 * (lx.l0 * v[1][0] + lx.l1 * v[1][1]);
      }
    }
  }
  MESH_STAMP(7);
}

}  // namespace shr

#ifdef MESH_TL
extern "C" int shr_mesh_debug_timeline(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(shr::mesh_tl), sizeof(shr::mesh_tl));
}
#endif

static int g_mesh_band = 1;   // SHR_TUNE_MESH_BAND
int shr::mesh_set_band(int on) { g_mesh_band = on; return SHR_OK; }

// the lattice kernel's launch (vertices: skinned ones in HBM, or nullptr with `skin` for the fused kernel); returns -1
// when the problem is not the lattice kernel's (non-integer ratio, lattice above 128 x 128, LDS)
static int mesh_lattice_launch(const float4 *v4, const shr::LatticeSkin *skin, const int32_t *faces, int B, int NV, int F,
                               int src_size, int S, float clamp_max, float *depth, hipStream_t s,
                               const shr::LatticePost *post = nullptr) {
  using namespace shr;
  static const int lattice_mode = [] { const char *e = getenv("SHR_MESH_LATTICE"); return e ? atoi(e) : 1; }();   // 0: off, 1: 16 x 32, 2: 12 x 64
  const bool single = (src_size % S == 0) && (((src_size / S) & 1) == 1);
  const bool even = (src_size % S == 0) && (((src_size / S) & 1) == 0);
  const int SLx = single ? 1 : 2;
  const int lw = lattice_mode == 2 ? 12 : 16, lf = lattice_mode == 2 ? 64 : 32;
  const size_t lat_lds = lattice_lds_bytes(SLx * S, F, lw, lf, skin ? NV : 0);
  if (lattice_mode == 0 || !(single || even) || SLx * S > kLatMax || F <= 0 || F > 65535 || lat_lds > 160 * 1024 - 2048) return -1;   // (+ the static arrays)
  if (skin && (skin->NB * 64 > lw * lat_scratch_bytes(lf))) return -1;     // (the matrices are staged in the scratch)
  static AttrDone attr_done[10];   // per (kernel, device)
  LatticeSkin sk = {};
  if (skin) sk = *skin;
  LatticePost po = {};
  if (post) po = *post;
  auto launch = [&](auto kernel, int which) -> int {
    const hipError_t e = allow_dynamic_lds(kernel, 160 * 1024 - 2048, &attr_done[which]);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)B), dim3(lw * 64), lat_lds, s, v4, faces, NV, F, src_size, S, clamp_max, depth, sk, po);
    return (int)hipGetLastError();
  };
  if (skin && post) {   // (the synthesizer's launch: the default wave x face shape only)
    if (lattice_mode == 2) return -1;
    return single ? launch(mesh_lattice_kernel<1, 16, 32, true, true>, 8) : launch(mesh_lattice_kernel<2, 16, 32, true, true>, 9);
  }
  if (skin) {
    if (lattice_mode == 2) return single ? launch(mesh_lattice_kernel<1, 12, 64, true>, 6) : launch(mesh_lattice_kernel<2, 12, 64, true>, 7);
    return single ? launch(mesh_lattice_kernel<1, 16, 32, true>, 4) : launch(mesh_lattice_kernel<2, 16, 32, true>, 5);
  }
  if (lattice_mode == 2) return single ? launch(mesh_lattice_kernel<1, 12, 64>, 2) : launch(mesh_lattice_kernel<2, 12, 64>, 3);
  return single ? launch(mesh_lattice_kernel<1, 16, 32>, 0) : launch(mesh_lattice_kernel<2, 16, 32>, 1);
}

extern "C" int shr_mesh_depth_fwd(const float *vertices, const int32_t *faces, int B, int NV, int F, int src_size,
                                  int S, float clamp_max, float *depth, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!vertices || (!faces && F > 0) || !depth || B < 0 || NV <= 0 || F < 0 || src_size <= 0 || S <= 0)
    return SHR_EINVAL;
  if (((uintptr_t)vertices & 15u) != 0) return SHR_EINVAL;
  if (B > 65535 || S > 16384 || src_size > 32767 || S > src_size || F > (1 << 24))
    return SHR_ETOOLARGE;  // no up-sampling; work items pack the face index in 24 bits, the face rows pixel ranges in 16
  hipStream_t s = (hipStream_t)stream;
  const float4 *v4 = reinterpret_cast<const float4 *>(vertices);
  // odd integer ratio: src = ratio * d + (ratio - 1) / 2 exactly, bilinear weights (1, 0)
  const bool single = (src_size % S == 0) && (((src_size / S) & 1) == 1);
  const bool even = (src_size % S == 0) && (((src_size / S) & 1) == 0);   // even integer ratio: two slots, closed forms
#define MESH_LAUNCH(TO, SL, EX)                                                                                  \
  do {                                                                                                           \
    const int t = (S + (TO) - 1) / (TO);                                                                         \
    hipLaunchKernelGGL((mesh_depth_kernel<TO, SL, EX>), dim3((unsigned)(t * t), (unsigned)B), dim3(1024), 0, s,  \
                       v4, faces, NV, F, src_size, S, clamp_max, depth);                                         \
  } while (0)
  {   // integer ratio and a lattice of sampled source pixels that fits one workgroup's LDS: the lattice kernel
      // (SHR_MESH_LATTICE=0 in the environment keeps the tile kernel: tests and tools compare the two)
    const int r = mesh_lattice_launch(v4, nullptr, faces, B, NV, F, src_size, S, clamp_max, depth, s);
    if (r >= 0) return r;
  }
  {   // no lattice, but the resize samples at least half of the source pixels (S = 256 from 640: 64 %): the triangle BAND
      // kernel at full resolution with clamp + resize as its stream-out (tri_raster.hip RESIZE) -- 64 crops @256 x 256:
      // 304 us with the tile kernel below.  SHR_TUNE_MESH_BAND 0 keeps the tile kernel (tests compare the two).
    if (g_mesh_band != 0 && 8LL * S * S >= (long long)src_size * src_size) {
      const int r = tri_band_resize(vertices, faces, B, NV, F, src_size, S, clamp_max, depth, s);
      if (r >= 0) return r;
    }
  }
  if (single) {
    if (S > 64) MESH_LAUNCH(128, 1, true); else MESH_LAUNCH(64, 1, true);
  } else if (even) {
    if (S > 32) MESH_LAUNCH(64, 2, true); else MESH_LAUNCH(32, 2, true);
  } else {
    if (S > 32) MESH_LAUNCH(64, 2, false); else MESH_LAUNCH(32, 2, false);
  }
#undef MESH_LAUNCH
  return (int)hipGetLastError();
}

/* DepthRender.forward (mesh/render.py:315-331) in ONE launch where the lattice kernel applies: skinning + camera
 * (shr_lbs_project's arguments, always projecting) + raster + clamp + resize (shr_mesh_depth_fwd's).  Otherwise the two
 * launches, through `vertices_ws` [B][NV][4] (16-byte aligned; may be NULL only if the caller knows the fused kernel
 * applies: SHR_EINVAL then).  Same images either way. */
extern "C" int shr_lbs_project(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start, const int32_t *skin_bone,
                               const float *skin_wv, int right_hand, int project, float cx, float cy, float fx, float fy,
                               const float *rand_f, float *out, void *stream);
extern "C" int shr_mesh_render_fwd(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                                   const int32_t *skin_bone, const float *skin_wv, int right_hand, float cx, float cy,
                                   float fx, float fy, const float *rand_f, const int32_t *faces, int F, int src_size,
                                   int S, float clamp_max, float *vertices_ws, float *depth, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!T || !skin_vertex_start || !skin_bone || !skin_wv || (!faces && F > 0) || !depth || B < 0 || NB <= 0 || NV <= 0 ||
      F < 0 || src_size <= 0 || S <= 0)
    return SHR_EINVAL;
  if (((uintptr_t)skin_wv & 15u) != 0) return SHR_EINVAL;
  if (B > 65535 || S > 16384 || src_size > 32767 || S > src_size || F > (1 << 24) || NB > 160) return SHR_ETOOLARGE;
  LatticeSkin sk;
  sk.T = T; sk.vstart = skin_vertex_start; sk.sbone = skin_bone; sk.swv = reinterpret_cast<const float4 *>(skin_wv);
  sk.rand_f = rand_f; sk.NB = NB; sk.right_hand = right_hand; sk.cx = cx; sk.cy = cy; sk.fx = fx; sk.fy = fy;
  const int r = mesh_lattice_launch(nullptr, &sk, faces, B, NV, F, src_size, S, clamp_max, depth, (hipStream_t)stream);
  if (r >= 0) return r;
  if (!vertices_ws) return SHR_EINVAL;
  const int e = shr_lbs_project(T, B, NB, NV, skin_vertex_start, skin_bone, skin_wv, right_hand, 1, cx, cy, fx, fy, rand_f,
                                vertices_ws, stream);
  if (e != SHR_OK) return e;
  return shr_mesh_depth_fwd(vertices_ws, faces, B, NV, F, src_size, S, clamp_max, depth, stream);
}

/* 1 where shr_mesh_render_fwd / shr_mesh_render_post_fwd take the problem as ONE launch (no workspaces needed), else 0 */
extern "C" int shr_mesh_render_one_launch(int NB, int NV, int F, int src_size, int S) {
  using namespace shr;
  static const int lattice_mode = [] { const char *e = getenv("SHR_MESH_LATTICE"); return e ? atoi(e) : 1; }();
  if (lattice_mode != 1 || S <= 0 || src_size <= 0 || src_size % S != 0) return 0;
  const int SLx = ((src_size / S) & 1) ? 1 : 2;
  if (SLx * S > kLatMax || F <= 0 || F > 65535) return 0;
  if (lattice_lds_bytes(SLx * S, F, 16, 32, NV) > 160 * 1024 - 2048 || NB * 64 > 16 * lat_scratch_bytes(32)) return 0;
  return 1;
}

/* HandSynthesizer.forward's depth branch (network/util_modules.py:111-116): DepthRender, `* depth_scale` and DepthNoise.
 * ONE launch where the lattice kernel applies (the tail runs in its epilogue); otherwise shr_mesh_render_fwd into
 * depth_ws[B,S,S] and the tail as a launch of its own -- the same draws, the same bits.  noise_keys[2][B]: the samples'
 * stream keys (shr_synth_pose_fwd's draws 4 and 5, as uint32), NULL = no noise.  rng_state: the generator's state whose
 * call counter this launch advances (NULL: left alone). */
extern "C" int shr_mesh_render_post_fwd(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                                        const int32_t *skin_bone, const float *skin_wv, int right_hand, float cx, float cy,
                                        float fx, float fy, const float *rand_f, const int32_t *faces, int F, int src_size,
                                        int S, float clamp_max, float depth_scale, const uint32_t *noise_keys,
                                        float sigma_xy, float sigma_z, unsigned long long *rng_state, float *vertices_ws,
                                        float *depth_ws, float *depth, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!T || !skin_vertex_start || !skin_bone || !skin_wv || (!faces && F > 0) || !depth || B < 0 || NB <= 0 || NV <= 0 ||
      F < 0 || src_size <= 0 || S <= 0)
    return SHR_EINVAL;
  if (((uintptr_t)skin_wv & 15u) != 0 || ((uintptr_t)depth & 15u) != 0 || ((uintptr_t)rng_state & 7u) != 0) return SHR_EINVAL;
  if (noise_keys && !(sigma_xy > 0.f && sigma_xy <= 0.6f)) return SHR_EINVAL;   // (shifts -1 .. +2 hold all but 1e-5 of the mass)
  if (B > 65535 || S > 16384 || src_size > 32767 || S > src_size || F > (1 << 24) || NB > 160) return SHR_ETOOLARGE;
  LatticePost po = {};
  po.depth_scale = depth_scale; po.keys = noise_keys; po.sigma_z = sigma_z; po.state = rng_state; po.B = B;
  {   // P(trunc(n sigma + 0.5) < k) = Phi((k - 0.5) / sigma) for k <= 0 (truncation towards zero: shift 0 holds (-1, 1))
    auto Phi = [](double x) { return 0.5 * erfc(-x / 1.4142135623730951); };
    const double s = noise_keys ? (double)sigma_xy : 0.5;
    po.t0 = (uint32_t)llround(65536.0 * Phi((-1.0 - 0.5) / s));     // shift <= -1  <=>  n sigma + 0.5 <= -1
    po.t1 = (uint32_t)llround(65536.0 * Phi((1.0 - 0.5) / s));      // shift <= 0   <=>  n sigma + 0.5 < 1
    po.t2 = (uint32_t)llround(65536.0 * Phi((2.0 - 0.5) / s));      // shift <= 1   <=>  n sigma + 0.5 < 2
  }
  LatticeSkin sk;
  sk.T = T; sk.vstart = skin_vertex_start; sk.sbone = skin_bone; sk.swv = reinterpret_cast<const float4 *>(skin_wv);
  sk.rand_f = rand_f; sk.NB = NB; sk.right_hand = right_hand; sk.cx = cx; sk.cy = cy; sk.fx = fx; sk.fy = fy;
  const int r = mesh_lattice_launch(nullptr, &sk, faces, B, NV, F, src_size, S, clamp_max, depth, (hipStream_t)stream, &po);
  if (r >= 0) return r;
  if (!depth_ws || depth_ws == depth) return SHR_EINVAL;
  const int e = shr_mesh_render_fwd(T, B, NB, NV, skin_vertex_start, skin_bone, skin_wv, right_hand, cx, cy, fx, fy, rand_f,
                                    faces, F, src_size, S, clamp_max, vertices_ws, depth_ws, stream);
  if (e != SHR_OK) return e;
  const int per = (S * S + 255) / 256;
  hipLaunchKernelGGL(depth_post_kernel, dim3((unsigned)(per > 64 ? 64 : per), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     depth_ws, B, S, S, po, depth);
  return (int)hipGetLastError();
}

/* HandSynthesizer.forward (network/util_modules.py:104-122) in ONE launch: every workgroup runs its crop's forward
 * kinematics + RandScale + draws (shr_synth_pose_fwd's arithmetic), skins, rasterizes, scales and noises it
 * (shr_mesh_render_post_fwd's) and paints its heat-maps (shr_heatmap_render_fwd's) -- the same bits as the three launches.
 * Only where shr_hand_synth_one_launch() says so (the lattice kernel's sizes, J <= 64, a power-of-two heat-map side);
 * SHR_EINVAL otherwise.  noise: 0 / 1; uv_hm = NULL: no heat-maps (then d_hm, xyz and the kp_* tables are not read).
 * rng_state: uint64 [3] = (seed, call counter, ticket -- zero between launches); the launch's last workgroup advances
 * the counter. */
extern "C" int shr_hand_synth_one_launch(int NB, int NV, int F, int src_size, int S, int J, int hm) {
  if (!shr_mesh_render_one_launch(NB, NV, F, src_size, S) || NB != shr::kBones) return 0;
  if (J < 0 || J > 64 || hm <= 0 || (hm & (hm - 1)) != 0 || hm > 1024) return 0;
  return 1;
}

extern "C" int shr_hand_synth_fwd(const float *params, int B, const float *offset, const float *offset_inv,
                                  unsigned long long *rng_state, float rand_scale, int NV,
                                  const int32_t *skin_vertex_start, const int32_t *skin_bone, const float *skin_wv,
                                  int right_hand, float cx, float cy, float fx, float fy, const int32_t *faces, int F,
                                  int src_size, int S, float clamp_max, float depth_scale, int noise, float sigma_xy,
                                  float sigma_z, int J, const int32_t *kp_start, const int32_t *kp_bone, const float *kp_wv,
                                  int hm, float hcx, float hcy, float hfx, float hfy, float hm_sigma, float uv_scale,
                                  float d_scale, float a00, float a03, float a11, float a13, float *draws, float *depth,
                                  float *uv_hm, float *d_hm, float *xyz, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!params || !offset || !offset_inv || !rng_state || !skin_vertex_start || !skin_bone || !skin_wv || !faces || !draws ||
      !depth || B < 0 || NV <= 0 || F <= 0 || src_size <= 0 || S <= 0)
    return SHR_EINVAL;
  if (uv_hm && (!d_hm || !xyz || !kp_start || !kp_bone || !kp_wv)) return SHR_EINVAL;
  if ((((uintptr_t)skin_wv | (uintptr_t)depth | (uintptr_t)offset | (uintptr_t)offset_inv | (uintptr_t)kp_wv | (uintptr_t)xyz) & 15u) != 0 ||
      ((uintptr_t)rng_state & 7u) != 0)
    return SHR_EINVAL;
  if (noise && !(sigma_xy > 0.f && sigma_xy <= 0.6f)) return SHR_EINVAL;
  if (B > 65535) return SHR_ETOOLARGE;
  if (!shr_hand_synth_one_launch(kBones, NV, F, src_size, S, uv_hm ? J : 0, uv_hm ? hm : 1)) return SHR_EINVAL;
  LatticePost po = {};
  po.depth_scale = depth_scale; po.keys = noise ? reinterpret_cast<const uint32_t *>(draws) : nullptr; po.sigma_z = sigma_z;
  po.state = rng_state; po.B = B;
  {
    auto Phi = [](double x) { return 0.5 * erfc(-x / 1.4142135623730951); };
    const double sg = noise ? (double)sigma_xy : 0.5;
    po.t0 = (uint32_t)llround(65536.0 * Phi(-1.5 / sg));
    po.t1 = (uint32_t)llround(65536.0 * Phi(0.5 / sg));
    po.t2 = (uint32_t)llround(65536.0 * Phi(1.5 / sg));
  }
  po.params = params; po.offset = offset; po.offset_inv = offset_inv;
  po.rand_scale = rand_scale; po.rand_half = (float)((double)rand_scale / 2.0); po.draws = draws;
  if (uv_hm) {
    po.kp_start = kp_start; po.kp_bone = kp_bone; po.kp_wv = reinterpret_cast<const float4 *>(kp_wv); po.J = J;
    int sh = 0;
    while ((1 << sh) < hm) sh++;
    po.hm_shift = sh;
    po.hcx = hcx; po.hcy = hcy; po.hfx = hfx; po.hfy = hfy; po.hsigma = hm_sigma; po.uv_scale = uv_scale; po.d_scale = d_scale;
    po.a00 = a00; po.a03 = a03; po.a11 = a11; po.a13 = a13;
    po.uv_hm = uv_hm; po.d_hm = d_hm; po.xyz = reinterpret_cast<float4 *>(xyz);
  }
  LatticeSkin sk;
  sk.T = nullptr; sk.vstart = skin_vertex_start; sk.sbone = skin_bone; sk.swv = reinterpret_cast<const float4 *>(skin_wv);
  sk.rand_f = nullptr; sk.NB = kBones; sk.right_hand = right_hand; sk.cx = cx; sk.cy = cy; sk.fx = fx; sk.fy = fy;
  const int r = mesh_lattice_launch(nullptr, &sk, faces, B, NV, F, src_size, S, clamp_max, depth, (hipStream_t)stream, &po);
  return r >= 0 ? r : SHR_EINVAL;
}
