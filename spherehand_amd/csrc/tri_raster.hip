// tri_raster.hip -- triangle-mesh depth rasterizer (forward only) + skinning.
//
// Replaces (reference file:line):
//   mesh/cuda_kernel/depth_rasterization_cuda_kernel.cu:18-113 `kernel` and :6-16
//   `atomicMin`, :115-134 depth_rasterization_cuda_forward  -> shr_tri_raster_fwd
//   mesh/pointTransformation.py:39-46 LinearBlendSkinning.forward and :84-99
//   OthographicalProjection.forward                           -> shr_lbs_project
//
// The reference launches one single-thread block per face that walks the face's
// pixel columns serially and CAS-loops a float min per pixel.  Here a wave takes
// 32 faces: lanes = faces for the set-up (cull, sort by x, inverse barycentric
// matrix), then lanes = the faces' pixel COLUMNS for the spans and lanes = the
// pixels inside the spans for the depth (raster_batch below).
// Every pixel repeats the reference's per-column span test and per-pixel
// arithmetic verbatim (fp32, one rounding per written operator, IEEE division; the
// `1. / x` the reference evaluates in fp64 and rounds to fp32 equals the fp32
// quotient exactly -- 53 >= 2*24+2 bits).  The float min is a native integer
// atomic on the fp32 bits (signed min / unsigned max by sign: order independent,
// hence deterministic); the image holds plain floats throughout: one fill pass,
// one raster pass.
#include "common.h"

namespace shr {

typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));

// CUDA double -> int32 conversion (cvt.rzi.s32.f64): truncate, saturate, NaN -> 0.
// The operands here are fp32 values promoted to double, so fp32 compares suffice.
__device__ __forceinline__ int cvt_rz_sat(float d) {
  if (d != d) return 0;
  if (d >= 2147483648.0f) return 2147483647;
  if (d <= -2147483648.0f) return (int)0x80000000;
  return (int)d;
}

struct FaceSetup {
  float p[3][3];   // vertices sorted by x
  float fi[9];     // inverse barycentric matrix / denominator
  // the three edge slopes (.cu:75-85): the reference divides per COLUMN, but the quotients depend on the face only --
  // one IEEE division each here, in the set-up (lanes = faces), instead of two per box pixel in the span test
  float s01, s12, s02;
  int sflags;      // bit 0: x1 - x0 != 0, bit 1: x2 - x1 != 0 (else the span end is y1, .cu:77, :83)
  int xi_min, xi_max, r_lo, r_hi;
  int live;
};

// .cu:25-69 for one face
__device__ __forceinline__ FaceSetup face_setup(const float f_[9], int width, int height) {
  // (opaque copies: the compiler turns the selects of the sort below -- "vertex order[a] of three" -- into ONE load from a
  // select of addresses, which pins the nine values to a scratch array: 3 scratch stores and 9 dependent scratch loads per
  // set-up, ScratchSize 48, in every kernel that sets faces up; values that are no longer loads stay in registers)
  float f[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { f[k] = f_[k]; asm("" : "+v"(f[k])); }
  FaceSetup s;
  s.live = 1;
  if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) s.live = 0;  // :33 back face
  int p0, p2;
  if (f[0] < f[3]) {
    p0 = (f[6] < f[0]) ? 2 : 0;
    p2 = (f[3] < f[6]) ? 2 : 1;
  } else {
    p0 = (f[6] < f[3]) ? 2 : 1;
    p2 = (f[0] < f[6]) ? 2 : 0;
  }
  int p1 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (p0 != k && p2 != k) p1 = k;
  const int order[3] = {p0, p1, p2};
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      // select without dynamic indexing (keeps everything in registers)
      const int o = order[a];
      s.p[a][d] = (o == 0) ? f[d] : ((o == 1) ? f[3 + d] : f[6 + d]);
    }
  if (s.p[0][0] == s.p[2][0]) s.live = 0;  // :54
  float (*p)[3] = s.p;
  s.fi[0] = p[1][1] - p[2][1]; s.fi[1] = p[2][0] - p[1][0]; s.fi[2] = p[1][0] * p[2][1] - p[2][0] * p[1][1];
  s.fi[3] = p[2][1] - p[0][1]; s.fi[4] = p[0][0] - p[2][0]; s.fi[5] = p[2][0] * p[0][1] - p[0][0] * p[2][1];
  s.fi[6] = p[0][1] - p[1][1]; s.fi[7] = p[1][0] - p[0][0]; s.fi[8] = p[0][0] * p[1][1] - p[1][0] * p[0][1];
  const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1])) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
  for (int k = 0; k < 9; k++) s.fi[k] = s.fi[k] / den;
  {
    const bool d01 = p[1][0] - p[0][0] != 0.f, d12 = p[2][0] - p[1][0] != 0.f;
    s.s01 = d01 ? (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) : 0.f;
    s.s12 = d12 ? (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]) : 0.f;
    s.s02 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]);
    s.sflags = (d01 ? 1 : 0) | (d12 ? 2 : 0);
  }
  // :68-69  max(ceil(x0), 0.) / min(x2, width - 1.)  (fmax/fmin drop a NaN operand)
  s.xi_min = cvt_rz_sat(fmaxf(ceilf(p[0][0]), 0.f));
  s.xi_max = cvt_rz_sat(fminf(p[2][0], (float)width - 1.f));
  // conservative row range of the columns' spans
  const float ylo = fminf(fminf(p[0][1], p[1][1]), p[2][1]);
  const float yhi = fmaxf(fmaxf(p[0][1], p[1][1]), p[2][1]);
  // (a face whose largest x lies in (-1, 0) still reaches column 0 -- the reference truncates x2 towards zero,
  // .cu:69 -- and the span there is an EXTRApolation of the edges: any row)
  const bool wild = !(fabsf(ylo) < 1e9f) || !(fabsf(yhi) < 1e9f) || p[2][0] < 0.f;
  // A column's span ends are edge interpolations slope * (x - xa) + ya at an x inside the edge:
  // convex combinations of the vertices' y up to 4 roundings (<= 2.4e-7 * |y|); rows
  // [ceil(min), trunc(max)] (.cu:89-90; a span end in (-1, 0) truncates to row 0).
  const float yeps = 1e-5f * (fabsf(ylo) + fabsf(yhi)) + 1e-4f;
  s.r_lo = wild ? 0 : max(0, (int)ceilf(ylo - yeps));
  s.r_hi = wild ? height - 1 : min(height - 1, max(0, (int)floorf(yhi + yeps)));
  if (s.xi_min > s.xi_max) s.live = 0;
  return s;
}

// Float min on the fp32 bits themselves, with native integer atomics: among non-negative
// floats the bits order like signed integers, among negative ones like unsigned integers
// reversed -- a signed MIN for a value whose sign bit is clear (an older negative entry is a
// smaller signed integer and stays), an unsigned MAX for one whose sign bit is set (it beats
// every non-negative entry, and the more negative of two negatives is the larger unsigned).
// The image holds plain floats at all times: no decode pass.  NaN is never offered.
__device__ __forceinline__ void zmin(float *cell, float v) {
  const uint32_t b = __float_as_uint(v);
  if (b >> 31) atomicMax(reinterpret_cast<unsigned int *>(cell), b);
  else atomicMin(reinterpret_cast<int *>(cell), (int)b);
}

// .cu:72-90: the rows [yi_min, yi_max] of column xi's span of the face.  (x0, y0), (x1, y1): the first two vertices
// sorted by x; the slopes and their flags from face_setup.
__device__ __forceinline__ void span_rows(float x0, float y0, float x1, float y1, float s01, float s12, float s02,
                                          int sflags, int xi, int height, int &yi_min, int &yi_max) {
  const float xf = (float)xi;
  float yi1;
  if (xf <= x1) yi1 = (sflags & 1) ? s01 * (xf - x0) + y0 : y1;
  else yi1 = (sflags & 2) ? s12 * (xf - x1) + y1 : y1;
  const float yi2 = s02 * (xf - x0) + y0;
  yi_min = cvt_rz_sat(fmaxf(0.f, ceilf(fminf(yi1, yi2))));
  yi_max = cvt_rz_sat(fminf(fmaxf(yi1, yi2), (float)height - 1.f));
}
// ... and whether row yi is inside it
__device__ __forceinline__ bool span_inside(float x0, float y0, float x1, float y1, float s01, float s12, float s02,
                                            int sflags, int xi, int yi, int height) {
  int yi_min, yi_max;
  span_rows(x0, y0, x1, y1, s01, s12, s02, sflags, xi, height, yi_min, yi_max);
  return yi >= yi_min && yi <= yi_max;
}

// .cu:97-110 for a pixel inside its column's span: the depth the reference offers to its atomicMin (NaN: none)
__device__ __forceinline__ float span_depth(const float (&pz)[3], const float (&rz)[3], bool tame, const float (&fi)[9], int xi,
                                            int yi) {
  const float xf = (float)xi, yf = (float)yi;
  float w[3];
  float w_sum = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    w[k] = (fi[3 * k + 0] * xf + fi[3 * k + 1] * yf) + fi[3 * k + 2];
    w[k] = fminf(fmaxf(w[k], 0.f), 1.f);
    w_sum += w[k];
  }
  return tri_pixel_depth(w[0], w[1], w[2], w_sum, pz, rz, tame);   // (the seven divisions of .cu:104-110)
}

// A wave sets up to 32 faces up with lanes = faces and parks each face's values in an LDS row (raster_batch below).
constexpr int kFaceRow = 28;          // x0 y0 x1 y1 x2 y2 s01 s12 | z0 z1 z2 fi[9] | x of column item 0, r_lo, r_hi, s02 | slope flags
constexpr int kFacesPerWave = 32;     // 4.5 KB of LDS per wave (rows, queue): eight waves per SIMD fit
// a wave's private scratch: face rows | pixel queue
constexpr int kScratchQueueBytes = 128 * 8;                                        // 1024
constexpr int kScratchMarkBytes = 64;                                               // start marks of a chunk's faces
constexpr int wave_scratch_bytes(int nf) { return nf * kFaceRow * 4 + kScratchQueueBytes + kScratchMarkBytes; }
constexpr int kWaveScratchBytes = wave_scratch_bytes(kFacesPerWave);               // 3584 + 1024

// One batch: lane l < 32 brings face set-up `s` (`have`: the lane holds a face; rows [s.r_lo, s.r_hi] already clipped
// to what the caller wants rasterized); every pixel inside its column's span goes to sink(xi, yi, depth) once.
// `scratch`: this wave's kWaveScratchBytes of LDS (16-byte aligned); nothing of it is live between two batches.
// One batch, the reference's own loop structure spread over a wave (.cu:70-111: for each column of the face its span
// of rows, for each row of the span a pixel):
//   columns  lanes = the COLUMNS of the batch's faces, 64 at a time whatever face they belong to: the column's span
//            (span_rows: two edge interpolations, the slopes divided once per face in the set-up), clipped to the
//            face's row range, and a prefix sum over the spans' lengths;
//   pixels   lanes = the pixels INSIDE those spans, 64 at a time, every lane with a pixel: queued as (face, x, y);
//   pass B   lanes = queued pixels, 64 at a time, every lane busy: the 7 divisions of the barycentric weights and the
//            perspective depth, then sink(x, y, depth).
// Rounds 1-4 walked every face's BOX (16-pixel groups; 8x8 patches of one face at a time for boxes above 256 pixels)
// with a span test per box pixel: three box pixels in four lie outside the spans, and the large boxes -- one face in
// six of the hand mesh, six tenths of its box pixels -- took 380 of the kernel's 630 us for 256 crops.
// NF = faces per batch (lanes 0 .. NF - 1 bring one each; `scratch` = wave_scratch_bytes(NF)).
// LEVELS: the pixels of a chunk of columns are queued LEVEL by level -- level t = row ylo + t of every column whose span
// is longer than t, compacted with a ballot -- instead of column after column: a level costs a ballot and a store where
// the pixel-order walk pays a run search and three shuffles per 64 pixels, and a drain's neighbouring lanes are then
// neighbouring columns at one height.  For an LDS sink (band kernel): 1 crop 54 -> 37 us.  NOT for the global atomics:
// 64 lanes in a few cache lines resolve slower in the L2 than 64 lanes in 64 lines (256 crops 461 -> 544 us;
// tools/exp_tri_level.py, EXPERIMENTS R3c).
// (tools/exp_tri_level.py builds this file with the two kernels' walks swapped: the measurements behind the defaults)
#ifndef SHR_TRI_ATOMIC_LEVELS
#define SHR_TRI_ATOMIC_LEVELS 0
#endif
#ifndef SHR_TRI_BAND_LEVELS
#define SHR_TRI_BAND_LEVELS 1
#endif
template <int NF, bool LEVELS, typename Sink>
__device__ __forceinline__ void raster_batch(const FaceSetup &s, bool have, int lane, unsigned char *scratch, int height,
                                             Sink &&sink) {
  float (*s_face)[kFaceRow] = reinterpret_cast<float (*)[kFaceRow]>(scratch);
  uint2 *s_queue = reinterpret_cast<uint2 *>(scratch + NF * kFaceRow * 4);
  unsigned char *s_mark = scratch + NF * kFaceRow * 4 + kScratchQueueBytes;
  const int bw = s.xi_max - s.xi_min + 1, bh = s.r_hi - s.r_lo + 1;
  const bool alive = have && s.live && bh > 0 && bw > 0;
  const int ncol = alive ? bw : 0;                     // (<= 65535 columns each, 32 faces: 32 bits)
  const int fincl = wave_scan_incl(ncol, lane);        // lanes = faces: the face's columns end here
  const int ncols = __builtin_amdgcn_readlane(fincl, 63);
  if (lane < NF) {
    float *r = s_face[lane];
#pragma unroll
    for (int a = 0; a < 3; a++) { r[2 * a] = s.p[a][0]; r[2 * a + 1] = s.p[a][1]; r[8 + a] = s.p[a][2]; }
#pragma unroll
    for (int k = 0; k < 9; k++) r[11 + k] = s.fi[k];
    r[6] = s.s01; r[7] = s.s12;
    r[20] = __int_as_float(s.xi_min - (fincl - ncol));   // column item k of the face is pixel column k + this
    r[21] = __int_as_float(s.r_lo); r[22] = __int_as_float(s.r_hi); r[23] = s.s02;
    {   // the corners' z: their refined reciprocals for the pixels' divisions (common.h tri_pixel_depth), bit 2: all tame
      const bool tame = div_tame_z(s.p[0][2]) && div_tame_z(s.p[1][2]) && div_tame_z(s.p[2][2]);
#pragma unroll
      for (int a = 0; a < 3; a++) r[25 + a] = tame ? div_rcp_refined(s.p[a][2]) : 0.f;
      r[24] = __int_as_float(s.sflags | (tame ? 4 : 0));
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's LDS writes (rows and queue are private to the wave)
  __builtin_amdgcn_wave_barrier();

  int qn = 0;   // queued pixels (wave-uniform)
  auto drain = [&](int take) {   // pass B on the last `take` queued pixels
    if (lane < take) {
      const uint2 e = s_queue[qn - take + lane];
      const float4 *r4 = reinterpret_cast<const float4 *>(s_face[e.x]);
      const float4 b2 = r4[2], b3 = r4[3], b4 = r4[4], b6 = r4[6];
      const float pz[3] = {b2.x, b2.y, b2.z};
      const float fi[9] = {b2.w, b3.x, b3.y, b3.z, b3.w, b4.x, b4.y, b4.z, b4.w};
      const float rz[3] = {b6.y, b6.z, b6.w};                                  // (row words 25 .. 27; word 24: the flags)
      const int xi = (int)(e.y & 0xffffu), yi = (int)(e.y >> 16);
      const float zp = span_depth(pz, rz, (__float_as_int(b6.x) & 4) != 0, fi, xi, yi);
      if (zp == zp) sink(xi, yi, zp);  // fminf(NaN, old) = old
    }
    qn -= take;
  };
  for (int k0 = 0; k0 < ncols; k0 += 64) {
    const int k = k0 + lane;
    const bool colv = k < ncols;
    // the face of column item k: the faces whose first column lies in this chunk mark it (their number + 1: the numbers
    // rise with the position), a max-scan over the lanes spreads the marks, and the face that covers the chunk's first
    // column is one ballot (run_of's loop over the runs that end inside the chunk: ~5 of them here, ~25 on the lattice
    // of mesh_depth.hip, where this replaced 1 000 cycles per chunk)
    s_mark[lane] = 0;
    const int start = fincl - ncol;
    const unsigned long long before = __ballot(ncol > 0 && start < k0);
    if (ncol > 0 && start >= k0 && start < k0 + 64) s_mark[start - k0] = (unsigned char)(lane + 1);
    const int cover = before ? (int)(63 - __builtin_clzll(before)) : 0;       // the last face that starts before the chunk
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
    const float4 *r4 = reinterpret_cast<const float4 *>(s_face[face]);
    const float4 a0 = r4[0], a1 = r4[1], i0 = r4[5];
    const int fl = __float_as_int(s_face[face][24]);
    const int xi = k + __float_as_int(i0.x);
    int ylo, yhi;
    span_rows(a0.x, a0.y, a0.z, a0.w, a1.z, a1.w, i0.w, fl, xi, height, ylo, yhi);
    ylo = max(ylo, __float_as_int(i0.y));
    yhi = min(yhi, __float_as_int(i0.z));
    const int cnt = (colv && yhi >= ylo) ? yhi - ylo + 1 : 0;   // (a span holds at most 65535 rows, a chunk 64 columns)
    if (LEVELS) {
      const unsigned packed = (unsigned)xi | ((unsigned)ylo << 16);   // (xi, yi < 65536)
      for (int t = 0;; t++) {
        const bool on = cnt > t;
        const unsigned long long m = __ballot(on);
        if (m == 0ull) break;
        if (on) {
          const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          s_queue[pos] = make_uint2((unsigned)face, packed + ((unsigned)t << 16));
        }
        qn += __popcll(m);
        if (qn >= 64) drain(64);
      }
    } else {
      const int cincl = wave_scan_incl(cnt, lane);
      const int ctotal = __builtin_amdgcn_readlane(cincl, 63), cexcl = cincl - cnt;
      const int packed = face | (xi << 8);                          // (face < 32, xi < 65536)
      for (int p0 = 0; p0 < ctotal; p0 += 64) {
        const int p = p0 + lane;
        const int col = min(run_of(cincl, colv, p0, p), 63);       // (empty columns among the runs count too)
        const int cy = __shfl(ylo, col), ce = __shfl(cexcl, col), fx = __shfl(packed, col);
        const bool inside = p < ctotal;
        const unsigned long long m = __ballot(inside);
        if (inside) {
          const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          s_queue[pos] = make_uint2((unsigned)(fx & 0xff), (unsigned)(fx >> 8) | ((unsigned)(cy + (p - ce)) << 16));
        }
        qn += __popcll(m);
        if (qn >= 64) drain(64);
      }
    }
  }
  if (qn > 0) drain(qn);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // the queue's and the rows' last reads: the next batch rewrites them
  __builtin_amdgcn_wave_barrier();
}

template <bool INDEXED>
__device__ __forceinline__ void load_face(const float *__restrict__ src, const int *__restrict__ faces, int b, int F, int NV,
                                          int fidx, float f[9]) {
  if (INDEXED) {  // vertices [B,NV,4] + faces [F,3]: the gather of mesh/render.py:308-309 fused
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float4 v = reinterpret_cast<const float4 *>(src)[(size_t)b * NV + faces[fidx * 3 + k]];
      f[3 * k] = v.x; f[3 * k + 1] = v.y; f[3 * k + 2] = v.z;
    }
  } else {
    const float *fp = src + ((size_t)b * F + fidx) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = fp[k];
  }
}

// ---- the global-atomic kernel (any size) --------------------------------------------------------------------------
// A wave takes 32 faces and offers every covered pixel to the image with a native integer atomic on the fp32 bits
// (zmin), after one fill pass: 256 hand crops @640x640 (a 419-MB image) 630 us in rounds 1-4, 440-480 us since
// raster_batch walks column spans instead of boxes.
template <bool INDEXED>
__global__ void __launch_bounds__(256)
tri_raster_kernel(const float *__restrict__ src, const int *__restrict__ faces, int B, int F, int NV, int width,
                  int height, float *__restrict__ zbuf) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int groups = (F + kFacesPerWave - 1) / kFacesPerWave;
  const int b = wave_global / groups;
  if (b >= B) return;
  const int fidx = (wave_global - b * groups) * kFacesPerWave + lane;
  float f[9];
  const bool have = lane < kFacesPerWave && fidx < F;
  if (have) {
    load_face<INDEXED>(src, faces, b, F, NV, fidx, f);
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = 0.f;
  }
  const FaceSetup s = face_setup(f, width, height);
  float *zimg = zbuf + (size_t)b * width * height;
  __shared__ __attribute__((aligned(16))) unsigned char s_scratch[4][kWaveScratchBytes];
  raster_batch<kFacesPerWave, SHR_TRI_ATOMIC_LEVELS != 0>(s, have, lane, s_scratch[threadIdx.x >> 6], height,
               [&](int xi, int yi, float zp) { zmin(zimg + (size_t)yi * width + xi, zp); });
}

// ---- the band kernel (round 5) ----------------------------------------------------------------------------------
// The image is produced in BANDS of R rows held in LDS: every covered pixel is an LDS integer minimum, every image
// pixel is written to HBM exactly once with full-line stores -- no fill pass, no L2 atomics, and the float image is
// the reference's bit for bit (a minimum does not depend on the order it is taken in).
//   grid = (B, segments), block = 16 waves, dynamic LDS = row ranges [F] u32 | band's faces [F] u16 | band [R x W] f32 |
//   16 wave scratches.
//   phase 1  lanes = faces, once per workgroup: the reference's culls (.cu:33, :54, :68-69) and the conservative row
//            range of the face's spans -> s_range[f] = r_lo | r_hi << 16 (0xFFFF: culled), and the number of faces that
//            reach each band;
//   phase 2  per band of the segment: a band no face reaches is 1000.0 straight from registers; otherwise the band is
//            initialised in LDS, every wave walks its share of the ranges (one LDS word per face), collects the faces
//            that reach the band and rasterizes them 32 at a time (raster_batch, rows clipped to the band) with an LDS
//            minimum per covered pixel, and the band is streamed out.
constexpr int kBandWaves = 16;
constexpr int kBandFaces = 16;            // faces per batch: a wave's share of a band's faces is about that, and sixteen rows of
                                          // scratch less per wave are eleven more rows of band (2.8 instead of 4.5 KB per wave)
constexpr int kBandScratchBytes = wave_scratch_bytes(kBandFaces);
constexpr int kBandMaxBands = 512;        // per-band face counters in LDS
constexpr uint32_t kFillBits = 0x447A0000u;   // 1000.0f, .cu:122

// the culls and the row range of face_setup without its twelve divisions (same comparisons, same values)
__device__ __forceinline__ uint32_t face_row_range(const float f[9], int width, int height) {
  bool live = !((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]));   // :33 back face
  int p0, p2;
  if (f[0] < f[3]) {
    p0 = (f[6] < f[0]) ? 2 : 0;
    p2 = (f[3] < f[6]) ? 2 : 1;
  } else {
    p0 = (f[6] < f[3]) ? 2 : 1;
    p2 = (f[0] < f[6]) ? 2 : 0;
  }
  const float x0 = p0 == 0 ? f[0] : (p0 == 1 ? f[3] : f[6]), x2 = p2 == 0 ? f[0] : (p2 == 1 ? f[3] : f[6]);
  if (x0 == x2) live = false;                                                       // :54
  const int xi_min = cvt_rz_sat(fmaxf(ceilf(x0), 0.f));
  const int xi_max = cvt_rz_sat(fminf(x2, (float)width - 1.f));
  if (xi_min > xi_max) live = false;
  const float ylo = fminf(fminf(f[1], f[4]), f[7]), yhi = fmaxf(fmaxf(f[1], f[4]), f[7]);
  const bool wild = !(fabsf(ylo) < 1e9f) || !(fabsf(yhi) < 1e9f) || x2 < 0.f;
  const float yeps = 1e-5f * (fabsf(ylo) + fabsf(yhi)) + 1e-4f;
  const int r_lo = wild ? 0 : max(0, (int)ceilf(ylo - yeps));
  const int r_hi = wild ? height - 1 : min(height - 1, max(0, (int)floorf(yhi + yeps)));
  if (r_hi < r_lo) live = false;
  return live ? ((uint32_t)r_lo | ((uint32_t)r_hi << 16)) : 0xFFFFu;
}

// RESIZE (DepthRasterization.forward's tail, mesh/render.py:286, :311, for sizes the lattice kernel does not take -- S = 256
// from 640): the band never leaves LDS as a 640 x 640 image; its stream-out IS clamp(max) + ATen's bilinear resize to
// S x S.  With src / S = p_src / p_out in lowest terms, output rows [k p_out, (k + 1) p_out) take their two source rows
// from [k p_src, (k + 1) p_src) (down-sampling: scale > 1), so bands of a multiple of p_src rows hold every tap of their
// own output rows: zbuf = out [B][S][S], R % p_src == 0, square images.  The rasterized values are the full-resolution
// kernel's (same code), the epilogue is mesh_depth_kernel's formula: the same bits as raster -> clamp -> F.interpolate.
template <bool INDEXED, bool RESIZE = false>
__global__ void __launch_bounds__(kBandWaves * 64)
tri_band_kernel(const float *__restrict__ src, const int *__restrict__ faces, int B, int F, int NV, int width, int height,
                float *__restrict__ zbuf, int R, int nbands, int S = 0, int p_src = 1, int p_out = 1, float clamp_max = 0.f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Fpad = (F + 7) & ~7;
  uint32_t *s_range = reinterpret_cast<uint32_t *>(smem);
  uint16_t *s_pend = reinterpret_cast<uint16_t *>(smem + (size_t)Fpad * 4);          // the faces that reach the current band
  float *s_band = reinterpret_cast<float *>(smem + (size_t)Fpad * 6);
  // (the band rounded up to 16 bytes: the scratch is read through float4 / uint2 -- odd widths would leave it 4-byte aligned)
  unsigned char *s_scr = smem + (size_t)Fpad * 6 + (((size_t)R * width * 4 + 15) & ~(size_t)15);
  __shared__ int s_bandcnt[kBandMaxBands];
  __shared__ int s_npend;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup y of a crop takes bands y, y + gridDim.y, ...: interleaved, so that every workgroup of a crop gets its
  // share of the hand's rows and of the empty ones (consecutive bands: the top and bottom segments were nearly free)
  const int band_first = blockIdx.y, band_step = gridDim.y;
  float *zimg = RESIZE ? zbuf + (size_t)b * S * S : zbuf + (size_t)b * width * height;
  const bool counted = nbands <= kBandMaxBands;
  for (int i = tid; i < min(nbands, kBandMaxBands); i += kBandWaves * 64) s_bandcnt[i] = 0;
  __syncthreads();
  // ---- phase 1: row ranges -------------------------------------------------------------------------------------
  const float rinv = 1.0f / (float)R;
  for (int fidx = tid; fidx < F; fidx += kBandWaves * 64) {
    float f[9];
    load_face<INDEXED>(src, faces, b, F, NV, fidx, f);
    const uint32_t rw = face_row_range(f, width, height);
    s_range[fidx] = rw;
    if (counted && rw != 0xFFFFu) {
      // (band numbers through a float quotient: a band too many on either side only costs that band its shortcut)
      const int ba = max(0, (int)((float)(rw & 0xffffu) * rinv) - 1), bb = min(nbands - 1, (int)((float)(rw >> 16) * rinv) + 1);
      for (int k = ba; k <= bb; k++)
        if ((int)(rw & 0xffffu) <= min(height, (k + 1) * R) - 1 && (int)(rw >> 16) >= k * R) atomicAdd(&s_bandcnt[k], 1);
    }
  }
  __syncthreads();
  // ---- phase 2: the bands ----------------------------------------------------------------------------------------
  const bool vec4 = (width & 3) == 0;
  unsigned char *scratch = s_scr + (size_t)wave * kBandScratchBytes;
  const int nchunks = (F + 63) >> 6;
  for (int band = band_first; band < nbands; band += band_step) {
    const int lo = band * R, hi = min(height, lo + R) - 1, rows = hi - lo + 1;
    const int npix = rows * width;
    // RESIZE: the band's output rows [oy_lo, oy_lo + orows)
    const int oy_lo = RESIZE ? (lo / p_src) * p_out : 0, orows = RESIZE ? (rows / p_src) * p_out : 0;
    float *gout = RESIZE ? zimg + (size_t)oy_lo * S : zimg + (size_t)lo * width;
    if (counted && s_bandcnt[band] == 0) {          // no face reaches the band (workgroup-uniform)
      if (RESIZE) {                                 // every tap is the clamped background: min(1000, clamp_max) exactly
        const float bgv = fminf(__uint_as_float(kFillBits), clamp_max);
        for (int i = tid; i < orows * S; i += kBandWaves * 64) gout[i] = bgv;
      } else
      if (vec4) {
        const v4u_t t = {kFillBits, kFillBits, kFillBits, kFillBits};
        for (int i = tid; i < (npix >> 2); i += kBandWaves * 64)
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(reinterpret_cast<float4 *>(gout) + i), "v"(t) : "memory");
      } else {
        for (int i = tid; i < npix; i += kBandWaves * 64) gout[i] = __uint_as_float(kFillBits);
      }
      continue;
    }
    // the faces that reach the band, collected by ALL waves into one list (face numbers cluster: a wave's own chunks
    // hold all of a band's faces or none), then dealt out in equal shares
    if (tid == 0) s_npend = 0;
    if (vec4) {
      const float4 fv = make_float4(__uint_as_float(kFillBits), __uint_as_float(kFillBits), __uint_as_float(kFillBits),
                                    __uint_as_float(kFillBits));
      for (int i = tid; i < (npix >> 2); i += kBandWaves * 64) reinterpret_cast<float4 *>(s_band)[i] = fv;
    } else {
      for (int i = tid; i < npix; i += kBandWaves * 64) s_band[i] = __uint_as_float(kFillBits);
    }
    __syncthreads();
    for (int c = wave; c < nchunks; c += kBandWaves) {
      const int fidx = (c << 6) + lane;
      const uint32_t rw = fidx < F ? s_range[fidx] : 0xFFFFu;
      const bool reach = (int)(rw & 0xffffu) <= hi && (int)(rw >> 16) >= lo;
      const unsigned long long m = __ballot(reach);
      if (m == 0ull) continue;
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_npend, __popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (reach)
        s_pend[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint16_t)fidx;
    }
    __syncthreads();
    float *cells = s_band - (size_t)lo * width;      // cell of pixel (yi, xi) = cells[yi * width + xi]
    auto sink = [&](int xi, int yi, float zp) {
      float *cell = cells + yi * width + xi;
      const uint32_t bits = __float_as_uint(zp);
      if (bits >> 31) atomicMax(reinterpret_cast<unsigned int *>(cell), bits);   // (zmin's rule on an LDS cell)
      else atomicMin(reinterpret_cast<int *>(cell), (int)bits);
    };
    {
      // equal shares of the list, 16 faces at a time.  (Drawing batches of 16 from a counter instead -- a face costs what
      // its box holds -- measured 1022 us against 634 for 256 crops: a batch costs its LATENCY, the gather of its
      // vertices and the set-up's chain of divisions, whatever it holds; fewer, fuller batches win.)
      // The shares are STRIDED (wave w takes entries w, w + 16, ...): the list is in face order, neighbouring faces are
      // neighbours on the hand and of one size -- a contiguous share was all palm or all finger tips, and the band
      // waited for the wave with the palm.
      const int n = s_npend;
      const int mine = (n - wave + kBandWaves - 1) / kBandWaves;            // entries wave, wave + 16, ... below n
      for (int at = 0; at < mine; at += kBandFaces) {
        const int count = min(kBandFaces, mine - at);
        float f[9];
        const bool have = lane < count;
        if (have) {
          load_face<INDEXED>(src, faces, b, F, NV, (int)s_pend[wave + kBandWaves * (at + lane)], f);
        } else {
#pragma unroll
          for (int k = 0; k < 9; k++) f[k] = 0.f;
        }
        FaceSetup fs = face_setup(f, width, height);
        fs.r_lo = max(fs.r_lo, lo);
        fs.r_hi = min(fs.r_hi, hi);
        raster_batch<kBandFaces, SHR_TRI_BAND_LEVELS != 0>(fs, have, lane, scratch, height, sink);
      }
    }
    __syncthreads();
    // stream the band out
    if (RESIZE) {
      const float scale = (float)height / (float)S;
      for (int i = tid; i < orows * S; i += kBandWaves * 64) {
        const int r = i / S, ox = i - r * S;
        const Lin ly = lin_index(oy_lo + r, scale, height), lx = lin_index(ox, scale, width);
        const float *r0 = cells + ly.i0 * width, *r1 = cells + min(ly.i1, hi) * width;   // (i1 <= hi whenever scale > 1)
        const float v00 = fminf(r0[lx.i0], clamp_max), v01 = fminf(r0[lx.i1], clamp_max);
        const float v10 = fminf(r1[lx.i0], clamp_max), v11 = fminf(r1[lx.i1], clamp_max);
        gout[i] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
      }
    } else
    if (vec4) {
      for (int i = tid; i < (npix >> 2); i += kBandWaves * 64) {
        const uint4 v = reinterpret_cast<const uint4 *>(s_band)[i];
        const v4u_t t = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(reinterpret_cast<float4 *>(gout) + i), "v"(t) : "memory");
      }
    } else {
      for (int i = tid; i < npix; i += kBandWaves * 64) gout[i] = s_band[i];
    }
    __syncthreads();                                  // the band's cells are re-initialised next
  }
}

__global__ void zbuf_fill_kernel(uint4 *__restrict__ z, size_t n4, uint32_t key, uint32_t *__restrict__ tail,
                                 int ntail) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const v4u_t v = {key, key, key, key};
  for (; i < n4; i += stride) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(z + i), "v"(v) : "memory");
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = key;
}

// ---------------------------------------------------------------------------------------
// Skinning + camera.  One thread per vertex and kLbsCrops samples: the samples' bone matrices are staged in LDS, a
// vertex's skin entries (bone, weight * vertex) are read ONCE for the kLbsCrops samples (one sample per workgroup row
// re-read the shared 0.5-MB table for every sample: 21 -> 1x us for 256 crops, round 3).  Visits only the non-zero
// (bone, vertex) pairs of the reference's dense sum, in ascending bone order (association documented in DESIGN.md);
// per sample the arithmetic is unchanged.
constexpr int kLbsCrops = 4;
__global__ void __launch_bounds__(256)
lbs_project_kernel(const float *__restrict__ T, int B, int NB, int NV, const int *__restrict__ vstart,
                   const int *__restrict__ sbone, const float4 *__restrict__ swv, int right_hand, int project,
                   float cx, float cy, float fx, float fy, const float *__restrict__ rand_f,
                   float4 *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_T[];   // [kLbsCrops][NB][16]
  const int b0 = blockIdx.y * kLbsCrops;
  const int nb = min(kLbsCrops, B - b0);
  for (int i = threadIdx.x; i < nb * NB * 16; i += blockDim.x) s_T[i] = T[(size_t)b0 * NB * 16 + i];
  __syncthreads();
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= NV) return;
  float acc[kLbsCrops][4];
#pragma unroll
  for (int c = 0; c < kLbsCrops; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) acc[c][r] = 0.f;
  for (int e = vstart[v]; e < vstart[v + 1]; e++) {
    const int bone = sbone[e];
    const float4 q = swv[e];
#pragma unroll
    for (int c = 0; c < kLbsCrops; c++) {
      if (c >= nb) continue;
      lbs_add_entry(acc[c], s_T + (c * NB + bone) * 16, q);
    }
  }
#pragma unroll
  for (int c = 0; c < kLbsCrops; c++) {
    if (c >= nb) continue;
    const int b = b0 + c;
    const float4 o = lbs_finish(acc[c], right_hand, project, cx, cy, fx, fy, rand_f != nullptr, rand_f ? rand_f[b] : 0.f);
    // (written through: the vertices are read next by the rasterizer, left dirty they are flushed at the kernel's end)
    const v4u_t t = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(out + (size_t)b * NV + v), "v"(t) : "memory");
  }
}

}  // namespace shr

using namespace shr;

static int g_tri_band = -1;   // SHR_TUNE_TRI_BAND: -1 = by batch size (below), 0 = never (the global-atomic kernel), n > 0 = always, bands of <= n rows
int shr::tri_set_band(int v) { g_tri_band = v; return SHR_OK; }

// resize_S > 0: the band kernel with the clamp + resize epilogue (RESIZE above; depth = [B][S][S]); -1 when it does not fit
static int tri_raster_common(bool indexed, const float *src, const int *faces, int B, int F, int NV, int W, int H,
                             float *depth, hipStream_t s, int resize_S = 0, float clamp_max = 0.f) {
  int p_src = 1, p_out = 1;
  if (resize_S > 0) {
    if (W != H || resize_S >= H) return -1;   // (scale 1: the second tap, weight 0, is the NEXT period's first row)
    int a = H, c = resize_S;
    while (c) { const int t = a % c; a = c; c = t; }
    p_src = H / a; p_out = resize_S / a;
  }
  // The band kernel: the row ranges of all faces (4 F bytes) + sixteen wave scratches + a band of at least 8 rows
  // in one CU's LDS, 16-bit face numbers and row numbers.
  constexpr int kLds = 160 * 1024;
  const long long fixed = (long long)((F + 7) & ~7) * 6 + (long long)kBandWaves * kBandScratchBytes + 4096 + 16;   // (+ the static arrays: band counters, pending faces; + the band's rounding to 16 bytes)
  long long Rmax = (kLds - fixed) / (4LL * W);
  if (Rmax > H) Rmax = H;
  Rmax -= Rmax % p_src;                                  // (resize: whole periods of source rows per band)
  if (resize_S > 0 && (Rmax < p_src || g_tri_band == 0 || !(F > 0 && F <= 65535 && H <= 65535))) return -1;
  // Which kernel: the band kernel wherever it fits (round 5, after raster_batch's level walk: hand mesh @640x640,
  // tools/exp_tri_band.py -- 1 crop 20.8 us against 78 for the atomic kernel, 48 crops 142 against 173, 256 crops 348
  // against 461); the atomic kernel for what does not (more than 65535 faces, rows too wide for 8 of them in LDS).
  // SHR_TUNE_TRI_BAND: 0 forces the atomic kernel, n > 0 bands of at most n rows.
  if (g_tri_band != 0 && F > 0 && F <= 65535 && (Rmax >= 8 || resize_S > 0) && H <= 65535 && W <= 65535) {
    const int cus = device_cus();   // (per device: common.h)
    // Rows per band and workgroups per crop.  A band costs a fixed part -- one batch's latency: the gather of its faces,
    // the set-up's chain of divisions, three barriers -- worth ~24 rows of raster and stream-out (256 crops: 8.8 us per
    // band of 8 rows, 13.4 per band of 25), a workgroup its pass over all faces' row ranges (~10 rows' worth), and a
    // workgroup has the CU's LDS: what does not fit the CUs at once waits for a CU.  The plan minimises
    //     rounds x (10 + bands per workgroup x (24 + R)),   rounds = 1 if B x segs <= cus, else B x segs / cus + 1/2
    // over R and the workgroups per crop (their bands interleaved): a crop per CU -> the tallest band, one workgroup
    // per crop; few crops -> cus / B workgroups per crop at the height that leaves each the fewest rows; somewhat more
    // crops than CUs -> several workgroups per crop again, so that the second round is not a whole crop long (288
    // crops: 517 -> see EXPERIMENTS R3c).
    int R = (int)Rmax, segs = 1;
    if (g_tri_band > 0) {
      if (g_tri_band < R) R = g_tri_band;
      R = R < p_src ? p_src : R - R % p_src;
      const int nb = (H + R - 1) / R;
      segs = B >= cus ? 1 : (cus + B - 1) / B;
      if (segs > nb) segs = nb;
    } else {
      float best = -1.f;
      const int lo = (int)(Rmax < 8 ? Rmax : 8);
      for (int r = (int)Rmax; r >= lo; r -= p_src) {
        const int nb = (H + r - 1) / r;
        // far fewer crops than CUs: cus / B workgroups per crop, all resident at once; from half a crop per CU on: up to
        // 8 per crop (200 crops as 200 workgroups leave 56 CUs idle for a whole crop's time)
        int sg_lo = B >= cus ? 1 : cus / B, sg_hi = 2 * B > cus ? 8 : cus / B;
        if (sg_lo > nb) sg_lo = nb;
        if (sg_hi > nb) sg_hi = nb;
        for (int sg = sg_lo; sg <= sg_hi; sg++) {
          const int per = (nb + sg - 1) / sg;
          if (sg > sg_lo && (nb + sg - 2) / (sg - 1) == per) continue;      // (one workgroup fewer does the same)
          const float load = (float)B * (float)sg / (float)cus;
          const float rounds = load <= 1.f ? 1.f : load + 0.5f;
          const float cost = rounds * (10.f + (float)per * (float)(24 + r));
          if (best < 0.f || cost < best) { best = cost; R = r; segs = sg; }
        }
      }
    }
    const int nbands = (H + R - 1) / R;
    if (segs > nbands) segs = nbands;
    if (segs > 65535) segs = 65535;
    const size_t lds = (size_t)((F + 7) & ~7) * 6 + (((size_t)R * W * 4 + 15) & ~(size_t)15) + (size_t)kBandWaves * kBandScratchBytes;
    static AttrDone attr_done[4];   // per (kernel, device)
    auto launch = [&](auto kernel, int which) -> int {
      const hipError_t e = allow_dynamic_lds(kernel, kLds - 4096, &attr_done[which]);   // (the static arrays take the rest)
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(kernel, dim3((unsigned)B, (unsigned)segs), dim3(kBandWaves * 64), lds, s, src, faces, B, F, NV, W, H,
                         depth, R, nbands, resize_S, p_src, p_out, clamp_max);
      return (int)hipGetLastError();
    };
    if (resize_S > 0) return indexed ? launch(tri_band_kernel<true, true>, 2) : launch(tri_band_kernel<false, true>, 3);
    return indexed ? launch(tri_band_kernel<true>, 0) : launch(tri_band_kernel<false>, 1);
  }
  if (resize_S > 0) return -1;
  const size_t n = (size_t)B * W * H;
  uint32_t *z = reinterpret_cast<uint32_t *>(depth);
  const size_t n4 = n / 4;
  const int ntail = (int)(n - n4 * 4);
  // (16384 workgroups at most: 4096 left the fill of 256 crops at 74 us, 16384 at ~60 -- torch's fill of the same bytes
  // takes 64; write-through / non-temporal stores change nothing, tools/exp_tri_fill.sh in round 5)
  const size_t want_blocks = (n4 + 255) / 256;
  const unsigned fill_blocks = (unsigned)(want_blocks > 16384 ? 16384 : (want_blocks ? want_blocks : 1));
  hipLaunchKernelGGL(zbuf_fill_kernel, dim3(fill_blocks), dim3(256), 0, s, reinterpret_cast<uint4 *>(z), n4,
                     0x447A0000u /* 1000.0f, .cu:122 */, z + n4 * 4, ntail);
  if (F > 0) {
    const long long waves = (long long)B * ((F + kFacesPerWave - 1) / kFacesPerWave);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (indexed)
      hipLaunchKernelGGL(tri_raster_kernel<true>, dim3(blocks), dim3(256), 0, s, src, faces, B, F, NV, W, H, depth);
    else
      hipLaunchKernelGGL(tri_raster_kernel<false>, dim3(blocks), dim3(256), 0, s, src, faces, B, F, NV, W, H, depth);
  }
  return (int)hipGetLastError();
}

// shr_mesh_depth_fwd's non-lattice sizes (mesh_depth.hip): vertices + faces -> clamp + resize of the 640 x 640 raster as
// ONE band-kernel launch; -1 when the band kernel does not take the problem
int shr::tri_band_resize(const float *vertices, const int *faces, int B, int NV, int F, int src_size, int S, float clamp_max,
                         float *depth, hipStream_t s) {
  return tri_raster_common(true, vertices, faces, B, F, NV, src_size, src_size, depth, s, S, clamp_max);
}

extern "C" int shr_tri_raster_fwd(const float *face_vertices, int B, int F, int W, int H, float *depth,
                                  void *stream) {
  if (B == 0) return SHR_OK;
  if (!depth || (!face_vertices && F > 0) || B < 0 || F < 0 || W <= 0 || H <= 0) return SHR_EINVAL;
  if ((long long)B * W * H > (1LL << 40) || (long long)B * ((F + 31) / 32) > (1LL << 31) || W > 65535 || H > 65535)
    return SHR_ETOOLARGE;   // (pixel coordinates travel in 16 bits)
  if (((uintptr_t)depth & 15u) != 0) return SHR_EINVAL;
  return tri_raster_common(false, face_vertices, nullptr, B, F, 0, W, H, depth, (hipStream_t)stream);
}

extern "C" int shr_tri_raster_indexed_fwd(const float *vertices, const int32_t *faces, int B, int NV, int F, int W,
                                          int H, float *depth, void *stream) {
  if (B == 0) return SHR_OK;
  if (!depth || !vertices || (!faces && F > 0) || B < 0 || F < 0 || NV <= 0 || W <= 0 || H <= 0) return SHR_EINVAL;
  if ((long long)B * W * H > (1LL << 40) || (long long)B * ((F + 31) / 32) > (1LL << 31) || W > 65535 || H > 65535)
    return SHR_ETOOLARGE;
  if ((((uintptr_t)depth | (uintptr_t)vertices) & 15u) != 0) return SHR_EINVAL;
  return tri_raster_common(true, vertices, faces, B, F, NV, W, H, depth, (hipStream_t)stream);
}

extern "C" int shr_lbs_project(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                               const int32_t *skin_bone, const float *skin_wv, int right_hand, int project, float cx,
                               float cy, float fx, float fy, const float *rand_f, float *out, void *stream) {
  if (B == 0 || NV == 0) return SHR_OK;
  if (!T || !skin_vertex_start || !skin_bone || !skin_wv || !out || B < 0 || NB <= 0 || NV < 0) return SHR_EINVAL;
  if ((((uintptr_t)skin_wv | (uintptr_t)out) & 15u) != 0) return SHR_EINVAL;
  if (B > 65535 * kLbsCrops || NB > 160) return SHR_ETOOLARGE;   // (kLbsCrops x NB matrices of 64 bytes in LDS)
  dim3 grid((unsigned)((NV + 255) / 256), (unsigned)((B + kLbsCrops - 1) / kLbsCrops));
  hipLaunchKernelGGL(lbs_project_kernel, grid, dim3(256), (size_t)kLbsCrops * NB * 64, (hipStream_t)stream, T, B, NB, NV,
                     skin_vertex_start, skin_bone, reinterpret_cast<const float4 *>(skin_wv), right_hand, project, cx,
                     cy, fx, fy, rand_f, reinterpret_cast<float4 *>(out));
  return (int)hipGetLastError();
}
