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
// matrix), then lanes = pixels of the faces' boxes for the span test and lanes =
// pixels inside their span for the depth (see tri_raster_kernel below).
// Every pixel repeats the reference's per-column span test and per-pixel
// arithmetic verbatim (fp32, one rounding per written operator, IEEE division; the
// `1. / x` the reference evaluates in fp64 and rounds to fp32 equals the fp32
// quotient exactly -- 53 >= 2*24+2 bits).  The float min is a native integer
// atomic on the fp32 bits (signed min / unsigned max by sign: order independent,
// hence deterministic); the image holds plain floats throughout: one fill pass,
// one raster pass.
#include "common.h"

namespace shr {

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
__device__ __forceinline__ FaceSetup face_setup(const float f[9], int width, int height) {
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

// .cu:72-90: is row yi inside column xi's span of the face?  (x0, y0), (x1, y1): the first two vertices sorted by
// x; the slopes and their flags from face_setup.
__device__ __forceinline__ bool span_inside(float x0, float y0, float x1, float y1, float s01, float s12, float s02,
                                            int sflags, int xi, int yi, int height) {
  const float xf = (float)xi;
  float yi1;
  if (xf <= x1) yi1 = (sflags & 1) ? s01 * (xf - x0) + y0 : y1;
  else yi1 = (sflags & 2) ? s12 * (xf - x1) + y1 : y1;
  const float yi2 = s02 * (xf - x0) + y0;
  const int yi_min = cvt_rz_sat(fmaxf(0.f, ceilf(fminf(yi1, yi2))));
  const int yi_max = cvt_rz_sat(fminf(fmaxf(yi1, yi2), (float)height - 1.f));
  return yi >= yi_min && yi <= yi_max;
}

// .cu:97-110 for a pixel inside its column's span
__device__ __forceinline__ void span_pixel(const float pz[3], const float fi[9], int xi, int yi, int width, float *zimg) {
  const float xf = (float)xi, yf = (float)yi;
  float w[3];
  float w_sum = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    w[k] = (fi[3 * k + 0] * xf + fi[3 * k + 1] * yf) + fi[3 * k + 2];
    w[k] = fminf(fmaxf(w[k], 0.f), 1.f);
    w_sum += w[k];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) w[k] = w[k] / w_sum;
  const float zp = 1.0f / ((w[0] / pz[0] + w[1] / pz[1]) + w[2] / pz[2]);
  if (zp == zp) zmin(zimg + (size_t)yi * width + xi, zp);  // fminf(NaN, old) = old
}

// A visible face of the hand mesh covers a ~16x16-pixel box at 640x640 and one box pixel in
// three lies inside its column's span.  A wave sets 64 faces up with lanes = faces and parks
// each face's values in an LDS row; then
//   pass A, lanes = box pixels: faces up to 256 box pixels are worked through in GROUPS of 16
//     consecutive box pixels, four groups -- of one face or of four -- per pass (larger faces:
//     the whole wave on 8x8 patches, values through SGPRs); a lane only runs the SPAN TEST
//     (3 of the 10 IEEE divisions) and queues (face, x, y) in LDS if its pixel is inside;
//   pass B, lanes = queued pixels, 64 at a time, every lane busy: the 7 divisions of the
//     barycentric weights and the perspective depth, then the atomic.
constexpr int kFaceRow = 28;          // x0 y0 x1 y1 x2 y2 s01 s12 | z0 z1 z2 fi[9] | box x0, r_lo, bw, area, first group, 2^20 / bw, s02, slope flags
constexpr int kGroupsPerFace = 16;    // faces with a box above 256 pixels are visited alone
constexpr int kFacesPerWave = 32;     // 5 KB of LDS per wave (rows, group table, queue): eight waves per SIMD fit

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
  bool have = lane < kFacesPerWave && fidx < F;
  if (have) {
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
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = 0.f;
  }
  FaceSetup s = face_setup(f, width, height);
  float *zimg = zbuf + (size_t)b * width * height;
  __shared__ __attribute__((aligned(16))) float s_face[4][kFacesPerWave][kFaceRow];
  __shared__ uint8_t s_gface[4][kFacesPerWave * kGroupsPerFace];
  __shared__ uint2 s_queue[4][128];
  const int wv = threadIdx.x >> 6;

  const int bw = s.xi_max - s.xi_min + 1, bh = s.r_hi - s.r_lo + 1;
  const bool alive = have && s.live && bh > 0 && bw > 0;
  const long long area_ll = (long long)bw * bh;
  const bool big = alive && area_ll > 16 * kGroupsPerFace;
  const int area = alive && !big ? (int)area_ll : 0;
  const int ng = (area + 15) >> 4;
  // inclusive scan of the group counts over the wave
  int incl = ng;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
  {
    const int r0s = __builtin_amdgcn_readlane(incl, 15), r1s = __builtin_amdgcn_readlane(incl, 31);
    const int r2s = __builtin_amdgcn_readlane(incl, 47);
    const int row = lane >> 4;
    incl += (row >= 1 ? r0s : 0) + (row >= 2 ? r1s : 0) + (row >= 3 ? r2s : 0);
  }
  const int first = incl - ng, total = __builtin_amdgcn_readlane(incl, 63);
  if (lane < kFacesPerWave) {
    float *r = s_face[wv][lane];
#pragma unroll
    for (int a = 0; a < 3; a++) { r[2 * a] = s.p[a][0]; r[2 * a + 1] = s.p[a][1]; r[8 + a] = s.p[a][2]; }
#pragma unroll
    for (int k = 0; k < 9; k++) r[11 + k] = s.fi[k];
    r[20] = __int_as_float(s.xi_min); r[21] = __int_as_float(s.r_lo); r[22] = __int_as_float(bw);
    r[23] = __int_as_float(area); r[24] = __int_as_float(first);
    r[25] = __int_as_float(bw > 0 ? (int)(((1u << 20) + (unsigned)bw - 1u) / (unsigned)bw) : 0);
    r[6] = s.s01; r[7] = s.s12; r[26] = s.s02; r[27] = __int_as_float(s.sflags);
  }
  {
    const int ngmax = (int)wave_minmax_all<false>((float)ng);
    for (int k = 0; k < ngmax; k++)
      if (k < ng) s_gface[wv][first + k] = (uint8_t)lane;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's LDS writes (rows, tables and queue are private to the wave)
  __builtin_amdgcn_wave_barrier();

  int qn = 0;   // queued pixels (wave-uniform)
  auto drain = [&](int take) {   // pass B on the last `take` queued pixels
    if (lane < take) {
      const uint2 e = s_queue[wv][qn - take + lane];
      const float4 *r4 = reinterpret_cast<const float4 *>(s_face[wv][e.x]);
      const float4 b2 = r4[2], b3 = r4[3], b4 = r4[4];
      const float pz[3] = {b2.x, b2.y, b2.z};
      const float fi[9] = {b2.w, b3.x, b3.y, b3.z, b3.w, b4.x, b4.y, b4.z, b4.w};
      span_pixel(pz, fi, (int)(e.y & 0xffffu), (int)(e.y >> 16), width, zimg);
    }
    qn -= take;
  };
  auto push = [&](bool inside, int face, int xi, int yi) {
    const unsigned long long m = __ballot(inside);
    if (m == 0ull) return;
    if (inside) {
      const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      s_queue[wv][pos] = make_uint2((unsigned)face, (unsigned)xi | ((unsigned)yi << 16));
    }
    qn += __popcll(m);
    if (qn >= 64) drain(64);
  };

  for (int g0 = 0; g0 < total; g0 += 4) {
    const int g = g0 + (lane >> 4);
    bool inside = false;
    int face = 0, xi = 0, yi = 0;
    if (g < total) {
      face = s_gface[wv][g];
      const float4 *r4 = reinterpret_cast<const float4 *>(s_face[wv][face]);
      const float4 a0 = r4[0], a1 = r4[1], i0 = r4[5], i1 = r4[6];
      const int x0 = __float_as_int(i0.x), y0 = __float_as_int(i0.y), fbw = __float_as_int(i0.z);
      const int farea = __float_as_int(i0.w), ffirst = __float_as_int(i1.x);
      const unsigned inv = (unsigned)__float_as_int(i1.y);
      const int t = ((g - ffirst) << 4) + (lane & 15);
      if (t < farea) {
        const int ly = (int)(((unsigned)t * inv) >> 20);   // t / bw: exact for t <= 1024, bw <= 1024
        xi = x0 + (t - ly * fbw);
        yi = y0 + ly;
        inside = span_inside(a0.x, a0.y, a0.z, a0.w, a1.z, a1.w, i1.z, __float_as_int(i1.w), xi, yi, height);
      }
    }
    push(inside, face, xi, yi);
  }

  // the faces with a large box: the whole wave on 8x8 patches of one face, values through SGPRs
  unsigned long long live = __ballot(big);
  while (live) {
    const int src_lane = __builtin_amdgcn_readfirstlane(__builtin_ctzll(live));
    live &= live - 1;
    const float fx0 = readlane_f(s.p[0][0], src_lane), fy0 = readlane_f(s.p[0][1], src_lane);
    const float fx1 = readlane_f(s.p[1][0], src_lane), fy1 = readlane_f(s.p[1][1], src_lane);
    const float f01 = readlane_f(s.s01, src_lane), f12 = readlane_f(s.s12, src_lane), f02 = readlane_f(s.s02, src_lane);
    const int ffl = __builtin_amdgcn_readlane(s.sflags, src_lane);
    const int x0 = __builtin_amdgcn_readlane(s.xi_min, src_lane), x1 = __builtin_amdgcn_readlane(s.xi_max, src_lane);
    const int r0 = __builtin_amdgcn_readlane(s.r_lo, src_lane), r1 = __builtin_amdgcn_readlane(s.r_hi, src_lane);
    for (int oy = 0; oy <= r1 - r0; oy += 8)
      for (int ox = 0; ox <= x1 - x0; ox += 8) {
        const int xi = x0 + ox + (lane & 7), yi = r0 + oy + (lane >> 3);
        push(xi <= x1 && yi <= r1 && span_inside(fx0, fy0, fx1, fy1, f01, f12, f02, ffl, xi, yi, height), src_lane, xi, yi);
      }
  }
  if (qn > 0) drain(qn);
}

__global__ void zbuf_fill_kernel(uint4 *__restrict__ z, size_t n4, uint32_t key, uint32_t *__restrict__ tail,
                                 int ntail) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint4 v = make_uint4(key, key, key, key);
  for (; i < n4; i += stride) z[i] = v;
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
      const float *M = s_T + (c * NB + bone) * 16;
#pragma unroll
      for (int r = 0; r < 4; r++)
        acc[c][r] += ((M[4 * r] * q.x + M[4 * r + 1] * q.y) + M[4 * r + 2] * q.z) + M[4 * r + 3] * q.w;
    }
  }
#pragma unroll
  for (int c = 0; c < kLbsCrops; c++) {
    if (c >= nb) continue;
    const int b = b0 + c;
    float a0 = acc[c][0];
    if (right_hand) a0 = -a0;
    float4 o;
    if (!project) {
      o = make_float4(a0, acc[c][1], acc[c][2], acc[c][3]);
    } else if (!rand_f) {
      o = make_float4(fx * a0 + cx * acc[c][3], fy * acc[c][1] + cy * acc[c][3], acc[c][2], acc[c][3]);
    } else {
      const float rf = rand_f[b];
      o = make_float4(a0 * rf * fx + cx, acc[c][1] * rf * fy + cy, acc[c][2], 1.0f);
    }
    // (written through: the vertices are read next by the rasterizer, left dirty they are flushed at the kernel's end)
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    const v4u_t t = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(out + (size_t)b * NV + v), "v"(t) : "memory");
  }
}

}  // namespace shr

using namespace shr;

static int tri_raster_common(bool indexed, const float *src, const int *faces, int B, int F, int NV, int W, int H,
                             float *depth, hipStream_t s) {
  const size_t n = (size_t)B * W * H;
  uint32_t *z = reinterpret_cast<uint32_t *>(depth);
  const size_t n4 = n / 4;
  const int ntail = (int)(n - n4 * 4);
  const unsigned fill_blocks = (unsigned)((n4 + 255) / 256 > 4096 ? 4096 : ((n4 + 255) / 256 ? (n4 + 255) / 256 : 1));
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
