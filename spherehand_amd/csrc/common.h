// common.h -- shared device helpers for libspherehand_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spherehand_hip.h"

// Parity with the reference is defined on its fp32 operation sequence: one
// correctly rounded IEEE operation per written operator, no FMA contraction
// (build also passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace shr {

constexpr int kWave = 64;  // CDNA4 wavefront

// Wave tile of the image rasterizers: 32 px wide x 8 px high; lane l owns the
// 4 consecutive pixels starting at column 4*(l&7) of row (l>>3): one 16-byte
// store per lane, eight lanes = one full 128-byte line per tile row.
constexpr int kTileW = 32;
constexpr int kTileH = 8;

// Image axis -> model-space millimetres, mesh/render.py:31-32:
//   (u - size/2) * 300.0 / size      (three fp32 ops: sub, mul, div)
// For a power-of-two size the division is an exact scaling, so
// round(t*300)/size == round(t*(300/size)) and the divide is skipped.
struct Axis {
  float half;   // size / 2
  float size;   // (float)size
  float mul;    // 300 / size   (used when pow2)
  int pow2;
};

__host__ __device__ inline Axis make_axis(int size) {
  Axis a;
  a.half = (float)((double)size / 2.0);
  a.size = (float)size;
  a.mul = 300.0f / (float)size;
  a.pow2 = (size & (size - 1)) == 0;
  return a;
}

// The same with 300 / size supplied by the launcher: the zbuf kernels take the two IEEE divisions of an image's axes
// (and the two of size / 300, the pixels per millimetre of the boxes) as launch constants -- sixteen waves per crop
// each spent ~45 VALU instructions on them (7 % of a crop's VALU work once the SIMDs are the contended resource).
struct AxisK { float mulx, muly, kx, ky; };   // 300 / W, 300 / H, W / 300, H / 300: fp32 divisions, host == device (IEEE)
inline AxisK make_axis_k(int W, int H) {
  AxisK k;
  k.mulx = 300.0f / (float)W;
  k.muly = 300.0f / (float)H;
  k.kx = (float)W / 300.0f;
  k.ky = (float)H / 300.0f;
  return k;
}
__device__ __forceinline__ Axis axis_of(int size, float mul) {
  Axis a;
  a.size = (float)size;
  a.half = a.size * 0.5f;   // exact (size < 2^24), == (float)((double)size / 2)
  a.mul = mul;
  a.pow2 = (size & (size - 1)) == 0;
  return a;
}

__device__ __forceinline__ float axis_coord(const Axis &a, int u) {
  const float t = (float)u - a.half;
  if (a.pow2) return t * a.mul;
  return (t * 300.0f) / a.size;
}

// ---- wave-level helpers ---------------------------------------------------
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  // lanes outside ROW_MASK (or whose DPP source is invalid) add 0
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(moved);
}

// Sum of v over the 64 lanes of the wave, valid in lane 63.  Fixed association
// (butterfly inside each row of 16, then row 0->1, 2->3, {0,1}->{2,3}):
// deterministic for a given lane assignment.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror       -> every lane holds its row's sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3
  return v;
}

// Four wave sums at once ("transposed" reduction): two quad exchanges leave ONE register
// whose lane l holds component l & 3 summed over its quad (9 ops for the 4 components),
// then the remaining four levels run on that single register -- row_shr:4, row_shr:8,
// and the gfx950 cross-row swaps (v_permlane16_swap / v_permlane32_swap: odd <-> even
// rows, upper <-> lower half).  15 VALU ops against 24 for four wave_sum_lane63, fixed
// association.  Lanes 12..15 of EVERY row return the totals of components 0..3; the
// other lanes return partial sums.
__device__ __forceinline__ float wave_sum4_transposed(float a0, float a1, float a2, float a3, int lane) {
  const bool odd = lane & 1, hi = lane & 2;
  const float k0 = odd ? a1 : a0, s0 = odd ? a0 : a1;
  const float k1 = odd ? a3 : a2, s1 = odd ? a2 : a3;
  const float b0 = k0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xF, 0xF, false));
  const float b1 = k1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xF, 0xF, false));
  const float k2 = hi ? b1 : b0, s2 = hi ? b0 : b1;
  float c = k2 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0x4E, 0xF, 0xF, false));
  c = dpp_add<0x114, 0xF>(c);   // row_shr:4
  c = dpp_add<0x118, 0xF>(c);   // row_shr:8  -> lanes 12..15 of a row: the row's sums
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(c), false, false);
  c = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);   // rows 0+1 | 2+3
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
  return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

// min / max over the 64 lanes, broadcast to every lane (same DPP butterfly; lanes whose
// DPP source is invalid or masked combine with their own value)
template <int CTRL, int ROW_MASK, bool IS_MIN>
__device__ __forceinline__ float dpp_minmax(float v) {
  const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
  return IS_MIN ? fminf(v, o) : fmaxf(v, o);
}
template <bool IS_MIN>
__device__ __forceinline__ float wave_minmax_all(float v) {
  v = dpp_minmax<0xB1, 0xF, IS_MIN>(v);
  v = dpp_minmax<0x4E, 0xF, IS_MIN>(v);
  v = dpp_minmax<0x141, 0xF, IS_MIN>(v);
  v = dpp_minmax<0x140, 0xF, IS_MIN>(v);
  v = dpp_minmax<0x142, 0xA, IS_MIN>(v);
  v = dpp_minmax<0x143, 0xC, IS_MIN>(v);
  return readlane_f(v, 63);
}

// Four wave minima at once, same scheme as wave_sum4_transposed: lanes 12..15 of every row
// return the minima of components 0..3 (fminf: a NaN operand is ignored).
__device__ __forceinline__ float wave_min4_transposed(float a0, float a1, float a2, float a3, int lane) {
  const bool odd = lane & 1, hi = lane & 2;
  const float k0 = odd ? a1 : a0, s0 = odd ? a0 : a1;
  const float k1 = odd ? a3 : a2, s1 = odd ? a2 : a3;
  const float b0 = fminf(k0, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(k0), __float_as_int(s0), 0xB1, 0xF, 0xF, false)));
  const float b1 = fminf(k1, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(k1), __float_as_int(s1), 0xB1, 0xF, 0xF, false)));
  const float k2 = hi ? b1 : b0, s2 = hi ? b0 : b1;
  float c = fminf(k2, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(k2), __float_as_int(s2), 0x4E, 0xF, 0xF, false)));
  c = dpp_minmax<0x114, 0xF, true>(c);   // row_shr:4 (lanes without a source keep their own value)
  c = dpp_minmax<0x118, 0xF, true>(c);   // row_shr:8  -> lanes 12..15 of a row: the row's minima
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(c), false, false);
  c = fminf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
  return fminf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}

// Correctly rounded fp32 square root for x in [0.01, 1e12] (sphere_zbuf.h documents the range and the
// exhaustive test).  s = v_sqrt_f32(x) is within one ulp u of the root, so the result is s - u, s or s + u, and
// which one is read off ONE residual r = fma(-s, s, x) (exact whenever it matters: |x - s^2| < 2^24 granules of
// s^2's last place there):  sqrt(x) > s + u/2  <=>  x - s^2 > s u (+ u^2/4, below a granule)  <=>  r > t,
// sqrt(x) < s - u/2  <=>  r <= -t,  with t = s * u formed on the bit pattern (exponent field of s added to itself).
// Eight VALU instructions where the two-neighbour form (two increments, two residuals, two selects) takes nine;
// the carry forms of the integer add / subtract take the comparison's result directly.
// NOT covered by the argument above: s an exact power of two, 2^k.  The ulp BELOW s is u / 2 there, so for
// x = prev(4^k) (23 inputs in the range) an estimate of 2^k would leave r = -4^k 2^-24 > -t and 2^k would be
// returned where the correctly rounded root is prev(2^k).  gfx950's v_sqrt_f32 returns prev(2^k) for those inputs
// (it never rounds the estimate UP across a binade boundary), which is what
// tests/test_sphere_raster_gpu.py::test_sqrt_rn_exhaustive establishes on the hardware, every fp32 value of the
// range, in the default GPU selection: the function is exact on gfx950, by measurement at those 23 points and by
// the argument everywhere else.  Another target needs the test re-run (or t halved for the lower test when s's
// mantissa is zero).  The s_nop counts below are gfx950's VALU-writes-vcc -> VALU-reads-vcc-as-carry wait states.
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float r = __builtin_fmaf(-s, s, x);
  const uint32_t sb = __float_as_uint(s);
  const float t = __uint_as_float(sb + (sb & 0x7f800000u) - 0x4B000000u);   // s * 2^(E - 23), E = s's exponent
  uint32_t out;
  // (s_nop 1: a VALU write of vcc needs two wait states before a VALU reads it as a carry / mask on gfx950, and
  // the hazard recognizer does not look inside an asm statement)
  asm("v_cmp_gt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, 0, %3, vcc\n\t"
      "v_cmp_le_f32 vcc, %1, -%2\n\ts_nop 1\n\tv_subbrev_co_u32 %0, vcc, 0, %0, vcc"
      : "=&v"(out) : "v"(r), "v"(t), "v"(sb) : "vcc");
  return __uint_as_float(out);
}

int d2m_set_waves(int waves);        // data_to_model.hip: launch-shape hooks behind SHR_TUNE_D2M_WAVES /
int d2m_set_band_units(int units);  // SHR_TUNE_D2M_BAND_UNITS
int d2m_set_tiled(int on);          // SHR_TUNE_D2M_TILED
int tri_set_band(int rows);         // tri_raster.hip: SHR_TUNE_TRI_BAND
int mesh_set_band(int on);          // mesh_depth.hip: SHR_TUNE_MESH_BAND
int tri_band_resize(const float *vertices, const int *faces, int B, int NV, int F, int src_size, int S, float clamp_max,
                    float *depth, hipStream_t s);   // tri_raster.hip: the band kernel with the clamp + resize epilogue (-1: not its problem)
// data_to_model.hip: the two halves of shr_data_to_model_compact (mutual_project.hip: shr_mv_project_compact)
int d2m_compact_check(const float *depth, int M, int H, int W, void *workspace, int **counts);
int d2m_compact_launch(const float *depth, int M, int H, int W, void *workspace, hipStream_t s);

// ---- the triangle pixel's seven IEEE divisions (.cu:97-110) with the denominators' work shared -------------------------
// hipcc's fp32 division a / d (-fhip-fp32-correctly-rounded-divide-sqrt) is
//     ds = v_div_scale(d, d, a); as = v_div_scale(a, d, a); r = v_rcp(ds); e = fma(-ds, r, 1); r1 = fma(e, r, r);
//     q0 = as * r1; e1 = fma(-ds, q0, as); q1 = fma(e1, r1, q0); e2 = fma(-ds, q1, as); q = v_div_fmas(e2, r1, q1);
//     v_div_fixup(q, d, a)
// and v_div_scale leaves BOTH operands alone (ds = d, as = a, v_div_fmas = fma, v_div_fixup = identity up to the sign it
// would give anyway) when d is normal and below 2^126, a is zero or at least 2^-103, and the exponents differ by less than
// 96 upwards and 126 downwards.  Inside that domain r1 depends on d only: three divisions by one denominator share it
// (w[k] / w_sum), and a denominator that is a constant of the face (its corners' z) brings it from the set-up.  These are
// the compiler's own instructions on the compiler's own operands -- the quotients are the same bits (shr_selftest_division
// compares them over random and edge operands; every parity test of the triangle kernels runs through them).
__device__ __forceinline__ float div_rcp_refined(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_with(float a, float d, float r1) {
  const float q0 = a * r1;
  const float e1 = __builtin_fmaf(-d, q0, a);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-d, q1, a);
  return __builtin_fmaf(e2, r1, q1);
}
// a corner depth whose reciprocal may be shared: 2^-40 <= |z| <= 2^40 (a hand's are within +-100 of the crop's centre; the
// sign rides through the same instructions as in the compiler's sequence, a zero quotient's included)
__device__ __forceinline__ bool div_tame_z(float z) { return fabsf(z) >= 0x1p-40f && fabsf(z) <= 0x1p40f; }
// The pixel: clamped barycentric weights w (each in [0, 1]), their sum, the corners' z and -- `tame`: all three
// div_tame_z -- their refined reciprocals rz.  Fast path when every weight is zero or at least 2^-60 (the sum is then
// within [2^-60, 3], w / w_sum zero or at least 2^-62, and that over z zero or at least 2^-102: all inside the domain
// above); the plain divisions otherwise.
__device__ __forceinline__ float tri_pixel_depth(float w0, float w1, float w2, float w_sum, const float (&pz)[3],
                                                 const float (&rz)[3], bool tame) {
  // "every weight is zero or at least 2^-60" on the bit patterns of the non-negative weights: bits - 1 wraps a zero
  // to the top, so one unsigned minimum and one compare.  (The sum needs no test of its own: it is at least the largest
  // weight; all three zero or a NaN among them give NaN on either path, and the pixel is skipped.)  As a chain of && / ||
  // over float compares the test compiled into a branch per clause and cost what the shared reciprocals save: 256 crops
  // 284 us, 275 as one mask of compares, against 264 with no test of the weights at all.
  const uint32_t t = __float_as_uint(0x1p-60f) - 1u;
  const uint32_t m = min(min(__float_as_uint(w0) - 1u, __float_as_uint(w1) - 1u), __float_as_uint(w2) - 1u);
  const bool ok = tame & (m >= t);
  if (ok) {
    const float rs = div_rcp_refined(w_sum);
    const float u0 = div_with(div_with(w0, w_sum, rs), pz[0], rz[0]);
    const float u1 = div_with(div_with(w1, w_sum, rs), pz[1], rz[1]);
    const float u2 = div_with(div_with(w2, w_sum, rs), pz[2], rz[2]);
    return 1.0f / ((u0 + u1) + u2);
  }
  w0 = w0 / w_sum; w1 = w1 / w_sum; w2 = w2 / w_sum;
  return 1.0f / ((w0 / pz[0] + w1 / pz[1]) + w2 / pz[2]);
}

// Linear blend skinning of one vertex for one sample, shared by lbs_project_kernel (tri_raster.hip) and the fused
// mesh_lattice_kernel (mesh_depth.hip): one (bone, weighted vertex) entry added to the four rows of the sum, and the
// sign flip + orthographic camera of mesh/render.py:320-329 on the finished sum.  M = the bone's 4 x 4 matrix.
__device__ __forceinline__ void lbs_add_entry(float (&acc)[4], const float *M, const float4 q) {
#pragma unroll
  for (int r = 0; r < 4; r++)
    acc[r] += ((M[4 * r] * q.x + M[4 * r + 1] * q.y) + M[4 * r + 2] * q.z) + M[4 * r + 3] * q.w;
}
__device__ __forceinline__ float4 lbs_finish(const float (&acc)[4], int right_hand, int project, float cx, float cy, float fx,
                                             float fy, bool has_rand, float rf) {
  float a0 = acc[0];
  if (right_hand) a0 = -a0;
  if (!project) return make_float4(a0, acc[1], acc[2], acc[3]);
  if (!has_rand) return make_float4(fx * a0 + cx * acc[3], fy * acc[1] + cy * acc[3], acc[2], acc[3]);
  return make_float4(a0 * rf * fx + cx, acc[1] * rf * fy + cy, acc[2], 1.0f);
}

// inclusive prefix sum over the 64 lanes (4 DPP steps inside each row of 16, then the row totals)
__device__ __forceinline__ int wave_scan_incl(int v, int lane) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
  const int r0s = __builtin_amdgcn_readlane(v, 15), r1s = __builtin_amdgcn_readlane(v, 31), r2s = __builtin_amdgcn_readlane(v, 47);
  const int row = lane >> 4;
  return v + (row >= 1 ? r0s : 0) + (row >= 2 ? r1s : 0) + (row >= 3 ? r2s : 0);
}
// Item k of a sequence cut into 64 consecutive runs whose inclusive ends the lanes hold (non-decreasing; `mine`: this
// lane's run exists): the run that holds k = the number of runs that END at or before k.  Those that end at or
// before k0 (wave-uniform, <= k) are counted by one ballot, the few that end inside [k0, k0 + 63] one by one.
__device__ __forceinline__ int run_of(int incl, bool mine, int k0, int k) {
  int run = __popcll(__ballot(mine && incl <= k0));
  unsigned long long inner = __ballot(mine && incl > k0 && incl <= k0 + 63);
  while (inner) {
    const int c = __builtin_ctzll(inner);
    inner &= inner - 1;
    run += (__builtin_amdgcn_readlane(incl, c) <= k) ? 1 : 0;
  }
  return run;
}


__device__ __forceinline__ bool is_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// ATen area_pixel_compute_source_index (align_corners=False) for output index d:
// src = scale*(d+0.5)-0.5 clamped at 0; i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0.
struct Lin { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lin lin_index(int d, float scale, int in_size) {
  // (one rounding: ATen's GPU kernel is compiled with contraction, scale * (d + 0.5) - 0.5 is an FMA there;
  // the two forms differ by an ulp of the source index at non-dyadic ratios, 3e-5 of a pixel at 256)
  float src = __builtin_fmaf(scale, (float)d + 0.5f, -0.5f);
  if (src < 0.f) src = 0.f;
  Lin r;
  r.i0 = min((int)src, in_size - 1);
  r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}

// ---- counter-based random numbers of the synthetic branch (HandSynthesizer: RandScale, focal jitter, DepthNoise) --------
// No state is carried between draws: a draw is a HASH of what it is for, so any launch geometry -- and a numpy
// restatement (spherehand_amd/synth_rng.py) -- produces the same numbers.
//   rng_hash        lowbias32 (C. Wellons' hash-prospector: xorshift-multiply, 2 rounds; bias 0.17 of an ideal hash)
//   rng_key         the stream word k of sample b in call `ctr` under `seed`: a chain of rng_hash over the six words
//   rng_uniform     the top 24 bits as a float in [0, 1) -- torch.rand's float32 construction
//   pixel p of a sample's noise field: h1 = rng_hash(key0 + p), h2 = rng_hash(key1 + p)  (mesh_depth.hip / synth_post.hip)
__host__ __device__ inline uint32_t rng_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t rng_key(unsigned long long seed, unsigned long long ctr, uint32_t b, uint32_t k) {
  uint32_t h = rng_hash((uint32_t)seed ^ 0x9e3779b9u);
  h = rng_hash(h ^ (uint32_t)(seed >> 32));
  h = rng_hash(h ^ (uint32_t)ctr);
  h = rng_hash(h ^ (uint32_t)(ctr >> 32) ^ 0x85ebca6bu);
  h = rng_hash(h ^ b);
  return rng_hash(h ^ (k + 0x27d4eb2fu));
}
__host__ __device__ inline float rng_uniform(uint32_t h) { return (float)(h >> 8) * 5.9604644775390625e-8f; }   // 2^-24

// DepthNoise (network/util_modules.py:46-84) on pixel `p` of a sample whose stream keys are (key0, key1):
//   shift per axis  trunc(n * sigma_xy + 0.5), n ~ N(0, 1): a 16-bit uniform against the three cumulative thresholds
//                   thr = (P(shift < 0), P(shift < 1), P(shift < 2)) x 65536 -- shifts -1 .. +2; what lies beyond them
//                   (3e-7 at the reference's sigma 0.5) is folded into the outer two.  x: h1's high half, y: its low half.
//   depth noise     Box-Muller on h2's halves: sqrt(-2 ln u1) cos(2 pi u2), u1 = (hi + 0.5) / 65536, u2 = lo / 65536.
struct NoiseShift { int dx, dy; };
__device__ __forceinline__ NoiseShift noise_shift(uint32_t key0, uint32_t p, uint32_t t0, uint32_t t1, uint32_t t2) {
  const uint32_t h = rng_hash(key0 + p);
  const uint32_t ux = h >> 16, uy = h & 0xffffu;
  NoiseShift s;
  s.dx = -1 + (int)(ux >= t0) + (int)(ux >= t1) + (int)(ux >= t2);
  s.dy = -1 + (int)(uy >= t0) + (int)(uy >= t1) + (int)(uy >= t2);
  return s;
}
__device__ __forceinline__ float noise_normal(uint32_t key1, uint32_t p) {
  const uint32_t h = rng_hash(key1 + p);
  const float u1 = ((float)(h >> 16) + 0.5f) * 1.52587890625e-5f;    // (0, 1)
  const float u2 = (float)(h & 0xffffu) * 1.52587890625e-5f;         // [0, 1): v_cos_f32 takes revolutions
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln 2 * log2(u1)
  return r * __builtin_amdgcn_cosf(u2);
}

// ---- host side: what a launcher caches PER DEVICE -------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device function attribute and the CU count a per-device number:
// a process that launches on a second GPU must set / read them there too (ops._on() may launch on a non-current device).
constexpr int kMaxDevices = 64;
struct AttrDone { bool dev[kMaxDevices] = {}; };

// the opt-in to `bytes` of dynamic LDS for `kernel` on the CURRENT device, once per (kernel, device)
template <typename K>
inline hipError_t allow_dynamic_lds(K kernel, int bytes, AttrDone *done) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = -1;
  if (d >= 0 && done->dev[d]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && d >= 0) done->dev[d] = true;
  return e;
}

// compute units of the current device (256 when it cannot be asked)
inline int device_cus() {
  static int cus[kMaxDevices] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return 256;
  if (cus[d] == 0) {
    int v = 0;
    cus[d] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && v > 0) ? v : 256;
  }
  return cus[d];
}

}  // namespace shr
