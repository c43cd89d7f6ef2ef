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

__device__ __forceinline__ bool is_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace shr
