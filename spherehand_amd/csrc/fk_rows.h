// fk_rows.h -- the row-chain arithmetic of the hand's forward kinematics (fk.hip documents the scheme), shared by the
// pose kernels (fk.hip) and the one-launch synthesizer (mesh_depth.hip: the rasterizer's workgroup runs its crop's FK).
#pragma once
#include "common.h"

namespace shr {

// u * M for an affine M with rows m0, m1, m2 (and an implied 0 0 0 1): u = (r0, r1, r2, t)
__device__ __forceinline__ float4 row_times(const float4 u, const float4 m0, const float4 m1, const float4 m2) {
  float4 o;
  o.x = __builtin_fmaf(u.z, m2.x, __builtin_fmaf(u.y, m1.x, u.x * m0.x));
  o.y = __builtin_fmaf(u.z, m2.y, __builtin_fmaf(u.y, m1.y, u.x * m0.y));
  o.z = __builtin_fmaf(u.z, m2.z, __builtin_fmaf(u.y, m1.z, u.x * m0.z));
  o.w = __builtin_fmaf(u.z, m2.w, __builtin_fmaf(u.y, m1.w, u.x * m0.w)) + u.w;
  return o;
}
// its adjoint: u_bar given o_bar
__device__ __forceinline__ float4 row_times_adj(const float4 ob, const float4 m0, const float4 m1, const float4 m2) {
  float4 u;
  u.x = __builtin_fmaf(ob.w, m0.w, __builtin_fmaf(ob.z, m0.z, __builtin_fmaf(ob.y, m0.y, ob.x * m0.x)));
  u.y = __builtin_fmaf(ob.w, m1.w, __builtin_fmaf(ob.z, m1.z, __builtin_fmaf(ob.y, m1.y, ob.x * m1.x)));
  u.z = __builtin_fmaf(ob.w, m2.w, __builtin_fmaf(ob.z, m2.z, __builtin_fmaf(ob.y, m2.y, ob.x * m2.x)));
  u.w = ob.w;
  return u;
}

// (sin, cos, (1 - cos) + cos) of one angle: the three numbers the reference's axis-aligned rotation matrix holds
struct Rot { float s, c, d; };

// u * R for the reference's rotation about +x: R = [[d,0,0],[0,c,-s],[0,s,c]] (AxisRotationMatrix with axis 1 0 0)
__device__ __forceinline__ float4 rot_x(const float4 u, const Rot r) {
  return make_float4(u.x * r.d, __builtin_fmaf(u.z, r.s, u.y * r.c), __builtin_fmaf(u.z, r.c, -(u.y * r.s)), u.w);
}
__device__ __forceinline__ float4 rot_x_adj(const float4 ob, const Rot r) {
  return make_float4(ob.x * r.d, __builtin_fmaf(-ob.z, r.s, ob.y * r.c), __builtin_fmaf(ob.z, r.c, ob.y * r.s), ob.w);
}
__device__ __forceinline__ float rot_x_dangle(const float4 ob, const float4 out) { return ob.y * out.z - ob.z * out.y; }
// about +z: R = [[c,-s,0],[s,c,0],[0,0,d]]
__device__ __forceinline__ float4 rot_z(const float4 u, const Rot r) {
  return make_float4(__builtin_fmaf(u.y, r.s, u.x * r.c), __builtin_fmaf(u.y, r.c, -(u.x * r.s)), u.z * r.d, u.w);
}
__device__ __forceinline__ float4 rot_z_adj(const float4 ob, const Rot r) {
  return make_float4(__builtin_fmaf(-ob.y, r.s, ob.x * r.c), __builtin_fmaf(ob.y, r.c, ob.x * r.s), ob.z * r.d, ob.w);
}
__device__ __forceinline__ float rot_z_dangle(const float4 ob, const float4 out) { return ob.x * out.y - ob.y * out.x; }
// about +y: R = [[c,0,s],[0,d,0],[-s,0,c]]
__device__ __forceinline__ float4 rot_y(const float4 u, const Rot r) {
  return make_float4(__builtin_fmaf(-u.z, r.s, u.x * r.c), u.y * r.d, __builtin_fmaf(u.z, r.c, u.x * r.s), u.w);
}
__device__ __forceinline__ float4 rot_y_adj(const float4 ob, const Rot r) {
  return make_float4(__builtin_fmaf(ob.z, r.s, ob.x * r.c), ob.y * r.d, __builtin_fmaf(ob.z, r.c, -(ob.x * r.s)), ob.w);
}
__device__ __forceinline__ float rot_y_dangle(const float4 ob, const float4 out) { return ob.z * out.x - ob.x * out.z; }

// The abduction axis is +z for fingers 0, 1, 4 and -y for fingers 2, 3 (:162-164).  About -y the matrix is
// [[c,0,-s],[0,d,0],[s,0,c]] = the +z form with the y and z components exchanged on both sides.
__device__ __forceinline__ float4 swap_yz(const float4 u, bool on) { return on ? make_float4(u.x, u.z, u.y, u.w) : u; }

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

constexpr int kBones = 17;
constexpr int kAngles = 23;          // palm Euler angles + 5 x 4 finger angles
constexpr int kMaxPoints = 128;      // key-points the one-launch backward stages per sample (the hand has 41)

// One sample's sincos table: lanes 0..22 take one angle each.
__device__ __forceinline__ void sincos_phase(const float *__restrict__ p, int lane, Rot *sc) {
  if (lane < kAngles) {
    const float a = p[lane < 3 ? lane : lane + 3];
    float s, c;
    sincosf(a, &s, &c);
    Rot r;
    r.s = s;
    r.c = c;
    r.d = (1.0f - c) + c;            // xx * i + c with xx = 1, i = 1 - c (:37-41)
    sc[lane] = r;
  }
}

// Row i of the palm transform P = Trans * (Rz * (Ry * Rx)) (:148-152) as a row chain: e_i * Rz, * Ry, * Rx.
struct PalmRow { float4 a, b, r; };   // after Rz, Ry, Rx (r.w = the translation)
__device__ __forceinline__ PalmRow palm_row(int i, const Rot rx, const Rot ry, const Rot rz, float t) {
  PalmRow o;
  o.a = i == 0 ? make_float4(rz.c, -rz.s, 0.f, 0.f) : i == 1 ? make_float4(rz.s, rz.c, 0.f, 0.f) : make_float4(0.f, 0.f, rz.d, 0.f);
  o.b = rot_y(o.a, ry);
  o.r = rot_x(o.b, rx);
  o.r.w = t;
  return o;
}

// The (I, O) rows of a finger's three bones, one lane's copy (the lanes of a finger read the same addresses)
struct BoneConst { float4 i0, i1, i2, o0, o1, o2; };
__device__ __forceinline__ BoneConst load_bone(const float *__restrict__ offset, const float *__restrict__ offset_inv, int nb) {
  BoneConst k;
  k.i0 = ld4(offset_inv + 16 * nb);     k.i1 = ld4(offset_inv + 16 * nb + 4); k.i2 = ld4(offset_inv + 16 * nb + 8);
  k.o0 = ld4(offset + 16 * nb);         k.o1 = ld4(offset + 16 * nb + 4);     k.o2 = ld4(offset + 16 * nb + 8);
  return k;
}

// SYNTH (shr_synth_pose_fwd: the head of HandSynthesizer.forward, network/util_modules.py:104-110): the sample's random
// draws are made HERE from the counter-based generator of common.h -- lanes 0..2 RandScale's three factors
// (mesh/pointTransformation.py:128-148: rand * rand_scale + 0.90 - rand_scale / 2), lane 3 the focal jitter
// (rand * 0.2 + 0.9, util_modules.py:110), lanes 4, 5 the two keys of the sample's pixel-noise stream -- written to
// draws[6][B] and the bone transforms leave as diag(s) * T, i.e. row i times s_i: the product RandScale.forward forms.
struct SynthDraws {
  const unsigned long long *state;   // [2]: seed, call counter (read only here; the render launch advances the counter)
  float rand_scale, rand_half;       // RandScale's width and (float)(rand_scale / 2)
  float *draws;                      // [6][B]
  int B;
};

// The sample's bone transforms by ONE wave, as pose_fwd_kernel<true, false, SYNTH> forms them (same operations, same bits):
// 17 x 4 float4 rows to Tb (LDS or global), SYNTH: scaled by RandScale's draws, which go to syn.draws and -- the focal
// jitter and the noise keys -- to s_scale[3 .. 5].  sc [kAngles] and s_scale [8]: this wave's LDS.  Wave-level
// synchronisation only.
template <bool SYNTH>
__device__ __forceinline__ void pose_transforms_wave(const float *__restrict__ params, const float *__restrict__ offset,
                                                     const float *__restrict__ offset_inv, int b, int lane, Rot *sc,
                                                     float *s_scale, float4 *Tb, const SynthDraws syn) {
  const int g = lane >> 2, i = lane & 3;
  const float *p = params + (size_t)b * 26;
  const bool chain = lane < 24 && i < 3, finger = chain && g < 5;
  const int b0 = 2 + 3 * (finger ? g : 0);
  BoneConst k1, k2, k3;
  if (finger) {
    k1 = load_bone(offset, offset_inv, b0);
    k2 = load_bone(offset, offset_inv, b0 + 1);
    k3 = load_bone(offset, offset_inv, b0 + 2);
  }
  const float t = chain ? p[3 + i] : 0.f;
  if (SYNTH && lane < 6) {
    const uint32_t h = rng_key(syn.state[0], syn.state[1], (uint32_t)b, (uint32_t)lane);
    float val = __uint_as_float(h);
    if (lane < 3) val = (rng_uniform(h) * syn.rand_scale + 0.90f) - syn.rand_half;
    if (lane == 3) val = rng_uniform(h) * 0.2f + 0.9f;
    syn.draws[(size_t)lane * syn.B + b] = val;
    s_scale[lane] = val;                             // (slots 3 .. 5: the focal jitter and the noise keys, for the caller)
  }
  sincos_phase(p, lane, sc);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const float si = (SYNTH && chain) ? s_scale[i] : 1.0f;
  auto scaled = [&](const float4 r) { return SYNTH ? make_float4(r.x * si, r.y * si, r.z * si, r.w * si) : r; };
  if (chain) {
    const PalmRow P = palm_row(i, sc[0], sc[1], sc[2], t);
    if (!finger) {
      Tb[i] = Tb[4 + i] = scaled(P.r);
    } else {
      const int a0 = 3 + 4 * g;
      const bool yaxis = g == 2 || g == 3;
      float4 u = row_times(P.r, k1.i0, k1.i1, k1.i2);
      u = swap_yz(rot_z(swap_yz(u, yaxis), sc[a0]), yaxis);
      u = rot_x(u, sc[a0 + 1]);
      const float4 G1 = row_times(u, k1.o0, k1.o1, k1.o2);
      u = rot_x(row_times(G1, k2.i0, k2.i1, k2.i2), sc[a0 + 2]);
      const float4 G2 = row_times(u, k2.o0, k2.o1, k2.o2);
      u = rot_x(row_times(G2, k3.i0, k3.i1, k3.i2), sc[a0 + 3]);
      const float4 G3 = row_times(u, k3.o0, k3.o1, k3.o2);
      Tb[4 * b0 + i] = scaled(G1); Tb[4 * b0 + 4 + i] = scaled(G2); Tb[4 * b0 + 8 + i] = scaled(G3);
    }
  } else if (lane < 24) {
    const float4 h = make_float4(0.f, 0.f, 0.f, 1.f);
    if (g == 5) {
      Tb[3] = Tb[7] = h;
    } else {
      const int q = 2 + 3 * g;
      Tb[4 * q + 3] = Tb[4 * q + 7] = Tb[4 * q + 11] = h;
    }
  }
}

}  // namespace shr
