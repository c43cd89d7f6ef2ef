// fk.hip -- forward kinematics of the hand, forward + analytic backward.
//
// Replaces (reference file:line): mesh/kinematicsTransformation.py:157-177
// HandTransformationMat.forward (Palm :145-155, Finger :123-127, FingerJoint
// :92-112, AxisRotationMatrix :29-54) -- ~100 tiny launches in the reference --
// and the autograd backward of that graph.
//
//   params[B,26] = palm Euler x,y,z | palm translation | 5 x (abduct, flex1..3)
//   T[B,17,4,4]  : bones 0,1 = palm P = Trans * Rz * Ry * Rx ; finger f (bones
//                  2+3f..4+3f): G1 = P*A1, G2 = G1*A2, G3 = G2*A3 with
//                  A_k = offset_k^-1 * L_k * offset_k, L1 = R_abduct(a0)*R_x(a1),
//                  L2 = R_x(a2), L3 = R_x(a3).
//
// 8 lanes per sample: lanes 0-4 own a finger (and recompute the palm), lane 5 adds
// the palm bones' gradient; the palm gradient is summed over the 8 lanes by an
// xor butterfly.  Reverse mode: dL/dG3 -> A3, G2 ; ... ; every rotation's angle
// gradient is <dL/dR, dR/dtheta>.
#include "common.h"

namespace shr {

struct M4 { float m[16]; };

__device__ __forceinline__ M4 mul(const M4 &a, const M4 &b) {
  M4 c;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) s += a.m[4 * i + k] * b.m[4 * k + j];
      c.m[4 * i + j] = s;
    }
  return c;
}
__device__ __forceinline__ M4 mul_bt(const M4 &a, const M4 &b) {  // a * b^T
  M4 c;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) s += a.m[4 * i + k] * b.m[4 * j + k];
      c.m[4 * i + j] = s;
    }
  return c;
}
__device__ __forceinline__ M4 mul_at(const M4 &a, const M4 &b) {  // a^T * b
  M4 c;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) s += a.m[4 * k + i] * b.m[4 * k + j];
      c.m[4 * i + j] = s;
    }
  return c;
}
__device__ __forceinline__ M4 load4(const float *p) {
  M4 r;
#pragma unroll
  for (int i = 0; i < 16; i++) r.m[i] = p[i];
  return r;
}
__device__ __forceinline__ void store4(float *p, const M4 &a) {
#pragma unroll
  for (int i = 0; i < 16; i++) p[i] = a.m[i];
}

// mesh/kinematicsTransformation.py:29-54 for a unit axis (x, y, z)
__device__ __forceinline__ M4 axis_rot(float x, float y, float z, float angle) {
  const float c = cosf(angle), s = sinf(angle), i = 1.0f - c;
  M4 r;
#pragma unroll
  for (int k = 0; k < 16; k++) r.m[k] = 0.f;
  r.m[15] = 1.0f;
  r.m[0] = (x * x) * i + c;     r.m[1] = (x * y) * i - z * s; r.m[2] = (x * z) * i + y * s;
  r.m[4] = (x * y) * i + z * s; r.m[5] = (y * y) * i + c;     r.m[6] = (y * z) * i - x * s;
  r.m[8] = (x * z) * i - y * s; r.m[9] = (y * z) * i + x * s; r.m[10] = (z * z) * i + c;
  return r;
}
// <G, dR/dangle> over the 3x3 block
__device__ __forceinline__ float axis_rot_grad(float x, float y, float z, float angle, const M4 &g) {
  const float c = cosf(angle), s = sinf(angle);
  // dR = axis axis^T * s - s I + c [axis]x
  float d[9] = {(x * x) * s - s, (x * y) * s - z * c, (x * z) * s + y * c,
                (x * y) * s + z * c, (y * y) * s - s, (y * z) * s - x * c,
                (x * z) * s - y * c, (y * z) * s + x * c, (z * z) * s - s};
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) acc += g.m[4 * r + k] * d[3 * r + k];
  return acc;
}

__device__ __forceinline__ void abduct_axis(int f, float &x, float &y, float &z) {  // :162-164
  x = 0.f;
  y = (f == 2 || f == 3) ? -1.f : 0.f;
  z = (f == 2 || f == 3) ? 0.f : 1.f;
}

__device__ __forceinline__ M4 palm_matrix(const float *p, M4 &Rx, M4 &Ry, M4 &Rz) {
  Rx = axis_rot(1.f, 0.f, 0.f, p[0]);
  Ry = axis_rot(0.f, 1.f, 0.f, p[1]);
  Rz = axis_rot(0.f, 0.f, 1.f, p[2]);
  M4 R = mul(Rz, mul(Ry, Rx));       // :148-150
  M4 Tr;
#pragma unroll
  for (int k = 0; k < 16; k++) Tr.m[k] = (k % 5 == 0) ? 1.f : 0.f;
  Tr.m[3] = p[3]; Tr.m[7] = p[4]; Tr.m[11] = p[5];
  return mul(Tr, R);                 // :152
}

__global__ void __launch_bounds__(256)
fk_fwd_kernel(const float *__restrict__ params, int B, const float *__restrict__ offset,
              const float *__restrict__ offset_inv, float *__restrict__ T) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, f = gid & 7;
  if (b >= B || f > 5) return;
  const float *p = params + (size_t)b * 26;
  float *Tb = T + (size_t)b * 17 * 16;
  M4 Rx, Ry, Rz;
  const M4 P = palm_matrix(p, Rx, Ry, Rz);
  if (f == 5) {  // bones 0 and 1 both carry the palm transform (:153-155)
    store4(Tb, P);
    store4(Tb + 16, P);
    return;
  }
  const float *a = p + 6 + 4 * f;
  const int b0 = 2 + 3 * f;
  float ax, ay, az;
  abduct_axis(f, ax, ay, az);
  const M4 L1 = mul(axis_rot(ax, ay, az, a[0]), axis_rot(1.f, 0.f, 0.f, a[1]));
  M4 G = mul(P, mul(mul(load4(offset_inv + 16 * b0), L1), load4(offset + 16 * b0)));   // :108-111
  store4(Tb + 16 * b0, G);
  G = mul(G, mul(mul(load4(offset_inv + 16 * (b0 + 1)), axis_rot(1.f, 0.f, 0.f, a[2])), load4(offset + 16 * (b0 + 1))));
  store4(Tb + 16 * (b0 + 1), G);
  G = mul(G, mul(mul(load4(offset_inv + 16 * (b0 + 2)), axis_rot(1.f, 0.f, 0.f, a[3])), load4(offset + 16 * (b0 + 2))));
  store4(Tb + 16 * (b0 + 2), G);
}

__global__ void __launch_bounds__(256)
fk_bwd_kernel(const float *__restrict__ params, int B, const float *__restrict__ offset,
              const float *__restrict__ offset_inv, const float *__restrict__ grad_T,
              float *__restrict__ grad_params) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, f = gid & 7;
  const bool active = b < B;
  const int bb = active ? b : 0;
  const float *p = params + (size_t)bb * 26;
  const float *gT = grad_T + (size_t)bb * 17 * 16;
  M4 Rx, Ry, Rz;
  const M4 P = palm_matrix(p, Rx, Ry, Rz);
  M4 Pbar;
#pragma unroll
  for (int k = 0; k < 16; k++) Pbar.m[k] = 0.f;
  float ga[4] = {0.f, 0.f, 0.f, 0.f};
  if (active && f == 5) {
#pragma unroll
    for (int k = 0; k < 16; k++) Pbar.m[k] = gT[k] + gT[16 + k];
  } else if (active && f < 5) {
    const float *a = p + 6 + 4 * f;
    const int b0 = 2 + 3 * f;
    float ax, ay, az;
    abduct_axis(f, ax, ay, az);
    const M4 Ra = axis_rot(ax, ay, az, a[0]), Rb = axis_rot(1.f, 0.f, 0.f, a[1]);
    const M4 O1 = load4(offset + 16 * b0), I1 = load4(offset_inv + 16 * b0);
    const M4 O2 = load4(offset + 16 * (b0 + 1)), I2 = load4(offset_inv + 16 * (b0 + 1));
    const M4 O3 = load4(offset + 16 * (b0 + 2)), I3 = load4(offset_inv + 16 * (b0 + 2));
    const M4 A1 = mul(mul(I1, mul(Ra, Rb)), O1);
    const M4 A2 = mul(mul(I2, axis_rot(1.f, 0.f, 0.f, a[2])), O2);
    const M4 A3 = mul(mul(I3, axis_rot(1.f, 0.f, 0.f, a[3])), O3);
    const M4 G1 = mul(P, A1), G2 = mul(G1, A2);
    M4 g3 = load4(gT + 16 * (b0 + 2)), g2 = load4(gT + 16 * (b0 + 1)), g1 = load4(gT + 16 * b0);
    // G3 = G2 A3
    const M4 A3bar = mul_at(G2, g3);
    M4 t = mul_bt(g3, A3);
#pragma unroll
    for (int k = 0; k < 16; k++) g2.m[k] += t.m[k];
    // G2 = G1 A2
    const M4 A2bar = mul_at(G1, g2);
    t = mul_bt(g2, A2);
#pragma unroll
    for (int k = 0; k < 16; k++) g1.m[k] += t.m[k];
    // G1 = P A1
    const M4 A1bar = mul_at(P, g1);
    Pbar = mul_bt(g1, A1);
    // A_k = I_k L_k O_k  ->  Lbar = I^T Abar O^T
    const M4 L3bar = mul_bt(mul_at(I3, A3bar), O3);
    const M4 L2bar = mul_bt(mul_at(I2, A2bar), O2);
    const M4 L1bar = mul_bt(mul_at(I1, A1bar), O1);
    ga[3] = axis_rot_grad(1.f, 0.f, 0.f, a[3], L3bar);
    ga[2] = axis_rot_grad(1.f, 0.f, 0.f, a[2], L2bar);
    // L1 = Ra Rb
    ga[0] = axis_rot_grad(ax, ay, az, a[0], mul_bt(L1bar, Rb));
    ga[1] = axis_rot_grad(1.f, 0.f, 0.f, a[1], mul_at(Ra, L1bar));
  }
  // palm gradient: sum over the sample's 8 lanes
#pragma unroll
  for (int k = 0; k < 16; k++) {
    float v = Pbar.m[k];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    Pbar.m[k] = v;
  }
  if (!active) return;
  float *gp = grad_params + (size_t)b * 26;
  if (f < 5) {
#pragma unroll
    for (int k = 0; k < 4; k++) gp[6 + 4 * f + k] = ga[k];
  }
  if (f == 5) {
    // P = Tr * R, R = Rz Ry Rx: translation gradient = Pbar[:3,3]; Rbar = Pbar[:3,:3]
    // (Tr's rotation block is the identity, so the 3x3 block passes through)
    M4 Rbar = Pbar;
    Rbar.m[3] = Rbar.m[7] = Rbar.m[11] = 0.f;
    Rbar.m[12] = Rbar.m[13] = Rbar.m[14] = Rbar.m[15] = 0.f;
    const M4 Ryx = mul(Ry, Rx);
    const M4 Ryxbar = mul_at(Rz, Rbar);
    gp[2] = axis_rot_grad(0.f, 0.f, 1.f, p[2], mul_bt(Rbar, Ryx));
    gp[1] = axis_rot_grad(0.f, 1.f, 0.f, p[1], mul_bt(Ryxbar, Rx));
    gp[0] = axis_rot_grad(1.f, 0.f, 0.f, p[0], mul_at(Ry, Ryxbar));
    gp[3] = Pbar.m[3]; gp[4] = Pbar.m[7]; gp[5] = Pbar.m[11];
  }
}

}  // namespace shr

extern "C" int shr_fk_fwd(const float *params, int B, const float *offset, const float *offset_inv, float *T,
                          void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!params || !offset || !offset_inv || !T || B < 0) return SHR_EINVAL;
  if (B > (1 << 27)) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(fk_fwd_kernel, dim3((unsigned)((B * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, B,
                     offset, offset_inv, T);
  return (int)hipGetLastError();
}

extern "C" int shr_fk_bwd(const float *params, int B, const float *offset, const float *offset_inv,
                          const float *grad_T, float *grad_params, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!params || !offset || !offset_inv || !grad_T || !grad_params || B < 0) return SHR_EINVAL;
  if (B > (1 << 27)) return SHR_ETOOLARGE;
  hipLaunchKernelGGL(fk_bwd_kernel, dim3((unsigned)((B * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, B,
                     offset, offset_inv, grad_T, grad_params);
  return (int)hipGetLastError();
}
