// fk.hip -- forward kinematics of the hand (pose -> bone transforms -> sphere records) and the analytic backward.
//
// Replaces (reference file:line): mesh/kinematicsTransformation.py:157-177 HandTransformationMat.forward (Palm
// :145-155, Finger :123-127, FingerJoint :92-112, AxisRotationMatrix :29-54) -- ~100 tiny launches in the reference --
// the autograd backward of that graph and, in the fused entries, the key-point skinning + cat of
// HandBallPrimitiveRender (mesh/render.py:65-88; keypoint_skin.hip has the stand-alone kernels and the formulae).
//
//   params[B,26] = palm Euler x,y,z | palm translation | 5 x (abduct, flex1..3)
//   T[B,17,4,4]  : bones 0,1 = palm P = Trans * Rz * Ry * Rx ; finger g (bones 2+3g..4+3g): G1 = P*A1, G2 = G1*A2,
//                  G3 = G2*A3 with A_k = offset_k^-1 * L_k * offset_k, L1 = R_abduct(a0)*R_x(a1), L2 = R_x(a2),
//                  L3 = R_x(a3).  Every factor is affine (last row 0 0 0 1: the bones' offset matrices are).
//
// Round 5 rebuild.  Rounds 1-4 gave a sample 8 lanes, five of which each ran ~25 dependent 4x4 products serially,
// 8 workgroups in all: 7.1 us forward, 14.6 us backward for 256 poses -- the longest kernel of the pose -> depth ->
// pose chain was not a rasterizer.  Two observations remove almost all of that work:
//   * row i of a product depends on row i of the left factor only, so a chain G_k = G_k-1 * (I_k L_k O_k) is three
//     INDEPENDENT row chains  g <- ((g * I_k) * L_k) * O_k  -- a 4-vector times an affine matrix is 12 FMAs, times an
//     axis-aligned rotation 5 multiplies/FMAs -- and no 4x4 product is ever formed (28 operations per bone and row
//     where forming A_k and multiplying took 3 x 64);
//   * the same holds in reverse mode: the adjoint of a row chain is a row chain, and an angle's gradient is the
//     2-term expression  <out_bar, d out / d theta> = ob_p * out_q - ob_q * out_p  of the rotated pair, summed over
//     the three rows.
// One wave per sample (256 poses = 256 single-wave workgroups, one per CU): lane 4 g + i owns row i of finger g
// (g = 5: the palm), 23 lanes take one sincos each first.  The axis-aligned rotations keep the reference's matrix
// entries, including its diagonal d = (1 - c) + c on the axis (:40-52), so the result is the reference's up to the
// association of the sums (FMAs here; the reference's own bmm order is the library's): <= 1e-4 mm from the
// reference's fp32 T on entries up to 150, 3e-5 from the fp64 value (the reference itself: 1e-4).
#include "fk_rows.h"

namespace shr {

// ---- forward -----------------------------------------------------------------------------------------------
// One wave per sample.  WANT_T: write T[b] (17 x 4 x 4).  WANT_SPH: the key-point records too (T stays in LDS).
template <bool WANT_T, bool WANT_SPH, bool SYNTH = false>
__global__ void __launch_bounds__(64)
pose_fwd_kernel(const float *__restrict__ params, const float *__restrict__ offset, const float *__restrict__ offset_inv,
                float *__restrict__ T, const int *__restrict__ bone, const float4 *__restrict__ wv,
                const float *__restrict__ radii, float sx, int J, float4 *__restrict__ spheres, SynthDraws syn) {
  __shared__ Rot sc[kAngles];
  __shared__ float4 rows[kBones * 3];
  __shared__ float s_scale[4];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (SYNTH && lane < 6) {
    const uint32_t h = rng_key(syn.state[0], syn.state[1], (uint32_t)b, (uint32_t)lane);
    float val = __uint_as_float(h);                                                   // lanes 4, 5: stream keys, as bits
    if (lane < 3) val = (rng_uniform(h) * syn.rand_scale + 0.90f) - syn.rand_half;
    if (lane == 3) val = rng_uniform(h) * 0.2f + 0.9f;
    syn.draws[(size_t)lane * syn.B + b] = val;
    if (lane < 3) s_scale[lane] = val;
  }
  const int g = lane >> 2, i = lane & 3;
  const float *p = params + (size_t)b * 26;
  const bool chain = lane < 24 && i < 3, finger = chain && g < 5;
  // everything the chain reads, requested before the sincos
  const int b0 = 2 + 3 * (finger ? g : 0);
  BoneConst k1, k2, k3;
  if (finger) {
    k1 = load_bone(offset, offset_inv, b0);
    k2 = load_bone(offset, offset_inv, b0 + 1);
    k3 = load_bone(offset, offset_inv, b0 + 2);
  }
  const float t = chain ? p[3 + i] : 0.f;
  int kb = 0;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  float rad = 0.f;
  if (WANT_SPH && lane < J) {
    kb = bone[lane];
    v = wv[lane];
    rad = radii[lane];
  }
  sincos_phase(p, lane, sc);
  __syncthreads();
  float4 *Tb = reinterpret_cast<float4 *>(T + (size_t)b * kBones * 16);
  // SYNTH: every entry of row i times s_i -- one fp32 multiply, the module's `transform_mats * diag` (T only)
  const float si = (SYNTH && chain) ? s_scale[i] : 1.0f;
  auto scaled = [&](const float4 r) { return SYNTH ? make_float4(r.x * si, r.y * si, r.z * si, r.w * si) : r; };
  if (chain) {
    const PalmRow P = palm_row(i, sc[0], sc[1], sc[2], t);
    if (!finger) {                     // bones 0 and 1 both carry the palm transform (:153-155)
      if (WANT_SPH) rows[i] = rows[3 + i] = P.r;
      if (WANT_T) Tb[i] = Tb[4 + i] = scaled(P.r);
    } else {
      const int a0 = 3 + 4 * g;        // sincos slot of the finger's first angle (parameter 6 + 4 g)
      const bool yaxis = g == 2 || g == 3;
      float4 u = row_times(P.r, k1.i0, k1.i1, k1.i2);                       // :108-111, row by row
      u = swap_yz(rot_z(swap_yz(u, yaxis), sc[a0]), yaxis);
      u = rot_x(u, sc[a0 + 1]);
      const float4 G1 = row_times(u, k1.o0, k1.o1, k1.o2);
      u = rot_x(row_times(G1, k2.i0, k2.i1, k2.i2), sc[a0 + 2]);
      const float4 G2 = row_times(u, k2.o0, k2.o1, k2.o2);
      u = rot_x(row_times(G2, k3.i0, k3.i1, k3.i2), sc[a0 + 3]);
      const float4 G3 = row_times(u, k3.o0, k3.o1, k3.o2);
      if (WANT_SPH) {
        rows[3 * b0 + i] = G1; rows[3 * b0 + 3 + i] = G2; rows[3 * b0 + 6 + i] = G3;
      }
      if (WANT_T) {
        Tb[4 * b0 + i] = scaled(G1); Tb[4 * b0 + 4 + i] = scaled(G2); Tb[4 * b0 + 8 + i] = scaled(G3);
      }
    }
  } else if (WANT_T && lane < 24) {    // i == 3: the homogeneous rows of the group's bones
    const float4 h = make_float4(0.f, 0.f, 0.f, 1.f);
    if (g == 5) {
      Tb[3] = Tb[7] = h;
    } else {
      const int q = 2 + 3 * g;
      Tb[4 * q + 3] = Tb[4 * q + 7] = Tb[4 * q + 11] = h;
    }
  }
  if (!WANT_SPH) return;
  __syncthreads();
  // keypoint_spheres_fwd_kernel's arithmetic on the rows in LDS: the same records, bit for bit, as the two launches
  for (int j = lane; j < J; j += 64) {
    if (j >= 64) {
      kb = bone[j];
      v = wv[j];
      rad = radii[j];
    }
    kb = min(max(kb, 0), kBones - 1);            // (a bone number outside the hand's 17 reads a row of the table, not beyond it)
    const float4 r0 = rows[3 * kb], r1 = rows[3 * kb + 1], r2 = rows[3 * kb + 2];
    const float x = ((r0.x * v.x + r0.y * v.y) + r0.z * v.z) + r0.w * v.w;
    const float y = ((r1.x * v.x + r1.y * v.y) + r1.z * v.z) + r1.w * v.w;
    const float z = ((r2.x * v.x + r2.y * v.y) + r2.z * v.z) + r2.w * v.w;
    spheres[(size_t)b * J + j] = make_float4(sx * x, y, z, rad);
  }
}

// ---- backward ----------------------------------------------------------------------------------------------
// grad_params[b] from grad_T[b] (FROM_SPH = false) or from grad_spheres[b] (true: keypoint_spheres_bwd_kernel's sums
// are formed in LDS first, same order, same bits).
template <bool FROM_SPH>
__global__ void __launch_bounds__(64)
pose_bwd_kernel(const float *__restrict__ params, const float *__restrict__ offset, const float *__restrict__ offset_inv,
                const float *__restrict__ grad_T, const float4 *__restrict__ grad_spheres,
                const int *__restrict__ bone_start, const int *__restrict__ bone_points, const float4 *__restrict__ wv,
                float sx, int J, float *__restrict__ grad_params) {
  __shared__ Rot sc[kAngles];
  __shared__ float4 gT[kBones * 3];    // d loss / d T[b, bone, row < 3, :]
  __shared__ float4 pbar[6 * 3];       // the six groups' contributions to d loss / d P
  __shared__ float4 sg[FROM_SPH ? kMaxPoints : 1], sw[FROM_SPH ? kMaxPoints : 1];
  __shared__ int sj[FROM_SPH ? kMaxPoints : 1];
  const int b = blockIdx.x, lane = threadIdx.x;
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
  if (FROM_SPH) {
    // Stage the sample's gradient records (as they lie: one coalesced load) and the key-points in bone order
    // (entry k = point bone_points[k]) in LDS, in flight while the sincos run.  (Walking the CSR lists straight from
    // HBM was a chain of dependent loads per point -- 11 for the palm -- and 3 us of this kernel's 6.2.)
    // (point numbers clamped into [0, J): a malformed table can then neither index wv[] nor the LDS copies out of range)
    for (int k = lane; k < J && k < kMaxPoints; k += 64) {
      const int j = min(max(bone_points[k], 0), min(J, kMaxPoints) - 1);
      sg[k] = grad_spheres[(size_t)b * J + k];
      sj[k] = j;
      sw[k] = wv[j];
    }
  }
  int ks = 0, ke = 0;
  float4 gdirect = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nb = lane / 3, r = lane - 3 * nb;   // lane = 3 * bone + row
  if (lane < kBones * 3) {
    if (FROM_SPH) {
      ks = bone_start[nb];
      ke = bone_start[nb + 1];
    } else {
      gdirect = ld4(grad_T + ((size_t)b * kBones + nb) * 16 + 4 * r);
    }
  }
  sincos_phase(p, lane, sc);
  if (FROM_SPH) __syncthreads();
  if (lane < kBones * 3) {
    if (FROM_SPH) {
      // keypoint_spheres_bwd_kernel's sums, point by point in index order (the same bits); four entries are read
      // ahead and the ones past the bone's end are skipped by predicate
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = ks; k < ke; k += 4) {
        float4 gs[4], w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int kk = min(max(k + q, 0), min(J, kMaxPoints) - 1);   // (entries past J are uninitialised LDS: never read)
          gs[q] = sg[sj[kk]];
          w[q] = sw[kk];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float gr = r == 0 ? sx * gs[q].x : r == 1 ? gs[q].y : gs[q].z;
          const bool on = k + q < ke;
          const float nx = a.x + gr * w[q].x, ny = a.y + gr * w[q].y, nz = a.z + gr * w[q].z, nw = a.w + gr * w[q].w;
          a = on ? make_float4(nx, ny, nz, nw) : a;
        }
      }
      gT[lane] = a;
    } else {
      gT[lane] = gdirect;
    }
  }
  __syncthreads();
  PalmRow P;
  float ga0 = 0.f, ga1 = 0.f, ga2 = 0.f, ga3 = 0.f;
  if (chain) {
    P = palm_row(i, sc[0], sc[1], sc[2], t);
    if (!finger) {
      const float4 x = gT[i], y = gT[3 + i];
      pbar[15 + i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    } else {
      const int a0 = 3 + 4 * g;
      const bool yaxis = g == 2 || g == 3;
      const Rot ra = sc[a0], rb = sc[a0 + 1], r2 = sc[a0 + 2], r3 = sc[a0 + 3];
      // forward, keeping what the adjoints read
      const float4 w1 = swap_yz(rot_z(swap_yz(row_times(P.r, k1.i0, k1.i1, k1.i2), yaxis), ra), yaxis);
      const float4 v1 = rot_x(w1, rb);
      const float4 G1 = row_times(v1, k1.o0, k1.o1, k1.o2);
      const float4 v2 = rot_x(row_times(G1, k2.i0, k2.i1, k2.i2), r2);
      const float4 G2 = row_times(v2, k2.o0, k2.o1, k2.o2);
      const float4 v3 = rot_x(row_times(G2, k3.i0, k3.i1, k3.i2), r3);
      // reverse
      float4 Gb = gT[3 * b0 + 6 + i];
      float4 vb = row_times_adj(Gb, k3.o0, k3.o1, k3.o2);
      ga3 = rot_x_dangle(vb, v3);
      float4 ub = row_times_adj(rot_x_adj(vb, r3), k3.i0, k3.i1, k3.i2);
      Gb = gT[3 * b0 + 3 + i];
      Gb = make_float4(Gb.x + ub.x, Gb.y + ub.y, Gb.z + ub.z, Gb.w + ub.w);
      vb = row_times_adj(Gb, k2.o0, k2.o1, k2.o2);
      ga2 = rot_x_dangle(vb, v2);
      ub = row_times_adj(rot_x_adj(vb, r2), k2.i0, k2.i1, k2.i2);
      Gb = gT[3 * b0 + i];
      Gb = make_float4(Gb.x + ub.x, Gb.y + ub.y, Gb.z + ub.z, Gb.w + ub.w);
      vb = row_times_adj(Gb, k1.o0, k1.o1, k1.o2);
      ga1 = rot_x_dangle(vb, v1);
      const float4 wb = swap_yz(rot_x_adj(vb, rb), yaxis);     // adjoint of w1, in the swapped frame
      ga0 = rot_z_dangle(wb, swap_yz(w1, yaxis));
      ub = row_times_adj(swap_yz(rot_z_adj(wb, ra), yaxis), k1.i0, k1.i1, k1.i2);
      pbar[3 * g + i] = ub;
    }
  }
  // an angle's gradient: the sum over the three rows (lanes 4 g + 0..2; lane 4 g + 3 holds zeros)
  ga0 = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(ga0));
  ga1 = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(ga1));
  ga2 = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(ga2));
  ga3 = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(ga3));
  float *gp = grad_params + (size_t)b * 26;
  if (lane < 20 && i == 0) {
    float2 *o = reinterpret_cast<float2 *>(gp + 6 + 4 * g);    // 8-byte aligned: 26 floats per sample
    o[0] = make_float2(ga0, ga1);
    o[1] = make_float2(ga2, ga3);
  }
  __syncthreads();
  // palm: d loss / d P row i = the palm bones' own gradient + the five fingers', in fixed order
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (chain && g == 5) {
    float4 Pb = pbar[15 + i];
#pragma unroll
    for (int f = 0; f < 5; f++) {
      const float4 c = pbar[3 * f + i];
      Pb = make_float4(Pb.x + c.x, Pb.y + c.y, Pb.z + c.z, Pb.w + c.w);
    }
    gp[3 + i] = Pb.w;                                           // P = Trans * R: the translation column
    const Rot rx = sc[0], ry = sc[1], rz = sc[2];
    Pb.w = 0.f;
    gx = rot_x_dangle(Pb, P.r);
    const float4 bb = rot_x_adj(Pb, rx);
    gy = rot_y_dangle(bb, P.b);
    const float4 ab = rot_y_adj(bb, ry);
    // row i of Rz: (c,-s,0) | (s,c,0) | (0,0,d): d/dz = (-s,-c,0) | (c,-s,0) | 0
    gz = i == 0 ? -(ab.x * rz.s) - ab.y * rz.c : i == 1 ? ab.x * rz.c - ab.y * rz.s : 0.f;
  }
  gx = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(gx));
  gy = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(gy));
  gz = dpp_add<0x4E, 0xF>(dpp_add<0xB1, 0xF>(gz));
  if (lane == 20) {
    gp[0] = gx; gp[1] = gy; gp[2] = gz;
  }
}

}  // namespace shr

static int fk_check(const void *params, int B, const void *offset, const void *offset_inv) {
  if (!params || !offset || !offset_inv || B < 0) return SHR_EINVAL;
  if ((((uintptr_t)offset | (uintptr_t)offset_inv) & 15u) != 0) return SHR_EINVAL;
  if (B > (1 << 24)) return SHR_ETOOLARGE;
  return SHR_OK;
}

extern "C" int shr_fk_fwd(const float *params, int B, const float *offset, const float *offset_inv, float *T,
                          void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!T || (((uintptr_t)T) & 15u) != 0) return SHR_EINVAL;
  if (int rc = fk_check(params, B, offset, offset_inv)) return rc;
  hipLaunchKernelGGL((pose_fwd_kernel<true, false>), dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, params, offset,
                     offset_inv, T, nullptr, nullptr, nullptr, 1.0f, 0, nullptr, SynthDraws{});
  return (int)hipGetLastError();
}

/* The head of HandSynthesizer.forward (network/util_modules.py:104-110) in one launch: forward kinematics, RandScale
 * and the sample's random draws (kernel comment above; generator: common.h). */
extern "C" int shr_synth_pose_fwd(const float *params, int B, const float *offset, const float *offset_inv,
                                  const unsigned long long *rng_state, float rand_scale, float *T, float *draws,
                                  void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!T || !draws || !rng_state || (((uintptr_t)T) & 15u) != 0 || (((uintptr_t)rng_state) & 7u) != 0) return SHR_EINVAL;
  if (int rc = fk_check(params, B, offset, offset_inv)) return rc;
  SynthDraws syn;
  syn.state = rng_state; syn.rand_scale = rand_scale; syn.rand_half = (float)((double)rand_scale / 2.0); syn.draws = draws; syn.B = B;
  hipLaunchKernelGGL((pose_fwd_kernel<true, false, true>), dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, params, offset,
                     offset_inv, T, nullptr, nullptr, nullptr, 1.0f, 0, nullptr, syn);
  return (int)hipGetLastError();
}

extern "C" int shr_fk_bwd(const float *params, int B, const float *offset, const float *offset_inv,
                          const float *grad_T, float *grad_params, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!grad_T || !grad_params || (((uintptr_t)grad_T) & 15u) != 0 || (((uintptr_t)grad_params) & 7u) != 0) return SHR_EINVAL;
  if (int rc = fk_check(params, B, offset, offset_inv)) return rc;
  hipLaunchKernelGGL((pose_bwd_kernel<false>), dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, params, offset,
                     offset_inv, grad_T, nullptr, nullptr, nullptr, nullptr, 1.0f, 0, grad_params);
  return (int)hipGetLastError();
}

extern "C" int shr_pose_spheres_fwd(const float *params, int B, const float *offset, const float *offset_inv, int J,
                                    const int32_t *bone, const float *wv, const float *radii, int right_hand,
                                    float *spheres, float *T, void *stream) {
  using namespace shr;
  if (B == 0 || J == 0) return SHR_OK;
  if (!bone || !wv || !radii || !spheres || J < 0) return SHR_EINVAL;
  if ((((uintptr_t)wv | (uintptr_t)spheres | (uintptr_t)T) & 15u) != 0) return SHR_EINVAL;
  if (int rc = fk_check(params, B, offset, offset_inv)) return rc;
  if ((long long)B * J > (1LL << 30)) return SHR_ETOOLARGE;
  const float sx = right_hand ? -1.0f : 1.0f;
  if (T)
    hipLaunchKernelGGL((pose_fwd_kernel<true, true>), dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, params, offset,
                       offset_inv, T, bone, reinterpret_cast<const float4 *>(wv), radii, sx, J,
                       reinterpret_cast<float4 *>(spheres), SynthDraws{});
  else
    hipLaunchKernelGGL((pose_fwd_kernel<false, true>), dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, params, offset,
                       offset_inv, nullptr, bone, reinterpret_cast<const float4 *>(wv), radii, sx, J,
                       reinterpret_cast<float4 *>(spheres), SynthDraws{});
  return (int)hipGetLastError();
}

extern "C" int shr_pose_spheres_bwd(const float *params, int B, const float *offset, const float *offset_inv, int J,
                                    const int32_t *bone_start, const int32_t *bone_points, const float *wv,
                                    int right_hand, const float *grad_spheres, float *grad_params, void *stream) {
  using namespace shr;
  if (B == 0) return SHR_OK;
  if (!bone_start || !bone_points || !wv || !grad_spheres || !grad_params || J < 0) return SHR_EINVAL;
  if (J > kMaxPoints) return SHR_ETOOLARGE;
  if ((((uintptr_t)wv | (uintptr_t)grad_spheres) & 15u) != 0 || (((uintptr_t)grad_params) & 7u) != 0) return SHR_EINVAL;
  if (int rc = fk_check(params, B, offset, offset_inv)) return rc;
  if ((long long)B * J > (1LL << 30)) return SHR_ETOOLARGE;
  hipLaunchKernelGGL((pose_bwd_kernel<true>), dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, params, offset,
                     offset_inv, nullptr, reinterpret_cast<const float4 *>(grad_spheres), bone_start, bone_points,
                     reinterpret_cast<const float4 *>(wv), right_hand ? -1.0f : 1.0f, J, grad_params);
  return (int)hipGetLastError();
}
