#include "../spherehand_amd/csrc/common.h"
namespace shr {
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

// A face set up like the reference kernel (.cu:33-69): culled, vertices sorted by x, pixel
// box.  `live` = front-facing, non-degenerate and its box meets the image.
struct FaceSetup {
  float p[3][3];
  int xi_min, xi_max, r_lo, r_hi;
  bool live;
};

__device__ __forceinline__ FaceSetup face_setup_sorted(const float4 *__restrict__ verts, const int *__restrict__ faces,
                                                       int f, int src) {
  FaceSetup s;
  s.live = false;
  float fv[9];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float4 v = verts[faces[f * 3 + k]];
    fv[3 * k] = v.x; fv[3 * k + 1] = v.y; fv[3 * k + 2] = v.z;
  }
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

constexpr int kMeshQueue = 3584;   // work items per round (14 KB next to the 66-KB slot array: two workgroups per CU)
constexpr int kMeshFaces = 4;      // faces per thread and round (one round for the 3382-face hand mesh)

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
template <int TO, int SL>
__global__ void __launch_bounds__(1024)
exp_mesh(const float4 *__restrict__ vertices, const int *__restrict__ faces, int NV, int F, int src,
                  int S, float clamp_max, float *__restrict__ depth, long long *tb) {
  long long tA = 0, tS = 0, tB = 0, tW = 0; int nitems = 0; const long long T0 = clock64();
  __shared__ uint32_t s_z[SL * TO][SL * TO + 1];   // [SL*dy + sy][SL*dx + sx], +1: bank spread
  __shared__ int s_queue[kMeshQueue];
  __shared__ int s_wave_cnt[16];
  const int b = blockIdx.y;
  const int tiles = (S + TO - 1) / TO;
  const int ty0 = (blockIdx.x / tiles) * TO, tx0 = (blockIdx.x % tiles) * TO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float scale = (float)src / (float)S;
  const float4 *verts = vertices + (size_t)b * NV;

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

  for (int f0 = 0; f0 < F; f0 += 1024 * kMeshFaces) {
    // ---- A. cull, count work items, block scan ---------------------------------------------
    const long long a0 = clock64();
    int nk[kMeshFaces], dx0[kMeshFaces];
    unsigned colmask[kMeshFaces];   // bit i: output column dx0 + i holds a sampled source column of the face
    int n = 0;
#pragma unroll
    for (int k = 0; k < kMeshFaces; k++) {
      const int f = f0 + k * 1024 + tid;
      nk[k] = 0; dx0[k] = 0; colmask[k] = 0u;
      if (f < F) {
        const FaceSetup fs = face_setup_sorted(verts, faces, f, src);
        if (fs.live && !(fs.xi_max < tlx.i0 || fs.xi_min > thx.i1 || fs.r_hi < tly.i0 || fs.r_lo > thy.i1)) {
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
      n += nk[k];
    }
    const long long a1 = clock64(); tA += a1 - a0;
    int incl = n;   // inclusive scan over the workgroup: DPP inside rows of 16, SGPR row totals, LDS wave totals
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
    __syncthreads();   // the previous round's queue is consumed, s_wave_cnt is free
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    int off = incl - n, total = 0;
    for (int w = 0; w < 16; w++) {
      const int c = s_wave_cnt[w];
      if (w < wave) off += c;
      total += c;
    }
    tS += clock64() - a1; nitems += total;
    for (int w0 = 0; w0 < total; w0 += kMeshQueue) {
      if (w0 > 0) __syncthreads();
      {
        int slot = off - w0;
#pragma unroll
        for (int k = 0; k < kMeshFaces; k++) {
          if (nk[k] == 0) continue;
          const int f = f0 + k * 1024 + tid;
          for (unsigned m = colmask[k]; m; m &= m - 1, slot++)
            if (slot >= 0 && slot < kMeshQueue) s_queue[slot] = (f << 7) | (dx0[k] + __builtin_ctz(m) - tx0);
          if (nk[k] > __builtin_popcount(colmask[k])) {   // a face wider than 32 output columns: the rest by re-enumeration
            const FaceSetup fs = face_setup_sorted(verts, faces, f, src);
            const int dx_hi = out_hi(fs.xi_max, tx0);
            for (int dx = dx0[k] + 32; dx <= dx_hi; dx++)
              if (column_slots(dx, fs.xi_min, fs.xi_max)) {
                if (slot >= 0 && slot < kMeshQueue) s_queue[slot] = (f << 7) | (dx - tx0);
                slot++;
              }
          }
        }
      }
      __syncthreads();
      // ---- B. rasterize the queued items (order is irrelevant: integer minima) ---------------
      const long long b0 = clock64();
      const int count = min(kMeshQueue, total - w0);
      for (int q = tid; q < count; q += blockDim.x) {
        const int item = s_queue[q];
        const FaceSetup fs = face_setup_sorted(verts, faces, item >> 7, src);
        const float (&p)[3][3] = fs.p;
        const int xi_min = fs.xi_min, xi_max = fs.xi_max;
        float fi[9];
        fi[0] = p[1][1] - p[2][1]; fi[1] = p[2][0] - p[1][0]; fi[2] = p[1][0] * p[2][1] - p[2][0] * p[1][1];
        fi[3] = p[2][1] - p[0][1]; fi[4] = p[0][0] - p[2][0]; fi[5] = p[2][0] * p[0][1] - p[0][0] * p[2][1];
        fi[6] = p[0][1] - p[1][1]; fi[7] = p[1][0] - p[0][0]; fi[8] = p[0][0] * p[1][1] - p[1][0] * p[0][1];
        const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1])) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
        for (int k = 0; k < 9; k++) fi[k] = fi[k] / den;
        const int dy_lo = out_lo(fs.r_lo, ty0), dy_hi = out_hi(fs.r_hi, ty0);
        {
          const int dx = tx0 + (item & 127);
          const Lin lx = lin_index(dx, scale, src);
#pragma unroll
          for (int sx = 0; sx < SL; sx++) {
            const int xi = sx ? lx.i1 : lx.i0;
            if ((sx ? lx.l1 : lx.l0) == 0.f || xi < xi_min || xi > xi_max) continue;
            // ---- column span (.cu:72-90) -------------------------------------------------
            const float xf = (float)xi;
            float yi1;
            if (xf <= p[1][0]) {
              if (p[1][0] - p[0][0] != 0.f) yi1 = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (xf - p[0][0]) + p[0][1];
              else yi1 = p[1][1];
            } else {
              if (p[2][0] - p[1][0] != 0.f) yi1 = (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]) * (xf - p[1][0]) + p[1][1];
              else yi1 = p[1][1];
            }
            const float yi2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (xf - p[0][0]) + p[0][1];
            const int yi_min = m_cvt_rz_sat(fmaxf(0.f, ceilf(fminf(yi1, yi2))));
            const int yi_max = m_cvt_rz_sat(fminf(fmaxf(yi1, yi2), (float)src - 1.f));
            // (only the output rows whose slots can fall inside the column's span)
            const int ry_lo = yi_min > yi_max ? 1 : max(dy_lo, out_lo(yi_min, ty0));
            const int ry_hi = yi_min > yi_max ? 0 : min(dy_hi, out_hi(yi_max, ty0));
            for (int dy = ry_lo; dy <= ry_hi; dy++) {
              const Lin ly = lin_index(dy, scale, src);
#pragma unroll
              for (int sy = 0; sy < SL; sy++) {
                const int yi = sy ? ly.i1 : ly.i0;
                if ((sy ? ly.l1 : ly.l0) == 0.f || yi < yi_min || yi > yi_max) continue;
                // ---- pixel (.cu:97-110) ----------------------------------------------------
                const float yf = (float)yi;
                float w[3], w_sum = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                  w[k] = (fi[3 * k] * xf + fi[3 * k + 1] * yf) + fi[3 * k + 2];
                  w[k] = fminf(fmaxf(w[k], 0.f), 1.f);
                  w_sum += w[k];
                }
#pragma unroll
                for (int k = 0; k < 3; k++) w[k] = w[k] / w_sum;
                const float zp = 1.0f / ((w[0] / p[0][2] + w[1] / p[1][2]) + w[2] / p[2][2]);
                if (zp == zp) atomicMin(&s_z[SL * (dy - ty0) + sy][SL * (dx - tx0) + sx], mkey(zp));
              }
            }
          }
        }
      }
      const long long b1 = clock64(); tB += b1 - b0;
    }
  }
  const long long e0 = clock64();
  __syncthreads();

  // ---- clamp + bilinear (mesh/render.py:286, :311; ATen upsample_bilinear2d) ---------------
  float *out = depth + (size_t)b * S * S;
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
  if (lane == 0) {
    long long *t = tb + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 8;
    t[0] = tA; t[1] = tS; t[2] = tB; t[3] = e0 - T0; t[4] = clock64() - T0; t[5] = nitems;
  }
}
}
extern "C" int exp_mesh_launch(const float *vertices, const int *faces, int B, int NV, int F, int src, int S, float *depth,
                               long long *tb, void *stream) {
  using namespace shr;
  if (S == 128) hipLaunchKernelGGL((exp_mesh<128, 1>), dim3(1, B), dim3(1024), 0, (hipStream_t)stream, (const float4 *)vertices, faces, NV, F, src, S, 100.f, depth, tb);
  else hipLaunchKernelGGL((exp_mesh<64, 2>), dim3(((S + 63) / 64) * ((S + 63) / 64), B), dim3(1024), 0, (hipStream_t)stream, (const float4 *)vertices, faces, NV, F, src, S, 100.f, depth, tb);
  return (int)hipGetLastError();
}
