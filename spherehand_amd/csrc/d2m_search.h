// d2m_search.h -- the data->model SEARCH shared by data_to_model.hip (stand-alone kernel) and the fused
// render-and-compare kernel (sphere_zbuf.h): <= 64 K foreground points against the crop's sphere table.
//
// Replaces (reference file:line): mesh/render.py:133-141 -- per point p = (xg, yg, depth):
//     e = min_j | ||p - c_j||_2 - r_j | ,  clamp(e, 0, 50)
// and the autograd gradient of that line w.r.t. the centres.  ONE implementation: both kernels produce the same
// fixed-point terms for the same point, so their integer sums are bit-identical whatever the grouping.
//
//   * lane l takes the K consecutive entries K l .. K l + K - 1 (neighbouring pixels: mostly one owner): a sphere's
//     record is read once for the 64 K points (uniform-address ds_read_b128 = an LDS broadcast, requested one
//     sphere ahead), 12 VALU instructions per (point, sphere), K independent chains per lane;
//   * the points lie in a thin strip of rows [y_lo, y_hi] (first / last entry: pixel order), and
//     | ||p - c|| - r | >= dist_y(c, strip) - r, so with lanes = spheres one ballot gives the spheres whose y extent
//     meets the strip.  They are searched first; the largest of the running minima (one wave maximum) then bounds
//     what any other sphere would have to beat, and only spheres whose gap is below it are searched as well -- exact
//     (a pruned sphere can neither win nor tie; ties keep the first index, torch.min's convention);
//   * loss and gradient are accumulated as FIXED-POINT integers (2^-20 mm / 2^-26 per unit-vector component, 64-bit
//     LDS atomics, a lane's K points combined first when they share their owner): order-independent sums.
// A non-finite sphere record or depth value sends the search through the exact index-order loop (torch.min / clamp
// propagate NaN).
#pragma once
#include "common.h"

namespace shr {

constexpr int kD2mTables = 4;               // copies of the gradient table (lane & 3): see d2m_search
constexpr float kLossScale = 1048576.f;     // 2^20: loss in units of 2^-20 mm (e <= 50 -> < 2^26 per point)
constexpr float kGradScale = 67108864.f;    // 2^26 per unit-vector component

struct D2mCtx {
  const float4 *s_c;            // LDS: (cx, cy, cz, r) per sphere
  float4 cj;                    // lanes = spheres: this lane's record (zeros beyond J)
  unsigned long long all;       // mask of the J spheres
  bool table_odd;               // a non-finite record somewhere
  int J, lane;
  Axis ax, ay;
  // fixed-point gradient rows [table][sphere][x, y, z, -] (u64): kD2mTables copies `acc_stride` apart, a lane adds
  // into copy lane & (kD2mTables - 1).  Neighbouring lanes hold neighbouring pixels -- mostly one owner -- and a 64-bit
  // LDS atomic is serialised over the lanes that share its ADDRESS; the copies start 4 banks apart (stride = 4 J' + 2).
  unsigned long long *s_acc;
  int acc_stride;
  int *s_nan;                   // set when a term is NaN (the crop's loss is then NaN, torch.clamp keeps it)
};

// entry(i), 0 <= i < count <= 64 K: the group's i-th point as (v << 16 | u, bits(z)).
// BOX2D = false: the entries are in pixel order and the bound is the strip of rows between the first and the last one.
// BOX2D = true (tile-sorted points): any order; the bound is the points' own bounding BOX in x, y and z (two transposed
// wave minima over the coordinates the lanes already hold),
//     | ||p - c|| - r | >= dist(c, box) - r,
// which a group of one or two 16 x 16-pixel tiles makes far tighter than a strip across the whole hand: at 256 x 256,
// 1 mm of pose noise, 256-point groups (tools/analyze_d2m_groups.py) 6.9 + 0.8 spheres are evaluated per point (stage 1
// + stage 2) against 11.8 + 1.4 with pixel-order strips; the depth extent of a tile alone removes a quarter (9.3 + 1.0
// with the x-y box only).  Both bounds are conservative: the minimum and its first index are
// exact either way, so the fixed-point terms -- and the sums -- do not depend on the grouping.
// (core: the group's points already in registers as COORDINATES -- px, py = the grid coordinates of mesh/render.py:31-32
// (axis_coord), pz = the depth value; lane l holds points K l + i; first_v / last_v = rows of the group's first and
// last entry, used by the strip bound only)
template <int K, bool WANT_GRAD, bool BOX2D>
__device__ __forceinline__ void d2m_search_core(const D2mCtx &cx, const float (&px)[K], const float (&py)[K],
                                                const float (&pz)[K], int count, int first_v, int last_v,
                                                long long &loss_fx) {
  const int lane = cx.lane;
  const Axis &ay = cx.ay;
  const float4 *s_c = cx.s_c;
  const float4 cj = cx.cj;
  // a sphere's record by an explicit LDS read whose wait is placed by hand (see stage 1 below)
  typedef float f4 __attribute__((ext_vector_type(4)));
  const unsigned table_base = (unsigned)(size_t)s_c;
  auto lds_request = [&](int j) {
    f4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(table_base + 16u * (unsigned)j));
    return r;
  };
  auto lds_arrived = [&](f4 &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r)); };

  float best[K];
  int bj[K];
  bool valid[K];
  bool zbad = false;
#pragma unroll
  for (int i = 0; i < K; i++) {
    valid[i] = K * lane + i < count;
    zbad |= !(fabsf(pz[i]) < __builtin_inff());
  }
  auto eval = [&](const float4 c, int j, bool tie_rule) {
#pragma unroll
    for (int i = 0; i < K; i++) {
      const float dx = px[i] - c.x, dy = py[i] - c.y, dz = pz[i] - c.z;
      const float t = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const float a = fabsf(__builtin_amdgcn_sqrtf(t) - c.w);   // <= 1 ulp root: the loss is continuous
      const bool lt = tie_rule ? (a < best[i] || (a == best[i] && j < bj[i])) : (a < best[i]);
      bj[i] = lt ? j : bj[i];
      best[i] = lt ? a : best[i];
    }
  };
  if (!cx.table_odd && __ballot(zbad) == 0ull) {
    // All inputs finite: no NaN can arise (an overflowing distance is +inf).  Bounds with lanes = spheres.
    float lb;
    if (BOX2D) {
      const float inf = __builtin_inff();
      float x0 = inf, x1 = inf, y0 = inf, y1 = inf, z0 = inf, z1 = inf;   // (min x, min -x, ... over the lane's valid points)
#pragma unroll
      for (int i = 0; i < K; i++) {
        x0 = fminf(x0, valid[i] ? px[i] : inf); x1 = fminf(x1, valid[i] ? -px[i] : inf);
        y0 = fminf(y0, valid[i] ? py[i] : inf); y1 = fminf(y1, valid[i] ? -py[i] : inf);
        z0 = fminf(z0, valid[i] ? pz[i] : inf); z1 = fminf(z1, valid[i] ? -pz[i] : inf);
      }
      const float m = wave_min4_transposed(x0, x1, y0, y1, lane), mz = wave_min4_transposed(z0, z1, z0, z1, lane);
      const float x_lo = readlane_f(m, 12), x_hi = -readlane_f(m, 13), y_lo = readlane_f(m, 14), y_hi = -readlane_f(m, 15);
      const float z_lo = readlane_f(mz, 12), z_hi = -readlane_f(mz, 13);
      const float ddx = fmaxf(fmaxf(x_lo - cj.x, cj.x - x_hi), 0.f), ddy = fmaxf(fmaxf(y_lo - cj.y, cj.y - y_hi), 0.f);
      const float ddz = fmaxf(fmaxf(z_lo - cj.z, cj.z - z_hi), 0.f);
      // (v_sqrt_f32 is within an ulp, the sum of squares within three: the margins below are 1e-5 relative + 1e-3 mm)
      lb = __builtin_amdgcn_sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz) - cj.w;
    } else {
      // rows of the first / last entry (pixel order inside a group)
      const float y_lo = axis_coord(ay, first_v), y_hi = axis_coord(ay, last_v);
      lb = fmaxf(fmaxf(y_lo - cj.y, cj.y - y_hi), 0.f) - cj.w;
    }
    // lb <= | ||p - c_j|| - r_j | for every point of the group, up to the rounding the margins below cover
    unsigned long long m1 = __ballot(!(lb > 1e-3f)) & cx.all;     // the sphere's extent meets the strip / box (or nearly)
    if (m1 == 0ull) {
      if (BOX2D) {   // observed points no sphere is near: start from the nearest one(s), the reach does the rest
        const float lbmin = wave_minmax_all<true>(lane < cx.J ? lb : __builtin_inff());
        m1 = __ballot(lb <= lbmin) & cx.all;
      }
      if (m1 == 0ull) m1 = cx.all;
    }
#pragma unroll
    for (int i = 0; i < K; i++) { best[i] = __builtin_inff(); bj[i] = 0; }
    // stage 1, ascending j, strict '<': ties keep the first index (torch.min's convention).
    // The NEXT sphere's record is requested before the current one is evaluated (explicit ds_read_b128 +
    // s_waitcnt: left to itself hipcc reads the record at the top of the iteration and waits for it at once,
    // one exposed LDS round trip per sphere).
    {
      unsigned long long m = m1;
      int j = __builtin_ctzll(m);
      f4 c = lds_request(j);
      lds_arrived(c);
      while (true) {
        m &= m - 1;
        const int jn = m ? __builtin_ctzll(m) : j;
        f4 cn = lds_request(jn);
        eval(make_float4(c.x, c.y, c.z, c.w), j, false);
        lds_arrived(cn);
        if (!m) break;
        j = jn; c = cn;
      }
    }
    // stage 2: what could still win or tie.  A point whose minimum stays above 50 is worth exactly 50
    // with no gradient whatever the owner, so 50 caps the reach.
    unsigned long long m2 = cx.all & ~m1;
    if (m2) {
      float wmax = -__builtin_inff();
#pragma unroll
      for (int i = 0; i < K; i++) wmax = fmaxf(wmax, valid[i] ? best[i] : -__builtin_inff());
      const float reach = fminf(wave_minmax_all<false>(wmax), 50.f) * 1.00001f + 1e-3f;
      m2 &= __ballot(!(lb * 0.99999f > reach));
      if (m2) {
        int j = __builtin_ctzll(m2);
        f4 c = lds_request(j);
        lds_arrived(c);
        while (true) {
          m2 &= m2 - 1;
          const int jn = m2 ? __builtin_ctzll(m2) : j;
          f4 cn = lds_request(jn);
          eval(make_float4(c.x, c.y, c.z, c.w), j, true);
          lds_arrived(cn);
          if (!m2) break;
          j = jn; c = cn;
        }
      }
    }
  }
  else {
    // a NaN / infinity somewhere: every sphere in index order with torch.min's NaN rule
    for (int j = 0; j < cx.J; j++) {
      const float4 c = s_c[j];
#pragma unroll
      for (int i = 0; i < K; i++) {
        const float dx = px[i] - c.x, dy = py[i] - c.y, dz = pz[i] - c.z;
        const float a = fabsf(__builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz) - c.w);
        if (j == 0 || ((best[i] == best[i]) && (a < best[i] || a != a))) { best[i] = a; bj[i] = j; }
      }
    }
  }
  // The search above compares candidates through v_sqrt_f32 (<= 1 ulp); the point's TERM is then evaluated once more
  // for its owner with a correctly rounded root and the reference's association -- | sqrtf((dx dx + dy dy) + dz dz) - r |,
  // mesh/render.py:131-137 on a host -- so that loss and gradient are the reference's arithmetic wherever the owner is
  // (two spheres within an ulp of each other can swap; the term then differs by that ulp).  sqrt_rn() is exact on
  // [0.01, 1e12]; a squared distance outside it (a point within 0.1 mm of a centre, or 1 km away, or not finite) takes
  // hipcc's IEEE sqrtf -- behind a wave-uniform branch: as a select, its ~20-instruction expansion ran for every point.
  float t2s[K], dists[K];
  bool wide = false;
#pragma unroll
  for (int i = 0; i < K; i++) {
    const float4 c = s_c[bj[i]];
    const float dx = px[i] - c.x, dy = py[i] - c.y, dz = pz[i] - c.z;
    t2s[i] = (dx * dx + dy * dy) + dz * dz;
    dists[i] = sqrt_rn(t2s[i]);
    wide |= !(t2s[i] >= 0.01f && t2s[i] <= 1e12f);
  }
  if (__ballot(wide) != 0ull) {
#pragma unroll
    for (int i = 0; i < K; i++) dists[i] = (t2s[i] >= 0.01f && t2s[i] <= 1e12f) ? dists[i] : __builtin_sqrtf(t2s[i]);
  }
  bool nan = false;
  int lsum = 0;                                       // (K terms of at most 50 * 2^20 < 2^26 each: 32 bits hold them)
  static_assert(K <= 16, "the lane's loss terms are added in 32 bits first");
  float term[K];
#pragma unroll
  for (int i = 0; i < K; i++) {
    term[i] = fabsf(dists[i] - s_c[bj[i]].w);
    nan |= valid[i] && term[i] != term[i];            // torch.clamp keeps NaN: the crop's loss is NaN
    lsum += (valid[i] && term[i] == term[i]) ? __float2int_rn(fminf(fmaxf(term[i], 0.f), 50.f) * kLossScale) : 0;
  }
  loss_fx += (long long)lsum;
  if (nan) *cx.s_nan = 1;
  if (WANT_GRAD) {
    int g[K][3];
    bool live[K];
#pragma unroll
    for (int i = 0; i < K; i++) {
      const float4 c = s_c[bj[i]];
      const float dx = px[i] - c.x, dy = py[i] - c.y, dz = pz[i] - c.z;
      const float dist = dists[i];
      const float t = dist - c.w;
      live[i] = valid[i] && term[i] <= 50.f && dist != 0.f && t != 0.f && dist < __builtin_inff();
      const float k = (t > 0.f ? -kGradScale : kGradScale) * __builtin_amdgcn_rcpf(dist);
      g[i][0] = live[i] ? __float2int_rn(k * dx) : 0;
      g[i][1] = live[i] ? __float2int_rn(k * dy) : 0;
      g[i][2] = live[i] ? __float2int_rn(k * dz) : 0;
    }
    // a lane's K points are neighbouring pixels: those sharing point 0's owner go with it
#pragma unroll
    for (int i = 1; i < K; i++) {
      const bool same = live[i] && live[0] && bj[i] == bj[0];
      g[0][0] += same ? g[i][0] : 0; g[0][1] += same ? g[i][1] : 0; g[0][2] += same ? g[i][2] : 0;
      live[i] = live[i] && !same;
    }
#pragma unroll
    for (int i = 0; i < K; i++) {
      if (live[i]) {
        unsigned long long *row = cx.s_acc + (lane & (kD2mTables - 1)) * cx.acc_stride + bj[i] * 4;
        atomicAdd(row + 0, (unsigned long long)(long long)g[i][0]);
        atomicAdd(row + 1, (unsigned long long)(long long)g[i][1]);
        atomicAdd(row + 2, (unsigned long long)(long long)g[i][2]);
      }
    }
  }
}

// entry(i), 0 <= i < count <= 64 K: the group's i-th point as (v << 16 | u, bits(z)) (a lane without a point reads
// entry 0: finite coordinates, never counted).
template <int K, bool WANT_GRAD, bool BOX2D = false, typename Entry>
__device__ __forceinline__ void d2m_search(const D2mCtx &cx, Entry &&entry, int count, long long &loss_fx) {
  float px[K], py[K], pz[K];
#pragma unroll
  for (int i = 0; i < K; i++) {
    const int idx = K * cx.lane + i;
    const uint2 e = entry(idx < count ? idx : 0);
    px[i] = axis_coord(cx.ax, (int)(e.x & 0xffffu));
    py[i] = axis_coord(cx.ay, (int)(e.x >> 16));
    pz[i] = __uint_as_float(e.y);
  }
  int first_v = 0, last_v = 0;
  if (!BOX2D) {
    first_v = __builtin_amdgcn_readfirstlane((int)(entry(0).x >> 16));
    last_v = __builtin_amdgcn_readfirstlane((int)(entry(count - 1).x >> 16));
  }
  d2m_search_core<K, WANT_GRAD, BOX2D>(cx, px, py, pz, count, first_v, last_v, loss_fx);
}

// The same for a group already in registers (the two-step path's lists): e[i] = (v << 16 | u, bits(depth)), box bound.
template <int K, bool WANT_GRAD>
__device__ __forceinline__ void d2m_search_points(const D2mCtx &cx, const uint2 (&e)[K], int count, long long &loss_fx) {
  float px[K], py[K], pz[K];
#pragma unroll
  for (int i = 0; i < K; i++) {
    px[i] = axis_coord(cx.ax, (int)(e[i].x & 0xffffu));
    py[i] = axis_coord(cx.ay, (int)(e[i].x >> 16));
    pz[i] = __uint_as_float(e[i].y);
  }
  d2m_search_core<K, WANT_GRAD, true>(cx, px, py, pz, count, 0, 0, loss_fx);
}

}  // namespace shr
