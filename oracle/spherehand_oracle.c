/* spherehand_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference (melonwan/sphereHand) algorithms on the
 * rasterize / fit hot path.  Build: see oracle/Makefile (gcc -O3
 * -ffp-contract=off, no fast-math: every fp32 operation below is a single
 * correctly rounded IEEE operation in exactly the written association).
 *
 * PARITY PINNING (details: DESIGN.md section 3)
 *   - sphere path (oracle_ball_render, oracle_sphere_raster_fwd/_bwd),
 *     oracle_data_to_model_*, oracle_lbs_project, oracle_fk_fwd,
 *     oracle_clamp_bilinear: PINNED against outputs of the imported PyTorch
 *     reference generated in the build container (tests/golden/ (the .npz files), made by
 *     tests/golden/make_goldens_*.py) -- bit-exact for the depth maps,
 *     tolerance for summed quantities (stated in the tests).
 *   - oracle_tri_raster_fwd: PINNED (round 4) against the reference's own
 *     device code executed on the MI355X: oracle/_ref/libref_tri.so is
 *     `kernel` + `atomicMin` of depth_rasterization_cuda_kernel.cu:1-113
 *     compiled for gfx950 where the file lies (oracle/Makefile, target
 *     ref_tri; -ffp-contract=off: every operation as the source writes it);
 *     tests/test_tri_reference_gpu.py: bit-exact on the hand mesh at 640x640,
 *     the quirk cases and random soups.  NOT reproducible here: nvcc's
 *     default mul+add fusion (-fmad=true) -- its pattern is the compiler's;
 *     clang's own fusion moves 2.6 % of the covered pixels by more than 1e-4
 *     relative (same test file).  Also cross-checked against an independent
 *     numpy restatement (tests/test_oracle_tri.py).
 *
 * Who may use this file: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.  The product path never links or loads it.
 */
#include "spherehand_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0; /* 0 = OpenMP default */

int oracle_num_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}
void oracle_set_num_threads(int n) { g_threads = n; }

#ifdef _OPENMP
#define OMP_FOR _Pragma("omp parallel for schedule(dynamic, 1) num_threads(oracle_num_threads())")
#else
#define OMP_FOR
#endif

/* Pixel grid, mesh/render.py:31-32:
 *   x_grid = (x_grid - self.width/2) * 300.0 / self.width
 * three fp32 tensor-scalar operations: sub, mul, div. */
static inline float grid_coord(int u, int W) {
  float half = (float)((double)W / 2.0);
  float t = (float)u - half;
  t = t * 300.0f;
  return t / (float)W;
}

/* mesh/render.py:37-52 for one (pixel, sphere):
 *   s = clamp(r**2 - (xg-x)**2 - (yg-y)**2, min=1e-2); hit = s != 1e-2
 *   d = z - sqrt(s) if hit else 100.0                                      */
static inline float ball_depth(float xg, float yg, float x, float y, float z, float rr,
                               int *hit, float *sq_out) {
  float dx = xg - x;
  float dy = yg - y;
  float q = (rr - dx * dx) - dy * dy;
  float s = (q < 0.01f) ? 0.01f : q; /* torch.clamp(min): NaN stays NaN */
  *hit = (s != 0.01f);
  float sq = sqrtf(s);
  *sq_out = sq;
  return *hit ? (z - sq) : 100.0f;
}

int oracle_ball_render(const float *centres, int stride, const float *radii, int n, int H,
                       int W, float *maps) {
  if (!centres || !radii || !maps || stride < 3 || n < 0 || H <= 0 || W <= 0) return -1;
  OMP_FOR
  for (int i = 0; i < n; i++) {
    const float x = centres[(size_t)i * stride + 0];
    const float y = centres[(size_t)i * stride + 1];
    const float z = centres[(size_t)i * stride + 2];
    const float rr = radii[i] * radii[i];
    float *m = maps + (size_t)i * H * W;
    for (int v = 0; v < H; v++) {
      const float yg = grid_coord(v, H);
      for (int u = 0; u < W; u++) {
        int hit;
        float sq;
        m[(size_t)v * W + u] = ball_depth(grid_coord(u, W), yg, x, y, z, rr, &hit, &sq);
      }
    }
  }
  return 0;
}

/* torch.min(dim) (mesh/render.py:89, multiview_utility.py:76): NaN propagates,
 * first index kept on ties. */
static inline int min_takes(float d, float best) { return (best == best) && (d < best || d != d); }

int oracle_sphere_raster_fwd(const float *spheres, int N, int J, int H, int W, float *depth,
                             uint8_t *argmin) {
  if (!spheres || !depth || N < 0 || J <= 0 || J > 255 || H <= 0 || W <= 0) return -1;
  float *xg = (float *)malloc(sizeof(float) * (size_t)W);
  if (!xg) return -2;
  for (int u = 0; u < W; u++) xg[u] = grid_coord(u, W);
  OMP_FOR
  for (int n = 0; n < N; n++) {
    const float *sp = spheres + (size_t)n * J * 4;
    float *out = depth + (size_t)n * H * W;
    uint8_t *am = argmin ? argmin + (size_t)n * H * W : NULL;
    uint8_t *arow = (uint8_t *)malloc((size_t)W * 2);
    uint8_t *hrow = arow + W;
    for (int v = 0; v < H; v++) {
      const float yg = grid_coord(v, H);
      float *row = out + (size_t)v * W;
      for (int j = 0; j < J; j++) {
        const float x = sp[4 * j + 0], y = sp[4 * j + 1], z = sp[4 * j + 2];
        const float rr = sp[4 * j + 3] * sp[4 * j + 3];
        for (int u = 0; u < W; u++) {
          int hit;
          float sq;
          float d = ball_depth(xg[u], yg, x, y, z, rr, &hit, &sq);
          if (j == 0 || min_takes(d, row[u])) {
            row[u] = d;
            arow[u] = (uint8_t)j;
            hrow[u] = (uint8_t)hit;
          }
        }
      }
      if (am)
        for (int u = 0; u < W; u++) am[(size_t)v * W + u] = hrow[u] ? arow[u] : 255;
    }
    free(arow);
  }
  free(xg);
  return 0;
}

/* Autograd of BallRender + min for upstream g.  Chain, in the
 * order autograd evaluates it (fp32):
 *   depth = z - sq           -> g_z = g ; g_sq = -g
 *   sq = sqrt(s)             -> g_s = g_sq / (2*sq)
 *   s  = clamp(q)            -> g_q = g_s          (hit pixels only)
 *   q  = (rr - xd) - yd      -> g_rr = g_q ; g_xd = g_yd = -g_q
 *   rr = r**2 ; xd = dx**2   -> g_r = g_rr*(2*r) ; g_dx = g_xd*(2*dx) ; g_x = -g_dx
 * Sum over pixels in fp64, rounded once. */
int oracle_sphere_raster_bwd(const float *spheres, const float *grad_depth, int N, int J,
                             int H, int W, float *grad_spheres) {
  if (!spheres || !grad_depth || !grad_spheres || N < 0 || J <= 0 || J > 255 || H <= 0 || W <= 0)
    return -1;
  float *xg = (float *)malloc(sizeof(float) * (size_t)W);
  if (!xg) return -2;
  for (int u = 0; u < W; u++) xg[u] = grid_coord(u, W);
  OMP_FOR
  for (int n = 0; n < N; n++) {
    const float *sp = spheres + (size_t)n * J * 4;
    const float *gd = grad_depth + (size_t)n * H * W;
    double *acc = (double *)calloc((size_t)J * 4, sizeof(double));
    for (int v = 0; v < H; v++) {
      const float yg = grid_coord(v, H);
      for (int u = 0; u < W; u++) {
        float best = 0.f, bsq = 0.f;
        int bj = 0, bhit = 0;
        for (int j = 0; j < J; j++) {
          int hit;
          float sq;
          float d = ball_depth(xg[u], yg, sp[4 * j], sp[4 * j + 1], sp[4 * j + 2],
                               sp[4 * j + 3] * sp[4 * j + 3], &hit, &sq);
          if (j == 0 || min_takes(d, best)) {
            best = d; bj = j; bhit = hit; bsq = sq;
          }
        }
        if (!bhit) continue;
        const float g = gd[(size_t)v * W + u];
        const float dx = xg[u] - sp[4 * bj], dy = yg - sp[4 * bj + 1], r = sp[4 * bj + 3];
        const float g_sq = -g;
        const float g_q = g_sq / (2.0f * bsq);
        const float g_xd = -g_q;
        const float g_dx = g_xd * (2.0f * dx);
        const float g_dy = g_xd * (2.0f * dy);
        acc[4 * bj + 0] += (double)(-g_dx);
        acc[4 * bj + 1] += (double)(-g_dy);
        acc[4 * bj + 2] += (double)g;
        acc[4 * bj + 3] += (double)(g_q * (2.0f * r));
      }
    }
    for (int k = 0; k < 4 * J; k++) grad_spheres[(size_t)n * J * 4 + k] = (float)acc[k];
    free(acc);
  }
  free(xg);
  return 0;
}

/* mesh/render.py:123-142.  Per pixel with depth <= 99 (":138 background = d>99"):
 *   e = min_j | ||(xg,yg,depth) - c_j||_2 - r_j | ; clamp(e, 0, 50)          */
static inline float d2m_pixel(float px, float py, float pz, const float *c, const float *radii,
                              int J, int *argj, float *dist_out, float *dxyz) {
  float best = 0.f;
  int bj = 0;
  for (int j = 0; j < J; j++) {
    float dx = px - c[3 * j], dy = py - c[3 * j + 1], dz = pz - c[3 * j + 2];
    float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
    float a = fabsf(dist - radii[j]);
    if (j == 0 || min_takes(a, best)) {
      best = a; bj = j;
      if (dist_out) { *dist_out = dist; dxyz[0] = dx; dxyz[1] = dy; dxyz[2] = dz; }
    }
  }
  *argj = bj;
  return best;
}

int oracle_data_to_model_fwd(const float *depth, const float *centres, const float *radii, int N,
                             int J, int H, int W, double *loss_sum_per_crop) {
  if (!depth || !centres || !radii || !loss_sum_per_crop || N < 0 || J <= 0 || H <= 0 || W <= 0)
    return -1;
  OMP_FOR
  for (int n = 0; n < N; n++) {
    const float *dm = depth + (size_t)n * H * W;
    const float *c = centres + (size_t)n * J * 3;
    double acc = 0.0;
    for (int v = 0; v < H; v++) {
      const float yg = grid_coord(v, H);
      for (int u = 0; u < W; u++) {
        const float z = dm[(size_t)v * W + u];
        if (z > 99.0f) continue; /* background contributes 0 */
        int bj;
        float e = d2m_pixel(grid_coord(u, W), yg, z, c, radii, J, &bj, NULL, NULL);
        e = e < 0.f ? 0.f : (e > 50.f ? 50.f : e);
        acc += (double)e;
      }
    }
    loss_sum_per_crop[n] = acc;
  }
  return 0;
}

/* autograd of mean(clamp(min_j |norm(p - c_j) - r_j|)):
 *   clamp passes grad iff 0 <= e <= 50; abs' = sign(dist - r); norm' = diff/dist
 *   (0 at dist == 0); d(p - c)/dc = -1.                                      */
int oracle_data_to_model_bwd(const float *depth, const float *centres, const float *radii, int N,
                             int J, int H, int W, float *grad_centres) {
  if (!depth || !centres || !radii || !grad_centres || N < 0 || J <= 0 || H <= 0 || W <= 0)
    return -1;
  const double inv_count = 1.0 / ((double)N * H * W);
  OMP_FOR
  for (int n = 0; n < N; n++) {
    const float *dm = depth + (size_t)n * H * W;
    const float *c = centres + (size_t)n * J * 3;
    double *acc = (double *)calloc((size_t)J * 3, sizeof(double));
    for (int v = 0; v < H; v++) {
      const float yg = grid_coord(v, H);
      for (int u = 0; u < W; u++) {
        const float z = dm[(size_t)v * W + u];
        if (z > 99.0f) continue;
        int bj;
        float dist = 0.f, d3[3] = {0.f, 0.f, 0.f};
        float e = d2m_pixel(grid_coord(u, W), yg, z, c, radii, J, &bj, &dist, d3);
        if (!(e <= 50.f)) continue;
        float t = dist - radii[bj];
        float sgn = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
        if (dist == 0.f || sgn == 0.f) continue;
        for (int k = 0; k < 3; k++) acc[3 * bj + k] += (double)(-(sgn * (d3[k] / dist)));
      }
    }
    for (int k = 0; k < 3 * J; k++)
      grad_centres[(size_t)n * J * 3 + k] = (float)(acc[k] * inv_count);
    free(acc);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Triangle z-buffer: mesh/cuda_kernel/depth_rasterization_cuda_kernel.cu.
 * CUDA semantics restated explicitly:
 *   max/min(double,double) are fmax/fmin (a NaN operand is dropped);
 *   double -> int32 conversion truncates toward zero and saturates (NaN -> 0);
 *   literals `0.`, `1.`, `width - 1.` promote the expression to double;
 *   no FMA contraction (the reference kernel compiled the same way gives these
 *   bits on the GPU; nvcc's default would contract: DESIGN.md section 3).     */
static inline int32_t cvt_rz_sat_i32(double d) {
  if (d != d) return 0;
  if (d >= 2147483647.0) return 2147483647;
  if (d <= -2147483648.0) return (int32_t)(-2147483647 - 1);
  return (int32_t)d;
}

static void tri_face(const float *face, int bn, int width, int height, float *depth_map) {
  /* .cu:33 back-face cull */
  if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) return;
  /* .cu:37-45 sort by x */
  int pi[3] = {0, 0, 0};
  if (face[0] < face[3]) {
    pi[0] = (face[6] < face[0]) ? 2 : 0;
    pi[2] = (face[3] < face[6]) ? 2 : 1;
  } else {
    pi[0] = (face[6] < face[3]) ? 2 : 1;
    pi[2] = (face[0] < face[6]) ? 2 : 0;
  }
  for (int k = 0; k < 3; k++)
    if (pi[0] != k && pi[2] != k) pi[1] = k;
  float p[3][3];
  for (int a = 0; a < 3; a++)
    for (int d = 0; d < 3; d++) p[a][d] = face[3 * pi[a] + d];
  if (p[0][0] == p[2][0]) return; /* .cu:54 */
  /* .cu:57-65 */
  float fi[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                 p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                 p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
  float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1])) +
              p[1][0] * (p[2][1] - p[0][1]);
  for (int k = 0; k < 9; k++) fi[k] /= den;
  /* .cu:68-69 */
  const int32_t xi_min = cvt_rz_sat_i32(fmax((double)ceilf(p[0][0]), 0.));
  const int32_t xi_max = cvt_rz_sat_i32(fmin((double)p[2][0], (double)width - 1.));
  for (int32_t xi = xi_min; xi <= xi_max; xi++) {
    float yi1, yi2;
    const float xf = (float)xi;
    if (xf <= p[1][0]) {
      if (p[1][0] - p[0][0] != 0)
        yi1 = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (xf - p[0][0]) + p[0][1];
      else
        yi1 = p[1][1];
    } else {
      if (p[2][0] - p[1][0] != 0)
        yi1 = (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]) * (xf - p[1][0]) + p[1][1];
      else
        yi1 = p[1][1];
    }
    yi2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (xf - p[0][0]) + p[0][1];
    /* .cu:89-90 */
    const int32_t yi_min = cvt_rz_sat_i32(fmax(0., (double)ceilf(fminf(yi1, yi2))));
    const int32_t yi_max = cvt_rz_sat_i32(fmin((double)fmaxf(yi1, yi2), (double)height - 1.));
    for (int32_t yi = yi_min; yi <= yi_max; yi++) {
      const float yf = (float)yi;
      float w[3];
      for (int k = 0; k < 3; k++) w[k] = (fi[3 * k + 0] * xf + fi[3 * k + 1] * yf) + fi[3 * k + 2];
      float w_sum = 0;
      for (int k = 0; k < 3; k++) {
        w[k] = (float)fmin(fmax((double)w[k], 0.), 1.);
        w_sum += w[k];
      }
      for (int k = 0; k < 3; k++) w[k] /= w_sum;
      const float zp = (float)(1. / (double)((w[0] / p[0][2] + w[1] / p[1][2]) + w[2] / p[2][2]));
      /* .cu:6-16 atomicMin == fminf(val, old) */
      float *dst = &depth_map[(size_t)bn * width * height + (size_t)yi * width + xi];
      *dst = fminf(zp, *dst);
    }
  }
}

int oracle_tri_raster_fwd(const float *face_vertices, int B, int F, int W, int H, float *depth) {
  if (!face_vertices || !depth || B < 0 || F < 0 || W <= 0 || H <= 0) return -1;
  OMP_FOR
  for (int b = 0; b < B; b++) {
    float *dm = depth + (size_t)b * W * H;
    for (size_t i = 0; i < (size_t)W * H; i++) dm[i] = 1000.0f; /* .cu:122 */
    for (int f = 0; f < F; f++) tri_face(face_vertices + ((size_t)b * F + f) * 9, b, W, H, depth);
  }
  return 0;
}

/* mesh/render.py:286 + :311.  ATen bilinear, align_corners=False:
 *   scale = (float)in/out ; src = scale*(dst+0.5)-0.5, clamped at 0 ;
 *   i0 = (int)src ; i1 = i0 + (i0 < in-1) ; l1 = src - i0 ; l0 = 1 - l1
 *   out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)                       */
static inline void lin_idx(int d, float scale, int in, int *i0, int *i1, float *l0, float *l1) {
  float src = scale * ((float)d + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int a = (int)src;
  if (a > in - 1) a = in - 1;
  *i0 = a;
  *i1 = a + ((a < in - 1) ? 1 : 0);
  *l1 = src - (float)a;
  *l0 = 1.0f - *l1;
}

int oracle_clamp_bilinear(const float *src, int B, int Hs, int Ws, int Hd, int Wd, float clamp_max,
                          float *dst) {
  if (!src || !dst || B < 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return -1;
  const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
  OMP_FOR
  for (int b = 0; b < B; b++) {
    const float *s = src + (size_t)b * Hs * Ws;
    float *o = dst + (size_t)b * Hd * Wd;
    for (int y = 0; y < Hd; y++) {
      int y0, y1;
      float h0, h1;
      lin_idx(y, sh, Hs, &y0, &y1, &h0, &h1);
      for (int x = 0; x < Wd; x++) {
        int x0, x1;
        float w0, w1;
        lin_idx(x, sw, Ws, &x0, &x1, &w0, &w1);
        float v00 = s[(size_t)y0 * Ws + x0], v01 = s[(size_t)y0 * Ws + x1];
        float v10 = s[(size_t)y1 * Ws + x0], v11 = s[(size_t)y1 * Ws + x1];
        v00 = v00 > clamp_max ? clamp_max : v00;
        v01 = v01 > clamp_max ? clamp_max : v01;
        v10 = v10 > clamp_max ? clamp_max : v10;
        v11 = v11 > clamp_max ? clamp_max : v11;
        o[(size_t)y * Wd + x] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
      }
    }
  }
  return 0;
}

/* mesh/pointTransformation.py:39-46 (+ :84-99).  The reference multiplies every
 * bone's 4x4 with a dense [17,NV,4] buffer that is zero except where a bone
 * skins the vertex, then sums over bones; adding exact zeros changes nothing,
 * so only the non-zero (bone, vertex) pairs are visited, in ascending bone
 * order.  skin_wv = float32(w * v) as the reference stores it (:31).        */
int oracle_lbs_project(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                       const int32_t *skin_bone, const float *skin_wv, int right_hand, int project,
                       float cx, float cy, float fx, float fy, const float *rand_f, float *out) {
  if (!T || !skin_vertex_start || !skin_bone || !skin_wv || !out || B < 0 || NB <= 0 || NV < 0)
    return -1;
  OMP_FOR
  for (int b = 0; b < B; b++) {
    const float *Tb = T + (size_t)b * NB * 16;
    for (int v = 0; v < NV; v++) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int e = skin_vertex_start[v]; e < skin_vertex_start[v + 1]; e++) {
        const float *M = Tb + (size_t)skin_bone[e] * 16;
        const float *q = skin_wv + (size_t)e * 4;
        for (int r = 0; r < 4; r++)
          acc[r] += ((M[4 * r] * q[0] + M[4 * r + 1] * q[1]) + M[4 * r + 2] * q[2]) + M[4 * r + 3] * q[3];
      }
      if (right_hand) acc[0] = -acc[0]; /* :44-45 */
      float *o = out + ((size_t)b * NV + v) * 4;
      if (!project) {
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
      } else if (!rand_f) { /* :88-89 k_mat @ p */
        o[0] = fx * acc[0] + cx * acc[3];
        o[1] = fy * acc[1] + cy * acc[3];
        o[2] = acc[2];
        o[3] = acc[3];
      } else { /* :91-97 */
        o[0] = acc[0] * rand_f[b] * fx + cx;
        o[1] = acc[1] * rand_f[b] * fy + cy;
        o[2] = acc[2];
        o[3] = 1.0f;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* mesh/kinematicsTransformation.py                                           */
static void mat4_mul(const float *A, const float *Bm, float *C) {
  float t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
      for (int k = 0; k < 4; k++) s += A[4 * i + k] * Bm[4 * k + j];
      t[4 * i + j] = s;
    }
  memcpy(C, t, sizeof t);
}

/* :29-54 AxisRotationMatrix.forward */
static void axis_rot(const float ax[3], float angle, float *R) {
  const float x = ax[0], y = ax[1], z = ax[2];
  const float c = cosf(angle), s = sinf(angle), i = 1.0f - c;
  memset(R, 0, 16 * sizeof(float));
  R[15] = 1.0f;
  R[0] = (x * x) * i + c;     R[1] = (x * y) * i - z * s; R[2] = (x * z) * i + y * s;
  R[4] = (x * y) * i + z * s; R[5] = (y * y) * i + c;     R[6] = (y * z) * i - x * s;
  R[8] = (x * z) * i - y * s; R[9] = (y * z) * i + x * s; R[10] = (z * z) * i + c;
}

/* :92-112 FingerJoint.forward: parent @ ((offset^-1 @ local) @ offset) */
static void finger_joint(const float *off, const float *off_inv, const float *local,
                         const float *parent, float *out) {
  float t[16];
  mat4_mul(off_inv, local, t);
  mat4_mul(t, off, t);
  mat4_mul(parent, t, out);
}

int oracle_fk_fwd(const float *params, int B, const float *offset, const float *offset_inv,
                  float *T) {
  if (!params || !offset || !offset_inv || !T || B < 0) return -1;
  static const float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1}, NY[3] = {0, -1, 0};
  const float *abduct[5] = {Z, Z, NY, NY, Z}; /* :162-164 */
  OMP_FOR
  for (int b = 0; b < B; b++) {
    const float *p = params + (size_t)b * 26;
    float *Tb = T + (size_t)b * 17 * 16;
    float Rx[16], Ry[16], Rz[16], R[16], Tr[16], palm[16];
    /* :145-155 Palm.forward: Rz @ (Ry @ Rx), then translation @ rotation */
    axis_rot(X, p[0], Rx);
    axis_rot(Y, p[1], Ry);
    axis_rot(Z, p[2], Rz);
    mat4_mul(Ry, Rx, R);
    mat4_mul(Rz, R, R);
    memset(Tr, 0, sizeof Tr);
    Tr[0] = Tr[5] = Tr[10] = Tr[15] = 1.0f;
    Tr[3] = p[3]; Tr[7] = p[4]; Tr[11] = p[5];
    mat4_mul(Tr, R, palm);
    memcpy(Tb, palm, sizeof palm);
    memcpy(Tb + 16, palm, sizeof palm);
    for (int f = 0; f < 5; f++) { /* :173-174, Finger.forward :123-127 */
      const float *a = p + 6 + 4 * f;
      const int b0 = 2 + 3 * f;
      float Ra[16], Rf[16], L[16];
      axis_rot(abduct[f], a[0], Ra);
      axis_rot(X, a[1], Rf);
      mat4_mul(Ra, Rf, L);
      finger_joint(offset + 16 * b0, offset_inv + 16 * b0, L, palm, Tb + 16 * b0);
      axis_rot(X, a[2], L);
      finger_joint(offset + 16 * (b0 + 1), offset_inv + 16 * (b0 + 1), L, Tb + 16 * b0,
                   Tb + 16 * (b0 + 1));
      axis_rot(X, a[3], L);
      finger_joint(offset + 16 * (b0 + 2), offset_inv + 16 * (b0 + 2), L, Tb + 16 * (b0 + 1),
                   Tb + 16 * (b0 + 2));
    }
  }
  return 0;
}
