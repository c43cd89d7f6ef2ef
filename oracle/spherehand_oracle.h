/* spherehand_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference algorithms on the hot path of
 * melonwan/sphereHand.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (spherehand_amd/) never
 * does and has no CPU fallback.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * it restates.  Parity pinning status is stated per function in
 * spherehand_oracle.c and in DESIGN.md section 3.
 *
 * All buffers are caller-allocated host memory, row-major, fp32 unless noted.
 * Return value: 0 on success, negative on argument error.
 */
#ifndef SPHEREHAND_ORACLE_H
#define SPHEREHAND_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* mesh/render.py:26-53 BallRender.forward: one [H,W] map per sphere, no min.
 * centres: n rows of `stride` floats (cols 0..2 used), radii[n]. */
int oracle_ball_render(const float *centres, int stride, const float *radii,
                       int n, int H, int W, float *maps);

/* mesh/render.py:81-90 / mesh/multiview_utility.py:72-76: BallRender + min over
 * the J spheres of a crop.  spheres[N,J,4] = (x,y,z,r).  argmin (may be NULL):
 * first index attaining the min among HIT spheres, 255 where no sphere hit or
 * every hit is >= 100 (background wins). */
int oracle_sphere_raster_fwd(const float *spheres, int N, int J, int H, int W,
                             float *depth, uint8_t *argmin);

/* autograd of the above: grad_spheres[N,J,4] = d<grad_depth,
 * depth>/d(x,y,z,r).  Per-pixel terms follow the autograd chain in fp32; the
 * sum over pixels is accumulated in fp64 and rounded once. */
int oracle_sphere_raster_bwd(const float *spheres, const float *grad_depth,
                             int N, int J, int H, int W, float *grad_spheres);

/* mesh/render.py:123-142 DataToModelLoss.forward.  depth[N,H,W], centres[N,J,3],
 * radii[J].  loss_sum_per_crop[N] = sum over pixels of clamp(min_j |..|,0,50)
 * (fp64 accumulate); the reference's scalar is sum(loss_sum)/(N*H*W). */
int oracle_data_to_model_fwd(const float *depth, const float *centres,
                             const float *radii, int N, int J, int H, int W,
                             double *loss_sum_per_crop);
/* autograd of the mean: grad_centres[N,J,3] for upstream scalar grad 1.0 */
int oracle_data_to_model_bwd(const float *depth, const float *centres,
                             const float *radii, int N, int J, int H, int W,
                             float *grad_centres);

/* mesh/cuda_kernel/depth_rasterization_cuda_kernel.cu:18-113 (+115-134 init):
 * face_vertices[B,F,3,3] pixel-space (x,y,z); depth[B,H,W] initialised to 1000. */
int oracle_tri_raster_fwd(const float *face_vertices, int B, int F, int W, int H,
                          float *depth);

/* mesh/render.py:286 clamp(max=100) + :311 F.interpolate(bilinear,
 * align_corners=False) from [B,Hs,Ws] to [B,Hd,Wd]. */
int oracle_clamp_bilinear(const float *src, int B, int Hs, int Ws, int Hd, int Wd,
                          float clamp_max, float *dst);

/* mesh/pointTransformation.py:39-46 LinearBlendSkinning (sparse restatement of
 * the dense sum) + :84-99 OthographicalProjection.  T[B,NB,4,4];
 * skin entries sorted by vertex: skin_vertex_start[NV+1], skin_bone[NS],
 * skin_wv[NS,4] (= float32(w*v), mesh/pointTransformation.py:31);
 * rand_f[B] or NULL.  out[B,NV,4]. project==0 skips the camera. */
int oracle_lbs_project(const float *T, int B, int NB, int NV,
                       const int32_t *skin_vertex_start, const int32_t *skin_bone,
                       const float *skin_wv, int right_hand, int project,
                       float cx, float cy, float fx, float fy,
                       const float *rand_f, float *out);

/* mesh/kinematicsTransformation.py:157-177 HandTransformationMat.forward.
 * params[B,26], offset[17,4,4], offset_inv[17,4,4] -> T[B,17,4,4]. */
int oracle_fk_fwd(const float *params, int B, const float *offset,
                  const float *offset_inv, float *T);

/* Threads used by the OpenMP loops over crops (1 if built without OpenMP). */
int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
