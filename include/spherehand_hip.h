/* spherehand_hip.h -- C ABI of libspherehand_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the rasterize-and-fit hot path of melonwan/sphereHand.
 * Plain pointers and sizes only: no torch / ATen types cross this boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - buffers are caller-allocated, contiguous, row-major fp32 unless noted;
 *     outputs are fully overwritten (no accumulate-into semantics);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     kernels are enqueued on it and the call returns without synchronising;
 *     nothing is allocated, freed or synchronised inside the library;
 *   - return value: 0 (SHR_OK) or a negative SHR_E* code; a positive value is
 *     a hipError_t reported by the launch.  shr_error_string() names either.
 *   - image grid: pixel (v,u) has model-space coordinates
 *       xg = (u - W/2)*300/W , yg = (v - H/2)*300/H            (mm)
 *     (reference mesh/render.py:31-32).
 *
 * Each entry point cites the reference interface it replaces
 * (file:line relative to the reference repo root).
 */
#ifndef SPHEREHAND_HIP_H
#define SPHEREHAND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHR_OK 0
#define SHR_EINVAL (-1)     /* null pointer / non-positive size / misaligned */
#define SHR_ETOOLARGE (-2)  /* size beyond what the kernels index (see each fn) */
#define SHR_ENODEVICE (-3)  /* no gfx950 device visible to the process */

#define SHR_MAX_SPHERES 64  /* spheres per crop (reference: 41) */
#define SHR_ARGMIN_NONE 255 /* argmin value of a pixel no sphere owns */

/* Library / device introspection ------------------------------------------ */
int shr_abi_version(void);               /* bumps on any signature change */
const char *shr_error_string(int code);  /* static string, never NULL */
/* 0 if a gfx950 device is current; fills name (<= name_len bytes, may be NULL),
 * compute-unit count and peak HBM clock-independent info are for logging only */
int shr_device_info(char *name_host, int name_len, int *num_cu_host);

/* Launch-shape tuning / test hooks (process-wide, not thread-safe; results never
 * depend on them).  *_LDS_BYTES cap the LDS a workgroup of the z-buffer kernels
 * may take (decides rows per region and workgroups per CU; FWD_LDS_BYTES 0 = the
 * default); FORCE_GENERAL = 1
 * routes every call to the general tile kernels. */
#define SHR_TUNE_FWD_LDS_BYTES 1
#define SHR_TUNE_FWD_OWNER_LDS_BYTES 2
#define SHR_TUNE_BWD_LDS_BYTES 3
#define SHR_TUNE_FORCE_GENERAL 4
#define SHR_TUNE_FWD_WAVES 5  /* waves per forward workgroup, 1..16 */
/* Shares of a crop's work list given to the four wave age groups of a 16-wave
 * workgroup (waves 0-3, 4-7, 8-11, 12-15; the SIMD arbitration favours the oldest):
 * one byte per group, oldest in the low byte, any scale (normalised internally). */
#define SHR_TUNE_FWD_SHARES 6
#define SHR_TUNE_BWD_SHARES 7
#define SHR_TUNE_D2M_WAVES 8       /* waves per workgroup of the data->model kernel: 0 = by batch size, 4, 8 or 16 */
#define SHR_TUNE_PERSISTENT 10     /* sphere rasterizer launches over more crops than the device holds: 0 = one workgroup per
                                    * crop, 1 = persistent workgroups (default), > 1 = that many workgroups */
#define SHR_TUNE_D2M_BAND_UNITS 9  /* 256-pixel units per band handed to a wave: 0 = by crop size */
#define SHR_TUNE_BWD_WAVES 12      /* waves per backward workgroup: 0 = by launch size (8 with half of a CU's LDS when the
                                    * launch has two workgroups per CU, else 16), 8, 16 */
#define SHR_TUNE_MSE_BOX 13        /* fused render-and-compare with the box z-buffer (two workgroups per CU): -1 = by launch
                                    * size, 0 = never, 1 = always (power-of-two images >= 32 wide), > 1 = with that many bytes
                                    * of LDS per workgroup */
#define SHR_TUNE_FWD_ZBUF_BYTES 11  /* forward z-buffer bytes per workgroup: 0 = by launch size (half of a CU's LDS when the
                                    * launch has two workgroups per CU, else one pass for any box); a box that exceeds
                                    * it is rasterized in passes over row bands */
#define SHR_TUNE_FWD_RUN_TABLE 14   /* forward, whole-crop workgroups on power-of-two images: runs on a sphere start from a
                                    * per-(sphere, lane) LDS table built by the idle waves: -1 = when it fits (default),
                                    * 0 = never, 1 = same as -1 */
#define SHR_TUNE_D2M_TILED 15        /* data->model kernel: 1 = units are 32 x 8-pixel tiles and a search is bounded by its
                                    * points' own box (needs W % 4 == 0 and 16-byte aligned images), 0 = 256 consecutive
                                    * pixels per unit and strip bounds (round 2's kernel), -1 = tiles from 192 x 192 pixels
                                    * on (default) */
#define SHR_TUNE_TRI_BAND 16         /* shr_tri_raster_fwd: -1 = the LDS band kernel wherever it fits (default: band height
                                    * and workgroups per crop planned per launch), 0 = always the global-atomic kernel,
                                    * n > 0 = the band kernel with bands of at most n rows */
#define SHR_TUNE_MESH_BAND 17        /* shr_mesh_depth_fwd / shr_mesh_render_fwd at sizes without a lattice kernel whose resize
                                    * samples at least half of the source pixels (S = 256 from 640): 1 = the triangle band
                                    * kernel with clamp + resize as its stream-out (default), 0 = the tile kernel */
int shr_set_tuning(int key, int value);
/* Self-test: adds to *mismatches (device, caller-zeroed u64) the number of fp32
 * bit patterns in [lo_bits, hi_bits) where the rasterizer's internal square root
 * differs from the correctly rounded one. */
int shr_selftest_sqrt(unsigned lo_bits, unsigned hi_bits,
                      unsigned long long *mismatches, void *stream);
/* Self-test: adds to *mismatches the number of pseudo-random (weights, corner depths) cases -- 4096 x 256 threads x
 * per_thread of them -- for which the triangle kernels' pixel depth with shared reciprocals (common.h tri_pixel_depth)
 * differs in any bit from the reference's seven plain IEEE divisions (.cu:97-110). */
int shr_selftest_division(unsigned seed, unsigned per_thread, unsigned long long *mismatches, void *stream);
/* Measurement hook (bench.py `roofline.launch_floor`, not part of the path): N workgroups of 1024 threads with
 * lds_bytes of dynamic LDS -- the sphere forward's launch shape at one crop per CU -- that only move the forward's
 * bytes: read spheres[N,J,4], write depth[N,H,W] (background) and the owner bytes of rows [row0, row1) of every crop
 * (argmin may be NULL), full-line 16-byte stores.  W % 16 == 0, buffers 16-byte aligned. */
int shr_selftest_launch_floor(const float *spheres, int N, int J, int H, int W, int row0, int row1,
                              float *depth, uint8_t *argmin, int lds_bytes, void *stream);

/* Sphere-set depth rasterizer ------------------------------------------------
 * Replaces BallRender.forward + the min over the sphere axis:
 *   mesh/render.py:26-53 (BallRender), :87-89 (HandBallPrimitiveRender),
 *   mesh/multiview_utility.py:72-76 (MutualProjection).
 * spheres[N,J,4] = (x, y, z, r) per sphere in the TARGET view's frame, mm.
 * depth[N,H,W]   = min_j ( hit_j ? z_j - sqrt(q_j) : 100 ),
 *                  q_j = (r*r - (xg-x)^2) - (yg-y)^2 , hit_j <=> q_j > 0.01f.
 * argmin[N,H,W] (uint8, may be NULL): index of the owning sphere, 255 where
 * the pixel is background.  J <= SHR_MAX_SPHERES.  With J == 1 this IS
 * BallRender.forward (one map per sphere).
 * Bit-exact with the reference's fp32 operation sequence (no FMA contraction,
 * IEEE sqrt). */
int shr_sphere_raster_fwd(const float *spheres, int N, int J, int H, int W,
                          float *depth, uint8_t *argmin, void *stream);
/* The same with `flags` (0 = shr_sphere_raster_fwd).
 * SHR_RASTER_OWNER_TOUCHED_ROWS: argmin is written only on the rows some sphere's pixel box touches; the
 * owner bytes of the other rows (background whatever the depths, half of a hand crop) are left as they were.
 * shr_sphere_raster_bwd never reads them -- it derives the same rows from the same records -- so the
 * autograd pair that replaces mesh/render.py:26-53 + :89 saves a tenth of the forward's stores; a caller that
 * wants the full owner map (the public `argmin` contract above) passes 0.  depth is always complete. */
#define SHR_RASTER_OWNER_TOUCHED_ROWS 1
int shr_sphere_raster_fwd_ex(const float *spheres, int N, int J, int H, int W,
                             float *depth, uint8_t *argmin, int flags, void *stream);

/* Analytic backward of the above (the reference gets it from autograd of
 * mesh/render.py:37-52 + torch.min): for upstream grad_depth[N,H,W],
 *   grad_spheres[N,J,4] = sum over the pixels sphere j owns of
 *     g * ( -(xg-x)/sqrt(q), -(yg-y)/sqrt(q), 1, -r/sqrt(q) ).
 * argmin: the owner map the forward wrote for the SAME spheres (fast path), or
 * NULL to recompute the owners (slower, no saved state needed).
 * Deterministic: fixed-order in-wave reductions, no float atomics. */
int shr_sphere_raster_bwd(const float *spheres, const float *grad_depth,
                          const uint8_t *argmin, int N, int J, int H, int W,
                          float *grad_spheres, void *stream);

/* Data-to-model loss ------------------------------------------------------------
 * Replaces DataToModelLoss.forward (mesh/render.py:123-142) and its autograd
 * backward in ONE pass.  depth[N,H,W] observed depth (background > 99),
 * centres[N,J,3], radii[J].  Per pixel with depth <= 99:
 *   e = min_j | ||(xg,yg,depth) - c_j|| - r_j | , clamped to [0,50].
 * loss_sum[N]: sum of e over the crop's pixels (the reference's scalar is
 *   sum(loss_sum) / (N*H*W)).
 * grad_centres[N,J,3] (may be NULL): d loss_sum[n] / d c_j; the caller scales it
 *   by upstream/(N*H*W).  Deterministic (no atomics). */
int shr_data_to_model(const float *depth, const float *centres, const float *radii,
                      int N, int J, int H, int W, float *loss_sum,
                      float *grad_centres, void *stream);
/* Same, crop n reading the depth image depth + depth_index[n]*H*W: the V*V view pairs
 * of mesh/multiview_utility.py:98-105 share V observed images, the reference's
 * expand().reshape() copy (9x the images) is not needed. */
int shr_data_to_model_indexed(const float *depth, const int32_t *depth_index,
                              const float *centres, const float *radii,
                              int N, int J, int H, int W, float *loss_sum,
                              float *grad_centres, void *stream);
/* The same in `parts` PARTIAL results per crop (1, 2 or 4): crop n's loss sum is
 * loss_parts[n*parts] + ... + loss_parts[n*parts + parts-1], likewise grad_parts[N*parts][J][3] -- large crops
 * are split over several workgroups on different CUs, and the caller adds the partials (fixed order:
 * deterministic).  shr_data_to_model_parts() returns the split this library prefers for a problem size
 * (1 below 192 x 192 pixels, else 2).  depth_index may be NULL (crop n reads image n).  The sums are accumulated
 * as fixed-point integers inside the kernel: a crop's result is bit-reproducible and does not depend on the
 * launch shape. */
int shr_data_to_model_parts(int N, int H, int W);
/* centre_stride: floats between consecutive centres (3, or 4 to read the rasterizer's (x, y, z, r) records in place). */
int shr_data_to_model_partial(const float *depth, const int32_t *depth_index,
                              const float *centres, int centre_stride, const float *radii,
                              int N, int J, int H, int W, int parts, float *loss_parts,
                              float *grad_parts, void *stream);

/* The same loss in TWO STEPS (round 4), for callers that compare one observed image with several sphere sets (the V*V
 * view pairs of mesh/multiview_utility.py:98-105 share V images) -- or that just want the faster path:
 *   shr_data_to_model_compact      every image of depth[M,H,W] once: its foreground pixels (<= 99) as 8-byte (v << 16 | u, depth)
 *                                  records sorted by 16 x 16-pixel tile, into `workspace`
 *                                  (shr_data_to_model_points_bytes(M, H, W) bytes, 16-byte aligned; 0 = image not
 *                                  taken: W % 4 != 0 or more than 2^28 pixels -- use the calls above);
 *   shr_data_to_model_from_points  crop n (sphere set n) against the points of image depth_index[n] (or n; then
 *                                  M == N): loss_parts [N*parts], grad_parts [N*parts][J][3] (may be NULL) exactly
 *                                  as shr_data_to_model_partial returns them, 1 <= parts <= 64.
 * The per-point search is the same code (csrc/d2m_search.h) with a tighter -- still conservative -- bound; the sums
 * are fixed-point integers: for parts = 1 the results are bit-identical to shr_data_to_model's and bit-reproducible.
 * (With parts > 1 a part owns every parts-th group of 256 points of the image's list, whose order follows the arrival
 * of the compaction's workgroups: the PARTIAL sums then vary from run to run, their total only to a float rounding.) */
long long shr_data_to_model_points_bytes(int M, int H, int W);
int shr_data_to_model_compact(const float *depth, int M, int H, int W, void *workspace, void *stream);
int shr_data_to_model_from_points(const void *workspace, int M, const int32_t *depth_index,
                                  const float *centres, int centre_stride, const float *radii,
                                  int N, int J, int H, int W, int parts, float *loss_parts,
                                  float *grad_parts, void *stream);
/* The same with crop n searching with record set centre_index[n] of `centres` (NULL: n): the V same-view pairs of a
 * [B,V,V] batch of projections without gathering their records first (MutualProjectionLoss with is_mv = False,
 * mesh/multiview_utility.py:107-127). */
int shr_data_to_model_from_points_indexed(const void *workspace, int M, const int32_t *depth_index,
                                          const int32_t *centre_index, const float *centres, int centre_stride,
                                          const float *radii, int N, int J, int H, int W, int parts,
                                          float *loss_parts, float *grad_parts, void *stream);

/* The same search in another LAUNCH ORDER: `order` is a permutation of the N record sets, workgroup w searches for set
 * order[w] against image depth_index[w] (the caller permutes that array alike) and writes that set's slots -- the
 * results of shr_data_to_model_from_points bit for bit.  What it is for: XCD placement -- workgroups w, w + 8, ... share
 * an L2, and the V pairs of a multi-view batch that read one image's list belong on one of them, one after the other. */
int shr_data_to_model_from_points_ordered(const void *workspace, int M, const int32_t *depth_index,
                                          const int32_t *order, const float *centres, int centre_stride,
                                          const float *radii, int N, int J, int H, int W, int parts,
                                          float *loss_parts, float *grad_parts, void *stream);

/* Fused render-and-compare: the model->data term of mesh/multiview_utility.py:98-101 and
 * :107-113 (MSELoss(BallRender(...).min(), observed)) with its whole backward, one
 * launch: e = raster(spheres[n]) - target[target_index ? target_index[n] : n],
 *   sse_partial[n*R + r]              = sum of e*e over region r of crop n,
 *   grad_spheres_partial[(n*R+r)*J+j] = d(sum e*e)/d(x, y, z, radius) of sphere j,
 *   depth[N,H,W] (optional, may be NULL) = the rendered depth.
 * R = shr_sphere_raster_mse_regions(H, W) row regions per crop (1 up to 128x128; 0 = image
 * rows too wide for the kernel); the caller sums the R partials (fixed order: deterministic)
 * and applies MSELoss's 1/(N*H*W) and its weights.  Requires W % 4 == 0 and 16-byte
 * aligned spheres / target / depth / grad buffers (SHR_EINVAL otherwise: compose
 * shr_sphere_raster_fwd and _bwd).  Results equal that composition: bit-identical depth,
 * gradients and sums to fp32 summation order. */
int shr_sphere_raster_mse_regions(int H, int W);
int shr_sphere_raster_mse(const float *spheres, int N, int J, int H, int W,
                          const float *target, const int32_t *target_index,
                          float *depth, float *sse_partial,
                          float *grad_spheres_partial, void *stream);
/* The same over a SELECTION of the batch: workgroup n (0 <= n < N) renders crop c = crop_index[n] (NULL: n) -- records
 * spheres[c], observed image target_index[c] (or c), depth[c] when depth is given -- and reports into slot n of
 * sse_partial / grad_spheres_partial ([N][R], [N][R][J][4]).  The same-view pairs of MutualProjectionLoss with
 * is_mv = False (mesh/multiview_utility.py:107-113) out of the [B,V,V] projections, no gather in front. */
int shr_sphere_raster_mse_indexed(const float *spheres, const int32_t *crop_index, int N, int J, int H, int W,
                                  const float *target, const int32_t *target_index, float *depth,
                                  float *sse_partial, float *grad_spheres_partial, void *stream);

/* shr_sphere_raster_mse in another LAUNCH ORDER: `order` is a permutation of the N crops, workgroup w renders crop
 * order[w] and writes that crop's slots of depth / sse_partial / grad_spheres_partial -- the same results bit for bit;
 * the caller places the crops that compare against one observed image on one XCD (workgroups w, w + 8, ...). */
int shr_sphere_raster_mse_ordered(const float *spheres, const int32_t *order, int N, int J, int H, int W,
                                  const float *target, const int32_t *target_index, float *depth,
                                  float *sse_partial, float *grad_spheres_partial, void *stream);

/* CollisionLoss and BoneLengthLoss (mesh/render.py:145-206) on M samples of J sphere centres (sample m at
 * joints + m*sample_stride floats, [J][3]) with their gradients, one launch.  Collision pairs: spheres
 * 0..num_palm-1 (palm) against every finger sphere, and finger spheres of different fingers (finger f =
 * spheres num_palm + f*per_finger ...); hinge relu(min_dist_sq - |c_a - c_b|^2).  Bone pairs (bone_a[k],
 * bone_b[k]), k < K: relu(bone_min_sq[k] - d^2) and relu(d^2 - bone_max_sq[k]).  Outputs per sample: the three
 * sums [M] and the three UNIT gradients [M][J][3]; the caller applies the reference's sum / means and weights. */
int shr_pair_losses(const float *joints, long long sample_stride, int M, int J, int num_palm, int per_finger,
                    float min_dist_sq, const int32_t *bone_a, const int32_t *bone_b,
                    const float *bone_min_sq, const float *bone_max_sq, int K,
                    float *coll_sum, float *bone_lo_sum, float *bone_hi_sum,
                    float *grad_coll, float *grad_bone_lo, float *grad_bone_hi, void *stream);

/* MultiviewConsistencyLoss (mesh/multiview_utility.py:138-167, hm_weight = None) with its gradient: cam [B,V,4,4]
 * (canonical = R joint + t, t in COLUMN 3), joints [B,V,J,3], V <= 8.  loss_sum[b] = sum over views, joints and
 * coordinates of (canonical - median over the views)^2 with torch.median's lower median; grad_joints [B,V,J,3]
 * (may be NULL) = d loss_sum[b] / d joints, the median's gradient routed to the view it was taken from.  The
 * caller applies MSELoss's 1/(B*V*J*3) and the weight. */
int shr_mv_consistency(const float *cam, const float *joints, int B, int V, int J, float *loss_sum,
                       float *grad_joints, void *stream);

/* DepthResample.forward (network/util_modules.py:10-43) on N scaled depth crops: pixels whose uniform[N][H][W]
 * draw exceeds sample_ratio become 1.0, then the module's fixed 3x3 or 5x5 Gaussian with zero padding
 * (out must not alias depth). */
int shr_depth_resample(const float *depth, const float *uniform, int N, int H, int W, float sample_ratio,
                       int kernel_size, float *out, void *stream);

/* Soft-argmax read-out of the network's heat-maps (network/util_modules.py:164-201), forward and
 * backward: hm[N][2J][h][w] fp32 with element (n, c, y, x) at n*stride_n + c*stride_c + (y*w + x)*stride_px
 * (NCHW: stride_c = h*w, stride_px = 1; channels-last: stride_c = 1, stride_px = 2J).  Channels 0..J-1 are
 * the uv heat-maps, J..2J-1 the depth heat-maps.  xyz[N][J][3] = ((u - cx)/fx, (v - cy)/fy, d * depth_scale_inv)
 * with u, v the softmax(20 * uv) expectation of the pixel grid and d the relu-normalised depth.  The backward
 * writes grad_hm in the input's layout (same strides).  _supported: the 2J maps of a sample fit LDS. */
int shr_soft_argmax_supported(int J, int h, int w);
int shr_soft_argmax_fwd(const float *hm, long long stride_n, long long stride_c, long long stride_px,
                        int N, int J, int h, int w, float cx, float cy, float fx, float fy,
                        float depth_scale_inv, float *xyz, void *stream);
int shr_soft_argmax_bwd(const float *hm, long long stride_n, long long stride_c, long long stride_px,
                        int N, int J, int h, int w, float cx, float cy, float fx, float fy,
                        float depth_scale_inv, const float *grad_xyz, float *grad_hm, void *stream);

/* Pointwise tail of the synthetic branch (network/util_modules.py:104-122), forward only.
 * shr_heatmap_paint: HeatmapRender.forward (mesh/render.py:226-248) for BJ = B*J key-points
 *   uvd[BJ][4] = (u, v, depth, w): uv_hm = uv_scale * exp(-0.5*sigma*((u-uj)^2+(v-vj)^2)) on the S x S
 *   grid, d_hm = d_scale * (depth where the Gaussian > 0.05, else 0), and xyz[BJ][4] = K^-1 uvd
 *   (mesh/pointTransformation.py:102-124; a00,a03 / a11,a13 = the non-trivial entries of rows 0 / 1
 *   of K^-1).
 * shr_depth_noise: DepthNoise.forward (network/util_modules.py:46-84): per pixel a rounded source
 *   shift and depth noise on foreground pixels from normal3[3][B][H][W] standard-normal draws
 *   (out must not alias depth). */
int shr_heatmap_paint(const float *uvd, int BJ, int S, float sigma, float uv_scale, float d_scale,
                      float a00, float a03, float a11, float a13,
                      float *uv_hm, float *d_hm, float *xyz, void *stream);
int shr_depth_noise(const float *depth, const float *normal3, int B, int H, int W,
                    float sigma_xy, float sigma_z, float *out, void *stream);

/* HandSynthesizer.forward (network/util_modules.py:104-122) as THREE launches, capturable in a hipGraph -- no host
 * random numbers, no torch launches in between:
 *   shr_synth_pose_fwd        forward kinematics (shr_fk_fwd) + RandScale (mesh/pointTransformation.py:128-148:
 *                             diag(s) T, s_k = rand * rand_scale + 0.90 - rand_scale / 2) + the sample's draws
 *                             draws[6][B] = s_x, s_y, s_z, the focal jitter rand * 0.2 + 0.9 (:110), and the two uint32
 *                             keys of the sample's pixel-noise stream (as float bits);
 *   shr_mesh_render_post_fwd  DepthRender (shr_mesh_render_fwd) + `* depth_scale` + DepthNoise (:46-84) in the
 *                             rasterizer's epilogue; advances the generator's call counter;
 *   shr_heatmap_render_fwd    Hand3DHeatmapRender (mesh/render.py:274-279): key-point skinning + heat-map camera
 *                             (shr_lbs_project on the key-point CSR table kp_start[J+1], kp_bone, kp_wv) + shr_heatmap_paint.
 * Random numbers: rng_state[3] = (seed, call counter, ticket) in device memory; a draw is a hash of (seed, counter, sample,
 * word) and, per pixel, of (sample key, pixel index) -- csrc/common.h rng_*, restated in numpy by
 * spherehand_amd/synth_rng.py.  Parity with the reference's torch generator is in distribution (as for any RNG-
 * dependent step); with the noise off and the same draws the images are the reference chain's bit for bit.
 * One launch at a time per rng_state: launches that share a state must be ordered (one stream, or events) -- the
 * counter is read by every workgroup and advanced by the last one to finish. */
int shr_synth_pose_fwd(const float *params, int B, const float *offset, const float *offset_inv,
                       const unsigned long long *rng_state, float rand_scale, float *T, float *draws, void *stream);
int shr_mesh_render_post_fwd(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                             const int32_t *skin_bone, const float *skin_wv, int right_hand, float cx, float cy,
                             float fx, float fy, const float *rand_f, const int32_t *faces, int F, int src_size,
                             int S, float clamp_max, float depth_scale, const uint32_t *noise_keys,
                             float sigma_xy, float sigma_z, unsigned long long *rng_state, float *vertices_ws,
                             float *depth_ws, float *depth, void *stream);
int shr_mesh_render_one_launch(int NB, int NV, int F, int src_size, int S);   /* 1: no workspaces needed by the two above */
/* ... and as ONE launch where shr_hand_synth_one_launch() returns 1 (integer resize ratio with a lattice of at most
 * 128 x 128 sampled pixels -- S = 128, 64, 32 from 640 --, NB = 17, J <= 64 key-points, a power-of-two heat-map side):
 * every workgroup runs its crop's forward kinematics + RandScale + draws, skins, rasterizes, scales, noises and paints its
 * heat-maps.  The same bits as the three entries above.  rng_state: uint64 [3] = (seed, call counter, ticket = 0 between
 * launches); noise: 0 / 1; uv_hm = NULL: depth only.  draws[6][B] receives the samples' draws. */
int shr_hand_synth_one_launch(int NB, int NV, int F, int src_size, int S, int J, int hm);
int shr_hand_synth_fwd(const float *params, int B, const float *offset, const float *offset_inv,
                       unsigned long long *rng_state, float rand_scale, int NV,
                       const int32_t *skin_vertex_start, const int32_t *skin_bone, const float *skin_wv,
                       int right_hand, float cx, float cy, float fx, float fy, const int32_t *faces, int F,
                       int src_size, int S, float clamp_max, float depth_scale, int noise, float sigma_xy,
                       float sigma_z, int J, const int32_t *kp_start, const int32_t *kp_bone, const float *kp_wv,
                       int hm, float hcx, float hcy, float hfx, float hfy, float hm_sigma, float uv_scale,
                       float d_scale, float a00, float a03, float a11, float a13, float *draws, float *depth,
                       float *uv_hm, float *d_hm, float *xyz, void *stream);
int shr_heatmap_render_fwd(const float *T, int B, int NB, int J, const int32_t *kp_start, const int32_t *kp_bone,
                           const float *kp_wv, int right_hand, float cx, float cy, float fx, float fy,
                           const float *rand_f, int S, float sigma, float uv_scale, float d_scale, float a00,
                           float a03, float a11, float a13, float *uv_hm, float *d_hm, float *xyz, void *stream);

/* y = relu(GroupNorm(x)) for channels-last (NHWC) fp32 activations x[N][HW][C], forward and
 * backward: the `F.relu(self.bnK(x))` pairs of network/hourglass.py:28-31 without the NCHW
 * round trips of torch's GroupNorm.  G groups of C/G consecutive channels, biased variance,
 * eps inside the root (torch.nn.GroupNorm).  mean / rstd are [N][G] (saved for the backward).
 * The backward returns dx, PER-SAMPLE partials dgamma_partial / dbeta_partial [N][C] and, when
 * dgamma / dbeta [C] are not NULL, their sums over the samples (in order: deterministic).
 * shr_group_norm_relu_supported: C % 32 == 0 and C/G in
 * {4, 8, 16, 32}; buffers 16-byte aligned. */
int shr_group_norm_relu_supported(int C, int G);
/* pre_bias [C] (may be NULL): the bias of the convolution that produced x, added on the fly (y = relu(GN(x +
 * pre_bias))): the convolution then runs without its bias-add pass and the backward returns that bias's
 * gradient (dpre [C], per-sample partials dpre_partial [N][C]; both NULL when pre_bias is) with dx, instead of
 * a separate reduction over dx. */
int shr_group_norm_relu_fwd(const float *x, const float *pre_bias, const float *gamma, const float *beta,
                            int N, int C, int HW, int G, float eps,
                            float *y, float *mean, float *rstd, void *stream);
int shr_group_norm_relu_bwd(const float *x, const float *pre_bias, const float *dy, const float *gamma,
                            const float *beta, const float *mean, const float *rstd,
                            int N, int C, int HW, int G, float *dx,
                            float *dgamma_partial, float *dbeta_partial, float *dpre_partial,
                            float *dgamma, float *dbeta, float *dpre, void *stream);


/* View-to-view projection of the sphere centres ---------------------------------
 * Replaces MutualTransformation + the projection in MutualProjection.forward
 * (mesh/multiview_utility.py:13-30, :62-72).  cam, inv_cam [B,V,4,4] (row-major,
 * p' = R p + t with t in COLUMN 3, as the reference reads them), joints [B,V,J,3],
 * radii [J].  spheres[B,V,V,J,4]: entry (b,i,j,k) = view-i joint k expressed in
 * view j, with its radius: the sphere records the rasterizer takes (N = B*V*V). */
int shr_mutual_project_fwd(const float *cam, const float *inv_cam, const float *joints,
                           const float *radii, int B, int V, int J, float *spheres,
                           void *stream);
/* The two preparations of MutualProjectionLoss (mesh/multiview_utility.py:90-105) in TWO launches instead of three:
 * shr_mutual_project_fwd (which also clears the point lists' fill counters) followed by the compaction of
 * shr_data_to_model_compact(depth[M,H,W] -> workspace) without its memset.  Same results as the two calls. */
int shr_mv_project_compact(const float *cam, const float *inv_cam, const float *joints,
                           const float *radii, int B, int V, int J, float *spheres,
                           const float *depth, int M, int H, int W, void *workspace, void *stream);
/* Assembly of MutualProjectionLoss (mesh/multiview_utility.py:98-129) and of its backward from the partial results
 * of shr_sphere_raster_mse (sse_part [N][Rm], grad_spheres_part [N][Rm][J][4], N = B*V*V pairs) and of
 * shr_data_to_model_partial (d2m_part [E][Rd], grad_d2m_part [E][Rd][J][3]; E = N pairs when is_mv, else the B*V
 * same-view pairs, entry b*V+i):
 *   loss[0] = 9 MSE + d2m_weight * 9 d2m over all pairs (is_mv), or 3 x the same over the V same-view pairs,
 *   grad_joints[B,V,J,3] (may be NULL) = d loss / d joints (both sphere gradients weighted, added and pulled back
 *   through the detached view transforms).  fp64 accumulation of the scalar in a fixed order: deterministic.
 * is_mv: 1 = all pairs; 0 = the same-view pairs, picked out of sse_part / grad_spheres_part of all N pairs; 2 = the
 * same-view pairs with sse_part [B*V][Rm] / grad_spheres_part [B*V][Rm][J][4] holding those pairs only (entry b*V+i:
 * the caller ran shr_sphere_raster_mse on them alone) -- same values as 0.
 * Limits: V <= 8 (the pairs of a view are summed by a 4- or 8-lane butterfly; SHR_ETOOLARGE beyond -- the reference
 * runs V = 3, mesh/multiview_utility.py:107), 4*B*V*V*J*max(Rm,Rd) below 2^31 (SHR_ETOOLARGE). */
int shr_mv_loss_combine(const float *cam, const float *inv_cam, const float *sse_part,
                        const float *grad_spheres_part, int Rm, const float *d2m_part,
                        const float *grad_d2m_part, int Rd, int B, int V, int J, int H, int W, int is_mv,
                        float d2m_weight, float *loss, float *grad_joints, void *stream);
/* grad_joints[B,V,J,3] = sum_j R(b,i,j)^T grad_spheres[b,i,j,k].xyz (the view
 * transforms are constants: detached at mesh/multiview_utility.py:68). */
int shr_mutual_project_bwd(const float *cam, const float *inv_cam, const float *grad_spheres,
                           int B, int V, int J, float *grad_joints, void *stream);

/* Triangle-mesh depth rasterizer (forward only) -----------------------------------
 * Replaces depth_rasterization.forward(width, height, vertices)
 * (mesh/cuda_kernel/depth_rasterization_cuda.cpp:15-25 ->
 *  depth_rasterization_cuda_kernel.cu:115-134, kernel :18-113).
 * face_vertices[B,F,3,3] = pixel-space (x, y, z) of each face's three vertices;
 * depth[B,H,W] is initialised to 1000 and receives, per covered pixel, the
 * minimum over faces of the 1/z-interpolated depth.  Coverage (back-face cull,
 * per-column spans with their truncation quirks) and arithmetic follow the
 * reference kernel; the result is order independent (integer atomics on the fp32
 * bits).  depth 16-byte aligned; W, H <= 65535 (SHR_ETOOLARGE beyond). */
int shr_tri_raster_fwd(const float *face_vertices, int B, int F, int W, int H,
                       float *depth, void *stream);
/* Same, with the face gather of DepthRasterization.forward (mesh/render.py:308-309)
 * fused: vertices[B,NV,4] (x,y,z,w) + faces[F,3] vertex indices (winding already
 * swapped for the right hand, mesh/render.py:298-300). */
int shr_tri_raster_indexed_fwd(const float *vertices, const int32_t *faces, int B,
                               int NV, int F, int W, int H, float *depth, void *stream);

/* Fused DepthRender back end: raster + clamp(max) + bilinear resize src_size -> S, touching
 * only the source pixels the resize reads.  Replaces the chain
 * DepthRasterizationFunction.apply(640,640,...) / clamp / F.interpolate of
 * mesh/render.py:284-287, :310-311.  vertices[B,NV,4] in src_size pixel space, faces[F,3]
 * (winding already swapped for the right hand).  depth[B,S,S].  Needs S <= src_size
 * (no up-sampling; S == src_size is the clamped raster itself). */
int shr_mesh_depth_fwd(const float *vertices, const int32_t *faces, int B, int NV, int F,
                       int src_size, int S, float clamp_max, float *depth, void *stream);

/* Skinning + orthographic camera -----------------------------------------------------
 * Replaces LinearBlendSkinning.forward (mesh/pointTransformation.py:39-46) and
 * OthographicalProjection.forward (:84-99).  T[B,NB,4,4]; the skin table is CSR by
 * vertex (skin_vertex_start[NV+1], skin_bone[NS], skin_wv[NS,4] = fp32(weight *
 * vertex) as the reference stores it, :31), bones ascending within a vertex.
 * right_hand: negate x (:44-45).  project: 0 = skinned points only; 1 = apply the
 * camera u = fx*x + cx*w, v = fy*y + cy*w (rand_f NULL, :88-89) or
 * u = x*rand_f[b]*fx + cx, ..., w = 1 (rand_f given, :91-97).  out[B,NV,4]. */
int shr_lbs_project(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                    const int32_t *skin_bone, const float *skin_wv, int right_hand,
                    int project, float cx, float cy, float fx, float fy,
                    const float *rand_f, float *out, void *stream);

/* DepthRender.forward (mesh/render.py:315-331: LinearBlendSkinning -> OthographicalProjection -> DepthRasterization ->
 * clamp -> resize) in ONE launch where the resize ratio src_size / S is an integer and the sampled source pixels form a
 * lattice of at most 128 x 128 (S = 128, 64, 32 from 640): every workgroup skins and projects its crop's vertices into
 * LDS (shr_lbs_project's arguments and arithmetic, project = 1) and rasterizes from there (shr_mesh_depth_fwd's
 * arguments).  Any other size: the two launches through vertices_ws[B,NV,4] (16-byte aligned; may be NULL when the
 * caller knows the fused kernel applies -- SHR_EINVAL otherwise).  The images are the same bits either way. */
int shr_mesh_render_fwd(const float *T, int B, int NB, int NV, const int32_t *skin_vertex_start,
                        const int32_t *skin_bone, const float *skin_wv, int right_hand, float cx, float cy,
                        float fx, float fy, const float *rand_f, const int32_t *faces, int F, int src_size,
                        int S, float clamp_max, float *vertices_ws, float *depth, void *stream);

/* Key-point skinning -> sphere records -----------------------------------------------------
 * Replaces, inside HandBallPrimitiveRender (mesh/render.py:65-88), the LinearBlendSkinning of the key-points (each
 * bound to ONE bone with weight 1: mesh/pointTransformation.py:39-46 reduces to p = T[bone[j]] @ wv[j], x -> -x for
 * the right hand) and the cat with the radii:  spheres[B,J,4] = (+-p.x, p.y, p.z, radii[j]).
 * T[B,NB,4,4]; bone[J] (bone of key-point j), wv[J,4] = fp32(weight * key-point) as the reference stores it (:31). */
int shr_keypoint_spheres_fwd(const float *T, int B, int NB, int J, const int32_t *bone, const float *wv,
                             const float *radii, int right_hand, float *spheres, void *stream);
/* Its autograd backward: grad_T[B,NB,4,4] = d<grad_spheres, spheres>/dT (the radii are buffers in the reference and
 * get no gradient, the homogeneous row gets none).  bone_start[NB+1], bone_points[J]: the key-points of every bone
 * (CSR, ascending: the sums run in index order). */
int shr_keypoint_spheres_bwd(const float *grad_spheres, int B, int NB, int J, const int32_t *bone_start,
                             const int32_t *bone_points, const float *wv, int right_hand, float *grad_T, void *stream);

/* Forward kinematics -------------------------------------------------------------------
 * Replaces HandTransformationMat.forward (mesh/kinematicsTransformation.py:157-177)
 * and its autograd backward.  params[B,26] (palm Euler xyz, palm translation, 5 x
 * (abduct, flex1, flex2, flex3)); offset, offset_inv [17,4,4] = the bones' offset
 * matrices and their inverses (16-byte aligned; AFFINE: last row 0 0 0 1, as every bone
 * offset matrix and its inverse is -- the kernels never read that row);
 * T[B,17,4,4] (bones 0 and 1 = palm transform). */
int shr_fk_fwd(const float *params, int B, const float *offset, const float *offset_inv,
               float *T, void *stream);
/* grad_params[B,26] = d<grad_T, T>/d params (analytic reverse mode; the homogeneous row of
 * T is constant and takes no gradient). */
int shr_fk_bwd(const float *params, int B, const float *offset, const float *offset_inv,
               const float *grad_T, float *grad_params, void *stream);

/* Pose -> sphere records in one launch per direction ---------------------------------------
 * shr_fk_fwd followed by shr_keypoint_spheres_fwd without T visiting HBM: the fit chain
 * HandBallPrimitiveRender(HandTransformationMat(pose)) of mesh/render.py:81-88 over
 * mesh/kinematicsTransformation.py:169-177.  Same arithmetic as the two entries, same bits.
 * bone[J] < 17, wv[J,4], radii[J] as for shr_keypoint_spheres_fwd; spheres[B,J,4];
 * T: NULL, or [B,17,4,4] to receive the bone transforms as well. */
int shr_pose_spheres_fwd(const float *params, int B, const float *offset, const float *offset_inv, int J,
                         const int32_t *bone, const float *wv, const float *radii, int right_hand,
                         float *spheres, float *T, void *stream);
/* grad_params[B,26] = d<grad_spheres, spheres>/d params: shr_keypoint_spheres_bwd and
 * shr_fk_bwd in one launch (bone_start[18], bone_points[J]: CSR as there). */
int shr_pose_spheres_bwd(const float *params, int B, const float *offset, const float *offset_inv, int J,
                         const int32_t *bone_start, const int32_t *bone_points, const float *wv,
                         int right_hand, const float *grad_spheres, float *grad_params, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPHEREHAND_HIP_H */
