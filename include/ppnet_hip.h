/* ppnet_hip.h — C ABI of libppnet_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the two data-parallel loops of AdamQLMeng/PPNet.  The reference has no
 * FFI layer of its own (it is pure Python); each entry point below names the reference
 * interface it replaces (paths relative to the reference repo).  INTEGRATION.md shows the
 * ctypes binding a maintainer would add under EDaGe-PP/ and GenNet/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     on that stream and never synchronise;
 *   - return value: PPN_OK or a negative PPN_E_* code; nothing throws or aborts;
 *   - no global mutable state: calls are re-entrant (constant tables are built once, lazily,
 *     under a lock);
 *   - points are (row, col) doubles; obstacles are [col, row, radius] doubles, as in the
 *     reference (EDaGe-PP/Path.py:495, MapGenerate.py:143);
 *   - R must be a multiple of 32 and 32 <= R <= 512.
 */
#ifndef PPNET_HIP_H
#define PPNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPN_OK             0
#define PPN_E_INVALID     -1   /* bad argument (NULL where required, R not a multiple of 32, ...) */
#define PPN_E_HIP         -2   /* a HIP runtime call failed; see ppn_last_hip_error() */
#define PPN_E_UNSUPPORTED -3

/* fixed geometry of the generator (EDaGe-PP/MapGenerate.py:21-23, PathSeg.py:22-23) */
#define PPN_SEGS            10
#define PPN_POLY             5          /* degree-4 coefficients, highest power first */
#define PPN_PATH_POINTS   1000
#define PPN_BOUNDARY_POINTS 1100
#define PPN_DRAWS_PER_SEG 1002          /* [straight flag][1000 samples][end abscissa] */
#define PPN_DRAWS_PER_PATH (1 + PPN_SEGS * PPN_DRAWS_PER_SEG)
#define PPN_MAX_HULL        64
#define PPN_MAX_ISLES       16
#define PPN_MAX_POCKET      64          /* pocket obstacles per path */
#define PPN_POCKET_TRY_CAP 256          /* per-isle cap on set_obstacles iterations */
#define PPN_PLACE_TRY_CAP 4096          /* per-map cap on placement attempts */
#define PPN_MAX_WAYPOINTS 2048          /* extract_path step cap (replaces the 1 s wall-clock timeout) */

/* occupancy grid codes (u8) */
#define PPN_GRID_OBST   0
#define PPN_GRID_FREE 255
#define PPN_GRID_MARK 128               /* start / goal marker squares */

/* per-instance flag bits */
#define PPN_FLAG_POCKET_CAP   1u        /* reference would loop forever (Path.py:478) */
#define PPN_FLAG_PLACE_CAP    2u        /* reference: "Repeated over 1000000 times" (MapGenerate.py:60-62) */
#define PPN_FLAG_EMPTY_ISLE   4u        /* reference raises IndexError (Path.py:529) */
#define PPN_FLAG_HULL_CAP     8u        /* more than PPN_MAX_HULL hull vertices */
#define PPN_FLAG_ISLE_CAP    16u
#define PPN_FLAG_POCKET_FULL 32u        /* more than PPN_MAX_POCKET pocket obstacles */
#define PPN_FLAG_CORRIDOR_PASS 64u       /* informational: an obstacle may touch the corridor, the raster ran its compose pass */
#define PPN_FLAG_POCKET_DRAWS 128u       /* caller-fed pocket draws ran out (pocket_stride too small): the path's pocket
                                            obstacles are not the reference's — call again with a larger buffer
                                            (3 * PPN_POCKET_TRY_CAP * PPN_MAX_ISLES floats always suffice) */

/* ABI version of this header.  ppn_version() returns the PPN_ABI_VERSION the LIBRARY was built with; a caller compares the two
 * before its first real call (ppnet_amd/_lib.py does, INTEGRATION.md shows the check) — argument lists are plain pointers and
 * sizes, so a caller built against another header would link and pass, say, a batch size where a workspace pointer is expected.
 * Bumped whenever an entry point's argument list changes or an entry point is removed: 100 = rounds 1-3; 101 = round 4
 * (ppn_conv3x3_relu_classify2_bf16 gained `partial`); 105 = round 5. */
#define PPN_ABI_VERSION 105
int         ppn_version(void);
const char* ppn_error_string(int code);
int         ppn_last_hip_error(void);   /* hipError_t of the most recent PPN_E_HIP on this thread */

/* Host-side constant: the 4 x 1000 least-squares operator W with Poly[0..3] = W . y for
 * x = arange(1000)/100, degree 4 (replaces the per-call np.polyfit at PathSeg.py:24; the
 * constant coefficient is overwritten with 0 at PathSeg.py:28 so its row is not needed).
 * `out` is a HOST buffer of 4*1000 doubles. */
int ppn_polyfit_table(double* out);

/* ---------------------------------------------------------------------------------------------
 * Stage A — target paths.  Replaces PathGroup.generate (PathGenerate.py:33-50) =
 * Path.generate + Path.draw_boundary + Path.path_obstacles (Path.py:78-98, 318-356, 144-155,
 * 113-142, 157-193, 388-404, 463-537) for n_paths independent paths.
 * Optional outputs may be NULL. */
typedef struct ppn_paths {
    double*   seg_poly;         /* [n][10][5]                                          */
    double*   seg_endpoint;     /* [n][10]                                             */
    double*   seg_rotation;     /* [n][10]   PathSeg.Rotation after Path.transform      */
    double*   seg_translation;  /* [n][10][2]                                          */
    double*   seg_length;       /* [n][10]   optional                                  */
    int32_t*  seg_straight;     /* [n][10]                                             */
    double*   segpoint_world;   /* [n][11][2]                                          */
    double*   pathpoint_world;  /* [n][1000][2]                                        */
    double*   boundary_world;   /* [n][1100][2] optional (Path.BoundaryPoint)          */
    uint32_t* canvas_bits;      /* [n][(2R*2R)/32] optional: pre-rotation corridor canvas, bit = row*2R+col */
    double*   hull_raw;         /* [n][64][2] optional: integer hull before normalisation */
    double*   hull;             /* [n][64][2] Path.ConvexHull after space_normalization */
    int32_t*  hull_n;           /* [n]                                                 */
    double*   rotation;         /* [n]  Path.Rotation (degrees)                        */
    double*   trans_rc;         /* [n][2] (t_row, t_col); Path.Translation = [t_col, t_row] */
    double*   segpoint_image;   /* [n][11][2]  Path.SegPointImage                      */
    double*   pathpoint_image;  /* [n][1000][2] Path.PathPoint after normalisation      */
    uint32_t* space_bits;       /* [n][R*R/32] Path.Space as a bit mask, bit = row*R+col */
    int32_t*  isles;            /* [n][16][2] slice bounds into PathPoint              */
    int32_t*  n_isles;          /* [n]                                                 */
    double*   obstacles;        /* [n][64][3] Path.obstacles as [col,row,r]            */
    int32_t*  n_obstacles;      /* [n]                                                 */
    double*   length;           /* [n] Path.Length                                     */
    int32_t*  straight;         /* [n] path-level is_straight                          */
    uint32_t* flags;            /* [n] PPN_FLAG_*                                      */
    double*   max_step_px;      /* [n] largest distance between consecutive path points (and the last one to
                                   the end point), in pixels; stage B uses it to prove that an obstacle cannot
                                   touch the corridor and skip the compose for it */
    double*   seg_grad;         /* [n][10][2] optional: PathSeg.GradSt, GradEnd (PathSeg.py:43-47)      */
    int32_t*  pocket_draws_used;/* [n] optional: torch.rand draws consumed by set_obstacles           */
} ppn_paths_t;

/* draws        : [n_paths][PPN_DRAWS_PER_PATH] uniform doubles in the fixed layout
 *                [path straight][seg0: flag, 1000 samples, end]...[seg9] or NULL = Philox4x32-10
 *                keyed by (seed, stream PATH, first_path_id + p, draw index);
 * pocket_draws : [n_paths][pocket_stride] uniform floats consumed in torch.rand order by
 *                set_obstacles, or NULL = Philox (stream POCKET). */
int ppn_edage_paths(int32_t n_paths, uint64_t first_path_id, int32_t R, double map_size,
                    double clearance, uint64_t seed,
                    const double* draws, const float* pocket_draws, int32_t pocket_stride,
                    const ppn_paths_t* out, void* stream);

/* Same, with the path-level is_straight flag supplied by the caller (Path(is_straight=...), Path.py:53):
 * force_straight[n] int8, -1 = decide from draw 0 as PathGroup.generate does, 0 / 1 = forced. May be NULL. */
int ppn_edage_paths_ex(int32_t n_paths, uint64_t first_path_id, int32_t R, double map_size,
                       double clearance, uint64_t seed,
                       const double* draws, const float* pocket_draws, int32_t pocket_stride,
                       const int8_t* force_straight, const ppn_paths_t* out, void* stream);

/* Same, with the hull's first vertex supplied by the caller: hull_start[n] int32 = index into the canonical vertex cycle
 * (lexicographically smallest lattice point first, counter-clockwise) at which the cycle is to begin, -1 = canonical; may be
 * NULL.  scipy.spatial.ConvexHull (Qhull, Path.py:392-393) returns the same cycle from an implementation-defined start, and
 * Path.search_isle / Path.set_obstacles (Path.py:463-537) walk the hull edges — and draw torch.rand — in that order: a caller
 * that replays the reference's random streams computes Qhull's start on the host and passes it here (dropin/Path.py). */
int ppn_edage_paths_ex2(int32_t n_paths, uint64_t first_path_id, int32_t R, double map_size,
                        double clearance, uint64_t seed,
                        const double* draws, const float* pocket_draws, int32_t pocket_stride,
                        const int8_t* force_straight, const int32_t* hull_start,
                        const ppn_paths_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage B — maps.  Replaces the body of MapGenerate.generate (MapGenerate.py:48-124) incl.
 * Path.boundary_check (Path.py:100-111), generate_map_randomly (MapGenerate.py:126-151), the
 * obstacle raster (Path.plot_obstacles, Path.py:36-49 — explicit rule) and
 * add_init_end_single (process_map.py:119-145).  Map m uses target path m / placements. */
typedef struct ppn_maps {
    uint8_t*  grid;             /* [n][R][R]  PPN_GRID_* codes                          */
    double*   angle;            /* [n]  degrees, U(-180,180)                           */
    int32_t*  translation;      /* [n][2] as drawn (MapGenerate.py:64-65)              */
    int32_t*  attempts;         /* [n]  placement attempts used                        */
    double*   segpoint;         /* [n][11][2] label                                     */
    double*   pathpoint;        /* [n][1000][2] label, optional                         */
    uint8_t*  accept;           /* [n][K] clearance-filter mask of the K random obstacles, optional */
    double*   obstacles;        /* [n][K+64][3] kept random obstacles then pocket obstacles */
    int32_t*  n_obstacles;      /* [n][2] (total, of which random)                      */
    uint32_t* flags;            /* [n]                                                  */
    double*   records;          /* [n][PPN_RECORD_WIDTH] optional: angle, flags, translation[2], segpoint[11][2] as doubles —
                                   the fixed-size per-instance record the ranks all-gather at the end of a batch
                                   (SURVEY 8e), written by the kernel so that no pack pass precedes the collective */
} ppn_maps_t;
#define PPN_RECORD_WIDTH 26

/* place_draws : [n_maps][3] (angle, t0, t1 uniforms of the ACCEPTED attempt) or NULL = Philox
 *               rejection loop (stream PLACE, draw index 3*attempt+i);
 * obst_draws  : [n_maps][3K] in the reference's draw order (K x, K y, K size) or NULL = Philox. */
int ppn_edage_maps(const ppn_paths_t* paths, int32_t n_paths, int32_t placements,
                   uint64_t first_map_id, int32_t R, double map_size, double obstacles_size,
                   int32_t K, double clearance, uint64_t seed,
                   const double* place_draws, const double* obst_draws,
                   const ppn_maps_t* out, void* stream);

/* The per-map label images written right after generation (process_map.py:148-191), for the n = n_paths*placements
 * maps of a ppn_edage_maps call:
 *   mask_path [n][R][R] u8 : generate_gen_path — every 5th label point with 0 < round(p) < bound set to 255;
 *   mask_space[n][R][R] u8 : generate_seg_space — the target path's Space rotated by -angle and translated, {0,1}.
 * `bound` is the reference's hard-coded 224 (pass R for other resolutions). Either output may be NULL. */
int ppn_label_masks(const ppn_paths_t* paths, const ppn_maps_t* maps, int32_t n_paths, int32_t placements,
                    int32_t R, int32_t bound, uint8_t* mask_path, uint8_t* mask_space, void* stream);

/* The two halves of ppn_edage_maps as separate launches (same arguments): `_place` runs the rejection loop, the label
 * transforms and the clearance filter and writes every output except `grid`; `_raster` turns the obstacle lists it
 * left into `grid`.  ppn_edage_maps == _place then _raster on one stream.  Apart they let the compute-bound half of
 * one batch overlap the store-bound half of another on different streams (double-buffer the ppn_maps_t). */
int ppn_edage_maps_place(const ppn_paths_t* paths, int32_t n_paths, int32_t placements,
                         uint64_t first_map_id, int32_t R, double map_size, double obstacles_size,
                         int32_t K, double clearance, uint64_t seed,
                         const double* place_draws, const double* obst_draws,
                         const ppn_maps_t* out, void* stream);
int ppn_edage_maps_raster(const ppn_paths_t* paths, int32_t n_paths, int32_t placements, int32_t R, int32_t K,
                          const ppn_maps_t* out, void* stream);

/* Path.boundary_check (Path.py:100-111) for n (angle, translation) pairs against one hull.
 * angle_deg[n] is the angle passed by the caller (MapGenerate passes -angle), translation_rc
 * [n][2] is (row, col).  ok[n] u8. */
int ppn_boundary_check(const double* hull, int32_t hull_n, const double* angle_deg,
                       const double* translation_rc, int32_t n, int32_t R, uint8_t* ok,
                       void* stream);
/* same, also returning the transformed hulls hull_out[n][hull_n][2] (second element of the reference's tuple) */
int ppn_boundary_check_ex(const double* hull, int32_t hull_n, const double* angle_deg,
                          const double* translation_rc, int32_t n, int32_t R, uint8_t* ok,
                          double* hull_out, void* stream);

/* The accept loop of MapGenerate.generate_map_randomly (MapGenerate.py:128-143) on its own: draws[n][3K]
 * (K x, K y, K size uniforms), pathpoint[n][1000][2]; accept[n][K] u8, obstacles[n][K][3] kept ones
 * compacted in draw order as [col,row,r], counts[n]. */
int ppn_obstacle_filter(const double* pathpoint, const double* draws, int32_t n, int32_t K, int32_t R,
                        double map_size, double obstacles_size, double clearance, uint8_t* accept,
                        double* obstacles, int32_t* counts, void* stream);

/* add_init_end_single (process_map.py:119-145) on n u8 grids [n][R][R]: PPN_GRID_MARK in the 7x7 squares
 * centred on the rounded init[n][2] and end[n][2] (row, col), clipped. */
int ppn_paint_markers(uint8_t* grid, int32_t n, int32_t R, const double* init, const double* end,
                      void* stream);

/* Obstacle raster rule standing in for plot_obstacles (Path.py:36-49): n_maps grids of R x R,
 * grid = PPN_GRID_OBST where the data point the pixel centre shows in the reference's cropped matplotlib figure
 * (X = (j + 0.5) * 444 / 446.4 + R / 446.4, Y = (i + 0.5) * 330 / 332.64 + 1.16 * R / 332.64) lies in the ellipse a stroked
 * circle (cx, cy, r) inks (semi-axes r + 0.625 R / 446.4 and r + 0.625 R / 332.64), else PPN_GRID_FREE; unfused IEEE double,
 * bit-exact against oracle/edage_np.py disc_raster.  obstacles [n_maps][stride][3] = (col, row, r), counts[n_maps]. */
int ppn_disc_raster(const double* obstacles, const int32_t* counts, int32_t stride,
                    int32_t n_maps, int32_t R, uint8_t* grid, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Planner tail (loop B).  collision_check_circle_edge (process_map.py:383-425) for n_seg
 * segments; segment i belongs to problem prob[i] whose obstacles are
 * obs[obs_off[p] .. obs_off[p+1]) as [ox, oy, size] floats.  hit[n_seg] u8. */
int ppn_collision_segments(const float* s, const float* e, const int32_t* prob, int32_t n_seg,
                           const float* obs, const int32_t* obs_off, float clearance,
                           uint8_t* hit, void* stream);
/* The reference's bounds test hard-codes its 224-pixel maps (process_map.py:384-387: a segment with a negative row or a
 * column above 224 is a collision); ppn_collision_segments keeps that constant.  `_bound` takes the map's resolution
 * instead, for maps of another size (configs 3 and 5: 256 and 512). */
int ppn_collision_segments_bound(const float* s, const float* e, const int32_t* prob, int32_t n_seg,
                                 const float* obs, const int32_t* obs_off, float clearance, float bound,
                                 uint8_t* hit, void* stream);

/* extract_path (process_map.py:293-365) on n heat maps `heat` [n][H][W] float32 already
 * down-sampled; init/end [n][2] doubles in down-sampled coordinates.  wp [n][max_wp][2] doubles,
 * wp_n[n], ok[n].  The 1 s wall-clock timeout becomes the max_wp step cap (<= PPN_MAX_WAYPOINTS).  One wave per problem;
 * the revisit rule runs on a bitmap of visited lattice offsets, and a heat map whose values are exactly k/255 is walked
 * from its 8-bit codes in LDS (any other values: from the float map in memory) — same waypoints either way.
 * Square maps only (H == W, else PPN_E_UNSUPPORTED): the reference's bounds test swaps the axes (process_map.py:318). */
int ppn_extract_paths(const float* heat, int32_t n, int32_t H, int32_t W, const double* init,
                      const double* end, int32_t max_wp, double* wp, int32_t* wp_n, uint8_t* ok,
                      void* stream);

/* Fused neighbourhood attention forward (replaces natten2dqkrpb + softmax + natten2dav behind
 * natten.NeighborhoodAttention2D, SegNet/nat.py:111-120,144).  qkv is the qkv Linear's output viewed as
 * [B][H][W][3][heads][32]; rpb [heads][13][13] float32; out [B][H][W][heads*32] (what `proj` consumes).
 * kernel size 7, head dim 32, dilation >= 1 with H, W >= 7*dilation (the module pads first, as NATTEN does).
 * dtype: 0 = float32, 1 = bfloat16 (float32 accumulation). q is multiplied by `scale` before QK. */
int ppn_na2d_fwd(const void* qkv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W, int32_t heads,
                 int32_t dilation, float scale, int32_t dtype, void* stream);
/* Same for a zero-padded token grid: qkv covers the padded H x W grid (the padded tokens' q/k/v are the qkv bias, as
 * when NATTEN's module pads before its projection); only the Hr x Wr real tokens are queries and out is the cropped
 * [B][Hr][Wr][heads*32] tensor.  Hr <= H, Wr <= W. */
int ppn_na2d_fwd_padded(const void* qkv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W, int32_t Hr,
                        int32_t Wr, int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream);
/* The same result without materialising the padded grid ("virtual padding"): qkv is [B][Hr][Wr][3][heads][32] (real
 * tokens only) and every padded position of the H x W grid has k / v = pad_kv[3][heads][32] — the qkv projection's bias
 * in the activation dtype (zeros when it has none), which is what projecting a zero-padded token yields.  Saves the
 * projection and the reads of the 1.7-3x larger padded grid on DiNAT's dilated layers. */
int ppn_na2d_fwd_vpad(const void* qkv, const void* pad_kv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W,
                      int32_t Hr, int32_t Wr, int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream);

/* Backward of ppn_na2d_fwd (the gradient NATTEN's natten2dqkrpb / natten2dav backward kernels compute; first brick of the
 * training step, GenNet/train.py:93-147, SegNet/mmseg/apis/train.py:67-167).  qkv [B][H][W][3][heads][32] and rpb as in the
 * forward, dout [B][H][W][heads*32]; outputs dqkv (same layout as qkv) and drpb [heads][13][13] float32, both fully WRITTEN (the
 * rpb gradient is summed in a fixed order: bit-reproducible).  Two passes over 8 x 8 regions staged through LDS, the
 * probabilities recomputed in the second; workspace: ppn_na2d_bwd_workspace(...) floats, 16-byte aligned (softmax statistics of
 * every query, 4 floats, + 169 partial sums of drpb per region and head) — workspace_floats is what the caller allocated and is
 * checked.  H, W >= 7 * dilation (the training module pads the tokens itself, as NATTEN's does).  dtype 0 = float32,
 * 1 = bfloat16 (float32 arithmetic). */
int64_t ppn_na2d_bwd_workspace(int32_t B, int32_t H, int32_t W, int32_t heads, int32_t dilation);   /* floats; < 0: invalid shape */
int ppn_na2d_bwd(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* workspace, int64_t workspace_floats,
                 int32_t B, int32_t H, int32_t W, int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream);

/* Fused residual + LayerScale + LayerNorm around the NAT layer's dense ops (SegNet/nat.py:140-153):
 *   a == NULL : y_out = LayerNorm(x)                                   (x_out ignored)
 *   a != NULL : x_out = x + gamma * a  (gamma NULL = 1);  y_out = LayerNorm(x_out) unless y_out is NULL.
 * rows x C row-major, C a multiple of 8 with C/8 a power of two <= 64, or C = 1024, or any C <= 64; w, b, gamma are [C] in the
 * same dtype (0 = float32, 1 = bfloat16; statistics in float32). x_out may alias x. */
int ppn_residual_layernorm(const void* x, const void* a, const void* gamma, const void* w, const void* b, void* x_out,
                           void* y_out, int64_t rows, int32_t C, float eps, int32_t dtype, void* stream);
/* y_out = LayerNorm(x + xoff) with a per-channel float32 offset xoff [C] (NULL = none; float32 whatever dtype x has): for a residual stream whose constant
 * (bias) part is carried outside the tensor — the projections then accumulate straight into x through the GEMM's
 * beta * C term and the residual add needs no pass of its own. */
int ppn_layernorm_offset(const void* x, const float* xoff, const void* w, const void* b, void* y_out, int64_t rows,
                         int32_t C, float eps, int32_t dtype, void* stream);
/* Same, with y_out scattered into a zero-padded token grid: rows = B*Hr*Wr tokens in [B][Hr][Wr] order are written to
 * y_out laid out [B][Hp][Wp][C] (the pad region is left untouched: the caller zero-fills it once). */
int ppn_residual_layernorm_padded(const void* x, const void* a, const void* gamma, const void* w, const void* b,
                                  void* x_out, void* y_out, int64_t rows, int32_t C, float eps, int32_t dtype,
                                  int32_t Hr, int32_t Wr, int32_t Hp, int32_t Wp, void* stream);

/* Bilinear x2 up-sampling of an NHWC tensor x [B][H][W][C] -> y [B][2H][2W][C] with PyTorch's
 * F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) arithmetic (the SETR-UP head's Upsample,
 * SegNet/mmseg/ops/wrappers.py:30-51); relu != 0 applies max(x, 0) to the input first (the ConvModule's ReLU,
 * setr_up_head.py:53-66). C a multiple of 8; dtype 0 = float32, 1 = bfloat16. */
int ppn_upsample2x_nhwc(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t relu,
                        int32_t dtype, void* stream);
/* Same with a per-channel bias [C] added before the ReLU: the preceding convolution then runs without its (BatchNorm-
 * folded) bias and its separate add pass (mmcv ConvModule conv -> bn -> ReLU -> Upsample, setr_up_head.py:56-66). */
int ppn_upsample2x_nhwc_bias(const void* x, const void* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t relu, int32_t dtype, void* stream);
/* y = add + (x resized x2): the FPN's top-down step (uper_head.py:103-108: `laterals[i - 1] += resize(laterals[i], size=prev_shape,
 * mode='bilinear', align_corners=False)`) when the finer level is exactly twice the coarser one.  x [B][H][W][C], add and y
 * [B][2H][2W][C]; add may be y.  The resized value is rounded to the tensor's type before the sum, as the two separate kernels do. */
int ppn_upsample2x_add_nhwc(const void* x, const void* add, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* UPerHead's FPN output assembly (mmseg/decode_heads/uper_head.py:117-127: `resize(fpn_outs[i], size=fpn_outs[0].shape[2:],
 * mode='bilinear', align_corners=False)` for i = 1..3, then `torch.cat(fpn_outs, dim=1)`) in one pass over NHWC tensors:
 * out [B][H0][W0][4 C], channels [l C, (l + 1) C) = level l resized to H0 x W0 (level 0 copied).  x_l is [B][hw[2 l]][hw[2 l + 1]][C]
 * (host array hw[8], no level larger than level 0); PyTorch's upsample_bilinear2d arithmetic per output.  C % 8 == 0; dtype 0
 * float32, 1 bfloat16. */
int ppn_resize_concat4_nhwc(const void* x0, const void* x1, const void* x2, const void* x3, const int32_t* hw, void* out, int32_t B, int32_t C,
                            int32_t dtype, void* stream);
/* The general form: n <= 8 NHWC tensors x[l] [B][hw[2 l]][hw[2 l + 1]][channels[l]] (HOST arrays of pointers / sizes; channels % 8 == 0),
 * each resized (bilinear, align_corners False; a tensor of level 0's size is copied) to level 0's size and written to its channel
 * range of out [B][H0][W0][sum channels].  Also the pyramid pooling module's output, psp_head.py:48-60 + uper_head.py:76-84:
 * `torch.cat([x] + [resize(ppm(x), size=x.shape[2:], mode='bilinear')...], dim=1)` with the pooled maps smaller than x. */
int ppn_resize_concat_nhwc(const void* const* x, const int32_t* hw, const int32_t* channels, int32_t n, void* out, int32_t B, int32_t dtype,
                           void* stream);
/* The pyramid pooling module's pools (psp_head.py:33-38: `nn.AdaptiveAvgPool2d(s)` for each pool scale) of one NHWC tensor
 * x [B][H][W][C] in one launch: y[k] [B][scales[k]][scales[k]][C], k < n <= 4 (HOST arrays), PyTorch's bins
 * (rows floor(i H / s) .. ceil((i + 1) H / s) - 1), summed in float32.  C % 8 == 0. */
int ppn_adaptive_pools_nhwc(const void* x, void* const* y, const int32_t* scales, int32_t n, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                            void* stream);
/* SegNet's input straight from stage B's occupancy codes: img [n_pixels][3] (NHWC) = (rgb - mean3) / std3 with rgb = (255,255,255)
 * for PPN_GRID_FREE, (255,0,0) for PPN_GRID_MARK, (0,0,0) otherwise (process_map.py:120,128; planning_seg.py:12-41).
 * mean3 / std3 are HOST pointers to three floats; n_pixels a multiple of 8. */
int ppn_grid_to_image(const uint8_t* grid, void* img, int64_t n_pixels, const float* mean3, const float* std3, int32_t dtype,
                      void* stream);
/* The segmentor's output tail for two classes (setr_up_head.py:78-80, encoder_decoder.py:76-79,242,257): logits [B][2][h][w]
 * (NCHW) -> bilinear x2 -> bilinear to [Ho][Wo] (both align_corners=False, each rounded to the logits' dtype as the
 * materialised tensors are) -> float32 softmax -> argmax, labels u8 [B][Ho][Wo] in {0,1}. */
int ppn_seg_labels_2class(const void* logits, uint8_t* labels, int32_t B, int32_t h, int32_t w, int32_t Ho, int32_t Wo,
                          int32_t dtype, void* stream);
/* In place x = leaky_relu(x + bias[c], negative_slope) on n elements of an NHWC tensor with C channels (C % 8 == 0;
 * slope 0 = ReLU, 1 = bias only): bias + BatchNorm(folded) + LeakyReLU of GenNet's conv stages (ae_vit.py:17-55) as
 * one pass behind a bias-free library convolution. */
int ppn_bias_act_nhwc(void* x, const void* bias, int64_t n, int32_t C, float negative_slope, int32_t dtype, void* stream);

/* The single-channel 3x3 convolutions at the two ends of GenNet's AE-ViT (ae_vit.py:17-20,58; stride 1, padding 1) as
 * direct kernels, float32 accumulation, weights and bias float32 on the device:
 *   _c1  : x [B][H][W] -> y [B][H][W][Cout] (NHWC), y = leaky_relu(conv(x, w[Cout][1][3][3]) + bias[Cout], negative_slope)
 *          (the BatchNorm folded into w / bias by the caller), Cout in {8,16,24,32};
 *   _to1 : x [B][H][W][Cin] (NHWC) -> y [B][H][W], y = conv(x, w[1][Cin][3][3]) + bias, Cin in {8,16,24,32}. */
int ppn_conv3x3_c1_nhwc(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W,
                        int32_t Cout, float negative_slope, int32_t dtype, void* stream);
int ppn_conv3x3_to1_nhwc(const void* x, const float* w, float bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cin,
                         int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 MFMA kernels (v_mfma_f32_16x16x32_bf16, float32 accumulation; ppnet_amd/csrc/mfma_gemm.h).
 *
 * 3x3 convolution, padding 1, stride 1 or 2, as an implicit GEMM on NHWC bfloat16: x [B][H][W][Cin] -> y [B][Ho][Wo][Cout],
 * y = act(conv(x, w) + bias), act = ReLU when relu != 0.  w is [Cout][3][3][Cin] (the torch weight [Cout][Cin][3][3]
 * permuted to (0, 2, 3, 1)); bias [Cout] float32 (zeros for a bias-free convolution).  Cin % 64 == 0, Cout % 8 == 0.
 * Replaces mmcv's ConvModule conv -> (folded) BatchNorm -> ReLU of the SETR-UP head (setr_up_head.py:53-66) and the bias-free
 * stride-2 convolution of NAT's ConvDownsampler (SegNet/nat.py:48-59). */
int ppn_conv3x3_mfma_bf16(const void* x, const void* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          int32_t Cout, int32_t stride, int32_t relu, void* stream);
/* The head's last stage (setr_up_head.py:78-80 with the 1x1 classifier commuted in front of the last up-sampling):
 * logits[m][c] += sum_n max(conv(x, w)[m][n] + bias[n], 0) * w2[c][n] for the B*H*W pixels m and c = 0, 1.
 * logits [B*H*W][2] float32 must hold the classifier's bias on entry; w2 [2][Cout] float32.  The Cout-channel activation is
 * never written: a workgroup adds the sums of its 256-column block in a fixed order (through LDS) and, when Cout <= 256, onto the
 * logits directly — ppn_conv3x3_relu_classify2_slots(Cout) is then 0 and `partial` may be NULL.  Wider convolutions leave one partial
 * sum per pixel, class and 256-column block in partial [ppn_conv3x3_relu_classify2_slots(Cout)][B*H*W][2] float32 (caller-owned
 * workspace, every element written; 16 bytes per pixel at Cout = 512 — round 4: 32) and a second launch adds the slots to logits in
 * slot order.  No atomics anywhere: bit-reproducible. */
int32_t ppn_conv3x3_relu_classify2_slots(int32_t Cout);
int ppn_conv3x3_relu_classify2_bf16(const void* x, const void* w, const float* bias, const float* w2, float* logits, float* partial, int32_t B,
                                    int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream);
/* GenNet's 24-channel stride-2 stages on MFMA, NHWC bfloat16, BatchNorm folded by the caller, bias + LeakyReLU fused:
 *   transposed == 0 (ae_vit.py:24-36, Conv2d(24, 24, 3, 2, 1)): x [B][H][W][24] -> y [B][H/2][W/2][24];
 *        w [32][224] bfloat16, row co (24..31 zero), column (ky*3+kx)*24 + ci (216..223 zero);
 *   transposed != 0 (ae_vit.py:44-55, ConvTranspose2d(24, 24, 3, 2, 1, output_padding=1)): x [B][H][W][24] -> y [B][2H][2W][24];
 *        w [4][32][96] bfloat16: output parity class (oy & 1) * 2 + (ox & 1), row co, column (dy*2+dx)*24 + ci over the 2x2
 *        input block under the 2x2 output block, zero where the class has no tap (ppnet_amd/gennet.py packs both).
 * bias [32] float32 (24..31 zero). */
int ppn_gennet_conv_s2_bf16(const void* x, const void* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W, float negative_slope,
                            int32_t transposed, void* stream);
/* GenNet's LAST decoder stage fused with its final convolution (ae_vit.py:44-58: ConvTranspose2d(24, 24, 3, 2, 1, output_padding=1) + BN +
 * LeakyReLU, then Conv2d(24, 1, 3, 1, 1)): x [B][H][W][24] bfloat16 -> y [B][2H][2W] bfloat16; the 24-channel tensor at the output
 * resolution never reaches memory.  w / bias / negative_slope: the decoder stage as for ppn_gennet_conv_s2_bf16 (transposed form);
 * w_final [24][9] float32 (input channel, tap ky*3+kx) = the final convolution's weight [1][24][3][3]; bias_final its bias.
 * Bit-identical to ppn_gennet_conv_s2_bf16 followed by ppn_conv3x3_to1_nhwc on bfloat16 tensors. */
int ppn_gennet_dec_final_bf16(const void* x, const void* w, const float* bias, float negative_slope, const float* w_final, float bias_final, void* y,
                              int32_t B, int32_t H, int32_t W, void* stream);
/* GenNet's first convolution fused into its first encoder stage (ae_vit.py:24-36: Conv2d(1, 24, 3, 1, 1) + BN + LeakyReLU, then
 * Conv2d(24, 24, 3, 2, 1) + BN + LeakyReLU; BatchNorms folded by the caller): x1 [B][H][W] bfloat16 -> y [B][H/2][W/2][24] bfloat16, the
 * 24-channel full-resolution tensor in between never reaches memory.  H, W even.
 *   w1 [2][16][32] bfloat16: row = channel (24..31 zero), columns 0..8 = bfloat16(w[c][tap]) (hi), 16..24 = bfloat16(w - hi) (lo),
 *       columns 9 / 25 = hi / lo of the bias (the kernel feeds a constant 1 there), else 0;  b1 [32] float32: unused, kept for the layout;
 *   the first LeakyReLU's slope must be <= 1;
 *   wk2 [2][9][16][32] bfloat16: [co tile][tap ky*3+kx][co row][k-slot], slot 8g + e = input channel 4g + e (e < 4) or
 *       16 + 4g + e - 4 (e >= 4, g < 2; zero for g >= 2);  bias2 [32] float32.  ppnet_amd/gennet.py packs all four. */
int ppn_gennet_first_enc_bf16(const void* x1, const void* w1, const float* b1, const void* wk2, const float* bias2, void* y, int32_t B, int32_t H, int32_t W,
                              float slope1, float slope2, void* stream);
/* The ViT blocks of GenNet's AE-ViT (ae_vit.py:38-42,68-70; vit.py:88-161: pre-LN attention + MLP residual blocks, dim 24, 3 heads,
 * MLP x4, LayerNorm eps 1e-6, erf GELU) as one kernel: x, y [B][N][24] bfloat16 token rows (the NHWC feature map), N <= 1024,
 * N % 8 == 0.  One workgroup per problem holds the residual stream in registers (float32) across all n_blocks blocks and the
 * block's K / V in LDS.  params: [n_blocks][PPN_GENNET_BLOCK_PARAMS] float32, per block in this order:
 *   norm1.weight[24] norm1.bias[24] attn.qkv.weight[72][24] attn.qkv.bias[72] attn.proj.weight[24][24] attn.proj.bias[24]
 *   norm2.weight[24] norm2.bias[24] mlp.fc1.weight[96][24] mlp.fc1.bias[96] mlp.fc2.weight TRANSPOSED [96][24] mlp.fc2.bias[24] */
#define PPN_GENNET_BLOCK_PARAMS 7224
int ppn_gennet_trunk_bf16(const void* x, void* y, const float* params, int32_t B, int32_t N, int32_t n_blocks, void* stream);
/* extract_path's result as a fixed-size polyline per problem (process_map.py:355-359): full[b] = [init[b]] + wp[b][:wp_n[b]] * rate +
 * [end[b]] in a [n][max_wp + 2][2] float64 array (rows beyond the plan: wp * rate, i.e. zeros), counts[b] = ok[b] ? wp_n[b] + 2 : 0.
 * wp [n][max_wp][2] / wp_n / ok are ppn_extract_paths' outputs, init / end the full-resolution states. */
int ppn_assemble_paths(const double* wp, const int32_t* wp_n, const uint8_t* ok, const double* init, const double* end, double rate, int32_t n,
                       int32_t max_wp, double* full, int32_t* counts, void* stream);
/* collision[b] = any consecutive-waypoint segment i < counts[b] - 1 of plan b hits any of its first n_obstacles[b] obstacle rows
 * (process_map.py:491-495 over collision_check_circle_edge, :383-425; same float32 arithmetic as ppn_collision_segments_bound).
 * waypoints [B][M][2] float64; obstacles [B][S][3] rows (ox, oy, size), float32 or float64 (obstacles_f64 != 0). */
int ppn_plan_collision(const double* waypoints, const int32_t* counts, const void* obstacles, int32_t obstacles_f64, const int32_t* n_obstacles,
                       int32_t B, int32_t M, int32_t S, float clearance, float bound, uint8_t* collision, void* stream);
/* GenNet's output -> 8-bit heat map, per sample (GenNet/predict.py:95-102): out[b][i] = (uint8)(((y[b][i] - min_b) / (max_b - min_b)) * 255),
 * float32 arithmetic in that order (bit-identical to the torch composition), y [B][n] float32 (dtype 0) or bfloat16 (1). */
int ppn_heatmap_u8(const void* y, uint8_t* out, int32_t B, int32_t n, int32_t dtype, void* stream);
/* SegNet's first tokenizer convolution (SegNet/nat.py:24-40, Conv2d(3, 64, 3, stride 2, padding 1)) straight from the occupancy
 * codes grid [B][H][W] u8: the input image is a 3-colour palette (see ppn_grid_to_image), so the convolution is a 64 x 28 table
 * times a one-hot column per output pixel.  lut [2][64][32] bfloat16: hi and lo halves of the float32 table
 * L[co][3 * (ky * 3 + kx) + colour] = sum_ci w[co][ci][ky][kx] * image_ci(colour)  (colour 0 free, 1 marker, 2 other), column 27 =
 * bias (hi half only), 28..31 zero.  out [B][H/2][W/2][64] bfloat16.  H even, W % 32 == 0 (else PPN_E_UNSUPPORTED). */
int ppn_tokenizer_conv1_codes_bf16(const uint8_t* grid, const void* lut, void* out, int32_t B, int32_t H, int32_t W, void* stream);
/* SegNet's whole tokenizer (SegNet/nat.py:17-46: Conv2d(3, 64, 3, 2, 1) -> Conv2d(64, 128, 3, 2, 1) -> LayerNorm(128)) from the occupancy
 * codes grid [B][H][W] u8 in one kernel: tokens [B][H/4][W/4][128] bfloat16.  H % 4 == 0, W % 64 == 0 (else PPN_E_UNSUPPORTED).
 *   lut [2][64][32] bfloat16: the palette table of ppn_tokenizer_conv1_codes_bf16 with rows in natural channel order;
 *   w2p [8][9][2][16][32] bfloat16: the second convolution's weight as MFMA A fragments — [output tile nt][tap ky*3+kx][k-step s][row i][slot],
 *       row (nt, i) = output channel (nt>>2)*64 + 16*(i>>2) + 4*(nt&3) + (i&3), slot 8g + e = input channel (2s + (e>>2))*16 + 4g + (e&3);
 *   vec [3][128] float32: the second convolution's bias, the LayerNorm weight, the LayerNorm bias.  ppnet_amd/fused.py packs all three. */
int ppn_tokenizer_codes_bf16(const uint8_t* grid, const void* lut, const void* w2p, const float* vec, void* tokens, int32_t B, int32_t H, int32_t W,
                             float eps, void* stream);
/* The dense half of a 128-channel NAT layer (DiNAT-B level 0) as two token-streaming kernels with the weights resident in LDS
 * (SegNet/nat.py:101-153), bfloat16 token rows, float32 accumulation.  tokens % 16 == 0 (else PPN_E_UNSUPPORTED).
 *   ppn_nat128_ln_qkv_bf16:  qkv[tokens][384] = LN(s + offset) . w[384][128]^T + bias     (norm1 -> attn.qkv; offset, bias may be NULL)
 *   ppn_nat128_ln_mlp_bf16:  s[tokens][128] += GELU(LN(s + offset) . w1[256][128]^T + b1) . w2[128][256]^T   in place
 *                            (norm2 -> mlp.fc1 -> erf GELU -> mlp.fc2 -> residual; fc2's bias is carried by the caller's offset)
 *   ppn_nat128_ln_mlp_add_bf16: the same with final_add [128] float32 (may be NULL) added to the result — the last layer of a level
 *                            gives the residual stream its accumulated constant back here instead of in a separate pass over the tensor
 * offset [128] float32 is the constant part of the residual stream carried outside the tensor (ppn_layernorm_offset). */
int ppn_nat128_ln_qkv_bf16(const void* s, const float* offset, const void* ln_w, const void* ln_b, const void* w, const void* bias, void* qkv,
                           int64_t tokens, float eps, void* stream);
int ppn_nat128_ln_mlp_bf16(void* s, const float* offset, const void* ln_w, const void* ln_b, const void* w1, const void* b1, const void* w2,
                           int64_t tokens, float eps, void* stream);
int ppn_nat128_ln_mlp_add_bf16(void* s, const float* offset, const void* ln_w, const void* ln_b, const void* w1, const void* b1, const void* w2,
                               const float* final_add, int64_t tokens, float eps, void* stream);
/* s[tokens][128] += a[tokens][128] . w[128][128]^T in place, bfloat16 (the output projection + residual of a 128-channel NAT layer,
 * SegNet/nat.py:144-146, LayerScale folded into w by the caller, no bias): one streaming kernel, w resident in LDS.  tokens % 16 == 0. */
int ppn_nat128_proj_add_bf16(void* s, const void* a, const void* w, int64_t tokens, void* stream);

/* Dense projection c[M][N] = epilogue(a[M][K] . w[N][K]^T) on bfloat16 (torch.nn.Linear layout; SegNet/nat.py:62-85,111-120).
 * epilogue: 0 = + bias[n]; 1 = gelu(+ bias[n]) (erf form); 2 = c += (the residual stream accumulates, bias unused);
 * 3 = max(+ bias[n], 0) (a 1x1 ConvModule: convolution + folded BatchNorm + ReLU, mmseg/models/decode_heads/uper_head.py:40-63).
 * K % 64 == 0, K >= 128, N % 8 == 0.  persistent_blocks: 0 = one tile per workgroup; else the number of workgroups (a multiple
 * of 8, normally the CU count) that walk the tiles with the LDS-DMA stream running across tile boundaries (M, N % 256 == 0).
 * Few rows (fewer than 64 tiles of 256 x 256, N % 64 == 0: batches of 1-16 problems) go to a kernel of their own whatever
 * persistent_blocks says: a workgroup per 32 x 64 block of c, its four waves splitting K (csrc/gemm_small.hip). */
int ppn_gemm_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int32_t N, int32_t K, int32_t epilogue,
                  int32_t persistent_blocks, void* stream);

/* Row-statistics partials per row that the accumulating mode of ppn_nat_gemm_bf16 writes for a residual stream of width C (and
 * that the LayerNorm modes expect to read for K = C): one per 128 columns at C = 256 (what ppn_nat_mlp_bf16 emits there too), one per
 * 256 columns for wider streams (round 4: per 128 up to C = 512).  C % 256 == 0; < 0: invalid. */
int32_t ppn_nat_gemm_partials(int32_t C);

/* The dense half of a NAT layer at C = 256 / 512 / 1024 with everything between two projections in the GEMMs' epilogues
 * (SegNet/nat.py:62-85 `Mlp.forward`, :140-153 `NATLayer.forward`; csrc/nat_gemm.hip).  a [M][K], w [N][K] (torch Linear
 * layout), c [M][N], all bfloat16; M, N % 256 == 0, K % 64 == 0.
 *   mode 0: c = LN(a) w0^T + b0 computed from the RAW rows of a: the caller passes w = w0 diag(gamma), bias = b0 + w0 beta,
 *           colsum[n] = sum_k w[n][k] (of the bfloat16 values), and stats_in [partials_in][M][2] = partial (sum, sum of squares) of
 *           every row of a (1 <= partials_in <= 4, summed in order; the LayerNorm is over the K features, eps as given):
 *           c = rstd (a w^T - mean colsum) + bias.  K >= 192.
 *   mode 1: c = gelu(mode 0) (erf form; evaluated through a logistic fit of erf, |error| < 3e-5).
 *   mode 2: c += a w^T + bias IN PLACE, and stats_out [P][M][2], P = ppn_nat_gemm_partials(N), receives per column tile (sum, sum of
 *           squares) of every row of the NEW c over the tile's columns — of the bfloat16 values stored: what mode 0 / 1 of the
 *           next projection reads as stats_in with partials_in = P.  colsum / stats_in unused.
 *   The partials of a row are summed in a fixed tree (lane quarter q takes partial q), not in index order: bit-reproducible, and
 *   equal to any other order to float32 rounding.
 * Behind it since round 5 (K >= 128): ONE kernel, the persistent 256 x 256 core of csrc/mfma_gemm.h with these three epilogues (mode 2
 * reads the old c in its epilogue).  Rounds 3-4's kernels (csrc/nat_gemm.hip: the old c through the matrix pipe against an identity;
 * csrc/nat_gemm128.hip: 128 x 128 tiles) serve K = 64 and the A/B knobs PPNET_NAT_LN=old / PPNET_NAT_ACC=old.  Bit-reproducible (no
 * atomics). */
int ppn_nat_gemm_bf16(const void* a, const void* w, const float* bias, const float* colsum, const float* stats_in, int32_t partials_in,
                      float* stats_out, void* c, int64_t M, int32_t N, int32_t K, int32_t mode, float eps, void* stream);

/* The MLP half of a NAT layer as ONE kernel (SegNet/nat.py:62-85 `Mlp.forward`, :147-153 of `NATLayer.forward`; csrc/nat_mlp.hip):
 *     s += gelu(LN(s) W1^T + b1) W2^T + b2        in place on the residual stream s [M][C] (bfloat16),
 * the hidden activation [M][HID] never written: it is produced and consumed chunk by chunk inside the compute unit.
 * ppn_nat_mlp_supported: 1 when (M, C, HID) is served (C = 256, M % 128 == 0 — a workgroup pass is 4 wave pairs x 32 rows —, 64 <= HID <= 2048, HID % 32 == 0), else 0 — callers fall back to two
 *   ppn_nat_gemm_bf16 calls (modes 1 and 2).
 * ppn_nat_mlp_pack_bf16: wpk [2 * C * HID] bfloat16 <- the two weights in the kernel's streaming order; w1 [HID][C] with the
 *   LayerNorm folded in (w1 = W1 diag(gamma)), w2 [C][HID] with LayerScale folded in (torch Linear layouts).  Once per weight change.
 * ppn_nat_mlp_bf16: hb [HID][2] float32 = (colsum_k w1[h][k] of the bfloat16 values, b1[h] + W1[h] . beta); b2 [C] float32;
 *   stats_out [C / 128][M][2] (or null) receives (sum, sum of squares) of every row of the NEW s per 128 columns, of the bfloat16
 *   values stored — the stats_in of the next ppn_nat_gemm_bf16 mode-0 call.  LayerNorm over the C features with `eps`; erf-GELU
 *   as x (1/2 + x~ q(x~^2)) with x~ = clamp(x, -4.25, 4.25) and q an odd-polynomial fit of (Phi(x) - 1/2) / x of degree 15 in x
 *   (no transcendental instruction; |gelu error| <= 9.2e-5 absolute, three times the 3e-5 of ppn_nat_gemm_bf16 mode 1's logistic
 *   fit — the bound bench.py's PARITY_TOLERANCE is sized against).  Bit-reproducible (no atomics). */
int32_t ppn_nat_mlp_supported(int64_t M, int32_t C, int32_t HID);
int ppn_nat_mlp_pack_bf16(const void* w1, const void* w2, void* wpk, int32_t C, int32_t HID, void* stream);
int ppn_nat_mlp_bf16(void* s, const void* wpk, const float* hb, const float* b2, float* stats_out, int64_t M, int32_t C, int32_t HID,
                     float eps, void* stream);

/* stats[rows][2] = (sum, sum of squares) of every row of the bfloat16 tensor x [rows][C], C % 8 == 0 in [64, 1024]: the
 * stats_in (partials_in = 1) of a level's first projection, whose input no mode-2 GEMM produced. */
int ppn_row_stats_bf16(const void* x, int64_t rows, int32_t C, float* stats, void* stream);

/* PIL.Image.resize(size, BILINEAR) for 8-bit single-channel images, bit-exact: Pillow's ImagingResample
 * (support = max(scale,1), 22-bit fixed-point coefficients, horizontal pass then vertical pass, each rounded
 * to 8 bits).  Used by extract_path's down-sampling (process_map.py:301).  in [n][H][W], tmp [n][H][outW],
 * out [n][outH][outW], all u8. */
int ppn_resize_bilinear_u8(const uint8_t* in, int32_t n, int32_t H, int32_t W, int32_t outH, int32_t outW,
                           uint8_t* tmp, uint8_t* out, void* stream);

/* `count` uniform doubles of Philox4x32-10 stream (seed, stream, instance), draw indices first..first+count-1,
 * exactly as the generator kernels draw them (tests pin this against oracle/philox_np.py). out[count] device. */
int ppn_philox_doubles(uint64_t seed, uint32_t stream_id, uint64_t instance, uint32_t first, int32_t count,
                       double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPNET_HIP_H */
