/*
 * d3fields_hip.h -- C ABI of libd3fields_hip.so (MI355X / gfx950).
 *
 * The reference (WangYixuan12/d3fields) has no native/FFI layer: its hot path is the Python
 * method surface of `class Fusion` (fusion.py:202) plus utils/corr_utils.py, executed as
 * torch ops.  This header is the boundary a maintainer binds instead of those torch ops;
 * each entry point names the reference lines it replaces.  INTEGRATION.md shows the ctypes
 * stub that goes into fusion.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the device the stream belongs to, unless noted;
 *   - the library never allocates, frees or retains caller memory, reads no environment variable and keeps no
 *     state between calls (re-entrant), with two documented thread-local exceptions: the text behind
 *     d3f_last_error() and the one-shot event pair armed by d3f_profile_next_eval() (a measurement
 *     hook, consumed by the same thread's next query); work is enqueued on `stream` (a
 *     hipStream_t, NULL = default stream) and NOT synchronised;
 *   - all tensors fp32, C-contiguous unless strides are part of the signature;
 *   - return value: D3F_OK or a negative D3F_ERR_* code; d3f_last_error() gives the text of
 *     the calling thread's last failure.  Nothing aborts.
 */
#ifndef D3FIELDS_HIP_H
#define D3FIELDS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3F_ABI_VERSION 6

#define D3F_OK 0
#define D3F_ERR_INVALID_ARG (-1)  /* null pointer, negative count, bad enum               */
#define D3F_ERR_BAD_SHAPE (-2)    /* V/H/W/C/fh/fw outside the supported range             */
#define D3F_ERR_BAD_DTYPE (-3)    /* channel map dtype not supported by this entry point   */
#define D3F_ERR_BAD_LAYOUT (-4)   /* stride/alignment the kernels cannot address           */
#define D3F_ERR_HIP (-5)          /* a HIP runtime call or kernel launch failed            */
#define D3F_ERR_WORKSPACE (-6)    /* workspace missing or too small                        */

#define D3F_MAX_VIEWS 64
#define D3F_MAX_MAPS 8

#define D3F_DTYPE_F32 0
#define D3F_DTYPE_F16 1 /* IEEE half STORAGE of a channel map (a data format of the producer side: DINOv2 run in   \
                           fp16, fusion.py:203,227,616).  Texels are widened to fp32 on load and every operation of \
                           the query stays fp32, so the result equals the fp32 query on the widened map bit for bit  \
                           at half the texel traffic.  Strides stay in ELEMENTS.                                    */

/* d3f_eval flags */
#define D3F_FLAG_FINITE_MAPS 1u /* caller has verified that depth and every channel map hold   \
                                   only finite values: views that are invalid for a point are \
                                   then skipped instead of multiplied by 0 (identical results; \
                                   without the flag 0*NaN / 0*Inf propagate as in the          \
                                   reference, fusion.py:385).  Device-side alternative without \
                                   a host sync: d3f_map_check + the `nonfinite` words below.   */
#define D3F_FLAG_REFERENCE_ROUNDING 4u /* every operation in the reference's own order, also for wide maps.  By default a   \
                                          wide map (more than 256 bytes per texel) takes, for finite operands, the FOLDED  \
                                          form: the view weight and the reciprocal of the view count are multiplied into   \
                                          the four bilinear weights once per (point, view) and the channels run four fma   \
                                          per view -- within a few ulp of the reference's order      \
                                          (1e-5 relative is the contract, ~2e-7 measured).  'dist', 'valid_mask', thin     \
                                          maps (instance masks, colours) and '<k>_inter' are in the reference's order and  \
                                          bit-exact either way.  Slower: the direct gather, no view skipping.              */
#define D3F_FLAG_UNORDERED_POINTS 2u /* the caller's point order has no spatial locality (shuffled or \
                                        uniformly random cloud; see d3f_point_order_locality): walk the \
                                        points in Hilbert order even when the maps are small.  Performance \
                                        only -- results never depend on it.                              */
#define D3F_FLAG_LOCAL_POINTS 128u /* (ABI 6) the caller's point order HAS spatial locality (d3f_points_probe: the mean step between \
                                      consecutive points is a small fraction of the cloud's extent -- a mesh, a scan, the surface    \
                                      points of a lattice in flat-index order): a cloud too small for the window kernel (< 262 144   \
                                      points) on maps that fit the Infinity Cache (<= 256 MiB) keeps the caller's order -- the      \
                                      Hilbert sort (five launches) costs more than it saves there.  Performance only.            */
#define D3F_FLAG_REUSE_POINT_ORDER 8u /* `workspace` still holds the point order that an earlier d3f_eval call wrote  \
                                         for the SAME pts / n (a static grid queried every frame): do not rebuild it   \
                                         (~0.12 ms per 1 M points).  Any permutation of 0..n-1 gives the same results; \
                                         a buffer that holds anything else is the caller's error.                      */

/* Tuning bits of `flags` (performance experiments; results never depend on them):
 *   bits 8..11  log2 of the points per workgroup (2..8), 0 = automatic
 *   bit  12     invert the default XCD mapping (default: XCD-contiguous tile ranges on the Hilbert walk,
 *               round-robin otherwise)
 *   bit  13     never reorder points, even when a workspace is supplied
 *   bit  14     always reorder points when a workspace is supplied
 *   bit   4     D3F_TUNE_DIRECT_GATHER: the plain direct gather in the chosen point order -- no LDS texel windows, no cell
 *               runs, no channel slices, thin maps view by view.  Every fast path is bit-identical to it (tests/).
 *   bit   5     D3F_TUNE_NO_WINDOW_GATE: a cloud never gets the gated pair of launches (LDS texel windows / cell runs chosen by a
 *               device-side probe, ABI 5): cell runs only.  bit 6  D3F_TUNE_WINDOW_SIDE: the gate always opens the window side.
 *   bits 24..25 ignored (round 1: the cell size of the point order; the ordering sizes its cells itself);  bit 26 / 27 force batched / load-use corner loads;
 *               bit 28 do not precompute corner set-ups in phase A;  bits 29..31 XCD-mapping chunk = 1024 << (k-1) tiles
 *   bits 16..23 extra dynamic LDS per workgroup in KiB (throttles workgroups per CU); 255 = none      */
#define D3F_TUNE_TILE_LOG2(k) (((uint32_t)(k) & 0xFu) << 8)
#define D3F_TUNE_XCD_REMAP (1u << 12)
#define D3F_TUNE_NO_REORDER (1u << 13)
#define D3F_TUNE_FORCE_REORDER (1u << 14)
#define D3F_TUNE_DIRECT_GATHER (1u << 4)
#define D3F_TUNE_NO_WINDOW_GATE (1u << 5)
#define D3F_TUNE_WINDOW_SIDE (1u << 6)
#define D3F_TUNE_LDS_PAD_KIB(k) (((uint32_t)(k) & 0xFFu) << 16)

/* Calibrated views: the part of Fusion.curr_obs_torch read by every query
 * (fusion.py:210-215, 707-712): 'depth' (V,H,W), 'K' (V,3,3), 'pose' (V,3,4) world->camera. */
typedef struct d3f_views {
    int32_t V, H, W;
    const float *depth; /* [V,H,W]                                                     */
    const float *K;     /* [V,3,3]                                                     */
    const float *pose;  /* [V,3,4]                                                     */
    const uint32_t *depth_nonfinite; /* NULL, or the device word d3f_map_check wrote for `depth` (ABI 4, see below) */
} d3f_views;

/* One channels-last per-view map: curr_obs_torch['dino_feats'|'mask'|'color_tensor'|...]
 * (fusion.py:373 passes these as .permute(0,3,1,2) views; here the channels-last storage is
 * addressed directly).  Element (v,y,x,c) lives at data[v*stride_v + y*stride_y + x*stride_x + c]
 * (strides in ELEMENTS; the channel stride must be 1). */
typedef struct d3f_channel_map {
    const void *data;
    int32_t fh, fw, C;
    int32_t dtype; /* D3F_DTYPE_F32 or D3F_DTYPE_F16 */
    int64_t stride_v, stride_y, stride_x;
    const uint32_t *nonfinite; /* NULL, or the device word d3f_map_check wrote for this map (ABI 4, see below) */
} d3f_channel_map;

/* Device-side replacement of the caller's `torch.isfinite(map).all()` (the precondition of D3F_FLAG_FINITE_MAPS):
 * one streaming pass over the V x fh x fw x C elements of `map` (16-byte loads, no temporaries) that leaves
 * *word_out != 0 iff the map holds a NaN or an Inf.  Enqueued on `stream`, NO host synchronisation: hand the word to the
 * queries through d3f_channel_map::nonfinite (for the depth images: describe them as a one-channel map {depth, H, W, 1,
 * F32, H*W, W, 1} and pass the word as d3f_views::depth_nonfinite).  A query whose depth and maps ALL carry a word reads
 * the words on the device and takes the exact invalid-view skip iff all are zero -- the same results as with / without
 * D3F_FLAG_FINITE_MAPS set by a host that looked, but capturable in a HIP graph and without the ~2 ms of ATen kernels +
 * host sync per new 1.9 GB map.  A caller that rewrites a map in place re-runs the check (on the same stream order).
 * word_out: one device uint32, 4-byte aligned; the call overwrites it. */
int d3f_map_check(const d3f_channel_map *map, int32_t V, uint32_t *word_out, void *stream);

/* ABI 5.  The same for n <= D3F_MAX_MAPS + 1 tensors in ONE launch (the per-frame refresh of a tracking loop checks depth +
 * features + mask: one launch instead of six): maps[k] with views[k] views leaves its verdict in *words_out[k].
 * D3F_CHECK_WORDS_ARE_ZERO: the caller hands over words that are zero already (fresh slots of a zeroed buffer), which saves
 * the clearing launch; without it the call zeroes them first.  A tensor that is not one contiguous 16-byte aligned block is
 * checked by its own d3f_map_check launch inside the call. */
#define D3F_CHECK_WORDS_ARE_ZERO 1u
int d3f_map_check_many(const d3f_channel_map *maps, const int32_t *views, int32_t n, uint32_t *const *words_out, uint32_t flags,
                       void *stream);

/* ---- library ---------------------------------------------------------------------- */
int d3f_abi_version(void);
const char *d3f_version(void);    /* "d3fields-hip <semver> gfx950" (host memory)          */
const char *d3f_last_error(void); /* host memory, valid until the thread's next failure    */
/* 1 when the library was compiled with -DD3F_EXPERIMENTS (tuning sessions: D3F_EXP_* environment variables select
 * kernel variants); 0 for the product build, which reads no environment variable at all. */
int d3f_build_has_experiments(void);

/* ---- fused field query --------------------------------------------------------------
 * Replaces Fusion.eval (fusion.py:305-394) together with project_points_coords
 * (fusion.py:32-55) and interpolate_feats (fusion.py:57-77); one launch covers any N, so it
 * also replaces the 60 000-point chunk loop + torch.cat of Fusion.batch_eval (fusion.py:526-545).
 *   pts        [n,3]
 *   maps       n_maps descriptors (host array), one per entry of `return_names`
 *   out_dist   [n]      'dist'        out_valid [n] (0/1 bytes) 'valid_mask'
 *   out_fused  host array of n_maps device pointers, [n,C_k] each
 *   out_inter  NULL, or host array of n_maps device pointers (entries may be NULL),
 *              [V,n,C_k] each: the reference's '<k>_inter' (return_inter=True, fusion.py:389)
 * The [n,C_k] rows are written once with non-temporal stores (they do not stay in the GPU's caches); like any kernel output
 * they are visible to whatever follows the call in `stream` order.
 */
int d3f_eval(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps,
             int32_t n_maps, float mu, uint32_t flags, float *out_dist, uint8_t *out_valid,
             float *const *out_fused, float *const *out_inter, void *workspace,
             int64_t workspace_bytes, void *stream);

/* Optional device scratch for d3f_eval (NULL/0 is always accepted).  With at least
 * d3f_eval_workspace_bytes(n) bytes the library may walk the points in a cache-friendlier
 * (Hilbert) order when the maps are much larger than the caches or the caller declares the points
 * unordered; outputs are unaffected. */
int64_t d3f_eval_workspace_bytes(int64_t n);
/* (ABI 6) Scratch for the DISTANCE-ONLY query (n_maps == 0; reference: batch_eval(pts, return_names=[]), vis_repr.py:93): with at
 * least this many bytes of workspace d3f_eval first copies the depth maps into tiles of 4 x 8 pixels (one cache line each) and looks
 * the nearest pixels up there -- consecutive query points of a grid column are neighbours along an image column, four cache lines per
 * four lanes in a row-major map.  0: the batch is too small for that to pay (< 2^22 points) or has more than 8 views.  Outputs are
 * unaffected; the workspace may be reused as soon as the call's work in `stream` is done. */
int64_t d3f_eval_dist_workspace_bytes(const d3f_views *views, int64_t n);

/* ABI 5.  For a CLOUD of >= 262 144 points in the Hilbert order on a patch-resolution wide map d3f_eval enqueues TWO fused
 * launches -- the LDS texel-window kernel and the cell-run kernel -- behind one device word: a probe kernel counts, over
 * D3F_GATE_SAMPLES evenly spaced 64-point tiles of the order, those whose touched texels fit the window kernel's pool, and the
 * workgroups of the side that lost return at once (no host sync; which kernel suits a cloud depends on its density against the
 * texel grid, which only the device knows).  The word lives in the workspace: uint32 at byte offset d3f_eval_gate_offset(n);
 * after the call has completed, a value >= D3F_GATE_MIN_FIT means the window kernel ran.  Diagnostics only. */
#define D3F_GATE_SAMPLES 128
#define D3F_GATE_MIN_FIT 96
int64_t d3f_eval_gate_offset(int64_t n);

/* Measurement hook: the calling thread's NEXT d3f_eval / d3f_eval_dist records these two hipEvent_t
 * (NULL = none) on its stream immediately before and after the fused kernel launch, i.e. around the
 * dominant kernel only (not the optional point-ordering kernels).  One-shot. */
void d3f_profile_next_eval(void *start_event, void *stop_event);

/* What d3f_eval would launch for these shapes (no device work; usable without a GPU): the launch
 * geometry and the per-map lane mapping the host logic picked.  For tests and tuning. */
typedef struct d3f_eval_plan {
    int32_t tile_points;                    /* query points per 256-thread workgroup                  */
    int32_t reorder;                        /* 1: points are walked in Hilbert order (sorted keys); 2: closed-form brick
                                               walk of a lattice (d3f_eval_grid / d3f_eval_lattice)    */
    int32_t lds_bytes;                      /* dynamic LDS per workgroup                              */
    int32_t reserved;                       /* cell-run gather: waves per SIMD its kernel variant is built for; channel-sliced
                                               launch: 100 + 10*log2(lanes per point) + views in flight; LDS texel-window
                                               kernel: 2000 + 100*U + 10*VC + W (its template arguments); else 0 */
    int64_t workgroups;
    int32_t vector_floats[D3F_MAX_MAPS];    /* 4 / 2 / 1 floats per load                              */
    int32_t lanes_per_point[D3F_MAX_MAPS];
    int32_t vectors_per_lane[D3F_MAX_MAPS];
    int32_t staged[D3F_MAX_MAPS];           /* 0: direct gather; 3: LDS texel windows per brick (patch-resolution wide
                                               map on a lattice); 16 + K: cell-run gather with runs of K points
                                               (patch-resolution wide maps); 5: the fused rows of a 32-point brick in
                                               registers (1024-channel patch maps, round 6)            */
    int32_t gated_window;                   /* ABI 5.  1: a cloud that gets the gated pair of launches; the fields above describe
                                               the cell-run side, the window side is the lattice's window plan on 64-point tiles */
    int32_t reserved2;                      /* gated_window: the window side's `reserved` code (2000 + 100*U + 10*VC + W)       */
    int32_t family;                         /* ABI 5.  row of the planner's family table that took the query: 0 dist-only, 1 lds-window,
                                               2 cell-runs, 3 channel-sliced, 4 direct, 5 register-rows (d3f_plan_family_name;
                                               csrc/d3f_plan.h)                                                                    */
    int32_t reserved3;
} d3f_eval_plan;
int d3f_eval_plan_query(const d3f_views *views, int64_t n, const d3f_channel_map *maps, int32_t n_maps,
                        uint32_t flags, int32_t have_workspace, int32_t want_inter, d3f_eval_plan *plan);
/* name of a family id ("lds-window", ...; NULL for an id outside the table), and what the row takes, as one line of text */
const char *d3f_plan_family_name(int32_t family);
const char *d3f_plan_family_takes(int32_t family);

/* ---- regular grids and keypoint selection (reference fusion.py:79-88, 1418-1475) ---------------------
 * A grid is given by its three axis coordinate arrays, exactly the `arange(lower, upper, step) + step/2`
 * tensors of create_init_grid (fusion.py:82-84) -- they are tiny, and reading them keeps the points
 * bit-identical to the reference's on any host.  Point (ix,iy,iz) has flat index (ix*ny + iy)*nz + iz
 * ('ij' meshgrid, z fastest), which is also the layout of every per-point output. */
typedef struct d3f_grid {
    const float *x, *y, *z;   /* device arrays of nx, ny, nz floats */
    int32_t nx, ny, nz;
    int32_t reserved;
} d3f_grid;

/* d3f_eval over all nx*ny*nz grid points without materialising them (replaces
 * batch_eval(create_init_grid(...)[0].to(device), ...), vis_repr.py:88-93): saves the 12 B/point read that
 * dominates a distance-only pass.  Outputs as d3f_eval, in flat grid order. */
int d3f_eval_grid(const d3f_views *views, const d3f_grid *grid, const d3f_channel_map *maps, int32_t n_maps,
                  float mu, uint32_t flags, float *out_dist, uint8_t *out_valid, float *const *out_fused,
                  void *stream);

/* d3f_eval for points that the caller has ARRANGED as a regular lattice -- pts[(ix*ny + iy)*nz + iz], e.g. the
 * materialised output of create_init_grid (fusion.py:79-88) handed to batch_eval (vis_repr.py:88-93) -- with n =
 * nx*ny*nz.  Coordinates are read from `pts` exactly as in d3f_eval and the outputs are identical; the dims only let
 * the library walk the index space brick by brick in closed form when the maps are large (no curve keys, no sort, no
 * index array, no workspace).  Any dims whose product is n are correct -- a wrong guess can only cost speed. */
int d3f_eval_lattice(const d3f_views *views, const float *pts, int32_t nx, int32_t ny, int32_t nz,
                     const d3f_channel_map *maps, int32_t n_maps, float mu, uint32_t flags, float *out_dist,
                     uint8_t *out_valid, float *const *out_fused, float *const *out_inter, void *stream);

/* d3f_eval_plan_query for d3f_eval_lattice / d3f_eval_grid launches. */
int d3f_eval_plan_query_lattice(const d3f_views *views, int32_t nx, int32_t ny, int32_t nz, const d3f_channel_map *maps,
                                int32_t n_maps, uint32_t flags, int32_t want_inter, d3f_eval_plan *plan);

/* Is pts[n,3] a z-fastest lattice (every point = (x[ix], y[iy], z[iz]) with strictly increasing axes, flat index
 * (ix*ny + iy)*nz + iz: the layout of create_init_grid, fusion.py:79-88)?  Sixteen workgroups find where the z and y
 * axes restart and 4096 evenly spaced points are compared with the implied axis values.  out_dims: FOUR device int32,
 * the caller zeroes out_dims[3] before the call; afterwards (nx, ny, nz) = out_dims[0..2] is a lattice iff out_dims[0] > 0
 * and out_dims[3] == 0 (axes longer than 65536 are not recognised).  The shim feeds the answer to d3f_eval_lattice,
 * where it only selects the walk order. */
int d3f_lattice_probe(const float *pts, int64_t n, int32_t *out_dims, void *stream);

/* Pre-filter of select_features_* (fusion.py:1430,1444): flat indices of the grid points with
 * valid_mask && |dist| < dist_thr, compacted into idx_out[0..min(count,capacity)) in ASCENDING order (the order
 * of the reference's boolean-mask indexing).  count_out: ONE device int64, the number of survivors (may exceed
 * capacity: then only the first `capacity` indices were stored).  workspace: d3f_grid_shell_workspace_bytes -- rounded up to a multiple of
 * 256 and followed by d3f_eval_dist_workspace_bytes(views, nx*ny*nz) further bytes the depth lookups of a big grid go to a tiled copy
 * (same survivors; optional). */
int64_t d3f_grid_shell_workspace_bytes(const d3f_grid *grid);
int d3f_grid_shell(const d3f_views *views, const d3f_grid *grid, float mu, float dist_thr, int64_t capacity,
                   int64_t *idx_out, int64_t *count_out, void *workspace, int64_t workspace_bytes, void *stream);

/* fps_np (utils/my_utils.py:478-497): k samples of pts[n,3] starting from init_idx, float32 Euclidean distances,
 * first maximum wins -> out_idx[k] (int64, device), out_maxdist (device float, may be NULL).  k may exceed n: like
 * fps_np the selection then continues with index 0 (every distance is 0).  k + 1 small dependent launches on `stream`
 * (one per round over up to 256 workgroups, maxima merged by the next round); workspace: d3f_fps_workspace_bytes(n)
 * bytes of device scratch, 16-byte aligned. */
int64_t d3f_fps_workspace_bytes(int64_t n);
int d3f_farthest_point_sampling(const float *pts, int64_t n, int32_t k, int64_t init_idx, int64_t *out_idx,
                                float *out_maxdist, void *workspace, void *stream);

/* ---- point clouds on the mask side of the path (fp64, like the reference's numpy) ----------------------
 * depth2fgpcd (utils/my_utils.py:522-537) + camera->world transform + boundary crop of
 * aggr_point_cloud_from_data (utils/draw_utils.py:325-413) for one view.  depth [H,W] fp64 (device);
 * mask [H,W] bytes or NULL (NULL: foreground = 0 < depth < 1.5, as in the reference); cam_params = {fx, fy, cx,
 * cy}, cam_to_world = inv(pose) as 16 row-major doubles, bounds = {x_lower, x_upper, y_lower, y_upper,
 * z_lower, z_upper} or NULL -- these three are HOST arrays.  Survivors are written in ascending pixel order
 * (numpy's boolean-mask order): out_pts [capacity,3] fp64, out_pixel [capacity] int32 (row*W + col) or NULL,
 * count_out: one device int64 (may exceed capacity).  workspace: d3f_backproject_workspace_bytes(H,W). */
int64_t d3f_backproject_workspace_bytes(int32_t H, int32_t W);
int d3f_backproject_view(const double *depth, const uint8_t *mask, int32_t H, int32_t W, const double *cam_params,
                         const double *cam_to_world, const double *bounds, int64_t capacity, double *out_pts,
                         int32_t *out_pixel, int64_t *count_out, void *workspace, void *stream);

/* One direction of Fusion.pcd_iou's nearest-neighbour search (fusion.py:731-735): for every point of a[na,3]
 * the Euclidean distance to, and index of, its nearest point in b[nb,3] (first minimum wins), fp64. */
int d3f_pcd_nearest(const double *a, int64_t na, const double *b, int64_t nb, double *min_dist, int64_t *argmin,
                    void *stream);

/* ---- multi-view instance association: the integer steps (fusion.py:118-180, 794-849, 1279-1297, 1539-1606) ----
 * d3f_pcd_to_index = pcd_to_voxel + voxel_to_index of _init_low_level_memory (fusion.py:118-180):
 *   voxel = floor((p - lower) / voxel_size) as int32 (fp64 arithmetic, numpy's cast), index = v0*num[1]*num[2] +
 *   v1*num[2] + v2 in wrapping int32.  pts [n,3] fp64 (device); lower[3], voxel_num[3] are HOST arrays;
 *   out_index [n] int32; out_voxel NULL or [n,3] int32 (pcd_to_voxel's result). */
int d3f_pcd_to_index(const double *pts, int64_t n, const double *lower, double voxel_size, const int32_t *voxel_num,
                     int32_t *out_index, int32_t *out_voxel, void *stream);

/* Fusion.vox_idx_iou (fusion.py:794-799) on two int32 index arrays (duplicates allowed, any values):
 * out_counts[0] = |set(a) & set(b)|, out_counts[1] = |set(a) | set(b)| (two device int64); the reference's
 * triple is (counts[0]/counts[1], n1/counts[1], n2/counts[1]).  workspace: d3f_vox_iou_workspace_bytes(n1, n2)
 * bytes of device scratch (one hash set), only used inside the call. */
int64_t d3f_vox_iou_workspace_bytes(int64_t n1, int64_t n2);
int d3f_vox_idx_iou(const int32_t *idx1, int64_t n1, const int32_t *idx2, int64_t n2, int64_t *out_counts,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* cv2.erode(src, np.ones([kh, kw], np.uint8), iterations=1) on an [H,W] uint8 image (fusion.py:1293: 2x2 on a
 * 0/255 mask; fusion.py:1561: 15x15): minimum over the window anchored at (kw/2, kh/2), out-of-image samples
 * ignored (cv2's default border for erosion).  src != dst. */
int d3f_erode(const uint8_t *src, int32_t H, int32_t W, int32_t kh, int32_t kw, uint8_t *dst, void *stream);

/* Fusion.swap_instance_mask (fusion.py:1052-1063) for ONE view: dets [n_dets, n_pix] uint8 (non-zero = the pixel belongs to the
 * detection; the rows of curr_obs_torch['mask_gs'][view]), label_of_det[n_dets] int32 device = the consensus instance index
 * the detection was assigned to (instances_info[k]['idx'][view] == d  =>  label_of_det[d] = k), < 0 = none.  out[n_pix] uint8 =
 * what painting the detections in instance order leaves: the largest instance index covering the pixel (as uint8), 0 where
 * no assigned detection does. */
int d3f_compose_labels(const uint8_t *dets, int32_t n_dets, int64_t n_pix, const int32_t *label_of_det, uint8_t *out, void *stream);

/* open3d's PointCloud.voxel_down_sample as the reference uses it (utils/draw_utils.py:318-323 voxel_downsample, :396-400
 * inside aggr_point_cloud_from_data; radius 0.01): voxels of side voxel_size anchored at min_bound - voxel_size/2, one
 * output point (and colour) per occupied voxel = the mean of its points.  pts / colors [n,3] float64 (colors may be NULL),
 * out_pts / out_colors [n,3] capacity; *count_out (device int64) = the number of voxels.  Output order: ASCENDING voxel
 * index (x, then y, then z) -- open3d's order is that of its hash map, so the SETS agree (means to ~1e-14 m: the sums are
 * exact 64-bit fixed point, hence deterministic), the order does not.  Axis extent < 2^21 voxels. */
int64_t d3f_voxel_downsample_workspace_bytes(int64_t n);
int d3f_voxel_downsample(const double *pts, const double *colors, int64_t n, double voxel_size, double *out_pts,
                         double *out_colors, int64_t *count_out, void *workspace, int64_t workspace_bytes, void *stream);

/* The pixel side of select_features_rand_v2 (fusion.py:1554-1565), on the device:
 *   d3f_mask_gate       out(y,x) = 255 where mask(y,x) != 0 and depth_lo < depth(y,x) < depth_hi, else 0 -- the reference's
 *                       `mask.astype(bool) & (depth > 0.0) & (depth < 1.5)` as the uint8 image cv2.erode takes.  The mask
 *                       channel is read in place from the channels-last one-hot tensor: element (y,x) at
 *                       mask_channel[y*stride_y + x*stride_x] (ELEMENT strides); depth, out: [H,W] contiguous.
 *   d3f_nonzero_pixels  np.array(image.nonzero()).T: the (row, col) int32 pairs of the nonzero pixels in ascending row-major
 *                       order (order-preserving compaction).  out_row_col [capacity,2]; *count_out (device int64) = the
 *                       number found, also when it exceeds capacity; workspace: d3f_backproject_workspace_bytes(H, W). */
int d3f_mask_gate(const float *mask_channel, int64_t stride_y, int64_t stride_x, const float *depth, int32_t H, int32_t W,
                  float depth_lo, float depth_hi, uint8_t *out, void *stream);
int d3f_nonzero_pixels(const uint8_t *image, int32_t H, int32_t W, int64_t capacity, int32_t *out_row_col, int64_t *count_out,
                       void *workspace, void *stream);

/* fps_np (utils/my_utils.py:478-497) on integer 2-D points, as select_features_rand_v2 calls it on the (row, col)
 * indices of a mask (fusion.py:1565-1566): pts [n,2] int32, exact squared distances (numpy's float64 norms of
 * integer differences order identically), first maximum wins.  out_idx [k] int64, out_maxdist one device double
 * or NULL; workspace: d3f_fps_pixels_workspace_bytes(n) bytes, 16-byte aligned; k + 1 launches like d3f_farthest_point_sampling. */
int64_t d3f_fps_pixels_workspace_bytes(int64_t n);
int d3f_fps_pixels(const int32_t *pts, int64_t n, int32_t k, int64_t init_idx, int64_t *out_idx, double *out_maxdist,
                   void *workspace, void *stream);

/* Gradient of d3f_eval's outputs w.r.t. the query points: what autograd through Fusion.eval gives
 * the reference's rigid_tracking (fusion.py:1643-1665).  grad_dist: [n] or NULL; grad_fused: host
 * array of n_maps device pointers ([n,C_k], entries may be NULL); grad_pts [n,3] is overwritten.
 * Differentiable paths: projection -> bilinear coordinates, exp weight, clamped distance;
 * nearest-depth lookup, validity and the view count carry no gradient (as in torch). */
int d3f_eval_backward(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps,
                      int32_t n_maps, float mu, const float *grad_dist, const float *const *grad_fused,
                      float *grad_pts, void *stream);

/* Locality of the caller's point order, decided on the device without a sort: out[0] = mean L1 step between
 * consecutive points, out[1] = mean L1 distance between points n/2 apart (<= 4096 evenly spaced samples,
 * non-finite pairs skipped).  out: 2 floats of DEVICE memory, written asynchronously on `stream`.
 * A grid / mesh / scan-ordered cloud gives out[0] << out[1]; the shim passes D3F_FLAG_UNORDERED_POINTS when
 * out[0] > 0.25 * out[1] (random clouds on patch-resolution maps: 1.93 -> 0.88 ms per 985 600 points). */
int d3f_point_order_locality(const float *pts, int64_t n, float *out, void *stream);

/* (ABI 6) d3f_lattice_probe AND d3f_point_order_locality in ONE launch, with nothing to clear beforehand: every one of the
 * D3F_PROBE_WORDS device words of `out` is written by the call.  out[0..2] = (nx, ny, nz) of the lattice or zeros; out[8 + b],
 * b < 16: non-zero when sample block b contradicts those dims (a lattice iff out[0] > 0 and all sixteen are zero);
 * out[24 + 3q .. 24 + 3q + 2], q < 4, as FLOATS: sum of the steps between consecutive points, sum of the distances between
 * points n/2 apart, number of finite samples of locality block q (mean step = sum of the first / sum of the third). */
#define D3F_PROBE_WORDS 40
int d3f_points_probe(const float *pts, int64_t n, int32_t *out, void *stream);

/* The same for Fusion.eval_dist: d(dist)/d(pts) = -mean over the valid views of row 2 of K@pose. */
int d3f_eval_dist_backward(const d3f_views *views, const float *pts, int64_t n, const float *grad_dist,
                           float *grad_pts, void *stream);

/* Replaces Fusion.eval_dist (fusion.py:396-436): no -mu gate, no clamp, no 1e3 sentinel. */
int d3f_eval_dist(const d3f_views *views, const float *pts, int64_t n, float *out_dist,
                  uint8_t *out_valid, void *stream);

/* ---- instance-mask helpers ----------------------------------------------------------
 * onehot2instance (fusion.py:109-116): argmax over the last dim -> uint8 (first max wins,
 * NaN counts as maximum, like torch.argmax).  instance2onehot (fusion.py:90-107). */
int d3f_onehot2instance(const float *onehot, int64_t n, int32_t NI, uint8_t *out, void *stream);
int d3f_instance2onehot(const uint8_t *instance, int64_t n, int32_t NI, uint8_t *out_bool,
                        void *stream);

/* ---- descriptor similarity (utils/corr_utils.py) --------------------------------------- */
#define D3F_DIST_L2 0     /* dist_type='l2'     : sqrt(sum (a-b)^2)                        */
#define D3F_DIST_SQUARE 1 /* dist_type='square' : sum (a-b)^2                              */

#define D3F_SIM_DIST 0         /* raw distance            compute_dist_tensor  corr_utils.py:44-61 */
#define D3F_SIM_EXP 1          /* exp(-d*scale)           compute_similarity   corr_utils.py:4-19  */
#define D3F_SIM_SOFTMAX_DIM0 2 /* softmax(-d*scale,dim=0) compute_similarity_tensor :21-42,
                                                          compute_similarity_tensor_multi :63-106 */

/* Device scratch needed by the D3F_SIM_SOFTMAX_DIM0 mode (and by argmax_out) for a
 * [rows, cols] result: per-column running max / sum-exp / argmax of each 64-row chunk. */
int64_t d3f_softmax_workspace_bytes(int64_t rows, int64_t cols);

/* Feature map vs ONE target descriptor.  src holds B*inner descriptors of C channels:
 * descriptor (b,j) channel c at src[b*stride_b + j*stride_i + c*stride_c] (elements), which
 * covers both [B,H,W,C] (compute_similarity) and [B,C,*dim] (compute_*_tensor).
 * out [B*inner].  D3F_SIM_SOFTMAX_DIM0 normalises over b for every j and needs
 * workspace >= d3f_softmax_workspace_bytes(B, inner); other modes accept NULL/0. */
int d3f_similarity_to_target(const float *src, int64_t B, int64_t inner, int32_t C,
                             int64_t stride_b, int64_t stride_i, int64_t stride_c,
                             const float *tgt, float scale, int32_t dist_type, int32_t mode,
                             float *out, void *workspace, int64_t workspace_bytes, void *stream);

/* compute_similarity_tensor_multi (corr_utils.py:63-106): src [B1,C], tgt [B2,C] ->
 * out [B1,B2]; the [B1,B2,C] difference tensor of the reference (and its OOM retry,
 * corr_utils.py:84-94) never exists.  `scale` is ignored by D3F_SIM_DIST.
 * argmax_out: NULL or [B2] int64, the row index of the best match of each target (largest
 * similarity == smallest distance; first wins).
 * workspace: >= d3f_softmax_workspace_bytes(B1, B2) bytes when mode is D3F_SIM_SOFTMAX_DIM0 or
 * argmax_out is given; only read/written inside the call. */
int d3f_pairwise_similarity(const float *src, const float *tgt, int64_t B1, int64_t B2, int32_t C,
                            float scale, int32_t dist_type, int32_t mode, float *out,
                            int64_t *argmax_out, void *workspace, int64_t workspace_bytes,
                            void *stream);

/* k-nearest-descriptor lookup (k <= 8): d3f_pairwise_similarity plus, per target column, the rows of the k SMALLEST
 * distances -- the k-NN generalisation of the reference's best match, compute_similarity_tensor_multi(...).argmax(0)
 * (the reference has no KNN of its own; this is what the north star's "KNN correspondence lookup" maps
 * to).  Neighbours are chosen on the raw distances (ties -> lower row index, NaN last), before exp / softmax can round
 * close values into ties.  topk_idx [k,B2] int64 (row j = the j-th nearest; -1 where B1 < k), topk_val [k,B2] or NULL:
 * the entries of `out` (in `mode`) at those rows.  workspace >= d3f_pairwise_topk_workspace_bytes(B1, B2), 16-B aligned. */
int64_t d3f_pairwise_topk_workspace_bytes(int64_t B1, int64_t B2);
int d3f_pairwise_similarity_topk(const float *src, const float *tgt, int64_t B1, int64_t B2, int32_t C, float scale,
                                 int32_t dist_type, int32_t mode, int32_t k, float *out, int64_t *topk_idx,
                                 float *topk_val, void *workspace, int64_t workspace_bytes, void *stream);

/* The two halves of the k-NN lookup for ROW-SHARDED sources (d3fields_amd/sharding.py: sharded_knn_descriptors):
 *   d3f_topk_smallest  per column of a [rows, cols] matrix (a rank's raw distances) the rows of the k smallest entries (ties ->
 *                      lower row, NaN last): idx_out [k,cols] int64 (-1 where rows < k), val_out [k,cols] or NULL;
 *                      workspace >= d3f_pairwise_topk_workspace_bytes(rows, cols), 16-byte aligned.
 *   d3f_topk_merge     n_parts such lists (indices already GLOBAL rows) -> the k best per column by (value, index);
 *                      parts_idx / parts_val [n_parts, k, cols]; entries with index < 0 are padding. */
int d3f_topk_smallest(const float *x, int64_t rows, int64_t cols, int32_t k, int64_t *idx_out, float *val_out,
                      void *workspace, int64_t workspace_bytes, void *stream);
int d3f_topk_merge(const int64_t *parts_idx, const float *parts_val, int64_t n_parts, int32_t k, int64_t cols,
                   int64_t *out_idx, float *out_val, void *stream);

/* ---- the same softmax with B1 sharded over GPUs -------------------------------
 * softmax(dim=0) of compute_similarity_tensor_multi (corr_utils.py:102) couples all B1 rows.  With the
 * rows split over ranks each rank runs
 *   1. d3f_pairwise_softmax_local : raw distances of ITS rows into out [B1_local,B2] and, per target
 *      column, the running statistics of -d*scale over its rows -> stats [B2]
 *      (argmax = row_offset + local row of the first maximum; B1_local == 0 gives (-inf, 0, INT64_MAX));
 *   2. an all-gather of the 16-B records (the only exchange step; host side, RCCL);
 *   3. d3f_softmax_merge : [n_parts,B2] records in rank order -> merged [B2] (+ global argmax, first wins);
 *   4. d3f_softmax_apply : out = exp(-out*scale - max) / sum in place.
 * workspace of step 1: >= d3f_softmax_workspace_bytes(B1_local, B2). */
typedef struct d3f_col_stat {
    float max_logit; /* max over rows of -d*scale               */
    float sum_exp;   /* sum over rows of exp(-d*scale - max)    */
    int64_t argmax;  /* global row index of the first maximum   */
} d3f_col_stat;

int d3f_pairwise_softmax_local(const float *src, const float *tgt, int64_t B1_local, int64_t B2, int32_t C,
                               float scale, int32_t dist_type, int64_t row_offset, float *out,
                               d3f_col_stat *stats, void *workspace, int64_t workspace_bytes, void *stream);
int d3f_softmax_merge(const d3f_col_stat *parts, int64_t n_parts, int64_t cols, d3f_col_stat *merged,
                      int64_t *argmax_out, void *stream);
int d3f_softmax_apply(float *x, int64_t rows, int64_t cols, float scale, const d3f_col_stat *merged,
                      void *stream);

/* ---- the optimiser step of Fusion.rigid_tracking (fusion.py:1608-1685) ---------------------------------
 * One step of the reference is  so3_exp_map -> rigid transform -> Fusion.eval -> loss -> autograd -> Adam
 * (pytorch3d + ~90 torch launches).  Around d3f_eval / d3f_eval_backward the rest is closed-form:
 *   1. d3f_rigid_transform : per instance i, R_i = so3_exp_map(w_i) (pytorch3d 0.7.5: Rodrigues, angle clamped
 *      at sqrt(1e-4)), out_pts[i,p] = last[i,p] @ R_i + t_i (row-vector convention); norms[0..1] = |t|_F, |w|_F.
 *   2. d3f_eval on out_pts (one channel map: the descriptors).
 *   3. d3f_track_loss_grad : loss[0] = mean(|f - src| * valid), loss[1] = dist_w * mean(max(dist*valid, 0))
 *      (fusion.py:1654-1657) and their gradients w.r.t. f and dist.
 *   4. d3f_eval_backward -> grad_pts.
 *   5. d3f_rigid_update : chain rule through the transform and the exponential map, + reg_w * d(|t|_F + |w|_F)
 *      (fusion.py:1658), then torch.optim.Adam's update of (t_i, w_i); adam_m / adam_v [n_inst,6] and
 *      step [n_inst] (float counters) are the optimiser state, zero at the start of a frame.
 * All arrays are device memory; last [n_inst,n,3], t / w [n_inst,3], valid = the uint8 mask of d3f_eval. */
int d3f_rigid_transform(const float *last, int32_t n_inst, int32_t n, const float *t, const float *w,
                        float *out_pts, float *norms, void *stream);
int d3f_track_loss_grad(const float *feats, const float *src, const float *dist, const uint8_t *valid,
                        int64_t N, int32_t C, float dist_w, float *grad_feats, float *grad_dist,
                        float *loss, void *stream);
int d3f_rigid_update(const float *last, int32_t n_inst, int32_t n, const float *grad_pts, float *t, float *w,
                     float *adam_m, float *adam_v, float *step, const float *norms, float reg_w, float lr,
                     float beta1, float beta2, float eps, void *stream);

/* The same optimiser step as ONE launch: a wave per keypoint transforms it, queries the descriptor field, forms the
 * loss gradient and back-propagates it to the keypoint with the corner texels held in registers; the last wave to
 * finish reduces per instance and steps Adam.  One descriptor map: fp32, C % 4 == 0, C <= 512, 16-byte aligned texels,
 * at most 8 views (otherwise D3F_ERR_BAD_SHAPE / D3F_ERR_BAD_LAYOUT: use the five launches above).  `state` holds device
 * pointers owned by the caller: t / w [n_inst,3] (updated in place), adam_m / adam_v [n_inst,6], step [n_inst] (float
 * counters), out_pts [n_inst*n,3] (the keypoints as evaluated in this step -- what rigid_tracking returns after its last
 * step), loss [3] (feature term, distance term, regulariser of this step) and scratch of
 * d3f_track_step_scratch_bytes(n_inst, n) bytes that must be ZERO before the first step of a frame (every step leaves
 * it zero again).  src [n_inst*n, C]: the instances' source descriptors. */
typedef struct d3f_track_state {
    float *t, *w;
    float *adam_m, *adam_v;
    float *step;
    float *out_pts;
    float *loss;
    void *scratch;
} d3f_track_state;
int64_t d3f_track_step_scratch_bytes(int32_t n_inst, int32_t n);
int d3f_track_step(const d3f_views *views, const d3f_channel_map *descriptors, const float *last, int32_t n_inst, int32_t n,
                   const float *src, float mu, float dist_w, float reg_w, float lr, float beta1, float beta2, float eps,
                   const d3f_track_state *state, void *stream);

/* `iters` optimiser steps (the whole loop of fusion.py:1650-1665 for one frame) in ONE launch: the waves of step k+1 wait
 * inside the kernel for the Adam update of step k (a device-wide arrival counter and step-tagged parameter words in `scratch`),
 * so K / pose / the source descriptors are read once per frame and no launch boundary separates the steps.  Same
 * arithmetic per step as d3f_track_step (same t / w / out_pts / loss as `iters` calls of it).  Every workgroup must be
 * resident while the others wait: n_inst*n <= d3f_track_run_max_keypoints() and n_inst <= 16, else D3F_ERR_BAD_SHAPE -- call
 * d3f_track_step per iteration then.  d3f_track_run_max_keypoints() is derived at run time for the CURRENT device: the
 * occupancy of the kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor) x its compute units, half of that (the other half is
 * left to whatever else runs), at most 512.  The wait is bounded: should a wave never see the next step's parameters (the
 * device held by other kernels for seconds), loss[0..2] become NaN, the launch ends, and t / w / the optimiser state are
 * UNDEFINED.  The kernel also leaves D3F_TRACK_STALL_SENTINEL in the uint32 at word d3f_track_stall_word(n_inst, n) of `scratch`
 * (ABI 5): a NaN loss WITHOUT the sentinel came out of the data, not out of a stall.  A caller that shares the device reads that
 * word, restores its state and repeats the frame with d3f_track_step (d3fields_amd/rigid.py: RigidTracker.run does exactly
 * that).  scratch is cleared at the start of every
 * d3f_track_run (a one-workgroup launch ahead of the step kernel); d3f_track_step expects the scratch its predecessor left. */
#define D3F_TRACK_STALL_SENTINEL 0x57A11EDu
int64_t d3f_track_stall_word(int32_t n_inst, int32_t n);      /* index (in 4-byte words) of the sentinel inside `scratch` */
int d3f_track_run(const d3f_views *views, const d3f_channel_map *descriptors, const float *last, int32_t n_inst, int32_t n,
                  const float *src, float mu, float dist_w, float reg_w, float lr, float beta1, float beta2, float eps,
                  int32_t iters, const d3f_track_state *state, void *stream);
int32_t d3f_track_run_max_keypoints(void);

#ifdef __cplusplus
}
#endif
#endif /* D3FIELDS_HIP_H */
