// d3f_internal.h -- declarations shared by the kernels and the C-ABI layer (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/d3fields_hip.h"

namespace d3f {

constexpr int kBlock = 256;                 // 4 waves of 64 lanes
#ifndef D3F_ROWS_PTS
#define D3F_ROWS_PTS 32                     // points per workgroup of the register-rows kernel (fuse_rows.hip; 16: a build-time experiment)
#endif
constexpr uint32_t kFlagFiniteMaps = D3F_FLAG_FINITE_MAPS;
constexpr uint32_t kFlagXcdRemap = D3F_TUNE_XCD_REMAP;

// One channel map as the kernel sees it (strides in elements, channel stride 1).
struct MapDesc {
    const float *data;
    float *out;      // [n, C]
    float *inter;    // [V, n, C] or nullptr
    int64_t sv, sy, sx;
    int32_t fh, fw, C;
    int32_t vw;        // channel-vector width in floats: 4, 2 or 1
    int32_t lpp_log2;  // log2(lanes per point) in phase B
    int32_t unroll;    // channel vectors per lane per pass: +1..+3 batched loads, -1..-4 load-use per vector
    int32_t pre_slot;  // >= 0: bilinear corner set-up of this map is precomputed per (point, view) in LDS slot pre_slot
    int32_t esize;     // bytes per stored channel: 4 (fp32) or 2 (fp16 storage, widened on load)
    int32_t runs;      // > 0: cell-run gather (gather_map_runs): a lane group walks `runs` consecutive points view by view
    int32_t fold;      // 1: wide map -- on the fast path the view weight and the reciprocal of the view count are folded into
                       //    the four bilinear weights once per (point, view): 4 fma per channel vector
};

struct EvalParams {
    const float *depth, *K, *pose, *pts;
    const uint32_t *order;  // nullptr, or n point indices: the kernel processes points in this order
    const float *grid_x, *grid_y, *grid_z;   // regular-grid mode (pts == nullptr): axis coordinate arrays
    int32_t grid_ny, grid_nz;
    float *out_dist;
    uint8_t *out_valid;
    int64_t n;
    int32_t V, H, W;
    int32_t n_maps;
    int32_t tile_pts;  // points per workgroup
    int32_t lds_pad;   // extra dynamic LDS bytes (occupancy throttle, tuning only)
    int32_t crec_offset;   // byte offset of the precomputed corner records, 16-B aligned
    int32_t n_pre;         // number of maps with precomputed corner records
    int32_t xcd_chunk;     // tiles per XCD-mapping chunk (multiple of 8), 0 = the whole launch
    // lattice walk: the n = walk_nx*walk_ny*walk_nz points form a regular lattice in index space (flat index
    // (ix*ny + iy)*nz + iz); workgroups take bricks of walk_tx x walk_ty x walk_tz points in a blocked order computed
    // from blockIdx alone (no keys, no sort, no index array).  walk_nx == 0: off.
    int32_t walk_nx, walk_ny, walk_nz;
    int32_t walk_tx, walk_ty, walk_tz;
    // channel-sliced launch (fused_eval_sliced_kernel): sl_slices > 0 selects it
    int32_t sl_unit;                   // workgroups (of 32 points) per unit
    int32_t sl_ilv;                    // units an XCD works on at the same time (1: one after the other)
    int32_t sl_slices, sl_lg, sl_vc;   // slices per texel of map 0, log2(lanes per point), views with loads in flight
    int64_t sl_tiles, sl_groups, sl_chunks;   // walk tiles, groups of 4 tiles, chunks of 128 groups
    // LDS texel windows (fused_eval_window_kernel): win_slices > 0 selects it
    int32_t win_slices;        // channel slices of 128 * win_u channels per texel of map 0 (looped inside the workgroup)
    int32_t win_u, win_vc;     // 16-byte vectors per lane (1..4), views with corner reads in flight
    int32_t win_pipe;          // 1 (default): software-pipelined point loop when the view count is 4 or 8
    int32_t win_pool_offset;   // byte offset of the two all-zero slices; the pool follows them
    int32_t win_pool_texels;   // pool capacity in texel slices of 512 * win_u bytes
    int32_t win_occ;           // workgroups per CU the kernel variant is built for (2 / 3 / 4)
    int32_t win_lpp;           // lanes per point in phase B: 32, or 16 (two vectors per lane inside a 512-byte slice)
    int32_t win_sparse;        // 1: the pool holds only the texels the tile's pairs touch (bitmap + ranks), 0: the views' whole rectangles
    int32_t rows;              // 1: the register-rows kernel (fuse_rows.hip): 32 points per workgroup, their fused rows in registers
    int32_t thin_max_views;    // 8 (default): thin maps with 2..8 views are gathered with the views in parallel across lanes
                               // (gather_map_thin); 0 switches that off (D3F_EXP_THIN=-1, tests)
    int32_t runs_occ;      // experiment: waves per SIMD of the (1,8) cell-run kernel variant (4 / 5 / 6)
    // the distance-only pass on a big batch (fuse_direct.hip): a copy of the depth maps in tiles of 4 x 8 pixels (one 128-byte line
    // each), [V][depth_th][depth_tw][8][4]; nullptr: the caller's row-major maps
    const float *depth_tiled;
    int32_t depth_tw, depth_th;
    int32_t dist_variant;  // the distance-only pass: 0 = fused_eval_dist_kernel, 8 = held to eight waves per SIMD with three and more views too, + 16 = the compiler's divisions, + 32 = no tiled depth copy, -1 = the branch of fused_eval_kernel (rounds 1-5)
    int32_t store_policy;  // 2 (default) = fused rows leave as non-temporal stores (nt), 3 = sc1 nt (the window kernel's own form), 1 = sc1, 0 = plain
    uint32_t flags;
    float mu;
    // device-side "this tensor holds a non-finite value" words written by d3f_map_check (depth first, then one per map);
    // n_words == 0: the host's D3F_FLAG_FINITE_MAPS alone decides.  All words zero <=> the exact invalid-view skip is allowed.
    int32_t n_words;
    const uint32_t *words[D3F_MAX_MAPS + 1];
    // device-side choice between the window kernel and the cell-run kernel for a cloud (fuse_common.h: gated_out)
    const uint32_t *gate;  // nullptr: this launch is not gated
    uint32_t gate_min;     // the window side runs iff *gate >= gate_min, the cell-run side iff *gate < gate_min
    int32_t gate_want;     // 1: the window side, 0: the cell-run side
    unsigned long long *exp_stamps;   // experiments builds (D3F_EXP_STAMPS=1): s_memtime stamps of every 64th workgroup's phases; else nullptr
    MapDesc maps[D3F_MAX_MAPS];
};

// LDS bytes in front of the stage buffers: records, cnt/flag/idx, KRt, per-view windows
inline int fused_lds_base(int tile_pts, int V) { return ((tile_pts * V * 24 + tile_pts * 12 + V * 48 + V * 16) + 15) / 16 * 16; }
hipError_t launch_fused_eval(const EvalParams &P, int mode, hipStream_t stream);      // fuse_launch.hip: dispatch on the plan
hipError_t launch_direct(const EvalParams &P, int mode, hipStream_t stream);          // fuse_direct.hip
hipError_t launch_depth_tiles(const EvalParams &P, float *tiled, hipStream_t stream);  // fuse_direct.hip: fills EvalParams::depth_tiled's buffer
inline int64_t depth_tiled_bytes(int V, int H, int W) { return (int64_t)V * ((H + 7) / 8) * ((W + 3) / 4) * 128; }
constexpr int64_t kDistTiledMin = 1LL << 22;      // the distance-only pass tiles the depth maps first from this many points on
hipError_t launch_runs(const EvalParams &P, hipStream_t stream);                      // fuse_runs.hip
hipError_t launch_sliced(const EvalParams &P, hipStream_t stream);                    // fuse_sliced.hip
hipError_t launch_window(const EvalParams &P, hipStream_t stream);                    // fuse_window.hip
hipError_t launch_rows(const EvalParams &P, hipStream_t stream);                      // fuse_rows.hip
constexpr int kGateSamples = D3F_GATE_SAMPLES;            // tiles the probe looks at (evenly spaced over the order)
hipError_t launch_window_gate_probe(const EvalParams &P, uint32_t *gate, int nsamples, hipStream_t stream);
int64_t order_gate_offset(int64_t n);
uint32_t *order_gate_words(void *workspace, int64_t n);      // 4 words at the end of a workspace of order_workspace_bytes(n)

// fuse_backward.hip
struct BackwardParams {
    const float *depth, *K, *pose, *pts;
    const float *grad_dist;                    // [n] or nullptr
    const float *grad_fused[D3F_MAX_MAPS];     // [n, C_k] or nullptr
    float *grad_pts;                           // [n, 3]
    int64_t n;
    int32_t V, H, W;
    int32_t n_maps;
    int32_t tile_pts;
    float mu;
    MapDesc maps[D3F_MAX_MAPS];                // out / inter unused
};
hipError_t launch_fused_backward(const BackwardParams &P, int mode, hipStream_t stream);

// scan_kernels.hip: exclusive prefix sum of uint32 counters (in == out allowed); scratch >= scan_scratch_bytes(n)
int64_t scan_scratch_bytes(int64_t n);
hipError_t launch_exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, void *scratch, hipStream_t s);
int64_t scan_status_words(int64_t n);      // single-launch form (decoupled look-back): zeroed status words, total < 2^30
hipError_t launch_exclusive_scan_lookback_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *status, hipStream_t s);

// order_kernels.hip
int64_t order_workspace_bytes(int64_t n);
hipError_t build_point_order(const float *pts, int64_t n, void *workspace, int64_t workspace_bytes,
                             const uint32_t **order_out, hipStream_t stream, int fine = 0);
const uint32_t *stored_point_order(void *workspace, int64_t n);
// is pts a z-fastest lattice?  out: 3 device int32 (nx, ny, nz), zeros when not
hipError_t launch_lattice_probe(const float *pts, int64_t n, int32_t *out_dims, hipStream_t stream);
// mean L1 step between consecutive points vs between points n/2 apart (out: 2 device floats)
hipError_t launch_point_locality(const float *pts, int64_t n, float *out, hipStream_t stream);
// both probes in one launch; out: D3F_PROBE_WORDS device words, every one written (no clearing needed)
hipError_t launch_points_probe(const float *pts, int64_t n, int32_t *out, hipStream_t stream);

// grid_kernels.hip
hipError_t launch_grid_shell(const float *depth, const float *K, const float *pose, int V, int H, int W, const float *gx,
                             const float *gy, const float *gz, int nx, int ny, int nz, float mu, float dist_thr,
                             int64_t capacity, int64_t *idx_out, unsigned long long *count, void *workspace, hipStream_t s,
                             float *tiled_scratch);
int64_t grid_shell_workspace_bytes(int64_t n);
hipError_t launch_fps(const float *pts, int64_t n, int k, int64_t init_idx, int64_t *out_idx, float *out_maxdist,
                      void *workspace, hipStream_t s);
int64_t fps_workspace_bytes(int64_t n, int dist_bytes);     // n distances + two arrays of per-workgroup maxima

// pcd_kernels.hip
hipError_t launch_backproject(const double *depth, const uint8_t *mask, int H, int W, const double *cam, const double *T,
                              const double *bounds, int64_t capacity, double *out_pts, int32_t *out_pixel, int64_t *count,
                              int64_t *block_counts, hipStream_t s);
hipError_t launch_nearest(const double *a, int64_t na, const double *b, int64_t nb, double *min_dist, int64_t *argmin, hipStream_t s);

// assoc_kernels.hip
hipError_t launch_pcd_to_index(const double *pts, int64_t n, const double *lower, double voxel_size, const int32_t *voxel_num,
                               int32_t *out_index, int32_t *out_voxel, hipStream_t s);
int64_t voxset_capacity(int64_t n1, int64_t n2);
hipError_t launch_voxset_iou(const int32_t *a, int64_t na, const int32_t *b, int64_t nb, int64_t *counts, void *workspace,
                             hipStream_t s);
int64_t voxmean_workspace_bytes(int64_t n);
hipError_t launch_voxel_mean(const double *pts, const double *col, int64_t n, double vs, double *out_pts, double *out_col, int64_t *count,
                             void *workspace, hipStream_t s);
hipError_t launch_erode(const uint8_t *src, int H, int W, int kh, int kw, uint8_t *dst, hipStream_t s);
hipError_t launch_compose_labels(const uint8_t *dets, int n_dets, int64_t n_pix, const int32_t *label_of_det, uint8_t *out, hipStream_t s);
hipError_t launch_mask_gate(const float *mask, int64_t sy, int64_t sx, const float *depth, int H, int W, float lo, float hi,
                            uint8_t *out, hipStream_t s);
hipError_t launch_nonzero_pixels(const uint8_t *img, int H, int W, int64_t capacity, int32_t *out_rc, int64_t *count,
                                 int64_t *block_counts, hipStream_t s);
hipError_t launch_fps_pixels(const int32_t *pts, int64_t n, int k, int64_t init_idx, int64_t *out_idx, double *out_maxdist,
                             void *workspace, hipStream_t s);

// misc_kernels.hip
hipError_t launch_map_check(const void *data, int V, int fh, int fw, int C, int64_t sv, int64_t sy, int64_t sx, int esize,
                            uint32_t *word, hipStream_t s);
bool map_is_flat(const void *data, int V, int fh, int fw, int C, int64_t sv, int64_t sy, int64_t sx);
hipError_t launch_map_check_many(const void *const *data, const int64_t *nbytes, const int *esize, uint32_t *const *words, int n,
                                 bool words_are_zero, hipStream_t s);
hipError_t launch_onehot2instance(const float *onehot, int64_t n, int NI, uint8_t *out, hipStream_t s);
hipError_t launch_instance2onehot(const uint8_t *inst, int64_t n, int NI, uint8_t *out, hipStream_t s);

// corr_kernels.hip
struct ColStat {   // running softmax statistics of one column (16 B)
    float m;       // max of -d*scale
    float s;       // sum exp(x - m)
    int64_t arg;   // row index of the first maximum
};
constexpr int kSoftmaxRowsPerBlock = 64;     // = the row tile of pairwise_dist_kernel, whose epilogue emits the statistics
hipError_t launch_dist_to_target(const float *src, int64_t B, int64_t inner, int C, int64_t sb, int64_t si,
                                 int64_t sc, const float *tgt, int dist_type, float *out, hipStream_t s);
hipError_t launch_pairwise_dist(const float *src, const float *tgt, int64_t B1, int64_t B2, int C,
                                int dist_type, float *out, hipStream_t s, ColStat *ws = nullptr, float stat_scale = 1.0f,
                                bool direct_only = false);
hipError_t launch_exp_neg_scale(float *x, int64_t n, float scale, hipStream_t s);
// softmax(-x*scale, dim=0) of a row-major [rows, cols] matrix in place (+ optional argmax)
hipError_t launch_softmax_dim0(float *x, int64_t rows, int64_t cols, float scale, int64_t *argmax_out,
                               ColStat *ws, bool have_stats, hipStream_t s);
// row-sharded softmax: local column statistics, cross-rank merge, normalisation with merged statistics
hipError_t launch_softmax_local_stats(const float *x, int64_t rows, int64_t cols, float scale, int64_t row_offset,
                                      ColStat *ws, ColStat *stats_out, bool have_stats, hipStream_t s);
hipError_t launch_softmax_merge(const ColStat *parts, int64_t nparts, int64_t cols, ColStat *merged, int64_t *argmax_out,
                                hipStream_t s);
hipError_t launch_softmax_apply(float *x, int64_t rows, int64_t cols, float scale, const ColStat *merged, hipStream_t s);
// argmin over dim 0 of raw distances (used for D3F_SIM_DIST + argmax_out)
hipError_t launch_argmin_dim0(const float *x, int64_t rows, int64_t cols, int64_t *arg_out, ColStat *ws,
                              bool have_stats, hipStream_t s);

// k smallest entries per column of a [rows, cols] distance matrix (k <= 8), ties -> lower row, NaN last
int64_t topk_workspace_bytes(int64_t rows, int64_t cols);
hipError_t launch_topk_select(const float *dist, int64_t rows, int64_t cols, void *workspace, const void **final_list, hipStream_t s);
hipError_t launch_topk_write(const void *final_list, const float *x, int64_t rows, int64_t cols, int k, int64_t *idx_out,
                             float *val_out, hipStream_t s);

hipError_t launch_topk_merge_parts(const int64_t *pidx, const float *pval, int64_t n_parts, int k, int64_t cols, int64_t *out_idx,
                                   float *out_val, hipStream_t s);

// track_kernels.hip: the closed-form parts of the rigid-tracking optimiser step
hipError_t launch_rigid_transform(const float *last, int I, int n, const float *t, const float *w, float eps, float *out_pts,
                                  float *norms, hipStream_t s);
hipError_t launch_track_loss_grad(const float *feats, const float *src, const float *dist, const uint8_t *valid, int N, int C,
                                  float dist_w, float *grad_feats, float *grad_dist, float *loss, hipStream_t s);
hipError_t launch_rigid_update(const float *last, int I, int n, const float *grad_pts, float *t, float *w, float *adam_m, float *adam_v,
                               float *step, const float *norms, float eps_rot, float reg_w, float lr, float beta1, float beta2,
                               float eps_adam, hipStream_t s);

struct TrackStepParams {
    const float *depth, *K, *pose;      // views
    int32_t V, H, W;
    MapDesc map;                        // the descriptor map (fp32, C % 4 == 0)
    const float *last;                  // [I, n, 3]
    const float *src;                   // [I*n, C]
    int32_t I, n;
    float mu, dist_w, reg_w, lr, beta1, beta2, eps_adam, eps_rot;
    double ln_beta1, ln_beta2;          // ln of the decimal numbers beta1 / beta2 stand for (log_of_decimal): Adam's bias corrections in double
    float *t, *w, *adam_m, *adam_v, *step;     // [I,3] [I,3] [I,6] [I,6] [I]
    float *out_pts;                     // [I*n, 3] the keypoints as evaluated in this step
    float *grad_pts;                    // [I*n, 3] scratch
    float *loss_acc;                    // [4] two slots of two accumulators (steps alternate), zero between launches
    unsigned long long *par;            // [I*6] (value, step tag) words: the parameters a multi-step launch publishes
    float *loss_out;                    // [3] feature loss, distance loss, regulariser of this step
    unsigned int *counter;              // [2] arrivals (the second word is spare); zero between launches
    int32_t iters;                      // optimiser steps in this launch (> 1: all I*n waves resident, <= kTrackMaxResident)
};
constexpr int kTrackMaxResident = 512;       // one wave per SIMD at this kernel's register count = 1024 on the chip; half of it

hipError_t launch_track_step(const TrackStepParams &P, hipStream_t s);
double log_of_decimal(float beta);
int track_run_capacity();          // keypoints one d3f_track_run launch may hold on the current device

}  // namespace d3f
