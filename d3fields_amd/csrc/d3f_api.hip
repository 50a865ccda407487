// d3f_api.hip -- the extern "C" boundary of libd3fields_hip.so (see include/d3fields_hip.h).
// Validates arguments, picks the lane mapping of every channel map and enqueues the kernels on
// the caller's stream.  No allocation, no synchronisation, no state besides the thread-local
// error text.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "d3f_internal.h"

namespace {

thread_local char g_err[512] = "";
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;   // one-shot, see d3f_profile_next_eval

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(D3F_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int check_views(const d3f_views *v)
{
    if (!v) return fail(D3F_ERR_INVALID_ARG, "views is NULL");
    if (!v->depth || !v->K || !v->pose) return fail(D3F_ERR_INVALID_ARG, "views: depth/K/pose must be non-NULL");
    if (v->V < 1 || v->V > D3F_MAX_VIEWS) return fail(D3F_ERR_BAD_SHAPE, "views: V=%d outside [1,%d]", v->V, D3F_MAX_VIEWS);
    if (v->H < 2 || v->W < 2) return fail(D3F_ERR_BAD_SHAPE, "views: H=%d W=%d must be >= 2", v->H, v->W);
    return D3F_OK;
}

// Phase-B lane mapping of one map: vector width, lanes per point (2^k) and vectors per lane.
// Minimises idle lane-slots (passes*lpp*U - cvec), then passes, then prefers wide groups
// (longer contiguous segments per load instruction).
// batch: issue all 4*U corner loads before the first use (best for cache-resident maps, U <= 3);
// otherwise load-use per vector, U <= 4 (best when the map misses the caches).
void pick_mapping(d3f::MapDesc &m, bool can16, bool can8, bool batch, int max_u = 4)
{
    m.vw = (m.C % 4 == 0 && can16) ? 4 : ((m.C % 2 == 0 && can8) ? 2 : 1);
    const int cvec = m.C / m.vw;
    long best_slots = -1;
    int best_passes = 0;
    for (int lg = 6; lg >= 0; --lg) {
        const int lpp = 1 << lg;
        for (int u = (batch ? 3 : 4) < max_u ? (batch ? 3 : 4) : max_u; u >= 1; --u) {
            const int per = lpp * u;
            const int passes = (cvec + per - 1) / per;
            const long slots = (long)passes * per;
            // thin family (max_u == 1: masks, colours): fewest PASSES first -- every pass repeats the per-(point, view)
            // set-up, and <= 4 lanes per point make the map eligible for the views-in-parallel gather (gather_map_thin)
            const bool better = best_slots < 0 || (max_u == 1 ? (passes < best_passes || (passes == best_passes && slots < best_slots))
                                                               : (slots < best_slots || (slots == best_slots && passes < best_passes)));
            if (better) {
                best_slots = slots;
                best_passes = passes;
                m.lpp_log2 = lg;
                m.unroll = u;
            }
        }
    }
    if (!batch) m.unroll = -m.unroll;
}

// ---- launch-plan thresholds (every row has a test in tests/test_abi.py::test_plan_table) ---------------------------------
//  kSmallBatch          fewer points than this: no reordering, no window / cell-run / sliced launch -- the set-up of those
//                       paths costs more than it saves, and small batches are spread over >= 1024 workgroups instead
//  kCacheResidentBytes  all requested maps together at most this big live in the L2s / Infinity Cache anyway: the caller's
//                       order is kept (unless the cloud has no locality at all), 128-point tiles
//  kBatchedLoadBytes    a map at most this big issues all 4*U corner loads of a view before the first use; bigger maps
//                       in caller order use load-use per vector (a smaller in-flight footprint measured faster)
//  kBeyondLlcBytes      maps beyond this in CALLER order without scratch: 64-point tiles at 2 workgroups per CU
//  kWindowCloudMin      a cloud of at least this many points (in the Hilbert order) may take the LDS-window kernel when the device-side
//                       probe finds its tiles compact (below: the window kernel's ~25-us workgroups do not fill the chip twice
//                       over and the cell-run kernel wins: 71 k surface points 0.17 vs 0.12 ms, 100 k keypoints 0.12 vs 0.09)
constexpr int64_t kSmallBatch = 65536;
constexpr int64_t kWindowCloudMin = 262144;
constexpr int kGatedSecondPass = 1;          // eval_common's internal "now enqueue the other side" status (never returned to callers)
constexpr int64_t kCacheResidentBytes = 64LL << 20;
constexpr int64_t kBatchedLoadBytes = 128LL << 20;
constexpr int64_t kBeyondLlcBytes = 512LL << 20;

// Validates one channel map and fills the kernel-side descriptor (out/inter may be NULL for backward).
int fill_map(const d3f_channel_map &c, int s, int V, float *out, float *inter, const float *extra_aligned,
             d3f::MapDesc &m, int64_t &map_bytes, uint32_t flags = 0)
{
    if (!c.data) return fail(D3F_ERR_INVALID_ARG, "map %d: data pointer is NULL", s);
    if (c.dtype != D3F_DTYPE_F32 && c.dtype != D3F_DTYPE_F16)
        return fail(D3F_ERR_BAD_DTYPE, "map %d: dtype %d unsupported (D3F_DTYPE_F32 or D3F_DTYPE_F16)", s, c.dtype);
    const int es = c.dtype == D3F_DTYPE_F16 ? 2 : 4;          // bytes per stored channel
    if (c.fh < 1 || c.fw < 1 || c.C < 1) return fail(D3F_ERR_BAD_SHAPE, "map %d: fh=%d fw=%d C=%d", s, c.fh, c.fw, c.C);
    if (c.stride_x < c.C || c.stride_y < 0 || c.stride_v < 0)
        return fail(D3F_ERR_BAD_LAYOUT, "map %d: strides (%lld,%lld,%lld) do not describe a channels-last map", s,
                    (long long)c.stride_v, (long long)c.stride_y, (long long)c.stride_x);
    if (((int64_t)(c.fh - 1) * c.stride_y + (int64_t)(c.fw - 1) * c.stride_x + c.C) * es >= (1LL << 32))
        return fail(D3F_ERR_BAD_SHAPE, "map %d: one view spans 4 GiB or more (32-bit texel offsets)", s);
    m.data = static_cast<const float *>(c.data);
    m.out = out;
    m.inter = inter;
    m.runs = 0;
    m.pre_slot = -1;
    m.esize = es;
    m.fold = ((int64_t)c.C * es > 256) ? 1 : 0;        // wide map: folded weights on the fast path (fuse_common.h)
    m.sv = c.stride_v; m.sy = c.stride_y; m.sx = c.stride_x;
    m.fh = c.fh; m.fw = c.fw; m.C = c.C;
    if (!aligned(m.data, es) || !aligned(out, 4) || !aligned(extra_aligned, 4))
        return fail(D3F_ERR_BAD_LAYOUT, "map %d: pointers must be aligned to their element size", s);
    // 4-channel vectors: 16 B of fp32 / 8 B of fp16 per load; outputs are fp32 either way
    const bool str4 = (c.stride_v % 4 == 0) && (c.stride_y % 4 == 0) && (c.stride_x % 4 == 0);
    const bool str2 = (c.stride_v % 2 == 0) && (c.stride_y % 2 == 0) && (c.stride_x % 2 == 0);
    const bool can16 = str4 && aligned(m.data, 4 * es) && aligned(out, 16) && aligned(inter, 16) && aligned(extra_aligned, 16);
    const bool can8 = str2 && aligned(m.data, 2 * es) && aligned(out, 8) && aligned(inter, 8) && aligned(extra_aligned, 8);
    const int64_t this_bytes = (int64_t)V * c.fh * c.fw * c.C * es;
    map_bytes += this_bytes;
    if (es == 2) {
        // fp16 storage: 8 channels per 16-B load (fp32 accumulators / 32-B stores), else scalar lanes; batched loads only
        const bool str8 = (c.stride_v % 8 == 0) && (c.stride_y % 8 == 0) && (c.stride_x % 8 == 0);
        const bool vec8 = (c.C % 8 == 0) && str8 && aligned(m.data, 16) && aligned(out, 16) && aligned(inter, 16) &&
                          aligned(extra_aligned, 16);      // backward: grad_fused is read as 32-byte f32x8 pieces of 16-B aligned rows
        const int max_u = m.fold ? 4 : 1;               // thin maps: one vector per lane (see below)
        pick_mapping(m, false, false, true, max_u);     // scalar lanes ...
        if (vec8) {                                     // ... or the same search over 8-channel vectors
            d3f::MapDesc t = m;
            t.C = c.C / 2;                               // cvec = C/8 = (C/2)/4: reuse the 4-wide search
            pick_mapping(t, true, true, true, max_u);
            m.vw = 8; m.lpp_log2 = t.lpp_log2; m.unroll = t.unroll;
        }
        return D3F_OK;
    }
    bool batch = this_bytes <= kBatchedLoadBytes;
    if (flags & (1u << 26)) batch = true;
    if (flags & (1u << 27)) batch = false;
    // thin maps (<= 256 bytes per texel: masks, colours) keep the reference's operation order and get ONE kernel form -- one
    // batched vector per lane; the wide ones have the whole family (and the folded weights).  Keeping the two families apart
    // halves the instantiations of gather_map a kernel carries (with both full families the generic kernels spilled 1 KiB).
    if (!m.fold) batch = true;
    pick_mapping(m, can16, can8, batch, m.fold ? 4 : 1);
    return D3F_OK;
}

// Experiment knobs read from the environment (integers; results never depend on them):
//   D3F_EXP_RUNS   cell-run gather on patch-resolution wide maps: -1 off, 0 automatic (default), 2 / 4 / 8 = run length
//   D3F_EXP_RUNS_U vectors per lane of the cell-run gather: 0 automatic, 1 / 2 / 3;  D3F_EXP_RUNS_OCC=5: the (1,8) variant
//                  held to 5 waves per SIMD
//   D3F_EXP_STORE  -1: write the fused rows with plain stores, 1: with sc1 ones, 3: with `sc1 nt` ones, instead of `nt` ones (store_out, fuse_common.h)
//                  also in the window kernel (plain there by default)
//   D3F_EXP_SLICED 1 / 2 / 3: force the channel-sliced launch for a dense wide map on a lattice (128- / 256- / 512-byte
//                  slices, fuse_eval.hip); -1: never (default: only with thin companion maps); _VC views in flight, _UNIT
//                  workgroups per unit
//   D3F_EXP_WALK_TILE  shape of the walk's tile as digits x y z with the same point count (222 default; 224 with a thin map)
//   D3F_EXP_WALK   lattice brick walk for grids on large maps: -1 off, 0 automatic (default)
//   D3F_EXP_THIN   -1: thin maps (mask, colours) through the view-sequential gather_map instead of gather_map_thin
//   D3F_EXP_WINDOW LDS texel-window kernel instead of the cell-run gather for a patch-resolution wide first map
//                  (fuse_eval.hip): 0 automatic = on lattices (64 points per workgroup), -1 never,
//                  32 / 64 / 128 = always, with that many points per workgroup; _U vectors per lane (1..4), _VC views
//                  in flight (U = 2 / 3), _OCC workgroups per CU (2..4), _POOL pool texels, _LPP 32: one vector per lane (default 16 x 2)
//   D3F_EXP_RUNS_OCC also: 4 = the (2,8) cell-run variant held to 4 waves per SIMD (default 3, spill-free)
#ifdef D3F_EXPERIMENTS
// built with -DD3F_EXPERIMENTS (python -m d3fields_amd.build --experiments): tuning sessions only
int exp_knob(const char *name)
{
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
}
// phase stamps of the window kernel (D3F_EXP_STAMPS=1): 32 x uint64 per sampled workgroup, read back with d3f_exp_read_stamps
constexpr int64_t kStampBytes = 8LL * 32 * 65536;
unsigned long long *exp_stamp_buffer(bool clear)
{
    static unsigned long long *buf = nullptr;
    if (!buf && hipMalloc(reinterpret_cast<void **>(&buf), kStampBytes) != hipSuccess) buf = nullptr;
    if (buf && clear) (void)hipMemset(buf, 0, kStampBytes);
    return buf;
}
#else
// the product build reads no environment: every knob is its default (0), the library keeps no hidden state
constexpr int exp_knob(const char *) { return 0; }
#endif

// Cell-run gather (fuse_eval.hip gather_map_runs): fp32 maps read as 16-byte vectors with >= 32 vectors per texel whose
// texels span >= 4 image pixels -- the patch-resolution feature maps of the reference (fusion.py:694-697).
bool runs_candidate(const d3f::MapDesc &m, int H, int W)
{
    return m.esize == 4 && m.vw == 4 && m.C >= 128 && (W - 1) >= 4 * (m.fw - 1) && (H - 1) >= 4 * (m.fh - 1);
}

// LDS texel windows (fuse_eval.hip fused_eval_window_kernel): a patch-resolution wide map in whole 128-channel slices whose texels
// start on 16-byte boundaries -- fp32 (512-byte slices), or stored in fp16 (256-byte slices, round 5: lattices only)
bool window_candidate(const d3f::MapDesc &m, const d3f_views *views, bool check_pointer)
{
    const int es = m.esize, per16 = 16 / es;            // channels per 16 bytes
    const bool vec = es == 4 ? m.vw == 4 : m.vw == 8;
    // (m.fold: a map of <= 256 bytes per texel -- 128 channels of fp16 -- belongs to the thin family, which keeps the reference's
    //  operation order in every kernel; the window kernel's arithmetic is the folded one)
    return vec && m.fold && m.C >= 128 && m.C % 128 == 0 && (views->W - 1) >= 4 * (m.fw - 1) && (views->H - 1) >= 4 * (m.fh - 1) &&
           (int64_t)views->V * m.sv * es < (1LL << 31) && (m.sx % per16) == 0 && (m.sy % per16) == 0 && (m.sv % per16) == 0 &&
           (!check_pointer || reinterpret_cast<uintptr_t>(m.data) % 16 == 0);
}

// (vectors per lane U, run length K) of the cell-run gather: the built variants are (1,8) (2,4) (2,8) (3,2) (3,4)
void pick_runs_mapping(d3f::MapDesc &m, int U, int K)
{
    const int cvec = m.C / 4;
    // Defaults from the MI355X sweeps (gpurun_out/r2h): 32-lane groups (C = 384) -> one vector per lane,
    // 4-point runs, 69 VGPR = 7 waves per SIMD (C2 patch 0.750 -> 0.633 ms, C3 patch 1.537 -> 1.288); 64-lane groups
    // (C = 1024) -> two vectors per lane x two passes, 8-point runs at 4 waves per SIMD (C4 patch 4.32 -> 3.35).
    const bool auto_u = U <= 0 || U > 3;
    if (auto_u) U = (cvec % 128 == 0) ? 2 : 1;
    long best_slots = -1;
    for (int lg = 6; lg >= 5; --lg) {           // 64 or 32 lanes per point; ties go to the wider group (fewer passes)
        const long per = (long)(1 << lg) * U;
        const long slots = (cvec + per - 1) / per * per;
        if (best_slots < 0 || slots < best_slots) { best_slots = slots; m.lpp_log2 = lg; }
    }
    if (auto_u && U == 2 && m.lpp_log2 != 6) {   // two vectors per lane only pays on full 64-lane groups
        U = 1;
        best_slots = -1;
        for (int lg = 6; lg >= 5; --lg) {
            const long per = (long)(1 << lg);
            const long slots = (cvec + per - 1) / per * per;
            if (best_slots < 0 || slots < best_slots) { best_slots = slots; m.lpp_log2 = lg; }
        }
    }
    if (U == 3 && K != 2) K = 4;
    if (U == 2 && K != 4) K = 8;
    if (U == 1 && K != 4 && K != 8) K = m.lpp_log2 == 5 ? 4 : 8;
    m.unroll = U;
    m.runs = K;
}

// Brick of the lattice one window workgroup takes: T points with power-of-two sides (the kernel decodes a slot with
// shifts), as few padded slots as possible, then as cubic as possible
void pick_window_brick(int nx, int ny, int nz, int T, int &bx, int &by, int &bz)
{
    double best = -1.0;
    bx = by = 1; bz = T;
    for (int x = 1; x <= T; x <<= 1)
        for (int y = 1; x * y <= T; y <<= 1) {
            const int z = T / (x * y);
            const double blocks = (double)((nx + x - 1) / x) * ((ny + y - 1) / y) * ((nz + z - 1) / z);
            const double eff = (double)nx * ny * nz / (blocks * T);
            const int hi = x > y ? (x > z ? x : z) : (y > z ? y : z), lo = x < y ? (x < z ? x : z) : (y < z ? y : z);
            const double score = eff * (1.0 - 0.03 * ((double)hi / lo - 1.0));
            if (score > best) { best = score; bx = x; by = y; bz = z; }
        }
}

int tile_points_for(int V)
{
    // LDS per workgroup = tile*V*24 B (+ small); keep it <= 32 KiB so >= 4 workgroups fit a CU.
    // 128 points measured best (985 600-pt grid, C=384: 128 -> 0.99 ms, 256 -> 1.06 ms patch-res).
    int t = 128;
    while (t > 32 && (long)t * V * 24 > 32 * 1024) t >>= 1;
    return t;
}

int eval_common(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps, int32_t n_maps,
                float mu, uint32_t flags, float *out_dist, uint8_t *out_valid, float *const *out_fused,
                float *const *out_inter, void *workspace, int64_t workspace_bytes, void *stream, int mode,
                d3f_eval_plan *plan_out = nullptr, const d3f_grid *grid = nullptr, const int32_t *lattice = nullptr, int cloud_side = 0)
{
    // cloud_side (eval_entry): 0 = plan queries and the callers that never gate; 1 = first pass of a query -- if the points are a
    // cloud the window kernel may take (kWindowCloudMin points, a patch-resolution wide map, the Hilbert order), this pass
    // enqueues order + probe + the GATED window launch and returns kGatedSecondPass; 2 = the second pass: the gated cell-run launch
    const bool plan_only = plan_out != nullptr;
    int rc = check_views(views);
    if (rc != D3F_OK) return rc;
    if (n < 0) return fail(D3F_ERR_INVALID_ARG, "n=%lld is negative", (long long)n);
    if (n == 0 && !plan_only) return D3F_OK;
    if (!plan_only && ((!pts && !grid) || !out_dist || !out_valid)) return fail(D3F_ERR_INVALID_ARG, "pts/out_dist/out_valid must be non-NULL");
    if (n_maps < 0 || n_maps > D3F_MAX_MAPS) return fail(D3F_ERR_BAD_SHAPE, "n_maps=%d outside [0,%d]", n_maps, D3F_MAX_MAPS);
    if (n_maps > 0 && (!maps || (!out_fused && !plan_only))) return fail(D3F_ERR_INVALID_ARG, "maps/out_fused must be non-NULL when n_maps > 0");
    if (!(mu > 0.0f)) return fail(D3F_ERR_INVALID_ARG, "mu must be > 0");

    // D3F_FLAG_REFERENCE_ROUNDING: no fast path at all -- every point takes the strict form (the reference's operation order)
    if (flags & D3F_FLAG_REFERENCE_ROUNDING) flags = (flags & ~D3F_FLAG_FINITE_MAPS) | D3F_TUNE_DIRECT_GATHER;
    d3f::EvalParams P;
    // device-side finiteness words (d3f_map_check): used when the host did not vouch for the maps and EVERY tensor of the
    // query carries one; the kernels then decide on the device, and the launch is planned for finite maps
    P.n_words = 0;
    if (!(flags & (D3F_FLAG_FINITE_MAPS | D3F_FLAG_REFERENCE_ROUNDING)) && views->depth_nonfinite && mode == 0) {
        bool all = true;
        for (int s = 0; s < n_maps; ++s) all = all && maps && maps[s].nonfinite;
        if (all) {
            P.words[P.n_words++] = views->depth_nonfinite;
            for (int s = 0; s < n_maps; ++s) P.words[P.n_words++] = maps[s].nonfinite;
        }
    }
    const bool finite_expected = (flags & D3F_FLAG_FINITE_MAPS) || P.n_words > 0;
    P.exp_stamps = nullptr;
#ifdef D3F_EXPERIMENTS
    if (exp_knob("D3F_EXP_STAMPS") > 0 && !plan_out) P.exp_stamps = exp_stamp_buffer(true);
#endif
    P.depth = views->depth; P.K = views->K; P.pose = views->pose; P.pts = pts;
    P.order = nullptr; P.lds_pad = 0;
    P.grid_x = grid ? grid->x : nullptr; P.grid_y = grid ? grid->y : nullptr; P.grid_z = grid ? grid->z : nullptr;
    P.grid_ny = grid ? grid->ny : 0; P.grid_nz = grid ? grid->nz : 0;
    P.walk_nx = P.walk_ny = P.walk_nz = 0; P.walk_tx = P.walk_ty = P.walk_tz = 1;
    P.sl_unit = 128; P.sl_ilv = 1; P.sl_slices = 0; P.sl_lg = 3; P.sl_vc = 4; P.sl_tiles = P.sl_groups = P.sl_chunks = 0;
    P.runs_occ = exp_knob("D3F_EXP_RUNS_OCC");
    P.thin_max_views = (exp_knob("D3F_EXP_THIN") < 0 || (flags & D3F_TUNE_DIRECT_GATHER)) ? 0 : 8;
    P.win_lpp = exp_knob("D3F_EXP_WINDOW_LPP") == 32 ? 32 : 16;     // 16 lanes x 2 vectors per point (C2 patch 0.565 -> 0.54 ms); U > 1: 32
    P.win_slices = 0; P.win_u = 1; P.win_vc = 1; P.win_pool_offset = 0; P.win_pool_texels = 0; P.win_occ = 4;
    P.win_pipe = exp_knob("D3F_EXP_WINDOW_PIPE") < 0 ? 0 : 1;
    P.win_sparse = 0;
    P.gate = nullptr; P.gate_min = 0u; P.gate_want = 0;
    P.store_policy = exp_knob("D3F_EXP_STORE") < 0 ? 0 : (exp_knob("D3F_EXP_STORE") == 1 ? 1 : (exp_knob("D3F_EXP_STORE") == 3 ? 3 : 2));     // non-temporal rows (fuse_common.h: store_out)
    int64_t map_bytes = 0;
    P.out_dist = out_dist; P.out_valid = out_valid;
    P.n = n; P.V = views->V; P.H = views->H; P.W = views->W;
    P.n_maps = n_maps; P.tile_pts = tile_points_for(views->V);
    P.flags = flags; P.mu = mu;
    // tuning bits (D3F_TUNE_*): experiments only, results never depend on them
    const int tl = (int)((flags >> 8) & 0xF);
    const int max_tile = tile_points_for(views->V) * 2;
    for (int s = 0; s < n_maps; ++s) {
        if (!plan_only && !out_fused[s]) return fail(D3F_ERR_INVALID_ARG, "map %d: output pointer is NULL", s);
        rc = fill_map(maps[s], s, views->V, out_fused ? out_fused[s] : nullptr, out_inter ? out_inter[s] : nullptr, nullptr,
                      P.maps[s], map_bytes, flags);
        if (rc != D3F_OK) return rc;
    }
    // The channel-sliced and the LDS-window kernels want THE wide map first (the thin ones ride along).  A map descriptor
    // carries its own output pointers, so the launch may take the maps in any order: return_names=['mask', 'dino_feats']
    // gets the same kernels as the reference's default ['dino_feats', 'mask'].  caller_map[k] = caller's index of P.maps[k].
    int caller_map[D3F_MAX_MAPS];
    bool want_inter[D3F_MAX_MAPS];
    for (int s = 0; s < D3F_MAX_MAPS; ++s) { caller_map[s] = s; want_inter[s] = false; }
    {
        int wide = -1, nwide = 0;
        for (int s = 0; s < n_maps; ++s)
            if ((int64_t)P.maps[s].C * P.maps[s].esize > 256) { if (wide < 0) wide = s; ++nwide; }
        if (nwide == 1 && wide > 0) {
            const d3f::MapDesc t = P.maps[0]; P.maps[0] = P.maps[wide]; P.maps[wide] = t;
            caller_map[0] = wide; caller_map[wide] = 0;
        }
        for (int s = 0; s < n_maps; ++s) want_inter[s] = out_inter && out_inter[caller_map[s]];
    }
    // Morton point order (performance only) when scratch is supplied and the maps exceed the L2s
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const bool may_reorder = (workspace || plan_only) && !grid && n_maps > 0 && n <= 0x7fffffffLL && !(flags & D3F_TUNE_NO_REORDER) &&
                             workspace_bytes >= d3f::order_workspace_bytes(n);
    // D3F_TUNE_DIRECT_GATHER: the plain direct gather in the chosen point order -- no texel windows, no cell runs, no channel
    // slices, thin maps view by view.  The reference the bit-identity tests compare every fast path with.
    const bool direct = (flags & D3F_TUNE_DIRECT_GATHER) != 0;
    // Cell-run gather for patch-resolution wide maps (at most two per launch: one corner-record slot each): consecutive
    // points of the processing order (a grid column in caller order, the Morton walk of a cloud) mostly stay inside one
    // texel cell of a view, so a lane group keeps the four corner vectors in registers across a run of points
    // (fuse_eval.hip).  Needs the exact invalid-view skip (finite maps), no '<k>_inter' output and fp32 maps only.
    // LDS texel windows (fuse_eval.hip, fused_eval_window_kernel): the FIRST map is a patch-resolution wide fp32 map with
    // whole 512-byte slices, every other map is thin; same preconditions as the cell-run gather, which it replaces.
    bool window = false;
    const int win_knob = exp_knob("D3F_EXP_WINDOW");          // 0 automatic (see below), -1 off, 32 / 64 / 128: points per workgroup
    // would this query's points be walked in the Hilbert order?  (the cloud half of `reorder` below)
    const bool reorder_cloud = may_reorder && !lattice && ((flags & D3F_TUNE_FORCE_REORDER) || (n >= kSmallBatch && (map_bytes > kCacheResidentBytes || (flags & D3F_FLAG_UNORDERED_POINTS))));
    const bool cloud_candidate = cloud_side == 1 && reorder_cloud && n >= kWindowCloudMin && !(flags & D3F_TUNE_NO_WINDOW_GATE);
    {
        // default: lattices (a brick's windows are compact), and clouds through the device-side gate (fuse_eval.hip: gated_out);
        // not when a cell-run variant is asked for explicitly
        const bool automatic = win_knob == 0 && (lattice != nullptr || cloud_candidate) && exp_knob("D3F_EXP_RUNS") == 0 && exp_knob("D3F_EXP_RUNS_U") == 0;
        const bool half0 = n_maps >= 1 && P.maps[0].esize == 2;      // fp16-stored: bricks of a lattice only (the cell-run side of a cloud's gate is fp32)
        window = (win_knob > 0 || automatic) && !direct && mode == 0 && n_maps >= 1 && finite_expected && n >= kSmallBatch &&
                 n <= 0x7fffffffLL && tl == 0 && views->V <= 8 && window_candidate(P.maps[0], views, !plan_only) &&
                 (!half0 || (lattice != nullptr && exp_knob("D3F_EXP_WINDOW_F16") >= 0));
        for (int s = 0; s < n_maps; ++s) window = window && !want_inter[s];
        for (int s = 1; s < n_maps; ++s) window = window && P.maps[s].esize == 4 && P.maps[s].C * 4 <= 256;
        if (window) {
            const int T = (win_knob == 32 || win_knob == 64 || win_knob == 128) ? win_knob : 64;
            const int VP = views->V <= 1 ? 1 : views->V <= 2 ? 2 : views->V <= 4 ? 4 : 8;
            int U = exp_knob("D3F_EXP_WINDOW_U");
            const int cv = P.maps[0].C / 128;                  // 128-channel granules per texel (512 bytes of fp32, 256 of fp16)
            const int slot = P.maps[0].esize == 2 ? 256 : 512;
            if (U < 1 || U > 4 || cv % U != 0 || slot == 256) U = 1;
            if (slot == 256) P.win_lpp = 16;
            // per (point, view): 32-byte window record (+ the 16-byte view record when thin maps ride along); per point 20 bytes
            const int base = T * (views->V * 32 + 16) + (n_maps > 1 ? T * views->V * 16 : 0) + T * 20 + views->V * 48;     // records at a padded point stride
            const int pool_offset = (base + 511) / 512 * 512;
            int occ = exp_knob("D3F_EXP_WINDOW_OCC");
            const bool occ_forced = occ >= 5 && occ <= 6;        // experiments: 5 / 6 workgroups per CU with the plain point loop
            if (occ < 2 || occ > 6) occ = 4;
            if (U > 1) occ = 2;                                 // those variants are built for 2 workgroups per CU
            // touched-texel pool (SPARSE) for clouds, whole rectangles for lattice bricks (which never overflow: 0.42 vs 0.455 ms on
            // C2-patch); experiments builds: D3F_EXP_WINDOW_SPARSE = 1 / -1 forces either
            P.win_sparse = exp_knob("D3F_EXP_WINDOW_SPARSE") > 0 ? 1 : (exp_knob("D3F_EXP_WINDOW_SPARSE") < 0 ? 0 : (lattice ? 0 : 1));
            if (slot == 256) P.win_sparse = 0;
            // static LDS of the kernel + allocation granularity: 3 workgroups per CU stop fitting with less (measured, round 5)
            const int slack = exp_knob("D3F_EXP_WINDOW_SLACK") > 0 ? exp_knob("D3F_EXP_WINDOW_SLACK") : (P.win_sparse ? 4096 : 2048);
            // slots per view worth a workgroup per CU: a brick's rectangles ~17; a cloud tile's touched texels ~12 (p90 14)
            const int want = exp_knob("D3F_EXP_WINDOW_WANT") > 0 ? exp_knob("D3F_EXP_WINDOW_WANT") : (P.win_sparse ? 14 : 17);
            int texels = 0;
            for (;; --occ) {
                const int budget = 160 * 1024 / occ - slack;
                texels = (budget - pool_offset) / (slot * U) - 2;
                // A 4x4x4 brick's window is ~3x4 texels per view once a texel is at least as wide as the brick's footprint
                // (config 4's slab: 2.5-mm lattice, 10-px texels), and a pool that cannot hold the views' windows sends the
                // overflowing pairs to the global gather: give every view enough slots, at the price of workgroups per CU
                // (MI355X, config 4 lattice slab: 4 / 3 / 2 workgroups per CU = 3.15 / 3.54 / 2.43 ms; config 2, four
                // views, fits at 4 and loses 18 % at 2)
                // (round 4, pipelined point loop: a point with a pair outside the pool is done a second time by the general path, so
                // overflow costs more than a workgroup per CU -- C2-patch: 55 slots at 4 per CU 0.525 ms, 80 slots at 3 per CU 0.493,
                // 40 / 32 slots 0.72 / 0.81: ask for ~17 slots per view)
                if (texels >= want * views->V || occ == 2 || occ_forced) break;
            }
            if (exp_knob("D3F_EXP_WINDOW_POOL") > 0 && exp_knob("D3F_EXP_WINDOW_POOL") < texels) texels = exp_knob("D3F_EXP_WINDOW_POOL");
            if (texels > 320) texels = 320;                      // kWinMaxTexels (fuse_eval.hip)
            texels &= ~1;
            window = texels >= 2 && (T * VP) % 64 == 0 && n / T < 0x7fffffffLL;
            if (U > 1) P.win_lpp = 32;
            P.win_u = U; P.win_occ = occ; P.win_pool_offset = pool_offset; P.win_pool_texels = texels;
            P.win_vc = exp_knob("D3F_EXP_WINDOW_VC") == 2 ? 2 : 1;
            P.win_slices = window ? cv / U : 0;
        }
    }
    bool any_runs = false;
    {
        const int knob = exp_knob("D3F_EXP_RUNS");
        bool blocked = window || direct || knob < 0 || !finite_expected || n < kSmallBatch || tl != 0;
        for (int s = 0; s < n_maps; ++s)
            blocked |= P.maps[s].esize == 2 || want_inter[s] ||
                       (P.maps[s].unroll == -4 && !runs_candidate(P.maps[s], views->H, views->W));
        // one cell-run map per launch (phase A keeps one "same cell as the previous point" flag per (point, view))
        for (int s = 0; s < n_maps && !blocked && !any_runs; ++s)
            if (runs_candidate(P.maps[s], views->H, views->W)) {
                pick_runs_mapping(P.maps[s], exp_knob("D3F_EXP_RUNS_U"), knob);
                any_runs = true;
            }
        if (any_runs)       // the cell-run kernel is built for one batched vector per lane on its other maps (register budget)
            for (int s = 0; s < n_maps; ++s)
                if (P.maps[s].runs == 0 && P.maps[s].unroll != 1)
                    pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
    }
    // Points on a regular lattice (a d3f_grid, or d3f_eval_lattice's dims): the brick walk is closed form -- no keys, no
    // sort, no index array, no scratch -- and replaces the Morton sort wherever that would be used.  (With the cell-run
    // gather the caller's z-fastest order is the one wanted: a grid column is one long run.)
    // (a flat lattice with more than 2^28 tiles per 16-tile slab would overflow the walk's 32-bit level arithmetic)
    const bool walk_fits = lattice && 16.0 * ((lattice[1] + 1) / 2) * ((lattice[2] + 1) / 2) < 4294967296.0;
    const bool walk = lattice && walk_fits && n_maps > 0 && n >= kSmallBatch && n <= 0x7fffffffLL && !(flags & D3F_TUNE_NO_REORDER) &&
                      exp_knob("D3F_EXP_WALK") >= 0 && !any_runs &&
                      ((flags & D3F_TUNE_FORCE_REORDER) || map_bytes > kCacheResidentBytes || window);
    const bool reorder = walk || (may_reorder && ((flags & D3F_TUNE_FORCE_REORDER) || (n >= kSmallBatch && (map_bytes > kCacheResidentBytes || (flags & D3F_FLAG_UNORDERED_POINTS)))));
    if (walk) {
        P.walk_nx = lattice[0]; P.walk_ny = lattice[1]; P.walk_nz = lattice[2];
    } else if (reorder && !plan_only && (flags & D3F_FLAG_REUSE_POINT_ORDER)) {
        P.order = d3f::stored_point_order(workspace, n);       // written by an earlier call for the same points
    } else if (reorder && !plan_only) {
        // Hilbert order of the 4-mm cells, exact (order_kernels.hip); experiments builds: D3F_EXP_ORDER_MORTON=1 = the Z curve of rounds 1-4
        hipError_t eo = d3f::build_point_order(pts, n, workspace, workspace_bytes, &P.order, hs, exp_knob("D3F_EXP_ORDER_MORTON") > 0 ? 1 : 0);
        if (eo != hipSuccess) return hip_fail(eo, "point ordering");
    }
    // Launch geometry (measured on MI355X):
    //  * Morton / lattice walk: 8-point tiles (one point per lane group; 16 when a thin map such as the mask is also
    //    requested), XCD k takes the k-th contiguous eighth of the walk, so the ~1 k points in flight on an
    //    XCD form one compact blob whose texels stay in that XCD's 4 MiB L2
    //    (C2 dense 2.84 -> 2.12 ms, C4 patch 13.0 -> 4.8 ms; 32-point tiles: 2.48 / 5.3 ms);
    //  * caller order: 128-point tiles, round-robin XCDs (0.80 ms on C2 patch); for maps far beyond the
    //    256 MiB Infinity Cache 64-point tiles at 2 workgroups per CU (3.2 -> 2.96 ms on C2 dense);
    //  * cell-run gather: 8 runs per workgroup (64 points with 32 lanes per point), either order.
    bool xcd_remap = false;
    if (reorder) {
        // on the walk the in-flight footprint is tiny, so batched corner loads win again wherever one pass
        // of <= 3 vectors per lane covers the channels (C2 dense 2.07 -> 1.99 ms); C = 1024 keeps load-use x 4
        for (int s = 0; s < n_maps; ++s) {
            d3f::MapDesc &m = P.maps[s];
            const bool forced = (flags & ((1u << 26) | (1u << 27))) != 0;
            if (!forced && m.unroll < 0 && (m.C / m.vw) <= 3 * 64) {
                const bool a16 = m.vw == 4, a8 = m.vw >= 2;
                pick_mapping(m, a16, a8, true, any_runs ? 1 : 4);
            }
        }
        // one point per lane group: 8 points when every map takes 32 lanes per point, else 16
        bool thin = false;
        for (int s = 0; s < n_maps; ++s) thin |= P.maps[s].lpp_log2 < 5;      // < 32 lanes per point: 16 groups have work
        P.tile_pts = thin ? 16 : 8; P.lds_pad = 0; xcd_remap = true;
        if (walk) {                                   // the tile is a brick of the lattice
            P.walk_tx = 2; P.walk_ty = 2; P.walk_tz = thin ? 4 : 2;
            const int shape = exp_knob("D3F_EXP_WALK_TILE");      // experiment: digits x y z, e.g. 224, 144, 422
            if (shape >= 111 && shape <= 888 && (shape / 100) * (shape / 10 % 10) * (shape % 10) == P.tile_pts && shape / 10 % 10 > 0 && shape % 10 > 0) {
                P.walk_tx = shape / 100; P.walk_ty = shape / 10 % 10; P.walk_tz = shape % 10;
            }
        }
        // maps that fit the L2s / Infinity Cache anyway (patch-resolution features, the mask): the walk is only there
        // to give a random cloud L1 locality and the big tiles of the caller-order path stay best
        // (C2 patch, random cloud: caller order 1.93 ms, walk with 8-point tiles 1.07, with 128-point tiles 0.76)
        if (map_bytes <= kCacheResidentBytes && !walk) P.tile_pts = tile_points_for(views->V);
    } else if (map_bytes > kBeyondLlcBytes && P.tile_pts > 64 && n >= kSmallBatch && !any_runs) {
        P.tile_pts = 64; P.lds_pad = 64 * 1024;
    }
    // small batches (keypoints, tracking): a 128-point tile is 16-32 serial rounds per lane group, so a few hundred
    // points would run on 3 CUs for ~200 us; spread them over >= 1024 workgroups instead (N = 300: 170 -> ~25 us)
    if (!reorder && n_maps > 0)
        while (P.tile_pts > 8 && n / P.tile_pts < 1024) P.tile_pts >>= 1;
    if (tl >= 2 && tl <= 8) { P.tile_pts = (1 << tl) <= max_tile ? (1 << tl) : max_tile; P.lds_pad = 0; }
    if ((flags >> 16) & 0xFF) P.lds_pad = ((int)((flags >> 16) & 0xFF) == 0xFF) ? 0 : (int)((flags >> 16) & 0xFF) * 1024;
    if (flags & D3F_TUNE_XCD_REMAP) xcd_remap = !xcd_remap;
    P.flags = (flags & ~D3F_TUNE_XCD_REMAP) | (xcd_remap ? D3F_TUNE_XCD_REMAP : 0u);
    if (any_runs) {
        int k = 8, lg = 6;
        for (int s = 0; s < n_maps; ++s)
            if (P.maps[s].runs > 0) { k = P.maps[s].runs; lg = P.maps[s].lpp_log2; }
        const int round = (d3f::kBlock >> lg) * k;    // one run per lane group
        P.tile_pts = round < 64 ? 64 : round;         // >= 64 points per workgroup (a lane group then takes several runs)
        // batches of less than ~2 workgroups per slot (256 CUs x 7): halve the tile so that the tail is shorter
        // (100 k keypoints: 0.126 -> 0.119 ms; the 985 600-point grid is slower with 32-point tiles: 0.632 -> 0.655)
        if (n / P.tile_pts < 4096 && P.tile_pts / 2 >= round) P.tile_pts /= 2;
        if (exp_knob("D3F_EXP_RUNS_TILE") >= round) P.tile_pts = exp_knob("D3F_EXP_RUNS_TILE");
        while ((long)P.tile_pts * views->V * 88 > 40 * 1024 && P.tile_pts > round) P.tile_pts >>= 1;   // records + 2 corner slots
        P.lds_pad = 0;
    }
    // Channel-sliced launch (fuse_eval.hip): a lattice on a dense wide fp32 map that is the FIRST map of the launch; any
    // other map must be thin (it rides along with slice 0).  512-byte slices, two views in flight; a wide map WITH thin
    // companions takes 32 points per workgroup (C3-dense, features + 8-channel mask: 3.04 -> 2.80 ms -- the whole-texel
    // kernel stalls on the thin map's gather, 16 points x 2 lanes per workgroup), a wide map ALONE 16 points per workgroup
    // (C2-dense 1.62 -> 1.52 ms: with 32 the slicing removed 20-29 % of the L2 fills and no time, with 64 it lost 20 %:
    // the points in flight per XCD are what the L2 window is made of).  D3F_EXP_SLICED = 1 / 2 / 3 forces
    // 128- / 256- / 512-byte slices, -1 disables; _TILE 8 / 16 / 32 / 64 points per workgroup.
    {
        int sl = exp_knob("D3F_EXP_SLICED");
        bool thin_rest = true;
        for (int s = 1; s < n_maps; ++s) thin_rest = thin_rest && P.maps[s].C * P.maps[s].esize <= 256 && P.maps[s].esize == 4;
        const bool half_sl = n_maps >= 1 && P.maps[0].esize == 2;      // fp16-stored wide map: 16 lanes x 8 channels = 128-channel slices
        const bool automatic = sl == 0 && thin_rest && n_maps >= 1 && P.maps[0].C % 128 == 0 && P.maps[0].C <= 1024 &&
                               (!half_sl || exp_knob("D3F_EXP_SLICED_F16") >= 0);
        if (automatic) sl = half_sl ? 2 : 3;
        if (half_sl && sl != 2) sl = 0;
        // ... or the Morton order of a cloud on maps beyond the caches (tiles of 16 / 32 consecutive points of the order)
        const bool cloud = reorder && !walk && !any_runs && map_bytes > kCacheResidentBytes && exp_knob("D3F_EXP_SLICED_CLOUD") >= 0;
        bool ok = (walk || cloud) && !window && !direct && (sl >= 1 && sl <= 3) && mode == 0 && n_maps >= 1 &&
                  ((P.maps[0].esize == 4 && P.maps[0].vw == 4) || (half_sl && P.maps[0].vw == 8 && P.maps[0].fold)) && !want_inter[0] && tl == 0;
        const int lg = sl + 2, lanes = 1 << lg;      // 1: 8 lanes (128-byte slices), 2: 16 lanes, 3: 32 lanes (512 bytes)
        P.sl_vc = exp_knob("D3F_EXP_SLICED_VC") > 0 ? exp_knob("D3F_EXP_SLICED_VC") : (automatic ? 2 : 4);
        if (half_sl) P.sl_vc = 2;
        const int cpl = half_sl ? 8 : 4;              // channels per lane (one 16-byte vector)
        ok = ok && P.maps[0].C % (cpl * lanes) == 0 && P.maps[0].C >= 128;
        for (int s = 1; s < n_maps && ok; ++s) ok = P.maps[s].C * P.maps[s].esize <= 256 && !want_inter[s] && P.maps[s].esize == 4;
        if (ok) {
            const int keep_tile = P.tile_pts, keep_pad = P.lds_pad, keep_t[3] = {P.walk_tx, P.walk_ty, P.walk_tz};
            d3f::MapDesc keep_maps[D3F_MAX_MAPS];
            for (int s = 0; s < n_maps; ++s) keep_maps[s] = P.maps[s];
            const int tile_knob = exp_knob("D3F_EXP_SLICED_TILE");
            const bool big = tile_knob == 64;                            // experiment: 64 points per workgroup (four 2x2x4 tiles)
            const bool tiny = tile_knob == 16 || (tile_knob == 0 && n_maps == 1);   // 16 points per workgroup (four 2x2x1 tiles)
            const bool mini = tile_knob == 8;                            // experiment: 8 points per workgroup (four 2x1x1 tiles)
            P.walk_tx = 2; P.walk_ty = mini ? 1 : 2; P.walk_tz = big ? 4 : ((tiny || mini) ? 1 : 2);
            P.sl_lg = lg;
            P.sl_slices = P.maps[0].C / (cpl * lanes);
            P.tile_pts = big ? 64 : (tiny ? 16 : (mini ? 8 : 32)); P.lds_pad = exp_knob("D3F_EXP_SLICED_PAD") > 0 ? exp_knob("D3F_EXP_SLICED_PAD") * 1024 : 0;
            if (walk) {
                P.sl_tiles = (int64_t)((P.walk_nx + 1) / 2) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) * ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
                P.sl_groups = (P.sl_tiles + 3) / 4;
            } else {
                P.sl_tiles = 0;
                P.sl_groups = (n + P.tile_pts - 1) / P.tile_pts;
            }
            P.sl_unit = exp_knob("D3F_EXP_SLICED_UNIT") > 0 ? exp_knob("D3F_EXP_SLICED_UNIT") : (big ? 64 : (tiny ? 256 : (mini ? 512 : 128)));   // 4096 points per unit (smaller: slower)
            P.sl_chunks = (P.sl_groups + P.sl_unit - 1) / P.sl_unit;
            P.sl_ilv = exp_knob("D3F_EXP_SLICED_ILV") >= 2 && exp_knob("D3F_EXP_SLICED_ILV") <= 4 ? exp_knob("D3F_EXP_SLICED_ILV") : 1;
            for (int s = 1; s < n_maps; ++s) pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
            if ((((P.sl_chunks * P.sl_slices + 7) / 8) + P.sl_ilv) * 8 * P.sl_unit > 0x7fffffffLL) P.sl_slices = 0;
            // its dynamic LDS (records + one corner record per (point, view)) must fit the 64 KiB a launch gets without opting in:
            // 32-point tiles with ~36 and more views do not -- such a query keeps the whole-texel kernel
            if ((int64_t)d3f::fused_lds_base(P.tile_pts, views->V) + (int64_t)P.tile_pts * views->V * 32 + P.lds_pad > 64 * 1024) P.sl_slices = 0;
            if (P.sl_slices == 0) {             // not this launch after all: the geometry of the whole-texel kernel again
                P.tile_pts = keep_tile; P.lds_pad = keep_pad; P.walk_tx = keep_t[0]; P.walk_ty = keep_t[1]; P.walk_tz = keep_t[2];
                for (int s = 0; s < n_maps; ++s) P.maps[s] = keep_maps[s];
            }
        }
    }
    if (window) {
        const int T = (win_knob == 32 || win_knob == 64 || win_knob == 128) ? win_knob : 64;
        P.tile_pts = T; P.lds_pad = 0;
        if (walk) pick_window_brick(P.walk_nx, P.walk_ny, P.walk_nz, T, P.walk_tx, P.walk_ty, P.walk_tz);
        for (int s = 1; s < n_maps; ++s) pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
        xcd_remap = false;
        P.flags &= ~D3F_TUNE_XCD_REMAP;
        // a cloud's tiles go round-robin over the XCDs (all eight work on one neighbourhood: C2-patch cloud 0.52 ms against 0.58
        // with contiguous eighths, which is the lattice bricks' mapping); experiments builds: D3F_EXP_WINDOW_RR=-1 = eighths
        if (!walk && exp_knob("D3F_EXP_WINDOW_RR") >= 0) P.flags |= D3F_TUNE_XCD_REMAP;
    }
    // walks: all eight XCDs stay inside one macro-brick of ~32 k points at a time (its texel footprint stays in
    // the 256 MiB Infinity Cache), each taking a contiguous eighth of it (C2 dense 1.97 -> 1.74 ms, C4 patch 4.75 -> 4.17)
    P.xcd_chunk = (reorder && xcd_remap) ? (int)((32768 / P.tile_pts + 7) / 8 * 8) : 0;
    if ((flags >> 29) & 0x7) P.xcd_chunk = 1024 << (((flags >> 29) & 0x7) - 1);   // tuning: 1024 .. 65536 tiles
    if (n_maps == 0) {
        // distance-only pass (return_names=[], eval_dist): one lane per point and nothing per point in LDS.  Rounds 1-3 ran it on
        // the 128-point tiles of the gathers -- half of every 256-lane workgroup idle (SQ_WAVES = 3.85 M for 123.2 M points):
        // four points per lane and workgroup instead, KRt computed once per 1024 points
        P.tile_pts = n >= (1LL << 22) ? 1024 : 256;
        P.lds_pad = 0;
    }
    P.crec_offset = d3f::fused_lds_base(n_maps == 0 ? 0 : P.tile_pts, views->V);
    // wide maps (>= 16 lanes per point): corner set-up once per (point, view) in phase A, 32 B of LDS each; the
    // cell-run maps come first -- they read their corners from these records
    P.n_pre = 0;
    for (int s = 0; s < n_maps; ++s)
        if (P.maps[s].runs > 0) P.maps[s].pre_slot = P.n_pre++;
    if (P.sl_slices > 0) {
        P.maps[0].pre_slot = 0; P.n_pre = 1;             // the sliced kernel keeps the wide map's corner records itself
    } else if (!(flags & (1u << 28)))
        for (int s = 0; s < n_maps && P.n_pre < 2; ++s) {
            // 32 B per (point, view) and map: only while records + set-ups stay within 48 KiB (>= 3 workgroups per CU)
            const long lds_after = (long)P.crec_offset + (long)(P.n_pre + 1) * P.tile_pts * views->V * 32;
            if (P.maps[s].pre_slot < 0 && P.maps[s].lpp_log2 >= 4 && !want_inter[s] && lds_after <= 48 * 1024)
                P.maps[s].pre_slot = P.n_pre++;
        }
    int64_t ntiles = (n + P.tile_pts - 1) / P.tile_pts;
    if (walk)
        ntiles = (int64_t)((P.walk_nx + P.walk_tx - 1) / P.walk_tx) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) *
                 ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
    if (ntiles > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "n=%lld needs more than 2^31 workgroups", (long long)n);
    if (plan_only) {
        plan_out->gated_window = 0; plan_out->reserved2 = 0;
        if (cloud_side == 0 && !lattice && !grid) {       // would d3f_eval's first pass take the window side?  (its plan: the lattice's)
            d3f_eval_plan side;
            if (eval_common(views, pts, n, maps, n_maps, mu, flags, out_dist, out_valid, out_fused, out_inter, workspace, workspace_bytes,
                            stream, mode, &side, grid, lattice, 1) == D3F_OK)
                for (int s = 0; s < n_maps; ++s)
                    if (side.staged[s] == 3 && side.reorder == 1) { plan_out->gated_window = 1; plan_out->reserved2 = side.reserved; }
        }
        plan_out->tile_points = P.tile_pts;
        plan_out->reorder = walk ? 2 : (reorder ? 1 : 0);
        plan_out->lds_bytes = P.crec_offset + P.n_pre * P.tile_pts * P.V * 32 + P.lds_pad;
        plan_out->workgroups = P.sl_slices > 0 ? (((P.sl_chunks * P.sl_slices + 7) / 8 + P.sl_ilv - 1) / P.sl_ilv * P.sl_ilv) * 8 * P.sl_unit : ntiles;
        if (P.win_slices > 0) {
            plan_out->lds_bytes = P.win_pool_offset + (2 + P.win_pool_texels) * (P.maps[0].esize == 2 ? 256 : 512) * P.win_u;
            plan_out->workgroups = ntiles;
        }
        plan_out->reserved = P.sl_slices > 0 ? 100 + P.sl_lg * 10 + P.sl_vc : (P.win_slices > 0 ? 2000 + 100 * P.win_u + 10 * (P.win_u == 1 ? (P.win_lpp == 16 ? P.win_vc : 4) : (P.win_u == 4 ? 1 : P.win_vc)) + (P.win_u == 1 ? (P.win_occ >= 4 ? 4 : (P.win_lpp == 16 ? 3 : P.win_occ)) : 2) : 0);   // 2UVW: the window kernel's template arguments      // 1LV: sliced launch, L = log2(lanes per point), V = views in flight
        for (int s = 0; s < n_maps; ++s)
            if (P.maps[s].runs > 0) {        // waves per SIMD the chosen cell-run kernel variant is built for
                const int ru = P.maps[s].unroll, rk = P.maps[s].runs;
                plan_out->reserved = (ru == 1 && rk == 4) ? (P.runs_occ == 6 ? 6 : 7) : (ru == 1 ? ((P.runs_occ == 4 || P.runs_occ == 6) ? P.runs_occ : 5) : ((ru == 2 && rk == 8 && P.runs_occ != 4) ? 3 : 4));
            }
        for (int s = 0; s < D3F_MAX_MAPS; ++s) {
            const bool on = s < n_maps;
            const int c = on ? caller_map[s] : s;         // reported in the caller's map order
            plan_out->vector_floats[c] = on ? P.maps[s].vw : 0;
            plan_out->lanes_per_point[c] = on ? ((P.win_slices > 0 && s == 0) ? P.win_lpp : (1 << P.maps[s].lpp_log2)) : 0;
            plan_out->vectors_per_lane[c] = on ? ((P.win_slices > 0 && s == 0) ? P.win_u * (32 / P.win_lpp) : P.maps[s].unroll) : 0;   /* negative: load-use per vector */
            plan_out->staged[c] = on ? (P.win_slices > 0 && s == 0 ? 3 : (P.maps[s].runs > 0 ? 16 + P.maps[s].runs : 0)) : 0;
        }
        return D3F_OK;
    }
    // a cloud on the gated pair of launches: the window side (this pass) and the cell-run side (the next) read one device word
    const bool gated_window = cloud_side == 1 && window && !walk && P.order != nullptr;
    const bool gated_runs = cloud_side == 2;
    if (gated_window || gated_runs) {
        P.gate = d3f::order_gate_words(workspace, n);
        P.gate_min = (uint32_t)D3F_GATE_MIN_FIT;                   // three quarters of the sampled tiles fit their pool
        if ((flags & D3F_TUNE_WINDOW_SIDE) || exp_knob("D3F_EXP_GATE") > 0) P.gate_min = 0u;          // always the window side
        if (exp_knob("D3F_EXP_GATE") < 0) P.gate_min = 0xffffffffu;                                    // experiments: always the cell runs
        P.gate_want = gated_window ? 1 : 0;
    }
    hipEvent_t ev0 = g_prof_start, ev1 = gated_window ? nullptr : g_prof_stop;
    g_prof_start = nullptr;
    if (!gated_window) g_prof_stop = nullptr;
    if (ev0) (void)hipEventRecord(ev0, hs);
    if (gated_window) {
        hipError_t ep = d3f::launch_window_gate_probe(P, d3f::order_gate_words(workspace, n), d3f::kGateSamples, hs);
        if (ep != hipSuccess) return hip_fail(ep, "window gate probe");
    }
    hipError_t e = d3f::launch_fused_eval(P, mode, hs);
    if (ev1) (void)hipEventRecord(ev1, hs);
    if (e != hipSuccess) return hip_fail(e, "fused_eval launch");
    return gated_window ? kGatedSecondPass : D3F_OK;
}

}  // namespace

extern "C" {

int d3f_abi_version(void) { return D3F_ABI_VERSION; }
const char *d3f_version(void) { return "d3fields-hip 0.3.0 gfx950"; }
int d3f_build_has_experiments(void)
{
#ifdef D3F_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
const char *d3f_last_error(void) { return g_err; }

int d3f_eval(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps, int32_t n_maps,
             float mu, uint32_t flags, float *out_dist, uint8_t *out_valid, float *const *out_fused,
             float *const *out_inter, void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = eval_common(views, pts, n, maps, n_maps, mu, flags, out_dist, out_valid, out_fused, out_inter, workspace,
                         workspace_bytes, stream, 0, nullptr, nullptr, nullptr, 1);
    if (rc == kGatedSecondPass)         // a cloud: the window launch is enqueued behind its gate; now the cell-run launch behind the same word
        rc = eval_common(views, pts, n, maps, n_maps, mu, flags | D3F_FLAG_REUSE_POINT_ORDER, out_dist, out_valid, out_fused, out_inter,
                         workspace, workspace_bytes, stream, 0, nullptr, nullptr, nullptr, 2);
    return rc;
}

int64_t d3f_eval_workspace_bytes(int64_t n) { return d3f::order_workspace_bytes(n); }

int64_t d3f_eval_gate_offset(int64_t n)
{
    return n <= 0 ? 0 : d3f::order_gate_offset(n);
}

void d3f_profile_next_eval(void *start_event, void *stop_event)
{
    g_prof_start = static_cast<hipEvent_t>(start_event);
    g_prof_stop = static_cast<hipEvent_t>(stop_event);
}

int d3f_eval_plan_query(const d3f_views *views, int64_t n, const d3f_channel_map *maps, int32_t n_maps, uint32_t flags,
                        int32_t have_workspace, int32_t want_inter, d3f_eval_plan *plan)
{
    if (!plan) return fail(D3F_ERR_INVALID_ARG, "plan is NULL");
    float *inter[D3F_MAX_MAPS];
    for (int s = 0; s < D3F_MAX_MAPS; ++s) inter[s] = want_inter ? reinterpret_cast<float *>(16) : nullptr;
    return eval_common(views, nullptr, n, maps, n_maps, 0.02f, flags, nullptr, nullptr, nullptr, want_inter ? inter : nullptr,
                       nullptr, have_workspace ? d3f::order_workspace_bytes(n) : 0, nullptr, 0, plan);
}

int d3f_eval_plan_query_lattice(const d3f_views *views, int32_t nx, int32_t ny, int32_t nz, const d3f_channel_map *maps,
                                int32_t n_maps, uint32_t flags, int32_t want_inter, d3f_eval_plan *plan)
{
    if (!plan) return fail(D3F_ERR_INVALID_ARG, "plan is NULL");
    if (nx < 0 || ny < 0 || nz < 0) return fail(D3F_ERR_BAD_SHAPE, "plan_query_lattice: negative size");
    float *inter[D3F_MAX_MAPS];
    for (int s = 0; s < D3F_MAX_MAPS; ++s) inter[s] = want_inter ? reinterpret_cast<float *>(16) : nullptr;
    const int32_t dims[3] = {nx, ny, nz};
    return eval_common(views, nullptr, (int64_t)nx * ny * nz, maps, n_maps, 0.02f, flags, nullptr, nullptr, nullptr,
                       want_inter ? inter : nullptr, nullptr, 0, nullptr, 0, plan, nullptr, dims);
}

int64_t d3f_backproject_workspace_bytes(int32_t H, int32_t W)
{
    if (H <= 0 || W <= 0) return 0;
    return (((int64_t)H * W + d3f::kBlock - 1) / d3f::kBlock + 1) * (int64_t)sizeof(int64_t);
}

int d3f_backproject_view(const double *depth, const uint8_t *mask, int32_t H, int32_t W, const double *cam_params,
                         const double *cam_to_world, const double *bounds, int64_t capacity, double *out_pts, int32_t *out_pixel,
                         int64_t *count_out, void *workspace, void *stream)
{
    if (H < 1 || W < 1 || (int64_t)H * W > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "backproject: H=%d W=%d", H, W);
    if (!depth || !cam_params || !cam_to_world || !count_out || !workspace || capacity < 0 || (capacity > 0 && !out_pts))
        return fail(D3F_ERR_INVALID_ARG, "backproject: NULL pointer or negative capacity");
    if (!(cam_params[0] != 0.0) || !(cam_params[1] != 0.0)) return fail(D3F_ERR_INVALID_ARG, "backproject: fx/fy must be non-zero");
    hipError_t e = d3f::launch_backproject(depth, mask, H, W, cam_params, cam_to_world, bounds, capacity, out_pts, out_pixel,
                                           count_out, static_cast<int64_t *>(workspace), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "backproject launch");
}

int d3f_pcd_nearest(const double *a, int64_t na, const double *b, int64_t nb, double *min_dist, int64_t *argmin, void *stream)
{
    if (na < 0 || nb < 1) return fail(D3F_ERR_BAD_SHAPE, "pcd_nearest: na=%lld nb=%lld (nb must be >= 1)", (long long)na, (long long)nb);
    if (na == 0) return D3F_OK;
    if (!a || !b || !min_dist || !argmin) return fail(D3F_ERR_INVALID_ARG, "pcd_nearest: NULL pointer");
    hipError_t e = d3f::launch_nearest(a, na, b, nb, min_dist, argmin, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "pcd_nearest launch");
}

int d3f_pcd_to_index(const double *pts, int64_t n, const double *lower, double voxel_size, const int32_t *voxel_num,
                     int32_t *out_index, int32_t *out_voxel, void *stream)
{
    if (n < 0) return fail(D3F_ERR_INVALID_ARG, "pcd_to_index: n=%lld is negative", (long long)n);
    if (!lower || !voxel_num) return fail(D3F_ERR_INVALID_ARG, "pcd_to_index: lower / voxel_num (host arrays) are NULL");
    if (!(voxel_size != 0.0)) return fail(D3F_ERR_INVALID_ARG, "pcd_to_index: voxel_size must be non-zero");
    if (n == 0) return D3F_OK;
    if (!pts || !out_index) return fail(D3F_ERR_INVALID_ARG, "pcd_to_index: NULL pointer");
    if ((n + d3f::kBlock - 1) / d3f::kBlock > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "pcd_to_index: n=%lld too large for one launch", (long long)n);
    hipError_t e = d3f::launch_pcd_to_index(pts, n, lower, voxel_size, voxel_num, out_index, out_voxel, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "pcd_to_index launch");
}

int64_t d3f_vox_iou_workspace_bytes(int64_t n1, int64_t n2)
{
    if (n1 < 0 || n2 < 0) return 0;
    return d3f::voxset_capacity(n1, n2) * (int64_t)sizeof(unsigned long long);
}

int d3f_vox_idx_iou(const int32_t *idx1, int64_t n1, const int32_t *idx2, int64_t n2, int64_t *out_counts, void *workspace,
                    int64_t workspace_bytes, void *stream)
{
    if (n1 < 0 || n2 < 0 || n1 + n2 > 0x3fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "vox_idx_iou: n1=%lld n2=%lld", (long long)n1, (long long)n2);
    if (!out_counts || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return fail(D3F_ERR_INVALID_ARG, "vox_idx_iou: NULL pointer");
    if (!workspace || workspace_bytes < d3f_vox_iou_workspace_bytes(n1, n2))
        return fail(D3F_ERR_WORKSPACE, "vox_idx_iou: needs %lld workspace bytes", (long long)d3f_vox_iou_workspace_bytes(n1, n2));
    if (!aligned(workspace, 8) || !aligned(out_counts, 8)) return fail(D3F_ERR_BAD_LAYOUT, "vox_idx_iou: workspace / out_counts must be 8-byte aligned");
    hipError_t e = d3f::launch_voxset_iou(idx1, n1, idx2, n2, out_counts, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "vox_idx_iou launch");
}

int d3f_erode(const uint8_t *src, int32_t H, int32_t W, int32_t kh, int32_t kw, uint8_t *dst, void *stream)
{
    if (H < 0 || W < 0 || kh < 1 || kw < 1 || kh > 255 || kw > 255) return fail(D3F_ERR_BAD_SHAPE, "erode: H=%d W=%d kernel %dx%d", H, W, kh, kw);
    if ((int64_t)H * W == 0) return D3F_OK;
    if (!src || !dst || src == dst) return fail(D3F_ERR_INVALID_ARG, "erode: src/dst must be distinct non-NULL images");
    if (((int64_t)H * W + d3f::kBlock - 1) / d3f::kBlock > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "erode: image too large for one launch");
    hipError_t e = d3f::launch_erode(src, H, W, kh, kw, dst, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "erode launch");
}

int64_t d3f_voxel_downsample_workspace_bytes(int64_t n) { return n > 0 ? d3f::voxmean_workspace_bytes(n) : 0; }

int d3f_voxel_downsample(const double *pts, const double *colors, int64_t n, double voxel_size, double *out_pts, double *out_colors,
                         int64_t *count_out, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (n < 0 || n > 0x3fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "voxel_downsample: n=%lld", (long long)n);
    if (!(voxel_size > 0.0)) return fail(D3F_ERR_INVALID_ARG, "voxel_downsample: voxel_size must be > 0 (open3d raises too)");
    if (!count_out) return fail(D3F_ERR_INVALID_ARG, "voxel_downsample: count_out is NULL");
    if (n == 0) return hipMemsetAsync(count_out, 0, sizeof(int64_t), static_cast<hipStream_t>(stream)) == hipSuccess ? D3F_OK : D3F_ERR_HIP;
    if (!pts || !out_pts || (colors && !out_colors)) return fail(D3F_ERR_INVALID_ARG, "voxel_downsample: NULL pointer");
    if (!workspace || workspace_bytes < d3f_voxel_downsample_workspace_bytes(n))
        return fail(D3F_ERR_WORKSPACE, "voxel_downsample: needs %lld workspace bytes", (long long)d3f_voxel_downsample_workspace_bytes(n));
    if (!aligned(workspace, 16)) return fail(D3F_ERR_BAD_LAYOUT, "voxel_downsample: workspace must be 16-byte aligned");
    hipError_t e = d3f::launch_voxel_mean(pts, colors, n, voxel_size, out_pts, colors ? out_colors : nullptr, count_out, workspace,
                                          static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "voxel_downsample launch");
}

int d3f_mask_gate(const float *mask_channel, int64_t stride_y, int64_t stride_x, const float *depth, int32_t H, int32_t W,
                  float depth_lo, float depth_hi, uint8_t *out, void *stream)
{
    if (H < 0 || W < 0 || (int64_t)H * W > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "mask_gate: H=%d W=%d", H, W);
    if ((int64_t)H * W == 0) return D3F_OK;
    if (!mask_channel || !depth || !out) return fail(D3F_ERR_INVALID_ARG, "mask_gate: NULL pointer");
    hipError_t e = d3f::launch_mask_gate(mask_channel, stride_y, stride_x, depth, H, W, depth_lo, depth_hi, out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "mask_gate launch");
}

int d3f_nonzero_pixels(const uint8_t *image, int32_t H, int32_t W, int64_t capacity, int32_t *out_row_col, int64_t *count_out,
                       void *workspace, void *stream)
{
    if (H < 1 || W < 1 || (int64_t)H * W > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "nonzero_pixels: H=%d W=%d", H, W);
    if (!image || !count_out || !workspace || capacity < 0 || (capacity > 0 && !out_row_col))
        return fail(D3F_ERR_INVALID_ARG, "nonzero_pixels: NULL pointer or negative capacity");
    hipError_t e = d3f::launch_nonzero_pixels(image, H, W, capacity, out_row_col, count_out, static_cast<int64_t *>(workspace),
                                              static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "nonzero_pixels launch");
}

int d3f_fps_pixels(const int32_t *pts, int64_t n, int32_t k, int64_t init_idx, int64_t *out_idx, double *out_maxdist,
                   void *workspace, void *stream)
{
    if (n < 1 || k < 0) return fail(D3F_ERR_BAD_SHAPE, "fps_pixels: n=%lld must be >= 1 (fps_np asserts a non-empty set), k=%d", (long long)n, k);
    if (k == 0) return D3F_OK;
    if (!pts || !out_idx || !workspace) return fail(D3F_ERR_INVALID_ARG, "fps_pixels: NULL pointer");
    if (!aligned(workspace, 16)) return fail(D3F_ERR_BAD_LAYOUT, "fps_pixels: workspace must be 16-byte aligned");
    if (init_idx < 0 || init_idx >= n) return fail(D3F_ERR_INVALID_ARG, "fps_pixels: init_idx=%lld outside [0,%lld)", (long long)init_idx, (long long)n);
    hipError_t e = d3f::launch_fps_pixels(pts, n, k, init_idx, out_idx, out_maxdist, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "fps_pixels launch");
}

static int check_grid(const d3f_grid *g)
{
    if (!g) return fail(D3F_ERR_INVALID_ARG, "grid is NULL");
    if (g->nx < 0 || g->ny < 0 || g->nz < 0) return fail(D3F_ERR_BAD_SHAPE, "grid: negative size");
    if ((int64_t)g->nx * g->ny * g->nz > 0 && (!g->x || !g->y || !g->z)) return fail(D3F_ERR_INVALID_ARG, "grid: axis arrays must be non-NULL");
    return D3F_OK;
}

int d3f_eval_grid(const d3f_views *views, const d3f_grid *grid, const d3f_channel_map *maps, int32_t n_maps, float mu,
                  uint32_t flags, float *out_dist, uint8_t *out_valid, float *const *out_fused, void *stream)
{
    int rc = check_grid(grid);
    if (rc != D3F_OK) return rc;
    const int32_t dims[3] = {grid->nx, grid->ny, grid->nz};
    return eval_common(views, nullptr, (int64_t)grid->nx * grid->ny * grid->nz, maps, n_maps, mu, flags, out_dist, out_valid,
                       out_fused, nullptr, nullptr, 0, stream, 0, nullptr, grid, dims);
}

int d3f_eval_lattice(const d3f_views *views, const float *pts, int32_t nx, int32_t ny, int32_t nz, const d3f_channel_map *maps,
                     int32_t n_maps, float mu, uint32_t flags, float *out_dist, uint8_t *out_valid, float *const *out_fused,
                     float *const *out_inter, void *stream)
{
    if (nx < 0 || ny < 0 || nz < 0) return fail(D3F_ERR_BAD_SHAPE, "eval_lattice: negative size");
    const int32_t dims[3] = {nx, ny, nz};
    return eval_common(views, pts, (int64_t)nx * ny * nz, maps, n_maps, mu, flags, out_dist, out_valid, out_fused, out_inter,
                       nullptr, 0, stream, 0, nullptr, nullptr, dims);
}

int64_t d3f_grid_shell_workspace_bytes(const d3f_grid *grid)
{
    if (!grid || grid->nx <= 0 || grid->ny <= 0 || grid->nz <= 0) return 0;
    return d3f::grid_shell_workspace_bytes((int64_t)grid->nx * grid->ny * grid->nz);
}

int d3f_grid_shell(const d3f_views *views, const d3f_grid *grid, float mu, float dist_thr, int64_t capacity, int64_t *idx_out,
                   int64_t *count_out, void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_views(views);
    if (rc != D3F_OK) return rc;
    rc = check_grid(grid);
    if (rc != D3F_OK) return rc;
    if (!count_out || capacity < 0 || (capacity > 0 && !idx_out)) return fail(D3F_ERR_INVALID_ARG, "grid_shell: idx_out/count_out/capacity");
    if ((int64_t)grid->nx * grid->ny * grid->nz > 0 && (!workspace || workspace_bytes < d3f_grid_shell_workspace_bytes(grid)))
        return fail(D3F_ERR_WORKSPACE, "grid_shell: needs %lld workspace bytes", (long long)d3f_grid_shell_workspace_bytes(grid));
    if (!(mu > 0.0f)) return fail(D3F_ERR_INVALID_ARG, "mu must be > 0");
    if (((int64_t)grid->nx * grid->ny * grid->nz + d3f::kBlock - 1) / d3f::kBlock > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "grid too large for one launch");
    hipError_t e = d3f::launch_grid_shell(views->depth, views->K, views->pose, views->V, views->H, views->W, grid->x, grid->y,
                                          grid->z, grid->nx, grid->ny, grid->nz, mu, dist_thr, capacity, idx_out,
                                          reinterpret_cast<unsigned long long *>(count_out), workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "grid_shell launch");
}

int64_t d3f_fps_workspace_bytes(int64_t n) { return n > 0 ? d3f::fps_workspace_bytes(n, 4) : 0; }
int64_t d3f_fps_pixels_workspace_bytes(int64_t n) { return n > 0 ? d3f::fps_workspace_bytes(n, 8) : 0; }

int d3f_farthest_point_sampling(const float *pts, int64_t n, int32_t k, int64_t init_idx, int64_t *out_idx, float *out_maxdist,
                                void *workspace, void *stream)
{
    if (n < 1 || k < 0) return fail(D3F_ERR_BAD_SHAPE, "fps: n=%lld must be >= 1 (fps_np asserts a non-empty cloud), k=%d", (long long)n, k);
    if (k == 0) return D3F_OK;
    if (!pts || !out_idx || !workspace) return fail(D3F_ERR_INVALID_ARG, "fps: NULL pointer");
    if (!aligned(workspace, 16)) return fail(D3F_ERR_BAD_LAYOUT, "fps: workspace must be 16-byte aligned");
    if (init_idx < 0 || init_idx >= n) return fail(D3F_ERR_INVALID_ARG, "fps: init_idx=%lld outside [0,%lld)", (long long)init_idx, (long long)n);
    hipError_t e = d3f::launch_fps(pts, n, k, init_idx, out_idx, out_maxdist, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "fps launch");
}

static int backward_common(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps, int32_t n_maps,
                           float mu, const float *grad_dist, const float *const *grad_fused, float *grad_pts, void *stream,
                           int mode)
{
    int rc = check_views(views);
    if (rc != D3F_OK) return rc;
    if (n < 0) return fail(D3F_ERR_INVALID_ARG, "n=%lld is negative", (long long)n);
    if (n == 0) return D3F_OK;
    if (!pts || !grad_pts) return fail(D3F_ERR_INVALID_ARG, "pts/grad_pts must be non-NULL");
    if (n_maps < 0 || n_maps > D3F_MAX_MAPS) return fail(D3F_ERR_BAD_SHAPE, "n_maps=%d outside [0,%d]", n_maps, D3F_MAX_MAPS);
    if (n_maps > 0 && (!maps || !grad_fused)) return fail(D3F_ERR_INVALID_ARG, "maps/grad_fused must be non-NULL when n_maps > 0");
    if (!(mu > 0.0f)) return fail(D3F_ERR_INVALID_ARG, "mu must be > 0");
    d3f::BackwardParams P;
    P.depth = views->depth; P.K = views->K; P.pose = views->pose; P.pts = pts;
    P.grad_dist = grad_dist; P.grad_pts = grad_pts;
    P.n = n; P.V = views->V; P.H = views->H; P.W = views->W; P.n_maps = n_maps; P.mu = mu;
    int64_t map_bytes = 0;
    for (int s = 0; s < n_maps; ++s) {
        P.grad_fused[s] = grad_fused[s];
        rc = fill_map(maps[s], s, views->V, nullptr, nullptr, grad_fused[s], P.maps[s], map_bytes);
        if (rc != D3F_OK) return rc;
    }
    int t = 128;                       // LDS: 44 B per (point, view)
    while (t > 16 && (long)t * views->V * 44 > 60 * 1024) t >>= 1;
    while (t > 8 && n / t < 1024) t >>= 1;     // small batches: spread over many workgroups (latency, not throughput)
    P.tile_pts = t;
    if ((n + t - 1) / t > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "n=%lld needs more than 2^31 workgroups", (long long)n);
    hipError_t e = d3f::launch_fused_backward(P, mode, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "fused_eval_backward launch");
}

int d3f_eval_backward(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps, int32_t n_maps,
                      float mu, const float *grad_dist, const float *const *grad_fused, float *grad_pts, void *stream)
{
    return backward_common(views, pts, n, maps, n_maps, mu, grad_dist, grad_fused, grad_pts, stream, 0);
}

int d3f_eval_dist_backward(const d3f_views *views, const float *pts, int64_t n, const float *grad_dist, float *grad_pts,
                           void *stream)
{
    return backward_common(views, pts, n, nullptr, 0, 1.0f, grad_dist, nullptr, grad_pts, stream, 1);
}

int d3f_point_order_locality(const float *pts, int64_t n, float *out, void *stream)
{
    if (n < 0) return fail(D3F_ERR_INVALID_ARG, "n=%lld is negative", (long long)n);
    if (!out || (n > 0 && !pts)) return fail(D3F_ERR_INVALID_ARG, "point_order_locality: NULL pointer");
    hipError_t e = d3f::launch_point_locality(pts, n, out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "point locality launch");
}

int d3f_lattice_probe(const float *pts, int64_t n, int32_t *out_dims, void *stream)
{
    if (n < 0) return fail(D3F_ERR_INVALID_ARG, "n=%lld is negative", (long long)n);
    if (!out_dims || (n > 0 && !pts)) return fail(D3F_ERR_INVALID_ARG, "lattice_probe: NULL pointer");
    hipError_t e = d3f::launch_lattice_probe(pts, n, out_dims, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "lattice probe launch");
}

int d3f_eval_dist(const d3f_views *views, const float *pts, int64_t n, float *out_dist, uint8_t *out_valid,
                  void *stream)
{
    return eval_common(views, pts, n, nullptr, 0, 1.0f, 0u, out_dist, out_valid, nullptr, nullptr, nullptr, 0, stream, 1);
}

#ifdef D3F_EXPERIMENTS
// experiments builds only (not in the header): the phase stamps of the last stamped launch, 32 x uint64 per sampled workgroup
int d3f_exp_read_stamps(unsigned long long *host_out, int64_t n_words)
{
    unsigned long long *dev = exp_stamp_buffer(false);
    if (!dev || !host_out || n_words < 0 || n_words * 8 > kStampBytes) return D3F_ERR_INVALID_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return D3F_ERR_HIP;
    return hipMemcpy(host_out, dev, (size_t)n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? D3F_OK : D3F_ERR_HIP;
}
#endif

int d3f_map_check(const d3f_channel_map *map, int32_t V, uint32_t *word_out, void *stream)
{
    if (!map || !word_out) return fail(D3F_ERR_INVALID_ARG, "map_check: NULL pointer");
    if (!aligned(word_out, 4)) return fail(D3F_ERR_BAD_LAYOUT, "map_check: word_out must be 4-byte aligned");
    if (V < 1 || map->fh < 1 || map->fw < 1 || map->C < 1) return fail(D3F_ERR_BAD_SHAPE, "map_check: V=%d fh=%d fw=%d C=%d", V, map->fh, map->fw, map->C);
    if (!map->data) return fail(D3F_ERR_INVALID_ARG, "map_check: data pointer is NULL");
    if (map->dtype != D3F_DTYPE_F32 && map->dtype != D3F_DTYPE_F16) return fail(D3F_ERR_BAD_DTYPE, "map_check: dtype %d unsupported", map->dtype);
    const int es = map->dtype == D3F_DTYPE_F16 ? 2 : 4;
    if (!aligned(map->data, es)) return fail(D3F_ERR_BAD_LAYOUT, "map_check: data must be aligned to its element size");
    if (map->stride_x < map->C || map->stride_y < 0 || map->stride_v < 0) return fail(D3F_ERR_BAD_LAYOUT, "map_check: strides do not describe a channels-last map");
    hipError_t e = d3f::launch_map_check(map->data, V, map->fh, map->fw, map->C, map->stride_v, map->stride_y, map->stride_x, es,
                                         word_out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "map_check launch");
}

int d3f_onehot2instance(const float *onehot, int64_t n, int32_t NI, uint8_t *out, void *stream)
{
    if (n < 0 || NI < 1 || NI > 256) return fail(D3F_ERR_BAD_SHAPE, "onehot2instance: n=%lld NI=%d (NI must be in [1,256])", (long long)n, NI);
    if (n == 0) return D3F_OK;
    if (!onehot || !out) return fail(D3F_ERR_INVALID_ARG, "onehot2instance: NULL pointer");
    hipError_t e = d3f::launch_onehot2instance(onehot, n, NI, out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "onehot2instance launch");
}

int d3f_instance2onehot(const uint8_t *instance, int64_t n, int32_t NI, uint8_t *out_bool, void *stream)
{
    if (n < 0 || NI < 1 || NI > 256) return fail(D3F_ERR_BAD_SHAPE, "instance2onehot: n=%lld NI=%d (NI must be in [1,256])", (long long)n, NI);
    if (n == 0) return D3F_OK;
    if (!instance || !out_bool) return fail(D3F_ERR_INVALID_ARG, "instance2onehot: NULL pointer");
    hipError_t e = d3f::launch_instance2onehot(instance, n, NI, out_bool, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "instance2onehot launch");
}

int64_t d3f_softmax_workspace_bytes(int64_t rows, int64_t cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    const int64_t nchunks = (rows + d3f::kSoftmaxRowsPerBlock - 1) / d3f::kSoftmaxRowsPerBlock;
    return (nchunks + 1) * cols * (int64_t)sizeof(d3f::ColStat);
}

static int check_sim_enums(int32_t dist_type, int32_t mode)
{
    if (dist_type != D3F_DIST_L2 && dist_type != D3F_DIST_SQUARE) return fail(D3F_ERR_INVALID_ARG, "dist_type=%d (expected D3F_DIST_L2 or D3F_DIST_SQUARE)", dist_type);
    if (mode < D3F_SIM_DIST || mode > D3F_SIM_SOFTMAX_DIM0) return fail(D3F_ERR_INVALID_ARG, "mode=%d is not a D3F_SIM_* value", mode);
    return D3F_OK;
}

int d3f_similarity_to_target(const float *src, int64_t B, int64_t inner, int32_t C, int64_t stride_b,
                             int64_t stride_i, int64_t stride_c, const float *tgt, float scale, int32_t dist_type,
                             int32_t mode, float *out, void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_sim_enums(dist_type, mode);
    if (rc != D3F_OK) return rc;
    if (B < 0 || inner < 0 || C < 1) return fail(D3F_ERR_BAD_SHAPE, "similarity_to_target: B=%lld inner=%lld C=%d", (long long)B, (long long)inner, C);
    if (B == 0 || inner == 0) return D3F_OK;
    if (!src || !tgt || !out) return fail(D3F_ERR_INVALID_ARG, "similarity_to_target: NULL pointer");
    if (mode == D3F_SIM_SOFTMAX_DIM0 && (!workspace || workspace_bytes < d3f_softmax_workspace_bytes(B, inner)))
        return fail(D3F_ERR_WORKSPACE, "similarity_to_target: softmax needs %lld workspace bytes", (long long)d3f_softmax_workspace_bytes(B, inner));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = d3f::launch_dist_to_target(src, B, inner, C, stride_b, stride_i, stride_c, tgt, dist_type, out, s);
    if (e != hipSuccess) return hip_fail(e, "dist_to_target launch");
    if (mode == D3F_SIM_EXP) {
        e = d3f::launch_exp_neg_scale(out, B * inner, scale, s);
        if (e != hipSuccess) return hip_fail(e, "exp launch");
    } else if (mode == D3F_SIM_SOFTMAX_DIM0) {
        e = d3f::launch_softmax_dim0(out, B, inner, scale, nullptr, static_cast<d3f::ColStat *>(workspace), false, s);
        if (e != hipSuccess) return hip_fail(e, "softmax launch");
    }
    return D3F_OK;
}

int d3f_pairwise_similarity(const float *src, const float *tgt, int64_t B1, int64_t B2, int32_t C, float scale,
                            int32_t dist_type, int32_t mode, float *out, int64_t *argmax_out, void *workspace,
                            int64_t workspace_bytes, void *stream)
{
    int rc = check_sim_enums(dist_type, mode);
    if (rc != D3F_OK) return rc;
    if (B1 < 0 || B2 < 0 || C < 1) return fail(D3F_ERR_BAD_SHAPE, "pairwise: B1=%lld B2=%lld C=%d", (long long)B1, (long long)B2, C);
    if (B1 == 0 || B2 == 0) return D3F_OK;
    if (!src || !tgt || !out) return fail(D3F_ERR_INVALID_ARG, "pairwise: NULL pointer");
    if ((B2 + 63) / 64 > 0x7fffffffLL || (B1 + 63) / 64 > 65535) return fail(D3F_ERR_BAD_SHAPE, "pairwise: B1=%lld exceeds 64*65535 rows per call", (long long)B1);
    const bool need_ws = (mode == D3F_SIM_SOFTMAX_DIM0) || (argmax_out != nullptr);
    if (need_ws && (!workspace || workspace_bytes < d3f_softmax_workspace_bytes(B1, B2)))
        return fail(D3F_ERR_WORKSPACE, "pairwise: needs %lld workspace bytes", (long long)d3f_softmax_workspace_bytes(B1, B2));
    hipStream_t s = static_cast<hipStream_t>(stream);
    d3f::ColStat *ws = static_cast<d3f::ColStat *>(workspace);
    // the column statistics (softmax / best match) come out of the distance kernel's epilogue, per 64-row tile
    hipError_t e = d3f::launch_pairwise_dist(src, tgt, B1, B2, C, dist_type, out, s, need_ws ? ws : nullptr,
                                             mode == D3F_SIM_SOFTMAX_DIM0 ? scale : 1.0f);
    if (e != hipSuccess) return hip_fail(e, "pairwise_dist launch");
    if (mode == D3F_SIM_SOFTMAX_DIM0) {
        e = d3f::launch_softmax_dim0(out, B1, B2, scale, argmax_out, ws, true, s);
        if (e != hipSuccess) return hip_fail(e, "softmax launch");
    } else {
        if (argmax_out) {   // best match = smallest distance, decided before exp() can tie values
            e = d3f::launch_argmin_dim0(out, B1, B2, argmax_out, ws, true, s);
            if (e != hipSuccess) return hip_fail(e, "argmin launch");
        }
        if (mode == D3F_SIM_EXP) {
            e = d3f::launch_exp_neg_scale(out, B1 * B2, scale, s);
            if (e != hipSuccess) return hip_fail(e, "exp launch");
        }
    }
    return D3F_OK;
}

int64_t d3f_pairwise_topk_workspace_bytes(int64_t B1, int64_t B2)
{
    if (B1 <= 0 || B2 <= 0) return 0;
    return d3f_softmax_workspace_bytes(B1, B2) + d3f::topk_workspace_bytes(B1, B2) + 256;
}

int d3f_pairwise_similarity_topk(const float *src, const float *tgt, int64_t B1, int64_t B2, int32_t C, float scale,
                                 int32_t dist_type, int32_t mode, int32_t k, float *out, int64_t *topk_idx, float *topk_val,
                                 void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_sim_enums(dist_type, mode);
    if (rc != D3F_OK) return rc;
    if (k < 1 || k > 8) return fail(D3F_ERR_INVALID_ARG, "pairwise_topk: k=%d outside [1,8]", k);
    if (B1 < 0 || B2 < 0 || C < 1) return fail(D3F_ERR_BAD_SHAPE, "pairwise_topk: B1=%lld B2=%lld C=%d", (long long)B1, (long long)B2, C);
    if (B2 == 0) return D3F_OK;
    if (!topk_idx) return fail(D3F_ERR_INVALID_ARG, "pairwise_topk: topk_idx is NULL");
    if (B1 > 0 && (!src || !tgt || !out)) return fail(D3F_ERR_INVALID_ARG, "pairwise_topk: NULL pointer");
    if (B1 > 0x7fffffffLL || (B2 + 63) / 64 > 0x7fffffffLL || (B1 + 63) / 64 > 65535)
        return fail(D3F_ERR_BAD_SHAPE, "pairwise_topk: B1=%lld exceeds 64*65535 rows per call", (long long)B1);
    if (B1 > 0 && (!workspace || workspace_bytes < d3f_pairwise_topk_workspace_bytes(B1, B2) || !aligned(workspace, 16)))
        return fail(D3F_ERR_WORKSPACE, "pairwise_topk: needs %lld bytes of 16-byte aligned workspace", (long long)d3f_pairwise_topk_workspace_bytes(B1, B2));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const void *final_list = nullptr;
    if (B1 > 0) {
        d3f::ColStat *ws = static_cast<d3f::ColStat *>(workspace);
        unsigned char *tk = static_cast<unsigned char *>(workspace) + (d3f_softmax_workspace_bytes(B1, B2) + 255) / 256 * 256;
        const bool stats = mode == D3F_SIM_SOFTMAX_DIM0;
        e = d3f::launch_pairwise_dist(src, tgt, B1, B2, C, dist_type, out, s, stats ? ws : nullptr, stats ? scale : 1.0f);
        if (e != hipSuccess) return hip_fail(e, "pairwise_dist launch");
        // neighbours are chosen on the raw distances, before exp() / softmax can round close values into ties
        e = d3f::launch_topk_select(out, B1, B2, tk, &final_list, s);
        if (e != hipSuccess) return hip_fail(e, "topk launch");
        if (mode == D3F_SIM_SOFTMAX_DIM0) e = d3f::launch_softmax_dim0(out, B1, B2, scale, nullptr, ws, true, s);
        else if (mode == D3F_SIM_EXP) e = d3f::launch_exp_neg_scale(out, B1 * B2, scale, s);
        if (e != hipSuccess) return hip_fail(e, "similarity launch");
    }
    e = d3f::launch_topk_write(final_list, out, B1, B2, k, topk_idx, topk_val, s);
    return e == hipSuccess ? D3F_OK : hip_fail(e, "topk write launch");
}

static_assert(sizeof(d3f_col_stat) == sizeof(d3f::ColStat) && sizeof(d3f_col_stat) == 16, "d3f_col_stat layout");

int d3f_topk_smallest(const float *x, int64_t rows, int64_t cols, int32_t k, int64_t *idx_out, float *val_out, void *workspace,
                      int64_t workspace_bytes, void *stream)
{
    if (k < 1 || k > 8) return fail(D3F_ERR_INVALID_ARG, "topk_smallest: k=%d outside [1,8]", k);
    if (rows < 0 || cols < 0 || rows > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "topk_smallest: rows=%lld cols=%lld", (long long)rows, (long long)cols);
    if (cols == 0) return D3F_OK;
    if (!idx_out || (rows > 0 && !x)) return fail(D3F_ERR_INVALID_ARG, "topk_smallest: NULL pointer");
    if (rows > 0 && (!workspace || workspace_bytes < d3f::topk_workspace_bytes(rows, cols) || !aligned(workspace, 16)))
        return fail(D3F_ERR_WORKSPACE, "topk_smallest: needs %lld bytes of 16-byte aligned workspace", (long long)d3f::topk_workspace_bytes(rows, cols));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const void *final_list = nullptr;
    if (rows > 0) {
        hipError_t e = d3f::launch_topk_select(x, rows, cols, workspace, &final_list, s);
        if (e != hipSuccess) return hip_fail(e, "topk launch");
    }
    hipError_t e = d3f::launch_topk_write(final_list, x, rows, cols, k, idx_out, val_out, s);
    return e == hipSuccess ? D3F_OK : hip_fail(e, "topk write launch");
}

int d3f_topk_merge(const int64_t *parts_idx, const float *parts_val, int64_t n_parts, int32_t k, int64_t cols, int64_t *out_idx,
                   float *out_val, void *stream)
{
    if (k < 1 || k > 8) return fail(D3F_ERR_INVALID_ARG, "topk_merge: k=%d outside [1,8]", k);
    if (n_parts < 0 || cols < 0) return fail(D3F_ERR_BAD_SHAPE, "topk_merge: n_parts=%lld cols=%lld", (long long)n_parts, (long long)cols);
    if (cols == 0) return D3F_OK;
    if (!out_idx || (n_parts > 0 && (!parts_idx || !parts_val))) return fail(D3F_ERR_INVALID_ARG, "topk_merge: NULL pointer");
    hipError_t e = d3f::launch_topk_merge_parts(parts_idx, parts_val, n_parts, k, cols, out_idx, out_val, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "topk merge launch");
}

int d3f_pairwise_softmax_local(const float *src, const float *tgt, int64_t B1, int64_t B2, int32_t C, float scale,
                               int32_t dist_type, int64_t row_offset, float *out, d3f_col_stat *stats, void *workspace,
                               int64_t workspace_bytes, void *stream)
{
    int rc = check_sim_enums(dist_type, D3F_SIM_SOFTMAX_DIM0);
    if (rc != D3F_OK) return rc;
    if (B1 < 0 || B2 < 0 || C < 1 || row_offset < 0) return fail(D3F_ERR_BAD_SHAPE, "pairwise_softmax_local: B1=%lld B2=%lld C=%d row_offset=%lld", (long long)B1, (long long)B2, C, (long long)row_offset);
    if (B2 == 0) return D3F_OK;
    if (!stats || !tgt) return fail(D3F_ERR_INVALID_ARG, "pairwise_softmax_local: NULL pointer");
    if (B1 > 0 && (!src || !out)) return fail(D3F_ERR_INVALID_ARG, "pairwise_softmax_local: NULL pointer");
    if ((B2 + 63) / 64 > 0x7fffffffLL || (B1 + 63) / 64 > 65535) return fail(D3F_ERR_BAD_SHAPE, "pairwise_softmax_local: B1=%lld exceeds 64*65535 rows per call", (long long)B1);
    if (B1 > 0 && (!workspace || workspace_bytes < d3f_softmax_workspace_bytes(B1, B2)))
        return fail(D3F_ERR_WORKSPACE, "pairwise_softmax_local: needs %lld workspace bytes", (long long)d3f_softmax_workspace_bytes(B1, B2));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    if (B1 > 0) {
        e = d3f::launch_pairwise_dist(src, tgt, B1, B2, C, dist_type, out, s, static_cast<d3f::ColStat *>(workspace), scale);
        if (e != hipSuccess) return hip_fail(e, "pairwise_dist launch");
    }
    e = d3f::launch_softmax_local_stats(out, B1, B2, scale, row_offset, static_cast<d3f::ColStat *>(workspace),
                                        reinterpret_cast<d3f::ColStat *>(stats), true, s);
    return e == hipSuccess ? D3F_OK : hip_fail(e, "softmax statistics launch");
}

int d3f_softmax_merge(const d3f_col_stat *parts, int64_t n_parts, int64_t cols, d3f_col_stat *merged, int64_t *argmax_out,
                      void *stream)
{
    if (n_parts < 0 || cols < 0) return fail(D3F_ERR_BAD_SHAPE, "softmax_merge: n_parts=%lld cols=%lld", (long long)n_parts, (long long)cols);
    if (cols == 0) return D3F_OK;
    if (!merged || (n_parts > 0 && !parts)) return fail(D3F_ERR_INVALID_ARG, "softmax_merge: NULL pointer");
    hipError_t e = d3f::launch_softmax_merge(reinterpret_cast<const d3f::ColStat *>(parts), n_parts, cols,
                                             reinterpret_cast<d3f::ColStat *>(merged), argmax_out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "softmax merge launch");
}

int d3f_softmax_apply(float *x, int64_t rows, int64_t cols, float scale, const d3f_col_stat *merged, void *stream)
{
    if (rows < 0 || cols < 0) return fail(D3F_ERR_BAD_SHAPE, "softmax_apply: rows=%lld cols=%lld", (long long)rows, (long long)cols);
    if (rows == 0 || cols == 0) return D3F_OK;
    if (!x || !merged) return fail(D3F_ERR_INVALID_ARG, "softmax_apply: NULL pointer");
    hipError_t e = d3f::launch_softmax_apply(x, rows, cols, scale, reinterpret_cast<const d3f::ColStat *>(merged),
                                             static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "softmax apply launch");
}

// ---- rigid tracking step (fusion.py:1643-1665) -----------------------------------------------------------------------
int d3f_rigid_transform(const float *last, int32_t n_inst, int32_t n, const float *t, const float *w, float *out_pts,
                        float *norms, void *stream)
{
    if (n_inst < 0 || n < 0 || (int64_t)n_inst * n > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "rigid_transform: n_inst=%d n=%d", n_inst, n);
    if (!norms || (n_inst > 0 && (!t || !w)) || ((int64_t)n_inst * n > 0 && (!last || !out_pts)))
        return fail(D3F_ERR_INVALID_ARG, "rigid_transform: NULL pointer");
    hipError_t e = d3f::launch_rigid_transform(last, n_inst, n, t, w, 1e-4f, out_pts, norms, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "rigid_transform launch");
}

int d3f_track_loss_grad(const float *feats, const float *src, const float *dist, const uint8_t *valid, int64_t N, int32_t C,
                        float dist_w, float *grad_feats, float *grad_dist, float *loss, void *stream)
{
    if (N < 0 || N > 0x7fffffffLL || C < 1) return fail(D3F_ERR_BAD_SHAPE, "track_loss_grad: N=%lld C=%d", (long long)N, C);
    if (!loss || (N > 0 && (!feats || !src || !dist || !valid || !grad_feats || !grad_dist)))
        return fail(D3F_ERR_INVALID_ARG, "track_loss_grad: NULL pointer");
    hipError_t e = d3f::launch_track_loss_grad(feats, src, dist, valid, (int)N, C, dist_w, grad_feats, grad_dist, loss,
                                               static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "track_loss_grad launch");
}

int d3f_rigid_update(const float *last, int32_t n_inst, int32_t n, const float *grad_pts, float *t, float *w, float *adam_m,
                     float *adam_v, float *step, const float *norms, float reg_w, float lr, float beta1, float beta2, float eps,
                     void *stream)
{
    if (n_inst < 0 || n < 0) return fail(D3F_ERR_BAD_SHAPE, "rigid_update: n_inst=%d n=%d", n_inst, n);
    if (n_inst == 0) return D3F_OK;
    if (!t || !w || !adam_m || !adam_v || !step || !norms || (n > 0 && (!last || !grad_pts)))
        return fail(D3F_ERR_INVALID_ARG, "rigid_update: NULL pointer");
    if (!(lr > 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps > 0.0f))
        return fail(D3F_ERR_INVALID_ARG, "rigid_update: lr=%g beta=(%g,%g) eps=%g", lr, beta1, beta2, eps);
    hipError_t e = d3f::launch_rigid_update(last, n_inst, n, grad_pts, t, w, adam_m, adam_v, step, norms, 1e-4f, reg_w, lr, beta1,
                                            beta2, eps, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "rigid_update launch");
}

int64_t d3f_track_step_scratch_bytes(int32_t n_inst, int32_t n)
{
    if (n_inst < 0 || n < 0) return 0;
    // gradients [n_inst*n*3], loss slots [4], arrival counter [2] (floats / words), then the tagged parameter words [n_inst*6] (8 bytes each)
    return (((int64_t)n_inst * n * 3 + 6) * (int64_t)sizeof(float) + 7) / 8 * 8 + (int64_t)n_inst * 6 * 8;
}

static int track_impl(const d3f_views *views, const d3f_channel_map *descriptors, const float *last, int32_t n_inst, int32_t n,
                      const float *src, float mu, float dist_w, float reg_w, float lr, float beta1, float beta2, float eps,
                      int32_t iters, const d3f_track_state *state, void *stream)
{
    int rc = check_views(views);
    if (rc != D3F_OK) return rc;
    if (n_inst < 0 || n < 0) return fail(D3F_ERR_BAD_SHAPE, "track_step: n_inst=%d n=%d", n_inst, n);
    if (iters < 0) return fail(D3F_ERR_INVALID_ARG, "track_run: iters=%d", iters);
    if ((int64_t)n_inst * n == 0 || iters == 0) return D3F_OK;
    if ((int64_t)n_inst * n > 0x7fffffLL) return fail(D3F_ERR_BAD_SHAPE, "track_step: %lld keypoints (one workgroup each) are too many", (long long)n_inst * n);
    if (iters > 1 && ((int64_t)n_inst * n > d3f::track_run_capacity() || n_inst > 16))
        return fail(D3F_ERR_BAD_SHAPE, "track_run: %lld keypoints of %d instances; the steps of one launch wait for one another, so every "
                                       "workgroup must be resident (<= %d keypoints on this device, <= 16 instances); call d3f_track_step per iteration",
                    (long long)n_inst * n, n_inst, d3f::track_run_capacity());
    if (!descriptors || !last || !src || !state) return fail(D3F_ERR_INVALID_ARG, "track_step: NULL pointer");
    if (!state->t || !state->w || !state->adam_m || !state->adam_v || !state->step || !state->out_pts || !state->loss || !state->scratch)
        return fail(D3F_ERR_INVALID_ARG, "track_step: NULL pointer in d3f_track_state");
    if (!(mu > 0.0f)) return fail(D3F_ERR_INVALID_ARG, "mu must be > 0");
    if (!(lr > 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps > 0.0f))
        return fail(D3F_ERR_INVALID_ARG, "track_step: lr=%g beta=(%g,%g) eps=%g", lr, beta1, beta2, eps);
    if (views->V > 8) return fail(D3F_ERR_BAD_SHAPE, "track_step: at most 8 views (V=%d); use the five-launch step", views->V);
    d3f::TrackStepParams P;
    int64_t map_bytes = 0;
    rc = fill_map(*descriptors, 0, views->V, nullptr, nullptr, nullptr, P.map, map_bytes);
    if (rc != D3F_OK) return rc;
    if (P.map.esize != 4 || P.map.C % 4 != 0 || P.map.C > 512 || (P.map.sx % 4) || (P.map.sy % 4) || (P.map.sv % 4) ||
        !aligned(descriptors->data, 16) || !aligned(src, 16))
        return fail(D3F_ERR_BAD_LAYOUT, "track_step: the descriptor map must be fp32 with C %% 4 == 0, C <= 512 and 16-byte aligned texels "
                                        "(C=%d); use the five-launch step", P.map.C);
    P.depth = views->depth; P.K = views->K; P.pose = views->pose; P.V = views->V; P.H = views->H; P.W = views->W;
    P.last = last; P.src = src; P.I = n_inst; P.n = n; P.iters = iters;
    P.mu = mu; P.dist_w = dist_w; P.reg_w = reg_w; P.lr = lr; P.beta1 = beta1; P.beta2 = beta2; P.eps_adam = eps; P.eps_rot = 1e-4f;
    P.ln_beta1 = d3f::log_of_decimal(beta1); P.ln_beta2 = d3f::log_of_decimal(beta2);
    P.t = state->t; P.w = state->w; P.adam_m = state->adam_m; P.adam_v = state->adam_v; P.step = state->step;
    P.out_pts = state->out_pts; P.loss_out = state->loss;
    float *scr = static_cast<float *>(state->scratch);
    P.grad_pts = scr; P.loss_acc = scr + (int64_t)n_inst * n * 3; P.counter = reinterpret_cast<unsigned int *>(P.loss_acc + 4);
    P.par = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(state->scratch) + (((int64_t)n_inst * n * 3 + 6) * 4 + 7) / 8 * 8);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    hipError_t e = d3f::launch_track_step(P, hs);
    return e == hipSuccess ? D3F_OK : hip_fail(e, "track_step launch");
}

int d3f_track_step(const d3f_views *views, const d3f_channel_map *descriptors, const float *last, int32_t n_inst, int32_t n,
                   const float *src, float mu, float dist_w, float reg_w, float lr, float beta1, float beta2, float eps,
                   const d3f_track_state *state, void *stream)
{
    return track_impl(views, descriptors, last, n_inst, n, src, mu, dist_w, reg_w, lr, beta1, beta2, eps, 1, state, stream);
}

int d3f_track_run(const d3f_views *views, const d3f_channel_map *descriptors, const float *last, int32_t n_inst, int32_t n,
                  const float *src, float mu, float dist_w, float reg_w, float lr, float beta1, float beta2, float eps,
                  int32_t iters, const d3f_track_state *state, void *stream)
{
    return track_impl(views, descriptors, last, n_inst, n, src, mu, dist_w, reg_w, lr, beta1, beta2, eps, iters, state, stream);
}

int32_t d3f_track_run_max_keypoints(void) { return d3f::track_run_capacity(); }

}  // extern "C"
