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

#ifdef D3F_EXPERIMENTS
// phase stamps of the window kernel (Tune::stamps): 32 x uint64 per sampled workgroup, read back with d3f_exp_read_stamps
constexpr int64_t kStampBytes = 8LL * 32 * 65536;
unsigned long long *exp_stamp_buffer(bool clear)
{
    static unsigned long long *buf = nullptr;
    if (!buf && hipMalloc(reinterpret_cast<void **>(&buf), kStampBytes) != hipSuccess) buf = nullptr;
    if (buf && clear) (void)hipMemset(buf, 0, kStampBytes);
    return buf;
}
#endif

// the launch planner: thresholds, the family table, one function per kernel family (host logic only)
#include "d3f_plan.h"

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

constexpr int kGatedSecondPass = 1;          // eval_common's internal "now enqueue the other side" status (never returned to callers)

// Validates, fills the kernel parameters, PLANS (d3f_plan.h), builds / reuses the point order, launches.
// cloud_side (d3f_eval): 0 = plan queries and the callers that never gate; 1 = first pass of a query -- if the points are a cloud the
// window kernel may take (kWindowCloudMin points, a patch-resolution wide map, the Hilbert order), this pass enqueues order +
// probe + the GATED window launch and returns kGatedSecondPass; 2 = the second pass: the gated cell-run launch.
int eval_common(const d3f_views *views, const float *pts, int64_t n, const d3f_channel_map *maps, int32_t n_maps,
                float mu, uint32_t flags, float *out_dist, uint8_t *out_valid, float *const *out_fused,
                float *const *out_inter, void *workspace, int64_t workspace_bytes, void *stream, int mode,
                d3f_eval_plan *plan_out = nullptr, const d3f_grid *grid = nullptr, const int32_t *lattice = nullptr, int cloud_side = 0)
{
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
    const Tune tune = load_tune();          // every knob 0 in the product build (d3f_plan.h)
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
    for (int k = 0; k < P.n_words; ++k)
        if (!aligned(P.words[k], 4)) return fail(D3F_ERR_BAD_LAYOUT, "nonfinite word %d: device pointer must be 4-byte aligned", k);
    P.exp_stamps = nullptr;
#ifdef D3F_EXPERIMENTS
    if (tune.stamps > 0 && !plan_out) P.exp_stamps = exp_stamp_buffer(true);
#endif
    P.depth = views->depth; P.K = views->K; P.pose = views->pose; P.pts = pts;
    P.order = nullptr; P.lds_pad = 0;
    P.grid_x = grid ? grid->x : nullptr; P.grid_y = grid ? grid->y : nullptr; P.grid_z = grid ? grid->z : nullptr;
    P.grid_ny = grid ? grid->ny : 0; P.grid_nz = grid ? grid->nz : 0;
    P.walk_nx = P.walk_ny = P.walk_nz = 0; P.walk_tx = P.walk_ty = P.walk_tz = 1;
    P.sl_unit = 128; P.sl_ilv = 1; P.sl_slices = 0; P.sl_lg = 3; P.sl_vc = 4; P.sl_tiles = P.sl_groups = P.sl_chunks = 0;
    P.runs_occ = tune.runs_occ;
    P.dist_variant = tune.dist;
    P.depth_tiled = nullptr; P.depth_tw = P.depth_th = 0;
    P.thin_max_views = (tune.thin < 0 || (flags & D3F_TUNE_DIRECT_GATHER)) ? 0 : 8;
    P.win_lpp = tune.window_lpp == 32 ? 32 : 16;     // 16 lanes x 2 vectors per point (C2 patch 0.565 -> 0.54 ms); U > 1: 32
    P.win_slices = 0; P.win_u = 1; P.win_vc = 1; P.win_pool_offset = 0; P.win_pool_texels = 0; P.win_occ = 4;
    P.win_pipe = tune.window_pipe < 0 ? 0 : 1;
    P.win_sparse = 0;
    P.rows = 0;
    P.gate = nullptr; P.gate_min = 0u; P.gate_want = 0;
    P.store_policy = tune.store < 0 ? 0 : (tune.store == 1 ? 1 : (tune.store == 3 ? 3 : 2));     // non-temporal rows (fuse_common.h: store_out)
    P.out_dist = out_dist; P.out_valid = out_valid;
    P.n = n; P.V = views->V; P.H = views->H; P.W = views->W;
    P.n_maps = n_maps; P.tile_pts = tile_points_for(views->V);
    P.flags = flags; P.mu = mu;

    Query q;
    q.views = views; q.n = n; q.n_maps = n_maps; q.flags = flags; q.mode = mode; q.lattice = lattice; q.grid = grid != nullptr;
    q.plan_only = plan_only; q.cloud_side = cloud_side; q.tune = tune;
    q.finite_expected = (flags & D3F_FLAG_FINITE_MAPS) || P.n_words > 0;
    q.direct = (flags & D3F_TUNE_DIRECT_GATHER) != 0;     // the plain direct gather in the chosen point order: the reference of the bit-identity tests
    q.tl = (int)((flags >> 8) & 0xF);
    q.map_bytes = 0;
    for (int s = 0; s < n_maps; ++s) {
        if (!plan_only && !out_fused[s]) return fail(D3F_ERR_INVALID_ARG, "map %d: output pointer is NULL", s);
        rc = fill_map(maps[s], s, views->V, out_fused ? out_fused[s] : nullptr, out_inter ? out_inter[s] : nullptr, nullptr,
                      P.maps[s], q.map_bytes, flags);
        if (rc != D3F_OK) return rc;
    }
    // The channel-sliced and the LDS-window kernels want THE wide map first (the thin ones ride along).  A map descriptor
    // carries its own output pointers, so the launch may take the maps in any order: return_names=['mask', 'dino_feats']
    // gets the same kernels as the reference's default ['dino_feats', 'mask'].  caller_map[k] = caller's index of P.maps[k].
    int caller_map[D3F_MAX_MAPS];
    for (int s = 0; s < D3F_MAX_MAPS; ++s) { caller_map[s] = s; q.want_inter[s] = false; }
    {
        int wide = -1, nwide = 0;
        for (int s = 0; s < n_maps; ++s)
            if ((int64_t)P.maps[s].C * P.maps[s].esize > 256) { if (wide < 0) wide = s; ++nwide; }
        if (nwide == 1 && wide > 0) {
            const d3f::MapDesc t = P.maps[0]; P.maps[0] = P.maps[wide]; P.maps[wide] = t;
            caller_map[0] = wide; caller_map[wide] = 0;
        }
        for (int s = 0; s < n_maps; ++s) q.want_inter[s] = out_inter && out_inter[caller_map[s]];
    }
    // Hilbert point order (performance only) when scratch is supplied
    hipStream_t hs = static_cast<hipStream_t>(stream);
    q.may_reorder = (workspace || plan_only) && !grid && n_maps > 0 && n <= 0x7fffffffLL && !(flags & D3F_TUNE_NO_REORDER) &&
                    workspace_bytes >= d3f::order_workspace_bytes(n);

    // ---- the plan: family + point order, then (device work) the order itself, then the geometry ----
    Plan pl;
    plan_family_and_order(q, P, pl);
    if (!pl.walk && pl.reorder && !plan_only) {
        if (flags & D3F_FLAG_REUSE_POINT_ORDER) {
            P.order = d3f::stored_point_order(workspace, n);       // written by an earlier call for the same points
        } else {
            // Hilbert order of the cells of a 512^3 grid over the cloud's box, exact (order_kernels.hip); experiments builds: D3F_EXP_ORDER_MORTON=1 = the Z curve of rounds 1-4
            hipError_t eo = d3f::build_point_order(pts, n, workspace, workspace_bytes, &P.order, hs,
                                                   (tune.order_morton > 0 ? 1 : 0) | (tune.scan3 > 0 ? 2 : 0) | (tune.order_fixed_grid > 0 ? 4 : 0) | (tune.order_bits > 0 ? tune.order_bits << 8 : 0));
            if (eo != hipSuccess) return hip_fail(eo, "point ordering");
        }
    }
    plan_geometry(q, P, pl);
    const int64_t ntiles = plan_workgroups(P, pl, n);
    if (ntiles > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "n=%lld needs more than 2^31 workgroups", (long long)n);
    // the window and the channel-sliced kernels keep a point's global index in 32 bits of LDS (their rows already require
    // n < 2^31; this is the guard that says so)
    if ((pl.window || pl.rows || P.sl_slices > 0) && n > 0x7fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "n=%lld: 32-bit point indices", (long long)n);
    if (plan_only) {
        plan_out->gated_window = 0; plan_out->reserved2 = 0;
        if (cloud_side == 0 && !lattice && !grid) {       // would d3f_eval's first pass take the window side?  (its plan: the lattice's)
            d3f_eval_plan side;
            if (eval_common(views, pts, n, maps, n_maps, mu, flags, out_dist, out_valid, out_fused, out_inter, workspace, workspace_bytes,
                            stream, mode, &side, grid, lattice, 1) == D3F_OK)
                for (int s = 0; s < n_maps; ++s)
                    if (side.staged[s] == 3 && side.reorder == 1) { plan_out->gated_window = 1; plan_out->reserved2 = side.reserved; }
        }
        report_plan(P, pl, caller_map, n_maps, ntiles, plan_out);
        return D3F_OK;
    }
    // a cloud on the gated pair of launches: the window side (this pass) and the cell-run side (the next) read one device word
    const bool gated_window = cloud_side == 1 && pl.window && !pl.walk && P.order != nullptr;
    const bool gated_runs = cloud_side == 2;
    if (gated_window || gated_runs) {
        P.gate = d3f::order_gate_words(workspace, n);
        P.gate_min = (uint32_t)D3F_GATE_MIN_FIT;                   // three quarters of the sampled tiles fit their pool
        if ((flags & D3F_TUNE_WINDOW_SIDE) || tune.gate > 0) P.gate_min = 0u;          // always the window side
        if (tune.gate < 0) P.gate_min = 0xffffffffu;                                    // experiments: always the cell runs
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
    // the distance-only pass over a big batch in caller order, with scratch: the depth maps are tiled first (inside the timed pair)
    if (n_maps == 0 && n >= d3f::kDistTiledMin && !P.order && P.walk_nx <= 0 && !P.grid_x && views->V <= 8 && tune.dist >= 0 && !(tune.dist & 32) &&
        workspace && workspace_bytes >= d3f::depth_tiled_bytes(views->V, views->H, views->W) &&
        d3f::depth_tiled_bytes(views->V, views->H, views->W) < (1LL << 32)) {         // (32-bit pixel indices in the tiled copy)
        P.depth_tw = (views->W + 3) / 4; P.depth_th = (views->H + 7) / 8;
        hipError_t et = d3f::launch_depth_tiles(P, static_cast<float *>(workspace), hs);
        if (et != hipSuccess) return hip_fail(et, "depth tiling");
        P.depth_tiled = static_cast<const float *>(workspace);
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
int64_t d3f_eval_dist_workspace_bytes(const d3f_views *views, int64_t n)
{
    if (!views || n < d3f::kDistTiledMin || views->V < 1 || views->V > 8 || views->H < 1 || views->W < 1) return 0;
    const int64_t bytes = d3f::depth_tiled_bytes(views->V, views->H, views->W);
    return bytes < (1LL << 32) ? bytes : 0;
}

const char *d3f_plan_family_name(int32_t family) { return (family >= 0 && family < (int32_t)(sizeof(kFamilies) / sizeof(kFamilies[0]))) ? kFamilies[family].name : nullptr; }
const char *d3f_plan_family_takes(int32_t family) { return (family >= 0 && family < (int32_t)(sizeof(kFamilies) / sizeof(kFamilies[0]))) ? kFamilies[family].takes : nullptr; }

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

int d3f_compose_labels(const uint8_t *dets, int32_t n_dets, int64_t n_pix, const int32_t *label_of_det, uint8_t *out, void *stream)
{
    if (n_dets < 0 || n_pix < 0 || n_pix > 0x3fffffffLL) return fail(D3F_ERR_BAD_SHAPE, "compose_labels: n_dets=%d n_pix=%lld", n_dets, (long long)n_pix);
    if (n_pix == 0) return D3F_OK;
    if (!out || (n_dets > 0 && (!dets || !label_of_det))) return fail(D3F_ERR_INVALID_ARG, "compose_labels: NULL pointer");
    if (!aligned(label_of_det, 4)) return fail(D3F_ERR_BAD_LAYOUT, "compose_labels: label_of_det must be 4-byte aligned");
    hipError_t e = d3f::launch_compose_labels(dets, n_dets, n_pix, label_of_det, out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "compose_labels launch");
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
    // a workspace with d3f_eval_dist_workspace_bytes(views, n) bytes MORE than the shell needs: the depth lookups go to a tiled copy there
    const int64_t n_grid = (int64_t)grid->nx * grid->ny * grid->nz, shell_bytes = (d3f_grid_shell_workspace_bytes(grid) + 255) / 256 * 256;
    const int64_t tiled_bytes = d3f_eval_dist_workspace_bytes(views, n_grid);
    float *tiled = (tiled_bytes > 0 && workspace_bytes >= shell_bytes + tiled_bytes)
                       ? reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + shell_bytes) : nullptr;
    hipError_t e = d3f::launch_grid_shell(views->depth, views->K, views->pose, views->V, views->H, views->W, grid->x, grid->y,
                                          grid->z, grid->nx, grid->ny, grid->nz, mu, dist_thr, capacity, idx_out,
                                          reinterpret_cast<unsigned long long *>(count_out), workspace, static_cast<hipStream_t>(stream),
                                          tiled);
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

int d3f_points_probe(const float *pts, int64_t n, int32_t *out, void *stream)
{
    if (n < 0) return fail(D3F_ERR_INVALID_ARG, "n=%lld is negative", (long long)n);
    if (!out || (n > 0 && !pts)) return fail(D3F_ERR_INVALID_ARG, "points_probe: NULL pointer");
    hipError_t e = d3f::launch_points_probe(pts, n, out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "points probe launch");
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

int d3f_map_check_many(const d3f_channel_map *maps, const int32_t *views, int32_t n, uint32_t *const *words_out, uint32_t flags, void *stream)
{
    if (n < 0 || n > D3F_MAX_MAPS + 1) return fail(D3F_ERR_BAD_SHAPE, "map_check_many: n=%d outside [0,%d]", n, D3F_MAX_MAPS + 1);
    if (n == 0) return D3F_OK;
    if (!maps || !views || !words_out) return fail(D3F_ERR_INVALID_ARG, "map_check_many: NULL pointer");
    const void *data[D3F_MAX_MAPS + 1];
    int64_t nbytes[D3F_MAX_MAPS + 1];
    int esize[D3F_MAX_MAPS + 1];
    uint32_t *words[D3F_MAX_MAPS + 1];
    int nf = 0;
    const bool zero = (flags & D3F_CHECK_WORDS_ARE_ZERO) != 0;
    for (int k = 0; k < n; ++k) {
        const d3f_channel_map &m = maps[k];
        // the same validation as d3f_map_check; a tensor that is not one flat 16-byte aligned block takes the single-tensor call
        if (!words_out[k] || !aligned(words_out[k], 4)) return fail(D3F_ERR_BAD_LAYOUT, "map_check_many: word %d must be a 4-byte aligned device pointer", k);
        if (views[k] < 1 || m.fh < 1 || m.fw < 1 || m.C < 1) return fail(D3F_ERR_BAD_SHAPE, "map_check_many: tensor %d: V=%d fh=%d fw=%d C=%d", k, views[k], m.fh, m.fw, m.C);
        if (!m.data) return fail(D3F_ERR_INVALID_ARG, "map_check_many: tensor %d: data pointer is NULL", k);
        if (m.dtype != D3F_DTYPE_F32 && m.dtype != D3F_DTYPE_F16) return fail(D3F_ERR_BAD_DTYPE, "map_check_many: tensor %d: dtype %d unsupported", k, m.dtype);
        const int es = m.dtype == D3F_DTYPE_F16 ? 2 : 4;
        if (!aligned(m.data, es)) return fail(D3F_ERR_BAD_LAYOUT, "map_check_many: tensor %d: data must be aligned to its element size", k);
        if (m.stride_x < m.C || m.stride_y < 0 || m.stride_v < 0) return fail(D3F_ERR_BAD_LAYOUT, "map_check_many: tensor %d: strides do not describe a channels-last map", k);
        if (d3f::map_is_flat(m.data, views[k], m.fh, m.fw, m.C, m.stride_v, m.stride_y, m.stride_x)) {
            data[nf] = m.data; nbytes[nf] = (int64_t)views[k] * m.fh * m.fw * m.C * es; esize[nf] = es; words[nf] = words_out[k];
            ++nf;
        } else {
            hipError_t e = d3f::launch_map_check(m.data, views[k], m.fh, m.fw, m.C, m.stride_v, m.stride_y, m.stride_x, es, words_out[k],
                                                 static_cast<hipStream_t>(stream));
            if (e != hipSuccess) return hip_fail(e, "map_check launch");
        }
    }
    hipError_t e = d3f::launch_map_check_many(data, nbytes, esize, words, nf, zero, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? D3F_OK : hip_fail(e, "map_check_many launch");
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

int64_t d3f_track_stall_word(int32_t n_inst, int32_t n) { return (n_inst < 0 || n < 0) ? 0 : (int64_t)n_inst * n * 3 + 5; }      // counter[1]

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
