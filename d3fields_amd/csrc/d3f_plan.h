// d3f_plan.h -- the launch planner of the field query (host logic only; included by d3f_api.hip).
//
// Which kernel family gathers a query, in which point order, with which geometry.  d3f_eval_plan_query reports exactly what
// this file decides (no device work happens here), and tests/test_abi.py holds one assertion per row and threshold.
//
//   THE FAMILY TABLE (kFamilies below; walked top down, the first row whose predicate holds takes the query)
//   family           kernel (fuse_<family>.hip)        takes the query when                                                  point order
//   ---------------  --------------------------------  --------------------------------------------------------------------  --------------------------
//   dist-only        fused_eval_dist_kernel<MODE,V,OCC,T> no channel maps (return_names=[], eval_dist)                          caller
//   register-rows    fused_eval_rows_kernel            a PATCH-resolution map of exactly 1024 fp32 channels, others thin,      lattice: brick walk (32 pts)
//                                                      finite maps, no '<k>_inter', >= 65 536 points, <= 8 views: with MORE   cloud: Hilbert / caller order
//                                                      than four views, or where the window row does not take the query
//                                                      (checked before lds-window; id 5 in d3f_eval_plan.family)
//   lds-window       fused_eval_window_kernel          the wide map is PATCH-resolution (texel >= 4 px), whole 128-channel     lattice: brick walk
//                                                      slices, fp32 or fp16, others thin, finite maps, no '<k>_inter',         cloud: Hilbert order, gated
//                                                      >= 65 536 points, <= 8 views; points: a lattice, or a cloud of
//                                                      >= 262 144 points in the Hilbert order (then gated against cell-runs)
//   cell-runs        fused_eval_runs_kernel            a patch-resolution wide fp32 map, finite maps, no '<k>_inter',          caller (lattice columns) /
//                                                      >= 65 536 points (and the cell-run side of a cloud's gate)             Hilbert order (clouds)
//   channel-sliced   fused_eval_sliced_kernel          the wide map is DENSE (beyond the caches), 128..1024 channels in       lattice: brick walk
//                                                      whole 128-channel slices, fp32 or fp16, others thin; lattice walk or    cloud: Hilbert order
//                                                      Hilbert-ordered cloud
//   direct           fused_eval_kernel / _wide / _f16  everything else (and D3F_TUNE_DIRECT_GATHER / REFERENCE_ROUNDING)       caller, walk or Hilbert
//
// Thresholds (one test row each, tests/test_abi.py::test_plan_table): kSmallBatch, kWindowCloudMin, kCacheResidentBytes,
// kBatchedLoadBytes, kBeyondLlcBytes.  The tuning knobs (struct Tune below) are all zero in the product build; an experiments
// build overlays them from D3F_EXP_* environment variables; results never depend on any of this -- every family computes the same numbers (tests: bit-identity).
#pragma once

// ---- thresholds ----------------------------------------------------------------------------------------------------------------
//  kSmallBatch          fewer points than this: no reordering, no window / cell-run / sliced launch -- the set-up of those
//                       paths costs more than it saves, and small batches are spread over >= 1024 workgroups instead
//  kWindowCloudMin      a cloud of at least this many points (in the Hilbert order) may take the LDS-window kernel when the device-side
//                       probe finds its tiles compact (below: the window kernel's ~25-us workgroups do not fill the chip twice
//                       over and the cell-run kernel wins: 71 k surface points 0.17 vs 0.12 ms, 100 k keypoints 0.12 vs 0.09)
//  kCacheResidentBytes  all requested maps together at most this big live in the L2s / Infinity Cache anyway: the caller's
//                       order is kept (unless the cloud has no locality at all), 128-point tiles
//  kBatchedLoadBytes    a map at most this big issues all 4*U corner loads of a view before the first use; bigger maps
//                       in caller order use load-use per vector (a smaller in-flight footprint measured faster)
//  kInfinityCacheBytes  maps up to this size stay in the Infinity Cache: a cloud below kWindowCloudMin points whose caller declares its order
//                       LOCAL (D3F_FLAG_LOCAL_POINTS) keeps that order -- five ordering launches of ~5 us each do not pay around a
//                       90-120 us query (the 71 k surface points of vis_repr.py:97-103 in flat-index order)
//  kBeyondLlcBytes      maps beyond this in CALLER order without scratch: 64-point tiles at 2 workgroups per CU
constexpr int64_t kSmallBatch = 65536;
constexpr int64_t kWindowCloudMin = 262144;
constexpr int64_t kCacheResidentBytes = 64LL << 20;
constexpr int64_t kBatchedLoadBytes = 128LL << 20;
constexpr int64_t kBeyondLlcBytes = 512LL << 20;
constexpr int64_t kInfinityCacheBytes = 256LL << 20;


// ---- tuning knobs ----------------------------------------------------------------------------------------------------------------
// One field per knob of the tuning sessions, all 0 = automatic in the product build (which reads no environment variable and keeps
// no hidden state).  An experiments build (python -m d3fields_amd.build --experiments, -DD3F_EXPERIMENTS) overlays them from the
// environment in load_tune(), the ONE place that does so.  Integers; results never depend on them.
//   runs / runs_u / runs_occ / runs_tile   cell-run gather: -1 off, run length 2 / 4 / 8; vectors per lane 1..3; waves per SIMD of the (1,8) / (2,8) variants; tile points
//   window (+ _u _vc _occ _pool _lpp _pipe _f16 _sparse _slack _want _rr)   LDS-window kernel: -1 never, 32 / 64 / 128 = always with that many points per
//                      workgroup; vectors per lane, views in flight, workgroups per CU, pool texels, lanes per point, -1 = plain view loop, ...
//   sliced (+ _vc _unit _ilv _tile _pad _f16 _cloud)   channel-sliced launch: 1 / 2 / 3 = 128- / 256- / 512-byte slices, -1 never
//   walk / walk_tile   lattice brick walk: -1 off; tile shape as digits x y z (222 default, 224 with a thin map)
//   rows / rows_tile   register-rows kernel (1024-channel patch maps): -1 never, 1 = also below kSmallBatch; brick shape as digits x y z (442)
//   thin               -1: thin maps through the view-sequential gather_map instead of gather_map_thin
//   store              row-store policy: -1 plain, 1 sc1, 3 `sc1 nt`, default `nt` (fuse_common.h: store_out)
//   dist               distance-only pass: -1 = the branch of fused_eval_kernel (rounds 1-5), 8 = fused_eval_dist_kernel held to eight waves per SIMD with three and more views too,
//                      + 16 = the compiler's divisions instead of the short form, + 32 = no tiled copy of the depth maps
//   gate               > 0 always the window side of a cloud's gate, < 0 always the cell runs
//   order_morton / order_fixed_grid / order_bits / scan3   point ordering: the Z curve of rounds 1-4, the fixed 4-mm grid, prefix bits, the three-launch scan
//   stamps             1: s_memtime phase stamps of the window kernel (d3f_exp_read_stamps)
#define D3F_TUNE_KNOBS(X)                                                                                                          \
    X(gate, "D3F_EXP_GATE") X(order_bits, "D3F_EXP_ORDER_BITS") X(order_fixed_grid, "D3F_EXP_ORDER_FIXED_GRID")                   \
    X(order_morton, "D3F_EXP_ORDER_MORTON") X(dist, "D3F_EXP_DIST") X(rows, "D3F_EXP_ROWS") X(rows_tile, "D3F_EXP_ROWS_TILE") X(runs, "D3F_EXP_RUNS") X(runs_occ, "D3F_EXP_RUNS_OCC") X(runs_tile, "D3F_EXP_RUNS_TILE") \
    X(runs_u, "D3F_EXP_RUNS_U") X(scan3, "D3F_EXP_SCAN3") X(sliced, "D3F_EXP_SLICED") X(sliced_cloud, "D3F_EXP_SLICED_CLOUD")     \
    X(sliced_f16, "D3F_EXP_SLICED_F16") X(sliced_ilv, "D3F_EXP_SLICED_ILV") X(sliced_pad, "D3F_EXP_SLICED_PAD")                   \
    X(sliced_tile, "D3F_EXP_SLICED_TILE") X(sliced_unit, "D3F_EXP_SLICED_UNIT") X(sliced_vc, "D3F_EXP_SLICED_VC")                 \
    X(stamps, "D3F_EXP_STAMPS") X(store, "D3F_EXP_STORE") X(thin, "D3F_EXP_THIN") X(walk, "D3F_EXP_WALK")                         \
    X(walk_tile, "D3F_EXP_WALK_TILE") X(window, "D3F_EXP_WINDOW") X(window_f16, "D3F_EXP_WINDOW_F16")                             \
    X(window_lpp, "D3F_EXP_WINDOW_LPP") X(window_occ, "D3F_EXP_WINDOW_OCC") X(window_pipe, "D3F_EXP_WINDOW_PIPE")                 \
    X(window_pool, "D3F_EXP_WINDOW_POOL") X(window_rr, "D3F_EXP_WINDOW_RR") X(window_slack, "D3F_EXP_WINDOW_SLACK")               \
    X(window_sparse, "D3F_EXP_WINDOW_SPARSE") X(window_u, "D3F_EXP_WINDOW_U") X(window_vc, "D3F_EXP_WINDOW_VC")                   \
    X(window_want, "D3F_EXP_WINDOW_WANT")
struct Tune {
#define D3F_TUNE_FIELD(f, env) int f = 0;
    D3F_TUNE_KNOBS(D3F_TUNE_FIELD)
#undef D3F_TUNE_FIELD
};
inline Tune load_tune()
{
    Tune t;
#ifdef D3F_EXPERIMENTS
#define D3F_TUNE_ENV(f, env) if (const char *v = getenv(env)) t.f = atoi(v);
    D3F_TUNE_KNOBS(D3F_TUNE_ENV)
#undef D3F_TUNE_ENV
#endif
    return t;
}

// Phase-B lane mapping of one map: vector width, lanes per point (2^k) and vectors per lane.
// Minimises idle lane-slots (passes*lpp*U - cvec), then passes, then prefers wide groups
// (longer contiguous segments per load instruction).
// batch: issue all 4*U corner loads before the first use (best for cache-resident maps, U <= 3);
// otherwise load-use per vector, U <= 4 (best when the map misses the caches).
inline void pick_mapping(d3f::MapDesc &m, bool can16, bool can8, bool batch, int max_u = 4)
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

// ---- what the planner may look at: host facts of one query ------------------------------------------------------------------------
struct Query {
    const d3f_views *views;
    int64_t n;
    int32_t n_maps;
    uint32_t flags;                 // after D3F_FLAG_REFERENCE_ROUNDING has been rewritten to D3F_TUNE_DIRECT_GATHER
    int mode;                       // 0 Fusion.eval, 1 Fusion.eval_dist
    const int32_t *lattice;         // the points are a z-fastest lattice of these dims (d3f_eval_lattice / d3f_eval_grid), or nullptr
    bool grid;                      // d3f_eval_grid: coordinates come from the axis arrays (no pts, no scratch)
    bool plan_only;
    bool may_reorder;               // scratch for the Hilbert order is there (and the points are a cloud the library may reorder)
    bool finite_expected;           // the host vouches for the maps, or every tensor carries a device-side check word
    bool direct;                    // D3F_TUNE_DIRECT_GATHER
    int tl;                         // D3F_TUNE_TILE_LOG2 (0 = automatic)
    int64_t map_bytes;              // all requested maps together
    bool want_inter[D3F_MAX_MAPS];  // '<k>_inter' requested for P.maps[k]
    Tune tune;                      // all zero in the product build
    int cloud_side;                 // eval_common: 0 plan queries / ungated callers, 1 first pass (may gate), 2 the gated cell-run pass

    // would the points be walked in the Hilbert order?  (clouds; performance only)
    bool reorder_cloud() const
    {
        if (!may_reorder || lattice) return false;
        if (flags & D3F_TUNE_FORCE_REORDER) return true;
        if (n < kSmallBatch) return false;
        if (flags & D3F_FLAG_UNORDERED_POINTS) return true;
        if (map_bytes <= kCacheResidentBytes) return false;
        // maps beyond the L2s: the Hilbert walk pays -- unless the caller's own order is local, the cloud is too small for the window
        // kernel and the maps sit in the Infinity Cache
        if ((flags & D3F_FLAG_LOCAL_POINTS) && n < kWindowCloudMin && map_bytes <= kInfinityCacheBytes) return false;
        return true;
    }
    // (a flat lattice with more than 2^28 tiles per 16-tile slab would overflow the walk's 32-bit level arithmetic)
    bool walk_possible() const
    {
        return lattice && 16.0 * ((lattice[1] + 1) / 2) * ((lattice[2] + 1) / 2) < 4294967296.0 && n_maps > 0 && n >= kSmallBatch &&
               n <= 0x7fffffffLL && !(flags & D3F_TUNE_NO_REORDER) && tune.walk >= 0;
    }
};

// What the planner decided besides the fields of EvalParams.
enum FamilyId { kFamDistOnly = 0, kFamWindow, kFamRuns, kFamSliced, kFamDirect, kFamRows };
struct Plan {
    FamilyId family = kFamDirect;
    bool walk = false;              // closed-form brick walk of a lattice
    bool reorder = false;           // walk, or the Hilbert order of a cloud
    bool xcd_remap = false;         // XCD k takes the k-th contiguous eighth of the tiles
    bool window = false, runs = false, sliced = false, rows = false;
};

// ---- predicates on one map -----------------------------------------------------------------------------------------------------------
// Cell-run gather (fuse_body.h gather_map_runs): fp32 maps read as 16-byte vectors with >= 32 vectors per texel whose
// texels span >= 4 image pixels -- the patch-resolution feature maps of the reference (fusion.py:694-697).
inline bool runs_candidate(const d3f::MapDesc &m, int H, int W)
{
    return m.esize == 4 && m.vw == 4 && m.C >= 128 && (W - 1) >= 4 * (m.fw - 1) && (H - 1) >= 4 * (m.fh - 1);
}

// LDS texel windows (fuse_window.hip fused_eval_window_kernel): a patch-resolution wide map in whole 128-channel slices whose texels
// start on 16-byte boundaries -- fp32 (512-byte slices), or stored in fp16 (256-byte slices, round 5: lattices only)
inline bool window_candidate(const d3f::MapDesc &m, const d3f_views *views, bool check_pointer)
{
    const int es = m.esize, per16 = 16 / es;            // channels per 16 bytes
    const bool vec = es == 4 ? m.vw == 4 : m.vw == 8;
    // (m.fold: a map of <= 256 bytes per texel -- 128 channels of fp16 -- belongs to the thin family, which keeps the reference's
    //  operation order in every kernel; the window kernel's arithmetic is the folded one)
    return vec && m.fold && m.C >= 128 && m.C % 128 == 0 && (views->W - 1) >= 4 * (m.fw - 1) && (views->H - 1) >= 4 * (m.fh - 1) &&
           (int64_t)views->V * m.sv * es < (1LL << 31) && (m.sx % per16) == 0 && (m.sy % per16) == 0 && (m.sv % per16) == 0 &&
           (!check_pointer || reinterpret_cast<uintptr_t>(m.data) % 16 == 0);
}

inline bool thin_fp32(const d3f::MapDesc &m) { return m.esize == 4 && m.C * 4 <= 256; }

// (vectors per lane U, run length K) of the cell-run gather: the built variants are (1,8) (2,4) (2,8) (3,2) (3,4)
inline void pick_runs_mapping(d3f::MapDesc &m, int U, int K)
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
inline void pick_window_brick(int nx, int ny, int nz, int T, int &bx, int &by, int &bz)
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

inline int tile_points_for(int V)
{
    // LDS per workgroup = tile*V*24 B (+ small); keep it <= 32 KiB so >= 4 workgroups fit a CU.
    // 128 points measured best (985 600-pt grid, C=384: 128 -> 0.99 ms, 256 -> 1.06 ms patch-res).
    int t = 128;
    while (t > 32 && (long)t * V * 24 > 32 * 1024) t >>= 1;
    return t;
}

inline int window_tile_points(const Tune &tune)
{
    const int k = tune.window;          // 0 automatic, -1 off, 32 / 64 / 128: points per workgroup (experiments)
    return (k == 32 || k == 64 || k == 128) ? k : 64;
}

// =================================================== family: lds-window ===========================================================
// Row predicate + pool sizing (the pool decides feasibility, so both live here).  On success the win_* fields of P are set.
// A cloud reaches this row only in the first pass of d3f_eval (cloud_side 1): the launch is then GATED against the cell runs.
inline bool window_row(const Query &q, d3f::EvalParams &P)
{
    const d3f_views *views = q.views;
    const int win_knob = q.tune.window;
    const bool cloud_candidate = q.cloud_side == 1 && q.reorder_cloud() && q.n >= kWindowCloudMin && !(q.flags & D3F_TUNE_NO_WINDOW_GATE);
    // default: lattices (a brick's windows are compact), and clouds through the device-side gate (fuse_common.h: gated_out);
    // not when a cell-run variant is asked for explicitly
    const bool automatic = win_knob == 0 && (q.lattice != nullptr || cloud_candidate) && q.tune.runs == 0 && q.tune.runs_u == 0;
    const bool half0 = q.n_maps >= 1 && P.maps[0].esize == 2;      // fp16-stored: bricks of a lattice only (the cell-run side of a cloud's gate is fp32)
    bool window = (win_knob > 0 || automatic) && !q.direct && q.mode == 0 && q.n_maps >= 1 && q.finite_expected && q.n >= kSmallBatch &&
                  q.n <= 0x7fffffffLL && q.tl == 0 && views->V <= 8 && window_candidate(P.maps[0], views, !q.plan_only) &&
                  (!half0 || (q.lattice != nullptr && q.tune.window_f16 >= 0));
    for (int s = 0; s < q.n_maps; ++s) window = window && !q.want_inter[s];
    for (int s = 1; s < q.n_maps; ++s) window = window && thin_fp32(P.maps[s]);
    if (!window) return false;
    const int T = window_tile_points(q.tune);
    const int VP = views->V <= 1 ? 1 : views->V <= 2 ? 2 : views->V <= 4 ? 4 : 8;
    int U = q.tune.window_u;
    const int cv = P.maps[0].C / 128;                  // 128-channel granules per texel (512 bytes of fp32, 256 of fp16)
    const int slot = P.maps[0].esize == 2 ? 256 : 512;
    if (U < 1 || U > 4 || cv % U != 0 || slot == 256) U = 1;
    if (slot == 256) P.win_lpp = 16;
    // per (point, view): 32-byte window record (+ the 16-byte view record when thin maps ride along); per point 20 bytes
    const int base = T * (views->V * 32 + 16) + (q.n_maps > 1 ? T * views->V * 16 : 0) + T * 20 + views->V * 48;     // records at a padded point stride
    const int pool_offset = (base + 511) / 512 * 512;
    int occ = q.tune.window_occ;
    const bool occ_forced = occ >= 5 && occ <= 6;        // experiments: 5 / 6 workgroups per CU with the plain point loop
    if (occ < 2 || occ > 6) occ = 4;
    if (U > 1) occ = 2;                                 // those variants are built for 2 workgroups per CU
    // touched-texel pool (SPARSE) for clouds, whole rectangles for lattice bricks (which never overflow: 0.42 vs 0.455 ms on
    // C2-patch); experiments builds: D3F_EXP_WINDOW_SPARSE = 1 / -1 forces either
    P.win_sparse = q.tune.window_sparse > 0 ? 1 : (q.tune.window_sparse < 0 ? 0 : (q.lattice ? 0 : 1));
    if (slot == 256) P.win_sparse = 0;
    // static LDS of the kernel + allocation granularity: 3 workgroups per CU stop fitting with less (measured, round 5)
    const int slack = q.tune.window_slack > 0 ? q.tune.window_slack : (P.win_sparse ? 4096 : 2048);
    // slots per view worth a workgroup per CU: a brick's rectangles ~17; a cloud tile's touched texels ~12 (p90 14)
    const int want = q.tune.window_want > 0 ? q.tune.window_want : (P.win_sparse ? 14 : 17);
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
    if (q.tune.window_pool > 0 && q.tune.window_pool < texels) texels = q.tune.window_pool;
    if (texels > 320) texels = 320;                      // kWinMaxTexels (fuse_window.hip)
    texels &= ~1;
    window = texels >= 2 && (T * VP) % 64 == 0 && q.n / T < 0x7fffffffLL;
    if (U > 1) P.win_lpp = 32;
    P.win_u = U; P.win_occ = occ; P.win_pool_offset = pool_offset; P.win_pool_texels = texels;
    P.win_vc = q.tune.window_vc == 2 ? 2 : 1;
    P.win_slices = window ? cv / U : 0;
    return window;
}

// geometry of the window launch (after the point order is decided)
inline void window_geometry(const Query &q, d3f::EvalParams &P, Plan &pl)
{
    P.tile_pts = window_tile_points(q.tune); P.lds_pad = 0;
    if (pl.walk) pick_window_brick(P.walk_nx, P.walk_ny, P.walk_nz, P.tile_pts, P.walk_tx, P.walk_ty, P.walk_tz);
    for (int s = 1; s < q.n_maps; ++s) pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
    pl.xcd_remap = false;
    P.flags &= ~D3F_TUNE_XCD_REMAP;
    // a cloud's tiles go round-robin over the XCDs (all eight work on one neighbourhood: C2-patch cloud 0.52 ms against 0.58
    // with contiguous eighths, which is the lattice bricks' mapping); experiments builds: D3F_EXP_WINDOW_RR=-1 = eighths
    if (!pl.walk && q.tune.window_rr >= 0) P.flags |= D3F_TUNE_XCD_REMAP;
}

// =================================================== family: register-rows =========================================================
// fuse_rows.hip: a patch-resolution map of exactly 1024 fp32 channels (256 lanes x one 16-byte vector: the reference's ViT-L
// features, config 4) whose texels start on 16-byte boundaries; finite maps, no '<k>_inter', the other maps thin; a lattice (bricks of
// 32 points) or any cloud -- Hilbert order when the planner reorders, the caller's otherwise.  Not gated: it replaces both sides.
inline bool rows_row(const Query &q, d3f::EvalParams &P)
{
    const int knob = q.tune.rows;
    bool rows = knob >= 0 && q.tune.window == 0 && q.tune.runs == 0 && q.tune.runs_u == 0 && !q.direct && q.mode == 0 && q.n_maps >= 1 &&
                q.finite_expected && (q.n >= kSmallBatch || knob > 0) && q.n <= 0x7fffffffLL && q.tl == 0 && q.views->V <= 8 &&
                P.maps[0].esize == 4 && P.maps[0].C == 1024 && window_candidate(P.maps[0], q.views, !q.plan_only);
    for (int s = 0; s < q.n_maps; ++s) rows = rows && !q.want_inter[s];
    for (int s = 1; s < q.n_maps; ++s) rows = rows && thin_fp32(P.maps[s]);
    return rows;
}

inline void rows_geometry(const Query &q, d3f::EvalParams &P, Plan &pl)
{
    P.rows = 1; P.tile_pts = D3F_ROWS_PTS; P.lds_pad = 0;
    if (pl.walk) pick_window_brick(P.walk_nx, P.walk_ny, P.walk_nz, P.tile_pts, P.walk_tx, P.walk_ty, P.walk_tz);
    const int shape = q.tune.rows_tile;                  // experiments: digits x y z (powers of two, product 32)
    if (pl.walk && shape >= 111 && (shape / 100) * (shape / 10 % 10) * (shape % 10) == D3F_ROWS_PTS) { P.walk_tx = shape / 100; P.walk_ty = shape / 10 % 10; P.walk_tz = shape % 10; }
    for (int s = 1; s < q.n_maps; ++s) pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
    pl.xcd_remap = false;
    P.flags &= ~D3F_TUNE_XCD_REMAP;
    // a big cloud's tiles go round-robin over the XCDs (config 4's cloud: 2.26 ms and 13.9 GB of L2 fills against 2.34 ms and 6.7 GB with
    // contiguous eighths -- all eight L2s on one neighbourhood win on time, as for the window kernel); a small one keeps the eighths
    // (the 71 k surface points: 0.125 vs 0.129 ms)
    if (!pl.walk && q.n >= kWindowCloudMin && q.tune.window_rr >= 0) P.flags |= D3F_TUNE_XCD_REMAP;
}

// =================================================== family: cell-runs ============================================================
// Cell-run gather for patch-resolution wide maps (one per launch: phase A keeps one "same cell as the previous point" flag per
// (point, view)): consecutive points of the processing order (a grid column in caller order, the Hilbert walk of a cloud) mostly
// stay inside one texel cell of a view, so a lane group keeps the four corner vectors in registers across a run of points.
// Needs the exact invalid-view skip (finite maps), no '<k>_inter' output and fp32 maps only.  On success the map's run geometry
// is set and the other maps are re-mapped to one batched vector per lane (the kernel's register budget).
inline bool runs_row(const Query &q, d3f::EvalParams &P, bool window_taken)
{
    const int knob = q.tune.runs;
    bool blocked = window_taken || q.direct || knob < 0 || !q.finite_expected || q.n < kSmallBatch || q.tl != 0;
    for (int s = 0; s < q.n_maps; ++s)
        blocked |= P.maps[s].esize == 2 || q.want_inter[s] ||
                   (P.maps[s].unroll == -4 && !runs_candidate(P.maps[s], q.views->H, q.views->W));
    bool any_runs = false;
    for (int s = 0; s < q.n_maps && !blocked && !any_runs; ++s)
        if (runs_candidate(P.maps[s], q.views->H, q.views->W)) {
            pick_runs_mapping(P.maps[s], q.tune.runs_u, knob);
            any_runs = true;
        }
    if (any_runs)
        for (int s = 0; s < q.n_maps; ++s)
            if (P.maps[s].runs == 0 && P.maps[s].unroll != 1)
                pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
    return any_runs;
}

inline void runs_geometry(const Query &q, d3f::EvalParams &P)
{
    int k = 8, lg = 6;
    for (int s = 0; s < q.n_maps; ++s)
        if (P.maps[s].runs > 0) { k = P.maps[s].runs; lg = P.maps[s].lpp_log2; }
    const int round = (d3f::kBlock >> lg) * k;    // one run per lane group
    P.tile_pts = round < 64 ? 64 : round;         // >= 64 points per workgroup (a lane group then takes several runs)
    // batches of less than ~2 workgroups per slot (256 CUs x 7): halve the tile so that the tail is shorter
    // (100 k keypoints: 0.126 -> 0.119 ms; the 985 600-point grid is slower with 32-point tiles: 0.632 -> 0.655)
    if (q.n / P.tile_pts < 4096 && P.tile_pts / 2 >= round) P.tile_pts /= 2;
    if (q.tune.runs_tile >= round) P.tile_pts = q.tune.runs_tile;
    while ((long)P.tile_pts * q.views->V * 88 > 40 * 1024 && P.tile_pts > round) P.tile_pts >>= 1;   // records + 2 corner slots
    P.lds_pad = 0;
}

// =================================================== family: direct (and the common order / tile geometry) =======================
// Geometry shared by every gather that takes whole texels per lane group (measured on MI355X):
//  * Hilbert / lattice walk: 8-point tiles (one point per lane group; 16 when a thin map such as the mask is also
//    requested), XCD k takes the k-th contiguous eighth of the walk, so the ~1 k points in flight on an
//    XCD form one compact blob whose texels stay in that XCD's 4 MiB L2
//    (C2 dense 2.84 -> 2.12 ms, C4 patch 13.0 -> 4.8 ms; 32-point tiles: 2.48 / 5.3 ms);
//  * caller order: 128-point tiles, round-robin XCDs (0.80 ms on C2 patch); for maps far beyond the
//    256 MiB Infinity Cache 64-point tiles at 2 workgroups per CU (3.2 -> 2.96 ms on C2 dense);
//  * small batches (keypoints, tracking): >= 1024 workgroups.
inline void direct_geometry(const Query &q, d3f::EvalParams &P, Plan &pl)
{
    const d3f_views *views = q.views;
    const int max_tile = tile_points_for(views->V) * 2;
    if (pl.reorder) {
        // on the walk the in-flight footprint is tiny, so batched corner loads win again wherever one pass
        // of <= 3 vectors per lane covers the channels (C2 dense 2.07 -> 1.99 ms); C = 1024 keeps load-use x 4
        for (int s = 0; s < q.n_maps; ++s) {
            d3f::MapDesc &m = P.maps[s];
            const bool forced = (q.flags & ((1u << 26) | (1u << 27))) != 0;
            if (!forced && m.unroll < 0 && (m.C / m.vw) <= 3 * 64) {
                const bool a16 = m.vw == 4, a8 = m.vw >= 2;
                pick_mapping(m, a16, a8, true, pl.runs ? 1 : 4);
            }
        }
        // one point per lane group: 8 points when every map takes 32 lanes per point, else 16
        bool thin = false;
        for (int s = 0; s < q.n_maps; ++s) thin |= P.maps[s].lpp_log2 < 5;      // < 32 lanes per point: 16 groups have work
        P.tile_pts = thin ? 16 : 8; P.lds_pad = 0; pl.xcd_remap = true;
        if (pl.walk) {                                   // the tile is a brick of the lattice
            P.walk_tx = 2; P.walk_ty = 2; P.walk_tz = thin ? 4 : 2;
            const int shape = q.tune.walk_tile;      // experiment: digits x y z, e.g. 224, 144, 422
            if (shape >= 111 && shape <= 888 && (shape / 100) * (shape / 10 % 10) * (shape % 10) == P.tile_pts && shape / 10 % 10 > 0 && shape % 10 > 0) {
                P.walk_tx = shape / 100; P.walk_ty = shape / 10 % 10; P.walk_tz = shape % 10;
            }
        }
        // maps that fit the L2s / Infinity Cache anyway (patch-resolution features, the mask): the walk is only there
        // to give a random cloud L1 locality and the big tiles of the caller-order path stay best
        // (C2 patch, random cloud: caller order 1.93 ms, walk with 8-point tiles 1.07, with 128-point tiles 0.76)
        if (q.map_bytes <= kCacheResidentBytes && !pl.walk) P.tile_pts = tile_points_for(views->V);
    } else if (q.map_bytes > kBeyondLlcBytes && P.tile_pts > 64 && q.n >= kSmallBatch && !pl.runs) {
        P.tile_pts = 64; P.lds_pad = 64 * 1024;
    }
    // small batches (keypoints, tracking): a 128-point tile is 16-32 serial rounds per lane group, so a few hundred
    // points would run on 3 CUs for ~200 us; spread them over >= 1024 workgroups instead (N = 300: 170 -> ~25 us)
    if (!pl.reorder && q.n_maps > 0)
        while (P.tile_pts > 8 && q.n / P.tile_pts < 1024) P.tile_pts >>= 1;
    // tuning bits (D3F_TUNE_*): experiments only, results never depend on them
    if (q.tl >= 2 && q.tl <= 8) { P.tile_pts = (1 << q.tl) <= max_tile ? (1 << q.tl) : max_tile; P.lds_pad = 0; }
    if ((q.flags >> 16) & 0xFF) P.lds_pad = ((int)((q.flags >> 16) & 0xFF) == 0xFF) ? 0 : (int)((q.flags >> 16) & 0xFF) * 1024;
    if (q.flags & D3F_TUNE_XCD_REMAP) pl.xcd_remap = !pl.xcd_remap;
    P.flags = (q.flags & ~D3F_TUNE_XCD_REMAP) | (pl.xcd_remap ? D3F_TUNE_XCD_REMAP : 0u);
}

// =================================================== family: channel-sliced =======================================================
// Channel-sliced launch (fuse_sliced.hip): a lattice walk (or the Hilbert order of a cloud) on a dense wide map that is the FIRST map
// of the launch; any other map must be thin (it rides along with slice 0).  512-byte slices of fp32 (256-byte ones of fp16: 16 lanes
// x 8 channels), two views in flight; a wide map WITH thin companions takes 32 points per workgroup (C3-dense, features + 8-channel
// mask: 3.04 -> 2.80 ms -- the whole-texel kernel stalls on the thin map's gather, 16 points x 2 lanes per workgroup), a wide map
// ALONE 16 points per workgroup (C2-dense 1.62 -> 1.52 ms: with 32 the slicing removed 20-29 % of the L2 fills and no time, with 64
// it lost 20 %: the points in flight per XCD are what the L2 window is made of).  D3F_EXP_SLICED = 1 / 2 / 3 forces
// 128- / 256- / 512-byte slices, -1 disables; _TILE 8 / 16 / 32 / 64 points per workgroup.
// Runs AFTER direct_geometry (it replaces that geometry, and restores it when the launch turns out infeasible).
inline bool sliced_row(const Query &q, d3f::EvalParams &P, const Plan &pl)
{
    const d3f_views *views = q.views;
    int sl = q.tune.sliced;
    bool thin_rest = true;
    for (int s = 1; s < q.n_maps; ++s) thin_rest = thin_rest && P.maps[s].C * P.maps[s].esize <= 256 && P.maps[s].esize == 4;
    const bool half_sl = q.n_maps >= 1 && P.maps[0].esize == 2;      // fp16-stored wide map: 16 lanes x 8 channels = 128-channel slices
    const bool automatic = sl == 0 && thin_rest && q.n_maps >= 1 && P.maps[0].C % 128 == 0 && P.maps[0].C <= 1024 &&
                           (!half_sl || q.tune.sliced_f16 >= 0);
    if (automatic) sl = half_sl ? 2 : 3;
    if (half_sl && sl != 2) sl = 0;
    // ... or the Hilbert order of a cloud on maps beyond the caches (tiles of 16 / 32 consecutive points of the order)
    const bool cloud = pl.reorder && !pl.walk && !pl.runs && q.map_bytes > kCacheResidentBytes && q.tune.sliced_cloud >= 0;
    bool ok = (pl.walk || cloud) && !pl.window && !q.direct && (sl >= 1 && sl <= 3) && q.mode == 0 && q.n_maps >= 1 &&
              ((P.maps[0].esize == 4 && P.maps[0].vw == 4) || (half_sl && P.maps[0].vw == 8 && P.maps[0].fold)) && !q.want_inter[0] && q.tl == 0;
    const int lg = sl + 2, lanes = 1 << lg;      // 1: 8 lanes (128-byte slices), 2: 16 lanes, 3: 32 lanes (512 bytes)
    P.sl_vc = q.tune.sliced_vc > 0 ? q.tune.sliced_vc : (automatic ? 2 : 4);
    if (half_sl) P.sl_vc = 2;
    const int cpl = half_sl ? 8 : 4;              // channels per lane (one 16-byte vector)
    ok = ok && P.maps[0].C % (cpl * lanes) == 0 && P.maps[0].C >= 128;
    for (int s = 1; s < q.n_maps && ok; ++s) ok = P.maps[s].C * P.maps[s].esize <= 256 && !q.want_inter[s] && P.maps[s].esize == 4;
    if (!ok) return false;
    const int keep_tile = P.tile_pts, keep_pad = P.lds_pad, keep_t[3] = {P.walk_tx, P.walk_ty, P.walk_tz};
    d3f::MapDesc keep_maps[D3F_MAX_MAPS];
    for (int s = 0; s < q.n_maps; ++s) keep_maps[s] = P.maps[s];
    const int tile_knob = q.tune.sliced_tile;
    const bool big = tile_knob == 64;                            // experiment: 64 points per workgroup (four 2x2x4 tiles)
    const bool tiny = tile_knob == 16 || (tile_knob == 0 && q.n_maps == 1);   // 16 points per workgroup (four 2x2x1 tiles)
    const bool mini = tile_knob == 8;                            // experiment: 8 points per workgroup (four 2x1x1 tiles)
    P.walk_tx = 2; P.walk_ty = mini ? 1 : 2; P.walk_tz = big ? 4 : ((tiny || mini) ? 1 : 2);
    P.sl_lg = lg;
    P.sl_slices = P.maps[0].C / (cpl * lanes);
    P.tile_pts = big ? 64 : (tiny ? 16 : (mini ? 8 : 32)); P.lds_pad = q.tune.sliced_pad > 0 ? q.tune.sliced_pad * 1024 : 0;
    if (pl.walk) {
        P.sl_tiles = (int64_t)((P.walk_nx + 1) / 2) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) * ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
        P.sl_groups = (P.sl_tiles + 3) / 4;
    } else {
        P.sl_tiles = 0;
        P.sl_groups = (q.n + P.tile_pts - 1) / P.tile_pts;
    }
    P.sl_unit = q.tune.sliced_unit > 0 ? q.tune.sliced_unit : (big ? 64 : (tiny ? 256 : (mini ? 512 : 128)));   // 4096 points per unit (smaller: slower)
    P.sl_chunks = (P.sl_groups + P.sl_unit - 1) / P.sl_unit;
    P.sl_ilv = q.tune.sliced_ilv >= 2 && q.tune.sliced_ilv <= 4 ? q.tune.sliced_ilv : 1;
    for (int s = 1; s < q.n_maps; ++s) pick_mapping(P.maps[s], P.maps[s].vw == 4, P.maps[s].vw >= 2, true, 1);
    if ((((P.sl_chunks * P.sl_slices + 7) / 8) + P.sl_ilv) * 8 * P.sl_unit > 0x7fffffffLL) P.sl_slices = 0;
    // its dynamic LDS (records + one corner record per (point, view)) must fit the 64 KiB a launch gets without opting in:
    // 32-point tiles with ~36 and more views do not -- such a query keeps the whole-texel kernel
    if ((int64_t)d3f::fused_lds_base(P.tile_pts, views->V) + (int64_t)P.tile_pts * views->V * 32 + P.lds_pad > 64 * 1024) P.sl_slices = 0;
    if (P.sl_slices == 0) {             // not this launch after all: the geometry of the whole-texel kernel again
        P.tile_pts = keep_tile; P.lds_pad = keep_pad; P.walk_tx = keep_t[0]; P.walk_ty = keep_t[1]; P.walk_tz = keep_t[2];
        for (int s = 0; s < q.n_maps; ++s) P.maps[s] = keep_maps[s];
        return false;
    }
    return true;
}

// =================================================== the table ====================================================================
struct FamilyRow {
    FamilyId id;
    const char *name, *kernel, *takes;
};
constexpr FamilyRow kFamilies[] = {
    {kFamDistOnly, "dist-only", "fused_eval_dist_kernel<MODE, V, OCC, TILED>", "no channel maps (return_names=[], eval_dist)"},
    {kFamWindow, "lds-window", "fused_eval_window_kernel", "a patch-resolution wide map in whole 128-channel slices on a lattice, or (gated on the device) on a cloud of >= 262 144 points in the Hilbert order"},
    {kFamRuns, "cell-runs", "fused_eval_runs_kernel", "a patch-resolution wide fp32 map, >= 65 536 points; the other side of a cloud's gate"},
    {kFamSliced, "channel-sliced", "fused_eval_sliced_kernel", "a dense wide map (128..1024 channels) on a lattice walk or a Hilbert-ordered cloud"},
    {kFamDirect, "direct", "fused_eval_kernel / _wide_kernel / _f16_kernel", "everything else"},
    {kFamRows, "register-rows", "fused_eval_rows_kernel", "a patch-resolution map of 1024 fp32 channels, finite maps, >= 65 536 points: a lattice (bricks of 32 points) or a cloud"},
};
inline const FamilyRow &family_row(FamilyId id) { return kFamilies[(int)id]; }

// ---- part 1 (before the point order exists): which family, which order --------------------------------------------------------------
// Walks the table: dist-only, lds-window, cell-runs are decided here (they also decide the point order); channel-sliced needs
// the geometry of the order first and is decided in part 2, with `direct` as what is left.
inline void plan_family_and_order(const Query &q, d3f::EvalParams &P, Plan &pl)
{
    if (q.n_maps == 0) {
        pl.family = kFamDistOnly;
    } else if (rows_row(q, P) && (q.views->V > 4 || !(pl.window = window_row(q, P)))) {
        // 1024-channel patch maps: the register rows with more than four views (config 4: 1.47 vs 1.68 ms on the lattice, 2.18 vs
        // 2.66 on the cloud) and wherever the windows do not apply (small clouds: the 71 k surface points 0.125 vs 0.137 ms); four
        // views on a lattice or a big cloud keep the windows (the reference's shape: 2.07 vs 2.37 ms)
        pl.rows = true; pl.window = false; P.win_slices = 0;
        pl.family = kFamRows;
    } else if (pl.window || (pl.window = window_row(q, P))) {
        pl.family = kFamWindow;
    } else if ((pl.runs = runs_row(q, P, false))) {
        pl.family = kFamRuns;
    }
    // Points on a regular lattice (a d3f_grid, or d3f_eval_lattice's dims): the brick walk is closed form -- no keys, no
    // sort, no index array, no scratch -- and replaces the Hilbert sort wherever that would be used.  (With the cell-run
    // gather the caller's z-fastest order is the one wanted: a grid column is one long run.)
    pl.walk = q.walk_possible() && !pl.runs && ((q.flags & D3F_TUNE_FORCE_REORDER) || q.map_bytes > kCacheResidentBytes || pl.window || pl.rows);
    pl.reorder = pl.walk || q.reorder_cloud();
    if (pl.walk) { P.walk_nx = q.lattice[0]; P.walk_ny = q.lattice[1]; P.walk_nz = q.lattice[2]; }
}

// ---- part 2: the geometry of the family (and the channel-sliced row) ----------------------------------------------------------------
inline void plan_geometry(const Query &q, d3f::EvalParams &P, Plan &pl)
{
    direct_geometry(q, P, pl);
    if (pl.runs) runs_geometry(q, P);
    if (pl.family == kFamDirect && !pl.rows && (pl.sliced = sliced_row(q, P, pl))) pl.family = kFamSliced;
    if (pl.window) window_geometry(q, P, pl);
    if (pl.rows) rows_geometry(q, P, pl);
    // walks: all eight XCDs stay inside one macro-brick of ~32 k points at a time (its texel footprint stays in
    // the 256 MiB Infinity Cache), each taking a contiguous eighth of it (C2 dense 1.97 -> 1.74 ms, C4 patch 4.75 -> 4.17)
    P.xcd_chunk = (pl.reorder && pl.xcd_remap) ? (int)((32768 / P.tile_pts + 7) / 8 * 8) : 0;
    if ((q.flags >> 29) & 0x7) P.xcd_chunk = 1024 << (((q.flags >> 29) & 0x7) - 1);   // tuning: 1024 .. 65536 tiles
    if (pl.family == kFamDistOnly) {
        // distance-only pass (return_names=[], eval_dist): one lane per point and nothing per point in LDS.  Rounds 1-3 ran it on
        // the 128-point tiles of the gathers -- half of every 256-lane workgroup idle (SQ_WAVES = 3.85 M for 123.2 M points):
        // four points per lane and workgroup instead, KRt computed once per 1024 points
        // (from 2^24 points on sixteen points per lane: a wave's KRt set-up -- ~100 instructions -- is then 1.5 % of its work instead of 6 %)
        P.tile_pts = q.n >= (1LL << 24) ? 4096 : (q.n >= (1LL << 22) ? 1024 : 256);
        P.lds_pad = 0;
    }
    P.crec_offset = d3f::fused_lds_base(q.n_maps == 0 ? 0 : P.tile_pts, q.views->V);
    // wide maps (>= 16 lanes per point): corner set-up once per (point, view) in phase A, 32 B of LDS each; the
    // cell-run maps come first -- they read their corners from these records
    P.n_pre = 0;
    for (int s = 0; s < q.n_maps; ++s)
        if (P.maps[s].runs > 0) P.maps[s].pre_slot = P.n_pre++;
    if (P.sl_slices > 0) {
        P.maps[0].pre_slot = 0; P.n_pre = 1;             // the sliced kernel keeps the wide map's corner records itself
    } else if (!(q.flags & (1u << 28)))
        for (int s = 0; s < q.n_maps && P.n_pre < 2; ++s) {
            // 32 B per (point, view) and map: only while records + set-ups stay within 48 KiB (>= 3 workgroups per CU)
            const long lds_after = (long)P.crec_offset + (long)(P.n_pre + 1) * P.tile_pts * q.views->V * 32;
            if (P.maps[s].pre_slot < 0 && P.maps[s].lpp_log2 >= 4 && !q.want_inter[s] && lds_after <= 48 * 1024)
                P.maps[s].pre_slot = P.n_pre++;
        }
}

inline int64_t plan_workgroups(const d3f::EvalParams &P, const Plan &pl, int64_t n)
{
    if (pl.walk)
        return (int64_t)((P.walk_nx + P.walk_tx - 1) / P.walk_tx) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) * ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
    return (n + P.tile_pts - 1) / P.tile_pts;
}

// what d3f_eval_plan_query reports (the maps in the caller's order)
inline void report_plan(const d3f::EvalParams &P, const Plan &pl, const int *caller_map, int n_maps, int64_t ntiles, d3f_eval_plan *out)
{
    out->family = (int32_t)pl.family; out->reserved3 = 0;
    out->tile_points = P.tile_pts;
    out->reorder = pl.walk ? 2 : (pl.reorder ? 1 : 0);
    out->lds_bytes = P.crec_offset + P.n_pre * P.tile_pts * P.V * 32 + P.lds_pad;
    out->workgroups = P.sl_slices > 0 ? (((P.sl_chunks * P.sl_slices + 7) / 8 + P.sl_ilv - 1) / P.sl_ilv * P.sl_ilv) * 8 * P.sl_unit : ntiles;
    if (P.win_slices > 0) {
        out->lds_bytes = P.win_pool_offset + (2 + P.win_pool_texels) * (P.maps[0].esize == 2 ? 256 : 512) * P.win_u;
        out->workgroups = ntiles;
    }
    if (P.rows > 0) out->lds_bytes = 24 * 1024;          // static: ops, cells, view records, keys (fuse_rows.hip)
    // 2UVW: the window kernel's template arguments (W: workgroups per CU the pool is sized for); 1LV: sliced launch, L = log2(lanes
    // per point), V = views in flight; cell runs: waves per SIMD the chosen variant is built for
    out->reserved = P.sl_slices > 0 ? 100 + P.sl_lg * 10 + P.sl_vc
                                    : (P.win_slices > 0 ? 2000 + 100 * P.win_u + 10 * (P.win_u == 1 ? (P.win_lpp == 16 ? P.win_vc : 4) : (P.win_u == 4 ? 1 : P.win_vc)) +
                                                              (P.win_u == 1 ? (P.win_occ >= 4 ? 4 : (P.win_lpp == 16 ? 3 : P.win_occ)) : 2)
                                                        : 0);
    for (int s = 0; s < n_maps; ++s)
        if (P.maps[s].runs > 0) {
            const int ru = P.maps[s].unroll, rk = P.maps[s].runs;
            out->reserved = (ru == 1 && rk == 4) ? (P.runs_occ == 6 ? 6 : 7) : (ru == 1 ? ((P.runs_occ == 4 || P.runs_occ == 6) ? P.runs_occ : 5) : ((ru == 2 && rk == 8 && P.runs_occ != 4) ? 3 : 4));
        }
    for (int s = 0; s < D3F_MAX_MAPS; ++s) {
        const bool on = s < n_maps;
        const int c = on ? caller_map[s] : s;
        out->vector_floats[c] = on ? P.maps[s].vw : 0;
        const bool rows0 = P.rows > 0 && s == 0;             // the register-rows kernel: 256 lanes x one vector on a point's row
        out->lanes_per_point[c] = on ? (rows0 ? 256 : ((P.win_slices > 0 && s == 0) ? P.win_lpp : (1 << P.maps[s].lpp_log2))) : 0;
        out->vectors_per_lane[c] = on ? (rows0 ? 1 : ((P.win_slices > 0 && s == 0) ? P.win_u * (32 / P.win_lpp) : P.maps[s].unroll)) : 0;   /* negative: load-use per vector */
        out->staged[c] = on ? (rows0 ? 5 : (P.win_slices > 0 && s == 0 ? 3 : (P.maps[s].runs > 0 ? 16 + P.maps[s].runs : 0))) : 0;
    }
}
