// fuse_body.h -- the generic fused field-query body: phase A (one lane per (point, view)) + phase B through gather_map /
// gather_map_thin (fuse_common.h) or, for patch-resolution wide maps, the cell-run gather below.  Included by fuse_direct.hip (the
// direct kernels) and fuse_runs.hip (the cell-run kernels): one body, one register allocation per entry point.
#pragma once
#include "fuse_common.h"

namespace d3f {

// ---- phase B, cell-run gather (patch-resolution wide maps) ------------------------------------------------
// When a texel spans many image pixels (the reference's dino_feats is (H/10, W/10), fusion.py:694-697) consecutive
// query points of a grid column / a Hilbert walk fall into the SAME texel cell of a view most of the time, and the
// direct gather above is limited by the vector-L1 request rate (64 B/clk/CU), not by misses.  Here a lane group owns a
// RUN of K consecutive points and U 16-byte channel vectors per lane, and walks the run view by view: the four corner
// vectors of a view stay in registers and are re-fetched only when the cell changes (a flag phase A computes once per
// (point, view) by comparing the four corner offsets with the previous point's); the K accumulators carry the view
// sums.  Per (point, view) the operations and their order are exactly those of gather_map's folded fast path -- four
// fma with the folded weights into the view sum, views in order -- so the results are bit-identical.  Points that need the strict path (non-finite projection) are left to gather_map(only_strict).
constexpr uint32_t kRunNonFinite = 1u;     // bits of the per-(point, view) state word (nfp_s)
constexpr uint32_t kRunNewCell = 2u;       // the four corner texels differ from those of the previous point of the run
constexpr uint32_t kRunValid = 4u;         // the view is valid for the point (its corner record is meaningful)

// Branch structure: only the FOUR LOADS of a new cell are conditional.  The arithmetic runs for every (point, view):
// phase A leaves an all-zero corner record for an invalid pair, so its term is (+-0) * wgt = +-0 and adding it to a sum
// that started at +0 changes no bit (the argument of gather_map's exact skip) -- fewer exec-mask
// round trips and LDS waits than skipping it.  A strict point (non-finite projection) takes part like any other and is
// simply not stored here.
// VFIX: the view count as a compile-time constant (4 = the reference's camera rig: LDS record addresses become
// immediates and the view loop unrolls), 0 = read it from the launch parameters.
template <int U, int K, int VFIX>
__device__ __forceinline__ void gather_map_runs(const MapDesc &m, const EvalParams &P, const ViewRec *rec,
                                                const uint32_t *state_s, const float *cnt_s, const uint32_t *flag_s,
                                                const uint32_t *idx_s, int64_t idx_base, int tile_n, const CornerRec *crec)
{
    using VT = f32x4;
    const int lpp = 1 << m.lpp_log2;
    const int g = threadIdx.x & (lpp - 1);
    const int grp = threadIdx.x >> m.lpp_log2;
    const int ngrp = kBlock >> m.lpp_log2;
    const int cvec = m.C / 4;
    const int V = VFIX > 0 ? VFIX : P.V;
    const char *__restrict__ data = reinterpret_cast<const char *>(m.data);
    const int last = tile_n * V - 1;

    for (int run0 = grp * K; run0 < tile_n; run0 += ngrp * K) {
        for (int c0 = 0; c0 < cvec; c0 += lpp * U) {
            uint32_t co[U];
#pragma unroll
            for (int u = 0; u < U; ++u) co[u] = (uint32_t)min(c0 + u * lpp + g, cvec - 1) * 16u;   // idle lanes re-read the last vector
            VT acc[K][U];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[k][u] = (VT)0.0f;
#pragma unroll 1
            for (int v = 0; v < V; ++v) {          // kept rolled: unrolled views let the scheduler interleave them and spill
                const char *bv = data + (int64_t)v * m.sv * 4;
                VT a[U], b[U], d[U], e[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { a[u] = (VT)0.0f; b[u] = (VT)0.0f; d[u] = (VT)0.0f; e[u] = (VT)0.0f; }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool inside = run0 + k < tile_n;
                    const int q = min((run0 + k) * V + v, last);      // beyond the tile: some valid record, result unused
                    const uint32_t st = state_s[q];
                    const CornerRec &cr = crec[q];
                    if (inside && (st & (kRunValid | kRunNewCell)) == (kRunValid | kRunNewCell)) {     // another texel cell
                        const uint32_t o0 = cr.o[0], o1 = cr.o[1], o2 = cr.o[2], o3 = cr.o[3];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            a[u] = load_texel<4, false>(bv + (o0 + co[u]));
                            b[u] = load_texel<4, false>(bv + (o1 + co[u]));
                            d[u] = load_texel<4, false>(bv + (o2 + co[u]));
                            e[u] = load_texel<4, false>(bv + (o3 + co[u]));
                        }
                    }
                    const float w0 = cr.w[0], w1 = cr.w[1], w2 = cr.w[2], w3 = cr.w[3];      // folded (fuse_common.h)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        acc[k][u] = v_fma<VT>(a[u], w0, acc[k][u]);      // corners nw, ne, sw, se; views in order
                        acc[k][u] = v_fma<VT>(b[u], w1, acc[k][u]);
                        acc[k][u] = v_fma<VT>(d[u], w2, acc[k][u]);
                        acc[k][u] = v_fma<VT>(e[u], w3, acc[k][u]);
                    }
                    // keep this point's arithmetic ahead of the next point's fetch: left alone, the optimiser sinks it below
                    // the next conditional load block, which needs a second set of corner registers (and spills).  The
                    // empty asm pins the accumulators (register operands) and, as a memory clobber, the later loads.
#pragma unroll
                    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(acc[k][u]) : : "memory");
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = run0 + k;
                if (p >= tile_n || flag_s[p] != 0u) continue;            // strict points: gather_map(only_strict) writes them
                // the weights carry 1/(cnt + 1e-6) already; no valid view: every weight is zero and so is the sum (fusion.py:386)
                const int64_t row = (idx_base + idx_s[p]) * m.C;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cv = c0 + u * lpp + g;
                    if (cv >= cvec) continue;
                    store_out<VT>(m.out + row + (int64_t)cv * 4, acc[k][u], P.store_policy);
                }
            }
        }
    }
}


// fp16-stored maps: the host maps them to 8-channel (16-B) or scalar lanes with batched loads only (1..3 vectors)
template <int VW>
__device__ __forceinline__ void gather_map_half_u(const MapDesc &m, const EvalParams &P, const ViewRec *rec,
                                                  const float *cnt_s, const uint32_t *flag_s,
                                                  const uint32_t *idx_s, int64_t idx_base, int tile_n, const CornerRec *crec)
{
    if (m.fold) {
        switch (m.unroll) {
        case 1: gather_map<VW, 1, true, true, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case 2: gather_map<VW, 2, true, true, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        default: gather_map<VW, 3, true, true, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        }
        return;
    }
    gather_map<VW, 1, true, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);      // thin: one vector per lane
}


template <int MODE, bool WIDE, bool ANYF16 = false, int RU = 0, int RK = 0>
__device__ __forceinline__ void fused_eval_body(const EvalParams &P)
{
    constexpr bool RUNS = RU > 0;
    extern __shared__ __align__(16) unsigned char smem[];
    const int V = P.V;
    const int TP = P.tile_pts;
    const int TL = P.n_maps == 0 ? 0 : TP;          // the distance-only pass keeps nothing per point in LDS (only KRt): its
                                                    // tiles may be large without costing workgroups per CU
    ViewRec *rec = reinterpret_cast<ViewRec *>(smem);                       // [TP*V]
    float *dcl_s = reinterpret_cast<float *>(rec + (size_t)TL * V);          // [TP*V] (unused since round 4)
    uint32_t *nfp_s = reinterpret_cast<uint32_t *>(dcl_s + (size_t)TL * V);  // [TP*V] per-pair state of the cell-run gather
    float *cnt_s = reinterpret_cast<float *>(nfp_s + (size_t)TL * V);        // [TP]
    uint32_t *flag_s = reinterpret_cast<uint32_t *>(cnt_s + TL);             // [TP]
    uint32_t *idx_s = flag_s + TL;                                           // [TP] global point index
    float *krt = reinterpret_cast<float *>(idx_s + TL);                      // [V*12]
    CornerRec *crec_s = reinterpret_cast<CornerRec *>(smem + P.crec_offset); // [n_pre][TP*V] (wide maps)

    __shared__ TileBox tb_s;                // lattice walk: decoded by one lane (12 integer divisions), read by all
    const bool walk = P.walk_nx > 0;
    const int64_t ntiles = walk ? (int64_t)gridDim.x : (P.n + TP - 1) / TP;
    int64_t tile = (int64_t)blockIdx.x;
    if (P.flags & kFlagXcdRemap) {
        // chunked XCD mapping: the walk is cut into chunks of `xcd_chunk` tiles (0 = one chunk); inside a chunk
        // XCD k takes the k-th contiguous eighth.  Small chunks keep all eight XCDs inside one region of space.
        const int64_t ch = P.xcd_chunk > 0 ? (int64_t)P.xcd_chunk : ntiles;
        const int64_t c0 = ((int64_t)blockIdx.x / ch) * ch;
        const int64_t len = min(ch, ntiles - c0);
        tile = c0 + xcd_tile((int64_t)blockIdx.x - c0, len);
    }
    if (walk && threadIdx.x == 0) tb_s = walk_tile(P, tile);
    compute_krt(P.K, P.pose, V, krt, kBlock);
    __syncthreads();
    const int64_t tile_base = tile * TP;
    TileBox tb = {0, 0, 0, 0, 0, 0};
    if (walk) tb = tb_s;
    const int tile_n = walk ? tb.sx * tb.sy * tb.sz : (int)min((int64_t)TP, P.n - tile_base);
    const int64_t idx_base = (P.order || walk) ? 0 : tile_base;   // idx_s holds 32-bit offsets from here
    const float mu = P.mu;
    const float Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);

    // ---------------- phase A ----------------
    if (P.n_maps == 0) {
        // distance-only query (return_names=[], eval_dist): one lane per point, nothing staged in LDS
        for (int p = threadIdx.x; p < tile_n; p += kBlock) {
            const int64_t i = tile_base + p;
            float px, py, pz;
            fetch_point(P, i, px, py, pz);
            float dsum = 0.0f, cnt = 0.0f;
            for (int v = 0; v < V; ++v) {
                float wgt;
                const ViewOut o = eval_view<MODE>(P.depth, P.H, P.W, krt + v * 12, v, px, py, pz, Wm1, Hm1, mu, wgt);
                dsum = dsum + o.dist * o.valid;                             // fusion.py:364
                cnt = cnt + o.valid;
            }
            const bool all_invalid = (cnt == 0.0f);                         // fusion.py:366
            float dist_out = dsum / (cnt + 1e-6f);
            if (MODE == 0 && all_invalid) dist_out = 1e3f;                  // fusion.py:367
            P.out_dist[i] = dist_out;
            P.out_valid[i] = all_invalid ? 0 : 1;
        }
        return;
    }
    // One lane per (point, view) pair, the views of a point in VP = 2^k >= V adjacent lanes: the V depth lookups of a point
    // are in flight together, the per-point sums over the views are rebuilt IN VIEW ORDER with wave shuffles (no second
    // pass over LDS), and every pair knows the point's view count when it writes its records -- which is what lets the
    // folded weights of wide maps (fuse_common.h) be final here.  Whole waves iterate (shuffles).
    {
        const bool finite_maps = maps_are_finite(P);
        const int vp_log2 = view_lanes_log2(V), VP = 1 << vp_log2;
        const int lane = threadIdx.x & 63, base = lane & ~(VP - 1);
        const int npair = tile_n << vp_log2;
        for (int idx0 = (int)(threadIdx.x & ~63u); idx0 < npair; idx0 += kBlock) {
            const int idx = idx0 + lane;
            const bool in = idx < npair;
            const int p = min(idx >> vp_log2, tile_n - 1), v = idx & (VP - 1);
            const bool act = in && v < V;
            // (indices are clamped: a stale buffer passed with D3F_FLAG_REUSE_POINT_ORDER must not fault the device)
            const int64_t i = walk ? walk_point(P, tb, p) : (P.order ? min((int64_t)P.order[tile_base + p], P.n - 1) : tile_base + p);
            ViewOut o;
            o.gx = 0.0f; o.gy = 0.0f; o.dist = 0.0f; o.valid = 0.0f;
            float wgt = 0.0f;
            uint32_t st = 0u;
            if (act) {
                float px, py, pz;
                fetch_point(P, i, px, py, pz);
                o = eval_view<MODE>(P.depth, P.H, P.W, krt + v * 12, v, px, py, pz, Wm1, Hm1, mu, wgt);
                if (!(isfinite(o.gx) && isfinite(o.gy) && isfinite(wgt))) st = kRunNonFinite;
            }
            float dsum, cnt;
            uint32_t nonfinite;
            view_sums(V, base, o.dist * o.valid, o.valid, st, dsum, cnt, nonfinite);      // fusion.py:364 (products), :368
            const float fsc = fold_scale(wgt, cnt);
            uint32_t c0 = 0u, c1 = 0u, c2 = 0u, c3 = 0u;          // corner offsets of the first cell-run map (slot 0)
            if (act) {
                ViewRec r;
                r.gx = o.gx; r.gy = o.gy; r.wgt = wgt; r.valid = o.valid;
                rec[p * V + v] = r;
                for (int s = 0; s < P.n_maps; ++s) {
                    const MapDesc &m = P.maps[s];
                    if (m.pre_slot >= 0 && o.valid != 0.0f) {
                        const Corner c = corner_setup(m, o.gx, o.gy);
                        const float sc = m.fold ? fsc : 1.0f;
                        CornerRec cr;
                        cr.o[0] = c.onw; cr.o[1] = c.one; cr.o[2] = c.osw; cr.o[3] = c.ose;
                        cr.w[0] = c.inw ? c.wnw : 0.0f; cr.w[1] = c.ine ? c.wne : 0.0f;
                        cr.w[2] = c.isw ? c.wsw : 0.0f; cr.w[3] = c.ise ? c.wse : 0.0f;
                        if (m.fold) { cr.w[0] = cr.w[0] * sc; cr.w[1] = cr.w[1] * sc; cr.w[2] = cr.w[2] * sc; cr.w[3] = cr.w[3] * sc; }
                        crec_s[(size_t)m.pre_slot * TP * V + p * V + v] = cr;
                        if (RUNS && m.pre_slot == 0) { c0 = c.onw; c1 = c.one; c2 = c.osw; c3 = c.ose; }
                    } else if (RUNS && m.runs > 0) {
                        // the cell-run gather multiplies instead of branching: an invalid pair contributes +-0
                        CornerRec cr;
                        cr.o[0] = cr.o[1] = cr.o[2] = cr.o[3] = 0u;
                        cr.w[0] = cr.w[1] = cr.w[2] = cr.w[3] = 0.0f;
                        crec_s[(size_t)m.pre_slot * TP * V + p * V + v] = cr;
                    }
                }
            }
            if (RUNS) {
                // cell-run gather: does this pair address the same four texels (of the first cell-run map) as the previous
                // point of the tile?  The previous point's pair of this view is VP lanes down; it counts only if it is valid
                // too -- an invalid or strict predecessor is handled by the consumer (the chain breaks there).
                const uint32_t q0 = __shfl_up(c0, VP, 64), q1 = __shfl_up(c1, VP, 64), q2 = __shfl_up(c2, VP, 64), q3 = __shfl_up(c3, VP, 64);
                const float pv = __shfl_up(o.valid, VP, 64);
                // a run starts at every RK-th point of the tile: its first valid pair always fetches
                const bool same = lane >= VP && (p % (RK > 0 ? RK : 1)) != 0 && pv != 0.0f && q0 == c0 && q1 == c1 && q2 == c2 && q3 == c3;
                if (!same) st |= kRunNewCell;
                if (o.valid != 0.0f) st |= kRunValid;
            }
            if (act) nfp_s[p * V + v] = st;
            if (in && v == 0) {
                // per point: outputs leave from the lane of view 0
                const bool all_invalid = (cnt == 0.0f);                             // fusion.py:366
                float dist_out = dsum / (cnt + 1e-6f);
                if (MODE == 0 && all_invalid) dist_out = 1e3f;                      // fusion.py:367
                P.out_dist[i] = dist_out;
                P.out_valid[i] = all_invalid ? 0 : 1;
                cnt_s[p] = cnt;
                idx_s[p] = (uint32_t)(i - idx_base);
                flag_s[p] = (nonfinite || !finite_maps) ? 1u : 0u;
            }
        }
    }
    __syncthreads();

    // ---------------- phase B: per map, 2^k lanes per point ----------------
    for (int s = 0; s < P.n_maps; ++s) {
        const MapDesc &m = P.maps[s];
        const CornerRec *crec = m.pre_slot >= 0 ? crec_s + (size_t)m.pre_slot * TP * V : nullptr;
        if (RUNS && m.runs > 0) {
            // non-strict points through the cell-run gather, the (rare) strict ones through the generic path
            // (the host gives such a map 16-byte vectors, one per lane, and a corner-record slot)
            if (V == 4) gather_map_runs<(RU > 0 ? RU : 1), (RK > 0 ? RK : 1), 4>(m, P, rec, nfp_s, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);
            else gather_map_runs<(RU > 0 ? RU : 1), (RK > 0 ? RK : 1), 0>(m, P, rec, nfp_s, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);
            gather_map<4, (RU > 0 ? RU : 1), true, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec, true);
            continue;
        }
        if (ANYF16 && m.esize == 2) {
            if (m.vw == 8) gather_map_half_u<8>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);
            else gather_map_half_u<1>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);
            continue;
        }
        switch (m.vw) {
        case 4: gather_map_u<4, WIDE, RUNS>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case 2: gather_map_u<2, WIDE, RUNS>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        default: gather_map_u<1, WIDE, RUNS>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        }
    }
}

}  // namespace d3f
