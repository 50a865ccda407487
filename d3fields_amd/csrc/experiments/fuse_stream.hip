// fuse_stream.hip -- the channel-sliced field query as PERSISTENT producer / consumer workgroups (round 3).
//
// Same work split as the channel-sliced launch it replaces: on maps far larger than the caches the
// query is bound by L2 misses, so a texel's channels are cut into slices, the slices of a stretch of the lattice walk
// are "units" spread over the XCDs, and every XCD's 4 MiB L2 sees only 1/S of every texel it touches.  What changes is
// how a workgroup spends its time.  The round-2 kernel was one short-lived workgroup per 16 points and slice: load K and
// pose, barrier, project on a quarter of the lanes, barrier, reduce, barrier, gather, exit -- three dependent global
// round trips and three barriers with no texel load in flight, 40 % of its lifetime (fabric read latency 930 cycles ==
// unloaded: the memory system was waiting for the kernel, profiles/r2_v3/c2_dense_summary.txt).  Here:
//   * a workgroup lives for R tiles of its unit (KRt once);
//   * its LAST wave is the PRODUCER: projection, depth test, weights and the bilinear corner set-up of tile t+2, one lane
//     per (point, view) with the four views of a point in adjacent lanes (ordered view sums by wave shuffles), written
//     as records into one of three LDS buffers;
//   * the other NC waves are CONSUMERS: LP lanes per point, they only read records, issue the corner loads of round r+1
//     BEFORE they consume round r (two register sets), and prefetch the first round of tile t+1 before the tile barrier,
//     so a consumer always has 8..16 texel loads in flight;
//   * one barrier per tile instead of three per workgroup.
// Arithmetic per (point, view, channel) is gather_map's fast path (strict points: its strict path), so results are
// bit-identical to the direct gather (tests/test_gpu_walks.py::test_stream_launch_is_bit_identical).
//
// Unit u = (chunk of WGU*R consecutive walk tiles, slice) runs on XCD u % 8 (its workgroups are consecutive in that XCD's
// dispatch stream); workgroup g of the unit takes tiles g, g + WGU, g + 2 WGU, ... of the chunk, so the ~WGU workgroups
// resident on the XCD sweep the chunk front to back together and the points open on an XCD at any time stay a compact
// stretch of the walk (the L2 window).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fuse_common.h"

namespace d3f {

constexpr int kStreamBufs = 3;
constexpr int kTicketStride = 64;          // uints between the eight ticket counters: one 256-byte line each (atomics on ONE line
                                           // serialise at ~88 per microsecond chip-wide, measured: 3.1 ms for 246 k tickets)

// LDS bytes of one record buffer for T points, V views
__host__ __device__ constexpr int stream_buf_bytes(int T, int V) { return T * V * (32 + 16 + 4) + T * 12; }

template <int V, int T>
struct StreamBuf {
    CornerRec *crec;    // [T*V]  corner offsets + zero-padded weights (all zero for an invalid pair)
    ViewRec *vrec;      // [T*V]  gx, gy, wgt, valid (strict points, thin maps)
    float *wgt;         // [T*V]
    float *cnt;         // [T]
    uint32_t *flag;     // [T]    1: the point takes the strict path
    uint32_t *idx;      // [T]    flat point index
    __device__ __forceinline__ StreamBuf(unsigned char *base)
    {
        crec = reinterpret_cast<CornerRec *>(base);
        vrec = reinterpret_cast<ViewRec *>(crec + T * V);
        wgt = reinterpret_cast<float *>(vrec + T * V);
        cnt = wgt + T * V;
        flag = reinterpret_cast<uint32_t *>(cnt + T);
        idx = flag + T;
    }
};

// ---- producer: phase A of one tile (fuse_eval.hip, fused_eval_body), one lane per (point, view) ----------------------
// Split in two so that the producer wave can run as a software pipeline: stream_fetch (tile decode + the point loads of
// tile i+1) is issued while the depth loads of tile i are in flight, and the ticket of tile i+2 before that.
template <int V, int T>
struct StreamPts {
    static constexpr int NP = (T * V + 63) / 64;    // passes of the producer wave over the tile's (point, view) pairs
    float px[NP], py[NP], pz[NP];
    uint32_t gi[NP];                                // flat point index (n < 2^31, host-checked)
};

template <int V, int T>
__device__ __forceinline__ void stream_fetch(StreamPts<V, T> &S, const EvalParams &P, int64_t tile, int lane)
{
    const TileBox tb = walk_tile(P, tile);                   // wave-uniform
    const int ltz = P.st_ltz, lty = P.st_lty;
#pragma unroll
    for (int q = 0; q < StreamPts<V, T>::NP; ++q) {
        const int idx = min(lane + 64 * q, T * V - 1);
        const int p = idx >> 2;
        // slots of a clipped brick repeat a neighbouring point: same inputs, same outputs, written twice
        const int dz = min(p & ((1 << ltz) - 1), tb.sz - 1), dy = min((p >> ltz) & ((1 << lty) - 1), tb.sy - 1);
        const int dx = min(p >> (ltz + lty), tb.sx - 1);
        const int64_t gi = ((int64_t)(tb.ox + dx) * P.walk_ny + (tb.oy + dy)) * P.walk_nz + (tb.oz + dz);
        S.gi[q] = (uint32_t)gi;
        if (P.grid_x) {
            S.px[q] = P.grid_x[tb.ox + dx]; S.py[q] = P.grid_y[tb.oy + dy]; S.pz[q] = P.grid_z[tb.oz + dz];
        } else {
            S.px[q] = P.pts[gi * 3 + 0]; S.py[q] = P.pts[gi * 3 + 1]; S.pz[q] = P.pts[gi * 3 + 2];
        }
    }
}

// projection + nearest-depth LOAD of one (point, view): the first half of eval_view<0> (d3f_device.h)
struct StreamProj {
    Proj pr;
    float d;
};

__device__ __forceinline__ StreamProj stream_project(const EvalParams &P, const float *M, int v, float px, float py, float pz)
{
    StreamProj r;
    r.pr = project_point(M, px, py, pz, (float)(P.W - 1), (float)(P.H - 1));
    r.d = nearest_depth(P.depth, v, P.H, P.W, r.pr.gx, r.pr.gy);
    return r;
}

// the rest of eval_view<0> and the records of one (point, view) pair; the four views of a point sit in adjacent lanes
template <int V, int T>
__device__ __forceinline__ void stream_records(const EvalParams &P, const StreamProj &sp, uint32_t gi, int idx,
                                               unsigned char *buf_base, bool writes_point_outputs, int lane)
{
    static_assert(V == 4, "the producer's lane layout (four views of a point in adjacent lanes) is built for V = 4");
    StreamBuf<V, T> B(buf_base);
    const MapDesc &m0 = P.maps[0];
    const float mu = P.mu;
    const int p = idx >> 2, v = idx & 3;
    // eval_view<0>, second half (fusion.py:343-358)
    float dist = sp.d - sp.pr.zc;                                               // fusion.py:343
    const bool valid = (sp.d > 0.0f) && sp.pr.ok && (dist > -mu);               // fusion.py:344
    float t = mu - fabsf(dist);                                                 // fusion.py:347
    t = t > 0.0f ? 0.0f : t;
    const float wgt = expf(t / mu);
    float dc = dist < -mu ? -mu : dist;                                         // fusion.py:358
    dc = dc > mu ? mu : dc;
    const float validf = valid ? 1.0f : 0.0f;
    const float gx = sp.pr.gx, gy = sp.pr.gy;
    ViewRec r;
    r.gx = gx; r.gy = gy; r.wgt = wgt; r.valid = validf;
    B.vrec[idx] = r;
    B.wgt[idx] = wgt;
    CornerRec cr;
    if (valid) {
        const Corner c = corner_setup(m0, gx, gy);
        cr.o[0] = c.onw; cr.o[1] = c.one; cr.o[2] = c.osw; cr.o[3] = c.ose;
        cr.w[0] = c.inw ? c.wnw : 0.0f; cr.w[1] = c.ine ? c.wne : 0.0f;
        cr.w[2] = c.isw ? c.wsw : 0.0f; cr.w[3] = c.ise ? c.wse : 0.0f;
    } else {                                    // invalid pair: texel 0 with zero weights, its term is (+-0) * wgt
        cr.o[0] = cr.o[1] = cr.o[2] = cr.o[3] = 0u;
        cr.w[0] = cr.w[1] = cr.w[2] = cr.w[3] = 0.0f;
    }
    B.crec[idx] = cr;
    const float dv = dc * validf;                                               // fusion.py:364 (product only)
    const uint32_t st = !(isfinite(gx) && isfinite(gy) && isfinite(wgt)) ? 1u : 0u;
    // sums over the views in view order (fusion.py:364-370)
    const int base = lane & ~3;
    float dsum = 0.0f, cnt = 0.0f;
    uint32_t stp = 0u;
#pragma unroll
    for (int vv = 0; vv < V; ++vv) {
        dsum = dsum + __shfl(dv, base + vv, 64);
        cnt = cnt + __shfl(validf, base + vv, 64);
        stp |= (uint32_t)__shfl((int)st, base + vv, 64);
    }
    if (v == 0) {
        const bool all_invalid = (cnt == 0.0f);                             // fusion.py:366
        float dist_out = dsum / (cnt + 1e-6f);
        if (all_invalid) dist_out = 1e3f;                                   // fusion.py:367
        if (writes_point_outputs) {
            P.out_dist[gi] = dist_out;
            P.out_valid[gi] = all_invalid ? 0 : 1;
        }
        B.cnt[p] = cnt;
        B.idx[p] = gi;
        B.flag[p] = (stp != 0u || !(P.flags & kFlagFiniteMaps)) ? 1u : 0u;
    }
}

// ---- geometry pre-pass (experiment): phase A ONCE per point into a compact record stream in walk order --------------
// Finer channel slices make the L2 hold more texels but repeat phase A per (point, slice).  With the geometry computed by
// a small first launch -- per (point, view) the 16-byte ViewRec, per point {flat index, view count, strict flag} -- the
// producer of the gather launch only loads and expands records (corner set-up), whatever the slice count.
struct __attribute__((aligned(16))) StreamAux {
    uint32_t gi;
    float cnt;
    uint32_t flag;
    uint32_t pad;
};

template <int V, int T>
__global__ __launch_bounds__(256) void stream_prepass_kernel(const EvalParams P)
{
    __shared__ float krt[V * 12];
    compute_krt(P.K, P.pose, V, krt, 256);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= P.sl_tiles) return;
    StreamPts<V, T> S;
    stream_fetch<V, T>(S, P, tile, lane);
    ViewRec *rec = reinterpret_cast<ViewRec *>(P.st_rec) + tile * (T * V);
    StreamAux *aux = reinterpret_cast<StreamAux *>(P.st_aux) + tile * T;
    const float mu = P.mu;
#pragma unroll
    for (int q = 0; q < StreamPts<V, T>::NP; ++q) {
        const int idx = lane + 64 * q;
        if (idx >= T * V) break;
        const int p = idx >> 2, v = idx & 3;
        const StreamProj sp = stream_project(P, krt + v * 12, v, S.px[q], S.py[q], S.pz[q]);
        float dist = sp.d - sp.pr.zc;                                               // fusion.py:343
        const bool valid = (sp.d > 0.0f) && sp.pr.ok && (dist > -mu);               // fusion.py:344
        float t = mu - fabsf(dist);                                                 // fusion.py:347
        t = t > 0.0f ? 0.0f : t;
        const float wgt = expf(t / mu);
        float dc = dist < -mu ? -mu : dist;                                         // fusion.py:358
        dc = dc > mu ? mu : dc;
        const float validf = valid ? 1.0f : 0.0f;
        ViewRec r;
        r.gx = sp.pr.gx; r.gy = sp.pr.gy; r.wgt = wgt; r.valid = validf;
        rec[idx] = r;
        const float dv = dc * validf;
        const uint32_t st = !(isfinite(r.gx) && isfinite(r.gy) && isfinite(wgt)) ? 1u : 0u;
        const int base = lane & ~3;
        float dsum = 0.0f, cnt = 0.0f;
        uint32_t stp = 0u;
#pragma unroll
        for (int vv = 0; vv < V; ++vv) {
            dsum = dsum + __shfl(dv, base + vv, 64);
            cnt = cnt + __shfl(validf, base + vv, 64);
            stp |= (uint32_t)__shfl((int)st, base + vv, 64);
        }
        if (v == 0) {
            const bool all_invalid = (cnt == 0.0f);
            float dist_out = dsum / (cnt + 1e-6f);
            if (all_invalid) dist_out = 1e3f;
            P.out_dist[S.gi[q]] = dist_out;
            P.out_valid[S.gi[q]] = all_invalid ? 0 : 1;
            StreamAux a;
            a.gi = S.gi[q]; a.cnt = cnt; a.flag = (stp != 0u || !(P.flags & kFlagFiniteMaps)) ? 1u : 0u; a.pad = 0u;
            aux[p] = a;
        }
    }
}

// what the producer carries per (point, view) lane in pre-pass mode
struct StreamLoaded {
    ViewRec r;
    StreamAux a;        // lanes with view 0 only
};

template <int V, int T>
__device__ __forceinline__ void stream_records_pre(const EvalParams &P, const StreamLoaded &L, int idx, unsigned char *buf_base)
{
    StreamBuf<V, T> B(buf_base);
    const MapDesc &m0 = P.maps[0];
    const int p = idx >> 2, v = idx & 3;
    B.vrec[idx] = L.r;
    B.wgt[idx] = L.r.wgt;
    CornerRec cr;
    if (L.r.valid != 0.0f) {
        const Corner c = corner_setup(m0, L.r.gx, L.r.gy);
        cr.o[0] = c.onw; cr.o[1] = c.one; cr.o[2] = c.osw; cr.o[3] = c.ose;
        cr.w[0] = c.inw ? c.wnw : 0.0f; cr.w[1] = c.ine ? c.wne : 0.0f;
        cr.w[2] = c.isw ? c.wsw : 0.0f; cr.w[3] = c.ise ? c.wse : 0.0f;
    } else {
        cr.o[0] = cr.o[1] = cr.o[2] = cr.o[3] = 0u;
        cr.w[0] = cr.w[1] = cr.w[2] = cr.w[3] = 0.0f;
    }
    B.crec[idx] = cr;
    if (v == 0) {
        B.cnt[p] = L.a.cnt;
        B.idx[p] = L.a.gi;
        B.flag[p] = L.a.flag;
    }
}

// ---- consumer ----------------------------------------------------------------------------------------------------------
// One round = the corner vectors of VC views of the PW points a wave serves at a time.
template <int VC>
struct StreamRound {
    f32x4 a[VC], b[VC], d[VC], e[VC];
};

template <int V, int T, int VC>
__device__ __forceinline__ void stream_issue(StreamRound<VC> &R, const StreamBuf<V, T> &B, const char *__restrict__ data,
                                             int64_t sv_bytes, int p, int v0, uint32_t co)
{
#pragma unroll
    for (int q = 0; q < VC; ++q) {
        const uint32_t *o = B.crec[p * V + v0 + q].o;
        const f32x4 ov = *reinterpret_cast<const f32x4 *>(o);           // one ds_read_b128: the four corner offsets
        const char *bv = data + (int64_t)(v0 + q) * sv_bytes;
        R.a[q] = load_texel<4, false>(bv + (__float_as_uint(ov.x) + co));
        R.b[q] = load_texel<4, false>(bv + (__float_as_uint(ov.y) + co));
        R.d[q] = load_texel<4, false>(bv + (__float_as_uint(ov.z) + co));
        R.e[q] = load_texel<4, false>(bv + (__float_as_uint(ov.w) + co));
    }
}

template <int V, int T, int VC>
__device__ __forceinline__ void stream_consume(f32x4 &acc, const StreamRound<VC> &R, const StreamBuf<V, T> &B, int p, int v0)
{
#pragma unroll
    for (int q = 0; q < VC; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(B.crec[p * V + v0 + q].w);
        const float wg = B.wgt[p * V + v0 + q];
        f32x4 s_ = R.a[q] * w.x;                                    // ATen bilinear: fma chain nw,ne,sw,se
        s_ = v_fma<f32x4>(R.b[q], w.y, s_);
        s_ = v_fma<f32x4>(R.d[q], w.z, s_);
        s_ = v_fma<f32x4>(R.e[q], w.w, s_);
        acc = acc + s_ * wg;                                        // fusion.py:385
    }
}

// division by (cnt + 1e-6) and the store of one point's slice (gather_map's fast path); strict points are left to
// stream_strict_point
template <int V, int T>
__device__ __forceinline__ void stream_finish(f32x4 acc, const EvalParams &P, const MapDesc &m, const StreamBuf<V, T> &B, int p,
                                              uint32_t co)
{
    using VT = f32x4;
    if (B.flag[p] != 0u) return;
    const float denom = B.cnt[p] + 1e-6f;                           // fusion.py:385
    const int64_t i = B.idx[p];
    // the shared-reciprocal IEEE division of gather_map's fast path; no view valid: every term was (+-0) * wgt,
    // acc is +0 and so is the quotient -- fusion.py:386 for free
    const float r0 = __builtin_amdgcn_rcpf(denom);
    const float rcp_d = fmaf(fmaf(-denom, r0, 1.0f), r0, r0);
    VT q = acc * rcp_d;
    q = v_fma<VT>(v_fma<VT>(q, -denom, acc), rcp_d, q);
    q = v_fma<VT>(v_fma<VT>(q, -denom, acc), rcp_d, q);
    store_out<VT>(m.out + i * m.C + (co >> 2), q, P.store_policy);
}

// a strict point (non-finite projection): gather_map's strict arithmetic on direct loads
template <int V, int T>
__device__ __forceinline__ void stream_strict_point(const EvalParams &P, const MapDesc &m, const StreamBuf<V, T> &B,
                                                    const char *__restrict__ data, int p, uint32_t co)
{
    using VT = f32x4;
    const float cnt = B.cnt[p];
    const float denom = cnt + 1e-6f;
    const int64_t i = B.idx[p];
    VT acc = (VT)0.0f;
#pragma unroll 1
    for (int v = 0; v < V; ++v) {
        const ViewRec r = B.vrec[p * V + v];
        const char *bv = data + (int64_t)v * m.sv * 4;
        const Corner c = corner_setup(m, r.gx, r.gy);
        const VT a = load_texel<4, false>(bv + (c.onw + co)), b = load_texel<4, false>(bv + (c.one + co));
        const VT d = load_texel<4, false>(bv + (c.osw + co)), e = load_texel<4, false>(bv + (c.ose + co));
        const VT av = c.inw ? a : (VT)0.0f, bvv = c.ine ? b : (VT)0.0f, dv = c.isw ? d : (VT)0.0f, ev = c.ise ? e : (VT)0.0f;
        VT s_ = av * c.wnw;
        s_ = v_fma<VT>(bvv, c.wne, s_);
        s_ = v_fma<VT>(dv, c.wsw, s_);
        s_ = v_fma<VT>(ev, c.wse, s_);
        acc = acc + (s_ * r.valid) * r.wgt;
    }
    VT o = (VT)0.0f;                                                // fusion.py:386
    if (cnt != 0.0f) o = strict_div<VT>(acc, denom);
    store_out<VT>(m.out + i * m.C + (co >> 2), o, P.store_policy);
}

// thin maps of the call (mask, colours) for one tile: gather_map_thin / the one-vector direct gather on the consumer lanes
template <int V, int T, int VW>
__device__ __forceinline__ void stream_thin(const MapDesc &m, const EvalParams &P, const StreamBuf<V, T> &B, int tid, int nth)
{
    if (thin_map(m, P, VW)) gather_map_thin<VW>(m, P, B.vrec, B.cnt, B.flag, B.idx, 0, T, tid, nth);
    else gather_map<VW, 1, true>(m, P, B.vrec, B.cnt, B.flag, B.idx, 0, T, nullptr, false, tid, nth);
}

// Per record buffer: what the producer decided for that tile
struct __attribute__((aligned(16))) StreamHdr {
    int32_t live;       // 0: no more tiles -- everybody leaves
    int32_t slice;      // channel slice of this tile's unit
    int32_t owner;      // this unit writes the tile's dist / valid_mask and gathers its thin maps
    int32_t pad;
};

// Tile hand-out.  With tickets (P.st_tickets: eight zeroed counters, one per XCD stream) the persistent workgroups of
// an XCD pull the tiles of that XCD's units strictly in order, one at a time, as they become free -- the same smooth
// front the hardware dispatcher gives short-lived workgroups: neighbouring tiles are processed shortly AFTER one another,
// not at the same instant, so the second one finds the shared texels in the L2 (lock-step sweeps of a static assignment
// measured 27 % more L2 fills, profiles/r3_*).  A workgroup whose own stream is dry steals from the others, so every
// tile is processed whatever the block -> XCD placement really is.  Without tickets: workgroup g of a unit takes tiles
// g, g + WGU, ... of the unit's chunk (static).
struct StreamCursor {
    int64_t tile;
    int slice;
    bool owner;
};

template <int LG, int T, int NC, int VC, bool PIPE, bool PRE>
__device__ __forceinline__ void fused_eval_stream_body(const EvalParams &P)
{
    constexpr int V = 4;
    constexpr int LP = 1 << LG, PW = 64 / LP;           // lanes per point, points per wave step
    static_assert(T % (PW * NC) == 0, "every consumer wave takes the same number of steps per tile");
    constexpr int KS = T / (PW * NC);                   // steps per consumer wave and tile
    constexpr int RPS = V / VC;                         // rounds per step
    constexpr int NR = KS * RPS;                        // rounds per consumer wave and tile
    static_assert(!PIPE || NR % 2 == 0, "the register sets alternate per round; a tile must leave them where it found them");
    constexpr int BUF = (stream_buf_bytes(T, V) + 15) / 16 * 16;
    extern __shared__ __align__(16) unsigned char smem[];
    StreamHdr *hdr = reinterpret_cast<StreamHdr *>(smem + kStreamBufs * BUF);          // [kStreamBufs]
    float *krt = reinterpret_cast<float *>(hdr + kStreamBufs);                          // [V*12]

    const int xcd = (int)(blockIdx.x & 7u);
    const int S = P.sl_slices;
    const int64_t CT = (int64_t)P.sl_unit * P.st_R;                  // tiles per chunk
    const int64_t units = (int64_t)P.sl_chunks * S;

    compute_krt(P.K, P.pose, V, krt, (NC + 1) * 64);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool producer = wave == NC;
    const MapDesc &m = P.maps[0];
    const char *__restrict__ data = reinterpret_cast<const char *>(m.data);
    const int64_t svb = m.sv * 4;
    auto slot = [&](int k) -> int { return (k * NC + wave) * PW + (lane >> LG); };   // neighbouring waves, neighbouring points
    auto lane_offset = [&](int slice) -> uint32_t { return (uint32_t)(slice * LP + (lane & (LP - 1))) * 16u; };

    // Producer and consumers run separate loops (their live registers never overlap) that meet at one barrier per tile.
    // Records are triple-buffered: during tile t the producer writes tile t+2, the consumers read tile t and, for the
    // prefetch of their first round, tile t+1 (complete since the previous barrier).
    if (producer) {
        // static assignment state
        const int64_t j = (int64_t)(blockIdx.x >> 3);
        const int64_t s_unit = (j / P.sl_unit) * 8 + xcd;
        const int64_t s_chunk = s_unit / S;
        const int s_slice = (int)(s_unit - s_chunk * S);
        int64_t s_tile = s_chunk * CT + (j % P.sl_unit);
        int s_left = (s_unit < units) ? P.st_R : 0;
        int steal = 0;                                              // streams given up on (ticket mode)
        bool dry = false;
        const bool tickets = P.st_tickets != nullptr;
        // a ticket is ISSUED (returning atomic, lane 0) one producer step before it is RESOLVED, so its round trip runs
        // under the records of the tile in between
        auto ticket_issue = [&]() -> unsigned int {
            unsigned int k = 0;
            if (tickets && !dry && lane == 0) k = atomicAdd(P.st_tickets + ((xcd + steal) & 7) * kTicketStride, 1u);
            return k;
        };
        // (32-bit arithmetic throughout: tiles, units and tickets are below 2^31, host-checked; a 64-bit division is a loop)
        const uint32_t uCT = (uint32_t)CT, uS = (uint32_t)S, uunits = (uint32_t)units, utiles = (uint32_t)P.sl_tiles;
        auto ticket_resolve = [&](unsigned int kraw, int ksx, StreamCursor &c) -> bool {
            if (dry) return false;
            if (!tickets) {
                if (s_left <= 0 || s_tile >= P.sl_tiles || s_tile >= (s_chunk + 1) * CT) { dry = true; return false; }
                c.tile = s_tile; c.slice = s_slice;
                s_tile += P.sl_unit; --s_left;
            } else {
                uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)kraw);
                int sx = ksx;                                       // the stream this ticket was drawn from
                for (;;) {
                    const uint32_t unit = (k / uCT) * 8u + (uint32_t)sx;            // the stream's units: sx, sx + 8, ...
                    if (unit < uunits) {
                        const uint32_t chunk = unit / uS;
                        const uint32_t tile = chunk * uCT + k % uCT;
                        if (tile < utiles) { c.tile = (int64_t)tile; c.slice = (int)(unit - chunk * uS); break; }
                        // the short last chunk: a void ticket -- draw again from the same stream
                    } else {
                        // stream sx is dry: move on to the next one (only the CURRENT stream advances the steal count)
                        if (sx == ((xcd + steal) & 7) && ++steal >= 8) { dry = true; return false; }
                        sx = (xcd + steal) & 7;
                    }
                    unsigned int k2 = 0;                                            // rare: a synchronous ticket
                    if (lane == 0) k2 = atomicAdd(P.st_tickets + sx * kTicketStride, 1u);
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k2);
                }
            }
            // dist / valid_mask of a tile are written by ONE of its S units, and the thin maps ride along with that
            // unit too: tile % S, so that the extra work is spread evenly over the slices (round 2: all on slice 0)
            c.owner = (int)((uint32_t)c.tile % uS) == c.slice;
            return true;
        };
        // The producer is a four-stage software pipeline over tiles, one stage per loop iteration, so that NO result is
        // used in the iteration that issued its memory operation (under load a producer load queues ~1-2 us behind the
        // consumers' bursts in the CU's texture-address FIFO; exposed, that made the producer the slower side):
        //   T(j) ticket (returning atomic)   R(j) resolve + tile decode + point loads   D(j) projection + depth loads
        //   W(j) weights, corner set-up, records -> LDS buffer j % 3
        // iteration `it` runs T(it+3), D(it+1), R(it+2), W(it).
        constexpr int NP = StreamPts<V, T>::NP;
        StreamPts<V, T> pts_d;                              // points of the tile entering D
        StreamProj sp_w[NP];                                // projection + depth of the tile entering W
        uint32_t gi_w[NP];
        StreamLoaded ld_d[NP], ld_w[NP];                    // pre-pass mode: the loaded records ride through the same stages
        StreamCursor c_d, c_w;
        bool live_d = false, live_w = false;
        unsigned int kraw = 0;
        int ksx = 0;
        bool have_ticket = false;
        int b = 0, cur = 0;
#pragma unroll 1
        for (int it = -3;; ++it) {
            if (it >= 2 && hdr[cur].live == 0) break;
            const bool idle = P.st_debug == 2 && it >= 3;               // experiment 2: stale records (consumer-bound time)
            // T(it+3)
            const int ksx_new = (xcd + steal) & 7;
            const unsigned int kraw_new = ticket_issue();
            // D(it+1): the points were loaded an iteration ago
            StreamProj sp_d[NP];
            if (!PRE && live_d && !idle) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int idx = min(lane + 64 * q, T * V - 1);
                    sp_d[q] = stream_project(P, krt + (idx & 3) * 12, idx & 3, pts_d.px[q], pts_d.py[q], pts_d.pz[q]);
                }
            }
            // R(it+2): the ticket was taken an iteration ago
            StreamCursor c_r;
            bool live_r = false;
            StreamPts<V, T> pts_r;
            StreamLoaded ld_r[NP];
            if (have_ticket) {
                live_r = ticket_resolve(kraw, ksx, c_r);
                if (live_r && !idle) {
                    if (PRE) {
                        const ViewRec *rec = reinterpret_cast<const ViewRec *>(P.st_rec) + c_r.tile * (T * V);
                        const StreamAux *aux = reinterpret_cast<const StreamAux *>(P.st_aux) + c_r.tile * T;
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const int idx = min(lane + 64 * q, T * V - 1);
                            ld_r[q].r = rec[idx];
                            ld_r[q].a = aux[idx >> 2];
                        }
                    } else {
                        stream_fetch<V, T>(pts_r, P, c_r.tile, lane);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // W(it): projection and depth were issued an iteration ago
            if (it >= 0) {
                if (live_w && !idle) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const int idx = lane + 64 * q;
                        if (idx < T * V) {
                            if (PRE) stream_records_pre<V, T>(P, ld_w[q], idx, smem + b * BUF);
                            else stream_records<V, T>(P, sp_w[q], gi_w[q], idx, smem + b * BUF, c_w.owner, lane);
                        }
                    }
                }
                if (lane == 0) {
                    StreamHdr h;
                    h.live = live_w ? 1 : 0; h.slice = live_w ? c_w.slice : 0; h.owner = (live_w && c_w.owner) ? 1 : 0; h.pad = 0;
                    hdr[b] = h;
                }
            }
            // shift the pipeline
#pragma unroll
            for (int q = 0; q < NP; ++q) { sp_w[q] = sp_d[q]; gi_w[q] = pts_d.gi[q]; ld_w[q] = ld_d[q]; ld_d[q] = ld_r[q]; }
            c_w = c_d; live_w = live_d;
            pts_d = pts_r; c_d = c_r; live_d = live_r;
            kraw = kraw_new; ksx = ksx_new; have_ticket = true;
            if (it >= 1) {
                __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): the records are in LDS
                __builtin_amdgcn_s_barrier();
            }
            if (it >= 2) cur = cur == kStreamBufs - 1 ? 0 : cur + 1;
            if (it >= 0) b = b == kStreamBufs - 1 ? 0 : b + 1;
        }
        return;
    }
    __builtin_amdgcn_s_barrier();

    StreamRound<VC> R0, R1;
    if (PIPE && hdr[0].live && P.st_debug != 1) {
        const StreamBuf<V, T> B0(smem);
        stream_issue<V, T, VC>(R0, B0, data, svb, slot(0), 0, lane_offset(hdr[0].slice));
    }
    int cur = 0;                                        // record buffer of tile t
#pragma unroll 1
    for (;;) {
        const StreamHdr h = hdr[cur];
        if (h.live == 0) break;
        const int nxt = cur == kStreamBufs - 1 ? 0 : cur + 1;
        const StreamBuf<V, T> B(smem + cur * BUF), BN(smem + nxt * BUF);
        const uint32_t co = lane_offset(h.slice);       // byte offset of this lane's vector in a texel
        f32x4 acc = (f32x4)0.0f;
        if (P.st_debug == 1) {
            // experiment: consumers idle (producer-bound time)
        } else if (PIPE) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int k = r / RPS, v0 = (r % RPS) * VC;
                StreamRound<VC> &Rc = (r & 1) ? R1 : R0, &Rn = (r & 1) ? R0 : R1;
                if (r + 1 < NR) {
                    stream_issue<V, T, VC>(Rn, B, data, svb, slot((r + 1) / RPS), ((r + 1) % RPS) * VC, co);
                } else {
                    const StreamHdr hn = hdr[nxt];
                    if (hn.live) stream_issue<V, T, VC>(Rn, BN, data, svb, slot(0), 0, lane_offset(hn.slice));   // first round of the next tile
                }
                __builtin_amdgcn_sched_barrier(0);              // the next round's loads stay ahead of this round's use
                if (v0 == 0) acc = (f32x4)0.0f;
                stream_consume<V, T, VC>(acc, Rc, B, slot(k), v0);
                if (v0 + VC == V) stream_finish<V, T>(acc, P, m, B, slot(k), co);
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int k = r / RPS, v0 = (r % RPS) * VC;
                stream_issue<V, T, VC>(R0, B, data, svb, slot(k), v0, co);
                if (v0 == 0) acc = (f32x4)0.0f;
                stream_consume<V, T, VC>(acc, R0, B, slot(k), v0);
                if (v0 + VC == V) stream_finish<V, T>(acc, P, m, B, slot(k), co);
            }
        }
        // strict points (rare): redone with the strict arithmetic on direct loads
#pragma unroll 1
        for (int k = 0; k < KS && P.st_debug != 1; ++k)
            if (B.flag[slot(k)] != 0u) stream_strict_point<V, T>(P, m, B, data, slot(k), co);
        if (P.n_maps > 1 && h.owner)
#pragma unroll 1
            for (int s = 1; s < P.n_maps; ++s) {
                const MapDesc &mt = P.maps[s];
                switch (mt.vw) {
                case 4: stream_thin<V, T, 4>(mt, P, B, threadIdx.x, NC * 64); break;
                case 2: stream_thin<V, T, 2>(mt, P, B, threadIdx.x, NC * 64); break;
                default: stream_thin<V, T, 1>(mt, P, B, threadIdx.x, NC * 64); break;
                }
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): this tile's record reads are done
        __builtin_amdgcn_s_barrier();
        cur = nxt;
    }
}

template <int LG, int T, int NC, int VC, bool PIPE, int WAVES, bool PRE = false>
__global__ __launch_bounds__((NC + 1) * 64, WAVES) void fused_eval_stream_kernel(const EvalParams P)
{
    fused_eval_stream_body<LG, T, NC, VC, PIPE, PRE>(P);
}

__global__ void stream_zero_tickets_kernel(unsigned int *t)
{
    if (threadIdx.x < 8) t[threadIdx.x * kTicketStride] = 0u;
}

#ifdef D3F_EXPERIMENTS
// tuning sessions only: ticket counters inside the library (the product takes them from the caller's workspace)
__device__ unsigned int g_exp_tickets[8 * kTicketStride];
unsigned int *stream_exp_tickets()
{
    void *p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(g_exp_tickets)) == hipSuccess ? static_cast<unsigned int *>(p) : nullptr;
}
void *stream_exp_scratch(int64_t bytes)
{
    static void *buf = nullptr;
    static int64_t cap = 0;
    if (bytes > cap) {
        if (buf) (void)hipFree(buf);
        buf = nullptr; cap = 0;
        if (hipMalloc(&buf, (size_t)bytes) == hipSuccess) cap = bytes;
    }
    return buf;
}
#endif

int stream_lds_bytes(int T, int V) { return kStreamBufs * ((stream_buf_bytes(T, V) + 15) / 16 * 16) + kStreamBufs * 16 + V * 48; }

hipError_t launch_fused_stream(const EvalParams &P, hipStream_t stream)
{
    const int64_t units = (int64_t)P.sl_chunks * P.sl_slices;
    // tickets: a fixed grid of persistent workgroups (st_grid per XCD) and eight zeroed counters; static: one workgroup
    // per st_R tiles of a unit
    const int64_t wgs = P.st_tickets ? (int64_t)8 * P.st_grid : (units + 7) / 8 * 8 * P.sl_unit;
    if (P.st_tickets) hipLaunchKernelGGL(stream_zero_tickets_kernel, dim3(1), dim3(64), 0, stream, P.st_tickets);
    if (P.st_rec) {
        const dim3 gp((unsigned)((P.sl_tiles + 3) / 4));
        if (P.tile_pts == 12) hipLaunchKernelGGL((stream_prepass_kernel<4, 12>), gp, dim3(256), 0, stream, P);
        else if (P.tile_pts == 24) hipLaunchKernelGGL((stream_prepass_kernel<4, 24>), gp, dim3(256), 0, stream, P);
        else return hipErrorInvalidValue;
    }
    const size_t lds = (size_t)stream_lds_bytes(P.tile_pts, P.V) + (size_t)P.lds_pad;
    const dim3 grid((unsigned)wgs);
#define D3F_STREAM(LG_, T_, NC_, VC_, PIPE_, W_)                                                                              \
    hipLaunchKernelGGL((fused_eval_stream_kernel<LG_, T_, NC_, VC_, PIPE_, W_>), grid, dim3((NC_ + 1) * 64), lds, stream, P)
    const int key = P.sl_lg * 1000 + P.tile_pts * 10 + P.st_variant + (P.st_rec ? 100000 : 0);
#define D3F_STREAM_PRE(LG_, T_, NC_, VC_, PIPE_, W_)                                                                          \
    hipLaunchKernelGGL((fused_eval_stream_kernel<LG_, T_, NC_, VC_, PIPE_, W_, true>), grid, dim3((NC_ + 1) * 64), lds, stream, P)
    switch (key) {
    // geometry pre-pass variants (experiment)
    case 100000 + 5000 + 120 + 1: D3F_STREAM_PRE(5, 12, 3, 2, false, 7); break;
    case 100000 + 5000 + 120 + 2: D3F_STREAM_PRE(5, 12, 3, 4, false, 5); break;
    case 100000 + 4000 + 120 + 1: D3F_STREAM_PRE(4, 12, 3, 2, false, 7); break;
    case 100000 + 4000 + 240 + 1: D3F_STREAM_PRE(4, 24, 3, 2, false, 7); break;
    case 100000 + 4000 + 240 + 2: D3F_STREAM_PRE(4, 24, 3, 4, false, 5); break;
    case 100000 + 4000 + 240 + 0: D3F_STREAM_PRE(4, 24, 3, 2, true, 3); break;
    case 100000 + 3000 + 240 + 1: D3F_STREAM_PRE(3, 24, 3, 2, false, 7); break;
    case 100000 + 3000 + 240 + 2: D3F_STREAM_PRE(3, 24, 3, 4, false, 5); break;
    // 512-byte slices.  variant 0 / 3: pipelined rounds at 3 / 4 waves per SIMD; 1: one register set, two views per round;
    // 2: one set, four views per round
    case 5000 + 120 + 0: D3F_STREAM(5, 12, 3, 2, true, 3); break;
    case 5000 + 120 + 3: D3F_STREAM(5, 12, 3, 2, true, 4); break;
    case 5000 + 120 + 1: D3F_STREAM(5, 12, 3, 2, false, 7); break;
    case 5000 + 120 + 2: D3F_STREAM(5, 12, 3, 4, false, 5); break;
    case 5000 + 240 + 0: D3F_STREAM(5, 24, 3, 2, true, 3); break;
    case 5000 + 240 + 1: D3F_STREAM(5, 24, 3, 2, false, 6); break;
    case 5000 + 240 + 2: D3F_STREAM(5, 24, 3, 4, false, 5); break;
    case 5000 + 160 + 0: D3F_STREAM(5, 16, 4, 2, true, 3); break;
    case 5000 + 160 + 1: D3F_STREAM(5, 16, 4, 2, false, 5); break;
    case 5000 + 160 + 2: D3F_STREAM(5, 16, 4, 4, false, 5); break;
    // 256-byte slices
    case 4000 + 120 + 0: D3F_STREAM(4, 12, 3, 2, true, 3); break;
    case 4000 + 120 + 1: D3F_STREAM(4, 12, 3, 2, false, 7); break;
    case 4000 + 240 + 0: D3F_STREAM(4, 24, 3, 2, true, 3); break;
    case 4000 + 240 + 1: D3F_STREAM(4, 24, 3, 2, false, 7); break;
    default: return hipErrorInvalidValue;
    }
#undef D3F_STREAM
#undef D3F_STREAM_PRE
    return hipGetLastError();
}

}  // namespace d3f
