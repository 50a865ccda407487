// fuse_window.hip -- the LDS-WINDOW kernel of the fused field query (gfx950) and the device-side gate that chooses between it and
// the cell runs for a cloud.  Patch-resolution wide maps (the reference's dino_feats, fusion.py:694-697).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"
#include "fuse_common.h"

namespace d3f {

// ---- texel windows in LDS for small wide maps (round 2) ---------------------------------------------------------------
// Patch-resolution feature maps (the reference's dino_feats, fusion.py:694-697) sit in the caches, and the direct gather
// of them is bound by the vector-L1 path (64 B/clk per CU, every (point, view) re-reading four whole texels) and by the
// VALU work around every load.  The LDS reads 256 B/clk.  Here a workgroup takes a compact set of tile_pts points (a
// power-of-two brick of a lattice, or that many consecutive points of the Hilbert order of a cloud):
//   1. the eight corners of the set's bounding box are projected into every view (one lane per (view, corner)); the
//      texel rectangle they span is the view's window (a projective map sends the box into the convex hull of its
//      projected corners);
//   2. phase A, one lane per (point, view) with the views of a point in adjacent lanes: the per-point sums over the
//      views are rebuilt in view order with wave shuffles (no second pass); it writes a 32-byte window record per
//      pair: LDS offset of the nw corner, row pitch, weight and validity of the view, the four bilinear weights (the
//      16-byte view records of the other kernels only when thin maps ride along), and per point the refined reciprocal
//      of the shared-divisor division.
//      Every corner is CHECKED against the window -- a pair whose corners are not all inside the map and the window
//      (rounding at the rim, an overflowing pool, a box behind the camera, the image border) is marked direct and
//      gathered from global memory as before, so the result never depends on step 1 being right;
//   3. per channel slice of 128*U channels: the windows' texel slices are copied global -> LDS by the DMA path
//      (global_load_lds_dwordx4: no registers, no ds_write; the source offset of every pool slot is tabulated once
//      per workgroup), then 16 lanes x two vectors per point (32 x U with wider slices) read records and corners with
//      ds_read_b128 and run the arithmetic of gather_map (same operands, same order: bit-identical results).  Invalid
//      pairs point at an all-zero texel with zero weights (exact skip); strict points take the strict
//      form on global loads.
// XCD k takes the k-th contiguous eighth of the bricks (window copies are then L2 hits left by the neighbours).
// Measured history, counters and the variants that lost: DESIGN.md 5.5.
// Slots of a clipped brick (or of the last, short tile of a cloud) repeat a neighbouring point: same inputs, same
// outputs, written twice.
constexpr float kWinDirectMark = 2.0f;       // WinRec::valid of a pair gathered from global memory (a view's validity is 0 or 1)
struct __attribute__((aligned(16))) WinRec {
    uint32_t nw;        // LDS byte offset of the nw corner's slice; ne = nw + slice bytes
    uint32_t sw;        // LDS byte offset of the sw corner's slice (one window row further); se = sw + slice bytes
    float wgt;          // a direct pair of a fast point: fold_scale(wgt, cnt); every pair of a strict point: ViewRec::wgt
    float valid;        // ViewRec::valid; kWinDirectMark: a DIRECT pair (corners outside the window / the pool) of a non-strict
                        // point -- nw and sw then point at the zero slices, so the pipelined loop may read them harmlessly
    float w[4];         // FOLDED bilinear weights nw, ne, sw, se (fuse_common.h); a direct pair (and every pair of a strict
                        // point) keeps gx, gy here instead -- the 16-byte view records exist only when thin maps ride along
};
struct WinView {
    int xmin, ymin, bw, bh;     // texel rectangle
    int base;                   // first pool slot
    int ok;                     // 0: no window for this view (every pair of it goes direct)
};
constexpr int kWinMaxViews = 8;
constexpr int kWinMaxTexels = 320;       // pool slots (host: win_pool_texels <= this)
constexpr uint32_t kWinStrict = 1u, kWinHasDirect = 2u;

// acc[NV] += corner vectors * w, one raw vector (four channels: 16 bytes of fp32, 8 of fp16) per accumulator
template <int NV, bool HALF>
__device__ __forceinline__ void win_accumulate(f32x4 (&acc)[NV], const typename WinRaw<HALF>::T (&c)[NV], float w)
{
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if constexpr (HALF) fma_mix4(acc[u], c[u], w);
        else acc[u] = v_fma<f32x4>(c[u], w, acc[u]);
    }
}

// all views of one point from the pool: VC views' corner reads in flight together
// NV accumulator vectors per lane; raw vectors VS bytes apart inside a slice of SBB bytes (the ne / se corners are one slice further)
template <int NV, int VC, int VS, int SBB, bool HALF = false>
__device__ __forceinline__ void window_point(f32x4 (&acc)[NV], const unsigned char *smem, const WinRec *wr_p, int V,
                                             uint32_t lane_off)
{
    using RT = typename WinRaw<HALF>::T;
    constexpr int NR = NV;
    int v0 = 0;
    for (; v0 + VC <= V; v0 += VC) {
        WinRec wr[VC];
#pragma unroll
        for (int q = 0; q < VC; ++q) {
            // two ds_read_b128 (4 LDS cycles each): left alone the compiler narrows the first to a ds_read_b96, which
            // the LDS serves in 8 cycles (MI355X_MICROARCH.md, LDS table) -- the asm keeps all four dwords "used"
            const f32x4 *r = reinterpret_cast<const f32x4 *>(wr_p + v0 + q);
            f32x4 h0 = r[0], h1 = r[1];
            asm volatile("" : "+v"(h0));
            wr[q].nw = __float_as_uint(h0.x); wr[q].sw = __float_as_uint(h0.y); wr[q].wgt = h0.z; wr[q].valid = h0.w;
            wr[q].w[0] = h1.x; wr[q].w[1] = h1.y; wr[q].w[2] = h1.z; wr[q].w[3] = h1.w;
        }
        RT a[VC][NR], b[VC][NR], d[VC][NR], e[VC][NR];
#pragma unroll
        for (int q = 0; q < VC; ++q) {
            const unsigned char *nw = smem + (wr[q].nw + lane_off);
            const unsigned char *sw = smem + (wr[q].sw + lane_off);
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                a[q][u] = *reinterpret_cast<const RT *>(nw + u * VS);
                b[q][u] = *reinterpret_cast<const RT *>(nw + SBB + u * VS);
                d[q][u] = *reinterpret_cast<const RT *>(sw + u * VS);
                e[q][u] = *reinterpret_cast<const RT *>(sw + SBB + u * VS);
            }
        }
#pragma unroll
        for (int q = 0; q < VC; ++q) {
            win_accumulate<NV, HALF>(acc, a[q], wr[q].w[0]);        // folded weights (fuse_common.h)
            win_accumulate<NV, HALF>(acc, b[q], wr[q].w[1]);
            win_accumulate<NV, HALF>(acc, d[q], wr[q].w[2]);
            win_accumulate<NV, HALF>(acc, e[q], wr[q].w[3]);
        }
    }
    if constexpr (VC > 1) {
        for (; v0 < V; ++v0) {
            const WinRec wr = wr_p[v0];
            const unsigned char *nw = smem + (wr.nw + lane_off);
            const unsigned char *sw = smem + (wr.sw + lane_off);
            RT a[NR], b[NR], d[NR], e[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                a[u] = *reinterpret_cast<const RT *>(nw + u * VS); b[u] = *reinterpret_cast<const RT *>(nw + SBB + u * VS);
                d[u] = *reinterpret_cast<const RT *>(sw + u * VS); e[u] = *reinterpret_cast<const RT *>(sw + SBB + u * VS);
            }
            win_accumulate<NV, HALF>(acc, a, wr.w[0]);
            win_accumulate<NV, HALF>(acc, b, wr.w[1]);
            win_accumulate<NV, HALF>(acc, d, wr.w[2]);
            win_accumulate<NV, HALF>(acc, e, wr.w[3]);
        }
    }
}

// The same for KI points of one lane group with the view count a compile-time constant, as ONE software pipeline over the
// KI * VF (point, view) steps (round 4).  The loop above costs two dependent LDS round trips per (point, view) -- record, then
// the corners the record points at -- with nothing else in flight: the counters of round 3's kernel showed waves waiting
// 57 % of their time with the LDS array 44 % and the VALU 39 % busy.  Here step j's sixteen fma run while the eight corner
// reads of step j + 1 and the record of step j + 2 are in flight (records: one ds_read_b64 {nw, sw} + one ds_read_b128
// {weights}; their addresses are immediates off one base register), so a wave always has >= 10 LDS reads behind its
// arithmetic.  Same operands in the same order as window_point: bit-identical.
//   rec0      LDS byte offset of the first point's first record;  KSTRIDE bytes between the lane group's consecutive points
//   done(k, acc)  called once per point, after its last view
// (The what-if builds of the tuning sessions -- parts of this kernel compiled out, only times read -- are a patch:
// scripts/notebook/patches/r6_whatif_macros.patch.)

template <int NV, bool HALF>
struct WinPipe {            // registers of the pipeline; every index below is a compile-time constant
    uint2 off[3];
    f32x4 wt[3];
    typename WinRaw<HALF>::T c[2][4][NV];
    f32x4 acc[NV];
};

template <int J, int NV, int VF, int KI, int VS, int SBB, int KSTRIDE, bool HALF>
__device__ __forceinline__ void win_pipe_record(WinPipe<NV, HALF> &st, const unsigned char *smem, uint32_t rec0)
{
    const unsigned char *r = smem + rec0 + (uint32_t)((J / VF) * KSTRIDE + (J % VF) * (int)sizeof(WinRec));
    st.off[J % 3] = *reinterpret_cast<const uint2 *>(r);
    st.wt[J % 3] = *reinterpret_cast<const f32x4 *>(r + 16);
}

template <int J, int NV, int VS, int SBB, bool HALF>
__device__ __forceinline__ void win_pipe_corners(WinPipe<NV, HALF> &st, const unsigned char *smem, uint32_t lane_off)
{
    using RT = typename WinRaw<HALF>::T;
    constexpr int NR = NV;
    const unsigned char *nw = smem + (st.off[J % 3].x + lane_off);
    const unsigned char *sw = smem + (st.off[J % 3].y + lane_off);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        st.c[J % 2][0][u] = *reinterpret_cast<const RT *>(nw + u * VS);
        st.c[J % 2][1][u] = *reinterpret_cast<const RT *>(nw + SBB + u * VS);
        st.c[J % 2][2][u] = *reinterpret_cast<const RT *>(sw + u * VS);
        st.c[J % 2][3][u] = *reinterpret_cast<const RT *>(sw + SBB + u * VS);
    }
}

template <int J, int NV, int VF, int KI, int VS, int SBB, int KSTRIDE, bool HALF, typename DONE>
__device__ __forceinline__ void win_pipe_step(WinPipe<NV, HALF> &st, const unsigned char *smem, uint32_t rec0, uint32_t lane_off, DONE &done)
{
    using VT = f32x4;
    constexpr int NS = KI * VF;
    if constexpr (J + 2 < NS) win_pipe_record<J + 2, NV, VF, KI, VS, SBB, KSTRIDE, HALF>(st, smem, rec0);
    if constexpr (J + 1 < NS) win_pipe_corners<J + 1, NV, VS, SBB, HALF>(st, smem, lane_off);
    win_accumulate<NV, HALF>(st.acc, st.c[J % 2][0], st.wt[J % 3].x);        // folded weights (fuse_common.h): nw, ne, sw, se
    win_accumulate<NV, HALF>(st.acc, st.c[J % 2][1], st.wt[J % 3].y);
    win_accumulate<NV, HALF>(st.acc, st.c[J % 2][2], st.wt[J % 3].z);
    win_accumulate<NV, HALF>(st.acc, st.c[J % 2][3], st.wt[J % 3].w);
    if constexpr (J % VF == VF - 1) {
        done(J / VF, st.acc);
#pragma unroll
        for (int u = 0; u < NV; ++u) st.acc[u] = (VT)0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the steps in program order (the pipeline IS the schedule)
    if constexpr (J + 1 < NS) win_pipe_step<J + 1, NV, VF, KI, VS, SBB, KSTRIDE, HALF>(st, smem, rec0, lane_off, done);
}

template <int NV, int VF, int KI, int VS, int SBB, int KSTRIDE, bool HALF, typename DONE>
__device__ __forceinline__ void window_points_pipelined(const unsigned char *smem, uint32_t rec0, uint32_t lane_off, DONE done)
{
    WinPipe<NV, HALF> st;
    win_pipe_record<0, NV, VF, KI, VS, SBB, KSTRIDE, HALF>(st, smem, rec0);
    if constexpr (KI * VF > 1) win_pipe_record<1, NV, VF, KI, VS, SBB, KSTRIDE, HALF>(st, smem, rec0);
    win_pipe_corners<0, NV, VS, SBB, HALF>(st, smem, lane_off);
#pragma unroll
    for (int u = 0; u < NV; ++u) st.acc[u] = (f32x4)0.0f;
    win_pipe_step<0, NV, VF, KI, VS, SBB, KSTRIDE, HALF>(st, smem, rec0, lane_off, done);
}

// LPP lanes per point in phase B: 32 (U vectors per lane, 512 bytes apart, slices of 512*U bytes) or 16 (U == 1: two
// vectors per lane 256 bytes apart inside the 512-byte slice, four points per wave instruction)
// VFIX: 4 / 8 = the view count as a compile-time constant (the host launches that variant when V matches): the point loop
// runs window_points_pipelined; 0: any V <= 8, view loop of window_point
// SPARSE (round 5): the pool holds only the texels the tile's valid pairs TOUCH, not the whole rectangles.  Phase A marks the four
// corners of every pair in a bitmap over the views' rectangles (LDS atomics), one wave ranks the set bits, and a pair's nw / sw
// slots are the ranks of its bits -- ne and se are the next bits of the same rows, hence the next slots, so the records and the
// point loop are unchanged.  A 64-point tile of a cloud's Hilbert order touches 47 texels where its rectangles hold 74 (C2-patch,
// scripts/notebook/sim_cloud_tiles.py): an 80-slot pool then overflows on 0.5 % of the tiles instead of 19 %.  The price: the copy of slice
// 0 starts after phase A instead of underneath it.
constexpr int kWinMaxBits = 2048;            // bitmap bits over all views' rectangles (rows that do not fit are left out)
// HALF: map 0 is stored in fp16 (D3F_DTYPE_F16): 256-byte slices of 128 channels, one 16-byte raw vector per lane (see fma_mix8)
template <int U, int VC, int NT, int LPP, int VFIX, bool SPARSE, bool HALF>
__device__ __forceinline__ void fused_eval_window_body(const EvalParams &P)
{
    using VT = f32x4;
    static_assert(LPP == 32 || (LPP == 16 && U == 1), "16 lanes per point only with 512-byte slices");
    static_assert(!HALF || (LPP == 16 && U == 1), "fp16-stored maps: 16 lanes x two 4-channel vectors per point");
    constexpr int ES = HALF ? 2 : 4;                   // bytes per stored channel of map 0
#ifdef D3F_EXPERIMENTS
    // phase stamps (D3F_EXP_STAMPS=1): lane 0 of wave 0 of every 64th workgroup writes s_memtime at the phase boundaries
    int stamp_k = 0;
    const bool stamping = P.exp_stamps != nullptr && (blockIdx.x & 63u) == 0u && threadIdx.x == 0 && (blockIdx.x >> 6) < 65536u;
#define D3F_STAMP() do { if (stamping && stamp_k < 31) P.exp_stamps[(size_t)(blockIdx.x >> 6) * 32 + 1 + stamp_k++] = __builtin_readcyclecounter(); } while (0)
#else
#define D3F_STAMP() do { } while (0)
#endif
    D3F_STAMP();                                    // 0: entry
    constexpr int NV = U * (32 / LPP);                 // vectors per lane
    constexpr int RB = HALF ? 8 : 16;                  // bytes of a lane's raw vector (four channels as stored)
    constexpr int VS = RB * LPP;                       // bytes between a lane's raw vectors inside a slice
    extern __shared__ __align__(16) unsigned char smem[];
    const int V = P.V;
    const int TP = P.tile_pts;
    const bool has_rec = P.n_maps > 1;                                       // thin maps read the view records
    // window records: V consecutive 32-byte records per point, points (V*32 + 16) bytes apart -- with V = 8 a stride of 256
    // bytes puts the records of the two points a ds_read lane group serves on the same banks (round 3's c4_patch counters:
    // SQ_LDS_BANK_CONFLICT = 16 % of the LDS cycles, exactly the two-way conflict of every record read)
    const uint32_t pstride = (uint32_t)(VFIX > 0 ? VFIX : V) * 32u + 16u;
    auto wrec_at = [&](int p) -> WinRec * { return reinterpret_cast<WinRec *>(smem + (uint32_t)p * pstride); };
    ViewRec *rec = reinterpret_cast<ViewRec *>(smem + (size_t)TP * pstride);   // [TP*V] if has_rec
    float *cnt_s = reinterpret_cast<float *>(rec + (has_rec ? (size_t)TP * V : 0));   // [TP]
    uint32_t *flag_s = reinterpret_cast<uint32_t *>(cnt_s + TP);             // [TP]
    uint32_t *idx_s = flag_s + TP;                                           // [TP]
    float *aux_s = reinterpret_cast<float *>(idx_s + TP);                    // [TP][2]: refined 1/(cnt + 1e-6), cnt + 1e-6
    float *krt = aux_s + 2 * TP;                                             // [V*12]
    constexpr uint32_t SB = (HALF ? 256u : 512u) * U;                        // bytes of one texel slice (128 * U channels as stored)
    constexpr uint32_t OSB = 512u * U;                                       // ... of the same channels in an output row (fp32)
    constexpr int OVS = 16 * LPP;                                            // bytes between a lane's accumulator vectors in an output row
    const uint32_t zero_off = (uint32_t)P.win_pool_offset;                   // two all-zero slices, then the pool
    const uint32_t pool_off = zero_off + 2u * SB;
    __shared__ float cpt_s[8][3];
    __shared__ float red_s[NT / 64][6];
    __shared__ WinView win_s[kWinMaxViews];
    __shared__ int total_s;
    __shared__ uint32_t texsrc_s[kWinMaxTexels];     // byte offset (from the map's base) of every pool slot's texel
    __shared__ uint32_t bits_s[SPARSE ? kWinMaxBits / 32 : 1];          // SPARSE: touched texels of the views' rectangles
    __shared__ uint32_t wpre_s[SPARSE ? kWinMaxBits / 32 : 1];          // ... set bits in front of every word

    const bool walk = P.walk_nx > 0;
    const MapDesc &m0 = P.maps[0];
    // the set of points: a brick of the lattice (bricks numbered z fastest) or tile_pts consecutive points
    // (XCD k = blockIdx % 8 takes the k-th contiguous eighth of the bricks, z fastest: the ~128 workgroups an XCD has in
    // flight are neighbours, so the window copies of one are L2 hits left behind by the others)
    const int lbz = __ffs(P.walk_tz) - 1, lby = __ffs(P.walk_ty) - 1;
    int ox = 0, oy = 0, oz = 0;
    if (walk) {
        const uint32_t nbz = (uint32_t)((P.walk_nz + P.walk_tz - 1) / P.walk_tz), nby = (uint32_t)((P.walk_ny + P.walk_ty - 1) / P.walk_ty);
        const uint32_t b = (uint32_t)xcd_tile((int64_t)blockIdx.x, (int64_t)gridDim.x);
        const uint32_t bxy = b / nbz;
        oz = (int)(b - bxy * nbz) * P.walk_tz;
        const uint32_t bx = bxy / nby;
        oy = (int)(bxy - bx * nby) * P.walk_ty;
        ox = (int)bx * P.walk_tx;
    }
    const int bsx = min(P.walk_tx, P.walk_nx - ox), bsy = min(P.walk_ty, P.walk_ny - oy), bsz = min(P.walk_tz, P.walk_nz - oz);
    // a cloud: XCD k takes the k-th contiguous eighth of the tiles of the Hilbert order, like the bricks of a lattice (experiments
    // builds: D3F_EXP_WINDOW_RR=1 = consecutive tiles round-robin over the XCDs, the form of rounds 2-4)
    const int64_t tile_base = ((P.flags & kFlagXcdRemap) ? (int64_t)blockIdx.x : xcd_tile((int64_t)blockIdx.x, (int64_t)gridDim.x)) * TP;
    const int tile_n = walk ? TP : (int)min((int64_t)TP, P.n - tile_base);
    auto slot_point = [&](int p) -> int64_t {
        if (walk) {
            const int lz = min(p & (P.walk_tz - 1), bsz - 1), ly = min((p >> lbz) & (P.walk_ty - 1), bsy - 1);
            const int lx = min(p >> (lbz + lby), bsx - 1);
            return ((int64_t)(ox + lx) * P.walk_ny + (oy + ly)) * P.walk_nz + (oz + lz);
        }
        const int64_t q = tile_base + min(p, tile_n - 1);
        return P.order ? min((int64_t)P.order[q], P.n - 1) : q;
    };
    // coordinates of slot p; on a d3f_grid they come straight from the axis arrays at the brick coordinates (fetch_point
    // would divide the flat index apart again: two 64-bit divisions per lane)
    auto slot_coords = [&](int p, int64_t i, float &px, float &py, float &pz) {
        if (walk && P.grid_x) {
            const int lz = min(p & (P.walk_tz - 1), bsz - 1), ly = min((p >> lbz) & (P.walk_ty - 1), bsy - 1);
            const int lx = min(p >> (lbz + lby), bsx - 1);
            px = P.grid_x[ox + lx]; py = P.grid_y[oy + ly]; pz = P.grid_z[oz + lz];
        } else {
            fetch_point(P, i, px, py, pz);
        }
    };
    const float mu = P.mu;
    const float Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);

    // ---- 0. this lane's query point of phase A's first pass, requested NOW: its load (a first touch of the point array: an HBM
    //         round trip) returns underneath the set-up below instead of in front of the projection (phase stamps, round 4:
    //         phase A was 10.5 k of a workgroup's 60 k cycles, ~4.5 k of them this load)
    const int pa_log2 = V <= 1 ? 0 : (V <= 2 ? 1 : (V <= 4 ? 2 : 3));
    float pre_x = 0.0f, pre_y = 0.0f, pre_z = 0.0f;
    if ((int)threadIdx.x < (TP << pa_log2) && (int)(threadIdx.x & ((1u << pa_log2) - 1u)) < V) {
        const int p = (int)threadIdx.x >> pa_log2;
        slot_coords(p, slot_point(p), pre_x, pre_y, pre_z);
    }
    // ---- 1. KRt, the zero slices, the eight corner points of the set's bounding box ----
    compute_krt(P.K, P.pose, V, krt, NT);
    for (uint32_t t = threadIdx.x; t < 2u * SB / 4u; t += NT) reinterpret_cast<uint32_t *>(smem + zero_off)[t] = 0u;
    if constexpr (SPARSE)
        if (threadIdx.x < kWinMaxBits / 32) bits_s[threadIdx.x] = 0u;
    if (walk) {
        if (threadIdx.x < 8) {
            const int c = threadIdx.x;
            const int lx = (c & 1) ? bsx - 1 : 0, ly = (c & 2) ? bsy - 1 : 0, lz = (c & 4) ? bsz - 1 : 0;
            const int64_t i = ((int64_t)(ox + lx) * P.walk_ny + (oy + ly)) * P.walk_nz + (oz + lz);
            float px, py, pz;
            fetch_point(P, i, px, py, pz);
            cpt_s[c][0] = px; cpt_s[c][1] = py; cpt_s[c][2] = pz;
        }
    } else {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int p = threadIdx.x; p < tile_n; p += NT) {
            float q[3];
            fetch_point(P, slot_point(p), q[0], q[1], q[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], q[k]); hi[k] = fmaxf(hi[k], q[k]); }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
            }
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < 3; ++k) { red_s[threadIdx.x >> 6][k] = lo[k]; red_s[threadIdx.x >> 6][3 + k] = hi[k]; }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int c = threadIdx.x;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float a = red_s[0][k], b = red_s[0][3 + k];
                for (int w = 1; w < NT / 64; ++w) { a = fminf(a, red_s[w][k]); b = fmaxf(b, red_s[w][3 + k]); }
                cpt_s[c][k] = ((c >> k) & 1) ? b : a;      // a NaN coordinate is dropped by fmin/fmax: such a point is strict
            }
        }
    }
    __syncthreads();
    D3F_STAMP();                                    // 1: KRt, zero slices, box corners
    // ---- 2. one window per view: wave 0, lane = view * 8 + corner ----
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, v = lane >> 3, c = lane & 7;
        const bool act = v < V && V <= kWinMaxViews;
        float xl = 0.0f, xh = 0.0f, yl = 0.0f, yh = 0.0f;
        int ok = 0;
        if (act) {
            const Proj pr = project_point(krt + v * 12, cpt_s[c][0], cpt_s[c][1], cpt_s[c][2], Wm1, Hm1);
            const float ix = unnormalize(pr.gx, m0.fw), iy = unnormalize(pr.gy, m0.fh);
            ok = (pr.ok && pr.zc > 1e-4f && isfinite(ix) && isfinite(iy)) ? 1 : 0;      // the whole box in front of the camera
            xl = xh = ix; yl = yh = iy;
        }
        // min / max / and over the view's eight corner lanes, on the VALU (DPP: lane ^ 1, lane ^ 2 inside the quad, then the
        // mirror of the 8-lane half row pairs the two quads) -- as ds_bpermute shuffles these were 27 LDS round trips of the
        // one wave every other wave of the workgroup is waiting for
#define D3F_WIN_RED(CTRL)                                                                                   \
        xl = fminf(xl, dpp_f<CTRL>(xl)); xh = fmaxf(xh, dpp_f<CTRL>(xh));                                   \
        yl = fminf(yl, dpp_f<CTRL>(yl)); yh = fmaxf(yh, dpp_f<CTRL>(yh));                                   \
        ok &= dpp_i<CTRL>(ok);
        D3F_WIN_RED(0xB1) D3F_WIN_RED(0x4E) D3F_WIN_RED(0x141)
#undef D3F_WIN_RED
        WinView w = {0, 0, 1, 1, 0, 0};
        int ntex = 0;
        if (ok) {
            const float fwm1 = (float)(m0.fw - 1), fhm1 = (float)(m0.fh - 1);
            const int x0 = (int)fminf(fmaxf(floorf(xl - 1e-3f), 0.0f), fwm1), x1 = (int)fminf(fmaxf(floorf(xh + 1e-3f) + 1.0f, 0.0f), fwm1);
            const int y0 = (int)fminf(fmaxf(floorf(yl - 1e-3f), 0.0f), fhm1), y1 = (int)fminf(fmaxf(floorf(yh + 1e-3f) + 1.0f, 0.0f), fhm1);
            w.xmin = x0; w.ymin = y0; w.bw = x1 - x0 + 1; w.bh = y1 - y0 + 1;
            ntex = w.bw * w.bh;
        }
        // pool slots in view order; a window that does not fit keeps the rows that do (pairs in the other rows go
        // direct, like every pair of a view left without a window)
        int run = 0;
        for (int vv = 0; vv < V && vv < kWinMaxViews; ++vv) {
            const int nv = __builtin_amdgcn_readlane(ntex, vv * 8), bwv = __builtin_amdgcn_readlane(w.bw, vv * 8);
            int rows = nv > 0 ? min(nv, (SPARSE ? kWinMaxBits : P.win_pool_texels) - run) / bwv : 0;      // SPARSE: `base` counts bitmap bits
            if (rows < 2) rows = 0;                              // a bilinear footprint needs two rows
            if (vv == v) { w.base = run; w.ok = rows > 0 ? 1 : 0; w.bh = rows > 0 ? rows : w.bh; }
            run += rows * bwv;
        }
        if (act && c == 0) win_s[v] = w;
        if (lane == 0) total_s = run;
    }
    __syncthreads();

    D3F_STAMP();                                    // 2: windows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // where every pool slot's texel lives in the map: one lane per slot, once per workgroup (the view search and the
    // division by the window width cost ~100 VALU instructions; done per copy instruction and slice they were a third
    // of the kernel's VALU work)
    // (texel of position `t` of the views' rectangles, row-major per view in view order: a pool slot, or with SPARSE a bitmap bit)
    auto rect_texel = [&](int t) -> uint32_t {
        int v = 0;
        for (int vv = 1; vv < V && vv < kWinMaxViews; ++vv)
            if (win_s[vv].ok && t >= win_s[vv].base) v = vv;          // bases ascend over the views that have a window
        const WinView w = win_s[v];
        const int local = t - w.base;
        // local / bw for 0 <= local < 2048, 1 <= bw <= 2048 through the float reciprocal: (local + 0.5) / bw is at least 1/(2 bw)
        // away from every integer, far more than the rounding of rcp and the product -- exact, at a tenth of the integer division
        const int y = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)w.bw)), x = local - y * w.bw;
        return (uint32_t)(((int64_t)v * m0.sv + (int64_t)(w.ymin + y) * m0.sy + (int64_t)(w.xmin + x) * m0.sx) * ES);
    };
    if constexpr (!SPARSE) {
        for (int t = threadIdx.x; t < total_s; t += NT) texsrc_s[t] = rect_texel(t);
        __syncthreads();
    }
    // copy of slice `sl` of every window into the pool: 512-byte granules, two per wave instruction (fp16 storage: 256-byte
    // granules, four per wave instruction); a wave instruction fills 1 KiB of consecutive pool slots
    auto stage = [&](int sl) {
        constexpr int GL = HALF ? 16 : 32, GPW = 64 / GL;        // lanes per granule, granules per wave instruction
        const int total = total_s * U;              // granules
        const int h = lane / GL, l = lane % GL;
        const char *data = reinterpret_cast<const char *>(m0.data) + (size_t)sl * SB + (size_t)l * 16;
        for (int g2 = wave; g2 * GPW < total; g2 += NT / 64) {
            const int hk = min(g2 * GPW + h, total - 1);
            const int t = hk / U, part = hk - t * U;
            const char *src = data + texsrc_s[t] + (uint32_t)part * (uint32_t)(SB / U);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(smem + pool_off + (size_t)g2 * 1024),
                                             16, 0, 0);
        }
    };
    if constexpr (!SPARSE) stage(0);
    D3F_STAMP();                                    // 3: slot table, DMA of slice 0 issued

    // ---- 3. phase A: lane = (point, view), the views of a point adjacent ----
    {
        const bool finite_maps = maps_are_finite(P);
        const int vp_log2 = pa_log2;
        const int VP = 1 << vp_log2;
        const int base = lane & ~(VP - 1);
        for (int idx = threadIdx.x; idx < TP * VP; idx += NT) {
            const int p = idx >> vp_log2, v = idx & (VP - 1);
            const int64_t i = slot_point(p);
            float dv = 0.0f, valid = 0.0f, gx = 0.0f, gy = 0.0f;
            uint32_t st = 0u;
            WinRec wr;
            wr.nw = zero_off; wr.sw = zero_off; wr.wgt = 0.0f; wr.valid = 0.0f;
            wr.w[0] = wr.w[1] = wr.w[2] = wr.w[3] = 0.0f;
            if (v < V) {
                float px = pre_x, py = pre_y, pz = pre_z;
                if (idx >= NT) slot_coords(p, i, px, py, pz);                   // later passes (more than NT pairs) load here
                float wgt;
                const ViewOut o = eval_view<0>(P.depth, P.H, P.W, krt + v * 12, v, px, py, pz, Wm1, Hm1, mu, wgt);
                if (has_rec) {
                    ViewRec r;
                    r.gx = o.gx; r.gy = o.gy; r.wgt = wgt; r.valid = o.valid;
                    rec[p * V + v] = r;
                }
                dv = o.dist * o.valid;                                          // fusion.py:364 (product only)
                valid = o.valid; gx = o.gx; gy = o.gy;
                if (!(isfinite(o.gx) && isfinite(o.gy) && isfinite(wgt))) st |= kWinStrict;
                wr.wgt = wgt; wr.valid = o.valid;
                if (o.valid != 0.0f) {
                    // the corner set-up of corner_setup(), in texel coordinates
                    const float ix = unnormalize(o.gx, m0.fw), iy = unnormalize(o.gy, m0.fh);
                    const float x0 = floorf(ix), y0 = floorf(iy);
                    const float tx = ix - x0, ty = iy - y0;
                    const float ex = 1.0f - tx, sy = 1.0f - ty;
                    const WinView w = win_s[v];
                    // all four corners inside the map (x0 in [0, fw-2]) and inside the window?
                    const bool inmap = x0 >= 0.0f && x0 <= (float)(m0.fw - 2) && y0 >= 0.0f && y0 <= (float)(m0.fh - 2);
                    const int ax = (int)fminf(fmaxf(x0, 0.0f), (float)m0.fw) - w.xmin, ay = (int)fminf(fmaxf(y0, 0.0f), (float)m0.fh) - w.ymin;
                    const bool inside = inmap && w.ok && ax >= 0 && ax + 1 < w.bw && ay >= 0 && ay + 1 < w.bh;
                    if (inside) {
                        if constexpr (SPARSE) {
                            // the four corners' bits: nw and ne are neighbours in a row of the rectangle, sw and se in the next one
                            const uint32_t b0 = (uint32_t)(w.base + ay * w.bw + ax), b1 = b0 + (uint32_t)w.bw;
                            atomicOr(&bits_s[b0 >> 5], 1u << (b0 & 31u)); atomicOr(&bits_s[(b0 + 1u) >> 5], 1u << ((b0 + 1u) & 31u));
                            atomicOr(&bits_s[b1 >> 5], 1u << (b1 & 31u)); atomicOr(&bits_s[(b1 + 1u) >> 5], 1u << ((b1 + 1u) & 31u));
                            wr.nw = b0; wr.sw = b1;                              // bit indices until the ranks exist (below)
                        } else {
                            wr.nw = pool_off + (uint32_t)(w.base + ay * w.bw + ax) * SB;
                            wr.sw = wr.nw + (uint32_t)w.bw * SB;
                        }
                        wr.w[0] = sy * ex; wr.w[1] = sy * tx; wr.w[2] = ty * ex; wr.w[3] = ty * tx;      // folded below
                    } else {
                        wr.valid = kWinDirectMark; wr.w[0] = gx; wr.w[1] = gy;       // (nw, sw stay on the zero slices)
                        st |= kWinHasDirect;
                    }
                }
            }
            // sums over the views in view order (fusion.py:364-370)
            float dsum, cnt;
            uint32_t stp;
            view_sums(V, base, dv, valid, st, dsum, cnt, stp);
            if (!finite_maps) stp |= kWinStrict;                               // (the host picks this kernel for maps it expects to be finite)
            if (v < V) {
                if (stp & kWinStrict) {                                         // strict point: every pair from global, reference order
                    wr.nw = zero_off; wr.sw = zero_off; wr.valid = valid; wr.w[0] = gx; wr.w[1] = gy;
                } else {
                    const float sc = fold_scale(wr.wgt, cnt);                   // folded weights (fuse_common.h)
                    if (wr.valid == kWinDirectMark) wr.wgt = sc;
                    else { wr.w[0] = wr.w[0] * sc; wr.w[1] = wr.w[1] * sc; wr.w[2] = wr.w[2] * sc; wr.w[3] = wr.w[3] * sc; }
                }
                wrec_at(p)[v] = wr;
            }
            if (v == 0) {
                const bool all_invalid = (cnt == 0.0f);                         // fusion.py:366
                float dist_out = dsum / (cnt + 1e-6f);
                if (all_invalid) dist_out = 1e3f;                               // fusion.py:367
                cnt_s[p] = cnt;
                idx_s[p] = (uint32_t)i;
                flag_s[p] = stp;
                aux_s[2 * p] = dist_out;                                         // 'dist' / 'valid_mask' leave at the end of the kernel:
                aux_s[2 * p + 1] = cnt + 1e-6f;                                  // no store in front of the pool's DMA wait; the strict path's divisor
            }
        }
    }

    if constexpr (SPARSE) {
        __syncthreads();                            // every pair has marked its corners; flag_s / cnt_s are written
        if (threadIdx.x < 64) {                     // set bits in front of every bitmap word: one wave, one word per lane
            const uint32_t pc = (uint32_t)__popc(bits_s[threadIdx.x & (kWinMaxBits / 32 - 1)]);
            uint32_t incl = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if ((int)threadIdx.x >= off) incl += up;
            }
            wpre_s[threadIdx.x] = incl - pc;
            if (threadIdx.x == 63) total_s = (int)min(incl, (uint32_t)P.win_pool_texels);
        }
        __syncthreads();
        const uint32_t pool_n = (uint32_t)P.win_pool_texels;
        auto slot_of = [&](uint32_t b) -> uint32_t { return wpre_s[b >> 5] + (uint32_t)__popc(bits_s[b >> 5] & ((1u << (b & 31u)) - 1u)); };
        int nbits = 0;
        for (int vv = 0; vv < V && vv < kWinMaxViews; ++vv)
            if (win_s[vv].ok) nbits = win_s[vv].base + win_s[vv].bw * win_s[vv].bh;
        for (int b = threadIdx.x; b < nbits; b += NT)
            if ((bits_s[b >> 5] >> (b & 31)) & 1u) {
                const uint32_t slot = slot_of((uint32_t)b);
                if (slot < pool_n) texsrc_s[slot] = rect_texel(b);
            }
        __syncthreads();
        stage(0);                                   // the copy of slice 0 runs underneath the record fix-up
        // the records' bit indices become pool offsets; a pair whose texels did not get a slot (more touched texels than the pool
        // holds: rare) becomes a direct pair -- its gx, gy are projected again (no depth lookup: the pair is valid)
        for (int idx = threadIdx.x; idx < TP * (1 << pa_log2); idx += NT) {
            const int p = idx >> pa_log2, v = idx & ((1 << pa_log2) - 1);
            if (v >= V || (flag_s[p] & kWinStrict)) continue;
            WinRec *wr = wrec_at(p) + v;
            if (wr->valid != 1.0f) continue;                 // invalid (zero slices) or direct already
            const uint32_t sn = slot_of(wr->nw), ss = slot_of(wr->sw);
            if (sn + 1u < pool_n && ss + 1u < pool_n) {
                wr->nw = pool_off + sn * SB; wr->sw = pool_off + ss * SB;
            } else {
                float px, py, pz;
                slot_coords(p, slot_point(p), px, py, pz);
                const Proj pr = project_point(krt + v * 12, px, py, pz, Wm1, Hm1);
                wr->nw = zero_off; wr->sw = zero_off; wr->valid = kWinDirectMark;
                wr->wgt = fold_scale(wr->wgt, cnt_s[p]);
                wr->w[0] = pr.gx; wr->w[1] = pr.gy;
                atomicOr(&flag_s[p], kWinHasDirect);
            }
        }
    }
    D3F_STAMP();                                    // 4: phase A
    // ---- 4. phase B per slice: LPP lanes per point ----
    // (Round 4 also measured storing a slice's rows only after the next slice's DMA is issued -- gfx950 counts loads and
    // stores in ONE in-order counter, so the wait for the DMA also waits for every store issued before it: C2-patch -2 %,
    // C3-patch +1.5 %, C4-patch -0.4 % with the pipelined point loop; not kept.  Barriers between the slices are LDS-only.)
    const MapDesc &m = m0;
    const int l = threadIdx.x & (LPP - 1), grp = threadIdx.x / LPP;
    const uint32_t lane_off = (uint32_t)l * (uint32_t)RB;
    const char *__restrict__ data = reinterpret_cast<const char *>(m.data);
    const int S = P.win_slices;
    constexpr int G = NT / LPP;                     // points in flight per workgroup pass
    constexpr int KI = 4;                           // points of a lane group in the pipelined loop (TP = 64, LPP = 16: all of them)
    // fused channels of slot p for this lane (NV vectors): everything but the store
    int Vg = V;                                     // the general path's view count, opaque: behind `pipe_ok` the compiler knows
    asm volatile("" : "+s"(Vg));                    // V == VFIX and would unroll these view loops too (32 corner loads in flight: spills)
    using RT = typename WinRaw<HALF>::T;
    constexpr int NR = NV;                          // raw vectors per lane and corner
    // the four corner vectors of one pair from global memory (texel byte offset `cu` of this lane's raw vector r)
    auto point_slice = [&](int p, uint32_t co, VT (&acc)[NV]) {
        const int V = Vg;
        const uint32_t fl = flag_s[p];
        const bool strict = (fl & kWinStrict) != 0u;
#pragma unroll
        for (int u = 0; u < NV; ++u) acc[u] = (VT)0.0f;
        if (fl == 0u) {
            window_point<NV, VC, VS, (int)SB, HALF>(acc, smem, wrec_at(p), V, lane_off);
        } else if (!strict) {
            // some pair of this point is gathered from global memory (folded weights like the pool pairs)
            for (int v = 0; v < V; ++v) {
                const WinRec wr = wrec_at(p)[v];
                RT a[NR], b[NR], d[NR], e[NR];
                float w0, w1, w2, w3;
                if (wr.valid != kWinDirectMark) {
                    const unsigned char *nw = smem + (wr.nw + lane_off);
                    const unsigned char *sw = smem + (wr.sw + lane_off);
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        a[u] = *reinterpret_cast<const RT *>(nw + u * VS); b[u] = *reinterpret_cast<const RT *>(nw + SB + u * VS);
                        d[u] = *reinterpret_cast<const RT *>(sw + u * VS); e[u] = *reinterpret_cast<const RT *>(sw + SB + u * VS);
                    }
                    w0 = wr.w[0]; w1 = wr.w[1]; w2 = wr.w[2]; w3 = wr.w[3];
                } else {
                    const Corner c = corner_setup(m, wr.w[0], wr.w[1]);
                    const char *bv = data + (int64_t)v * m.sv * ES;
                    const float sc = wr.wgt;                                 // fold_scale of this pair (phase A)
                    w0 = (c.inw ? c.wnw : 0.0f) * sc; w1 = (c.ine ? c.wne : 0.0f) * sc;
                    w2 = (c.isw ? c.wsw : 0.0f) * sc; w3 = (c.ise ? c.wse : 0.0f) * sc;
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        const uint32_t cu = co + (uint32_t)u * (uint32_t)VS;
                        a[u] = *reinterpret_cast<const RT *>(bv + (c.onw + cu)); b[u] = *reinterpret_cast<const RT *>(bv + (c.one + cu));
                        d[u] = *reinterpret_cast<const RT *>(bv + (c.osw + cu)); e[u] = *reinterpret_cast<const RT *>(bv + (c.ose + cu));
                    }
                }
                win_accumulate<NV, HALF>(acc, a, w0);
                win_accumulate<NV, HALF>(acc, b, w1);
                win_accumulate<NV, HALF>(acc, d, w2);
                win_accumulate<NV, HALF>(acc, e, w3);
            }
        } else {
            for (int v = 0; v < V; ++v) {
                const WinRec wr = wrec_at(p)[v];
                const char *bv = data + (int64_t)v * m.sv * ES;
                const Corner c = corner_setup(m, wr.w[0], wr.w[1]);
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    // accumulator vector u = the four channels of raw vector u (fp16 storage: widened)
                    VT a, b, d, e;
                    if constexpr (HALF) {
                        const uint32_t cu = co + (uint32_t)u * (uint32_t)VS;
                        a = __builtin_convertvector(*reinterpret_cast<const f16x4 *>(bv + (c.onw + cu)), f32x4);
                        b = __builtin_convertvector(*reinterpret_cast<const f16x4 *>(bv + (c.one + cu)), f32x4);
                        d = __builtin_convertvector(*reinterpret_cast<const f16x4 *>(bv + (c.osw + cu)), f32x4);
                        e = __builtin_convertvector(*reinterpret_cast<const f16x4 *>(bv + (c.ose + cu)), f32x4);
                    } else {
                        const uint32_t cu = co + (uint32_t)u * (uint32_t)VS;
                        a = load_texel<4, false>(bv + (c.onw + cu)); b = load_texel<4, false>(bv + (c.one + cu));
                        d = load_texel<4, false>(bv + (c.osw + cu)); e = load_texel<4, false>(bv + (c.ose + cu));
                    }
                    const VT av = c.inw ? a : (VT)0.0f, bvv = c.ine ? b : (VT)0.0f, dv = c.isw ? d : (VT)0.0f, ev = c.ise ? e : (VT)0.0f;
                    VT s_ = av * c.wnw;
                    s_ = v_fma<VT>(bvv, c.wne, s_);
                    s_ = v_fma<VT>(dv, c.wsw, s_);
                    s_ = v_fma<VT>(ev, c.wse, s_);
                    acc[u] = acc[u] + (s_ * wr.valid) * wr.wgt;
                }
            }
            // the reference's division (fusion.py:385-386); the fast path's weights carry 1/(cnt + 1e-6) already, and with no
            // valid view every weight is zero, acc is +0 -- fusion.py:386 for free
            const float cnt = cnt_s[p], denom = aux_s[2 * p + 1];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                VT o = (VT)0.0f;
                if (cnt != 0.0f) o = strict_div<VT>(acc[u], denom);
                acc[u] = o;
            }
        }
    };
    // NON-TEMPORAL write-through stores (`sc1 nt`, store_row_vec), one 64-bit multiply-add per row and no store-flavour branches
    // inside the point loop.  The rows are written once and never read by the launch; as plain (or sc1) stores they allocate in
    // the L2s and the Infinity Cache on their way out and push out the texels the next bricks' window copies would have hit
    // (round 4, found with what-if builds: with the rows stored over each other in 8 MiB the gather ran 26 % faster, with `nt`
    // stores to their real addresses 19 %): C2-patch 0.48-0.52 -> 0.445 ms, C3-patch 1.07-1.17 -> 0.92, C4-patch 1.83 -> 1.70,
    // the reference's shape 2.79 -> 2.12; `sc1 nt` instead of `nt` alone: C2-patch 0.455 -> 0.415, the reference's shape 2.11 -> 2.02.
    // (A flagged point's row is stored twice by the same lane to the same address: program order holds for those.)
    const uint32_t row_bytes = (uint32_t)m.C * 4u;
    char *const out_bytes = reinterpret_cast<char *>(m.out);
    // (oco: byte offset of this lane's first accumulator vector in an output row -- the texel offset `co` for fp32 maps)
    auto store_point = [&](int p, uint32_t oco, const VT (&acc)[NV]) {
        char *row = out_bytes + ((uint64_t)idx_s[p] * row_bytes + oco);
#pragma unroll
        for (int u = 0; u < NV; ++u) store_row_vec(row + u * OVS, acc[u]);
    };
    // The pipelined point loop runs for ALL points of the lane group: the records of a point with a direct pair, and of a
    // strict point, point at the zero slices (harmless reads); such a point (rare: rim rounding, pool overflow, non-finite
    // projection) is then done again by the general path and its row stored a second time -- same lane, same address, later in
    // program order.  (Round 4's first form let one flagged point send its whole wave to the general path for every slice:
    // C2-patch 0.52 ms at 4 workgroups per CU against 0.50 at 3 with the larger pool -- the overflow, not the occupancy.)
    const bool pipe_ok = VFIX > 0 && TP == KI * G && V == VFIX;
    uint32_t pidx[KI] = {0u, 0u, 0u, 0u};          // global index of the lane group's points (read once: an LDS read inside the
                                                   // pipelined loop would drain it -- LDS reads return in order)
    if (pipe_ok) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");          // idx_s of phase A
#pragma unroll
        for (int k = 0; k < KI; ++k) pidx[k] = idx_s[grp + k * G];
    }
    for (int sl = 0; sl < S; ++sl) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the DMA of this slice has landed (and phase A's records)
        D3F_STAMP();                                // 5 + 3 sl: pool of this slice ready
        const uint32_t co = (uint32_t)sl * SB + lane_off;               // byte offset of this lane's first vector in a texel
        const uint32_t oco = (uint32_t)sl * OSB + (uint32_t)l * 16u;             // ... of its first four channels in an output row (fp32)
        if (pipe_ok) {
            window_points_pipelined<NV, (VFIX > 0 ? VFIX : 1), KI, VS, (int)SB, G * ((VFIX > 0 ? VFIX : 1) * 32 + 16), HALF>(
                smem, (uint32_t)grp * pstride, lane_off, [&](int k, const VT (&acc)[NV]) {
                    char *row = out_bytes + ((uint64_t)pidx[k] * row_bytes + oco);
#pragma unroll
                    for (int u = 0; u < NV; ++u) store_row_vec(row + u * OVS, acc[u]);
                });
#pragma unroll 1
            for (int p = grp; p < TP; p += G)
                if (flag_s[p] != 0u) {              // a direct pair or a strict point: the general path, stored over the row above
                    VT acc[NV];
                    point_slice(p, co, acc);
                    store_point(p, oco, acc);
                }
        } else {
            for (int p = grp; p < TP; p += G) {
                VT acc[NV];
                point_slice(p, co, acc);
                store_point(p, oco, acc);
            }
        }
        D3F_STAMP();                                // 6 + 3 sl: this wave's points of the slice done
        if (sl + 1 < S) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: everyone is done with this slice's pool
            D3F_STAMP();                            // 7 + 3 sl: all waves done
            stage(sl + 1);
        } else {
            D3F_STAMP();
        }
    }
    for (int p = threadIdx.x; p < TP; p += NT) {       // per-point outputs (clipped slots repeat a neighbour: same values twice)
        P.out_dist[idx_s[p]] = aux_s[2 * p];
        P.out_valid[idx_s[p]] = cnt_s[p] == 0.0f ? 0 : 1;
    }
    D3F_STAMP();                                    // last: rows stored
#ifdef D3F_EXPERIMENTS
    if (stamping) P.exp_stamps[(size_t)(blockIdx.x >> 6) * 32] = (unsigned long long)stamp_k;
#endif
#undef D3F_STAMP
    // the other (thin) maps of the call (their gather is written for kBlock lanes)
    if (NT > kBlock && threadIdx.x >= kBlock) return;
    for (int s = 1; s < P.n_maps; ++s) {
        const MapDesc &mt = P.maps[s];
        switch (mt.vw) {
        case 4: gather_map_u<4, false, true>(mt, P, rec, cnt_s, flag_s, idx_s, 0, TP, nullptr); break;
        case 2: gather_map_u<2, false, true>(mt, P, rec, cnt_s, flag_s, idx_s, 0, TP, nullptr); break;
        default: gather_map_u<1, false, true>(mt, P, rec, cnt_s, flag_s, idx_s, 0, TP, nullptr); break;
        }
    }
}

// (NT lanes per workgroup: 512 lanes over the same 64-point brick -- twice the waves per pool -- measured no faster
// at 2, 3 or 4 workgroups per CU: 0.58-0.71 ms on C2 patch against 0.58; only the 256-lane form is built)
// ---- window kernel or cell runs for a cloud?  decided on the device (round 5) --------------------------------------------------
// Whether 64 consecutive points of a cloud's Hilbert order touch few enough texels for the pool depends on the cloud's density
// against the texel grid (C2-patch, 1 M points: 47 texels per tile; 100 k keypoints: 105; config 4's 8 views x 6.7-mm texels: 176) --
// nothing the host knows without reading the points.  So for a cloud BOTH launches are enqueued, each gated on one device word:
// window_gate_probe_kernel counts, over <= kGateSamples evenly spaced tiles, those whose touched texels fit the pool (the window
// kernel's own steps 1-3: box, rectangles, bitmap), the last workgroup to finish publishes the count, and each kernel's workgroups
// return at once unless the count is on their side of `gate_min`.  No host sync, capturable in a HIP graph; the losing launch
// costs its dispatch (a few microseconds).
template <int U, int VC, int WAVES, int NT = kBlock, int LPP = 32, int VFIX = 0, bool SPARSE = false, bool HALF = false>
__global__ __launch_bounds__(NT, WAVES) void fused_eval_window_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    fused_eval_window_body<U, VC, NT, LPP, VFIX, SPARSE, HALF>(P);
}

// gate[0] verdict (tiles that fit), gate[1] running count, gate[2] workgroups done; [1] and [2] are zero between launches
// (order_prepare_kernel zeroes them with the counting table; the last workgroup resets them)
__global__ __launch_bounds__(kBlock) void window_gate_probe_kernel(const EvalParams P, uint32_t *__restrict__ gate, int nsamples)
{
    __shared__ float krt[kWinMaxViews * 12];
    __shared__ float cpt_s[8][3];
    __shared__ float red_s[kBlock / 64][6];
    __shared__ WinView win_s[kWinMaxViews];
    __shared__ uint32_t bits_s[kWinMaxBits / 32];
    const int V = P.V, TP = P.tile_pts;
    const MapDesc &m0 = P.maps[0];
    const int64_t ntiles = (P.n + TP - 1) / TP;
    const int64_t tile = (int64_t)blockIdx.x * ntiles / nsamples;
    const int64_t tile_base = tile * TP;
    const int tile_n = (int)min((int64_t)TP, P.n - tile_base);
    const float Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);
    auto point_at = [&](int p, float &px, float &py, float &pz) {
        const int64_t q = tile_base + min(p, tile_n - 1);
        fetch_point(P, P.order ? min((int64_t)P.order[q], P.n - 1) : q, px, py, pz);
    };
    compute_krt(P.K, P.pose, V, krt, kBlock);
    if (threadIdx.x < kWinMaxBits / 32) bits_s[threadIdx.x] = 0u;
    // step 1: the tile's bounding box
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int p = threadIdx.x; p < tile_n; p += kBlock) {
            float q[3];
            point_at(p, q[0], q[1], q[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], q[k]); hi[k] = fmaxf(hi[k], q[k]); }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
            }
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < 3; ++k) { red_s[threadIdx.x >> 6][k] = lo[k]; red_s[threadIdx.x >> 6][3 + k] = hi[k]; }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int c = threadIdx.x;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float a = red_s[0][k], b = red_s[0][3 + k];
                for (int w = 1; w < kBlock / 64; ++w) { a = fminf(a, red_s[w][k]); b = fmaxf(b, red_s[w][3 + k]); }
                cpt_s[c][k] = ((c >> k) & 1) ? b : a;
            }
        }
    }
    __syncthreads();
    // step 2: one texel rectangle per view (wave 0, lane = view * 8 + corner), bitmap bits in view order
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, v = lane >> 3, c = lane & 7;
        const bool act = v < V && V <= kWinMaxViews;
        float xl = 0.0f, xh = 0.0f, yl = 0.0f, yh = 0.0f;
        int ok = 0;
        if (act) {
            const Proj pr = project_point(krt + v * 12, cpt_s[c][0], cpt_s[c][1], cpt_s[c][2], Wm1, Hm1);
            const float ix = unnormalize(pr.gx, m0.fw), iy = unnormalize(pr.gy, m0.fh);
            ok = (pr.ok && pr.zc > 1e-4f && isfinite(ix) && isfinite(iy)) ? 1 : 0;
            xl = xh = ix; yl = yh = iy;
        }
#define D3F_WIN_RED(CTRL)                                                                                   \
        xl = fminf(xl, dpp_f<CTRL>(xl)); xh = fmaxf(xh, dpp_f<CTRL>(xh));                                   \
        yl = fminf(yl, dpp_f<CTRL>(yl)); yh = fmaxf(yh, dpp_f<CTRL>(yh));                                   \
        ok &= dpp_i<CTRL>(ok);
        D3F_WIN_RED(0xB1) D3F_WIN_RED(0x4E) D3F_WIN_RED(0x141)
#undef D3F_WIN_RED
        WinView w = {0, 0, 1, 1, 0, 0};
        int ntex = 0;
        if (ok) {
            const float fwm1 = (float)(m0.fw - 1), fhm1 = (float)(m0.fh - 1);
            const int x0 = (int)fminf(fmaxf(floorf(xl - 1e-3f), 0.0f), fwm1), x1 = (int)fminf(fmaxf(floorf(xh + 1e-3f) + 1.0f, 0.0f), fwm1);
            const int y0 = (int)fminf(fmaxf(floorf(yl - 1e-3f), 0.0f), fhm1), y1 = (int)fminf(fmaxf(floorf(yh + 1e-3f) + 1.0f, 0.0f), fhm1);
            w.xmin = x0; w.ymin = y0; w.bw = x1 - x0 + 1; w.bh = y1 - y0 + 1;
            ntex = w.bw * w.bh;
        }
        int run = 0;
        for (int vv = 0; vv < V && vv < kWinMaxViews; ++vv) {
            const int nv = __builtin_amdgcn_readlane(ntex, vv * 8), bwv = __builtin_amdgcn_readlane(w.bw, vv * 8);
            int rows = nv > 0 ? min(nv, kWinMaxBits - run) / bwv : 0;
            if (rows < 2) rows = 0;
            if (vv == v) { w.base = run; w.ok = rows > 0 ? 1 : 0; w.bh = rows > 0 ? rows : w.bh; }
            run += rows * bwv;
        }
        if (act && c == 0) win_s[v] = w;
    }
    __syncthreads();
    // step 3: every valid pair marks its four corner texels; a pair outside its rectangle counts as a miss for the tile
    const int vp_log2 = V <= 1 ? 0 : (V <= 2 ? 1 : (V <= 4 ? 2 : 3));
    int outside = 0;
    for (int idx = threadIdx.x; idx < (tile_n << vp_log2); idx += kBlock) {
        const int p = idx >> vp_log2, v = idx & ((1 << vp_log2) - 1);
        if (v >= V) continue;
        float px, py, pz, wgt;
        point_at(p, px, py, pz);
        const ViewOut o = eval_view<0>(P.depth, P.H, P.W, krt + v * 12, v, px, py, pz, Wm1, Hm1, P.mu, wgt);
        if (o.valid == 0.0f) continue;
        const float x0 = floorf(unnormalize(o.gx, m0.fw)), y0 = floorf(unnormalize(o.gy, m0.fh));
        const WinView w = win_s[v];
        const bool inmap = x0 >= 0.0f && x0 <= (float)(m0.fw - 2) && y0 >= 0.0f && y0 <= (float)(m0.fh - 2);
        if (!inmap) continue;                               // image border: a direct pair in any case
        const int ax = (int)x0 - w.xmin, ay = (int)y0 - w.ymin;
        if (w.ok && ax >= 0 && ax + 1 < w.bw && ay >= 0 && ay + 1 < w.bh) {
            const uint32_t b0 = (uint32_t)(w.base + ay * w.bw + ax), b1 = b0 + (uint32_t)w.bw;
            atomicOr(&bits_s[b0 >> 5], 1u << (b0 & 31u)); atomicOr(&bits_s[(b0 + 1u) >> 5], 1u << ((b0 + 1u) & 31u));
            atomicOr(&bits_s[b1 >> 5], 1u << (b1 & 31u)); atomicOr(&bits_s[(b1 + 1u) >> 5], 1u << ((b1 + 1u) & 31u));
        } else {
            outside = 1;
        }
    }
    const int any_outside = __syncthreads_or(outside);
    if (threadIdx.x < 64) {
        uint32_t pc = (uint32_t)__popc(bits_s[threadIdx.x]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pc += __shfl_xor(pc, off, 64);
        if (threadIdx.x == 0) {
            const uint32_t fits = (!any_outside && pc <= (uint32_t)P.win_pool_texels) ? 1u : 0u;
            const uint32_t before = atomicAdd(&gate[1], fits);
            __threadfence();
            const uint32_t done = atomicAdd(&gate[2], 1u);
            (void)before;
            if (done == (uint32_t)nsamples - 1u) {          // the last workgroup: publish and reset
                __threadfence();
                const uint32_t total = atomicAdd(&gate[1], 0u);
                gate[0] = total;
                gate[1] = 0u; gate[2] = 0u;
            }
        }
    }
}

hipError_t launch_window_gate_probe(const EvalParams &P, uint32_t *gate, int nsamples, hipStream_t stream)
{
    hipLaunchKernelGGL(window_gate_probe_kernel, dim3((unsigned)nsamples), dim3(kBlock), 0, stream, P, gate, nsamples);
    return hipGetLastError();
}

hipError_t launch_window(const EvalParams &P, hipStream_t stream)
{
    int64_t ntiles = (P.n + P.tile_pts - 1) / P.tile_pts;
    if (P.walk_nx > 0)
        ntiles = (int64_t)((P.walk_nx + P.walk_tx - 1) / P.walk_tx) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) *
                 ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
    dim3 block(kBlock);
    const bool half = P.maps[0].esize == 2;        // fp16-stored map: 256-byte slices (lattices only, 16 lanes per point)
    const size_t lds_w = (size_t)P.win_pool_offset + (size_t)(2 + P.win_pool_texels) * (half ? 256 : 512) * P.win_u;
    dim3 gw((unsigned)((P.n + P.tile_pts - 1) / P.tile_pts));
    if (P.walk_nx > 0) gw = dim3((unsigned)ntiles);
#define D3F_WIN_LAUNCH_S(U_, VC_, W_, LPP_, VF_, SP_)                                                                          \
    do {                                                                                                                       \
        if (lds_w > 64 * 1024) {                                                                                               \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_eval_window_kernel<U_, VC_, W_, kBlock, LPP_, VF_, SP_>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);                      \
            if (ea != hipSuccess) return ea;                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((fused_eval_window_kernel<U_, VC_, W_, kBlock, LPP_, VF_, SP_>), gw, block, lds_w, stream, P);      \
    } while (0)
#define D3F_WIN_LAUNCH_F(U_, VC_, W_, LPP_, VF_)                                                                               \
    do {                                                                                                                       \
        if (P.win_sparse) D3F_WIN_LAUNCH_S(U_, VC_, W_, LPP_, VF_, true);                                                      \
        else D3F_WIN_LAUNCH_S(U_, VC_, W_, LPP_, VF_, false);                                                                  \
    } while (0)
#define D3F_WIN_LAUNCH(U_, VC_, W_, LPP_) D3F_WIN_LAUNCH_F(U_, VC_, W_, LPP_, 0)
    // the product library holds the variants the planner picks by itself: 16 lanes x two vectors per point, at 4 or 3
    // workgroups per CU, with the view count fixed at 4 / 8 (software-pipelined point loop) or free; the others exist in
    // experiments builds only (measured and dropped)
    const bool lpp16 = P.win_u == 1 && P.win_lpp == 16;
    const int vfix = (P.win_pipe && P.tile_pts == 64) ? (P.V == 4 ? 4 : (P.V == 8 ? 8 : 0)) : 0;
#ifdef D3F_EXPERIMENTS
    // win_vc 2: two views' corner reads in flight in the plain view loop (round 3's form)
    if (lpp16 && vfix == 0 && P.win_occ == 6) D3F_WIN_LAUNCH_F(1, 1, 6, 16, 0);          // plain loop at 6 / 5 workgroups per CU (smaller pools)
    else if (lpp16 && vfix == 0 && P.win_occ == 5) D3F_WIN_LAUNCH_F(1, 1, 5, 16, 0);
    else if (lpp16 && vfix == 0 && P.win_vc == 2 && P.win_occ >= 4) D3F_WIN_LAUNCH_F(1, 2, 4, 16, 0);
    else if (lpp16 && vfix == 0 && P.win_vc == 2) D3F_WIN_LAUNCH_F(1, 2, 3, 16, 0);
    else
#endif
    // ONE register budget (<= 128 VGPRs: four waves per SIMD) serves every pool size: the workgroups per CU follow from the
    // dynamic LDS of the launch (win_occ sized the pool), not from the kernel variant -- up to round 4 a second set held to
    // __launch_bounds__(256, 3) existed and allocated 121 instead of 125 registers, the same occupancy step
#define D3F_WIN_LAUNCH_H(VF_)                                                                                                 \
    do {                                                                                                                       \
        if (lds_w > 64 * 1024) {                                                                                               \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_eval_window_kernel<1, 1, 4, kBlock, 16, VF_, false, true>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);                      \
            if (ea != hipSuccess) return ea;                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((fused_eval_window_kernel<1, 1, 4, kBlock, 16, VF_, false, true>), gw, block, lds_w, stream, P);    \
    } while (0)
    if (half) {
        if (!lpp16 || P.win_sparse) return hipErrorInvalidValue;
        if (vfix == 4) D3F_WIN_LAUNCH_H(4);
        else if (vfix == 8) D3F_WIN_LAUNCH_H(8);
        else D3F_WIN_LAUNCH_H(0);
        return hipGetLastError();
    }
#undef D3F_WIN_LAUNCH_H
    if (lpp16 && vfix == 4) D3F_WIN_LAUNCH_F(1, 1, 4, 16, 4);
    else if (lpp16 && vfix == 8) D3F_WIN_LAUNCH_F(1, 1, 4, 16, 8);
    else if (lpp16) D3F_WIN_LAUNCH(1, 1, 4, 16);
#ifdef D3F_EXPERIMENTS
    else if (P.win_u == 1 && P.win_occ >= 4) D3F_WIN_LAUNCH(1, 4, 4, 32);
    else if (P.win_u == 1 && P.win_occ == 3) D3F_WIN_LAUNCH(1, 4, 3, 32);
    else if (P.win_u == 1) D3F_WIN_LAUNCH(1, 4, 2, 32);
    else if (P.win_u == 2 && P.win_vc == 2) D3F_WIN_LAUNCH(2, 2, 2, 32);
    else if (P.win_u == 2) D3F_WIN_LAUNCH(2, 1, 2, 32);
    else if (P.win_u == 3 && P.win_vc == 2) D3F_WIN_LAUNCH(3, 2, 2, 32);
    else if (P.win_u == 3) D3F_WIN_LAUNCH(3, 1, 2, 32);
    else D3F_WIN_LAUNCH(4, 1, 2, 32);
#else
    else return hipErrorInvalidValue;
#endif
#undef D3F_WIN_LAUNCH
#undef D3F_WIN_LAUNCH_F
#undef D3F_WIN_LAUNCH_S
    return hipGetLastError();
}

}  // namespace d3f
