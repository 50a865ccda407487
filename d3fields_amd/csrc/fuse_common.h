// fuse_common.h -- device code shared by the fused field-query kernels (fuse_direct / fuse_runs / fuse_sliced / fuse_window .hip): per-(point, view) records, the bilinear corner set-up, the direct gather of one map, the thin-map
// gather, output stores, the closed-form lattice walk.  Arithmetic contract: DESIGN.md section 2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"

namespace d3f {

// ---- per-(point,view) arithmetic: identical, operation for operation, to the oracle -----

struct ViewRec {
    float gx, gy;   // normalised image coordinates   (fusion.py:72-73)
    float wgt;      // exp(clamp(mu-|dist|,max=0)/mu)  (fusion.py:347)
    float valid;    // 1.0f / 0.0f                     (fusion.py:344)
};

// Bilinear corner set-up of one (point, view) for one map (grid_sample, align_corners=True, zeros padding).
struct Corner {
    uint32_t onw, one, osw, ose;    // 32-bit BYTE offsets of the clamped corner texels from the view's base
    float wnw, wne, wsw, wse;       // bilinear weights
    bool inw, ine, isw, ise;        // corner inside the map?
};

// Coordinates are clamped into the map so that every address is valid; out-of-bounds corners become the
// zeros of padding_mode='zeros' by zeroing the weight (finite operands) or by a select on the value.
__device__ __forceinline__ Corner corner_setup(const MapDesc &m, float gx, float gy)
{
    Corner c;
    const float fwm1 = (float)(m.fw - 1), fhm1 = (float)(m.fh - 1);
    const uint32_t es = (uint32_t)m.esize;
    const uint32_t sy_b = (uint32_t)m.sy * es, sx_b = (uint32_t)m.sx * es;   // host guarantees a view spans < 4 GiB
    const float ix = unnormalize(gx, m.fw), iy = unnormalize(gy, m.fh);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float tx = ix - x0, ty = iy - y0;
    const float ex = 1.0f - tx, sy = 1.0f - ty;
    c.wnw = sy * ex; c.wne = sy * tx; c.wsw = ty * ex; c.wse = ty * tx;
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    c.inw = in_bounds(x0, y0, m.fw, m.fh); c.ine = in_bounds(x1, y0, m.fw, m.fh);
    c.isw = in_bounds(x0, y1, m.fw, m.fh); c.ise = in_bounds(x1, y1, m.fw, m.fh);
    const int xi0 = (int)fminf(fmaxf(x0, 0.0f), fwm1), xi1 = (int)fminf(fmaxf(x1, 0.0f), fwm1);
    const int yi0 = (int)fminf(fmaxf(y0, 0.0f), fhm1), yi1 = (int)fminf(fmaxf(y1, 0.0f), fhm1);
    const uint32_t r0 = (uint32_t)yi0 * sy_b, r1 = (uint32_t)yi1 * sy_b;
    const uint32_t q0 = (uint32_t)xi0 * sx_b, q1 = (uint32_t)xi1 * sx_b;
    c.onw = r0 + q0; c.one = r0 + q1; c.osw = r1 + q0; c.ose = r1 + q1;
    return c;
}

// The same set-up computed ONCE per (point, view) in phase A (one lane per pair) for wide maps, so that the
// 2^k lanes of a point's group read 32 bytes from LDS instead of repeating ~45 VALU instructions each.
struct __attribute__((aligned(16))) CornerRec {
    uint32_t o[4];      // onw, one, osw, ose
    float w[4];         // weights with out-of-bounds corners already zeroed (non-strict path only); of a FOLDED map
                        // (MapDesc::fold) they are already multiplied by fold_scale()
};

// ---- folded weights of wide maps (fast path only) ------------------------------------------------
// The reference computes  out = (sum_v (bilinear_v * valid_v) * wgt_v) / (cnt + 1e-6)  (fusion.py:373-386): per channel
// and view a 4-term bilinear chain, a product with the weight, an addition, and one division per channel.  On the fast
// path (finite operands, valid_v == 1 for every view that takes part) the scalar factors of a (point, view) are folded
// into its four bilinear weights ONCE:   w'_k = w_k * (wgt_v * rcp(cnt + 1e-6)),
// and the channels run   acc = fma(nw, w'_0, acc); fma(ne, w'_1, .); fma(sw, w'_2, .); fma(se, w'_3, .)   in view order;
// out = acc.  Four packed fma per channel vector and view instead of six operations, and no division epilogue.  Every
// kernel that takes the fast path uses exactly these operations in this order (bit-identical among themselves); against
// the reference's order the result moves by a few ulp of the largest term (tests: <= 1e-5 of max|ref|, measured ~2e-7).
// Thin maps (the instance mask, colours: <= 256 bytes per texel) and every strict point keep the reference's order, so
// instance indices and '<k>_inter' stay bit-exact.
__device__ __forceinline__ float refined_rcp(float denom)
{
    // 1/denom to within an ulp: hardware estimate + one Newton step (denom = cnt + 1e-6 lies in [1e-6, V + 1))
    const float r0 = __builtin_amdgcn_rcpf(denom);
    return fmaf(fmaf(-denom, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float fold_scale(float wgt, float cnt) { return wgt * refined_rcp(cnt + 1e-6f); }

// lanes per point in phase A: the views of a point sit in VP = 2^k >= V adjacent lanes
__device__ __forceinline__ int view_lanes_log2(int V) { return V <= 1 ? 0 : 32 - __clz(V - 1); }

// ---- cross-lane moves on the VALU (DPP) instead of the LDS crossbar -----------------------------------------------------
// hipcc's __shfl / __shfl_xor are ds_bpermute_b32: an LDS-pipe round trip (~100+ cycles under load) per value.  Phase A
// needs 3 V of them per lane for the per-point sums, the window set-up 27 for its min / max reductions -- 1.5-3 k cycles
// of a workgroup's serial prologue (phase stamps, round 4).  Inside a quad / a row DPP does the same in one VALU
// instruction.  dpp_ctrl: quad_perm [a,b,c,d] = a | b<<2 | c<<4 | d<<6; 0x141 = row_half_mirror (lane i <-> 7 - i of its 8).
template <int CTRL> __device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) { return __int_as_float(dpp_i<CTRL>(__float_as_int(x))); }
template <int K> __device__ __forceinline__ float quad_bcast(float x) { return dpp_f<K * 0x55>(x); }       // lane K of the quad
template <int K> __device__ __forceinline__ int quad_bcast_i(int x) { return dpp_i<K * 0x55>(x); }

// Sums over the views of a point IN VIEW ORDER (fusion.py:364-370) from the per-(point, view) lanes of phase A: lane
// `base` + v holds view v (VP = 2^k >= V lanes per point, base a multiple of VP).  Every lane of the wave must call this
// (cross-lane reads of inactive lanes are undefined).  V <= 4: quad broadcasts; V <= 8: two quads chained through a row
// shift; more views: the LDS crossbar.  The additions are the same operands in the same order in every branch.
__device__ __forceinline__ void view_sums(int V, int base, float dv, float valid, uint32_t st, float &dsum, float &cnt,
                                          uint32_t &st_any)
{
    dsum = 0.0f; cnt = 0.0f; st_any = 0u;
    if (V <= 8) {
        // views 0..3: lanes 0..3 of the point's first quad (a point with V <= 4 owns <= one quad: VP = 1, 2, 4 lanes; for
        // VP < 4 a quad holds several points and lane k of the quad is view k - (base & 3) of ITS point: select by lane)
        const int q = base & 3;                         // first lane of the point inside its quad (0 unless VP < 4)
#define D3F_VS_STEP(K)                                                                                              \
        {                                                                                                           \
            const float a = quad_bcast<K>(dv), b = quad_bcast<K>(valid);                                            \
            const int c = quad_bcast_i<K>((int)st);                                                                 \
            const bool mine = (K >= q) && (K - q < V) && (K - q < 4);                                               \
            dsum = mine ? dsum + a : dsum; cnt = mine ? cnt + b : cnt; st_any |= mine ? (uint32_t)c : 0u;          \
        }
        D3F_VS_STEP(0) D3F_VS_STEP(1) D3F_VS_STEP(2) D3F_VS_STEP(3)
        if (V > 4) {
            // views 4..7 sit in the next quad: hand the running sums over (row_shr:4 = lane i-4 -> lane i), continue the
            // chain there, and hand the totals back (row_shl:4) -- VP = 8, so base is a multiple of 8 and q = 0
            const bool hi = (threadIdx.x & 4) != 0;
            const float d0 = dpp_f<0x114>(dsum), c0 = dpp_f<0x114>(cnt);
            const int s0 = dpp_i<0x114>((int)st_any);
            float d1 = d0, c1 = c0;
            uint32_t s1 = (uint32_t)s0;
#define D3F_VS_HI(K)                                                                                                \
            {                                                                                                       \
                const float a = quad_bcast<K>(dv), b = quad_bcast<K>(valid);                                        \
                const int c = quad_bcast_i<K>((int)st);                                                             \
                const bool mine = 4 + K < V;                                                                        \
                d1 = mine ? d1 + a : d1; c1 = mine ? c1 + b : c1; s1 |= mine ? (uint32_t)c : 0u;                    \
            }
            D3F_VS_HI(0) D3F_VS_HI(1) D3F_VS_HI(2) D3F_VS_HI(3)
            const float dl = dpp_f<0x104>(d1), cl = dpp_f<0x104>(c1);          // back to the low quad
            const int sl = dpp_i<0x104>((int)s1);
            dsum = hi ? d1 : dl; cnt = hi ? c1 : cl; st_any = hi ? s1 : (uint32_t)sl;
#undef D3F_VS_HI
        }
#undef D3F_VS_STEP
        return;
    }
    for (int vv = 0; vv < V; ++vv) {
        dsum = dsum + __shfl(dv, base + vv, 64);
        cnt = cnt + __shfl(valid, base + vv, 64);
        st_any |= (uint32_t)__shfl((int)st, base + vv, 64);
    }
}

// may invalid views be skipped exactly?  (the host verified the maps, or every device-side check word is zero)
__device__ __forceinline__ bool maps_are_finite(const EvalParams &P)
{
    if (P.flags & kFlagFiniteMaps) return true;
    if (P.n_words == 0) return false;
    uint32_t bad = 0u;
    for (int k = 0; k < P.n_words; ++k) bad |= __builtin_nontemporal_load(P.words[k]);
    return bad == 0u;
}

// Gathers map `m` for the points of this workgroup's tile.
//   VW  channel-vector width in floats (4 when C%4==0 and 16-B aligned, else 2 or 1)
//   U   channel vectors per lane per pass
// A group of LPP = 1<<lpp_log2 lanes serves one point; lane g of the group owns channel
// vectors  pass*LPP*U + u*LPP + g  (u < U), so one load instruction of a group covers
// LPP*VW*4 contiguous bytes of a texel.
//   HALF  the map is stored in fp16 (D3F_DTYPE_F16): VW = 8 channels per 16-B load (or scalar lanes), widened to
//         fp32 on load; everything after the load is the fp32 path
// Output rows are written once and never read again by the launch: they leave as NON-TEMPORAL stores (policy 2: `nt`; the window
// kernel adds write-through, `sc1 nt` = store_row_vec, which is policy 3 here, experiments builds: it gains 4-9 % there and loses
// 0-5 % in these kernels, scripts/notebook/gpu_sessions/r4_gpu34/35.sh), which stream
// to memory without allocating in the L2s and the Infinity Cache, i.e. without pushing out the texels the gather lives on
// (round 4: C2 dense 1.53 -> 1.47 ms, C3 dense 2.83 -> 2.66, C4 dense 8.97 -> 8.94; the window kernel, which stores with the same
// policy, gained 8-23 %).  Round 2's `sc1` stores (policy 1: write-through, the line dropped from the XCD's L2 after the
// write) had bought 1 % over plain ones (policy 0); experiments builds: D3F_EXP_STORE=1 / -1.
// One 16-byte piece of an output row, non-temporal and write-through: `global_store_dwordx4 ... sc1 nt` (no builtin emits the
// pair; `nt` alone is __builtin_nontemporal_store).  s_nop 1: see store_out.
__device__ __forceinline__ void store_row_vec(void *p, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <typename VT>
__device__ __forceinline__ void store_out(float *p, VT v, int policy)
{
    if (policy >= 2) {                              // non-temporal: streams to memory past the L2's and the Infinity Cache's allocation
        if constexpr (sizeof(VT) == 16) {
            if (policy == 3) { store_row_vec(p, v); return; }       // ... and write-through (experiments; the window kernel's form)
        }
        __builtin_nontemporal_store(v, reinterpret_cast<VT *>(p));
        return;
    }
    if constexpr (sizeof(VT) == 16) {
        if (policy == 1) {
            // s_nop 1: a VMEM store of more than 64 bits reads its data registers for two more cycles on gfx940+; the
            // compiler inserts those wait states for its own stores but cannot see inside the asm, and a VALU write
            // to v scheduled right behind it tore dwords of some lanes (found with the 2-vector window kernel)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
            return;
        }
    }
    store_vec<VT>(p, v);
}

// The full IEEE division of the strict path.  The empty volatile asm keeps it inside its (rare) branch: without it the
// compiler if-converts `strict ? a / d : fast` and every point pays the ~11 instructions per channel of the division
// it does not use (44 of ~120 VALU instructions per point and channel vector in the epilogues, measured round 2).
// the same store addressed as (uniform base, 32-bit byte offset): rows of outputs below 4 GiB need no 64-bit arithmetic
__device__ __forceinline__ void store_out_off(float *base, uint32_t off, f32x4 v, int policy)
{
    if (policy == 1) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
        return;
    }
    if (policy == 3) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
        return;
    }
    if (policy == 2) {
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(base) + off));
        return;
    }
    *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(base) + off) = v;
}

template <typename VT>
__device__ __forceinline__ VT strict_div(VT a, float d)
{
    asm volatile("" ::);
    return a / d;
}

template <int VW, int U, bool BATCH, bool HALF = false, bool FOLD = false>
__device__ __forceinline__ void gather_map(const MapDesc &m, const EvalParams &P, const ViewRec *rec,
                                           const float *cnt_s, const uint32_t *flag_s,
                                           const uint32_t *idx_s, int64_t idx_base, int tile_n, const CornerRec *crec,
                                           bool only_strict = false, int tid = threadIdx.x, int nth = kBlock)
{
    // tid / nth: the calling lane's index among the nth lanes that share the tile (a whole workgroup by default)
    using VT = typename Vec<VW>::T;
    const int lpp = 1 << m.lpp_log2;
    const int g = tid & (lpp - 1);
    const int grp = tid >> m.lpp_log2;
    const int ngrp = nth >> m.lpp_log2;
    const int cvec = m.C / VW;
    const int V = P.V;
    const float *__restrict__ data = m.data;
    constexpr int ES = HALF ? 2 : 4;                  // bytes per stored channel

    for (int p = grp; p < tile_n; p += ngrp) {
        const int64_t i = idx_base + idx_s[p];
        const float cnt = cnt_s[p];
        const bool all_invalid = (cnt == 0.0f);           // fusion.py:366
        const float denom = cnt + 1e-6f;                  // fusion.py:385
        const bool strict = (flag_s[p] != 0u) || (m.inter != nullptr);
        if (only_strict && !strict) continue;             // the cell-run gather already wrote this point
        for (int c0 = 0; c0 < cvec; c0 += lpp * U) {
            VT acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = (VT)0.0f;
            for (int v = 0; v < V; ++v) {
                const ViewRec r = rec[p * V + v];
                if (!strict && r.valid == 0.0f) continue;  // exact: +0 + (+-0) == +0, x + (+-0) == x
                // Branch-free corner fetch: all 4*U loads are unconditional `global_load v, v_off32, s[base]`
                // (wave-uniform per-view base + 32-bit byte offset of a clamped, always valid texel).
                Corner c;
                float w0, w1, w2, w3;
                if (!strict && crec) {
                    const CornerRec cr = crec[p * V + v];
                    c.onw = cr.o[0]; c.one = cr.o[1]; c.osw = cr.o[2]; c.ose = cr.o[3];
                    w0 = cr.w[0]; w1 = cr.w[1]; w2 = cr.w[2]; w3 = cr.w[3];      // FOLD: folded in phase A
                } else {
                    c = corner_setup(m, r.gx, r.gy);
                    w0 = c.inw ? c.wnw : 0.0f; w1 = c.ine ? c.wne : 0.0f; w2 = c.isw ? c.wsw : 0.0f; w3 = c.ise ? c.wse : 0.0f;
                    if (FOLD && !strict) {
                        const float sc = fold_scale(r.wgt, cnt);
                        w0 = w0 * sc; w1 = w1 * sc; w2 = w2 * sc; w3 = w3 * sc;
                    }
                }
                const char *bv = reinterpret_cast<const char *>(data) + (int64_t)v * m.sv * ES;
                typename Raw<VW, HALF>::T a[U], b[U], d[U], e[U];      // as stored; widened to fp32 at the use
                if (BATCH) {
                    // cache-resident maps: all 4*U loads in flight before the first use (latency-bound regime)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int cv = min(c0 + u * lpp + g, cvec - 1);     // idle lanes re-read the last vector
                        const uint32_t co = (uint32_t)cv * (VW * ES);
                        a[u] = load_texel<VW, HALF>(bv + (c.onw + co));
                        b[u] = load_texel<VW, HALF>(bv + (c.one + co));
                        d[u] = load_texel<VW, HALF>(bv + (c.osw + co));
                        e[u] = load_texel<VW, HALF>(bv + (c.ose + co));
                    }
                }
                if (!strict) {
                    // Finite maps, finite coordinates, valid view: a zero WEIGHT is the zeros padding
                    // (x*0 == +-0 for finite x, and +-0 never changes the sums below), valid_v == 1.
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (!BATCH) {      // maps larger than the caches: a smaller in-flight footprint measured faster
                            const uint32_t co = (uint32_t)min(c0 + u * lpp + g, cvec - 1) * (VW * ES);
                            a[u] = load_texel<VW, HALF>(bv + (c.onw + co));
                            b[u] = load_texel<VW, HALF>(bv + (c.one + co));
                            d[u] = load_texel<VW, HALF>(bv + (c.osw + co));
                            e[u] = load_texel<VW, HALF>(bv + (c.ose + co));
                        }
                        if constexpr (HALF) {
                            // keep the fp16 -> fp32 conversions of one vector HERE: left alone, the optimiser hoists the
                            // conversions of all 4*U corner vectors above the strict / fast branch (both sides widen them):
                            // 96 extra VGPRs for U = 3, one wave per SIMD less
                            asm volatile("" : "+v"(a[u]), "+v"(b[u]), "+v"(d[u]), "+v"(e[u]));
                        }
                        if (FOLD) {                        // folded weights: four fma straight into the view sum
                            acc[u] = v_fma<VT>(widen<VW, HALF>(a[u]), w0, acc[u]);
                            acc[u] = v_fma<VT>(widen<VW, HALF>(b[u]), w1, acc[u]);
                            acc[u] = v_fma<VT>(widen<VW, HALF>(d[u]), w2, acc[u]);
                            acc[u] = v_fma<VT>(widen<VW, HALF>(e[u]), w3, acc[u]);
                        } else {
                            VT s = widen<VW, HALF>(a[u]) * w0; // ATen bilinear: fma chain nw,ne,sw,se
                            s = v_fma<VT>(widen<VW, HALF>(b[u]), w1, s);
                            s = v_fma<VT>(widen<VW, HALF>(d[u]), w2, s);
                            s = v_fma<VT>(widen<VW, HALF>(e[u]), w3, s);
                            acc[u] = acc[u] + s * r.wgt;       // fusion.py:385
                        }
                        if (!BATCH) __builtin_amdgcn_sched_barrier(0);   // keep the next vector's loads behind this use
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (!BATCH) {
                            const uint32_t co = (uint32_t)min(c0 + u * lpp + g, cvec - 1) * (VW * ES);
                            a[u] = load_texel<VW, HALF>(bv + (c.onw + co));
                            b[u] = load_texel<VW, HALF>(bv + (c.one + co));
                            d[u] = load_texel<VW, HALF>(bv + (c.osw + co));
                            e[u] = load_texel<VW, HALF>(bv + (c.ose + co));
                        }
                        const VT av = c.inw ? widen<VW, HALF>(a[u]) : (VT)0.0f, bvv = c.ine ? widen<VW, HALF>(b[u]) : (VT)0.0f;
                        const VT dv = c.isw ? widen<VW, HALF>(d[u]) : (VT)0.0f, ev = c.ise ? widen<VW, HALF>(e[u]) : (VT)0.0f;
                        VT s = av * c.wnw;
                        s = v_fma<VT>(bvv, c.wne, s);
                        s = v_fma<VT>(dv, c.wsw, s);
                        s = v_fma<VT>(ev, c.wse, s);
                        const int cv = c0 + u * lpp + g;
                        if (m.inter && cv < cvec)          // '<k>_inter' [V,n,C]  fusion.py:389
                            store_out<VT>(m.inter + ((int64_t)v * P.n + i) * m.C + cv * VW, s, P.store_policy == 1 ? 0 : P.store_policy);
                        acc[u] = acc[u] + (s * r.valid) * r.wgt;        // fusion.py:385
                    }
                }
            }
            // acc / (cnt + 1e-6)  (fusion.py:385).  All channels of a point share the divisor, so the IEEE
            // division is unrolled by hand with the reciprocal refined ONCE: the same rcp + fma sequence the
            // compiler expands `/` into (v_rcp, 2 fma on the reciprocal, then mul + 4 fma per quotient), minus
            // its operand pre-scaling and special-case fix-up, which cannot trigger here: the divisor lies in
            // [1, V+1) and on this path every numerator is finite.  Bit-identical quotients, 5 instead of 11
            // instructions per channel.  Strict points (non-finite operands possible) keep the full division.
            float rcp_d = 0.0f;
            if (!strict && !FOLD) rcp_d = refined_rcp(denom);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cv = c0 + u * lpp + g;
                if (cv < cvec) {
                    VT o;
                    if (all_invalid) {
                        o = (VT)0.0f;                                          // fusion.py:386
                    } else if (strict) {
                        o = strict_div<VT>(acc[u], denom);
                    } else if (FOLD) {
                        o = acc[u];                                            // the reciprocal is inside the weights
                    } else {
                        VT q = acc[u] * rcp_d;
                        q = v_fma<VT>(v_fma<VT>(q, -denom, acc[u]), rcp_d, q);
                        q = v_fma<VT>(v_fma<VT>(q, -denom, acc[u]), rcp_d, q);
                        o = q;
                    }
                    store_out<VT>(m.out + i * m.C + (int64_t)cv * VW, o, P.store_policy);
                }
            }
        }
    }
}

// ---- thin maps (the instance mask, colours: <= 4 lanes per point): the views in parallel across lanes ----------------
// gather_map walks the views one after the other, which for a map of one vector per lane is V dependent
// load round trips per point with next to nothing to overlap them.  Here a lane owns (point, view, vector): the
// 4 corner loads of all views of a point are in flight together, and the ordered sum over the views
// (((0 + t_0) + t_1) + ...) is rebuilt with V wave shuffles -- the same operands in the same order as the
// sequential loop (a skipped invalid view adds +0, which is exact, see gather_map), bit-identical results.
template <typename VT> __device__ __forceinline__ VT shfl_vec(VT x, int src);
template <> __device__ __forceinline__ float shfl_vec<float>(float x, int src) { return __shfl(x, src, 64); }
template <> __device__ __forceinline__ f32x2 shfl_vec<f32x2>(f32x2 x, int src)
{
    f32x2 r; r.x = __shfl(x.x, src, 64); r.y = __shfl(x.y, src, 64); return r;
}
template <> __device__ __forceinline__ f32x4 shfl_vec<f32x4>(f32x4 x, int src)
{
    f32x4 r; r.x = __shfl(x.x, src, 64); r.y = __shfl(x.y, src, 64); r.z = __shfl(x.z, src, 64); r.w = __shfl(x.w, src, 64);
    return r;
}

__device__ __forceinline__ bool thin_map(const MapDesc &m, const EvalParams &P, int VW)
{
    return m.unroll == 1 && m.lpp_log2 <= 2 && m.C / VW <= (1 << m.lpp_log2) && P.V >= 2 && P.V <= P.thin_max_views;
}

template <int VW>
__device__ __forceinline__ void gather_map_thin(const MapDesc &m, const EvalParams &P, const ViewRec *rec,
                                                const float *cnt_s, const uint32_t *flag_s,
                                                const uint32_t *idx_s, int64_t idx_base, int tile_n,
                                                int tid = threadIdx.x, int nth = kBlock)
{
    using VT = typename Vec<VW>::T;
    const int V = P.V;
    const int vp_log2 = V <= 2 ? 1 : V <= 4 ? 2 : 3;
    const int sh = m.lpp_log2 + vp_log2;               // lanes of one point: <= 32, inside one wave
    const int lpp = 1 << m.lpp_log2;
    const int g = tid & (lpp - 1);
    const int v = (tid >> m.lpp_log2) & ((1 << vp_log2) - 1);
    const int npts = nth >> sh;
    const int base = (tid & 63) & ~((1 << sh) - 1);
    const int cvec = m.C / VW;
    const uint32_t co = (uint32_t)min(g, cvec - 1) * (VW * 4);

    for (int p = tid >> sh; p < tile_n; p += npts) {
        const int64_t i = idx_base + idx_s[p];
        const float cnt = cnt_s[p];
        const bool strict = (flag_s[p] != 0u) || (m.inter != nullptr);
        VT t = (VT)0.0f;
        if (v < V) {
            const ViewRec r = rec[p * V + v];
            if (strict || r.valid != 0.0f) {
                const Corner c = corner_setup(m, r.gx, r.gy);
                const char *bv = reinterpret_cast<const char *>(m.data) + (int64_t)v * m.sv * 4;
                const VT a = load_texel<VW, false>(bv + (c.onw + co));
                const VT b = load_texel<VW, false>(bv + (c.one + co));
                const VT d = load_texel<VW, false>(bv + (c.osw + co));
                const VT e = load_texel<VW, false>(bv + (c.ose + co));
                if (!strict) {
                    const float w0 = c.inw ? c.wnw : 0.0f, w1 = c.ine ? c.wne : 0.0f;
                    const float w2 = c.isw ? c.wsw : 0.0f, w3 = c.ise ? c.wse : 0.0f;
                    VT s = a * w0;
                    s = v_fma<VT>(b, w1, s);
                    s = v_fma<VT>(d, w2, s);
                    s = v_fma<VT>(e, w3, s);
                    t = s * r.wgt;
                } else {
                    const VT av = c.inw ? a : (VT)0.0f, bvv = c.ine ? b : (VT)0.0f;
                    const VT dv = c.isw ? d : (VT)0.0f, ev = c.ise ? e : (VT)0.0f;
                    VT s = av * c.wnw;
                    s = v_fma<VT>(bvv, c.wne, s);
                    s = v_fma<VT>(dv, c.wsw, s);
                    s = v_fma<VT>(ev, c.wse, s);
                    if (m.inter && g < cvec)
                        store_out<VT>(m.inter + ((int64_t)v * P.n + i) * m.C + g * VW, s, P.store_policy == 1 ? 0 : P.store_policy);
                    t = (s * r.valid) * r.wgt;
                }
            }
        }
        VT acc = (VT)0.0f;
        for (int vv = 0; vv < V; ++vv) acc = acc + shfl_vec<VT>(t, base + (vv << m.lpp_log2) + g);
        if (v == 0 && g < cvec) {
            const float denom = cnt + 1e-6f;
            VT o;
            if (cnt == 0.0f) {
                o = (VT)0.0f;
            } else if (strict) {
                o = strict_div<VT>(acc, denom);
            } else {
                const float r0 = __builtin_amdgcn_rcpf(denom);
                const float rcp_d = fmaf(fmaf(-denom, r0, 1.0f), r0, r0);
                VT q = acc * rcp_d;
                q = v_fma<VT>(v_fma<VT>(q, -denom, acc), rcp_d, q);
                q = v_fma<VT>(v_fma<VT>(q, -denom, acc), rcp_d, q);
                o = q;
            }
            store_out<VT>(m.out + i * m.C + (int64_t)g * VW, o, P.store_policy);
        }
    }
}

// SMALL: the cell-run kernel keeps <= 96 VGPRs; its other maps (the mask, colours) are mapped to one vector per lane, batched
template <int VW, bool WIDE, bool SMALL = false>
__device__ __forceinline__ void gather_map_u(const MapDesc &m, const EvalParams &P, const ViewRec *rec,
                                             const float *cnt_s, const uint32_t *flag_s,
                                             const uint32_t *idx_s, int64_t idx_base, int tile_n, const CornerRec *crec)
{
    if (thin_map(m, P, VW)) {
        gather_map_thin<VW>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n);
        return;
    }
    if (m.fold) {                 // wide map: folded weights on the fast path
        switch (m.unroll) {
        case 1: gather_map<VW, 1, true, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case 2: if (!SMALL) gather_map<VW, 2, true, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case 3: if (!SMALL) gather_map<VW, 3, true, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case -1: if (!SMALL) gather_map<VW, 1, false, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case -2: if (!SMALL) gather_map<VW, 2, false, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        case -3: if (!SMALL) gather_map<VW, 3, false, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec); break;
        default:
            if (WIDE) gather_map<VW, 4, false, false, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);
            break;
        }
        return;
    }
    // thin maps (<= 256 bytes per texel): the host maps them to one batched vector per lane, reference order
    gather_map<VW, 1, true>(m, P, rec, cnt_s, flag_s, idx_s, idx_base, tile_n, crec);
}

// XCD-aware tile order: the dispatcher places workgroup b on XCD b%8 (observed, speed only).
// Giving XCD k the k-th contiguous eighth of the tiles keeps the texel footprints of the
// eight private L2s (4 MiB each) disjoint instead of replicated.  Bijective for any count.
__device__ __forceinline__ int64_t xcd_tile(int64_t b, int64_t nb)
{
    const int64_t q = nb / 8, r = nb % 8, xcd = b % 8, j = b / 8;
    const int64_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + j;
}

// MODE 0: Fusion.eval semantics; MODE 1: Fusion.eval_dist semantics (fusion.py:396-436).
// Query point i: from the caller's [n,3] array, or generated from the three axis arrays of a regular
// grid in the reference's order (create_init_grid, fusion.py:79-88: 'ij' meshgrid, z fastest).
__device__ __forceinline__ void fetch_point(const EvalParams &P, int64_t i, float &px, float &py, float &pz)
{
    if (P.grid_x) {
        if (P.n <= 0xffffffffLL) {          // (wave-uniform) two 32-bit divisions instead of two 64-bit ones: ~50 instead of ~300 instructions
            const uint32_t j = (uint32_t)i, nz = (uint32_t)P.grid_nz, ny = (uint32_t)P.grid_ny;
            const uint32_t ixy = j / nz, iz = j - ixy * nz, ix = ixy / ny;
            px = P.grid_x[ix];
            py = P.grid_y[ixy - ix * ny];
            pz = P.grid_z[iz];
        } else {
            const int64_t iz = i % P.grid_nz, ixy = i / P.grid_nz;
            px = P.grid_x[ixy / P.grid_ny];
            py = P.grid_y[ixy % P.grid_ny];
            pz = P.grid_z[iz];
        }
    } else {
        px = P.pts[i * 3 + 0]; py = P.pts[i * 3 + 1]; pz = P.pts[i * 3 + 2];
    }
}

// ---- lattice walk: blockIdx -> brick of points, closed form ----------------------------------------------------
// The points of a regular grid (create_init_grid, fusion.py:79-88) need no keys, no sort and no index array to be
// walked brick by brick: the flat walk position t decodes to a tile of walk_tx x walk_ty x walk_tz points through a
// three-level BLOCKED row-major order over the tile lattice -- 16^3-tile macro-bricks (32 k points with 2x2x2 tiles:
// the Infinity-Cache window all eight XCDs share), 8^3-tile sub-bricks (the contiguous eighth one XCD takes, its L2
// window), 4^3-tile mini-bricks, tiles row-major inside.  Blocks at the upper faces are clipped, not padded, so every
// workgroup has work and the count is exactly ceil(nx/tx)*ceil(ny/ty)*ceil(nz/tz).  Wave-uniform integer arithmetic.
struct TileBox {
    int ox, oy, oz;     // first point of the tile (lattice coordinates)
    int sx, sy, sz;     // clipped size in points
};

__device__ __forceinline__ void walk_level(uint32_t &rem, int b, int &ox, int &oy, int &oz, int &ex, int &ey, int &ez)
{
    // the box at (ox,oy,oz) with extents (ex,ey,ez) tiles is cut into b^3 blocks (upper ones clipped), row-major;
    // on return the box is the block holding position `rem`, and rem is the position inside it
    const uint32_t slab = (uint32_t)b * (uint32_t)ey * (uint32_t)ez;
    const uint32_t i = rem / slab;
    rem -= i * slab;
    const int bx = min(b, ex - (int)i * b);
    const uint32_t col = (uint32_t)bx * (uint32_t)b * (uint32_t)ez;
    const uint32_t j = rem / col;
    rem -= j * col;
    const int by = min(b, ey - (int)j * b);
    const uint32_t cell = (uint32_t)bx * (uint32_t)by * (uint32_t)b;
    const uint32_t k = rem / cell;
    rem -= k * cell;
    ox += (int)i * b; oy += (int)j * b; oz += (int)k * b;
    ex = bx; ey = by; ez = min(b, ez - (int)k * b);
}

__device__ __forceinline__ TileBox walk_tile(const EvalParams &P, int64_t t)
{
    int ex = (P.walk_nx + P.walk_tx - 1) / P.walk_tx, ey = (P.walk_ny + P.walk_ty - 1) / P.walk_ty,
        ez = (P.walk_nz + P.walk_tz - 1) / P.walk_tz;
    int ox = 0, oy = 0, oz = 0;
    uint32_t rem = (uint32_t)t;
    walk_level(rem, 16, ox, oy, oz, ex, ey, ez);
    walk_level(rem, 8, ox, oy, oz, ex, ey, ez);
    walk_level(rem, 4, ox, oy, oz, ex, ey, ez);
    const uint32_t yz = (uint32_t)ey * (uint32_t)ez;
    const uint32_t lx = rem / yz, r2 = rem - lx * yz;
    const uint32_t ly = r2 / (uint32_t)ez, lz = r2 - ly * (uint32_t)ez;
    TileBox tb;
    tb.ox = (ox + (int)lx) * P.walk_tx; tb.oy = (oy + (int)ly) * P.walk_ty; tb.oz = (oz + (int)lz) * P.walk_tz;
    tb.sx = min(P.walk_tx, P.walk_nx - tb.ox); tb.sy = min(P.walk_ty, P.walk_ny - tb.oy); tb.sz = min(P.walk_tz, P.walk_nz - tb.oz);
    return tb;
}

// flat index of the p-th point of a tile (z fastest inside the tile, like the lattice itself)
__device__ __forceinline__ int64_t walk_point(const EvalParams &P, const TileBox &tb, int p)
{
    const int yz = tb.sy * tb.sz;
    const int dx = p / yz, r = p - dx * yz;
    const int dy = r / tb.sz, dz = r - dy * tb.sz;
    return ((int64_t)(tb.ox + dx) * P.walk_ny + (tb.oy + dy)) * P.walk_nz + (tb.oz + dz);
}

// ---- fp16-STORED maps in the window kernel (round 5) --------------------------------------------------------------------------
// The pool then holds the texels as stored: a slice of 128 channels is 256 bytes, a lane's corner read is an 8-byte vector of four
// halves where the fp32 form reads 16 bytes (half the LDS bytes, the same instruction count, the same lane -> channel map, hence
// the same coalesced row stores), and the arithmetic is v_fma_mix_f32 -- the fp16 operand widened inside the fp32 fma, one
// rounding: bit for bit fma(float(h), w, acc), i.e. the fp32 kernel run on the widened map, at one instruction per channel (the
// compiler's own form of that expression is v_cvt + v_pk_fma: 12 instead of 8 instructions per eight channels).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <bool HALF> struct WinRaw { using T = f32x4; };
template <> struct WinRaw<true> { using T = f16x4; };

__device__ __forceinline__ void fma_mix4(f32x4 &acc, f16x4 r, float w)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 p = __builtin_bit_cast(u32x2, r);
    const uint32_t p0 = p.x, p1 = p.y;
    float a0 = acc.x, a1 = acc.y, a2 = acc.z, a3 = acc.w;
    // op_sel_hi[0] = 1: source 0 is fp16; op_sel[0] picks its high half
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a0) : "v"(p0), "v"(w));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a1) : "v"(p0), "v"(w));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a2) : "v"(p1), "v"(w));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a3) : "v"(p1), "v"(w));
    acc = f32x4{a0, a1, a2, a3};
}

// the same for a 16-byte vector of eight halves (the channel-sliced kernel's lane): two accumulators
__device__ __forceinline__ void fma_mix8(f32x4 &lo, f32x4 &hi, f16x8 r, float w)
{
    struct Pair { f16x4 a, b; };
    const Pair pr = __builtin_bit_cast(Pair, r);
    fma_mix4(lo, pr.a, w);
    fma_mix4(hi, pr.b, w);
}
// ---- the device-side gate of a cloud's two launches (fuse_window.hip: window_gate_probe_kernel) ------------------------
// Every fused entry point starts with it: a gated launch whose side lost returns at once (an ungated launch pays one scalar compare).
__device__ __forceinline__ bool gated_out(const EvalParams &P)
{
    if (!P.gate) return false;
    const uint32_t fit = __builtin_nontemporal_load(P.gate);
    return (fit >= P.gate_min) != (P.gate_want != 0);
}

}  // namespace d3f
