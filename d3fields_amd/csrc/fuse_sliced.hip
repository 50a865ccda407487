// fuse_sliced.hip -- the CHANNEL-SLICED kernel of the fused field query (gfx950): dense wide maps, far larger than the caches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"
#include "fuse_common.h"


namespace d3f {

// ---- channel-sliced launch for a dense wide map on a lattice (the default there since round 2) -------------------------
// On maps much larger than the caches the kernel is bound by L2 misses, and an L2 holds only a ~512-point window of
// whole texels.  Here a workgroup handles 16 points (four 2x2x1 tiles of the brick walk; 32 when thin
// maps ride along) x ONE 512-byte channel slice of the wide map, and the slices of a 4096-point stretch of the walk (a
// "chunk") are units (chunk, slice) spread over the XCDs: the XCD that owns a unit has its 256 workgroups in flight
// together and its 4 MiB L2 sees a third of every texel, i.e. a ~3x larger window in points (read hit rate 58 -> 66 %,
// C2-dense 1.62 -> 1.52 ms).  The price: phase A (projection, depth test, weights, corner set-up) runs once per (point,
// slice).  512 bytes is the finest slice that pays: narrower ones repeat phase A more often and remove no fills (round 3
// counters: 61 M line fills at 512, 256 and 128 bytes alike, profiles/r3_stream -- the fills follow the points in flight,
// not the L2's capacity: DESIGN.md 5.6 e).  With C = 1024 there are eight slices, one per XCD: all eight L2s work on the
// same chunk (C4-dense lattice 13.05 -> 9.05 ms).  Clouds take tiles of consecutive points of their Hilbert order instead
// of lattice bricks.  Arithmetic per (point, view, channel) is that of gather_map (fast or strict form), so results are
// identical.
// HALF (round 5): the sliced map is stored in fp16 -- a lane's 16-byte vector is eight channels, widened inside v_fma_mix_f32
// (fma_mix8 below: the fp32 arithmetic on the widened map, bit for bit); 16 lanes x 8 channels = the same 128-channel slices.
template <int LG, int VC, bool HALF = false>   // lanes per point = 1 << LG: 8 (128-byte slices), 16 (256 B) or 32 (512 B)
__device__ __forceinline__ void fused_eval_sliced_body(const EvalParams &P)
{
    constexpr int LP = 1 << LG, PTS = kBlock / LP;
    constexpr int ES = HALF ? 2 : 4;
    const int TP = P.tile_pts;                 // 32 (four 2x2x2 walk tiles) or 64 (four 2x2x4 ones)
    extern __shared__ __align__(16) unsigned char smem[];
    const int V = P.V;
    ViewRec *rec = reinterpret_cast<ViewRec *>(smem);                       // same layout as fused_eval_body
    float *dcl_s = reinterpret_cast<float *>(rec + (size_t)TP * V);
    uint32_t *nfp_s = reinterpret_cast<uint32_t *>(dcl_s + (size_t)TP * V);
    float *cnt_s = reinterpret_cast<float *>(nfp_s + (size_t)TP * V);
    uint32_t *flag_s = reinterpret_cast<uint32_t *>(cnt_s + TP);
    uint32_t *idx_s = flag_s + TP;
    float *krt = reinterpret_cast<float *>(idx_s + TP);
    CornerRec *crec_s = reinterpret_cast<CornerRec *>(smem + P.crec_offset);
    __shared__ TileBox tbs[4];

    // unit (chunk, slice) -> XCD blockIdx % 8; the unit's workgroups are consecutive in that XCD's stream
    const int xcd = (int)(blockIdx.x & 7u);
    const int64_t j = (int64_t)(blockIdx.x >> 3);
    // sl_ilv > 1: the XCD's consecutive workgroups alternate between sl_ilv units (other slices of other chunks)
    const int64_t jj = j / P.sl_ilv;
    const int64_t unit = ((jj / P.sl_unit) * P.sl_ilv + (j - jj * P.sl_ilv)) * 8 + xcd;
    const int wg = (int)(jj % P.sl_unit);
    if (unit >= (int64_t)P.sl_chunks * P.sl_slices) return;
    const int64_t chunk = unit / P.sl_slices;
    const int slice = (int)(unit - chunk * P.sl_slices);
    const int64_t grp4 = chunk * P.sl_unit + wg;                              // group of four consecutive walk tiles,
    if (grp4 >= P.sl_groups) return;                                          // or of TP consecutive points of a cloud's order
    const bool lat = P.walk_nx > 0;
    if (lat && threadIdx.x < 4) {
        const int64_t t = grp4 * 4 + threadIdx.x;
        TileBox tb = {0, 0, 0, 0, 0, 0};
        if (t < P.sl_tiles) tb = walk_tile(P, t);
        tbs[threadIdx.x] = tb;
    }
    compute_krt(P.K, P.pose, V, krt, kBlock);
    __syncthreads();
    int start[5];
    start[0] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) start[q + 1] = lat ? start[q] + tbs[q].sx * tbs[q].sy * tbs[q].sz : 0;
    const int64_t cloud_base = grp4 * TP;
    const int tile_n = lat ? start[4] : (int)min((int64_t)TP, P.n - cloud_base);
    auto point_of = [&](int p) -> int64_t {
        if (!lat) return P.order ? min((int64_t)P.order[cloud_base + p], P.n - 1) : cloud_base + p;
        int q = 0;
        if (p >= start[1]) q = 1;
        if (p >= start[2]) q = 2;
        if (p >= start[3]) q = 3;
        return walk_point(P, tbs[q], p - start[q]);
    };
    const float mu = P.mu;
    const float Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);
    const MapDesc &m0 = P.maps[0];                                           // the sliced wide map

    // ---------------- phase A (as fused_eval_body: lane = (point, view), the views of a point adjacent) ----------------
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
            const int64_t i = point_of(p);
            ViewOut o;
            o.gx = 0.0f; o.gy = 0.0f; o.dist = 0.0f; o.valid = 0.0f;
            float wgt = 0.0f;
            uint32_t st = 0u;
            if (act) {
                float px, py, pz;
                fetch_point(P, i, px, py, pz);
                o = eval_view<0>(P.depth, P.H, P.W, krt + v * 12, v, px, py, pz, Wm1, Hm1, mu, wgt);
                if (!(isfinite(o.gx) && isfinite(o.gy) && isfinite(wgt))) st = 1u;
            }
            float dsum, cnt;
            uint32_t nonfinite;
            view_sums(V, base, o.dist * o.valid, o.valid, st, dsum, cnt, nonfinite);
            if (act) {
                ViewRec r;
                r.gx = o.gx; r.gy = o.gy; r.wgt = wgt; r.valid = o.valid;
                rec[p * V + v] = r;
                CornerRec cr;                           // invalid pair: texel 0 with zero weights (see phase B)
                cr.o[0] = cr.o[1] = cr.o[2] = cr.o[3] = 0u;
                cr.w[0] = cr.w[1] = cr.w[2] = cr.w[3] = 0.0f;
                if (o.valid != 0.0f) {
                    const Corner c = corner_setup(m0, o.gx, o.gy);
                    const float sc = fold_scale(wgt, cnt);          // the sliced map is wide: folded weights (fuse_common.h)
                    cr.o[0] = c.onw; cr.o[1] = c.one; cr.o[2] = c.osw; cr.o[3] = c.ose;
                    cr.w[0] = (c.inw ? c.wnw : 0.0f) * sc; cr.w[1] = (c.ine ? c.wne : 0.0f) * sc;
                    cr.w[2] = (c.isw ? c.wsw : 0.0f) * sc; cr.w[3] = (c.ise ? c.wse : 0.0f) * sc;
                }
                crec_s[p * V + v] = cr;
            }
            if (in && v == 0) {
                const bool all_invalid = (cnt == 0.0f);
                float dist_out = dsum / (cnt + 1e-6f);
                if (all_invalid) dist_out = 1e3f;
                if (slice == 0) {                                                    // one slice writes the per-point outputs
                    P.out_dist[i] = dist_out;
                    P.out_valid[i] = all_invalid ? 0 : 1;
                }
                cnt_s[p] = cnt;
                idx_s[p] = (uint32_t)i;
                flag_s[p] = (nonfinite || !finite_maps) ? 1u : 0u;
            }
        }
    }
    __syncthreads();

    // ---------------- phase B: the slice of the wide map, LP lanes per point ----------------
    {
        using VT = f32x4;
        using RT = typename std::conditional<HALF, f16x8, f32x4>::type;     // a lane's 16-byte vector as stored
        const MapDesc &m = m0;
        const int lg = threadIdx.x & (LP - 1), grp = threadIdx.x >> LG;
        const uint32_t co = (uint32_t)(slice * LP + lg) * 16u;               // byte offset of this lane's vector in a texel
        const char *__restrict__ data = reinterpret_cast<const char *>(m.data);
        auto raw = [&](const char *bv, uint32_t off) -> RT { return *reinterpret_cast<const RT *>(bv + (off + co)); };
        for (int p = grp; p < tile_n; p += PTS) {
            const int64_t i = idx_s[p];
            const float cnt = cnt_s[p];
            const float denom = cnt + 1e-6f;
            const bool strict = flag_s[p] != 0u;
            VT acc = (VT)0.0f, acc2 = (VT)0.0f;                              // (acc2: channels 4..7 of a fp16 vector)
            auto accumulate = [&](RT t, float w) {
                if constexpr (HALF) fma_mix8(acc, acc2, t, w);
                else acc = v_fma<VT>(t, w, acc);
            };
            if (!strict) {
                // fast path, branch-free: phase A left an all-zero corner record for an invalid (point, view), so its
                // loads hit texel 0 of the view and its term is +-0 -- adding it changes no bit.  The corner
                // loads of VC views are in flight together, then the views are consumed in view order with the folded
                // weights: four fma per view straight into the sum.
                int v0 = 0;
                for (; v0 + VC <= V; v0 += VC) {
                    RT a[VC], b[VC], d[VC], e[VC];
                    f32x4 w[VC];
#pragma unroll
                    for (int q = 0; q < VC; ++q) {
                        const CornerRec cr = crec_s[p * V + v0 + q];
                        const char *bv = data + (int64_t)(v0 + q) * m.sv * ES;
                        a[q] = raw(bv, cr.o[0]); b[q] = raw(bv, cr.o[1]); d[q] = raw(bv, cr.o[2]); e[q] = raw(bv, cr.o[3]);
                        w[q] = f32x4{cr.w[0], cr.w[1], cr.w[2], cr.w[3]};
                    }
#pragma unroll
                    for (int q = 0; q < VC; ++q) {
                        accumulate(a[q], w[q].x);
                        accumulate(b[q], w[q].y);
                        accumulate(d[q], w[q].z);
                        accumulate(e[q], w[q].w);
                    }
                }
                for (; v0 < V; ++v0) {
                    const CornerRec cr = crec_s[p * V + v0];
                    const char *bv = data + (int64_t)v0 * m.sv * ES;
                    const RT a = raw(bv, cr.o[0]), b = raw(bv, cr.o[1]), d = raw(bv, cr.o[2]), e = raw(bv, cr.o[3]);
                    accumulate(a, cr.w[0]);
                    accumulate(b, cr.w[1]);
                    accumulate(d, cr.w[2]);
                    accumulate(e, cr.w[3]);
                }
            } else {
                for (int v = 0; v < V; ++v) {
                    const ViewRec r = rec[p * V + v];
                    const char *bv = data + (int64_t)v * m.sv * ES;
                    const Corner c = corner_setup(m, r.gx, r.gy);
                    const RT ra = raw(bv, c.onw), rb = raw(bv, c.one), rd = raw(bv, c.osw), re = raw(bv, c.ose);
#pragma unroll
                    for (int hh = 0; hh < (HALF ? 2 : 1); ++hh) {
                        VT a, b, d, e;
                        if constexpr (HALF) {
                            const f32x8 wa = __builtin_convertvector(ra, f32x8), wb = __builtin_convertvector(rb, f32x8);
                            const f32x8 wd = __builtin_convertvector(rd, f32x8), we = __builtin_convertvector(re, f32x8);
                            a = hh ? wa.hi : wa.lo; b = hh ? wb.hi : wb.lo; d = hh ? wd.hi : wd.lo; e = hh ? we.hi : we.lo;
                        } else {
                            a = ra; b = rb; d = rd; e = re;
                        }
                        const VT av = c.inw ? a : (VT)0.0f, bvv = c.ine ? b : (VT)0.0f, dv = c.isw ? d : (VT)0.0f, ev = c.ise ? e : (VT)0.0f;
                        VT s_ = av * c.wnw;
                        s_ = v_fma<VT>(bvv, c.wne, s_);
                        s_ = v_fma<VT>(dv, c.wsw, s_);
                        s_ = v_fma<VT>(ev, c.wse, s_);
                        if (hh) acc2 = acc2 + (s_ * r.valid) * r.wgt;
                        else acc = acc + (s_ * r.valid) * r.wgt;
                    }
                }
            }
            VT o = acc, o2 = acc2;              // fast path: the weights carry 1/(cnt + 1e-6); no valid view: every weight is 0
            if (strict) {
                o = (VT)0.0f; o2 = (VT)0.0f;    // fusion.py:386
                if (cnt != 0.0f) { o = strict_div<VT>(acc, denom); if constexpr (HALF) o2 = strict_div<VT>(acc2, denom); }
            }
            if constexpr (HALF) {
                store_out<VT>(m.out + i * m.C + (co >> 1), o, P.store_policy);
                store_out<VT>(m.out + i * m.C + (co >> 1) + 4, o2, P.store_policy);
            } else {
                store_out<VT>(m.out + i * m.C + (co >> 2), o, P.store_policy);
            }
        }
    }
    // the other (thin) maps of the launch ride along with slice 0
    if (slice == 0)
        for (int s = 1; s < P.n_maps; ++s) {
            const MapDesc &m = P.maps[s];
            switch (m.vw) {
            case 4: gather_map_u<4, false, true>(m, P, rec, cnt_s, flag_s, idx_s, 0, tile_n, nullptr); break;
            case 2: gather_map_u<2, false, true>(m, P, rec, cnt_s, flag_s, idx_s, 0, tile_n, nullptr); break;
            default: gather_map_u<1, false, true>(m, P, rec, cnt_s, flag_s, idx_s, 0, tile_n, nullptr); break;
            }
        }
}

template <int LG, int VC, int WAVES, bool HALF = false>
__global__ __launch_bounds__(kBlock, WAVES) void fused_eval_sliced_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    fused_eval_sliced_body<LG, VC, HALF>(P);
}

hipError_t launch_sliced(const EvalParams &P, hipStream_t stream)
{
    dim3 block(kBlock);
    const int64_t units = (int64_t)P.sl_chunks * P.sl_slices;
    const int64_t wgs = ((units + 7) / 8 + P.sl_ilv - 1) / P.sl_ilv * P.sl_ilv * 8 * P.sl_unit;
    const size_t lds_s = (size_t)P.crec_offset + (size_t)P.tile_pts * P.V * 32 + (size_t)P.lds_pad;
    const dim3 gs((unsigned)wgs);
    if (P.maps[0].esize == 2) {                 // fp16-stored map: 16 lanes x 8 channels = 128-channel (256-byte) slices
        if (P.sl_lg != 4 || P.sl_vc != 2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((fused_eval_sliced_kernel<4, 2, 7, true>), gs, block, lds_s, stream, P);
        return hipGetLastError();
    }
    if (P.sl_lg == 5 && P.sl_vc == 2) hipLaunchKernelGGL((fused_eval_sliced_kernel<5, 2, 7>), gs, block, lds_s, stream, P);
#ifndef D3F_EXPERIMENTS
    else return hipErrorInvalidValue;          // (other slice widths / views in flight: experiments builds only)
#else
    else if (P.sl_lg == 5) hipLaunchKernelGGL((fused_eval_sliced_kernel<5, 4, 5>), gs, block, lds_s, stream, P);
    else if (P.sl_lg == 4 && P.sl_vc == 2) hipLaunchKernelGGL((fused_eval_sliced_kernel<4, 2, 7>), gs, block, lds_s, stream, P);
    else if (P.sl_lg == 4 && P.sl_vc == 1) hipLaunchKernelGGL((fused_eval_sliced_kernel<4, 1, 8>), gs, block, lds_s, stream, P);
    else if (P.sl_lg == 4) hipLaunchKernelGGL((fused_eval_sliced_kernel<4, 4, 5>), gs, block, lds_s, stream, P);
    else if (P.sl_vc == 2) hipLaunchKernelGGL((fused_eval_sliced_kernel<3, 2, 7>), gs, block, lds_s, stream, P);
    else hipLaunchKernelGGL((fused_eval_sliced_kernel<3, 4, 5>), gs, block, lds_s, stream, P);
#endif
    return hipGetLastError();
}

}  // namespace d3f
