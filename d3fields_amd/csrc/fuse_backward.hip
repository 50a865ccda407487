// fuse_backward.hip -- gradient of the fused field query w.r.t. the query points (gfx950).
//
// The reference gets this from autograd through Fusion.eval: rigid_tracking optimises an SE(3)
// pose by back-propagating  loss(eval(pts)['dino_feats'], eval(pts)['dist'])  to the points,
// 100 Adam steps per frame (fusion.py:1643-1665).  The differentiable paths of the forward
// (fusion.py:305-394) are, per view v of a point p:
//     zc      -> dist_v = d - zc            (d: nearest-mode sample, zero gradient)
//     dist_v  -> clamp(dist_v,-mu,mu)       -> 'dist'          (gradient 1 inside [-mu, mu])
//     dist_v  -> wgt_v = exp(min(mu-|dist_v|,0)/mu)            (gradient -sign*wgt/mu where mu-|dist| <= 0)
//     (u, w)  -> bilinear sample s_kv       (grid_sample backward: corner differences)
// valid_v, the view count and the 1e3 / 0 overrides carry no gradient.  One launch computes
//     grad_pts[p] = sum_v  dL/dxc * KRt_v[0,:3] + dL/dyc * KRt_v[1,:3] + dL/dzc * KRt_v[2,:3]
// with the same work decomposition as the forward kernel: phase A one lane per point (records in
// LDS), phase B 2^k lanes per point per map computing three dot products per view
// <g, s>, <g, ds/dix>, <g, ds/diy> with a shuffle reduction inside the lane group, phase C one
// lane per point combining the per-view scalars.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"

namespace d3f {

struct BwdRec {          // 32 B per (point, view)
    float gx, gy, wgt, valid;
    float zc, u, w, dist;   // dist: UNclamped d - zc
};

// HALF: the map is stored in fp16 (8 channels per 16-B load or scalar lanes), widened to fp32 on load like the forward
template <int VW, int U, bool HALF = false>
__device__ __forceinline__ void backward_map(const MapDesc &m, const float *__restrict__ gout, const BackwardParams &P,
                                             const BwdRec *rec, float *dots, int64_t tile_base, int tile_n)
{
    using VT = typename Vec<VW>::T;
    constexpr int ES = HALF ? 2 : 4;
    const int lpp = 1 << m.lpp_log2;
    const int g = threadIdx.x & (lpp - 1);
    const int grp = threadIdx.x >> m.lpp_log2;
    const int ngrp = kBlock >> m.lpp_log2;
    const int cvec = m.C / VW;
    const int V = P.V;
    const float sx = 0.5f * (float)(m.fw - 1), sy_ = 0.5f * (float)(m.fh - 1);   // d(ix)/d(gx), d(iy)/d(gy)

    for (int p0 = 0; p0 < tile_n; p0 += ngrp) {       // uniform trip count: shuffles below need whole groups
        const int p = p0 + grp;
        const bool live = p < tile_n;
        const int64_t i = tile_base + (live ? p : 0);
        for (int v = 0; v < V; ++v) {
            const BwdRec r = rec[(live ? p : 0) * V + v];
            const bool use = live && r.valid != 0.0f;
            float ds = 0.0f, dx = 0.0f, dy = 0.0f;
            if (use) {
                const float ix = unnormalize(r.gx, m.fw), iy = unnormalize(r.gy, m.fh);
                const float x0 = floorf(ix), y0 = floorf(iy);
                const float tx = ix - x0, ty = iy - y0;
                const float ex = 1.0f - tx, sy = 1.0f - ty;
                const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
                const bool inw = in_bounds(x0, y0, m.fw, m.fh), ine = in_bounds(x1, y0, m.fw, m.fh);
                const bool isw = in_bounds(x0, y1, m.fw, m.fh), ise = in_bounds(x1, y1, m.fw, m.fh);
                const int xi0 = (inw || isw) ? (int)x0 : 0, yi0 = (inw || ine) ? (int)y0 : 0;
                const int xi1 = (ine || ise) ? (int)x1 : 0, yi1 = (isw || ise) ? (int)y1 : 0;
                const char *bv = reinterpret_cast<const char *>(m.data) + (int64_t)v * m.sv * ES;
                const char *pnw = bv + ((int64_t)yi0 * m.sy + (int64_t)xi0 * m.sx) * ES;
                const char *pne = bv + ((int64_t)yi0 * m.sy + (int64_t)xi1 * m.sx) * ES;
                const char *psw = bv + ((int64_t)yi1 * m.sy + (int64_t)xi0 * m.sx) * ES;
                const char *pse = bv + ((int64_t)yi1 * m.sy + (int64_t)xi1 * m.sx) * ES;
                for (int c0 = 0; c0 < cvec; c0 += lpp * U) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int cv = c0 + u * lpp + g;
                        if (cv < cvec) {
                            const int co = cv * VW;
                            const int cb = co * ES;
                            const VT a = inw ? widen<VW, HALF>(load_texel<VW, HALF>(pnw + cb)) : (VT)0.0f;
                            const VT b = ine ? widen<VW, HALF>(load_texel<VW, HALF>(pne + cb)) : (VT)0.0f;
                            const VT d = isw ? widen<VW, HALF>(load_texel<VW, HALF>(psw + cb)) : (VT)0.0f;
                            const VT e = ise ? widen<VW, HALF>(load_texel<VW, HALF>(pse + cb)) : (VT)0.0f;
                            const VT go = load_vec<VT>(gout + i * m.C + co);
                            VT s = a * (sy * ex);
                            s = v_fma<VT>(b, sy * tx, s);
                            s = v_fma<VT>(d, ty * ex, s);
                            s = v_fma<VT>(e, ty * tx, s);
                            const VT dsx = (b - a) * sy + (e - d) * ty;      // ds/dix
                            const VT dsy = (d - a) * ex + (e - b) * tx;      // ds/diy
                            ds += hsum<VT>(go * s);
                            dx += hsum<VT>(go * dsx);
                            dy += hsum<VT>(go * dsy);
                        }
                    }
                }
            }
            for (int off = lpp >> 1; off > 0; off >>= 1) {
                ds += __shfl_xor(ds, off, 64);
                dx += __shfl_xor(dx, off, 64);
                dy += __shfl_xor(dy, off, 64);
            }
            if (use && g == 0) {
                float *o = dots + (p * V + v) * 3;
                o[0] += ds;
                o[1] += dx * sx;
                o[2] += dy * sy_;
            }
        }
    }
}

template <int VW, bool HALF = false>
__device__ __forceinline__ void backward_map_u(const MapDesc &m, const float *gout, const BackwardParams &P,
                                               const BwdRec *rec, float *dots, int64_t tile_base, int tile_n)
{
    switch (m.unroll < 0 ? -m.unroll : m.unroll) {
    case 1: backward_map<VW, 1, HALF>(m, gout, P, rec, dots, tile_base, tile_n); break;
    case 2: backward_map<VW, 2, HALF>(m, gout, P, rec, dots, tile_base, tile_n); break;
    case 3: backward_map<VW, 3, HALF>(m, gout, P, rec, dots, tile_base, tile_n); break;
    default:
        if (!HALF) backward_map<VW, 4, false>(m, gout, P, rec, dots, tile_base, tile_n);    // fp16 maps: <= 3 vectors
        break;
    }
}

// MODE 0: gradient of Fusion.eval; MODE 1: of Fusion.eval_dist (fusion.py:396-436: dist = mean over valid views of
// d - zc, validity without the -mu gate, no clamp, no weight, no channels).
template <int MODE>
__global__ __launch_bounds__(kBlock) void fused_eval_backward_kernel(const BackwardParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int V = P.V, TP = P.tile_pts;
    BwdRec *rec = reinterpret_cast<BwdRec *>(smem);                          // [TP*V]
    float *dots = reinterpret_cast<float *>(rec + (size_t)TP * V);           // [TP*V*3]
    float *cnt_s = dots + (size_t)TP * V * 3;                                // [TP]
    float *krt = cnt_s + TP;                                                 // [V*12]

    compute_krt(P.K, P.pose, V, krt, kBlock);
    for (int t = threadIdx.x; t < TP * V * 3; t += kBlock) dots[t] = 0.0f;
    __syncthreads();

    const int64_t tile_base = (int64_t)blockIdx.x * TP;
    const int tile_n = (int)min((int64_t)TP, P.n - tile_base);
    const float mu = P.mu;
    const float Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);

    // phase A: recompute the forward's per-view scalars
    for (int p = threadIdx.x; p < tile_n; p += kBlock) {
        const int64_t i = tile_base + p;
        const float px = P.pts[i * 3 + 0], py = P.pts[i * 3 + 1], pz = P.pts[i * 3 + 2];
        float cnt = 0.0f;
        for (int v = 0; v < V; ++v) {
            const Proj pr = project_point(krt + v * 12, px, py, pz, Wm1, Hm1);
            const float d = nearest_depth(P.depth, v, P.H, P.W, pr.gx, pr.gy);
            const float dist = d - pr.zc;
            const bool valid = (d > 0.0f) && pr.ok && (MODE == 1 || dist > -mu);
            float t = mu - fabsf(dist);
            t = t > 0.0f ? 0.0f : t;
            BwdRec r;
            r.gx = pr.gx; r.gy = pr.gy; r.wgt = expf(t / mu); r.valid = valid ? 1.0f : 0.0f;
            r.zc = pr.zc; r.u = pr.u; r.w = pr.w; r.dist = dist;
            rec[p * V + v] = r;
            cnt = cnt + r.valid;
        }
        cnt_s[p] = cnt;
    }
    __syncthreads();

    // phase B: per map, the three dot products per (point, view)
    for (int s = 0; s < P.n_maps; ++s) {
        const MapDesc &m = P.maps[s];
        const float *gout = P.grad_fused[s];
        if (gout && m.esize == 2) {
            if (m.vw == 8) backward_map_u<8, true>(m, gout, P, rec, dots, tile_base, tile_n);
            else backward_map_u<1, true>(m, gout, P, rec, dots, tile_base, tile_n);
        } else if (gout) {
            switch (m.vw) {
            case 4: backward_map_u<4>(m, gout, P, rec, dots, tile_base, tile_n); break;
            case 2: backward_map_u<2>(m, gout, P, rec, dots, tile_base, tile_n); break;
            default: backward_map_u<1>(m, gout, P, rec, dots, tile_base, tile_n); break;
            }
        }
        __syncthreads();      // the next map may assign a point to other lanes
    }

    // phase C: chain rule through weight, distance and projection; one lane per point
    for (int p = threadIdx.x; p < tile_n; p += kBlock) {
        const int64_t i = tile_base + p;
        const float cnt = cnt_s[p];
        const float inv = 1.0f / (cnt + 1e-6f);
        // eval: an all-invalid point has dist := 1e3 (constant); eval_dist: 0/(0+1e-6), no valid view contributes
        const float gd = (P.grad_dist && cnt != 0.0f) ? P.grad_dist[i] : 0.0f;
        float gxw = 0.0f, gyw = 0.0f, gzw = 0.0f;
        for (int v = 0; v < V; ++v) {
            const BwdRec r = rec[p * V + v];
            if (r.valid == 0.0f) continue;
            const float A = inv;                                   // valid_v / (cnt + 1e-6)
            const float *o = dots + (p * V + v) * 3;
            const float g_wgt = A * o[0];                          // dL/dwgt_v
            const float g_gx = A * r.wgt * o[1];                   // dL/dgx_v  (o[1] already times d ix / d gx)
            const float g_gy = A * r.wgt * o[2];
            // dL/ddist_v: clamp passes inside [-mu, mu]; weight passes where mu - |dist| <= 0
            float g_dist = (MODE == 1 || (r.dist >= -mu && r.dist <= mu)) ? gd * A : 0.0f;
            if (MODE == 0 && mu - fabsf(r.dist) <= 0.0f) {
                const float sgn = r.dist > 0.0f ? 1.0f : (r.dist < 0.0f ? -1.0f : 0.0f);
                g_dist += g_wgt * r.wgt * (-sgn) / mu;
            }
            const float g_u = g_gx * 2.0f / Wm1, g_w = g_gy * 2.0f / Hm1;
            const float g_xc = g_u / r.zc, g_yc = g_w / r.zc;
            // valid implies ok, so zc is the live camera depth here (fusion.py:52-53)
            const float g_zc = -(g_u * r.u + g_w * r.w) / r.zc - g_dist;
            const float *M = krt + v * 12;
            gxw += g_xc * M[0] + g_yc * M[4] + g_zc * M[8];
            gyw += g_xc * M[1] + g_yc * M[5] + g_zc * M[9];
            gzw += g_xc * M[2] + g_yc * M[6] + g_zc * M[10];
        }
        P.grad_pts[i * 3 + 0] = gxw;
        P.grad_pts[i * 3 + 1] = gyw;
        P.grad_pts[i * 3 + 2] = gzw;
    }
}

hipError_t launch_fused_backward(const BackwardParams &P, int mode, hipStream_t stream)
{
    if (P.n == 0) return hipSuccess;
    const int64_t ntiles = (P.n + P.tile_pts - 1) / P.tile_pts;
    const size_t lds = (size_t)P.tile_pts * P.V * (sizeof(BwdRec) + 12) + (size_t)P.tile_pts * 4 + (size_t)P.V * 48;
    if (mode == 0)
        hipLaunchKernelGGL(fused_eval_backward_kernel<0>, dim3((unsigned)ntiles), dim3(kBlock), lds, stream, P);
    else
        hipLaunchKernelGGL(fused_eval_backward_kernel<1>, dim3((unsigned)ntiles), dim3(kBlock), lds, stream, P);
    return hipGetLastError();
}

}  // namespace d3f
