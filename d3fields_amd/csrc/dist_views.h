// dist_views.h -- the distance part of a field query for ONE point and up to four views at a time, as the distance-only kernel
// (fuse_direct.hip: fused_eval_dist_kernel) and the keypoint pre-filter (grid_kernels.hip: grid_shell_flag_fast_kernel)
// run it: KRt in wave-uniform registers, the views stage by stage, the IEEE divisions in their short form (d3f_device.h), the depth
// pixels from the caller's row-major maps or from the copy tiled 4 x 8 pixels per cache line.
#pragma once
#include "d3f_internal.h"
#include "d3f_device.h"

namespace d3f {

// the NVQ views of one batch for one point, STAGE BY STAGE across the views: a wave's own instruction stream then carries NVQ
// independent chains (the divisions are serial fma chains: with one view after the other the counters showed the waves
// issue-stalled 54 % of the time with the VALU 76 % busy), the depth lookups leave together, and the short divisions'
// check (d3f_device.h: project_point_short) is ONE wave-uniform branch per point.  Same operations on the same operands as
// eval_view_straight view by view; the sums in view order.
template <int MODE, int NVQ, bool TILED>
__device__ __forceinline__ void dist_views(const EvalParams &P, const float (&M)[4][12], int v0, float px, float py, float pz,
                                           const DivConst &cw, const DivConst &ch, float mu, float &ds, float &cn)
{
    float xc[NVQ], yc[NVQ], zc[NVQ], a[NVQ], b[NVQ];
    bool ok[NVQ];
#pragma unroll
    for (int q = 0; q < NVQ; ++q) {
        xc[q] = ((M[q][0] * px + M[q][1] * py) + M[q][2] * pz) + M[q][3] * 1.0f;
        yc[q] = ((M[q][4] * px + M[q][5] * py) + M[q][6] * pz) + M[q][7] * 1.0f;
        zc[q] = ((M[q][8] * px + M[q][9] * py) + M[q][10] * pz) + M[q][11] * 1.0f;
        ok[q] = !(fabsf(zc[q]) < 1e-4f);                                    // fusion.py:52
        if (!ok[q]) zc[q] = 1e-3f;                                          // fusion.py:53
    }
    DivConst cz[NVQ];
#pragma unroll
    for (int q = 0; q < NVQ; ++q) cz[q] = div_const(zc[q]);
    float u[NVQ], w[NVQ];
#pragma unroll
    for (int q = 0; q < NVQ; ++q) { u[q] = div_short(xc[q], cz[q]); w[q] = div_short(yc[q], cz[q]); }      // fusion.py:54
    int plain = 1;              // (int, bitwise: no short-circuit branches)
#pragma unroll
    for (int q = 0; q < NVQ; ++q) {
        a[q] = div_short(u[q], cw); b[q] = div_short(w[q], ch);                                          // fusion.py:72-73
        plain &= (int)div_result_plain(a[q]) & (int)div_result_plain(b[q]) & (int)(fabsf(zc[q]) <= 0x1p60f);
    }
#ifdef D3F_EXPERIMENTS
    const bool long_form = (P.dist_variant & 16) != 0;        // wave-uniform A/B switch (a run-time test in this loop costs: product builds have none)
#else
    constexpr bool long_form = false;
#endif
    if (__builtin_expect(long_form || !__all(plain), 0)) {
#pragma unroll
        for (int q = 0; q < NVQ; ++q) {
            const float uu = xc[q] / zc[q], ww = yc[q] / zc[q];
            a[q] = uu / cw.d; b[q] = ww / ch.d;
        }
    }
    float d[NVQ];
#pragma unroll
    for (int q = 0; q < NVQ; ++q) {
        const float gx = a[q] * 2.0f - 1.0f, gy = b[q] * 2.0f - 1.0f;
        if constexpr (TILED) d[q] = nearest_depth<true, true>(P.depth_tiled, v0 + q, P.H, P.W, gx, gy, P.depth_tw, P.depth_th);
        else d[q] = nearest_depth<true>(P.depth, v0 + q, P.H, P.W, gx, gy);
    }
#pragma unroll
    for (int q = 0; q < NVQ; ++q) {
        Proj pr;
        pr.zc = zc[q]; pr.ok = ok[q]; pr.gx = 0.0f; pr.gy = 0.0f; pr.u = 0.0f; pr.w = 0.0f;
        float wgt;
        const ViewOut o = view_result<MODE>(pr, d[q], mu, wgt);
        ds = ds + o.dist * o.valid;                             // fusion.py:364
        cn = cn + o.valid;
    }
}

// KRt of views v0 .. v0+3 (fusion.py:44): entry t = lane, as compute_krt computes it ...
__device__ __forceinline__ float dist_krt_lane(const EvalParams &P, int v0)
{
    const int lane = threadIdx.x & 63;
    const int v = v0 + lane / 12, ij = lane % 12, i = ij / 4, j = ij % 4;
    float kr = 0.0f;
    if (lane < 48 && v < P.V) {
        const float *Kv = P.K + v * 9, *Rv = P.pose + v * 12;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float pr = Kv[i * 3 + k] * Rv[k * 4 + j];
            acc = acc + pr;
        }
        kr = acc;
    }
    return kr;
}
// ... and into wave-uniform registers (SGPRs)
__device__ __forceinline__ void dist_krt_uniform(float kr, float (&M)[4][12])
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j2 = 0; j2 < 12; ++j2)
            M[q][j2] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(kr), q * 12 + j2));
}

}  // namespace d3f
