// d3f_device.h -- device helpers shared by the forward and backward field-query kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace d3f {

__device__ __forceinline__ float unnormalize(float g, int size)
{
    // grid_sample(align_corners=True): ((g+1)/2)*(size-1)
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}

__device__ __forceinline__ bool in_bounds(float x, float y, int fw, int fh)
{
    return (x > -1.0f) && (x < (float)fw) && (y > -1.0f) && (y < (float)fh);
}

// clang extended vectors: elementwise * + / are native, fma is explicit.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <int VW> struct Vec;
template <> struct Vec<8> { using T = f32x8; };     // fp16-stored maps: 8 channels = one 16-B load
template <> struct Vec<4> { using T = f32x4; };
template <> struct Vec<2> { using T = f32x2; };
template <> struct Vec<1> { using T = float; };

template <typename VT> __device__ __forceinline__ VT v_fma(VT a, float s, VT c) { return __builtin_elementwise_fma(a, (VT)s, c); }
template <> __device__ __forceinline__ float v_fma<float>(float a, float s, float c) { return fmaf(a, s, c); }

template <typename VT> __device__ __forceinline__ VT load_vec(const float *p) { return *reinterpret_cast<const VT *>(p); }
template <typename VT> __device__ __forceinline__ void store_vec(float *p, VT v) { *reinterpret_cast<VT *>(p) = v; }

// One channel vector of a texel: fp32 storage as is, fp16 storage (D3F_DTYPE_F16) widened to fp32 on load --
// all arithmetic stays fp32, so the result equals the fp32 path run on the widened map bit for bit.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Raw (as stored) channel vector and its widening: corner vectors stay in their stored form until they are used, so
// twelve in-flight fp16 loads cost 48 VGPRs, not 96.
template <int VW, bool HALF> struct Raw { using T = typename Vec<VW>::T; };
template <> struct Raw<8, true> { using T = f16x8; };
template <> struct Raw<1, true> { using T = _Float16; };
template <int VW, bool HALF> __device__ __forceinline__ typename Raw<VW, HALF>::T load_texel(const char *p)
{
    static_assert(!HALF || VW == 8 || VW == 1, "fp16-stored maps use 8-channel (16-B) or scalar lanes");
    return *reinterpret_cast<const typename Raw<VW, HALF>::T *>(p);
}
template <int VW, bool HALF> __device__ __forceinline__ typename Vec<VW>::T widen(typename Raw<VW, HALF>::T r)
{
    if constexpr (!HALF) return r;
    else if constexpr (VW == 8) return __builtin_convertvector(r, f32x8);
    else return (float)r;
}


template <typename VT> __device__ __forceinline__ float hsum(VT v);
template <> __device__ __forceinline__ float hsum<f32x8>(f32x8 v) { return ((v.s0 + v.s1) + (v.s2 + v.s3)) + ((v.s4 + v.s5) + (v.s6 + v.s7)); }
template <> __device__ __forceinline__ float hsum<f32x4>(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }
template <> __device__ __forceinline__ float hsum<f32x2>(f32x2 v) { return v.x + v.y; }
template <> __device__ __forceinline__ float hsum<float>(float v) { return v; }

// One view's projection of one point: the arithmetic contract of DESIGN.md section 2
// (reference fusion.py:45-55, 72-73), shared so that forward and backward agree bit for bit.
struct Proj {
    float gx, gy, zc, u, w;
    bool ok;
};

__device__ __forceinline__ Proj project_point(const float *M, float px, float py, float pz, float Wm1, float Hm1)
{
    Proj r;
    float xc = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    float yc = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    float zc = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    r.ok = !(fabsf(zc) < 1e-4f);                                    // fusion.py:52
    if (!r.ok) zc = 1e-3f;                                          // fusion.py:53
    r.u = xc / zc;                                                  // fusion.py:54
    r.w = yc / zc;
    r.gx = r.u / Wm1 * 2.0f - 1.0f;                                 // fusion.py:72
    r.gy = r.w / Hm1 * 2.0f - 1.0f;                                 // fusion.py:73
    r.zc = zc;
    return r;
}

// KRt = K @ pose (fusion.py:44): k-sequential, unfused, like the 3x3@3x4 bmm on the host
__device__ __forceinline__ void compute_krt(const float *K, const float *pose, int V, float *krt, int nthreads)
{
    for (int t = threadIdx.x; t < V * 12; t += nthreads) {
        const int v = t / 12, ij = t % 12, i = ij / 4, j = ij % 4;
        const float *Kv = K + v * 9, *Rv = pose + v * 12;
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) {
            const float pr = Kv[i * 3 + k] * Rv[k * 4 + j];
            acc = acc + pr;
        }
        krt[t] = acc;
    }
}

// nearest depth pixel with zeros padding (fusion.py:327-333)
// STRAIGHT: no branch around the load -- a pixel out of bounds reads pixel (0, 0) of the view and discards it -- so that the
// lookups of several views of one lane can be in flight together (the distance-only kernel, fuse_direct.hip); same value.
// TILED (with STRAIGHT): `depth` is the copy in tiles of 4 x 8 pixels, tw x th tiles per view (fuse_direct.hip: depth_tile_kernel)
template <bool STRAIGHT = false, bool TILED = false>
__device__ __forceinline__ float nearest_depth(const float *depth, int v, int H, int W, float gx, float gy, int tw = 0, int th = 0)
{
    const float rx = rintf(unnormalize(gx, W)), ry = rintf(unnormalize(gy, H));
    if constexpr (STRAIGHT) {
        const bool in = ((int)(rx > -1.0f) & (int)(rx < (float)W) & (int)(ry > -1.0f) & (int)(ry < (float)H)) != 0;      // in_bounds() without its short circuits
        const int ix = in ? (int)rx : 0, iy = in ? (int)ry : 0;
        float d;
        // (32-bit index: the host tiles only maps of fewer than 2^31 pixels -- an SGPR base + a 32-bit lane offset, no 64-bit adds)
        if constexpr (TILED) d = depth[(((uint32_t)(v * th + (iy >> 3)) * (uint32_t)tw + (uint32_t)(ix >> 2)) << 5) + (uint32_t)(((iy & 7) << 2) | (ix & 3))];
        else d = depth[((int64_t)v * H + iy) * W + ix];
        return in ? d : 0.0f;
    } else {
        float d = 0.0f;
        if (in_bounds(rx, ry, W, H)) d = depth[((int64_t)v * H + (int64_t)ry) * W + (int64_t)rx];
        return d;
    }
}

// One (point, view) of the forward: projection, nearest depth, validity, weight.
struct ViewOut {
    float gx, gy;
    float dist;     // clamp(d - zc, -mu, mu) for eval, d - zc for eval_dist
    float valid;    // 1.0f / 0.0f
};

// from the projection and the depth pixel to the view's distance / validity / weight
template <int MODE>
__device__ __forceinline__ ViewOut view_result(const Proj &pr, float d, float mu, float &wgt)
{
    float dist = d - pr.zc;                                                 // fusion.py:343
    bool valid;
    wgt = 1.0f;
    if (MODE == 0) {
        valid = (d > 0.0f) && pr.ok && (dist > -mu);                        // fusion.py:344
        float t = mu - fabsf(dist);                                         // fusion.py:347
        t = t > 0.0f ? 0.0f : t;
        wgt = expf(t / mu);
        float dc = dist < -mu ? -mu : dist;                                 // fusion.py:358
        dc = dc > mu ? mu : dc;
        dist = dc;
    } else {
        valid = (d > 0.0f) && pr.ok;                                        // fusion.py:426
    }
    ViewOut o;
    o.gx = pr.gx; o.gy = pr.gy; o.dist = dist; o.valid = valid ? 1.0f : 0.0f;
    return o;
}

template <int MODE>
__device__ __forceinline__ ViewOut eval_view(const float *depth, int H, int W, const float *M, int v, float px, float py,
                                             float pz, float Wm1, float Hm1, float mu, float &wgt)
{
    const Proj pr = project_point(M, px, py, pz, Wm1, Hm1);
    const float d = nearest_depth(depth, v, H, W, pr.gx, pr.gy);
    return view_result<MODE>(pr, d, mu, wgt);
}

// ---- the four IEEE divisions of a projection with the denominators' share hoisted (round 6; the distance-only kernel) ------------------
// hipcc expands x / y into v_div_scale x2, v_rcp, two fma refining the reciprocal, v_mul + three fma refining the quotient,
// v_div_fmas, v_div_fixup: 11 instructions, 17.4 ns per wave on a SIMD at eight waves (scripts/notebook/microbench/div_rate.hip).
// The two scalings and the fix-up are the identity unless an operand is zero, denormal, infinite, NaN or the exponents are extreme
// (ISA manual, V_DIV_SCALE / V_DIV_FMAS / V_DIV_FIXUP), and the reciprocal's part depends on the denominator alone: xc / zc and
// yc / zc share it, and for u / (W-1), w / (H-1) it is a constant of the launch.  What is left per quotient is v_mul + four fma in
// the SAME order on the SAME operands (6.2 ns) -- bit-identical whenever no scaling would have happened.  That is checked on the
// RESULTS (a quotient of the short form inside [2^-30, 2^30] and |zc| <= 2^60 imply operands for which every scaling is the identity:
// a zero, infinite, NaN or extreme operand drives the short form's quotient out of that range); a wave with one lane outside
// redoes the four divisions in the compiler's form.
struct DivConst {
    float d, r1;        // a denominator and the refined reciprocal the expansion derives from it
};
__device__ __forceinline__ DivConst div_const(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    DivConst c;
    c.d = d; c.r1 = fmaf(fmaf(-d, r, 1.0f), r, r);
    return c;
}
__device__ __forceinline__ float div_short(float n, const DivConst &c)
{
    float q = n * c.r1;
    float e = fmaf(-c.d, q, n);
    q = fmaf(e, c.r1, q);
    e = fmaf(-c.d, q, n);
    return fmaf(e, c.r1, q);
}
__device__ __forceinline__ bool div_result_plain(float q)
{
    const float a = fabsf(q);
    return __builtin_amdgcn_fmed3f(a, 0x1p-30f, 0x1p30f) == a;       // false for NaN
}

__device__ __forceinline__ Proj project_point_short(const float *M, float px, float py, float pz, const DivConst &cw, const DivConst &ch,
                                                    bool long_form = false)
{
    Proj r;
    float xc = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    float yc = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    float zc = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    r.ok = !(fabsf(zc) < 1e-4f);                                    // fusion.py:52
    if (!r.ok) zc = 1e-3f;                                          // fusion.py:53
    const DivConst cz = div_const(zc);
    r.u = div_short(xc, cz);                                        // fusion.py:54
    r.w = div_short(yc, cz);
    float a = div_short(r.u, cw), b = div_short(r.w, ch);           // fusion.py:72-73
    const bool plain = div_result_plain(a) && div_result_plain(b) && fabsf(zc) <= 0x1p60f;
    if (long_form || !__all(plain)) {         // (long_form: wave-uniform, the A/B switch of experiments builds)
        r.u = xc / zc; r.w = yc / zc;
        a = r.u / cw.d; b = r.w / ch.d;
    }
    r.gx = a * 2.0f - 1.0f;
    r.gy = b * 2.0f - 1.0f;
    r.zc = zc;
    return r;
}

// eval_view for the distance-only kernel: short divisions, no branch around the depth lookup
template <int MODE>
__device__ __forceinline__ ViewOut eval_view_straight(const float *depth, int H, int W, const float *M, int v, float px, float py,
                                                      float pz, const DivConst &cw, const DivConst &ch, float mu, float &wgt,
                                                      bool long_form = false)
{
    const Proj pr = project_point_short(M, px, py, pz, cw, ch, long_form);
    const float d = nearest_depth<true>(depth, v, H, W, pr.gx, pr.gy);
    return view_result<MODE>(pr, d, mu, wgt);
}

}  // namespace d3f
