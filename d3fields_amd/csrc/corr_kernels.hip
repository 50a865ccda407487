// corr_kernels.hip -- descriptor similarity of utils/corr_utils.py for gfx950.
//
// The reference forms the full difference tensor ([B1,B2,C] for compute_similarity_tensor_multi,
// corr_utils.py:80-83, with a 5000-row retry when that does not fit, :84-94) and reduces it.
// Here the difference never leaves registers: distances use the DIRECT form sum((a-b)^2) -- not
// |a|^2+|b|^2-2ab, whose cancellation breaks 1e-5 relative parity for near-identical
// descriptors -- so this is fp32 VALU work staged through LDS, not an MFMA contraction.
#include "d3f_internal.h"

namespace d3f {

// ---- feature map vs one target (corr_utils.py:4-61) -------------------------------------
// G = 2^k lanes cooperate on one descriptor; lane g takes channels g, g+G, ...  The host picks
// G = 64 when channels are contiguous (stride_c == 1: consecutive lanes read consecutive
// floats) and G = 1 when positions are contiguous (stride_i == 1: consecutive lanes read
// consecutive positions of the same channel).  Either way every wave load is one segment.
__global__ __launch_bounds__(kBlock) void dist_to_target_kernel(const float *__restrict__ src, int64_t total,
                                                               int64_t inner, int C, int64_t sb, int64_t si,
                                                               int64_t sc, const float *__restrict__ tgt,
                                                               int dist_type, int g_log2, float *__restrict__ out)
{
    const int G = 1 << g_log2;
    const int g = threadIdx.x & (G - 1);
    const int64_t pos = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> g_log2;
    const bool live = pos < total;
    const int64_t b = live ? pos / inner : 0, j = live ? pos - (pos / inner) * inner : 0;
    const float *a = src + b * sb + j * si;
    float acc = 0.0f;
    if (live)
        for (int c = g; c < C; c += G) {
            const float d = a[(int64_t)c * sc] - tgt[c];
            acc = fmaf(d, d, acc);
        }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && g == 0) out[pos] = dist_type == D3F_DIST_L2 ? sqrtf(acc) : acc;
}

hipError_t launch_dist_to_target(const float *src, int64_t B, int64_t inner, int C, int64_t sb, int64_t si,
                                 int64_t sc, const float *tgt, int dist_type, float *out, hipStream_t s)
{
    const int64_t total = B * inner;
    if (total == 0) return hipSuccess;
    int g_log2 = 0;
    if (sc == 1) while ((1 << g_log2) < C && g_log2 < 6) ++g_log2;
    const int64_t threads = total << g_log2;
    hipLaunchKernelGGL(dist_to_target_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       src, total, inner, C, sb, si, sc, tgt, dist_type, g_log2, out);
    return hipGetLastError();
}

// ---- pairwise distances src[B1,C] x tgt[B2,C] -> out[B1,B2] (corr_utils.py:78-83) -------
// 64x64 output tile per workgroup, 4x4 outputs per lane, channels staged 32 at a time through
// LDS (row stride 33 floats: the 16 distinct tgt rows a wave reads fall on >= 8 banks).
constexpr int kPT = 64, kPK = 32;

__global__ __launch_bounds__(kBlock) void pairwise_dist_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ tgt, int64_t B1, int64_t B2,
                                                              int C, int dist_type, float *__restrict__ out)
{
    __shared__ float As[kPT][kPK + 1];
    __shared__ float Bs[kPT][kPK + 1];
    const int64_t i0 = (int64_t)blockIdx.y * kPT, j0 = (int64_t)blockIdx.x * kPT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;

    for (int k0 = 0; k0 < C; k0 += kPK) {
        // 64 rows x 32 channels per operand = 2048 floats, 8 per lane, channel fastest
        for (int e = threadIdx.x; e < kPT * kPK; e += kBlock) {
            const int r = e / kPK, kk = e % kPK;
            const bool kin = (k0 + kk) < C;
            As[r][kk] = (kin && (i0 + r) < B1) ? src[(i0 + r) * C + k0 + kk] : 0.0f;
            Bs[r][kk] = (kin && (j0 + r) < B2) ? tgt[(j0 + r) * C + k0 + kk] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < kPK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = As[ty * 4 + q][kk];
                b[q] = Bs[tx * 4 + q][kk];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float d = a[q] - b[w];
                    acc[q][w] = fmaf(d, d, acc[q][w]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + ty * 4 + q;
        if (i >= B1) continue;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int64_t j = j0 + tx * 4 + w;
            if (j < B2) out[i * B2 + j] = dist_type == D3F_DIST_L2 ? sqrtf(acc[q][w]) : acc[q][w];
        }
    }
}

hipError_t launch_pairwise_dist(const float *src, const float *tgt, int64_t B1, int64_t B2, int C, int dist_type,
                                float *out, hipStream_t s)
{
    if (B1 == 0 || B2 == 0) return hipSuccess;
    dim3 grid((unsigned)((B2 + kPT - 1) / kPT), (unsigned)((B1 + kPT - 1) / kPT));
    hipLaunchKernelGGL(pairwise_dist_kernel, grid, dim3(kBlock), 0, s, src, tgt, B1, B2, C, dist_type, out);
    return hipGetLastError();
}

// ---- exp(-d*scale)  (corr_utils.py:17) ---------------------------------------------------
__global__ __launch_bounds__(kBlock) void exp_neg_scale_kernel(float *__restrict__ x, int64_t n, float scale)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k < n) x[k] = expf(-x[k] * scale);
}

hipError_t launch_exp_neg_scale(float *x, int64_t n, float scale, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(exp_neg_scale_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, x, n,
                       scale);
    return hipGetLastError();
}

// ---- softmax(-d*scale, dim=0) of a row-major [rows, cols] matrix (corr_utils.py:39,102) --
// dim 0 is the LONG axis (B1, up to 1e5+), cols the short one, and consecutive columns are
// contiguous, so lanes map to columns and rows are split into chunks of 256 across workgroups:
//   pass 1  per (row chunk, column): online max / sum-exp / first-argmax  -> ws[chunk][col]
//   pass 2  per column: merge the chunk statistics in chunk order        -> ws[nchunks][col]
//   pass 3  elementwise normalise
__global__ __launch_bounds__(kBlock) void softmax_stats_kernel(const float *__restrict__ x, int64_t rows,
                                                              int64_t cols, float scale, ColStat *__restrict__ ws)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * kSoftmaxRowsPerBlock;
    const int64_t r1 = min(rows, r0 + kSoftmaxRowsPerBlock);
    float m = -INFINITY, s = 0.0f;
    int64_t arg = r0;
    for (int64_t i = r0; i < r1; ++i) {
        const float v = -x[i * cols + j] * scale;
        if (v > m) {
            s = s * expf(m - v) + 1.0f;
            m = v;
            arg = i;
        } else {
            s += expf(v - m);
        }
    }
    ColStat o;
    o.m = m; o.s = s; o.arg = arg;
    ws[(int64_t)blockIdx.y * cols + j] = o;
}

__global__ __launch_bounds__(kBlock) void softmax_merge_kernel(ColStat *__restrict__ ws, int64_t nchunks,
                                                              int64_t cols, int64_t *__restrict__ argmax_out)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cols) return;
    float m = -INFINITY;
    int64_t arg = 0;
    for (int64_t c = 0; c < nchunks; ++c) {
        const ColStat t = ws[c * cols + j];
        if (t.m > m) { m = t.m; arg = t.arg; }
    }
    float s = 0.0f;
    for (int64_t c = 0; c < nchunks; ++c) {
        const ColStat t = ws[c * cols + j];
        s += t.s * expf(t.m - m);
    }
    ColStat o;
    o.m = m; o.s = s; o.arg = arg;
    ws[nchunks * cols + j] = o;
    if (argmax_out) argmax_out[j] = arg;
}

__global__ __launch_bounds__(kBlock) void softmax_apply_kernel(float *__restrict__ x, int64_t total, int64_t cols,
                                                              float scale, const ColStat *__restrict__ fin)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= total) return;
    const ColStat t = fin[k % cols];
    x[k] = expf(-x[k] * scale - t.m) / t.s;
}

static hipError_t softmax_impl(float *x, int64_t rows, int64_t cols, float scale, int64_t *argmax_out, ColStat *ws,
                               bool normalise, hipStream_t s)
{
    if (rows == 0 || cols == 0) return hipSuccess;
    const int64_t nchunks = (rows + kSoftmaxRowsPerBlock - 1) / kSoftmaxRowsPerBlock;
    const unsigned gx = (unsigned)((cols + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(softmax_stats_kernel, dim3(gx, (unsigned)nchunks), dim3(kBlock), 0, s, x, rows, cols, scale, ws);
    hipLaunchKernelGGL(softmax_merge_kernel, dim3(gx), dim3(kBlock), 0, s, ws, nchunks, cols, argmax_out);
    if (normalise) {
        const int64_t total = rows * cols;
        hipLaunchKernelGGL(softmax_apply_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, x,
                           total, cols, scale, ws + nchunks * cols);
    }
    return hipGetLastError();
}

hipError_t launch_softmax_dim0(float *x, int64_t rows, int64_t cols, float scale, int64_t *argmax_out, ColStat *ws,
                               hipStream_t s)
{
    return softmax_impl(x, rows, cols, scale, argmax_out, ws, true, s);
}

hipError_t launch_argmin_dim0(const float *x, int64_t rows, int64_t cols, int64_t *arg_out, ColStat *ws,
                              hipStream_t s)
{
    return softmax_impl(const_cast<float *>(x), rows, cols, 1.0f, arg_out, ws, false, s);
}

}  // namespace d3f
