// corr_kernels.hip -- descriptor similarity of utils/corr_utils.py for gfx950.
//
// The reference forms the full difference tensor ([B1,B2,C] for compute_similarity_tensor_multi,
// corr_utils.py:80-83, with a 5000-row retry when that does not fit, :84-94) and reduces it.
// Here the difference never leaves registers: distances use the DIRECT form sum((a-b)^2) -- not
// |a|^2+|b|^2-2ab, whose cancellation breaks 1e-5 relative parity for near-identical
// descriptors -- so this is fp32 VALU work staged through LDS, not an MFMA contraction.
#include <type_traits>

#include "d3f_internal.h"
#include "d3f_device.h"

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
// 64x64 output tile per workgroup, 4x4 outputs per lane, 32 channels per LDS stage.
//  * LDS holds the stage as float4 columns [k/4][row]: lane (tx,ty) owns rows ty+16q and columns tx+16w, so
//    the 16 lanes of a ds_read_b128 group read 16 consecutive float4 (conflict-free) or one address
//    (broadcast); 8 b128 reads feed 64 (difference, square-accumulate) pairs.
//  * the arithmetic is packed: d = a - b as two v_pk_add_f32, acc2 += d*d as two v_pk_fma_f32 on a float2
//    accumulator (even / odd channels), i.e. one VALU instruction per (pair, channel).  The target stage is
//    stored NEGATED in LDS: the compiler has no packed form for a vector fsub (it emitted 64 scalar v_sub_f32
//    per float4 step), while a + (-b) is a v_pk_add_f32 and the same IEEE result bit for bit.
// fp32 VALU bound: 3*B1*B2*C flop (sub, mul, add).
constexpr int kPT = 64, kPK = 32;
static_assert(kPT == kSoftmaxRowsPerBlock, "the fused column statistics use one row tile per statistics chunk");

// merges two running (max, sum-exp, first-argmax) triples of the same column
__device__ __forceinline__ void merge_stat(float &m, float &s, int64_t &arg, float m2, float s2, int64_t arg2)
{
    const bool take2 = (m2 > m) || (m2 == m && arg2 < arg);
    const float M = fmaxf(m, m2);
    const float sa = (m == -INFINITY) ? 0.0f : s * expf(m - M);
    const float sb = (m2 == -INFINITY) ? 0.0f : s2 * expf(m2 - M);
    s = sa + sb;
    m = M;
    arg = take2 ? arg2 : arg;
}


// FAST (host-checked: 16-B aligned operands, C a multiple of the 32-channel stage, operands below 2^30 floats):
// every lane's two float4 loads per operand and stage go through a uniform base + a loop-invariant 32-bit offset,
// with rows beyond the matrix clamped to the last row (their results are never stored), so the fetch costs no
// per-element bounds or address arithmetic -- the generic path spent 22 % of its VALU instructions there.
// NW: 16-column groups of the tile that hold live columns (4 everywhere but in the last column tile of a B2 that is
// not a multiple of 64: B2 = 300 leaves 44 columns = 3 groups there, and the dead group's 1/4 of the tile's arithmetic
// -- 5 % of the whole launch -- is skipped instead of computed and dropped).
template <bool FAST, int NW>
__device__ __forceinline__ void pairwise_dist_body(const float *__restrict__ src, const float *__restrict__ tgt, int64_t B1,
                                                   int64_t B2, int C, int dist_type, float *__restrict__ out,
                                                   ColStat *__restrict__ ws, float stat_scale, f32x4 (&As)[kPK / 4][kPT],
                                                   f32x4 (&Bs)[kPK / 4][kPT])
{
    const int64_t i0 = (int64_t)blockIdx.y * kPT, j0 = (int64_t)blockIdx.x * kPT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const bool vec_ok = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(tgt)) % 16 == 0);
    f32x2 acc[4][NW];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int w = 0; w < NW; ++w) acc[q][w] = (f32x2)0.0f;

    // software pipeline: the next stage's global loads are issued (into registers) before the current stage
    // is consumed; lane -> (row, k/4) pairs e = tid, tid + 256 of the 64 x 8 float4 stage
    f32x4 pa[2], pb[2];
    uint32_t aoff[2], boff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = threadIdx.x + h * kBlock;
        const int r = e % kPT, k4 = e / kPT;
        aoff[h] = (uint32_t)(min(i0 + r, B1 - 1) * C + k4 * 4);
        boff[h] = (uint32_t)(min(j0 + r, B2 - 1) * C + k4 * 4);
    }
    auto fetch = [&](int k0) {
        if (FAST) {
            const float *sa = src + k0, *sb = tgt + k0;          // uniform bases
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pa[h] = *reinterpret_cast<const f32x4 *>(sa + aoff[h]);
                pb[h] = *reinterpret_cast<const f32x4 *>(sb + boff[h]);
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = threadIdx.x + h * kBlock;
            const int r = e % kPT, k4 = e / kPT;
            const int kc = k0 + k4 * 4;
            f32x4 va = (f32x4)0.0f, vb = (f32x4)0.0f;
            if (vec_ok && kc + 3 < C) {
                if (i0 + r < B1) va = *reinterpret_cast<const f32x4 *>(src + (i0 + r) * C + kc);
                if (j0 + r < B2) vb = *reinterpret_cast<const f32x4 *>(tgt + (j0 + r) * C + kc);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (kc + t < C && i0 + r < B1) va[t] = src[(i0 + r) * C + kc + t];
                    if (kc + t < C && j0 + r < B2) vb[t] = tgt[(j0 + r) * C + kc + t];
                }
            }
            pa[h] = va;
            pb[h] = vb;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < C; k0 += kPK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = threadIdx.x + h * kBlock;
            As[e / kPT][e % kPT] = pa[h];
            Bs[e / kPT][e % kPT] = -pb[h];
        }
        __syncthreads();
        if (k0 + kPK < C) fetch(k0 + kPK);
#pragma unroll
        for (int k4 = 0; k4 < kPK / 4; ++k4) {
            f32x4 a[4], b[NW];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = As[k4][ty + 16 * q];
#pragma unroll
            for (int w = 0; w < NW; ++w) b[w] = Bs[k4][tx + 16 * w];
            // per row q: the four differences first, then the even-pair FMAs, then the odd-pair FMAs, so the two
            // dependent v_pk_fma_f32 of one accumulator are four issue slots apart (back to back they cost an s_nop)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 d[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) d[w] = a[q] + b[w];              // b holds -tgt
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const f32x2 lo = __builtin_shufflevector(d[w], d[w], 0, 1);
                    acc[q][w] = __builtin_elementwise_fma(lo, lo, acc[q][w]);
                }
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const f32x2 hi = __builtin_shufflevector(d[w], d[w], 2, 3);
                    acc[q][w] = __builtin_elementwise_fma(hi, hi, acc[q][w]);
                }
            }
        }
        __syncthreads();
    }
    // Epilogue, twice: FULL (every row and column of the tile's NW groups is inside the matrix -- all but the rim tiles;
    // workgroup-uniform) stores and reduces without predicates.  The fast kernel addresses `out` with one 32-bit element
    // offset per lane (B1*B2 < 2^31, host-checked) and the 16 outputs at uniform strides from it.
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int rows_left = (int)min((int64_t)kPT, B1 - i0) - ty, cols_left = (int)min((int64_t)kPT, B2 - j0) - tx;
        float dv[4][NW];
        const uint32_t o00 = FAST ? (uint32_t)(i0 + ty) * (uint32_t)B2 + (uint32_t)(j0 + tx) : 0u, rs = 16u * (uint32_t)B2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float ssum = acc[q][w].x + acc[q][w].y;
                dv[q][w] = dist_type == D3F_DIST_L2 ? sqrtf(ssum) : ssum;
                if (FULL || (16 * q < rows_left && 16 * w < cols_left)) {
                    if (FAST) out[o00 + q * rs + 16 * w] = dv[q][w];
                    else out[(i0 + ty + 16 * q) * B2 + (j0 + tx + 16 * w)] = dv[q][w];
                }
            }
        if (!ws) return;                               // uniform
        // Fused column statistics of softmax(-d*stat_scale, dim=0) over this tile's 64 rows (what softmax_stats_kernel
        // would compute in a second pass over `out`): per lane its 4 rows, then the 4 lane-rows of the wave by shuffles,
        // then the 4 waves through LDS; ws[row tile][column].  The winning row travels as its index inside the tile.
        ColStat *red = reinterpret_cast<ColStat *>(&As[0][0]);      // [4 waves][64 columns] = 4 KiB, stage buffer is free
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            // maximum over the wave's 16 rows of this column first (cheap fmax shuffles), then ONE expf per element
            float v[4], m = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = (FULL || 16 * q < rows_left) ? -dv[q][w] * stat_scale : -INFINITY;
                m = fmaxf(m, v[q]);                    // NaN rows are caught by the sum below
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.0f;
            int arg = 0x7fffffff;
#pragma unroll
            for (int q = 3; q >= 0; --q)
                if (FULL || 16 * q < rows_left) {
                    sum += expf(v[q] - m);             // exp(-inf - -inf) cannot occur: a live row makes m finite or NaN
                    if (v[q] == m) arg = ty + 16 * q;  // descending q: the smallest row index wins
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            arg = min(arg, __shfl_xor(arg, 16, 64));
            arg = min(arg, __shfl_xor(arg, 32, 64));
            if ((threadIdx.x & 63) < 16) {
                ColStat o;
                o.m = m; o.s = sum; o.arg = arg == 0x7fffffff ? 0x7fffffffffffffffLL : i0 + arg;
                red[(threadIdx.x >> 6) * kPT + tx + 16 * w] = o;
            }
        }
        __syncthreads();
        if (threadIdx.x < kPT && j0 + threadIdx.x < B2) {
            ColStat t = red[threadIdx.x];
#pragma unroll
            for (int wv = 1; wv < kBlock / 64; ++wv) {
                const ColStat u = red[wv * kPT + threadIdx.x];
                merge_stat(t.m, t.s, t.arg, u.m, u.s, u.arg);
            }
            ws[(int64_t)blockIdx.y * B2 + j0 + threadIdx.x] = t;
        }
    };
    if (FAST && B1 - i0 >= kPT && B2 - j0 >= 16 * NW) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// NWT: live 16-column groups of the LAST column tile (the other column tiles are full)
template <bool FAST, int NWT>
__global__ __launch_bounds__(kBlock) void pairwise_dist_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ tgt, int64_t B1, int64_t B2,
                                                              int C, int dist_type, float *__restrict__ out,
                                                              ColStat *__restrict__ ws, float stat_scale)
{
    __shared__ f32x4 As[kPK / 4][kPT];
    __shared__ f32x4 Bs[kPK / 4][kPT];
    if (NWT < 4 && blockIdx.x == gridDim.x - 1)          // uniform per workgroup
        pairwise_dist_body<FAST, NWT>(src, tgt, B1, B2, C, dist_type, out, ws, stat_scale, As, Bs);
    else
        pairwise_dist_body<FAST, 4>(src, tgt, B1, B2, C, dist_type, out, ws, stat_scale, As, Bs);
}

// ---- the same distances through the contraction |a|^2 + |b|^2 - 2 a.b on the fp32 matrix cores, GUARDED (round 6) -----------------
// The direct form above costs two packed VALU instructions per two (pair, channel) terms -- a difference and a square-accumulate --
// and four 16-byte LDS operand reads per sixteen of them.  The contraction needs ONE multiply-accumulate per term, and
// v_mfma_f32_16x16x4_f32 does 1024 of them per instruction from two operand registers per lane: the same 32 MAC / cycle / SIMD as
// v_pk_fma_f32 (MI355X_MICROARCH.md: exact fp32, bit for bit an fmaf chain), but half the issue slots of the direct form, a tenth
// of its LDS operand traffic, and on a pipe of its own.  What the contraction loses is accuracy where it cancels: its error is
// ~2e-7 * (|a|^2 + |b|^2) ABSOLUTE on d^2, harmless for far pairs and fatal for near matches -- the pairs a correspondence lookup
// is for.  Hence the guard: a pair whose contraction result is not at least a QUARTER of |a|^2 + |b|^2 (or is not finite) is
// recomputed in the direct form -- by the wave, cooperatively (64 lanes x one coalesced row pair), when a tile has few such pairs;
// by pairwise_dist_body over the whole tile when it has many.  Beyond the guard d is good to 4e-7 relative, which moves a
// softmax(-scale * d) output by at most ~2e-7 * scale * d_tie (two rows tied at distance d_tie and nothing nearer).
// Tile and stage layout are those of pairwise_dist_body (64 x 64 outputs per workgroup, 32 channels per LDS stage, float4 columns);
// wave w owns rows 16 w .. 16 w + 15 of the tile and all 64 columns: four accumulator tiles of v_mfma_f32_16x16x4_f32.
// A lane reads ONE float4 per operand and k-half of a stage and feeds its four components to four successive MFMAs: MFMA c of
// half r contracts the channels {4 (4 r + s) + c : s = 0..3} -- every channel of the stage exactly once, in a fixed order.
constexpr int kGuardDenseFlags = 96;          // flagged pairs per 64 x 64 tile beyond which the whole tile takes the direct form
constexpr int64_t kMfmaMinRows = 512;         // fewer source rows: the direct kernel (nothing to win, and small softmaxes have O(1) entries)

template <int NW>
__device__ __forceinline__ void pairwise_mfma_body(const float *__restrict__ src, const float *__restrict__ tgt, int64_t B1,
                                                   int64_t B2, int C, int dist_type, float *__restrict__ out,
                                                   ColStat *__restrict__ ws, float stat_scale, f32x4 (&As2)[2][kPK / 4][kPT],
                                                   f32x4 (&Bs2)[2][kPK / 4][kPT])
{
    f32x4 (&As)[kPK / 4][kPT] = As2[0];          // (the epilogue and the direct fallback reuse the first stage buffer)
    f32x4 (&Bs)[kPK / 4][kPT] = Bs2[0];
    const int64_t i0 = (int64_t)blockIdx.y * kPT, j0 = (int64_t)blockIdx.x * kPT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lc = lane & 15, ls = lane >> 4;                    // column inside a 16-wide block / k slot (operands), row quad (results)
    __shared__ float na_s[kBlock / 64][kPT], nb_s[kBlock / 64][kPT];
    __shared__ int dense_s;
    f32x4 acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = (f32x4)0.0f;
    // stage fetch: as pairwise_dist_body<FAST> (uniform base + loop-invariant 32-bit offsets, rows clamped into the matrix)
    f32x4 pa[2], pb[2];
    uint32_t aoff[2], boff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = threadIdx.x + h * kBlock;
        const int r = e % kPT, k4 = e / kPT;
        aoff[h] = (uint32_t)(min(i0 + r, B1 - 1) * C + k4 * 4);
        boff[h] = (uint32_t)(min(j0 + r, B2 - 1) * C + k4 * 4);
    }
    auto fetch = [&](int k0) {
        const float *sa = src + k0, *sb = tgt + k0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pa[h] = *reinterpret_cast<const f32x4 *>(sa + aoff[h]);
            pb[h] = *reinterpret_cast<const f32x4 *>(sb + boff[h]);
        }
    };
    if (threadIdx.x == 0) dense_s = 0;
    // |a|^2, |b|^2: every lane squares what it stages (row threadIdx % 64, k quads threadIdx / 64 and + 4 of every stage), four
    // lanes -- one per wave -- share a row; sixteen short partial sums per row instead of one chain of C terms
    f32x4 na4 = (f32x4)0.0f, nb4 = (f32x4)0.0f;
    // two stage buffers, ONE barrier per stage: while the matrix cores contract stage k out of buffer k & 1, the registers fetched a
    // stage ahead are stored into the other buffer (whose readers all passed the previous barrier)
    auto stash = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = threadIdx.x + h * kBlock;
            As2[buf][e / kPT][e % kPT] = pa[h];
            Bs2[buf][e / kPT][e % kPT] = pb[h];
            na4 = __builtin_elementwise_fma(pa[h], pa[h], na4);
            nb4 = __builtin_elementwise_fma(pb[h], pb[h], nb4);
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < C; k0 += kPK, cur ^= 1) {
        const bool more = k0 + kPK < C;
        if (more) fetch(k0 + kPK);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const f32x4 a = As2[cur][4 * r + ls][16 * wv + lc];
            f32x4 b[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) b[w] = Bs2[cur][4 * r + ls][16 * w + lc];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int w = 0; w < NW; ++w) acc[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[w][c], acc[w], 0, 0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
    }
    na_s[wv][lane] = (na4.x + na4.y) + (na4.z + na4.w);
    nb_s[wv][lane] = (nb4.x + nb4.y) + (nb4.z + nb4.w);
    __syncthreads();
    // this lane's results: rows 16 wv + 4 ls + i (i < 4), columns 16 w + lc
    float na[4], nb[NW];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 16 * wv + 4 * ls + i;
        na[i] = (na_s[0][r] + na_s[1][r]) + (na_s[2][r] + na_s[3][r]);
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int cidx = 16 * w + lc;
        nb[w] = (nb_s[0][cidx] + nb_s[1][cidx]) + (nb_s[2][cidx] + nb_s[3][cidx]);
    }
    const int rows_left = (int)min((int64_t)kPT, B1 - i0) - (16 * wv + 4 * ls), cols_left = (int)min((int64_t)kPT, B2 - j0) - lc;
    float d2[4][NW];
    uint32_t flagged = 0u;                       // bit 4 w + i: this lane's pair (i, w) is inside the matrix and fails the guard
    int nflag = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float nsum = na[i] + nb[w];
            d2[i][w] = fmaf(-2.0f, acc[w][i], nsum);
            const bool live = i < rows_left && 16 * w < cols_left;
            if (live && !(d2[i][w] >= 0.25f * nsum && nsum < INFINITY)) { flagged |= 1u << (4 * w + i); ++nflag; }     // NaN / Inf: flagged
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nflag += __shfl_xor(nflag, off, 64);
    if (lane == 0 && nflag > 0) atomicAdd(&dense_s, nflag);
    __syncthreads();
    if (dense_s > kGuardDenseFlags) {            // (workgroup-uniform) many near pairs: the whole tile in the direct form
        __syncthreads();                         // everyone has read dense_s and the norms before the stage buffers are reused
        pairwise_dist_body<true, NW>(src, tgt, B1, B2, C, dist_type, out, ws, stat_scale, As, Bs);
        return;
    }
    // few near pairs: each is recomputed by its wave, all 64 lanes on one (row, column) -- coalesced float4 reads of both rows
    if (dense_s > 0) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned long long m = __ballot((flagged >> (4 * w + i)) & 1u);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int64_t row = i0 + 16 * wv + 4 * (l >> 4) + i, col = j0 + 16 * w + (l & 15);
                    const f32x4 *pr = reinterpret_cast<const f32x4 *>(src + row * C), *pc = reinterpret_cast<const f32x4 *>(tgt + col * C);
                    f32x2 s2 = (f32x2)0.0f;
                    for (int k4 = lane; k4 < C / 4; k4 += 64) {
                        const f32x4 d = pr[k4] - pc[k4];
                        const f32x2 lo = __builtin_shufflevector(d, d, 0, 1), hi = __builtin_shufflevector(d, d, 2, 3);
                        s2 = __builtin_elementwise_fma(lo, lo, s2);
                        s2 = __builtin_elementwise_fma(hi, hi, s2);
                    }
                    float ssum = s2.x + s2.y;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) ssum += __shfl_xor(ssum, off, 64);
                    if (lane == l) d2[i][w] = ssum;
                }
            }
    }
    // ---- epilogue: distances out, column statistics of the softmax (as pairwise_dist_body's; this lane's rows are 4 ls + i) ----
    float dv[4][NW];
    const uint32_t o00 = (uint32_t)(i0 + 16 * wv + 4 * ls) * (uint32_t)B2 + (uint32_t)(j0 + lc);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            // (an unflagged d2 is >= 0: the guard; a recomputed one is a sum of squares)
            dv[i][w] = dist_type == D3F_DIST_L2 ? sqrtf(d2[i][w]) : d2[i][w];
            if (i < rows_left && 16 * w < cols_left) out[o00 + (uint32_t)i * (uint32_t)B2 + 16 * w] = dv[i][w];
        }
    if (!ws) return;                                   // uniform
    ColStat *red = reinterpret_cast<ColStat *>(&As[0][0]);      // [4 waves][64 columns] = 4 KiB, the stage buffer is free
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        float v[4], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = i < rows_left ? -dv[i][w] * stat_scale : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.0f;
        int arg = 0x7fffffff;
#pragma unroll
        for (int i = 3; i >= 0; --i)
            if (i < rows_left) {
                sum += expf(v[i] - m);
                if (v[i] == m) arg = 16 * wv + 4 * ls + i;          // descending i: the smallest row index of this lane wins
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        arg = min(arg, __shfl_xor(arg, 16, 64));
        arg = min(arg, __shfl_xor(arg, 32, 64));
        if (lane < 16) {
            ColStat o;
            o.m = m; o.s = sum; o.arg = arg == 0x7fffffff ? 0x7fffffffffffffffLL : i0 + arg;
            red[wv * kPT + lc + 16 * w] = o;
        }
    }
    __syncthreads();
    if (threadIdx.x < kPT && threadIdx.x < 16 * NW && j0 + threadIdx.x < B2) {
        ColStat t = red[threadIdx.x];
#pragma unroll
        for (int q = 1; q < kBlock / 64; ++q) {
            const ColStat u = red[q * kPT + threadIdx.x];
            merge_stat(t.m, t.s, t.arg, u.m, u.s, u.arg);
        }
        ws[(int64_t)blockIdx.y * B2 + j0 + threadIdx.x] = t;
    }
}

template <int NWT>
__global__ __launch_bounds__(kBlock, 4) void pairwise_mfma_kernel(const float *__restrict__ src, const float *__restrict__ tgt, int64_t B1,
                                                              int64_t B2, int C, int dist_type, float *__restrict__ out,
                                                              ColStat *__restrict__ ws, float stat_scale)
{
    __shared__ f32x4 As[2][kPK / 4][kPT];
    __shared__ f32x4 Bs[2][kPK / 4][kPT];
    if (NWT < 4 && blockIdx.x == gridDim.x - 1)          // uniform per workgroup
        pairwise_mfma_body<NWT>(src, tgt, B1, B2, C, dist_type, out, ws, stat_scale, As, Bs);
    else
        pairwise_mfma_body<4>(src, tgt, B1, B2, C, dist_type, out, ws, stat_scale, As, Bs);
}

// ws != nullptr: also writes the column statistics of softmax(-d*stat_scale, dim=0) per 64-row tile into
// ws[tile][column] (the layout softmax_merge_kernel reads), saving the separate pass over `out`
hipError_t launch_pairwise_dist(const float *src, const float *tgt, int64_t B1, int64_t B2, int C, int dist_type,
                                float *out, hipStream_t s, ColStat *ws, float stat_scale, bool direct_only)
{
    if (B1 == 0 || B2 == 0) return hipSuccess;
    dim3 grid((unsigned)((B2 + kPT - 1) / kPT), (unsigned)((B1 + kPT - 1) / kPT));
    const bool fast = (C % kPK == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(tgt)) % 16 == 0) &&
                      B1 * (int64_t)C < (1LL << 30) && B2 * (int64_t)C < (1LL << 30) && B1 * B2 < (1LL << 31);
    const int tail_groups = (int)((B2 - (int64_t)(grid.x - 1) * kPT + 15) / 16);         // 1..4
    // the guarded contraction on the matrix cores (above) for the shapes of a correspondence lookup; `direct_only`: the direct form
    // everywhere (the reference arithmetic of the tests)
    const bool mfma = fast && !direct_only && B1 >= kMfmaMinRows;
#define D3F_PAIRWISE(NWT_)                                                                                                      \
    do {                                                                                                                        \
        if (mfma) hipLaunchKernelGGL((pairwise_mfma_kernel<NWT_>), grid, dim3(kBlock), 0, s, src, tgt, B1, B2, C, dist_type, out, ws, stat_scale); \
        else hipLaunchKernelGGL((pairwise_dist_kernel<true, NWT_>), grid, dim3(kBlock), 0, s, src, tgt, B1, B2, C, dist_type, out, ws, stat_scale); \
    } while (0)
    if (!fast)
        hipLaunchKernelGGL((pairwise_dist_kernel<false, 4>), grid, dim3(kBlock), 0, s, src, tgt, B1, B2, C, dist_type, out, ws, stat_scale);
    else if (tail_groups == 1) D3F_PAIRWISE(1);
    else if (tail_groups == 2) D3F_PAIRWISE(2);
    else if (tail_groups == 3) D3F_PAIRWISE(3);
    else D3F_PAIRWISE(4);
#undef D3F_PAIRWISE
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
    const int64_t j = (int64_t)blockIdx.y * kBlock + threadIdx.x;
    if (j >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * kSoftmaxRowsPerBlock;
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
    ws[(int64_t)blockIdx.x * cols + j] = o;
}

// Eight columns per 1024-lane workgroup: thread t owns column (t & 7) and every 128th row chunk from (t >> 3), so that
// one wave load covers eight chunks x one 128-byte line (8 columns x 16 B); four chunks' loads are in flight per lane.
// The 128 partial triples of a column are merged by shuffles (lanes 8, 16, 32 apart) and across the sixteen waves through
// LDS, in a fixed order.  The kernel is a chain of memory round trips, so its time is the number of rounds: one wave per
// column with the lanes striding over the chunks (64 different lines per wave load, no loads in flight together) took
// 18.6 us for 1563 chunks x 300 columns; 256 lanes in this layout 16.3 us (13 rounds of four loads); 1024 lanes 4 rounds.

// `in` [nchunks, cols] -> `out` [cols]; arg_offset shifts the winning row index (a rank's first global row when
// the rows are sharded over GPUs; 0 otherwise).  nchunks == 0 writes the identity (-inf, 0, INT64_MAX).
constexpr int kMergeCols = 8, kMergeBlock = 1024;
__global__ __launch_bounds__(kMergeBlock) void softmax_merge_kernel(const ColStat *__restrict__ in, int64_t nchunks,
                                                              int64_t cols, ColStat *__restrict__ out,
                                                              int64_t *__restrict__ argmax_out, int64_t arg_offset)
{
    __shared__ ColStat red[kMergeBlock / 64][kMergeCols];
    constexpr int kLanes = kMergeBlock / kMergeCols;             // chunk lanes per column: 128
    const int cj = threadIdx.x & (kMergeCols - 1), cl = threadIdx.x >> 3;
    const int64_t j = min((int64_t)blockIdx.x * kMergeCols + cj, cols - 1);      // a short last group repeats its last column
    float m = -INFINITY, s = 0.0f;
    int64_t arg = 0x7fffffffffffffffLL;
    int64_t c = cl;
    for (; c + 3 * kLanes < nchunks; c += 4 * kLanes) {
        ColStat t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = in[(c + u * kLanes) * cols + j];
#pragma unroll
        for (int u = 0; u < 4; ++u) merge_stat(m, s, arg, t[u].m, t[u].s, t[u].arg);
    }
    for (; c < nchunks; c += kLanes) {
        const ColStat t = in[c * cols + j];
        merge_stat(m, s, arg, t.m, t.s, t.arg);
    }
#pragma unroll
    for (int off = kMergeCols; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        const int64_t a2 = __shfl_xor(arg, off, 64);
        merge_stat(m, s, arg, m2, s2, a2);
    }
    if ((threadIdx.x & 63) < kMergeCols) {
        ColStat o;
        o.m = m; o.s = s; o.arg = arg;
        red[threadIdx.x >> 6][cj] = o;
    }
    __syncthreads();
    if (threadIdx.x < kMergeCols && (int64_t)blockIdx.x * kMergeCols + threadIdx.x < cols) {
        ColStat t = red[0][threadIdx.x];
#pragma unroll
        for (int wv = 1; wv < kMergeBlock / 64; ++wv) {
            const ColStat u = red[wv][threadIdx.x];
            merge_stat(t.m, t.s, t.arg, u.m, u.s, u.arg);
        }
        if (t.arg != 0x7fffffffffffffffLL) t.arg += arg_offset;
        out[j] = t;
        if (argmax_out) argmax_out[j] = t.arg;
    }
}

__global__ __launch_bounds__(kBlock) void softmax_apply_kernel(float *__restrict__ x, int64_t total, int64_t cols,
                                                              float scale, const ColStat *__restrict__ fin)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= total) return;
    const ColStat t = fin[k % cols];
    x[k] = expf(-x[k] * scale - t.m) / t.s;
}

// Four columns per lane (cols % 4 == 0, 16-byte aligned matrix): the launch stride is a multiple of cols / 4, so a lane
// stays on its column quad -- its statistics are loaded once, the column index costs one 32-bit remainder per lane instead
// of a 64-bit one per element -- and walks down the rows with four 16-byte loads in flight.  Same expression per element
// as the scalar form (bit-identical).
__global__ __launch_bounds__(kBlock) void softmax_apply_vec_kernel(f32x4 *__restrict__ x, int64_t quads, uint32_t cq,
                                                                  float scale, const ColStat *__restrict__ fin)
{
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x, stride = gridDim.x * kBlock;
    const uint32_t c4 = (g % cq) * 4u;
    float m[4], sum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const ColStat st = fin[c4 + t];
        m[t] = st.m;
        sum[t] = st.s;
    }
    auto norm = [&](f32x4 v) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = expf(-v[t] * scale - m[t]) / sum[t];
        return v;
    };
    int64_t q = g;
    for (; q + 3 * (int64_t)stride < quads; q += 4 * (int64_t)stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = x[q + u * (int64_t)stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[q + u * (int64_t)stride] = norm(v[u]);
    }
    for (; q < quads; q += stride) x[q] = norm(x[q]);
}

static void apply_launch(float *x, int64_t total, int64_t cols, float scale, const ColStat *fin, hipStream_t s)
{
    const int64_t cq = cols / 4;
    if (cols % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && cq <= 4096) {
        // workgroups: a multiple of cq (=> stride % cq == 0) near 2048 (8 per CU), never more than the quads need
        const int64_t quads = total / 4;
        int64_t groups = cq * std::max<int64_t>(1, 2048 / cq);
        while (groups > cq && (groups - cq) * kBlock >= quads) groups -= cq;
        hipLaunchKernelGGL(softmax_apply_vec_kernel, dim3((unsigned)groups), dim3(kBlock), 0, s, reinterpret_cast<f32x4 *>(x), quads,
                           (uint32_t)cq, scale, fin);
        return;
    }
    hipLaunchKernelGGL(softmax_apply_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, x, total, cols, scale,
                       fin);
}

static void merge_launch(const ColStat *in, int64_t nchunks, int64_t cols, ColStat *out, int64_t *argmax_out,
                         int64_t arg_offset, hipStream_t s)
{
    hipLaunchKernelGGL(softmax_merge_kernel, dim3((unsigned)((cols + kMergeCols - 1) / kMergeCols)), dim3(kMergeBlock), 0, s, in,
                       nchunks, cols, out, argmax_out, arg_offset);
}

// have_stats: ws[chunk][column] was already filled by the producer of x (pairwise_dist_kernel's epilogue)
static hipError_t softmax_impl(float *x, int64_t rows, int64_t cols, float scale, int64_t *argmax_out, ColStat *ws,
                               bool normalise, bool have_stats, hipStream_t s)
{
    if (rows == 0 || cols == 0) return hipSuccess;
    const int64_t nchunks = (rows + kSoftmaxRowsPerBlock - 1) / kSoftmaxRowsPerBlock;
    const unsigned gx = (unsigned)((cols + kBlock - 1) / kBlock);
    if (!have_stats)
        hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)nchunks, gx), dim3(kBlock), 0, s, x, rows, cols, scale, ws);
    merge_launch(ws, nchunks, cols, ws + nchunks * cols, argmax_out, 0, s);
    if (normalise) {
        apply_launch(x, rows * cols, cols, scale, ws + nchunks * cols, s);
    }
    return hipGetLastError();
}

// ---- row-sharded softmax (shard B1 over GPUs, exchange 16 B per column) --------------
// local statistics of this rank's rows (global row index = row_offset + local row); rows == 0 gives the identity
hipError_t launch_softmax_local_stats(const float *x, int64_t rows, int64_t cols, float scale, int64_t row_offset,
                                      ColStat *ws, ColStat *stats_out, bool have_stats, hipStream_t s)
{
    if (cols == 0) return hipSuccess;
    const int64_t nchunks = (rows + kSoftmaxRowsPerBlock - 1) / kSoftmaxRowsPerBlock;
    if (nchunks > 0 && !have_stats) {
        const unsigned gx = (unsigned)((cols + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)nchunks, gx), dim3(kBlock), 0, s, x, rows, cols, scale, ws);
    }
    merge_launch(ws, nchunks, cols, stats_out, nullptr, row_offset, s);
    return hipGetLastError();
}

// merges the per-rank statistics [parts, cols] (rank order) into [cols] (+ global argmax)
hipError_t launch_softmax_merge(const ColStat *parts, int64_t nparts, int64_t cols, ColStat *merged, int64_t *argmax_out,
                                hipStream_t s)
{
    if (cols == 0) return hipSuccess;
    merge_launch(parts, nparts, cols, merged, argmax_out, 0, s);
    return hipGetLastError();
}

hipError_t launch_softmax_apply(float *x, int64_t rows, int64_t cols, float scale, const ColStat *merged, hipStream_t s)
{
    const int64_t total = rows * cols;
    if (total == 0) return hipSuccess;
    apply_launch(x, total, cols, scale, merged, s);
    return hipGetLastError();
}

hipError_t launch_softmax_dim0(float *x, int64_t rows, int64_t cols, float scale, int64_t *argmax_out, ColStat *ws,
                               bool have_stats, hipStream_t s)
{
    return softmax_impl(x, rows, cols, scale, argmax_out, ws, true, have_stats, s);
}

hipError_t launch_argmin_dim0(const float *x, int64_t rows, int64_t cols, int64_t *arg_out, ColStat *ws,
                              bool have_stats, hipStream_t s)
{
    return softmax_impl(const_cast<float *>(x), rows, cols, 1.0f, arg_out, ws, false, have_stats, s);
}

// ---- k nearest descriptors per target column (k <= 8) ------------------------------------------------------------------
// The north star names a "KNN correspondence lookup"; the reference has none: its best match is
// compute_similarity_tensor_multi(...).argmax(0), i.e. the smallest distance of a column.  The k-NN extension keeps that
// definition: per column of the [rows, cols] DISTANCE matrix the k smallest entries, ties -> lower row index, NaN last;
// selection happens on the distances, before exp()/softmax can round neighbours into ties.
// Two steps, both reading the matrix row-wise (lanes = consecutive columns, coalesced):
//   topk_chunk_kernel  64 columns x 256 rows per workgroup; thread (c, r) scans rows r, r+4, ... with an 8-entry sorted
//                      list in registers, the four row lanes of a column merge through LDS -> 8 candidates per chunk;
//   topk_merge_kernel  the same scan over the candidates of all chunks of a column.
constexpr int kTopK = 8;
constexpr int kTopkRows = 256;

struct Cand {
    float v;
    int32_t i;
};

__device__ __forceinline__ bool cand_before(float v, int32_t i, const Cand &o) { return v < o.v || (v == o.v && i < o.i); }

__device__ __forceinline__ void cand_insert(Cand (&best)[kTopK], float v, int32_t i)
{
    if (!cand_before(v, i, best[kTopK - 1])) return;
    best[kTopK - 1].v = v; best[kTopK - 1].i = i;
#pragma unroll
    for (int j = kTopK - 1; j > 0; --j) {
        if (cand_before(best[j].v, best[j].i, best[j - 1])) {
            const Cand t = best[j - 1]; best[j - 1] = best[j]; best[j] = t;
        }
    }
}

// in: dist != nullptr -> rows of the matrix (row index = global row); else cand_in [n_in, cols] candidate lists
__global__ __launch_bounds__(kBlock) void topk_chunk_kernel(const float *__restrict__ dist, const Cand *__restrict__ cand_in,
                                                           int64_t rows, int64_t cols, Cand *__restrict__ out)
{
    __shared__ Cand lds[4][64][kTopK + 1];          // +1: 9 x 8 B rows spread the banks
    const int c_local = threadIdx.x & 63, r_lane = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c_local;
    const int64_t row0 = (int64_t)blockIdx.y * kTopkRows, row1 = min(rows, row0 + kTopkRows);
    Cand best[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { best[j].v = INFINITY; best[j].i = 0x7fffffff; }
    if (col < cols) {
        for (int64_t r = row0 + r_lane; r < row1; r += 4) {
            float v;
            int32_t i;
            if (dist) { v = dist[r * cols + col]; i = (int32_t)r; }
            else { const Cand cnd = cand_in[r * cols + col]; v = cnd.v; i = cnd.i; }
            if (v != v) v = INFINITY;               // NaN sorts last (by row index among the infinities)
            cand_insert(best, v, i);
        }
    }
#pragma unroll
    for (int j = 0; j < kTopK; ++j) lds[r_lane][c_local][j] = best[j];
    __syncthreads();
    if (r_lane == 0 && col < cols) {
        for (int q = 1; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < kTopK; ++j) {
                const Cand cnd = lds[q][c_local][j];
                cand_insert(best, cnd.v, cnd.i);
            }
#pragma unroll
        for (int j = 0; j < kTopK; ++j) out[((int64_t)blockIdx.y * kTopK + j) * cols + col] = best[j];
    }
}

__global__ __launch_bounds__(kBlock) void topk_write_kernel(const Cand *__restrict__ best, const float *__restrict__ x, int64_t rows,
                                                           int64_t cols, int k, int64_t *__restrict__ idx_out,
                                                           float *__restrict__ val_out)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (int64_t)k * cols) return;
    const int64_t j = t / cols, col = t - j * cols;
    const bool real = j < rows;                    // fewer rows than k: the tail is (-1, NaN)
    Cand cnd = {INFINITY, 0};
    if (real) cnd = best[j * cols + col];
    idx_out[t] = real ? (int64_t)cnd.i : -1;
    if (val_out) val_out[t] = real ? x[(int64_t)cnd.i * cols + col] : NAN;     // the matrix AFTER exp / softmax was applied
}

int64_t topk_workspace_bytes(int64_t rows, int64_t cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    int64_t total = 0, n = rows;
    while (true) {                                  // levels of 256-fold reduction until one candidate list per column
        const int64_t chunks = (n + kTopkRows - 1) / kTopkRows;
        total += chunks * kTopK * cols * (int64_t)sizeof(Cand);
        if (chunks == 1) break;
        n = chunks * kTopK;
    }
    return total;
}

// dist [rows, cols] -> workspace ends with the final [kTopK, cols] list; returns a pointer to it
hipError_t launch_topk_select(const float *dist, int64_t rows, int64_t cols, void *workspace, const void **final_list, hipStream_t s)
{
    Cand *level = static_cast<Cand *>(workspace);
    const Cand *in = nullptr;
    int64_t n = rows;
    while (true) {
        const int64_t chunks = (n + kTopkRows - 1) / kTopkRows;
        hipLaunchKernelGGL(topk_chunk_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)chunks), dim3(kBlock), 0, s,
                           in ? nullptr : dist, in, n, cols, level);
        if (chunks == 1) break;
        in = level;
        n = chunks * kTopK;
        level += chunks * kTopK * cols;
    }
    *final_list = level;
    return hipGetLastError();
}

hipError_t launch_topk_write(const void *final_list, const float *x, int64_t rows, int64_t cols, int k, int64_t *idx_out,
                             float *val_out, hipStream_t s)
{
    const int64_t total = (int64_t)k * cols;
    hipLaunchKernelGGL(topk_write_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       static_cast<const Cand *>(final_list), x, rows, cols, k, idx_out, val_out);
    return hipGetLastError();
}

// merge of per-rank k-nearest lists (rows sharded over ranks: sharding.sharded_knn_descriptors): per column the k best of
// the n_parts * k candidates by (value ascending, GLOBAL row index ascending); entries with index < 0 are padding
__global__ __launch_bounds__(kBlock) void topk_merge_parts_kernel(const int64_t *__restrict__ pidx, const float *__restrict__ pval, int64_t n_parts,
                                                                 int k, int64_t cols, int64_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int64_t col = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (col >= cols) return;
    float bv[kTopK];
    int64_t bi[kTopK];
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { bv[j] = INFINITY; bi[j] = -1; }
    auto before = [](float v, int64_t i, float ov, int64_t oi) { return oi < 0 || v < ov || (v == ov && i < oi); };
    for (int64_t t = 0; t < n_parts * k; ++t) {
        const int64_t i = pidx[t * cols + col];
        if (i < 0) continue;
        float v = pval[t * cols + col];
        if (v != v) v = INFINITY;
        if (!before(v, i, bv[kTopK - 1], bi[kTopK - 1])) continue;
        bv[kTopK - 1] = v; bi[kTopK - 1] = i;
#pragma unroll
        for (int j = kTopK - 1; j > 0; --j)
            if (before(bv[j], bi[j], bv[j - 1], bi[j - 1])) {
                const float tv = bv[j - 1]; const int64_t ti = bi[j - 1];
                bv[j - 1] = bv[j]; bi[j - 1] = bi[j]; bv[j] = tv; bi[j] = ti;
            }
    }
    for (int j = 0; j < k; ++j) {
        out_idx[(int64_t)j * cols + col] = bi[j];
        if (out_val) out_val[(int64_t)j * cols + col] = bi[j] < 0 ? NAN : bv[j];
    }
}

hipError_t launch_topk_merge_parts(const int64_t *pidx, const float *pval, int64_t n_parts, int k, int64_t cols, int64_t *out_idx,
                                   float *out_val, hipStream_t s)
{
    hipLaunchKernelGGL(topk_merge_parts_kernel, dim3((unsigned)((cols + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, pidx, pval, n_parts, k, cols,
                       out_idx, out_val);
    return hipGetLastError();
}

}  // namespace d3f
