// grid_kernels.hip -- keypoint selection helpers of the reference, kept on the device.
//
// select_features_rand (fusion.py:1418-1475) evaluates a 1-mm grid (~1e8 points) with ['mask'], keeps
// the points with |dist| < 5 mm that are valid and mostly one instance, copies them to the host and runs a
// numpy farthest-point sampling (utils/my_utils.py:478-497).  Here
//   grid_shell_kernel   generates the grid points from the axis arrays (nothing of size N is read),
//                       evaluates dist/valid with the forward's arithmetic and COMPACTS the indices of
//                       the thin shell |dist| < thr & valid (about 3 % of the grid) IN ASCENDING ORDER (ballot
//                       words + workgroup counts, exclusive scan, ordered write), so that the wide mask query
//                       runs on the survivors only and no [N,NI] tensor is ever written;
//   fps_kernel          farthest point sampling of the survivors, one workgroup, same float32
//                       arithmetic and first-maximum tie rule as numpy (bit-identical selection).
#include <cstring>
#include <string.h>

#include "d3f_internal.h"
#include "d3f_device.h"
#include "dist_views.h"

namespace d3f {

// Pass 1: survivor flag of every grid point as one ballot word per wave + survivor count per workgroup.
__global__ __launch_bounds__(kBlock) void grid_shell_flag_kernel(const float *__restrict__ depth, const float *__restrict__ K,
                                                                const float *__restrict__ pose, int V, int H, int W,
                                                                const float *__restrict__ gx, const float *__restrict__ gy,
                                                                const float *__restrict__ gz, int ny, int nz, int64_t n, float mu,
                                                                float dist_thr, unsigned long long *__restrict__ ballots,
                                                                uint32_t *__restrict__ block_counts)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *krt = reinterpret_cast<float *>(smem);
    __shared__ int wave_cnt[kBlock / 64];
    compute_krt(K, pose, V, krt, kBlock);
    __syncthreads();
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1);
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const int64_t iz = i % nz, ixy = i / nz;
        const float px = gx[ixy / ny], py = gy[ixy % ny], pz = gz[iz];
        float dsum = 0.0f, cnt = 0.0f;
        for (int v = 0; v < V; ++v) {
            float wgt;
            const ViewOut o = eval_view<0>(depth, H, W, krt + v * 12, v, px, py, pz, Wm1, Hm1, mu, wgt);
            dsum = dsum + o.dist * o.valid;
            cnt = cnt + o.valid;
        }
        // fusion.py:1430, 1444: |dist| < dist_threshold and valid_mask (an all-invalid point has dist = 1e3)
        keep = (cnt != 0.0f) && (fabsf(dsum / (cnt + 1e-6f)) < dist_thr);
    }
    const unsigned long long ballot = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        ballots[(int64_t)blockIdx.x * (kBlock / 64) + wave] = ballot;
        wave_cnt[wave] = __popcll(ballot);
    }
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// Pass 1 for up to eight views (round 6, last): the distance of a grid point as the distance-only kernel computes it (dist_views.h:
// KRt in SGPRs, the views stage by stage, short IEEE divisions, depth pixels from the tiled copy when the caller's scratch has room
// for one) -- the same operations on the same operands as the kernel above, so the same survivors.  A workgroup takes kShellBlocks
// consecutive 256-point blocks, one per WAVE (KRt once per wave and four points); the flat index is taken apart in 32-bit arithmetic (n < 2^32).
constexpr int kShellBlocks = 4;      // 256-point blocks per wave on big grids (>= 2^18 blocks); smaller grids: one, so that the chip still gets many rounds of workgroups
template <int NVQ, bool TILED>
__global__ __launch_bounds__(kBlock, (NVQ == 1 || NVQ == 2) ? 8 : 6) void grid_shell_flag_fast_kernel(const EvalParams P, float dist_thr,
                                                                                                unsigned long long *__restrict__ ballots,
                                                                                                uint32_t *__restrict__ block_counts, int blocks_per_wave)
{
    const float mu = P.mu;
    const DivConst Wm1 = div_const((float)(P.W - 1)), Hm1 = div_const((float)(P.H - 1));
    const int V = P.V;
    const float kr0 = dist_krt_lane(P, 0), kr1 = NVQ == 0 ? dist_krt_lane(P, 4) : 0.0f;
    float M[4][12];
    if constexpr (NVQ > 0) dist_krt_uniform(kr0, M);
    const uint32_t n = (uint32_t)P.n, nz = (uint32_t)P.grid_nz, ny = (uint32_t)P.grid_ny;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nb = ((int64_t)P.n + kBlock - 1) / kBlock;
    // a WAVE owns whole 256-point blocks (four ballot words and their count each), kShellBlocks of them one after the other: no LDS, no
    // barrier, KRt once per wave and sixteen points per lane
#pragma unroll 1
    for (int b = 0; b < blocks_per_wave; ++b) {
    const int64_t blk = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * blocks_per_wave + b;
    if (blk >= nb) return;                                          // wave-uniform
    int count = 0;
#pragma unroll 1
    for (int k = 0; k < kBlock / 64; ++k) {
        const uint32_t i = (uint32_t)blk * kBlock + (uint32_t)(k * 64 + lane);
        bool keep = false;
        if (i < n) {
            const uint32_t ixy = i / nz, iz = i - ixy * nz;
            const uint32_t ix = ixy / ny, iy = ixy - ix * ny;
            const float px = P.grid_x[ix], py = P.grid_y[iy], pz = P.grid_z[iz];
            float dsum = 0.0f, cnt = 0.0f;
            if constexpr (NVQ > 0) {
                dist_views<0, NVQ, TILED>(P, M, 0, px, py, pz, Wm1, Hm1, mu, dsum, cnt);
            } else {
                dist_krt_uniform(kr0, M);
                dist_views<0, 4, TILED>(P, M, 0, px, py, pz, Wm1, Hm1, mu, dsum, cnt);
                __builtin_amdgcn_sched_barrier(0);
                dist_krt_uniform(kr1, M);
                if (V == 8) dist_views<0, 4, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, dsum, cnt);
                else if (V == 7) dist_views<0, 3, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, dsum, cnt);
                else if (V == 6) dist_views<0, 2, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, dsum, cnt);
                else dist_views<0, 1, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, dsum, cnt);
            }
            // fusion.py:1430, 1444: |dist| < dist_threshold and valid_mask (an all-invalid point has dist = 1e3)
            keep = (cnt != 0.0f) && (fabsf(dsum / (cnt + 1e-6f)) < dist_thr);
        }
        const unsigned long long ballot = __ballot(keep);
        if (lane == 0) ballots[blk * (kBlock / 64) + k] = ballot;
        count += __popcll(ballot);
    }
    if (lane == 0) block_counts[blk] = (uint32_t)count;
    }
}

template <bool TILED>
static void launch_shell_flag_fast(const EvalParams &P, float dist_thr, unsigned long long *ballots, uint32_t *counts, int64_t nb, hipStream_t s)
{
    const int bpw = nb >= (1LL << 18) ? kShellBlocks : 1;
    const dim3 grid((unsigned)((nb + bpw * (kBlock / 64) - 1) / (bpw * (kBlock / 64)))), block(kBlock);
    switch (P.V <= 4 ? P.V : 0) {
    case 1: hipLaunchKernelGGL((grid_shell_flag_fast_kernel<1, TILED>), grid, block, 0, s, P, dist_thr, ballots, counts, bpw); break;
    case 2: hipLaunchKernelGGL((grid_shell_flag_fast_kernel<2, TILED>), grid, block, 0, s, P, dist_thr, ballots, counts, bpw); break;
    case 3: hipLaunchKernelGGL((grid_shell_flag_fast_kernel<3, TILED>), grid, block, 0, s, P, dist_thr, ballots, counts, bpw); break;
    case 4: hipLaunchKernelGGL((grid_shell_flag_fast_kernel<4, TILED>), grid, block, 0, s, P, dist_thr, ballots, counts, bpw); break;
    default: hipLaunchKernelGGL((grid_shell_flag_fast_kernel<0, TILED>), grid, block, 0, s, P, dist_thr, ballots, counts, bpw); break;
    }
}

// Pass 2 (after an exclusive scan of the workgroup counts): survivors are written in ascending flat index --
// the order of the reference's boolean-mask indexing -- so no sort is needed afterwards.
__global__ __launch_bounds__(kBlock) void grid_shell_write_kernel(const unsigned long long *__restrict__ ballots,
                                                                 const uint32_t *__restrict__ block_offsets, int64_t n,
                                                                 int64_t capacity, int64_t *__restrict__ idx_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long *bw = ballots + (int64_t)blockIdx.x * (kBlock / 64);
    int before = 0;
    for (int w = 0; w < wave; ++w) before += __popcll(bw[w]);
    const unsigned long long mine = bw[wave];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n && ((mine >> lane) & 1ull)) {
        const int64_t slot = (int64_t)block_offsets[blockIdx.x] + before + __popcll(mine & ((1ull << lane) - 1ull));
        if (slot < capacity) idx_out[slot] = i;
    }
}

__global__ void grid_shell_total_kernel(const uint32_t *__restrict__ block_counts, const uint32_t *__restrict__ block_offsets,
                                        int64_t nb, unsigned long long *__restrict__ count)
{
    *count = (unsigned long long)block_offsets[nb - 1] + block_counts[nb - 1];
}

int64_t grid_shell_workspace_bytes(int64_t n)
{
    const int64_t nb = (n + kBlock - 1) / kBlock;
    return nb * (kBlock / 64) * 8 + 2 * ((nb * 4 + 255) / 256 * 256) + scan_scratch_bytes(nb);   // ballots + counts + offsets + scan scratch
}

// tiled_scratch: nullptr, or depth_tiled_bytes(V, H, W) bytes BEHIND the workspace proper (d3f_grid_shell: the caller handed
// over d3f_grid_shell_workspace_bytes + d3f_eval_dist_workspace_bytes)
hipError_t launch_grid_shell(const float *depth, const float *K, const float *pose, int V, int H, int W, const float *gx,
                             const float *gy, const float *gz, int nx, int ny, int nz, float mu, float dist_thr,
                             int64_t capacity, int64_t *idx_out, unsigned long long *count, void *workspace, hipStream_t s,
                             float *tiled_scratch)
{
    const int64_t n = (int64_t)nx * ny * nz;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0) return e;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    unsigned char *base = static_cast<unsigned char *>(workspace);
    unsigned long long *ballots = reinterpret_cast<unsigned long long *>(base);
    const size_t seg = (size_t)((nb * 4 + 255) / 256 * 256);
    uint32_t *counts = reinterpret_cast<uint32_t *>(base + nb * (kBlock / 64) * 8);
    uint32_t *offsets = reinterpret_cast<uint32_t *>(base + nb * (kBlock / 64) * 8 + seg);
    void *scratch = base + nb * (kBlock / 64) * 8 + 2 * seg;
    if (V <= 8 && n < 0xffffffffLL) {
        EvalParams P;
        memset(&P, 0, sizeof(P));
        P.depth = depth; P.K = K; P.pose = pose; P.V = V; P.H = H; P.W = W; P.mu = mu; P.n = n;
        P.grid_x = gx; P.grid_y = gy; P.grid_z = gz; P.grid_ny = ny; P.grid_nz = nz;
        if (tiled_scratch && n >= kDistTiledMin && depth_tiled_bytes(V, H, W) < (1LL << 32)) {
            P.depth_tw = (W + 3) / 4; P.depth_th = (H + 7) / 8;
            e = launch_depth_tiles(P, tiled_scratch, s);
            if (e != hipSuccess) return e;
            P.depth_tiled = tiled_scratch;
            launch_shell_flag_fast<true>(P, dist_thr, ballots, counts, nb, s);
        } else {
            launch_shell_flag_fast<false>(P, dist_thr, ballots, counts, nb, s);
        }
    } else {
        hipLaunchKernelGGL(grid_shell_flag_kernel, dim3((unsigned)nb), dim3(kBlock), (size_t)V * 48, s, depth, K, pose, V, H, W, gx, gy,
                           gz, ny, nz, n, mu, dist_thr, ballots, counts);
    }
    e = launch_exclusive_scan_u32(counts, offsets, nb, scratch, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(grid_shell_write_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, ballots, offsets, n, capacity, idx_out);
    hipLaunchKernelGGL(grid_shell_total_kernel, dim3(1), dim3(1), 0, s, counts, offsets, nb, count);
    return hipGetLastError();
}

// ---- farthest point sampling (utils/my_utils.py:478-497 fps_np) ---------------------------------------
// dist_i = min over chosen of sqrt((dx*dx + dy*dy) + dz*dz)   (float32, numpy's summation order, no fma)
// next   = first index of the maximum.
// k sequential rounds over n points.  Round 2's kernel was ONE 1024-thread workgroup on one of 256 CUs (7.2 ms for 100
// of 200 k points).  Here every round is one launch over up to 256 workgroups: a workgroup first merges the previous
// round's per-workgroup maxima (value, first index) into the chosen point -- every workgroup redundantly, a few KB of
// L2 reads --, then updates its contiguous chunk of `dist` and leaves its own maximum for the next round; maxima are
// double-buffered by round parity, and launch k only merges (fps_np's third return value, dist.max()).  The comparisons
// and their tie rule (larger value, else smaller index) are those of np.argmax, so the index sequence is identical.
// k + 1 small dependent launches: ~2 us each back to back or in a HIP graph (the stream is the caller's, nothing here
// synchronises).  Shared by the 3-D float32 variant and the integer-pixel variant of assoc_kernels.hip.
struct Fps3 {
    using dist_t = float;
    static constexpr int kDim = 3;
    using coord_t = float;
    __device__ static dist_t dist(const coord_t *p, int64_t i, coord_t c0, coord_t c1, coord_t c2)
    {
        const float dx = p[i * 3 + 0] - c0, dy = p[i * 3 + 1] - c1, dz = p[i * 3 + 2] - c2;
        return sqrtf((dx * dx + dy * dy) + dz * dz);
    }
    __device__ static dist_t lowest() { return -1.0f; }
    __device__ static dist_t vmin(dist_t a, dist_t b) { return fminf(a, b); }       // np.minimum on finite values
};
struct FpsPix {        // exact int64 squared distances order like numpy's float64 norms of integer differences (< 2^53)
    using dist_t = long long;
    static constexpr int kDim = 2;
    using coord_t = int32_t;
    __device__ static dist_t dist(const coord_t *p, int64_t i, coord_t c0, coord_t c1, coord_t)
    {
        const long long dy = (long long)p[i * 2 + 0] - c0, dx = (long long)p[i * 2 + 1] - c1;
        return dy * dy + dx * dx;
    }
    __device__ static dist_t lowest() { return -1; }
    __device__ static dist_t vmin(dist_t a, dist_t b) { return a < b ? a : b; }     // min on the squares == np.minimum on the norms
};

template <typename M>
struct FpsMax {
    typename M::dist_t v;
    long long i;
};

constexpr int kFpsMaxBlocks = 256;

int64_t fps_blocks(int64_t n)
{
    const int64_t per = n / kFpsMaxBlocks > 1024 ? (n + kFpsMaxBlocks - 1) / kFpsMaxBlocks : 1024;     // >= 1024 points per workgroup
    return (n + per - 1) / per;
}

int64_t fps_workspace_bytes(int64_t n, int dist_bytes)
{
    // n distances, then two arrays of per-workgroup maxima (16 bytes each), 16-byte aligned
    return (n * dist_bytes + 15) / 16 * 16 + 2 * kFpsMaxBlocks * 16;
}

template <typename M>
__global__ __launch_bounds__(kBlock) void fps_round_kernel(const typename M::coord_t *__restrict__ pts, int64_t n, int round, int k,
                                                          int64_t init_idx, int64_t *__restrict__ out_idx, void *out_maxdist,
                                                          typename M::dist_t *__restrict__ dist, FpsMax<M> *__restrict__ maxima, int nb)
{
    using D = typename M::dist_t;
    __shared__ D red_v[kBlock / 64];
    __shared__ long long red_i[kBlock / 64];
    __shared__ long long cur_s;
    __shared__ D best_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto better = [](D ov, long long oi, D v, long long i) { return ov > v || (ov == v && oi < i); };
    auto block_argmax = [&](D &bv, long long &bi) {
        for (int off = 32; off > 0; off >>= 1) {
            const D ov = __shfl_xor(bv, off, 64);
            const long long oi = __shfl_xor(bi, off, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            D v = red_v[0];
            long long ix = red_i[0];
            for (int w = 1; w < kBlock / 64; ++w)
                if (better(red_v[w], red_i[w], v, ix)) { v = red_v[w]; ix = red_i[w]; }
            cur_s = ix; best_s = v;
        }
        __syncthreads();
    };
    // 1. the point this round adds: the start, or the first maximum of `dist` after the previous round
    long long cur = init_idx;
    if (round > 0) {
        const FpsMax<M> *prev = maxima + (size_t)((round - 1) & 1) * kFpsMaxBlocks;
        D bv = M::lowest();
        long long bi = 0x7fffffffffffffffLL;
        for (int b = tid; b < nb; b += kBlock) {
            const FpsMax<M> m = prev[b];
            if (better(m.v, m.i, bv, bi)) { bv = m.v; bi = m.i; }
        }
        block_argmax(bv, bi);
        cur = cur_s;
        if (round == k) {                                   // the extra launch: fps_np's dist.max() after the last update
            if (blockIdx.x == 0 && tid == 0 && out_maxdist) {
                if constexpr (sizeof(D) == 4) *static_cast<float *>(out_maxdist) = (float)best_s;
                else *static_cast<double *>(out_maxdist) = sqrt((double)best_s);
            }
            return;
        }
    }
    if (blockIdx.x == 0 && tid == 0) out_idx[round] = cur;
    // 2. update this workgroup's chunk of dist, leave its maximum (first index on ties: ascending scan with '>')
    const typename M::coord_t c0 = pts[cur * M::kDim + 0], c1 = pts[cur * M::kDim + 1];
    typename M::coord_t c2 = 0;
    if constexpr (M::kDim > 2) c2 = pts[cur * M::kDim + 2];
    const int64_t per = (n + nb - 1) / nb, lo = (int64_t)blockIdx.x * per, hi = min(n, lo + per);
    D bv = M::lowest();
    long long bi = 0x7fffffffffffffffLL;
    for (int64_t i = lo + tid; i < hi; i += kBlock) {
        D d = M::dist(pts, i, c0, c1, c2);
        if (round > 0) d = M::vmin(dist[i], d);
        dist[i] = d;
        if (d > bv) { bv = d; bi = i; }
    }
    block_argmax(bv, bi);
    if (tid == 0) {
        FpsMax<M> m;
        m.v = best_s; m.i = cur_s;
        maxima[(size_t)(round & 1) * kFpsMaxBlocks + blockIdx.x] = m;
    }
}

template <typename M>
static hipError_t launch_fps_rounds(const typename M::coord_t *pts, int64_t n, int k, int64_t init_idx, int64_t *out_idx, void *out_maxdist,
                                    void *workspace, hipStream_t s)
{
    using D = typename M::dist_t;
    D *dist = static_cast<D *>(workspace);
    FpsMax<M> *maxima = reinterpret_cast<FpsMax<M> *>(static_cast<char *>(workspace) + (n * (int64_t)sizeof(D) + 15) / 16 * 16);
    const int nb = (int)fps_blocks(n);
    for (int round = 0; round <= k; ++round)
        hipLaunchKernelGGL(fps_round_kernel<M>, dim3(round == k ? 1 : nb), dim3(kBlock), 0, s, pts, n, round, k, init_idx, out_idx, out_maxdist,
                           dist, maxima, nb);
    return hipGetLastError();
}

hipError_t launch_fps(const float *pts, int64_t n, int k, int64_t init_idx, int64_t *out_idx, float *out_maxdist,
                      void *workspace, hipStream_t s)
{
    return launch_fps_rounds<Fps3>(pts, n, k, init_idx, out_idx, out_maxdist, workspace, s);
}

hipError_t launch_fps_pixels(const int32_t *pts, int64_t n, int k, int64_t init_idx, int64_t *out_idx, double *out_maxdist,
                             void *workspace, hipStream_t s)
{
    return launch_fps_rounds<FpsPix>(pts, n, k, init_idx, out_idx, out_maxdist, workspace, s);
}

}  // namespace d3f
