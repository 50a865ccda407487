// order_kernels.hip -- locality ordering of query points (performance only; results never
// depend on it because points are independent, fusion.py:305-394).
//
// Why: with maps larger than the caches (C2 dense: 1.89 GB vs 8 x 4 MiB L2 + 256 MiB Infinity
// Cache) the fused kernel is bound by texel re-fetches; points that are close in 3-D project
// close together in EVERY view, so walking the points along a space-filling curve keeps the in-flight
// texel footprint compact (measured 3.2 -> 2.8 ms on the 985 600-point grid, 3.4 -> 2.6 ms on a
// shuffled cloud).  Grids do not come here at all (closed-form brick walk, fuse_common.h); this is the path of clouds.
//
// Round 5: the curve is a HILBERT curve and the order inside a counting cell is exact.  Rounds 1-4 walked a Morton (Z)
// curve, whose consecutive cells are up to a whole parent cell apart at every octant boundary: 64 consecutive points of it
// -- one workgroup's tile -- then span a box several cells wide, and the LDS texel windows of the window kernel
// (fuse_window.hip) overflow their pool on 2 tiles of 3 (scripts/notebook/sim_cloud_tiles.py: C2-patch cloud, 66 % of the tiles over
// an 80-slot pool, 18 % of the valid (point, view) pairs outside their window; Hilbert: 19 % / 2.9 %), which is why the
// window kernel lost on clouds.  Consecutive cells of a Hilbert curve always share a face, at every level, so ANY run of
// consecutive points is a compact blob.  Any prefix of the key is still a valid coarser cell (the curve is hierarchical).
//
// Hand-written since round 2 (round 1 sorted (key, index) pairs with rocPRIM's Onesweep radix sort: histogram +
// 3-4 digit passes + 7 memset launches = 0.12-0.16 ms per 1 M points).  A counting sort by cell plus an exact rank inside the
// cell gives the total order of the keys:
//   0. order_prepare_kernel  counters / gate / scan status cleared AND the box of the finite points as <= 128 partial boxes (round 6:
//                            one launch, no atomics; round 5: two): the 512^3 key grid is laid over that box
//   1. cell_count_kernel     27-bit Hilbert key of the grid cell of every point (Skilling's transpose form; kept for step 4)
//                            and a histogram of a 15..20-bit prefix = the counting cell (~4 x 4 x 8 key cells at 1 M points; counters
//                            in caller scratch, cleared by order_clear_kernel); the returning atomic also gives the point its
//                            arrival rank inside the cell -- the only atomic per point
//   2. exclusive scan of the counters (scan_kernels.hip: ONE launch, chained scan with decoupled look-back; rounds 2-4: three)
//   3. scatter_kernel        index i and its key go to slot offset[cell] + rank -- cells in curve order, arrival order inside
//   4. cell_rank_kernel      every slot counts the (key, index) pairs of ITS cell that sort before its own (a cell is a few
//                            consecutive slots: L1 hits) and moves to that place: the exact order of the full key, ties by
//                            index.  A cell of more than 256 points (a dense clump) is ranked in aligned pieces of 256 slots.
// Five launches (seven up to round 4, six in round 5); 1 M points: 0.10 ms of kernel time (0.135 with the fixed 4-mm key grid this round began with,
// whose returning atomics serialised on the ~33 points of an occupied 16-mm cell).  The key kernel runs at the rate the L2s retire
// returning atomics (~21 G/s).  Measured and not kept (round 5, scripts/notebook/gpu_sessions/r5_gpu20-22, 26, 27.sh): on the fixed
// grid fewer counters (2^19 / 2^18: the atomics 130 us, the rank loops 72-110 us) -- on the box-relative grid 2^20 .. 2^18 are
// within 1 % of each other and 2^21 only lengthens the scan; counters private to an XCD (eight planes indexed by the hardware XCC
// id, workgroup-scope atomics: no faster -- the limit is not coherence traffic -- and 3 bits of cell resolution lost); four points
// per lane in the key kernel (47 vs 47 us).  The result is a deterministic function of the points (the arrival order does not
// survive step 4 in cells of <= 256 points).
#include "d3f_internal.h"

namespace d3f {

__device__ __forceinline__ uint32_t spread3(uint32_t x)
{
    x &= 0x3ffu;                      // up to 10 bits per axis
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

// Index of the cell (x, y, z), 9 bits per axis, along the 3-D Hilbert curve: J. Skilling, "Programming the Hilbert
// curve", AIP Conf. Proc. 707 (2004) -- axes to transpose (an inverse-undo pass of the per-level reflections / swaps, then
// a Gray decode), then the transpose's bits interleaved with axis 0 the most significant of every 3-bit digit.
__device__ __forceinline__ uint32_t hilbert27(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t X[3] = {x & 511u, y & 511u, z & 511u};
#pragma unroll
    for (uint32_t Q = 256u; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t; X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    uint32_t t = 0u;
#pragma unroll
    for (uint32_t Q = 256u; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (spread3(X[0]) << 2) | (spread3(X[1]) << 1) | spread3(X[2]);
}

constexpr float kFineCell = 0.004f;       // the coarsest fine cell: 4 mm x 512 per axis = 2.05 m (wider clouds: fixed grid, keys wrap)
// 27-bit fine key = key of the counting cell (15 .. 20 bits, chosen per call; the scratch is sized for 21) << shift | sub-cell
constexpr int64_t kCells = 1 << 21;
constexpr int kExactCell = 256;           // cells of more points than this are ranked in aligned pieces of this many slots

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- the cloud's bounding box (round 5) -----------------------------------------------------------------------------------------
// The keys of rounds 1-4 quantised the coordinates on a FIXED 4-mm grid of 512^3 cells (2.05 m): a cloud in the reference's
// 0.8 x 0.7 x 0.22 m workspace occupies 1.4 % of that key space, its ~30 k occupied 16-mm counting cells take ~33 points each,
// and the returning atomics of those points serialise on their counter (63 us per 1 M points, the largest part of the ordering).
// Now the grid is laid over the cloud's own box: the longest side is cut into 511 cells (never coarser than 4 mm; a cloud wider
// than 2.04 m keeps the fixed grid), so 2^20 counters resolve ~6 x 6 x 12-mm cells of ~4 points.  The box is the min / max over the
// FINITE coordinates (|x| < 1e6), kept in six words after the gate as order-preserving unsigned images of the floats.
__device__ __forceinline__ uint32_t ordered_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // monotone: a < b  <=>  ordered_bits(a) < ordered_bits(b)
}
__device__ __forceinline__ float from_ordered_bits(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

constexpr int kBoxBlock = 1024;          // few, large workgroups (each leaves ONE partial box)
constexpr int kBoxParts = 128;           // at most this many of them
constexpr int kBoxLanes = 6;             // words of a partial box: lo x y z, hi x y z (ordered_bits images)

// ONE launch in front of the key kernel (round 6; rounds 2-5: order_clear_kernel, then order_bbox_kernel with six same-address
// atomics per workgroup on words the first kernel had to initialise): workgroups [0, clear_blocks) zero the counting table, the gate
// words and the status words of the single-launch scan; workgroups [clear_blocks, clear_blocks + parts) each reduce a strided share
// of the points to a partial box and STORE it to parts[b] -- no atomics, nothing to initialise, so both roles run side by side.
// The key kernel folds the <= 128 partial boxes itself (one wave, L2 hits).
__global__ __launch_bounds__(kBoxBlock) void order_prepare_kernel(uint32_t *__restrict__ table, uint32_t *__restrict__ gate,
                                                                 uint32_t *__restrict__ status, int status_words, int clear_blocks,
                                                                 const float *__restrict__ pts, int64_t n, uint32_t *__restrict__ parts)
{
    if ((int)blockIdx.x < clear_blocks) {
        reinterpret_cast<uint4 *>(table)[(int64_t)blockIdx.x * kBoxBlock + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0 && threadIdx.x < 4) gate[threadIdx.x] = 0u;       // the window / cell-run gate of this order (fuse_window.hip)
        if ((int)blockIdx.x == clear_blocks - 1)                              // the status words of the single-launch scan
            for (int k = threadIdx.x; k < status_words; k += kBoxBlock) status[k] = 0u;
        return;
    }
    const int b = (int)blockIdx.x - clear_blocks, nparts = (int)gridDim.x - clear_blocks;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int64_t stride = (int64_t)nparts * kBoxBlock;
    for (int64_t i = (int64_t)b * kBoxBlock + threadIdx.x; i < n; i += 4 * stride) {
        float x[4][3];                                                     // four points in flight (past the end: the last point again)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t ij = i + j * stride < n ? i + j * stride : n - 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) x[j][k] = pts[ij * 3 + k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (fabsf(x[j][k]) < 1e6f) { lo[k] = fminf(lo[k], x[j][k]); hi[k] = fmaxf(hi[k], x[j][k]); }       // false for NaN / Inf
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    __shared__ float part[kBoxBlock / 64][6];
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) { part[threadIdx.x >> 6][k] = lo[k]; part[threadIdx.x >> 6][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = part[0][k], h = part[0][3 + k];
#pragma unroll
        for (int w = 1; w < kBoxBlock / 64; ++w) { l = fminf(l, part[w][k]); h = fmaxf(h, part[w][3 + k]); }
        // a workgroup that saw no finite coordinate leaves the empty box (lo = +inf, hi = -inf): the fold's min / max drop it
        parts[b * kBoxLanes + k] = ordered_bits(l);
        parts[b * kBoxLanes + 3 + k] = ordered_bits(h);
    }
}

// origin and cells-per-metre of the key grid, from the box (every lane computes the same values from six uniform loads)
struct KeyGrid { float ox, oy, oz, inv; };
// (every workgroup folds the partial boxes by itself: wave 0, two partial boxes per lane, six DPP-free shuffle reductions; the result
// is the same in every workgroup -- min / max are exact and order-free)
__device__ __forceinline__ KeyGrid key_grid(const uint32_t *__restrict__ parts, int nparts)
{
    __shared__ KeyGrid g_s;
    if (threadIdx.x < 64) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (parts)
            for (int b = threadIdx.x; b < nparts; b += 64)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    lo[k] = fminf(lo[k], from_ordered_bits(parts[b * kBoxLanes + k]));
                    hi[k] = fmaxf(hi[k], from_ordered_bits(parts[b * kBoxLanes + 3 + k]));
                }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
            }
        if (threadIdx.x == 0) {
            KeyGrid g = {0.0f, 0.0f, 0.0f, 1.0f / kFineCell};               // the fixed 4-mm grid (keys wrap beyond 2.05 m: harmless)
            const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
            if (parts && ext > 0.0f && ext <= 511.0f * kFineCell) { g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2]; g.inv = 511.0f / ext; }    // false for NaN / no finite point
            g_s = g;
        }
    }
    __syncthreads();
    return g_s;
}

// MORTON: the Z-curve keys of rounds 1-4 (experiments builds keep them for same-box comparisons)
// (four points per lane -- all loads, then all returning atomics in flight together -- changes nothing: 47 vs 47 us per 1 M points,
// session 27.  The kernel runs at the rate the L2s retire returning atomics, ~21 G/s.)
template <bool MORTON>
__global__ __launch_bounds__(kBlock) void cell_count_kernel(const float *__restrict__ pts, int64_t n, uint32_t *__restrict__ keys,
                                                           uint32_t *__restrict__ ranks, uint32_t *__restrict__ table, int shift,
                                                           const uint32_t *__restrict__ parts, int nparts)
{
    const KeyGrid g = key_grid(parts, nparts);          // (whole workgroup: a barrier inside)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // non-finite / huge coordinates just land in some cell: only locality is at stake
    const int qx = (int)fminf(fmaxf(floorf((pts[i * 3 + 0] - g.ox) * g.inv), -1e9f), 1e9f);
    const int qy = (int)fminf(fmaxf(floorf((pts[i * 3 + 1] - g.oy) * g.inv), -1e9f), 1e9f);
    const int qz = (int)fminf(fmaxf(floorf((pts[i * 3 + 2] - g.oz) * g.inv), -1e9f), 1e9f);
    const uint32_t key = MORTON ? (spread3((uint32_t)qx & 511u) | (spread3((uint32_t)qy & 511u) << 1) | (spread3((uint32_t)qz & 511u) << 2))
                                : hilbert27((uint32_t)qx, (uint32_t)qy, (uint32_t)qz);
    keys[i] = key;
    ranks[i] = atomicAdd(&table[key >> shift], 1u);
}

// slot = (key << 32) | index: ONE 8-byte store per point (two 4-byte arrays were two partial-sector writes: 25 -> 14 us per 1 M
// points), and the rank kernel compares the words as they are
__global__ __launch_bounds__(kBlock) void scatter_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ ranks, int64_t n,
                                                        const uint32_t *__restrict__ offsets, unsigned long long *__restrict__ slots, int shift)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i];
    slots[offsets[key >> shift] + ranks[i]] = ((unsigned long long)key << 32) | (uint32_t)i;
}

// One lane per slot: its place among the (key, index) words of its own counting cell.
__global__ __launch_bounds__(kBlock) void cell_rank_kernel(const unsigned long long *__restrict__ slots, int64_t n,
                                                          const uint32_t *__restrict__ offsets, int64_t cells, int shift,
                                                          uint32_t *__restrict__ order)
{
    const int64_t pos = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (pos >= n) return;
    const unsigned long long mine = slots[pos];                 // (key, index): a strict total order
    const int64_t c = (int64_t)(mine >> (32 + shift));
    int64_t s = offsets[c], e = c + 1 < cells ? (int64_t)offsets[c + 1] : n;
    if (e - s > kExactCell) {                      // a clump: aligned pieces of kExactCell slots, each ranked by itself
        const int64_t lo = pos / kExactCell * kExactCell;
        s = max(s, lo); e = min(e, lo + kExactCell);
    }
    int rank = 0;
    for (int64_t j = s; j < e; ++j) rank += slots[j] < mine ? 1 : 0;
    order[s + rank] = (uint32_t)mine;
}

int64_t order_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    // keys | ranks | order | slots (8-byte (key, index) words: two segments) | cell counters + scan scratch
    // | 256 bytes of gate words | the partial boxes of order_prepare_kernel
    return (int64_t)(5 * align_up((size_t)n * 4, 256) + (size_t)kCells * 4 + (size_t)scan_scratch_bytes(kCells) + 256 + (size_t)kBoxParts * kBoxLanes * 4);
}

int64_t order_gate_offset(int64_t n)
{
    return (int64_t)(5 * align_up((size_t)n * 4, 256) + (size_t)kCells * 4 + (size_t)scan_scratch_bytes(kCells));
}

uint32_t *order_gate_words(void *workspace, int64_t n)
{
    return reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(workspace) + order_gate_offset(n));
}

// Fills *order_out with a pointer (inside the workspace) to n uint32 indices in Hilbert-cell order.
// curve: 0 = Hilbert (the product's); experiments builds: bit 0 = Morton (the order of rounds 1-4), bit 1 = the three-launch scan,
// bit 2 = the fixed 4-mm key grid instead of the one laid over the cloud's box
hipError_t build_point_order(const float *pts, int64_t n, void *workspace, int64_t workspace_bytes,
                             const uint32_t **order_out, hipStream_t stream, int curve)
{
    *order_out = nullptr;
    if (n <= 0 || n > 0x7fffffffLL || workspace_bytes < order_workspace_bytes(n)) return hipErrorInvalidValue;
    const size_t seg = align_up((size_t)n * 4, 256);
    unsigned char *base = static_cast<unsigned char *>(workspace);
    uint32_t *keys = reinterpret_cast<uint32_t *>(base), *ranks = reinterpret_cast<uint32_t *>(base + seg);
    uint32_t *order = reinterpret_cast<uint32_t *>(base + 2 * seg);
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(base + 3 * seg);      // n 8-byte words: segments 3 and 4
    uint32_t *table = reinterpret_cast<uint32_t *>(base + 5 * seg);
    void *scan_scratch = base + 5 * seg + (size_t)kCells * 4;
    const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
    // counters: >= 2 per point (any prefix of the key is a valid coarser cell of the curve), 2^15 .. 2^20 -- clearing and scanning
    // the table must not dominate small batches, and with the grid laid over the cloud's box 2 per point already leave ~4 points
    // per occupied cell (rounds 2-4, fixed grid: 4 per point, fewer and the atomics collided)
    int bits = 15;
    while (bits < 20 && (1LL << bits) < 2 * n) bits += 1;      // 2 counters per point, at most 2^20 (session 26: 2^21 only lengthens the scan)
    if ((curve >> 8) >= 15 && (curve >> 8) <= 21) bits = curve >> 8;      // experiments builds only
    const int shift = 27 - bits;
    const int64_t cells = 1LL << bits;
    // (the scan scratch is sized for the recursive three-launch scan of 2^21 counters: 1024 + 1 words and more)
    uint32_t *status = static_cast<uint32_t *>(scan_scratch);
    const int status_words = (int)scan_status_words(cells);
    // ONE launch: the counters, gate and status words cleared AND the cloud's box as <= 128 partial boxes; then keys on a grid laid over
    // the box (experiments builds: bit 2 = the fixed 4-mm grid of rounds 1-4, bit 0 = Morton keys on it)
    uint32_t *parts = (curve & 5) ? nullptr : order_gate_words(workspace, n) + 64;
    int nparts = 0;
    if (parts) {
        nparts = (int)((n + (int64_t)kBoxBlock * 8 - 1) / ((int64_t)kBoxBlock * 8));
        if (nparts > kBoxParts) nparts = kBoxParts;
    }
    const int clear_blocks = (int)(cells / 4 / kBoxBlock);
    hipLaunchKernelGGL(order_prepare_kernel, dim3((unsigned)(clear_blocks + nparts)), dim3(kBoxBlock), 0, stream, table,
                       order_gate_words(workspace, n), status, status_words, clear_blocks, pts, n, parts);
    if (curve & 1) hipLaunchKernelGGL(cell_count_kernel<true>, dim3(nb), dim3(kBlock), 0, stream, pts, n, keys, ranks, table, shift, parts, nparts);
    else hipLaunchKernelGGL(cell_count_kernel<false>, dim3(nb), dim3(kBlock), 0, stream, pts, n, keys, ranks, table, shift, parts, nparts);
    // n < 2^31 points: the counts sum to n; the look-back scan carries 30-bit values -- above 2^30 points the recursive scan
    hipError_t e = (n < (1LL << 30) && !(curve & 2)) ? launch_exclusive_scan_lookback_u32(table, table, cells, status, stream)
                                   : launch_exclusive_scan_u32(table, table, cells, scan_scratch, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(kBlock), 0, stream, keys, ranks, n, table, slots, shift);
    hipLaunchKernelGGL(cell_rank_kernel, dim3(nb), dim3(kBlock), 0, stream, slots, n, table, cells, shift, order);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the finished order always sits in the same place, so that a later call can find it again (D3F_FLAG_REUSE_POINT_ORDER)
    *order_out = order;
    return hipSuccess;
}

// where build_point_order leaves the order inside a workspace of order_workspace_bytes(n)
const uint32_t *stored_point_order(void *workspace, int64_t n)
{
    return reinterpret_cast<const uint32_t *>(static_cast<unsigned char *>(workspace) + 2 * align_up((size_t)n * 4, 256));
}

// ---- does the caller's order have spatial locality? ------------------------------------------------
// out[0] = mean L1 step between consecutive points, out[1] = mean L1 distance between points n/2 apart, both
// over <= kProbeSamples evenly spaced positions (non-finite pairs are left out).  A voxel grid, a mesh or a
// depth-ordered cloud gives out[0] << out[1]; a shuffled / uniformly random cloud gives out[0] ~ out[1].
constexpr int kProbeSamples = 1024;

// One workgroup of 1024 lanes, one sample per lane, three 12-byte loads each (the pair of consecutive points and the
// far one), all in flight together: one memory round trip.  A single CU looks up one cache line per lane and load
// instruction, so the kernel's time is its count of (lane, load) pairs: 4096 samples read as 36 dword loads per lane
// were 37 k look-ups = 25 us -- a third of the Morton-order set-up of a 100 000-point cloud; this form has 3 k.
constexpr int kProbeBlock = 1024;
static_assert(kProbeSamples == kProbeBlock && (kProbeSamples & (kProbeSamples - 1)) == 0, "one sample per lane; the position is a shift");
struct __attribute__((packed, aligned(4))) ProbePoint { float x, y, z; };

__global__ __launch_bounds__(kProbeBlock) void point_locality_kernel(const float *__restrict__ pts, int64_t n, float *__restrict__ out)
{
    __shared__ float red[3][kProbeBlock / 64];
    float near_d = 0.0f, far_d = 0.0f, cnt = 0.0f;
    if (n >= 2) {
        // sample k of kProbeSamples evenly spaced positions in [0, n-2]
        const int64_t i = (int64_t)((uint64_t)threadIdx.x * (uint64_t)(n - 1) / (uint64_t)kProbeSamples);       // i + 1 <= n - 1
        const int64_t j = i + n / 2 - (i + n / 2 >= n ? n : 0);
        const ProbePoint *pp = reinterpret_cast<const ProbePoint *>(pts);
        const ProbePoint a = pp[i], b = pp[i + 1], c = pp[j];
        const float dn = fabsf(b.x - a.x) + fabsf(b.y - a.y) + fabsf(b.z - a.z);
        const float df = fabsf(c.x - a.x) + fabsf(c.y - a.y) + fabsf(c.z - a.z);
        if (dn < INFINITY && df < INFINITY) { near_d = dn; far_d = df; cnt = 1.0f; }      // false for NaN too
    }
    for (int off = 32; off > 0; off >>= 1) {
        near_d += __shfl_xor(near_d, off, 64);
        far_d += __shfl_xor(far_d, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = near_d; red[1][threadIdx.x >> 6] = far_d; red[2][threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[3] = {0.0f, 0.0f, 0.0f};
        for (int w = 0; w < kProbeBlock / 64; ++w) { t[0] += red[0][w]; t[1] += red[1][w]; t[2] += red[2][w]; }
        out[0] = t[2] > 0.0f ? t[0] / t[2] : 0.0f;
        out[1] = t[2] > 0.0f ? t[1] / t[2] : 0.0f;
    }
}

// ---- is the caller's array a z-fastest lattice (create_init_grid's layout, fusion.py:79-88)? --------------------------
// One workgroup: nz = first i with z[i] <= z[i-1] (the z axis restarts), ny = first j with y[j*nz] <= y[(j-1)*nz],
// nx = n / (ny*nz); then <= 4096 evenly spaced points are compared bit for bit with the axis values implied by the
// first column / row / plane.  out = (nx, ny, nz), or zeros when the array is not such a lattice.  The answer only
// steers the walk order of d3f_eval_lattice (coordinates are always read from pts), so a wrong answer costs time only.
constexpr int kProbeSpan = 1 << 16;      // axes longer than this are not recognised
constexpr int kProbeBlocks = 16;         // 16 x 256 threads: one sample per thread

// first i in [1, count) with !(a[i*stride] > a[(i-1)*stride]) for coordinate `axis`, or count; whole workgroup
__device__ __forceinline__ int first_restart(const float *__restrict__ pts, int64_t stride, int64_t count, int axis, int *first_s)
{
    const int span = (int)min((int64_t)kProbeSpan, count);
    if (threadIdx.x == 0) *first_s = span;
    __syncthreads();
    for (int base = 1; base < span; base += kBlock) {
        const int i = base + threadIdx.x;
        if (i < span && !(pts[(int64_t)i * stride * 3 + axis] > pts[(int64_t)(i - 1) * stride * 3 + axis])) atomicMin(first_s, i);
        __syncthreads();
        if (*first_s < base + kBlock) break;            // found inside this chunk (uniform: read after the barrier)
    }
    __syncthreads();
    const int found = *first_s;
    __syncthreads();
    return found;
}

// out[0..2] = dims (written by workgroup 0), out[3] |= 1 by any workgroup whose samples contradict the lattice.
// The caller zeroes out[3] before the launch and accepts the dims only when out[0] > 0 and out[3] == 0.
__global__ __launch_bounds__(kBlock) void lattice_probe_kernel(const float *__restrict__ pts, int64_t n, int32_t *__restrict__ out)
{
    __shared__ int first_s;
    int nz = n >= 2 ? first_restart(pts, 1, n, 2, &first_s) : 0;
    bool ok = nz >= 2 && n % nz == 0;
    int ny = ok ? first_restart(pts, nz, n / nz, 1, &first_s) : 0;
    ok = ok && ny >= 1 && (n / nz) % ny == 0 && (ny >= 2 || n / nz == 1);
    int64_t nx = 0;
    if (ok) {
        nx = n / ((int64_t)nz * ny);
        ok = nx >= 1 && nx <= 0x7fffffffLL;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = ok ? (int32_t)nx : 0; out[1] = ok ? ny : 0; out[2] = ok ? nz : 0;
    }
    if (!ok) return;
    const int64_t samples = min((int64_t)kProbeBlocks * kBlock, n);
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k < samples) {
        const int64_t i = samples > 1 ? (int64_t)((double)k * (double)(n - 1) / (double)(samples - 1)) : 0;
        const int64_t iz = i % nz, ixy = i / nz, iy = ixy % ny, ix = ixy / ny;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(pts);
        const bool same = p[i * 3 + 2] == p[iz * 3 + 2] && p[i * 3 + 1] == p[iy * nz * 3 + 1] &&
                          p[i * 3 + 0] == p[ix * ny * nz * 3 + 0];
        if (!same) atomicOr(&out[3], 1);
    }
}

// ---- both probes in ONE launch (round 6) ------------------------------------------------------------------------------------------
// A new query tensor used to cost lattice_probe_kernel + point_locality_kernel + a fill of their output words (three launches of
// ~5 us each in front of a 90-120 us query of a small cloud).  Here workgroups 0..15 are the lattice probe, workgroups 16..19 the
// locality probe (256 samples each), and every output word is WRITTEN by exactly one workgroup, so the buffer needs no clearing:
//   out[0..2] lattice dims (zeros: not a lattice)      out[8 + b], b < 16: 1 if workgroup b's samples contradict the dims
//   out[24 + 3 q .. +2], q < 4: sum of near steps, sum of far distances, number of finite samples (floats) of locality workgroup q
static_assert(D3F_PROBE_WORDS == 40, "points_probe_kernel writes words 0..3, 8..23 and 24..35");
constexpr int kLocalityBlocks = 4;
__global__ __launch_bounds__(kBlock) void points_probe_kernel(const float *__restrict__ pts, int64_t n, int32_t *__restrict__ out)
{
    __shared__ int first_s;
    __shared__ float red[3][kBlock / 64];
    if (blockIdx.x >= kProbeBlocks) {
        const int q = (int)blockIdx.x - kProbeBlocks;
        float near_d = 0.0f, far_d = 0.0f, cnt = 0.0f;
        if (n >= 2) {
            const int sample = q * kBlock + (int)threadIdx.x;                  // one of kLocalityBlocks * kBlock = kProbeSamples positions
            const int64_t i = (int64_t)((uint64_t)sample * (uint64_t)(n - 1) / (uint64_t)(kLocalityBlocks * kBlock));
            const int64_t j = i + n / 2 - (i + n / 2 >= n ? n : 0);
            const ProbePoint *pp = reinterpret_cast<const ProbePoint *>(pts);
            const ProbePoint a = pp[i], b = pp[i + 1], c = pp[j];
            const float dn = fabsf(b.x - a.x) + fabsf(b.y - a.y) + fabsf(b.z - a.z);
            const float df = fabsf(c.x - a.x) + fabsf(c.y - a.y) + fabsf(c.z - a.z);
            if (dn < INFINITY && df < INFINITY) { near_d = dn; far_d = df; cnt = 1.0f; }      // false for NaN too
        }
        for (int off = 32; off > 0; off >>= 1) {
            near_d += __shfl_xor(near_d, off, 64);
            far_d += __shfl_xor(far_d, off, 64);
            cnt += __shfl_xor(cnt, off, 64);
        }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = near_d; red[1][threadIdx.x >> 6] = far_d; red[2][threadIdx.x >> 6] = cnt; }
        __syncthreads();
        if (threadIdx.x < 3) {
            float t = 0.0f;
            for (int w = 0; w < kBlock / 64; ++w) t += red[threadIdx.x][w];
            reinterpret_cast<float *>(out)[24 + 3 * q + threadIdx.x] = t;
        }
        return;
    }
    int nz = n >= 2 ? first_restart(pts, 1, n, 2, &first_s) : 0;
    bool ok = nz >= 2 && n % nz == 0;
    int ny = ok ? first_restart(pts, nz, n / nz, 1, &first_s) : 0;
    ok = ok && ny >= 1 && (n / nz) % ny == 0 && (ny >= 2 || n / nz == 1);
    int64_t nx = 0;
    if (ok) {
        nx = n / ((int64_t)nz * ny);
        ok = nx >= 1 && nx <= 0x7fffffffLL;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = ok ? (int32_t)nx : 0; out[1] = ok ? ny : 0; out[2] = ok ? nz : 0; out[3] = 0;
    }
    int bad = 0;
    if (ok) {
        const int64_t samples = min((int64_t)kProbeBlocks * kBlock, n);
        const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (k < samples) {
            const int64_t i = samples > 1 ? (int64_t)((double)k * (double)(n - 1) / (double)(samples - 1)) : 0;
            const int64_t iz = i % nz, ixy = i / nz, iy = ixy % ny, ix = ixy / ny;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(pts);
            const bool same = p[i * 3 + 2] == p[iz * 3 + 2] && p[i * 3 + 1] == p[iy * nz * 3 + 1] &&
                              p[i * 3 + 0] == p[ix * ny * nz * 3 + 0];
            bad = same ? 0 : 1;
        }
    }
    const int any_bad = __syncthreads_or(bad);          // (uniform control flow: `ok` is the same in every lane)
    if (threadIdx.x == 0) out[8 + blockIdx.x] = any_bad ? 1 : 0;
}

hipError_t launch_points_probe(const float *pts, int64_t n, int32_t *out, hipStream_t stream)
{
    hipLaunchKernelGGL(points_probe_kernel, dim3(kProbeBlocks + kLocalityBlocks), dim3(kBlock), 0, stream, pts, n, out);
    return hipGetLastError();
}

hipError_t launch_lattice_probe(const float *pts, int64_t n, int32_t *out_dims, hipStream_t stream)
{
    hipLaunchKernelGGL(lattice_probe_kernel, dim3(kProbeBlocks), dim3(kBlock), 0, stream, pts, n, out_dims);
    return hipGetLastError();
}

// one workgroup (a few microseconds); out: 2 device floats
hipError_t launch_point_locality(const float *pts, int64_t n, float *out, hipStream_t stream)
{
    hipLaunchKernelGGL(point_locality_kernel, dim3(1), dim3(kProbeBlock), 0, stream, pts, n, out);
    return hipGetLastError();
}

}  // namespace d3f
