// order_kernels.hip -- locality ordering of query points (performance only; results never
// depend on it because points are independent, fusion.py:305-394).
//
// Why: with maps larger than the caches (C2 dense: 1.89 GB vs 8 x 4 MiB L2 + 256 MiB Infinity
// Cache) the fused kernel is bound by texel re-fetches; points that are close in 3-D project
// close together in EVERY view, so walking the points along a Morton curve keeps the in-flight
// texel footprint compact (measured 3.2 -> 2.8 ms on the 985 600-point grid, 3.4 -> 2.6 ms on a
// shuffled cloud).  Grids do not come here at all (closed-form brick walk, fuse_eval.hip); this is the path of clouds.
//
// Hand-written since round 2 (round 1 sorted (key, index) pairs with rocPRIM's Onesweep radix sort: histogram +
// 3-4 digit passes + 7 memset launches = 0.12-0.16 ms per 1 M points).  The fused kernel needs locality, not a total
// order, so a counting sort by cell plus a local refinement is enough:
//   1. morton_count_kernel   27-bit Morton key of the 4-mm cell of every point (kept for step 4) and a histogram of its
//                            top 21 bits = the 16-mm cell (2 M counters in caller scratch, cleared by order_clear_kernel);
//                            the returning atomic also gives the point its arrival rank inside the cell -- the only
//                            atomic per point (device-scope atomics run at ~17 per ns chip-wide: 59 us per 1 M)
//   2. exclusive scan of the counters (scan_kernels.hip, three small launches)
//   3. scatter_kernel        index i goes to slot offset[cell] + rank -- cells in Morton order, arrival order inside
//   4. window_rank_kernel    every wave re-orders its 64 consecutive slots by (27-bit key, index): inside a 16-mm
//                            cell (~30 points at 1 M points per 0.12 m^3) the walk follows the 4-mm sub-cells, which is
//                            what the cell-run gather lives on; runs of equal cells that straddle a window are simply
//                            refined piecewise.
// Seven launches; the order inside a cell depends on atomic arrival, the results do not.
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

constexpr float kFineCell = 0.004f;       // 4-mm sub-cells x 512 per axis = 2.05 m before the keys wrap (harmless)
// 27-bit fine key = key of the counting cell (15 .. 21 bits, chosen per call) << shift | sub-cell
constexpr int64_t kCells = 1 << 21;

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__global__ __launch_bounds__(kBlock) void order_clear_kernel(uint32_t *__restrict__ table)
{
    reinterpret_cast<uint4 *>(table)[(int64_t)blockIdx.x * kBlock + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(kBlock) void morton_count_kernel(const float *__restrict__ pts, int64_t n, uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ ranks, uint32_t *__restrict__ table, int shift)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float inv = 1.0f / kFineCell;
    // non-finite / huge coordinates just land in some cell: only locality is at stake
    const int qx = (int)fminf(fmaxf(floorf(pts[i * 3 + 0] * inv), -1e9f), 1e9f);
    const int qy = (int)fminf(fmaxf(floorf(pts[i * 3 + 1] * inv), -1e9f), 1e9f);
    const int qz = (int)fminf(fmaxf(floorf(pts[i * 3 + 2] * inv), -1e9f), 1e9f);
    const uint32_t key = spread3((uint32_t)qx & 511u) | (spread3((uint32_t)qy & 511u) << 1) | (spread3((uint32_t)qz & 511u) << 2);
    keys[i] = key;
    ranks[i] = atomicAdd(&table[key >> shift], 1u);
}

__global__ __launch_bounds__(kBlock) void scatter_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ ranks, int64_t n,
                                                        const uint32_t *__restrict__ offsets, uint32_t *__restrict__ slots, int shift)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    slots[offsets[keys[i] >> shift] + ranks[i]] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void window_rank_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ slots,
                                                            int64_t n, uint32_t *__restrict__ order)
{
    const int64_t base = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64;
    const int lane = threadIdx.x & 63;
    const int64_t pos = base + lane;
    const bool live = pos < n;
    const uint32_t idx = live ? slots[pos] : 0xffffffffu;
    const uint32_t key = live ? keys[idx] : 0xffffffffu;
    const unsigned long long mine = ((unsigned long long)key << 32) | idx;      // (key, index): a strict total order
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const unsigned long long other = __shfl(mine, j, 64);
        rank += other < mine ? 1 : 0;
    }
    if (live) order[base + rank] = idx;            // dead lanes hold the maximum and rank last
}

int64_t order_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    // keys | ranks | order | slots | cell counters + scan scratch
    return (int64_t)(4 * align_up((size_t)n * 4, 256) + (size_t)kCells * 4 + (size_t)scan_scratch_bytes(kCells));
}

// Fills *order_out with a pointer (inside the workspace) to n uint32 indices in Morton-cell order.
hipError_t build_point_order(const float *pts, int64_t n, void *workspace, int64_t workspace_bytes,
                             const uint32_t **order_out, hipStream_t stream, int /*fine: the walk is always refined to 4 mm*/)
{
    *order_out = nullptr;
    if (n <= 0 || n > 0x7fffffffLL || workspace_bytes < order_workspace_bytes(n)) return hipErrorInvalidValue;
    const size_t seg = align_up((size_t)n * 4, 256);
    unsigned char *base = static_cast<unsigned char *>(workspace);
    uint32_t *keys = reinterpret_cast<uint32_t *>(base), *ranks = reinterpret_cast<uint32_t *>(base + seg);
    uint32_t *order = reinterpret_cast<uint32_t *>(base + 2 * seg), *slots = reinterpret_cast<uint32_t *>(base + 3 * seg);
    uint32_t *table = reinterpret_cast<uint32_t *>(base + 4 * seg);
    void *scan_scratch = base + 4 * seg + (size_t)kCells * 4;
    const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
    // counters: >= 4 per point (any prefix of the Morton key is a valid coarser Z-order cell), 2^15 .. 2^21: 16-mm cells
    // for 1 M points, 16 x 32 x 32 mm for 100 k -- the 64-slot refinement below still resolves them, and clearing +
    // scanning the table stops dominating small batches (fewer counters than that and the atomics start to collide:
    // 2.6 per point measured 7 -> 14 us in morton_count_kernel at 100 k points)
    int bits = 15;
    while (bits < 21 && (1LL << bits) < 4 * n) bits += 1;
    const int shift = 27 - bits;
    const int64_t cells = 1LL << bits;
    hipLaunchKernelGGL(order_clear_kernel, dim3((unsigned)(cells / 4 / kBlock)), dim3(kBlock), 0, stream, table);
    hipLaunchKernelGGL(morton_count_kernel, dim3(nb), dim3(kBlock), 0, stream, pts, n, keys, ranks, table, shift);
    hipError_t e = launch_exclusive_scan_u32(table, table, cells, scan_scratch, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(kBlock), 0, stream, keys, ranks, n, table, slots, shift);
    hipLaunchKernelGGL(window_rank_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, keys, slots, n, order);
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
