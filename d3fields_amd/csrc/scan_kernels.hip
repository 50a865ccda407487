// scan_kernels.hip -- exclusive prefix sum of uint32 counters (hand-written; used by the point ordering and by the
// order-preserving compactions).  2048 elements per 256-thread workgroup (8 per lane: lane-local scan, wave scan by
// DPP shuffles, four wave totals through LDS); the workgroup totals are scanned recursively and added back.
// In-place operation (in == out) is allowed: every lane reads its eight elements before it writes them.
#include "d3f_internal.h"

namespace d3f {

constexpr int kScanPerLane = 8;
constexpr int kScanPerBlock = kBlock * kScanPerLane;      // 2048

__global__ __launch_bounds__(kBlock) void scan_block_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                           uint32_t *__restrict__ block_sums, int64_t n)
{
    __shared__ uint32_t wave_tot[kBlock / 64];
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kScanPerLane;
    uint32_t v[kScanPerLane];
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k) v[k] = base + k < n ? in[base + k] : 0u;
    uint32_t lane_sum = 0u;
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k) { const uint32_t t = v[k]; v[k] = lane_sum; lane_sum += t; }   // exclusive inside the lane
    uint32_t incl = lane_sum;                                                                             // inclusive over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63) >= off) incl += up;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - lane_sum;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k)
        if (base + k < n) out[base + k] = v[k] + before;
    if (block_sums && threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = before + lane_sum;
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ block_offsets, int64_t n)
{
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kScanPerLane;
    const uint32_t add = block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k)
        if (base + k < n) out[base + k] += add;
}

// ---- the same scan in ONE launch (round 5: the point ordering of a 100 000-point cloud was seven launches of 4-12 us, three of
// them this scan) -- chained scan with decoupled look-back (Merrill & Garland, 2016): a workgroup takes the next tile from a
// ticket counter (so tiles start in order whatever the dispatcher does), scans it, publishes the tile's total, adds up the totals /
// inclusive prefixes its predecessors have published, and publishes its own inclusive prefix.  status[0] = the ticket counter,
// status[1 + t] = (flag << 30) | value with flag 1 = tile total, 2 = inclusive prefix; all zero at launch (the caller clears
// them: order_prepare_kernel).  Values < 2^30 (the ordering scans counts of < 2^31 / 2 points; the host checks).
__global__ __launch_bounds__(kBlock) void scan_lookback_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int64_t n,
                                                              uint32_t *__restrict__ status)
{
    __shared__ uint32_t wave_tot[kBlock / 64];
    __shared__ uint32_t tile_s, prefix_s;
    if (threadIdx.x == 0) tile_s = atomicAdd(&status[0], 1u);
    __syncthreads();
    const uint32_t tile = tile_s;
    const int64_t base = ((int64_t)tile * kBlock + threadIdx.x) * kScanPerLane;
    uint32_t v[kScanPerLane];
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k) v[k] = base + k < n ? in[base + k] : 0u;
    uint32_t lane_sum = 0u;
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k) { const uint32_t t = v[k]; v[k] = lane_sum; lane_sum += t; }
    uint32_t incl = lane_sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63) >= off) incl += up;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - lane_sum;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    // the last wave looks back 64 predecessors at a time (one status word per lane): everything from the nearest published
    // inclusive prefix up to tile - 1 is summed; a lane whose predecessor has published nothing yet makes the wave read again
    // (the status words ARE the data: nothing else a predecessor wrote is read here, so relaxed device-scope atomics suffice --
    //  acquire / release ones invalidate and write back the caches on every poll: 21 us instead of 9 for 2^19 counters)
    if (wave == kBlock / 64 - 1) {
        const int lane = threadIdx.x & 63;
        const uint32_t total = __shfl(before + lane_sum, 63, 64);
        uint32_t *st = status + 1;
        uint32_t prefix = 0u;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&st[tile], (1u << 30) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t hi = (int64_t)tile - 1;      // nearest predecessor not summed yet
            bool done = false;
            while (!done) {
                const int64_t j = hi - lane;
                uint32_t sw = 2u << 30;          // beyond tile 0: an inclusive prefix of zero
                if (j >= 0) sw = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long missing = __ballot((sw >> 30) == 0u), incl_at = __ballot((sw >> 30) == 2u);
                // usable lanes: below the first missing one; stop at the first inclusive prefix among them
                const int first_missing = missing ? __builtin_ctzll(missing) : 64;
                const int first_incl = incl_at ? __builtin_ctzll(incl_at) : 64;
                const int upto = first_incl < first_missing ? first_incl + 1 : first_missing;      // lanes [0, upto) are summed
                uint32_t part = lane < upto ? (sw & 0x3fffffffu) : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
                prefix += part;
                hi -= upto;
                done = first_incl < first_missing;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&st[tile], (2u << 30) | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prefix_s = prefix;
        }
    }
    __syncthreads();
    const uint32_t add = prefix_s + before;
#pragma unroll
    for (int k = 0; k < kScanPerLane; ++k)
        if (base + k < n) out[base + k] = v[k] + add;
}

int64_t scan_status_words(int64_t n) { return 1 + (n + kScanPerBlock - 1) / kScanPerBlock; }

// `status`: scan_status_words(n) zeroed words; total of the scanned values < 2^30
hipError_t launch_exclusive_scan_lookback_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *status, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int64_t nb = (n + kScanPerBlock - 1) / kScanPerBlock;
    hipLaunchKernelGGL(scan_lookback_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, in, out, n, status);
    return hipGetLastError();
}

int64_t scan_scratch_bytes(int64_t n)
{
    int64_t total = 0;
    while (n > kScanPerBlock) {
        n = (n + kScanPerBlock - 1) / kScanPerBlock;
        total += (n * 4 + 255) / 256 * 256;
    }
    return total + 256;
}

hipError_t launch_exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, void *scratch, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int64_t nb = (n + kScanPerBlock - 1) / kScanPerBlock;
    uint32_t *sums = static_cast<uint32_t *>(scratch);
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, in, out, nb > 1 ? sums : nullptr, n);
    if (nb > 1) {
        unsigned char *next = static_cast<unsigned char *>(scratch) + (nb * 4 + 255) / 256 * 256;
        hipError_t e = launch_exclusive_scan_u32(sums, sums, nb, next, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, out, sums, n);
    }
    return hipGetLastError();
}

}  // namespace d3f
