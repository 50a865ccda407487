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
