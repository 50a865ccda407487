// Micro-benchmark: what rate can the memory system deliver to gathers of the fused kernel's shape?
// Every half-wave (32 lanes x 16 B) reads one random, 512-byte-aligned 512-B segment per load instruction -- a texel
// third of a 384-channel fp32 map -- with 12 independent loads in flight per lane and 4 waves per SIMD, from a buffer of
// S bytes: S <= 2 MiB per XCD-local working set hits L2, tens of MB hit the Infinity Cache, GBs go to HBM.
// The vector L1 (32 KiB) never hits.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.cpp -o /tmp/ubench_gather && /tmp/ubench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int SEG_LANES>
__global__ __launch_bounds__(256, 4) void gather_kernel(const char *__restrict__ buf, uint32_t seg_mask, int iters, float *__restrict__ out)
{
    const uint32_t group = (blockIdx.x * 256u + threadIdx.x) / SEG_LANES;      // lanes sharing a segment
    const uint32_t lane = threadIdx.x % SEG_LANES;
    f32x4 acc = (f32x4)0.0f;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t seg = mix(group * 7919u + (uint32_t)it * 12u + (uint32_t)k) & seg_mask;
            v[k] = *reinterpret_cast<const f32x4 *>(buf + (size_t)seg * (SEG_LANES * 16) + lane * 16);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) acc += v[k];
    }
    if (acc.x == 123.456f) out[group] = acc.y;      // never true: keeps the loads alive
}

template <int SEG_LANES> static double run(const char *dbuf, size_t bytes, float *dout, int iters)
{
    const uint32_t nseg = (uint32_t)(bytes / (SEG_LANES * 16));
    const uint32_t mask = nseg - 1;                 // bytes is a power of two
    const int blocks = 256 * 4 * 8;                 // 32 workgroups per CU in total, 4 resident at a time
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(gather_kernel<SEG_LANES>, dim3(blocks), dim3(256), 0, 0, dbuf, mask, 2, dout);
    hipEventRecord(a);
    hipLaunchKernelGGL(gather_kernel<SEG_LANES>, dim3(blocks), dim3(256), 0, 0, dbuf, mask, iters, dout);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, a, b);
    const double moved = (double)blocks * 256 * 16.0 * 12 * iters;
    return moved / (ms * 1e-3) / 1e12;
}

int main()
{
    const size_t maxb = 4ull << 30;
    char *dbuf = nullptr;
    float *dout = nullptr;
    if (hipMalloc(&dbuf, maxb) != hipSuccess || hipMalloc(&dout, 64 << 20) != hipSuccess) return 1;
    hipMemset(dbuf, 0, maxb);
    printf("%-12s %-22s %-22s\n", "buffer", "512-B segments TB/s", "128-B segments TB/s");
    for (size_t bytes : {1ull << 20, 8ull << 20, 16ull << 20, 64ull << 20, 128ull << 20, 512ull << 20, 2ull << 30, 4ull << 30}) {
        const double t512 = run<32>(dbuf, bytes, dout, 40);
        const double t128 = run<8>(dbuf, bytes, dout, 40);
        printf("%8.0f MiB %-22.2f %-22.2f\n", bytes / 1048576.0, t512, t128);
    }
    return 0;
}
