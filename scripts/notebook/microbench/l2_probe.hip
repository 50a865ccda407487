// l2_probe.hip -- how many strided items does ONE XCD's L2 (4 MiB, 16 channels) hold?  (gfx950, standalone)
//
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/l2_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
//
// Every XCD sweeps the SAME set of n items (item = `item_bytes` contiguous bytes, consecutive items `stride_bytes` apart)
// `passes` times; which wave reads which item rotates from pass to pass, so the 32-KiB vector L1s do not help and every
// repeat is served by the XCD's L2 or by the fabric.  Output per configuration: time per pass and GB/s of item bytes; the
// knee of GB/s over n is the L2's effective capacity for that (item, stride).  DESIGN.md 5.6(e) uses it for the question
// "does a 256-byte slice of a 1536-byte texel reach all 16 channels of the L2?".  Run under
// rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum for hit rates (one dispatch per configuration, in print order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(const char *__restrict__ base, int n_items, int lanes_per_item, long stride_bytes,
                                             int passes, float *__restrict__ sink)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int groups_per_wave = 64 / lanes_per_item;
    const int lane = threadIdx.x & 63, sub = lane / lanes_per_item, l = lane % lanes_per_item;
    const int waves = (gridDim.x >> 3) * 4;
    const int wv = j * 4 + (threadIdx.x >> 6);
    const int slots = waves * groups_per_wave;                 // items read per step by this XCD
    f32x4 acc = (f32x4)0.0f;
    for (int p = 0; p < passes; ++p) {
        const int rot = (int)(((long)p * 7919) % slots);
        for (int s0 = 0; s0 < n_items; s0 += slots) {
            int slot = (wv * groups_per_wave + sub + rot) % slots;
            int i = s0 + slot;
            if (i < n_items) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long)i * stride_bytes + l * 16);
                acc += v;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = acc.x + xcd;   // never true: keeps the loads
}

int main()
{
    const long bytes = 1L << 30;
    char *buf; float *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipMalloc(&sink, 8 * 2048 * 256 * sizeof(float)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    struct Cfg { int item, stride; };
    // (items wider than 1024 bytes need more than one wave per item: not built)
    const Cfg cfgs[] = {{256, 256}, {256, 1536}, {512, 1536}, {128, 1536}, {1024, 1536}, {512, 4096}, {256, 4096}, {512, 512}, {512, 2048}, {512, 1024}, {512, 3072}};
    const int ns[] = {1024, 2048, 4096, 6144, 8192, 10240, 12288, 16384, 24576, 32768};
    const int passes = 40;
    printf("%-22s %8s %12s %12s %10s\n", "item B @ stride B", "items", "set KiB", "us / pass", "GB/s/XCD");
    for (const Cfg &c : cfgs)
        for (int n : ns) {
            if ((long)n * c.stride > bytes || c.item > 1024 || 1024 % c.item != 0) continue;
            const int grid = 8 * 32 * 4;                       // 4 workgroups per CU on every XCD
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, buf, n, c.item / 16, (long)c.stride, 3, sink);   // warm the L2s
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, buf, n, c.item / 16, (long)c.stride, passes, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double us = ms * 1e3 / passes;
            printf("%6d @ %-13d %8d %12.0f %12.2f %10.1f\n", c.item, c.stride, n, n * (double)c.item / 1024, us, n * (double)c.item / us / 1e3);
            fflush(stdout);
        }
    return 0;
}
