// l2_bw.hip -- peak rate at which one XCD's L2 feeds its CUs with 512-byte gathers that all hit (gfx950, standalone)
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/l2_bw.hip -o /tmp/l2_bw && /tmp/l2_bw
// Every lane group of 32 lanes reads whole 512-byte items (16 B per lane) picked pseudo-randomly from a set that fits the
// L2 (and is far larger than the 32-KiB L1s), U independent loads in flight per lane, W workgroups of 256 lanes per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void gather(const char *__restrict__ base, unsigned n_items, long stride, int iters, float *sink)
{
    const unsigned lane32 = threadIdx.x & 31u, grp = (blockIdx.x * 256u + threadIdx.x) >> 5;
    unsigned h = grp * 2654435761u + 12345u;
    f32x4 acc = (f32x4)0.0f;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned item = (h >> 8) % n_items;
            v[u] = *reinterpret_cast<const f32x4 *>(base + (long)item * stride + lane32 * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = acc.y;
}

int main()
{
    char *buf; float *sink;
    CHECK(hipMalloc(&buf, 1L << 30)); CHECK(hipMemset(buf, 0, 1L << 30));
    CHECK(hipMalloc(&sink, 4L << 20));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    printf("%-16s %8s %6s %4s %12s %14s\n", "item @ stride", "set KiB", "WG/CU", "U", "TB/s chip", "TB/s per XCD");
    const long strides[] = {512, 1536, 2048};
    const unsigned sets[] = {2048, 4096};          // items: 1 MiB and 2 MiB of 512-byte items
    for (long stride : strides) for (unsigned n : sets) for (int wgcu : {4, 8}) for (int U : {4, 8, 16}) {
        const int grid = 256 * wgcu, iters = 400;
        auto launch = [&](int its) {
            if (U == 4) hipLaunchKernelGGL(gather<4>, dim3(grid), dim3(256), 0, 0, buf, n, stride, its, sink);
            else if (U == 8) hipLaunchKernelGGL(gather<8>, dim3(grid), dim3(256), 0, 0, buf, n, stride, its, sink);
            else hipLaunchKernelGGL(gather<16>, dim3(grid), dim3(256), 0, 0, buf, n, stride, its, sink);
        };
        launch(20);
        CHECK(hipEventRecord(a)); launch(iters); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)grid * 256 * 16 * U * iters;
        printf("512 @ %-10ld %8u %6d %4d %12.2f %14.2f\n", stride, n / 2, wgcu, U, bytes / ms / 1e9, bytes / ms / 1e9 / 8);
        fflush(stdout);
    }
    return 0;
}
