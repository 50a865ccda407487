// Does a ds_read_b128 with only some 16-lane groups of the wave active cost fewer LDS cycles?  (round 5: would skipping the corner
// reads of a lane group whose point shares the previous point's texel cell relieve the LDS pipe of the window kernel?)
//   hipcc --offload-arch=gfx950 -O3 lds_partial_exec.cpp -o lds_partial_exec && ./lds_partial_exec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACTIVE_GROUPS>      // 16-lane groups of every wave that read (1..4)
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    extern __shared__ __align__(16) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float *>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    f32x4 acc = (f32x4)0.0f;
    uint32_t off = (uint32_t)(threadIdx.x & 15) * 16u + (uint32_t)grp * 4096u;
    if (grp < ACTIVE_GROUPS) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + ((off + u * 512u) & 0xffffu));
                acc += v;
            }
            off += 256u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int A> float run(float *out, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<A>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<A><<<256 * 8, 256, 65536>>>(out, iters);
    hipEventRecord(a);
    k<A><<<256 * 8, 256, 65536>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 2000;
    const float t4 = run<4>(out, iters), t3 = run<3>(out, iters), t2 = run<2>(out, iters), t1 = run<1>(out, iters);
    // per CU: 8 workgroups x 4 waves x iters x 8 reads
    printf("ds_read_b128, lane groups active per wave: 4: %.3f ms  3: %.3f  2: %.3f  1: %.3f   (ratio to 4: %.2f %.2f %.2f)\n", t4, t3, t2, t1, t3 / t4, t2 / t4, t1 / t4);
    const double bytes = 256.0 * 8 * 4 * iters * 8 * 1024;
    printf("full-wave rate: %.1f TB/s over the chip (%.0f B/clk/CU at 2.4 GHz)\n", bytes / (t4 * 1e-3) / 1e12, bytes / (t4 * 1e-3) / 256 / 2.4e9);
    return 0;
}
