// fma_rate.hip -- what does a wave64 v_pk_fma_f32 cost on gfx950 next to two v_fma_f32 of the same work?  (MI355X_MICROARCH.md lists v_fma_f32 at
// 2 cycles on the SIMD-32 and calls packed fp32 an anti-lever beside MFMAs.)  Register-only loops, 8 independent accumulator chains of
// depth 4 per step (the window kernel's corner arithmetic: four fma per accumulator vector), 1 / 2 / 4 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate && ./fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PACKED>
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, int nsteps, float *__restrict__ out)
{
    const int l = threadIdx.x;
    f32x4 c[4];
    for (int q = 0; q < 4; ++q) c[q] = reinterpret_cast<const f32x4 *>(in)[l + 256 * q];
    f32x4 w = reinterpret_cast<const f32x4 *>(in)[l & 3];
    f32x4 acc0 = (f32x4)0.0f, acc1 = (f32x4)0.0f;       // two accumulator vectors: 16 fma per step like one (point, view) of the window loop
    for (int s = 0; s < nsteps; ++s) {
        if (PACKED) {
            f32x2 a0 = {acc0.x, acc0.y}, a1 = {acc0.z, acc0.w}, a2 = {acc1.x, acc1.y}, a3 = {acc1.z, acc1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 lo = {c[q].x, c[q].y}, hi = {c[q].z, c[q].w};
                const f32x2 ww = {w[q], w[q]};
                asm volatile("v_pk_fma_f32 %0, %4, %6, %0\n\tv_pk_fma_f32 %1, %5, %6, %1\n\tv_pk_fma_f32 %2, %5, %6, %2\n\tv_pk_fma_f32 %3, %4, %6, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(lo), "v"(hi), "v"(ww));
            }
            acc0 = f32x4{a0.x, a0.y, a1.x, a1.y}; acc1 = f32x4{a2.x, a2.y, a3.x, a3.y};
        } else {
            float a[8] = {acc0.x, acc0.y, acc0.z, acc0.w, acc1.x, acc1.y, acc1.z, acc1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float wq = w[q];
                asm volatile("v_fma_f32 %0, %8, %12, %0\n\tv_fma_f32 %1, %9, %12, %1\n\tv_fma_f32 %2, %10, %12, %2\n\tv_fma_f32 %3, %11, %12, %3\n\t"
                             "v_fma_f32 %4, %10, %12, %4\n\tv_fma_f32 %5, %11, %12, %5\n\tv_fma_f32 %6, %8, %12, %6\n\tv_fma_f32 %7, %9, %12, %7"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(c[q].x), "v"(c[q].y), "v"(c[q].z), "v"(c[q].w), "v"(wq));
            }
            acc0 = f32x4{a[0], a[1], a[2], a[3]}; acc1 = f32x4{a[4], a[5], a[6], a[7]};
        }
    }
    reinterpret_cast<f32x4 *>(out)[(size_t)blockIdx.x * 512 + l] = acc0;
    reinterpret_cast<f32x4 *>(out)[(size_t)blockIdx.x * 512 + 256 + l] = acc1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main()
{
    float *d_in, *d_out;
    const int nsteps = 20000;
    CK(hipMalloc(&d_in, 1 << 20)); CK(hipMemset(d_in, 0, 1 << 20));
    CK(hipMalloc(&d_out, (size_t)256 * 8 * 512 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int packed = 0; packed < 2; ++packed) {
            const int nblocks = 256 * wps;      // 256-lane workgroups: one wave per SIMD each
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (packed) hipLaunchKernelGGL(k<1>, dim3(nblocks), dim3(256), 0, 0, d_in, nsteps, d_out);
                else hipLaunchKernelGGL(k<0>, dim3(nblocks), dim3(256), 0, 0, d_in, nsteps, d_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double flop = (double)nblocks * 256.0 * nsteps * 32.0 * 2.0;
            printf("%d wave(s) per SIMD, %s: %.3f ms, %.1f TFLOP/s, %.2f ns per step and wave slot (32 fma lanes-ops per lane)\n", wps,
                   packed ? "16 v_pk_fma_f32 per step" : "32 v_fma_f32 per step   ", best, flop / (best * 1e-3) / 1e12, best * 1e6 / nsteps / wps);
        }
    return 0;
}
