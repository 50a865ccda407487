// div_rate.hip -- what does the IEEE fp32 division of the projection (fusion.py:54, 72-73) cost on gfx950, and what would its pieces cost?
// Register-only loops at 2 / 4 / 8 waves per SIMD on every CU; cycles from s_memtime inside the kernel (wave 0 of every workgroup), so the
// figures do not depend on the clock the box runs at.  Per step and lane: FOUR independent chains of the variant.
//   0  x / y as hipcc expands it without fast-math (v_div_scale x2, v_rcp, 4 fma, v_mul, v_div_fmas, v_div_fixup: 11 instructions)
//   1  the same division with the denominator's part (scale, rcp, two refinement fma) hoisted: v_mul + 4 v_fma per quotient
//   2  11 plain v_fma_f32 (the instruction count of variant 0 at full rate)
//   3  v_rcp_f32 alone          4  v_div_scale_f32 alone          5  v_div_fmas_f32 + v_div_fixup_f32
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math div_rate.hip -o div_rate && ./div_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VAR>
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, int nsteps, float *__restrict__ out, unsigned long long *cyc)
{
    const int l = threadIdx.x;
    float x[4], y[4];
    for (int q = 0; q < 4; ++q) { x[q] = in[l + 256 * q] + 1.5f + q; y[q] = in[l + 256 * (q + 4)] + 3.25f + q; }
    const float den = y[0], r1 = 1.0f / den;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (VAR == 0) {
                x[q] = x[q] / y[q];
                asm volatile("" : "+v"(x[q]));
            } else if (VAR == 1) {
                float n = x[q], qq, e;
                asm volatile("v_mul_f32 %0, %2, %3\n\tv_fma_f32 %1, -%4, %0, %2\n\tv_fma_f32 %0, %1, %3, %0\n\tv_fma_f32 %1, -%4, %0, %2\n\tv_fma_f32 %0, %1, %3, %0"
                             : "=&v"(qq), "=&v"(e) : "v"(n), "v"(r1), "v"(den));
                x[q] = qq;
            } else if (VAR == 2) {
                asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\t"
                             "v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1"
                             : "+v"(x[q]) : "v"(y[q]));
            } else if (VAR == 3) {
                asm volatile("v_rcp_f32 %0, %0" : "+v"(x[q]));
            } else if (VAR == 4) {
                asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x[q]) : "v"(y[q]) : "vcc");
            } else {
                asm volatile("v_div_fmas_f32 %0, %0, %1, %1\n\tv_div_fixup_f32 %0, %0, %1, %1" : "+v"(x[q]) : "v"(y[q]) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 256 + l] = (x[0] + x[1]) + (x[2] + x[3]);
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int VAR> static void go(int nblocks, const float *in, int nsteps, float *out, unsigned long long *cyc) { hipLaunchKernelGGL(k<VAR>, dim3(nblocks), dim3(256), 0, 0, in, nsteps, out, cyc); }
int main()
{
    float *d_in, *d_out; unsigned long long *d_cyc;
    const int nsteps = 20000;
    CK(hipMalloc(&d_in, 1 << 20)); CK(hipMemset(d_in, 0, 1 << 20));
    CK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4)); CK(hipMalloc(&d_cyc, 256 * 8 * 8));
    static unsigned long long h[256 * 8];
    const char *names[6] = {"x / y (IEEE, 11 instructions)", "hoisted denominator (5)", "11 v_fma_f32", "v_rcp_f32 (1)", "v_div_scale_f32 (1)", "v_div_fmas + v_div_fixup (2)"};
    const int ninst[6] = {11, 5, 11, 1, 1, 2};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 2; wps <= 8; wps *= 2)
        for (int var = 0; var < 6; ++var) {
            const int nblocks = 256 * wps;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                switch (var) { case 0: go<0>(nblocks, d_in, nsteps, d_out, d_cyc); break; case 1: go<1>(nblocks, d_in, nsteps, d_out, d_cyc); break;
                               case 2: go<2>(nblocks, d_in, nsteps, d_out, d_cyc); break; case 3: go<3>(nblocks, d_in, nsteps, d_out, d_cyc); break;
                               case 4: go<4>(nblocks, d_in, nsteps, d_out, d_cyc); break; default: go<5>(nblocks, d_in, nsteps, d_out, d_cyc); break; }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            CK(hipMemcpy(h, d_cyc, nblocks * 8, hipMemcpyDeviceToHost));
            double c = 0; for (int b = 0; b < nblocks; ++b) c += (double)h[b]; c /= nblocks;
            // a SIMD holds wps waves, each running nsteps * 4 chains: cycles of the SIMD per chain = c / (nsteps * 4 * wps)
            const double per_chain = c / ((double)nsteps * 4.0 * wps);
            printf("%d waves/SIMD  %-32s %.3f ms  %8.1f cycles per wave-chain per SIMD  = %.2f cycles per wave instruction\n", wps, names[var], best, per_chain, per_chain / ninst[var]);
        }
    return 0;
}
