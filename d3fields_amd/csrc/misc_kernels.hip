// misc_kernels.hip -- instance-mask format helpers (reference fusion.py:90-116) for gfx950.
// HBM-bound byte/float streaming; one lane per row, NI is small (number of instances).
#include "d3f_internal.h"

namespace d3f {

// onehot2instance (fusion.py:109-116): argmax over the last dim -> uint8.
// Tie rule of torch.argmax / np.argmax: first maximum wins, a NaN is the maximum.
__global__ __launch_bounds__(kBlock) void onehot2instance_kernel(const float *__restrict__ onehot, int64_t n,
                                                                int NI, uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *r = onehot + i * NI;
    int best = 0;
    float bv = r[0];
    for (int c = 1; c < NI; ++c) {
        const float x = r[c];
        const bool take = !(bv != bv) && ((x > bv) || (x != x));
        bv = take ? x : bv;
        best = take ? c : best;
    }
    out[i] = (uint8_t)best;
}

// instance2onehot (fusion.py:90-107): out[i, c] = (instance[i] == c), bool bytes.
__global__ __launch_bounds__(kBlock) void instance2onehot_kernel(const uint8_t *__restrict__ inst, int64_t total,
                                                                int NI, uint8_t *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= total) return;
    const int64_t i = k / NI;
    const int c = (int)(k - i * NI);
    out[k] = (inst[i] == c) ? 1 : 0;
}

hipError_t launch_onehot2instance(const float *onehot, int64_t n, int NI, uint8_t *out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(onehot2instance_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       onehot, n, NI, out);
    return hipGetLastError();
}

hipError_t launch_instance2onehot(const uint8_t *inst, int64_t n, int NI, uint8_t *out, hipStream_t s)
{
    const int64_t total = n * NI;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(instance2onehot_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       inst, total, NI, out);
    return hipGetLastError();
}

}  // namespace d3f
