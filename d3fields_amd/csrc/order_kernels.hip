// order_kernels.hip -- locality ordering of query points (performance only; results never
// depend on it because points are independent, fusion.py:305-394).
//
// Why: with maps larger than the caches (C2 dense: 1.89 GB vs 8 x 4 MiB L2 + 256 MiB Infinity
// Cache) the fused kernel is bound by texel re-fetches; points that are close in 3-D project
// close together in EVERY view, so walking the points along a Morton curve keeps the in-flight
// texel footprint compact (measured 3.2 -> 2.8 ms on the 985 600-point grid, 3.4 -> 2.6 ms on a
// shuffled cloud).  Keys are 21-bit Morton codes of the 16-mm cell of each point (a 64..128-point
// tile spans a few such cells; the stable sort keeps the caller's order inside a cell); the pairs
// (key, index) are sorted with rocPRIM's device radix sort in caller-provided workspace and the
// fused kernel then reads its points through the index array.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

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

__global__ __launch_bounds__(kBlock) void morton_keys_kernel(const float *__restrict__ pts, int64_t n,
                                                            float inv_cell, uint32_t axis_mask, uint32_t *__restrict__ keys,
                                                            uint32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    // non-finite / huge coordinates just land in some cell: only locality is at stake
    const int qx = (int)fminf(fmaxf(floorf(x * inv_cell), -1e9f), 1e9f);
    const int qy = (int)fminf(fmaxf(floorf(y * inv_cell), -1e9f), 1e9f);
    const int qz = (int)fminf(fmaxf(floorf(z * inv_cell), -1e9f), 1e9f);
    keys[i] = spread3((uint32_t)qx & axis_mask) | (spread3((uint32_t)qy & axis_mask) << 1) | (spread3((uint32_t)qz & axis_mask) << 2);
    idx[i] = (uint32_t)i;
}

// rocPRIM's default switches to a ~20-launch merge sort below 1M items (measured 160 us at n = 985 600);
// Onesweep (histogram + 3 digit passes for 21-bit keys) is the right algorithm from a few 10k items up.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr float kCell = 0.016f;          // 16-mm cells x 128 per axis = 2.05 m before keys wrap (harmless)
constexpr size_t kSortScratch = 8u << 20;   // rocPRIM histogram/scan scratch (it needs far less)

int64_t order_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return (int64_t)(4 * align_up((size_t)n * 4, 256) + kSortScratch);
}

// Fills *order_out with a pointer (inside the workspace) to n uint32 indices in Morton order.
hipError_t build_point_order(const float *pts, int64_t n, void *workspace, int64_t workspace_bytes,
                             const uint32_t **order_out, hipStream_t stream, int fine)
{
    // fine = 0..2: cell = 16 mm >> fine, 7 + fine bits per axis (always ~2 m before the keys wrap)
    const float cell = kCell / (float)(1 << fine);
    const unsigned bits_axis = 7u + (unsigned)fine, key_bits = 3u * bits_axis;
    *order_out = nullptr;
    if (n <= 0 || n > 0x7fffffffLL || workspace_bytes < order_workspace_bytes(n)) return hipErrorInvalidValue;
    const size_t seg = align_up((size_t)n * 4, 256);
    unsigned char *base = static_cast<unsigned char *>(workspace);
    uint32_t *k0 = reinterpret_cast<uint32_t *>(base), *k1 = reinterpret_cast<uint32_t *>(base + seg);
    uint32_t *v0 = reinterpret_cast<uint32_t *>(base + 2 * seg), *v1 = reinterpret_cast<uint32_t *>(base + 3 * seg);
    void *scratch = base + 4 * seg;
    hipLaunchKernelGGL(morton_keys_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, pts, n,
                       1.0f / cell, (1u << bits_axis) - 1u, k0, v0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    rocprim::double_buffer<uint32_t> keys(k0, k1), vals(v0, v1);
    size_t need = 0;
    e = rocprim::radix_sort_pairs<SortConfig>(nullptr, need, keys, vals, (size_t)n, 0u, key_bits, stream);
    if (e != hipSuccess) return e;
    if (need > kSortScratch) return hipErrorOutOfMemory;
    e = rocprim::radix_sort_pairs<SortConfig>(scratch, need, keys, vals, (size_t)n, 0u, key_bits, stream);
    if (e != hipSuccess) return e;
    // the finished order always ends in the FIRST index buffer, so that a later call can find it again
    // (D3F_FLAG_REUSE_POINT_ORDER) without knowing how many digit passes ran
    if (vals.current() != v0) {
        e = hipMemcpyAsync(v0, vals.current(), (size_t)n * 4, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    *order_out = v0;
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
constexpr int kProbeSamples = 4096;

__global__ __launch_bounds__(kBlock) void point_locality_kernel(const float *__restrict__ pts, int64_t n, int samples,
                                                               float *__restrict__ out)
{
    __shared__ float red[3][kBlock / 64];
    float near_d = 0.0f, far_d = 0.0f, cnt = 0.0f;
    for (int k = threadIdx.x; k < samples && n >= 2; k += kBlock) {
        const int64_t i = (int64_t)((double)k * (double)(n - 1) / (double)samples);      // i + 1 <= n - 1
        const int64_t j = (i + n / 2) % n;
        const float ax = pts[i * 3], ay = pts[i * 3 + 1], az = pts[i * 3 + 2];
        const float dn = fabsf(pts[i * 3 + 3] - ax) + fabsf(pts[i * 3 + 4] - ay) + fabsf(pts[i * 3 + 5] - az);
        const float df = fabsf(pts[j * 3] - ax) + fabsf(pts[j * 3 + 1] - ay) + fabsf(pts[j * 3 + 2] - az);
        if (dn < INFINITY && df < INFINITY) { near_d += dn; far_d += df; cnt += 1.0f; }  // false for NaN too
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
        for (int w = 0; w < kBlock / 64; ++w) { t[0] += red[0][w]; t[1] += red[1][w]; t[2] += red[2][w]; }
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
    hipLaunchKernelGGL(point_locality_kernel, dim3(1), dim3(kBlock), 0, stream, pts, n, kProbeSamples, out);
    return hipGetLastError();
}

}  // namespace d3f
