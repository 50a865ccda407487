// pcd_kernels.hip -- the point-cloud steps on the mask side of the path, fp64 like numpy.
//
//   backproject_*   depth2fgpcd (utils/my_utils.py:522-537) fused with the camera->world transform and the
//                   boundary crop of aggr_point_cloud_from_data (utils/draw_utils.py:325-413): one lane per
//                   pixel, ORDER-PRESERVING compaction (numpy's boolean-mask order = ascending pixel index)
//                   in two passes: per-workgroup survivor counts, then prefix (sum of the counts before the
//                   workgroup) + in-workgroup rank.
//   nearest_kernel  the two directed nearest-neighbour searches of Fusion.pcd_iou (fusion.py:724-741), which
//                   the reference does through an [N,M] distance matrix: one lane per query point, the other
//                   cloud staged through LDS in 256-point tiles, first minimum wins (np.argmin).
#include "d3f_internal.h"

namespace d3f {

struct BackprojectParams {
    const double *depth;      // [H*W]
    const uint8_t *mask;      // [H*W] or nullptr
    int64_t npix;
    int32_t W;
    double fx, fy, cx, cy;
    double T[12];             // first three rows of inv(pose), row-major
    double lo[3], hi[3];
    int32_t crop;
};

__device__ __forceinline__ bool backproject_pixel(const BackprojectParams &P, int64_t i, double &wx, double &wy, double &wz)
{
    const double d = P.depth[i];
    // aggr_point_cloud_from_data: masks is None -> (depth > 0) & (depth < 1.5); else masks & (depth > 0)
    const bool fg = P.mask ? (P.mask[i] != 0 && d > 0.0) : (d > 0.0 && d < 1.5);
    if (!fg) return false;
    const double px = (double)(i % P.W), py = (double)(i / P.W);
    const double x = (px - P.cx) * d / P.fx;            // depth2fgpcd: (pos_x - cx) * depth / fx
    const double y = (py - P.cy) * d / P.fy;
    wx = ((P.T[0] * x + P.T[1] * y) + P.T[2] * d) + P.T[3];
    wy = ((P.T[4] * x + P.T[5] * y) + P.T[6] * d) + P.T[7];
    wz = ((P.T[8] * x + P.T[9] * y) + P.T[10] * d) + P.T[11];
    if (P.crop)
        return wx > P.lo[0] && wx < P.hi[0] && wy > P.lo[1] && wy < P.hi[1] && wz > P.lo[2] && wz < P.hi[2];
    return true;
}

__global__ __launch_bounds__(kBlock) void backproject_count_kernel(const BackprojectParams P, int64_t *__restrict__ counts)
{
    __shared__ int wave_cnt[kBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double x, y, z;
    const bool keep = i < P.npix && backproject_pixel(P, i, x, y, z);
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(kBlock) void backproject_write_kernel(const BackprojectParams P, const int64_t *__restrict__ counts,
                                                                  int64_t capacity, double *__restrict__ out_pts,
                                                                  int32_t *__restrict__ out_pixel, int64_t *__restrict__ total)
{
    __shared__ long long part[kBlock];
    __shared__ int wave_cnt[kBlock / 64];
    // exclusive prefix of this workgroup = sum of the counts of all earlier workgroups
    long long acc = 0;
    for (int64_t b = threadIdx.x; b < (int64_t)blockIdx.x; b += kBlock) acc += counts[b];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    const long long prefix = part[0];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double x = 0, y = 0, z = 0;
    const bool keep = i < P.npix && backproject_pixel(P, i, x, y, z);
    const unsigned long long b = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    if (keep) {
        const long long slot = prefix + before + __popcll(b & ((1ull << lane) - 1ull));
        if (slot < capacity) {
            out_pts[slot * 3 + 0] = x; out_pts[slot * 3 + 1] = y; out_pts[slot * 3 + 2] = z;
            if (out_pixel) out_pixel[slot] = (int32_t)i;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        *total = prefix + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

hipError_t launch_backproject(const double *depth, const uint8_t *mask, int H, int W, const double *cam, const double *T,
                              const double *bounds, int64_t capacity, double *out_pts, int32_t *out_pixel, int64_t *count,
                              int64_t *block_counts, hipStream_t s)
{
    BackprojectParams P;
    P.depth = depth; P.mask = mask; P.npix = (int64_t)H * W; P.W = W;
    P.fx = cam[0]; P.fy = cam[1]; P.cx = cam[2]; P.cy = cam[3];
    for (int k = 0; k < 12; ++k) P.T[k] = T[k];
    P.crop = bounds ? 1 : 0;
    for (int k = 0; k < 3; ++k) { P.lo[k] = bounds ? bounds[2 * k] : 0.0; P.hi[k] = bounds ? bounds[2 * k + 1] : 0.0; }
    const unsigned nb = (unsigned)((P.npix + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(backproject_count_kernel, dim3(nb), dim3(kBlock), 0, s, P, block_counts);
    hipLaunchKernelGGL(backproject_write_kernel, dim3(nb), dim3(kBlock), 0, s, P, block_counts, capacity, out_pts, out_pixel, count);
    return hipGetLastError();
}

// ---- directed nearest neighbour (fusion.py:731-735): min / first argmin of sqrt((dx*dx + dy*dy) + dz*dz) ----
__global__ __launch_bounds__(kBlock) void nearest_kernel(const double *__restrict__ a, int64_t na, const double *__restrict__ b,
                                                        int64_t nb, double *__restrict__ min_dist, int64_t *__restrict__ argmin)
{
    __shared__ double tile[kBlock * 3];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < na;
    const double ax = live ? a[i * 3 + 0] : 0.0, ay = live ? a[i * 3 + 1] : 0.0, az = live ? a[i * 3 + 2] : 0.0;
    double best = INFINITY;
    int64_t arg = 0;
    for (int64_t j0 = 0; j0 < nb; j0 += kBlock) {
        const int64_t cnt = min((int64_t)kBlock, nb - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += kBlock) tile[t] = b[j0 * 3 + t];
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const double dx = ax - tile[t * 3 + 0], dy = ay - tile[t * 3 + 1], dz = az - tile[t * 3 + 2];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; arg = j0 + t; }      // sqrt is monotone: compare squares, root once
        }
    }
    if (live) { min_dist[i] = sqrt(best); argmin[i] = arg; }
}

hipError_t launch_nearest(const double *a, int64_t na, const double *b, int64_t nb, double *min_dist, int64_t *argmin, hipStream_t s)
{
    if (na == 0) return hipSuccess;
    hipLaunchKernelGGL(nearest_kernel, dim3((unsigned)((na + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, na, b, nb, min_dist, argmin);
    return hipGetLastError();
}

}  // namespace d3f
