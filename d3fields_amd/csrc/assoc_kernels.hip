// assoc_kernels.hip -- the integer steps of the reference's multi-view instance association
// and of select_features_rand_v2 (fusion.py:1539-1606).  All results are integers and must be BIT-EXACT.
//
//   pcd_to_index_kernel   pcd_to_voxel + voxel_to_index of _init_low_level_memory (fusion.py:118-180): one lane per
//                         point, fp64 floor-divide like numpy, int32 linearisation with numpy's wrap-around.
//   voxset_*              Fusion.vox_idx_iou (fusion.py:794-799): |set(a) & set(b)| and |set(a) | set(b)| of two int32
//                         index arrays (duplicates allowed -- the reference passes un-deduplicated pcd_to_index output)
//                         through ONE open-addressing hash set in caller scratch: every key is inserted once with a
//                         membership bit per array (64-bit CAS on an empty slot, atomicOr on a found key), then the
//                         occupied slots are counted.  Exact for any key values, O(n), no sort.
//   erode_kernel          cv2.erode of a binary uint8 image with a kh x kw all-ones kernel, anchor at the kernel
//                         centre (kw/2, kh/2), one iteration, cv2's default border (out-of-image pixels never
//                         constrain the minimum) -- fusion.py:1293 (2x2) and :1561 (15x15).
//   (fps_np on INTEGER 2-D pixel coordinates, fusion.py:1566, lives with the 3-D variant in grid_kernels.hip: numpy
//                         computes float64 norms of exact integer differences, so comparing the exact int64 squared
//                         distances gives the same argmax sequence -- sqrt is monotone and correctly rounded, equal
//                         norms <=> equal squares below 2^53 --, first maximum wins.)
#include "d3f_internal.h"

namespace d3f {

// numpy's float64 -> int32 cast on x86-64 (cvttsd2si): out-of-range and NaN become INT32_MIN
__device__ __forceinline__ int32_t numpy_f64_to_i32(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int32_t)x;      // truncation toward zero
}

__global__ __launch_bounds__(kBlock) void pcd_to_index_kernel(const double *__restrict__ pts, int64_t n, double lx, double ly,
                                                             double lz, double vs, uint32_t n1, uint32_t n2,
                                                             int32_t *__restrict__ out_index, int32_t *__restrict__ out_voxel)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // voxels = np.floor((pcds - lower_bound) / voxel_size).astype(np.int32)            (fusion.py:124)
    const int32_t v0 = numpy_f64_to_i32(floor((pts[i * 3 + 0] - lx) / vs));
    const int32_t v1 = numpy_f64_to_i32(floor((pts[i * 3 + 1] - ly) / vs));
    const int32_t v2 = numpy_f64_to_i32(floor((pts[i * 3 + 2] - lz) / vs));
    // indexes = v0 * voxel_num[1] * voxel_num[2] + v1 * voxel_num[2] + v2   in int32, wrapping like numpy (fusion.py:140-144)
    const uint32_t idx = ((uint32_t)v0 * n1) * n2 + (uint32_t)v1 * n2 + (uint32_t)v2;
    out_index[i] = (int32_t)idx;
    if (out_voxel) { out_voxel[i * 3 + 0] = v0; out_voxel[i * 3 + 1] = v1; out_voxel[i * 3 + 2] = v2; }
}

hipError_t launch_pcd_to_index(const double *pts, int64_t n, const double *lower, double voxel_size, const int32_t *voxel_num,
                               int32_t *out_index, int32_t *out_voxel, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pcd_to_index_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, pts, n, lower[0],
                       lower[1], lower[2], voxel_size, (uint32_t)voxel_num[1], (uint32_t)voxel_num[2], out_index, out_voxel);
    return hipGetLastError();
}

// ---- hash set of int32 keys with two membership bits ----------------------------------------------------------------
constexpr unsigned long long kEmptySlot = ~0ULL;       // a live slot is (uint32 key) << 2 | bits, i.e. < 2^34

int64_t voxset_capacity(int64_t n1, int64_t n2)
{
    int64_t cap = 1024;
    while (cap < 2 * (n1 + n2)) cap <<= 1;              // load factor <= 0.5
    return cap;
}

__global__ __launch_bounds__(kBlock) void voxset_clear_kernel(unsigned long long *__restrict__ table, int64_t cap,
                                                             unsigned long long *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < cap) table[i] = kEmptySlot;
    if (i < 2) counts[i] = 0ULL;
}

__device__ __forceinline__ uint32_t hash_u32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;     // lowbias32 finaliser
    return x;
}

__global__ __launch_bounds__(kBlock) void voxset_insert_kernel(const int32_t *__restrict__ a, int64_t na,
                                                              const int32_t *__restrict__ b, int64_t nb,
                                                              unsigned long long *__restrict__ table, int64_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= na + nb) return;
    const uint32_t key = (uint32_t)(i < na ? a[i] : b[i - na]);
    const unsigned long long bit = i < na ? 1ULL : 2ULL;
    const unsigned long long mine = ((unsigned long long)key << 2) | bit;
    const uint64_t mask = (uint64_t)cap - 1;
    uint64_t h = hash_u32(key) & mask;
    for (int64_t probe = 0; probe < cap; ++probe) {
        unsigned long long cur = table[h];
        if (cur == kEmptySlot) {
            cur = atomicCAS(&table[h], kEmptySlot, mine);
            if (cur == kEmptySlot) return;              // this lane created the entry
        }
        if ((uint32_t)(cur >> 2) == key) {              // (cur is a live slot here: < 2^34)
            atomicOr(&table[h], bit);
            return;
        }
        h = (h + 1) & mask;
    }
}

__global__ __launch_bounds__(kBlock) void voxset_count_kernel(const unsigned long long *__restrict__ table, int64_t cap,
                                                             unsigned long long *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long cur = i < cap ? table[i] : kEmptySlot;
    const bool live = cur != kEmptySlot;
    const unsigned long long both = __ballot(live && (cur & 3ULL) == 3ULL), any = __ballot(live);
    if ((threadIdx.x & 63) == 0) {
        if (both) atomicAdd(&counts[0], (unsigned long long)__popcll(both));      // |A & B|
        if (any) atomicAdd(&counts[1], (unsigned long long)__popcll(any));        // |A | B|
    }
}

hipError_t launch_voxset_iou(const int32_t *a, int64_t na, const int32_t *b, int64_t nb, int64_t *counts, void *workspace,
                             hipStream_t s)
{
    const int64_t cap = voxset_capacity(na, nb);
    unsigned long long *table = static_cast<unsigned long long *>(workspace);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(counts);
    const unsigned gcap = (unsigned)((cap + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(voxset_clear_kernel, dim3(gcap), dim3(kBlock), 0, s, table, cap, cnt);
    if (na + nb > 0)
        hipLaunchKernelGGL(voxset_insert_kernel, dim3((unsigned)((na + nb + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, na, b,
                           nb, table, cap);
    hipLaunchKernelGGL(voxset_count_kernel, dim3(gcap), dim3(kBlock), 0, s, table, cap, cnt);
    return hipGetLastError();
}

// ---- cv2.erode, all-ones kh x kw structuring element ----------------------------------------------------------------
// dst(y,x) = min over i < kh, j < kw of src(y + i - kh/2, x + j - kw/2), out-of-image samples skipped (cv2's default
// border value for erosion is +max).  Binary images: dst = 255 iff no in-image sample of the window is 0... for general
// uint8 input the minimum is returned, as cv2 does.  Separable: rows then columns through LDS-free two passes would
// need scratch; the windows here are <= 15x15 on <= 1 MPixel images, so one pass with early exit is enough.
__global__ __launch_bounds__(kBlock) void erode_kernel(const uint8_t *__restrict__ src, int H, int W, int kh, int kw,
                                                      uint8_t *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const int y0 = max(y - kh / 2, 0), y1 = min(y - kh / 2 + kh - 1, H - 1);
    const int x0 = max(x - kw / 2, 0), x1 = min(x - kw / 2 + kw - 1, W - 1);
    unsigned m = 255u;
    for (int yy = y0; yy <= y1 && m != 0u; ++yy)
        for (int xx = x0; xx <= x1; ++xx) m = min(m, (unsigned)src[(int64_t)yy * W + xx]);
    dst[i] = (uint8_t)m;
}

hipError_t launch_erode(const uint8_t *src, int H, int W, int kh, int kw, uint8_t *dst, hipStream_t s)
{
    const int64_t n = (int64_t)H * W;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(erode_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, src, H, W, kh, kw, dst);
    return hipGetLastError();
}

// ---- swap_instance_mask (fusion.py:1052-1063): the consensus label image of one view ------------------------------------------
// The reference paints the view's detections one after the other in instance order (new_mask[mask_gs[i][idx]] = inst_idx), so a
// pixel ends up with the LARGEST instance index among the detections covering it (0 where none does; the value wraps into uint8
// as numpy's assignment does).  One lane per four pixels; the n detections are [n, n_pix] uint8 (non-zero = member).
__global__ __launch_bounds__(kBlock) void compose_labels_kernel(const uint8_t *__restrict__ dets, int n_dets, int64_t n_pix,
                                                               const int32_t *__restrict__ label_of_det, uint8_t *__restrict__ out)
{
    const int64_t p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (p0 >= n_pix) return;
    const int np = (int)min((int64_t)4, n_pix - p0);
    int best[4] = {0, 0, 0, 0};
    for (int m = 0; m < n_dets; ++m) {
        const int lab = label_of_det[m];
        if (lab < 0) continue;                                   // this detection belongs to no instance
        const uint8_t *row = dets + (int64_t)m * n_pix + p0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < np && row[k] != 0) best[k] = max(best[k], lab);
    }
    for (int k = 0; k < np; ++k) out[p0 + k] = (uint8_t)best[k];
}

hipError_t launch_compose_labels(const uint8_t *dets, int n_dets, int64_t n_pix, const int32_t *label_of_det, uint8_t *out, hipStream_t s)
{
    if (n_pix == 0) return hipSuccess;
    const int64_t lanes = (n_pix + 3) / 4;
    hipLaunchKernelGGL(compose_labels_kernel, dim3((unsigned)((lanes + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, dets, n_dets, n_pix,
                       label_of_det, out);
    return hipGetLastError();
}

// ---- voxel-grid mean: open3d's PointCloud::VoxelDownSample (utils/draw_utils.py:318-323, 396-400) ------------------------
// open3d buckets the points into voxels of side `vs` anchored at min_bound - vs/2 (voxel index = floor((p - anchor) / vs))
// and returns the mean point (and colour) of every occupied voxel, in the order of its hash map.  Here: one open-addressing
// hash table keyed by the packed voxel index (21 bits per axis); every point adds its offset inside its voxel to exact
// 64-bit fixed-point sums (2^-40 of a voxel side: order-independent, so the result is deterministic although the adds are
// atomic), the occupied slots are compacted, ranked by key (counting rank, V^2 / 2 comparisons through LDS) and written in
// ASCENDING key order: the same set of points as open3d's to ~1e-14 m, in a defined order.
struct VoxSlot {
    unsigned long long key;         // packed (ix, iy, iz), ~0 = empty
    unsigned long long cnt;
    long long sum[6];               // fixed-point offsets inside the voxel (x, y, z) and colours (r, g, b)
};
constexpr unsigned long long kVoxEmpty = ~0ULL;
constexpr double kVoxFix = 1099511627776.0;     // 2^40

int64_t voxmean_capacity(int64_t n)
{
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

int64_t voxmean_workspace_bytes(int64_t n)
{
    const int64_t cap = voxmean_capacity(n);
    // min bound (3 doubles + pad), table, per-workgroup slot counts, the compacted slot list (cap worst case is n entries)
    return 64 + cap * (int64_t)sizeof(VoxSlot) + ((cap + kBlock - 1) / kBlock + 1) * 8 + n * 8;
}

__global__ __launch_bounds__(1024) void voxmean_minbound_kernel(const double *__restrict__ pts, int64_t n, double *__restrict__ out)
{
    __shared__ double red[3][16];
    double m[3] = {INFINITY, INFINITY, INFINITY};
    for (int64_t i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int k = 0; k < 3; ++k) m[k] = fmin(m[k], pts[i * 3 + k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) m[k] = fmin(m[k], __shfl_xor(m[k], off, 64));
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red[threadIdx.x][0];
        for (int w = 1; w < 16; ++w) v = fmin(v, red[threadIdx.x][w]);
        out[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(kBlock) void voxmean_clear_kernel(VoxSlot *__restrict__ table, int64_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap) return;
    VoxSlot z;
    z.key = kVoxEmpty; z.cnt = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) z.sum[k] = 0;
    table[i] = z;
}

__global__ __launch_bounds__(kBlock) void voxmean_insert_kernel(const double *__restrict__ pts, const double *__restrict__ col, int64_t n,
                                                               double vs, const double *__restrict__ minb, VoxSlot *__restrict__ table,
                                                               int64_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0;
    long long fix[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double anchor = minb[k] - vs * 0.5;                      // open3d: voxel_min_bound = min_bound - voxel_size * 0.5
        const double ref = (pts[i * 3 + k] - anchor) / vs;
        double id = floor(ref);
        id = id < 0.0 ? 0.0 : (id > 2097151.0 ? 2097151.0 : id);      // 21 bits per axis
        key = (key << 21) | (unsigned long long)id;
        fix[k] = (long long)(((pts[i * 3 + k] - (anchor + id * vs)) / vs) * kVoxFix);
        if (col) fix[3 + k] = (long long)(col[i * 3 + k] * kVoxFix);
    }
    const uint64_t mask = (uint64_t)cap - 1;
    uint64_t h = (key * 0x9E3779B97F4A7C15ULL) >> 20 & mask;
    for (int64_t probe = 0; probe < cap; ++probe) {
        unsigned long long cur = table[h].key;
        if (cur == kVoxEmpty) cur = atomicCAS(&table[h].key, kVoxEmpty, key);
        if (cur == kVoxEmpty || cur == key) {
            atomicAdd(&table[h].cnt, 1ULL);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < 3 || col) atomicAdd(reinterpret_cast<unsigned long long *>(&table[h].sum[k]), (unsigned long long)fix[k]);
            return;
        }
        h = (h + 1) & mask;
    }
}

__global__ __launch_bounds__(kBlock) void voxmean_count_kernel(const VoxSlot *__restrict__ table, int64_t cap, int64_t *__restrict__ counts)
{
    __shared__ int wave_cnt[kBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < cap && table[i].key != kVoxEmpty;
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(kBlock) void voxmean_list_kernel(const VoxSlot *__restrict__ table, int64_t cap, const int64_t *__restrict__ counts,
                                                             int64_t *__restrict__ list, int64_t *__restrict__ total)
{
    __shared__ long long part[kBlock];
    __shared__ int wave_cnt[kBlock / 64];
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
    const bool live = i < cap && table[i].key != kVoxEmpty;
    const unsigned long long b = __ballot(live);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    if (live) list[prefix + before + __popcll(b & ((1ull << lane) - 1ull))] = i;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = prefix + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// rank of every occupied voxel among the keys (unique), then its mean at that position
__global__ __launch_bounds__(kBlock) void voxmean_write_kernel(const VoxSlot *__restrict__ table, const int64_t *__restrict__ list,
                                                              const int64_t *__restrict__ total, double vs, const double *__restrict__ minb,
                                                              double *__restrict__ out_pts, double *__restrict__ out_col)
{
    __shared__ unsigned long long keys[kBlock];
    const int64_t V = *total;
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if ((int64_t)blockIdx.x * kBlock >= V) return;                     // whole workgroups leave together
    const bool live = j < V;
    const VoxSlot me = table[list[live ? j : 0]];
    long long rank = 0;
    for (int64_t t0 = 0; t0 < V; t0 += kBlock) {
        __syncthreads();
        keys[threadIdx.x] = t0 + threadIdx.x < V ? table[list[t0 + threadIdx.x]].key : kVoxEmpty;
        __syncthreads();
        const int cnt = (int)min((int64_t)kBlock, V - t0);
        for (int t = 0; t < cnt; ++t) rank += keys[t] < me.key ? 1 : 0;
    }
    if (!live) return;
    const double inv = 1.0 / (double)me.cnt;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double id = (double)((me.key >> (21 * (2 - k))) & 0x1FFFFFULL);
        const double anchor = minb[k] - vs * 0.5;
        out_pts[rank * 3 + k] = (anchor + id * vs) + ((double)me.sum[k] * inv / kVoxFix) * vs;
        if (out_col) out_col[rank * 3 + k] = (double)me.sum[3 + k] * inv / kVoxFix;
    }
}

hipError_t launch_voxel_mean(const double *pts, const double *col, int64_t n, double vs, double *out_pts, double *out_col, int64_t *count,
                             void *workspace, hipStream_t s)
{
    const int64_t cap = voxmean_capacity(n);
    char *ws = static_cast<char *>(workspace);
    double *minb = reinterpret_cast<double *>(ws);
    VoxSlot *table = reinterpret_cast<VoxSlot *>(ws + 64);
    const unsigned gcap = (unsigned)((cap + kBlock - 1) / kBlock);
    int64_t *counts = reinterpret_cast<int64_t *>(ws + 64 + cap * (int64_t)sizeof(VoxSlot));
    int64_t *list = counts + gcap + 1;
    hipLaunchKernelGGL(voxmean_minbound_kernel, dim3(1), dim3(1024), 0, s, pts, n, minb);
    hipLaunchKernelGGL(voxmean_clear_kernel, dim3(gcap), dim3(kBlock), 0, s, table, cap);
    hipLaunchKernelGGL(voxmean_insert_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, pts, col, n, vs, minb, table, cap);
    hipLaunchKernelGGL(voxmean_count_kernel, dim3(gcap), dim3(kBlock), 0, s, table, cap, counts);
    hipLaunchKernelGGL(voxmean_list_kernel, dim3(gcap), dim3(kBlock), 0, s, table, cap, counts, list, count);
    hipLaunchKernelGGL(voxmean_write_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, table, list, count, vs, minb, out_pts, out_col);
    return hipGetLastError();
}

// ---- instance-mask gate and row-major nonzero (select_features_rand_v2, fusion.py:1554-1565) ------------------------------
// gate:    out(y,x) = 255 if mask(y,x) != 0 && depth(y,x) > lo && depth(y,x) < hi else 0 -- the reference's
//          `mask.astype(bool) & (depth > 0.0) & (depth < 1.5)` scaled to the uint8 image cv2.erode takes (fusion.py:1557-1561);
//          the mask channel is read in place from the channels-last one-hot tensor (element strides).
// nonzero: np.array(img.nonzero()).T -- the (row, col) pairs of the nonzero pixels in ascending (row-major) order, by the
//          two-pass order-preserving compaction of pcd_kernels.hip (per-workgroup counts, then prefix + in-workgroup rank).
__global__ __launch_bounds__(kBlock) void mask_gate_kernel(const float *__restrict__ mask, int64_t sy, int64_t sx,
                                                          const float *__restrict__ depth, int H, int W, float lo, float hi,
                                                          uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const float d = depth[i];
    out[i] = (mask[y * sy + x * sx] != 0.0f && d > lo && d < hi) ? 255 : 0;
}

hipError_t launch_mask_gate(const float *mask, int64_t sy, int64_t sx, const float *depth, int H, int W, float lo, float hi,
                            uint8_t *out, hipStream_t s)
{
    const int64_t n = (int64_t)H * W;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(mask_gate_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, mask, sy, sx, depth, H, W, lo, hi, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void nonzero_count_kernel(const uint8_t *__restrict__ img, int64_t npix, int64_t *__restrict__ counts)
{
    __shared__ int wave_cnt[kBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool keep = i < npix && img[i] != 0;
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(kBlock) void nonzero_write_kernel(const uint8_t *__restrict__ img, int64_t npix, int W,
                                                              const int64_t *__restrict__ counts, int64_t capacity,
                                                              int32_t *__restrict__ out_rc, int64_t *__restrict__ total)
{
    __shared__ long long part[kBlock];
    __shared__ int wave_cnt[kBlock / 64];
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
    const bool keep = i < npix && img[i] != 0;
    const unsigned long long b = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    if (keep) {
        const long long slot = prefix + before + __popcll(b & ((1ull << lane) - 1ull));
        if (slot < capacity) { out_rc[slot * 2 + 0] = (int32_t)(i / W); out_rc[slot * 2 + 1] = (int32_t)(i % W); }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        *total = prefix + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

hipError_t launch_nonzero_pixels(const uint8_t *img, int H, int W, int64_t capacity, int32_t *out_rc, int64_t *count,
                                 int64_t *block_counts, hipStream_t s)
{
    const int64_t npix = (int64_t)H * W;
    const unsigned nb = (unsigned)((npix + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(nonzero_count_kernel, dim3(nb), dim3(kBlock), 0, s, img, npix, block_counts);
    hipLaunchKernelGGL(nonzero_write_kernel, dim3(nb), dim3(kBlock), 0, s, img, npix, W, block_counts, capacity, out_rc, count);
    return hipGetLastError();
}

}  // namespace d3f
