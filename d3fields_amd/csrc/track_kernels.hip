// track_kernels.hip -- the optimiser iteration of Fusion.rigid_tracking as HIP kernels (gfx950).
//
// The reference (fusion.py:1608-1685) optimises, per tracked instance i, a translation t_i and an axis-angle w_i with
// 100 Adam steps; one step is  so3_exp_map -> rigid transform of the keypoints -> Fusion.eval -> loss -> autograd ->
// Adam, about ninety tiny torch launches.  Around the two field-query kernels (d3f_eval, d3f_eval_backward) the rest
// of the step is closed-form, so it is written out here as three kernels and the whole step becomes five launches
// that a HIP graph replays (d3fields_amd/rigid.py, RigidTracker):
//
//   rigid_transform_kernel   R_i = I + sin(th)/th K + (1-cos th)/th^2 K^2,  th = sqrt(max(|w_i|^2, 1e-4)),  K = hat(w_i)
//                            (pytorch3d 0.7.5 so3_exp_map);  p' = p R_i + t_i  (row-vector convention of Transform3d);
//                            also the Frobenius norms of t and w over ALL instances (the regulariser, fusion.py:1658)
//   track_loss_grad_kernel   loss = mean(|f - s| * valid) + dist_w * mean(max(dist * valid, 0))   (fusion.py:1654-1657)
//                            and its gradients w.r.t. f and dist (what autograd hands to the backward of eval)
//   rigid_update_kernel      d loss / d(t_i, w_i) from d loss / d p' by the chain rule through the transform and the
//                            exponential map, + reg_w * d(|t|_F + |w|_F), then torch.optim.Adam's update of the six
//                            parameters of the instance (one workgroup per instance)
//
// Sums run in a different order than torch's (block reductions vs GEMM / TensorIterator), so results agree with the
// autograd loop to rounding, not bit for bit; tests pin both against the keypoints the reference's own loop returned.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"

namespace d3f {

struct Rot { float m[9]; float th, a, b; bool clamped; };

// so3_exp_map of one axis-angle vector
__device__ __forceinline__ Rot exp_map(float wx, float wy, float wz, float eps)
{
    Rot r;
    const float nrm = (wx * wx + wy * wy) + wz * wz;
    r.clamped = !(nrm >= eps);                       // torch.clamp(nrm, eps): gradient passes where nrm >= eps
    const float th = sqrtf(fmaxf(nrm, eps));
    const float inv = 1.0f / th;
    r.th = th;
    r.a = inv * sinf(th);
    r.b = inv * inv * (1.0f - cosf(th));
    // K = hat(w); KK = K @ K
    const float K[9] = {0.0f, -wz, wy, wz, 0.0f, -wx, -wy, wx, 0.0f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float kk = (K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j]) + K[i * 3 + 2] * K[2 * 3 + j];
            r.m[i * 3 + j] = (r.a * K[i * 3 + j] + r.b * kk) + (i == j ? 1.0f : 0.0f);
        }
    return r;
}

__global__ __launch_bounds__(kBlock) void rigid_transform_kernel(const float *__restrict__ last, int I, int n,
                                                                const float *__restrict__ t, const float *__restrict__ w,
                                                                float eps, float *__restrict__ out_pts, float *__restrict__ norms)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx == 0) {      // |t|_F and |w|_F over all instances (torch.norm of the whole [I,3] tensors)
        float st = 0.0f, sw = 0.0f;
        for (int k = 0; k < I * 3; ++k) { st += t[k] * t[k]; sw += w[k] * w[k]; }
        norms[0] = sqrtf(st);
        norms[1] = sqrtf(sw);
    }
    if (idx >= I * n) return;
    const int i = idx / n;
    const Rot r = exp_map(w[i * 3], w[i * 3 + 1], w[i * 3 + 2], eps);
    const float px = last[idx * 3], py = last[idx * 3 + 1], pz = last[idx * 3 + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = px * r.m[0 * 3 + j];
        acc = fmaf(py, r.m[1 * 3 + j], acc);
        acc = fmaf(pz, r.m[2 * 3 + j], acc);
        out_pts[idx * 3 + j] = acc + t[i * 3 + j];
    }
}

// one 64-lane wave per keypoint
__global__ __launch_bounds__(kBlock) void track_loss_grad_kernel(const float *__restrict__ feats, const float *__restrict__ src,
                                                                const float *__restrict__ dist, const uint8_t *__restrict__ valid,
                                                                int N, int C, float dist_w, float *__restrict__ grad_feats,
                                                                float *__restrict__ grad_dist, float *__restrict__ loss)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= N) return;                                  // whole waves leave together
    const float *f = feats + (int64_t)p * C, *s = src + (int64_t)p * C;
    float ss = 0.0f;
    for (int c = lane; c < C; c += 64) {
        const float d = f[c] - s[c];
        ss = fmaf(d, d, ss);
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float nrm = sqrtf(ss);
    const float vf = valid[p] ? 1.0f : 0.0f;
    const float invN = 1.0f / (float)N;
    const float scale = (nrm > 0.0f) ? (vf * invN) / nrm : 0.0f;     // norm backward: 0 at a zero difference
    float *g = grad_feats + (int64_t)p * C;
    for (int c = lane; c < C; c += 64) g[c] = (f[c] - s[c]) * scale;
    if (lane == 0) {
        const float cd = dist[p] * vf;
        grad_dist[p] = (cd >= 0.0f) ? dist_w * invN * vf : 0.0f;    // clamp(min=0) passes the gradient where x >= 0
        atomicAdd(loss + 0, nrm * vf * invN);
        atomicAdd(loss + 1, dist_w * fmaxf(cd, 0.0f) * invN);
    }
}

// one workgroup per instance: reduce over its n keypoints, chain rule, Adam
__global__ __launch_bounds__(kBlock) void rigid_update_kernel(const float *__restrict__ last, int n, const float *__restrict__ grad_pts,
                                                             float *__restrict__ t, float *__restrict__ w, float *__restrict__ adam_m,
                                                             float *__restrict__ adam_v, float *__restrict__ step,
                                                             const float *__restrict__ norms, float eps_rot, float reg_w, float lr,
                                                             float beta1, float beta2, float eps_adam)
{
    __shared__ float red[12][kBlock / 64];
    const int i = blockIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    for (int p = threadIdx.x; p < n; p += kBlock) {
        const int64_t q = ((int64_t)i * n + p) * 3;
        const float gx = grad_pts[q], gy = grad_pts[q + 1], gz = grad_pts[q + 2];
        const float px = last[q], py = last[q + 1], pz = last[q + 2];
        acc[0] += gx; acc[1] += gy; acc[2] += gz;                       // d/dt
        acc[3] += px * gx; acc[4] += px * gy; acc[5] += px * gz;        // d/dR[k][j] = sum p_k g_j
        acc[6] += py * gx; acc[7] += py * gy; acc[8] += py * gz;
        acc[9] += pz * gx; acc[10] += pz * gy; acc[11] += pz * gz;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float G[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float v = 0.0f;
        for (int wv = 0; wv < kBlock / 64; ++wv) v += red[k][wv];
        G[k] = v;
    }
    const float wx = w[i * 3], wy = w[i * 3 + 1], wz = w[i * 3 + 2];
    const Rot r = exp_map(wx, wy, wz, eps_rot);
    const float K[9] = {0.0f, -wz, wy, wz, 0.0f, -wx, -wy, wx, 0.0f};
    const float *GR = G + 3;                                            // 3x3, row k, column j
    // R = a K + b K^2 + I:  dL/da = <G, K>, dL/db = <G, K^2>, dL/dK = a G + b (G K^T + K^T G)
    float KK[9], dK[9];
    float da = 0.0f, db = 0.0f;
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            KK[x * 3 + y] = (K[x * 3 + 0] * K[0 * 3 + y] + K[x * 3 + 1] * K[1 * 3 + y]) + K[x * 3 + 2] * K[2 * 3 + y];
            da += GR[x * 3 + y] * K[x * 3 + y];
        }
#pragma unroll
    for (int x = 0; x < 9; ++x) db += GR[x] * KK[x];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            float gkT = 0.0f, kTg = 0.0f;                               // (G K^T)[x][y] = sum_z G[x][z] K[y][z];  (K^T G)[x][y] = sum_z K[z][x] G[z][y]
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                gkT += GR[x * 3 + z] * K[y * 3 + z];
                kTg += K[z * 3 + x] * GR[z * 3 + y];
            }
            dK[x * 3 + y] = r.a * GR[x * 3 + y] + r.b * (gkT + kTg);
        }
    float gw[3] = {dK[2 * 3 + 1] - dK[1 * 3 + 2], dK[0 * 3 + 2] - dK[2 * 3 + 0], dK[1 * 3 + 0] - dK[0 * 3 + 1]};
    if (!r.clamped) {                                                   // through a(th), b(th), th = |w|
        const float th = r.th, s = sinf(th), c = cosf(th);
        const float da_dth = (th * c - s) / (th * th);
        const float db_dth = (th * s - 2.0f * (1.0f - c)) / (th * th * th);
        const float dth = da * da_dth + db * db_dth;
        gw[0] += dth * wx / th; gw[1] += dth * wy / th; gw[2] += dth * wz / th;
    }
    float g6[6] = {G[0], G[1], G[2], gw[0], gw[1], gw[2]};
    // regulariser reg_w * (|t|_F + |w|_F): gradient x / |x|_F, 0 at the origin (torch.norm backward)
    const float nt = norms[0], nw = norms[1];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (nt > 0.0f) g6[k] += reg_w * t[i * 3 + k] / nt;
        if (nw > 0.0f) g6[3 + k] += reg_w * w[i * 3 + k] / nw;
    }
    // torch.optim.Adam (amsgrad off, no weight decay): one step of the six parameters of this instance
    const float st = step[i] + 1.0f;
    step[i] = st;
    const float bc1 = 1.0f - powf(beta1, st), bc2 = 1.0f - powf(beta2, st);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float *par = k < 3 ? t + i * 3 + k : w + i * 3 + (k - 3);
        const float g = g6[k];
        float m = adam_m[i * 6 + k], v = adam_v[i * 6 + k];
        m = m + (g - m) * (1.0f - beta1);
        v = v * beta2 + (1.0f - beta2) * (g * g);
        adam_m[i * 6 + k] = m;
        adam_v[i * 6 + k] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps_adam;
        *par = *par - step_size * (m / denom);
    }
}

hipError_t launch_rigid_transform(const float *last, int I, int n, const float *t, const float *w, float eps, float *out_pts,
                                  float *norms, hipStream_t s)
{
    const int total = I * n;
    hipLaunchKernelGGL(rigid_transform_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock > 0 ? (total + kBlock - 1) / kBlock : 1)),
                       dim3(kBlock), 0, s, last, I, n, t, w, eps, out_pts, norms);
    return hipGetLastError();
}

hipError_t launch_track_loss_grad(const float *feats, const float *src, const float *dist, const uint8_t *valid, int N, int C,
                                  float dist_w, float *grad_feats, float *grad_dist, float *loss, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(loss, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return e;
    if (N == 0) return hipSuccess;
    const int per = kBlock / 64;
    hipLaunchKernelGGL(track_loss_grad_kernel, dim3((unsigned)((N + per - 1) / per)), dim3(kBlock), 0, s, feats, src, dist, valid, N, C,
                       dist_w, grad_feats, grad_dist, loss);
    return hipGetLastError();
}

hipError_t launch_rigid_update(const float *last, int I, int n, const float *grad_pts, float *t, float *w, float *adam_m, float *adam_v,
                               float *step, const float *norms, float eps_rot, float reg_w, float lr, float beta1, float beta2,
                               float eps_adam, hipStream_t s)
{
    if (I == 0) return hipSuccess;
    hipLaunchKernelGGL(rigid_update_kernel, dim3((unsigned)I), dim3(kBlock), 0, s, last, n, grad_pts, t, w, adam_m, adam_v, step, norms,
                       eps_rot, reg_w, lr, beta1, beta2, eps_adam);
    return hipGetLastError();
}

}  // namespace d3f
