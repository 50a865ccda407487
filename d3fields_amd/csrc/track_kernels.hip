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

#include <atomic>

#include "d3f_internal.h"
#include "d3f_device.h"

namespace d3f {

struct Rot { float m[9]; float th, a, b, s, c; bool clamped; };           // s, c: sin / cos of th (the update's chain rule reuses them)

// so3_exp_map of one axis-angle vector
__device__ __forceinline__ Rot exp_map(float wx, float wy, float wz, float eps)
{
    Rot r;
    const float nrm = (wx * wx + wy * wy) + wz * wz;
    r.clamped = !(nrm >= eps);                       // torch.clamp(nrm, eps): gradient passes where nrm >= eps
    const float th = sqrtf(fmaxf(nrm, eps));
    const float inv = 1.0f / th;
    r.th = th;
    r.s = sinf(th);
    r.c = cosf(th);
    r.a = inv * r.s;
    r.b = inv * inv * (1.0f - r.c);
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

// chain rule through the transform and the exponential map, the regulariser's gradient and torch.optim.Adam's update of the
// six parameters of one instance.  G[0..2] = sum_p dL/dp', G[3..11] = sum_p p_k dL/dp'_j (row k, column j).  Pure
// arithmetic on registers: the callers load and store the state (plain accesses in rigid_update_kernel; coherent ones
// inside the multi-step launch, where the wave that runs the update changes from step to step).
struct AdamState { float t[3], w[3], m[6], v[6], step; };
struct AdamShared { float g6[6]; float step, step_size, bc2_sqrt; };      // what the six parameter updates of an instance share

// gradient of the six parameters (chain rule + regulariser) and Adam's bias corrections for this step
__device__ __forceinline__ AdamShared rigid_adam_shared(const float *G, const float *t3, const float *w3, float step_in, float nt, float nw,
                                                        float eps_rot, float reg_w, float lr, double ln_b1, double ln_b2)
{
    const float wx = w3[0], wy = w3[1], wz = w3[2];
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
        const float th = r.th, s = r.s, c = r.c;
        const float da_dth = (th * c - s) / (th * th);
        const float db_dth = (th * s - 2.0f * (1.0f - c)) / (th * th * th);
        const float dth = da * da_dth + db * db_dth;
        gw[0] += dth * wx / th; gw[1] += dth * wy / th; gw[2] += dth * wz / th;
    }
    AdamShared o;
    o.g6[0] = G[0]; o.g6[1] = G[1]; o.g6[2] = G[2]; o.g6[3] = gw[0]; o.g6[4] = gw[1]; o.g6[5] = gw[2];
    // regulariser reg_w * (|t|_F + |w|_F): gradient x / |x|_F, 0 at the origin (torch.norm backward)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (nt > 0.0f) o.g6[k] += reg_w * t3[k] / nt;
        if (nw > 0.0f) o.g6[3 + k] += reg_w * w3[k] / nw;
    }
    // torch.optim.Adam (amsgrad off, no weight decay): bias corrections of this step
    const float st = step_in + 1.0f;
    o.step = st;
    // 1 - beta^step in DOUBLE like torch.optim.Adam forms them on the host: -expm1(step * ln beta), with ln beta computed on
    // the host from the decimal value the caller meant (0.999f is 0.99900001287: as a float it already moves 1 - beta2 by 1.3e-5,
    // and an fp32 exp2 / log2 pair moves the early-step corrections by ~6e-5 -- round 3's form).  Two fp64 expm1 per step and wave.
    const float bc1 = (float)(-expm1((double)st * ln_b1));
    const float bc2 = (float)(-expm1((double)st * ln_b2));
    o.step_size = lr / bc1;
    o.bc2_sqrt = sqrtf(bc2);
    return o;
}

// Adam's update of one parameter
__device__ __forceinline__ void rigid_adam_param(float g, float &m, float &v, float &par, float step_size, float bc2_sqrt, float beta1,
                                                 float beta2, float eps_adam)
{
    m = m + (g - m) * (1.0f - beta1);
    v = v * beta2 + (1.0f - beta2) * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps_adam;
    par = par - step_size * (m / denom);
}

__device__ __forceinline__ AdamState rigid_adam_math(const float *G, const AdamState in, float nt, float nw, float eps_rot, float reg_w,
                                                     float lr, float beta1, float beta2, float eps_adam, double ln_b1, double ln_b2)
{
    const AdamShared sh = rigid_adam_shared(G, in.t, in.w, in.step, nt, nw, eps_rot, reg_w, lr, ln_b1, ln_b2);
    AdamState out = in;
    out.step = sh.step;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        rigid_adam_param(sh.g6[k], out.m[k], out.v[k], k < 3 ? out.t[k] : out.w[k - 3], sh.step_size, sh.bc2_sqrt, beta1, beta2, eps_adam);
    return out;
}

__device__ __forceinline__ void rigid_adam_update(int i, const float *G, float *__restrict__ t, float *__restrict__ w,
                                                  float *__restrict__ adam_m, float *__restrict__ adam_v, float *__restrict__ step,
                                                  float nt, float nw, float eps_rot, float reg_w, float lr, float beta1, float beta2,
                                                  float eps_adam, double ln_b1, double ln_b2)
{
    AdamState in;
#pragma unroll
    for (int k = 0; k < 3; ++k) { in.t[k] = t[i * 3 + k]; in.w[k] = w[i * 3 + k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { in.m[k] = adam_m[i * 6 + k]; in.v[k] = adam_v[i * 6 + k]; }
    in.step = step[i];
    const AdamState o = rigid_adam_math(G, in, nt, nw, eps_rot, reg_w, lr, beta1, beta2, eps_adam, ln_b1, ln_b2);
#pragma unroll
    for (int k = 0; k < 3; ++k) { t[i * 3 + k] = o.t[k]; w[i * 3 + k] = o.w[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { adam_m[i * 6 + k] = o.m[k]; adam_v[i * 6 + k] = o.v[k]; }
    step[i] = o.step;
}

// one workgroup per instance: reduce over its n keypoints, chain rule, Adam
__global__ __launch_bounds__(kBlock) void rigid_update_kernel(const float *__restrict__ last, int n, const float *__restrict__ grad_pts,
                                                             float *__restrict__ t, float *__restrict__ w, float *__restrict__ adam_m,
                                                             float *__restrict__ adam_v, float *__restrict__ step,
                                                             const float *__restrict__ norms, float eps_rot, float reg_w, float lr,
                                                             float beta1, float beta2, float eps_adam, double ln_b1, double ln_b2)
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
    rigid_adam_update(i, G, t, w, adam_m, adam_v, step, norms[0], norms[1], eps_rot, reg_w, lr, beta1, beta2, eps_adam, ln_b1, ln_b2);
}

// ---- the WHOLE optimiser step as one launch (round 3) -------------------------------------------------------------------
// The five-launch step above spends its 37 us per iteration in five latency-bound kernels over ~100 keypoints and their
// boundaries.  Per keypoint, everything between the pose parameters and dL/dp' is local: transform, projection, nearest
// depth, weights, the bilinear gather of the descriptor, |f - src|, its gradient, and the backward of the query (three dot
// products per view over the same corner texels).  So ONE wave per keypoint does all of it -- the corner texels are
// gathered once and kept in registers for the forward sum AND the backward dot products -- and leaves dL/dp' (12 bytes).
// Only the reduction over an instance's keypoints and Adam couple the waves: the last wave to arrive (a returning atomic)
// does that for every instance -- instances side by side in lane groups, one parameter per lane -- so a step is one launch
// (d3f_track_step), and ALL steps of a frame are one launch too (d3f_track_run): the waves of step k+1 wait for step k's
// parameters, which travel as step-tagged words.  A step is a chain of dependent operations on ONE wave per SIMD, so its
// time is latency: what can run side by side does (lane v prepares view v's projection, corner set-up and chain rule; the
// views' loads and reductions are in flight together).  Same formulas as rigid_transform / fused_eval / track_loss_grad /
// fused_eval_backward / rigid_update; sums run in another order, so the result agrees with the five-launch step to
// rounding (tests pin both to the keypoints the reference's own loop returned).
constexpr int kTrackMaxViews = 8;
constexpr int kTrackMaxInst = 16;           // instances of the coherent small-problem path (state staged through LDS)

// Device-coherent scalar accesses (agent scope, relaxed: they go to the memory side, past this XCD's L2) for the few words
// the waves of a launch exchange: per-keypoint gradients, pose parameters, Adam's state, the loss.  With them a step needs
// no device-scope FENCE -- which on gfx950 writes back and invalidates the whole L2 of the XCD and sends the next step's
// depth and texel reads back to memory -- only program order: a wave's stores are complete (s_waitcnt) before its
// arrival is counted, and the updated parameters travel as (value, step tag) words that are their own flag.
__device__ __forceinline__ float ld_coh(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void order_release()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // compiler order; no cache maintenance at this scope
    __builtin_amdgcn_s_waitcnt(0);                              // every store / atomic of this wave has been acknowledged
}
__device__ __forceinline__ void order_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

template <int NVEC>        // float4 channel vectors per lane: ceil(C / 256)
__global__ __launch_bounds__(64) void track_step_kernel(const TrackStepParams P)
{
    __shared__ float krt[kTrackMaxViews * 12];
    __shared__ float sh_g[kTrackMaxResident * 3], sh_l[kTrackMaxResident * 3];    // the updating wave: gradients, keypoints
    __shared__ float sh_p[kTrackMaxInst * 19 + 2];                                // ... t, w, Adam m / v / step, loss sums
    const int lane = threadIdx.x;
    const int p = blockIdx.x;                               // one wave per keypoint
    const int V = P.V, N = P.I * P.n, C = P.map.C, cvec = C / 4;
    const MapDesc &m = P.map;
    // small problems (every tracking frame of the reference: ~100 keypoints, a few instances) exchange their words with
    // coherent accesses and need no device-scope fence; larger ones (one step per launch only) keep fences + plain accesses
    const bool small = N <= kTrackMaxResident && P.I <= kTrackMaxInst;
    // every load that depends on nothing is issued before the first wait: pose parameters, the keypoint, its source
    // descriptor, and K / pose inside compute_krt
    const int inst = p / P.n;
    const float lx = P.last[p * 3], ly = P.last[p * 3 + 1], lz = P.last[p * 3 + 2];
    f32x4 srcv[NVEC];
#pragma unroll
    for (int k = 0; k < NVEC; ++k) {
        const int cv = lane + 64 * k;
        srcv[k] = cv < cvec ? load_vec<f32x4>(P.src + (int64_t)p * C + cv * 4) : (f32x4)0.0f;
    }
    compute_krt(P.K, P.pose, V, krt, 64);
    __syncthreads();
    // P.iters optimiser steps in this launch (d3f_track_run: > 1, every wave resident, a device-wide barrier between steps)
    for (int it = 0; it < P.iters; ++it) {
    float w0, w1, w2, t0, t1, t2;                                                        // step it-1's update
    if (small && it > 0) {
        // Waiting for step it-1's update IS reading its result: the six parameters of this instance are published as 64-bit
        // words (value, step tag), lanes 0..5 poll one word each until all six carry tag `it` (bounded; a wave that never
        // sees them poisons the loss with NaN and leaves)
        unsigned long long word = 0ull;
        unsigned int polls = 0;
        bool seen = false;
        while (!seen) {
            if (lane < 6) word = __hip_atomic_load(P.par + inst * 6 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = __all(lane >= 6 || (unsigned int)(word >> 32) == (unsigned int)it);
            if (!seen) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > (1u << 21)) break;
            }
        }
        if (!seen) {
            // the bounded wait expired (the device was held by other work): poison the loss AND leave a sentinel in the spare
            // counter word, so that the host can tell a stall from a NaN that came out of the data
            if (lane < 3) st_coh(P.loss_out + lane, __builtin_nanf(""));
            if (lane == 0) st_coh(P.counter + 1, D3F_TRACK_STALL_SENTINEL);
            return;
        }
        const float val = __uint_as_float((unsigned int)word);
        t0 = __shfl(val, 0, 64); t1 = __shfl(val, 1, 64); t2 = __shfl(val, 2, 64);
        w0 = __shfl(val, 3, 64); w1 = __shfl(val, 4, 64); w2 = __shfl(val, 5, 64);
    } else if (small) {
        w0 = ld_coh(P.w + inst * 3); w1 = ld_coh(P.w + inst * 3 + 1); w2 = ld_coh(P.w + inst * 3 + 2);
        t0 = ld_coh(P.t + inst * 3); t1 = ld_coh(P.t + inst * 3 + 1); t2 = ld_coh(P.t + inst * 3 + 2);
    } else {
        w0 = P.w[inst * 3]; w1 = P.w[inst * 3 + 1]; w2 = P.w[inst * 3 + 2];
        t0 = P.t[inst * 3]; t1 = P.t[inst * 3 + 1]; t2 = P.t[inst * 3 + 2];
    }
    // ---- transform (rigid_transform_kernel) ----
    const Rot R = exp_map(w0, w1, w2, P.eps_rot);
    const float tt3[3] = {t0, t1, t2};
    float q[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = lx * R.m[0 * 3 + j];
        acc = fmaf(ly, R.m[1 * 3 + j], acc);
        acc = fmaf(lz, R.m[2 * 3 + j], acc);
        q[j] = acc + tt3[j];
    }
    if (lane < 3) P.out_pts[p * 3 + lane] = q[lane];
    // ---- phase A: lane v evaluates view v (fused_eval / fused_eval_backward phase A) ----
    const float mu = P.mu, Wm1 = (float)(P.W - 1), Hm1 = (float)(P.H - 1);
    float a_gx = 0.0f, a_gy = 0.0f, a_wgt = 0.0f, a_valid = 0.0f, a_zc = 1.0f, a_u = 0.0f, a_w = 0.0f, a_dist = 0.0f;
    if (lane < V) {
        const Proj pr = project_point(krt + lane * 12, q[0], q[1], q[2], Wm1, Hm1);
        const float d = nearest_depth(P.depth, lane, P.H, P.W, pr.gx, pr.gy);
        const float dist = d - pr.zc;
        const bool valid = (d > 0.0f) && pr.ok && (dist > -mu);
        float tt = mu - fabsf(dist);
        tt = tt > 0.0f ? 0.0f : tt;
        a_gx = pr.gx; a_gy = pr.gy; a_wgt = expf(tt / mu); a_valid = valid ? 1.0f : 0.0f;
        a_zc = pr.zc; a_u = pr.u; a_w = pr.w; a_dist = dist;
    }
    float cnt = 0.0f, dsum = 0.0f;
    for (int v = 0; v < V; ++v) {
        const float vv = __shfl(a_valid, v, 64), dd = __shfl(a_dist, v, 64);
        float dc = dd < -mu ? -mu : dd;
        dc = dc > mu ? mu : dc;
        dsum = dsum + dc * vv;                                            // fusion.py:358-364
        cnt = cnt + vv;
    }
    const bool any_valid = cnt != 0.0f;
    const float inv = 1.0f / (cnt + 1e-6f);
    const float dist_out = any_valid ? dsum / (cnt + 1e-6f) : 1e3f;      // fusion.py:366-367
    // ---- forward gather: the corner vectors of every valid view stay in registers ----
    // Lane v prepares view v's corner set-up (texel coordinates, bounds, weights, element offsets) -- all views at once,
    // not one after the other -- and the views' loads are issued from what lane v hands over (v_readlane: v is a constant
    // after unrolling).
    f32x4 ca[kTrackMaxViews][NVEC], cb[kTrackMaxViews][NVEC], cd[kTrackMaxViews][NVEC], ce[kTrackMaxViews][NVEC];
    float wsy[kTrackMaxViews], wex[kTrackMaxViews], wtx[kTrackMaxViews], wty[kTrackMaxViews];
    f32x4 acc[NVEC];
#pragma unroll
    for (int k = 0; k < NVEC; ++k) acc[k] = (f32x4)0.0f;
    float c_sy = 0.0f, c_ex = 0.0f, c_tx = 0.0f, c_ty = 0.0f;
    int c_in = 0;                                                         // bit 0..3: nw, ne, sw, se inside the map; 0: view not valid
    int64_t c_onw = 0, c_one = 0, c_osw = 0, c_ose = 0;                   // element offsets of the four corner texels
    if (lane < V && a_valid != 0.0f) {
        const float ix = unnormalize(a_gx, m.fw), iy = unnormalize(a_gy, m.fh);
        const float x0 = floorf(ix), y0 = floorf(iy);
        c_tx = ix - x0; c_ty = iy - y0;
        c_ex = 1.0f - c_tx; c_sy = 1.0f - c_ty;
        const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
        const bool inw = in_bounds(x0, y0, m.fw, m.fh), ine = in_bounds(x1, y0, m.fw, m.fh);
        const bool isw = in_bounds(x0, y1, m.fw, m.fh), ise = in_bounds(x1, y1, m.fw, m.fh);
        const int xi0 = (inw || isw) ? (int)x0 : 0, yi0 = (inw || ine) ? (int)y0 : 0;
        const int xi1 = (ine || ise) ? (int)x1 : 0, yi1 = (isw || ise) ? (int)y1 : 0;
        const int64_t bvo = (int64_t)lane * m.sv;
        c_onw = bvo + (int64_t)yi0 * m.sy + (int64_t)xi0 * m.sx; c_one = bvo + (int64_t)yi0 * m.sy + (int64_t)xi1 * m.sx;
        c_osw = bvo + (int64_t)yi1 * m.sy + (int64_t)xi0 * m.sx; c_ose = bvo + (int64_t)yi1 * m.sy + (int64_t)xi1 * m.sx;
        c_in = (inw ? 1 : 0) | (ine ? 2 : 0) | (isw ? 4 : 0) | (ise ? 8 : 0) | 16;
    }
#pragma unroll
    for (int v = 0; v < kTrackMaxViews; ++v) {
        if (v >= V) break;
        const int in_v = __shfl(c_in, v, 64);
        wsy[v] = __shfl(c_sy, v, 64); wex[v] = __shfl(c_ex, v, 64); wtx[v] = __shfl(c_tx, v, 64); wty[v] = __shfl(c_ty, v, 64);
#pragma unroll
        for (int k = 0; k < NVEC; ++k) ca[v][k] = cb[v][k] = cd[v][k] = ce[v][k] = (f32x4)0.0f;
        if (in_v == 0) continue;                                          // exact: the term is (+-0) for finite maps
        const float *pnw = m.data + __shfl(c_onw, v, 64), *pne = m.data + __shfl(c_one, v, 64);
        const float *psw = m.data + __shfl(c_osw, v, 64), *pse = m.data + __shfl(c_ose, v, 64);
#pragma unroll
        for (int k = 0; k < NVEC; ++k) {
            const int cv = lane + 64 * k;
            if (cv < cvec) {
                ca[v][k] = (in_v & 1) ? load_vec<f32x4>(pnw + cv * 4) : (f32x4)0.0f;
                cb[v][k] = (in_v & 2) ? load_vec<f32x4>(pne + cv * 4) : (f32x4)0.0f;
                cd[v][k] = (in_v & 4) ? load_vec<f32x4>(psw + cv * 4) : (f32x4)0.0f;
                ce[v][k] = (in_v & 8) ? load_vec<f32x4>(pse + cv * 4) : (f32x4)0.0f;
            }
        }
    }
    // the corner loads of EVERY view are in flight before the first is used (one memory round trip, not one per view)
#pragma unroll
    for (int v = 0; v < kTrackMaxViews; ++v) {
        if (v >= V) break;
        const float vv = __shfl(a_valid, v, 64), wg = __shfl(a_wgt, v, 64);
        if (vv == 0.0f) continue;                                         // exact: the term is (+-0) for finite maps
#pragma unroll
        for (int k = 0; k < NVEC; ++k) {
            f32x4 s_ = ca[v][k] * (wsy[v] * wex[v]);                      // ATen bilinear: fma chain nw,ne,sw,se
            s_ = v_fma<f32x4>(cb[v][k], wsy[v] * wtx[v], s_);
            s_ = v_fma<f32x4>(cd[v][k], wty[v] * wex[v], s_);
            s_ = v_fma<f32x4>(ce[v][k], wty[v] * wtx[v], s_);
            acc[k] = acc[k] + (s_ * vv) * wg;                             // fusion.py:385
        }
    }
    // ---- loss and its gradient w.r.t. the fused descriptor (track_loss_grad_kernel) ----
    f32x4 g[NVEC];
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < NVEC; ++k) {
        const int cv = lane + 64 * k;
        g[k] = (f32x4)0.0f;
        if (cv < cvec) {
            const f32x4 f = any_valid ? acc[k] / (cnt + 1e-6f) : (f32x4)0.0f;     // fusion.py:385-386
            const f32x4 d = f - srcv[k];
            g[k] = d;
            ss = fmaf(d.x, d.x, ss); ss = fmaf(d.y, d.y, ss); ss = fmaf(d.z, d.z, ss); ss = fmaf(d.w, d.w, ss);
        }
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float nrm = sqrtf(ss);
    const float vf = any_valid ? 1.0f : 0.0f;
    const float invN = 1.0f / (float)N;
    const float scale = (nrm > 0.0f) ? (vf * invN) / nrm : 0.0f;          // norm backward: 0 at a zero difference
    const float cdist = dist_out * vf;
    const float gd = (any_valid && cdist >= 0.0f) ? P.dist_w * invN * vf : 0.0f;   // clamp(min=0) passes where x >= 0
    float *const loss_slot = P.loss_acc + 2 * (it & 1);       // two slots: a slot's clearing has landed long before its next use
    if (lane == 0) {
        atomicAdd(loss_slot + 0, nrm * vf * invN);
        atomicAdd(loss_slot + 1, P.dist_w * fmaxf(cdist, 0.0f) * invN);
    }
    // ---- backward of the query (fused_eval_backward_kernel): per valid view three dot products, then the chain rule ----
    // The dot products of all views are reduced together (independent shuffle chains), then lane v runs view v's chain rule
    // on the values it already holds from phase A, and the views' contributions are summed over lanes 0..7.
    const float sxm = 0.5f * (float)(m.fw - 1), sym = 0.5f * (float)(m.fh - 1);    // d(ix)/d(gx), d(iy)/d(gy)
    float ds[kTrackMaxViews], dx[kTrackMaxViews], dy[kTrackMaxViews];
#pragma unroll
    for (int v = 0; v < kTrackMaxViews; ++v) {
        ds[v] = dx[v] = dy[v] = 0.0f;
        if (v >= V) continue;
#pragma unroll
        for (int k = 0; k < NVEC; ++k) {
            const f32x4 go = g[k] * scale;                               // dL/df (zero on idle lanes: g = 0)
            f32x4 s_ = ca[v][k] * (wsy[v] * wex[v]);
            s_ = v_fma<f32x4>(cb[v][k], wsy[v] * wtx[v], s_);
            s_ = v_fma<f32x4>(cd[v][k], wty[v] * wex[v], s_);
            s_ = v_fma<f32x4>(ce[v][k], wty[v] * wtx[v], s_);
            const f32x4 dsx = (cb[v][k] - ca[v][k]) * wsy[v] + (ce[v][k] - cd[v][k]) * wty[v];      // ds/dix
            const f32x4 dsy = (cd[v][k] - ca[v][k]) * wex[v] + (ce[v][k] - cb[v][k]) * wtx[v];      // ds/diy
            ds[v] += hsum<f32x4>(go * s_);
            dx[v] += hsum<f32x4>(go * dsx);
            dy[v] += hsum<f32x4>(go * dsy);
        }
    }
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int v = 0; v < kTrackMaxViews; ++v) {
            if (v >= V) continue;
            ds[v] += __shfl_xor(ds[v], off, 64);
            dx[v] += __shfl_xor(dx[v], off, 64);
            dy[v] += __shfl_xor(dy[v], off, 64);
        }
    float my_ds = 0.0f, my_dx = 0.0f, my_dy = 0.0f;
#pragma unroll
    for (int v = 0; v < kTrackMaxViews; ++v)
        if (lane == v) { my_ds = ds[v]; my_dx = dx[v]; my_dy = dy[v]; }
    float gxw = 0.0f, gyw = 0.0f, gzw = 0.0f;
    if (lane < V && a_valid != 0.0f) {
        const float wg = a_wgt, zc = a_zc, uu = a_u, ww = a_w, dd = a_dist;
        const float g_wgt = inv * my_ds;                                  // dL/dwgt_v
        const float g_gx = inv * wg * (my_dx * sxm), g_gy = inv * wg * (my_dy * sym);
        float g_dist = (dd >= -mu && dd <= mu) ? gd * inv : 0.0f;         // clamp passes inside [-mu, mu]
        if (mu - fabsf(dd) <= 0.0f) {                                     // the weight passes where mu - |dist| <= 0
            const float sgn = dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f);
            g_dist += g_wgt * wg * (-sgn) / mu;
        }
        const float g_u = g_gx * 2.0f / Wm1, g_w = g_gy * 2.0f / Hm1;
        const float g_xc = g_u / zc, g_yc = g_w / zc;
        const float g_zc = -(g_u * uu + g_w * ww) / zc - g_dist;
        const float *M = krt + lane * 12;
        gxw = g_xc * M[0] + g_yc * M[4] + g_zc * M[8];
        gyw = g_xc * M[1] + g_yc * M[5] + g_zc * M[9];
        gzw = g_xc * M[2] + g_yc * M[6] + g_zc * M[10];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {                               // lanes 0..7 (kTrackMaxViews): lane 0 gets the sum
        gxw += __shfl_xor(gxw, off, 64);
        gyw += __shfl_xor(gyw, off, 64);
        gzw += __shfl_xor(gzw, off, 64);
    }
    if (lane == 0) {
        if (small) { st_coh(P.grad_pts + p * 3 + 0, gxw); st_coh(P.grad_pts + p * 3 + 1, gyw); st_coh(P.grad_pts + p * 3 + 2, gzw); }
        else { P.grad_pts[p * 3 + 0] = gxw; P.grad_pts[p * 3 + 1] = gyw; P.grad_pts[p * 3 + 2] = gzw; }
    }
    // ---- the last wave to finish reduces per instance and steps Adam (rigid_update_kernel) ----
    // The arrival counter runs on across the steps of a launch (step `it` is complete at N*(it+1) arrivals); the wave that
    // completes it updates the parameters and publishes them tagged with it+1, which is what the waves of the next step
    // wait for at its top.  Nothing else is ordered by hand: whatever the updating wave stores besides the tagged words
    // (Adam's state, the cleared loss slot) is read again only by a later updating wave, i.e. after this wave's own next
    // arrival, before which it waits for its stores (order_release).
    if (small) order_release(); else __threadfence();                     // grad_pts / loss are out before the arrival counts
    unsigned int ticket = 0;
    if (lane == 0) ticket = atomicAdd(P.counter, 1u);
    ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
    const bool last_step = it + 1 == P.iters;
    if (ticket != (unsigned int)N * (unsigned int)(it + 1) - 1u) {
        if (last_step) return;
        continue;                                                         // (a multi-step launch is always `small`)
    }
    if (small) {
        // ---- the update, coherent form: ONE round of loads (everything the update reads, spread over the lanes, into LDS),
        // arithmetic, one round of stores ----
        order_acquire();
        for (int k = lane; k < N * 3; k += 64) { sh_g[k] = ld_coh(P.grad_pts + k); sh_l[k] = P.last[k]; }
        const int I = P.I;
        for (int k = lane; k < I * 19 + 2; k += 64) {
            const float *src_k = k < 3 * I ? P.t + k : k < 6 * I ? P.w + (k - 3 * I) : k < 12 * I ? P.adam_m + (k - 6 * I)
                               : k < 18 * I ? P.adam_v + (k - 12 * I) : k < 19 * I ? P.step + (k - 18 * I) : loss_slot + (k - 19 * I);
            sh_p[k] = ld_coh(src_k);
        }
        __syncthreads();
        float st = 0.0f, sw = 0.0f;                                       // |t|_F, |w|_F over ALL instances, before the update
        for (int k = 0; k < I * 3; ++k) { st += sh_p[k] * sh_p[k]; sw += sh_p[3 * I + k] * sh_p[3 * I + k]; }
        const float nt = sqrtf(st), nw = sqrtf(sw);
        // Instances side by side: 64 / 32 / 16 / 8 lanes per instance (I = 1 / 2 / <= 4 / more; eight instances per pass),
        // a lane group reduces its instance's keypoints, every lane of the group forms the shared part (chain rule, bias
        // corrections: one latency for all instances) and lanes 0..5 of the group update one parameter each.
        const int glog = I <= 1 ? 6 : I <= 2 ? 5 : I <= 4 ? 4 : 3, gsz = 1 << glog;
        for (int ib = 0; ib < I; ib += 64 >> glog) {
            const int i_raw = ib + (lane >> glog), k = lane & (gsz - 1);
            const bool active = i_raw < I;
            const int i = active ? i_raw : I - 1;
            float G[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) G[q] = 0.0f;
            for (int pp = k; pp < P.n; pp += gsz) {
                const int b = (i * P.n + pp) * 3;
                const float gx = sh_g[b], gy = sh_g[b + 1], gz = sh_g[b + 2];
                const float px = sh_l[b], py = sh_l[b + 1], pz = sh_l[b + 2];
                G[0] += gx; G[1] += gy; G[2] += gz;
                G[3] += px * gx; G[4] += px * gy; G[5] += px * gz;
                G[6] += py * gx; G[7] += py * gy; G[8] += py * gz;
                G[9] += pz * gx; G[10] += pz * gy; G[11] += pz * gz;
            }
            for (int off = 1; off < gsz; off <<= 1)
#pragma unroll
                for (int q = 0; q < 12; ++q) G[q] += __shfl_xor(G[q], off, 64);
            const float t3[3] = {sh_p[i * 3], sh_p[i * 3 + 1], sh_p[i * 3 + 2]};
            const float w3[3] = {sh_p[3 * I + i * 3], sh_p[3 * I + i * 3 + 1], sh_p[3 * I + i * 3 + 2]};
            const AdamShared sh = rigid_adam_shared(G, t3, w3, sh_p[18 * I + i], nt, nw, P.eps_rot, P.reg_w, P.lr, P.ln_beta1, P.ln_beta2);
            if (active && k < 6) {
                const float g = k == 0 ? sh.g6[0] : k == 1 ? sh.g6[1] : k == 2 ? sh.g6[2] : k == 3 ? sh.g6[3] : k == 4 ? sh.g6[4] : sh.g6[5];
                float mk = sh_p[6 * I + i * 6 + k], vk = sh_p[12 * I + i * 6 + k];
                float par = k < 3 ? sh_p[i * 3 + k] : sh_p[3 * I + i * 3 + (k - 3)];
                rigid_adam_param(g, mk, vk, par, sh.step_size, sh.bc2_sqrt, P.beta1, P.beta2, P.eps_adam);
                st_coh(k < 3 ? P.t + i * 3 + k : P.w + i * 3 + (k - 3), par);
                st_coh(P.adam_m + i * 6 + k, mk);
                st_coh(P.adam_v + i * 6 + k, vk);
                if (!last_step) {                                         // what the next step's waves wait for
                    const unsigned long long word = ((unsigned long long)(unsigned int)(it + 1) << 32) | (unsigned long long)__float_as_uint(par);
                    __hip_atomic_store(P.par + i * 6 + k, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (active && k == 6) st_coh(P.step + i, sh.step);
        }
        if (lane == 0) {
            st_coh(P.loss_out + 0, sh_p[19 * I]);
            st_coh(P.loss_out + 1, sh_p[19 * I + 1]);
            st_coh(P.loss_out + 2, P.reg_w * (nt + nw));
            st_coh(loss_slot + 0, 0.0f); st_coh(loss_slot + 1, 0.0f);    // clean for the step after the next
            if (last_step) { st_coh(P.counter, 0u); st_coh(P.counter + 1, 0u); }      // every wave has arrived: nobody waits any more
        }
        if (last_step) return;
        __syncthreads();                                                  // LDS reads above are done before this wave stages again
        continue;
    }
    // ---- the update for larger problems (one step per launch): fences + plain accesses ----
    __threadfence();                                                      // acquire
    float st = 0.0f, sw = 0.0f;                                           // |t|_F, |w|_F over ALL instances, before the update
    for (int k = 0; k < P.I * 3; ++k) { st += P.t[k] * P.t[k]; sw += P.w[k] * P.w[k]; }
    const float nt = sqrtf(st), nw = sqrtf(sw);
    for (int i = 0; i < P.I; ++i) {
        float G[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) G[k] = 0.0f;
        for (int pp = lane; pp < P.n; pp += 64) {
            const int64_t b = ((int64_t)i * P.n + pp) * 3;
            const float gx = __builtin_nontemporal_load(P.grad_pts + b), gy = __builtin_nontemporal_load(P.grad_pts + b + 1), gz = __builtin_nontemporal_load(P.grad_pts + b + 2);
            const float px = P.last[b], py = P.last[b + 1], pz = P.last[b + 2];
            G[0] += gx; G[1] += gy; G[2] += gz;
            G[3] += px * gx; G[4] += px * gy; G[5] += px * gz;
            G[6] += py * gx; G[7] += py * gy; G[8] += py * gz;
            G[9] += pz * gx; G[10] += pz * gy; G[11] += pz * gz;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k)
            for (int off = 32; off > 0; off >>= 1) G[k] += __shfl_xor(G[k], off, 64);
        if (lane == 0) rigid_adam_update(i, G, P.t, P.w, P.adam_m, P.adam_v, P.step, nt, nw, P.eps_rot, P.reg_w, P.lr, P.beta1, P.beta2, P.eps_adam, P.ln_beta1, P.ln_beta2);
    }
    if (lane == 0) {
        P.loss_out[0] = __builtin_nontemporal_load(P.loss_acc + 0);
        P.loss_out[1] = __builtin_nontemporal_load(P.loss_acc + 1);
        P.loss_out[2] = P.reg_w * (nt + nw);
        P.loss_acc[0] = 0.0f; P.loss_acc[1] = 0.0f;                       // clean for the next step
        P.counter[0] = 0u; P.counter[1] = 0u;
    }
    return;                                                               // (not `small`: one step per launch)
    }
}

// clears what a multi-step launch synchronises through: the loss slots, the arrival counter and the tagged parameter
// words (a memset node of a captured HIP graph replayed 16 such bytes wrongly on ROCm 7.2 -- the second replay found a
// host pointer in them; a kernel node replays as recorded)
__global__ void track_reset_kernel(float *loss_acc, unsigned int *counter, unsigned long long *par, int n_par)
{
    if (threadIdx.x < 4) loss_acc[threadIdx.x] = 0.0f;
    else if (threadIdx.x < 6) counter[threadIdx.x - 4] = 0u;
    for (int k = threadIdx.x; k < n_par; k += 64) par[k] = 0ull;
}

// ln(beta) in double for the decimal number the caller's float stands for (0.9f -> 0.9, 0.999f -> 0.999: the shortest decimal that
// rounds to the float, which is what a Python caller typed and torch.optim.Adam computes with)
double log_of_decimal(float beta)
{
    char buf[32];
    double d = (double)beta;
    for (int digits = 1; digits <= 9; ++digits) {
        snprintf(buf, sizeof(buf), "%.*g", digits, (double)beta);
        d = strtod(buf, nullptr);
        if ((float)d == beta) break;
    }
    return log(d);
}

// How many one-wave workgroups of the multi-step kernel the device can hold at once (they wait for one another, so all of a
// launch must be resident): the runtime's occupancy for THIS kernel x the CUs of THIS device (a partition in CPX mode, a part
// with fewer CUs, spills -- none of which the round-3 constant knew), half of it left to whatever else runs, capped by the
// kernel's static LDS arrays.
// Keypoints one d3f_track_run launch may hold on the CURRENT device: half of the waves that are resident together, the smaller of
// the two kernel variants' figures (the bound protects the in-kernel wait of a multi-step launch).  Cached per device id -- a
// process may drive several GPUs or switch devices between calls -- behind relaxed atomics (a racing first call computes the same
// value twice).
int track_run_capacity()
{
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cached[kMaxDevices];          // 0 = not computed yet (a capacity is never 0 on a working device), stored + 1
    int dev = 0, cus = 0, per1 = 0, per2 = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < kMaxDevices) {
        const int hit = cached[dev].load(std::memory_order_relaxed);
        if (hit > 0) return hit - 1;
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per1, track_step_kernel<1>, 64, 0) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per2, track_step_kernel<2>, 64, 0) != hipSuccess) return 0;
    int cap = cus * (per1 < per2 ? per1 : per2) / 2;
    cap = cap < kTrackMaxResident ? cap : kTrackMaxResident;
    if (dev >= 0 && dev < kMaxDevices) cached[dev].store(cap + 1, std::memory_order_relaxed);
    return cap;
}

hipError_t launch_track_step(const TrackStepParams &P, hipStream_t s)
{
    const int N = P.I * P.n;
    if (N == 0 || P.iters <= 0) return hipSuccess;
    // several steps per launch wait for one another inside the kernel: every wave must be resident
    if (P.iters > 1 && (N > track_run_capacity() || P.I > kTrackMaxInst)) return hipErrorInvalidValue;
    if (P.iters > 1) hipLaunchKernelGGL(track_reset_kernel, dim3(1), dim3(64), 0, s, P.loss_acc, P.counter, P.par, P.I * 6);
    const int nvec = (P.map.C / 4 + 63) / 64;
    if (nvec <= 1) hipLaunchKernelGGL(track_step_kernel<1>, dim3((unsigned)N), dim3(64), 0, s, P);
    else if (nvec == 2) hipLaunchKernelGGL(track_step_kernel<2>, dim3((unsigned)N), dim3(64), 0, s, P);
    else return hipErrorInvalidValue;
    return hipGetLastError();
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
                       eps_rot, reg_w, lr, beta1, beta2, eps_adam, log_of_decimal(beta1), log_of_decimal(beta2));
    return hipGetLastError();
}

}  // namespace d3f
