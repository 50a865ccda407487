/*
 * oracle/d3f_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp32 CPU restatement of the d3fields field-query hot path.  It exists only
 * so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check
 * (never replace) the HIP path.  Nothing under d3fields_amd/ may import, link or
 * call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py compares every function here
 * with golden vectors produced by importing the reference (oracle/gen_golden.py,
 * run in the build container where /root/reference is mounted).
 *
 * Every rounding step below is deliberate: the file is compiled with
 * -ffp-contract=off, so a*b+c is two roundings unless written fmaf(a,b,c).  The
 * sequence was matched bit-for-bit against the reference running on torch-CPU:
 *   - 4x4 @ 4x1 bmm of fusion.py:45      -> left-to-right sum of 4 rounded products
 *   - F.grid_sample(bilinear) fusion.py:75 -> fmaf chain nw,ne,sw,se
 *   - .sum(0) over views fusion.py:364,385 -> sequential v = 0..V-1 starting from +0
 *
 * Reference lines restated (paths relative to the reference tree):
 *   fusion.py:32-55    project_points_coords
 *   fusion.py:57-77    interpolate_feats
 *   fusion.py:305-394  Fusion.eval
 *   fusion.py:396-436  Fusion.eval_dist
 *   fusion.py:109-116  onehot2instance
 *   utils/corr_utils.py:4-106  descriptor similarity helpers
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads the OpenMP split of d3f_oracle_eval will use (bench.py reports it beside the timing) */
int d3f_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define ORACLE_MODE_EVAL 0
#define ORACLE_MODE_EVAL_DIST 1

/* fusion.py:44  KRt = K @ Rt  (3x3 @ 3x4, fp32, k-sequential, no fma) */
static void oracle_krt(const float *K, const float *Rt, float *M)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) {
                float p = K[i * 3 + k] * Rt[k * 4 + j];
                acc = acc + p;
            }
            M[i * 4 + j] = acc;
        }
}

typedef struct {
    float gx, gy;   /* normalised image coords, fusion.py:72-73 */
    float zc;       /* camera depth after the |z|<1e-4 patch, fusion.py:52-53 */
    int ok;         /* ~invalid_mask, fusion.py:55 */
} oracle_proj;

/* fusion.py:42-55 + 72-73 for one (view, point) */
static oracle_proj oracle_project(const float *M, const float *p, int H, int W)
{
    oracle_proj r;
    float xc = ((M[0] * p[0] + M[1] * p[1]) + M[2] * p[2]) + M[3] * 1.0f;
    float yc = ((M[4] * p[0] + M[5] * p[1]) + M[6] * p[2]) + M[7] * 1.0f;
    float zc = ((M[8] * p[0] + M[9] * p[1]) + M[10] * p[2]) + M[11] * 1.0f;
    r.ok = !(fabsf(zc) < 1e-4f);
    if (!r.ok) zc = 1e-3f;
    float u = xc / zc;
    float w = yc / zc;
    r.gx = u / (float)(W - 1) * 2.0f - 1.0f;
    r.gy = w / (float)(H - 1) * 2.0f - 1.0f;
    r.zc = zc;
    return r;
}

/* grid_sample align_corners=True un-normalisation (ATen grid_sampler_unnormalize) */
static inline float oracle_unnorm(float g, int size)
{
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}

static inline int oracle_inb(float x, float y, int fw, int fh)
{
    /* float compares so that NaN/Inf fall out of bounds like ATen's masks */
    return (x > -1.0f) && (x < (float)fw) && (y > -1.0f) && (y < (float)fh);
}

/* fusion.py:327-333  nearest depth lookup with zeros padding */
static float oracle_nearest_depth(const float *depth_v, int H, int W, float gx, float gy,
                                  float *margin)
{
    float ix = oracle_unnorm(gx, W), iy = oracle_unnorm(gy, H);
    float rx = nearbyintf(ix), ry = nearbyintf(iy);
    if (margin) {
        /* distance (pixels) of ix/iy from the nearest rounding tie */
        float mx = fabsf(fabsf(ix - floorf(ix)) - 0.5f);
        float my = fabsf(fabsf(iy - floorf(iy)) - 0.5f);
        float m = mx < my ? mx : my;
        if (!(m == m)) m = 0.0f;
        *margin = m;
    }
    if (!oracle_inb(rx, ry, W, H)) return 0.0f;
    return depth_v[(int64_t)ry * W + (int64_t)rx];
}

/* fusion.py:373-379  bilinear sample of one texel row of C channels (zeros padding) */
static void oracle_bilinear(const float *map_v, int fh, int fw, int C, float gx, float gy,
                            float *out)
{
    float ix = oracle_unnorm(gx, fw), iy = oracle_unnorm(gy, fh);
    float x0 = floorf(ix), y0 = floorf(iy);
    float tx = ix - x0, ty = iy - y0;
    float ex = 1.0f - tx, sy = 1.0f - ty;
    float wnw = sy * ex, wne = sy * tx, wsw = ty * ex, wse = ty * tx;
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    int inw = oracle_inb(x0, y0, fw, fh), ine = oracle_inb(x1, y0, fw, fh);
    int isw = oracle_inb(x0, y1, fw, fh), ise = oracle_inb(x1, y1, fw, fh);
    const float *pnw = inw ? map_v + ((int64_t)y0 * fw + (int64_t)x0) * C : NULL;
    const float *pne = ine ? map_v + ((int64_t)y0 * fw + (int64_t)x1) * C : NULL;
    const float *psw = isw ? map_v + ((int64_t)y1 * fw + (int64_t)x0) * C : NULL;
    const float *pse = ise ? map_v + ((int64_t)y1 * fw + (int64_t)x1) * C : NULL;
    for (int c = 0; c < C; ++c) {
        float a = pnw ? pnw[c] : 0.0f, b = pne ? pne[c] : 0.0f;
        float d = psw ? psw[c] : 0.0f, e = pse ? pse[c] : 0.0f;
        float r = a * wnw;
        r = fmaf(b, wne, r);
        r = fmaf(d, wsw, r);
        r = fmaf(e, wse, r);
        out[c] = r;
    }
}

/*
 * Fusion.eval (mode 0, fusion.py:305-394) / Fusion.eval_dist (mode 1, fusion.py:396-436).
 *   depth [V,H,W], K [V,3,3], Rt [V,3,4], pts [n,3]
 *   maps[s] [V,fh[s],fw[s],C[s]] channels-last; out_sets[s] [n,C[s]];
 *   out_inter[s] (nullable) [V,n,C[s]]  (the reference's '<k>_inter', fusion.py:389-390)
 *   out_margin (nullable) [n]: smallest distance of any per-view quantity of the point from
 *   a discontinuous decision (nearest-pixel tie in pixels, |z|-1e-4, dist+mu, depth>0);
 *   tests use it to list knife-edge points.
 */
int d3f_oracle_eval(int V, int H, int W, const float *depth, const float *K, const float *Rt,
                    const float *pts, int64_t n, int n_sets, const float *const *maps,
                    const int *fh, const int *fw, const int *C, float mu, int mode,
                    float *out_dist, uint8_t *out_valid, float *const *out_sets,
                    float *const *out_inter, float *out_margin)
{
    if (V <= 0 || H <= 1 || W <= 1 || n < 0 || n_sets < 0) return -1;
    float *M = (float *)malloc(sizeof(float) * 12 * (size_t)V);
    int maxC = 1;
    for (int s = 0; s < n_sets; ++s)
        if (C[s] > maxC) maxC = C[s];
    for (int v = 0; v < V; ++v) oracle_krt(K + 9 * v, Rt + 12 * v, M + 12 * v);

    /* points are independent (no cross-point term anywhere in fusion.py:305-394), so the
     * OpenMP split below cannot change any result; OMP_NUM_THREADS=1 gives the scalar port */
#pragma omp parallel
    {
    float *wgt = (float *)malloc(sizeof(float) * (size_t)V);
    float *vld = (float *)malloc(sizeof(float) * (size_t)V);
    oracle_proj *pr = (oracle_proj *)malloc(sizeof(oracle_proj) * (size_t)V);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)maxC);
    float *acc = (float *)malloc(sizeof(float) * (size_t)maxC);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *p = pts + 3 * i;
        float dsum = 0.0f, cnt = 0.0f, margin = INFINITY;
        for (int v = 0; v < V; ++v) {
            pr[v] = oracle_project(M + 12 * v, p, H, W);
            float mpx;
            float d = oracle_nearest_depth(depth + (int64_t)v * H * W, H, W, pr[v].gx, pr[v].gy,
                                           out_margin ? &mpx : NULL);
            float dist = d - pr[v].zc;                                  /* fusion.py:343 / 425 */
            int valid;
            if (mode == ORACLE_MODE_EVAL) {
                valid = (d > 0.0f) && pr[v].ok && (dist > -mu);         /* fusion.py:344 */
                float t = mu - fabsf(dist);                             /* fusion.py:347 */
                t = t > 0.0f ? 0.0f : t;       /* clamp(max=0); NaN stays NaN */
                wgt[v] = expf(t / mu);
                float dc = dist < -mu ? -mu : dist;                     /* fusion.py:358 */
                dc = dc > mu ? mu : dc;
                dist = dc;
            } else {
                valid = (d > 0.0f) && pr[v].ok;                         /* fusion.py:426 */
                wgt[v] = 1.0f;
            }
            vld[v] = valid ? 1.0f : 0.0f;
            dsum = dsum + dist * vld[v];                                /* fusion.py:364 / 429 */
            cnt = cnt + vld[v];
            if (out_margin) {
                float m = mpx;
                float mz = fabsf(fabsf(pr[v].zc) - 1e-4f);
                if (mz < m) m = mz;
                float md = fabsf((d - pr[v].zc) + mu);
                if (mode == ORACLE_MODE_EVAL && md < m) m = md;
                if (m < margin) margin = m;
            }
        }
        int all_invalid = (cnt == 0.0f);                                /* fusion.py:366 / 431 */
        float denom = cnt + 1e-6f;
        float dist_out = dsum / denom;
        if (mode == ORACLE_MODE_EVAL && all_invalid) dist_out = 1e3f;   /* fusion.py:367 */
        out_dist[i] = dist_out;
        out_valid[i] = all_invalid ? 0 : 1;
        if (out_margin) out_margin[i] = margin;

        for (int s = 0; s < n_sets; ++s) {
            int Cs = C[s];
            for (int c = 0; c < Cs; ++c) acc[c] = 0.0f;
            for (int v = 0; v < V; ++v) {
                oracle_bilinear(maps[s] + (int64_t)v * fh[s] * fw[s] * Cs, fh[s], fw[s], Cs,
                                pr[v].gx, pr[v].gy, tmp);
                if (out_inter && out_inter[s])
                    memcpy(out_inter[s] + ((int64_t)v * n + i) * Cs, tmp, sizeof(float) * Cs);
                for (int c = 0; c < Cs; ++c) {
                    float t = tmp[c] * vld[v];                          /* fusion.py:385 */
                    t = t * wgt[v];
                    acc[c] = acc[c] + t;
                }
            }
            float *o = out_sets[s] + i * Cs;
            for (int c = 0; c < Cs; ++c) o[c] = all_invalid ? 0.0f : acc[c] / denom; /* :385-386 */
        }
    }
    free(wgt); free(vld); free(pr); free(tmp); free(acc);
    }
    free(M);
    return 0;
}

/* fusion.py:109-116 onehot2instance: argmax over the last dim (first max wins; a NaN
 * counts as the maximum, like torch.argmax / np.argmax) -> uint8 */
int d3f_oracle_onehot2instance(const float *onehot, int64_t n, int NI, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *r = onehot + i * NI;
        int best = 0;
        float bv = r[0];
        for (int c = 1; c < NI; ++c) {
            float x = r[c];
            if (bv != bv) break;            /* NaN already found: it stays the winner */
            if (x > bv || x != x) { bv = x; best = c; }
        }
        out[i] = (uint8_t)best;
    }
    return 0;
}

/* fusion.py:90-107 instance2onehot: out[i, c] = (instance[i] == c) */
int d3f_oracle_instance2onehot(const uint8_t *inst, int64_t n, int NI, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < NI; ++c) out[i * NI + c] = (inst[i] == c) ? 1 : 0;
    return 0;
}

/*
 * corr_utils distances.  dist_type: 0 = 'l2' (torch.norm / np.linalg.norm), 1 = 'square'.
 * The reductions over C are accumulated in double and rounded once: the reference's
 * own summation order is a vectorised/cascaded one that differs between numpy and
 * torch, so the oracle states the mathematically exact value and tests compare with
 * a relative tolerance (1e-5, BASELINE.json north_star).
 *
 * src is addressed through explicit strides so that both layouts of the reference are
 * covered: compute_similarity takes [B,H,W,C] (channel stride 1), compute_similarity_tensor
 * and compute_dist_tensor take [B,C,*dim] (channel stride prod(dim)).
 *   pos = b*inner + j  (j in [0,inner));   element (pos,c) at src[b*stride_b + j*stride_i + c*stride_c]
 */
static double oracle_dist(const float *a, int64_t stride_c, const float *t, int C, int dist_type)
{
    double s = 0.0;
    for (int c = 0; c < C; ++c) {
        float d = a[c * stride_c] - t[c];        /* fp32 subtraction, as in the reference */
        s += (double)d * (double)d;
    }
    return dist_type == 0 ? sqrt(s) : s;
}

/* corr_utils.py:44-61 compute_dist_tensor -> out [B*inner] */
int d3f_oracle_dist_to_target(const float *src, int64_t B, int64_t inner, int C, int64_t stride_b,
                              int64_t stride_i, int64_t stride_c, const float *tgt, int dist_type,
                              float *out)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t j = 0; j < inner; ++j)
            out[b * inner + j] =
                (float)oracle_dist(src + b * stride_b + j * stride_i, stride_c, tgt, C, dist_type);
    return 0;
}

/* corr_utils.py:4-19 compute_similarity: exp(-dist*scale) */
int d3f_oracle_similarity_exp(const float *src, int64_t B, int64_t inner, int C, int64_t stride_b,
                              int64_t stride_i, int64_t stride_c, const float *tgt, float scale,
                              int dist_type, float *out)
{
    d3f_oracle_dist_to_target(src, B, inner, C, stride_b, stride_i, stride_c, tgt, dist_type, out);
    for (int64_t k = 0; k < B * inner; ++k) out[k] = expf(-out[k] * scale);
    return 0;
}

/* softmax over the leading dim of a [R, Ccols] fp32 matrix, column-wise, in place
 * (torch.softmax(x, dim=0): max-subtracted, corr_utils.py:39,102) */
static void oracle_softmax_dim0(float *x, int64_t R, int64_t Ccols, float scale)
{
    for (int64_t j = 0; j < Ccols; ++j) {
        float m = -INFINITY;
        for (int64_t i = 0; i < R; ++i) {
            float v = -x[i * Ccols + j] * scale;
            x[i * Ccols + j] = v;
            if (v > m) m = v;
        }
        double s = 0.0;
        for (int64_t i = 0; i < R; ++i) s += (double)expf(x[i * Ccols + j] - m);
        for (int64_t i = 0; i < R; ++i)
            x[i * Ccols + j] = (float)((double)expf(x[i * Ccols + j] - m) / s);
    }
}

/* corr_utils.py:21-42 compute_similarity_tensor: softmax(-dist*scale, dim=0) over B */
int d3f_oracle_similarity_softmax(const float *src, int64_t B, int64_t inner, int C,
                                  int64_t stride_b, int64_t stride_i, int64_t stride_c,
                                  const float *tgt, float scale, int dist_type, float *out)
{
    d3f_oracle_dist_to_target(src, B, inner, C, stride_b, stride_i, stride_c, tgt, dist_type, out);
    oracle_softmax_dim0(out, B, inner, scale);
    return 0;
}

/* corr_utils.py:63-106 compute_similarity_tensor_multi.
 * mode 0: softmax over B1 (the reference's output); mode 1: raw distances (intermediate).
 * argmax_out (nullable) [B2]: argmax over dim 0 of the output (first max wins). */
int d3f_oracle_pairwise(const float *src, const float *tgt, int64_t B1, int64_t B2, int C,
                        float scale, int dist_type, int mode, float *out, int64_t *argmax_out)
{
    for (int64_t i = 0; i < B1; ++i)
        for (int64_t j = 0; j < B2; ++j)
            out[i * B2 + j] = (float)oracle_dist(src + i * C, 1, tgt + j * C, C, dist_type);
    if (mode == 0) oracle_softmax_dim0(out, B1, B2, scale);
    if (argmax_out)
        for (int64_t j = 0; j < B2; ++j) {
            int64_t best = 0;
            float bv = out[j];
            for (int64_t i = 1; i < B1; ++i) {
                float v = out[i * B2 + j];
                if (mode == 0 ? (v > bv) : (v < bv)) { bv = v; best = i; }
            }
            argmax_out[j] = best;
        }
    return 0;
}

/* utils/my_utils.py:478-497 fps_np: farthest point sampling, float32 distances
 * sqrt((dx*dx + dy*dy) + dz*dz) (numpy's reduction order for 3 terms), first maximum wins.
 * out_idx [k]; returns dist.max() after the last update through *out_maxdist. */
int d3f_oracle_fps(const float *pts, int64_t n, int k, int64_t init_idx, int64_t *out_idx, float *out_maxdist)
{
    if (n < 1 || k < 1 || init_idx < 0 || init_idx >= n) return -1;
    float *dist = (float *)malloc(sizeof(float) * (size_t)n);
    int64_t cur = init_idx;
    float best = 0.0f;
    for (int r = 0; r < k; ++r) {
        out_idx[r] = cur;
        const float cx = pts[cur * 3], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
        int64_t arg = 0;
        best = -1.0f;
        for (int64_t i = 0; i < n; ++i) {
            float dx = pts[i * 3] - cx, dy = pts[i * 3 + 1] - cy, dz = pts[i * 3 + 2] - cz;
            float d = sqrtf((dx * dx + dy * dy) + dz * dz);
            if (r > 0 && dist[i] < d) d = dist[i];
            dist[i] = d;
            if (d > best) { best = d; arg = i; }
        }
        cur = arg;
    }
    if (out_maxdist) *out_maxdist = best;
    free(dist);
    return 0;
}
