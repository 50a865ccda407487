// fuse_direct.hip -- the fused 3-D field query of d3fields for gfx950 (MI355X, CDNA4).
//
// One launch does what Fusion.eval does with ~20 torch ops and three [V,N,C] temporaries
// (reference fusion.py:305-394, helpers :32-77): project every query point into the V
// calibrated views, look the nearest depth pixel up, derive the truncated signed distance,
// the per-view validity bit and exp weight, bilinearly sample every requested channels-last
// map and reduce over the views.  The arithmetic contract (operation order, where an fma is
// and is not used) is stated in DESIGN.md §Arithmetic and restated by oracle/d3f_oracle.c;
// this file is compiled with -ffp-contract=off so that a*b+c below is two roundings and
// only fmaf() fuses.
//
// Work decomposition (wave = 64 lanes, 256-thread workgroups, no MFMA: this is gather work)
//   phase A  one LANE per point: projection, depth test, weights for all V views; 'dist' and
//            'valid_mask' leave coalesced; the per-(point,view) record {gx,gy,wgt,valid}
//            goes to LDS (16 B, one ds_write_b128).
//   phase B  per channel map, a GROUP of 2^k lanes per point walks the channel vectors of the
//            four bilinear corners (16-byte loads, consecutive lanes = consecutive channels, so
//            every texel is fetched as whole 64-B..1-KiB coalesced segments), accumulates the
//            V views in registers in view order and stores the fused row once.
// Nothing of size [V,N,C] ever exists and N is not chunked.
//
// This file: the DIRECT kernels (every query no other family takes: small batches, odd channel counts, '<k>_inter', maps not known
// finite, the distance-only pass).  Families and the planner: DESIGN.md 5.2, d3f_plan.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"
#include "fuse_common.h"
#include "fuse_body.h"
#include "dist_views.h"

namespace d3f {

// Entry points over one body: the plain kernel (<= 3 channel vectors per lane) is held to 128 VGPRs = 4 waves per SIMD
// (125 allocated) -- the gather lives on memory-level parallelism; the WIDE variant adds the 4-vector load-use path
// (C = 1024: a whole wave per point) with its natural register count.
template <int MODE>
__global__ __launch_bounds__(kBlock, 4) void fused_eval_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    fused_eval_body<MODE, false>(P);
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void fused_eval_wide_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    fused_eval_body<MODE, true>(P);
}

// fp16-stored maps get their own entry point (all vector counts, fp32 and fp16 maps may be mixed in one call) so
// that the fp32 kernels above keep their register allocation (folding both into one body made them spill)
// (163 VGPR = 3 waves per SIMD; held to 4 it spills 750 B per lane)
template <int MODE>
__global__ __launch_bounds__(kBlock) void fused_eval_f16_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    fused_eval_body<MODE, true, true>(P);
}


// ---- the DISTANCE-ONLY pass (return_names=[], eval_dist; fusion.py:396-436, vis_repr.py:93) as its own entry point (round 6) -------------
// Up to round 6 this pass was a branch of fused_eval_body and inherited the gathers' register allocation (119 VGPRs: four
// waves per SIMD) and a rolled view loop whose depth lookups -- one dependent L2 round trip per view -- were issued one after
// the other: counters said 0.22 wave instructions per cycle and SIMD with the waves parked 54 % of the time.  The work per
// (point, view) is ~100 VALU instructions of IEEE divisions and unfused multiply-adds that the arithmetic contract fixes
//  plus ONE 4-byte load, so what the pass needs is waves and independent chains, not registers:
//   * KRt lives in SGPRs: lanes 0..47 of every wave compute the entries of a batch of four views (compute_krt's
//     k-sequential unfused sums), v_readlane hands them to the scalar file -- no LDS, no barrier, no VGPRs;
//   * the four views of a batch are unrolled: four projections, four depth lookups in flight per lane, then the sums IN VIEW
//     ORDER (dsum and cnt start at +0 and take view 0, 1, ... exactly as the rolled loop did: bit-identical);
//   * five to eight views (config 4: eight): a second batch of four continues the same sums; more: the branch of fused_eval_kernel;
//   * 45-64 VGPRs: seven or eight waves per SIMD (launch_dist_v);
//   * the IEEE divisions in their short form (d3f_device.h: project_point_short) and, on a big batch, the depth pixels looked up in a
//     copy tiled 4 x 8 pixels per cache line (depth_tile_kernel below): 1.89 -> 1.185 ms on the 123 M-point grid of bench.py.
__device__ __forceinline__ void dist_store(const EvalParams &P, int mode, int64_t i, float ds, float cn)
{
    const bool all_invalid = (cn == 0.0f);                      // fusion.py:366
    float dist_out = ds / (cn + 1e-6f);
    if (mode == 0 && all_invalid) dist_out = 1e3f;              // fusion.py:367
    P.out_dist[i] = dist_out;
    P.out_valid[i] = all_invalid ? 0 : 1;
}

// The points of one lane, one after the other, in CALLER order: tile_pts consecutive points per workgroup, lane t takes t, t + 256, ...
// (point loads and the 'dist' / 'valid_mask' stores are then whole lines per wave; bricks of a lattice -- sixteen 4 x 4 x 4
// sub-bricks per workgroup, any lane order -- fragment both and measured 2-8 % slower, rows of 64 x 3.5 times: session 42,
// scripts/notebook/patches/r6_dist_bricks_and_whatifs.patch)
// GRID: the points come from the axis arrays of a d3f_grid (d3f_eval_grid), else from the [n, 3] array -- a template argument, not
// fetch_point's run-time branch: the loop of this kernel is short enough for every scalar branch in it to show (two more
// wave-uniform tests per point cost 8 % in session 52's experiments build)
template <bool GRID, typename BODY>
__device__ __forceinline__ void dist_for_each_point(const EvalParams &P, BODY body)
{
    const int64_t tile_base = (int64_t)blockIdx.x * P.tile_pts;
    const int64_t end = min(tile_base + P.tile_pts, P.n);
#pragma unroll 1
    for (int64_t i = tile_base + threadIdx.x; i < end; i += kBlock) {
        float px, py, pz;
        if constexpr (GRID) fetch_point(P, i, px, py, pz);
        else { px = P.pts[i * 3 + 0]; py = P.pts[i * 3 + 1]; pz = P.pts[i * 3 + 2]; }
        body(i, px, py, pz);
    }
}

// NVQ: the view count (1..4) as a compile-time constant, 0 = five to eight views (two batches of four)
// OCC: waves per SIMD the entry point is held to (8: 64 VGPRs / ~80 SGPRs; 6: 80 / 102 -- see launch_dist_v)
template <int MODE, int NVQ, int OCC, bool TILED, bool GRID>
__global__ __launch_bounds__(kBlock, OCC) void fused_eval_dist_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    const float mu = P.mu;
    const DivConst Wm1 = div_const((float)(P.W - 1)), Hm1 = div_const((float)(P.H - 1));
    if constexpr (NVQ > 0) {
        float M[4][12];
        dist_krt_uniform(dist_krt_lane(P, 0), M);
        dist_for_each_point<GRID>(P, [&](int64_t i, float px, float py, float pz) {
            float ds = 0.0f, cn = 0.0f;
            dist_views<MODE, NVQ, TILED>(P, M, 0, px, py, pz, Wm1, Hm1, mu, ds, cn);
            dist_store(P, MODE, i, ds, cn);
        });
    } else {
        // five to eight views (config 4: eight): two batches of four per point; both batches' KRt entries wait in two VGPRs and
        // are handed to the scalar file in front of each batch (48 v_readlane: ~14 % on top of a batch's four views)
        const int V = P.V;
        const float kr0 = dist_krt_lane(P, 0), kr1 = dist_krt_lane(P, 4);
        dist_for_each_point<GRID>(P, [&](int64_t i, float px, float py, float pz) {
            float ds = 0.0f, cn = 0.0f;
            float M[4][12];
            dist_krt_uniform(kr0, M);
            dist_views<MODE, 4, TILED>(P, M, 0, px, py, pz, Wm1, Hm1, mu, ds, cn);
            __builtin_amdgcn_sched_barrier(0);
            dist_krt_uniform(kr1, M);
            if (V == 8) dist_views<MODE, 4, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, ds, cn);
            else if (V == 7) dist_views<MODE, 3, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, ds, cn);
            else if (V == 6) dist_views<MODE, 2, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, ds, cn);
            else dist_views<MODE, 1, TILED>(P, M, 4, px, py, pz, Wm1, Hm1, mu, ds, cn);
            dist_store(P, MODE, i, ds, cn);
        });
    }
}

// Waves per SIMD the entry point is held to: KRt's 48 SGPRs (three or four views) do not fit beside the rest at eight waves (~80
// SGPRs: 32-47 spilled to VGPR lanes and read back per point); held to six the allocator has 102 and the kernel still runs seven
// waves (1.185 vs 1.235 ms on the 123 M-point grid, session 47).  One or two views fit at eight.  occ8: experiments (D3F_EXP_DIST=8).
template <int MODE, bool TILED, bool GRID>
static void launch_dist_v(const EvalParams &P, dim3 grid, dim3 block, hipStream_t stream, bool occ8)
{
    switch (P.V <= 4 ? P.V : 0) {
    case 1: hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 1, 8, TILED, GRID>), grid, block, 0, stream, P); break;
    case 2: hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 2, 8, TILED, GRID>), grid, block, 0, stream, P); break;
    case 3:
#ifdef D3F_EXPERIMENTS
        if (occ8) { hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 3, 8, TILED, GRID>), grid, block, 0, stream, P); break; }
#endif
        hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 3, 6, TILED, GRID>), grid, block, 0, stream, P);
        break;
    case 4:
#ifdef D3F_EXPERIMENTS
        if (occ8) { hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 4, 8, TILED, GRID>), grid, block, 0, stream, P); break; }
#endif
        hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 4, 6, TILED, GRID>), grid, block, 0, stream, P);
        break;
    default:
#ifdef D3F_EXPERIMENTS
        if (occ8) { hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 0, 8, TILED, GRID>), grid, block, 0, stream, P); break; }
#endif
        hipLaunchKernelGGL((fused_eval_dist_kernel<MODE, 0, 6, TILED, GRID>), grid, block, 0, stream, P);
        break;
    }
}
template <int MODE>
static void launch_dist(const EvalParams &P, dim3 grid, dim3 block, hipStream_t stream)
{
    const bool occ8 = (P.dist_variant & 15) == 8;
    if constexpr (MODE == 0) {          // (eval_dist, MODE 1, has no scratch or grid parameter: keypoint batches)
        if (P.depth_tiled) { launch_dist_v<MODE, true, false>(P, grid, block, stream, occ8); return; }      // (d3f_eval_grid has no scratch parameter either)
        if (P.grid_x) { launch_dist_v<MODE, false, true>(P, grid, block, stream, occ8); return; }
    }
    launch_dist_v<MODE, false, false>(P, grid, block, stream, occ8);
}

// The depth maps in tiles of 4 x 8 pixels -- one 128-byte line each -- for the distance-only pass over a big batch.  Its lookups are
// nearest-pixel gathers whose four lanes of a quad are four consecutive points of the caller's order: a lattice's z column, i.e.
// neighbouring pixels ALONG AN IMAGE COLUMN for an upright camera -- in a row-major map four lines.  The texture addresser serves a
// quad per cycle when its lanes share a line and a lane per cycle otherwise, and with one lookup per ~100 VALU instructions that is
// half of the pass (what-if builds, sessions 43-45: no lookups 0.89 ms, quad-coherent lookups 1.28-1.35 ms, as is 1.75 ms).  In
// 4 x 8 tiles a quad of the 1-mm grid touches 1.26 lines instead of 2.60, of a 5-mm grid 2.1 instead of 3.95
// (scripts/notebook/sim_depth_tiles.py).  One thread per output float; pixels beyond the map are written as 0 and never read.
__global__ __launch_bounds__(kBlock) void depth_tile_kernel(const float *__restrict__ depth, int V, int H, int W, int tw, int th,
                                                            float *__restrict__ out)
{
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t total = (int64_t)V * th * tw * 32;
    if (o >= total) return;
    const int in_tile = (int)(o & 31);
    const int64_t t = o >> 5;
    const int tx = (int)(t % tw);
    const int64_t r = t / tw;
    const int ty = (int)(r % th), v = (int)(r / th);
    const int ix = tx * 4 + (in_tile & 3), iy = ty * 8 + (in_tile >> 2);
    out[o] = (ix < W && iy < H) ? depth[((int64_t)v * H + iy) * W + ix] : 0.0f;
}

hipError_t launch_depth_tiles(const EvalParams &P, float *tiled, hipStream_t stream)
{
    const int64_t total = (int64_t)P.V * P.depth_th * P.depth_tw * 32;
    hipLaunchKernelGGL(depth_tile_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, P.depth, P.V, P.H, P.W,
                       P.depth_tw, P.depth_th, tiled);
    return hipGetLastError();
}

hipError_t launch_direct(const EvalParams &P, int mode, hipStream_t stream)
{
    int64_t ntiles = (P.n + P.tile_pts - 1) / P.tile_pts;
    if (P.walk_nx > 0)
        ntiles = (int64_t)((P.walk_nx + P.walk_tx - 1) / P.walk_tx) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) *
                 ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
    const size_t lds = (size_t)P.crec_offset + (size_t)P.n_pre * P.tile_pts * P.V * 32 + (size_t)P.lds_pad;
    dim3 grid((unsigned)ntiles), block(kBlock);
    if (P.n_maps == 0 && P.walk_nx <= 0 && P.order == nullptr && P.V <= 8 && P.dist_variant >= 0) {
        // the distance-only pass in caller order: its own entry point (KRt in SGPRs, the views of a point in flight together)
        if (mode == 0) launch_dist<0>(P, grid, block, stream);
        else launch_dist<1>(P, grid, block, stream);
        return hipGetLastError();
    }
    bool wide = false, f16 = false;
    for (int s = 0; s < P.n_maps; ++s) {
        wide |= (P.maps[s].unroll == -4);
        f16 |= (P.maps[s].esize == 2);
    }
    if (mode == 0 && f16)
        hipLaunchKernelGGL((fused_eval_f16_kernel<0>), grid, block, lds, stream, P);
    else if (mode == 0 && wide)
        hipLaunchKernelGGL((fused_eval_wide_kernel<0>), grid, block, lds, stream, P);
    else if (mode == 0)
        hipLaunchKernelGGL((fused_eval_kernel<0>), grid, block, lds, stream, P);
    else
        hipLaunchKernelGGL((fused_eval_kernel<1>), grid, block, lds, stream, P);
    return hipGetLastError();
}

}  // namespace d3f
