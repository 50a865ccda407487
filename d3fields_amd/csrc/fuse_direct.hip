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

hipError_t launch_direct(const EvalParams &P, int mode, hipStream_t stream)
{
    int64_t ntiles = (P.n + P.tile_pts - 1) / P.tile_pts;
    if (P.walk_nx > 0)
        ntiles = (int64_t)((P.walk_nx + P.walk_tx - 1) / P.walk_tx) * ((P.walk_ny + P.walk_ty - 1) / P.walk_ty) *
                 ((P.walk_nz + P.walk_tz - 1) / P.walk_tz);
    const size_t lds = (size_t)P.crec_offset + (size_t)P.n_pre * P.tile_pts * P.V * 32 + (size_t)P.lds_pad;
    dim3 grid((unsigned)ntiles), block(kBlock);
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
