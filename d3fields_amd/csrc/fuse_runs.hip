// fuse_runs.hip -- the CELL-RUN kernels of the fused field query (gfx950): patch-resolution wide fp32 maps on clouds (and the
// other side of the device gate).  The gather itself is gather_map_runs in fuse_body.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3f_internal.h"
#include "d3f_device.h"
#include "fuse_common.h"
#include "fuse_body.h"

namespace d3f {

// cell-run gather for patch-resolution wide maps, one entry point per (vectors per lane, run length, waves per SIMD) so
// that every variant gets its own register allocation.  The planner's choices (launch_fused_eval): <1,4,7> for 32-lane
// groups (C = 384: C2 patch clouds 0.750 -> 0.633 ms), <2,8,3> for 64-lane groups x two vectors (C = 1024: C4 patch
// 4.32 -> 3.31 ms; held to 4 waves it spills 48 bytes per lane), <1,8,5> otherwise; all three are spill-free.  The other
// instantiations (some spill) are compiled into experiments builds only.
template <int MODE, int RU, int RK, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void fused_eval_runs_kernel(const EvalParams P)
{
    if (gated_out(P)) return;
    fused_eval_body<MODE, false, false, RU, RK>(P);
}

hipError_t launch_runs(const EvalParams &P, hipStream_t stream)
{
    const int64_t ntiles = (P.n + P.tile_pts - 1) / P.tile_pts;
    const size_t lds = (size_t)P.crec_offset + (size_t)P.n_pre * P.tile_pts * P.V * 32 + (size_t)P.lds_pad;
    dim3 grid((unsigned)ntiles), block(kBlock);
    int ru = 1, rk = 8;
    for (int s = 0; s < P.n_maps; ++s)
        if (P.maps[s].runs > 0) { ru = P.maps[s].unroll; rk = P.maps[s].runs; }
    // product library: the three spill-free variants the planner picks by itself -- (2,8) at 3 waves per SIMD (64-lane
    // groups x 2 vectors, C = 1024), (1,4) at 7 waves (32-lane groups, C = 384), (1,8) at 5 waves (64-lane groups x 1)
    if (ru == 2 && rk == 8 && P.runs_occ != 4) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 2, 8, 3>), grid, block, lds, stream, P);
    else if (ru == 1 && rk == 4 && P.runs_occ != 6) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 1, 4, 7>), grid, block, lds, stream, P);
    else if (ru == 1 && rk == 8 && P.runs_occ != 4 && P.runs_occ != 6) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 1, 8, 5>), grid, block, lds, stream, P);
#ifdef D3F_EXPERIMENTS
    else if (ru == 3 && rk == 4) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 3, 4, 4>), grid, block, lds, stream, P);
    else if (ru == 3 && rk == 2) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 3, 2, 4>), grid, block, lds, stream, P);
    else if (ru == 2 && rk == 4) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 2, 4, 4>), grid, block, lds, stream, P);
    else if (ru == 2 && rk == 8) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 2, 8, 4>), grid, block, lds, stream, P);
    else if (ru == 1 && rk == 4) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 1, 4, 6>), grid, block, lds, stream, P);
    else if (P.runs_occ == 4) hipLaunchKernelGGL((fused_eval_runs_kernel<0, 1, 8, 4>), grid, block, lds, stream, P);
    else hipLaunchKernelGGL((fused_eval_runs_kernel<0, 1, 8, 6>), grid, block, lds, stream, P);
#else
    else return hipErrorInvalidValue;
#endif
    return hipGetLastError();
}

}  // namespace d3f
