// fuse_launch.hip -- host side of the fused field query: the planner's choice (EvalParams, d3f_plan.h) -> the family's launcher.
#include <hip/hip_runtime.h>

#include "d3f_internal.h"

namespace d3f {

hipError_t launch_fused_eval(const EvalParams &P, int mode, hipStream_t stream)
{
    if (P.n == 0) return hipSuccess;
    bool wide = false, f16 = false, runs = false;
    for (int s = 0; s < P.n_maps; ++s) {
        wide |= (P.maps[s].unroll == -4);
        f16 |= (P.maps[s].esize == 2);
        runs |= (P.maps[s].runs > 0);
    }
    if (mode == 0 && P.rows > 0) return launch_rows(P, stream);
    if (mode == 0 && P.win_slices > 0) return launch_window(P, stream);
    if (mode == 0 && P.sl_slices > 0) return launch_sliced(P, stream);
    if (mode == 0 && runs && !f16 && !wide) return launch_runs(P, stream);
    return launch_direct(P, mode, stream);
}

}  // namespace d3f
