// vaeq_dp_wave_bk.hip -- single-wave shapes of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h) beyond the baked B = 100, for M = 25:
// any even B <= 128 runs on the FIXED LDS layout of B = 128 (template parameter BL: offsets and row strides immediate, the pipelined tap
// loops and 16 accumulator chains of the baked shapes; B itself -- masks, trip counts -- stays a run-time value).  The fully run-time
// instantiation (layout from B) is 16-18 % slower per step: 250 SGPR + 18 VGPR spills (5.16 vs 4.46 us per step at B = 98 vs 100).
// A translation unit of its own so that it compiles beside the others.
#include <stdlib.h>

#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

// VAEQ_DP_RUNTIME_LAYOUT=1: the fully run-time instantiations for every shape that is not B = 100 / 200 / 400 (A/B switch)
bool dp_wave_fixl(int B, int M)
{
    const char *e = getenv("VAEQ_DP_RUNTIME_LAYOUT");
    return M == 25 && B != 100 && !(e && e[0] == '1');
}

int launch_dp_wave_bk(const vaeq_dp_args &a, hipStream_t st)
{
    switch (a.n_lev) {
    case 2: return launch_wave_fixl<25, 2, 128, 1>(a, st);
    case 4: return launch_wave_fixl<25, 4, 128, 1>(a, st);
    case 8: return launch_wave_fixl<25, 8, 128, 1>(a, st);
    }
    return VAEQ_ERR_SHAPE;
}

int64_t dp_wave_bk_resident(int n_lev)
{
    switch (n_lev) {
    case 2: return wave_resident_fixl<25, 2, 128, 1>();
    case 4: return wave_resident_fixl<25, 4, 128, 1>();
    case 8: return wave_resident_fixl<25, 8, 128, 1>();
    }
    return VAEQ_ERR_SHAPE;
}

}  // namespace vaeq
