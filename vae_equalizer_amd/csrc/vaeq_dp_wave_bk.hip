// vaeq_dp_wave_bk.hip -- further baked single-wave shapes of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h): M = 25 with B = 64 and B = 128
// (every lane of the wavefront owns a symbol pair at 128).  Like B = 100 they get immediate LDS offsets, scalar trip counts and the pipelined tap
// loops; the run-time-shape instantiation is 16-18 % slower per step (250 SGPR + 18 VGPR spills; 5.16 vs 4.46 us per step at B = 98 vs 100).
// A translation unit of its own so that it compiles beside the others.
#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

bool dp_wave_baked(int B, int M) { return M == 25 && (B == 64 || B == 128); }

int launch_dp_wave_bk(const vaeq_dp_args &a, hipStream_t st)
{
    return a.B == 64 ? launch_wave_lev<25, 64, 1>(a, st) : launch_wave_lev<25, 128, 1>(a, st);
}

int64_t dp_wave_bk_resident(int B, int n_lev)
{
    return B == 64 ? wave_resident_lev<25, 64, 1>(B, n_lev) : wave_resident_lev<25, 128, 1>(B, n_lev);
}

}  // namespace vaeq
