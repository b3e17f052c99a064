// vaeq_dp_wave_fl.hip -- M = 25, multi-wave shapes of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h) on FIXED LDS layouts (template parameter BL):
// two wavefronts per run on the layout of B = 256 for 128 < B <= 256, four on that of 512 for B <= 512, eight on that of 1024 for B <= 1024 (B = 200 / 400
// stay baked, vaeq_dp_wave_mw.hip).  Offsets and row strides are immediate, the tap loops pipelined; B itself stays a run-time value.
// The layouts fit the residency the register file allows anyway (2 waves per SIMD): 4 x 35 KB, 2 x 66 KB, 1 x 130 KB per CU.
#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

template <int BL, int NW>
static int launch_fl(const vaeq_dp_args &a, hipStream_t st)
{
    switch (a.n_lev) {
    case 2: return launch_wave_fixl<25, 2, BL, NW>(a, st);
    case 4: return launch_wave_fixl<25, 4, BL, NW>(a, st);
    case 8: return launch_wave_fixl<25, 8, BL, NW>(a, st);
    }
    return VAEQ_ERR_SHAPE;
}
template <int BL, int NW>
static int64_t resident_fl(int n_lev)
{
    switch (n_lev) {
    case 2: return wave_resident_fixl<25, 2, BL, NW>();
    case 4: return wave_resident_fixl<25, 4, BL, NW>();
    case 8: return wave_resident_fixl<25, 8, BL, NW>();
    }
    return VAEQ_ERR_SHAPE;
}

int launch_dp_wave_fl(const vaeq_dp_args &a, hipStream_t st)
{
    return a.B <= 256 ? launch_fl<256, 2>(a, st) : a.B <= 512 ? launch_fl<512, 4>(a, st) : launch_fl<1024, 8>(a, st);
}
int64_t dp_wave_fl_resident(int B, int n_lev) { return B <= 256 ? resident_fl<256, 2>(n_lev) : B <= 512 ? resident_fl<512, 4>(n_lev) : resident_fl<1024, 8>(n_lev); }

}  // namespace vaeq
