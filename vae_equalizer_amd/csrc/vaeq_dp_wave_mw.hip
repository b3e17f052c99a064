// vaeq_dp_wave_mw.hip -- the multi-wave instantiations of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h): two wavefronts per
// run for 128 < B <= 256, four for 256 < B <= 512.  A translation unit of its own so that the two compile side by side.
#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

template <int NW>
static int launch_mw(const vaeq_dp_args &a, hipStream_t st)
{
    switch (a.M) {
    case 25: return launch_wave_lev<25, 0, NW>(a, st);
    case 31: return launch_wave_lev<31, 0, NW>(a, st);
    case 21: return launch_wave_lev<21, 0, NW>(a, st);
    case 17: return launch_wave_lev<17, 0, NW>(a, st);
    case 13: return launch_wave_lev<13, 0, NW>(a, st);
    case 9: return launch_wave_lev<9, 0, NW>(a, st);
    }
    return VAEQ_ERR_SHAPE;
}

template <int NW>
static int64_t resident_mw(int B, int M, int n_lev)
{
    switch (M) {
    case 25: return wave_resident_lev<25, 0, NW>(B, n_lev);
    case 31: return wave_resident_lev<31, 0, NW>(B, n_lev);
    case 21: return wave_resident_lev<21, 0, NW>(B, n_lev);
    case 17: return wave_resident_lev<17, 0, NW>(B, n_lev);
    case 13: return wave_resident_lev<13, 0, NW>(B, n_lev);
    case 9: return wave_resident_lev<9, 0, NW>(B, n_lev);
    }
    return VAEQ_ERR_SHAPE;
}

int launch_dp_wave_mw(const vaeq_dp_args &a, hipStream_t st) { return a.B <= 256 ? launch_mw<2>(a, st) : launch_mw<4>(a, st); }

int64_t dp_wave_mw_resident(int B, int M, int n_lev) { return B <= 256 ? resident_mw<2>(B, M, n_lev) : resident_mw<4>(B, M, n_lev); }

}  // namespace vaeq
