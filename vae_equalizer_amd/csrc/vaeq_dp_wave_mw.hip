// vaeq_dp_wave_mw.hip -- the two- and four-wave instantiations of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h): two wavefronts
// per run for 128 < B <= 256, four for 256 < B <= 512.  Translation units of their own so that they compile side by side.
#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

int launch_dp_wave_mw8(const vaeq_dp_args &a, hipStream_t st);         // vaeq_dp_wave_mw8.hip
int64_t dp_wave_mw8_resident(int B, int M, int n_lev);
int launch_dp_wave_fl(const vaeq_dp_args &a, hipStream_t st);          // vaeq_dp_wave_fl.hip: M = 25 on fixed layouts
int64_t dp_wave_fl_resident(int B, int n_lev);
bool dp_wave_fixl(int B, int M);                                       // vaeq_dp_wave_bk.hip

// M = 25 with B = 200 / 400 (the reference's longer minibatch sweeps, Eval_run_DP.py batch_len_vec) are baked like B = 100: immediate LDS offsets, scalar
// trip counts, pipelined tap loops -- the run-time-shape instantiation spills at these sizes
int launch_dp_wave_mw(const vaeq_dp_args &a, hipStream_t st)
{
    if (a.M == 25 && a.B == 200) return launch_wave_lev<25, 200, 2>(a, st);
    if (a.M == 25 && a.B == 400) return launch_wave_lev<25, 400, 4>(a, st);
    if (dp_wave_fixl(a.B, a.M)) return launch_dp_wave_fl(a, st);
    return a.B <= 256 ? launch_wave_any<2>(a, st) : a.B <= 512 ? launch_wave_any<4>(a, st) : launch_dp_wave_mw8(a, st);
}

int launch_dp_wave_nw(const vaeq_dp_args &a, hipStream_t st, int nw) { return nw == 2 ? launch_wave_any<2>(a, st) : launch_wave_any<4>(a, st); }

int64_t dp_wave_mw_resident(int B, int M, int n_lev)
{
    if (M == 25 && B == 200) return wave_resident_lev<25, 200, 2>(B, n_lev);
    if (M == 25 && B == 400) return wave_resident_lev<25, 400, 4>(B, n_lev);
    if (dp_wave_fixl(B, M)) return dp_wave_fl_resident(B, n_lev);
    return B <= 256 ? wave_resident_any<2>(B, M, n_lev) : B <= 512 ? wave_resident_any<4>(B, M, n_lev) : dp_wave_mw8_resident(B, M, n_lev);
}

}  // namespace vaeq
