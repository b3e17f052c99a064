// vaeq_dp_wave.hip -- dispatch of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h): one wavefront per run for B <= 128 (M = 25: B = 100 and 128 baked,
// every other B -- even or odd -- on the fixed layout of B = 128, like all B of the other tap counts; the run-time layout only as M = 25's A/B form); the
// multi-wave variants for 128 < B <= 1024 are instantiated in vaeq_dp_wave_mw.hip / vaeq_dp_wave_mw8.hip.
#include <stdlib.h>

#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

int launch_dp_wave_mw(const vaeq_dp_args &a, hipStream_t st);          // vaeq_dp_wave_mw.hip
int64_t dp_wave_mw_resident(int B, int M, int n_lev);
int launch_dp_wave_nw(const vaeq_dp_args &a, hipStream_t st, int nw);   // vaeq_dp_wave_mw.hip: any B on nw wavefronts per run (run-time layout)
bool dp_wave_fixl(int B, int M);                                        // vaeq_dp_wave_bk.hip (false under VAEQ_DP_RUNTIME_LAYOUT=1: A/B switch)
int launch_dp_wave_bk(const vaeq_dp_args &a, hipStream_t st);
int64_t dp_wave_bk_resident(int n_lev);
int launch_dp_wave_b128(const vaeq_dp_args &a, hipStream_t st);         // vaeq_dp_wave_b128.hip
int64_t dp_wave_b128_resident(int n_lev);

// Whether the wave-per-run kernel covers this call (else the generic kernel runs).
bool dp_wave_supported(const vaeq_dp_args &a)
{
    // minibatches down to the shortest one with a residual (nm = 2 B - 2 (M / 2) >= 2 samples: the reference's short batch_len options, Eval_run_DP.py:38,
    // e.g. B = 20 with M = 25 -- the KL slice mh <= n < B - mh is then empty, as in the reference): ten of the 64 lanes own a symbol pair, still
    // several times the generic kernel's rate (one workgroup of 256 threads and seven barriers per step for 20 symbols)
    if (a.sps != 2 || a.B > 1024 || 2 * a.B - 2 * (a.M / 2) < 2 || a.B < 4) return false;
    if (!(a.M == 25 || a.M == 31 || a.M == 21 || a.M == 17 || a.M == 13 || a.M == 9)) return false;
    // odd minibatch lengths run on the class layouts (the last lane's pair is half empty, vaeq_dp_wave_kernel.h: ODDB), i.e. not on M = 25's
    // run-time-layout A/B form; their windows start 8-byte aligned (16-byte buffer loads need dword alignment only)
    if ((a.B & 1) && (a.B > 1023 || (a.M == 25 && !dp_wave_fixl(a.B, 25)))) return false;
    if ((a.S & 1) || (reinterpret_cast<uintptr_t>(a.rx) & 15)) return false;
    if (!(a.B & 1) && ((a.S & 3) || ((a.stride_sym * 2) & 3))) return false;                             // even B: 16-byte aligned window loads as before
    if (a.q_out && (reinterpret_cast<uintptr_t>(a.q_out) & 7)) return false;
    if (a.y_out && (reinterpret_cast<uintptr_t>(a.y_out) & 7)) return false;
    if (a.eq_out && (reinterpret_cast<uintptr_t>(a.eq_out) & 7)) return false;
    if (a.dec_out && (reinterpret_cast<uintptr_t>(a.dec_out) & 1)) return false;
    if ((a.dbg_gW == nullptr) != (a.dbg_gh == nullptr)) return false;
    return true;
}

int64_t dp_wave_resident(int B, int M, int n_lev)
{
    if (B > 128) return dp_wave_mw_resident(B, M, n_lev);
    if (M == 25 && B == 128 && dp_wave_fixl(64, 25)) return dp_wave_b128_resident(n_lev);
    if (dp_wave_fixl(B, M)) return dp_wave_bk_resident(n_lev);
    if (M == 25 && B == 100) return wave_resident_lev<25, 100, 1>(B, n_lev);
    return wave_resident_any<1>(B, M, n_lev);
}

int launch_dp_wave(const vaeq_dp_args &a, hipStream_t st)
{
    if (a.B > 128) return launch_dp_wave_mw(a, st);
    if (const char *e = getenv("VAEQ_DP_FORCE_NW")) {          // experiment knob: B <= 128 on two / four wavefronts per run (run-time layout)
        if ((e[0] == '2' || e[0] == '4') && !(a.B & 1)) return launch_dp_wave_nw(a, st, e[0] - '0');   // (run-time layout: even B only)
    }
    if (a.M == 25 && a.B == 128 && dp_wave_fixl(64, 25)) return launch_dp_wave_b128(a, st);
    if (dp_wave_fixl(a.B, a.M)) return launch_dp_wave_bk(a, st);
    if (a.M == 25 && a.B == 100) return launch_wave_lev<25, 100, 1>(a, st);
    return launch_wave_any<1>(a, st);
}

}  // namespace vaeq
