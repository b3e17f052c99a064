// vaeq_dp_wave_mw8.hip -- the eight-wave instantiations of the wave-per-run DP kernel (vaeq_dp_wave_kernel.h): 512 < B <= 1024.
#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

int launch_dp_wave_mw8(const vaeq_dp_args &a, hipStream_t st) { return launch_wave_any<8>(a, st); }

int64_t dp_wave_mw8_resident(int B, int M, int n_lev) { return wave_resident_any<8>(B, M, n_lev); }

}  // namespace vaeq
