// vaeq_dp_wave_b128.hip -- M = 25, B = 128 baked like B = 100 (vaeq_dp_wave_kernel.h): the single-wave shape in which every lane owns a symbol pair,
// 5 % faster per step than the same minibatch on the fixed-layout instantiation of vaeq_dp_wave_bk.hip.  A translation unit of its own.
#include "vaeq_dp_wave_kernel.h"

namespace vaeq {

int launch_dp_wave_b128(const vaeq_dp_args &a, hipStream_t st) { return launch_wave_lev<25, 128, 1>(a, st); }
int64_t dp_wave_b128_resident(int n_lev) { return wave_resident_lev<25, 128, 1>(128, n_lev); }

}  // namespace vaeq
