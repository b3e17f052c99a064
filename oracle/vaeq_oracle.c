/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see vaeq_oracle_impl.h for the full header).
 * Instantiates the CPU restatement for float (_f32) and double (_f64) and adds
 * an OpenMP batch driver used by bench.py's cpu_baseline leg (kind "port").
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL float
#define REAL_IS_FLOAT 1
#define SFX _f32
#include "vaeq_oracle_impl.h"
#undef REAL
#undef REAL_IS_FLOAT
#undef SFX

#define REAL double
#define REAL_IS_FLOAT 0
#define SFX _f64
#include "vaeq_oracle_impl.h"
#undef REAL
#undef REAL_IS_FLOAT
#undef SFX

/* R independent runs (one sweep point / seed each, Eval_run_DP.py:68-86) of one frame,
 * fp32, spread over host threads.  Layouts are the per-run layouts of
 * vaeq_oracle_dp_train_f32 with a leading [R] axis; amp shared, P/var/nu_sc/lr per run.
 * Returns the number of threads used. */
int vaeq_oracle_dp_train_batch_f32(int R, int n_threads, int n_steps, int B, int sps, int M, int n, int stride,
                                   int keep_off, int keep_len, int S, const float *rx, float *W, float *h, float *mW,
                                   float *vW, float *mh_, float *vh, int *step, const float *amp, const float *P,
                                   const float *var, const float *nu_sc, const float *lr_W, const float *lr_h,
                                   float *q_out, float *y_out, float *loss, float *var_est)
{
    const size_t No = (size_t)n_steps * keep_len;
    int used = 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
    used = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int r = 0; r < R; r++) {
        vaeq_oracle_dp_train_f32(n_steps, B, sps, M, n, stride, keep_off, keep_len, S, rx + (size_t)r * 4 * S,
                                 W + (size_t)r * 8 * M, h + (size_t)r * 8 * M, mW + (size_t)r * 8 * M,
                                 vW + (size_t)r * 8 * M, mh_ + (size_t)r * 8 * M, vh + (size_t)r * 8 * M, step + r, amp,
                                 P + (size_t)r * n, var + (size_t)r * 2, nu_sc[r], (double)lr_W[r], (double)lr_h[r],
                                 q_out ? q_out + (size_t)r * 4 * n * No : NULL, y_out ? y_out + (size_t)r * 4 * No : NULL,
                                 loss + (size_t)r * n_steps, var_est + (size_t)r * 2 * n_steps);
    }
    return used;
}
