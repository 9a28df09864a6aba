// ode_dopri5.hip -- the adaptive integrator of the reference's DEFAULT sampler, resident on the device (gfx950).
//
// The reference samples with Sampler.sample_ode(sampling_method="dopri5", atol=1e-6, rtol=1e-3)
// (/root/reference/transport/transport.py:384-431, nsr/lsgm/flow_matching_trainer.py:715) through torchdiffeq.odeint
// (/root/reference/transport/integrators.py:111-118; third party, unpinned: semantics restated in SURVEY.md A.3 and
// oracle/ode.py): Dormand-Prince 5(4) with FSAL, error ratio = RMS over the WHOLE state of err / (atol + rtol max(|y0|, |y1|)),
// step factor clip(0.9 ratio^(-1/5), 0.2 (1 if accepted), 10), 4th-order dense output at the requested times.
//
// Round 3 ran this as a host loop (one float() synchronisation per attempted step, ~290 eager launches per evaluation, ~250 x 8
// interpolation launches): 3.9 ms per function evaluation against 3.2 in the replayed Euler loop.  Here one ATTEMPTED STEP is a fixed
// sequence of launches -- six (stage input, function evaluation) pairs, the error norm, a one-thread controller, a predicated
// accept kernel -- that reads and writes time, step size, decisions and counters in a small device block, so the host captures it
// into a HIP graph once and replays it; it only looks at the `done` word after a replay.
//   ga_ode_dopri5_stage    ystage = y + dt sum_j B[i][j] k_j ; timesteps[] = t + A[i] dt           (i = 0..5; i = 5 gives y1)
//   ga_ode_dopri5_error    ctl.partial[block] = sum over the block's elements of ((dt sum_j Cerr[j] k_j) / (atol + rtol max(|y|, |y1|)))^2
//                          (fixed element -> thread -> wave -> block assignment; the controller adds the partials in a fixed order:
//                          the accept / reject decision is bit-reproducible from run to run and from replay to replay)
//   ga_ode_dopri5_control  ratio -> accept / reject, next dt, the grid times inside an accepted step, done
//   ga_ode_dopri5_accept   (accepted steps only) dense output at those grid times, y <- y1, k1 <- k7
// The arithmetic follows gaussiananything_amd/transport/odeint.py operation by operation (time and step size in fp64, the
// coefficient dt * c rounded to fp32, products and sums of the state update rounded separately).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/ga_dit.h"

namespace gaode {

__constant__ double kA[6] = {1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
__constant__ double kB[6][6] = {{1.0 / 5, 0, 0, 0, 0, 0},
                                {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
                                {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
                                {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
                                {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
                                {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
__constant__ double kCerr[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720,
                                -2187.0 / 6784 - -12231.0 / 42400, 11.0 / 84 - 649.0 / 6300, -1.0 / 60.0};
__constant__ double kCmid[7] = {6025192743.0 / 30085553152.0 / 2, 0, 51252292925.0 / 65400821598.0 / 2, -2691868925.0 / 45128329728.0 / 2,
                                187940372067.0 / 1594534317056.0 / 2, -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};

// separately rounded product and sum (this TU may contract to FMA; the eager loop's tensor ops do not)
__device__ __forceinline__ float mul_rn(float a, float b) { float p = a * b; asm volatile("" : "+v"(p)); return p; }
__device__ __forceinline__ float axpy(float y, float c, float k) { return y + mul_rn(c, k); }

struct KPtrs { const float *k[7]; };

__global__ __launch_bounds__(256) void stage_kernel(int64_t n, int stage, const float *__restrict__ y, KPtrs kp, float *__restrict__ ystage,
                                                    const double *__restrict__ ctl, float *__restrict__ timesteps, int batch)
{
    const double t = ctl[GA_ODE_T], dt = ctl[GA_ODE_DT];
    if (blockIdx.x == 0 && (int)threadIdx.x < batch) timesteps[threadIdx.x] = (float)(t + kA[stage] * dt);
    float c[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) c[j] = (float)(dt * kB[stage][j]);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float v = y[e];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j <= stage && kB[stage][j] != 0.0) v = axpy(v, c[j], kp.k[j][e]);
        ystage[e] = v;
    }
}

__global__ __launch_bounds__(256) void error_kernel(int64_t n, const float *__restrict__ y, const float *__restrict__ y1, KPtrs kp,
                                                    double *__restrict__ ctl)
{
    const double dt = ctl[GA_ODE_DT];
    const float atol = (float)ctl[GA_ODE_ATOL], rtol = (float)ctl[GA_ODE_RTOL];
    float c[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) c[j] = (float)(dt * kCerr[j]);
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float err = 0.0f;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (kCerr[j] != 0.0) err = axpy(err, c[j], kp.k[j][e]);
        const float tol = atol + mul_rn(rtol, fmaxf(fabsf(y[e]), fabsf(y1[e])));
        const float r = err / tol;
        acc += (double)mul_rn(r, r);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ctl[GA_ODE_CTL_WORDS + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);    // (no atomics: order-free)
}

// one thread: torchdiffeq's controller (rk_common._optimal_step_size) and the bookkeeping of the attempted step
__global__ __launch_bounds__(64) void control_kernel(int64_t n, int nparts, double *__restrict__ ctl, const double *__restrict__ t_grid, int ngrid)
{
    // the error norm: lane l adds the partials l, l + 64, ... in that order, then a fixed shuffle tree
    double sumsq = 0.0;
    for (int b = (int)threadIdx.x; b < nparts; b += 64) sumsq += ctl[GA_ODE_CTL_WORDS + b];
    for (int o = 32; o > 0; o >>= 1) sumsq += __shfl_down(sumsq, o, 64);
    if (threadIdx.x != 0) return;
    ctl[GA_ODE_ACCEPT] = 0.0;
    ctl[GA_ODE_JCOUNT] = 0.0;
    if (ctl[GA_ODE_DONE] != 0.0) return;                 // (a replay past the end: nothing moves, no counter either)
    const double t = ctl[GA_ODE_T], dt = ctl[GA_ODE_DT];
    const double ratio = sqrt(sumsq / (double)n);
    ctl[GA_ODE_SUMSQ] = sumsq;                           // (diagnostic: the sum the decision was taken on)
    ctl[GA_ODE_RATIO] = ratio;
    ctl[GA_ODE_STEPS] += 1.0;
    if (!(ratio == ratio) || isinf(ratio) || !(dt == dt) || isinf(dt) || t + dt == t) {   // NaN / inf model output, step size underflow
        ctl[GA_ODE_ERROR] = !(ratio == ratio) || isinf(ratio) ? 1.0 : 2.0;
        ctl[GA_ODE_DONE] = 1.0;
        return;
    }
    const bool accept = ratio <= 1.0;
    double factor;
    if (ratio == 0.0) factor = 10.0;
    else factor = fmin(10.0, fmax(0.9 / pow(ratio, 0.2), ratio < 1.0 ? 1.0 : 0.2));
    ctl[GA_ODE_DT_USED] = dt;
    ctl[GA_ODE_DT] = dt * factor;
    if (!accept) { ctl[GA_ODE_REJECTED] += 1.0; return; }
    const double tb = t + dt;
    ctl[GA_ODE_ACCEPT] = 1.0;
    ctl[GA_ODE_TA] = t;
    ctl[GA_ODE_TB] = tb;
    ctl[GA_ODE_T] = tb;
    // the requested times inside (t, tb]: interpolated by the accept kernel
    int j0 = (int)ctl[GA_ODE_JNEXT], j1 = j0;
    while (j1 < ngrid && !(t_grid[j1] > tb)) ++j1;
    ctl[GA_ODE_JBEG] = (double)j0;
    ctl[GA_ODE_JCOUNT] = (double)(j1 - j0);
    ctl[GA_ODE_JNEXT] = (double)j1;
    if (j1 >= ngrid) ctl[GA_ODE_DONE] = 1.0;
}

__global__ __launch_bounds__(256) void accept_kernel(int64_t n, float *__restrict__ y, const float *__restrict__ y1, float *__restrict__ k0,
                                                     KPtrs kp, const double *__restrict__ ctl, const double *__restrict__ t_grid,
                                                     float *__restrict__ out)
{
    if (ctl[GA_ODE_ACCEPT] == 0.0) return;
    const double dt = ctl[GA_ODE_DT_USED], ta = ctl[GA_ODE_TA], tb = ctl[GA_ODE_TB];
    const int j0 = (int)ctl[GA_ODE_JBEG], jc = (int)ctl[GA_ODE_JCOUNT];
    const float dtf = (float)dt;
    float cm[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cm[j] = (float)(dt * kCmid[j]);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float y0 = y[e], yn = y1[e], f0 = kp.k[0][e], f1 = kp.k[6][e];
        if (jc > 0) {
            float mid = 0.0f;     // ymid = y0 + sum (dt c) k   (the eager loop sums the increments first)
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (kCmid[j] != 0.0) mid = axpy(mid, cm[j], kp.k[j][e]);
            const float ym = y0 + mid;
            // torchdiffeq's quartic through (y0, f0), ymid, (y1, f1)
            const float a = 2.0f * dtf * (f1 - f0) - 8.0f * (yn + y0) + 16.0f * ym;
            const float b = dtf * (5.0f * f0 - 3.0f * f1) + 18.0f * y0 + 14.0f * yn - 32.0f * ym;
            const float c = dtf * (f1 - 4.0f * f0) - 11.0f * y0 - 5.0f * yn + 16.0f * ym;
            const float d = dtf * f0;
            for (int j = 0; j < jc; ++j) {
                const float x = (float)((t_grid[j0 + j] - ta) / (tb - ta));
                out[(size_t)(j0 + j) * n + e] = y0 + x * (d + x * (c + x * (b + x * a)));
            }
        }
        y[e] = yn;
        k0[e] = f1;
    }
}

inline int grid_for(int64_t n) { return (int)((n + 255) / 256 < GA_ODE_MAX_PARTIALS ? (n + 255) / 256 : GA_ODE_MAX_PARTIALS); }

}  // namespace gaode

extern "C" {

int ga_ode_dopri5_stage(const GaOdeDopri5 *o, int32_t stage, void *stream)
{
    using namespace gaode;
    if (!o || !o->y || !o->ystage || !o->ctl || !o->timesteps) return GA_DIT_ERR_NULL_ARG;
    if (stage < 0 || stage > 5 || o->n <= 0 || o->batch <= 0 || o->batch > 64) return GA_DIT_ERR_BAD_SHAPE;
    KPtrs kp;
    for (int j = 0; j < 7; ++j) { if (!o->k[j]) return GA_DIT_ERR_NULL_ARG; kp.k[j] = o->k[j]; }
    hipLaunchKernelGGL(stage_kernel, dim3(grid_for(o->n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), o->n, stage, o->y, kp,
                       o->ystage, o->ctl, o->timesteps, o->batch);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

int ga_ode_dopri5_finish(const GaOdeDopri5 *o, void *stream)
{
    using namespace gaode;
    if (!o || !o->y || !o->ystage || !o->ctl || !o->t_grid || !o->out) return GA_DIT_ERR_NULL_ARG;
    if (o->n <= 0 || o->grid_len <= 0) return GA_DIT_ERR_BAD_SHAPE;
    if (o->ctl_words < (int64_t)GA_ODE_CTL_WORDS + grid_for(o->n)) return GA_DIT_ERR_BAD_SHAPE;   // the partials would land outside `ctl`
    KPtrs kp;
    for (int j = 0; j < 7; ++j) { if (!o->k[j]) return GA_DIT_ERR_NULL_ARG; kp.k[j] = o->k[j]; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(error_kernel, dim3(grid_for(o->n)), dim3(256), 0, s, o->n, o->y, o->ystage, kp, o->ctl);
    hipLaunchKernelGGL(control_kernel, dim3(1), dim3(64), 0, s, o->n, grid_for(o->n), o->ctl, o->t_grid, o->grid_len);
    hipLaunchKernelGGL(accept_kernel, dim3(grid_for(o->n)), dim3(256), 0, s, o->n, o->y, o->ystage, o->k[0], kp, o->ctl, o->t_grid, o->out);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

}  // extern "C"
