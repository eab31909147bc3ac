// mpcqp_plant.h -- the wheeled inverted pendulum's control period on the device, shared by the stand-alone plant kernel
// (mpcqp_model.hip: mpcqp_wip_advance_stats_batch) and by the epilogue of the stage-wise solver kernel
// (mpcqp_stage.hip: mpcqp_wip_period_batch, the whole period in ONE launch).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace mpcqp {

// sin and cos of a pendulum angle: on |x| <= 0.5 (where an upright pendulum lives) the Taylor series to x^17 / x^16 in
// Horner form (truncation < 2e-23, i.e. below one ulp; 18 FMAs instead of the library's ~100 instructions), the library
// beyond
__device__ __forceinline__ void sincos_t(double x, double *s, double *c)
{
    if (fabs(x) <= 0.5) {
        const double z = x * x;
        double ps = 1.0 / 355687428096000.0;  // 1/17!
        ps = fma(ps, z, -1.0 / 1307674368000.0);
        ps = fma(ps, z, 1.0 / 6227020800.0);
        ps = fma(ps, z, -1.0 / 39916800.0);
        ps = fma(ps, z, 1.0 / 362880.0);
        ps = fma(ps, z, -1.0 / 5040.0);
        ps = fma(ps, z, 1.0 / 120.0);
        ps = fma(ps, z, -1.0 / 6.0);
        *s = fma(x * z, ps, x);
        double pc = 1.0 / 20922789888000.0;  // 1/16!
        pc = fma(pc, z, -1.0 / 87178291200.0);
        pc = fma(pc, z, 1.0 / 479001600.0);
        pc = fma(pc, z, -1.0 / 3628800.0);
        pc = fma(pc, z, 1.0 / 40320.0);
        pc = fma(pc, z, -1.0 / 720.0);
        pc = fma(pc, z, 1.0 / 24.0);
        pc = fma(pc, z, -0.5);
        *c = fma(z, pc, 1.0);
    } else {
        sincos(x, s, c);
    }
}
__device__ __forceinline__ void sincos_t(float x, float *s, float *c) { sincosf(x, s, c); }

// One control period of ONE loop, executed by a whole wavefront (every lane integrates: same cost as one lane; the lanes
// then write the N reference rows side by side): the input a is applied to the second-order Taylor plant for nsub
// sub-steps (qpmpc/systems/wheeled_inverted_pendulum.py:127-160), then the next MPC problem's x0 [4], goal [4] and
// targets [N * 4] are written (examples/wheeled_inverted_pendulum.py:65-83,101-108). st / x0 / goal / tg: this loop's.
// (s0: the loop's state as it stands in st -- the fused period requests it at the top of the solver kernel, so that its
// round trip is not on the period's tail)
template <typename T>
__device__ __forceinline__ void wip_period_wave(int lane, T *st, const T (&s0)[4], T a, int N, T Tp, T vel, T omega2, T g, int nsub,
                                                T *x0, T *goal, T *tg, T (&s1)[4])  // s1: the state after the period, in every lane
{
    T r = s0[0], th = s0[1], rd = s0[2], thd = s0[3];
    const T dt = nsub > 0 ? Tp / (T)nsub : T(0), ag = a / g;  // (nsub = 0: the state stays, the problem of that state is written)
    for (int i = 0; i < nsub; ++i) {
        T sn, cs;
        sincos_t(th, &sn, &cs);  // (one argument reduction for both)
        const T thdd = omega2 * (sn - ag * cs);
        const T r2 = r + dt * (rd + dt * (a / 2));
        const T th2 = th + dt * (thd + dt * (thdd / 2));
        rd = rd + dt * a;
        thd = thd + dt * thdd;
        r = r2;
        th = th2;
    }
    s1[0] = r;
    s1[1] = th;
    s1[2] = rd;
    s1[3] = thd;
    __builtin_amdgcn_wave_barrier();  // (every lane has read the state before any lane overwrites it)
    if (lane < 4) {
        const T v = lane == 0 ? r : lane == 1 ? th : lane == 2 ? rd : thd;
        st[lane] = v;
        x0[lane] = v;
        goal[lane] = lane == 0 ? r + ((T)N * Tp) * vel : lane == 2 ? vel : T(0);
    }
    for (int k = lane; k < N; k += 64) {
        tg[k * 4 + 0] = r + ((T)k * Tp) * vel;
        tg[k * 4 + 1] = T(0);
        tg[k * 4 + 2] = vel;
        tg[k * 4 + 3] = T(0);
    }
}
template <typename T>
__device__ __forceinline__ void wip_period_wave(int lane, T *st, T a, int N, T Tp, T vel, T omega2, T g, int nsub, T *x0, T *goal,
                                                T *tg)
{
    const T s0[4] = {st[0], st[1], st[2], st[3]};
    T s1[4];
    wip_period_wave<T>(lane, st, s0, a, N, Tp, vel, omega2, g, nsub, x0, goal, tg, s1);
}

}  // namespace mpcqp
