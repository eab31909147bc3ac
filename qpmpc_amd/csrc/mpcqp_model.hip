// mpcqp_model.hip -- the shared-model path: factor once, re-solve for new states.
//
// Replaces, for batches whose problems differ only by x0 / goal / targets, the
// reference's build-once usage: MPCQP(problem) once (qpmpc/mpc_qp.py:39-122), then
// update_cost_vector / update_constraint_vector (mpc_qp.py:129-163) per problem.
//
// mpcqp_model_kernel (one workgroup, one-time work): Cholesky P = L L' in LDS, then the
// forward substitution  row <- row L^-T  for every row of
//     [ G ; Qx' ; Qg' ; Qt' ; I ]
// giving M = G L^-T, the maps Wx, Wg, Wt (L^-1 q = Wx x0 - Wg goal - Wt targets) and
// the rows of L^-T; plus e and Hx (h = e - Hx x0) and 1/|M_i|. The Q* columns are the q
// vectors of pseudo-problems with unit states (include/mpcqp.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpcqp.h"
#include "mpcqp_internal.h"
#include "mpcqp_plant.h"

namespace mpcqp {

template <typename T>
__global__ void __launch_bounds__(256) mpcqp_model_kernel(const KernelArgs ka, const ModelLayout ml,
                                                          const T *__restrict__ P, const T *__restrict__ G,
                                                          const T *__restrict__ qb, const T *__restrict__ hb,
                                                          T *__restrict__ model)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *Lm = (T *)smem_raw;
    const int n = ka.n, m = ka.m, nx = ka.nx, N = ka.N, nc = ml.nc, tid = threadIdx.x;
    const int ld = n | 1;
    T *inv = Lm + n * ld;
    T *flag = model + ml.total;  // 0 = ok, 1 = P not positive definite
    for (int i = tid; i < n * n; i += 256) Lm[(i / n) * ld + (i % n)] = P[i];
    __syncthreads();
    // left-looking Cholesky, every thread recomputes the pivot (one barrier per column)
    bool notpd = false;
    for (int j = 0; j < n; ++j) {
        T piv = Lm[j * ld + j];
        for (int k = 0; k < j; ++k) piv -= Lm[j * ld + k] * Lm[j * ld + k];
        if (!(piv > T(0))) {
            notpd = true;
            break;
        }
        const T rinv = T(1) / sqrt(piv);
        for (int i = j + 1 + tid; i < n; i += 256) {
            T v = Lm[i * ld + j];
            for (int k = 0; k < j; ++k) v -= Lm[i * ld + k] * Lm[j * ld + k];
            Lm[i * ld + j] = v * rinv;
        }
        if (tid == 0) inv[j] = rinv;
        __syncthreads();
    }
    if (tid == 0) *flag = notpd ? T(1) : T(0);
    if (notpd) return;
    // rows through L^-T; results are written (and re-read) in place in the model
    const int nT = N * nx;
    const int rows = m + 2 * nx + nT + nc;
    T *Mo = model + ml.off_M, *Lt = model + ml.off_LinvT;
    T *Wx = model + ml.off_Wx, *Wg = model + ml.off_Wg, *Wt = model + ml.off_Wt;
    for (int row = tid; row < rows; row += 256) {
        const T *src = nullptr;
        T sign = T(1);
        T *dst;
        int dstride, unit = -1;
        if (row < m) {
            src = G + (size_t)row * n;
            dst = Mo + (size_t)row * nc;
            dstride = 1;
        } else if (row < m + nx) {
            const int c = row - m;
            src = qb + (size_t)(1 + c) * n;
            dst = Wx + c;
            dstride = nx;
        } else if (row < m + 2 * nx) {
            const int c = row - m - nx;
            src = qb + (size_t)(1 + nx + c) * n;
            sign = T(-1);
            dst = Wg + c;
            dstride = nx;
        } else if (row < m + 2 * nx + nT) {
            const int c = row - m - 2 * nx;
            src = qb + (size_t)(1 + 2 * nx + c) * n;
            sign = T(-1);
            dst = Wt + c;
            dstride = nT;
        } else {
            unit = row - (m + 2 * nx + nT);
            dst = Lt + (size_t)unit * nc;
            dstride = 1;
        }
        T nn = T(0);
        for (int j = 0; j < nc; ++j) {
            T v;
            if (j < n) {
                v = src ? sign * src[j] : ((unit == j) ? T(1) : T(0));
                for (int k = 0; k < j; ++k) v -= dst[(size_t)k * dstride] * Lm[j * ld + k];
                v *= inv[j];
            } else {
                v = (unit == j) ? T(1) : T(0);  // padded unit variables
            }
            dst[(size_t)j * dstride] = v;
            nn += v * v;
        }
        if (row < m) model[ml.off_invn + row] = (nn > T(0)) ? T(1) / sqrt(nn) : T(1);
    }
    for (int i = tid; i < m; i += 256) {
        const T e = hb[i];
        model[ml.off_e + i] = e;
        for (int c = 0; c < nx; ++c) model[ml.off_Hx + (size_t)i * nx + c] = e - hb[(size_t)(1 + c) * m + i];
    }
}

// Plant + reference update of one control period (include/mpcqp.h: mpcqp_wip_advance_batch), optionally with the
// loops' bookkeeping (stats[0] += failures, stats[1] += iterations). State = [r, theta, r', theta'].
// ONE WAVEFRONT PER LOOP, 16 loops per workgroup: every lane integrates the plant (same cost as one lane), then the
// lanes write the N reference rows side by side (a thread per loop kept 4 CUs busy for 20 us at 1024 loops).
template <typename T>
__global__ void __launch_bounds__(1024) mpcqp_wip_advance_kernel(T *__restrict__ states, const T *__restrict__ U,
                                                                 int64_t u_stride, const int32_t *__restrict__ status,
                                                                 const int32_t *__restrict__ iters,
                                                                 unsigned long long *__restrict__ stats, int N, T Tp, T vel,
                                                                 T omega2, T g, int nsub, T *__restrict__ x0,
                                                                 T *__restrict__ goal, T *__restrict__ targets, int64_t batch)
{
    __shared__ unsigned long long red[2 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 16 + wave;
    unsigned long long f = 0, it = 0;
    if (b < batch) {
        const bool failed = status && status[b] != 0;
        const T a = failed ? T(0) : U[b * u_stride];
        f = failed;
        it = iters ? (unsigned long long)iters[b] : 0;
        wip_period_wave<T>(lane, states + b * 4, a, N, Tp, vel, omega2, g, nsub, x0 + b * 4, goal + b * 4,
                           targets + b * (int64_t)N * 4);
    }
    if (stats) {
        if (lane == 0) {
            red[wave] = f;
            red[16 + wave] = it;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long fs = 0, is = 0;
            for (int w = 0; w < 16; ++w) {
                fs += red[w];
                is += red[16 + w];
            }
            if (fs) atomicAdd(stats, fs);
            if (is) atomicAdd(stats + 1, is);
        }
    }
}

int launch_wip_advance(int dtype, void *states, const void *U, int64_t u_stride, const int32_t *status,
                       const int32_t *iters, int64_t *stats, int N, double Tp, double vel, double length, double gravity,
                       int nsub, void *x0, void *goal, void *targets, int64_t batch, hipStream_t st)
{
    const unsigned grid = (unsigned)((batch + 15) / 16);
    const double omega2 = gravity / length;
    unsigned long long *sp = (unsigned long long *)stats;
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL(mpcqp_wip_advance_kernel<double>, dim3(grid), dim3(1024), 0, st, (double *)states,
                           (const double *)U, u_stride, status, iters, sp, N, Tp, vel, omega2, gravity, nsub, (double *)x0,
                           (double *)goal, (double *)targets, batch);
    else
        hipLaunchKernelGGL(mpcqp_wip_advance_kernel<float>, dim3(grid), dim3(1024), 0, st, (float *)states,
                           (const float *)U, u_stride, status, iters, sp, N, (float)Tp, (float)vel, (float)omega2,
                           (float)gravity, nsub, (float *)x0, (float *)goal, (float *)targets, batch);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- loop statistics
// stats[0] += #(status != 0), stats[1] += sum(iters): one launch instead of the half-dozen tiny tensor kernels the
// closed loops used per period for the same bookkeeping.
__global__ void __launch_bounds__(256) mpcqp_stats_kernel(const int32_t *__restrict__ status, const int32_t *__restrict__ iters,
                                                          int64_t batch, unsigned long long *__restrict__ stats)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long f = 0, it = 0;
    if (i < batch) {
        f = status[i] != 0;
        it = (unsigned long long)iters[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
        f += __shfl_xor(f, o);
        it += __shfl_xor(it, o);
    }
    if ((threadIdx.x & 63) == 0 && (f | it)) {
        if (f) atomicAdd(stats, f);
        if (it) atomicAdd(stats + 1, it);
    }
}

int launch_stats(const int32_t *status, const int32_t *iters, int64_t batch, int64_t *stats, hipStream_t st)
{
    hipLaunchKernelGGL(mpcqp_stats_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, status, iters, batch,
                       (unsigned long long *)stats);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------- LIPM walking loop, one period
// examples/lipm_walking_controller.py:304-333 for `batch` walkers, one thread each: integrate the first
// jerk of the plan exactly for nsub sub-steps (:216-236), advance the footstep phase (:125-132, :329-332),
// then write the next problem: x0, goal and the per-step ZMP bounds e [N, 2] of the receding horizon
// (PhaseStepper.get_nb_steps :134-165 + update_goal_and_constraints :179-213).
template <typename T>
__global__ void __launch_bounds__(256) mpcqp_lipm_advance_kernel(
    T *__restrict__ states, const T *__restrict__ U, int64_t u_stride, const int32_t *__restrict__ status, int N, T Tp,
    int nsub, int nb_dsp, int nb_ssp, T max_zmp, int64_t *__restrict__ index, int64_t *__restrict__ stride_index,
    T *__restrict__ support, const T *__restrict__ strides, const T *__restrict__ foot_size, T *__restrict__ x0,
    T *__restrict__ goal, T *__restrict__ e, int64_t batch, const int32_t *__restrict__ iters,
    unsigned long long *__restrict__ stats)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (stats) {  // the loops' bookkeeping (as mpcqp_accumulate_stats): per-wavefront sums, one atomic pair each
        unsigned long long f = 0, it = 0;
        if (b < batch) {
            f = status && status[b] != 0;
            it = iters ? (unsigned long long)iters[b] : 0;
        }
        for (int o = 32; o > 0; o >>= 1) {
            f += __shfl_xor(f, o);
            it += __shfl_xor(it, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (f) atomicAdd(stats, f);
            if (it) atomicAdd(stats + 1, it);
        }
    }
    if (b >= batch) return;
    T p = states[b * 3 + 0], v = states[b * 3 + 1], a = states[b * 3 + 2];
    int idx = (int)index[b], sidx = (int)stride_index[b];
    T cur = support[b];
    const T s0 = strides[b * 2 + 0], s1 = strides[b * 2 + 1];
    if (U) {
        const T jerk = (status && status[b] != 0) ? T(0) : U[b * u_stride];
        const T dt = Tp / (T)nsub;
        for (int i = 0; i < nsub; ++i) {
            const T p2 = p + dt * (v + dt * (a / 2 + dt * jerk / 6));
            const T v2 = v + dt * (a + dt * (jerk / 2));
            a = a + dt * jerk;
            p = p2;
            v = v2;
        }
        idx += 1;
        if (idx >= nb_dsp + nb_ssp) idx = 0;
        if (idx == 0) {  // the swing foot lands: it becomes the support foot
            cur = cur + (sidx == 0 ? s0 : s1);
            sidx = (sidx + 1) % 2;
        }
        states[b * 3 + 0] = p;
        states[b * 3 + 1] = v;
        states[b * 3 + 2] = a;
        index[b] = idx;
        stride_index[b] = sidx;
        support[b] = cur;
    }
    x0[b * 3 + 0] = p;
    x0[b * 3 + 1] = v;
    x0[b * 3 + 2] = a;
    // segments of the horizon
    int offset = idx;
    const int init_dsp = max(0, nb_dsp - offset);
    offset = max(0, offset - nb_dsp);
    const int init_ssp = max(0, nb_ssp - offset);
    int remaining = N - init_dsp - init_ssp;
    const int next_dsp = min(nb_dsp, remaining);
    remaining = max(0, remaining - nb_dsp);
    const int next_ssp = min(nb_ssp, remaining);
    remaining = max(0, remaining - nb_ssp);
    const int last_dsp = min(nb_dsp, remaining);
    const T nxt = cur + (sidx == 0 ? s0 : s1);
    const T last = nxt + (sidx == 0 ? s1 : s0);
    const T half = T(0.5) * foot_size[b];
    const int e1 = init_dsp, e2 = e1 + init_ssp, e3 = e2 + next_dsp, e4 = e3 + next_ssp, e5 = e4 + last_dsp;
    T *eb = e + b * (int64_t)N * 2;
    for (int k = 0; k < N; ++k) {
        T hi = max_zmp, lo = max_zmp;
        if (k >= e1 && k < e2) {
            hi = cur + half;
            lo = -(cur - half);
        } else if (k >= e3 && k < e4) {
            hi = nxt + half;
            lo = -(nxt - half);
        } else if (k >= e5) {
            hi = last + half;
            lo = -(last - half);
        }
        eb[2 * k] = hi;
        eb[2 * k + 1] = lo;
    }
    goal[b * 3 + 0] = last_dsp > 0 ? last : nxt;
    goal[b * 3 + 1] = T(0);
    goal[b * 3 + 2] = T(0);
}

int launch_lipm_advance(int dtype, void *states, const void *U, int64_t u_stride, const int32_t *status, int N,
                        double Tp, int nsub, int nb_dsp, int nb_ssp, double max_zmp, int64_t *index,
                        int64_t *stride_index, void *support, const void *strides, const void *foot_size, void *x0,
                        void *goal, void *e, int64_t batch, const int32_t *iters, int64_t *stats, hipStream_t st)
{
    const unsigned grid = (unsigned)((batch + 255) / 256);
    unsigned long long *sp = (unsigned long long *)stats;
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL(mpcqp_lipm_advance_kernel<double>, dim3(grid), dim3(256), 0, st, (double *)states,
                           (const double *)U, u_stride, status, N, Tp, nsub, nb_dsp, nb_ssp, max_zmp, index,
                           stride_index, (double *)support, (const double *)strides, (const double *)foot_size,
                           (double *)x0, (double *)goal, (double *)e, batch, iters, sp);
    else
        hipLaunchKernelGGL(mpcqp_lipm_advance_kernel<float>, dim3(grid), dim3(256), 0, st, (float *)states,
                           (const float *)U, u_stride, status, N, (float)Tp, nsub, nb_dsp, nb_ssp, (float)max_zmp,
                           index, stride_index, (float *)support, (const float *)strides, (const float *)foot_size,
                           (float *)x0, (float *)goal, (float *)e, batch, iters, sp);
    return (int)hipGetLastError();
}

int launch_factor_model(const KernelArgs &ka, int dtype, const void *P, const void *G, const void *qb, const void *hb,
                        void *model, hipStream_t st)
{
    const ModelLayout ml = make_model_layout(ka.nx, ka.N, ka.n, ka.m);
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    const size_t lds = ((size_t)ka.n * (ka.n | 1) + ka.n) * esz;
    if (lds > kLdsBytesPerCU) return MPCQP_ETOOLARGE;
    if (dtype == MPCQP_F64) {
        auto kern = mpcqp_model_kernel<double>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, st, ka, ml, (const double *)P, (const double *)G,
                           (const double *)qb, (const double *)hb, (double *)model);
    } else {
        auto kern = mpcqp_model_kernel<float>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, st, ka, ml, (const float *)P, (const float *)G,
                           (const float *)qb, (const float *)hb, (float *)model);
    }
    return (int)hipGetLastError();
}

}  // namespace mpcqp
