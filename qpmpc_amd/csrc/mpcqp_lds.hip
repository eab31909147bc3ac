// mpcqp_lds.hip -- gfx950 kernels for the on-chip path: one MPC problem per
// workgroup, every matrix of the problem resident in that CU's LDS.
//
// Replaces (reference file:line)
//   build  : qpmpc/mpc_qp.py:53-114 (Phi/Psi propagation, G_k, h_k, P) and
//            :129-149 (q), :151-163 (h update)
//   solve  : qpsolvers.solve_problem(...) at qpmpc/solve_mpc.py:43
//   rollout: qpmpc/mpc_problem.py:316-335
//
// Solver: dual active-set method (Goldfarb & Idnani 1983) re-derived for a
// 64-wide wavefront instead of a scalar core:
//   * works in y = L'u coordinates (P = L L'), so the GI matrix J is an
//     orthogonal Q and the constraint matrix becomes M = G L^-T;
//   * adding a constraint applies ONE Householder reflector to Q2 (a rank-1
//     update, one row per lane) instead of a serial chain of Givens rotations;
//   * S = R^-1 is kept explicitly: r = S d1 is a mat-vec, and the new column
//     after an add is [-r; 1]/R_qq (no back-substitution on the critical path);
//   * dropping a constraint uses rotations whose coefficients are prefix norms
//     of one row of S, so they are computed lane-parallel and then applied
//     row-wise without inter-lane dependencies.
// Data layout in LDS: row-major with an ODD row stride ld >= n+1, so that both
// "one row per lane" (stride ld, conflict-free for 8-byte accesses) and "one
// column per lane" (stride 1) sweeps are bank-conflict free.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpcqp.h"
#include "mpcqp_internal.h"

namespace mpcqp {

// ------------------------------------------------------------- reductions
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <typename T>
__device__ __forceinline__ void wave_argmin(T &v, int &i)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        if (ov < v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
    }
}

// Workgroup barrier. With __launch_bounds__(64) the backend lowers s_barrier of
// a single-wave workgroup to a wave_barrier (no hardware barrier is issued).
__device__ __forceinline__ void bsync() { __syncthreads(); }

template <typename T, int WAVES>
__device__ __forceinline__ T block_sum(T v, T *redv, int tid)
{
    v = wave_sum(v);
    if constexpr (WAVES > 1) {
        if ((tid & 63) == 0) redv[tid >> 6] = v;
        bsync();
        v = redv[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += redv[w];
        bsync();
    }
    return v;
}

template <typename T, int WAVES>
__device__ __forceinline__ void block_argmin(T &v, int &i, T *redv, int *redi, int tid)
{
    wave_argmin(v, i);
    if constexpr (WAVES > 1) {
        if ((tid & 63) == 0) {
            redv[tid >> 6] = v;
            redi[tid >> 6] = i;
        }
        bsync();
        v = redv[0];
        i = redi[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            const T ov = redv[w];
            const int oi = redi[w];
            if (ov < v || (ov == v && oi < i)) {
                v = ov;
                i = oi;
            }
        }
        bsync();
    }
}

template <typename T> struct Lim;
template <> struct Lim<double> {
    static __device__ __forceinline__ double inf() { return HUGE_VAL; }
    // |d2|^2 / |d|^2 below this: the selected row depends on the active ones. (1e-28 until round 2: rounding noise let
    // dependent rows through and a few INFEASIBLE problems in 10^4 came back 'solved' with |u| ~ 1e13 and violated rows --
    // found by tools/stress_pair.py. The other kernels use 1e-14 .. 1e-10 on this ratio; with 1e-14 HERE one borderline
    // problem of the stress run lost accuracy (1e-7 against the other kernels), 1e-18 rejects the noise and keeps it.)
    static __device__ __forceinline__ double tiny() { return 1e-18; }
};
template <> struct Lim<float> {
    static __device__ __forceinline__ float inf() { return HUGE_VALF; }
    static __device__ __forceinline__ float tiny() { return 1e-10f; }
};

// ------------------------------------------------------------------ build
// Psi blocks k = 1..N live in X: block k at X + (k-1)*nx*ld, nx rows of ld.
// Column n of each block is the free response xf_k = Phi_k x0; after h is
// formed it is overwritten by the tracking residual xf_k - ref_k.
template <typename T, int WAVES>
__device__ __forceinline__ void build_phase(const KernelArgs &ka, const Layout &L, T *sm,
                                            int64_t prob, int tid)
{
    constexpr int BS = 64 * WAVES;
    const int nx = ka.nx, nu = ka.nu, N = ka.N, mk = ka.mk, n = ka.n, m = ka.m, ld = L.ld;
    T *X = sm + L.off_X, *Pm = sm + L.off_P, *Mm = sm + L.off_M, *hv = sm + L.off_h;
    T *Ast = sm + L.off_A, *Bst = sm + L.off_B, *x0s = sm + L.off_x0;
    const T *gA = (const T *)ka.A.ptr + prob * ka.A.batch_stride;
    const T *gB = (const T *)ka.B.ptr + prob * ka.B.batch_stride;
    const T *gC = ka.C.ptr ? (const T *)ka.C.ptr + prob * ka.C.batch_stride : nullptr;
    const T *gD = ka.D.ptr ? (const T *)ka.D.ptr + prob * ka.D.batch_stride : nullptr;
    const T *ge = (const T *)ka.e.ptr + prob * ka.e.batch_stride;
    const T *gx0 = (const T *)ka.x0.ptr + prob * ka.x0.batch_stride;
    const T *ggoal = ka.goal.ptr ? (const T *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
    const T *gtgt = ka.targets.ptr ? (const T *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
    const int sA = ka.A.step_stride ? nx * nx : 0, sB = ka.B.step_stride ? nx * nu : 0;
    const int sC = (int)ka.C.step_stride, sD = (int)ka.D.step_stride, se = (int)ka.e.step_stride;

    // stage the dynamics (coalesced: a problem's steps are packed) and clear Psi
    const int nA = (sA ? N : 1) * nx * nx, nB = (sB ? N : 1) * nx * nu;
    for (int i = tid; i < nA; i += BS) Ast[i] = gA[i];
    for (int i = tid; i < nB; i += BS) Bst[i] = gB[i];
    for (int i = tid; i < nx; i += BS) x0s[i] = gx0[i];
    for (int i = tid; i < N * nx * ld; i += BS) X[i] = T(0);
    bsync();

    // chains: column c<n is A_{k-1}..A_{j+1} B_j e_i (mpc_qp.py:89-90); column n is
    // the free response (the reference's Phi_k x0). One column per lane, no
    // inter-lane dependency, so no barrier inside the k loop.
    for (int c = tid; c <= n; c += BS) {
        int k0;
        if (c < n) {
            const int j = c / nu, i = c - j * nu;
            for (int s = 0; s < nx; ++s) X[(j * nx + s) * ld + c] = Bst[j * sB + s * nu + i];
            k0 = j + 1;
        } else {
            for (int r = 0; r < nx; ++r) {
                T acc = T(0);
                for (int s = 0; s < nx; ++s) acc += Ast[r * nx + s] * x0s[s];
                X[r * ld + c] = acc;
            }
            k0 = 1;
        }
        for (int k = k0; k < N; ++k) {  // block k+1 <- A_k * block k
            const T *Ak = Ast + k * sA;
            const T *src = X + (k - 1) * nx * ld + c;
            T *dst = X + k * nx * ld + c;
            for (int r = 0; r < nx; ++r) {
                T acc = T(0);
                for (int s = 0; s < nx; ++s) acc += Ak[r * nx + s] * src[s * ld];
                dst[r * ld] = acc;
            }
        }
    }
    bsync();

    // G_k = [D_k in block k] + C_k Psi_k (mpc_qp.py:67,73-78), h_k = e_k - C_k xf_k (:68-72)
    for (int idx = tid; idx < m * (n + 1); idx += BS) {
        const int row = idx / (n + 1), c = idx - row * (n + 1);
        const int k = row / mk, i = row - k * mk;
        T acc = T(0);
        if (gC) {
            const T *Ci = gC + k * sC + i * nx;
            if (k >= 1) {
                const T *src = X + (k - 1) * nx * ld + c;
                for (int s = 0; s < nx; ++s) acc += Ci[s] * src[s * ld];
            } else if (c == n) {
                for (int s = 0; s < nx; ++s) acc += Ci[s] * x0s[s];
            }
        }
        if (c < n) {
            if (gD && c >= k * nu && c < (k + 1) * nu) acc += gD[k * sD + i * nu + (c - k * nu)];
            Mm[row * ld + c] = acc;
        } else {
            hv[row] = ge[k * se + i] - acc;
        }
    }
    bsync();

    // tracking residuals in column n: block k<N against targets[k], block N against goal
    for (int idx = tid; idx < N * nx; idx += BS) {
        const int k = idx / nx + 1, s = idx - (k - 1) * nx;
        T ref = T(0);
        if (k < N) {
            if ((ka.flags & MPCQP_Q_STAGE) && gtgt) ref = gtgt[k * nx + s];
        } else {
            if ((ka.flags & MPCQP_Q_TERMINAL) && ggoal) ref = ggoal[s];
        }
        X[((k - 1) * nx + s) * ld + n] -= ref;
    }
    bsync();

    // P = w_u I + w_t psi_N' psi_N + w_x Psi' Psi (mpc_qp.py:99-105), lower triangle,
    // using the block-triangular zero pattern: Psi_k[:, a] = 0 for k <= a/nu.
    // Row n of the extended Gram is q (mpc_qp.py:139-149): q lands in row m of M.
    const T wt = (T)ka.wt, wx = (T)ka.wx, wu = (T)ka.wu;
    for (int idx = tid; idx < (n + 1) * n; idx += BS) {
        const int a = idx / n, b = idx - a * n;  // a = n  <=> the q row
        if (a < n && b > a) continue;
        const bool isq = (a == n);
        const int hi = isq ? b : a;
        T acc = (a == b) ? wu : T(0);
        const bool use_t = isq ? (ka.flags & MPCQP_Q_TERMINAL) : (ka.flags & MPCQP_P_TERMINAL);
        const bool use_s = isq ? (ka.flags & MPCQP_Q_STAGE) : (ka.flags & MPCQP_P_STAGE);
        if (use_t) {
            const T *blk = X + (N - 1) * nx * ld;
            T t = T(0);
            for (int s = 0; s < nx; ++s) t += blk[s * ld + a] * blk[s * ld + b];
            acc += wt * t;
        }
        if (use_s) {
            T t = T(0);
            for (int k = hi / nu + 1; k < N; ++k) {
                const T *blk = X + (k - 1) * nx * ld;
                for (int s = 0; s < nx; ++s) t += blk[s * ld + a] * blk[s * ld + b];
            }
            acc += wx * t;
        }
        if (isq)
            Mm[m * ld + b] = acc;
        else
            Pm[a * ld + b] = acc;
    }
    bsync();
}

// ------------------------------------------------------------------ solve
// In: Pm lower triangle of P (n x ld), Mm rows 0..m-1 = G, row m = q, hv = h.
// Out: xs[n] (solution in original coordinates), u/act/where (multipliers).
// PREFAC (shared-model path): Mm already holds M = G L^-T, y the start point y0, hs the
// 1/|M_i|; nothing is factorised here and u = L^-T y comes from the model's L^-T rows.
template <typename T, int WAVES, bool PREFAC = false>
__device__ __forceinline__ void solve_phase(const Layout &L, T *sm, int n, int m, int max_iter,
                                            T tol, int tid, int &status, int &iters,
                                            const T *linvT = nullptr, int nc = 0)
{
    constexpr int BS = 64 * WAVES;
    const int ld = L.ld;
    T *Pm = sm + L.off_P, *Mm = sm + L.off_M, *Qm = sm + L.off_X, *Sm = Qm + n * ld;
    T *hv = sm + L.off_h, *hs = sm + L.off_hs, *sv = sm + L.off_s, *y = sm + L.off_y;
    T *z = sm + L.off_z, *d = sm + L.off_d, *r = sm + L.off_r, *u = sm + L.off_u;
    T *cs = sm + L.off_cs, *sn = sm + L.off_sn, *inv = sm + L.off_inv, *xs = sm + L.off_xs;
    T *redv = sm + L.off_red;
    int *act = (int *)(sm + L.off_int), *where = act + (n + 2), *redi = where + m;
    const T INF = Lim<T>::inf();

    status = MPCQP_MAX_ITER;
    iters = 0;

    if constexpr (!PREFAC) {
    // Cholesky P = L L' in place (strict lower part + inv[] = 1/L_jj), left-looking.
    // Every lane recomputes the pivot of column j so one barrier per column suffices.
    for (int j = 0; j < n; ++j) {
        T piv = Pm[j * ld + j];
        for (int k = 0; k < j; ++k) piv -= Pm[j * ld + k] * Pm[j * ld + k];
        if (!(piv > T(0))) {  // wave-uniform: every lane computed the same pivot
            status = MPCQP_NOT_PD;
            return;
        }
        const T rinv = T(1) / sqrt(piv);
        for (int i = j + 1 + tid; i < n; i += BS) {
            T v = Pm[i * ld + j];
            for (int k = 0; k < j; ++k) v -= Pm[i * ld + k] * Pm[j * ld + k];
            Pm[i * ld + j] = v * rinv;
        }
        if (tid == 0) inv[j] = rinv;
        bsync();
    }
    // M = G L^-T and w = L^-1 q: one row per lane, forward substitution
    for (int i = tid; i <= m; i += BS) {
        T *row = Mm + i * ld;
        for (int j = 0; j < n; ++j) {
            T v = row[j];
            for (int k = 0; k < j; ++k) v -= row[k] * Pm[j * ld + k];
            row[j] = v * inv[j];
        }
    }
    }
    for (int i = tid; i < n * ld; i += BS) {
        const int a = i / ld, b = i - a * ld;
        Qm[i] = (a == b) ? T(1) : T(0);
        Sm[i] = T(0);
    }
    // hs[i] = 1/|M_i|: rows are ranked by their distance to the hyperplane in the
    // P^-1 metric (classic Goldfarb-Idnani rule: fewest iterations, hardly any drop)
    for (int i = tid; i < m; i += BS) {
        where[i] = -1;
        if constexpr (!PREFAC) {
            const T *row = Mm + i * ld;
            T nn = T(0);
            for (int k = 0; k < n; ++k) nn += row[k] * row[k];
            hs[i] = (nn > T(0)) ? T(1) / sqrt(nn) : T(1);
        }
    }
    for (int i = tid; i <= n; i += BS) u[i] = T(0);
    bsync();
    if constexpr (!PREFAC) {
        for (int k = tid; k < n; k += BS) y[k] = -Mm[m * ld + k];
    }
    bsync();

    int nq = 0, refined = 0;
    for (;;) {
        // step 1: slacks in y coordinates and the most violated inactive row
        T best = INF;
        int bi = 0x7fffffff;
        for (int i = tid; i < m; i += BS) {
            const T *row = Mm + i * ld;
            T s = hv[i];
            for (int k = 0; k < n; ++k) s -= row[k] * y[k];
            sv[i] = s;
            const bool violated = (where[i] < 0) && (s < -(tol + tol * fabs(hv[i])));
            const T key = violated ? s * hs[i] : INF;
            if (key < best) {
                best = key;
                bi = i;
            }
        }
        block_argmin<T, WAVES>(best, bi, redv, redi, tid);
        if (!(best < INF)) {
            // No inactive row is violated. The ACTIVE rows are on their bounds by construction -- but y is moved by increments,
            // and after several hundred iterations of a degenerate problem they drift (a stress run: a row 5e-7 inside its
            // bound after 455 iterations). Their slacks rho_A were just evaluated with all the others: when one is off, y is put
            // back on the active hyperplanes, dy = M_A' (M_A M_A')^-1 rho_A = -Q1 S' rho_A (-M_A' = Q1 R, S = R^-1), the
            // multipliers follow (u -= S S' rho_A), and the slacks are evaluated again (at most twice).
            if (nq > 0 && refined < 2) {
                T flag = T(0);
                for (int i = tid; i < nq; i += BS) {
                    const int a = act[i];
                    const T rho = sv[a];
                    r[i] = rho;
                    if (!(fabs(rho) <= T(64) * (tol + tol * fabs(hv[a])))) flag = T(1);
                }
                if (block_sum<T, WAVES>(flag, redv, tid) > T(0)) {
                    ++refined;
                    bsync();
                    for (int j = tid; j < nq; j += BS) {  // g = S' rho
                        T acc = T(0);
                        for (int i = 0; i <= j; ++i) acc += Sm[i * ld + j] * r[i];
                        d[j] = acc;
                    }
                    bsync();
                    for (int k = tid; k < n; k += BS) {  // y -= Q1 g
                        const T *row = Qm + k * ld;
                        T acc = T(0);
                        for (int j = 0; j < nq; ++j) acc += row[j] * d[j];
                        y[k] -= acc;
                    }
                    for (int i = tid; i < nq; i += BS) {  // u -= S g
                        const T *row = Sm + i * ld;
                        T acc = T(0);
                        for (int j = i; j < nq; ++j) acc += row[j] * d[j];
                        const T v = u[i] - acc;
                        u[i] = v < T(0) ? T(0) : v;
                    }
                    bsync();
                    continue;
                }
            }
            status = MPCQP_SOLVED;
            break;
        }
        bsync();
        const int p = bi;
        T sp = sv[p];
        const T *Mp = Mm + p * ld;
        // d = Q' n+ with n+ = -M_p
        for (int j = tid; j < n; j += BS) {
            T acc = T(0);
            for (int k = 0; k < n; ++k) acc -= Qm[k * ld + j] * Mp[k];
            d[j] = acc;
        }
        bsync();
        T up = T(0);
        bool added = false;
        while (!added) {
            if (iters >= max_iter) return;  // status stays MAX_ITER
            ++iters;
            // |d2|^2 and |d|^2
            T p2 = T(0), pa = T(0);
            for (int j = tid; j < n; j += BS) {
                const T dj = d[j];
                pa += dj * dj;
                if (j >= nq) p2 += dj * dj;
            }
            const T delta2 = block_sum<T, WAVES>(p2, redv, tid);
            const T dd = block_sum<T, WAVES>(pa, redv, tid);
            // z = Q2 d2 (primal direction), r = S d1 (dual direction), ratio test
            for (int k = tid; k < n; k += BS) {
                const T *row = Qm + k * ld;
                T acc = T(0);
                for (int j = nq; j < n; ++j) acc += row[j] * d[j];
                z[k] = acc;
            }
            T t1 = INF;
            int l = 0x7fffffff;
            for (int i = tid; i < nq; i += BS) {
                const T *row = Sm + i * ld;
                T acc = T(0);
                for (int j = i; j < nq; ++j) acc += row[j] * d[j];
                r[i] = acc;
                if (acc > T(0)) {
                    const T ratio = u[i] / acc;
                    if (ratio < t1) {
                        t1 = ratio;
                        l = i;
                    }
                }
            }
            block_argmin<T, WAVES>(t1, l, redv, redi, tid);
            const bool can_move = (nq < n) && (delta2 > Lim<T>::tiny() * dd) && (delta2 > T(0));
            const T t2 = can_move ? -sp / delta2 : INF;
            const T t = t1 < t2 ? t1 : t2;
            if (!(t < INF)) {
                status = MPCQP_INFEASIBLE;
                return;
            }
            bsync();  // z, r visible
            if (can_move)
                for (int k = tid; k < n; k += BS) y[k] += t * z[k];
            for (int i = tid; i < nq; i += BS) u[i] -= t * r[i];
            up += t;
            if (t2 <= t1) {
                // full step: add row p. Householder H = I - beta v v', v = d2 + sigma*delta*e1,
                // Q2 <- Q2 H = Q2 - beta (Q2 v) v', and Q2 v = z + sigma*delta*Q[:,nq].
                const T delta = sqrt(delta2);
                const T dq = d[nq];
                const T sig = dq >= T(0) ? T(1) : T(-1);
                const T beta = T(1) / (delta * (delta + fabs(dq)));
                const T v0 = dq + sig * delta;
                for (int k = tid; k < n; k += BS) {
                    T *row = Qm + k * ld;
                    const T wk = beta * (z[k] + sig * delta * row[nq]);
                    row[nq] -= wk * v0;
                    for (int j = nq + 1; j < n; ++j) row[j] -= wk * d[j];
                }
                const T rinv = T(-1) / (sig * delta);  // 1 / R_qq
                for (int i = tid; i < nq; i += BS) Sm[i * ld + nq] = -r[i] * rinv;
                if (tid == 0) {
                    Sm[nq * ld + nq] = rinv;
                    u[nq] = up;
                    act[nq] = p;
                    where[p] = nq;
                }
                ++nq;
                added = true;
                bsync();
            } else {
                // partial step: drop the blocking row at position l.
                // Rotation jj acts on columns (l+jj, l+jj+1) and zeroes S[l][l+jj];
                // its coefficients are prefix norms of a = S[l][l..nq-1].
                const int K = nq - l;
                const T *a = Sm + l * ld + l;
                for (int jj = tid; jj < K - 1; jj += BS) {
                    T acc = T(0);
                    for (int i = 0; i <= jj; ++i) acc += a[i] * a[i];
                    const T lo = (jj == 0) ? a[0] : sqrt(acc);
                    const T nxt = a[jj + 1];
                    const T hi = sqrt(acc + nxt * nxt);
                    T c = T(1), s = T(0);
                    if (hi > T(0)) {
                        c = nxt / hi;
                        s = lo / hi;
                    }
                    cs[jj] = c;
                    sn[jj] = s;
                }
                bsync();
                // apply to every row of Q (n), every live row of S (nq) and to d
                for (int t_ = tid; t_ < n + nq + 1; t_ += BS) {
                    T *row = (t_ < n) ? (Qm + t_ * ld) : (t_ < n + nq ? Sm + (t_ - n) * ld : d);
                    T xj = row[l];
                    for (int jj = 0; jj < K - 1; ++jj) {
                        const T c = cs[jj], s = sn[jj];
                        const T xn = row[l + jj + 1];
                        row[l + jj] = c * xj - s * xn;
                        xj = s * xj + c * xn;
                    }
                    row[nq - 1] = xj;
                }
                bsync();
                // S <- rows without l (shift up), last row and last column cleared
                for (int j = tid; j < nq; j += BS) {
                    if (j == nq - 1) {
                        for (int i = 0; i < nq; ++i) Sm[i * ld + j] = T(0);
                    } else {
                        for (int i = l; i < nq - 1; ++i) Sm[i * ld + j] = Sm[(i + 1) * ld + j];
                        Sm[(nq - 1) * ld + j] = T(0);
                    }
                }
                // act/u/where: positions l+1..nq-1 move down by one (nq - l - 1 < BS always)
                const int pos = l + tid;
                int a_next = 0;
                T u_next = T(0);
                if (pos < nq - 1) {
                    a_next = act[pos + 1];
                    u_next = u[pos + 1];
                }
                if (tid == 0) where[act[l]] = -1;
                bsync();
                if (pos < nq - 1) {
                    act[pos] = a_next;
                    u[pos] = u_next;
                    where[a_next] = pos;
                }
                if (tid == 0) u[nq - 1] = T(0);
                --nq;
                if (can_move) sp = sp * (T(1) - t / t2);
                bsync();
            }
        }
    }
    if constexpr (PREFAC) {
        // u = L^-T y with the rows of L^-T taken from the shared model
        for (int k = tid; k < n; k += BS) {
            const T *row = linvT + (size_t)k * nc;
            T acc = T(0);
            for (int jj = 0; jj < n; ++jj) acc += row[jj] * y[jj];
            xs[k] = acc;
        }
        bsync();
    } else {
        // u = L^-T y (column-oriented back substitution, one barrier per column)
        for (int i = n - 1; i >= 0; --i) {
            const T xi = y[i] * inv[i];
            if (tid == 0) xs[i] = xi;
            for (int k = tid; k < i; k += BS) y[k] -= Pm[i * ld + k] * xi;
            bsync();
        }
    }
}

// ------------------------------------------------------------------ kernels
// GWS: the carve lives in a per-problem slice of a global workspace (ka.ws) instead of
// LDS -- the same code path for problems too large for one CU (n = 256, m = 1024).
template <typename T, int WAVES, int MODE, bool GWS = false>
__global__ void __launch_bounds__(64 * WAVES) mpcqp_lds_kernel(const KernelArgs ka, const Layout L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sm = GWS ? (T *)ka.ws + (int64_t)blockIdx.x * L.total : (T *)smem_raw;
    constexpr int BS = 64 * WAVES;
    const int tid = threadIdx.x;
    const int64_t prob = blockIdx.x;
    const int n = ka.n, m = ka.m, ld = L.ld;
    T *Pm = sm + L.off_P, *Mm = sm + L.off_M, *hv = sm + L.off_h;

    if constexpr (MODE == MODE_MODEL) {
        // shared model: M rows and 1/|M_i| from the model, h and y0 from this problem's states
        const ModelLayout ml = make_model_layout(ka.nx, ka.N, n, m);
        const T *model = (const T *)ka.model;
        const int nx = ka.nx, nT = ka.N * ka.nx, nc = ml.nc;
        const T *x0 = (const T *)ka.x0.ptr + prob * ka.x0.batch_stride;
        const T *goal = ka.goal.ptr ? (const T *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
        const T *tgt = ka.targets.ptr ? (const T *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
        T *hs = sm + L.off_hs, *y = sm + L.off_y;
        for (int i = tid; i < m * n; i += BS) Mm[(i / n) * ld + (i % n)] = model[ml.off_M + (size_t)(i / n) * nc + (i % n)];
        for (int i = tid; i < m; i += BS) {
            T hh = model[ml.off_e + i];
            for (int c = 0; c < nx; ++c) hh -= model[ml.off_Hx + (size_t)i * nx + c] * x0[c];
            hv[i] = hh;
            hs[i] = model[ml.off_invn + i];
        }
        for (int k = tid; k < n; k += BS) {
            T wk = T(0);
            for (int c = 0; c < nx; ++c) wk += model[ml.off_Wx + (size_t)k * nx + c] * x0[c];
            if ((ka.flags & MPCQP_Q_TERMINAL) && goal)
                for (int c = 0; c < nx; ++c) wk -= model[ml.off_Wg + (size_t)k * nx + c] * goal[c];
            if ((ka.flags & MPCQP_Q_STAGE) && tgt)
                for (int j2 = 0; j2 < nT; ++j2) wk -= model[ml.off_Wt + (size_t)k * nT + j2] * tgt[j2];
            y[k] = -wk;  // y0 = -L^-1 q
        }
        bsync();
    } else if constexpr (MODE == MODE_SOLVE) {
        const T *gP = (const T *)ka.P + prob * (int64_t)n * n;
        const T *gG = (const T *)ka.G + prob * (int64_t)m * n;
        const T *gq = (const T *)ka.q + prob * (int64_t)n;
        const T *gh = (const T *)ka.h + prob * (int64_t)m;
        for (int i = tid; i < n * n; i += BS) Pm[(i / n) * ld + (i % n)] = gP[i];
        for (int i = tid; i < m * n; i += BS) Mm[(i / n) * ld + (i % n)] = gG[i];
        for (int i = tid; i < n; i += BS) Mm[m * ld + i] = gq[i];
        for (int i = tid; i < m; i += BS) hv[i] = gh[i];
        bsync();
    } else {
        build_phase<T, WAVES>(ka, L, sm, prob, tid);
    }

    if constexpr (MODE == MODE_CONDENSE) {
        T *oP = (T *)ka.P + prob * (int64_t)n * n;
        T *oG = (T *)ka.G + prob * (int64_t)m * n;
        T *oq = (T *)ka.q + prob * (int64_t)n;
        T *oh = (T *)ka.h + prob * (int64_t)m;
        for (int i = tid; i < n * n; i += BS) {
            const int a = i / n, b = i - a * n;
            oP[i] = (b <= a) ? Pm[a * ld + b] : Pm[b * ld + a];
        }
        for (int i = tid; i < m * n; i += BS) oG[i] = Mm[(i / n) * ld + (i % n)];
        for (int i = tid; i < n; i += BS) oq[i] = Mm[m * ld + i];
        for (int i = tid; i < m; i += BS) oh[i] = hv[i];
        if (ka.Psi) {
            const int nx = ka.nx, N = ka.N;
            T *oPsi = (T *)ka.Psi + prob * (int64_t)(N + 1) * nx * n;
            const T *X = sm + L.off_X;
            for (int i = tid; i < (N + 1) * nx * n; i += BS) {
                const int rowi = i / n, c = i - rowi * n;
                oPsi[i] = (rowi < nx) ? T(0) : X[(rowi - nx) * ld + c];
            }
        }
        return;
    } else {
        int status, iters;
        if constexpr (MODE == MODE_MODEL) {
            const ModelLayout ml = make_model_layout(ka.nx, ka.N, n, m);
            const T *model = (const T *)ka.model;
            if (model[ml.total] != T(0)) {
                status = MPCQP_NOT_PD;
                iters = 0;
            } else {
                solve_phase<T, WAVES, true>(L, sm, n, m, ka.max_iter, (T)ka.tol, tid, status, iters,
                                            model + ml.off_LinvT, ml.nc);
            }
        } else {
            solve_phase<T, WAVES>(L, sm, n, m, ka.max_iter, (T)ka.tol, tid, status, iters);
        }
        const T *xs = sm + L.off_xs, *u = sm + L.off_u;
        const int *act = (const int *)(sm + L.off_int), *where = act + (n + 2);
        T *oU = (T *)ka.U + prob * (int64_t)n;
        const bool ok = (status == MPCQP_SOLVED);
        for (int i = tid; i < n; i += BS) oU[i] = ok ? xs[i] : T(0);
        if (ka.lam) {
            T *ol = (T *)ka.lam + prob * (int64_t)m;
            for (int i = tid; i < m; i += BS) ol[i] = (ok && where[i] >= 0) ? u[where[i]] : T(0);
        }
        if (tid == 0) {
            if (ka.status) ka.status[prob] = status;
            if (ka.iters) ka.iters[prob] = iters;
        }
    }
}

// Phi_0..Phi_N (mpc_qp.py:53,88) for callers that ask for them: nx chains per
// problem, one lane each; written straight to HBM (only MPCQP.Phi/phi_last use it).
template <typename T>
__global__ void __launch_bounds__(64) mpcqp_phi_kernel(const KernelArgs ka)
{
    const int nx = ka.nx, N = ka.N;
    const int64_t prob = blockIdx.x;
    const T *gA = (const T *)ka.A.ptr + prob * ka.A.batch_stride;
    const int64_t sA = ka.A.step_stride;
    T *oPhi = (T *)ka.Phi + prob * (int64_t)(N + 1) * nx * nx;
    // a lane owns columns c, c + 64, ... (any state dimension); a column's chain only reads what the same
    // lane wrote one step earlier
    for (int c = threadIdx.x; c < nx; c += 64) {
        for (int s = 0; s < nx; ++s) oPhi[s * nx + c] = (s == c) ? T(1) : T(0);
        for (int k = 0; k < N; ++k) {
            const T *Ak = gA + k * sA;
            const T *src = oPhi + (int64_t)k * nx * nx;
            T *dst = oPhi + (int64_t)(k + 1) * nx * nx;
            for (int r = 0; r < nx; ++r) {
                T acc = T(0);
                for (int s = 0; s < nx; ++s) acc += Ak[r * nx + s] * src[s * nx + c];
                dst[r * nx + c] = acc;
            }
        }
    }
}

// q = w_t (Phi_N x0 - goal)' psi_N + w_x (Phi x0 - targets)' Psi  (mpc_qp.py:129-149)
// h = e - C Phi x0                                                   (mpc_qp.py:151-163)
// One workgroup per problem; xf = Phi_all x0 staged in LDS.
template <typename T>
__global__ void __launch_bounds__(256) mpcqp_update_kernel(const KernelArgs ka, int64_t phi_bs,
                                                           int64_t psi_bs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *xf = (T *)smem_raw;  // (N+1)*nx residuals
    const int nx = ka.nx, N = ka.N, mk = ka.mk, n = ka.n, m = ka.m, tid = threadIdx.x;
    const int64_t prob = blockIdx.x;
    const T *Phi = (const T *)ka.Phi + prob * phi_bs;
    const T *Psi = (const T *)ka.Psi + prob * psi_bs;
    const T *gx0 = (const T *)ka.x0.ptr + prob * ka.x0.batch_stride;
    const T *ggoal = ka.goal.ptr ? (const T *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
    const T *gtgt = ka.targets.ptr ? (const T *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
    for (int i = tid; i < (N + 1) * nx; i += 256) {
        T acc = T(0);
        for (int s = 0; s < nx; ++s) acc += Phi[(int64_t)i * nx + s] * gx0[s];
        xf[i] = acc;
    }
    __syncthreads();
    if (ka.h) {
        const T *gC = ka.C.ptr ? (const T *)ka.C.ptr + prob * ka.C.batch_stride : nullptr;
        const T *ge = (const T *)ka.e.ptr + prob * ka.e.batch_stride;
        T *oh = (T *)ka.h + prob * (int64_t)m;
        for (int row = tid; row < m; row += 256) {
            const int k = row / mk, i = row - k * mk;
            T acc = T(0);
            if (gC)
                for (int s = 0; s < nx; ++s) acc += gC[k * ka.C.step_stride + i * nx + s] * xf[k * nx + s];
            oh[row] = ge[k * ka.e.step_stride + i] - acc;
        }
    }
    if (ka.q) {
        T *oq = (T *)ka.q + prob * (int64_t)n;
        const T wt = (T)ka.wt, wx = (T)ka.wx;
        for (int a = tid; a < n; a += 256) {
            T acc = T(0);
            if ((ka.flags & MPCQP_Q_TERMINAL) && ggoal) {
                T t = T(0);
                for (int s = 0; s < nx; ++s) t += (xf[N * nx + s] - ggoal[s]) * Psi[((int64_t)N * nx + s) * n + a];
                acc += wt * t;
            }
            if ((ka.flags & MPCQP_Q_STAGE) && gtgt) {
                T t = T(0);
                for (int s = 0; s < N * nx; ++s) t += (xf[s] - gtgt[s]) * Psi[(int64_t)s * n + a];
                acc += wx * t;
            }
            oq[a] = acc;
        }
    }
}

// X_0 = x0, X_{k+1} = A_k X_k + B_k U_k  (mpc_problem.py:316-335). GP lanes per
// problem (GP = power of two >= nx): lane r owns state row r, the other rows
// arrive by wavefront shuffles; 64/GP problems per wavefront.
template <typename T>
__global__ void __launch_bounds__(64) mpcqp_rollout_kernel(const KernelArgs ka, int gp, int64_t batch)
{
    const int nx = ka.nx, nu = ka.nu, N = ka.N;
    const int lane = threadIdx.x, per = 64 / gp;
    const int sub = lane / gp, r = lane - sub * gp;
    const int64_t prob = (int64_t)blockIdx.x * per + sub;
    const bool live = prob < batch && r < nx;
    const int64_t pb = prob < batch ? prob : batch - 1;
    const T *gA = (const T *)ka.A.ptr + pb * ka.A.batch_stride;
    const T *gB = (const T *)ka.B.ptr + pb * ka.B.batch_stride;
    const T *gx0 = (const T *)ka.x0.ptr + pb * ka.x0.batch_stride;
    const T *gU = (const T *)ka.U + pb * (int64_t)N * nu;
    T *oX = (T *)ka.X + pb * (int64_t)(N + 1) * nx;
    T x = (r < nx) ? gx0[r] : T(0);
    if (live) oX[r] = x;
    for (int k = 0; k < N; ++k) {
        T acc = T(0);
        for (int s = 0; s < nx; ++s) {
            const T xs = __shfl(x, sub * gp + s);
            if (r < nx) acc += gA[k * ka.A.step_stride + r * nx + s] * xs;
        }
        if (r < nx)
            for (int c = 0; c < nu; ++c) acc += gB[k * ka.B.step_stride + r * nu + c] * gU[k * nu + c];
        x = acc;
        if (live) oX[(k + 1) * nx + r] = x;
    }
}

// The same roll-out for state dimensions above 64: one problem per wavefront, the state in two LDS
// buffers, lane r owns rows r, r + 64, ...
template <typename T>
__global__ void __launch_bounds__(64) mpcqp_rollout_wide_kernel(const KernelArgs ka)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rollout_smem[];
    T *xa = (T *)rollout_smem, *xb = xa + ka.nx;
    const int nx = ka.nx, nu = ka.nu, N = ka.N, lane = threadIdx.x;
    const int64_t pb = blockIdx.x;
    const T *gA = (const T *)ka.A.ptr + pb * ka.A.batch_stride;
    const T *gB = (const T *)ka.B.ptr + pb * ka.B.batch_stride;
    const T *gx0 = (const T *)ka.x0.ptr + pb * ka.x0.batch_stride;
    const T *gU = (const T *)ka.U + pb * (int64_t)N * nu;
    T *oX = (T *)ka.X + pb * (int64_t)(N + 1) * nx;
    for (int r = lane; r < nx; r += 64) oX[r] = xa[r] = gx0[r];
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const T *Ak = gA + k * ka.A.step_stride, *Bk = gB + k * ka.B.step_stride;
        for (int r = lane; r < nx; r += 64) {
            T acc = T(0);
            for (int s = 0; s < nx; ++s) acc += Ak[r * nx + s] * xa[s];
            for (int c = 0; c < nu; ++c) acc += Bk[r * nu + c] * gU[k * nu + c];
            oX[(k + 1) * nx + r] = xb[r] = acc;
        }
        __syncthreads();
        T *t = xa;
        xa = xb;
        xb = t;
    }
}

// ------------------------------------------------------------ host launchers
template <typename T, int WAVES, int MODE>
static int launch_lds(const KernelArgs &ka, const Layout &L, int64_t batch, hipStream_t st)
{
    auto kern = mpcqp_lds_kernel<T, WAVES, MODE>;
    const size_t bytes = (size_t)L.total * sizeof(T);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(64 * WAVES), bytes, st, ka, L);
    return (int)hipGetLastError();
}

template <int MODE>
int dispatch_lds(const KernelArgs &ka, const Layout &L, int dtype, int64_t batch, hipStream_t st)
{
    const bool small = (ka.n <= 32 && ka.m <= 64);
    if (dtype == MPCQP_F64)
        return small ? launch_lds<double, 1, MODE>(ka, L, batch, st) : launch_lds<double, 4, MODE>(ka, L, batch, st);
    return small ? launch_lds<float, 1, MODE>(ka, L, batch, st) : launch_lds<float, 4, MODE>(ka, L, batch, st);
}

int dispatch_gws_solve(const KernelArgs &ka, const Layout &L, int dtype, int64_t batch, hipStream_t st)
{
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL((mpcqp_lds_kernel<double, 4, MODE_SOLVE, true>), dim3((unsigned)batch), dim3(256), 0, st, ka, L);
    else
        hipLaunchKernelGGL((mpcqp_lds_kernel<float, 4, MODE_SOLVE, true>), dim3((unsigned)batch), dim3(256), 0, st, ka, L);
    return (int)hipGetLastError();
}

template int dispatch_lds<MODE_FUSED>(const KernelArgs &, const Layout &, int, int64_t, hipStream_t);
template int dispatch_lds<MODE_CONDENSE>(const KernelArgs &, const Layout &, int, int64_t, hipStream_t);
template int dispatch_lds<MODE_SOLVE>(const KernelArgs &, const Layout &, int, int64_t, hipStream_t);
template int dispatch_lds<MODE_MODEL>(const KernelArgs &, const Layout &, int, int64_t, hipStream_t);

int launch_phi(const KernelArgs &ka, int dtype, int64_t batch, hipStream_t st)
{
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL(mpcqp_phi_kernel<double>, dim3((unsigned)batch), dim3(64), 0, st, ka);
    else
        hipLaunchKernelGGL(mpcqp_phi_kernel<float>, dim3((unsigned)batch), dim3(64), 0, st, ka);
    return (int)hipGetLastError();
}

int launch_update(const KernelArgs &ka, int dtype, int64_t phi_bs, int64_t psi_bs, int64_t batch, hipStream_t st)
{
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    const size_t bytes = (size_t)(ka.N + 1) * ka.nx * esz;
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL(mpcqp_update_kernel<double>, dim3((unsigned)batch), dim3(256), bytes, st, ka, phi_bs, psi_bs);
    else
        hipLaunchKernelGGL(mpcqp_update_kernel<float>, dim3((unsigned)batch), dim3(256), bytes, st, ka, phi_bs, psi_bs);
    return (int)hipGetLastError();
}

int launch_rollout(const KernelArgs &ka, int dtype, int64_t batch, hipStream_t st)
{
    if (ka.nx > 64) {
        const size_t bytes = 2 * (size_t)ka.nx * (dtype == MPCQP_F64 ? 8 : 4);
        if (bytes > 64 * 1024) return MPCQP_ETOOLARGE;
        if (dtype == MPCQP_F64)
            hipLaunchKernelGGL(mpcqp_rollout_wide_kernel<double>, dim3((unsigned)batch), dim3(64), bytes, st, ka);
        else
            hipLaunchKernelGGL(mpcqp_rollout_wide_kernel<float>, dim3((unsigned)batch), dim3(64), bytes, st, ka);
        return (int)hipGetLastError();
    }
    int gp = 1;
    while (gp < ka.nx) gp <<= 1;
    const int per = 64 / gp;
    const unsigned grid = (unsigned)((batch + per - 1) / per);
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL(mpcqp_rollout_kernel<double>, dim3(grid), dim3(64), 0, st, ka, gp, batch);
    else
        hipLaunchKernelGGL(mpcqp_rollout_kernel<float>, dim3(grid), dim3(64), 0, st, ka, gp, batch);
    return (int)hipGetLastError();
}

}  // namespace mpcqp
