/*
 * ORACLE -- test infrastructure only. Never linked, imported or called by the
 * product path (qpmpc_amd/); only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * Plain-C float64 restatement of the reference hot path
 *   MPCProblem -> MPCQP (condense) -> QP solve -> inputs
 * (reference: qpmpc/solve_mpc.py:42-44).
 *
 * Build half  : follows qpmpc/mpc_qp.py:53-114 (Phi/Psi propagation, G_k, h_k,
 *               P) and :129-149 (q). PINNED against tests/golden/ (captured
 *               from the real reference by tools/gen_golden.py) and against
 *               oracle/condense_np.py in tests/test_oracle.py.
 *
 * Solve half  : the reference delegates to the third-party package
 *               `qpsolvers` (pyproject.toml:29-31: ">=1.8.0", no lock file, so
 *               the version is unpinned) and one of its backends; BASELINE.json
 *               names `quadprog`, i.e. the dual active-set method of
 *               D. Goldfarb and A. Idnani, "A numerically stable dual method
 *               for solving strictly convex quadratic programs", Math. Prog.
 *               27 (1983). Neither package is present in /root/reference nor
 *               installable here, so this file restates the PUBLISHED
 *               algorithm (Cholesky P = L L^T, J = L^-T, R updated by Givens
 *               rotations, most-violated-constraint rule, partial/full steps).
 *               PARITY UNPINNED at the solver boundary: the reference's tests
 *               hold no numeric solution except the known answer U* = 0 of
 *               tests/test_wheeled_inverted_pendulum.py:23-41. It is anchored
 *               instead on (i) that known answer, (ii) uniqueness of the
 *               minimiser (P >= w_u I > 0, mpc_problem.py:104-107) and (iii)
 *               the SLSQP+KKT-certified solutions of the reference-built QPs
 *               stored in tests/golden/.
 *
 * Layout: all matrices row-major, contiguous. Constraints are G u <= h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FLAG_P_TERMINAL 1 /* terminal weight "is not None"  (mpc_qp.py:102) */
#define FLAG_P_STAGE 2    /* stage weight "is not None"     (mpc_qp.py:104) */
#define FLAG_Q_TERMINAL 4 /* has_terminal_cost              (mpc_problem.py:141-153) */
#define FLAG_Q_STAGE 8    /* has_stage_state_cost           (mpc_problem.py:155-166) */

/* ------------------------------------------------------------------ build */

/* Psi_all: (N+1) blocks of nx x n; block k is Psi_k (block 0 is zero), block N
 * is psi_last. xf: (N+1) x nx, xf_k = Phi_k x0. Phi_all (optional): (N+1) blocks
 * of nx x nx. */
void oracle_condense_one(int nx, int nu, int N, int mk,
                         const double *A, int64_t sA, const double *B, int64_t sB,
                         const double *C, int64_t sC, const double *D, int64_t sD,
                         const double *e, int64_t se, const double *x0,
                         const double *goal, const double *targets, int flags,
                         double wt, double wx, double wu, double *P, double *q,
                         double *G, double *h, double *Psi_all, double *xf,
                         double *Phi_all)
{
    const int n = N * nu;
    const int m = N * mk;
    memset(Psi_all, 0, sizeof(double) * (size_t)(N + 1) * nx * n);
    /* mpc_qp.py:53 phi = I ; :54 psi = 0 */
    double *phi = (double *)calloc((size_t)nx * nx, sizeof(double));
    double *phin = (double *)calloc((size_t)nx * nx, sizeof(double));
    for (int i = 0; i < nx; ++i) phi[i * nx + i] = 1.0;
    for (int i = 0; i < nx; ++i) xf[i] = x0[i];
    for (int k = 0; k < N; ++k) {
        const double *Ak = A + k * sA, *Bk = B + k * sB;
        const double *psi = Psi_all + (size_t)k * nx * n;
        double *psin = Psi_all + (size_t)(k + 1) * nx * n;
        if (Phi_all) memcpy(Phi_all + (size_t)k * nx * nx, phi, sizeof(double) * nx * nx);
        /* :67-78 G_k = [D_k in block k] + C_k psi ; :68-72 h_k = e_k - (C_k phi) x0 */
        for (int i = 0; i < mk; ++i) {
            double *Gi = G + (size_t)(k * mk + i) * n;
            for (int c = 0; c < n; ++c) Gi[c] = 0.0;
            if (D)
                for (int c = 0; c < nu; ++c) Gi[k * nu + c] = D[k * sD + i * nu + c];
            double hk = e[k * se + i];
            if (C) {
                const double *Ci = C + k * sC + (size_t)i * nx;
                for (int c = 0; c < n; ++c) {
                    double acc = 0.0;
                    for (int s = 0; s < nx; ++s) acc += Ci[s] * psi[s * n + c];
                    Gi[c] += acc;
                }
                /* (C_k phi) x0, same association as the reference */
                double acc = 0.0;
                for (int c = 0; c < nx; ++c) {
                    double cp = 0.0;
                    for (int s = 0; s < nx; ++s) cp += Ci[s] * phi[s * nx + c];
                    acc += cp * x0[c];
                }
                hk -= acc;
            }
            h[k * mk + i] = hk;
        }
        /* :88 phi = A_k phi ; :89 psi = A_k psi ; :90 psi[:, block k] = B_k */
        for (int r = 0; r < nx; ++r)
            for (int c = 0; c < nx; ++c) {
                double acc = 0.0;
                for (int s = 0; s < nx; ++s) acc += Ak[r * nx + s] * phi[s * nx + c];
                phin[r * nx + c] = acc;
            }
        memcpy(phi, phin, sizeof(double) * nx * nx);
        for (int r = 0; r < nx; ++r) {
            for (int c = 0; c < n; ++c) {
                double acc = 0.0;
                for (int s = 0; s < nx; ++s) acc += Ak[r * nx + s] * psi[s * n + c];
                psin[r * n + c] = acc;
            }
            for (int c = 0; c < nu; ++c) psin[r * n + k * nu + c] = Bk[r * nu + c];
        }
        /* free response, used for q below: xf_{k+1} = Phi_{k+1} x0 */
        for (int r = 0; r < nx; ++r) {
            double acc = 0.0;
            for (int c = 0; c < nx; ++c) acc += phi[r * nx + c] * x0[c];
            xf[(k + 1) * nx + r] = acc;
        }
    }
    if (Phi_all) memcpy(Phi_all + (size_t)N * nx * nx, phi, sizeof(double) * nx * nx);
    /* :99-105 P = wu I + wt psi_N^T psi_N + wx Psi^T Psi */
    const double *psiN = Psi_all + (size_t)N * nx * n;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double acc = (a == b) ? wu : 0.0;
            if (flags & FLAG_P_TERMINAL) {
                double t = 0.0;
                for (int s = 0; s < nx; ++s) t += psiN[s * n + a] * psiN[s * n + b];
                acc += wt * t;
            }
            if (flags & FLAG_P_STAGE) {
                double t = 0.0;
                for (int s = 0; s < N * nx; ++s) t += Psi_all[(size_t)s * n + a] * Psi_all[(size_t)s * n + b];
                acc += wx * t;
            }
            P[a * n + b] = acc;
        }
    /* :139-149 q */
    for (int a = 0; a < n; ++a) q[a] = 0.0;
    if (flags & FLAG_Q_TERMINAL)
        for (int a = 0; a < n; ++a) {
            double t = 0.0;
            for (int s = 0; s < nx; ++s) t += (xf[N * nx + s] - goal[s]) * psiN[s * n + a];
            q[a] += wt * t;
        }
    if (flags & FLAG_Q_STAGE)
        for (int a = 0; a < n; ++a) {
            double t = 0.0;
            for (int s = 0; s < N * nx; ++s) t += (xf[s] - targets[s]) * Psi_all[(size_t)s * n + a];
            q[a] += wx * t;
        }
    free(phi);
    free(phin);
    (void)m;
}

/* X[0]=x0, X[k+1]=A_k X[k]+B_k U[k]   (mpc_problem.py:316-335) */
void oracle_rollout_one(int nx, int nu, int N, const double *A, int64_t sA,
                        const double *B, int64_t sB, const double *x0,
                        const double *U, double *X)
{
    for (int i = 0; i < nx; ++i) X[i] = x0[i];
    for (int k = 0; k < N; ++k)
        for (int r = 0; r < nx; ++r) {
            double acc = 0.0;
            for (int s = 0; s < nx; ++s) acc += A[k * sA + r * nx + s] * X[k * nx + s];
            for (int c = 0; c < nu; ++c) acc += B[k * sB + r * nu + c] * U[k * nu + c];
            X[(k + 1) * nx + r] = acc;
        }
}

/* ------------------------------------------------------------------ solve */
/* Goldfarb-Idnani dual active-set method for
 *     min 1/2 x'Px + q'x   s.t.  G x <= h ,  P symmetric positive definite.
 * In the paper's notation the constraints are n_i' x >= b_i with
 * n_i = -G_i, b_i = -h_i, slack s_i = n_i'x - b_i = h_i - G_i x.
 * status: 0 solved, 1 iteration limit, 2 infeasible, 3 P not positive definite.
 */
static double hyp(double a, double b) { return hypot(a, b); }

/* Iterative refinement of the final active set's KKT system at (x, u): see the comment at the end of oracle_gi_solve.
 * r: scratch of iq + 1 doubles. */
static void refine_active(int n, int iq, const int *act, const double *P, const double *qv, const double *G, const double *h,
                          const double *J, const double *R, double *x, double *u, double *r)
{
    long double *r1 = (long double *)calloc((size_t)n, sizeof(long double));
    long double *r2 = (long double *)calloc((size_t)(iq > 0 ? iq : 1), sizeof(long double));
    double *w = (double *)calloc((size_t)n, sizeof(double));
    double *v = (double *)calloc((size_t)(iq > 0 ? iq : 1), sizeof(double));
    double *xs = (double *)calloc((size_t)n, sizeof(double));
    double *us = (double *)calloc((size_t)(iq > 0 ? iq : 1), sizeof(double));
    long double best = INFINITY;
    for (int pass = 0; pass < 4 && iq > 0; ++pass) {
        /* residuals of the active set's KKT system at (x, u), extended precision */
        long double worst = 0.0L;
        for (int k = 0; k < n; ++k) {
            long double acc = (long double)qv[k];
            for (int j = 0; j < n; ++j) acc += (long double)P[(size_t)k * n + j] * (long double)x[j];
            for (int i = 0; i < iq; ++i) acc += (long double)G[(size_t)act[i] * n + k] * (long double)u[i];
            r1[k] = -acc;
        }
        for (int i = 0; i < iq; ++i) {
            const int a = act[i];
            long double acc = (long double)h[a];
            for (int k = 0; k < n; ++k) acc -= (long double)G[(size_t)a * n + k] * (long double)x[k];
            r2[i] = acc;
            const long double rel = fabsl(acc) / (1.0L + fabsl((long double)h[a]));
            if (rel > worst) worst = rel;
        }
        if (!(worst < best)) {  /* no longer improving (or not a number): keep the better point */
            if (pass > 0) {
                for (int k = 0; k < n; ++k) x[k] = xs[k];
                for (int i = 0; i < iq; ++i) u[i] = us[i];
            }
            break;
        }
        best = worst;
        if (pass == 3 || worst == 0.0L) break;
        for (int k = 0; k < n; ++k) xs[k] = x[k];
        for (int i = 0; i < iq; ++i) us[i] = u[i];
        for (int j = 0; j < n; ++j) {  /* w = J' r1 */
            long double acc = 0.0L;
            for (int k = 0; k < n; ++k) acc += (long double)J[k * n + j] * r1[k];
            w[j] = (double)acc;
        }
        for (int i = 0; i < iq; ++i) {  /* R' v = r2 */
            long double acc = r2[i];
            for (int j = 0; j < i; ++j) acc -= (long double)R[j * n + i] * (long double)v[j];
            v[i] = (double)(acc / (long double)R[i * n + i]);
        }
        for (int k = 0; k < n; ++k) {  /* dx = J2 w2 - J1 v */
            long double acc = 0.0L;
            for (int j = 0; j < iq; ++j) acc -= (long double)J[k * n + j] * (long double)v[j];
            for (int j = iq; j < n; ++j) acc += (long double)J[k * n + j] * (long double)w[j];
            x[k] = (double)((long double)x[k] + acc);
        }
        for (int i = iq - 1; i >= 0; --i) {  /* du = -R^-1 (w1 + v), into r (free now) */
            long double acc = -((long double)w[i] + (long double)v[i]);
            for (int j = i + 1; j < iq; ++j) acc -= (long double)R[i * n + j] * (long double)r[j];
            r[i] = (double)(acc / (long double)R[i * n + i]);
        }
        for (int i = 0; i < iq; ++i) u[i] += r[i];
    }
    free(r1); free(r2); free(w); free(v); free(xs); free(us);
}

int oracle_gi_solve(int n, int m, const double *P, const double *qv,
                    const double *G, const double *h, int max_iter, double tol,
                    double *x, double *lam, int *iters_out)
{
    int status = 1, iters = 0;
    double *L = (double *)calloc((size_t)n * n, sizeof(double));
    double *J = (double *)calloc((size_t)n * n, sizeof(double));
    double *R = (double *)calloc((size_t)n * n, sizeof(double));
    double *d = (double *)calloc(n, sizeof(double));
    double *z = (double *)calloc(n, sizeof(double));
    double *r = (double *)calloc(n + 1, sizeof(double));
    double *u = (double *)calloc(n + 1, sizeof(double));
    double *np = (double *)calloc(n, sizeof(double));
    int *act = (int *)calloc(n + 1, sizeof(int));
    char *is_active = (char *)calloc(m > 0 ? m : 1, 1);
    int iq = 0, refined = 0;

    /* Step 0a: Cholesky P = L L' (lower) */
    for (int j = 0; j < n; ++j) {
        double sjj = P[j * n + j];
        for (int k = 0; k < j; ++k) sjj -= L[j * n + k] * L[j * n + k];
        if (!(sjj > 0.0)) { status = 3; goto done; }
        L[j * n + j] = sqrt(sjj);
        for (int i = j + 1; i < n; ++i) {
            double sij = P[i * n + j];
            for (int k = 0; k < j; ++k) sij -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = sij / L[j * n + j];
        }
    }
    /* Step 0b: J = L^-T (upper triangular): column c solves L' J[:,c] = e_c */
    for (int c = 0; c < n; ++c)
        for (int i = n - 1; i >= 0; --i) {
            double acc = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k < n; ++k) acc -= L[k * n + i] * J[k * n + c];
            J[i * n + c] = acc / L[i * n + i];
        }
    /* Step 0c: unconstrained minimiser x = -P^-1 q = -J J' q */
    for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += J[k * n + j] * qv[k];
        d[j] = acc;
    }
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += J[i * n + j] * d[j];
        x[i] = -acc;
    }

    while (1) {
        /* Step 1: most violated inactive constraint */
        int p = -1;
        double worst = -tol, sp = 0.0;
        for (int i = 0; i < m; ++i) {
            if (is_active[i]) continue;
            double s = h[i];
            for (int k = 0; k < n; ++k) s -= G[(size_t)i * n + k] * x[k];
            double key = s / (1.0 + fabs(h[i]));
            if (key < worst) { worst = key; p = i; sp = s; }
        }
        if (p < 0) {
            /* no inactive row is violated: refine the point on its active set once (round 6, see the end of this function) and look
               again -- a row that was feasible by less than the refinement moved the point is taken up like any other */
            if (iq > 0 && !refined) {
                refine_active(n, iq, act, P, qv, G, h, J, R, x, u, r);
                for (int i = 0; i < iq; ++i)
                    if (u[i] < 0.0) u[i] = 0.0; /* (rounding level: a weakly active row) */
                refined = 1;
                continue;
            }
            status = 0;
            break;
        }
        refined = 0;
        for (int k = 0; k < n; ++k) np[k] = -G[(size_t)p * n + k];
        u[iq] = 0.0;

        int added = 0;
        while (!added) {
            if (iters++ >= max_iter) { status = 1; goto done; }
            /* Step 2a: d = J' n+ ; z = J2 d2 ; r = R^-1 d1 */
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int k = 0; k < n; ++k) acc += J[k * n + j] * np[k];
                d[j] = acc;
            }
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int j = iq; j < n; ++j) acc += J[i * n + j] * d[j];
                z[i] = acc;
            }
            for (int i = iq - 1; i >= 0; --i) {
                double acc = d[i];
                for (int j = i + 1; j < iq; ++j) acc -= R[i * n + j] * r[j];
                r[i] = acc / R[i * n + i];
            }
            /* Step 2b: step lengths */
            double t1 = INFINITY, t2 = INFINITY;
            int l = -1;
            for (int i = 0; i < iq; ++i)
                if (r[i] > 0.0 && u[i] / r[i] < t1) { t1 = u[i] / r[i]; l = i; }
            double zz = 0.0, ztn = 0.0, nn = 0.0;
            for (int k = 0; k < n; ++k) { zz += z[k] * z[k]; ztn += z[k] * np[k]; nn += np[k] * np[k]; }
            /* z == 0 (n+ in the span of the active normals): compare with |J'n+|^2 scale. The threshold sits at
               rounding-noise level: on INCONSISTENT problems a dependent row can slip through, the step along it has
               length ~1e13 and the loop ends 'solved' with the ACTIVE rows far off their bounds (step 1 only looks at
               inactive rows). The pivot test is kept as it is -- this is the pinned checker, and a tighter test
               (1e-14) cost 2e-7 of accuracy on one legitimate borderline problem -- and such an end is caught by the
               acceptance check after the loop instead. */
            double dd = 0.0;
            for (int j = 0; j < n; ++j) dd += d[j] * d[j];
            if (iq < n && ztn > 1e-28 * (dd > 0 ? dd : 1.0) && zz > 0.0) t2 = -sp / ztn;
            double t = t1 < t2 ? t1 : t2;
            (void)nn;
            if (!isfinite(t)) { status = 2; goto done; }
            /* Step 2c */
            if (isfinite(t2))
                for (int k = 0; k < n; ++k) x[k] += t * z[k];
            for (int i = 0; i < iq; ++i) u[i] -= t * r[i];
            u[iq] += t;
            if (t2 <= t1) {
                /* full step: add constraint p. Givens sweep reduces d2 to one entry. */
                for (int j = n - 1; j > iq; --j) {
                    double a = d[j - 1], b = d[j];
                    double hh = hyp(a, b);
                    if (hh == 0.0) continue;
                    double c = a / hh, s = b / hh;
                    d[j - 1] = hh;
                    d[j] = 0.0;
                    for (int k = 0; k < n; ++k) {
                        double t1k = J[k * n + j - 1], t2k = J[k * n + j];
                        J[k * n + j - 1] = c * t1k + s * t2k;
                        J[k * n + j] = -s * t1k + c * t2k;
                    }
                }
                for (int i = 0; i <= iq; ++i) R[i * n + iq] = d[i];
                act[iq] = p;
                is_active[p] = 1;
                ++iq;
                added = 1;
            } else {
                /* partial step: drop the blocking constraint at position l */
                is_active[act[l]] = 0;
                for (int j = l; j < iq - 1; ++j) {
                    act[j] = act[j + 1];
                    u[j] = u[j + 1];
                    for (int i = 0; i <= j + 1; ++i) R[i * n + j] = R[i * n + j + 1];
                }
                u[iq - 1] = u[iq];
                u[iq] = 0.0;
                for (int i = 0; i < iq; ++i) R[i * n + iq - 1] = 0.0;
                --iq;
                /* restore triangular R with row rotations; mirror on J's columns */
                for (int j = l; j < iq; ++j) {
                    double a = R[j * n + j], b = R[(j + 1) * n + j];
                    double hh = hyp(a, b);
                    if (hh == 0.0) continue;
                    double c = a / hh, s = b / hh;
                    R[j * n + j] = hh;
                    R[(j + 1) * n + j] = 0.0;
                    for (int k = j + 1; k < iq; ++k) {
                        double t1k = R[j * n + k], t2k = R[(j + 1) * n + k];
                        R[j * n + k] = c * t1k + s * t2k;
                        R[(j + 1) * n + k] = -s * t1k + c * t2k;
                    }
                    for (int k = 0; k < n; ++k) {
                        double t1k = J[k * n + j], t2k = J[k * n + j + 1];
                        J[k * n + j] = c * t1k + s * t2k;
                        J[k * n + j + 1] = -s * t1k + c * t2k;
                    }
                }
                if (isfinite(t2)) {
                    sp = h[p];
                    for (int k = 0; k < n; ++k) sp -= G[(size_t)p * n + k] * x[k];
                }
            }
        }
    }
    /* Refinement + acceptance (round 6). The loop above ends with every inactive row feasible to tol (1 + |h|), but the point it
       carries is the SUM of its steps: on a nearly fully active problem (a vertex: as many active rows as variables) the active
       rows end 1e-8 off their bounds, and that is 2e-6 in the plan -- looser than the 1e-6 contract this oracle checks (SURVEY 2.1
       asks a "quadprog-class" KKT residual of 1e-9). So the final active set's KKT system
           P dx + G_A' du = r1 = -(P x + q + G_A' u),     G_A dx = r2 = h_A - G_A x
       is solved once more -- up to three steps of iterative refinement with the factors the method holds (J J' = P^-1,
       J' N = [R; 0] with N = -G_A'), the residuals formed in extended precision:
           w = J' r1,  v = R^-T r2,  dx = J2 w2 - J1 v,  du = -R^-1 (w1 + v).
       Acceptance, from scratch: the point is finite, every multiplier is >= 0 (to rounding) and every ACTIVE row sits on its bound
       to 1e-9 (1 + |h_i|) (it was 1e-6 until round 5) -- inactive rows were just checked by step 1. A failure means a dependent row
       of an inconsistent problem slipped through the pivot test above: no feasible point was found -> status 2, what qpsolvers
       reports as found=False (plan.py:35-40). */
    if (status == 0) {
        int bad = 0;
        double umax = 0.0;
        for (int i = 0; i < iq; ++i)
            if (u[i] > umax) umax = u[i];
        for (int k = 0; k < n; ++k)
            if (!isfinite(x[k])) bad = 1;
        for (int i = 0; i < iq && !bad; ++i) {
            const int a = act[i];
            long double s = (long double)h[a];
            for (int k = 0; k < n; ++k) s -= (long double)G[(size_t)a * n + k] * (long double)x[k];
            if (!(fabsl(s) <= 1e-9L * (1.0L + fabsl((long double)h[a])))) bad = 1;
            if (u[i] < 0.0 && u[i] >= -1e-12 * (1.0 + umax)) u[i] = 0.0;  /* a weakly active row's multiplier, at rounding level */
            if (!(u[i] >= 0.0)) bad = 1;
        }
        if (bad) status = 2;
    }
done:
    if (lam) {
        for (int i = 0; i < m; ++i) lam[i] = 0.0;
        if (status == 0)
            for (int i = 0; i < iq; ++i) lam[act[i]] = u[i];
    }
    if (iters_out) *iters_out = iters;
    free(L); free(J); free(R); free(d); free(z); free(r); free(u); free(np);
    free(act); free(is_active);
    return status;
}

/* ------------------------------------------------------------ whole path */
/* One problem per call of the inner body, exactly like solve_mpc.py:42-44.
 * Operands carry an element stride per batch item (0 = shared) and per step
 * (0 = time-invariant). */
typedef struct {
    const double *ptr;
    int64_t batch_stride;
    int64_t step_stride;
} oracle_operand;

int oracle_build_solve_batch(int nx, int nu, int N, int mk, int flags, double wt,
                             double wx, double wu, const oracle_operand *A,
                             const oracle_operand *B, const oracle_operand *C,
                             const oracle_operand *D, const oracle_operand *e,
                             const oracle_operand *x0, const oracle_operand *goal,
                             const oracle_operand *targets, int64_t batch,
                             int max_iter, double tol, double *U, double *lam,
                             int32_t *status, int32_t *iters)
{
    const int n = N * nu, m = N * mk;
    double *P = (double *)malloc(sizeof(double) * n * n);
    double *q = (double *)malloc(sizeof(double) * n);
    double *G = (double *)malloc(sizeof(double) * (m > 0 ? m : 1) * n);
    double *h = (double *)malloc(sizeof(double) * (m > 0 ? m : 1));
    double *Psi = (double *)malloc(sizeof(double) * (size_t)(N + 1) * nx * n);
    double *xf = (double *)malloc(sizeof(double) * (N + 1) * nx);
    for (int64_t b = 0; b < batch; ++b) {
        oracle_condense_one(nx, nu, N, mk, A->ptr + b * A->batch_stride, A->step_stride,
                            B->ptr + b * B->batch_stride, B->step_stride,
                            C->ptr ? C->ptr + b * C->batch_stride : NULL, C->step_stride,
                            D->ptr ? D->ptr + b * D->batch_stride : NULL, D->step_stride,
                            e->ptr + b * e->batch_stride, e->step_stride,
                            x0->ptr + b * x0->batch_stride,
                            goal->ptr ? goal->ptr + b * goal->batch_stride : NULL,
                            targets->ptr ? targets->ptr + b * targets->batch_stride : NULL,
                            flags, wt, wx, wu, P, q, G, h, Psi, xf, NULL);
        int it = 0;
        int st = oracle_gi_solve(n, m, P, q, G, h, max_iter, tol, U + b * n,
                                 lam ? lam + b * m : NULL, &it);
        if (status) status[b] = st;
        if (iters) iters[b] = it;
    }
    free(P); free(q); free(G); free(h); free(Psi); free(xf);
    return 0;
}
