// mpcqp_stageg.hip -- gfx950: stage-wise (uncondensed) solve of the MPC QP for WIDE systems, any horizon.
//
// Replaces the same reference code as the other solver units -- MPCQP.__init__ + qpsolvers.solve_problem, i.e. the whole
// of solve_mpc (qpmpc/solve_mpc.py:42-44), which accepts ANY (nx, nu, N) -- for the dimensions no other kernel of this
// library serves: nx > 16 or nu > 4 (the wide stage-wise kernel's MFMA tiles stop there) together with n = N nu > 256
// (the dense HBM-resident path stops there). Until round 4 such a problem -- a 20-state, 6-input model at N = 64 -- came
// back MPCQP_ETOOLARGE.
//
// Same method as mpcqp_stage.hip / mpcqp_stagew.hip (oracle/stagewise_np.py restates it): Riccati factor of the LQR problem
// whose Hessian is the condensed P (once per problem), v -> P^-1 v as one backward and one forward sweep, a row of G applied
// to a vector as a read of that vector's trajectory, Goldfarb-Idnani's dual active set in the metric of P with
// a thin QR FACTORISATION of the active rows' whitened vectors (round 5; rounds 2-4 kept W = (G_A P^-1 G_A')^-1 by rank-one updates,
// see below); per active row the slot keeps V_a = P^-1 g_a' (inputs) and h_a = G V_a (all m rows), so an iteration is one sweep pair,
// two passes over Q, one triangular solve and AXPYs over m-long arrays.
//
// The active-set operator. With c = G_A V_p and d = g_p V_p the step needs r = S^-1 c and |z|^2 = d - c' S^-1 c, S = G_A P^-1 G_A'.
// An explicit inverse W = S^-1 kept by bordering / deflation computes the latter with an error of eps cond(S) d -- near a full
// active set, cond(S) ~ 1e10 and more, a row that can enter looks dependent (or the other way round) and the loop wanders
// (tools/stress_general.py seed 7, tools/stress_tight.py general seeds 2, 3, 8: MAX_ITER where the oracle solves); a Cholesky
// factor of S gets eps sqrt(cond(S)) d, still a DIFFERENCE. Goldfarb and Idnani's own form has no difference in it: with
// P = L L', y_a = L^-1 g_a' and Y_A = Q R (Q orthonormal), |z|^2 = |y_p - Q Q' y_p|^2 is a sum of squares, good to eps^2 |y_p|^2,
// and r = R^-1 Q' y_p. The Riccati recursion IS a block Cholesky factorisation of P, and the backward sweep already produces the
// whitened vector: with S_k = R + B' P_{k+1} B = Ls_k Ls_k' the stage Hessians and ff_k = -S_k^-1 t_k the feed-forward terms,
// y = (Ls_k^-1 t_k)_k = (-Ls_k' ff_k)_k and g_a P^-1 g_b' = y_a . y_b. So: Q (n x slots, explicit, in the workspace) and R (upper
// triangular) are kept; a candidate is orthogonalised against Q twice (classical Gram-Schmidt with re-orthogonalisation), a new
// row appends a column to R and a vector to Q, a leaving row deletes its column of R and a sweep of Givens rotations over the
// rows of R / the vectors of Q restores the triangle. The whitened vectors Y are kept too: Q, R can be rebuilt from them.
//
// Written for GENERALITY first: float64 only (float32 launches are converted, mpcqp_capi.hip), one workgroup of 256 threads per
// problem, every per-problem array in a caller-owned HBM workspace, the per-step matrices of the recursion (at most 32 x 32)
// in LDS. nx <= 32, nu <= 8, any N, any mk. The 2 N serial steps of a sweep pair are run by ONE wavefront on scalars
// (v_readlane) while the other three stream the factor's per-step blocks into an LDS ring ahead of it (round 4, second
// version: 2.9x the first one on nx = 20, nu = 6, N = 40); what is left per iteration besides the sweeps are the m-row passes
// and the updates of W, handed from thread to thread through the workspace (a barrier and a round trip each).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "mpcqp.h"
#include "mpcqp_internal.h"

#ifndef STAGEG_VPASS
// passes of the final verification: each one re-evaluates every slack from scratch and, while an active row is off its bound,
// corrects the multipliers by S^-1 times the residuals (iterative refinement through the factor)
#define STAGEG_VPASS 6
#endif
#ifndef STAGEG_REFRESH
// iterations between rebuilds of the factor (from the Gram matrix the slots hold), the multipliers and the slacks from scratch
#define STAGEG_REFRESH 1024
#endif
#ifndef STAGEG_ACC
// acceptance: active rows within this (1 + |e_i|) of their bounds on the roll-out of the returned inputs. 1e-6 until the end of round 6
// (the oracle's rule of rounds 1-5); the oracle accepts 1e-9 since its refinement, which this kernel's explicit primal point does not
// reach on nearly fully active problems: at 1e-8 one problem of `stress_tight general` (STRESS_TIGHT 0.5, 128 rounds) that the oracle
// solves ends MPCQP_MAX_ITER; at 1e-7 none does, and the verdicts that differ from the oracle's on the tighter families -- plans
// with |U| ~ 1e7 accepted where the oracle says infeasible -- go from 1 / 3 / 5 to 1 / 2 / 2 rounds of 128 (0.3 / 0.15 / 0.05).
#define STAGEG_ACC 1e-7
#endif
#ifndef STAGEG_DBG
#define STAGEG_DBG 0 /* timing experiments only (wrong results) */
#endif

namespace mpcqp {
namespace stageg {

constexpr int NXM = 32, NUM = 8, BS = 256;

struct Ws {  // per-problem workspace carve in doubles (host-computed, passed by value)
    int64_t Blk, U0, ff, Xt, s0, s, invn, thr, V, H, W, Q, Y, LS, lam, cv, rv, yv, dv, ev, ints, total;
    int maxq;
};

__host__ __device__ inline int block_doubles(int nx, int nu) { return (nx * nx + 2 * nx * nu + nu * nu + 1) & ~1; }

inline Ws make_ws(int nx, int nu, int N, int mk, int maxq)
{
    Ws w{};
    int64_t o = 0;
    auto take = [&](int64_t c) {
        const int64_t at = o;
        o += (c + 1) & ~(int64_t)1;
        return at;
    };
    const int64_t n = (int64_t)N * nu, m = (int64_t)N * mk;
    // the factor, one packed block per step: [Acl' nx x nx | B nx x nu | S^-1 nu x nu | K nu x nx] -- what a sweep step reads, in the
    // order in which the sweeps' loaders copy whole runs of steps into LDS
    w.Blk = take((int64_t)N * block_doubles(nx, nu));
    w.U0 = take(n);
    w.ff = take(n);
    w.Xt = take((int64_t)N * nx);  // state trajectory of the latest forward sweep (the rows of G are applied to it afterwards)
    w.s0 = take(2);  // (unused since round 5)
    w.s = take(m);
    w.invn = take(m);
    w.thr = take(m);
    w.V = take((int64_t)2 * n);  // the step in the inputs z_u, a roll-out's scratch inputs
    w.H = take((int64_t)2 * m);  // G z of the step, a roll-out's rows
    w.W = take((int64_t)(maxq + 1) * maxq);  // R (upper triangular) by rows: row j = basis vector j, column b = slot b
    w.Q = take((int64_t)(maxq + 1) * n);     // Q by vectors: vector j at j n (vector nq: the candidate's projection)
    w.Y = take(n);                           // the candidate's whitened vector y_p = L^-1 g_p'
    w.LS = take((int64_t)N * nu * nu);       // Cholesky factors of the stage Hessians S_k (the recursion)
    w.lam = take(maxq + 1);
    w.cv = take(maxq + 1);
    w.rv = take(maxq + 1);
    w.yv = take(maxq + 1);  // vectors of the triangular solves when the slots outgrow their LDS copies
    w.dv = take(maxq + 1);
    w.ev = take(maxq + 1);
    w.ints = take((m + 2 * (maxq + 2)) / 2 + 2);  // int32: pos[m], actrow[maxq + 1], phys[maxq + 1]
    w.total = (o + 15) & ~(int64_t)15;
    w.maxq = maxq;
    return w;
}

__device__ __forceinline__ void bsync() { __syncthreads(); }

// block-wide arg-min of (v, i): ties -> lowest index; every thread gets the result
__device__ __forceinline__ void block_argmin(double &v, int &i, double *redv, int *redi, int tid)
{
    wave_argmin_dpp(v, i);
    if ((tid & 63) == 0) {
        redv[tid >> 6] = v;
        redi[tid >> 6] = i;
    }
    bsync();
    v = redv[0];
    i = redi[0];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w) {
        const double ov = redv[w];
        const int oi = redi[w];
        if (ov < v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
    }
    bsync();
}
__device__ __forceinline__ double block_sum(double v, double *redv, int tid)
{
    v = wave_sum_dpp(v);
    if ((tid & 63) == 0) redv[tid >> 6] = v;
    bsync();
    double s = redv[0];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w) s += redv[w];
    bsync();
    return s;
}
__device__ __forceinline__ bool block_any(bool p, int *redi, int tid)
{
    const bool w = __ballot(p) != 0ull;
    if ((tid & 63) == 0) redi[tid >> 6] = w;
    bsync();
    bool r = false;
#pragma unroll
    for (int k = 0; k < BS / 64; ++k) r = r || redi[k];
    bsync();
    return r;
}

}  // namespace stageg

using namespace stageg;

__global__ void __launch_bounds__(BS, 2) mpcqp_stageg_kernel(const KernelArgs ka, const Ws wl, double *__restrict__ wsbase, int ring_doubles)
{
    using T = double;
    // The matrices of the recursion and the sweeps' ring are never alive together: both are carved from the SAME dynamic LDS
    // (64 KB; a workgroup's own 4 KB of static LDS next to it -- two workgroups per CU at the kernel's 235 VGPRs)
    extern __shared__ __attribute__((aligned(16))) unsigned char stageg_dyn[];
    T *Pm = (T *)stageg_dyn, *PAm = Pm + NXM * NXM, *Tm = PAm + NXM * NXM, *Am = Tm + NXM * NXM, *PBm = Am + NXM * NXM;
    T *Bm = PBm + NXM * NUM, *G1 = Bm + NXM * NUM, *Km = G1 + NUM * NXM, *Sm = Km + NUM * NXM, *Sim = Sm + NUM * NUM, *Ga = Sim + NUM * NUM;
    static_assert(4 * NXM * NXM + 4 * NXM * NUM + 2 * NUM * NUM + NUM * 2 * NUM <= 6144, "the recursion's matrices fit the ring");
    __shared__ T redv[BS / 64];
    constexpr int LQ = 256;  // slots whose coefficient / place are staged in LDS for the m-row passes
    __shared__ T rvl[LQ];
    __shared__ int physl[LQ];
    __shared__ int redi[BS / 64], flag;
    __shared__ T yl_s[LQ], dl_s[LQ], el_s[LQ], cs_s[4];  // d = Q' y, 1 / R_jj, the re-orthogonalisation's correction, a rotation's (c, s)
    const int tid = threadIdx.x;
    const int64_t prob = blockIdx.x;
    const int nx = ka.nx, nu = ka.nu, N = ka.N, mk = ka.mk, maxq = wl.maxq;
    const int n = N * nu, M = N * mk;
    const T INF = HUGE_VAL;
    T *ws = wsbase + prob * wl.total;
    T *Blk = ws + wl.Blk, *U0 = ws + wl.U0, *ffv = ws + wl.ff, *Xt = ws + wl.Xt, *s0 = ws + wl.s0, *sl = ws + wl.s;
    T *invn = ws + wl.invn, *thr = ws + wl.thr, *Vs = ws + wl.V, *Hs = ws + wl.H, *Wm = ws + wl.W, *lamv = ws + wl.lam;
    T *cv = ws + wl.cv, *rv = ws + wl.rv;
    T *Rm = Wm, *Qs = ws + wl.Q, *Ys = ws + wl.Y;  // Y_A = Q R: R by rows (row j at j maxq), Q by vectors (vector j at j n)
    // the vectors over the slots: in LDS while the launch's slots fit there, in the workspace beyond (maxq <= NRM BS)
    constexpr int NRM = 4;  // rows / columns of R per thread: maxq <= 1024 (launch_stageg)
    const bool vlds = maxq <= LQ;
    T *yv = vlds ? yl_s : ws + wl.yv, *dv = vlds ? dl_s : ws + wl.dv, *ev = vlds ? el_s : ws + wl.ev, *ro = vlds ? rvl : rv;
    int *pos = (int *)(ws + wl.ints), *actrow = pos + M, *phys = actrow + maxq + 1;
    const T *gA = (const T *)ka.A.ptr + prob * ka.A.batch_stride;
    const T *gB = (const T *)ka.B.ptr + prob * ka.B.batch_stride;
    const T *gC = ka.C.ptr ? (const T *)ka.C.ptr + prob * ka.C.batch_stride : nullptr;
    const T *gD = ka.D.ptr ? (const T *)ka.D.ptr + prob * ka.D.batch_stride : nullptr;
    const T *ge = (const T *)ka.e.ptr + prob * ka.e.batch_stride;
    const T *gx0 = (const T *)ka.x0.ptr + prob * ka.x0.batch_stride;
    const T *ggoal = ka.goal.ptr ? (const T *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
    const T *gtgt = ka.targets.ptr ? (const T *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
    const int64_t sA = ka.A.step_stride, sB = ka.B.step_stride, sC = ka.C.step_stride, sD = ka.D.step_stride, sE = ka.e.step_stride;
    const bool stageP = ka.flags & MPCQP_P_STAGE, termP = ka.flags & MPCQP_P_TERMINAL;
    const bool stageQ = (ka.flags & MPCQP_Q_STAGE) && gtgt, termQ = (ka.flags & MPCQP_Q_TERMINAL) && ggoal;
    const T wu = ka.wu, wx = stageP ? ka.wx : 0.0, wt = termP ? ka.wt : 0.0, tol = ka.tol;

    const int oB = nx * nx, oS = oB + nx * nu, oK = oS + nu * nu, PSW = block_doubles(nx, nu);  // a step's block (make_ws)
    const long long t_start = (long long)__builtin_readcyclecounter();
    // ================================================================= factor: Riccati recursion (oracle/stagewise_np.py::Riccati)
    for (int i = tid; i < nx * nx; i += BS) Pm[i] = (i / nx == i % nx) ? wt : 0.0;
    if (tid == 0) flag = 0;
    bsync();
    for (int k = N - 1; k >= 0; --k) {
        const T *A = gA + k * sA, *B = gB + k * sB;
        for (int i = tid; i < nx * nx; i += BS) Am[i] = A[i];
        for (int i = tid; i < nx * nu; i += BS) Bm[i] = B[i];
        bsync();
        for (int e = tid; e < nx * nx; e += BS) {  // PA = P A
            const int i = e / nx, j = e - i * nx;
            T acc = 0.0;
            for (int l = 0; l < nx; ++l) acc += Pm[i * nx + l] * Am[l * nx + j];
            PAm[e] = acc;
        }
        for (int e = tid; e < nx * nu; e += BS) {  // PB = P B
            const int i = e / nu, j = e - i * nu;
            T acc = 0.0;
            for (int l = 0; l < nx; ++l) acc += Pm[i * nx + l] * Bm[l * nu + j];
            PBm[e] = acc;
        }
        bsync();
        for (int e = tid; e < nu * nu; e += BS) {  // S = w_u I + B' P B
            const int a = e / nu, b = e - a * nu;
            T acc = (a == b) ? wu : 0.0;
            for (int l = 0; l < nx; ++l) acc += Bm[l * nu + a] * PBm[l * nu + b];
            Sm[e] = acc;
        }
        for (int e = tid; e < nu * nx; e += BS) {  // G1 = B' P A
            const int a = e / nx, j = e - a * nx;
            T acc = 0.0;
            for (int l = 0; l < nx; ++l) acc += Bm[l * nu + a] * PAm[l * nx + j];
            G1[e] = acc;
        }
        bsync();
        // Ls_k with S_k = Ls Ls' (lower, nu <= 8: one thread; whitens the feed-forward terms of the sweeps, see the header)
        if (tid == 64) {
            T *ls = ws + wl.LS + (int64_t)k * nu * nu;
            for (int j = 0; j < nu; ++j) {
                T dsum = Sm[j * nu + j];
                for (int c2 = 0; c2 < j; ++c2) dsum -= ls[j * nu + c2] * ls[j * nu + c2];
                const T dj = dsum > 0.0 ? sqrt(dsum) : 0.0;  // (a non-positive pivot is reported by the inversion below)
                ls[j * nu + j] = dj;
                const T idj = dj > 0.0 ? 1.0 / dj : 0.0;
                for (int i = j + 1; i < nu; ++i) {
                    T v2 = Sm[i * nu + j];
                    for (int c2 = 0; c2 < j; ++c2) v2 -= ls[i * nu + c2] * ls[j * nu + c2];
                    ls[i * nu + j] = v2 * idj;
                }
                for (int i = 0; i < j; ++i) ls[i * nu + j] = 0.0;
            }
        }
        // S^-1 by Gauss-Jordan on [S | I] in LDS, one thread per entry (nu <= 8; S is a Schur complement of the condensed
        // Hessian: pivots must be positive). (Round 4: it was ONE thread on a private array -- 70 us per step in scratch memory.)
        {
            const int gr_ = tid / (2 * NUM), gc_ = tid - gr_ * 2 * NUM;  // entry (gr_, gc_) of the nu x 2 nu tableau, tid < 128
            const bool mine = tid < 2 * NUM * NUM && gr_ < nu && gc_ < 2 * nu;
            if (mine) Ga[gr_ * 2 * NUM + gc_] = gc_ < nu ? Sm[gr_ * nu + gc_] : (gc_ - nu == gr_ ? 1.0 : 0.0);
            bsync();
            bool bad = false;
            for (int c = 0; c < nu; ++c) {
                const T piv = Ga[c * 2 * NUM + c];
                if (!(piv > 0.0)) {
                    bad = true;  // (uniform: every thread reads the same pivot)
                    break;
                }
                const T prow = mine ? Ga[c * 2 * NUM + gc_] / piv : 0.0, fcol = mine ? Ga[gr_ * 2 * NUM + c] : 0.0;
                bsync();
                if (mine) Ga[gr_ * 2 * NUM + gc_] = gr_ == c ? prow : Ga[gr_ * 2 * NUM + gc_] - fcol * prow;
                bsync();
            }
            if (bad && tid == 0) flag = 1;
            if (tid < nu * nu) Sim[tid] = bad ? 0.0 : Ga[(tid / nu) * 2 * NUM + nu + tid % nu];
        }
        bsync();
        T *blkk = Blk + (int64_t)k * PSW;
        for (int e = tid; e < nu * nu; e += BS) blkk[oS + e] = Sim[e];
        for (int e = tid; e < nx * nu; e += BS) blkk[oB + e] = Bm[e];
        for (int e = tid; e < nu * nx; e += BS) {  // K = S^-1 B' P A
            const int a = e / nx, j = e - a * nx;
            T acc = 0.0;
            for (int b = 0; b < nu; ++b) acc += Sim[a * nu + b] * G1[b * nx + j];
            Km[e] = acc;
            blkk[oK + e] = acc;
        }
        bsync();
        for (int e = tid; e < nx * nx; e += BS) {  // Acl = A - B K ; T = P Acl = PA - PB K
            const int i = e / nx, j = e - i * nx;
            T a1 = Am[e], a2 = PAm[e];
            for (int a = 0; a < nu; ++a) {
                a1 -= Bm[i * nu + a] * Km[a * nx + j];
                a2 -= PBm[i * nu + a] * Km[a * nx + j];
            }
            blkk[j * nx + i] = a1;  // (transposed: the backward sweep reads columns)
            Tm[e] = a2;
        }
        bsync();
        for (int e = tid; e < nx * nx; e += BS) {  // Pn = Q_k + A' P Acl (x_0 is data: Q_0 = 0)
            const int i = e / nx, j = e - i * nx;
            T acc = (i == j && k >= 1) ? wx : 0.0;
            for (int l = 0; l < nx; ++l) acc += Am[l * nx + i] * Tm[l * nx + j];
            PAm[e] = acc;
        }
        bsync();
        for (int e = tid; e < nx * nx; e += BS) {
            const int i = e / nx, j = e - i * nx;
            Pm[e] = 0.5 * (PAm[e] + PAm[j * nx + i]);
        }
        bsync();
    }
    const bool notpd = flag != 0;
    // (developer probe, MpcqpSolveOpts.probe: cycles of the whole problem [0], of the recursion [1], of the sweeps [2], their number [3])
    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 16 : nullptr;
    const long long t_ric = (long long)__builtin_readcyclecounter();
    long long t_sweeps = 0, n_sweeps = 0, t_bwd = 0, t_fwd = 0, t_rows = 0, t_w0 = 0, t_ortho = 0, t_solve = 0, t_drop = 0, t_whiten = 0;

    // ---- one LQR solve: backward sweep from stage kp (row right-hand side) or from N (tracking terms), forward sweep from
    //      x_start; writes the inputs to Vout [n] and G (x, u) to Hout [M]
    // (round 4, second version) The 2 N serial steps are run by ONE wavefront, without a barrier per step: lane l < nx owns
    // component l of the running vector and the row of the step's state matrix that produces it, lanes 32 .. 32 + nu - 1 own
    // the input-sized rows; the vector reaches every lane as scalars (v_readlane of the register that holds it, one component
    // at a time, feeding FMAs with a scalar operand). The step's matrices do not fit a CU for a whole horizon (N nx^2 doubles),
    // and a step is shorter than a round trip to HBM: the OTHER three wavefronts stream the next RG steps' blocks
    // ([Acl' | B | S^-1 | K | targets | ff], as they lie in memory) into one half of an LDS ring while the first wavefront works
    // through the half loaded before -- one workgroup barrier per RG steps. The rows of G are applied afterwards, by all
    // 256 threads, to the stored trajectory. Rows are read with CLAMPED indices instead of being zero-padded: an entry beyond
    // nx / nu multiplies a component that is zero. (First version: 256 threads, one role per wavefront, 2-3 barriers and one
    // exposed memory round trip per step: 4.7 k cycles per step; this one: ~1 k.)
    T *ring = (T *)stageg_dyn;
    // a half of the ring: RG blocks as they lie in the workspace (one flat copy), then RG x (nx targets + nu feed-forward terms)
    const int XS = nx + nu, oT = 0, oF = nx;
    const int RG = ring_doubles / (2 * (PSW + XS)) < 1 ? 1 : (ring_doubles / (2 * (PSW + XS)) > 16 ? 16 : ring_doubles / (2 * (PSW + XS)));
    const int HALF = RG * (PSW + XS);
    auto rlane = [&](T x, int l) {  // (l: compile-time after unrolling)
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
    };
    // sum_l row[l rs] v_l over the (clamped) 32 components of the vector held one per lane in `v`: every LDS read issued before
    // the first use, four partial sums (a dependent float64 FMA issues every 8 cycles, an independent one every 4)
    auto dot32 = [&](const T *row, int rs, T v) {
#if STAGEG_DBG & 1
        return row[0] * v;  // (timing experiment: no dot product)
#endif
#if STAGEG_DBG & 2
        {  // (timing experiment: the LDS reads without the lane reads)
            T a = 0.0;
#pragma unroll
            for (int l = 0; l < NXM; ++l) a += row[(l < nx ? l : 0) * rs] * v;
            return a;
        }
#endif
#if STAGEG_DBG & 4
        {  // (timing experiment: the lane reads without the LDS reads)
            T a = 0.0;
            const T r0 = row[0];
#pragma unroll
            for (int l = 0; l < NXM; ++l) a += r0 * rlane(v, l);
            return a;
        }
#endif
        T rv_[NXM];
#pragma unroll
        for (int l = 0; l < NXM; ++l) rv_[l] = row[(l < nx ? l : 0) * rs];
        T a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int l = 0; l < NXM; l += 4) {
            a0 += rv_[l] * rlane(v, l);
            a1 += rv_[l + 1] * rlane(v, l + 1);
            a2 += rv_[l + 2] * rlane(v, l + 2);
            a3 += rv_[l + 3] * rlane(v, l + 3);
        }
        return (a0 + a1) + (a2 + a3);
    };
    // part: 1 = the backward sweep only (leaves the feed-forward terms in ffv), 2 = the forward sweep + the rows of G only (takes
    // them from ffv), 3 = both. given_u: ffv holds the INPUTS themselves (a roll-out: u = ffv, x+ = A x + B u = Acl x + B (u + K x)).
    auto sweep = [&](int kp, int rp, bool tracking, const T *xstart, T *Vout, T *Hout, int part, bool given_u) {
        const int ktop = tracking ? N - 1 : kp;
        if (part & 1)
            for (int i = tid; i < n; i += BS)
                if (i >= (ktop + 1) * nu) ffv[i] = 0.0;
        // steps kfirst, kfirst + dir, ... (RG of them, inside the horizon) into half `half` of the ring, by threads t, t + nth, ...:
        // the blocks are one contiguous run of the workspace (eight loads in flight per thread), slot sidx <-> step kfirst + dir sidx
        auto load_chunk = [&](int half, int kfirst, int dir, int t, int nth, bool fwd) {
            int cnt = 0;
            for (int sidx = 0; sidx < RG; ++sidx) cnt += (kfirst + dir * sidx >= 0 && kfirst + dir * sidx < N) ? 1 : 0;
            if (cnt == 0) return;
            const int klo = dir > 0 ? kfirst : kfirst - (cnt - 1);  // lowest step of the run
            const T *src = Blk + (int64_t)klo * PSW;
            T *dst = ring + (size_t)half * HALF;  // (block of step k at (k - klo) PSW)
            typedef T T2 __attribute__((ext_vector_type(2)));
            const int total = cnt * PSW / 2;  // (16-byte pairs: PSW is even, the blocks and the ring are 16-byte aligned)
            const T2 *src2 = (const T2 *)src;
            T2 *dst2 = (T2 *)dst;
            for (int j0 = t; j0 < total; j0 += nth * 8) {
                T2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src2[j0 + u * nth < total ? j0 + u * nth : total - 1];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u * nth < total) dst2[j0 + u * nth] = v[u];
            }
            T *ext = dst + RG * PSW;
            for (int j = t; j < cnt * XS; j += nth) {
                const int kk = j / XS, i = j - kk * XS, k = klo + kk;
                ext[j] = i < nx ? ((!fwd && tracking && stageQ && k >= 1) ? gtgt[(int64_t)k * nx + i] : 0.0) : (fwd ? ffv[k * nu + (i - nx)] : 0.0);
            }
        };
        const int lane = tid & 63, ti = lane - 32;
        const bool w0 = tid < 64, roleP = w0 && lane < nx, roleT = w0 && ti >= 0 && ti < nu;
        const int lp = lane < nx ? lane : 0, tq = (ti >= 0 && ti < nu) ? ti : 0;  // (clamped: every lane reads SOME row)
        // ---- backward: p_k = ql + Acl' p - K' rl, t = B' p + rl, ff = -S^-1 t  (ql = -C[kp, rp], rl = -D[kp, rp] at the row's stage)
        T pc = (roleP && tracking && termQ) ? -ka.wt * ggoal[lane] : 0.0;  // p (lane l: component l)
        // (what a step needs besides its slot is formed once per sweep: a kernel argument or a 64-bit address formed per step is
        // a scalar load / a chain of integer operations in the serial wavefront's path)
        const T wxs = ka.wx;
        const T *DrK = (!tracking && gD) ? gD + kp * sD + rp * nu : nullptr, *CrK = (!tracking && gC) ? gC + kp * sC + rp * nx : nullptr;
        const T crl = (CrK && roleP) ? -CrK[lane] : 0.0, drl = (DrK && roleT) ? -DrK[ti] : 0.0;  // the row's own entries, by lane
        const long long tb0 = (long long)__builtin_readcyclecounter();
        bsync();
        if (part & 1) load_chunk(0, ktop, -1, tid, BS, false);
        bsync();
        for (int c = 0, kc = ktop; (part & 1) && kc >= 0; ++c, kc -= RG) {
            if (!w0) {
                load_chunk((c + 1) & 1, kc - RG, -1, tid - 64, BS - 64, false);
            } else {
                const long long tc0 = (long long)__builtin_readcyclecounter();
                const int cntb = kc + 1 < RG ? kc + 1 : RG, klo = kc - (cntb - 1);
                for (int sidx = 0; sidx < RG && kc - sidx >= 0; ++sidx) {
                    const int k = kc - sidx;
                    const T *blk = ring + (size_t)(c & 1) * HALF + (size_t)(k - klo) * PSW;
                    const T *ext = ring + (size_t)(c & 1) * HALF + (size_t)RG * PSW + (size_t)(k - klo) * XS;
                    const bool here = !tracking && k == kp;
                    // this lane's row: state lanes a column of Acl (row lp of the stored transpose), input lanes a column of B
                    const T *row = ti >= 0 ? blk + oB + tq : blk + lp * nx;
                    const int rs = ti >= 0 ? nu : 1;
                    T acc = here ? crl + drl : 0.0;
                    if (roleP) acc -= wxs * ext[oT + lp];
                    acc += dot32(row, rs, pc);
                    if (here && DrK && roleP) {
                        const T *Kk = blk + oK;
                        for (int a = 0; a < nu; ++a) acc += Kk[a * nx + lane] * DrK[a];
                    }
                    // (input lanes: ff = -S^-1 t, t = acc of those lanes. No branch per term -- a taken or untaken branch costs a lone
                    // wavefront more than the term: lanes that are no input lane hold a zero, the clamped entry of S^-1 is a number)
                    const T tm = roleT ? acc : 0.0;
                    const T *srow = blk + oS + tq * nu;
                    T f0 = 0.0, f1 = 0.0;
#pragma unroll
                    for (int bb = 0; bb < NUM; bb += 2) {
                        f0 -= srow[bb < nu ? bb : 0] * rlane(tm, 32 + bb);
                        f1 -= srow[bb + 1 < nu ? bb + 1 : 0] * rlane(tm, 32 + bb + 1);
                    }
                    const T f = f0 + f1;
                    if (roleT) ffv[k * nu + ti] = f;
                    pc = roleP ? acc : 0.0;
                }
                t_w0 += (long long)__builtin_readcyclecounter() - tc0;
            }
            bsync();
        }
        // ---- forward: u = ff - K x, x+ = Acl x + B ff  (= A x + B u)
        const long long tf0 = (long long)__builtin_readcyclecounter();
        t_bwd += tf0 - tb0;
        T xc = (roleP && xstart) ? xstart[lane] : 0.0;  // x (lane l: component l)
        if (!(part & 2)) return;
        load_chunk(0, 0, 1, tid, BS, true);  // (the barrier above made the feed-forward terms visible)
        bsync();
        for (int c = 0, kc = 0; kc < N; ++c, kc += RG) {
            if (!w0) {
                load_chunk((c + 1) & 1, kc + RG, 1, tid - 64, BS - 64, true);
            } else {
                const long long tc0 = (long long)__builtin_readcyclecounter();
                for (int sidx = 0; sidx < RG && kc + sidx < N; ++sidx) {
                    const int k = kc + sidx;
                    const T *blk = ring + (size_t)(c & 1) * HALF + (size_t)sidx * PSW;
                    const T *ext = ring + (size_t)(c & 1) * HALF + (size_t)RG * PSW + (size_t)sidx * XS;
                    if (roleP) Xt[k * nx + lane] = xc;
                    // state lanes: row lp of Acl (a column of the stored transpose) and of B; input lanes: row tq of K, negated below
                    const T *row = ti >= 0 ? blk + oK + tq * nx : blk + lp;
                    const int rs = ti >= 0 ? 1 : nx;
                    const T sg = ti >= 0 ? -1.0 : 1.0;
                    const T accd = sg * dot32(row, rs, xc);  // state lanes: Acl x ; input lanes: -K x
                    const T gu = roleT ? ext[oF + tq] : 0.0;  // (input lanes: their feed-forward term, or their input; zero elsewhere)
                    const T ffown = given_u ? (roleT ? gu - accd : 0.0) : gu;  // the feed-forward term in effect
                    T acc = roleT ? (given_u ? gu : accd + gu) : accd;
                    const T *brow = blk + oB + lp * nu;
                    T b0 = 0.0, b1 = 0.0;  // B ff (no branch per term: see the backward sweep)
#pragma unroll
                    for (int a = 0; a < NUM; a += 2) {
                        b0 += brow[a < nu ? a : 0] * rlane(ffown, 32 + a);
                        b1 += brow[a + 1 < nu ? a + 1 : 0] * rlane(ffown, 32 + a + 1);
                    }
                    if (roleP) acc += b0 + b1;
                    if (roleT) Vout[k * nu + ti] = acc;
                    xc = roleP ? acc : 0.0;
                }
                t_w0 += (long long)__builtin_readcyclecounter() - tc0;
            }
            bsync();
        }
        const long long tr0 = (long long)__builtin_readcyclecounter();
        t_fwd += tr0 - tf0;
        for (int i = tid; i < M; i += BS) {  // h = G (x, u), one row per thread (the barrier above made the trajectory visible)
            const int k = i / mk, r = i - k * mk;
            T acc = 0.0;
            if (gC) {
                const T *c = gC + k * sC + r * nx, *x = Xt + k * nx;
                for (int l = 0; l < nx; ++l) acc += c[l] * x[l];
            }
            if (gD) {
                const T *d = gD + k * sD + r * nu, *u = Vout + k * nu;
                for (int a = 0; a < nu; ++a) acc += d[a] * u[a];
            }
            Hout[i] = acc;
        }
        bsync();
        t_rows += (long long)__builtin_readcyclecounter() - tr0;
    };

    int status = notpd ? (int)MPCQP_NOT_PD : (int)MPCQP_MAX_ITER, iters = 0, nq = 0;
    // ---- the factorisation's routines (every thread of the workgroup calls them)
    auto load_dinv = [&](int k) {
        for (int a2 = tid; a2 < k; a2 += BS) dv[a2] = 1.0 / Rm[(int64_t)a2 * maxq + a2];
        bsync();
    };
    // r = R^-1 d for the first k slots (d in yv), into ro. Thread j owns row j: the owner of row b publishes r_b, every row above
    // takes its entry of column b times r_b off its own sum -- one barrier per step; the thread's own row is fetched eight entries
    // (contiguous along the row) per round trip to the workspace.
    auto solveR_n = [&](int k, auto nrc) {  // (NR rows per thread: 1 while the slots fit one row per thread)
        constexpr int NR = decltype(nrc)::value;
        T acc[NR];
        const T *lr[NR];
#pragma unroll
        for (int q2 = 0; q2 < NR; ++q2) {
            const int j = tid + q2 * BS;
            acc[q2] = j < k ? yv[j] : 0.0;
            lr[q2] = Rm + (int64_t)(j < k ? j : 0) * maxq;
        }
        for (int hi = k - 1; hi >= 0; hi -= 8) {  // columns hi, hi - 1, ..., hi - 7
            T ent[NR][8];
#pragma unroll
            for (int q2 = 0; q2 < NR; ++q2) {
                const int j = tid + q2 * BS;
#pragma unroll
                for (int u = 0; u < 8; ++u) ent[q2][u] = (hi - u >= 0 && j < hi - u) ? lr[q2][hi - u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b2 = hi - u;
                if (b2 < 0) break;  // (uniform)
#pragma unroll
                for (int q2 = 0; q2 < NR; ++q2)
                    if (tid + q2 * BS == b2) ro[b2] = acc[q2] * dv[b2];
                bsync();
                const T rb = ro[b2];
#pragma unroll
                for (int q2 = 0; q2 < NR; ++q2) acc[q2] -= ent[q2][u] * rb;  // (zero for the rows at and below b2)
            }
        }
        bsync();
    };
    auto solveR = [&](int k) {
        if (maxq <= BS)
            solveR_n(k, std::integral_constant<int, 1>{});
        else
            solveR_n(k, std::integral_constant<int, NRM>{});
    };
    // The candidate y (n entries, |y|^2 = yy) against the first k vectors of Q: d = Q' y into yv, z = y - Q d into zout; when that
    // cancelled (|z|^2 < |y|^2 / 4) once more on z, the second pass's coefficients added to d (classical Gram-Schmidt with
    // re-orthogonalisation on demand, "twice is enough"); returns |z|^2 as the sum of z's squares. A wavefront per four vectors for
    // the dot products (coalesced, DPP reduction), a thread per entry for the AXPYs, eight loads in flight per thread: the passes are
    // bound by the round trips to Q in the workspace.
    auto ortho = [&](const T *y, T *zout, int k, T yy) -> T {
        const int wv = tid >> 6, ln = tid & 63;
        T zz = yy;
        for (int pass = 0; pass < 2; ++pass) {
            const T *src = pass == 0 ? y : zout;
            T *co = pass == 0 ? yv : ev;
            for (int a0i = 4 * wv; a0i < k; a0i += 4 * (BS / 64)) {  // four vectors per wavefront and round: their loads overlap
                const T *qa = Qs + (int64_t)a0i * n;
                const int64_t s1 = a0i + 1 < k ? n : 0, s2 = a0i + 2 < k ? 2 * (int64_t)n : 0, s3 = a0i + 3 < k ? 3 * (int64_t)n : 0;
                T p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll 2
                for (int i = ln; i < n; i += 64) {
                    const T sv = src[i];
                    p0 += qa[i] * sv;
                    p1 += qa[s1 + i] * sv;
                    p2 += qa[s2 + i] * sv;
                    p3 += qa[s3 + i] * sv;
                }
                p0 = wave_sum_dpp(p0);
                p1 = wave_sum_dpp(p1);
                p2 = wave_sum_dpp(p2);
                p3 = wave_sum_dpp(p3);
                if (ln == 0) {
                    co[a0i] = p0;
                    if (a0i + 1 < k) co[a0i + 1] = p1;
                    if (a0i + 2 < k) co[a0i + 2] = p2;
                    if (a0i + 3 < k) co[a0i + 3] = p3;
                }
            }
            bsync();
            T part = 0.0;
            for (int i = tid; i < n; i += BS) {
                T a0 = src[i], a1 = 0.0;
                int a2 = 0;
                for (; a2 + 8 <= k; a2 += 8) {
                    T qv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) qv[u] = Qs[(int64_t)(a2 + u) * n + i];
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        a0 -= co[a2 + u] * qv[u];
                        a1 -= co[a2 + u + 1] * qv[u + 1];
                    }
                }
                for (; a2 < k; ++a2) a0 -= co[a2] * Qs[(int64_t)a2 * n + i];
                const T zi = a0 + a1;
                zout[i] = zi;
                part += zi * zi;
            }
            if (pass == 1)
                for (int a2 = tid; a2 < k; a2 += BS) yv[a2] += ev[a2];
            const T prev = zz;
            zz = block_sum(part, redv, tid);  // (its barriers publish z and d)
            if (pass == 0 && (k == 0 || zz >= 0.25 * prev)) break;  // no cancellation: Q' z is at rounding level already
        }
        return zz;
    };
    // the candidate becomes basis vector k = slot k: Q gains z / |z|, R the column [d; |z|]
    auto append = [&](int k, T zn2) {
        const T zn = sqrt(zn2), izn = 1.0 / zn;
        T *qk = Qs + (int64_t)k * n;
        for (int i = tid; i < n; i += BS) qk[i] *= izn;
        for (int j = tid; j < k; j += BS) Rm[(int64_t)j * maxq + k] = yv[j];
        if (tid == 0) Rm[(int64_t)k * maxq + k] = zn;
    };
    // slot l leaves: its column of R goes (every thread closes the gap in its own rows; the small per-slot arrays move down by
    // one), and one Givens rotation per column behind it -- on two rows of R and two vectors of Q -- restores the triangle
    auto drop_slot = [&](int l) {
        const int k = nq - 1;  // slots after the drop
        int na[NRM];
        T nl[NRM];
        const int rowl = actrow[l];
#pragma unroll
        for (int q2 = 0; q2 < NRM; ++q2) {
            const int j = tid + q2 * BS;
            na[q2] = 0;
            nl[q2] = 0.0;
            if (j < nq) {
                T *row = Rm + (int64_t)j * maxq;
                for (int b2 = (j > l ? j - 1 : l); b2 < k; ++b2) row[b2] = row[b2 + 1];  // (own row, ascending; row j > l gains entry j - 1)
            }
            if (j > l && j < nq) {
                na[q2] = actrow[j];
                nl[q2] = lamv[j];
            }
        }
        bsync();
#pragma unroll
        for (int q2 = 0; q2 < NRM; ++q2) {
            const int j = tid + q2 * BS;
            if (j > l && j < nq) {
                actrow[j - 1] = na[q2];
                lamv[j - 1] = nl[q2];
                pos[na[q2]] = j - 1;
            }
        }
        if (tid == 0) pos[rowl] = -1;  // (its slack is zero now and moves with the next steps)
        bsync();
        // The rotation of step j mixes rows j, j + 1 of R and vectors j, j + 1 of Q. Thread b carries column b's entry of the
        // MOVING row (row j after the rotations before it) in a register, so what a step reads from memory -- row j + 1 -- no earlier
        // step has written: one barrier per rotation (the owner of column j publishes (c, s)). The rotations are recorded and Q
        // takes them afterwards in one pass, every thread walking the chain for its own entries without any barrier.
        T *cr = vlds ? yl_s : ws + wl.yv, *sr = vlds ? el_s : ws + wl.ev;  // (c_j, s_j), j = l .. k - 1
        T carry[NRM], nxt[NRM];
#pragma unroll
        for (int q2 = 0; q2 < NRM; ++q2) {
            const int b2 = tid + q2 * BS;
            carry[q2] = (b2 >= l && b2 < k) ? Rm[(int64_t)l * maxq + b2] : 0.0;
            nxt[q2] = (b2 >= l && b2 < k && l < k) ? Rm[(int64_t)(l + 1) * maxq + b2] : 0.0;
        }
        for (int j = l; j < k; ++j) {  // zero R[j + 1][j] against R[j][j]
            T u[NRM];
#pragma unroll
            for (int q2 = 0; q2 < NRM; ++q2) {
                const int b2 = tid + q2 * BS;
                u[q2] = nxt[q2];
                if (j + 1 < k && b2 > j && b2 < k) nxt[q2] = Rm[(int64_t)(j + 2) * maxq + b2];  // (row j + 2: next step's)
                if (b2 == j) {
                    const T hh = sqrt(carry[q2] * carry[q2] + u[q2] * u[q2]);
                    cr[j] = hh > 0.0 ? carry[q2] / hh : 1.0;
                    sr[j] = hh > 0.0 ? u[q2] / hh : 0.0;
                }
            }
            bsync();
            const T cc = cr[j], ss = sr[j];
#pragma unroll
            for (int q2 = 0; q2 < NRM; ++q2) {
                const int b2 = tid + q2 * BS;
                if (b2 >= j && b2 < k) {
                    Rm[(int64_t)j * maxq + b2] = cc * carry[q2] + ss * u[q2];
                    carry[q2] = cc * u[q2] - ss * carry[q2];
                }
            }
        }
        bsync();
        for (int i = tid; i < n; i += BS) {
            T t1 = Qs[(int64_t)l * n + i];
            T un = l < k ? Qs[(int64_t)(l + 1) * n + i] : 0.0;
            for (int j = l; j < k; ++j) {
                const T u2 = un;
                if (j + 1 < k) un = Qs[(int64_t)(j + 2) * n + i];
                const T cc = cr[j], ss = sr[j];
                Qs[(int64_t)j * n + i] = cc * t1 + ss * u2;
                t1 = cc * u2 - ss * t1;
            }
        }
        bsync();
    };
    // every slack from scratch, at the point the loop has reached: a roll-out of the inputs Ucur through the dynamics (the forward
    // sweep with the inputs given) and the rows of G on it -- what the oracle does after a full step (sp = h_p - G_p x)
    T *Ucur = U0, *Zu = Vs, *Usc = Vs + n, *gz = Hs, *Hsc = Hs + M, *Yp = Ys;
    auto eval_slacks = [&](T *out) {
        for (int i = tid; i < n; i += BS) ffv[i] = Ucur[i];
        bsync();
        sweep(0, 0, true, gx0, Usc, out, 2, true);
        for (int i = tid; i < M; i += BS) {
            const int k = i / mk, r = i - k * mk;
            out[i] = ge[k * sE + r] - out[i];
        }
        bsync();
    };
    if (!notpd) {
        // unconstrained minimiser and its slacks (tracking terms as linear costs: q of mpc_qp.py:129-149)
        {
            const long long t0 = (long long)__builtin_readcyclecounter();
            sweep(0, 0, true, gx0, Ucur, sl, 3, false);
            t_sweeps += (long long)__builtin_readcyclecounter() - t0;
            ++n_sweeps;
        }
        for (int i = tid; i < M; i += BS) {
            const int k = i / mk, r = i - k * mk;
            const T ev2 = ge[k * sE + r];
            sl[i] = ev2 - sl[i];
            thr[i] = tol + tol * fabs(ev2);
            T nn = 0.0;
            if (gC)
                for (int j = 0; j < nx; ++j) nn += gC[k * sC + r * nx + j] * gC[k * sC + r * nx + j];
            if (gD)
                for (int a = 0; a < nu; ++a) nn += gD[k * sD + r * nu + a] * gD[k * sD + r * nu + a];
            invn[i] = nn > 0.0 ? rsqrt(nn) : 1.0;
            pos[i] = ev2 < 1e29 ? -1 : -2;  // -2: padded row, never selectable
        }
        bsync();
        const int max_iter = ka.max_iter;
        bool fail = false, slotsfull = false;
        int fails = 0, next_refresh = STAGEG_REFRESH;
        for (;;) {
            // ---- active-set loop (Goldfarb-Idnani with the thin QR of the whitened active rows; oracle/mpc_oracle.c is the dense form)
            for (;;) {
                if (nq > 0 && iters >= next_refresh) {  // every STAGEG_REFRESH iterations: the slacks from scratch (rounding drift)
                    next_refresh = iters + STAGEG_REFRESH;
                    eval_slacks(Hsc);
                    for (int i = tid; i < M; i += BS) sl[i] = pos[i] >= 0 ? 0.0 : Hsc[i];
                    bsync();
                }
                T best = INF;
                int bi = 0x7fffffff;
                for (int i = tid; i < M; i += BS) {
                    const T sv = sl[i];
                    if (pos[i] == -1 && sv < -thr[i]) {
                        const T sc = sv * invn[i];
                        if (sc < best || (sc == best && i < bi)) {
                            best = sc;
                            bi = i;
                        }
                    }
                }
                block_argmin(best, bi, redv, redi, tid);
                if (!(best < INF)) {
                    status = MPCQP_SOLVED;
                    break;
                }
                const int rowp = bi, kp = rowp / mk, rp = rowp - kp * mk;
                T up = 0.0;
                bool added = false, stop = false;
                while (!added) {
                    if (iters >= max_iter) {
                        if (stamp && tid == 0) stamp[8] = 1;  // (developer probe: why the problem stopped)
                        status = MPCQP_MAX_ITER;
                        stop = fail = true;
                        break;
                    }
                    ++iters;
                    {  // backward sweep on the row: ffv = -S_k^-1 t_k, the feed-forward terms of P^-1 g_p'
                        const long long t0 = (long long)__builtin_readcyclecounter();
                        sweep(kp, rp, false, nullptr, nullptr, nullptr, 1, false);
                        t_sweeps += (long long)__builtin_readcyclecounter() - t0;
                        ++n_sweeps;
                    }
                    // the row's whitened vector y_p = L^-1 g_p' = (-Ls_k' ff_k)_k (header); d = Q' y_p, z = y_p - Q d into Q's next
                    // vector, |z|^2 as a sum of squares; r = R^-1 d
                    T *Zw = Qs + (int64_t)nq * n;
                    T part = 0.0;
                    const long long tw0 = (long long)__builtin_readcyclecounter();
                    for (int i = tid; i < n; i += BS) {
                        const int k2 = i / nu, a2 = i - k2 * nu;
                        const T *ls = ws + wl.LS + (int64_t)k2 * nu * nu;
                        T acc = 0.0;
                        for (int b2 = a2; b2 < nu; ++b2) acc -= ls[b2 * nu + a2] * ffv[k2 * nu + b2];
                        Yp[i] = acc;
                        part += acc * acc;
                    }
                    const T dpp = block_sum(part, redv, tid);  // |y_p|^2 = g_p P^-1 g_p'  (its barriers publish y_p)
                    const long long to0 = (long long)__builtin_readcyclecounter();
                    t_whiten += to0 - tw0;
                    const T d2 = ortho(Yp, Zw, nq, dpp);
                    const long long to1 = (long long)__builtin_readcyclecounter();
                    t_ortho += to1 - to0;
                    load_dinv(nq);
                    solveR(nq);
                    t_solve += (long long)__builtin_readcyclecounter() - to1;
                    if (vlds) {
                        for (int a = tid; a < nq; a += BS) rv[a] = ro[a];
                        bsync();
                    }
                    // (the oracle's pivot test: |z|^2 > 1e-28 |y_p|^2, oracle/mpc_oracle.c; two orders above it here)
                    const bool can_move = nq < n && d2 > 1e-26 * dpp && d2 > 0.0;
                    if (can_move && nq >= maxq) {  // the step would need one more slot than this launch holds
                        slotsfull = stop = fail = true;
                        break;
                    }
                    T t1 = INF;
                    int l = 0x7fffffff;
                    for (int a = tid; a < nq; a += BS) {
                        const T ra = rv[a];
                        if (ra > 0.0) {
                            const T q = lamv[a] / ra;
                            if (q < t1 || (q == t1 && a < l)) {
                                t1 = q;
                                l = a;
                            }
                        }
                    }
                    const T slp = sl[rowp];
                    block_argmin(t1, l, redv, redi, tid);
                    const T t2 = can_move ? -slp / d2 : INF;
                    const T t = t1 < t2 ? t1 : t2;
                    if (!(t < INF)) {  // no step possible: the row depends on the active ones and no multiplier blocks
                        status = MPCQP_INFEASIBLE;
                        if (stamp && tid == 0) stamp[8] = 5;  // (developer probe: why the problem stopped)
                        stop = fail = true;
                        break;
                    }
                    if (can_move) {
                        // the step in the inputs: z_u = L'^-1 z = the forward sweep on the feed-forward terms -Ls_k'^-1 z_k (the
                        // PROJECTED vector goes through the sweep: no difference of large vectors anywhere), and G z with it
                        for (int k2 = tid; k2 < N; k2 += BS) {
                            const T *ls = ws + wl.LS + (int64_t)k2 * nu * nu;
                            T f[NUM];  // (fully unrolled below: registers)
#pragma unroll
                            for (int a2 = NUM - 1; a2 >= 0; --a2) {  // Ls' f = -z_k (upper triangular: back substitution)
                                T acc = a2 < nu ? -Zw[k2 * nu + a2] : 0.0;
#pragma unroll
                                for (int b2 = a2 + 1; b2 < NUM; ++b2)
                                    if (b2 < nu) acc -= ls[b2 * nu + (a2 < nu ? a2 : 0)] * f[b2];
                                f[a2] = a2 < nu ? acc / ls[a2 * nu + a2] : 0.0;
                            }
#pragma unroll
                            for (int a2 = 0; a2 < NUM; ++a2)
                                if (a2 < nu) ffv[k2 * nu + a2] = f[a2];
                        }
                        bsync();
                        {
                            const long long t0 = (long long)__builtin_readcyclecounter();
                            sweep(0, 0, false, nullptr, Zu, gz, 2, false);
                            t_sweeps += (long long)__builtin_readcyclecounter() - t0;
                        }
                        // z_u = P^-1 (g_p' - G_A' r): the point moves against it, u -= t z_u ; s = e - G u gains t G z_u
                        for (int i = tid; i < n; i += BS) Ucur[i] -= t * Zu[i];
                        for (int i = tid; i < M; i += BS) sl[i] = pos[i] >= 0 ? 0.0 : sl[i] + t * gz[i];
                    }
                    for (int a = tid; a < nq; a += BS) {
                        const T v = lamv[a] - t * rv[a];
                        lamv[a] = v < 0.0 ? 0.0 : v;
                    }
                    up += t;
                    bsync();
                    if (t2 <= t1) {  // full step: p takes slot nq: Q gains z / |z|, R the column [d; |z|]
                        append(nq, d2);
                        if (tid == 0) {
                            lamv[nq] = up;
                            actrow[nq] = rowp;
                            pos[rowp] = nq;
                            sl[rowp] = 0.0;
                        }
                        ++nq;
                        added = true;
                        bsync();
                    } else {  // partial step: slot l leaves, the slots behind it close the gap
                        const long long td0 = (long long)__builtin_readcyclecounter();
                        drop_slot(l);
                        t_drop += (long long)__builtin_readcyclecounter() - td0;
                        --nq;
                    }
                }
                if (stop) break;
            }
            if (fail) break;
            // ---- acceptance, from scratch (the oracle's: oracle/mpc_oracle.c): every multiplier >= 0, every active row on its bound
            //      to STAGEG_ACC (1 + |e_i|), no inactive row violated -- evaluated on the roll-out of the inputs that are returned
            eval_slacks(Hsc);
            bool dirty = false, offa = false;
            for (int a = tid; a < nq; a += BS) offa |= !(lamv[a] >= 0.0);
            for (int i = tid; i < M; i += BS) {
                const T fr = Hsc[i];
                const bool act = pos[i] >= 0;
                const T fac = 1000.0 * (STAGEG_ACC / 1e-6) > STAGEG_ACC / tol ? 1000.0 * (STAGEG_ACC / 1e-6) : STAGEG_ACC / tol;
                if (act)
                    offa |= !(fabs(fr) <= fac * thr[i]);
                else if (pos[i] == -1 && !(fr >= -4.0 * thr[i]))
                    dirty = true;
                sl[i] = act ? 0.0 : fr;
            }
            offa = block_any(offa, redi, tid);
            dirty = block_any(dirty, redi, tid);
            if (offa) {
                if (stamp && tid == 0) stamp[8] = 3;  // (developer probe: why the problem stopped)
                status = MPCQP_MAX_ITER;
                break;
            }
            if (!dirty) {
                status = MPCQP_SOLVED;
                break;
            }
            if (++fails >= 4) {
                if (stamp && tid == 0) stamp[8] = 4;  // (developer probe: why the problem stopped)
                status = MPCQP_MAX_ITER;
                break;
            }
        }
        if (slotsfull) status = MPCQP_SLOTS_FULL;
    }
    const bool ok = status == MPCQP_SOLVED;
    T *ou = (T *)ka.U + prob * (int64_t)n;
    for (int i = tid; i < n; i += BS) ou[i] = ok ? U0[i] : 0.0;  // (U0 holds the point the loop reached)
    if (ka.lam) {
        T *ol = (T *)ka.lam + prob * (int64_t)M;
        for (int i = tid; i < M; i += BS) ol[i] = (ok && !notpd && pos[i] >= 0) ? lamv[pos[i]] : 0.0;
    }
    if (tid == 0) {
        if (stamp) {
            stamp[0] = (long long)__builtin_readcyclecounter() - t_start;
            stamp[1] = t_ric - t_start;
            stamp[2] = t_sweeps;
            stamp[3] = n_sweeps;
            stamp[4] = t_bwd;
            stamp[5] = t_fwd;
            stamp[6] = t_rows;
            stamp[7] = t_w0;
            stamp[12] = t_ortho;
            stamp[13] = t_solve;
            stamp[14] = t_drop;
            stamp[15] = t_whiten;
        }
        if (ka.status) ka.status[prob] = status;
        if (ka.iters) ka.iters[prob] = iters;
    }
}

// ------------------------------------------------------------ host side
bool stageg_supported(const KernelArgs &ka, int dtype)
{
    return dtype == MPCQP_F64 && ka.nx >= 1 && ka.nx <= NXM && ka.nu >= 1 && ka.nu <= NUM && ka.m >= 1;
}
int stageg_default_maxq(const KernelArgs &ka)
{
    int q = ka.n < ka.m ? ka.n : ka.m;
    return q < 256 ? q : 256;
}
size_t stageg_ws_doubles(const KernelArgs &ka, int maxq) { return (size_t)make_ws(ka.nx, ka.nu, ka.N, ka.mk, maxq).total; }

int launch_stageg(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st)
{
    const Ws wl = make_ws(ka.nx, ka.nu, ka.N, ka.mk, maxq);
    // the sweeps' LDS ring: two halves of RG steps each (the recursion's matrices live in the same 64 KB before the first sweep)
    constexpr int ring_doubles = 8192;
    const bool attr_ok = hipFuncSetAttribute((const void *)mpcqp_stageg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             ring_doubles * (int)sizeof(double)) == hipSuccess;  // (per call: no state between calls)
    if (!attr_ok || maxq > 4 * BS) return MPCQP_EUNSUPPORTED;  // (a thread owns at most four rows of the active rows' factor)
    hipLaunchKernelGGL(mpcqp_stageg_kernel, dim3((unsigned)batch), dim3(BS), ring_doubles * sizeof(double), st, ka, wl, (double *)ws,
                       ring_doubles);
    return (int)hipGetLastError();
}

}  // namespace mpcqp
