// mpcqp_stagew.hip -- the stage-wise (uncondensed) formulation for WIDER systems: 2 <= nx <= 16, 1 <= nu <= 4,
// float64 and float32 (BASELINE config 5's dimensions nx = 12, nu = 4 among them).
//
// Same problem, same method and same per-problem data as mpcqp_stage.hip (see there: Goldfarb-Idnani in the metric of
// the condensed Hessian P with P never formed; a product with P^-1 is one LQR solve; qpmpc/mpc_qp.py:39,108-109 and
// qpmpc/solve_mpc.py:31-32 are what it replaces). What differs is how one wavefront handles nx x nx matrices that no
// longer fit a lane's registers:
//   * RICCATI RECURSION: the step's matrices (P, A_k, A_k', P A, A_cl, ...) live in LDS as 16 x 16 tiles; a product is
//     spread over the 64 lanes -- lane (p, c) = (lane / 16, lane % 16) owns the outputs (p + 4 t, c), t = 0..3 -- and reads
//     its operands from LDS (the B-side entry is shared by the lane's four outputs). Transposed operands are staged
//     transposed (A_k', B_k') so that every read runs along a row.
//   * SWEEPS of the LQR solve: serial over the horizon, each step a mat-vec whose 12-16 outputs sit on the 16 lanes of a
//     row group, the four row groups summing a quarter of the inner dimension each (two xor-shuffles); the running vector
//     (costate / state) lives in LDS. The factor stores A_cl and its transpose so that both sweeps read rows.
//   * the O(|A| N) part of an iteration (slot vectors, slacks, selection) is spread over the lanes by chunks of the
//     horizon exactly as in the narrow kernel, and the code is the same.
// One wavefront per problem and ~10 KB (f32) of LDS: 8+ problems per CU are in flight, against ONE for the dense
// large-problem solver (its packed L^-1 fills the LDS) -- that, not the flop count, is where most of the gain at
// config 5's size comes from.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpcqp.h"
#include "mpcqp_internal.h"

namespace mpcqp {

namespace stagew {

constexpr int NX = 16, NU = 4;  // capacities (register arrays, LDS tiles); the true dimensions nx, nu are run-time values
constexpr int LD = 17;          // row stride of the 16 x 16 LDS tiles (odd: the MFMA operand reads are conflict-free)
constexpr int REC = 12;         // per-lane record of a sweep step: 4 + 4 matrix entries + 4 small ones

struct Ws {  // per-problem workspace carve, in elements of T (host-computed, passed by value)
    int64_t Rb, Rf, Kt, ff, U0, X0, s, invn, rowslot, V, XV, W, total;
    int maxq;
};

inline Ws make_ws(int nx, int nu, int N, int mk, int maxq, size_t esz)
{
    Ws w{};
    int64_t o = 0;
    auto take = [&](int64_t cnt) {
        const int64_t at = o;
        o += (cnt + 3) & ~(int64_t)3;
        return at;
    };
    const int64_t NP = 64 * (int64_t)((N + 63) / 64);  // per-step arrays transposed by chunk, see mpcqp_stage.hip
    const int64_t m = NP * mk;
    w.Rb = take((int64_t)N * 64 * REC);   // lane-ordered records of the backward / forward sweep (one 48-byte load per
    w.Rf = take((int64_t)N * 64 * REC);   // lane and step instead of a dozen scattered ones)
    w.Kt = take((int64_t)N * nx * nu);
    w.ff = take((int64_t)N * nu);
    w.U0 = take(NP * nu);
    w.X0 = take(NP * nx);
    w.s = take(m);
    w.invn = take(m);
    w.rowslot = take((m * 4 + esz - 1) / esz);  // int32 per row
    w.V = take((int64_t)(maxq + 1) * NP * nu);
    w.XV = take((int64_t)(maxq + 1) * NP * nx);
    w.W = take((int64_t)maxq * maxq);
    o = (o + 127) & ~(int64_t)127;  // odd multiple of 512 B / 1 KB between problems (memory channels)
    if (((o >> 7) & 1) == 0) o += 128;
    w.total = o;
    w.maxq = maxq;
    return w;
}

template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <typename T> __device__ __forceinline__ void wave_argmin(T &v, int &idx)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        const bool take = (ov < v) || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
}
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
template <typename T> struct Tol;
template <> struct Tol<double> { static constexpr double dep = 1e-13; };
template <> struct Tol<float> { static constexpr float dep = 1e-6f; };
// 16 x 16 (+)= 16 x 4 times 4 x 16 on the matrix cores; operands A[lane % 16][lane / 16], B[lane / 16][lane % 16];
// result register t of a lane: column lane % 16, row 4 (lane / 16) + t in float32, (lane / 16) + 4 t in float64
// (checked on MI355X, tools/ note in DESIGN.md).
template <typename T> struct Mfma;
template <> struct Mfma<float> {
    using V = __attribute__((ext_vector_type(4))) float;
    static __device__ __forceinline__ V run(float a, float b, V c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int pg, int t) { return 4 * pg + t; }
};
template <> struct Mfma<double> {
    using V = __attribute__((ext_vector_type(4))) double;
    static __device__ __forceinline__ V run(double a, double b, V c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int pg, int t) { return pg + 4 * t; }
};

}  // namespace stagew

using namespace stagew;

template <typename T>
__global__ void __launch_bounds__(64)
    mpcqp_stagew_kernel(const KernelArgs ka, const Ws wl, T *__restrict__ wsbase, const int64_t batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stagew_smem[];
    const int lane = threadIdx.x, pg = lane >> 4, c16 = lane & 15;
    const int64_t prob = blockIdx.x;
    const int nx = ka.nx, nu = ka.nu, N = ka.N, mk = ka.mk, maxq = wl.maxq;
    const int L = (N + 63) / 64;
    const int k0 = lane * L < N ? lane * L : N;
    const int k1 = (k0 + L < N) ? k0 + L : N;
    const int64_t NP = 64 * (int64_t)L;
    auto wq = [&](int k) { return (int64_t)(k - k0) * 64 + lane; };
    auto wg = [&](int k) {
        const int j = k / L;
        return (int64_t)(k - j * L) * 64 + j;
    };
    const T INF = (T)HUGE_VAL;
    const T DEPTOL = Tol<T>::dep;
    // ---- LDS: matrix tiles of the Riccati step, the sweeps' running vectors, the vectors shared by the lanes
    T *Pm = (T *)stagew_smem, *PAm = Pm + 16 * LD, *Mm = PAm + 16 * LD, *Am = Mm + 16 * LD, *Atm = Am + 16 * LD;
    T *Acm = Atm + 16 * LD, *PBm = Acm + 16 * LD, *Bm = PBm + 16 * 4, *Btm = Bm + 16 * 4, *BPAm = Btm + 4 * LD;
    T *Km = BPAm + 4 * LD, *Sm = Km + 4 * LD, *Sim = Sm + 16, *vec = Sim + 16, *tv = vec + 16;
    T *cv = tv + 8, *rv = cv + maxq, *lamv = rv + maxq;
    int *actk = (int *)(lamv + maxq), *actr = actk + maxq;
    // ---- workspace
    T *ws = wsbase + prob * wl.total;
    T *Rb = ws + wl.Rb, *Rf = ws + wl.Rf, *Kt = ws + wl.Kt, *ffv = ws + wl.ff;
    T *U0 = ws + wl.U0, *X0 = ws + wl.X0, *sl = ws + wl.s, *invn = ws + wl.invn, *Vs = ws + wl.V, *XVs = ws + wl.XV, *Wm = ws + wl.W;
    int *rowslot = (int *)(ws + wl.rowslot);
    // ---- operands
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
    const T wu = (T)ka.wu, wx = stageP ? (T)ka.wx : T(0), wt = termP ? (T)ka.wt : T(0);

    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 16 : nullptr;
    auto tick = [&](int slot) {
        if (stamp && lane == 0) stamp[slot] = (long long)__builtin_readcyclecounter();
    };
    tick(0);

    // C[r][c] = alpha sum_k A[r][k] B[k][c] + beta Add[r][c] (r < nr, c < nc) on the matrix cores: one MFMA per
    // chunk of 4 along k, operands read from the LDS tiles (every tile is ZERO outside its valid part, so the last
    // chunk may run past nk; rows / columns past nr / nc read neighbouring tiles and are not stored).
    auto mm = [&](T *C, int ldc, const T *A, int lda, const T *B, int ldb, int nr, int nc, int nk, T alpha, const T *Add,
                  int ldadd, T beta) {
        typename Mfma<T>::V acc = {T(0), T(0), T(0), T(0)};
        for (int k = 0; k < nk; k += 4) acc = Mfma<T>::run(A[c16 * lda + k + pg], B[(k + pg) * ldb + c16], acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = Mfma<T>::row(pg, t);
            if (r < nr && c16 < nc) C[r * ldc + c16] = alpha * acc[t] + (Add ? beta * Add[r * ldadd + c16] : T(0));
        }
    };

    // ================================================================= factor: Riccati recursion in LDS
    for (int i = lane; i < (int)(vec - Pm) + 24; i += 64) Pm[i] = T(0);  // every tile, vec and tv
    wsync();
    if (lane < nx) Pm[lane * LD + lane] = wt;
    wsync();
    // A_k, B_k are requested one step ahead (<= 4 + 1 entries per lane) and land in the LDS tiles at the top of their step
    T pfa[4], pfb;
    auto request = [&](int k) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = lane + 64 * u;
            pfa[u] = (i < nx * nx) ? gA[k * sA + i] : T(0);
        }
        pfb = (lane < nx * nu) ? gB[k * sB + lane] : T(0);
    };
    request(N - 1);
    for (int k = N - 1; k >= 0; --k) {
        // stage A_k, A_k', B_k, B_k'
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = lane + 64 * u;
            if (i < nx * nx) {
                const int r = i / nx, c = i - r * nx;
                Am[r * LD + c] = pfa[u];
                Atm[c * LD + r] = pfa[u];
            }
        }
        if (lane < nx * nu) {
            const int r = lane / nu, c = lane - r * nu;
            Bm[r * 4 + c] = pfb;
            Btm[c * LD + r] = pfb;
        }
        wsync();
        if (k > 0) request(k - 1);
        mm(PAm, LD, Pm, LD, Am, LD, nx, nx, nx, T(1), nullptr, 0, T(0));   // PA = P A
        mm(PBm, 4, Pm, LD, Bm, 4, nx, nu, nx, T(1), nullptr, 0, T(0));     // PB = P B
        wsync();
        mm(Sm, 4, Btm, LD, PBm, 4, nu, nu, nx, T(1), nullptr, 0, T(0));    // B' P B
        mm(BPAm, LD, Btm, LD, PAm, LD, nu, nx, nx, T(1), nullptr, 0, T(0));  // B' P A
        wsync();
        // S^-1 (nu <= 4): Gauss-Jordan on the symmetric positive definite S = w_u I + B'PB, every lane the same
        {
            T S[NU][NU], Si[NU][NU];
#pragma unroll
            for (int i = 0; i < NU; ++i)
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    S[i][j] = (i < nu && j < nu) ? Sm[i * 4 + j] + ((i == j) ? wu : T(0)) : ((i == j) ? T(1) : T(0));
                    Si[i][j] = (i == j) ? T(1) : T(0);
                }
#pragma unroll
            for (int c = 0; c < NU; ++c) {
                const T ip = T(1) / S[c][c];
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    S[c][j] *= ip;
                    Si[c][j] *= ip;
                }
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    if (r == c) continue;
                    const T f = S[r][c];
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        S[r][j] -= f * S[c][j];
                        Si[r][j] -= f * Si[c][j];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NU; ++i)
#pragma unroll
                for (int j = 0; j < NU; ++j) {  // (compile-time register indices: no scratch)
                    if (lane == i * 4 + j) Sim[lane] = Si[i][j];
                }
        }
        wsync();
        mm(Km, LD, Sim, 4, BPAm, LD, nu, nx, nu, T(1), nullptr, 0, T(0));  // K = S^-1 B'PA
        wsync();
        mm(Acm, LD, Bm, 4, Km, LD, nx, nx, nu, T(-1), Am, LD, T(1));       // Acl = A - B K
        mm(Mm, LD, PBm, 4, Km, LD, nx, nx, nu, T(-1), PAm, LD, T(1));      // M = P Acl = PA - PB K
        wsync();
        mm(PAm, LD, Atm, LD, Mm, LD, nx, nx, nx, T(1), nullptr, 0, T(0));  // A' P Acl (into the PA tile)
        // factors to the workspace as the sweeps' per-lane records, and K' (only read at the candidate row's step)
        {
            T *rb = Rb + ((int64_t)k * 64 + lane) * REC, *rf = Rf + ((int64_t)k * 64 + lane) * REC;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = pg + 4 * u;
                rb[u] = (j < nx && c16 < nx) ? Acm[j * LD + c16] : T(0);      // Acl[j][c]   (Acl' p)
                rb[4 + u] = (j < nx && c16 < nu) ? Bm[j * 4 + c16] : T(0);    // B[j][i]     (B' p)
                rb[8 + u] = (lane < nu && u < nu) ? Sim[lane * 4 + u] : T(0); // S^-1[i][l]
                rf[u] = (j < nx && c16 < nx) ? Acm[c16 * LD + j] : T(0);      // Acl[c][j]   (Acl x)
                rf[4 + u] = (j < nx && c16 < nu) ? Km[c16 * LD + j] : T(0);   // K[i][j]     (K x)
            }
            rf[8] = (c16 < nx && pg < nu) ? Bm[c16 * 4 + pg] : T(0);          // B[c][pg]    (B ff, one input per row group)
            rf[9] = rf[10] = rf[11] = T(0);
        }
        for (int i = lane; i < nx * nu; i += 64) {
            const int r = i / nu, c = i - r * nu;  // Kt[r][c] = K[c][r]
            Kt[(int64_t)k * nx * nu + i] = Km[c * LD + r];
        }
        wsync();
        // P_k = Q_k + sym(A' P Acl)   (x_0 is data: Q_0 = 0)
        {
            const T qk = (k >= 1) ? wx : T(0);
            T pn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = pg + 4 * t;
                pn[t] = (r < nx && c16 < nx) ? T(0.5) * (PAm[r * LD + c16] + PAm[c16 * LD + r]) + ((r == c16) ? qk : T(0)) : T(0);
            }
            wsync();
#pragma unroll
            for (int t = 0; t < 4; ++t) Pm[(pg + 4 * t) * LD + c16] = pn[t];
        }
        wsync();
    }
    tick(1);

    // ================================================================= the LQR solve: two serial sweeps
    // A step of a sweep is two mat-vecs  out[c] = sum_j Mat[j][c] v[j]: the 16 lanes of a row group hold the outputs c,
    // row group pg sums the quarter j = pg, pg + 4, ... of the inner dimension, two xor-shuffles add the quarters. The
    // matrix entries of step k -+ 1 are requested into registers while step k computes (they sit in HBM / L2).
    auto quarter_sum = [&](T a) {
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        return a;
    };
    // backward: p_k = g_k + Acl_k' p_{k+1}, g_k = q_k - K_k' r_k ; ff_k = -S_k^-1 (B_k' p_{k+1} + r_k).
    // Linear costs: q_k = -qrow, r_k = -rrow at k == kq (kq < 0: none); tracking terms when `track`.
    auto backward = [&](int kq, const T (&qrow)[NX], const T (&rrow)[NU], bool track) {
        if (lane < 16) vec[lane] = (track && termQ && lane < nx) ? -(T)ka.wt * ggoal[lane] : T(0);  // p_N
        T ma[4], mb[4], si[NU], tg;
        auto request = [&](int k) {
            const T *rb = Rb + ((int64_t)k * 64 + lane) * REC;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ma[u] = rb[u];
                mb[u] = rb[4 + u];
                si[u] = rb[8 + u];
            }
            tg = (track && stageQ && k >= 1 && c16 < nx) ? (T)ka.wx * gtgt[(int64_t)k * nx + c16] : T(0);
        };
        request(N - 1);
        wsync();
        for (int k = N - 1; k >= 0; --k) {
            T a4[4], b4[4], s4[NU];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a4[u] = ma[u];
                b4[u] = mb[u];
            }
#pragma unroll
            for (int l = 0; l < NU; ++l) s4[l] = si[l];
            const T tgk = tg;
            if (k > 0) request(k - 1);
            T pn = T(0), tb = T(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T vj = vec[(pg + 4 * u) & 15];  // entries >= nx are zero
                pn += a4[u] * vj;
                tb += b4[u] * vj;
            }
            pn = quarter_sum(pn);  // (Acl' p)_c
            tb = quarter_sum(tb);  // (B' p)_i in lanes c16 = i < nu
            if (k == kq) {
                const T *Kk = Kt + (int64_t)k * nx * nu;
                if (c16 < nu) tb -= rrow[c16];
                if (c16 < nx) {
                    T g = -qrow[c16];
#pragma unroll
                    for (int i = 0; i < NU; ++i)
                        if (i < nu) g += Kk[c16 * nu + i] * rrow[i];  // - K' r with r = -rrow
                    pn += g;
                }
            }
            pn -= tgk;
            // ff_i = - sum_l Sinv[i][l] tb_l : the tb_l sit in lanes l of this row group
            T ff = T(0);
#pragma unroll
            for (int l = 0; l < NU; ++l) ff -= s4[l] * __shfl(tb, l);
            if (lane < nu) ffv[(int64_t)k * nu + lane] = ff;
            wsync();
            if (lane < 16) vec[lane] = (lane < nx) ? pn : T(0);
            wsync();
        }
    };
    // forward: u_k = -K_k x_k + ff_k, x_{k+1} = Acl_k x_k + B_k ff_k from x_0 = xs; writes Uo, Xo (chunk-transposed)
    auto forward = [&](const T *xs, T *Uo, T *Xo) {
        if (lane < 16) vec[lane] = (xs && lane < nx) ? xs[lane] : T(0);
        T ma[4], mk4[4], bf;
        auto request = [&](int k) {
            const T *rf = Rf + ((int64_t)k * 64 + lane) * REC;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ma[u] = rf[u];
                mk4[u] = rf[4 + u];
            }
            // (B ff)_c, row group pg adds the term of input pg
            bf = (pg < nu) ? rf[8] * ffv[(int64_t)k * nu + pg] : T(0);
        };
        request(0);
        wsync();
        for (int k = 0; k < N; ++k) {
            T a4[4], k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a4[u] = ma[u];
                k4[u] = mk4[u];
            }
            const T bfk = bf;
            if (k + 1 < N) request(k + 1);
            const int64_t w = wg(k);
            T xn = bfk, kx = T(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T vj = vec[(pg + 4 * u) & 15];
                xn += a4[u] * vj;
                kx += k4[u] * vj;
            }
            xn = quarter_sum(xn);  // (Acl x + B ff)_c
            kx = quarter_sum(kx);  // (K x)_i
            if (lane < nx) Xo[w * nx + lane] = vec[lane];
            if (lane < nu) Uo[w * nu + lane] = ffv[(int64_t)k * nu + lane] - kx;
            wsync();
            if (lane < 16) vec[lane] = (lane < nx) ? xn : T(0);
            wsync();
        }
    };
    // g_(k,r) . (U, X) = C_k[r] x_k + D_k[r] u_k   (w = workspace index of step k)
    auto gdot = [&](int k, int64_t w, int r, const T *Uv, const T *Xv) {
        T a = T(0);
        if (gC) {
            const T *c = gC + k * sC + r * nx;
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (i < nx) a += c[i] * Xv[w * nx + i];
        }
        if (gD) {
            const T *d = gD + k * sD + r * nu;
#pragma unroll
            for (int i = 0; i < NU; ++i)
                if (i < nu) a += d[i] * Uv[w * nu + i];
        }
        return a;
    };

    // ================================================================= unconstrained minimiser, slacks
    {
        T zq[NX], zr[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) zq[i] = T(0);
#pragma unroll
        for (int i = 0; i < NU; ++i) zr[i] = T(0);
        tick(2);
        backward(-1, zq, zr, true);
    }
    wsync();
    tick(3);
    forward(gx0, U0, X0);
    wsync();
    tick(4);
    const T tol = ka.tol;
    for (int k = k0; k < k1; ++k)
        for (int r = 0; r < mk; ++r) {
            const int64_t i = wq(k) * mk + r;
            const T ev = ge[k * sE + r];
            sl[i] = ev - gdot(k, wq(k), r, U0, X0);
            T nn = 0.0;
            if (gC)
                for (int c = 0; c < NX; ++c) if (c < nx) nn += gC[k * sC + r * nx + c] * gC[k * sC + r * nx + c];
            if (gD)
                for (int c = 0; c < NU; ++c) if (c < nu) nn += gD[k * sD + r * nu + c] * gD[k * sD + r * nu + c];
            invn[i] = nn > 0.0 ? (T)rsqrt((double)nn) : 1.0;
            rowslot[i] = -1;
        }
    wsync();

    tick(5);
    // ================================================================= active-set loop
    int nq = 0, iters = 0, status = MPCQP_MAX_ITER;
    const int max_iter = ka.max_iter;
    const int nvar = N * nu;
    T *Vp = Vs + (int64_t)maxq * NP * nu, *Xp = XVs + (int64_t)maxq * NP * nx;  // the candidate's slot
    bool fail = false;
    for (int round = 0; round < 4 && !fail; ++round) {
        for (;;) {
            // ---- selection: the violated row farthest from its hyperplane
            T best = INF;
            int bi = 0x7fffffff;
            for (int k = k0; k < k1; ++k)
                for (int r = 0; r < mk; ++r) {
                    const int64_t i = wq(k) * mk + r;
                    const T ev = ge[k * sE + r], sv = sl[i];
                    const bool viol = ev < 1e29 && rowslot[i] < 0 && sv < -(tol + tol * fabs((double)ev));
                    const T sc = sv * invn[i];
                    if (viol && sc < best) {
                        best = sc;
                        bi = k * mk + r;  // natural row id: ties go to the lowest one, like the restatement
                    }
                }
            wave_argmin(best, bi);
            if (!(best < INF)) {
                status = MPCQP_SOLVED;
                break;
            }
            const int kp = bi / mk, rp = bi - kp * mk;
            const int64_t wp = wg(kp), bw = wp * mk + rp;  // workspace index of step kp / of row p
            T qrow[NX], rrow[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) if (i < nx) qrow[i] = gC ? gC[kp * sC + rp * nx + i] : 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) if (i < nu) rrow[i] = gD ? gD[kp * sD + rp * nu + i] : 0.0;
            T up = 0.0;
            bool added = false;
            // V_p = P^-1 g_p' and its trajectory do not change while p waits for room: solved once
            backward(kp, qrow, rrow, false);
            wsync();
            forward(nullptr, Vp, Xp);
            wsync();
            const T dpp = gdot(kp, wp, rp, Vp, Xp);
            while (!added) {
                if (iters >= max_iter || nq >= maxq) {
                    fail = true;
                    break;
                }
                ++iters;
                // ---- c_a = g_a . V_p ; r = W c ; d2 = g_p . V_p - c . r
                for (int a = lane; a < nq; a += 64) cv[a] = gdot(actk[a], wg(actk[a]), actr[a], Vp, Xp);
                wsync();
                T cr = 0.0;
                for (int a = lane; a < nq; a += 64) {
                    T acc = 0.0;
                    for (int b = 0; b < nq; ++b) acc += Wm[(int64_t)b * maxq + a] * cv[b];  // W is symmetric
                    rv[a] = acc;
                    cr += acc * cv[a];
                }
                cr = wave_sum(cr);
                wsync();
                const T d2 = dpp - cr;
                const bool can_move = (nq < nvar) && (d2 > DEPTOL * dpp) && (d2 > 0.0);
                // ---- ratio test on the multipliers
                T t1 = INF;
                int l = 0x7fffffff;
                for (int a = lane; a < nq; a += 64) {
                    const T ra = rv[a];
                    if (ra > 0.0) {
                        const T q = lamv[a] / ra;
                        if (q < t1) {
                            t1 = q;
                            l = a;
                        }
                    }
                }
                wave_argmin(t1, l);
                const T sp = sl[bw];
                const T t2 = can_move ? -sp / d2 : INF;
                const T t = t1 < t2 ? t1 : t2;
                if (!(t < INF)) {
                    status = MPCQP_INFEASIBLE;
                    fail = true;
                    break;
                }
                const bool full = (t2 <= t1);
                // ---- slacks: s_i -= t g_i . z ,  z = -(V_p - sum_a r_a V_a)  (this lane's chunk)
                for (int k = k0; k < k1; ++k) {
                    T zu[NU], zx[NX];
#pragma unroll
                    for (int i = 0; i < NU; ++i) if (i < nu) zu[i] = -Vp[wq(k) * nu + i];
#pragma unroll
                    for (int i = 0; i < NX; ++i) if (i < nx) zx[i] = -Xp[wq(k) * nx + i];
                    for (int a = 0; a < nq; ++a) {
                        const T ra = rv[a];
                        const T *va = Vs + ((int64_t)a * NP + wq(k)) * nu, *xa = XVs + ((int64_t)a * NP + wq(k)) * nx;
#pragma unroll
                        for (int i = 0; i < NU; ++i) if (i < nu) zu[i] += ra * va[i];
#pragma unroll
                        for (int i = 0; i < NX; ++i) if (i < nx) zx[i] += ra * xa[i];
                    }
                    for (int r = 0; r < mk; ++r) {
                        T gz = 0.0;
                        if (gC)
#pragma unroll
                            for (int i = 0; i < NX; ++i) if (i < nx) gz += gC[k * sC + r * nx + i] * zx[i];
                        if (gD)
#pragma unroll
                            for (int i = 0; i < NU; ++i) if (i < nu) gz += gD[k * sD + r * nu + i] * zu[i];
                        const int64_t i = wq(k) * mk + r;
                        sl[i] = (rowslot[i] >= 0) ? 0.0 : sl[i] - t * gz;
                    }
                }
                // ---- multipliers
                for (int a = lane; a < nq; a += 64) {
                    const T v = lamv[a] - t * rv[a];
                    lamv[a] = v < 0.0 ? 0.0 : v;
                }
                up += t;
                wsync();
                if (full) {
                    // p becomes active in slot nq: W is bordered, the candidate's vectors move into the slot
                    const T id2 = 1.0 / d2;
                    for (int a = lane; a < nq; a += 64) {
                        const T ra = rv[a];
                        for (int b = 0; b < nq; ++b) Wm[(int64_t)b * maxq + a] += rv[b] * ra * id2;
                        Wm[(int64_t)nq * maxq + a] = -ra * id2;
                        Wm[(int64_t)a * maxq + nq] = -ra * id2;
                    }
                    if (lane == 0) {
                        Wm[(int64_t)nq * maxq + nq] = id2;
                        lamv[nq] = up;
                        actk[nq] = kp;
                        actr[nq] = rp;
                        rowslot[bw] = nq;
                        sl[bw] = 0.0;
                    }
                    T *vd = Vs + (int64_t)nq * NP * nu, *xd = XVs + (int64_t)nq * NP * nx;
                    for (int k = k0; k < k1; ++k) {
#pragma unroll
                        for (int i = 0; i < NU; ++i) if (i < nu) vd[wq(k) * nu + i] = Vp[wq(k) * nu + i];
#pragma unroll
                        for (int i = 0; i < NX; ++i) if (i < nx) xd[wq(k) * nx + i] = Xp[wq(k) * nx + i];
                    }
                    ++nq;
                    added = true;
                } else {
                    // partial step: slot l leaves; W is deflated and the last slot moves into the hole
                    const T wll = Wm[(int64_t)l * maxq + l];
                    const T iw = 1.0 / wll;
                    for (int a = lane; a < nq; a += 64) cv[a] = Wm[(int64_t)l * maxq + a];  // row l before the update
                    wsync();
                    for (int a = lane; a < nq; a += 64) {
                        const T wa = cv[a];
                        for (int b = 0; b < nq; ++b) Wm[(int64_t)b * maxq + a] -= cv[b] * wa * iw;
                    }
                    wsync();
                    const int last = nq - 1;
                    const int64_t drow = wg(actk[l]) * mk + actr[l];
                    if (l != last) {
                        for (int a = lane; a < nq; a += 64) Wm[(int64_t)l * maxq + a] = Wm[(int64_t)last * maxq + a];
                        wsync();
                        for (int b = lane; b < nq; b += 64) Wm[(int64_t)b * maxq + l] = Wm[(int64_t)b * maxq + last];
                        wsync();
                        const T *vs = Vs + (int64_t)last * NP * nu, *xs = XVs + (int64_t)last * NP * nx;
                        T *vd = Vs + (int64_t)l * NP * nu, *xd = XVs + (int64_t)l * NP * nx;
                        for (int k = k0; k < k1; ++k) {
#pragma unroll
                            for (int i = 0; i < NU; ++i) if (i < nu) vd[wq(k) * nu + i] = vs[wq(k) * nu + i];
#pragma unroll
                            for (int i = 0; i < NX; ++i) if (i < nx) xd[wq(k) * nx + i] = xs[wq(k) * nx + i];
                        }
                    }
                    if (lane == 0) {
                        rowslot[drow] = -1;
                        if (l != last) {
                            lamv[l] = lamv[last];
                            actk[l] = actk[last];
                            actr[l] = actr[last];
                            rowslot[wg(actk[last]) * mk + actr[last]] = l;
                        }
                    }
                    --nq;
                }
                wsync();
            }
            if (fail) break;
        }
        if (fail) break;
        tick(6);
        // ================================================================= primal point, verification
        // u = u0 - sum_a lam_a V_a ; slacks from scratch through x = x0 - sum_a lam_a X_a
        bool dirty = false;
        for (int k = k0; k < k1; ++k) {
            T u[NU], x[NX];
#pragma unroll
            for (int i = 0; i < NU; ++i) if (i < nu) u[i] = U0[wq(k) * nu + i];
#pragma unroll
            for (int i = 0; i < NX; ++i) if (i < nx) x[i] = X0[wq(k) * nx + i];
            for (int a = 0; a < nq; ++a) {
                const T la = lamv[a];
                const T *va = Vs + ((int64_t)a * NP + wq(k)) * nu, *xa = XVs + ((int64_t)a * NP + wq(k)) * nx;
#pragma unroll
                for (int i = 0; i < NU; ++i) if (i < nu) u[i] -= la * va[i];
#pragma unroll
                for (int i = 0; i < NX; ++i) if (i < nx) x[i] -= la * xa[i];
            }
            T *ou = (T *)ka.U + prob * (int64_t)nvar + (int64_t)k * nu;
#pragma unroll
            for (int i = 0; i < NU; ++i) if (i < nu) ou[i] = u[i];
            for (int r = 0; r < mk; ++r) {
                const int64_t i = wq(k) * mk + r;
                const T ev = ge[k * sE + r];
                T g = 0.0;
                if (gC)
#pragma unroll
                    for (int c = 0; c < NX; ++c) if (c < nx) g += gC[k * sC + r * nx + c] * x[c];
                if (gD)
#pragma unroll
                    for (int c = 0; c < NU; ++c) if (c < nu) g += gD[k * sD + r * nu + c] * u[c];
                const T fresh = ev - g;
                const bool act = rowslot[i] >= 0;
                if (ev < 1e29 && !act && !(fresh >= -4.0 * (tol + tol * fabs((double)ev)))) dirty = true;
                sl[i] = act ? 0.0 : fresh;
            }
        }
        dirty = __ballot(dirty) != 0ull;
        wsync();
        if (!dirty) {
            status = MPCQP_SOLVED;
            break;
        }
        status = MPCQP_MAX_ITER;  // continue from the re-evaluated slacks
    }
    tick(7);
    if (fail && status == MPCQP_SOLVED) status = MPCQP_MAX_ITER;
    const bool ok = status == MPCQP_SOLVED;
    if (!ok) {
        T *ou = (T *)ka.U + prob * (int64_t)nvar;
        for (int k = k0; k < k1; ++k)
#pragma unroll
            for (int i = 0; i < NU; ++i) if (i < nu) ou[(int64_t)k * nu + i] = 0.0;
    }
    if (ka.lam) {
        T *ol = (T *)ka.lam + prob * (int64_t)N * mk;
        for (int k = k0; k < k1; ++k)
            for (int r = 0; r < mk; ++r) {
                const int sidx = rowslot[wq(k) * mk + r];
                ol[(int64_t)k * mk + r] = (ok && sidx >= 0) ? lamv[sidx] : 0.0;
            }
    }
    if (lane == 0) {
        if (ka.status) ka.status[prob] = status;
        if (ka.iters) ka.iters[prob] = iters;
    }
}

// ------------------------------------------------------------ host side
bool stagew_supported(const KernelArgs &ka, int dtype)
{
    return (dtype == MPCQP_F64 || dtype == MPCQP_F32) && ka.nx >= 2 && ka.nx <= NX && ka.nu >= 1 && ka.nu <= NU && ka.mk >= 1;
}

size_t stagew_ws_elems(const KernelArgs &ka, int maxq, int dtype)
{
    return (size_t)make_ws(ka.nx, ka.nu, ka.N, ka.mk, maxq, dtype == MPCQP_F64 ? 8 : 4).total;
}

template <typename T> static int launch_stagew_t(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st)
{
    const Ws wl = make_ws(ka.nx, ka.nu, ka.N, ka.mk, maxq, sizeof(T));
    const size_t tiles = (size_t)(6 * 16 * LD + 2 * 16 * 4 + 3 * 4 * LD + 16 + 16 + 16 + 8);
    const size_t lds = tiles * sizeof(T) + (size_t)maxq * (3 * sizeof(T) + 2 * sizeof(int)) + 16;
    auto kern = mpcqp_stagew_kernel<T>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(64), lds, st, ka, wl, (T *)ws, batch);
    return (int)hipGetLastError();
}

int launch_stagew(const KernelArgs &ka, int dtype, int maxq, int64_t batch, void *ws, hipStream_t st)
{
    return dtype == MPCQP_F64 ? launch_stagew_t<double>(ka, maxq, batch, ws, st) : launch_stagew_t<float>(ka, maxq, batch, ws, st);
}

}  // namespace mpcqp
