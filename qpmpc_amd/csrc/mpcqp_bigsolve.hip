// mpcqp_bigsolve.hip -- dual active-set solver for LARGE dense QPs (config 5:
// n = 256, m = 1024, f32), one problem per workgroup of 256 threads.
//
// Replaces qpsolvers.solve_problem(...) at qpmpc/solve_mpc.py:43 for problems whose
// matrices do not fit one CU's LDS. What does fit is the packed lower triangle of ONE
// n x n matrix (131 KB in f32), and the method is arranged around that:
//   * P is factorised in LDS (packed Cholesky) and the factor is INVERTED in place, so
//     that everything needed later is a mat-vec with L^-1 (parallel, no dependent sweep):
//         M_p = L^-1 G_p'  (the row of M = G L^-T of a selected constraint, computed lazily:
//                           only the handful of rows that ever get selected, not all m),
//         z_x = L^-T z     (primal direction back in the original coordinates);
//   * the Goldfarb-Idnani operator N* (rows T_a) and the active rows M_a live in an HBM
//     workspace (nq x n each, read column-wise by thread k: coalesced);
//   * the slack update s -= t G z_x reads G through its TRANSPOSE (written by the
//     propagation kernel), thread i owning rows i, i+256, ...: coalesced, no reductions.
// Same algorithm and selection rule as mpcqp_w64.hip (explicit N*, rows ranked by their
// distance to the hyperplane -- here in the Euclidean metric of the original rows, whose
// norms are cheap, since the rows of M are not formed).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "mpcqp.h"
#include "mpcqp_internal.h"

namespace mpcqp {

namespace bigs {
constexpr int BS = 256;
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }
// Per-problem stride of the solver's workspace rows (M_A and N*, n x n each). The problems of a launch walk their
// slices in lock step: 2 n^2 elements is a power of two for n = 256 and would put every problem on the same memory
// channels (measured on the stage-wise kernel: 8x). 160 extra elements break the alignment.
__host__ __device__ inline int64_t ws_stride_elems(int n) { return (int64_t)2 * n * n + 160; }

template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
template <int NWV, typename T> __device__ __forceinline__ T block_sum(T v, T *red, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    T s = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) s += red[w];
    return s;
}
// arg-min of (v, i) over the block; ties -> lowest index
template <int NWV, typename T> __device__ __forceinline__ void block_argmin(T &v, int &i, T *redv, int *redi, int tid)
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
    __syncthreads();
    if ((tid & 63) == 0) {
        redv[tid >> 6] = v;
        redi[tid >> 6] = i;
    }
    __syncthreads();
    v = redv[0];
    i = redi[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w)
        if (redv[w] < v || (redv[w] == v && redi[w] < i)) {
            v = redv[w];
            i = redi[w];
        }
}
// 1/sqrt(x) and 1/x from the hardware estimates plus Newton steps (full precision of T; the IEEE
// sqrt + divide sequences are ~100 instructions each in float64 and sit on the factorisation's
// critical path once per column)
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}
__device__ __forceinline__ float fast_rsqrt(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
}
__device__ __forceinline__ double fast_recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ float fast_recip(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    return fmaf(y, fmaf(-x, y, 1.0f), y);
}
}  // namespace bigs
using namespace bigs;

// Packed Cholesky followed by the in-place inverse of the factor, scalar version (any n <= 256,
// any dtype): thread i owns row i. Returns false when a pivot is not positive.
// WAVE = true: n <= 64, called by the first wavefront only -- every participant is in one wavefront, so
// the ~4 n workgroup barriers become wavefront-level fences (LDS is in order within a wavefront).
template <typename T, bool WAVE = false>
__device__ __forceinline__ bool factor_invert_scalar(T *Li, int n, int tid, T *red)
{
    auto sync = [&]() __attribute__((always_inline)) {
        if constexpr (WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    };
    // ---- Cholesky, left-looking by columns: thread i owns row i
    bool notpd = false;
    {
        const int i = tid;  // n <= BS
        const T *ri = Li + tri(i < n ? i : 0, 0);
        int rowj = 0;  // tri(j, 0)
        for (int j = 0; j < n; rowj += ++j) {
            // column j below the diagonal (and the pivot by thread j)
            T v = T(0);
            if (i >= j && i < n) {
                T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
                const T *rj = Li + rowj;
                int k = 0;
                for (; k + 8 <= j; k += 8) {
                    a0 += ri[k] * rj[k];
                    a1 += ri[k + 1] * rj[k + 1];
                    a2 += ri[k + 2] * rj[k + 2];
                    a3 += ri[k + 3] * rj[k + 3];
                    a0 += ri[k + 4] * rj[k + 4];
                    a1 += ri[k + 5] * rj[k + 5];
                    a2 += ri[k + 6] * rj[k + 6];
                    a3 += ri[k + 7] * rj[k + 7];
                }
                for (; k < j; ++k) a0 += ri[k] * rj[k];
                v = ri[j] - ((a0 + a1) + (a2 + a3));
            }
            if (i == j) red[0] = v;  // pivot
            sync();
            const T piv = red[0];
            if (!(piv > T(0))) {
                notpd = true;
                break;
            }
            const T rinv = fast_rsqrt(piv);
            if (i >= j && i < n) Li[tri(i, 0) + j] = (i == j) ? piv * rinv : v * rinv;
            sync();
        }
    }
    if (notpd) return false;
    {
        // ---- invert L in place, row by row: X[i][j] = -(sum_{k=j}^{i-1} L[i][k] X[k][j]) / L[i][i]
        // k runs uniformly over the wavefront (from its first column), so L[i][k] is a broadcast
        // read, X[k][j] a conflict-free one, and the row bases stay in scalar registers.
        {
            const int j = tid, wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
            int rowi = 0;
            for (int i = 0; i < n; rowi += ++i) {
                const T *ri = Li + rowi;
                const T dinv = fast_recip(ri[i]);
                T x = T(0);
                if (wbase < i) {
                    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
                    int k = wbase, rowk = tri(wbase, 0);
                    for (; k + 4 <= i; k += 4) {
                        const int r1 = rowk + k + 1, r2 = r1 + k + 2, r3 = r2 + k + 3;
                        const int jc = j < n ? j : 0;  // unconditional, in-bounds loads; selected below
                        const T y0 = Li[rowk + jc], y1 = Li[r1 + jc], y2 = Li[r2 + jc], y3 = Li[r3 + jc];
                        const T x0 = (k >= j) ? y0 : T(0);
                        const T x1 = (k + 1 >= j) ? y1 : T(0);
                        const T x2 = (k + 2 >= j) ? y2 : T(0);
                        const T x3 = (k + 3 >= j) ? y3 : T(0);
                        a0 += ri[k] * x0;
                        a1 += ri[k + 1] * x1;
                        a2 += ri[k + 2] * x2;
                        a3 += ri[k + 3] * x3;
                        rowk = r3 + k + 4;
                    }
                    for (; k < i; ++k) {
                        a0 += ri[k] * ((k >= j) ? Li[rowk + j] : T(0));
                        rowk += k + 1;
                    }
                    x = -((a0 + a1) + (a2 + a3)) * dinv;
                }
                sync();
                if (j < i) Li[rowi + j] = x;
                if (j == i) Li[rowi + i] = dinv;
                sync();
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------ blocked factor + inverse on the matrix cores (f32)
// The packed matrix is cut into 32 x 32 blocks. Per block column J: (1) the diagonal block is
// factorised and inverted by one wavefront in registers (lanes = rows, v_readlane broadcasts) and
// REPLACED by its inverse W_J; (2) the panel below becomes L[I,J] = A[I,J] W_J' and (3) the trailing
// blocks A[I,I2] -= L[I,J] L[I2,J]', both as v_mfma_f32_32x32x2_f32 tiles shared out over the four
// wavefronts. The inverse X = L^-1 then follows block row by block row:
//     X[I,J] = - sum_{K=J}^{I-1} (W_I L[I,K]) X[K,J],   X[J,J] = W_J,
// again as MFMA tiles, in place. f32 MFMA is an exact fmaf chain, so this is the same arithmetic as
// the scalar version in another summation order. ~3600 MFMAs per 256 x 256 problem.
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float rlane(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// row inside a 32 x 32 accumulator tile held by register t of this lane (column = lane & 31)
__device__ __forceinline__ int crow(int t, int lane) { return (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5); }

// One wavefront: diagonal block J -> its Cholesky factor -> the inverse of that factor, in place.
// Lane r keeps row r in registers. Column c is broadcast through a 32-float LDS buffer (one write,
// eight 16-byte broadcast reads) rather than lane by lane; the inverse W = L^-1 is then one forward
// substitution per lane (lane r = column r) against the rows of L read back as broadcasts.
__device__ __forceinline__ bool diag_block_invert(float *Li, int J, int lane, float *colbuf)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    int r = lane & 31;
    // opaque to the optimiser: otherwise the 100+ lane predicates below are hoisted out of the
    // caller's loop over J and kept alive through the MFMA phases (hundreds of spills)
    asm volatile("" : "+v"(r));
    const int c0 = 32 * J;
    float *row = Li + tri(c0 + r, 0) + c0;
    float d[32], w[32], rinvs[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const float dl = row[c];
        d[c] = (c <= r) ? dl : 0.0f;
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        if (lane < 32) colbuf[r] = d[c];  // column c before scaling
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float cv[32];
#pragma unroll
        for (int g = c / 4; g < 8; ++g) {
            const f4 v = *reinterpret_cast<const f4 *>(colbuf + 4 * g);
            cv[4 * g] = v[0];
            cv[4 * g + 1] = v[1];
            cv[4 * g + 2] = v[2];
            cv[4 * g + 3] = v[3];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float piv = cv[c];
        ok = ok && (piv > 0.0f);
        float rinv = __builtin_amdgcn_rsqf(piv);  // v_rsq_f32 + one Newton step
        rinv = rinv * (1.5f - 0.5f * piv * rinv * rinv);
        rinvs[c] = rinv;
        // L[r][c] = a[r][c] rinv ; a[r][c2] -= L[r][c] L[c2][c] = (a[r][c] rinv^2) a[c2][c]
        const float f = (r >= c) ? d[c] * (rinv * rinv) : 0.0f;
        d[c] *= rinv;
#pragma unroll
        for (int c2 = c + 1; c2 < 32; ++c2) {
            d[c2] -= f * cv[c2];
            asm volatile("" : "+v"(d[c2]));  // right-looking on purpose: do not defer the updates
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c <= r) row[c] = d[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // lane r = column r of W = L^-1
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float *ri = Li + tri(c0 + i, 0) + c0;  // uniform address: broadcast reads
        float acc = (i == r) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < i; ++k) acc -= ri[k] * w[k];
        w[i] = acc * rinvs[i];
        asm volatile("" : "+v"(w[i]));  // computed here, not sunk into the predicated stores below
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 32) {
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i >= r) Li[tri(c0 + i, 0) + c0 + r] = w[i];
    }
    return ok;
}

__device__ __forceinline__ bool factor_invert_mfma(float *Li, int n, int tid, float *red, float *scratch, long long *st)
{
    long long t_last = 0, t_acc[5] = {0, 0, 0, 0, 0};
    auto lapf = [&](int idx) {
        if (st && tid == 0) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (idx >= 0) t_acc[idx] += now - t_last;
            t_last = now;
        }
    };
    lapf(-1);
    const int nb = n >> 5, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    float *flag = red + 7;
    if (tid == 0) flag[0] = 1.0f;
    __syncthreads();
    for (int J = 0; J < nb; ++J) {
        const int c0 = 32 * J;
        if (wv == 0) {
            const bool ok = diag_block_invert(Li, J, lane, scratch);
            if (!ok && lane == 0) flag[0] = 0.0f;
        }
        __syncthreads();
        lapf(0);
        // panel: L[I,J] = A[I,J] W_J'   (B operand B[k][c] = W_J[c][k], zero above the diagonal)
        for (int I = J + 1 + wv; I < nb; I += 4) {
            const float *arow = Li + tri(32 * I + l31, 0) + c0;
            const float *brow = Li + tri(c0 + l31, 0) + c0;
            float a[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) a[s2] = arow[2 * s2 + kh];
            f32x16 acc;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int k = 2 * s2 + kh;
                const float bl = brow[k];  // unconditional load, selected (see lower_matvec)
                const float bv = (k <= l31) ? bl : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) Li[tri(32 * I + crow(t, lane), 0) + c0 + l31] = acc[t];
        }
        __syncthreads();
        lapf(1);
        // trailing update: A[I,I2] -= L[I,J] L[I2,J]'
        int cnt = 0;
        for (int I = J + 1; I < nb; ++I)
            for (int I2 = J + 1; I2 <= I; ++I2, ++cnt) {
                if ((cnt & 3) != wv) continue;
                const float *arow = Li + tri(32 * I + l31, 0) + c0;
                const float *brow = Li + tri(32 * I2 + l31, 0) + c0;
                const bool dg = (I2 == I);
                f32x16 acc;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int rr = crow(t, lane);
                    const float cl = Li[tri(32 * I + rr, 0) + 32 * I2 + l31];
                    acc[t] = (!dg || l31 <= rr) ? cl : 0.0f;
                }
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
                    const int k = 2 * s2 + kh;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-arow[k], brow[k], acc, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int rr = crow(t, lane);
                    if (!dg || l31 <= rr) Li[tri(32 * I + rr, 0) + 32 * I2 + l31] = acc[t];
                }
            }
        __syncthreads();
        lapf(2);
    }
    // inverse, block row by block row
    for (int I = 1; I < nb; ++I) {
        const int r0 = 32 * I;
        // L'[I,K] = W_I L[I,K]   (A operand W_I[r][k], zero above the diagonal)
        for (int K = wv; K < I; K += 4) {
            const float *wrow = Li + tri(r0 + l31, 0) + r0;
            float bv[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) bv[s2] = Li[tri(r0 + 2 * s2 + kh, 0) + 32 * K + l31];
            f32x16 acc;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int k = 2 * s2 + kh;
                const float al = wrow[k];
                const float av = (k <= l31) ? al : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s2], acc, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) Li[tri(r0 + crow(t, lane), 0) + 32 * K + l31] = acc[t];
        }
        __syncthreads();
        lapf(3);
        // X[I,Jt] = - sum_K L'[I,K] X[K,Jt] : kept in registers until every wavefront has read row block I
        f32x16 acc2[2];
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            const int Jt = wv + 4 * qd;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc2[qd][t] = 0.0f;
            if (Jt < I) {
                for (int K = Jt; K < I; ++K) {
                    const float *arow = Li + tri(r0 + l31, 0) + 32 * K;
                    const bool dg = (K == Jt);
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2) {
                        const int k = 2 * s2 + kh;
                        const float bl = Li[tri(32 * K + k, 0) + 32 * Jt + l31];
                        const float bv = (!dg || l31 <= k) ? bl : 0.0f;
                        acc2[qd] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k], bv, acc2[qd], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            const int Jt = wv + 4 * qd;
            if (Jt < I) {
#pragma unroll
                for (int t = 0; t < 16; ++t) Li[tri(r0 + crow(t, lane), 0) + 32 * Jt + l31] = -acc2[qd][t];
            }
        }
        __syncthreads();
        lapf(4);
    }
    if (st && tid == 0)
        for (int u = 0; u < 5; ++u) st[u] = t_acc[u];
    return flag[0] != 0.0f;
}

// ------------------------------------------------------------------ blocked factor + inverse, float64 (mid-size kind)
// Same scheme as factor_invert_mfma with 16 x 16 blocks on v_mfma_f64_16x16x4_f64. The matrix is padded to a
// multiple of 16 with an identity block (chol(diag(P, I)) = diag(L, I)), which the caller provides: packed
// triangle of npad = 16 ceil(n / 16) rows. Operand maps: A lane l -> A[l & 15][l >> 4], B lane l ->
// B[l >> 4][l & 15]; C/D register t of lane l -> row (l >> 4) + 4 t, column l & 15 (NOT the f32 map).
using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ bool diag_block_invert64(double *Li, int J, int lane, double *colbuf)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    int r = lane & 15;
    asm volatile("" : "+v"(r));  // see diag_block_invert: keeps the lane predicates out of the caller's loops
    const int c0 = 16 * J;
    double *row = Li + tri(c0 + r, 0) + c0;
    double d[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double dl = row[c];
        d[c] = (c <= r) ? dl : 0.0;
    }
    bool ok = true;
    double *rinvs = colbuf + 16;  // 1 / L_cc, read back as broadcasts by the inverse
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (lane < 16) colbuf[r] = d[c];  // column c before scaling
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double cv[16];
#pragma unroll
        for (int g = c / 2; g < 8; ++g) {
            const d2 v = *reinterpret_cast<const d2 *>(colbuf + 2 * g);
            cv[2 * g] = v[0];
            cv[2 * g + 1] = v[1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const double piv = cv[c];
        ok = ok && (piv > 0.0);
        const double rinv = fast_rsqrt(piv);
        if (lane == 0) rinvs[c] = rinv;
        const double f = (r >= c) ? d[c] * (rinv * rinv) : 0.0;
        d[c] *= rinv;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) {
            d[c2] -= f * cv[c2];
            asm volatile("" : "+v"(d[c2]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= r) row[c] = d[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // lane r = column r of W = L^-1
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double *ri = Li + tri(c0 + i, 0) + c0;  // uniform address: broadcast reads
        double acc = (i == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) acc -= ri[k] * w[k];
        w[i] = acc * rinvs[i];
        asm volatile("" : "+v"(w[i]));
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i >= r) Li[tri(c0 + i, 0) + c0 + r] = w[i];
    }
    return ok;
}

// npad: multiple of 16 (rows n..npad-1 of the packed triangle hold the identity); four wavefronts.
__device__ __forceinline__ bool factor_invert_mfma64(double *Li, int npad, int tid, double *red, double *scratch)
{
    const int nb = npad >> 4, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    double *flag = red + 7;
    if (tid == 0) flag[0] = 1.0;
    __syncthreads();
    for (int J = 0; J < nb; ++J) {
        const int c0 = 16 * J;
        if (wv == 0) {
            const bool ok = diag_block_invert64(Li, J, lane, scratch);
            if (!ok && lane == 0) flag[0] = 0.0;
        }
        __syncthreads();
        // panel: L[I,J] = A[I,J] W_J'
        for (int I = J + 1 + wv; I < nb; I += 4) {
            const double *arow = Li + tri(16 * I + l15, 0) + c0;
            const double *brow = Li + tri(c0 + l15, 0) + c0;
            double a[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) a[s2] = arow[4 * s2 + kq];
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int k = 4 * s2 + kq;
                const double bl = brow[k];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], (k <= l15) ? bl : 0.0, acc, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) Li[tri(16 * I + kq + 4 * t, 0) + c0 + l15] = acc[t];
        }
        __syncthreads();
        // trailing update: A[I,I2] -= L[I,J] L[I2,J]'
        int cnt = 0;
        for (int I = J + 1; I < nb; ++I)
            for (int I2 = J + 1; I2 <= I; ++I2, ++cnt) {
                if ((cnt & 3) != wv) continue;
                const double *arow = Li + tri(16 * I + l15, 0) + c0;
                const double *brow = Li + tri(16 * I2 + l15, 0) + c0;
                const bool dg = (I2 == I);
                f64x4 acc;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = kq + 4 * t;
                    const double cl = Li[tri(16 * I + rr, 0) + 16 * I2 + l15];
                    acc[t] = (!dg || l15 <= rr) ? cl : 0.0;
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int k = 4 * s2 + kq;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-arow[k], brow[k], acc, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = kq + 4 * t;
                    if (!dg || l15 <= rr) Li[tri(16 * I + rr, 0) + 16 * I2 + l15] = acc[t];
                }
            }
        __syncthreads();
    }
    // inverse, block row by block row
    for (int I = 1; I < nb; ++I) {
        const int r0 = 16 * I;
        for (int K = wv; K < I; K += 4) {  // L'[I,K] = W_I L[I,K]
            const double *wrow = Li + tri(r0 + l15, 0) + r0;
            double bv[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) bv[s2] = Li[tri(r0 + 4 * s2 + kq, 0) + 16 * K + l15];
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int k = 4 * s2 + kq;
                const double al = wrow[k];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((k <= l15) ? al : 0.0, bv[s2], acc, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) Li[tri(r0 + kq + 4 * t, 0) + 16 * K + l15] = acc[t];
        }
        __syncthreads();
        // X[I,Jt] = - sum_K L'[I,K] X[K,Jt]: in registers until every wavefront has read row block I
        f64x4 acc2[3];
#pragma unroll
        for (int qd = 0; qd < 3; ++qd) {
            const int Jt = wv + 4 * qd;
            acc2[qd] = f64x4{0.0, 0.0, 0.0, 0.0};
            if (Jt < I) {
                for (int K = Jt; K < I; ++K) {
                    const double *arow = Li + tri(r0 + l15, 0) + 16 * K;
                    const bool dg = (K == Jt);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        const int k = 4 * s2 + kq;
                        const double bl = Li[tri(16 * K + k, 0) + 16 * Jt + l15];
                        acc2[qd] = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k], (!dg || l15 <= k) ? bl : 0.0, acc2[qd], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int qd = 0; qd < 3; ++qd) {
            const int Jt = wv + 4 * qd;
            if (Jt < I) {
#pragma unroll
                for (int t = 0; t < 4; ++t) Li[tri(r0 + kq + 4 * t, 0) + 16 * Jt + l15] = -acc2[qd][t];
            }
        }
        __syncthreads();
    }
    return flag[0] != 0.0;
}

// quad-local sums (lanes 4q .. 4q+3) on the DPP path
__device__ __forceinline__ float quad_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ double quad_sum(double v)
{
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

// Workspace per problem (elements of T): MA [n][n], Tm [n][n].
//
// STRUCT = true: G is never formed. The constraint matrix of a condensed MPC problem is
// G = blockrows(C_k Psi_k + D_k E_k) (qpmpc/mpc_qp.py:62-78), so
//   * G v (slack updates) is one roll-out  dx_{k+1} = A_k dx_k + B_k v_k  (one wavefront, operands
//     prefetched eight steps ahead) followed by  (G v)_{k,r} = C_k[r] dx_k + D_k[r] v_k : ~0.1 MB of
//     operands per product instead of streaming 1 MB of G;
//   * a row G_p (only for the selected constraint) is C_k[r] Psi_k + D_k[r] E_k from the Psi blocks
//     the Gram kernel needed anyway;
//   * 1/|G_i| comes from the propagation kernel (aux2).
// aux = G' (dense mode) or Psi_all (structured mode).
//
// KIND = K_MID: mid-size problems (n up to ~128, BASELINE config 3: n = 50, m = 100, N = 50), fused
// build + solve in ONE launch with a small LDS footprint (several problems per CU):
//   * the problem's operands A, B, C, D are staged into LDS once (a few KB);
//   * front end: Psi_k (nx x n, two LDS buffers) is propagated step by step and CONSUMED on the fly --
//     P += w_k Psi_k' Psi_k straight into the packed triangle, q += w_k Psi_k' resid_k, h_k, and the row
//     norms from S_k = Psi_k Psi_k' -- so neither Psi (N nx n) nor G (m n) nor M = G L^-T ever exists;
//   * the same solver as above, G applied through the roll-out (operands from LDS), a selected row
//     G_p = C_k[r] Psi_k + D_k[r] E_k rebuilt by the adjoint recursion mu' <- mu' A_j, g_j = B_j' mu.
enum { K_DENSE = 0, K_STRUCT = 1, K_MID = 2 };

template <typename T, int KIND, int NWV>
__global__ void __launch_bounds__(64 * NWV, (KIND == 2 && NWV == 4 && sizeof(T) == 8) ? 4 : 1) mpcqp_bigsolve_kernel(const KernelArgs ka, const T *__restrict__ Pall,
                                                            const T *__restrict__ qall, const T *__restrict__ Gall,
                                                            const T *__restrict__ aux, const T *__restrict__ hall,
                                                            const T *__restrict__ aux2, T *__restrict__ wsall)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = ka.n, m = ka.m, tid = threadIdx.x;
    const int64_t prob = blockIdx.x;
    const T INF = (T)HUGE_VALF;
    typedef T V4 __attribute__((ext_vector_type(4)));
    constexpr bool STRUCT = (KIND == K_STRUCT), MID = (KIND == K_MID), MFREE = (KIND != K_DENSE);
    constexpr int BS = 64 * NWV;  // threads per problem: 256, or 128 for mid-size problems with n <= 128
    static_assert(NWV == 4 || KIND == K_MID, "the large-problem kinds are laid out for four wavefronts");
    // LDS carve
    T *Li = (T *)smem_raw;              // packed lower triangle: P -> L -> L^-1
    const int npad = MID ? ((n + 15) & ~15) : n;  // mid-size kind: triangle padded to 16-row blocks (identity rows)
    T *sv = Li + npad * (npad + 1) / 2;  // slacks            [m]
    T *gin = sv + m;                    // 1 / |G_i|          [m]
    T *tolv = gin + m;                  // tol (1 + |h_i|)    [m]
    T *y0 = tolv + m;                   // -L^-1 q            [n]
    T *mp = y0 + n;                     // M_p                [n]
    T *zv = mp + n;                     // z (y coordinates)  [n]
    T *zx = zv + n;                     // L^-T z             [n]
    T *rv = zx + n;                     // r by slot          [n]
    T *lam = rv + n;                    // multipliers by slot[n]
    T *tmp = lam + n;                   // scratch            [n]
    T *red = tmp + n;                   // reductions         [8]
    T *dxs = red + 8;                   // roll-out states [N][nx] (structured mode only)
    // mid-size kind: the operands of every step, two Psi_k buffers, small front-end vectors
    const int sAk = MID ? (ka.A.step_stride ? ka.nx * ka.nx : 0) : 0, sBk = MID ? (ka.B.step_stride ? ka.nx * ka.nu : 0) : 0;
    const int sCk = MID ? (ka.C.step_stride ? ka.mk * ka.nx : 0) : 0, sDk = MID ? (ka.D.step_stride ? ka.mk * ka.nu : 0) : 0;
    T *opA = dxs + (MFREE ? ka.N * ka.nx : 0);
    T *opB = opA + (MID ? (sAk ? ka.N : 1) * ka.nx * ka.nx : 0);
    T *opC = opB + (MID ? (sBk ? ka.N : 1) * ka.nx * ka.nu : 0);
    T *opD = opC + ((MID && ka.C.ptr) ? (sCk ? ka.N : 1) * ka.mk * ka.nx : 0);
    T *psiA = opD + ((MID && ka.D.ptr) ? (sDk ? ka.N : 1) * ka.mk * ka.nu : 0);
    T *psiB = psiA + (MID ? ka.nx * n : 0);
    T *fe = psiB + (MID ? ka.nx * n : 0);  // phi [2 nx], resid [2 nx], S [2 nx nx], T1 [nx nx], mu [2 nx]
    int *act = (int *)(fe + (MID ? 6 * ka.nx + 3 * ka.nx * ka.nx : 0));  // constraint of slot [n]
    int *pos = act + n;                 // slot of constraint, -1 [m]
    int *redi = pos + m;                // [4]

    const T *P = Pall + prob * (int64_t)n * n;
    const T *q = qall + prob * (int64_t)n;
    const T *G = MFREE ? nullptr : Gall + prob * (int64_t)m * n;
    const T *GT = MFREE ? nullptr : aux + prob * (int64_t)m * n;
    const T *h = MID ? nullptr : hall + prob * (int64_t)m;
    // structured mode: the problem's own operands
    const int nx = ka.nx, nu = ka.nu, N = ka.N, mk = ka.mk;
    const T *gA = MFREE ? (const T *)ka.A.ptr + prob * ka.A.batch_stride : nullptr;
    const T *gB = MFREE ? (const T *)ka.B.ptr + prob * ka.B.batch_stride : nullptr;
    const T *gC = (MFREE && ka.C.ptr) ? (const T *)ka.C.ptr + prob * ka.C.batch_stride : nullptr;
    const T *gD = (MFREE && ka.D.ptr) ? (const T *)ka.D.ptr + prob * ka.D.batch_stride : nullptr;
    const T *Psi = STRUCT ? aux + prob * (int64_t)(N + 1) * nx * n : nullptr;
    const T *nrm = STRUCT ? aux2 + prob * (int64_t)m : nullptr;

    // Structured mode keeps the problem's operands in REGISTERS for the whole solve (they are
    // read once): wavefront w holds A_k, B_k for its 16 steps k = 16 w + d, lane (r, c) =
    // (lane / 4, lane % 4) taking row r and every fourth column from c; thread t holds the rows
    // C_k[r], D_k[r] of its four constraints i = t + 256 j.
    constexpr int SPW = 16, RS = 4;
    T ra[SPW][4], rb[SPW][2], rc[RS][16], rd[RS][8];
    int krow[RS];
    auto load_operands = [&]() __attribute__((always_inline)) {
        const int lane = tid & 63, wv = tid >> 6, r = lane >> 2, c = lane & 3;
#pragma unroll
        for (int d = 0; d < SPW; ++d) {
            const int k = wv * SPW + d;
            const bool live = (k < N - 1) && (r < nx);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int sc = c + 4 * jj;
                ra[d][jj] = (live && sc < nx) ? gA[(int64_t)k * ka.A.step_stride + r * nx + sc] : T(0);
                if (jj < 2) rb[d][jj] = (live && sc < nu) ? gB[(int64_t)k * ka.B.step_stride + r * nu + sc] : T(0);
            }
        }
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const int i = tid + BS * j;
            const bool live = i < m;
            const int k = live ? i / mk : 0, r = live ? i - k * mk : 0;
            krow[j] = k;
#pragma unroll
            for (int sc = 0; sc < 16; ++sc)
                rc[j][sc] = (live && gC && sc < nx) ? gC[(int64_t)k * ka.C.step_stride + r * nx + sc] : T(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                rd[j][u] = (live && gD && u < nu) ? gD[(int64_t)k * ka.D.step_stride + r * nu + u] : T(0);
        }
    };
    // dxs[k] = Psi_k zx for k < N: dx_{k+1} = A_k dx_k + B_k zx_k, the wavefronts taking turns
    auto rollout = [&]() __attribute__((always_inline)) {
        if constexpr (MID) {
            // operands in LDS: the first wavefront walks the horizon, lane r = row r of the state
            if (tid < 64) {
                if (tid < nx) dxs[tid] = T(0);
                for (int k = 0; k < N - 1; ++k) {
                    T acc = T(0);
                    if (tid < nx) {
                        const T *ar = opA + k * sAk + tid * nx, *br = opB + k * sBk + tid * nu;
                        for (int sc = 0; sc < nx; ++sc) acc += ar[sc] * dxs[k * nx + sc];
                        for (int u = 0; u < nu; ++u) acc += br[u] * zx[k * nu + u];
                        dxs[(k + 1) * nx + tid] = acc;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            __syncthreads();
            return;
        }
        const int lane = tid & 63, wv = tid >> 6, r = lane >> 2, c = lane & 3;
        if (tid < nx) dxs[tid] = T(0);
        for (int w = 0; w < NWV; ++w) {
            __syncthreads();
            if (wv == w) {
                // The state travels from step to step in REGISTERS: after the quad sum every lane of quad r
                // holds dx_{k+1}[r]; lane (r, c) fetches the entries c, c+4, c+8, c+12 it multiplies next from
                // quads c + 4 jj with ds_bpermute (no LDS round trip on the critical path). dxs is still
                // written -- the constraint rows and the next wavefront read it -- but nobody waits for it here.
                T xc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) xc[jj] = dxs[(w * SPW) * nx + min(c + 4 * jj, nx - 1)];
#pragma unroll
                for (int d = 0; d < SPW; ++d) {
                    const int k = w * SPW + d;
                    if (k < N - 1) {
                        T acc = T(0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int sc = c + 4 * jj;  // operands past nx / nu are zero: clamp the index
                            acc += ra[d][jj] * xc[jj];
                            if (jj < 2) acc += rb[d][jj] * zx[k * nu + min(sc, nu - 1)];
                        }
                        acc = quad_sum(acc);
                        if (c == 0 && r < nx) dxs[(k + 1) * nx + r] = acc;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) xc[jj] = __shfl(acc, 4 * min(c + 4 * jj, 15));
                    }
                }
            }
        }
        __syncthreads();
    };
    // (G zx)_i for the thread's j-th constraint, from the roll-out states
    auto struct_row = [&](int j) __attribute__((always_inline)) -> T {
        const int k = krow[j];
        T acc = T(0);
#pragma unroll
        for (int sc = 0; sc < 16; ++sc) acc += rc[j][sc] * dxs[k * nx + min(sc, nx - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += rd[j][u] * zx[k * nu + min(u, nu - 1)];
        return acc;
    };
    // (G zx)_i with the operands in LDS (mid-size kind)
    auto mid_row = [&](int i) __attribute__((always_inline)) -> T {
        const int k = i / mk, r = i - k * mk;
        T acc = T(0);
        if (gC) {
            const T *cr = opC + k * sCk + r * nx;
            for (int sc = 0; sc < nx; ++sc) acc += cr[sc] * dxs[k * nx + sc];
        }
        if (gD) {
            const T *dr = opD + k * sDk + r * nu;
            for (int u = 0; u < nu; ++u) acc += dr[u] * zx[k * nu + u];
        }
        return acc;
    };
    T *MA = wsall + prob * ws_stride_elems(n);
    T *Tm = MA + (int64_t)n * n;
    T *oU = (T *)ka.U + prob * (int64_t)n;
    const T tol = (T)ka.tol;
    int status = MPCQP_MAX_ITER, iters = 0;
    // optional phase timestamps (tools/probe_big_phases.py): MpcqpSolveOpts.probe -> long long[8] per problem
    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 32 : nullptr;
    auto mark = [&](int slot) {
        if (stamp && tid == 0) stamp[slot] = (long long)__builtin_readcyclecounter();
    };
    long long lap_t = 0, lap_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // per-phase totals inside the loop
    auto lap = [&](int idx) {
        if (stamp && tid == 0) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (idx >= 0) lap_acc[idx] += now - lap_t;
            lap_t = now;
        }
    };
    mark(0);

    // dst = L^-1 src : thread j takes row j of the packed inverse (k uniform across the wavefront)
    auto lower_matvec = [&](const T *src) __attribute__((always_inline)) -> T {
        T a0 = T(0), a1 = T(0);
        if (tid < n) {
            const T *rj = Li + tri(tid, 0);
            const int kend = __builtin_amdgcn_readfirstlane(min(n, (tid | 63) + 1));  // the wavefront's longest row (uniform)
            int k = 0;
#pragma unroll 4
            // loads are unconditional (past the diagonal they land in later rows: finite values,
            // selected away) so that the compiler can batch them instead of branching per element
            for (; k + 1 < kend; k += 2) {
                const T l0 = rj[k], l1 = rj[k + 1];
                a0 += ((k <= tid) ? l0 : T(0)) * src[k];
                a1 += ((k + 1 <= tid) ? l1 : T(0)) * src[k + 1];
            }
            if (k < kend) {
                const T l0 = rj[k];
                a0 += ((k <= tid) ? l0 : T(0)) * src[k];
            }
        }
        return a0 + a1;
    };
    // dst = L^-T src : thread j takes column j (row index uniform, row base in scalar registers)
    auto upper_matvec = [&](const T *src) __attribute__((always_inline)) -> T {
        T a0 = T(0), a1 = T(0);
        if (tid < n) {
            int i = __builtin_amdgcn_readfirstlane(tid & ~63);  // uniform: keeps the loop scalar
            int rowi = tri(i, 0);
#pragma unroll 4
            for (; i + 1 < n; i += 2) {
                const T l0 = Li[rowi + tid];
                rowi += i + 1;
                const T l1 = Li[rowi + tid];
                rowi += i + 2;
                a0 += ((i >= tid) ? l0 : T(0)) * src[i];
                a1 += ((i + 1 >= tid) ? l1 : T(0)) * src[i + 1];
            }
            if (i < n) {
                const T l0 = Li[rowi + tid];
                a0 += ((i >= tid) ? l0 : T(0)) * src[i];
            }
        }
        return a0 + a1;
    };

    if constexpr (MID) {
        // ---------------------------------------------------------------- on-chip front end
        const T wxp = (ka.flags & MPCQP_P_STAGE) ? (T)ka.wx : T(0), wtp = (ka.flags & MPCQP_P_TERMINAL) ? (T)ka.wt : T(0);
        const T *gx0 = (const T *)ka.x0.ptr + prob * ka.x0.batch_stride;
        const T *ggoal = ka.goal.ptr ? (const T *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
        const T *gtgt = ka.targets.ptr ? (const T *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
        const T *ge = (const T *)ka.e.ptr + prob * ka.e.batch_stride;
        const bool qs = (ka.flags & MPCQP_Q_STAGE) && gtgt, qt = (ka.flags & MPCQP_Q_TERMINAL) && ggoal;
        const T wxq = qs ? (T)ka.wx : T(0), wtq = qt ? (T)ka.wt : T(0);
        T *phi = fe, *resid = fe + 2 * nx, *Sm = resid + 2 * nx, *T1 = Sm + 2 * nx * nx;
        // stage the operands (LTI operands once)
        for (int i = tid; i < (sAk ? N : 1) * nx * nx; i += BS) opA[i] = gA[i];
        for (int i = tid; i < (sBk ? N : 1) * nx * nu; i += BS) opB[i] = gB[i];
        if (gC)
            for (int i = tid; i < (sCk ? N : 1) * mk * nx; i += BS) opC[i] = gC[i];
        if (gD)
            for (int i = tid; i < (sDk ? N : 1) * mk * nu; i += BS) opD[i] = gD[i];
        for (int i = tid; i < npad * (npad + 1) / 2; i += BS) Li[i] = T(0);
        for (int i = tid; i < nx * n; i += BS) psiA[i] = T(0);
        for (int i = tid; i < nx * nx; i += BS) Sm[i] = T(0);
        // e and the reference trajectory go to LDS as well (no global load inside the step loop):
        // e into tolv (turned into h in place), the targets into the roll-out table (free until the solve)
        for (int i = tid; i < m; i += BS) tolv[i] = ge[(int64_t)(i / mk) * ka.e.step_stride + (i % mk)];
        if (qs)
            for (int i = tid; i < N * nx; i += BS) dxs[i] = gtgt[i];
        T goal_l = T(0);  // goal entry of lane l of the last wavefront
        if (qt && tid >= BS - 64 && tid - (BS - 64) < nx) goal_l = ggoal[tid - (BS - 64)];
        if (tid < nx) {
            const T x = gx0[tid];
            phi[tid] = x;
            resid[tid] = x - (qs ? gtgt[tid] : T(0));  // resid_0 (multiplies Psi_0 = 0)
        }
        // entries of the packed triangle are shared out as (row i, every NC-th column from c)
        const int NC = max(1, BS / n), pi = tid % n, pc = tid / n;
        // ... or, when the lower triangle has at most BS tiles of 4 x 4 (n <= 88), as one register tile each
        const int T4 = (n + 3) / 4;
        const bool tiled = T4 * (T4 + 1) / 2 <= BS;
        int ti = 0, tj = 0;
        {
            int rem = tid;
            while (rem > ti) {  // tile number tid -> (ti, tj), tj <= ti
                rem -= ti + 1;
                ++ti;
            }
            tj = rem;
        }
        const bool tlive = tiled && ti < T4;
        T pacc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) pacc[u][v2] = T(0);
        T qacc = T(0);
        const int tq = (2 * n <= BS) ? tid - BS / 2 : tid;          // column of q this thread accumulates
        const int th = (mk <= 64 && BS >= 192) ? tid - 128 : tid;   // first h row of this thread
        __syncthreads();
        // ONE barrier per step: everything step k reads (Psi_k, Phi_k x0, resid_k, S_k = Psi_k Psi_k') was
        // written by step k-1 into the other half of a double buffer
        long long wv_busy = 0;  // dev probe: cycles this wavefront works per step (excluding the barrier)
        for (int k = 0; k <= N; ++k) {
            const long long t_in = stamp ? (long long)__builtin_readcyclecounter() : 0;
            T *cur = (k & 1) ? psiB : psiA, *nxt = (k & 1) ? psiA : psiB;
            const T *ph = phi + (k & 1) * nx, *rc = resid + (k & 1) * nx, *Sc = Sm + (k & 1) * nx * nx;
            T *phn = phi + ((k + 1) & 1) * nx, *rn = resid + ((k + 1) & 1) * nx, *Sn = Sm + ((k + 1) & 1) * nx * nx;
            const int ncol = min(n, k * nu);  // Psi_k is zero from column k nu on
            const bool last = (k == N);
            // P += w_k Psi_k' Psi_k (lower triangle, packed) ; q += w_k Psi_k' resid_k
            const T wp = last ? wtp : wxp, wq = last ? wtq : wxq;
            if (tiled) {
                // 4 x 4 register tile of P per thread: per state row 8 LDS reads feed 16 FMAs, and the
                // accumulators never leave the registers until the triangle is complete
                if (wp != T(0) && tlive && 4 * tj < ncol) {
                    for (int sc = 0; sc < nx; ++sc) {
                        const T *pr = cur + sc * n;
                        T a4[4], b4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            a4[u] = wp * pr[min(4 * ti + u, n - 1)];
                            b4[u] = pr[min(4 * tj + u, n - 1)];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v2 = 0; v2 < 4; ++v2) pacc[u][v2] += a4[u] * b4[v2];
                    }
                }
            } else if (wp != T(0) && pc < NC && pi < ncol) {
                // padded to 4, 8 or 16 state rows so that the coefficients stay in registers
                auto accumulate = [&](auto nxc_tag) __attribute__((always_inline)) {
                    constexpr int NXC = decltype(nxc_tag)::value;
                    T ci[NXC];
#pragma unroll
                    for (int sc = 0; sc < NXC; ++sc) ci[sc] = (sc < nx) ? wp * cur[min(sc, nx - 1) * n + pi] : T(0);
                    T *row = Li + tri(pi, 0);
                    for (int j = pc; j <= pi; j += NC) {
                        T acc = T(0);
#pragma unroll
                        for (int sc = 0; sc < NXC; ++sc) acc += ci[sc] * cur[min(sc, nx - 1) * n + j];
                        row[j] += acc;
                    }
                };
                if (nx <= 4)
                    accumulate(std::integral_constant<int, 4>{});
                else if (nx <= 8)
                    accumulate(std::integral_constant<int, 8>{});
                else
                    accumulate(std::integral_constant<int, 16>{});
            }
            // (the per-step jobs are spread over the wavefronts -- tiles from thread 0 up, q and the Psi update
            //  from the top down, h / norms in the third wavefront, the small recursions in the last -- so that no
            //  wavefront runs all of them back to back)
            if (tq >= 0 && tq < ncol && wq != T(0)) {
                T acc = T(0);
                for (int sc = 0; sc < nx; ++sc) acc += cur[sc * n + tq] * rc[sc];
                qacc += wq * acc;
            }
            if (!last) {
                const T *Ak = opA + k * sAk, *Bk = opB + k * sBk;
                // h_k = e_k - C_k Phi_k x0 (kept in tolv until the tolerances are formed) ; 1/|G_i| with
                // |G_i|^2 = C_i S_k C_i' + |D_i|^2
                for (int r = th; r >= 0 && r < mk; r += BS) {
                    T acc = T(0), nn = T(0);
                    if (gC) {
                        const T *cr = opC + k * sCk + r * nx;
                        for (int a = 0; a < nx; ++a) {
                            acc += cr[a] * ph[a];
                            T t1 = T(0);
                            for (int b = 0; b < nx; ++b) t1 += Sc[a * nx + b] * cr[b];
                            nn += cr[a] * t1;
                        }
                    }
                    if (gD) {
                        const T *dr = opD + k * sDk + r * nu;
                        for (int u = 0; u < nu; ++u) nn += dr[u] * dr[u];
                    }
                    tolv[k * mk + r] -= acc;
                    gin[k * mk + r] = (nn > T(0)) ? fast_rsqrt(nn) : T(1);
                }
                // Psi_{k+1} = A_k Psi_k, block k <- B_k
                for (int e2 = BS - 1 - tid; e2 < nx * n; e2 += BS) {
                    const int a = e2 / n, c = e2 - a * n;
                    T acc = T(0);
                    const int jb = c / nu;
                    if (jb == k) {
                        acc = Bk[a * nu + (c - jb * nu)];
                    } else if (c < ncol) {
                        for (int b = 0; b < nx; ++b) acc += Ak[a * nx + b] * cur[b * n + c];
                    }
                    nxt[e2] = acc;
                }
                // the last wavefront: Phi_{k+1} x0, resid_{k+1}, S_{k+1} = A_k S_k A_k' + B_k B_k'
                if (tid >= BS - 64) {
                    const int l = tid - (BS - 64);
                    if (l < nx) {
                        T acc = T(0);
                        for (int b = 0; b < nx; ++b) acc += Ak[l * nx + b] * ph[b];
                        phn[l] = acc;
                        T ref = T(0);
                        if (k + 1 < N) {
                            if (qs) ref = dxs[(k + 1) * nx + l];
                        } else {
                            ref = goal_l;
                        }
                        rn[l] = acc - ref;
                    }
                    // S is only needed for the norms of rows with a state part: skipped without C (config 3)
                    for (int e2 = l; gC && e2 < nx * nx; e2 += 64) {
                        const int a = e2 / nx, b = e2 - a * nx;
                        T acc = T(0);
                        for (int u = 0; u < nx; ++u) acc += Ak[a * nx + u] * Sc[u * nx + b];
                        T1[e2] = acc;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (int e2 = l; gC && e2 < nx * nx; e2 += 64) {
                        const int a = e2 / nx, b = e2 - a * nx;
                        T acc = T(0);
                        for (int u = 0; u < nx; ++u) acc += T1[a * nx + u] * Ak[b * nx + u];
                        for (int u = 0; u < nu; ++u) acc += Bk[a * nu + u] * Bk[b * nu + u];
                        Sn[e2] = acc;
                    }
                }
            }
            if (stamp) wv_busy += (long long)__builtin_readcyclecounter() - t_in;
            __syncthreads();
        }
        if (stamp && (tid & 63) == 0) stamp[24 + (tid >> 6)] = wv_busy;
        if (tlive) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v2 = 0; v2 < 4; ++v2) {
                    const int i = 4 * ti + u, j = 4 * tj + v2;
                    if (i < n && j <= i) Li[tri(i, j)] = pacc[u][v2];
                }
        }
        __syncthreads();
        if (tq >= 0 && tq < n) tmp[tq] = qacc;  // q, consumed below
        if (tid < n) Li[tri(tid, tid)] += (T)ka.wu;
        for (int i = n + tid; i < npad; i += BS) Li[tri(i, i)] = T(1);
    } else {
        // ---- packed lower triangle of P
        if ((n & 3) == 0) {
            // whole rows as 16-byte loads, sixteen in flight per thread. Branch-free so that the loads
            // batch: groups above the diagonal re-read their row's diagonal group and store to a dump
            // slot (the vectors after y0 are not in use yet; n >= 64 here, see bigsolve_supported).
            const int nv = n * n / 4;
            T *dump = y0 + tid;
    #pragma unroll 16
            for (int v4 = tid; v4 < nv; v4 += BS) {
                const int idx = 4 * v4, i = idx / n, j = idx - i * n;
                const bool need = j <= i;
                const V4 pv = *reinterpret_cast<const V4 *>(P + (need ? idx : i * n + (i & ~3)));
                T *d = Li + tri(i, j);
    #pragma unroll
                for (int c = 0; c < 4; ++c) *((need && j + c <= i) ? d + c : dump) = pv[c];
            }
        } else {
    #pragma unroll 8
            for (int i = 0; i < n; ++i)
                if (tid <= i) Li[tri(i, tid)] = P[(int64_t)i * n + tid];
        }
    }
    __syncthreads();
    mark(1);
    // ---- L = chol(P), then L^-1 in place
    auto factor_small_or_scalar = [&](T *A, int nn, int t, T *rd) __attribute__((always_inline)) -> bool {
        if (nn > 64) return factor_invert_scalar<T, false>(A, nn, t, rd);
        // one wavefront owns every row: no workgroup barrier inside the factorisation
        if (t < 64) {
            const bool ok = factor_invert_scalar<T, true>(A, nn, t, rd);
            if (t == 0) rd[7] = ok ? T(1) : T(0);
        }
        __syncthreads();
        return rd[7] != T(0);
    };
    bool pd;
    if constexpr (sizeof(T) == 4) {
        if constexpr (NWV == 4) {
            if ((n & 31) == 0)
                pd = factor_invert_mfma(Li, n, tid, red, sv, stamp ? stamp + 16 : nullptr);
            else
                pd = factor_small_or_scalar(Li, n, tid, red);
        } else {
            pd = factor_small_or_scalar(Li, n, tid, red);
        }
    } else {
        if constexpr (MID && NWV == 4)
            pd = factor_invert_mfma64(Li, npad, tid, red, mp);  // mp (n >= 32 entries) is free until the first iteration
        else if constexpr (NWV == 4)
            pd = ((n & 15) == 0 && n <= 192) ? factor_invert_mfma64(Li, n, tid, red, mp) : factor_small_or_scalar(Li, n, tid, red);
        else
            pd = factor_small_or_scalar(Li, n, tid, red);
    }
    if (!pd) {
        status = MPCQP_NOT_PD;
    } else {
        mark(3);
        if constexpr (STRUCT) load_operands();  // after the factorisation: 190 registers it should not have to carry
        // ---- y0 = -L^-1 q ; slacks at the unconstrained minimiser need x0 = L^-T y0
        if constexpr (!MID) {
            if (tid < n) tmp[tid] = q[tid];
        }
        __syncthreads();
        {
            const T a = lower_matvec(tmp);
            if (tid < n) y0[tid] = -a;
        }
        __syncthreads();
        {
            const T a = upper_matvec(y0);
            if (tid < n) zx[tid] = a;  // unconstrained minimiser in x coordinates
        }
        __syncthreads();
        if constexpr (MID) {
            rollout();
            for (int i = tid; i < m; i += BS) {
                const T hi = tolv[i];  // h was parked here by the front end
                sv[i] = hi - mid_row(i);
                // (no room for h itself in LDS: the tolerance carries its sign, the acceptance test at the end recovers it)
                tolv[i] = (hi < T(1e29)) ? (T)copysign((double)(tol + tol * fabs(hi)), (double)hi) : INF;
                pos[i] = -1;
            }
        } else if constexpr (STRUCT) {
            rollout();
#pragma unroll
            for (int j = 0; j < RS; ++j) {
                const int i = tid + BS * j;
                if (i < m) {
                    const T hi = h[i];
                    sv[i] = hi - struct_row(j);
                    gin[i] = nrm[i];
                    tolv[i] = (hi < T(1e29)) ? tol + tol * fabs(hi) : INF;
                    pos[i] = -1;
                }
            }
        } else {
            if ((m & 3) == 0) {
                // four consecutive rows per thread, 16-byte loads of G', eight k in flight
                for (int i4 = tid; i4 < (m >> 2); i4 += BS) {
                    V4 a = {T(0), T(0), T(0), T(0)}, nn = {T(0), T(0), T(0), T(0)};
    #pragma unroll 8
                    for (int k = 0; k < n; ++k) {
                        const V4 g = *reinterpret_cast<const V4 *>(GT + (int64_t)k * m + 4 * i4);
                        a += g * zx[k];
                        nn += g * g;
                    }
    #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = 4 * i4 + c;
                        const T hi = h[i];
                        sv[i] = hi - a[c];
                        gin[i] = (nn[c] > T(0)) ? T(1) / sqrt(nn[c]) : T(1);
                        tolv[i] = (hi < T(1e29)) ? tol + tol * fabs(hi) : INF;  // padded rows are never selected
                        pos[i] = -1;
                    }
                }
            } else {
                for (int i = tid; i < m; i += BS) {
                    T a = T(0), nn = T(0);
                    for (int k = 0; k < n; ++k) {
                        const T g = GT[(int64_t)k * m + i];
                        a += g * zx[k];
                        nn += g * g;
                    }
                    const T hi = h[i];
                    sv[i] = hi - a;
                    gin[i] = (nn > T(0)) ? T(1) / sqrt(nn) : T(1);
                    tolv[i] = (hi < T(1e29)) ? tol + tol * fabs(hi) : INF;
                    pos[i] = -1;
                }
            }
        }
        if (tid < n) lam[tid] = T(0);
        __syncthreads();

        mark(4);
        int nq = 0;       // number of occupied slots (slots are compact here: 0..nq-1)
        const int max_iter = ka.max_iter;
        bool fail = false;
        // y = y0 - M_A' lam ; u = L^-T y
        auto primal_point = [&]() __attribute__((always_inline)) {
            __syncthreads();
            if (tid < n) {
                T yk = y0[tid];
                for (int a = 0; a < nq; ++a) yk -= lam[a] * MA[(int64_t)a * n + tid];
                zv[tid] = yk;
            }
            __syncthreads();
            {
                const T a = upper_matvec(zv);
                if (tid < n) zx[tid] = a;
            }
            __syncthreads();
        };
        int reeval = 0;  // (mid-size kind) from-scratch evaluations of the slacks that found a violated row
        for (;;) {
            lap(-1);
            // ---- select the violated row farthest from its hyperplane
            T best = INF;
            int bi = 0x7fffffff;
            for (int i = tid; i < m; i += BS) {
                const T s = sv[i];
                if (pos[i] < 0 && s < -(T)fabs(tolv[i])) {
                    const T key = s * gin[i];
                    if (key < best) {
                        best = key;
                        bi = i;
                    }
                }
            }
            block_argmin<NWV>(best, bi, red, redi, tid);
            if (!(best < INF)) {
                if constexpr (MID) {
                    // The slacks of the inactive rows are carried along by increments: after hundreds of iterations of a
                    // degenerate problem they drift (a stress run found plans with rows violated by 5e-7 accepted). Before a
                    // point is accepted they are evaluated FROM SCRATCH -- like the small-problem and the stage-wise kernels
                    // do --, and the loop goes on from the fresh values if a row is violated after all.
                    if (iters >= 32 && reeval < 4) {  // (drift needs many updates: short solves are accepted as they stand)
                        primal_point();
                        rollout();
                        bool dirty = false;
                        for (int i = tid; i < m; i += BS) {
                            if (pos[i] < 0) {
                                const T tv = tolv[i];
                                if (tv < INF && tv > -INF) {
                                    const T hi = (T)copysign((double)(((T)fabs(tv) - tol) / tol), (double)tv);
                                    const T fresh = hi - mid_row(i);
                                    sv[i] = fresh;
                                    dirty |= !(fresh >= -T(4) * (T)fabs(tv));
                                }
                            }
                        }
                        if (__syncthreads_or(dirty)) {
                            ++reeval;
                            continue;
                        }
                    }
                }
                status = MPCQP_SOLVED;
                break;
            }
            const int p = bi;
            lap(0);
            // ---- M_p = L^-1 G_p'   (thread j: row j of L^-1 against G_p)
            __syncthreads();
            if constexpr (MID) {
                // G_p = C_k[r] Psi_k + D_k[r] E_k by the adjoint recursion (first wavefront):
                // mu = C_k[r]' ; for j = k-1 .. 0 : G_p[block j] = B_j' mu, mu <- A_j' mu
                const int k = p / mk, r = p - k * mk;
                if (tid < n) {
                    const int jb = tid / nu;
                    tmp[tid] = (gD && jb == k) ? opD[k * sDk + r * nu + (tid - jb * nu)] : T(0);
                }
                if (gC) __syncthreads();  // the recursion below overwrites blocks j < k of tmp
                if (gC && tid < 64) {
                    T *mu = fe + 4 * nx + 3 * nx * nx;  // two buffers of nx
                    if (tid < nx) mu[tid] = opC[k * sCk + r * nx + tid];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (int j = k - 1; j >= 0; --j) {
                        const T *mc = mu + ((k - 1 - j) & 1) * nx;
                        T *mn = mu + ((k - j) & 1) * nx;
                        const T *Aj = opA + j * sAk, *Bj = opB + j * sBk;
                        if (tid < nu) {
                            T acc = T(0);
                            for (int a = 0; a < nx; ++a) acc += Bj[a * nu + tid] * mc[a];
                            tmp[j * nu + tid] = acc;
                        } else if (tid >= 32 && tid < 32 + nx) {
                            const int b = tid - 32;
                            T acc = T(0);
                            for (int a = 0; a < nx; ++a) acc += Aj[a * nx + b] * mc[a];
                            mn[b] = acc;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                }
            } else if constexpr (STRUCT) {
                if (tid < n) {
                    const int k = p / mk, r = p - k * mk;
                    T g = T(0);
                    if (gC) {
                        const T *cr = gC + (int64_t)k * ka.C.step_stride + r * nx;
                        const T *pk = Psi + (int64_t)k * nx * n + tid;
#pragma unroll
                        for (int sc = 0; sc < 16; ++sc) {
                            const int sl = min(sc, nx - 1);
                            const T cv = cr[sl];
                            g += ((sc < nx) ? cv : T(0)) * pk[(int64_t)sl * n];
                        }
                    }
                    const int jb = tid / nu;
                    if (gD && jb == k) g += gD[(int64_t)k * ka.D.step_stride + r * nu + (tid - jb * nu)];
                    tmp[tid] = g;
                }
            } else {
                if (tid < n) tmp[tid] = G[(int64_t)p * n + tid];
            }
            __syncthreads();
            lap(1);
            {
                const T a = lower_matvec(tmp);
                if (tid < n) mp[tid] = a;
            }
            __syncthreads();
            T kpp;
            {
                const T v = (tid < n) ? mp[tid] * mp[tid] : T(0);
                kpp = block_sum<NWV>(v, red, tid);
            }
            T up = T(0);
            bool added = false;
            lap(2);
            while (!added) {
                if (iters >= max_iter) {
                    fail = true;
                    break;
                }
                ++iters;
                // r_a = T_a . M_p : one wavefront per row, lanes across columns
                __syncthreads();
                for (int a = tid >> 6; a < nq; a += NWV) {
                    T acc = T(0);
                    for (int k = tid & 63; k < n; k += 64) acc += Tm[(int64_t)a * n + k] * mp[k];
                    acc = wave_sum(acc);
                    if ((tid & 63) == 0) rv[a] = acc;
                }
                __syncthreads();
                lap(3);
                // z = -M_p + sum_a r_a M_a   (thread k: column k, coalesced)
                T zk = T(0);
                if (tid < n) {
                    zk = -mp[tid];
#pragma unroll 4
                    for (int a = 0; a < nq; ++a) zk += rv[a] * MA[(int64_t)a * n + tid];
                    zv[tid] = zk;
                }
                const T d2 = block_sum<NWV>(zk * zk, red, tid);
                // ratio test
                T t1 = INF;
                int l = 0x7fffffff;
                if (tid < nq && rv[tid] > T(0)) {
                    t1 = lam[tid] / rv[tid];
                    l = tid;
                }
                block_argmin<NWV>(t1, l, red, redi, tid);
                const bool can_move = (nq < n) && (d2 > (T)1e-10 * kpp) && (d2 > T(0));
                const T sp = sv[p];
                const T t2 = can_move ? -sp / d2 : INF;
                const T t = t1 < t2 ? t1 : t2;
                if (!(t < INF)) {
                    status = MPCQP_INFEASIBLE;
                    fail = true;
                    break;
                }
                const bool full = (t2 <= t1);
                // z_x = L^-T z, then s_i -= t G_i . z_x through the transposed G
                __syncthreads();
                lap(4);
                {
                    const T a = upper_matvec(zv);
                    if (tid < n) zx[tid] = a;
                }
                __syncthreads();
                lap(5);
                if constexpr (MID) {
                    rollout();
                    for (int i = tid; i < m; i += BS) sv[i] = (pos[i] >= 0) ? T(0) : sv[i] - t * mid_row(i);
                } else if constexpr (STRUCT) {
                    rollout();
#pragma unroll
                    for (int j = 0; j < RS; ++j) {
                        const int i = tid + BS * j;
                        if (i < m) sv[i] = (pos[i] >= 0) ? T(0) : sv[i] - t * struct_row(j);
                    }
                } else {
                    if ((m & 3) == 0) {
                        for (int i4 = tid; i4 < (m >> 2); i4 += BS) {
                            V4 a = {T(0), T(0), T(0), T(0)};
    #pragma unroll 8
                            for (int k = 0; k < n; ++k)
                                a += *reinterpret_cast<const V4 *>(GT + (int64_t)k * m + 4 * i4) * zx[k];
    #pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int i = 4 * i4 + c;
                                sv[i] = (pos[i] >= 0) ? T(0) : sv[i] - t * a[c];
                            }
                        }
                    } else {
                        for (int i = tid; i < m; i += BS) {
                            T a0 = T(0), a1 = T(0);
                            for (int k = 0; k < n; k += 2) {
                                a0 += GT[(int64_t)k * m + i] * zx[k];
                                a1 += GT[(int64_t)(k + 1) * m + i] * zx[k + 1];
                            }
                            sv[i] = (pos[i] >= 0) ? T(0) : sv[i] - t * (a0 + a1);
                        }
                    }
                }
                if (tid < nq) {
                    T lv = lam[tid] - t * rv[tid];
                    lam[tid] = lv < T(0) ? T(0) : lv;
                }
                up += t;
                __syncthreads();
                lap(6);
                if (full) {
                    // T_a += (r_a / d2) z ; new row T_nq = -z / d2 ; M_nq = M_p
                    const T inv = T(1) / d2;
                    if (tid < n) {
                        const T zt = zv[tid];
#pragma unroll 4
                        for (int a = 0; a < nq; ++a) Tm[(int64_t)a * n + tid] += (rv[a] * inv) * zt;
                        Tm[(int64_t)nq * n + tid] = -zt * inv;
                        MA[(int64_t)nq * n + tid] = mp[tid];
                    }
                    if (tid == 0) {
                        lam[nq] = up;
                        act[nq] = p;
                        pos[p] = nq;
                        sv[p] = T(0);
                    }
                    ++nq;
                    added = true;
                } else {
                    // drop slot l: T_a -= (T_a . T_l / T_l . T_l) T_l, then compact (last slot moves to l)
                    __syncthreads();
                    if (tid < n) tmp[tid] = Tm[(int64_t)l * n + tid];
                    __syncthreads();
                    for (int a = tid >> 6; a < nq; a += NWV) {
                        T acc = T(0);
                        for (int k = tid & 63; k < n; k += 64) acc += Tm[(int64_t)a * n + k] * tmp[k];
                        acc = wave_sum(acc);
                        if ((tid & 63) == 0) rv[a] = acc;  // rv reused: T_a . T_l
                    }
                    __syncthreads();
                    const T tll = rv[l];
                    const int last = nq - 1;
                    if (tid < n) {
                        const T tl = tmp[tid];
                        for (int a = 0; a < nq; ++a)
                            if (a != l) Tm[(int64_t)a * n + tid] -= (rv[a] / tll) * tl;
                        if (l != last) {
                            Tm[(int64_t)l * n + tid] = Tm[(int64_t)last * n + tid];
                            MA[(int64_t)l * n + tid] = MA[(int64_t)last * n + tid];
                        }
                    }
                    __syncthreads();
                    if (tid == 0) {
                        pos[act[l]] = -1;
                        if (l != last) {
                            act[l] = act[last];
                            lam[l] = lam[last];
                            pos[act[l]] = l;
                        }
                        lam[last] = T(0);
                    }
                    --nq;
                }
                __syncthreads();
                lap(7);
            }
            if (fail) break;
        }
        mark(5);
        if (!fail || status == MPCQP_SOLVED) {
            // y = y0 - M_A' lam ; u = L^-T y
            primal_point();
            bool verified = false;  // (mid-size kind) the active rows were found on their bounds by the refinement's own check
            if constexpr (MID) {
                // Refinement (float64 mid-size kind): the operator T = N* is only ever updated, and after a few hundred
                // iterations of a degenerate problem the active rows can sit 1e-6 .. 1e-4 off their bounds. Up to two steps
                // dlam = -T (T' rho_A) with the rows' residuals rho_A bring them back (T T' = (M_A M_A')^-1) before the
                // acceptance test below decides; problems whose rows are on their bounds skip it.
                for (int pass = 0; pass < 2 && status == MPCQP_SOLVED && nq > 0; ++pass) {
                    rollout();
                    bool off = false;
                    for (int a = tid; a < nq; a += BS) {
                        const int i = act[a];
                        const T tv = tolv[i], hi = (T)copysign((double)(((T)fabs(tv) - tol) / tol), (double)tv);
                        const T rho = hi - mid_row(i);
                        rv[a] = rho;
                        off |= !((T)fabs(rho) <= T(64) * (T)fabs(tv));
                    }
                    if (!__syncthreads_or(off)) {
                        verified = true;
                        break;
                    }
                    if (tid < n) {  // t = T' rho
                        T acc = T(0);
                        for (int a = 0; a < nq; ++a) acc += Tm[(int64_t)a * n + tid] * rv[a];
                        tmp[tid] = acc;
                    }
                    __syncthreads();
                    for (int a = tid >> 6; a < nq; a += BS / 64) {  // dlam_a = -T_a . t (one wavefront per row)
                        T acc = T(0);
                        for (int k = tid & 63; k < n; k += 64) acc += Tm[(int64_t)a * n + k] * tmp[k];
                        acc = wave_sum(acc);
                        if ((tid & 63) == 0) {
                            const T v = lam[a] - acc;
                            lam[a] = v < T(0) ? T(0) : v;
                        }
                    }
                    primal_point();
                }
            }
            // Acceptance, from scratch: every ACTIVE row must sit on its bound (the loop only ever looks at inactive
            // rows, and sets the active ones' slacks to zero). On an inconsistent problem a row that depends on the
            // active ones can slip past the pivot test on rounding noise -- float32 (4, 1, 50, 2) of the fuzz test came
            // back 'solved' with |u| ~ 2e6 and rows violated by 6e6 -- and then the active rows are far off.
            if (status == MPCQP_SOLVED) {
                // |residual| <= (1 + |h_i|) max(1e3 tol, 1e-6): the contract's 1e-6 in float64 (no refinement step here: an
                // ill-conditioned but legitimate plan leaves ~1e-8), 1e-2 in float32
                const T kacc = T(1000) > T(1e-6) / tol ? T(1000) : T(1e-6) / tol;
                bool bad = false;
                if constexpr (MID) {
                    if (!verified) rollout();
                    for (int a = tid; a < nq; a += BS) {
                        const int i = act[a];
                        const T tv = tolv[i], hi = (T)copysign((double)(((T)fabs(tv) - tol) / tol), (double)tv);
                        bad |= (!verified && !((T)fabs(hi - mid_row(i)) <= kacc * (T)fabs(tv))) || !(lam[a] >= T(0));
                    }
                } else if constexpr (STRUCT) {
                    rollout();
#pragma unroll
                    for (int j = 0; j < RS; ++j) {
                        const int i = tid + BS * j;
                        if (i < m && pos[i] >= 0) bad |= !((T)fabs(h[i] - struct_row(j)) <= kacc * tolv[i]);
                    }
                } else {
                    for (int a = tid; a < nq; a += BS) {
                        const int i = act[a];
                        T dot = T(0);
                        for (int k = 0; k < n; ++k) dot += GT[(int64_t)k * m + i] * zx[k];
                        bad |= !((T)fabs(h[i] - dot) <= kacc * tolv[i]) || !(lam[a] >= T(0));
                    }
                }
                if (tid < n) bad |= !(zx[tid] - zx[tid] == T(0));  // (finite)
                if (__syncthreads_or(bad)) status = MPCQP_MAX_ITER;
            }
        }
        if (fail && status == MPCQP_SOLVED) status = MPCQP_MAX_ITER;
    }
    mark(6);
    if (stamp && tid == 0)
        for (int u = 0; u < 8; ++u) stamp[8 + u] = lap_acc[u];
    const bool ok = (status == MPCQP_SOLVED);
    if (tid < n) oU[tid] = ok ? zx[tid] : T(0);
    if (ka.lam) {
        T *ol = (T *)ka.lam + prob * (int64_t)m;
        for (int i = tid; i < m; i += BS) ol[i] = (ok && pos[i] >= 0) ? lam[pos[i]] : T(0);
    }
    if (tid == 0) {
        if (ka.status) ka.status[prob] = status;
        if (ka.iters) ka.iters[prob] = iters;
    }
}

// G [m][n] -> G' [n][m] per problem, 64 x 64 tiles through LDS (both sides coalesced).
template <typename T>
__global__ void __launch_bounds__(256) mpcqp_transpose_kernel(const T *__restrict__ Gall, T *__restrict__ GTall, int m, int n)
{
    __shared__ T tile[64][65];
    const int tiles_c = (n + 63) / 64;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const T *G = Gall + (int64_t)blockIdx.y * m * n;
    T *GT = GTall + (int64_t)blockIdx.y * m * n;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int r = ly; r < 64; r += 4) {
        const int row = tr * 64 + r, c = tc * 64 + lx;
        tile[r][lx] = (row < m && c < n) ? G[(int64_t)row * n + c] : T(0);
    }
    __syncthreads();
    for (int r = ly; r < 64; r += 4) {
        const int c = tc * 64 + r, row = tr * 64 + lx;
        if (row < m && c < n) GT[(int64_t)c * m + row] = tile[lx][r];
    }
}

// ------------------------------------------------------------------ host side
int launch_transpose(const void *G, void *GT, int m, int n, int dtype, int64_t batch, hipStream_t st)
{
    const dim3 grid((unsigned)(((m + 63) / 64) * ((n + 63) / 64)), (unsigned)batch);
    if (dtype == MPCQP_F64)
        hipLaunchKernelGGL(mpcqp_transpose_kernel<double>, grid, dim3(256), 0, st, (const double *)G, (double *)GT, m, n);
    else
        hipLaunchKernelGGL(mpcqp_transpose_kernel<float>, grid, dim3(256), 0, st, (const float *)G, (float *)GT, m, n);
    return (int)hipGetLastError();
}

size_t bigsolve_lds_bytes(int n, int m, size_t esz, int rollout_elems)
{
    const size_t el = (size_t)n * (n + 1) / 2 + 3 * (size_t)m + 7 * (size_t)n + 8 + (size_t)rollout_elems;
    return el * esz + ((size_t)n + m + 4) * 4 + 16;
}
bool bigsolve_supported(int n, int m, int dtype)
{
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    return n >= 64 && n <= bigs::BS && (n % 2 == 0) && bigsolve_lds_bytes(n, m, esz, 0) <= kLdsBytesPerCU;
}
// structured (matrix-free G) mode: needs the roll-out table in LDS too, nx and nu within the lane map
bool bigsolve_struct_supported(const KernelArgs &ka, int dtype)
{
    // float32 only: with the problem's A, B, C, D resident in registers the float64 instantiation needed ~860 VGPRs (512 +
    // 352 spilled, 484 B of scratch) and was 4 % faster than forming G, which is what float64 problems get now
    if (dtype == MPCQP_F64) return false;
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    return ka.m > 0 && ka.m <= 4 * bigs::BS && ka.mk > 0 && ka.nx <= 16 && ka.nu <= 8 && ka.N <= 65 &&
           ka.n >= 64 && ka.n <= bigs::BS && (ka.n % 2 == 0) &&
           bigsolve_lds_bytes(ka.n, ka.m, esz, ka.N * ka.nx) <= kLdsBytesPerCU;
}
size_t bigsolve_ws_elems(int n) { return (size_t)ws_stride_elems(n); }

// extra LDS elements of the mid-size kind: roll-out table, operands, two Psi_k buffers, front-end vectors
static size_t mid_extra_elems(const KernelArgs &ka)
{
    size_t el = (size_t)ka.N * ka.nx;
    el += (size_t)(ka.A.step_stride ? ka.N : 1) * ka.nx * ka.nx + (size_t)(ka.B.step_stride ? ka.N : 1) * ka.nx * ka.nu;
    if (ka.C.ptr) el += (size_t)(ka.C.step_stride ? ka.N : 1) * ka.mk * ka.nx;
    if (ka.D.ptr) el += (size_t)(ka.D.step_stride ? ka.N : 1) * ka.mk * ka.nu;
    el += 2 * (size_t)ka.nx * ka.n + 6 * (size_t)ka.nx + 3 * (size_t)ka.nx * ka.nx;
    const size_t np = ((size_t)ka.n + 15) & ~(size_t)15;
    el += np * (np + 1) / 2 - (size_t)ka.n * (ka.n + 1) / 2;  // triangle padded to 16-row blocks
    return el;
}
// Fused build+solve of mid-size problems in one launch; LDS capped so that at least two problems share a CU.
bool mid_supported(const KernelArgs &ka, int dtype)
{
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    return ka.m > 0 && ka.mk > 0 && ka.nx <= 16 && ka.nu <= 32 && ka.n >= 32 && ka.n <= 160 &&
           bigsolve_lds_bytes(ka.n, ka.m, esz, (int)mid_extra_elems(ka)) <= 72 * 1024;
}

template <typename T, int KIND, int NWV = 4>
static int launch_bigsolve_t(const KernelArgs &ka, int64_t batch, const void *P, const void *q, const void *G,
                             const void *aux, const void *h, const void *aux2, void *ws, hipStream_t st)
{
    const size_t lds = bigsolve_lds_bytes(ka.n, ka.m, sizeof(T),
                                          KIND == K_MID ? (int)mid_extra_elems(ka) : (KIND == K_STRUCT ? ka.N * ka.nx : 0));
    auto kern = mpcqp_bigsolve_kernel<T, KIND, NWV>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(64 * NWV), lds, st, ka, (const T *)P, (const T *)q,
                       (const T *)G, (const T *)aux, (const T *)h, (const T *)aux2, (T *)ws);
    return (int)hipGetLastError();
}

int launch_bigsolve(const KernelArgs &ka, int dtype, int64_t batch, const void *P, const void *q, const void *G,
                    const void *GT, const void *h, void *ws, hipStream_t st)
{
    if (dtype == MPCQP_F64) return launch_bigsolve_t<double, K_DENSE>(ka, batch, P, q, G, GT, h, nullptr, ws, st);
    return launch_bigsolve_t<float, K_DENSE>(ka, batch, P, q, G, GT, h, nullptr, ws, st);
}

int launch_bigsolve_struct(const KernelArgs &ka, int dtype, int64_t batch, const void *P, const void *q,
                           const void *Psi_all, const void *h, const void *rownorm_inv, void *ws, hipStream_t st)
{
    if (dtype == MPCQP_F64) return MPCQP_EUNSUPPORTED;  // (bigsolve_struct_supported: float32 only)
    return launch_bigsolve_t<float, K_STRUCT>(ka, batch, P, q, nullptr, Psi_all, h, rownorm_inv, ws, st);
}

int launch_mid(const KernelArgs &ka, int dtype, int64_t batch, void *ws, hipStream_t st)
{
    // two wavefronts per problem pay only when the smaller block really doubles the problems resident on a
    // CU, i.e. when LDS (not the 16 wavefront slots) allows eight of them; config 3 (30 KB) stays at four
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    const bool two = ka.n <= 128 && bigsolve_lds_bytes(ka.n, ka.m, esz, (int)mid_extra_elems(ka)) <= 20 * 1024;
    if (dtype == MPCQP_F64)
        return two ? launch_bigsolve_t<double, K_MID, 2>(ka, batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, st)
                   : launch_bigsolve_t<double, K_MID, 4>(ka, batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, st);
    return two ? launch_bigsolve_t<float, K_MID, 2>(ka, batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, st)
               : launch_bigsolve_t<float, K_MID, 4>(ka, batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, st);
}

}  // namespace mpcqp
