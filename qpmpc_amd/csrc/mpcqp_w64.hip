// mpcqp_w64.hip -- gfx950 kernel for SMALL problems (n <= 16 variables,
// m <= 63 inequality rows, nx <= 4): ONE PROBLEM PER WAVEFRONT.
//
// Replaces the same reference code as mpcqp_lds.hip (qpmpc/mpc_qp.py:53-149 for
// the build, qpsolvers.solve_problem at qpmpc/solve_mpc.py:43 for the solve),
// for the sizes of BASELINE configs 1, 2 and 4 (nx=3, nu=1, N=16 -> n=16, m=32).
//
// Design rule (measured): with four wavefronts per SIMD the
// kernel is bound by INSTRUCTION ISSUE, so everything is arranged to need few,
// wide instructions: 16-element vectors are exchanged through LDS and read back
// as broadcast ds_read_b128; row data that is only ever indexed statically lives
// in VGPRs; nothing is indexed dynamically in registers; no per-element uniform
// branches; the active set lives in 16 fixed SLOTS (no compaction on a drop).
//
// Lane roles (a 64-wide wavefront is the whole machine of one problem):
//   lanes  0..15  SLOT a of the active set: multiplier lam_a, constraint act_a and
//                 row a of T = N* in 16 registers. Earlier: row a of P -> L during
//                 the factorisation; lane 0 carries q through the forward substitution.
//   lanes 16..47  CONSTRAINT i = lane-16 (m <= 32): row M_i of M = G L^-T in the
//                 same 16 registers, slack s_i.
//   lanes 48..63  row k = lane-48 of L^-T (identity rows pushed through the forward
//                 substitution), so that u = L^-T y is one dot product at the end.
// Every lane owns ONE 16-register row R; "r = T M_p", "M_i . z" and the rank-1
// update of T are the same instruction stream over R for all three roles.
// LDS per problem (9.6 KB, 16 wavefronts per CU): the image of M (row-p broadcast),
// the active rows M_A by slot, and a few 16-vectors used as broadcast buffers. The
// counters say the kernel is bound by the CU-shared LDS pipe, so T lives in
// registers and nothing is read-modify-written in LDS inside the loop.
//
// Solver = dual active set (Goldfarb-Idnani 1983) with their operator
// N* = (M_A M_A')^-1 M_A kept EXPLICITLY as T (16 slot rows of 16): no factor, no
// Gram column, no gather. The primal iterate is implied by the multipliers
// (y = y0 - M_A' lam). Per step, for the selected row p:
//   r = T M_p ;  z = -M_p + M_A' r ;  d2 = |z|^2
//   step t = min(t1 = min lam_a / r_a, t2 = -s_p / d2) ;  s_i -= t M_i . z
//   add  : T_a += (r_a/d2) z  (a active),  T_new = -z/d2  -> one rank-1 update by z
//   drop : T_a -= (T_a.T_l / T_l.T_l) T_l,  slot l cleared   (W = T T' is implicit)
// d2 comes from z itself (no cancellation); the final multipliers get one step of
// iterative refinement and the slacks are re-evaluated from scratch before the
// solution is accepted.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpcqp.h"
#include "mpcqp_internal.h"

namespace mpcqp {

namespace w64 {

constexpr int NV = 16;  // padded number of variables / slots

// ------------------------------------------------------------ lane primitives
__device__ __forceinline__ double bcast(double x, int lane)  // lane must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast(float x, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ __forceinline__ int bcast(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }

template <int CTRL> __device__ __forceinline__ int dpp(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ unsigned dpp(unsigned x) { return (unsigned)dpp<CTRL>((int)x); }
template <int CTRL> __device__ __forceinline__ float dpp(float x) { return __int_as_float(dpp<CTRL>(__float_as_int(x))); }
template <int CTRL> __device__ __forceinline__ double dpp(double x)
{
    return __hiloint2double(dpp<CTRL>(__double2hiint(x)), dpp<CTRL>(__double2loint(x)));
}
constexpr int ROR8 = 0x128, ROR4 = 0x124, ROR2 = 0x122, ROR1 = 0x121;  // rotate within a row of 16

template <typename T> __device__ __forceinline__ T row_sum(T v)  // all-reduce inside each row of 16
{
    v += dpp<ROR8>(v);
    v += dpp<ROR4>(v);
    v += dpp<ROR2>(v);
    v += dpp<ROR1>(v);
    return v;
}
__device__ __forceinline__ unsigned row_min(unsigned v)
{
    v = min(v, dpp<ROR8>(v));
    v = min(v, dpp<ROR4>(v));
    v = min(v, dpp<ROR2>(v));
    v = min(v, dpp<ROR1>(v));
    return v;
}
__device__ __forceinline__ unsigned wave_min(unsigned v)  // uniform result
{
    v = row_min(v);
    const unsigned a = (unsigned)bcast((int)v, 0), b = (unsigned)bcast((int)v, 16);
    const unsigned c = (unsigned)bcast((int)v, 32), d = (unsigned)bcast((int)v, 48);
    return min(min(a, b), min(c, d));
}

// order-preserving map of a floating-point value onto unsigned integers
__device__ __forceinline__ void ordered(double x, unsigned &hi, unsigned &lo)
{
    const unsigned h = (unsigned)__double2hiint(x), l = (unsigned)__double2loint(x);
    const bool neg = h & 0x80000000u;
    hi = neg ? ~h : (h | 0x80000000u);
    lo = neg ? ~l : l;
}
__device__ __forceinline__ void ordered(float x, unsigned &hi, unsigned &lo)
{
    const unsigned h = __float_as_uint(x);
    hi = (h & 0x80000000u) ? ~h : (h | 0x80000000u);
    lo = 0;
}
// arg-min over row 0 (lanes 0..15) with take==true; ties -> lowest lane; 64 if none.
// Exact except for the 6 low mantissa bits replaced by the lane id.
template <typename T> __device__ __forceinline__ int argmin_row0(T x, bool take, int lane)
{
    unsigned hi, lo;
    ordered(x, hi, lo);
    hi = take ? hi : 0xffffffffu;
    const unsigned mhi = (unsigned)bcast((int)row_min(hi), 0);
    if (mhi == 0xffffffffu) return 64;
    const unsigned low = (sizeof(T) == 8) ? ((lo & ~63u) | (unsigned)lane) : (unsigned)lane;
    const unsigned k2 = (take && hi == mhi) ? low : 0xffffffffu;
    return (int)((unsigned)bcast((int)row_min(k2), 0) & 63u);
}
// cheap selection over the wavefront (heuristic quality is enough): one reduction
template <typename T> __device__ __forceinline__ int argmin_coarse(T x, bool take, int lane)
{
    unsigned hi, lo;
    ordered(x, hi, lo);
    const unsigned key = take ? ((hi & ~63u) | (unsigned)lane) : 0xffffffffu;
    const unsigned mk = wave_min(key);
    return mk == 0xffffffffu ? 64 : (int)(mk & 63u);
}

// ------------------------------------------------------------ 16-vectors in LDS
template <typename T> struct Vec;
template <> struct Vec<double> {
    using type = double2;
    static constexpr int W = 2;
    static constexpr int LDW = 18;  // W row stride: 144 B, 16 lanes hit 16 distinct 16-B slots
};
template <> struct Vec<float> {
    using type = float4;
    static constexpr int W = 4;
    static constexpr int LDW = 20;  // 80 B rows, same property
};
template <typename T> __device__ __forceinline__ void ld16(T (&d)[NV], const T *src)
{
    using V = typename Vec<T>::type;
    const V *p = reinterpret_cast<const V *>(src);
#pragma unroll
    for (int i = 0; i < NV / Vec<T>::W; ++i) {
        const V t = p[i];
        if constexpr (Vec<T>::W == 2) {
            d[2 * i] = t.x;
            d[2 * i + 1] = t.y;
        } else {
            d[4 * i] = t.x;
            d[4 * i + 1] = t.y;
            d[4 * i + 2] = t.z;
            d[4 * i + 3] = t.w;
        }
    }
}
template <typename T> __device__ __forceinline__ void st16(T *dst, const T (&s)[NV])
{
    using V = typename Vec<T>::type;
    V *p = reinterpret_cast<V *>(dst);
#pragma unroll
    for (int i = 0; i < NV / Vec<T>::W; ++i) {
        V t;
        if constexpr (Vec<T>::W == 2) {
            t.x = s[2 * i];
            t.y = s[2 * i + 1];
        } else {
            t.x = s[4 * i];
            t.y = s[4 * i + 1];
            t.z = s[4 * i + 2];
            t.w = s[4 * i + 3];
        }
        p[i] = t;
    }
}
// Register pressure is what decides occupancy here (4 wavefronts per SIMD need
// <= 128 VGPRs), so 16-vectors coming from LDS are consumed in two halves of 8 and
// a scheduling barrier keeps the second half's loads from being hoisted.
constexpr int HV = 8;
template <typename T> __device__ __forceinline__ void ld8(T (&d)[HV], const T *src)
{
    using V = typename Vec<T>::type;
    const V *p = reinterpret_cast<const V *>(src);
#pragma unroll
    for (int i = 0; i < HV / Vec<T>::W; ++i) {
        const V t = p[i];
        if constexpr (Vec<T>::W == 2) {
            d[2 * i] = t.x;
            d[2 * i + 1] = t.y;
        } else {
            d[4 * i] = t.x;
            d[4 * i + 1] = t.y;
            d[4 * i + 2] = t.z;
            d[4 * i + 3] = t.w;
        }
    }
}
__device__ __forceinline__ void half_fence() { __builtin_amdgcn_sched_barrier(0); }
// Make a value opaque at this point: the compiler can neither sink the
// computation that produced it below here nor keep its operands alive instead
// (without this the trailing updates of the factorisation are deferred and every
// exchanged column stays live -> hundreds of bytes of scratch).
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }

// sum_k a[k] * v[k] with a in registers and v a 16-vector in LDS (broadcast reads)
template <typename T> __device__ __forceinline__ T dot_reg_lds(const T (&a)[NV], const T *v)
{
    T acc0 = T(0), acc1 = T(0);
#pragma unroll
    for (int h = 0; h < NV; h += HV) {
        T b[HV];
        ld8(b, v + h);
#pragma unroll
        for (int k = 0; k < HV; k += 2) {
            acc0 += a[h + k] * b[k];
            acc1 += a[h + k + 1] * b[k + 1];
        }
        half_fence();
    }
    return acc0 + acc1;
}
// sum_k a[k] * v[k] with both operands in LDS (a: per-lane row, v: broadcast vector)
template <typename T> __device__ __forceinline__ T dot_lds_lds(const T *a, const T *v)
{
    T acc0 = T(0), acc1 = T(0);
#pragma unroll
    for (int h = 0; h < NV; h += HV) {
        T x[HV], b[HV];
        ld8(x, a + h);
        ld8(b, v + h);
#pragma unroll
        for (int k = 0; k < HV; k += 2) {
            acc0 += x[k] * b[k];
            acc1 += x[k + 1] * b[k + 1];
        }
        half_fence();
    }
    return acc0 + acc1;
}
// row[k] += c * v[k] for a 16-row in LDS (read-modify-write) and a broadcast vector v;
// zero==true clears the row instead
template <typename T> __device__ __forceinline__ void axpy_row_lds(T *row, T c, const T *v, bool zero)
{
    using V = typename Vec<T>::type;
#pragma unroll
    for (int h = 0; h < NV; h += HV) {
        T x[HV], b[HV];
        ld8(x, row + h);
        ld8(b, v + h);
#pragma unroll
        for (int k = 0; k < HV; ++k) x[k] = zero ? T(0) : x[k] + c * b[k];
        V *p = reinterpret_cast<V *>(row + h);
#pragma unroll
        for (int i = 0; i < HV / Vec<T>::W; ++i) {
            V t;
            if constexpr (Vec<T>::W == 2) {
                t.x = x[2 * i];
                t.y = x[2 * i + 1];
            } else {
                t.x = x[4 * i];
                t.y = x[4 * i + 1];
                t.z = x[4 * i + 2];
                t.w = x[4 * i + 3];
            }
            p[i] = t;
        }
        half_fence();
    }
}
// acc += sum_a v[a] * col[a * STRIDE] : a 16-vector against a strided column (slot rows)
template <int STRIDE, typename T> __device__ __forceinline__ T dot_vec_col(const T *v, const T *col, T acc)
{
    T acc1 = T(0);
#pragma unroll
    for (int h = 0; h < NV; h += HV) {
        T b[HV];
        ld8(b, v + h);
#pragma unroll
        for (int a = 0; a < HV; a += 2) {
            acc += b[a] * col[(h + a) * STRIDE];
            acc1 += b[a + 1] * col[(h + a + 1) * STRIDE];
        }
        half_fence();
    }
    return acc + acc1;
}
// 1/x from the hardware estimate plus Newton steps (a full IEEE division costs ~3x
// the instructions; the operands here are never subnormal or zero when the result is used)
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ float fast_rcp(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    const float e = fmaf(-x, y, 1.0f);
    return fmaf(y, e, y);
}
// wavefront-level ordering of LDS traffic (a 64-thread workgroup needs no s_barrier)
__device__ __forceinline__ void wsync()
{
    // One wavefront per workgroup: LDS operations of a wavefront complete in order, so what one lane
    // wrote is what another lane reads next without any wait. Only the COMPILER has to keep the order
    // (wavefront-scope fence); __syncthreads() would also drain the LDS queue (s_waitcnt lgkmcnt(0)) at
    // every exchange, six times per active-set iteration.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T> struct Cst;
template <> struct Cst<double> {
    static __device__ __forceinline__ double inf() { return HUGE_VAL; }
    static __device__ __forceinline__ double dep() { return 1e-14; }  // |z|^2/|M_p|^2 below: dependent
    static __device__ __forceinline__ double rs(double x) { return rsqrt(x); }
};
template <> struct Cst<float> {
    static __device__ __forceinline__ float inf() { return HUGE_VALF; }
    static __device__ __forceinline__ float dep() { return 1e-6f; }
    static __device__ __forceinline__ float rs(float x) { return rsqrtf(x); }
};

// Stores that only lanes 0..15 should perform are made UNCONDITIONAL: lanes >= 16
// are pointed at shadow ("junk") copies -- a 17th row of M_A and a second set of
// the exchange vectors -- so the hot loops carry no exec-mask branches.
constexpr int CB = 16;    // first constraint lane
constexpr int LB = 48;    // first L^-T lane
constexpr int MMAX = 32;  // constraints this kernel can hold
constexpr int LDM = 18;   // row stride of the L and M images: 144 B, lanes writing rows hit distinct 16-B slots
struct Lay {    // LDS carve in elements of T (host-computed, passed by value)
    int off_X;  // build: G image (m+1) x 16 | main: M_A 17 x 16, then the T image 16 x 16 (refinement)
    int off_Y;  // build: exchange + staged operands | L 16 x 16 | main: M image m x 16
    int off_hv; // h_i by lane (64), inside Y behind everything else
    int off_v;  // kAv, rv, zv, their shadows, y0v, invv
    int off_stage, nA, nB, nC, nD;
    int total;
    // MODE_MODEL: element offsets inside the shared model (ModelLayout)
    int mo_M, mo_LinvT, mo_invn, mo_e, mo_Hx, mo_Wx, mo_Wg, mo_Wt, mo_flag;
};

}  // namespace w64

using namespace w64;

// MODE_FUSED: build from the MPC problem (A..targets). MODE_SOLVE: gA=P, gB=q, gC=G, ge=h.
// MK > 0: compile-time number of inequality rows per step for the software-pipelined
// chain (terminal cost only, state constraints only); MK == 0: generic chain.
template <typename T, int NX, int MODE, int MK>
__global__ void __launch_bounds__(64, 4)
    mpcqp_w64_kernel(const T *__restrict__ gA, const T *__restrict__ gB, const T *__restrict__ gC,
                     const T *__restrict__ gD, const T *__restrict__ ge, const T *__restrict__ gx0,
                     const T *__restrict__ ggoal, const T *__restrict__ gtgt, T *__restrict__ oU,
                     T *__restrict__ olam, int32_t *__restrict__ ostatus, int32_t *__restrict__ oiters,
                     const KernelArgs ka, const Lay L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sm = (T *)smem_raw;
    const int lane = threadIdx.x;
    const int l15 = lane & 15;
    const bool low = lane < NV;
    const int vofs = low ? lane : 3 * NV + l15;  // element of an exchange vector (shadow for lanes >= 16)
    const int64_t prob = blockIdx.x;
    const int n = ka.n, m = ka.m;
    const int cid = lane - CB;                   // constraint id of this lane
    const bool isc = (lane >= CB) && (cid < m);  // this lane owns a constraint
    const T INF = Cst<T>::inf();
    T *Gimg = sm + L.off_X, *MAl = sm + L.off_X, *Timg = sm + L.off_X + (NV + 1) * NV;
    const int GS = (m + 1) | 1;  // G image is stored by COLUMN with an odd stride: Gimg[c * GS + row],
                                 // conflict-free both for the column-wise writes and the row-wise fetch
    T *Ll = sm + L.off_Y, *Ml = sm + L.off_Y, *hv = sm + L.off_hv;
    T *kAv = sm + L.off_v, *rv = kAv + NV, *zv = rv + NV;
    // hv[lane] is only meaningful for the constraint lanes 16..47: its last 16 entries
    // double as y0 (written after hv is filled)
    T *y0v = hv + LB;

    T R[NV];   // this lane's row: T_a (slots) | M_i (constraints) | row of L^-T
    T Pr[NV];  // lane a < 16: row a of P, then of L
    // optional phase timestamps (tools/probe_phases.py): MpcqpSolveOpts.probe -> long long[8] per problem
    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 8 : nullptr;
    auto tick = [&](int slot) {
        if (stamp && lane == 0) stamp[slot] = (long long)__builtin_readcyclecounter();
    };
    tick(0);

    T invn_model = T(1);
    bool notpd = false;
    if constexpr (MODE == MODE_MODEL) {
        // Shared model (gA): M, L^-T and the maps from the states are already factored;
        // this problem only differs by x0 / goal / targets.
        const T *model = gA;
        const int nx = ka.nx, nT = ka.N * ka.nx;
        const T *x0 = gx0 + prob * ka.x0.batch_stride;
        const T *goal = ggoal ? ggoal + prob * ka.goal.batch_stride : nullptr;
        const T *tgt = gtgt ? gtgt + prob * ka.targets.batch_stride : nullptr;
        notpd = model[L.mo_flag] != T(0);
        const T *rowsrc = isc ? model + L.mo_M + cid * NV : model + L.mo_LinvT + (lane >= LB ? lane - LB : 0) * NV;
        const bool has_row = isc || lane >= LB;
        if (has_row) {
            ld16(R, rowsrc);
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) R[k] = T(0);
        }
        T hh = INF;
        if (isc) {
            hh = model[L.mo_e + cid];
            for (int c = 0; c < nx; ++c) hh -= model[L.mo_Hx + cid * nx + c] * x0[c];
            invn_model = model[L.mo_invn + cid];
        }
        hv[lane] = hh;
        wsync();
        if (low) {  // w = L^-1 q = Wx x0 - Wg goal - Wt targets (lane k: component k)
            T wk = T(0);
            for (int c = 0; c < nx; ++c) wk += model[L.mo_Wx + lane * nx + c] * x0[c];
            if ((ka.flags & MPCQP_Q_TERMINAL) && goal)
                for (int c = 0; c < nx; ++c) wk -= model[L.mo_Wg + lane * nx + c] * goal[c];
            if ((ka.flags & MPCQP_Q_STAGE) && tgt)
                for (int j2 = 0; j2 < nT; ++j2) wk -= model[L.mo_Wt + lane * nT + j2] * tgt[j2];
            y0v[lane] = wk;
        }
        for (int i = lane; i < (NV + 1) * NV; i += 64) MAl[i] = T(0);
        wsync();
    } else if constexpr (MODE == MODE_SOLVE) {
        const T *P = gA + prob * (int64_t)n * n;
        const T *G = gC + prob * (int64_t)m * n;
        const T *q = gB + prob * (int64_t)n;
#pragma unroll
        for (int b = 0; b < NV; ++b)
            Pr[b] = (lane < n && b < n) ? P[lane * n + b] : ((lane == b) ? T(1) : T(0));
        hv[lane] = isc ? ge[prob * (int64_t)m + cid] : INF;
        // rows of G (row m: q) are parked in the LDS image until P is factorised,
        // so that P's rows and G's rows are never live in registers together
        for (int i = lane; i < (m + 1) * NV; i += 64) {
            const int row = i / NV, b = i - row * NV;
            Gimg[b * GS + row] = (b < n) ? (row < m ? G[row * n + b] : q[b]) : T(0);
        }
        wsync();
    } else {
        // ---------------------------------------------------------------- build
        // nx == NX here (the host dispatches on it), so the small loops are exact.
        constexpr int nx = NX;
        const int nu = ka.nu, N = ka.N, mk = ka.mk;
        const T *A = gA + prob * ka.A.batch_stride;
        const T *B = gB + prob * ka.B.batch_stride;
        const T *Cm = gC ? gC + prob * ka.C.batch_stride : nullptr;
        const T *Dm = gD ? gD + prob * ka.D.batch_stride : nullptr;
        const T *x0 = gx0 + prob * ka.x0.batch_stride;
        const T *goal = ggoal ? ggoal + prob * ka.goal.batch_stride : nullptr;
        const T *tgt = gtgt ? gtgt + prob * ka.targets.batch_stride : nullptr;
        const int sA = ka.A.step_stride ? nx * nx : 0, sB = ka.B.step_stride ? nx * nu : 0;
        const int sC = ka.C.step_stride ? mk * nx : 0, sD = ka.D.step_stride ? mk * nu : 0;
        const bool stageP = ka.flags & MPCQP_P_STAGE, stageQ = (ka.flags & MPCQP_Q_STAGE) && tgt;
        const bool termP = ka.flags & MPCQP_P_TERMINAL, termQ = (ka.flags & MPCQP_Q_TERMINAL) && goal;
        T *ex = sm + L.off_Y;           // exchange: ex[s*32 + c] = Psi_k[s][c] (c < 16), ex[s*32 + 16] = residual
        T *hp = sm + L.off_Y + 4 * 32;  // hp[row] = C_k Phi_k x0 (m <= 32 entries)
        // One coalesced pass stages the problem's operands in LDS: a single HBM
        // latency instead of one per horizon step (a problem's steps are packed).
        // (The pipelined chain, MK > 0, reads A_k, C_k and its B column straight from HBM
        // one step ahead instead: fewer LDS operations, which is what this kernel is short of.)
        T *As = sm + L.off_stage, *Bs = As + L.nA, *Cs = Bs + L.nB, *Ds = Cs + L.nC;
        if constexpr (MK == 0) {
            for (int i = lane; i < L.nA; i += 64) As[i] = A[i];
            for (int i = lane; i < L.nB; i += 64) Bs[i] = B[i];
            for (int i = lane; i < L.nC; i += 64) Cs[i] = Cm[i];
            for (int i = lane; i < L.nD; i += 64) Ds[i] = Dm[i];
        }
        const bool isx = (lane == NV), col = (lane < n);
        const int j = col ? lane / nu : -1, ii = col ? lane - j * nu : 0;
        const T eval = isc ? ge[prob * ka.e.batch_stride + (cid / mk) * ka.e.step_stride + (cid % mk)] : INF;
        T v[NX], gref[NX];
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            v[s] = isx ? x0[s] : T(0);
            gref[s] = (isx && termQ) ? goal[s] : T(0);
        }
        const T wu = (T)ka.wu;
#pragma unroll
        for (int b = 0; b < NV; ++b) Pr[b] = (lane == b) ? (col ? wu : T(1)) : T(0);
        T qa = T(0);
        wsync();
        T bcol[NX];  // this lane's column of B_j (enters the chain at step j)
#pragma unroll
        for (int r = 0; r < NX; ++r) bcol[r] = col ? ((MK > 0) ? B[j * sB + r * nu + ii] : Bs[j * sB + r * nu + ii]) : T(0);

        // Gram accumulation of one block: Pr[b] += w v_a . v_b, qa += w resid . v_a;
        // ref[] is this lane's reference (non-zero only in lane 16).
        auto gram = [&](T w, bool useP, bool useQ, const T (&ref)[NX]) {
            if (!useP && !useQ) return;
            wsync();
            if (lane < 32) {
#pragma unroll
                for (int s = 0; s < NX; ++s) ex[s * 32 + lane] = v[s] - ref[s];
            }
            wsync();
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                const T t = w * v[s];
                if (useP) {
#pragma unroll
                    for (int h = 0; h < NV; h += HV) {
                        T vb[HV];
                        ld8(vb, ex + s * 32 + h);
#pragma unroll
                        for (int b = 0; b < HV; ++b) {
                            Pr[h + b] += t * vb[b];
                            pin(Pr[h + b]);
                        }
                        half_fence();
                    }
                }
                if (useQ) qa += t * ex[s * 32 + NV];
            }
        };

        // G rows of step k from v = Psi_k[:, lane] (lane 16: Phi_k x0); lanes 0..15 fill
        // column `lane` of the G image, lane 16 the C_k Phi_k x0 part of h (mpc_qp.py:62-78)
        T *gd = low ? (Gimg + lane * GS) : hp;
        constexpr int gs = 1;
        auto g_rows = [&](int k) {
            const bool here = (j == k);
            for (int i2 = 0; i2 < mk; ++i2) {
                T acc = T(0);
                if (L.nC) {
                    const T *Ci = Cs + k * sC + i2 * nx;  // broadcast LDS reads
#pragma unroll
                    for (int s = 0; s < NX; ++s) acc += Ci[s] * v[s];
                }
                if (L.nD) {
                    const T dv = Ds[k * sD + i2 * nu + ii];
                    acc += here ? dv : T(0);
                }
                gd[(k * mk + i2) * gs] = acc;
            }
        };
        // Psi_{k+1} = A_k Psi_k, then column block k <- B_k (mpc_qp.py:88-90)
        auto advance = [&](int k) {
            const T *Ak = As + k * sA;
            const bool here = (j == k);
            T w[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                T acc = T(0);
#pragma unroll
                for (int s = 0; s < NX; ++s) acc += Ak[r * nx + s] * v[s];
                w[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) v[r] = here ? bcol[r] : w[r];
        };
        if constexpr (MK > 0) {
            // Terminal cost only, C only, mk == MK (configs 1, 2, 4; the host checks).
            // Software-pipelined chain on 17 lanes: the broadcast reads of A_{k+1}, C_{k+1}
            // are in flight while step k computes [G_k; Psi_{k+1}] = [C_k; A_k] Psi_k.
            if (lane <= NV) {
                T a0[NX * NX], c0[MK * NX];
#pragma unroll
                for (int e = 0; e < NX * NX; ++e) a0[e] = A[e];
#pragma unroll
                for (int e = 0; e < MK * NX; ++e) c0[e] = Cm[e];
                for (int k = 0; k < N; ++k) {
                    const int kn = (k + 1 < N) ? k + 1 : k;
                    T a1[NX * NX], c1[MK * NX];
#pragma unroll
                    for (int e = 0; e < NX * NX; ++e) a1[e] = A[kn * sA + e];
#pragma unroll
                    for (int e = 0; e < MK * NX; ++e) c1[e] = Cm[kn * sC + e];
#pragma unroll
                    for (int i2 = 0; i2 < MK; ++i2) {
                        T acc = T(0);
#pragma unroll
                        for (int s2 = 0; s2 < NX; ++s2) acc += c0[i2 * NX + s2] * v[s2];
                        gd[(k * MK + i2) * gs] = acc;
                    }
                    const bool here = (j == k);
                    T w[NX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        T acc = T(0);
#pragma unroll
                        for (int s2 = 0; s2 < NX; ++s2) acc += a0[r * NX + s2] * v[s2];
                        w[r] = acc;
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) v[r] = here ? bcol[r] : w[r];
#pragma unroll
                    for (int e = 0; e < NX * NX; ++e) a0[e] = a1[e];
#pragma unroll
                    for (int e = 0; e < MK * NX; ++e) c0[e] = c1[e];
                }
            }
        } else if (!stageP && !stageQ) {
            // terminal cost only: a branch-light chain on 17 lanes
            if (lane <= NV) {
                for (int k = 0; k < N; ++k) {
                    g_rows(k);
                    advance(k);
                }
            }
        } else {
            for (int k = 0; k < N; ++k) {
                if (lane <= NV) g_rows(k);
                if (k >= 1) {
                    T tref[NX];
#pragma unroll
                    for (int s = 0; s < NX; ++s) tref[s] = (isx && stageQ) ? tgt[k * nx + s] : T(0);
                    gram((T)ka.wx, stageP, stageQ, tref);
                }
                advance(k);
            }
        }
        gram((T)ka.wt, termP, termQ, gref);  // v = Psi_N
        wsync();
        if (low) Gimg[lane * GS + m] = col ? qa : T(0);  // the q row
        wsync();
        // h_i = e_i - C_k Phi_k x0 goes to LDS; the rows of G stay in the LDS image for now
        hv[lane] = (isc && L.nC) ? eval - hp[cid] : eval;
        wsync();
    }

    tick(1);
    // ------------------------------------------------------------ factorise
    // Right-looking Cholesky on the rows held by lanes 0..15. Everything is broadcast
    // with v_readlane (the LDS pipe is this kernel's bottleneck, the VALU is not); the
    // trailing update needs no sqrt: P[i][k] -= P[i][j] P[k][j] / piv.
    T myinv = T(1);  // lane j keeps 1 / L_jj
    if constexpr (MODE != MODE_MODEL) {
    // Column j (one entry per lane) is broadcast through a double-buffered 16-entry LDS vector: column j+1
    // is brought up to date and written FIRST in step j, so its round trip overlaps the rest of step j's
    // updates. One write and <= 8 wavefront-uniform 16-byte reads per column replace 2 (15 - j) v_readlane.
    {
        using V = typename Vec<T>::type;
        constexpr int W = Vec<T>::W;
        T *cb = Ll + NV * LDM;  // two buffers of NV (+ a shadow entry for lanes >= 16), behind the L image
        const int cw = low ? lane : NV;
        cb[cw] = Pr[0];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const T *cbj = cb + (j & 1) * (NV + W);
            T cv[NV];
#pragma unroll
            for (int g = j / W; g < NV / W; ++g) {
                const V t = reinterpret_cast<const V *>(cbj)[g];
                if constexpr (W == 2) {
                    cv[2 * g] = t.x;
                    cv[2 * g + 1] = t.y;
                } else {
                    cv[4 * g] = t.x;
                    cv[4 * g + 1] = t.y;
                    cv[4 * g + 2] = t.z;
                    cv[4 * g + 3] = t.w;
                }
            }
            const T piv = cv[j];
            if (!(piv > T(0))) notpd = true;
            const T rinv = Cst<T>::rs(piv);
            const T pij = Pr[j];             // P[i][j] of this lane's row, before scaling
            const T t2 = pij * rinv * rinv;  // P[i][j] / piv
            if (j + 1 < NV) {
                Pr[j + 1] -= t2 * cv[j + 1];
                pin(Pr[j + 1]);
                (cb + ((j + 1) & 1) * (NV + W))[cw] = Pr[j + 1];
            }
#pragma unroll
            for (int k = j + 2; k < NV; ++k) {
                Pr[k] -= t2 * cv[k];  // cv[k] = P[k][j]
                pin(Pr[k]);
            }
            Pr[j] = pij * rinv;  // L[i][j] (lane j: sqrt(piv))
            if (lane == j) myinv = rinv;
        }
    }
    }
    tick(2);
    int status = MPCQP_MAX_ITER, iters = 0;
    T xsol = T(0), lam_out = T(0);
    if (notpd) {
        status = MPCQP_NOT_PD;
    } else {
        // Rows fetched only now (register pressure): lane 0 takes q, constraint lanes
        // their row of G, lanes 48.. the identity (-> rows of L^-T), all others zero.
        if constexpr (MODE != MODE_MODEL) {
            const bool fetch = isc || lane == 0;
            const int row = isc ? cid : m;
            if (fetch) {
#pragma unroll
                for (int k = 0; k < NV; ++k) R[k] = Gimg[k * GS + row];
            } else {
#pragma unroll
                for (int k = 0; k < NV; ++k) R[k] = (lane == LB + k) ? T(1) : T(0);
            }
        // R <- R L^-T, one row per lane. L[j][k] lives in lane j's Pr[k]; it is broadcast through an LDS
        // image (16-byte reads at a wavefront-uniform address: 80 LDS reads) instead of 272 v_readlane --
        // in this phase every wavefront of the CU is VALU-bound at the same time and the LDS pipe is idle.
        if (low) {
            st16(Ll + lane * LDM, Pr);
            Ll[lane * LDM + NV] = myinv;
        }
        wsync();
        {
            using V = typename Vec<T>::type;
            constexpr int W = Vec<T>::W;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const V *Lr = reinterpret_cast<const V *>(Ll + j * LDM);
                T acc = R[j];
#pragma unroll
                for (int kk = 0; kk * W < j; ++kk) {
                    const V t = Lr[kk];
                    if constexpr (W == 2) {
                        acc -= R[2 * kk] * t.x;
                        if (2 * kk + 1 < j) acc -= R[2 * kk + 1] * t.y;
                    } else {
                        acc -= R[4 * kk] * t.x;
                        if (4 * kk + 1 < j) acc -= R[4 * kk + 1] * t.y;
                        if (4 * kk + 2 < j) acc -= R[4 * kk + 2] * t.z;
                        if (4 * kk + 3 < j) acc -= R[4 * kk + 3] * t.w;
                    }
                }
                R[j] = acc * Lr[NV / W].x;
                pin(R[j]);
            }
        }
        wsync();  // the M image below reuses the L image
        tick(3);
        if (lane == 0) st16(y0v, R);           // w = L^-1 q ; y0 = -w
        if (isc) st16(Ml + cid * LDM, R);      // image of M for the row-p broadcasts
        for (int i = lane; i < (NV + 1) * NV; i += 64) MAl[i] = T(0);
        if (low) {
#pragma unroll
            for (int k = 0; k < NV; ++k) R[k] = T(0);  // T = N* starts empty
        }
        wsync();
        }
        // rows of M for the row-p broadcasts: the LDS image, or the shared model itself
        // (4 KB read by every wavefront of the launch: it stays in L1/L2)
        const T *Mbase = (MODE == MODE_MODEL) ? gA + L.mo_M : Ml;
        const int mstride = (MODE == MODE_MODEL) ? NV : LDM;
        const T hval = hv[lane];
        T s = hval + dot_reg_lds(R, y0v);  // h - M y0  (y0 = -L^-1 q)
        s = isc ? s : INF;
        // Selection rule (the classic Goldfarb-Idnani one): among the rows violated
        // beyond the tolerance, take the one FARTHEST from its hyperplane in the
        // P^-1 metric, s_i / |M_i|. On the triple-integrator family this needs
        // 10.8 iterations on average and 16 at most, against 12.3 / 26 when the
        // slack is only scaled by 1 + |h_i|, and it practically removes the drops.
        T invn = invn_model;
        if constexpr (MODE != MODE_MODEL) {
            T nn = T(0);
#pragma unroll
            for (int k = 0; k < NV; ++k) nn += R[k] * R[k];
            invn = (nn > T(0)) ? Cst<T>::rs(nn) : T(1);
        }
        const bool selectable = isc && (hval < T(1e29));
        const T tol = (T)ka.tol;
        const T tolh = tol + tol * fabs(hval);  // row i is violated when s_i < -tol (1 + |h_i|)
        const int max_iter = ka.max_iter;

        T lam = T(0);
        int myact = 0, pos = -1, nq = 0;
        bool occ = false;
        unsigned mask = 0;  // occupied slots (wave-uniform)
        bool fail = false;
        tick(4);
        for (int round = 0; round < 4 && !fail; ++round) {
            // ===================================================== active-set loop
            for (;;) {
                const int p = argmin_coarse<T>(s * invn, selectable && pos < 0 && s < -tolh, lane);
                if (p == 64) {
                    status = MPCQP_SOLVED;
                    break;
                }
                const T *mprow = Mbase + (p - CB) * mstride;  // row p of M, read as broadcast
                const T ip = bcast(invn, p);          // 1 / |M_p|
                T up = T(0);
                bool added = false;
                // Each trip of this loop makes exactly ONE pass "R += c * vec" over the row
                // registers: either the step along z (s and, on a full step, T are updated)
                // or, right after a partial step, the removal of slot ldrop from T. Keeping
                // a single modification site of R is what keeps R in registers.
                bool dropping = false;
                int ldrop = 0;
                while (!added) {
                    const T *vec;
                    T c, t = T(0), r = T(0);
                    bool full = false;
                    int sl = 0;
                    if (!dropping) {
                        if (iters >= max_iter) {
                            fail = true;
                            break;
                        }
                        ++iters;
                        // r = T M_p (slot lanes; the other roles compute and ignore)
                        r = dot_reg_lds(R, mprow);
                        r = occ ? r : T(0);
                        rv[vofs] = r;
                        wsync();
                        // z = -M_p + M_A' r (lane k < 16)
                        T z = dot_vec_col<NV>(rv, MAl + l15, -mprow[l15]);
                        z = low ? z : T(0);
                        zv[vofs] = z;
                        const T d2 = bcast(row_sum(z * z), 0);
                        // ratio test on the multipliers
                        const bool cand = occ && (r > T(0));
                        const T ratio = cand ? lam * fast_rcp(r) : INF;
                        const int l = argmin_row0<T>(ratio, cand, lane);
                        const T t1 = (l < 64) ? bcast(ratio, l) : INF;
                        const bool can_move = (nq < n) && (d2 * ip * ip > Cst<T>::dep()) && (d2 > T(0));
                        const T sp = bcast(s, p);
                        const T inv = can_move ? fast_rcp(d2) : T(0);
                        const T t2 = can_move ? -sp * inv : INF;
                        t = t1 < t2 ? t1 : t2;
                        if (!(t < INF)) {
                            status = MPCQP_INFEASIBLE;
                            fail = true;
                            break;
                        }
                        full = (t2 <= t1);
                        sl = __builtin_ctz(~mask);  // lowest free slot
                        ldrop = l;
                        // on a full step T gets its rank-1 update T_a += (r_a/d2) z, T_sl = -z/d2
                        c = full ? ((lane == sl) ? -inv : r * inv) : T(0);
                        vec = zv;
                    } else {
                        // slot ldrop leaves (its row T_l was copied to kAv). With W = T T' implicit,
                        // T_a -= (T_a . T_l / T_l . T_l) T_l ; row l becomes exactly zero (f = 1).
                        const T tl = dot_reg_lds(R, kAv);
                        const T f = tl * fast_rcp(bcast(tl, ldrop));
                        c = (lane == ldrop) ? T(-1) : (occ ? -f : T(0));
                        vec = kAv;
                    }
                    wsync();
                    // the single pass over vec: m_i = R_i . vec (before the update), R += c vec
                    T mz0 = T(0), mz1 = T(0);
#pragma unroll
                    for (int h = 0; h < NV; h += HV) {
                        T zz[HV];
                        ld8(zz, vec + h);
#pragma unroll
                        for (int k = 0; k < HV; k += 2) {
                            mz0 += R[h + k] * zz[k];
                            mz1 += R[h + k + 1] * zz[k + 1];
                        }
#pragma unroll
                        for (int k = 0; k < HV; ++k) {
                            R[h + k] += c * zz[k];
                            pin(R[h + k]);
                        }
                        half_fence();
                    }
                    if (!dropping) {
                        // the implied primal point moved by t z: s_i -= t M_i . z
                        if (isc) s = (pos >= 0) ? T(0) : s - t * (mz0 + mz1);
                        lam -= t * r;
                        lam = (occ && lam < T(0)) ? T(0) : lam;
                        up += t;
                        if (full) {
                            // p takes slot sl
                            MAl[(low ? sl : NV) * NV + l15] = mprow[l15];
                            if (lane == sl) {
                                lam = up;
                                myact = p;
                                occ = true;
                            }
                            if (lane == p) {
                                pos = sl;
                                s = T(0);
                            }
                            mask |= 1u << sl;
                            ++nq;
                            added = true;
                        } else {
                            // partial step: the next trip removes slot ldrop from T
                            const int cl = bcast(myact, ldrop);
                            wsync();
                            if (lane == ldrop) st16(kAv, R);
                            if (lane == cl) pos = -1;
                            dropping = true;
                        }
                    } else {
                        if (lane == ldrop) {
                            lam = T(0);
                            occ = false;
                        }
                        mask &= ~(1u << ldrop);
                        --nq;
                        dropping = false;
                    }
                    wsync();
                }
                if (fail) break;
            }
            if (fail) break;
            tick(5);
            // ================================== refine multipliers, verify slacks
            // y = y0 - M_A' lam (lane k < 16)
            wsync();
            rv[vofs] = lam;
            wsync();
            T y = dot_vec_col<NV>(rv, MAl + l15, T(0));
            y = -y0v[l15] - y;  // y0 = -L^-1 q
            zv[vofs] = y;
            wsync();
            T fresh = hv[lane] - dot_reg_lds(R, zv);
            fresh = isc ? fresh : INF;
            if (nq > 0) {
                // active residuals rho_a = h_a - M_a y should vanish:
                // dlam = -W rho_A = -T (T' rho_A)
                T rho = __shfl(fresh, myact);
                rho = occ ? rho : T(0);
                wsync();
                kAv[vofs] = rho;
                if (low) st16(Timg + lane * NV, R);  // T by columns is only needed here
                wsync();
                const T uk = dot_vec_col<NV>(kAv, Timg + l15, T(0));  // (T' rho)_k, lane k < 16
                rv[vofs] = low ? uk : T(0);
                wsync();
                T dl = -dot_reg_lds(R, rv);
                dl = occ ? dl : T(0);
                lam += dl;
                lam = (occ && lam < T(0)) ? T(0) : lam;
                wsync();
                rv[vofs] = dl;
                wsync();
                y -= dot_vec_col<NV>(rv, MAl + l15, T(0));
                wsync();
                zv[vofs] = y;
                wsync();
                fresh = hv[lane] - dot_reg_lds(R, zv);
                fresh = isc ? fresh : INF;
            }
            // accept when no inactive row is violated at the re-evaluated point
            const bool clean = __ballot(selectable && pos < 0 && fresh < -T(4) * tolh) == 0ull;
            if (clean || round == 3) {
                xsol = dot_reg_lds(R, zv);  // u = L^-T y in lanes 48..63 (zv holds y)
                status = clean ? MPCQP_SOLVED : MPCQP_MAX_ITER;
                break;
            }
            // otherwise continue the active-set loop from the re-evaluated slacks
            s = (pos >= 0) ? T(0) : fresh;
            status = MPCQP_MAX_ITER;
        }
        if (fail && status == MPCQP_SOLVED) status = MPCQP_MAX_ITER;
        if (status == MPCQP_SOLVED && olam) {
            const T lv = __shfl(lam, pos < 0 ? 0 : pos);
            lam_out = (pos >= 0) ? lv : T(0);
        }
    }

    tick(6);
    const bool ok = (status == MPCQP_SOLVED);
    {
        const int k = lane - LB;  // lanes 48.. hold u_k
        if (k >= 0 && k < n) oU[prob * (int64_t)n + k] = ok ? xsol : T(0);
    }
    if (olam && isc) olam[prob * (int64_t)m + cid] = ok ? lam_out : T(0);
    if (lane == 0) {
        if (ostatus) ostatus[prob] = status;
        if (oiters) oiters[prob] = iters;
    }
}

// ------------------------------------------------------------ host side
template <typename T> static Lay make_lay(const KernelArgs &ka)
{
    Lay L{};
    auto al = [](int c) { return (c + 3) & ~3; };  // 16-byte alignment for float and double
    const int gimg = NV * ((ka.m + 1) | 1), main_x = (NV + 1) * NV + NV * NV;
    L.off_X = 0;
    int o = al(gimg > main_x ? gimg : main_x);
    L.off_Y = o;
    int y_build = 4 * 32 + 64;  // ex (4 x 32), hp
    L.off_stage = L.off_Y + y_build;
    if (ka.A.ptr) {  // fused mode: the staged operands sit behind the exchange buffers
        L.nA = (ka.A.step_stride ? ka.N : 1) * ka.nx * ka.nx;
        L.nB = (ka.B.step_stride ? ka.N : 1) * ka.nx * ka.nu;
        L.nC = ka.C.ptr ? (ka.C.step_stride ? ka.N : 1) * ka.mk * ka.nx : 0;
        L.nD = ka.D.ptr ? (ka.D.step_stride ? ka.N : 1) * ka.mk * ka.nu : 0;
        y_build += al(L.nA) + al(L.nB) + al(L.nC) + al(L.nD);
    }
    int y_main = ka.m * LDM;  // the M image; the L image (16 x LDM) fits inside
    if (y_main < NV * LDM + 2 * (NV + 4)) y_main = NV * LDM + 2 * (NV + 4);  // L image + the two column buffers of the factorisation
    const int y_sz = al(y_build > y_main ? y_build : y_main);
    L.off_hv = L.off_Y + y_sz;
    o = L.off_hv + 64;
    L.off_v = o;
    o += 6 * NV;  // kAv rv zv | their shadows
    L.total = o;
    if (ka.model) {
        const ModelLayout ml = make_model_layout(ka.nx, ka.N, ka.n, ka.m);
        L.mo_M = (int)ml.off_M;
        L.mo_LinvT = (int)ml.off_LinvT;
        L.mo_invn = (int)ml.off_invn;
        L.mo_e = (int)ml.off_e;
        L.mo_Hx = (int)ml.off_Hx;
        L.mo_Wx = (int)ml.off_Wx;
        L.mo_Wg = (int)ml.off_Wg;
        L.mo_Wt = (int)ml.off_Wt;
        L.mo_flag = (int)ml.total;
    }
    return L;
}

// float64 only: in single precision the explicit operator T loses too much on
// ill-conditioned active sets (cond up to 2e8 in the humanoid sweep); float32
// problems go to the Q-based kernel of mpcqp_lds.hip instead. m <= 32 and n <= 16
// are what the lane map holds.
bool w64_eligible(const KernelArgs &ka, int mode, int dtype)
{
    if (dtype != MPCQP_F64) return false;
    if (ka.n > NV || ka.m > MMAX) return false;
    if (mode == MODE_FUSED && ka.nx != 3 && ka.nx != 4) return false;
    if (mode == MODE_MODEL && ka.n > NV) return false;
    return mode == MODE_FUSED || mode == MODE_SOLVE || mode == MODE_MODEL;
}

template <typename T, int MODE, int NX, int MK>
static int launch_w64_t(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    const Lay L = make_lay<T>(ka);
    const size_t bytes = (size_t)L.total * sizeof(T);
    if constexpr (MODE == MODE_MODEL) {
        hipLaunchKernelGGL((mpcqp_w64_kernel<T, NX, MODE, MK>), dim3((unsigned)batch), dim3(64), bytes, st,
                           (const T *)ka.model, (const T *)nullptr, (const T *)nullptr, (const T *)nullptr,
                           (const T *)nullptr, (const T *)ka.x0.ptr, (const T *)ka.goal.ptr,
                           (const T *)ka.targets.ptr, (T *)ka.U, (T *)ka.lam, ka.status, ka.iters, ka, L);
    } else if constexpr (MODE == MODE_FUSED) {
        hipLaunchKernelGGL((mpcqp_w64_kernel<T, NX, MODE, MK>), dim3((unsigned)batch), dim3(64), bytes, st,
                           (const T *)ka.A.ptr, (const T *)ka.B.ptr, (const T *)ka.C.ptr, (const T *)ka.D.ptr,
                           (const T *)ka.e.ptr, (const T *)ka.x0.ptr, (const T *)ka.goal.ptr,
                           (const T *)ka.targets.ptr, (T *)ka.U, (T *)ka.lam, ka.status, ka.iters, ka, L);
    } else {
        hipLaunchKernelGGL((mpcqp_w64_kernel<T, NX, MODE, MK>), dim3((unsigned)batch), dim3(64), bytes, st,
                           (const T *)ka.P, (const T *)ka.q, (const T *)ka.G, (const T *)nullptr,
                           (const T *)ka.h, (const T *)nullptr, (const T *)nullptr, (const T *)nullptr,
                           (T *)ka.U, (T *)ka.lam, ka.status, ka.iters, ka, L);
    }
    return (int)hipGetLastError();
}

int launch_w64(const KernelArgs &ka, int mode, int dtype, int64_t batch, hipStream_t st)
{
    (void)dtype;  // w64_eligible() admits MPCQP_F64 only
    if (mode == MODE_FUSED) {
        // the pipelined chain: terminal cost only, state constraints only, two rows per step
        const bool lean = ka.mk == 2 && ka.C.ptr && !ka.D.ptr && !(ka.flags & (MPCQP_P_STAGE | MPCQP_Q_STAGE));
        if (ka.nx == 3)
            return lean ? launch_w64_t<double, MODE_FUSED, 3, 2>(ka, batch, st)
                        : launch_w64_t<double, MODE_FUSED, 3, 0>(ka, batch, st);
        return lean ? launch_w64_t<double, MODE_FUSED, 4, 2>(ka, batch, st)
                    : launch_w64_t<double, MODE_FUSED, 4, 0>(ka, batch, st);
    }
    if (mode == MODE_MODEL) return launch_w64_t<double, MODE_MODEL, 4, 0>(ka, batch, st);
    return launch_w64_t<double, MODE_SOLVE, 4, 0>(ka, batch, st);
}

}  // namespace mpcqp
