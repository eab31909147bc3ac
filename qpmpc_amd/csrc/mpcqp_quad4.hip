// mpcqp_quad4.hip -- gfx950: the four-problems-per-wavefront kernel of mpcqp_quad.hip for problems with MORE THAN 32 ROWS: n <= 16
// variables, 33 <= m <= 64 inequality rows (three or four rows per step at N = 16: a state box next to an input box), nx <= 8, float64,
// cold launches. Replaces the same reference code (qpmpc/mpc_qp.py:53-149 for the build, qpsolvers.solve_problem at
// qpmpc/solve_mpc.py:43 for the solve); until the end of round 6 such problems fell to the workgroup / one-per-wavefront kernels,
// 9-12 x slower than the same problem with two rows per step.
//
// Same method, same layout as mpcqp_quad.hip (read that file first: the comments below are its comments) with FOUR constraint rows per
// lane instead of two -- rows l, l + 16, l + 32, l + 48 of M in registers, their slacks, norms and flags as arrays of four -- and the
// selection key carrying a six-bit row id. It is a copy and not a template parameter of that kernel on purpose: renaming the tuned
// kernel's per-row scalars to arrays alone cost the headline launch 2.4 % (profiles/HISTORY.md). One wavefront per SIMD at most
// (~330 registers of the 512 a lone wavefront may hold); 49.9 KB of LDS per wavefront (M image 64 x 16), three wavefronts per CU.
// Only the general build (GEN) in the one-round carve is instantiated.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "mpcqp.h"
#include "mpcqp_internal.h"

namespace mpcqp {

namespace quad4 {

constexpr int NV = 16;    // padded number of variables / slots = lanes per problem
constexpr int MMAX = 64;  // constraints a problem can hold (four per lane)
constexpr int ROWS = 4;   // constraint rows per lane
#ifndef QUAD4_MKG
#define QUAD4_MKG 8       // rows per step the general build is unrolled for
#endif
// row stride of the M image: 18 (144 B: rows start in distinct 16-B slots, conflict-free stores) in the roomy carve, 16 in the slim one
constexpr int MK = 2;     // inequality rows per step

template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned x)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
constexpr int ROR8 = 0x128, ROR4 = 0x124, ROR2 = 0x122, ROR1 = 0x121;  // rotate within a row of 16

// all-reduce (min) over the 16 lanes of each row
__device__ __forceinline__ unsigned row_min(unsigned v)
{
    v = min(v, dpp_u<ROR8>(v));
    v = min(v, dpp_u<ROR4>(v));
    v = min(v, dpp_u<ROR2>(v));
    v = min(v, dpp_u<ROR1>(v));
    return v;
}
// value of lane `idx` (0..15; per lane, usually uniform inside a row) of the caller's own row
__device__ __forceinline__ int row_get(int x, int rb, int idx) { return __builtin_amdgcn_ds_bpermute((rb + idx) << 2, x); }
__device__ __forceinline__ double row_get(double x, int rb, int idx)
{
    const int a = (rb + idx) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
// true in every lane of a row iff `pred` holds in one of its lanes
__device__ __forceinline__ bool row_any(bool pred, int rb)
{
    const unsigned long long b = __ballot(pred);
    return ((unsigned)(b >> rb) & 0xffffu) != 0u;
}
// order-preserving map of a double onto two unsigned words
__device__ __forceinline__ void ordered(double x, unsigned &hi, unsigned &lo)
{
    const unsigned h = (unsigned)__double2hiint(x), l = (unsigned)__double2loint(x);
    const bool neg = h & 0x80000000u;
    hi = neg ? ~h : (h | 0x80000000u);
    lo = neg ? ~l : l;
}
__device__ __forceinline__ void ld16(double (&d)[NV], const double *src)
{
    const double2 *p = reinterpret_cast<const double2 *>(src);
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
        const double2 t = p[i];
        d[2 * i] = t.x;
        d[2 * i + 1] = t.y;
    }
}
__device__ __forceinline__ void st16(double *dst, const double (&s)[NV])
{
    double2 *p = reinterpret_cast<double2 *>(dst);
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
        double2 t;
        t.x = s[2 * i];
        t.y = s[2 * i + 1];
        p[i] = t;
    }
}
__device__ __forceinline__ double dot16(const double (&a)[NV], const double (&b)[NV])
{
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k += 4) {
        acc0 += a[k] * b[k];
        acc1 += a[k + 1] * b[k + 1];
        acc2 += a[k + 2] * b[k + 2];
        acc3 += a[k + 3] * b[k + 3];
    }
    return (acc0 + acc1) + (acc2 + acc3);
}
// two dot products with one vector, two chains each (a lone wavefront issues a dependent FMA every 8.5 cycles and an
// independent one every 5.1: four chains in flight are enough, and accumulators are registers the loop does not have)
__device__ __forceinline__ void dot16x2(const double (&a)[NV], const double (&b)[NV], const double (&x)[NV], double &ra, double &rb)
{
    double a0 = a[0] * x[0], a1 = a[1] * x[1], b0 = b[0] * x[0], b1 = b[1] * x[1];
#pragma unroll
    for (int k = 2; k < NV; k += 2) {
        a0 = fma(a[k], x[k], a0);
        b0 = fma(b[k], x[k], b0);
        a1 = fma(a[k + 1], x[k + 1], a1);
        b1 = fma(b[k + 1], x[k + 1], b1);
    }
    ra = a0 + a1;
    rb = b0 + b1;
}
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }

// The value held by lane N of the caller's 16-lane row, in every lane of that row (v_mov_b64_dpp row_newbcast:N).
template <int N> __device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_mov_dpp(x, 0x150 + N, 0xf, 0xf, true); }
// acc += (x of lane N of the caller's row) * m in ONE instruction (v_fmac_f64_dpp). The compiler cannot see inside the asm:
// a register written by a VALU instruction needs two wait states before a DPP read, so every batch of these is preceded by
// dpp_ready(x) on its broadcast source (tools/check_dpp_hazards.py verifies that on the assembly).
template <int N> __device__ __forceinline__ void fmac_bcast(double &acc, double x, double m)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(N));
}
__device__ __forceinline__ void dpp_ready(double &x) { asm volatile("s_nop 1" : "+v"(x)); }
// Compile-time loop: f(integral_constant<int, I>) for I = B .. E-1. The DPP lane select is an immediate, so the loops over lanes
// are unrolled by the front end (a `switch` on an unrolled loop's counter is only folded AFTER the unroller has priced the
// body with all sixteen cases in it -- the fused factorisation then exceeds the unroller's budget and stays a loop of jump tables).
template <int I> using ic = std::integral_constant<int, I>;
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}
// sum_k (x_k of lane k of the caller's row) * m[k]: a dot product with a vector spread over the row's lanes, two chains
__device__ __forceinline__ double dot_bcast(double x, const double (&m)[NV], double init)
{
    double a0 = init, a1 = 0.0;
    dpp_ready(x);
    static_for<0, NV / 2>([&](auto kk) {
        constexpr int k = 2 * decltype(kk)::value;
        fmac_bcast<k>(a0, x, m[k]);
        fmac_bcast<k + 1>(a1, x, m[k + 1]);
    });
    return a0 + a1;
}

// 1/x from the hardware estimate plus two Newton steps (operands are never subnormal or zero when the result is used)
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
// 1/sqrt(x) from the hardware estimate, one Newton step and one third-order step (x is a positive, normal number wherever the
// result is used: a pivot of a positive definite matrix, a squared row norm; the library's rsqrt spends two thirds of its
// instructions on subnormals and infinities)
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}
// The wavefronts of a workgroup share nothing and a wavefront's LDS operations complete in order: only the COMPILER has
// to keep the order of an exchange (no s_barrier, no queue drain).
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS carve of ONE problem, in doubles. Two of them, ONE wavefront per SIMD in both (the kernel holds 334 / 373 registers):
//  * ROOMY (up to three wavefronts per CU): M image 64 x 16, the full L^-T image, a T image for the rare refinement, the leaving slot's
//    row, the slots' constraint ids: 1560 doubles = 12.2 KB per problem, 49.9 KB per wavefront, three on a CU's 160 KB.
//  * SLIM (launches beyond three wavefronts per CU: 3073 problems and more on an MI355X): the M image and the strict upper triangle of
//    L^-T, packed (the diagonal stays in a register): 1152 doubles, 36.9 KB per wavefront, FOUR on a CU. Gone: the T image (T' rho by
//    row sums over the lanes), the leaving slot's row (fetched from its lane by ds_bpermute in the rare drop trip), the slots' ids
//    (DPP). 4096 problems with m = 64: 119 us roomy (two rounds of three per CU), 81 us slim (one round).
template <bool SLIM> struct Carve {
    static constexpr int LDM = 16;  // (64 x 16: three wavefronts of 49.9 KB on a CU; with 18 only two)
    static constexpr int OFF_M = 0;  // build: G image by column, 16 x GS (GS = 65: 1040 doubles, over the start of the region behind the
                                     // M image, which is not alive yet) | main: M image, 64 x LDM
    static constexpr int OFF_LT = MMAX * LDM;  // roomy: rows of L^-T, 16 x 16 | slim: strict upper triangle by rows, packed: row l at
                                               // l (31 - l) / 2, 15 - l entries
    static constexpr int NLT = SLIM ? NV * (NV - 1) / 2 + 8 : NV * NV;  // (slim: eight spare doubles keep the G image inside the carve)
    static constexpr int OFF_T = OFF_LT + NLT;                    // roomy only from here on: T by rows (refinement)
    static constexpr int OFF_KA = OFF_T + NV * NV;                // the row of a leaving slot
    static constexpr int OFF_ACT = OFF_KA + NV;                   // 16 int32: constraint held by each slot
    static constexpr int PER = SLIM ? OFF_LT + NLT : OFF_ACT + NV / 2;  // 640 | 1112 doubles per problem
    static_assert(NV * 65 <= PER, "the G image must fit the problem's carve");
    static_assert(PER % 2 == 0 && (!SLIM || PER * 4 * 8 <= 40 * 1024), "16-byte alignment; slim: 40 KB per wavefront (four on a CU)");
};

constexpr double DEP = 1e-14;      // |z|^2 / |M_p|^2 below this: M_p depends on the active rows
constexpr double DEP_FAST = 1e-6;  // K_p . M_p is trusted as |z|^2 only above this (mpcqp_pair.hip); |K_p|^2 otherwise

// a[i] for a lane-uniform-per-row index i = 0 .. 3 (straight-line selects)
template <typename X> __device__ __forceinline__ X pick(const X (&a)[4], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }

}  // namespace quad4

using namespace quad4;

// ORD: the launch carries a pairing order (MpcqpSolveOpts.order): row i of the launch takes problem order[i].
// WPB: wavefronts per workgroup (they share nothing).
// MODEL: the launch shares one factored model (mpcqp_factor_model: gA points at it); the per-problem vectors are x0, goal, targets and,
//        optionally, the bounds e. No build, no factorisation: M, L^-T and the linear maps of h and w are read from the model.
// GEN:   the build serves every cost and constraint layout of one to four rows per step (round 6): input rows D_k next to / instead of
//        the state rows C_k, a stage cost w_x sum |x_k - xref_k|^2 (the Gram matrix accumulated over every Psi_k of the chain, as
//        mpcqp_pair.hip's generic build does) -- the reference's own wheeled-inverted-pendulum example
//        (examples/wheeled_inverted_pendulum.py:90-94: input box, stage + terminal cost) is of this kind.
template <int NX, bool ORD, int WPB, bool SLIM, bool MODEL = false, bool GEN = false>
__global__ void __launch_bounds__(64 * WPB, 1)  // (one wavefront per SIMD in both carves: ~330 registers)
    mpcqp_quad4_kernel(const double *__restrict__ gA, const double *__restrict__ gB, const double *__restrict__ gC,
                      const double *__restrict__ ge, const double *__restrict__ gx0, const double *__restrict__ ggoal,
                      const double *__restrict__ gtgt, double *__restrict__ oU, double *__restrict__ olam, int32_t *__restrict__ ostatus,
                      int32_t *__restrict__ oiters, const KernelArgs ka, const int64_t batch)
{
    using T = double;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    {  // every kernel argument the operand addresses need, requested in ONE batch of scalar loads at the top (mpcqp_pair.hip)
        const int64_t b0 = ka.A.batch_stride, b1 = ka.B.batch_stride, b2 = ka.C.batch_stride, b3 = ka.e.batch_stride,
                      b4 = ka.x0.batch_stride, b5 = ka.goal.batch_stride;
        const int64_t s0 = ka.A.step_stride, s1 = ka.B.step_stride, s2 = ka.C.step_stride, s3 = ka.e.step_stride;
        asm volatile("" ::"s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(ka.N), "s"(ka.nu),
                     "s"(ka.flags), "s"(ka.probe), "s"(gA), "s"(gB), "s"(gC), "s"(ge), "s"(gx0), "s"(ggoal), "s"(ka.n), "s"(ka.m),
                     "s"(batch));
    }
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int rb = lane & 48;  // first lane of this row
    const int l = lane & 15;   // lane inside the row
    int64_t prob = 4 * ((int64_t)blockIdx.x * WPB + wv) + (rb >> 4);
    const bool valid = prob < batch;  // a batch that is no multiple of four leaves rows idle: they repeat the last problem, store nothing
    prob = valid ? prob : batch - 1;
    if constexpr (ORD) {  // (an index outside the batch is clamped: a bad order costs wrong pairings, never an access out of bounds)
        const int64_t o = ka.order[prob];
        prob = o < 0 ? 0 : (o >= batch ? batch - 1 : o);
    }
    using CV = Carve<SLIM>;
    constexpr int LDM = CV::LDM, PER = CV::PER;
    T *sm = (T *)smem_raw + (4 * wv + (rb >> 4)) * PER;
    const int n = ka.n, m = ka.m;
    int rowi[ROWS];   // the constraints of this lane: rows l, l + 16, l + 32, l + 48
    bool isc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        rowi[r] = l + NV * r;
        isc[r] = rowi[r] < m;
    }
    const T INF = HUGE_VAL;
    constexpr int GS = 65;  // the G image is stored by COLUMN with an odd stride
    T *Gimg = sm + CV::OFF_M, *Ml = sm + CV::OFF_M;
    T *LTp = sm + CV::OFF_LT + (l * (31 - l)) / 2 - (l + 1);  // slim: LTp[k] = entry (l, k) of L^-T, k > l
    T *LTimg = sm + CV::OFF_LT, *Timg = sm + CV::OFF_T, *kAv = sm + CV::OFF_KA;  // roomy
    int *actv = reinterpret_cast<int *>(sm + CV::OFF_ACT);

    // optional phase timestamps (developer probe, MpcqpSolveOpts.probe): long long[16] per problem, slots as mpcqp_pair.hip
    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 16 : nullptr;
    auto tick = [&](int slot) {
        if (stamp && l == 0 && valid) {
            stamp[slot] = (long long)__builtin_readcyclecounter();
            if (slot == 0 || slot == 6) stamp[slot ? 13 : 12] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    };
    tick(0);

    T hval[ROWS];
    bool notpd = false;
    T RM[ROWS][NV], RLt[NV];
    T wv_ = T(0);  // w = L^-1 q, component l
    static_assert(!MODEL && GEN && !ORD, "mpcqp_quad4.hip: the general build only");
    {
    // ---------------------------------------------------------------- build (mpc_qp.py:53-114)
    T Pr[NV];  // row l of P, then of L
    T qa;
    T g15[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) g15[r] = T(0);
    {
        // WIDE (round 6, NX = 8 / 12 / 16 as padded sizes for nx = 7 .. 16): the operands of a step do not fit registers for the whole
        // horizon -- they are streamed, two steps ahead of the chain, and the chain is a loop over the steps (see below)
        constexpr bool WIDE = GEN && NX > 6;
        const int nx = WIDE ? ka.nx : NX;
        const int nu = ka.nu, N = ka.N;
        // rows per step: two in the lean build; one to four in the general one (a run-time number: the chain's loops over the rows of
        // a step are unrolled four wide behind wavefront-uniform tests)
        constexpr int MKG = GEN ? QUAD4_MKG : MK;  // (this file: up to EIGHT rows per step -- m <= 64 holds them for horizons of up to eight steps)
        const int mk = GEN ? ka.mk : MK;
        auto stepof = [&](int row) { return mk == 2 ? row >> 1 : (mk == 1 ? row : (mk == 4 ? row >> 2 : (mk == 8 ? row >> 3 : row / mk))); };
        const T *A = gA + prob * ka.A.batch_stride;
        const T *B = gB + prob * ka.B.batch_stride;
        const bool hasC = !GEN || gC != nullptr;
        const T *Cm = hasC ? gC + prob * ka.C.batch_stride : A;  // (no state rows: the lanes load valid addresses, the products are dropped)
        const T *x0 = gx0 + prob * ka.x0.batch_stride;
        const T *goal = ggoal ? ggoal + prob * ka.goal.batch_stride : nullptr;
        const int sA = ka.A.step_stride ? nx * nx : 0, sB = ka.B.step_stride ? nx * nu : 0, sC = (hasC && ka.C.step_stride) ? mk * nx : 0;
        const bool termP = ka.flags & MPCQP_P_TERMINAL, termQ = (ka.flags & MPCQP_Q_TERMINAL) && goal;
        // (GEN) input rows and the stage cost
        const T *Dm = (GEN && ka.D.ptr) ? (const T *)ka.D.ptr + prob * ka.D.batch_stride : nullptr;
        const T *tgt = (GEN && gtgt) ? gtgt + prob * ka.targets.batch_stride : nullptr;
        const int sD = (GEN && ka.D.step_stride) ? mk * nu : 0;
        const bool stageP = GEN && (ka.flags & MPCQP_P_STAGE), stageQ = GEN && (ka.flags & MPCQP_Q_STAGE) && tgt;
        constexpr int NAe = NX * NX, NEe = NAe + MKG * NX;  // elements of [A_k | C_k]
        const int NEr = hasC ? NAe + mk * NX : NAe;          // ... that exist
        const bool col = (l < n);
        const int j = col ? (nu == 1 ? l : l / nu) : -1, ii = col ? l - j * nu : 0;  // (nu == 1: no division before the loads)
        // every load is issued before the first use: the whole build costs ONE HBM latency
        const int64_t eb = prob * ka.e.batch_stride;
        int kq[ROWS];
        T evl[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            kq[r] = stepof(rowi[r]);
            evl[r] = isc[r] ? ge[eb + kq[r] * ka.e.step_stride + (rowi[r] - kq[r] * mk)] : INF;
        }
        // The free response Phi_k x0 rides in LANE 15 of the row: that lane's own column belongs to the horizon's last step (or to
        // no variable at all), so it is zero in every G_k and enters Psi_N only as B's column itself -- its registers are idle for the
        // whole chain. The lane starts from x0 instead of zero, the chain's FMAs propagate it with everyone else's columns, and a row's
        // C_k Phi_k x0 is a row broadcast of what lane 15 computes as "its G entry" (round 5; before: a second chain of 15 FMAs per step
        // in every lane).
        const bool xl15 = (l == NV - 1);
        T v[NX], gref[NX], bcol[NX];
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            v[s] = (xl15 && s < nx) ? x0[s < nx ? s : 0] : T(0);
            gref[s] = (termQ && s < nx) ? goal[s < nx ? s : 0] : T(0);
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) bcol[r] = (col && r < nx) ? B[j * sB + (r < nx ? r : 0) * nu + ii] : T(0);
        // (GEN) this lane's column of D_j: the G entries of its variable, rows (j, 0) and (j, 1). Lane 15's cells of the image carry
        // the free response through the chain: when it owns a variable (n = 16: an input of the LAST step) the two rows of that step
        // fetch its entries of D straight from memory, behind the chain. Targets: lane e keeps xref element e, e + 16, e + 32, e + 48
        // (N nx <= 64), the chain fetches them as row broadcasts.
        T dcol[MKG], d15[ROWS], tg[NX];  // (targets: N nx <= 16 nx values, one per lane and register)
#pragma unroll
        for (int u = 0; u < NX; ++u) tg[u] = T(0);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) d15[r] = T(0);
#pragma unroll
        for (int i2 = 0; i2 < MKG; ++i2) dcol[i2] = T(0);
        if constexpr (GEN) {
            if (Dm) {
#pragma unroll
                for (int i2 = 0; i2 < MKG; ++i2) dcol[i2] = (col && !xl15 && i2 < mk) ? Dm[j * sD + i2 * nu + ii] : T(0);
                if (n == NV) {
                    const int j15 = nu == 1 ? NV - 1 : (NV - 1) / nu, i15 = NV - 1 - j15 * nu;
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) d15[r] = (isc[r] && kq[r] == j15) ? Dm[j15 * sD + (rowi[r] - kq[r] * mk) * nu + i15] : T(0);
                }
            }
            if (stageQ && !WIDE) {
#pragma unroll
                for (int u = 0; u < NX; ++u) tg[u] = (l + 16 * u < N * NX) ? tgt[l + 16 * u] : T(0);
            }
        }
        if constexpr (!WIDE) {
        // lane e of the row keeps element e (and e + 16) of [A_k | C_k] for every step k, straight from HBM; the chain
        // fetches an operand as a DPP row broadcast (lanes without an element load a valid address and are never read)
        constexpr int NOP = (NEe + 15) / 16;  // operand registers per step: two up to nx = 4, three / four for nx = 5 / 6
        T op[NOP][NV];
        static_for<0, NOP>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int e = l + 16 * r;
            const bool ok = e < NEr;
            const T *p = (e < NAe) ? A + e : Cm + (ok ? e - NAe : 0);
            const int st = (e < NAe) ? sA : sC;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int kc = (k < N) ? k : N - 1;
                op[r][k] = p[kc * st];
            }
        });
        tick(8);
        const T wu = (T)ka.wu;
        qa = T(0);
#pragma unroll
        for (int b = 0; b < NV; ++b) Pr[b] = (l == b) ? (col ? wu : T(1)) : T(0);
        tick(9);
        // acc += (element IDX of [A_k | C_k]) * xx
        auto mac = [&](auto idx, auto kc_, T &acc, T xx) {
            constexpr int IDX = decltype(idx)::value, K = decltype(kc_)::value;
            fmac_bcast<IDX % 16>(acc, op[IDX / 16][K], xx);
        };
        // [G_k; Psi_{k+1}] = [C_k; A_k] Psi_k (column l in lane l; lane 15: [C_k Phi_k x0; Phi_{k+1} x0])
        static_for<0, NV>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            if (k < N) {
                static_for<0, NOP>([&](auto rc) { dpp_ready(op[decltype(rc)::value][k]); });  // (as in the streamed build below)
                // (lane 15 stores C_k Phi_k x0 into column 15 of the image -- zero in G by construction --: the rows read their
                // entry of it behind the chain)
                if constexpr (!GEN) {
                    T g[MK];
                    static_for<0, MK>([&](auto i2c) {
                        constexpr int i2 = decltype(i2c)::value;
                        T acc = T(0);
                        static_for<0, NX>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value;
                            mac(ic<NAe + i2 * NX + s2>{}, kk, acc, v[s2]);
                        });
                        g[i2] = acc;
                    });
#pragma unroll
                    for (int i2 = 0; i2 < MK; ++i2) Gimg[l * GS + k * MK + i2] = g[i2];
                } else {
                    static_for<0, MKG>([&](auto i2c) {
                        constexpr int i2 = decltype(i2c)::value;
                        if (i2 < mk) {  // (wavefront-uniform)
                            T acc = T(0);
                            if (hasC) {
                                static_for<0, NX>([&](auto sc) {
                                    constexpr int s2 = decltype(sc)::value;
                                    mac(ic<NAe + i2 * NX + s2>{}, kk, acc, v[s2]);
                                });
                            }
                            acc += (j == k) ? dcol[i2] : T(0);
                            Gimg[l * GS + k * mk + i2] = acc;
                        }
                    });
                }
                if constexpr (GEN && k >= 1) {
                    // stage cost on x_k: P += w_x Psi_k' Psi_k, q += w_x Psi_k' (Phi_k x0 - xref_k)  (mpc_qp.py:99-105, 129-149; lane 15's
                    // own column is zero in every Psi_k of the chain, its registers hold the free response)
                    if (stageP || stageQ) {
                        const T wxs = (T)ka.wx;
                        static_for<0, NX>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value;
                            constexpr int te = k * NX + s2;
                            T src = xl15 ? T(0) : v[s2];
                            const T t = wxs * src;
                            if (stageQ) qa += t * (row_bcast<NV - 1>(v[s2]) - row_bcast<te % 16>(tg[te / 16]));
                            if (stageP) {
                                dpp_ready(src);
                                static_for<0, NV>([&](auto bc) { fmac_bcast<decltype(bc)::value>(Pr[decltype(bc)::value], src, t); });
                            }
                        });
                    }
                }
                // column j of Psi_k is zero up to step j, so B_j's column enters as the start value of lane j's sums (lane 15 carries
                // the free response: its own column is put in place behind the chain)
                const T hk = (j == k && !xl15) ? T(1) : T(0);
                T w[NX];
                static_for<0, NX>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    T acc = hk * bcol[r];
                    static_for<0, NX>([&](auto sc) {
                        constexpr int s2 = decltype(sc)::value;
                        mac(ic<r * NX + s2>{}, kk, acc, v[s2]);
                    });
                    w[r] = acc;
                });
#pragma unroll
                for (int r = 0; r < NX; ++r) v[r] = w[r];
            }
        });
        } else {
        tick(8);
        const T wu = (T)ka.wu;
        qa = T(0);
#pragma unroll
        for (int b = 0; b < NV; ++b) Pr[b] = (l == b) ? (col ? wu : T(1)) : T(0);
        tick(9);
        // ---- streamed operands. Element e of a step's [A_k (NX x NX) | C_k (4 x NX) | xref_k (NX)] -- padded to NX, entries beyond nx /
        // mk read as zero -- sits in lane e % 16 of register e / 16; every register holds one kind of element (NX^2 and 4 NX are
        // multiples of 16). Two steps' registers are alive: step k + 2 is requested when step k has been consumed.
        constexpr int NAw = NX * NX, NCw = MKG * NX, NEw = NAw + NCw + NX, NOPW = (NEw + 15) / 16;
        T opw[2][NOPW];
        auto fetch = [&](auto dc, int k) {
            constexpr int d = decltype(dc)::value;
            const int kc = k < N ? k : N - 1;
            static_for<0, NOPW>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr int e0 = 16 * r;
                const int e = e0 + l;
                T val;
                if constexpr (e0 < NAw) {
                    const int rr = e / NX, ss = e - rr * NX;
                    const bool ok = rr < nx && ss < nx;
                    val = A[kc * sA + (ok ? rr * nx + ss : 0)];
                    val = ok ? val : T(0);
                } else if constexpr (e0 < NAw + NCw) {
                    const int e2 = e - NAw, i2 = e2 / NX, ss = e2 - i2 * NX;
                    const bool ok = hasC && i2 < mk && ss < nx;
                    val = Cm[ok ? kc * sC + i2 * nx + ss : 0];
                    val = ok ? val : T(0);
                } else {
                    const int ss = e - NAw - NCw;
                    const bool ok = stageQ && ss < nx;
                    val = ok ? tgt[kc * nx + ss] : T(0);
                }
                opw[d][r] = val;
            });
        };
        fetch(ic<0>{}, 0);
        fetch(ic<1>{}, 1);
        const T wxs = (T)ka.wx;
        auto cstep = [&](auto dc, int k) {
            constexpr int d = decltype(dc)::value;
            // (the step's operand registers are DPP sources of hand-written instructions: pinned in vector registers two wait states
            // ahead -- an instantiation that runs on more than 256 registers may otherwise fetch one from an accumulation register
            // right in front of its first use)
            static_for<0, NOPW>([&](auto rc) { dpp_ready(opw[d][decltype(rc)::value]); });
            // G rows of step k from Psi_k (lane 15: C_k Phi_k x0)
            static_for<0, MKG>([&](auto i2c) {
                constexpr int i2 = decltype(i2c)::value;
                if (i2 < mk) {  // (wavefront-uniform)
                    T acc = T(0);
                    if (hasC) {
                        static_for<0, NX>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value, E = NAw + i2 * NX + s2;
                            fmac_bcast<E % 16>(acc, opw[d][E / 16], v[s2]);
                        });
                    }
                    acc += (j == k) ? dcol[i2] : T(0);
                    Gimg[l * GS + k * mk + i2] = acc;
                }
            });
            // stage cost on x_k (k >= 1)
            if (k >= 1 && (stageP || stageQ)) {
                static_for<0, NX>([&](auto sc) {
                    constexpr int s2 = decltype(sc)::value, E = NAw + NCw + s2;
                    T src = xl15 ? T(0) : v[s2];
                    const T t = wxs * src;
                    if (stageQ) qa += t * (row_bcast<NV - 1>(v[s2]) - row_bcast<E % 16>(opw[d][E / 16]));
                    if (stageP) {
                        dpp_ready(src);
                        static_for<0, NV>([&](auto bc) { fmac_bcast<decltype(bc)::value>(Pr[decltype(bc)::value], src, t); });
                    }
                });
            }
            // Psi_{k+1} = A_k Psi_k, B_k's column entering at step j
            const T hk = (j == k && !xl15) ? T(1) : T(0);
            T w[NX];
            static_for<0, NX>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                T acc = hk * bcol[r];
                static_for<0, NX>([&](auto sc) {
                    constexpr int s2 = decltype(sc)::value, E = r * NX + s2;
                    fmac_bcast<E % 16>(acc, opw[d][E / 16], v[s2]);
                });
                w[r] = acc;
            });
#pragma unroll
            for (int r = 0; r < NX; ++r) v[r] = w[r];
            fetch(dc, k + 2);
        };
        for (int k = 0; k < N; k += 2) {
            cstep(ic<0>{}, k);
            if (k + 1 < N) cstep(ic<1>{}, k + 1);
        }
        }
        tick(10);
        // P = wu I + wt psi_N' psi_N ; q = wt psi_N' (Phi_N x0 - goal)   (mpc_qp.py:99-105, 129-149)
        const T wt = (T)ka.wt;
        // Phi_N x0 from lane 15, whose own column of Psi_N is B's column of the last step (or nothing)
        T x[NX];
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            x[s] = row_bcast<NV - 1>(v[s]);
            v[s] = xl15 ? bcol[s] : v[s];
        }
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            const T t = wt * v[s];
            if (termQ) qa += t * (x[s] - gref[s]);
            if (termP) {
                T src = v[s];
                dpp_ready(src);
                static_for<0, NV>([&](auto bc) { fmac_bcast<decltype(bc)::value>(Pr[decltype(bc)::value], src, t); });
            }
        }
        qa = col ? qa : T(0);
        tick(11);
        wsync();  // the G image is complete
#pragma unroll
        for (int r = 0; r < ROWS; ++r) hval[r] = isc[r] ? evl[r] - Gimg[(NV - 1) * GS + rowi[r]] : INF;  // h_i = e_i - C_k Phi_k x0 (column 15 of the image)
        if constexpr (GEN) {  // (column 15 of G: the last step's input rows)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) g15[r] = d15[r];
        }
    }
    tick(1);
    // ------------------------------------------------------------ factorise + forward substitution, one pass
    // Right-looking Cholesky; step j scales column j of L and applies it at once to everything that waits for it:
    //   P[:, k] -= L[:, j] L[k][j]                       (trailing columns of P)
    //   x[k]    -= (x[j] / L_jj) L[k][j], x[j] /= L_jj   for the rows x = G_l, G_{l+16}, e_l: -> M_l, M_{l+16}, (L^-T)_l
    //   q_k     -= L[k][j] w_j, w_j = q_j / L_jj         (q_k in lane k: L[k][j] is local, w_j the broadcast)
    // L[k][j] of lane k is a DPP row broadcast; the four FMA streams are independent of each other.
    {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            // (column 15 of G is zero: the column of the horizon's last step, or of no variable -- its cells hold the free response)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) RM[r][k] = (isc[r] && k < NV - 1) ? Gimg[k * GS + rowi[r]] : (GEN && k == NV - 1 ? g15[r] : T(0));
            RLt[k] = (l == k) ? T(1) : T(0);
        }
        static_for<0, NV>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const T pij = Pr[j];                 // P[l][j] of this lane's row, before scaling
            const T piv = row_bcast<j>(pij);     // P[j][j]
            if (!(piv > T(0))) notpd = true;
            const T rinv = fast_rsqrt(piv);
            const T nt2 = -(pij * rinv * rinv);  // -P[l][j] / piv
            T nl = -(pij * rinv);                // -L[l][j] (column j of L is never read after this step: not kept)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) RM[r][j] *= rinv;
            RLt[j] *= rinv;
            const T wj = qa * rinv;              // lane j: w_j
            wv_ = (l == j) ? wj : wv_;
            {
                T src = pij, wsrc = wj;
                asm volatile("s_nop 1" : "+v"(src), "+v"(nl), "+v"(wsrc) : "v"(RM[0][j]), "v"(RM[1][j]), "v"(RM[2][j]), "v"(RM[3][j]), "v"(RLt[j]));  // (the DPP wait states)
                fmac_bcast<j>(qa, wsrc, nl);  // q_k -= L[k][j] w_j (lanes k <= j hold values nobody reads again)
                static_for<j + 1, NV>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    fmac_bcast<k>(Pr[k], src, nt2);     // P[k][j] from lane k
                    fmac_bcast<k>(RM[0][k], nl, RM[0][j]);  // -L[k][j] from lane k
                    fmac_bcast<k>(RM[1][k], nl, RM[1][j]);
                    fmac_bcast<k>(RM[2][k], nl, RM[2][j]);
                    fmac_bcast<k>(RM[3][k], nl, RM[3][j]);
                    fmac_bcast<k>(RLt[k], nl, RLt[j]);
                });
            }
        });
        wsync();  // the M image below reuses the G image
    }
    }  // (!MODEL)
    tick(2);
    tick(3);
    int status = MPCQP_MAX_ITER, iters = 0;
    T xsol = T(0);
    bool done = notpd;      // this row has left the active-set loop
    bool finished = notpd;  // ... and needs no refinement any more (failed, or accepted)
    if (notpd) status = MPCQP_NOT_PD;

#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        if (isc[r]) st16(Ml + rowi[r] * LDM, RM[r]);  // image of M: row-p broadcasts, the active rows in the refinement
    // the rows of L^-T leave the registers. Slim: L^-T is upper triangular (the identity pushed through a forward substitution with a
    // lower triangular factor): the strict upper part packed in LDS, the diagonal in a register
    T ltd = T(0);
    if constexpr (SLIM) {
        static_for<0, NV>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            ltd = (l == k) ? RLt[k] : ltd;
            if (k > l) LTp[k] = RLt[k];
        });
    } else {
        st16(LTimg + l * NV, RLt);
    }
    const T y0 = -wv_;          // y0 = -L^-1 q, component l
    T s[ROWS], invn[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        // (computed by EVERY lane: a DPP read from a lane that a branch has switched off returns zero)
        const T d = dot_bcast(wv_, RM[r], hval[r]);  // h - M y0 = h + M w
        s[r] = isc[r] ? d : INF;
        const T nn = dot16(RM[r], RM[r]);
        invn[r] = (nn > T(0)) ? fast_rsqrt(nn) : T(1);
    }
    // Selection rule (the classic Goldfarb-Idnani one): among the rows violated beyond the tolerance, the one FARTHEST
    // from its hyperplane in the P^-1 metric, s_i / |M_i|.
    bool sel[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) sel[r] = isc[r] && (hval[r] < T(1e29));
    const T tol = (T)ka.tol;
    T tolh[ROWS];  // row i is violated when s_i < -tol (1 + |h_i|)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) tolh[r] = tol + tol * fabs(hval[r]);
    const int max_iter = ka.max_iter;
    T RT[NV], RH[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        RT[k] = T(0);                      // T = N* starts empty
        RH[k] = (l == k) ? T(1) : T(0);    // H = I
    }
    T lam = T(0);      // multiplier of slot l
    int myact = 0;     // constraint held by slot l
    bool occ = false;  // slot l occupied
    bool e[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) e[r] = sel[r];
    // e: this lane's constraints may be selected: they have a bound and are not active (the slack of an
                                // active row is never read: it stays whatever the steps make of it, zero up to rounding)
    // row-uniform state
    int nq = 0, p = 0, ldrop = 0;
    unsigned mask = 0;  // occupied slots
    bool needp = true, dropping = false;
    T up = T(0);
    int fails = 0;
    // ---- selection, for the rows that start a new constraint (straight-line selects: no divergent branches), and the
    //      fetch of row p of M (a broadcast read inside the row). Called between the two halves of the (deferred) rank-one
    //      update: the update's first FMAs cover the reduction's dependent chain, its last ones the LDS round trip.
    T mp[NV];
    auto select = [&]() {
        // a violated row's scaled slack is negative: the order of the magnitudes is the order of the high words, so
        // the most violated row has the smallest complement
        const bool want = needp & !done & !dropping;
        unsigned kmin = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const unsigned h = ~(unsigned)__double2hiint(s[r] * invn[r]);
            const bool v = want & e[r] & (s[r] < -tolh[r]);
            const unsigned kk = v ? ((h & ~63u) | (unsigned)rowi[r]) : 0xffffffffu;
            kmin = min(kmin, kk);
        }
        const unsigned mkey = row_min(kmin);
        const bool none = want & (mkey == 0xffffffffu);
        const bool got = want & !none;
        done = done | none;
        status = none ? (int)MPCQP_SOLVED : status;
        p = got ? (int)(mkey & 63u) : p;
        up = got ? T(0) : up;
        needp = needp & !got;
    };
    // R += c v for the two maintained register rows, v spread over the row (component k in lane k), columns B .. E-1
    auto update = [&](auto bc, auto ec, T zn, T cT, T cH) {
        static_for<decltype(bc)::value, decltype(ec)::value>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            fmac_bcast<k>(RT[k], zn, cT);
            fmac_bcast<k>(RH[k], zn, cH);
        });
    };
#ifndef QUAD_USPLIT
#define QUAD_USPLIT 8
#endif
    constexpr int USPLIT = QUAD_USPLIT;  // columns of the update issued before the selection's row fetch
    // pending rank-one update R += c v, applied at the top of the next trip -- the ONE site that writes the register rows,
    // selects and fetches
    T zn = T(0), cT = T(0), cH = T(0);
    wsync();
    tick(4);
    for (;;) {
        // ===================================================== active-set loop
        for (;;) {
            // ---- the previous trip's rank-one update  T_a += (r_a/d2) z, T_new = -z/d2 ; H -= z z'/d2  (or the same with a
            //      leaving slot's T_l), wrapped around this trip's selection and row fetch, which only read the slacks
            dpp_ready(zn);
            update(ic<0>{}, ic<USPLIT>{}, zn, cT, cH);
            select();
            ld16(mp, Ml + p * LDM);
            update(ic<USPLIT>{}, ic<NV>{}, zn, cT, cH);
            cT = cH = T(0);
            if (__ballot(!done) == 0ull) break;
            const bool st = !done & !dropping;  // this row steps
            const bool drp = !done & dropping;  // ... or drops a slot
            const int pr = p >> 4;              // (row-uniform) which of the lane's registers row p is
            const int pl = p & 15;
            // ---- r_a = T_a . M_p ; -z_l = H_l . M_p ; then -M_i . z = sum_k M_i[k] (-z_k) for this lane's two rows, with -z
            //      spread over the row: the projected rows K_i = H M_i of mpcqp_pair.hip are NOT maintained (32 FMAs of update
            //      and 32 of dot products per trip against the 32 of these two)
            const T hd = dot16(RH, mp);
            T kd[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) kd[r] = dot_bcast(hd, RM[r], T(0));
            // ---- step length, by lane p from its own row: |z|^2 = -M_p . z (trusted well away from dependence, DEP_FAST),
            //      1/|z|^2, t2 = -s_p/|z|^2; the two results go to the row (inv = -1: too close to dependence)
            T inv, t2;
            {
                const T kp = pick(kd, pr), sq = pick(s, pr), iq = pick(invn, pr);
                const bool okf = kp * iq * iq > T(DEP_FAST);
                const T iv = fast_rcp(kp);
                inv = row_get(okf ? iv : T(-1), rb, pl);
                t2 = row_get(-sq * iv, rb, pl);
            }
            T rd = dot16(RT, mp);  // (while the exchange is in flight)
            pin(rd);
            if (__ballot(st & !(inv > T(0))) != 0ull) {  // rare: |z|^2 as a sum of squares, robust near dependence
                T z2 = hd * hd;
                z2 += dpp_mov<ROR8>(z2);
                z2 += dpp_mov<ROR4>(z2);
                z2 += dpp_mov<ROR2>(z2);
                z2 += dpp_mov<ROR1>(z2);
                z2 = row_bcast<0>(z2);  // (the rotations sum in another order in every lane)
                const T sq = row_get(pick(s, pr), rb, pl), iq = row_get(pick(invn, pr), rb, pl);
                const bool ok2 = (z2 * iq * iq > T(DEP)) & (z2 > T(0));
                const T iv2 = fast_rcp(z2);
                const bool nearp = !(inv > T(0));
                t2 = nearp ? -sq * iv2 : t2;
                inv = nearp ? (ok2 ? iv2 : T(0)) : inv;
            }
            const bool can_move = (nq < n) & (inv > T(0));
            inv = (can_move & st) ? inv : T(0);
            t2 = can_move ? t2 : INF;
            const int sl = (int)__builtin_ctz(~mask);  // lowest free slot
            const T r = (occ & st) ? rd : T(0);
            // a blocking multiplier exists iff lam_a / r_a < t2 for some slot
            const bool blk = (r > T(0)) & (lam < t2 * r);
            if (__ballot(drp | (st & (!can_move | (iters >= max_iter) | blk))) == 0ull) {
                // ---- PLAIN TRIP: every row still in the loop takes a full step. Coefficients of the (deferred) update with
                //      -z: slot sl takes -z/d2, the occupied slots r_a/d2, H row l -z_l/d2
                const T tt = st ? t2 : T(0);
                const bool isnew = st & (l == sl), isp = st & (l == pl);
                iters += st ? 1 : 0;
                zn = hd;
                cT = (l == sl) ? inv : -(r * inv);
                cH = -(hd * inv);
#pragma unroll
                for (int r = 0; r < ROWS; ++r) s[r] = fma(tt, kd[r], s[r]);  // s_i -= t M_i . z
                T ln = fma(-tt, r, lam);
                ln = (occ & (ln < T(0))) ? T(0) : ln;
                lam = isnew ? up + tt : ln;
                myact = isnew ? p : myact;
                occ = occ | isnew;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) e[r] = e[r] & !(isp & (pr == r));
                mask |= st ? (1u << sl) : 0u;
                nq += st ? 1 : 0;
                needp = needp | st;
                continue;
            }
            // ---- GENERAL TRIP: limits, partial steps, drops
            bool stepping = st;
            {
                const bool lim = stepping & (iters >= max_iter);
                done = done | lim;
                finished = finished | lim;
                status = lim ? (int)MPCQP_MAX_ITER : status;
                stepping = stepping & !lim;
            }
            iters += stepping ? 1 : 0;
            const bool cand = occ & stepping & (r > T(0));
            T t1 = INF;
            int lq = 0;
            const unsigned long long bl = __ballot(stepping & blk);
            if (bl != 0ull) {  // ratio test on the multipliers
                const T ratio = cand ? lam * fast_rcp(r) : INF;
                unsigned hi, lo;
                ordered(ratio, hi, lo);
                hi = cand ? hi : 0xffffffffu;
                const unsigned mhi = row_min(hi);
                const unsigned k2 = (cand && hi == mhi) ? ((lo & ~31u) | (unsigned)l) : 0xffffffffu;
                const unsigned ml = row_min(k2);
                lq = (int)(ml & 15u);
                const T tl1 = row_get(ratio, rb, lq);
                // (only for the rows that are blocked: a row's result must not depend on what its wavefront's other rows need)
                const bool blocked = ((unsigned)(bl >> rb) & 0xffffu) != 0u;
                t1 = (blocked && mhi != 0xffffffffu) ? tl1 : INF;
            }
            T t = t1 < t2 ? t1 : t2;
            {
                const bool inf = stepping & !(t < INF);  // no step possible: the constraints are inconsistent
                done = done | inf;
                finished = finished | inf;
                status = inf ? (int)MPCQP_INFEASIBLE : status;
                stepping = stepping & !inf;
            }
            t = stepping ? t : T(0);
            const bool full = stepping & (t2 <= t1);
            zn = hd;
            cT = full ? ((l == sl) ? inv : -(r * inv)) : T(0);
            cH = full ? -(hd * inv) : T(0);
            if (__ballot(drp) != 0ull) {
                // slot ldrop leaves (its row T_l still sits in lane ldrop -- a partial step has no update of its own --, roomy: and in kAv).
                // With W = T T' implicit, T_a -= (T_a . T_l / T_l . T_l) T_l
                // (row l becomes exactly zero); the null space of the active rows gains the direction T_l:
                // H += T_l T_l' / T_l . T_l.
                T vv[NV];
                T vl = T(0);
                if constexpr (SLIM) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) {  // (fetched from its lane: a rare trip, and LDS has no room for the row)
                        vv[k] = row_get(RT[k], rb, ldrop);
                        vl = (l == k) ? vv[k] : vl;
                    }
                } else {
                    ld16(vv, kAv);
                    vl = kAv[l];
                }
                const T tl = dot16(RT, vv);
                const T tld = row_get(tl, rb, ldrop);
                const T itl = fast_rcp(tld);
                if (drp) {
                    zn = vl;
                    cT = (l == ldrop) ? T(-1) : (occ ? -tl * itl : T(0));
                    cH = vl * itl;
                    if (l == ldrop) {
                        lam = T(0);
                        occ = false;
                    }
                    mask &= ~(1u << ldrop);
                    --nq;
                    dropping = false;
                }
            }
            // ---- bookkeeping: the implied primal point moved by t z: s_i -= t M_i . z
            if (stepping) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) s[r] = fma(t, kd[r], s[r]);
                lam -= t * r;
                lam = (occ && lam < T(0)) ? T(0) : lam;
                up += t;
            }
            if (full) {  // p takes slot sl
                if (l == sl) {
                    lam = up;
                    myact = p;
                    occ = true;
                }
                if (l == pl) {
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) e[r] = e[r] & (pr != r);
                }
                mask |= 1u << sl;
                ++nq;
                needp = true;
            }
            const bool partial = stepping & !full;
            if (__ballot(partial) != 0ull) {
                // partial step: the next trip removes slot lq from T (no update is pending for this row: RT is current)
                const int cl = row_get(myact, rb, lq);
                if constexpr (!SLIM) {
                    wsync();
                    if (partial && l == lq) st16(kAv, RT);
                }
                if (partial) {
                    if (l == (cl & 15)) {  // the row that leaves may be selected again
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) e[r] = ((cl >> 4) == r) ? sel[r] : e[r];
                    }
                    dropping = true;
                    ldrop = lq;
                }
                if constexpr (!SLIM) wsync();
            }
        }
        tick(5);
        if (__ballot(!finished) == 0ull) break;
        // ================================== multipliers by refinement, slacks re-evaluated
        // (rows that are already finished compute along and change nothing)
        int aa[NV];  // constraint held by each slot (an empty one: row 0 with a zero coefficient)
        if constexpr (SLIM) {
            const int mine = occ ? myact : 0;
            static_for<0, NV>([&](auto ac) { aa[decltype(ac)::value] = __builtin_amdgcn_update_dpp(0, mine, 0x150 + decltype(ac)::value, 0xf, 0xf, false); });
        } else {
            actv[l] = occ ? myact : 0;
            wsync();
            const int4 *ap = reinterpret_cast<const int4 *>(actv);
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) {
                const int4 t4 = ap[q];
                aa[4 * q] = t4.x;
                aa[4 * q + 1] = t4.y;
                aa[4 * q + 2] = t4.z;
                aa[4 * q + 3] = t4.w;
            }
        }
        // (M_A' cf)_l, cf_a in lane a (an empty slot carries a zero coefficient)
        auto ma_dot = [&](T cf) {
            T ma[NV];
#pragma unroll
            for (int a = 0; a < NV; ++a) ma[a] = Ml[aa[a] * LDM + l];
            return dot_bcast(cf, ma, T(0));
        };
        // slacks of this lane's two rows at the point y (component k in lane k)
        T fresh[ROWS];
        auto slacks = [&](T yv) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const T f = dot_bcast(yv, RM[r], T(0));
                fresh[r] = isc[r] ? hval[r] - f : INF;
            }
        };
        // value of this slot's own constraint row (register myact / 16 of lane myact % 16)
        auto of_act = [&](const T (&xv)[ROWS]) {
            T got[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) got[r] = row_get(xv[r], rb, myact & 15);
            return pick(got, myact >> 4);
        };
        T y = y0 - ma_dot(occ ? lam : T(0));  // y = y0 - M_A' lam
        slacks(y);
        // active residuals rho_a = h_a - M_a y should vanish. When they already do to REFTOL (1 + |h_a|) in every row --
        // the usual case: a dozen rank-one updates of T in float64 -- the refinement step below would move y by less than
        // that and is skipped.
        T rho = of_act(fresh);  // (fetched by EVERY lane: an exchange only sees the lanes that take part in it)
        rho = occ ? rho : T(0);
        constexpr double REFTOL = 1e-11;
        const T hact = of_act(hval);
        const bool needref = occ & !finished & !(fabs(rho) <= T(REFTOL) * (T(1) + fabs(hact)));
        const unsigned long long nr = __ballot(needref);
        if (nr != 0ull) {
            rho = (((unsigned)(nr >> rb) & 0xffffu) != 0u) ? rho : T(0);  // (rows that need none take a zero step: see the ratio test)
            // dlam = -W rho_A = -T (T' rho_A). (T' rho)_k = sum_a T_a[k] rho_a: lane a holds row a of T, so the sum runs over the
            // lanes -- slim: sixteen row reductions, lane k keeps the k-th (a rare path); roomy: through an image of T in LDS
            T uk = T(0);
            if constexpr (SLIM) {
                static_for<0, NV>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    T v = RT[k] * rho;
                    v += dpp_mov<ROR8>(v);
                    v += dpp_mov<ROR4>(v);
                    v += dpp_mov<ROR2>(v);
                    v += dpp_mov<ROR1>(v);
                    uk = (l == k) ? v : uk;
                });
            } else {
                st16(Timg + l * NV, RT);
                wsync();
                T tc[NV];
#pragma unroll
                for (int a = 0; a < NV; ++a) tc[a] = Timg[a * NV + l];
                uk = dot_bcast(rho, tc, T(0));  // (T' rho)_l
            }
            T dl = -dot_bcast(uk, RT, T(0));
            dl = occ ? dl : T(0);
            if (!finished) {
                const T lraw = lam + dl;
                lam = (occ && lraw < T(0)) ? T(0) : lraw;
            }
            // with dl as it is (not clamped) y moves exactly onto the active hyperplanes: y - M_A' dl = y + M_A' T (T' rho)
            // = y + T' rho, because T' rho lies in the range of M_A' where M_A' T = I - H is the identity
            y += uk;
            wsync();
            slacks(y);
        }
        // ---- acceptance: no inactive row violated, every active row on its bound, lam >= 0 -- with stationarity by
        //      construction these are the KKT conditions of the strictly convex QP
        bool viol = false;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) viol = viol || (e[r] && !(fresh[r] >= -T(4) * tolh[r]));
        bool dirty = row_any(viol, rb);
        {
            const T ra = of_act(fresh);
            const T ta = of_act(tolh);
            const bool off = row_any(occ && !(fabs(ra) <= T(1e3) * ta), rb);
            const bool neg = row_any(occ && !(lam >= T(0)), rb);
            if (stamp && l == 0 && valid && !finished && done) {  // developer probe: why the last acceptance test failed
                T worst = T(0);
                stamp[14] = (long long)(dirty ? 1 : 0) | (off ? 2 : 0) | (neg ? 4 : 0) | ((long long)fails << 8) | ((long long)nq << 16);
                (void)worst;
            }
            dirty = dirty || off || neg;
        }
        // u = L^-T y (component l). Slim: the diagonal from its register, the strict upper triangle from its packed image
        auto primal = [&]() {
            T lt[NV];
            if constexpr (SLIM) {
                static_for<0, NV>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    lt[k] = (k > l) ? LTp[k > l ? k : l + 1] : ((k == l) ? ltd : T(0));
                });
            } else {
                ld16(lt, LTimg + l * NV);
            }
            return dot_bcast(y, lt, T(0));
        };
        if (!finished && done) {
            if (!dirty) {
                xsol = primal();
                status = MPCQP_SOLVED;
                finished = true;
            } else if (++fails < 4) {
                // continue the active-set loop from the re-evaluated slacks
#pragma unroll
                for (int r = 0; r < ROWS; ++r) s[r] = fresh[r];
                status = MPCQP_MAX_ITER;
                done = false;
                needp = true;
            } else {
                xsol = primal();
                status = MPCQP_MAX_ITER;
                finished = true;
            }
        }
        wsync();
        if (__ballot(!finished) == 0ull) break;
    }
    tick(6);
    const bool ok = (status == MPCQP_SOLVED);
    T lo[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) lo[r] = T(0);
    if (olam) {  // multipliers by constraint: every occupied slot drops its multiplier at its row's place
        T *lamv = Ml;  // (the M image is dead: every row has finished; 64 doubles)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) lamv[rowi[r]] = T(0);
        wsync();
        if (occ) lamv[myact] = lam;
        wsync();
#pragma unroll
        for (int r = 0; r < ROWS; ++r) lo[r] = ok ? lamv[rowi[r]] : T(0);
    }
    if (valid) {
        if (l < n) oU[prob * (int64_t)n + l] = ok ? xsol : T(0);
        if (olam) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if (isc[r]) olam[prob * (int64_t)m + rowi[r]] = lo[r];
        }
        if (l == 0) {
            if (ostatus) ostatus[prob] = status;
            if (oiters) oiters[prob] = iters;
        }
    }
}

// ------------------------------------------------------------ host side
// cold launches of problems with n <= 16 and 33 .. 64 rows, or five to eight rows per step (m <= 64), nx <= 8 (the streamed build's
// padded size 8 serves nx = 7, 8); everything else keeps the kernel it had
#ifndef MPCQP_QUAD_WIDE_UNIT
bool quad4_applies(const KernelArgs &ka)
{
    if (ka.n > NV || ka.m > MMAX || (ka.m <= 32 && ka.mk <= 4) || ka.nx < 2 || ka.nx > 8) return false;  // (m <= 32 with mk <= 4: mpcqp_quad.hip)
    if (ka.mk < 1 || ka.mk > 8 || (!ka.C.ptr && !ka.D.ptr)) return false;
    if (ka.N * ka.mk != ka.m || ka.N > NV) return false;
    if (ka.warm_state || ka.order || (ka.opt_flags & (MPCQP_OPT_SEED_VIOLATED | MPCQP_OPT_TWO_PER_WAVE))) return false;
    return true;
}

#endif

template <int NX> static int launch_quad4_t(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    const int64_t waves = (batch + 3) / 4;
    auto go = [&](auto kern, size_t per) -> int {
        const size_t bytes = per * 4 * sizeof(double);
        if (bytes > 48 * 1024) {  // (per call: no state between calls)
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)waves), dim3(64), bytes, st, (const double *)ka.A.ptr, (const double *)ka.B.ptr,
                           (const double *)ka.C.ptr, (const double *)ka.e.ptr, (const double *)ka.x0.ptr, (const double *)ka.goal.ptr,
                           (const double *)ka.targets.ptr, (double *)ka.U, (double *)ka.lam, ka.status, ka.iters, ka, batch);
        return (int)hipGetLastError();
    };
    // The roomy carve (49.9 KB per wavefront) puts three wavefronts on a CU, the slim one (36.9 KB: packed L^-T, no T image, the leaving
    // slot's row by ds_bpermute) four: launches that do not fit three per CU take the slim one (STILL one wavefront per SIMD: the kernel
    // holds ~330 registers in either)
    const bool slim = waves > 3 * (int64_t)(device_simds_now() / 4);
    return slim ? go(mpcqp_quad4_kernel<NX, false, 1, true, false, true>, Carve<true>::PER)
                : go(mpcqp_quad4_kernel<NX, false, 1, false, false, true>, Carve<false>::PER);
}

// (compiled twice, like mpcqp_quad.hip: as itself -- nx = 2 .. 4 -- and through mpcqp_quad4w.hip -- nx = 5 .. 8, the streamed build --)
#ifdef MPCQP_QUAD_WIDE_UNIT
int launch_quad4_wide(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    // nx = 5 .. 8: the streamed build in its padded size 8. (Not the all-steps-in-registers build for nx = 5, 6: with eight rows per
    // step it keeps 80 / 96 operand registers alive next to this kernel's ~330, the allocator parks some in accumulation registers and
    // moves them back right in front of the hand-written v_fmac_f64_dpp that reads them -- without the two wait states a DPP read
    // needs: wrong plans (tools/check_dpp_hazards.py finds some of those sites, the oracle found the rest).)
    return launch_quad4_t<8>(ka, batch, st);
}
#else
int launch_quad4_wide(const KernelArgs &ka, int64_t batch, hipStream_t st);  // (mpcqp_quad4w.hip)

int launch_quad4(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    switch (ka.nx) {
    case 2: return launch_quad4_t<2>(ka, batch, st);
    case 3: return launch_quad4_t<3>(ka, batch, st);
    case 4: return launch_quad4_t<4>(ka, batch, st);
    default: return launch_quad4_wide(ka, batch, st);
    }
}
#endif

}  // namespace mpcqp
