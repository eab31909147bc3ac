// mpcqp_pair.hip -- gfx950 kernel for SMALL problems (n <= 16 variables, m <= 32
// inequality rows, nx in {3, 4}, float64): TWO PROBLEMS PER WAVEFRONT.
//
// Replaces the same reference code as mpcqp_w64.hip (qpmpc/mpc_qp.py:53-149 for the
// build, qpsolvers.solve_problem at qpmpc/solve_mpc.py:43 for the solve) for the fused
// build+solve of BASELINE configs 1, 2 and 4 (nx=3, nu=1, N=16 -> n=16, m=32).
//
// Why two per wavefront (round-2 counters, profiles/r02_*): with one problem per
// wavefront and four wavefronts per SIMD the launch is bound by VALU ISSUE -- about
// 4.1 k vector instructions per problem of which 1.5 k are f64 FMAs -- and every role
// of the 64 lanes executed the whole stream although each instruction was useful for
// one role only. Here a problem owns 32 lanes and each lane owns TWO 16-register rows,
// so one instruction stream serves two problems:
//   lane l of a half (l = 0..31):
//     RM  row M_l of M = G L^-T (constraint l), slack s_l                (all 32 lanes)
//     RT  l < 16: row l of T = N* (active-set slot l, multiplier, constraint id)
//         l >= 16: row l-16 of L^-T (identity pushed through the forward substitution; parked in LDS
//                  afterwards, so that u = L^-T y is one dot product at the end); in the active-set
//                  loop row l-16 of the projector H = I - M_A' T, and RM holds K_l = H M_l
//   during the build lanes 0..15 own a column of Psi and row l of P -> L, lane 16 the
//   free response Phi_k x0.
// The two halves take their own branches of the active-set iteration through
// predication: every per-problem scalar (selected row, slot, step length, status) is a
// per-lane value that is uniform inside a half. Half-wide reductions are four DPP row
// rotations plus one v_permlane16_swap; a value of one lane is fetched with ds_bpermute.
// 2048 wavefronts for the 4096 problems of config 2 -> two wavefronts per SIMD, a
// 256-register budget (no scratch), 19.6 KB of LDS per wavefront.
//
// Solver: the same dual active-set method (Goldfarb-Idnani 1983) with the explicit
// operator T = N* as mpcqp_w64.hip; see that file for the derivation. Here their
// operator H (the projector onto the null space of the active rows, in y-coordinates) is
// kept explicitly as well, and the constraint rows are kept projected, K_i = H M_i, so that
// ONE pass over the two register rows with the broadcast row M_p gives, for the selected p:
//   r_a = T_a.M_p (slot lanes) ; -z_k = H_k.M_p (lanes 16..31) ; -M_i.z = K_i.M_p ; |z|^2 = K_p.M_p
//   t = min(t1 = min lam_a/r_a, t2 = -s_p/d2) ; s_i -= t M_i.z ;
//   full step: T_a += (r_a/d2) z, T_new = -z/d2, H -= z z'/d2, K_i -= (M_i.z/d2) z ;
//   partial step: slot l leaves, T_a -= (T_a.T_l / T_l.T_l) T_l, H += T_l T_l'/T_l.T_l,
//   K_i += (M_i.T_l / T_l.T_l) T_l.
// The rank-one update is deferred to the top of the next trip (the one site that writes the
// register rows); trips that are plain full steps run in a small loop of their own.
//
// Broadcasts: wherever the value to broadcast sits in a lane known at compile time -- column j of the
// Cholesky factor, L[k][j] in the forward substitution, the operands of the Psi chain (lane e keeps
// element e of [A_k | C_k] for every step), v_b in the Gram accumulation, component k of the update
// vector in the fast loop -- it is a 64-bit DPP row broadcast folded into the FMA
// (v_fmac_f64_dpp ... row_newbcast:n), not an LDS round trip: both 16-lane rows of a half hold what
// the other needs (a v_permlane16_swap pair copies one row over the other). LDS is left with the
// images addressed by run-time indices: the G image, M (row p of the selected constraint), T by
// columns in the refinement, the rows of L^-T.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpcqp.h"
#include "mpcqp_internal.h"

#ifndef PAIR_EARLY_ARGS
#define PAIR_EARLY_ARGS 2
#endif
namespace mpcqp {

namespace pair {

constexpr int NV = 16;    // padded number of variables / slots
constexpr int HL = 32;    // lanes per problem
constexpr int MMAX = 32;  // constraints a half can hold
constexpr int LDM = 18;   // row stride of the L and M images (144 B: rows start in distinct 16-B slots)

// ------------------------------------------------------------ lane primitives
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned x)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
constexpr int ROR8 = 0x128, ROR4 = 0x124, ROR2 = 0x122, ROR1 = 0x121;  // rotate within a row of 16

// all-reduce (min) over the 32 lanes of each half
__device__ __forceinline__ unsigned half_min(unsigned v)
{
    v = min(v, dpp_u<ROR8>(v));
    v = min(v, dpp_u<ROR4>(v));
    v = min(v, dpp_u<ROR2>(v));
    v = min(v, dpp_u<ROR1>(v));
    // rows 1 and 3 of the first operand are exchanged with rows 0 and 2 of the second
    const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return min((unsigned)sw[0], (unsigned)sw[1]);
}
// all-reduce (or) over the 32 lanes of each half
__device__ __forceinline__ unsigned half_or(unsigned v)
{
    v |= dpp_u<ROR8>(v);
    v |= dpp_u<ROR4>(v);
    v |= dpp_u<ROR2>(v);
    v |= dpp_u<ROR1>(v);
    const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (unsigned)sw[0] | (unsigned)sw[1];
}
// value of lane `idx` (0..31, uniform inside a half) of the caller's own half
__device__ __forceinline__ int half_get(int x, int hb, int idx)
{
    return __builtin_amdgcn_ds_bpermute((hb + idx) << 2, x);
}
__device__ __forceinline__ double half_get(double x, int hb, int idx)
{
    const int a = (hb + idx) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
// true in every lane of a half iff `pred` holds in one of its lanes
__device__ __forceinline__ bool half_any(bool pred, int hb)
{
    const unsigned long long b = __ballot(pred);
    return ((unsigned)(b >> hb)) != 0u;
}

// order-preserving map of a double onto two unsigned words
__device__ __forceinline__ void ordered(double x, unsigned &hi, unsigned &lo)
{
    const unsigned h = (unsigned)__double2hiint(x), l = (unsigned)__double2loint(x);
    const bool neg = h & 0x80000000u;
    hi = neg ? ~h : (h | 0x80000000u);
    lo = neg ? ~l : l;
}

// ------------------------------------------------------------ 16-vectors in LDS
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
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }

// The value held by lane N of the caller's 16-lane row, in every lane of that row: a 64-bit DPP move
// (v_mov_b64_dpp row_newbcast:N) -- a register-to-register broadcast on the vector pipe, no LDS round trip.
template <int N> __device__ __forceinline__ double row_bcast(double x)
{
    return __builtin_amdgcn_mov_dpp(x, 0x150 + N, 0xf, 0xf, true);
}
// acc += (x of lane N of the caller's row) * m in ONE instruction (v_fmac_f64_dpp). The compiler does not fold the DPP
// move into the FMA, and it cannot see inside the asm: a register written by a VALU instruction needs two wait states
// before a DPP read, so every batch of these is preceded by dpp_ready(x) on its broadcast source.
template <int N> __device__ __forceinline__ void fmac_bcast(double &acc, double x, double m)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(N));
}
__device__ __forceinline__ void dpp_ready(double &x) { asm volatile("s_nop 1" : "+v"(x)); }
__device__ __forceinline__ void fmac_bcast_at(double &acc, double x, double m, int k)
{
    switch (k) {
    case 0: return fmac_bcast<0>(acc, x, m);
    case 1: return fmac_bcast<1>(acc, x, m);
    case 2: return fmac_bcast<2>(acc, x, m);
    case 3: return fmac_bcast<3>(acc, x, m);
    case 4: return fmac_bcast<4>(acc, x, m);
    case 5: return fmac_bcast<5>(acc, x, m);
    case 6: return fmac_bcast<6>(acc, x, m);
    case 7: return fmac_bcast<7>(acc, x, m);
    case 8: return fmac_bcast<8>(acc, x, m);
    case 9: return fmac_bcast<9>(acc, x, m);
    case 10: return fmac_bcast<10>(acc, x, m);
    case 11: return fmac_bcast<11>(acc, x, m);
    case 12: return fmac_bcast<12>(acc, x, m);
    case 13: return fmac_bcast<13>(acc, x, m);
    case 14: return fmac_bcast<14>(acc, x, m);
    default: return fmac_bcast<15>(acc, x, m);
    }
}
// The value held by the SECOND 16-lane row of the caller's half (lanes 16..31), lane for lane, in both rows.
__device__ __forceinline__ double from_high_row(double x)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)sh[1], (int)sl[1]);
}
// row_bcast with the lane given by a loop counter of a fully unrolled loop (the DPP lane select is an immediate:
// the switch folds once the counter is a constant)
__device__ __forceinline__ double row_bcast_at(double x, int k)
{
    switch (k) {
    case 0: return row_bcast<0>(x);
    case 1: return row_bcast<1>(x);
    case 2: return row_bcast<2>(x);
    case 3: return row_bcast<3>(x);
    case 4: return row_bcast<4>(x);
    case 5: return row_bcast<5>(x);
    case 6: return row_bcast<6>(x);
    case 7: return row_bcast<7>(x);
    case 8: return row_bcast<8>(x);
    case 9: return row_bcast<9>(x);
    case 10: return row_bcast<10>(x);
    case 11: return row_bcast<11>(x);
    case 12: return row_bcast<12>(x);
    case 13: return row_bcast<13>(x);
    case 14: return row_bcast<14>(x);
    default: return row_bcast<15>(x);
    }
}

// 1/x from the hardware estimate plus two Newton steps (operands are never subnormal
// or zero when the result is used)
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
// One wavefront per workgroup: its LDS operations complete in order, so only the
// COMPILER has to keep the order of an exchange (no s_barrier, no queue drain).
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct Lay {     // LDS carve of ONE problem in doubles (host-computed, passed by value)
    int off_X;   // build: G image 16 x GS | main: rows of L^-T 16 x 16, the T image 16 x 16 (refinement), 16 slot ids
    int off_Y;   // build: exchange + staged operands | L 16 x LDM + column buffers | main: M image m x LDM
    int off_hv;  // h_i by lane (32), then y0 (16)
    int off_v;   // kAv, rv, zv and their shadows (6 x 16)
    int off_stage, nA, nB, nC, nD;
    int per;     // doubles per problem (the second half's carve starts here)
};

constexpr double DEP = 1e-14;  // |z|^2 / |M_p|^2 below this: M_p depends on the active rows
// |z|^2 = M_p' H M_p is read off as K_p . M_p, whose rounding error is LINEAR in the error of the maintained row K_p
// (about 1e-16 |M_p|^2 times the growth of the updates) while |K_p|^2 is quadratic in it like the |z|^2 of a
// recomputed z. The fast loop therefore only trusts K_p . M_p well away from dependence; anything closer goes
// to the general trip, which forms |K_p|^2 (a stress run with inconsistent rows returned 'solved' for a few
// infeasible problems before: a dependent row slipped past the 1e-14 test on rounding noise).
constexpr double DEP_FAST = 1e-6;

}  // namespace pair

using namespace pair;

// MK > 0: compile-time number of inequality rows per step for the register-pipelined
// chain (terminal cost only, state constraints only); MK == 0: generic chain.
// MODEL: the problems share a factored model (mpcqp_factor_model: M, L^-T and the linear maps from the states to
// h and L^-1 q, gA = the model); build, factorisation and forward substitution are skipped (mpc_qp.py:129-163 usage).
// WARM: the launch carries a warm-start state (MpcqpSolveOpts.warm_state); the cold instantiations drop the repair
// machinery from the loop.
// WPB: wavefronts per workgroup, each with its own two problems and its own LDS carve (they share nothing: no barrier
// anywhere). A launch that fills the machine exactly once (two wavefronts per SIMD) is 3 % shorter with workgroups of two -- 25.6 against
// 26.4 us for 4096 problems; 3, 4 and 8 lose: 35.7 / 27.5 / 27.6 --, launches of several rounds are faster with single
// wavefronts (8192 problems: 47.2 against 53.3 us; 65,536: 310 against 320), see launch_pair_t.
// SEED: the instantiation that carries the seed steps (MPCQP_OPT_SEED_VIOLATED, MPCQP_WARM_ACTIVE_SET): compiled into the plain
// cold instantiation they cost its launches 2 % (198 instead of 192 registers, one more ballot per trip).
// ORD: the launch carries a pairing order (MpcqpSolveOpts.order): half-wavefront i takes problem order[i] instead of problem i. A
// wavefront runs max(trips of its two problems), 13.1 trips against 10.75 per problem on config 4: a launch of several rounds whose
// order puts problems of similar trip counts next to each other (mpcqp_order_by_count on last period's counts) is up to 12 % shorter.
// Its own instantiation: the plain one's prologue (hand-placed kernel-argument loads) is left as measured.
template <int NX, int MK, bool MODEL = false, bool WARM = false, int WPB = 1, bool SEED = false, bool ORD = false>
__global__ void __launch_bounds__(64 * WPB, 2)
    mpcqp_pair_kernel(const double *__restrict__ gA, const double *__restrict__ gB, const double *__restrict__ gC,
                      const double *__restrict__ gD, const double *__restrict__ ge, const double *__restrict__ gx0,
                      const double *__restrict__ ggoal, const double *__restrict__ gtgt, double *__restrict__ oU,
                      double *__restrict__ olam, int32_t *__restrict__ ostatus, int32_t *__restrict__ oiters,
                      const KernelArgs ka, const Lay L, const int64_t batch)
{
    using T = double;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#if PAIR_EARLY_ARGS
    {  // every kernel argument the operand addresses and the LDS carve need, requested in ONE batch of scalar loads at the top:
       // left to itself the compiler fetches them where they are first used -- eight dependent scalar-load round trips before
       // the first operand load is issued (25.6 -> 25.0 us per 4096 problems, tools/ab_unit.sh)
        const int64_t b0 = ka.A.batch_stride, b1 = ka.B.batch_stride, b2 = ka.C.batch_stride, b3 = ka.e.batch_stride,
                      b4 = ka.x0.batch_stride, b5 = ka.goal.batch_stride;
        const int64_t s0 = ka.A.step_stride, s1 = ka.B.step_stride, s2 = ka.C.step_stride, s3 = ka.e.step_stride;
        asm volatile("" ::"s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(ka.N), "s"(ka.nu),
                     "s"(ka.mk), "s"(ka.flags), "s"(ka.probe), "s"(gA), "s"(gB), "s"(gC), "s"(ge), "s"(gx0), "s"(ggoal));
#if PAIR_EARLY_ARGS >= 2
        asm volatile("" ::"s"(gD), "s"(gtgt), "s"(ka.D.step_stride), "s"(ka.n), "s"(ka.m), "s"(L.off_X), "s"(L.off_Y), "s"(L.off_hv),
                     "s"(L.off_v), "s"(L.off_stage), "s"(L.per), "s"(L.nA), "s"(L.nB), "s"(L.nC), "s"(L.nD), "s"(batch));
#endif
    }
#endif
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;  // wavefront of the workgroup (the wavefronts share nothing: no barrier anywhere)
    const int hb = lane & 32;   // first lane of this half
    const int hl = lane & 31;   // lane inside the half
    const int l15 = lane & 15;
    const bool low = hl < NV;
    int64_t prob = 2 * ((int64_t)blockIdx.x * WPB + wv) + (hb >> 5);
    const bool valid = prob < batch;  // an odd batch leaves the last half idle: it repeats the last problem, stores nothing
    prob = valid ? prob : batch - 1;
    if constexpr (ORD) {  // (an index outside the batch is clamped: a bad order costs wrong pairings, never an access out of bounds)
        const int64_t o = ka.order[prob];
        prob = o < 0 ? 0 : (o >= batch ? batch - 1 : o);
    }
    T *sm = (T *)smem_raw + (2 * wv + (hb ? 1 : 0)) * L.per;
    const int vofs = low ? hl : 3 * NV + l15;  // element of an exchange vector (shadow copy for lanes >= 16)
    const int n = ka.n, m = ka.m;
    const bool isc = hl < m;  // this lane owns a constraint
    const T INF = HUGE_VAL;
    T *Gimg = sm + L.off_X;
    const int GS = (m + 1) | 1;  // the G image is stored by COLUMN with an odd stride
    T *Ll = sm + L.off_Y, *Ml = sm + L.off_Y, *hv = sm + L.off_hv;
    T *y0v = hv + HL;
    T *kAv = sm + L.off_v, *rv = kAv + NV, *zv = rv + NV;

    // optional phase timestamps (developer probe, MpcqpSolveOpts.probe): long long[16] per problem
    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 16 : nullptr;
    auto tick = [&](int slot) {
        if (stamp && hl == 0 && valid) {
            stamp[slot] = (long long)__builtin_readcyclecounter();  // (shader clock: its base differs from CU to CU)
            if (slot == 0 || slot == 6) stamp[slot ? 13 : 12] = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz, one base
        }
    };
    tick(0);

    T Pr[NV];  // lane a < 16: row a of P, then of L
    // ---------------------------------------------------------------- build (mpc_qp.py:53-114)
    if constexpr (!MODEL) {
        constexpr int nx = NX;
        const int nu = ka.nu, N = ka.N, mk = MK > 0 ? MK : ka.mk;  // (a compile-time mk spares the prologue an integer division)
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
        T *hp = sm + L.off_Y + 4 * 32;  // hp[row] = C_k Phi_k x0 (m <= 32 entries; the 4 x 32 doubles before it are spare since the
                                        // Gram accumulation exchanges through DPP)
        auto al4 = [](int c) { return (c + 3) & ~3; };  // as make_lay: every staged array starts 16-byte aligned
        T *As = sm + L.off_stage, *Bs = As + al4(L.nA), *Cs = Bs + al4(L.nB), *Ds = Cs + al4(L.nC);
        // The problem's operands are staged in LDS by the 32 lanes of its half: every load of the
        // four arrays is issued before the first store, so the whole stage costs ONE HBM latency.
        // The lean instantiations (MK > 0) do not stage A and C at all: lane e of each 16-lane row keeps element e
        // (and e + 16) of [A_k | C_k] for every step k in registers, straight from HBM, and the chain fetches an
        // operand as a DPP row broadcast -- the chain used to be bound by the LDS return path (15 doubles broadcast
        // to every lane per step); now its only LDS traffic is the G rows it writes.
        constexpr bool LEAN = MK > 0;
        constexpr int NAe = NX * NX, NEe = NAe + MK * NX;  // elements of [A_k | C_k]
        // the per-lane scalars are requested first, so that their latency is the staging's
        const bool isx = (hl == NV), col = (hl < n);
        const int j = col ? (nu == 1 ? hl : hl / nu) : -1, ii = col ? hl - j * nu : 0;  // (nu == 1: no division before the loads)
        const T eval = isc ? ge[prob * ka.e.batch_stride + (hl / mk) * ka.e.step_stride + (hl % mk)] : INF;
        T v[NX], gref[NX];
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            v[s] = isx ? x0[s] : T(0);
            gref[s] = (isx && termQ) ? goal[s] : T(0);
        }
        T bcol[NX];  // this lane's column of B_j (enters the chain at step j)
        if constexpr (LEAN) {
#pragma unroll
            for (int r = 0; r < NX; ++r) bcol[r] = col ? B[j * sB + r * nu + ii] : T(0);
        }
        T opa[NV], opb[NV];
        if constexpr (LEAN) {
            // (lanes without an element and steps beyond the horizon load a valid address and are never read: no
            // select, so these registers are written by the loads only -- see dpp_ready)
            const int e0 = l15, e1 = l15 + 16;
            const bool ok0 = e0 < NEe, ok1 = e1 < NEe;
            const T *p0 = (e0 < NAe) ? A + e0 : Cm + (ok0 ? e0 - NAe : 0);
            const T *p1 = (e1 < NAe) ? A + e1 : Cm + (ok1 ? e1 - NAe : 0);
            const int s0 = (e0 < NAe) ? sA : sC, s1 = (e1 < NAe) ? sA : sC;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int kc = (k < N) ? k : N - 1;
                opa[k] = p0[kc * s0];
                opb[k] = (NEe > 16) ? p1[kc * s1] : T(0);
            }
        }
        {
            constexpr int CA = 8, CB2 = 2, CC = 4, CD = 4;  // 32-element chunks held in registers per array
            T ta[CA], tb[CB2], tc[CC], td[CD];
            const int nAs = LEAN ? 0 : L.nA, nCs = LEAN ? 0 : L.nC, nBs = LEAN ? 0 : L.nB;  // (lean: D is absent, B per lane above)
#pragma unroll
            for (int u = 0; u < CA; ++u) ta[u] = (u * HL + hl < nAs) ? A[u * HL + hl] : T(0);
#pragma unroll
            for (int u = 0; u < CB2; ++u) tb[u] = (u * HL + hl < nBs) ? B[u * HL + hl] : T(0);
#pragma unroll
            for (int u = 0; u < CC; ++u) tc[u] = (u * HL + hl < nCs) ? Cm[u * HL + hl] : T(0);
#pragma unroll
            for (int u = 0; u < CD; ++u) td[u] = (u * HL + hl < L.nD) ? Dm[u * HL + hl] : T(0);
#pragma unroll
            for (int u = 0; u < CA; ++u)
                if (u * HL + hl < nAs) As[u * HL + hl] = ta[u];
#pragma unroll
            for (int u = 0; u < CB2; ++u)
                if (u * HL + hl < nBs) Bs[u * HL + hl] = tb[u];
#pragma unroll
            for (int u = 0; u < CC; ++u)
                if (u * HL + hl < nCs) Cs[u * HL + hl] = tc[u];
#pragma unroll
            for (int u = 0; u < CD; ++u)
                if (u * HL + hl < L.nD) Ds[u * HL + hl] = td[u];
            // anything beyond the register chunks (long horizons of tiny systems never get here; kept for safety)
            for (int i = CA * HL + hl; i < nAs; i += HL) As[i] = A[i];
            for (int i = CB2 * HL + hl; i < nBs; i += HL) Bs[i] = B[i];
            for (int i = CC * HL + hl; i < nCs; i += HL) Cs[i] = Cm[i];
            for (int i = CD * HL + hl; i < L.nD; i += HL) Ds[i] = Dm[i];
        }
        tick(8);
        const T wu = (T)ka.wu;
#pragma unroll
        for (int b = 0; b < NV; ++b) Pr[b] = (hl == b) ? (col ? wu : T(1)) : T(0);
        T qa = T(0);
        wsync();
        tick(9);
        if constexpr (!LEAN) {
#pragma unroll
            for (int r = 0; r < NX; ++r) bcol[r] = col ? Bs[j * sB + r * nu + ii] : T(0);
        }

        // Gram accumulation of one block: Pr[b] += w v_a . v_b, qa += w resid . v_a;
        // ref[] is this lane's reference (non-zero only in lane 16).
        auto gram = [&](T w, bool useP, bool useQ, const T (&ref)[NX]) {
            if (!useP && !useQ) return;
            // v_b[s] of column lane b is a DPP row broadcast; the residual of lane 16 (first lane of the half's second
            // row) comes over with one row swap. Lanes 16..31 accumulate rows nobody reads (they take a copy of
            // lanes 0..15's rows before the factorisation).
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                T res = v[s] - ref[s];
                const T t = w * v[s];
                if (useQ) qa += t * row_bcast<0>(from_high_row(res));
                if (useP) {
                    dpp_ready(res);
#pragma unroll
                    for (int b = 0; b < NV; ++b) fmac_bcast_at(Pr[b], res, t, b);
                }
            }
        };
        // G rows of step k from v = Psi_k[:, lane] (lane 16: Phi_k x0); lanes 0..15 fill
        // column `lane` of the G image, lane 16 the C_k Phi_k x0 part of h (mpc_qp.py:62-78)
        T *gd = low ? (Gimg + hl * GS) : hp;
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
                gd[k * mk + i2] = acc;
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
            // Terminal cost only, C only, mk == MK (configs 1, 2, 4; the host checks). [G_k; Psi_{k+1}] = [C_k; A_k] Psi_k
            // with the operands broadcast from the lanes' registers. EVERY lane runs the chain (a DPP read from a lane
            // that a branch has switched off returns zero); lanes beyond 16 carry zeros and store nothing.
            // acc += (element idx of [A_k | C_k]) * x  (idx, k: constants once unrolled)
            auto mac = [&](T &acc, int idx, int k, T x) {
                if (idx < 16)
                    fmac_bcast_at(acc, opa[k], x, idx);
                else
                    fmac_bcast_at(acc, opb[k], x, idx - 16);
            };
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (k < N) {
                    T g[MK];
#pragma unroll
                    for (int i2 = 0; i2 < MK; ++i2) {
                        T acc = T(0);
#pragma unroll
                        for (int s2 = 0; s2 < NX; ++s2) mac(acc, NAe + i2 * NX + s2, k, v[s2]);
                        g[i2] = acc;
                    }
                    if (hl <= NV) {
#pragma unroll
                        for (int i2 = 0; i2 < MK; ++i2) gd[k * MK + i2] = g[i2];
                    }
                    const bool here = (j == k);
                    T w[NX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        T acc = T(0);
#pragma unroll
                        for (int s2 = 0; s2 < NX; ++s2) mac(acc, r * NX + s2, k, v[s2]);
                        w[r] = acc;
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) v[r] = here ? bcol[r] : w[r];
                }
            }
        } else if (!stageP && !stageQ) {
            if (hl <= NV) {
                for (int k = 0; k < N; ++k) {
                    g_rows(k);
                    advance(k);
                }
            }
        } else {
            for (int k = 0; k < N; ++k) {
                if (hl <= NV) g_rows(k);
                if (k >= 1) {
                    T tref[NX];
#pragma unroll
                    for (int s = 0; s < NX; ++s) tref[s] = (isx && stageQ) ? tgt[k * nx + s] : T(0);
                    gram((T)ka.wx, stageP, stageQ, tref);
                }
                if (hl <= NV) advance(k);
            }
        }
        tick(10);
        gram((T)ka.wt, termP, termQ, gref);  // v = Psi_N
        wsync();
        tick(11);
        if (low) Gimg[hl * GS + m] = col ? qa : T(0);  // the q row
        wsync();
        hv[hl] = (isc && L.nC) ? eval - hp[hl] : eval;  // h_i = e_i - C_k Phi_k x0
        wsync();
    }

    tick(1);
    // ------------------------------------------------------------ factorise
    // Right-looking Cholesky. Lanes 16..31 of each half take a copy of the rows of P first, so that BOTH 16-lane
    // rows of the half hold the factor: every broadcast of the factorisation and of the forward substitution
    // is then a DPP row broadcast (lane k's entry of column j to its whole row) instead of an LDS round trip --
    // these two phases used to be bound by the LDS return path (about 150 sixteen-byte broadcast reads per wavefront).
    bool notpd = false;
    T myinv = T(1);  // lanes j and 16 + j keep 1 / L_jj
    if constexpr (!MODEL) {
        if (low) st16(Ll + hl * LDM, Pr);
        wsync();
        ld16(Pr, Ll + l15 * LDM);
        wsync();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const T pij = Pr[j];                    // P[i][j] of this lane's row, before scaling
            const T piv = row_bcast_at(pij, j);     // P[j][j]
            if (!(piv > T(0))) notpd = true;
            const T rinv = rsqrt(piv);
            const T nt2 = -(pij * rinv * rinv);     // -P[i][j] / piv
            {
                T src = pij;
                dpp_ready(src);
#pragma unroll
                for (int k = j + 1; k < NV; ++k) fmac_bcast_at(Pr[k], src, nt2, k);  // P[k][j] from lane k
            }
            Pr[j] = pij * rinv;                     // L[i][j] (lane j: sqrt(piv))
            if (l15 == j) myinv = rinv;
            pin(Pr[j]);
        }
        wsync();  // (keeps the next phase's LDS loads out of the factorisation: register pressure)
        __builtin_amdgcn_sched_barrier(0);
    }
    tick(2);

    // the stored warm-start state is requested now and consumed after the forward substitution
    double *wstate = ((WARM || SEED) && ka.warm_state) ? (double *)ka.warm_state + prob * (int64_t)kPairWarmDoubles : nullptr;
    const bool wload = WARM && wstate && ka.warm_start == MPCQP_WARM_OPERATOR && !notpd;
    // MPCQP_WARM_ACTIVE_SET: only the stored row ids are read (the cold instantiation serves it: see the seeded start)
    const bool wseed = SEED && !WARM && wstate && ka.warm_start == MPCQP_WARM_ACTIVE_SET;
    int sid = -1;
    if (wseed && low) sid = reinterpret_cast<const int *>(wstate + NV * NV)[hl];
    int wid = -1;
    T wrow[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) wrow[k] = T(0);
    if (wload && low) {
        wid = reinterpret_cast<const int *>(wstate + NV * NV)[hl];
        const double2 *src = reinterpret_cast<const double2 *>(wstate + hl * NV);
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
            const double2 t = src[i];
            wrow[2 * i] = t.x;
            wrow[2 * i + 1] = t.y;
        }
    }
    // ------------------------------------------------------------ rows, forward substitution
    T RM[NV], RT[NV];
    if constexpr (MODEL) {
        // rows of M and of L^-T straight from the model (4 KB, read by every wavefront of the launch: L1/L2),
        // h = e - Hx x0 and w = L^-1 q = Wx x0 - Wg goal - Wt targets from the model's linear maps
        const T *model = gA;
        const ModelLayout ml = make_model_layout(ka.nx, ka.N, n, m);
        const int nxr = ka.nx, nT = ka.N * ka.nx;
        const T *x0 = gx0 + prob * ka.x0.batch_stride;
        const T *goal = ggoal ? ggoal + prob * ka.goal.batch_stride : nullptr;
        const T *tgt = gtgt ? gtgt + prob * ka.targets.batch_stride : nullptr;
        notpd = model[ml.total] != T(0);
        {
            T rm[NV], rt[NV];
            ld16(rm, model + ml.off_M + (size_t)(isc ? hl : 0) * NV);
            ld16(rt, model + ml.off_LinvT + (size_t)(low ? 0 : hl - NV) * NV);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                RM[k] = isc ? rm[k] : T(0);
                RT[k] = low ? T(0) : rt[k];
            }
        }
        T hh = INF;
        if (isc) {
            // the bounds come with the model, or per problem (mpcqp_solve_model_bounds_batch: matrices shared, e moving)
            const int mkr = ka.mk, kq = mkr == 2 ? hl >> 1 : hl / mkr;  // (two rows per step: no integer division in the prologue)
            hh = ge ? ge[prob * ka.e.batch_stride + kq * ka.e.step_stride + (hl - kq * mkr)] : model[ml.off_e + hl];
            for (int c = 0; c < nxr; ++c) hh -= model[ml.off_Hx + (size_t)hl * nxr + c] * x0[c];
        }
        hv[hl] = hh;
        if (low) {
            T wk = T(0);
            for (int c = 0; c < nxr; ++c) wk += model[ml.off_Wx + (size_t)hl * nxr + c] * x0[c];
            if ((ka.flags & MPCQP_Q_TERMINAL) && goal)
                for (int c = 0; c < nxr; ++c) wk -= model[ml.off_Wg + (size_t)hl * nxr + c] * goal[c];
            if ((ka.flags & MPCQP_Q_STAGE) && tgt)
                for (int j2 = 0; j2 < nT; ++j2) wk -= model[ml.off_Wt + (size_t)hl * nT + j2] * tgt[j2];
            y0v[hl] = wk;
        }
        wsync();
    } else {
        // Rows fetched only now: constraint lanes their row of G; lane 0 takes q in RT, lanes 16..31
        // the identity (-> rows of L^-T), the other slot lanes zero.
#pragma unroll
        for (int k = 0; k < NV; ++k) RM[k] = isc ? Gimg[k * GS + hl] : T(0);
        if (hl == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) RT[k] = Gimg[k * GS + m];
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) RT[k] = (hl == NV + k) ? T(1) : T(0);
        }
        // [RM; RT] <- [RM; RT] L^-T, right-looking: y_j = x_j / L_jj, then x_k -= y_j L[k][j] for k > j (the updates of
        // a step are independent of each other); L[k][j] is lane k's entry of column j, a DPP row broadcast.
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const T ij = row_bcast_at(myinv, j);
            RM[j] *= ij;
            RT[j] *= ij;
            // (the step's broadcasts stay behind its scaling, and both rows finish the step before the next one starts:
            // otherwise the scheduler runs the two chains apart and keeps all 120 broadcast values alive in between)
            T colj = Pr[j];
            asm volatile("s_nop 1" : "+v"(colj) : "v"(RM[j]), "v"(RT[j]));  // (also the DPP wait states, see fmac_bcast)
            const T nm = -RM[j], nt = -RT[j];
#pragma unroll
            for (int k = j + 1; k < NV; ++k) {
                fmac_bcast_at(RM[k], colj, nm, k);
                fmac_bcast_at(RT[k], colj, nt, k);
            }
#pragma unroll
            for (int k = j + 1; k < NV; ++k) asm volatile("" : "+v"(RM[k]), "+v"(RT[k]));
        }
        wsync();  // the M image below reuses the L image
    }
    tick(3);
    int status = MPCQP_MAX_ITER, iters = 0;
    T xsol = T(0), lam_out = T(0);
    bool done = notpd;  // this half has left the active-set loop
    bool finished = notpd;  // ... and needs no refinement any more (failed, or accepted)
    if (notpd) status = MPCQP_NOT_PD;

    // Main-phase carve of the X region: the rows of L^-T (only read when a point is accepted), the T image
    // (T by columns, refinement only) and the 16 constraint ids of the slots.
    T *LTimg = sm + L.off_X, *Timg = LTimg + NV * NV;
    int *actv = reinterpret_cast<int *>(Timg + NV * NV);
    // colp(a)[0] = element l15 of the row of M held by slot a (actv published by the caller)
    auto ma_dot = [&](const T (&cf)[NV]) {
        int aa[NV];
        const int4 *ap = reinterpret_cast<const int4 *>(actv);
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            const int4 t = ap[q];
            aa[4 * q] = t.x;
            aa[4 * q + 1] = t.y;
            aa[4 * q + 2] = t.z;
            aa[4 * q + 3] = t.w;
        }
        T a0 = T(0), a1 = T(0);
#pragma unroll
        for (int a = 0; a < NV; a += 2) {
            a0 += cf[a] * Ml[aa[a] * LDM + l15];
            a1 += cf[a + 1] * Ml[aa[a + 1] * LDM + l15];
        }
        return a0 + a1;  // (M_A' cf)_l15 ; empty slots carry a zero coefficient
    };

    if (!MODEL && hl == 0) st16(y0v, RT);  // w = L^-1 q ; y0 = -w
    if (isc) st16(Ml + hl * LDM, RM);  // image of M: row-p broadcasts, and the only copy of M_i once RM holds H M_i
    if (!low) st16(LTimg + l15 * NV, RT);  // the rows of L^-T leave the registers
#pragma unroll
    for (int k = 0; k < NV; ++k) RT[k] = (!low && l15 == k) ? T(1) : T(0);  // T = N* starts empty, H = I
    zv[l15] = T(0);  // the pending rank-one update's vector
    wsync();
    const T hval = hv[hl];
    T s;
    {
        T y0[NV];
        ld16(y0, y0v);
        s = hval + dot16(RM, y0);  // h - M y0  (y0 = -L^-1 q)
    }
    s = isc ? s : INF;
    // Selection rule (the classic Goldfarb-Idnani one): among the rows violated beyond the
    // tolerance, the one FARTHEST from its hyperplane in the P^-1 metric, s_i / |M_i|.
    T invn;
    {
        const T nn = dot16(RM, RM);
        invn = (nn > T(0)) ? rsqrt(nn) : T(1);
    }
    const bool selectable = isc && (hval < T(1e29));
    const T tol = (T)ka.tol;
    const T tolh = tol + tol * fabs(hval);  // row i is violated when s_i < -tol (1 + |h_i|)
    const int max_iter = ka.max_iter;

    T lam = T(0);      // slot lanes: multiplier of the slot
    int myact = 0;     // slot lanes: constraint held by the slot
    bool occ = false;  // slot lanes: slot occupied
    int pos = -1;      // constraint lanes: slot of this constraint, or -1
    // half-uniform state
    int nq = 0, p = 0, ldrop = 0;
    unsigned mask = 0;  // occupied slots
    bool needp = true, dropping = false;
    T up = T(0);
    T cT = T(0), cK = T(0);  // pending rank-one update RT += cT v, RM += cK v (applied at the top of the next trip:
                             // the ONE site that writes the two register rows)
    bool pdrop = false;      // ... whose vector v is T_l in kAv (a slot leaves) instead of z in zv
    bool pstore = false;     // slot lanes: this lane's row of T goes to kAv once the pending update is applied
    bool hkstale = false;    // this half runs on a stored operator while H, K are still those of the empty set
    const T s0 = s;      // slacks at the unconstrained minimiser (cold restart)
    bool warm = false;   // this half started from a stored active set
    bool wfix = false;   // ... and is still repairing it (multipliers that turned negative leave one by one)
    bool wdrop = false;  // the drop in flight belongs to that repair: back to the multiplier solve afterwards
    bool negl = false;   // slot lanes: this slot's multiplier came out negative, the slot leaves during the repair
    int fails = 0;       // verifications that found a violated row (per half)
    bool recold = false; // a warm-started half failed inside the loop: restart it from the empty set
    // back to the empty active set (after a warm start that did not lead to a certified point)
    auto cold_reset = [&]() {
        warm = wfix = wdrop = recold = negl = false;
        fails = 0;
        iters = 0;
        T mi[NV];
        ld16(mi, Ml + (isc ? hl : 0) * LDM);
        int kk = low ? -1 : l15;
        asm volatile("" : "+v"(kk));  // (recomputed here: otherwise the 16 unit-vector entries stay live from the start)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            RT[k] = (kk == k) ? T(1) : T(0);
            RM[k] = isc ? mi[k] : T(0);
        }
        cT = cK = T(0);
        pdrop = pstore = false;
        hkstale = false;
        lam = T(0);
        occ = false;
        pos = -1;
        mask = 0;
        nq = 0;
        s = s0;
        status = MPCQP_MAX_ITER;
        done = false;
        needp = true;
        dropping = false;
    };

    // ------------------------------------------------------------ warm start (MpcqpSolveOpts.warm_state)
    // The stored operator T = N* depends on M only (not on q, h), so the previous period's T and
    // active set are a valid operator for this period as long as the matrices did not change. The
    // multipliers are recomputed from scratch (lam = -T T' s0_A); while one is negative its row
    // leaves and they are recomputed; then (y, A) is an S-pair in Goldfarb-Idnani's sense and the
    // usual iterations go on from it.
    if (wload) {
        int *slotof = reinterpret_cast<int *>(kAv);  // 32 ints: slot claiming each constraint
        const int a = wid;
        const bool okrow = low && a >= 0 && a < m && hv[(a >= 0 && a < m) ? a : 0] < T(1e29);
        slotof[hl] = -1;
        wsync();
        if (okrow) slotof[a] = hl;  // two slots naming one row: one of them wins
        wsync();
        const bool win = okrow && slotof[a] == hl;
        const int mypos = slotof[hl];
        const unsigned wmask = (unsigned)(__ballot(win) >> hb) & 0xffffu;
        const int cnt = __builtin_popcount(wmask);
        wsync();
        bool finite = true;
#pragma unroll
        for (int k = 0; k < NV; ++k) finite = finite && (fabs(wrow[k]) < T(1e150));  // false for NaN / inf too
        const bool sane = !half_any(win && !finite, hb);
        if (cnt > 0 && cnt <= n && sane) {  // (half-uniform)
            if (low) {
#pragma unroll
                for (int k = 0; k < NV; ++k) RT[k] = win ? wrow[k] : T(0);
            }
            occ = win;
            myact = win ? a : 0;
            pos = isc ? mypos : -1;
            mask = wmask;
            nq = cnt;
            warm = wfix = true;
            hkstale = true;
            done = true;  // straight to the multiplier solve
            needp = false;
        }
        wsync();
    }
    // The projector H = I - M_A' T (rows in lanes 16..31) and the projected rows K_i = H M_i that go with a stored
    // operator, for the halves in `take`. Built only when such a half really enters the active-set loop: a stored
    // state that is accepted as it stands (the common case) never needs them.
    auto rebuild_hk = [&](const bool take) {
        if (low) {
            st16(Timg + hl * NV, RT);  // T by rows
            actv[hl] = occ ? myact : 0;
        }
        wsync();
        T hrow[NV];
        int kk = l15;
        asm volatile("" : "+v"(kk));
#pragma unroll
        for (int k = 0; k < NV; ++k) hrow[k] = (kk == k) ? T(1) : T(0);
        {
#pragma unroll 2
            for (int a = 0; a < NV; ++a) {  // (rolled: this block runs once per warm start, registers are scarce here)
                const T ma = Ml[actv[a] * LDM + l15];  // M_A[a][l15] (the row of T is zero for an empty slot)
                T ta[NV];
                ld16(ta, Timg + a * NV);
#pragma unroll
                for (int k = 0; k < NV; ++k) hrow[k] -= ma * ta[k];
            }
        }
        wsync();
        if (!low) st16(Timg + l15 * NV, hrow);  // H by rows (symmetric)
        wsync();
        {
            T kr[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) kr[k] = T(0);
            const T *mrow = Ml + (isc ? hl : 0) * LDM;  // M_i again (RM still holds it, but indexing it needs unrolling)
#pragma unroll 2
            for (int jj = 0; jj < NV; ++jj) {
                const T mij = isc ? mrow[jj] : T(0);
                T hj[NV];
                ld16(hj, Timg + jj * NV);
#pragma unroll
                for (int k = 0; k < NV; ++k) kr[k] += mij * hj[k];
            }
            if (take) {
#pragma unroll
                for (int k = 0; k < NV; ++k) RM[k] = kr[k];
            }
        }
        if (take && !low) {
#pragma unroll
            for (int k = 0; k < NV; ++k) RT[k] = hrow[k];
        }
        wsync();
    };
    // ------------------------------------------------------------ seeded start
    // The rows violated at the unconstrained minimiser predict the final active set well when the bounds are simple
    // (BASELINE config 2: 98 % of them end up active, 83 % of the final set is among them), and what a trip of the
    // loop below costs is mostly NOT arithmetic: the selection's reduction, the row of M whose address depends on it, the
    // ratio test and the flags form one dependent chain of ~1.5 k cycles around 64 FMAs. A SEED STEP adds the next row of
    // that list -- lowest index first, only while it is still violated -- with everything the selection needed known one
    // step ahead and without a ratio test: the same rank-one updates of T, H, K, the same move of the implied primal
    // point, multipliers allowed to turn negative. What it reaches is the minimiser on the seeded rows' hyperplanes; rows
    // whose multiplier came out negative then leave one by one (`negfix`, first thing in the loop below: the ordinary drop
    // pass, extended by the move a non-zero multiplier implies), which restores an S-pair in Goldfarb-Idnani's sense, and
    // the ordinary iterations finish from there. Same minimiser (strict convexity). Measured (tools/ab_seed.py): a seed step
    // costs 1.04 k cycles against 1.56 k for a trip, but the rows it misses cost full trips afterwards and the slowest wavefront of a
    // one-round launch gets no shorter -- 25.7 against 24.4 us per 4096 config-2 problems --, so seeding from the violated rows is
    // opt-in (MPCQP_OPT_SEED_VIOLATED); the seed steps are what MPCQP_WARM_ACTIVE_SET -- last period's active rows, moved with the
    // horizon -- starts from: those rows enter whether or not they are violated yet.
    bool negfix = false;  // this half holds negative multipliers left by the seed steps
    if (SEED && ((ka.opt_flags & MPCQP_OPT_SEED_VIOLATED) || wseed)) {
        // rows that enter whether or not they are violated at the moment: last period's active set, moved with the horizon
        unsigned force = 0u;
        if (wseed) {
            const int a = sid - ka.warm_shift;
            const bool okrow = low && sid >= 0 && a >= 0 && a < m && hv[(a >= 0 && a < m) ? a : 0] < T(1e29);
            force = half_or(okrow ? (1u << a) : 0u);
            force = done ? 0u : force;
        }
        unsigned rem = force;  // (half-uniform)
        if (ka.opt_flags & MPCQP_OPT_SEED_VIOLATED) rem |= (unsigned)(__ballot(!done & !warm & selectable & (s < -tolh)) >> hb);
        rem = (__builtin_popcount(rem) > n) ? 0u : rem;  // more rows than slots: left to the ordinary iterations
        force &= rem;
        const bool forced = (force >> hl) & 1u;  // this lane's row is one of them
        if (__ballot(rem != 0u) != 0ull) {
            // Software pipeline: while step t runs, the seed of step t + 1 is chosen -- from the slacks as they are BEFORE
            // step t, one step stale -- and its row of M is on its way from LDS; lane p itself re-checks, with the slacks of
            // the moment, that its row is still violated (a row that step t repaired makes step t + 1 an empty one).
            T zn = T(0), cTn = T(0), cKn = T(0);
            int ps = 0;
            bool st = false;
            // next seed of this half: lowest remaining row that is violated now
            auto next_seed = [&](T (&mpn)[NV]) {
                const unsigned vnow = (unsigned)(__ballot(selectable & (pos < 0) & (s < -tolh)) >> hb);
                const unsigned cand = rem & (vnow | force);
                st = (cand != 0u);
                ps = st ? (int)__builtin_ctz(cand) : 0;
                rem = st ? (cand & ~(1u << ps)) : 0u;
                ld16(mpn, Ml + ps * LDM);
            };
            auto seed_step = [&](const T (&mp)[NV], T (&mpn)[NV]) {
                const int pc = ps;        // this step's row (chosen one step ago)
                const bool sc = st & (__builtin_popcount(mask) < n);
                next_seed(mpn);
                dpp_ready(zn);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    fmac_bcast_at(RT[k], zn, cTn, k);
                    fmac_bcast_at(RM[k], zn, cKn, k);
                }
                const T rd = dot16(RT, mp), kd = dot16(RM, mp);
                // |z|^2 = K_p . M_p, trusted only well away from dependence (see DEP_FAST); lane pc decides for its row,
                // which must still be violated
                const T kx = ((kd * invn * invn > T(DEP_FAST)) & ((s < -tolh) | forced)) ? kd : T(0);
                const T d2 = half_get(kx, hb, pc);
                const T sp = half_get(s, hb, pc);
                const bool go = sc & (d2 > T(0));
                const T inv = go ? fast_rcp(d2) : T(0);
                const T tt = -sp * inv;  // (zero for a half that does not step)
                zn = from_high_row(rd);  // -z_k in lanes k and 16 + k
                const int sl = (int)__builtin_ctz(~mask);
                const bool isnew = go & (hl == sl);
                // T_a += (r_a/d2) z, T_new = -z/d2, H_k -= (z_k/d2) z, K_i -= (M_i.z/d2) z, as multiples of -z (an empty
                // slot's row of T is zero, so is its r_a; rows of lanes without a constraint are zero)
                cTn = (hl == sl) ? inv : -(rd * inv);
                cKn = -(kd * inv);
                s = fma(tt, kd, s);  // s_i -= t M_i . z ; an active row's K_i vanishes (their slacks are reset below)
                const T ln = low ? fma(-tt, rd, lam) : lam;  // (no ratio test, no clamp)
                lam = isnew ? tt : ln;
                myact = isnew ? pc : myact;
                occ = occ | isnew;
                pos = (go & (hl == pc)) ? sl : pos;
                mask |= go ? (1u << sl) : 0u;
            };
            T mpa[NV], mpb[NV];
            next_seed(mpa);
            for (;;) {
                if (__ballot(st) == 0ull) break;
                seed_step(mpa, mpb);
                if (__ballot(st) == 0ull) break;
                seed_step(mpb, mpa);
            }
            // the last step's update
            dpp_ready(zn);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                fmac_bcast_at(RT[k], zn, cTn, k);
                fmac_bcast_at(RM[k], zn, cKn, k);
            }
            nq = __builtin_popcount(mask);
            iters += nq;
            s = (pos >= 0) ? T(0) : s;
            negfix = half_any(occ & (lam < T(0)), hb);
        }
    }
    tick(4);
    for (;;) {
        // ===================================================== active-set loop
        for (;;) {
            // ---- the previous trip's rank-one update:  T_a += (r_a/d2) z, T_new = -z/d2 ; H -= z z'/d2 ;
            //      K_i -= (M_i.z/d2) z   (independent of the selection below, which only reads the slacks)
            {
                T vv[NV];
                ld16(vv, pdrop ? kAv : zv);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    RT[k] += cT * vv[k];
                    RM[k] += cK * vv[k];
                }
            }
            if (WARM && __ballot(pstore) != 0ull) {  // warm-start repair chain: the next slot to leave publishes its row
                wsync();
                if (pstore) st16(kAv, RT);
                pstore = false;
                wsync();
            }
            cT = cK = T(0);
            pdrop = false;
            if (WARM && __ballot(hkstale & !done) != 0ull) {
                const bool take = hkstale & !done;
                rebuild_hk(take);
                hkstale = hkstale & !take;
            }
            // ---- FAST LOOP. While every half still in the loop is about to take a new constraint and the step
            //      turns out to be a FULL one (no multiplier blocks, nothing leaves, no limit reached), a trip needs
            //      none of the general machinery: one ballot per trip checks that, anything else leaves this loop
            //      and the same trip is redone by the general code below.
            if (__ballot(!done & (!needp | dropping | (SEED & negfix))) == 0ull) {
                // Inside this loop the update vector never goes through LDS: lane 16 + k of a half produces -z_k, one
                // v_permlane16_swap pair copies the high row over the low one, and the next trip's update reads
                // component k as a DPP row broadcast. cTn, cKn are the coefficients of -z.
                T zn = T(0), cTn = T(0), cKn = T(0);
                for (;;) {
                    dpp_ready(zn);
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        fmac_bcast_at(RT[k], zn, cTn, k);
                        fmac_bcast_at(RM[k], zn, cKn, k);
                    }
                    cTn = cKn = T(0);
                    {
                        // a violated row's scaled slack is negative: the order of the magnitudes is the order of
                        // the high words, so the most violated row has the smallest complement
                        const unsigned hi = ~(unsigned)__double2hiint(s * invn);
                        const bool viol = !done & selectable & (pos < 0) & (s < -tolh);
                        const unsigned key = viol ? ((hi & ~31u) | (unsigned)hl) : 0xffffffffu;
                        const unsigned mkey = half_min(key);
                        const bool none = !done & (mkey == 0xffffffffu);
                        done = done | none;
                        status = none ? (int)MPCQP_SOLVED : status;
                        p = done ? p : (int)(mkey & 31u);
                    }
                    up = T(0);
                    if (__ballot(!done) == 0ull) break;
                    T rd, kd;
                    {
                        T mp[NV];
                        ld16(mp, Ml + p * LDM);
                        rd = dot16(RT, mp);
                        kd = dot16(RM, mp);
                    }
                    const bool st = !done;
                    const T d2 = half_get(kd, hb, p);
                    const T sp = half_get(s, hb, p);
                    const T ip = half_get(invn, hb, p);
                    const bool can_move = (nq < n) & (d2 * ip * ip > DEP_FAST);
                    const T inv = can_move ? fast_rcp(d2) : T(0);
                    const T t2 = can_move ? -sp * inv : INF;
                    const int sl = (int)__builtin_ctz(~mask);
                    const T r0 = occ ? rd : T(0);
                    const bool blk = (r0 > T(0)) & (lam < t2 * r0);
                    const bool odd = st & (!can_move | (iters >= max_iter) | blk);
                    if (__ballot(odd) != 0ull) {
                        needp = done;  // the halves still in the loop hold a selected row and have not stepped
                        break;
                    }
                    zn = from_high_row(rd);  // -z_k in lanes k and 16 + k
                    const bool isnew = st & (hl == sl), isp = st & (hl == p);
                    iters += st ? 1 : 0;
                    // T_a += (r_a/d2) z, T_new = -z/d2, H_k -= (z_k/d2) z, K_i -= (M_i.z/d2) z, as multiples of -z
                    cTn = st ? ((hl == sl) ? inv : -((low ? r0 : rd) * inv)) : T(0);
                    cKn = (st & isc) ? -(kd * inv) : T(0);
                    const T tt = st ? t2 : T(0);
                    const T sn = (pos >= 0) ? T(0) : s + t2 * kd;  // s_i -= t M_i . z
                    s = (st & isc) ? sn : s;
                    T ln = lam - tt * r0;
                    ln = (occ & (ln < T(0))) ? T(0) : ln;
                    lam = isnew ? tt : ln;
                    myact = isnew ? p : myact;
                    occ = occ | isnew;
                    pos = isp ? sl : pos;
                    s = isp ? T(0) : s;
                    mask |= st ? (1u << sl) : 0u;
                    nq += st ? 1 : 0;
                }
                // (both exits leave no update pending: the coefficients are cleared right after each update)
            }
            // ---- seeded start: a slot whose multiplier came out negative leaves -- the most negative one first, one per
            //      trip -- before anything else is selected (rare: ~0.2 rows per config-2 problem)
            bool ndrop = false;  // this trip's drop pass carries a non-zero multiplier
            if (SEED && __ballot(negfix & !done & !dropping) != 0ull) {
                const bool rep = negfix & !done & !dropping;
                unsigned hi, lo;
                ordered(lam, hi, lo);
                const bool ng = rep & occ & (lam < T(0));
                const unsigned mkey = half_min(ng ? ((hi & ~31u) | (unsigned)hl) : 0xffffffffu);
                const bool any = rep & (mkey != 0xffffffffu);
                const int ln = (int)(mkey & 31u);
                const int cl = half_get(myact, hb, ln);
                wsync();
                if (any && hl == ln) st16(kAv, RT);  // (the pending update was applied at the top of this trip)
                if (any && hl == cl) pos = -1;
                wsync();
                ldrop = any ? ln : ldrop;
                dropping = dropping | any;
                ndrop = any;
                negfix = negfix & !(rep & !any);  // none left: (y, A) is an S-pair, the ordinary iterations take over
            }
            // ---- selection, for the halves that start a new constraint (straight-line selects: no divergent branches)
            {
                const unsigned hi = ~(unsigned)__double2hiint(s * invn);  // (negative for every candidate: see the fast loop)
                const bool want = needp & !done & (SEED ? !dropping : true);
                const bool viol = want & selectable & (pos < 0) & (s < -tolh);
                const unsigned key = viol ? ((hi & ~31u) | (unsigned)hl) : 0xffffffffu;
                const unsigned mkey = half_min(key);
                const bool none = want & (mkey == 0xffffffffu);
                const bool got = want & !none;
                done = done | none;
                status = none ? (int)MPCQP_SOLVED : status;
                p = got ? (int)(mkey & 31u) : p;
                up = got ? T(0) : up;
                needp = needp & !got;
            }
            if (__ballot(!done) == 0ull) break;
            bool stepping = !done && !dropping;
            const bool drp = !done && dropping;
            // ---- one pass over both register rows with row p of M (a broadcast read inside the half):
            //      slot lanes r_a = T_a . M_p ; lanes 16..31 -z_k = H_k . M_p ; constraint lanes -M_i . z = K_i . M_p
            T rd, kd;
            {
                T mp[NV];
                ld16(mp, Ml + p * LDM);
                rd = dot16(RT, mp);
                kd = dot16(RM, mp);
            }
            // z by the H lanes (the slot lanes write the shadow); a step cancelled below leaves its coefficients zero
            zv[low ? 3 * NV + l15 : l15] = stepping ? -rd : T(0);
            const T mz = -kd;
            // ---- step length
            const T d2 = half_get(dot16(RM, RM), hb, p);  // |z|^2 = |H M_p|^2 = |K_p|^2 (robust near dependence, see DEP_FAST)
            const T sp = half_get(s, hb, p);
            const T ip = half_get(invn, hb, p);
            const bool can_move = (nq < n) & (d2 * ip * ip > DEP) & (d2 > T(0));
            const T inv = can_move ? fast_rcp(d2) : T(0);
            const T t2 = can_move ? -sp * inv : INF;
            const int sl = (int)__builtin_ctz(~mask);  // lowest free slot
            // ---- general tail
            if (stepping && iters >= max_iter) {
                done = true;
                finished = !warm;
                recold = warm;
                status = MPCQP_MAX_ITER;
                stepping = false;
            }
            iters += stepping ? 1 : 0;
            const T r = (occ && stepping) ? rd : T(0);
            const bool cand = occ && (r > T(0));
            // a blocking multiplier exists iff lam_a / r_a < t2 for some slot
            const bool blocked = half_any(stepping && cand && (lam < t2 * r), hb);
            T t1 = INF;
            int l = 0;
            if (__ballot(blocked) != 0ull) {  // rare: ratio test on the multipliers
                const T ratio = cand ? lam * fast_rcp(r) : INF;
                unsigned hi, lo;
                ordered(ratio, hi, lo);
                hi = cand ? hi : 0xffffffffu;
                const unsigned mhi = half_min(hi);
                const unsigned k2 = (cand && hi == mhi) ? ((lo & ~31u) | (unsigned)hl) : 0xffffffffu;
                const unsigned ml = half_min(k2);
                l = (int)(ml & 31u);
                const T tl1 = half_get(ratio, hb, l);
                t1 = (mhi != 0xffffffffu) ? tl1 : INF;
            }
            T t = t1 < t2 ? t1 : t2;
            if (stepping && !(t < INF)) {  // no step possible: the constraints are inconsistent
                done = true;
                finished = !warm;  // after a warm start the verdict is only trusted from a cold operator
                recold = warm;
                status = MPCQP_INFEASIBLE;
                stepping = false;
            }
            t = stepping ? t : T(0);
            const bool full = stepping && (t2 <= t1);
            // ---- coefficients of the (deferred) update with z: slot sl takes -z/d2, the occupied slots r_a/d2,
            //      H row k -z_k/d2 = (H_k . M_p)/d2, K row i -(M_i . z)/d2 = (K_i . M_p)/d2
            if (full) {
                cT = (hl == sl) ? -inv : ((low ? r : rd) * inv);
                cK = isc ? kd * inv : T(0);
            }
            if (__ballot(drp) != 0ull) {
                // slot ldrop leaves (its row T_l is in kAv). With W = T T' implicit,
                // T_a -= (T_a . T_l / T_l . T_l) T_l (row l becomes exactly zero); the null space of the active
                // rows gains the direction T_l:  H += T_l T_l' / T_l . T_l,  K_i += (M_i . T_l / T_l . T_l) T_l.
                // Only the coefficients are formed here; the update itself is the next trip's.
                T g = T(0);  // M_i . T_l, in four pieces (register pressure peaks here)
                {
                    const double2 *mr = reinterpret_cast<const double2 *>(Ml + (isc ? hl : 0) * LDM);
                    const double2 *tr = reinterpret_cast<const double2 *>(kAv);
#pragma unroll
                    for (int q = 0; q < NV / 2; q += 2) {
                        const double2 a0 = mr[q], b0 = tr[q], a1 = mr[q + 1], b1 = tr[q + 1];
                        g += a0.x * b0.x + a0.y * b0.y + a1.x * b1.x + a1.y * b1.y;
                        pin(g);
                    }
                }
                T tl;
                {
                    T vv[NV];
                    ld16(vv, kAv);
                    tl = dot16(RT, vv);
                }
                const T tld = half_get(tl, hb, ldrop);
                const T itl = fast_rcp(tld);
                if (drp) {
                    cT = low ? ((hl == ldrop) ? T(-1) : (occ ? -tl * itl : T(0))) : kAv[l15] * itl;
                    cK = isc ? g * itl : T(0);
                    pdrop = true;
                }
                if (SEED && __ballot(ndrop) != 0ull) {
                    // the slot leaves with a multiplier lam_l < 0 (seeded start): the minimiser on the remaining rows'
                    // hyperplanes is y + (lam_l / |T_l|^2) T_l, its multipliers lam_a - lam_l (T_a . T_l) / |T_l|^2
                    const T f = half_get(lam, hb, ldrop) * itl;
                    if (drp && ndrop) {
                        if (isc) s = (pos >= 0) ? T(0) : s - f * g;
                        lam = occ ? lam - f * tl : lam;
                    }
                }
            }
            // ---- bookkeeping
            if (stepping) {
                // the implied primal point moved by t z: s_i -= t M_i . z
                if (isc) s = (pos >= 0) ? T(0) : s - t * mz;
                lam -= t * r;
                lam = (occ && lam < T(0)) ? T(0) : lam;
                up += t;
            }
            if (full) {  // p takes slot sl
                if (hl == sl) {
                    lam = up;
                    myact = p;
                    occ = true;
                }
                if (hl == p) {
                    pos = sl;
                    s = T(0);
                }
                mask |= 1u << sl;
                ++nq;
                needp = true;
            }
            const bool partial = stepping && !full;
            if (drp) {
                if (hl == ldrop) {
                    lam = T(0);
                    occ = false;
                }
                mask &= ~(1u << ldrop);
                --nq;
                dropping = false;
            }
            if (WARM && __ballot(drp && wdrop) != 0ull) {
                // warm-start repair: every slot whose multiplier was negative leaves, one pass each, then the
                // multipliers are solved again (rows dropped in excess come back through the usual iterations)
                const bool mine = drp && wdrop;
                if (hl == ldrop) negl = false;
                const unsigned nk = half_min((mine && negl && occ) ? (unsigned)hl : 0xffffffffu);
                const bool more = mine && nk != 0xffffffffu;
                const int nl = more ? (int)nk : 0;
                const int cl = half_get(myact, hb, nl);
                wsync();
                pstore = more && hl == nl;  // its row AFTER this pass: stored once the pending update is applied
                if (more && hl == cl) pos = -1;
                if (more) {
                    ldrop = nl;
                    dropping = true;
                } else if (mine) {
                    wdrop = false;
                    done = true;
                }
            }
            if (__ballot(partial) != 0ull) {
                // partial step: the next trip removes slot l from T (no update is pending for this half)
                const int cl = half_get(myact, hb, l);
                wsync();
                if (partial && hl == l) st16(kAv, RT);
                if (partial) {
                    if (hl == cl) pos = -1;
                    dropping = true;
                    ldrop = l;
                }
            }
            wsync();
        }
        tick(5);
        if (WARM && __ballot(recold) != 0ull) {
            wsync();
            if (recold) cold_reset();
            wsync();
            if (__ballot(!finished && done) == 0ull) continue;
        }
        if (__ballot(!finished) == 0ull) break;
        // ================================== multipliers by refinement, slacks re-evaluated
        // (halves that are already finished compute along and change nothing)
        if (wfix) lam = T(0);  // repair phase: lam = -T T' s0_A from scratch
        T y = -y0v[l15];       // y0 = -L^-1 q
        T yy[NV];
        T fresh = s0;          // slacks at y0
        // slacks of all rows at the point whose coordinates are in zv (M_i from the image)
        auto slacks = [&]() {
            T mi[NV];
            ld16(mi, Ml + (isc ? hl : 0) * LDM);
            const T f = hv[hl] - dot16(mi, yy);
            return isc ? f : INF;
        };
        wsync();
        if (low) actv[hl] = occ ? myact : 0;
        if (__ballot(!wfix && !finished) != 0ull) {
            rv[vofs] = lam;
            wsync();
            {
                T rr[NV];
                ld16(rr, rv);
                y -= ma_dot(rr);  // y = y0 - M_A' lam
            }
            zv[vofs] = y;
            wsync();
            ld16(yy, zv);
            fresh = slacks();
        } else {
            zv[vofs] = y;
            wsync();
            ld16(yy, zv);
        }
        T lraw = lam;  // multipliers before the clamp at zero (repair phase)
        // active residuals rho_a = h_a - M_a y should vanish. When they already do to REFTOL (1 + |h_a|) in both halves --
        // the usual case: a dozen rank-one updates of T in float64 -- the refinement step below would move y by less than
        // that and is skipped (two exchanged mat-vecs and a second evaluation of all slacks).
        T rho = half_get(fresh, hb, myact);
        rho = occ ? rho : T(0);
        // (warm-started launches always take it: there it is what computes the multipliers of a stored set)
        constexpr double REFTOL = 1e-11;
        const bool needref = WARM ? (nq > 0 && !finished)
                                  : (occ && !finished && !(fabs(rho) <= T(REFTOL) * (T(1) + fabs(half_get(hval, hb, myact)))));
        if (__ballot(needref) != 0ull) {
            // dlam = -W rho_A = -T (T' rho_A)
            wsync();
            kAv[vofs] = rho;
            if (low) st16(Timg + hl * NV, RT);  // T by columns is only needed here
            wsync();
            T uk;
            {
                T rr[NV];
                ld16(rr, kAv);
                const T *colp = Timg + l15;
                T a0 = T(0), a1 = T(0);
#pragma unroll
                for (int a = 0; a < NV; a += 2) {
                    a0 += rr[a] * colp[a * NV];
                    a1 += rr[a + 1] * colp[(a + 1) * NV];
                }
                uk = a0 + a1;  // (T' rho)_k, lane k < 16
            }
            rv[vofs] = low ? uk : T(0);
            wsync();
            T dl;
            {
                T rr[NV];
                ld16(rr, rv);
                dl = -dot16(RT, rr);
            }
            dl = occ ? dl : T(0);
            if (!finished) {
                lraw = lam + dl;
                lam = (occ && lraw < T(0)) ? T(0) : lraw;
            }
            // with dl as it is (not clamped) y moves exactly onto the active hyperplanes
            if constexpr (WARM) {
                // (a stored operator is not trusted: y must stay y0 - M_A' lam for the acceptance test to mean anything)
                wsync();
                rv[vofs] = dl;
                wsync();
                T rr[NV];
                ld16(rr, rv);
                y -= ma_dot(rr);
            } else {
                // y - M_A' dl = y + M_A' T (T' rho) = y + T' rho, because T' rho lies in the range of M_A' where
                // M_A' T = I - H is the identity (T is this launch's own operator)
                y += uk;
            }
            wsync();
            zv[vofs] = y;
            wsync();
            ld16(yy, zv);
            fresh = slacks();
        }
        // the loop's pending-update vector lives in zv: no update is pending here (the loop was left at its
        // break, after the update was applied and cleared), but the coefficients multiply whatever zv holds
        // ---- warm-start repair: the most negative multiplier's row leaves, then everything is solved again
        if (WARM && __ballot(wfix && !finished) != 0ull) {
            unsigned hi, lo;
            ordered(lraw, hi, lo);
            negl = wfix && !finished && occ && lraw < T(0);
            const unsigned key = negl ? ((hi & ~31u) | (unsigned)hl) : 0xffffffffu;
            const unsigned mkey = half_min(key);
            const bool rep = wfix && !finished;
            if (rep && mkey != 0xffffffffu) {
                const int l = (int)(mkey & 31u);
                ldrop = l;
                dropping = true;
                wdrop = true;
                done = false;
                needp = false;
            }
            const int cl = half_get(myact, hb, ldrop);
            wsync();
            if (rep && dropping && hl == ldrop) st16(kAv, RT);
            if (rep && dropping && hl == cl) pos = -1;
            wsync();
            if (rep && !dropping) {
                // every multiplier is >= 0: (y, A) is an S-pair. It goes through the acceptance test below
                // (lam, y and the slacks are those of this very set); violated rows, if any, enter through the
                // usual iterations from there.
                wfix = false;
                fails = -1;  // this first test is not a failed verification
            }
            if (__ballot(!finished && done) == 0ull) continue;
        }
        // ---- acceptance: no inactive row violated, and (after a warm start, whose operator is not trusted)
        //      every active row on its bound -- with stationarity by construction and lam >= 0 these are
        //      the KKT conditions of the strictly convex QP
        bool dirty = half_any(selectable && pos < 0 && !(fresh >= -T(4) * tolh), hb);  // (NaN counts as violated)
        if (__ballot(!finished) != 0ull) {
            // (needed after a warm start, whose operator is not trusted; cheap insurance for the cold solves, whose
            // refinement relies on the maintained operator as well)
            const T ra = half_get(fresh, hb, myact);
            const T ta = half_get(tolh, hb, myact);
            const bool off = half_any(occ && !(fabs(ra) <= T(1e3) * ta), hb);
            const bool neg = half_any(occ && !(lam >= T(0)), hb);
            dirty = dirty || off || neg;
        }
        bool coldnow = false;
        // u = L^-T y in lanes 16..31 (yy holds y; the rows of L^-T come back from their image)
        auto primal = [&]() {
            T lt[NV];
            ld16(lt, LTimg + l15 * NV);
            return dot16(lt, yy);
        };
        if (!finished && done && !wfix) {
            if (!dirty) {
                xsol = primal();
                status = MPCQP_SOLVED;
                finished = true;
            } else if (++fails < 4) {
                // continue the active-set loop from the re-evaluated slacks
                s = (pos >= 0) ? T(0) : fresh;
                status = MPCQP_MAX_ITER;
                done = false;
                needp = true;
            } else if (warm) {
                coldnow = true;  // the stored state did not lead to a certified point
            } else {
                xsol = primal();
                status = MPCQP_MAX_ITER;
                finished = true;
            }
        }
        if (WARM && __ballot(coldnow) != 0ull) {
            wsync();
            if (coldnow) cold_reset();
            wsync();
        }
        if (__ballot(!finished) == 0ull) break;
    }
    if (status == MPCQP_SOLVED && olam) {
        const T lv = half_get(lam, hb, pos < 0 ? 0 : pos);
        lam_out = (pos >= 0) ? lv : T(0);
    }

    tick(6);
    const bool ok = (status == MPCQP_SOLVED);
    if (valid) {
        const int k = hl - NV;  // lanes 16..31 hold u_k
        if (k >= 0 && k < n) oU[prob * (int64_t)n + k] = ok ? xsol : T(0);
        if (olam && isc) olam[prob * (int64_t)m + hl] = ok ? lam_out : T(0);
        if (hl == 0) {
            if (ostatus) ostatus[prob] = status;
            if (oiters) oiters[prob] = iters;
        }
        if (wstate && low) {  // the operator and the active set for the next period's warm start
            if constexpr (WARM) {
                double2 *dst = reinterpret_cast<double2 *>(wstate + hl * NV);
#pragma unroll
                for (int i = 0; i < NV / 2; ++i) dst[i] = double2{RT[2 * i], RT[2 * i + 1]};
            } else {
                // (MPCQP_WARM_ACTIVE_SET launches keep the row ids only: the operator's rows are marked as absent, so that
                // an MPCQP_WARM_OPERATOR launch that meets this record starts cold instead of trusting stale rows)
                wstate[hl * NV] = __builtin_nan("");
            }
            reinterpret_cast<int *>(wstate + NV * NV)[hl] = (ok && occ) ? myact : -1;
        }
    }
}

// ------------------------------------------------------------ host side
static Lay make_lay(const KernelArgs &ka)
{
    Lay L{};
    auto al = [](int c) { return (c + 3) & ~3; };  // 32-byte granules keep every region 16-byte aligned
    const int gimg = NV * ((ka.m + 1) | 1), main_x = 2 * NV * NV + NV;
    L.off_X = 0;
    int o = al(gimg > main_x ? gimg : main_x);
    L.off_Y = o;
    int y_build = 4 * 32 + 32;  // (spare 4 x 32), hp
    L.off_stage = L.off_Y + y_build;
    L.nA = (ka.A.step_stride ? ka.N : 1) * ka.nx * ka.nx;
    L.nB = (ka.B.step_stride ? ka.N : 1) * ka.nx * ka.nu;
    L.nC = ka.C.ptr ? (ka.C.step_stride ? ka.N : 1) * ka.mk * ka.nx : 0;
    L.nD = ka.D.ptr ? (ka.D.step_stride ? ka.N : 1) * ka.mk * ka.nu : 0;
    y_build += al(L.nA) + al(L.nB) + al(L.nC) + al(L.nD) + 2 * 16;  // + slack for the pair fetch past an odd horizon's last step
    int y_main = ka.m * LDM;                                                 // the M image
    if (y_main < NV * LDM + 2 * (NV + 2)) y_main = NV * LDM + 2 * (NV + 2);  // L image + the factorisation's column buffers
    const int y_sz = al(y_build > y_main ? y_build : y_main);
    L.off_hv = L.off_Y + y_sz;
    o = L.off_hv + HL + NV;
    L.off_v = o;
    o += 6 * NV;  // kAv rv zv | their shadows
    L.per = al(o);
    return L;
}

bool pair_eligible(const KernelArgs &ka, int mode, int dtype)
{
    if (dtype != MPCQP_F64 || (mode != MODE_FUSED && mode != MODE_MODEL)) return false;
    if (ka.n > NV || ka.m > MMAX || ka.m < 1) return false;
    if (mode == MODE_FUSED && ka.nx != 3 && ka.nx != 4) return false;
    const Lay L = make_lay(ka);
    return (size_t)L.per * 2 * sizeof(double) <= 64 * 1024;
}

// true when the launch fills the machine exactly once (two wavefronts on every SIMD: BASELINE config 2's 4096 problems on
// this chip, per GPU): workgroups of two wavefronts pay there and only there -- a partly filled machine is better served by
// single wavefronts, which the dispatcher spreads over all compute units (2048 problems: 18.9 against 22.8 us; 3500: 24.2
// against 26.5; 3900: 26.7 against 27.5; 4000: 27.0 against 28.0; 4096: 26.4 against 25.6)
static bool one_round(int64_t waves)
{
    const int simds = device_simds_now();
    return waves <= 2 * (int64_t)simds && waves > 2 * (int64_t)simds - 8;
}

template <int NX, int MK> static int launch_pair_t(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    const Lay L = make_lay(ka);
    const int64_t waves = (batch + 1) / 2;
    auto go = [&](auto kern, int wpb) {
        const size_t bytes = (size_t)L.per * 2 * sizeof(double) * wpb;
        const unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), bytes, st, (const double *)ka.A.ptr, (const double *)ka.B.ptr,
                           (const double *)ka.C.ptr, (const double *)ka.D.ptr, (const double *)ka.e.ptr,
                           (const double *)ka.x0.ptr, (const double *)ka.goal.ptr, (const double *)ka.targets.ptr,
                           (double *)ka.U, (double *)ka.lam, ka.status, ka.iters, ka, L, batch);
    };
    bool two = waves >= 2 && one_round(waves) && (size_t)L.per * 4 * sizeof(double) <= 64 * 1024;
#ifdef PAIR_FORCE_WPB1
    two = false;
#endif
    const bool seeded = (ka.opt_flags & MPCQP_OPT_SEED_VIOLATED) || (ka.warm_state && ka.warm_start == MPCQP_WARM_ACTIVE_SET);
    if (seeded && !(ka.warm_state && ka.warm_start != MPCQP_WARM_ACTIVE_SET)) {
        go(mpcqp_pair_kernel<NX, MK, false, false, 1, true>, 1);
    } else if (ka.warm_state && ka.warm_start != MPCQP_WARM_ACTIVE_SET) {  // (workgroups of two measured no different here: 23.1 us either way for a stored state that is accepted)
        go(mpcqp_pair_kernel<NX, MK, false, true>, 1);
    } else if (ka.order) {  // (cold launches only: mpcqp_capi.hip refuses the other combinations)
        go(mpcqp_pair_kernel<NX, MK, false, false, 1, false, true>, 1);
    } else if (two) {
        go(mpcqp_pair_kernel<NX, MK, false, false, 2>, 2);
    } else {
        go(mpcqp_pair_kernel<NX, MK, false, false>, 1);
    }
    return (int)hipGetLastError();
}

int launch_pair_model(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    const Lay L = make_lay(ka);
    const int64_t waves = (batch + 1) / 2;
    auto go = [&](auto kern, int wpb) {
        const size_t bytes = (size_t)L.per * 2 * sizeof(double) * wpb;
        const unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), bytes, st, (const double *)ka.model, (const double *)nullptr,
                           (const double *)nullptr, (const double *)nullptr, (const double *)ka.e.ptr, (const double *)ka.x0.ptr,
                           (const double *)ka.goal.ptr, (const double *)ka.targets.ptr, (double *)ka.U, (double *)ka.lam, ka.status,
                           ka.iters, ka, L, batch);
    };
    if (ka.order)
        go(mpcqp_pair_kernel<3, 0, true, false, 1, false, true>, 1);
    else if (waves >= 2 && one_round(waves) && (size_t)L.per * 4 * sizeof(double) <= 64 * 1024)
        go(mpcqp_pair_kernel<3, 0, true, false, 2>, 2);
    else
        go(mpcqp_pair_kernel<3, 0, true, false, 1>, 1);
    return (int)hipGetLastError();
}

int launch_pair(const KernelArgs &ka, int64_t batch, hipStream_t st)
{
    // the register-pipelined chain: terminal cost only, state constraints only, two rows per step
    const bool lean = ka.mk == 2 && ka.C.ptr && !ka.D.ptr && !(ka.flags & (MPCQP_P_STAGE | MPCQP_Q_STAGE));
    if (quad_eligible(ka, batch)) return launch_quad(ka, batch, st);  // cold lean launches that fill the machine about once: four problems per wavefront
    if (ka.nx == 3) return lean ? launch_pair_t<3, 2>(ka, batch, st) : launch_pair_t<3, 0>(ka, batch, st);
    return lean ? launch_pair_t<4, 2>(ka, batch, st) : launch_pair_t<4, 0>(ka, batch, st);
}

}  // namespace mpcqp
