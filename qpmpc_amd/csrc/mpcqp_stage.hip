// mpcqp_stage.hip -- gfx950 kernel for the STAGE-WISE (uncondensed) formulation: long horizons.
//
// The reference has no sparse formulation (sparse=True only wraps the dense condensed matrices in CSC,
// qpmpc/mpc_qp.py:39,108-109; qpmpc/solve_mpc.py:31-32): its build is O(N^2) memory and O(N^3) time and the
// dense kernels of this library stop at n = N nu = 256. This kernel solves the same QP
//
//   min 1/2 sum w_u |u_k|^2 + 1/2 sum_{1<=k<N} w_x |x_k - xref_k|^2 + 1/2 w_t |x_N - x_goal|^2
//   s.t. x_{k+1} = A_k x_k + B_k u_k,  C_k x_k + D_k u_k <= e_k
//
// without ever forming P or G: O(N) memory, O(N) work per active-set iteration. Restated on the CPU in
// oracle/stagewise_np.py; same minimiser as the dense path (tests/golden/stagewise_*.npz come from the
// reference-built dense QP).
//
// Method -- Goldfarb-Idnani's dual active set in the metric of the condensed Hessian P:
//   * v -> P^-1 v is ONE LQR SOLVE: Riccati gains (Acl_k = A_k - B_k K_k, K_k, S_k^-1) computed once per
//     problem, then a backward sweep (costate p_k, feed-forward ff_k) and a forward sweep (u_k, x_k);
//   * a row of G applied to a vector is a read of that vector's state trajectory:
//     g_i . v = C_k[r] x_k(v) + D_k[r] v_k;
//   * per active row a the slot keeps V_a = P^-1 g_a' and its trajectory X_a, and W = (G_A P^-1 G_A')^-1
//     (|A| x |A|) is bordered / deflated by rank-one updates. Step along z = -(V_p - sum_a r_a V_a),
//     r = W c, c_a = g_a . V_p, d2 = g_p . V_p - c . r; full step t2 = -s_p / d2, partial step
//     t1 = min lam_a / r_a; the primal point is implied: u = u0 - sum_a lam_a V_a.
//
// Mapping -- ONE PROBLEM PER WAVEFRONT, the horizon cut into 64 CHUNKS of L = ceil(N/64) steps, lane j
// owning chunk j for everything (slacks, slot vectors, trajectories): the O(|A| N) work of an iteration is
// spread over the 64 lanes, and the two sweeps of the LQR solve -- affine recurrences, serial in k -- are
// chunked scans: every lane runs its chunk from a zero inflow, the 64 chunk boundaries are chained through
// the chunks' transition matrices (readlane, 63 small mat-vecs), then every lane re-runs its chunk from
// its true inflow: depth 2 L + 63 instead of N. All per-problem arrays live in a caller-owned HBM
// workspace (they are re-read by the same CU: L1/L2-resident for the horizons this is meant for); the
// small vectors shared by the lanes (c, r, multipliers, the active list) live in LDS.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "mpcqp.h"
#include "mpcqp_internal.h"
#include "mpcqp_plant.h"

namespace mpcqp {

// (developer knobs: tools/ab_unit.sh builds variants)
// priority of a solving wavefront of the five-wavefront workgroups (the factor wavefront runs at 0): worth 1-2 % on config 3
#ifndef STAGE_PRIO
#define STAGE_PRIO 3
#endif
#ifndef STAGE_FPRIO
#define STAGE_FPRIO 0
#endif
#ifndef STAGE_SRD
#define STAGE_SRD 4
#endif
#ifndef STAGE_PW
// problems per workgroup of the pipelined instantiation (4 when four of them fit the CU's LDS; else 1).
// 1: two wavefronts per workgroup, the problem's solving wavefront and its factor wavefront, on neighbouring SIMDs: every SIMD
//    hosts the solving wavefront of one problem and the factor wavefront of another.
// 4 (round 5): FIVE wavefronts -- four solving ones and ONE factor wavefront for all four problems: v_mfma_f64_4x4x4 carries four
//    independent products, one per lane quad of a 16-lane row, so quad b of the factor wavefront runs the recursion of problem b
//    on the instruction stream that served one problem before. The chip issues a quarter of the factor instructions, three of
//    four solving wavefronts have their SIMD to themselves, and the one that shares issues first (s_setprio).
//    (Round 4's variant of 4 -- eight wavefronts, a problem's two on ONE SIMD -- made every workgroup as slow as the slow ones
//    of 1: a SIMD issued one solving and one factor wavefront's instructions per period either way.)
#define STAGE_PW 4
#endif

namespace stage {

constexpr int kSerialMaxN = 128;  // horizons up to here take the serial sweeps of the LQR solve (if their factor fits LDS)
constexpr int serial_fs(int nu) { return (16 + 12 * nu + nu * nu + 1) & ~1; }  // doubles per step of the LDS factor image
// LDS doubles of a serial instantiation behind the active-set vectors: exchange cells, the image (STAGE_SRD steps of slack on
// either side: the sweeps request that far ahead without clamping), the feed-forward terms, the targets / the staged trajectory
// ... and one cell of nu doubles per lane that takes the stores of the lanes whose value is not wanted (no exec masking)
constexpr int64_t serial_lds_doubles(int N, int nx, int nu)
{
    return 32 + 2 * STAGE_SRD * serial_fs(nu) + (int64_t)N * (serial_fs(nu) + nu + nx) + 64 * nu;
}

struct Ws {  // per-problem workspace carve, in doubles (host-computed, passed by value)
    int64_t Acl, Kg, Sinv, Fimg, ff, U0, X0, s, invn, rowslot, V, XV, W, total;
    int maxq;
};

__host__ __device__ inline Ws make_ws(int nx, int nu, int N, int mk, int maxq)
{
    Ws w{};
    int64_t o = 0;
    auto take = [&](int64_t cnt) {
        const int64_t at = o;
        o += (cnt + 1) & ~(int64_t)1;
        return at;
    };
    // per-step arrays are stored TRANSPOSED by chunk: step k = lane L + kk lives at index kk 64 + lane, so that the 64
    // lanes of a wavefront, each walking its own chunk, touch 64 consecutive entries per instruction (coalesced)
    // instead of 64 cache lines; they are sized for the padded horizon NP = 64 ceil(N / 64)
    const int64_t NP = 64 * (int64_t)((N + 63) / 64);
    const int64_t m = NP * mk;
    w.Acl = take(NP * nx * nx);
    w.Kg = take(NP * nu * nx);
    w.Sinv = take(NP * nu * nu);
    // two copies of the LDS factor image (MPCQP_OPT_KEEP_FACTOR / REUSE_FACTOR: slot MpcqpSolveOpts.factor_slot;
    // MPCQP_OPT_PIPELINE_FACTOR: the solve reads one while the next launch's factor is written into the other)
    w.Fimg = take(N <= kSerialMaxN ? 2 * (int64_t)N * serial_fs(nu) : 0);
    w.ff = take(NP * nu);
    w.U0 = take(NP * nu);
    w.X0 = take(NP * nx);
    w.s = take(m);
    w.invn = take(m);
    w.rowslot = take((m + 1) / 2);  // int32 per row
    w.V = take((int64_t)(maxq + 1) * NP * nu);   // slot maxq: the candidate row of the current iteration
    w.XV = take((int64_t)(maxq + 1) * NP * nx);
    w.W = take((int64_t)maxq * maxq);
    // the problems of a launch walk their workspaces in lock step: a stride that is a large power of two would put
    // all of them on the same memory channels. Odd multiple of 512 B.
    o = (o + 63) & ~(int64_t)63;
    if (((o >> 6) & 1) == 0) o += 64;
    w.total = o;
    w.maxq = maxq;
    return w;
}

__device__ __forceinline__ double rl(double x, int lane)  // lane must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
// reciprocal: hardware estimate + two Newton steps (the IEEE division sequence is ~15 dependent instructions)
__device__ __forceinline__ double frcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }
// (value, index) arg-min over the wavefront; ties -> lowest index; every lane gets the result
__device__ __forceinline__ void wave_argmin(double &v, int &idx) { wave_argmin_dpp(v, idx); }
// stores of one lane become visible to the other lanes of the wavefront (same CU, same L1)
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// hand-over through LDS inside ONE wavefront: its LDS operations execute in order, so nothing has to be waited for --
// only the compiler must not move the accesses across this point
__device__ __forceinline__ void lsync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace stage

using namespace stage;

// bytes per problem of the warm-state record (MpcqpSolveOpts.warm_state) for `maxq` slots
__host__ __device__ size_t stage_warm_bytes(int maxq) { return ((size_t)(4 + 2 * maxq) * sizeof(int) + 15) & ~(size_t)15; }

// PIPE (MPCQP_OPT_PIPELINE_FACTOR; serial instantiations): TWO wavefronts per problem. Wavefront 0 solves with the factor a
// previous launch left in the workspace (as MPCQP_OPT_REUSE_FACTOR does), wavefront 1 -- on another SIMD, at the same time --
// runs the Riccati recursion on the problem's operands as they are NOW and leaves that factor in the other image for the
// NEXT launch. In a receding-horizon loop whose operands for the next period are known when this period is solved
// (time-invariant or pre-scheduled LTV dynamics: examples/wheeled_inverted_pendulum.py:99-118) the factor is still rebuilt
// every period, like the reference's solve_mpc does, but off the period's critical path.
// WARM: the warm-start machinery (MpcqpSolveOpts.warm_state) is compiled only into the instantiations that a launch with a
// warm-state record selects: the cold instantiations keep the registers and the code size they had without it.
// The kernel's ONE argument (round-3 advisor finding): the period below reads its arguments where they lie in the kernarg
// segment, so their offsets are offsetof() of this type -- the single by-value argument starts the segment -- instead of
// hand-computed positions of four separate parameters.
struct StageArgs {
    KernelArgs ka;
    Ws wl;
    double *wsbase;
    int64_t batch;
    int64_t lds_problem;  // dynamic LDS bytes of ONE problem (the pipelined instantiation packs STAGE_PW problems per workgroup)
};
static_assert(offsetof(StageArgs, ka) == 0, "the kernel argument block starts with KernelArgs");

template <int NX, int NU, bool SERIAL, bool PIPE, bool WARM, int PWT = 1>
__global__ void __launch_bounds__(PIPE ? (PWT == 4 ? 320 : 128) : 64, PIPE && PWT == 4 ? 1 : 2) mpcqp_stage_kernel(const StageArgs sa_)
{
    const KernelArgs &ka_ = sa_.ka;
    // ONE PERIOD = one build + solve (+ the fused plant epilogue). A launch runs ka.ep_periods of them back to back
    // (mpcqp_wip_periods_batch: the loop's next problem is written by the epilogue, so the wavefront carries on with it --
    // no launch boundary, no dispatch gap between the periods); every other entry point runs one. (Everything, the
    // address arithmetic included, is inside the period: nothing but the kernel's arguments stays live across periods.)
    // (the one thing a period hands to the next: the loop's plant state after its epilogue -- the next problem's x0, goal and
    // targets are functions of it, so a later period of a launch forms them in registers instead of reading back what the
    // epilogue has just stored: one global round trip less at the top of every period but the first)
    double carry[4] = {0.0, 0.0, 0.0, 0.0};
    auto period = [&](const int per) {
    const long long t_entry = (long long)__builtin_readcyclecounter();  // (developer probe: slot 11)
    // (the lane and problem indices pass through an empty asm: the optimiser must not hoist the period's address
    // arithmetic out of the period loop, where all of it would stay live across the whole period -- that version of the
    // kernel spilled 50-200 VGPRs)
    typedef const __attribute__((address_space(4))) unsigned char *KargPtr;
    KargPtr kbase = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    int tid = threadIdx.x;
    constexpr int PW = PIPE ? PWT : 1;  // problems per workgroup
    const int wvi = PIPE ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;  // wavefront of the workgroup
    int64_t prob = (int64_t)blockIdx.x * PW + (wvi % PW);
    // MULTI: the instantiations that mpcqp_wip_periods_batch reaches (the plant of the fused period has nx = 4, nu = 1)
    constexpr bool MULTI = SERIAL && NX == 4 && NU == 1;
    if constexpr (MULTI) asm volatile("" : "+s"(kbase), "+s"(prob), "+v"(tid));
    // (the kernel's arguments, read where they lie in the kernarg segment: the fields of StageArgs)
    constexpr size_t off_wl = offsetof(StageArgs, wl), off_ws = offsetof(StageArgs, wsbase);
    const __attribute__((address_space(4))) KernelArgs &ka = *(const __attribute__((address_space(4))) KernelArgs *)kbase;
    const __attribute__((address_space(4))) Ws &wl = *(const __attribute__((address_space(4))) Ws *)(kbase + off_wl);
    double *wsbase = *(double *const __attribute__((address_space(4))) *)(kbase + off_ws);
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_smem_all[];
    unsigned char *stage_smem = stage_smem_all + (PW > 1 ? (size_t)(wvi % PW) * (size_t)sa_.lds_problem : 0);
    const int lane = tid & 63;
    // serial sweeps: the running vector sits the way the matrix cores take a B operand (and return a result): component sq in
    // every lane of the 16-lane row sq; sc: the column of a matrix element this lane fetches as an A operand
    const int sq = lane >> 4, sc = lane & 3;
    const bool sqin = sq < NX;
    auto mm44 = [](double a, double b, double cc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, cc, 0, 0, 0); };
    const bool factor_wave = PIPE && wvi >= PW;
    if constexpr (PIPE && PW == 4) {  // the solving wavefronts are the period's critical path: the one that shares its SIMD with the
                                      // factor wavefront issues first, the factor wavefront fills the gaps
        if (factor_wave)
            __builtin_amdgcn_s_setprio(STAGE_FPRIO);
        else
            __builtin_amdgcn_s_setprio(STAGE_PRIO);
    }
    const int N = ka.N, mk = ka.mk, maxq = wl.maxq;
    const int L = (N + 63) / 64;                    // steps per chunk
    const int k0 = lane * L < N ? lane * L : N;     // this lane's chunk [k0, k1)
    const int k1 = (k0 + L < N) ? k0 + L : N;
    const int nch = (N + L - 1) / L;                // chunks that hold steps
    const int64_t NP = 64 * (int64_t)L;             // padded horizon (slot stride of the per-step arrays)
    auto wq = [&](int k) { return (int64_t)(k - k0) * 64 + lane; };  // workspace index of a step of THIS lane's chunk
    auto wg = [&](int k) {                                             // ... of any step
        if (L == 1) return (int64_t)k;  // (one step per lane: no division in the serial sweeps)
        const int j = k / L;
        return (int64_t)(k - j * L) * 64 + j;
    };
    const double INF = HUGE_VAL;
    // ---- LDS: vectors shared by the lanes
    double *cv = (double *)stage_smem, *rv = cv + maxq, *lamv = rv + maxq;
    int *actk = (int *)(lamv + maxq), *actr = actk + maxq;
    // SERIAL: the factor of every step stays in LDS -- the serial sweeps read nothing else, and nothing of it goes to the
    // workspace -- in the order in which quad q of a 16-lane row consumes it (16-byte reads):
    //   FA  Acl[q][j] at q 4 + j   (the sweeps on the matrix cores fetch it by element: as it is for the backward sweep's Acl', transposed
    //       for the forward one -- round 4 kept a second, transposed copy for its 16-byte row reads)
    //   FKN -K[i][j] at j NU + i     FBS -(S^-1 B')[i][j] at j NU + i     (the same 4 NU values for every lane)
    //   FBO B[q][i]      FSI S^-1
    constexpr int FA = 0, FKN = 16, FBS = FKN + 4 * NU, FBO = FBS + 4 * NU, FSI = FBO + 4 * NU;
    constexpr int FS = serial_fs(NU);
    static_assert(FS >= FSI + NU * NU && (FS & 1) == 0, "factor image");
    typedef double D2 __attribute__((ext_vector_type(2)));
    double *rsc = (double *)(actr + maxq);  // 32 doubles of exchange for the factor (32 maxq bytes precede it: 16-byte aligned)
    double *Fl = rsc + 32 + STAGE_SRD * FS;  // (SRD steps of slack below the image and above the targets: serial_lds_doubles)
    double *ffl = Fl + N * FS;             // ... and the feed-forward terms of the latest backward sweep (N x NU)
    double *tgl = ffl + N * NU;            // ... and the targets of the tracking sweep (N x NX)
    // LDSIMG (four problems per workgroup, one factor wavefront): a SECOND image behind that wavefront's operand copies. The factor
    // wavefront writes the next period's factor straight into the image the solving wavefront is not reading, and the two swap
    // from period to period: inside a multi-period launch the factor never travels through the workspace (only the launch's
    // first period loads one from there, only its last period leaves one there).
    constexpr bool LDSIMG = PIPE && PW == 4;
    double *const Fl0 = Fl, *const Fl1 = rsc + serial_lds_doubles(N, NX, NU) + 32 + (int64_t)N * (16 + 4 * NU);
    if constexpr (LDSIMG) Fl = (per & 1) ? Fl1 : Fl0;
    // ---- workspace
    double *ws = wsbase + prob * wl.total;
    double *Acl = ws + wl.Acl, *Kg = ws + wl.Kg, *Sinv = ws + wl.Sinv, *ffv = ws + wl.ff, *U0 = ws + wl.U0, *X0 = ws + wl.X0;
    double *sl = ws + wl.s, *invn = ws + wl.invn, *Vs = ws + wl.V, *XVs = ws + wl.XV, *Wm = ws + wl.W;
    int *rowslot = (int *)(ws + wl.rowslot);
    // ---- operands
    const double *gA = (const double *)ka.A.ptr + prob * ka.A.batch_stride;
    const double *gB = (const double *)ka.B.ptr + prob * ka.B.batch_stride;
    const double *gC = ka.C.ptr ? (const double *)ka.C.ptr + prob * ka.C.batch_stride : nullptr;
    const double *gD = ka.D.ptr ? (const double *)ka.D.ptr + prob * ka.D.batch_stride : nullptr;
    const double *ge = (const double *)ka.e.ptr + prob * ka.e.batch_stride;
    const double *gx0 = (const double *)ka.x0.ptr + prob * ka.x0.batch_stride;
    const double *ggoal = ka.goal.ptr ? (const double *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
    const double *gtgt = ka.targets.ptr ? (const double *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
    const int64_t sA = ka.A.step_stride, sB = ka.B.step_stride, sC = ka.C.step_stride, sD = ka.D.step_stride, sE = ka.e.step_stride;
    const bool stageP = ka.flags & MPCQP_P_STAGE, termP = ka.flags & MPCQP_P_TERMINAL;
    const bool stageQ = (ka.flags & MPCQP_Q_STAGE) && gtgt, termQ = (ka.flags & MPCQP_Q_TERMINAL) && ggoal;
    const double wu = ka.wu, wx = stageP ? ka.wx : 0.0, wt = termP ? ka.wt : 0.0;

    long long *stamp = ka.probe ? (long long *)ka.probe + prob * 16 : nullptr;  // developer probe (MpcqpSolveOpts.probe)
    auto tick = [&](int slot) {
        if (stamp && lane == 0) stamp[slot] = (long long)__builtin_readcyclecounter();
    };
    // (developer probe, launches of ONE period without the pipelined factor: slots 9, 10, 12, 13 are free there) time spent in the
    // parts of an active-set iteration: sweeps [9], c and r = W c [10], ratio test [12], slack update [13]; the rest is the W update
    long long tlast = 0;
    const bool acc_on = stamp && !PIPE && !(ka.ep_on && ka.ep_periods > 1);
    auto tacc = [&](int slot) {
        if (acc_on) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0 && slot >= 0) stamp[slot] += now - tlast;
            tlast = now;
        }
    };
    if (acc_on && lane == 0) stamp[9] = stamp[10] = stamp[12] = stamp[13] = 0;
    // ================================================================= factor: Riccati recursion
    // Serial in k and nonlinear; the lane layout is described where the recursion starts (below).
    // MPCQP_OPT_REUSE_FACTOR: A, B and the weights are those of the launch that left its factor in this workspace
    // (MPCQP_OPT_KEEP_FACTOR): the recursion is skipped -- build once, re-solve (mpc_qp.py:129-163 usage).
    const bool reuse = PIPE ? !factor_wave : (ka.opt_flags & MPCQP_OPT_REUSE_FACTOR) != 0;
    const bool keep = !PIPE && (ka.opt_flags & MPCQP_OPT_KEEP_FACTOR);
    if constexpr (PIPE) rsc += serial_lds_doubles(N, NX, NU);  // (the factor wavefront's own exchange cells, after everything)
    tick(factor_wave ? 9 : 0);
    if (stamp && lane == 0 && !factor_wave) stamp[11] = t_entry;
    if (stamp && lane == 0)  // HW_ID (wave, SIMD, CU, SH, SE) | XCC_ID << 32
        stamp[factor_wave ? 15 : 14] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) << 32);
    // the fused period's plant state (epilogue): requested now, used ~70 k cycles later
    double ep_s0[4] = {0.0, 0.0, 0.0, 0.0};
    const bool carried = SERIAL && NX == 4 && NU == 1 && ka.ep_on && per > 0;
    if (ka.ep_on && !factor_wave) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ep_s0[i] = carried ? carry[i] : ((const double *)ka.ep_states)[prob * 4 + i];
    }
    // (serial sweeps) the problem's own data, requested now and used once the factor is in place: x0 and the goal (this
    // lane's component), the targets (64 consecutive values per register; they go to LDS in the tracking sweep)
    constexpr int TGV = SERIAL ? (kSerialMaxN * NX + 63) / 64 : 1;
    double x0own = 0.0, goalown = 0.0, tgv[TGV];
#pragma unroll
    for (int u = 0; u < TGV; ++u) tgv[u] = 0.0;
    if constexpr (SERIAL) {
        if (!factor_wave && carried) {
            // the problem the last epilogue wrote (wip_period_wave, mpcqp_plant.h: the same expressions, bit for bit)
            const double rr = carry[0], vel = ka.ep_vel, Tp = ka.ep_Tp;
            x0own = sq == 0 ? carry[0] : sq == 1 ? carry[1] : sq == 2 ? carry[2] : carry[3];
            goalown = termQ ? (sq == 0 ? rr + ((double)N * Tp) * vel : sq == 2 ? vel : 0.0) : 0.0;
#pragma unroll
            for (int u = 0; u < TGV; ++u) {
                const int i = lane + 64 * u, kk = i >> 2, j = i & 3;
                tgv[u] = (stageQ && i < N * NX) ? (j == 0 ? rr + ((double)kk * Tp) * vel : j == 2 ? vel : 0.0) : 0.0;
            }
        } else if (!factor_wave) {
            x0own = sqin ? gx0[sq] : 0.0;
            goalown = (termQ && sqin) ? ggoal[sq] : 0.0;
#pragma unroll
            for (int u = 0; u < TGV; ++u) tgv[u] = (stageQ && lane + 64 * u < N * NX) ? gtgt[lane + 64 * u] : 0.0;
        }
    }
    // factor images in the workspace: this period's (read by REUSE / the solving wavefront, written by KEEP) and the next one's
    // (PIPE: the two alternate from period to period)
    const int slot = (ka.factor_slot + (PIPE ? per : 0)) & 1;
    double *img = ws + wl.Fimg + (int64_t)slot * N * FS;
    double *img_next = ws + wl.Fimg + (int64_t)(slot ^ 1) * N * FS;
    // S_k = w_u I + B_k' P_{k+1} B_k are the Schur complements of the condensed Hessian in the order u_{N-1}, ..., u_0: P is
    // positive definite iff every S_k is (mpc_problem.py:104-107 only guarantees w_u > 0; a negative state weight can
    // still make P indefinite). A pivot that is not positive -> MPCQP_NOT_PD, like the condensed kernels' Cholesky.
    bool notpd = false;
    if (!reuse) {
        // (round 5) The recursion runs on the MATRIX CORES: v_mfma_f64_4x4x4 multiplies 4 x 4 float64 matrices held ONE ELEMENT PER
        // LANE -- operand A[i][k] in lane i + 16 k, operand B[k][j] in lane j + 16 k, result D[i][j] in lane j + 16 i, the same in
        // each of the four lane quads b = (lane / 4) % 4 of a 16-lane row (four independent products; here four copies of one).
        // (tools/ubench/mfma_f64_4x4.hip prints that layout; 17 cycles per instruction, 28 from a result to its dependent product.)
        // A result is already in B-operand order, and read as an A operand it is its own TRANSPOSE, so a step is a chain of
        // products with no lane exchange at all (element [r][c] of every matrix in lane 16 r + c, r = lane / 16, c = lane % 4):
        //   T = P A, PB = P B (P symmetric: P as the A operand)   BPA = (PB)' A, S = w_u I + (PB)' B   K = S^-1 BPA
        //   A_cl = A - B K   P_k = Q + (A_cl' T + T' A_cl) / 2  (the two products are exact mirrors: P stays exactly symmetric)
        // B is held with its columns repeated (B[r][c % NU]), which repeats the rows of BPA, S and K the same way: every lane
        // has the entries it needs. ~30 instructions per step (six or eight on the matrix pipe, which runs beside the vector
        // pipe the solving wavefront of the same SIMD lives on) against ~85 of float64 DPP arithmetic in round 4.
        const int r = lane >> 4, c = lane & 3;
        const bool in = r < NX && c < NX;
        auto mm = [](double a, double b, double cc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, cc, 0, 0, 0); };
        // SF: ONE factor wavefront for the workgroup's four problems -- quad b = (lane / 4) % 4 of every row runs problem b (the
        // products of the four quads are independent; every other instruction of a step is per lane): per-lane pointers
        constexpr bool SF = PIPE && PW == 4;
        const int quad = (lane >> 2) & 3;
        const int64_t lds_pd = sa_.lds_problem / (int64_t)sizeof(double);
        double *imgq = img_next;                              // where this lane's problem takes its next factor: the workspace ...
        const int nper_ = (ka.ep_on && ka.ep_periods > 1) ? ka.ep_periods : 1;
        const bool to_lds = LDSIMG && per + 1 < nper_;        // ... or (LDSIMG, not the launch's last period) its other LDS image
        double *lq = (((per + 1) & 1) ? Fl1 : Fl0) + (SF ? quad * lds_pd : 0);
        const double *laq = rsc + 32 + (SF ? quad * lds_pd : 0);  // ... and finds its operands (PIPE)
        if constexpr (SF) {
            int64_t pq = (int64_t)blockIdx.x * PW + quad;
            pq = pq < sa_.batch ? pq : sa_.batch - 1;  // (the last workgroup may hold fewer than four problems: a quad repeats the last)
            imgq = wsbase + pq * wl.total + wl.Fimg + (int64_t)(slot ^ 1) * N * FS;
        }
        double P = (in && r == c) ? wt : 0.0;  // P[r][c]
        // operands are requested RD steps ahead into a register ring (a step is shorter than an HBM round trip)
        constexpr int RD = 3;
        double Amn[RD], Brn[RD][NU], Bik[RD];  // A[r][c]; B[r][u]; (NU = 2) -B[c][r], r < NU: the A operand of B K
        // PIPE: the factor wavefront writes its factor to the WORKSPACE, and a load issued behind those stores would wait for
        // them (vector memory operations retire in order): the operands come through LDS instead, one bulk copy up front --
        // A transposed and both padded to four rows (zeros)
        const int sAl = sA ? 16 : 0, sBl = sB ? 4 * NU : 0;
        const int na = (sA ? N : 1) * 16, nb = (sB ? N : 1) * 4 * NU;
        const double *la = laq, *lb = laq + na;
        // (SF: the loads of all four problems are in flight together -- one round trip per turn, not one per problem)
        // (A LATER period of a multi-period launch finds the copy of the first one: the launch's operands are its arguments, nothing
        // writes them while it runs, and nothing else lives in that part of LDS.)
        constexpr int NPQ = SF ? PW : 1;
        if (PIPE && per == 0) {
            const double *srcA[NPQ], *srcB[NPQ];
            double *dst[NPQ];
#pragma unroll
            for (int j = 0; j < NPQ; ++j) {
                int64_t pj = SF ? (int64_t)blockIdx.x * PW + j : prob;
                pj = pj < sa_.batch ? pj : sa_.batch - 1;
                srcA[j] = (const double *)ka.A.ptr + pj * ka.A.batch_stride;
                srcB[j] = (const double *)ka.B.ptr + pj * ka.B.batch_stride;
                dst[j] = rsc + 32 + j * lds_pd;
            }
            for (int i0 = lane; i0 < na; i0 += 64 * 8) {
                double v[NPQ][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 64 * u < na ? i0 + 64 * u : na - 1;
                    const int kk = i >> 4, cc = (i >> 2) & 3, rr = i & 3;  // la[k][c][r] = A_k[r][c]
#pragma unroll
                    for (int j = 0; j < NPQ; ++j) v[j][u] = (rr < NX && cc < NX) ? srcA[j][(int64_t)kk * sA + rr * NX + cc] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 64 * u < na)
#pragma unroll
                        for (int j = 0; j < NPQ; ++j) dst[j][i0 + 64 * u] = v[j][u];
            }
            for (int i0 = lane; i0 < nb; i0 += 64 * 4) {
                double v[NPQ][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 64 * u < nb ? i0 + 64 * u : nb - 1;
                    const int kk = i / (4 * NU), e = i - kk * 4 * NU;  // lb[k][l][u] = B_k[l][u], l < 4
#pragma unroll
                    for (int j = 0; j < NPQ; ++j) v[j][u] = e < NX * NU ? srcB[j][(int64_t)kk * sB + e] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + 64 * u < nb)
#pragma unroll
                        for (int j = 0; j < NPQ; ++j) dst[j][na + i0 + 64 * u] = v[j][u];
            }
            wsync();
        }
        auto request = [&](int d, int k) {
            if constexpr (PIPE) {
                const double *A = la + k * sAl, *B = lb + k * sBl;
                Amn[d] = A[c * 4 + r];
#pragma unroll
                for (int u = 0; u < NU; ++u) Brn[d][u] = B[r * NU + u];
                Bik[d] = (NU > 1 && r < NU) ? -B[c * NU + (r < NU ? r : 0)] : 0.0;
            } else {
                const double *A = gA + k * sA, *B = gB + k * sB;
                Amn[d] = in ? A[r * NX + c] : 0.0;
#pragma unroll
                for (int u = 0; u < NU; ++u) Brn[d][u] = r < NX ? B[r * NU + u] : 0.0;
                Bik[d] = (NU > 1 && r < NU && c < NX) ? -B[c * NU + (r < NU ? r : 0)] : 0.0;
            }
        };
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            request(d, N - 1 - d >= 0 ? N - 1 - d : 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int cu = c & (NU - 1), ru = r & (NU - 1);
        auto step = [&](int d, int k) {
            const double Am = Amn[d];
            const double Bk = (NU == 1 || cu == 0) ? Brn[d][0] : Brn[d][NU - 1];  // B[r][c % NU]
            const double T = mm(P, Am, 0.0);    // (P A)[r][c]
            const double PB = mm(P, Bk, 0.0);   // (P B)[r][c % NU]
            const double BPA = mm(PB, Am, 0.0); // (B' P A)[r % NU][c]
            const double Sm = mm(PB, Bk, 0.0);  // (B' P B)[r % NU][c % NU]
            double Si[NU * NU], Kd, Aclo;       // K[r % NU][c], Acl[r][c]
            auto si = [&](int i) { return Si[i < NU * NU ? i : 0]; };  // (an index the NU = 1 instantiation never takes stays in range)
            if constexpr (NU == 1) {
                const double S = Sm + wu;
                notpd |= !(S > 0.0);
                Si[0] = frcp(S);
                Kd = Si[0] * BPA;
                Aclo = Am - Bk * Kd;
            } else {
                // (SF: the entries of this lane's own quad, by ds_bpermute)
                auto sget = [&](int rr, int cc) {
                    if constexpr (SF) {
                        const int src = 4 * (16 * rr + 4 * quad + cc);
                        return __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(Sm)),
                                                __builtin_amdgcn_ds_bpermute(src, __double2loint(Sm)));
                    } else {
                        return rl(Sm, 16 * rr + cc);
                    }
                };
                const double s00 = sget(0, 0) + wu, s01 = sget(0, 1), s10 = sget(1, 0), s11 = sget(1, 1) + wu;
                const double det = s00 * s11 - s01 * s10, id = frcp(det);
                notpd |= !(s00 > 0.0) | !(det > 0.0);
                Si[0] = s11 * id;
                Si[1] = -s01 * id;
                Si[2] = -s10 * id;
                Si[3] = s00 * id;
                // K = S^-1 BPA and A_cl = A - B K as two more products: A operand [i][k] in lane i + 16 k
                const double SiA = r < NU ? (cu == 0 ? (r == 0 ? si(0) : si(1)) : (r == 0 ? si(2) : si(3))) : 0.0;  // Si[c % 2][r]
                Kd = mm(SiA, BPA, 0.0);
                Aclo = mm(Bik[d], Kd, Am);
            }
            // P_k = Q_k + A_cl' (P A), symmetrised (x_0 is data: Q_0 = 0): the mirrored product forms the same sums in the same
            // order, so the mean is exactly symmetric -- without it the antisymmetric rounding error is multiplied by ~|Acl| |A|
            // per step (1e-6 after 40)
            const double s1 = mm(Aclo, T, 0.0), s2 = mm(T, Aclo, 0.0);
            const double Pn = 0.5 * (s1 + s2) + ((in && r == c && k >= 1) ? wx : 0.0);
            if constexpr (SERIAL) {
                // every lane stores (the four quads hold bitwise equal copies; zero outside NX x NX); PIPE: straight into the next
                // launch's image in the workspace (the LDS image belongs to the solving wavefront)
                double bsv = 0.0;  // -(S^-1 B')[u][r], u = c % NU: the backward sweep's feed-forward row, S^-1 folded in here
#pragma unroll
                for (int v = 0; v < NU; ++v) bsv -= ((NU == 1 || cu == 0) ? si(v) : si(NU + v)) * Brn[d][v];
                const double siv = ru == 0 ? (cu == 0 ? si(0) : si(1)) : (cu == 0 ? si(2) : si(3));
                auto put = [&](double *f) {
                    f[FA + r * 4 + c] = Aclo;
                    f[FKN + c * NU + ru] = -Kd;
                    f[FBS + r * NU + cu] = bsv;
                    f[FBO + r * NU + cu] = Bk;
                    f[FSI + ru * NU + cu] = siv;
                };
                if constexpr (LDSIMG) {
                    if (to_lds)
                        put(lq + k * FS);
                    else
                        put(imgq + k * FS);
                } else {
                    put((PIPE ? imgq : Fl) + k * FS);
                }
            } else {
                const int64_t w = wg(k);
                if ((lane & 12) == 0) {  // the first quad of every row
                    if (in) Acl[w * NX * NX + r * NX + c] = Aclo;
                    if (r < NU && c < NX) Kg[w * NU * NX + r * NX + c] = Kd;
                    if (r < NU && c < NU) Sinv[w * NU * NU + r * NU + c] = r == 0 ? (c == 0 ? si(0) : si(1)) : (c == 0 ? si(2) : si(3));
                }
            }
            P = Pn;
            request(d, k - RD >= 0 ? k - RD : 0);
        };
        // full groups of RD steps (every step re-requests, clamped at the end: the same loads in flight on every path),
        // then the remainder
        int k = N - 1;
        for (int g = N / RD; g > 0; --g) {
#pragma unroll
            for (int d = 0; d < RD; ++d) step(d, k - d);
            k -= RD;
        }
#pragma unroll
        for (int d = 0; d < RD - 1; ++d)
            if (k - d >= 0) step(d, k - d);
    }
    if constexpr (PIPE && PW == 4) {
        if (factor_wave) {  // this wavefront's work is done: mark the factors that do not exist (per quad), like KEEP does
            if (notpd) {
                const int quad = (lane >> 2) & 3;
                int64_t pq = (int64_t)blockIdx.x * PW + quad;
                pq = pq < sa_.batch ? pq : sa_.batch - 1;
                const int nper_ = (ka.ep_on && ka.ep_periods > 1) ? ka.ep_periods : 1;
                if (per + 1 < nper_)
                    ((((per + 1) & 1) ? Fl1 : Fl0) + quad * (sa_.lds_problem / (int64_t)sizeof(double)))[FSI] = __builtin_nan("");
                else
                    (wsbase + pq * wl.total + wl.Fimg + (int64_t)(slot ^ 1) * N * FS)[FSI] = __builtin_nan("");
            }
            wsync();
            tick(10);
            return;
        }
    }
    notpd = __ballot(notpd) != 0ull;
    wsync();
    if constexpr (PIPE) {
        if (factor_wave) {  // this wavefront's work is done: mark a factor that does not exist, like KEEP does
            if (notpd && lane == 0) img_next[FSI] = __builtin_nan("");
            tick(10);
            return;
        }
    }
    if constexpr (SERIAL) {
        // the factor image travels between LDS and the workspace as it is (coalesced, one round trip); a factor that
        // does not exist is marked by a NaN in its first S^-1 so that a launch reusing it reports MPCQP_NOT_PD as well
        if (!reuse && notpd && lane == 0) Fl[FSI] = __builtin_nan("");
        if (!reuse && notpd) wsync();
        if (reuse) {
            // straight into LDS (global_load_lds_dwordx4: 1 KB per instruction, no staging registers), every request in
            // flight at once: ONE round trip for the image (a load -> LDS store loop paid one per turn of eight requests)
            typedef __attribute__((address_space(3))) void lds_void;
            typedef __attribute__((address_space(1))) const void glb_void;
            // (a later period of a multi-period launch that REUSES the factor finds the image where the first one put it:
            // nothing of a solve writes into it)
            if ((PIPE && !LDSIMG) || per == 0) {
                const D2 *src = (const D2 *)img;
                D2 *dst = (D2 *)Fl;
                const int n2 = N * FS / 2;  // (FS is even)
                const int nfull = n2 >> 6;
                for (int c = 0; c < nfull; ++c)
                    __builtin_amdgcn_global_load_lds((glb_void *)(src + c * 64 + lane), (lds_void *)(dst + c * 64), 16, 0, 0);
                if (nfull * 64 + lane < n2) dst[nfull * 64 + lane] = src[nfull * 64 + lane];
                wsync();
            }
            notpd = Fl[FSI] != Fl[FSI];
        } else if (keep) {
            for (int i = lane; i < N * FS; i += 64) img[i] = Fl[i];
        }
    }
    tick(1);
    // Short horizons run the two sweeps of an LQR solve SERIALLY (below): ~165 cycles per step with the vector spread
    // over the quads of a 16-lane row beat the chunked scans (and their 15 k cycles of prefix products) up to here.
    constexpr bool serial = SERIAL;  // (a template parameter: the scans' prefix matrices must not stay live here)
    // chunk transition matrix Phi_j = Acl_{k1-1} ... Acl_{k0} of this lane's chunk (identity if empty)
    double Phi[NX * NX];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) Phi[i] = (i / NX == i % NX) ? 1.0 : 0.0;
    for (int k = k0; k < (serial ? k0 : k1); ++k) {
        const double *Ac = Acl + wq(k) * NX * NX;
        double T[NX * NX];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a += Ac[i * NX + l] * Phi[l * NX + j];
                T[i * NX + j] = a;
            }
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) Phi[i] = T[i];
    }

    // ---- boundary scans in two levels. The 64 chunk boundaries obey x_{j+1} = Phi_j x_j + beta_j (forward) and
    // p_{j-1} = Phi_j' p_j + gamma_j (backward). The lanes form 8 groups of 8: (a) inside every group the offsets
    // are chained lane to lane (7 steps, all groups at once), (b) the 8 group inflows are chained through the
    // groups' total transition matrices (7 steps, readlane), (c) every lane gets its inflow from its group's inflow
    // and its neighbour's prefix. The prefix products Qf_j = Phi_j ... Phi_{8g}, Qb_j = Phi_j' ... Phi_{8g+7}' depend
    // on the problem only and stay in registers: 21 dependent small mat-vecs per scan instead of 63, no memory.
    double Qf[NX * NX], Qb[NX * NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            Qf[i * NX + j] = Phi[i * NX + j];
            Qb[i * NX + j] = Phi[j * NX + i];
        }
#pragma unroll 1
    for (int sdx = 1; sdx < (serial ? 1 : 8); ++sdx) {
        double Pf[NX * NX], Pb[NX * NX];
#pragma unroll
        for (int e = 0; e < NX * NX; ++e) {
            Pf[e] = __shfl_up(Qf[e], 1);
            Pb[e] = __shfl_down(Qb[e], 1);
        }
        const bool hf = (lane & 7) == sdx, hb = (lane & 7) == 7 - sdx;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double tf[NX], tb[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double a = 0.0, c = 0.0;
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    a += Phi[i * NX + q] * Pf[q * NX + j];  // Phi_j Qf_{j-1}
                    c += Phi[q * NX + i] * Pb[q * NX + j];  // Phi_j' Qb_{j+1}
                }
                tf[j] = a;
                tb[j] = c;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                Qf[i * NX + j] = hf ? tf[j] : Qf[i * NX + j];
                Qb[i * NX + j] = hb ? tb[j] : Qb[i * NX + j];
            }
        }
    }
    // inflow of every chunk for the forward recurrence, from the chunk offsets beta_j (in `off`) and x_s
    auto inflow_fwd = [&](double (&off)[NX], const double *xs, double (&xin)[NX]) {
#pragma unroll 1
        for (int sdx = 1; sdx < 8; ++sdx) {  // (a) off_j <- Phi_j off_{j-1} + beta_j inside the groups
            double o[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) o[i] = __shfl_up(off[i], 1);
            const bool h = (lane & 7) == sdx;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = off[i];
#pragma unroll
                for (int q = 0; q < NX; ++q) a += Phi[i * NX + q] * o[q];
                off[i] = h ? a : off[i];
            }
        }
        double xg[NX], cur[NX];  // (b) inflow of this lane's group
#pragma unroll
        for (int i = 0; i < NX; ++i) xg[i] = cur[i] = xs ? xs[i] : 0.0;
#pragma unroll 1
        for (int g = 0; g < 7; ++g) {
            const int src = 8 * g + 7;
            double nxt[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = rl(off[i], src);
#pragma unroll
                for (int q = 0; q < NX; ++q) a += rl(Qf[i * NX + q], src) * cur[q];
                nxt[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                cur[i] = nxt[i];
                xg[i] = ((lane >> 3) == g + 1) ? cur[i] : xg[i];
            }
        }
        double xo[NX];  // (c) outflow of this lane's chunk, handed to the next lane
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = off[i];
#pragma unroll
            for (int q = 0; q < NX; ++q) a += Qf[i * NX + q] * xg[q];
            xo[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double up = __shfl_up(xo[i], 1);
            xin[i] = ((lane & 7) == 0) ? xg[i] : up;
        }
    };
    // the same for the backward recurrence (gamma_j in `off`, terminal costate pN at the top of chunk 63)
    auto inflow_bwd = [&](double (&off)[NX], const double (&pN)[NX], double (&pin)[NX]) {
#pragma unroll 1
        for (int sdx = 1; sdx < 8; ++sdx) {
            double o[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) o[i] = __shfl_down(off[i], 1);
            const bool h = (lane & 7) == 7 - sdx;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = off[i];
#pragma unroll
                for (int q = 0; q < NX; ++q) a += Phi[q * NX + i] * o[q];
                off[i] = h ? a : off[i];
            }
        }
        double pg[NX], cur[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) pg[i] = cur[i] = pN[i];
#pragma unroll 1
        for (int g = 7; g >= 1; --g) {
            const int src = 8 * g;
            double nxt[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = rl(off[i], src);
#pragma unroll
                for (int q = 0; q < NX; ++q) a += rl(Qb[i * NX + q], src) * cur[q];
                nxt[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                cur[i] = nxt[i];
                pg[i] = ((lane >> 3) == g - 1) ? cur[i] : pg[i];
            }
        }
        double po[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = off[i];
#pragma unroll
            for (int q = 0; q < NX; ++q) a += Qb[i * NX + q] * pg[q];
            po[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double dn = __shfl_down(po[i], 1);
            pin[i] = ((lane & 7) == 7) ? pg[i] : dn;
        }
    };

    // ================================================================= the LQR solve as two chunked scans
    // linear costs: state cost q_k = qs * (row vector qrow) at k == kq plus the dense tracking term when
    // `track`; input cost r_k = rrow at k == kq.   (kq < 0: none)
    // backward: p_k = g_k + Acl_k' p_{k+1},  g_k = q_k - K_k' r_k ;  ff_k = -Sinv_k (B_k' p_{k+1} + r_k)
    auto lin_q = [&](int k, int kq, const double *qrow, bool track, double (&q)[NX]) {
#pragma unroll
        for (int i = 0; i < NX; ++i) q[i] = 0.0;
        if (track && stageQ && k >= 1) {
#pragma unroll
            for (int i = 0; i < NX; ++i) q[i] = -ka.wx * gtgt[(int64_t)k * NX + i];
        }
        if (k == kq) {
#pragma unroll
            for (int i = 0; i < NX; ++i) q[i] -= qrow[i];
        }
    };
    auto backward = [&](int kq, const double *qrow, const double *rrow, bool track) {
        // pass 1: chunk offset gamma_j (outflow at the chunk's bottom for a zero inflow at its top)
        double p[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) p[i] = 0.0;
        for (int k = k1 - 1; k >= k0; --k) {
            const double *Ac = Acl + wq(k) * NX * NX, *Kk = Kg + wq(k) * NU * NX;
            double q[NX], np_[NX];
            lin_q(k, kq, qrow, track, q);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = q[i];
#pragma unroll
                for (int l = 0; l < NX; ++l) a += Ac[l * NX + i] * p[l];
                if (k == kq) {
#pragma unroll
                    for (int l = 0; l < NU; ++l) a += Kk[l * NX + i] * rrow[l];  // - K' r with r = -rrow
                }
                np_[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) p[i] = np_[i];
        }
        // pass 2: inflow at the top of every chunk (empty chunks are identities)
        double pin[NX];
        {
            double pN[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) pN[i] = (track && termQ) ? -ka.wt * ggoal[i] : 0.0;
            inflow_bwd(p, pN, pin);
        }
        // pass 3: the chunk again from its true inflow; feed-forward terms written out
#pragma unroll
        for (int i = 0; i < NX; ++i) p[i] = pin[i];
        for (int k = k1 - 1; k >= k0; --k) {
            const double *Ac = Acl + wq(k) * NX * NX, *Kk = Kg + wq(k) * NU * NX, *Si = Sinv + wq(k) * NU * NU;
            const double *B = gB + k * sB;
            double t[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = (k == kq) ? -rrow[i] : 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a += B[l * NU + i] * p[l];
                t[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NU; ++l) a -= Si[i * NU + l] * t[l];
                ffv[wq(k) * NU + i] = a;
            }
            double q[NX], np_[NX];
            lin_q(k, kq, qrow, track, q);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = q[i];
#pragma unroll
                for (int l = 0; l < NX; ++l) a += Ac[l * NX + i] * p[l];
                if (k == kq) {
#pragma unroll
                    for (int l = 0; l < NU; ++l) a += Kk[l * NX + i] * rrow[l];
                }
                np_[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) p[i] = np_[i];
        }
    };
    // forward: u_k = -K_k x_k + ff_k, x_{k+1} = Acl_k x_k + B_k ff_k from x_0 = xs; writes Uo [N, NU], Xo [N, NX]
    auto forward = [&](const double *xs, double *Uo, double *Xo) {
        double y[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) y[i] = 0.0;
        for (int k = k0; k < k1; ++k) {
            const double *Ac = Acl + wq(k) * NX * NX, *B = gB + k * sB, *f = ffv + wq(k) * NU;
            double ny[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a += Ac[i * NX + l] * y[l];
#pragma unroll
                for (int l = 0; l < NU; ++l) a += B[i * NU + l] * f[l];
                ny[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) y[i] = ny[i];
        }
        double xin[NX];
        inflow_fwd(y, xs, xin);
        double x[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xin[i];
        for (int k = k0; k < k1; ++k) {
            const double *Ac = Acl + wq(k) * NX * NX, *Kk = Kg + wq(k) * NU * NX, *B = gB + k * sB;
            const double *f = ffv + wq(k) * NU;
#pragma unroll
            for (int i = 0; i < NX; ++i) Xo[wq(k) * NX + i] = x[i];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = f[i];
#pragma unroll
                for (int l = 0; l < NX; ++l) a -= Kk[i * NX + l] * x[l];
                Uo[wq(k) * NU + i] = a;
            }
            double nx_[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a += Ac[i * NX + l] * x[l];
#pragma unroll
                for (int l = 0; l < NU; ++l) a += B[i * NU + l] * f[l];
                nx_[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = nx_[i];
        }
    };
    // ---- the same two sweeps, serial in k (N <= kSerialMaxN), on the matrix cores (round 5). The running vector sits in B-operand
    // order -- component sq in the lanes of 16-lane row sq, which is also where a result lands --, the step's matrix is fetched
    // from the LDS image one ELEMENT per lane in A-operand order, the affine term enters as the C operand: a step of the serial
    // chain is ONE v_mfma_f64_4x4x4 whose result is the next step's operand (28 cycles from result to dependent product,
    // tools/ubench/mfma_f64_4x4.hip); a second product off the chain gives the feed-forward term (backward) or the input
    // (forward). The operands are requested STAGE_SRD steps ahead through running pointers (no clamping: the image has that
    // many steps of slack on either side); S^-1 is folded into the feed-forward rows by the factor; nothing is stored under an
    // exec mask. ~12 instructions per step. Round 4: component q in quad q of a row, eight dependent v_fmac_f64_dpp per step,
    // 22 instructions, ~165 cycles.
    constexpr int SRD = STAGE_SRD;  // request distance of the serial sweeps, in steps (their operands sit in LDS: ~100+ cycles, a step is ~80)
    // A value that only one lane (or one lane per quad) has to store is stored by EVERY lane, each to an address of its own:
    // the wanted lanes walk the array, the others hit their cell of `junkl` (a masked store costs the wavefront two exec
    // writes and a branch in the middle of a ~20-instruction step).
    double *junkl = tgl + N * NX + STAGE_SRD * FS + lane * NU;
    // TRACK: the sweep of the unconstrained minimiser (terminal costate, targets); otherwise the costate of ONE row (kq, crow, drow):
    // zero above its step, the row itself at its step -- that step is formed directly, the loop starts below it
    auto backward_s = [&](int kq, const double *crow, const double *drow, auto trackc) {
        constexpr bool track = decltype(trackc)::value;
        double own;
        int kstart;
        if constexpr (track) {
            own = (termQ && sqin) ? -ka.wt * goalown : 0.0;
            kstart = N - 1;
            // the targets come through LDS, already scaled (-w_x target; step 0 carries no state cost), and the step starts its
            // sum from them (they were requested at the top of the period)
            const double mwx = -ka.wx;
#pragma unroll
            for (int u = 0; u < TGV; ++u)
                if (lane + 64 * u < N * NX) tgl[lane + 64 * u] = (lane + 64 * u >= NX) ? mwx * tgv[u] : 0.0;
            lsync();
        } else {
            const double *f = Fl + kq * FS;
            double pn = 0.0;
            if (crow && sqin) pn -= crow[sq];
            double dv[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                dv[i] = drow ? drow[i] : 0.0;
                pn -= f[FKN + sq * NU + i] * dv[i];  // - K' r with r = -D row   (FKN holds -K[i][q])
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    double a = 0.0;
#pragma unroll
                    for (int l = 0; l < NU; ++l) a += f[FSI + i * NU + l] * dv[l];
                    ffl[kq * NU + i] = a;
                }
            }
            // (the feed-forward terms above the row's step count as zero in the forward sweep)
            for (int i = (kq + 1) * NU + lane; i < N * NU; i += 64) ffl[i] = 0.0;
            own = sqin ? pn : 0.0;
            kstart = kq - 1;
        }
        // element [sq][sc] of the step's matrix, as the matrix cores take an A operand (A[i][k] in lane i + 16 k): Acl[r][c] read that
        // way is Acl' (p_k = Acl' p_{k+1}). The step stores its operand p_{k+1} into cell k + 1 of the target array (that target
        // was consumed a step earlier; cell N lies in the slack behind the array), one step late -- nothing waits for a product.
        double at[SRD], tg[SRD];
        const double *fa = Fl + FA + sq * 4 + sc + kstart * FS;
        const double *tq = tgl + ((NX == 4 || sqin) ? sq : 0) + kstart * NX;
        double *pw = (NX == 4 || sqin) ? tgl + (kstart + 1) * NX + sq : junkl;  // cell of p_{kstart+1}
        const int pws = (NX == 4 || sqin) ? NX : 0;
        auto req = [&](int d, int off, int offt) {
            at[d] = fa[off];
            if constexpr (track) tg[d] = tq[offt];
        };
#pragma unroll
        for (int d = 0; d < SRD; ++d) {
            req(d, -d * FS, -d * NX);
            __builtin_amdgcn_sched_barrier(0);
        }
        fa -= SRD * FS;  // (the pointers run SRD steps ahead of the step that computes)
        tq -= SRD * NX;
        auto step = [&](int d, bool again) {
            const double c0 = track ? ((NX == 4 || sqin) ? tg[d] : 0.0) : 0.0;
            *pw = own;
            pw -= pws;
            own = mm44(at[d], own, c0);  // p_k = (-w_x target_k) + Acl' p_{k+1}: ONE product per step
            if (again) req(d, 0, 0);
            fa -= FS;
            tq -= NX;
        };
        int k = kstart;
        for (int g = (kstart + 1) / SRD; g > 0; --g) {
#pragma unroll
            for (int d = 0; d < SRD; ++d) step(d, true);
            k -= SRD;
        }
#pragma unroll
        for (int d = 0; d < SRD - 1; ++d)
            if (k - d >= 0) step(d, false);
        lsync();
        // the feed-forward terms ff_k = -(S^-1 B') p_{k+1} of all the steps at once, lane <-> step (they were a second product in
        // every step of the chain; a lone wavefront pays ~16 cycles of issue for each)
        for (int kk = lane; kk <= kstart; kk += 64) {
            const double *f = Fl + kk * FS + FBS, *pc = tgl + (kk + 1) * NX;
            double a[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) a[u] = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const double pj = pc[j];
#pragma unroll
                for (int u = 0; u < NU; ++u) a[u] += f[j * NU + u] * pj;
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) ffl[kk * NU + u] = a[u];
        }
    };
    // The trajectory is staged in LDS while the sweep runs (x_k over the targets, u_k over the feed-forward term it was formed
    // from: both are dead by then); the lanes copy their chunks to the workspace arrays (Uo, Xo) in one coalesced pass
    // afterwards. (The feed-forward terms above a single row's step were zeroed by its backward sweep.)
    double *xl = tgl, *ul = ffl;
    auto forward_s = [&](double x0q, double *Uo, double *Xo) {  // x0q: this lane's component of the initial state
        double own = x0q;
        // c_k = B_k ff_k of all the steps at once (lane <-> step) into cell k + 1 of the trajectory array, where the step that forms
        // x_{k+1} = Acl x_k + c_k reads it as its C operand and a step later x_{k+1} itself lands (x_k is stored one step late)
        lsync();
        for (int kk = lane; kk < N; kk += 64) {
            const double *f = Fl + kk * FS + FBO;
            double fk[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) fk[u] = ffl[kk * NU + u];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double a = 0.0;
#pragma unroll
                for (int u = 0; u < NU; ++u) a += f[j * NU + u] * fk[u];
                xl[(kk + 1) * NX + j] = a;
            }
        }
        lsync();
        // A operand by element: Acl[c][r] -> x_{k+1} = Acl x_k + c_k: ONE product per step
        double ar[SRD], cx[SRD];
        const double *fa = Fl + FA + sc * 4 + sq, *cq = xl + NX + ((NX == 4 || sqin) ? sq : 0);
        double *xw = (NX == 4 || sqin) ? xl + sq : junkl;
        const int xws = (NX == 4 || sqin) ? NX : 0;
        auto req = [&](int d, int off, int offc) {
            ar[d] = fa[off];
            cx[d] = cq[offc];
        };
        auto run = [&]() {
#pragma unroll
        for (int d = 0; d < SRD; ++d) {
            req(d, d * FS, d * NX);
            __builtin_amdgcn_sched_barrier(0);
        }
        fa += SRD * FS;
        cq += SRD * NX;
        auto step = [&](int d, bool again) {
            *xw = own;
            xw += xws;
            const double xn = mm44(ar[d], own, (NX == 4 || sqin) ? cx[d] : 0.0);
            own = (NX == 4 || sqin) ? xn : 0.0;
            if (again) req(d, 0, 0);
            fa += FS;
            cq += NX;
        };
        int k = 0;
        for (int g = N / SRD; g > 0; --g) {
#pragma unroll
            for (int d = 0; d < SRD; ++d) step(d, true);
            k += SRD;
        }
#pragma unroll
        for (int d = 0; d < SRD - 1; ++d)
            if (k + d < N) step(d, false);
        };
        run();
        lsync();
        // the inputs u_k = ff_k - K_k x_k of all the steps at once (over the feed-forward terms they are formed from)
        for (int kk = lane; kk < N; kk += 64) {
            const double *f = Fl + kk * FS + FKN, *xc = xl + kk * NX;
            double a[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) a[u] = ffl[kk * NU + u];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const double xj = xc[j];
#pragma unroll
                for (int u = 0; u < NU; ++u) a[u] += f[j * NU + u] * xj;  // (FKN holds -K[u][j] at j NU + u)
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) ul[kk * NU + u] = a[u];
        }
        lsync();
        for (int kk = k0; kk < k1; ++kk) {
#pragma unroll
            for (int i = 0; i < NU; ++i) Uo[wq(kk) * NU + i] = ul[kk * NU + i];
#pragma unroll
            for (int i = 0; i < NX; ++i) Xo[wq(kk) * NX + i] = xl[kk * NX + i];
        }
    };
    // g_(k,r) . (U, X) = C_k[r] x_k + D_k[r] u_k
    auto gdot = [&](int k, int64_t w, int r, const double *Uv, const double *Xv) {  // w = workspace index of step k
        double a = 0.0;
        if (gC) {
            const double *c = gC + k * sC + r * NX;
#pragma unroll
            for (int i = 0; i < NX; ++i) a += c[i] * Xv[w * NX + i];
        }
        if (gD) {
            const double *d = gD + k * sD + r * NU;
#pragma unroll
            for (int i = 0; i < NU; ++i) a += d[i] * Uv[w * NU + i];
        }
        return a;
    };

    // ================================================================= unconstrained minimiser, slacks
    double zero_row[NX > NU ? NX : NU];
#pragma unroll
    for (int i = 0; i < (NX > NU ? NX : NU); ++i) zero_row[i] = 0.0;
    tick(2);
    if (!notpd) {
        if constexpr (serial)
            backward_s(-1, nullptr, nullptr, std::true_type{});
        else
            backward(-1, zero_row, zero_row, true);
    }
    if constexpr (serial)
        lsync();  // (the feed-forward terms go from lane to lane through LDS: nothing to wait for)
    else
        wsync();
    tick(3);
    if (!notpd) {
        if constexpr (serial)
            forward_s(x0own, U0, X0);
        else
            forward(gx0, U0, X0);
    }
    if constexpr (!serial) wsync();  // (serial: the slack pass below reads the trajectory where the sweep staged it, in LDS)
    tick(4);
    const double tol = ka.tol;
    // The same pass makes the FIRST selection (the values are in registers: the loop's own selection pass would wait for
    // them to come back from the workspace) and keeps the unconstrained inputs of the lane's first step: a problem none of
    // whose rows is violated there -- most periods of a well-conditioned loop -- is finished after this pass.
    double best = INF, u_first[NU];
    int bi = 0x7fffffff;
    bool presel = SERIAL;  // (the long-horizon instantiations have no registers to spare for it: 254 VGPRs)
#pragma unroll
    for (int i = 0; i < NU; ++i) u_first[i] = 0.0;
    // (SERIAL: the trajectory is still staged in LDS, in step order -- no wait for the workspace copy to come back)
    const double *Us = SERIAL ? ul : U0, *Xs = SERIAL ? xl : X0;
    for (int k = k0; k < (notpd ? k0 : k1); ++k) {
        const int64_t wk = SERIAL ? (int64_t)k : wq(k);
        if (SERIAL && k == k0)
#pragma unroll
            for (int i = 0; i < NU; ++i) u_first[i] = Us[wk * NU + i];
        for (int r = 0; r < mk; ++r) {
            const int64_t i = wq(k) * mk + r;
            const double ev = ge[k * sE + r];
            const double sv = ev - gdot(k, wk, r, Us, Xs);
            sl[i] = sv;
            double nn = 0.0;
            if (gC)
                for (int c = 0; c < NX; ++c) nn += gC[k * sC + r * NX + c] * gC[k * sC + r * NX + c];
            if (gD)
                for (int c = 0; c < NU; ++c) nn += gD[k * sD + r * NU + c] * gD[k * sD + r * NU + c];
            const double in = nn > 0.0 ? rsqrt(nn) : 1.0;
            invn[i] = in;
            rowslot[i] = -1;
            if constexpr (SERIAL) {
                const bool viol = ev < 1e29 && sv < -(tol + tol * fabs(ev));
                const double sc = sv * in;
                if (viol && sc < best) {
                    best = sc;
                    bi = k * mk + r;
                }
            }
        }
    }
    // (the slacks, norms and slot table just stored are read by other lanes from the first iteration on. SERIAL cold start: the
    // first selection comes from registers, so the wait for those stores is taken only if a row IS violated -- below)
    constexpr bool late_fence = SERIAL && !WARM;
    if constexpr (!late_fence) wsync();

    tick(5);
    // ================================================================= active-set loop
    int nq = 0, iters = 0, status = MPCQP_MAX_ITER;
    // SERIAL cold start: whether ANY row is violated at the unconstrained minimiser is decided here, before the active-set loop's
    // set-up code (its loop-invariant addresses, divisions and scalar spills: ~3 k cycles that most periods of a
    // well-conditioned receding-horizon loop do not need)
    bool early_plan = false;
    if constexpr (late_fence) {
        if (!notpd) {
            wave_argmin(best, bi);  // (the loop's own reduction of the reduced pair changes nothing)
            early_plan = !(best < INF);
        }
    }
    const int max_iter = ka.max_iter;
    const int nvar = N * NU;
    double *Vp = Vs + (int64_t)maxq * NP * NU, *Xp = XVs + (int64_t)maxq * NP * NX;  // the candidate's slot
    bool fail = false, slotsfull = false, unconstrained = false;
    // slot l leaves the active set: W is deflated and the last slot moves into the hole (nq is decremented by the caller)
    auto drop_slot = [&](int l) {
        const double wll = Wm[(int64_t)l * maxq + l];
        const double iw = 1.0 / wll;
        for (int a = lane; a < nq; a += 64) cv[a] = Wm[(int64_t)l * maxq + a];  // row l before the update
        wsync();
        for (int a = lane; a < nq; a += 64) {
            const double wa = cv[a];
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
            const double *vs = Vs + (int64_t)last * NP * NU, *xs = XVs + (int64_t)last * NP * NX;
            double *vd = Vs + (int64_t)l * NP * NU, *xd = XVs + (int64_t)l * NP * NX;
            for (int k = k0; k < k1; ++k) {
#pragma unroll
                for (int i = 0; i < NU; ++i) vd[wq(k) * NU + i] = vs[wq(k) * NU + i];
#pragma unroll
                for (int i = 0; i < NX; ++i) xd[wq(k) * NX + i] = xs[wq(k) * NX + i];
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
    };
    // the primal point of the current multipliers, u = u0 - sum_a lam_a V_a (x likewise), and every row's slack from scratch:
    // inactive rows get their fresh slack (dirty: one of them is infeasible beyond 4 tol), active rows are set to zero after
    // checking that they really sit on their bounds (offa: W is only ever updated, never refactored -- and a warm start takes
    // it from a previous launch); with `write_u` the inputs go to the output
    // (an active row may be off by (1 + |e_i|) max(1e3 tol, 1e-6): the contract's 1e-6 -- an ill-conditioned but legitimate
    // plan leaves ~1e-9 here, no refinement step in this kernel)
    const double kacc = 1e3 > 1e-6 / tol ? 1e3 : 1e-6 / tol;
    bool rough = false;  // (set by eval_point) an active row is further than 1e3 tol (1 + |e|) from its bound
    auto eval_point = [&](bool write_u, bool &dirty, bool &offa) {
        rough = false;
        for (int k = k0; k < k1; ++k) {
            double u[NU], x[NX];
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = U0[wq(k) * NU + i];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = X0[wq(k) * NX + i];
            for (int a = 0; a < nq; ++a) {
                const double la = lamv[a];
                const double *va = Vs + ((int64_t)a * NP + wq(k)) * NU, *xa = XVs + ((int64_t)a * NP + wq(k)) * NX;
#pragma unroll
                for (int i = 0; i < NU; ++i) u[i] -= la * va[i];
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] -= la * xa[i];
            }
            if (write_u) {
                double *ou = (double *)ka.U + prob * (int64_t)nvar + (int64_t)k * NU;
#pragma unroll
                for (int i = 0; i < NU; ++i) ou[i] = u[i];
                if (SERIAL && k == k0)
#pragma unroll
                    for (int i = 0; i < NU; ++i) u_first[i] = u[i];
            }
            for (int r = 0; r < mk; ++r) {
                const int64_t i = wq(k) * mk + r;
                const double ev = ge[k * sE + r];
                double g = 0.0;
                if (gC)
#pragma unroll
                    for (int c = 0; c < NX; ++c) g += gC[k * sC + r * NX + c] * x[c];
                if (gD)
#pragma unroll
                    for (int c = 0; c < NU; ++c) g += gD[k * sD + r * NU + c] * u[c];
                const double fresh = ev - g, th = tol + tol * fabs(ev);
                const int rsl = rowslot[i];
                const bool act = rsl >= 0;
                if (ev < 1e29 && !act && !(fresh >= -4.0 * th)) dirty = true;
                if (act && !(fabs(fresh) <= kacc * th)) offa = true;
                if (act) {  // the active rows' residuals, for the refinement step of the final evaluation
                    cv[rsl] = fresh;
                    if (!(fabs(fresh) <= 1e3 * th)) rough = true;
                }
                sl[i] = act ? 0.0 : fresh;
            }
        }
        dirty = __ballot(dirty) != 0ull;
        offa = __ballot(offa) != 0ull;
        rough = __ballot(rough) != 0ull;
        wsync();
    };
    // back to the empty active set at the unconstrained minimiser
    auto cold_start = [&]() {
        for (int a = lane; a < nq; a += 64) rowslot[wg(actk[a]) * mk + actr[a]] = -1;
        wsync();
        nq = 0;
        bool d = false, o = false;
        eval_point(false, d, o);
    };
    // ---- warm start (MpcqpSolveOpts.warm_state / warm_start with MPCQP_OPT_REUSE_FACTOR, include/mpcqp.h): the previous
    // launch left its active rows' vectors V_a, X_a and W = (G_A P^-1 G_A')^-1 in this workspace and the rows' ids in the
    // warm-state record. They depend on the matrices only, so for new x0 / goal / targets / e the multipliers are
    // lam = -W s0_A; rows whose multiplier comes out negative leave (one deflation each), the rest is the starting active
    // set: a period whose active set did not move costs no sweep at all.
    const int wrec = (int)(stage_warm_bytes(maxq) / sizeof(int));  // ints per problem: nq, maxq, workspace tag (2), row steps, row indices
    int *wst = (WARM && ka.warm_state) ? (int *)ka.warm_state + prob * (int64_t)wrec : nullptr;
    const unsigned long long wtag = (unsigned long long)(uintptr_t)ws;
    if constexpr (WARM) if (wst && ka.warm_start && reuse && !notpd) {
        presel = false;  // (the slacks move with the stored active set)
        int nqs = wst[0];
        const bool same = wst[1] == maxq && (unsigned)wst[2] == (unsigned)wtag && (unsigned)wst[3] == (unsigned)(wtag >> 32);
        if (!same || nqs < 0 || nqs > maxq) nqs = 0;
        bool bad = false;
        for (int a = lane; a < nqs; a += 64) {
            const int k = wst[4 + a], r = wst[4 + maxq + a];
            actk[a] = k;
            actr[a] = r;
            bad |= k < 0 || k >= N || r < 0 || r >= mk;
        }
        if (__ballot(bad) != 0ull) nqs = 0;
        wsync();
        for (int a = lane; a < nqs; a += 64) rowslot[wg(actk[a]) * mk + actr[a]] = a;
        wsync();
        bad = false;
        for (int a = lane; a < nqs; a += 64) bad |= rowslot[wg(actk[a]) * mk + actr[a]] != a;  // (a row listed twice)
        nq = nqs;
        if (__ballot(bad) != 0ull) cold_start();
        while (nq > 0) {
            for (int a = lane; a < nq; a += 64) cv[a] = sl[wg(actk[a]) * mk + actr[a]];  // slacks at the unconstrained minimiser
            wsync();
            double lmin = INF;
            int l = 0x7fffffff;
            bool nan = false;
            for (int a = lane; a < nq; a += 64) {
                double acc = 0.0;
                for (int b = 0; b < nq; ++b) acc += Wm[(int64_t)b * maxq + a] * cv[b];
                lamv[a] = -acc;
                nan |= !(acc == acc) || fabs(acc) > 1e300;
                if (-acc < lmin) {
                    lmin = -acc;
                    l = a;
                }
            }
            if (__ballot(nan) != 0ull) {  // (not a state this kernel left: start cold)
                cold_start();
                break;
            }
            wave_argmin(lmin, l);
            wsync();
            if (!(lmin < 0.0)) break;
            drop_slot(l);
            --nq;
            ++iters;
            wsync();
        }
        if (nq > 0) {
            bool d = false, o = false;
            eval_point(false, d, o);
            if (o) cold_start();  // the stored rows do not sit on their bounds with these vectors: not this problem's state
        }
    }
    for (int round = 0; round < 4 && !fail && !notpd && !early_plan; ++round) {
        for (;;) {
            // ---- selection: the violated row farthest from its hyperplane (the first one was made by the slack pass)
            if (!presel) {
                best = INF;
                bi = 0x7fffffff;
            }
            for (int k = k0; k < (presel ? k0 : k1); ++k)
                for (int r = 0; r < mk; ++r) {
                    const int64_t i = wq(k) * mk + r;
                    const double ev = ge[k * sE + r], sv = sl[i];
                    const bool viol = ev < 1e29 && rowslot[i] < 0 && sv < -(tol + tol * fabs(ev));
                    const double sc = sv * invn[i];
                    if (viol && sc < best) {
                        best = sc;
                        bi = k * mk + r;  // natural row id: ties go to the lowest one, like the restatement
                    }
                }
            const bool first_sel = presel;
            presel = false;
            wave_argmin(best, bi);
            if (!(best < INF)) {
                status = MPCQP_SOLVED;
                if (first_sel) unconstrained = true;  // nothing was violated at the unconstrained minimiser: it is the plan
                break;
            }
            if constexpr (late_fence)
                if (first_sel) wsync();
            const int kp = bi / mk, rp = bi - kp * mk;
            const int64_t wp = wg(kp), bw = wp * mk + rp;  // workspace index of step kp / of row p
            double qrow[NX], rrow[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) qrow[i] = gC ? gC[kp * sC + rp * NX + i] : 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) rrow[i] = gD ? gD[kp * sD + rp * NU + i] : 0.0;
            double up = 0.0;
            bool added = false;
            // V_p = P^-1 g_p' and its trajectory do not change while p waits for room: solved once
            tacc(-1);
            if constexpr (serial)
                backward_s(kp, gC ? gC + kp * sC + rp * NX : nullptr, gD ? gD + kp * sD + rp * NU : nullptr, std::false_type{});
            else
                backward(kp, qrow, rrow, false);
            if constexpr (serial)
                lsync();
            else
                wsync();
            if constexpr (serial)
                forward_s(0.0, Vp, Xp);
            else
                forward(nullptr, Vp, Xp);
            // SERIAL: the candidate's vectors are read where the sweep staged them, in LDS (step order) -- no wait for the workspace
            // copy, which only the slot copy of a row that becomes active reads (each lane its own chunk)
            const double *Vc = SERIAL ? ul : Vp, *Xc = SERIAL ? xl : Xp;
            if constexpr (!serial) wsync();
            tacc(9);  // sweeps
            const double dpp = gdot(kp, SERIAL ? (int64_t)kp : wp, rp, Vc, Xc);
            while (!added) {
                if (iters >= max_iter) {
                    fail = true;
                    break;
                }
                ++iters;
                // ---- c_a = g_a . V_p ; r = W c ; d2 = g_p . V_p - c . r
                for (int a = lane; a < nq; a += 64) cv[a] = gdot(actk[a], SERIAL ? (int64_t)actk[a] : wg(actk[a]), actr[a], Vc, Xc);
                if constexpr (serial)
                    lsync();  // (c goes from lane to lane through LDS)
                else
                    wsync();
                double cr = 0.0;
                for (int a = lane; a < nq; a += 64) {
                    double acc = 0.0;
                    for (int b = 0; b < nq; ++b) acc += Wm[(int64_t)b * maxq + a] * cv[b];  // W is symmetric
                    rv[a] = acc;
                    cr += acc * cv[a];
                }
                cr = wave_sum(cr);
                if constexpr (serial)
                    lsync();  // (r likewise)
                else
                    wsync();
                tacc(10);  // c, r = W c
                const double d2 = dpp - cr;
                const bool can_move = (nq < nvar) && (d2 > 1e-13 * dpp) && (d2 > 0.0);
                // ---- ratio test on the multipliers
                double t1 = INF;
                int l = 0x7fffffff;
                for (int a = lane; a < nq; a += 64) {
                    const double ra = rv[a];
                    if (ra > 0.0) {
                        const double q = lamv[a] / ra;
                        if (q < t1) {
                            t1 = q;
                            l = a;
                        }
                    }
                }
                wave_argmin(t1, l);
                const double sp = sl[bw];
                const double t2 = can_move ? -sp / d2 : INF;
                const double t = t1 < t2 ? t1 : t2;
                if (!(t < INF)) {
                    status = MPCQP_INFEASIBLE;
                    fail = true;
                    break;
                }
                const bool full = (t2 <= t1);
                if (full && nq >= maxq) {  // the row would enter, but every slot is taken (max_active < min(n, m)): a
                    fail = true;           // drop can go on with full slots, an addition cannot -> MPCQP_SLOTS_FULL
                    slotsfull = true;
                    break;
                }
                tacc(12);  // ratio test, step
                // ---- slacks: s_i -= t g_i . z ,  z = -(V_p - sum_a r_a V_a)  (this lane's chunk)
                for (int k = k0; k < k1; ++k) {
                    double zu[NU], zx[NX];
#pragma unroll
                    for (int i = 0; i < NU; ++i) zu[i] = -Vc[(SERIAL ? (int64_t)k : wq(k)) * NU + i];
#pragma unroll
                    for (int i = 0; i < NX; ++i) zx[i] = -Xc[(SERIAL ? (int64_t)k : wq(k)) * NX + i];
                    for (int a = 0; a < nq; ++a) {
                        const double ra = rv[a];
                        const double *va = Vs + ((int64_t)a * NP + wq(k)) * NU, *xa = XVs + ((int64_t)a * NP + wq(k)) * NX;
#pragma unroll
                        for (int i = 0; i < NU; ++i) zu[i] += ra * va[i];
#pragma unroll
                        for (int i = 0; i < NX; ++i) zx[i] += ra * xa[i];
                    }
                    for (int r = 0; r < mk; ++r) {
                        double gz = 0.0;
                        if (gC)
#pragma unroll
                            for (int i = 0; i < NX; ++i) gz += gC[k * sC + r * NX + i] * zx[i];
                        if (gD)
#pragma unroll
                            for (int i = 0; i < NU; ++i) gz += gD[k * sD + r * NU + i] * zu[i];
                        const int64_t i = wq(k) * mk + r;
                        sl[i] = (rowslot[i] >= 0) ? 0.0 : sl[i] - t * gz;
                    }
                }
                // ---- multipliers
                for (int a = lane; a < nq; a += 64) {
                    const double v = lamv[a] - t * rv[a];
                    lamv[a] = v < 0.0 ? 0.0 : v;
                }
                up += t;
                wsync();
                tacc(13);  // slack update
                if (full) {
                    // p becomes active in slot nq: W is bordered, the candidate's vectors move into the slot
                    const double id2 = 1.0 / d2;
                    for (int a = lane; a < nq; a += 64) {
                        const double ra = rv[a];
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
                    double *vd = Vs + (int64_t)nq * NP * NU, *xd = XVs + (int64_t)nq * NP * NX;
                    for (int k = k0; k < k1; ++k) {
#pragma unroll
                        for (int i = 0; i < NU; ++i) vd[wq(k) * NU + i] = Vp[wq(k) * NU + i];
#pragma unroll
                        for (int i = 0; i < NX; ++i) xd[wq(k) * NX + i] = Xp[wq(k) * NX + i];
                    }
                    ++nq;
                    added = true;
                } else {
                    // partial step: slot l leaves
                    drop_slot(l);
                    --nq;
                }
                wsync();
                tacc(-1);  // (W update, slot copy: the rest)
            }
            if (fail) break;
        }
        if (fail) break;
        tick(6);
        // ================================================================= primal point, verification
        // u = u0 - sum_a lam_a V_a ; slacks from scratch through x = x0 - sum_a lam_a X_a
        bool dirty = false, offa = false;
        if (unconstrained) {
            // the slack pass evaluated exactly this point (no active row, u = u0): only the inputs are left to write
            for (int k = k0; k < k1; ++k) {
                double *ou = (double *)ka.U + prob * (int64_t)nvar + (int64_t)k * NU;
#pragma unroll
                for (int i = 0; i < NU; ++i) ou[i] = k == k0 ? u_first[i] : U0[wq(k) * NU + i];
            }
        } else {
            eval_point(true, dirty, offa);
            if (rough && nq > 0) {
                // W is only ever updated: after a hundred iterations the active rows drift off their bounds (1e-8 .. 1e-6). One
                // step of refinement, lam -= W rho_A with the residuals the evaluation just left in cv, and the evaluation again.
                for (int a = lane; a < nq; a += 64) {
                    double acc = 0.0;
                    for (int b = 0; b < nq; ++b) acc += Wm[(int64_t)b * maxq + a] * cv[b];  // W is symmetric
                    rv[a] = acc;
                }
                wsync();
                for (int a = lane; a < nq; a += 64) {
                    const double v = lamv[a] - rv[a];
                    lamv[a] = v < 0.0 ? 0.0 : v;
                }
                wsync();
                dirty = offa = false;
                eval_point(true, dirty, offa);
            }
        }
        if (offa) {  // an active row is off its bound (W drifted, or a warm state that was not this problem's): start cold
            cold_start();
            status = MPCQP_MAX_ITER;
            continue;
        }
        if (!dirty) {
            status = MPCQP_SOLVED;
            break;
        }
        status = MPCQP_MAX_ITER;  // continue from the re-evaluated slacks
    }
    if (early_plan) {
        // the slack pass evaluated exactly this point (no active row, u = u0): it is the plan, only the inputs are left to write
        status = MPCQP_SOLVED;
        tick(6);
        for (int k = k0; k < k1; ++k) {
            double *ou = (double *)ka.U + prob * (int64_t)(N * NU) + (int64_t)k * NU;
#pragma unroll
            for (int i = 0; i < NU; ++i) ou[i] = k == k0 ? u_first[i] : U0[wq(k) * NU + i];
        }
    }
    tick(7);
    if (fail && status == MPCQP_SOLVED) status = MPCQP_MAX_ITER;
    if (slotsfull) status = MPCQP_SLOTS_FULL;
    if (notpd) status = MPCQP_NOT_PD;
    const bool ok = status == MPCQP_SOLVED;
    if (!ok) {
        double *ou = (double *)ka.U + prob * (int64_t)nvar;
        for (int k = k0; k < k1; ++k)
#pragma unroll
            for (int i = 0; i < NU; ++i) ou[(int64_t)k * NU + i] = 0.0;
    }
    if (ka.lam) {
        double *ol = (double *)ka.lam + prob * (int64_t)N * mk;
        for (int k = k0; k < k1; ++k)
            for (int r = 0; r < mk; ++r) {
                const int sidx = (ok && nq > 0) ? rowslot[wq(k) * mk + r] : -1;
                ol[(int64_t)k * mk + r] = sidx >= 0 ? lamv[sidx] : 0.0;
            }
    }
    if constexpr (WARM) if (wst) {  // the warm-state record of the next launch: the active rows (their vectors and W stay in the workspace)
        for (int a = lane; a < (ok ? nq : 0); a += 64) {
            wst[4 + a] = actk[a];
            wst[4 + maxq + a] = actr[a];
        }
        if (lane == 0) {
            wst[0] = ok ? nq : 0;
            wst[1] = maxq;
            wst[2] = (int)(unsigned)wtag;
            wst[3] = (int)(unsigned)(wtag >> 32);
        }
    }
    if (lane == 0) {
        if (ka.status) ka.status[prob] = status;
        if (ka.iters) ka.iters[prob] = iters;
    }
    if (ka.ep_on) {
        // the rest of the control period: plant step with the plan's first input (zero if there is no plan), then the
        // loop's next problem written over the one just solved (this wavefront is its only reader)
        // (the plan's first input is lane 0's: step 0 is the first step of its chunk)
        double a = 0.0;
        if constexpr (SERIAL) {
            a = ok ? __shfl(u_first[0], 0) : 0.0;
        } else {
            wsync();
            a = ok ? ((const double *)ka.U)[prob * (int64_t)N * NU] : 0.0;
        }
        wip_period_wave<double>(lane, (double *)ka.ep_states + prob * 4, ep_s0, a, N, ka.ep_Tp, ka.ep_vel, ka.ep_omega2, ka.ep_g,
                                ka.ep_nsub, const_cast<double *>(gx0), const_cast<double *>(ggoal), const_cast<double *>(gtgt), carry);
        if (lane == 0 && ka.ep_loopstats) {
            // (atomics without a return value: fire and forget -- a load + add + store is a dependent round trip on the period's tail;
            // this wavefront is the only writer of its loop's counters)
            if (!ok) __hip_atomic_fetch_add(ka.ep_loopstats + 2 * prob, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (iters) __hip_atomic_fetch_add(ka.ep_loopstats + 2 * prob + 1, (long long)iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    tick(8);
    };  // period
    if constexpr (!(SERIAL && NX == 4 && NU == 1)) {  // (one period per launch: mpcqp_wip_periods_batch refuses more)
        constexpr int PW1 = PIPE ? PWT : 1;
        if ((int64_t)blockIdx.x * PW1 + (((int)threadIdx.x >> 6) % PW1) < sa_.batch) period(0);
        return;
    }
    const int nper = (ka_.ep_on && ka_.ep_periods > 1) ? ka_.ep_periods : 1;
    constexpr int PWo = PIPE ? PWT : 1;
    const int wvo = (int)threadIdx.x >> 6;
    const int64_t probo = (int64_t)blockIdx.x * PWo + (wvo % PWo);
    const bool valid = probo < sa_.batch;  // (the last workgroup of a pipelined launch may hold fewer than PW problems)
    const bool stamper = ka_.probe && valid && (threadIdx.x & 63) == 0 && wvo < PWo;
    for (int per = 0; per < nper; ++per) {
        if (valid) period(per);
        if (per + 1 < nper) {
            if (stamper)  // (developer probe: slot 13 = the period's end, slot 12 = the end of the hand-over)
                ((long long *)ka_.probe)[probo * 16 + 13] = (long long)__builtin_readcyclecounter();
            // the next period's x0 / goal / targets / state were written by this wavefront, its factor image by the other
            // one (PIPE): same CU, same L1 -- the stores have to be complete, nothing has to be invalidated but the
            // scalar cache
            if constexpr (PIPE && PWT == 4) {
                // four loops per workgroup: the next period reads NOTHING this one stored to global memory -- its factor is in the
                // other LDS image, its problem comes from the carried plant state, the factor wavefront's operands from their LDS
                // copy, and a wavefront's own workspace arrays are written before they are read in every period -- so only the LDS
                // writes have to be complete at the barrier: the global stores (plan, next problem, counters) drain behind it
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                wsync();
                __builtin_amdgcn_s_dcache_inv();
                if constexpr (PIPE) __syncthreads();
            }
            if (stamper)  // (developer probe: the end of the hand-over to the next period)
                ((long long *)ka_.probe)[probo * 16 + 12] = (long long)__builtin_readcyclecounter();
            // (no vector-L1 invalidate: the two wavefronts of a workgroup share their CU's L1, which its own stores keep
            // coherent -- workgroup scope in the AMDGPU memory model; an agent-scope acquire here cost 4-8 us per period)
        }
    }
}

// ------------------------------------------------------------ host side
bool stage_supported(const KernelArgs &ka, int dtype)
{
    return dtype == MPCQP_F64 && ka.nx >= 2 && ka.nx <= 4 && ka.nu >= 1 && ka.nu <= 2 && ka.mk >= 1;
}

int stage_default_maxq(const KernelArgs &ka)
{
    int q = ka.n < ka.m ? ka.n : ka.m;
    return q < 128 ? q : 128;
}

size_t stage_ws_doubles(const KernelArgs &ka, int maxq) { return (size_t)make_ws(ka.nx, ka.nu, ka.N, ka.mk, maxq).total; }

template <int NX, int NU, bool SERIAL, bool PIPE, bool WARM, int PW = 1>
static int launch_stage_p(const KernelArgs &ka, const Ws &wl, size_t lds_problem, int64_t batch, void *ws, hipStream_t st)
{
    auto kern = mpcqp_stage_kernel<NX, NU, SERIAL, PIPE, WARM, PW>;
    const size_t lds = lds_problem * PW;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const StageArgs sa{ka, wl, (double *)ws, batch, (int64_t)lds_problem};
    hipLaunchKernelGGL(kern, dim3((unsigned)((batch + PW - 1) / PW)), dim3(PIPE ? (PW == 4 ? 320 : 128) : 64), lds, st, sa);
    return (int)hipGetLastError();
}

template <int NX, int NU, bool SERIAL, bool PIPE, bool WARM>
static int launch_stage_s(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st)
{
    const Ws wl = make_ws(ka.nx, ka.nu, ka.N, ka.mk, maxq);
    size_t lds = (size_t)maxq * (3 * sizeof(double) + 2 * sizeof(int)) + 32 * sizeof(double);
    if (SERIAL) lds += (size_t)(serial_lds_doubles(ka.N, NX, NU) - 32) * sizeof(double);
    if (PIPE) lds += (32 + (size_t)ka.N * (16 + 4 * NU)) * sizeof(double);  // the factor wavefront's exchange cells + operands (padded)
    lds = (lds + 15) & ~(size_t)15;
    if constexpr (PIPE && STAGE_PW == 4) {
        // four problems per workgroup and ONE factor wavefront for them, when they fit the CU's 160 KB
        // (... with the second LDS image of every problem, and the sweeps' request distance of slack behind it)
        const size_t lds4 = lds + (size_t)(ka.N + STAGE_SRD) * serial_fs(NU) * sizeof(double);
        if (4 * lds4 <= 160 * 1024) return launch_stage_p<NX, NU, SERIAL, PIPE, WARM, 4>(ka, wl, lds4, batch, ws, st);
    }
    return launch_stage_p<NX, NU, SERIAL, PIPE, WARM, 1>(ka, wl, lds, batch, ws, st);
}

static bool stage_serial(const KernelArgs &ka)
{
    // serial sweeps: short horizons whose factor fits 32 KB of LDS next to the active-set vectors
    return ka.N <= kSerialMaxN && (size_t)serial_lds_doubles(ka.N, ka.nx, ka.nu) * sizeof(double) <= 40 * 1024;
}

// MPCQP_OPT_PIPELINE_FACTOR needs the factor image of the serial instantiations
bool stage_pipeline_supported(const KernelArgs &ka, int dtype) { return stage_supported(ka, dtype) && stage_serial(ka); }

template <int NX, int NU> static int launch_stage_t(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st)
{
    const bool serial = stage_serial(ka);
    if (ka.opt_flags & MPCQP_OPT_PIPELINE_FACTOR)  // (a rebuilt factor: the stored active set is not used, warm_state is ignored)
        return serial ? launch_stage_s<NX, NU, true, true, false>(ka, maxq, batch, ws, st) : MPCQP_EUNSUPPORTED;
    if (ka.warm_state)
        return serial ? launch_stage_s<NX, NU, true, false, true>(ka, maxq, batch, ws, st)
                      : launch_stage_s<NX, NU, false, false, true>(ka, maxq, batch, ws, st);
    return serial ? launch_stage_s<NX, NU, true, false, false>(ka, maxq, batch, ws, st)
                  : launch_stage_s<NX, NU, false, false, false>(ka, maxq, batch, ws, st);
}

int launch_stage(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st)
{
    switch (ka.nx * 10 + ka.nu) {
    case 21: return launch_stage_t<2, 1>(ka, maxq, batch, ws, st);
    case 22: return launch_stage_t<2, 2>(ka, maxq, batch, ws, st);
    case 31: return launch_stage_t<3, 1>(ka, maxq, batch, ws, st);
    case 32: return launch_stage_t<3, 2>(ka, maxq, batch, ws, st);
    case 41: return launch_stage_t<4, 1>(ka, maxq, batch, ws, st);
    case 42: return launch_stage_t<4, 2>(ka, maxq, batch, ws, st);
    default: return MPCQP_EUNSUPPORTED;
    }
}

}  // namespace mpcqp
