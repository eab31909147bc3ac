// Internal declarations shared by the kernel translation units and the C ABI.
#ifndef MPCQP_INTERNAL_H_
#define MPCQP_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpcqp.h"

namespace mpcqp {

enum { MODE_FUSED = 0, MODE_CONDENSE = 1, MODE_SOLVE = 2, MODE_MODEL = 3 };

constexpr size_t kLdsBytesPerCU = 160 * 1024;  // gfx950: 160 KiB per CU

// SIMDs of the device that is current for this call (4 per compute unit). Asked per call: the library keeps no state between
// calls (include/mpcqp.h), and a process may drive several devices. 0 when the runtime cannot tell.
inline int device_simds_now()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return 4 * cus;
}

// INTERNAL bit of KernelArgs.opt_flags (never taken from MpcqpSolveOpts.flags: fill_opts clears it): the wide stage-wise kernel runs as
// the SECOND OPINION behind the narrow one -- a wavefront whose problem the first launch settled (solved, not positive definite,
// slots full) exits at once, the others (MPCQP_MAX_ITER, MPCQP_INFEASIBLE) are solved again from scratch. Round 6: the narrow kernel
// still keeps the explicit inverse of the active rows' Gram matrix and, when every variable is pinned, gives a wrong verdict on one
// problem in some hundreds; the wide kernel's thin-QR operator does not, for the price of a launch of wavefronts that exit.
constexpr int kOptSecondOpinion = 1 << 30;

// Everything a kernel needs, passed by value (kernarg segment, scalar loads).
struct KernelArgs {
    int nx, nu, N, mk, n, m, flags, max_iter;
    double wt, wx, wu, tol;
    MpcqpOperand A, B, C, D, e, x0, goal, targets;
    // QP buffers: outputs of CONDENSE, inputs of SOLVE
    void *P, *q, *G, *h, *Phi, *Psi;
    // solve outputs
    void *U, *lam;
    int32_t *status, *iters;
    // rollout
    void *X;
    // per-problem solver arrays in HBM (large problems): batch * Layout.total elements
    void *ws;
    // shared-model path: the factored model (ModelLayout) 
    const void *model;
    // MpcqpSolveOpts beyond max_iter / feas_tol
    int opt_flags;                // MPCQP_OPT_*
    void *warm_state;             // per-problem active set + operator (read if warm_start, written at the end), or null
    int warm_start;               // 0 | MPCQP_WARM_OPERATOR | MPCQP_WARM_ACTIVE_SET
    int warm_shift;               // MPCQP_WARM_ACTIVE_SET: rows the stored ids move down by
    int factor_slot;              // which of the two factor images of the stage-wise kernel this launch keeps / reuses
    size_t warm_state_bytes;      // size of the buffer behind warm_state (checked on the host before the launch)
    void *probe;                  // developer probe: int64 stamps per problem, or null
    // closed-loop epilogue of the stage-wise kernel (mpcqp_wip_period_batch): after its solve every wavefront applies
    // the first input of its plan to the wheeled-inverted-pendulum plant and writes its loop's NEXT problem in place
    int ep_on, ep_nsub, ep_periods;  // (ep_periods: control periods per launch, mpcqp_wip_periods_batch; 0 / 1 = one)
    double ep_Tp, ep_vel, ep_omega2, ep_g;
    void *ep_states;              // [batch, 4], updated in place
    long long *ep_loopstats;      // [batch, 2]: += (failed, iterations), or null
    // (last: the fields above keep the offsets the stage-wise kernels' hand-placed kernel-argument loads were measured with)
    const int32_t *order;         // pairing order of the small-problem fused kernel (a permutation of the batch), or null
};

// LDS carve, in elements of T. Matrices are row-major with odd row stride ld.
struct Layout {
    int ld;
    int off_A, off_B, off_x0;  // staged dynamics (build only)
    int off_X;                 // union: Psi blocks 1..N (build) | Q, S (solve)
    int off_P, off_M;          // P -> L ; G -> M = G L^-T, row m holds q -> L^-1 q
    int off_h, off_hs, off_s, off_y, off_z, off_d, off_r, off_u, off_cs, off_sn, off_inv, off_xs, off_red;
    int off_int;               // int act[n+2], where[m], redi[8]
    int total;
};

inline Layout make_layout(int nx, int nu, int N, int n, int m, bool stepA, bool stepB, int mode, size_t esz)
{
    Layout L{};
    L.ld = (n + 1) | 1;
    int o = 0;
    auto take = [&](int cnt) {
        int at = o;
        o += (cnt + 1) & ~1;  // keep 8-byte alignment for float too
        return at;
    };
    if (mode != MODE_SOLVE) {
        L.off_A = take((stepA ? N : 1) * nx * nx);
        L.off_B = take((stepB ? N : 1) * nx * nu);
        L.off_x0 = take(nx);
        const int psi = N * nx * L.ld, qs = 2 * n * L.ld;
        L.off_X = take(psi > qs ? psi : qs);
    } else {
        L.off_A = L.off_B = L.off_x0 = 0;
        L.off_X = take(2 * n * L.ld);
    }
    L.off_P = take(n * L.ld);
    L.off_M = take((m + 1) * L.ld);
    L.off_h = take(m);
    L.off_hs = take(m);
    L.off_s = take(m);
    L.off_y = take(n);
    L.off_z = take(n);
    L.off_d = take(n);
    L.off_r = take(n);
    L.off_u = take(n + 1);
    L.off_cs = take(n);
    L.off_sn = take(n);
    L.off_inv = take(n);
    L.off_xs = take(n);
    L.off_red = take(8);
    L.off_int = o;
    const int ints = (n + 2) + m + 8;
    o += (int)(((size_t)ints * 4 + esz - 1) / esz);
    o = (o + 1) & ~1;
    L.total = o;
    return L;
}

// Layout of a factored shared model, in elements of T. nc = max(n, 16) columns: models
// with n < 16 are padded with unit variables so that the wavefront kernel can use them.
struct ModelLayout {
    int nc, nb;  // padded variable count; number of pseudo-problems 1 + 2 nx + N nx
    size_t off_M, off_LinvT, off_invn, off_e, off_Hx, off_Wx, off_Wg, off_Wt, total;
};
__host__ __device__ inline ModelLayout make_model_layout(int nx, int N, int n, int m)
{
    ModelLayout ml{};
    ml.nc = n < 16 ? 16 : n;
    ml.nb = 1 + 2 * nx + N * nx;
    size_t o = 0;
#define MPCQP_TAKE(field, cnt) \
    ml.field = o;              \
    o += ((size_t)(cnt) + 3) & ~(size_t)3;
    MPCQP_TAKE(off_M, (size_t)m * ml.nc)
    MPCQP_TAKE(off_LinvT, (size_t)ml.nc * ml.nc)
    MPCQP_TAKE(off_invn, m)
    MPCQP_TAKE(off_e, m)
    MPCQP_TAKE(off_Hx, (size_t)m * nx)
    MPCQP_TAKE(off_Wx, (size_t)ml.nc * nx)
    MPCQP_TAKE(off_Wg, (size_t)ml.nc * nx)
    MPCQP_TAKE(off_Wt, (size_t)ml.nc * N * nx)
#undef MPCQP_TAKE
    ml.total = o;  // one more element follows: the "P not positive definite" flag
    return ml;
}
int launch_wip_advance(int dtype, void *states, const void *U, int64_t u_stride, const int32_t *status,
                       const int32_t *iters, int64_t *stats, int N, double Tp, double vel, double length, double gravity,
                       int nsub, void *x0, void *goal, void *targets, int64_t batch, hipStream_t st);
int launch_lipm_advance(int dtype, void *states, const void *U, int64_t u_stride, const int32_t *status, int N,
                        double Tp, int nsub, int nb_dsp, int nb_ssp, double max_zmp, int64_t *index,
                        int64_t *stride_index, void *support, const void *strides, const void *foot_size, void *x0,
                        void *goal, void *e, int64_t batch, const int32_t *iters, int64_t *stats, hipStream_t st);
int launch_stats(const int32_t *status, const int32_t *iters, int64_t batch, int64_t *stats, hipStream_t st);
int launch_factor_model(const KernelArgs &ka, int dtype, const void *P, const void *G, const void *qb, const void *hb,
                        void *model, hipStream_t st);

template <int MODE>
int dispatch_lds(const KernelArgs &ka, const Layout &L, int dtype, int64_t batch, hipStream_t st);
// same solver with its arrays in a global workspace (ka.ws) instead of LDS
int dispatch_gws_solve(const KernelArgs &ka, const Layout &L, int dtype, int64_t batch, hipStream_t st);
// large-problem condensing (mpcqp_big.hip)
size_t big_condense_ws_elems(const KernelArgs &ka);
bool big_supported(const KernelArgs &ka);
// G may be null (not formed); rownorm_inv (1/|G_i|, [batch, m]) is optional
int launch_big_condense(const KernelArgs &ka, int dtype, int64_t batch, void *Psi_ws, void *res_ws, void *P, void *q,
                        void *G, void *h, void *rownorm_inv, hipStream_t st, int phase = 0);
// large-problem solver (mpcqp_bigsolve.hip): one problem per workgroup, L^-1 packed in LDS
#ifdef __HIPCC__
// ---- wavefront all-reductions on the vector pipe (the stage-wise kernels; __shfl_xor is a ds_bpermute per dword and step:
// 18 dependent LDS round trips for one (double, int) arg-min). Four DPP steps reduce every 16-lane row in all of its lanes
// (xor 1, xor 2 inside the quads, row_half_mirror, row_mirror: every step pairs each lane with a lane of the other half of
// its group, and the operations are commutative, so the lanes of a row end with the same bits); the four row results meet
// through v_readlane. Call with all 64 lanes active.
template <int CTRL, typename T> __device__ __forceinline__ T dpp_mov(T x)
{
    if constexpr (sizeof(T) == 8) {
        const long long b = __builtin_bit_cast(long long, x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
        return __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned)lo);
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
    }
}
template <typename T> __device__ __forceinline__ T lane_get(T x, int l)  // (l: wavefront-uniform)
{
    if constexpr (sizeof(T) == 8) {
        const long long b = __builtin_bit_cast(long long, x);
        const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
        return __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned)lo);
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
    }
}
template <typename T> __device__ __forceinline__ T wave_sum_dpp(T v)
{
    v += dpp_mov<0xb1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4e>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
// (value, index) arg-min; ties -> lowest index (a total order: any reduction tree gives the same pair); every lane gets it
template <typename T> __device__ __forceinline__ void wave_argmin_dpp(T &v, int &idx)
{
    auto merge = [&](T ov, int oi) {
        const bool take = (ov < v) || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    };
    merge(dpp_mov<0xb1>(v), dpp_mov<0xb1>(idx));
    merge(dpp_mov<0x4e>(v), dpp_mov<0x4e>(idx));
    merge(dpp_mov<0x141>(v), dpp_mov<0x141>(idx));
    merge(dpp_mov<0x140>(v), dpp_mov<0x140>(idx));
    const T v1 = lane_get(v, 16), v2 = lane_get(v, 32), v3 = lane_get(v, 48);
    const int i1 = lane_get(idx, 16), i2 = lane_get(idx, 32), i3 = lane_get(idx, 48);
    v = lane_get(v, 0);
    idx = lane_get(idx, 0);
    merge(v1, i1);
    merge(v2, i2);
    merge(v3, i3);
}
#endif
bool bigsolve_supported(int n, int m, int dtype);
size_t bigsolve_ws_elems(int n);
bool bigsolve_struct_supported(const KernelArgs &ka, int dtype);
int launch_bigsolve_struct(const KernelArgs &ka, int dtype, int64_t batch, const void *P, const void *q,
                           const void *Psi_all, const void *h, const void *rownorm_inv, void *ws, hipStream_t st);
// mid-size fused build+solve (mpcqp_bigsolve.hip, KIND = K_MID): ws = 2 n^2 elements per problem
bool mid_supported(const KernelArgs &ka, int dtype);
int launch_mid(const KernelArgs &ka, int dtype, int64_t batch, void *ws, hipStream_t st);
int launch_transpose(const void *G, void *GT, int m, int n, int dtype, int64_t batch, hipStream_t st);
int launch_bigsolve(const KernelArgs &ka, int dtype, int64_t batch, const void *P, const void *q, const void *G,
                    const void *GT, const void *h, void *ws, hipStream_t st);
// stage-wise formulation (mpcqp_stage.hip): one problem per wavefront, O(N) per iteration
bool stage_supported(const KernelArgs &ka, int dtype);
int stage_default_maxq(const KernelArgs &ka);
bool stage_pipeline_supported(const KernelArgs &ka, int dtype);
__host__ __device__ size_t stage_warm_bytes(int maxq);
size_t stage_ws_doubles(const KernelArgs &ka, int maxq);
int launch_stage(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st);
// ... for wider systems (mpcqp_stagew.hip): nx <= 16, nu <= 4, f64 and f32; workspace in elements of the dtype
bool stagew_supported(const KernelArgs &ka, int dtype);
size_t stagew_ws_elems(const KernelArgs &ka, int maxq, int dtype);
int launch_stagew(const KernelArgs &ka, int dtype, int maxq, int64_t batch, void *ws, hipStream_t st);
// warm-state record of the wide stage-wise kernel (MPCQP_WARM_ACTIVE_SET): int32 count, then the active rows' ids
__host__ __device__ inline size_t stagew_warm_bytes(int maxq) { return ((size_t)(maxq + 1) * 4 + 15) & ~(size_t)15; }
// ... and for every other system (mpcqp_stageg.hip): nx <= 32, nu <= 8, float64, any horizon; one workgroup per problem, all
// arrays in the workspace -- a general fallback, not a tuned path
bool stageg_supported(const KernelArgs &ka, int dtype);
int stageg_default_maxq(const KernelArgs &ka);
size_t stageg_ws_doubles(const KernelArgs &ka, int maxq);
int launch_stageg(const KernelArgs &ka, int maxq, int64_t batch, void *ws, hipStream_t st);
// small-problem kernel (mpcqp_pair.hip): two problems per wavefront, fused build+solve
bool pair_eligible(const KernelArgs &ka, int mode, int dtype);
constexpr size_t kPairWarmDoubles = 16 * 16 + 8;  // T (16 x 16), then 16 int32 constraint ids
int launch_pair(const KernelArgs &ka, int64_t batch, hipStream_t st);
int launch_pair_model(const KernelArgs &ka, int64_t batch, hipStream_t st);  // shared model (ka.model)
// small-problem kernel (mpcqp_quad.hip): four problems per wavefront, cold lean fused build+solve
bool quad_applies(const KernelArgs &ka);                  // the kernel serves this launch's layout
bool quad_model_eligible(const KernelArgs &ka, int64_t batch);  // shared-model launches of that layout
int launch_quad_model(const KernelArgs &ka, int64_t batch, hipStream_t st);
bool quad_eligible(const KernelArgs &ka, int64_t batch);  // ... and the dispatch takes it (batch size, MPCQP_OPT_TWO / FOUR_PER_WAVE)
int launch_quad(const KernelArgs &ka, int64_t batch, hipStream_t st);
// ... and its four-rows-per-lane copy for 33 .. 64 rows (mpcqp_quad4.hip): cold launches, any batch size
bool quad4_applies(const KernelArgs &ka);
int launch_quad4(const KernelArgs &ka, int64_t batch, hipStream_t st);
// small-problem kernel (mpcqp_w64.hip): one problem per wavefront
bool w64_eligible(const KernelArgs &ka, int mode, int dtype);
int launch_w64(const KernelArgs &ka, int mode, int dtype, int64_t batch, hipStream_t st);
int launch_phi(const KernelArgs &ka, int dtype, int64_t batch, hipStream_t st);
int launch_update(const KernelArgs &ka, int dtype, int64_t phi_bs, int64_t psi_bs, int64_t batch, hipStream_t st);
int launch_rollout(const KernelArgs &ka, int dtype, int64_t batch, hipStream_t st);

}  // namespace mpcqp
#endif
