// mpcqp_capi.hip -- the extern "C" boundary declared in include/mpcqp.h.
// Argument validation happens here, on the host, before any launch; kernels
// live in mpcqp_lds.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "mpcqp.h"
#include "mpcqp_internal.h"

using namespace mpcqp;

namespace {

size_t elem_size(int dtype) { return dtype == MPCQP_F64 ? 8 : 4; }

int check_dims(const MpcqpDims *d)
{
    if (!d) return MPCQP_EINVAL;
    if (d->dtype != MPCQP_F64 && d->dtype != MPCQP_F32) return MPCQP_EDTYPE;
    if (d->nx <= 0 || d->nu <= 0 || d->N <= 0 || d->mk < 0) return MPCQP_EINVAL;
    if (!(d->w_input > 0.0)) return MPCQP_EINVAL;  // mpc_problem.py:104-107
    return 0;
}

int check_step(const MpcqpOperand &op, int64_t block)
{
    if (op.ptr && op.step_stride != 0 && op.step_stride != block) return MPCQP_ELAYOUT;
    return 0;
}

int check_problem(const MpcqpDims *d, const MpcqpProblem *p)
{
    if (!p || !p->A.ptr || !p->B.ptr || !p->x0.ptr) return MPCQP_EINVAL;
    if (d->mk > 0 && !p->e.ptr) return MPCQP_EINVAL;
    if ((d->flags & MPCQP_Q_TERMINAL) && !p->goal.ptr) return MPCQP_EINVAL;
    if ((d->flags & MPCQP_Q_STAGE) && !p->targets.ptr) return MPCQP_EINVAL;
    int rc;
    if ((rc = check_step(p->A, (int64_t)d->nx * d->nx))) return rc;
    if ((rc = check_step(p->B, (int64_t)d->nx * d->nu))) return rc;
    if ((rc = check_step(p->C, (int64_t)d->mk * d->nx))) return rc;
    if ((rc = check_step(p->D, (int64_t)d->mk * d->nu))) return rc;
    if ((rc = check_step(p->e, (int64_t)d->mk))) return rc;
    return 0;
}

void fill_args(KernelArgs &ka, const MpcqpDims *d, const MpcqpProblem *p)
{
    memset(&ka, 0, sizeof(ka));
    ka.nx = d->nx;
    ka.nu = d->nu;
    ka.N = d->N;
    ka.mk = d->mk;
    ka.n = d->N * d->nu;
    ka.m = d->N * d->mk;
    ka.flags = d->flags;
    ka.wt = d->w_terminal;
    ka.wx = d->w_stage;
    ka.wu = d->w_input;
    if (p) {
        ka.A = p->A;
        ka.B = p->B;
        ka.C = p->C;
        ka.D = p->D;
        ka.e = p->e;
        ka.x0 = p->x0;
        ka.goal = p->goal;
        ka.targets = p->targets;
    }
}

int fill_opts(KernelArgs &ka, const MpcqpSolveOpts *o, int dtype, bool order_ok = false)
{
    ka.max_iter = (o && o->max_iter > 0) ? o->max_iter : 10 * (ka.n + ka.m) + 10;
    ka.tol = (o && o->feas_tol > 0.0) ? o->feas_tol : (dtype == MPCQP_F64 ? 1e-12 : 1e-5);
    if (!o) return 0;
    ka.opt_flags = o->flags & ~kOptSecondOpinion;
    ka.probe = o->probe;
    ka.warm_state = o->warm_state;
    ka.warm_start = o->warm_state ? o->warm_start : 0;
    ka.warm_shift = o->warm_state ? o->warm_shift : 0;
    if (ka.warm_start < 0 || ka.warm_start > MPCQP_WARM_ACTIVE_SET) return MPCQP_EINVAL;
    ka.warm_state_bytes = o->warm_state ? o->warm_state_bytes : 0;
    ka.factor_slot = o->factor_slot & 1;
    if (o->order && !order_ok) return MPCQP_EUNSUPPORTED;  // (only mpcqp_build_solve_batch's small-problem kernel takes one)
    ka.order = o->order;
    return 0;
}

int layout_for(const KernelArgs &ka, bool stepA, bool stepB, int mode, int dtype, Layout &L)
{
    L = make_layout(ka.nx, ka.nu, ka.N, ka.n, ka.m, stepA, stepB, mode, elem_size(dtype));
    if ((size_t)L.total * elem_size(dtype) > kLdsBytesPerCU) return MPCQP_ETOOLARGE;
    if (ka.n > 256) return MPCQP_ETOOLARGE;
    return 0;
}

// Dispatch overrides are explicit bits of MpcqpSolveOpts.flags (no process-wide state):
// MPCQP_OPT_FORCE_LDS routes small problems through the general LDS kernel too, MPCQP_OPT_FORCE_GWS keeps
// large QPs on the general kernel with its arrays in the workspace, MPCQP_OPT_FORCE_DENSE_G makes the fused
// large path form G instead of applying it through the roll-out (cross-checks of two formulations).
bool force_lds(int fl) { return fl & MPCQP_OPT_FORCE_LDS; }
bool force_gws(int fl) { return fl & MPCQP_OPT_FORCE_GWS; }
bool force_dense_g(int fl) { return fl & MPCQP_OPT_FORCE_DENSE_G; }
// every combination of overrides a launch may carry: the workspace queries, which see no opts, report the
// largest amount any of them needs
const int kFlagVariants[] = {0, MPCQP_OPT_FORCE_LDS, MPCQP_OPT_FORCE_GWS, MPCQP_OPT_FORCE_DENSE_G, MPCQP_OPT_FORCE_CONDENSED};

// float64 problems that would take the dense HBM-resident path (nx <= 16, nu > 4, n <= 256: condense + one QP per workgroup)
// go to the general stage-wise kernel instead, unless an override flag asks for the dense solvers: since its second version
// (round 4) it is 15-20x faster there (512 problems of nx = 12, nu = 6, N = 40: 124 against 8.4 ms). float32 keeps the dense
// path (MFMA Gram, float32 solver), which is what makes that size affordable in float32.
static bool prefer_general(const KernelArgs &ka, int dtype)
{
    const int override_bits = MPCQP_OPT_FORCE_LDS | MPCQP_OPT_FORCE_GWS | MPCQP_OPT_FORCE_DENSE_G | MPCQP_OPT_FORCE_CONDENSED |
                              MPCQP_OPT_ONE_PER_WAVE;
    return dtype == MPCQP_F64 && !(ka.opt_flags & override_bits) && !ka.warm_state && stageg_supported(ka, MPCQP_F64);
}

// Fused build+solve of mid-size problems of small systems goes to the stage-wise kernel (mpcqp_stage.hip): same
// minimiser (tests), 1.5-1.9x the mid-size condensed kernel on config 3. Its slots hold min(n, m) <= 128 active rows, i.e.
// every row that can be active at once, so nothing is lost against the condensed kernels.
bool use_stage_auto(const KernelArgs &ka, int dtype)
{
    const int override_bits = MPCQP_OPT_FORCE_LDS | MPCQP_OPT_FORCE_GWS | MPCQP_OPT_FORCE_DENSE_G | MPCQP_OPT_FORCE_CONDENSED |
                              MPCQP_OPT_ONE_PER_WAVE;
    return !(ka.opt_flags & override_bits) && stage_supported(ka, dtype) && ka.n > 16 && ka.n <= 128 && ka.m >= 1;
}

// Fused build+solve of problems that do NOT fit the on-chip condensed kernels goes to the wide stage-wise kernel
// (mpcqp_stagew.hip) when the system fits it (nx <= 16, nu <= 4): same minimiser (tests), 7-10x the HBM-resident
// condensed path at BASELINE config 5's size, and float32 errors ~300x smaller (nothing is squared into P). Its slots
// hold min(n, m, 256) active rows -- every row that can be active at once where the condensed path exists (n <= 256).
bool fits_on_chip(const KernelArgs &ka, bool stepA, bool stepB, int mode, int dtype);
bool use_stagew_auto(const KernelArgs &ka, int dtype)
{
    const int override_bits = MPCQP_OPT_FORCE_LDS | MPCQP_OPT_FORCE_GWS | MPCQP_OPT_FORCE_DENSE_G | MPCQP_OPT_FORCE_CONDENSED |
                              MPCQP_OPT_ONE_PER_WAVE;
    // Round 3: ... and what DOES fit on chip but is no small problem (n > 24): measured on batches of 512 random LTV problems
    // (tools/probe_f32_dispatch.py), the stage-wise kernel is 2-2.5x the mid-size / LDS condensed kernels in float64 (nx = 6 .. 12,
    // n = 37 .. 64: 590-980 us against 1180-2170 us) and 2-3x in float32, where it is also 5-30x closer to the float64 oracle
    // (the condensed float32 path squares the conditioning into P: 1e-3 at n ~ 130). Small problems stay on chip: up to n = 20 since
    // round 6 (4096 problems, float64, default / wide stage-wise, us: (nx, nu, N) = (8, 2, 10) n = 20: 824 / 843; (8, 2, 12) n = 24:
    // 1471 / 1087; (12, 4, 6): 860 / 490; (7, 1, 24): 2332 / 1123; (6, 2, 12): 683 / 535; n <= 16: the on-chip kernels by 1.2-2.3x).
    return !(ka.opt_flags & override_bits) && !(ka.warm_state && ka.warm_start == MPCQP_WARM_OPERATOR) && stagew_supported(ka, dtype) && ka.m >= 1 &&
           ((ka.n > 20 && (dtype == MPCQP_F64 || ka.nx <= 12)) || !fits_on_chip(ka, true, true, MODE_FUSED, dtype));
    // (float32 with nx > 12 -- the LDS-tiled Riccati recursion -- stays on chip while it fits: on borderline problems of that size
    // the condensed float32 kernel was the closer one, 1e-3 against 3e-3)
}
// ... and the narrow stage-wise kernel (float64, nx <= 4, nu <= 2: chunked scans, depth ~2 N / 64 per sweep instead of N
// serial steps) takes what does not fit on chip among the systems it serves: horizons of any length (n > 256 included)
bool use_stage_long(const KernelArgs &ka, int dtype)
{
    const int override_bits = MPCQP_OPT_FORCE_LDS | MPCQP_OPT_FORCE_GWS | MPCQP_OPT_FORCE_DENSE_G | MPCQP_OPT_FORCE_CONDENSED |
                              MPCQP_OPT_ONE_PER_WAVE;
    // Round 3, late: no longer taken by the automatic dispatch. On long horizons the two stage-wise kernels tie when few rows become
    // active (triple integrator, N = 256 / 1024 / 4096, 3.5-12 iterations: 4.97 / 52.2 / 399 ms narrow against 6.03 / 54.2 / 387 ms
    // wide) and the wide one -- active-set state in LDS, eight right-hand sides per sweep pair, 256 slots -- is 1.3-2x faster when
    // many do (random LTV, n = 160 .. 256, 50-140 iterations: 7.5 / 40.3 / 13.4 / 42.9 ms against 5.0 / 26.8 / 6.9 / 32.7 ms;
    // tools/probe_long_narrow.py, tools/probe_narrow_vs_wide.py): n > 128 goes to the wide kernel (use_stagew_auto). The narrow
    // kernel stays reachable through mpcqp_stagewise_solve_batch.
    (void)override_bits;
    (void)dtype;
    return false;
}
// the narrow stage-wise kernel's unsolved verdicts get a second opinion from the wide one (kOptSecondOpinion) in one-shot launches of
// mpcqp_build_solve_batch: not when the launch keeps state in its workspace or its warm-state record for a later one
bool second_opinion_applies(const KernelArgs &ka, int dtype, bool size_query = false)
{
    // (a size query carries neither outputs nor options: it prices the launch that takes the second opinion)
    return stagew_supported(ka, dtype) && (size_query || ka.status) && !ka.warm_state &&
           !(ka.opt_flags & (MPCQP_OPT_KEEP_FACTOR | MPCQP_OPT_REUSE_FACTOR | MPCQP_OPT_PIPELINE_FACTOR));
}
int stagew_auto_maxq(const KernelArgs &ka)
{
    const int q = ka.n < ka.m ? ka.n : ka.m;
    return q < 256 ? q : 256;
}

// warm start: the small-problem pair kernel (operator + slot ids) and the narrow stage-wise kernel (row ids; the rows'
// vectors stay in its workspace); 0 = not offered for these dimensions
bool promote_f32(const KernelArgs &ka, int dtype);
// (kind: MPCQP_WARM_KIND_* -- whose record it is; the host side reads it instead of re-deriving the dispatch)
size_t warm_bytes_per_problem(KernelArgs ka, int dtype, int *kind = nullptr)
{
    ka.opt_flags = 0;
    ka.warm_state = nullptr;
    int k = MPCQP_WARM_KIND_NONE;
    size_t bytes = 0;
    if (dtype == MPCQP_F32 && promote_f32(ka, dtype)) dtype = MPCQP_F64;  // (the launch that is solved in float64: its kernel's record)
    if (pair_eligible(ka, MODE_FUSED, dtype)) {
        k = MPCQP_WARM_KIND_OPERATOR;
        bytes = kPairWarmDoubles * sizeof(double);
    } else if (use_stage_auto(ka, dtype)) {
        k = MPCQP_WARM_KIND_STAGE;
        bytes = stage_warm_bytes(stage_default_maxq(ka));
    } else if (use_stagew_auto(ka, dtype)) {
        k = MPCQP_WARM_KIND_ROWS;
        bytes = stagew_warm_bytes(stagew_auto_maxq(ka));  // row ids only (MPCQP_WARM_ACTIVE_SET)
    }
    if (kind) *kind = k;
    return bytes;
}

bool use_bigsolve(int n, int m, int dtype, int fl) { return !force_gws(fl) && m > 0 && bigsolve_supported(n, m, dtype); }

// Per-problem solver scratch (elements) for QPs that do not fit the on-chip kernels.
size_t solver_ws_elems(int n, int m, int dtype, int fl)
{
    if (use_bigsolve(n, m, dtype, fl)) return (size_t)m * n + bigsolve_ws_elems(n);  // G' + M_A + N*
    const Layout L = make_layout(1, 1, n, n, m, false, false, MODE_SOLVE, elem_size(dtype));
    return (size_t)L.total;
}

bool use_struct(const KernelArgs &ka, int dtype)
{
    return !force_gws(ka.opt_flags) && !force_dense_g(ka.opt_flags) && bigsolve_struct_supported(ka, dtype);
}

// Sizes (in elements) of the pieces of the large-path workspace, per problem.
struct BigPlan {
    size_t psi, P, q, G, h, nrm, solver;  // psi includes the residual vector
    bool matrix_free;                     // G applied through the roll-out, never formed
    size_t total(bool with_qp) const { return psi + (with_qp ? P + q + G + h + nrm : 0) + solver; }
};

BigPlan big_plan(const KernelArgs &ka, int dtype, bool condense, bool solve)
{
    BigPlan b{};
    if (condense) b.psi = big_condense_ws_elems(ka);
    if (solve) {
        b.matrix_free = use_struct(ka, dtype);
        b.P = (size_t)ka.n * ka.n;
        b.q = ka.n;
        b.h = ka.m;
        if (b.matrix_free) {
            b.nrm = ka.m;
            b.solver = bigsolve_ws_elems(ka.n);
        } else {
            b.G = (size_t)ka.m * ka.n;
            b.solver = solver_ws_elems(ka.n, ka.m, dtype, ka.opt_flags);
        }
    }
    return b;
}

// Mid-size fused problems (config 3) go to the lean one-launch kernel unless the small-problem kernel
// takes them or MPCQP_OPT_FORCE_LDS asks for the all-in-LDS kernel (cross-checks).
bool use_mid(const KernelArgs &ka, int dtype)
{
    return !force_lds(ka.opt_flags) && !w64_eligible(ka, MODE_FUSED, dtype) && mid_supported(ka, dtype);
}

// mpcqp_workspace_bytes sees the dimensions only, not the operand strides: it reports the mid-size
// kernel's workspace whenever the most compact operand layout (LTI, no C/D) would be taken. A launch with
// bulkier operands may still fall back to the all-in-LDS kernel, which simply ignores the workspace.
bool problem_strides_unknown_mid(KernelArgs ka, int dtype)
{
    ka.A.step_stride = ka.B.step_stride = ka.C.step_stride = ka.D.step_stride = 0;
    ka.C.ptr = ka.D.ptr = nullptr;
    return use_mid(ka, dtype);
}

bool fits_on_chip(const KernelArgs &ka, bool stepA, bool stepB, int mode, int dtype)
{
    if (!force_lds(ka.opt_flags) && w64_eligible(ka, mode, dtype)) return true;
    Layout L;
    return layout_for(ka, stepA, stepB, mode, dtype, L) == 0;
}

// Solve QPs given in HBM (ka.P/q/G/h) with the solver arrays in the workspace.
int run_gws_solve(KernelArgs ka, int dtype, int64_t batch, void *ws, size_t ws_bytes, hipStream_t st)
{
    if (ka.n > 256) return MPCQP_ETOOLARGE;
    const size_t esz = elem_size(dtype);
    if (!ws || ws_bytes < solver_ws_elems(ka.n, ka.m, dtype, ka.opt_flags) * esz * (size_t)batch) return MPCQP_EWORKSPACE;
    if (use_bigsolve(ka.n, ka.m, dtype, ka.opt_flags)) {
        // L^-1 packed in LDS, lazy rows of M; needs G transposed (coalesced slack updates)
        void *GT = ws;
        void *rest = (char *)ws + (size_t)ka.m * ka.n * esz * (size_t)batch;
        int rc = launch_transpose(ka.G, GT, ka.m, ka.n, dtype, batch, st);
        if (rc) return rc;
        return launch_bigsolve(ka, dtype, batch, ka.P, ka.q, ka.G, GT, ka.h, rest, st);
    }
    const Layout L = make_layout(ka.nx, ka.nu, ka.N, ka.n, ka.m, false, false, MODE_SOLVE, esz);
    ka.ws = ws;
    return dispatch_gws_solve(ka, L, dtype, batch, st);
}

template <int MODE>
int run_solver(const KernelArgs &ka, bool stepA, bool stepB, int dtype, int64_t batch, hipStream_t st)
{
    if (!force_lds(ka.opt_flags)) {
        // (nx = 2 and nx > 4 have no two-per-wavefront instantiation: the four-per-wavefront kernel takes them where it pays -- the general
        // layouts and nx > 4 at every batch size, the lean layout of nx = 2 from more than two problems per SIMD --, on its own)
        // (more than 32 rows at n <= 16: the four-rows-per-lane copy of that kernel, at every batch size -- the workgroup / one-per-wavefront
        // kernels it replaces there are 5-9 x slower)
        if (MODE == MODE_FUSED && dtype == MPCQP_F64 && !(ka.opt_flags & MPCQP_OPT_ONE_PER_WAVE) && quad4_applies(ka)) return launch_quad4(ka, batch, st);
        if (MODE == MODE_FUSED && dtype == MPCQP_F64 && !(ka.opt_flags & MPCQP_OPT_ONE_PER_WAVE) && (ka.nx == 2 || ka.nx > 4) && quad_eligible(ka, batch))
            return launch_quad(ka, batch, st);
        if (!(ka.opt_flags & MPCQP_OPT_ONE_PER_WAVE) && pair_eligible(ka, MODE, dtype)) return launch_pair(ka, batch, st);
        if (ka.warm_state) return MPCQP_EUNSUPPORTED;
        if (w64_eligible(ka, MODE, dtype)) return launch_w64(ka, MODE, dtype, batch, st);
    }
    if (ka.warm_state) return MPCQP_EUNSUPPORTED;
    Layout L;
    int rc = layout_for(ka, stepA, stepB, MODE, dtype, L);
    if (rc) return rc;
    return dispatch_lds<MODE>(ka, L, dtype, batch, st);
}

// ---- float32 problems of the on-chip condensed kernels' sizes are SOLVED IN FLOAT64 (round 4). Those kernels form
// P = w_u I + Psi' W Psi in the launch's arithmetic, which squares the conditioning: a stress run (tools/stress_f32.py)
// returned float32 plans 2e-2 from the float64 ones as SOLVED on ill-conditioned small problems, and nothing the float32
// kernel holds can certify such a plan. The problems in question are a few KB each, so the launch converts the operands
// into the caller's workspace (one kernel), runs the float64 dispatch on the copies and rounds the plan (and the
// multipliers) back: the float32 contract (1e-3) is then met with five digits to spare. Large problems that the wide
// stage-wise kernel takes (n > 160: BASELINE config 5) stay in float32 -- nothing is squared there --, and so does every
// launch that carries a dispatch override (the cross-check tests of the float32 kernels).
struct ConvSeg {
    const void *src;
    void *dst;
    int64_t count;       // elements written (densely packed)
    int64_t row = 0;     // > 0: the source is `count / row` runs of `row` elements, `src_stride` elements apart (a padded batch stride)
    int64_t src_stride = 0;
};
struct ConvPlan {
    ConvSeg seg[10];
    int nseg;
    int to_double;  // float -> double, else double -> float
};
__global__ void __launch_bounds__(256) mpcqp_convert_kernel(const ConvPlan cp)
{
    const ConvSeg sg = cp.seg[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (cp.to_double) {
        const float *a = (const float *)sg.src;
        double *b = (double *)sg.dst;
        if (sg.row > 0) {  // (rows of a padded source packed densely)
            for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sg.count; i += stride) {
                const int64_t r = i / sg.row;
                b[i] = (double)a[r * sg.src_stride + (i - r * sg.row)];
            }
            return;
        }
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sg.count; i += stride) b[i] = (double)a[i];
    } else {
        const double *a = (const double *)sg.src;
        float *b = (float *)sg.dst;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sg.count; i += stride) b[i] = (float)a[i];
    }
}
int launch_convert(const ConvPlan &cp, hipStream_t st)
{
    if (cp.nseg == 0) return 0;
    int64_t mx = 0;
    for (int i = 0; i < cp.nseg; ++i) mx = cp.seg[i].count > mx ? cp.seg[i].count : mx;
    int64_t gx = (mx + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(mpcqp_convert_kernel, dim3((unsigned)gx, (unsigned)cp.nseg), dim3(256), 0, st, cp);
    return (int)hipGetLastError();
}

// elements one operand occupies: `block` per step, N steps unless shared along the horizon, `batch` problems unless shared
int64_t operand_elems(const MpcqpOperand &op, int64_t block, int N, int64_t batch, bool per_step)
{
    if (!op.ptr) return 0;
    const int64_t per_problem = (per_step && op.step_stride) ? (int64_t)N * block : block;
    return op.batch_stride ? (batch - 1) * op.batch_stride + per_problem : per_problem;
}
int64_t al256(int64_t bytes) { return (bytes + 255) & ~(int64_t)255; }

bool promote_f32(const KernelArgs &ka, int dtype)
{
    const int override_bits = MPCQP_OPT_FORCE_LDS | MPCQP_OPT_FORCE_GWS | MPCQP_OPT_FORCE_DENSE_G | MPCQP_OPT_FORCE_CONDENSED |
                              MPCQP_OPT_ONE_PER_WAVE;
    if (dtype != MPCQP_F32 || (ka.opt_flags & override_bits) || ka.m < 1) return false;
    // The wide stage-wise kernel squares nothing, and BASELINE config 5 (n = 256) comes out 1e-6 from the float64 plan in
    // float32 -- but on adversarial mid-size families (bounds at the edge of consistency, 60-270 iterations) one plan in
    // ~1500 came back SOLVED 1.4e-3 .. 3.6e-3 away with every active row on its bound to rounding noise: the error sits in the
    // float32 Riccati sweeps themselves (V_a = P^-1 g_a'), where no residual this kernel can evaluate in float32 sees it.
    // Mid-size float32 problems (n <= kPromoteN) are therefore solved in float64 as well (1.5x the float32 time at
    // nx = 8, n = 40); float32 arithmetic is kept where it is what makes the size affordable.
    constexpr int kPromoteN = 160;
    if (use_stagew_auto(ka, MPCQP_F32)) return ka.n <= kPromoteN && use_stagew_auto(ka, MPCQP_F64);
    if (!fits_on_chip(ka, true, true, MODE_FUSED, MPCQP_F32) && !use_mid(ka, MPCQP_F32))
        // Round 5: whatever the general stage-wise kernel serves (nx <= 32, nu <= 8; float64 only) goes there, converted -- also
        // the float32 launches the dense HBM-resident path could hold (nx <= 16, 4 < nu <= 8, n <= 256): that path solves 330 k
        // problems/s where the general kernel, in float64, is an order of magnitude faster (tools/probe_dense_vs_general.py). The
        // dense solvers are left with nu > 8, MPCQP_OPT_FORCE_CONDENSED / _GWS / _DENSE_G and mpcqp_condense_batch + mpcqp_solve_batch.
        return stageg_supported(ka, MPCQP_F64);
    // ... and the float64 dispatch must have an on-chip / stage-wise kernel for it
    return pair_eligible(ka, MODE_FUSED, MPCQP_F64) || use_stage_auto(ka, MPCQP_F64) || use_stagew_auto(ka, MPCQP_F64) ||
           use_mid(ka, MPCQP_F64) || fits_on_chip(ka, true, true, MODE_FUSED, MPCQP_F64);
}

}  // namespace

extern "C" {

int mpcqp_abi_version(void) { return MPCQP_ABI_VERSION; }

const char *mpcqp_error_string(int code)
{
    switch (code) {
    case 0: return "ok";
    case MPCQP_EINVAL: return "invalid argument";
    case MPCQP_ETOOLARGE: return "no kernel for these dimensions (the stage-wise kernels serve nx <= 32, nu <= 8 at any horizon; the dense HBM-resident path any system with n <= 256)";
    case MPCQP_EDTYPE: return "dtype must be MPCQP_F64 or MPCQP_F32";
    case MPCQP_ELAYOUT: return "step stride must be 0 or the block size (float32 launches solved in float64: batch stride 0 or the packed size)";
    case MPCQP_EWORKSPACE: return "workspace missing or too small (see mpcqp_workspace_bytes)";
    case MPCQP_EUNSUPPORTED: return "option not available for these dimensions / this dtype (warm start: n <= 16, m <= 32, float64)";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

int mpcqp_lds_bytes(const MpcqpDims *dims, size_t *bytes)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!bytes) return MPCQP_EINVAL;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    Layout L = make_layout(ka.nx, ka.nu, ka.N, ka.n, ka.m, true, true, MODE_FUSED, elem_size(dims->dtype));
    *bytes = (size_t)L.total * elem_size(dims->dtype);
    return (*bytes > kLdsBytesPerCU || ka.n > 256) ? MPCQP_ETOOLARGE : 0;
}

int mpcqp_warm_state_bytes(const MpcqpDims *dims, size_t *bytes)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!bytes) return MPCQP_EINVAL;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    *bytes = warm_bytes_per_problem(ka, dims->dtype);
    return 0;
}

int mpcqp_warm_state_kind(const MpcqpDims *dims, int32_t *kind)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!kind) return MPCQP_EINVAL;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    int k = MPCQP_WARM_KIND_NONE;
    warm_bytes_per_problem(ka, dims->dtype, &k);
    *kind = k;
    return 0;
}

int mpcqp_workspace_bytes(const MpcqpDims *dims, int64_t batch, int32_t for_solve, size_t *bytes)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!bytes || batch < 0) return MPCQP_EINVAL;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    *bytes = 0;
    const int mode = for_solve ? MODE_FUSED : MODE_CONDENSE;
    // The query sees dimensions, not operand strides or MpcqpSolveOpts.flags, while the launch picks its kernel
    // with both (bulkier operands need more LDS): report the LARGEST workspace any path the launch may take needs.
    size_t need = 0;
    bool served = false;  // the automatic dispatch (fl == 0, first) has a kernel for these dimensions
    bool toolarge = false;
    for (int fl : kFlagVariants) {
        ka.opt_flags = fl;
        const bool maybe_mid = for_solve && problem_strides_unknown_mid(ka, dims->dtype);
        size_t v = 0;
        if (maybe_mid) v = bigsolve_ws_elems(ka.n) * elem_size(dims->dtype) * (size_t)batch;  // N* and M_A rows
        if (for_solve && use_stage_auto(ka, dims->dtype)) {
            const size_t sw = stage_ws_doubles(ka, stage_default_maxq(ka)) * sizeof(double) * (size_t)batch;
            if (sw > v) v = sw;
            if (second_opinion_applies(ka, dims->dtype, true)) {  // (the wide kernel behind the narrow one, in the same buffer)
                const size_t s2 = stagew_ws_elems(ka, stagew_auto_maxq(ka), dims->dtype) * elem_size(dims->dtype) * (size_t)batch;
                if (s2 > v) v = s2;
            }
        }
        if (for_solve && use_stage_long(ka, dims->dtype)) {
            const size_t sw = stage_ws_doubles(ka, stage_default_maxq(ka)) * sizeof(double) * (size_t)batch;
            if (sw > v) v = sw;
            served = true;
        }
        if (for_solve && use_stagew_auto(ka, dims->dtype)) {
            const size_t sw = stagew_ws_elems(ka, stagew_auto_maxq(ka), dims->dtype) * elem_size(dims->dtype) * (size_t)batch;
            if (sw > v) v = sw;
            served = true;
        } else if (!fits_on_chip(ka, true, true, mode, dims->dtype)) {
            if (big_supported(ka) && ka.n <= 256) {
                const BigPlan b = big_plan(ka, dims->dtype, true, for_solve != 0);
                const size_t big = b.total(for_solve != 0) * elem_size(dims->dtype) * (size_t)batch;
                if (big > v) v = big;
                if (for_solve && prefer_general(ka, dims->dtype)) {  // (the launch takes the general kernel: the larger of the two)
                    const size_t sg = stageg_ws_doubles(ka, stageg_default_maxq(ka)) * sizeof(double) * (size_t)batch;
                    if (sg > v) v = sg;
                }
                served = true;
            } else if (for_solve && stageg_supported(ka, dims->dtype)) {
                const size_t sg = stageg_ws_doubles(ka, stageg_default_maxq(ka)) * sizeof(double) * (size_t)batch;
                if (sg > v) v = sg;
                served = true;
            } else if (!maybe_mid && fl == 0 && !served) {
                // (only the automatic dispatch decides: an override combination that has no kernel for these
                // dimensions is refused by the launch that carries it, not by the size query)
                if (!(for_solve && dims->dtype == MPCQP_F32)) return MPCQP_ETOOLARGE;
                toolarge = true;  // (float32: unless the launch is one that is solved in float64, below)
            }
        }
        if (v > need) need = v;
    }
    if (for_solve && dims->dtype == MPCQP_F32) {
        // a float32 launch of an on-chip condensed kernel's size is solved in float64 on copies of its operands (promote_f32):
        // the copies at their largest (nothing shared), the float64 plan and multipliers, and the float64 launch's own scratch
        ka.opt_flags = 0;
        if (promote_f32(ka, MPCQP_F32)) {
            MpcqpDims d64 = *dims;
            d64.dtype = MPCQP_F64;
            size_t inner = 0;
            const int rc64 = mpcqp_workspace_bytes(&d64, batch, 1, &inner);
            if (rc64 == 0) {
                const int64_t N = ka.N, nx = ka.nx, nu = ka.nu, mk = ka.mk;
                const int64_t per = N * (nx * nx + nx * nu + mk * nx + mk * nu + mk) + 2 * nx + N * nx + ka.n + ka.m;
                const size_t v = (size_t)(per * 8 * batch + 10 * 256) + inner;  // (every segment starts 256-byte aligned)
                if (v > need) need = v;
                toolarge = false;
            }
        }
    }
    if (toolarge) return MPCQP_ETOOLARGE;
    *bytes = need;
    return 0;
}

int mpcqp_solve_workspace_bytes(int32_t n, int32_t m, int32_t dtype, int64_t batch, size_t *bytes)
{
    if (dtype != MPCQP_F64 && dtype != MPCQP_F32) return MPCQP_EDTYPE;
    if (n <= 0 || m < 0 || batch < 0 || !bytes) return MPCQP_EINVAL;
    KernelArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.n = n;
    ka.m = m;
    ka.nx = ka.nu = 1;
    ka.N = n;
    *bytes = 0;
    if (fits_on_chip(ka, false, false, MODE_SOLVE, dtype)) return 0;
    if (n > 256) return MPCQP_ETOOLARGE;
    size_t need = 0;
    for (int fl : kFlagVariants) {
        const size_t v = solver_ws_elems(n, m, dtype, fl) * elem_size(dtype) * (size_t)batch;
        if (v > need) need = v;
    }
    *bytes = need;
    return 0;
}

int mpcqp_condense_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch, void *P,
                         void *q, void *G, void *h, void *Phi, void *Psi, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if ((rc = check_problem(dims, problem))) return rc;
    if (batch < 0 || !P || !q || (dims->mk > 0 && (!G || !h))) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, problem);
    ka.P = P;
    ka.q = q;
    ka.G = G;
    ka.h = h;
    ka.Phi = Phi;
    ka.Psi = Psi;
    hipStream_t st = (hipStream_t)stream;
    Layout L;
    rc = layout_for(ka, problem->A.step_stride != 0, problem->B.step_stride != 0, MODE_CONDENSE, dims->dtype, L);
    if (rc == 0) {
        if ((rc = dispatch_lds<MODE_CONDENSE>(ka, L, dims->dtype, batch, st))) return rc;
    } else if (rc == MPCQP_ETOOLARGE && big_supported(ka)) {
        // HBM-resident path: Psi goes to the caller's Psi buffer when given, else to the workspace
        const size_t esz = elem_size(dims->dtype);
        const size_t psi_el = (size_t)(ka.N + 1) * ka.nx * ka.n * (size_t)batch;
        const size_t res_el = (size_t)(ka.N + 1) * ka.nx * (size_t)batch;
        const size_t need = (Psi ? res_el : psi_el + res_el) * esz;
        if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
        void *psi_ws = Psi ? Psi : workspace;
        void *res_ws = Psi ? workspace : (void *)((char *)workspace + psi_el * esz);
        if ((rc = launch_big_condense(ka, dims->dtype, batch, psi_ws, res_ws, P, q, G, h, nullptr, st))) return rc;
    } else {
        return rc;
    }
    if (Phi) rc = launch_phi(ka, dims->dtype, batch, st);
    return rc;
}

int mpcqp_update_vectors_batch(const MpcqpDims *dims, const MpcqpProblem *problem, const void *Phi,
                               int64_t phi_batch_stride, const void *Psi, int64_t psi_batch_stride,
                               int64_t batch, void *q, void *h, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!problem || !problem->x0.ptr || !Phi || batch < 0) return MPCQP_EINVAL;
    if (q && !Psi) return MPCQP_EINVAL;
    if (h && dims->mk > 0 && !problem->e.ptr) return MPCQP_EINVAL;
    if ((dims->flags & MPCQP_Q_TERMINAL) && q && !problem->goal.ptr) return MPCQP_EINVAL;
    if ((dims->flags & MPCQP_Q_STAGE) && q && !problem->targets.ptr) return MPCQP_EINVAL;
    if (batch == 0 || (!q && !h)) return 0;
    KernelArgs ka;
    fill_args(ka, dims, problem);
    ka.Phi = const_cast<void *>(Phi);
    ka.Psi = const_cast<void *>(Psi);
    ka.q = q;
    ka.h = h;
    return launch_update(ka, dims->dtype, phi_batch_stride, psi_batch_stride, batch, (hipStream_t)stream);
}

int mpcqp_condense_phase_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch, int32_t phase, void *P,
                               void *q, void *G, void *h, void *Psi, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if ((rc = check_problem(dims, problem))) return rc;
    if (batch < 0 || (phase != 1 && phase != 2) || !Psi) return MPCQP_EINVAL;
    if (phase == 1 && dims->mk > 0 && (!G || !h)) return MPCQP_EINVAL;
    if (phase == 2 && (!P || !q)) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, problem);
    ka.P = P;
    ka.q = q;
    ka.G = G;
    ka.h = h;
    ka.Psi = Psi;
    Layout L;
    // only where mpcqp_condense_batch itself runs as two launches: problems that do not fit a CU's LDS
    if (layout_for(ka, problem->A.step_stride != 0, problem->B.step_stride != 0, MODE_CONDENSE, dims->dtype, L) == 0 || !big_supported(ka))
        return MPCQP_EUNSUPPORTED;
    const size_t need = (size_t)(ka.N + 1) * ka.nx * (size_t)batch * elem_size(dims->dtype);  // the tracking residuals
    if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
    return launch_big_condense(ka, dims->dtype, batch, Psi, workspace, P, q, G, h, nullptr, (hipStream_t)stream, phase);
}

int mpcqp_solve_batch(int32_t n, int32_t m, int32_t dtype, const void *P, const void *q, const void *G,
                      const void *h, int64_t batch, const MpcqpSolveOpts *opts, void *x, void *lam,
                      int32_t *status, int32_t *iters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (dtype != MPCQP_F64 && dtype != MPCQP_F32) return MPCQP_EDTYPE;
    if (n <= 0 || m < 0 || batch < 0 || !P || !q || !x || (m > 0 && (!G || !h))) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.n = n;
    ka.m = m;
    ka.nx = 1;
    ka.nu = 1;
    ka.N = n;
    ka.P = const_cast<void *>(P);
    ka.q = const_cast<void *>(q);
    ka.G = const_cast<void *>(G);
    ka.h = const_cast<void *>(h);
    ka.U = x;
    ka.lam = lam;
    ka.status = status;
    ka.iters = iters;
    if (int rc = fill_opts(ka, opts, dtype)) return rc;
    if (ka.warm_state) return MPCQP_EUNSUPPORTED;
    if (fits_on_chip(ka, false, false, MODE_SOLVE, dtype))
        return run_solver<MODE_SOLVE>(ka, false, false, dtype, batch, (hipStream_t)stream);
    return run_gws_solve(ka, dtype, batch, workspace, workspace_bytes, (hipStream_t)stream);
}

int mpcqp_build_solve_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch,
                            const MpcqpSolveOpts *opts, void *U, void *lam, int32_t *status,
                            int32_t *iters, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if ((rc = check_problem(dims, problem))) return rc;
    if (batch < 0 || !U) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, problem);
    ka.U = U;
    ka.lam = lam;
    ka.status = status;
    ka.iters = iters;
    if ((rc = fill_opts(ka, opts, dims->dtype, true))) return rc;
    // a pairing order: cold launches of the small-problem fused kernel (float32 launches of its size arrive here converted)
    if (ka.order && (ka.warm_state || (ka.opt_flags & (MPCQP_OPT_FORCE_LDS | MPCQP_OPT_ONE_PER_WAVE | MPCQP_OPT_SEED_VIOLATED)) ||
                     !pair_eligible(ka, MODE_FUSED, MPCQP_F64)))
        return MPCQP_EUNSUPPORTED;
    // four problems per wavefront on request: only where that kernel applies (the dispatch picks it by batch size otherwise)
    if ((ka.opt_flags & MPCQP_OPT_FOUR_PER_WAVE) &&
        ((ka.opt_flags & (MPCQP_OPT_FORCE_LDS | MPCQP_OPT_ONE_PER_WAVE | MPCQP_OPT_TWO_PER_WAVE)) ||
         (!pair_eligible(ka, MODE_FUSED, MPCQP_F64) && ka.nx != 2 && ka.nx <= 4 && !quad4_applies(ka)) || (!quad_applies(ka) && !quad4_applies(ka))))
        return MPCQP_EUNSUPPORTED;
    const bool stepA = problem->A.step_stride != 0, stepB = problem->B.step_stride != 0;
    hipStream_t st = (hipStream_t)stream;
    if (promote_f32(ka, dims->dtype)) {
        // float32 at an on-chip condensed kernel's size: solved in float64 on converted copies (see promote_f32)
        const int64_t N = ka.N, nx = ka.nx, nu = ka.nu, mk = ka.mk;
        const MpcqpOperand *src[8] = {&problem->A, &problem->B, &problem->C, &problem->D, &problem->e, &problem->x0, &problem->goal, &problem->targets};
        const int64_t block[8] = {nx * nx, nx * nu, mk * nx, mk * nu, mk, nx, nx, N * nx};
        const bool per_step[8] = {true, true, true, true, true, false, false, false};
        MpcqpProblem p64 = *problem;
        MpcqpOperand *dst[8] = {&p64.A, &p64.B, &p64.C, &p64.D, &p64.e, &p64.x0, &p64.goal, &p64.targets};
        ConvPlan in{};
        in.to_double = 1;
        char *w = (char *)workspace;
        int64_t off = 0;
        for (int i = 0; i < 8; ++i) {
            if (!operand_elems(*src[i], block[i], (int)N, batch, per_step[i])) continue;
            // (mpcqp_workspace_bytes prices the copies densely packed. A source with a PADDED batch stride is packed on the way --
            // round 6; until then it was refused, MPCQP_ELAYOUT --; a padded step stride still is: no caller of this library makes one)
            const int64_t dense = (per_step[i] && src[i]->step_stride) ? N * block[i] : block[i];
            if (per_step[i] && src[i]->step_stride && src[i]->step_stride != block[i]) return MPCQP_ELAYOUT;
            const bool padded = src[i]->batch_stride != 0 && src[i]->batch_stride != dense;
            if (src[i]->batch_stride != 0 && src[i]->batch_stride < dense) return MPCQP_ELAYOUT;
            const int64_t cnt = src[i]->batch_stride ? batch * dense : dense;
            if (w) dst[i]->ptr = w + off;
            if (padded) dst[i]->batch_stride = dense;
            ConvSeg sg{src[i]->ptr, w ? w + off : nullptr, cnt};
            if (padded) {
                sg.row = dense;
                sg.src_stride = src[i]->batch_stride;
            }
            in.seg[in.nseg++] = sg;
            off += al256(cnt * 8);
        }
        const int64_t offU = off;
        off += al256(batch * ka.n * 8);
        const int64_t offL = off;
        if (lam) off += al256(batch * ka.m * 8);
        MpcqpDims d64 = *dims;
        d64.dtype = MPCQP_F64;
        size_t inner = 0;
        if ((rc = mpcqp_workspace_bytes(&d64, batch, 1, &inner))) return rc;
        if (!workspace || workspace_bytes < (size_t)off + inner) return MPCQP_EWORKSPACE;
        if ((rc = launch_convert(in, st))) return rc;
        MpcqpSolveOpts o64{};
        if (opts) o64 = *opts;
        if (!(o64.feas_tol > 0.0)) o64.feas_tol = 1e-9;  // (float64 arithmetic; the float32 default of 1e-5 would only loosen the plan)
        rc = mpcqp_build_solve_batch(&d64, &p64, batch, &o64, w + offU, lam ? w + offL : nullptr, status, iters,
                                     inner ? w + off : nullptr, inner, stream);
        if (rc) return rc;
        ConvPlan out{};
        out.to_double = 0;
        out.seg[out.nseg++] = ConvSeg{w + offU, U, batch * ka.n};
        if (lam) out.seg[out.nseg++] = ConvSeg{w + offL, lam, batch * ka.m};
        return launch_convert(out, st);
    }
    if (ka.warm_state && !pair_eligible(ka, MODE_FUSED, dims->dtype) && !use_stage_auto(ka, dims->dtype) && !use_stagew_auto(ka, dims->dtype))
        return MPCQP_EUNSUPPORTED;
    // (row-id warm starts: the pair kernel and the wide stage-wise kernel; the narrow stage-wise kernel, which the dispatch
    // prefers for small systems with 16 < n <= 128, has its own kind of record)
    if (ka.warm_start == MPCQP_WARM_ACTIVE_SET && !pair_eligible(ka, MODE_FUSED, dims->dtype) &&
        (use_stage_auto(ka, dims->dtype) || !use_stagew_auto(ka, dims->dtype)))
        return MPCQP_EUNSUPPORTED;
    if ((ka.opt_flags & MPCQP_OPT_PIPELINE_FACTOR) && !(use_stage_auto(ka, dims->dtype) && stage_pipeline_supported(ka, dims->dtype)))
        return MPCQP_EUNSUPPORTED;
    // the state is indexed by problem: a buffer made for a smaller batch would be read and written out of bounds
    if (ka.warm_state && ka.warm_state_bytes < (size_t)batch * warm_bytes_per_problem(ka, dims->dtype)) return MPCQP_EWORKSPACE;
    if (use_stage_auto(ka, dims->dtype)) {
        const int maxq = stage_default_maxq(ka);
        size_t need = stage_ws_doubles(ka, maxq) * sizeof(double) * (size_t)batch;
        const bool second = second_opinion_applies(ka, dims->dtype);
        const int maxq2 = stagew_auto_maxq(ka);
        if (second) {
            const size_t need2 = stagew_ws_elems(ka, maxq2, dims->dtype) * elem_size(dims->dtype) * (size_t)batch;
            need = need2 > need ? need2 : need;
        }
        if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
        if ((rc = launch_stage(ka, maxq, batch, workspace, st)) || !second) return rc;
        // the second opinion (mpcqp_internal.h, kOptSecondOpinion): same stream, same workspace (the first launch is done with it)
        KernelArgs kb = ka;
        kb.opt_flags |= kOptSecondOpinion;
        kb.probe = nullptr;
        rc = launch_stagew(kb, dims->dtype, maxq2, batch, workspace, st);
        return rc == MPCQP_ETOOLARGE ? 0 : rc;  // (a horizon beyond the wide kernel's 32-bit offsets: the narrow kernel's verdicts stand)
    }
    if (use_stage_long(ka, dims->dtype)) {
        const int maxq = stage_default_maxq(ka);
        const size_t need = stage_ws_doubles(ka, maxq) * sizeof(double) * (size_t)batch;
        if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
        return launch_stage(ka, maxq, batch, workspace, st);
    }
    if (use_stagew_auto(ka, dims->dtype)) {
        const int maxq = stagew_auto_maxq(ka);
        const size_t need = stagew_ws_elems(ka, maxq, dims->dtype) * elem_size(dims->dtype) * (size_t)batch;
        if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
        return launch_stagew(ka, dims->dtype, maxq, batch, workspace, st);
    }
    if (use_mid(ka, dims->dtype)) {
        const size_t need = bigsolve_ws_elems(ka.n) * elem_size(dims->dtype) * (size_t)batch;
        if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
        return launch_mid(ka, dims->dtype, batch, workspace, st);
    }
    if (fits_on_chip(ka, stepA, stepB, MODE_FUSED, dims->dtype))
        return run_solver<MODE_FUSED>(ka, stepA, stepB, dims->dtype, batch, st);
    // HBM-resident path: propagate + Gram (MFMA for f32) into the workspace, then the
    // general solver with its arrays in the workspace as well
    if (!big_supported(ka) || ka.n > 256 || prefer_general(ka, dims->dtype)) {
        // wide systems on horizons the dense path cannot hold -- and every float64 problem it could hold (prefer_general): the
        // general stage-wise kernel (float64; float32 launches of these dimensions arrive here converted, promote_f32)
        if (!stageg_supported(ka, dims->dtype) || ka.warm_state) return MPCQP_ETOOLARGE;
        const int maxq = stageg_default_maxq(ka);
        const size_t need = stageg_ws_doubles(ka, maxq) * sizeof(double) * (size_t)batch;
        if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
        return launch_stageg(ka, maxq, batch, workspace, st);
    }
    const size_t esz = elem_size(dims->dtype), nb = (size_t)batch;
    const BigPlan b = big_plan(ka, dims->dtype, true, true);
    if (!workspace || workspace_bytes < b.total(true) * esz * nb) return MPCQP_EWORKSPACE;
    char *w = (char *)workspace;
    void *psi_ws = w;
    void *res_ws = w + (size_t)(ka.N + 1) * ka.nx * ka.n * nb * esz;
    w += b.psi * nb * esz;
    void *Pw = w;
    w += b.P * nb * esz;
    void *qw = w;
    w += b.q * nb * esz;
    void *Gw = w;
    w += b.G * nb * esz;
    void *hw = w;
    w += b.h * nb * esz;
    if (b.matrix_free) {
        void *nw = w;
        w += b.nrm * nb * esz;
        if ((rc = launch_big_condense(ka, dims->dtype, batch, psi_ws, res_ws, Pw, qw, nullptr, hw, nw, st))) return rc;
        return launch_bigsolve_struct(ka, dims->dtype, batch, Pw, qw, psi_ws, hw, nw, w, st);
    }
    if ((rc = launch_big_condense(ka, dims->dtype, batch, psi_ws, res_ws, Pw, qw, Gw, hw, nullptr, st))) return rc;
    ka.P = Pw;
    ka.q = qw;
    ka.G = Gw;
    ka.h = hw;
    return run_gws_solve(ka, dims->dtype, batch, w, b.solver * nb * esz, st);
}

int mpcqp_stagewise_workspace_bytes(const MpcqpDims *dims, int64_t batch, int32_t max_active, size_t *bytes)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!bytes || batch < 0) return MPCQP_EINVAL;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    if (max_active < 0) {  // the general kernel's workspace (MPCQP_OPT_STAGE_GENERAL): -1 default slots, -k: k slots
        if (!stageg_supported(ka, dims->dtype)) return MPCQP_EUNSUPPORTED;
        *bytes = stageg_ws_doubles(ka, max_active < -1 ? -max_active : stageg_default_maxq(ka)) * sizeof(double) * (size_t)batch;
        return 0;
    }
    const int maxq = max_active > 0 ? max_active : stage_default_maxq(ka);
    // (the query does not see MpcqpSolveOpts.flags: where both kernels apply it reports the larger workspace)
    size_t a = 0, b = 0;
    if (stage_supported(ka, dims->dtype)) a = stage_ws_doubles(ka, maxq) * sizeof(double) * (size_t)batch;
    if (stagew_supported(ka, dims->dtype)) b = stagew_ws_elems(ka, maxq, dims->dtype) * elem_size(dims->dtype) * (size_t)batch;
    if (!stage_supported(ka, dims->dtype) && !stagew_supported(ka, dims->dtype)) {
        if (!stageg_supported(ka, dims->dtype)) return MPCQP_EUNSUPPORTED;
        *bytes = stageg_ws_doubles(ka, max_active > 0 ? max_active : stageg_default_maxq(ka)) * sizeof(double) * (size_t)batch;
        return 0;
    }
    if (!a && !b && batch > 0) return MPCQP_EUNSUPPORTED;
    *bytes = a > b ? a : b;
    return 0;
}

int mpcqp_stagewise_solve_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch,
                                const MpcqpSolveOpts *opts, int32_t max_active, void *U, void *lam, int32_t *status,
                                int32_t *iters, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if ((rc = check_problem(dims, problem))) return rc;
    if (batch < 0 || !U) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, problem);
    bool narrow = stage_supported(ka, dims->dtype);
    const bool general = opts && (opts->flags & MPCQP_OPT_STAGE_GENERAL);
    if (general || (!narrow && !stagew_supported(ka, dims->dtype))) {
        if (!stageg_supported(ka, dims->dtype)) return MPCQP_EUNSUPPORTED;  // (float32: through mpcqp_build_solve_batch, which converts)
        ka.U = U;
        ka.lam = lam;
        ka.status = status;
        ka.iters = iters;
        if ((rc = fill_opts(ka, opts, dims->dtype))) return rc;
        if (ka.warm_state) return MPCQP_EUNSUPPORTED;
        const int mq = max_active > 0 ? max_active : stageg_default_maxq(ka);
        if (!workspace || workspace_bytes < stageg_ws_doubles(ka, mq) * sizeof(double) * (size_t)batch) return MPCQP_EWORKSPACE;
        return launch_stageg(ka, mq, batch, workspace, (hipStream_t)stream);
    }
    if (opts && (opts->flags & MPCQP_OPT_STAGE_WIDE) && stagew_supported(ka, dims->dtype)) narrow = false;
    ka.U = U;
    ka.lam = lam;
    ka.status = status;
    ka.iters = iters;
    if ((rc = fill_opts(ka, opts, dims->dtype))) return rc;
    if (ka.warm_state) return MPCQP_EUNSUPPORTED;
    const int maxq = max_active > 0 ? max_active : stage_default_maxq(ka);
    const size_t need_w = stagew_supported(ka, dims->dtype) ? stagew_ws_elems(ka, maxq, dims->dtype) * elem_size(dims->dtype) * (size_t)batch : 0;
    // (as in mpcqp_build_solve_batch: what the narrow kernel leaves MPCQP_MAX_ITER / MPCQP_INFEASIBLE goes through the wide one, with
    // the same slots, in the same workspace -- mpcqp_stagewise_workspace_bytes reports the larger of the two)
    const bool second = narrow && second_opinion_applies(ka, dims->dtype);
    size_t need = narrow ? stage_ws_doubles(ka, maxq) * sizeof(double) * (size_t)batch : need_w;
    if (second && need_w > need) need = need_w;
    if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
    if (!narrow) return launch_stagew(ka, dims->dtype, maxq, batch, workspace, (hipStream_t)stream);
    if ((rc = launch_stage(ka, maxq, batch, workspace, (hipStream_t)stream)) || !second) return rc;
    KernelArgs kb = ka;
    kb.opt_flags |= kOptSecondOpinion;
    kb.probe = nullptr;
    rc = launch_stagew(kb, dims->dtype, maxq, batch, workspace, (hipStream_t)stream);
    return rc == MPCQP_ETOOLARGE ? 0 : rc;  // (a horizon beyond the wide kernel's 32-bit offsets: the narrow kernel's verdicts stand)
}

int mpcqp_model_bytes(const MpcqpDims *dims, size_t *bytes)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!bytes) return MPCQP_EINVAL;
    const ModelLayout ml = make_model_layout(dims->nx, dims->N, dims->N * dims->nu, dims->N * dims->mk);
    *bytes = (ml.total + 4) * elem_size(dims->dtype);
    return 0;
}

int mpcqp_factor_model(const MpcqpDims *dims, const void *P, const void *G, const void *q_basis,
                       const void *h_basis, void *model, size_t model_bytes, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!P || !q_basis || !model || (dims->mk > 0 && (!G || !h_basis))) return MPCQP_EINVAL;
    size_t need = 0;
    mpcqp_model_bytes(dims, &need);
    if (model_bytes < need) return MPCQP_EWORKSPACE;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    return launch_factor_model(ka, dims->dtype, P, G, q_basis, h_basis, model, (hipStream_t)stream);
}

int mpcqp_solve_model_batch(const MpcqpDims *dims, const void *model, const MpcqpOperand *x0,
                            const MpcqpOperand *goal, const MpcqpOperand *targets, int64_t batch,
                            const MpcqpSolveOpts *opts, void *U, void *lam, int32_t *status, int32_t *iters,
                            void *stream)
{
    return mpcqp_solve_model_bounds_batch(dims, model, nullptr, x0, goal, targets, batch, opts, U, lam, status, iters,
                                          stream);
}

int mpcqp_solve_model_bounds_batch(const MpcqpDims *dims, const void *model, const MpcqpOperand *e,
                                   const MpcqpOperand *x0, const MpcqpOperand *goal, const MpcqpOperand *targets,
                                   int64_t batch, const MpcqpSolveOpts *opts, void *U, void *lam, int32_t *status,
                                   int32_t *iters, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!model || !x0 || !x0->ptr || !U || batch < 0) return MPCQP_EINVAL;
    if ((dims->flags & MPCQP_Q_TERMINAL) && !(goal && goal->ptr)) return MPCQP_EINVAL;
    if ((dims->flags & MPCQP_Q_STAGE) && !(targets && targets->ptr)) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    ka.x0 = *x0;
    if (goal) ka.goal = *goal;
    if (targets) ka.targets = *targets;
    ka.model = model;
    ka.U = U;
    ka.lam = lam;
    ka.status = status;
    ka.iters = iters;
    if ((rc = fill_opts(ka, opts, dims->dtype, true))) return rc;
    if (ka.warm_state) return MPCQP_EUNSUPPORTED;
    // (a pairing order: the pair kernel's model mode only)
    if (ka.order && ((ka.opt_flags & (MPCQP_OPT_FORCE_LDS | MPCQP_OPT_ONE_PER_WAVE)) || !pair_eligible(ka, MODE_MODEL, dims->dtype)))
        return MPCQP_EUNSUPPORTED;
    const bool small = pair_eligible(ka, MODE_MODEL, dims->dtype);
    if ((ka.opt_flags & MPCQP_OPT_FOUR_PER_WAVE) &&
        ((ka.opt_flags & (MPCQP_OPT_FORCE_LDS | MPCQP_OPT_ONE_PER_WAVE | MPCQP_OPT_TWO_PER_WAVE | MPCQP_OPT_SEED_VIOLATED)) || !small))
        return MPCQP_EUNSUPPORTED;  // (the four-per-wavefront kernel has no seed steps: asked for by name, it is refused by name)
    hipStream_t st = (hipStream_t)stream;
    const bool own_e = e && e->ptr;  // per-problem bounds: the small-problem kernels' model mode only
    if (own_e) {
        if (ka.m < 1) return MPCQP_EINVAL;
        ka.e = *e;
        if (!small) return MPCQP_EUNSUPPORTED;
        return quad_model_eligible(ka, batch) ? launch_quad_model(ka, batch, st) : launch_pair_model(ka, batch, st);
    }
    if (!force_lds(ka.opt_flags) && !(ka.opt_flags & MPCQP_OPT_ONE_PER_WAVE) && small)
        return quad_model_eligible(ka, batch) ? launch_quad_model(ka, batch, st) : launch_pair_model(ka, batch, st);
    if (!force_lds(ka.opt_flags) && w64_eligible(ka, MODE_MODEL, dims->dtype)) return launch_w64(ka, MODE_MODEL, dims->dtype, batch, st);
    Layout L;
    if ((rc = layout_for(ka, false, false, MODE_SOLVE, dims->dtype, L))) return rc;
    return dispatch_lds<MODE_MODEL>(ka, L, dims->dtype, batch, st);
}

int mpcqp_rollout_batch(const MpcqpDims *dims, const MpcqpOperand *A, const MpcqpOperand *B,
                        const MpcqpOperand *x0, const void *U, int64_t batch, void *X, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!A || !B || !x0 || !A->ptr || !B->ptr || !x0->ptr || !U || !X || batch < 0) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, nullptr);
    ka.A = *A;
    ka.B = *B;
    ka.x0 = *x0;
    ka.U = const_cast<void *>(U);
    ka.X = X;
    return launch_rollout(ka, dims->dtype, batch, (hipStream_t)stream);
}

int mpcqp_wip_period_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch, const MpcqpSolveOpts *opts,
                           void *U, void *lam, int32_t *status, int32_t *iters, void *workspace, size_t workspace_bytes,
                           void *states, int64_t *loop_stats, double sampling_period, double target_vel, double length,
                           double gravity, int32_t nsub, void *stream)
{
    return mpcqp_wip_periods_batch(dims, problem, batch, opts, U, lam, status, iters, workspace, workspace_bytes, states,
                                   loop_stats, sampling_period, target_vel, length, gravity, nsub, 1, stream);
}

int mpcqp_wip_periods_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch, const MpcqpSolveOpts *opts,
                            void *U, void *lam, int32_t *status, int32_t *iters, void *workspace, size_t workspace_bytes,
                            void *states, int64_t *loop_stats, double sampling_period, double target_vel, double length,
                            double gravity, int32_t nsub, int32_t nperiods, void *stream)
{
    if (nperiods < 1) return MPCQP_EINVAL;
    int rc = check_dims(dims);
    if (rc) return rc;
    if ((rc = check_problem(dims, problem))) return rc;
    if (batch < 0 || !U || !states || nsub <= 0 || !(length > 0) || !(gravity > 0)) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    KernelArgs ka;
    fill_args(ka, dims, problem);
    ka.U = U;
    ka.lam = lam;
    ka.status = status;
    ka.iters = iters;
    if ((rc = fill_opts(ka, opts, dims->dtype))) return rc;
    // the plant is the 4-state, 1-input pendulum, every loop with its own x0, goal and targets; only the stage-wise
    // kernel carries the epilogue
    if (ka.nx != 4 || ka.nu != 1 || !problem->x0.ptr || !problem->goal.ptr || !problem->targets.ptr ||
        problem->x0.batch_stride != 4 || problem->goal.batch_stride != 4 || problem->targets.batch_stride != (int64_t)ka.N * 4 ||
        !use_stage_auto(ka, dims->dtype))
        return MPCQP_EUNSUPPORTED;
    const int maxq = stage_default_maxq(ka);
    const size_t need = stage_ws_doubles(ka, maxq) * sizeof(double) * (size_t)batch;
    if (!workspace || workspace_bytes < need) return MPCQP_EWORKSPACE;
    // (as in mpcqp_build_solve_batch: the warm-state record is indexed by problem and written by the kernel, so a buffer
    // that is too small -- or an old caller that leaves warm_state_bytes at zero -- is refused before anything is launched)
    if (ka.warm_state && ka.warm_state_bytes < (size_t)batch * warm_bytes_per_problem(ka, dims->dtype)) return MPCQP_EWORKSPACE;
    // several periods per launch: a factor that is kept is the FIRST period's business (the caller's next launch reuses
    // or pipelines it), and the warm-state record is per launch
    if (nperiods > 1 && ((ka.opt_flags & MPCQP_OPT_KEEP_FACTOR) || ka.warm_state || !stage_pipeline_supported(ka, dims->dtype)))
        return MPCQP_EUNSUPPORTED;  // (... and the period loop is compiled into the short-horizon instantiations only)
    ka.ep_on = 1;
    ka.ep_periods = nperiods;
    ka.ep_nsub = nsub;
    ka.ep_Tp = sampling_period;
    ka.ep_vel = target_vel;
    ka.ep_omega2 = gravity / length;
    ka.ep_g = gravity;
    ka.ep_states = states;
    ka.ep_loopstats = (long long *)loop_stats;
    return launch_stage(ka, maxq, batch, workspace, (hipStream_t)stream);
}

int mpcqp_wip_advance_stats_batch(int32_t dtype, void *states, const void *U, int64_t u_stride, const int32_t *status,
                                  const int32_t *iters, int64_t *stats, int32_t N, double sampling_period,
                                  double target_vel, double length, double gravity, int32_t nsub, void *x0, void *goal,
                                  void *targets, int64_t batch, void *stream)
{
    if (dtype != MPCQP_F64 && dtype != MPCQP_F32) return MPCQP_EDTYPE;
    if (!states || !U || !x0 || !goal || !targets || N <= 0 || nsub < 0 || batch < 0 || !(length > 0) || !(gravity > 0))
        return MPCQP_EINVAL;
    if (stats && !status) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    return launch_wip_advance(dtype, states, U, u_stride, status, iters, stats, N, sampling_period, target_vel, length,
                              gravity, nsub, x0, goal, targets, batch, (hipStream_t)stream);
}

int mpcqp_wip_advance_batch(int32_t dtype, void *states, const void *U, int64_t u_stride, const int32_t *status,
                            int32_t N, double sampling_period, double target_vel, double length, double gravity,
                            int32_t nsub, void *x0, void *goal, void *targets, int64_t batch, void *stream)
{
    return mpcqp_wip_advance_stats_batch(dtype, states, U, u_stride, status, nullptr, nullptr, N, sampling_period,
                                         target_vel, length, gravity, nsub, x0, goal, targets, batch, stream);
}

int mpcqp_accumulate_stats(const int32_t *status, const int32_t *iters, int64_t batch, int64_t *stats, void *stream)
{
    if (!status || !iters || !stats || batch < 0) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    return launch_stats(status, iters, batch, stats, (hipStream_t)stream);
}

// ---- mpcqp_order_by_count: a counting sort of the batch by (clamped) count, longest first. Three small launches: per-chunk
// histograms, one workgroup that turns them into start offsets (bucket-major, chunk-minor), the scatter. Inside one (bucket, chunk)
// cell the places are handed out by an LDS atomic, so the order of equal counts within a chunk is not reproducible -- it is a
// pairing hint, every order is a valid one.
namespace {
constexpr int kOrdBuckets = 1024, kOrdChunk = 4096, kOrdThreads = 256;
__device__ __forceinline__ int ord_bucket(int c) { return kOrdBuckets - 1 - (c < 0 ? 0 : (c > kOrdBuckets - 1 ? kOrdBuckets - 1 : c)); }

__global__ void __launch_bounds__(kOrdThreads) order_hist_kernel(const int32_t *__restrict__ counts, int64_t batch, int32_t *__restrict__ hist, int chunks)
{
    __shared__ int h[kOrdBuckets];
    for (int b = threadIdx.x; b < kOrdBuckets; b += kOrdThreads) h[b] = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * kOrdChunk;
    for (int i = threadIdx.x; i < kOrdChunk && first + i < batch; i += kOrdThreads) atomicAdd(&h[ord_bucket(counts[first + i])], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < kOrdBuckets; b += kOrdThreads) hist[(int64_t)blockIdx.x * kOrdBuckets + b] = h[b];
}

__global__ void __launch_bounds__(kOrdBuckets) order_scan_kernel(int32_t *__restrict__ hist, int chunks)
{
    __shared__ int tot[kOrdBuckets];
    const int b = threadIdx.x;
    int sum = 0;
#pragma unroll 8
    for (int g = 0; g < chunks; ++g) sum += hist[(int64_t)g * kOrdBuckets + b];  // (independent, coalesced loads)
    tot[b] = sum;
    __syncthreads();
    for (int d = 1; d < kOrdBuckets; d <<= 1) {  // inclusive scan over the buckets
        const int v = b >= d ? tot[b - d] : 0;
        __syncthreads();
        tot[b] += v;
        __syncthreads();
    }
    int run = tot[b] - sum;
#pragma unroll 8
    for (int g = 0; g < chunks; ++g) {
        const int c = hist[(int64_t)g * kOrdBuckets + b];
        hist[(int64_t)g * kOrdBuckets + b] = run;
        run += c;
    }
}

__global__ void __launch_bounds__(kOrdThreads) order_scatter_kernel(const int32_t *__restrict__ counts, int64_t batch, const int32_t *__restrict__ hist,
                                                                     int chunks, int32_t *__restrict__ order)
{
    __shared__ int off[kOrdBuckets];
    for (int b = threadIdx.x; b < kOrdBuckets; b += kOrdThreads) off[b] = hist[(int64_t)blockIdx.x * kOrdBuckets + b];
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * kOrdChunk;
    for (int i = threadIdx.x; i < kOrdChunk && first + i < batch; i += kOrdThreads) {
        const int at = atomicAdd(&off[ord_bucket(counts[first + i])], 1);
        order[at] = (int32_t)(first + i);
    }
}
}  // namespace

size_t mpcqp_order_workspace_bytes(int64_t batch)
{
    return batch <= 0 ? 0 : (size_t)kOrdBuckets * sizeof(int32_t) * (size_t)((batch + kOrdChunk - 1) / kOrdChunk);
}

int mpcqp_order_by_count(const int32_t *counts, int64_t batch, int32_t *order, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!counts || !order || batch < 0 || batch > INT32_MAX) return MPCQP_EINVAL;
    if (batch == 0) return 0;
    if (!workspace || workspace_bytes < mpcqp_order_workspace_bytes(batch)) return MPCQP_EWORKSPACE;
    const int chunks = (int)((batch + kOrdChunk - 1) / kOrdChunk);
    hipStream_t st = (hipStream_t)stream;
    int32_t *hist = (int32_t *)workspace;
    hipLaunchKernelGGL(order_hist_kernel, dim3(chunks), dim3(kOrdThreads), 0, st, counts, batch, hist, chunks);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(kOrdBuckets), 0, st, hist, chunks);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(chunks), dim3(kOrdThreads), 0, st, counts, batch, hist, chunks, order);
    return (int)hipGetLastError();
}

int mpcqp_lipm_advance_stats_batch(int32_t dtype, void *states, const void *U, int64_t u_stride, const int32_t *status,
                                   const int32_t *iters, int64_t *stats, int32_t N, double sampling_period, int32_t nsub,
                                   int32_t nb_dsp, int32_t nb_ssp, double max_zmp_dist, int64_t *index,
                                   int64_t *stride_index, void *support, const void *strides, const void *foot_size,
                                   void *x0, void *goal, void *e, int64_t batch, void *stream)
{
    if (dtype != MPCQP_F64 && dtype != MPCQP_F32) return MPCQP_EDTYPE;
    if (!states || !index || !stride_index || !support || !strides || !foot_size || !x0 || !goal || !e || N <= 0 ||
        nsub <= 0 || nb_dsp < 0 || nb_ssp < 0 || batch < 0)
        return MPCQP_EINVAL;
    if (stats && !status) return MPCQP_EINVAL;
    if (2 * (nb_dsp + nb_ssp) < N) return MPCQP_EINVAL;  // more than two steps in the receding horizon
    if (batch == 0) return 0;
    return launch_lipm_advance(dtype, states, U, u_stride, status, N, sampling_period, nsub, nb_dsp, nb_ssp,
                               max_zmp_dist, index, stride_index, support, strides, foot_size, x0, goal, e, batch,
                               iters, stats, (hipStream_t)stream);
}

int mpcqp_lipm_advance_batch(int32_t dtype, void *states, const void *U, int64_t u_stride, const int32_t *status,
                             int32_t N, double sampling_period, int32_t nsub, int32_t nb_dsp, int32_t nb_ssp,
                             double max_zmp_dist, int64_t *index, int64_t *stride_index, void *support,
                             const void *strides, const void *foot_size, void *x0, void *goal, void *e,
                             int64_t batch, void *stream)
{
    return mpcqp_lipm_advance_stats_batch(dtype, states, U, u_stride, status, nullptr, nullptr, N, sampling_period, nsub,
                                          nb_dsp, nb_ssp, max_zmp_dist, index, stride_index, support, strides,
                                          foot_size, x0, goal, e, batch, stream);
}

}  // extern "C"
