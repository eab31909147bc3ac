/*
 * mpcqp.h -- C ABI of libmpcqp_hip.so, the MI355X (gfx950) implementation of
 * qpmpc's hot path:  MPCProblem -> MPCQP (condense) -> dense QP solve -> inputs.
 *
 * The reference (stephane-caron/qpmpc v3.1.0) is pure Python and has no FFI
 * layer of its own; its operator boundary for this path is
 *     solve_mpc(problem, solver, sparse=False, **kw) -> Plan   qpmpc/solve_mpc.py:16-44
 *     MPCQP(problem, sparse=False)                             qpmpc/mpc_qp.py:39-122
 * Each entry point below names the reference code it replaces. A Python
 * maintainer binds them with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the
 *    caller (the library never allocates or frees device memory); problems that do
 *    not fit the on-chip path need a caller-provided scratch `workspace` whose size
 *    the *_workspace_bytes functions report (0 for the on-chip path; NULL is then fine);
 *  - every call is asynchronous on the caller's hipStream_t (passed as void*;
 *    NULL = the null stream), re-entrant, no global state (nothing is read from the
 *    process environment; dispatch overrides are explicit MpcqpSolveOpts.flags);
 *  - return value: 0 ok, <0 bad argument (MPCQP_E*), >0 a hipError_t;
 *    per-problem outcome is reported in status[b], never by the return value;
 *  - matrices are row-major and densely packed; one batch item after another
 *    unless a stride says otherwise;
 *  - constraints are  G u <= h ; the QP is  min 1/2 u'Pu + q'u.
 */
#ifndef MPCQP_H_
#define MPCQP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCQP_ABI_VERSION 11

/* element type of every floating-point buffer of a call. It is the STORAGE type: mpcqp_build_solve_batch computes
 * MPCQP_F32 problems of at most 160 variables (and every float32 problem only the general stage-wise kernel serves) in
 * float64 on copies of the operands converted into the workspace -- the float32 condensed kernels square the
 * conditioning into P and returned plans 2e-2 from the float64 ones as solved on ill-conditioned problems;
 * larger float32 problems (BASELINE config 5) run the float32 stage-wise kernel, whose acceptance test is 16 tol (1 + |e|)
 * on the active rows. Contract either way: |u - u_float64| <= 1e-3 max(1, |u|), or status != MPCQP_SOLVED. */
#define MPCQP_F64 0
#define MPCQP_F32 1

/* MpcqpDims.flags -- which cost terms enter P and q. They differ on purpose:
 * the reference adds a term to P when its weight "is not None"
 * (mpc_qp.py:102,104) but to q only when weight > 1e-10 and the matching
 * state is defined (mpc_problem.py:141-166, mpc_qp.py:119-122). */
#define MPCQP_P_TERMINAL 1
#define MPCQP_P_STAGE 2
#define MPCQP_Q_TERMINAL 4
#define MPCQP_Q_STAGE 8

/* per-problem status[b]  (Plan.is_empty <=> status != 0, plan.py:35-40) */
#define MPCQP_SOLVED 0
#define MPCQP_MAX_ITER 1
#define MPCQP_INFEASIBLE 2
#define MPCQP_NOT_PD 3
#define MPCQP_SLOTS_FULL 4 /* stage-wise kernels only: more rows wanted to be active at once than the launch's max_active
                              slots hold (possible when max_active < min(n, m), i.e. n, m > 128 / 256 with the defaults);
                              solve the problem again with a larger max_active (the Python host does: solve_mpc_batch) */

/* negative return codes */
#define MPCQP_EINVAL (-1)    /* NULL/negative/inconsistent argument           */
#define MPCQP_ETOOLARGE (-2) /* no kernel for these dimensions: does not fit a CU's LDS and nx > 32 or nu > 8 with n > 256
                                (ABI 8: systems with nx <= 32, nu <= 8 are served at any horizon -- the general stage-wise
                                kernel, float64 arithmetic, takes what the MFMA stage-wise kernels (nx <= 16, nu <= 4) and
                                the dense path (n <= 256) do not: qpmpc/solve_mpc.py:42-44 accepts any dimension) */
#define MPCQP_EDTYPE (-3)    /* dtype not MPCQP_F64 / MPCQP_F32               */
#define MPCQP_ELAYOUT (-4)   /* step stride is neither 0 nor the block size, or a batch stride smaller than a problem's block
                                (a float32 launch that is solved in float64 -- at most 160 variables -- takes any larger batch
                                stride since ABI 11: the conversion packs the operands) */
#define MPCQP_EWORKSPACE (-5) /* workspace missing or too small (see *_workspace_bytes) */
#define MPCQP_EUNSUPPORTED (-6) /* option not available for these dimensions / this dtype    */

/* Problem dimensions and cost weights (mpc_problem.py:88-139).
 * n = N*nu decision variables, m = N*mk inequality rows. Problems whose
 * per-step row count varies are padded by the host to mk rows with C=D=0 and
 * e=+1e30 (an inequality that can never be active). */
typedef struct MpcqpDims {
    int32_t nx;    /* state_dim                                   */
    int32_t nu;    /* input_dim                                   */
    int32_t N;     /* nb_timesteps                                */
    int32_t mk;    /* inequality rows per step (after padding)    */
    int32_t dtype; /* MPCQP_F64 | MPCQP_F32                       */
    int32_t flags; /* MPCQP_P_* | MPCQP_Q_*                       */
    double w_terminal; /* terminal_cost_weight     (ignored unless flagged) */
    double w_stage;    /* stage_state_cost_weight  (ignored unless flagged) */
    double w_input;    /* stage_input_cost_weight  > 0                      */
} MpcqpDims;

/* One operand of a batch of problems, addressed as
 *     ptr + b*batch_stride + k*step_stride      (strides in ELEMENTS)
 * batch_stride == 0: shared by every problem of the batch;
 * step_stride  == 0: time-invariant (the reference's "array, not list" case,
 *                    mpc_problem.py:177-245); otherwise it must equal the
 *                    block size (rows*cols) so a problem's steps are packed.
 * ptr == NULL means "None" where the reference allows it (C, D, goal, targets). */
typedef struct MpcqpOperand {
    const void *ptr;
    int64_t batch_stride;
    int64_t step_stride;
} MpcqpOperand;

typedef struct MpcqpProblem {
    MpcqpOperand A;       /* [N] nx x nx   transition_state_matrix              */
    MpcqpOperand B;       /* [N] nx x nu   transition_input_matrix              */
    MpcqpOperand C;       /* [N] mk x nx   ineq_state_matrix        (nullable)  */
    MpcqpOperand D;       /* [N] mk x nu   ineq_input_matrix        (nullable)  */
    MpcqpOperand e;       /* [N] mk        ineq_vector                          */
    MpcqpOperand x0;      /* nx            initial_state   (step_stride unused) */
    MpcqpOperand goal;    /* nx            goal_state      (nullable)           */
    MpcqpOperand targets; /* N*nx          target_states   (nullable)           */
} MpcqpProblem;

/* MpcqpSolveOpts.flags -- explicit kernel-dispatch overrides. 0 in production: the library picks the
 * kernel from the dimensions. Tests set them to cross-check two formulations of the same solver on the
 * same batch (they replace the MPCQP_FORCE_* environment switches of ABI 4: no hidden process state). */
#define MPCQP_OPT_FORCE_LDS 1      /* everything that fits one CU's LDS goes to the workgroup kernel      */
#define MPCQP_OPT_FORCE_GWS 2      /* large QPs: general kernel with its arrays in the workspace          */
#define MPCQP_OPT_FORCE_DENSE_G 4  /* large fused path: form G (and its transpose) instead of applying it */
#define MPCQP_OPT_ONE_PER_WAVE 8   /* small problems: one problem per wavefront instead of two            */
#define MPCQP_OPT_FORCE_CONDENSED 16 /* keep the condensed kernels (mpcqp_build_solve_batch otherwise hands 16 < n <= 128,
                                        nx <= 4, nu <= 2, float64 -- and every other problem of more than 24 variables with
                                        nx <= 16 (float32: 12), nu <= 4 -- to the stage-wise kernels, which are faster
                                        there and return the same minimiser)                                      */

#define MPCQP_OPT_STAGE_WIDE 32   /* mpcqp_stagewise_solve_batch: take the wide kernel (nx <= 16, nu <= 4, MFMA
                                     sweeps) also where the narrow one (nx <= 4, nu <= 2, float64) applies       */

#define MPCQP_OPT_KEEP_FACTOR 64  /* stage-wise kernels: leave the Riccati factor of every problem in the workspace ...  */
#define MPCQP_OPT_REUSE_FACTOR 128 /* ... and start from it: the caller asserts that A, B and the weights are those of the
                                      launch that kept it, in the same workspace (build once, then only x0 / goal /
                                      targets / e change: the reference's update_cost_vector / update_constraint_vector
                                      usage, mpc_qp.py:129-163). Ignored by the other kernels (they rebuild).          */

#define MPCQP_OPT_PIPELINE_FACTOR 256 /* stage-wise kernel (float64, nx <= 4, nu <= 2, horizons whose factor fits LDS: N <= 64
                                      for nx = 4, nu = 1): TWO wavefronts per problem. One solves with the factor image
                                      `factor_slot` that a previous launch left in the workspace (as REUSE_FACTOR); the other,
                                      concurrently on another SIMD, factors the problem's operands as they are NOW into the
                                      other image, for the NEXT launch (which passes factor_slot ^ 1). The factor is still
                                      rebuilt once per launch -- what the reference's solve_mpc does every period,
                                      solve_mpc.py:42 -- but off the critical path. Contract: the operands A, B and the weights
                                      this launch sees are the ones the next launch will solve with (time-invariant or
                                      pre-scheduled dynamics); the first launch of a sequence uses KEEP_FACTOR.
                                      MPCQP_EUNSUPPORTED for other dimensions. */

#define MPCQP_OPT_SEED_VIOLATED 512 /* small-problem fused kernel (n <= 16, m <= 32): before the Goldfarb-Idnani iterations, the
                                 rows violated at the unconstrained minimiser enter by SEED STEPS (same rank-one updates, no
                                 selection, no ratio test; rows whose multiplier comes out negative leave again). Same
                                 minimiser; measured 5 % SLOWER than the plain iterations on BASELINE config 2,
                                 hence opt-in: the seed steps are what MPCQP_WARM_ACTIVE_SET starts from. */

#define MPCQP_OPT_EXACT_SELECTION 1024 /* accepted, no effect (ABI 11). Until ABI 10 it switched the wide stage-wise kernel from "lazy
                                 slacks" to a pass over the m slacks per step. Since round 6 that kernel never updates slacks
                                 incrementally: between two evaluations of the point it takes the most violated row among those
                                 whose whitened vectors the latest backward sweep cached (a valid Goldfarb-Idnani choice:
                                 same minimiser), see csrc/mpcqp_stagew.hip. */

#define MPCQP_OPT_TWO_PER_WAVE 2048 /* small-problem fused kernel: keep TWO problems per wavefront (mpcqp_pair.hip) where the
                                 dispatch would put FOUR on one (mpcqp_quad.hip: cold launches of problems with nx <= 16 and at
                                 most four rows per step -- BASELINE configs 1, 2, 4, the reference's WIP example -- from a few
                                 thousand problems up, see MPCQP_OPT_FOUR_PER_WAVE). Same method, same pivots: a cross-check
                                 (nx = 2 and nx >= 5 have no two-per-wavefront instantiation: the flag sends them to the one-per-wavefront kernel). */
#define MPCQP_OPT_STAGE_GENERAL 8192 /* mpcqp_stagewise_solve_batch: take the general stage-wise kernel (float64, nx <= 32, nu <= 8) also
                                 where the narrow or the wide one applies (a cross-check; until ABI 10 the formulation the host
                                 side re-solved through, when only this kernel kept a thin QR factor of the active rows -- the
                                 wide kernel does since ABI 11). MPCQP_EUNSUPPORTED for float32 and wider systems. */
#define MPCQP_OPT_FOUR_PER_WAVE 4096 /* ... and FOUR per wavefront for every batch size that kernel is eligible for (the dispatch
                                 takes it from more than two problems per SIMD of the device up: 2049 and more on an MI355X, where a
                                 wavefront per SIMD with four problems beats two wavefronts with two, and launches of several
                                 rounds keep two such wavefronts on every SIMD; smaller batches leave SIMDs idle either way;
                                 problems of 33 .. 64 rows with nx <= 8 run on its four-rows-per-lane copy, mpcqp_quad4.hip, at
                                 every batch size).
                                 MPCQP_EUNSUPPORTED where the kernel does not apply (nx > 16, more than four rows per step -- eight with nx <= 8 --, warm
                                 starts, seed steps, a pairing order together with input rows / a stage cost). The shared-model solves (mpcqp_solve_model_batch / _bounds_batch) take
                                 it too, with the same batch-size rule, for every model with n <= 16, m <= 32. */

/* MpcqpSolveOpts.warm_start */
#define MPCQP_WARM_OPERATOR 1   /* begin from the stored active set AND operator N* (contract: matrices unchanged)          */
#define MPCQP_WARM_ACTIVE_SET 2 /* begin from the stored active set's ROW IDS only, moved by warm_shift rows: the rows enter
                                   by seed steps on THIS problem's matrices, so the matrices, bounds and states may all have
                                   changed -- the receding-horizon case, where row (k, i) of last period is row (k - 1, i)
                                   now: warm_shift = mk. Small-problem fused kernel only (MPCQP_EUNSUPPORTED elsewhere).   */

typedef struct MpcqpSolveOpts {
    int32_t max_iter; /* active-set iterations per problem; <=0 -> 10*(n+m)     */
    int32_t flags;    /* MPCQP_OPT_* (0 = automatic dispatch)                    */
    double feas_tol;  /* a row is violated when (h_i-G_i u)/(1+|h_i|) < -tol;
                         <=0 -> 1e-12 (f64) / 1e-5 (f32)                         */
    /* Warm start for receding-horizon loops (the reference reaches its backends' warm start through
     * **kwargs -> qpsolvers' initvals, qpmpc/solve_mpc.py:20,43; loops at
     * examples/wheeled_inverted_pendulum.py:99-118, examples/lipm_walking_controller.py:307-335).
     * warm_state: caller-owned DEVICE buffer of mpcqp_warm_state_bytes() per problem, packed by problem.
     *   When non-NULL every solve ENDS by storing its final active set and the active-set operator
     *   N* = (M_A M_A')^-1 M_A there (M = G L^-T); problems that were not solved store an empty set.
     * warm_start != 0: the solve BEGINS from the stored state instead of the empty set: multipliers are
     *   recomputed for the new q and h, rows whose multiplier turned negative leave, violated rows enter
     *   through the usual iterations -- a period whose active set moved by two rows costs about two
     *   iterations instead of one per active row.
     * Contract: N* depends on the matrices only, so the state is meaningful while A, B, C, D and the
     *   weights are those of the solve that stored it; e, x0, goal and targets may change freely. The
     *   contract is not trusted: with a warm start a solution is accepted only after its KKT conditions
     *   were re-checked from scratch (stationarity holds by construction, multipliers >= 0, active rows
     *   on their bounds, inactive rows feasible); a stale state costs a cold restart, never a wrong plan.
     * Supported by the small-problem fused kernel (n <= 16, m <= 32, float64) as described, and by the stage-wise kernel
     * that mpcqp_build_solve_batch picks for small systems with 16 < n <= 128 (float64, nx <= 4, nu <= 2): there the record
     * holds the active rows' ids only -- their vectors V_a = P^-1 g_a', the trajectories and W = (G_A P^-1 G_A')^-1 stay
     * in the WORKSPACE, so a warm start needs the same workspace as the launch before and MPCQP_OPT_REUSE_FACTOR (same
     * contract: matrices and weights unchanged); the multipliers are lam = -W s_A at the new unconstrained minimiser,
     * rows with a negative one leave, and a state whose active rows do not end up on their bounds (another problem's, or
     * another workspace's -- the record carries the workspace's address) costs a cold restart. Other dimensions return
     * MPCQP_EUNSUPPORTED when warm_state is given. */
    void *warm_state;
    int32_t warm_start;
    int32_t factor_slot; /* 0 | 1: the factor image of the stage-wise workspace that KEEP_FACTOR writes, REUSE_FACTOR and
                            PIPELINE_FACTOR read (PIPELINE_FACTOR writes the other one). 0 unless a loop alternates them. */
    /* Developer probe, NULL in production: DEVICE buffer of int64 per problem (16 for the small-problem
     * kernels, 32 for the mid-size / large ones) that receives shader-clock stamps at phase boundaries. */
    void *probe;
    /* Size in bytes of the buffer behind warm_state (ABI 6): a launch over `batch` problems needs
     * batch * mpcqp_warm_state_bytes(); a smaller buffer (a state allocated for another batch or other
     * dimensions) is refused with MPCQP_EWORKSPACE before anything is launched. Ignored when warm_state is NULL. */
    size_t warm_state_bytes;
    /* MPCQP_WARM_ACTIVE_SET (ABI 8): stored row id r is taken as row r - warm_shift of this launch's problem; ids that
     * fall off the horizon's start (or name a padded row) are dropped. 0: the rows kept their places. */
    int32_t warm_shift;
    int32_t reserved_;
    /* Pairing order (ABI 9), NULL = natural: DEVICE array of `batch` int32, a permutation of 0 .. batch-1. The small-problem
     * fused kernel (n <= 16, m <= 32) puts TWO problems on a wavefront, which then runs max(trips_a, trips_b) active-set trips --
     * 13.1 against 10.75 per problem on BASELINE config 4 (examples/humanoid_one_step.py:42-80 times 65,536). With an order,
     * half-wavefront i takes problem order[i]; outputs stay where they were (U, lam, status, iters are indexed by problem), every
     * problem's iterations are the same (the two halves of a wavefront sum in different orders: plans equal to rounding). A receding-horizon loop passes mpcqp_order_by_count() of last period's iteration counts
     * (qpmpc/solve_mpc.py:43 is called once per period, examples/lipm_walking_controller.py:307-335): launches of several rounds
     * get up to 12 % shorter, a launch that fills the machine once gains nothing. Cold launches of mpcqp_build_solve_batch and launches
     * of mpcqp_solve_model_batch / mpcqp_solve_model_bounds_batch that the small-problem kernel serves only: MPCQP_EUNSUPPORTED
     * with warm_state, with a dispatch override, for other dimensions and from every other entry point. The array is read during the launch (keep it alive until the stream has
     * passed it). It is not checked for being a permutation (a problem named twice is solved twice, one left out keeps its old
     * outputs); an index outside the batch is clamped into it. */
    const int32_t *order;
} MpcqpSolveOpts;

/* ABI version of the loaded library (== MPCQP_ABI_VERSION of its build). */
int mpcqp_abi_version(void);

/* Static text for a return code (<0: MPCQP_E*, >0: hipGetErrorString). */
const char *mpcqp_error_string(int code);

/* Dynamic LDS bytes one problem of these dimensions needs on the fused path;
 * MPCQP_ETOOLARGE when it exceeds the 160 KiB of a gfx950 CU (such problems take the
 * HBM-resident path and need a workspace). */
int mpcqp_lds_bytes(const MpcqpDims *dims, size_t *bytes);

/* Device scratch bytes that mpcqp_condense_batch (for_solve == 0) or
 * mpcqp_build_solve_batch (for_solve != 0) need for `batch` problems of these
 * dimensions: 0 for small problems solved entirely on chip; 2 n^2 elements per problem
 * (rows of the active-set operator) for the mid-size fused kernel; the propagators, P
 * and the same rows for large problems. The query sees dimensions only, so it reports the
 * largest amount any kernel the launch may pick (it depends on the operand strides) needs. */
int mpcqp_workspace_bytes(const MpcqpDims *dims, int64_t batch, int32_t for_solve, size_t *bytes);

/* Same for mpcqp_solve_batch (n variables, m rows). */
int mpcqp_solve_workspace_bytes(int32_t n, int32_t m, int32_t dtype, int64_t batch, size_t *bytes);

/* Bytes per problem of MpcqpSolveOpts.warm_state for these dimensions (0: no warm start for them). */
int mpcqp_warm_state_bytes(const MpcqpDims *dims, size_t *bytes);

/* Whose record that buffer holds (ABI 10) -- the kernel mpcqp_build_solve_batch picks for these dimensions and this dtype (a
 * float32 launch of at most 160 variables is solved by the float64 kernels and keeps THEIR record):
 *   MPCQP_WARM_KIND_OPERATOR  small-problem fused kernel: the operator N* (16 x 16 float64, by slot), then 16 int32 constraint
 *                             ids (-1 = empty slot). MPCQP_WARM_OPERATOR and MPCQP_WARM_ACTIVE_SET.
 *   MPCQP_WARM_KIND_STAGE     narrow stage-wise kernel: int32 [count, slots, workspace tag (2), step of each row ..., index in its
 *                             step ...]; the rows' vectors stay in the launch's workspace (MPCQP_OPT_REUSE_FACTOR contract).
 *   MPCQP_WARM_KIND_ROWS      wide stage-wise kernel: int32 count, then the active rows' ids. MPCQP_WARM_ACTIVE_SET only.
 * A binding reads the layout from here instead of re-deriving the dispatch. */
#define MPCQP_WARM_KIND_NONE 0
#define MPCQP_WARM_KIND_OPERATOR 1
#define MPCQP_WARM_KIND_STAGE 2
#define MPCQP_WARM_KIND_ROWS 3
int mpcqp_warm_state_kind(const MpcqpDims *dims, int32_t *kind);

/* Replaces MPCQP.__init__ (mpc_qp.py:39-122) for a batch: Phi/Psi propagation
 * (:53-54,:88-90), G_k/h_k (:62-78), P (:99-105), q (:129-149).
 * Outputs per problem, packed: P[n*n], q[n], G[m*n], h[m];
 * Phi[(N+1)*nx*nx] (blocks Phi_0..Phi_N; the last one is phi_last) and
 * Psi[(N+1)*nx*n]  (blocks Psi_0..Psi_N; the last one is psi_last) may be NULL. */
int mpcqp_condense_batch(const MpcqpDims *dims, const MpcqpProblem *problem,
                         int64_t batch, void *P, void *q, void *G, void *h,
                         void *Phi, void *Psi, void *workspace,
                         size_t workspace_bytes, void *stream);

/* The two launches of mpcqp_condense_batch for problems that do not fit a CU's LDS (BASELINE config 5's size), one at a
 * time -- for measurement (bench.py times the Gram product alone: `roofline.gram_mfma`) and for callers that overlap them
 * with other work. phase 1 replaces the propagation loop of MPCQP.__init__ (qpmpc/mpc_qp.py:53-98): Psi (caller's buffer,
 * same packing as mpcqp_condense_batch), G, h and the tracking residuals (workspace: (N + 1) nx elements per problem).
 * phase 2 replaces the Hessian and cost-vector products (qpmpc/mpc_qp.py:99-105, 129-149): P = w_u I + Psi' W Psi and
 * q = Psi' W resid from what phase 1 left in Psi and the workspace -- float32 with n a multiple of 32: on the matrix cores
 * (v_mfma_f32_32x32x2_f32, the causal lower triangle only). MPCQP_EUNSUPPORTED for problems the on-chip kernel condenses. */
int mpcqp_condense_phase_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch, int32_t phase,
                               void *P, void *q, void *G, void *h, void *Psi, void *workspace,
                               size_t workspace_bytes, void *stream);

/* Replaces MPCQP.update_cost_vector (mpc_qp.py:129-149) and
 * MPCQP.update_constraint_vector (mpc_qp.py:151-163) for a batch, from the
 * Phi/Psi kept by mpcqp_condense_batch (same packing; batch strides
 * phi_batch_stride/psi_batch_stride in elements, 0 = one shared copy) and new
 * x0/goal/targets. Reads problem->C, ->e, ->x0, ->goal, ->targets only.
 * q and/or h may be NULL to skip that update. */
int mpcqp_update_vectors_batch(const MpcqpDims *dims, const MpcqpProblem *problem,
                               const void *Phi, int64_t phi_batch_stride,
                               const void *Psi, int64_t psi_batch_stride,
                               int64_t batch, void *q, void *h, void *stream);

/* Replaces qpsolvers.solve_problem(Problem(P,q,G,h), solver=...) at its call
 * site qpmpc/solve_mpc.py:43 for a batch of dense strictly convex QPs:
 * x[batch*n], lam[batch*m] (multipliers of G u <= h, nullable),
 * status[batch], iters[batch] (nullable). Dual active-set method. */
int mpcqp_solve_batch(int32_t n, int32_t m, int32_t dtype, const void *P,
                      const void *q, const void *G, const void *h, int64_t batch,
                      const MpcqpSolveOpts *opts, void *x, void *lam,
                      int32_t *status, int32_t *iters, void *workspace,
                      size_t workspace_bytes, void *stream);

/* Replaces the whole of solve_mpc (qpmpc/solve_mpc.py:42-44) for a batch, fused:
 * P, G and their factors never leave the CU. U[batch*n] is the stacked input
 * sequence (Plan.inputs = U.reshape(N, nu), plan.py:36-39).
 * status[b] is what an exact backend of the reference would report (found / not found, plan.py:35-40): problems of 17 .. 128
 * variables with nx <= 4, nu <= 2 -- the narrow stage-wise kernel, whose active-set operator is an explicit inverse -- that this
 * kernel leaves MPCQP_MAX_ITER or MPCQP_INFEASIBLE are solved once more by the wide stage-wise kernel (thin QR factor of the
 * active rows) in a second launch on the same stream and in the same workspace, before the call returns (ABI 11; not when the
 * launch keeps state for a later one: MPCQP_OPT_KEEP_FACTOR / _REUSE_FACTOR / _PIPELINE_FACTOR, a warm state). */
int mpcqp_build_solve_batch(const MpcqpDims *dims, const MpcqpProblem *problem,
                            int64_t batch, const MpcqpSolveOpts *opts, void *U,
                            void *lam, int32_t *status, int32_t *iters,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage-wise (uncondensed) formulation: long horizons -------------------------------------
 * The reference has no sparse formulation: sparse=True only wraps the dense condensed matrices in CSC
 * (qpmpc/mpc_qp.py:39,108-109; qpmpc/solve_mpc.py:31-32), so its build is O(N^2) memory, O(N^3) time.
 * This entry point solves the same QP as mpcqp_build_solve_batch -- same operands, same outputs, same
 * minimiser -- without forming P or G: Riccati gains once per problem, then per active-set iteration
 * one LQR solve (two sweeps over the horizon) and O(|A| N) vector work; O(N) memory. No cap on
 * N; float64 with 2 <= nx <= 4, 1 <= nu <= 2 (chunked scans over the horizon), or float64 / float32 with nx <= 16,
 * nu <= 4 (matrix-core sweeps; MPCQP_EUNSUPPORTED otherwise). `max_active` bounds the number of simultaneously active
 * rows (<= 0: min(n, m, 128)); a problem that needs more returns status MPCQP_SLOTS_FULL. The workspace is caller-owned
 * (mpcqp_stagewise_workspace_bytes). mpcqp_build_solve_batch reaches the same kernels by itself for every problem that
 * does not fit the on-chip condensed kernels (any n = N nu). MPCQP_OPT_STAGE_GENERAL asks for the general kernel (float64,
 * nx <= 32, nu <= 8: thin-QR active-set operator, like the wide kernel's since ABI 11) whatever
 * the width; its workspace is sized by the query with max_active = -1 (default slots) or -k (k slots). As in
 * mpcqp_build_solve_batch, what the narrow kernel leaves MPCQP_MAX_ITER / MPCQP_INFEASIBLE is solved once more by the wide kernel
 * (same slots, same workspace, same stream) before the call returns. */
int mpcqp_stagewise_workspace_bytes(const MpcqpDims *dims, int64_t batch, int32_t max_active, size_t *bytes);
int mpcqp_stagewise_solve_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch,
                                const MpcqpSolveOpts *opts, int32_t max_active, void *U, void *lam,
                                int32_t *status, int32_t *iters, void *workspace, size_t workspace_bytes,
                                void *stream);

/* ---- shared-model path: build once, re-solve for new states ----------------------
 * The reference's own fast path (doc/src/developer-notes.rst:10, CHANGELOG.md:12,44):
 * build MPCQP once, then only update_cost_vector / update_constraint_vector
 * (mpc_qp.py:129-163) and re-solve. When A, B, C, D, e and the weights are the same for
 * every problem of a batch (an initial-state sweep, or successive steps of
 * time-invariant receding-horizon loops), everything that does not depend on
 * x0/goal/targets is computed ONCE into a "model": L = chol(P), M = G L^-T, L^-T and
 * the linear maps  h = e - Hx x0,  L^-1 q = Wx x0 - Wg goal - Wt targets.
 *
 * The model is filled from condensed data of 1 + 2 nx + N nx PSEUDO-problems sharing the
 * operands (mpcqp_condense_batch on them): pseudo-problem 0 has x0 = goal = targets = 0,
 * then x0 = e_c (c < nx), then goal = e_c, then targets = e_j (j < N nx):
 *   P, G        [n*n], [m*n]        of pseudo-problem 0
 *   q_basis     [(1+2nx+N nx) * n]  their q vectors, in that order
 *   h_basis     [(1+2nx+N nx) * m]  their h vectors                                      */
int mpcqp_model_bytes(const MpcqpDims *dims, size_t *bytes);
int mpcqp_factor_model(const MpcqpDims *dims, const void *P, const void *G,
                       const void *q_basis, const void *h_basis, void *model,
                       size_t model_bytes, void *stream);

/* Solve `batch` problems that share `model`: only x0 [nx], goal [nx] and targets [N nx]
 * differ (batch_stride 0 = shared; goal/targets may be NULL when their cost term is
 * not flagged). Replaces update_cost_vector + update_constraint_vector + the solver
 * call for every problem. Outputs as mpcqp_build_solve_batch. */
int mpcqp_solve_model_batch(const MpcqpDims *dims, const void *model,
                            const MpcqpOperand *x0, const MpcqpOperand *goal,
                            const MpcqpOperand *targets, int64_t batch,
                            const MpcqpSolveOpts *opts, void *U, void *lam,
                            int32_t *status, int32_t *iters, void *stream);

/* As mpcqp_solve_model_batch, with the inequality vectors e given PER PROBLEM ([batch, N|1, mk] through `e`'s strides)
 * instead of taken from the model: the matrices are shared and constant, the bounds move -- the per-step ZMP bounds
 * that examples/lipm_walking_controller.py:179-213 (update_goal_and_constraints) rewrites every control period while
 * A, B, C stay, i.e. update_constraint_vector (mpc_qp.py:151-163) with a new e. h = e - (C Phi) x0 uses the model's map.
 * e == NULL or e->ptr == NULL is mpcqp_solve_model_batch. Served by the pair kernel (n <= 16, m <= 32, float64);
 * MPCQP_EUNSUPPORTED elsewhere. */
int mpcqp_solve_model_bounds_batch(const MpcqpDims *dims, const void *model, const MpcqpOperand *e,
                                   const MpcqpOperand *x0, const MpcqpOperand *goal,
                                   const MpcqpOperand *targets, int64_t batch,
                                   const MpcqpSolveOpts *opts, void *U, void *lam,
                                   int32_t *status, int32_t *iters, void *stream);

/* Replaces MPCProblem.integrate (mpc_problem.py:316-335) as used by Plan.states
 * (plan.py:81-109) for a batch: X[batch*(N+1)*nx], X_0 = x0,
 * X_{k+1} = A_k X_k + B_k U_k. U is packed [batch*N*nu]. */
int mpcqp_rollout_batch(const MpcqpDims *dims, const MpcqpOperand *A,
                        const MpcqpOperand *B, const MpcqpOperand *x0,
                        const void *U, int64_t batch, void *X, void *stream);

/* One period of `batch` wheeled-inverted-pendulum control loops, fused: apply the first
 * input of each plan (U[b*u_stride]) to the nonlinear plant for `nsub` Taylor sub-steps of
 * sampling_period/nsub (WheeledInvertedPendulum.integrate,
 * qpmpc/systems/wheeled_inverted_pendulum.py:127-160, called from
 * examples/wheeled_inverted_pendulum.py:110-111), then write the next MPC problem's
 * x0 [4], goal [4] and targets [N*4] reference ramp (get_target_states, same example
 * :65-83,:101-108). states [batch*4] is updated in place; a loop whose plan was not
 * found (status[b] != 0, may be NULL) applies a zero input. nsub = 0 leaves the plant where
 * it is and writes the problem of the CURRENT state (the first problem of an episode). */
int mpcqp_wip_advance_batch(int32_t dtype, void *states, const void *U, int64_t u_stride,
                            const int32_t *status, int32_t N, double sampling_period,
                            double target_vel, double length, double gravity, int32_t nsub,
                            void *x0, void *goal, void *targets, int64_t batch, void *stream);

/* The same with the loops' bookkeeping fused in (one launch per period instead of two): if stats is not NULL,
 * stats[0] += number of loops with status != 0 and stats[1] += sum of iters (iters may be NULL), as
 * mpcqp_accumulate_stats does. */
int mpcqp_wip_advance_stats_batch(int32_t dtype, void *states, const void *U, int64_t u_stride,
                                  const int32_t *status, const int32_t *iters, int64_t *stats, int32_t N,
                                  double sampling_period, double target_vel, double length, double gravity,
                                  int32_t nsub, void *x0, void *goal, void *targets, int64_t batch, void *stream);

/* A WHOLE control period of `batch` wheeled-inverted-pendulum loops in ONE launch: mpcqp_build_solve_batch of every
 * loop's problem, then -- as the epilogue of the solver kernel, by the wavefront that solved it -- the plant step with
 * the plan's first input (zero when status != 0) and the loop's next problem written IN PLACE over x0 / goal / targets
 * of `problem`, exactly as mpcqp_wip_advance_batch does (examples/wheeled_inverted_pendulum.py:99-118, one iteration of
 * the loop). states [batch, 4] is updated in place; loop_stats (int64 [batch, 2], device, may be NULL): [b][0] += 1 if
 * the plan was not found, [b][1] += iterations. Only for problems mpcqp_build_solve_batch hands to the stage-wise
 * kernel (float64, nx = 4, nu = 1, 16 < N <= 128, per-loop x0 / goal / targets): MPCQP_EUNSUPPORTED otherwise -- the
 * caller then launches the solve and mpcqp_wip_advance_stats_batch separately. Workspace: mpcqp_workspace_bytes. */
int mpcqp_wip_period_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch,
                           const MpcqpSolveOpts *opts, void *U, void *lam, int32_t *status, int32_t *iters,
                           void *workspace, size_t workspace_bytes, void *states, int64_t *loop_stats,
                           double sampling_period, double target_vel, double length, double gravity, int32_t nsub,
                           void *stream);

/* `nperiods` consecutive control periods of every loop in ONE launch (ABI 7): what nperiods calls of
 * mpcqp_wip_period_batch do -- the epilogue writes the loop's next problem in place, so the wavefront that solved period t
 * carries on with period t + 1 -- without the launch boundary between them (a period of 1024 loops is ~30 us of work
 * behind ~8 us of dispatch; the loops never interact, so nothing has to meet at a period's end). Trajectories, U / lam /
 * status / iters (those of the LAST period) and loop_stats are bitwise what the one-period calls leave. With
 * MPCQP_OPT_PIPELINE_FACTOR the two factor images alternate from period to period starting at opts->factor_slot: the
 * caller's next launch passes factor_slot ^ (nperiods & 1). MPCQP_OPT_KEEP_FACTOR, warm_state and horizons whose factor
 * does not fit LDS (N > 64 for nx = 4, nu = 1) need nperiods == 1 (MPCQP_EUNSUPPORTED otherwise); nperiods < 1 is
 * MPCQP_EINVAL. examples/wheeled_inverted_pendulum.py:99-118, nperiods
 * iterations of the loop. */
int mpcqp_wip_periods_batch(const MpcqpDims *dims, const MpcqpProblem *problem, int64_t batch,
                            const MpcqpSolveOpts *opts, void *U, void *lam, int32_t *status, int32_t *iters,
                            void *workspace, size_t workspace_bytes, void *states, int64_t *loop_stats,
                            double sampling_period, double target_vel, double length, double gravity, int32_t nsub,
                            int32_t nperiods, void *stream);

/* Bookkeeping of closed loops (the reference's loops count nothing; ours report failures and iterations):
 * stats[0] += number of problems with status != 0, stats[1] += sum of iters. stats: two int64 in DEVICE memory. */
int mpcqp_accumulate_stats(const int32_t *status, const int32_t *iters, int64_t batch, int64_t *stats, void *stream);

/* A pairing order for MpcqpSolveOpts.order (ABI 9): order = the problems sorted by counts[] (last period's `iters`, clamped to
 * 0 .. 1023), longest first, by a counting sort on the device (three small launches, ~10 us for 65,536). Equal counts come out in
 * an unspecified order. counts, order: DEVICE int32 [batch]; workspace: mpcqp_order_workspace_bytes(batch) bytes of DEVICE memory
 * (MPCQP_EWORKSPACE if smaller). The reference has no counterpart: its solver is called per problem (qpmpc/solve_mpc.py:43). */
size_t mpcqp_order_workspace_bytes(int64_t batch);

int mpcqp_order_by_count(const int32_t *counts, int64_t batch, int32_t *order, void *workspace, size_t workspace_bytes,
                         void *stream);

/* One period of `batch` LIPM walking controllers, fused (examples/lipm_walking_controller.py:304-333):
 * if U is not NULL, apply the first jerk of each plan (U[b*u_stride], zero when status[b] != 0) to
 * the constant-jerk plant for `nsub` exact sub-steps (integrate, :216-236) and advance the footstep
 * phase (PhaseStepper.advance / advance_stride :125-132, main loop :329-332); then write the NEXT MPC
 * problem of every walker: x0 [3], goal [3] and the per-step ZMP bounds e [N, 2] of its receding
 * horizon (PhaseStepper.get_nb_steps :134-165, update_goal_and_constraints :179-213; rows without a
 * bound hold max_zmp_dist). U == NULL only writes the problem of the current phase (first period).
 * Per walker, updated in place: states [3], index, stride_index (int64), support; read: strides [2],
 * foot_size. Refuses (EINVAL) horizons spanning more than two steps, like the reference (:107-110). */
int mpcqp_lipm_advance_batch(int32_t dtype, void *states, const void *U, int64_t u_stride,
                             const int32_t *status, int32_t N, double sampling_period, int32_t nsub,
                             int32_t nb_dsp, int32_t nb_ssp, double max_zmp_dist, int64_t *index,
                             int64_t *stride_index, void *support, const void *strides,
                             const void *foot_size, void *x0, void *goal, void *e, int64_t batch,
                             void *stream);

/* The same with the loops' bookkeeping fused in: if stats is not NULL, stats[0] += number of walkers with
 * status != 0 and stats[1] += sum of iters (iters may be NULL), as mpcqp_accumulate_stats does. */
int mpcqp_lipm_advance_stats_batch(int32_t dtype, void *states, const void *U, int64_t u_stride,
                                   const int32_t *status, const int32_t *iters, int64_t *stats, int32_t N,
                                   double sampling_period, int32_t nsub, int32_t nb_dsp, int32_t nb_ssp,
                                   double max_zmp_dist, int64_t *index, int64_t *stride_index, void *support,
                                   const void *strides, const void *foot_size, void *x0, void *goal, void *e,
                                   int64_t batch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCQP_H_ */
