// mpcqp_big.hip -- gfx950 kernels for problems too large for one CU's LDS
// (BASELINE config 5: nx=12, nu=4, N=64 -> n=256 variables, m=1024 rows, f32).
// Psi, G and P live in HBM (per-problem slices of a caller-owned workspace);
// the condensing is split into
//   mpcqp_propagate_kernel : Phi/Psi propagation, G_k, h_k, tracking residuals
//                            (qpmpc/mpc_qp.py:53-98)        -- HBM-bound streaming
//   mpcqp_gram_mfma_f32    : P = w_u I + Psi' W Psi, q = Psi' W resid
//                            (qpmpc/mpc_qp.py:99-105,129-149) -- MFMA-bound SYRK
// The Gram contraction is the only GEMM-shaped step of the path (1.0e8 of the
// 1.15e8 flops of a config-5 build) and the only one that goes to
// the matrix cores: v_mfma_f32_32x32x2_f32, exact f32 (bitwise an fmaf chain).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mpcqp.h"
#include "mpcqp_internal.h"

namespace mpcqp {

namespace big {
constexpr int NXMAX = 16;  // state dimension held in registers by the propagation
constexpr int KC = 32;     // rows of Psi staged per Gram step
}  // namespace big
using namespace big;

// ------------------------------------------------------------------ propagate
// One workgroup per problem, thread c < n <= 256 owns column c of Psi; the last wavefront carries the
// free response Phi_k x0 (in LDS), h and the tracking residuals. A_k, B_k, C_k, D_k of the current step are staged in LDS and read
// as broadcast.
// Outputs (per problem): Psi_all [(N+1)*nx, n] (block 0 is zero), resid [(N+1)*nx] =
// Phi_k x0 - ref_k, h [m], and optionally G [m, n] and the inverse row norms 1/|G_i| [m].
// The norms come from the nx x nx recursion S_{k+1} = A_k S_k A_k' + B_k B_k' (S_k = Psi_k Psi_k',
// |G_i|^2 = C_i S_k C_i' + |D_i|^2 because Psi_k is zero in the columns of u_k).
// NXC > 0: the state dimension at compile time (straight-line 16-byte broadcast reads of A_k, C_k); 0: any nx <= NXMAX.
template <typename T, bool NRM, int NXC>
__global__ void __launch_bounds__(320, (sizeof(T) == 8 || NXC == 0) ? 2 : 5) mpcqp_propagate_kernel(const KernelArgs ka, T *__restrict__ Psi_ws,
                                                              T *__restrict__ res_ws, T *__restrict__ oG,
                                                              T *__restrict__ oh, T *__restrict__ onrm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sm = (T *)smem_raw;
    const int nx = NXC > 0 ? NXC : ka.nx, nu = ka.nu, N = ka.N, mk = ka.mk, n = ka.n, m = ka.m;
    constexpr int NXU = NXC > 0 ? NXC : NXMAX;  // unroll bound of the per-column loops
    const int tid = threadIdx.x;
    const int64_t prob = blockIdx.x;
    const int nA = nx * nx, nB = nx * nu, nC = mk * nx, nD = mk * nu;
    auto al4 = [](int c) { return (c + 3) & ~3; };  // every staged array starts 16-byte aligned
    T *As = sm, *Bs = As + al4(nA), *Cs = Bs + al4(nB), *Ds = Cs + al4(nC), *es = Ds + al4(nD);
    T *Ss = es + al4(mk), *T1s = Ss + al4(nA), *Ys = T1s + al4(nA);
    const T *gA = (const T *)ka.A.ptr + prob * ka.A.batch_stride;
    const T *gB = (const T *)ka.B.ptr + prob * ka.B.batch_stride;
    const T *gC = ka.C.ptr ? (const T *)ka.C.ptr + prob * ka.C.batch_stride : nullptr;
    const T *gD = ka.D.ptr ? (const T *)ka.D.ptr + prob * ka.D.batch_stride : nullptr;
    const T *ge = (const T *)ka.e.ptr + prob * ka.e.batch_stride;
    const T *gx0 = (const T *)ka.x0.ptr + prob * ka.x0.batch_stride;
    const T *ggoal = ka.goal.ptr ? (const T *)ka.goal.ptr + prob * ka.goal.batch_stride : nullptr;
    const T *gtgt = ka.targets.ptr ? (const T *)ka.targets.ptr + prob * ka.targets.batch_stride : nullptr;
    T *Psi = Psi_ws + prob * (int64_t)(N + 1) * nx * n;
    T *res = res_ws + prob * (int64_t)(N + 1) * nx;
    T *G = oG ? oG + prob * (int64_t)m * n : nullptr;
    T *h = oh + prob * (int64_t)m;
    T *nrm = NRM ? onrm + prob * (int64_t)m : nullptr;
    const bool col = (tid < n);
    const int j = col ? tid / nu : -1, ii = col ? tid - j * nu : 0;
    const bool qs = (ka.flags & MPCQP_Q_STAGE) && gtgt, qt = (ka.flags & MPCQP_Q_TERMINAL) && ggoal;
    // first step at which some column of this wavefront is non-zero (uniform): before it the
    // wavefront only stores zeros (causality, mpc_qp.py:80-90)
    const int jfirst = __builtin_amdgcn_readfirstlane(min((tid & ~63) / nu, N));
    // the free response Phi_k x0 lives in LDS (double-buffered) and is advanced by the last wavefront,
    // one lane per row, so that no single lane carries a serial chain of mk + nx dot products
    T *xs = Ys + al4(nC);
    const int l4 = tid - 256;  // lane of the last wavefront (threads 256..319)

    if constexpr (NRM)
        for (int e2 = tid; e2 < nA; e2 += 320) Ss[e2] = T(0);
    if (l4 >= 0 && l4 < nx) xs[l4] = gx0[l4];

    T v[NXU];
#pragma unroll
    for (int s = 0; s < NXU; ++s) v[s] = T(0);
    // The operands of step k+1 are requested into registers while step k computes and land in LDS at the next
    // barrier: the HBM latency of the staging is paid once, not once per step. (Two elements per thread and array
    // cover 640 entries; larger blocks -- mk nx > 640 -- are staged the plain way.)
    const bool pfok = nA <= 640 && nB <= 640 && nC <= 640 && nD <= 640 && mk <= 640;
    T pa[2], pb[2], pc[2], pd[2], pe[2];
    auto request = [&](int k) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 320;
            pa[u] = (i < nA) ? gA[k * ka.A.step_stride + i] : T(0);
            pb[u] = (i < nB) ? gB[k * ka.B.step_stride + i] : T(0);
            pc[u] = (gC && i < nC) ? gC[k * ka.C.step_stride + i] : T(0);
            pd[u] = (gD && i < nD) ? gD[k * ka.D.step_stride + i] : T(0);
            pe[u] = (i < mk) ? ge[k * ka.e.step_stride + i] : T(0);
        }
    };
    auto land = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 320;
            if (i < nA) As[i] = pa[u];
            if (i < nB) Bs[i] = pb[u];
            if (gC && i < nC) Cs[i] = pc[u];
            if (gD && i < nD) Ds[i] = pd[u];
            if (i < mk) es[i] = pe[u];
        }
    };
    if (pfok) request(0);
    for (int k = 0; k <= N; ++k) {
        const T *xc = xs + (k & 1) * nx;
        T *xn = xs + ((k + 1) & 1) * nx;
        __syncthreads();
        if (k < N) {  // stage the operands of step k (coalesced)
            if (pfok) {
                land();
            } else {
                for (int i = tid; i < nA; i += blockDim.x) As[i] = gA[k * ka.A.step_stride + i];
                for (int i = tid; i < nB; i += blockDim.x) Bs[i] = gB[k * ka.B.step_stride + i];
                if (gC)
                    for (int i = tid; i < nC; i += blockDim.x) Cs[i] = gC[k * ka.C.step_stride + i];
                if (gD)
                    for (int i = tid; i < nD; i += blockDim.x) Ds[i] = gD[k * ka.D.step_stride + i];
                for (int i = tid; i < mk; i += blockDim.x) es[i] = ge[k * ka.e.step_stride + i];
            }
        }
        __syncthreads();
        if (pfok && k + 1 < N) request(k + 1);
        // v = Psi_k[:, c]
        if (col) {
#pragma unroll
            for (int s = 0; s < NXU; ++s)
                if (s < nx) Psi[((int64_t)k * nx + s) * n + tid] = v[s];
        } else if (l4 >= 0 && l4 < nx) {
            T ref = T(0);
            if (k < N) {
                if (qs) ref = gtgt[k * nx + l4];
            } else if (qt) {
                ref = ggoal[l4];
            }
            res[k * nx + l4] = xc[l4] - ref;
        }
        if (k == N) break;
        if (col) {
            // rows of G for step k (mpc_qp.py:62-78)
            if (G) {
                for (int i2 = 0; i2 < mk; ++i2) {
                    T acc = T(0);
                    if (gC && k > jfirst) {
#pragma unroll
                        for (int s = 0; s < NXU; ++s)
                            if (s < nx) acc += Cs[i2 * nx + s] * v[s];
                    }
                    if (gD && j == k) acc += Ds[i2 * nu + ii];
                    G[((int64_t)k * mk + i2) * n + tid] = acc;
                }
            }
            if (k >= jfirst) {
                // advance (mpc_qp.py:88-90)
                T w[NXU];
#pragma unroll
                for (int r = 0; r < NXU; ++r) {
                    T acc = T(0);
                    if (r < nx) {
#pragma unroll
                        for (int s = 0; s < NXU; ++s)
                            if (s < nx) acc += As[r * nx + s] * v[s];
                    }
                    w[r] = acc;
                }
                if (j == k) {
#pragma unroll
                    for (int r = 0; r < NXU; ++r)
                        if (r < nx) w[r] = Bs[r * nu + ii];
                }
#pragma unroll
                for (int r = 0; r < NXU; ++r) v[r] = w[r];
            }
        } else if (l4 >= 0) {
            // h rows (mpc_qp.py:74-78) and the free response, one lane per row
            for (int i2 = l4; i2 < mk; i2 += 64) {
                T acc = T(0);
                if (gC)
                    for (int s = 0; s < nx; ++s) acc += Cs[i2 * nx + s] * xc[s];
                h[k * mk + i2] = es[i2] - acc;
            }
            if (l4 < nx) {
                T acc = T(0);
                for (int s = 0; s < nx; ++s) acc += As[l4 * nx + s] * xc[s];
                xn[l4] = acc;
            }
        }
        if constexpr (NRM) {
            // inverse row norms of step k: Y = C S_k and T1 = A S_k now, the row-wise dots and
            // S_{k+1} = T1 A' + B B' after a barrier -- a few dozen FMAs per thread, shared by all
            if (gC)
                for (int e2 = tid; e2 < nC; e2 += 320) {
                    const int i2 = e2 / nx, s = e2 - i2 * nx;
                    T acc = T(0);
                    for (int u = 0; u < nx; ++u) acc += Cs[i2 * nx + u] * Ss[u * nx + s];
                    Ys[e2] = acc;
                }
            for (int e2 = tid; e2 < nA; e2 += 320) {
                const int r = e2 / nx, s = e2 - r * nx;
                T acc = T(0);
                for (int u = 0; u < nx; ++u) acc += As[r * nx + u] * Ss[u * nx + s];
                T1s[e2] = acc;
            }
            __syncthreads();
            // the last wavefront takes the norms, the others S_{k+1}
            for (int i2 = 319 - tid; i2 < mk; i2 += 320) {
                T acc = T(0);
                if (gC)
                    for (int u = 0; u < nx; ++u) acc += Ys[i2 * nx + u] * Cs[i2 * nx + u];
                if (gD)
                    for (int u = 0; u < nu; ++u) acc += Ds[i2 * nu + u] * Ds[i2 * nu + u];
                nrm[k * mk + i2] = (acc > T(0)) ? T(1) / sqrt(acc) : T(1);
            }
            for (int e2 = tid; e2 < nA; e2 += 320) {
                const int r = e2 / nx, s = e2 - r * nx;
                T acc = T(0);
                for (int u = 0; u < nx; ++u) acc += T1s[r * nx + u] * As[s * nx + u];
                for (int u = 0; u < nu; ++u) acc += Bs[r * nu + u] * Bs[s * nu + u];
                Ss[e2] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------ Gram, MFMA f32
// P = w_u I + sum_rows w_row Psi[row,:]' Psi[row,:]  (n x n, n a multiple of 32, <= 256)
// One workgroup of 8 wavefronts per problem; the NT (NT + 1) / 2 tiles of 32 x 32 of the LOWER triangle are
// accumulators of v_mfma_f32_32x32x2_f32 (the upper triangle is mirrored in the epilogue through LDS). Psi is streamed
// through LDS KC rows at a time, the next chunk requested into registers while this one feeds the matrix cores.
// Causality is used twice: row block k of Psi is zero from column k nu on (u_j with j >= k does not reach x_k,
// qpmpc/mpc_qp.py:80-90), so a chunk of rows is only staged up to that column and only the tiles whose ROW block starts
// below it do any MFMA: with the symmetry 5760 MFMAs per config-5 problem instead of the dense product's 24,576.
// Which wavefront owns which tile decides how busy the matrix cores are: the chunks are walked in lock step (one barrier
// pair per chunk), and at the chunk of step k only the tile rows I < ceil(k nu / 32) are active. Round 1-2 gave wavefront w
// tile row w: the last chunks kept one wavefront busy with 8 tiles while wavefront 0 had one (matrix cores <= 42 % busy by
// construction, 24 % measured). Now the tiles are dealt round-robin IN THEIR ORDER OF ACTIVATION (row-major over the lower
// triangle: tile t -> wavefront t mod 8, its slot t / 8): at every chunk the active tiles are the first c (c + 1) / 2 of
// that order, so the wavefronts' counts differ by at most one (>= 79 % by construction), and a wavefront holds at most
// ceil(36 / 8) = 5 accumulators instead of 8 (80 instead of 128 registers: two problems per CU).
// The tile's row stride in LDS is 256 + 32 floats: the two k-rows an MFMA operand read touches (lanes 0..31 / 32..63) fall
// into different halves of the 64 banks.
// q = Psi' W resid is accumulated by the first n threads from the same tile.
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NT>  // NT = n / 32 tile rows
__global__ void __launch_bounds__(512, 4) mpcqp_gram_mfma_f32_kernel(const KernelArgs ka, const float *__restrict__ Psi_ws,
                                                                      const float *__restrict__ res_ws,
                                                                      float *__restrict__ oP, float *__restrict__ oq)
{
    constexpr int NTL = NT * (NT + 1) / 2;  // tiles of the lower triangle
    constexpr int SL = (NTL + 7) / 8;       // accumulators per wavefront
    constexpr int LDT = 256 + 32;           // row stride of the staged chunk
    // two chunk buffers (the next chunk lands in one while the matrix cores read the other), the weighted residuals of every
    // row; the epilogue's transposers reuse the first buffer
    extern __shared__ float rr[];  // (dynamic: (N + 1) nx + KC floats, rows of Psi_all rounded up past the last chunk)
    __shared__ __attribute__((aligned(16))) float tile0[KC * LDT];
    __shared__ __attribute__((aligned(16))) float tile1[KC * LDT];
    static_assert(8 * 32 * 33 <= KC * LDT, "the transposers must fit the first chunk buffer");
    const int nx = ka.nx, nu = ka.nu, N = ka.N, n = ka.n;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t prob = blockIdx.x;
    const int K = (N + 1) * nx;
    const float *Psi = Psi_ws + prob * (int64_t)K * n;
    const float *res = res_ws + prob * (int64_t)K;
    const float wxp = (ka.flags & MPCQP_P_STAGE) ? (float)ka.wx : 0.0f;
    const float wtp = (ka.flags & MPCQP_P_TERMINAL) ? (float)ka.wt : 0.0f;
    const float wxq = (ka.flags & MPCQP_Q_STAGE) ? (float)ka.wx : 0.0f;
    const float wtq = (ka.flags & MPCQP_Q_TERMINAL) ? (float)ka.wt : 0.0f;
    // this wavefront's tiles: t = wv + 8 s -> (I, J) with t = I (I + 1) / 2 + J, J <= I (wavefront-uniform)
    int tI[SL], tJ[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int t = wv + 8 * s;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        tI[s] = t < NTL ? I : NT;  // (a slot without a tile never becomes active: its row block starts at n)
        tJ[s] = t < NTL ? t - I * (I + 1) / 2 : 0;
    }
    f32x16 acc[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
    float qacc = 0.0f;
    auto chunk_cols = [&](int row0, int &cnz, int &cmax) {
        const int klast = min(row0 + KC - 1, K - 1) / nx;  // columns that can be non-zero: below k_last nu
        cnz = min(n, klast * nu);
        cmax = min(n, (cnz + 31) & ~31);
    };
    // Both buffers start as zeros: a chunk is only staged up to its last non-zero column block, and that bound never
    // shrinks along the rows, so what lies beyond it in a buffer has never been written. (Rows past the end of Psi in the
    // last chunk keep an older chunk's values: their weight is zero.)
    for (int i = tid; i < KC * LDT; i += 512) tile0[i] = tile1[i] = 0.0f;
    for (int i = tid; i < K + KC; i += 512) {
        const bool term = i >= N * nx;
        rr[i] = i < K ? (term ? wtq : wxq) * res[i] : 0.0f;
    }
    __syncthreads();
    // staging: straight into LDS (global_load_lds_dwordx4: no registers, no LDS store pass): wavefront w carries rows w, w + 8,
    // ... of the chunk, one instruction per row (lane l the columns 4 l .. 4 l + 3: 1 KB contiguous in LDS)
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    auto request = [&](int row0, float *dst) {
        int cnz, cmax;
        chunk_cols(row0, cnz, cmax);
#pragma unroll
        for (int u = 0; u < KC / 8; ++u) {
            const int r = wv + 8 * u, row = row0 + r;
            if (row < K && 4 * lane < cmax)
                __builtin_amdgcn_global_load_lds((glb_void *)(Psi + (int64_t)row * n + 4 * lane), (lds_void *)(dst + r * LDT), 16, 0, 0);
        }
    };
    int cA[SL], cB[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        cA[s] = __builtin_amdgcn_readfirstlane(32 * (tI[s] < NT ? tI[s] : 0));
        cB[s] = __builtin_amdgcn_readfirstlane(32 * tJ[s]);
    }
    // one chunk: wait for it (this wavefront's requests, then everybody's: the barrier also says that every wavefront has
    // finished reading the OTHER buffer), request the next chunk into that other buffer, feed the matrix cores
    auto chunk = [&](int row0, const float *cur, float *nxt) __attribute__((always_inline)) {
        int cnz, cmax;
        chunk_cols(row0, cnz, cmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (row0 + KC < K) request(row0 + KC, nxt);
        // slots are in activation order: the active ones are a prefix
        int nact = 0;
#pragma unroll
        for (int s = 0; s < SL; ++s) nact += (32 * tI[s] < cnz) ? 1 : 0;
        // operand reads: this lane's k-row of an MFMA step and its column inside a tile (a register), the tile's column
        // block (wavefront-uniform: a scalar), the step (an immediate)
        const float *tlane = cur + (lane >> 5) * LDT + l31;
        if (nact > 0) {
#pragma unroll 1
            for (int k0 = 0; k0 < KC; k0 += 8) {
#pragma unroll
                for (int kk = 0; kk < 8; kk += 2) {
                    // A[i][k] = w_k Psi[k][32 I + i],  B[k][j] = Psi[k][32 J + j]; k = k0 + kk + (lane >> 5)
                    const int row = row0 + k0 + kk + (lane >> 5);
                    const float wk = row < N * nx ? wxp : (row < K ? wtp : 0.0f);
#pragma unroll
                    for (int s = 0; s < SL; ++s) {
                        if (s < nact) {
                            const float av = wk * tlane[cA[s] + (k0 + kk) * LDT];
                            const float bv = tlane[cB[s] + (k0 + kk) * LDT];
                            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[s], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (tid < cmax) {
#pragma unroll
            for (int r = 0; r < KC; ++r) qacc += rr[row0 + r] * cur[r * LDT + tid];
        }
    };
    // block 0 of Psi is zero: start at row nx
    request(nx, tile0);
    for (int row0 = nx; row0 < K; row0 += 2 * KC) {
        chunk(row0, tile0, tile1);
        if (row0 + KC < K) chunk(row0 + KC, tile1, tile0);
    }
    __syncthreads();  // (the transposers below overwrite the first chunk buffer)
    float(*tr)[32 * 33] = reinterpret_cast<float(*)[32 * 33]>(tile0);
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *P = oP + prob * (int64_t)n * n;
    const float wu = (float)ka.wu;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int I = tI[s], J = tJ[s];
        if (I >= NT) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int i = 32 * I + il, jj = 32 * J + l31;
            const float v = acc[s][r] + ((i == jj) ? wu : 0.0f);
            P[(int64_t)i * n + jj] = v;
            if (J < I) tr[wv][il * 33 + l31] = v;
        }
        if (J < I) {
            // mirrored tile P[32 J + j][32 I + i], written with i across the lanes (coalesced)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                P[(int64_t)(32 * J + jl) * n + 32 * I + l31] = tr[wv][l31 * 33 + jl];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (tid < n) oq[prob * (int64_t)n + tid] = qacc;
}

// Generic Gram (any n, any dtype): one thread per entry of the lower triangle is not
// worth an MFMA path for float64 here; plain LDS-tiled VALU version, used for f64 or
// when n is not a multiple of 32.
template <typename T>
__global__ void __launch_bounds__(256) mpcqp_gram_valu_kernel(const KernelArgs ka, const T *__restrict__ Psi_ws,
                                                              const T *__restrict__ res_ws, T *__restrict__ oP,
                                                              T *__restrict__ oq)
{
    const int nx = ka.nx, N = ka.N, n = ka.n;
    const int64_t prob = blockIdx.x;
    const int K = (N + 1) * nx;
    const T *Psi = Psi_ws + prob * (int64_t)K * n;
    const T *res = res_ws + prob * (int64_t)K;
    const T wxp = (ka.flags & MPCQP_P_STAGE) ? (T)ka.wx : T(0), wtp = (ka.flags & MPCQP_P_TERMINAL) ? (T)ka.wt : T(0);
    const T wxq = (ka.flags & MPCQP_Q_STAGE) ? (T)ka.wx : T(0), wtq = (ka.flags & MPCQP_Q_TERMINAL) ? (T)ka.wt : T(0);
    T *P = oP + prob * (int64_t)n * n;
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int a = idx / n, b = idx - a * n;
        T acc = (a == b) ? (T)ka.wu : T(0), st = T(0), tt = T(0);
        for (int row = nx; row < N * nx; ++row) st += Psi[(int64_t)row * n + a] * Psi[(int64_t)row * n + b];
        for (int row = N * nx; row < K; ++row) tt += Psi[(int64_t)row * n + a] * Psi[(int64_t)row * n + b];
        P[idx] = acc + wtp * tt + wxp * st;
    }
    for (int a = threadIdx.x; a < n; a += blockDim.x) {
        T st = T(0), tt = T(0);
        for (int row = nx; row < N * nx; ++row) st += res[row] * Psi[(int64_t)row * n + a];
        for (int row = N * nx; row < K; ++row) tt += res[row] * Psi[(int64_t)row * n + a];
        oq[prob * (int64_t)n + a] = wtq * tt + wxq * st;
    }
}

// ------------------------------------------------------------------ host side
// Workspace (elements of T) the large path needs per problem for the condensing:
// Psi_all and the residual vector.
size_t big_condense_ws_elems(const KernelArgs &ka) { return (size_t)(ka.N + 1) * ka.nx * (ka.n + 1); }

bool big_supported(const KernelArgs &ka) { return ka.nx <= NXMAX && ka.n <= 256; }

int launch_big_condense(const KernelArgs &ka, int dtype, int64_t batch, void *Psi_ws, void *res_ws, void *P, void *q,
                        void *G, void *h, void *rownorm_inv, hipStream_t st, int phase)
{
    // phase 0: both launches; 1: propagation only (Psi, residuals, G, h); 2: the Gram product only (P, q from Psi)
    const size_t esz = dtype == MPCQP_F64 ? 8 : 4;
    auto al4 = [](size_t c) { return (c + 3) & ~(size_t)3; };
    const size_t lds = (3 * al4((size_t)ka.nx * ka.nx) + al4((size_t)ka.nx * ka.nu) + 2 * al4((size_t)ka.mk * ka.nx) +
                        al4((size_t)ka.mk * ka.nu) + al4((size_t)ka.mk) + 2 * (size_t)ka.nx + 8) * esz;
#define PROPAGATE(TY, NRMV, NXV)                                                                                \
    hipLaunchKernelGGL((mpcqp_propagate_kernel<TY, NRMV, NXV>), dim3((unsigned)batch), dim3(320), lds, st, ka,  \
                       (TY *)Psi_ws, (TY *)res_ws, (TY *)G, (TY *)h, (TY *)rownorm_inv)
    if (phase == 2) {
    } else if (dtype == MPCQP_F64) {
        if (rownorm_inv) PROPAGATE(double, true, 0); else PROPAGATE(double, false, 0);
    } else if (ka.nx == 12) {  // config 5's state dimension at compile time
        if (rownorm_inv) PROPAGATE(float, true, 12); else PROPAGATE(float, false, 12);
    } else {
        if (rownorm_inv) PROPAGATE(float, true, 0); else PROPAGATE(float, false, 0);
    }
#undef PROPAGATE
    int rc = (int)hipGetLastError();
    if (rc || phase == 1) return rc;
    const bool mfma = (dtype == MPCQP_F32) && (ka.n % 32 == 0) && (ka.n <= 256);
    if (mfma) {
        const float *ps = (const float *)Psi_ws, *rs = (const float *)res_ws;
        switch (ka.n / 32) {
#define GRAM_CASE(NTV)                                                                                              \
    case NTV:                                                                                                       \
        hipLaunchKernelGGL(mpcqp_gram_mfma_f32_kernel<NTV>, dim3((unsigned)batch), dim3(512),                      \
                           ((size_t)(ka.N + 1) * ka.nx + KC) * sizeof(float), st, ka, ps, rs,                      \
                           (float *)P, (float *)q);                                                                 \
        break;
            GRAM_CASE(1) GRAM_CASE(2) GRAM_CASE(3) GRAM_CASE(4) GRAM_CASE(5) GRAM_CASE(6) GRAM_CASE(7) GRAM_CASE(8)
#undef GRAM_CASE
        }
    } else if (dtype == MPCQP_F64) {
        hipLaunchKernelGGL(mpcqp_gram_valu_kernel<double>, dim3((unsigned)batch), dim3(256), 0, st, ka,
                           (const double *)Psi_ws, (const double *)res_ws, (double *)P, (double *)q);
    } else {
        hipLaunchKernelGGL(mpcqp_gram_valu_kernel<float>, dim3((unsigned)batch), dim3(256), 0, st, ka,
                           (const float *)Psi_ws, (const float *)res_ws, (float *)P, (float *)q);
    }
    return (int)hipGetLastError();
}

}  // namespace mpcqp
