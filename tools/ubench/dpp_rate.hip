// Micro-benchmark (dev tool): issue cost of float64 FMAs with and without a DPP row broadcast on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dpp_rate tools/ubench/dpp_rate.hip && gpurun_out/dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define NACC 16
#define ITER 512
template <int MODE> __global__ void k(double *out, long long *cyc, double xin, double min_)
{
    double acc[NACC];
    double x = xin + threadIdx.x * 1e-9, m = min_;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i * 1e-3;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(m));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x), "v"(m));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                double t;
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(t), "v"(m));
            }
        } else if (MODE == 3) {  // dependent chain, plain
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[0]) : "v"(x), "v"(m));
        } else if (MODE == 4) {  // dependent chain, dpp
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[0]) : "v"(x), "v"(m));
        } else if (MODE == 5) {  // mov dpp only
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(acc[i]) : "v"(x));
        } else if (MODE == 6) {  // v_mul_f64
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(acc[i]) : "v"(x), "v"(m));
        } else if (MODE == 7) {  // 32-bit dpp mov pair (two v_mov_b32_dpp)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                unsigned lo, hi;
                asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"((unsigned)__double2loint(x)));
                asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(hi) : "v"((unsigned)__double2hiint(x)));
                acc[i] += __hiloint2double((int)hi, (int)lo);
            }
        } else if (MODE == 8) {  // v_readlane pair + fma with SGPR operand
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 3), __builtin_amdgcn_readlane(__double2loint(x), 3));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "s"(s), "v"(m));
            }
        } else if (MODE >= 10 && MODE <= 15) {
            // the serial sweep's step: x_next = c + sum_j bcast_j(x) a_j, a second sum f interleaved; NACC/2 steps per iteration
#pragma unroll
            for (int i = 0; i < NACC / 2; ++i) {
                double pn = acc[1], fn = 0.0, p2 = 0.0;
                if (MODE == 10) {  // one dependent sum of four + interleaved second sum (what the kernel runs)
                    asm volatile("s_nop 1\n\t"
                                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %2, %3 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %2, %3 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                                 : "+v"(pn), "+v"(fn) : "v"(x), "v"(m));
                } else if (MODE == 11) {  // the same without DPP
                    asm volatile("s_nop 1\n\t"
                                 "v_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %3\n\tv_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %3\n\t"
                                 "v_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %3\n\tv_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %3"
                                 : "+v"(pn), "+v"(fn) : "v"(x), "v"(m));
                } else if (MODE == 12) {  // x's sum split in two halves + one add
                    asm volatile("s_nop 1\n\t"
                                 "v_fmac_f64_dpp %0, %3, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %2, %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %3, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %3, %4 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %2, %3, %4 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f64 %0, %0, %2\n\t"
                                 "v_fmac_f64_dpp %1, %3, %4 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %1, %3, %4 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                                 : "+v"(pn), "+v"(fn), "+v"(p2) : "v"(x), "v"(m));
                } else if (MODE == 13) {  // only the four dependent DPP FMAs
                    asm volatile("s_nop 1\n\t"
                                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                                 : "+v"(pn) : "v"(x), "v"(m));
                } else if (MODE == 14) {  // ONE DPP FMA per step (the bare loop-carried latency)
                    asm volatile("s_nop 1\n\t"
                                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf"
                                 : "+v"(pn) : "v"(x), "v"(m));
                } else {  // 15: one plain FMA per step
                    asm volatile("s_nop 1\n\tv_fmac_f64 %0, %1, %2" : "+v"(pn) : "v"(x), "v"(m));
                }
                x = pn;
                acc[0] += fn;
            }
        } else if (MODE == 9) {  // v_fma_f32 for reference
            float a32[NACC];
#pragma unroll
            for (int i = 0; i < NACC; ++i) a32[i] = (float)acc[i];
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a32[i]) : "v"((float)x), "v"((float)m));
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = a32[i];
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> void run(const char *name, int threads)
{
    double *out; long long *cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8 * 64);
    k<MODE><<<1, threads>>>(out, cyc, 1.000001, 0.999999);
    k<MODE><<<1, threads>>>(out, cyc, 1.000001, 0.999999);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-34s waves/SIMD %d: %7.2f clock64 ticks per instruction-slot (%d x %d)\n", name, threads / 256 ? threads / 256 : 1, (double)mx / (ITER * NACC), ITER, NACC);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int threads : {64, 256, 512}) {
        printf("---- block of %d threads\n", threads);
        run<0>("v_fmac_f64", threads);
        run<1>("v_fmac_f64_dpp row_newbcast", threads);
        run<2>("v_mov_b64_dpp + v_fmac_f64", threads);
        run<3>("v_fmac_f64 dependent", threads);
        run<4>("v_fmac_f64_dpp dependent", threads);
        run<5>("v_mov_b64_dpp", threads);
        run<6>("v_mul_f64", threads);
        run<7>("2 x v_mov_b32_dpp + v_add_f64", threads);
        run<8>("2 x v_readlane + v_fmac_f64 sgpr", threads);
        run<9>("v_fmac_f32 (+cvt outside?)", threads);
        run<10>("sweep step: 4+4 dpp fmac (x2 slots)", threads);
        run<11>("sweep step: 4+4 plain fmac", threads);
        run<12>("sweep step: split sum + add", threads);
        run<13>("sweep step: 4 dependent dpp fmac", threads);
        run<14>("sweep step: 1 dpp fmac", threads);
        run<15>("sweep step: 1 plain fmac", threads);
    }
    return 0;
}
