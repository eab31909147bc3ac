// Micro-benchmark (dev tool): lane layout and cost of v_mfma_f64_4x4x4_4b_f64 on gfx950 (four independent 4 x 4 x 4 products,
// one value of A, B, C / D per lane).
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma44 tools/ubench/mfma_f64_4x4.hip && gpurun_out/mfma44
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void one(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int MODE> __global__ void rate(double *out, long long *cyc, double x)
{
    double acc[8];
    double a = x + threadIdx.x * 1e-9, b = 1.0 - x;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = i * 1e-3;
    long long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
        if (MODE == 0) {  // independent
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {  // dependent through C
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 2) {  // dependent through A (result feeds the next A operand)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[0], b, 0.0, 0, 0, 0);
        } else if (MODE == 4) {  // dependent through B
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc[0], 0.0, 0, 0, 0);
        } else if (MODE == 5) {  // a sweep step: the chain through B, an independent product on the same operand between two links
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, acc[0], 0.0, 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc[0], b, 0, 0, 0);
                asm volatile("" : "+v"(acc[1]));
            }
        } else if (MODE == 6) {  // ... and the side product stored to LDS
            __shared__ double buf[64 * 8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, acc[0], 0.0, 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, acc[0], b, 0, 0, 0);
                buf[threadIdx.x + 64 * i] = acc[1];
            }
        } else {  // MFMA -> VALU -> MFMA
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[0], b, 0.0, 0, 0, 0);
                acc[0] = acc[0] * x;
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    hipMalloc(&da, 512), hipMalloc(&db, 512), hipMalloc(&dd, 512);
    // A value = 1000 block + 10 (l % 4) + (l / 4) % 4 coded so that products identify their operands: use powers instead
    // unit probes: A = e_(p), B = all distinct -> D tells which B lanes a given A lane meets
    for (int p = 0; p < 16; ++p) {
        for (int l = 0; l < 64; ++l) ha[l] = (l % 16 == p) ? 1.0 : 0.0, hb[l] = 100 * (l / 16) + (l % 16) + 1;
        hipMemcpy(da, ha, 512, hipMemcpyHostToDevice), hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
        printf("A lane %2d (every block) = 1:", p);
        for (int l = 0; l < 16; ++l)
            if (hd[l] != 0.0) printf("  D[%d]=B[%d]", l, (int)hd[l] - 1);
        printf("   | block 2:");
        for (int l = 32; l < 48; ++l)
            if (hd[l] != 0.0) printf(" D[%d]=%g", l, hd[l]);
        printf("\n");
    }
    double *out;
    long long *cyc, hc;
    hipMalloc(&out, 64 * 8), hipMalloc(&cyc, 8);
    const char *names[] = {"independent", "dependent via C", "dependent via A", "MFMA -> v_mul -> MFMA", "dependent via B",
                           "chain via B + side product", "chain via B + side product + ds_write"};
    for (int m = 0; m < 7; ++m) {
        for (int r = 0; r < 2; ++r) {
            if (m == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            if (m == 1) hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            if (m == 2) hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            if (m == 3) hipLaunchKernelGGL(rate<3>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            if (m == 4) hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            if (m == 5) hipLaunchKernelGGL(rate<5>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            if (m == 6) hipLaunchKernelGGL(rate<6>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
            hipDeviceSynchronize();
        }
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-40s %.1f cycles per %s\n", names[m], hc / (256.0 * 8), m == 3 ? "MFMA + mul" : m >= 5 ? "step (two MFMAs)" : "MFMA");
    }
    return 0;
}
