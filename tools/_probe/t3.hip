#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ __forceinline__ double bc(double x)
{
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + N, 0xf, 0xf, false);
}
__global__ void k(double *o, const double *a)
{
    double x = a[threadIdx.x];
    double r5 = bc<5>(x), r0 = bc<0>(x), r15 = bc<15>(x);
    o[threadIdx.x] = r5;
    o[64 + threadIdx.x] = r0;
    o[128 + threadIdx.x] = r15;
    // permlane16_swap of both words
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    o[192 + threadIdx.x] = __hiloint2double((int)sh[0], (int)sl[0]);
    o[256 + threadIdx.x] = __hiloint2double((int)sh[1], (int)sl[1]);
}
int main()
{
    double h[64], *d, *o, ho[320];
    for (int i = 0; i < 64; ++i) h[i] = 100.0 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int r = 0; r < 5; ++r) { for (int i = 0; i < 64; ++i) printf("%g ", ho[r * 64 + i]); printf("\n"); }
    return 0;
}
