// Developer probe (not part of the product): is v_mfma_f64_16x16x4_f64 on gfx950 a k-ordered chain of fp64 fmas per element?
// (k_any_gram's arithmetic is one fma chain per element, k ascending; the products of two fp32 values are exact in fp64.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma64(const double* A /*[16][K]*/, const double* B /*[K][16]*/, const double* C /*[16][16]*/, double* D, int K)
{
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f64x4 acc;
    for (int r = 0; r < 4; r++) acc[r] = C[(g + 4 * r) * 16 + i];
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[i * K + k0 + g], B[(k0 + g) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(g + 4 * r) * 16 + i] = acc[r];
}

extern "C" int probe64_run(const double* A, const double* B, const double* C, double* D, int K)
{
    double *a, *b, *c, *d;
    hipMalloc(&a, 16 * K * 8); hipMalloc(&b, 16 * K * 8); hipMalloc(&c, 2048); hipMalloc(&d, 2048);
    hipMemcpy(a, A, 16 * K * 8, hipMemcpyHostToDevice); hipMemcpy(b, B, 16 * K * 8, hipMemcpyHostToDevice);
    hipMemcpy(c, C, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma64, dim3(1), dim3(64), 0, 0, a, b, c, d, K);
    hipMemcpy(D, d, 2048, hipMemcpyDeviceToHost);
    hipFree(a); hipFree(b); hipFree(c); hipFree(d);
    return (int)hipDeviceSynchronize();
}

template <int NACC>
__global__ void k_rate64(double* out, unsigned long long* cyc, int iters)
{
    int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

extern "C" int probe64_rate(double* rates /*2: s_memtime ticks per MFMA, dependent / 4-way independent*/)
{
    double* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int iters = 4096;
    hipLaunchKernelGGL((k_rate64<1>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); rates[0] = (double)h / iters;
    hipLaunchKernelGGL((k_rate64<4>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); rates[1] = (double)h / iters / 4;
    return (int)hipDeviceSynchronize();
}
