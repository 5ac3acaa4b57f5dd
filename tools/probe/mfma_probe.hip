// Developer probe (not part of the product): lane layout and cost of the f32 MFMA forms on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* a, const float* b, float* d)
{
    int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[l * 4 + r] = c[r];
}

template <int KIND, int NACC>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters)
{
    int l = threadIdx.x;
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

extern "C" int probe_run(float* d_host /*256*/, double* rates /*4*/)
{
    float ha[64], hb[64];
    for (int l = 0; l < 64; l++) { ha[l] = (float)(l + 1); hb[l] = (float)(67 + l); }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(d_host, d, 1024, hipMemcpyDeviceToHost);
    float* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64 * 4 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int iters = 4096;
    hipLaunchKernelGGL((k_rate<0, 1>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); rates[0] = (double)h / iters;
    hipLaunchKernelGGL((k_rate<0, 4>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); rates[1] = (double)h / iters / 4;
    hipLaunchKernelGGL((k_rate<1, 1>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); rates[2] = (double)h / iters;
    hipLaunchKernelGGL((k_rate<1, 4>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); rates[3] = (double)h / iters / 4;
    hipDeviceSynchronize();
    return 0;
}
