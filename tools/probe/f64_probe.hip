// Developer probe (not part of the product): issue cost of the fp64 vector instructions the Householder tridiagonalisations
// are built from (k_init, k_any_tridiag_reg), of v_readlane_b32 (a wave-uniform operand out of a VGPR), of broadcast LDS
// reads and of the fp64 matrix instruction — shader cycles per instruction and wave (s_memtime ticks), one wave per SIMD
// (64 / 256 threads) and two (512) or three (768).
//   hipcc -O3 --offload-arch=gfx950 -o f64_probe f64_probe.hip && ./f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

#define NACC 8
template <int KIND>
__global__ void k_rate(double* out, unsigned long long* cyc, int iters, double seed)
{
    __shared__ __attribute__((aligned(16))) double lds[512];
    const int l = threadIdx.x;
    double acc[NACC];
    double a = seed + l, b = seed * 3 + l;
    int si = 0;
    for (int i = 0; i < NACC; i++) acc[i] = i + l;
    lds[l & 511] = a;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 2) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
                if (KIND == 3) { // a wave-uniform fp64 operand fetched from lane 5 of a register: two readlanes into SGPRs
                    const int lo = __builtin_amdgcn_readlane(__double2loint(acc[i]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(acc[i]), 5);
                    asm volatile("" ::"s"(lo), "s"(hi));
                }
                if (KIND == 4) { // the same with the fma that uses it
                    const double u = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 5 + i), __builtin_amdgcn_readlane(__double2loint(a), 5 + i));
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(u), "v"(b));
                }
                if (KIND == 5) { // two doubles per broadcast LDS read (every lane the same address), then two fmas
                    const f64x2 q = *reinterpret_cast<const volatile f64x2*>(&lds[2 * i + 16 * rep]);
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(q[0]), "v"(b));
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(q[1]), "v"(b));
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = si;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + l] = s + a;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (l >> 6)] = t1 - t0;
}

__global__ void k_chain(double* out, unsigned long long* cyc, int iters, double seed)
{
    const int l = threadIdx.x;
    double acc = l, a = seed + l, b = seed * 3 + l;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 32; rep++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = acc;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (l >> 6)] = t1 - t0;
}

__global__ void k_mfma_f64(double* out, unsigned long long* cyc, int iters)
{
    const int l = threadIdx.x;
    double a = l, b = l * 3;
    f64x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f64x4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = s;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (l >> 6)] = t1 - t0;
}

static const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "2 x v_readlane_b32", "2 x v_readlane_b32 + v_fma_f64 (SGPR operand)",
                              "ds_read_b128 (broadcast) + 2 x v_fma_f64"};

template <int KIND>
static void run_rate(double* out, unsigned long long* cyc)
{
    const int iters = 512;
    for (int threads : {64, 256, 512, 768}) {
        unsigned long long h[12];
        hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 7.0);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(unsigned long long) * (threads / 64), hipMemcpyDeviceToHost);
        double m = 0;
        for (int w = 0; w < threads / 64; w++) m = h[w] > m ? h[w] : m;
        printf("%-50s %d wave(s)/SIMD (%3d threads): %.2f cycles per group per wave\n", names[KIND], threads <= 256 ? 1 : threads / 256, threads,
               m / (iters * 4.0 * NACC));
    }
}

int main()
{
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 8 * 1024);
    hipMalloc(&cyc, 8 * 64);
    run_rate<0>(out, cyc); run_rate<1>(out, cyc); run_rate<2>(out, cyc); run_rate<3>(out, cyc); run_rate<4>(out, cyc); run_rate<5>(out, cyc);
    {
        unsigned long long h;
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, cyc, 512, 7.0);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_fma_f64 dependent chain: %.2f cycles per instruction\n", (double)h / (512 * 32.0));
    }
    for (int threads : {64, 256, 512}) {
        unsigned long long h[8];
        hipLaunchKernelGGL(k_mfma_f64, dim3(1), dim3(threads), 0, 0, out, cyc, 512);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
        double m = 0;
        for (int w = 0; w < threads / 64; w++) m = h[w] > m ? h[w] : m;
        printf("v_mfma_f64_16x16x4_f64, %d threads: %.2f cycles per MFMA per wave (2048 flops each)\n", threads, m / (512 * 4.0));
    }
    return 0;
}
