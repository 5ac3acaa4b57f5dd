// Developer probe (not part of the product): issue cost of the vector instructions the rank-17..32 Gauss-Seidel could be
// built from, one wave per SIMD and two waves per SIMD (shader cycles per instruction and wave, s_memtime ticks).
//   hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define NACC 8
template <int KIND>
__global__ void k_rate(int* out, unsigned long long* cyc, int iters, int seed)
{
    const int l = threadIdx.x;
    int acc[NACC];
    int a = seed + l, b = seed * 3 + l;
    for (int i = 0; i < NACC; i++) acc[i] = i + l;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 2) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 3) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 4) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 5) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 6) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 7) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 8) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(acc[i]) : "v"(a));
                if (KIND == 9) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 10) asm volatile("v_rndne_f32 %0, %1" : "=v"(acc[i]) : "v"(a));
                if (KIND == 11) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(acc[i]), "+v"(a));
                if (KIND == 12) asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(acc[i]) : "v"(a));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + l] = s + a;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (l >> 6)] = t1 - t0;
}

// dependent chain: each instruction consumes the previous one's result
template <int KIND>
__global__ void k_chain(int* out, unsigned long long* cyc, int iters, int seed)
{
    const int l = threadIdx.x;
    int acc = l, a = seed + l, b = seed * 3 + l;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 32; rep++) {
            if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            if (KIND == 3) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = acc;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (l >> 6)] = t1 - t0;
}

// i8 MFMA 16x16x64 cost, independent accumulators
__global__ void k_mfma_i8(int* out, unsigned long long* cyc, int iters)
{
    const int l = threadIdx.x;
    i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9};
    i32x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (i32x4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    int s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = s;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (l >> 6)] = t1 - t0;
}

static const char* names[] = {"v_fmac_f32", "v_dot2c_i32_i16", "v_mad_i32_i16", "v_dot4c_i32_i8", "v_mad_i32_i24",
                              "v_mad_i32_i16 op_sel hi", "v_dot2_i32_i16 (VOP3P)", "v_fmac_f32_dpp row_newbcast", "v_cvt_f32_i32",
                              "v_perm_b32", "v_rndne_f32", "v_permlane16_swap_b32", "v_lshl_or_b32"};

template <int KIND>
static void run_rate(int* out, unsigned long long* cyc)
{
    const int iters = 1024;
    for (int threads : {64, 256, 512}) {
        unsigned long long h[8];
        hipLaunchKernelGGL((k_rate<KIND>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 7);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(unsigned long long) * (threads / 64), hipMemcpyDeviceToHost);
        double m = 0;
        for (int w = 0; w < threads / 64; w++) m = h[w] > m ? h[w] : m;
        printf("%-30s %d wave(s)/SIMD-set (%3d threads): %.2f cycles per instruction per wave\n", names[KIND], threads == 512 ? 2 : 1, threads,
               m / (iters * 4.0 * NACC));
    }
}
template <int KIND>
static void run_chain(int* out, unsigned long long* cyc)
{
    const int iters = 1024;
    unsigned long long h;
    hipLaunchKernelGGL((k_chain<KIND>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 7);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-30s dependent chain: %.2f cycles per instruction\n", names[KIND], (double)h / (iters * 32.0));
}

int main()
{
    int* out;
    unsigned long long* cyc;
    hipMalloc(&out, 4 * 1024);
    hipMalloc(&cyc, 8 * 64);
    run_rate<0>(out, cyc); run_rate<1>(out, cyc); run_rate<2>(out, cyc); run_rate<3>(out, cyc); run_rate<4>(out, cyc);
    run_rate<5>(out, cyc); run_rate<6>(out, cyc); run_rate<7>(out, cyc); run_rate<8>(out, cyc); run_rate<9>(out, cyc);
    run_rate<10>(out, cyc); run_rate<11>(out, cyc); run_rate<12>(out, cyc);
    run_chain<0>(out, cyc); run_chain<1>(out, cyc); run_chain<2>(out, cyc); run_chain<3>(out, cyc);
    for (int threads : {64, 256, 512}) {
        unsigned long long h[8];
        hipLaunchKernelGGL(k_mfma_i8, dim3(1), dim3(threads), 0, 0, out, cyc, 1024);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
        double m = 0;
        for (int w = 0; w < threads / 64; w++) m = h[w] > m ? h[w] : m;
        printf("v_mfma_i32_16x16x64_i8, %d threads: %.2f cycles per MFMA per wave\n", threads, m / (1024 * 4.0));
    }
    return 0;
}
