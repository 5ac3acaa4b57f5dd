// read_probe.hip — what a streaming read of 617 MB reaches on this part as a function of the bytes each wave keeps in
// flight and of the number of waves per CU (the two things k_bcd_w is short of).  Each wave owns contiguous 96 KB blocks
// (like a k_bcd_w block) and reads them in chunks of DEPTH x 1 KB (DEPTH float4 loads per lane, all issued before the first
// is used).  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libread_probe.so read_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(64) void k_read(const float* __restrict__ X, float* __restrict__ out, long block_floats, int lds_pad)
{
    extern __shared__ float pad[]; // occupancy control only
    const float* p = X + (long)blockIdx.x * block_floats + 4 * threadIdx.x;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long nchunk = block_floats / (256 * DEPTH);
    for (long c = 0; c < nchunk; c++) {
        f32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) v[d] = *reinterpret_cast<const f32x4*>(p + (c * DEPTH + d) * 256);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += v[d];
    }
    if (lds_pad < 0) pad[threadIdx.x] = acc[0];
    out[(long)blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

extern "C" int probe_read(double* ms_out /*[n]*/, const int* depth, const int* lds_kb, int n)
{
    const long total = 617349120L / 4, block_floats = 24576; // 96 KB blocks
    const int nblocks = (int)(total / block_floats);
    float *X, *out;
    if (hipMalloc(&X, total * 4) != hipSuccess || hipMalloc(&out, (size_t)nblocks * 256) != hipSuccess) return 1;
    hipMemset(X, 0, total * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < n; i++) {
        size_t lds = (size_t)lds_kb[i] * 1024;
        auto launch = [&]() {
            switch (depth[i]) {
            case 4: hipFuncSetAttribute((const void*)k_read<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k_read<4>, dim3(nblocks), dim3(64), lds, 0, X, out, block_floats, 0); break;
            case 8: hipFuncSetAttribute((const void*)k_read<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k_read<8>, dim3(nblocks), dim3(64), lds, 0, X, out, block_floats, 0); break;
            case 16: hipFuncSetAttribute((const void*)k_read<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k_read<16>, dim3(nblocks), dim3(64), lds, 0, X, out, block_floats, 0); break;
            default: hipFuncSetAttribute((const void*)k_read<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k_read<32>, dim3(nblocks), dim3(64), lds, 0, X, out, block_floats, 0); break;
            }
        };
        for (int w = 0; w < 3; w++) launch();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; r++) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        ms_out[i] = ms / 10;
    }
    hipFree(X); hipFree(out);
    return 0;
}
