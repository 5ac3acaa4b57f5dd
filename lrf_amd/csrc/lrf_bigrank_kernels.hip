#pragma once
// lrf_bigrank_kernels.hip — table layouts and helpers shared by the kernels for ranks above 16 (rank pitch 64): the gt table of
// b = v.mT @ v, its builder, the generic ordered Gauss-Seidel terms (ATen-native order of tiny matrices), the LDS carve of the
// V update.  The kernels themselves: lrf_midrank_kernels.hip (ranks 17..32); ranks 33..64 run their iterations on the
// any-shape kernels (lrf_api.hip, LRF_BIG_TO_ANY_RANK).  A first, correctness-only kernel pair for ranks 17..64 lived here
// (k_bcd_big / k_vupdate_big: rank padded to four MFMA tiles, no prefetch, one wave of four solving each sub-tile): 5.1 ms
// per 64 images at ranks (20,10,10) against 2.5 ms for its successor.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lrf_device.h"

#define XS_LD 66 // LDS row stride (dwords) of the X sub-tile: conflict-free B-operand reads

__global__ __launch_bounds__(256) void k_bprep_big(const PlaneDesc* __restrict__ planes, const float* __restrict__ Vf,
                                                   float* __restrict__ Bf, int plane0)
{
    __shared__ float v_s[64 * LRF_RPB];
    const int pli = blockIdx.x + plane0;
    for (int i = threadIdx.x; i < 64 * LRF_RPB; i += 256) v_s[i] = Vf[(long)pli * 64 * LRF_RPB + i];
    __syncthreads();
    make_gtable_big(v_s, 64, planes[pli].R, Bf + (long)pli * LRF_GTB_STRIDE, threadIdx.x, 256);
}

struct BigVLds {
    float a_s[64 * LRF_RPB];
    float v_s[64 * LRF_RPB];
    float gt_s[LRF_GTB_STRIDE];
};
