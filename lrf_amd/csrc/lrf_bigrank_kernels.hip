// lrf_bigrank_kernels.hip — table layouts and helpers shared by the kernels for ranks above 16 (rank pitch 64): the gt table of
// b = v.mT @ v, its builder, the generic ordered Gauss-Seidel terms (ATen-native order of tiny matrices), the LDS carve of the
// V update.  The kernels themselves: lrf_midrank_kernels.hip (ranks 17..32); ranks 33..64 run their iterations on the
// any-shape kernels (lrf_api.hip, LRF_BIG_TO_ANY_RANK).  A first, correctness-only kernel pair for ranks 17..64 lived here
// (k_bcd_big / k_vupdate_big: rank padded to four MFMA tiles, no prefetch, one wave of four solving each sub-tile): 5.1 ms
// per 64 images at ranks (20,10,10) against 2.5 ms for its successor.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XS_LD 66 // LDS row stride (dwords) of the X sub-tile: conflict-free B-operand reads
#define LRF_RPB 64                      // padded rank
#define LRF_GTB_LD 68                   // gt table pitch: <= 63 `bb` entries, [64] = 1/den (unused here), [65] = den
#define LRF_GTB_DEN 65
#define LRF_GTB_STRIDE (LRF_RPB * LRF_GTB_LD)

// acc += uu[n] * bb[n] for n = start, start + step, ... (count terms, in that order; uu = the row without column r).
// Eight terms at a time: the sixteen LDS reads of a chunk are issued together instead of one exposed latency per term.
__device__ __forceinline__ float gs_chain(const float* u_row, int r, const float* bb, int start, int step, int count, float acc)
{
    int i = 0;
    for (; i + 8 <= count; i += 8) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int n = start + (i + j) * step;
            p[j] = u_row[n < r ? n : n + 1] * bb[n];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc = acc + p[j];
    }
    for (; i < count; i++) {
        const int n = start + i * step;
        acc = acc + u_row[n < r ? n : n + 1] * bb[n];
    }
    return acc;
}

// term2 = uu . bb of column r in the reference's order (qmf.py:115): ATen native chain, or the MKL single-column tree
// ((fma(u1,b1,u0*b0) + p_lastodd + ... + p3) + (p2 + p4 + ...)), oracle/lrf_oracle.c dot_mkl_n1
__device__ __forceinline__ float gs_term2_generic(const float* u_row, int r, const float* bb, int K, bool native)
{
    if (K <= 0) return 0.f;
#define UU(n) u_row[(n) < r ? (n) : (n) + 1]
    if (native) return gs_chain(u_row, r, bb, 0, 1, K, 0.f);
    if (K == 1) return UU(0) * bb[0];
    float odd = fmaf(UU(1), bb[1], UU(0) * bb[0]);
    const int last_odd = ((K - 1) & 1) ? K - 1 : K - 2;
    if (last_odd >= 3) odd = gs_chain(u_row, r, bb, last_odd, -2, (last_odd - 3) / 2 + 1, odd);
    if (K < 3) return odd;
    float even = UU(2) * bb[2];
    if (K > 4) even = gs_chain(u_row, r, bb, 4, 2, (K - 1 - 4) / 2 + 1, even);
#undef UU
    return odd + even;
}

// gt table (pitch LRF_GTB_LD) of b = v.mT @ v from a [depth][LRF_RPB] factor
__device__ __forceinline__ void make_gtable_big(const float* Vp, int depth, int R, float* gt, int tid, int nthreads)
{
    bool native = (long)depth * R * R < 400;
    for (int i = tid; i < R * R; i += nthreads) {
        int j = i / R, r = i - j * R;
        float acc = 0.f;
        if (native) {
            for (int k = 0; k < depth; k++) {
                float p = Vp[k * LRF_RPB + j] * Vp[k * LRF_RPB + r];
                acc = acc + p;
            }
        } else {
            for (int k = 0; k < depth; k++) acc = fmaf(Vp[k * LRF_RPB + j], Vp[k * LRF_RPB + r], acc);
        }
        if (j == r) gt[r * LRF_GTB_LD + LRF_GTB_DEN] = (acc + 0.f) + LRF_EPS;
        else gt[r * LRF_GTB_LD + (j < r ? j : j - 1)] = acc;
    }
}

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
