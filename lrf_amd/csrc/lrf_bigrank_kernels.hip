// lrf_bigrank_kernels.hip — the BCD iteration for ranks 17..64 (quality sweeps beyond ~25: R = round(64 q / 100)).
//
// Same arithmetic and summation order as k_bcd / k_vupdate (lrf_kernels.hip; reference lrf/factorization/qmf.py:93-139),
// with the rank padded to 64 (four 16-wide MFMA tiles) and a generic, non-unrolled Gauss-Seidel that always uses the
// IEEE division.  Correctness-first: no prefetch pipeline, one workgroup per CU (about 100 KB of LDS).
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

// One row, all R columns (qmf.py:108-119), u_row updated in place (LDS); gt: table of b = v.mT @ v (layout as in gs_row).
__device__ __forceinline__ void gs_row_generic(int R, const float* a_row, float* u_row, const float* gt, bool native,
                                               float lo, float hi)
{
    const int K = R - 1;
    for (int r = 0; r < R; r++) {
        const float* bb = gt + r * LRF_GTB_LD;
        float term2 = gs_term2_generic(u_row, r, bb, K, native);
        float num = (a_row[r] - term2) + LRF_EPS;
        float val = rintf(num / bb[LRF_GTB_DEN]);
        u_row[r] = fminf(fmaxf(val, lo), hi);
    }
}

// The same row update when every product and partial sum of `uu @ bb` is an exact integer in fp32 (iterations >= 2: v is
// integer valued, |b| <= 64 mx^2, and the host checks (R - 1) * 64 * mx^3 < 2^24 for mx = max(|lo|, |hi|)): the order of
// the sum no longer matters, so the R (R - 1) dependent multiply-adds of the reference's chain become R (R - 1) independent
// fmas on RT accumulators — T[r] starts as the part of the sum over the not yet updated columns j > r (old values) and
// receives u_r b[r][r'] for every later column r' as soon as u_r is known.  Bit-identical results, a third of the time.
// gt rows are contiguous and wave-uniform: scalar loads.
template <int RT>
__device__ __forceinline__ void gs_row_exact(int R, const float* a_row, float* u_row, const float* __restrict__ gt, float lo, float hi)
{
    float T[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) T[r] = 0.f;
#pragma unroll
    for (int j = 1; j < RT; j++) {
        if (j < R) {
            const float uo = u_row[j];
            const float* bj = gt + j * LRF_GTB_LD; // bj[r] = b[j][r] for r < j
#pragma unroll
            for (int r = 0; r < j; r++) T[r] = fmaf(uo, bj[r], T[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RT; r++) {
        if (r < R) {
            const float* br = gt + r * LRF_GTB_LD; // br[r' - 1] = b[r][r'] for r' > r
            const float num = (a_row[r] - T[r]) + LRF_EPS;
            float val = rintf(num / br[LRF_GTB_DEN]);
            val = fminf(fmaxf(val, lo), hi);
            u_row[r] = val;
#pragma unroll
            for (int rn = r + 1; rn < RT; rn++) T[rn] = fmaf(val, br[rn - 1], T[rn]); // columns >= R: never read
        }
    }
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
                                                   float* __restrict__ Bf)
{
    __shared__ float v_s[64 * LRF_RPB];
    for (int i = threadIdx.x; i < 64 * LRF_RPB; i += 256) v_s[i] = Vf[(long)blockIdx.x * 64 * LRF_RPB + i];
    __syncthreads();
    make_gtable_big(v_s, 64, planes[blockIdx.x].R, Bf + (long)blockIdx.x * LRF_GTB_STRIDE, threadIdx.x, 256);
}

// 49 KB (65 KB in the first iteration): three workgroups per CU, i.e. three Gauss-Seidel waves at work per CU.  The b
// table stays in global memory (only the Gauss-Seidel wave reads it, with wave-uniform addresses: scalar loads) and the
// V operand of the U phase lives in 64 registers per lane (the W0 operand of the first iteration in LDS).
template <int MODE>
struct BigLds {
    float Xs[64 * XS_LD];
    float a_s[64 * LRF_RPB];
    float u_s[64 * LRF_RPB];
    float wa_s[MODE == 1 ? 4 * 16 * 64 : 4];
};

// MODE as in k_bcd.  Ppart / Qpart: per block [64][64] fp32.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_bcd_big(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                                 const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                                 const float* __restrict__ Wf, const float* __restrict__ Bf,
                                                 const float* __restrict__ U0, int8_t* __restrict__ U,
                                                 float* __restrict__ Ppart, float* __restrict__ Qpart, float lo, float hi,
                                                 int gs_exact /* see gs_row_exact; honoured for MODE 0, R <= 32 */)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BigLds<MODE>& L = *reinterpret_cast<BigLds<MODE>*>(smem);
    const BlockDesc bd = blocks[blockIdx.x];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RPB;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;

    const float* gt = Bf + (long)bd.plane * LRF_GTB_STRIDE;
    // A operand of a^T = V^T X^T for tile nt: lane needs V[4s + lq][16 nt + li] at k-step s
    float va[4][16];
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int s_ = 0; s_ < 16; s_++) va[nt][s_] = Vp[(4 * s_ + lq) * LRF_RPB + 16 * nt + li];
    if (MODE == 1) {
        for (int e = tid; e < 4 * 16 * 64; e += 256) {
            int nt = e >> 10, s_ = (e >> 6) & 15, l = e & 63;
            L.wa_s[e] = Wf[(long)bd.plane * 64 * LRF_RPB + (4 * s_ + (l >> 4)) * LRF_RPB + 16 * nt + (l & 15)];
        }
    }
    f32x4 accP[4], accQ[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { accP[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accQ[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int e = i * 256 + tid, row = e >> 4, c4 = e & 15;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r0 + row < nrows) v = *reinterpret_cast<const f32x4*>(Xp + (long)(r0 + row) * 64 + 4 * c4);
            float2* d = reinterpret_cast<float2*>(&L.Xs[row * XS_LD + 4 * c4]);
            d[0] = make_float2(v[0], v[1]);
            d[1] = make_float2(v[2], v[3]);
        }
        if (MODE != 1) { // old U rows of the sub-tile -> u_s, all threads, coalesced (a lane-by-lane loop in the
                         // Gauss-Seidel wave paid one exposed global latency per element)
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
            for (int e = tid; e < lim; e += 256) {
                const int row = e / R, r = e - row * R;
                L.u_s[row * LRF_RPB + r] = (MODE == 0) ? (float)Ub[(long)r0 * R + e]
                                                       : U0[pd.u0_off + ((long)bd.row0 + r0) * R + e];
            }
        }
        __syncthreads();
        { // a^T tiles for rows 16*wave..+15
            f32x4 acc[4], accw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accw[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            const float* xr = &L.Xs[(16 * wave + li) * XS_LD + lq];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                float bx = xr[4 * s];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[nt][s], bx, acc[nt], 0, 0, 0);
                    if (MODE == 1)
                        accw[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(L.wa_s[(nt * 16 + s) * 64 + lane], bx, accw[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                *reinterpret_cast<f32x4*>(&L.a_s[(16 * wave + li) * LRF_RPB + 16 * nt + 4 * lq]) = acc[nt];
                if (MODE == 1) *reinterpret_cast<f32x4*>(&L.u_s[(16 * wave + li) * LRF_RPB + 16 * nt + 4 * lq]) = accw[nt];
            }
        }
        __syncthreads();
        if (wave == (t & 3)) { // Gauss-Seidel, lane = row
            int row = r0 + lane;
            float* ur = &L.u_s[lane * LRF_RPB];
            if (row < nrows) {
#ifndef LRF_BIG_NO_GS
                if (MODE == 0 && gs_exact && R <= 24) gs_row_exact<24>(R, &L.a_s[lane * LRF_RPB], ur, gt, lo, hi);
                else if (MODE == 0 && gs_exact && R <= 32) gs_row_exact<32>(R, &L.a_s[lane * LRF_RPB], ur, gt, lo, hi);
                else gs_row_generic(R, &L.a_s[lane * LRF_RPB], ur, gt, pd.native_t2_u != 0, lo, hi);
#else
                for (int r = 0; r < R; r++) ur[r] = fminf(fmaxf(rintf(L.a_s[lane * LRF_RPB + r] * 1e-4f), lo), hi);
#endif
                for (int r = R; r < LRF_RPB; r++) ur[r] = 0.f;
            } else {
                for (int r = 0; r < LRF_RPB; r++) ur[r] = 0.f;
            }
        }
        __syncthreads();
        { // int8 U out, coalesced
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
            for (int e = tid; e < lim; e += 256) {
                const int row = e / R, r = e - row * R;
                Ub[(long)r0 * R + e] = (int8_t)L.u_s[row * LRF_RPB + r];
            }
        }
        { // X^T U for columns 16*wave..+15 (all four rank tiles); U^T U tile row `wave`
            const float* xc = &L.Xs[lq * XS_LD + 16 * wave + li];
#ifdef LRF_BIG_NO_P
            for (int s = 0; s < 0; s++) {
#else
            for (int s = 0; s < 16; s++) {
#endif
                float px = xc[4 * s * XS_LD];
                const float* urow = &L.u_s[(4 * s + lq) * LRF_RPB];
                float qa = urow[16 * wave + li];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    float ub = urow[16 * nt + li];
                    accP[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, ub, accP[nt], 0, 0, 0);
                    accQ[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, ub, accQ[nt], 0, 0, 0);
                }
            }
        }
    }
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RPB;
    float* Qp = Qpart + slot * LRF_RPB * LRF_RPB;
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            Pp[(16 * wave + 4 * lq + reg) * LRF_RPB + 16 * nt + li] = accP[nt][reg]; // D[i = X column][j = r]
            Qp[(16 * wave + 4 * lq + reg) * LRF_RPB + 16 * nt + li] = accQ[nt][reg]; // D[i = r (tile row wave)][j = r']
        }
}

struct BigVLds {
    float a_s[64 * LRF_RPB];
    float v_s[64 * LRF_RPB];
    float gt_s[LRF_GTB_STRIDE];
};

__global__ __launch_bounds__(256) void k_vupdate_big(const PlaneDesc* __restrict__ planes, const float* __restrict__ Ppart,
                                                     const float* __restrict__ Qpart, float* __restrict__ Vf,
                                                     float* __restrict__ Bf, int8_t* __restrict__ V8, float lo, float hi,
                                                     int write_i8)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BigVLds& L = *reinterpret_cast<BigVLds*>(smem);
    const PlaneDesc pd = planes[blockIdx.x];
    const int R = pd.R, tid = threadIdx.x;
    for (int i = tid; i < 64 * LRF_RPB; i += 256) {
        const float* Pp = Ppart + (long)pd.blk0 * 64 * LRF_RPB + i;
        const float* Qp = Qpart + (long)pd.blk0 * LRF_RPB * LRF_RPB + i;
        float acc = 0.f, q = 0.f;
        for (int b0 = 0; b0 < pd.nblk; b0 += 8) {
            float v[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                v[k] = (b0 + k < pd.nblk) ? Pp[(long)(b0 + k) * 64 * LRF_RPB] : 0.f;
                w[k] = (b0 + k < pd.nblk) ? Qp[(long)(b0 + k) * LRF_RPB * LRF_RPB] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (b0 + k < pd.nblk) {
                    acc = (b0 + k == 0) ? v[k] : acc + v[k];
                    q = (b0 + k == 0) ? w[k] : q + w[k];
                }
        }
        L.a_s[i] = acc;
        L.v_s[i] = Vf[(long)blockIdx.x * 64 * LRF_RPB + i];
        int j = i >> 6, r = i & 63; // b' = U^T U entry (j, r)
        if (j < R && r < R) {
            if (j == r) L.gt_s[r * LRF_GTB_LD + LRF_GTB_DEN] = (q + 0.f) + LRF_EPS;
            else L.gt_s[r * LRF_GTB_LD + (j < r ? j : j - 1)] = q;
        }
    }
    __syncthreads();
    if (tid < 64) {
        bool native = (long)(R - 1) * 64 < 400;
        gs_row_generic(R, &L.a_s[tid * LRF_RPB], &L.v_s[tid * LRF_RPB], L.gt_s, native, lo, hi);
        float* Vp = Vf + (long)blockIdx.x * 64 * LRF_RPB + tid * LRF_RPB;
        for (int r = 0; r < R; r++) Vp[r] = L.v_s[tid * LRF_RPB + r];
        if (write_i8) {
            int8_t* vo = V8 + pd.v_off + (long)tid * R;
            for (int r = 0; r < R; r++) vo[r] = (int8_t)L.v_s[tid * LRF_RPB + r];
        }
    }
    __syncthreads();
    if (!write_i8) make_gtable_big(L.v_s, 64, R, Bf + (long)blockIdx.x * LRF_GTB_STRIDE, tid, 256);
}
