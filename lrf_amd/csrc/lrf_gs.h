// lrf_gs.h — the Gauss-Seidel column sweep of CoordinateDescent.update_u / update_v (lrf/factorization/qmf.py:108-119) and the
// b = v.mT @ v table (qmf.py:107) as inline device functions shared by the BCD kernels of every rank family: the reference's
// summation orders (`uu @ bb`: ATen's native chain or MKL's single-column tree), the speculative reciprocal with the exact-
// division fallback, round-half-even + clamp (qmf.py:191-195), the exact-integer form of ranks 9..16.
#ifndef LRF_GS_H
#define LRF_GS_H
#include "lrf_device.h"

// Exact-integer Gauss-Seidel of ranks 9..16 (see gs_row_lds) on a table in seventeen VGPRs: tabv[j], lane l = b[j][l & 15]
// (symmetric, the diagonal holds den).  T[r] += u0[J] b[J][r] for r < J;  T[r'] += u_R0 b[R0][r'] for r' > R0.
template <int R, int J, int... Rs>
__device__ __forceinline__ void gsx_s_row(float (&T)[R], float tab, float u, std::integer_sequence<int, Rs...>)
{
    (fmac_bc16<Rs>(T[Rs], tab, u), ...);
}
template <int R, int J>
__device__ __forceinline__ void gsx_s(float (&T)[R], const float (&tabv)[17], const float (&u0)[R])
{
    if constexpr (J < R) {
        gsx_s_row<R, J>(T, tabv[J], u0[J], std::make_integer_sequence<int, J>{});
        gsx_s<R, J + 1>(T, tabv, u0);
    }
}
template <int R, int R0, int... Is>
__device__ __forceinline__ void gsx_p_row(float (&T)[R], float tab, float u, std::integer_sequence<int, Is...>)
{
    (fmac_bc16<R0 + 1 + Is>(T[R0 + 1 + Is], tab, u), ...);
}
// FAST: the quotient as num * (1 / den) with the tie / range test of gs_row (returns "some column was too close to call":
// the caller then repeats the row with the IEEE division); the division sits on the column-to-column dependency chain,
// which is what bounds this solve.
template <int R, int R0, bool FAST>
__device__ __forceinline__ bool gsx_p(float (&T)[R], const float (&tabv)[17], float rdenv, const float (&a)[R], float (&u)[R],
                                      const GsParams& gp)
{
    if constexpr (R0 < R) {
        const float num = (a[R0] - T[R0]) + LRF_EPS;
        float val;
        bool unsafe = false;
        if (FAST) {
            const float q = num * get_bc16<R0>(rdenv);
            const float nq = rintf(q);
            const bool inside = fabsf(q) < gp.flimit;
            unsafe = inside && !(fabsf(q - nq) <= gp.fthr);
            val = inside ? nq : q;
        } else {
            val = rintf(num / get_bc16<R0>(tabv[R0]));
        }
        u[R0] = fminf(fmaxf(val, gp.lo), gp.hi);
        gsx_p_row<R, R0>(T, tabv[R0], u[R0], std::make_integer_sequence<int, R - 1 - R0>{});
        return gsx_p<R, R0 + 1, FAST>(T, tabv, rdenv, a, u, gp) || unsafe;
    } else {
        return false;
    }
}

// keeps the scalar loads of one table row next to their use (hoisted together they overflow the SGPR file)
#define LRF_TABLE_ROW_FENCE() asm volatile("" ::: "memory")


// term2 = uu . bb in the reference's order (qmf.py:115): ATen native chain or the MKL single-column tree
template <int K, bool NATIVE>
__device__ __forceinline__ float gs_term2(const float* uu, const float* bb)
{
    if (K == 0) return 0.f;
    if (NATIVE) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            float p = uu[k] * bb[k];
            acc = acc + p;
        }
        return acc;
    }
    if (K == 1) return uu[0] * bb[0];
    float odd = fmaf(uu[1], bb[1], uu[0] * bb[0]); // oracle/lrf_oracle.c dot_mkl_n1
    constexpr int last_odd = ((K - 1) & 1) ? K - 1 : K - 2;
#pragma unroll
    for (int k = last_odd; k >= 3; k -= 2) odd = odd + uu[k] * bb[k];
    if (K < 3) return odd;
    float even = uu[2] * bb[2];
#pragma unroll
    for (int k = 4; k < K; k += 2) even = even + uu[k] * bb[k];
    return odd + even;
}

// One row, all R columns.  EXACT = false: branch-free speculative solve with q~ = num * (1/den); returns true
// when some column sat too close to a rounding tie (or anything else made the shortcut unsafe) — the caller
// then re-solves the row with EXACT = true (IEEE division), which is what the reference computes.
template <int R, bool NATIVE, bool EXACT>
__device__ __forceinline__ bool gs_row(const float* a, float* u, const float* __restrict__ gt, const GsParams gp)
{
    constexpr int K = R - 1;
    // Preload the whole table in straight-line code: one batch of (scalar or LDS) loads and a single wait.
    float bbv[R][K > 0 ? K : 1], rden[R], den[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int k = 0; k < K; k++) bbv[r][k] = gt[r * LRF_GT_LD + k];
        rden[r] = gt[r * LRF_GT_LD + LRF_GT_RDEN];
        den[r] = gt[r * LRF_GT_LD + LRF_GT_DEN];
    }
#ifdef LRF_STAMPS
    STAMP(gq0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(gq1);
    if (!EXACT) GSP_ADD(1, gq0, gq1);
#endif
    bool unsafe = false;
#pragma unroll
    for (int r = 0; r < R; r++) {
        float uu[K > 0 ? K : 1];
        int n = 0;
#pragma unroll
        for (int j = 0; j < R; j++)
            if (j != r) uu[n++] = u[j];
        float num = (a[r] - gs_term2<K, NATIVE>(uu, bbv[r])) + LRF_EPS;
        float val;
        if (EXACT) {
            val = rintf(num / den[r]);
        } else {
            float q = num * rden[r];
            float nq = rintf(q);
            bool inside = fabsf(q) < gp.flimit;          // false for NaN: falls to val = q, clamp handles it
            unsafe |= inside && !(fabsf(q - nq) <= gp.fthr);
            val = inside ? nq : q;
        }
        u[r] = fminf(fmaxf(val, gp.lo), gp.hi);
    }
#ifdef LRF_STAMPS
    asm volatile("" ::"v"(u[R - 1]));
    STAMP(gq2);
    if (!EXACT) GSP_ADD(2, gq1, gq2);
#endif
    return unsafe;
}

// One row through LDS: a_row[0..R) in, u_row[0..LRF_RP) out (zero padded).  The old row comes from int8 bytes
// (uold_row, packed R per row) when FROM_I8, else from u_row itself.  All loops have compile-time bounds so the
// LDS reads/writes are issued as batches (a runtime-R loop costs one exposed LDS latency per element).
template <int R, bool FROM_I8>
__device__ __forceinline__ void gs_row_lds(const float* a_row, float* u_row, const int8_t* uold_row,
                                           const float* __restrict__ gt, bool native, const GsParams gp, const float (&tabv)[17])
{
    float a[R], u0[R], u[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        a[r] = a_row[r];
        u0[r] = FROM_I8 ? (float)uold_row[r] : u_row[r];
        u[r] = u0[r];
    }
#ifdef LRF_STAMPS
    STAMP(gl0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(gl1);
    GSP_ADD(0, gl0, gl1);
#endif
    if (FROM_I8 && R > 8 && gp.exact_int) {
        // Ranks 9..16 from the second iteration on: integer u, integer b, sums below 2^24 — the reference's dependent chain
        // per column (and the R (R-1)-entry register copy of the table it needs) becomes R (R-1) independent fmas on R
        // accumulators: T[r] starts as the sum over the columns j > r still holding old values and receives u_r b[r][r']
        // for every later column as soon as u_r is known.  Bit-identical; the table sits in sixteen VGPRs (tabv) and is
        // broadcast by DPP (scalar loads of it, even fenced row by row, left one exposed scalar-cache latency per row).
        float T[R];
#pragma unroll
        for (int r = 0; r < R; r++) T[r] = 0.f;
        gsx_s<R, 1>(T, tabv, u0);
        float T0[R];
#pragma unroll
        for (int r = 0; r < R; r++) T0[r] = T[r];
        const float rdenv = tabv[16]; // lane l: 1 / den[l & 15]
        if (__any(gsx_p<R, 0, true>(T, tabv, rdenv, a, u, gp))) { // rare: repeat with the reference's IEEE division
#pragma unroll
            for (int r = 0; r < R; r++) T[r] = T0[r];
            gsx_p<R, 0, false>(T, tabv, rdenv, a, u, gp);
        }
    } else {
        bool unsafe = native ? gs_row<R, true, false>(a, u, gt, gp) : gs_row<R, false, false>(a, u, gt, gp);
        if (__any(unsafe)) { // rare (about one wave in a few hundred): redo with the reference's IEEE division
#pragma unroll
            for (int r = 0; r < R; r++) u[r] = u0[r];
            if (native) gs_row<R, true, true>(a, u, gt, gp);
            else gs_row<R, false, true>(a, u, gt, gp);
        }
    }
#ifdef LRF_STAMPS
    STAMP(gl2);
    GSP_ADD(3, gl1, gl2);
#endif
    float o[LRF_RP];
#pragma unroll
    for (int r = 0; r < LRF_RP; r++) o[r] = (r < R) ? u[r < R ? r : 0] : 0.f;
#pragma unroll
    for (int r = 0; r < LRF_RP; r += 4) *reinterpret_cast<f32x4*>(u_row + r) = (f32x4){o[r], o[r + 1], o[r + 2], o[r + 3]};
}

template <int RMAX, bool FROM_I8>
__device__ __forceinline__ void gs_dispatch(int R, const float* a_row, float* u_row, const int8_t* uold_rows, int lane,
                                            const float* __restrict__ gt, bool native, const GsParams gp, const float (&tabv)[17])
{
    switch (R) {
#define LRF_CASE(r)                                                                                              \
    case r:                                                                                                      \
        if (r <= RMAX) gs_row_lds<(r <= RMAX ? r : 1), FROM_I8>(a_row, u_row, uold_rows + lane * r, gt, native, gp, tabv); \
        break;
        LRF_CASE(1) LRF_CASE(2) LRF_CASE(3) LRF_CASE(4) LRF_CASE(5) LRF_CASE(6) LRF_CASE(7) LRF_CASE(8)
        LRF_CASE(9) LRF_CASE(10) LRF_CASE(11) LRF_CASE(12) LRF_CASE(13) LRF_CASE(14) LRF_CASE(15) LRF_CASE(16)
#undef LRF_CASE
    }
}

// gt table of b = v.mT @ v (R x R) from a [depth][LRF_RP] factor: thread (j, r).
// ATen uses its native kernel when depth*R*R < 400, MKL (k-ordered fma chain) otherwise.
__device__ __forceinline__ void make_gtable(const float* Vp, int depth, int R, float* gt, int tid, int nthreads)
{
    bool native = (long)depth * R * R < 400;
    for (int i = tid; i < R * R; i += nthreads) {
        int j = i / R, r = i - j * R;
        float acc = 0.f;
        if (native) {
            for (int k = 0; k < depth; k++) {
                float p = Vp[k * LRF_RP + j] * Vp[k * LRF_RP + r];
                acc = acc + p;
            }
        } else {
            for (int k = 0; k < depth; k++) acc = fmaf(Vp[k * LRF_RP + j], Vp[k * LRF_RP + r], acc);
        }
        if (j == r) {
            float den = (acc + 0.f) + LRF_EPS;
            gt[r * LRF_GT_LD + LRF_GT_DEN] = den;
            gt[r * LRF_GT_LD + LRF_GT_RDEN] = 1.0f / den;
        } else {
            gt[r * LRF_GT_LD + (j < r ? j : j - 1)] = acc;
        }
    }
}

// ---- ranks above 16 (rank pitch LRF_RPB, gt pitch LRF_GTB_LD) and the generic ordered chain ------------------------------
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

// The ordered Gauss-Seidel of one row (qmf.py:108-119) with the row in REGISTERS:
// u[0..32) in/out, a[0..32), the b table read from LDS with wave-uniform addresses (gt_l: pitch LRF_GTB_LD, row r = the `bb`
// vector of column r, [LRF_GTB_DEN] = den).  Every index below is a compile-time constant, so the R (R-1) terms are register
// multiplies and adds in the reference's MKL single-column order (oracle dot_mkl_n1: the odd terms descending after
// fma(u1, b1, u0 b0), then the even ones ascending) with no memory latency inside the chains; the run-time rank only
// guards terms (wave-uniform branches).  Not for the ATen-native order (tiny matrices): callers keep gs_term2_generic there.
template <int RR>
__device__ __forceinline__ float mid_term2(int K, const float (&u)[32], const float (&bb)[32])
{
    // uu[n] = u[n < RR ? n : n + 1]
#define MID_UU(n) u[(n) < RR ? (n) : ((n) + 1 < 32 ? (n) + 1 : 31)]
    if (K <= 0) return 0.f;
    if (K == 1) return MID_UU(0) * bb[0];
    float odd = fmaf(MID_UU(1), bb[1], MID_UU(0) * bb[0]);
#pragma unroll
    for (int n = 29; n >= 3; n -= 2)
        if (n < K) odd = odd + MID_UU(n) * bb[n];
    if (K < 3) return odd;
    float even = MID_UU(2) * bb[2];
#pragma unroll
    for (int n = 4; n <= 30; n += 2)
        if (n < K) even = even + MID_UU(n) * bb[n];
#undef MID_UU
    return odd + even;
}

template <int RR>
__device__ __forceinline__ void mid_ordered_col(int R, const float (&a)[32], float (&u)[32], const float* __restrict__ gt_l, float lo, float hi)
{
    if (RR < R) {
        // the whole `bb` row up front (eight wave-uniform 16-byte reads, pitch LRF_GTB_LD * 4 = 272 bytes): a read inside each
        // guarded term would put an LDS round trip into every link of the chain
        float bb[32];
#pragma unroll
        for (int n = 0; n < 32; n += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gt_l + RR * LRF_GTB_LD + n);
            bb[n] = v[0]; bb[n + 1] = v[1]; bb[n + 2] = v[2]; bb[n + 3] = v[3];
        }
        const float den = gt_l[RR * LRF_GTB_LD + LRF_GTB_DEN];
        const float term2 = mid_term2<RR>(R - 1, u, bb);
        const float num = (a[RR] - term2) + LRF_EPS;
        const float val = rintf(num / den);
        u[RR] = fminf(fmaxf(val, lo), hi);
    }
}

template <int... Rs>
__device__ __forceinline__ void mid_ordered_row(int R, const float (&a)[32], float (&u)[32], const float* __restrict__ gt_l, float lo, float hi,
                                                std::integer_sequence<int, Rs...>)
{
    (mid_ordered_col<Rs>(R, a, u, gt_l, lo, hi), ...);
}

#endif
