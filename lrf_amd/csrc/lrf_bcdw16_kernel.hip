#pragma once
// lrf_bcdw16_kernel.hip — k_bcd_w16: the BCD half-iteration (U update + partials of the V update) of iterations >= 2 for
// ranks up to 16 with one *wave* per (matrix, 384-row block) and no workgroup barrier — k_bcd_w's shape for the rank
// family 9..16 (lrf/factorization/qmf.py:93-126, 128-139).  Included by lrf_api.hip after lrf_bcdw_kernel.hip.
//
// What differs from k_bcd_w (ranks <= 8):
//   * a = x V runs on the matrix cores: at ranks above 8 the 16-wide f32 MFMA tile is mostly used, and 64 fmas per row and
//     rank column on the VALU (1024 at R = 16) would make the vector ALU the bound.  a^T = V^T X^T: V is the A operand,
//     resident in 16 VGPRs; the B operand X[16T + li][4s + lq] is read from the wave's LDS tile with one ds_read_b32 per
//     MFMA (conflict-free in the tile's XOR swizzle); four independent chains (one per 16 rows) of 16 MFMAs, each the
//     k-ordered fma chain of the reference's sgemm.  The four D tiles become lane = row with 16 v_permlane swaps.
//   * the Gauss-Seidel is the EXACT-INTEGER form (gsx_s / gsx_p, lrf_kernels.hip): from the second iteration on u and b
//     are integers and every partial sum of `uu @ bb` stays below 2^24 for the bounds this kernel is launched with (host
//     check (R-1) 64 mx^3 < 2^24), so the order of that sum is immaterial and the reference's dependent chain per column
//     becomes R (R-1) independent fmas; the symmetric b table sits in 17 VGPRs behind DPP row_newbcast.  Bit-identical.
//   * the new row is kept as int8 in LDS (1 KB per wave; the packed dwords are what goes to global memory anyway) and
//     widened by the reads of the MFMA operand (ds_read_i8 + v_cvt): 17 KB of LDS per wave, two waves per SIMD.
//   * b' = u^T u is a full 16 x 16 tile (16 MFMAs per sub-tile).
// MODE 0: iterations >= 2 (old U from int8, exact-integer Gauss-Seidel: launched only for bounds inside the exact range).
// MODE 1: the first iteration — old U = X @ W0 computed here by a second set of MFMA chains (float values, so the
// Gauss-Seidel is the reference's ordered chain, gs16_* below: `uu @ bb` in MKL's single-column order with the table
// operands broadcast from the same 17 VGPRs); any bounds.  Runs with a plane small enough for ATen's native order
// ((R-1) M < 400) and MODE 2 (caller's fp32 U0) stay on the workgroup kernel k_bcd<., 16>.

#define LRF_BCDW16_WAVES 4
#define LRF_BCDW16_WAVE_LDS (64 * 64 * 4 + 64 * 16)
#define LRF_BCDW16_LDS (LRF_BCDW16_WAVES * LRF_BCDW16_WAVE_LDS)

// acc[T][i] (lane (li, lq)) = a[16T + li][4lq + i]  ->  out[4j + i] (lane L) = a[L][4j + i]
__device__ __forceinline__ void w16_tiles_to_rows(const f32x4 (&acc)[4], float (&out)[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned t0 = __float_as_uint(acc[0][i]), t1 = __float_as_uint(acc[1][i]);
        unsigned t2 = __float_as_uint(acc[2][i]), t3 = __float_as_uint(acc[3][i]);
        auto s01 = __builtin_amdgcn_permlane16_swap(t0, t1, false, false);
        auto s23 = __builtin_amdgcn_permlane16_swap(t2, t3, false, false);
        auto s02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
        out[i] = __uint_as_float(s02[0]);
        out[4 + i] = __uint_as_float(s13[0]);
        out[8 + i] = __uint_as_float(s02[1]);
        out[12 + i] = __uint_as_float(s13[1]);
    }
}

// The old int8 row of a lane, R bytes from `up` (any alignment), requested one sub-tile ahead and finished (w16_row_dwords) when
// it is used: dword d of the finished row = bytes 4d .. 4d+3 (bytes at or past R: unspecified — the solve reads R of them).
//   MemLaunch: ceil(R/4) unaligned dword loads from offsets min(4d, R-4) (the last one overlapping its predecessor; R < 4: byte
//              loads, a dword would leave the row's allocation), the partial last dword shifted into place when finished;
//   MemSc1   : the ALIGNED dwords that cover the row as compiler-tracked sc1 loads (k_bcd_p's scheme: a hand-issued asm load
//              would leave its result register open to compiler copies before the data has landed), funnel-shifted
//              (v_alignbyte) when finished; the last dword is clamped to the one that holds the row's last byte.
struct W16RawRow {
    unsigned d[5];
    unsigned sh;
};
template <int R, class MEM>
__device__ __forceinline__ void w16_load_row(const int8_t* up, W16RawRow& w)
{
    if constexpr (MEM::kSc1) {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(up);
        const uintptr_t base = a0 & ~(uintptr_t)3, last = (a0 + R - 1) & ~(uintptr_t)3;
        w.sh = (unsigned)(a0 & 3);
        constexpr int ND = (R + 3) / 4 + 1; // aligned dwords that can hold R bytes at any offset
#pragma unroll
        for (int d = 0; d < 5; d++)
            if (d < ND) {
                const uintptr_t a = base + 4 * d <= last ? base + 4 * d : last;
                w.d[d] = MEM::ld_u32(reinterpret_cast<const unsigned*>(a));
            }
    } else if constexpr (R < 4) {
        unsigned v = (uint8_t)up[0];
        if constexpr (R > 1) v |= (unsigned)(uint8_t)up[R > 1 ? 1 : 0] << 8;
        if constexpr (R > 2) v |= (unsigned)(uint8_t)up[R > 2 ? 2 : 0] << 16;
        w.d[0] = v;
    } else {
        w.d[0] = *reinterpret_cast<const u32_unaligned*>(up);
        if constexpr (R > 4) w.d[1] = *reinterpret_cast<const u32_unaligned*>(up + (R >= 8 ? 4 : R - 4));
        if constexpr (R > 8) w.d[2] = *reinterpret_cast<const u32_unaligned*>(up + (R >= 12 ? 8 : R - 4));
        if constexpr (R > 12) w.d[3] = *reinterpret_cast<const u32_unaligned*>(up + (R >= 16 ? 12 : R - 4));
    }
}
template <int R, class MEM>
__device__ __forceinline__ uint4 w16_row_dwords(const W16RawRow& w)
{
    unsigned o[4] = {0u, 0u, 0u, 0u};
    if constexpr (MEM::kSc1) {
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (4 * d < R) o[d] = __builtin_amdgcn_alignbyte(w.d[d + 1 < (R + 3) / 4 + 1 ? d + 1 : d], w.d[d], w.sh);
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (4 * d < R) o[d] = w.d[d];
        // the partial last dword was loaded from offset R - 4, so byte 4d of the row sits at index 4 - (R & 3) of it
        if constexpr (R >= 4 && (R & 3) != 0) o[R / 4] >>= 8 * (4 - (R & 3));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
template <int R, class MEM>
__device__ __forceinline__ void w16_store_row(int8_t* uo, const uint4 w)
{
    if constexpr (R < 4) {
        MEM::st_u8(uo, (int8_t)w.x);
        if constexpr (R > 1) MEM::st_u8(uo + 1, (int8_t)(w.x >> 8));
        if constexpr (R > 2) MEM::st_u8(uo + 2, (int8_t)(w.x >> 16));
    } else {
        MEM::st_u32(uo, w.x);
        if constexpr (R >= 8) MEM::st_u32(uo + 4, w.y);
        if constexpr (R >= 12) MEM::st_u32(uo + 8, w.z);
        if constexpr (R >= 16) MEM::st_u32(uo + 12, w.w);
        if constexpr ((R & 3) != 0) { // bytes R-4 .. R-1: the tail of the last full dword and the head of the partial one
            const unsigned lo = R / 4 == 1 ? w.x : (R / 4 == 2 ? w.y : w.z), hi = R / 4 == 1 ? w.y : (R / 4 == 2 ? w.z : w.w);
            MEM::st_u32(uo + R - 4, __builtin_amdgcn_alignbyte(hi, lo, R & 3));
        }
    }
}

// One row: old int8 row (w16_row_dwords) and a = x V -> new int8 row (bytes past R zero).  Every lane of the wave must be
// active (the table operands are DPP broadcasts out of other lanes' registers).
template <int R>
__device__ __forceinline__ uint4 w16_row(const float (&a16)[16], const uint4 wo, const float (&tabv)[17], const GsParams& gp)
{
    float a[R], u0[R], u[R], T[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned word = (r >> 2) == 0 ? wo.x : ((r >> 2) == 1 ? wo.y : ((r >> 2) == 2 ? wo.z : wo.w));
        a[r] = a16[r];
        u0[r] = (float)(int)(int8_t)(word >> (8 * (r & 3)));
        u[r] = u0[r];
        T[r] = 0.f;
    }
    gsx_s<R, 1>(T, tabv, u0);
    const float rdenv = tabv[16]; // lane l: 1 / den[l & 15]
    if (__any(gsx_p<R, 0, true>(T, tabv, rdenv, a, u, gp))) { // rare: repeat with the reference's IEEE division
#pragma unroll
        for (int r = 0; r < R; r++) T[r] = 0.f;
        gsx_s<R, 1>(T, tabv, u0);
        gsx_p<R, 0, false>(T, tabv, rdenv, a, u, gp);
    }
    unsigned o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < R; r++) o[r >> 2] |= ((unsigned)(int)u[r] & 0xffu) << (8 * (r & 3));
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- the reference's ordered Gauss-Seidel (first iteration: float old U) on the symmetric table in VGPRs ----------
// tabv[j], lane l = b[j][l & 15] (diagonal: den), so b[j][RR] is lane RR of tabv[j]: a DPP row_newbcast operand.
template <int RR>
__device__ __forceinline__ float gs16_mul(float tab, float x)
{
    float out;
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab), "v"(x), "n"(RR));
    return out;
}
// column index of the k-th "other" column of column RR (increasing order, RR skipped)
template <int RR>
__device__ __forceinline__ constexpr int gs16_col(int k) { return k < RR ? k : k + 1; }
// uu . bb for column RR in the order of oracle/lrf_oracle.c dot_mkl_n1 (K = R - 1 >= 2 terms):
// ((fma(u1, b1, u0 b0) + p_last_odd + ... + p3) + (p2 + p4 + ...)); K == 1: the product; K == 0: nothing
template <int R, int RR>
__device__ __forceinline__ float gs16_term2(const float (&u)[R], const float (&tabv)[17])
{
    constexpr int K = R - 1;
    if constexpr (K == 0) return 0.f;
    else if constexpr (K == 1) return gs16_mul<RR>(tabv[gs16_col<RR>(0)], u[gs16_col<RR>(0)]);
    else {
        float odd = gs16_mul<RR>(tabv[gs16_col<RR>(0)], u[gs16_col<RR>(0)]);
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
            : "+v"(odd) : "v"(tabv[gs16_col<RR>(1)]), "v"(u[gs16_col<RR>(1)]), "n"(RR));
        constexpr int last_odd = ((K - 1) & 1) ? K - 1 : K - 2;
#pragma unroll
        for (int k = last_odd; k >= 3; k -= 2) odd = odd + gs16_mul<RR>(tabv[gs16_col<RR>(k)], u[gs16_col<RR>(k)]);
        if constexpr (K < 3) return odd;
        else {
            float even = gs16_mul<RR>(tabv[gs16_col<RR>(2)], u[gs16_col<RR>(2)]);
#pragma unroll
            for (int k = 4; k < K; k += 2) even = even + gs16_mul<RR>(tabv[gs16_col<RR>(k)], u[gs16_col<RR>(k)]);
            return odd + even;
        }
    }
}
template <int R, int RR, bool EXACT>
__device__ __forceinline__ bool gs16_cols(const float (&a)[R], float (&u)[R], const float (&tabv)[17], const GsParams& gp)
{
    if constexpr (RR < R) {
        const float num = (a[RR] - gs16_term2<R, RR>(u, tabv)) + LRF_EPS;
        float val;
        bool unsafe = false;
        if (EXACT) {
            val = rintf(num / get_bc16<RR>(tabv[RR]));
        } else {
            const float q = gs16_mul<RR>(tabv[16], num);
            const float nq = rintf(q);
            const bool inside = fabsf(q) < gp.flimit;
            unsafe = inside && !(fabsf(q - nq) <= gp.fthr);
            val = inside ? nq : q;
        }
        u[RR] = fminf(fmaxf(val, gp.lo), gp.hi);
        return gs16_cols<R, RR + 1, EXACT>(a, u, tabv, gp) || unsafe;
    } else {
        return false;
    }
}
// first iteration of one row: a = x V, old u = x W0 (floats) -> new int8 row
template <int R>
__device__ __forceinline__ uint4 w16_row_first(const float (&a16)[16], const float (&u16)[16], const float (&tabv)[17], const GsParams& gp)
{
    float a[R], u[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        a[r] = a16[r];
        u[r] = u16[r];
    }
    if (__any(gs16_cols<R, 0, false>(a, u, tabv, gp))) { // rare: repeat with the reference's IEEE division
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = u16[r];
        gs16_cols<R, 0, true>(a, u, tabv, gp);
    }
    unsigned o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < R; r++) o[r >> 2] |= ((unsigned)(int)u[r] & 0xffu) << (8 * (r & 3));
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// One (matrix, 384-row block) on one wave.  Xs: the wave's LRF_BCDW16_WAVE_LDS bytes of LDS (X tile, then the int8 u tile).
// MEM (lrf_device.h): MemLaunch for the launch-per-iteration kernel below, MemSc1 inside the persistent kernel (lrf_bcdp_kernel.hip),
// where the V and b tables, the old int8 rows and the partial tables are handed from wave to wave within the launch.
template <int MODE, class MEM>
__device__ __forceinline__ void w16_block(const float* __restrict__ X, const PlaneDesc& pd, const BlockDesc& bd, const float* __restrict__ Vf,
                                          const float* __restrict__ Wf, const float* __restrict__ Bf, int8_t* __restrict__ U,
                                          float* __restrict__ Ppart, float* __restrict__ Qpart, const GsParams& gp, float* Xs, const int lane)
{
    int8_t* us8 = reinterpret_cast<int8_t*>(Xs + 64 * 64);
    const int R = pd.R;
    const int li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RP;
    const float* gt = Bf + (long)bd.plane * LRF_GT_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;

    // A operand of a^T = V^T X^T, resident: va[s] = V[4s + lq][li] (columns >= R of the table are zero)
    float va[16], wa[MODE == 1 ? 16 : 1];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        va[s] = MEM::ld(Vp + (4 * s + lq) * LRF_RP + li);
        if (MODE == 1) wa[s] = Wf[(long)bd.plane * 64 * LRF_RP + (4 * s + lq) * LRF_RP + li];
    }
    // the symmetric b table of the exact Gauss-Seidel: tabv[j], lane l = b[j][l & 15] (diagonal: den); [16]: 1 / den
    float tabv[17];
#pragma unroll
    for (int j = 0; j < 16; j++)
        tabv[j] = (j < R && li < R) ? ((j == li) ? MEM::ld(gt + li * LRF_GT_LD + LRF_GT_DEN) : MEM::ld(gt + li * LRF_GT_LD + (j < li ? j : j - 1))) : 0.f;
    tabv[16] = (li < R) ? MEM::ld(gt + li * LRF_GT_LD + LRF_GT_RDEN) : 0.f;

    // prefetch registers: xq[T][q] = X[r0 + 16T + 4q + lq][4li .. +3] (each load instruction: four whole rows, 1 KB);
    // upre = the old int8 row of this lane (w16_load_row).  Rows past the end of the block are clamped to its last row.
    f32x4 xq[4][4];
    W16RawRow upre;
#pragma unroll
    for (int d = 0; d < 5; d++) upre.d[d] = 0u;
    upre.sh = 0u;
    auto issue_x = [&](int t, int T0, int T1) {
#ifdef LRF_W16_NO_X // ablation (timing only): the X tile is loaded once per block
        if (t > 0) return;
#endif
        const int r0 = t * 64;
#pragma unroll
        for (int T = T0; T < T1; T++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int row = r0 + 16 * T + 4 * q + lq;
                row = row < nrows ? row : nrows - 1;
                xq[T][q] = *reinterpret_cast<const f32x4*>(Xp + (long)row * 64 + 4 * li);
            }
        }
    };
    auto issue_u = [&](int t) {
        if (MODE != 0) return;
#ifdef LRF_W16_NO_ULOAD // ablation (timing only)
        if (t > 0) return;
#endif
        int row = t * 64 + lane;
        row = row < nrows ? row : nrows - 1;
        const int8_t* up = Ub + (long)row * R;
        switch (R) {
#define LRF_CASE(r) case r: w16_load_row<r, MEM>(up, upre); break;
            LRF_CASE(1) LRF_CASE(2) LRF_CASE(3) LRF_CASE(4) LRF_CASE(5) LRF_CASE(6) LRF_CASE(7) LRF_CASE(8)
            LRF_CASE(9) LRF_CASE(10) LRF_CASE(11) LRF_CASE(12) LRF_CASE(13) LRF_CASE(14) LRF_CASE(15)
#undef LRF_CASE
        default: w16_load_row<16, MEM>(up, upre); break;
        }
    };
    // B operand of a^T: X[16T + li][4s + lq] lives at byte (16T + li) * 256 + ((16 s) ^ (16 li)) + 4 lq of the tile
    const char* xrow_b = reinterpret_cast<const char*>(Xs) + li * 256 + 4 * lq;
    const int g16 = 16 * li;
    // A operand of a' = X^T u, all four column tiles at once: chunk li of row 4s + lq (k_bcd_w)
    const float* xp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xp[e] = &Xs[lq * 64 + 4 * (li ^ (4 * e + lq))];
    // B operand of a' = X^T u and both operands of b' = u^T u: u[4s + lq][li], a byte of the int8 tile
    const int8_t* ub8 = us8 + lq * 16 + li;

    f32x4 accP[4], accQ = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) accP[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_x(0, 0, 4);
    issue_u(0);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        __builtin_amdgcn_sched_barrier(0);
        // ---- 1. sub-tile -> LDS, next sub-tile's loads into the same registers (three bursts, as in k_bcd_w)
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int m = 16 * T + 4 * q + lq;
                *reinterpret_cast<f32x4*>(&Xs[m * 64 + 4 * (li ^ (4 * q + lq))]) = xq[T][q];
            }
        const W16RawRow wraw = upre;
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        const int tn = t + 1;
        const bool more = tn < nsub;
        if (more) {
            issue_x(tn, 0, 2);
            issue_u(tn);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        // ---- 2. a^T = V^T X^T: four independent chains of 16 MFMAs
        float a[16], uf[MODE == 1 ? 16 : 1];
        {
            f32x4 acc[4], accw[MODE == 1 ? 4 : 1];
#pragma unroll
            for (int T = 0; T < 4; T++) {
                acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (MODE == 1) accw[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int h = 0; h < 2; h++) { // the operand reads in two halves of 32 registers
                float bx[8][4];
#pragma unroll
                for (int s = 0; s < 8; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
                        bx[s][T] = *reinterpret_cast<const float*>(xrow_b + T * 16 * 256 + ((16 * (8 * h + s)) ^ g16));
#ifndef LRF_W16_NO_A // ablation (timing only)
#pragma unroll
                for (int s = 0; s < 8; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++) {
                        acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[8 * h + s], bx[s][T], acc[T], 0, 0, 0);
                        if constexpr (MODE == 1) accw[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[8 * h + s], bx[s][T], accw[T], 0, 0, 0);
                    }
#else
#pragma unroll
                for (int T = 0; T < 4; T++) acc[T][0] += bx[0][T] + bx[7][T] + va[8 * h];
#endif
            }
            w16_tiles_to_rows(acc, a);
            if constexpr (MODE == 1) w16_tiles_to_rows(accw, uf);
        }
        if (more) issue_x(tn, 2, 3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. Gauss-Seidel in registers: w = old int8 row in (MODE 0, exact-integer form), new int8 row out
#ifdef LRF_W16_NO_GS // ablation (timing only)
        w = make_uint4(__float_as_uint(a[0]) ^ wraw.d[0], __float_as_uint(a[5]), __float_as_uint(a[10]) ^ wraw.d[1], __float_as_uint(a[15]));
        if (false)
#endif
        switch (R) {
#define LRF_CASE(r) case r: if constexpr (MODE == 1) w = w16_row_first<r>(a, uf, tabv, gp); else w = w16_row<r>(a, w16_row_dwords<r, MEM>(wraw), tabv, gp); break;
            LRF_CASE(1) LRF_CASE(2) LRF_CASE(3) LRF_CASE(4) LRF_CASE(5) LRF_CASE(6) LRF_CASE(7) LRF_CASE(8)
            LRF_CASE(9) LRF_CASE(10) LRF_CASE(11) LRF_CASE(12) LRF_CASE(13) LRF_CASE(14) LRF_CASE(15)
#undef LRF_CASE
        default: if constexpr (MODE == 1) w = w16_row_first<16>(a, uf, tabv, gp); else w = w16_row<16>(a, w16_row_dwords<16, MEM>(wraw), tabv, gp); break;
        }
        const int row = r0 + lane;
        if (row >= nrows) w = make_uint4(0u, 0u, 0u, 0u);
        if (more) issue_x(tn, 3, 4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4. the int8 row to LDS (operand of the partial products) and to global memory
        *reinterpret_cast<uint4*>(us8 + lane * 16) = w;
#ifdef LRF_W16_NO_USTORE // ablation (timing only)
        if (row < nrows && t == 0) {
#else
        if (row < nrows) {
#endif
            int8_t* uo = Ub + (long)row * R;
            switch (R) {
#define LRF_CASE(r) case r: w16_store_row<r, MEM>(uo, w); break;
                LRF_CASE(1) LRF_CASE(2) LRF_CASE(3) LRF_CASE(4) LRF_CASE(5) LRF_CASE(6) LRF_CASE(7) LRF_CASE(8)
                LRF_CASE(9) LRF_CASE(10) LRF_CASE(11) LRF_CASE(12) LRF_CASE(13) LRF_CASE(14) LRF_CASE(15)
#undef LRF_CASE
            default: w16_store_row<16, MEM>(uo, w); break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
#ifndef LRF_W16_NO_PQ // ablation (timing only)
        // ---- 5. a' += X^T u (four strided column tiles per LDS read), b' += u^T u
        float pu[16];
#pragma unroll
        for (int s = 0; s < 16; s++) pu[s] = (float)(int)ub8[64 * s];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f32x4 px[8];
#pragma unroll
            for (int s = 0; s < 8; s++) px[s] = *reinterpret_cast<const f32x4*>(xp[s & 3] + 256 * (8 * h + s));
#pragma unroll
            for (int s = 0; s < 8; s++) {
#pragma unroll
                for (int c = 0; c < 4; c++) accP[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s][c], pu[8 * h + s], accP[c], 0, 0, 0);
                accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(pu[8 * h + s], pu[8 * h + s], accQ, 0, 0, 0);
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // a' partial: tile c holds the columns 4 i + c: D[i = 4*lq + reg][j = li (r)] -> a'[4 i + c][r]
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RP;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) MEM::st(Pp + (4 * (4 * lq + reg) + c) * LRF_RP + li, accP[c][reg]);
    // b' partial: D[i = 4*lq + reg][j = li], exact integers
    float* Qp = Qpart + slot * LRF_RP * LRF_RP;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) MEM::st(Qp + (4 * lq + reg) * LRF_RP + li, accQ[reg]);
}

// Two waves per SIMD for both modes.  Left to itself the compiler gave MODE 1 (both MFMA operand sets resident) 264 registers, one
// wave per SIMD; bounded to 256 it needs 232 and spills nothing: 256 x 512x768 at (16,8,8) 2.40 -> 2.35 ms (round 5).
template <int MODE>
__global__ __launch_bounds__(64 * LRF_BCDW16_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bcd_w16(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                                                  const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                                                  const float* __restrict__ Wf, const float* __restrict__ Bf, int8_t* __restrict__ U,
                                                                  float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp,
                                                                  int nblocks)
{
    extern __shared__ __attribute__((aligned(16))) float bcdw16_lds[]; // LRF_BCDW16_LDS bytes, per wave: X tile, then int8 u
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * LRF_BCDW16_WAVES + wave;
    if (blk >= nblocks) return; // the waves of a workgroup never synchronise with each other
    float* Xs = reinterpret_cast<float*>(reinterpret_cast<char*>(bcdw16_lds) + wave * LRF_BCDW16_WAVE_LDS);
    const BlockDesc bd = blocks[blk];
    const PlaneDesc pd = planes[bd.plane];
    w16_block<MODE, MemLaunch>(X, pd, bd, Vf, Wf, Bf, U, Ppart, Qpart, gp, Xs, (int)(threadIdx.x & 63));
}
