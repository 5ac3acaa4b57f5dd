#pragma once
// lrf_bcdw_kernel.hip — k_bcd_w: the BCD half-iteration (U update + partials of the V update) for ranks <= 8 with
// one *wave* per (matrix, 384-row block) and no workgroup barrier at all.  Included by lrf_api.hip after lrf_kernels.hip.
//
// Per 64-row sub-tile the wave
//   1. stores the sub-tile (prefetched one sub-tile ahead into registers: 16 float4 loads of four whole rows each, issued
//      in three bursts) to its private, XOR-swizzled LDS tile and reads it back with lane = row (16 ds_read_b128,
//      conflict-free);
//   2. computes a = x V on the VALU: one fma per (k, column), x[k] from the lane's own row and V[k][r] broadcast out of
//      32 resident VGPRs by the DPP row_newbcast operand modifier (V is wave-uniform and constant for the block).
//      With R <= 8 the 16-wide f32 MFMA tile would be at most half used, the VALU has the same f32 peak, and the
//      result is already lane = row — the same k-ordered fma chain as the MFMA, bit for bit;
//   3. solves the Gauss-Seidel recurrence in registers (gs_row_tab below: gs_row of lrf_kernels.hip with the b table
//      broadcast from VGPRs the same way);
//   4. writes the int8 row (two overlapping unaligned dword stores), parks u in LDS and accumulates a' += X^T u (MFMA,
//      four independent chains over four strided 16-column tiles {4i + c}, so that one ds_read_b128 of the LDS tile
//      feeds all four) and b' += u^T u (MFMA, two row groups per instruction in the two diagonal 8 x 8 blocks).
// The summation orders are those of k_bcd (and therefore of the reference): k-ordered fma chains, one chain per
// 384-row block, block partials added in order by k_vupdate.
// Two waves per SIMD (18 KB LDS per wave): while one wave is in its VALU phases the other feeds the MFMA pipe; the
// next sub-tile's global loads are in flight during steps 2 to 4.


// diagnostic stamps of this kernel: -DLRF_STAMPS -DLRF_W_STAMPS (tools/dev_stamps_w.py)
#if defined(LRF_STAMPS) && defined(LRF_W_STAMPS)
#define WSTAMP(var) STAMP(var)
#define WSTAMP_ADD(acc, a, b) STAMP_ADD(acc, a, b)
#else
#undef LRF_W_STAMPS
#define WSTAMP(var)
#define WSTAMP_ADD(acc, a, b)
#endif

// LDS X tile: element (m, n) of the 64 x 64 sub-tile lives at dword m*64 + 4*((n >> 2) ^ xsw(m)) + (n & 3).
// All three access patterns are float4 and conflict-free with xsw(m) = m mod 16: the stores (8 consecutive rows, one
// chunk), the row reads of the U phase (16 rows, one chunk) and the reads of the a' = X^T u operand (lane (li, lq):
// chunk li of row 4s + lq — the four components are the operands of four *strided* 16-column tiles {4 li + c}).
__device__ __forceinline__ int xsw(int m) { return m & 15; }

// V (wave-uniform, constant for the whole block) lives in 32 VGPRs: vreg[r][kb], lane l = V[16 kb + (l & 15)][r].
// DPP row_newbcast:n hands lane n of every 16-lane row to all lanes of that row, folded into the fma's operand
// fetch, so V[k][r] reaches the VALU as a uniform operand without scalar loads or LDS traffic.
// acc = fma(lane KL of v's 16-lane row, x, acc) in one instruction (the compiler does not fold a DPP move into an fma)
template <int KL>
__device__ __forceinline__ void fmac_row_bcast(float& acc, float v, float x)
{
#ifndef LRF_W_NODPP
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(x), "n"(KL));
#else
    asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(v), "v"(x)); // ablation: wrong values, same instruction count
#endif
}

// acc[r] = fma(V[k][r], x[k], acc[r]) for k = 0..63 in order: the k-ordered chain of the reference's sgemm
template <int NR, int J>
__device__ __forceinline__ void row_times_v_step(const f32x4 xv, const float (&vreg)[8][4], float (&acc)[8])
{
#pragma unroll
    for (int r = 0; r < NR; r++) fmac_row_bcast<(4 * J + 0) & 15>(acc[r], vreg[r][J >> 2], xv[0]);
#pragma unroll
    for (int r = 0; r < NR; r++) fmac_row_bcast<(4 * J + 1) & 15>(acc[r], vreg[r][J >> 2], xv[1]);
#pragma unroll
    for (int r = 0; r < NR; r++) fmac_row_bcast<(4 * J + 2) & 15>(acc[r], vreg[r][J >> 2], xv[2]);
#pragma unroll
    for (int r = 0; r < NR; r++) fmac_row_bcast<(4 * J + 3) & 15>(acc[r], vreg[r][J >> 2], xv[3]);
}
template <int NR>
__device__ __forceinline__ void row_times_v(const float* __restrict__ xrow_lds, int g16, const float (&vreg)[8][4], float (&acc)[8])
{
#define LRF_STEP(J)                                                                                                   \
    row_times_v_step<NR, J>(*reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xrow_lds) + ((16 * J) ^ g16)), vreg, acc);
    LRF_STEP(0) LRF_STEP(1) LRF_STEP(2) LRF_STEP(3) LRF_STEP(4) LRF_STEP(5) LRF_STEP(6) LRF_STEP(7)
    LRF_STEP(8) LRF_STEP(9) LRF_STEP(10) LRF_STEP(11) LRF_STEP(12) LRF_STEP(13) LRF_STEP(14) LRF_STEP(15)
#undef LRF_STEP
}
__device__ __forceinline__ void row_times_v_dispatch(int R, const float* xrow_lds, int g16, const float (&vreg)[8][4], float (&acc)[8])
{
    switch (R) {
    case 1: row_times_v<1>(xrow_lds, g16, vreg, acc); break;
    case 2: row_times_v<2>(xrow_lds, g16, vreg, acc); break;
    case 3: row_times_v<3>(xrow_lds, g16, vreg, acc); break;
    case 4: row_times_v<4>(xrow_lds, g16, vreg, acc); break;
    case 5: row_times_v<5>(xrow_lds, g16, vreg, acc); break;
    case 6: row_times_v<6>(xrow_lds, g16, vreg, acc); break;
    case 7: row_times_v<7>(xrow_lds, g16, vreg, acc); break;
    default: row_times_v<8>(xrow_lds, g16, vreg, acc); break;
    }
}

// ---- Gauss-Seidel with the b table broadcast out of VGPRs ------------------------------------------------------
// The per-matrix table (gt, lrf_kernels.hip) is wave-uniform.  Scalar loads of it cost a scalar-cache round trip per
// batch inside a dependent chain, so it is kept in five VGPRs instead and reaches the VALU through DPP
// row_newbcast, like V above:  tab[j], lane l = ct[16 j + (l & 15)] with ct[8 r + n] = b[j_n][r] (n < 7),
// ct[8 r + 7] = 1/den[r];  tab[4], lane l = den[l & 7].
template <int IDX>
__device__ __forceinline__ float mul_tab(const float (&tab)[5], float x)
{
    float out;
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab[IDX >> 4]), "v"(x), "n"(IDX & 15));
    return out;
}
template <int IDX>
__device__ __forceinline__ float fma_tab(const float (&tab)[5], float x, float acc)
{
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab[IDX >> 4]), "v"(x), "n"(IDX & 15));
    return acc;
}
template <int IDX>
__device__ __forceinline__ float get_tab(const float (&tab)[5])
{
    float out;
    asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab[IDX >> 4]), "n"(IDX & 15));
    return out;
}

// uu . bb for column RR (the others' values uu[0..K) in increasing column order), same orders as gs_term2
template <int K, int RR, bool NATIVE>
__device__ __forceinline__ float gs_term2_tab(const float* uu, const float (&tab)[5])
{
    if constexpr (K == 0) return 0.f;
    else if constexpr (NATIVE) {
        float acc = 0.f;
        acc = acc + mul_tab<8 * RR + 0>(tab, uu[0]);
        if constexpr (K > 1) acc = acc + mul_tab<8 * RR + (K > 1 ? 1 : 0)>(tab, uu[K > 1 ? 1 : 0]);
        if constexpr (K > 2) acc = acc + mul_tab<8 * RR + (K > 2 ? 2 : 0)>(tab, uu[K > 2 ? 2 : 0]);
        if constexpr (K > 3) acc = acc + mul_tab<8 * RR + (K > 3 ? 3 : 0)>(tab, uu[K > 3 ? 3 : 0]);
        if constexpr (K > 4) acc = acc + mul_tab<8 * RR + (K > 4 ? 4 : 0)>(tab, uu[K > 4 ? 4 : 0]);
        if constexpr (K > 5) acc = acc + mul_tab<8 * RR + (K > 5 ? 5 : 0)>(tab, uu[K > 5 ? 5 : 0]);
        if constexpr (K > 6) acc = acc + mul_tab<8 * RR + (K > 6 ? 6 : 0)>(tab, uu[K > 6 ? 6 : 0]);
        return acc;
    } else if constexpr (K == 1) {
        return mul_tab<8 * RR>(tab, uu[0]);
    } else {
        // oracle/lrf_oracle.c dot_mkl_n1: ((fma(u1,b1,u0*b0) + p5) + p3) + (p2 + p4 [+ p6])
        float odd = fma_tab<8 * RR + 1>(tab, uu[1], mul_tab<8 * RR>(tab, uu[0]));
        if constexpr (K >= 6) odd = odd + mul_tab<8 * RR + 5>(tab, uu[K >= 6 ? 5 : 0]);
        if constexpr (K >= 4) odd = odd + mul_tab<8 * RR + 3>(tab, uu[K >= 4 ? 3 : 0]);
        if constexpr (K < 3) return odd;
        else {
            float even = mul_tab<8 * RR + 2>(tab, uu[2]);
            if constexpr (K >= 5) even = even + mul_tab<8 * RR + 4>(tab, uu[K >= 5 ? 4 : 0]);
            if constexpr (K >= 7) even = even + mul_tab<8 * RR + 6>(tab, uu[K >= 7 ? 6 : 0]);
            return odd + even;
        }
    }
}

template <int R, int RR, bool NATIVE, bool EXACT>
__device__ __forceinline__ void gs_col_tab(const float* a, float* u, const float (&tab)[5], const GsParams gp, bool& unsafe)
{
    constexpr int K = R - 1;
    float uu[K > 0 ? K : 1];
    int n = 0;
#pragma unroll
    for (int j = 0; j < R; j++)
        if (j != RR) uu[n++] = u[j];
    float num = (a[RR] - gs_term2_tab<K, RR, NATIVE>(uu, tab)) + LRF_EPS;
    float val;
    if (EXACT) {
        val = rintf(num / get_tab<64 + RR>(tab));
    } else {
        float q = mul_tab<8 * RR + 7>(tab, num);
        float nq = rintf(q);
        bool inside = fabsf(q) < gp.flimit;
        unsafe |= inside && !(fabsf(q - nq) <= gp.fthr);
        val = inside ? nq : q;
    }
    u[RR] = fminf(fmaxf(val, gp.lo), gp.hi);
}

// one row, all R columns: gs_row (lrf_kernels.hip) with the table operands broadcast from VGPRs
template <int R, bool NATIVE, bool EXACT>
__device__ __forceinline__ bool gs_row_tab(const float* a, float* u, const float (&tab)[5], const GsParams gp)
{
    bool unsafe = false;
    gs_col_tab<R, 0, NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 1) gs_col_tab<R, (R > 1 ? 1 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 2) gs_col_tab<R, (R > 2 ? 2 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 3) gs_col_tab<R, (R > 3 ? 3 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 4) gs_col_tab<R, (R > 4 ? 4 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 5) gs_col_tab<R, (R > 5 ? 5 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 6) gs_col_tab<R, (R > 6 ? 6 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    if constexpr (R > 7) gs_col_tab<R, (R > 7 ? 7 : 0), NATIVE, EXACT>(a, u, tab, gp, unsafe);
    return unsafe;
}

// Gauss-Seidel of one row held in registers: a[0..R), u[0..R) in/out (entries >= R are zeroed)
template <int R, int RMAX>
__device__ __forceinline__ void gs_regs(const float (&a)[RMAX], float (&u)[RMAX], const float (&tab)[5], bool native,
                                        const GsParams gp)
{
    float aa[R], uu[R], u0[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        aa[r] = a[r];
        u0[r] = u[r];
        uu[r] = u[r];
    }
    bool unsafe = native ? gs_row_tab<R, true, false>(aa, uu, tab, gp) : gs_row_tab<R, false, false>(aa, uu, tab, gp);
    if (__any(unsafe)) { // rare: redo with the reference's IEEE division
#pragma unroll
        for (int r = 0; r < R; r++) uu[r] = u0[r];
        if (native) gs_row_tab<R, true, true>(aa, uu, tab, gp);
        else gs_row_tab<R, false, true>(aa, uu, tab, gp);
    }
#pragma unroll
    for (int r = 0; r < RMAX; r++) u[r] = (r < R) ? uu[r < R ? r : 0] : 0.f;
}

template <int RMAX>
__device__ __forceinline__ void gs_regs_dispatch(int R, const float (&a)[RMAX], float (&u)[RMAX], const float (&tab)[5],
                                                 bool native, const GsParams gp)
{
    switch (R) {
#define LRF_CASE(r)                                                                \
    case r:                                                                        \
        if (r <= RMAX) gs_regs<(r <= RMAX ? r : 1), RMAX>(a, u, tab, native, gp);  \
        break;
        LRF_CASE(1) LRF_CASE(2) LRF_CASE(3) LRF_CASE(4) LRF_CASE(5) LRF_CASE(6) LRF_CASE(7) LRF_CASE(8)
#undef LRF_CASE
    }
}

#define LRF_BCDW_LDS (LRF_BCDW_WAVES * (64 * 64 + 64 * 8) * 4)
#define LRF_BCDW_WAVES 4 // waves per workgroup: consecutive blocks, i.e. mostly the same matrix (shared V in the scalar cache)
// aligned(4096): with identical instructions this kernel ran 0.153 ms when its entry was 4 KB aligned (or at 0x100 /
// 0x500 past a 2 KB boundary) and 0.238 ms at 0x800 past a 4 KB boundary, so the placement is pinned.
template <int MODE>
__global__ __launch_bounds__(64 * LRF_BCDW_WAVES) __attribute__((aligned(4096))) void k_bcd_w(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                              const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                              const float* __restrict__ Wf, const float* __restrict__ Bf,
                                              const float* __restrict__ U0, int8_t* __restrict__ U,
                                              float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp, int nblocks)
{
    constexpr int RMAX = 8;
    extern __shared__ __attribute__((aligned(16))) float bcdw_lds[]; // LRF_BCDW_LDS bytes, per wave: X tile, then u

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * LRF_BCDW_WAVES + wave;
    if (blk >= nblocks) return; // the waves of a workgroup never synchronise with each other
    float* Xs = bcdw_lds + wave * (64 * 64 + 64 * RMAX);
    float* us = Xs + 64 * 64;
    const BlockDesc bd = blocks[blk];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RP;
    const float* Wp = Wf + (long)bd.plane * 64 * LRF_RP;
    const float* gt = Bf + (long)bd.plane * LRF_GT_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;
    const bool native = pd.native_t2_u != 0;

    // V (and, first iteration, W0) resident in registers for row_bcast: vreg[r][kb] = V[16 kb + li][r]
    float vreg[8][4], wreg[MODE == 1 ? 8 : 1][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            vreg[r][kb] = Vp[(16 * kb + li) * LRF_RP + r];
            if (MODE == 1) wreg[r][kb] = Wp[(16 * kb + li) * LRF_RP + r];
        }

    // the b table in five registers (gs_row_tab)
    float tab[5];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ci = 16 * j + li, tr = ci >> 3, tn = ci & 7;
        tab[j] = gt[tr * LRF_GT_LD + (tn < 7 ? tn : LRF_GT_RDEN)];
    }
    tab[4] = gt[(li & 7) * LRF_GT_LD + LRF_GT_DEN];

    // prefetch registers: xq[T][q] = X[r0 + 16T + 4q + lq][4li .. +3] (chunk li of the row); upre = old int8 U[r0 + lane][:].
    // Rows past the end of the block are clamped to its last row (finite data, no per-lane branches): their u is forced to 0.
    f32x4 xq[4][4];
    unsigned upre[2]; // the row's R bytes: bytes 0..3 and bytes R-4..R-1 (R >= 4), or bytes 0..R-1 gathered (R < 4)
    auto issue_x = [&](int t, int T0, int T1) { // rows 16 T0 .. 16 T1 - 1 of sub-tile t: each load is four whole rows (1 KB)
        const int r0 = t * 64;
#pragma unroll
        for (int T = T0; T < T1; T++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int row = r0 + 16 * T + 4 * q + lq;
                row = row < nrows ? row : nrows - 1;
#ifndef LRF_W_NO_LOADS
                xq[T][q] = *reinterpret_cast<const f32x4*>(Xp + (long)row * 64 + 4 * li);
#else
                xq[T][q] = (f32x4){(float)lane, 1.f + q, 2.f + T, (float)t}; // ablation: no HBM traffic for X
#endif
            }
        }
    };
    auto issue_u = [&](int t) {
        const int r0 = t * 64;
        if (MODE == 0) {
            int row = r0 + lane;
            row = row < nrows ? row : nrows - 1;
            const int8_t* up = Ub + (long)row * R;
            if (R >= 4) { // two (unaligned, overlapping) dword loads instead of R byte loads
                upre[0] = *reinterpret_cast<const u32_unaligned*>(up);
                upre[1] = *reinterpret_cast<const u32_unaligned*>(up + R - 4);
            } else
            {
                unsigned b0 = (uint8_t)up[0], b1 = (uint8_t)up[R > 1 ? 1 : 0], b2 = (uint8_t)up[R > 2 ? 2 : 0];
                upre[0] = b0 | (b1 << 8) | (b2 << 16);
                upre[1] = 0;
            }
        }
    };

    // A operand of a' = X^T u, all four column tiles at once: chunk li of row 4s + lq, xp[(s & 3)][64 * 4 * s floats]
    // (xsw(4s + lq) = 4 (s & 3) + lq: four lane-constant bases, the rest is an immediate offset)
    const float* xp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xp[e] = &Xs[lq * 64 + 4 * (li ^ (4 * e + lq))];
    // B operand of a' = X^T u: u[4s + lq][li] (columns >= 8 are zero); A/B operand of b' = u^T u: two row groups per
    // MFMA, u[4(2h) + lq][li] in columns/rows 0..7 and u[4(2h+1) + lq][li - 8] in 8..15
    const float* ub = &us[lq * RMAX + (li & 7)];
    const float* uq = &us[(lq + 4 * (li >> 3)) * RMAX + (li & 7)];
    const float* xrow = &Xs[lane * 64];
    const int g16 = 16 * xsw(lane);

    f32x4 accP[4], accQ = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) accP[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef LRF_W_STAMPS
    unsigned long long c_w1 = 0, c_w2 = 0, c_w3 = 0, c_w4 = 0, c_w5 = 0, c_w6 = 0, c_w7 = 0;
#endif
    WSTAMP(t_begin);
    issue_x(0, 0, 4);
    issue_u(0);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        WSTAMP(s0);
#ifdef LRF_W_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s1);
        WSTAMP_ADD(c_w1, s0, s1); // wait for the prefetch
        // ---- 1. sub-tile -> LDS, next sub-tile's loads into the same registers
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int m = 16 * T + 4 * q + lq; // xsw(m) = 4 q + lq: one lane-constant address, the rest immediate
                *reinterpret_cast<f32x4*>(&Xs[m * 64 + 4 * (li ^ (4 * q + lq))]) = xq[T][q];
            }
        float u[RMAX];
        const int row = r0 + lane;
        if constexpr (MODE == 0) {
            // bytes 4..7 of the row sit in upre[1] from byte 8 - R on
            const unsigned lo = upre[0], hi = (R > 4) ? upre[1] >> (8 * (8 - R)) : 0u;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                u[r] = (float)(int)(int8_t)(lo >> (8 * r));
                u[4 + r] = (float)(int)(int8_t)(hi >> (8 * r));
            }
        }
        // The next sub-tile's 16 KB go out in three bursts (here, after the U phase, after the Gauss-Seidel) rather
        // than as one: at ~10 B/cycle/CU the kernel runs at the memory system's pace and a wave that issues sixteen
        // loads into full queues just stalls.  The last sub-tile of a block prefetches nothing.
        const int tn = t + 1;
        const bool more = tn < nsub; // wave-uniform: the last sub-tile prefetches nothing
        if (more) {
            issue_x(tn, 0, 2);
            issue_u(tn);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s2);
        WSTAMP_ADD(c_w2, s1, s2); // LDS stores + prefetch issue
        // ---- 2. a = x V (and, first iteration, u_old = x W0) for this lane's row
        float a[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; r++) a[r] = 0.f;
#ifndef LRF_W_NO_U
        row_times_v_dispatch(R, xrow, g16, vreg, a);
#else
        a[0] = xrow[lane & 3]; a[1] = vreg[1][1];
#endif
        if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = 0.f;
            row_times_v_dispatch(R, xrow, g16, wreg, u);
        }
        if constexpr (MODE == 2) {
            const float* up = U0 + pd.u0_off + ((long)bd.row0 + (row < nrows ? row : nrows - 1)) * R;
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = up[r < R ? r : R - 1];
        }
#ifdef LRF_W_STAMPS
        asm volatile("" ::"v"(a[0]), "v"(a[RMAX - 1]));
#endif
        if (more) issue_x(tn, 2, 3);
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s3);
        WSTAMP_ADD(c_w3, s2, s3); // row reads + packed fma
        // ---- 3. Gauss-Seidel in registers
#ifndef LRF_W_NO_GS
        gs_regs_dispatch<RMAX>(R, a, u, tab, native, gp);
#else
#pragma unroll
        for (int r = 0; r < RMAX; r++) u[r] = (r < R) ? fminf(fmaxf(rintf(a[r] * 1e-4f + u[r]), gp.lo), gp.hi) : 0.f;
#endif
        if (row >= nrows) {
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = 0.f;
        }
#ifdef LRF_W_STAMPS
        asm volatile("" ::"v"(u[0]), "v"(u[RMAX - 1]));
#endif
        if (more) issue_x(tn, 3, 4);
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s5);
        WSTAMP_ADD(c_w5, s3, s5); // Gauss-Seidel
        // ---- 4. u -> LDS (operand of the partial products), int8 row out
#pragma unroll
        for (int r = 0; r < RMAX; r += 4) *reinterpret_cast<f32x4*>(&us[lane * RMAX + r]) = (f32x4){u[r], u[r + 1], u[r + 2], u[r + 3]};
        if (row < nrows) {
            int8_t* uo = Ub + (long)row * R;
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                lo |= ((unsigned)(int)u[r] & 0xffu) << (8 * r);
                hi |= ((unsigned)(int)u[4 + r] & 0xffu) << (8 * r);
            }
            if (R >= 4) { // bytes 0..3 and bytes R-4..R-1 (overlapping, same values): two dword stores
                *reinterpret_cast<u32_unaligned*>(uo) = lo;
                const unsigned long long w = ((unsigned long long)hi << 32) | lo;
                *reinterpret_cast<u32_unaligned*>(uo + R - 4) = (unsigned)(w >> (8 * (R - 4)));
            } else {
                uo[0] = (int8_t)lo;
                if (R > 1) uo[1] = (int8_t)(lo >> 8);
                if (R > 2) uo[2] = (int8_t)(lo >> 16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s6);
        WSTAMP_ADD(c_w6, s5, s6); // u -> LDS, int8 stores
        float pu[16], qu[8];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            float v = ub[4 * s * RMAX];
            pu[s] = (li < RMAX) ? v : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 8; h++) qu[h] = uq[8 * h * RMAX];
#ifdef LRF_W_NO_PQ
        accP[0][0] += pu[3] + qu[2];
#else
        f32x4 px[16];
#pragma unroll
        for (int s = 0; s < 16; s++) px[s] = *reinterpret_cast<const f32x4*>(xp[s & 3] + 256 * s);
#pragma unroll
        for (int s = 0; s < 16; s++) {
#pragma unroll
            for (int c = 0; c < 4; c++) accP[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s][c], pu[s], accP[c], 0, 0, 0);
            if (s & 1) accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[s >> 1], qu[s >> 1], accQ, 0, 0, 0);
        }
#endif
#ifdef LRF_W_STAMPS
        asm volatile("" ::"v"(accP[0][0]), "v"(accP[3][3]), "v"(accQ[0]));
#endif
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s7);
        WSTAMP_ADD(c_w7, s6, s7); // P / Q MFMAs
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
        for (int reg = 0; reg < 4; reg++) Pp[(4 * (4 * lq + reg) + c) * LRF_RP + li] = accP[c][reg];
    // b' partial: D holds the even row groups in its upper-left 8 x 8 block and the odd ones in the lower-right block
    // (exact integers: the order of the final addition is immaterial); the cross blocks are not used
    float* Qp = Qpart + slot * LRF_RP * LRF_RP;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int i = 4 * lq + reg;
        const float mine = accQ[reg];
        // lane (li, lq) with i, li < 8 needs D[i + 8][li + 8]: lane (li + 8, lq + 2), same register
        const float other = __shfl(mine, ((lq + 2) & 3) * 16 + ((li + 8) & 15), 64);
        Qp[i * LRF_RP + li] = (i < 8 && li < 8) ? mine + other : 0.f;
    }
#ifdef LRF_W_STAMPS
    if (lane == 0 && blk < 16384) {
        WSTAMP(t_end);
        unsigned long long* o = g_stamps + 8 * blk;
        o[0] = t_end - t_begin; o[1] = c_w1; o[2] = c_w2; o[3] = c_w3; o[4] = c_w4; o[5] = c_w5; o[6] = c_w6; o[7] = c_w7;
    }
#endif
}
