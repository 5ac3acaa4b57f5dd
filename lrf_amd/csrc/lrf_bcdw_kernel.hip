// lrf_bcdw_kernel.hip — k_bcd_w: the BCD half-iteration (U update + partials of the V update) with one *wave* per
// (matrix, 384-row block) and no workgroup barrier at all.  Included by lrf_api.hip after lrf_kernels.hip.
//
// Per 64-row sub-tile the wave
//   1. takes the sub-tile from registers (prefetched one sub-tile ahead: 16 float4 loads, lane = (row, 16-byte chunk)),
//      stores it to its private, XOR-swizzled LDS tile and transposes it in registers (v_permlane*_swap) into the
//      MFMA operand layout of a = X V;
//   2. runs four independent 16-step MFMA chains (one per 16 rows; V operand resident in 16 VGPRs);
//   3. turns the four D tiles into lane = row with 16 more permlane swaps and solves the Gauss-Seidel recurrence in
//      registers (gs_row, lrf_kernels.hip) — no LDS round trip, no other wave to wait for;
//   4. writes the int8 row, parks u in LDS and accumulates a' += X^T u (four independent chains, operand read
//      transposed from the LDS tile) and b' += u^T u.
// The summation orders are those of k_bcd (and therefore of the reference): k-ordered fma chains, one chain per
// 384-row block, block partials added in order by k_vupdate.
// Two waves per SIMD (<= 256 VGPRs, 18 KB LDS per wave): while one wave is in its Gauss-Seidel the other feeds the
// MFMA pipe; the next sub-tile's global loads are in flight during steps 3 and 4.

// LDS X tile: element (m, n) of the 64 x 64 sub-tile lives at dword m*64 + 4*((n >> 2) ^ xsw(m)) + (n & 3):
// the float4 stores (lane = row, 8 consecutive rows per LDS cycle) and the transposed dword loads (16 columns of
// two rows per LDS cycle) both touch 32 distinct banks.
__device__ __forceinline__ int xsw(int m) { return ((m & 1) << 2) | ((m >> 1) & 3); }

// acc[T][i] (lane (li, lq)) = a[16T + li][4lq + i]  ->  out[4j + i] (lane L) = a[L][4j + i]
template <int RMAX>
__device__ __forceinline__ void tiles_to_rows(const f32x4 (&acc)[4], float (&out)[RMAX])
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
        if (RMAX == 16) {
            out[8 + i] = __uint_as_float(s02[1]);
            out[12 + i] = __uint_as_float(s13[1]);
        }
    }
}

// Gauss-Seidel of one row held in registers: a[0..R), u[0..R) in/out (entries >= R are zeroed)
template <int R, int RMAX>
__device__ __forceinline__ void gs_regs(const float (&a)[RMAX], float (&u)[RMAX], const float* __restrict__ gt, bool native,
                                        const GsParams gp)
{
    float aa[R], uu[R], u0[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        aa[r] = a[r];
        u0[r] = u[r];
        uu[r] = u[r];
    }
    bool unsafe = native ? gs_row<R, true, false>(aa, uu, gt, gp) : gs_row<R, false, false>(aa, uu, gt, gp);
    if (__any(unsafe)) { // rare: redo with the reference's IEEE division
#pragma unroll
        for (int r = 0; r < R; r++) uu[r] = u0[r];
        if (native) gs_row<R, true, true>(aa, uu, gt, gp);
        else gs_row<R, false, true>(aa, uu, gt, gp);
    }
#pragma unroll
    for (int r = 0; r < RMAX; r++) u[r] = (r < R) ? uu[r < R ? r : 0] : 0.f;
}

template <int RMAX>
__device__ __forceinline__ void gs_regs_dispatch(int R, const float (&a)[RMAX], float (&u)[RMAX], const float* __restrict__ gt,
                                                 bool native, const GsParams gp)
{
    switch (R) {
#define LRF_CASE(r)                                                                \
    case r:                                                                        \
        if (r <= RMAX) gs_regs<(r <= RMAX ? r : 1), RMAX>(a, u, gt, native, gp);   \
        break;
        LRF_CASE(1) LRF_CASE(2) LRF_CASE(3) LRF_CASE(4) LRF_CASE(5) LRF_CASE(6) LRF_CASE(7) LRF_CASE(8)
        LRF_CASE(9) LRF_CASE(10) LRF_CASE(11) LRF_CASE(12) LRF_CASE(13) LRF_CASE(14) LRF_CASE(15) LRF_CASE(16)
#undef LRF_CASE
    }
}

// MODE 0: old U from int8 (iterations >= 2); MODE 1: first iteration, old U = X @ W0 computed here;
// MODE 2: first iteration, old U = caller's fp32 U0.
template <int MODE, int RMAX>
__global__ __launch_bounds__(64) void k_bcd_w(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                              const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                              const float* __restrict__ Wf, const float* __restrict__ Bf,
                                              const float* __restrict__ U0, int8_t* __restrict__ U,
                                              float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp)
{
    __shared__ __attribute__((aligned(16))) float Xs[64 * 64];
    __shared__ __attribute__((aligned(16))) float us[64 * RMAX];

    const BlockDesc bd = blocks[blockIdx.x];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RP;
    const float* gt = Bf + (long)bd.plane * LRF_GT_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;
    const bool native = pd.native_t2_u != 0;

    // A operand of a^T = V^T X^T, resident: va[s] = V[4s + lq][li]
    float va[16], wa[MODE == 1 ? 16 : 1];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        va[s] = Vp[(4 * s + lq) * LRF_RP + li];
        if (MODE == 1) wa[s] = Wf[(long)bd.plane * 64 * LRF_RP + (4 * s + lq) * LRF_RP + li];
    }

    // prefetch registers: xq[T][q] = X[r0 + 16T + li][16q + 4lq .. +3]; upre[r] = old int8 U[r0 + lane][r].
    // Rows past the end of the block are clamped to its last row (finite data, no branches): their u is forced to 0.
    f32x4 xq[4][4];
    int8_t upre[RMAX];
    auto issue = [&](int t) {
        const int r0 = t * 64;
#pragma unroll
        for (int T = 0; T < 4; T++) {
            int row = r0 + 16 * T + li;
            row = row < nrows ? row : nrows - 1;
            const float* src = Xp + (long)row * 64 + 4 * lq;
#pragma unroll
            for (int q = 0; q < 4; q++) xq[T][q] = *reinterpret_cast<const f32x4*>(src + 16 * q);
        }
        if (MODE == 0) {
            int row = r0 + lane;
            row = row < nrows ? row : nrows - 1;
            const int8_t* up = Ub + (long)row * R;
#pragma unroll
            for (int r = 0; r < RMAX; r++) upre[r] = up[r < R ? r : R - 1];
        }
    };

    // transposed-operand read bases: X[4s + lq][16c + li] at xb[s & 1][c & 1][256 s + 32 (c >> 1)]
    const float* xb[2][2];
    {
        const int cb = (li >> 2) ^ (((lq & 1) << 2) | (lq >> 1));
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int k = 0; k < 2; k++) xb[e][k] = &Xs[lq * 64 + 4 * ((cb ^ (4 * k)) ^ (2 * e)) + (li & 3)];
    }
    const float* ub = &us[lq * RMAX + (li < RMAX ? li : 0)];

    f32x4 accP[4], accQ = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) accP[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef LRF_STAMPS
    unsigned long long c_w1 = 0, c_w2 = 0, c_w3 = 0, c_w4 = 0, c_w5 = 0, c_w6 = 0, c_w7 = 0;
#endif
    STAMP(t_begin);
    issue(0);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        STAMP(s0);
#ifdef LRF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(s1);
        STAMP_ADD(c_w1, s0, s1); // wait for the prefetch
        // ---- 1. sub-tile -> LDS (raw layout), then the in-register transposes
#pragma unroll
        for (int T = 0; T < 4; T++) {
            const int m = 16 * T + li;
            const int g = xsw(m);
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4*>(&Xs[m * 64 + 4 * ((4 * q + lq) ^ g)]) = xq[T][q];
        }
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) rows_transpose4(xq[T][q]); // xq[T][q][i] = X[row][16q + 4i + lq]
#ifdef LRF_STAMPS
        asm volatile("" ::"v"(xq[3][3][3]), "v"(xq[0][0][0]));
#endif
        STAMP(s2);
        STAMP_ADD(c_w2, s1, s2); // LDS stores + transposes
        // ---- 2. a^T tiles: four independent chains of 16 MFMAs
        f32x4 acc[4], accw[4];
#pragma unroll
        for (int T = 0; T < 4; T++) {
            acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) accw[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < 16; s++)
#pragma unroll
            for (int T = 0; T < 4; T++) {
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s], xq[T][s >> 2][s & 3], acc[T], 0, 0, 0);
                if (MODE == 1) accw[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xq[T][s >> 2][s & 3], accw[T], 0, 0, 0);
            }
        // ---- 3. lane = row: a, old u; Gauss-Seidel in registers
        float a[RMAX], u[RMAX];
        tiles_to_rows<RMAX>(acc, a);
#ifdef LRF_STAMPS
        asm volatile("" ::"v"(a[0]), "v"(a[RMAX - 1]));
#endif
        STAMP(s3);
        STAMP_ADD(c_w3, s2, s3); // U-phase MFMAs + shuffle
        const int row = r0 + lane;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = (float)upre[r];
        } else if constexpr (MODE == 1) {
            tiles_to_rows<RMAX>(accw, u);
        } else {
            const float* up = U0 + pd.u0_off + ((long)bd.row0 + (row < nrows ? row : nrows - 1)) * R;
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = up[r < R ? r : R - 1];
        }
        issue(t + 1 < nsub ? t + 1 : t); // the registers are free again; unconditional: exact s_waitcnt counts
        STAMP(s4);
        STAMP_ADD(c_w4, s3, s4); // old u + prefetch issue
        gs_regs_dispatch<RMAX>(R, a, u, gt, native, gp);
#ifdef LRF_STAMPS
        asm volatile("" ::"v"(u[0]), "v"(u[RMAX - 1]));
#endif
        STAMP(s5);
        STAMP_ADD(c_w5, s4, s5); // Gauss-Seidel
        if (row >= nrows) {
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = 0.f;
        }
        // ---- 4. u -> LDS (B operand of the partial products), int8 row out
#pragma unroll
        for (int r = 0; r < RMAX; r += 4) *reinterpret_cast<f32x4*>(&us[lane * RMAX + r]) = (f32x4){u[r], u[r + 1], u[r + 2], u[r + 3]};
        if (row < nrows) {
            int8_t* uo = Ub + (long)row * R;
#pragma unroll
            for (int r = 0; r < RMAX; r++)
                if (r < R) uo[r] = (int8_t)u[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(s6);
        STAMP_ADD(c_w6, s5, s6); // u -> LDS, int8 stores
        float pu[16];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            float v = ub[4 * s * RMAX];
            pu[s] = (li < RMAX) ? v : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; s++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float px = xb[s & 1][c & 1][256 * s + 32 * (c >> 1)];
                accP[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, pu[s], accP[c], 0, 0, 0);
            }
            accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(pu[s], pu[s], accQ, 0, 0, 0);
        }
#ifdef LRF_STAMPS
        asm volatile("" ::"v"(accP[0][0]), "v"(accP[3][3]), "v"(accQ[0]));
#endif
        STAMP(s7);
        STAMP_ADD(c_w7, s6, s7); // P / Q MFMAs
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // a' partial: D[i = 4*lq + reg (column 16c + i)][j = li (r)];  b' partial: D[i = 4*lq + reg][j = li]
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RP;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Pp[(16 * c + 4 * lq + reg) * LRF_RP + li] = accP[c][reg];
    float* Qp = Qpart + slot * LRF_RP * LRF_RP;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) Qp[(4 * lq + reg) * LRF_RP + li] = accQ[reg];
#ifdef LRF_STAMPS
    if (lane == 0 && blockIdx.x < 16384) {
        STAMP(t_end);
        unsigned long long* o = g_stamps + 8 * blockIdx.x;
        o[0] = t_end - t_begin; o[1] = c_w1; o[2] = c_w2; o[3] = c_w3; o[4] = c_w4; o[5] = c_w5; o[6] = c_w6; o[7] = c_w7;
    }
#endif
}
