#pragma once
// lrf_bcdw32_kernel.hip — k_bcd_w32: the BCD half-iteration (U update + partials of the V update) of iterations >= 2 for
// ranks 17..32 with one *wave* per (matrix, 384-row block) and no workgroup barrier — k_bcd_w16's frame for the rank
// family the reference's quality sweep reaches beyond quality 25 (lrf/factorization/qmf.py:93-126, 128-139;
// experiments/comparison/eval.py:83).  Included by lrf_api.hip after lrf_bcdw16_kernel.hip.  Replaces k_bcd_mid<0>
// (four waves per block, sixteen rows x four lanes in the Gauss-Seidel, three workgroup barriers per sub-tile) where the
// exact-integer conditions below hold; k_bcd_mid keeps the first iteration (float old U), the caller's-U0 mode and the
// wider bounds.
//
// Per 64-row sub-tile, one wave:
//   1. the prefetched X sub-tile (64 VGPRs) goes to the wave's XOR-swizzled LDS tile, the next one is requested;
//   2. a^T = V^T X^T on the f32 matrix cores, two rank tiles: 128 v_mfma_f32_16x16x4_f32 in eight independent chains (each the
//      k-ordered fma chain of the reference's sgemm), the eight D tiles become lane = row by 32 v_permlane swaps;
//   3. Gauss-Seidel with LANE = ROW (all 64 lanes on 64 rows) in EXACT INTEGERS.  From the second iteration on u and
//      b = v.mT @ v are integers and every partial sum of `uu @ bb` stays below 2^24 (host check (R-1) 64 mx^3 < 2^24), so
//      the reference's fp32 sum is the integer sum in any order.  The row lives as sixteen int16 PAIRS W[p] = (w[2p],
//      w[2p+1]), updated in place; column r's sum is the dot product of W with row r of the symmetric table (diagonal
//      zero): sixteen v_dot2c_i32_i16 — two terms per instruction — whose table operand is ONE VGPR (the row's sixteen
//      pair-dwords, one ds_read_b32 per column, lane l reading dword l & 15) behind the DPP row_newbcast modifier.  Only
//      the dot product with the pair that holds the column just solved sits on the column-to-column chain:
//      dot2 -> cvt -> a - T -> + eps -> * (1/den) -> rndne -> med3 -> magic add -> perm into W.  The quotient is
//      num * (1/den) with gs_row's tie test; a wave with a lane too close to call repeats the sub-tile's solve with the
//      IEEE division (rare).  Needs |b| <= 32767, i.e. 64 mx^2 <= 32767 (mx <= 22): checked by the host.
//   4. the new row leaves as int8 (global memory: R bytes per row; LDS: 32 bytes per row);
//   5. a' += X^T u on the f32 matrix cores (128 MFMAs: four strided column tiles per ds_read_b128, two rank tiles; the
//      reference's k-ordered chain over the block's rows) and b' += u^T u on the INT8 matrix cores: the sixteen bytes
//      u[4 e + lq][16 t + li] a lane has just read for the f32 operand ARE an operand of v_mfma_i32_16x16x64_i8 (the k
//      index is summed over, so any fixed bijection rows <-> (lane group, byte) serves): four instructions instead of 64.
// 20 KB of LDS per wave (X tile 16 KB, int8 u tile 2 KB, int16 table 2 KB): two 4-wave workgroups fill a CU's 160 KB.

#define LRF_BCDW32_WAVES 4
#define LRF_BCDW32_WAVE_LDS(NP) (64 * 64 * 4 + 64 * 32 + 2 * (NP) * 16 * 4) // the table rows past 2 NP are never read
#define LRF_BCDW32_LDS(NP) (LRF_BCDW32_WAVES * LRF_BCDW32_WAVE_LDS(NP))

typedef short w32_s16x2 __attribute__((ext_vector_type(2)));

// diagnostic stamps of this kernel: -DLRF_STAMPS -DLRF_W32_STAMPS (tools/dev_stamps_w32.py; never the shipped library)
#if defined(LRF_STAMPS) && defined(LRF_W32_STAMPS)
#define W32STAMP(var) STAMP(var)
#define W32STAMP_ADD(acc, a, b) STAMP_ADD(acc, a, b)
#else
#undef LRF_W32_STAMPS
#define W32STAMP(var)
#define W32STAMP_ADD(acc, a, b)
#endif

// acc += dot2(tab[lane N of each 16-lane row], w): both int16 pairs
template <int N>
__device__ __forceinline__ void w32_dot2_bc16(int& acc, int tab, int w)
{
    asm("v_dot2c_i32_i16_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(w), "n"(N));
}
template <int N>
__device__ __forceinline__ float w32_mul_bc16(float tab, float x)
{
    float out;
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab), "v"(x), "n"(N));
    return out;
}

// bytes (b0, b1, b2, b3) of d -> the int16 pairs (b0, b1) and (b2, b3), sign-extended
__device__ __forceinline__ void w32_bytes_to_pairs(unsigned d, int& lo, int& hi)
{
    w32_s16x2 l = __builtin_bit_cast(w32_s16x2, __builtin_amdgcn_perm(0u, d, 0x010C000Cu)); // (b0 << 8, b1 << 8)
    w32_s16x2 h = __builtin_bit_cast(w32_s16x2, __builtin_amdgcn_perm(0u, d, 0x030C020Cu)); // (b2 << 8, b3 << 8)
    l = l >> 8;
    h = h >> 8;
    lo = __builtin_bit_cast(int, l);
    hi = __builtin_bit_cast(int, h);
}

// One column of the solve.  `row` = the table row of column RR (lane l: pair-dword l & 15).  NP = pairs in use.
// FAST: q~ = num * (1/den); q~ + 1.5 * 2^23 rounds it to the nearest-even integer n in the mantissa (|q~| < 2^22; beyond
// that the bit pattern still orders like q~, so the integer clamp decides), the clamp is a v_med3_i32 on the bit pattern
// against (magic + lo, magic + hi) and the low 16 bits are the int16 to store.  emax gathers |q~ - n|: above gp.fthr for
// some column (or a |q~| >= 2^22, where n is not its rounding) the caller repeats the sub-tile with the IEEE division
// (gs_row's argument: q~ is within 3 ulp of fl(num / den), so unless it sits that close to a tie both round alike).
template <int NP, int RR, bool FAST, int... Ps>
__device__ __forceinline__ void w32_col(const float a_r, int (&W)[16], const int row, const float rdv, const float dnv,
                                        const GsParams& gp, const int lob, const int hib, float& emax, std::integer_sequence<int, Ps...>)
{
    constexpr int PN = RR >= 1 ? (RR - 1) >> 1 : -1; // the pair that holds the column solved last
    int accA = 0, accB = 0;
    // the pairs whose values are older: two chains, off the column-to-column dependency
    ((Ps != PN ? w32_dot2_bc16<Ps>((Ps & 1) ? accB : accA, row, W[Ps]) : (void)0), ...);
    int s = accA + accB;
    if constexpr (PN >= 0) w32_dot2_bc16<(PN >= 0 ? PN : 0)>(s, row, W[PN >= 0 ? PN : 0]);
    const float num = (a_r - (float)s) + LRF_EPS;
    unsigned ub;
    if (FAST) {
        const float q = w32_mul_bc16<RR & 15>(rdv, num);
        const float t = q + 12582912.0f;
        emax = fmaxf(emax, fabsf(q - (t - 12582912.0f)));
        int c;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(__float_as_int(t)), "s"(lob), "v"(hib));
        ub = (unsigned)c;
    } else {
        const float val = rintf(num / get_bc16<RR & 15>(dnv));
        const float u = __builtin_amdgcn_fmed3f(val, gp.lo, gp.hi);
        ub = __float_as_uint(u + 12582912.0f); // low 16 bits: u as int16
    }
    W[RR >> 1] = (int)((RR & 1) ? __builtin_amdgcn_perm(ub, (unsigned)W[RR >> 1], 0x05040100u)
                                : __builtin_amdgcn_perm(ub, (unsigned)W[RR >> 1], 0x03020504u));
}

// All columns of one row.  tabl: this lane's column of the wave's LDS table (tabl[16 * r] = pair-dword l & 15 of row r);
// rd0 / rd1: lane l = 1 / den[l & 15] and 1 / den[16 + (l & 15)] (1 for columns past R); dn0 / dn1: den likewise.
template <int NP, bool FAST, int... Rs>
__device__ __forceinline__ bool w32_solve(const float (&a)[32], int (&W)[16], const int* tabl, const float rd0, const float rd1,
                                          const float dn0, const float dn1, const GsParams& gp, std::integer_sequence<int, Rs...>)
{
    float emax = 0.f;
    const int lob = 0x4B400000 + (int)gp.lo, hib = 0x4B400000 + (int)gp.hi;
    int rows[2 * NP + 2];
    rows[0] = tabl[0];
    rows[1] = tabl[16];
    ((rows[Rs + 2] = tabl[16 * (Rs + 2 < 2 * NP ? Rs + 2 : 2 * NP - 1)],
      w32_col<NP, Rs, FAST>(a[Rs], W, rows[Rs], Rs < 16 ? rd0 : rd1, Rs < 16 ? dn0 : dn1, gp, lob, hib, emax, std::make_integer_sequence<int, NP>{})),
     ...);
    return FAST && !(emax <= gp.fthr);
}

// acc[T][i] (lane (li, lq)) = a[16T + li][4lq + i]  ->  out[4j + i] (lane L) = a[L][4j + i]   (w16_tiles_to_rows on a slice)
__device__ __forceinline__ void w32_tiles_to_rows(const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3, float* out)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned t0 = __float_as_uint(a0[i]), t1 = __float_as_uint(a1[i]);
        unsigned t2 = __float_as_uint(a2[i]), t3 = __float_as_uint(a3[i]);
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

// NP: pairs of rank columns in use (ceil(R / 2), 9..16); an odd R solves one padding column whose table row, a and result are zero
// One (matrix, 384-row block) on one wave.  Xs: the wave's LRF_BCDW32_WAVE_LDS(NP) bytes of LDS (X tile, int8 u tile, int16 table).
// MEM (lrf_device.h): MemLaunch for the launch-per-iteration kernel below, MemSc1 inside the persistent kernel (lrf_bcdp_kernel.hip).
// blk_id: only the diagnostic stamps use it.
template <int NP, class MEM>
__device__ __forceinline__ void w32_block(const float* __restrict__ X, const PlaneDesc& pd, const BlockDesc& bd, const float* __restrict__ Vf,
                                          const float* __restrict__ Bf, int8_t* __restrict__ U, float* __restrict__ Ppart,
                                          float* __restrict__ Qpart, const GsParams& gp, float* Xs, const int lane, const int blk_id)
{
    int8_t* us8 = reinterpret_cast<int8_t*>(Xs + 64 * 64);
    int* tab16 = reinterpret_cast<int*>(us8 + 64 * 32);
    const int R = pd.R; // 2 NP - 1 or 2 NP
    const int li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RPB;
    const float* gt = Bf + (long)bd.plane * LRF_GTB_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;

    // A operand of a^T = V^T X^T, resident: va[t][s] = V[4s + lq][16 t + li] (columns >= R of the table are zero).
    // T1V (ranks up to 24): the second rank tile holds at most eight columns — its a = x V runs on the VALU with lane = row
    // (k_bcd_w's device: 64 k-steps x NC v_fmac_f32_dpp per row, the same k-ordered fma chain as the MFMA; V[k][16 + c]
    // broadcast out of vb[c][k >> 4] by DPP row_newbcast) instead of 64 MFMAs of which a quarter to a half would be used.
    constexpr bool T1V = NP <= 12;
    constexpr int NC = T1V ? 2 * NP - 16 : 1; // columns 16 .. 2 NP - 1
    float va[T1V ? 1 : 2][16], vb[NC][4];
#pragma unroll
    for (int t = 0; t < (T1V ? 1 : 2); t++)
#pragma unroll
        for (int s = 0; s < 16; s++) va[t][s] = MEM::ld(Vp + (4 * s + lq) * LRF_RPB + 16 * t + li);
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) vb[c][g4] = T1V ? MEM::ld(Vp + (16 * g4 + li) * LRF_RPB + 16 + c) : 0.f;
    // the symmetric int16 table, diagonal zero: dword (r, p) = (b[r][2p], b[r][2p+1]); lane l builds dwords l, l + 64, ...
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int idx = e * 64 + lane, r = idx >> 4, c0 = 2 * (idx & 15), c1 = c0 + 1;
        float b0 = 0.f, b1 = 0.f;
        if (r < R && c0 < R && c0 != r) b0 = MEM::ld(gt + c0 * LRF_GTB_LD + (r < c0 ? r : r - 1)); // b[r][c0]: gt row c lists b[j][c], j != c
        if (r < R && c1 < R && c1 != r) b1 = MEM::ld(gt + c1 * LRF_GTB_LD + (r < c1 ? r : r - 1));
        if (r < 2 * NP) tab16[idx] = (int)(((unsigned)(int)b0 & 0xffffu) | ((unsigned)(int)b1 << 16));
    }
    // 1 / den and den of the columns li and 16 + li (padding columns: 1)
    const float dn0 = (li < R) ? MEM::ld(gt + li * LRF_GTB_LD + LRF_GTB_DEN) : 1.f;
    const float dn1 = (16 + li < R) ? MEM::ld(gt + (16 + li) * LRF_GTB_LD + LRF_GTB_DEN) : 1.f;
    const float rd0 = 1.0f / dn0, rd1 = 1.0f / dn1;
    const int* tabl = tab16 + li;

    // prefetch registers: xq[T][q] = X[r0 + 16T + 4q + lq][4li .. +3] (each load instruction: four whole rows, 1 KB);
    // upre = the old int8 row of this lane.  MemLaunch: dword d from byte offset min(4d, R - 4) (unaligned loads); MemSc1: the
    // nine ALIGNED dwords that cover a row of up to 32 bytes at any offset, as compiler-tracked sc1 loads (w16_load_row's
    // scheme), funnel-shifted when the row is used.  Rows past the block's end: its last row.
    f32x4 xq[4][4];
    constexpr int NRAW = MEM::kSc1 ? 9 : 8;
    unsigned upre[NRAW];
    unsigned ush = 0u; // MemSc1: the byte offset of the prefetched row inside its first aligned dword
    // `live` false (no next sub-tile): the registers are cleared instead — a load under a bare `if` would keep their old
    // values alive across the whole body (76 spilled registers at NP = 16)
    auto issue_x = [&](int t, int T0, int T1, bool live) {
        const int r0 = t * 64;
#pragma unroll
        for (int T = T0; T < T1; T++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int row = r0 + 16 * T + 4 * q + lq;
                row = row < nrows ? row : nrows - 1;
                if (live) xq[T][q] = *reinterpret_cast<const f32x4*>(Xp + (long)row * 64 + 4 * li);
                else xq[T][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto load_u = [&](int t, unsigned (&w)[NRAW], unsigned& sh, bool live) {
        int row = t * 64 + lane;
        row = row < nrows ? row : nrows - 1;
        const int8_t* up = Ub + (long)row * R;
        if constexpr (MEM::kSc1) {
            const uintptr_t a0 = reinterpret_cast<uintptr_t>(up);
            const uintptr_t base = a0 & ~(uintptr_t)3, last = (a0 + R - 1) & ~(uintptr_t)3;
            sh = (unsigned)(a0 & 3);
#pragma unroll
            for (int d = 0; d < NRAW; d++) {
                const uintptr_t a = base + 4 * d <= last ? base + 4 * d : last;
                if (live) w[d] = MEM::ld_u32(reinterpret_cast<const unsigned*>(a));
                else w[d] = 0u;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const int off = 4 * d < R - 4 ? 4 * d : R - 4; // wave-uniform
                if (live) w[d] = *reinterpret_cast<const u32_unaligned*>(up + off);
                else w[d] = 0u;
            }
        }
    };
    // the partial dword (index R / 4 when R is not a multiple of 4) was loaded from offset R - 4: its row bytes sit in its
    // upper part.  Bytes at or past R hold neighbouring bytes of the row: harmless (their table entries are zero) and
    // never stored (the solve writes every pair in use; the pairs past NP are cleared)
    const int part_dw = (R & 3) ? (R >> 2) : 99, part_sh = 8 * (4 - (R & 3));

    // B operand of a^T: X[16T + li][4s + lq] lives at byte (16T + li) * 256 + ((16 s) ^ (16 li)) + 4 lq of the tile
    const char* xrow_b = reinterpret_cast<const char*>(Xs) + li * 256 + 4 * lq;
    const int g16 = 16 * li;
    // A operand of a' = X^T u, all four column tiles at once: chunk li of row 4s + lq (k_bcd_w)
    const float* xp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xp[e] = &Xs[lq * 64 + 4 * (li ^ (4 * e + lq))];
    // B operand of a' = X^T u and both operands of b' = u^T u: u[4s + lq][16 t + li], a byte of the int8 tile
    const int8_t* ub8 = us8 + lq * 32 + li;

    f32x4 accP[4][2];
    i32x4 accQ[2][2];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int t = 0; t < 2; t++) accP[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) accQ[i][j] = (i32x4){0, 0, 0, 0};

#ifdef LRF_W32_STAMPS
    unsigned long long c_w1 = 0, c_w2 = 0, c_w3 = 0, c_w4 = 0, c_w5 = 0, c_w6 = 0, c_w7 = 0;
#endif
    W32STAMP(t_begin);
    issue_x(0, 0, 4, true);
    load_u(0, upre, ush, true);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        W32STAMP(s0);
#ifdef LRF_W32_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
        W32STAMP(s1);
        W32STAMP_ADD(c_w1, s0, s1); // wait for the prefetch
        // ---- 1. sub-tile -> LDS, next sub-tile's loads into the same registers (in bursts, as in k_bcd_w)
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int m = 16 * T + 4 * q + lq;
                *reinterpret_cast<f32x4*>(&Xs[m * 64 + 4 * (li ^ (4 * q + lq))]) = xq[T][q];
            }
        int W[16];
        auto row_to_pairs = [&](const unsigned (&w8)[NRAW], const unsigned sh) {
#pragma unroll
            for (int d = 0; d < 8; d++) {
                unsigned w;
                if constexpr (MEM::kSc1) w = __builtin_amdgcn_alignbyte(w8[d + 1 < NRAW ? d + 1 : d], w8[d], sh);
                else w = (d == part_dw) ? (w8[d] >> part_sh) : w8[d];
                w32_bytes_to_pairs(w, W[2 * d], W[2 * d + 1]);
            }
#pragma unroll
            for (int p = NP; p < 16; p++) W[p < 16 ? p : 15] = 0;
        };
        row_to_pairs(upre, ush);
        const int tn = t + 1;
        const bool more = tn < nsub;
        issue_x(tn, 0, 1, more);
        load_u(tn, upre, ush, more);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        W32STAMP(s2);
        W32STAMP_ADD(c_w2, s1, s2); // LDS stores, old row -> pairs, prefetch issue
        // ---- 2. a^T = V^T X^T: eight independent chains of 16 MFMAs, then lane = row
        float a[32];
#ifdef LRF_W32_STAMPS
        unsigned long long s3 = 0;
#endif
        {
            constexpr int NT = T1V ? 1 : 2;
            f32x4 acc[4][NT];
#pragma unroll
            for (int T = 0; T < 4; T++)
#pragma unroll
                for (int tt = 0; tt < NT; tt++) acc[T][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float a1[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) a1[c] = 0.f;
            static_for<4>([&](auto hc) { // the operand reads in four quarters of 16 registers
                constexpr int h = decltype(hc)::value;
                float bx[4][4];
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
                        bx[s][T] = *reinterpret_cast<const float*>(xrow_b + T * 16 * 256 + ((16 * (4 * h + s)) ^ g16));
                // T1V: the lane's own row, k = 16 h .. 16 h + 15 (chunks 4 h .. 4 h + 3 of the swizzled tile)
                f32x4 xr[T1V ? 4 : 1];
                if constexpr (T1V) {
#pragma unroll
                    for (int cch = 0; cch < 4; cch++)
                        xr[cch] = *reinterpret_cast<const f32x4*>(&Xs[lane * 64 + 4 * ((4 * h + cch) ^ (lane & 15))]);
                }
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
#pragma unroll
                        for (int tt = 0; tt < NT; tt++)
                            acc[T][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[tt][4 * h + s], bx[s][T], acc[T][tt], 0, 0, 0);
                if constexpr (T1V) {
                    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                        (([&] {
                             constexpr int k = Ks; // k-step 16 h + k: V[16 h + k][16 + c] = lane k of vb[c][h]
#pragma unroll
                             for (int c = 0; c < NC; c++) {
                                 const float tabv = h == 0 ? vb[c][0] : (h == 1 ? vb[c][1] : (h == 2 ? vb[c][2] : vb[c][3]));
                                 fmac_bc16<k>(a1[c], tabv, xr[k >> 2][k & 3]);
                             }
                         }()),
                         ...);
                    }(std::make_integer_sequence<int, 16>{});
                }
            });
#ifdef LRF_W32_STAMPS
            asm volatile("" ::"v"(acc[3][NT - 1][3]), "v"(acc[0][0][0]));
            s3 = stamp_now();
            W32STAMP_ADD(c_w3, s2, s3); // operand reads + 128 MFMAs
#endif
            w32_tiles_to_rows(acc[0][0], acc[1][0], acc[2][0], acc[3][0], a);
            if constexpr (T1V) {
#pragma unroll
                for (int c = 0; c < 16; c++) a[16 + c] = c < NC ? a1[c < NC ? c : 0] : 0.f;
            } else {
                w32_tiles_to_rows(acc[0][NT - 1], acc[1][NT - 1], acc[2][NT - 1], acc[3][NT - 1], a + 16);
            }
        }
        issue_x(tn, 1, 2, more);
        __builtin_amdgcn_sched_barrier(0);
#ifdef LRF_W32_STAMPS
        asm volatile("" ::"v"(a[31]), "v"(a[0]));
        W32STAMP(s4);
        W32STAMP_ADD(c_w4, s3, s4); // tiles -> rows
#endif
        // ---- 3. Gauss-Seidel, lane = row, exact integers
        if (__any(w32_solve<NP, true>(a, W, tabl, rd0, rd1, dn0, dn1, gp, std::make_integer_sequence<int, 2 * NP>{}))) {
            // rare: repeat with the reference's IEEE division; the old row is read again (its stores come after the solve)
            // rather than kept in sixteen registers across the solve
            unsigned wr[NRAW], wsh = 0u;
            load_u(t, wr, wsh, true);
            row_to_pairs(wr, wsh);
            w32_solve<NP, false>(a, W, tabl, rd0, rd1, dn0, dn1, gp, std::make_integer_sequence<int, 2 * NP>{});
        }
        // int16 pairs -> bytes: dword d = columns 4d .. 4d+3 (rows past the block's end: zero)
        const int row = r0 + lane;
        unsigned o[8];
#pragma unroll
        for (int d = 0; d < 8; d++) {
            o[d] = __builtin_amdgcn_perm((unsigned)W[2 * d + 1], (unsigned)W[2 * d], 0x06040200u);
            if (row >= nrows) o[d] = 0u;
        }
        issue_x(tn, 2, 4, more);
        __builtin_amdgcn_sched_barrier(0);
#ifdef LRF_W32_STAMPS
        asm volatile("" ::"v"(o[7]), "v"(o[0]));
        W32STAMP(s5);
        W32STAMP_ADD(c_w5, s4, s5); // Gauss-Seidel + pack
#endif
        // ---- 4. the int8 row to LDS (operand of the partial products) and to global memory (R bytes)
        *reinterpret_cast<uint4*>(us8 + lane * 32) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(us8 + lane * 32 + 16) = make_uint4(o[4], o[5], o[6], o[7]);
        if (row < nrows) {
            int8_t* uo = Ub + (long)row * R;
#pragma unroll
            for (int d = 0; d < 4; d++) MEM::st_u32(uo + 4 * d, o[d]); // R >= 16
#pragma unroll
            for (int d = 4; d < 8; d++)
                if (4 * d + 4 <= R) MEM::st_u32(uo + 4 * d, o[d]); // wave-uniform
            if (R & 3) { // bytes R-4 .. R-1: the tail of the last full dword and the head of the partial one
                const int dl = R >> 2; // 4 .. 7
                unsigned lo_w = o[3], hi_w = o[4];
#pragma unroll
                for (int d = 4; d < 8; d++)
                    if (d == dl) {
                        lo_w = o[d - 1];
                        hi_w = o[d];
                    }
                MEM::st_u32(uo + R - 4, __builtin_amdgcn_alignbyte(hi_w, lo_w, (unsigned)(R & 3)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        W32STAMP(s6);
        W32STAMP_ADD(c_w6, s5, s6); // u -> LDS, int8 stores
        // ---- 5. a' += X^T u (four strided column tiles per LDS read, two rank tiles), b' += u^T u on the int8 matrix cores
        {
            int ui[2][16];
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int s = 0; s < 16; s++) ui[tt][s] = (int)ub8[128 * s + 16 * tt];
            i32x4 qa[2];
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned p01 = __builtin_amdgcn_perm((unsigned)ui[tt][4 * d + 1], (unsigned)ui[tt][4 * d], 0x0C0C0400u);
                    const unsigned p23 = __builtin_amdgcn_perm((unsigned)ui[tt][4 * d + 3], (unsigned)ui[tt][4 * d + 2], 0x04000C0Cu);
                    qa[tt][d] = (int)(p01 | p23);
                }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) accQ[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qa[i], qa[j], accQ[i][j], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 4; h++) {
                f32x4 px[4];
#pragma unroll
                for (int s = 0; s < 4; s++) px[s] = *reinterpret_cast<const f32x4*>(xp[s & 3] + 256 * (4 * h + s));
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int tt = 0; tt < 2; tt++) {
                        const float pu = (float)ui[tt][4 * h + s];
#pragma unroll
                        for (int c = 0; c < 4; c++) accP[c][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s][c], pu, accP[c][tt], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef LRF_W32_STAMPS
        asm volatile("" ::"v"(accP[3][1][3]), "v"(accP[0][0][0]), "v"(accQ[1][1][3]));
        W32STAMP(s7);
        W32STAMP_ADD(c_w7, s6, s7); // P / Q MFMAs
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef LRF_W32_STAMPS
    if (lane == 0 && blk_id < 16384) {
        W32STAMP(t_end);
        unsigned long long* o = g_stamps + 8 * blk_id;
        o[0] = t_end - t_begin; o[1] = c_w1; o[2] = c_w2; o[3] = c_w3; o[4] = c_w4; o[5] = c_w5; o[6] = c_w6; o[7] = c_w7;
    }
#endif
    // a' partial: tile c holds the columns 4 i + c: D[i = 4*lq + reg][j = li (r)] -> a'[4 i + c][16 t + li]
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RPB;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) MEM::st(Pp + (4 * (4 * lq + reg) + c) * LRF_RPB + 16 * tt + li, accP[c][tt][reg]);
    // b' partial: D[i = 4*lq + reg][j = li] of tile (ti, tj), exact integers (below 384 mx^2)
    float* Qp = Qpart + slot * LRF_RPB * LRF_RPB;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) MEM::st(Qp + (16 * i + 4 * lq + reg) * LRF_RPB + 16 * j + li, (float)accQ[i][j][reg]);
    (void)blk_id;
}

template <int NP>
__global__ __launch_bounds__(64 * LRF_BCDW32_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_bcd_w32(const float* __restrict__ X, const PlaneDesc* __restrict__ planes, const BlockDesc* __restrict__ blocks,
               const float* __restrict__ Vf, const float* __restrict__ Bf, int8_t* __restrict__ U, float* __restrict__ Ppart,
               float* __restrict__ Qpart, GsParams gp, int nblocks)
{
    extern __shared__ __attribute__((aligned(16))) float bcdw32_lds[]; // LRF_BCDW32_LDS bytes, per wave: X tile, int8 u, table
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * LRF_BCDW32_WAVES + wave;
    if (blk >= nblocks) return; // the waves of a workgroup never synchronise with each other
    float* Xs = reinterpret_cast<float*>(reinterpret_cast<char*>(bcdw32_lds) + wave * LRF_BCDW32_WAVE_LDS(NP));
    const BlockDesc bd = blocks[blk];
    const PlaneDesc pd = planes[bd.plane];
    w32_block<NP, MemLaunch>(X, pd, bd, Vf, Bf, U, Ppart, Qpart, gp, Xs, (int)(threadIdx.x & 63), blk);
}

// =====================================================================================================================
// k_bcd_w32f<R>: the FIRST iteration of ranks 17..32 in the same frame (one wave per block, lane = row, no workgroup barrier;
// replaces k_bcd_mid<1>, whose Gauss-Seidel of a sub-tile runs on one wave of four while three wait).  The old U is X @ W0 —
// floats, computed here by a second set of MFMA chains — and b = v0.mT @ v0 is not integer either, so the Gauss-Seidel is
// the reference's ORDERED chain (lrf/factorization/qmf.py:108-119): per column r the sum `uu @ bb` in MKL's single-column
// order (oracle/lrf_oracle.c dot_mkl_n1: fma(u1, b1, u0 b0), then the odd terms descending, the even ones ascending, the two
// chains added), every product rounded before it is added.  The table row of a column (b[j][r], j = 0..31; symmetric) sits
// in TWO VGPRs (one ds_read_b32 each, lane l reading entries l & 15 and 16 + (l & 15) of the wave's fp32 LDS table) and
// reaches the products through the DPP row_newbcast modifier, so a column costs R - 1 v_mul_f32_dpp + R - 2 v_add_f32 +
// the division chain, all 64 lanes on 64 rows.  R is a template parameter: every index of the ordered chains is a
// compile-time constant.  Any bounds (the result rows are int8 either way); planes small enough for ATen's native order
// ((R - 1) M < 400) stay on k_bcd_mid<1> (host check).  22.5 KB of LDS per wave (the table is fp32 here) and ~350 registers
// (both MFMA operand sets resident, a and the old row as floats): one-wave workgroups, ONE wave per SIMD — at two (256
// registers) the compiler spills 69..116 registers; one launch of ten, so the simple form was kept.
// =====================================================================================================================
#define LRF_BCDW32F_LDS (64 * 64 * 4 + 64 * 32 + 32 * 32 * 4)
#ifndef W32F_WAVES_PER_EU
#define W32F_WAVES_PER_EU 1
#endif

template <int J>
__device__ __forceinline__ float w32f_prod(const float t0, const float t1, const float u)
{
    return w32_mul_bc16<J & 15>(J < 16 ? t0 : t1, u);
}
// column index of the n-th "other" column of column RR
template <int RR>
__device__ __forceinline__ constexpr int w32f_col(int n) { return n < RR ? n : n + 1; }

// uu . bb of column RR in dot_mkl_n1's order, K = R - 1 >= 16 terms: each product is rounded, then added to its chain
template <int RR, int N>
__device__ __forceinline__ float w32f_prod_n(const float (&u)[32], const float t0, const float t1)
{
    constexpr int j = w32f_col<RR>(N);
    return w32f_prod<j>(t0, t1, u[j]);
}
template <int RR, int N> // odd terms N, N - 2, ..., 3
__device__ __forceinline__ void w32f_odd_chain(float& odd, const float (&u)[32], const float t0, const float t1)
{
    if constexpr (N >= 3) {
        odd = odd + w32f_prod_n<RR, N>(u, t0, t1);
        w32f_odd_chain<RR, N - 2>(odd, u, t0, t1);
    }
}
template <int RR, int N, int K> // even terms N, N + 2, ... < K
__device__ __forceinline__ void w32f_even_chain(float& even, const float (&u)[32], const float t0, const float t1)
{
    if constexpr (N < K) {
        even = even + w32f_prod_n<RR, N>(u, t0, t1);
        w32f_even_chain<RR, N + 2, K>(even, u, t0, t1);
    }
}
template <int R, int RR>
__device__ __forceinline__ float w32f_term2(const float (&u)[32], const float t0, const float t1)
{
    constexpr int K = R - 1;
    float odd = w32f_prod_n<RR, 0>(u, t0, t1);
    {
        constexpr int j1 = w32f_col<RR>(1);
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(odd) : "v"(j1 < 16 ? t0 : t1), "v"(u[j1]), "n"(j1 & 15));
    }
    constexpr int last_odd = ((K - 1) & 1) ? K - 1 : K - 2;
    w32f_odd_chain<RR, last_odd>(odd, u, t0, t1);
    float even = w32f_prod_n<RR, 2>(u, t0, t1);
    w32f_even_chain<RR, 4, K>(even, u, t0, t1);
    return odd + even;
}

// t0 / t1: the table row of column RR (prefetched by the caller's caller: two columns ahead, as in w32_solve — left to itself
// the compiler hoists the reads of ALL columns above the first one, 40-64 registers)
template <int R, int RR, bool FAST>
__device__ __forceinline__ void w32f_cols(const float (&a)[32], float (&u)[32], const float* tabl, float t0, float t1, float n0, float n1,
                                          const float rd0, const float rd1, const float dn0, const float dn1, const GsParams& gp,
                                          const int lob, const int hib, float& emax)
{
    if constexpr (RR < R) {
        constexpr int RN = RR + 2 < R ? RR + 2 : R - 1;
        const float m0 = tabl[32 * RN], m1 = tabl[32 * RN + 16]; // the row two columns ahead
        const float num = (a[RR] - w32f_term2<R, RR>(u, t0, t1)) + LRF_EPS;
        if (FAST) {
            const float q = w32_mul_bc16<RR & 15>(RR < 16 ? rd0 : rd1, num);
            const float t = q + 12582912.0f;
            emax = fmaxf(emax, fabsf(q - (t - 12582912.0f)));
            int c;
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(__float_as_int(t)), "s"(lob), "v"(hib));
            u[RR] = __int_as_float(c) - 12582912.0f;
        } else {
            const float val = rintf(num / get_bc16<RR & 15>(RR < 16 ? dn0 : dn1));
            u[RR] = __builtin_amdgcn_fmed3f(val, gp.lo, gp.hi);
        }
#ifdef W32F_COLUMN_FENCE
        __builtin_amdgcn_sched_barrier(0); // the products of later columns stay behind this column (register pressure)
#endif
        w32f_cols<R, RR + 1, FAST>(a, u, tabl, n0, n1, m0, m1, rd0, rd1, dn0, dn1, gp, lob, hib, emax);
    }
}

template <int R>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W32F_WAVES_PER_EU, W32F_WAVES_PER_EU)))
void k_bcd_w32f(const float* __restrict__ X, const PlaneDesc* __restrict__ planes, const BlockDesc* __restrict__ blocks,
                const float* __restrict__ Vf, const float* __restrict__ Wf, const float* __restrict__ Bf, int8_t* __restrict__ U,
                float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp, int nblocks)
{
    extern __shared__ __attribute__((aligned(16))) float bcdw32f_lds[]; // LRF_BCDW32F_LDS bytes: X tile, int8 u, fp32 table
    const int blk = blockIdx.x;
    if (blk >= nblocks) return;
    float* Xs = bcdw32f_lds;
    int8_t* us8 = reinterpret_cast<int8_t*>(Xs + 64 * 64);
    float* tab = reinterpret_cast<float*>(us8 + 64 * 32);
    const BlockDesc bd = blocks[blk];
    const PlaneDesc pd = planes[bd.plane]; // pd.R == R (host)
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RPB;
    const float* Wp = Wf + (long)bd.plane * 64 * LRF_RPB;
    const float* gt = Bf + (long)bd.plane * LRF_GTB_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;

    // A operands of a^T = V^T X^T and of (old u)^T = W0^T X^T, resident
    float va[2][16], wa[2][16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s = 0; s < 16; s++) {
            va[t][s] = Vp[(4 * s + lq) * LRF_RPB + 16 * t + li];
            wa[t][s] = Wp[(4 * s + lq) * LRF_RPB + 16 * t + li];
        }
    // the symmetric fp32 table tab[r][j] = b[j][r] (diagonal and everything past R: zero); lane l fills entries l, l + 64, ...
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int idx = e * 64 + lane, r = idx >> 5, j = idx & 31;
        tab[idx] = (r < R && j < R && j != r) ? gt[r * LRF_GTB_LD + (j < r ? j : j - 1)] : 0.f;
    }
    const float dn0 = (li < R) ? gt[li * LRF_GTB_LD + LRF_GTB_DEN] : 1.f;
    const float dn1 = (16 + li < R) ? gt[(16 + li) * LRF_GTB_LD + LRF_GTB_DEN] : 1.f;
    const float rd0 = 1.0f / dn0, rd1 = 1.0f / dn1;
    const float* tabl = tab + li;
    const int lob = 0x4B400000 + (int)gp.lo, hib = 0x4B400000 + (int)gp.hi;

    f32x4 xq[4][4];
    auto issue_x = [&](int t, int T0, int T1, bool live) {
        const int r0 = t * 64;
#pragma unroll
        for (int T = T0; T < T1; T++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int row = r0 + 16 * T + 4 * q + lq;
                row = row < nrows ? row : nrows - 1;
                if (live) xq[T][q] = *reinterpret_cast<const f32x4*>(Xp + (long)row * 64 + 4 * li);
                else xq[T][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    const char* xrow_b = reinterpret_cast<const char*>(Xs) + li * 256 + 4 * lq;
    const int g16 = 16 * li;
    const float* xp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xp[e] = &Xs[lq * 64 + 4 * (li ^ (4 * e + lq))];
    const int8_t* ub8 = us8 + lq * 32 + li;

    f32x4 accP[4][2];
    i32x4 accQ[2][2];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int t = 0; t < 2; t++) accP[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) accQ[i][j] = (i32x4){0, 0, 0, 0};

    issue_x(0, 0, 4, true);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        __builtin_amdgcn_sched_barrier(0);
        // ---- 1. sub-tile -> LDS, next sub-tile's loads
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int m = 16 * T + 4 * q + lq;
                *reinterpret_cast<f32x4*>(&Xs[m * 64 + 4 * (li ^ (4 * q + lq))]) = xq[T][q];
            }
        const int tn = t + 1;
        const bool more = tn < nsub;
        issue_x(tn, 0, 1, more);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        // ---- 2. a^T = V^T X^T and (old u)^T = W0^T X^T: sixteen chains of 16 MFMAs, then lane = row
        float a[32], u[32];
        auto old_u = [&](float (&dst)[32]) { // (also the rare repeat with the IEEE division starts from it)
            f32x4 acc[4][2];
#pragma unroll
            for (int T = 0; T < 4; T++)
#pragma unroll
                for (int tt = 0; tt < 2; tt++) acc[T][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 4; h++) {
                float bx[4][4];
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
                        bx[s][T] = *reinterpret_cast<const float*>(xrow_b + T * 16 * 256 + ((16 * (4 * h + s)) ^ g16));
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
#pragma unroll
                        for (int tt = 0; tt < 2; tt++)
                            acc[T][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tt][4 * h + s], bx[s][T], acc[T][tt], 0, 0, 0);
            }
            w32_tiles_to_rows(acc[0][0], acc[1][0], acc[2][0], acc[3][0], dst);
            w32_tiles_to_rows(acc[0][1], acc[1][1], acc[2][1], acc[3][1], dst + 16);
        };
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int T = 0; T < 4; T++)
#pragma unroll
                for (int tt = 0; tt < 2; tt++) acc[T][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 4; h++) {
                float bx[4][4];
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
                        bx[s][T] = *reinterpret_cast<const float*>(xrow_b + T * 16 * 256 + ((16 * (4 * h + s)) ^ g16));
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int T = 0; T < 4; T++)
#pragma unroll
                        for (int tt = 0; tt < 2; tt++)
                            acc[T][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[tt][4 * h + s], bx[s][T], acc[T][tt], 0, 0, 0);
            }
            w32_tiles_to_rows(acc[0][0], acc[1][0], acc[2][0], acc[3][0], a);
            w32_tiles_to_rows(acc[0][1], acc[1][1], acc[2][1], acc[3][1], a + 16);
        }
        old_u(u);
        issue_x(tn, 1, 2, more);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. the ordered Gauss-Seidel, lane = row
        {
            float emax = 0.f;
            w32f_cols<R, 0, true>(a, u, tabl, tabl[0], tabl[16], tabl[32], tabl[48], rd0, rd1, dn0, dn1, gp, lob, hib, emax);
            if (__any(!(emax <= gp.fthr))) { // rare: repeat with the reference's IEEE division
                old_u(u);
                w32f_cols<R, 0, false>(a, u, tabl, tabl[0], tabl[16], tabl[32], tabl[48], rd0, rd1, dn0, dn1, gp, lob, hib, emax);
            }
        }
        const int row = r0 + lane;
        unsigned o[8];
#pragma unroll
        for (int d = 0; d < 8; d++) {
            unsigned w = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (4 * d + k < R) w |= ((unsigned)(int)u[4 * d + k] & 0xffu) << (8 * k);
            o[d] = row < nrows ? w : 0u;
        }
        issue_x(tn, 2, 4, more);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4. the int8 row to LDS and to global memory (R bytes)
        *reinterpret_cast<uint4*>(us8 + lane * 32) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(us8 + lane * 32 + 16) = make_uint4(o[4], o[5], o[6], o[7]);
        if (row < nrows) {
            int8_t* uo = Ub + (long)row * R;
#pragma unroll
            for (int d = 0; d < 8; d++)
                if (4 * d + 4 <= R) *reinterpret_cast<u32_unaligned*>(uo + 4 * d) = o[d];
            if constexpr ((R & 3) != 0) // bytes R-4 .. R-1: the tail of the last full dword and the head of the partial one
                *reinterpret_cast<u32_unaligned*>(uo + R - 4) = __builtin_amdgcn_alignbyte(o[R >> 2], o[(R >> 2) - 1], (unsigned)(R & 3));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        // ---- 5. a' += X^T u, b' += u^T u (as k_bcd_w32)
        {
            int ui[2][16];
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int s = 0; s < 16; s++) ui[tt][s] = (int)ub8[128 * s + 16 * tt];
            i32x4 qa[2];
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned p01 = __builtin_amdgcn_perm((unsigned)ui[tt][4 * d + 1], (unsigned)ui[tt][4 * d], 0x0C0C0400u);
                    const unsigned p23 = __builtin_amdgcn_perm((unsigned)ui[tt][4 * d + 3], (unsigned)ui[tt][4 * d + 2], 0x04000C0Cu);
                    qa[tt][d] = (int)(p01 | p23);
                }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) accQ[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qa[i], qa[j], accQ[i][j], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 4; h++) {
                f32x4 px[4];
#pragma unroll
                for (int s = 0; s < 4; s++) px[s] = *reinterpret_cast<const f32x4*>(xp[s & 3] + 256 * (4 * h + s));
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int tt = 0; tt < 2; tt++) {
                        const float pu = (float)ui[tt][4 * h + s];
#pragma unroll
                        for (int c = 0; c < 4; c++) accP[c][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s][c], pu, accP[c][tt], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RPB;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Pp[(4 * (4 * lq + reg) + c) * LRF_RPB + 16 * tt + li] = accP[c][tt][reg];
    float* Qp = Qpart + slot * LRF_RPB * LRF_RPB;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Qp[(16 * i + 4 * lq + reg) * LRF_RPB + 16 * j + li] = (float)accQ[i][j][reg];
}
