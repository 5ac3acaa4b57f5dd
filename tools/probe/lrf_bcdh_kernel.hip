// NOT PART OF THE LIBRARY — an experiment kept for the record ("k_bcd_w, round 2"): bit-exact, 15 % SLOWER
// than k_bcd_w (0.162 against 0.141 ms per launch at the bench workload).  To try it again: include it from lrf_api.hip after
// lrf_bcdw_kernel.hip and launch it for MODE 0 when gp.exact_int (grid ceil(nb / 4), 256 threads, LRF_BCDH_LDS bytes).
// lrf_bcdh_kernel.hip — k_bcd_h: the BCD half-iteration of k_bcd_w (ranks <= 8; one wave per (matrix, 384-row block), no
// workgroup barrier) restructured for FOUR waves per SIMD.  Included by lrf_api.hip after lrf_bcdw_kernel.hip.
//
// k_bcd_w is bound by the dependent chain of each wave with only two waves per SIMD to interleave:
// its 16 KB X tile and 64 landing registers per wave allow no more.  Here a sub-tile is 32 rows: the 64 lanes are 32 rows x
// two column halves (lane = row + 32 h; half h owns the columns r = h mod 2), the tile is 8 KB, 32 registers land the
// prefetch, V takes 16 registers instead of 32 — 9 KB of LDS and <= 128 VGPRs per wave, sixteen waves per CU.
//   * a = x V: every lane walks its row's 64 k (both halves read the same LDS row) for its own <= 4 columns, V[k][r] through DPP
//     row_newbcast as in k_bcd_w (a 16-lane row lies inside one half) — the same k-ordered fma chain per element;
//   * Gauss-Seidel: iterations >= 2 with bounds where every term of `uu @ bb` is an exact integer (run_bcd checks
//     (R - 1) 64 mx^3 < 2^24; true for the default bounds at every rank <= 8), so the ordered chain of the reference equals
//     T[r] = sum_{j > r} u_old[j] b[j][r], then column by column u_r = project((a_r - T[r] + eps) / den_r) by the owning half,
//     v_permlane32_swap to the other half of the row, T[r'] += u_r b[r][r'] for the later columns (the b table in registers,
//     one value per lane of a quad, DPP quad_perm broadcast folded into the fma).  num * (1 / den) with gs_row's tie test; a
//     wave with any lane too close to call repeats its solve with the IEEE division.  The first iteration and wider bounds
//     stay on k_bcd_w.
//   * a' += X^T u and b' += u^T u as in k_bcd_w (MFMA, strided column tiles, the same row order: bit-identical partials).
#define LRF_BCDH_WAVES 4
#define LRF_BCDH_LDS (LRF_BCDH_WAVES * (32 * 64 + 32 * 8) * 4)

template <int I>
__device__ __forceinline__ void fmac_quad_bcast(float& acc, float tab, float x)
{
    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(I));
}
template <int I>
__device__ __forceinline__ float mul_quad_bcast(float tab, float x)
{
    float out;
    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab), "v"(x), "n"(I));
    return out;
}
template <int I>
__device__ __forceinline__ float get_quad_bcast(float tab)
{
    float out;
    asm("v_mov_b32_dpp %0, %1 quad_perm:[%2,%2,%2,%2] row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(tab), "n"(I));
    return out;
}
// the value the lane 32 away holds (the other column half of the same row)
__device__ __forceinline__ float other_half(float v, int h)
{
    const int x = __float_as_int(v);
    auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false); // s[0]: lanes 32-63 hold lanes 0-31; s[1]: lanes 0-31 hold lanes 32-63
    return __int_as_float(h ? s[0] : s[1]);
}
__device__ __forceinline__ unsigned other_half_u(unsigned v, int h)
{
    auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return h ? s[0] : s[1];
}

// a[i] = chain over k = 0..63 of V[k][2 i + h] x[k] for the lane's NR own columns (vreg[i][kb], lane l = V[16 kb + (l & 15)][2 i + h])
template <int NR, int J>
__device__ __forceinline__ void h_row_times_v_step(const f32x4 xv, const float (&vreg)[4][4], float (&acc)[4])
{
#pragma unroll
    for (int i = 0; i < NR; i++) fmac_row_bcast<(4 * J + 0) & 15>(acc[i], vreg[i][J >> 2], xv[0]);
#pragma unroll
    for (int i = 0; i < NR; i++) fmac_row_bcast<(4 * J + 1) & 15>(acc[i], vreg[i][J >> 2], xv[1]);
#pragma unroll
    for (int i = 0; i < NR; i++) fmac_row_bcast<(4 * J + 2) & 15>(acc[i], vreg[i][J >> 2], xv[2]);
#pragma unroll
    for (int i = 0; i < NR; i++) fmac_row_bcast<(4 * J + 3) & 15>(acc[i], vreg[i][J >> 2], xv[3]);
}
template <int NR>
__device__ __forceinline__ void h_row_times_v(const float* __restrict__ xrow_lds, int g16, const float (&vreg)[4][4], float (&acc)[4])
{
#define LRF_STEP(J)                                                                                                   \
    h_row_times_v_step<NR, J>(*reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xrow_lds) + ((16 * J) ^ g16)), vreg, acc);
    LRF_STEP(0) LRF_STEP(1) LRF_STEP(2) LRF_STEP(3) LRF_STEP(4) LRF_STEP(5) LRF_STEP(6) LRF_STEP(7)
    LRF_STEP(8) LRF_STEP(9) LRF_STEP(10) LRF_STEP(11) LRF_STEP(12) LRF_STEP(13) LRF_STEP(14) LRF_STEP(15)
#undef LRF_STEP
}

// One Gauss-Seidel step: column R0 (compile time), owned by half R0 & 1 in slot R0 >> 1.  tb[R0]: lane l holds b[R0][c] for its
// quad position's column c = 2 (l & 3) + h; rdenq / denq: lane l holds 1 / den[c], den[c].  The update reaches the columns
// c > R0: slot i for both halves when 2 i > R0, for the odd half only when 2 i = R0.
template <int R0, bool FAST>
__device__ __forceinline__ void h_gs_step(int h, const float (&a)[4], float (&T)[4], float (&un)[4], const float (&tbu)[8], float rdenq,
                                          float denq, const GsParams& gp, bool& unsafe)
{
    constexpr int HO = R0 & 1, IO = R0 >> 1;
    const float num = (a[IO] - T[IO]) + LRF_EPS;
    float val;
    if (FAST) {
        const float qt = mul_quad_bcast<IO>(rdenq, num);
        const float nq = rintf(qt);
        const bool inside = fabsf(qt) < gp.flimit;
        unsafe |= (h == HO) && inside && !(fabsf(qt - nq) <= gp.fthr);
        val = inside ? nq : qt;
    } else {
        val = rintf(num / get_quad_bcast<IO>(denq));
    }
    val = fminf(fmaxf(val, gp.lo), gp.hi);
    const float oth = other_half(val, h);
    const float vb = (h == HO) ? val : oth;
    un[IO] = (h == HO) ? val : un[IO];
    const float vb1 = h ? vb : 0.f; // slot R0 / 2 (R0 even): only the odd half's column R0 + 1 lies beyond R0
    if (0 > R0) fmac_quad_bcast<0>(T[0], tbu[R0], vb); else if (0 == R0) fmac_quad_bcast<0>(T[0], tbu[R0], vb1);
    if (2 > R0) fmac_quad_bcast<1>(T[1], tbu[R0], vb); else if (2 == R0) fmac_quad_bcast<1>(T[1], tbu[R0], vb1);
    if (4 > R0) fmac_quad_bcast<2>(T[2], tbu[R0], vb); else if (4 == R0) fmac_quad_bcast<2>(T[2], tbu[R0], vb1);
    if (6 > R0) fmac_quad_bcast<3>(T[3], tbu[R0], vb); else if (6 == R0) fmac_quad_bcast<3>(T[3], tbu[R0], vb1);
}

template <bool FAST>
__device__ __forceinline__ bool h_gs_solve(int R, int h, const float (&a)[4], float (&T)[4], float (&un)[4], const float (&tbu)[8],
                                           float rdenq, float denq, const GsParams& gp)
{
    bool unsafe = false;
    h_gs_step<0, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    if (R > 1) h_gs_step<1, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe); // R is wave-uniform
    if (R > 2) h_gs_step<2, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    if (R > 3) h_gs_step<3, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    if (R > 4) h_gs_step<4, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    if (R > 5) h_gs_step<5, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    if (R > 6) h_gs_step<6, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    if (R > 7) h_gs_step<7, FAST>(h, a, T, un, tbu, rdenq, denq, gp, unsafe);
    return unsafe;
}

__global__ __launch_bounds__(64 * LRF_BCDH_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_bcd_h(
    const float* __restrict__ X, const PlaneDesc* __restrict__ planes, const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
    const float* __restrict__ Bf, int8_t* __restrict__ U, float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp, int nblocks)
{
    constexpr int RMAX = 8;
    extern __shared__ __attribute__((aligned(16))) float bcdh_lds[]; // per wave: X tile [32][64] (swizzled), then u [32][8]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * LRF_BCDH_WAVES + wave;
    if (blk >= nblocks) return; // the waves of a workgroup never synchronise with each other
    float* Xs = bcdh_lds + wave * (32 * 64 + 32 * RMAX);
    float* us = Xs + 32 * 64;
    const BlockDesc bd = blocks[blk];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const int row32 = lane & 31, h = lane >> 5, qi = lane & 3;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RP;
    const float* gt = Bf + (long)bd.plane * LRF_GT_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 31) >> 5;
    const int nrh = (R + 1) >> 1; // columns per half: r = 2 i + h, i < nrh (the last one of the odd half may not exist)

    // V resident in registers: vreg[i][kb] = V[16 kb + li][2 i + h] (0 beyond the rank: the table is zero padded)
    float vreg[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int kb = 0; kb < 4; kb++) vreg[i][kb] = Vp[(16 * kb + li) * LRF_RP + 2 * i + h];

    // b table of the exact solve, one value per lane of a quad: tb[j], lane l = b[j][c] for the column c = 2 qi + h of its quad
    // position (0 on the diagonal and beyond the rank).  Which (j, c) pairs enter a sum — c < j for the T initialisation over the
    // columns still holding old values, c > j for the updates — is a compile-time fact except where it hinges on the half.
    float tb[8], rdenq, denq;
    {
        const int c = 2 * qi + h;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float b = 0.f;
            if (j < R && c < R && c != j) b = gt[c * LRF_GT_LD + (j < c ? j : j - 1)]; // gt row c lists b[j][c], j != c
            tb[j] = b;
        }
        denq = (c < R) ? gt[c * LRF_GT_LD + LRF_GT_DEN] : 1.f;
        rdenq = (c < R) ? gt[c * LRF_GT_LD + LRF_GT_RDEN] : 1.f;
    }

    // prefetch registers: xq[j] = X[r0 + 4 j + lq][4 li .. + 3]; upre: the row's old int8 U bytes (both halves load the same)
    f32x4 xq[8];
    unsigned upre[2];
    auto issue_x = [&](int t, int J0, int J1) {
        const int r0 = t * 32;
#pragma unroll
        for (int j = J0; j < J1; j++) {
            int row = r0 + 4 * j + lq;
            row = row < nrows ? row : nrows - 1; // clamped (finite data, no per-lane branches): u of rows past the end is forced to 0
            xq[j] = *reinterpret_cast<const f32x4*>(Xp + (long)row * 64 + 4 * li);
        }
    };
    auto issue_u = [&](int t) {
        int row = t * 32 + row32;
        row = row < nrows ? row : nrows - 1;
        const int8_t* up = Ub + (long)row * R;
        if (R >= 4) { // bytes 0..3 and bytes R-4..R-1 (overlapping): two unaligned dword loads
            upre[0] = *reinterpret_cast<const u32_unaligned*>(up);
            upre[1] = *reinterpret_cast<const u32_unaligned*>(up + R - 4);
        } else {
            unsigned b0 = (uint8_t)up[0], b1 = (uint8_t)up[R > 1 ? 1 : 0], b2 = (uint8_t)up[R > 2 ? 2 : 0];
            upre[0] = b0 | (b1 << 8) | (b2 << 16);
            upre[1] = 0;
        }
    };

    const float* xp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xp[e] = &Xs[lq * 64 + 4 * (li ^ (4 * e + lq))];
    const float* ub = &us[lq * RMAX + (li & 7)];
    const float* uq = &us[(lq + 4 * (li >> 3)) * RMAX + (li & 7)];
    const float* xrow = &Xs[row32 * 64];
    const int g16 = 16 * xsw(row32);

    f32x4 accP[4], accQ = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) accP[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_x(0, 0, 8);
    issue_u(0);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 32;
        __builtin_amdgcn_sched_barrier(0);
        // ---- 1. sub-tile -> LDS (row m = 4 j + lq, xsw(m) = 4 (j & 3) + lq), old U bytes -> floats, next sub-tile's loads
#pragma unroll
        for (int j = 0; j < 8; j++) *reinterpret_cast<f32x4*>(&Xs[(4 * j + lq) * 64 + 4 * (li ^ (4 * (j & 3) + lq))]) = xq[j];
        float uo[RMAX];
        {
            const unsigned lo = upre[0], hi = (R > 4) ? upre[1] >> (8 * (8 - R)) : 0u;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uo[r] = (float)(int)(int8_t)(lo >> (8 * r));
                uo[4 + r] = (float)(int)(int8_t)(hi >> (8 * r));
            }
        }
        const int tn = t + 1;
        const bool more = tn < nsub; // wave-uniform
        if (more) {
            issue_x(tn, 0, 3);
            issue_u(tn);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        // ---- 2. a = x V for the lane's own columns
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int g16v = g16;
        asm volatile("" : "+v"(g16v)); // the sixteen swizzled row addresses are recomputed per sub-tile (one xor each): hoisted out of
                                       // the loop they would hold sixteen registers this kernel does not have
        switch (nrh) {
        case 1: h_row_times_v<1>(xrow, g16v, vreg, a); break;
        case 2: h_row_times_v<2>(xrow, g16v, vreg, a); break;
        case 3: h_row_times_v<3>(xrow, g16v, vreg, a); break;
        default: h_row_times_v<4>(xrow, g16v, vreg, a); break;
        }
        if (more) issue_x(tn, 3, 6);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. Gauss-Seidel (exact-integer form), the two halves of a row in lock-step
        float T0[4] = {0.f, 0.f, 0.f, 0.f};
        // T[i] = sum over j > c = 2 i + h of u_old[j] b[j][c]: j > 2 i + 1 for both halves, j = 2 i + 1 for the even half only
#define LRF_H_TINIT(J)                                                                                  \
        if (J < R) { /* wave-uniform */                                                                     \
            const float uj = uo[J], uj0 = h ? 0.f : uo[J];                                                  \
            if (J > 1) fmac_quad_bcast<0>(T0[0], tb[J], uj); else if (J == 1) fmac_quad_bcast<0>(T0[0], tb[J], uj0); \
            if (J > 3) fmac_quad_bcast<1>(T0[1], tb[J], uj); else if (J == 3) fmac_quad_bcast<1>(T0[1], tb[J], uj0); \
            if (J > 5) fmac_quad_bcast<2>(T0[2], tb[J], uj); else if (J == 5) fmac_quad_bcast<2>(T0[2], tb[J], uj0); \
            if (J > 7) fmac_quad_bcast<3>(T0[3], tb[J], uj); else if (J == 7) fmac_quad_bcast<3>(T0[3], tb[J], uj0); \
        }
        LRF_H_TINIT(1) LRF_H_TINIT(2) LRF_H_TINIT(3) LRF_H_TINIT(4) LRF_H_TINIT(5) LRF_H_TINIT(6) LRF_H_TINIT(7)
#undef LRF_H_TINIT
        float T[4], un[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { T[i] = T0[i]; un[i] = 0.f; }
        if (__any(h_gs_solve<true>(R, h, a, T, un, tb, rdenq, denq, gp))) { // rare: repeat with the reference's IEEE division
#pragma unroll
            for (int i = 0; i < 4; i++) { T[i] = T0[i]; un[i] = 0.f; }
            h_gs_solve<false>(R, h, a, T, un, tb, rdenq, denq, gp);
        }
        const int row = r0 + row32;
        const bool live = row < nrows;
#pragma unroll
        for (int i = 0; i < 4; i++) un[i] = (live && 2 * i + h < R) ? un[i] : 0.f;
        if (more) issue_x(tn, 6, 8);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4. u -> LDS (natural column order), int8 row out
#pragma unroll
        for (int i = 0; i < 4; i++) us[row32 * RMAX + 2 * i + h] = un[i];
        {
            unsigned mine = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) mine |= ((unsigned)(int)un[i] & 0xffu) << (8 * i);
            const unsigned oth = other_half_u(mine, h);
            const unsigned ev = h ? oth : mine, od = h ? mine : oth; // bytes of the even / odd columns
            // row bytes: r even -> ev byte r/2, r odd -> od byte r/2: lo = [e0 o0 e1 o1], hi = [e2 o2 e3 o3]
            const unsigned lo = __builtin_amdgcn_perm(od, ev, 0x05010400u), hi = __builtin_amdgcn_perm(od, ev, 0x07030602u);
            if (live) {
                int8_t* uo8 = Ub + (long)row * R;
                if (R >= 4) {
                    if (h == 0) {
                        *reinterpret_cast<u32_unaligned*>(uo8) = lo;
                    } else {
                        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
                        *reinterpret_cast<u32_unaligned*>(uo8 + R - 4) = (unsigned)(w >> (8 * (R - 4)));
                    }
                } else if (h == 0) {
                    uo8[0] = (int8_t)lo;
                    if (R > 1) uo8[1] = (int8_t)(lo >> 8);
                    if (R > 2) uo8[2] = (int8_t)(lo >> 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        // ---- 5. a' += X^T u (four strided column tiles), b' += u^T u (two row groups per MFMA)
        float pu[8], qu[4];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const float v = ub[4 * s * RMAX];
            pu[s] = (li < RMAX) ? v : 0.f;
        }
#pragma unroll
        for (int h2 = 0; h2 < 4; h2++) qu[h2] = uq[8 * h2 * RMAX];
#pragma unroll
        for (int s0 = 0; s0 < 8; s0 += 4) { // the operand in two batches of four row steps: sixteen registers instead of thirty-two
            f32x4 px[4];
#pragma unroll
            for (int s = 0; s < 4; s++) px[s] = *reinterpret_cast<const f32x4*>(xp[s] + 256 * (s0 + s));
#pragma unroll
            for (int s = 0; s < 4; s++) {
#pragma unroll
                for (int c = 0; c < 4; c++) accP[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s][c], pu[s0 + s], accP[c], 0, 0, 0);
                if (s & 1) accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[(s0 + s) >> 1], qu[(s0 + s) >> 1], accQ, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // partials, as k_bcd_w writes them
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RP;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Pp[(4 * (4 * lq + reg) + c) * LRF_RP + li] = accP[c][reg];
    float* Qp = Qpart + slot * LRF_RP * LRF_RP;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int i = 4 * lq + reg;
        const float mine = accQ[reg];
        const float other = __shfl(mine, ((lq + 2) & 3) * 16 + ((li + 8) & 15), 64);
        Qp[i * LRF_RP + li] = (i < 8 && li < 8) ? mine + other : 0.f;
    }
}
