#pragma once
// lrf_midrank_kernels.hip — the BCD half-iteration (U update + partials of the V update) for ranks 17..32, the part of
// the reference's quality sweep (experiments/comparison/eval.py:83: linspace(0, 40, 80)) beyond quality 25.  Included by
// lrf_api.hip after lrf_bigrank_kernels.hip, whose table layouts (rank pitch LRF_RPB = 64, gt pitch LRF_GTB_LD) and V
// update LDS carve (BigVLds) it shares.
//
// Same arithmetic and summation orders as k_bcd (reference lrf/factorization/qmf.py:93-139).  Against the first kernel
// for these ranks, which padded every rank to four 16-wide MFMA tiles and let one wave of four solve a sub-tile's Gauss-Seidel
// lane = row while the others waited:
//   * two rank tiles: half the MFMAs of the U phase and of the X^T U / U^T U phase;
//   * the X sub-tile and the old int8 U rows are prefetched one sub-tile ahead into registers;
//   * the Gauss-Seidel runs on ALL waves: wave w solves the sixteen rows whose a = x V it has just computed (no workgroup
//     barrier in between), four lanes per row, lane q owning the columns r = q mod 4.  From the second iteration on
//     every term of `uu @ bb` is an exact integer in fp32 (the host checks (R - 1) 64 mx^3 < 2^24: run_bcd, lrf_api.hip), so
//     the reference's ordered chain becomes: T[r] = sum of u_old[j] b[j][r] over the columns j > r still holding old values,
//     then column by column  u_r = project((a_r - T[r] + eps) / den_r)  by the owning lane, a quad broadcast (DPP
//     quad_perm), and T[r'] += u_r b[r][r'] for the later columns — eight independent fmas per lane and step instead of
//     thirty-one on one.  The quotient is num * (1 / den) with gs_row's tie test, repeated with the IEEE division when any
//     lane is too close to call.  Bit-identical to the ordered chain / the oracle.
//   The first iteration (float u_old: not exact) keeps the ordered chain, lane = row on one wave, with the row in registers
//   (mid_ordered_row).
#define MID_RP 32  // LDS row pitch of the a / u tiles: two rank tiles

template <int MODE>
struct MidLds {
    float Xs[64 * XS_LD];
    float a_p[64 * MID_RP]; // a = x V; quad path: column r of a row at (r & 3) * 8 + (r >> 2) (a lane's eight columns contiguous)
    float u_s[64 * MID_RP]; // u, natural column order
    float bu_p[32 * 32];    // bu_p[r][q * 8 + i] = b[r][4 i + q] if 4 i + q > r (both < R) else 0
    float bl_p[32 * 32];    // bl_p[j][q * 8 + i] = b[j][4 i + q] if 4 i + q < j (both < R) else 0
    float rden[32], den[32];
    float gt_l[32 * LRF_GTB_LD]; // the gt table rows of the ordered solve (first iteration / wide bounds)
    float wa_s[MODE == 1 ? 2 * 16 * 64 : 4];
};

template <int Q>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), Q * 0x55, 0xf, 0xf, true)); // quad_perm:[Q,Q,Q,Q]
}

// one Gauss-Seidel step of the quad solve: column R0 (compile time), owner lane q = R0 & 3, its slot R0 >> 2
template <int R0, bool FAST>
__device__ __forceinline__ void mid_gs_step(const MidLds<0>& L, int q, const float (&a)[8], float (&T)[8], float (&un)[8],
                                            const GsParams& gp, bool& unsafe)
{
    constexpr int QO = R0 & 3, IO = R0 >> 2;
    const float num = (a[IO] - T[IO]) + LRF_EPS;
    float val;
    if (FAST) {
        const float qt = num * L.rden[R0];
        const float nq = rintf(qt);
        const bool inside = fabsf(qt) < gp.flimit;
        unsafe |= (q == QO) && inside && !(fabsf(qt - nq) <= gp.fthr);
        val = inside ? nq : qt;
    } else {
        val = rintf(num / L.den[R0]);
    }
    val = fminf(fmaxf(val, gp.lo), gp.hi);
    const float vb = quad_bcast<QO>(val);
    un[IO] = (q == QO) ? vb : un[IO];
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(&L.bu_p[R0 * 32 + q * 8]);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(&L.bu_p[R0 * 32 + q * 8 + 4]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        T[i] = fmaf(vb, b0[i], T[i]);
        T[4 + i] = fmaf(vb, b1[i], T[4 + i]);
    }
}

template <bool FAST, int... Rs>
__device__ __forceinline__ bool mid_gs_steps(const MidLds<0>& L, int R, int q, const float (&a)[8], float (&T)[8], float (&un)[8],
                                             const GsParams& gp, std::integer_sequence<int, Rs...>)
{
    bool unsafe = false;
    ((Rs < R ? mid_gs_step<Rs, FAST>(L, q, a, T, un, gp, unsafe) : (void)0), ...); // R is wave-uniform: scalar branches
    return unsafe;
}


template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_bcd_mid(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                                 const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                                 const float* __restrict__ Wf, const float* __restrict__ Bf,
                                                 const float* __restrict__ U0, int8_t* __restrict__ U,
                                                 float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MidLds<MODE>& L = *reinterpret_cast<MidLds<MODE>*>(smem);
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
    const bool quad = (MODE == 0) && gp.exact_int; // the exact-integer solve on all waves

    // A operand of a^T = V^T X^T for rank tile nt: lane needs V[4s + lq][16 nt + li] at k-step s
    float va[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int s_ = 0; s_ < 16; s_++) va[nt][s_] = Vp[(4 * s_ + lq) * LRF_RPB + 16 * nt + li];
    if (MODE == 1) {
        for (int e = tid; e < 2 * 16 * 64; e += 256) {
            int nt = e >> 10, s_ = (e >> 6) & 15, l = e & 63;
            L.wa_s[e] = Wf[(long)bd.plane * 64 * LRF_RPB + (4 * s_ + (l >> 4)) * LRF_RPB + 16 * nt + (l & 15)];
        }
    }
    if (!quad)
        for (int e = tid; e < R * LRF_GTB_LD; e += 256) L.gt_l[e] = gt[e];
    if (MODE == 0) { // the symmetric b table, split into its strictly upper and strictly lower part, quad-permuted columns
        for (int e = tid; e < 32 * 32; e += 256) {
            const int r = e >> 5, p = e & 31, c = 4 * (p & 7) + (p >> 3); // position p = q * 8 + i holds column 4 i + q
            float b = 0.f;
            if (r < R && c < R && c != r) b = gt[c * LRF_GTB_LD + (r < c ? r : r - 1)]; // b[r][c] (gt row c lists b[j][c], j != c)
            L.bu_p[e] = (c > r) ? b : 0.f;
            L.bl_p[e] = (c < r) ? b : 0.f;
        }
        if (tid < 32) {
            const float d = (tid < R) ? gt[tid * LRF_GTB_LD + LRF_GTB_DEN] : 1.f;
            L.den[tid] = d;
            L.rden[tid] = 1.0f / d;
        }
    }

    // prefetch registers: the thread's four float4 of the next X sub-tile and up to eight old int8 U bytes
    f32x4 xq[4];
    int8_t upre[8];
    auto issue = [&](int t) {
        const int r0 = t * 64;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = i * 256 + tid;
            int row = r0 + (e >> 4);
            row = row < nrows ? row : nrows - 1; // clamped, not masked: rows past the end get u = 0 below
            xq[i] = *reinterpret_cast<const f32x4*>(Xp + (long)row * 64 + 4 * (e & 15));
        }
        if (MODE == 0) {
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int e = i * 256 + tid;
                upre[i] = Ub[(long)r0 * R + (e < lim ? e : lim - 1)];
            }
        }
    };
    const int invR = (65536 + R - 1) / R; // (e * invR) >> 16 == e / R for e < 64 * R

    f32x4 accP[2], accQ[2];
#pragma unroll
    for (int i = 0; i < 2; i++) { accP[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accQ[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    issue(0);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        __syncthreads(); // the previous sub-tile's X^T U phase has read Xs / u_s
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = i * 256 + tid, row = e >> 4, c4 = e & 15;
            float2* d = reinterpret_cast<float2*>(&L.Xs[row * XS_LD + 4 * c4]);
            d[0] = make_float2(xq[i][0], xq[i][1]);
            d[1] = make_float2(xq[i][2], xq[i][3]);
        }
        if (MODE == 0) {
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int e = i * 256 + tid;
                const int row = (e * invR) >> 16, r = e - row * R;
                if (e < lim) L.u_s[row * MID_RP + r] = (float)upre[i];
            }
        }
        if (MODE == 2) {
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
            for (int e = tid; e < lim; e += 256) {
                const int row = e / R, r = e - row * R;
                L.u_s[row * MID_RP + r] = U0[pd.u0_off + ((long)bd.row0 + r0) * R + e];
            }
        }
        issue(t + 1 < nsub ? t + 1 : t); // unconditional (the last one re-reads its own tile): exact wait counts
        __syncthreads();
        { // a^T tiles for rows 16*wave..+15
            f32x4 acc[2], accw[2];
#pragma unroll
            for (int i = 0; i < 2; i++) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accw[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            const float* xr = &L.Xs[(16 * wave + li) * XS_LD + lq];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float bx = xr[4 * s];
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[nt][s], bx, acc[nt], 0, 0, 0);
                    if (MODE == 1)
                        accw[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(L.wa_s[(nt * 16 + s) * 64 + lane], bx, accw[nt], 0, 0, 0);
                }
            }
            // D[i = 4 lq + reg (column 16 nt + i)][j = li (row)]
            float* ar = &L.a_p[(16 * wave + li) * MID_RP];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                if (quad) { // column r = 16 nt + 4 lq + reg -> position (r & 3) * 8 + (r >> 2) = reg * 8 + 4 nt + lq
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) ar[reg * 8 + 4 * nt + lq] = acc[nt][reg];
                } else {
                    *reinterpret_cast<f32x4*>(&ar[16 * nt + 4 * lq]) = acc[nt];
                }
                if (MODE == 1) *reinterpret_cast<f32x4*>(&L.u_s[(16 * wave + li) * MID_RP + 16 * nt + 4 * lq]) = accw[nt];
            }
        }
        if (quad) {
            // ---- Gauss-Seidel of this wave's own sixteen rows: only wave-local LDS traffic, no workgroup barrier
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int rl = 16 * wave + (lane >> 2), q = lane & 3;
            const MidLds<0>& L0 = *reinterpret_cast<const MidLds<0>*>(smem); // same layout up to wa_s
            float a[8], T0[8];
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&L.a_p[rl * MID_RP + q * 8 + i]);
                a[i] = v[0]; a[i + 1] = v[1]; a[i + 2] = v[2]; a[i + 3] = v[3];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) T0[i] = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < 32; j4 += 4) {
                if (j4 < R) { // wave-uniform; the old row four columns at a time (all of it at once costs 32 registers)
                    const f32x4 uo = *reinterpret_cast<const f32x4*>(&L.u_s[rl * MID_RP + j4]);
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int j = j4 + jj;
                        if (j >= 1 && j < R) {
                            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&L.bl_p[j * 32 + q * 8]);
                            const f32x4 b1 = *reinterpret_cast<const f32x4*>(&L.bl_p[j * 32 + q * 8 + 4]);
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                T0[i] = fmaf(uo[jj], b0[i], T0[i]);
                                T0[4 + i] = fmaf(uo[jj], b1[i], T0[4 + i]);
                            }
                        }
                    }
                }
            }
            float T[8], un[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { T[i] = T0[i]; un[i] = 0.f; }
            if (__any(mid_gs_steps<true>(L0, R, q, a, T, un, gp, std::make_integer_sequence<int, 32>{}))) {
#pragma unroll
                for (int i = 0; i < 8; i++) { T[i] = T0[i]; un[i] = 0.f; }
                mid_gs_steps<false>(L0, R, q, a, T, un, gp, std::make_integer_sequence<int, 32>{});
            }
            const bool live = r0 + rl < nrows;
#pragma unroll
            for (int i = 0; i < 8; i++) L.u_s[rl * MID_RP + 4 * i + q] = (live && 4 * i + q < R) ? un[i] : 0.f;
        } else {
            __syncthreads();
            if (wave == (t & 3)) { // ordered chain, lane = row (first iteration, or bounds too wide for the exact solve)
                const int row = r0 + lane;
                float* ur = &L.u_s[lane * MID_RP];
                if (row < nrows && pd.native_t2_u == 0) { // the row in registers: no memory latency inside the ordered chains
                    float a[32], u[32];
#pragma unroll
                    for (int j = 0; j < 32; j += 4) {
                        const f32x4 va4 = *reinterpret_cast<const f32x4*>(&L.a_p[lane * MID_RP + j]);
                        const f32x4 vu4 = *reinterpret_cast<const f32x4*>(&ur[j]);
#pragma unroll
                        for (int i = 0; i < 4; i++) { a[j + i] = va4[i]; u[j + i] = vu4[i]; }
                    }
                    mid_ordered_row(R, a, u, L.gt_l, gp.lo, gp.hi, std::make_integer_sequence<int, 32>{});
#pragma unroll
                    for (int j = 0; j < 32; j += 4)
                        *reinterpret_cast<f32x4*>(&ur[j]) = (f32x4){j < R ? u[j] : 0.f, j + 1 < R ? u[j + 1] : 0.f, j + 2 < R ? u[j + 2] : 0.f,
                                                                     j + 3 < R ? u[j + 3] : 0.f};
                } else if (row < nrows) { // ATen-native order (tiny matrices): the generic chain through LDS
                    const int K = R - 1;
                    for (int r = 0; r < R; r++) {
                        const float* bb = L.gt_l + r * LRF_GTB_LD;
                        const float term2 = gs_term2_generic(ur, r, bb, K, true);
                        const float num = (L.a_p[lane * MID_RP + r] - term2) + LRF_EPS;
                        const float val = rintf(num / bb[LRF_GTB_DEN]);
                        ur[r] = fminf(fmaxf(val, gp.lo), gp.hi);
                    }
                    for (int r = R; r < MID_RP; r++) ur[r] = 0.f;
                } else {
                    for (int r = 0; r < MID_RP; r++) ur[r] = 0.f;
                }
            }
        }
        __syncthreads();
        { // int8 U out, coalesced
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int e = i * 256 + tid;
                const int row = (e * invR) >> 16, r = e - row * R;
                if (e < lim) Ub[(long)r0 * R + e] = (int8_t)L.u_s[row * MID_RP + r];
            }
        }
        { // X^T U for columns 16*wave..+15 (both rank tiles); U^T U tile row `wave` (waves 0, 1)
            const float* xc = &L.Xs[lq * XS_LD + 16 * wave + li];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float px = xc[4 * s * XS_LD];
                const float* urow = &L.u_s[(4 * s + lq) * MID_RP];
                const float qa = urow[16 * (wave & 1) + li];
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const float ub = urow[16 * nt + li];
                    accP[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(px, ub, accP[nt], 0, 0, 0);
                    if (wave < 2) accQ[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, ub, accQ[nt], 0, 0, 0);
                }
            }
        }
    }
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RPB;
    float* Qp = Qpart + slot * LRF_RPB * LRF_RPB;
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            Pp[(16 * wave + 4 * lq + reg) * LRF_RPB + 16 * nt + li] = accP[nt][reg]; // D[i = X column][j = r]
            if (wave < 2) Qp[(16 * wave + 4 * lq + reg) * LRF_RPB + 16 * nt + li] = accQ[nt][reg]; // D[i = r][j = r']
        }
}

// V update for ranks 17..32 (k_vupdate's structure at rank pitch 64) with the sixty-four rows of V solved in registers (mid_ordered_row) — the ordered
// chain is mandatory here (u.mT @ u is far beyond the exact-integer range), but its R (R-1) terms per row no longer pay an
// LDS round trip each: 0.054 -> 0.027 ms per launch at ranks (20,10,10), 64 images.
__global__ __launch_bounds__(256) void k_vupdate_mid(const PlaneDesc* __restrict__ planes, const float* __restrict__ Ppart,
                                                     const float* __restrict__ Qpart, float* __restrict__ Vf,
                                                     float* __restrict__ Bf, int8_t* __restrict__ V8, float lo, float hi,
                                                     int write_i8, int plane0)
{
    const int pli = blockIdx.x + plane0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BigVLds& L = *reinterpret_cast<BigVLds*>(smem);
    const PlaneDesc pd = planes[pli];
    const int R = pd.R, tid = threadIdx.x;
    // a' = ((P0 + P1) + P2) + ... per element, b' likewise (block order); only the first 32 columns exist at these ranks
    // The eight elements of a thread advance together, eight blocks per round: 128 loads in flight per thread instead of 16
    // (the kernel is a chain of memory round trips: 27 -> 21 us per launch at 192 matrices); per element the order stays b = 0, 1, ...
    {
        constexpr int NE = 64 * 32 / 256, NB = 8;
        float acc[NE], q[NE];
#pragma unroll
        for (int m = 0; m < NE; m++) acc[m] = q[m] = 0.f;
        const float* P0 = Ppart + (long)pd.blk0 * 64 * LRF_RPB;
        const float* Q0 = Qpart + (long)pd.blk0 * LRF_RPB * LRF_RPB;
        for (int b0 = 0; b0 < pd.nblk; b0 += NB) {
            float v[NE][NB], w[NE][NB];
#pragma unroll
            for (int m = 0; m < NE; m++) {
                const int i2 = tid + 256 * m, i = (i2 >> 5) * LRF_RPB + (i2 & 31);
                const bool needq = (i2 >> 5) < 32; // b' is [R][R]
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    v[m][k] = (b0 + k < pd.nblk) ? P0[(long)(b0 + k) * 64 * LRF_RPB + i] : 0.f;
                    w[m][k] = (needq && b0 + k < pd.nblk) ? Q0[(long)(b0 + k) * LRF_RPB * LRF_RPB + i] : 0.f;
                }
            }
#pragma unroll
            for (int m = 0; m < NE; m++)
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if (b0 + k < pd.nblk) {
                        acc[m] = (b0 + k == 0) ? v[m][k] : acc[m] + v[m][k];
                        q[m] = (b0 + k == 0) ? w[m][k] : q[m] + w[m][k];
                    }
        }
#pragma unroll
        for (int m = 0; m < NE; m++) {
            const int i2 = tid + 256 * m, i = (i2 >> 5) * LRF_RPB + (i2 & 31);
            L.a_s[i] = acc[m];
            L.v_s[i] = Vf[(long)pli * 64 * LRF_RPB + i];
            const int j = i2 >> 5, r = i2 & 31; // b' = U^T U entry (j, r)
            if (j < R && r < R) {
                if (j == r) L.gt_s[r * LRF_GTB_LD + LRF_GTB_DEN] = (q[m] + 0.f) + LRF_EPS;
                else L.gt_s[r * LRF_GTB_LD + (j < r ? j : j - 1)] = q[m];
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        float a[32], v[32];
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            const f32x4 va4 = *reinterpret_cast<const f32x4*>(&L.a_s[tid * LRF_RPB + j]);
            const f32x4 vv4 = *reinterpret_cast<const f32x4*>(&L.v_s[tid * LRF_RPB + j]);
#pragma unroll
            for (int i = 0; i < 4; i++) { a[j + i] = va4[i]; v[j + i] = vv4[i]; }
        }
        mid_ordered_row(R, a, v, L.gt_s, lo, hi, std::make_integer_sequence<int, 32>{}); // (R - 1) * 64 >= 400: never the native order
        float* Vp = Vf + (long)pli * 64 * LRF_RPB + tid * LRF_RPB;
        int8_t* vo = V8 + pd.v_off + (long)tid * R;
#pragma unroll
        for (int r = 0; r < 32; r++)
            if (r < R) {
                L.v_s[tid * LRF_RPB + r] = v[r];
                Vp[r] = v[r];
                if (write_i8) vo[r] = (int8_t)v[r];
            }
    }
    __syncthreads();
    if (!write_i8) make_gtable_big(L.v_s, 64, R, Bf + (long)pli * LRF_GTB_STRIDE, tid, 256);
}
