#pragma once
// lrf_gram_kernels.hip — the exact Gram matrix G = X^T X of the 64-column path (input of the SVD initialisation,
// lrf/factorization/qmf.py:42-48), on the int8 matrix cores.  Included by lrf_api.hip after lrf_kernels.hip.
//
// Definition (oracle/lrf_oracle.c lrf_oracle_gram_exact): every element is placed on the fixed-point grid 2^(E-35),
// n = rint(x 2^(35-E)) with max|x| < 2^E, the integer sums S_ij = sum_m n_mi n_mj are accumulated EXACTLY and rounded once
// to fp64.  For the matrices qmf_encode forms (fp32 values that are 0 or in [0.114, 255.5]) the grid step holds every
// value exactly, so G is the exact Gram matrix of X.  Exact means order-free: rows may be summed in any grouping, by any
// number of workgroups, and the result is the oracle's bit for bit.
//
// n (|n| < 2^35) is cut into five 7-bit digits d_a (sign carried by every digit), so that
//     S_ij = sum_{a,b} 2^(7(a+b)) sum_m d_a[m,i] d_b[m,j]
// and each digit-pair sum is one chain of v_mfma_i32_16x16x64_i8 (64 rows per instruction, int32 accumulation: nine
// accumulators per 16 x 16 tile, one per weight a + b, exact for up to 26,000 rows).  One workgroup per chunk of
// LRF_GRAM_ROWS rows: wave t converts column tile t of each 64-row block to digit bytes in MFMA operand layout (lane
// (i, kq): column 16 t + i, rows 16 kq .. + 15 — the same bytes serve as A operand (X^T) and as B operand (X)), the four
// tiles meet in LDS, and each wave owns two or three of the ten upper-triangle tile pairs.  At the end of the chunk the
// nine weights of every element are folded into one 128-bit integer and written as the chunk's partial; k_init adds the
// partials of its matrix and rounds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lrf_device.h"

#include "lrf_internal.h"


#define LRF_GRAM_BITS 35

#ifndef LRF_GRAM_DEVICE_ONLY // (a second translation unit takes the device functions below without this kernel: lrf_planes_gram.hip)
// E with max|x| < 2^E per matrix, from the largest magnitude's bit pattern (oracle: lrf_oracle_gram_exponent)
__global__ __launch_bounds__(256) void k_gram_exponent(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                                       int* __restrict__ gexp)
{
    __shared__ unsigned red[4];
    const PlaneDesc pd = planes[blockIdx.x];
    const uint4* xp = reinterpret_cast<const uint4*>(X + pd.x_off);
    const long n4 = (long)pd.M * 16;
    unsigned mx = 0;
    for (long i = threadIdx.x; i < n4; i += 256) {
        const uint4 v = xp[i];
        mx = max(max(mx, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(red[0], red[1]), max(red[2], red[3]));
        gexp[blockIdx.x] = mx ? (int)(mx >> 23) - 126 : 0;
    }
}

#endif

// five signed 7-bit digits of n = clamp(rint(x * scale)), one per byte lane `b` of the packed operand dwords
__device__ __forceinline__ void gram_digits(float x, double scale, unsigned (&pk)[5], int b)
{
    const double lim = 34359738367.0; // 2^35 - 1
    double r = rint((double)x * scale);
    r = fmin(fmax(r, -lim), lim);
    const double hi = trunc(r * 4.76837158203125e-07); // 2^-21: |hi| < 2^14
    const double lo = fma(-hi, 2097152.0, r);           // exact, |lo| < 2^21, sign of r
    const int ih = (int)hi, il = (int)lo;
    const int ah = ih < 0 ? -ih : ih, al = il < 0 ? -il : il;
    int d[5] = {al & 127, (al >> 7) & 127, al >> 14, ah & 127, ah >> 7};
    const bool neg = r < 0.0;
#pragma unroll
    for (int a = 0; a < 5; a++) {
        const int v = neg ? -d[a] : d[a];
        pk[a] |= ((unsigned)v & 0xffu) << (8 * b);
    }
}

// The ten tile pairs over the four waves.  Wave w keeps its own column tile w as the B operand of all its MFMAs ("hub":
// five ds_read_b128 per block) and owns
//   slot 0: tile pair (w+1, w)      A = the five digits of tile w+1                         25 MFMAs
//   slot 1: tile pair (w, w)        A = B                                                   25 MFMAs, no loads
//   slot 2: HALF of the pair of tiles w and w+2 (mod 4), A = the five digits of the other tile: the digit pairs (a, b) with
//           a <= b — in waves 2 and 3 the a == b products take a zeroed A operand, because waves 0 and 1, which hold the same
//           two tiles the other way round, cover them —                                    15 MFMAs
// 65 MFMAs and 15 operand reads per block and wave (75 / 75 / 50 / 50 and 30 / 30 / 20 / 20 with whole pairs; on a SIMD the
// MFMA and the VALU time of its waves add up, so the slowest wave sets the pace).  The two halves of a split pair meet once
// per chunk: waves 2 / 3 pass their nine int32 weight sums, transposed, through LDS to waves 0 / 1.
// All waves run the same instructions on statically indexed accumulators (branches around MFMAs cost accumulator copies and
// spills); what differs is data: LDS addresses and the zeroed operand.
// A tile pair may come out in either orientation: G is symmetric and k_init fills both triangles (gram_pair_tiles).
__device__ __forceinline__ void gram_accumulate(const uint4* __restrict__ lbuf, int lane, int wave, i32x4 (&acc)[3][9])
{
    i32x4 A[5], Bv[5];
    const uint4* lb = lbuf + wave * 5 * 64 + lane;
    const uint4* la = lbuf + ((wave + 1) & 3) * 5 * 64 + lane;
#pragma unroll
    for (int a = 0; a < 5; a++) {
        const uint4 va = la[a * 64], vb = lb[a * 64];
        A[a] = (i32x4){(int)va.x, (int)va.y, (int)va.z, (int)va.w};
        Bv[a] = (i32x4){(int)vb.x, (int)vb.y, (int)vb.z, (int)vb.w};
    }
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) acc[0][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a], Bv[b], acc[0][a + b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) acc[1][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bv[a], Bv[b], acc[1][a + b], 0, 0, 0);
    // no scheduling barrier here: the operand reads of slot 2 move up under the MFMAs of slot 1 as far as the registers allow
    const uint4* lc = lbuf + ((wave + 2) & 3) * 5 * 64 + lane;
    const int keep = wave < 2 ? -1 : 0; // wave-uniform mask of the a == b products
#pragma unroll
    for (int a = 0; a < 5; a++) {
        const uint4 va = lc[a * 64];
        A[a] = (i32x4){(int)va.x, (int)va.y, (int)va.z, (int)va.w};
        const i32x4 Ad = (i32x4){A[a][0] & keep, A[a][1] & keep, A[a][2] & keep, A[a][3] & keep};
        acc[2][2 * a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ad, Bv[a], acc[2][2 * a], 0, 0, 0);
#pragma unroll
        for (int b = a + 1; b < 5; b++) acc[2][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a], Bv[b], acc[2][a + b], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// sum += (int128)v << sh  (sh <= 64), two's complement in (lo, hi)
__device__ __forceinline__ void add_shifted_i128(unsigned long long& lo, long long& hi, int v, int sh)
{
    const long long x = (long long)v;
    if (sh == 64) { hi += x; return; }
    const unsigned long long l = (unsigned long long)x << sh;
    const long long h = sh ? (x >> (64 - sh)) : (x >> 63);
    const unsigned long long nl = lo + l;
    hi += h + (nl < lo ? 1 : 0);
    lo = nl;
}

// The five digits of four values (one packed dword per digit) for the matrices qmf_encode forms: every element is 0 or in
// [2^-4, 2^8) and non-negative, so on the grid 2^-27 (E = 8) n = x 2^27 is an integer below 2^35, no rounding, no sign.
// Here the digits are BALANCED base-256 ones, d_k in [-128, 127]: n + 0x80808080 has the bytes d_k + 128, i.e. the bytes
// of (n + 0x80808080) ^ 0x80808080 read as int8 ARE the four low digits, and the fifth is the carry-adjusted top word
// (0..8) — no shifts or masks per digit, a 4 x 4 byte transpose (eight v_perm_b32) packs four values, and the weights become
// 2^(8(a+b)) (k_gram64's fold).  172 VALU instructions per 64-row block and wave instead of the 360 of 7-bit digits.
__device__ __forceinline__ void gram_digits4_planes(const float (&x)[4], unsigned (&pk)[5])
{
    unsigned xb[4], d4[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        // n = x 2^27 < 2^35 split without 64-bit shifts: the top word is trunc(x / 32) (0..7), the remainder x - 32 top < 32
        // is exact in fp32 and so is its product with 2^27, an integer below 2^32
        const float top = truncf(x[b] * 0x1p-5f);
        const float rem = fmaf(-32.f, top, x[b]);
        const unsigned lo = (unsigned)(rem * 0x1p27f), hi = (unsigned)top;
        const unsigned lob = lo + 0x80808080u;
        xb[b] = lob ^ 0x80808080u;
        d4[b] = hi + (lob < lo ? 1u : 0u);
    }
    // v_perm_b32(s0, s1, sel): byte selectors 0-3 pick from s1, 4-7 from s0
    const unsigned t0 = __builtin_amdgcn_perm(xb[1], xb[0], 0x05010400u); // x0.b0 x1.b0 x0.b1 x1.b1
    const unsigned t1 = __builtin_amdgcn_perm(xb[1], xb[0], 0x07030602u); // x0.b2 x1.b2 x0.b3 x1.b3
    const unsigned t2 = __builtin_amdgcn_perm(xb[3], xb[2], 0x05010400u);
    const unsigned t3 = __builtin_amdgcn_perm(xb[3], xb[2], 0x07030602u);
    pk[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); // x0.b0 x1.b0 x2.b0 x3.b0
    pk[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    pk[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    pk[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
    pk[4] = d4[0] | (d4[1] << 8) | (d4[2] << 16) | (d4[3] << 24);
}

// The epilogue of k_gram64: the halves of the split pairs meet, the nine weights fold into one 128-bit
// integer per element, the chunk's partial goes out as [pair][reg][lane].
template <bool PLANES>
__device__ __forceinline__ void gram_finish(i32x4 (&acc)[3][9], uint4* lds0, int lane, int wave, ulonglong2* __restrict__ out)
{
    // waves 2 / 3 hold tile (w-2, w) (rows i from tile w-2), waves 0 / 1 tile (w+2, w) of the same two tiles the other way
    // round: element [i][j] of one is element [j][i] of the other
    __syncthreads(); // every wave is done with the operand tiles
    int* xch = reinterpret_cast<int*>(lds0); // [half pair 0 / 1][weight][j][i]: 2 x 9 KB
    const int li16 = lane & 15, lq4 = lane >> 4;
    if (wave >= 2) {
#pragma unroll
        for (int w = 0; w < 9; w++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) xch[((wave - 2) * 9 + w) * 256 + li16 * 16 + 4 * lq4 + reg] = acc[2][w][reg];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int w = 0; w < 9; w++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) acc[2][w][reg] += xch[(wave * 9 + w) * 256 + (4 * lq4 + reg) * 16 + li16];
    }
    const int npairs = wave < 2 ? 3 : 2;
    const int pair0 = wave == 0 ? 0 : wave == 1 ? 3 : wave == 2 ? 6 : 8;
#pragma unroll
    for (int p = 0; p < 3; p++) {
        if (p < npairs) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                unsigned long long lo = 0;
                long long hi = 0;
#pragma unroll
                for (int w = 0; w < 9; w++) add_shifted_i128(lo, hi, acc[p][w][reg], (PLANES ? 8 : 7) * w); // digit base of the path
                out[(pair0 + p) * 256 + reg * 64 + lane] = make_ulonglong2(lo, (unsigned long long)hi);
            }
        }
    }
}

#ifdef LRF_GRAM_STAMPS // diagnostic build only (tools/dev_stamps_gram.py)
__device__ unsigned long long g_gram_stamps[8 * 16384];
__device__ __forceinline__ unsigned long long gram_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define GSTAMP(v) const unsigned long long v = gram_stamp()
#else
#define GSTAMP(v)
#endif

// PLANES: the matrices come from k_planes / k_planes16 (fixed_exp = 8, exact integer digit extraction)
// Tried on this kernel and not kept (all bit-identical, 0.18-0.19 ms either way): whole tile pairs per wave (75 / 75 / 50 / 50
// MFMAs, 100 operand reads per block) against the hub assignment of gram_accumulate; workgroups de-phased by their wave slot
// (s_sleep at entry); a software-pipelined form — the digits of block b + 1 cut into twelve pieces and placed between the
// MFMAs of block b by scheduling group barriers, two register sets of values, 250 VGPRs, two workgroups per CU —: 0.207 ms.
template <bool PLANES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_gram64(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                                const GramChunk* __restrict__ chunks, const int* __restrict__ gexp, int fixed_exp,
                                                ulonglong2* __restrict__ Gpart)
{
    __shared__ uint4 lds[2][4 * 5 * 64]; // [buffer][tile][digit][lane]: 40 KB
    const GramChunk ch = chunks[blockIdx.x];
    const PlaneDesc pd = planes[ch.plane];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int E = fixed_exp != LRF_GRAM_EXP_FROM_DATA ? fixed_exp : gexp[ch.plane];
    const double scale = scalbn(1.0, LRF_GRAM_BITS - E);
    const float* Xp = X + pd.x_off + (long)ch.row0 * 64 + 16 * wave + li;
    const int nrows = ch.nrows, nblk = (nrows + 63) >> 6;

    i32x4 acc[3][9];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int w = 0; w < 9; w++) acc[p][w] = (i32x4){0, 0, 0, 0};


    float vals[16];
    auto load_block = [&](int blk) {
        const float* bp = Xp + (long)(blk * 64 + 16 * kq) * 64; // one address per lane, the sixteen rows at immediate offsets
        if (blk * 64 + 64 <= nrows) {                            // wave-uniform
#pragma unroll
#ifndef LRF_GRAM_NO_LOADS
            for (int j = 0; j < 16; j++) vals[j] = bp[j * 64];
#else
            for (int j = 0; j < 16; j++) vals[j] = 1.5f * (float)(lane + j + blk); // ablation builds only (tools/dev_lib_kernels.py)
#endif
        } else { // the last block of a matrix: rows past its end count as zeros
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int row = blk * 64 + 16 * kq + j;
                const float v = Xp[(long)(row < nrows ? row : nrows - 1) * 64];
                vals[j] = (row < nrows) ? v : 0.f;
            }
        }
    };
    load_block(0);
#ifdef LRF_GRAM_STAMPS
    unsigned long long ga = 0, gb = 0, gc = 0, gd = 0;
    const unsigned long long g_begin = gram_stamp();
#endif
    for (int blk = 0; blk < nblk; blk++) {
        GSTAMP(g0);
#ifdef LRF_GRAM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        GSTAMP(g1);
        unsigned pk[5][4];
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) pk[a][q] = 0u;
        if constexpr (PLANES) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float x4[4] = {vals[4 * q], vals[4 * q + 1], vals[4 * q + 2], vals[4 * q + 3]};
                unsigned one[5];
                gram_digits4_planes(x4, one);
#pragma unroll
                for (int a = 0; a < 5; a++) pk[a][q] = one[a];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                unsigned one[5] = {0u, 0u, 0u, 0u, 0u};
                gram_digits(vals[j], scale, one, j & 3);
#pragma unroll
                for (int a = 0; a < 5; a++) pk[a][j >> 2] |= one[a];
            }
        }
        __builtin_amdgcn_sched_barrier(0); // keep the phases apart: interleaved they overflow the register file
        GSTAMP(g2);
        if (blk + 1 < nblk) load_block(blk + 1); // wave-uniform; lands under the MFMAs below
        uint4* lb = lds[blk & 1];
#pragma unroll
        for (int a = 0; a < 5; a++) lb[(wave * 5 + a) * 64 + lane] = make_uint4(pk[a][0], pk[a][1], pk[a][2], pk[a][3]);
        __syncthreads(); // one barrier per block: the other buffer is not written before every wave has passed this point again
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP(g3);
#ifndef LRF_GRAM_NO_MFMA
        gram_accumulate(lb, lane, wave, acc);
#else
        acc[0][0][0] += (int)lb[lane].x; // ablation builds only
#endif
#ifdef LRF_GRAM_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][8]));
        {
            GSTAMP(g4);
            ga += g1 - g0; gb += g2 - g1; gc += g3 - g2; gd += g4 - g3;
        }
#endif
    }
#ifdef LRF_GRAM_STAMPS
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_gram_stamps + 8 * (4 * blockIdx.x + wave);
        o[0] = gram_stamp() - g_begin; o[1] = ga; o[2] = gb; o[3] = gc; o[4] = gd;
    }
#endif
    gram_finish<PLANES>(acc, lds[0], lane, wave, Gpart + (long)ch.slot * LRF_GRAM_SLOT);
}

// signed 128-bit integer (two's complement lo, hi) -> fp64, round to nearest even (oracle: u128_to_double_rne)
__device__ __forceinline__ double i128_to_double_rne(unsigned long long lo, long long hi)
{
    const bool neg = hi < 0;
    unsigned long long uhi = (unsigned long long)hi;
    if (neg) {
        lo = ~lo + 1ull;
        uhi = ~uhi + (lo == 0ull ? 1ull : 0ull);
    }
    if (uhi == 0ull && lo == 0ull) return 0.0;
    const int nbits = uhi ? 128 - __clzll((long long)uhi) : 64 - __clzll((long long)lo);
    double v;
    if (nbits <= 53) {
        v = (double)lo;
    } else {
        const int sh = nbits - 53; // 1..75
        unsigned long long mant = sh >= 64 ? (uhi >> (sh - 64)) : ((lo >> sh) | (uhi << (64 - sh)));
        const int rb = sh - 1; // position of the round bit
        const bool round_bit = rb >= 64 ? ((uhi >> (rb - 64)) & 1ull) : ((lo >> rb) & 1ull);
        bool sticky;
        if (rb >= 64) sticky = lo != 0ull || (uhi & ((1ull << (rb - 64)) - 1ull)) != 0ull;
        else sticky = (lo & ((1ull << rb) - 1ull)) != 0ull;
        if (round_bit && (sticky || (mant & 1ull))) mant++;
        v = scalbn((double)mant, sh);
    }
    return neg ? -v : v;
}

// pair id -> (tile row, tile column)
__device__ __forceinline__ void gram_pair_tiles(int p, int& ti, int& tj)
{
    // pair slots of k_gram64 (gram_accumulate): wave 0: 0 1 2, wave 1: 3 4 5, wave 2: 6 7, wave 3: 8 9; rows from tile TI
    const int TI[10] = {1, 0, 2, 2, 1, 3, 3, 2, 0, 3}, TJ[10] = {0, 0, 0, 1, 1, 1, 2, 2, 3, 3};
    ti = TI[p];
    tj = TJ[p];
}
