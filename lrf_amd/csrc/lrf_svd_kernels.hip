#pragma once
// lrf_svd_kernels.hip — the SVD baseline codec (lrf.svd_encode / svd_decode, default RGB branch) on gfx950.
//
// Reference: lrf/compression/svd.py:156-193 (encode), :310-326 (decode), lrf/compression/utils.py:185-243
// (quantize / dequantize).  X is [M, 192] (three 8x8 colour patches per row), so the 64-wide kernels of the QMF
// path do not apply; the top-R singular pairs come from the Gram matrix (192 x 192: exact on the int8 matrix cores for the
// uint8-valued matrices, k_gram192_u8, or fp64, k_gram_blk below) through the
// eigen-solver of the any-shape path (k_any_eig<1>, lrf_anyshape_kernels.hip: Householder tridiagonalisation, multisection,
// twisted factorisation, Gram-Schmidt, back-transformation) and its ordered product (k_any_prod) for u = X w.  Parity for
// this path against the reference is by tolerance (LAPACK's SVD there); against the oracle, which
// restates this solver operation for operation (oracle/lrf_oracle_any.c), the encoder is byte for byte.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lrf_device.h"

// ---- "c (h p) (w q) -> (h w) (c p q)" with reflect padding: one workgroup per (patch row, image) ----
// T = float: the matrix the factorisation kernels take.  T = uint8_t (round 3, svd_encode): the same matrix as bytes — a quarter
// of the traffic for the three passes of svd_encode over it (this one, the exact Gram matrix, u = X w).
template <typename T>
__global__ __launch_bounds__(256) void k_patchify_rgb(const uint8_t* __restrict__ rgb, int H, int W, int top, int left,
                                                      int nw, long img_floats, T* __restrict__ X)
{
    const int hh = blockIdx.x;
    const uint8_t* img = rgb + (long)blockIdx.y * 3 * H * W;
    T* Xp = X + (long)blockIdx.y * img_floats + (long)hh * nw * 192;
    if constexpr (sizeof(T) == 1) {
        // bytes: an item is one 8-pixel patch row (ww, c, a) — one 8-byte load, one 8-byte store, the 24 items of a patch
        // 192 contiguous bytes of X (four pixels per item: 0.165 ms per 256 images, 3.7 TB/s)
        for (int it = threadIdx.x; it < nw * 24; it += 256) {
            const int ww = it / 24, rem = it - ww * 24;
            const int c = rem >> 3, a = rem & 7;
            const int y = reflect_idx(hh * 8 + a - top, H);
            const int x0 = ww * 8 - left;
            const uint8_t* rowp = img + (long)c * H * W + (long)y * W;
            uint64_t w8;
            if (x0 >= 0 && x0 + 7 < W) { // no reflection inside these eight pixels: one (unaligned) 8-byte word
                typedef uint64_t __attribute__((aligned(1))) u64u;
                w8 = *reinterpret_cast<const u64u*>(rowp + x0);
            } else {
                w8 = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) w8 |= (uint64_t)rowp[reflect_idx(x0 + i, W)] << (8 * i);
            }
            *reinterpret_cast<uint64_t*>(Xp + ww * 192 + c * 64 + a * 8) = w8;
        }
    } else {
        for (int it = threadIdx.x; it < nw * 48; it += 256) {
            const int ww = it / 48, rem = it - ww * 48;
            const int c = rem >> 4, a = (rem >> 1) & 7, b4 = (rem & 1) * 4;
            const int y = reflect_idx(hh * 8 + a - top, H);
            f32x4 out;
            const int x0 = ww * 8 + b4 - left;
            const uint8_t* rowp = img + (long)c * H * W + (long)y * W;
            uint32_t w4;
            if (x0 >= 0 && x0 + 3 < W) { // no reflection inside these four pixels: one (unaligned) word instead of four byte loads
                typedef uint32_t __attribute__((aligned(1))) u32u;
                w4 = *reinterpret_cast<const u32u*>(rowp + x0);
            } else {
                w4 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) w4 |= (uint32_t)rowp[reflect_idx(x0 + i, W)] << (8 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) out[i] = (float)((w4 >> (8 * i)) & 255u);
            *reinterpret_cast<f32x4*>(Xp + ww * 192 + c * 64 + a * 8 + b4) = out;
        }
    }
}

// ---- fp64 Gram block G[cg][cg'] = X[:, 64cg..]^T X[:, 64cg'..] with MFMA f64; grid (block pair, matrix) ----
__global__ __launch_bounds__(256) void k_gram_blk(const float* __restrict__ X, long x_stride, int M, int N, int nc,
                                                  double* __restrict__ G)
{
    __shared__ double Gs[64 * 64];
    int pair = blockIdx.x, cg = 0, cgp = 0; // upper-triangular pair index -> (cg, cg')
    for (int a = 0, p = 0; a < nc; a++)
        for (int b = a; b < nc; b++, p++)
            if (p == pair) { cg = a; cgp = b; }
    const float* Xp = X + (long)blockIdx.y * x_stride;
    double* Gp = G + (long)blockIdx.y * N * N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    f64x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int nsteps = (M + 3) >> 2;
    auto load_step = [&](int s, f32x4& xa, f32x4& xb) __attribute__((always_inline)) {
        const int row = 4 * s + lq;
        xa = (f32x4){0.f, 0.f, 0.f, 0.f};
        xb = xa;
        if (s < nsteps && row < M) {
            xa = *reinterpret_cast<const f32x4*>(Xp + (long)row * N + 64 * cg + 4 * li);
            xb = *reinterpret_cast<const f32x4*>(Xp + (long)row * N + 64 * cgp + 4 * li);
        }
    };
    // two steps in flight: the loads of the next two steps are issued before the sixteen MFMAs of the current one (a step
    // that waits for its own loads runs at a third of the fp64 MFMA rate)
    f32x4 xa0, xb0, xa1, xb1;
    load_step(wave, xa0, xb0);
    load_step(wave + 4, xa1, xb1);
    for (int s = wave; s < nsteps; s += 4) {
        const f32x4 xa = xa0, xb = xb0;
        xa0 = xa1;
        xb0 = xb1;
        load_step(s + 8, xa1, xb1);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++)
                acc[4 * t + u] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xa[t], (double)xb[u], acc[4 * t + u], 0, 0, 0);
    }
    for (int w = 0; w < 4; w++) { // sum the four waves' row subsets in wave order
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        int gi = 4 * (lq + 4 * reg) + t, gj = 4 * li + u;
                        double v = acc[4 * t + u][reg];
                        if (w) v = Gs[gi * 64 + gj] + v;
                        Gs[gi * 64 + gj] = v;
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < 64 * 64; i += 256) {
        int gi = i >> 6, gj = i & 63;
        double v = Gs[i];
        Gp[(long)(64 * cg + gi) * N + 64 * cgp + gj] = v;
        if (cg != cgp) Gp[(long)(64 * cgp + gj) * N + 64 * cg + gi] = v;
    }
}

// ---- exact Gram matrix of the [M,192] matrices that hold uint8 pixels (k_patchify_rgb: every entry an integer 0..255) ----
// x = a + 128 with a in [-128, 127] one int8 digit: S_ij = sum_m a_mi a_mj on the int8 matrix cores
// (v_mfma_i32_16x16x64_i8: 64 rows per instruction, where the fp64 MFMA of k_gram_blk takes 4 rows in four times the cycles),
// column sums s_i on the VALU, and G_ij = S_ij + 128 (s_i + s_j) + 128^2 M — exact integers (< 2^53), so G is the exact Gram
// matrix and does not depend on how the rows are grouped.  One workgroup per (1536-row chunk, matrix): the twelve 16-column
// tiles of a 64-row block meet in LDS as int8 in MFMA operand layout (double-buffered, one barrier per block), and wave w
// accumulates the tile products (row tiles 0 .. jt) x (its own three column tiles jt) — the upper triangle, the same code in
// every wave — plus the column sums of its tiles as products with a tile of ones.  k_gram192_fold adds the chunks' int32
// partials and the offset terms.  256 x [6144,192]: 2.49 ms (k_gram_blk, fp64 MFMA at 39 % of its peak) -> 0.8 ms with strided dword loads -> see DESIGN.md.
#define LRF_G192_ROWS 1536
template <typename T> // float: uint8-valued floats; uint8_t: the bytes themselves (k_patchify_rgb<uint8_t>)
__global__ __launch_bounds__(256) void k_gram192_u8(const T* __restrict__ X, long x_stride, int M, int* __restrict__ P, int nchunks)
{
    __shared__ uint4 lds[2][12 * 64]; // [buffer][tile][lane = 16 kq + li]: rows 16 kq .. + 15 of column 16 tile + li; 24 KB
    const int chunk = blockIdx.x;
    const T* Xp = X + (long)blockIdx.y * x_stride;
    int* Pp = P + ((long)blockIdx.y * nchunks + chunk) * (192 * 192 + 192);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int row_lo = chunk * LRF_G192_ROWS, row_hi = min(M, row_lo + LRF_G192_ROWS);
    const int nblk = (row_hi - row_lo + 63) >> 6;
    // The upper triangle only (round 4; k_gram192_fold mirrors it): wave w owns the column tiles jt = w, 7 - w, 8 + w and for
    // each the row tiles 0 .. jt — as straight-line code the same in every wave: 4 + 8 + 12 products (the ones with a row
    // tile beyond jt are computed and dropped), 24 MFMAs per 64-row block and wave instead of 36, 96 accumulator registers
    // instead of 144 (three waves per SIMD instead of two).
    const int jt0 = wave, jt1 = 7 - wave, jt2 = 8 + wave;
    i32x4 acc0[4], acc1[8], acc2[12], accs[3];
#pragma unroll
    for (int j = 0; j < 3; j++) accs[j] = (i32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) acc0[i] = (i32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) acc1[i] = (i32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 12; i++) acc2[i] = (i32x4){0, 0, 0, 0};
    // A 64 x 192 block is loaded as float4s along the rows (768 contiguous bytes per row: the strided dword loads of the
    // operand layout ran at 1.5 TB/s): item = (row group g of 4 rows, column group c4 of 4 columns), 16 x 48 items, three
    // per thread; its 4 x 4 values become int8, are transposed in registers (v_perm_b32) and go to LDS as four dwords, each
    // the four rows of one column — so that an operand (sixteen rows of a column) is one ds_read_b128.
    constexpr bool BYTES = sizeof(T) == 1;
    f32x4 vals[BYTES ? 1 : 3][BYTES ? 1 : 4];
    unsigned valb[BYTES ? 3 : 1][BYTES ? 4 : 1]; // bytes: the four columns of a row as one dword
    auto load_block = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int item = tid + 256 * it, g = item / 48, c4 = item - 48 * g;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = row_lo + blk * 64 + 4 * g + r;
                const long off = (long)(row < row_hi ? row : row_hi - 1) * 192 + 4 * c4;
                if constexpr (BYTES) {
                    const unsigned v = *reinterpret_cast<const unsigned*>(Xp + off);
                    valb[it][r] = (row < row_hi) ? v : 0x80808080u; // rows past the end: a = 0
                } else {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Xp + off);
                    vals[it][r] = (row < row_hi) ? v : (f32x4){128.f, 128.f, 128.f, 128.f};
                }
            }
        }
    };
    load_block(0);
    for (int blk = 0; blk < nblk; blk++) {
        unsigned* lw = reinterpret_cast<unsigned*>(lds[blk & 1]);
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int item = tid + 256 * it, g = item / 48, c4 = item - 48 * g;
            unsigned rw[4]; // row r: its four columns as bytes
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if constexpr (BYTES) {
                    rw[r] = valb[it][r] ^ 0x80808080u; // x - 128 as int8, four at a time
                } else {
                    const f32x4 v = vals[it][r];
                    rw[r] = ((unsigned)((int)v[0] - 128) & 255u) | (((unsigned)((int)v[1] - 128) & 255u) << 8) |
                            (((unsigned)((int)v[2] - 128) & 255u) << 16) | (((unsigned)((int)v[3] - 128) & 255u) << 24);
                }
            }
            // 4 x 4 byte transpose: cw[i] = column i, its four rows
            const unsigned t0 = __builtin_amdgcn_perm(rw[1], rw[0], 0x05010400u), t1 = __builtin_amdgcn_perm(rw[1], rw[0], 0x07030602u);
            const unsigned t2 = __builtin_amdgcn_perm(rw[3], rw[2], 0x05010400u), t3 = __builtin_amdgcn_perm(rw[3], rw[2], 0x07030602u);
            const unsigned cw[4] = {__builtin_amdgcn_perm(t2, t0, 0x05040100u), __builtin_amdgcn_perm(t2, t0, 0x07060302u),
                                    __builtin_amdgcn_perm(t3, t1, 0x05040100u), __builtin_amdgcn_perm(t3, t1, 0x07060302u)};
            const int tile = c4 >> 2;
#pragma unroll
            for (int i = 0; i < 4; i++) lw[(tile * 64 + (g >> 2) * 16 + 4 * (c4 & 3) + i) * 4 + (g & 3)] = cw[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < nblk) load_block(blk + 1); // lands under the MFMAs below
        __syncthreads(); // one barrier per block: the other buffer is not written before every wave has passed this point again
        const uint4* lb = lds[blk & 1];
        i32x4 Bv[3];
        const i32x4 ones = (i32x4){0x01010101, 0x01010101, 0x01010101, 0x01010101};
        const int jts[3] = {jt0, jt1, jt2};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint4 x = lb[jts[j] * 64 + lane];
            Bv[j] = (i32x4){(int)x.x, (int)x.y, (int)x.z, (int)x.w};
            accs[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bv[j], ones, accs[j], 0, 0, 0); // column sums of tile jts[j] (every column of D)
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint4 x = lb[i * 64 + lane];
            const i32x4 Av = (i32x4){(int)x.x, (int)x.y, (int)x.z, (int)x.w};
            if (i < 4) acc0[i < 4 ? i : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bv[0], acc0[i < 4 ? i : 0], 0, 0, 0);
            if (i < 8) acc1[i < 8 ? i : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bv[1], acc1[i < 8 ? i : 0], 0, 0, 0);
            acc2[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Av, Bv[2], acc2[i], 0, 0, 0);
        }
    }
    // D[r][c] of tile product (i, jt): rows r = 4 (lane >> 4) + reg of tile i, column c = lane & 15 of tile jt; i <= jt only
    auto store_tile = [&](const i32x4& a, int i, int jt) __attribute__((always_inline)) {
        if (i <= jt) { // wave-uniform
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Pp[(16 * i + 4 * kq + reg) * 192 + 16 * jt + li] = a[reg];
        }
    };
#pragma unroll
    for (int i = 0; i < 4; i++) store_tile(acc0[i], i, jt0);
#pragma unroll
    for (int i = 0; i < 8; i++) store_tile(acc1[i], i, jt1);
#pragma unroll
    for (int i = 0; i < 12; i++) store_tile(acc2[i], i, jt2);
    // column sums: D[r][c] = sum of column r of tile jt for every c; the lanes with c = 0 write them
    if (li == 0) {
        const int jts[3] = {jt0, jt1, jt2};
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Pp[192 * 192 + 16 * jts[j] + 4 * kq + reg] = accs[j][reg];
    }
}

// One workgroup per 16 x 16 tile on or above the diagonal (78 of them; the partials hold those tiles only) and matrix: the
// chunks' int32 partials and the offset terms, then the tile and — through LDS — its mirror image, both with row-contiguous
// stores.  grid (78, B).
__global__ __launch_bounds__(256) void k_gram192_fold(const int* __restrict__ P, int nchunks, int M, double* __restrict__ G)
{
    __shared__ double tile[16][17];
    const int* Pp = P + (long)blockIdx.y * nchunks * (192 * 192 + 192);
    int it = 0, jt = 0; // blockIdx.x -> (it <= jt): row it of the triangle starts at it * 12 - it (it - 1) / 2
    {
        int t = blockIdx.x;
        while (t >= 12 - it) { t -= 12 - it; it++; }
        jt = it + t;
    }
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int i = 16 * it + r, j = 16 * jt + c, e = i * 192 + j;
    long long S = 0, si = 0, sj = 0;
    constexpr long PS = 192 * 192 + 192;
    int ch = 0;
    for (; ch + 4 <= nchunks; ch += 4) { // (integer sums: any order) the loads of four chunks in flight
        const int* pc = Pp + (long)ch * PS;
        const int a0 = pc[e], a1 = pc[PS + e], a2 = pc[2 * PS + e], a3 = pc[3 * PS + e];
        const int b0 = pc[192 * 192 + i], b1 = pc[PS + 192 * 192 + i], b2 = pc[2 * PS + 192 * 192 + i], b3 = pc[3 * PS + 192 * 192 + i];
        const int d0 = pc[192 * 192 + j], d1 = pc[PS + 192 * 192 + j], d2 = pc[2 * PS + 192 * 192 + j], d3 = pc[3 * PS + 192 * 192 + j];
        S += ((long long)a0 + a1) + ((long long)a2 + a3);
        si += ((long long)b0 + b1) + ((long long)b2 + b3);
        sj += ((long long)d0 + d1) + ((long long)d2 + d3);
    }
    for (; ch < nchunks; ch++) {
        const int* pc = Pp + (long)ch * PS;
        S += pc[e];
        si += pc[192 * 192 + i];
        sj += pc[192 * 192 + j];
    }
    const double v = (double)(S + 128 * (si + sj) + 16384ll * M);
    double* Gb = G + (long)blockIdx.y * 192 * 192;
    Gb[e] = v;
    if (it != jt) { // (block-uniform)
        tile[r][c] = v;
        __syncthreads();
        Gb[(16 * jt + r) * 192 + 16 * it + c] = tile[c][r];
    }
}

// ---- u = X w for the byte matrix [M,192] and R <= 8 columns (svd_encode): U[m][r] = the k-ordered fp32 fma chain over
// k = 0..191 of x[m][k] w[k][r] — one block of the reference's K-blocking, the chain k_any_prod's MFMA tiles compute — with
// lane = row.  Round 4: the table w lives in VGPRs (register t of column r holds w[16 t + (lane & 15)][r], 12 R registers)
// and reaches the VALU through the DPP row_newbcast operand of v_fmac_f32 (k_bcd_w's device): one instruction per product
// and no LDS at all.  Round 3's form read w[k][0..RP) from an LDS table by broadcast ds_read_b128s, RP = 4 or 8: 384 KB of
// LDS reads per 12 KB of rows and 8 products per k where 5 are wanted — 0.268 ms for 256 x [6144,192] x [192,5], five times
// the bytes' time.  A wave walks LRF_PROD192_GROUPS groups of 64 rows (the table is loaded once per wave); a thread loads its
// row's 192 bytes as twelve 16-byte loads, the next row group's first piece requested under the arithmetic.
// grid (ceil(M / (256 LRF_PROD192_GROUPS)), B), 256 threads.
#define LRF_PROD192_GROUPS 4
template <int J, int R>
__device__ __forceinline__ void prod192_step(float (&acc)[R], const float (&wt)[R], float x)
{
#pragma unroll
    for (int r = 0; r < R; r++) fmac_bc16<J>(acc[r], wt[r], x);
}
template <int R>
__global__ __launch_bounds__(256) void k_prod192_u8(const uint8_t* __restrict__ X, long x_stride, int M, const float* __restrict__ Wn,
                                                    float* __restrict__ Uf)
{
    const int lane = threadIdx.x & 63, li = lane & 15, wave = threadIdx.x >> 6;
    const float* Wb = Wn + (long)blockIdx.y * 192 * R;
    float wt[12][R];
#pragma unroll
    for (int t = 0; t < 12; t++)
#pragma unroll
        for (int r = 0; r < R; r++) wt[t][r] = Wb[(16 * t + li) * R + r];
    const uint8_t* Xb = X + (long)blockIdx.y * x_stride;
    const int g0 = (blockIdx.x * 4 + wave) * LRF_PROD192_GROUPS; // first 64-row group of this wave
    uint4 xq[12], xn[12];
    auto load_group = [&](int g, uint4 (&q)[12]) __attribute__((always_inline)) {
        const int m = (g0 + g) * 64 + lane;
        const uint8_t* xr = Xb + (long)(m < M ? m : M - 1) * 192;
#pragma unroll
        for (int t = 0; t < 12; t++) q[t] = *reinterpret_cast<const uint4*>(xr + 16 * t);
    };
    if (g0 * 64 < M) load_group(0, xq);
    for (int g = 0; g < LRF_PROD192_GROUPS; g++) {
        const int m = (g0 + g) * 64 + lane;
        if ((g0 + g) * 64 >= M) break; // wave-uniform
        const bool more = g + 1 < LRF_PROD192_GROUPS && (g0 + g + 1) * 64 < M; // wave-uniform
        if (more) load_group(g + 1, xn); // the next group's rows are requested before this group's arithmetic
        else {
#pragma unroll
            for (int t = 0; t < 12; t++) xn[t] = make_uint4(0u, 0u, 0u, 0u);
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 12; t++) {
            const unsigned w4[4] = {xq[t].x, xq[t].y, xq[t].z, xq[t].w};
#define LRF_P192(J) prod192_step<J, R>(acc, wt[t], (float)((w4[(J) >> 2] >> (8 * ((J) & 3))) & 255u))
            LRF_P192(0); LRF_P192(1); LRF_P192(2); LRF_P192(3); LRF_P192(4); LRF_P192(5); LRF_P192(6); LRF_P192(7);
            LRF_P192(8); LRF_P192(9); LRF_P192(10); LRF_P192(11); LRF_P192(12); LRF_P192(13); LRF_P192(14); LRF_P192(15);
#undef LRF_P192
        }
        if (m < M) {
            float* uo = Uf + ((long)blockIdx.y * M + m) * R;
#pragma unroll
            for (int r = 0; r < R; r++) uo[r] = acc[r];
        }
#pragma unroll
        for (int t = 0; t < 12; t++) xq[t] = xn[t];
    }
}

// ---- reductions shared with the eigen-solver of the any-shape path (k_any_eig, lrf_anyshape_kernels.hip) ----
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double* part, int tid)
{
    v = wave_sum(v);
    if ((tid & 63) == 0) part[tid >> 6] = v;
    __syncthreads();
    double r = ((part[0] + part[1]) + part[2]) + part[3];
    __syncthreads();
    return r;
}

// ---- quantize(tensor, uint8): lrf/compression/utils.py:185-220.  One workgroup per tensor finds min / max ----
__global__ __launch_bounds__(256) void k_minmax(const float* __restrict__ T, long stride, long n, float* __restrict__ mm /*[nb][2]*/)
{
    __shared__ float smin[4], smax[4];
    const float* t = T + (long)blockIdx.x * stride;
    float mn = 3.4e38f, mx = -3.4e38f;
    // (min / max do not depend on the order.)  16-byte loads, four of them in flight per thread, where the tensor allows: one
    // workgroup walking 30720 floats with a 4-byte load per trip took 50 us of svd_encode's step (a chain of memory round trips)
    long i0 = 0;
    if (((reinterpret_cast<uintptr_t>(t) & 15) == 0)) {
        const float4* t4 = reinterpret_cast<const float4*>(t);
        const long n4 = n >> 2;
        long q = threadIdx.x;
        for (; q + 768 < n4; q += 1024) {
            const float4 a = t4[q], b = t4[q + 256], c = t4[q + 512], d = t4[q + 768];
            mn = fminf(fminf(fminf(mn, fminf(a.x, a.y)), fminf(fminf(a.z, a.w), fminf(b.x, b.y))), fminf(fminf(b.z, b.w), fminf(fminf(c.x, c.y), fminf(c.z, c.w))));
            mn = fminf(mn, fminf(fminf(d.x, d.y), fminf(d.z, d.w)));
            mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(a.x, a.y)), fmaxf(fmaxf(a.z, a.w), fmaxf(b.x, b.y))), fmaxf(fmaxf(b.z, b.w), fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w))));
            mx = fmaxf(mx, fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w)));
        }
        for (; q < n4; q += 256) {
            const float4 a = t4[q];
            mn = fminf(mn, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
            mx = fmaxf(mx, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        }
        i0 = n4 << 2;
    }
    for (long i = i0 + threadIdx.x; i < n; i += 256) {
        float v = t[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mm[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

// q = clamp((t - min) / scale + 0, 0, 255) truncated; scale = (max - min) / 255; also records (scale, min)
__global__ __launch_bounds__(256) void k_quantize_u8(const float* __restrict__ T, long stride, long n, const float* __restrict__ mm,
                                                     uint8_t* __restrict__ Q, float* __restrict__ qp /*[nb][qp_ld]*/, int qp_ld, int qp_off)
{
    const float mn = mm[2 * blockIdx.y], mx = mm[2 * blockIdx.y + 1];
    const float scale = (mx - mn) / 255.f;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { qp[(long)blockIdx.y * qp_ld + qp_off] = scale; qp[(long)blockIdx.y * qp_ld + qp_off + 1] = mn; }
    if (i >= n) return;
    float v = (T[(long)blockIdx.y * stride + i] - mn) / scale + 0.f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    Q[(long)blockIdx.y * stride + i] = (uint8_t)v;
}

// ---- svd_decode, RGB branch: dequantize, u @ v.mT, depatchify, unpad, to_dtype(uint8); 4 pixels per thread ----
__global__ __launch_bounds__(256) void k_svd_decode(const uint8_t* __restrict__ U, const uint8_t* __restrict__ V, int H, int W,
                                                    int top, int left, int nw, int M, int R, const float* __restrict__ qp /*[B][6]*/,
                                                    uint8_t* __restrict__ rgb)
{
    int w4 = (W + 3) >> 2;
    long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long)3 * H * w4) return;
    int c = (int)(o / ((long)H * w4));
    long rem = o - (long)c * H * w4;
    int y = (int)(rem / w4), x0 = (int)(rem - (long)y * w4) * 4;
    const float* q = qp + (long)blockIdx.y * 6; // scale_u, min_u, qmin_u, scale_v, min_v, qmin_v
    const uint8_t* Ub = U + (long)blockIdx.y * M * R;
    const uint8_t* Vb = V + (long)blockIdx.y * 192 * R;
    uint8_t* out = rgb + (long)blockIdx.y * 3 * H * W + (long)c * H * W + (long)y * W;
    int yy = y + top;
    for (int i = 0; i < 4; i++) {
        int x = x0 + i;
        if (x >= W) break;
        int xx = x + left;
        int m = (yy >> 3) * nw + (xx >> 3), n = c * 64 + (yy & 7) * 8 + (xx & 7);
        float acc = 0.f;
        for (int r = 0; r < R; r++) {
            float uf = ((float)Ub[(long)m * R + r] - q[2]) * q[0] + q[1]; // dequantize: (q - q.min()) * scale + min
            float vf = ((float)Vb[n * R + r] - q[5]) * q[3] + q[4];
            acc = fmaf(uf, vf, acc);
        }
        acc = fminf(fmaxf(acc, 0.f), 255.f);
        out[x] = (uint8_t)acc;
    }
}

// ---- the RGB colour-space branch of qmf_encode (lrf/compression/qmf.py:164-187, 311-323) shares this file's [M, 192] matrices:
// k_patchify_rgb forms them, the any-shape kernels factorise them (a dedicated VALU kernel set was half as fast), and:
// qmf_decode, RGB colour-space branch (qmf.py:311-323): u @ v.mT (exact integers), depatchify, unpad,
// to_dtype(uint8) = clamp + truncate; one thread per pixel
__global__ __launch_bounds__(256) void k_qmf_decode_rgbspace(const int8_t* __restrict__ U, const int8_t* __restrict__ V, int H, int W,
                                                             int top, int left, int nw, int M, int R, uint8_t* __restrict__ rgb)
{
    const long n = 3L * H * W;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e / ((long)H * W));
    const int rem = (int)(e - (long)c * H * W);
    const int y = rem / W, x = rem - y * W;
    const int yy = y + top, xx = x + left;
    const int m = (yy >> 3) * nw + (xx >> 3), col = c * 64 + (yy & 7) * 8 + (xx & 7);
    const int8_t* u = U + (long)blockIdx.y * M * R + (long)m * R;
    const int8_t* v = V + (long)blockIdx.y * 192 * R + (long)col * R;
    float acc = 0.f;
    for (int r = 0; r < R; r++) acc = fmaf((float)u[r], (float)v[r], acc); // exact: |sum| < 2^24
    acc = fminf(fmaxf(acc, 0.f), 255.f);
    rgb[(long)blockIdx.y * n + e] = (uint8_t)acc;
}

// The same branch for any patch size (qmf.py:164-171: X [M, 3 p q], columns in (c, a, b) order) and for patch=False
// (qmf.py:193-194: the three channel planes themselves, X [3][H][W]; p = 0).  One thread per matrix element.
__global__ __launch_bounds__(256) void k_rgb_matrix_any(const uint8_t* __restrict__ rgb, int H, int W, int p, int q, int top, int left,
                                                        int nw, long elems, float* __restrict__ X)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    const uint8_t* img = rgb + (long)blockIdx.y * 3 * H * W;
    float v;
    if (p == 0) {
        v = (float)img[e]; // [3][H][W] as it is
    } else {
        const int N = 3 * p * q;
        const long m = e / N;
        const int n = (int)(e - m * N);
        const int c = n / (p * q), rem = n - c * p * q, a = rem / q, b = rem - a * q;
        const int hh = (int)(m / nw), ww = (int)(m - (long)hh * nw);
        const int y = reflect_idx(hh * p + a - top, H), x = reflect_idx(ww * q + b - left, W);
        v = (float)img[(long)c * H * W + (long)y * W + x];
    }
    X[(long)blockIdx.y * elems + e] = v;
}

// qmf_decode of those streams (qmf.py:311-323, 351): u @ v.mT (exact integers), depatchify + unpad (p > 0), clamp + truncate.
// p > 0: U [B][M][R], V [B][3 p q][R];  p = 0: U [B][3][H][R], V [B][3][W][R].  One thread per pixel.
__global__ __launch_bounds__(256) void k_rgb_decode_any(const int8_t* __restrict__ U, const int8_t* __restrict__ V, int H, int W, int p,
                                                        int q, int top, int left, int nw, long u_img, long v_img, int R,
                                                        uint8_t* __restrict__ rgb)
{
    const long n = 3L * H * W;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e / ((long)H * W));
    const int rem = (int)(e - (long)c * H * W);
    const int y = rem / W, x = rem - y * W;
    const int8_t *u, *v;
    if (p == 0) {
        u = U + (long)blockIdx.y * u_img + ((long)c * H + y) * R;
        v = V + (long)blockIdx.y * v_img + ((long)c * W + x) * R;
    } else {
        const int yy = y + top, xx = x + left;
        const long m = (long)(yy / p) * nw + (xx / q);
        const int col = c * p * q + (yy % p) * q + (xx % q);
        u = U + (long)blockIdx.y * u_img + m * R;
        v = V + (long)blockIdx.y * v_img + (long)col * R;
    }
    float acc = 0.f;
    for (int r = 0; r < R; r++) acc = fmaf((float)u[r], (float)v[r], acc); // k-ordered chain; exact while |sum| < 2^24
    acc = fminf(fmaxf(acc, 0.f), 255.f);
    rgb[(long)blockIdx.y * n + e] = (uint8_t)acc;
}

// svd_decode of the RGB branch for any patch size / no patches and for quantised (uint8) or float factors
// (lrf/compression/svd.py:310-326, 359): dequantize (utils.py:223-243: (q - q.min()) * scale + min) when QUANT, u @ v.mT as
// a k-ordered fp32 fma chain, depatchify + unpad (p > 0), clamp + truncate.  Layouts as k_rgb_decode_any.  One thread per pixel.
template <bool QUANT>
__global__ __launch_bounds__(256) void k_svd_decode_any(const void* __restrict__ Uv, const void* __restrict__ Vv, int H, int W, int p, int q,
                                                        int top, int left, int nw, long u_img, long v_img, int R,
                                                        const float* __restrict__ qp /*[B][6] or NULL*/, uint8_t* __restrict__ rgb)
{
    const long n = 3L * H * W;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e / ((long)H * W));
    const int rem = (int)(e - (long)c * H * W);
    const int y = rem / W, x = rem - y * W;
    long uo, vo;
    if (p == 0) {
        uo = (long)blockIdx.y * u_img + ((long)c * H + y) * R;
        vo = (long)blockIdx.y * v_img + ((long)c * W + x) * R;
    } else {
        const int yy = y + top, xx = x + left;
        uo = (long)blockIdx.y * u_img + ((long)(yy / p) * nw + (xx / q)) * R;
        vo = (long)blockIdx.y * v_img + (long)(c * p * q + (yy % p) * q + (xx % q)) * R;
    }
    float acc = 0.f;
    if (QUANT) {
        const uint8_t* u = static_cast<const uint8_t*>(Uv) + uo;
        const uint8_t* v = static_cast<const uint8_t*>(Vv) + vo;
        const float* g = qp + (long)blockIdx.y * 6; // scale_u, min_u, qmin_u, scale_v, min_v, qmin_v
        for (int r = 0; r < R; r++) {
            const float uf = ((float)u[r] - g[2]) * g[0] + g[1];
            const float vf = ((float)v[r] - g[5]) * g[3] + g[4];
            acc = fmaf(uf, vf, acc);
        }
    } else {
        const float* u = static_cast<const float*>(Uv) + uo;
        const float* v = static_cast<const float*>(Vv) + vo;
        for (int r = 0; r < R; r++) acc = fmaf(u[r], v[r], acc);
    }
    acc = fminf(fmaxf(acc, 0.f), 255.f);
    rgb[(long)blockIdx.y * n + e] = (uint8_t)acc;
}
