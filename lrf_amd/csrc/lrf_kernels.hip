#pragma once
// lrf_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the QMF hot path.
//
// Compiled with -ffp-contract=off: every fused multiply-add below is explicit, because the int8
// factors must come out bit-identical to the reference's CPU arithmetic (see DESIGN.md "fp32 order").
// The f32 MFMA (v_mfma_f32_16x16x4_f32) is a k-ordered fmaf chain, which is exactly the order MKL's
// sgemm micro-kernel uses for the reference's `x @ v` and, in 384-row blocks, for `x.mT @ u`.
//
// Reference code restated here (paths relative to the reference root):
//   k_planes   lrf/compression/utils.py:24-47,76-95,108-132 + lrf/compression/qmf.py:43-56
//   k_init     lrf/factorization/qmf.py:42-71 (SVDInit; LAPACK replaced by the exact Gram matrix + a tridiagonal eigen-solver)
//   k_bcd      lrf/factorization/qmf.py:93-126 (update_u) + the a = x.mT @ u half of :128-139
//   k_vupdate  lrf/factorization/qmf.py:128-139 (update_v), :191-195 (_project)
//   k_decode   lrf/compression/qmf.py:329-351, lrf/compression/utils.py:50-73,98-105,135-182
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "lrf_internal.h"

#include "lrf_device.h"

// ------------------------------------------------------------------------------------------------
// K1: uint8 RGB -> patch matrices (YCbCr, area down-sampled chroma, reflect pad, patchify)
// ------------------------------------------------------------------------------------------------
// One workgroup per (patch row, image): a luma patch row consumes 8 image rows, a chroma patch row 16 (+ 1) and yields the
// patch row of BOTH chroma planes — the two planes have the same geometry and share every source byte, so the Cb and Cr
// samples of a window come from one set of loads (separate workgroups per chroma plane read the image three times:
// 512 x 1365x2048 4.8 -> 3.9 ms).  It writes nw complete 256-byte patches per plane.  Thread item = one float4 of a patch (16
// consecutive items = one patch, so stores are fully coalesced); word loads on the aligned interior fast paths.
__global__ __launch_bounds__(256) void k_planes(const uint8_t* __restrict__ rgb, int H, int W, ImageGeom g,
                                                float* __restrict__ X)
{
    const int pr = blockIdx.x; // luma patch rows, then chroma patch rows (grid: p[1].pr0 + p[1].nh)
    const int c = (pr >= g.p[1].pr0) ? 1 : 0;
    const PlaneGeom pg = g.p[c];
    const int hh = pr - pg.pr0;
    const uint8_t* img = rgb + (long)blockIdx.y * 3 * H * W;
    float* Xp = X + (long)blockIdx.y * g.img_floats + pg.xoff + (long)hh * pg.nw * 64;
    float* Xp2 = X + (long)blockIdx.y * g.img_floats + g.p[2].xoff + (long)hh * pg.nw * 64; // the Cr patch row (c == 1)
    const int hw = H * W;
    // F.interpolate(scale 0.5, "area") = adaptive average pooling to (floor(H/2), floor(W/2)): the window of sample (y, x)
    // starts at (2y, 2x) and is 2 wide for an even side, 3 wide for an odd one (floor(y H / h) = 2y, ceil((y+1) H / h) =
    // 2y + 2 + [H odd] for every y < h = floor(H/2)), which the general path below evaluates term by term.
    const int kh = 2 + (H & 1), kw = 2 + (W & 1);
    typedef uint32_t __attribute__((aligned(1))) u32u;
    typedef uint64_t __attribute__((aligned(1))) u64u;
    for (int it = threadIdx.x; it < pg.nw * 16; it += 256) {
        const int ww = it >> 4, a = (it >> 1) & 7, b4 = (it & 1) * 4;
        const int y = reflect_idx(hh * 8 + a - pg.top, pg.h);
        const int x0 = ww * 8 + b4 - pg.left;
        const bool interior = x0 >= 0 && x0 + 3 < pg.w; // no horizontal reflection inside this float4
        f32x4 out, out2;
        if (c == 0) {
            if (interior) { // four consecutive pixels of one row: one (unaligned) word per channel
                const uint8_t* p0 = img + y * W + x0;
                uint32_t r4 = *reinterpret_cast<const u32u*>(p0);
                uint32_t g4 = *reinterpret_cast<const u32u*>(p0 + hw);
                uint32_t b4w = *reinterpret_cast<const u32u*>(p0 + 2 * hw);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    out[i] = ycc_of((float)((r4 >> (8 * i)) & 255u), (float)((g4 >> (8 * i)) & 255u),
                                    (float)((b4w >> (8 * i)) & 255u), 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int x = reflect_idx(x0 + i, pg.w);
                    const uint8_t* p0 = img + y * W + x;
                    out[i] = ycc_of((float)p0[0], (float)p0[hw], (float)p0[2 * hw], 0);
                }
            }
        } else if (interior) {
            // windows of samples x0 .. x0+3: rows 2y .. 2y+kh-1, columns 2 x0 .. 2 x0 + 7 (+ 1 more when W is odd): per channel
            // and row one 8-byte word (unaligned when W is odd) and, for 3-wide windows, the ninth byte.  Window sum in
            // row-major order from 0, then / kh / kw — the order of the general path below.
            float sum[4] = {0.f, 0.f, 0.f, 0.f}, sum2[4] = {0.f, 0.f, 0.f, 0.f};
            for (int dy = 0; dy < kh; dy++) {
                const uint8_t* p0 = img + (long)(2 * y + dy) * W + 2 * x0;
                uint64_t ch[3];
                uint32_t ex[3] = {0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    ch[k] = *reinterpret_cast<const u64u*>(p0 + (long)k * hw);
                    if (kw == 3) ex[k] = p0[(long)k * hw + 8];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    for (int dx = 0; dx < kw; dx++) {
                        const int col = 2 * i + dx; // 0..8
                        float r_, g_, b_;
                        if (col < 8) {
                            r_ = (float)((ch[0] >> (8 * col)) & 255u);
                            g_ = (float)((ch[1] >> (8 * col)) & 255u);
                            b_ = (float)((ch[2] >> (8 * col)) & 255u);
                        } else {
                            r_ = (float)ex[0]; g_ = (float)ex[1]; b_ = (float)ex[2];
                        }
                        sum[i] = sum[i] + ycc_of(r_, g_, b_, 1);
                        sum2[i] = sum2[i] + ycc_of(r_, g_, b_, 2);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                out[i] = sum[i] / (float)kh / (float)kw;
                out2[i] = sum2[i] / (float)kh / (float)kw;
            }
        } else { // general adaptive-average-pool window (reflected columns), row-major fp32 sum, then / kh / kw
            const int h0 = (y * H) / pg.h, h1 = ((y + 1) * H + pg.h - 1) / pg.h;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int x = reflect_idx(x0 + i, pg.w);
                int w0 = (x * W) / pg.w, w1 = ((x + 1) * W + pg.w - 1) / pg.w;
                float sum = 0.f, sum2 = 0.f;
                for (int yy = h0; yy < h1; yy++)
                    for (int xx = w0; xx < w1; xx++) {
                        const uint8_t* p0 = img + yy * W + xx;
                        const float r_ = (float)p0[0], g_ = (float)p0[hw], b_ = (float)p0[2 * hw];
                        sum = sum + ycc_of(r_, g_, b_, 1);
                        sum2 = sum2 + ycc_of(r_, g_, b_, 2);
                    }
                out[i] = sum / (float)(h1 - h0) / (float)(w1 - w0);
                out2[i] = sum2 / (float)(h1 - h0) / (float)(w1 - w0);
            }
        }
        *reinterpret_cast<f32x4*>(Xp + ww * 64 + a * 8 + b4) = out;
        if (c) *reinterpret_cast<f32x4*>(Xp2 + ww * 64 + a * 8 + b4) = out2;
    }
}

// Fast path of K1 for images whose sides are multiples of 16 (no padding anywhere, exact 2x2 chroma windows, 8-byte
// aligned rows): every RGB byte is read once.  One workgroup per 16-row strip; a thread takes a 2 x 8 pixel block
// (six 8-byte loads) and produces its sixteen luma samples and its four Cb and four Cr samples with the arithmetic and
// orders of k_planes (ycc_of; the window sum is row-major: (0,0), (0,1), (1,0), (1,1)).
__global__ __launch_bounds__(256) void k_planes16(const uint8_t* __restrict__ rgb, int H, int W, ImageGeom g,
                                                  float* __restrict__ X)
{
    // luma staging: [h = patch row 0/1 of the strip][patch 0..31][16 float4], float4 slot q stored at q ^ swz(h, q)
    __shared__ __attribute__((aligned(16))) float Ls[2 * 32 * 64];
    const int hw = H * W, nwl = g.p[0].nw, nwc = g.p[1].nw;
    const int per_strip = (nwl + 31) / 32; // workgroups per strip: 32 luma patches (256 blocks of 2 x 8 pixels) each
    const int strip = blockIdx.x / per_strip;
    const int ww0 = (blockIdx.x - strip * per_strip) * 32;
    const uint8_t* img = rgb + (long)blockIdx.y * 3 * H * W;
    float* Xi = X + (long)blockIdx.y * g.img_floats;
    const int tid = threadIdx.x;
    const int wwl = tid >> 3, ww = ww0 + wwl, rp = tid & 7; // 8-pixel column block, row pair inside the strip
    if (ww < nwl) {
        const int y = 16 * strip + 2 * rp, x = 8 * ww;
        uint64_t ch[3][2];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++) ch[k][rr] = *reinterpret_cast<const uint64_t*>(img + k * hw + (y + rr) * W + x);
        // luma: rows y, y + 1 -> patch row 2 strip + (rp >> 2), rows a = 2 (rp & 3), a + 1 of patch ww: through LDS, so that
        // the global stores below are lane-contiguous (a thread's own 64 bytes would go out as 16-byte pieces 64 bytes apart:
        // 0.24 ms against 0.16 ms for the kernel)
        float* Lp = Ls + ((rp >> 2) * 32 + wwl) * 64;
        const int swz = (rp >> 1) & 3;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            f32x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o0[i] = ycc_of((float)((ch[0][rr] >> (8 * i)) & 255u), (float)((ch[1][rr] >> (8 * i)) & 255u),
                               (float)((ch[2][rr] >> (8 * i)) & 255u), 0);
                o1[i] = ycc_of((float)((ch[0][rr] >> (8 * (i + 4))) & 255u), (float)((ch[1][rr] >> (8 * (i + 4))) & 255u),
                               (float)((ch[2][rr] >> (8 * (i + 4))) & 255u), 0);
            }
            const int q = 4 * (rp & 3) + 2 * rr; // float4 slot of (row a + rr, left half) in the patch
            *reinterpret_cast<f32x4*>(Lp + 4 * (q ^ swz)) = o0;
            *reinterpret_cast<f32x4*>(Lp + 4 * ((q + 1) ^ swz)) = o1;
        }
        // chroma: samples (8 strip + rp, 4 ww .. 4 ww + 3) -> patch row strip, row a = rp of patch ww >> 1, columns 4 (ww & 1) ..
        // (lanes of an even / odd ww pair fill whole 32-byte rows: already contiguous)
#pragma unroll
        for (int c = 1; c < 3; c++) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float sum = 0.f;
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int cc = 0; cc < 2; cc++) {
                        const int sh = 8 * (2 * i + cc);
                        sum = sum + ycc_of((float)((ch[0][rr] >> sh) & 255u), (float)((ch[1][rr] >> sh) & 255u),
                                           (float)((ch[2][rr] >> sh) & 255u), c);
                    }
                o[i] = sum / 2.f / 2.f;
            }
            float* Cp = Xi + g.p[c].xoff + ((long)strip * nwc + (ww >> 1)) * 64 + rp * 8 + 4 * (ww & 1);
            *reinterpret_cast<f32x4*>(Cp) = o;
        }
    }
    __syncthreads();
    const int npw = nwl - ww0 < 32 ? nwl - ww0 : 32; // patches this workgroup holds per patch row
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int f = k * 256 + tid, h = f >> 9, rem = f & 511, pw = rem >> 4, q = rem & 15;
        if (pw < npw) {
            const int sw = (h << 1) | (q >> 3); // swz of the writer: rp = 4 h + (q >> 2)
            const f32x4 v = *reinterpret_cast<const f32x4*>(Ls + (h * 32 + pw) * 64 + 4 * (q ^ sw));
            *reinterpret_cast<f32x4*>(Xi + g.p[0].xoff + ((long)(2 * strip + h) * nwl + ww0) * 64 + rem * 4) = v;
        }
    }
}

// K1 for every other size, in one pass over the image (CLIC-sized 1365 x 2048: the odd height means 3-row pooling windows,
// one reflected luma row on top, two below, three reflected chroma rows on either side): k_planes16's tiling laid over the
// PADDED planes.  A workgroup owns padded luma rows 16 s .. 16 s + 15 (two luma patch rows) and padded chroma rows
// 8 s .. 8 s + 7 (one patch row of Cb and of Cr) of 32 luma patches' width; a thread takes a 2 x 8 block of padded luma
// pixels and four padded chroma samples per plane, maps them to source rows / columns with the reflect rule and loads whole
// 8-byte words wherever the eight source columns are not reflected (always, away from the left / right border).  The luma
// and chroma windows of a workgroup overlap in all but the halo rows, so every source byte comes from HBM once and from
// the cache otherwise; consecutive strips of an image run on the same XCD (lrf_api.hip maps the block index), which keeps
// the halo rows in that XCD's L2.  Arithmetic and orders are k_planes' (ycc_of; row-major window sums from 0, / kh / kw).
// k_planes (one workgroup per patch row, four-pixel items) read every byte twice with 4-byte loads and stays as the
// reference implementation of the geometry (LRF_PLANES_NO_TILED=1).
// x / 3 for 0 <= x < 2^14, correctly rounded (= the IEEE division the reference performs) in three instructions: with
// c = RN(1/3), q0 = RN(x c), r = x - 3 q0 (exact, one fma), q = RN(q0 + r c) (Markstein).  Checked against x / 3.0f for
// every float in [2^-20, 2^14) (285 M values, none differs; 0 maps to 0).  The generic division costs ten instructions.
__device__ __forceinline__ float div3_exact(float x)
{
    const float c = 1.0f / 3.0f;
    const float q0 = x * c;
    const float r = fmaf(-3.0f, q0, x);
    return fmaf(r, c, q0);
}
template <int K>
__device__ __forceinline__ float div_win(float x) { return K == 2 ? x * 0.5f : div3_exact(x); }
// byte j (0..7) of an 8-byte word held as two dwords -> float
template <int J>
__device__ __forceinline__ float byte_f(uint32_t lo, uint32_t hi) { return (float)(((J < 4 ? lo : hi) >> (8 * (J & 3))) & 255u); }
// luma of one pixel: ycc_of(.., 0) without its `0.f + acc` (acc >= +0: the same bits)
__device__ __forceinline__ float luma_of(float r, float g, float b)
{
    float acc = 0.299f * r; // = fmaf(0.299f, r, 0.f)
    acc = fmaf(0.587f, g, acc);
    return fmaf(0.114f, b, acc);
}

// sixteen luma samples of a 2 x 8 pixel block (channel words lo / hi = bytes 0..3 / 4..7 of each row) into the staging tile
__device__ __forceinline__ void strip_luma(const uint32_t (&lo)[2][3], const uint32_t (&hi)[2][3], float* Lp, int rp)
{
    const int swz = (rp >> 1) & 3;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const uint32_t r0 = lo[rr][0], g0 = lo[rr][1], b0 = lo[rr][2], r1 = hi[rr][0], g1 = hi[rr][1], b1 = hi[rr][2];
        const f32x4 o0 = (f32x4){luma_of(byte_f<0>(r0, r1), byte_f<0>(g0, g1), byte_f<0>(b0, b1)),
                                 luma_of(byte_f<1>(r0, r1), byte_f<1>(g0, g1), byte_f<1>(b0, b1)),
                                 luma_of(byte_f<2>(r0, r1), byte_f<2>(g0, g1), byte_f<2>(b0, b1)),
                                 luma_of(byte_f<3>(r0, r1), byte_f<3>(g0, g1), byte_f<3>(b0, b1))};
        const f32x4 o1 = (f32x4){luma_of(byte_f<4>(r0, r1), byte_f<4>(g0, g1), byte_f<4>(b0, b1)),
                                 luma_of(byte_f<5>(r0, r1), byte_f<5>(g0, g1), byte_f<5>(b0, b1)),
                                 luma_of(byte_f<6>(r0, r1), byte_f<6>(g0, g1), byte_f<6>(b0, b1)),
                                 luma_of(byte_f<7>(r0, r1), byte_f<7>(g0, g1), byte_f<7>(b0, b1))};
        const int q = 4 * (rp & 3) + 2 * rr; // float4 slot of (row a + rr, left half) in the patch
        *reinterpret_cast<f32x4*>(Lp + 4 * (q ^ swz)) = o0;
        *reinterpret_cast<f32x4*>(Lp + 4 * ((q + 1) ^ swz)) = o1;
    }
}
// four Cb and four Cr samples from their KH x (8 + [KW == 3]) source bytes per channel: row-major window sums from 0, / kh / kw
template <int KH, int KW>
__device__ __forceinline__ void strip_chroma(const uint32_t (&lo)[KH][3], const uint32_t (&hi)[KH][3], const uint32_t (&ex)[KH][3],
                                             f32x4& o, f32x4& o2)
{
    float sum[4] = {0.f, 0.f, 0.f, 0.f}, sum2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < KH; dy++) {
        float rf[9], gf[9], bf[9];
#define LRF_CVT(J) rf[J] = byte_f<J>(lo[dy][0], hi[dy][0]); gf[J] = byte_f<J>(lo[dy][1], hi[dy][1]); bf[J] = byte_f<J>(lo[dy][2], hi[dy][2]);
        LRF_CVT(0) LRF_CVT(1) LRF_CVT(2) LRF_CVT(3) LRF_CVT(4) LRF_CVT(5) LRF_CVT(6) LRF_CVT(7)
#undef LRF_CVT
        rf[8] = (float)ex[dy][0]; gf[8] = (float)ex[dy][1]; bf[8] = (float)ex[dy][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int dx = 0; dx < KW; dx++) {
                const int col = 2 * i + dx; // 0..8
                sum[i] = sum[i] + ycc_of(rf[col], gf[col], bf[col], 1);
                sum2[i] = sum2[i] + ycc_of(rf[col], gf[col], bf[col], 2);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { // sum / kh / kw
        o[i] = div_win<KW>(div_win<KH>(sum[i]));
        o2[i] = div_win<KW>(div_win<KH>(sum2[i]));
    }
}

template <int KH, int KW>
__global__ __launch_bounds__(256) void k_planes_strip(const uint8_t* __restrict__ rgb, int H, int W, ImageGeom g,
                                                      float* __restrict__ X, int per_strip, int nblk, int xcd_chunk)
{
    __shared__ __attribute__((aligned(16))) float Ls[2 * 32 * 64]; // luma staging, as in k_planes16
    typedef uint64_t __attribute__((aligned(1))) u64u;
    // blocks are dealt to the XCDs round-robin: XCD j gets the contiguous range [j chunk, (j + 1) chunk) of (strip, column group)
    const int bid = xcd_chunk ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (bid >= nblk) return;
    const int hw = H * W;
    const PlaneGeom pl = g.p[0], pc = g.p[1];
    const int nwl = pl.nw, nwc = pc.nw;
    const int strip = bid / per_strip;
    const int ww0 = (bid - strip * per_strip) * 32;
    const uint8_t* img = rgb + (long)blockIdx.y * 3 * H * W;
    float* Xi = X + (long)blockIdx.y * g.img_floats;
    const int tid = threadIdx.x;
    const int wwl = tid >> 3, ww = ww0 + wwl, rp = tid & 7; // 8-column block of padded luma pixels, row pair inside the strip
    float* Lp = Ls + ((rp >> 2) * 32 + wwl) * 64;
    const bool luma_row = 2 * strip + (rp >> 2) < pl.nh, chroma_row = strip < pc.nh;
    const long coff = ((long)strip * nwc + (ww >> 1)) * 64 + rp * 8 + 4 * (ww & 1);
    // Workgroups whose 256 padded columns need no reflection and hold whole patches (all of them, away from the left / right
    // border): every load of the thread — two luma rows, KH chroma window rows, three channels each — is issued before the
    // first use, without a branch in between (rows past the last patch row are clamped into the image and not stored).
    const int xl = 8 * ww0 - pl.left, xc = 4 * ww0 - pc.left;
    const bool xfast = xl >= 0 && xl + 256 <= pl.w && ww0 + 32 <= nwl && xc >= 0 && xc + 128 <= pc.w && (ww0 >> 1) + 16 <= nwc;
    // Strips whose luma rows and chroma windows are the SAME source rows (round 3, after the ablations of DESIGN.md section 4):
    // the two luma rows y0, y0 + 1 of a thread lie inside the window of chroma sample row floor(y0 / 2) — for KH = 3 always
    // (rows y0 - 1 .. y0 + 1 or y0 .. y0 + 2 by the parity of the top pad), for KH = 2 when the top pad is even — and the
    // columns coincide when the left pads do (left = 2 * chroma left).  Such a strip loads KH rows per thread instead of
    // 2 + KH and converts their bytes once; the price is that its eight chroma rows are shifted by d = chroma top -
    // ceil(top / 2) against the chroma patch row of the strip (they straddle two patch rows: stores of 32-byte rows, not whole
    // patches) and that a strip next to one that is not shared fills the |d| rows between them the separate way.
    const int tpar = pl.top & 1, dsh = pc.top - ((pl.top + 1) >> 1);
    auto strip_shared = [&](int s2) {
        if (s2 < 0) return false;
        const int ya = 16 * s2 - pl.top - ((KH == 3 && tpar) ? 1 : 0), yb = 16 * s2 + 15 - pl.top + ((KH == 3 && !tpar) ? 1 : 0);
        return (KH == 3 || !tpar) && pl.left == 2 * pc.left && ya >= 0 && yb <= H - 1 && 2 * s2 + 1 < pl.nh && 8 * s2 + dsh >= 0 &&
               8 * s2 + 7 + dsh < pc.hp;
    };
    const bool shared = xfast && strip_shared(strip);
    if (shared) {
        constexpr int NR = KH; // rows per thread: the window rows; the luma rows are rows l0, l0 + 1 of them
        const int l0 = (KH == 3 && tpar) ? 1 : 0;
        const int x0 = 8 * ww - pl.left, y0 = 16 * strip + 2 * rp - pl.top;
        uint32_t wlo[NR][3], whi[NR][3], wex[NR][3];
#pragma unroll
        for (int dy = 0; dy < NR; dy++) {
            const uint8_t* p0 = img + (long)(y0 - l0 + dy) * W + x0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint64_t v = *reinterpret_cast<const u64u*>(p0 + (long)k * hw);
                wlo[dy][k] = (uint32_t)v;
                whi[dy][k] = (uint32_t)(v >> 32);
                wex[dy][k] = KW == 3 ? p0[(long)k * hw + 8] : 0u;
            }
        }
        uint32_t lo[2][3], hi[2][3];
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                lo[rr][k] = wlo[l0 + rr][k];
                hi[rr][k] = whi[l0 + rr][k];
            }
        strip_luma(lo, hi, Lp, rp);
        f32x4 o, o2;
        strip_chroma<KH, KW>(wlo, whi, wex, o, o2);
        const int q = 8 * strip + rp + dsh; // padded chroma row of sample row floor(y0 / 2)
        const long co = ((long)(q >> 3) * nwc + (ww >> 1)) * 64 + (q & 7) * 8 + 4 * (ww & 1);
        *reinterpret_cast<f32x4*>(Xi + g.p[1].xoff + co) = o;
        *reinterpret_cast<f32x4*>(Xi + g.p[2].xoff + co) = o2;
        // the rows between this strip's chroma rows and those of a neighbour that is not shared
        const bool fill = dsh > 0 ? !strip_shared(strip - 1) : (dsh < 0 && !strip_shared(strip + 1));
        const int nfill = dsh > 0 ? dsh : -dsh, qf = dsh > 0 ? 8 * strip + rp : 8 * strip + 8 + dsh + rp;
        if (fill && rp < nfill && qf >= 0 && qf < pc.hp) {
            const int cy = reflect_idx(qf - pc.top, pc.h), cx0 = 4 * ww - pc.left;
            uint32_t clo[KH][3], chi[KH][3], cex[KH][3];
#pragma unroll
            for (int dy = 0; dy < KH; dy++) {
                const uint8_t* p0 = img + (long)(2 * cy + dy) * W + 2 * cx0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint64_t v = *reinterpret_cast<const u64u*>(p0 + (long)k * hw);
                    clo[dy][k] = (uint32_t)v;
                    chi[dy][k] = (uint32_t)(v >> 32);
                    cex[dy][k] = KW == 3 ? p0[(long)k * hw + 8] : 0u;
                }
            }
            strip_chroma<KH, KW>(clo, chi, cex, o, o2);
            const long cf = ((long)(qf >> 3) * nwc + (ww >> 1)) * 64 + (qf & 7) * 8 + 4 * (ww & 1);
            *reinterpret_cast<f32x4*>(Xi + g.p[1].xoff + cf) = o;
            *reinterpret_cast<f32x4*>(Xi + g.p[2].xoff + cf) = o2;
        }
    } else if (xfast) {
        uint32_t lo[2][3], hi[2][3], clo[KH][3], chi[KH][3], cex[KH][3];
        const int x0 = 8 * ww - pl.left, cx0 = 4 * ww - pc.left;
        int yl = 16 * strip + 2 * rp - pl.top, cy = 8 * strip + rp - pc.top;
        yl = yl < pl.hp - pl.top - 1 ? yl : pl.hp - pl.top - 2; // (only rows that are not stored)
        cy = cy < pc.hp - pc.top ? cy : pc.hp - pc.top - 1;
        cy = reflect_idx(cy, pc.h);
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint8_t* row = img + (long)reflect_idx(yl + rr, pl.h) * W + x0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint64_t v = *reinterpret_cast<const u64u*>(row + (long)k * hw);
                lo[rr][k] = (uint32_t)v;
                hi[rr][k] = (uint32_t)(v >> 32);
            }
        }
#pragma unroll
        for (int dy = 0; dy < KH; dy++) {
            const uint8_t* p0 = img + (long)(2 * cy + dy) * W + 2 * cx0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint64_t v = *reinterpret_cast<const u64u*>(p0 + (long)k * hw);
                clo[dy][k] = (uint32_t)v;
                chi[dy][k] = (uint32_t)(v >> 32);
                cex[dy][k] = KW == 3 ? p0[(long)k * hw + 8] : 0u;
            }
        }
        strip_luma(lo, hi, Lp, rp);
        f32x4 o, o2;
        strip_chroma<KH, KW>(clo, chi, cex, o, o2);
        if (chroma_row) {
            *reinterpret_cast<f32x4*>(Xi + g.p[1].xoff + coff) = o;
            *reinterpret_cast<f32x4*>(Xi + g.p[2].xoff + coff) = o2;
        }
    } else {
        // ---- luma: padded rows 16 strip + 2 rp (+ 1), padded columns 8 ww .. 8 ww + 7
        if (ww < nwl && luma_row) {
            const int x0 = 8 * ww - pl.left;
            const bool xin = x0 >= 0 && x0 + 7 < pl.w;
            uint32_t lo[2][3], hi[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int y = reflect_idx(16 * strip + 2 * rp + rr - pl.top, pl.h);
                const uint8_t* row = img + (long)y * W;
                if (xin) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint64_t v = *reinterpret_cast<const u64u*>(row + (long)k * hw + x0);
                        lo[rr][k] = (uint32_t)v;
                        hi[rr][k] = (uint32_t)(v >> 32);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        uint32_t a = 0, b = 0;
                        for (int j = 0; j < 4; j++) {
                            a |= (uint32_t)row[(long)k * hw + reflect_idx(x0 + j, pl.w)] << (8 * j);
                            b |= (uint32_t)row[(long)k * hw + reflect_idx(x0 + 4 + j, pl.w)] << (8 * j);
                        }
                        lo[rr][k] = a;
                        hi[rr][k] = b;
                    }
                }
            }
            strip_luma(lo, hi, Lp, rp);
        }
        // ---- chroma: padded row 8 strip + rp, padded columns 4 ww .. 4 ww + 3 of both planes -> patch (strip, ww >> 1).
        // F.interpolate(scale 0.5, "area"): the window of sample (y, x) starts at (2y, 2x), KH x KW = 2 for an even side, 3 for an odd one
        if ((ww >> 1) < nwc && chroma_row) {
            const int cy = reflect_idx(8 * strip + rp - pc.top, pc.h);
            const int cx0 = 4 * ww - pc.left;
            f32x4 o, o2;
            if (cx0 >= 0 && cx0 + 3 < pc.w) { // source columns 2 cx0 .. 2 cx0 + 7 (+ 1): one word (+ the ninth byte) per channel and row
                uint32_t clo[KH][3], chi[KH][3], cex[KH][3];
#pragma unroll
                for (int dy = 0; dy < KH; dy++) {
                    const uint8_t* p0 = img + (long)(2 * cy + dy) * W + 2 * cx0;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint64_t v = *reinterpret_cast<const u64u*>(p0 + (long)k * hw);
                        clo[dy][k] = (uint32_t)v;
                        chi[dy][k] = (uint32_t)(v >> 32);
                        cex[dy][k] = KW == 3 ? p0[(long)k * hw + 8] : 0u;
                    }
                }
                strip_chroma<KH, KW>(clo, chi, cex, o, o2);
            } else { // reflected columns: sample by sample
                float sum[4] = {0.f, 0.f, 0.f, 0.f}, sum2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int x = reflect_idx(cx0 + i, pc.w);
                    for (int dy = 0; dy < KH; dy++)
                        for (int dx = 0; dx < KW; dx++) {
                            const uint8_t* p0 = img + (long)(2 * cy + dy) * W + 2 * x + dx;
                            const float r_ = (float)p0[0], g_ = (float)p0[hw], b_ = (float)p0[2 * hw];
                            sum[i] = sum[i] + ycc_of(r_, g_, b_, 1);
                            sum2[i] = sum2[i] + ycc_of(r_, g_, b_, 2);
                        }
                    o[i] = div_win<KW>(div_win<KH>(sum[i]));
                    o2[i] = div_win<KW>(div_win<KH>(sum2[i]));
                }
            }
            *reinterpret_cast<f32x4*>(Xi + g.p[1].xoff + coff) = o;
            *reinterpret_cast<f32x4*>(Xi + g.p[2].xoff + coff) = o2;
        }
    }
    __syncthreads();
    const int npw = nwl - ww0 < 32 ? nwl - ww0 : 32; // luma patches this workgroup holds per patch row
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int f = k * 256 + tid, h = f >> 9, rem = f & 511, pw = rem >> 4, q = rem & 15;
        if (pw < npw && 2 * strip + h < pl.nh) {
            const int sw = (h << 1) | (q >> 3); // swz of the writer: rp = 4 h + (q >> 2)
            const f32x4 v = *reinterpret_cast<const f32x4*>(Ls + (h * 32 + pw) * 64 + 4 * (q ^ sw));
            *reinterpret_cast<f32x4*>(Xi + g.p[0].xoff + ((long)(2 * strip + h) * nwl + ww0) * 64 + rem * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2: SVD initialisation = the exact Gram matrix (k_gram64, lrf_gram_kernels.hip: 128-bit partials per row chunk, added
// here) + top-R eigen-pairs of the 64 x 64 Gram matrix (Householder tridiagonalisation, 64-way multisection on
// division-free Sturm counts, twisted factorisation, Gram-Schmidt, back-transformation) + scaling to (v0, w0).
// One workgroup (4 waves) per matrix.  Mirrors oracle/lrf_oracle.c (lrf_oracle_gram_exact, tridiagonalize, sturm_count,
// top_eigenvalues, twisted_vector, lrf_oracle_top_eig_f64, init_from_gram) operation for operation.
// ------------------------------------------------------------------------------------------------
// LDS carve of k_init; ZR = 8 or 16 eigenvectors' worth of scratch (chosen on the host from the largest rank
// of the call, so that R <= 8 batches fit three workgroups per CU)
// (ZR = 8: 36.2 KB, three workgroups per CU; 16: 44.5 KB, three per CU, or one beside two ZR = 8 ones, which is how the luma and
// chroma initialisations of a (9..16, <= 8, <= 8) call run side by side; 32: 68.6 KB, two per CU or one beside two ZR = 16 ones.)
#define LRF_INIT_VPACK 2016 /* doubles: v_0 .. v_61 packed take 63 + 62 + ... + 2 = 2015 */
template <int ZR>
struct InitLds {
    // The Gram matrix (32 KB); once it sits in the waves' registers: the Householder vectors v_k PACKED (v_k has 63 - k entries:
    // LRF_INIT_VPACK doubles in all, read again by the back-transformation) and, in the 16 KB behind them, as much of the
    // twisted-factorisation scratch D1 / D2 ([i][r], 64 * ZR doubles each) as fits: both at ZR <= 16, D1 at ZR = 32 (round 5:
    // 49 / 62 / 84 KB -> 41 / 45 / 69 KB, so that a ZR = 32 workgroup has room for two ZR = 16 ones beside it, and a CU for three
    // ZR = 16 or two ZR = 32 ones).  D1's first 2 KB hold `cpart` first: the matvec partial chains of the tridiagonalisation, then
    // the (d', e'^2) table of the eigenvalue searches.
    static constexpr int kDInA = ZR <= 16 ? 2 : (ZR == 32 ? 1 : 0); // how many of D1, D2 live inside A
    static constexpr int kDx = (2 - kDInA) * 64 * ZR;
    static constexpr bool kZInA = ZR == 8; // (and the eigenvectors too at ZR = 8: 37 KB, three workgroups beside a ZR = 16 one)
    double A[64 * 64];
    double Dx[kDx > 0 ? kDx : 2];
    double Zx[kZInA ? 2 : ZR * 64]; // eigenvectors in tridiagonal coordinates, then in the original basis
    double v[128], w[128], d[64], e[64], e2[64], tau[64], lam[ZR < 16 ? 16 : ZR]; // v, w: two buffers (tridiagonalisation)
    double scal[8];         // [0] t, [1] pivmin, [2] lo, [3] hi
    int flag[4];
    __device__ double* Zp() { return kZInA ? A + LRF_INIT_VPACK + 2 * 64 * ZR : Zx; }
    __device__ double* D1() { return kDInA >= 1 ? A + LRF_INIT_VPACK : Dx; }
    __device__ double* D2() { return kDInA == 2 ? A + LRF_INIT_VPACK + 64 * ZR : (kDInA == 1 ? Dx : Dx + 64 * ZR); }
};
static_assert(LRF_INIT_VPACK + 3 * 64 * 8 <= 64 * 64, "ZR = 8: packed v_k + D1 + D2 + Z inside A");
static_assert(LRF_INIT_VPACK % 2 == 0 && LRF_INIT_VPACK >= 63 * 62 / 2 + 62 && LRF_INIT_VPACK + 2 * 64 * 16 <= 64 * 64, "packed v_k + D1 + D2 inside A");

// NW waves per workgroup: 4 hold the matrix during the tridiagonalisation; with NW = 8 (ranks above 8: ZR = 16 / 64, where LDS
// leaves a CU two workgroups or one and its SIMDs mostly idle) waves 4..7 wait at the barriers of that stage and then take
// their share of what scales with the rank — sixteen eigenvalue searches and sixteen back-transformations per round instead of
// eight (round 5: k_init<16> 197 -> see DESIGN.md, k_init<64> at rank 26: 298 ->).  Which wave computes a vector does not change a bit of it.
// Waves per SIMD: 3 (three 4-wave workgroups per CU at ZR = 8) or 6 (three 8-wave workgroups: 80 registers, a dozen spilled
// dwords; the bound is a maximum too: with 3 a second 8-wave workgroup did not fit and 512 chroma planes of rank 13 ran in two
// rounds, 362 us; with 4 — 94 registers, no spills — two fit, and the luma workgroup of a (26,13,13) call had ONE chroma
// workgroup beside it: 256 x 512x768 at (20,10,10) 3.09 -> 2.97 ms, (26,13,13) 3.41 -> 3.28 ms with 6 and the packed LDS layout).
#ifndef LRF_INIT_MINW4
#define LRF_INIT_MINW4 3
#endif
#ifndef LRF_INIT_MINW8
#define LRF_INIT_MINW8 6
#endif
#ifndef LRF_INIT_MAXW8
#define LRF_INIT_MAXW8 6
#endif
// DENSE (eight-wave workgroups only): six waves per SIMD as described — for calls with more matrices than CUs, where workgroups
// of different families share a CU; otherwise four (94 registers, no spills): a workgroup alone on its CU is a pure latency
// chain and the spilled dwords cost it 10 % (64 x 512x768 at (16,8,8): the stage 303 us with four, 339 with six).
template <int ZR, int NW = (ZR > 8 ? 8 : 4), bool DENSE = true>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? (DENSE ? LRF_INIT_MINW8 : 4) : LRF_INIT_MINW4, NW == 8 ? LRF_INIT_MAXW8 : 4))) void k_init(const ulonglong2* __restrict__ Gpart, const int* __restrict__ gexp,
                                              int fixed_exp, const PlaneDesc* __restrict__ planes,
                                              const int8_t* __restrict__ sign, float* __restrict__ Vf,
                                              float* __restrict__ Wf, int debug_stop, int rp, int plane0 /* first plane of this launch's run */)
{
    const int pli = blockIdx.x + plane0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    InitLds<ZR>& L = *reinterpret_cast<InitLds<ZR>*>(smem);
    double* G = L.A;
    double* const Zs = L.Zp();
    double* cpart = L.D1(); // [4][64] (2 KB of the >= 4 KB twisted-factorisation scratch, which is written two stages later)
    static_assert(64 * ZR >= 4 * 64, "cpart lives in D1");

    const PlaneDesc pd = planes[pli];
    const int M = pd.M, R = pd.R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- Gram matrix: the exact integer partials of k_gram64 (lrf_gram_kernels.hip), one per row chunk, are added as 128-bit
    // integers (order-free) and rounded once to fp64
    {
        const int E = fixed_exp != LRF_GRAM_EXP_FROM_DATA ? fixed_exp : gexp[pli];
        const double back = scalbn(1.0, 2 * (E - LRF_GRAM_BITS));
        const ulonglong2* gp = Gpart + (long)pd.gch0 * LRF_GRAM_SLOT;
        // chunk by chunk with the ten elements of a thread in flight together (one memory round trip per chunk, not per load)
        constexpr int NE = LRF_GRAM_SLOT / 256;
        unsigned long long lo[NE];
        long long hi[NE];
#pragma unroll
        for (int m = 0; m < NE; m++) { lo[m] = 0; hi[m] = 0; }
        for (int c = 0; c < (wave < 4 ? pd.ngch : 0); c++) {
            ulonglong2 v[NE];
#pragma unroll
            for (int m = 0; m < NE; m++) v[m] = gp[(long)c * LRF_GRAM_SLOT + tid + 256 * m];
#pragma unroll
            for (int m = 0; m < NE; m++) {
                const unsigned long long nl = lo[m] + v[m].x;
                hi[m] += (long long)v[m].y + (nl < lo[m] ? 1 : 0);
                lo[m] = nl;
            }
        }
#pragma unroll
        for (int m = 0; m < (wave < 4 ? NE : 0); m++) {
            const int e = tid + 256 * m;
            const double g = i128_to_double_rne(lo[m], hi[m]) * back;
            int ti, tj;
            gram_pair_tiles(e >> 8, ti, tj);
            const int ln = e & 63, reg = (e >> 6) & 3;
            const int gi = 16 * ti + 4 * (ln >> 4) + reg, gj = 16 * tj + (ln & 15);
            G[gi * 64 + gj] = g;
            G[gj * 64 + gi] = g; // diagonal tiles hold both triangles: the same exact value either way
        }
        __syncthreads();
    }
    if (debug_stop == 1) return;

    // ---- Householder tridiagonalisation (oracle: tridiagonalize).  The matrix lives in registers: thread (lane i,
    // wave g) holds A[16g + jj][i] (= A[i][16g + jj], the matrix stays exactly symmetric), jj = 0..15.  The LDS copy of
    // the Gram matrix is dead from here on and its space is reused for the Householder vectors v_k (read again by the
    // back-transformation) and the scratch of the later stages (InitLds).  Terms the oracle skips (j <= k) are fma(a, 0, c) = c here: v_k[j] = 0 there.
    d16 Ar; // a vector, not an array: row k is picked with a wave-uniform register index (s_set_gpr_idx), not 15 selects
    const int wrow = wave & 3; // (waves 4..7 of an eight-wave workgroup hold nothing: they only keep the barriers of this stage)
#pragma unroll
    for (int jj = 0; jj < 16; jj++) Ar[jj] = G[(16 * wrow + jj) * 64 + lane];
    __syncthreads();
#ifdef LRF_INIT_STAMPS
    unsigned long long acc_a = 0, acc_b = 0, acc_c = 0, acc_d = 0;
    const unsigned long long t_begin = stamp_now();
#define ISTAMP(var) unsigned long long var = stamp_now()
#else
#define ISTAMP(var)
#endif
    for (int k = 0; k < 62; k++) {
        ISTAMP(s0);
        double* vbuf = L.v + 64 * (k & 1); // v, w double-buffered: one barrier fewer per step
        double* wbuf = L.w + 64 * (k & 1);
        if (wave == (k >> 4)) { // the wave that holds row k
            const int i = lane;
            const double xk = Ar[k & 15];
            double x = (i > k) ? xk : 0.0;
            double sigma = wave_tree64(x * x);
            double hk = 0.0, ek = 0.0, vi = 0.0;
            if (sigma > LRF_SIGMA_TINY) {
                const double x0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k + 1),
                                                   __builtin_amdgcn_readlane(__double2loint(x), k + 1));
                double nrm = sqrt(sigma);
                double alpha = (x0 >= 0.0) ? -nrm : nrm;
                vi = (i > k + 1) ? x : 0.0;
                if (i == k + 1) vi = x0 - alpha;
                hk = fma(fabs(x0), nrm, sigma); // |v|^2 / 2
                ek = alpha;
            }
            vbuf[i] = vi;
            if (i > k) G[(62 * k - ((k * (k - 1)) >> 1) - 1) + i] = vi; // v_k, packed (InitLds), for the back-transformation
            if (i == 0) { L.tau[k] = 0.0; L.e[k] = ek; L.scal[0] = hk; L.flag[k & 1] = (sigma > LRF_SIGMA_TINY); }
        }
        __syncthreads();
        ISTAMP(s1);
#ifdef LRF_INIT_STAMPS
        acc_a += s1 - s0;
#endif
        if (L.flag[k & 1]) { // flag double-buffered like v: a wave that skips ahead must not overwrite what others still read
            double vj[16];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) vj[jj] = vbuf[16 * wrow + jj];
            double t = 0.0;
            if (wave == 0) t = 1.0 / L.scal[0]; // 2 / |v|^2: the division runs under the chains below
            if (wave < 4) { // matvec partial chains: thread (row i, column group g), two chains of eight
                double ca = 0.0, cb = 0.0;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    ca = fma(Ar[jj], vj[jj], ca);
                    cb = fma(Ar[8 + jj], vj[8 + jj], cb);
                }
                cpart[wave * 64 + lane] = (lane > k) ? ca + cb : 0.0;
            }
            __syncthreads();
            ISTAMP(s2);
            if (wave == 0) {
                const int i = lane;
                double p = t * (((cpart[i] + cpart[64 + i]) + cpart[128 + i]) + cpart[192 + i]);
                double vi = vbuf[i];
                double K = (0.5 * t) * wave_tree64(p * vi);
                wbuf[i] = fma(-K, vi, p);
                if (i == 0) L.tau[k] = t;
            }
            __syncthreads();
            ISTAMP(s3);
#ifdef LRF_INIT_STAMPS
            acc_b += s2 - s1;
            acc_c += s3 - s2;
#endif
            { // rank-2 update A -= v w^T + w v^T: the two products are rounded, then added (commutative), so element (r, c) and
              // its mirror image get the same bits without choosing an order per element.  v and w are zero up to index k,
              // which leaves the finished rows and columns as they are.  (Waves 4..7 would repeat waves 0..3's rows: skipped.)
                const double vc = vbuf[lane], wc = wbuf[lane];
                if (wave < 4) {
#pragma unroll
                    for (int jj = 0; jj < 16; jj++) {
                        const double wr = wbuf[16 * wrow + jj];
                        const double m1 = vj[jj] * wc, m2 = wr * vc;
                        Ar[jj] = Ar[jj] - (m1 + m2);
                    }
                }
            }
#ifdef LRF_INIT_STAMPS
            {
                asm volatile("" ::"v"(Ar[0]), "v"(Ar[15]));
                ISTAMP(s4);
                acc_d += s4 - s3;
            }
#endif
        }
    }
#ifdef LRF_INIT_STAMPS
    if (tid == 0 && blockIdx.x < 16384) {
        unsigned long long* o = g_stamps + 8 * blockIdx.x;
        o[0] = stamp_now() - t_begin; o[1] = acc_a; o[2] = acc_b; o[3] = acc_c; o[4] = acc_d;
    }
#endif
    __syncthreads();
    { // d = diag, e[62] = A[63][62]
        const int i = lane;
        if (wave == (i >> 4)) { // (waves 0..3)
            double dv = Ar[0];
#pragma unroll
            for (int jj = 1; jj < 16; jj++) dv = ((i & 15) == jj) ? Ar[jj] : dv;
            L.d[i] = dv;
        }
        if (wave == 3 && i == 62) { L.e[62] = Ar[15]; L.tau[62] = 0.0; }
        if (wave == 3 && i == 63) { L.e[63] = 0.0; L.tau[63] = 0.0; }
    }
    __syncthreads();
    if (debug_stop == 2) return;

    // ---- eigenvalues: Gershgorin hull, pivmin, the matrix scaled into [-1, 1] (oracle: top_eigenvalues)
    const int rmax = M < 64 ? M : 64;
    const int Rc = R < rmax ? R : rmax;
    double2* de = reinterpret_cast<double2*>(cpart); // (d'_i, e'_{i-1}^2): the matvec partials are dead
    if (wave == 0) {
        const int i = lane;
        double ei = (i < 63) ? L.e[i] : 0.0, eim = (i > 0) ? L.e[i - 1] : 0.0;
        double e2i = ei * ei;
        L.e2[i] = e2i;
        double rad = fabs(eim) + fabs(ei);
        double a = L.d[i] - rad, b = L.d[i] + rad, m2 = (i < 63) ? e2i : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a = fmin(a, __shfl_xor(a, off, 64));
            b = fmax(b, __shfl_xor(b, off, 64));
            m2 = fmax(m2, __shfl_xor(m2, off, 64));
        }
        double tn = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
        double pivmin = 2.2250738585072014e-300 * (m2 > 1.0 ? m2 : 1.0);
        double slack = 2.0 * tn * 2.220446049250313e-16 * 64 + 2.0 * pivmin;
        a -= slack;
        b += slack;
        const int s = __builtin_amdgcn_frexp_exp(fabs(a) > fabs(b) ? fabs(a) : fabs(b));
        const double es = ldexp(eim, -s);
        de[i] = make_double2(ldexp(L.d[i], -s), es * es);
        if (i == 0) { L.scal[1] = pivmin; L.scal[2] = ldexp(a, -s); L.scal[3] = ldexp(b, -s); L.flag[2] = s; }
    }
    __syncthreads();
    {
        const int sc = L.flag[2];
        // A wave runs two searches at once (eigenvalues r0 and r0 + NW, 64 shifts per pass each): at three workgroups per CU the
        // stage is paced by the LDS reads of the table, and one read now serves both.  A minor that comes out as zero needs
        // the oracle's replacement rule: such passes (hardly ever) are redone by sturm_count_slow.
        auto sturm_count_slow = [&](double x) {
            double p = 1.0, pp = 0.0;
            int cnt = 0;
            for (int i = 0; i < 64; i++) {
                const double2 q = de[i];
                double pn = fma(q.x - x, p, -(q.y * pp));
                if (pn == 0.0) pn = (__double2hiint(p) < 0) ? 0x1p-200 : -0x1p-200;
                cnt += ((__double2hiint(pn) ^ __double2hiint(p)) < 0);
                pp = p;
                p = pn;
                if ((i & 7) == 7) {
                    const int ea = __builtin_amdgcn_frexp_exp(p), eb = __builtin_amdgcn_frexp_exp(pp);
                    const int m = ea > eb ? ea : eb;
                    p = ldexp(p, -m);
                    pp = ldexp(pp, -m);
                }
            }
            return cnt;
        };
        auto narrow = [&](double x, int cnt, int kk, double& a, double& b) {
            unsigned long long mask = __ballot(cnt > kk);
            int j = mask ? (int)__builtin_ctzll(mask) : 64;
            double xm = __shfl(x, j > 0 ? j - 1 : 0, 64), xj = __shfl(x, j < 64 ? j : 63, 64);
            double na = (j == 0) ? a : xm, nb = (j == 64) ? b : xj;
            a = na;
            b = nb;
        };
        auto search = [&](int r0, auto two_tag) {
            constexpr bool TWO = decltype(two_tag)::value;
            const int r1 = r0 + NW;
            const int kk0 = 63 - r0, kk1 = 63 - r1;
            double a0 = L.scal[2], b0 = L.scal[3], a1 = a0, b1 = b0;
            for (int pass = 0; pass < 10; pass++) {
                const double x0 = a0 + ((b0 - a0) / 65.0) * (double)(lane + 1);
                const double x1 = a1 + ((b1 - a1) / 65.0) * (double)(lane + 1);
                // Sturm counts, division-free (oracle: sturm_count): one dependent fma per step and search
                double p0 = 1.0, pp0 = 0.0, p1 = 1.0, pp1 = 0.0;
                bool zero = false;
                unsigned sg0[2] = {0u, 0u}, sg1[2] = {0u, 0u}; // the sign bits of p_1 .. p_64, first at the top: one v_alignbit per step
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
#pragma unroll 1
                    for (int g = 4 * hf; g < 4 * hf + 4; g++) { // not unrolled: the table reads would be hoisted out of the pass loop
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const double2 q = de[8 * g + u];
                            const double n0 = fma(q.x - x0, p0, -(q.y * pp0));
                            zero |= (n0 == 0.0);
                            sg0[hf] = __builtin_amdgcn_alignbit(sg0[hf], (unsigned)__double2hiint(n0), 31);
                            pp0 = p0; p0 = n0;
                            if constexpr (TWO) {
                                const double n1 = fma(q.x - x1, p1, -(q.y * pp1));
                                zero |= (n1 == 0.0);
                                sg1[hf] = __builtin_amdgcn_alignbit(sg1[hf], (unsigned)__double2hiint(n1), 31);
                                pp1 = p1; p1 = n1;
                            }
                        }
                        const int ea0 = __builtin_amdgcn_frexp_exp(p0), eb0 = __builtin_amdgcn_frexp_exp(pp0);
                        const int m0 = ea0 > eb0 ? ea0 : eb0;
                        p0 = ldexp(p0, -m0); pp0 = ldexp(pp0, -m0);
                        if constexpr (TWO) {
                            const int ea1 = __builtin_amdgcn_frexp_exp(p1), eb1 = __builtin_amdgcn_frexp_exp(pp1);
                            const int m1 = ea1 > eb1 ? ea1 : eb1;
                            p1 = ldexp(p1, -m1); pp1 = ldexp(pp1, -m1);
                        }
                    }
                }
                const unsigned long long s0 = ((unsigned long long)sg0[0] << 32) | sg0[1], s1 = ((unsigned long long)sg1[0] << 32) | sg1[1];
                int cnt0 = __popcll(s0 ^ (s0 >> 1)), cnt1 = __popcll(s1 ^ (s1 >> 1)); // sign changes along 1, p_1, ..., p_64
                if (__any(zero) || debug_stop == 100) { // wave-uniform (100: the tests force this path, LRF_DEBUG_INIT_SWEEPS)
                    cnt0 = sturm_count_slow(x0);
                    if constexpr (TWO) cnt1 = sturm_count_slow(x1);
                }
                narrow(x0, cnt0, kk0, a0, b0);
                if constexpr (TWO) narrow(x1, cnt1, kk1, a1, b1);
            }
            if (lane == 0) {
                L.lam[r0] = ldexp(0.5 * (a0 + b0), sc);
                if constexpr (TWO) L.lam[r1] = ldexp(0.5 * (a1 + b1), sc);
            }
        };
        for (int r0 = wave; r0 < Rc; r0 += 2 * NW) {
            if (r0 + NW < Rc) search(r0, std::true_type{}); // wave-uniform
            else search(r0, std::false_type{});
        }
    }
    __syncthreads();
    if (debug_stop == 3) return;

    // ---- eigenvectors of T by twisted factorisation (oracle: twisted_vector), scratch [i][r].  The two pivot recurrences of
    // vector r are independent chains of 63 divisions: thread r runs the forward one, thread 64 + r the backward one; after
    // the barrier both find the twist index and each fills its side of the vector.
    {
        const int r = tid & 63;
        const bool fwd = tid < 64, mine = tid < 128 && r < Rc;
        const double lam = mine ? L.lam[r] : 0.0, pivmin = L.scal[1];
        double* Dp = L.D1() + r;
        double* Dm = L.D2() + r;
        if (mine && fwd) {
            double q = L.d[0] - lam;
            Dp[0] = q;
            for (int i = 1; i < 64; i++) {
                if (fabs(q) < pivmin) q = -pivmin;
                q = (L.d[i] - lam) - L.e2[i - 1] / q;
                Dp[i * ZR] = q;
            }
        }
        if (mine && !fwd) {
            double q = L.d[63] - lam;
            Dm[63 * ZR] = q;
            for (int i = 62; i >= 0; i--) {
                if (fabs(q) < pivmin) q = -pivmin;
                q = (L.d[i] - lam) - L.e2[i] / q;
                Dm[i * ZR] = q;
            }
        }
        __syncthreads();
        if (mine) {
            int kt = 0;
            double best = 0.0;
            for (int i = 0; i < 64; i++) {
                double g = fabs((Dp[i * ZR] + Dm[i * ZR]) - (L.d[i] - lam));
                if (i == 0 || g < best) { best = g; kt = i; }
            }
            double* x = Zs + r * 64;
            double xv = 1.0;
            if (fwd) {
                x[kt] = 1.0;
                for (int i = kt - 1; i >= 0; i--) {
                    double qq = Dp[i * ZR];
                    if (fabs(qq) < pivmin) qq = -pivmin;
                    xv = -(L.e[i] / qq) * xv;
                    x[i] = xv;
                }
            } else {
                for (int i = kt; i < 63; i++) {
                    double qq = Dm[(i + 1) * ZR];
                    if (fabs(qq) < pivmin) qq = -pivmin;
                    xv = -(L.e[i] / qq) * xv;
                    x[i + 1] = xv;
                }
            }
        }
    }
    __syncthreads();
    if (debug_stop == 4) return;

    // ---- scale, modified Gram-Schmidt, normalise (sequential over r, wave 0, lane = element)
    if (wave == 0) {
        const int i = lane;
        for (int r = 0; r < Rc; r++) {
            double x = Zs[r * 64 + i];
            bool use_twisted = __all(isfinite(x));
            int uidx = 0;
            for (;;) {
                if (use_twisted) {
                    double n0 = sqrt(wave_tree64(x * x));
                    x = x / n0;
                } else {
                    if (uidx >= 64) break; // unreachable for finite input (mirrors the oracle's failure exit)
                    x = (i == uidx) ? 1.0 : 0.0;
                    uidx++;
                }
                for (int pr = 0; pr < r; pr++) {
                    double pv = Zs[pr * 64 + i];
                    double c = wave_tree64(pv * x);
                    x = fma(-c, pv, x);
                }
                double n2 = wave_tree64(x * x);
                if (n2 > 1e-6 && n2 < 1e300) {
                    x = x / sqrt(n2);
                    break;
                }
                use_twisted = false;
            }
            Zs[r * 64 + i] = x;
        }
    }
    __syncthreads();
    if (debug_stop == 5) return;

    // ---- back-transformation x <- H_0 ... H_61 x, sign, scaling, output: one wave per vector
    float* Vp = Vf + (long)pli * 64 * rp; // rp: padded rank (row pitch) of the V / W tables, a power of two
    float* Wp = Wf + (long)pli * 64 * rp;
    for (int i = tid; i < 64 * rp; i += 64 * NW) {
        if ((i & (rp - 1)) >= Rc) { Vp[i] = 0.f; Wp[i] = 0.f; } // padding and the r >= min(M,N) columns
    }
    for (int r0 = wave; r0 < Rc; r0 += 2 * NW) { // two vectors per wave and pass (r0 and r0 + NW): their reduction trees interleave
        const int i = lane;
        const int r1 = r0 + NW;
        const bool two = r1 < Rc; // wave-uniform
        double xa = Zs[r0 * 64 + i], xb = two ? Zs[r1 * 64 + i] : 0.0;
        for (int k = 61; k >= 0; k--) {
            double tk = L.tau[k];
            if (tk == 0.0) continue;
            double v = (i > k) ? G[(62 * k - ((k * (k - 1)) >> 1) - 1) + i] : 0.0; // (packed: InitLds)
            double sa = tk * wave_tree64(v * xa);
            double sb = tk * wave_tree64(v * xb);
            xa = fma(-sa, v, xa);
            xb = fma(-sb, v, xb);
        }
        Zs[r0 * 64 + i] = xa;
        if (two) Zs[r1 * 64 + i] = xb;
      for (int half = 0; half < (two ? 2 : 1); half++) {
        const int r = half ? r1 : r0;
        const double x = half ? xb : xa;
        double dot = 0.0;
        for (int j = 0; j < 64; j++) dot = fma((double)(j + 1), Zs[r * 64 + j], dot); // every lane: same chain
        double lam = L.lam[r];
        double sigma = sqrt(lam > 1e-200 ? lam : 0.0); // noise-floor eigenvalues count as zero (oracle: same)
        double sr = sqrt(sigma);
        int sg = (pd.sign_off >= 0 && sign) ? (int)sign[pd.sign_off + r] : 0;
        double want = sg ? (double)sg : -1.0;
        double flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.0 : -1.0;
        double ev = flip * x;
        Vp[i * rp + r] = (float)(ev * sr);
        Wp[i * rp + r] = (sr > 0.0) ? (float)(ev / sr) : 0.f;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Gauss-Seidel over the R columns of one row (qmf.py:108-119).  a[] = (x @ v) row, u[] = current
// row of the factor being updated (in/out).  gt: the per-matrix table of b = v.mT @ v (layout below)
// with b = v.mT @ v, den[r] = (b[r][r] + 0) + eps, rden = 1/den (uniform across lanes: scalar loads
// when it lives in global memory).  `native` selects the ATen small-product order for `uu @ bb`.
//
// Division: the reference computes round(fl(num / den)).  q~ = num * rden is within 3 ulp of that
// quotient, so unless q~ sits within `1/2 - fthr` of a rounding tie its nearest integer is the same;
// only then (or never, beyond the clamp range) is the IEEE division evaluated.  Results are identical.
#include "lrf_gs.h"

// Sweep calls (lrf_qmf_encode_sweep_rgb_u8): a plane whose matrix X is also factorised at a larger rank takes the first R columns
// of that plane's initial tables (PlaneDesc::init_src) — the top-R singular pairs are the first R of the top-R' ones, bit for bit:
// every stage of k_init treats component r without reference to the rank asked for (its eigenvalue search, its twisted vector,
// its Gram-Schmidt against components < r, its back-transformation and sign).  One workgroup per plane; the two planes may sit
// in tables of different rank pitch (V16 / W16: the pitch-16 set of a call that mixes pitches; split: every plane on the family of
// its own rank, else all on pitch rp_main).
__global__ __launch_bounds__(256) void k_init_share(const PlaneDesc* __restrict__ planes, float* __restrict__ Vm, float* __restrict__ Wm,
                                                    float* __restrict__ V16, float* __restrict__ W16, int split, int mixed, int rp_main)
{
    const int pli = blockIdx.x;
    const PlaneDesc pd = planes[pli];
    if (pd.init_src == pli) return;
    const PlaneDesc ps = planes[pd.init_src];
    const int pitch_d = split ? (LRF_FAM_OF_RANK(pd.R) == 2 ? LRF_RPB : 16) : rp_main;
    const int pitch_s = split ? (LRF_FAM_OF_RANK(ps.R) == 2 ? LRF_RPB : 16) : rp_main;
    float* Vd = ((mixed && pitch_d == 16) ? V16 : Vm) + (long)pli * 64 * pitch_d;
    float* Wd = ((mixed && pitch_d == 16) ? W16 : Wm) + (long)pli * 64 * pitch_d;
    const float* Vs = ((mixed && pitch_s == 16) ? V16 : Vm) + (long)pd.init_src * 64 * pitch_s;
    const float* Ws = ((mixed && pitch_s == 16) ? W16 : Wm) + (long)pd.init_src * 64 * pitch_s;
    for (int i = threadIdx.x; i < 64 * pitch_d; i += 256) {
        const int row = i / pitch_d, col = i - row * pitch_d;
        const bool in = col < pd.R;
        Vd[i] = in ? Vs[row * pitch_s + col] : 0.f;
        Wd[i] = in ? Ws[row * pitch_s + col] : 0.f;
    }
}

// b table of the initial V (after k_init or k_load_v0): one workgroup per matrix
__global__ __launch_bounds__(256) void k_bprep(const PlaneDesc* __restrict__ planes, const float* __restrict__ Vf,
                                               float* __restrict__ Bf, int plane0)
{
    __shared__ float v_s[64 * LRF_RP];
    const int pli = blockIdx.x + plane0;
    for (int i = threadIdx.x; i < 64 * LRF_RP; i += 256) v_s[i] = Vf[(long)pli * 64 * LRF_RP + i];
    __syncthreads();
    make_gtable(v_s, 64, planes[pli].R, Bf + (long)pli * LRF_GT_STRIDE, threadIdx.x, 256);
}

// ------------------------------------------------------------------------------------------------
// K3: one BCD half-iteration over X: U update (row local) fused with the partials of the following V
// update, a' = X^T U (fp32 chain over the block's rows) and b' = U^T U (exact).  One workgroup (4 waves)
// per (matrix, 384-row block), six 64-row sub-tiles.
// The row-major MFMA operand (a = X V: lane = row) comes straight from global memory: float4 loads one
// sub-tile ahead, then a 4x4 transpose across the four 16-lane rows with v_permlane16_swap /
// v_permlane32_swap.  The same registers are then stored to LDS (XOR-swizzled, conflict-free both ways)
// in the shadow of the Gauss-Seidel, for the transposed operand of a' = X^T U.
// MODE 0: old U from int8 (iterations >= 2); MODE 1: first iteration, old U = X @ W0 computed here;
// MODE 2: first iteration, old U = caller's fp32 U0.
// ------------------------------------------------------------------------------------------------
// 4x4 transpose between the four 16-lane rows of a wave and the four components of c:
// in: component i of lane row j = M[j][i]; out: component t of lane row j = M[t][j].
__device__ __forceinline__ void rows_transpose4(f32x4& c)
{
    unsigned c0 = __float_as_uint(c[0]), c1 = __float_as_uint(c[1]), c2 = __float_as_uint(c[2]), c3 = __float_as_uint(c[3]);
    auto s01 = __builtin_amdgcn_permlane16_swap(c0, c1, false, false); // c0.row1 <-> c1.row0, c0.row3 <-> c1.row2
    auto s23 = __builtin_amdgcn_permlane16_swap(c2, c3, false, false);
    auto s02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false); // upper half of the first <-> lower half of the second
    auto s13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
    c[0] = __uint_as_float(s02[0]);
    c[1] = __uint_as_float(s13[0]);
    c[2] = __uint_as_float(s02[1]);
    c[3] = __uint_as_float(s13[1]);
}

template <int MODE, int RMAX>
__global__ __launch_bounds__(256) void k_bcd(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                             const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                             const float* __restrict__ Wf, const float* __restrict__ Bf,
                                             const float* __restrict__ U0, int8_t* __restrict__ U,
                                             float* __restrict__ Ppart, float* __restrict__ Qpart, GsParams gp)
{
    // X sub-tile for the a' = X^T U operand: element (m, n) at m * 64 + (n ^ xs_swz(m)); both the dword stores
    // (lane = row m, 4 columns apart) and the dword loads (lane = column n, 4 rows) touch 32 distinct banks per half wave
    __shared__ __attribute__((aligned(16))) float Xs[64 * 64];
    __shared__ __attribute__((aligned(16))) float a_s[64 * LRF_RP];
    __shared__ __attribute__((aligned(16))) float u_s[64 * LRF_RP];
    __shared__ __attribute__((aligned(16))) int8_t uold_s[64 * LRF_RP];
    __shared__ __attribute__((aligned(16))) float va_s[16 * 64];
    __shared__ __attribute__((aligned(16))) float wa_s[MODE == 1 ? 16 * 64 : 4];

    const BlockDesc bd = blocks[blockIdx.x];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform (scalar branches below)
    const int li = lane & 15, lq = lane >> 4;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RP;
    const float* gt = Bf + (long)bd.plane * LRF_GT_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;
    const int invR = (65536 + R - 1) / R; // (i * invR) >> 16 == i / R for i < 64 * R
    // the symmetric b table of the exact Gauss-Seidel (ranks 9..16, iterations >= 2): tabv[j], lane l = b[j][l & 15]
    float tabv[17]; // [16]: lane l = 1 / den[l & 15]
#pragma unroll
    for (int j = 0; j < 17; j++) tabv[j] = 0.f;
    if (RMAX > 8 && MODE == 0 && gp.exact_int && R > 8) {
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < R && li < R) tabv[j] = (j == li) ? gt[li * LRF_GT_LD + LRF_GT_DEN] : gt[li * LRF_GT_LD + (j < li ? j : j - 1)];
        if (li < R) tabv[16] = gt[li * LRF_GT_LD + LRF_GT_RDEN];
    }

    // A operand of a^T = V^T X^T : A[i = r][k]; lane needs V[4s + lq][li] at k-step s.  Kept in LDS in
    // [step][lane] order (conflict-free, one ds_read per MFMA) rather than in 16 VGPRs per wave.
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int s_ = 4 * wave + i;
        va_s[s_ * 64 + lane] = Vp[(4 * s_ + lq) * LRF_RP + li];
        if (MODE == 1) wa_s[s_ * 64 + lane] = Wf[(long)bd.plane * 64 * LRF_RP + (4 * s_ + lq) * LRF_RP + li];
    }

    // prefetch registers: this lane's quarter of its row of the next sub-tile (row 16*wave + li, columns
    // 16q + 4lq .. +3) and the old int8 U rows (<= 4 bytes per thread)
    f32x4 xq[4];
    int8_t upre[RMAX / 4];
    // Rows past the end of the block are clamped to its last row instead of masked (no branches, so the compiler
    // counts outstanding loads exactly): their `a` values are never used and their U rows are zero, so they add
    // fma(x, 0, acc) = acc to the partials.
    constexpr int NB = RMAX / 4; // int8 U bytes per thread and sub-tile: 64 * RMAX / 256
    auto issue = [&](int t) {
        const int r0 = t * 64;
        int row = r0 + 16 * wave + li;
        row = row < nrows ? row : nrows - 1;
        const float* src = Xp + (long)row * 64 + 4 * lq;
#pragma unroll
        for (int q = 0; q < 4; q++) {
#ifndef LRF_ABLATE_LOADS
            xq[q] = *reinterpret_cast<const f32x4*>(src + 16 * q);
#else
            xq[q] = (f32x4){(float)lane, 1.f, 2.f, (float)t};
#endif
        }
        if (MODE == 0) {
            int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                int e = i * 256 + tid;
                upre[i] = Ub[(long)r0 * R + (e < lim ? e : lim - 1)];
            }
        }
    };

    f32x4 accP = (f32x4){0.f, 0.f, 0.f, 0.f}, accQ = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef LRF_STAMPS
    unsigned long long c_stage = 0, c_umfma = 0, c_gs = 0, c_pq = 0, c_g1 = 0, c_g2 = 0, c_g3 = 0;
#endif
    STAMP(t_begin);
    issue(0);
    __syncthreads(); // va_s / wa_s complete
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        STAMP(t1);
        if (MODE != 0) __syncthreads(); // first iteration only: u_s is also written in the U phase below
        // ---- B operand of a^T = V^T X^T for rows 16*wave .. +15: bx[4q + t'] = X[row][16q + 4t' + lq]
        float bx[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            rows_transpose4(xq[q]);
#pragma unroll
            for (int i = 0; i < 4; i++) bx[4 * q + i] = xq[q][i];
        }
#ifdef LRF_STAMPS
        asm volatile("" ::"v"(bx[15]), "v"(bx[0]));
        STAMP(u1);
        STAMP_ADD(c_stage, t1, u1);
#endif
        // ---- a^T tile: 16 chained MFMAs over k
        {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accw = (f32x4){0.f, 0.f, 0.f, 0.f};
            float av[16], aw[MODE == 1 ? 16 : 1];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                av[s] = va_s[s * 64 + lane];
                if (MODE == 1) aw[s] = wa_s[s * 64 + lane];
            }
#pragma unroll
            for (int s = 0; s < 16; s++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bx[s], acc, 0, 0, 0);
                if (MODE == 1) accw = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], bx[s], accw, 0, 0, 0);
            }
            // D[i = 4*lq + reg (r)][j = li (row)]
            *reinterpret_cast<f32x4*>(&a_s[(16 * wave + li) * LRF_RP + 4 * lq]) = acc;
            if (MODE == 1) *reinterpret_cast<f32x4*>(&u_s[(16 * wave + li) * LRF_RP + 4 * lq]) = accw;
        }
#ifdef LRF_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(u2);
        STAMP_ADD(c_g1, u1, u2);
#endif
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NB; i++) uold_s[i * 256 + tid] = upre[i];
        }
#ifdef LRF_STAMPS
        STAMP(u3);
        STAMP_ADD(c_g2, u2, u3);
#endif
        __syncthreads();
        STAMP(t2);
#ifdef LRF_STAMPS
        STAMP_ADD(c_g3, u3, t2);
#endif
        STAMP_ADD(c_umfma, t1, t2);
        // ---- this wave's 16 rows -> LDS (bx[4q + i] = X[m][(16q + 4i) ^ lq], the bits of lq and 16q + 4i are disjoint),
        //      then the prefetch of the next sub-tile into the same registers
        {
            const int m = 16 * wave + li;
            const int e = lq ^ (((m & 1) << 4) | (((m >> 1) & 7) << 1));
            float* xw = &Xs[m * 64];
#pragma unroll
            for (int c = 0; c < 16; c++) xw[(4 * c) ^ e] = bx[c];
        }
        issue(t + 1 < nsub ? t + 1 : t); // unconditional (the last one re-reads its own tile): exact s_waitcnt counts
        // ---- Gauss-Seidel: one wave, lane = row (rotating wave so the VALU work spreads over SIMDs)
        if (wave == (t & 3)) {
            STAMP(g0);
            int row = r0 + lane;
            float* ur = &u_s[lane * LRF_RP];
            // the DPP broadcasts of the exact solve read lanes of the table registers: every lane of the wave must be active,
            // so rows past the end are solved too (their inputs are zeros / stale bytes, finite) and zeroed afterwards
            const bool all_lanes = (MODE == 0) && RMAX > 8 && gp.exact_int && R > 8;
            if (row < nrows || all_lanes) {
                if (MODE == 2) {
                    const float* up = U0 + pd.u0_off + ((long)bd.row0 + row) * R;
                    for (int r = 0; r < R; r++) ur[r] = up[r];
                }
                STAMP(g1);
#ifdef LRF_ABLATE_GS
                {
                    f32x4 a0 = *reinterpret_cast<const f32x4*>(&a_s[lane * LRF_RP]), a1 = *reinterpret_cast<const f32x4*>(&a_s[lane * LRF_RP + 4]);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        a0[q] = (q < R) ? fminf(fmaxf(rintf(a0[q] * 1e-4f), gp.lo), gp.hi) : 0.f;
                        a1[q] = (q + 4 < R) ? fminf(fmaxf(rintf(a1[q] * 1e-4f), gp.lo), gp.hi) : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(ur) = a0;
                    *reinterpret_cast<f32x4*>(ur + 4) = a1;
                    *reinterpret_cast<f32x4*>(ur + 8) = (f32x4){0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(ur + 12) = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#else
                gs_dispatch<RMAX, MODE == 0>(R, &a_s[lane * LRF_RP], ur, uold_s, lane, gt, pd.native_t2_u != 0, gp, tabv);
#endif
                STAMP(g2);
#ifdef LRF_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                STAMP(g3);
#endif
            }
            if (!(row < nrows)) {
#pragma unroll
                for (int r = 0; r < LRF_RP; r += 4) *reinterpret_cast<f32x4*>(ur + r) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        STAMP(t3);
        STAMP_ADD(c_gs, t2, t3);
        // ---- int8 U out (coalesced bytes), partial a' = X^T U for columns 16*wave..+15, partial b' = U^T U
        {
            // A operand: X[4s + lq][16*wave + li]; xs_swz(4s + lq) = F0(lq) | 4 * (s & 3)
            const int f0 = ((lq & 1) << 4) | ((lq >> 1) << 1);
            const float* xc = &Xs[lq * 64];
            const int nb = (16 * wave + li) ^ f0;
            const float* uc = &u_s[lq * LRF_RP + li];
            float px[16], pu[16], qu[4];
#pragma unroll
            for (int s = 0; s < 16; s++) px[s] = xc[256 * s + (nb ^ (4 * (s & 3)))];
            const float* uq = uc + 16 * wave * LRF_RP; // wave w: row steps 4w .. 4w+3 of the sub-tile
#pragma unroll
            for (int s = 0; s < 16; s++) pu[s] = uc[4 * s * LRF_RP];
#pragma unroll
            for (int s = 0; s < 4; s++) qu[s] = uq[4 * s * LRF_RP];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                accP = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s], pu[s], accP, 0, 0, 0);
                if ((s & 3) == 3) accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[s >> 2], qu[s >> 2], accQ, 0, 0, 0);
            }
            // int8 U out: coalesced bytes, a fixed number of (predicated) stores per thread
            int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                int e = i * 256 + tid;
                int row = (e * invR) >> 16, r = e - row * R;
                if (e < lim) Ub[(long)r0 * R + e] = (int8_t)u_s[row * LRF_RP + r];
            }
        }
#ifdef LRF_STAMPS
        asm volatile("" ::"v"(accP[0]), "v"(accQ[0]));
        STAMP(t4);
        STAMP_ADD(c_pq, t3, t4);
#endif
    }
    // a' partial: D[i = 4*lq + reg (column 16*wave + i)][j = li (r)]
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RP;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) Pp[(16 * wave + 4 * lq + reg) * LRF_RP + li] = accP[reg];
    // b' partial: the four waves hold disjoint row subsets; integers, so the sum order is immaterial
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 4; reg++) a_s[wave * 256 + (4 * lq + reg) * LRF_RP + li] = accQ[reg];
    __syncthreads();
    Qpart[slot * LRF_RP * LRF_RP + tid] = ((a_s[tid] + a_s[256 + tid]) + a_s[512 + tid]) + a_s[768 + tid];
#ifdef LRF_STAMPS
    if (tid == 0 && blockIdx.x < 16384) {
        STAMP(t_end);
        unsigned long long* o = g_stamps + 8 * blockIdx.x;
        o[0] = t_end - t_begin; o[1] = c_stage; o[2] = c_umfma; o[3] = c_gs; o[4] = c_pq; o[5] = c_g1; o[6] = c_g2; o[7] = c_g3;
        if (getenv_probe_dummy == 1) { // GS probe view: replaces the U-phase split
            __threadfence();
            o[1] = g_gsp[4 * blockIdx.x + 0]; o[5] = g_gsp[4 * blockIdx.x + 1]; o[6] = g_gsp[4 * blockIdx.x + 2]; o[7] = g_gsp[4 * blockIdx.x + 3];
            g_gsp[4 * blockIdx.x + 0] = 0; g_gsp[4 * blockIdx.x + 1] = 0; g_gsp[4 * blockIdx.x + 2] = 0; g_gsp[4 * blockIdx.x + 3] = 0;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// K3b: V update for one matrix per workgroup: a' = sum of the block partials (in block order),
// b' = U^T U (sum of exact integer partials), Gauss-Seidel over the R columns for the 64 rows of V,
// then the b table (v.mT @ v) of the new V for the next U update.
// ------------------------------------------------------------------------------------------------
template <int RMAX>
__global__ __launch_bounds__(256) void k_vupdate(const PlaneDesc* __restrict__ planes, const float* __restrict__ Ppart,
                                                 const float* __restrict__ Qpart, float* __restrict__ Vf,
                                                 float* __restrict__ Bf, int8_t* __restrict__ V8, GsParams gp,
                                                 int write_i8, int plane0 /* first plane of this launch's run */)
{
    const int pli = blockIdx.x + plane0;
    __shared__ __attribute__((aligned(16))) float gt_s[LRF_GT_STRIDE];
    __shared__ __attribute__((aligned(16))) float a_s[64 * LRF_RP];
    __shared__ __attribute__((aligned(16))) float v_s[64 * LRF_RP];

    const PlaneDesc pd = planes[pli];
    const int R = pd.R, tid = threadIdx.x;
    // a' = ((P0 + P1) + P2) + ... and b' likewise: all the partials of up to 16 blocks (4 elements of a' and one of b'
    // per thread) are requested before the first one is used — one exposed memory latency per 16 blocks — and then
    // added in block order.
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, q = 0.f;
    {
        const float* Pp = Ppart + (long)pd.blk0 * 64 * LRF_RP + tid;
        const float* Qp = Qpart + (long)pd.blk0 * LRF_RP * LRF_RP + tid;
        const float vold[4] = {Vf[(long)pli * 64 * LRF_RP + tid], Vf[(long)pli * 64 * LRF_RP + 256 + tid],
                               Vf[(long)pli * 64 * LRF_RP + 512 + tid], Vf[(long)pli * 64 * LRF_RP + 768 + tid]};
        for (int b0 = 0; b0 < pd.nblk; b0 += 16) {
            float pv[4][16], qv[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const bool ok = b0 + k < pd.nblk;
                const long blk = ok ? b0 + k : b0;
#pragma unroll
                for (int e = 0; e < 4; e++) pv[e][k] = Pp[blk * 64 * LRF_RP + 256 * e];
                qv[k] = Qp[blk * LRF_RP * LRF_RP];
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (b0 + k < pd.nblk) {
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = (b0 + k == 0) ? pv[e][k] : acc[e] + pv[e][k];
                    q = (b0 + k == 0) ? qv[k] : q + qv[k];
                }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            a_s[256 * e + tid] = acc[e];
            v_s[256 * e + tid] = vold[e];
        }
    }
    {
        int j = tid >> 4, r = tid & 15;
        if (j < R && r < R) {
            if (j == r) {
                float den = (q + 0.f) + LRF_EPS;
                gt_s[r * LRF_GT_LD + LRF_GT_DEN] = den;
                gt_s[r * LRF_GT_LD + LRF_GT_RDEN] = 1.0f / den;
            } else {
                gt_s[r * LRF_GT_LD + (j < r ? j : j - 1)] = q;
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        bool native = (long)(R - 1) * 64 < 400;
        const float no_tab[17] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // FROM_I8 = false: unused
        gs_dispatch<RMAX, false>(R, &a_s[tid * LRF_RP], &v_s[tid * LRF_RP], nullptr, 0, gt_s, native, gp, no_tab);
        float* Vp = Vf + (long)pli * 64 * LRF_RP + tid * LRF_RP;
        for (int r = 0; r < R; r++) Vp[r] = v_s[tid * LRF_RP + r];
        if (write_i8) {
            int8_t* vo = V8 + pd.v_off + (long)tid * R;
            for (int r = 0; r < R; r++) vo[r] = (int8_t)v_s[tid * LRF_RP + r];
        }
    }
    __syncthreads();
    if (!write_i8) make_gtable(v_s, 64, R, Bf + (long)pli * LRF_GT_STRIDE, tid, 256);
}

// loads caller-supplied fp32 V0 [B][64][R] into the padded Vf table (lrf_qmf_bcd_f32); rp = padded rank
__global__ void k_load_v0(const PlaneDesc* __restrict__ planes, const float* __restrict__ V0, float* __restrict__ Vf, int rp)
{
    const PlaneDesc pd = planes[blockIdx.x];
    for (int i = threadIdx.x; i < 64 * rp; i += blockDim.x) {
        int j = i / rp, r = i - j * rp;
        Vf[(long)blockIdx.x * 64 * rp + i] = (r < pd.R) ? V0[pd.v0_off + (long)j * pd.R + r] : 0.f;
    }
}

// u0 = X @ W0 (fp32 out), v0 from Vf: lrf_qmf_svd_init_f32
__global__ __launch_bounds__(256) void k_emit_init(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                                   const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                                   const float* __restrict__ Wf, float* __restrict__ U0,
                                                   float* __restrict__ V0, int rp)
{
    const BlockDesc bd = blocks[blockIdx.x];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const float* Wp = Wf + (long)bd.plane * 64 * rp;
    for (int i = threadIdx.x; i < nrows * R; i += 256) {
        int m = i / R, r = i - m * R;
        const float* x = X + pd.x_off + (long)(bd.row0 + m) * 64;
        float acc = 0.f;
        for (int k = 0; k < 64; k++) acc = fmaf(x[k], Wp[k * rp + r], acc);
        U0[pd.u0_off + (long)(bd.row0 + m) * R + r] = acc;
    }
    if (bd.blk == 0)
        for (int i = threadIdx.x; i < 64 * R; i += 256) {
            int j = i / R, r = i - j * R;
            V0[pd.v0_off + i] = Vf[(long)bd.plane * 64 * rp + j * rp + r];
        }
}

// ------------------------------------------------------------------------------------------------
// K4: int8 factors -> uint8 RGB.  One thread per 4 horizontally adjacent pixels.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float recon_at(const int8_t* __restrict__ Uc, const int8_t* __restrict__ Vc, int R,
                                          const PlaneGeom& pg, int y, int x)
{
    // depatchify(unpad): padded coordinates, patch index, element index
    int yy = y + pg.top_crop, xx = x + pg.left_crop;
    int m = (yy >> 3) * pg.nw + (xx >> 3), n = (yy & 7) * 8 + (xx & 7);
    float acc = 0.f; // u @ v.mT : k-ordered fma chain (exact: small integers)
    for (int r = 0; r < R; r++) acc = fmaf((float)Uc[(long)m * R + r], (float)Vc[n * R + r], acc);
    return acc;
}

// Fast path for ranks <= 8 (the default sweeps): the three V tables of the image sit in LDS as floats (zero padded
// to 8 columns), a thread keeps the u row of the patch it is in and reloads it only when the patch changes (four
// horizontally adjacent pixels touch at most two luma and two chroma patches), and the four output bytes of a
// channel leave as one dword.  Same arithmetic and order as k_decode.
__global__ __launch_bounds__(256) void k_decode8(const int8_t* __restrict__ U, const int8_t* __restrict__ V, int H, int W,
                                                 ImageGeom g, int R0, int R1, int R2, long u_img, long v_img,
                                                 uint8_t* __restrict__ rgb, int reps)
{
    __shared__ float Vs[3][64 * 8];
    const int8_t* Ui = U + (long)blockIdx.y * u_img;
    const int8_t* Vi = V + (long)blockIdx.y * v_img;
    const int8_t* Uc[3] = {Ui, Ui + (long)g.p[0].M * R0, Ui + (long)g.p[0].M * R0 + (long)g.p[1].M * R1};
    const int8_t* Vc[3] = {Vi, Vi + 64 * R0, Vi + 64 * R0 + 64 * R1};
    const int Rc[3] = {R0, R1, R2};
    for (int e = threadIdx.x; e < 3 * 64 * 8; e += 256) {
        int c = e >> 9, n = (e >> 3) & 63, r = e & 7;
        Vs[c][n * 8 + r] = (r < Rc[c]) ? (float)Vc[c][n * Rc[c] + r] : 0.f;
    }
    __syncthreads();
    int w4 = (W + 3) >> 2;
    // `reps` groups of four pixels per thread (the host picks up to 16 for large calls): the V table above (six dependent byte
    // loads per thread and a barrier) is then staged once for up to 16384 pixels instead of 1024 — with one group per thread
    // that prologue, not the arithmetic, set the pace (512 x 1365x2048: 4.8 -> 3.4 ms)
    for (int rep = 0; rep < reps; rep++) {
    long o = ((long)blockIdx.x * reps + rep) * 256 + threadIdx.x;
    if (o >= (long)H * w4) return;
    int y = (int)(o / w4), x0 = (int)(o - (long)y * w4) * 4;
    uint8_t* out = rgb + (long)blockIdx.y * 3 * H * W;
    float sh = (float)g.p[1].h / (float)H, sw = (float)g.p[1].w / (float)W;
    int sy = (int)floorf((float)y * sh);
    if (sy > g.p[1].h - 1) sy = g.p[1].h - 1;
    float ur[3][8];
    int mcur[3] = {-1, -1, -1};
    auto recon = [&](int c, int yy, int xx) { // padded coordinates of plane c
        const int m = (yy >> 3) * g.p[c].nw + (xx >> 3), n = (yy & 7) * 8 + (xx & 7);
        if (m != mcur[c]) {
            mcur[c] = m;
#pragma unroll
            for (int r = 0; r < 8; r++) ur[c][r] = (r < Rc[c]) ? (float)Uc[c][(long)m * Rc[c] + r] : 0.f;
        }
        float acc = 0.f; // k-ordered fma chain; the padded terms are fma(0, 0, acc) = acc
#pragma unroll
        for (int r = 0; r < 8; r++) acc = fmaf(ur[c][r], Vs[c][n * 8 + r], acc);
        return acc;
    };
    uint32_t packed[3] = {0u, 0u, 0u};
    const int yyl = y + g.p[0].top_crop, yyc = sy + g.p[1].top_crop;
    int psx = -1;
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int x = x0 + i;
        if (x >= W) break;
        int sx = (int)floorf((float)x * sw);
        if (sx > g.p[1].w - 1) sx = g.p[1].w - 1;
        const float c0 = recon(0, yyl, x + g.p[0].left_crop) + 0.f;
        if (sx != psx) { // neighbouring pixels mostly share their chroma sample
            psx = sx;
            c1 = recon(1, yyc, sx + g.p[1].left_crop) + -128.f;
            c2 = recon(2, yyc, sx + g.p[2].left_crop) + -128.f;
        }
        // the colour chain of k_decode (k-ordered fmas from 0) without the steps that cannot change a bit for finite values:
        // fma(1, c0, 0) = c0, fma(0, c, acc) = acc; clamp + truncation as one v_med3 + conversion
        const float chv[3] = {fmaf(1.402f, c2, c0), fmaf(-0.714136f, c2, fmaf(-0.344136f, c1, c0)), fmaf(1.772f, c1, c0)};
#pragma unroll
        for (int ch = 0; ch < 3; ch++) packed[ch] |= (uint32_t)__builtin_amdgcn_fmed3f(chv[ch], 0.f, 255.f) << (8 * i);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        uint8_t* dst = out + (long)ch * H * W + (long)y * W + x0;
        if (x0 + 3 < W) {
            *reinterpret_cast<uint32_t __attribute__((aligned(1)))*>(dst) = packed[ch];
        } else {
            for (int i = 0; x0 + i < W; i++) dst[i] = (uint8_t)(packed[ch] >> (8 * i));
        }
    }
    }
}

// Fast path of K4 for images whose sides are multiples of 16 (no padding, no crop, exact 2x nearest
// up-sampling): the inverse of k_planes16's tiling.  One workgroup per 16-row strip x 32 luma patches; a thread owns a
// 2 x 8 pixel block — two rows of one luma patch, and the four chroma samples under them, one row of one chroma patch —
// so it loads its three u rows once (two unaligned dwords each), reads V from an LDS table laid out [plane][r][n] (a
// ds_read_b128 per (r, row)) and writes six 8-byte pieces; the 32 lanes of a row pair cover 256 contiguous bytes per store
// instruction.  44-49 registers, eight waves per SIMD.  The kernel is bound by its vector instructions (counters, 256 x
// 512x768: VALU busy 73 % of the 0.106 ms, 490 instructions per 16 pixels before the trimming below, 380 after; LDS 27 %);
// tried and not kept: two patches or eight strips per thread to reuse the V values (the compiler holds them in 160
// registers, two waves per SIMD: 1.4-2x slower), v_pk_fma_f32 for the sums (counted and paced as two instructions: no
// gain), 64 lanes per image row (512-byte store segments, but a quarter of the lanes idle at 96 patches per row: slower).
// Arithmetic: that of k_decode8 / k_decode (sums of products of small integers: exact in any order; "+ -128.f"; the colour
// chain; clamp; truncate) with the steps dropped that cannot change a bit for finite values: fma(1, y, 0) = y,
// fma(0, c, acc) = acc.  Template parameters: the rank bounds of the chroma planes (4, 8, 16) and of luma (8, 16, 32) — table
// sizes and loop lengths; the u rows are zero padded to them (round 3: until then ranks above 8 fell to the one-thread-per-
// four-pixels kernel k_decode, 13-16x slower: 256 x 512x768 at ranks (16,8,8) 2.31 -> 0.146 ms, (26,13,13) 3.74 -> 0.227).
// The int8 row of R <= RM bytes (RM = 8, 16, 32) as RM / 4 dwords, bytes past R zero: dword d comes from offset 4 d while it lies
// inside the row, the partial last one from offset R - 4 (unaligned, overlapping) shifted down; R < 4: byte loads.  R is
// uniform across the wave (one plane per call), so the branches are scalar.
template <int RM>
__device__ __forceinline__ void decode_u_load(const int8_t* up, int R, unsigned (&w)[RM / 4])
{
#pragma unroll
    for (int d = 0; d < RM / 4; d++) w[d] = 0u;
    if (R < 4) {
        w[0] = (unsigned)(uint8_t)up[0] | ((unsigned)(uint8_t)up[R > 1 ? 1 : 0] << 8) | ((unsigned)(uint8_t)up[R > 2 ? 2 : 0] << 16);
        if (R < 3) w[0] &= (R == 1) ? 0xffu : 0xffffu;
        return;
    }
#pragma unroll
    for (int d = 0; d < RM / 4; d++) {
        if (4 * d + 4 <= R) w[d] = *reinterpret_cast<const unsigned __attribute__((aligned(1)))*>(up + 4 * d);
        else if (4 * d < R) w[d] = *reinterpret_cast<const unsigned __attribute__((aligned(1)))*>(up + R - 4) >> (8 * (4 * d + 4 - R));
    }
}
template <int RM>
__device__ __forceinline__ void decode_u_unpack(const unsigned (&w)[RM / 4], float (&u)[RM])
{
#pragma unroll
    for (int r = 0; r < RM; r++) u[r] = (float)(int)(int8_t)(w[r >> 2] >> (8 * (r & 3)));
}

// clamp to [0, 255] and truncate four values, packed into one dword (values are finite)
__device__ __forceinline__ unsigned decode16_pack4(float a, float b, float c, float d)
{
    const unsigned ua = (unsigned)__builtin_amdgcn_fmed3f(a, 0.f, 255.f), ub = (unsigned)__builtin_amdgcn_fmed3f(b, 0.f, 255.f);
    const unsigned uc = (unsigned)__builtin_amdgcn_fmed3f(c, 0.f, 255.f), ud = (unsigned)__builtin_amdgcn_fmed3f(d, 0.f, 255.f);
    return (ua | (ub << 8)) | ((uc | (ud << 8)) << 16);
}

template <int RC, int RL> // rank bounds of the chroma planes (4, 8, 16) and of luma (8, 16, 32): table sizes and loop lengths
__global__ __launch_bounds__(256) void k_decode16(const int8_t* __restrict__ U, const int8_t* __restrict__ V, int H, int W,
                                                  ImageGeom g, int R0, int R1, int R2, long u_img, long v_img,
                                                  uint8_t* __restrict__ rgb)
{
    __shared__ __attribute__((aligned(16))) float VsL[RL][64], VsC[2][RC][64];
    const int8_t* Ui = U + (long)blockIdx.y * u_img;
    const int8_t* Vi = V + (long)blockIdx.y * v_img;
    const int8_t* Uc[3] = {Ui, Ui + (long)g.p[0].M * R0, Ui + (long)g.p[0].M * R0 + (long)g.p[1].M * R1};
    const int8_t* Vc[3] = {Vi, Vi + 64 * R0, Vi + 64 * R0 + 64 * R1};
    const int Rc[3] = {R0, R1, R2};
    const int nwl = g.p[0].nw, nwc = g.p[1].nw;
    const int per_strip = (nwl + 31) / 32;
    const int strip = blockIdx.x / per_strip;
    const int ww = (blockIdx.x - strip * per_strip) * 32 + (threadIdx.x & 31);
    const int wwc = ww < nwl ? ww : nwl - 1; // threads past the last patch load what the last one loads and store nothing
    const int rp = threadIdx.x >> 5; // row pair inside the strip: image rows 16 strip + 2 rp, + 1
    // the u rows of the luma patch and of the two chroma patches: issued before the V table is staged, so that the two
    // memory round trips of a workgroup overlap
    unsigned wl[RL / 4], wb[RC / 4], wr[RC / 4];
    const long mrow[3] = {(long)(2 * strip + (rp >> 2)) * nwl + wwc, (long)strip * nwc + (wwc >> 1), (long)strip * nwc + (wwc >> 1)};
    decode_u_load<RL>(Uc[0] + mrow[0] * R0, R0, wl);
    decode_u_load<RC>(Uc[1] + mrow[1] * R1, R1, wb);
    decode_u_load<RC>(Uc[2] + mrow[2] * R2, R2, wr);
    for (int e = threadIdx.x; e < RL * 64; e += 256) {
        const int r = e >> 6, n = e & 63;
        VsL[r][n] = (r < R0) ? (float)Vc[0][n * R0 + r] : 0.f;
    }
    for (int e = threadIdx.x; e < 2 * RC * 64; e += 256) {
        const int c = e / (RC * 64), r = (e >> 6) % RC, n = e & 63;
        VsC[c][r][n] = (r < Rc[1 + c]) ? (float)Vc[1 + c][n * Rc[1 + c] + r] : 0.f;
    }
    __syncthreads();
    if (ww >= nwl) return;
    float ul[RL], ub[RC], ur[RC]; // zero padded to the rank bounds
    decode_u_unpack<RL>(wl, ul);
    decode_u_unpack<RC>(wb, ub);
    decode_u_unpack<RC>(wr, ur);
    // chroma: samples (row 8 strip + rp of the plane = row rp of the patch, columns 4 (ww & 1) .. + 3)
    float cb[4] = {0.f, 0.f, 0.f, 0.f}, cr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RC; r++) {
        const f32x4 vb = *reinterpret_cast<const f32x4*>(&VsC[0][r][rp * 8 + 4 * (ww & 1)]);
        const f32x4 vr = *reinterpret_cast<const f32x4*>(&VsC[1][r][rp * 8 + 4 * (ww & 1)]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            cb[i] = fmaf(ub[r], vb[i], cb[i]);
            cr[i] = fmaf(ur[r], vr[i], cr[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        cb[i] = cb[i] + -128.f;
        cr[i] = cr[i] + -128.f;
    }
    const long hw = (long)H * W;
    uint8_t* out = rgb + (long)blockIdx.y * 3 * hw + (long)(16 * strip + 2 * rp) * W + 8 * ww;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int n0 = (2 * (rp & 3) + rr) * 8;
#pragma unroll
        for (int r = 0; r < RL; r++) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&VsL[r][n0]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&VsL[r][n0 + 4]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                y[i] = fmaf(ul[r], v0[i], y[i]);
                y[4 + i] = fmaf(ul[r], v1[i], y[4 + i]);
            }
        }
        uint2 pk[3];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float ch[3][4];
#pragma unroll
            for (int i = 0; i < 4; i++) { // the k-ordered chains without their no-op steps
                const float c0 = y[4 * h + i], c1 = cb[2 * h + (i >> 1)], c2 = cr[2 * h + (i >> 1)];
                ch[0][i] = fmaf(1.402f, c2, c0);
                ch[1][i] = fmaf(-0.714136f, c2, fmaf(-0.344136f, c1, c0));
                ch[2][i] = fmaf(1.772f, c1, c0);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const unsigned w = decode16_pack4(ch[k][0], ch[k][1], ch[k][2], ch[k][3]);
                if (h == 0) pk[k].x = w; else pk[k].y = w;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) *reinterpret_cast<uint2*>(out + k * hw + (long)rr * W) = pk[k];
    }
}

// k_decode16's tiling for every height and for the widths where the four chroma samples under a thread's eight pixels are
// the four aligned columns of one chroma patch (W even, left pad even, left-pad difference a multiple of four: e.g. every
// W that is a multiple of 16, whatever H — CLIC's 1365 x 2048): the tiles lie over the PADDED luma plane, a thread owns
// padded luma rows 16 s + 2 rp (+ 1) of one patch and stores the image rows / columns that survive the centre crop.  The
// nearest-neighbour chroma row of an image row y is min(floor(y * (h_c / H)), h_c - 1) in fp32 (F.interpolate "nearest",
// lrf/compression/utils.py:98-105): the two rows of a thread usually share it (then the chroma sums are computed once, as
// in k_decode16), otherwise the second row's are computed separately, from the u rows of its own chroma patch.
// Same arithmetic as k_decode8 / k_decode16.
template <int RC, int RL> // rank bounds of the chroma planes (4, 8, 16) and of luma (8, 16, 32)
__global__ __launch_bounds__(256) void k_decode_strip(const int8_t* __restrict__ U, const int8_t* __restrict__ V, int H, int W,
                                                      ImageGeom g, int R0, int R1, int R2, long u_img, long v_img,
                                                      uint8_t* __restrict__ rgb, int per_strip)
{
    __shared__ __attribute__((aligned(16))) float VsL[RL][64], VsC[2][RC][64];
    const int8_t* Ui = U + (long)blockIdx.y * u_img;
    const int8_t* Vi = V + (long)blockIdx.y * v_img;
    const int8_t* Uc[3] = {Ui, Ui + (long)g.p[0].M * R0, Ui + (long)g.p[0].M * R0 + (long)g.p[1].M * R1};
    const int8_t* Vc[3] = {Vi, Vi + 64 * R0, Vi + 64 * R0 + 64 * R1};
    const int Rc[3] = {R0, R1, R2};
    const PlaneGeom pl = g.p[0], pc = g.p[1];
    const int nwl = pl.nw, nwc = pc.nw;
    const int strip = blockIdx.x / per_strip;
    const int ww = (blockIdx.x - strip * per_strip) * 32 + (threadIdx.x & 31);
    const int rp = threadIdx.x >> 5; // row pair inside the strip: padded luma rows 16 strip + 2 rp, + 1
    const int prow = 2 * strip + (rp >> 2);
    const bool live = ww < nwl && prow < pl.nh;
    const int wwc = ww < nwl ? ww : nwl - 1, prc = prow < pl.nh ? prow : pl.nh - 1;
    // image rows / columns of this thread and their chroma samples (padded chroma coordinates)
    const int y0 = 16 * strip + 2 * rp - pl.top_crop, x0 = 8 * wwc - pl.left_crop;
    const float sh = (float)pc.h / (float)H;
    int q[2];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        int y = y0 + rr;
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        int sy = (int)floorf((float)y * sh);
        sy = sy > pc.h - 1 ? pc.h - 1 : sy;
        q[rr] = sy + pc.top_crop;
    }
    int cx = (x0 >> 1) + pc.left_crop; // x0 is even, cx a multiple of four (host check); columns cropped away are clamped
    cx = cx < 0 ? 0 : (cx > pc.wp - 4 ? pc.wp - 4 : cx);
    const long mc0 = (long)(q[0] >> 3) * nwc + (cx >> 3), mc1 = (long)(q[1] >> 3) * nwc + (cx >> 3);
    // the u rows of the luma patch and of the chroma patches of the first row: issued before the V table is staged
    unsigned wl[RL / 4], wb[RC / 4], wr[RC / 4];
    decode_u_load<RL>(Uc[0] + ((long)prc * nwl + wwc) * R0, R0, wl);
    decode_u_load<RC>(Uc[1] + mc0 * R1, R1, wb);
    decode_u_load<RC>(Uc[2] + mc0 * R2, R2, wr);
    for (int e = threadIdx.x; e < RL * 64; e += 256) {
        const int r = e >> 6, n = e & 63;
        VsL[r][n] = (r < R0) ? (float)Vc[0][n * R0 + r] : 0.f;
    }
    for (int e = threadIdx.x; e < 2 * RC * 64; e += 256) {
        const int c = e / (RC * 64), r = (e >> 6) % RC, n = e & 63;
        VsC[c][r][n] = (r < Rc[1 + c]) ? (float)Vc[1 + c][n * Rc[1 + c] + r] : 0.f;
    }
    __syncthreads();
    if (!live) return;
    float ul[RL], ub[RC], ur[RC]; // zero padded to the rank bounds
    decode_u_unpack<RL>(wl, ul);
    decode_u_unpack<RC>(wb, ub);
    decode_u_unpack<RC>(wr, ur);
    float cb[4], cr[4];
    auto chroma = [&](int qq) { // samples (padded row qq, padded columns cx .. cx + 3) of both planes, "+ -128.f"
#pragma unroll
        for (int i = 0; i < 4; i++) cb[i] = cr[i] = 0.f;
        const int nc = (qq & 7) * 8 + (cx & 7);
#pragma unroll
        for (int r = 0; r < RC; r++) {
            const f32x4 vb = *reinterpret_cast<const f32x4*>(&VsC[0][r][nc]);
            const f32x4 vr = *reinterpret_cast<const f32x4*>(&VsC[1][r][nc]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                cb[i] = fmaf(ub[r], vb[i], cb[i]);
                cr[i] = fmaf(ur[r], vr[i], cr[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            cb[i] = cb[i] + -128.f;
            cr[i] = cr[i] + -128.f;
        }
    };
    chroma(q[0]);
    const long hw = (long)H * W;
    uint8_t* out = rgb + (long)blockIdx.y * 3 * hw;
    const bool xfull = x0 >= 0 && x0 + 8 <= W;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int y = y0 + rr;
        if (rr == 1 && q[1] != q[0]) { // the second row sits over another chroma row (possibly of the next chroma patch row)
            if (mc1 != mc0) {
                decode_u_load<RC>(Uc[1] + mc1 * R1, R1, wb);
                decode_u_load<RC>(Uc[2] + mc1 * R2, R2, wr);
                decode_u_unpack<RC>(wb, ub);
                decode_u_unpack<RC>(wr, ur);
            }
            chroma(q[1]);
        }
        if (y < 0 || y >= H) continue;
        float yv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int n0 = (2 * (rp & 3) + rr) * 8;
#pragma unroll
        for (int r = 0; r < RL; r++) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&VsL[r][n0]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&VsL[r][n0 + 4]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                yv[i] = fmaf(ul[r], v0[i], yv[i]);
                yv[4 + i] = fmaf(ul[r], v1[i], yv[4 + i]);
            }
        }
        uint2 pk[3];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float ch[3][4];
#pragma unroll
            for (int i = 0; i < 4; i++) { // the k-ordered chains without their no-op steps
                const float c0 = yv[4 * h + i], c1 = cb[2 * h + (i >> 1)], c2 = cr[2 * h + (i >> 1)];
                ch[0][i] = fmaf(1.402f, c2, c0);
                ch[1][i] = fmaf(-0.714136f, c2, fmaf(-0.344136f, c1, c0));
                ch[2][i] = fmaf(1.772f, c1, c0);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const unsigned w = decode16_pack4(ch[k][0], ch[k][1], ch[k][2], ch[k][3]);
                if (h == 0) pk[k].x = w; else pk[k].y = w;
            }
        }
        uint8_t* dst = out + (long)y * W + x0;
        if (xfull) {
#pragma unroll
            for (int k = 0; k < 3; k++) *reinterpret_cast<uint2 __attribute__((aligned(1)))*>(dst + k * hw) = pk[k];
        } else { // the crop cuts this thread's run: byte by byte
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const unsigned long long w = ((unsigned long long)pk[k].y << 32) | pk[k].x;
                for (int j = 0; j < 8; j++)
                    if (x0 + j >= 0 && x0 + j < W) dst[k * hw + j] = (uint8_t)(w >> (8 * j));
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_decode(const int8_t* __restrict__ U, const int8_t* __restrict__ V, int H, int W,
                                                ImageGeom g, int R0, int R1, int R2, long u_img, long v_img,
                                                uint8_t* __restrict__ rgb)
{
    int w4 = (W + 3) >> 2;
    long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long)H * w4) return;
    int y = (int)(o / w4), x0 = (int)(o - (long)y * w4) * 4;
    const int8_t* Ui = U + (long)blockIdx.y * u_img;
    const int8_t* Vi = V + (long)blockIdx.y * v_img;
    const int8_t* Uc[3] = {Ui, Ui + (long)g.p[0].M * R0, Ui + (long)g.p[0].M * R0 + (long)g.p[1].M * R1};
    const int8_t* Vc[3] = {Vi, Vi + 64 * R0, Vi + 64 * R0 + 64 * R1};
    const int Rc[3] = {R0, R1, R2};
    uint8_t* out = rgb + (long)blockIdx.y * 3 * H * W;
    const float T[3][3] = {{1.0f, 0.0f, 1.402f}, {1.0f, -0.344136f, -0.714136f}, {1.0f, 1.772f, 0.0f}};
    // nearest up-sampling source rows/cols (ATen: floor(dst * (in/out)) in fp32, clamped)
    float sh = (float)g.p[1].h / (float)H, sw = (float)g.p[1].w / (float)W;
    int sy = (int)floorf((float)y * sh);
    if (sy > g.p[1].h - 1) sy = g.p[1].h - 1;
    for (int i = 0; i < 4; i++) {
        int x = x0 + i;
        if (x >= W) break;
        int sx = (int)floorf((float)x * sw);
        if (sx > g.p[1].w - 1) sx = g.p[1].w - 1;
        float c[3];
        c[0] = recon_at(Uc[0], Vc[0], Rc[0], g.p[0], y, x) + 0.f;
        c[1] = recon_at(Uc[1], Vc[1], Rc[1], g.p[1], sy, sx) + -128.f;
        c[2] = recon_at(Uc[2], Vc[2], Rc[2], g.p[2], sy, sx) + -128.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float acc = 0.f;
            acc = fmaf(T[ch][0], c[0], acc);
            acc = fmaf(T[ch][1], c[1], acc);
            acc = fmaf(T[ch][2], c[2], acc);
            acc = fminf(fmaxf(acc, 0.f), 255.f);
            out[(long)ch * H * W + (long)y * W + x] = (uint8_t)acc; // truncation (to_dtype)
        }
    }
}
