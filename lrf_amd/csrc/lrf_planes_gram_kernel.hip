#pragma once
// lrf_planes_gram_kernel.hip — k_planes16_gram: k_planes16 (patch matrices of images whose sides are multiples of 16,
// lrf/compression/utils.py:24-47,76-95,108-132; qmf.py:43-56) and the LUMA half of k_gram64 (the exact Gram matrix, input of the
// SVD initialisation, lrf/factorization/qmf.py:42-48) in one kernel.  Included by lrf_encode8.hip after lrf_gram_kernels.hip and
// lrf_kernels.hip.
//
// k_planes16 stages the 64 luma patches of a (16-row strip, 32-patch segment) unit in LDS before it writes them out — exactly
// one 64-row block of k_gram64.  Here a workgroup owns one Gram chunk of a luma plane, walks the units that make up that chunk
// and, while a unit's rows sit in LDS, takes the Gram digits from there: the 403 MB of luma X (256 x 512x768) are written once
// and not read back by a Gram pass (k_gram64 then runs over the chroma planes only: a third of its rows).  The sums are exact
// integers: which rows a chunk's partial covers, and in which order, does not change a bit of what k_init adds up, so a chunk
// is "units [g U / n, (g + 1) U / n) of the image" here and "rows [row0, row0 + nrows)" there.
// The RGB bytes of the next unit but one are in flight while a unit is processed (k_planes16 hides that latency with eight
// workgroups per CU; this kernel has two).
#include "lrf_device.h"
#include "lrf_internal.h"

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_planes16_gram(
    const uint8_t* __restrict__ rgb, int H, int W, ImageGeom g, float* __restrict__ X, const PlaneDesc* __restrict__ planes,
    const GramChunk* __restrict__ chunks, ulonglong2* __restrict__ Gpart)
{
    // luma staging as in k_planes16: [h = patch row 0/1 of the strip][patch 0..31][16 float4], float4 slot q stored at q ^ swz(h, q);
    // two buffers, and two of k_gram64's operand tiles [tile][digit][lane]: ONE barrier per unit (below)
    __shared__ __attribute__((aligned(16))) float Lsb[2][2 * 32 * 64];
    __shared__ uint4 lds[2][4 * 5 * 64];
    const GramChunk ch = chunks[blockIdx.x];
    const PlaneDesc pd = planes[ch.plane]; // a luma plane: x_off = image * img_floats + g.p[0].xoff
    const long image = (pd.x_off - g.p[0].xoff) / g.img_floats;
    const int hw = H * W, nwl = g.p[0].nw, nwc = g.p[1].nw;
    const int per_strip = (nwl + 31) / 32;
    const int nunits = (H / 16) * per_strip;
    const int gi = ch.slot - pd.gch0;
    const int u0 = (int)((long)gi * nunits / pd.ngch), u1 = (int)((long)(gi + 1) * nunits / pd.ngch);
    const uint8_t* img = rgb + image * 3 * (long)hw;
    float* Xi = X + image * g.img_floats;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int wwl = tid >> 3, rp = tid & 7; // 8-pixel column block of the segment, row pair inside the strip

    i32x4 acc[3][9];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int w = 0; w < 9; w++) acc[p][w] = (i32x4){0, 0, 0, 0};

    uint64_t chn[3][2];
    auto issue_rgb = [&](int unit) {
        const int strip = unit / per_strip, ww = (unit - strip * per_strip) * 32 + wwl;
        const int wc = ww < nwl ? ww : nwl - 1; // (threads past the last patch column load a valid address and produce zeros)
        const int y = 16 * strip + 2 * rp, x = 8 * wc;
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++) chn[k][rr] = *reinterpret_cast<const uint64_t*>(img + k * hw + (y + rr) * W + x);
    };
    // k_planes16's arithmetic for this thread's 2 x 8 pixel block of `unit` (ycc_of; chroma window sums row-major): luma into the
    // staging buffer, chroma straight out (lanes of an even / odd column-block pair fill whole 32-byte rows)
    auto compute = [&](int unit, float* Ls) {
        const int strip = unit / per_strip;
        const int ww = (unit - strip * per_strip) * 32 + wwl;
        const bool live = ww < nwl;
        float* Lp = Ls + ((rp >> 2) * 32 + wwl) * 64;
        const int swz = (rp >> 1) & 3;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            f32x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o0[i] = ycc_of((float)((chn[0][rr] >> (8 * i)) & 255u), (float)((chn[1][rr] >> (8 * i)) & 255u),
                               (float)((chn[2][rr] >> (8 * i)) & 255u), 0);
                o1[i] = ycc_of((float)((chn[0][rr] >> (8 * (i + 4))) & 255u), (float)((chn[1][rr] >> (8 * (i + 4))) & 255u),
                               (float)((chn[2][rr] >> (8 * (i + 4))) & 255u), 0);
            }
            if (!live) { // a patch column past the plane: zero rows of the Gram block
                o0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                o1 = o0;
            }
            const int q = 4 * (rp & 3) + 2 * rr;
            *reinterpret_cast<f32x4*>(Lp + 4 * (q ^ swz)) = o0;
            *reinterpret_cast<f32x4*>(Lp + 4 * ((q + 1) ^ swz)) = o1;
        }
        if (live) {
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
                            sum = sum + ycc_of((float)((chn[0][rr] >> sh) & 255u), (float)((chn[1][rr] >> sh) & 255u),
                                               (float)((chn[2][rr] >> sh) & 255u), c);
                        }
                    o[i] = sum / 2.f / 2.f;
                }
                float* Cp = Xi + g.p[c].xoff + ((long)strip * nwc + (ww >> 1)) * 64 + rp * 8 + 4 * (ww & 1);
                *reinterpret_cast<f32x4*>(Cp) = o;
            }
        }
    };
    // One barrier per unit: behind barrier u - 1 a wave takes unit u's rows from Ls[u & 1] (digits into lds[u & 1], rows out to
    // X), forms unit u + 1 into Ls[(u + 1) & 1], meets the others at barrier u and runs unit u's MFMAs from lds[u & 1].  Ls[b] is
    // rewritten two barriers after its last reader left it, lds[b] likewise.
    if (u0 < u1) {
        issue_rgb(u0);
        compute(u0, Lsb[0]);
        if (u0 + 1 < u1) issue_rgb(u0 + 1);
        __syncthreads();
    }
    for (int unit = u0; unit < u1; unit++) {
        const int bsel = (unit - u0) & 1;
        const float* Ls = Lsb[bsel];
        const int strip = unit / per_strip;
        const int ww0 = (unit - strip * per_strip) * 32;
        // ---- the Gram digits of these 64 rows (k_gram64's loop body): lane (li, kq) of wave t takes column 16 t + li of the rows
        // 16 kq .. 16 kq + 15 of the tile — row r = 32 h + pw sits at Ls[r * 64 + 4 * (slot ^ swz) + (column & 3)], slot = column >> 2,
        // swz = (2 h + (slot >> 3)) & 3 as written by compute()
        unsigned pk[5][4];
        {
            const int col = 16 * wave + li, slot = col >> 2;
            const int hh = kq >> 1; // rows 16 kq .. 16 kq + 15 lie in patch row h = kq >> 1
            const int swz = ((hh << 1) | (slot >> 3)) & 3;
            const float* lp = Ls + (16 * kq) * 64 + 4 * (slot ^ swz) + (col & 3);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float x4[4] = {lp[(4 * q) * 64], lp[(4 * q + 1) * 64], lp[(4 * q + 2) * 64], lp[(4 * q + 3) * 64]};
                unsigned one[5];
                gram_digits4_planes(x4, one);
#pragma unroll
                for (int a = 0; a < 5; a++) pk[a][q] = one[a];
            }
        }
        uint4* lb = lds[bsel];
#pragma unroll
        for (int a = 0; a < 5; a++) lb[(wave * 5 + a) * 64 + lane] = make_uint4(pk[a][0], pk[a][1], pk[a][2], pk[a][3]);
        // ---- luma rows out, lane-contiguous (k_planes16's second phase)
        const int npw = nwl - ww0 < 32 ? nwl - ww0 : 32;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int f = k * 256 + tid, h = f >> 9, rem = f & 511, pw = rem >> 4, q = rem & 15;
            if (pw < npw) {
                const int sw = (h << 1) | (q >> 3);
                const f32x4 v = *reinterpret_cast<const f32x4*>(Ls + (h * 32 + pw) * 64 + 4 * (q ^ sw));
                *reinterpret_cast<f32x4*>(Xi + g.p[0].xoff + ((long)(2 * strip + h) * nwl + ww0) * 64 + rem * 4) = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the next unit into the other staging buffer (its bytes were requested a unit ago), the one after that requested
        if (unit + 1 < u1) { // (wave-uniform)
            compute(unit + 1, Lsb[bsel ^ 1]);
            if (unit + 2 < u1) issue_rgb(unit + 2);
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        gram_accumulate(lb, lane, wave, acc);
    }
    gram_finish<true>(acc, lds[0], lane, wave, Gpart + (long)ch.slot * LRF_GRAM_SLOT);
}
