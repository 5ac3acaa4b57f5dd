#pragma once
// lrf_bcdp_kernel.hip — k_bcd_p<F16, NP32, FIRST>: the iterations of a large call in ONE launch, for every rank family of the
// 64-column path (round 4: ranks <= 8, iterations 2..K; round 5: the planes of ranks 9..16 and 17..32 and their mixes — (16,8,8),
// (26,13,13), ... — too, and, FIRST, at ranks <= 16 the call's first iteration as well: all K in the launch).
// Included by lrf_bcd_persist.hip after the block bodies it repeats operation for operation — ranks <= 8: k_bcd_w<0>
// (lrf_bcdw_kernel.hip; restated here with sc1 accesses, bcdp_w_block), 9..16: w16_block (lrf_bcdw16_kernel.hip), 17..32:
// w32_block (lrf_bcdw32_kernel.hip) — so the outputs are bit-identical to the launch-per-iteration path.
//
// What it removes: per iteration two kernel boundaries per family, the V-update launches (latency chains per matrix that leave
// the chip idle, or — when the families of a call run on streams of their own — starve behind the other family's U update:
// k_vupdate<16> took 180 us per launch beside k_bcd_w32 at (26,13,13)) and the drain / refill of every U-update launch.  How:
//   * items (iteration, block) are pulled in iteration-major order from one device-scope counter by whatever waves are
//     resident (a plain grid; a wave that finds the queue empty leaves); a block runs the body of its plane's rank family;
//   * a block stores its partials of X^T u / u^T u, drains its stores, and takes a ticket of its matrix; the LAST arriver of
//     (matrix, iteration) performs that matrix's V update on its own wave (partial sums in block order, the 64-row
//     Gauss-Seidel lane = row, the b table of the new V: bcdp_vupdate / bcdp_vupdate16 / bcdp_vupdate32 restate k_vupdate<8>,
//     k_vupdate<16>, k_vupdate_mid) and publishes flag[matrix] = iteration + 1;
//   * a block of iteration i + 1 polls its matrix's flag before it loads V.  Queue order makes this deadlock-free without
//     any assumption on dispatch: an item of iteration i + 1 is pulled only after every item of iteration i has been pulled
//     by a running wave, and running waves of iteration i never wait for later items.  In steady state nobody spins: the
//     same matrix's next item comes a full round of the queue later.  Every poll loop is BOUNDED (LRF_BCDP_MAX_POLLS): on
//     expiry the wave sets the error word (host memory) and returns, so the grid always drains; the host refuses the result.
//   * visibility across CUs / XCDs (MI355X_MICROARCH.md, inter-workgroup visibility): everything one wave writes and another
//     reads inside the launch — int8 U rows, partials, V table, b table — is stored `sc1` (write-through) and loaded `sc1`
//     (L1 bypass; MemSc1, lrf_device.h), every storing wave drains (`s_waitcnt vmcnt(0)`) before its ticket / flag, tickets
//     and flags are agent-scope atomics.  X is read-only and loaded plainly.

#ifndef LRF_BCDP_MAX_POLLS
#define LRF_BCDP_MAX_POLLS (1 << 20) // x ~1 us per poll: a second, then the wave gives up (error word)
#endif

struct BcdpSync {
    int head;        // next item
    int done;        // waves that have left
    int err;         // a poll of some launch on this state has expired: later launches leave at once (the host clears it)
    int pad[29];     // (head on a line of its own)
    int cell[1];     // [nplanes] tickets, then [nplanes] flags
};
// The queue head, tickets and flags start at zero.  The buffer is zeroed when it is allocated (and after a failed launch);
// after that every launch leaves it zeroed: the last wave to leave (`done`) clears what the launch used — no clearing kernel
// and no copy of an error word per call (the error word lives in page-locked host memory the kernel writes to directly).
// A FAILED launch (a poll expired) writes its sequence number `seq` (>= 1, counted per context by the host) to that word and
// sets `err` in this state; launches queued behind it find `err` set and leave at once without touching anything, so the
// host word keeps the number of the FIRST failed launch: that call's results and those of every later one are invalid until
// the host has looked (lrf_ctx_check / lrf_ctx_synchronize / lrf_pipe_wait_next / the next call's entry) and cleared the state.

// the tables of one rank pitch: 16 (ranks <= 16: LRF_RP, gt pitch LRF_GT_LD) or 64 (ranks 17..32: LRF_RPB, LRF_GTB_LD)
struct BcdpTabs {
    float *vf, *bf, *pp, *qp;
    const float* wf; // the initialisation's W0 = V0 / sigma (k_bcd_p<.., .., true>: the first iteration's old U is X @ W0)
};

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v)
{
#ifdef LRF_BCDP_PLAIN_PSTORE // timing experiment only
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void st_sc1_u32_unaligned(void* p, unsigned v)
{
#ifdef LRF_BCDP_PLAIN_USTORE // timing experiment only
    *reinterpret_cast<u32_unaligned*>(p) = v;
#else
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ unsigned ld_sc1_u32_unaligned(const void* p) // the caller waits (s_waitcnt vmcnt(0)) before the use
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ---- V updates on ONE wave (lane = row of V).  lds: the wave's own share (>= 17 KB) --------------------------------------------
// a' = ((P0 + P1) + P2) + ... per element, b' likewise, at rank pitch 16: element e of the [64][16] table = lane + 64 j
// (j < 16), of the [16][16] table = lane + 64 j (j < 4); four blocks' loads in flight at a time
__device__ __forceinline__ void bcdp_sum16(const PlaneDesc& pd, const float* __restrict__ Ppart, const float* __restrict__ Qpart, int lane,
                                           float (&acc)[16], float (&q)[4])
{
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = 0.f;
    const float* Pp = Ppart + (long)pd.blk0 * 64 * LRF_RP + lane;
    const float* Qp = Qpart + (long)pd.blk0 * LRF_RP * LRF_RP + lane;
    constexpr int NB = 4; // blocks in flight: 80 registers (the kernel must stay inside 256 for two waves per SIMD)
    for (int b0 = 0; b0 < pd.nblk; b0 += NB) {
        float pv[NB][16], qv[NB][4];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const long blk = b0 + k < pd.nblk ? b0 + k : b0;
#pragma unroll
            for (int j = 0; j < 16; j++) pv[k][j] = ld_sc1(Pp + blk * 64 * LRF_RP + 64 * j);
#pragma unroll
            for (int j = 0; j < 4; j++) qv[k][j] = ld_sc1(Qp + blk * LRF_RP * LRF_RP + 64 * j);
        }
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (b0 + k < pd.nblk) {
#pragma unroll
                for (int j = 0; j < 16; j++) acc[j] = (b0 + k == 0) ? pv[k][j] : acc[j] + pv[k][j];
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = (b0 + k == 0) ? qv[k][j] : q[j] + qv[k][j];
            }
    }
}
__device__ __forceinline__ void bcdp_wave_sync() // LDS written by some lanes is read by others of the same wave
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ranks <= 8: k_vupdate<8> (a_s [64][16], v_s [64][16], gt_s)
__device__ __forceinline__ void bcdp_vupdate(const PlaneDesc& pd, int pli, const float* __restrict__ Ppart, const float* __restrict__ Qpart,
                                             float* __restrict__ Vf, float* __restrict__ Bf, int8_t* __restrict__ V8, const GsParams gp,
                                             int write_i8, float* lds, int lane)
{
    float* a_s = lds;
    float* v_s = lds + 64 * LRF_RP;
    float* gt_s = lds + 2 * 64 * LRF_RP;
    const int R = pd.R;
    float acc[16], q[4];
    bcdp_sum16(pd, Ppart, Qpart, lane, acc, q);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        a_s[lane + 64 * j] = acc[j];
        v_s[lane + 64 * j] = ld_sc1(Vf + (long)pli * 64 * LRF_RP + lane + 64 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int idx = lane + 64 * j, jj = idx >> 4, r = idx & 15; // b' entry (jj, r)
        if (jj < R && r < R) {
            if (jj == r) {
                const float den = (q[j] + 0.f) + LRF_EPS;
                gt_s[r * LRF_GT_LD + LRF_GT_DEN] = den;
                gt_s[r * LRF_GT_LD + LRF_GT_RDEN] = 1.0f / den;
            } else {
                gt_s[r * LRF_GT_LD + (jj < r ? jj : jj - 1)] = q[j];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const bool native = (long)(R - 1) * 64 < 400;
        const float no_tab[17] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        gs_dispatch<8, false>(R, &a_s[lane * LRF_RP], &v_s[lane * LRF_RP], nullptr, 0, gt_s, native, gp, no_tab);
        float* Vp = Vf + (long)pli * 64 * LRF_RP + lane * LRF_RP;
        for (int r = 0; r < R; r++) st_sc1(Vp + r, v_s[lane * LRF_RP + r]);
        if (write_i8) {
            int8_t* vo = V8 + pd.v_off + (long)lane * R;
            for (int r = 0; r < R; r++) vo[r] = (int8_t)v_s[lane * LRF_RP + r];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!write_i8) {
        // the b table of the new V: into LDS (make_gtable's stores), then out with write-through stores
        float* gt_n = a_s; // a_s is free now
        make_gtable(v_s, 64, R, gt_n, lane, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* gt_g = Bf + (long)pli * LRF_GT_STRIDE;
        for (int i = lane; i < R * LRF_GT_LD; i += 64) st_sc1(gt_g + i, gt_n[i]);
    }
}


// ranks 9..16: k_vupdate<16>.  lds: a_s [64][16], v_s [64][16], the b' table in the layout mid_ordered_row reads (pitch
// LRF_GTB_LD; 12.5 KB in all).  The ordered chain of a row runs in registers (mid_ordered_row: MKL's single-column order with
// the IEEE division — what gs_row's speculative reciprocal provably reproduces); (R - 1) 64 >= 400: never ATen's native order.
__device__ __forceinline__ void bcdp_vupdate16(const PlaneDesc& pd, int pli, const float* __restrict__ Ppart, const float* __restrict__ Qpart,
                                               float* __restrict__ Vf, float* __restrict__ Bf, int8_t* __restrict__ V8, const GsParams gp,
                                               int write_i8, float* lds, int lane)
{
    float* a_s = lds;
    float* v_s = lds + 64 * LRF_RP;
    float* gt_b = lds + 2 * 64 * LRF_RP; // [R][LRF_GTB_LD]
    const int R = pd.R;
    {
        float acc[16], q[4];
        bcdp_sum16(pd, Ppart, Qpart, lane, acc, q);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            a_s[lane + 64 * j] = acc[j];
            v_s[lane + 64 * j] = ld_sc1(Vf + (long)pli * 64 * LRF_RP + lane + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = lane + 64 * j, jj = idx >> 4, r = idx & 15; // b' entry (jj, r)
            if (jj < R && r < R) {
                if (jj == r) gt_b[r * LRF_GTB_LD + LRF_GTB_DEN] = (q[j] + 0.f) + LRF_EPS;
                else gt_b[r * LRF_GTB_LD + (jj < r ? jj : jj - 1)] = q[j];
            }
        }
    }
    bcdp_wave_sync();
    {
        float a[32], v[32];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            const f32x4 va4 = *reinterpret_cast<const f32x4*>(&a_s[lane * LRF_RP + j]);
            const f32x4 vv4 = *reinterpret_cast<const f32x4*>(&v_s[lane * LRF_RP + j]);
#pragma unroll
            for (int i = 0; i < 4; i++) { a[j + i] = va4[i]; v[j + i] = vv4[i]; }
        }
#pragma unroll
        for (int j = 16; j < 32; j++) a[j] = v[j] = 0.f;
        mid_ordered_row(R, a, v, gt_b, gp.lo, gp.hi, std::make_integer_sequence<int, 16>{});
        float* Vp = Vf + (long)pli * 64 * LRF_RP + lane * LRF_RP;
        int8_t* vo = V8 + pd.v_off + (long)lane * R;
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (r < R) {
                v_s[lane * LRF_RP + r] = v[r];
                st_sc1(Vp + r, v[r]);
                if (write_i8) vo[r] = (int8_t)v[r];
            }
    }
    bcdp_wave_sync();
    if (!write_i8) {
        float* gt_n = a_s; // a_s is free now
        make_gtable(v_s, 64, R, gt_n, lane, 64);
        bcdp_wave_sync();
        float* gt_g = Bf + (long)pli * LRF_GT_STRIDE;
        for (int i = lane; i < R * LRF_GT_LD; i += 64) st_sc1(gt_g + i, gt_n[i]);
    }
}

// ranks 17..32: k_vupdate_mid (rank pitch LRF_RPB; only the first 32 columns exist at these ranks).  lds: t_s [64][32] (a',
// then the old V, then the new V: a lane's row is contiguous there, the global tables are read and written element-wise, 128
// contiguous bytes per row pair) and the b' table [R][LRF_GTB_LD]: 16.9 KB.  Elements: lane + 64 m -> (row = e >> 5, col = e & 31).
__device__ __forceinline__ void bcdp_vupdate32(const PlaneDesc& pd, int pli, const float* __restrict__ Ppart, const float* __restrict__ Qpart,
                                               float* __restrict__ Vf, float* __restrict__ Bf, int8_t* __restrict__ V8, const GsParams gp,
                                               int write_i8, float* lds, int lane)
{
    float* t_s = lds;            // [64][32]
    float* gt_b = lds + 64 * 32; // [R][LRF_GTB_LD]
    const int R = pd.R;
    float a[32], v[32];
    {
        float acc[32], q[16];
#pragma unroll
        for (int m = 0; m < 32; m++) acc[m] = 0.f;
#pragma unroll
        for (int m = 0; m < 16; m++) q[m] = 0.f;
        const float* P0 = Ppart + (long)pd.blk0 * 64 * LRF_RPB + (lane >> 5) * LRF_RPB + (lane & 31);
        const float* Q0 = Qpart + (long)pd.blk0 * LRF_RPB * LRF_RPB + (lane >> 5) * LRF_RPB + (lane & 31);
        constexpr int NB = 2; // blocks in flight: 96 registers
        for (int b0 = 0; b0 < pd.nblk; b0 += NB) {
            float pv[NB][32], qv[NB][16];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const long blk = b0 + k < pd.nblk ? b0 + k : b0;
#pragma unroll
                for (int m = 0; m < 32; m++) pv[k][m] = ld_sc1(P0 + blk * 64 * LRF_RPB + 2 * m * LRF_RPB); // rows 2 m, 2 m + 1
#pragma unroll
                for (int m = 0; m < 16; m++) qv[k][m] = ld_sc1(Q0 + blk * LRF_RPB * LRF_RPB + 2 * m * LRF_RPB);
            }
#pragma unroll
            for (int k = 0; k < NB; k++)
                if (b0 + k < pd.nblk) {
#pragma unroll
                    for (int m = 0; m < 32; m++) acc[m] = (b0 + k == 0) ? pv[k][m] : acc[m] + pv[k][m];
#pragma unroll
                    for (int m = 0; m < 16; m++) q[m] = (b0 + k == 0) ? qv[k][m] : q[m] + qv[k][m];
                }
        }
#pragma unroll
        for (int m = 0; m < 32; m++) t_s[lane + 64 * m] = acc[m];
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int j = 2 * m + (lane >> 5), r = lane & 31; // b' = U^T U entry (j, r)
            if (j < R && r < R) {
                if (j == r) gt_b[r * LRF_GTB_LD + LRF_GTB_DEN] = (q[m] + 0.f) + LRF_EPS;
                else gt_b[r * LRF_GTB_LD + (j < r ? j : j - 1)] = q[m];
            }
        }
    }
    bcdp_wave_sync();
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&t_s[lane * 32 + j]);
#pragma unroll
        for (int i = 0; i < 4; i++) a[j + i] = t4[i];
    }
    bcdp_wave_sync();
    {
        const float* Vg = Vf + (long)pli * 64 * LRF_RPB + (lane >> 5) * LRF_RPB + (lane & 31);
        float vv[32];
#pragma unroll
        for (int m = 0; m < 32; m++) vv[m] = ld_sc1(Vg + 2 * m * LRF_RPB);
#pragma unroll
        for (int m = 0; m < 32; m++) t_s[lane + 64 * m] = vv[m];
    }
    bcdp_wave_sync();
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&t_s[lane * 32 + j]);
#pragma unroll
        for (int i = 0; i < 4; i++) v[j + i] = t4[i];
    }
    mid_ordered_row(R, a, v, gt_b, gp.lo, gp.hi, std::make_integer_sequence<int, 32>{});
    bcdp_wave_sync(); // every lane has read its old row
    {
        int8_t* vo = V8 + pd.v_off + (long)lane * R;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            t_s[lane * 32 + r] = r < R ? v[r] : 0.f;
            if (write_i8 && r < R) vo[r] = (int8_t)v[r];
        }
    }
    bcdp_wave_sync();
    {
        float* Vg = Vf + (long)pli * 64 * LRF_RPB + (lane >> 5) * LRF_RPB + (lane & 31);
        if ((lane & 31) < R) {
#pragma unroll
            for (int m = 0; m < 32; m++) st_sc1(Vg + 2 * m * LRF_RPB, t_s[lane + 64 * m]);
        }
    }
    if (!write_i8) {
        // the b table of the new V (make_gtable_big's arithmetic: one k-ordered fma chain per entry; 64 R R >= 400: never native)
        float* gt_g = Bf + (long)pli * LRF_GTB_STRIDE;
        for (int i = lane; i < R * R; i += 64) {
            const int j = i / R, r = i - j * R;
            float acc = 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; k++) acc = fmaf(t_s[k * 32 + j], t_s[k * 32 + r], acc);
            if (j == r) st_sc1(gt_g + r * LRF_GTB_LD + LRF_GTB_DEN, (acc + 0.f) + LRF_EPS);
            else st_sc1(gt_g + r * LRF_GTB_LD + (j < r ? j : j - 1), acc);
        }
    }
}

// ---- ranks <= 8: one (matrix, 384-row block) on one wave — k_bcd_w<0> (lrf_bcdw_kernel.hip) operation for operation, with the
// V table, the b table, the old int8 rows and the partial tables reached through sc1 accesses.  Xs: the wave's LDS share
// (X tile 16 KB, then the fp32 u tile 2 KB).
// MODE 1 (round 5, k_bcd_p<.., .., true>): the call's FIRST iteration, k_bcd_w<1> — the old U is X @ W0 (the initialisation's
// table Wf, written before the launch), no int8 rows are read.
template <int MODE>
__device__ __forceinline__ void bcdp_w_block(const float* __restrict__ X, const PlaneDesc& pd, const BlockDesc& bd, const float* __restrict__ Vf,
                                             const float* __restrict__ Wf, const float* __restrict__ Bf, int8_t* __restrict__ U, float* __restrict__ Ppart,
                                             float* __restrict__ Qpart, const GsParams& gp, float* Xs, const int lane)
{
    constexpr int RMAX = 8;
    const int li = lane & 15, lq = lane >> 4;
    float* us = Xs + 64 * 64;
    const float* xp[4];
#pragma unroll
    for (int e = 0; e < 4; e++) xp[e] = &Xs[lq * 64 + 4 * (li ^ (4 * e + lq))];
    const float* ub = &us[lq * RMAX + (li & 7)];
    const float* uq = &us[(lq + 4 * (li >> 3)) * RMAX + (li & 7)];
    const float* xrow = &Xs[lane * 64];
    const int g16 = 16 * xsw(lane);
    const int R = pd.R;
    const float* Xp = X + pd.x_off + (long)bd.row0 * 64;
    const float* Vp = Vf + (long)bd.plane * 64 * LRF_RP;
    const float* gt = Bf + (long)bd.plane * LRF_GT_STRIDE;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;
    const bool native = pd.native_t2_u != 0;

    float vreg[8][4], wreg[MODE == 1 ? 8 : 1][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            vreg[r][kb] = ld_sc1(Vp + (16 * kb + li) * LRF_RP + r);
            if constexpr (MODE == 1) wreg[r][kb] = Wf[((long)bd.plane * 64 + 16 * kb + li) * LRF_RP + r];
        }
    float tab[5];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ci = 16 * j + li, tr = ci >> 3, tn = ci & 7;
        tab[j] = ld_sc1(gt + tr * LRF_GT_LD + (tn < 7 ? tn : LRF_GT_RDEN));
    }
    tab[4] = ld_sc1(gt + (li & 7) * LRF_GT_LD + LRF_GT_DEN);

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
    // The old int8 row with compiler-tracked sc1 loads: three ALIGNED dwords that cover the row's R <= 8 bytes (a hand-issued
    // asm load would leave its result register open to compiler copies before the data has landed); row_bytes() shifts
    // them into place.  The last dword is clamped to the one that holds the row's last byte (never past the allocation).
    unsigned uraw[3];
    int ush = 0;
    auto issue_u = [&](int t) {
        int row = t * 64 + lane;
        row = row < nrows ? row : nrows - 1;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(Ub + (long)row * R);
        const uintptr_t base = a0 & ~(uintptr_t)3, last = (a0 + R - 1) & ~(uintptr_t)3;
        ush = (int)(a0 & 3);
        const unsigned* p0 = reinterpret_cast<const unsigned*>(base);
        const unsigned* p1 = reinterpret_cast<const unsigned*>(base + 4 <= last ? base + 4 : last);
        const unsigned* p2 = reinterpret_cast<const unsigned*>(base + 8 <= last ? base + 8 : last);
#ifdef LRF_BCDP_PLAIN_U
        uraw[0] = *p0; uraw[1] = *p1; uraw[2] = *p2;
#else
        uraw[0] = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uraw[1] = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uraw[2] = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    };
    auto row_bytes = [&](unsigned& lo, unsigned& hi) { // bytes 0..3 and 4..7 of the row (bytes at or past R: unspecified)
        lo = __builtin_amdgcn_alignbyte(uraw[1], uraw[0], (unsigned)ush);
        hi = __builtin_amdgcn_alignbyte(uraw[2], uraw[1], (unsigned)ush);
    };

    f32x4 accP[4], accQ = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) accP[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (MODE == 0) issue_u(0);
    issue_x(0, 0, 4, true);
    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int m = 16 * T + 4 * q + lq;
                *reinterpret_cast<f32x4*>(&Xs[m * 64 + 4 * (li ^ (4 * q + lq))]) = xq[T][q];
            }
        float u[RMAX];
        const int row = r0 + lane;
        if constexpr (MODE == 0) {
            unsigned lo, hi;
            row_bytes(lo, hi);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                u[r] = (float)(int)(int8_t)(lo >> (8 * r));
                u[4 + r] = (float)(int)(int8_t)(hi >> (8 * r));
            }
        }
        const int tn = t + 1;
        const bool more = tn < nsub;
        if constexpr (MODE == 0) {
            if (more) issue_u(tn);
        }
        issue_x(tn, 0, 2, more);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        float a[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; r++) a[r] = 0.f;
        row_times_v_dispatch(R, xrow, g16, vreg, a);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = 0.f;
            row_times_v_dispatch(R, xrow, g16, wreg, u);
        }
        issue_x(tn, 2, 3, more);
        __builtin_amdgcn_sched_barrier(0);
        gs_regs_dispatch<RMAX>(R, a, u, tab, native, gp);
        if (row >= nrows) {
#pragma unroll
            for (int r = 0; r < RMAX; r++) u[r] = 0.f;
        }
        issue_x(tn, 3, 4, more);
        __builtin_amdgcn_sched_barrier(0);
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
            if (R >= 4) {
                st_sc1_u32_unaligned(uo, lo);
                const unsigned long long w = ((unsigned long long)hi << 32) | lo;
                st_sc1_u32_unaligned(uo + R - 4, (unsigned)(w >> (8 * (R - 4))));
            } else {
                __hip_atomic_store(uo, (int8_t)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (R > 1) __hip_atomic_store(uo + 1, (int8_t)(lo >> 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (R > 2) __hip_atomic_store(uo + 2, (int8_t)(lo >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        float pu[16], qu[8];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            float v = ub[4 * s * RMAX];
            pu[s] = (li < RMAX) ? v : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 8; h++) qu[h] = uq[8 * h * RMAX];
        f32x4 px[16];
#pragma unroll
        for (int s = 0; s < 16; s++) px[s] = *reinterpret_cast<const f32x4*>(xp[s & 3] + 256 * s);
#pragma unroll
        for (int s = 0; s < 16; s++) {
#pragma unroll
            for (int c = 0; c < 4; c++) accP[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(px[s][c], pu[s], accP[c], 0, 0, 0);
            if (s & 1) accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[s >> 1], qu[s >> 1], accQ, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const long slot = (long)pd.blk0 + bd.blk;
    float* Pp = Ppart + slot * 64 * LRF_RP;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) st_sc1(Pp + (4 * (4 * lq + reg) + c) * LRF_RP + li, accP[c][reg]);
    float* Qp = Qpart + slot * LRF_RP * LRF_RP;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int i = 4 * lq + reg;
        const float mine = accQ[reg];
        const float other = __shfl(mine, ((lq + 2) & 3) * 16 + ((li + 8) & 15), 64);
        st_sc1(Qp + i * LRF_RP + li, (i < 8 && li < 8) ? mine + other : 0.f);
    }
}

// F16: planes of ranks 9..16 may occur (w16_block); NP32 > 0: planes of ranks 2 NP32 - 1 / 2 NP32 (17..32) may occur
// (w32_block<NP32>).  t16 / t64: the table sets of the two rank pitches (a call without ranks above 16 has only t16).
// wave_lds: bytes of LDS per wave (the largest share a family of the call needs; the V updates fit the smallest).
// FIRST (round 5; only without ranks above 16): item iteration 0 is the call's FIRST iteration — the bodies of k_bcd_w<1> /
// k_bcd_w16<1>, old U = X @ W0 from the initialisation's tables, which are complete before the launch: no flag to wait for —,
// so that the launch covers all K iterations and the call has no U-update or V-update launch of its own.
template <bool F16, int NP32, bool FIRST = false>
__global__ __launch_bounds__(64 * LRF_BCDW_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bcd_p(
    const float* __restrict__ X, const PlaneDesc* __restrict__ planes, const BlockDesc* __restrict__ blocks, const BcdpTabs t16,
    const BcdpTabs t64, int8_t* __restrict__ U, int8_t* __restrict__ V8, GsParams gp, int nblocks, int niter, int nplanes, int plane0,
    BcdpSync* sync, int* err_host, int ncells, int seq, int wave_lds)
{
    extern __shared__ __attribute__((aligned(16))) float bcdp_lds[]; // LRF_BCDW_WAVES x wave_lds bytes
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* Xs = reinterpret_cast<float*>(reinterpret_cast<char*>(bcdp_lds) + wave * wave_lds);
    int* ticket = sync->cell;
    int* flag = sync->cell + nplanes;
    const int total = niter * nblocks;
    // leaving: the wave's own stores to the flags have landed before it is counted; the last one out zeroes the state
    auto leave = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int l = 0;
        if (lane == 0) l = __hip_atomic_fetch_add(&sync->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int d = __builtin_amdgcn_readfirstlane(__shfl(l, 0, 64));
        if (d == (int)(gridDim.x * LRF_BCDW_WAVES) - 1) {
            for (int i = lane; i < ncells; i += 64) __hip_atomic_store(&sync->cell[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync->head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // a launch queued behind a failed one: the queue state is dirty (head past the end, tickets of the failed launch): nothing
    // here may run on it.  The host word already names the first failed launch; this one is invalid by its sequence number.
    if (__hip_atomic_load(&sync->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    for (;;) {
        // lane 0 pulls, EVERY lane then holds lane 0's value (an explicit lane-0 broadcast: with `readfirstlane` of a variable
        // that is 0 in the other lanes the compiler's control flow let lanes 1..63 go on with item 0 after lane 0 had left)
        int idx_l = 0;
        if (lane == 0) idx_l = __hip_atomic_fetch_add(&sync->head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int idx = __builtin_amdgcn_readfirstlane(__shfl(idx_l, 0, 64));
        if (idx >= total) {
            leave();
            return;
        }
        // (iteration-major, the blocks of an iteration in table order.  Tried and not kept, round 5: the rank families of a call
        // interleaved within an iteration — luma blocks beside chroma blocks on a SIMD: within noise —, and chunks of ~86 images
        // (2064 blocks, 200 MB of X) running all their iterations before the next chunk, so that X stays in the 256 MB Infinity
        // Cache: 139 -> 164 us per iteration at (7,3,3), 246 -> 353 at (26,13,13): a chunk is barely one round of the resident
        // waves, so blocks poll for V tables that are still being computed — and the kernel is not bound by HBM bandwidth)
        const int it = idx / nblocks, blk = idx - it * nblocks;
#ifdef LRF_BCDP_DEBUG
        atomicAdd(&sync->cell[2 * nplanes + blk], 1 + 1000 * it); // every active lane
#endif
        const BlockDesc bd = blocks[blk];
        const PlaneDesc pd = planes[bd.plane];
        const int pl = bd.plane - plane0; // index into the tickets / flags
        if (it > 0) { // the V update of (matrix, it - 1) must have been published
            int polls = 0;
            while (__hip_atomic_load(&flag[pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it) {
                __builtin_amdgcn_s_sleep(8);
                if (++polls > LRF_BCDP_MAX_POLLS) { // (the state stays dirty: the host sees the error word and zeroes it)
                    __hip_atomic_store(&sync->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(err_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
            }
        }
        // nothing below may move above the poll: the hardware path is the sc1 loads behind the flag (MI355X_MICROARCH.md,
        // inter-workgroup visibility, first row of the sc1 table); this pins the compiler to the same order (relaxed atomics
        // to different addresses are not ordered among themselves by the memory model)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        asm volatile("" ::: "memory");
#ifdef LRF_BCDP_FENCES
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        // the block, by the rank family of its plane (wave-uniform).  `ln` is the lane number behind an opaque move: left
        // visible as loop-invariant, the per-lane LDS addresses of EVERY family's body are hoisted out of the item loop and
        // stay live across the other families' bodies (15-25 registers: <true, 0> spilled 89 registers, the V operand of
        // w16_block's MFMAs among them, reloaded at every use)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int fam = (!F16 || pd.R <= 8) ? 0 : ((NP32 == 0 || pd.R <= 16) ? 1 : 2);
        static_assert(!FIRST || NP32 == 0, "the first iteration of ranks 17..32 stays a launch of its own (k_bcd_w32f)");
        const bool first = FIRST && it == 0; // wave-uniform
        if (fam == 0) {
            if (first) {
                if constexpr (FIRST) bcdp_w_block<1>(X, pd, bd, t16.vf, t16.wf, t16.bf, U, t16.pp, t16.qp, gp, Xs, ln);
            } else
                bcdp_w_block<0>(X, pd, bd, t16.vf, nullptr, t16.bf, U, t16.pp, t16.qp, gp, Xs, ln);
        }
        if constexpr (F16) {
            if (fam == 1) {
                if (first) {
                    if constexpr (FIRST) w16_block<1, MemSc1>(X, pd, bd, t16.vf, t16.wf, t16.bf, U, t16.pp, t16.qp, gp, Xs, ln);
                } else
                    w16_block<0, MemSc1>(X, pd, bd, t16.vf, nullptr, t16.bf, U, t16.pp, t16.qp, gp, Xs, ln);
            }
        }
        if constexpr (NP32 > 0) {
            if (fam == 2) w32_block<NP32, MemSc1>(X, pd, bd, t64.vf, t64.bf, U, t64.pp, t64.qp, gp, Xs, ln, 0);
        }
        // every store of this wave has left (write-through) before it is counted
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LRF_BCDP_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        int arrived_l = 0;
        if (lane == 0) arrived_l = __hip_atomic_fetch_add(&ticket[pl], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int arrived = __builtin_amdgcn_readfirstlane(__shfl(arrived_l, 0, 64));
        __atomic_signal_fence(__ATOMIC_SEQ_CST); // the partial loads of the V update stay behind the returned ticket
        asm volatile("" ::: "memory");
        if (arrived == (it + 1) * pd.nblk - 1) { // the last block of (matrix, iteration): its V update
#ifdef LRF_BCDP_FENCES
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            const int last = it == niter - 1 ? 1 : 0;
            int lv = lane;
            asm volatile("" : "+v"(lv)); // (as `ln` above)
            if (fam == 0) bcdp_vupdate(pd, bd.plane, t16.pp, t16.qp, t16.vf, t16.bf, V8, gp, last, Xs, lv);
            if constexpr (F16) {
                if (fam == 1) bcdp_vupdate16(pd, bd.plane, t16.pp, t16.qp, t16.vf, t16.bf, V8, gp, last, Xs, lv);
            }
            if constexpr (NP32 > 0) {
                if (fam == 2) bcdp_vupdate32(pd, bd.plane, t64.pp, t64.qp, t64.vf, t64.bf, V8, gp, last, Xs, lv);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LRF_BCDP_FENCES
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            // every lane stores the same word (no `if (lane == 0)` here: followed by the loop head's `if (lane == 0)` pull it let
            // the compiler thread lane 0 through both and retire it from the loop alone — lanes 1..63 then went on with item 0)
#ifdef LRF_BCDP_TEST_SKIP_FLAG // tests/test_persist_error.py, tools/dev_persist_expiry.py: in a call of FOUR iterations matrix 0 never
                               // publishes its first V update, its later blocks' polls expire; other calls of that build work
            if (!(pl == 0 && it == (FIRST ? 1 : 0) && niter == (FIRST ? 4 : 3)))
#endif
            __hip_atomic_store(&flag[pl], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
