#pragma once
// lrf_anyshape_kernels.hip — QMF for matrices of any shape [M, N] and any rank: the branches of qmf_encode that do not
// produce 64-column patch matrices (lrf/compression/qmf.py:227-286 with patch_size = (4,4), (16,16), (32,32) — the sweep of
// experiments/ablation_patchsize/eval.py:49-55 — and patch=False, where X is the whole plane [H, W]).
//
// Same arithmetic as the 64-column kernels, restated shape-free (oracle/lrf_oracle.c lrf_oracle_bcd reproduces the
// reference bit for bit for every shape and rank tried, tools/gen_golden.py `anyshape`):
//   * every matrix product (x @ v, v.mT @ v, x.mT @ u, u.mT @ u; lrf/factorization/qmf.py:107) is one k-ordered fma chain
//     per output element over blocks of LRF_KC = 384 of the contraction, the block sums added in block order
//     (k_any_prod writes the block partials, k_any_fold adds them); ATen's native kernel (rounded product, then sum) when
//     contraction * rows * cols < 400;
//   * the Gauss-Seidel sweep over the columns of a factor (qmf.py:108-119) with `uu @ bb` in MKL's single-column order
//     (k_any_gs), round-half-even and clamp (qmf.py:191-195).
// The SVD initialisation (qmf.py:42-71) is this project's own: fp64 Gram matrix of the SHORT side (n = min(M, N)),
// Householder tridiagonalisation, multisection eigenvalues, twisted-factorisation vectors, Gram-Schmidt, back-transformation
// (k_any_gram, k_any_eig), then the long factor as X (or X^T) times e / sqrt(sigma).
// Correctness-first, not tuned.  Included by lrf_api.hip after lrf_svd_kernels.hip (wave_sum, block_sum).

// ---- batched products with the reference's summation order ------------------------------------------------------
// P[b][blk][i][r] = chain over k in block blk of A(i,k) * Bm[k][r];  A(i,k) = A[b*a_batch + i*sai + k*sak].
// grid (ceil(I/64), ceil(R/32) * nblk, B).  MKL order: f32 MFMA tiles (a k-ordered fma chain per element); ATen's native
// order (tiny products): scalar loop, each thread 8 rows x 1 column.
__global__ __launch_bounds__(256) void k_any_prod(const float* __restrict__ A, long a_batch, long sai, long sak,
                                                  const float* __restrict__ Bm, long b_batch, float* __restrict__ P,
                                                  int I, int D, int R, int nblk, int native, int tiles /* 64-row tiles per workgroup */)
{
    __shared__ float As[64 * 65];
    __shared__ float Bs[64 * 32];
    const int tid = threadIdx.x, tr = tid & 31, ti = tid >> 5;
    const int lane = tid & 63, li = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = blockIdx.y / nblk, blk = blockIdx.y - rt * nblk;
    const int r0 = rt * 32;
    const float* Ab = A + (long)blockIdx.z * a_batch;
    const float* Bb = Bm + (long)blockIdx.z * b_batch;
    const int kbeg = blk * LRF_KC;
    const int kend = (kbeg + LRF_KC < D) ? kbeg + LRF_KC : D;
    // a workgroup takes `tiles` consecutive 64-row tiles (tall matrices with a short contraction — [24576,16] for 4x4 patches —
    // are otherwise hundreds of thousands of workgroups with a few hundred multiply-adds each)
    for (int tile = 0; tile < tiles; tile++) {
    const int i0 = (blockIdx.x * tiles + tile) * 64;
    if (i0 >= I) break; // workgroup-uniform
    f32x4 macc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    // staging width along k: the next power of two >= the contraction (64 at most), so that a short contraction (N = 16 for
    // 4x4 patches) does not walk 64-wide tiles of mostly absent entries
    int kw = 64, kws = 6;
    while (kw > 4 && (kw >> 1) >= D) { kw >>= 1; kws--; } // at least 4: one MFMA step reads four k
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        const int klen = (kend - k0 < 64) ? kend - k0 : 64;
        __syncthreads();
        if (sak == 1) { // rows of A contiguous along k
            for (int e = tid; e < 64 * kw; e += 256) {
                const int kk = e & (kw - 1), ii = e >> kws;
                float v = 0.f;
                if (kk < klen && i0 + ii < I) v = Ab[(long)(i0 + ii) * sai + (k0 + kk)];
                As[kk * 65 + ii] = v;
            }
        } else { // contiguous along i (transposed views)
            for (int e = tid; e < 64 * kw; e += 256) {
                const int ii = e & 63, kk = e >> 6;
                float v = 0.f;
                if (kk < klen && i0 + ii < I) v = Ab[(long)(i0 + ii) * sai + (long)(k0 + kk) * sak];
                As[kk * 65 + ii] = v;
            }
        }
        for (int e = tid; e < kw * 32; e += 256) {
            const int rr = e & 31, kk = e >> 5;
            float v = 0.f;
            if (kk < klen && r0 + rr < R) v = Bb[(long)(k0 + kk) * R + r0 + rr];
            Bs[e] = v;
        }
        __syncthreads();
        if (native) {
            for (int kk = 0; kk < klen; kk++) {
                const float b = Bs[kk * 32 + tr];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float p = As[kk * 65 + ti + 8 * j] * b;
                    acc[j] = acc[j] + p;
                }
            }
        } else {
            // wave w: rows 16w .. 16w+15 of the tile, both 16-wide column tiles.  v_mfma_f32_16x16x4_f32 adds its four
            // products to the accumulator one after the other in k order — the same chain of fmas as the scalar loop — and
            // entries past klen are zeros staged above (fma(0, 0, acc) = acc).
            const int steps = (klen + 3) >> 2;
            if (r0 + 16 < R) {
                for (int s4 = 0; s4 < steps; s4++) {
                    const float av = As[(4 * s4 + lq) * 65 + 16 * wave + li];
                    macc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[(4 * s4 + lq) * 32 + li], macc[0], 0, 0, 0);
                    macc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[(4 * s4 + lq) * 32 + 16 + li], macc[1], 0, 0, 0);
                }
            } else { // ranks up to 16 (in this rank tile): the second column tile is empty
                for (int s4 = 0; s4 < steps; s4++) {
                    const float av = As[(4 * s4 + lq) * 65 + 16 * wave + li];
                    macc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[(4 * s4 + lq) * 32 + li], macc[0], 0, 0, 0);
                }
            }
        }
    }
    float* Pb = P + (((long)blockIdx.z * nblk + blk) * I) * R;
    if (native) {
        if (r0 + tr < R) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + ti + 8 * j;
                if (i < I) Pb[(long)i * R + r0 + tr] = acc[j];
            }
        }
    } else { // C/D layout of the 16x16 tile: column = lane & 15, rows 4 (lane >> 4) + reg
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            const int r = r0 + 16 * ct + li;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = i0 + 16 * wave + 4 * lq + reg;
                if (i < I && r < R) Pb[(long)i * R + r] = macc[ct][reg];
            }
        }
    }
    }
}

// The same products for contractions deeper than 32 (round 3): a 128 x 64 tile of P per workgroup, a wave 32 rows x 64
// columns as 2 x 4 MFMA tiles (six LDS operand reads per eight MFMAs; k_any_prod: two or three per one or two), the next
// 32-deep chunk fetched into registers under the MFMAs of the current one, loads unconditional on clamped addresses (a
// guarded load is an exec-mask branch).  Per element still the k-ordered fma chain of its 384-block: bit-identical.
// grid (ceil(I/128), ceil(R/64) * nblk, B)
__global__ __launch_bounds__(256) void k_any_prod_big(const float* __restrict__ A, long a_batch, long sai, long sak,
                                                      const float* __restrict__ Bm, long b_batch, float* __restrict__ P,
                                                      int I, int D, int R, int nblk)
{
    constexpr int LSA = 145, LSB = 81; // strides 17 mod 64: operand reads (16 lanes along i / r, 4 along k) and staging stores conflict-free
    __shared__ float As[32 * LSA];
    __shared__ float Bs[32 * LSB];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = blockIdx.y / nblk, blk = blockIdx.y - rt * nblk;
    const int r0 = rt * 64, i0 = blockIdx.x * 128;
    const float* Ab = A + (long)blockIdx.z * a_batch;
    const float* Bb = Bm + (long)blockIdx.z * b_batch;
    const int kbeg = blk * LRF_KC;
    const int kend = (kbeg + LRF_KC < D) ? kbeg + LRF_KC : D;
    f32x4 acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging roles: A 32 x 128 = 16 values per thread, B 32 x 64 = 8 per thread.  Element offsets relative to the chunk's first
    // k are fixed per thread (rows / columns clamped once), so a chunk costs one scalar base and 24 loads; only the last,
    // partial chunk of a block guards k.  (The host sends matrices whose offsets do not fit 32 bits to k_any_prod.)
    float ra[16], rb[8];
    unsigned offa[16], offb[8];
    int kka[16], kkb[8];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int e = tid + 256 * t;
        int kk, ii;
        if (sak == 1) { kk = e & 31; ii = e >> 5; } else { ii = e & 127; kk = e >> 7; }
        const int i = (i0 + ii < I) ? i0 + ii : I - 1;
        kka[t] = kk;
        offa[t] = (unsigned)((long)i * sai + (long)kk * sak);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int e = tid + 256 * t;
        const int rr = e & 63, kk = e >> 6;
        kkb[t] = kk;
        offb[t] = (unsigned)(kk * R + ((r0 + rr < R) ? r0 + rr : R - 1));
    }
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        const float* pa = Ab + (long)k0 * sak; // wave-uniform bases
        const float* pb = Bb + (long)k0 * R;
        if (k0 + 32 <= kend) {
#pragma unroll
            for (int t = 0; t < 16; t++) ra[t] = pa[offa[t]];
#pragma unroll
            for (int t = 0; t < 8; t++) rb[t] = pb[offb[t]];
        } else { // the block's last chunk: k past kend reads as zero (and is not dereferenced)
            const int klen = kend - k0;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const float v = pa[(kka[t] < klen) ? offa[t] : offa[t] - (unsigned)((long)(kka[t] - (klen - 1)) * sak)];
                ra[t] = (kka[t] < klen) ? v : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const float v = pb[(kkb[t] < klen) ? offb[t] : offb[t] - (unsigned)((kkb[t] - (klen - 1)) * R)];
                rb[t] = (kkb[t] < klen) ? v : 0.f;
            }
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        const int klen = (kend - k0 < 32) ? kend - k0 : 32;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int e = tid + 256 * t;
            int kk, ii;
            if (sak == 1) { kk = e & 31; ii = e >> 5; } else { ii = e & 127; kk = e >> 7; }
            As[kk * LSA + ii] = ra[t];
        }
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int e = tid + 256 * t;
            Bs[(e >> 6) * LSB + (e & 63)] = rb[t];
        }
        __syncthreads();
        fetch(k0 + 32 < kend ? k0 + 32 : k0); // unconditional (the last one re-reads its own chunk): exact wait counts
        const int steps = (klen + 3) >> 2; // entries past klen are zeros: fma(0, 0, acc) = acc
        for (int s4 = 0; s4 < steps; s4++) {
            const float* ar = As + (4 * s4 + lq) * LSA + 32 * wave + li;
            const float* br = Bs + (4 * s4 + lq) * LSB + li;
            const float a0 = ar[0], a1 = ar[16];
            const float b0 = br[0], b1 = br[16], b2 = br[32], b3 = br[48];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b2, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b3, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b2, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b3, acc[1][3], 0, 0, 0);
        }
    }
    float* Pb = P + (((long)blockIdx.z * nblk + blk) * I) * R;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = r0 + 16 * q + li;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = i0 + 32 * wave + 16 * p + 4 * lq + reg;
                if (i < I && r < R) Pb[(long)i * R + r] = acc[p][q][reg];
            }
        }
}

// Thin products (4 x 4 patches: X [24576, 16], rank 3 — x.mT @ u and u.mT @ u contract 24576 rows into a 16 x 3 or 3 x 3
// result, x @ v contracts 16 columns for 24576 rows): k_any_prod stages 64 x 64 tiles of which a sixteenth is used.  Here the
// MFMA operands come straight from global memory, one wave per work item, eight steps' loads in flight; per element the same
// k-ordered chain.
// (1) I <= 16, R <= 16, long contraction: one wave per (384-block, matrix).  grid (nblk, B), 64 threads
__global__ __launch_bounds__(64) void k_any_prod_thin_long(const float* __restrict__ A, long a_batch, long sai, long sak,
                                                           const float* __restrict__ Bm, long b_batch, float* __restrict__ P,
                                                           int I, int D, int R, int nblk)
{
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
    const int blk = blockIdx.x;
    const int kbeg = blk * LRF_KC, kend = (kbeg + LRF_KC < D) ? kbeg + LRF_KC : D;
    const float* pa = A + (long)blockIdx.y * a_batch + (long)((li < I) ? li : I - 1) * sai;
    const float* pb = Bm + (long)blockIdx.y * b_batch + ((li < R) ? li : R - 1);
    const bool va = li < I, vb = li < R;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int k = k0 + 4 * q + lq, kc = (k < kend) ? k : kend - 1;
            const float x = pa[(long)kc * sak], y = pb[(long)kc * R];
            av[q] = (va && k < kend) ? x : 0.f;
            bv[q] = (vb && k < kend) ? y : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (k0 + 4 * q < kend) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0); // wave-uniform
    }
    float* Pb = P + (((long)blockIdx.y * nblk + blk) * I) * R;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int i = 4 * lq + reg;
        if (i < I && li < R) Pb[(long)i * R + li] = acc[reg];
    }
}

// (2) contraction D <= 64, R <= 16, many rows: a wave takes TPW consecutive 16-row tiles; the B operand (D x R) stays in
// registers.  Single 384-block (D <= 64 < LRF_KC): P is the result.  grid (ceil(I / (16 TPW)), B), 64 threads
template <int STEPS>
__global__ __launch_bounds__(64) void k_any_prod_thin_short(const float* __restrict__ A, long a_batch, long sai, long sak,
                                                            const float* __restrict__ Bm, long b_batch, float* __restrict__ P,
                                                            int I, int D, int R, int tpw)
{
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
    const float* Ab = A + (long)blockIdx.y * a_batch;
    const float* pb = Bm + (long)blockIdx.y * b_batch + ((li < R) ? li : R - 1);
    float bv[STEPS];
#pragma unroll
    for (int q = 0; q < STEPS; q++) {
        const int k = 4 * q + lq, kc = (k < D) ? k : D - 1;
        const float y = pb[(long)kc * R];
        bv[q] = (li < R && k < D) ? y : 0.f;
    }
    float* Pb = P + (long)blockIdx.y * I * R;
    for (int t = 0; t < tpw; t++) {
        const int i0 = (blockIdx.x * tpw + t) * 16;
        if (i0 >= I) break; // wave-uniform
        const int ic = (i0 + li < I) ? i0 + li : I - 1;
        const float* pa = Ab + (long)ic * sai;
        float av[STEPS];
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const int k = 4 * q + lq, kc = (k < D) ? k : D - 1;
            const float x = pa[(long)kc * sak];
            av[q] = (k < D) ? x : 0.f;
        }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < STEPS; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int i = i0 + 4 * lq + reg;
            if (i < I && li < R) Pb[(long)i * R + li] = acc[reg];
        }
    }
}

// C[b][e] = ((P[b][0][e] + P[b][1][e]) + P[b][2][e]) + ...   e < IR;  grid (ceil(IR/256), B)
__global__ __launch_bounds__(256) void k_any_fold(const float* __restrict__ P, float* __restrict__ C, long IR, int nblk)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= IR) return;
    const float* Pp = P + (long)blockIdx.y * nblk * IR + e;
    float acc = Pp[0];
    for (int b = 1; b < nblk; b++) acc = acc + Pp[(long)b * IR];
    C[(long)blockIdx.y * IR + e] = acc;
}

// `uu @ bb` of qmf.py:115 for column r of one row: the K = R - 1 terms j != r, in MKL's single-output-column order
// (oracle dot_mkl_n1) or ATen's native order.  u: the row (LDS, pitch 1), b: row r of the symmetric b (uniform address).
// A chain of the sum in j space: cnt terms u[j] b[j] from j on in steps of STEP, added to acc in that order.  The operands of
// eight terms are fetched together (they do not depend on the sum): taken one at a time, every term waited for its LDS read
// and its uniform load, and the index n -> j = n + (n >= r) cost a dozen scalar instructions per term — the skipped column
// splits each chain into two plain runs instead.
template <int STEP, typename UT>
__device__ __forceinline__ float any_chain(float acc, const UT* u, const float* __restrict__ b, int j, int cnt)
{
    int t = 0;
    for (; t + 8 <= cnt; t += 8) {
        float pq[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pq[q] = (float)u[j + STEP * q] * b[j + STEP * q];
#pragma unroll
        for (int q = 0; q < 8; q++) acc = acc + pq[q];
        j += 8 * STEP;
    }
    for (; t < cnt; t++) {
        const float pv = (float)u[j] * b[j];
        acc = acc + pv;
        j += STEP;
    }
    return acc;
}

template <typename UT>
__device__ __forceinline__ float any_term2(const UT* u, const float* __restrict__ b, int r, int R, bool native)
{
    const int K = R - 1;
    if (K <= 0) return 0.f;
#define ANY_J(n) ((n) < r ? (n) : (n) + 1)
    if (native) { // one chain, n ascending: j = 0 .. r - 1, then r + 1 .. R - 1
        float acc = any_chain<1, UT>(0.f, u, b, 0, r);
        return any_chain<1, UT>(acc, u, b, r + 1, K - r);
    }
    const int j0 = ANY_J(0);
    if (K == 1) return (float)u[j0] * b[j0];
    const int j1 = ANY_J(1);
    float odd = fmaf((float)u[j1], b[j1], (float)u[j0] * b[j0]);
    if (K < 3) return odd;
    const int j2 = ANY_J(2);
    float even = (float)u[j2] * b[j2];
    // odd n from last_odd down to 3: first the part with n >= r (j = n + 1), then n < r (j = n)
    const int last_odd = ((K - 1) & 1) ? K - 1 : K - 2;
    {
        const int m = r > 3 ? r : 3, n0 = (m & 1) ? m : m + 1;             // smallest odd n >= max(r, 3)
        const int chi = (last_odd >= n0) ? (last_odd - n0) / 2 + 1 : 0;
        odd = any_chain<-2, UT>(odd, u, b, last_odd + 1, chi);
        int n1 = (r & 1) ? r - 2 : r - 1;                                  // largest odd n < r
        if (n1 > last_odd) n1 = last_odd;
        const int clo = (n1 >= 3) ? (n1 - 3) / 2 + 1 : 0;
        odd = any_chain<-2, UT>(odd, u, b, n1, clo);
    }
    // even n from 4 up to K - 1: first n < r (j = n), then n >= r (j = n + 1)
    {
        const int lim = r < K ? r : K;                                     // n < lim
        const int n2 = ((lim - 1) & 1) ? lim - 2 : lim - 1;                // largest even n < lim
        const int clo = (n2 >= 4) ? (n2 - 4) / 2 + 1 : 0;
        even = any_chain<2, UT>(even, u, b, 4, clo);
        const int m = r > 4 ? r : 4, n3 = (m & 1) ? m + 1 : m;             // smallest even n >= max(r, 4)
        const int chi = (n3 < K) ? (K - 1 - n3) / 2 + 1 : 0;
        even = any_chain<2, UT>(even, u, b, n3 + 1, chi);
    }
    return odd + even;
#undef ANY_J
}

// Gauss-Seidel over the R columns of 64 rows (lane = row); the rows sit in LDS with an odd pitch.
// UT = float, or int8_t when the factor on entry holds integers of the int8 range (every sweep after the first one of a
// bounded factorisation): a quarter of the LDS per wave, i.e. four times the waves per CU for a kernel that is a chain of
// dependent additions per lane (at rank 102: 26 KB per wave, six waves per CU).
// a [B][I][R], bm [B][R][R] (symmetric), F [B][I][R] updated in place.  grid (ceil(I/64), B), 64 threads,
// dynamic LDS 64 * (R | 1) floats.
// l1, l2: the elastic-net terms of CoordinateDescent.update_u (qmf.py:116-118; 0 for qmf_encode): numerator
// soft_thresholding(term1 - term2, l1) (factorization/utils.py:36-40), denominator b_rr + l2.
__device__ __forceinline__ float any_soft_threshold(float x, float thr)
{
    if (thr == 0.f) return x;
    const float ax = fabsf(x) - thr;
    const float sg = (x > 0.f) ? 1.f : (x < 0.f ? -1.f : 0.f);
    return sg * (ax > 0.f ? ax : 0.f);
}

template <typename UT, bool BLDS>
__global__ __launch_bounds__(64) void k_any_gs(const float* __restrict__ a, const float* __restrict__ bm, float* __restrict__ F,
                                               int I, int R, int native, float lo, float hi, float l1, float l2, float eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    UT* us = reinterpret_cast<UT*>(smem);
    // row pitch in elements: an odd number of dwords, so that the 64 lanes' reads of one column fall into 64 banks
    const int RP = (sizeof(UT) == 4) ? (R | 1) : 4 * (((R + 3) >> 2) | 1), lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    const int nrows = (I - row0 < 64) ? I - row0 : 64;
    float* Fb = F + ((long)blockIdx.y * I + row0) * R;
    const float* ab = a + ((long)blockIdx.y * I + row0) * R;
    const float* bb = bm + (long)blockIdx.y * R * R;
    // the block's rows are one contiguous run of nrows * R floats: walk it with a running (row, column) pair instead of a
    // division per element
    {
        int row = lane / R, r = lane - row * R;
        const int drow = 64 / R, dr = 64 - drow * R;
        for (int e = lane; e < nrows * R; e += 64) {
            us[row * RP + r] = (UT)Fb[e];
            row += drow;
            r += dr;
            if (r >= R) { r -= R; row++; }
        }
    }
    // Row r of b goes through LDS (bs), fetched from global memory a column ahead: as uniform (scalar) loads every pair of
    // terms waited for its own round trip — scalar loads return out of order, so each wait is for all of them (~75 cycles per
    // term at rank 102); LDS reads return in order and sixteen terms' operands are in flight together.
    float* bs = reinterpret_cast<float*>(smem + (((size_t)64 * RP * sizeof(UT) + 15) & ~(size_t)15));
    constexpr int NBR = (LRF_ANY_MAX_RANK + 63) / 64;
    float bn[NBR];
    auto fetch_b = [&](int r) __attribute__((always_inline)) {
        const float* brow = bb + (long)r * R;
#pragma unroll
        for (int t = 0; t < NBR; t++)
            if (64 * t < R) bn[t] = brow[(lane + 64 * t < R) ? lane + 64 * t : R - 1]; // wave-uniform guard
    };
    if (BLDS) fetch_b(0);
    UT* u = us + ((lane < nrows) ? lane : 0) * RP;
    const float* al = ab + (long)((lane < nrows) ? lane : 0) * R;
    float a_next = al[0];
    for (int r = 0; r < R; r++) {
        const float* brow = bs;
        if (BLDS) {
            __syncthreads(); // the reads of the previous column's row are over
#pragma unroll
            for (int t = 0; t < NBR; t++)
                if (64 * t < R && lane + 64 * t < R) bs[lane + 64 * t] = bn[t];
            __syncthreads();
            fetch_b(r + 1 < R ? r + 1 : r); // unconditional: a guarded prefetch makes the wait counts conservative
        } else { // ranks whose float rows fill the LDS (R > 628): uniform loads from global memory
            brow = bb + (long)r * R;
        }
        const float a_cur = a_next;
        a_next = al[r + 1 < R ? r + 1 : r]; // requested a column ahead (unconditionally): the sweep does not wait for it
        if (lane < nrows) {
            const float term2 = any_term2(u, brow, r, R, native != 0);
            const float num = any_soft_threshold(a_cur - term2, l1) + eps; // CoordinateDescent's eps (qmf.py:90, 117-118)
            const float den = (brow[r] + l2) + eps;
            const float val = rintf(num / den);
            u[r] = (UT)fminf(fmaxf(val, lo), hi);
        }
    }
    __syncthreads();
    {
        int row = lane / R, r = lane - row * R;
        const int drow = 64 / R, dr = 64 - drow * R;
        for (int e = lane; e < nrows * R; e += 64) {
            Fb[e] = (float)us[row * RP + r];
            row += drow;
            r += dr;
            if (r >= R) { r -= R; row++; }
        }
    }
}

// ---- the affine pair w of the general QMF class (x ~ w0 + w1 u v^T; qmf.py:104-105, 141-147) ---------------------------
// Xp = safe_divide(X - w0, w1) (factorization/utils.py:18-33: |w1| < eps -> eps * sign(w1)); W [B][2]
__global__ __launch_bounds__(256) void k_any_affine(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Xp, long per)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const float w0 = W[2 * blockIdx.y], w1 = W[2 * blockIdx.y + 1];
    float den = w1;
    if (fabsf(w1) < LRF_EPS) den = LRF_EPS * ((w1 > 0.f) ? 1.f : (w1 < 0.f ? -1.f : 0.f));
    Xp[(long)blockIdx.y * per + e] = (X[(long)blockIdx.y * per + e] - w0) / den;
}

// update_w, first half: per 64-row tile the fp64 sums of z, z^2, x, x z over its elements, z = u v^T (k-ordered fp32 fma
// chain like u @ v.mT).  grid (ceil(M/64), B), 256 threads; part [B][ntiles][4].
__global__ __launch_bounds__(256) void k_any_wstats(const float* __restrict__ X, const float* __restrict__ Uf, const float* __restrict__ Vf,
                                                    int M, int N, int R, double* __restrict__ part)
{
    __shared__ double red[4][4];
    const int row0 = blockIdx.x * 64, nrows = (M - row0 < 64) ? M - row0 : 64;
    const float* Xb = X + ((long)blockIdx.y * M + row0) * N;
    const float* Ub = Uf + ((long)blockIdx.y * M + row0) * R;
    const float* Vb = Vf + (long)blockIdx.y * N * R;
    double sz = 0.0, szz = 0.0, sx = 0.0, sxz = 0.0;
    for (long e = threadIdx.x; e < (long)nrows * N; e += 256) {
        const int m = (int)(e / N), j = (int)(e - (long)m * N);
        float z = 0.f;
        for (int r = 0; r < R; r++) z = fmaf(Ub[(long)m * R + r], Vb[(long)j * R + r], z);
        const double zd = (double)z, xd = (double)Xb[e];
        sz += zd; szz = fma(zd, zd, szz); sx += xd; sxz = fma(xd, zd, sxz);
    }
    double v[4] = {sz, szz, sx, sxz};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        v[q] = wave_sum(v[q]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// update_w, second half: the tiles' sums added in order, then the least-squares line of x on z by the 2 x 2 normal equations
// (the reference: torch.linalg.lstsq on [1, z]; equal to ~1e-7 relative, parity by tolerance).  One thread per matrix.
__global__ void k_any_wsolve(const double* __restrict__ part, int ntiles, double n, float* __restrict__ W, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = 0; t < ntiles; t++)
        for (int q = 0; q < 4; q++) s[q] += part[((long)b * ntiles + t) * 4 + q];
    const double det = n * s[1] - s[0] * s[0];
    const double w1 = (det != 0.0) ? (n * s[3] - s[0] * s[2]) / det : 0.0;
    const double w0 = (s[2] - w1 * s[0]) / n;
    W[2 * b] = (float)w0;
    W[2 * b + 1] = (float)w1;
}

// fp32 factors [B][per] -> int8, matrix b at O + b * o_batch (the fused encode interleaves the planes of an image)
__global__ __launch_bounds__(256) void k_any_to_i8(const float* __restrict__ F, int8_t* __restrict__ O, long per, long o_batch)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < per) O[(long)blockIdx.y * o_batch + e] = (int8_t)F[(long)blockIdx.y * per + e];
}

// ---- initialisation -------------------------------------------------------------------------------------------
// fp64 Gram matrix of the short side: G[i][j] = sum_k A(k,i) A(k,j), A(k,i) = X[k*sgk + i*sgi]; k ascending, one fma
// chain per element (the products of two fp32 are exact in fp64).  32 x 32 tile per workgroup, upper tiles mirrored.
// grid (nt, nt, B)
__global__ __launch_bounds__(256) void k_any_gram(const float* __restrict__ X, long x_batch, long sgk, long sgi, int n, int D,
                                                  double* __restrict__ G)
{
    if (blockIdx.y < blockIdx.x) return;
    __shared__ float Is[64 * 33], Js[64 * 33];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const float* Xb = X + (long)blockIdx.z * x_batch;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int k0 = 0; k0 < D; k0 += 64) {
        const int klen = (D - k0 < 64) ? D - k0 : 64;
        __syncthreads();
        for (int e = tid; e < 64 * 32; e += 256) {
            int kk, cc;
            if (sgk == 1) { kk = e & 63; cc = e >> 6; } else { cc = e & 31; kk = e >> 5; }
            float vi = 0.f, vj = 0.f;
            if (kk < klen) {
                if (i0 + cc < n) vi = Xb[(long)(k0 + kk) * sgk + (long)(i0 + cc) * sgi];
                if (j0 + cc < n) vj = Xb[(long)(k0 + kk) * sgk + (long)(j0 + cc) * sgi];
            }
            Is[kk * 33 + cc] = vi;
            Js[kk * 33 + cc] = vj;
        }
        __syncthreads();
        for (int kk = 0; kk < klen; kk++) {
            const double a0 = (double)Is[kk * 33 + ty], a1 = (double)Is[kk * 33 + ty + 16];
            const double b0 = (double)Js[kk * 33 + tx], b1 = (double)Js[kk * 33 + tx + 16];
            acc[0][0] = fma(a0, b0, acc[0][0]);
            acc[0][1] = fma(a0, b1, acc[0][1]);
            acc[1][0] = fma(a1, b0, acc[1][0]);
            acc[1][1] = fma(a1, b1, acc[1][1]);
        }
    }
    double* Gb = G + (long)blockIdx.z * n * n;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = i0 + ty + 16 * p, j = j0 + tx + 16 * q;
            if (i < n && j < n) {
                Gb[(long)i * n + j] = acc[p][q];
                if (blockIdx.x != blockIdx.y) Gb[(long)j * n + i] = acc[p][q];
            }
        }
}

// the diagonal tiles above compute (i,j) and (j,i) by the same chain of the same commuting products: exactly symmetric.

// The same Gram matrix on the fp64 matrix cores (round 3).  v_mfma_f64_16x16x4_f64 IS the chain above: per element
// d = fma(a_3, b_3, fma(a_2, b_2, fma(a_1, b_1, fma(a_0, b_0, c)))), k ascending (tools/probe/run_mfma64.py: 0 of 3072 elements
// differ from the fma chain, products of fp32 values being exact in fp64), so the result is bit for bit k_any_gram's.  The
// VALU kernel is bound by its LDS operand reads (a 2 x 2 register tile: one read per fma); here a wave owns a 32 x 32
// sub-tile as 2 x 2 MFMA tiles and reads four operands per four MFMAs.  64 x 64 tile of G per workgroup, upper tiles
// mirrored; operand layout: A lane l -> (i = l % 16, k = l / 16), B lane l -> (k = l / 16, j = l % 16), D register r ->
// (i = l / 16 + 4 r, j = l % 16).  grid (nt64, nt64, B)
typedef double f64x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_any_gram_mfma(const float* __restrict__ X, long x_batch, long sgk, long sgi, int n, int D,
                                                       double* __restrict__ G)
{
    if (blockIdx.y < blockIdx.x) return;
    constexpr int LS = 81; // row stride of the staged tiles (17 mod 64): the four k rows of an operand read fall into (all but) disjoint
                           // banks, and the staging stores are conflict-free along k as well as along i
    __shared__ __attribute__((aligned(16))) float Ss[2 * 64 * LS]; // the two staged tiles; afterwards the transposed output tile
    float* Is = Ss;
    float* Js = Ss + 64 * LS;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const bool diag = blockIdx.x == blockIdx.y;
    const float* Xb = X + (long)blockIdx.z * x_batch;
    f64x4_t acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 2; q++) acc[p][q] = (f64x4_t){0.0, 0.0, 0.0, 0.0};
    // the next 64-deep chunk travels from global memory into registers while the matrix cores work on the current one
    float ri[16], rj[16];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        const int klen = (D - k0 < 64) ? D - k0 : 64;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int e = tid + 256 * t;
            int kk, cc;
            if (sgk == 1) { kk = e & 63; cc = e >> 6; } else { cc = e & 63; kk = e >> 6; }
            float vi = 0.f, vj = 0.f;
            if (kk < klen) {
                if (i0 + cc < n) vi = Xb[(long)(k0 + kk) * sgk + (long)(i0 + cc) * sgi];
                if (!diag && j0 + cc < n) vj = Xb[(long)(k0 + kk) * sgk + (long)(j0 + cc) * sgi];
            }
            ri[t] = vi;
            rj[t] = vj;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < D; k0 += 64) {
        const int klen = (D - k0 < 64) ? D - k0 : 64;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int e = tid + 256 * t;
            int kk, cc;
            if (sgk == 1) { kk = e & 63; cc = e >> 6; } else { cc = e & 63; kk = e >> 6; }
            Is[kk * LS + cc] = ri[t];
            if (!diag) Js[kk * LS + cc] = rj[t];
        }
        __syncthreads();
        if (k0 + 64 < D) fetch(k0 + 64);
        const float* Jt = diag ? Is : Js;
        const int ksteps = (klen + 3) >> 2; // rows past klen hold zeros: fma(0, 0, acc) = acc
        for (int ks = 0; ks < ksteps; ks++) {
            const int kr = (4 * ks + g) * LS;
            const double a0 = (double)Is[kr + 32 * wi + li], a1 = (double)Is[kr + 32 * wi + 16 + li];
            const double b0 = (double)Jt[kr + 32 * wj + li], b1 = (double)Jt[kr + 32 * wj + 16 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    double* Gb = G + (long)blockIdx.z * n * n;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + 32 * wi + 16 * p + g + 4 * r, j = j0 + 32 * wj + 16 * q + li;
                if (i < n && j < n) Gb[(long)i * n + j] = acc[p][q][r];
            }
    if (diag) return;
    // the mirror image G[j][i]: straight from the registers a lane would write 8 bytes every n doubles (a 64-byte sector per
    // element; the two Gram kernels spent most of their time there), so the tile goes through LDS and leaves row by row
    constexpr int TS = 65;
    double* Tt = reinterpret_cast<double*>(Ss); // [64 j][TS]: 33 KB of the 41 KB
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) Tt[(32 * wj + 16 * q + li) * TS + 32 * wi + 16 * p + g + 4 * r] = acc[p][q][r];
    __syncthreads();
    for (int jl = wave; jl < 64; jl += 4) {
        const int j = j0 + jl, i = i0 + lane;
        if (j < n && i < n) Gb[(long)j * n + i] = Tt[jl * TS + lane];
    }
}

// Top-R eigen-pairs of the n x n Gram matrix G (global, destroyed): E1 = e sqrt(sigma), E2 = e / sqrt(sigma), fp32 [n][R].
// Householder tridiagonalisation of a symmetric n x n matrix, 64 < n <= 64 NC (NC = 2, 3), with the matrix in REGISTERS: the
// [M,192] Gram matrices of svd_encode and of the RGB colour-space branch, 16 x 8 / 8 x 16 patches.  k_any_eig<1> walks its
// matrix in global memory three times per step (8.3 ms per 256 matrices of 192 x 192, ~95 % of svd_encode's initialisation);
// here 256 NC threads hold it in registers — thread (lane i, wave b) keeps A[16 b + j][64 c + i], c < NC, j < 16, as NC
// 16-double vectors, so row k is read with a register index.  Arithmetic: k_init's step (one reduction for sigma, t = 1 / (sigma + |x0| nrm),
// p = t A v from chains over 16-row sub-blocks), the rank-2 update as two fmas per element; four barriers per step.  Output: row k of A keeps the reflector v_k (i > k),
// td = d[n], e[n], tau[n] for k_any_eig<1>, which then starts at its eigenvalue stage.
// lane `src_lane` of a double register, as a wave-uniform value (two v_readlane_b32: the result lives in SGPRs)
__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// lane j of the quad (CTRL = j * 0x55: quad_perm [j, j, j, j]) of a double register
template <int CTRL>
__device__ __forceinline__ double quad_bcast_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int NC>
__global__ __launch_bounds__(256 * NC) void k_any_tridiag_reg(double* __restrict__ G, int n, double* __restrict__ TD)
{
    constexpr int NP = 64 * NC; // padded side
    constexpr int CP = NP + 2; // pitch of cpart: the four lanes of a quad read four sub-block rows of one column (distinct banks)
    __shared__ __attribute__((aligned(16))) double xrow2[2 * NP], cpart[4 * NC * CP], pbuf[NP], vbuf[NP], tbuf[2]; // cpart: one chain sum per 16-row sub-block and column
    __shared__ int sflag[2];
    double* A = G + (long)blockIdx.x * n * n;
    double* td = TD + (long)blockIdx.x * 3 * n;
    const int tid = threadIdx.x, lane = tid & 63;
    // Wave b holds the 16-row sub-block b (rows 16 b .. 16 b + 15) on ALL columns: a thread keeps NC 16-double vectors, one per
    // column chunk (columns lane, 64 + lane, ...).  Round 3's layout gave a wave one column chunk and 16 NC consecutive rows:
    // every wave-uniform operand (v and w at a ROW: two v_readlane_b32 per double) then served one product, and the
    // instruction count per element was 3 in the product A v and 6 in the update; with all chunks in one thread it serves NC
    // of them: (2 + NC) / NC and (4 + 2 NC) / NC.  The trailing matrix shrinks from the top, so the last wave is the
    // critical one: 16 rows x the live chunks (240 / 176 / 112 instructions per step as the chunks finish, against 432 in
    // every step for round 3's last row group).  Which thread forms a sub-block's chain does not change a bit: the chains
    // (sixteen consecutive rows from zero) and the order their sums are added in are the oracle's.
    const int b = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the waves of a SIMD get different priorities, the later sub-blocks (the ones that work until the last step) first
#ifndef LRF_REG_NO_PRIO
    if (b >= 3 * NC) __builtin_amdgcn_s_setprio(3);
    else if (b >= 2 * NC) __builtin_amdgcn_s_setprio(2);
    else if (b >= NC) __builtin_amdgcn_s_setprio(1);
#endif
    d16 Ar[NC];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int r = 16 * b + jj, col = 64 * c + lane;
            Ar[c][jj] = (r < n && col < n) ? A[(long)r * n + col] : 0.0;
        }
    // Every wave holds v and w on ALL columns (NC values per lane: v from the reflector waves or LDS, w computed in every
    // wave, same bits), so the values a thread needs at its ROWS come out of the wave's own registers by v_readlane
    // (wave-uniform, in SGPRs) instead of broadcast LDS reads — twelve waves reading 144 doubles each per step had made the
    // LDS the bottleneck (1.26 -> 0.5 ms per 256 matrices of 192 x 192).  The rows of sub-block b are the lanes
    // 16 (b & 3) .. + 15 of chunk b >> 2.
    auto pick = [&](const double (&q)[NC], int chunk) __attribute__((always_inline)) {
        double r = q[0];
        if (NC > 1 && chunk == 1) r = q[1];
        if (NC > 2 && chunk == 2) r = q[NC - 1];
        return r;
    };
    const int rl0 = 16 * (b & 3), rch = b >> 2; // first lane and chunk of this wave's rows in v / w
#ifdef LRF_REG_STAMPS
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime(), tq0 = tq;
#define REG_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); st[i] += tn_ - tq; tq = tn_; __builtin_amdgcn_sched_barrier(0); }
#else
#define REG_STAMP(i)
#endif
    for (int k = 0; k < n - 2; k++) {
        double* xrow = xrow2 + NP * (k & 1); // double-buffered: a slower wave may still be reading the other one
        const bool holds_k = b == (k >> 4); // the wave that holds row k publishes it (entries up to column k as zeros)
        if (holds_k) {
            const int jj = k & 15;
#pragma unroll
            for (int c = 0; c < NC; c++) xrow[64 * c + lane] = (64 * c + lane > k) ? Ar[c][jj] : 0.0;
        }
        __syncthreads();
        REG_STAMP(1);
        // The reflector's scalars: ONE wave per SIMD computes them (the last four waves: same bits in all of them), the others
        // wait and take v from LDS.  With every wave computing them the three waves of a SIMD issue the same ~150 dependent
        // fp64 instructions one after the other, and a step is bound by what a SIMD issues (stamps: ~4.2 k cycles per step,
        // a third of them these chains).
        double vk[NC], t = 0.0, alpha = 0.0; // v on the columns lane, 64 + lane, ...
        bool live_step;
        if (b >= 4 * NC - 4) {
            double xs[NC], sq = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < NC; c2++) {
                xs[c2] = xrow[64 * c2 + lane];
                sq = fma(xs[c2], xs[c2], sq);
            }
            const double sigma = wave_tree64(sq); // DPP / permlane tree of k_init (lrf_kernels.hip): no LDS crossbar
            live_step = sigma > LRF_SIGMA_TINY;
            if (live_step) {
                const double x0 = readlane_f64(pick(xs, (k + 1) >> 6), (k + 1) & 63);
                const double nrm = sqrt(sigma);
                alpha = (x0 >= 0.0) ? -nrm : nrm;
                const double vfix = x0 - alpha;
                t = 1.0 / fma(fabs(x0), nrm, sigma);
#pragma unroll
                for (int c2 = 0; c2 < NC; c2++) vk[c2] = (64 * c2 + lane == k + 1) ? vfix : xs[c2];
            } else {
#pragma unroll
                for (int c2 = 0; c2 < NC; c2++) vk[c2] = 0.0;
            }
            if (b == 4 * NC - 1) {
#pragma unroll
                for (int c2 = 0; c2 < NC; c2++) vbuf[64 * c2 + lane] = vk[c2];
                if (lane == 0) { sflag[0] = live_step ? 1 : 0; tbuf[0] = t; }
            }
        }
        __syncthreads();
        if (b < 4 * NC - 4) {
            live_step = sflag[0] != 0;
            t = tbuf[0];
#pragma unroll
            for (int c2 = 0; c2 < NC; c2++) vk[c2] = vbuf[64 * c2 + lane];
        }
        if (!live_step) { // wave-uniform, the same in every wave
            if (tid == 0) { td[n + k] = 0.0; td[2 * n + k] = 0.0; }
            continue; // the next step writes the other buffer, and its barrier orders the step after
        }
        REG_STAMP(2);
        if (holds_k) { // v_k for the back-transformation
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int col = 64 * c + lane;
                if (col > k && col < n) A[(long)k * n + col] = vk[c];
            }
        }
        const bool rows_live = 16 * b + 15 > k; // wave-uniform: some row of this wave is still in the trailing matrix
        { // matvec partials of this thread's sub-block: one chain per column; finished rows (v = 0 there) and chunks are skipped
            double cs[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) cs[c] = 0.0;
            if (rows_live) {
                const double src = pick(vk, rch);
#pragma unroll
                for (int jj = 0; jj < 16; jj++) { // (finished chunks are not skipped: v = 0 there, and their sums are masked below)
                    const double vj = readlane_f64(src, rl0 + jj);
#pragma unroll
                    for (int c = 0; c < NC; c++) cs[c] = fma(Ar[c][jj], vj, cs[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) cpart[b * CP + 64 * c + lane] = (64 * c + lane > k) ? cs[c] : 0.0;
        }
        REG_STAMP(3);
        __syncthreads();
        // p = t A v, ONCE per column: wave w combines the sub-block sums of the columns 16 w .. 16 w + 15 — lane (column
        // 16 w + (lane >> 2), row group g = lane & 3) adds its group's NC sums in order, the four groups meet inside the quad
        // (DPP), in the oracle's order ((g0 + g1) + g2) + g3 — and publishes them; a third barrier.  (With every wave reading
        // all 4 NC sums of all its columns the LDS sets the pace of the step: 221 KB per step at n = 192.  One wave doing all
        // scalar work of a step — the reflector, then p, K, w — with v and w handed out through LDS was measured too: four
        // barriers, slower: the redundant form lets the waves of a SIMD drift apart, so that one's scalar chain runs under
        // another's products.)
        {
            const int g = lane & 3, ci = 16 * b + (lane >> 2);
            double cgv = cpart[(NC * g) * CP + ci];
#pragma unroll
            for (int s2i = 1; s2i < NC; s2i++) cgv = cgv + cpart[(NC * g + s2i) * CP + ci];
            const double c0 = quad_bcast_f64<0x00>(cgv), c1 = quad_bcast_f64<0x55>(cgv), c2q = quad_bcast_f64<0xaa>(cgv), c3 = quad_bcast_f64<0xff>(cgv);
            const double pv = t * (((c0 + c1) + c2q) + c3);
            if (g == 0) pbuf[ci] = pv;
        }
        __syncthreads();
        REG_STAMP(4);
        // every wave: K, w on all columns
        double wk[NC];
        {
            double pk[NC], s2 = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < NC; c2++) {
                pk[c2] = pbuf[64 * c2 + lane];
                s2 = fma(pk[c2], vk[c2], s2);
            }
            const double K = (0.5 * t) * wave_tree64(s2);
#pragma unroll
            for (int c2 = 0; c2 < NC; c2++) wk[c2] = fma(-K, vk[c2], pk[c2]);
            if (tid == 256 * NC - 1) { td[n + k] = alpha; td[2 * n + k] = t; } // (a thread of the last wave: it has alpha)
        }
        REG_STAMP(5);
        // rank-2 update as two fmas per element (the four-instruction commutative form of k_init doubled the kernel's dominant
        // term).  Element (r, c) and its mirror image may differ in the last bit: the FULL matrix is carried.  v, w are zero up
        // to k: finished rows and chunks are skipped.
        if (rows_live) {
            const double sv = pick(vk, rch), sw = pick(wk, rch);
#pragma unroll
            for (int jj = 0; jj < 16; jj++) { // (finished chunks are not skipped: v = w = 0 there leaves the element as it is)
                const double vj = readlane_f64(sv, rl0 + jj), wj = readlane_f64(sw, rl0 + jj);
#pragma unroll
                for (int c = 0; c < NC; c++) Ar[c][jj] = fma(-vj, wk[c], fma(-wj, vk[c], Ar[c][jj]));
            }
        }
#ifdef LRF_REG_STAMPS
        asm volatile("" ::"v"(Ar[0][0]), "v"(Ar[NC - 1][15]));
#endif
        REG_STAMP(6);
    }
#ifdef LRF_REG_STAMPS
    if (lane == 0 && blockIdx.x < 1024) { // every wave's lane 0: [matrix][wave][8]
        unsigned long long* o = g_stamps + 8 * (16 * blockIdx.x + b);
        o[0] = __builtin_amdgcn_s_memtime() - tq0;
        for (int q = 1; q < 7; q++) o[q] = st[q];
    }
#endif
    // d = diagonal, e[n-2] = A[n-1][n-2]
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int r = 16 * b + jj, col = 64 * c + lane;
            if (r == col && r < n) td[r] = Ar[c][jj];
            if (r == n - 1 && col == n - 2) td[n + n - 2] = Ar[c][jj];
        }
    if (tid == 0) { td[n + n - 1] = 0.0; td[2 * n + n - 2] = 0.0; td[2 * n + n - 1] = 0.0; }
}

#ifndef LRF_ANY_BLK_ROWS
#define LRF_ANY_BLK_ROWS 16 // rows per load batch of k_any_tridiag_blk, times the columns a thread owns
#endif
// Blocked Householder tridiagonalisation for n > 192 (round 3; oracle: any_tridiag_blocked).  The unblocked loops of
// k_any_eig move the whole trailing matrix through the CU two or three times per step (read for the product, read + write
// for the rank-2 update): with hundreds of matrices in flight that traffic, not latency, is the time (256 matrices of
// 512 x 512: ~180 GB).  Here a panel of NB = 16 (n <= 512), 8 (n <= 1024) or 4 steps leaves the trailing matrix in memory as it was at the panel's
// start (A0) and keeps the panel's reflectors V_m and their companions W_m in LDS (2 NB n doubles, 128 KB at most):
//   row k of the current matrix   x = A0[k][.] - sum_m (V_m[k] W_m + W_m[k] V_m)
//   its product                   A v = A0 v - sum_m (V_m (W_m . v) + W_m (V_m . v))      (ONE read pass over A0 per step)
//   once per panel                A -= sum_m (V_m W_m^T + W_m V_m^T)                      (one read + write pass per NB steps)
// Thread t owns the columns t, t + 256, ... and touches only those columns of A, so no global-memory hand-off between
// threads exists; LDS carries v, the panel and the reduction partials.  Output as k_any_tridiag_reg: row k of A keeps v_k
// (columns > k), td = d[n], e[n], tau[n].  Dynamic LDS: (2 NB (n rounded up to even) + 64 + 8 NB + 24) doubles.
template <int NCT>
__global__ __launch_bounds__(256) void k_any_tridiag_blk(double* __restrict__ G, int n, double* __restrict__ TD)
{
    constexpr int NB = NCT == 1 ? 16 : 32 / NCT, UB = LRF_ANY_BLK_ROWS / NCT; // steps per panel (64 KB of LDS at n <= 256, up to 128 KB
                                                                             // above); rows per load batch and thread (two batches in flight)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ns = (n + 1) & ~1;                  // row stride of the panel buffers
    double* Vp = reinterpret_cast<double*>(smem); // [NB][ns]
    double* Wp = Vp + (size_t)NB * ns;            // [NB][ns] (+ 64 doubles of slack: row reads run past n unguarded)
    double* Lgh = Wp + (size_t)NB * ns + 64;      // [4 waves][2 NB]
    double* Lpart = Lgh + 8 * NB;                 // [16]: two sets of four wave partials, used in turn
    double* Lscal = Lpart + 16;                   // [8]
    double* A = G + (long)blockIdx.x * n * n;
    double* td = TD + (long)blockIdx.x * 3 * n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block_sum's value (wave trees, then ((p0 + p1) + p2) + p3) on the DPP tree and with ONE barrier: consecutive sums use
    // different partial slots, and between two uses of the same slot lies the barrier of the sum in between
    auto bsum = [&](double v, int slot) __attribute__((always_inline)) {
        v = wave_tree64(v);
        if (lane == 0) Lpart[4 * slot + wave] = v;
        __syncthreads();
        return ((Lpart[4 * slot] + Lpart[4 * slot + 1]) + Lpart[4 * slot + 2]) + Lpart[4 * slot + 3];
    };
    // Loads are unconditional (a guarded load costs an exec-mask branch and a dozen scalar instructions — the first version of
    // this kernel spent ~150 cycles per row on them): rows clamp to n - 1, columns to the thread's last valid one, and what a
    // pass makes of columns it does not own is discarded where it is used (cc is zeroed for i <= k, stores are guarded).
    int ci[NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) ci[c] = (tid + 256 * c < n) ? tid + 256 * c : n - 1;
    auto load_rows = [&](int r0, double (&a)[UB][NCT]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int r = (r0 + u < n) ? r0 + u : n - 1; // wave-uniform
            const double* Ar = A + (long)r * n;
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                a[u][c] = Ar[ci[c]];
            }
        }
    };
    // A step is a chain of memory round trips unless they are taken off it: row k of A0 comes out of the registers of the
    // pass before (it is the first row of the previous step's product, or of the panel update), and the first two batches of
    // this step's product pass are requested before the reflector is computed (they do not depend on it).
#ifdef LRF_BLK_STAMPS
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime(), tq0 = tq;
#define BLK_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); st[i] += tn_ - tq; tq = tn_; }
#else
#define BLK_STAMP(i)
#endif
    double xrow[NCT];
    bool have_row = false; // xrow = A0[k][own columns >= k] for the coming step
    for (int k0 = 0; k0 < n - 2; k0 += NB) {
        const int np = (n - 2 - k0 < NB) ? n - 2 - k0 : NB;
        for (int j = 0; j < np; j++) {
            const int k = k0 + j;
            double a0[UB][NCT], a1[UB][NCT];
            load_rows(k + 1, a0);
            load_rows(k + 1 + UB, a1);
            // ---- the current row k from A0 and the panel so far
            double x[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (have_row) x[c] = (i < n && i >= k) ? xrow[c] : 0.0;
                else x[c] = (i < n && i >= k) ? A[(long)k * n + i] : 0.0;
            }
#pragma unroll 4
            for (int m = 0; m < j; m++) {
                const double vk = Vp[(size_t)m * ns + k], wk = Wp[(size_t)m * ns + k];
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n && i >= k) {
                        x[c] = fma(-vk, Wp[(size_t)m * ns + i], x[c]);
                        x[c] = fma(-wk, Vp[(size_t)m * ns + i], x[c]);
                    }
                }
            }
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (i == k) { td[k] = x[c]; x[c] = 0.0; }
                if (i == k + 1) Lscal[0] = x[c];
                s = fma(x[c], x[c], s);
            }
            const double sigma = bsum(s, 0); // its barrier publishes Lscal[0]
            if (!(sigma > LRF_SIGMA_TINY)) { // the same in every thread
                if (tid == 0) { td[n + k] = 0.0; td[2 * n + k] = 0.0; }
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n) { Vp[(size_t)j * ns + i] = 0.0; Wp[(size_t)j * ns + i] = 0.0; }
                    xrow[c] = a0[0][c]; // row k + 1 of A0 (columns >= k + 1)
                }
                have_row = true;
                __syncthreads();
                continue;
            }
            BLK_STAMP(1)
            const double x0 = Lscal[0];
            const double nrm = sqrt(sigma);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            const double t = 1.0 / fma(fabs(x0), nrm, sigma);
            double v[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                v[c] = (i == k + 1) ? x0 - alpha : x[c];
                if (i < n) {
                    Vp[(size_t)j * ns + i] = v[c];
                    if (i > k) A[(long)k * n + i] = v[c];
                }
            }
            if (tid == 0) { td[n + k] = alpha; td[2 * n + k] = t; }
            // ---- g_m = W_m . v, h_m = V_m . v: thread partials over its own columns, one tree per wave, four partials in LDS
#pragma unroll 4
            for (int m = 0; m < j; m++) {
                double sg = 0.0, sh = 0.0;
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n) {
                        sg = fma(Wp[(size_t)m * ns + i], v[c], sg);
                        sh = fma(Vp[(size_t)m * ns + i], v[c], sh);
                    }
                }
                sg = wave_tree64(sg);
                sh = wave_tree64(sh);
                if (lane == 0) { Lgh[wave * 2 * NB + 2 * m] = sg; Lgh[wave * 2 * NB + 2 * m + 1] = sh; }
            }
            __syncthreads(); // V_j and the partials are visible
            BLK_STAMP(2)
            // ---- A0 v: rows in batches of UB, two batches in flight (a read-only pass)
            const double* Lv = Vp + (size_t)j * ns;
            double cc[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) { cc[c] = 0.0; xrow[c] = a0[0][c]; }
            have_row = true;
            {
                auto matvec_rows = [&](int j0, const double (&a)[UB][NCT]) __attribute__((always_inline)) {
                    double lv[UB]; // read without the row guard (the buffers have slack), so that the reads pair up
#pragma unroll
                    for (int u = 0; u < UB; u++) lv[u] = Lv[j0 + u];
#pragma unroll
                    for (int u = 0; u < UB; u++) {
                        const double vj = (j0 + u < n) ? lv[u] : 0.0; // rows past n: fma(a, 0, cc) = cc (a is the clamped row, finite)
#pragma unroll
                        for (int c = 0; c < NCT; c++) cc[c] = fma(a[u][c], vj, cc[c]);
                    }
                };
                for (int j0 = k + 1; j0 < n; j0 += 2 * UB) {
                    matvec_rows(j0, a0);
                    load_rows(j0 + 2 * UB, a0); // unconditional (rows clamp): a guarded prefetch makes every wait count conservative
                    matvec_rows(j0 + UB, a1);
                    load_rows(j0 + 3 * UB, a1);
                }
            }
            BLK_STAMP(3)
#pragma unroll 4
            for (int m = 0; m < j; m++) {
                const double gm = ((Lgh[2 * m] + Lgh[2 * NB + 2 * m]) + Lgh[4 * NB + 2 * m]) + Lgh[6 * NB + 2 * m];
                const double hm = ((Lgh[2 * m + 1] + Lgh[2 * NB + 2 * m + 1]) + Lgh[4 * NB + 2 * m + 1]) + Lgh[6 * NB + 2 * m + 1];
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n && i > k) {
                        cc[c] = fma(-Vp[(size_t)m * ns + i], gm, cc[c]);
                        cc[c] = fma(-Wp[(size_t)m * ns + i], hm, cc[c]);
                    }
                }
            }
            s = 0.0;
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (!(i < n && i > k)) cc[c] = 0.0;
                cc[c] = t * cc[c];
                s = fma(cc[c], v[c], s);
            }
            const double Kc = (0.5 * t) * bsum(s, 1);
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (i < n) Wp[(size_t)j * ns + i] = fma(-Kc, v[c], cc[c]);
            }
            __syncthreads();
        }
        BLK_STAMP(4)
        // ---- the panel's rank-2 np update of the trailing matrix (rows and columns >= kend): the thread's own columns of the
        // panel in registers, the rows' values by broadcast LDS reads; the next batch's loads go out before this one's stores
        const int kend = k0 + np;
        // per reflector m and batch: the rows' V_m / W_m values (UB consecutive doubles each, broadcast reads that pair up) and the
        // thread's own columns of V_m / W_m; every element still receives its updates in ascending m
        auto update_rows = [&](int r0, double (&a)[UB][NCT]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < NB; m++) {
                if (m < np) {
                    double vr[UB], wr[UB], vic[NCT], wic[NCT];
#pragma unroll
                    for (int u = 0; u < UB; u++) { vr[u] = Vp[(size_t)m * ns + r0 + u]; wr[u] = Wp[(size_t)m * ns + r0 + u]; }
#pragma unroll
                    for (int c = 0; c < NCT; c++) {
                        const int i = tid + 256 * c;
                        vic[c] = (i < n) ? Vp[(size_t)m * ns + i] : 0.0;
                        wic[c] = (i < n) ? Wp[(size_t)m * ns + i] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < UB; u++)
#pragma unroll
                        for (int c = 0; c < NCT; c++) a[u][c] = fma(-wr[u], vic[c], fma(-vr[u], wic[c], a[u][c]));
                }
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int r = r0 + u;
                if (r < n) {
                    double* Ar = A + (long)r * n;
#pragma unroll
                    for (int c = 0; c < NCT; c++) {
                        const int i = tid + 256 * c;
                        if (i < n && i >= kend) Ar[i] = a[u][c];
                    }
                }
            }
        };
        {
            double a0[UB][NCT], a1[UB][NCT];
            load_rows(kend, a0);
            load_rows(kend + UB, a1);
            for (int r0 = kend; r0 < n; r0 += 2 * UB) {
                update_rows(r0, a0);
                if (r0 == kend) {
#pragma unroll
                    for (int c = 0; c < NCT; c++) xrow[c] = a0[0][c]; // the updated row kend: the next panel's first row
                }
                load_rows(r0 + 2 * UB, a0);
                update_rows(r0 + UB, a1);
                load_rows(r0 + 3 * UB, a1);
            }
            have_row = true;
        }
        __syncthreads(); // the panel buffers are rewritten by the next panel
        BLK_STAMP(5)
    }
    // d[n-2], d[n-1], e[n-2] from the updated matrix (each element read by the thread that owns its column)
#pragma unroll
    for (int c = 0; c < NCT; c++) {
        const int i = tid + 256 * c;
        if (i == n - 2) { td[n - 2] = A[(long)(n - 2) * n + i]; td[n + n - 2] = A[(long)(n - 1) * n + i]; }
        if (i == n - 1) td[n - 1] = A[(long)(n - 1) * n + i];
    }
    if (tid == 0) { td[n + n - 1] = 0.0; td[2 * n + n - 2] = 0.0; td[2 * n + n - 1] = 0.0; }
#ifdef LRF_BLK_STAMPS
    if (tid == 0 && blockIdx.x < 16384) {
        unsigned long long* o = g_stamps + 8 * blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - tq0;
        for (int q = 1; q < 6; q++) o[q] = st[q];
    }
#endif
#undef BLK_STAMP
}

// The blocked tridiagonalisation on the LOWER TRIANGLE only, 192 < n <= 512 (round 3; oracle: any_tridiag_sym).
// k_any_tridiag_blk's product pass reads the whole trailing square of a symmetric matrix: at n = 512 that pass runs at the
// HBM rate, at n = 256 at what a CU draws from the Infinity Cache.  Here an element A0[r][i], r >= i, is loaded once and
// used twice: for the column part cc[i] += A0[r][i] v[r] and for the row part y[r] += A0[r][i] v[i].  Work is dealt by ROWS:
// sub-tile r0 (sixteen rows) belongs to wave (r0 / 16) % NW, which walks the 64-column chunks J <= r0 / 64 of it; lane =
// column.  A body (r0, J): sixteen row loads (the next body's are in flight), the column part as a 16-fma chain added to the
// wave's partial Cp[wave][column] (LDS), the sixteen products a v[column] through a wave-private LDS tile to lane = (row,
// quarter): four 16-term sums per row, combined ((q0 + q1) + q2) + q3 into Yrow[J][row] — one writer per (J, row).  The
// owner thread of column i then takes the waves' partials in wave order plus the chunks' row values in order.  Row k of the current
// matrix is column k of the triangle: the lane that owns column k + 1 leaves what it loaded in Lx for the next step.  The
// panel update touches the triangle only; the upper triangle keeps the reflectors (row k: v_k).  Everything else — panel
// algebra, reductions, outputs — as k_any_tridiag_blk.  NB = 16 / NCT.  LDS at n = 512: 155 KB (one workgroup per CU).
#define LRF_WAVE_SYNC()                                     \
    do {                                                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                    \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
template <int NCT, int NW>
__global__ __launch_bounds__(64 * NW) void k_any_tridiag_sym(double* __restrict__ G, int n, double* __restrict__ TD)
{
    constexpr int NB = 8, NCH = 4 * NCT, TP = 65; // NW waves: 4, or 8 for n <= 256 (two per SIMD; the threads past 256 own no column)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ns = ((n + 1) & ~1) + 16; // row stride of the LDS vectors: sixteen zeros behind every panel row (row reads run past n)
    const int nch = (n + 63) >> 6;
    double* Vp = reinterpret_cast<double*>(smem); // [NB][ns]
    double* Wp = Vp + (size_t)NB * ns;            // [NB][ns]
    double* Lx = Wp + (size_t)NB * ns;            // [ns]: column k of the triangle for the coming step
    double* Yrow = Lx + ns;                       // [NCH][ns]
    double* Cp = Yrow + (size_t)NCH * ns;         // [NW][ns]
    double* Lt = Cp + (size_t)NW * ns;            // [NW][16 TP]
    double* Lq = Lt + NW * 16 * TP;               // [NW][64]
    double* Lgh = Lq + 64 * NW;                   // [4][2 NB]
    double* Lpart = Lgh + 8 * NB;                 // [16]
    double* Lscal = Lpart + 16;                   // [8]
    double* A = G + (long)blockIdx.x * n * n;
    double* td = TD + (long)blockIdx.x * 3 * n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* Ltw = Lt + wave * 16 * TP;
    double* Lqw = Lq + wave * 64;
    double* Cpw = Cp + (size_t)wave * ns;
    auto bsum = [&](double v, int slot) __attribute__((always_inline)) {
        v = wave_tree64(v);
        if (lane == 0 && wave < 4) Lpart[4 * slot + wave] = v; // (waves past the fourth own no column: their sums are zero)
        __syncthreads();
        return ((Lpart[4 * slot] + Lpart[4 * slot + 1]) + Lpart[4 * slot + 2]) + Lpart[4 * slot + 3];
    };
    // body (r0, J): rows r0 .. r0 + 15 (clamped), column 64 J + lane (clamped): unconditional loads
    auto load_body = [&](int r0, int J, double (&a)[16]) __attribute__((always_inline)) {
        const int col = 64 * J + lane, colc = col < n ? col : n - 1;
        if (r0 + 16 <= n) { // wave-uniform: no row clamps
            const double* p0 = A + (long)r0 * n + colc;
#pragma unroll
            for (int u = 0; u < 16; u++) a[u] = p0[(long)u * n];
        } else {
            const int rb = (r0 < n) ? r0 : n - 1;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int r = (rb + u < n) ? rb + u : n - 1;
                a[u] = A[(long)r * n + colc];
            }
        }
    };
    // the wave's bodies from sub-tile row rbeg0 (a multiple of 16) on, chunks Jlo .. r0 / 64: (r0, J) -> next; r0 >= n: done
    auto advance = [&](int& r0, int& J, int Jlo) __attribute__((always_inline)) {
        if (J < (r0 >> 6)) J++;
        else { r0 += 16 * NW; J = Jlo; }
    };
    // the wave's bodies in order, the loads of the next two in flight (one workgroup of four waves per CU: what hides the
    // memory latency is what a wave itself has outstanding — 16 KB per wave here)
    auto pipeline = [&](int rbeg, int Jlo, auto&& fn) __attribute__((always_inline)) {
        int rA = rbeg + 16 * ((wave - (rbeg >> 4)) & (NW - 1)), JA = Jlo;
        if (!(rA < n)) return;
        // every prefetch is issued whether or not its body exists (a body past the end re-reads clamped rows): with a load
        // under a branch the compiler must assume it was not issued and waits for younger loads than the one it needs
        double b0[16], b1[16], b2[16];
        int rB = rA, JB = JA;
        advance(rB, JB, Jlo);
        load_body(rA, JA, b0);
        load_body(rB, JB, b1);
        for (;;) {
            int rC = rB, JC = JB;
            advance(rC, JC, Jlo);
            load_body(rC, JC, b2);
            fn(rA, JA, b0);
            if (!(rB < n)) break;
            int rD = rC, JD = JC;
            advance(rD, JD, Jlo);
            load_body(rD, JD, b0);
            fn(rB, JB, b1);
            if (!(rC < n)) break;
            int rE = rD, JE = JD;
            advance(rE, JE, Jlo);
            load_body(rE, JE, b1);
            fn(rC, JC, b2);
            if (!(rD < n)) break;
            rA = rD; JA = JD; rB = rE; JB = JE;
        }
    };
#ifdef LRF_BLK_STAMPS
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime(), tq0 = tq;
#define SYM_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); st[i] += tn_ - tq; tq = tn_; }
#else
#define SYM_STAMP(i)
#endif
    for (int e = tid; e < 2 * NB * ns; e += 64 * NW) Vp[e] = 0.0; // Vp and Wp: the padding stays zero
    __syncthreads();
    bool have_x = false;
    for (int k0 = 0; k0 < n - 2; k0 += NB) {
        const int np = (n - 2 - k0 < NB) ? n - 2 - k0 : NB;
        for (int j = 0; j < np; j++) {
            const int k = k0 + j;
            // ---- row k of the current matrix: column k of the triangle and the panel so far
            double x[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (have_x) x[c] = (i < n && i >= k) ? Lx[i] : 0.0;
                else x[c] = (i < n && i >= k) ? A[(long)i * n + k] : 0.0;
            }
#pragma unroll 4
            for (int m = 0; m < j; m++) {
                const double vk = Vp[(size_t)m * ns + k], wk = Wp[(size_t)m * ns + k];
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n && i >= k) {
                        x[c] = fma(-vk, Wp[(size_t)m * ns + i], x[c]);
                        x[c] = fma(-wk, Vp[(size_t)m * ns + i], x[c]);
                    }
                }
            }
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (i == k) { td[k] = x[c]; x[c] = 0.0; }
                if (i == k + 1) Lscal[0] = x[c];
                s = fma(x[c], x[c], s);
            }
            const double sigma = bsum(s, 0); // its barrier publishes Lscal[0]
            if (!(sigma > LRF_SIGMA_TINY)) { // the same in every thread
                if (tid == 0) { td[n + k] = 0.0; td[2 * n + k] = 0.0; }
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n) { Vp[(size_t)j * ns + i] = 0.0; Wp[(size_t)j * ns + i] = 0.0; }
                }
                have_x = false; // no pass ran: the next step reads its column from memory
                __syncthreads();
                continue;
            }
            SYM_STAMP(1)
            const double x0 = Lscal[0];
            const double nrm = sqrt(sigma);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            const double t = 1.0 / fma(fabs(x0), nrm, sigma);
            double v[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                v[c] = (i == k + 1) ? x0 - alpha : x[c];
                if (i < n) {
                    Vp[(size_t)j * ns + i] = v[c];
                    if (i > k) A[(long)k * n + i] = v[c]; // upper triangle: the reflector, for the back-transformation
                }
            }
            if (tid == 0) { td[n + k] = alpha; td[2 * n + k] = t; }
#pragma unroll 4
            for (int m = 0; m < j; m++) {
                double sg = 0.0, sh = 0.0;
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n) {
                        sg = fma(Wp[(size_t)m * ns + i], v[c], sg);
                        sh = fma(Vp[(size_t)m * ns + i], v[c], sh);
                    }
                }
                sg = wave_tree64(sg);
                sh = wave_tree64(sh);
                if (lane == 0 && wave < 4) { Lgh[wave * 2 * NB + 2 * m] = sg; Lgh[wave * 2 * NB + 2 * m + 1] = sh; }
            }
            for (int J = 0; J < nch; J++) Cpw[64 * J + lane < n ? 64 * J + lane : n - 1] = 0.0; // (the clamped lanes rewrite n - 1 with 0)
            __syncthreads(); // V_j, the partials and the cleared column sums are visible
            SYM_STAMP(2)
            // ---- the product pass over the triangle
            {
                const double* Lv = Vp + (size_t)j * ns;
                const int rbeg = (k + 1) & ~15, Jmin = (k + 1) >> 6;
                auto body = [&](int r0, int J, double (&a)[16]) __attribute__((always_inline)) {
                    const int col = 64 * J + lane;
                    const bool diag = (J == (r0 >> 6)); // wave-uniform
                    const double vim = (col < n && col > k) ? Lv[col < n ? col : n - 1] : 0.0;
                    double lv[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) lv[u] = Lv[r0 + u];
                    double cl = 0.0, pr[16];
                    if (diag) { // (v reads as zero past n: the padding of the panel rows)
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            const double av = (r0 + u >= col) ? a[u] : 0.0;
                            cl = fma(av, lv[u], cl);
                            pr[u] = (col < r0 + u) ? a[u] * vim : 0.0;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            cl = fma(a[u], lv[u], cl);
                            pr[u] = a[u] * vim;
                        }
                    }
                    if (col < n) Cpw[col] = Cpw[col] + cl;
                    if (J == Jmin && col == k + 1) { // wave-uniform first test; one lane: column k + 1 of the triangle for the next step
#pragma unroll
                        for (int u = 0; u < 16; u++)
                            if (r0 + u < n) Lx[r0 + u] = a[u];
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) Ltw[u * TP + lane] = pr[u];
                    LRF_WAVE_SYNC();
                    const int ur = lane & 15, q = lane >> 4;
                    double sq = 0.0;
#pragma unroll
                    for (int m = 0; m < 16; m++) sq = sq + Ltw[ur * TP + 16 * q + m];
                    Lqw[16 * q + ur] = sq;
                    LRF_WAVE_SYNC();
                    if (lane < 16 && r0 + lane < n) Yrow[(size_t)J * ns + r0 + lane] = ((Lqw[lane] + Lqw[16 + lane]) + Lqw[32 + lane]) + Lqw[48 + lane];
                    LRF_WAVE_SYNC();
                };
                pipeline(rbeg, Jmin, body);
            }
            have_x = true;
            __syncthreads(); // Cp, Yrow, Lx complete
            SYM_STAMP(3)
            double cc[NCT];
            {
                const int Jmin = (k + 1) >> 6;
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    double cv = 0.0;
                    if (i < n && i > k) {
                        cv = Cp[i];
#pragma unroll
                        for (int w = 1; w < NW; w++) cv = cv + Cp[(size_t)w * ns + i]; // the waves' partials in wave order
                        double yr = 0.0;
                        for (int J = Jmin; J <= (i >> 6); J++) yr = yr + Yrow[(size_t)J * ns + i];
                        cv = cv + yr;
                    }
                    cc[c] = cv;
                }
            }
#pragma unroll 4
            for (int m = 0; m < j; m++) {
                const double gm = ((Lgh[2 * m] + Lgh[2 * NB + 2 * m]) + Lgh[4 * NB + 2 * m]) + Lgh[6 * NB + 2 * m];
                const double hm = ((Lgh[2 * m + 1] + Lgh[2 * NB + 2 * m + 1]) + Lgh[4 * NB + 2 * m + 1]) + Lgh[6 * NB + 2 * m + 1];
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n && i > k) {
                        cc[c] = fma(-Vp[(size_t)m * ns + i], gm, cc[c]);
                        cc[c] = fma(-Wp[(size_t)m * ns + i], hm, cc[c]);
                    }
                }
            }
            s = 0.0;
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                cc[c] = t * cc[c];
                s = fma(cc[c], v[c], s);
            }
            const double Kc = (0.5 * t) * bsum(s, 1);
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (i < n) Wp[(size_t)j * ns + i] = fma(-Kc, v[c], cc[c]);
            }
            __syncthreads();
        }
        SYM_STAMP(4)
        // ---- the panel's rank-2 np update of the triangle (rows >= columns >= kend), dealt by rows as the product pass
        const int kend = k0 + np;
        {
            const int rbeg = kend & ~15, Jlo = kend >> 6;
            auto ubody = [&](int r0, int J, double (&a)[16]) __attribute__((always_inline)) {
                const int col = 64 * J + lane, colc = col < n ? col : n - 1;
#pragma unroll
                for (int m = 0; m < NB; m++) {
                    if (m < np) {
                        double vr[16], wr[16];
#pragma unroll
                        for (int u = 0; u < 16; u++) { vr[u] = Vp[(size_t)m * ns + r0 + u]; wr[u] = Wp[(size_t)m * ns + r0 + u]; }
                        const double vic = Vp[(size_t)m * ns + colc], wic = Wp[(size_t)m * ns + colc];
#pragma unroll
                        for (int u = 0; u < 16; u++) a[u] = fma(-wr[u], vic, fma(-vr[u], wic, a[u]));
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int r = r0 + u;
                    if (r < n && r >= kend && col < n && col >= kend && r >= col) {
                        A[(long)r * n + col] = a[u];
                        if (col == kend) Lx[r] = a[u]; // column kend of the updated triangle: the next panel's first row
                    }
                }
            };
            pipeline(rbeg, Jlo, ubody);
            have_x = true;
        }
        __syncthreads(); // the panel buffers are rewritten by the next panel; the updated triangle and Lx are visible
        SYM_STAMP(5)
    }
    // d[n-2], d[n-1], e[n-2] from the updated triangle
#pragma unroll
    for (int c = 0; c < NCT; c++) {
        const int i = tid + 256 * c;
        if (i == n - 2) { td[n - 2] = A[(long)(n - 2) * n + i]; td[n + n - 2] = A[(long)(n - 1) * n + i]; }
        if (i == n - 1) td[n - 1] = A[(long)(n - 1) * n + i];
    }
    if (tid == 0) { td[n + n - 1] = 0.0; td[2 * n + n - 2] = 0.0; td[2 * n + n - 1] = 0.0; }
#ifdef LRF_BLK_STAMPS
    if (tid == 0 && blockIdx.x < 16384) {
        unsigned long long* o = g_stamps + 8 * blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - tq0;
        for (int q = 1; q < 6; q++) o[q] = st[q];
    }
#endif
#undef SYM_STAMP
}

// One workgroup per matrix; thread t owns the columns t, t + 256, ... (NC = ceil(n/256) <= NCT of them).
// Workspaces per matrix: Z [R][n] doubles (vectors), Dw [2][n][Rc] doubles (the two pivots sequences of the twisted
// factorisation).  Dynamic LDS: six n-vectors, lam[R], reduction scratch.
template <int NCT>
__global__ __launch_bounds__(256) void k_any_eig(double* __restrict__ G, int n, int R, const int8_t* __restrict__ sign,
                                                 float* __restrict__ E1, float* __restrict__ E2, double* __restrict__ Zw,
                                                 double* __restrict__ Dw, int stop_after /* developer timing aid, 0 = run all */,
                                                 int rcap /* rank of the matrix at most this: columns beyond are zero */,
                                                 const double* __restrict__ td_in /* d, e, tau of k_any_tridiag_reg, or NULL */)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef LRF_REG_STAMPS // (tools/dev_stamps_eig.py: stage boundaries of every wave's lane 0, [matrix][12 + wave][8]: k_any_tridiag_reg has 0..11)
    unsigned long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int esi = 0;
#define EIG_STAMP() { __builtin_amdgcn_sched_barrier(0); est[esi++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define EIG_STAMPS_OUT() { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) { unsigned long long* o_ = g_stamps + 8 * (16 * blockIdx.x + 12 + (threadIdx.x >> 6)); for (int q_ = 0; q_ < 8; q_++) o_[q_] = est[q_]; } }
    EIG_STAMP();
#else
#define EIG_STAMP()
#define EIG_STAMPS_OUT()
#endif
    double* Lv = reinterpret_cast<double*>(smem);
    double* Lw = Lv + n;
    double* Ld = Lw + n;
    double* Le = Ld + n;
    double* Le2 = Le + n;
    double* Ltau = Le2 + n;
    double* Llam = Ltau + n;      // [R]
    double* Lpart = Llam + R;     // [16]
    double* Lscal = Lpart + 16;   // [8]
    const int Rn = R < n ? R : n;
    const int Rc = Rn < rcap ? Rn : rcap;
    double* A = G + (long)blockIdx.x * n * n;
    double* Z = Zw + (long)blockIdx.x * Rc * n;
    double* Dp = Dw + (long)blockIdx.x * 2 * n * Rc;
    double* Dm = Dp + (long)n * Rc;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int UB = 16 / NCT; // rows per batch and thread in the passes over the matrix (two batches in flight; 32 / NCT changed nothing)

    if (td_in) { // tridiagonalised already (k_any_tridiag_reg): row k of A holds v_k
        const double* tdm = td_in + (long)blockIdx.x * 3 * n;
        for (int i = tid; i < n; i += 256) { Ld[i] = tdm[i]; Le[i] = tdm[n + i]; Ltau[i] = tdm[2 * n + i]; }
        __syncthreads();
    } else if constexpr (NCT == 1) {
    // n <= 256 (svd_encode's and the RGB colour space's [M,192] matrices, 16x16 patches): three passes per step over a matrix
    // that sits in the L2; the fused form below costs one more exposed load round trip per step there (svd_encode of 256 images
    // 8.07 -> 8.46 ms), so these sizes keep the plain loop.
        for (int k = 0; k < n - 2; k++) {
            double s = 0.0;
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (i < n && i > k) { const double x = A[(long)k * n + i]; s = fma(x, x, s); }
            }
            const double sigma = block_sum(s, Lpart, tid);
            if (!(sigma > LRF_SIGMA_TINY)) {
                if (tid == 0) { Ltau[k] = 0.0; Le[k] = 0.0; }
                continue;
            }
            const double x0 = A[(long)k * n + k + 1];
            const double nrm = sqrt(sigma);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            double vi[NCT];
            s = 0.0;
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                double v = 0.0;
                if (i < n) {
                    if (i > k + 1) v = A[(long)k * n + i];
                    else if (i == k + 1) v = x0 - alpha;
                    Lv[i] = v;
                    if (i > k) A[(long)k * n + i] = v;
                }
                vi[c] = v;
                s = fma(v, v, s);
            }
            const double vn = block_sum(s, Lpart, tid); // its barriers publish Lv
            const double t = 2.0 / vn;
            if (tid == 0) { Ltau[k] = t; Le[k] = alpha; }
            double cc[NCT];
    #pragma unroll
            for (int c = 0; c < NCT; c++) cc[c] = 0.0;
            // rows in batches of UB, two batches in flight: the loads of the next batch are issued before the current one is
            // used (a plain loop waits for every single load; the matrix sits in L2 / Infinity Cache, ~1 us away)
            auto load_rows = [&](int r0, double (&a)[UB][NCT]) __attribute__((always_inline)) {
    #pragma unroll
                for (int u = 0; u < UB; u++) {
                    const int r = (r0 + u < n) ? r0 + u : n - 1;
                    const double* Ar = A + (long)r * n;
    #pragma unroll
                    for (int c = 0; c < NCT; c++) {
                        const int i = tid + 256 * c;
                        a[u][c] = Ar[i < n ? i : n - 1];
                    }
                }
            };
            auto matvec_rows = [&](int j0, const double (&a)[UB][NCT]) __attribute__((always_inline)) {
    #pragma unroll
                for (int u = 0; u < UB; u++) {
                    if (j0 + u < n) {
                        const double vj = Lv[j0 + u];
    #pragma unroll
                        for (int c = 0; c < NCT; c++) cc[c] = fma(a[u][c], vj, cc[c]);
                    }
                }
            };
            {
                double a0[UB][NCT], a1[UB][NCT];
                load_rows(k + 1, a0);
                for (int j0 = k + 1; j0 < n; j0 += 2 * UB) {
                    load_rows(j0 + UB, a1);
                    matvec_rows(j0, a0);
                    load_rows(j0 + 2 * UB, a0);
                    matvec_rows(j0 + UB, a1);
                }
            }
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (!(i < n && i > k)) cc[c] = 0.0;
            }
            s = 0.0;
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                cc[c] = t * cc[c];
                s = fma(cc[c], vi[c], s);
            }
            const double Kc = (0.5 * t) * block_sum(s, Lpart, tid);
            double wi[NCT];
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                wi[c] = fma(-Kc, vi[c], cc[c]);
                if (i < n) Lw[i] = wi[c];
            }
            __syncthreads();
            auto update_rows = [&](int r0, const double (&a)[UB][NCT]) __attribute__((always_inline)) {
    #pragma unroll
                for (int u = 0; u < UB; u++) {
                    const int r = r0 + u;
                    if (r < n) {
                        const double vr = Lv[r], wr = Lw[r];
                        double* Ar = A + (long)r * n;
    #pragma unroll
                        for (int c = 0; c < NCT; c++) {
                            const int i = tid + 256 * c;
                            if (i < n && i > k) {
                                const bool rc = r >= i; // canonical (row >= column) operand order: the matrix stays exactly symmetric
                                const double va = rc ? vr : vi[c], wa = rc ? wr : wi[c], vb = rc ? vi[c] : vr, wb = rc ? wi[c] : wr;
                                Ar[i] = fma(-wa, vb, fma(-va, wb, a[u][c]));
                            }
                        }
                    }
                }
            };
            { // the loads of the next batch go out before the stores of this one, so a batch waits for one latency, not two
                double a0[UB][NCT], a1[UB][NCT];
                load_rows(k + 1, a0);
                for (int r0 = k + 1; r0 < n; r0 += 2 * UB) {
                    load_rows(r0 + UB, a1);
                    update_rows(r0, a0);
                    load_rows(r0 + 2 * UB, a0);
                    update_rows(r0 + UB, a1);
                }
            }
            __syncthreads();
        }
    } else {
        // ---- Householder tridiagonalisation; row k of A keeps the reflector v_k.
        // Two passes over the trailing matrix per step instead of three: the rank-2 update of step k also accumulates the
        // matrix-vector product of step k + 1 (row k + 1 is updated first, its reflector v_{k+1} formed, then every updated row r
        // adds A[r][i] v_{k+1}[r] to the next product while it is still in registers), and columns i <= k are neither loaded nor
        // stored (a thread owns columns; the products and the updates only ever use i > k).
        double* Lv2 = Ld; // the next reflector (Ld is filled after the loop)
        auto load_rows = [&](int r0, int k, double (&a)[UB][NCT]) __attribute__((always_inline)) {
    #pragma unroll
            for (int u = 0; u < UB; u++) {
                const int r = (r0 + u < n) ? r0 + u : n - 1;
                const double* Ar = A + (long)r * n;
    #pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    a[u][c] = (i < n && i > k) ? Ar[i] : 0.0;
                }
            }
        };
        double vi[NCT], cc[NCT];
        double t = 0.0;
        bool have = false; // v_k (Lv, vi), t and cc = sum_{j > k} A[j][.] v_k[j] are ready for the current k
        for (int k = 0; k < n - 2; k++) {
            double* Lvk = (k & 1) ? Lv2 : Lv;       // this step's reflector
            double* Lvn = (k & 1) ? Lv : Lv2;       // the next step's
            if (!have) {
                double s = 0.0;
    #pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n && i > k) { const double x = A[(long)k * n + i]; s = fma(x, x, s); }
                }
                const double sigma = block_sum(s, Lpart, tid);
                if (!(sigma > LRF_SIGMA_TINY)) {
                    if (tid == 0) { Ltau[k] = 0.0; Le[k] = 0.0; }
                    continue;
                }
                const double x0 = A[(long)k * n + k + 1];
                const double nrm = sqrt(sigma);
                const double alpha = (x0 >= 0.0) ? -nrm : nrm;
                s = 0.0;
    #pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    double v = 0.0;
                    if (i < n) {
                        if (i > k + 1) v = A[(long)k * n + i];
                        else if (i == k + 1) v = x0 - alpha;
                        Lvk[i] = v;
                        if (i > k) A[(long)k * n + i] = v;
                    }
                    vi[c] = v;
                    s = fma(v, v, s);
                }
                const double vn = block_sum(s, Lpart, tid); // its barriers publish Lvk
                t = 2.0 / vn;
                if (tid == 0) { Ltau[k] = t; Le[k] = alpha; }
    #pragma unroll
                for (int c = 0; c < NCT; c++) cc[c] = 0.0;
                // rows in batches of UB, two batches in flight: the loads of the next batch are issued before the current one is used
                auto matvec_rows = [&](int j0, const double (&a)[UB][NCT]) __attribute__((always_inline)) {
    #pragma unroll
                    for (int u = 0; u < UB; u++) {
                        if (j0 + u < n) {
                            const double vj = Lvk[j0 + u];
    #pragma unroll
                            for (int c = 0; c < NCT; c++) cc[c] = fma(a[u][c], vj, cc[c]);
                        }
                    }
                };
                double a0[UB][NCT], a1[UB][NCT];
                load_rows(k + 1, k, a0);
                for (int j0 = k + 1; j0 < n; j0 += 2 * UB) {
                    load_rows(j0 + UB, k, a1);
                    matvec_rows(j0, a0);
                    load_rows(j0 + 2 * UB, k, a0);
                    matvec_rows(j0 + UB, a1);
                }
            }
            // ---- w_k = t A v - (t/2 (t A v . v)) v
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                if (!(i < n && i > k)) cc[c] = 0.0;
            }
            double s = 0.0;
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                cc[c] = t * cc[c];
                s = fma(cc[c], vi[c], s);
            }
            const double Kc = (0.5 * t) * block_sum(s, Lpart, tid);
            double wi[NCT];
    #pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int i = tid + 256 * c;
                wi[c] = fma(-Kc, vi[c], cc[c]);
                if (i < n) Lw[i] = wi[c];
            }
            __syncthreads();
            // the rank-2 update of one element, canonical (row >= column) operand order: the matrix stays exactly symmetric
            auto upd = [&](int r, int c, double a) __attribute__((always_inline)) {
                const int i = tid + 256 * c;
                const double vr = Lvk[r], wr = Lw[r];
                const bool rc = r >= i;
                const double va = rc ? vr : vi[c], wa = rc ? wr : wi[c], vb = rc ? vi[c] : vr, wb = rc ? wi[c] : wr;
                return fma(-wa, vb, fma(-va, wb, a));
            };
            // ---- row k + 1 first: it carries the next reflector
            bool next = false;
            double t2 = 0.0, vi2[NCT], cc2[NCT];
    #pragma unroll
            for (int c = 0; c < NCT; c++) { vi2[c] = 0.0; cc2[c] = 0.0; }
            {
                const int r = k + 1;
                double rowv[NCT];
                double s2 = 0.0;
    #pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    rowv[c] = 0.0;
                    if (i < n && i > k) {
                        rowv[c] = upd(r, c, A[(long)r * n + i]);
                        A[(long)r * n + i] = rowv[c];
                        if (i > r) s2 = fma(rowv[c], rowv[c], s2);
                        if (i == r + 1) Lscal[4] = rowv[c];
                    }
                }
                if (k + 1 < n - 2) { // there is a step k + 1
                    const double sigma2 = block_sum(s2, Lpart, tid); // its barriers publish Lscal[4]
                    if (sigma2 > LRF_SIGMA_TINY) {
                        next = true;
                        const double x0 = Lscal[4];
                        const double nrm = sqrt(sigma2);
                        const double alpha = (x0 >= 0.0) ? -nrm : nrm;
                        double s3 = 0.0;
    #pragma unroll
                        for (int c = 0; c < NCT; c++) {
                            const int i = tid + 256 * c;
                            double v = 0.0;
                            if (i < n) {
                                if (i > r + 1) v = rowv[c];
                                else if (i == r + 1) v = x0 - alpha;
                                Lvn[i] = v;
                                if (i > r) A[(long)r * n + i] = v;
                            }
                            vi2[c] = v;
                            s3 = fma(v, v, s3);
                        }
                        const double vn = block_sum(s3, Lpart, tid); // publishes Lvn
                        t2 = 2.0 / vn;
                        if (tid == 0) { Ltau[r] = t2; Le[r] = alpha; }
                    }
                }
            }
            // ---- the remaining rows: update, and (when there is a next reflector) its matrix-vector product on the fly
            auto update_rows = [&](int r0, const double (&a)[UB][NCT]) __attribute__((always_inline)) {
    #pragma unroll
                for (int u = 0; u < UB; u++) {
                    const int r = r0 + u;
                    if (r < n) {
                        double* Ar = A + (long)r * n;
                        const double vn_r = next ? Lvn[r] : 0.0;
    #pragma unroll
                        for (int c = 0; c < NCT; c++) {
                            const int i = tid + 256 * c;
                            if (i < n && i > k) {
                                const double nv = upd(r, c, a[u][c]);
                                Ar[i] = nv;
                                cc2[c] = fma(nv, vn_r, cc2[c]);
                            }
                        }
                    }
                }
            };
            { // the loads of the next batch go out before the stores of this one, so a batch waits for one latency, not two
                double a0[UB][NCT], a1[UB][NCT];
                load_rows(k + 2, k, a0);
                for (int r0 = k + 2; r0 < n; r0 += 2 * UB) {
                    load_rows(r0 + UB, k, a1);
                    update_rows(r0, a0);
                    load_rows(r0 + 2 * UB, k, a0);
                    update_rows(r0 + UB, a1);
                }
            }
            __syncthreads();
            have = next;
            if (next) {
                t = t2;
    #pragma unroll
                for (int c = 0; c < NCT; c++) { vi[c] = vi2[c]; cc[c] = cc2[c]; }
            }
        }
    }
    if (!td_in) {
#pragma unroll
        for (int c = 0; c < NCT; c++) {
            const int i = tid + 256 * c;
            if (i < n) Ld[i] = A[(long)i * n + i];
        }
        if (tid == 0) {
            if (n >= 2) { Le[n - 2] = A[(long)(n - 1) * n + n - 2]; Ltau[n - 2] = 0.0; }
            Le[n - 1] = 0.0;
            Ltau[n - 1] = 0.0;
        }
    }
    __syncthreads();

    EIG_STAMP();
    if (stop_after == 1) return;
    // ---- Gershgorin hull, pivmin
    {
        double a = 1e300, b = -1e300, m2 = 0.0;
#pragma unroll
        for (int c = 0; c < NCT; c++) {
            const int i = tid + 256 * c;
            if (i < n) {
                const double ei = (i < n - 1) ? Le[i] : 0.0, eim = (i > 0) ? Le[i - 1] : 0.0;
                Le2[i] = ei * ei;
                const double rad = fabs(eim) + fabs(ei);
                a = fmin(a, Ld[i] - rad);
                b = fmax(b, Ld[i] + rad);
                m2 = fmax(m2, ei * ei);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a = fmin(a, __shfl_xor(a, off, 64));
            b = fmax(b, __shfl_xor(b, off, 64));
            m2 = fmax(m2, __shfl_xor(m2, off, 64));
        }
        if (lane == 0) { Lpart[wave] = a; Lpart[4 + wave] = b; Lpart[8 + wave] = m2; }
        __syncthreads();
        if (tid == 0) {
            const double lo = fmin(fmin(Lpart[0], Lpart[1]), fmin(Lpart[2], Lpart[3]));
            const double hi = fmax(fmax(Lpart[4], Lpart[5]), fmax(Lpart[6], Lpart[7]));
            const double e2m = fmax(fmax(Lpart[8], Lpart[9]), fmax(Lpart[10], Lpart[11]));
            const double tn = fabs(lo) > fabs(hi) ? fabs(lo) : fabs(hi);
            const double pivmin = 2.2250738585072014e-300 * (e2m > 1.0 ? e2m : 1.0);
            const double slack = 2.0 * tn * 2.220446049250313e-16 * n + 2.0 * pivmin;
            const double lo2 = lo - slack, hi2 = hi + slack;
            const int sc = __builtin_amdgcn_frexp_exp(fabs(lo2) > fabs(hi2) ? fabs(lo2) : fabs(hi2)); // hull within [-2^sc, 2^sc]
            Lscal[1] = pivmin; Lscal[2] = ldexp(lo2, -sc); Lscal[3] = ldexp(hi2, -sc); Lscal[4] = (double)sc;
        }
        __syncthreads();
    }
    const double pivmin = Lscal[1];
    // ---- eigenvalues: one wave per eigenvalue, 64 shifts per pass, 10 passes (interval / 65^10).  Sturm counts in the
    // division-free form of k_init (lrf_kernels.hip; oracle: sturm_count): the leading principal minors of T / 2^sc - x,
    // one dependent fma per step, the pair rescaled by a power of two every eighth step, the sign changes counted from the
    // collected sign bits; a pass in which a minor came out as zero is redone with the replacement rule.
    double2* de = reinterpret_cast<double2*>(Lv); // (d'_j, e'_{j-1}^2): Lv and Lw are free between the stages
    {
        const int sc = (int)Lscal[4];
        for (int j = tid; j < n; j += 256) {
            const double es = (j > 0) ? ldexp(Le[j - 1], -sc) : 0.0;
            de[j] = make_double2(ldexp(Ld[j], -sc), es * es);
        }
        __syncthreads();
        const int n8 = n & ~7;
        // A wave runs TWO searches at once (eigenvalues r and r + 4, as k_init does): the two dependent chains and their table
        // reads interleave, and a call with 5..8 eigenvalues needs one round instead of two (the stage issues ~7.5 vector
        // instructions per Sturm step and wave, so two interleaved searches cost what two rounds did: 70 us at R = 5, n = 192).
        // A wave with one eigenvalue left runs it twice (same bits, result stored once).
        auto sturm_slow = [&](double x) {
            double p = 1.0, pp = 0.0;
            int cnt = 0;
            for (int j = 0; j < n; j++) {
                const double2 q = de[j];
                double pn = fma(q.x - x, p, -(q.y * pp));
                if (pn == 0.0) pn = (__double2hiint(p) < 0) ? 0x1p-200 : -0x1p-200;
                cnt += ((__double2hiint(pn) ^ __double2hiint(p)) < 0);
                pp = p;
                p = pn;
                if ((j & 7) == 7) {
                    const int ea = __builtin_amdgcn_frexp_exp(p), eb = __builtin_amdgcn_frexp_exp(pp);
                    const int m = ea > eb ? ea : eb;
                    p = ldexp(p, -m);
                    pp = ldexp(pp, -m);
                }
            }
            return cnt;
        };
        for (int r0 = wave; r0 < Rc; r0 += 8) {
            const int r1 = r0 + 4 < Rc ? r0 + 4 : r0;
            const int kk[2] = {n - 1 - r0, n - 1 - r1};
            double a[2] = {Lscal[2], Lscal[2]}, b[2] = {Lscal[3], Lscal[3]};
            for (int pass = 0; pass < 10; pass++) {
                double x[2], p[2] = {1.0, 1.0}, pp[2] = {0.0, 0.0};
                unsigned sg[2] = {0u, 0u};
                int cnt[2] = {0, 0};
                double zm[2] = {1.0, 1.0}; // the smallest |minor| of the pass: a zero is looked for once, at its end (one v_min_f64 per
                                           // step instead of a compare, a select and an OR)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const double h = (b[s2] - a[s2]) / 65.0;
                    x[s2] = a[s2] + h * (double)(lane + 1);
                }
#pragma unroll 1
                for (int j0 = 0; j0 < n8; j0 += 8) { // not unrolled further: the table reads must stay inside the pass loop
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const double2 q = de[j0 + u];
#pragma unroll
                        for (int s2 = 0; s2 < 2; s2++) {
                            const double pn = fma(q.x - x[s2], p[s2], -(q.y * pp[s2]));
                            zm[s2] = fmin(zm[s2], fabs(pn));
                            sg[s2] = __builtin_amdgcn_alignbit(sg[s2], (unsigned)__double2hiint(pn), 31);
                            pp[s2] = p[s2];
                            p[s2] = pn;
                        }
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        cnt[s2] += __popc((sg[s2] ^ (sg[s2] >> 1)) & 0xffu);
                        const int ea = __builtin_amdgcn_frexp_exp(p[s2]), eb = __builtin_amdgcn_frexp_exp(pp[s2]);
                        const int m = ea > eb ? ea : eb;
                        p[s2] = ldexp(p[s2], -m);
                        pp[s2] = ldexp(pp[s2], -m);
                    }
                }
                for (int j = n8; j < n; j++) {
                    const double2 q = de[j];
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        const double pn = fma(q.x - x[s2], p[s2], -(q.y * pp[s2]));
                        zm[s2] = fmin(zm[s2], fabs(pn));
                        cnt[s2] += ((__double2hiint(pn) ^ __double2hiint(p[s2])) < 0);
                        pp[s2] = p[s2];
                        p[s2] = pn;
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    if (__any(zm[s2] == 0.0)) cnt[s2] = sturm_slow(x[s2]); // wave-uniform
                    const unsigned long long mask = __ballot(cnt[s2] > kk[s2]);
                    const int jj = mask ? (int)__builtin_ctzll(mask) : 64;
                    const double xm = __shfl(x[s2], jj > 0 ? jj - 1 : 0, 64), xj = __shfl(x[s2], jj < 64 ? jj : 63, 64);
                    const double na = (jj == 0) ? a[s2] : xm, nb = (jj == 64) ? b[s2] : xj;
                    a[s2] = na;
                    b[s2] = nb;
                }
            }
            if (lane == 0) {
                Llam[r0] = ldexp(0.5 * (a[0] + b[0]), sc);
                if (r1 != r0) Llam[r1] = ldexp(0.5 * (a[1] + b[1]), sc);
            }
        }
    }
    __syncthreads();
    EIG_STAMP();
    if (stop_after == 2) return;
    if (stop_after == 9) { // developer aid (tools/dev_any_init_bits.py): d, e, lambda as raw doubles in the E1 output
        double* dbg = reinterpret_cast<double*>(E1 + (long)blockIdx.x * n * R);
        if ((long)n * R * 4 >= (long)(2 * n + Rc) * 8) {
            for (int i = tid; i < n; i += 256) { dbg[i] = Ld[i]; dbg[n + i] = Le[i]; }
            for (int r = tid; r < Rc; r += 256) dbg[2 * n + r] = Llam[r];
        }
        return;
    }
    // ---- twisted factorisation, TWO threads per eigenvalue (t = 2 r + dir): the two pivot sequences are independent chains
    // of one division per step, and so are the two halves of the vector on either side of the twist index — one lane each
    // instead of one after the other.  The two lanes of a pair sit in one wave, so the loops are written ONCE with the
    // direction as data (a branch per direction would run the two one after the other); the arithmetic per element is
    // unchanged.  The twist index is searched by both lanes, half of the range each.
    for (int t = tid; t < 2 * Rc; t += 256) { // (t and t ^ 1 are lanes of one wave: 256 is even)
        const int r = t >> 1, dir = t & 1;
        const double lam = Llam[r];
        double* Dsel = dir ? Dm : Dp; // D+ runs forward from row 0, D- backward from row n - 1
        {
            const int j0 = dir ? n - 1 : 0;
            double q = Ld[j0] - lam;
            Dsel[(long)j0 * Rc + r] = q;
            for (int i = 1; i < n; i++) {
                const int j = dir ? n - 1 - i : i;
                const double e2 = Le2[dir ? j : j - 1];
                if (fabs(q) < pivmin) q = -pivmin;
                q = (Ld[j] - lam) - e2 / q;
                Dsel[(long)j * Rc + r] = q;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the other lane of the pair reads this lane's sequence below
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the first index of the smallest |(D+ + D-) - (d - lambda)|: the sequential scan's rule (j == 0 || g < best), the
        // second half scanned by the other lane from best = +inf and merged with the same strict comparison
        const int half = n >> 1;
        int kt = dir ? -1 : 0;
        double best = dir ? __builtin_inf() : 0.0;
        const int jb = dir ? half : 0, je = dir ? n : half;
        for (int j = jb; j < je; j++) {
            const double g = fabs((Dp[(long)j * Rc + r] + Dm[(long)j * Rc + r]) - (Ld[j] - lam));
            if ((!dir && j == 0) || g < best) { best = g; kt = j; }
        }
        {
            const double ob = __shfl_xor(best, 1, 64);
            const int ok = __shfl_xor(kt, 1, 64);
            const double b0 = dir ? ob : best, b1 = dir ? best : ob; // (first half, second half)
            const int k0 = dir ? ok : kt, k1 = dir ? kt : ok;
            kt = (half > 0 && b1 < b0) ? k1 : k0;
        }
        double* x = Z + (long)r * n;
        double xv = 1.0;
        if (dir == 0) x[kt] = 1.0;
        const int cnt = dir ? n - 1 - kt : kt; // below the twist: j = kt - 1 .. 0 with D+; above: j = kt .. n - 2 with D-[j + 1]
        for (int i = 0; i < cnt; i++) {
            const int j = dir ? kt + i : kt - 1 - i;
            double qq = Dsel[(long)(dir ? j + 1 : j) * Rc + r];
            if (fabs(qq) < pivmin) qq = -pivmin;
            xv = -(Le[j] / qq) * xv;
            x[dir ? j + 1 : j] = xv;
        }
    }
    __syncthreads();
    EIG_STAMP();
    if (stop_after == 3) return;
    if (stop_after >= 1300 && stop_after < 1400) { // developer aid: vectors r0.. of Z as raw doubles in the E1 output
        double* dbg = reinterpret_cast<double*>(E1 + (long)blockIdx.x * n * R);
        const int r0 = stop_after - 1300;
        const long cap = (long)n * R / 2;
        for (long q = tid; q < cap; q += 256) {
            const long r = r0 + q / n;
            dbg[q] = (r < Rc) ? Z[r * n + q % n] : 0.0;
        }
        return;
    }

    // ---- orthonormalisation: classical Gram-Schmidt, twice, against the vectors already fixed
    for (int r = 0; r < Rc; r++) {
        double* Zr = Z + (long)r * n;
        double x[NCT];
        bool fin = true;
#pragma unroll
        for (int c = 0; c < NCT; c++) {
            const int i = tid + 256 * c;
            x[c] = (i < n) ? Zr[i] : 0.0;
            fin = fin && isfinite(x[c]);
        }
        bool use_twisted = __syncthreads_and(fin) != 0;
        int uidx = 0;
        for (;;) {
            if (use_twisted) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < NCT; c++) s = fma(x[c], x[c], s);
                const double n0 = sqrt(block_sum(s, Lpart, tid));
#pragma unroll
                for (int c = 0; c < NCT; c++) x[c] = x[c] / n0;
            } else {
                if (uidx >= n) break;
#pragma unroll
                for (int c = 0; c < NCT; c++) x[c] = (tid + 256 * c == uidx) ? 1.0 : 0.0;
                uidx++;
            }
            for (int pass = 0; pass < 2 && r > 0; pass++) {
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const int i = tid + 256 * c;
                    if (i < n) Lv[i] = x[c];
                }
                __syncthreads();
                // the earlier vectors live in global memory (L2): their loads are issued in batches (all 4 NCT pieces of a
                // dot product at once, eight vectors per step of the update) instead of one per loop trip; same chains
                // (DB earlier vectors per trip and wave, their loads in flight together and their reduction trees interleaved: one at
                // a time a trip waited for its own L2 round trip — the stage was 4.1 of k_any_eig<2>'s 9.2 ms at rank 102, side
                // 512; each dot product's arithmetic is what it was)
                constexpr int DB = NCT <= 2 ? 4 : 2;
                for (int pr0 = wave; pr0 < r; pr0 += 4 * DB) {
                    double zv[DB][4 * NCT];
#pragma unroll
                    for (int d = 0; d < DB; d++) {
                        const int pr = pr0 + 4 * d;
                        const double* Zp = Z + (long)(pr < r ? pr : r - 1) * n;
#pragma unroll
                        for (int e = 0; e < 4 * NCT; e++) {
                            const int i = lane + 64 * e;
                            zv[d][e] = Zp[i < n ? i : n - 1];
                        }
                    }
                    double dsum[DB];
#pragma unroll
                    for (int d = 0; d < DB; d++) {
                        dsum[d] = 0.0;
#pragma unroll
                        for (int e = 0; e < 4 * NCT; e++) {
                            const int i = lane + 64 * e;
                            if (i < n) dsum[d] = fma(zv[d][e], Lv[i], dsum[d]);
                        }
                    }
#pragma unroll
                    for (int d = 0; d < DB; d++) dsum[d] = wave_sum(dsum[d]);
#pragma unroll
                    for (int d = 0; d < DB; d++)
                        if (lane == 0 && pr0 + 4 * d < r) Lw[pr0 + 4 * d] = dsum[d];
                }
                __syncthreads();
                // (two batches of eight earlier vectors in flight: the loads of batch b + 1 are issued before batch b's fmas —
                // a batch at a time waited out one L2 round trip per eight vectors)
                auto load_batch = [&](int p0, double (&zv)[8][NCT]) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const double* Zp = Z + (long)(p0 + u < r ? p0 + u : r - 1) * n;
#pragma unroll
                        for (int c = 0; c < NCT; c++) {
                            const int i = tid + 256 * c;
                            const double zz = Zp[i < n ? i : n - 1];
                            // entries past the side must stay zero: they are part of the norm below (until round 3 the clamped
                            // load went into x there: harmless while the coefficients are ~1e-16, but the vectors of a cluster
                            // — rank-deficient matrices — came out with norms below one; found by the oracle's restatement)
                            zv[u][c] = (i < n) ? zz : 0.0;
                        }
                    }
                };
                auto apply_batch = [&](int p0, const double (&zv)[8][NCT]) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (p0 + u < r) {
                            const double cf = Lw[p0 + u];
#pragma unroll
                            for (int c = 0; c < NCT; c++) x[c] = fma(-cf, zv[u][c], x[c]);
                        }
                    }
                };
                if constexpr (NCT <= 2) {
                    double za[8][NCT], zb[8][NCT];
                    load_batch(0, za);
                    for (int p0 = 0; p0 < r; p0 += 16) {
                        if (p0 + 8 < r) load_batch(p0 + 8, zb);
                        apply_batch(p0, za);
                        if (p0 + 16 < r) load_batch(p0 + 16, za);
                        if (p0 + 8 < r) apply_batch(p0 + 8, zb);
                    }
                } else { // (sides above 512: the second batch does not fit the registers)
                    for (int p0 = 0; p0 < r; p0 += 8) {
                        double za[8][NCT];
                        load_batch(p0, za);
                        apply_batch(p0, za);
                    }
                }
                __syncthreads();
            }
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < NCT; c++) s = fma(x[c], x[c], s);
            const double n2 = block_sum(s, Lpart, tid);
            if (n2 > 1e-6 && n2 < 1e300) {
                const double nr = sqrt(n2);
#pragma unroll
                for (int c = 0; c < NCT; c++) x[c] = x[c] / nr;
                break;
            }
            use_twisted = false;
        }
#pragma unroll
        for (int c = 0; c < NCT; c++) {
            const int i = tid + 256 * c;
            if (i < n) Zr[i] = x[c];
        }
        __syncthreads();
    }
    if (stop_after >= 1400 && stop_after < 1500) { // developer aid: vectors r0.. of Z as raw doubles in the E1 output
        double* dbg = reinterpret_cast<double*>(E1 + (long)blockIdx.x * n * R);
        const int r0 = stop_after - 1400;
        const long cap = (long)n * R / 2;
        for (long q = tid; q < cap; q += 256) {
            const long r = r0 + q / n;
            dbg[q] = (r < Rc) ? Z[r * n + q % n] : 0.0;
        }
        return;
    }
    EIG_STAMP();
    if (stop_after == 4) return;
    // ---- back-transformation x <- H_0 H_1 ... H_{n-3} x, up to CW vectors per wave at a time, lane owns i = lane + 64 e.  The
    // vectors are dealt out over the four waves (slot w of wave g: vector base + g + 4 w): with consecutive vectors per wave
    // a call of rank <= CW ran on one wave (svd_encode, R = 5: 173 us of the kernel's 370)
    constexpr int NE = 4 * NCT, CW = 32 / NE;
    float* E1b = E1 + (long)blockIdx.x * n * R;
    float* E2b = E2 + (long)blockIdx.x * n * R;
    for (int g0 = wave; g0 < Rc; g0 += 4 * CW) { // vector of slot w: g0 + 4 w
        double x[CW][NE];
#pragma unroll
        for (int w = 0; w < CW; w++)
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int i = lane + 64 * e;
                x[w][e] = (g0 + 4 * w < Rc && i < n) ? Z[(long)(g0 + 4 * w) * n + i] : 0.0;
            }
        // the reflectors of the next PF steps are in flight (global memory) while step k is computed; the dot products reduce
        // through the DPP / permlane tree of k_init (wave_tree64), not the LDS crossbar.  What a step costs is that tree: ~50
        // dependent instructions per vector and step, ~1000 cycles (ablations, 256 matrices of 192 x 192: synthetic reflectors
        // instead of the loads change nothing, a plain sum instead of the tree takes a third off)
        constexpr int PF = 4;
        double vq[PF][NE];
        auto load_v = [&](int k, double (&v)[NE]) __attribute__((always_inline)) {
            const double* Ak = A + (long)(k > 0 ? k : 0) * n;
#pragma unroll
            for (int e = 0; e < NE; e++) { // unconditional loads from clamped addresses, then a select: a load under a condition
                const int i = lane + 64 * e; // becomes a branch around it, sixteen of them per step
                const double a = Ak[i < n ? i : n - 1];
                v[e] = (k >= 0 && i < n && i > k) ? a : 0.0;
            }
        };
#pragma unroll
        for (int pf = 0; pf < PF; pf++) load_v(n - 3 - pf, vq[pf]);
        for (int k0 = n - 3; k0 >= 0; k0 -= PF) {
#pragma unroll
            for (int pf = 0; pf < PF; pf++) {
                const int k = k0 - pf;
                double v[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) v[e] = vq[pf][e];
                load_v(k - PF, vq[pf]);
                const double tk = k >= 0 ? Ltau[k] : 0.0;
                if (tk != 0.0) {
#pragma unroll
                    for (int w = 0; w < CW; w++) {
                        if (g0 + 4 * w >= Rc) continue; // wave-uniform
                        double dsum = 0.0;
#pragma unroll
                        for (int e = 0; e < NE; e++) dsum = fma(v[e], x[w][e], dsum);
                        const double sc = tk * wave_tree64(dsum);
#pragma unroll
                        for (int e = 0; e < NE; e++) x[w][e] = fma(-sc, v[e], x[w][e]);
                    }
                }
            }
        }
#ifdef LRF_REG_STAMPS
        if (esi == 5) EIG_STAMP();
#endif
#pragma unroll
        for (int w = 0; w < CW; w++) {
            const int r = g0 + 4 * w;
            if (r >= Rc) continue;
            double dsum = 0.0;
#pragma unroll
            for (int e = 0; e < NE; e++) dsum = fma((double)(lane + 64 * e + 1), x[w][e], dsum);
            const double dot = wave_sum(dsum);
            const double lam = Llam[r];
            const double sr = sqrt(sqrt(lam > 1e-200 ? lam : 0.0));
            const int sg = sign ? (int)sign[(long)blockIdx.x * R + r] : 0;
            const double want = sg ? (double)sg : -1.0;
            const double flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.0 : -1.0;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int i = lane + 64 * e;
                if (i < n) {
                    const double ev = flip * x[w][e];
                    E1b[(long)i * R + r] = (float)(ev * sr);
                    E2b[(long)i * R + r] = (sr > 0.0) ? (float)(ev / sr) : 0.f;
                }
            }
        }
    }
#ifdef LRF_REG_STAMPS
    while (esi < 7) EIG_STAMP();
    EIG_STAMPS_OUT();
#endif
    // columns r >= min(M, N): zero (qmf.py:50-52)
    for (long e = tid; e < (long)n * (R - Rc); e += 256) {
        const long i = e / (R - Rc), r = Rc + (e - i * (R - Rc));
        E1b[i * R + r] = 0.f;
        E2b[i * R + r] = 0.f;
    }
}

// Column signs when the eigenvectors belong to the M side (M < N): the convention of include/lrf_hip.h speaks of v0, so the
// sign of sum_j (j+1) v0[j,r] is evaluated on the finished factor and both columns are flipped where it differs from the
// wanted one (sign[r], or -1).  One workgroup per matrix.
__global__ __launch_bounds__(256) void k_any_signfix(float* __restrict__ Uf, float* __restrict__ Vf, int M, int N, int R,
                                                     const int8_t* __restrict__ sign)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* flip = reinterpret_cast<float*>(smem); // [R]
    float* Ub = Uf + (long)blockIdx.x * M * R;
    float* Vb = Vf + (long)blockIdx.x * N * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < R; r += 4) {
        double acc = 0.0;
        for (int j = lane; j < N; j += 64) acc = fma((double)(j + 1), (double)Vb[(long)j * R + r], acc);
        acc = wave_sum(acc);
        const int sg = sign ? (int)sign[(long)blockIdx.x * R + r] : 0;
        const double want = sg ? (double)sg : -1.0;
        if (lane == 0) flip[r] = ((acc < 0.0 ? -1.0 : 1.0) == want) ? 1.f : -1.f;
    }
    __syncthreads();
    for (long e = tid; e < (long)M * R; e += 256) Ub[e] = flip[e % R] * Ub[e];
    for (long e = tid; e < (long)N * R; e += 256) Vb[e] = flip[e % R] * Vb[e];
}

// ---- colour planes for any patch size, and the matching decode ---------------------------------------------------
// Plane c of B images as matrices: patches of p x q (reflect-padded plane, patchify "c (h p) (w q) -> (h w) (c p q)",
// qmf.py:43-56) or, with p = 0, the plane itself [h, w].  One thread per matrix element; arithmetic as k_planes.
__global__ __launch_bounds__(256) void k_any_planes(const uint8_t* __restrict__ rgb, int H, int W, int c, int h, int w, int p, int q,
                                                    int top, int left, int nw, long mat_elems, float* __restrict__ X)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= mat_elems) return;
    int y, x;
    if (p > 0) {
        const int N = p * q;
        const long m = e / N;
        const int col = (int)(e - m * N);
        const int ph = (int)(m / nw), pw = (int)(m - (long)ph * nw);
        y = reflect_idx(ph * p + col / q - top, h);
        x = reflect_idx(pw * q + col % q - left, w);
    } else {
        y = (int)(e / w);
        x = (int)(e - (long)y * w);
    }
    const uint8_t* img = rgb + (long)blockIdx.y * 3 * H * W;
    const int hw = H * W;
    float out;
    if (c == 0) {
        const uint8_t* p0 = img + (long)y * W + x;
        out = ycc_of((float)p0[0], (float)p0[hw], (float)p0[2 * hw], 0);
    } else { // adaptive average pooling window, row-major fp32 sum, then / kh / kw
        const int h0 = (int)(((long)y * H) / h), h1 = (int)((((long)y + 1) * H + h - 1) / h);
        const int w0 = (int)(((long)x * W) / w), w1 = (int)((((long)x + 1) * W + w - 1) / w);
        float sum = 0.f;
        for (int yy = h0; yy < h1; yy++)
            for (int xx = w0; xx < w1; xx++) {
                const uint8_t* p0 = img + (long)yy * W + xx;
                sum = sum + ycc_of((float)p0[0], (float)p0[hw], (float)p0[2 * hw], c);
            }
        out = sum / (float)(h1 - h0) / (float)(w1 - w0);
    }
    X[(long)blockIdx.y * mat_elems + e] = out;
}

struct AnyDecodePlane {
    const int8_t* U;
    const int8_t* V;
    long u_img, v_img; // elements per image
    int h, w, p, q, top, left, nw, R;
};

__device__ __forceinline__ float any_recon(const AnyDecodePlane& d, long img, int y, int x)
{
    long m;
    int col;
    if (d.p > 0) {
        const int yy = y + d.top, xx = x + d.left;
        m = (long)(yy / d.p) * d.nw + xx / d.q;
        col = (yy % d.p) * d.q + xx % d.q;
    } else {
        m = y;
        col = x;
    }
    const int8_t* u = d.U + img * d.u_img + m * d.R;
    const int8_t* v = d.V + img * d.v_img + (long)col * d.R;
    float acc = 0.f; // u @ v.mT: exact small integers, any order
    for (int r = 0; r < d.R; r++) acc = fmaf((float)u[r], (float)v[r], acc);
    return acc;
}

// qmf_decode of the YCbCr branch for any patch size / no patches (qmf.py:325-351): one thread per pixel
__global__ __launch_bounds__(256) void k_any_decode(AnyDecodePlane d0, AnyDecodePlane d1, AnyDecodePlane d2, int H, int W,
                                                    uint8_t* __restrict__ rgb)
{
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long)H * W) return;
    const int y = (int)(o / W), x = (int)(o - (long)y * W);
    const float T[3][3] = {{1.0f, 0.0f, 1.402f}, {1.0f, -0.344136f, -0.714136f}, {1.0f, 1.772f, 0.0f}};
    const float sh = (float)d1.h / (float)H, sw = (float)d1.w / (float)W;
    int sy = (int)floorf((float)y * sh), sx = (int)floorf((float)x * sw);
    if (sy > d1.h - 1) sy = d1.h - 1;
    if (sx > d1.w - 1) sx = d1.w - 1;
    float cv[3];
    cv[0] = any_recon(d0, blockIdx.y, y, x) + 0.f;
    cv[1] = any_recon(d1, blockIdx.y, sy, sx) + -128.f;
    cv[2] = any_recon(d2, blockIdx.y, sy, sx) + -128.f;
    uint8_t* out = rgb + (long)blockIdx.y * 3 * H * W;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float acc = 0.f;
        acc = fmaf(T[ch][0], cv[0], acc);
        acc = fmaf(T[ch][1], cv[1], acc);
        acc = fmaf(T[ch][2], cv[2], acc);
        acc = fminf(fmaxf(acc, 0.f), 255.f);
        out[(long)ch * H * W + o] = (uint8_t)acc; // truncation (to_dtype)
    }
}

// ---- QMF.loss (lrf/factorization/qmf.py:225-227; relative_error, lrf/factorization/utils.py:12-15): per matrix the squared
// norms of x - (w0 + w1 u v^T) and of x, in fp64 (k_any_loss, one workgroup per 64-row tile: partial sums added by atomics),
// then sqrt(num) / (sqrt(den) + 1e-16) (k_any_loss_finish).  A diagnostic (QMF(verbose=True)), not part of the encoder.
__global__ __launch_bounds__(256) void k_any_loss(const float* __restrict__ X, const float* __restrict__ U, const float* __restrict__ V,
                                                  const float* __restrict__ Wp, int M, int N, int R, double* __restrict__ acc)
{
    const int b = blockIdx.y, row0 = blockIdx.x * 64;
    const float* x = X + (long)b * M * N;
    const float* u = U + (long)b * M * R;
    const float* v = V + (long)b * N * R;
    const float w0 = Wp ? Wp[2 * b] : 0.f, w1 = Wp ? Wp[2 * b + 1] : 1.f;
    __shared__ double part[4];
    double num = 0.0, den = 0.0;
    const int rows = M - row0 < 64 ? M - row0 : 64;
    for (long e = threadIdx.x; e < (long)rows * N; e += 256) {
        const int m = row0 + (int)(e / N), n = (int)(e % N);
        float y = 0.f; // u @ v.mT: a k-ordered fma chain
        for (int r = 0; r < R; r++) y = fmaf(u[(long)m * R + r], v[(long)n * R + r], y);
        y = Wp ? w0 + w1 * y : y;
        const float xv = x[(long)m * N + n];
        const double d = (double)(xv - y);
        num += d * d;
        den += (double)xv * (double)xv;
    }
    num = block_sum(num, part, threadIdx.x);
    den = block_sum(den, part, threadIdx.x);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[2 * b], num);
        atomicAdd(&acc[2 * b + 1], den);
    }
}
__global__ void k_any_loss_finish(const double* __restrict__ acc, int B, float* __restrict__ loss)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) loss[b] = (float)(sqrt(acc[2 * b]) / (sqrt(acc[2 * b + 1]) + 1e-16));
}
