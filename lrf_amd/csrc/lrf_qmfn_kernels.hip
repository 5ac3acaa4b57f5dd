// lrf_qmfn_kernels.hip — the BCD iteration of the RGB colour-space branch of qmf_encode (lrf/compression/qmf.py:164-187):
// one matrix X [M, 192] per image (three 8x8 colour patches per row), rank R = round(192 q / 100) <= 32.
//
// Same arithmetic and summation orders as the 64-column kernels (reference lrf/factorization/qmf.py:93-139; oracle
// lrf_oracle_bcd with N = 192 reproduces the reference bit for bit, tools/gen_golden.py `rgbspace`): k-ordered fma
// chains for x @ v (K = 192, one MKL block) and x.mT @ u (one chain per 384-row block, block partials added in order
// by k_vupdaten), exact integers for u.mT @ u, the generic Gauss-Seidel of lrf_bigrank_kernels.hip.
// Correctness-first, not tuned: plain VALU fma chains with LDS operands, one workgroup per CU.
// Included by lrf_api.hip after lrf_kernels.hip and lrf_bigrank_kernels.hip.

#define LRF_RPN 32                    // padded rank
#define LRF_GTN_LD 36                 // gt table pitch: <= 31 `bb` entries, [33] = den
#define LRF_GTN_DEN 33
#define LRF_GTN_STRIDE (LRF_RPN * LRF_GTN_LD)

// One row, all R columns (qmf.py:108-119), u_row updated in place; gt with pitch LRF_GTN_LD (layout as in gs_row);
// the dot product in the reference's order is gs_term2_generic (lrf_bigrank_kernels.hip).
__device__ __forceinline__ void gs_row_n(int R, const float* a_row, float* u_row, const float* gt, bool native, float lo, float hi)
{
    const int K = R - 1;
    for (int r = 0; r < R; r++) {
        const float* bb = gt + r * LRF_GTN_LD;
        float term2 = gs_term2_generic(u_row, r, bb, K, native);
        float num = (a_row[r] - term2) + LRF_EPS;
        float val = rintf(num / bb[LRF_GTN_DEN]);
        u_row[r] = fminf(fmaxf(val, lo), hi);
    }
}

// gt table of b = v.mT @ v from a [depth][LRF_RPN] factor; ATen's native kernel when depth*R*R < 400
__device__ __forceinline__ void make_gtable_n(const float* Vp, int depth, int R, float* gt, int tid, int nthreads)
{
    bool native = (long)depth * R * R < 400;
    for (int i = tid; i < R * R; i += nthreads) {
        int j = i / R, r = i - j * R;
        float acc = 0.f;
        if (native) {
            for (int k = 0; k < depth; k++) {
                float p = Vp[k * LRF_RPN + j] * Vp[k * LRF_RPN + r];
                acc = acc + p;
            }
        } else {
            for (int k = 0; k < depth; k++) acc = fmaf(Vp[k * LRF_RPN + j], Vp[k * LRF_RPN + r], acc);
        }
        if (j == r) gt[r * LRF_GTN_LD + LRF_GTN_DEN] = (acc + 0.f) + LRF_EPS;
        else gt[r * LRF_GTN_LD + (j < r ? j : j - 1)] = acc;
    }
}

// V0 [plane][NN][R] fp32 (k_any_eig output) -> padded Vf [plane][NN][LRF_RPN], and the first b table
template <int NN>
__global__ __launch_bounds__(256) void k_bprepn(const PlaneDesc* __restrict__ planes, const float* __restrict__ V0,
                                                float* __restrict__ Vf, float* __restrict__ Bf)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* v_s = reinterpret_cast<float*>(smem); // [NN][LRF_RPN]
    const PlaneDesc pd = planes[blockIdx.x];
    for (int i = threadIdx.x; i < NN * LRF_RPN; i += 256) {
        int j = i / LRF_RPN, r = i - j * LRF_RPN;
        float v = (r < pd.R) ? V0[pd.v0_off + (long)j * pd.R + r] : 0.f;
        v_s[i] = v;
        Vf[(long)blockIdx.x * NN * LRF_RPN + i] = v;
    }
    __syncthreads();
    make_gtable_n(v_s, NN, pd.R, Bf + (long)blockIdx.x * LRF_GTN_STRIDE, threadIdx.x, 256);
}

template <int NN>
struct BcdnLds {
    float Xs[64 * (NN + 1)];   // sub-tile, odd pitch: conflict-free for lane = row and for lane = column
    float Vs[NN * LRF_RPN];
    float a_s[64 * LRF_RPN];
    float u_s[64 * LRF_RPN];
    float gt_s[LRF_GTN_STRIDE];
};

// MODE 0: old U from int8 (iterations >= 2); MODE 2: first iteration, old U = fp32 U0 (u0 = X w0 from k_any_prod).
// Ppart: per block [NN][LRF_RPN]; Qpart: per block [LRF_RPN][LRF_RPN].
template <int NN, int MODE>
__global__ __launch_bounds__(256) void k_bcdn(const float* __restrict__ X, const PlaneDesc* __restrict__ planes,
                                              const BlockDesc* __restrict__ blocks, const float* __restrict__ Vf,
                                              const float* __restrict__ Bf, const float* __restrict__ U0,
                                              int8_t* __restrict__ U, float* __restrict__ Ppart, float* __restrict__ Qpart,
                                              float lo, float hi)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BcdnLds<NN>& L = *reinterpret_cast<BcdnLds<NN>*>(smem);
    constexpr int XLD = NN + 1;
    const BlockDesc bd = blocks[blockIdx.x];
    const PlaneDesc pd = planes[bd.plane];
    const int R = pd.R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* Xp = X + pd.x_off + (long)bd.row0 * NN;
    int8_t* Ub = U + pd.u_off + (long)bd.row0 * R;
    int nrows = pd.M - bd.row0;
    if (nrows > LRF_KC) nrows = LRF_KC;
    const int nsub = (nrows + 63) >> 6;

    for (int i = tid; i < NN * LRF_RPN; i += 256) L.Vs[i] = Vf[(long)bd.plane * NN * LRF_RPN + i];
    for (int i = tid; i < R * LRF_GTN_LD; i += 256) L.gt_s[i] = Bf[(long)bd.plane * LRF_GTN_STRIDE + i];

    float accP[LRF_RPN], accQ[4];
#pragma unroll
    for (int r = 0; r < LRF_RPN; r++) accP[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) accQ[e] = 0.f;

    for (int t = 0; t < nsub; t++) {
        const int r0 = t * 64;
        __syncthreads(); // previous sub-tile consumed
        for (int e = tid; e < 64 * (NN / 4); e += 256) {
            int row = e / (NN / 4), c4 = e - row * (NN / 4);
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r0 + row < nrows) v = *reinterpret_cast<const f32x4*>(Xp + (long)(r0 + row) * NN + 4 * c4);
            float* d = &L.Xs[row * XLD + 4 * c4];
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        { // old U rows of the sub-tile -> u_s, all threads, coalesced
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
            for (int e = tid; e < lim; e += 256) {
                const int row = e / R, r = e - row * R;
                L.u_s[row * LRF_RPN + r] = (MODE == 0) ? (float)Ub[(long)r0 * R + e]
                                                       : U0[pd.u0_off + ((long)bd.row0 + r0) * R + e];
            }
        }
        __syncthreads();
        { // a = x @ v: thread (row = lane, column group = wave) takes the columns r = wave, wave + 4, ...
            float acc[LRF_RPN / 4];
#pragma unroll
            for (int j = 0; j < LRF_RPN / 4; j++) acc[j] = 0.f;
            const float* xr = &L.Xs[lane * XLD];
            for (int k = 0; k < NN; k++) {
                const float x = xr[k];
                const float* vk = &L.Vs[k * LRF_RPN + wave];
#pragma unroll
                for (int j = 0; j < LRF_RPN / 4; j++) acc[j] = fmaf(x, vk[4 * j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < LRF_RPN / 4; j++) L.a_s[lane * LRF_RPN + wave + 4 * j] = acc[j];
        }
        __syncthreads();
        if (wave == 0) { // Gauss-Seidel, lane = row
            const int row = r0 + lane;
            float* ur = &L.u_s[lane * LRF_RPN];
            if (row < nrows) {
                gs_row_n(R, &L.a_s[lane * LRF_RPN], ur, L.gt_s, pd.native_t2_u != 0, lo, hi);
                for (int r = R; r < LRF_RPN; r++) ur[r] = 0.f;
            } else {
                for (int r = 0; r < LRF_RPN; r++) ur[r] = 0.f;
            }
        }
        __syncthreads();
        { // int8 U out, coalesced
            const int lim = (nrows - r0 < 64 ? nrows - r0 : 64) * R;
            for (int e = tid; e < lim; e += 256) {
                const int row = e / R, r = e - row * R;
                Ub[(long)r0 * R + e] = (int8_t)L.u_s[row * LRF_RPN + r];
            }
        }
        if (tid < NN) { // a' += x.mT @ u: thread = column n; rows in order; padded / missing rows have u = 0
            const float* xc = &L.Xs[tid];
            for (int m = 0; m < 64; m++) {
                const float x = xc[m * XLD];
                const float* um = &L.u_s[m * LRF_RPN];
#pragma unroll
                for (int r = 0; r < LRF_RPN; r++) accP[r] = fmaf(x, um[r], accP[r]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) { // b' += u.mT @ u (exact integers)
            const int idx = tid + 256 * e, j = idx >> 5, r = idx & 31;
            float q = accQ[e];
            for (int m = 0; m < 64; m++) q = fmaf(L.u_s[m * LRF_RPN + j], L.u_s[m * LRF_RPN + r], q);
            accQ[e] = q;
        }
    }
    const long slot = (long)pd.blk0 + bd.blk;
    if (tid < NN) {
        float* Pp = Ppart + slot * NN * LRF_RPN + (long)tid * LRF_RPN;
#pragma unroll
        for (int r = 0; r < LRF_RPN; r++) Pp[r] = accP[r];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) Qpart[slot * LRF_RPN * LRF_RPN + tid + 256 * e] = accQ[e];
}

template <int NN>
struct VupdnLds {
    float a_s[NN * LRF_RPN];
    float v_s[NN * LRF_RPN];
    float gt_s[LRF_GTN_STRIDE];
};

// V update of one matrix: a' = sum of block partials in block order, b' = u.mT @ u, Gauss-Seidel over the NN rows
// of V, then the b table of the new V (iterations before the last) or the int8 V (last iteration).
template <int NN>
__global__ __launch_bounds__(256) void k_vupdaten(const PlaneDesc* __restrict__ planes, const float* __restrict__ Ppart,
                                                  const float* __restrict__ Qpart, float* __restrict__ Vf,
                                                  float* __restrict__ Bf, int8_t* __restrict__ V8, float lo, float hi,
                                                  int write_i8)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    VupdnLds<NN>& L = *reinterpret_cast<VupdnLds<NN>*>(smem);
    const PlaneDesc pd = planes[blockIdx.x];
    const int R = pd.R, tid = threadIdx.x;
    for (int i = tid; i < NN * LRF_RPN; i += 256) {
        const float* Pp = Ppart + (long)pd.blk0 * NN * LRF_RPN + i;
        float acc = 0.f;
        for (int b = 0; b < pd.nblk; b++) {
            float v = Pp[(long)b * NN * LRF_RPN];
            acc = (b == 0) ? v : acc + v;
        }
        L.a_s[i] = acc;
        L.v_s[i] = Vf[(long)blockIdx.x * NN * LRF_RPN + i];
    }
    for (int i = tid; i < LRF_RPN * LRF_RPN; i += 256) {
        const float* Qp = Qpart + (long)pd.blk0 * LRF_RPN * LRF_RPN + i;
        float q = 0.f;
        for (int b = 0; b < pd.nblk; b++) {
            float v = Qp[(long)b * LRF_RPN * LRF_RPN];
            q = (b == 0) ? v : q + v;
        }
        int j = i >> 5, r = i & 31;
        if (j < R && r < R) {
            if (j == r) L.gt_s[r * LRF_GTN_LD + LRF_GTN_DEN] = (q + 0.f) + LRF_EPS;
            else L.gt_s[r * LRF_GTN_LD + (j < r ? j : j - 1)] = q;
        }
    }
    __syncthreads();
    if (tid < NN) {
        bool native = (long)(R - 1) * NN < 400;
        gs_row_n(R, &L.a_s[tid * LRF_RPN], &L.v_s[tid * LRF_RPN], L.gt_s, native, lo, hi);
        float* Vp = Vf + (long)blockIdx.x * NN * LRF_RPN + tid * LRF_RPN;
        for (int r = 0; r < R; r++) Vp[r] = L.v_s[tid * LRF_RPN + r];
        if (write_i8) {
            int8_t* vo = V8 + pd.v_off + (long)tid * R;
            for (int r = 0; r < R; r++) vo[r] = (int8_t)L.v_s[tid * LRF_RPN + r];
        }
    }
    __syncthreads();
    if (!write_i8) make_gtable_n(L.v_s, NN, R, Bf + (long)blockIdx.x * LRF_GTN_STRIDE, tid, 256);
}

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
