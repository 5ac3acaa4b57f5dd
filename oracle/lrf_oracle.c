/*
 * lrf_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's QMF / SVD codec arithmetic
 * (pashtari/lrf @ 2025-02-15).  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (lrf_amd/) never links, imports or calls anything in this directory.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference root).  Floating point is fp32 unless stated, compiled with
 * -ffp-contract=off so every fma below is explicit.
 *
 * Summation orders.  The reference delegates its products to torch CPU kernels; the
 * orders below were identified by bit-comparison against torch 2.10.0 (MKL 2024.2,
 * AVX512) run with torch.set_num_threads(1) (tools/pin_oracle.py re-checks them):
 *   - MKL sgemm, output n >= 2: per output element a k-ordered fma chain, K cut into
 *     blocks of 384; block sums are added in block order.                 (mm_mkl)
 *   - ATen native bmm kernel, used when contraction*rows*cols < 400
 *     (aten/src/ATen/native/LinearAlgebra.cpp baddbmm_cpu_kernel): k-ordered chain of
 *     separately rounded multiply then add, starting from 0.              (mm_native)
 *   - MKL with a single output column (the Gauss-Seidel `uu @ bb` product of
 *     lrf/factorization/qmf.py:115 when it is large enough for MKL): for K <= 6 the
 *     tree (((fma(a1,b1,a0*b0) + p5) + p3) + (p2 + p4)) with absent terms dropped;
 *     the same pattern continued for K >= 7 (descending odd terms, ascending even ones)
 *     reproduces the reference for every K up to 204 (tools/pin_oracle_anyshape.py:
 *     ranks 1..205 on all patch sizes and on whole planes).               (dot_mkl_n1)
 * With more than one BLAS thread the reference itself changes the order of the long
 * X^T U reduction, so "the reference" is pinned at one thread.
 *
 * The SVD initialisation (lrf/factorization/qmf.py:42-71 uses LAPACK sgesdd) is NOT a
 * restatement of LAPACK: it is this project's own algorithm (fp64 Gram matrix, Householder
 * tridiagonalisation, multisection + twisted factorisation for the top-R eigen-pairs,
 * deterministic sign rule), defined here and mirrored operation-for-operation by the HIP
 * kernels.  See DESIGN.md.  (lrf_oracle_jacobi_f64 is kept as an independent cross-check.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LRF_KC 384
#define LRF_EPS 1e-16f

/* ------------------------------------------------------------------------------------------------
 * products
 * ---------------------------------------------------------------------------------------------- */

/* MKL-ordered product: C[m*ldc+n] = sum_k A[m*sam+k*sak] * B[k*sbk+n*sbn].
 * Every output element is its own k-ordered fma chain (blocks of LRF_KC, block sums added in order);
 * the two fast paths below only reorder the loops ACROSS independent elements, never inside a chain. */
static void mm_mkl(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                   float* C, long ldc, long M, long K, long N)
{
    if (N <= 64 && sak == 1 && sbn == 1 && K <= LRF_KC) {
        /* rows of A contiguous (x @ v): N accumulators per row */
        for (long m = 0; m < M; m++) {
            float acc[64];
            for (long n = 0; n < N; n++) acc[n] = 0.f;
            const float* a = A + m * sam;
            for (long k = 0; k < K; k++) {
                const float ak = a[k];
                const float* b = B + k * sbk;
                for (long n = 0; n < N; n++) acc[n] = fmaf(ak, b[n], acc[n]);
            }
            for (long n = 0; n < N; n++) C[m * ldc + n] = acc[n];
        }
        return;
    }
    if (M <= 64 && N <= 64 && sam == 1 && sbn == 1) {
        /* A is a transposed view (x.mT @ u): stream over k, M x N accumulators stored [n][m] */
        float s[64][64], tot[64][64];
        for (long k0 = 0; k0 < K; k0 += LRF_KC) {
            long k1 = k0 + LRF_KC < K ? k0 + LRF_KC : K;
            for (long n = 0; n < N; n++)
                for (long m = 0; m < M; m++) s[n][m] = 0.f;
            for (long k = k0; k < k1; k++) {
                const float* a = A + k * sak;
                const float* b = B + k * sbk;
                for (long n = 0; n < N; n++) {
                    const float bn = b[n];
                    float* sn = s[n];
                    for (long m = 0; m < M; m++) sn[m] = fmaf(a[m], bn, sn[m]);
                }
            }
            for (long n = 0; n < N; n++)
                for (long m = 0; m < M; m++) tot[n][m] = (k0 == 0) ? s[n][m] : tot[n][m] + s[n][m];
        }
        for (long m = 0; m < M; m++)
            for (long n = 0; n < N; n++) C[m * ldc + n] = tot[n][m];
        return;
    }
    for (long m = 0; m < M; m++)
        for (long n = 0; n < N; n++) {
            float acc = 0.f;
            for (long k0 = 0; k0 < K; k0 += LRF_KC) {
                long k1 = k0 + LRF_KC < K ? k0 + LRF_KC : K;
                float s = 0.f;
                for (long k = k0; k < k1; k++) s = fmaf(A[m * sam + k * sak], B[k * sbk + n * sbn], s);
                acc = (k0 == 0) ? s : acc + s;
            }
            C[m * ldc + n] = acc;
        }
}

/* ATen native small-product kernel: acc = 0; acc += a*b (product rounded, then sum rounded). */
static void mm_native(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                      float* C, long ldc, long M, long K, long N)
{
    for (long m = 0; m < M; m++)
        for (long n = 0; n < N; n++) {
            float acc = 0.f;
            for (long k = 0; k < K; k++) {
                float p = A[m * sam + k * sak] * B[k * sbk + n * sbn];
                acc = acc + p;
            }
            C[m * ldc + n] = acc;
        }
}

static int aten_uses_native(long contraction, long rows, long cols)
{
    return contraction * rows * cols < 400;
}

/* torch `a @ b` for [1,M,K] @ [1,K,N] as dispatched on CPU (bmm_out_or_baddbmm_). */
static void mm_torch(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                     float* C, long ldc, long M, long K, long N)
{
    if (aten_uses_native(K, M, N)) mm_native(A, sam, sak, B, sbk, sbn, C, ldc, M, K, N);
    else mm_mkl(A, sam, sak, B, sbk, sbn, C, ldc, M, K, N);
}

/* MKL single-output-column dot of length K (see header).  a, b are K-vectors. */
static float dot_mkl_n1(const float* a, const float* b, int K)
{
    if (K == 0) return 0.f;
    if (K == 1) return a[0] * b[0];
    float odd = fmaf(a[1], b[1], a[0] * b[0]);
    int last_odd = ((K - 1) & 1) ? K - 1 : K - 2;
    for (int k = last_odd; k >= 3; k -= 2) odd = odd + a[k] * b[k];
    if (K < 3) return odd;
    float even = a[2] * b[2];
    for (int k = 4; k < K; k += 2) even = even + a[k] * b[k];
    return odd + even;
}

static float dot_native(const float* a, const float* b, int K)
{
    float acc = 0.f;
    for (int k = 0; k < K; k++) {
        float p = a[k] * b[k];
        acc = acc + p;
    }
    return acc;
}

/* ------------------------------------------------------------------------------------------------
 * projection  — lrf/factorization/qmf.py:191-195 (torch.round = ties-to-even, then clamp)
 * ---------------------------------------------------------------------------------------------- */
static inline float project(float x, int bounded, float lo, float hi)
{
    x = nearbyintf(x);
    if (bounded) {
        if (x < lo) x = lo;
        if (x > hi) x = hi;
    }
    return x;
}

/* ------------------------------------------------------------------------------------------------
 * CoordinateDescent.update_u — lrf/factorization/qmf.py:93-126 with w = [0;1], l1 = l2 = 0
 * (safe_divide(x-0, 1) and soft_thresholding(., 0) are identities: factorization/utils.py:18-40).
 *
 *   X  : rows x depth, element (i,k) at X[i*sxi + k*sxk]     (update_v passes the transposed strides)
 *   Uo : rows x R row-major, updated in place (Gauss-Seidel over columns, qmf.py:109-119)
 *   Vf : depth x R row-major, the fixed factor
 * ---------------------------------------------------------------------------------------------- */
static void update_factor(const float* X, long sxi, long sxk, long rows, long depth, int R,
                          float* Uo, const float* Vf, int bounded, float lo, float hi,
                          float* a_ws /* rows*R */, float* b_ws /* R*R */)
{
    /* qmf.py:107  a = x @ v ; b = v.mT @ v */
    mm_torch(X, sxi, sxk, Vf, R, 1, a_ws, R, rows, depth, R);
    mm_torch(Vf, 1, R, Vf, R, 1, b_ws, R, R, depth, R);

    if (R == 1) { /* qmf.py:120-124 */
        for (long i = 0; i < rows; i++)
            Uo[i] = project((a_ws[i] + LRF_EPS) / (b_ws[0] + 0.f + LRF_EPS), bounded, lo, hi);
        return;
    }
    int native = aten_uses_native(R - 1, rows, 1);
    float* uu = (float*)malloc(sizeof(float) * 2 * (size_t)R);
    float* bb = uu + R;
    for (int r = 0; r < R; r++) {
        int n = 0;
        for (int j = 0; j < R; j++)
            if (j != r) bb[n++] = b_ws[j * R + r];      /* qmf.py:114 */
        float den = (b_ws[r * R + r] + 0.f) + LRF_EPS;  /* qmf.py:117-118 (l2 = 0) */
        for (long i = 0; i < rows; i++) {
            n = 0;
            for (int j = 0; j < R; j++)
                if (j != r) uu[n++] = Uo[i * R + j];    /* qmf.py:113 (already-updated columns) */
            float term2 = native ? dot_native(uu, bb, R - 1) : dot_mkl_n1(uu, bb, R - 1); /* :115 */
            float num = a_ws[i * R + r] - term2;        /* qmf.py:116 */
            Uo[i * R + r] = project((num + LRF_EPS) / den, bounded, lo, hi); /* :118-119 */
        }
    }
    free(uu);
}

/* QMF.decompose iterations — lrf/factorization/qmf.py:207-212 / CoordinateDescent.forward :149-164.
 * U [M,R], V [N,R] hold the initial factors on entry and the result on return (fp32, integer valued
 * when num_iters >= 1).  Any M, N, R >= 1.  Returns 0, or -1 on allocation failure. */
int lrf_oracle_bcd(const float* X, long M, long N, int R, int num_iters,
                   int bounded, float lo, float hi, float* U, float* V)
{
    if (R < 1) return -1;
    long mx = M > N ? M : N;
    float* a_ws = (float*)malloc(sizeof(float) * ((size_t)mx * R + (size_t)R * R));
    if (!a_ws) return -1;
    float* b_ws = a_ws + (size_t)mx * R;
    if (bounded) { lo = ceilf(lo); hi = floorf(hi); } /* qmf.py:194 math.ceil / math.floor */
    for (int it = 0; it < num_iters; it++) {
        update_factor(X, N, 1, M, N, R, U, V, bounded, lo, hi, a_ws, b_ws); /* update_u :159 */
        update_factor(X, 1, N, N, M, R, V, U, bounded, lo, hi, a_ws, b_ws); /* update_v :161 (x.mT) */
    }
    free(a_ws);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * The general CoordinateDescent.forward — lrf/factorization/qmf.py:93-164 with every option the class has:
 * unbounded or bounded projection (QMF._project :191-195), elastic-net terms l1 / l2 (:116-118,
 * soft_thresholding factorization/utils.py:36-40), any subset of the factors (0 = u, 1 = v, 2 = w), and the affine pair w
 * (x ~ w0 + w1 u v^T): x is replaced by safe_divide(x - w0, w1) in the u / v updates (:104-105, utils.py:18-33) and
 * update_w (:141-147) refits (w0, w1) by least squares of x on [1, u v^T].
 * update_w: the reference calls torch.linalg.lstsq (LAPACK gelsy on the [M N, 2] design matrix); here the 2 x 2 normal
 * equations in fp64 — equal to ~1e-7 relative, NOT bit for bit: everything downstream of an updated w is parity by
 * tolerance.  Without factor 2 (w stays [0; 1], the affine map is the identity) the results are the reference's bit for bit.
 * ---------------------------------------------------------------------------------------------- */
static float soft_threshold(float x, float thr)
{
    if (thr == 0.f) return x; /* utils.py:37-38 */
    float ax = fabsf(x) - thr; /* sign(x) * relu(|x| - thr) */
    float sg = (x > 0.f) ? 1.f : (x < 0.f ? -1.f : 0.f);
    return sg * (ax > 0.f ? ax : 0.f);
}

static void update_factor_ex(const float* X, long sxi, long sxk, long rows, long depth, int R,
                             float* Uo, const float* Vf, int bounded, float lo, float hi, float l1, float l2,
                             float* a_ws, float* b_ws, float eps)
{
    mm_torch(X, sxi, sxk, Vf, R, 1, a_ws, R, rows, depth, R);
    mm_torch(Vf, 1, R, Vf, R, 1, b_ws, R, R, depth, R);
    if (R == 1) { /* qmf.py:120-124 */
        for (long i = 0; i < rows; i++)
            Uo[i] = project((soft_threshold(a_ws[i], l1) + eps) / ((b_ws[0] + l2) + eps), bounded, lo, hi);
        return;
    }
    int native = aten_uses_native(R - 1, rows, 1);
    float* uu = (float*)malloc(sizeof(float) * 2 * (size_t)R);
    float* bb = uu + R;
    for (int r = 0; r < R; r++) {
        int n = 0;
        for (int j = 0; j < R; j++)
            if (j != r) bb[n++] = b_ws[j * R + r];
        float den = (b_ws[r * R + r] + l2) + eps; /* qmf.py:117-118 */
        for (long i = 0; i < rows; i++) {
            n = 0;
            for (int j = 0; j < R; j++)
                if (j != r) uu[n++] = Uo[i * R + j];
            float term2 = native ? dot_native(uu, bb, R - 1) : dot_mkl_n1(uu, bb, R - 1);
            float num = soft_threshold(a_ws[i * R + r] - term2, l1); /* qmf.py:116 */
            Uo[i * R + r] = project((num + eps) / den, bounded, lo, hi);
        }
    }
    free(uu);
}

/* factors: bit 0 = update u, bit 1 = update v, bit 2 = update w.  l2u, l2v: the `l2` pair; l1_ratio as in the class
 * (l1_u = l2u * l1_ratio * N, l2_u = l2u * (1 - l1_ratio) * N, and with M for v: qmf.py:154-157, evaluated in double like
 * Python and rounded to fp32 where they meet fp32 tensors).  W[2] = (w0, w1), in: the initial pair ([0; 1] from SVDInit). */
int lrf_oracle_bcd_ex_eps(const float* X, long M, long N, int R, int num_iters, int bounded, float lo, float hi,
                          double l2u, double l2v, double l1_ratio, int factors, float* U, float* V, float* W, double eps_d);
int lrf_oracle_bcd_ex(const float* X, long M, long N, int R, int num_iters, int bounded, float lo, float hi,
                      double l2u, double l2v, double l1_ratio, int factors, float* U, float* V, float* W)
{
    return lrf_oracle_bcd_ex_eps(X, M, N, R, num_iters, bounded, lo, hi, l2u, l2v, l1_ratio, factors, U, V, W, 1e-16);
}
/* eps_d: CoordinateDescent's eps (qmf.py:82, 90, 117-118), a Python float that meets fp32 tensors: rounded to fp32.  The eps of
 * safe_divide (utils.py:18) is that function's own default and stays 1e-16. */
int lrf_oracle_bcd_ex_eps(const float* X, long M, long N, int R, int num_iters, int bounded, float lo, float hi,
                          double l2u, double l2v, double l1_ratio, int factors, float* U, float* V, float* W, double eps_d)
{
    const float eps = (float)eps_d;
    if (R < 1) return -1;
    long mx = M > N ? M : N;
    float* a_ws = (float*)malloc(sizeof(float) * ((size_t)mx * R + (size_t)R * R));
    float* Xp = (float*)malloc(sizeof(float) * (size_t)M * N);
    if (!a_ws || !Xp) return -1;
    float* b_ws = a_ws + (size_t)mx * R;
    if (bounded) { lo = ceilf(lo); hi = floorf(hi); }
    const float l1_u = (float)(l2u * l1_ratio * (double)N), l2_u = (float)(l2u * (1.0 - l1_ratio) * (double)N);
    const float l1_v = (float)(l2v * l1_ratio * (double)M), l2_v = (float)(l2v * (1.0 - l1_ratio) * (double)M);
    for (int it = 0; it < num_iters; it++) {
        /* x <- safe_divide(x - w0, w1): utils.py:18-33 (|w1| < eps -> eps * sign(w1)) */
        const float w0 = W[0], w1 = W[1];
        float den = w1;
        if (fabsf(w1) < LRF_EPS) den = LRF_EPS * ((w1 > 0.f) ? 1.f : (w1 < 0.f ? -1.f : 0.f));
        for (long e = 0; e < M * N; e++) Xp[e] = (X[e] - w0) / den;
        if (factors & 1) update_factor_ex(Xp, N, 1, M, N, R, U, V, bounded, lo, hi, l1_u, l2_u, a_ws, b_ws, eps);
        if (factors & 2) update_factor_ex(Xp, 1, N, N, M, R, V, U, bounded, lo, hi, l1_v, l2_v, a_ws, b_ws, eps);
        if (factors & 4) { /* update_w: least squares of x on [1, z], z = u v^T (fp32 product, k-ordered like u @ v.mT) */
            double n = (double)M * (double)N, sz = 0, szz = 0, sx = 0, sxz = 0;
            for (long m = 0; m < M; m++)
                for (long j = 0; j < N; j++) {
                    float z = 0.f;
                    for (int r = 0; r < R; r++) z = fmaf(U[m * R + r], V[j * R + r], z);
                    double zd = (double)z, xd = (double)X[m * N + j];
                    sz += zd; szz += zd * zd; sx += xd; sxz += xd * zd;
                }
            double det = n * szz - sz * sz;
            double w1n = (det != 0.0) ? (n * sxz - sz * sx) / det : 0.0;
            double w0n = (sx - w1n * sz) / n;
            W[0] = (float)w0n;
            W[1] = (float)w1n;
        }
    }
    free(Xp);
    free(a_ws);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SVD initialisation (this project's algorithm; replaces LAPACK at lrf/factorization/qmf.py:44-48)
 * ---------------------------------------------------------------------------------------------- */

/* fp64 Gram matrix G = X^T X of an M x N fp32 matrix, N <= 64... general N.
 * Order: four interleaved accumulators; accumulator w takes the 4-row steps s with s % 4 == w
 * (rows 4s..4s+3 in order, fma chain); G = ((g0 + g1) + g2) + g3.  Mirrors one workgroup of four
 * waves on the GPU. */
void lrf_oracle_gram_f64(const float* X, long M, long N, double* G)
{
    double* acc = (double*)calloc((size_t)4 * N * N, sizeof(double));
    double* xd = (double*)malloc(sizeof(double) * N);
    long nsteps = (M + 3) / 4;
    for (long s = 0; s < nsteps; s++) {
        double* g = acc + (size_t)(s & 3) * N * N;
        for (long m = 4 * s; m < 4 * s + 4 && m < M; m++) {
            const float* x = X + m * N;
            for (long i = 0; i < N; i++) xd[i] = (double)x[i];
            for (long i = 0; i < N; i++) { /* upper triangle; (j,i) is the same fma sequence */
                double xi = xd[i];
                double* gi = g + i * N;
                for (long j = i; j < N; j++) gi[j] = fma(xi, xd[j], gi[j]);
            }
        }
    }
    for (long i = 0; i < N; i++)
        for (long j = i; j < N; j++) {
            long e = i * N + j;
            double v = ((acc[e] + acc[N * N + e]) + acc[2 * N * N + e]) + acc[3 * N * N + e];
            G[i * N + j] = v;
            G[j * N + i] = v;
        }
    free(xd);
    free(acc);
}

/* ---- exact Gram matrix of the 64-column path -------------------------------------------------------
 * G = X^T X is computed EXACTLY and rounded once: every element x is placed on the fixed-point grid
 * 2^(E-35), n = rint(x * 2^(35-E)) with max|x| < 2^E (so |n| < 2^35; for the planes qmf_encode forms —
 * fp32 values that are 0 or in [0.114, 255.5] — no rounding happens for any E <= 8), the integer sums
 * S_ij = sum_m n_mi n_mj (< 2^87) are accumulated exactly and G_ij = rne_f64(S_ij) * 4^(E-35).
 * Being exact, the result does not depend on any summation order: the GPU is free to accumulate it with
 * int8 MFMAs over 7-bit digits, per row chunk, per workgroup, fused into the kernel that produces X.
 */
#define LRF_GRAM_BITS 35

/* E with max|x| < 2^E from the largest magnitude's bit pattern: x = 1.m * 2^(e-127) < 2^(e-126); 0 for an all-zero matrix */
int lrf_oracle_gram_exponent(const float* X, long n)
{
    uint32_t mx = 0;
    for (long i = 0; i < n; i++) {
        uint32_t b;
        memcpy(&b, X + i, 4);
        b &= 0x7fffffffu;
        if (b > mx) mx = b;
    }
    if (mx == 0) return 0;
    return (int)(mx >> 23) - 126;
}

/* magnitude (hi:lo, 128 bits) -> double, round to nearest even; spelled out so that the GPU runs the same steps */
static double u128_to_double_rne(uint64_t hi, uint64_t lo)
{
    if (hi == 0 && lo == 0) return 0.0;
    int nbits = hi ? 128 - __builtin_clzll(hi) : 64 - __builtin_clzll(lo);
    if (nbits <= 53) return (double)lo;
    int sh = nbits - 53; /* 1..75 */
    unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    uint64_t mant = (uint64_t)(v >> sh);
    unsigned __int128 rem = v & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
    if (rem > half || (rem == half && (mant & 1))) mant++;
    return ldexp((double)mant, sh);
}

void lrf_oracle_gram_exact(const float* X, long M, long N, int E, double* G)
{
    /* n = h 2^18 + l with |h| < 2^17, |l| < 2^18 (same sign): the three digit-pair sums of a chunk of <= 2^16 rows fit
     * int64 exactly (plain 32 x 32 -> 64-bit multiplies, which vectorise); chunks are added as 128-bit integers. */
    const long CH = 1L << 16;
    int32_t* h = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* l = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int64_t* hh = (int64_t*)malloc(sizeof(int64_t) * (size_t)N * N * 3);
    int64_t *hl = hh + N * N, *ll = hl + N * N;
    __int128* S = (__int128*)calloc((size_t)N * N, sizeof(__int128));
    const double scale = ldexp(1.0, LRF_GRAM_BITS - E);
    const long long lim = (1LL << LRF_GRAM_BITS) - 1;
    for (long m0 = 0; m0 < M; m0 += CH) {
        long m1 = m0 + CH < M ? m0 + CH : M;
        memset(hh, 0, sizeof(int64_t) * (size_t)N * N * 3);
        for (long m = m0; m < m1; m++) {
            for (long i = 0; i < N; i++) {
                double r = nearbyint((double)X[m * N + i] * scale);
                long long n = (r >= (double)lim) ? lim : (r <= -(double)lim) ? -lim : (long long)r;
                long long a = n < 0 ? -n : n;
                int32_t hv = (int32_t)(a >> 18), lv = (int32_t)(a & ((1 << 18) - 1));
                h[i] = n < 0 ? -hv : hv;
                l[i] = n < 0 ? -lv : lv;
            }
            for (long i = 0; i < N; i++) {
                const int32_t hi_ = h[i], li_ = l[i];
                int64_t* restrict ph = hh + i * N;
                int64_t* restrict pm = hl + i * N;
                int64_t* restrict pl = ll + i * N;
                for (long j = i; j < N; j++) { /* 32 x 32 -> 64-bit products (vpmuldq) */
                    ph[j] += (int64_t)hi_ * (int64_t)h[j];
                    pm[j] += (int64_t)hi_ * (int64_t)l[j] + (int64_t)li_ * (int64_t)h[j];
                    pl[j] += (int64_t)li_ * (int64_t)l[j];
                }
            }
        }
        for (long e = 0; e < N * N; e++) S[e] += ((__int128)hh[e] << 36) + ((__int128)hl[e] << 18) + (__int128)ll[e];
    }
    const double back = ldexp(1.0, 2 * (E - LRF_GRAM_BITS));
    for (long i = 0; i < N; i++)
        for (long j = i; j < N; j++) {
            __int128 s_ = S[i * N + j];
            int neg = s_ < 0;
            unsigned __int128 a = neg ? (unsigned __int128)(-s_) : (unsigned __int128)s_;
            double v = u128_to_double_rne((uint64_t)(a >> 64), (uint64_t)a) * back;
            if (neg) v = -v;
            G[i * N + j] = v;
            G[j * N + i] = v;
        }
    free(S);
    free(hh);
    free(l);
    free(h);
}

/* Cyclic Jacobi eigen-solve of a symmetric n x n fp64 matrix (n even), round-robin parallel order.
 * A is destroyed (diagonal = eigenvalues); E (n x n, row-major) receives eigenvectors in columns.
 * Each round: rotation parameters for the n/2 disjoint pairs from the current A (upper triangle),
 * then a row phase, then a column phase (A and E).  Returns the number of sweeps used. */
static void rr_pair(int n, int t, int i, int* p, int* q)
{
    int a, b;
    if (i == 0) { a = n - 1; b = t % (n - 1); }
    else { a = (t + i) % (n - 1); b = (t - i + (n - 1)) % (n - 1); }
    if (a < b) { *p = a; *q = b; } else { *p = b; *q = a; }
}

int lrf_oracle_jacobi_f64(double* A, int n, double* E, int max_sweeps)
{
    int np = n / 2;
    double* cs = (double*)malloc(sizeof(double) * 2 * np);
    int* pq = (int*)malloc(sizeof(int) * 2 * np);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) E[i * n + j] = (i == j) ? 1.0 : 0.0;
    int sweep;
    for (sweep = 0; sweep < max_sweeps; sweep++) {
        int rotated = 0;
        for (int t = 0; t < n - 1; t++) {
            for (int i = 0; i < np; i++) {
                int p, q;
                rr_pair(n, t, i, &p, &q);
                pq[2 * i] = p; pq[2 * i + 1] = q;
                double apq = A[p * n + q], app = A[p * n + p], aqq = A[q * n + q];
                double c = 1.0, s = 0.0;
                /* skip when |apq| <= 2^-40 sqrt(|app aqq|): negligible for fp32 factors */
                if (apq * apq > 8.271806125530277e-25 * fabs(app * aqq)) {
                    double tau = (aqq - app) / (2.0 * apq);
                    double tt = 1.0 / (fabs(tau) + sqrt(1.0 + tau * tau));
                    if (tau < 0.0) tt = -tt;
                    c = 1.0 / sqrt(1.0 + tt * tt);
                    s = tt * c;
                    rotated = 1;
                }
                cs[2 * i] = c; cs[2 * i + 1] = s;
            }
            for (int i = 0; i < np; i++) { /* row phase: A <- J^T A */
                int p = pq[2 * i], q = pq[2 * i + 1];
                double c = cs[2 * i], s = cs[2 * i + 1];
                if (s == 0.0) continue;
                for (int k = 0; k < n; k++) {
                    double gp = A[p * n + k], gq = A[q * n + k];
                    A[p * n + k] = c * gp - s * gq;
                    A[q * n + k] = s * gp + c * gq;
                }
            }
            for (int k = 0; k < n; k++) { /* column phase: A <- A J ; E <- E J (pairs are disjoint) */
                double* ak = A + k * n;
                double* ek = E + k * n;
                for (int i = 0; i < np; i++) {
                    double c = cs[2 * i], s = cs[2 * i + 1];
                    if (s == 0.0) continue;
                    int p = pq[2 * i], q = pq[2 * i + 1];
                    double gp = ak[p], gq = ak[q];
                    ak[p] = c * gp - s * gq;
                    ak[q] = s * gp + c * gq;
                    double ep = ek[p], eq = ek[q];
                    ek[p] = c * ep - s * eq;
                    ek[q] = s * ep + c * eq;
                }
            }
            for (int i = 0; i < np; i++) {
                if (cs[2 * i + 1] == 0.0) continue;
                int p = pq[2 * i], q = pq[2 * i + 1];
                A[p * n + q] = 0.0;
                A[q * n + p] = 0.0;
            }
        }
        if (!rotated) break;
    }
    free(cs);
    free(pq);
    return sweep;
}

/* ------------------------------------------------------------------------------------------------
 * Top-R eigen-pairs of the 64 x 64 Gram matrix: Householder tridiagonalisation, eigenvalues by
 * 64-way multisection on Sturm counts, eigenvectors by twisted factorisation + Gram-Schmidt, then
 * back-transformation.  Every reduction has a fixed shape so that the GPU kernel (one workgroup per
 * matrix) reproduces it bit for bit:
 *   tree64(s)   : for off = 32,16,...,1: s[i] += s[i+off] (i < off); result s[0]   (wave butterfly)
 *   matvec      : eight partial k-ordered fma chains over column groups j>>3, combined pairwise, then in order
 * ---------------------------------------------------------------------------------------------- */
#define EN 64

static double tree64(double* s)
{
    for (int off = 32; off >= 1; off >>= 1)
        for (int i = 0; i < off; i++) s[i] = s[i] + s[i + off];
    return s[0];
}

/* A (EN x EN, symmetric, destroyed) -> d[EN], e[EN-1]; reflectors H_k = I - tau_k v_k v_k^T with
 * v_k stored in Vh[k*EN + i] (zero for i <= k). */
#define LRF_SIGMA_TINY 1e-280
static void tridiagonalize(double* A, double* d, double* e, double* Vh, double* tau)
{
    double s[EN], v[EN], p[EN], w[EN];
    memset(Vh, 0, sizeof(double) * EN * EN);
    for (int k = 0; k < EN - 2; k++) {
        for (int i = 0; i < EN; i++) { double x = (i > k) ? A[i * EN + k] : 0.0; s[i] = x * x; }
        double sigma = tree64(s);
        tau[k] = 0.0;
        e[k] = 0.0;
        /* a column that is zero up to cascaded rounding noise (sigma in the denormal range would overflow t = 2 / vn):
         * rank-deficient Gram matrices, e.g. constant planes */
        if (!(sigma > LRF_SIGMA_TINY)) continue;
        double x0 = A[(k + 1) * EN + k];
        double nrm = sqrt(sigma);
        double alpha = (x0 >= 0.0) ? -nrm : nrm;
        for (int i = 0; i < EN; i++) v[i] = (i > k + 1) ? A[i * EN + k] : 0.0;
        v[k + 1] = x0 - alpha;
        /* |v|^2 = sigma - x0^2 + (x0 - alpha)^2 = 2 (sigma + |x0| nrm): t = 2 / |v|^2 without a second reduction */
        double t = 1.0 / fma(fabs(x0), nrm, sigma);
        for (int i = 0; i < EN; i++) {
            double c[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (i > k)
                for (int j = k + 1; j < EN; j++) c[j >> 3] = fma(A[i * EN + j], v[j], c[j >> 3]);
            p[i] = t * ((((c[0] + c[1]) + (c[2] + c[3])) + (c[4] + c[5])) + (c[6] + c[7]));
        }
        for (int i = 0; i < EN; i++) s[i] = p[i] * v[i];
        double K = (0.5 * t) * tree64(s);
        for (int i = 0; i < EN; i++) w[i] = fma(-K, v[i], p[i]);
        for (int i = k + 1; i < EN; i++)
            for (int j = k + 1; j <= i; j++) { /* both products rounded, then added (commutative): exactly symmetric */
                double m1 = v[i] * w[j], m2 = w[i] * v[j];
                double val = A[i * EN + j] - (m1 + m2);
                A[i * EN + j] = val;
                A[j * EN + i] = val;
            }
        e[k] = alpha;
        tau[k] = t;
        for (int i = 0; i < EN; i++) Vh[k * EN + i] = v[i];
    }
    for (int i = 0; i < EN; i++) d[i] = A[i * EN + i];
    e[EN - 2] = A[(EN - 1) * EN + (EN - 2)];
}

/* Number of eigenvalues of T' smaller than x: sign changes along the leading principal minors
 *   p_0 = 1, p_{i+1} = (d'_i - x) p_i - e'_{i-1}^2 p_{i-1}
 * — the Sturm sequence in its division-free form (the kernel's inner loop is one dependent fma per step instead of an fp64
 * division).  T' = T / 2^s is scaled so that its Gershgorin hull lies in [-1, 1]; every eighth step the pair (p_i, p_{i-1})
 * is multiplied by the power of two that brings the larger into [0.5, 1) (exact).  A minor that comes out as zero takes a
 * tiny value of the sign opposite to its predecessor's, which also restarts the sequence correctly behind e' = 0. */
static int sturm_count(const double* ds, const double* e2s, double x)
{
    double p = 1.0, pp = 0.0;
    int cnt = 0;
    for (int i = 0; i < EN; i++) {
        double dx = ds[i] - x;
        double t = (i ? e2s[i - 1] : 0.0) * pp;
        double pn = fma(dx, p, -t);
        if (pn == 0.0) pn = signbit(p) ? 0x1p-200 : -0x1p-200;
        cnt += (signbit(pn) != 0) != (signbit(p) != 0);
        pp = p;
        p = pn;
        if ((i & 7) == 7) {
            int ea, eb;
            frexp(p, &ea);
            frexp(pp, &eb);
            int m = ea > eb ? ea : eb;
            p = ldexp(p, -m);
            pp = ldexp(pp, -m);
        }
    }
    return cnt;
}

/* the R largest eigenvalues of T, descending: 10 passes of 64-way multisection from the Gershgorin hull, on the scaled
 * matrix; pivmin is the pivot guard of the twisted factorisation */
static void top_eigenvalues(const double* d, const double* e, int R, double* lam, double* pivmin_out)
{
    double ds[EN], e2s[EN], lo = 0, hi = 0, e2max = 0.0;
    for (int i = 0; i < EN - 1; i++) { double e2 = e[i] * e[i]; if (e2 > e2max) e2max = e2; }
    for (int i = 0; i < EN; i++) {
        double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i < EN - 1 ? fabs(e[i]) : 0.0);
        double a = d[i] - rad, b = d[i] + rad;
        if (i == 0 || a < lo) lo = a;
        if (i == 0 || b > hi) hi = b;
    }
    double tn = fabs(lo) > fabs(hi) ? fabs(lo) : fabs(hi);
    double pivmin = 2.2250738585072014e-300 * (e2max > 1.0 ? e2max : 1.0);
    double slack = 2.0 * tn * 2.220446049250313e-16 * EN + 2.0 * pivmin;
    lo -= slack;
    hi += slack;
    int s;
    frexp(fabs(lo) > fabs(hi) ? fabs(lo) : fabs(hi), &s); /* hull within [-2^s, 2^s] */
    for (int i = 0; i < EN; i++) ds[i] = ldexp(d[i], -s);
    for (int i = 0; i < EN - 1; i++) { double es = ldexp(e[i], -s); e2s[i] = es * es; }
    lo = ldexp(lo, -s);
    hi = ldexp(hi, -s);
    for (int r = 0; r < R; r++) {
        int k = EN - 1 - r; /* eigenvalue with exactly k eigenvalues below it */
        double a = lo, b = hi;
        for (int pass = 0; pass < 10; pass++) {
            double h = (b - a) / 65.0;
            int j = EN;
            double xs[EN];
            for (int i = 0; i < EN; i++) {
                xs[i] = a + h * (double)(i + 1);
                if (j == EN && sturm_count(ds, e2s, xs[i]) > k) j = i;
            }
            double na = (j == 0) ? a : xs[j - 1], nb = (j == EN) ? b : xs[j];
            a = na;
            b = nb;
        }
        lam[r] = ldexp(0.5 * (a + b), s);
    }
    *pivmin_out = pivmin;
}

/* eigenvector of T for eigenvalue lam by twisted factorisation (not normalised) */
static void twisted_vector(const double* d, const double* e, double lam, double pivmin, double* x)
{
    double Dp[EN], Dm[EN];
    Dp[0] = d[0] - lam;
    for (int i = 1; i < EN; i++) {
        double q = Dp[i - 1];
        if (fabs(q) < pivmin) q = -pivmin;
        Dp[i] = (d[i] - lam) - (e[i - 1] * e[i - 1]) / q;
    }
    Dm[EN - 1] = d[EN - 1] - lam;
    for (int i = EN - 2; i >= 0; i--) {
        double q = Dm[i + 1];
        if (fabs(q) < pivmin) q = -pivmin;
        Dm[i] = (d[i] - lam) - (e[i] * e[i]) / q;
    }
    int k = 0;
    double best = 0.0;
    for (int i = 0; i < EN; i++) {
        double g = fabs((Dp[i] + Dm[i]) - (d[i] - lam));
        if (i == 0 || g < best) { best = g; k = i; }
    }
    x[k] = 1.0;
    for (int i = k - 1; i >= 0; i--) {
        double q = Dp[i];
        if (fabs(q) < pivmin) q = -pivmin;
        x[i] = -(e[i] / q) * x[i + 1];
    }
    for (int i = k; i < EN - 1; i++) {
        double q = Dm[i + 1];
        if (fabs(q) < pivmin) q = -pivmin;
        x[i + 1] = -(e[i] / q) * x[i];
    }
}

/* G (EN x EN symmetric, destroyed) -> lam[R] (descending) and orthonormal eigenvectors Ev[r*EN + j]. */
int lrf_oracle_top_eig_f64(double* G, int R, double* lam, double* Ev)
{
    double d[EN], e[EN], tau[EN], s[EN], pivmin;
    double* Vh = (double*)malloc(sizeof(double) * EN * EN);
    double* Z = (double*)malloc(sizeof(double) * EN * R); /* vectors in tridiagonal coordinates */
    tridiagonalize(G, d, e, Vh, tau);
    top_eigenvalues(d, e, R, lam, &pivmin);
    for (int r = 0; r < R; r++) {
        double* x = Z + r * EN;
        twisted_vector(d, e, lam[r], pivmin, x);
        int use_twisted = 1, uidx = 0;
        for (int i = 0; i < EN; i++) use_twisted &= isfinite(x[i]) != 0;
        for (;;) {
            if (use_twisted) { /* scale first: the Gram-Schmidt loss test below is relative */
                for (int i = 0; i < EN; i++) s[i] = x[i] * x[i];
                double n0 = sqrt(tree64(s));
                for (int i = 0; i < EN; i++) x[i] = x[i] / n0;
            } else { /* deterministic fallback: unit vectors in turn */
                if (uidx >= EN) { free(Vh); free(Z); return -1; }
                for (int i = 0; i < EN; i++) x[i] = (i == uidx) ? 1.0 : 0.0;
                uidx++;
            }
            for (int pr = 0; pr < r; pr++) { /* modified Gram-Schmidt against the vectors already fixed */
                const double* pv = Z + pr * EN;
                for (int i = 0; i < EN; i++) s[i] = pv[i] * x[i];
                double c = tree64(s);
                for (int i = 0; i < EN; i++) x[i] = fma(-c, pv[i], x[i]);
            }
            for (int i = 0; i < EN; i++) s[i] = x[i] * x[i];
            double n2 = tree64(s);
            if (n2 > 1e-6 && n2 < 1e300) { /* kept (a repeated eigenvalue collapses to ~0 here; NaN fails both) */
                double nr = sqrt(n2);
                for (int i = 0; i < EN; i++) x[i] = x[i] / nr;
                break;
            }
            use_twisted = 0;
        }
    }
    for (int r = 0; r < R; r++) { /* back-transform: x <- H_0 H_1 ... H_{n-3} x */
        double* x = Ev + r * EN;
        memcpy(x, Z + r * EN, sizeof(double) * EN);
        for (int k = EN - 3; k >= 0; k--) {
            if (tau[k] == 0.0) continue;
            const double* v = Vh + k * EN;
            for (int i = 0; i < EN; i++) s[i] = v[i] * x[i];
            double sc = tau[k] * tree64(s);
            for (int i = 0; i < EN; i++) x[i] = fma(-sc, v[i], x[i]);
        }
    }
    free(Vh);
    free(Z);
    return 0;
}

/* Top-R factors from the Gram eigen-pairs:
 *   sigma_r = sqrt(max(lambda_r,0)); s_r = sqrt(sigma_r)
 *   v0[:,r] = e_r * s_r                 (lrf/factorization/qmf.py:46,48  v = (sqrt(s) Vh)^T)
 *   w0[:,r] = e_r / s_r  (0 if s_r = 0) so that u0 = X w0 = U sqrt(s)   (qmf.py:46-47)
 * Column sign: sign[r] if non-zero, else -1, is imposed on  sum_j (j+1) e_r[j]  (the reference's
 * LAPACK sign is arbitrary; component 0 usually comes out negative there, which -1 reproduces).
 * Columns r >= min(M,N) are zero (qmf.py:50-52).  G is N x N (destroyed), N must be 64. */
int lrf_oracle_init_from_gram(double* G, long M, long N, int R, const int8_t* sign, float* v0, float* w0)
{
    int n = (int)N;
    if (n != EN || R > EN) return -1;
    long rmax = M < N ? M : N;
    int Rc = R < rmax ? R : (int)rmax;
    double lam[EN];
    double* Ev = (double*)malloc(sizeof(double) * EN * EN);
    if (lrf_oracle_top_eig_f64(G, Rc, lam, Ev)) { free(Ev); return -1; }
    for (int r = 0; r < R; r++) {
        if (r >= rmax) {
            for (int j = 0; j < n; j++) { v0[j * R + r] = 0.f; w0[j * R + r] = 0.f; }
            continue;
        }
        const double* E = Ev + r * EN;
        /* eigenvalues at the multisection noise floor (hull slack ~1e-300) count as zero: no inf in w0 */
        double sigma = sqrt(lam[r] > 1e-200 ? lam[r] : 0.0);
        double sr = sqrt(sigma);
        double dot = 0.0;
        for (int j = 0; j < n; j++) dot = fma((double)(j + 1), E[j], dot);
        double want = (sign && sign[r]) ? (double)sign[r] : -1.0;
        double flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.0 : -1.0;
        for (int j = 0; j < n; j++) {
            double ev = flip * E[j];
            v0[j * R + r] = (float)(ev * sr);
            w0[j * R + r] = (sr > 0.0) ? (float)(ev / sr) : 0.f;
        }
    }
    free(Ev);
    return 0;
}

/* Full init: u0 [M,R], v0 [N,R].  u0 = X w0 with the same k-ordered fma chain as x @ v. */
int lrf_oracle_svd_init(const float* X, long M, long N, int R, const int8_t* sign, float* u0, float* v0)
{
    double* G = (double*)malloc(sizeof(double) * N * N);
    float* w0 = (float*)malloc(sizeof(float) * N * R);
    /* the 64-column path (the only one init_from_gram takes): the exact Gram matrix */
    lrf_oracle_gram_exact(X, M, N, lrf_oracle_gram_exponent(X, M * N), G);
    int rc = lrf_oracle_init_from_gram(G, M, N, R, sign, v0, w0);
    if (rc == 0) mm_mkl(X, N, 1, w0, R, 1, u0, R, M, N, R);
    free(G);
    free(w0);
    return rc;
}

/* QMF.decompose — lrf/factorization/qmf.py:197-214 (init + num_iters BCD iterations). */
int lrf_oracle_qmf_decompose(const float* X, long M, long N, int R, int num_iters, int bounded,
                             float lo, float hi, const int8_t* sign, float* U, float* V)
{
    int rc = lrf_oracle_svd_init(X, M, N, R, sign, U, V);
    if (rc) return rc;
    return lrf_oracle_bcd(X, M, N, R, num_iters, bounded, lo, hi, U, V);
}

/* ------------------------------------------------------------------------------------------------
 * colour, resampling, padding, patches — lrf/compression/utils.py, lrf/compression/qmf.py
 * ---------------------------------------------------------------------------------------------- */

/* rgb_to_ycbcr — lrf/compression/utils.py:24-47: offset + einsum("ij,j...->i...", T, rgb.float()).
 * The einsum is an sgemm with K = 3: k-ordered fma chain from 0.  rgb is CHW uint8. */
void lrf_oracle_rgb_to_ycbcr(const uint8_t* rgb, long H, long W, float* ycc)
{
    static const float T[3][3] = {{0.299f, 0.587f, 0.114f},
                                  {-0.168736f, -0.331264f, 0.5f},
                                  {0.5f, -0.418688f, -0.081312f}};
    static const float off[3] = {0.f, 128.f, 128.f};
    long hw = H * W;
    for (int i = 0; i < 3; i++)
        for (long p = 0; p < hw; p++) {
            float acc = 0.f;
            for (int j = 0; j < 3; j++) acc = fmaf(T[i][j], (float)rgb[j * hw + p], acc);
            ycc[i * hw + p] = off[i] + acc;
        }
}

/* ycbcr_to_rgb — lrf/compression/utils.py:50-73: einsum(T', ycc + offset). */
void lrf_oracle_ycbcr_to_rgb(const float* ycc, long H, long W, float* rgb)
{
    static const float T[3][3] = {{1.0f, 0.0f, 1.402f}, {1.0f, -0.344136f, -0.714136f}, {1.0f, 1.772f, 0.0f}};
    static const float off[3] = {0.f, -128.f, -128.f};
    long hw = H * W;
    for (int i = 0; i < 3; i++)
        for (long p = 0; p < hw; p++) {
            float acc = 0.f;
            for (int j = 0; j < 3; j++) acc = fmaf(T[i][j], ycc[j * hw + p] + off[j], acc);
            rgb[i * hw + p] = acc;
        }
}

/* F.interpolate(mode="area") = adaptive_avg_pool2d (lrf/compression/utils.py:92-94 called with
 * scale_factor from lrf/compression/qmf.py:230).  Window [floor(i*H/oh), ceil((i+1)*H/oh));
 * row-major fp32 sum, then / kh / kw (ATen AdaptiveAvgPoolKernel.cpp). */
void lrf_oracle_area_downsample(const float* in, long H, long W, long oh, long ow, float* out)
{
    for (long i = 0; i < oh; i++) {
        long h0 = (i * H) / oh, h1 = ((i + 1) * H + oh - 1) / oh;
        for (long j = 0; j < ow; j++) {
            long w0 = (j * W) / ow, w1 = ((j + 1) * W + ow - 1) / ow;
            float sum = 0.f;
            for (long a = h0; a < h1; a++)
                for (long b = w0; b < w1; b++) sum = sum + in[a * W + b];
            out[i * ow + j] = sum / (float)(h1 - h0) / (float)(w1 - w0);
        }
    }
}

/* F.interpolate(mode="nearest", size=...) — lrf/compression/utils.py:101-103 via qmf.py:346-348.
 * src = min(floor(dst * in/out), in-1) with the scale computed in fp32 (ATen nearest_idx). */
void lrf_oracle_nearest_upsample(const float* in, long h, long w, long H, long W, float* out)
{
    float sh = (float)h / (float)H, sw = (float)w / (float)W;
    for (long i = 0; i < H; i++) {
        long si = (long)floorf((float)i * sh);
        if (si > h - 1) si = h - 1;
        for (long j = 0; j < W; j++) {
            long sj = (long)floorf((float)j * sw);
            if (sj > w - 1) sj = w - 1;
            out[i * W + j] = in[si * w + sj];
        }
    }
}

static long reflect_idx(long i, long n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

/* pad_image(reflect) + patchify: lrf/compression/utils.py:108-132, lrf/compression/qmf.py:43-56.
 * in: C planes of H x W fp32 (CHW).  X: [(Hp/p)*(Wp/q)] x [C*p*q], "c (h p) (w q) -> (h w) (c p q)". */
void lrf_oracle_pad_patchify(const float* in, long C, long H, long W, long p, long q, float* X)
{
    long ph = (p - H % p) % p, pw = (q - W % q) % q;
    long top = ph / 2, left = pw / 2;
    long Hp = H + ph, Wp = W + pw, nh = Hp / p, nw = Wp / q, N = C * p * q;
    for (long hh = 0; hh < nh; hh++)
        for (long ww = 0; ww < nw; ww++)
            for (long c = 0; c < C; c++)
                for (long a = 0; a < p; a++)
                    for (long b = 0; b < q; b++) {
                        long y = reflect_idx(hh * p + a - top, H), x = reflect_idx(ww * q + b - left, W);
                        X[(hh * nw + ww) * N + (c * p + a) * q + b] = in[(c * H + y) * W + x];
                    }
}

/* depatchify + unpad_image: lrf/compression/qmf.py:59-75, lrf/compression/utils.py:135-153.
 * X: [(Hp/p)*(Wp/q)] x [C*p*q] -> out: C planes of H x W (centre crop, start = (Hp-H)//2). */
void lrf_oracle_depatchify_unpad(const float* X, long C, long Hp, long Wp, long H, long W, long p, long q, float* out)
{
    long nw = Wp / q, N = C * p * q;
    long sh = (Hp - H) / 2, sw = (Wp - W) / 2;
    for (long c = 0; c < C; c++)
        for (long y = 0; y < H; y++)
            for (long x = 0; x < W; x++) {
                long yy = y + sh, xx = x + sw;
                long hh = yy / p, a = yy % p, ww = xx / q, b = xx % q;
                out[(c * H + y) * W + x] = X[(hh * nw + ww) * N + (c * p + a) * q + b];
            }
}

/* QMF.reconstruct — lrf/factorization/qmf.py:216-223 with w=None: u @ v.mT (k-ordered fma chain;
 * exact for the integer factors of the QMF path). */
void lrf_oracle_reconstruct(const float* U, const float* V, long M, long N, int R, float* X)
{
    mm_torch(U, R, 1, V, 1, R, X, N, M, R, N);
}

/* to_dtype(uint8) — lrf/compression/utils.py:156-182: clamp(0,255) then a truncating cast. */
void lrf_oracle_to_u8(const float* in, long n, uint8_t* out)
{
    for (long i = 0; i < n; i++) {
        float v = in[i];
        if (v < 0.f) v = 0.f;
        if (v > 255.f) v = 255.f;
        out[i] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * whole-image helpers for the default branch of qmf_encode / qmf_decode
 * (color_space="YCbCr", patch=True): lrf/compression/qmf.py:214-262 and :323-351
 * ---------------------------------------------------------------------------------------------- */

/* sizes of plane c (0 = Y, 1/2 = chroma) for an H x W image, scale 0.5: lrf/compression/qmf.py:230 */
void lrf_oracle_plane_dims(long H, long W, long p, long q, int c, long* h, long* w, long* hp, long* wp, long* M)
{
    long ph = c ? (long)floor((double)H * 0.5) : H, pw = c ? (long)floor((double)W * 0.5) : W;
    *h = ph; *w = pw;
    *hp = ph + (p - ph % p) % p;
    *wp = pw + (q - pw % q) % q;
    *M = (*hp / p) * (*wp / q);
}

/* uint8 RGB (CHW) -> the three patch matrices X_Y, X_Cb, X_Cr (row-major [M_c, p*q]). */
int lrf_oracle_rgb_to_planes(const uint8_t* rgb, long H, long W, long p, long q, float* XY, float* XCb, float* XCr)
{
    float* ycc = (float*)malloc(sizeof(float) * 3 * H * W);
    if (!ycc) return -1;
    lrf_oracle_rgb_to_ycbcr(rgb, H, W, ycc);
    lrf_oracle_pad_patchify(ycc, 1, H, W, p, q, XY);
    long h, w, hp, wp, M;
    lrf_oracle_plane_dims(H, W, p, q, 1, &h, &w, &hp, &wp, &M);
    float* ds = (float*)malloc(sizeof(float) * h * w);
    float* Xc[2] = {XCb, XCr};
    for (int c = 0; c < 2; c++) {
        lrf_oracle_area_downsample(ycc + (c + 1) * H * W, H, W, h, w, ds);
        lrf_oracle_pad_patchify(ds, 1, h, w, p, q, Xc[c]);
    }
    free(ds);
    free(ycc);
    return 0;
}

/* int8 factors of the three planes -> uint8 RGB (CHW): lrf/compression/qmf.py:329-351. */
int lrf_oracle_planes_to_rgb(const int8_t* const U[3], const int8_t* const V[3], const int R[3],
                             long H, long W, long p, long q, uint8_t* rgb)
{
    long N = p * q;
    float* ycc = (float*)malloc(sizeof(float) * 3 * H * W);
    float* out = (float*)malloc(sizeof(float) * 3 * H * W);
    if (!ycc || !out) return -1;
    for (int c = 0; c < 3; c++) {
        long h, w, hp, wp, M;
        lrf_oracle_plane_dims(H, W, p, q, c, &h, &w, &hp, &wp, &M);
        float* uf = (float*)malloc(sizeof(float) * M * R[c]);
        float* vf = (float*)malloc(sizeof(float) * N * R[c]);
        float* X = (float*)malloc(sizeof(float) * M * N);
        float* pl = (float*)malloc(sizeof(float) * h * w);
        for (long i = 0; i < M * R[c]; i++) uf[i] = (float)U[c][i];
        for (long i = 0; i < N * R[c]; i++) vf[i] = (float)V[c][i];
        lrf_oracle_reconstruct(uf, vf, M, N, R[c], X);
        lrf_oracle_depatchify_unpad(X, 1, hp, wp, h, w, p, q, pl);
        if (c == 0) memcpy(ycc, pl, sizeof(float) * H * W);
        else lrf_oracle_nearest_upsample(pl, h, w, H, W, ycc + c * H * W);
        free(uf); free(vf); free(X); free(pl);
    }
    lrf_oracle_ycbcr_to_rgb(ycc, H, W, out);
    lrf_oracle_to_u8(out, 3 * H * W, rgb);
    free(ycc);
    free(out);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SVD baseline — lrf/compression/svd.py:179-187 and lrf/compression/utils.py:185-243
 * ---------------------------------------------------------------------------------------------- */

/* quantize(tensor, uint8): scale = (max-min)/(255-0); q = clamp((t-min)/scale + 0, 0, 255) truncated. */
void lrf_oracle_quantize_u8(const float* t, long n, uint8_t* qv, float* scale, float* minv)
{
    float mn = t[0], mx = t[0];
    for (long i = 1; i < n; i++) {
        if (t[i] < mn) mn = t[i];
        if (t[i] > mx) mx = t[i];
    }
    float sc = (mx - mn) / 255.f;
    for (long i = 0; i < n; i++) {
        float v = (t[i] - mn) / sc + 0.f;
        if (v < 0.f) v = 0.f;
        if (v > 255.f) v = 255.f;
        qv[i] = (uint8_t)v;
    }
    *scale = sc;
    *minv = mn;
}

/* dequantize: (q - q.min()) * scale + min_val  (utils.py:241; scale/min are Python floats there,
 * so the multiply-add runs in fp32 on the tensor with fp32-cast scalars). */
void lrf_oracle_dequantize_u8(const uint8_t* qv, long n, float scale, float minv, float* t)
{
    uint8_t qm = qv[0];
    for (long i = 1; i < n; i++)
        if (qv[i] < qm) qm = qv[i];
    for (long i = 0; i < n; i++) t[i] = ((float)qv[i] - (float)qm) * scale + minv;
}

/* ------------------------------------------------------------------------------------------------
 * SVD baseline, default branch of svd_encode (color_space="RGB", patch 8x8, uint8 factors):
 * lrf/compression/svd.py:156-193.  Tolerance-checked path: the top-R pairs come
 * from the fp64 Gram matrix and the cyclic Jacobi solver above (any even n), not from LAPACK.
 * ---------------------------------------------------------------------------------------------- */

/* u = U sqrt(s) [M,R], v = (sqrt(s) Vh)^T [N,R] (svd.py:179-183); sign as in lrf_oracle_init_from_gram. */
int lrf_oracle_svd_topr(const float* X, long M, long N, int R, const int8_t* sign, float* u, float* v)
{
    int n = (int)N;
    if (n & 1) return -1;
    double* G = (double*)malloc(sizeof(double) * n * n);
    double* E = (double*)malloc(sizeof(double) * n * n);
    float* w = (float*)malloc(sizeof(float) * n * R);
    int* order = (int*)malloc(sizeof(int) * n);
    lrf_oracle_gram_f64(X, M, N, G);
    lrf_oracle_jacobi_f64(G, n, E, 40);
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 1; i < n; i++) {
        int o = order[i];
        double key = G[o * n + o];
        int j = i - 1;
        while (j >= 0 && G[order[j] * n + order[j]] < key) { order[j + 1] = order[j]; j--; }
        order[j + 1] = o;
    }
    for (int r = 0; r < R; r++) {
        int c = order[r];
        double lam = G[c * n + c];
        double sr = sqrt(sqrt(lam > 1e-200 ? lam : 0.0));
        double dot = 0.0;
        for (int j = 0; j < n; j++) dot = fma((double)(j + 1), E[j * n + c], dot);
        double want = (sign && sign[r]) ? (double)sign[r] : -1.0;
        double flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.0 : -1.0;
        for (int j = 0; j < n; j++) {
            double ev = flip * E[j * n + c];
            v[j * R + r] = (float)(ev * sr);
            w[j * R + r] = (sr > 0.0) ? (float)(ev / sr) : 0.f;
        }
    }
    mm_mkl(X, N, 1, w, R, 1, u, R, M, N, R);
    free(G); free(E); free(w); free(order);
    return 0;
}

/* The same for any shape, by the cyclic Jacobi solver — a SECOND OPINION on lrf_oracle_svd_topr_any (lrf_oracle_any.c, the
 * restatement of the GPU's own eigen-solver): the two agree to fp32 accuracy, not bit for bit.  The eigen-problem is solved
 * on the SHORT side (n = min(M, N)); when M < N the roles of the factors swap (u = e sqrt(sigma), v = X^T e / sqrt(sigma))
 * and the column sign is imposed on the finished v, as include/lrf_hip.h states it.  Odd n: the Gram matrix is bordered
 * with a zero row / column (the Jacobi sweep pairs an even number of indices); the extra eigenvector is e_n with
 * eigenvalue 0 and contributes zero columns at most. */
int lrf_oracle_svd_topr_any_jacobi(const float* X, long M, long N, int R, const int8_t* sign, float* u, float* v)
{
    const int tall = N <= M;
    const long n = tall ? N : M, D = tall ? M : N;
    float* Xt = NULL;
    const float* A = X; /* D x n, row-major */
    if (!tall) {
        Xt = (float*)malloc(sizeof(float) * M * N);
        for (long i = 0; i < M; i++)
            for (long j = 0; j < N; j++) Xt[j * M + i] = X[i * N + j];
        A = Xt;
    }
    const int ne = (int)(n + (n & 1));
    double* G0 = (double*)malloc(sizeof(double) * n * n);
    double* G = (double*)calloc((size_t)ne * ne, sizeof(double));
    double* E = (double*)malloc(sizeof(double) * ne * ne);
    float* e1 = (float*)calloc((size_t)n * R, sizeof(float));
    float* e2 = (float*)calloc((size_t)n * R, sizeof(float));
    int* order = (int*)malloc(sizeof(int) * ne);
    lrf_oracle_gram_f64(A, D, n, G0);
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++) G[i * ne + j] = G0[i * n + j];
    lrf_oracle_jacobi_f64(G, ne, E, 60);
    for (int i = 0; i < ne; i++) order[i] = i;
    for (int i = 1; i < ne; i++) {
        int o = order[i];
        double key = G[o * ne + o];
        int j = i - 1;
        while (j >= 0 && G[order[j] * ne + order[j]] < key) { order[j + 1] = order[j]; j--; }
        order[j + 1] = o;
    }
    for (int r = 0; r < R && r < n; r++) {
        int c = order[r];
        double lam = G[c * ne + c];
        double sr = sqrt(sqrt(lam > 1e-200 ? lam : 0.0));
        double flip = 1.0;
        if (tall) {
            double dot = 0.0;
            for (long j = 0; j < n; j++) dot = fma((double)(j + 1), E[j * ne + c], dot);
            double want = (sign && sign[r]) ? (double)sign[r] : -1.0;
            flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.0 : -1.0;
        }
        for (long j = 0; j < n; j++) {
            double ev = flip * E[j * ne + c];
            e1[j * R + r] = (float)(ev * sr);
            e2[j * R + r] = (sr > 0.0) ? (float)(ev / sr) : 0.f;
        }
    }
    if (tall) {
        memcpy(v, e1, sizeof(float) * n * R);
        mm_mkl(X, N, 1, e2, R, 1, u, R, M, N, R);
    } else {
        memcpy(u, e1, sizeof(float) * n * R);
        mm_mkl(X, 1, N, e2, R, 1, v, R, N, M, R);
        for (int r = 0; r < R; r++) {
            double dot = 0.0;
            for (long j = 0; j < N; j++) dot = fma((double)(j + 1), (double)v[j * R + r], dot);
            double want = (sign && sign[r]) ? (double)sign[r] : -1.0;
            if ((dot < 0.0 ? -1.0 : 1.0) != want) {
                for (long j = 0; j < N; j++) v[j * R + r] = -v[j * R + r];
                for (long i = 0; i < M; i++) u[i * R + r] = -u[i * R + r];
            }
        }
    }
    free(Xt); free(G0); free(G); free(E); free(e1); free(e2); free(order);
    return 0;
}

/* svd_decode arithmetic for the RGB branch (svd.py:310-326): dequantize, u @ v.mT, depatchify, unpad, to_dtype. */
int lrf_oracle_svd_decode_rgb(const uint8_t* qu, const uint8_t* qv, long M, int R, float su, float mu, float sv, float mv,
                              long H, long W, uint8_t* rgb)
{
    long N = 192, Hp = H + (8 - H % 8) % 8, Wp = W + (8 - W % 8) % 8;
    float* u = (float*)malloc(sizeof(float) * M * R);
    float* v = (float*)malloc(sizeof(float) * N * R);
    float* X = (float*)malloc(sizeof(float) * M * N);
    float* img = (float*)malloc(sizeof(float) * 3 * H * W);
    lrf_oracle_dequantize_u8(qu, M * R, su, mu, u);
    lrf_oracle_dequantize_u8(qv, N * R, sv, mv, v);
    lrf_oracle_reconstruct(u, v, M, N, R, X);
    lrf_oracle_depatchify_unpad(X, 3, Hp, Wp, H, W, 8, 8, img);
    lrf_oracle_to_u8(img, 3 * H * W, rgb);
    free(u); free(v); free(X); free(img);
    return 0;
}

#include "lrf_oracle_any.c"
