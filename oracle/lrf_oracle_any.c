/* lrf_oracle_any.c — CPU restatement of the any-shape SVD initialisation of liblrf_hip.so, operation for operation
 * (TEST INFRASTRUCTURE ONLY; included at the end of lrf_oracle.c, whose helpers it uses).
 *
 * What it follows.  The reference computes torch.linalg.svd (LAPACK sgesdd) at lrf/factorization/qmf.py:42-48 and
 * lrf/compression/svd.py:179-183; the GPU library replaces it by this project's own deterministic algorithm (DESIGN.md
 * section 2), so there is no reference arithmetic to restate here: this file DEFINES that algorithm on the CPU and the
 * kernels of lrf_amd/csrc/lrf_anyshape_kernels.hip (k_any_gram, k_any_tridiag_reg<NC>, k_any_eig<NCT>, k_any_signfix) and
 * lrf_svd_kernels.hip (k_gram192_u8 + k_gram192_fold) must produce the same bits.  The LAPACK result is reached to fp32
 * accuracy (tests compare against lrf_oracle_svd_topr_any_jacobi and against reference fixtures by tolerance).
 *
 * Every reduction below is written in the order of the kernel it mirrors:
 *   tree64(s)       wave butterfly (wave_sum / wave_tree64): for off = 32..1: s[i] += s[i + off]
 *   block_sum256(s) four wave trees, then ((p0 + p1) + p2) + p3
 *   "thread t owns columns t, t + 256, ..." partial sums, "lane l owns i = l, l + 64, ..." partial sums
 * Tridiagonalisations, chosen by the side n = min(M, N) exactly as lrf_anyshape_host.inc does:
 *   n <= 64                     k_any_eig<1>, plain three-pass Householder step              (any_tridiag_plain)
 *   64 < n <= 192               k_any_tridiag_reg<2 or 3>, matrix in registers                (any_tridiag_reg)
 *   192 < n <= 512              k_any_tridiag_sym<1,8> / <2,4>, panels of 8 steps, lower triangle (any_tridiag_sym)
 *   n > 512                     k_any_tridiag_blk<4|8>, panels of 8 / 4 steps, full square     (any_tridiag_blocked)
 * (LRF_ORACLE_ANY_TRIDIAG=blocked: the full-square panels also for 193..512 — the library's LRF_ANY_TRIDIAG_SQUARE=1;
 *  =unblocked: round 2's any_tridiag_plain / any_tridiag_fused above 192 — the library's LRF_ANY_TRIDIAG_UNBLOCKED=1.)
 */

static double tree64c(const double* v)
{
    double s[64];
    memcpy(s, v, sizeof(s));
    return tree64(s);
}

/* per-thread values of a 256-thread workgroup -> block_sum (lrf_svd_kernels.hip) */
static double block_sum256(const double* tv)
{
    const double p0 = tree64c(tv), p1 = tree64c(tv + 64), p2 = tree64c(tv + 128), p3 = tree64c(tv + 192);
    return ((p0 + p1) + p2) + p3;
}

/* k_any_gram: G[i][j] = sum_k A(k,i) A(k,j) as one fp64 fma chain per element, k ascending; A(k,i) = X[k*sgk + i*sgi] */
static void any_gram(const float* X, long sgk, long sgi, long n, long D, double* G)
{
    for (long i = 0; i < n; i++)
        for (long j = i; j < n; j++) {
            double acc = 0.0;
            for (long k = 0; k < D; k++) acc = fma((double)X[k * sgk + i * sgi], (double)X[k * sgk + j * sgi], acc);
            G[i * n + j] = acc;
            G[j * n + i] = acc;
        }
}

/* k_gram192_u8 + k_gram192_fold: X holds uint8-valued floats; the Gram matrix is an exact integer (< 2^53) */
void lrf_oracle_gram_u8_exact(const float* X, long M, long N, double* G)
{
    for (long i = 0; i < N; i++)
        for (long j = i; j < N; j++) {
            int64_t acc = 0;
            for (long k = 0; k < M; k++) acc += (int64_t)X[k * N + i] * (int64_t)X[k * N + j];
            G[i * N + j] = (double)acc;
            G[j * N + i] = (double)acc;
        }
}

/* ---- k_any_eig<NCT>, NCT == 1 branch: plain Householder step, thread t owns column t (n <= 256).
 * A: n x n, row k keeps v_k (columns > k) afterwards; d, e, tau: n doubles each. */
static void any_tridiag_plain(double* A, int n, double* d, double* e, double* tau)
{
    double tv[256], *v = (double*)malloc(sizeof(double) * n), *w = (double*)malloc(sizeof(double) * n),
                    *cc = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) { e[i] = 0.0; tau[i] = 0.0; }
    for (int k = 0; k < n - 2; k++) {
        for (int t = 0; t < 256; t++) {
            double s = 0.0;
            if (t < n && t > k) { const double x = A[(long)k * n + t]; s = fma(x, x, s); }
            tv[t] = s;
        }
        const double sigma = block_sum256(tv);
        if (!(sigma > LRF_SIGMA_TINY)) { tau[k] = 0.0; e[k] = 0.0; continue; }
        const double x0 = A[(long)k * n + k + 1];
        const double nrm = sqrt(sigma);
        const double alpha = (x0 >= 0.0) ? -nrm : nrm;
        for (int t = 0; t < 256; t++) {
            double vv = 0.0;
            if (t < n) {
                if (t > k + 1) vv = A[(long)k * n + t];
                else if (t == k + 1) vv = x0 - alpha;
                v[t] = vv;
                if (t > k) A[(long)k * n + t] = vv;
            }
            tv[t] = fma(vv, vv, 0.0);
        }
        const double vn = block_sum256(tv);
        const double t_ = 2.0 / vn;
        tau[k] = t_;
        e[k] = alpha;
        for (int i = 0; i < n; i++) {
            double c = 0.0;
            if (i > k)
                for (int j = k + 1; j < n; j++) c = fma(A[(long)j * n + i], v[j], c);
            cc[i] = c;
        }
        for (int t = 0; t < 256; t++) {
            double s = 0.0;
            if (t < n) {
                cc[t] = t_ * cc[t];
                s = fma(cc[t], v[t], s);
            }
            tv[t] = s;
        }
        const double Kc = (0.5 * t_) * block_sum256(tv);
        for (int i = 0; i < n; i++) w[i] = fma(-Kc, v[i], cc[i]);
        for (int r = k + 1; r < n; r++)
            for (int i = k + 1; i < n; i++) { /* canonical (row >= column) operand order: exactly symmetric */
                const int rc = r >= i;
                const double va = rc ? v[r] : v[i], wa = rc ? w[r] : w[i], vb = rc ? v[i] : v[r], wb = rc ? w[i] : w[r];
                A[(long)r * n + i] = fma(-wa, vb, fma(-va, wb, A[(long)r * n + i]));
            }
    }
    for (int i = 0; i < n; i++) d[i] = A[(long)i * n + i];
    if (n >= 2) { e[n - 2] = A[(long)(n - 1) * n + n - 2]; tau[n - 2] = 0.0; }
    e[n - 1] = 0.0;
    tau[n - 1] = 0.0;
    free(v); free(w); free(cc);
}

/* ---- k_any_tridiag_reg<NC> (64 < n <= 64 NC, NC = 2, 3): 4 NC waves; wave (cc, rg) holds rows 16 NC rg .. of the columns
 * 64 cc .. 64 cc + 63.  Non-symmetric two-fma update, so the FULL matrix is carried. */
static void any_tridiag_reg(double* A, int n, int NC, double* d, double* e, double* tau)
{
    const int NP = 64 * NC, RPT = 16 * NC;
    double* M = (double*)calloc((size_t)NP * NP, sizeof(double)); /* padded copy: zeros outside n x n */
    double *xs = (double*)malloc(sizeof(double) * NP), *vk = (double*)malloc(sizeof(double) * NP),
           *wk = (double*)malloc(sizeof(double) * NP), *pk = (double*)malloc(sizeof(double) * NP),
           *cpart = (double*)malloc(sizeof(double) * 4 * NP);
    double lanes[64];
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) M[(long)r * NP + c] = A[(long)r * n + c];
    for (int i = 0; i < n; i++) { e[i] = 0.0; tau[i] = 0.0; }
    for (int k = 0; k < n - 2; k++) {
        for (int c = 0; c < NP; c++) xs[c] = (c > k) ? M[(long)k * NP + c] : 0.0;
        for (int l = 0; l < 64; l++) {
            double sq = 0.0;
            for (int c2 = 0; c2 < NC; c2++) sq = fma(xs[64 * c2 + l], xs[64 * c2 + l], sq);
            lanes[l] = sq;
        }
        const double sigma = tree64c(lanes);
        if (!(sigma > LRF_SIGMA_TINY)) { e[k] = 0.0; tau[k] = 0.0; continue; }
        const double x0 = xs[k + 1];
        const double nrm = sqrt(sigma);
        const double alpha = (x0 >= 0.0) ? -nrm : nrm;
        const double vfix = x0 - alpha;
        const double t = 1.0 / fma(fabs(x0), nrm, sigma);
        for (int c = 0; c < NP; c++) vk[c] = (c == k + 1) ? vfix : xs[c];
        for (int c = k + 1; c < n; c++) A[(long)k * n + c] = vk[c]; /* v_k for the back-transformation */
        /* matvec partials: thread (col, rg): one chain per 16-row sub-block of its row group, the sub-block sums added in order */
        for (int rg = 0; rg < 4; rg++)
            for (int col = 0; col < NP; col++) {
                double cs = 0.0;
                const int live = 64 * (col >> 6) + 63 > k;
                for (int s = 0; s < NC; s++) {
                    const int r0 = rg * RPT + 16 * s;
                    double c = 0.0;
                    if (live && r0 + 15 > k)
                        for (int jj = 0; jj < 16; jj++) c = fma(M[(long)(r0 + jj) * NP + col], vk[r0 + jj], c);
                    cs = s ? cs + c : c;
                }
                cpart[rg * NP + col] = (col > k) ? cs : 0.0;
            }
        for (int l = 0; l < 64; l++) {
            double s2 = 0.0;
            for (int c2 = 0; c2 < NC; c2++) {
                const int i = 64 * c2 + l;
                pk[i] = t * (((cpart[i] + cpart[NP + i]) + cpart[2 * NP + i]) + cpart[3 * NP + i]);
                s2 = fma(pk[i], vk[i], s2);
            }
            lanes[l] = s2;
        }
        const double K = (0.5 * t) * tree64c(lanes);
        for (int c = 0; c < NP; c++) wk[c] = fma(-K, vk[c], pk[c]);
        e[k] = alpha;
        tau[k] = t;
        for (int rg = 0; rg < 4; rg++)
            for (int s = 0; s < NC; s++) {
                const int r0 = rg * RPT + 16 * s;
                if (!(r0 + 15 > k)) continue;
                for (int col = 0; col < NP; col++) {
                    if (!(64 * (col >> 6) + 63 > k)) continue;
                    for (int jj = 0; jj < 16; jj++) {
                        double* a = &M[(long)(r0 + jj) * NP + col];
                        *a = fma(-vk[r0 + jj], wk[col], fma(-wk[r0 + jj], vk[col], *a));
                    }
                }
            }
    }
    for (int i = 0; i < n; i++) d[i] = M[(long)i * NP + i];
    e[n - 2] = M[(long)(n - 1) * NP + n - 2];
    e[n - 1] = 0.0;
    tau[n - 2] = 0.0;
    tau[n - 1] = 0.0;
    free(M); free(xs); free(vk); free(wk); free(pk); free(cpart);
}

/* ---- k_any_eig<NCT>, NCT > 1 branch (n > 256): thread t owns the columns t, t + 256, ...; the rank-2 update of step k also
 * accumulates the matrix-vector product of step k + 1 (row k + 1 first, then the rows below, each adding A[r][i] v_{k+1}[r]). */
static void any_tridiag_fused(double* A, int n, int NCT, double* d, double* e, double* tau)
{
    double tv[256];
    double *Lvk = (double*)calloc(n, sizeof(double)), *Lvn = (double*)calloc(n, sizeof(double)), *w = (double*)malloc(sizeof(double) * n),
           *cc = (double*)calloc(n, sizeof(double)), *cc2 = (double*)calloc(n, sizeof(double)), *rowv = (double*)malloc(sizeof(double) * n);
    double t = 0.0;
    int have = 0;
    for (int i = 0; i < n; i++) { e[i] = 0.0; tau[i] = 0.0; }
    for (int k = 0; k < n - 2; k++) {
        /* (the kernel swaps two LDS buffers by the parity of k; here Lvk always is this step's reflector) */
        if (!have) {
            for (int tt = 0; tt < 256; tt++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    if (i < n && i > k) { const double x = A[(long)k * n + i]; s = fma(x, x, s); }
                }
                tv[tt] = s;
            }
            const double sigma = block_sum256(tv);
            if (!(sigma > LRF_SIGMA_TINY)) { tau[k] = 0.0; e[k] = 0.0; continue; }
            const double x0 = A[(long)k * n + k + 1];
            const double nrm = sqrt(sigma);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            for (int tt = 0; tt < 256; tt++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    double vv = 0.0;
                    if (i < n) {
                        if (i > k + 1) vv = A[(long)k * n + i];
                        else if (i == k + 1) vv = x0 - alpha;
                        Lvk[i] = vv;
                        if (i > k) A[(long)k * n + i] = vv;
                    }
                    s = fma(vv, vv, s);
                }
                tv[tt] = s;
            }
            const double vn = block_sum256(tv);
            t = 2.0 / vn;
            tau[k] = t;
            e[k] = alpha;
            for (int i = 0; i < n; i++) {
                double c = 0.0;
                if (i > k)
                    for (int j = k + 1; j < n; j++) c = fma(A[(long)j * n + i], Lvk[j], c);
                cc[i] = c;
            }
        }
        /* w_k = t A v - (t/2 (t A v . v)) v */
        for (int i = 0; i < n; i++)
            if (!(i > k)) cc[i] = 0.0;
        for (int tt = 0; tt < 256; tt++) {
            double s = 0.0;
            for (int c = 0; c < NCT; c++) {
                const int i = tt + 256 * c;
                if (i < n) {
                    cc[i] = t * cc[i];
                    s = fma(cc[i], Lvk[i], s);
                } /* (columns past n hold zeros in the kernel: fma(0, 0, s) = s) */
            }
            tv[tt] = s;
        }
        const double Kc = (0.5 * t) * block_sum256(tv);
        for (int i = 0; i < n; i++) w[i] = fma(-Kc, Lvk[i], cc[i]);
#define ANY_UPD(r, i, a) (((r) >= (i)) ? fma(-w[r], Lvk[i], fma(-Lvk[r], w[i], (a))) : fma(-w[i], Lvk[r], fma(-Lvk[i], w[r], (a))))
        /* row k + 1 first: it carries the next reflector */
        int next = 0;
        double t2 = 0.0;
        for (int i = 0; i < n; i++) { cc2[i] = 0.0; Lvn[i] = 0.0; }
        {
            const int r = k + 1;
            double x0n = 0.0;
            for (int tt = 0; tt < 256; tt++) {
                double s2 = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    if (i < n) rowv[i] = 0.0;
                    if (i < n && i > k) {
                        rowv[i] = ANY_UPD(r, i, A[(long)r * n + i]);
                        A[(long)r * n + i] = rowv[i];
                        if (i > r) s2 = fma(rowv[i], rowv[i], s2);
                        if (i == r + 1) x0n = rowv[i];
                    }
                }
                tv[tt] = s2;
            }
            if (k + 1 < n - 2) {
                const double sigma2 = block_sum256(tv);
                if (sigma2 > LRF_SIGMA_TINY) {
                    next = 1;
                    const double nrm = sqrt(sigma2);
                    const double alpha = (x0n >= 0.0) ? -nrm : nrm;
                    for (int tt = 0; tt < 256; tt++) {
                        double s3 = 0.0;
                        for (int c = 0; c < NCT; c++) {
                            const int i = tt + 256 * c;
                            double vv = 0.0;
                            if (i < n) {
                                if (i > r + 1) vv = rowv[i];
                                else if (i == r + 1) vv = x0n - alpha;
                                Lvn[i] = vv;
                                if (i > r) A[(long)r * n + i] = vv;
                            }
                            s3 = fma(vv, vv, s3);
                        }
                        tv[tt] = s3;
                    }
                    const double vn = block_sum256(tv);
                    t2 = 2.0 / vn;
                    tau[r] = t2;
                    e[r] = alpha;
                }
            }
        }
        /* the remaining rows: update, and (when there is a next reflector) its matrix-vector product on the fly */
        for (int r = k + 2; r < n; r++) {
            const double vn_r = next ? Lvn[r] : 0.0;
            for (int i = k + 1; i < n; i++) {
                const double nv = ANY_UPD(r, i, A[(long)r * n + i]);
                A[(long)r * n + i] = nv;
                cc2[i] = fma(nv, vn_r, cc2[i]);
            }
        }
#undef ANY_UPD
        have = next;
        if (next) {
            t = t2;
            for (int i = 0; i < n; i++) { Lvk[i] = Lvn[i]; cc[i] = cc2[i]; }
        }
    }
    for (int i = 0; i < n; i++) d[i] = A[(long)i * n + i];
    if (n >= 2) { e[n - 2] = A[(long)(n - 1) * n + n - 2]; tau[n - 2] = 0.0; }
    e[n - 1] = 0.0;
    tau[n - 1] = 0.0;
    free(Lvk); free(Lvn); free(w); free(cc); free(cc2); free(rowv);
}

/* ---- k_any_tridiag_blk<NCT> (n > 192): panels of NB = 16 (n <= 512), 8 (n <= 1024), 4 Householder steps (LAPACK's dlatrd shape).  Inside a panel
 * the trailing matrix in memory stays as it was at the panel's start (A0); step k forms its row from A0 and the panel's
 * reflectors, x = A0[k][.] - sum_m (V_m[k] W_m + W_m[k] V_m), its product as A0 v - sum_m (V_m (W_m . v) + W_m (V_m . v)),
 * and the rank-2 NB update is applied once per panel (two fmas per pair, not symmetrised).  Thread t owns the columns
 * t, t + 256, ...; every thread only ever touches its own columns of A.  t = 1 / (sigma + |x0| nrm) as in k_any_tridiag_reg. */
static void any_tridiag_blocked(double* A, int n, int NCT, int NB, double* d, double* e, double* tau)
{
    double tv[256], tg[256], th[256], g[16], h[16];
    double *Vp = (double*)calloc((size_t)NB * n, sizeof(double)), *Wp = (double*)calloc((size_t)NB * n, sizeof(double)),
           *x = (double*)malloc(sizeof(double) * n), *v = (double*)malloc(sizeof(double) * n), *p = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) { d[i] = 0.0; e[i] = 0.0; tau[i] = 0.0; }
    for (int k0 = 0; k0 < n - 2; k0 += NB) {
        const int np = (n - 2 - k0 < NB) ? n - 2 - k0 : NB;
        for (int j = 0; j < np; j++) {
            const int k = k0 + j;
            for (int i = 0; i < n; i++) { /* the current row k (and the diagonal element) */
                double xx = 0.0;
                if (i >= k) {
                    xx = A[(long)k * n + i];
                    for (int m = 0; m < j; m++) {
                        xx = fma(-Vp[(long)m * n + k], Wp[(long)m * n + i], xx);
                        xx = fma(-Wp[(long)m * n + k], Vp[(long)m * n + i], xx);
                    }
                }
                x[i] = xx;
            }
            d[k] = x[k];
            x[k] = 0.0;
            for (int tt = 0; tt < 256; tt++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    if (i < n) s = fma(x[i], x[i], s);
                }
                tv[tt] = s;
            }
            const double sigma = block_sum256(tv);
            if (!(sigma > LRF_SIGMA_TINY)) {
                tau[k] = 0.0;
                e[k] = 0.0;
                for (int i = 0; i < n; i++) { Vp[(long)j * n + i] = 0.0; Wp[(long)j * n + i] = 0.0; }
                continue;
            }
            const double x0 = x[k + 1];
            const double nrm = sqrt(sigma);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            const double t = 1.0 / fma(fabs(x0), nrm, sigma);
            for (int i = 0; i < n; i++) {
                v[i] = (i == k + 1) ? x0 - alpha : x[i];
                Vp[(long)j * n + i] = v[i];
                if (i > k) A[(long)k * n + i] = v[i];
            }
            tau[k] = t;
            e[k] = alpha;
            for (int m = 0; m < j; m++) { /* g_m = W_m . v, h_m = V_m . v */
                for (int tt = 0; tt < 256; tt++) {
                    double sg = 0.0, sh = 0.0;
                    for (int c = 0; c < NCT; c++) {
                        const int i = tt + 256 * c;
                        if (i < n) {
                            sg = fma(Wp[(long)m * n + i], v[i], sg);
                            sh = fma(Vp[(long)m * n + i], v[i], sh);
                        }
                    }
                    tg[tt] = sg;
                    th[tt] = sh;
                }
                g[m] = block_sum256(tg);
                h[m] = block_sum256(th);
            }
            for (int i = 0; i < n; i++) {
                double c = 0.0;
                if (i > k) {
                    for (int r = k + 1; r < n; r++) c = fma(A[(long)r * n + i], v[r], c);
                    for (int m = 0; m < j; m++) {
                        c = fma(-Vp[(long)m * n + i], g[m], c);
                        c = fma(-Wp[(long)m * n + i], h[m], c);
                    }
                }
                p[i] = c;
            }
            for (int tt = 0; tt < 256; tt++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    if (i < n) {
                        p[i] = t * p[i];
                        s = fma(p[i], v[i], s);
                    }
                }
                tv[tt] = s;
            }
            const double Kc = (0.5 * t) * block_sum256(tv);
            for (int i = 0; i < n; i++) Wp[(long)j * n + i] = fma(-Kc, v[i], p[i]);
        }
        const int kend = k0 + np;
        for (int r = kend; r < n; r++)
            for (int i = kend; i < n; i++) {
                double a = A[(long)r * n + i];
                for (int m = 0; m < np; m++) {
                    a = fma(-Vp[(long)m * n + r], Wp[(long)m * n + i], a);
                    a = fma(-Wp[(long)m * n + r], Vp[(long)m * n + i], a);
                }
                A[(long)r * n + i] = a;
            }
    }
    if (n >= 2) {
        d[n - 2] = A[(long)(n - 2) * n + n - 2];
        e[n - 2] = A[(long)(n - 1) * n + n - 2];
    }
    d[n - 1] = A[(long)(n - 1) * n + n - 1];
    if (n == 1) d[0] = A[0];
    free(Vp); free(Wp); free(x); free(v); free(p);
}

/* ---- k_any_tridiag_sym<NCT> (192 < n <= 512): the blocked algorithm above on the LOWER TRIANGLE only.  A symmetric matrix
 * read in full moves every element twice per product; here an element A0[r][i] (r >= i) is read once and used twice — for
 * the column part cc[i] += A0[r][i] v[r] (chains over sub-tiles of sixteen rows, dealt to the four waves, below) and for
 * the row part of row r, the sum over the columns i < r of A0[r][i] v[i]: per 64-column chunk J four quarter sums (sixteen
 * products each, added in column order to 0.0), ((q0 + q1) + q2) + q3, and the chunks' values added in chunk order.
 * p = cc + rowpart, then dlatrd's corrections as in any_tridiag_blocked.  Row k of the current matrix is column k of the
 * lower triangle.  The panel update touches the lower triangle only; the upper triangle keeps the reflectors. */
static void any_tridiag_sym(double* A, int n, int NCT, int NB, int NW, double* d, double* e, double* tau)
{
    double tv[256], tg[256], th[256], g[16], h[16];
    double *Vp = (double*)calloc((size_t)NB * n, sizeof(double)), *Wp = (double*)calloc((size_t)NB * n, sizeof(double)),
           *x = (double*)malloc(sizeof(double) * n), *v = (double*)malloc(sizeof(double) * n), *p = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) { d[i] = 0.0; e[i] = 0.0; tau[i] = 0.0; }
    for (int k0 = 0; k0 < n - 2; k0 += NB) {
        const int np = (n - 2 - k0 < NB) ? n - 2 - k0 : NB;
        for (int j = 0; j < np; j++) {
            const int k = k0 + j;
            for (int i = 0; i < n; i++) { /* row k of the current matrix = column k of the lower triangle */
                double xx = 0.0;
                if (i >= k) {
                    xx = A[(long)i * n + k];
                    for (int m = 0; m < j; m++) {
                        xx = fma(-Vp[(long)m * n + k], Wp[(long)m * n + i], xx);
                        xx = fma(-Wp[(long)m * n + k], Vp[(long)m * n + i], xx);
                    }
                }
                x[i] = xx;
            }
            d[k] = x[k];
            x[k] = 0.0;
            for (int tt = 0; tt < 256; tt++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    if (i < n) s = fma(x[i], x[i], s);
                }
                tv[tt] = s;
            }
            const double sigma = block_sum256(tv);
            if (!(sigma > LRF_SIGMA_TINY)) {
                tau[k] = 0.0;
                e[k] = 0.0;
                for (int i = 0; i < n; i++) { Vp[(long)j * n + i] = 0.0; Wp[(long)j * n + i] = 0.0; }
                continue;
            }
            const double x0 = x[k + 1];
            const double nrm = sqrt(sigma);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            const double t = 1.0 / fma(fabs(x0), nrm, sigma);
            for (int i = 0; i < n; i++) {
                v[i] = (i == k + 1) ? x0 - alpha : x[i];
                Vp[(long)j * n + i] = v[i];
                if (i > k) A[(long)k * n + i] = v[i]; /* upper triangle: the reflector for the back-transformation */
            }
            tau[k] = t;
            e[k] = alpha;
            for (int m = 0; m < j; m++) {
                for (int tt = 0; tt < 256; tt++) {
                    double sg = 0.0, sh = 0.0;
                    for (int c = 0; c < NCT; c++) {
                        const int i = tt + 256 * c;
                        if (i < n) {
                            sg = fma(Wp[(long)m * n + i], v[i], sg);
                            sh = fma(Vp[(long)m * n + i], v[i], sh);
                        }
                    }
                    tg[tt] = sg;
                    th[tt] = sh;
                }
                g[m] = block_sum256(tg);
                h[m] = block_sum256(th);
            }
            const int Jmin = (k + 1) / 64;
            for (int i = 0; i < n; i++) {
                double c = 0.0;
                if (i > k) {
                    /* column part: rows r >= i (the diagonal included).  The rows are dealt to the four waves in sub-tiles of
                     * sixteen (sub-tile r0 / 16 to wave (r0 / 16) % NW — eight waves for n <= 256, four above —, from the sub-tile
                     * that holds row k + 1, inside the chunks at or below the diagonal one): per sub-tile a chain from 0.0, a
                     * wave's sub-tiles added in order, then the waves' sums in wave order */
                    double sw[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                    const int rbeg = (k + 1) & ~15, rdiag = 64 * (i / 64);
                    for (int r0 = (rbeg > rdiag ? rbeg : rdiag); r0 < n; r0 += 16) {
                        double cl = 0.0;
                        for (int u = 0; u < 16; u++) {
                            const int r = r0 + u;
                            const double av = (r < n && r >= i) ? A[(long)r * n + i] : 0.0, vj = (r < n) ? v[r] : 0.0;
                            cl = fma(av, vj, cl);
                        }
                        sw[(r0 >> 4) & (NW - 1)] = sw[(r0 >> 4) & (NW - 1)] + cl;
                    }
                    c = sw[0];
                    for (int w = 1; w < NW; w++) c = c + sw[w];
                    /* row part of row i: chunks Jmin .. i / 64 */
                    double yr = 0.0;
                    for (int J = Jmin; J <= i / 64; J++) {
                        double q[4];
                        for (int qq = 0; qq < 4; qq++) {
                            double sq = 0.0;
                            for (int m = 0; m < 16; m++) {
                                const int cc_ = 64 * J + 16 * qq + m;
                                const double pr = (cc_ < i && cc_ > k && cc_ < n) ? A[(long)i * n + cc_] * v[cc_] : 0.0;
                                sq = sq + pr;
                            }
                            q[qq] = sq;
                        }
                        yr = yr + (((q[0] + q[1]) + q[2]) + q[3]);
                    }
                    c = c + yr;
                    for (int m = 0; m < j; m++) {
                        c = fma(-Vp[(long)m * n + i], g[m], c);
                        c = fma(-Wp[(long)m * n + i], h[m], c);
                    }
                }
                p[i] = c;
            }
            for (int tt = 0; tt < 256; tt++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) {
                    const int i = tt + 256 * c;
                    if (i < n) {
                        p[i] = t * p[i];
                        s = fma(p[i], v[i], s);
                    }
                }
                tv[tt] = s;
            }
            const double Kc = (0.5 * t) * block_sum256(tv);
            for (int i = 0; i < n; i++) Wp[(long)j * n + i] = fma(-Kc, v[i], p[i]);
        }
        const int kend = k0 + np;
        for (int r = kend; r < n; r++)
            for (int i = kend; i <= r; i++) { /* lower triangle only */
                double a = A[(long)r * n + i];
                for (int m = 0; m < np; m++) {
                    a = fma(-Vp[(long)m * n + r], Wp[(long)m * n + i], a);
                    a = fma(-Wp[(long)m * n + r], Vp[(long)m * n + i], a);
                }
                A[(long)r * n + i] = a;
            }
    }
    if (n >= 2) {
        d[n - 2] = A[(long)(n - 2) * n + n - 2];
        e[n - 2] = A[(long)(n - 1) * n + n - 2];
    }
    d[n - 1] = A[(long)(n - 1) * n + n - 1];
    free(Vp); free(Wp); free(x); free(v); free(p);
}

/* Sturm count of k_any_eig / k_init for a side n (sturm_count above is the n = 64 case): de[j] = (d'_j, e'_{j-1}^2) */
static int any_sturm_count(const double* ds, const double* e2s, int n, double x)
{
    double p = 1.0, pp = 0.0;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        double pn = fma(ds[i] - x, p, -(e2s[i] * pp));
        if (pn == 0.0) pn = signbit(p) ? 0x1p-200 : -0x1p-200;
        cnt += (signbit(pn) != 0) != (signbit(p) != 0);
        pp = p;
        p = pn;
        if ((i & 7) == 7) {
            int ea, eb;
            frexp(p, &ea);
            frexp(pp, &eb);
            const int m = ea > eb ? ea : eb;
            p = ldexp(p, -m);
            pp = ldexp(pp, -m);
        }
    }
    return cnt;
}

/* Top-R eigen-pairs of the n x n Gram matrix G (destroyed) -> E1 = e sqrt(sigma), E2 = e / sqrt(sigma), fp32 [n][R]:
 * the tridiagonalisation the host code picks for this n, then k_any_eig's stages.  rcap: rank of the matrix at most this. */
int lrf_oracle_any_eig(double* G, int n, int R, int rcap, const int8_t* sign, float* E1, float* E2)
{
    const int NCT = n <= 256 ? 1 : (n <= 512 ? 2 : (n <= 1024 ? 4 : 8));
    const int Rn = R < n ? R : n, Rc = Rn < rcap ? Rn : rcap;
    double *d = (double*)malloc(sizeof(double) * n), *e = (double*)malloc(sizeof(double) * n), *tau = (double*)malloc(sizeof(double) * n),
           *e2 = (double*)malloc(sizeof(double) * n), *ds = (double*)malloc(sizeof(double) * n), *e2s = (double*)malloc(sizeof(double) * n),
           *lam = (double*)malloc(sizeof(double) * (Rc > 0 ? Rc : 1)), *Z = (double*)calloc((size_t)(Rc > 0 ? Rc : 1) * n, sizeof(double)),
           *Dp = (double*)malloc(sizeof(double) * n), *Dm = (double*)malloc(sizeof(double) * n), *x = (double*)malloc(sizeof(double) * n),
           *cf = (double*)malloc(sizeof(double) * (Rc > 0 ? Rc : 1));
    double tv[256], lanes[64];
    const char* tdv = getenv("LRF_ORACLE_ANY_TRIDIAG"); /* developer aid: "unblocked" = the round-2 variants above n = 192 */
    if (n > 64 && n <= 192) any_tridiag_reg(G, n, n <= 128 ? 2 : 3, d, e, tau);
    else if (n <= 64) any_tridiag_plain(G, n, d, e, tau);
    else if (tdv && tdv[0] == 'u') { if (NCT == 1) any_tridiag_plain(G, n, d, e, tau); else any_tridiag_fused(G, n, NCT, d, e, tau); }
    else if (tdv && tdv[0] == 'b') any_tridiag_blocked(G, n, NCT, NCT == 1 ? 16 : 32 / NCT, d, e, tau); /* round 3's first blocked form */
    else if (n <= 512) any_tridiag_sym(G, n, NCT, 8, NCT == 1 ? 8 : 4, d, e, tau);
    else any_tridiag_blocked(G, n, NCT, 32 / NCT, d, e, tau);
    /* Gershgorin hull, pivmin (min / max: order-free) */
    double lo = 1e300, hi = -1e300, e2m = 0.0;
    for (int i = 0; i < n; i++) {
        const double ei = (i < n - 1) ? e[i] : 0.0, eim = (i > 0) ? e[i - 1] : 0.0;
        e2[i] = ei * ei;
        const double rad = fabs(eim) + fabs(ei);
        lo = fmin(lo, d[i] - rad);
        hi = fmax(hi, d[i] + rad);
        e2m = fmax(e2m, ei * ei);
    }
    const double tn = fabs(lo) > fabs(hi) ? fabs(lo) : fabs(hi);
    const double pivmin = 2.2250738585072014e-300 * (e2m > 1.0 ? e2m : 1.0);
    const double slack = 2.0 * tn * 2.220446049250313e-16 * n + 2.0 * pivmin;
    const double lo2 = lo - slack, hi2 = hi + slack;
    int sc;
    frexp(fabs(lo2) > fabs(hi2) ? fabs(lo2) : fabs(hi2), &sc);
    const double a0 = ldexp(lo2, -sc), b0 = ldexp(hi2, -sc);
    for (int j = 0; j < n; j++) {
        const double es = (j > 0) ? ldexp(e[j - 1], -sc) : 0.0;
        ds[j] = ldexp(d[j], -sc);
        e2s[j] = es * es;
    }
    for (int r = 0; r < Rc; r++) { /* ten passes of 64-way multisection */
        const int kk = n - 1 - r;
        double a = a0, b = b0;
        for (int pass = 0; pass < 10; pass++) {
            const double h = (b - a) / 65.0;
            double xs[64];
            int jj = 64;
            for (int l = 0; l < 64; l++) {
                xs[l] = a + h * (double)(l + 1);
                if (jj == 64 && any_sturm_count(ds, e2s, n, xs[l]) > kk) jj = l;
            }
            const double na = (jj == 0) ? a : xs[jj - 1], nb = (jj == 64) ? b : xs[jj];
            a = na;
            b = nb;
        }
        lam[r] = ldexp(0.5 * (a + b), sc);
    }
    if (getenv("LRF_ORACLE_ANY_DEBUG")) { /* developer aid: d, e, lambda as raw doubles in E1 (tools/dev_any_init_bits.py) */
        double* dbg = (double*)E1;
        if ((long)n * R * 4 >= (long)(2 * n + Rc) * 8) {
            for (int i = 0; i < n; i++) { dbg[i] = d[i]; dbg[n + i] = e[i]; }
            for (int r = 0; r < Rc; r++) dbg[2 * n + r] = lam[r];
        }
        return 0;
    }
    /* twisted factorisation */
    for (int r = 0; r < Rc; r++) {
        const double lm = lam[r];
        double q = d[0] - lm;
        Dp[0] = q;
        for (int j = 1; j < n; j++) {
            if (fabs(q) < pivmin) q = -pivmin;
            q = (d[j] - lm) - e2[j - 1] / q;
            Dp[j] = q;
        }
        q = d[n - 1] - lm;
        Dm[n - 1] = q;
        for (int j = n - 2; j >= 0; j--) {
            if (fabs(q) < pivmin) q = -pivmin;
            q = (d[j] - lm) - e2[j] / q;
            Dm[j] = q;
        }
        int kt = 0;
        double best = 0.0;
        for (int j = 0; j < n; j++) {
            const double g = fabs((Dp[j] + Dm[j]) - (d[j] - lm));
            if (j == 0 || g < best) { best = g; kt = j; }
        }
        double* z = Z + (long)r * n;
        double xv = 1.0;
        z[kt] = 1.0;
        for (int j = kt - 1; j >= 0; j--) {
            double qq = Dp[j];
            if (fabs(qq) < pivmin) qq = -pivmin;
            xv = -(e[j] / qq) * xv;
            z[j] = xv;
        }
        xv = 1.0;
        for (int j = kt; j < n - 1; j++) {
            double qq = Dm[j + 1];
            if (fabs(qq) < pivmin) qq = -pivmin;
            xv = -(e[j] / qq) * xv;
            z[j + 1] = xv;
        }
    }
    if (getenv("LRF_ORACLE_ANY_STAGE") && atoi(getenv("LRF_ORACLE_ANY_STAGE")) / 100 == 13) { /* developer aid */
        double* dbg = (double*)E1;
        const int r0 = atoi(getenv("LRF_ORACLE_ANY_STAGE")) % 100;
        const long cap = (long)n * R / 2;
        for (long q = 0; q < cap; q++) {
            const long r = r0 + q / n;
            dbg[q] = (r < Rc) ? Z[r * n + q % n] : 0.0;
        }
        return 0;
    }
    /* orthonormalisation: classical Gram-Schmidt, twice, against the vectors already fixed */
    int failed = 0;
    for (int r = 0; r < Rc && !failed; r++) {
        double* Zr = Z + (long)r * n;
        int use_twisted = 1, uidx = 0;
        for (int i = 0; i < n; i++) { x[i] = Zr[i]; use_twisted &= isfinite(x[i]) != 0; }
        for (;;) {
            if (use_twisted) {
                for (int t = 0; t < 256; t++) {
                    double s = 0.0;
                    for (int c = 0; c < NCT; c++) { const int i = t + 256 * c; if (i < n) s = fma(x[i], x[i], s); }
                    tv[t] = s;
                }
                const double n0 = sqrt(block_sum256(tv));
                for (int i = 0; i < n; i++) x[i] = x[i] / n0;
            } else {
                if (uidx >= n) { failed = 1; break; }
                for (int i = 0; i < n; i++) x[i] = (i == uidx) ? 1.0 : 0.0;
                uidx++;
            }
            for (int pass = 0; pass < 2 && r > 0; pass++) {
                for (int pr = 0; pr < r; pr++) { /* lane l owns i = l, l + 64, ... (4 NCT of them) */
                    const double* Zp = Z + (long)pr * n;
                    for (int l = 0; l < 64; l++) {
                        double ds_ = 0.0;
                        for (int ee = 0; ee < 4 * NCT; ee++) { const int i = l + 64 * ee; if (i < n) ds_ = fma(Zp[i], x[i], ds_); }
                        lanes[l] = ds_;
                    }
                    cf[pr] = tree64c(lanes);
                }
                for (int pr = 0; pr < r; pr++) {
                    const double* Zp = Z + (long)pr * n;
                    for (int i = 0; i < n; i++) x[i] = fma(-cf[pr], Zp[i], x[i]);
                }
            }
            for (int t = 0; t < 256; t++) {
                double s = 0.0;
                for (int c = 0; c < NCT; c++) { const int i = t + 256 * c; if (i < n) s = fma(x[i], x[i], s); }
                tv[t] = s;
            }
            const double n2 = block_sum256(tv);
            if (n2 > 1e-6 && n2 < 1e300) {
                const double nr = sqrt(n2);
                for (int i = 0; i < n; i++) x[i] = x[i] / nr;
                break;
            }
            use_twisted = 0;
        }
        for (int i = 0; i < n; i++) Zr[i] = x[i];
    }
    if (getenv("LRF_ORACLE_ANY_STAGE") && atoi(getenv("LRF_ORACLE_ANY_STAGE")) / 100 == 14) { /* developer aid */
        double* dbg = (double*)E1;
        const int r0 = atoi(getenv("LRF_ORACLE_ANY_STAGE")) % 100;
        const long cap = (long)n * R / 2;
        for (long q = 0; q < cap; q++) {
            const long r = r0 + q / n;
            dbg[q] = (r < Rc) ? Z[r * n + q % n] : 0.0;
        }
        return 0;
    }
    /* back-transformation x <- H_0 H_1 ... H_{n-3} x (lane l owns i = l + 64 e), scaling, column sign */
    const int NE = 4 * NCT;
    for (long q = 0; q < (long)n * R; q++) { E1[q] = 0.f; E2[q] = 0.f; }
    for (int r = 0; r < Rc && !failed; r++) {
        for (int i = 0; i < n; i++) x[i] = Z[(long)r * n + i];
        for (int k = n - 3; k >= 0; k--) {
            const double tk = tau[k];
            if (tk == 0.0) continue;
            const double* vrow = G + (long)k * n;
            for (int l = 0; l < 64; l++) {
                double ds_ = 0.0;
                for (int ee = 0; ee < NE; ee++) {
                    const int i = l + 64 * ee;
                    const double vv = (i < n && i > k) ? vrow[i] : 0.0, xx = (i < n) ? x[i] : 0.0;
                    ds_ = fma(vv, xx, ds_);
                }
                lanes[l] = ds_;
            }
            const double scl = tk * tree64c(lanes);
            /* every entry, with v = 0 up to k, as the kernel's lanes do: fma(-scl, 0, x) turns an entry -0.0 into +0.0 when
             * scl < 0 (exact zeros occur in the unit-vector fallbacks of rank-deficient matrices) */
            for (int i = 0; i < n; i++) x[i] = fma(-scl, (i > k) ? vrow[i] : 0.0, x[i]);
        }
        for (int l = 0; l < 64; l++) {
            double ds_ = 0.0;
            for (int ee = 0; ee < NE; ee++) {
                const int i = l + 64 * ee;
                ds_ = fma((double)(i + 1), (i < n) ? x[i] : 0.0, ds_);
            }
            lanes[l] = ds_;
        }
        const double dot = tree64c(lanes);
        const double lm = lam[r];
        const double sr = sqrt(sqrt(lm > 1e-200 ? lm : 0.0));
        const int sg = sign ? (int)sign[r] : 0;
        const double want = sg ? (double)sg : -1.0;
        const double flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.0 : -1.0;
        for (int i = 0; i < n; i++) {
            const double ev = flip * x[i];
            E1[(long)i * R + r] = (float)(ev * sr);
            E2[(long)i * R + r] = (sr > 0.0) ? (float)(ev / sr) : 0.f;
        }
    }
    free(d); free(e); free(tau); free(e2); free(ds); free(e2s); free(lam); free(Z); free(Dp); free(Dm); free(x); free(cf);
    return failed ? -1 : 0;
}

/* any_run_init (lrf_anyshape_host.inc): u0 = U sqrt(s) [M,R], v0 = V sqrt(s) [N,R] for any shape.  The eigen-problem is
 * solved on the SHORT side; when M < N the roles of the factors swap and the column sign is imposed on the finished v0
 * (k_any_signfix), as include/lrf_hip.h states it. */
int lrf_oracle_svd_topr_any(const float* X, long M, long N, int R, const int8_t* sign, float* u, float* v)
{
    const int tall = N <= M;
    const long n = tall ? N : M, D = tall ? M : N;
    double* G = (double*)malloc(sizeof(double) * n * n);
    float* e1 = (float*)malloc(sizeof(float) * n * R);
    float* e2 = (float*)malloc(sizeof(float) * n * R);
    any_gram(X, tall ? N : 1, tall ? 1 : N, n, D, G);
    int rc = lrf_oracle_any_eig(G, (int)n, R, (int)n, tall ? sign : NULL, e1, e2);
    if (rc == 0) {
        if (tall) {
            memcpy(v, e1, sizeof(float) * n * R);
            mm_mkl(X, N, 1, e2, R, 1, u, R, M, N, R);
        } else {
            memcpy(u, e1, sizeof(float) * n * R);
            mm_mkl(X, 1, N, e2, R, 1, v, R, N, M, R);
            for (int r = 0; r < R; r++) { /* k_any_signfix: lane l owns j = l, l + 64, ...; wave_sum */
                double lanes[64];
                for (int l = 0; l < 64; l++) {
                    double acc = 0.0;
                    for (long j = l; j < N; j += 64) acc = fma((double)(j + 1), (double)v[j * R + r], acc);
                    lanes[l] = acc;
                }
                const double dot = tree64c(lanes);
                const double want = (sign && sign[r]) ? (double)sign[r] : -1.0;
                const float flip = ((dot < 0.0 ? -1.0 : 1.0) == want) ? 1.f : -1.f;
                for (long j = 0; j < N; j++) v[j * R + r] = flip * v[j * R + r];
                for (long i = 0; i < M; i++) u[i * R + r] = flip * u[i * R + r];
            }
        }
    }
    free(G); free(e1); free(e2);
    return rc;
}

/* any_factors_from_gram for the [M,N] matrices of svd_encode and of the RGB colour-space branch (N = c p q, uint8-valued X):
 * v = e sqrt(sigma) [N,R], u = X (e / sqrt(sigma)) [M,R] from the EXACT Gram matrix. */
int lrf_oracle_svd_topr_u8(const float* X, long M, long N, int R, const int8_t* sign, float* u, float* v)
{
    double* G = (double*)malloc(sizeof(double) * N * N);
    float* e2 = (float*)malloc(sizeof(float) * N * R);
    lrf_oracle_gram_u8_exact(X, M, N, G);
    int rc = lrf_oracle_any_eig(G, (int)N, R, (int)(M < N ? M : N), sign, v, e2);
    if (rc == 0) mm_mkl(X, N, 1, e2, R, 1, u, R, M, N, R);
    free(G); free(e2);
    return rc;
}
