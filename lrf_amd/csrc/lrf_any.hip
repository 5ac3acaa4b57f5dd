// lrf_any.hip — the any-shape path of liblrf_hip.so: patch sizes other than 8x8, patch=False, the RGB colour space, the
// general QMF solver, ranks 33..64 of the 64-column path, and the SVD baseline (svd_encode / svd_decode) — kernels
// (lrf_svd_kernels.hip, lrf_anyshape_kernels.hip), their launch sequences (lrf_anyshape_host.inc) and the C ABI entry points
// built on them (include/lrf_hip.h).
#include "lrf_host.h"
#include "lrf_svd_kernels.hip"
#include "lrf_anyshape_kernels.hip"

#include "lrf_anyshape_host.inc"

extern "C" {

int lrf_qmf_loss_f32(lrf_ctx* c, const float* X, const float* U, const float* V, const float* W, int64_t B, int64_t M, int64_t N, int R,
                     float* loss)
{
    if (!c || !X || !U || !V || !loss) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535 || M < 1 || N < 1 || R < 1 || M > (1 << 24) || N > (1 << 24)) return set_err(LRF_EINVAL, "bad sizes");
    LRF_ON_DEVICE(c);
    int rc;
    if ((rc = ensure(c, c->smm, (size_t)B * 2 * sizeof(double)))) return rc;
    double* acc = (double*)c->smm.p;
    HIP_TRY(hipMemsetAsync(acc, 0, (size_t)B * 2 * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_any_loss, dim3((unsigned)((M + 63) / 64), (unsigned)B), dim3(256), 0, c->stream, X, U, V, W, (int)M, (int)N, R, acc);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_any_loss_finish, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, c->stream, (const double*)acc, (int)B, loss);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_qmf_decompose_ex_f32(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, int K, const lrf_qmf_opts* o,
                             const int8_t* sign, const float* U0, const float* V0, float* U, float* V, float* W)
{
    if (!c || !X || !o || !U || !V || !W) return set_err(LRF_EINVAL, "NULL argument");
    if ((U0 == nullptr) != (V0 == nullptr)) return set_err(LRF_EINVAL, "U0 and V0 must be given together");
    if (K < 0) return set_err(LRF_EINVAL, "num_iters must be >= 0");
    if (o->factors & ~7) return set_err(LRF_EINVAL, "factors: bits 0 (u), 1 (v), 2 (w) only");
    if (o->bounded && !(o->lo <= o->hi)) return set_err(LRF_EINVAL, "bounds (%g, %g)", (double)o->lo, (double)o->hi);
    if (!(o->l2_u >= 0.0) || !(o->l2_v >= 0.0) || !(o->l1_ratio >= 0.0 && o->l1_ratio <= 1.0))
        return set_err(LRF_EINVAL, "l2 must be >= 0 and l1_ratio in [0, 1]");
    if (!(o->eps >= 0.0)) return set_err(LRF_EINVAL, "eps must be >= 0 (0 selects the default 1e-16)");
    if (o->w_init && !U0) return set_err(LRF_EINVAL, "w_init needs the initial factors (U0, V0) it belongs to");
    int rc = any_check(B, M, N, R, -128, 127);
    if (rc) return rc;
    LRF_ON_DEVICE(c);
    if ((rc = any_workspace(c, (int)B, (int)M, (int)N, R))) return rc;
    if (U0) {
        HIP_TRY(hipMemcpyAsync(c->any_uf.p, U0, (size_t)B * M * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->any_vf.p, V0, (size_t)B * N * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else if ((rc = any_run_init(c, X, (int)B, (int)M, (int)N, R, sign))) {
        return rc;
    }
    // w = [0; 1] (SVDInit, qmf.py:54,70), on the device — or the caller's initial pair (SVDInit(num_levels=...), qmf.py:56-68)
    if ((rc = ensure(c, c->sign, (size_t)2 * B * sizeof(float)))) return rc; // the (otherwise unused here) sign scratch holds w
    float* Wd = (float*)c->sign.p;
    if (o->w_init) {
        HIP_TRY(hipMemcpyAsync(Wd, W, (size_t)2 * B * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        std::vector<float> w0((size_t)2 * B);
        for (int64_t b = 0; b < B; b++) { w0[2 * b] = 0.f; w0[2 * b + 1] = 1.f; }
        if ((rc = upload(c, c->sign, w0.data(), w0.size() * sizeof(float)))) return rc;
        Wd = (float*)c->sign.p;
    }
    // qmf.py:154-157: the products in double like Python, fp32 where they meet fp32 tensors; bounds through ceil / floor (:194)
    const float l1_u = (float)(o->l2_u * o->l1_ratio * (double)N), l2_u = (float)(o->l2_u * (1.0 - o->l1_ratio) * (double)N);
    const float l1_v = (float)(o->l2_v * o->l1_ratio * (double)M), l2_v = (float)(o->l2_v * (1.0 - o->l1_ratio) * (double)M);
    const float lo = o->bounded ? ceilf(o->lo) : -INFINITY, hi = o->bounded ? floorf(o->hi) : INFINITY;
    const float eps = o->eps > 0.0 ? (float)o->eps : LRF_EPS; // a Python float meeting fp32 tensors: rounded to fp32 (qmf.py:117-118)
    if ((rc = any_run_bcd_general(c, X, (int)B, (int)M, (int)N, R, K, lo, hi, l1_u, l2_u, l1_v, l2_v, o->factors, Wd, eps, o->w_init != 0)))
        return rc;
    HIP_TRY(hipMemcpyAsync(U, c->any_uf.p, (size_t)B * M * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(V, c->any_vf.p, (size_t)B * N * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(W, Wd, (size_t)2 * B * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return LRF_OK;
}

// Gram matrices of B matrices [M,192] of uint8-valued floats (svd_encode, RGB colour-space branch): exact, int8 MFMA
// int32 sums hold 128^2 x 131072 rows; longer matrices take the fp64 kernel (float X only)
#define LRF_G192_MAX_ROWS 100000
extern "C++" {
template <typename T>
static int gram192_u8(lrf_ctx* c, const T* X, long xs, int B, int M, double* G)
{
    static const bool use_f64 = dev_flag("LRF_GRAM192_F64");
    if constexpr (sizeof(T) == 4) {
        if (use_f64 || M > LRF_G192_MAX_ROWS) {
            hipLaunchKernelGGL(k_gram_blk, dim3(6, (unsigned)B), dim3(256), 0, c->stream, X, xs, M, 192, 3, G);
            LAUNCH_CHECK();
            return LRF_OK;
        }
    }
    const int nchunks = (M + LRF_G192_ROWS - 1) / LRF_G192_ROWS;
    int rc;
    if ((rc = ensure(c, c->any_td, (size_t)B * nchunks * (192 * 192 + 192) * sizeof(int)))) return rc;
    int* P = (int*)c->any_td.p; // consumed by the fold before the eigen-solver reuses the buffer (same stream)
    hipLaunchKernelGGL((k_gram192_u8<T>), dim3((unsigned)nchunks, (unsigned)B), dim3(256), 0, c->stream, X, xs, M, P, nchunks);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gram192_fold, dim3(78, (unsigned)B), dim3(256), 0, c->stream, (const int*)P, nchunks, M, G);
    LAUNCH_CHECK();
    return LRF_OK;
}
} // extern "C++"

/* ---- SVD baseline (lrf.svd_encode / svd_decode, default RGB branch) ---- */
static int svd_geom(int64_t H, int64_t W, int* hp, int* wp, int* top, int* left, int* nw, int* M)
{
    if (H < 1 || W < 1) return set_err(LRF_EINVAL, "bad image size");
    int64_t ph = (8 - H % 8) % 8, pw = (8 - W % 8) % 8;
    if (ph / 2 >= H || ph - ph / 2 >= H || pw / 2 >= W || pw - pw / 2 >= W)
        return set_err(LRF_EINVAL, "reflect padding larger than the image (%ldx%ld)", (long)H, (long)W);
    *hp = (int)(H + ph); *wp = (int)(W + pw); *top = (int)(ph / 2); *left = (int)(pw / 2);
    *nw = *wp / 8; *M = (*hp / 8) * (*wp / 8);
    return LRF_OK;
}

int lrf_svd_encode_rgb_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int R, const int8_t* sign,
                          uint8_t* U, uint8_t* V, float* qparams)
{
    if (!c || !rgb || !U || !V || !qparams) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (R < 1) return set_err(LRF_EINVAL, "rank must be >= 1 (got %d)", R);
    if (R > 192) return set_err(LRF_EINVAL, "svd_encode: rank %d > 192 columns", R);
    int hp, wp, top, left, nw, M, rc;
    if ((rc = svd_geom(H, W, &hp, &wp, &top, &left, &nw, &M))) return rc;
    const int N = 192;
    LRF_ON_DEVICE(c);
    long xs = (long)M * N;
    if ((rc = ensure(c, c->sx, (size_t)B * xs * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->sg, (size_t)B * N * N * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->svn, (size_t)B * N * R * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->swn, (size_t)B * N * R * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->suf, (size_t)B * M * R * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->smm, (size_t)B * 4 * sizeof(float)))) return rc;
    float* X = (float*)c->sx.p;
    double* G = (double*)c->sg.p;
    float* Vn = (float*)c->svn.p;
    float* Wn = (float*)c->swn.p;
    float* Uf = (float*)c->suf.p;
    float* mm = (float*)c->smm.p;
    static const bool f32_matrix = dev_flag("LRF_SVD_F32_MATRIX");
    if (R <= 8 && M <= LRF_G192_MAX_ROWS && !f32_matrix) {
        // the matrix as BYTES (round 3): its three passes — this one, the exact Gram matrix, u = X w — move a quarter of the bytes
        uint8_t* X8 = (uint8_t*)c->sx.p; // (allocated for the float matrix: four times what the bytes need)
        hipLaunchKernelGGL((k_patchify_rgb<uint8_t>), dim3(hp / 8, (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H, (int)W, top, left, nw, xs, X8);
        LAUNCH_CHECK();
        if ((rc = gram192_u8(c, (const uint8_t*)X8, xs, (int)B, M, G))) return rc;
        if ((rc = any_eig_from_gram(c, G, (int)B, M, N, R, sign, Vn, Wn))) return rc;
        const dim3 pg((unsigned)((M + 256 * LRF_PROD192_GROUPS - 1) / (256 * LRF_PROD192_GROUPS)), (unsigned)B);
#define LRF_LAUNCH_P192(RR) \
    case RR: hipLaunchKernelGGL((k_prod192_u8<RR>), pg, dim3(256), 0, c->stream, (const uint8_t*)X8, xs, M, (const float*)Wn, Uf); break;
        switch (R) {
            LRF_LAUNCH_P192(1) LRF_LAUNCH_P192(2) LRF_LAUNCH_P192(3) LRF_LAUNCH_P192(4) LRF_LAUNCH_P192(5) LRF_LAUNCH_P192(6)
            LRF_LAUNCH_P192(7) LRF_LAUNCH_P192(8)
        }
#undef LRF_LAUNCH_P192
        LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL((k_patchify_rgb<float>), dim3(hp / 8, (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H, (int)W, top, left, nw, xs, X);
        LAUNCH_CHECK();
        if ((rc = gram192_u8(c, (const float*)X, xs, (int)B, M, G))) return rc;
        if ((rc = any_factors_from_gram(c, X, G, (int)B, M, N, R, sign, Vn, Wn, Uf))) return rc;
    }
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)B), dim3(256), 0, c->stream, (const float*)Uf, (long)M * R, (long)M * R, mm);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)B), dim3(256), 0, c->stream, (const float*)Vn, (long)N * R, (long)N * R, mm + 2 * B);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_quantize_u8, dim3((unsigned)(((long)M * R + 255) / 256), (unsigned)B), dim3(256), 0, c->stream,
                       (const float*)Uf, (long)M * R, (long)M * R, (const float*)mm, U, qparams, 4, 0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_quantize_u8, dim3((unsigned)(((long)N * R + 255) / 256), (unsigned)B), dim3(256), 0, c->stream,
                       (const float*)Vn, (long)N * R, (long)N * R, (const float*)(mm + 2 * B), V, qparams, 4, 2);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_svd_decode_rgb_u8(lrf_ctx* c, const uint8_t* U, const uint8_t* V, int64_t B, int64_t H, int64_t W, int R,
                          const float* qparams6, uint8_t* rgb)
{
    if (!c || !U || !V || !qparams6 || !rgb) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (R < 1 || R > 192) return set_err(LRF_EINVAL, "rank %d out of range", R);
    int hp, wp, top, left, nw, M, rc;
    if ((rc = svd_geom(H, W, &hp, &wp, &top, &left, &nw, &M))) return rc;
    LRF_ON_DEVICE(c);
    long n4 = 3L * H * ((W + 3) / 4);
    hipLaunchKernelGGL(k_svd_decode, dim3((unsigned)((n4 + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W, top,
                       left, nw, M, R, qparams6, rgb);
    LAUNCH_CHECK();
    return LRF_OK;
}

/* ---- QMF, RGB colour-space branch (qmf_encode(color_space="RGB", patch=True): qmf.py:164-187; decode :311-323) ---- */
static int rgbspace_check(int64_t B, int64_t H, int64_t W, int R, int K, int lo, int hi, int M)
{
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (R < 1) return set_err(LRF_EINVAL, "rank must be >= 1 (got %d)", R);
    if (R > 192) return set_err(LRF_ENOTSUP, "RGB colour space: rank %d > 192 columns not implemented", R);
    if (K < 1) return set_err(LRF_ENOTSUP, "RGB colour space: num_iters=%d not implemented (K >= 1)", K);
    if (lo > hi || lo < -128 || hi > 127) return set_err(LRF_EINVAL, "bounds (%d,%d) outside int8", lo, hi);
    (void)M; // u.mT @ u stays the reference's for any int8 bounds: see check_params
    (void)H; (void)W;
    return LRF_OK;
}

int lrf_qmf_rgbspace_encode_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int R, int K, int lo, int hi,
                               const int8_t* sign, const float* U0, const float* V0, int8_t* U, int8_t* V)
{
    if (!c || !rgb || !U || !V) return set_err(LRF_EINVAL, "NULL argument");
    if ((U0 == nullptr) != (V0 == nullptr)) return set_err(LRF_EINVAL, "U0 and V0 must be given together");
    int hp, wp, top, left, nw, M, rc;
    if ((rc = svd_geom(H, W, &hp, &wp, &top, &left, &nw, &M))) return rc;
    if ((rc = rgbspace_check(B, H, W, R, K, lo, hi, M))) return rc;
    const int N = 192;
    LRF_ON_DEVICE(c);
    const long xs = (long)M * N;
    if ((rc = ensure(c, c->sx, (size_t)B * xs * sizeof(float)))) return rc;
    float* X = (float*)c->sx.p;
    {
        Prof p(c, LRF_K_PLANES);
        hipLaunchKernelGGL((k_patchify_rgb<float>), dim3(hp / 8, (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H, (int)W, top, left, nw, xs, X);
        LAUNCH_CHECK();
    }
    // The factorisation itself runs on the any-shape kernels (lrf_anyshape_kernels.hip): measured against a dedicated
    // [M,192] kernel set with plain VALU chains they took half the time, so that set is gone.
    if ((rc = any_workspace(c, (int)B, M, N, R))) return rc;
    float* Uf = (float*)c->any_uf.p;
    float* Vf = (float*)c->any_vf.p;
    if (U0) {
        HIP_TRY(hipMemcpyAsync(Uf, U0, (size_t)B * M * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(Vf, V0, (size_t)B * N * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else { // SVD initialisation: u0 = U sqrt(s), v0 = V sqrt(s) (qmf.py:42-71): fp64 MFMA Gram, then the any-shape eigen-solver
        if ((rc = ensure(c, c->sg, (size_t)B * N * N * sizeof(double)))) return rc;
        if ((rc = ensure(c, c->swn, (size_t)B * N * R * sizeof(float)))) return rc;
        double* G = (double*)c->sg.p;
        float* Wn = (float*)c->swn.p;
        Prof p(c, LRF_K_INIT);
        if ((rc = gram192_u8(c, (const float*)X, xs, (int)B, M, G))) return rc;
        if ((rc = any_factors_from_gram(c, X, G, (int)B, M, N, R, sign, Vf, Wn, Uf))) return rc;
    }
    return any_run_bcd(c, X, (int)B, M, N, R, K, lo, hi, U, V);
}

int lrf_qmf_rgbspace_decode_u8(lrf_ctx* c, const int8_t* U, const int8_t* V, int64_t B, int64_t H, int64_t W, int R, uint8_t* rgb)
{
    if (!c || !U || !V || !rgb) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (R < 1 || R > 192) return set_err(LRF_EINVAL, "rank %d out of range", R);
    int hp, wp, top, left, nw, M, rc;
    if ((rc = svd_geom(H, W, &hp, &wp, &top, &left, &nw, &M))) return rc;
    LRF_ON_DEVICE(c);
    const long n = 3L * H * W;
    Prof p(c, LRF_K_DECODE);
    hipLaunchKernelGGL(k_qmf_decode_rgbspace, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W,
                       top, left, nw, M, R, rgb);
    LAUNCH_CHECK();
    return LRF_OK;
}

// geometry of the RGB colour-space branch for patches (p, q) (reflect padding to multiples, lrf/compression/utils.py:108-132) or
// none (p = q = 0: per channel the plane [H, W])
static int rgbspace_geom_any(int64_t H, int64_t W, int p, int q, int* hp, int* wp, int* top, int* left, int* nw, long* M, long* N)
{
    if (H < 1 || W < 1) return set_err(LRF_EINVAL, "bad image size");
    if ((p == 0) != (q == 0) || p < 0 || q < 0) return set_err(LRF_EINVAL, "patch size (%d, %d)", p, q);
    if (p == 0) {
        *hp = (int)H; *wp = (int)W; *top = 0; *left = 0; *nw = 0; *M = H; *N = W;
        return LRF_OK;
    }
    const int64_t ph = (p - H % p) % p, pw = (q - W % q) % q;
    if (ph / 2 >= H || ph - ph / 2 >= H || pw / 2 >= W || pw - pw / 2 >= W)
        return set_err(LRF_EINVAL, "reflect padding larger than the image (%ldx%ld, patches %dx%d)", (long)H, (long)W, p, q);
    *hp = (int)(H + ph); *wp = (int)(W + pw); *top = (int)(ph / 2); *left = (int)(pw / 2);
    *nw = *wp / q;
    *M = (long)(*hp / p) * (*wp / q);
    *N = 3L * p * q;
    return LRF_OK;
}

int lrf_rgbspace_dims_any(int64_t H, int64_t W, int p, int q, int64_t* hp, int64_t* wp, int64_t* M, int64_t* N)
{
    if (!hp || !wp || !M || !N) return set_err(LRF_EINVAL, "NULL argument");
    int h2, w2, top, left, nw, rc;
    long m, n;
    if ((rc = rgbspace_geom_any(H, W, p, q, &h2, &w2, &top, &left, &nw, &m, &n))) return rc;
    *hp = h2; *wp = w2; *M = m; *N = n;
    return LRF_OK;
}

int lrf_qmf_rgbspace_matrix_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int p, int q, float* X)
{
    if (!c || !rgb || !X) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    int hp, wp, top, left, nw, rc;
    long M, N;
    if ((rc = rgbspace_geom_any(H, W, p, q, &hp, &wp, &top, &left, &nw, &M, &N))) return rc;
    LRF_ON_DEVICE(c);
    const long elems = p ? M * N : 3L * H * W;
    Prof pr(c, LRF_K_PLANES);
    hipLaunchKernelGGL(k_rgb_matrix_any, dim3((unsigned)((elems + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H, (int)W, p, q,
                       top, left, nw, elems, X);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_qmf_rgbspace_decode_any_u8(lrf_ctx* c, const int8_t* U, const int8_t* V, int64_t B, int64_t H, int64_t W, int p, int q, int R,
                                   uint8_t* rgb)
{
    if (!c || !U || !V || !rgb) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (R < 1 || R > LRF_ANY_MAX_RANK) return set_err(LRF_EINVAL, "rank %d out of range", R);
    int hp, wp, top, left, nw, rc;
    long M, N;
    if ((rc = rgbspace_geom_any(H, W, p, q, &hp, &wp, &top, &left, &nw, &M, &N))) return rc;
    LRF_ON_DEVICE(c);
    const long u_img = (p ? M : 3L * H) * R, v_img = (p ? N : 3L * W) * R;
    const long n = 3L * H * W;
    Prof pr(c, LRF_K_DECODE);
    hipLaunchKernelGGL(k_rgb_decode_any, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W, p, q, top,
                       left, nw, u_img, v_img, R, rgb);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_quantize_u8(lrf_ctx* c, const float* T, int64_t B, int64_t per, uint8_t* Q, float* qparams)
{
    if (!c || !T || !Q || !qparams) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535 || per < 1) return set_err(LRF_EINVAL, "bad sizes");
    LRF_ON_DEVICE(c);
    int rc;
    if ((rc = ensure(c, c->smm, (size_t)B * 2 * sizeof(float)))) return rc;
    float* mm = (float*)c->smm.p;
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)B), dim3(256), 0, c->stream, T, (long)per, (long)per, mm);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_quantize_u8, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, T, (long)per, (long)per,
                       (const float*)mm, Q, qparams, 2, 0);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_svd_decode_any_u8(lrf_ctx* c, const void* U, const void* V, int factors_are_float, int64_t B, int64_t H, int64_t W, int p, int q,
                          int R, const float* qparams6, uint8_t* rgb)
{
    if (!c || !U || !V || !rgb) return set_err(LRF_EINVAL, "NULL argument");
    if (!factors_are_float && !qparams6) return set_err(LRF_EINVAL, "quantised factors need their (scale, min, qmin) parameters");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (R < 1 || R > 16384) return set_err(LRF_EINVAL, "rank %d out of range", R);
    int hp, wp, top, left, nw, rc;
    long M, N;
    if ((rc = rgbspace_geom_any(H, W, p, q, &hp, &wp, &top, &left, &nw, &M, &N))) return rc;
    LRF_ON_DEVICE(c);
    const long u_img = (p ? M : 3L * H) * R, v_img = (p ? N : 3L * W) * R;
    const long n = 3L * H * W;
    Prof pr(c, LRF_K_DECODE);
    if (factors_are_float)
        hipLaunchKernelGGL(k_svd_decode_any<false>, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W, p,
                           q, top, left, nw, u_img, v_img, R, qparams6, rgb);
    else
        hipLaunchKernelGGL(k_svd_decode_any<true>, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W, p,
                           q, top, left, nw, u_img, v_img, R, qparams6, rgb);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_plane_dims_any_hw(int64_t H, int64_t W, int64_t hc, int64_t wc, int p, int q, int ch, int64_t* h, int64_t* w, int64_t* hp,
                          int64_t* wp, int64_t* M, int64_t* N)
{
    if (!h || !w || !hp || !wp || !M || !N) return set_err(LRF_EINVAL, "NULL argument");
    AnyGeom g;
    int rc = any_geom(H, W, p, q, ch, &g, hc, wc);
    if (rc) return rc;
    *h = g.h; *w = g.w; *hp = g.hp; *wp = g.wp; *M = g.M; *N = g.N;
    return LRF_OK;
}

int lrf_plane_dims_any(int64_t H, int64_t W, int p, int q, int ch, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M,
                       int64_t* N)
{
    return lrf_plane_dims_any_hw(H, W, 0, 0, p, q, ch, h, w, hp, wp, M, N);
}

int lrf_qmf_planes_any_hw_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int64_t hc, int64_t wc, int p, int q, int ch,
                             float* X)
{
    if (!c || !rgb || !X) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    AnyGeom g;
    int rc = any_geom(H, W, p, q, ch, &g, hc, wc);
    if (rc) return rc;
    LRF_ON_DEVICE(c);
    const long elems = g.M * g.N;
    Prof pr(c, LRF_K_PLANES);
    hipLaunchKernelGGL(k_any_planes, dim3((unsigned)((elems + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H, (int)W, ch,
                       g.h, g.w, p, q, g.top, g.left, g.nw, elems, X);
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_qmf_planes_any_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int p, int q, int ch, float* X)
{
    return lrf_qmf_planes_any_hw_u8(c, rgb, B, H, W, 0, 0, p, q, ch, X);
}

int lrf_qmf_decode_any_u8(lrf_ctx* c, const int8_t* U0, const int8_t* V0, const int8_t* U1, const int8_t* V1, const int8_t* U2,
                          const int8_t* V2, int64_t B, int64_t H, int64_t W, int p, int q, const int R[3], uint8_t* rgb)
{
    return lrf_qmf_decode_any_hw_u8(c, U0, V0, U1, V1, U2, V2, B, H, W, 0, 0, p, q, R, rgb);
}

int lrf_qmf_decode_any_hw_u8(lrf_ctx* c, const int8_t* U0, const int8_t* V0, const int8_t* U1, const int8_t* V1, const int8_t* U2,
                             const int8_t* V2, int64_t B, int64_t H, int64_t W, int64_t hc, int64_t wc, int p, int q, const int R[3],
                             uint8_t* rgb)
{
    if (!c || !U0 || !V0 || !U1 || !V1 || !U2 || !V2 || !R || !rgb) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    const int8_t* Us[3] = {U0, U1, U2};
    const int8_t* Vs[3] = {V0, V1, V2};
    AnyDecodePlane d[3];
    for (int ch = 0; ch < 3; ch++) {
        AnyGeom g;
        int rc = any_geom(H, W, p, q, ch, &g, hc, wc);
        if (rc) return rc;
        if (R[ch] < 1 || R[ch] > 16384) return set_err(LRF_EINVAL, "rank %d out of range", R[ch]);
        d[ch].U = Us[ch]; d[ch].V = Vs[ch];
        d[ch].u_img = g.M * R[ch]; d[ch].v_img = g.N * R[ch];
        d[ch].h = g.h; d[ch].w = g.w; d[ch].p = p; d[ch].q = q; d[ch].top = g.top; d[ch].left = g.left; d[ch].nw = g.nw;
        d[ch].R = R[ch];
    }
    LRF_ON_DEVICE(c);
    Prof pr(c, LRF_K_DECODE);
    hipLaunchKernelGGL(k_any_decode, dim3((unsigned)(((long)H * W + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, d[0], d[1], d[2],
                       (int)H, (int)W, rgb);
    LAUNCH_CHECK();
    return LRF_OK;
}

} // extern "C"
