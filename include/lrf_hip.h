/*
 * lrf_hip.h — C ABI of liblrf_hip.so: the MI355X (gfx950) implementation of pashtari/lrf's
 * QMF factorisation hot path.
 *
 * The reference has no FFI: it is pure Python on torch CPU tensors.  The seam this library cuts is
 * the one SURVEY.md §8(b) names, and every entry point cites the reference code it replaces
 * (paths relative to the reference root).  A maintainer of the reference binds these with ctypes;
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer (HBM) unless the parameter name ends in `_host`.
 *     lrf_malloc / lrf_free / lrf_memcpy_* let a host without any GPU runtime of its own
 *     (plain ctypes + numpy) move data.  No torch types cross this boundary.
 *   - Caller owns inputs and outputs.  The library owns only the per-context scratch workspace,
 *     grown on demand and released by lrf_ctx_destroy.
 *   - All work is enqueued on the context's HIP stream (lrf_ctx_set_stream); calls return after
 *     enqueueing unless stated.  lrf_ctx_synchronize waits for the stream.
 *   - Return value: 0 = ok, negative = LRF_E*.  lrf_last_error() returns a thread-local message.
 *   - A context is not thread-safe; use one per (host thread, device).  Distinct contexts are
 *     independent.
 *   - Factors are int8 (requires -128 <= lo <= hi <= 127; the reference default is (-16, 15),
 *     lrf/compression/qmf.py:124) in row-major [M,R] / [N,R] layout, exactly the layout
 *     `u.to(int8)`, `v.to(int8)` have at lrf/compression/qmf.py:258-260.
 */
#ifndef LRF_HIP_H
#define LRF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRF_OK 0
#define LRF_EINVAL (-1)      /* bad argument (the reference raises AssertionError / ValueError) */
#define LRF_ENOTSUP (-2)     /* valid in the reference but not implemented on this path yet */
#define LRF_EHIP (-3)        /* a HIP runtime call failed; see lrf_last_error() */
#define LRF_ENOMEM (-4)

#define LRF_MAX_RANK 64      /* largest rank of the 64-column kernels (ranks above 16 run on the untuned big-rank kernels) */
#define LRF_PATCH_ELEMS 64   /* N = p*q of the tuned kernels (8x8 patches); other N run on the any-shape kernels */
#define LRF_ANY_MAX_SIDE 2048 /* any-shape path: largest min(M, N) of the SVD initialisation */
#define LRF_ANY_MAX_RANK 639  /* any-shape path: largest rank */

typedef struct lrf_ctx lrf_ctx;

/* ---- runtime ------------------------------------------------------------------------------ */
const char* lrf_last_error(void);
int lrf_device_count(void);
int lrf_version(void);

int lrf_ctx_create(int device, lrf_ctx** out);
void lrf_ctx_destroy(lrf_ctx* ctx);
/* A new context enqueues on a non-blocking stream of its own.  lrf_ctx_set_stream switches to the
 * caller's hipStream_t, taken as is (NULL = HIP's default stream, which is what a torch process uses
 * unless it changed streams); lrf_ctx_use_own_stream switches back.  Both first wait for the stream in use. */
int lrf_ctx_set_stream(lrf_ctx* ctx, void* hip_stream);
int lrf_ctx_use_own_stream(lrf_ctx* ctx);
int lrf_ctx_synchronize(lrf_ctx* ctx);
/* Asynchronous failures.  Large calls run their iterations (all K at ranks <= 16, else 2..K) in ONE launch (k_bcd_p) whose waves wait for each
 * other with BOUNDED polls; a poll that expires (never observed in service; the exit every wave reaches) makes the launch
 * give up and report its number in page-locked host memory.  Because calls only enqueue, the report is read where results are
 * handed back: lrf_ctx_synchronize (after its wait), lrf_pipe_wait_next (for the piece it returns), the next persistent
 * call's entry, and lrf_ctx_check — which does NOT wait: call it after the stream has been waited for by other means (a
 * device-to-host copy on the stream, an event).  A non-zero return means: the factors of the named call and of every later
 * call on this context up to the check are invalid; the context itself stays usable (the next call clears the state).
 * Replaces nothing in the reference (its calls are synchronous Python): this is the error half of the async boundary. */
int lrf_ctx_check(lrf_ctx* ctx);
/* bytes of scratch the context currently holds.  The scratch grows to the largest call seen (2.4 KB per input pixel of the
 * default branch: a 512 x 1365x2048 batch holds ~9 GB) and is kept for reuse; lrf_ctx_trim waits for the stream and gives
 * all of it back (the next call allocates again). */
size_t lrf_ctx_workspace_bytes(const lrf_ctx* ctx);
int lrf_ctx_trim(lrf_ctx* ctx);

/* Per-kernel timing with HIP events on the context's stream (off by default; when on, every launch
 * of the kernels below is bracketed by events).  kernel ids: LRF_K_*.  lrf_ctx_kernel_time
 * synchronises the stream and returns the accumulated milliseconds and launch count. */
#define LRF_K_PLANES 0       /* rgb -> patch matrices          */
#define LRF_K_INIT 1         /* SVD initialisation             */
#define LRF_K_BCD 2          /* U update + X^T U partials      */
#define LRF_K_VUPDATE 3      /* V update                       */
#define LRF_K_DECODE 4       /* factors -> rgb                 */
#define LRF_K_GRAM 5         /* exact Gram matrices (input of the SVD initialisation) */
#define LRF_K_BCD_PERSIST 6  /* the iterations of a large call in ONE launch (k_bcd_p: U updates + V updates) */
#define LRF_K_PLANES_GRAM 7  /* rgb -> patch matrices + the luma planes' exact Gram partials in one kernel (k_planes16_gram: large calls) */
#define LRF_K_COUNT 8
int lrf_ctx_profile(lrf_ctx* ctx, int enable);
/* The same for a subset of the kernels: bit (1 << LRF_K_x) per kernel id, 0 = off.  An event pair costs a few
 * microseconds of stream time per launch (0.15 ms per 22-launch encode when every kernel is timed); bench.py times
 * only the dominant kernel inside its timed region. */
int lrf_ctx_profile_kernels(lrf_ctx* ctx, unsigned mask);
int lrf_ctx_kernel_time(lrf_ctx* ctx, int kernel_id, double* total_ms, long* launches);
int lrf_ctx_profile_reset(lrf_ctx* ctx);

/* ---- memory helpers (for hosts with no GPU runtime of their own) --------------------------- */
int lrf_malloc(lrf_ctx* ctx, size_t bytes, void** out_dev);
int lrf_free(lrf_ctx* ctx, void* dev);
int lrf_memcpy_h2d(lrf_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* synchronous */
int lrf_memcpy_d2h(lrf_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* synchronous */

/* ---- geometry ------------------------------------------------------------------------------
 * Plane c (0 = Y, 1 = Cb, 2 = Cr) of an H x W image under the default qmf_encode branch
 * (color_space="YCbCr", scale_factor=(0.5,0.5), patch 8x8): size after chroma down-sampling
 * (lrf/compression/qmf.py:230), after reflect padding (lrf/compression/utils.py:108-132) and the
 * number of patches M (lrf/compression/qmf.py:43-56). */
int lrf_plane_dims(int64_t H, int64_t W, int c, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M);

/* ---- the hot path -------------------------------------------------------------------------- */

/*
 * uint8 RGB images [B,3,H,W] -> patch matrices.  Replaces, for every image, image.float();
 * rgb_to_ycbcr; chroma_downsampling(mode="area"); pad_image(reflect); patchify:
 * lrf/compression/qmf.py:227-242, lrf/compression/utils.py:24-47,76-95,108-132.
 * X: per image the three matrices back to back, [M_Y,64] [M_Cb,64] [M_Cr,64], fp32 row-major
 * (image stride = (M_Y + M_Cb + M_Cr) * 64 floats).
 */
int lrf_qmf_planes_from_rgb_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, float* X);

/*
 * QMF(rank=R, num_iters=K, bounds=(lo,hi), factor=(0,1)).decompose(x) for a batch of B matrices of
 * one shape: replaces the call sites lrf/compression/qmf.py:190,209,257,281 and the body
 * lrf/factorization/qmf.py:197-214 (SVDInit :42-71, CoordinateDescent.update_u/update_v :93-139,
 * QMF._project :191-195).
 *   X    [B,M,N] fp32, K >= 1.  N == LRF_PATCH_ELEMS with R <= LRF_MAX_RANK runs on the tuned kernels; every other shape
 *        (the patch sizes of experiments/ablation_patchsize/eval.py:49-55, patch=False) on the any-shape kernels
 *        (lrf_anyshape_kernels.hip; R <= LRF_ANY_MAX_RANK, min(M,N) <= LRF_ANY_MAX_SIDE for the initialisation)
 *   sign optional [B,R] int8 (NULL = default): sign imposed on sum_j (j+1) v0[j,r] of initial
 *        component r; 0 entries mean default (-1).  The reference's sign is LAPACK's arbitrary
 *        choice (SURVEY.md §7 hard part 1); passing the reference's signs reproduces its factors.
 *   U    [B,M,R] int8,  V [B,N,R] int8
 */
int lrf_qmf_decompose_f32(lrf_ctx* ctx, const float* X, int64_t B, int64_t M, int64_t N, int R, int K,
                          int lo, int hi, const int8_t* sign, int8_t* U, int8_t* V);

/*
 * The K block-coordinate-descent iterations alone, from caller-supplied initial factors: the loop
 * lrf/factorization/qmf.py:207-212 (CoordinateDescent.forward :149-164).  U0 [B,M,R], V0 [B,N,R] fp32.
 */
int lrf_qmf_bcd_f32(lrf_ctx* ctx, const float* X, int64_t B, int64_t M, int64_t N, int R, int K,
                    int lo, int hi, const float* U0, const float* V0, int8_t* U, int8_t* V);

/*
 * QMF.decompose in its general form — the class as the reference defines it, beyond what qmf_encode asks of it
 * (lrf/factorization/qmf.py:74-214; the reference's own smoke test, test/test_factorization.py:5-10, is
 * QMF(rank=5, num_iters=10): unbounded, all three factors).  Float factors out (an unbounded factor need not fit int8).
 *   bounded / lo / hi   QMF._project (:191-195): round, then clamp to [ceil(lo), floor(hi)] when bounded
 *   l2_u, l2_v, l1_ratio  the elastic-net terms of CoordinateDescent (:77-91, 154-157, 116-118; soft_thresholding
 *                       lrf/factorization/utils.py:36-40)
 *   factors             bit 0: update u, bit 1: update v, bit 2: update w — the affine pair of x ~ w0 + w1 u v^T: the u / v
 *                       updates then see safe_divide(x - w0, w1) (:104-105, utils.py:18-33) and update_w (:141-147) refits
 *                       (w0, w1) after them.  The reference solves that least-squares problem with torch.linalg.lstsq
 *                       (LAPACK); this library with the 2 x 2 normal equations in fp64: everything downstream of an updated
 *                       w is parity by tolerance (loss to 2e-4), everything else is the reference's bit for bit.
 *   U0 [B,M,R], V0 [B,N,R] fp32, both or neither: initial factors; NULL = this library's SVD initialisation (sign as in
 *                       lrf_qmf_decompose_f32).
 *   U [B,M,R], V [B,N,R], W [B,2] = (w0, w1) fp32, device memory.  eps is the class default 1e-16.  K = 0 returns the
 *   initial factors.  Runs on the any-shape kernels for every shape (R <= LRF_ANY_MAX_RANK).
 */
typedef struct lrf_qmf_opts {
    int bounded;
    float lo, hi;
    double l2_u, l2_v, l1_ratio;
    int factors;
    double eps;   /* CoordinateDescent(eps=...) (lrf/factorization/qmf.py:82, 90, 117-118); 0 selects the default 1e-16 */
    int w_init;   /* non-zero: W [B][2] holds the INITIAL pair (w0, w1) on entry — SVDInit(num_levels=...), qmf.py:56-68 —
                     and U0 / V0 the factors it belongs to; the u / v updates then see safe_divide(x - w0, w1) (qmf.py:104-105)
                     also when bit 2 of `factors` is clear.  Zero: [0; 1]. */
} lrf_qmf_opts;
int lrf_qmf_decompose_ex_f32(lrf_ctx* ctx, const float* X, int64_t B, int64_t M, int64_t N, int R, int K, const lrf_qmf_opts* opts,
                             const int8_t* sign, const float* U0, const float* V0, float* U, float* V, float* W);

/*
 * The initial factors alone (what SVDInit.forward returns, lrf/factorization/qmf.py:42-71):
 * u0 = U sqrt(s) [B,M,R], v0 = (sqrt(s) Vh)^T [B,N,R], fp32.  Also the arithmetic of svd_encode's
 * lrf/compression/svd.py:179-183 for N == LRF_PATCH_ELEMS.
 */
int lrf_qmf_svd_init_f32(lrf_ctx* ctx, const float* X, int64_t B, int64_t M, int64_t N, int R,
                         const int8_t* sign, float* U0, float* V0);

/*
 * QMF.loss(x, u, v, w) for a batch: the relative error ||x - (w0 + w1 u v^T)||_F / (||x||_F + 1e-16) per matrix —
 * lrf/factorization/qmf.py:225-227, relative_error lrf/factorization/utils.py:12-15 — which QMF(verbose=True) prints before every
 * iteration (qmf.py:208-210).  X [B,M,N], U [B,M,R], V [B,N,R] fp32; W [B,2] = (w0, w1) or NULL (= (0, 1)); loss [B] fp32.  The
 * sums are accumulated in fp64 (the reference: torch.norm in fp32; agreement to ~1e-6 relative).
 */
int lrf_qmf_loss_f32(lrf_ctx* ctx, const float* X, const float* U, const float* V, const float* W, int64_t B, int64_t M, int64_t N, int R,
                     float* loss);

/*
 * Fused encode of B images (default qmf_encode branch, everything between image.float() and the
 * byte container): lrf/compression/qmf.py:227-262.
 *   rgb  [B,3,H,W] uint8;  R[3] ranks of (Y, Cb, Cr);  sign optional [B, R[0]+R[1]+R[2]] int8
 *   U    per image [M_Y,R0] [M_Cb,R1] [M_Cr,R2] int8 back to back; V likewise with 64 rows each.
 */
int lrf_qmf_encode_rgb_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, const int R[3],
                          int K, int lo, int hi, const int8_t* sign, int8_t* U, int8_t* V);

/*
 * The same encode for ONE batch at Q rank triples in one call: the R-D sweep of the reference's experiments
 * (experiments/comparison/eval.py:83-110 calls qmf_encode per image and quality; BASELINE config 3: 24 images x quality 1..32).
 * What does not depend on the rank is computed once per image (patch matrices, exact Gram matrices), the SVD initialisation
 * once per (image, channel) at the largest rank asked for that channel — the lower ranks take its leading columns, which are
 * the same singular pairs bit for bit — and the BCD of all (triple, image) pairs runs as one call of Q x B matrices sets that
 * share X (large launches per rank family; the persistent kernel from 3584 blocks, 2304 for one rank family).  Every (triple, image) result is
 * byte-identical to lrf_qmf_encode_rgb_u8's for that triple.
 *   R     [Q][3] ranks (Y, Cb, Cr) per triple, each 1..32 (larger ranks: one lrf_qmf_encode_rgb_u8 call per triple)
 *   sign  optional [B][Rmax_Y + Rmax_Cb + Rmax_Cr] int8 (Rmax_c = the largest rank of channel c over the triples): the signs
 *         of the initial components; every triple uses the leading ones of its ranks
 *   U, V  for q = 0 .. Q-1 the factors of the B images at triple q back to back, each block in lrf_qmf_encode_rgb_u8's
 *         layout: U offset sum_{q' < q} B u_img(q'), u_img(q) = sum_c M_c R[q][c]; V likewise with 64 R[q][c]
 */
int lrf_qmf_encode_sweep_rgb_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int Q, const int* R, int K, int lo, int hi,
                                const int8_t* sign, int8_t* U, int8_t* V);

/*
 * Fused decode of B images: QMF.reconstruct, depatchify, unpad_image, chroma_upsampling(nearest),
 * ycbcr_to_rgb, to_dtype(uint8): lrf/compression/qmf.py:329-351, lrf/factorization/qmf.py:216-223,
 * lrf/compression/utils.py:50-73,98-105,135-182.  U, V laid out as lrf_qmf_encode_rgb_u8 writes them.
 */
int lrf_qmf_decode_rgb_u8(lrf_ctx* ctx, const int8_t* U, const int8_t* V, int64_t B, int64_t H, int64_t W,
                          const int R[3], uint8_t* rgb);

/* ---- the SVD baseline ------------------------------------------------- */

/*
 * svd_encode, default branch (color_space="RGB", patch 8x8, uint8 factors), everything between image.float() and the byte
 * container: pad_image(reflect) + patchify to X [M,192] (lrf/compression/svd.py:160-162), the top-R singular pairs
 * u = U sqrt(s), v = (sqrt(s) Vh)^T (:179-183; Gram matrix + eigen-solve instead of LAPACK, tolerance-checked) and
 * quantize(., uint8) of both (:185-187, lrf/compression/utils.py:185-220).
 *   U [B,M,R] uint8, V [B,192,R] uint8, qparams [B,4] float = (scale_u, min_u, scale_v, min_v), all device memory.
 *   sign optional [B,R] as in lrf_qmf_decompose_f32.
 */
int lrf_svd_encode_rgb_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int R, const int8_t* sign,
                          uint8_t* U, uint8_t* V, float* qparams);

/*
 * svd_decode, RGB branch: dequantize (lrf/compression/utils.py:223-243), u @ v.mT, depatchify, unpad_image, to_dtype(uint8)
 * (lrf/compression/svd.py:310-326,359).  qparams6 [B,6] float (device) = (scale_u, min_u, qmin_u, scale_v, min_v, qmin_v)
 * where qmin is the smallest stored code of the tensor (dequantize subtracts `q.min()`).
 */
int lrf_svd_decode_rgb_u8(lrf_ctx* ctx, const uint8_t* U, const uint8_t* V, int64_t B, int64_t H, int64_t W, int R,
                          const float* qparams6, uint8_t* rgb);

/* svd_encode's RGB branch for the other patch sizes, patch=False and float factors (lrf/compression/svd.py:157-193): the host
 * forms the matrices with lrf_qmf_rgbspace_matrix_u8 (the same X, svd.py:160-167), takes u = U sqrt(s), v = (sqrt(s) Vh)^T from
 * lrf_qmf_svd_init_f32 (any shape) and, unless dtype is a float type, quantises each factor tensor as a whole:
 *   lrf_quantize_u8: quantize(t, uint8) (lrf/compression/utils.py:185-220) of B tensors of `per` floats each:
 *                    scale = (max - min) / 255, q = clamp((t - min) / scale + 0, 0, 255) truncated; qparams [B][2] = (scale, min).
 *   lrf_svd_decode_any_u8: svd_decode of those streams (svd.py:310-326, 359); U / V uint8 with qparams6 [B][6] = (scale_u, min_u,
 *                    qmin_u, scale_v, min_v, qmin_v), or float factors (factors_are_float, qparams6 NULL); layouts as
 *                    lrf_qmf_rgbspace_decode_any_u8.
 * The YCbCr branch of svd_encode is not built: in the reference it raises TypeError for an integer rank (svd.py:234, 267) and
 * its streams do not decode ("padded size" is appended twice per plane, svd.py:226,237, so svd_decode reads the wrong entry). */
int lrf_quantize_u8(lrf_ctx* ctx, const float* T, int64_t B, int64_t per, uint8_t* Q, float* qparams);
int lrf_svd_decode_any_u8(lrf_ctx* ctx, const void* U, const void* V, int factors_are_float, int64_t B, int64_t H, int64_t W, int p, int q,
                          int R, const float* qparams6, uint8_t* rgb);

/* ---------------------------------------------------------------------------------------------------
 * QMF, RGB colour-space branch: qmf_encode(color_space="RGB", patch=True, patch_size=(8,8))
 * (lrf/compression/qmf.py:164-187).  One matrix X [M,192] per image (reflect padding to multiples of 8, rows =
 * patches in row-major order, columns = (c, p, q)), rank R = max(round(min(M,192) * quality / 100), 1) chosen by
 * the host, QMF(rank=R, bounds, factor=(0,1)).decompose, int8 factors U [B,M,R], V [B,192,R] (device memory).
 *   U0 [B,M,R] / V0 [B,192,R] fp32 (device), both or neither: the initial factors (lrf/factorization/qmf.py:42-71);
 *   NULL = this library's SVD initialisation (sign optional [B,R] as in lrf_qmf_decompose_f32).
 * Ranks up to 192 (the reference's colour-space ablation sweeps quality 0..10 -> R <= 19), K >= 1.
 * The factorisation runs on the any-shape kernels (lrf_anyshape_kernels.hip).
 */
int lrf_qmf_rgbspace_encode_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int R, int K, int lo, int hi,
                               const int8_t* sign, const float* U0, const float* V0, int8_t* U, int8_t* V);

/* qmf_decode, RGB colour-space branch (lrf/compression/qmf.py:311-323, :351): u @ v.mT, depatchify, unpad_image,
 * to_dtype(uint8).  rgb [B,3,H,W] uint8 (device). */
int lrf_qmf_rgbspace_decode_u8(lrf_ctx* ctx, const int8_t* U, const int8_t* V, int64_t B, int64_t H, int64_t W, int R, uint8_t* rgb);

/* The RGB colour-space branch for the other patch sizes and for patch=False (lrf/compression/qmf.py:164-212, decode :309-323).
 * The host forms the matrices with lrf_qmf_rgbspace_matrix_u8, factorises them with lrf_qmf_decompose_f32 /
 * lrf_qmf_bcd_f32 / lrf_qmf_svd_init_f32 (any shape) and decodes with lrf_qmf_rgbspace_decode_any_u8.
 *   p, q > 0: X [B, M, 3 p q] — reflect-padded image, rows = patches, columns in (c, a, b) order (qmf.py:167-169);
 *   p = q = 0 (patch=False): X [B, 3, H, W], the channel planes as three matrices per image (qmf.py:193-194), factors
 *   U [B,3,H,R], V [B,3,W,R].  lrf_rgbspace_dims_any: padded size and the matrix shape [M, N] (p = 0: M = H, N = W). */
int lrf_rgbspace_dims_any(int64_t H, int64_t W, int p, int q, int64_t* hp, int64_t* wp, int64_t* M, int64_t* N);
int lrf_qmf_rgbspace_matrix_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int p, int q, float* X);
int lrf_qmf_rgbspace_decode_any_u8(lrf_ctx* ctx, const int8_t* U, const int8_t* V, int64_t B, int64_t H, int64_t W, int p, int q, int R,
                                   uint8_t* rgb);

/* ---------------------------------------------------------------------------------------------------
 * YCbCr branch with any patch size, or none: qmf_encode(color_space="YCbCr", patch_size=(p,q)) and patch=False
 * (lrf/compression/qmf.py:227-262 and :264-286; swept by experiments/ablation_patchsize/eval.py:49-55).
 * The host forms the matrices of one plane, factorises them with lrf_qmf_decompose_f32 (any M, N, R) and decodes with
 * lrf_qmf_decode_any_u8.  p = q = 0 means patch=False: the matrix is the plane itself, [h, w].
 */

/* plane ch (0 = Y, 1 = Cb, 2 = Cr): size after chroma down-sampling, after reflect padding to multiples of (p,q)
 * (lrf/compression/utils.py:108-132), and the matrix shape [M, N] (N = p*q; p = 0: M = h, N = w) */
int lrf_plane_dims_any(int64_t H, int64_t W, int p, int q, int ch, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M,
                       int64_t* N);

/* rgb [B,3,H,W] uint8 -> X [B,M,N] fp32 for plane ch: rgb_to_ycbcr, chroma_downsampling(area), pad_image(reflect),
 * patchify (qmf.py:227-242 with patch_size = (p,q); :264-269 when p = 0).  Device pointers. */
int lrf_qmf_planes_any_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int p, int q, int ch, float* X);

/* qmf_decode of that branch (qmf.py:325-351): per plane u @ v.mT, depatchify + unpad_image (p > 0), nearest chroma
 * up-sampling, ycbcr_to_rgb, to_dtype(uint8).  Uc [B,M_c,R_c], Vc [B,N_c,R_c] int8 per plane c; rgb [B,3,H,W] uint8. */
int lrf_qmf_decode_any_u8(lrf_ctx* ctx, const int8_t* U0, const int8_t* V0, const int8_t* U1, const int8_t* V1, const int8_t* U2,
                          const int8_t* V2, int64_t B, int64_t H, int64_t W, int p, int q, const int R[3], uint8_t* rgb);

/* ---------------------------------------------------------------------------------------------------
 * Host -> host pipelined encoder.  This is the protocol a caller of the reference experiences — a host tensor goes in,
 * the encoded factors come back on the host (lrf/utils/misc.py:90-100 times exactly that around `encoder(image)`;
 * experiments/comparison/eval.py:105-110 loops it over a dataset) — and the metric SURVEY.md section 8(d) defines.
 * A pipe owns `slots` encoder contexts (stream + scratch + device staging each).  A batch is cut into sub-batches of
 * `sub_batch` images (0 = chosen by the library: ~40 MB of input); sub-batch i runs on slot i % slots as
 *     H2D(rgb) -> lrf_qmf_encode_rgb_u8 -> D2H(U, V)
 * — the uploads of all sub-batches in order on one upload stream (each gets the whole link, the first lands early), the
 * kernels and the download on the slot's stream — so uploads, kernels and downloads of different sub-batches overlap.
 * Two slots are the measured optimum on MI355X (three or more streams of kernels plus the upload stream exceed the HIP
 * runtime's default of four hardware queues, and streams that share a queue serialise).
 *   rgb_host [B,3,H,W] uint8, U_host / V_host as lrf_qmf_encode_rgb_u8 lays them out, sign_host optional
 *   [B, R[0]+R[1]+R[2]] int8: HOST pointers.  Page-locked memory (lrf_host_alloc, lrf_host_register, or torch's
 *   pin_memory) is what lets the copies run asynchronously at link speed; pageable memory works, slowly.
 * The results equal lrf_qmf_encode_rgb_u8's on the same images bit for bit (images are independent).
 * A pipe is not thread-safe: one per (host thread, device), which is also the multi-GPU model (one process per GPU).
 */
typedef struct lrf_pipe lrf_pipe;
int lrf_pipe_create(int device, int slots /* 1..8 */, int64_t sub_batch /* images, 0 = auto */, lrf_pipe** out);
void lrf_pipe_destroy(lrf_pipe* pipe);
int lrf_pipe_slots(const lrf_pipe* pipe);
/* the encoder context of one slot (owned by the pipe): for lrf_ctx_profile* / lrf_ctx_kernel_time */
lrf_ctx* lrf_pipe_slot_ctx(lrf_pipe* pipe, int slot);
size_t lrf_pipe_workspace_bytes(const lrf_pipe* pipe);
/* whole batch, returns when every factor is in U_host / V_host */
int lrf_pipe_qmf_encode_rgb_u8_host(lrf_pipe* pipe, const uint8_t* rgb_host, int64_t B, int64_t H, int64_t W, const int R[3],
                                    int K, int lo, int hi, const int8_t* sign_host, int8_t* U_host, int8_t* V_host);
/* the same in two steps, so that the host can pack the container of finished sub-batches (lrf/compression/utils.py:354-455)
 * while the GPU works on the next ones: submit enqueues everything and returns the number of sub-batches; each
 * lrf_pipe_wait_next blocks until the next sub-batch (in order) is on the host and reports its image range
 * (n_images = 0: nothing left).  The host buffers must stay valid until the last wait returns. */
int lrf_pipe_qmf_encode_submit(lrf_pipe* pipe, const uint8_t* rgb_host, int64_t B, int64_t H, int64_t W, const int R[3], int K,
                               int lo, int hi, const int8_t* sign_host, int8_t* U_host, int8_t* V_host, int* n_sub);
int lrf_pipe_wait_next(lrf_pipe* pipe, int64_t* first_image, int64_t* n_images);

/* page-locked host memory for the pipe's callers (hosts without a GPU runtime of their own) */
int lrf_host_alloc(size_t bytes, void** out_host);
int lrf_host_free(void* host);
int lrf_host_register(void* host, size_t bytes);   /* page-locks memory the caller already owns */
int lrf_host_unregister(void* host);

/* The same three with an explicit chroma plane size hc x wc, for scale_factor other than (0.5, 0.5) (lrf/compression/qmf.py:230:
 * F.interpolate(scale_factor, mode="area") produces floor(H * s_h) x floor(W * s_w), which the host computes; the pooling
 * windows follow from the two sizes alone, lrf/compression/utils.py:92-94; decode :346-348 up-samples to the luma size with
 * mode="nearest").  hc, wc <= 0: the default halves. */
int lrf_plane_dims_any_hw(int64_t H, int64_t W, int64_t hc, int64_t wc, int p, int q, int ch, int64_t* h, int64_t* w, int64_t* hp,
                          int64_t* wp, int64_t* M, int64_t* N);
int lrf_qmf_planes_any_hw_u8(lrf_ctx* ctx, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int64_t hc, int64_t wc, int p, int q, int ch,
                             float* X);
int lrf_qmf_decode_any_hw_u8(lrf_ctx* ctx, const int8_t* U0, const int8_t* V0, const int8_t* U1, const int8_t* V1, const int8_t* U2,
                             const int8_t* V2, int64_t B, int64_t H, int64_t W, int64_t hc, int64_t wc, int p, int q, const int R[3],
                             uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif /* LRF_HIP_H */
