/*
 * lrf_pack.h — C ABI of liblrf_pack.so: the reference's byte container for QMF factor sets, packed on host threads.
 *
 * Restates lrf/compression/utils.py:246-300 (combine_bytes: left fold of len32_be(p1) || p1 || p2),
 * :354-390 (encode_matrix, mode "col": every column zlib-compressed at level 9 behind the JSON header
 * {"num_fibers": R, "mode": "col", "dtype": "int8"}) and the stream layout of lrf/compression/qmf.py:288-290.
 * Host memory only; zlib is the system libz (the one CPython's zlib module uses), so the bytes equal the reference's.
 */
#ifndef LRF_PACK_H
#define LRF_PACK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/*
 * Packs B images.  Image b's factors: U + b*u_stride holds [M0,R0] [M1,R1] [M2,R2] int8 row-major back to back,
 * V + b*v_stride three [64,R_c] matrices (the layout lrf_qmf_encode_rgb_u8 writes).  `metadata` is the UTF-8 JSON of
 * the image metadata (identical for every image of the batch).  On return out[b] points to a malloc'ed stream of
 * out_len[b] bytes (release with lrf_pack_free).  threads <= 0: one per hardware thread, at most 64.
 * Returns 0, or a negative value (-1 bad argument, -4 out of memory, -5 zlib error).
 */
int lrf_pack_qmf_streams(const int8_t* U, int64_t u_stride, const int8_t* V, int64_t v_stride, int64_t B,
                         const int64_t M[3], const int R[3], const char* metadata, int64_t metadata_len,
                         int threads, uint8_t** out, int64_t* out_len);
void lrf_pack_free(uint8_t* p);

/*
 * The same for the other patch sizes and for patch=False (lrf/compression/qmf.py:232-286): F[f], f = u_Y, v_Y, u_Cb, v_Cb,
 * u_Cr, v_Cr, holds [B][rows[f]][cols[f]] int8.  whole == 0: every factor as encode_matrix does (per-column zlib-9,
 * utils.py:354-390); whole != 0: the patch=False form, where the reference's factors keep the plane's channel axis and
 * encode_tensor (utils.py:429-455) compresses each 3-D array [1, rows, cols] in one piece behind {"shape": ..., "dtype": "int8"}.
 * Output and return values as lrf_pack_qmf_streams.
 */
int lrf_pack_qmf_streams_planes(const int8_t* const F[6], const int64_t rows[6], const int cols[6], int64_t B, int whole,
                                const char* metadata, int64_t metadata_len, int threads, uint8_t** out, int64_t* out_len);

/*
 * The reverse for decoding (lrf/compression/utils.py:393-426, decode_matrix; qmf.py:306-327): factor_blobs[b] is the second
 * payload of stream b — combine_bytes of the six encoded matrices u_Y, v_Y, u_Cb, v_Cb, u_Cr, v_Cr — of blob_len[b] bytes.
 * Writes image b's factors to U + b*u_stride ([M0,R0] [M1,R1] [M2,R2] int8 row-major back to back) and V + b*v_stride
 * (three [64,R_c]), the layout lrf_qmf_decode_rgb_u8 reads.  Every length is checked against the blob and against (M, R).
 * Returns 0; -1 bad argument; -6 when a blob is not exactly that layout (another dtype or mode, other shapes, truncated or
 * corrupt data): the caller then parses the stream itself to say what is wrong.
 */
int lrf_pack_unpack_qmf_factors(const uint8_t* const* factor_blobs, const int64_t* blob_len, int64_t B, const int64_t M[3],
                                const int R[3], int threads, int8_t* U, int64_t u_stride, int8_t* V, int64_t v_stride);
const char* lrf_pack_zlib_version(void);

#ifdef __cplusplus
}
#endif
#endif
