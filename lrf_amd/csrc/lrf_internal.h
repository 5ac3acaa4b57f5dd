// lrf_internal.h — structures shared by the kernels and the host side of liblrf_hip.so.
#ifndef LRF_INTERNAL_H
#define LRF_INTERNAL_H
#include <stdint.h>

#define LRF_FAM_OF_RANK(R) ((R) <= 8 ? 0 : ((R) <= 16 ? 1 : 2)) // kernel family of a rank: k_bcd_w / k_bcd_w16 / k_bcd_w32 (+ workgroup kernels)
#define LRF_RP 16    // rank padded to one 16-wide MFMA tile (== LRF_MAX_RANK)
#define LRF_KC 384   // rows per X^T U reduction block (the reference's MKL K-blocking)

// Table layout ("gt") of b = v.mT @ v at rank pitch 16, LRF_GT_LD floats per column r, so that one column's operands are
// contiguous:  gt[r*LD + n], n < R-1 : b[j_n][r] for the j != r in increasing order (the `bb` vector of qmf.py:114);
//   gt[r*LD + 16] = 1/den[r],  gt[r*LD + 17] = den[r] = (b[r][r] + 0) + eps
#define LRF_GT_LD 20
#define LRF_GT_RDEN 16
#define LRF_GT_DEN 17
#define LRF_GT_STRIDE (LRF_RP * LRF_GT_LD)
// ranks above 16: rank pitch 64 (lrf_bigrank_kernels.hip)
#define LRF_RPB 64                      // padded rank
#define LRF_GTB_LD 68                   // gt table pitch: <= 63 `bb` entries, [64] = 1/den (unused here), [65] = den
#define LRF_GTB_DEN 65
#define LRF_GTB_STRIDE (LRF_RPB * LRF_GTB_LD)
// the exact Gram pass (lrf_gram_kernels.hip)
#define LRF_GRAM_ROWS 1536            // rows per chunk: 24 blocks of 64
#define LRF_GRAM_ROWS_FUSED 1536 // rows per Gram chunk of a plane whose partials k_planes16_gram computes
#define LRF_GRAM_PAIRS 10             // upper-triangle pairs of the four 16-column tiles
#define LRF_GRAM_SLOT (LRF_GRAM_PAIRS * 256) // 128-bit sums per partial, [pair][reg][lane]
#define LRF_GRAM_EXP_FROM_DATA (-100000)

// geometry of one colour plane of an image (lrf/compression/qmf.py:230-242)
struct PlaneGeom {
    int h, w;          // plane size (chroma: floor(H/2), floor(W/2))
    int hp, wp;        // after reflect padding to multiples of 8
    int top, left;     // pad_top / pad_left == unpad start (utils.py:127-130, :148-150)
    int top_crop, left_crop;
    int nw;            // patches per row
    int nh;            // patch rows
    int pr0;           // first patch-row index of this plane inside the image (k_planes grid)
    int M;             // number of patches
    long xoff;         // floats from the image's X base
    long o4;           // first float4 output index of this plane inside the image
};
struct ImageGeom {
    PlaneGeom p[3];
    long tot4;         // float4 outputs per image
    long img_floats;   // floats per image in X
};

// one matrix to factorise
struct PlaneDesc {
    long x_off;        // floats from the X base
    long u_off;        // elements from the U (int8) base
    long v_off;        // elements from the V (int8) base
    long u0_off;       // elements from the fp32 U0 base (init in/out)
    long v0_off;       // elements from the fp32 V0 base
    int M, R;
    int blk0, nblk;    // slots in the X^T U partial table
    int native_t2_u;   // ATen native order for `uu @ bb` in update_u: (R-1)*M < 400
    int sign_off;      // offset into the sign vector, or -1
    int gch0, ngch;    // slots in the Gram partial table (lrf_gram_kernels.hip): one per chunk of LRF_GRAM_ROWS rows
    int init_src;      // the plane whose SVD initialisation this plane takes its first R columns from: itself, or — in a sweep
                       // call (lrf_qmf_encode_sweep_rgb_u8) — the plane of the same matrix X with the call's largest rank
    int gram_fused;    // 1: a luma plane whose Gram partials the planes kernel itself computes (k_planes16_gram); k_gram64 skips it
};
struct BlockDesc {
    int plane;         // index into the PlaneDesc table
    int row0;          // first row of the block
    int blk;           // block number inside the plane
    int pad;
};
struct GramChunk {
    int plane; // index into the PlaneDesc table
    int row0;  // first row of the chunk
    int slot;  // partial slot (pd.gch0 + chunk number)
    int nrows;
};

struct GsParams {
    float lo, hi;      // clamp
    float flimit;      // |q~| >= flimit: certainly outside [lo,hi] after rounding
    float fthr;        // |q~ - rint(q~)| <= fthr: rint(q~) == rint(fl(num/den))
    int exact_int;     // iterations >= 2 only: every term and partial sum of `uu @ bb` is an exact integer in fp32 for the
                       // call's largest rank and bounds ((R-1) 64 mx^3 < 2^24), so the order of that sum is immaterial
};
#endif
