// lrf_planes_gram.hip — the patch matrices of a large encode call together with the luma planes' exact Gram partials
// (k_planes16_gram, lrf_planes_gram_kernel.hip): when a call takes it, and its launch.  A translation unit of its own because it is
// compiled without SLP vectorisation (Makefile: packed fp32 math issues slower than the scalars it replaces, and this kernel is
// bound by what it issues: 243 -> 218 us) — a flag that costs the workgroup kernels of lrf_encode8.hip 9 % (64 images at ranks
// (20,10,10): 1.19 -> 1.30 ms, tools/run_r05_w.sh).
#include "lrf_host.h"
#define LRF_GRAM_DEVICE_ONLY // (k_gram_exponent lives in lrf_encode8.hip's copy of this file; a unity build has included it already)
#include "lrf_gram_kernels.hip"
#include "lrf_planes_gram_kernel.hip"

// Whether an encode call forms its patch matrices with k_planes16_gram (lrf_planes_gram_kernel.hip): the sizes and alignment
// k_planes16 asks for, and a batch of two full rounds of that kernel's workgroups or more (1024 chunks of LRF_GRAM_ROWS_FUSED luma
// rows = 256 x 512x768).  The kernel is bound by the SUM of what its two halves issue — on a SIMD the int8 MFMAs of the Gram
// blocks and the vector instructions of the colour conversion and the digit extraction do not overlap — so all it saves is the
// re-read of the luma matrices: 256 x 512x768: 218 + 67 us (the chroma planes' k_gram64) against 146 + 183; below that size the
// two-kernel form is as fast or faster (128 images: +15 us), tools/run_r05_p.sh / _q.sh / _r.sh.
// The caller marks the luma planes that compute an initialisation `gram_fused` before the tables are uploaded, and calls
// planes_gram_from_rgb instead of lrf_qmf_planes_from_rgb_u8.
bool planes_gram_eligible(const uint8_t* rgb, int64_t B, int64_t H, int64_t W)
{
    static const bool off = dev_flag("LRF_NO_FUSED_GRAM") || dev_flag("LRF_PLANES_NO_TILED");
    static const long min_chunks = env_long("LRF_FUSED_GRAM_MIN_CHUNKS", 1024); // test hook (lrf_env.h): 1 = every eligible call
    if (off || H % 16 != 0 || W % 16 != 0 || (reinterpret_cast<uintptr_t>(rgb) & 7) != 0 || (long)H * W * 3 >= (1L << 31)) return false;
    return min_chunks <= 1 || B * (H / 8) * (W / 8) >= min_chunks * LRF_GRAM_ROWS_FUSED; // (luma rows of the call)
}
int planes_gram_from_rgb(lrf_ctx* c, const uint8_t* rgb, int64_t H, int64_t W, const ImageGeom& g, const Tables& t, float* X)
{
    const int nfused = (int)t.gchunks.size() - t.ngram_rest;
    if (nfused < 1) return set_err(LRF_EINVAL, "internal: no plane is marked for k_planes16_gram");
    Prof p(c, LRF_K_PLANES_GRAM);
    hipLaunchKernelGGL(k_planes16_gram, dim3((unsigned)nfused), dim3(256), 0, c->stream, rgb, (int)H, (int)W, g, X, (const PlaneDesc*)c->planes.p,
                       (const GramChunk*)c->gchunks.p + t.ngram_rest, (ulonglong2*)c->gpart.p);
    LAUNCH_CHECK();
    return LRF_OK;
}

