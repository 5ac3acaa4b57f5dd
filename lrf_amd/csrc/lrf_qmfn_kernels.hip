// lrf_qmfn_kernels.hip — decode of the RGB colour-space branch of qmf_encode (lrf/compression/qmf.py:164-187, 311-323):
// one matrix X [M, 192] per image (three 8x8 colour patches per row).  The encode side has no kernels of its own any more:
// the matrices are formed by k_patchify_rgb (lrf_svd_kernels.hip) and factorised by the any-shape kernels
// (lrf_anyshape_kernels.hip), which beat the dedicated VALU kernels that used to live here by a factor of two.
// Included by lrf_api.hip.

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
