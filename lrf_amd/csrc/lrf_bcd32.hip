// lrf_bcd32.hip — the BCD kernels of ranks 17..32 of the 64-column path (the reference's quality sweep beyond quality 25,
// experiments/comparison/eval.py:83) behind launch functions (lrf_host.h): the b table of the initial V (k_bprep_big), the U
// update by k_bcd_w32 (iterations >= 2, exact-integer bounds), k_bcd_w32f (the first iteration) or the workgroup kernel
// k_bcd_mid (small runs, wide bounds, caller-supplied U0), the V update (k_vupdate_mid).  lrf/factorization/qmf.py:93-139.
#include "lrf_host.h"
#include "lrf_gs.h"
#include "lrf_bigrank_kernels.hip"
#include "lrf_midrank_kernels.hip"
#include "lrf_bcdw32_kernel.hip"

static int bcd32_attrs(lrf_ctx* c)
{
    if (c->attr_done & (1u << 2)) return LRF_OK;
    HIP_TRY(hipFuncSetAttribute((const void*)k_vupdate_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BigVLds)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_mid<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MidLds<0>)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_mid<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MidLds<1>)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_mid<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MidLds<2>)));
#define LRF_W32_ATTR(NP) HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w32<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW32_LDS(NP)))
    LRF_W32_ATTR(9); LRF_W32_ATTR(10); LRF_W32_ATTR(11); LRF_W32_ATTR(12); LRF_W32_ATTR(13); LRF_W32_ATTR(14); LRF_W32_ATTR(15); LRF_W32_ATTR(16);
#undef LRF_W32_ATTR
#define LRF_W32F_ATTR(R) HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w32f<R>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW32F_LDS))
    LRF_W32F_ATTR(17); LRF_W32F_ATTR(18); LRF_W32F_ATTR(19); LRF_W32F_ATTR(20); LRF_W32F_ATTR(21); LRF_W32F_ATTR(22); LRF_W32F_ATTR(23); LRF_W32F_ATTR(24);
    LRF_W32F_ATTR(25); LRF_W32F_ATTR(26); LRF_W32F_ATTR(27); LRF_W32F_ATTR(28); LRF_W32F_ATTR(29); LRF_W32F_ATTR(30); LRF_W32F_ATTR(31); LRF_W32F_ATTR(32);
#undef LRF_W32F_ATTR
    c->attr_done |= 1u << 2;
    return LRF_OK;
}

int bcd32_bprep(hipStream_t rs, const PlaneDesc* pl, const float* vf, float* bf, int nplanes, int plane0)
{
    hipLaunchKernelGGL(k_bprep_big, dim3(nplanes), dim3(256), 0, rs, pl, vf, bf, plane0);
    LAUNCH_CHECK();
    return LRF_OK;
}

// The wave kernels take a run whose planes all have ranks 17..32, from LRF_BCDW32_MIN_BLOCKS blocks on:
//   mode 0 (iterations >= 2): exact-integer bounds ((R - 1) 64 mx^3 < 2^24) with |b| within int16 (64 mx^2 <= 32767): the
//           lane = row Gauss-Seidel on int16 pairs (k_bcd_w32);
//   mode 1 (the first iteration, old U = X W0): one rank for the whole run and no plane small enough for ATen's native
//           order of `uu @ bb` (k_bcd_w32f).
bool bcd32_wave_kernels_apply(const FamRun& r, bool exact_int, long mx_b, int mode)
{
    static const bool w32_off = dev_flag("LRF_NO_BCDW32"); // dev build: k_bcd_mid instead
    static const long w32_min = env_long("LRF_BCDW32_MIN_BLOCKS", LRF_BCDW32_MIN_BLOCKS); // test hook (lrf_env.h)
    if (r.fam != 2 || !bcd_wave_variant() || w32_off || r.rmin < 17 || r.nblocks < w32_min) return false;
    if (mode == 0) return exact_int && 64 * mx_b * mx_b <= 32767;
    if (mode == 1) return !r.any_native && r.rmin == r.rmax;
    return false;
}

int bcd32_update_u(lrf_ctx* c, hipStream_t rs, const BcdLaunch& a, const FamRun& r, long mx_b)
{
    int rc = bcd32_attrs(c);
    if (rc) return rc;
    const int nbr = a.nblocks;
    if (a.mode == 0 && bcd32_wave_kernels_apply(r, a.gp.exact_int != 0, mx_b, 0)) {
#define LRF_LAUNCH_W32(NP)                                                                                           \
    hipLaunchKernelGGL((k_bcd_w32<NP>), dim3((nbr + LRF_BCDW32_WAVES - 1) / LRF_BCDW32_WAVES), dim3(64 * LRF_BCDW32_WAVES), LRF_BCDW32_LDS(NP), \
                       rs, a.X, a.pl, a.bl, a.vf, a.bf, a.U, a.pp, a.qp, a.gp, nbr)
        switch ((r.rmax + 1) >> 1) {
        case 9: LRF_LAUNCH_W32(9); break;
        case 10: LRF_LAUNCH_W32(10); break;
        case 11: LRF_LAUNCH_W32(11); break;
        case 12: LRF_LAUNCH_W32(12); break;
        case 13: LRF_LAUNCH_W32(13); break;
        case 14: LRF_LAUNCH_W32(14); break;
        case 15: LRF_LAUNCH_W32(15); break;
        default: LRF_LAUNCH_W32(16); break;
        }
#undef LRF_LAUNCH_W32
    } else if (a.mode == 1 && bcd32_wave_kernels_apply(r, a.gp.exact_int != 0, mx_b, 1)) {
#define LRF_LAUNCH_W32F(RR)                                                                                          \
    case RR:                                                                                                         \
        hipLaunchKernelGGL((k_bcd_w32f<RR>), dim3(nbr), dim3(64), LRF_BCDW32F_LDS, rs, a.X, a.pl, a.bl, a.vf, a.wf, a.bf, a.U, a.pp, a.qp, a.gp, nbr); \
        break;
        switch (r.rmax) {
            LRF_LAUNCH_W32F(17) LRF_LAUNCH_W32F(18) LRF_LAUNCH_W32F(19) LRF_LAUNCH_W32F(20) LRF_LAUNCH_W32F(21) LRF_LAUNCH_W32F(22)
            LRF_LAUNCH_W32F(23) LRF_LAUNCH_W32F(24) LRF_LAUNCH_W32F(25) LRF_LAUNCH_W32F(26) LRF_LAUNCH_W32F(27) LRF_LAUNCH_W32F(28)
            LRF_LAUNCH_W32F(29) LRF_LAUNCH_W32F(30) LRF_LAUNCH_W32F(31) LRF_LAUNCH_W32F(32)
        }
#undef LRF_LAUNCH_W32F
    } else {
#define LRF_LAUNCH_MID(MODE)                                                                                         \
    hipLaunchKernelGGL((k_bcd_mid<MODE>), dim3(nbr), dim3(256), sizeof(MidLds<MODE>), rs, a.X, a.pl, a.bl, a.vf, a.wf, a.bf, a.U0, a.U, a.pp, \
                       a.qp, a.gp)
        if (a.mode == 1) LRF_LAUNCH_MID(1);
        else if (a.mode == 2) LRF_LAUNCH_MID(2);
        else LRF_LAUNCH_MID(0);
#undef LRF_LAUNCH_MID
    }
    LAUNCH_CHECK();
    return LRF_OK;
}

int bcd32_update_v(lrf_ctx* c, hipStream_t rs, const PlaneDesc* pl, const float* pp, const float* qp, float* vf, float* bf, int8_t* V, float lo,
                   float hi, int last, int nplanes, int plane0)
{
    int rc = bcd32_attrs(c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_vupdate_mid, dim3(nplanes), dim3(256), sizeof(BigVLds), rs, pl, pp, qp, vf, bf, V, lo, hi, last, plane0);
    LAUNCH_CHECK();
    return LRF_OK;
}
