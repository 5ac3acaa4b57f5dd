// lrf_bcd_persist.hip — the iterations of a large call of the 64-column path in ONE launch (k_bcd_p<F16, NP32, FIRST>,
// lrf_bcdp_kernel.hip: iterations 2..K, or all K at ranks <= 16): which calls take it (bcdp_plan) and its launch (bcdp_launch:
// queue state, error word, grid).
// lrf/factorization/qmf.py:93-139, 197-214 (the num_iters loop).
#include "lrf_host.h"
#include "lrf_gs.h"
#include "lrf_bcdw_kernel.hip"
#include "lrf_bcdw16_kernel.hip"
#include "lrf_bcdw32_kernel.hip"
#include "lrf_bcdp_kernel.hip"

// The persistent kernel takes a call when
//   * the device is the part its in-launch hand-offs were validated on (gfx950);
//   * every plane sits on the kernel family of its own rank (the split plan of plan_runs: calls of 1024 blocks or more, 256
//     with a rank above 16) and the families' exact-integer conditions hold for iterations >= 2 — ranks 9..16: (R - 1) 64 mx^3 <
//     2^24; ranks 17..32: that and 64 mx^2 <= 32767, one pair count NP for all such planes; no plane of ranks above 8 small
//     enough for ATen's native order of `uu @ bb`;
//   * the call has LRF_PERSIST_MIN_BLOCKS blocks or more (256 x 512x768 at ranks <= 8: 2.05 -> 1.92 ms per step; 48 / 64 such
//     images lose 15 %: a round and a half of the 2048 wave slots; tools/dev_persist_threshold.py) — from
//     LRF_PERSIST_MIN_BLOCKS_ONE_FAMILY when all planes are of one rank family (96 x 512x768: (7,3,3) 0.96 -> 0.92 ms, (12,12,12)
//     1.25 -> 1.13, (20,20,20) 1.83 -> 1.72; calls that mix families lose at 96 and 128 images: tools/run_r05_o.sh).
//     LRF_PERSIST=0 turns it off, =1 lowers the threshold to LRF_BCDW_MIN_BLOCKS (tests).
PersistPlan bcdp_plan(lrf_ctx* c, const std::vector<FamRun>& runs, int K, int lo, int hi)
{
    PersistPlan pp;
    static const int persist_env = (int)env_long("LRF_PERSIST", -1); // test hook (lrf_env.h)
    static const bool exact_off = dev_flag("LRF_GENERIC_GS");
    if (!c->persist_arch || persist_env == 0 || !bcd_wave_variant() || K < 2 || runs.empty()) return pp;
    const long mx = abs(lo) > abs(hi) ? abs(lo) : abs(hi);
    long nblocks = 0;
    bool f16 = false;
    int np32 = 0;
    for (const FamRun& r : runs) {
        if (fam_of_rank(r.rmin) != r.fam || fam_of_rank(r.rmax) != r.fam) return pp; // a small call: one family for all planes
        nblocks += r.nblocks;
        if (r.fam == 0) continue;
        const bool exact = !exact_off && (long)(r.rmax - 1) * 64 * mx * mx * mx < (1L << 24);
        if (!exact || r.any_native) return pp;
        if (r.fam == 1) {
            f16 = true;
        } else {
            const int np = (r.rmax + 1) >> 1;
            if (64 * mx * mx > 32767 || r.rmin < 2 * np - 1 || (np32 != 0 && np32 != np)) return pp;
            np32 = np;
        }
    }
    const long min_blocks = runs.size() == 1 ? LRF_PERSIST_MIN_BLOCKS_ONE_FAMILY : LRF_PERSIST_MIN_BLOCKS;
    if (nblocks < (persist_env == 1 ? LRF_BCDW_MIN_BLOCKS : min_blocks)) return pp;
    pp.use = true;
    pp.f16 = f16 || np32 != 0; // (the instantiations with ranks 17..32 carry the 9..16 body too: their chroma planes)
    pp.np32 = np32;
    // the call's first iteration inside the launch too (k_bcd_p<.., 0, true>): ranks <= 16 only — the caller adds its own
    // conditions (the old U comes from the initialisation's W0, run_bcd)
    static const bool first_off = dev_flag("LRF_NO_PERSIST_FIRST");
    pp.first = np32 == 0 && !first_off;
    return pp;
}

template <bool F16, int NP32, bool FIRST = false>
static int bcdp_launch_t(lrf_ctx* c, int attr_bit, int wgs, int wave_lds, const float* X, const PlaneDesc* pl, const BlockDesc* bl, int nblocks,
                         int nplanes, int plane0, const BcdpTabs& t16, const BcdpTabs& t64, int8_t* U, int8_t* V, GsParams gp, int niter)
{
    if (!(c->attr_persist & (1u << attr_bit))) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_p<F16, NP32, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_WAVES * wave_lds));
        c->attr_persist |= 1u << attr_bit;
    }
    Prof p(c, LRF_K_BCD_PERSIST);
    hipLaunchKernelGGL((k_bcd_p<F16, NP32, FIRST>), dim3((unsigned)wgs), dim3(64 * LRF_BCDW_WAVES), (size_t)LRF_BCDW_WAVES * wave_lds, c->stream, X, pl, bl, t16,
                       t64, U, V, gp, nblocks, niter, nplanes, plane0, (BcdpSync*)c->psync.p, c->h_perr, 2 * nplanes, ++c->pseq, wave_lds);
    LAUNCH_CHECK();
    return LRF_OK;
}

int bcdp_launch(lrf_ctx* c, const PersistPlan& pp, const float* X, const PlaneDesc* pl, const BlockDesc* bl, int nblocks, int nplanes, int plane0,
                const FamBufs& f16, const FamBufs& f64, int8_t* U, int8_t* V, GsParams gp, int niter, bool first)
{
    if (first && (!pp.first || pp.np32 != 0)) return set_err(LRF_EINVAL, "internal: first iteration inside k_bcd_p with ranks above 16");
    if (!c->h_perr) {
        HIP_TRY(hipHostMalloc((void**)&c->h_perr, sizeof(int), hipHostMallocDefault));
        *c->h_perr = 0;
    }
    // a failure nobody has looked at yet (the results of the calls it names were never checked): refuse to go on silently
    int rc = ctx_check(c);
    if (rc) return rc;
    const size_t sbytes = sizeof(BcdpSync) + (2 * (size_t)nplanes + (size_t)nblocks) * sizeof(int); // (+ a debug count per block)
    const void* before = c->psync.p;
    if ((rc = ensure(c, c->psync, sbytes))) return rc;
    if (c->psync.p != before || c->psync_dirty) { // a launch leaves the state zeroed (its last wave); a new buffer or a failed launch does not
        HIP_TRY(hipMemsetAsync(c->psync.p, 0, c->psync.cap, c->stream));
        c->psync_dirty = false;
    }
    const long total_waves = (long)niter * nblocks;
    long wgs = (total_waves + LRF_BCDW_WAVES - 1) / LRF_BCDW_WAVES;
    if (wgs > 512) wgs = 512; // two workgroups per CU resident; later ones would only find the queue empty
    // LDS per wave: the largest share a family of the call needs
    int wave_lds = (64 * 64 + 64 * 8) * 4; // ranks <= 8 (LRF_BCDW_LDS / LRF_BCDW_WAVES)
    if (pp.f16 && LRF_BCDW16_WAVE_LDS > wave_lds) wave_lds = LRF_BCDW16_WAVE_LDS;
    const BcdpTabs t16{f16.vf, f16.bf, f16.pp, f16.qp, f16.wf}, t64{f64.vf, f64.bf, f64.pp, f64.qp, f64.wf};
    gp.exact_int = 1;
#define LRF_P(F16, NP, BIT)                                                                                                          \
    return bcdp_launch_t<F16, NP>(c, BIT, (int)wgs, NP ? (LRF_BCDW32_WAVE_LDS(NP) > wave_lds ? LRF_BCDW32_WAVE_LDS(NP) : wave_lds) : wave_lds, X, pl, bl, \
                                  nblocks, nplanes, plane0, t16, t64, U, V, gp, niter)
    if (first) { // (niter = K: item iteration 0 is the call's first iteration)
        if (!pp.f16) return bcdp_launch_t<false, 0, true>(c, 10, (int)wgs, wave_lds, X, pl, bl, nblocks, nplanes, plane0, t16, t64, U, V, gp, niter);
        return bcdp_launch_t<true, 0, true>(c, 11, (int)wgs, wave_lds, X, pl, bl, nblocks, nplanes, plane0, t16, t64, U, V, gp, niter);
    }
    if (!pp.f16) LRF_P(false, 0, 0);
    switch (pp.np32) {
    case 0: LRF_P(true, 0, 1);
    case 9: LRF_P(true, 9, 2);
    case 10: LRF_P(true, 10, 3);
    case 11: LRF_P(true, 11, 4);
    case 12: LRF_P(true, 12, 5);
    case 13: LRF_P(true, 13, 6);
    case 14: LRF_P(true, 14, 7);
    case 15: LRF_P(true, 15, 8);
    default: LRF_P(true, 16, 9);
    }
#undef LRF_P
}
