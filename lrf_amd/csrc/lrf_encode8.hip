// lrf_encode8.hip — the 64-column path of liblrf_hip.so (8x8 patches; the default branch of qmf_encode / qmf_decode): its
// kernels and their launch sequences — patch matrices, exact Gram matrix, SVD initialisation, the BCD iterations by rank
// family, decode — and the C ABI entry points built on them (include/lrf_hip.h).
#include "lrf_host.h"
#include "lrf_gram_kernels.hip"
#include "lrf_kernels.hip"
#include "lrf_bcdw_kernel.hip"
#include "lrf_bcdw16_kernel.hip"

// gram_exp: the fixed-point grid exponent of the exact Gram matrix (max|x| < 2^gram_exp) when the caller knows it — 8 for the
// planes of qmf_encode — or LRF_GRAM_EXP_FROM_DATA: one more pass over X finds it per matrix
static int run_init(lrf_ctx* c, const float* X, const Tables& t, const int8_t* sign_dev, int gram_exp)
{
    if (!(c->attr_done & (1u << 0))) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<8>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<16>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<32>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<64>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<16, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<16>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<32, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<32>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<64, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<64>)));
        c->attr_done |= 1u << 0;
    }
    int rmax = table_rmax(t), rp = table_rp(t), nplanes = (int)t.planes.size();
    {
        Prof p(c, LRF_K_GRAM);
        if (gram_exp == LRF_GRAM_EXP_FROM_DATA) {
            hipLaunchKernelGGL(k_gram_exponent, dim3(nplanes), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p, (int*)c->gexp.p);
            LAUNCH_CHECK();
        }
        // (t.ngram_rest: all chunks, or — behind k_planes16_gram, which has written the luma planes' partials itself — the chroma ones)
        if (t.ngram_rest != (int)t.gchunks.size() && gram_exp != LRF_PLANES_GRAM_EXP)
            return set_err(LRF_EINVAL, "internal: fused Gram partials exist only for the planes of qmf_encode");
        if (t.ngram_rest > 0) {
            if (gram_exp == LRF_PLANES_GRAM_EXP)
                hipLaunchKernelGGL(k_gram64<true>, dim3((unsigned)t.ngram_rest), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p,
                                   (const GramChunk*)c->gchunks.p, (const int*)c->gexp.p, gram_exp, (ulonglong2*)c->gpart.p);
            else
                hipLaunchKernelGGL(k_gram64<false>, dim3((unsigned)t.ngram_rest), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p,
                                   (const GramChunk*)c->gchunks.p, (const int*)c->gexp.p, gram_exp, (ulonglong2*)c->gpart.p);
            LAUNCH_CHECK();
        }
    }
    Prof p(c, LRF_K_INIT);
    const std::vector<FamRun> runs = plan_runs(t);
    const bool mixed = plan_is_mixed(runs);
    // A call whose iterations run in the persistent kernel keeps one stream — except here: the initialisation kernels of its
    // families are per-matrix latency chains that leave most of a CU idle (k_init<16> 190 us for 256 luma planes, k_init<8> 180
    // us for 512 chroma planes, one round of workgroups each), and LDS admits one workgroup of the first beside two of the
    // second: forked for this stage only, the later runs (chroma: more, smaller workgroups) enqueued first, joined at once.
    // (The order of the enqueues does not show in the step time — measured both ways at five rank triples — because the
    // streams' queues place their workgroups side by side either way.)  Since round 5's LDS layout (InitLds) a ZR = 32 workgroup
    // has two ZR = 16 ones beside it as well ((26,13,13): the stage 484 -> ~340 us).
    static const bool init_fork_off = dev_flag("LRF_NO_INIT_FORK");
    const bool init_only_fork = !c->fam_parallel && c->init_parallel && !init_fork_off;
    if (init_only_fork) c->fam_parallel = true;
    {
        int rcf = fam_fork_streams(c, runs.size(), init_only_fork);
        if (init_only_fork) c->fam_parallel = false;
        if (rcf) return rcf;
    }
    long ninit = 0;
    for (const FamRun& r : runs) ninit += r.nbase;
    const bool dense = ninit > 256; // more initialisation workgroups than CUs (k_init's DENSE)
    for (size_t rj = 0; rj < runs.size(); rj++) {
        const size_t ri = init_only_fork ? runs.size() - 1 - rj : rj;
        const FamRun& r = runs[ri];
        hipStream_t rs = run_stream(c, ri);
        const FamBufs fb = run_bufs(c, r, mixed);
        if (r.nbase == 0) continue; // (a sweep call: every plane of this run takes its columns from a plane of another run)
#define LRF_LAUNCH_INIT(ZR, ...)                                                                                     \
    hipLaunchKernelGGL((k_init<ZR, ##__VA_ARGS__>), dim3(r.nbase), dim3(ZR > 8 ? 512 : 256), sizeof(InitLds<ZR>), rs, (const ulonglong2*)c->gpart.p, \
                       (const int*)c->gexp.p, gram_exp, (const PlaneDesc*)c->planes.p, sign_dev, fb.vf, fb.wf, c->init_sweeps, r.pitch, r.plane0)
        if (r.rmax <= 8) LRF_LAUNCH_INIT(8);
        else if (dense) {
            if (r.rmax <= 16) LRF_LAUNCH_INIT(16);
            else if (r.rmax <= 32) LRF_LAUNCH_INIT(32);
            else LRF_LAUNCH_INIT(64);
        } else { // (no more matrices than CUs: the eight-wave workgroups without spills, k_init<.., 8, false>)
            if (r.rmax <= 16) LRF_LAUNCH_INIT(16, 8, false);
            else if (r.rmax <= 32) LRF_LAUNCH_INIT(32, 8, false);
            else LRF_LAUNCH_INIT(64, 8, false);
        }
#undef LRF_LAUNCH_INIT
        LAUNCH_CHECK();
    }
    (void)rmax;
    if (init_only_fork) {
        int rcj = fam_join_streams(c, runs.size());
        if (rcj) return rcj;
    }
    bool shares = false;
    for (const FamRun& r : runs) shares = shares || r.nbase != r.nplanes;
    if (shares) { // a sweep call (its entry point keeps one stream): the other ranks' planes take their columns from the base planes
        if (c->fam_forked) return set_err(LRF_EINVAL, "internal: a call that shares initialisations between planes must not fork its families");
        hipLaunchKernelGGL(k_init_share, dim3(nplanes), dim3(256), 0, c->stream, (const PlaneDesc*)c->planes.p, (float*)c->vf.p, (float*)c->wf.p,
                           (float*)c->vf16.p, (float*)c->wf16.p, plan_splits((long)t.blocks.size(), table_rmax(t)) ? 1 : 0, mixed ? 1 : 0, rp);
        LAUNCH_CHECK();
    }
    return LRF_OK;
}

static GsParams make_gs(int lo, int hi)
{
    GsParams gp;
    gp.lo = (float)lo;
    gp.hi = (float)hi;
    int mx = abs(lo) > abs(hi) ? abs(lo) : abs(hi);
    gp.flimit = (float)(mx + 2);
    gp.fthr = 0.5f - 8e-7f * (float)(mx + 2); // see gs_row: q~ is within 3 ulp (< 2e-7 |q|) of fl(num/den)
    gp.exact_int = 0; // set per call by run_bcd (depends on the largest rank)
    return gp;
}

// mode: 1 = old U from X @ W0 (after run_init), 2 = old U from caller's fp32 U0
static int run_bcd(lrf_ctx* c, const float* X, const Tables& t, int K, int lo, int hi, int first_mode, const float* U0,
                   int8_t* U, int8_t* V)
{
    const PlaneDesc* pl = (const PlaneDesc*)c->planes.p;
    const BlockDesc* bl = (const BlockDesc*)c->blocks.p;
    GsParams gp = make_gs(lo, hi);
    if (table_rmax(t) > LRF_BIG_TO_ANY_RANK) return set_err(LRF_ENOTSUP, "internal: ranks above %d iterate on the any-shape kernels", LRF_BIG_TO_ANY_RANK);
    const std::vector<FamRun> runs = plan_runs(t);
    const bool mixed = plan_is_mixed(runs);
    // k_bcd_w (one wave per block, no barriers) for rank <= 8 runs — of LRF_BCDW_MIN_BLOCKS blocks or more: with fewer than a
    // wave per SIMD what counts is the latency of ONE block, and there the four waves of the workgroup kernel k_bcd share a
    // block's sub-tile (one 512x768 image: 27.9 -> 17.0 us per launch, 8 images 28.5 -> 18.2, 32 images 31.7 -> 27.6; equal at 48)
    const bool wave_variant = bcd_wave_variant();
    if (!(c->attr_done & (1u << 1))) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW16_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW16_LDS));
        c->attr_done |= 1u << 1;
    }
    // the b tables of the initial V
    for (size_t ri = 0; ri < runs.size(); ri++) {
        const FamRun& r = runs[ri];
        hipStream_t rs = run_stream(c, ri);
        const FamBufs fb = run_bufs(c, r, mixed);
        if (r.pitch == 16) {
            hipLaunchKernelGGL(k_bprep, dim3(r.nplanes), dim3(256), 0, rs, pl, (const float*)fb.vf, fb.bf, r.plane0);
            LAUNCH_CHECK();
        } else {
            int rcb = bcd32_bprep(rs, pl, fb.vf, fb.bf, r.nplanes, r.plane0);
            if (rcb) return rcb;
        }
    }
    // Iterations >= 2 with bounds where every term and partial sum of `uu @ bb` is an exact integer in fp32 for the largest
    // rank of a run ((R - 1) 64 mx^3 < 2^24): the order of that sum is immaterial, which lets ranks 9..16 (gs_row_lds) and
    // 17..32 (k_bcd_w32, k_bcd_mid) replace the reference's dependent chain by independent fmas, bit for bit
    const long mx_b = abs(lo) > abs(hi) ? abs(lo) : abs(hi);
    static const bool exact_off = dev_flag("LRF_GENERIC_GS");
    static const long w16_min = env_long("LRF_BCDW16_MIN_BLOCKS", LRF_BCDW16_MIN_BLOCKS); // test hook (lrf_env.h)
    // Iterations 2..K of a large call in ONE launch (k_bcd_p<F16, NP32>, lrf_bcd_persist.hip): the U updates of all iterations
    // and planes pulled from a queue, each matrix's V update done by the last of its blocks to finish (bcdp_plan says which
    // calls).  Such a call never forks its families onto streams (plan_fam_parallel).
    const PersistPlan persist = bcdp_plan(c, runs, K, lo, hi);
    // ... and the first iteration as well when its old U is X @ W0 of the initialisation just run (first_mode 1) and no plane of
    // ranks 9..16 is small enough for ATen's native order (bcdp_plan has checked that already): then the b tables above are
    // the last launch before k_bcd_p.
    const bool persist_first = persist.use && persist.first && first_mode == 1;
    for (int it = 0; it < K; it++) {
        if (persist.use && it == (persist_first ? 0 : 1)) {
            const FamRun& r0 = runs.front();
            long nb = 0, np = 0;
            for (const FamRun& r : runs) { nb += r.nblocks; np += r.nplanes; }
            // the table sets of the two rank pitches (run_bufs): a call without ranks above 16 has only the pitch-16 one
            FamBufs f16{nullptr, nullptr, nullptr, nullptr, nullptr}, f64 = f16;
            for (const FamRun& r : runs) (r.pitch == 16 ? f16 : f64) = run_bufs(c, r, mixed);
            int rcp = bcdp_launch(c, persist, X, pl, bl + r0.block0, (int)nb, (int)np, r0.plane0, f16, f64, U, V, gp, persist_first ? K : K - 1, persist_first);
            if (rcp) return rcp;
            break;
        }
        {
            Prof p(c, LRF_K_BCD);
            const int mode = (it == 0) ? first_mode : 0;
            for (size_t ri = 0; ri < runs.size(); ri++) {
                const FamRun& r = runs[ri];
                hipStream_t rs = run_stream(c, ri);
                const FamBufs fb = run_bufs(c, r, mixed);
                const BlockDesc* blr = bl + r.block0;
                const int nbr = r.nblocks;
                GsParams gpr = gp;
                gpr.exact_int = (!exact_off && (long)(r.rmax - 1) * 64 * mx_b * mx_b * mx_b < (1L << 24)) ? 1 : 0;
#define LRF_LAUNCH_W(MODE)                                                                                           \
    hipLaunchKernelGGL((k_bcd_w<MODE>), dim3((nbr + LRF_BCDW_WAVES - 1) / LRF_BCDW_WAVES), dim3(64 * LRF_BCDW_WAVES), LRF_BCDW_LDS, rs, X, pl, blr, \
                       (const float*)fb.vf, (const float*)fb.wf, (const float*)fb.bf, U0, U, fb.pp, fb.qp, gpr, nbr)
#define LRF_LAUNCH_WG(MODE, RMAX)                                                                                    \
    hipLaunchKernelGGL((k_bcd<MODE, RMAX>), dim3(nbr), dim3(256), 0, rs, X, pl, blr, (const float*)fb.vf, (const float*)fb.wf, \
                       (const float*)fb.bf, U0, U, fb.pp, fb.qp, gpr)
                if (r.fam == 2) {
                    // ranks 17..32: k_bcd_w32 / k_bcd_w32f (one wave per block) or the workgroup kernel k_bcd_mid (lrf_bcd32.hip)
                    const BcdLaunch a{X, pl, blr, nbr, fb.vf, fb.wf, fb.bf, U0, U, fb.pp, fb.qp, gpr, mode};
                    int rcu = bcd32_update_u(c, rs, a, r, mx_b);
                    if (rcu) return rcu;
                    continue;
                } else if (r.fam == 0 && wave_variant && nbr >= LRF_BCDW_MIN_BLOCKS) {
                    if (mode == 1) LRF_LAUNCH_W(1);
                    else if (mode == 2) LRF_LAUNCH_W(2);
                    else LRF_LAUNCH_W(0);
                } else if (r.fam == 0) {
                    if (mode == 1) LRF_LAUNCH_WG(1, 8);
                    else if (mode == 2) LRF_LAUNCH_WG(2, 8);
                    else LRF_LAUNCH_WG(0, 8);
                } else if (wave_variant && nbr >= w16_min && ((mode == 0 && gpr.exact_int) || (mode == 1 && !r.any_native))) {
                    // ranks 9..16 (and the lower-rank planes of such a run): iterations >= 2 with exact-integer bounds, and
                    // the first iteration from the initialisation's W0 unless a plane is small enough for ATen's native order
#define LRF_LAUNCH_W16(MODE)                                                                                         \
    hipLaunchKernelGGL((k_bcd_w16<MODE>), dim3((nbr + LRF_BCDW16_WAVES - 1) / LRF_BCDW16_WAVES), dim3(64 * LRF_BCDW16_WAVES), LRF_BCDW16_LDS, \
                       rs, X, pl, blr, (const float*)fb.vf, (const float*)fb.wf, (const float*)fb.bf, U, fb.pp, fb.qp, gpr, nbr)
                    if (mode == 1) LRF_LAUNCH_W16(1);
                    else LRF_LAUNCH_W16(0);
#undef LRF_LAUNCH_W16
                } else {
                    if (mode == 1) LRF_LAUNCH_WG(1, 16);
                    else if (mode == 2) LRF_LAUNCH_WG(2, 16);
                    else LRF_LAUNCH_WG(0, 16);
                }
#undef LRF_LAUNCH_W
#undef LRF_LAUNCH_WG
                LAUNCH_CHECK();
            }
        }
        {
            Prof p(c, LRF_K_VUPDATE);
            const int last = it == K - 1 ? 1 : 0;
            for (size_t ri = 0; ri < runs.size(); ri++) {
                const FamRun& r = runs[ri];
                hipStream_t rs = run_stream(c, ri);
                const FamBufs fb = run_bufs(c, r, mixed);
                if (r.fam == 2) {
                    int rcv = bcd32_update_v(c, rs, pl, fb.pp, fb.qp, fb.vf, fb.bf, V, gp.lo, gp.hi, last, r.nplanes, r.plane0);
                    if (rcv) return rcv;
                    continue;
                } else if (r.fam == 0)
                    hipLaunchKernelGGL(k_vupdate<8>, dim3(r.nplanes), dim3(256), 0, rs, pl, (const float*)fb.pp, (const float*)fb.qp,
                                       fb.vf, fb.bf, V, gp, last, r.plane0);
                else
                    hipLaunchKernelGGL(k_vupdate<16>, dim3(r.nplanes), dim3(256), 0, rs, pl, (const float*)fb.pp, (const float*)fb.qp,
                                       fb.vf, fb.bf, V, gp, last, r.plane0);
                LAUNCH_CHECK();
            }
        }
    }
    return fam_join_streams(c, runs.size());
}

// Whether the kernel families of a call may run on streams of their own (run_init forks, run_bcd joins): not when the call's
// iterations 2..K run in the persistent kernel — one launch for all families, behind a first iteration whose family kernels
// run one after the other (side by side they were SLOWER: k_bcd_w32f 305 us and k_bcd_w16<1> 80 us alone, 590 us together).
static bool plan_fam_parallel(lrf_ctx* c, const Tables& t, int K, int lo, int hi) { return !bcdp_plan(c, plan_runs(t), K, lo, hi).use; }

// ---- C ABI ------------------------------------------------------------------------------------
extern "C" {

int lrf_qmf_planes_from_rgb_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, float* X)
{
    if (!c || !rgb || !X) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    ImageGeom g;
    int rc = make_geom(H, W, &g);
    if (rc) return rc;
    LRF_ON_DEVICE(c);
    Prof p(c, LRF_K_PLANES);
    if ((long)H * W * 3 >= (1L << 31)) return set_err(LRF_ENOTSUP, "image too large for 32-bit pixel indexing");
    static const bool no_tiled = dev_flag("LRF_PLANES_NO_TILED");
    if (H % 16 == 0 && W % 16 == 0 && (reinterpret_cast<uintptr_t>(rgb) & 7) == 0 && !no_tiled)
        hipLaunchKernelGGL(k_planes16, dim3((unsigned)((H / 16) * ((g.p[0].nw + 31) / 32)), (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H,
                           (int)W, g, X);
    else if (!no_tiled) {
        // any other size: the same tiling over the padded planes (k_planes_strip); blocks dealt so that the strips of an image
        // stay on one XCD (block index mod 8 is the XCD when the grid's x extent is a multiple of 8)
        static const bool no_xcd = dev_flag("LRF_PLANES_NO_XCD");
        const int ncols = g.p[0].nw > 2 * g.p[1].nw ? g.p[0].nw : 2 * g.p[1].nw;
        const int per_strip = (ncols + 31) / 32;
        const int nstrips = (g.p[0].nh + 1) / 2 > g.p[1].nh ? (g.p[0].nh + 1) / 2 : g.p[1].nh;
        const int nblk = nstrips * per_strip;
        const int chunk = no_xcd ? 0 : (nblk + 7) / 8;
        const dim3 grid((unsigned)(chunk ? 8 * chunk : nblk), (unsigned)B);
#define LRF_LAUNCH_STRIP(KH, KW) \
    hipLaunchKernelGGL((k_planes_strip<KH, KW>), grid, dim3(256), 0, c->stream, rgb, (int)H, (int)W, g, X, per_strip, nblk, chunk)
        if (H & 1) {
            if (W & 1) LRF_LAUNCH_STRIP(3, 3);
            else LRF_LAUNCH_STRIP(3, 2);
        } else {
            if (W & 1) LRF_LAUNCH_STRIP(2, 3);
            else LRF_LAUNCH_STRIP(2, 2);
        }
#undef LRF_LAUNCH_STRIP
    } else
        hipLaunchKernelGGL(k_planes, dim3((unsigned)(g.p[1].pr0 + g.p[1].nh), (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H,
                           (int)W, g, X);
    LAUNCH_CHECK();
    return LRF_OK;
}

static void uniform_tables(Tables& t, int64_t B, int64_t M, int R, bool with_sign)
{
    for (int64_t b = 0; b < B; b++)
        add_plane(t, b * M * 64, b * M * R, b * 64 * R, b * M * R, b * 64 * R, (int)M, R, with_sign ? (int)(b * R) : -1);
}

// Ranks 33..64 of the 64-column path iterate on the any-shape kernels, which spread the ordered Gauss-Seidel chain over all
// waves (a first rank-64 workgroup kernel with the chain on one wave of four was 2x slower there: (40,20) 20.8 against 11.4 ms
// per 64 images, (64,32) 45.6 against 20.6).  The initialisation stays with k_init, which mirrors the oracle operation for
// operation: k_emit_init writes its factors out as fp32 and the any-shape iteration takes over.

// one class of B equal-shaped 64-column matrices whose initial factors sit contiguously at U0c / V0c
static int any_bcd_from_init(lrf_ctx* c, const float* X, long x_batch, int B, int M, int R, int K, int lo, int hi, const float* U0c,
                             const float* V0c, int8_t* U, long u_batch, int8_t* V, long v_batch)
{
    int rc = any_workspace(c, B, M, 64, R);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->any_uf.p, U0c, (size_t)B * M * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->any_vf.p, V0c, (size_t)B * 64 * R * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return any_run_bcd_ex(c, X, x_batch, B, M, 64, R, K, lo, hi, U, u_batch, V, v_batch);
}

// k_init on the uploaded table, then its factors as fp32 into c->any_e2 (U0 at [0], V0 behind it): offsets from the table
static int init_to_fp32(lrf_ctx* c, const float* X, const Tables& t, const int8_t* sign, size_t u0_floats, size_t v0_floats, float** U0,
                        float** V0, int gram_exp)
{
    int rc = run_init(c, X, t, sign, gram_exp);
    if (rc) return rc;
    if ((rc = ensure(c, c->any_e2, (u0_floats + v0_floats) * sizeof(float)))) return rc;
    *U0 = (float*)c->any_e2.p;
    *V0 = *U0 + u0_floats;
    hipLaunchKernelGGL(k_emit_init, dim3((unsigned)t.blocks.size()), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p,
                       (const BlockDesc*)c->blocks.p, (const float*)c->vf.p, (const float*)c->wf.p, *U0, *V0, table_rp(t));
    LAUNCH_CHECK();
    return LRF_OK;
}

int lrf_qmf_decompose_f32(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, int K, int lo, int hi,
                          const int8_t* sign, int8_t* U, int8_t* V)
{
    if (!c || !X || !U || !V) return set_err(LRF_EINVAL, "NULL argument");
    static const bool force_any = dev_flag("LRF_FORCE_ANY");
    if (N != LRF_PATCH_ELEMS || R > LRF_MAX_RANK || force_any) return any_decompose(c, X, B, M, N, R, K, lo, hi, sign, U, V);
    int rc = check_params(M, N, R, K, lo, hi);
    if (rc) return rc;
    if (B < 1) return set_err(LRF_EINVAL, "B must be >= 1");
    LRF_ON_DEVICE(c);
    Tables t;
    uniform_tables(t, B, M, R, sign != nullptr);
    if ((rc = upload_tables(c, t))) return rc;
    if (R > LRF_BIG_TO_ANY_RANK) {
        float *U0, *V0;
        if ((rc = init_to_fp32(c, X, t, sign, (size_t)B * M * R, (size_t)B * 64 * R, &U0, &V0, LRF_GRAM_EXP_FROM_DATA))) return rc;
        return any_bcd_from_init(c, X, M * 64, (int)B, (int)M, R, K, lo, hi, U0, V0, U, M * R, V, 64L * R);
    }
    c->fam_parallel = plan_fam_parallel(c, t, K, lo, hi); // run_init is followed by run_bcd at once: the kernel families of the call may run side by side
    c->init_parallel = !c->fam_parallel;
    rc = run_init(c, X, t, sign, LRF_GRAM_EXP_FROM_DATA);
    c->fam_parallel = c->init_parallel = false;
    if (rc) {
        (void)fam_join_streams(c, 3);
        return rc;
    }
    return run_bcd(c, X, t, K, lo, hi, 1, nullptr, U, V);
}

int lrf_qmf_bcd_f32(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, int K, int lo, int hi,
                    const float* U0, const float* V0, int8_t* U, int8_t* V)
{
    if (!c || !X || !U || !V || !U0 || !V0) return set_err(LRF_EINVAL, "NULL argument");
    if (N != LRF_PATCH_ELEMS || R > LRF_BIG_TO_ANY_RANK) return any_bcd(c, X, B, M, N, R, K, lo, hi, U0, V0, U, V);
    int rc = check_params(M, N, R, K, lo, hi);
    if (rc) return rc;
    if (B < 1) return set_err(LRF_EINVAL, "B must be >= 1");
    LRF_ON_DEVICE(c);
    Tables t;
    uniform_tables(t, B, M, R, false);
    if ((rc = upload_tables(c, t))) return rc;
    hipLaunchKernelGGL(k_load_v0, dim3((unsigned)t.planes.size()), dim3(256), 0, c->stream, (const PlaneDesc*)c->planes.p, V0,
                       (float*)c->vf.p, table_rp(t));
    LAUNCH_CHECK();
    return run_bcd(c, X, t, K, lo, hi, 2, U0, U, V);
}

int lrf_qmf_svd_init_f32(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, const int8_t* sign,
                         float* U0, float* V0)
{
    if (!c || !X || !U0 || !V0) return set_err(LRF_EINVAL, "NULL argument");
    if (N != LRF_PATCH_ELEMS || R > LRF_MAX_RANK) return any_svd_init(c, X, B, M, N, R, sign, U0, V0);
    int rc = check_params(M, N, R, 1, -16, 15);
    if (rc) return rc;
    if (B < 1) return set_err(LRF_EINVAL, "B must be >= 1");
    LRF_ON_DEVICE(c);
    Tables t;
    uniform_tables(t, B, M, R, sign != nullptr);
    if ((rc = upload_tables(c, t))) return rc;
    if ((rc = run_init(c, X, t, sign, LRF_GRAM_EXP_FROM_DATA))) return rc;
    hipLaunchKernelGGL(k_emit_init, dim3((unsigned)t.blocks.size()), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p,
                       (const BlockDesc*)c->blocks.p, (const float*)c->vf.p, (const float*)c->wf.p, U0, V0, table_rp(t));
    LAUNCH_CHECK();
    return LRF_OK;
}

} // extern "C"

int encode_rgb_prepare(lrf_ctx* c, int64_t B, int64_t H, int64_t W, const int R[3], int K, int lo, int hi, bool with_sign,
                       bool fuse_gram /* planes_gram_eligible(rgb, H, W): the luma planes' Gram partials come from k_planes16_gram */, EncodePlan& ep)
{
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    int rc = make_geom(H, W, &ep.g);
    if (rc) return rc;
    const ImageGeom& g = ep.g;
    for (int ch = 0; ch < 3; ch++)
        if ((rc = check_params(g.p[ch].M, 64, R[ch], K, lo, hi))) return rc;
    if ((rc = ensure(c, c->x, (size_t)B * g.img_floats * sizeof(float)))) return rc;
    // plane table: all Y planes first (four times the work of a chroma plane), then Cb, then Cr
    const long s_img = R[0] + R[1] + R[2];
    const long soff[3] = {0, R[0], R[0] + R[1]};
    for (int ch = 0; ch < 3; ch++) {
        ep.uoff[ch] = ep.u_img; ep.voff[ch] = ep.v_img;
        ep.u_img += (long)g.p[ch].M * R[ch];
        ep.v_img += 64L * R[ch];
    }
    // fp32 initial factors, if they are wanted (ranks above LRF_BIG_TO_ANY_RANK): per plane class contiguous [B][M][R] / [B][64][R]
    for (int ch = 0; ch < 3; ch++) {
        ep.u0c[ch + 1] = ep.u0c[ch] + B * (long)g.p[ch].M * R[ch];
        ep.v0c[ch + 1] = ep.v0c[ch] + B * 64L * R[ch];
    }
    for (int ch = 0; ch < 3; ch++)
        for (int64_t b = 0; b < B; b++) {
            add_plane(ep.t, b * g.img_floats + g.p[ch].xoff, b * ep.u_img + ep.uoff[ch], b * ep.v_img + ep.voff[ch],
                      ep.u0c[ch] + b * (long)g.p[ch].M * R[ch], ep.v0c[ch] + b * 64L * R[ch], g.p[ch].M, R[ch],
                      with_sign ? (int)(b * s_img + soff[ch]) : -1);
            if (fuse_gram && ch == 0) ep.t.planes.back().gram_fused = 1;
        }
    return upload_tables(c, ep.t);
}

extern "C" {

int lrf_qmf_encode_rgb_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, const int R[3], int K, int lo,
                          int hi, const int8_t* sign, int8_t* U, int8_t* V)
{
    if (!c || !rgb || !R || !U || !V) return set_err(LRF_EINVAL, "NULL argument");
    LRF_ON_DEVICE(c);
    EncodePlan ep;
    const bool fuse = planes_gram_eligible(rgb, B, H, W);
    int rc = encode_rgb_prepare(c, B, H, W, R, K, lo, hi, sign != nullptr, fuse, ep);
    if (rc) return rc;
    const ImageGeom& g = ep.g;
    Tables& t = ep.t;
    const long u_img = ep.u_img, v_img = ep.v_img;
    const long *uoff = ep.uoff, *voff = ep.voff, *u0c = ep.u0c, *v0c = ep.v0c;
    float* X = (float*)c->x.p;
    if ((rc = fuse ? planes_gram_from_rgb(c, rgb, H, W, g, t, X) : lrf_qmf_planes_from_rgb_u8(c, rgb, B, H, W, X))) return rc;
    if (c->planes_done) HIP_TRY(hipEventRecord(c->planes_done, c->stream)); // the RGB bytes are not read again
    if (table_rmax(t) > LRF_BIG_TO_ANY_RANK) {
        float *U0, *V0;
        if ((rc = init_to_fp32(c, X, t, sign, (size_t)u0c[3], (size_t)v0c[3], &U0, &V0, LRF_PLANES_GRAM_EXP))) return rc;
        for (int ch = 0; ch < 3; ch++)
            if ((rc = any_bcd_from_init(c, X + g.p[ch].xoff, g.img_floats, (int)B, g.p[ch].M, R[ch], K, lo, hi, U0 + u0c[ch],
                                        V0 + v0c[ch], U + uoff[ch], u_img, V + voff[ch], v_img)))
                return rc;
        return LRF_OK;
    }
    c->fam_parallel = plan_fam_parallel(c, t, K, lo, hi); // run_init is followed by run_bcd at once: the kernel families of the call may run side by side
    c->init_parallel = !c->fam_parallel;
    rc = run_init(c, X, t, sign, LRF_PLANES_GRAM_EXP);
    c->fam_parallel = c->init_parallel = false;
    if (rc) {
        (void)fam_join_streams(c, 3);
        return rc;
    }
    return run_bcd(c, X, t, K, lo, hi, 1, nullptr, U, V);
}

// One batch at Q rank triples in one call (BASELINE config 3: an R-D sweep of 24 images x qualities 1..32; the reference's loop
// experiments/comparison/eval.py:83-110 calls qmf_encode once per image and quality).  What does not depend on the rank is done
// once per image — patch matrices, exact Gram matrices — and the SVD initialisation once per (image, channel) at the largest
// rank asked for that channel: the lower ranks take its leading columns (k_init_share).  The BCD of ALL (quality, image) pairs
// runs as one call of Q x B "virtual images" that share X: large per-family launches, the persistent kernel from 3584 blocks.
// Output: for q = 0..Q-1 the factors of the B images at triple q back to back, each in lrf_qmf_encode_rgb_u8's layout
// (U: offset sum_{q' < q} B u_img(q'), image stride u_img(q) = sum_c M_c R[q][c]; V likewise with 64 R[q][c]).
// sign: optional [B][Rmax_Y + Rmax_Cb + Rmax_Cr] int8 (component signs of the largest ranks; every triple uses its leading ones).
int lrf_qmf_encode_sweep_rgb_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, int Q, const int* R /*[Q][3]*/, int K, int lo,
                                int hi, const int8_t* sign, int8_t* U, int8_t* V)
{
    if (!c || !rgb || !R || !U || !V) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    if (Q < 1 || Q > 4096) return set_err(LRF_EINVAL, "Q=%d out of range [1,4096]", Q);
    LRF_ON_DEVICE(c);
    ImageGeom g;
    int rc = make_geom(H, W, &g);
    if (rc) return rc;
    int rmaxc[3] = {0, 0, 0}, qbase[3] = {0, 0, 0}, rmax_t = 0;
    for (int q = 0; q < Q; q++)
        for (int ch = 0; ch < 3; ch++) {
            const int r = R[3 * q + ch];
            if ((rc = check_params(g.p[ch].M, 64, r, K, lo, hi))) return rc;
            if (r > LRF_BIG_TO_ANY_RANK) return set_err(LRF_ENOTSUP, "sweep call: rank %d > %d (ranks above it iterate on the any-shape kernels: one call per triple)", r, LRF_BIG_TO_ANY_RANK);
            if (r > rmaxc[ch]) { rmaxc[ch] = r; qbase[ch] = q; }
            rmax_t = r > rmax_t ? r : rmax_t;
        }
    if ((rc = ensure(c, c->x, (size_t)B * g.img_floats * sizeof(float)))) return rc;
    // output offsets of the triples
    std::vector<long> uq((size_t)Q + 1, 0), vq((size_t)Q + 1, 0), u_img((size_t)Q), v_img((size_t)Q);
    for (int q = 0; q < Q; q++) {
        u_img[q] = v_img[q] = 0;
        for (int ch = 0; ch < 3; ch++) { u_img[q] += (long)g.p[ch].M * R[3 * q + ch]; v_img[q] += 64L * R[3 * q + ch]; }
        uq[q + 1] = uq[q] + B * u_img[q];
        vq[q + 1] = vq[q] + B * v_img[q];
    }
    // The plane table: Q x B x 3 planes on B x 3 matrices.  Order: by kernel family (so that plan_runs finds at most three runs),
    // inside a family the planes that compute an initialisation first (run_init launches k_init for a run's leading planes), then
    // luma before chroma; a call too small to split its families keeps one run: all initialising planes first.
    long nblk_img = 0;
    for (int ch = 0; ch < 3; ch++) nblk_img += (g.p[ch].M + LRF_KC - 1) / LRF_KC;
    const bool split = plan_splits((long)Q * B * nblk_img, rmax_t);
    const long s_img = rmaxc[0] + rmaxc[1] + rmaxc[2];
    const long soff[3] = {0, rmaxc[0], (long)rmaxc[0] + rmaxc[1]};
    struct Spec { int q, ch; long b; };
    std::vector<Spec> order;
    order.reserve((size_t)Q * B * 3);
    for (int fam = 0; fam < (split ? 3 : 1); fam++)
        for (int base = 1; base >= 0; base--)
            for (int ch = 0; ch < 3; ch++)
                for (int q = 0; q < Q; q++) {
                    if (split && fam_of_rank(R[3 * q + ch]) != fam) continue;
                    if ((q == qbase[ch]) != (base == 1)) continue;
                    for (long b = 0; b < B; b++) order.push_back(Spec{q, ch, b});
                }
    Tables t;
    std::vector<int> base_index((size_t)B * 3, -1);
    for (const Spec& sp : order) {
        const int r = R[3 * sp.q + sp.ch];
        long uo = uq[sp.q] + sp.b * u_img[sp.q], vo = vq[sp.q] + sp.b * v_img[sp.q];
        for (int c2 = 0; c2 < sp.ch; c2++) { uo += (long)g.p[c2].M * R[3 * sp.q + c2]; vo += 64L * R[3 * sp.q + c2]; }
        if (sp.q == qbase[sp.ch]) base_index[(size_t)sp.b * 3 + sp.ch] = (int)t.planes.size();
        add_plane(t, sp.b * g.img_floats + g.p[sp.ch].xoff, uo, vo, 0, 0, g.p[sp.ch].M, r, sign ? (int)(sp.b * s_img + soff[sp.ch]) : -1);
    }
    const bool fuse = planes_gram_eligible(rgb, B, H, W);
    for (size_t pi = 0; pi < t.planes.size(); pi++) {
        const Spec& sp = order[pi];
        t.planes[pi].init_src = base_index[(size_t)sp.b * 3 + sp.ch];
        if (fuse && sp.ch == 0 && t.planes[pi].init_src == (int)pi) t.planes[pi].gram_fused = 1; // one luma plane per image
    }
    if ((rc = upload_tables(c, t))) return rc;
    float* X = (float*)c->x.p;
    if ((rc = fuse ? planes_gram_from_rgb(c, rgb, H, W, g, t, X) : lrf_qmf_planes_from_rgb_u8(c, rgb, B, H, W, X))) return rc;
    if (c->planes_done) HIP_TRY(hipEventRecord(c->planes_done, c->stream));
    c->fam_parallel = false; // one stream: the shared initialisations tie the families together
    c->init_parallel = true;
    rc = run_init(c, X, t, sign, LRF_PLANES_GRAM_EXP);
    c->init_parallel = false;
    if (rc) {
        (void)fam_join_streams(c, 3);
        return rc;
    }
    return run_bcd(c, X, t, K, lo, hi, 1, nullptr, U, V);
}

int lrf_qmf_decode_rgb_u8(lrf_ctx* c, const int8_t* U, const int8_t* V, int64_t B, int64_t H, int64_t W, const int R[3],
                          uint8_t* rgb)
{
    if (!c || !U || !V || !R || !rgb) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1 || B > 65535) return set_err(LRF_EINVAL, "B=%ld out of range [1,65535]", (long)B);
    ImageGeom g;
    int rc = make_geom(H, W, &g);
    if (rc) return rc;
    for (int ch = 0; ch < 3; ch++)
        if (R[ch] < 1 || R[ch] > 64) return set_err(LRF_EINVAL, "rank %d out of range", R[ch]);
    LRF_ON_DEVICE(c);
    long u_img = 0, v_img = 0;
    for (int ch = 0; ch < 3; ch++) {
        u_img += (long)g.p[ch].M * R[ch];
        v_img += 64L * R[ch];
    }
    long n4 = (long)H * ((W + 3) / 4);
    Prof p(c, LRF_K_DECODE);
    static const bool no_tiled = dev_flag("LRF_DECODE_NO_TILED");
    // the tiled kernels (k_decode16: sides multiples of 16; k_decode_strip: any height, four-aligned chroma columns) are
    // instantiated for the rank bounds (chroma, luma) = (4,8) (8,8) (8,16) (16,16) (16,32): the reference's quality sweep up to 40
    const int rcm = R[1] > R[2] ? R[1] : R[2];
    const int RCb = rcm <= 4 ? 4 : (rcm <= 8 ? 8 : 16), RLb = R[0] <= 8 ? 8 : (R[0] <= 16 ? 16 : 32);
    const bool tiled_ranks = R[0] <= 32 && rcm <= 16 && !no_tiled;
    const bool sides16 = H % 16 == 0 && W % 16 == 0 && (reinterpret_cast<uintptr_t>(rgb) & 7) == 0;
    const bool strip_ok = W % 2 == 0 && g.p[0].left_crop % 2 == 0 && (g.p[1].left_crop - g.p[0].left_crop / 2) % 4 == 0 && g.p[1].w == W / 2;
    if (tiled_ranks && (sides16 || strip_ok)) {
        const int per_strip = (g.p[0].nw + 31) / 32;
        const dim3 grid16((unsigned)((H / 16) * per_strip), (unsigned)B), grids((unsigned)(((g.p[0].nh + 1) / 2) * per_strip), (unsigned)B);
#define LRF_DECODE_TILED(RC, RL)                                                                                                  \
    do {                                                                                                                         \
        if (sides16)                                                                                                             \
            hipLaunchKernelGGL((k_decode16<RC, RL>), grid16, dim3(256), 0, c->stream, U, V, (int)H, (int)W, g, R[0], R[1], R[2], u_img, v_img, rgb); \
        else                                                                                                                     \
            hipLaunchKernelGGL((k_decode_strip<RC, RL>), grids, dim3(256), 0, c->stream, U, V, (int)H, (int)W, g, R[0], R[1], R[2], u_img, v_img, rgb, per_strip); \
    } while (0)
        if (RLb == 8 && RCb == 4) LRF_DECODE_TILED(4, 8);
        else if (RLb == 8 && RCb == 8) LRF_DECODE_TILED(8, 8);
        else if (RLb == 16 && RCb <= 8) LRF_DECODE_TILED(8, 16);
        else if (RLb <= 16) LRF_DECODE_TILED(16, 16);
        else LRF_DECODE_TILED(16, 32);
#undef LRF_DECODE_TILED
    }
    else if (R[0] <= 8 && R[1] <= 8 && R[2] <= 8)
{
        // groups of four pixels per thread: as many as leave the call ~2048 workgroups (small calls keep one group per thread)
        long reps = (long)B * ((n4 + 255) / 256) / 2048;
        reps = reps < 1 ? 1 : (reps > 16 ? 16 : reps);
        hipLaunchKernelGGL(k_decode8, dim3((unsigned)((n4 + 256 * reps - 1) / (256 * reps)), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W,
                           g, R[0], R[1], R[2], u_img, v_img, rgb, (int)reps);
    }
    else
        hipLaunchKernelGGL(k_decode, dim3((unsigned)((n4 + 255) / 256), (unsigned)B), dim3(256), 0, c->stream, U, V, (int)H, (int)W,
                           g, R[0], R[1], R[2], u_img, v_img, rgb);
    LAUNCH_CHECK();
    return LRF_OK;
}


#if defined(LRF_STAMPS) || defined(LRF_INIT_STAMPS) || defined(LRF_BLK_STAMPS) || defined(LRF_REG_STAMPS)
int lrf_debug_read_stamps(lrf_ctx* c, unsigned long long* out_host, int n)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)n));
    return LRF_OK;
}
#endif

#ifdef LRF_GRAM_STAMPS
int lrf_debug_read_gram_stamps(lrf_ctx* c, unsigned long long* out_host, int n)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_gram_stamps), sizeof(unsigned long long) * (size_t)n));
    return LRF_OK;
}
#endif

} // extern "C"
