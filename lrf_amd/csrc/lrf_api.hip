// lrf_api.hip — host side of liblrf_hip.so: context, workspace, launch sequencing, C ABI (include/lrf_hip.h).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <algorithm>
#include <functional>
#include <vector>

#include "../../include/lrf_hip.h"
#include "lrf_gram_kernels.hip"
#include "lrf_kernels.hip"
#include "lrf_svd_kernels.hip"
#include "lrf_bigrank_kernels.hip"
#include "lrf_midrank_kernels.hip"
#include "lrf_bcdw_kernel.hip"
#include "lrf_bcdp_kernel.hip"
#include "lrf_bcdw16_kernel.hip"
#include "lrf_bcdw32_kernel.hip"
#include "lrf_anyshape_kernels.hip"

// the planes qmf_encode forms hold YCbCr samples, 0 or in [0.114, 255.5]: all below 2^8 and exact on the grid 2^(8-35)
// (run_init: selects k_gram64's integer digit extraction; callers with arbitrary X pass LRF_GRAM_EXP_FROM_DATA)
#define LRF_PLANES_GRAM_EXP 8
// largest rank of the 64-column BCD kernels (k_bcd_w <= 8, k_bcd <= 16, k_bcd_mid <= 32); above it the any-shape kernels iterate
#define LRF_BIG_TO_ANY_RANK 32
#define LRF_TABLE_SETS 6 // descriptor-table sets a context keeps resident (upload_tables)
#define LRF_BCDW_MIN_BLOCKS 1024 // smaller rank <= 8 runs iterate on the workgroup kernel k_bcd (run_bcd)
#define LRF_BCDW16_MIN_BLOCKS 1024 // likewise for rank <= 16 runs and k_bcd_w16
#define LRF_PERSIST_MIN_BLOCKS 3584 // a rank <= 8 call of this many blocks runs its iterations 2..K in one launch (k_bcd_p)
#define LRF_SHARE_MIN_BLOCKS 3072   // LRF_SHARES=2|3: a rank <= 8 call of this many blocks runs as two image shares (run_two_shares)
#define LRF_BCDW32_MIN_BLOCKS 128  // likewise for rank 17..32 runs and k_bcd_w32 / k_bcd_w32f (12 images: 1.06 -> 0.99 ms at (20,10,10))

static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return set_err(LRF_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                             __FILE__, __LINE__);                                         \
    } while (0)

// Makes the context's device current for the duration of one ABI call and restores the caller's device on the way out
// (a torch process encoding a cuda:1 tensor while its current device is cuda:0 must not find cuda:1 current afterwards).
struct DevGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DevGuard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DevGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};
#define LRF_ON_DEVICE(c)                                                                                  \
    DevGuard dev_guard_((c)->device);                                                                     \
    if (dev_guard_.err != hipSuccess)                                                                     \
        return set_err(LRF_EHIP, "selecting device %d failed: %s", (c)->device, hipGetErrorString(dev_guard_.err))

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct lrf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    DevBuf planes, blocks, gchunks, vf, wf, bf, ppart, qpart, x, sign;
    DevBuf gpart, gexp; // exact Gram partials (128-bit integers per chunk) and per-matrix grid exponents (lrf_gram_kernels.hip)
    DevBuf sx, sg, svn, swn, suf, smm; // SVD baseline workspace
    DevBuf any_uf, any_vf, any_a, any_b, any_p, any_e2, any_g, any_td; // any-shape path (lrf_anyshape_host.inc)
    DevBuf vf16, wf16, bf16, pp16, qp16; // the pitch-16 tables of a call that mixes kernel families (plan_runs)
    // host staging for descriptor tables (pinned)
    void* h_stage = nullptr;
    size_t h_stage_cap = 0;
    // profiling
    bool profile = false;
    unsigned profile_mask = ~0u; // kernel ids (bit per LRF_K_*) that get event pairs while `profile` is on
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[LRF_K_COUNT];
    std::vector<hipEvent_t> ev_pool;
    double acc_ms[LRF_K_COUNT] = {0};
    long acc_n[LRF_K_COUNT] = {0};
    int init_sweeps = 0; // developer aid: stop k_init after stage n (0 = run everything)
    std::vector<char> table_key; // bytes of the descriptor tables now resident on the device (planes / blocks)
    // earlier tables, least recently used one replaced: calls that alternate between a few geometries (a pipeline slot sees
    // its full sub-batch size and the two or three sizes of the tapered tail) find them resident and skip the synchronising upload
    struct TableSet {
        DevBuf planes, blocks, gchunks;
        std::vector<char> key;
        unsigned long stamp = 0;
    };
    TableSet talt[LRF_TABLE_SETS - 1];
    unsigned long tstamp = 0;
    unsigned attr_done = 0;      // hipFuncSetAttribute call sites already executed for this context's device (bit per site)
    // Kernel families of one call on streams of their own (run_init / run_bcd): the runs of plan_runs touch disjoint planes, so
    // the whole chain of a run — initialisation, b table, K x (U update, V update) — is independent of the other runs'; the
    // first run stays on `stream`, the others fork behind the Gram pass and are joined at the end of run_bcd.  Created on
    // first use (a call with 1024 blocks or more — 256 with a rank above 16 — that mixes rank families); never while kernel profiling is on.
    hipStream_t fam_stream[2] = {nullptr, nullptr};
    hipEvent_t fam_fork = nullptr, fam_join[2] = {nullptr, nullptr};
    bool fam_parallel = false;   // set by the fused entry points whose run_init is followed by run_bcd at once
    bool fam_forked = false;     // run_init forked: run_bcd uses the same streams and joins
    hipEvent_t planes_done = nullptr; // set by a pipe: recorded after the planes kernel of lrf_qmf_encode_rgb_u8 (input buffer free)
    // two image shares of one large call on two streams (run_two_shares; an experiment, off unless LRF_SHARES=2 or 3): per
    // share the events Gram done / initial tables done / U update done / V update done
    // the persistent iteration kernel (k_bcd_p; experiment, LRF_PERSIST=1): its queue head, tickets and flags; its error word
    // comes back into page-locked host memory behind every launch and is looked at by the next call / lrf_ctx_synchronize
    DevBuf psync;
    int* h_perr = nullptr;     // page-locked: k_bcd_p writes it directly when a poll expires
    bool psync_dirty = false;  // the queue state of k_bcd_p is not all-zero (a failed launch)
    hipEvent_t share_ev[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
};

static int ensure(lrf_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return LRF_OK;
    if (b.p) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + bytes / 8;
    hipError_t e = hipMalloc(&b.p, cap);
    if (e != hipSuccess) return set_err(LRF_ENOMEM, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
    b.cap = cap;
    return LRF_OK;
}

static int upload(lrf_ctx* c, DevBuf& b, const void* src, size_t bytes)
{
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    if (bytes > c->h_stage_cap) {
        if (c->h_stage) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipHostFree(c->h_stage));
        }
        HIP_TRY(hipHostMalloc(&c->h_stage, bytes * 2, hipHostMallocDefault));
        c->h_stage_cap = bytes * 2;
    }
    // the staging buffer may still be read by an earlier async copy
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(c->h_stage, src, bytes);
    HIP_TRY(hipMemcpyAsync(b.p, c->h_stage, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRF_OK;
}

struct Prof {
    lrf_ctx* c;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    Prof(lrf_ctx* c_, int id_) : c(c_), id(id_), on(c_->profile && ((c_->profile_mask >> id_) & 1u))
    {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, c->stream);
    }
    ~Prof()
    {
        if (!on) return;
        (void)hipEventRecord(b, c->stream);
        c->ev[id].push_back({a, b});
    }
};

static void fold_events(lrf_ctx* c)
{
    for (int k = 0; k < LRF_K_COUNT; k++) {
        for (auto& pr : c->ev[k]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                c->acc_ms[k] += ms;
                c->acc_n[k] += 1;
            }
            c->ev_pool.push_back(pr.first);
            c->ev_pool.push_back(pr.second);
        }
        c->ev[k].clear();
    }
}

// ---- geometry ---------------------------------------------------------------------------------
static void plane_dims(int64_t H, int64_t W, int c, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M)
{
    // F.interpolate(scale_factor=0.5): output size = floor(input * 0.5) (lrf/compression/qmf.py:230)
    int64_t ph = c ? (int64_t)floor((double)H * 0.5) : H, pw = c ? (int64_t)floor((double)W * 0.5) : W;
    *h = ph;
    *w = pw;
    *hp = ph + (8 - ph % 8) % 8;
    *wp = pw + (8 - pw % 8) % 8;
    *M = (*hp / 8) * (*wp / 8);
}

static int make_geom(int64_t H, int64_t W, ImageGeom* g)
{
    long xoff = 0;
    for (int c = 0; c < 3; c++) {
        int64_t h, w, hp, wp, M;
        plane_dims(H, W, c, &h, &w, &hp, &wp, &M);
        if (h < 1 || w < 1) return set_err(LRF_EINVAL, "image %ldx%ld too small", (long)H, (long)W);
        // reflect padding needs pad < size (torch raises otherwise)
        if ((hp - h) / 2 >= h || (hp - h) - (hp - h) / 2 >= h || (wp - w) / 2 >= w || (wp - w) - (wp - w) / 2 >= w)
            return set_err(LRF_EINVAL, "reflect padding larger than the plane (%ldx%ld)", (long)h, (long)w);
        PlaneGeom& p = g->p[c];
        p.h = (int)h; p.w = (int)w; p.hp = (int)hp; p.wp = (int)wp;
        p.top = (int)((hp - h) / 2); p.left = (int)((wp - w) / 2);
        p.top_crop = p.top; p.left_crop = p.left;
        p.nw = (int)(wp / 8);
        p.nh = (int)(hp / 8);
        p.pr0 = c ? g->p[c - 1].pr0 + g->p[c - 1].nh : 0;
        p.M = (int)M;
        p.xoff = xoff;
        p.o4 = xoff / 4;
        xoff += M * 64;
    }
    g->img_floats = xoff;
    g->tot4 = xoff / 4;
    return LRF_OK;
}

// ---- descriptor tables ------------------------------------------------------------------------
struct Tables {
    std::vector<PlaneDesc> planes;
    std::vector<BlockDesc> blocks;
    std::vector<GramChunk> gchunks;
    // image shares (encode_rgb_prepare / run_two_shares): the table is ordered share by share; share s owns the images
    // [share_img0[s], share_img0[s+1]) and the planes [share_plane0[s], share_plane0[s+1])
    int shares = 1;
    long share_img0[3] = {0, 0, 0};
    int share_plane0[3] = {0, 0, 0};
};

static void add_plane(Tables& t, long x_off, long u_off, long v_off, long u0_off, long v0_off, int M, int R, int sign_off)
{
    PlaneDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.x_off = x_off; pd.u_off = u_off; pd.v_off = v_off; pd.u0_off = u0_off; pd.v0_off = v0_off;
    pd.M = M; pd.R = R;
    pd.blk0 = (int)t.blocks.size();
    pd.nblk = (M + LRF_KC - 1) / LRF_KC;
    pd.native_t2_u = ((long)(R - 1) * M < 400) ? 1 : 0;
    pd.sign_off = sign_off;
    int pi = (int)t.planes.size();
    for (int b = 0; b < pd.nblk; b++) t.blocks.push_back(BlockDesc{pi, b * LRF_KC, b, 0});
    pd.gch0 = 0; // the Gram chunks are cut when the table is complete (finish_gram_chunks)
    pd.ngch = 0;
    t.planes.push_back(pd);
}

static int check_params(int64_t M, int64_t N, int R, int K, int lo, int hi)
{
    if (N != LRF_PATCH_ELEMS) return set_err(LRF_ENOTSUP, "N=%ld: only N=%d (8x8 patches) is implemented", (long)N, LRF_PATCH_ELEMS);
    if (R < 1) return set_err(LRF_EINVAL, "rank must be >= 1 (got %d)", R);
    if (R > LRF_MAX_RANK) return set_err(LRF_ENOTSUP, "rank %d > %d not implemented", R, LRF_MAX_RANK);
    if (K < 1) return set_err(LRF_ENOTSUP, "num_iters=%d: use lrf_qmf_svd_init_f32 for K=0", K);
    if (lo > hi || lo < -128 || hi > 127) return set_err(LRF_EINVAL, "bounds (%d,%d) outside int8", lo, hi);
    if (M < 1) return set_err(LRF_EINVAL, "M must be >= 1");
    // u.mT @ u: each 384-row block partial is an exact integer in fp32 for any int8 bounds (384 * 128^2 < 2^24), whatever the
    // order inside the block, and the block partials are added in block order like the reference's sgemm (K blocked by 384),
    // so the result is the reference's even where the running sum leaves the exact range.
    (void)M;
    return LRF_OK;
}

// padded rank of the V / W / partial tables: 16 (one MFMA tile, the tuned kernels) or 64 (lrf_bigrank_kernels.hip)
static int table_rmax(const Tables& t)
{
    int rmax = 1;
    for (const PlaneDesc& pd : t.planes) rmax = pd.R > rmax ? pd.R : rmax;
    return rmax;
}
static int table_rp(const Tables& t) { return table_rmax(t) <= 16 ? 16 : LRF_RPB; }

// ---- kernel families of a call -----------------------------------------------------------------
// A run: consecutive planes (and their blocks) that iterate on one kernel family — 0: rank <= 8 (k_bcd_w), 1: rank <= 16
// (k_bcd<., 16>), 2: rank <= 32 (k_bcd_mid) — with that family's table pitch (16 or LRF_RPB).  A small call takes ONE family,
// the one its largest rank needs: its launches are latency chains per block and a second launch per iteration costs more than
// a faster kernel saves.  From 1024 blocks on (256 with a rank above 16: plan_runs) every plane
// goes to its own family (256 images: (16,8,8) 4.05 -> 3.78 ms, (20,10,10) 7.07 -> see DESIGN.md); the planes of the fused
// encode are ordered by channel, so that is at most three runs.  Pitch-16 runs of a call whose table pitch is LRF_RPB use
// the second table set (vf16 ...): the regions of the two pitches would overlap in one buffer.
struct FamRun {
    int plane0, nplanes, block0, nblocks, rmax, fam, pitch;
    int rmin;        // smallest rank of the run (k_bcd_w32 takes runs whose ranks are all 17..32)
    bool any_native; // some plane of the run is small enough for ATen's native order of `uu @ bb` ((R-1) M < 400)
};
static int fam_of_rank(int R) { return R <= 8 ? 0 : (R <= 16 ? 1 : 2); }
static bool bcd_wave_variant()
{
    static const bool v = !(getenv("LRF_BCD_WG") && getenv("LRF_BCD_WG")[0] == '1'); // LRF_BCD_WG=1: k_bcd instead of k_bcd_w
    return v;
}
static std::vector<FamRun> plan_runs(const Tables& t)
{
    static const bool no_split = getenv("LRF_NO_FAMILY_SPLIT") && getenv("LRF_NO_FAMILY_SPLIT")[0] == '1'; // developer comparison aid
    const int rmax_t = table_rmax(t);
    // since the families of a call run side by side on streams of their own (round 3) the split pays from 1024 blocks on
    // (64 x 512x768: (16,8,8) 0.93 -> 0.89 ms, (20,10,10) 2.04 -> 1.36 with k_bcd_w32 on the luma run); calls with a rank above
    // 16 split from 256 blocks (24 images: (20,10,10) 1.27 -> 1.04 ms, 12 images 1.06 -> 0.99).  It was 3072 while the runs shared one stream.
    static const long env_blocks = getenv("LRF_FAMILY_SPLIT_BLOCKS") ? atol(getenv("LRF_FAMILY_SPLIT_BLOCKS")) : -1; // developer aid
    const long min_blocks = env_blocks >= 0 ? env_blocks : (rmax_t > 16 ? 256 : 1024);
    const bool split = !no_split && bcd_wave_variant() && rmax_t <= LRF_BIG_TO_ANY_RANK && (long)t.blocks.size() >= min_blocks;
    std::vector<FamRun> runs;
    for (int p = 0; p < (int)t.planes.size(); p++) {
        const PlaneDesc& pd = t.planes[p];
        const int fam = split ? fam_of_rank(pd.R) : (rmax_t > 16 ? 2 : fam_of_rank(rmax_t));
        if (runs.empty() || runs.back().fam != fam) runs.push_back(FamRun{p, 0, pd.blk0, 0, 1, fam, fam == 2 ? LRF_RPB : 16, pd.R, false});
        FamRun& r = runs.back();
        r.any_native = r.any_native || pd.native_t2_u != 0;
        r.nplanes++;
        r.nblocks += pd.nblk;
        r.rmax = pd.R > r.rmax ? pd.R : r.rmax;
        r.rmin = pd.R < r.rmin ? pd.R : r.rmin;
    }
    return runs;
}
static bool plan_is_mixed(const std::vector<FamRun>& runs)
{
    bool p16 = false, p64 = false;
    for (const FamRun& r : runs) (r.pitch == 16 ? p16 : p64) = true;
    return p16 && p64;
}
// the V / W / b / partial tables a run uses
struct FamBufs {
    float *vf, *wf, *bf, *pp, *qp;
};
static FamBufs run_bufs(lrf_ctx* c, const FamRun& r, bool mixed)
{
    if (mixed && r.pitch == 16) return FamBufs{(float*)c->vf16.p, (float*)c->wf16.p, (float*)c->bf16.p, (float*)c->pp16.p, (float*)c->qp16.p};
    return FamBufs{(float*)c->vf.p, (float*)c->wf.p, (float*)c->bf.p, (float*)c->ppart.p, (float*)c->qpart.p};
}

static hipStream_t run_stream(lrf_ctx* c, size_t run_idx) { return (c->fam_forked && run_idx > 0) ? c->fam_stream[run_idx - 1] : c->stream; }
static int fam_fork_streams(lrf_ctx* c, size_t nruns)
{
    static const bool off = getenv("LRF_NO_FAMILY_STREAMS") && getenv("LRF_NO_FAMILY_STREAMS")[0] == '1'; // developer comparison aid
    c->fam_forked = false;
    if (off || !c->fam_parallel || c->profile || nruns < 2 || nruns > 3) return LRF_OK;
    if (!c->fam_fork) HIP_TRY(hipEventCreateWithFlags(&c->fam_fork, hipEventDisableTiming));
    for (size_t i = 0; i + 1 < nruns; i++) {
        if (!c->fam_stream[i]) HIP_TRY(hipStreamCreateWithFlags(&c->fam_stream[i], hipStreamNonBlocking));
        if (!c->fam_join[i]) HIP_TRY(hipEventCreateWithFlags(&c->fam_join[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(c->fam_fork, c->stream));
    for (size_t i = 0; i + 1 < nruns; i++) HIP_TRY(hipStreamWaitEvent(c->fam_stream[i], c->fam_fork, 0));
    c->fam_forked = true;
    return LRF_OK;
}
static int fam_join_streams(lrf_ctx* c, size_t nruns)
{
    if (!c->fam_forked) return LRF_OK;
    c->fam_forked = false;
    for (size_t i = 0; i + 1 < nruns && i < 2; i++) {
        if (!c->fam_stream[i] || !c->fam_join[i]) continue;
        HIP_TRY(hipEventRecord(c->fam_join[i], c->fam_stream[i]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->fam_join[i], 0));
    }
    return LRF_OK;
}

// Row chunks of the exact Gram pass (k_gram64: one workgroup per chunk, one 128-bit partial per chunk for k_init to add):
// LRF_GRAM_ROWS rows each — fewer for small calls, so that the pass still has a few workgroups per CU (a chunk is a latency
// chain of 64-row blocks: one 512x768 image 38 -> 14 us, 64 images 57 -> 44 us).  Exact integer sums: the cut does not change a bit.
static void finish_gram_chunks(Tables& t)
{
    long rows = 0;
    for (const PlaneDesc& pd : t.planes) rows += pd.M;
    int per = LRF_GRAM_ROWS;
    while (per > 384 && rows / per < 768) per >>= 1;
    t.gchunks.clear();
    for (int pi = 0; pi < (int)t.planes.size(); pi++) {
        PlaneDesc& pd = t.planes[pi];
        pd.gch0 = (int)t.gchunks.size();
        pd.ngch = (pd.M + per - 1) / per;
        for (int g = 0; g < pd.ngch; g++) {
            const int row0 = g * per;
            t.gchunks.push_back(GramChunk{pi, row0, pd.gch0 + g, pd.M - row0 < per ? pd.M - row0 : per});
        }
    }
}

static int upload_tables(lrf_ctx* c, Tables& t)
{
    finish_gram_chunks(t);
    // the tables only depend on the call's geometry: skip the (synchronising) upload when nothing changed
    // (the Gram chunk table follows from the plane table: it is not part of the key)
    size_t pb = t.planes.size() * sizeof(PlaneDesc), bb = t.blocks.size() * sizeof(BlockDesc);
    std::vector<char> key(pb + bb);
    memcpy(key.data(), t.planes.data(), pb);
    memcpy(key.data() + pb, t.blocks.data(), bb);
    if (key == c->table_key) return LRF_OK;
    int victim = 0;
    for (int j = 0; j < LRF_TABLE_SETS - 1; j++) {
        lrf_ctx::TableSet& a = c->talt[j];
        const bool hit = a.key == key;
        if (hit || a.stamp < c->talt[victim].stamp) victim = j;
        if (hit) break;
    }
    { // the current set goes to the victim's place, the victim's buffers become current (its tables, if this was a hit)
        lrf_ctx::TableSet& a = c->talt[victim];
        std::swap(c->planes, a.planes);
        std::swap(c->blocks, a.blocks);
        std::swap(c->gchunks, a.gchunks);
        c->table_key.swap(a.key);
        a.stamp = ++c->tstamp;
    }
    if (key == c->table_key) return LRF_OK;
    c->table_key.clear();
    int rc = upload(c, c->planes, t.planes.data(), pb);
    if (rc) return rc;
    rc = upload(c, c->blocks, t.blocks.data(), bb);
    if (rc) return rc;
    rc = upload(c, c->gchunks, t.gchunks.data(), t.gchunks.size() * sizeof(GramChunk));
    if (rc) return rc;
    if ((rc = ensure(c, c->gpart, t.gchunks.size() * (size_t)LRF_GRAM_SLOT * sizeof(ulonglong2)))) return rc;
    if ((rc = ensure(c, c->gexp, t.planes.size() * sizeof(int)))) return rc;
    size_t np = t.planes.size(), nb = t.blocks.size();
    size_t rp = (size_t)table_rp(t), gts = rp == 16 ? (size_t)LRF_GT_STRIDE : (size_t)LRF_GTB_STRIDE;
    if ((rc = ensure(c, c->vf, np * 64 * rp * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->wf, np * 64 * rp * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->bf, np * gts * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->ppart, nb * 64 * rp * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->qpart, nb * rp * rp * sizeof(float)))) return rc;
    if (plan_is_mixed(plan_runs(t))) {
        if ((rc = ensure(c, c->vf16, np * 64 * 16 * sizeof(float)))) return rc;
        if ((rc = ensure(c, c->wf16, np * 64 * 16 * sizeof(float)))) return rc;
        if ((rc = ensure(c, c->bf16, np * (size_t)LRF_GT_STRIDE * sizeof(float)))) return rc;
        if ((rc = ensure(c, c->pp16, nb * 64 * 16 * sizeof(float)))) return rc;
        if ((rc = ensure(c, c->qp16, nb * 16 * 16 * sizeof(float)))) return rc;
    }
    c->table_key.swap(key);
    return LRF_OK;
}

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// gram_exp: the fixed-point grid exponent of the exact Gram matrix (max|x| < 2^gram_exp) when the caller knows it — 8 for the
// planes of qmf_encode — or LRF_GRAM_EXP_FROM_DATA: one more pass over X finds it per matrix
static int run_init(lrf_ctx* c, const float* X, const Tables& t, const int8_t* sign_dev, int gram_exp)
{
    if (!(c->attr_done & (1u << 0))) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<8>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<16>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<64>)));
        c->attr_done |= 1u << 0;
    }
    int rmax = table_rmax(t), rp = table_rp(t), nplanes = (int)t.planes.size();
    {
        Prof p(c, LRF_K_GRAM);
        if (gram_exp == LRF_GRAM_EXP_FROM_DATA) {
            hipLaunchKernelGGL(k_gram_exponent, dim3(nplanes), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p, (int*)c->gexp.p);
            LAUNCH_CHECK();
        }
        if (gram_exp == LRF_PLANES_GRAM_EXP)
            hipLaunchKernelGGL(k_gram64<true>, dim3((unsigned)t.gchunks.size()), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p,
                               (const GramChunk*)c->gchunks.p, (const int*)c->gexp.p, gram_exp, (ulonglong2*)c->gpart.p);
        else
            hipLaunchKernelGGL(k_gram64<false>, dim3((unsigned)t.gchunks.size()), dim3(256), 0, c->stream, X, (const PlaneDesc*)c->planes.p,
                               (const GramChunk*)c->gchunks.p, (const int*)c->gexp.p, gram_exp, (ulonglong2*)c->gpart.p);
        LAUNCH_CHECK();
    }
    Prof p(c, LRF_K_INIT);
    const std::vector<FamRun> runs = plan_runs(t);
    const bool mixed = plan_is_mixed(runs);
    {
        int rcf = fam_fork_streams(c, runs.size());
        if (rcf) return rcf;
    }
    for (size_t ri = 0; ri < runs.size(); ri++) {
        const FamRun& r = runs[ri];
        hipStream_t rs = run_stream(c, ri);
        const FamBufs fb = run_bufs(c, r, mixed);
#define LRF_LAUNCH_INIT(ZR)                                                                                          \
    hipLaunchKernelGGL(k_init<ZR>, dim3(r.nplanes), dim3(256), sizeof(InitLds<ZR>), rs, (const ulonglong2*)c->gpart.p, \
                       (const int*)c->gexp.p, gram_exp, (const PlaneDesc*)c->planes.p, sign_dev, fb.vf, fb.wf, c->init_sweeps, r.pitch, r.plane0)
        if (r.rmax <= 8) LRF_LAUNCH_INIT(8);
        else if (r.rmax <= 16) LRF_LAUNCH_INIT(16);
        else LRF_LAUNCH_INIT(64);
#undef LRF_LAUNCH_INIT
        LAUNCH_CHECK();
    }
    (void)rmax;
    (void)rp;
    (void)nplanes;
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
    if (!(c->attr_done & (1u << 2))) {
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
    }
    // the b tables of the initial V
    for (size_t ri = 0; ri < runs.size(); ri++) {
        const FamRun& r = runs[ri];
        hipStream_t rs = run_stream(c, ri);
        const FamBufs fb = run_bufs(c, r, mixed);
        if (r.pitch == 16)
            hipLaunchKernelGGL(k_bprep, dim3(r.nplanes), dim3(256), 0, rs, pl, (const float*)fb.vf, fb.bf, r.plane0);
        else
            hipLaunchKernelGGL(k_bprep_big, dim3(r.nplanes), dim3(256), 0, rs, pl, (const float*)fb.vf, fb.bf, r.plane0);
        LAUNCH_CHECK();
    }
    // Iterations >= 2 with bounds where every term and partial sum of `uu @ bb` is an exact integer in fp32 for the largest
    // rank of a run ((R - 1) 64 mx^3 < 2^24): the order of that sum is immaterial, which lets ranks 9..16 (gs_row_lds) and
    // 17..32 (k_bcd_mid) replace the reference's dependent chain by independent fmas, bit for bit
    const long mx_b = abs(lo) > abs(hi) ? abs(lo) : abs(hi);
    static const bool exact_off = getenv("LRF_GENERIC_GS") && getenv("LRF_GENERIC_GS")[0] == '1'; // developer comparison aid
    static const bool w32_off = getenv("LRF_NO_BCDW32") && getenv("LRF_NO_BCDW32")[0] == '1'; // developer comparison aid: k_bcd_mid<0> instead
    static const long w32_min = getenv("LRF_BCDW32_MIN_BLOCKS") ? atol(getenv("LRF_BCDW32_MIN_BLOCKS")) : LRF_BCDW32_MIN_BLOCKS; // developer aid
    static const long w16_min = getenv("LRF_BCDW16_MIN_BLOCKS") ? atol(getenv("LRF_BCDW16_MIN_BLOCKS")) : LRF_BCDW16_MIN_BLOCKS; // developer aid
    // Iterations 2..K of a large single-run rank <= 8 call in ONE launch (k_bcd_p, lrf_bcdp_kernel.hip; round 4): the U updates of
    // all iterations pulled from a queue, each matrix's V update done by the last of its blocks to finish.  From 3584 blocks
    // on (256 x 512x768: 2.05 -> 1.93 ms per step; 64 x 1365x2048: 3.47 -> 3.27 ms; tools/dev_persist_threshold.py, 512x768
    // images: 48 / 64 images lose 15 %, 96 win 3 %, 128 lose 1.5 % — a round and a half of the 2048 wave slots —, 160 win 3 %,
    // 192 win 7 %); what decides is the number of blocks, not the matrices' size (tools/dev_persist_small.py: 1200 x 173x264
    // -3 %, 600 x 352x288 -9 %; 300 x 173x264, 1200 blocks: +8 %).  LRF_PERSIST=0 turns it off, =1 forces it from
    // LRF_BCDW_MIN_BLOCKS blocks on (tests).
    static const int persist_env = getenv("LRF_PERSIST") ? atoi(getenv("LRF_PERSIST")) : -1;
    const bool persist_ok = runs.size() == 1 && runs[0].fam == 0 && wave_variant && K >= 2 && !c->fam_forked &&
                            !(c->profile && (c->profile_mask & ~((1u << LRF_K_BCD) | (1u << LRF_K_BCD_PERSIST))) != 0);
    const bool use_persist = persist_ok && persist_env != 0 &&
                             runs[0].nblocks >= (persist_env == 1 ? LRF_BCDW_MIN_BLOCKS : LRF_PERSIST_MIN_BLOCKS);
    if (use_persist) {
        if (!(c->attr_done & (1u << 3))) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_p, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
            c->attr_done |= 1u << 3;
        }
        if (!c->h_perr) {
            HIP_TRY(hipHostMalloc((void**)&c->h_perr, sizeof(int), hipHostMallocDefault));
            *c->h_perr = 0;
        }
        if (*c->h_perr) {
            *c->h_perr = 0;
            c->psync_dirty = true;
            return set_err(LRF_EHIP, "k_bcd_p: a wave's poll for a V update expired in an earlier call on this context");
        }
    }
    for (int it = 0; it < K; it++) {
        if (use_persist && it == 1) {
            const FamRun& r = runs[0];
            const size_t sbytes = sizeof(BcdpSync) + (2 * (size_t)r.nplanes + (size_t)r.nblocks) * sizeof(int); // (+ a debug count per block)
            const void* before = c->psync.p;
            int rcp = ensure(c, c->psync, sbytes);
            if (rcp) return rcp;
            if (c->psync.p != before || c->psync_dirty) { // a launch leaves the state zeroed (its last wave); a new buffer or a failed launch does not
                HIP_TRY(hipMemsetAsync(c->psync.p, 0, c->psync.cap, c->stream));
                c->psync_dirty = false;
            }
            const int total_waves = (K - 1) * r.nblocks;
            int wgs = (total_waves + LRF_BCDW_WAVES - 1) / LRF_BCDW_WAVES;
            if (wgs > 512) wgs = 512; // two workgroups per CU resident; later ones would only find the queue empty
            {
                Prof p(c, LRF_K_BCD_PERSIST);
                hipLaunchKernelGGL(k_bcd_p, dim3((unsigned)wgs), dim3(64 * LRF_BCDW_WAVES), LRF_BCDW_LDS, c->stream, X, pl, bl + r.block0,
                                   (float*)c->vf.p, (float*)c->bf.p, U, (float*)c->ppart.p, (float*)c->qpart.p, V, gp, r.nblocks, K - 1,
                                   r.nplanes, r.plane0, (BcdpSync*)c->psync.p, c->h_perr, 2 * r.nplanes);
                LAUNCH_CHECK();
            }
#ifdef LRF_BCDP_DEBUG
            {
                std::vector<int> cnt(r.nblocks);
                HIP_TRY(hipStreamSynchronize(c->stream));
                HIP_TRY(hipMemcpy(cnt.data(), ((BcdpSync*)c->psync.p)->cell + 2 * r.nplanes, r.nblocks * sizeof(int), hipMemcpyDeviceToHost));
                int want = 0, bad = 0;
                for (int i = 0; i < K - 1; i++) want += 64 * (1 + 1000 * i);
                for (int b = 0; b < r.nblocks; b++)
                    if (cnt[b] != want && bad++ < 8) fprintf(stderr, "[k_bcd_p debug] block %d processed-count code %d (want %d)\n", b, cnt[b], want);
                fprintf(stderr, "[k_bcd_p debug] %d blocks, %d with a wrong count, err %d\n", r.nblocks, bad, *c->h_perr);
                c->psync_dirty = true; // the per-block counts are not cleared by the kernel
            }
#endif
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
#define LRF_LAUNCH_MID(MODE)                                                                                         \
    hipLaunchKernelGGL((k_bcd_mid<MODE>), dim3(nbr), dim3(256), sizeof(MidLds<MODE>), rs, X, pl, blr, (const float*)fb.vf, \
                       (const float*)fb.wf, (const float*)fb.bf, U0, U, fb.pp, fb.qp, gpr)
                if (r.fam == 2 && mode == 0 && wave_variant && !w32_off && gpr.exact_int && 64 * mx_b * mx_b <= 32767 && r.rmin >= 17 &&
                    nbr >= w32_min) {
                    // ranks 17..32, iterations >= 2, exact-integer bounds with |b| within int16: one wave per block, lane = row
                    // Gauss-Seidel on int16 pairs (lrf_bcdw32_kernel.hip)
#define LRF_LAUNCH_W32(NP)                                                                                           \
    hipLaunchKernelGGL((k_bcd_w32<NP>), dim3((nbr + LRF_BCDW32_WAVES - 1) / LRF_BCDW32_WAVES), dim3(64 * LRF_BCDW32_WAVES), LRF_BCDW32_LDS(NP), \
                       rs, X, pl, blr, (const float*)fb.vf, (const float*)fb.bf, U, fb.pp, fb.qp, gpr, nbr)
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
                } else if (r.fam == 2 && mode == 1 && wave_variant && !w32_off && !r.any_native && r.rmin == r.rmax && r.rmin >= 17 && nbr >= w32_min) {
                    // ranks 17..32, first iteration (old U = X W0, the reference's ordered chain): one wave per block, lane = row
#define LRF_LAUNCH_W32F(RR)                                                                                          \
    case RR:                                                                                                         \
        hipLaunchKernelGGL((k_bcd_w32f<RR>), dim3(nbr), dim3(64), LRF_BCDW32F_LDS, rs, X, pl, blr, (const float*)fb.vf,  \
                           (const float*)fb.wf, (const float*)fb.bf, U, fb.pp, fb.qp, gpr, nbr);                     \
        break;
                    switch (r.rmax) {
                        LRF_LAUNCH_W32F(17) LRF_LAUNCH_W32F(18) LRF_LAUNCH_W32F(19) LRF_LAUNCH_W32F(20) LRF_LAUNCH_W32F(21) LRF_LAUNCH_W32F(22)
                        LRF_LAUNCH_W32F(23) LRF_LAUNCH_W32F(24) LRF_LAUNCH_W32F(25) LRF_LAUNCH_W32F(26) LRF_LAUNCH_W32F(27) LRF_LAUNCH_W32F(28)
                        LRF_LAUNCH_W32F(29) LRF_LAUNCH_W32F(30) LRF_LAUNCH_W32F(31) LRF_LAUNCH_W32F(32)
                    }
#undef LRF_LAUNCH_W32F
                } else if (r.fam == 2) {
                    if (mode == 1) LRF_LAUNCH_MID(1);
                    else if (mode == 2) LRF_LAUNCH_MID(2);
                    else LRF_LAUNCH_MID(0);
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
#undef LRF_LAUNCH_MID
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
                if (r.fam == 2)
                    hipLaunchKernelGGL(k_vupdate_mid, dim3(r.nplanes), dim3(256), sizeof(BigVLds), rs, pl, (const float*)fb.pp,
                                       (const float*)fb.qp, fb.vf, fb.bf, V, gp.lo, gp.hi, last, r.plane0);
                else if (r.fam == 0)
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

// ---- experiment (off by default): two image shares of one large call on two streams -----------------------------------
// A large rank <= 8 call is a chain of 3 + 2 K launches, each waiting for the last workgroup of its predecessor: the chip
// drains and refills 2 K + 2 times, and three of the kernels (k_init, k_bprep, k_vupdate) are latency chains per matrix that
// leave it mostly idle.  Cut into two image shares A, B the same launches can interleave on TWO streams:
//   LRF_SHARES=2  heavy (the caller's stream): planes A, Gram A, planes B, Gram B, then U-update A1, B1, A2, B2, ...
//                 latency (the context's own): init A, b-table A, init B, b-table B, then V-update A1, B1, A2, ...
//                 with an event per dependency (Gram s -> init s; b-table / V-update s -> next U-update s; U-update s -> V-update s):
//                 the streaming kernels never overlap each other, the latency kernels of one share run under the U update of the other;
//   LRF_SHARES=3  each share's whole chain on a stream of its own (one fork, one join).
// Arithmetic, tables and outputs are those of run_init + run_bcd (same kernels on sub-ranges of the same tables): bit-identical.
// Measured: both are SLOWER or no faster than the single stream on this runtime, which is why
// neither is the default.
static int run_two_shares(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, const ImageGeom& g, const Tables& t, int K,
                          int lo, int hi, const int8_t* sign_dev, int8_t* U, int8_t* V)
{
    const PlaneDesc* pl = (const PlaneDesc*)c->planes.p;
    const BlockDesc* bl = (const BlockDesc*)c->blocks.p;
    float* X = (float*)c->x.p;
    if (!(c->attr_done & (1u << 0))) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<8>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<16>)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_init<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(InitLds<64>)));
        c->attr_done |= 1u << 0;
    }
    if (!(c->attr_done & (1u << 1))) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW16_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)k_bcd_w16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LRF_BCDW16_LDS));
        c->attr_done |= 1u << 1;
    }
    if (!c->fam_stream[0]) HIP_TRY(hipStreamCreateWithFlags(&c->fam_stream[0], hipStreamNonBlocking));
    for (int sh = 0; sh < 2; sh++)
        for (int j = 0; j < 4; j++)
            if (!c->share_ev[sh][j]) HIP_TRY(hipEventCreateWithFlags(&c->share_ev[sh][j], hipEventDisableTiming));
    static const bool indep = getenv("LRF_SHARES") && atoi(getenv("LRF_SHARES")) == 3;
    // hs[s]: the stream of share s's streaming kernels (planes, Gram, U update); ls[s]: of its latency kernels
    hipStream_t hs[2] = {c->stream, indep ? c->fam_stream[0] : c->stream};
    hipStream_t ls[2] = {indep ? c->stream : c->fam_stream[0], c->fam_stream[0]};
    if (indep) {
        if (!c->fam_fork) HIP_TRY(hipEventCreateWithFlags(&c->fam_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->fam_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->fam_stream[0], c->fam_fork, 0));
    }
    auto link = [&](hipEvent_t ev, hipStream_t from, hipStream_t to) -> int { // `to` continues behind what `from` has queued
        if (from == to) return LRF_OK;
        HIP_TRY(hipEventRecord(ev, from));
        HIP_TRY(hipStreamWaitEvent(to, ev, 0));
        return LRF_OK;
    };
    enum { EV_GRAM = 0, EV_TABLES = 1, EV_U = 2, EV_V = 3 };
    struct Share { int plane0, nplanes, block0, nblocks, gch0, ngch; long img0, nimg; } shr[2];
    for (int sh = 0; sh < 2; sh++) {
        Share& s = shr[sh];
        s.plane0 = t.share_plane0[sh];
        s.nplanes = t.share_plane0[sh + 1] - s.plane0;
        s.block0 = t.planes[s.plane0].blk0;
        s.gch0 = t.planes[s.plane0].gch0;
        s.nblocks = 0;
        s.ngch = 0;
        for (int p = s.plane0; p < s.plane0 + s.nplanes; p++) {
            s.nblocks += t.planes[p].nblk;
            s.ngch += t.planes[p].ngch;
        }
        s.img0 = t.share_img0[sh];
        s.nimg = t.share_img0[sh + 1] - s.img0;
    }
    GsParams gp = make_gs(lo, hi);
    hipStream_t caller = c->stream;
    int rc = LRF_OK;
    for (int sh = 0; sh < 2 && !rc; sh++) {
        const Share& s = shr[sh];
        c->stream = hs[sh]; // lrf_qmf_planes_from_rgb_u8 and Prof launch / record on c->stream
        rc = lrf_qmf_planes_from_rgb_u8(c, rgb + (size_t)s.img0 * 3 * H * W, s.nimg, H, W, X + s.img0 * g.img_floats);
        if (!rc && sh == 1 && c->planes_done) {
            if (hs[1] != caller) rc = link(c->share_ev[1][EV_GRAM], hs[1], caller);
            if (!rc && hipEventRecord(c->planes_done, caller) != hipSuccess) rc = set_err(LRF_EHIP, "hipEventRecord failed");
        }
        if (!rc) {
            Prof p(c, LRF_K_GRAM);
            hipLaunchKernelGGL(k_gram64<true>, dim3((unsigned)s.ngch), dim3(256), 0, hs[sh], X, pl, (const GramChunk*)c->gchunks.p + s.gch0,
                               (const int*)c->gexp.p, LRF_PLANES_GRAM_EXP, (ulonglong2*)c->gpart.p);
        }
        c->stream = caller;
        if (rc) break;
        LAUNCH_CHECK();
        if ((rc = link(c->share_ev[sh][EV_GRAM], hs[sh], ls[sh]))) break;
        hipLaunchKernelGGL(k_init<8>, dim3(s.nplanes), dim3(256), sizeof(InitLds<8>), ls[sh], (const ulonglong2*)c->gpart.p, (const int*)c->gexp.p,
                           LRF_PLANES_GRAM_EXP, pl, sign_dev, (float*)c->vf.p, (float*)c->wf.p, c->init_sweeps, 16, s.plane0);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bprep, dim3(s.nplanes), dim3(256), 0, ls[sh], pl, (const float*)c->vf.p, (float*)c->bf.p, s.plane0);
        LAUNCH_CHECK();
    }
    if (rc) return rc;
    for (int it = 0; it < K; it++) {
        const int last = it == K - 1 ? 1 : 0;
        for (int sh = 0; sh < 2; sh++) {
            const Share& s = shr[sh];
            const BlockDesc* blr = bl + s.block0;
            if ((rc = link(c->share_ev[sh][it == 0 ? EV_TABLES : EV_V], ls[sh], hs[sh]))) return rc;
            {
                c->stream = hs[sh];
                Prof p(c, LRF_K_BCD);
                c->stream = caller;
                const dim3 grid((unsigned)((s.nblocks + LRF_BCDW_WAVES - 1) / LRF_BCDW_WAVES)), block(64 * LRF_BCDW_WAVES);
                if (it == 0)
                    hipLaunchKernelGGL((k_bcd_w<1>), grid, block, LRF_BCDW_LDS, hs[sh], (const float*)X, pl, blr, (const float*)c->vf.p,
                                       (const float*)c->wf.p, (const float*)c->bf.p, (const float*)nullptr, U, (float*)c->ppart.p,
                                       (float*)c->qpart.p, gp, s.nblocks);
                else
                    hipLaunchKernelGGL((k_bcd_w<0>), grid, block, LRF_BCDW_LDS, hs[sh], (const float*)X, pl, blr, (const float*)c->vf.p,
                                       (const float*)c->wf.p, (const float*)c->bf.p, (const float*)nullptr, U, (float*)c->ppart.p,
                                       (float*)c->qpart.p, gp, s.nblocks);
                c->stream = hs[sh];
            }
            c->stream = caller;
            LAUNCH_CHECK();
            if ((rc = link(c->share_ev[sh][EV_U], hs[sh], ls[sh]))) return rc;
            hipLaunchKernelGGL(k_vupdate<8>, dim3(s.nplanes), dim3(256), 0, ls[sh], pl, (const float*)c->ppart.p, (const float*)c->qpart.p,
                               (float*)c->vf.p, (float*)c->bf.p, V, gp, last, s.plane0);
            LAUNCH_CHECK();
        }
    }
    for (int sh = 0; sh < 2; sh++)
        if ((rc = link(c->share_ev[sh][EV_V], ls[sh], caller))) return rc;
    return LRF_OK;
}

// ---- C ABI ------------------------------------------------------------------------------------
#include "lrf_anyshape_host.inc"

extern "C" {

const char* lrf_last_error(void) { return g_err; }


int lrf_version(void) { return 1; }

int lrf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lrf_ctx_create(int device, lrf_ctx** out)
{
    if (!out) return set_err(LRF_EINVAL, "out is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return set_err(LRF_EINVAL, "device %d out of range (%d visible)", device, n);
    DevGuard dev_guard_(device);
    if (dev_guard_.err != hipSuccess) return set_err(LRF_EHIP, "selecting device %d failed: %s", device, hipGetErrorString(dev_guard_.err));
    lrf_ctx* c = new lrf_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return set_err(LRF_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    if (const char* e = getenv("LRF_DEBUG_INIT_SWEEPS")) c->init_sweeps = atoi(e); // developer timing aid only
    *out = c;
    return LRF_OK;
}

void lrf_ctx_destroy(lrf_ctx* c)
{
    if (!c) return;
    DevGuard dev_guard_(c->device);
    (void)hipStreamSynchronize(c->stream);
    fold_events(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->gchunks, &c->gpart, &c->gexp, &c->planes, &c->blocks, &c->vf, &c->wf, &c->bf, &c->ppart, &c->qpart, &c->x, &c->sign,
                      &c->sx, &c->sg, &c->svn, &c->swn, &c->suf, &c->smm,
                      &c->any_uf, &c->any_vf, &c->any_a, &c->any_b, &c->any_p, &c->any_e2, &c->any_g, &c->any_td,
                      &c->vf16, &c->wf16, &c->bf16, &c->pp16, &c->qp16};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& a : c->talt) {
        DevBuf* tb[] = {&a.planes, &a.blocks, &a.gchunks};
        for (DevBuf* b : tb)
            if (b->p) (void)hipFree(b->p);
    }
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_perr) (void)hipHostFree(c->h_perr);
    if (c->psync.p) (void)hipFree(c->psync.p);
    for (int i = 0; i < 2; i++) {
        if (c->fam_stream[i]) {
            (void)hipStreamSynchronize(c->fam_stream[i]);
            (void)hipStreamDestroy(c->fam_stream[i]);
        }
        if (c->fam_join[i]) (void)hipEventDestroy(c->fam_join[i]);
    }
    if (c->fam_fork) (void)hipEventDestroy(c->fam_fork);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++)
            if (c->share_ev[i][j]) (void)hipEventDestroy(c->share_ev[i][j]);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int lrf_ctx_set_stream(lrf_ctx* c, void* hip_stream)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    hipStream_t s = (hipStream_t)hip_stream; // NULL is HIP's default stream, a valid handle
    if (s == c->stream) return LRF_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = s;
    return LRF_OK;
}

int lrf_ctx_use_own_stream(lrf_ctx* c)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    if (c->stream == c->own_stream) return LRF_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return LRF_OK;
}

int lrf_ctx_synchronize(lrf_ctx* c)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    LRF_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_perr && *c->h_perr) {
        *c->h_perr = 0;
        c->psync_dirty = true;
        return set_err(LRF_EHIP, "k_bcd_p: a wave's poll for a V update expired (the results of that call are invalid)");
    }
    return LRF_OK;
}

size_t lrf_ctx_workspace_bytes(const lrf_ctx* c)
{
    if (!c) return 0;
    const DevBuf* bufs[] = {&c->gchunks, &c->gpart, &c->gexp, &c->planes, &c->blocks, &c->vf, &c->wf, &c->bf, &c->ppart, &c->qpart, &c->x, &c->sign,
                            &c->sx, &c->sg, &c->svn, &c->swn, &c->suf, &c->smm,
                            &c->any_uf, &c->any_vf, &c->any_a, &c->any_b, &c->any_p, &c->any_e2, &c->any_g, &c->any_td,
                      &c->vf16, &c->wf16, &c->bf16, &c->pp16, &c->qp16};
    size_t total = 0;
    for (const DevBuf* b : bufs) total += b->cap;
    for (const auto& a : c->talt) total += a.planes.cap + a.blocks.cap + a.gchunks.cap;
    return total;
}

int lrf_ctx_trim(lrf_ctx* c)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    LRF_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    DevBuf* bufs[] = {&c->gchunks, &c->gpart, &c->gexp, &c->planes, &c->blocks, &c->vf,
                      &c->wf, &c->bf, &c->ppart, &c->qpart, &c->x, &c->sign, &c->sx, &c->sg, &c->svn, &c->swn, &c->suf, &c->smm,
                      &c->any_uf, &c->any_vf, &c->any_a, &c->any_b, &c->any_p, &c->any_e2, &c->any_g, &c->any_td,
                      &c->vf16, &c->wf16, &c->bf16, &c->pp16, &c->qp16};
    for (DevBuf* b : bufs) {
        if (b->p) HIP_TRY(hipFree(b->p));
        b->p = nullptr;
        b->cap = 0;
    }
    for (auto& a : c->talt) {
        DevBuf* tb[] = {&a.planes, &a.blocks, &a.gchunks};
        for (DevBuf* b : tb) {
            if (b->p) HIP_TRY(hipFree(b->p));
            b->p = nullptr;
            b->cap = 0;
        }
        a.key.clear();
        a.stamp = 0;
    }
    c->table_key.clear();     // the descriptor tables went with their buffers
    return LRF_OK;
}

int lrf_ctx_profile(lrf_ctx* c, int enable)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    c->profile = enable != 0;
    c->profile_mask = ~0u;
    return LRF_OK;
}

int lrf_ctx_profile_kernels(lrf_ctx* c, unsigned mask)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    c->profile = mask != 0;
    c->profile_mask = mask;
    return LRF_OK;
}

int lrf_ctx_kernel_time(lrf_ctx* c, int id, double* total_ms, long* launches)
{
    if (!c || id < 0 || id >= LRF_K_COUNT) return set_err(LRF_EINVAL, "bad kernel id");
    HIP_TRY(hipStreamSynchronize(c->stream));
    fold_events(c);
    if (total_ms) *total_ms = c->acc_ms[id];
    if (launches) *launches = c->acc_n[id];
    return LRF_OK;
}

int lrf_ctx_profile_reset(lrf_ctx* c)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    fold_events(c);
    for (int k = 0; k < LRF_K_COUNT; k++) { c->acc_ms[k] = 0; c->acc_n[k] = 0; }
    return LRF_OK;
}

int lrf_malloc(lrf_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return set_err(LRF_EINVAL, "NULL argument");
    LRF_ON_DEVICE(c);
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) return set_err(LRF_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return LRF_OK;
}

int lrf_free(lrf_ctx* c, void* p)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    if (p) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(p));
    }
    return LRF_OK;
}

int lrf_memcpy_h2d(lrf_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRF_OK;
}

int lrf_memcpy_d2h(lrf_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRF_OK;
}

int lrf_plane_dims(int64_t H, int64_t W, int c, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M)
{
    if (c < 0 || c > 2 || H < 1 || W < 1 || !h || !w || !hp || !wp || !M) return set_err(LRF_EINVAL, "bad argument");
    plane_dims(H, W, c, h, w, hp, wp, M);
    return LRF_OK;
}

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
    static const bool no_tiled = getenv("LRF_PLANES_NO_TILED") && getenv("LRF_PLANES_NO_TILED")[0] == '1'; // developer comparison aid
    if (H % 16 == 0 && W % 16 == 0 && (reinterpret_cast<uintptr_t>(rgb) & 7) == 0 && !no_tiled)
        hipLaunchKernelGGL(k_planes16, dim3((unsigned)((H / 16) * ((g.p[0].nw + 31) / 32)), (unsigned)B), dim3(256), 0, c->stream, rgb, (int)H,
                           (int)W, g, X);
    else if (!no_tiled) {
        // any other size: the same tiling over the padded planes (k_planes_strip); blocks dealt so that the strips of an image
        // stay on one XCD (block index mod 8 is the XCD when the grid's x extent is a multiple of 8)
        static const bool no_xcd = getenv("LRF_PLANES_NO_XCD") && getenv("LRF_PLANES_NO_XCD")[0] == '1'; // developer comparison aid
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
    static const bool force_any = getenv("LRF_FORCE_ANY") && getenv("LRF_FORCE_ANY")[0] == '1'; // developer comparison aid
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
    c->fam_parallel = true; // run_init is followed by run_bcd at once: the kernel families of the call may run side by side
    rc = run_init(c, X, t, sign, LRF_GRAM_EXP_FROM_DATA);
    c->fam_parallel = false;
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

// The part of lrf_qmf_encode_rgb_u8 that may allocate or upload: argument checks, the X workspace, the plane / block tables of
// B images (resident afterwards: upload_tables).  A pipe calls it for every sub-batch size of a submission before any
// transfer is in flight, so that nothing synchronises or allocates once its threads and streams are busy.
struct EncodePlan {
    ImageGeom g;
    Tables t;
    long u_img = 0, v_img = 0, uoff[3], voff[3], u0c[4] = {0, 0, 0, 0}, v0c[4] = {0, 0, 0, 0};
};
static int encode_rgb_prepare(lrf_ctx* c, int64_t B, int64_t H, int64_t W, const int R[3], int K, int lo, int hi, bool with_sign,
                              EncodePlan& ep)
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
    // Two image shares (run_two_shares; an experiment: LRF_SHARES=2 or 3) for a large call whose planes all iterate on
    // k_bcd_w: the first half of the images, then the second, each luma first
    long nblk_img = 0;
    for (int ch = 0; ch < 3; ch++) nblk_img += (g.p[ch].M + LRF_KC - 1) / LRF_KC;
    static const int shares_mode = getenv("LRF_SHARES") ? atoi(getenv("LRF_SHARES")) : 0;
    const bool two = shares_mode >= 2 && bcd_wave_variant() && R[0] <= 8 && R[1] <= 8 && R[2] <= 8 && B >= 2 && B * nblk_img >= LRF_SHARE_MIN_BLOCKS;
    ep.t.shares = two ? 2 : 1;
    ep.t.share_img0[0] = 0;
    ep.t.share_img0[1] = two ? B / 2 : B;
    ep.t.share_img0[2] = B;
    for (int sh = 0; sh < ep.t.shares; sh++) {
        ep.t.share_plane0[sh] = (int)ep.t.planes.size();
        for (int ch = 0; ch < 3; ch++)
            for (int64_t b = ep.t.share_img0[sh]; b < ep.t.share_img0[sh + 1]; b++)
                add_plane(ep.t, b * g.img_floats + g.p[ch].xoff, b * ep.u_img + ep.uoff[ch], b * ep.v_img + ep.voff[ch],
                          ep.u0c[ch] + b * (long)g.p[ch].M * R[ch], ep.v0c[ch] + b * 64L * R[ch], g.p[ch].M, R[ch],
                          with_sign ? (int)(b * s_img + soff[ch]) : -1);
    }
    for (int sh = ep.t.shares; sh <= 2; sh++) ep.t.share_plane0[sh] = (int)ep.t.planes.size();
    return upload_tables(c, ep.t);
}

int lrf_qmf_encode_rgb_u8(lrf_ctx* c, const uint8_t* rgb, int64_t B, int64_t H, int64_t W, const int R[3], int K, int lo,
                          int hi, const int8_t* sign, int8_t* U, int8_t* V)
{
    if (!c || !rgb || !R || !U || !V) return set_err(LRF_EINVAL, "NULL argument");
    LRF_ON_DEVICE(c);
    EncodePlan ep;
    int rc = encode_rgb_prepare(c, B, H, W, R, K, lo, hi, sign != nullptr, ep);
    if (rc) return rc;
    const ImageGeom& g = ep.g;
    Tables& t = ep.t;
    const long u_img = ep.u_img, v_img = ep.v_img;
    const long *uoff = ep.uoff, *voff = ep.voff, *u0c = ep.u0c, *v0c = ep.v0c;
    float* X = (float*)c->x.p;
    // two image shares (experiment): only while no kernel carries profiling events (those are recorded on the caller's stream)
    if (t.shares == 2 && !c->profile) return run_two_shares(c, rgb, B, H, W, g, t, K, lo, hi, sign, U, V);
    if ((rc = lrf_qmf_planes_from_rgb_u8(c, rgb, B, H, W, X))) return rc;
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
    c->fam_parallel = true; // run_init is followed by run_bcd at once: the kernel families of the call may run side by side
    rc = run_init(c, X, t, sign, LRF_PLANES_GRAM_EXP);
    c->fam_parallel = false;
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
    static const bool no_tiled = getenv("LRF_DECODE_NO_TILED") && getenv("LRF_DECODE_NO_TILED")[0] == '1'; // developer comparison aid
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


/* ---- host -> host pipelined encoder (the protocol of lrf/utils/misc.py:90-100: host tensor in,
 * encoded factors back on the host) ----
 * A pipe owns `slots` independent encoder contexts, each with its own stream, scratch workspace and device staging for one
 * sub-batch.  Sub-batch i goes to slot i % slots as  H2D(rgb) -> planes -> init -> K x (U update, V update) -> D2H(U, V),
 * all on the slot's stream: the copies of one slot overlap the kernels of the others (separate SDMA queues, full-duplex link),
 * and the latency-bound initialisation of one sub-batch overlaps the HBM-bound iterations of another. */
struct PipeSlot {
    lrf_ctx* ctx = nullptr;
    DevBuf rgb, u, v, sign;
    hipEvent_t h2d_done = nullptr;  // recorded on the upload stream: this slot's input has landed
    hipEvent_t rgb_free = nullptr;  // recorded on the slot's stream after the planes kernel: the input may be overwritten
};
struct lrf_pipe {
    int device = 0;
    int64_t sub_batch = 0;
    hipStream_t h2d = nullptr;      // all uploads, in order: one sub-batch at a time gets the whole link, so the first one
                                    // lands early and its kernels run under the uploads of the following ones
    hipStream_t d2h = nullptr;      // all downloads of factors, in order: a slot's next kernels do not queue behind its copies
                                    // (LRF_PIPE_NO_D2H=1 at creation: downloads on the slot's own stream, as in round 2)
    hipEvent_t sign_done = nullptr; // the batch's sign vectors have landed
    std::vector<PipeSlot> slots;
    DevBuf sign;                    // the whole batch's sign vectors, uploaded once per call ahead of the first sub-batch
    std::vector<hipEvent_t> done;   // one per sub-batch of the call in flight: its factors are in the caller's buffers
    std::vector<hipEvent_t> kdone;  // one per sub-batch: its kernels have finished (the download stream waits for it)
    std::vector<int64_t> first, count;
    size_t next_wait = 0;
};

static int pipe_ensure(lrf_pipe* p, PipeSlot& s, DevBuf& b, size_t bytes) { (void)p; return ensure(s.ctx, b, bytes); }

int lrf_pipe_create(int device, int slots, int64_t sub_batch, lrf_pipe** out)
{
    if (!out) return set_err(LRF_EINVAL, "out is NULL");
    if (slots < 1 || slots > 8) return set_err(LRF_EINVAL, "slots=%d out of range [1,8]", slots);
    if (sub_batch < 0 || sub_batch > 65535) return set_err(LRF_EINVAL, "sub_batch=%ld out of range [0,65535]", (long)sub_batch);
    lrf_pipe* p = new lrf_pipe();
    p->device = device;
    p->sub_batch = sub_batch;
    p->slots.resize((size_t)slots);
    for (auto& s : p->slots) {
        int rc = lrf_ctx_create(device, &s.ctx);
        if (rc) {
            lrf_pipe_destroy(p);
            return rc;
        }
    }
    DevGuard dev_guard_(device);
    hipError_t e = hipStreamCreateWithFlags(&p->h2d, hipStreamNonBlocking);
    if (e == hipSuccess && !getenv("LRF_PIPE_NO_D2H")) e = hipStreamCreateWithFlags(&p->d2h, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->sign_done, hipEventDisableTiming);
    for (auto& s : p->slots) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.h2d_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.rgb_free, hipEventDisableTiming);
        s.ctx->planes_done = s.rgb_free;
    }
    if (e != hipSuccess) {
        lrf_pipe_destroy(p);
        return set_err(LRF_EHIP, "creating the pipe's stream / events failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return LRF_OK;
}

void lrf_pipe_destroy(lrf_pipe* p)
{
    if (!p) return;
    DevGuard dev_guard_(p->device);
    if (p->h2d) (void)hipStreamSynchronize(p->h2d);
    if (p->d2h) (void)hipStreamSynchronize(p->d2h);
    for (auto& s : p->slots) {
        if (!s.ctx) continue;
        (void)hipStreamSynchronize(s.ctx->stream);
        DevBuf* bufs[] = {&s.rgb, &s.u, &s.v, &s.sign};
        for (DevBuf* b : bufs)
            if (b->p) (void)hipFree(b->p);
        s.ctx->planes_done = nullptr;
        lrf_ctx_destroy(s.ctx);
        if (s.h2d_done) (void)hipEventDestroy(s.h2d_done);
        if (s.rgb_free) (void)hipEventDestroy(s.rgb_free);
    }
    if (p->h2d) (void)hipStreamDestroy(p->h2d);
    if (p->d2h) (void)hipStreamDestroy(p->d2h);
    if (p->sign_done) (void)hipEventDestroy(p->sign_done);
    if (p->sign.p) (void)hipFree(p->sign.p);
    for (auto e : p->done) (void)hipEventDestroy(e);
    for (auto e : p->kdone) (void)hipEventDestroy(e);
    delete p;
}

int lrf_pipe_slots(const lrf_pipe* p) { return p ? (int)p->slots.size() : 0; }

lrf_ctx* lrf_pipe_slot_ctx(lrf_pipe* p, int slot)
{
    if (!p || slot < 0 || slot >= lrf_pipe_slots(p)) return nullptr;
    return p->slots[(size_t)slot].ctx;
}

size_t lrf_pipe_workspace_bytes(const lrf_pipe* p)
{
    if (!p) return 0;
    size_t total = 0;
    for (const auto& s : p->slots) total += lrf_ctx_workspace_bytes(s.ctx) + s.rgb.cap + s.u.cap + s.v.cap + s.sign.cap;
    return total + p->sign.cap;
}

/* The sub-batches of a submission.  A pipe created with an explicit sub_batch cuts the batch into equal pieces (the last one
 * shorter).  Left to the library (sub_batch 0): pieces of about 80 MB of input (1.4 ms of a Gen5 x16 link; 64 images of
 * 512x768) or a sixteenth of the batch, whichever is larger — smaller ones are bound by the per-matrix latency chain of the
 * initialisation and by this thread's launch rate (a CLIC-sized batch in 4-image pieces: 128 x 22 launches), and every piece
 * costs ~20 us of idle link between two copies of the upload stream — and a tapered tail: what happens after the last byte
 * has landed is the kernels of the LAST piece, which no transfer hides, so the batch ends with a piece of three quarters and
 * one of a quarter of the regular size (256 x 512x768: 64, 64, 64, 48, 16; the 48's kernels run on the other slot's stream
 * under the upload and the kernels of the 16).  Measured (tools/dev_pipe_taper.py): the taper and the
 * larger pieces are worth 1-2 % (6.14 -> 6.07 ms); a stream of its own for the last piece's kernels, or two upload streams
 * with two copies in flight, made it slower (every further stream costs more in the runtime's cross-stream waits than
 * the idle time it removes).  LRF_PIPE_TAIL="a,b,.." overrides the tail and
 * LRF_PIPE_BULK the regular size (developer aids: tools/dev_pipe_sweep.py, tests/test_pipeline.py). */
static std::vector<int64_t> pipe_schedule(const lrf_pipe* p, int64_t B, int64_t H, int64_t W)
{
    std::vector<int64_t> sizes;
    if (p->sub_batch > 0) {
        for (int64_t b = 0; b < B; b += p->sub_batch) sizes.push_back(b + p->sub_batch <= B ? p->sub_batch : B - b);
        return sizes;
    }
    const int64_t img = 3 * H * W;
    int64_t bytes = (int64_t)80 << 20;
    if (B * img / 16 > bytes) bytes = B * img / 16;
    int64_t sb = bytes / img;
    if (sb >= 8) sb -= sb % 8;
    if (sb < 1) sb = 1;
    if (sb > 1024) sb = 1024;
    if (const char* e = getenv("LRF_PIPE_BULK")) // developer aid: images per regular piece of the tapered schedule
        if (atol(e) > 0) sb = atol(e);
    std::vector<int64_t> tail;
    if (const char* e = getenv("LRF_PIPE_TAIL")) {
        for (const char* q = e; *q;) {
            char* end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            if (v > 0) tail.push_back(v);
            q = *end == ',' ? end + 1 : end;
        }
    } else if (sb >= 4) {
        tail.push_back(sb - sb / 4);
        tail.push_back(sb / 4);
    }
    int64_t tail_sum = 0;
    for (int64_t v : tail) tail_sum += v;
    if (B < sb + tail_sum) { // too small for a regular piece and the tail: equal pieces, as for an explicit size
        for (int64_t b = 0; b < B; b += sb) sizes.push_back(b + sb <= B ? sb : B - b);
        return sizes;
    }
    int64_t rem = B - tail_sum;
    while (rem >= 2 * sb) {
        sizes.push_back(sb);
        rem -= sb;
    }
    if (rem > sb) { // two pieces of about half of what is left, rather than a regular one and a sliver
        sizes.push_back(rem - rem / 2);
        rem = rem / 2;
    }
    if (rem > 0) sizes.push_back(rem);
    for (int64_t v : tail) sizes.push_back(v);
    return sizes;
}

int lrf_pipe_qmf_encode_submit(lrf_pipe* p, const uint8_t* rgb_host, int64_t B, int64_t H, int64_t W, const int R[3], int K, int lo,
                               int hi, const int8_t* sign_host, int8_t* U_host, int8_t* V_host, int* n_sub)
{
    if (!p || !rgb_host || !R || !U_host || !V_host) return set_err(LRF_EINVAL, "NULL argument");
    if (B < 1) return set_err(LRF_EINVAL, "B must be >= 1");
    if (p->next_wait < p->count.size()) return set_err(LRF_EINVAL, "the previous submission has not been waited for");
    ImageGeom g;
    int rc = make_geom(H, W, &g);
    if (rc) return rc;
    for (int ch = 0; ch < 3; ch++)
        if ((rc = check_params(g.p[ch].M, 64, R[ch], K, lo, hi))) return rc;
    long u_img = 0, v_img = 0, s_img = R[0] + R[1] + R[2];
    for (int ch = 0; ch < 3; ch++) {
        u_img += (long)g.p[ch].M * R[ch];
        v_img += 64L * R[ch];
    }
    const size_t img_bytes = (size_t)3 * H * W;
    const std::vector<int64_t> sizes = pipe_schedule(p, B, H, W);
    const size_t nsub = sizes.size(), S = (size_t)lrf_pipe_slots(p);
    // sub-batch -> slot: the slots in turn.  prev[i]: the sub-batch that had i's slot before.
    std::vector<int64_t> first(nsub);
    std::vector<size_t> slot_of(nsub);
    std::vector<long> prev(nsub, -1);
    int64_t sb_max = 0;
    for (size_t i = 0, b = 0; i < nsub; b += (size_t)sizes[i], i++) {
        first[i] = (int64_t)b;
        sb_max = sizes[i] > sb_max ? sizes[i] : sb_max;
        slot_of[i] = i % S;
        if (slot_of[i] < S && i >= S) prev[i] = (long)(i - S);
    }
    DevGuard dev_guard_(p->device);
    if (dev_guard_.err != hipSuccess) return set_err(LRF_EHIP, "selecting device %d failed", p->device);
    while (p->done.size() < nsub) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->done.push_back(e);
    }
    while (p->kdone.size() < nsub) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->kdone.push_back(e);
    }
    p->first.clear();
    p->count.clear();
    p->next_wait = 0;
    // Everything that allocates, uploads a descriptor table or synchronises happens here, before the first transfer: per slot
    // the buffers for its largest piece, the workspace and the tables of every piece size it will see (largest first, so
    // that the workspace is allocated once; a context keeps LRF_TABLE_SETS table sets resident).
    if (sign_host && (rc = ensure(p->slots[0].ctx, p->sign, (size_t)B * s_img))) return rc;
    for (size_t sl = 0; sl < p->slots.size(); sl++) {
        PipeSlot& s = p->slots[sl];
        std::vector<int64_t> seen;
        for (size_t i = 0; i < nsub; i++)
            if (slot_of[i] == sl && std::find(seen.begin(), seen.end(), sizes[i]) == seen.end()) seen.push_back(sizes[i]);
        if (seen.empty()) continue;
        std::sort(seen.begin(), seen.end(), std::greater<int64_t>());
        if ((rc = pipe_ensure(p, s, s.rgb, (size_t)seen[0] * img_bytes))) return rc;
        if ((rc = pipe_ensure(p, s, s.u, (size_t)seen[0] * u_img))) return rc;
        if ((rc = pipe_ensure(p, s, s.v, (size_t)seen[0] * v_img))) return rc;
        if (seen.size() <= LRF_TABLE_SETS) // (more sizes than table sets: the calls below upload as they go, still correct)
            for (int64_t nb : seen) {
                EncodePlan ep;
                if ((rc = encode_rgb_prepare(s.ctx, nb, H, W, R, K, lo, hi, sign_host != nullptr, ep))) return rc;
            }
    }
    // upload of sub-batch i: on the upload stream, once the planes kernel of the sub-batch that used the slot before has read
    // its input; enqueued one sub-batch ahead of the kernels so that the link never waits for this host thread
    auto enqueue_upload = [&](size_t i) -> int {
        PipeSlot& s = p->slots[slot_of[i]];
        hipStream_t up = p->h2d;
        HIP_TRY(hipStreamWaitEvent(up, s.rgb_free, 0));
        HIP_TRY(hipMemcpyAsync(s.rgb.p, rgb_host + (size_t)first[i] * img_bytes, (size_t)sizes[i] * img_bytes, hipMemcpyHostToDevice, up));
        HIP_TRY(hipEventRecord(s.h2d_done, up));
        return LRF_OK;
    };
    // The sign vectors are read by the initialisation kernel, long after the planes kernel has released the slot's input
    // buffer: they get a buffer of their own for the whole batch (a few bytes per image), first thing on an upload stream.
    if (sign_host) {
        HIP_TRY(hipMemcpyAsync(p->sign.p, sign_host, (size_t)B * s_img, hipMemcpyHostToDevice, p->h2d));
        HIP_TRY(hipEventRecord(p->sign_done, p->h2d));
    }
    // kernels of sub-batch i on its slot's stream, behind its upload and behind the download of the slot's previous factors
    auto enqueue_kernels = [&](size_t i) -> int {
        PipeSlot& s = p->slots[slot_of[i]];
        hipStream_t st = s.ctx->stream;
        int rc2;
        HIP_TRY(hipStreamWaitEvent(st, s.h2d_done, 0));
        if (sign_host) HIP_TRY(hipStreamWaitEvent(st, p->sign_done, 0));
        if (prev[i] >= 0 && p->d2h) HIP_TRY(hipStreamWaitEvent(st, p->done[(size_t)prev[i]], 0));
        if ((rc2 = lrf_qmf_encode_rgb_u8(s.ctx, (const uint8_t*)s.rgb.p, sizes[i], H, W, R, K, lo, hi,
                                         sign_host ? (const int8_t*)p->sign.p + (size_t)first[i] * s_img : nullptr, (int8_t*)s.u.p, (int8_t*)s.v.p)))
            return rc2;
        if (p->d2h) HIP_TRY(hipEventRecord(p->kdone[i], st));
        return LRF_OK;
    };
    // its factors to the caller's buffers, on the download stream
    auto enqueue_download = [&](size_t i) -> int {
        PipeSlot& s = p->slots[slot_of[i]];
        hipStream_t dn = p->d2h ? p->d2h : s.ctx->stream;
        if (p->d2h) HIP_TRY(hipStreamWaitEvent(dn, p->kdone[i], 0));
        HIP_TRY(hipMemcpyAsync(U_host + (size_t)first[i] * u_img, s.u.p, (size_t)sizes[i] * u_img, hipMemcpyDeviceToHost, dn));
        HIP_TRY(hipMemcpyAsync(V_host + (size_t)first[i] * v_img, s.v.p, (size_t)sizes[i] * v_img, hipMemcpyDeviceToHost, dn));
        HIP_TRY(hipEventRecord(p->done[i], dn));
        p->first.push_back(first[i]);
        p->count.push_back(sizes[i]);
        return LRF_OK;
    };
    // Pageable source memory (what a torch tensor is unless it was pinned): hipMemcpyAsync from it returns when the transfer is
    // over, so a single submitting thread enqueues the kernels of sub-batch i only after upload i + 1 and the pipeline drains
    // (256 x 512x768: 9.7 ms against 6.1 ms from page-locked memory).  Then a second thread issues the uploads, in order;
    // the two threads hand over through two counters: the kernels of sub-batch i wait for upload i to have been issued (its
    // event must be RECORDED before a stream can wait for it), the upload into a slot for the kernels of the sub-batch that
    // had the slot before (which record rgb_free, the event that says the slot's input buffer may be overwritten).  The uploader only copies and records
    // events; allocations and table uploads were done above.
    bool pageable = false;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, rgb_host) == hipSuccess) pageable = attr.type == hipMemoryTypeUnregistered;
        else { (void)hipGetLastError(); pageable = true; } // some runtimes report an error for an unregistered pointer
    }
    if (pageable && S > 1 && nsub > 1) {
        struct Handover {
            std::mutex m;
            std::condition_variable cv;
            size_t uploaded = 0, enqueued = 0;
            int err = 0;
            char msg[512] = "";
        } ho;
        std::thread uploader;
        try {
            uploader = std::thread([&]() {
                DevGuard guard(p->device);
                for (size_t i = 0; i < nsub; i++) {
                    {
                        std::unique_lock<std::mutex> lk(ho.m);
                        ho.cv.wait(lk, [&] { return ho.err != 0 || prev[i] < 0 || ho.enqueued > (size_t)prev[i]; });
                        if (ho.err) return;
                    }
                    const int rcu = enqueue_upload(i);
                    std::unique_lock<std::mutex> lk(ho.m);
                    if (rcu) {
                        ho.err = rcu;
                        snprintf(ho.msg, sizeof(ho.msg), "%s", lrf_last_error()); // this thread's message, for the caller's thread
                    } else {
                        ho.uploaded = i + 1;
                    }
                    ho.cv.notify_all();
                    if (rcu) return;
                }
            });
        } catch (const std::exception& ex) { // no exception crosses the C ABI
            return set_err(LRF_EHIP, "starting the upload thread failed: %s", ex.what());
        }
        int rcm = LRF_OK;
        for (size_t i = 0; i < nsub && rcm == LRF_OK; i++) {
            {
                std::unique_lock<std::mutex> lk(ho.m);
                ho.cv.wait(lk, [&] { return ho.err != 0 || ho.uploaded > i; });
                if (ho.err) break;
            }
            rcm = enqueue_kernels(i);
            if (rcm == LRF_OK) rcm = enqueue_download(i);
            std::unique_lock<std::mutex> lk(ho.m);
            if (rcm) ho.err = rcm;
            else ho.enqueued = i + 1;
            ho.cv.notify_all();
        }
        uploader.join();
        if (rcm) return rcm;
        if (ho.err) return set_err(ho.err, "%s", ho.msg);
    } else {
        if ((rc = enqueue_upload(0))) return rc;
        for (size_t i = 0; i < nsub; i++) {
            // With one slot the next upload overwrites the buffer this sub-batch still has to read: it is enqueued after the
            // kernels (which record rgb_free); with more slots it goes first, so that the link never waits for this thread.
            if (i + 1 < nsub && S > 1 && (rc = enqueue_upload(i + 1))) return rc;
            if ((rc = enqueue_kernels(i))) return rc;
            if (i + 1 < nsub && S == 1 && (rc = enqueue_upload(i + 1))) return rc;
            if ((rc = enqueue_download(i))) return rc;
        }
    }
    if (n_sub) *n_sub = (int)nsub;
    return LRF_OK;
}

int lrf_pipe_wait_next(lrf_pipe* p, int64_t* first_image, int64_t* n_images)
{
    if (!p) return set_err(LRF_EINVAL, "pipe is NULL");
    if (p->next_wait >= p->count.size()) {
        if (first_image) *first_image = 0;
        if (n_images) *n_images = 0;
        return LRF_OK;
    }
    DevGuard dev_guard_(p->device);
    HIP_TRY(hipEventSynchronize(p->done[p->next_wait]));
    if (first_image) *first_image = p->first[p->next_wait];
    if (n_images) *n_images = p->count[p->next_wait];
    p->next_wait++;
    return LRF_OK;
}

int lrf_pipe_qmf_encode_rgb_u8_host(lrf_pipe* p, const uint8_t* rgb_host, int64_t B, int64_t H, int64_t W, const int R[3], int K,
                                    int lo, int hi, const int8_t* sign_host, int8_t* U_host, int8_t* V_host)
{
    int rc = lrf_pipe_qmf_encode_submit(p, rgb_host, B, H, W, R, K, lo, hi, sign_host, U_host, V_host, nullptr);
    // whatever was enqueued must have finished before the caller may touch (or free) its buffers, also after an error
    int64_t n = 1;
    while (n > 0) {
        int rc2 = lrf_pipe_wait_next(p, nullptr, &n);
        if (rc2) return rc ? rc : rc2;
    }
    if (rc && p) {
        DevGuard dev_guard_(p->device);
        (void)hipStreamSynchronize(p->h2d);
        for (auto& s : p->slots) (void)hipStreamSynchronize(s.ctx->stream);
    }
    return rc;
}

int lrf_host_alloc(size_t bytes, void** out)
{
    if (!out) return set_err(LRF_EINVAL, "out is NULL");
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return set_err(LRF_ENOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return LRF_OK;
}

int lrf_host_free(void* p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return LRF_OK;
}

int lrf_host_register(void* p, size_t bytes)
{
    if (!p) return set_err(LRF_EINVAL, "NULL argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return LRF_OK;
}

int lrf_host_unregister(void* p)
{
    if (!p) return set_err(LRF_EINVAL, "NULL argument");
    HIP_TRY(hipHostUnregister(p));
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

// Gram matrices of B matrices [M,192] of uint8-valued floats (svd_encode, RGB colour-space branch): exact, int8 MFMA
// int32 sums hold 128^2 x 131072 rows; longer matrices take the fp64 kernel (float X only)
#define LRF_G192_MAX_ROWS 100000
extern "C++" {
template <typename T>
static int gram192_u8(lrf_ctx* c, const T* X, long xs, int B, int M, double* G)
{
    static const bool use_f64 = getenv("LRF_GRAM192_F64") && getenv("LRF_GRAM192_F64")[0] == '1'; // developer comparison aid
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
    static const bool f32_matrix = getenv("LRF_SVD_F32_MATRIX") && getenv("LRF_SVD_F32_MATRIX")[0] == '1'; // developer comparison aid
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
