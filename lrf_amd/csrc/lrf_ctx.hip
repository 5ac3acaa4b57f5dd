// lrf_ctx.hip — host side of liblrf_hip.so that owns no kernel: error convention, the context and its workspace, image
// geometry, descriptor tables and the launch plan of a call (kernel families, their streams), the memory helpers of the C ABI
// (include/lrf_hip.h).  The kernels and their launch sequences: lrf_encode8.hip (the 64-column path), lrf_any.hip (any-shape,
// RGB colour space, svd); the host -> host pipeline: lrf_pipe.hip.
#include "lrf_host.h"

static thread_local char g_err[512] = "";

int set_err(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_err() { return g_err; }

int ensure(lrf_ctx* c, DevBuf& b, size_t bytes)
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

int upload(lrf_ctx* c, DevBuf& b, const void* src, size_t bytes)
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

void fold_events(lrf_ctx* c)
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
void plane_dims(int64_t H, int64_t W, int c, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M)
{
    // F.interpolate(scale_factor=0.5): output size = floor(input * 0.5) (lrf/compression/qmf.py:230)
    int64_t ph = c ? (int64_t)floor((double)H * 0.5) : H, pw = c ? (int64_t)floor((double)W * 0.5) : W;
    *h = ph;
    *w = pw;
    *hp = ph + (8 - ph % 8) % 8;
    *wp = pw + (8 - pw % 8) % 8;
    *M = (*hp / 8) * (*wp / 8);
}

int make_geom(int64_t H, int64_t W, ImageGeom* g)
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
void add_plane(Tables& t, long x_off, long u_off, long v_off, long u0_off, long v0_off, int M, int R, int sign_off)
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
    pd.init_src = pi; // (a sweep call's table builder points the lower-rank planes of a matrix at its largest-rank plane)
    for (int b = 0; b < pd.nblk; b++) t.blocks.push_back(BlockDesc{pi, b * LRF_KC, b, 0});
    pd.gch0 = 0; // the Gram chunks are cut when the table is complete (finish_gram_chunks)
    pd.ngch = 0;
    t.planes.push_back(pd);
}

int check_params(int64_t M, int64_t N, int R, int K, int lo, int hi)
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
int table_rmax(const Tables& t)
{
    int rmax = 1;
    for (const PlaneDesc& pd : t.planes) rmax = pd.R > rmax ? pd.R : rmax;
    return rmax;
}
int table_rp(const Tables& t) { return table_rmax(t) <= 16 ? 16 : LRF_RPB; }

// ---- kernel families of a call (struct FamRun, lrf_host.h) -----------------------------------
bool bcd_wave_variant()
{
    static const bool v = !dev_flag("LRF_BCD_WG"); // LRF_BCD_WG=1 (dev build): the workgroup kernel k_bcd instead of k_bcd_w
    return v;
}
// Since the families of a call run side by side on streams of their own (round 3) — or in one persistent launch (round 5) — the
// split pays from 1024 blocks on (64 x 512x768: (16,8,8) 0.93 -> 0.89 ms, (20,10,10) 2.04 -> 1.36 with k_bcd_w32 on the luma run);
// calls with a rank above 16 split from 256 blocks (24 images: (20,10,10) 1.27 -> 1.04 ms, 12 images 1.06 -> 0.99).
bool plan_splits(long nblocks, int rmax_t)
{
    static const bool no_split = dev_flag("LRF_NO_FAMILY_SPLIT");
    static const long env_blocks = env_long("LRF_FAMILY_SPLIT_BLOCKS", -1); // test hook (lrf_env.h)
    const long min_blocks = env_blocks >= 0 ? env_blocks : (rmax_t > 16 ? 256 : 1024);
    return !no_split && bcd_wave_variant() && rmax_t <= LRF_BIG_TO_ANY_RANK && nblocks >= min_blocks;
}
std::vector<FamRun> plan_runs(const Tables& t)
{
    const int rmax_t = table_rmax(t);
    const bool split = plan_splits((long)t.blocks.size(), rmax_t);
    std::vector<FamRun> runs;
    for (int p = 0; p < (int)t.planes.size(); p++) {
        const PlaneDesc& pd = t.planes[p];
        const int fam = split ? fam_of_rank(pd.R) : (rmax_t > 16 ? 2 : fam_of_rank(rmax_t));
        if (runs.empty() || runs.back().fam != fam) runs.push_back(FamRun{p, 0, pd.blk0, 0, 1, fam, fam == 2 ? LRF_RPB : 16, pd.R, false, 0});
        FamRun& r = runs.back();
        r.any_native = r.any_native || pd.native_t2_u != 0;
        if (pd.init_src == p && r.nbase == r.nplanes) r.nbase++; // (the table builders put a run's self-initialising planes first)
        r.nplanes++;
        r.nblocks += pd.nblk;
        r.rmax = pd.R > r.rmax ? pd.R : r.rmax;
        r.rmin = pd.R < r.rmin ? pd.R : r.rmin;
    }
    return runs;
}
bool plan_is_mixed(const std::vector<FamRun>& runs)
{
    bool p16 = false, p64 = false;
    for (const FamRun& r : runs) (r.pitch == 16 ? p16 : p64) = true;
    return p16 && p64;
}
FamBufs run_bufs(lrf_ctx* c, const FamRun& r, bool mixed)
{
    if (mixed && r.pitch == 16) return FamBufs{(float*)c->vf16.p, (float*)c->wf16.p, (float*)c->bf16.p, (float*)c->pp16.p, (float*)c->qp16.p};
    return FamBufs{(float*)c->vf.p, (float*)c->wf.p, (float*)c->bf.p, (float*)c->ppart.p, (float*)c->qpart.p};
}

hipStream_t run_stream(lrf_ctx* c, size_t run_idx) { return (c->fam_forked && run_idx > 0) ? c->fam_stream[run_idx - 1] : c->stream; }
// stage_only: the fork is joined again inside the caller's stage (run_init's initialisation kernels) — kernel profiling may stay
// on (the stage's event pair on the caller's stream encloses fork and join); a fork that lasts into run_bcd is refused while
// profiling is on (the per-launch event pairs are recorded on the caller's stream only).
int fam_fork_streams(lrf_ctx* c, size_t nruns, bool stage_only)
{
    static const bool off = dev_flag("LRF_NO_FAMILY_STREAMS");
    c->fam_forked = false;
    if (off || !c->fam_parallel || (c->profile && !stage_only) || nruns < 2 || nruns > 3) return LRF_OK;
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
int fam_join_streams(lrf_ctx* c, size_t nruns)
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
    for (size_t pi = 0; pi < t.planes.size(); pi++)
        if (t.planes[pi].init_src == (int)pi) rows += t.planes[pi].M; // (planes that share another's initialisation need no Gram matrix)
    int per = LRF_GRAM_ROWS;
    while (per > 384 && rows / per < 768) per >>= 1;
    // the fused planes (k_planes16_gram: two workgroups per CU, 512 of them a round): chunks of LRF_GRAM_ROWS_FUSED rows
    long rows_f = 0;
    for (size_t pi = 0; pi < t.planes.size(); pi++)
        if (t.planes[pi].init_src == (int)pi && t.planes[pi].gram_fused) rows_f += t.planes[pi].M;
    int per_f = LRF_GRAM_ROWS_FUSED;
    while (per_f > 384 && rows_f / per_f < 768) per_f >>= 1;
    const int per_k = per;
    t.gchunks.clear();
    for (int pass = 0; pass < 2; pass++) { // the chunks of k_gram64 first, then those the planes kernel computes (k_planes16_gram)
        per = pass ? per_f : per_k;
        for (int pi = 0; pi < (int)t.planes.size(); pi++) {
            PlaneDesc& pd = t.planes[pi];
            if ((pd.gram_fused != 0) != (pass == 1)) continue;
            pd.gch0 = (int)t.gchunks.size();
            pd.ngch = pd.init_src == pi ? (pd.M + per - 1) / per : 0;
            for (int g = 0; g < pd.ngch; g++) {
                const int row0 = g * per;
                t.gchunks.push_back(GramChunk{pi, row0, pd.gch0 + g, pd.M - row0 < per ? pd.M - row0 : per});
            }
        }
        if (pass == 0) t.ngram_rest = (int)t.gchunks.size();
    }
}

int upload_tables(lrf_ctx* c, Tables& t)
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

// The error word of k_bcd_p (no synchronisation here: the caller has waited for the stream, by whatever means).  A poll that
// expired invalidates the call it belongs to and every persistent call queued behind it on this context (those find the
// queue state marked and leave at once): the message names the first one, the state is cleared by the next launch.
int ctx_check(lrf_ctx* c)
{
    if (!c->h_perr || !*c->h_perr) return LRF_OK;
    const int first = *c->h_perr;
    *c->h_perr = 0;
    c->psync_dirty = true;
    return set_err(LRF_EHIP, "k_bcd_p: a wave's poll for a V update expired in persistent launch %d of this context (%d issued so far): "
                             "the factors of that call and of every later call up to this check are invalid", first, c->pseq);
}


// ---- C ABI ------------------------------------------------------------------------------------
extern "C" {

const char* lrf_last_error(void) { return last_err(); }


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
    {
        // k_bcd_p hands data from wave to wave inside one launch with sc1 stores / loads and agent-scope tickets: a hardware
        // path measured on gfx950 (MI355X_MICROARCH.md), not a guarantee of the programming model — other parts keep the
        // launch-per-iteration path
        hipDeviceProp_t prop;
        c->persist_arch = hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }
    c->init_sweeps = (int)env_long("LRF_DEBUG_INIT_SWEEPS", 0); // test hook (lrf_env.h)
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
    return ctx_check(c);
}

int lrf_ctx_check(lrf_ctx* c)
{
    if (!c) return set_err(LRF_EINVAL, "ctx is NULL");
    return ctx_check(c);
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

} // extern "C"
