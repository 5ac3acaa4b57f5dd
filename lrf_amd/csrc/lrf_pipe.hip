// lrf_pipe.hip — lrf_pipe: the host -> host pipelined encoder of the C ABI (include/lrf_hip.h).
// Host code only: it sequences uploads, lrf_qmf_encode_rgb_u8 (lrf_encode8.hip) and downloads over streams.
#include <condition_variable>
#include <mutex>
#include <thread>

#include "lrf_host.h"

extern "C" {

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
    std::vector<size_t> slot;       // the slot each enqueued sub-batch ran on (its context holds k_bcd_p's error word)
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
    if (e == hipSuccess && !dev_flag("LRF_PIPE_NO_D2H")) e = hipStreamCreateWithFlags(&p->d2h, hipStreamNonBlocking);
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
    p->slot.clear();
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
                if ((rc = encode_rgb_prepare(s.ctx, nb, H, W, R, K, lo, hi, sign_host != nullptr, planes_gram_eligible((const uint8_t*)s.rgb.p, nb, H, W), ep)))
                    return rc;
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
        p->slot.push_back(slot_of[i]);
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
    lrf_ctx* sc = p->slots[p->slot[p->next_wait]].ctx;
    p->next_wait++;
    // the piece's kernels have finished: did its persistent launch (k_bcd_p) report an expired poll?  The piece counts as
    // waited for either way (the caller's drain loop goes on); its factors in the caller's buffers are invalid.
    return ctx_check(sc);
}

int lrf_pipe_qmf_encode_rgb_u8_host(lrf_pipe* p, const uint8_t* rgb_host, int64_t B, int64_t H, int64_t W, const int R[3], int K,
                                    int lo, int hi, const int8_t* sign_host, int8_t* U_host, int8_t* V_host)
{
    int rc = lrf_pipe_qmf_encode_submit(p, rgb_host, B, H, W, R, K, lo, hi, sign_host, U_host, V_host, nullptr);
    // whatever was enqueued must have finished before the caller may touch (or free) its buffers, also after an error
    int64_t n = 1;
    int rc_wait = LRF_OK;
    char msg_wait[512] = "";
    while (n > 0) {
        const size_t before = p ? p->next_wait : 0;
        int rc2 = lrf_pipe_wait_next(p, nullptr, &n);
        if (rc2 && !rc_wait) { // keep the first failure, go on draining: nothing may stay in flight into the caller's buffers
            rc_wait = rc2;
            snprintf(msg_wait, sizeof(msg_wait), "%s", last_err());
        }
        if (rc2 && (!p || p->next_wait == before)) break; // no progress (a HIP error): give up
    }
    if (!rc && rc_wait) return set_err(rc_wait, "%s", msg_wait);
    if (rc && p) {
        DevGuard dev_guard_(p->device);
        (void)hipStreamSynchronize(p->h2d);
        for (auto& s : p->slots) (void)hipStreamSynchronize(s.ctx->stream);
    }
    return rc;
}

} // extern "C"
