// lrf_pack.cpp — host-side packer of liblrf_pack.so (include/lrf_pack.h).  No GPU code.
#include "../../include/lrf_pack.h"

#include <zlib.h>

#include <pthread.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// Worker threads that outlive the call: one image per call is the reference's own usage pattern, and creating and joining
// fifteen threads per call cost more than compressing its sixteen columns (0.55 -> ~0.2 ms per 512x768 image).  The threads
// are detached and sleep on a condition variable between jobs.  A forked child has none of these threads, and the pool's
// mutexes and condition variables were copied in whatever state the parent's sleeping workers had left them (a condition
// variable still counts them as waiters: the child's second notify_all would wait for them for ever), so the child never
// touches the inherited pool: worker_pool() below hands it a brand-new one (pthread_atfork child handler).
class WorkerPool {
public:
    // runs job() on `helpers` pool threads and on the calling thread; returns when all of them have returned
    void run(int helpers, const std::function<void()>& job)
    {
        std::unique_lock<std::mutex> call(call_mutex_); // one job at a time
        {
            std::unique_lock<std::mutex> lk(m_);
            if (pid_ < 0) pid_ = getpid(); // first use (a forked child gets a new pool, never this one)
            while (nthreads_ < helpers) {
                std::thread(&WorkerPool::loop, this, pid_).detach();
                nthreads_++;
            }
            job_ = &job;
            slots_ = helpers;
            running_ = helpers;
            generation_++;
        }
        work_cv_.notify_all();
        job();
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return running_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(pid_t owner)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void()>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                work_cv_.wait(lk, [&] { return pid_ != owner || (generation_ != seen && slots_ > 0); });
                if (pid_ != owner) return; // cannot happen in the process that created it; kept for symmetry
                seen = generation_;
                slots_--;
                job = job_;
            }
            (*job)();
            {
                std::unique_lock<std::mutex> lk(m_);
                if (--running_ == 0) done_cv_.notify_all();
            }
        }
    }
    std::mutex call_mutex_, m_;
    std::condition_variable work_cv_, done_cv_;
    const std::function<void()>* job_ = nullptr;
    unsigned long long generation_ = 0;
    int nthreads_ = 0, slots_ = 0, running_ = 0;
    pid_t pid_ = -1;
};

std::atomic<WorkerPool*> g_pool{nullptr};

// runs in the child of a fork(): forget the parent's pool (leaked on purpose: its synchronisation objects are unusable,
// see above); the next call allocates a fresh one with mutexes and condition variables of its own
void forget_pool_after_fork() { g_pool.store(nullptr, std::memory_order_release); }

WorkerPool& worker_pool()
{
    static const int registered = pthread_atfork(nullptr, nullptr, forget_pool_after_fork);
    (void)registered;
    WorkerPool* p = g_pool.load(std::memory_order_acquire);
    if (!p) {
        WorkerPool* fresh = new WorkerPool(); // never destroyed: its threads may be asleep on it when the process exits
        if (g_pool.compare_exchange_strong(p, fresh, std::memory_order_acq_rel))
            p = fresh;
        else
            delete fresh; // another thread was first; `fresh` has started no thread yet
    }
    return *p;
}

void append_be32(std::string& s, size_t n)
{
    char b[4] = {(char)((n >> 24) & 0xff), (char)((n >> 16) & 0xff), (char)((n >> 8) & 0xff), (char)(n & 0xff)};
    s.append(b, 4);
}

// combine_bytes: reduce(_combine_bytes, payloads) with _combine(p1, p2) = len32_be(p1) || p1 || p2
std::string combine(const std::vector<std::string>& parts)
{
    std::string acc = parts[0];
    for (size_t i = 1; i < parts.size(); i++) {
        std::string nxt;
        nxt.reserve(acc.size() + parts[i].size() + 4);
        append_be32(nxt, acc.size());
        nxt += acc;
        nxt += parts[i];
        acc.swap(nxt);
    }
    return acc;
}

// one column of a row-major [M,R] int8 matrix, zlib level 9 (encode_matrix, mode "col": utils.py:372-378)
int compress_column(const int8_t* A, int64_t M, int R, int r, std::string& out)
{
    std::vector<unsigned char> col((size_t)M), buf(compressBound((uLong)M));
    for (int64_t m = 0; m < M; m++) col[(size_t)m] = (unsigned char)A[m * R + r];
    uLongf n = (uLongf)buf.size();
    if (compress2(buf.data(), &n, col.data(), (uLong)M, 9) != Z_OK) return -5;
    out.assign((const char*)buf.data(), (size_t)n);
    return 0;
}

// header + fibers of one matrix (utils.py:380-390)
std::string assemble_matrix(const std::vector<std::string>& fibers)
{
    char header[96];
    int hl = snprintf(header, sizeof(header), "{\"num_fibers\": %d, \"mode\": \"col\", \"dtype\": \"int8\"}", (int)fibers.size());
    return combine({std::string(header, (size_t)hl), combine(fibers)});
}

} // namespace

extern "C" {

const char* lrf_pack_zlib_version(void) { return zlibVersion(); }

void lrf_pack_free(uint8_t* p) { free(p); }

int lrf_pack_qmf_streams(const int8_t* U, int64_t u_stride, const int8_t* V, int64_t v_stride, int64_t B, const int64_t M[3],
                         const int R[3], const char* metadata, int64_t metadata_len, int threads, uint8_t** out, int64_t* out_len)
{
    if (!U || !V || !M || !R || !metadata || !out || !out_len || B < 1) return -1;
    for (int c = 0; c < 3; c++)
        if (M[c] < 1 || R[c] < 1) return -1;
    // work item = one column of one factor of one image: a single image spreads over the threads as well as a batch does
    int64_t cols_per_image = 0;
    for (int c = 0; c < 3; c++) cols_per_image += 2 * (int64_t)R[c];
    const int64_t items = B * cols_per_image;
    unsigned hw = std::thread::hardware_concurrency();
    int nt = threads > 0 ? threads : (int)(hw ? (hw > 64 ? 64 : hw) : 1);
    if (nt > items) nt = (int)items;
    std::vector<std::string> fibers((size_t)items);
    std::atomic<int64_t> next(0);
    std::atomic<int> status(0);
    auto work = [&]() {
        for (;;) {
            int64_t it = next.fetch_add(1);
            if (it >= items || status.load() != 0) return;
            const int64_t b = it / cols_per_image;
            int64_t k = it - b * cols_per_image;
            const int8_t* u = U + b * u_stride;
            const int8_t* v = V + b * v_stride;
            int rc = 0;
            for (int c = 0; c < 3; c++) {
                if (k < R[c]) { rc = compress_column(u, M[c], R[c], (int)k, fibers[(size_t)it]); break; }
                k -= R[c];
                if (k < R[c]) { rc = compress_column(v, 64, R[c], (int)k, fibers[(size_t)it]); break; }
                k -= R[c];
                u += M[c] * R[c];
                v += 64 * R[c];
            }
            if (rc) { status.store(rc); return; }
        }
    };
    for (int64_t b = 0; b < B; b++) out[b] = nullptr;
    if (nt > 1) worker_pool().run(nt - 1, work);
    else work();
    if (status.load() != 0) return status.load();
    const std::string meta(metadata, (size_t)metadata_len);
    for (int64_t b = 0; b < B; b++) {
        std::vector<std::string> enc;
        size_t at = (size_t)(b * cols_per_image);
        for (int c = 0; c < 3; c++)
            for (int f = 0; f < 2; f++) {
                enc.push_back(assemble_matrix(std::vector<std::string>(fibers.begin() + at, fibers.begin() + at + R[c])));
                at += (size_t)R[c];
            }
        std::string stream = combine({meta, combine(enc)});
        uint8_t* p = (uint8_t*)malloc(stream.size() ? stream.size() : 1);
        if (!p) {
            for (int64_t j = 0; j < b; j++) { free(out[j]); out[j] = nullptr; }
            return -4;
        }
        memcpy(p, stream.data(), stream.size());
        out[b] = p;
        out_len[b] = (int64_t)stream.size();
    }
    return 0;
}

/* The streams of the other patch sizes and of patch=False (lrf/compression/qmf.py:232-286): six factor arrays per batch,
 * F[f] = [B][rows[f]][cols[f]] int8 (u_Y, v_Y, u_Cb, v_Cb, u_Cr, v_Cr).  whole == 0: every factor as encode_matrix does
 * (per-column zlib, utils.py:354-390); whole != 0: the patch=False form — the reference keeps the plane's channel axis, so the
 * factors are 3-D [1, rows, cols] and encode_tensor compresses each in one piece behind {"shape": [...], "dtype": "int8"}
 * (utils.py:429-455). */
int lrf_pack_qmf_streams_planes(const int8_t* const F[6], const int64_t rows[6], const int cols[6], int64_t B, int whole,
                                const char* metadata, int64_t metadata_len, int threads, uint8_t** out, int64_t* out_len)
{
    if (!F || !rows || !cols || !metadata || !out || !out_len || B < 1) return -1;
    for (int f = 0; f < 6; f++)
        if (!F[f] || rows[f] < 1 || cols[f] < 1) return -1;
    int64_t per_image = 0; // work items per image: columns, or whole factors
    for (int f = 0; f < 6; f++) per_image += whole ? 1 : cols[f];
    const int64_t items = B * per_image;
    unsigned hw = std::thread::hardware_concurrency();
    int nt = threads > 0 ? threads : (int)(hw ? (hw > 64 ? 64 : hw) : 1);
    if (nt > items) nt = (int)items;
    std::vector<std::string> pieces((size_t)items);
    std::atomic<int64_t> next(0);
    std::atomic<int> status(0);
    auto work = [&]() {
        for (;;) {
            int64_t it = next.fetch_add(1);
            if (it >= items || status.load() != 0) return;
            const int64_t b = it / per_image;
            int64_t k = it - b * per_image;
            int rc = 0;
            for (int f = 0; f < 6; f++) {
                const int64_t n = whole ? 1 : cols[f];
                if (k < n) {
                    const int8_t* A = F[f] + b * rows[f] * cols[f];
                    if (whole) {
                        const uLong len = (uLong)(rows[f] * cols[f]);
                        std::vector<unsigned char> buf(compressBound(len));
                        uLongf m = (uLongf)buf.size();
                        if (compress2(buf.data(), &m, (const unsigned char*)A, len, 9) != Z_OK) rc = -5;
                        else pieces[(size_t)it].assign((const char*)buf.data(), (size_t)m);
                    } else {
                        rc = compress_column(A, rows[f], cols[f], (int)k, pieces[(size_t)it]);
                    }
                    break;
                }
                k -= n;
            }
            if (rc) { status.store(rc); return; }
        }
    };
    for (int64_t b = 0; b < B; b++) out[b] = nullptr;
    if (nt > 1) worker_pool().run(nt - 1, work);
    else work();
    if (status.load() != 0) return status.load();
    const std::string meta(metadata, (size_t)metadata_len);
    for (int64_t b = 0; b < B; b++) {
        std::vector<std::string> enc;
        size_t at = (size_t)(b * per_image);
        for (int f = 0; f < 6; f++) {
            if (whole) {
                char header[128];
                int hl = snprintf(header, sizeof(header), "{\"shape\": [1, %lld, %d], \"dtype\": \"int8\"}", (long long)rows[f], cols[f]);
                enc.push_back(combine({std::string(header, (size_t)hl), pieces[at]}));
                at += 1;
            } else {
                enc.push_back(assemble_matrix(std::vector<std::string>(pieces.begin() + at, pieces.begin() + at + cols[f])));
                at += (size_t)cols[f];
            }
        }
        std::string stream = combine({meta, combine(enc)});
        uint8_t* p = (uint8_t*)malloc(stream.size() ? stream.size() : 1);
        if (!p) {
            for (int64_t j = 0; j < b; j++) { free(out[j]); out[j] = nullptr; }
            return -4;
        }
        memcpy(p, stream.data(), stream.size());
        out[b] = p;
        out_len[b] = (int64_t)stream.size();
    }
    return 0;
}

/* ---- the reverse: streams -> factor matrices (decode_matrix, utils.py:393-426) ---- */

namespace {
struct Span {
    const uint8_t* p;
    size_t n;
};
// _split2 (utils.py:268-287): len32_be(first) || first || second
bool split2(Span in, Span& first, Span& second)
{
    if (in.n < 4) return false;
    const size_t n = ((size_t)in.p[0] << 24) | ((size_t)in.p[1] << 16) | ((size_t)in.p[2] << 8) | (size_t)in.p[3];
    if (n > in.n - 4) return false;
    first = Span{in.p + 4, n};
    second = Span{in.p + 4 + n, in.n - 4 - n};
    return true;
}
// separate_bytes(combined, k): the payloads are peeled off the END of the left fold
bool separate(Span in, int k, std::vector<Span>& out)
{
    out.assign((size_t)k, Span{nullptr, 0});
    Span head = in;
    for (int i = k - 1; i >= 1; i--) {
        Span h, t;
        if (!split2(head, h, t)) return false;
        out[(size_t)i] = t;
        head = h;
    }
    out[0] = head;
    return true;
}
// the header json.dumps writes for an int8 matrix packed column by column, and nothing else
bool header_is(Span h, int num_fibers)
{
    char want[96];
    const int n = snprintf(want, sizeof(want), "{\"num_fibers\": %d, \"mode\": \"col\", \"dtype\": \"int8\"}", num_fibers);
    return n > 0 && (size_t)n == h.n && memcmp(want, h.p, h.n) == 0;
}
} // namespace

int lrf_pack_unpack_qmf_factors(const uint8_t* const* factor_blobs, const int64_t* blob_len, int64_t B, const int64_t M[3],
                                const int R[3], int threads, int8_t* U, int64_t u_stride, int8_t* V, int64_t v_stride)
{
    if (!factor_blobs || !blob_len || !M || !R || !U || !V || B < 1) return -1;
    int64_t usz = 0, vsz = 0;
    for (int c = 0; c < 3; c++) {
        if (M[c] < 1 || R[c] < 1) return -1;
        usz += M[c] * R[c];
        vsz += 64 * (int64_t)R[c];
    }
    if (u_stride < usz || v_stride < vsz) return -1;
    // pass 1 (serial, cheap): locate every compressed column; anything that is not exactly the layout described above is
    // refused (-6) and left to the caller's own parser and its error messages
    struct Item {
        Span z;
        int8_t* dst; // first element of the column in the row-major destination
        int64_t rows;
        int cols;
    };
    std::vector<Item> items;
    std::vector<Span> mats, parts, fibers;
    for (int64_t b = 0; b < B; b++) {
        if (!factor_blobs[b] || blob_len[b] < 0) return -1;
        if (!separate(Span{factor_blobs[b], (size_t)blob_len[b]}, 6, mats)) return -6;
        int8_t* u = U + b * u_stride;
        int8_t* v = V + b * v_stride;
        for (int c = 0; c < 3; c++) {
            for (int f = 0; f < 2; f++) {
                const int64_t rows = f ? 64 : M[c];
                int8_t* dst = f ? v : u;
                if (!separate(mats[(size_t)(2 * c + f)], 2, parts) || !header_is(parts[0], R[c])) return -6;
                if (!separate(parts[1], R[c], fibers)) return -6;
                for (int r = 0; r < R[c]; r++) items.push_back(Item{fibers[(size_t)r], dst + r, rows, R[c]});
            }
            u += M[c] * R[c];
            v += 64 * R[c];
        }
    }
    std::atomic<int64_t> next(0);
    std::atomic<int> status(0);
    const int64_t nitems = (int64_t)items.size();
    auto work = [&]() {
        std::vector<unsigned char> col;
        for (;;) {
            const int64_t it = next.fetch_add(1);
            if (it >= nitems || status.load() != 0) return;
            const Item& w = items[(size_t)it];
            col.resize((size_t)w.rows);
            uLongf n = (uLongf)w.rows;
            if (uncompress(col.data(), &n, w.z.p, (uLong)w.z.n) != Z_OK || (int64_t)n != w.rows) { status.store(-6); return; }
            for (int64_t i = 0; i < w.rows; i++) w.dst[i * w.cols] = (int8_t)col[(size_t)i];
        }
    };
    unsigned hw = std::thread::hardware_concurrency();
    int nt = threads > 0 ? threads : (int)(hw ? (hw > 64 ? 64 : hw) : 1);
    if (nt > nitems) nt = (int)nitems;
    if (nt > 1) worker_pool().run(nt - 1, work);
    else work();
    return status.load();
}

} // extern "C"
