// lrf_pack.cpp — host-side packer of liblrf_pack.so (include/lrf_pack.h).  No GPU code.
#include "../../include/lrf_pack.h"

#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

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

// encode_matrix(mode="col") of a row-major [M,R] int8 matrix
int encode_matrix(const int8_t* A, int64_t M, int R, std::string& out)
{
    std::vector<std::string> fibers((size_t)R);
    std::vector<unsigned char> col((size_t)M), buf(compressBound((uLong)M));
    for (int r = 0; r < R; r++) {
        for (int64_t m = 0; m < M; m++) col[(size_t)m] = (unsigned char)A[m * R + r];
        uLongf n = (uLongf)buf.size();
        if (compress2(buf.data(), &n, col.data(), (uLong)M, 9) != Z_OK) return -5;
        fibers[(size_t)r].assign((const char*)buf.data(), (size_t)n);
    }
    char header[96];
    int hl = snprintf(header, sizeof(header), "{\"num_fibers\": %d, \"mode\": \"col\", \"dtype\": \"int8\"}", R);
    out = combine({std::string(header, (size_t)hl), combine(fibers)});
    return 0;
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
    unsigned hw = std::thread::hardware_concurrency();
    int nt = threads > 0 ? threads : (int)(hw ? (hw > 64 ? 64 : hw) : 1);
    if (nt > B) nt = (int)B;
    std::atomic<int64_t> next(0);
    std::atomic<int> status(0);
    const std::string meta(metadata, (size_t)metadata_len);
    auto work = [&]() {
        for (;;) {
            int64_t b = next.fetch_add(1);
            if (b >= B || status.load() != 0) return;
            const int8_t* u = U + b * u_stride;
            const int8_t* v = V + b * v_stride;
            std::vector<std::string> enc(6);
            int rc = 0;
            for (int c = 0; c < 3 && rc == 0; c++) {
                rc = encode_matrix(u, M[c], R[c], enc[2 * c]);
                if (rc == 0) rc = encode_matrix(v, 64, R[c], enc[2 * c + 1]);
                u += M[c] * R[c];
                v += 64 * R[c];
            }
            if (rc) { status.store(rc); return; }
            std::string stream = combine({meta, combine(enc)});
            uint8_t* p = (uint8_t*)malloc(stream.size() ? stream.size() : 1);
            if (!p) { status.store(-4); return; }
            memcpy(p, stream.data(), stream.size());
            out[b] = p;
            out_len[b] = (int64_t)stream.size();
        }
    };
    for (int64_t b = 0; b < B; b++) out[b] = nullptr;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (status.load() != 0) {
        for (int64_t b = 0; b < B; b++) { free(out[b]); out[b] = nullptr; }
        return status.load();
    }
    return 0;
}

} // extern "C"
