// lrf_host.h — what the host-side translation units of liblrf_hip.so share: error convention, device guard, the context,
// descriptor tables and the launch plan of a call.  lrf_ctx.hip defines the functions declared here unless noted.
#ifndef LRF_HOST_H
#define LRF_HOST_H
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../include/lrf_hip.h"
#include "lrf_internal.h"
#include "lrf_env.h"

// the planes qmf_encode forms hold YCbCr samples, 0 or in [0.114, 255.5]: all below 2^8 and exact on the grid 2^(8-35)
// (run_init: selects k_gram64's integer digit extraction; callers with arbitrary X pass LRF_GRAM_EXP_FROM_DATA)
#define LRF_PLANES_GRAM_EXP 8
// largest rank of the 64-column BCD kernels (k_bcd_w <= 8, k_bcd <= 16, k_bcd_mid <= 32); above it the any-shape kernels iterate
#define LRF_BIG_TO_ANY_RANK 32
#define LRF_TABLE_SETS 6 // descriptor-table sets a context keeps resident (upload_tables)
#define LRF_BCDW_MIN_BLOCKS 1024 // smaller rank <= 8 runs iterate on the workgroup kernel k_bcd (run_bcd)
#define LRF_BCDW16_MIN_BLOCKS 1024 // likewise for rank <= 16 runs and k_bcd_w16
#define LRF_PERSIST_MIN_BLOCKS 3584 // a call of this many blocks runs its iterations in one launch (k_bcd_p) ...
#define LRF_PERSIST_MIN_BLOCKS_ONE_FAMILY 2304 // ... of this many when all its planes are of one rank family (lrf_bcd_persist.hip)
#define LRF_BCDW32_MIN_BLOCKS 128  // likewise for rank 17..32 runs and k_bcd_w32 / k_bcd_w32f (12 images: 1.06 -> 0.99 ms at (20,10,10))

int set_err(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_err();

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
    unsigned attr_persist = 0;   // likewise, one bit per instantiation of k_bcd_p (lrf_bcd_persist.hip)
    // Kernel families of one call on streams of their own (run_init / run_bcd): the runs of plan_runs touch disjoint planes, so
    // the whole chain of a run — initialisation, b table, K x (U update, V update) — is independent of the other runs'; the
    // first run stays on `stream`, the others fork behind the Gram pass and are joined at the end of run_bcd.  Created on
    // first use (a call with 1024 blocks or more — 256 with a rank above 16 — that mixes rank families); never while kernel profiling is on.
    hipStream_t fam_stream[2] = {nullptr, nullptr};
    hipEvent_t fam_fork = nullptr, fam_join[2] = {nullptr, nullptr};
    bool fam_parallel = false;   // set by the fused entry points whose run_init is followed by run_bcd at once
    bool fam_forked = false;     // run_init forked: run_bcd uses the same streams and joins
    bool init_parallel = false;  // set by the fused entry points of a call whose iterations run in k_bcd_p: only the
                                 // initialisation kernels of its families (per-matrix latency chains) run side by side
    hipEvent_t planes_done = nullptr; // set by a pipe: recorded after the planes kernel of lrf_qmf_encode_rgb_u8 (input buffer free)
    // the persistent iteration kernel (k_bcd_p, default for large rank <= 8 calls): its queue head, tickets and flags; its error
    // word is page-locked host memory the kernel writes directly — the sequence number of the first launch whose poll expired.
    // It is looked at wherever results are handed back (ctx_check: lrf_ctx_check, lrf_ctx_synchronize, lrf_pipe_wait_next) and at
    // the next persistent call's entry.
    DevBuf psync;
    int* h_perr = nullptr;     // page-locked: k_bcd_p writes its launch number here when a poll expires
    int pseq = 0;              // persistent launches issued on this context so far (the numbers start at 1)
    bool psync_dirty = false;  // the queue state of k_bcd_p is not all-zero (a failed launch)
    bool persist_arch = false; // the device is the part the in-launch hand-offs of k_bcd_p were validated on (gfx950)
};


int ensure(lrf_ctx* c, DevBuf& b, size_t bytes);
int upload(lrf_ctx* c, DevBuf& b, const void* src, size_t bytes);
void fold_events(lrf_ctx* c);
int ctx_check(lrf_ctx* c); // the error word of k_bcd_p (no synchronisation): LRF_OK, or the failure and its message

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


#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// ---- geometry (lrf_ctx.hip)
void plane_dims(int64_t H, int64_t W, int c, int64_t* h, int64_t* w, int64_t* hp, int64_t* wp, int64_t* M);
int make_geom(int64_t H, int64_t W, ImageGeom* g);

// ---- descriptor tables ------------------------------------------------------------------------
struct Tables {
    std::vector<PlaneDesc> planes;
    std::vector<BlockDesc> blocks;
    std::vector<GramChunk> gchunks; // the chunks k_gram64 computes first, then those of the gram_fused planes (k_planes16_gram)
    int ngram_rest = 0;             // how many of them k_gram64 computes (finish_gram_chunks)
};

void add_plane(Tables& t, long x_off, long u_off, long v_off, long u0_off, long v0_off, int M, int R, int sign_off);
int check_params(int64_t M, int64_t N, int R, int K, int lo, int hi);
int table_rmax(const Tables& t);
int table_rp(const Tables& t); // padded rank of the V / W / partial tables: 16 (one MFMA tile) or LRF_RPB
int upload_tables(lrf_ctx* c, Tables& t);

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
    int nbase;       // leading planes of the run that compute their own SVD initialisation (all of them, except in a sweep call:
                     // there the other planes take their columns from a plane of the same matrix, PlaneDesc::init_src)
};
inline int fam_of_rank(int R) { return LRF_FAM_OF_RANK(R); }
bool plan_splits(long nblocks, int rmax_t); // whether a call of that size gives every plane the kernel family of its own rank
bool bcd_wave_variant();
std::vector<FamRun> plan_runs(const Tables& t);
bool plan_is_mixed(const std::vector<FamRun>& runs);
// the V / W / b / partial tables a run uses
struct FamBufs {
    float *vf, *wf, *bf, *pp, *qp;
};
FamBufs run_bufs(lrf_ctx* c, const FamRun& r, bool mixed);
hipStream_t run_stream(lrf_ctx* c, size_t run_idx);
int fam_fork_streams(lrf_ctx* c, size_t nruns, bool stage_only = false);
int fam_join_streams(lrf_ctx* c, size_t nruns);

// ---- the 64-column encoder (lrf_encode8.hip)
// The part of lrf_qmf_encode_rgb_u8 that may allocate or upload: argument checks, the X workspace, the plane / block tables of
// B images (resident afterwards: upload_tables).  A pipe calls it for every sub-batch size of a submission before any
// transfer is in flight, so that nothing synchronises or allocates once its threads and streams are busy.
struct EncodePlan {
    ImageGeom g;
    Tables t;
    long u_img = 0, v_img = 0, uoff[3], voff[3], u0c[4] = {0, 0, 0, 0}, v0c[4] = {0, 0, 0, 0};
};
int encode_rgb_prepare(lrf_ctx* c, int64_t B, int64_t H, int64_t W, const int R[3], int K, int lo, int hi, bool with_sign, bool fuse_gram,
                       EncodePlan& ep);
bool planes_gram_eligible(const uint8_t* rgb, int64_t B, int64_t H, int64_t W); // lrf_planes_gram.hip
int planes_gram_from_rgb(lrf_ctx* c, const uint8_t* rgb, int64_t H, int64_t W, const ImageGeom& g, const Tables& t, float* X);

// ---- kernels of other translation units behind launch functions ---------------------------------------------------------
// one BCD half-iteration of a run (U update + partials of the V update): what every family's kernel takes
struct BcdLaunch {
    const float* X;
    const PlaneDesc* pl;
    const BlockDesc* bl; // the run's first block
    int nblocks;
    const float *vf, *wf, *bf;
    const float* U0;     // mode 2: the caller's fp32 initial U
    int8_t* U;
    float *pp, *qp;
    GsParams gp;         // exact_int set for the run
    int mode;            // 0: iterations >= 2 (old U from int8); 1: the first, old U = X W0; 2: the first, old U = U0
};
// ranks 17..32 (lrf_bcd32.hip: k_bprep_big, k_bcd_w32 / k_bcd_w32f / k_bcd_mid, k_vupdate_mid)
int bcd32_bprep(hipStream_t rs, const PlaneDesc* pl, const float* vf, float* bf, int nplanes, int plane0);
int bcd32_update_u(lrf_ctx* c, hipStream_t rs, const BcdLaunch& a, const FamRun& r, long mx_bound);
int bcd32_update_v(lrf_ctx* c, hipStream_t rs, const PlaneDesc* pl, const float* pp, const float* qp, float* vf, float* bf, int8_t* V, float lo,
                   float hi, int last, int nplanes, int plane0);
bool bcd32_wave_kernels_apply(const FamRun& r, bool exact_int, long mx_bound, int mode); // k_bcd_w32 (mode 0) / k_bcd_w32f (mode 1) take the run
// iterations 2..K of a whole call in one launch (lrf_bcd_persist.hip: k_bcd_p<F16, NP32>)
struct PersistPlan {
    bool use = false;
    bool f16 = false; // planes of ranks 9..16 occur
    int np32 = 0;     // pairs of rank columns of the planes of ranks 17..32 (0: none)
    bool first = false; // the launch may carry the call's first iteration too (ranks <= 16)
};
PersistPlan bcdp_plan(lrf_ctx* c, const std::vector<FamRun>& runs, int K, int lo, int hi);
int bcdp_launch(lrf_ctx* c, const PersistPlan& pp, const float* X, const PlaneDesc* pl, const BlockDesc* bl, int nblocks, int nplanes, int plane0,
                const FamBufs& t16, const FamBufs& t64, int8_t* U, int8_t* V, GsParams gp, int niter, bool first);

// ---- the any-shape path (lrf_any.hip: other patch sizes, patch=False, the RGB colour space, ranks 33..64 of the 64-column path)
int any_workspace(lrf_ctx* c, int B, int M, int N, int R);
int any_run_bcd_ex(lrf_ctx* c, const float* X, long x_batch, int B, int M, int N, int R, int K, int lo, int hi, int8_t* U, long u_batch,
                   int8_t* V, long v_batch);
int any_decompose(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, int K, int lo, int hi, const int8_t* sign, int8_t* U,
                  int8_t* V);
int any_bcd(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, int K, int lo, int hi, const float* U0, const float* V0,
            int8_t* U, int8_t* V);
int any_svd_init(lrf_ctx* c, const float* X, int64_t B, int64_t M, int64_t N, int R, const int8_t* sign, float* U0, float* V0);
#endif
