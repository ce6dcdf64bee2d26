// Host engine + C ABI of the MI355X bundle-adjustment library (see include/ba_hip.h).
//
// One BaHandle = one problem resident in HBM: SoA observations in the engine's point-sorted
// order, chunk tables, the block-sparse reduced camera system, PCG vectors and the LM state.
// All kernels run on the handle's own HIP stream; the host only reads back one small LmState
// per LM step (accept flag / termination) -- the damping and accept logic itself runs on device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"
#include "ba_banded.h"
#include "ba_bcr.h"
#include "ba_dense.h"
#include "ba_dense_big.h"
#include "ba_kernels.h"
#include "ba_heavy.h"
#include "ba_det.h"
#include "ba_rows.h"

// ---- scratch of the no-allocation entry point (ba_eval_cost_noalloc, end of this file) ------------------------------
// A file-scope device array instead of hipMalloc'd buffers: every GPU the library runs on holds its own copy (about
// 4.4 MB) until the process ends.  In host code the NAME below is only a stand-in, not a device address: the address
// on the current device comes from hipGetSymbolAddress (noalloc_scratch), and only that pointer is ever handed to a
// kernel, a memset or a copy.
#define BA_NOALLOC_MAX_CAMS 16384
#define BA_NOALLOC_MAX_OBS (1LL << 24)
#define BA_NOALLOC_MAX_BLOCKS ((int)(BA_NOALLOC_MAX_OBS / 256))
#define BA_SCR_AUX 0                                                   /* [MAX_CAMS * BA_CAM_AUX] per-camera constants */
#define BA_SCR_BC (BA_SCR_AUX + (size_t)BA_NOALLOC_MAX_CAMS * BA_CAM_AUX) /* [MAX_BLOCKS] per-block cost partials */
#define BA_SCR_BR (BA_SCR_BC + BA_NOALLOC_MAX_BLOCKS)                  /* [MAX_BLOCKS] per-block residual sums */
#define BA_SCR_FLAG (BA_SCR_BR + BA_NOALLOC_MAX_BLOCKS)                /* one int: an index was out of range */
#define BA_SCR_DOUBLES (BA_SCR_FLAG + 32)
static __device__ __align__(256) double scratch[BA_SCR_DOUBLES];

namespace {

thread_local std::string g_err;
thread_local int g_code = 0;

int fail(int code, const char *fmt, ...) {
    g_code = code;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return fail(BA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define BA_MAX_WORLD 64  // ranks of one sharded solve

// One-time uploads of small problems are bound by the call, not by the bytes: a blocking hipMemcpy from pageable memory
// costs 20-35 us, and a handle is created with some thirty of them.  While a StageScope is open on the calling thread,
// DevBuf::upload / refill copy their source into a process-wide page-locked staging buffer instead and enqueue a truly
// asynchronous copy on the handle's stream; the scope's end synchronises the stream once (only then is the buffer handed
// out again) and lets the buffer grow -- up to STAGE_MAX -- to what did not fit.  Large problems overflow it and take the
// blocking path for the rest, where the call overhead does not matter.
struct Stager {
    std::mutex mu;
    char *buf = nullptr;
    size_t cap = 0, used = 0, missed = 0;
    hipStream_t st = nullptr;   // the open scope's stream
};
constexpr size_t STAGE_MAX = (size_t)64 << 20, STAGE_FIRST = (size_t)8 << 20;   // (8 MB: the reference's own problem sizes)
Stager g_stager;
thread_local Stager *t_stage = nullptr;
void *stage_take(size_t bytes) {
    Stager *s = t_stage;
    if (!s) return nullptr;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (s->used + need > s->cap) {
        s->missed += need;
        return nullptr;
    }
    void *r = s->buf + s->used;
    s->used += need;
    return r;
}
struct StageScope {
    hipStream_t st;
    bool on = false;
    explicit StageScope(hipStream_t stream) : st(stream) {
        if (t_stage || getenv("BA_NO_STAGING")) return;   // (nested: the outer scope owns the buffer)
        if (!g_stager.mu.try_lock()) return;   // (another thread is staging: this one takes the blocking copies)
        g_stager.used = g_stager.missed = 0;
        if (!g_stager.buf && hipHostMalloc((void **)&g_stager.buf, STAGE_FIRST, hipHostMallocPortable) == hipSuccess) g_stager.cap = STAGE_FIRST;
        g_stager.st = stream;
        t_stage = &g_stager;
        on = true;
    }
    ~StageScope() {
        if (!on) return;
        (void)hipStreamSynchronize(st);   // every staged copy has left the buffer
        Stager &g = g_stager;
        if (g.missed > 0 && g.cap < STAGE_MAX) {
            size_t want = g.used + g.missed + ((size_t)1 << 20);   // (geometric: a growing reconstruction re-allocates rarely)
            if (want < 2 * g.cap) want = 2 * g.cap;
            if (want > STAGE_MAX) want = STAGE_MAX;
            if (g.buf) (void)hipHostFree(g.buf);
            g.buf = nullptr;
            g.cap = 0;
            if (hipHostMalloc((void **)&g.buf, want, hipHostMallocPortable) == hipSuccess) g.cap = want;
            else g.buf = nullptr;
        }
        t_stage = nullptr;
        g.mu.unlock();
    }
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
    }
    // (contents are NOT kept; an allocation that is already large enough is re-used -- ba_append / ba_set_values size their
    // work buffers again and again, and a hipFree + hipMalloc pair costs tens of microseconds)
    hipError_t alloc(size_t count) {
        if (p && count <= cap && count > 0) {
            n = count;
            return hipSuccess;
        }
        release();
        n = count;
        if (count == 0) count = 1;
        cap = count;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    // the same with head room for a structure that is expected to grow (ba_append)
    hipError_t alloc_grow(size_t count) {
        if (p && count <= cap && count > 0) {
            n = count;
            return hipSuccess;
        }
        release();
        n = count;
        cap = count + count / 4 + 16;
        return hipMalloc((void **)&p, cap * sizeof(T));
    }
    // ba_append: new elements behind the first `keep` ones, which stay where they are in HBM.  The allocation grows
    // geometrically (device-to-device copy of the kept part) so that a sequence of appends re-allocates rarely.
    hipError_t append_at(size_t keep, const T *tail, size_t count) {
        if (keep > n) keep = n;
        const size_t need = keep + count;
        if (!p || need > cap) {
            const size_t ncap = need + need / 2 + 64;
            T *q = nullptr;
            hipError_t e = hipMalloc((void **)&q, ncap * sizeof(T));
            if (e != hipSuccess) return e;
            if (p && keep) e = hipMemcpy(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice);
            if (p) (void)hipFree(p);
            p = q;
            cap = ncap;
            if (e != hipSuccess) return e;
        }
        n = need;
        if (count == 0) return hipSuccess;
        if (void *stg = stage_take(count * sizeof(T))) {
            memcpy(stg, tail, count * sizeof(T));
            return hipMemcpyAsync(p + keep, stg, count * sizeof(T), hipMemcpyHostToDevice, t_stage->st);
        }
        return hipMemcpy(p + keep, tail, count * sizeof(T), hipMemcpyHostToDevice);
    }
    // (a BLOCKING copy: the sources of these one-time uploads are often temporaries that die before the stream is
    // synchronised, and HIP does not promise that a pageable-memory hipMemcpyAsync has read its source on return)
    // -- unless a StageScope is open: then through the page-locked staging buffer, asynchronously on st
    hipError_t upload(const std::vector<T> &v, hipStream_t st) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        if (void *stg = stage_take(v.size() * sizeof(T))) {
            memcpy(stg, v.data(), v.size() * sizeof(T));
            return hipMemcpyAsync(p, stg, v.size() * sizeof(T), hipMemcpyHostToDevice, st);
        }
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    // same as upload, but keeps the allocation when the size has not changed (ba_set_values)
    hipError_t refill(const std::vector<T> &v, hipStream_t s) {
        if (!p || n != v.size()) return upload(v, s);
        if (v.empty()) return hipSuccess;
        if (void *stg = stage_take(v.size() * sizeof(T))) {
            memcpy(stg, v.data(), v.size() * sizeof(T));
            return hipMemcpyAsync(p, stg, v.size() * sizeof(T), hipMemcpyHostToDevice, s);
        }
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int validate(const BaProblem *p) {
    if (!p) return fail(BA_ERR_INVALID, "null problem");
    if (p->n_cams < 0 || p->n_points < 0 || p->n_obs < 0) return fail(BA_ERR_INVALID, "negative size");
    if (p->n_cams > 0 && (!p->cam_pose || !p->cam_intr || !p->cam_fixed)) return fail(BA_ERR_INVALID, "null camera array");
    if (p->n_points > 0 && !p->points) return fail(BA_ERR_INVALID, "null points array");
    if (p->n_obs > 0 && (!p->obs_cam || !p->obs_pt || !p->obs_xy)) return fail(BA_ERR_INVALID, "null observation array");
    if (p->memory != BA_MEM_HOST && p->memory != BA_MEM_DEVICE) return fail(BA_ERR_INVALID, "bad memory kind");
    if (p->n_obs > 2000000000LL) return fail(BA_ERR_UNSUPPORTED, "more than 2e9 observations per handle");
    return BA_OK;
}

// host copies of the caller's arrays (the symbolic analysis and the re-ordering run on the host)
struct HostProblem {
    std::vector<double> cam_pose, cam_intr, points, obs_xy;
    std::vector<uint8_t> cam_fixed;
    std::vector<int32_t> obs_cam, obs_pt;
};

template <typename T>
int fetch(std::vector<T> &dst, const T *src, size_t n, int memory) {
    dst.resize(n);
    if (n == 0) return BA_OK;
    if (memory == BA_MEM_HOST) memcpy(dst.data(), src, n * sizeof(T));
    else HIPCHK(hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    return BA_OK;
}

int fetch_problem(const BaProblem *p, HostProblem &h, bool indices_only) {
    int e;
    if ((e = fetch(h.cam_fixed, p->cam_fixed, (size_t)p->n_cams, p->memory))) return e;
    if ((e = fetch(h.obs_cam, p->obs_cam, (size_t)p->n_obs, p->memory))) return e;
    if ((e = fetch(h.obs_pt, p->obs_pt, (size_t)p->n_obs, p->memory))) return e;
    if (indices_only) return BA_OK;
    if ((e = fetch(h.cam_pose, (const double *)p->cam_pose, (size_t)p->n_cams * 6, p->memory))) return e;
    if ((e = fetch(h.cam_intr, p->cam_intr, (size_t)p->n_cams * 4, p->memory))) return e;
    if ((e = fetch(h.points, (const double *)p->points, (size_t)p->n_points * 3, p->memory))) return e;
    if ((e = fetch(h.obs_xy, p->obs_xy, (size_t)p->n_obs * 2, p->memory))) return e;
    return BA_OK;
}

// ---- RCCL, bound at run time (no link-time dependency: the library also loads on machines without it) -------------
// Only the five entry points the sharded solve needs; types as in rccl.h (ncclUniqueId = 128 opaque bytes,
// ncclDouble = 8, ncclSum = 0, ncclMax = 2).
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy that is already in the process (torch ships its own librccl.so) first: two RCCL instances in one
        // process would each bring up their own transports
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *lib = nullptr;
        for (const char *n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (size_t i = 0; !lib && i < sizeof names / sizeof names[0]; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(lib, "ncclBroadcast"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Broadcast;
    });
    return api.ok ? &api : nullptr;
}

}  // namespace

using namespace ba;

struct BaHandle {
    BaOptions opt{};
    Symbolic sym;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    int n_chunks = 0, n_heavy = 0, n_upper = 0, n_full = 0;
    double lambda_host = 0.0;
    bool pattern_uploaded = false;
    double x2_const = 0.0;  // |x|^2 of the unobserved points (never change)
    // state (two copies: current / trial)
    DevBuf<double> pose[2], aux[2], points[2];
    int cur = 0;
    DevBuf<double> intr;
    DevBuf<int> cam_free, free_cam;
    // observations, chunks
    DevBuf<int> o_cam, o_pt, ck_obs, ck_pt, ck_cam, ck_cam_list, ck_pair, ck_pair_slot;
    DevBuf<short> o_lcam;
    DevBuf<double> o_xy;
    DevBuf<long long> ck_pe;
    DevBuf<int> ck_pgrp;
    // heavy points (ba_heavy.h)
    DevBuf<long long> hv_obs, hv_pair;
    DevBuf<int> hv_pair_slot;
    DevBuf<double> hv_t;
    DevBuf<unsigned> pair_entries;
    // reduced system: one contiguous buffer [S | U | gc | gred | sums(4)]
    DevBuf<double> red;
    double *S = nullptr, *U = nullptr, *gc = nullptr, *gred = nullptr, *sums = nullptr;
    DevBuf<int> diag_slot, bsr_row, bsr_col, bsr_src;
    DevBuf<double> Minv, cam_gmax, gm2;
    bool minv_valid = false;   // Minv belongs to the S of the latest build
    DevBuf<double> pt_linv, pt_u, ck_cost, ck_gmax, ck_part, cam_part, step_sums, cam_sums;
    DevBuf<double> dbg_v, dbg_gp;
    DevBuf<unsigned long long> stamps;  // diagnostic (ba_debug_stamps)
    DevBuf<double> px, pr, pz, pp, pq, pqv;  // PCG: x, r, z, p (double-buffered: pp / pq), q = S p
    DevBuf<PcgScalars> pcg_sc;
    DevBuf<double> pcg_part;         // per-workgroup partial sums of the PCG's dot products (PcgPartials)
    // deterministic build (ba_det.h): contribution tables and slabs
    DetTables det;
    DevBuf<long long> det_ck_lg;
    DevBuf<int> det_c_ptr, det_c_idx, det_s_ptr;
    DevBuf<unsigned> det_s_idx;
    DevBuf<double> det_cam_slab, det_blk_slab;
    // rows mode (ba_rows.h): per-observation rows, global entry lists
    DevBuf<double> rows_buf;
    DevBuf<int> rw_ent, rw_xblk, rw_diag, rw_slot;
    DevBuf<long long> rw_ptr;
    int rw_nx = 0;
    long long rw_rows = 0, rw_row_off = 0;   // rows held by rows_buf, observation index of the first of them
    DevBuf<double> dense_a;          // dense reduced system (one-workgroup solver, or the blocked multi-workgroup one)
    DevBuf<double> dense_linv;       // blocked solver: inverse factors of the diagonal blocks
    DevBuf<double> band_l, band_y, band_g, band_mid;  // block-banded factor, work vectors, separator system
    BcrPlan bcr_plan;                                  // block cyclic reduction (ba_bcr.h): schedule of this pattern,
    int bcr_state = 0;                                 //   0 not planned yet, 1 usable, -1 the pattern is no chain / ring
    DevBuf<int32_t> bcr_tab;                           //   elimination / update / block tables
    DevBuf<double> bcr_dg, bcr_bv, bcr_linv;           //   diagonal blocks + coupling slots, block vectors, tile inverses
    size_t bcr_off[6] = {0, 0, 0, 0, 0, 0};            //   offsets of the tables inside bcr_tab
    DevBuf<long long> upper_keys;
    DevBuf<int> dense_info;
    int dense_fallbacks = 0;
    // AUTO between the blocked dense Cholesky and block-Jacobi PCG on a (nearly) full pattern: both are timed once, then
    // the faster one is kept (do_step)
    int auto_phase = -1;           // -1: first use of the direct solver (allocations, attributes: not timed), 0: time the
                                   // direct solver next, 1: time PCG next, 2: decided
    bool auto_use_pcg = false;
    float auto_ms_direct = 0.f, auto_ms_pcg = 0.f;
    int auto_pcg_iters = 0;
    bool dense_attr_set = false;
    DevBuf<LmState> state;
    LmState hstate{};
    LmState *hstate_pinned = nullptr;   // page-locked landing place of read_state
    PcgScalars hpcg{};
    PcgScalars *hpcg_pinned = nullptr;   // page-locked landing place of the PCG's convergence poll
    int *dense_info_pinned = nullptr;    // ... and of the pivot flag of the preconditioner's factorisation (do_bcr_pcg)
    // collectives
    BaCollectiveFn coll = nullptr;
    void *coll_user = nullptr;
    int rank = 0, world = 1;
    bool coll_stream_ordered = false;  // the callback enqueues on `stream`; no host synchronisation around it
    void *rccl_comm = nullptr;         // ncclComm_t of the built-in RCCL collectives (ba_rccl_init); owned by the handle
    // bookkeeping
    int total_pcg = 0;
    double initial_cost = NAN;
    bool built = false;
    bool poisoned = false;   // a ba_append failed half way: the host structure and the device arrays may disagree
    bool pattern_stale = false;   // sharded handle after ba_append: the ranks must agree on a new pattern (ba_set_pattern) first
    double t_setup_ms = 0.0;

    ~BaHandle() {
        if (rccl_comm) {
            if (RcclApi *api = rccl_api()) (void)api->CommDestroy(rccl_comm);
        }
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev2) (void)hipEventDestroy(ev2);
        if (ev3) (void)hipEventDestroy(ev3);
        if (hstate_pinned) (void)hipHostFree(hstate_pinned);
        if (hpcg_pinned) (void)hipHostFree(hpcg_pinned);
        if (dense_info_pinned) (void)hipHostFree(dense_info_pinned);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int upload_pattern(BaHandle *h) {
    Symbolic &s = h->sym;
    hipStream_t st = h->stream;
    h->n_upper = (int)s.upper_keys.size();
    h->n_full = (int)s.bsr_col.size();
    h->bcr_state = 0;   // (the elimination schedule belongs to the pattern)
    const int nf = s.n_free;
    HIPCHK(h->ck_pair_slot.upload(s.ck_pair_slot, st));
    HIPCHK(h->hv_pair_slot.upload(s.hv_pair_slot, st));
    HIPCHK(h->rw_slot.upload(s.rw_slot, st));
    HIPCHK(h->diag_slot.upload(s.diag_slot, st));
    HIPCHK(h->bsr_row.upload(s.bsr_row, st));
    HIPCHK(h->bsr_col.upload(s.bsr_col, st));
    HIPCHK(h->bsr_src.upload(s.bsr_src, st));
    {
        std::vector<long long> keys(s.upper_keys.begin(), s.upper_keys.end());
        HIPCHK(h->upper_keys.upload(keys, st));
    }
    const size_t nS = (size_t)h->n_upper * 36, nU = (size_t)nf * 21, ng = (size_t)nf * 6;
    HIPCHK(h->red.alloc_grow(nS + nU + 2 * ng + 4 + BA_MAX_WORLD));  // tail: one slot per rank (see do_build)
    h->S = h->red.p;
    h->U = h->S + nS;
    h->gc = h->U + nU;
    h->gred = h->gc + ng;
    h->sums = h->gred + ng;
    if (h->opt.deterministic && !s.rows) {
        build_det_tables(s, h->det);
        const DetTables &t = h->det;
        {
            std::vector<long long> lg(t.ck_lg.begin(), t.ck_lg.end());
            HIPCHK(h->det_ck_lg.upload(lg, st));
        }
        HIPCHK(h->det_c_ptr.upload(t.c_ptr, st));
        HIPCHK(h->det_c_idx.upload(t.c_idx, st));
        HIPCHK(h->det_s_ptr.upload(t.s_ptr, st));
        HIPCHK(h->det_s_idx.upload(t.s_idx, st));
        HIPCHK(h->det_cam_slab.alloc_grow((size_t)t.cam_rows * 33));
        HIPCHK(h->det_blk_slab.alloc_grow((size_t)t.blk_rows * 36));
    }
    HIPCHK(hipStreamSynchronize(st));
    h->pattern_uploaded = true;
    return BA_OK;
}

DetArgs det_args(BaHandle *h) {
    DetArgs D{};
    if (!h->opt.deterministic || h->sym.rows) return D;   // (the rows mode has no atomics to replace)
    D.cam_slab = h->det_cam_slab.p;
    D.blk_slab = h->det_blk_slab.p;
    D.ck_lg = h->det_ck_lg.p;
    D.n_cl = h->det.n_cl; D.n_raw = h->det.n_raw; D.n_lg = h->det.n_lg;
    return D;
}

int call_coll(BaHandle *h, void *ptr, int64_t n, int op) {
    if (h->rccl_comm) {
        // built-in collectives: enqueued on the handle's stream, in place, no host code in between (the LM loop's only
        // host round trip stays the accept / reject read-back)
        RcclApi *api = rccl_api();
        const int rc = op == BA_BROADCAST0 ? api->Broadcast(ptr, ptr, (size_t)n, 8 /* ncclDouble */, 0, h->rccl_comm, h->stream)
                                           : api->AllReduce(ptr, ptr, (size_t)n, 8, op == BA_REDUCE_MAX ? 2 /* ncclMax */ : 0 /* ncclSum */,
                                                            h->rccl_comm, h->stream);
        if (rc != 0) return fail(BA_ERR_COMM, "RCCL collective failed: %s", api->GetErrorString ? api->GetErrorString(rc) : "?");
        return BA_OK;
    }
    if (!h->coll) return BA_OK;
    if (!h->coll_stream_ordered) HIPCHK(hipStreamSynchronize(h->stream));
    if (h->opt.verbose >= 3) { fprintf(stderr, "[ba_hip r%d] collective op %d n %lld\n", h->rank, op, (long long)n); fflush(stderr); }
    if (h->coll(ptr, n, op, h->coll_user) != 0) return fail(BA_ERR_COMM, "collective callback failed");
    if (h->opt.verbose >= 3) { fprintf(stderr, "[ba_hip r%d] collective done\n", h->rank); fflush(stderr); }
    return BA_OK;
}

BuildArgs build_args(BaHandle *h) {
    BuildArgs A{};
    A.aux = h->aux[h->cur].p;
    A.points = h->points[h->cur].p;
    A.o_cam = h->o_cam.p; A.o_pt = h->o_pt.p; A.o_lcam = h->o_lcam.p; A.o_xy = h->o_xy.p;
    A.cam_free = h->cam_free.p;
    A.ck_obs = h->ck_obs.p; A.ck_pt = h->ck_pt.p; A.ck_cam = h->ck_cam.p; A.ck_cam_list = h->ck_cam_list.p;
    A.ck_pair = h->ck_pair.p; A.ck_pair_slot = h->ck_pair_slot.p; A.ck_pe = h->ck_pe.p; A.ck_pgrp = h->ck_pgrp.p;
    A.pair_entries = h->pair_entries.p;
    A.S = h->S; A.U = h->U; A.gc = h->gc; A.gred = h->gred;
    A.pt_linv = h->pt_linv.p; A.pt_u = h->pt_u.p;
    A.ck_cost = h->ck_cost.p; A.ck_gmax = h->ck_gmax.p;
    A.dbg_v = h->dbg_v.n ? h->dbg_v.p : nullptr;
    A.dbg_gp = h->dbg_gp.n ? h->dbg_gp.p : nullptr;
    A.state = h->state.p;
    A.loss = h->opt.loss;
    A.inv_fs2 = 1.0 / (h->opt.f_scale * h->opt.f_scale);
    A.half_fs2 = 0.5 * h->opt.f_scale * h->opt.f_scale;
    return A;
}

HeavyArgs heavy_args(BaHandle *h) {
    HeavyArgs H{};
    H.n_chunks = h->n_chunks;
    H.pt0 = h->sym.n_pts_chunked;
    H.hv_obs = h->hv_obs.p;
    H.hv_pair = h->hv_pair.p;
    H.hv_pair_slot = h->hv_pair_slot.p;
    H.hv_t = h->hv_t.p;
    return H;
}

// (x, lambda on device) -> cost, S, g_red, block-Jacobi inverse.  Timed pieces: ev0..ev1 fused kernel.
int do_build(BaHandle *h, bool timed) {
    if (h->pattern_stale)
        return fail(BA_ERR_INVALID, "sharded handle after ba_append: call ba_set_pattern with the union of the ranks' patterns first");
    if (!h->pattern_uploaded) {
        int e = upload_pattern(h);
        if (e) return e;
    }
    hipStream_t st = h->stream;
    const int nf = h->sym.n_free;
    const size_t nred = (size_t)h->n_upper * 36 + (size_t)nf * 33 + 4;
    const bool det = h->opt.deterministic != 0 && !h->sym.rows;   // (the rows kernels are deterministic as they are)
    // (deterministic build: k_det_reduce writes every element of S, U, g_c, g_red -- only the sums and the tail are cleared)
    // (rows mode: every block of S that has an entry is STORED by its owner; when these are all the blocks of the pattern -- no
    // block that only another rank or a heavy point contributes to -- S needs no clearing either)
    const bool rows_all = h->sym.rows && (int)h->sym.rw_keys.size() == h->n_upper;
    if (det) HIPCHK(hipMemsetAsync(h->red.p + nred - 4, 0, (4 + BA_MAX_WORLD) * sizeof(double), st));
    else if (rows_all) HIPCHK(hipMemsetAsync(h->red.p + (size_t)h->n_upper * 36, 0, (nred - (size_t)h->n_upper * 36 + BA_MAX_WORLD) * sizeof(double), st));
    else HIPCHK(hipMemsetAsync(h->red.p, 0, (nred + BA_MAX_WORLD) * sizeof(double), st));
    if (timed) HIPCHK(hipEventRecord(h->ev0, st));
    RowsArgs R{};
    R.rows = h->rows_buf.p; R.tails = h->rows_buf.p + (size_t)std::max<long long>(h->rw_rows, 1) * ROW_A; R.ent = h->rw_ent.p; R.ptr = h->rw_ptr.p;
    R.key_slot = h->rw_slot.p; R.row_off = h->rw_row_off; R.xblk = h->rw_xblk.p; R.diag_key = h->rw_diag.p; R.n_x = h->rw_nx;
    // heavy points (tracks no chunk holds): their rows come from k_lin_heavy, their pairs sit in the global entry lists
    auto lin_heavy = [&]() {
        if (h->opt.loss == BA_LOSS_HUBER) hipLaunchKernelGGL((k_lin_heavy<BA_LOSS_HUBER>), dim3(h->n_heavy), dim3(256), 0, st, build_args(h), heavy_args(h), R.rows, R.tails, R.row_off);
        else hipLaunchKernelGGL((k_lin_heavy<-1>), dim3(h->n_heavy), dim3(256), 0, st, build_args(h), heavy_args(h), R.rows, R.tails, R.row_off);
    };
    if (h->n_chunks > 0 && det) {
        if (h->opt.loss == BA_LOSS_HUBER) hipLaunchKernelGGL((k_build_det<BA_LOSS_HUBER>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, build_args(h), det_args(h));
        else hipLaunchKernelGGL((k_build_det<-1>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, build_args(h), det_args(h));
    } else if (h->sym.rows) {
        if (h->n_chunks > 0) {
            if (h->opt.loss == BA_LOSS_HUBER) hipLaunchKernelGGL((k_lin<BA_LOSS_HUBER>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, build_args(h), R.rows, R.tails);
            else hipLaunchKernelGGL((k_lin<-1>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, build_args(h), R.rows, R.tails);
        }
        if (h->n_heavy > 0) lin_heavy();
        {
            const int n_diag = (nf > 0 && h->n_chunks + h->n_heavy > 0) ? nf : 0, n_cross = (h->rw_nx + 127) / 128;
            if (n_diag + n_cross > 0)
                hipLaunchKernelGGL((k_rows_schur<false>), dim3(n_diag + n_cross), dim3(256), 0, st, R, h->S, h->U, h->gc, h->gred, n_diag);
        }
    } else if (h->n_chunks > 0) {
        BuildArgs A = build_args(h);
        if (h->stamps.n) {
            A.stamps = h->stamps.p;
            if (h->opt.loss == BA_LOSS_HUBER) hipLaunchKernelGGL((k_build<true, BA_LOSS_HUBER>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, A);
            else hipLaunchKernelGGL((k_build<true, -1>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, A);
        } else if (h->opt.loss == BA_LOSS_HUBER) {
            hipLaunchKernelGGL((k_build<false, BA_LOSS_HUBER>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, A);
        } else {
            hipLaunchKernelGGL((k_build<false, -1>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, A);
        }
    }
    if (timed) HIPCHK(hipEventRecord(h->ev1, st));
    if (h->n_heavy > 0 && det) {
        hipLaunchKernelGGL(k_build_heavy, dim3(h->n_heavy), dim3(256), 0, st, build_args(h), heavy_args(h), det_args(h));
    } else if (h->n_heavy > 0 && !h->sym.rows) {
        // chunk mode: the lists hold the heavy points only; their blocks are ADDED behind k_build (one owner per block)
        // (no table at all when every observation of the heavy points belongs to a fixed camera)
        lin_heavy();
        {
            const int n_diag = (nf > 0 && h->rw_rows > 0) ? nf : 0, n_cross = (h->rw_nx + 127) / 128;
            if (n_diag + n_cross > 0)
                hipLaunchKernelGGL((k_rows_schur<true>), dim3(n_diag + n_cross), dim3(256), 0, st, R, h->S, h->U, h->gc, h->gred, n_diag);
        }
    }
    if (det) {
        const long long n_el = (long long)h->n_upper * 36 + (long long)nf * 33;
        if (n_el > 0)
            hipLaunchKernelGGL(k_det_reduce, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st, nf, h->n_upper, h->det_c_ptr.p, h->det_c_idx.p,
                               h->det_s_ptr.p, h->det_s_idx.p, h->det_cam_slab.p, h->det_blk_slab.p, h->S, h->U, h->gc, h->gred);
    }
    if (!h->coll) {
        // single GPU: nothing has to happen between the kernels behind the build -- two launches instead of four
        if (nf > 0)
            hipLaunchKernelGGL(k_apply_finalize, dim3((h->n_upper + nf + 127) / 128), dim3(128), 0, st, nf, h->n_upper, h->upper_keys.p,
                               h->free_cam.p, h->aux[h->cur].p, h->S, h->U, h->gc, h->gred, h->diag_slot.p, h->cam_gmax.p, h->state.p);
        h->minv_valid = false;   // (do_pcg computes the block-Jacobi inverse when it needs it)
        hipLaunchKernelGGL(k_totals_state, dim3(1), dim3(1024), 0, st, h->n_chunks + h->n_heavy, h->ck_cost.p, h->ck_gmax.p, h->sums,
                           h->gm2.p, h->state.p, nf, h->cam_gmax.p);
    } else {
        hipLaunchKernelGGL(k_chunk_totals, dim3(1), dim3(1024), 0, st, h->n_chunks + h->n_heavy, h->ck_cost.p, h->ck_gmax.p, h->sums, h->gm2.p);
        // ONE collective per build: the maximum over ranks of the point-gradient norm rides in the sum buffer,
        // each rank's value in its own slot of the tail (zero elsewhere), and is maximised locally afterwards
        hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, st, h->gm2.p, h->red.p + nred + h->rank);
        int e = call_coll(h, h->red.p, (int64_t)nred + h->world, BA_REDUCE_SUM);
        if (e) return e;
        hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(256), 0, st, h->world, h->red.p + nred, h->gm2.p);
        if (nf > 0) {  // (after the all-reduce: the transform is linear, and every rank holds the same cameras)
            hipLaunchKernelGGL(k_apply_jl, dim3((h->n_upper + nf + 127) / 128), dim3(128), 0, st, nf, h->n_upper, h->upper_keys.p,
                               h->free_cam.p, h->aux[h->cur].p, h->S, h->U, h->gc, h->gred);
            hipLaunchKernelGGL(k_finalize, dim3((nf + 63) / 64), dim3(64), 0, st, nf, h->U, h->gc, h->gred, h->S, h->diag_slot.p,
                               h->Minv.p, h->cam_gmax.p, h->state.p);
            h->minv_valid = true;
        }
        hipLaunchKernelGGL(k_state_after_build, dim3(1), dim3(256), 0, st, h->state.p, h->sums, h->gm2.p, nf, h->cam_gmax.p);
    }
    HIPCHK(hipGetLastError());
    h->built = true;
    return BA_OK;
}

int read_state(BaHandle *h) {
    // (into page-locked memory: a copy to pageable memory goes through the runtime's staging buffer, and this read-back is
    // the one host round trip of every LM step)
    if (!h->hstate_pinned) HIPCHK(hipHostMalloc((void **)&h->hstate_pinned, sizeof(LmState), hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(h->hstate_pinned, h->state.p, sizeof(LmState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->hstate = *h->hstate_pinned;
    return BA_OK;
}

// Relative residual at which the iterative solves of this step stop.  BaOptions.pcg_tol > 0: that, every step.
// pcg_tol = 0 (the default): an inexact-Newton forcing term, eta = 1e-2 min(0.5, |g|inf) -- the rule of scipy's own
// lsmr_tol = 'auto' (scipy/optimize/_lsq/dogbox.py; the reference's TRF runs lsmr at atol = btol = 1e-6,
// BundleAdjustment.py:210-220 with scipy/optimize/_lsq/trf.py) -- loose while the LM step is far from the minimum (the
// damping absorbs an inexact step: same number of LM steps, same minimum on every shape of the bench at a third of the
// iterations), tightening with the gradient as the steps become Newton steps.  |g|inf is the handle's latest read-back
// (the state one build ago at worst), identical on every rank of a sharded solve.
double pcg_tol_now(const BaHandle *h) {
    if (h->opt.pcg_tol > 0.0) return h->opt.pcg_tol;
    const double g = h->hstate.gmax;
    double eta = 1e-2 * (g < 0.5 ? g : 0.5);
    if (!(eta > 1e-10)) eta = 1e-10;
    return eta;
}

// block-Jacobi PCG on S x = -g_red; result in h->px
int do_pcg(BaHandle *h, int *iters_out) {
    hipStream_t st = h->stream;
    const int nf = h->sym.n_free;
    *iters_out = 0;
    if (nf == 0) return BA_OK;
    const int nb_cam = (nf + 255) / 256, nb_row = (nf + 3) / 4;
    if (!h->minv_valid) {
        hipLaunchKernelGGL(k_block_jacobi, dim3((nf + 63) / 64), dim3(64), 0, st, nf, h->S, h->diag_slot.p, h->Minv.p);
        h->minv_valid = true;
    }
    HIPCHK(hipMemsetAsync(h->pcg_sc.p, 0, sizeof(PcgScalars), st));
    PcgPartials pp{};
    {
        HIPCHK(h->pcg_part.alloc_grow((size_t)7 * nb_cam + (size_t)2 * nb_row));
        double *q = h->pcg_part.p;
        for (int k = 0; k < 3; ++k) { pp.rz[k] = q; q += nb_cam; pp.rn[k] = q; q += nb_cam; }
        pp.bb = q; q += nb_cam;
        pp.pq[0] = q; pp.pq[1] = q + nb_row;
        pp.nb_cam = nb_cam; pp.nb_row = nb_row;
    }
    hipLaunchKernelGGL(k_pcg_init, dim3(nb_cam), dim3(256), 0, st, nf, h->gred, h->Minv.p, h->px.p, h->pr.p, h->pz.p, pp);
    const double tol_now = pcg_tol_now(h), tol2 = tol_now * tol_now;
    const int max_it = h->opt.pcg_max_iter > 0 ? h->opt.pcg_max_iter : 500;
    const int batch = h->opt.pcg_tol > 0.0 ? 16 : 8;   // (iterations queued between two polls of the convergence flag)
    int it = 0;
    while (it < max_it) {
        const int end = std::min(max_it, it + batch);
        for (; it < end; ++it) {
            double *p_in = (it & 1) ? h->pq.p : h->pp.p, *p_out = (it & 1) ? h->pp.p : h->pq.p;
            hipLaunchKernelGGL(k_pcg_a, dim3(nb_row), dim3(256), 0, st, nf, it, tol2, h->bsr_row.p, h->bsr_col.p,
                               h->bsr_src.p, h->S, h->pz.p, p_in, p_out, h->pqv.p, h->pcg_sc.p, pp);
            hipLaunchKernelGGL(k_pcg_b, dim3(nb_cam), dim3(256), 0, st, nf, it, h->Minv.p, p_out, h->pqv.p, h->px.p,
                               h->pr.p, h->pz.p, h->pcg_sc.p, pp);
        }
        if (!h->hpcg_pinned) HIPCHK(hipHostMalloc((void **)&h->hpcg_pinned, sizeof(PcgScalars), hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(h->hpcg_pinned, h->pcg_sc.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        h->hpcg = *h->hpcg_pinned;
        if (h->hpcg.done != 0.0) break;
    }
    *iters_out = (int)h->hpcg.iters;
    return BA_OK;
}

// dense Cholesky solve of S x = -g_red (small systems), launched only; *launched = false when there is nothing to solve
int do_dense(BaHandle *h, bool *launched) {
    hipStream_t st = h->stream;
    const int n = h->sym.n_free * 6;
    *launched = false;
    if (n == 0) return BA_OK;
    *launched = true;
    if (h->dense_a.n == 0) {
        HIPCHK(h->dense_a.alloc((size_t)n * n + 8));  // (+8: the trailing update reads whole 4-column strips of the last row)
        HIPCHK(h->dense_info.alloc(1));
    }
    HIPCHK(hipMemsetAsync(h->dense_a.p, 0, (size_t)n * n * sizeof(double), st));
    const int ne = h->n_upper * 36;
    hipLaunchKernelGGL(k_dense_assemble, dim3((ne + 255) / 256), dim3(256), 0, st, h->n_upper, h->upper_keys.p, h->S, n,
                       h->dense_a.p);
    {
        // the panel of a block column lives in dynamic LDS when it fits (beyond the 64 KB a kernel gets by default)
        const int rows = n > BA_NB ? n - BA_NB : 0, lds_rows = rows < BA_DENSE_LDS_ROWS ? rows : BA_DENSE_LDS_ROWS;
        const size_t dyn = (size_t)lds_rows * BA_PS * sizeof(double);
        if (!h->dense_attr_set) {  // (per handle: the attribute belongs to the function on the handle's device)
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_cholesky_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       BA_DENSE_LDS_ROWS * BA_PS * (int)sizeof(double)));
            h->dense_attr_set = true;
        }
        hipLaunchKernelGGL(k_dense_cholesky_solve, dim3(1), dim3(BA_DENSE_THREADS), dyn, st, n, h->dense_a.p, h->gred, h->px.p,
                           h->dense_info.p, lds_rows);
    }
    HIPCHK(hipGetLastError());
    return BA_OK;  // (the pivot flag in dense_info is read together with the step's outcome, see do_step)
}

// blocked multi-workgroup dense Cholesky (ba_dense_big.h), launched only
int do_dense_big(BaHandle *h, bool *launched) {
    const int n = h->sym.n_free * 6;
    *launched = false;
    if (n == 0) return BA_OK;
    *launched = true;
    if (h->dense_a.n != big_matrix_doubles(n)) {
        HIPCHK(h->dense_a.alloc(big_matrix_doubles(n)));
        HIPCHK(h->dense_linv.alloc(big_linv_doubles(n)));
        if (h->dense_info.n == 0) HIPCHK(h->dense_info.alloc(1));
    }
    HIPCHK(big_solve_launch(h->stream, n, h->n_upper, h->upper_keys.p, h->S, h->gred, h->px.p, h->dense_a.p, h->dense_linv.p,
                            h->dense_info.p));
    return BA_OK;  // (pivot flag: see do_step)
}

// block-banded Cholesky solve of S x = -g_red, launched only
int do_banded(BaHandle *h, bool *launched) {
    hipStream_t st = h->stream;
    const int N = h->sym.n_free, B = h->sym.band;
    *launched = false;
    if (N == 0) return BA_OK;
    *launched = true;
    if (h->band_l.n != band_l_doubles(N, B)) {
        HIPCHK(h->band_l.alloc(band_l_doubles(N, B)));
        HIPCHK(h->band_y.alloc(band_vec_doubles(N, B)));
        HIPCHK(h->band_g.alloc(band_vec_doubles(N, B)));
        HIPCHK(h->band_mid.alloc(band_mid_doubles(B)));
        if (h->dense_info.n == 0) HIPCHK(h->dense_info.alloc(1));
    }
    const BandWork w{h->band_l.p, h->band_y.p, h->band_g.p, h->band_mid.p, h->dense_info.p};
    HIPCHK(band_solve_launch(st, N, B, h->n_upper, h->upper_keys.p, h->S, h->gred, h->px.p, w));
    return BA_OK;  // (pivot flag: see do_step)
}

// block cyclic reduction solve of S x = -g_red (chain / ring patterns), launched only.  *usable = false when the
// pattern has no such schedule (nothing is launched then).
int bcr_plan_for(BaHandle *h) {
    if (h->bcr_state != 0) return BA_OK;
    const Symbolic &s = h->sym;
    h->bcr_state = -1;
    try {
        // (a plan for the in-band part only -- a chain / ring plus a remainder -- is accepted too: do_bcr_pcg)
        if (!bcr_make_plan(s.n_free, s.upper_keys.data(), s.upper_keys.size(), h->bcr_plan, true)) return BA_OK;
    } catch (const std::bad_alloc &) {
        return fail(BA_ERR_NOMEM, "out of host memory while planning the cyclic reduction");
    }
    const BcrPlan &P = h->bcr_plan;
    std::vector<int32_t> tab;
    auto put = [&](const void *src, size_t ints) {
        const size_t at = tab.size();
        tab.resize(at + ints + 1);   // (+1: no empty table)
        if (ints) memcpy(tab.data() + at, src, ints * sizeof(int32_t));
        return at;
    };
    h->bcr_off[0] = put(P.elims.data(), P.elims.size() * sizeof(BcrElim) / 4);
    h->bcr_off[1] = put(P.ops.data(), P.ops.size() * sizeof(BcrOp) / 4);
    h->bcr_off[2] = put(P.src.data(), P.src.size());
    h->bcr_off[3] = put(P.pos_f.data(), P.pos_f.size());
    h->bcr_off[4] = put(P.fwd.data(), P.fwd.size());
    HIPCHK(h->bcr_tab.upload(tab, h->stream));
    HIPCHK(h->bcr_dg.alloc(bcr_matrix_doubles(P)));
    HIPCHK(h->bcr_bv.alloc(bcr_vec_doubles(P)));
    HIPCHK(h->bcr_linv.alloc(bcr_linv_doubles(P)));
    if (h->dense_info.n == 0) HIPCHK(h->dense_info.alloc(1));
    HIPCHK(bcr_prepare_device(P.nt));
    h->bcr_state = 1;
    return BA_OK;
}
BcrDev bcr_dev(BaHandle *h) {
    const int32_t *t = h->bcr_tab.p;
    BcrDev d{};
    d.elims = reinterpret_cast<const BcrElim *>(t + h->bcr_off[0]);
    d.ops = reinterpret_cast<const BcrOp *>(t + h->bcr_off[1]);
    d.src = t + h->bcr_off[2];
    d.pos_f = t + h->bcr_off[3];
    d.fwd = t + h->bcr_off[4];
    d.DG = h->bcr_dg.p; d.bv = h->bcr_bv.p; d.bacc = h->bcr_bv.p + (size_t)h->bcr_plan.m * h->bcr_plan.nb; d.linv = h->bcr_linv.p; d.info = h->dense_info.p;
    return d;
}
int do_bcr(BaHandle *h, bool *launched) {
    *launched = false;
    if (h->sym.n_free == 0 || h->bcr_state != 1) return BA_OK;
    *launched = true;
    HIPCHK(bcr_solve_launch(h->stream, h->bcr_plan, bcr_dev(h), h->S, h->gred, h->px.p));
    return BA_OK;  // (pivot flag: see do_step)
}

// A chain / ring PLUS a remainder (a revisit, a few long tracks; bcr_plan.n_rest > 0): CG on the whole S,
// preconditioned by the cyclic-reduction factorisation of its in-band part M.  The factorisation is computed once per
// step together with z0 = M^-1 (-g) (the direct solver's own launch sequence); every further iteration applies the
// stored factors to the new residual (k_bcr_fwd per level, then k_bcr_back per level).  Block-Jacobi PCG needs
// hundreds of iterations on such a chain (its low-frequency modes); with M they are solved exactly and what is left
// for CG is the remainder.  Result in h->px; *ok = false when the factorisation of M met a bad pivot (M is S without
// some off-diagonal blocks: not necessarily positive definite) -- the caller falls back to block-Jacobi PCG.
int do_bcr_pcg(BaHandle *h, int *iters_out, bool *ok) {
    hipStream_t st = h->stream;
    const int nf = h->sym.n_free;
    *iters_out = 0;
    *ok = true;
    if (nf == 0) return BA_OK;
    const int nb_cam = (nf + 255) / 256, nb_row = (nf + 3) / 4;
    const BcrDev d = bcr_dev(h);
    HIPCHK(bcr_solve_launch(st, h->bcr_plan, d, h->S, h->gred, h->pz.p));   // factors of M, z0 = M^-1 (-g)
    HIPCHK(bcr_invert_launch(st, h->bcr_plan, d));                           // X_k = L_k^-1: the sweeps below are matrix-vector products
    HIPCHK(hipMemsetAsync(h->pcg_sc.p, 0, sizeof(PcgScalars), st));
    PcgPartials pp{};
    {
        HIPCHK(h->pcg_part.alloc_grow((size_t)7 * nb_cam + (size_t)2 * nb_row));
        double *q = h->pcg_part.p;
        for (int k = 0; k < 3; ++k) { pp.rz[k] = q; q += nb_cam; pp.rn[k] = q; q += nb_cam; }
        pp.bb = q; q += nb_cam;
        pp.pq[0] = q; pp.pq[1] = q + nb_row;
        pp.nb_cam = nb_cam; pp.nb_row = nb_row;
    }
    hipLaunchKernelGGL(k_pcgx_init, dim3(nb_cam), dim3(256), 0, st, nf, h->gred, h->pz.p, h->px.p, h->pr.p, pp);
    const double tol_now = pcg_tol_now(h), tol2 = tol_now * tol_now;
    // (with a useful preconditioner CG converges in a handful of iterations; one that has not after BCR_PCG_CAP is no
    // help on this problem: the plan is dropped and block-Jacobi PCG takes over, this step and from now on)
    constexpr int BCR_PCG_CAP = 120;
    const int max_it = std::min(h->opt.pcg_max_iter > 0 ? h->opt.pcg_max_iter : 500, BCR_PCG_CAP);
    const double *done = &h->pcg_sc.p->done;
    if (!h->hpcg_pinned) HIPCHK(hipHostMalloc((void **)&h->hpcg_pinned, sizeof(PcgScalars), hipHostMallocDefault));
    if (h->dense_info_pinned == nullptr) HIPCHK(hipHostMalloc((void **)&h->dense_info_pinned, sizeof(int), hipHostMallocDefault));
    const int batch = 2;   // (an iteration is some twenty launches: poll the convergence flag often)
    int it = 0;
    while (it < max_it) {
        const int end = std::min(max_it, it + batch);
        for (; it < end; ++it) {
            double *p_in = (it & 1) ? h->pq.p : h->pp.p, *p_out = (it & 1) ? h->pp.p : h->pq.p;
            hipLaunchKernelGGL(k_pcg_a, dim3(nb_row), dim3(256), 0, st, nf, it, tol2, h->bsr_row.p, h->bsr_col.p,
                               h->bsr_src.p, h->S, h->pz.p, p_in, p_out, h->pqv.p, h->pcg_sc.p, pp);
            hipLaunchKernelGGL(k_pcg_b, dim3(nb_cam), dim3(256), 0, st, nf, it, (const double *)nullptr, p_out, h->pqv.p, h->px.p,
                               h->pr.p, h->pz.p, h->pcg_sc.p, pp);
            HIPCHK(bcr_apply_launch(st, h->bcr_plan, d, h->pr.p, h->pz.p, done));
            hipLaunchKernelGGL(k_pcgx_dot, dim3(nb_cam), dim3(256), 0, st, nf, it, h->pr.p, h->pz.p, h->pcg_sc.p, pp);
        }
        HIPCHK(hipMemcpyAsync(h->hpcg_pinned, h->pcg_sc.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, st));
        if (it <= batch) HIPCHK(hipMemcpyAsync(h->dense_info_pinned, h->dense_info.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        h->hpcg = *h->hpcg_pinned;
        if (it <= batch && *h->dense_info_pinned != 0) {   // the factorisation of M failed: not a preconditioner
            *ok = false;
            return BA_OK;
        }
        if (h->hpcg.done != 0.0) break;
    }
    *iters_out = (int)h->hpcg.iters;
    if (h->hpcg.done == 0.0 && max_it == BCR_PCG_CAP) {
        *ok = false;
        h->bcr_state = -1;
    }
    return BA_OK;
}

// the part of a trial step behind the linear solve: camera update, back-substitution + trial cost, reductions, decision
int step_tail(BaHandle *h, const int *solve_info) {
    hipStream_t st = h->stream;
    const Symbolic &s = h->sym;
    int e;
    if (h->coll && s.n_free > 0 && (e = call_coll(h, h->px.p, (int64_t)s.n_free * 6, BA_BROADCAST0))) return e;
    const int cur = h->cur, tr = cur ^ 1;
    hipLaunchKernelGGL(k_cam_update, dim3((s.n_cams + 63) / 64), dim3(64), 0, st, s.n_cams, h->cam_free.p, h->pose[cur].p,
                       h->intr.p, h->px.p, h->pose[tr].p, h->aux[tr].p, h->cam_part.p, h->aux[cur].p, h->pr.p);
    // (h->pr, the PCG's residual vector, is free between two solves: it carries the step in the local rotation tangent)
    if (h->n_chunks + h->n_heavy > 0) {
        BacksubArgs B{};
        B.aux = h->aux[cur].p; B.aux_t = h->aux[tr].p;
        B.points = h->points[cur].p; B.points_t = h->points[tr].p;
        B.o_cam = h->o_cam.p; B.o_pt = h->o_pt.p; B.o_xy = h->o_xy.p;
        B.cam_free = h->cam_free.p;
        B.ck_obs = h->ck_obs.p; B.ck_pt = h->ck_pt.p;
        B.dx = h->pr.p;
        B.dx_rvec = h->px.p;
        B.pt_linv = h->pt_linv.p; B.pt_u = h->pt_u.p;
        B.ck_part = h->ck_part.p;
        B.loss = h->opt.loss;
        B.state = h->state.p;
        B.inv_fs2 = 1.0 / (h->opt.f_scale * h->opt.f_scale);
        B.half_fs2 = 0.5 * h->opt.f_scale * h->opt.f_scale;
        if (h->n_chunks > 0 && h->opt.deterministic) {
            if (h->opt.loss == BA_LOSS_HUBER) hipLaunchKernelGGL((k_backsub_det<BA_LOSS_HUBER>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, B);
            else hipLaunchKernelGGL((k_backsub_det<-1>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, B);
        } else if (h->n_chunks > 0) {
            if (h->opt.loss == BA_LOSS_HUBER) hipLaunchKernelGGL((k_backsub<BA_LOSS_HUBER>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, B);
            else hipLaunchKernelGGL((k_backsub<-1>), dim3(h->n_chunks), dim3(CHUNK_OBS), 0, st, B);
        }
        if (h->n_heavy > 0) hipLaunchKernelGGL(k_backsub_heavy, dim3(h->n_heavy), dim3(256), 0, st, B, heavy_args(h));
    }
    if (!h->coll) {   // single GPU: reductions, |x|^2 of the unobserved points and the decision in ONE launch
        hipLaunchKernelGGL(k_reduce_decide, dim3(1), dim3(1024), 0, st, h->n_chunks + h->n_heavy, h->ck_part.p, s.n_cams, h->cam_part.p,
                           h->step_sums.p, h->cam_sums.p, h->state.p, h->x2_const, h->opt.ftol, h->opt.xtol, h->opt.gtol, solve_info);
        HIPCHK(hipGetLastError());
        return BA_OK;
    }
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, st, h->n_chunks + h->n_heavy, 4, h->ck_part.p, h->step_sums.p);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, st, s.n_cams, 2, h->cam_part.p, h->cam_sums.p);
    // |x|^2 of this rank's unobserved points joins the partial sums BEFORE the all-reduce so that
    // every rank takes the same decision
    hipLaunchKernelGGL(k_add_scalar, dim3(1), dim3(1), 0, st, h->step_sums.p + 3, h->x2_const);
    if (h->coll && (e = call_coll(h, h->step_sums.p, 4, BA_REDUCE_SUM))) return e;
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(1), 0, st, h->state.p, h->step_sums.p, h->cam_sums.p, 0.0,
                       h->opt.ftol, h->opt.xtol, h->opt.gtol, solve_info);
    HIPCHK(hipGetLastError());
    return BA_OK;
}

// one LM trial step from the current (built) state; leaves the decision in h->hstate
int do_step(BaHandle *h) {
    const Symbolic &s = h->sym;
    int pcg_it = 0;
    int e;
    const int n_red = s.n_free * 6;
    bool direct = false;
    const int ls = h->opt.linear_solver;
    // (the two-sided banded solver merges its two halves with fp64 atomics: not in deterministic mode)
    const bool can_dense = n_red <= BA_DENSE_MAX, can_band = s.band <= BA_BAND_MAX && !h->opt.deterministic, can_big = n_red <= BA_DENSE_BIG_MAX;
    // "a genuine dense contraction": at least a quarter of the 6x6 blocks of S are there (every camera shares landmarks
    // with most others); a sparse S of that size is cheaper through PCG
    const bool dense_pattern = (double)h->n_full >= 0.25 * (double)s.n_free * (double)s.n_free;
    // chain- or ring-shaped co-visibility beyond the one-workgroup dense solver: block cyclic reduction (whole chip)
    bool can_bcr = false;
    if ((ls == BA_SOLVER_CYCLIC_REDUCTION || (ls == BA_SOLVER_AUTO && !can_dense)) && s.n_free > 0) {
        if ((e = bcr_plan_for(h))) return e;
        can_bcr = h->bcr_state == 1;
    }
    // A plan with a remainder (chain / ring plus a few far-apart pairs) is no direct solver: it preconditions CG (do_bcr_pcg)
    const bool bcr_rest = can_bcr && h->bcr_plan.n_rest > 0;
    // AUTO on a (nearly) full pattern beyond the one-workgroup solver: blocked dense Cholesky or block-Jacobi PCG, whichever
    // is faster HERE -- the dense solve costs n^3 / 3 whatever the problem, PCG costs its iteration count, which depends on
    // the conditioning (object-centric captures: a few dozen iterations).  Both are timed once with events on the stream
    // (the step's one host synchronisation is there anyway), then the faster one is kept; PCG is timed again when its
    // iteration count has grown by half.
    const bool auto_pick = ls == BA_SOLVER_AUTO && !can_dense && !can_bcr && !can_band && can_big && dense_pattern;
    bool time_direct = false, time_pcg = false, skip_direct = false;
    if (auto_pick) {
        if (h->auto_phase < 0) {   // (the direct solver's first use allocates its work space: not a timing)
            time_direct = h->dense_a.n == big_matrix_doubles(n_red);
            h->auto_phase = 0;
        } else if (h->auto_phase == 0) time_direct = true;
        else if (h->auto_phase == 1) { time_pcg = true; skip_direct = true; }
        else skip_direct = h->auto_use_pcg;
    }
    if ((ls == BA_SOLVER_AUTO || ls == BA_SOLVER_DENSE_CHOLESKY) && can_dense) {
        if ((e = do_dense(h, &direct))) return e;
    } else if (can_bcr && !bcr_rest) {
        if ((e = do_bcr(h, &direct))) return e;
    } else if (bcr_rest) {
        // (iterative: below)
    } else if ((ls == BA_SOLVER_AUTO || ls == BA_SOLVER_BANDED_CHOLESKY) && can_band) {
        if ((e = do_banded(h, &direct))) return e;
    } else if (can_big && !skip_direct && (ls == BA_SOLVER_DENSE_CHOLESKY || (ls == BA_SOLVER_AUTO && dense_pattern))) {
        if (time_direct) HIPCHK(hipEventRecord(h->ev2, h->stream));
        if ((e = do_dense_big(h, &direct))) return e;
        if (time_direct) HIPCHK(hipEventRecord(h->ev3, h->stream));
    }
    if (direct) {
        // The direct solvers leave a pivot flag on the device.  The rest of the step is queued behind them without
        // looking at it (no host round trip in the middle of the step); a step whose flag is set is void -- k_decide
        // leaves the LM state alone -- and is repeated with PCG.
        if (getenv("BA_DEBUG_FAIL_DIRECT"))  // test hook: pretend the factorisation met a bad pivot
            HIPCHK(hipMemsetAsync(h->dense_info.p, 1, sizeof(int), h->stream));
        if ((e = step_tail(h, h->dense_info.p))) return e;
        if ((e = read_state(h))) return e;  // (k_decide copies the flag into the state: one copy brings both)
        const int info = h->hstate.solve_info;
        if (h->opt.verbose >= 3) { fprintf(stderr, "[ba_hip r%d] direct solve info=%d\n", h->rank, info); fflush(stderr); }
        if (time_direct) {
            if (hipEventElapsedTime(&h->auto_ms_direct, h->ev2, h->ev3) != hipSuccess) h->auto_ms_direct = 0.f;
            h->auto_phase = 1;
        }
        if (info == 0) {
            if (h->hstate.accept) h->cur ^= 1;
            return BA_OK;
        }
        h->dense_fallbacks++;
    }
    bool solved = false;
    if (bcr_rest && !getenv("BA_DEBUG_FAIL_DIRECT")) {
        if ((e = do_bcr_pcg(h, &pcg_it, &solved))) return e;
        if (!solved) h->dense_fallbacks++;
    }
    if (!solved) {
        if (time_pcg) HIPCHK(hipEventRecord(h->ev2, h->stream));
        if ((e = do_pcg(h, &pcg_it))) return e;
        if (time_pcg) {
            HIPCHK(hipEventRecord(h->ev3, h->stream));
            HIPCHK(hipEventSynchronize(h->ev3));
            if (hipEventElapsedTime(&h->auto_ms_pcg, h->ev2, h->ev3) != hipSuccess) h->auto_ms_pcg = 0.f;
            const int max_it = h->opt.pcg_max_iter > 0 ? h->opt.pcg_max_iter : 500;
            h->auto_pcg_iters = pcg_it;
            // (a PCG that ran into its iteration cap did not solve the system: never the choice)
            h->auto_use_pcg = pcg_it < max_it && h->auto_ms_pcg > 0.f && h->auto_ms_direct > 0.f && h->auto_ms_pcg < h->auto_ms_direct;
            h->auto_phase = 2;
            if (h->opt.verbose >= 3)
                printf("[ba_hip] AUTO on a full pattern: blocked dense Cholesky %.2f ms, block-Jacobi PCG %.2f ms (%d iterations) -> %s\n",
                       h->auto_ms_direct, h->auto_ms_pcg, pcg_it, h->auto_use_pcg ? "PCG" : "dense");
        } else if (auto_pick && h->auto_phase == 2 && h->auto_use_pcg && 2 * pcg_it > 3 * h->auto_pcg_iters + 20) {
            h->auto_phase = 1;   // the system has become harder for PCG: time it again next step
        }
    }
    h->total_pcg += pcg_it;
    if ((e = step_tail(h, nullptr))) return e;
    if ((e = read_state(h))) return e;
    if (h->hstate.accept) h->cur ^= 1;
    return BA_OK;
}

int set_lambda(BaHandle *h, double lambda) {
    HIPCHK(hipStreamSynchronize(h->stream));
    h->lambda_host = lambda;
    HIPCHK(hipMemcpyAsync(&h->state.p->lambda, &h->lambda_host, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return BA_OK;
}

// The numeric half of a problem: values in the internal order, both state copies, camera constants, and a fresh
// LM state.  The host vectors are read by asynchronous copies, hence the synchronise at the end.
int upload_values(BaHandle *h, const HostProblem &hp) {
    const Symbolic &s = h->sym;
    hipStream_t st = h->stream;
    std::vector<double> pts((size_t)s.n_points * 3), xy((size_t)s.n_obs * 2);
    for (int i = 0; i < s.n_points; ++i)
        for (int k = 0; k < 3; ++k) pts[3 * (size_t)i + k] = hp.points[3 * (size_t)s.pt_order[i] + k];
    for (int64_t k = 0; k < s.n_obs; ++k) {
        xy[2 * k] = hp.obs_xy[2 * s.obs_order[k]];
        xy[2 * k + 1] = hp.obs_xy[2 * s.obs_order[k] + 1];
    }
    h->x2_const = 0.0;
    for (int i = s.n_pts_obs; i < s.n_points; ++i)
        for (int k = 0; k < 3; ++k) h->x2_const += pts[3 * (size_t)i + k] * pts[3 * (size_t)i + k];
    for (int b = 0; b < 2; ++b) {
        HIPCHK(h->pose[b].refill(hp.cam_pose, st));
        HIPCHK(h->points[b].refill(pts, st));
    }
    HIPCHK(h->intr.refill(hp.cam_intr, st));
    HIPCHK(h->o_xy.refill(xy, st));
    LmState s0{};
    s0.lambda = h->opt.lambda0;
    s0.nu = 2.0;
    s0.floor = 1.0;
    s0.floor_min = h->opt.curv_floor;
    h->hstate = s0;
    h->cur = 0;
    h->total_pcg = 0;
    h->dense_fallbacks = 0;
    h->auto_phase = -1;
    h->initial_cost = NAN;
    h->built = false;
    HIPCHK(hipMemcpyAsync(h->state.p, &h->hstate, sizeof(LmState), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(h->gm2.p, 0, 2 * sizeof(double), st));
    if (s.n_cams > 0)
        hipLaunchKernelGGL(k_cam_prep, dim3((s.n_cams + 63) / 64), dim3(64), 0, st, s.n_cams, h->pose[0].p, h->intr.p,
                           h->aux[0].p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return BA_OK;
}

int check_options(BaOptions &o) {
    if (!(o.f_scale > 0.0)) return fail(BA_ERR_INVALID, "f_scale must be positive");
    if (o.loss < 0 || o.loss > 4) return fail(BA_ERR_INVALID, "unknown loss");
    if (!(o.lambda0 > 0.0)) o.lambda0 = 1e-4;
    if (!(o.pcg_tol > 0.0)) o.pcg_tol = 0.0;   // adaptive (pcg_tol_now)
    if (!(o.curv_floor > 0.0) || o.curv_floor > 1.0) o.curv_floor = 0.1;
    o.deterministic = o.deterministic ? 1 : 0;
    if (o.schur_mode < 0 || o.schur_mode > 2) return fail(BA_ERR_INVALID, "schur_mode must be 0 (auto), 1 (chunks) or 2 (rows)");
    return BA_OK;
}

// work buffers whose sizes follow the structure (no contents to keep): called after the structure (re)build
int alloc_work(BaHandle *h) {
    const Symbolic &s = h->sym;
    const int nf = s.n_free;
    for (int b = 0; b < 2; ++b) HIPCHK(h->aux[b].alloc_grow((size_t)s.n_cams * BA_CAM_AUX));
    HIPCHK(h->Minv.alloc_grow((size_t)nf * 36));
    HIPCHK(h->cam_gmax.alloc_grow((size_t)nf));
    HIPCHK(h->gm2.alloc(2));
    HIPCHK(h->pt_linv.alloc_grow((size_t)s.n_pts_obs * 6));
    HIPCHK(h->pt_u.alloc_grow((size_t)s.n_pts_obs * 3));
    HIPCHK(h->ck_cost.alloc_grow((size_t)(h->n_chunks + h->n_heavy)));
    HIPCHK(h->ck_gmax.alloc_grow((size_t)(h->n_chunks + h->n_heavy)));
    HIPCHK(h->ck_part.alloc_grow((size_t)(h->n_chunks + h->n_heavy) * 4));
    HIPCHK(h->cam_part.alloc_grow((size_t)s.n_cams * 2));
    HIPCHK(h->step_sums.alloc(4));
    HIPCHK(h->cam_sums.alloc(2));
    HIPCHK(h->px.alloc_grow((size_t)nf * 6));
    HIPCHK(h->pr.alloc_grow((size_t)nf * 6));
    HIPCHK(h->pz.alloc_grow((size_t)nf * 6));
    HIPCHK(h->pp.alloc_grow((size_t)nf * 6));
    HIPCHK(h->pq.alloc_grow((size_t)nf * 6));
    HIPCHK(h->pqv.alloc_grow((size_t)nf * 6));
    HIPCHK(h->pcg_sc.alloc(1));
    HIPCHK(h->pcg_part.alloc_grow((size_t)7 * ((nf + 255) / 256) + (size_t)2 * ((nf + 3) / 4) + 1));
    if (!h->hpcg_pinned) HIPCHK(hipHostMalloc((void **)&h->hpcg_pinned, sizeof(PcgScalars), hipHostMallocDefault));
    HIPCHK(h->state.alloc(1));
    // solver work space is sized on first use: drop what belonged to another structure
    h->dense_a.release(); h->dense_linv.release();
    h->band_l.release(); h->band_y.release(); h->band_g.release(); h->band_mid.release();
    h->bcr_tab.release(); h->bcr_dg.release(); h->bcr_bv.release(); h->bcr_linv.release();
    h->dbg_v.release(); h->dbg_gp.release();
    h->pattern_uploaded = false;
    return BA_OK;
}

// whole structure of a problem: symbolic analysis on the host, every index array to the device
int build_structure(BaHandle *h, const HostProblem &hp, int n_cams, int n_points, int64_t n_obs) {
    int e;
    const char *msg = "";
    // BaOptions.schur_mode: 0 AUTO (chunks; rows when the chunk-local groups are mostly padding), 1 chunks, 2 rows.
    const bool det = h->opt.deterministic != 0;
    auto analyse = [&](bool rows) -> int {
        const int rc = analyze_structure(n_cams, n_points, n_obs, hp.cam_fixed.data(), hp.obs_cam.data(), hp.obs_pt.data(), h->sym, &msg, det, rows);
        return rc ? fail(rc, "%s", msg) : BA_OK;
    };
    if ((e = analyse(h->opt.schur_mode == 2))) return e;
    if (h->opt.schur_mode == 0 && wants_rows_mode(h->sym) && (double)n_obs * ROW_DOUBLES * 8 < 16e9 && (e = analyse(true))) return e;
    const bool tm_on = getenv("BA_TIME") != nullptr;
    double tl = now_ms();
    finalize_pattern(h->sym, {});
    if (tm_on) { fprintf(stderr, "[create]   %-22s %7.2f ms\n", "finalize_pattern", now_ms() - tl); tl = now_ms(); }
    Symbolic &s = h->sym;
    hipStream_t st = h->stream;
    h->n_chunks = (int)s.ck_obs.size() - 1;
    h->n_heavy = s.n_heavy;
    if (s.n_heavy > 0) {
        std::vector<long long> ho(s.hv_obs.begin(), s.hv_obs.end()), hq(s.hv_pair.begin(), s.hv_pair.end());
        HIPCHK(h->hv_obs.upload(ho, st));
        HIPCHK(h->hv_pair.upload(hq, st));
        if (!s.hv_keys.empty()) HIPCHK(h->hv_t.alloc((size_t)(s.hv_obs.back() - s.hv_obs.front()) * 18));   // (k_build_heavy: deterministic chunk mode)
        else h->hv_t.release();
    }
    HIPCHK(h->cam_free.upload(s.cam_free, st));
    HIPCHK(h->free_cam.upload(s.free_cam, st));
    HIPCHK(h->o_cam.upload(s.o_cam, st));
    HIPCHK(h->o_pt.upload(s.o_pt, st));
    HIPCHK(h->o_lcam.upload(s.o_lcam, st));
    HIPCHK(h->ck_obs.upload(s.ck_obs, st));
    HIPCHK(h->ck_pt.upload(s.ck_pt, st));
    HIPCHK(h->ck_cam.upload(s.ck_cam, st));
    HIPCHK(h->ck_cam_list.upload(s.ck_cam_list, st));
    HIPCHK(h->ck_pair.upload(s.ck_pair, st));
    {
        std::vector<long long> pe(s.ck_pe.begin(), s.ck_pe.end());
        HIPCHK(h->ck_pe.upload(pe, st));
    }
    HIPCHK(h->pair_entries.upload(s.pair_entries, st));
    HIPCHK(h->ck_pgrp.upload(s.ck_pgrp, st));
    h->rw_nx = 0;
    h->rw_rows = h->rw_row_off = 0;
    if (s.rows || !s.rw_keys.empty()) {
        // rows mode: every observation has a row; chunk mode: the heavy points' observations (the last of the internal order)
        h->rw_row_off = s.rows ? 0 : (long long)s.hv_obs.front();
        h->rw_rows = (long long)s.n_obs - h->rw_row_off;
        std::vector<int32_t> xblk, diag((size_t)std::max(s.n_free, 1), -1);
        for (size_t k = 0; k < s.rw_keys.size(); ++k) {
            const int i = (int)(s.rw_keys[k] >> 32), j = (int)(s.rw_keys[k] & 0xffffffff);
            if (i == j) diag[i] = (int32_t)k;
            else xblk.push_back((int32_t)k);
        }
        h->rw_nx = (int)xblk.size();
        if (xblk.empty()) xblk.push_back(0);
        std::vector<long long> ptr(s.rw_ptr.begin(), s.rw_ptr.end());
        if (ptr.empty()) ptr.push_back(0);
        if (s.rw_ent.empty()) s.rw_ent.assign(2, 0);
        HIPCHK(h->rw_ent.upload(s.rw_ent, st));
        HIPCHK(h->rw_ptr.upload(ptr, st));
        HIPCHK(h->rw_xblk.upload(xblk, st));
        HIPCHK(h->rw_diag.upload(diag, st));
        HIPCHK(h->rows_buf.alloc_grow((size_t)std::max<long long>(h->rw_rows, 1) * ROW_DOUBLES));
    } else {
        h->rows_buf.release(); h->rw_ent.release(); h->rw_ptr.release(); h->rw_xblk.release(); h->rw_diag.release();
    }
    if (tm_on) { fprintf(stderr, "[create]   %-22s %7.2f ms%s\n", "index uploads", now_ms() - tl, s.rows ? "  (rows mode)" : ""); tl = now_ms(); }
    const int ea = alloc_work(h);
    if (tm_on) fprintf(stderr, "[create]   %-22s %7.2f ms\n", "work buffers", now_ms() - tl);
    return ea;
}

int create_impl(const BaProblem *prob, const BaOptions *opt, BaHandle *h) {
    const double t0 = now_ms();
    int e = validate(prob);
    if (e) return e;
    BaOptions o;
    if (opt) o = *opt;
    else ba_default_options(&o);
    if ((e = check_options(o))) return e;
    h->opt = o;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(BA_ERR_HIP, "no HIP device available");
    if (o.device >= 0) HIPCHK(hipSetDevice(o.device));
    HIPCHK(hipGetDevice(&h->device));
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    HIPCHK(hipEventCreate(&h->ev2));
    HIPCHK(hipEventCreate(&h->ev3));
    const bool tm_on = getenv("BA_TIME") != nullptr;
    double tl = now_ms();
    auto tick = [&](const char *what) { if (tm_on) { const double t = now_ms(); fprintf(stderr, "[create]   %-22s %7.2f ms\n", what, t - tl); tl = t; } };
    tick("device, stream, events");
    HostProblem hp;
    if ((e = fetch_problem(prob, hp, false))) return e;
    tick("fetch");
    StageScope staging(h->stream);
    if ((e = build_structure(h, hp, prob->n_cams, prob->n_points, prob->n_obs))) return e;
    tick("structure + uploads");
    if ((e = upload_values(h, hp))) return e;
    tick("values");
    h->t_setup_ms = now_ms() - t0;
    return BA_OK;
}

// ba_append, fast path: the grown problem differs from the handle's by new cameras, new points and new observations
// that only see points nobody saw before (what the reference's driver produces, Reconstruction.py:244-251).  Then the
// old chunks, their pair entries and their block slots are all still valid: only the new points are analysed (as a
// problem of their own), and their chunks are appended behind the old ones, on the host and in HBM.
// Returns BA_OK with *done = false when the pattern does not apply (the caller rebuilds everything).
int append_tail(BaHandle *h, const HostProblem &hp, int nc2, int np2, int64_t no2, bool *done) {
    *done = false;
    Symbolic &s = h->sym;
    const int64_t no1 = s.n_obs;
    if (s.n_heavy > 0 || no2 == no1 || s.rows) return BA_OK;   // (rows mode: the entry lists are global -- rebuilt)
    // points seen before (by the handle's observations)
    std::vector<uint8_t> seen((size_t)np2, 0);
    for (int i = 0; i < s.n_pts_obs; ++i) seen[s.pt_order[i]] = 1;
    std::vector<int32_t> local((size_t)np2, -1), tail_pts;
    for (int64_t k = no1; k < no2; ++k) {
        const int32_t p = hp.obs_pt[k];
        if (p < 0 || p >= np2 || hp.obs_cam[k] < 0 || hp.obs_cam[k] >= nc2) return fail(BA_ERR_INVALID, "observation index out of range");
        if (seen[p]) return BA_OK;   // an old track grew: its chunk changes
        if (local[p] < 0) { local[p] = 0; tail_pts.push_back(p); }
    }
    std::sort(tail_pts.begin(), tail_pts.end());
    for (size_t i = 0; i < tail_pts.size(); ++i) local[tail_pts[i]] = (int32_t)i;
    std::vector<int32_t> t_cam(hp.obs_cam.begin() + no1, hp.obs_cam.begin() + no2), t_pt((size_t)(no2 - no1));
    for (int64_t k = no1; k < no2; ++k) t_pt[k - no1] = local[hp.obs_pt[k]];
    Symbolic t;
    const char *msg = "";
    int e;
    if ((e = analyze_structure(nc2, (int)tail_pts.size(), no2 - no1, hp.cam_fixed.data(), t_cam.data(), t_pt.data(), t, &msg, s.deterministic)))
        return fail(e, "%s", msg);
    if (t.n_heavy > 0) return BA_OK;
    // ---- merge the tail into the handle's structure ----
    const int P1 = s.n_pts_chunked, nck1 = (int)s.ck_obs.size() - 1;
    const size_t pad = (size_t)(CHUNK_OBS / 2) * PAIR_RUN;
    const size_t pe1 = (size_t)s.ck_pe.back(), cl1 = s.ck_cam_list.size(), rk1 = s.raw_keys.size(), pg1 = s.ck_pgrp.size();
    s.n_cams = nc2; s.n_points = np2; s.n_obs = no2;
    s.cam_free = t.cam_free; s.free_cam = t.free_cam; s.n_free = t.n_free;
    {
        std::vector<int32_t> order;
        order.reserve((size_t)np2);
        order.insert(order.end(), s.pt_order.begin(), s.pt_order.begin() + P1);
        for (int i = 0; i < t.n_pts_chunked; ++i) order.push_back(tail_pts[t.pt_order[i]]);
        for (int i = 0; i < P1; ++i) seen[s.pt_order[i]] = 1;
        for (int32_t p : tail_pts) seen[p] = 1;
        for (int p = 0; p < np2; ++p)
            if (!seen[p]) order.push_back(p);   // unobserved points last, in index order
        s.pt_order.swap(order);
    }
    s.n_pts_chunked = s.n_pts_obs = P1 + t.n_pts_chunked;
    s.o_cam.insert(s.o_cam.end(), t.o_cam.begin(), t.o_cam.end());
    s.o_lcam.insert(s.o_lcam.end(), t.o_lcam.begin(), t.o_lcam.end());
    for (int32_t v : t.o_pt) s.o_pt.push_back(v + P1);
    for (int64_t v : t.obs_order) s.obs_order.push_back(no1 + v);
    for (size_t c = 1; c < t.ck_obs.size(); ++c) {
        s.ck_obs.push_back(t.ck_obs[c] + (int32_t)no1);
        s.ck_pt.push_back(t.ck_pt[c] + P1);
        s.ck_cam.push_back(t.ck_cam[c] + (int32_t)cl1);
        s.ck_pair.push_back(t.ck_pair[c] + (int32_t)rk1);
        s.ck_pe.push_back(t.ck_pe[c] + (int64_t)pe1);
    }
    s.ck_cam_list.insert(s.ck_cam_list.end(), t.ck_cam_list.begin(), t.ck_cam_list.end());
    s.raw_keys.insert(s.raw_keys.end(), t.raw_keys.begin(), t.raw_keys.end());
    s.ck_pair_slot.resize(s.raw_keys.size(), -1);
    s.ck_pgrp.insert(s.ck_pgrp.end(), t.ck_pgrp.begin(), t.ck_pgrp.end());
    s.pair_entries.resize(pe1);                                   // (drop the old read-ahead padding; the tail brings its own)
    s.pair_entries.insert(s.pair_entries.end(), t.pair_entries.begin(), t.pair_entries.end());
    s.n_pair_real += t.n_pair_real;
    s.n_direct_pairs += t.n_direct_pairs;
    s.n_direct_cam_obs += t.n_direct_cam_obs;
    s.max_track = std::max(s.max_track, t.max_track);
    s.max_chunk_cams = std::max(s.max_chunk_cams, t.max_chunk_cams);
    s.max_chunk_pairs = std::max(s.max_chunk_pairs, t.max_chunk_pairs);
    s.mean_track = s.n_pts_obs ? (double)no2 / s.n_pts_obs : 0.0;
    s.hv_obs.assign(1, no2);
    s.hv_pair.assign(1, 0);
    finalize_pattern(s, {});   // new keys only involve new cameras when the gauge is unchanged: old slots keep their numbers
    (void)pad;
    // ---- device: the old index arrays stay where they are, the tails go behind them ----
    h->n_chunks = (int)s.ck_obs.size() - 1;
    h->n_heavy = 0;
    HIPCHK(h->cam_free.upload(s.cam_free, h->stream));
    HIPCHK(h->free_cam.upload(s.free_cam, h->stream));
    HIPCHK(h->o_cam.append_at((size_t)no1, t.o_cam.data(), t.o_cam.size()));
    HIPCHK(h->o_pt.append_at((size_t)no1, s.o_pt.data() + no1, t.o_pt.size()));
    HIPCHK(h->o_lcam.append_at((size_t)no1, t.o_lcam.data(), t.o_lcam.size()));
    const size_t ntc = t.ck_obs.size() - 1;
    HIPCHK(h->ck_obs.append_at((size_t)nck1 + 1, s.ck_obs.data() + nck1 + 1, ntc));
    HIPCHK(h->ck_pt.append_at((size_t)nck1 + 1, s.ck_pt.data() + nck1 + 1, ntc));
    HIPCHK(h->ck_cam.append_at((size_t)nck1 + 1, s.ck_cam.data() + nck1 + 1, ntc));
    HIPCHK(h->ck_pair.append_at((size_t)nck1 + 1, s.ck_pair.data() + nck1 + 1, ntc));
    {
        std::vector<long long> pe(s.ck_pe.begin() + nck1 + 1, s.ck_pe.end());
        HIPCHK(h->ck_pe.append_at((size_t)nck1 + 1, pe.data(), pe.size()));
    }
    HIPCHK(h->ck_cam_list.append_at(cl1, t.ck_cam_list.data(), t.ck_cam_list.size()));
    HIPCHK(h->ck_pgrp.append_at(pg1, t.ck_pgrp.data(), t.ck_pgrp.size()));
    HIPCHK(h->pair_entries.append_at(pe1, t.pair_entries.data(), t.pair_entries.size()));
    if ((e = alloc_work(h))) return e;
    *done = true;
    return BA_OK;
}

void fill_summary(BaHandle *h, BaSummary *out, int status, double t_solve) {
    if (!out) return;
    memset(out, 0, sizeof *out);
    out->status = status;
    out->iterations = h->hstate.iter;
    out->accepted = h->hstate.n_accept;
    out->pcg_iterations = h->total_pcg;
    out->initial_cost = h->initial_cost;
    out->final_cost = h->hstate.F;
    out->grad_inf = h->hstate.gmax;
    out->lambda_final = h->hstate.lambda;
    out->t_setup_ms = h->t_setup_ms;
    out->t_solve_ms = t_solve;
    out->n_chunks = h->n_chunks;
    out->nnz_blocks = h->n_upper;
}

// residual sum at the current state (the reference's success quirk, BundleAdjustment.py:223-224)
int residual_sum(BaHandle *h, double *cost, double *rsum) {
    const Symbolic &s = h->sym;
    const long long n = s.n_obs;
    const int nb = (int)((n + 255) / 256);
    *cost = 0.0;
    *rsum = 0.0;
    if (nb == 0 && !h->coll) return BA_OK;
    DevBuf<double> bc, br, out;
    HIPCHK(bc.alloc(nb));
    HIPCHK(br.alloc(nb));
    HIPCHK(out.alloc(2));
    HIPCHK(hipMemsetAsync(out.p, 0, 2 * sizeof(double), h->stream));  // (a rank without observations still joins the all-reduce)
    if (nb > 0) hipLaunchKernelGGL(k_eval_cost<false>, dim3(nb), dim3(256), 0, h->stream, n, h->aux[h->cur].p, h->points[h->cur].p, h->o_cam.p,
                       h->o_pt.p, h->o_xy.p, h->opt.loss, 1.0 / (h->opt.f_scale * h->opt.f_scale),
                       0.5 * h->opt.f_scale * h->opt.f_scale, bc.p, br.p, 0, 0, (int *)nullptr);
    if (nb > 0) hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, h->stream, nb, 1, bc.p, out.p);
    if (nb > 0) hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, h->stream, nb, 1, br.p, out.p + 1);
    // point-sharded solve: cost and residual sum of the WHOLE problem on every rank (the reference's success rule
    // compares the two, BundleAdjustment.py:223-224; final_cost is all-reduced as well)
    int e = call_coll(h, out.p, 2, BA_REDUCE_SUM);
    if (e) return e;
    double host[2];
    HIPCHK(hipMemcpyAsync(host, out.p, sizeof host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *cost = host[0];
    *rsum = host[1];
    return BA_OK;
}

// every entry point that reads the handle's structure starts here
int usable(BaHandle *h) {
    if (!h) return fail(BA_ERR_INVALID, "null handle");
    if (h->poisoned) return fail(BA_ERR_INVALID, "the handle is unusable after a failed ba_append: call ba_append again (it rebuilds) or ba_destroy");
    return BA_OK;
}

}  // namespace

extern "C" {

int ba_version(void) { return 100; }
const char *ba_last_error(void) { return g_err.c_str(); }

void ba_default_options(BaOptions *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->loss = BA_LOSS_HUBER;
    o->max_iter = 100;
    o->f_scale = 1.0;
    o->ftol = o->xtol = o->gtol = 1e-8;
    o->lambda0 = 1e-4;
    o->pcg_tol = 0.0;   // adaptive forcing term (pcg_tol_now)
    o->pcg_max_iter = 500;
    o->linear_solver = BA_SOLVER_AUTO;
    o->verbose = 0;
    o->device = -1;
    o->curv_floor = 0.1;
}

BaHandle *ba_create(const BaProblem *prob, const BaOptions *opt) {
    BaHandle *h = new (std::nothrow) BaHandle();
    if (!h) {
        fail(BA_ERR_NOMEM, "out of host memory");
        return nullptr;
    }
    int e = BA_OK;
    try {
        e = create_impl(prob, opt, h);
    } catch (const std::bad_alloc &) {
        e = fail(BA_ERR_NOMEM, "out of host memory");
    } catch (...) {
        e = fail(BA_ERR_INVALID, "unexpected exception in ba_create");
    }
    if (e) {
        delete h;
        return nullptr;
    }
    return h;
}

void ba_destroy(BaHandle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
}

int ba_set_values(BaHandle *h, const BaProblem *prob, const BaOptions *opt) {
    if (int eu = usable(h)) return eu;
    const double t0 = now_ms();
    int e = validate(prob);
    if (e) return e;
    const Symbolic &s = h->sym;
    if (prob->n_cams != s.n_cams || prob->n_points != s.n_points || prob->n_obs != s.n_obs)
        return fail(BA_ERR_INVALID, "ba_set_values: sizes (%d cameras, %d points, %lld observations) differ from the handle's (%d, %d, %lld)",
                    prob->n_cams, prob->n_points, (long long)prob->n_obs, s.n_cams, s.n_points, (long long)s.n_obs);
    BaOptions o = h->opt;
    if (opt) {
        o = *opt;
        o.device = h->opt.device;
        o.schur_mode = h->opt.schur_mode;   // (part of the analysed structure)
        if ((e = check_options(o))) return e;
        if (o.deterministic != h->opt.deterministic)
            return fail(BA_ERR_INVALID, "ba_set_values: BaOptions.deterministic belongs to the analysed structure and cannot change on a handle");
    }
    (void)hipSetDevice(h->device);
    HIPCHK(hipStreamSynchronize(h->stream));
    StageScope staging(h->stream);
    HostProblem hp;
    try {
        if ((e = fetch_problem(prob, hp, false))) return e;
        // the structure the handle was analysed for must be the caller's: same free set, same (camera, point) per observation
        for (int c = 0; c < s.n_cams; ++c)
            if ((hp.cam_fixed[c] != 0) != (s.cam_free[c] < 0)) return fail(BA_ERR_INVALID, "ba_set_values: cam_fixed[%d] changed", c);
        for (int64_t k = 0; k < s.n_obs; ++k) {
            const int64_t src = s.obs_order[k];
            if (hp.obs_cam[src] != s.o_cam[k] || hp.obs_pt[src] != s.pt_order[s.o_pt[k]])
                return fail(BA_ERR_INVALID, "ba_set_values: observation %lld names another (camera, point) than at ba_create", (long long)src);
        }
        h->opt = o;
        if ((e = upload_values(h, hp))) return e;
        h->t_setup_ms = now_ms() - t0;  // what this solve paid before its first build
        return BA_OK;
    } catch (const std::bad_alloc &) {
        return fail(BA_ERR_NOMEM, "out of host memory");
    }
}

int ba_append(BaHandle *h, const BaProblem *grown, const BaOptions *opt, int32_t *mode) {
    if (!h) return fail(BA_ERR_INVALID, "null handle");
    if (mode) *mode = 0;
    const double t0 = now_ms();
    int e = validate(grown);
    if (e) return e;
    Symbolic &s = h->sym;
    if (!h->poisoned && (grown->n_cams < s.n_cams || grown->n_points < s.n_points || grown->n_obs < s.n_obs))
        return fail(BA_ERR_INVALID, "ba_append: the problem shrank (%d cameras, %d points, %lld observations against the handle's %d, %d, %lld)",
                    grown->n_cams, grown->n_points, (long long)grown->n_obs, s.n_cams, s.n_points, (long long)s.n_obs);
    BaOptions o = h->opt;
    if (opt) {
        o = *opt;
        o.device = h->opt.device;
        o.schur_mode = h->opt.schur_mode;
        if ((e = check_options(o))) return e;
        if (o.deterministic != h->opt.deterministic)
            return fail(BA_ERR_INVALID, "ba_append: BaOptions.deterministic belongs to the analysed structure and cannot change on a handle");
    }
    (void)hipSetDevice(h->device);
    HIPCHK(hipStreamSynchronize(h->stream));
    StageScope staging(h->stream);
    try {
        HostProblem hp;
        if ((e = fetch_problem(grown, hp, false))) return e;
        // the handle's problem must be a prefix of the grown one: same gauge for the old cameras, same (camera, point) for
        // every old observation
        bool prefix = !h->poisoned;
        for (int c = 0; c < s.n_cams && prefix; ++c) prefix = (hp.cam_fixed[c] != 0) == (s.cam_free[c] < 0);
        for (int64_t k = 0; k < s.n_obs && prefix; ++k) {
            const int64_t src = s.obs_order[k];
            prefix = hp.obs_cam[src] == s.o_cam[k] && hp.obs_pt[src] == s.pt_order[s.o_pt[k]];
        }
        // From here on the handle's structure is being replaced: an error (or an exception) half way leaves the host
        // tables ahead of the device arrays.  The flag makes every later call on the handle fail cleanly -- except
        // another ba_append, which then rebuilds the whole structure -- instead of launching over inconsistent tables.
        const bool was_poisoned = h->poisoned;
        h->poisoned = true;
        h->built = false;
        bool appended = false;
        if (prefix && !was_poisoned && (e = append_tail(h, hp, grown->n_cams, grown->n_points, grown->n_obs, &appended))) return e;
        if (!appended && (e = build_structure(h, hp, grown->n_cams, grown->n_points, grown->n_obs))) return e;
        h->opt = o;
        if ((e = upload_values(h, hp))) return e;
        h->poisoned = false;
        // (a sharded handle: this rank's own pattern has changed, and so may the others' -- the union has to be made again)
        h->pattern_stale = h->coll != nullptr;
        if (mode) *mode = appended ? 1 : 0;
        h->t_setup_ms = now_ms() - t0;
        return BA_OK;
    } catch (const std::bad_alloc &) {
        return fail(BA_ERR_NOMEM, "out of host memory");
    } catch (...) {
        return fail(BA_ERR_INVALID, "unexpected exception in ba_append");
    }
}

static int bcr_plan_serialise(int32_t n_free, const int32_t *pairs, int64_t n_pairs, bool allow_rest, int32_t *out, int64_t capacity,
                              int64_t *needed) {
    if (!needed || n_pairs < 0 || (n_pairs > 0 && !pairs) || n_free < 0) return fail(BA_ERR_INVALID, "bad argument");
    try {
        std::vector<int64_t> keys;
        keys.reserve((size_t)n_pairs);
        for (int64_t i = 0; i < n_pairs; ++i) {
            int a = pairs[2 * i], b = pairs[2 * i + 1];
            if (a < 0 || b < 0 || a >= n_free || b >= n_free) return fail(BA_ERR_INVALID, "pattern pair out of range");
            if (a > b) std::swap(a, b);
            keys.push_back(((int64_t)a << 32) | (uint32_t)b);
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        BcrPlan P;
        const bool ok = bcr_make_plan(n_free, keys.data(), keys.size(), P, allow_rest);
        std::vector<int32_t> w = {ok ? 1 : 0, (P.cyclic ? 1 : 0) | (P.reordered ? 2 : 0) | (P.n_rest > 0 ? 4 : 0), P.m, P.nt, P.nb, P.n_slots,
                                  P.n_init_slots, ok ? P.levels() : 0};
        if (ok) {
            // src refers to the position of a pair in the SORTED UNIQUE list = the block order of S
            w.insert(w.end(), P.blk_start.begin(), P.blk_start.end());
            w.insert(w.end(), P.pos_f.begin(), P.pos_f.end());
            w.insert(w.end(), P.src.begin(), P.src.end());
            w.insert(w.end(), P.lvl_elim.begin(), P.lvl_elim.end());
            w.insert(w.end(), P.lvl_op.begin(), P.lvl_op.end());
            for (const BcrElim &e : P.elims) w.insert(w.end(), {e.k, e.gu, e.gv, e.u, e.v});
            for (const BcrOp &o : P.ops) w.insert(w.end(), {o.kind, o.dst, o.a0, o.b0, o.a1, o.b1, o.k0, o.k1, o.init});
            w.push_back((int32_t)std::min<int64_t>(P.n_rest, INT32_MAX));
            w.insert(w.end(), P.pend_ptr.begin(), P.pend_ptr.end());
            w.insert(w.end(), P.pend.begin(), P.pend.end());
            w.insert(w.end(), P.lvl_fwd.begin(), P.lvl_fwd.end());
            w.insert(w.end(), P.fwd.begin(), P.fwd.end());
        }
        *needed = (int64_t)w.size();
        if (out) {
            if (capacity < (int64_t)w.size()) return fail(BA_ERR_INVALID, "plan buffer too small");
            std::copy(w.begin(), w.end(), out);
        }
        return BA_OK;
    } catch (const std::bad_alloc &) {
        return fail(BA_ERR_NOMEM, "out of host memory");
    }
}

int ba_bcr_plan(int32_t n_free, const int32_t *pairs, int64_t n_pairs, int32_t *out, int64_t capacity, int64_t *needed) {
    return bcr_plan_serialise(n_free, pairs, n_pairs, false, out, capacity, needed);
}
int ba_bcr_plan_rest(int32_t n_free, const int32_t *pairs, int64_t n_pairs, int32_t *out, int64_t capacity, int64_t *needed) {
    return bcr_plan_serialise(n_free, pairs, n_pairs, true, out, capacity, needed);
}

int ba_get_pattern(BaHandle *h, int32_t *out_pairs, int64_t capacity, int64_t *n_pairs) {
    if (!h || !n_pairs) return fail(BA_ERR_INVALID, "null argument");
    const auto &k = h->sym.upper_keys;
    *n_pairs = (int64_t)k.size();
    if (out_pairs) {
        if (capacity < (int64_t)k.size()) return fail(BA_ERR_INVALID, "pattern buffer too small");
        for (size_t i = 0; i < k.size(); ++i) {
            out_pairs[2 * i] = (int32_t)(k[i] >> 32);
            out_pairs[2 * i + 1] = (int32_t)(k[i] & 0xffffffff);
        }
    }
    return BA_OK;
}

int ba_set_pattern(BaHandle *h, const int32_t *pairs, int64_t n_pairs) {
    if (!h || (n_pairs > 0 && !pairs)) return fail(BA_ERR_INVALID, "null argument");
    if (int eu = usable(h)) return eu;
    if (n_pairs < 0) return fail(BA_ERR_INVALID, "negative pair count");
    try {
        std::vector<int64_t> extra;
        extra.reserve((size_t)n_pairs);
        for (int64_t i = 0; i < n_pairs; ++i) {
            int a = pairs[2 * i], b = pairs[2 * i + 1];
            if (a < 0 || b < 0 || a >= h->sym.n_free || b >= h->sym.n_free) return fail(BA_ERR_INVALID, "pattern pair out of range");
            if (a > b) std::swap(a, b);
            extra.push_back(((int64_t)a << 32) | (uint32_t)b);
        }
        finalize_pattern(h->sym, extra);
        (void)hipSetDevice(h->device);
        h->built = false;
        h->pattern_stale = false;
        return upload_pattern(h);
    } catch (const std::bad_alloc &) {
        return fail(BA_ERR_NOMEM, "out of host memory");
    } catch (...) {
        return fail(BA_ERR_INVALID, "unexpected exception in ba_set_pattern");
    }
}

int ba_set_collective(BaHandle *h, BaCollectiveFn fn, void *user, int32_t rank, int32_t world) {
    if (!h) return fail(BA_ERR_INVALID, "null handle");
    if (world > BA_MAX_WORLD || rank < 0 || (world >= 1 && rank >= world)) return fail(BA_ERR_INVALID, "rank %d of %d (at most %d ranks)", rank, world, BA_MAX_WORLD);
    if (h->rccl_comm) {   // a callback replaces the built-in RCCL collectives
        if (RcclApi *api = rccl_api()) (void)api->CommDestroy(h->rccl_comm);
        h->rccl_comm = nullptr;
    }
    h->coll = fn;
    h->coll_user = user;
    h->rank = rank;
    h->world = world < 1 ? 1 : world;
    return BA_OK;
}

int ba_rccl_unique_id(void *out_id128) {
    if (!out_id128) return fail(BA_ERR_INVALID, "null argument");
    RcclApi *api = rccl_api();
    if (!api) return fail(BA_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    RcclUniqueId id;
    const int rc = api->GetUniqueId(&id);
    if (rc != 0) return fail(BA_ERR_COMM, "ncclGetUniqueId failed: %s", api->GetErrorString ? api->GetErrorString(rc) : "?");
    memcpy(out_id128, &id, sizeof id);
    return BA_OK;
}

namespace {
int sentinel_collective(void *, int64_t, int32_t, void *) { return 1; }  // never called: call_coll takes the RCCL path first
}

int ba_rccl_init(BaHandle *h, const void *id128, int32_t rank, int32_t world) {
    if (!h || !id128) return fail(BA_ERR_INVALID, "null argument");
    if (world < 1 || world > BA_MAX_WORLD || rank < 0 || rank >= world) return fail(BA_ERR_INVALID, "rank %d of %d (at most %d ranks)", rank, world, BA_MAX_WORLD);
    RcclApi *api = rccl_api();
    if (!api) return fail(BA_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    (void)hipSetDevice(h->device);
    if (h->rccl_comm) {
        (void)api->CommDestroy(h->rccl_comm);
        h->rccl_comm = nullptr;
    }
    RcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    void *comm = nullptr;
    const int rc = api->CommInitRank(&comm, world, id, rank);
    if (rc != 0 || !comm) return fail(BA_ERR_COMM, "ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(rc) : "?");
    h->rccl_comm = comm;
    h->coll = sentinel_collective;   // "this solve is sharded" for the code that only asks whether collectives are on
    h->coll_user = nullptr;
    h->coll_stream_ordered = true;
    h->rank = rank;
    h->world = world;
    return BA_OK;
}

int ba_set_collective_mode(BaHandle *h, int32_t stream_ordered) {
    if (!h) return fail(BA_ERR_INVALID, "null handle");
    h->coll_stream_ordered = stream_ordered != 0;
    return BA_OK;
}

int ba_get_stream(BaHandle *h, void **stream) {
    if (!h || !stream) return fail(BA_ERR_INVALID, "null argument");
    *stream = (void *)h->stream;
    return BA_OK;
}

int ba_build(BaHandle *h, double lambda) {
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    int e = set_lambda(h, lambda);
    if (e) return e;
    return do_build(h, false);
}

int ba_build_bench(BaHandle *h, double lambda, int32_t warmup, int32_t reps, double *ms_per_pass, double *ms_fused_kernel) {
    if (!h || reps <= 0) return fail(BA_ERR_INVALID, "bad argument");
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    int e = set_lambda(h, lambda);
    if (e) return e;
    for (int i = 0; i < warmup; ++i)
        if ((e = do_build(h, false))) return e;
    HIPCHK(hipStreamSynchronize(h->stream));
    // pass A: whole build passes back to back (no host syncs inside the bracket)
    HIPCHK(hipEventRecord(h->ev2, h->stream));
    for (int i = 0; i < reps; ++i)
        if ((e = do_build(h, false))) return e;
    HIPCHK(hipEventRecord(h->ev3, h->stream));
    HIPCHK(hipEventSynchronize(h->ev3));
    float total = 0.f;
    HIPCHK(hipEventElapsedTime(&total, h->ev2, h->ev3));
    // pass B (optional): the fused kernel alone, events directly around its launch
    double kern_ms = 0.0;
    for (int i = 0; ms_fused_kernel && i < reps; ++i) {
        if ((e = do_build(h, true))) return e;
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        kern_ms += ms;
    }
    if (ms_per_pass) *ms_per_pass = total / reps;
    if (ms_fused_kernel) *ms_fused_kernel = kern_ms / reps;
    return BA_OK;
}

int ba_run(BaHandle *h, int32_t max_steps, BaSummary *out) {
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    const double t0 = now_ms();
    int e;
    if ((e = do_build(h, false))) return e;
    if ((e = read_state(h))) return e;
    if (std::isnan(h->initial_cost)) h->initial_cost = h->hstate.F;
    if (!std::isfinite(h->hstate.F)) {
        fill_summary(h, out, BA_ERR_NUMERIC, now_ms() - t0);
        return fail(BA_ERR_NUMERIC, "non-finite initial cost");
    }
    // verbose = 2 is what the reference passes (least_squares(verbose=2), BundleAdjustment.py:219): scipy's own table --
    // print_header_nonlinear / print_iteration_nonlinear of scipy/optimize/_lsq/common.py, one row for the starting point
    // and one per ACCEPTED step (scipy's iterations; nfev counts every trial step) -- and its two closing lines.
    // verbose >= 3: the engine's table (every trial step, damping, PCG iterations) and collective traces.
    const int verbose = h->opt.verbose;
    auto centred = [](char *dst, const char *fmt, double v) {   // Python's '{:^15...}': the odd blank goes to the right
        char num[32];
        const int n = snprintf(num, sizeof num, fmt, v);
        const int left = n < 15 ? (15 - n) / 2 : 0;
        memset(dst, ' ', 15);
        memcpy(dst + left, num, (size_t)(n < 15 - left ? n : 15 - left));
        dst[15] = 0;
    };
    int scipy_iter = 0;
    auto scipy_row = [&](int nfev, double cost, bool first, double reduction, double step_norm, double optimality) {
        char c0[16], c1[16], c2[16], c3[16], c4[16], c5[16];
        centred(c0, "%.0f", (double)scipy_iter);
        centred(c1, "%.0f", (double)nfev);
        centred(c2, "%.4e", cost);
        if (first) { memset(c3, ' ', 15); c3[15] = 0; memset(c4, ' ', 15); c4[15] = 0; }
        else { centred(c3, "%.2e", reduction); centred(c4, "%.2e", step_norm); }
        centred(c5, "%.2e", optimality);
        printf("%s%s%s%s%s%s\n", c0, c1, c2, c3, c4, c5);
        ++scipy_iter;
    };
    if (verbose == 2) {
        printf("   Iteration     Total nfev        Cost      Cost reduction    Step norm     Optimality   \n");
        scipy_row(1, h->hstate.F, true, 0.0, 0.0, h->hstate.gmax);
    } else if (verbose >= 3) {
        printf("   Iteration     Total nfev        Cost      Cost reduction    Step norm     Optimality      lambda   pcg\n"
               "%12d %12d %16.4e %46.2e\n", 0, 1, h->hstate.F, h->hstate.gmax);
    }
    int status = 0;
    // (sharded solve: the point count is this rank's, and a rank that left the loop alone would leave its peers in a
    // collective; there gmax -- all-reduced -- ends an empty problem on every rank at once)
    const int n_unknown = h->coll ? 1 : h->sym.n_free * 6 + h->sym.n_pts_obs * 3;
    // One host round trip per step (the accept / reject decision, which selects the state buffers of the next launches).
    // The state is NOT read back after the build that follows: the gradient test of that build is made on the device by
    // the next step's k_decide (terminate = 1, step void), the host's copy of F / gmax is refreshed at the end.
    bool fresh = true;   // hstate holds the outcome of the latest build
    for (int step = 0; step < max_steps; ++step) {
        if ((fresh && h->hstate.gmax < h->opt.gtol) || n_unknown == 0) {
            status = 1;
            break;
        }
        const int pcg_before = h->total_pcg;
        if ((e = do_step(h))) return e;
        const LmState &s = h->hstate;
        if (s.terminate == 1 && !s.accept && s.solve_info == 0) {   // the build before this step had already converged (gtol)
            status = 1;
            fresh = true;   // (nothing was built since the state was read)
            break;
        }
        // (scipy's row shows the gradient AT the new iterate: printed once the build that follows has been read back)
        const bool row = verbose == 2 && s.accept;
        const int row_nfev = s.iter + 1;
        const double row_dF = s.dF, row_step = s.step_norm;
        if (verbose >= 3)
            printf("%12d %12d %16.4e %16.2e %14.2e %14.2e %11.1e %5d %s\n", s.iter, s.iter + 1, s.F, s.accept ? s.dF : 0.0,
                   s.step_norm, s.gmax, s.lambda, h->total_pcg - pcg_before, s.accept ? "" : "(rejected)");
        if (s.terminate) {
            status = s.terminate;
            // the last build belongs to the pre-step state: refresh F / gradient if the step was taken
            if (s.accept && (e = do_build(h, false))) return e;
            if (s.accept && (e = read_state(h))) return e;
            fresh = true;
            if (row) scipy_row(row_nfev, h->hstate.F, false, row_dF, row_step, h->hstate.gmax);
            break;
        }
        if ((e = do_build(h, false))) return e;  // new linearisation (accepted) or new damping (rejected)
        fresh = false;
        if (row) {   // one more read-back per accepted step, in this mode only; the loop's own state is left as it was
            const LmState keep = h->hstate;
            if ((e = read_state(h))) return e;
            scipy_row(row_nfev, h->hstate.F, false, row_dF, row_step, h->hstate.gmax);
            h->hstate = keep;
        }
    }
    if (!fresh && (e = read_state(h))) return e;
    fill_summary(h, out, status, now_ms() - t0);
    if (out) {
        double c, rs;
        if ((e = residual_sum(h, &c, &rs))) return e;
        out->residual_sum = rs;
    }
    if (verbose == 2) {
        // least_squares' closing lines (scipy/optimize/_lsq/least_squares.py: TERMINATION_MESSAGES, then the summary)
        static const char *const msg[5] = {"The maximum number of function evaluations is exceeded.", "`gtol` termination condition is satisfied.",
                                           "`ftol` termination condition is satisfied.", "`xtol` termination condition is satisfied.",
                                           "Both `ftol` and `xtol` termination conditions are satisfied."};
        printf("%s\nFunction evaluations %d, initial cost %.4e, final cost %.4e, first-order optimality %.2e.\n",
               msg[status >= 0 && status <= 4 ? status : 0], h->hstate.iter + 1, h->initial_cost, h->hstate.F, h->hstate.gmax);
    }
    if (verbose == 1 || verbose >= 3)
        printf("[ba_hip] status %d after %d steps (%d accepted, %d PCG iterations): cost %.6e -> %.6e, |g|inf %.2e\n", status,
               h->hstate.iter, h->hstate.n_accept, h->total_pcg, h->initial_cost, h->hstate.F, h->hstate.gmax);
    return BA_OK;
}

int ba_get_state(BaHandle *h, double *cam_pose, double *points) {
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    const Symbolic &s = h->sym;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (cam_pose && s.n_cams)
        HIPCHK(hipMemcpy(cam_pose, h->pose[h->cur].p, (size_t)s.n_cams * 6 * sizeof(double), hipMemcpyDeviceToHost));
    if (points && s.n_points) {
        std::vector<double> tmp((size_t)s.n_points * 3);
        HIPCHK(hipMemcpy(tmp.data(), h->points[h->cur].p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < s.n_points; ++i)
            for (int k = 0; k < 3; ++k) points[3 * (size_t)s.pt_order[i] + k] = tmp[3 * (size_t)i + k];
    }
    return BA_OK;
}

int ba_get_cost(BaHandle *h, double *cost) {
    if (!h || !cost) return fail(BA_ERR_INVALID, "null argument");
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    double rs;
    return residual_sum(h, cost, &rs);
}

int ba_get_reduced_blocks(BaHandle *h, int64_t *keys, double *blocks, int64_t capacity, int64_t *n_blocks) {
    if (!h || !n_blocks) return fail(BA_ERR_INVALID, "null argument");
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    if (!h->built) return fail(BA_ERR_INVALID, "ba_build has not run");
    const Symbolic &s = h->sym;
    *n_blocks = (int64_t)s.upper_keys.size();
    if (!keys && !blocks) return BA_OK;
    if (capacity < *n_blocks) return fail(BA_ERR_INVALID, "capacity %lld < %lld blocks", (long long)capacity, (long long)*n_blocks);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (keys) std::copy(s.upper_keys.begin(), s.upper_keys.end(), keys);
    if (blocks && *n_blocks) HIPCHK(hipMemcpy(blocks, h->S, (size_t)*n_blocks * 36 * sizeof(double), hipMemcpyDeviceToHost));
    return BA_OK;
}

int ba_get_reduced_system(BaHandle *h, double *s_out, double *g_red, double *u, double *g_c, double *v, double *g_p) {
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    const Symbolic &s = h->sym;
    if ((v || g_p) && h->dbg_v.n == 0) {
        // probes requested: allocate the debug arrays and rebuild at the same lambda
        HIPCHK(h->dbg_v.alloc((size_t)s.n_pts_obs * 6 + 1));
        HIPCHK(h->dbg_gp.alloc((size_t)s.n_pts_obs * 3 + 1));
        int e = do_build(h, false);
        if (e) return e;
    }
    if (!h->built) return fail(BA_ERR_INVALID, "ba_build has not run");
    HIPCHK(hipStreamSynchronize(h->stream));
    const int nf = s.n_free;
    const size_t n6 = (size_t)nf * 6;
    if (s_out) {
        std::vector<double> blocks((size_t)h->n_upper * 36);
        if (!blocks.empty()) HIPCHK(hipMemcpy(blocks.data(), h->S, blocks.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::fill(s_out, s_out + n6 * n6, 0.0);
        for (size_t k = 0; k < s.upper_keys.size(); ++k) {
            const int i = (int)(s.upper_keys[k] >> 32), j = (int)(s.upper_keys[k] & 0xffffffff);
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double val = blocks[k * 36 + a * 6 + b];
                    s_out[((size_t)i * 6 + a) * n6 + (size_t)j * 6 + b] = val;
                    if (i != j) s_out[((size_t)j * 6 + b) * n6 + (size_t)i * 6 + a] = val;
                }
        }
    }
    if (g_red && n6) HIPCHK(hipMemcpy(g_red, h->gred, n6 * sizeof(double), hipMemcpyDeviceToHost));
    if (u || g_c) {
        std::vector<double> hu((size_t)nf * 21), hg(n6);
        if (nf) {
            HIPCHK(hipMemcpy(hu.data(), h->U, hu.size() * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hg.data(), h->gc, hg.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        if (u) std::fill(u, u + (size_t)s.n_cams * 36, 0.0);
        if (g_c) std::fill(g_c, g_c + (size_t)s.n_cams * 6, 0.0);
        for (int f = 0; f < nf; ++f) {
            const int c = s.free_cam[f];
            for (int a = 0; a < 6; ++a) {
                if (g_c) g_c[(size_t)c * 6 + a] = hg[(size_t)f * 6 + a];
                if (u)
                    for (int b = a; b < 6; ++b) {
                        const double val = hu[(size_t)f * 21 + ut6(a, b)];
                        u[(size_t)c * 36 + a * 6 + b] = val;
                        u[(size_t)c * 36 + b * 6 + a] = val;
                    }
            }
        }
    }
    if (v || g_p) {
        std::vector<double> hv((size_t)s.n_pts_obs * 6), hg((size_t)s.n_pts_obs * 3);
        if (s.n_pts_obs) {
            HIPCHK(hipMemcpy(hv.data(), h->dbg_v.p, hv.size() * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hg.data(), h->dbg_gp.p, hg.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        if (v) std::fill(v, v + (size_t)s.n_points * 9, 0.0);
        if (g_p) std::fill(g_p, g_p + (size_t)s.n_points * 3, 0.0);
        static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
        for (int i = 0; i < s.n_pts_obs; ++i) {
            const size_t p = (size_t)s.pt_order[i];
            for (int a = 0; a < 3; ++a) {
                if (g_p) g_p[p * 3 + a] = hg[(size_t)i * 3 + a];
                if (v)
                    for (int b = 0; b < 3; ++b) v[p * 9 + a * 3 + b] = hv[(size_t)i * 6 + idx[a][b]];
            }
        }
    }
    return BA_OK;
}

// Diagnostic only (not in include/ba_hip.h): one build pass with the stamped kernel variant; out gets
// 12 s_memtime values per chunk.  Used by tools/stamps.py to see where a chunk spends its cycles.
int ba_debug_stamps(BaHandle *h, unsigned long long *out, long long capacity) {
    if (!h || !out) return fail(BA_ERR_INVALID, "null argument");
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    if (capacity < (long long)h->n_chunks * 12) return fail(BA_ERR_INVALID, "stamp buffer too small");
    if (h->opt.deterministic || h->sym.rows) return fail(BA_ERR_UNSUPPORTED, "no stamped variant of the deterministic build or of the rows mode");
    HIPCHK(h->stamps.alloc((size_t)h->n_chunks * 12));
    HIPCHK(hipMemset(h->stamps.p, 0, (size_t)h->n_chunks * 12 * 8));
    int e = do_build(h, false);
    if (e == BA_OK && (e = do_build(h, false)) == BA_OK) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(out, h->stamps.p, (size_t)h->n_chunks * 12 * 8, hipMemcpyDeviceToHost));
    }
    h->stamps.release();
    return e;
}

// Diagnostic only (not in include/ba_hip.h): the cyclic reduction's preconditioner path against its direct solve.
// After a build on a pattern with an exact schedule: x = the fused factor + solve of S x = -g, then the explicit inverses
// and z = M^-1 (-g) through the sweeps (k_bcr_inv, k_bcr_fwd, k_bcr_backx).  *rel = max |x - z| / max |x| (0 when equal).
int ba_debug_bcr_apply(BaHandle *h, double *rel) {
    if (!h || !rel) return fail(BA_ERR_INVALID, "null argument");
    if (int eu = usable(h)) return eu;
    (void)hipSetDevice(h->device);
    if (!h->built) return fail(BA_ERR_INVALID, "ba_build has not run");
    int e;
    if ((e = bcr_plan_for(h))) return e;
    if (h->bcr_state != 1) return fail(BA_ERR_UNSUPPORTED, "the pattern has no cyclic-reduction schedule");
    const int n = h->sym.n_free * 6;
    const BcrDev d = bcr_dev(h);
    HIPCHK(bcr_solve_launch(h->stream, h->bcr_plan, d, h->S, h->gred, h->px.p));
    HIPCHK(bcr_invert_launch(h->stream, h->bcr_plan, d));
    hipLaunchKernelGGL(k_negate, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, h->gred, h->pr.p);
    HIPCHK(bcr_apply_launch(h->stream, h->bcr_plan, d, h->pr.p, h->pz.p, nullptr));
    std::vector<double> x(n), z(n);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(x.data(), h->px.p, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(z.data(), h->pz.p, n * sizeof(double), hipMemcpyDeviceToHost));
    double dmax = 0.0, xmax = 0.0;
    for (int i = 0; i < n; ++i) { dmax = std::max(dmax, std::fabs(x[i] - z[i])); xmax = std::max(xmax, std::fabs(x[i])); }
    *rel = xmax > 0.0 ? dmax / xmax : dmax;
    return BA_OK;
}

int ba_synchronize(BaHandle *h) {
    if (!h) return fail(BA_ERR_INVALID, "null handle");
    (void)hipSetDevice(h->device);
    HIPCHK(hipStreamSynchronize(h->stream));
    return BA_OK;
}

int ba_solve(BaProblem *prob, const BaOptions *opt, BaSummary *out) {
    BaOptions o;
    if (opt) o = *opt;
    else ba_default_options(&o);
    BaHandle *h = ba_create(prob, &o);
    if (!h) {
        if (out) {
            memset(out, 0, sizeof *out);
            out->status = BA_ERR_INVALID;
        }
        return g_code ? g_code : BA_ERR_INVALID;
    }
    int e = ba_run(h, o.max_iter, out);
    if (e == BA_OK) {
        const Symbolic &s = h->sym;
        std::vector<double> cp((size_t)s.n_cams * 6), pts((size_t)s.n_points * 3);
        e = ba_get_state(h, cp.data(), pts.data());
        if (e == BA_OK) {
            const hipMemcpyKind kind = prob->memory == BA_MEM_HOST ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
            if (!cp.empty() && hipMemcpy(prob->cam_pose, cp.data(), cp.size() * sizeof(double), kind) != hipSuccess)
                e = fail(BA_ERR_HIP, "copy-back of cam_pose failed");
            if (e == BA_OK && !pts.empty() && hipMemcpy(prob->points, pts.data(), pts.size() * sizeof(double), kind) != hipSuccess)
                e = fail(BA_ERR_HIP, "copy-back of points failed");
        }
    }
    ba_destroy(h);
    return e;
}

int ba_warm_up(void) {
    // One tiny solve: the first call into a fresh process pays for the HIP runtime, the code objects of every kernel on
    // the reference-sized path (chunk kernels, one-workgroup dense solver, reductions) and the page-locked staging buffer
    // -- some 50 ms that would otherwise land on the caller's first inter_ba.
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(BA_ERR_HIP, "no HIP device available");
    const int nc = 4, np_ = 12;
    std::vector<double> pose(nc * 6, 0.0), intr(nc * 4), pts(np_ * 3), xy;
    std::vector<uint8_t> fixed(nc, 0);
    std::vector<int32_t> oc, op;
    fixed[0] = fixed[1] = 1;
    for (int c = 0; c < nc; ++c) {
        pose[c * 6 + 3] = -0.5 * c;   // cameras along x, looking down +z
        intr[c * 4] = intr[c * 4 + 1] = 1000.0; intr[c * 4 + 2] = 500.0; intr[c * 4 + 3] = 400.0;
    }
    for (int p = 0; p < np_; ++p) {
        pts[p * 3] = 0.3 * (p % 4); pts[p * 3 + 1] = 0.2 * (p / 4) - 0.2; pts[p * 3 + 2] = 5.0 + 0.1 * p;
        for (int c = 0; c < nc; ++c) {
            const double x = pts[p * 3] + pose[c * 6 + 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
            oc.push_back(c); op.push_back(p);
            xy.push_back(1000.0 * x / z + 500.0 + 0.1 * ((p + c) % 3 - 1));
            xy.push_back(1000.0 * y / z + 400.0 - 0.1 * ((p * c) % 3 - 1));
        }
    }
    BaProblem pr{};
    pr.n_cams = nc; pr.n_points = np_; pr.n_obs = (int64_t)oc.size();
    pr.cam_pose = pose.data(); pr.cam_intr = intr.data(); pr.cam_fixed = fixed.data(); pr.points = pts.data();
    pr.obs_cam = oc.data(); pr.obs_pt = op.data(); pr.obs_xy = xy.data(); pr.memory = BA_MEM_HOST;
    BaOptions o;
    ba_default_options(&o);
    o.max_iter = 3;
    BaSummary s;
    return ba_solve(&pr, &o, &s);
}

int ba_analyze(const BaProblem *prob, BaAnalysis *out) {
    int e = validate(prob);
    if (e) return e;
    if (!out) return fail(BA_ERR_INVALID, "null output");
    HostProblem hp;
    if ((e = fetch_problem(prob, hp, true))) return e;
    Symbolic s;
    const char *msg = "";
    try {
        if ((e = analyze_structure(prob->n_cams, prob->n_points, prob->n_obs, hp.cam_fixed.data(), hp.obs_cam.data(),
                                   hp.obs_pt.data(), s, &msg)))
            return fail(e, "%s", msg);
        finalize_pattern(s, {});
    } catch (const std::bad_alloc &) {
        return fail(BA_ERR_NOMEM, "out of host memory");
    }
    memset(out, 0, sizeof *out);
    out->n_chunks = (int64_t)s.ck_obs.size() - 1;
    out->n_pair_entries = s.n_pair_real;
    out->nnz_blocks_upper = (int64_t)s.upper_keys.size();
    out->nnz_blocks_full = (int64_t)s.bsr_col.size();
    out->n_direct_pairs = s.n_direct_pairs;
    out->n_direct_cam_obs = s.n_direct_cam_obs;
    out->n_free_cams = s.n_free;
    out->n_observed_points = s.n_pts_obs;
    out->max_track = s.max_track;
    out->n_heavy_points = s.n_heavy;
    out->max_chunk_cams = s.max_chunk_cams;
    out->max_chunk_pairs = s.max_chunk_pairs;
    out->mean_track = s.mean_track;
    out->n_groups = s.n_groups;
    out->auto_rows_mode = wants_rows_mode(s) ? 1 : 0;
    return BA_OK;
}

// ---- one-shot evaluation entry points (caller's observation order) ---------------------------
namespace {
struct EvalCtx {
    DevBuf<double> pose, intr, points, xy, aux;
    DevBuf<int> ocam, opt;
    const double *d_points = nullptr, *d_xy = nullptr;
    const int *d_ocam = nullptr, *d_opt = nullptr;
};

int eval_setup(const BaProblem *p, EvalCtx &c, bool need_xy) {
    int e = validate(p);
    if (e) return e;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(BA_ERR_HIP, "no HIP device available");
    // index validation on the host (a bad index would fault the kernel)
    {
        std::vector<int32_t> oc, op;
        if ((e = fetch(oc, p->obs_cam, (size_t)p->n_obs, p->memory))) return e;
        if ((e = fetch(op, p->obs_pt, (size_t)p->n_obs, p->memory))) return e;
        for (int64_t k = 0; k < p->n_obs; ++k)
            if (oc[k] < 0 || oc[k] >= p->n_cams || op[k] < 0 || op[k] >= p->n_points)
                return fail(BA_ERR_INVALID, "observation index out of range");
    }
    const double *d_pose, *d_intr;
    if (p->memory == BA_MEM_HOST) {
        std::vector<double> tmp;
        auto up = [&](DevBuf<double> &b, const double *src, size_t n) -> hipError_t {
            hipError_t er = b.alloc(n);
            if (er != hipSuccess || n == 0) return er;
            return hipMemcpy(b.p, src, n * sizeof(double), hipMemcpyHostToDevice);
        };
        HIPCHK(up(c.pose, p->cam_pose, (size_t)p->n_cams * 6));
        HIPCHK(up(c.intr, p->cam_intr, (size_t)p->n_cams * 4));
        HIPCHK(up(c.points, p->points, (size_t)p->n_points * 3));
        if (need_xy) HIPCHK(up(c.xy, p->obs_xy, (size_t)p->n_obs * 2));
        HIPCHK(c.ocam.alloc((size_t)p->n_obs));
        HIPCHK(c.opt.alloc((size_t)p->n_obs));
        if (p->n_obs) {
            HIPCHK(hipMemcpy(c.ocam.p, p->obs_cam, (size_t)p->n_obs * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(c.opt.p, p->obs_pt, (size_t)p->n_obs * sizeof(int), hipMemcpyHostToDevice));
        }
        d_pose = c.pose.p; d_intr = c.intr.p;
        c.d_points = c.points.p; c.d_xy = c.xy.p; c.d_ocam = c.ocam.p; c.d_opt = c.opt.p;
    } else {
        d_pose = p->cam_pose; d_intr = p->cam_intr;
        c.d_points = p->points; c.d_xy = p->obs_xy; c.d_ocam = p->obs_cam; c.d_opt = p->obs_pt;
    }
    HIPCHK(c.aux.alloc((size_t)p->n_cams * BA_CAM_AUX));
    if (p->n_cams)
        hipLaunchKernelGGL(k_cam_prep, dim3((p->n_cams + 63) / 64), dim3(64), 0, 0, p->n_cams, d_pose, d_intr, c.aux.p);
    HIPCHK(hipGetLastError());
    return BA_OK;
}

int copy_out(double *dst, const double *src_dev, size_t n, int memory) {
    if (n == 0) return BA_OK;
    HIPCHK(hipMemcpy(dst, src_dev, n * sizeof(double), memory == BA_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return BA_OK;
}
}  // namespace

int ba_eval_residuals(const BaProblem *p, double *out) {
    EvalCtx c;
    int e = eval_setup(p, c, true);
    if (e) return e;
    if (!out) return fail(BA_ERR_INVALID, "null output");
    if (p->n_obs == 0) return BA_OK;
    DevBuf<double> res;
    HIPCHK(res.alloc((size_t)p->n_obs * 2));
    hipLaunchKernelGGL(k_eval_residuals, dim3((unsigned)((p->n_obs + 255) / 256)), dim3(256), 0, 0, (long long)p->n_obs,
                       c.aux.p, c.d_points, c.d_ocam, c.d_opt, c.d_xy, res.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return copy_out(out, res.p, (size_t)p->n_obs * 2, p->memory);
}

int ba_eval_cost(const BaProblem *p, int32_t loss, double f_scale, double *out_cost) {
    EvalCtx c;
    int e = eval_setup(p, c, true);
    if (e) return e;
    if (!out_cost) return fail(BA_ERR_INVALID, "null output");
    if (loss < 0 || loss > 4 || !(f_scale > 0.0)) return fail(BA_ERR_INVALID, "bad loss / f_scale");
    *out_cost = 0.0;
    const int nb = (int)((p->n_obs + 255) / 256);
    if (nb == 0) return BA_OK;
    DevBuf<double> bc, br, o2;
    HIPCHK(bc.alloc(nb));
    HIPCHK(br.alloc(nb));
    HIPCHK(o2.alloc(1));
    hipLaunchKernelGGL(k_eval_cost<false>, dim3(nb), dim3(256), 0, 0, (long long)p->n_obs, c.aux.p, c.d_points, c.d_ocam, c.d_opt,
                       c.d_xy, loss, 1.0 / (f_scale * f_scale), 0.5 * f_scale * f_scale, bc.p, br.p, 0, 0, (int *)nullptr);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, 0, nb, 1, bc.p, o2.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out_cost, o2.p, sizeof(double), hipMemcpyDeviceToHost));
    return BA_OK;
}

int ba_eval_jacobian_blocks(const BaProblem *p, double *out_jc, double *out_jp) {
    EvalCtx c;
    int e = eval_setup(p, c, false);
    if (e) return e;
    if (!out_jc || !out_jp) return fail(BA_ERR_INVALID, "null output");
    if (p->n_obs == 0) return BA_OK;
    DevBuf<double> jc, jp;
    HIPCHK(jc.alloc((size_t)p->n_obs * 12));
    HIPCHK(jp.alloc((size_t)p->n_obs * 6));
    hipLaunchKernelGGL(k_eval_jacobian, dim3((unsigned)((p->n_obs + 255) / 256)), dim3(256), 0, 0, (long long)p->n_obs,
                       c.aux.p, c.d_points, c.d_ocam, c.d_opt, jc.p, jp.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if ((e = copy_out(out_jc, jc.p, (size_t)p->n_obs * 12, p->memory))) return e;
    return copy_out(out_jp, jp.p, (size_t)p->n_obs * 6, p->memory);
}

namespace {
// shared tail of the three reprojection-error entry points: launch over (rt rows, K, points, observations) that are all
// on the device already, per-camera means to out_err (host pointer unless out_on_device)
int reproj_launch(hipStream_t st, int n_cams, long long n_obs, const double *rt, int stride, int t_off, const double *cam_k_dev,
                  const double *points, const int *ocam, const int *opt, const double *xy, double *out_err, bool out_on_device) {
    const size_t nc = (size_t)n_cams;
    if (nc == 0) return BA_OK;
    DevBuf<double> sum, cnt, mean;
    HIPCHK(sum.alloc(nc));
    HIPCHK(cnt.alloc(nc));
    HIPCHK(mean.alloc(nc));
    HIPCHK(hipMemsetAsync(sum.p, 0, nc * sizeof(double), st));
    HIPCHK(hipMemsetAsync(cnt.p, 0, nc * sizeof(double), st));
    if (n_obs)
        hipLaunchKernelGGL(k_eval_reproj, dim3((unsigned)((n_obs + 255) / 256)), dim3(256), 0, st, n_obs, rt, stride, t_off,
                           cam_k_dev, points, ocam, opt, xy, sum.p, cnt.p);
    hipLaunchKernelGGL(k_mean_or_nan, dim3((n_cams + 255) / 256), dim3(256), 0, st, n_cams, sum.p, cnt.p, mean.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_err, mean.p, nc * sizeof(double), out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return BA_OK;
}
}  // namespace

int ba_eval_reprojection_errors(const BaProblem *p, const double *cam_k, double *out_err) {
    EvalCtx c;
    int e = eval_setup(p, c, true);
    if (e) return e;
    if (!cam_k || !out_err) return fail(BA_ERR_INVALID, "null argument");
    const size_t nc = (size_t)p->n_cams;
    if (nc == 0) return BA_OK;
    DevBuf<double> k;
    HIPCHK(k.alloc(nc * 9));
    HIPCHK(hipMemcpy(k.p, cam_k, nc * 9 * sizeof(double), p->memory == BA_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
    return reproj_launch(0, p->n_cams, (long long)p->n_obs, c.aux.p, BA_CAM_AUX, 18, k.p, c.d_points, c.d_ocam, c.d_opt, c.d_xy,
                         out_err, p->memory != BA_MEM_HOST);
}

int ba_eval_reprojection_errors_rt(const BaProblem *p, const double *cam_rt, const double *cam_k, double *out_err) {
    if (!p || !cam_rt || !cam_k || !out_err) return fail(BA_ERR_INVALID, "null argument");
    BaProblem q = *p;
    q.cam_pose = const_cast<double *>(cam_rt);  // (only so that the shared set-up finds a pose array; the rvec-derived rows are not used)
    // (cam_intr is not used either, but the shared set-up wants the pointer)
    if (!q.cam_intr) q.cam_intr = cam_k;
    EvalCtx c;
    int e = eval_setup(&q, c, true);
    if (e) return e;
    const size_t nc = (size_t)p->n_cams;
    if (nc == 0) return BA_OK;
    const hipMemcpyKind kind = p->memory == BA_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    DevBuf<double> k, rt;
    HIPCHK(k.alloc(nc * 9));
    HIPCHK(rt.alloc(nc * 12));
    HIPCHK(hipMemcpy(k.p, cam_k, nc * 9 * sizeof(double), kind));
    HIPCHK(hipMemcpy(rt.p, cam_rt, nc * 12 * sizeof(double), kind));
    return reproj_launch(0, p->n_cams, (long long)p->n_obs, rt.p, 12, 9, k.p, c.d_points, c.d_ocam, c.d_opt, c.d_xy, out_err,
                         p->memory != BA_MEM_HOST);
}

int ba_get_reprojection_errors(BaHandle *h, const double *cam_rt, const double *cam_k, double *out_err) {
    if (!h || !out_err) return fail(BA_ERR_INVALID, "null argument");
    if (int eu = usable(h)) return eu;
    // (a sharded handle only holds its own shard's observations: a per-camera mean over them is not the camera's mean)
    if (h->coll) return fail(BA_ERR_UNSUPPORTED, "ba_get_reprojection_errors on a sharded handle (each rank holds only its shard's observations)");
    (void)hipSetDevice(h->device);
    const Symbolic &s = h->sym;
    const size_t nc = (size_t)s.n_cams;
    if (nc == 0) return BA_OK;
    DevBuf<double> k, rt;
    HIPCHK(k.alloc(nc * 9));
    if (cam_k) HIPCHK(hipMemcpyAsync(k.p, cam_k, nc * 9 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    else hipLaunchKernelGGL(k_intr_to_k, dim3((s.n_cams + 63) / 64), dim3(64), 0, h->stream, s.n_cams, h->intr.p, k.p);
    if (cam_rt) {
        HIPCHK(rt.alloc(nc * 12));
        HIPCHK(hipMemcpyAsync(rt.p, cam_rt, nc * 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    // resident state: points and observations in the engine's order (a per-camera mean does not depend on the order),
    // and -- unless the caller hands the matrices over -- the current poses' aux rows (R = Rodrigues(rvec), t)
    return reproj_launch(h->stream, s.n_cams, (long long)s.n_obs, cam_rt ? rt.p : h->aux[h->cur].p, cam_rt ? 12 : BA_CAM_AUX,
                         cam_rt ? 9 : 18, k.p, h->points[h->cur].p, h->o_cam.p, h->o_pt.p, h->o_xy.p, out_err, false);
}

// ---- the one entry point that allocates nothing -------------------------------------------------------------------
namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_scratch_mu;
double *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// device address of `scratch` on the CURRENT device (looked up once per device, with that device current)
int noalloc_scratch(double **out) {
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(BA_ERR_UNSUPPORTED, "device ordinal %d (at most %d devices)", dev, NOALLOC_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(BA_ERR_HIP, "hipGetSymbolAddress returned a null pointer");
        g_scratch_ptr[dev] = static_cast<double *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return BA_OK;
}
}  // namespace

int ba_eval_cost_noalloc(const BaProblem *p, int32_t loss, double f_scale, double *out_cost_dev, void *stream) {
    int e = validate(p);
    if (e) return e;
    if (!out_cost_dev) return fail(BA_ERR_INVALID, "null output");
    if (p->memory != BA_MEM_DEVICE) return fail(BA_ERR_INVALID, "ba_eval_cost_noalloc needs device-resident arrays (BA_MEM_DEVICE): staging host arrays would allocate");
    if (loss < 0 || loss > 4 || !(f_scale > 0.0)) return fail(BA_ERR_INVALID, "bad loss / f_scale");
    if (p->n_cams > BA_NOALLOC_MAX_CAMS || p->n_obs > BA_NOALLOC_MAX_OBS)
        return fail(BA_ERR_UNSUPPORTED, "ba_eval_cost_noalloc: %d cameras / %lld observations exceed the static scratch (%d / %lld)",
                    p->n_cams, (long long)p->n_obs, BA_NOALLOC_MAX_CAMS, (long long)BA_NOALLOC_MAX_OBS);
    double *scr = nullptr;
    if ((e = noalloc_scratch(&scr))) return e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *aux = scr + BA_SCR_AUX, *bc = scr + BA_SCR_BC, *br = scr + BA_SCR_BR;
    int *flag = reinterpret_cast<int *>(scr + BA_SCR_FLAG);
    const int nb = (int)((p->n_obs + 255) / 256);
    if (nb == 0) {
        HIPCHK(hipMemsetAsync(out_cost_dev, 0, sizeof(double), st));
        return BA_OK;
    }
    // the array keeps the last call's contents: the flag must start at zero; aux and the partials are overwritten
    // for exactly the cameras / blocks this call reads
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), st));
    // (no cameras: nothing to prepare -- a grid of zero blocks is an invalid launch -- and k_eval_cost<true> flags every
    // observation, so the result is NaN as the header promises)
    if (p->n_cams > 0)
        hipLaunchKernelGGL(k_cam_prep, dim3((p->n_cams + 63) / 64), dim3(64), 0, st, p->n_cams, (const double *)p->cam_pose, p->cam_intr, aux);
    hipLaunchKernelGGL(k_eval_cost<true>, dim3(nb), dim3(256), 0, st, (long long)p->n_obs, aux, (const double *)p->points, p->obs_cam,
                       p->obs_pt, p->obs_xy, loss, 1.0 / (f_scale * f_scale), 0.5 * f_scale * f_scale, bc, br, p->n_cams,
                       p->n_points, flag);
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(1024), 0, st, nb, 1, bc, out_cost_dev);
    hipLaunchKernelGGL(k_nan_if, dim3(1), dim3(1), 0, st, flag, out_cost_dev);
    HIPCHK(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
