// nwknn_comm.cpp -- the multi-GPU face of the C ABI (include/nwknn.h, "one process, N GPUs of one node"): base rows sharded over the
// devices, queries replicated, per-shard partial top-k, ONE RCCL all-gather of the [Q,k] partial lists over xGMI, merge by (distance,
// index).  The reference has nothing to mirror here -- it is pinned to GPU 0 (cu_knn.py:65) and folds per-base-tile partial tables on
// the CPU (merge.py:56-101); this is SURVEY 8b/8e's design, reachable from any host language, not only through torch.distributed
// (neighborhoodwatch_amd/distributed.py is the one-process-per-GPU flavour of the same exchange).
//
// One host thread per device drives that device's engine handle (nwknn_search blocks on the handle's stream and returns with the partial
// lists complete in device memory); the collective is issued from the calling thread inside ncclGroupStart/End, one call per device, on
// that device's EXCHANGE stream (DevPart::stream: collective, merge and D2H of the device, created here, not the engine's); afterwards
// device g merges only ITS slice of the query rows (rows [Q g / N, Q (g+1) / N) of every gathered list) and copies that slice to the
// caller's host buffers.  An error after work has been queued never returns with that work in flight: an open RCCL group is closed
// and every exchange stream is drained first (the caller's host buffers are the targets of asynchronous copies).
// STATUS: more than one device has never executed on hardware (no multi-GPU box was available to the builder).  What a one-GPU box can
// run is covered there: the whole path with one device, and the sliced / two-level merge for any number of lists through
// nwknn_comm_merge_lists (tests/test_gpu_distributed_rccl.py).
// RCCL is bound at run time (dlopen of librccl.so.1: the copy already in the process -- torch ships one -- or the system's), so the
// library itself carries no link-time dependency on it and single-GPU users never load it... except through this file's entry points.
#include "../../include/nwknn.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "nwknn_internal.h"

static int cfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return nwknn_internal_fail(code, buf);
}

#define CHIP(expr)                                                                                                              \
    do {                                                                                                                        \
        hipError_t e_ = (expr);                                                                                                 \
        if (e_ != hipSuccess)                                                                                                   \
            return cfail(e_ == hipErrorOutOfMemory ? NWKNN_ENOMEM : NWKNN_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                                                   \
    } while (0)

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
#ifdef NWKNN_DEV
    bool shares_one_device = false;               // developer builds only: the loaded transport says its ranks may live on one GPU (tests/fake_rccl: not RCCL)
#endif
};

static int load_rccl(Rccl* r)
{
    static Rccl cached;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        // NWKNN_RCCL_LIBRARY: a site's own build of RCCL, by path.  (The test suite points a DEVELOPER build of this library -- -DNWKNN_DEV,
        // ab/libnwknn_devcomm.so -- at tests/fake_rccl, a stand-in whose ranks share one GPU, to execute the N > 1 orchestration on a one-GPU
        // box; the shipped library has no such relaxation: one shard per GPU, whatever transport is loaded.)
        const char* forced = getenv("NWKNN_RCCL_LIBRARY");
        if (forced && *forced) {
            lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            fprintf(stderr, "libnwknn: NWKNN_RCCL_LIBRARY=%s %s\n", forced, lib ? "loaded as the collective transport" : "could not be loaded");      // never silent
        } else {
            for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }     // a copy already in the process first
            for (const char* n : names) { if (lib) break; lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
        }
        if (lib) {
            cached.lib = lib;
            cached.CommInitAll = (decltype(cached.CommInitAll))dlsym(lib, "ncclCommInitAll");
            cached.CommDestroy = (decltype(cached.CommDestroy))dlsym(lib, "ncclCommDestroy");
            cached.AllGather = (decltype(cached.AllGather))dlsym(lib, "ncclAllGather");
            cached.GroupStart = (decltype(cached.GroupStart))dlsym(lib, "ncclGroupStart");
            cached.GroupEnd = (decltype(cached.GroupEnd))dlsym(lib, "ncclGroupEnd");
            cached.GetErrorString = (decltype(cached.GetErrorString))dlsym(lib, "ncclGetErrorString");
            cached.GetVersion = (decltype(cached.GetVersion))dlsym(lib, "ncclGetVersion");
#ifdef NWKNN_DEV
            cached.shares_one_device = dlsym(lib, "nwknn_test_transport_shares_one_device") != nullptr;
            if (cached.shares_one_device)
                fprintf(stderr, "libnwknn (DEVELOPER build): the loaded transport is a TEST stand-in whose ranks share one GPU; the one-shard-per-GPU guard is off\n");
#endif
        }
    }
    if (!cached.lib) return cfail(NWKNN_EHIP, "RCCL is not available: dlopen(librccl.so.1) failed: %s", dlerror());
    if (!cached.CommInitAll || !cached.CommDestroy || !cached.AllGather || !cached.GroupStart || !cached.GroupEnd || !cached.GetErrorString)
        return cfail(NWKNN_EHIP, "RCCL library lacks a required symbol (ncclCommInitAll / ncclAllGather / ncclGroupStart ...)");
    *r = cached;
    return NWKNN_OK;
}

#define CNCCL(c, expr)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return cfail(NWKNN_EHIP, "%s failed: %s (%s:%d)", #expr, (c)->rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct DBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return NWKNN_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) { p = nullptr; return cfail(NWKNN_ENOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e)); }
        bytes = need;
        return NWKNN_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct DevPart {
    int device = -1;
    nwknn_handle h = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;                 // collective + merge + D2H of this device
    DBuf b_part_d, b_part_i;                      // [Q][k] this shard's partial lists
    DBuf b_all_d, b_all_i;                        // [N][Q][k] gathered
    DBuf b_grp_d, b_grp_i;                        // fold buffer when N * k exceeds one merge (see merge_slice)
    DBuf b_out_d, b_out_i;                        // [Q slice][k] merged
    float* part_d = nullptr; long long* part_i = nullptr;
    float* all_d = nullptr; long long* all_i = nullptr;
    float* grp_d = nullptr; long long* grp_i = nullptr;
    float* out_d = nullptr; long long* out_i = nullptr;
    long long row_lo = 0, row_hi = 0;             // this shard's base rows
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
};

struct nwknn_comm {
    Rccl rccl;
    std::vector<DevPart> dev;
    long long rows = 0;
    long long ms_docs = 0;                        // > 0: the devices hold MaxSim documents (rows counts documents then)
    int d = 0;
    nwknn_comm_stats stats;
};

static void shard_bounds(long long n, int world, int r, long long* lo, long long* hi)       // distributed.shard_bounds
{
    *lo = n * r / world;
    *hi = n * (r + 1) / world;
}

template <class F>
static int on_every_device(nwknn_comm* c, F&& f)
{
    // one host thread per device (the engine calls block on their stream); the first failure is reported, its message kept
    const int n = (int)c->dev.size();
    std::vector<int> rc((size_t)n, NWKNN_OK);
    std::vector<std::string> msg((size_t)n);
    std::vector<std::thread> pool;
    auto work = [&](int g) {
        rc[g] = f(g);
        if (rc[g] != NWKNN_OK) msg[g] = nwknn_last_error();      // thread-local in the worker: carry it over
    };
    pool.reserve((size_t)n);
    for (int g = 1; g < n; ++g) {
        try { pool.emplace_back(work, g); } catch (...) { work(g); }
    }
    work(0);
    for (auto& t : pool) t.join();
    for (int g = 0; g < n; ++g)
        if (rc[g] != NWKNN_OK) return cfail(rc[g], "device %d: %s", c->dev[g].device, msg[g].c_str());
    return NWKNN_OK;
}

extern "C" int nwknn_comm_destroy(nwknn_comm_handle c)
{
    if (!c) return NWKNN_OK;
    for (DevPart& p : c->dev) {
        if (!p.h) continue;                                   // never created (nwknn_create refused the device): nothing to release
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        if (p.comm && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(p.comm);
        for (DBuf* b : {&p.b_part_d, &p.b_part_i, &p.b_all_d, &p.b_all_i, &p.b_grp_d, &p.b_grp_i, &p.b_out_d, &p.b_out_i}) b->release();
        for (hipEvent_t e : {p.ev0, p.ev1, p.ev2}) if (e) (void)hipEventDestroy(e);
        if (p.stream) (void)hipStreamDestroy(p.stream);
        (void)nwknn_destroy(p.h);
    }
    (void)hipGetLastError();                                  // HIP's last-error slot is sticky: leave none behind for the next launch check
    delete c;
    return NWKNN_OK;
}

extern "C" int nwknn_comm_create(nwknn_comm_handle* out, const int* devices, int n_dev)
{
    if (!out || !devices || n_dev <= 0) return cfail(NWKNN_EINVAL, "nwknn_comm_create: null argument or n_dev <= 0");
    if (n_dev > 64) return cfail(NWKNN_EINVAL, "nwknn_comm_create: at most 64 devices");
    bool twice = false;
    int dup = -1;
    for (int a = 0; a < n_dev; ++a)
        for (int b = a + 1; b < n_dev; ++b)
            if (devices[a] == devices[b]) { twice = true; dup = devices[a]; }
#ifndef NWKNN_DEV
    if (twice)                                                        // refused before anything is loaded (RCCL refuses it too)
        return cfail(NWKNN_EINVAL, "nwknn_comm_create: device %d listed twice (one shard per GPU)", dup);
#endif
    nwknn_comm* c = new (std::nothrow) nwknn_comm();
    if (!c) return cfail(NWKNN_ENOMEM, "out of host memory");
    memset(&c->stats, 0, sizeof c->stats);
    int rc = load_rccl(&c->rccl);
    if (rc != NWKNN_OK) { delete c; return rc; }
#ifdef NWKNN_DEV
    if (twice && !c->rccl.shares_one_device) { delete c; return cfail(NWKNN_EINVAL, "nwknn_comm_create: device %d listed twice (one shard per GPU)", dup); }
#endif
    try { c->dev.resize((size_t)n_dev); } catch (...) { delete c; return cfail(NWKNN_ENOMEM, "out of host memory"); }
    auto fail_with = [&](int code) { (void)nwknn_comm_destroy(c); return code; };
    for (int g = 0; g < n_dev; ++g) {
        DevPart& p = c->dev[g];
        rc = nwknn_create(&p.h, devices[g]);                  // checks the ordinal and that the device is a gfx950
        if (rc != NWKNN_OK) { p.h = nullptr; return fail_with(rc); }
        p.device = devices[g];
        if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&p.ev0) != hipSuccess || hipEventCreate(&p.ev1) != hipSuccess || hipEventCreate(&p.ev2) != hipSuccess)
            return fail_with(cfail(NWKNN_EHIP, "nwknn_comm_create: stream / event creation failed on device %d", devices[g]));
    }
    std::vector<ncclComm_t> comms((size_t)n_dev, nullptr);
    ncclResult_t nr = c->rccl.CommInitAll(comms.data(), n_dev, devices);      // one communicator per device, all in this process
    if (nr != ncclSuccess) return fail_with(cfail(NWKNN_EHIP, "ncclCommInitAll over %d device(s) failed: %s", n_dev, c->rccl.GetErrorString(nr)));
    for (int g = 0; g < n_dev; ++g) c->dev[g].comm = comms[g];
    *out = c;
    return NWKNN_OK;
}

extern "C" int nwknn_comm_device_count(nwknn_comm_handle c) { return c ? (int)c->dev.size() : 0; }

extern "C" nwknn_handle nwknn_comm_engine(nwknn_comm_handle c, int i)
{
    if (!c || i < 0 || i >= (int)c->dev.size()) return nullptr;
    return c->dev[i].h;
}

extern "C" int nwknn_comm_set_option(nwknn_comm_handle c, const char* name, int64_t value)
{
    if (!c) return cfail(NWKNN_EINVAL, "null communicator");
    for (DevPart& p : c->dev) {
        const int rc = nwknn_set_option(p.h, name, value);
        if (rc != NWKNN_OK) return rc;
    }
    return NWKNN_OK;
}

static int check_shards(nwknn_comm* c, int64_t rows, int32_t d, const char* what)
{
    if (!c) return cfail(NWKNN_EINVAL, "%s: null communicator", what);
    if (rows <= 0 || d <= 0) return cfail(NWKNN_EINVAL, "%s: rows=%lld d=%d must be positive", what, (long long)rows, d);
    if (rows < (int64_t)c->dev.size()) return cfail(NWKNN_EINVAL, "%s: %lld base rows cannot be sharded over %zu devices", what, (long long)rows, c->dev.size());
    const int n = (int)c->dev.size();
    for (int g = 0; g < n; ++g) shard_bounds(rows, n, g, &c->dev[g].row_lo, &c->dev[g].row_hi);
    c->rows = rows; c->d = d; c->ms_docs = 0;
    return NWKNN_OK;
}

extern "C" int nwknn_comm_set_base(nwknn_comm_handle c, const void* base_host, int64_t rows, int32_t d, int dtype)
{
    if (!base_host) return cfail(NWKNN_EINVAL, "nwknn_comm_set_base: null base");
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16) return cfail(NWKNN_EINVAL, "nwknn_comm_set_base: unsupported dtype %d", dtype);
    const int rc = check_shards(c, rows, d, "nwknn_comm_set_base");
    if (rc != NWKNN_OK) return rc;
    const size_t es = dtype == NWKNN_DTYPE_F16 ? 2 : 4;
    return on_every_device(c, [&](int g) {
        DevPart& p = c->dev[g];
        return nwknn_set_base(p.h, (const char*)base_host + (size_t)p.row_lo * d * es, p.row_hi - p.row_lo, d, dtype, NWKNN_HOST);
    });
}

extern "C" int nwknn_comm_generate_base(nwknn_comm_handle c, uint64_t seed, int64_t rows, int32_t d, int dtype)
{
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16) return cfail(NWKNN_EINVAL, "nwknn_comm_generate_base: unsupported dtype %d", dtype);
    const int rc = check_shards(c, rows, d, "nwknn_comm_generate_base");
    if (rc != NWKNN_OK) return rc;
    const size_t es = dtype == NWKNN_DTYPE_F16 ? 2 : 4;
    return on_every_device(c, [&](int g) -> int {
        DevPart& p = c->dev[g];
        CHIP(hipSetDevice(p.device));
        void* raw = nullptr;
        CHIP(hipMalloc(&raw, (size_t)(p.row_hi - p.row_lo) * d * es));
        int r = nwknn_generate(p.h, seed, p.row_lo, p.row_hi - p.row_lo, d, dtype, raw);      // rows of the GLOBAL numbering: shard g holds [lo, hi)
        if (r == NWKNN_OK) r = nwknn_set_base(p.h, raw, p.row_hi - p.row_lo, d, dtype, NWKNN_DEVICE);
        if (r == NWKNN_OK) r = nwknn_synchronize(p.h);
        (void)hipFree(raw);
        return r;
    });
}

// device g: the k best of its query slice over the N gathered lists.  One merge sorts at most 4096 (list, rank) entries per query in
// LDS; more (8 shards x k = 1000, 8 x 2048, 64 lists x 1000) are folded in groups of lists, level after level until one merge takes what
// is left -- the same k smallest, as KnnEngine.merge folds in the torch flavour.  A level needs at least two lists per merge: k <= 2048.
static int merge_levels(int n, int k)                                 // 0: one merge; >= 1: that many group levels before it; -1: cannot fold
{
    if ((long long)n * k <= 4096) return 0;
    const int per = 4096 / k;
    if (per < 2) return -1;
    int levels = 0;
    while ((long long)n * k > 4096) { n = (n + per - 1) / per; ++levels; }
    return levels;
}

static int check_merge_plan(const char* what, int n, int k)
{
    if (merge_levels(n, k) >= 0) return NWKNN_OK;
    return cfail(NWKNN_EINVAL, "%s: %d lists of k=%d cannot be merged: more than 4096 entries per query need k <= 2048 (a fold level merges at least two lists)", what, n, k);
}

static int merge_slice(nwknn_comm* c, DevPart& p, int n, long long nq, int k, long long qlo, long long qn, int descending = 0)
{
    (void)c;
    if (qn <= 0) return NWKNN_OK;
    const float* src_d = p.all_d + qlo * k;
    const long long* src_i = p.all_i + qlo * k;
    long long stride = nq * k;                                        // entries between two lists of the current level
    if (merge_levels(n, k) < 0) return check_merge_plan("merge", n, k);
    const int per = std::max(2, 4096 / k);
    // fold buffer: region 0 takes the groups of level 1, region 1 those of level 2, region 0 again those of level 3 (level 1 is consumed by then) ...
    const int g1 = (n + per - 1) / per;
    float* reg_d[2] = {p.grp_d, p.grp_d + (long long)g1 * qn * k};
    long long* reg_i[2] = {p.grp_i, p.grp_i + (long long)g1 * qn * k};
    int level = 0;
    while ((long long)n * k > 4096) {
        const int groups = (n + per - 1) / per;
        float* dst_d = reg_d[level & 1];
        long long* dst_i = reg_i[level & 1];
        for (int gidx = 0; gidx < groups; ++gidx) {
            const int l0 = gidx * per, ln = std::min(per, n - l0);
            hipError_t e = nwk_launch_merge(p.stream, src_d + (long long)l0 * stride, src_i + (long long)l0 * stride, ln, stride, qn, k, k,
                                            dst_d + (long long)gidx * qn * k, dst_i + (long long)gidx * qn * k, descending);
            if (e != hipSuccess) return cfail(NWKNN_EHIP, "merge launch failed: %s", hipGetErrorString(e));
        }
        src_d = dst_d; src_i = dst_i; stride = qn * k; n = groups; ++level;
    }
    hipError_t e = nwk_launch_merge(p.stream, src_d, src_i, n, stride, qn, k, k, p.out_d, p.out_i, descending);
    if (e != hipSuccess) return cfail(NWKNN_EHIP, "merge launch failed: %s", hipGetErrorString(e));
    return NWKNN_OK;
}

// device buffers of the exchange step on one device: the gathered lists of n_lists shards and the merged slice [qlo, qhi) of the query rows
static int ensure_exchange_buffers(DevPart& p, int n_lists, long long nq, int k, long long qlo, long long qhi, bool with_partial)
{
    const size_t qk = (size_t)nq * k;
    const size_t slice = (size_t)(qhi - qlo) * k + 1;
    int rc;
    if (with_partial && ((rc = p.b_part_d.ensure(qk * 4)) || (rc = p.b_part_i.ensure(qk * 8)))) return rc;
    if ((rc = p.b_all_d.ensure(qk * n_lists * 4)) || (rc = p.b_all_i.ensure(qk * n_lists * 8)) || (rc = p.b_out_d.ensure(slice * 4)) ||
        (rc = p.b_out_i.ensure(slice * 8)))
        return rc;
    if ((long long)n_lists * k > 4096) {
        const int per = std::max(2, 4096 / k);
        const size_t g1 = (size_t)((n_lists + per - 1) / per), g2 = (g1 + per - 1) / per;      // the two regions merge_slice alternates between
        const size_t grp = slice * (g1 + g2);
        if ((rc = p.b_grp_d.ensure(grp * 4)) || (rc = p.b_grp_i.ensure(grp * 8))) return rc;
    }
    p.part_d = (float*)p.b_part_d.p; p.part_i = (long long*)p.b_part_i.p;
    p.all_d = (float*)p.b_all_d.p; p.all_i = (long long*)p.b_all_i.p;
    p.grp_d = (float*)p.b_grp_d.p; p.grp_i = (long long*)p.b_grp_i.p;
    p.out_d = (float*)p.b_out_d.p; p.out_i = (long long*)p.b_out_i.p;
    return NWKNN_OK;
}

// wait for everything queued on every exchange stream (an error path must not return while copies into the caller's buffers are in flight)
static void drain_exchange_streams(nwknn_comm* c)
{
    for (DevPart& p : c->dev) {
        if (!p.stream) continue;
        (void)hipSetDevice(p.device);
        (void)hipStreamSynchronize(p.stream);
    }
    (void)hipGetLastError();
}

// step 3 of the exchange on device `p`: merge rows [qlo, qhi) of the n_lists gathered lists in p.all_*, queue the copy to the caller
static int merge_and_return_slice(nwknn_comm* c, DevPart& p, int n_lists, long long nq, int k, long long qlo, long long qhi, float* out_dist,
                                  int64_t* out_idx, int descending = 0)
{
    CHIP(hipSetDevice(p.device));
    CHIP(hipEventRecord(p.ev1, p.stream));
    const int rc = merge_slice(c, p, n_lists, nq, k, qlo, qhi - qlo, descending);
    if (rc != NWKNN_OK) return rc;
    CHIP(hipEventRecord(p.ev2, p.stream));
    if (qhi > qlo) {
        CHIP(hipMemcpyAsync(out_dist + qlo * k, p.out_d, (size_t)(qhi - qlo) * k * 4, hipMemcpyDeviceToHost, p.stream));
        CHIP(hipMemcpyAsync(out_idx + qlo * k, p.out_i, (size_t)(qhi - qlo) * k * 8, hipMemcpyDeviceToHost, p.stream));
    }
    return NWKNN_OK;
}

// The exchange's merge for lists the CALLER gathered (another transport, partial files, a test): dists / idx are HOST arrays
// [n_lists][nq][k] (idx < 0 = empty slot); the query rows are cut into n_slices contiguous slices exactly as nwknn_search_sharded cuts
// them over its devices, slice s is merged on device s mod N of the communicator from the lists' own stride (no repacking), more than
// 4096 entries per query are folded in two levels.  Writes out_dist / out_idx [nq][k] on the host.
extern "C" int nwknn_comm_merge_lists(nwknn_comm_handle c, const float* dists_host, const int64_t* idx_host, int32_t n_lists, int64_t nq, int32_t k,
                                      int32_t n_slices, int descending, float* out_dist, int64_t* out_idx)
{
    if (!c || !dists_host || !idx_host || !out_dist || !out_idx) return cfail(NWKNN_EINVAL, "nwknn_comm_merge_lists: null argument");
    if (n_lists <= 0 || nq <= 0 || k <= 0 || n_slices <= 0 || n_slices > 4096) return cfail(NWKNN_EINVAL, "nwknn_comm_merge_lists: n_lists, nq, k, n_slices must be positive");
    if (k > 4096) return cfail(NWKNN_EINVAL, "nwknn_comm_merge_lists: k=%d exceeds 4096 entries per merge", k);
    if (check_merge_plan("nwknn_comm_merge_lists", n_lists, k) != NWKNN_OK) return NWKNN_EINVAL;      // before anything is uploaded
    const int n = (int)c->dev.size();
    const size_t total = (size_t)n_lists * nq * k;
    int rc = NWKNN_OK;
    std::vector<char> uploaded((size_t)n, 0);
    for (int s = 0; s < n_slices && rc == NWKNN_OK; ++s) {
        DevPart& p = c->dev[s % n];
        long long qlo, qhi;
        shard_bounds(nq, n_slices, s, &qlo, &qhi);
        if (qhi <= qlo) continue;
        if (hipSetDevice(p.device) != hipSuccess) { rc = cfail(NWKNN_EHIP, "hipSetDevice(%d) failed", p.device); break; }
        if (uploaded[(size_t)(s % n)]) {                             // the device's out buffer is reused: its previous slice must have left
            if (hipStreamSynchronize(p.stream) != hipSuccess) { rc = cfail(NWKNN_EHIP, "stream synchronize failed on device %d", p.device); break; }
        }
        rc = ensure_exchange_buffers(p, n_lists, nq, k, 0, (nq + n_slices - 1) / n_slices + 1, false);
        if (rc != NWKNN_OK) break;
        if (!uploaded[(size_t)(s % n)]) {
            if (hipMemcpyAsync(p.all_d, dists_host, total * 4, hipMemcpyHostToDevice, p.stream) != hipSuccess ||
                hipMemcpyAsync(p.all_i, idx_host, total * 8, hipMemcpyHostToDevice, p.stream) != hipSuccess) {
                rc = cfail(NWKNN_EHIP, "upload of the gathered lists to device %d failed", p.device);
                break;
            }
            uploaded[(size_t)(s % n)] = 1;
        }
        rc = merge_and_return_slice(c, p, n_lists, nq, k, qlo, qhi, out_dist, out_idx, descending ? 1 : 0);
    }
    drain_exchange_streams(c);
    return rc;
}

// steps 2 and 3 of a sharded search plus the accounting, shared by kNN and MaxSim: the devices' partial lists (p.part_*) are complete
static int exchange_and_merge(nwknn_comm* c, long long nq, int k, int descending, float* out_dist, int64_t* out_idx)
{
    const int n = (int)c->dev.size();
    const size_t qk = (size_t)nq * k;
    // 2. one all-gather per array over RCCL (xGMI): every device ends up with the N partial lists, in device order == row order.
    //    From here on work is queued on the exchange streams: no early return -- the first error is kept, the group is closed and the
    //    streams are drained before it is handed back.
    auto exchange = [&]() -> int {
        for (int g = 0; g < n; ++g) { CHIP(hipSetDevice(c->dev[g].device)); CHIP(hipEventRecord(c->dev[g].ev0, c->dev[g].stream)); }
        CNCCL(c, c->rccl.GroupStart());
        int in_group = NWKNN_OK;
        for (int g = 0; g < n && in_group == NWKNN_OK; ++g) {
            DevPart& p = c->dev[g];
            in_group = [&]() -> int {
                CNCCL(c, c->rccl.AllGather(p.part_d, p.all_d, qk, ncclFloat32, p.comm, p.stream));
                CNCCL(c, c->rccl.AllGather(p.part_i, p.all_i, qk, ncclInt64, p.comm, p.stream));
                return NWKNN_OK;
            }();
        }
        const ncclResult_t ge = c->rccl.GroupEnd();                  // always: a group left open would swallow the next caller's collectives
        if (in_group != NWKNN_OK) return in_group;
        if (ge != ncclSuccess) return cfail(NWKNN_EHIP, "ncclGroupEnd failed: %s", c->rccl.GetErrorString(ge));
        // 3. device g merges its slice of the query rows and hands it to the caller
        for (int g = 0; g < n; ++g) {
            long long qlo, qhi;
            shard_bounds(nq, n, g, &qlo, &qhi);
            const int r3 = merge_and_return_slice(c, c->dev[g], n, nq, k, qlo, qhi, out_dist, out_idx, descending);
            if (r3 != NWKNN_OK) return r3;
        }
        return NWKNN_OK;
    };
    int rc = exchange();
    if (rc != NWKNN_OK) {
        const std::string keep = nwknn_last_error();                 // draining must not lose the cause
        drain_exchange_streams(c);
        return nwknn_internal_fail(rc, keep.c_str());
    }
    double ag = 0.0, mg = 0.0, sr = 0.0, flops = 0.0, tile = 0.0;
    int rc_sync = NWKNN_OK;
    for (int g = 0; g < n; ++g) {
        DevPart& p = c->dev[g];
        float a = 0.f, m = 0.f;
        if (hipSetDevice(p.device) != hipSuccess || hipStreamSynchronize(p.stream) != hipSuccess) {     // keep draining the others
            if (rc_sync == NWKNN_OK) rc_sync = cfail(NWKNN_EHIP, "exchange stream of device %d failed: %s", p.device, hipGetErrorString(hipGetLastError()));
            continue;
        }
        if (hipEventElapsedTime(&a, p.ev0, p.ev1) != hipSuccess || hipEventElapsedTime(&m, p.ev1, p.ev2) != hipSuccess) { a = m = 0.f; (void)hipGetLastError(); }
        ag = std::max(ag, (double)a); mg = std::max(mg, (double)m);
        nwknn_stats st;
        if (nwknn_get_stats(p.h, &st) == NWKNN_OK) { sr = std::max(sr, st.total_ms); flops += st.tile_flops; tile += st.tile_ms; }
    }
    if (rc_sync != NWKNN_OK) return rc_sync;
    c->stats.n_devices = n; c->stats.nq = nq; c->stats.k = k;
    c->stats.search_ms = sr; c->stats.allgather_ms = ag; c->stats.merge_ms = mg;
    c->stats.allgather_bytes_per_device = (double)qk * 12.0 * n;
    c->stats.tile_flops = flops; c->stats.tile_ms = tile;
    return NWKNN_OK;
}

static int prepare_exchange(nwknn_comm* c, long long nq, int k)
{
    const int n = (int)c->dev.size();
    for (int g = 0; g < n; ++g) {
        DevPart& p = c->dev[g];
        CHIP(hipSetDevice(p.device));
        long long qlo, qhi;
        shard_bounds(nq, n, g, &qlo, &qhi);
        const int rc = ensure_exchange_buffers(p, n, nq, k, qlo, qhi, true);
        if (rc != NWKNN_OK) return rc;
    }
    return NWKNN_OK;
}

extern "C" int nwknn_search_sharded(nwknn_comm_handle c, const void* queries_host, int64_t nq, int dtype, int32_t k, int metric, float* out_dist,
                                    int64_t* out_idx)
{
    if (!c || !queries_host || !out_dist || !out_idx) return cfail(NWKNN_EINVAL, "nwknn_search_sharded: null argument");
    if (c->rows <= 0 || c->ms_docs > 0) return cfail(NWKNN_ESTATE, "nwknn_search_sharded: call nwknn_comm_set_base / nwknn_comm_generate_base first");
    if (nq <= 0) return cfail(NWKNN_EINVAL, "nwknn_search_sharded: nq must be positive");
    for (const DevPart& p : c->dev)
        if (k <= 0 || k > p.row_hi - p.row_lo)
            return cfail(NWKNN_EINVAL, "nwknn_search_sharded: k=%d must satisfy 0 < k <= %lld rows of the smallest shard (cu_knn.py:177-178,237)", k,
                         (long long)(p.row_hi - p.row_lo));
    int rc = check_merge_plan("nwknn_search_sharded", (int)c->dev.size(), k);      // before any shard is searched
    if (rc != NWKNN_OK) return rc;
    rc = prepare_exchange(c, nq, k);
    if (rc != NWKNN_OK) return rc;
    // 1. every device searches its shard; ids are global (index_offset = first row of the shard: "+ batch_offset", cu_knn.py:273)
    rc = on_every_device(c, [&](int g) {
        DevPart& p = c->dev[g];
        return nwknn_search(p.h, queries_host, nq, dtype, NWKNN_HOST, k, metric, p.row_lo, p.part_d, (int64_t*)p.part_i, NWKNN_DEVICE);
    });
    if (rc != NWKNN_OK) return rc;                                   // nwknn_search returns with its stream drained: nothing is in flight
    return exchange_and_merge(c, nq, k, 0, out_dist, out_idx);
}

// ---- MaxSim over the devices (BASELINE config 5 "on 8 GPUs"; an extension like nwknn_maxsim_*: no reference parity) --------------------------
// Documents sharded contiguously (device g holds docs [n g / N, n (g+1) / N)), queries replicated, per-shard top-k with global doc ids,
// the same ONE all-gather per array, and the sliced merge ranks by (score descending, doc id ascending).
extern "C" int nwknn_comm_maxsim_set_docs(nwknn_comm_handle c, const void* doc_tokens_host, int64_t n_docs, int32_t tokens_per_doc, int32_t d, int dtype)
{
    if (!doc_tokens_host) return cfail(NWKNN_EINVAL, "nwknn_comm_maxsim_set_docs: null documents");
    if (dtype != NWKNN_DTYPE_F32) return cfail(NWKNN_EINVAL, "nwknn_comm_maxsim_set_docs: float32 tokens only");
    if (tokens_per_doc <= 0) return cfail(NWKNN_EINVAL, "nwknn_comm_maxsim_set_docs: tokens_per_doc must be positive");
    const int rc = check_shards(c, n_docs, d, "nwknn_comm_maxsim_set_docs");
    if (rc != NWKNN_OK) return rc;
    const int rc2 = on_every_device(c, [&](int g) {
        DevPart& p = c->dev[g];
        return nwknn_maxsim_set_docs(p.h, (const char*)doc_tokens_host + (size_t)p.row_lo * tokens_per_doc * d * 4, p.row_hi - p.row_lo, tokens_per_doc, d, dtype,
                                     NWKNN_HOST);
    });
    c->ms_docs = rc2 == NWKNN_OK ? n_docs : 0;
    if (rc2 != NWKNN_OK) c->rows = 0;
    return rc2;
}

extern "C" int nwknn_maxsim_search_sharded(nwknn_comm_handle c, const void* query_tokens_host, int64_t n_queries, int32_t tokens_per_query, int dtype,
                                           int32_t k, float* out_score, int64_t* out_doc)
{
    if (!c || !query_tokens_host || !out_score || !out_doc) return cfail(NWKNN_EINVAL, "nwknn_maxsim_search_sharded: null argument");
    if (c->ms_docs <= 0) return cfail(NWKNN_ESTATE, "nwknn_maxsim_search_sharded: call nwknn_comm_maxsim_set_docs first");
    if (n_queries <= 0) return cfail(NWKNN_EINVAL, "nwknn_maxsim_search_sharded: n_queries must be positive");
    for (const DevPart& p : c->dev)
        if (k <= 0 || k > p.row_hi - p.row_lo)
            return cfail(NWKNN_EINVAL, "nwknn_maxsim_search_sharded: k=%d must satisfy 0 < k <= %lld documents of the smallest shard", k,
                         (long long)(p.row_hi - p.row_lo));
    int rc = check_merge_plan("nwknn_maxsim_search_sharded", (int)c->dev.size(), k);
    if (rc != NWKNN_OK) return rc;
    rc = prepare_exchange(c, n_queries, k);
    if (rc != NWKNN_OK) return rc;
    rc = on_every_device(c, [&](int g) {
        DevPart& p = c->dev[g];
        return nwknn_maxsim_search(p.h, query_tokens_host, n_queries, tokens_per_query, dtype, NWKNN_HOST, k, p.row_lo, p.part_d, (int64_t*)p.part_i, NWKNN_DEVICE);
    });
    if (rc != NWKNN_OK) return rc;
    return exchange_and_merge(c, n_queries, k, 1, out_score, out_doc);
}

extern "C" int nwknn_comm_get_stats(nwknn_comm_handle c, nwknn_comm_stats* out)
{
    if (!c || !out) return cfail(NWKNN_EINVAL, "null communicator or output");
    *out = c->stats;
    return NWKNN_OK;
}
