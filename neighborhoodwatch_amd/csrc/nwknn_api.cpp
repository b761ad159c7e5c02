// nwknn_api.cpp -- C ABI (include/nwknn.h) and host-side orchestration of the kNN hot path.
//
// Reference call stack this file stands in for (paths relative to /root/reference):
//   neighborhoodwatch/cu_knn.py:193-287  process_batches(): base tiles x query tiles, knn() per pair, + batch_offset
//   neighborhoodwatch/cu_knn.py:54-109   tune_memory(): NVML memory probe            -> nwknn_mem_info
//   neighborhoodwatch/merge.py:56-101    fold of partial lists                         -> nwknn_merge
// Instead of materialising per-tile top-k tables and folding them on the CPU, one search keeps a running
// per-query candidate list on the device and walks the base in growing chunks.
#include "../../include/nwknn.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "nwknn_internal.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// Developer builds (-DNWKNN_DEV: tools/mk_ab_lib.sh) add the switches whose results are wrong or unguaranteed by construction -- timing
// ablations ("diag"), in-kernel stamps ("dbg_ptr"), a shrunken pre-screen bound ("pre_eps_pct" < 100), unpruned rescoring.  The
// shipped library has none of them: nwknn_set_option answers NWKNN_EINVAL (the reference has no mode that returns wrong neighbours:
// cu_knn.py:281-282 asserts its output).
#ifdef NWKNN_DEV
#define NWK_DEV_BUILD 1
#else
#define NWK_DEV_BUILD 0
#endif

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int nwknn_internal_fail(int code, const char* msg) { g_err = msg ? msg : ""; return code; }       // for the other translation units of the library

#define HIPCHK(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? NWKNN_ENOMEM : NWKNN_EHIP, "%s failed: %s (%s:%d)", #expr,   \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                              \
    } while (0)

#define RCCHK(expr)                \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != NWKNN_OK) return rc_; \
    } while (0)

static inline long long round_up(long long x, long long m) { return (x + m - 1) / m * m; }

// No C++ exception crosses the C ABI (include/nwknn.h): entry points that reach code which allocates host memory run through this.
template <class F>
static int guarded(F&& f)
{
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(NWKNN_ENOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(NWKNN_EHIP, "internal error: %s", e.what()); }
    catch (...) { return fail(NWKNN_EHIP, "internal error (unknown C++ exception)"); }
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return NWKNN_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) { p = nullptr; return fail(NWKNN_ENOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e)); }
        bytes = need;
        return NWKNN_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

#define NWK_UPLOAD_THREADS 8

struct EvPair { hipEvent_t a, b; int kind; };   // kind: 0 tile, 1 compact, 2 prep, 3 rescore

struct SearchPlan {
    int cap; long long first, max_chunk, safe; int tile_rows; int granule;
    bool pre;                                   // filtered chunks run the fp16 pre-screen + fp32 rescoring
    bool native16;                              // exact fp16 mode: the fp16 rows ARE the data; rescoring walks them, the dense chunk is rescored too
    bool copies16;                              // the tile kernel reads the fp16 COPIES (fp32 data; cosine's unit rows), not the native fp16 rows
    float pre_scale, pre_eps_half, pre_aq, pre_ab, pre_cst;
    float chk_scale = 1.0f;                     // fraction of the bound the rescoring kernel holds the invariant against (option pre_check_ppm on attempt 0)
};

// A search that rides a base which is still arriving (nwknn_search_attach / nwknn_search_collect): the plan of the walk and how far it got.
struct FollowState {
    int active = 0;
    SearchPlan pl;
    int k = 0, metric = 0;
    long long nq = 0, nq_pad = 0, index_offset = 0, seen = 0, growth = 1, rows = 0;
};

struct nwknn_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // column upload pipeline (nwknn_set_base_columns / nwknn_search_columns): two pinned host slabs, a copy stream, four events
    hipStream_t copy_stream = nullptr;
    void* pin[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_prepped[2] = {nullptr, nullptr};
    hipEvent_t ev_cal[3] = {nullptr, nullptr, nullptr};      // walk calibration of the direct fp32 kernel (search_attempt)
    long long walk_growth = 3;                               // chunk growth of the current walk (the deferred pass sizes its per-query limit by it)
    int walk_auto_q = 0, walk_auto_sq = 3;                   // what "auto" means for the fp32 kernel's super-tile walk in the current search
    FollowState fw;
    hipStream_t fw_stream = nullptr;                         // the attached search's chunks run here, beside the layout kernels of later pieces
    hipEvent_t ev_fw_ready = nullptr, ev_fw_done = nullptr;
    long long up_seq = 0;           // slabs the column upload pipeline has sent so far (never reset: the two pinned / device slabs alternate across calls)
    // base shard
    DevBuf bp, bn, braw, bmax;      // bmax: float bits of max |x| of the base being laid out (folded into the layout pass when the pre-screen is on)
    int bmax_fused = 0;             // 1: the layout launches of the current base feed bmax
    // a base that is being handed over in pieces (nwknn_base_begin / _append* / _end): rows expected, rows laid out so far; 0 = none
    long long sb_rows = 0, sb_done = 0, sb_rows_pad = 0;
    int sb_d = 0, sb_dp = 0, sb_f16 = 0, sb_pre = 0;
    long long nb = 0, nb_pad = 0;
    int d = 0, dp = 0;
    long long ms_docs = 0;          // > 0: the base holds ms_docs documents of NWK_MS_TOKENS_PER_DOC token rows each (MaxSim)
    int f16 = 0;                    // 1: the prepared base (and queries) are fp16, distance tiles run on the fp16 MFMA kernel
    // query side
    DevBuf qraw, qp, qn;
    // fp16 pre-screen of the exact fp32 search (option "prescreen"): scaled fp16 copies of bp / qp
    DevBuf bp16, qp16, cnt0, scal, ms_rq, ms_cd, ms_surv, ms_nsurv, dskip;
    int ms_pre = 0;                 // MaxSim: the documents have a usable fp16 copy (option "prescreen" at nwknn_maxsim_set_docs)
    float ms_eps_half = 0.f, ms_aq = 0.f, ms_ab = 0.f, ms_cst = 0.f, ms_scale = 1.f;   // constants of the current MaxSim search
    int pre_ok = 0;                 // the current base can be pre-screened (finite, sane magnitudes); bp16 holds a copy of kind pre_kind
    int f16_finite = 0;             // fp16 compute path: the prepared base holds no Inf/NaN (exact fp16 mode is possible)
    int pre_kind = -1;              // 0: rows scaled by b_scale (sqeuclidean, one_minus_dot); 1: unit rows x 2^14 (cosine); -1: none yet
    int dp16 = 0;
    float b_scale = 1.f, b_maxabs = 0.f;
    long long nb_pad_for_copy = 0;
    int dp_for_copy = 0;
    // running top-k state
    DevBuf tau, cnt, buf, ovf, ovf_any, qtiles, out_d, out_i, mrg_d, mrg_i, ctab, chk, calib;
    // options
    long long opt_capacity = 0, opt_first_chunk = 0, opt_max_chunk = 1 << 20, opt_profile = 0, opt_tile_rows = 0, opt_diag = 0, opt_dbg_ptr = 0, opt_compute_f16 = 0, opt_sq_shift = -1, opt_walk_q = -1, opt_prescreen = 0, opt_pre_prune = 1, opt_f16_shape = -1, opt_pre_eps_pct = 100, opt_f16_strip = 0, opt_f16_agg = -1, opt_f16_exact = 1, opt_tile_occ = 3, opt_chunk_growth = 0, opt_upload_threads = NWK_UPLOAD_THREADS, opt_ms_strip = 0, opt_ms_waves = 8, opt_ms_form = 3, opt_ms_rescore_valu = 0, opt_upload_slab_mib = 128, opt_pre_check_ppm = 1000000, opt_pre_direct_rows = -1, opt_rescore_coop = 1, opt_pre_defer = 1;
    int num_cu = 256;
    // profiling
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    double base_prep_ms = 0.0;
    nwknn_stats stats;
    char tile_kernel[96] = "";      // name (rocprofv3's spelling) of the distance-tile kernel behind the last search's filtered launches
};

static int use_device(nwknn_ctx* h)
{
    HIPCHK(hipSetDevice(h->device));
    return NWKNN_OK;
}

// ---- profiling helpers ------------------------------------------------------------------------------
static EvPair* ev_begin(nwknn_ctx* h, int kind)
{
    if (!h->opt_profile) return nullptr;
    if (h->ev_used == h->ev_pool.size()) {
        EvPair e;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    EvPair* e = &h->ev_pool[h->ev_used++];
    e->kind = kind;
    (void)hipEventRecord(e->a, h->stream);
    return e;
}
static void ev_end(nwknn_ctx* h, EvPair* e)
{
    if (e) (void)hipEventRecord(e->b, h->stream);
}
static void ev_collect(nwknn_ctx* h, double* tile_ms, double* compact_ms, double* prep_ms, double* rescore_ms = nullptr)
{
    *tile_ms = *compact_ms = *prep_ms = 0.0;
    if (rescore_ms) *rescore_ms = 0.0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b) != hipSuccess) continue;
        if (h->ev_pool[i].kind == 0) *tile_ms += ms;
        else if (h->ev_pool[i].kind == 1) *compact_ms += ms;
        else if (h->ev_pool[i].kind == 3) { if (rescore_ms) *rescore_ms += ms; }
        else *prep_ms += ms;
    }
    h->ev_used = 0;
}

// ---- library / device -------------------------------------------------------------------------------
extern "C" int nwknn_version(void) { return NWKNN_ABI_VERSION; }
extern "C" const char* nwknn_last_error(void) { return g_err.c_str(); }

extern "C" int nwknn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int nwknn_mem_info(int device, size_t* free_bytes, size_t* total_bytes)
{
    if (!free_bytes || !total_bytes) return fail(NWKNN_EINVAL, "nwknn_mem_info: null output pointer");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
    return NWKNN_OK;
}

static int create_device_objects(nwknn_ctx* h)
{
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamCreate(&h->own_stream));
    h->stream = h->own_stream;
    HIPCHK(hipEventCreate(&h->ev_begin));
    HIPCHK(hipEventCreate(&h->ev_end));
    return NWKNN_OK;
}

extern "C" int nwknn_destroy(nwknn_handle h);
extern "C" int nwknn_set_option(nwknn_handle h, const char* name, int64_t value);

extern "C" int nwknn_create(nwknn_handle* out, int device)
{
    if (!out) return fail(NWKNN_EINVAL, "nwknn_create: null handle pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(NWKNN_ENODEV, "no HIP device visible");
    if (device < 0 || device >= n) return fail(NWKNN_EINVAL, "device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NWKNN_ENODEV, "device %d is %s; libnwknn is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    nwknn_ctx* h = new (std::nothrow) nwknn_ctx();
    if (!h) return fail(NWKNN_ENOMEM, "out of host memory");
    h->device = device;
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    memset(&h->stats, 0, sizeof h->stats);
    int rc = create_device_objects(h);
    // NWKNN_DEFAULT_OPTS="name=value,name=value": options applied to every new handle (developers run the whole test suite against
    // an alternative kernel build this way, e.g. f16_mfma=1); unset in production.  It goes through nwknn_set_option, so in the shipped
    // library it reaches no switch that changes results (those exist under -DNWKNN_DEV only), and its use is reported on stderr.
    if (rc == NWKNN_OK) rc = guarded([&] {
        const char* env = getenv("NWKNN_DEFAULT_OPTS");
        if (!env || !*env) return (int)NWKNN_OK;
        const std::string all(env);
        fprintf(stderr, "libnwknn: NWKNN_DEFAULT_OPTS=\"%s\" applied to the handle on device %d\n", env, device);   // never silent
        for (size_t pos = 0; pos < all.size();) {
            const size_t end = std::min(all.find(',', pos), all.size());
            const std::string kv = all.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) return fail(NWKNN_EINVAL, "NWKNN_DEFAULT_OPTS: '%s' is not name=value", kv.c_str());
            RCCHK(nwknn_set_option(h, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1)));
            pos = end + 1;
        }
        return (int)NWKNN_OK;
    });
    if (rc != NWKNN_OK) { (void)nwknn_destroy(h); return rc; }      // releases whatever was created; last_error keeps the cause
    *out = h;
    return NWKNN_OK;
}

extern "C" int nwknn_destroy(nwknn_handle h)
{
    if (!h) return NWKNN_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (DevBuf* b : {&h->bp, &h->bn, &h->braw, &h->qraw, &h->qp, &h->qn, &h->tau, &h->cnt, &h->buf, &h->ovf, &h->ovf_any, &h->qtiles,
                      &h->out_d, &h->out_i, &h->mrg_d, &h->mrg_i, &h->ctab, &h->chk, &h->calib, &h->bp16, &h->qp16, &h->cnt0, &h->scal, &h->ms_rq, &h->ms_cd, &h->ms_surv, &h->ms_nsurv, &h->dskip})
        b->release();
    for (EvPair& e : h->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int i = 0; i < 2; ++i) {
        if (h->pin[i]) (void)hipHostFree(h->pin[i]);
        if (h->ev_h2d[i]) (void)hipEventDestroy(h->ev_h2d[i]);
        if (h->ev_prepped[i]) (void)hipEventDestroy(h->ev_prepped[i]);
    }
    for (int i = 0; i < 3; ++i) if (h->ev_cal[i]) (void)hipEventDestroy(h->ev_cal[i]);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->ev_fw_ready) (void)hipEventDestroy(h->ev_fw_ready);
    if (h->ev_fw_done) (void)hipEventDestroy(h->ev_fw_done);
    if (h->fw_stream) (void)hipStreamDestroy(h->fw_stream);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return NWKNN_OK;
}

extern "C" int nwknn_set_stream(nwknn_handle h, void* hip_stream)
{
    if (!h) return fail(NWKNN_EINVAL, "null handle");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return NWKNN_OK;
}

extern "C" int nwknn_set_option(nwknn_handle h, const char* name, int64_t value)
{
    if (!name) return fail(NWKNN_EINVAL, "null handle or option name");
    // switches that change results exist in developer builds only; refused by name, before anything else is looked at
    if (!NWK_DEV_BUILD && (!strcmp(name, "diag") || !strcmp(name, "dbg_ptr") || !strcmp(name, "prescreen_prune") || (!strcmp(name, "pre_eps_pct") && value < 100)))
        return fail(NWKNN_EINVAL, "option '%s'=%lld exists in developer builds of libnwknn only (-DNWKNN_DEV): it changes or unguards results", name, (long long)value);
    if (!h) return fail(NWKNN_EINVAL, "null handle or option name");
    if (!strcmp(name, "capacity")) {
        if (value != 0 && (value < 256 || value > NWK_MAX_CAPACITY || (value & (value - 1)))) return fail(NWKNN_EINVAL, "capacity must be 0 (auto) or a power of two in [256, %d]", NWK_MAX_CAPACITY);
        h->opt_capacity = value;
    } else if (!strcmp(name, "first_chunk")) {
        if (value < 0) return fail(NWKNN_EINVAL, "first_chunk must be non-negative");
        h->opt_first_chunk = value;      // base rows (kNN) or docs (MaxSim) of the dense first chunk; rounded up to the tile size
    } else if (!strcmp(name, "max_chunk")) {
        if (value < NWK_TILE_ROWS || value % NWK_TILE_ROWS) return fail(NWKNN_EINVAL, "max_chunk must be a positive multiple of %d", NWK_TILE_ROWS);
        h->opt_max_chunk = value;
    } else if (!strcmp(name, "profile")) {
        h->opt_profile = value ? 1 : 0;
    } else if (!strcmp(name, "compute_dtype")) {
        if (value != NWKNN_DTYPE_F32 && value != NWKNN_DTYPE_F16) return fail(NWKNN_EINVAL, "compute_dtype must be 0 (f32) or 1 (f16)");
        h->opt_compute_f16 = value;      // takes effect at the next nwknn_set_base
    } else if (!strcmp(name, "prescreen")) {
        h->opt_prescreen = value ? 1 : 0;   // takes effect at the next nwknn_set_base (it builds the fp16 copy of the base)
    } else if (!strcmp(name, "f16_exact")) {
        h->opt_f16_exact = value ? 1 : 0;   // fp16 compute path: 1 (default) rescoring to the canonical fp32 chain over the fp16 values, 0 raw MFMA sums
    } else if (!strcmp(name, "f16_agg")) {
        if (value < -1 || value > 1) return fail(NWKNN_EINVAL, "f16_agg must be -1 (auto), 0 (never) or 1 (every filtered fp16 chunk)");
        h->opt_f16_agg = value;             // which chunks run the row-aggregating build of the streaming fp16 kernel (auto: the loose-threshold ones)
    } else if (!strcmp(name, "rescore_coop")) {
        h->opt_rescore_coop = value ? 1 : 0; // rescoring kernel: 1 (default) a wave fetches the rows of its 64 pairs together through LDS, 0 every lane its own row
    } else if (!strcmp(name, "pre_defer")) {
        h->opt_pre_defer = value ? 1 : 0;    // pre-screened walk: 1 (default) only the last chunk rescores, the chunks before it prune on the estimates' bounds (run_chunk); 0 every chunk rescores
    } else if (!strcmp(name, "pre_direct_rows")) {
        if (value < -1) return fail(NWKNN_EINVAL, "pre_direct_rows must be -1 (auto) or a row count >= 0");
        h->opt_pre_direct_rows = value;     // pre-screened search of fp32 data: filtered chunks that START below this many rows run the direct fp32 kernel (run_chunk)
    } else if (!strcmp(name, "f16_strip")) {
        if (value < 0 || value > 64) return fail(NWKNN_EINVAL, "f16_strip must be 0 (auto) or 1..64");
        h->opt_f16_strip = value;           // base tiles a workgroup of the fp16 kernels walks in a row (auto: 2 for the ping-pong form, 4 for the others)
    } else if (!strcmp(name, "chunk_growth")) {
        if (value < 0 || value > 7) return fail(NWKNN_EINVAL, "chunk_growth must be 0 (auto) or 1..7");
        h->opt_chunk_growth = value;        // a filtered chunk holds up to chunk_growth x the rows seen so far (1: chunks double)
    } else if (!strcmp(name, "ms_rescore_valu")) {
        h->opt_ms_rescore_valu = value ? 1 : 0;   // MaxSim rescoring on the vector pipes (round-1 kernel) instead of the matrix pipes
    } else if (!strcmp(name, "ms_waves")) {
        if (value != 4 && value != 8) return fail(NWKNN_EINVAL, "ms_waves must be 8 (256 x 256 tiles of 8 queries, the default) or 4 (256 x 128 tiles of 4 queries)");
        h->opt_ms_waves = value;
    } else if (!strcmp(name, "ms_form")) {
        if (value < 1 || value > 3)
            return fail(NWKNN_EINVAL, "ms_form (eight-wave MaxSim pre-screen tile) must be 3 (tile-granular ring, epilogue pipelined under the MFMAs: the default), "
                                      "2 (eight-slab ring, one barrier per two k-slabs) or 1 (four-slab ring, one barrier per k-slab)");
        h->opt_ms_form = value;
    } else if (!strcmp(name, "ms_strip")) {
        if (value < 0 || value > 64) return fail(NWKNN_EINVAL, "ms_strip must be 0..64");
        h->opt_ms_strip = value;            // MaxSim pre-screen, d <= 128: document tiles a workgroup of the ring kernel walks per query group (0: auto, 1: the any-d kernel)
    } else if (!strcmp(name, "upload_threads")) {
        if (value < 1 || value > 64) return fail(NWKNN_EINVAL, "upload_threads must be 1..64");
        h->opt_upload_threads = value;      // host threads that gather column slabs into the pinned upload buffers
    } else if (!strcmp(name, "upload_slab_mib")) {
        if (value < 1 || value > 1024) return fail(NWKNN_EINVAL, "upload_slab_mib must be 1..1024");
        h->opt_upload_slab_mib = value;     // size of one slab of the column upload pipeline (two pinned + two device slabs of this size)
    } else if (!strcmp(name, "tile_occ")) {
        if (value != 2 && value != 3) return fail(NWKNN_EINVAL, "tile_occ must be 2 or 3");
        h->opt_tile_occ = value;            // workgroups per CU of the fp32 tile kernel's 256-row shape (3: default)
    } else if (!strcmp(name, "super_walk_q")) {
        if (value < -1 || value > 1) return fail(NWKNN_EINVAL, "super_walk_q must be -1 (auto), 0 (base tile index fastest) or 1 (query tile index fastest)");
        h->opt_walk_q = value;              // walk order inside a super-tile
    } else if (!strcmp(name, "pre_eps_pct")) {
        // scales the pre-screen error bound.  Results are guaranteed at >= 100 only: smaller values exist in developer builds
        if (value < 0 || value > 100000) return fail(NWKNN_EINVAL, "pre_eps_pct must be 100..100000 (percent of the proven bound)");
        h->opt_pre_eps_pct = value;
    } else if (!strcmp(name, "pre_check_ppm")) {
        // the invariant |estimate - exact| <= bound is checked against this share of the bound.  No value changes a result: a violation
        // sends the search down its fallback, which returns the same bits
        if (value < 1 || value > 1000000) return fail(NWKNN_EINVAL, "pre_check_ppm must be 1..1000000 (millionths of the pre-screen bound the invariant is checked against)");
        h->opt_pre_check_ppm = value;
    } else if (!strcmp(name, "f16_mfma")) {
        if (value != -1 && value != 1 && value != 3 && value != 4 && value != 5)
            return fail(NWKNN_EINVAL, "f16_mfma must be -1 (auto), 1 (burst form), 3 (streaming form), 4 (ping-pong form) or 5 (two-issuing-waves form)");
        h->opt_f16_shape = value;
#if NWK_DEV_BUILD
    } else if (!strcmp(name, "prescreen_prune")) {
        h->opt_pre_prune = value ? 1 : 0;   // dev: 0 rescoring every listed pair (A/B of the bound-based pruning)
    } else if (!strcmp(name, "diag")) {
        h->opt_diag = value;             // dev: timing ablations of the tile kernel (results are wrong when != 0)
    } else if (!strcmp(name, "dbg_ptr")) {
        h->opt_dbg_ptr = value;          // dev: device buffer of >= grid*64 bytes receiving per-workgroup cycle stamps
#endif
    } else if (!strcmp(name, "super_q_shift")) {
        if (value < -1 || value > 6) return fail(NWKNN_EINVAL, "super_q_shift must be -1 (auto) or 0..6");
        h->opt_sq_shift = value;         // super-tile of an XCD = 2^v query tiles x 2^(6-v) base tiles
    } else if (!strcmp(name, "tile_rows")) {
        if (value != 0 && value != 128 && value != 256) return fail(NWKNN_EINVAL, "tile_rows must be 0 (auto), 128 or 256");
        h->opt_tile_rows = value;
    } else {
        return fail(NWKNN_EINVAL, "unknown option '%s'", name);
    }
    return NWKNN_OK;
}

extern "C" int nwknn_get_stats(nwknn_handle h, nwknn_stats* out)
{
    if (!h || !out) return fail(NWKNN_EINVAL, "null handle or output");
    *out = h->stats;
    return NWKNN_OK;
}

extern "C" const char* nwknn_tile_kernel_name(nwknn_handle h)
{
    return h ? h->tile_kernel : "";
}

extern "C" int nwknn_synchronize(nwknn_handle h)
{
    if (!h) return fail(NWKNN_EINVAL, "null handle");
    RCCHK(use_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    return NWKNN_OK;
}

// ---- dataset side -----------------------------------------------------------------------------------
// Stages a caller matrix onto the device (if it is on the host) and runs the prep kernel into dst/norms.
// in_f16: element type of the caller's matrix; out_f16: element type of the prepared layout (the compute path).
static int stage_and_prep(nwknn_ctx* h, const void* src, int location, int in_f16, int out_f16, long long rows, int d, long long rows_pad,
                          int dp, DevBuf& raw, void* dst, float* norms, unsigned int* maxabs_bits = nullptr)
{
    const size_t in_es = in_f16 ? 2 : 4, out_es = out_f16 ? 2 : 4;
    auto launch = [&](const void* dsrc, long long real, long long pad, void* ddst, float* dn) -> hipError_t {
        if (out_f16) return nwk_launch_prep_f16(h->stream, dsrc, in_f16, real, d, pad, dp, ddst, dn);
        return nwk_launch_prep(h->stream, (const float*)dsrc, real, d, pad, dp, (float*)ddst, dn, maxabs_bits);
    };
    if (in_f16 && !out_f16) return fail(NWKNN_EINVAL, "fp16 input needs the fp16 compute path (set option compute_dtype=1 or pass an fp16 base)");
    if (location == NWKNN_HOST) {
        // slab-wise so the raw staging copy stays small next to a 288 GB shard
        const long long slab_rows = std::max<long long>(256, ((256ll << 20) / ((long long)d * (long long)in_es)) / 256 * 256);
        RCCHK(raw.ensure((size_t)std::min(slab_rows, std::max<long long>(rows, 1)) * d * in_es));
        for (long long r0 = 0; r0 < rows_pad; r0 += slab_rows) {
            const long long pad_here = std::min(slab_rows, rows_pad - r0);
            const long long real_here = std::max<long long>(0, std::min(slab_rows, rows - r0));
            if (real_here > 0)
                HIPCHK(hipMemcpyAsync(raw.p, (const char*)src + (size_t)r0 * d * in_es, (size_t)real_here * d * in_es, hipMemcpyHostToDevice, h->stream));
            EvPair* e = ev_begin(h, 2);
            HIPCHK(launch(raw.p, real_here, pad_here, (char*)dst + (size_t)r0 * dp * out_es, norms + r0));
            ev_end(h, e);
            HIPCHK(hipStreamSynchronize(h->stream));    // the slab is reused by the next copy
        }
        return NWKNN_OK;
    }
    EvPair* e = ev_begin(h, 2);
    HIPCHK(launch(src, rows, rows_pad, dst, norms));
    ev_end(h, e);
    return NWKNN_OK;
}

// Column-major host input (one pointer per dimension, as parquet/arrow hold embeddings): each slab of rows is staged as
// [d][slab_rows] by d plain copies (no host-side transpose) and the strided prep kernel writes the prepared layout.
// Column-major host data as the caller holds it: chunk c of column j at chunks[c * d + j], chunk_rows[c] rows each, the same chunking
// for every column (a pyarrow ChunkedArray per dimension has one chunk per parquet row group; a plain array is one chunk).
struct ColumnSource {
    const void* const* chunks;
    const int64_t* chunk_rows;
    int n_chunks;
};

// Column-major host data (one array per dimension: what parquet / arrow hold) -> prepared layout, as a three-stage pipeline:
//   host threads copy a slab of every column into a PINNED buffer (the arrow buffers are pageable: a plain hipMemcpy of them
//   stages through the driver's own bounce buffers at ~11 GB/s)  ->  one H2D copy of the whole slab on a copy stream
//   ->  the strided layout kernel on the compute stream.  Two pinned slabs and two device slabs: slab s+1 is gathered and copied
//   while slab s is laid out.  C2's base (1M x 768 float32 columns = 3.07 GB): 0.28 s -> see DESIGN.md section 6.
// (slab size: option "upload_slab_mib", 128 by default)

static int ensure_upload_pipeline(nwknn_ctx* h, size_t slab_bytes)
{
    if (!h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!h->ev_h2d[i]) HIPCHK(hipEventCreateWithFlags(&h->ev_h2d[i], hipEventDisableTiming));
        if (!h->ev_prepped[i]) HIPCHK(hipEventCreateWithFlags(&h->ev_prepped[i], hipEventDisableTiming));
    }
    if (h->pin_bytes < slab_bytes) {
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) { (void)hipHostFree(h->pin[i]); h->pin[i] = nullptr; }
            hipError_t e = hipHostMalloc(&h->pin[i], slab_bytes, hipHostMallocDefault);
            if (e != hipSuccess) { h->pin[i] = nullptr; h->pin_bytes = 0; return fail(NWKNN_ENOMEM, "hipHostMalloc(%zu bytes) failed: %s", slab_bytes, hipGetErrorString(e)); }
        }
        h->pin_bytes = slab_bytes;
    }
    return NWKNN_OK;
}

// finish = false (a piece of a streamed base, nwknn_base_append_column_chunks): the call returns as soon as the caller's arrays have been
// gathered into the pinned slabs -- the H2D copy and the layout kernel of the last slab(s) are still in flight on the copy / compute
// streams, ordered against the next call's by the same events that order slab s against slab s+2 inside one call.
static int stage_and_prep_columns(nwknn_ctx* h, const ColumnSource& src, int in_dtype, int out_f16, long long rows, int d, long long rows_pad,
                                  int dp, DevBuf& raw, void* dst, float* norms, bool finish = true, unsigned int* maxabs_bits = nullptr)
{
    const size_t in_es = in_dtype == NWKNN_DTYPE_F64 ? 8 : (in_dtype == NWKNN_DTYPE_F16 ? 2 : 4), out_es = out_f16 ? 2 : 4;
    if (in_dtype == NWKNN_DTYPE_F16 && !out_f16) return fail(NWKNN_EINVAL, "fp16 columns need the fp16 compute path");
    if (src.n_chunks <= 0 || !src.chunks || !src.chunk_rows) return fail(NWKNN_EINVAL, "column table: no chunks");
    std::vector<long long> chunk_start((size_t)src.n_chunks + 1, 0);
    for (int c = 0; c < src.n_chunks; ++c) {
        if (src.chunk_rows[c] < 0) return fail(NWKNN_EINVAL, "column chunk %d has %lld rows", c, (long long)src.chunk_rows[c]);
        chunk_start[c + 1] = chunk_start[c] + src.chunk_rows[c];
        for (int j = 0; j < d; ++j)
            if (src.chunk_rows[c] > 0 && !src.chunks[(size_t)c * d + j]) return fail(NWKNN_EINVAL, "column %d, chunk %d is a null pointer", j, c);
    }
    if (chunk_start[src.n_chunks] != rows) return fail(NWKNN_EINVAL, "column chunks hold %lld rows, expected %lld", chunk_start[src.n_chunks], rows);
    // slab = as many rows (a multiple of 256) as fit NWK_UPLOAD_SLAB_BYTES with all d columns
    const long long slab_rows = std::max<long long>(256, (long long)(((size_t)h->opt_upload_slab_mib << 20) / ((size_t)d * in_es)) / 256 * 256);
    const long long stage_rows = std::min(slab_rows, round_up(std::max<long long>(rows, 1), 256));
    const size_t slab_bytes = (size_t)stage_rows * d * in_es;
    if (raw.bytes < 2 * slab_bytes || h->pin_bytes < slab_bytes || finish) {
        // whatever the streams still do with the slabs (a previous call's copy or layout kernel) must finish before they are re-allocated;
        // a one-call upload also starts from idle streams, as it always did
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->copy_stream) HIPCHK(hipStreamSynchronize(h->copy_stream));
    }
    RCCHK(raw.ensure(2 * slab_bytes));
    RCCHK(ensure_upload_pipeline(h, slab_bytes));
    // device slab b of THIS call's geometry: the second one starts where the first one of the LARGEST slab size seen ends, so that pieces of
    // different lengths (a short last row group) never overlap a slab that is still being read
    const size_t dev_half = raw.bytes / 2 / 256 * 256;
    for (long long r0 = 0; r0 < rows_pad; r0 += slab_rows) {
        const long long s = h->up_seq++;
        const int b = (int)(s & 1);
        const long long pad_here = std::min(slab_rows, rows_pad - r0);
        const long long real_here = std::max<long long>(0, std::min(slab_rows, rows - r0));
        char* dev_slab = (char*)raw.p + (size_t)b * dev_half;
        if (real_here > 0) {
            if (s >= 2) HIPCHK(hipEventSynchronize(h->ev_h2d[b]));            // the copy of slab s-2 has left this pinned buffer
            {   // gather: column j of the slab at offset j * stage_rows (the layout kernel's row stride 1, column stride stage_rows)
                char* pin = (char*)h->pin[b];
                const size_t seg = (size_t)real_here * in_es;
                const int nt = (int)std::min<long long>(h->opt_upload_threads, std::max<long long>(1, (long long)((size_t)d * seg >> 22)));   // >= 4 MiB per thread
                // rows [r0, r0 + real_here) of column j: the pieces of the chunks that overlap the slab, in row order
                int c0 = 0;
                while (chunk_start[c0 + 1] <= r0) ++c0;
                auto work = [&](int t) {
                    for (int j = t; j < d; j += nt) {
                        char* out = pin + (size_t)j * stage_rows * in_es;
                        long long r = r0;
                        for (int c = c0; r < r0 + real_here; ++c) {
                            const long long take = std::min(chunk_start[c + 1], r0 + real_here) - r;
                            if (take > 0) {
                                memcpy(out + (size_t)(r - r0) * in_es, (const char*)src.chunks[(size_t)c * d + j] + (size_t)(r - chunk_start[c]) * in_es,
                                       (size_t)take * in_es);
                                r += take;
                            }
                        }
                    }
                };
                (void)seg;
                if (nt <= 1) work(0);
                else {
                    // no C++ exception may cross the C ABI: if a thread cannot be started, this thread does that share itself
                    std::vector<std::thread> pool;
                    std::vector<int> mine;
                    mine.reserve(nt);               // nothing below may throw once a thread runs
                    mine.push_back(0);
                    pool.reserve(nt - 1);
                    for (int t = 1; t < nt; ++t) {
                        try { pool.emplace_back(work, t); } catch (...) { mine.push_back(t); }
                    }
                    for (int t : mine) work(t);
                    for (auto& th : pool) th.join();
                }
            }
            if (s >= 2) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_prepped[b], 0));   // the layout kernel of slab s-2 has read this device slab
            // column j sits at j * stage_rows: on a short (last) slab the copy ends with the last column's real rows
            const size_t copy_bytes = ((size_t)(d - 1) * stage_rows + (size_t)real_here) * in_es;
            HIPCHK(hipMemcpyAsync(dev_slab, h->pin[b], copy_bytes, hipMemcpyHostToDevice, h->copy_stream));
            HIPCHK(hipEventRecord(h->ev_h2d[b], h->copy_stream));
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_h2d[b], 0));
        }
        EvPair* e = ev_begin(h, 2);
        HIPCHK(nwk_launch_prep_strided(h->stream, dev_slab, in_dtype, 1, stage_rows, real_here, d, pad_here, dp, out_f16,
                                       (char*)dst + (size_t)r0 * dp * out_es, norms + r0, maxabs_bits));
        ev_end(h, e);
        HIPCHK(hipEventRecord(h->ev_prepped[b], h->stream));
    }
    if (finish) HIPCHK(hipStreamSynchronize(h->stream));        // the device slabs are idle again; (the caller's columns were free after the last gather)
    return NWKNN_OK;
}

// ---- fp16 pre-screen of the exact fp32 search --------------------------------------------------------
// max |x| of a prepared fp32 matrix; usable = finite and inside [2^-40, 2^40] (keeps every scaled threshold inside fp32)
static int matrix_maxabs(nwknn_ctx* h, const float* x, long long n, float* maxabs, bool* usable)
{
    RCCHK(h->scal.ensure(16));
    HIPCHK(hipMemsetAsync(h->scal.p, 0, 4, h->stream));
    HIPCHK(nwk_launch_maxabs(h->stream, x, n, h->scal.as<u32>()));
    u32 bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, h->scal.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float v;
    memcpy(&v, &bits, 4);
    *maxabs = v;
    *usable = bits < 0x7F800000u && v >= 0x1p-40f && v <= 0x1p40f;
    return NWKNN_OK;
}
// fp16 matrix: true when it holds no Inf/NaN
static int matrix_finite_f16(nwknn_ctx* h, const void* x, long long n, bool* finite)
{
    RCCHK(h->scal.ensure(16));
    HIPCHK(hipMemsetAsync(h->scal.p, 0, 4, h->stream));
    HIPCHK(nwk_launch_maxabs_f16(h->stream, x, n, h->scal.as<u32>()));
    u32 bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, h->scal.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *finite = bits < 0x7C00u;
    return NWKNN_OK;
}
// power of two that puts maxabs in [2^14, 2^15)
static float prescreen_scale(float maxabs)
{
    int e = 0;
    (void)frexpf(maxabs, &e);            // maxabs = m * 2^e, m in [0.5, 1)
    return ldexpf(1.0f, 15 - e);
}

static const float NWK_COS_SCALE = 0x1p14f;       // unit rows times this: elements in [-2^14, 2^14]

// (re)build the fp16 copy of the base in the flavour a search needs
static int ensure_base_copy(nwknn_ctx* h, int kind)
{
    if (h->pre_kind == kind) return NWKNN_OK;
    RCCHK(h->bp16.ensure((size_t)h->nb_pad_for_copy * h->dp16 * 2));
    EvPair* e = ev_begin(h, 2);
    if (kind == 2) HIPCHK(nwk_launch_half_copy16(h->stream, h->bp.p, h->nb_pad_for_copy, h->dp16, NWK_COS_SCALE, h->bn.as<float>(), h->bp16.p));
    else if (kind == 0) HIPCHK(nwk_launch_half_copy(h->stream, h->bp.as<float>(), h->nb_pad_for_copy, h->dp_for_copy, h->dp16, h->b_scale, nullptr, h->bp16.p));
    else HIPCHK(nwk_launch_half_copy(h->stream, h->bp.as<float>(), h->nb_pad_for_copy, h->dp_for_copy, h->dp16, NWK_COS_SCALE, h->bn.as<float>(), h->bp16.p));
    ev_end(h, e);
    h->pre_kind = kind;
    return NWKNN_OK;
}

// geometry of a prepared base and its buffers (what set_base decides before a byte is moved)
struct BasePlan { long long rows, rows_pad; int d, dp, f16; bool pre; };

static int base_plan(nwknn_ctx* h, int64_t rows, int32_t d, int dtype, bool cols, BasePlan* bp)
{
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16 && !(cols && dtype == NWKNN_DTYPE_F64))
        return fail(NWKNN_EINVAL, "nwknn_set_base: unsupported dtype %d", dtype);
    if (rows <= 0 || d <= 0) return fail(NWKNN_EINVAL, "nwknn_set_base: rows=%lld d=%d must be positive", (long long)rows, d);
    if (rows >= (1ll << 32) - 512) return fail(NWKNN_EINVAL, "nwknn_set_base: a shard holds at most 2^32-513 rows");
    RCCHK(use_device(h));
    bp->rows = rows; bp->d = d;
    bp->f16 = (dtype == NWKNN_DTYPE_F16 || h->opt_compute_f16) ? 1 : 0;
    bp->pre = h->opt_prescreen && !bp->f16;
    bp->rows_pad = round_up(rows, (bp->f16 || bp->pre) ? NWK_F16_TILE_ROWS : NWK_TILE_ROWS);
    bp->dp = (int)round_up(d, bp->f16 ? NWK_K_ALIGN_F16 : NWK_K_ALIGN);
    h->sb_rows = 0;                                                 // a half-streamed base is abandoned by any new base
    if (h->fw.active) {                                             // ... and so is a search that was riding it (its chunks may still read bp)
        h->fw.active = 0;
        if (h->fw_stream) HIPCHK(hipStreamSynchronize(h->fw_stream));
    }
    h->nb = 0;                                                      // ... and the previous base is gone as soon as its buffers may move
    RCCHK(h->bp.ensure((size_t)bp->rows_pad * bp->dp * (bp->f16 ? 2 : 4)));
    RCCHK(h->bn.ensure((size_t)bp->rows_pad * 4));
    h->ev_used = 0;
    h->bmax_fused = 0;
    if (bp->pre) {                                                  // max |x| rides the layout launches of this base (base_finish reads it)
        RCCHK(h->bmax.ensure(16));
        HIPCHK(hipMemsetAsync(h->bmax.p, 0, 4, h->stream));
        h->bmax_fused = 1;
    }
    return NWKNN_OK;
}

// the rows are laid out (bp / bn complete, padding included): checks and copies the searches need, then the base becomes current
static int base_finish(nwknn_ctx* h, const BasePlan& b)
{
    const long long rows_pad = b.rows_pad;
    const int d = b.d, dp = b.dp, f16 = b.f16;
    h->nb = b.rows; h->nb_pad = rows_pad; h->d = d; h->dp = dp; h->f16 = f16; h->ms_docs = 0;
    h->pre_ok = 0;
    h->f16_finite = 0;
    if (f16 && h->opt_f16_exact) {
        bool finite = false;
        RCCHK(matrix_finite_f16(h, h->bp.p, rows_pad * (long long)dp, &finite));
        h->f16_finite = finite ? 1 : 0;
        h->nb_pad_for_copy = rows_pad; h->dp_for_copy = dp; h->dp16 = dp; h->pre_kind = -1;      // cosine builds unit-row copies on demand
    }
    if (b.pre) {
        bool usable = false;
        if (h->bmax_fused) {                                        // folded into the layout launches: no pass over the prepared base
            u32 bits = 0;
            HIPCHK(hipMemcpyAsync(&bits, h->bmax.p, 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            memcpy(&h->b_maxabs, &bits, 4);
            usable = bits < 0x7F800000u && h->b_maxabs >= 0x1p-40f && h->b_maxabs <= 0x1p40f;
            h->bmax_fused = 0;
        } else {
            RCCHK(matrix_maxabs(h, h->bp.as<float>(), rows_pad * (long long)dp, &h->b_maxabs, &usable));
        }
        h->nb_pad_for_copy = rows_pad; h->dp_for_copy = dp;
        if (usable) {       // else (NaN/Inf/extreme magnitudes/all zero): searches on this base run the direct fp32 kernel
            h->dp16 = (int)round_up(d, NWK_K_ALIGN_F16);
            h->b_scale = prescreen_scale(h->b_maxabs);
            h->pre_ok = 1;
            h->pre_kind = -1;
            const int rc = ensure_base_copy(h, 0);  // the cosine flavour is made on demand by the first cosine search
            if (rc == NWKNN_ENOMEM) h->pre_ok = 0;  // no room for the fp16 copy next to the fp32 base: searches use the direct kernel
            else RCCHK(rc);
        }
    }
    if (h->opt_profile && !h->fw.active) {                          // (an attached search keeps the events: its own collection counts them all)
        HIPCHK(hipStreamSynchronize(h->stream));
        double t, c, p;
        ev_collect(h, &t, &c, &p);
        h->base_prep_ms = p;
    }
    return NWKNN_OK;
}

static int set_base_common(nwknn_ctx* h, const void* base, const ColumnSource* columns, int64_t rows, int32_t d, int dtype, int location)
{
    if (!h || (!base && !columns)) return fail(NWKNN_EINVAL, "nwknn_set_base: null handle or base pointer");
    const bool cols = columns != nullptr;
    if (!cols && location != NWKNN_HOST && location != NWKNN_DEVICE) return fail(NWKNN_EINVAL, "bad location %d", location);
    BasePlan b;
    RCCHK(base_plan(h, rows, d, dtype, cols, &b));
    unsigned int* const mx = h->bmax_fused ? h->bmax.as<u32>() : nullptr;
    if (cols) RCCHK(stage_and_prep_columns(h, *columns, dtype, b.f16, rows, d, b.rows_pad, b.dp, h->braw, h->bp.p, h->bn.as<float>(), true, mx));
    else RCCHK(stage_and_prep(h, base, location, dtype == NWKNN_DTYPE_F16, b.f16, rows, d, b.rows_pad, b.dp, h->braw, h->bp.p, h->bn.as<float>(), mx));
    return base_finish(h, b);
}

// ---- a base handed over in pieces ---------------------------------------------------------------------------------------------------
// The reference's tile loop uploads a tile after it has been read whole (cu_knn.py:201-211).  With mem_tune on 288 GB a "tile" is the
// whole file, so nothing reached the GPU before the last row group was decoded (round 4: kNN stage 8.4 s through files, 0.87 s from
// memory).  Here a tile is announced (begin), its row groups are appended as the reader threads finish them -- an append returns once the
// piece has been gathered into the pinned slabs, its H2D copy and layout kernel run under the decoding of the next piece -- and end pads
// the last tile, runs the checks / copies of nwknn_set_base and makes the base current.  The prepared layout is per-row, so the result is
// the layout nwknn_set_base* would have built from the concatenation: the same bits.
static int base_begin_common(nwknn_ctx* h, int64_t rows, int32_t d, int dtype)
{
    if (!h) return fail(NWKNN_EINVAL, "nwknn_base_begin: null handle");
    BasePlan b;
    RCCHK(base_plan(h, rows, d, dtype, false, &b));
    h->sb_rows = rows; h->sb_done = 0; h->sb_rows_pad = b.rows_pad; h->sb_d = d; h->sb_dp = b.dp; h->sb_f16 = b.f16; h->sb_pre = b.pre ? 1 : 0;
    return NWKNN_OK;
}

static int follow_advance(nwknn_ctx* h, bool final);

static int base_append_check(nwknn_ctx* h, const char* what, int64_t n)
{
    if (!h) return fail(NWKNN_EINVAL, "%s: null handle", what);
    if (h->sb_rows <= 0) return fail(NWKNN_ESTATE, "%s: call nwknn_base_begin first", what);
    if (n <= 0) return fail(NWKNN_EINVAL, "%s: a piece holds at least one row", what);
    if (h->sb_done + n > h->sb_rows) return fail(NWKNN_EINVAL, "%s: %lld + %lld rows exceed the %lld announced", what, h->sb_done, (long long)n, h->sb_rows);
    return use_device(h);
}

static int base_append_columns_common(nwknn_ctx* h, const ColumnSource& src, int64_t n, int dtype)
{
    RCCHK(base_append_check(h, "nwknn_base_append_column_chunks", n));
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16 && dtype != NWKNN_DTYPE_F64) return fail(NWKNN_EINVAL, "nwknn_base_append_column_chunks: unsupported dtype %d", dtype);
    const size_t out_es = h->sb_f16 ? 2 : 4;
    const int rc = stage_and_prep_columns(h, src, dtype, h->sb_f16, n, h->sb_d, n, h->sb_dp, h->braw, (char*)h->bp.p + (size_t)h->sb_done * h->sb_dp * out_es,
                                          h->bn.as<float>() + h->sb_done, false, h->bmax_fused ? h->bmax.as<u32>() : nullptr);
    if (rc != NWKNN_OK) { h->sb_rows = 0; return rc; }              // a failed piece abandons the base (streams may hold half of it)
    h->sb_done += n;
    return follow_advance(h, false);
}

static int base_append_rows_common(nwknn_ctx* h, const void* rows_ptr, int64_t n, int dtype, int location)
{
    RCCHK(base_append_check(h, "nwknn_base_append", n));
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16) return fail(NWKNN_EINVAL, "nwknn_base_append: unsupported dtype %d", dtype);
    if (location != NWKNN_HOST && location != NWKNN_DEVICE) return fail(NWKNN_EINVAL, "bad location %d", location);
    const size_t out_es = h->sb_f16 ? 2 : 4;
    const int rc = stage_and_prep(h, rows_ptr, location, dtype == NWKNN_DTYPE_F16, h->sb_f16, n, h->sb_d, n, h->sb_dp, h->braw,
                                  (char*)h->bp.p + (size_t)h->sb_done * h->sb_dp * out_es, h->bn.as<float>() + h->sb_done,
                                  h->bmax_fused ? h->bmax.as<u32>() : nullptr);
    if (rc != NWKNN_OK) { h->sb_rows = 0; return rc; }
    h->sb_done += n;
    return follow_advance(h, false);
}

static int base_end_common(nwknn_ctx* h)
{
    if (!h) return fail(NWKNN_EINVAL, "nwknn_base_end: null handle");
    if (h->sb_rows <= 0) return fail(NWKNN_ESTATE, "nwknn_base_end: call nwknn_base_begin first");
    RCCHK(use_device(h));
    if (h->sb_done != h->sb_rows) {
        const long long done = h->sb_done, want = h->sb_rows;
        h->sb_rows = 0;
        return fail(NWKNN_ESTATE, "nwknn_base_end: %lld of the %lld announced rows were appended", done, want);
    }
    BasePlan b{h->sb_rows, h->sb_rows_pad, h->sb_d, h->sb_dp, h->sb_f16, h->sb_pre != 0};
    h->sb_rows = 0;
    if (b.rows_pad > b.rows) {                                        // zero rows (norm 0) up to the tile granule, as a one-call set_base leaves them
        const size_t out_es = b.f16 ? 2 : 4;
        EvPair* e = ev_begin(h, 2);
        HIPCHK(nwk_launch_prep_strided(h->stream, h->bp.p, NWKNN_DTYPE_F32, 1, 1, 0, b.d, b.rows_pad - b.rows, b.dp, b.f16,
                                       (char*)h->bp.p + (size_t)b.rows * b.dp * out_es, h->bn.as<float>() + b.rows));
        ev_end(h, e);
    }
    if (h->fw.active) {                                               // the attached search walks what is left; the compute stream then waits for it
        const int rc = follow_advance(h, true);
        if (rc != NWKNN_OK) return rc;
    }
    return base_finish(h, b);                                         // its first reduction synchronises the compute stream: every piece has landed
}

extern "C" int nwknn_set_base(nwknn_handle h, const void* base, int64_t rows, int32_t d, int dtype, int location)
{
    if (!base) return fail(NWKNN_EINVAL, "nwknn_set_base: null handle or base pointer");
    return guarded([&] { return set_base_common(h, base, nullptr, rows, d, dtype, location); });
}

extern "C" int nwknn_set_base_columns(nwknn_handle h, const void* const* columns, int64_t rows, int32_t d, int dtype)
{
    if (!columns) return fail(NWKNN_EINVAL, "nwknn_set_base_columns: null column table");
    const int64_t one = rows;
    const ColumnSource src{columns, &one, 1};
    return guarded([&] { return set_base_common(h, nullptr, &src, rows, d, dtype, NWKNN_HOST); });
}

extern "C" int nwknn_set_base_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t rows,
                                            int32_t d, int dtype)
{
    if (!chunks || !chunk_rows || n_chunks <= 0) return fail(NWKNN_EINVAL, "nwknn_set_base_column_chunks: null or empty chunk table");
    const ColumnSource src{chunks, chunk_rows, n_chunks};
    return guarded([&] { return set_base_common(h, nullptr, &src, rows, d, dtype, NWKNN_HOST); });
}

extern "C" int nwknn_base_begin(nwknn_handle h, int64_t rows, int32_t d, int dtype)
{
    return guarded([&] { return base_begin_common(h, rows, d, dtype); });
}

extern "C" int nwknn_base_append(nwknn_handle h, const void* rows_ptr, int64_t n, int dtype, int location)
{
    if (!rows_ptr) return fail(NWKNN_EINVAL, "nwknn_base_append: null rows");
    return guarded([&] { return base_append_rows_common(h, rows_ptr, n, dtype, location); });
}

extern "C" int nwknn_base_append_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t n, int dtype)
{
    if (!chunks || !chunk_rows || n_chunks <= 0) return fail(NWKNN_EINVAL, "nwknn_base_append_column_chunks: null or empty chunk table");
    const ColumnSource src{chunks, chunk_rows, n_chunks};
    return guarded([&] { return base_append_columns_common(h, src, n, dtype); });
}

extern "C" int nwknn_base_end(nwknn_handle h)
{
    return guarded([&] { return base_end_common(h); });
}

// ---- search -----------------------------------------------------------------------------------------
static int make_plan(nwknn_ctx* h, int k, long long nq, bool pre, SearchPlan* pl)
{
    long long cap = h->opt_capacity;
    if (cap == 0) { cap = 1024; while (cap < 4ll * k) cap <<= 1; }
    if (cap < 2ll * k + NWK_TILE_ROWS) return fail(NWKNN_EINVAL, "capacity %lld too small for k=%d (need >= 2k+%d)", cap, k, NWK_TILE_ROWS);
    if (cap > NWK_MAX_CAPACITY) return fail(NWKNN_EINVAL, "k=%d needs capacity %lld > %d", k, cap, NWK_MAX_CAPACITY);
    const long long G = (h->f16 || pre) ? NWK_F16_TILE_ROWS : NWK_TILE_ROWS;  // base rows per tile = granule of every chunk
    long long first = h->opt_first_chunk ? round_up(h->opt_first_chunk, G) : cap / 2;
    first = std::max<long long>(first, round_up(k, G));
    first = std::min<long long>(first, cap / G * G);
    pl->cap = (int)cap;
    pl->first = first;
    pl->max_chunk = std::max<long long>(G, h->opt_max_chunk / G * G);
    pl->safe = (cap - k) / G * G;                              // a range this small cannot overflow a compacted list
    pl->granule = (int)G;
    if (pl->safe < G || first < k) return fail(NWKNN_EINVAL, "capacity %lld too small for k=%d on this compute path", cap, k);
    // query tile height: 256 rows (fewer DMA instructions and tile turnarounds per MFMA) once there are enough queries
    // to fill the chip with such tiles, else 128 (less padding waste)
    pl->tile_rows = (h->f16 || pre) ? NWK_F16_TILE_ROWS : (h->opt_tile_rows ? (int)h->opt_tile_rows : (nq >= 2048 ? 256 : 128));
    pl->pre = pre;
    pl->native16 = false;
    pl->copies16 = pre;
    pl->pre_scale = 1.f; pl->pre_eps_half = pl->pre_aq = pl->pre_ab = pl->pre_cst = 0.f;
    return NWKNN_OK;
}

static NwkTileParams tile_params(nwknn_ctx* h, const SearchPlan& pl, long long nq_pad)
{
    NwkTileParams p;
    memset(&p, 0, sizeof p);
    p.qp = h->qp.as<float>(); p.bp = h->bp.as<float>();          // halves on the fp16 path (the kernel casts)
    p.qn = h->qn.as<float>(); p.bn = h->bn.as<float>();
    p.tau = h->tau.as<u64>(); p.cnt = h->cnt.as<u32>(); p.buf = h->buf.as<u64>();
    p.qtile_list = nullptr;
    p.nb_valid = h->nb;
    p.nq_valid = (h->opt_diag == 32) ? 1 : h->stats.nq;   // dev diag 32: only query row 0 may list anything (timing of a filter that never fires)
    p.n_qtiles = (int)(nq_pad / pl.tile_rows);
    p.dp = h->dp; p.cap = pl.cap;
    p.sq_shift = h->opt_sq_shift >= 0 ? (int)h->opt_sq_shift : 3;      // auto is resolved per kernel in run_chunk
    p.walk_q_fastest = h->opt_walk_q > 0 ? 1 : 0;
    p.f16_shape = (int)h->opt_f16_shape;              // -1 (auto) is resolved in run_chunk, once the row length the kernel sees is known
    p.tiles_per_wg = (int)h->opt_f16_strip;
    p.dbg = h->opt_dbg_ptr ? (unsigned long long*)(uintptr_t)h->opt_dbg_ptr : nullptr;
    p.diag = NWK_DEV_BUILD ? (int)h->opt_diag : 0;
    return p;
}

static NwkCompactParams compact_params(nwknn_ctx* h, const SearchPlan& pl, int k)
{
    NwkCompactParams c;
    memset(&c, 0, sizeof c);
    c.buf = h->buf.as<u64>(); c.cnt = h->cnt.as<u32>(); c.tau = h->tau.as<u64>();
    c.ovf = h->ovf.as<u32>(); c.ovf_any = h->ovf_any.as<u32>();
    c.cnt0 = pl.pre ? h->cnt0.as<u32>() : nullptr;
    c.cap = pl.cap; c.k = k; c.qtile_rows = pl.tile_rows;
    return c;
}

#define NWK_DEV_BUILD_DIAG(h) (NWK_DEV_BUILD && (h)->opt_diag != 0)
static int run_chunk(nwknn_ctx* h, const SearchPlan& pl, int metric, int k, long long nq_pad, long long lo, long long rows, bool dense,
                     const int* qtile_list, int n_list_tiles, bool dedupe, bool rescan, bool last = true)
{
    NwkTileParams tp = tile_params(h, pl, nq_pad);
    // fp32 data: the dense first chunk runs the fp32 kernel -- and so do the filtered chunks that start below pre_direct_rows.  Right behind the
    // dense chunk the threshold is the k-th best of a few hundred rows: a quarter of all pairs pass, the pre-screen lists them and the
    // rescoring kernel then walks the fp32 chain for each, lane by lane (C3 shard, 1 536-row chunk: 1.5 ms of fp16 tiles + 5.2 ms of
    // rescoring, against 2.1 ms of fp32 MFMA tiles that produce the exact keys at once).  Auto: only the chunk that follows the dense one.
    const long long direct_until = h->opt_pre_direct_rows >= 0 ? h->opt_pre_direct_rows : pl.first + 1;
    const bool prescreen = pl.pre && (pl.native16 || (!dense && (rescan || lo >= direct_until)));
    // fp16 kernel build.  Auto (round 3, same-box A/B on 32k x 4M, tools/f16_forms.py, strips of 2 / 4 tiles): the ping-pong form from
    // 768 halves per row on (d = 768: 1 129 vs 1 084 TF for the streaming form, +4.2 %; 1024: +5.5 % over round 2's default; 1536 fp16
    // input: +4.2 % with the first version of the form; 2048: +6.3 % / +1.0 % over streaming / burst; 4096: +8.3 % / +3.4 %), the
    // streaming form below (d = 512: 1 005 vs 1 023; 384: 913 vs 953; 128: 461 vs 569 -- a tile is only a few slabs there and the
    // ping-pong form pays two phases per tile boundary); dense chunks run the burst form (the launcher sees to that).  Strips: 2 tiles
    // for the ping-pong form (1-3: equal within noise; 4: -2.5 %, 16: -5 %), 4 for the others.
    const int row_halves = (prescreen && pl.copies16) ? h->dp16 : h->dp;
    // Round 5: below 768 halves per row the two-issuing-waves form (5: one program for all eight waves, one mid-slab barrier, priority ladder --
    // nwk_tile_f16y.h) replaces the streaming form: same-box A/B on 32k x 4M pre-screened, forms 3 / 4 / 5 -- d = 128: 568 / 597 / 672 TF,
    // 256: 831 / 905 / 944, 384: 944 / 1 037 / 1 054, 512: 1 000 / 1 120 / 1 135, 640: 1 045 / 1 173 / 1 180; from 768 on forms 4 and 5 tie
    // within 0.2 % (768: 1 224 / 1 225, 1024: 1 268 / 1 267, 2048: 1 312 / 1 311; fp16 input d = 1536: 1 404 / 1 378) -- two different
    // structures at one rate: busy x clock is what the power limit hands out (PMC: 0.689 x 1.760 GHz against 0.673 x 1.793 GHz), see
    // tools/f16y_probe.hip and profiles/r05q_*, r05r_*.  The ping-pong form stays where it is not slower.
    const int f16_shape_auto = h->opt_f16_shape >= 0 ? (int)h->opt_f16_shape : (row_halves >= 768 ? 4 : 5);
    // Loose-threshold chunks.  After `lo` rows tau admits ~k / lo of the next ones (x 1.4 pre-screened): 27 % of all pairs in the chunk
    // that follows the dense one, 7 %, 1.7 %, ... as the walk quadruples.  There the tile kernel is bound by the list-length atomics --
    // device scope, resolved beyond L2, ~8 G/s for the whole chip whatever the addresses (C2 pre-screened, tools/chunk_trace.sh: 0.82 ms
    // for a chunk with 0.02 ms of MFMAs; 0.26 ms with the atomic compiled out; 0.65 ms with every counter in a cache line of its own) --
    // so those chunks run a build of the streaming kernel that reserves slots per query ROW and block (tile_epilogue16, AGG): 0.82 / 0.79 /
    // 1.03 ms -> 0.38 / 0.39 / 0.84 ms for C2's first three filtered chunks.  That epilogue costs the hot path registers (5 spills, -17 %
    // on tight-threshold chunks; in the ping-pong kernel -20 %), hence a build of its own and only below a pass rate of ~1 %.
    const bool agg = prescreen && !dense && !rescan && (h->opt_f16_agg >= 0 ? h->opt_f16_agg == 1 : (h->opt_f16_shape < 0 && lo <= 140ll * k));   // (a forced f16_mfma is honoured)
    const int f16_shape = agg ? 3 : f16_shape_auto;
    tp.f16_shape = f16_shape;
    tp.agg = agg ? 1 : 0;
    tp.tiles_per_wg = h->opt_f16_strip > 0 ? (int)h->opt_f16_strip : (f16_shape >= 4 ? 2 : 4);
    // super-tile shape and walk order (see tile_from_block).  Auto = the fastest setting measured, base tile index fastest in both:
    // fp32 kernel 8 x 8 (same-box A/B on C3: 0.27 % faster than 2 x 32 / query fastest, which moves 38 % fewer bytes over the
    // fabric); fp16 kernels 16 x 4 (C3 pre-screen tile kernel: +1.4 % over 8 x 8, +0.7 % over 32 x 2; 4 x 16 / query fastest moves
    // 31 % fewer bytes and is 1 % slower).  Neither kernel is bound by that traffic (the Infinity Cache serves it), so speed
    // decides; options super_q_shift / super_walk_q select the other walks.
    const bool fp32_kernel = !(h->f16 || prescreen);
    if (h->opt_sq_shift < 0) tp.sq_shift = fp32_kernel ? h->walk_auto_sq : 4;
    if (h->opt_walk_q < 0) tp.walk_q_fastest = fp32_kernel ? h->walk_auto_q : 0;
    // rows per base tile of the kernel the launchers below will pick: 128 for the fp32 kernel, 256 for the fp16 kernels
    const int btile = (h->f16 || prescreen) ? NWK_F16_TILE_ROWS : NWK_TILE_ROWS;
    if (rows % btile != 0) return fail(NWKNN_EINVAL, "internal: chunk of %lld rows is not a whole number of %d-row tiles", rows, btile);
    tp.bt0 = (int)(lo / btile);
    tp.n_btiles = (int)(rows / btile);
    tp.dense_col0 = lo;
    if (qtile_list) { tp.qtile_list = qtile_list; tp.n_qtiles = n_list_tiles; }
    EvPair* e = ev_begin(h, 0);
    if (prescreen) {
        if (pl.copies16) { tp.qp = h->qp16.as<float>(); tp.bp = h->bp16.as<float>(); tp.dp = h->dp16; }
        tp.pre_inv_scale = 1.0f / pl.pre_scale;
        tp.pre_scale = pl.pre_scale; tp.pre_eps_half = pl.pre_eps_half; tp.pre_aq = pl.pre_aq; tp.pre_ab = pl.pre_ab; tp.pre_cst = pl.pre_cst;
        tp.ctab = h->ctab.as<float>();
    }
    {   // HIP refuses a launch of 2^32 threads or more.  A chunk of very many query tiles x base tiles (hundreds of thousands of queries)
        // goes out as several launches over consecutive base-tile ranges: same thresholds, same lists, so the same result
        const long long per_launch = std::max<long long>(64, (NWK_MAX_WORKGROUPS_PER_LAUNCH / round_up(std::max(1, tp.n_qtiles), 64)) / 64 * 64);
        const int bt_first = tp.bt0, bt_total = tp.n_btiles;
        for (long long done = 0; done < bt_total; done += per_launch) {
            tp.bt0 = bt_first + (int)done;
            tp.n_btiles = (int)std::min<long long>(per_launch, bt_total - done);
            if (prescreen || h->f16) HIPCHK(nwk_launch_tile_f16(h->stream, tp, metric, dense, prescreen ? 1 : 0));
            else HIPCHK(nwk_launch_tile(h->stream, tp, metric, dense, pl.tile_rows, (int)h->opt_diag, (int)h->opt_tile_occ));
            h->stats.tile_launches += 1;                  // HIP launches, as a kernel trace counts them (a chunk of very many tiles is several)
            if (rescan) h->stats.rescan_launches += 1;
        }
        tp.bt0 = bt_first; tp.n_btiles = bt_total;
    }
    ev_end(h, e);
    if (!dense && !rescan) {                               // which kernel carried this launch, spelled as rocprofv3 prints it
        if (!(prescreen || h->f16)) {
            const bool occ3 = pl.tile_rows == 256 && !NWK_DEV_BUILD_DIAG(h) && h->opt_tile_occ >= 3;
            snprintf(h->tile_kernel, sizeof h->tile_kernel, "nwk_tile_kernel<%d,false,%d,%d,0,%d>", metric, pl.tile_rows, pl.tile_rows == 256 ? 16 : 32, occ3 ? 3 : 2);
        } else if (f16_shape == 3) snprintf(h->tile_kernel, sizeof h->tile_kernel, "nwk_tile_kernel_f16p<%d,%d,%d>", metric, prescreen ? 1 : 0, agg ? 1 : 0);
        else if (f16_shape == 4) snprintf(h->tile_kernel, sizeof h->tile_kernel, "nwk_tile_kernel_f16x<%d,%d>", metric, prescreen ? 1 : 0);
        else if (f16_shape == 5) snprintf(h->tile_kernel, sizeof h->tile_kernel, "nwk_tile_kernel_f16y<%d,%d>", metric, prescreen ? 1 : 0);
        else snprintf(h->tile_kernel, sizeof h->tile_kernel, "nwk_tile_kernel_f16s<%d,false,%d>", metric, prescreen ? 1 : 0);
    }
    bool deferred = false;
    if (prescreen) {                                       // listed pairs -> canonical fp32 keys, in place
        if (dense) {                                       // (exact fp16 mode) every slot of the dense chunk is a candidate: [0, rows)
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->cnt.p, (int)rows, (size_t)nq_pad, h->stream));
            HIPCHK(hipMemsetAsync(h->cnt0.p, 0, (size_t)nq_pad * 4, h->stream));
        }
        NwkRescoreParams rp;
        rp.qp = h->qp.as<float>(); rp.bp = h->bp.as<float>(); rp.qn = h->qn.as<float>(); rp.bn = h->bn.as<float>();
        rp.buf = h->buf.as<u64>(); rp.cnt = h->cnt.as<u32>(); rp.cnt0 = h->cnt0.as<u32>();
        rp.dp = h->dp; rp.cap = pl.cap; rp.k = k;
        // no pruning on a repair pass: it re-finds pairs that are already listed, and the k-th smallest upper bound over a list
        // with duplicates can lie below the k-th distinct distance.  (Round 4: the dense first chunk of exact fp16 mode IS pruned -- all of
        // its cap/2 slots carry estimates, and only the ~k of them whose lower bound reaches the k-th smallest upper bound can be among
        // the k nearest: 512 -> ~110 rescored pairs per query, a third of a C4 search's rescoring.)
        rp.prune = (h->opt_pre_prune && !dedupe) ? 1 : 0;
        // Deferred rescoring (round 5): a chunk that is not the last of its walk only PRUNES -- bounds of the listed estimates, the k-th
        // smallest upper bound U, entries whose lower bound exceeds it dropped, the rest packed as they are, tau = U -- and neither
        // rescores nor sorts (but for the queries the pass hands back, see defer_limit below: the compaction launch stays, its blocks
        // return at once for everyone else); the walk's last chunk rescores what is left of all of them (nwk_rescore_kernel, `defer`).
        // Not for the dense chunk of exact fp16 mode (it builds the exact prefix the pruning counts on), repair passes, or with pruning off.
        deferred = rp.prune && h->opt_pre_defer && !last && !dense && !rescan && !qtile_list;
        rp.defer = deferred ? 1 : 0;
        rp.tau = h->tau.as<u64>(); rp.cnt0_w = h->cnt0.as<u32>();
        rp.skip = nullptr; rp.defer_limit = 0;
        if (deferred) {
            // a query is left in the deferred state only while the next chunk's candidates still fit behind what it keeps; otherwise it is
            // rescored and sorted now (nwk_rescore_kernel, `stay`).  Two estimates of "the next chunk's candidates": ~1.6 k per unit of growth
            // (what a threshold at the k-th distance admits, the spread on top), and growth x 1.3 x what the query keeps NOW -- the kept entries
            // are the rows seen so far that lie under the bound, and a chunk `growth` times as long brings that many times as many (a query
            // sitting in a cloud of near-ties keeps hundreds: it must go back to exact thresholds before the cloud fills its list)
            RCCHK(h->dskip.ensure((size_t)nq_pad * 4));
            rp.skip = h->dskip.as<u32>();
            const long long by_k = pl.cap - (long long)(1.8 * (double)h->walk_growth * k) - 32;
            const long long by_kept = (long long)((double)pl.cap / (1.0 + 1.3 * (double)h->walk_growth));
            rp.defer_limit = (int)std::max<long long>(0, std::min(by_k, by_kept));
        }
        rp.rows16 = pl.native16 ? 1 : 0;
        rp.coop = h->opt_rescore_coop ? 1 : 0;
        rp.eps_half = pl.pre_eps_half; rp.aq = pl.pre_aq; rp.ab = pl.pre_ab; rp.cst = pl.pre_cst;
        rp.chk_scale = pl.chk_scale;
        rp.chk = h->chk.as<u64>();
        e = ev_begin(h, 3);
        HIPCHK(nwk_launch_rescore(h->stream, rp, metric, nq_pad));
        ev_end(h, e);
    }
    {   // algorithmic FLOPs of this chunk: real (unpadded) query rows x real base rows x 2d
        const long long real_rows = std::max<long long>(0, std::min<long long>(lo + rows, h->nb) - lo);
        const double q_rows = qtile_list ? (double)n_list_tiles * pl.tile_rows : (double)h->stats.nq;
        h->stats.tile_flops += 2.0 * q_rows * (double)real_rows * (double)h->d;
    }

    NwkCompactParams cp = compact_params(h, pl, k);
    if (deferred) cp.skip = h->dskip.as<u32>();           // the pruning pass packed the lists and set the thresholds: only queries it rescored after all are sorted
    cp.dense_count = (dense && !prescreen) ? (int)rows : 0;      // (a rescored dense chunk -- exact fp16 mode -- has its length in cnt: rows, or what the rescoring kernel kept)
    cp.dedupe = dedupe ? 1 : 0;
    cp.qtile_list = qtile_list;
    const long long nqc = qtile_list ? (long long)n_list_tiles * pl.tile_rows : nq_pad;
    e = ev_begin(h, 1);
    HIPCHK(nwk_launch_compact(h->stream, cp, nqc));
    ev_end(h, e);
    h->stats.compact_launches += 1;
    return NWKNN_OK;
}

// Overflow repair: re-walk row range [lo,hi) for the query tiles whose lists overflowed, halving the range
// until it fits.  A range of <= pl.safe rows can add at most pl.safe entries to a compacted (<= k) list, so it
// cannot overflow: the recursion terminates.
static int repair(nwknn_ctx* h, const SearchPlan& pl, int metric, int k, long long nq_pad)
{
    const int TR = pl.tile_rows;
    const long long n_tiles_all = nq_pad / TR;
    std::vector<u32> flags((size_t)nq_pad);
    HIPCHK(hipMemcpy(flags.data(), h->ovf.p, (size_t)nq_pad * 4, hipMemcpyDeviceToHost));
    std::vector<int> tiles;
    int nflag = 0;
    for (long long t = 0; t < n_tiles_all; ++t) {
        bool any = false;
        for (int j = 0; j < TR; ++j) if (flags[(size_t)t * TR + j]) { any = true; ++nflag; }
        if (any) tiles.push_back((int)t);
    }
    h->stats.overflowed_queries += nflag;
    RCCHK(h->qtiles.ensure((size_t)n_tiles_all * 4 * 2));

    struct Work { long long lo, hi; std::vector<int> tiles; };
    std::vector<Work> stack;
    for (long long lo = h->nb_pad; lo > 0;) {            // pushed high-to-low so ranges pop in ascending order
        const long long hi = lo;
        lo = std::max<long long>(0, hi - pl.max_chunk);
        stack.push_back({lo, hi, tiles});
    }
    while (!stack.empty()) {
        Work w = std::move(stack.back());
        stack.pop_back();
        if (w.tiles.empty()) continue;
        HIPCHK(hipMemsetAsync(h->ovf.p, 0, (size_t)nq_pad * 4, h->stream));
        HIPCHK(hipMemsetAsync(h->ovf_any.p, 0, 4, h->stream));
        HIPCHK(hipMemcpyAsync(h->qtiles.p, w.tiles.data(), w.tiles.size() * 4, hipMemcpyHostToDevice, h->stream));
        RCCHK(run_chunk(h, pl, metric, k, nq_pad, w.lo, w.hi - w.lo, false, h->qtiles.as<int>(), (int)w.tiles.size(), true, true));
        u32 any = 0;
        HIPCHK(hipMemcpyAsync(&any, h->ovf_any.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (!any) continue;
        if (w.hi - w.lo <= pl.safe) return fail(NWKNN_EHIP, "internal: list overflow on a %lld-row range (safe size %lld)", w.hi - w.lo, pl.safe);
        HIPCHK(hipMemcpy(flags.data(), h->ovf.p, (size_t)nq_pad * 4, hipMemcpyDeviceToHost));
        std::vector<int> again;
        for (int t : w.tiles) {
            bool a = false;
            for (int j = 0; j < TR; ++j) a |= flags[(size_t)t * TR + j] != 0;
            if (a) again.push_back(t);
        }
        const long long mid = w.lo + round_up((w.hi - w.lo) / 2, pl.granule);
        stack.push_back({mid, w.hi, again});
        stack.push_back({w.lo, mid, again});
    }
    return NWKNN_OK;
}

// The end of a walk: overflow repair, the pre-screen's invariant, the k nearest of every list in order, the copy to the caller.
static int search_finish(nwknn_ctx* h, const SearchPlan& pl, int metric, int k, long long nq, long long nq_pad, int64_t index_offset,
                         float* out_dist, int64_t* out_idx, int out_location, long long* violations)
{
    u32 any = 0;
    HIPCHK(hipMemcpyAsync(&any, h->ovf_any.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (any) RCCHK(repair(h, pl, metric, k, nq_pad));
    if (pl.pre) {                                                  // the invariant of the pre-screen, counted by every rescoring launch
        u64 chk[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(chk, h->chk.p, sizeof chk, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->stats.prescreen_checked = (int64_t)chk[0];
        float w;
        const u32 wb = (u32)chk[2];
        memcpy(&w, &wb, 4);
        h->stats.prescreen_worst = (double)w;
        if (chk[1] != 0) { *violations = (long long)chk[1]; return NWKNN_OK; }      // this attempt is discarded by the caller
    }

    float* od = out_dist;
    long long* oi = (long long*)out_idx;
    if (out_location == NWKNN_HOST) {
        RCCHK(h->out_d.ensure((size_t)nq * k * 4));
        RCCHK(h->out_i.ensure((size_t)nq * k * 8));
        od = h->out_d.as<float>(); oi = h->out_i.as<long long>();
    }
    HIPCHK(nwk_launch_finalize(h->stream, h->buf.as<u64>(), h->cnt.as<u32>(), pl.cap, k, nq, index_offset, od, oi, 0));
    if (h->opt_profile) HIPCHK(hipEventRecord(h->ev_end, h->stream));
    if (out_location == NWKNN_HOST) {
        HIPCHK(hipMemcpyAsync(out_dist, od, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(out_idx, oi, (size_t)nq * k * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->opt_profile) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
        h->stats.total_ms = ms;
        ev_collect(h, &h->stats.tile_ms, &h->stats.compact_ms, &h->stats.prep_ms, &h->stats.rescore_ms);
        h->stats.prep_ms += h->base_prep_ms;      // the base layout/norm pass that preceded this search
        h->base_prep_ms = 0.0;
    }
    return NWKNN_OK;
}

// One attempt of a search.  attempt 0 is the search as configured.  A later attempt follows a pre-screened one whose invariant failed
// (*violations > 0 on return: nothing was written to the outputs): fp32 data then runs the direct kernel; exact fp16 mode, whose
// rows have no fp32 form, widens its bound 16x per attempt and is checked again.
static int search_attempt(nwknn_ctx* h, const void* queries, const ColumnSource* qcolumns, int64_t nq, int dtype, int location, int32_t k, int metric,
                          int64_t index_offset, float* out_dist, int64_t* out_idx, int out_location, int attempt, long long* violations)
{
    *violations = 0;
    if (!h || (!queries && !qcolumns) || !out_dist || !out_idx) return fail(NWKNN_EINVAL, "nwknn_search: null argument");
    if (!h->bp.p || h->nb <= 0) return fail(NWKNN_ESTATE, "nwknn_search: call nwknn_set_base first");
    if (h->ms_docs > 0) return fail(NWKNN_ESTATE, "nwknn_search: the handle holds MaxSim documents; call nwknn_set_base for kNN");
    if (h->fw.active) return fail(NWKNN_ESTATE, "nwknn_search: an attached search holds the lists; call nwknn_search_collect first");
    if (nq <= 0) return fail(NWKNN_EINVAL, "nwknn_search: nq=%lld must be positive", (long long)nq);
    if (k <= 0 || k > h->nb) return fail(NWKNN_EINVAL, "nwknn_search: k=%d must satisfy 0 < k <= %lld base rows", k, h->nb);   // cu_knn.py:237
    if (k > NWK_MAX_K) return fail(NWKNN_EINVAL, "nwknn_search: k=%d exceeds the supported maximum %d", k, NWK_MAX_K);
    if (metric < 0 || metric > 2) return fail(NWKNN_EINVAL, "nwknn_search: unknown metric %d", metric);
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16 && !(qcolumns && dtype == NWKNN_DTYPE_F64))
        return fail(NWKNN_EINVAL, "nwknn_search: unsupported query dtype %d", dtype);
    if ((location != NWKNN_HOST && location != NWKNN_DEVICE) || (out_location != NWKNN_HOST && out_location != NWKNN_DEVICE))
        return fail(NWKNN_EINVAL, "nwknn_search: bad location");
    RCCHK(use_device(h));
    bool pre = h->pre_ok && !h->f16 && attempt == 0;
    const double widen = std::pow(16.0, (double)attempt);        // exact fp16 mode after a failed invariant
    // exact fp16 mode (fp16 compute path, option f16_exact): the fp16 MFMA kernel pre-screens, every listed pair is rescored with
    // the fp32 chain over the fp16 values.  (Cosine pre-screens on unit-row copies of the fp16 rows.)
    bool native16 = h->f16 && h->opt_f16_exact && h->f16_finite;
    SearchPlan pl;
    RCCHK(make_plan(h, k, nq, pre, &pl));
    // pre_check_ppm tightens the check of the search AS CONFIGURED (tests drive the fallback with it).  An attempt that follows a failed
    // one -- exact fp16 mode under a 16x wider bound -- is held against the whole of its own bound: otherwise a small ppm would fail every
    // attempt and turn a correct search into an error (16^3 x a few ppm is still below the deviations a healthy MFMA shows).
    pl.chk_scale = attempt == 0 ? (float)((double)h->opt_pre_check_ppm * 1e-6) : 1.0f;

    const long long nq_pad = round_up(nq, NWK_QPAD_ROWS);
    RCCHK(h->qp.ensure((size_t)nq_pad * h->dp * (h->f16 ? 2 : 4)));
    RCCHK(h->qn.ensure((size_t)nq_pad * 4));
    RCCHK(h->tau.ensure((size_t)nq_pad * 8));
    RCCHK(h->cnt.ensure((size_t)nq_pad * 4));
    RCCHK(h->ovf.ensure((size_t)nq_pad * 4));
    RCCHK(h->ovf_any.ensure(16));
    RCCHK(h->buf.ensure((size_t)nq_pad * pl.cap * 8));

    memset(&h->stats, 0, sizeof h->stats);
    h->stats.nq = nq; h->stats.nb = h->nb; h->stats.d = h->d; h->stats.k = k;
    h->ev_used = 0;
    if (h->opt_profile) HIPCHK(hipEventRecord(h->ev_begin, h->stream));

    if (qcolumns) RCCHK(stage_and_prep_columns(h, *qcolumns, dtype, h->f16, nq, h->d, nq_pad, h->dp, h->qraw, h->qp.p, h->qn.as<float>()));
    else RCCHK(stage_and_prep(h, queries, location, dtype == NWKNN_DTYPE_F16, h->f16, nq, h->d, nq_pad, h->dp, h->qraw, h->qp.p, h->qn.as<float>()));
    if (pre) {
        // scaled fp16 copy of the prepared queries and the constants of the error bound (DESIGN.md "pre-screen bound")
        float q_maxabs = 0.f;
        bool usable = false;
        RCCHK(matrix_maxabs(h, h->qp.as<float>(), nq_pad * (long long)h->dp, &q_maxabs, &usable));
        if (!usable) {
            pre = false;
            RCCHK(make_plan(h, k, nq, false, &pl));                          // same cap; granule/tile shape of the direct path
        } else {
            const bool cosine = metric == NWKNN_METRIC_COSINE;
            const float q_scale = cosine ? NWK_COS_SCALE : prescreen_scale(q_maxabs);
            RCCHK(ensure_base_copy(h, cosine ? 1 : 0));
            RCCHK(h->qp16.ensure((size_t)nq_pad * h->dp16 * 2));
            RCCHK(h->cnt0.ensure((size_t)nq_pad * 4));
            EvPair* e = ev_begin(h, 2);
            HIPCHK(nwk_launch_half_copy(h->stream, h->qp.as<float>(), nq_pad, h->dp, h->dp16, q_scale, cosine ? h->qn.as<float>() : nullptr, h->qp16.p));
            ev_end(h, e);
            HIPCHK(hipMemsetAsync(h->cnt0.p, 0, (size_t)nq_pad * 4, h->stream));
            const double u = 0x1p-24, dd = (double)h->dp16;
            // input rounding (1+2^-11)^2-1, the canonical chain (d u), the fp16 MFMA accumulation (allowed 4 d u), 2 % slack
            const double eps = 1.02 * (0x1p-10 * (1.0 + 0x1p-11) + 5.5 * dd * u);
            const double a_q = 0x1p-14 / (double)q_scale, a_b = 0x1p-14 / (double)h->b_scale;   // absolute error per element
            const double sd = sqrt((double)h->d);
            pl.pre_scale = q_scale * h->b_scale;
            pl.pre_eps_half = (float)(0.5 * eps * (double)h->opt_pre_eps_pct / 100.0);
            pl.pre_aq = (float)(1.01 * a_q * sd);       // times sqrt(|b|^2) >= |b|_1 / sqrt(d)
            pl.pre_ab = (float)(1.01 * a_b * sd);       // times sqrt(|q|^2)
            pl.pre_cst = (float)(1.01 * (double)h->d * a_q * a_b);
            if (cosine) {
                // unit rows: |S^-1 dot16 - cos| <= input rounding + fp16 MFMA sum (4 d u) + normalisation (d u + 8 u) + flushed elements;
                // |canonical cosine distance - (1 - cos)| <= 2.1 d u + 8 u.  One constant, no norm-dependent terms.
                const double a = 0x1p-14 / (double)NWK_COS_SCALE;
                pl.pre_scale = NWK_COS_SCALE * NWK_COS_SCALE;
                pl.pre_eps_half = 0.f; pl.pre_aq = 0.f; pl.pre_ab = 0.f;
                pl.pre_cst = (float)((1.02 * (0x1p-10 * (1.0 + 0x1p-11) + 7.5 * dd * u + 16.0 * u) + 2.02 * a * sd + (double)h->d * a * a)
                                     * (double)h->opt_pre_eps_pct / 100.0);
            }
        }
    }
    if (native16) {
        bool finite = false;
        RCCHK(matrix_finite_f16(h, h->qp.p, nq_pad * (long long)h->dp, &finite));
        if (finite) {
            RCCHK(h->cnt0.ensure((size_t)nq_pad * 4));
            HIPCHK(hipMemsetAsync(h->cnt0.p, 0, (size_t)nq_pad * 4, h->stream));
            const double u = 0x1p-24, dd = (double)h->dp;
            // the inputs are exact (the fp16 MFMA keeps subnormals: tools/f16_denorm.py): only the two summations differ --
            // the canonical chain (d u) and the MFMA's own (allowed 4 d u, measured ~2 u), 2 % slack
            pl.pre = true; pl.native16 = true;
            pl.pre_scale = 1.0f;
            pl.pre_eps_half = (float)(0.5 * 1.02 * 5.5 * dd * u * (double)h->opt_pre_eps_pct / 100.0 * widen);
            pl.pre_aq = pl.pre_ab = pl.pre_cst = 0.0f;
            pl.copies16 = false;
            if (metric == NWKNN_METRIC_COSINE) {           // unit-row copies of both sides; the constant bound of the fp32-data cosine path
                RCCHK(ensure_base_copy(h, 2));
                RCCHK(h->qp16.ensure((size_t)nq_pad * h->dp16 * 2));
                EvPair* e = ev_begin(h, 2);
                HIPCHK(nwk_launch_half_copy16(h->stream, h->qp.p, nq_pad, h->dp16, NWK_COS_SCALE, h->qn.as<float>(), h->qp16.p));
                ev_end(h, e);
                const double a = 0x1p-14 / (double)NWK_COS_SCALE, sd = sqrt((double)h->d);
                pl.copies16 = true;
                pl.pre_scale = NWK_COS_SCALE * NWK_COS_SCALE;
                pl.pre_eps_half = 0.f;
                pl.pre_cst = (float)((1.02 * (0x1p-10 * (1.0 + 0x1p-11) + 7.5 * dd * u + 16.0 * u) + 2.02 * a * sd + (double)h->d * a * a)
                                     * (double)h->opt_pre_eps_pct / 100.0 * widen);
            }
        } else {
            native16 = false;
        }
    }
    if (pl.pre) {                                                  // the column term of every base row under this search's constants (ping-pong kernel)
        RCCHK(h->chk.ensure(32));
        HIPCHK(hipMemsetAsync(h->chk.p, 0, 32, h->stream));
        RCCHK(h->ctab.ensure((size_t)h->nb_pad * 4));
        NwkTileParams tp = tile_params(h, pl, nq_pad);
        tp.pre_inv_scale = 1.0f / pl.pre_scale;
        tp.pre_scale = pl.pre_scale; tp.pre_eps_half = pl.pre_eps_half; tp.pre_aq = pl.pre_aq; tp.pre_ab = pl.pre_ab; tp.pre_cst = pl.pre_cst;
        EvPair* e = ev_begin(h, 2);
        HIPCHK(nwk_launch_col_terms(h->stream, tp, metric, h->nb_pad, h->ctab.as<float>()));
        ev_end(h, e);
    }
    h->stats.prescreen = (pre || native16) ? 1 : 0;
    HIPCHK(hipMemsetAsync(h->tau.p, 0xFF, (size_t)nq_pad * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->cnt.p, 0, (size_t)nq_pad * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->ovf.p, 0, (size_t)nq_pad * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->ovf_any.p, 0, 4, h->stream));

    // walk the base: one dense chunk, then filtered chunks that double (tau from n rows admits ~k*m/n of the next m)
    // chunk sizes: after n rows tau admits ~k*m/n of the next m, so a chunk of g*n rows adds ~g*k entries (x1.4 when pre-screened)
    // to a compacted list of k.  Auto: the largest g <= 3 that keeps that within 80 % of the list capacity (k = 100: 3, i.e. chunks
    // quadruple -- fewer launches, -1.6 % on a 1.25M-row shard, +3 % on C2 pre-screened, nothing on C3; k >= 200 at capacity 1024: 1).
    long long growth = h->opt_chunk_growth;
    if (growth <= 0) growth = std::max<long long>(1, std::min<long long>(3, (long long)((0.8 * pl.cap / k - 1.0) / 1.5)));
    // (deferred rescoring keeps ~0.4 k estimates beside the k exact entries between chunks and its threshold admits ~1.15 x as many)
    if (h->opt_chunk_growth <= 0 && pl.pre && h->opt_pre_defer && h->opt_pre_prune)
        growth = std::max<long long>(1, std::min<long long>(3, (long long)((0.8 * pl.cap / k - 1.4) / 1.6)));
    h->walk_growth = growth;
    long long seen = std::min(h->nb_pad, pl.first);
    h->walk_auto_q = 0; h->walk_auto_sq = 3;
    RCCHK(run_chunk(h, pl, metric, k, nq_pad, 0, seen, true, nullptr, 0, false, false));
    // Walk calibration (direct fp32 kernel, walk options at auto).  Which super-tile walk is faster is decided by the device's fabric, not by
    // the data: 8 x 8 base-fastest won by 0.27 % on the builder's boxes, 2 x 32 query-fastest moves 38 % fewer bytes beyond L2 -- and the
    // driver's box of round 4 ran the same kernel 2 % slower with the same MFMA ceiling.  So the first two full-size chunks are timed, one
    // under each walk, and the rest of the walk -- at least four more chunks, or it is not worth two event waits -- runs the faster.
    bool calibrate = !h->f16 && !pl.pre && h->opt_walk_q < 0 && h->opt_sq_shift < 0;
    while (seen < h->nb_pad) {
        const long long m = std::min(std::min(pl.max_chunk, seen * growth), h->nb_pad - seen);
        if (calibrate && m == pl.max_chunk && h->nb_pad - seen >= 6 * pl.max_chunk) {
            calibrate = false;
            for (int i = 0; i < 3; ++i) if (!h->ev_cal[i]) HIPCHK(hipEventCreate(&h->ev_cal[i]));
            HIPCHK(hipEventRecord(h->ev_cal[0], h->stream));
            RCCHK(run_chunk(h, pl, metric, k, nq_pad, seen, m, false, nullptr, 0, false, false));
            HIPCHK(hipEventRecord(h->ev_cal[1], h->stream));
            h->walk_auto_q = 1; h->walk_auto_sq = 1;
            RCCHK(run_chunk(h, pl, metric, k, nq_pad, seen + m, m, false, nullptr, 0, false, false));
            HIPCHK(hipEventRecord(h->ev_cal[2], h->stream));
            HIPCHK(hipEventSynchronize(h->ev_cal[2]));
            float a = 0.f, b = 0.f;
            HIPCHK(hipEventElapsedTime(&a, h->ev_cal[0], h->ev_cal[1]));
            HIPCHK(hipEventElapsedTime(&b, h->ev_cal[1], h->ev_cal[2]));
            // (both spans hold one tile launch and one compaction of equal size; the later chunk filters under a slightly tighter
            // threshold -- a few survivors fewer in a million rows -- which is far inside the differences this is after)
            const bool q_fastest = b < a;
            h->walk_auto_q = q_fastest ? 1 : 0; h->walk_auto_sq = q_fastest ? 1 : 3;
            h->stats.walk_calibrated = 1; h->stats.walk_query_fastest = q_fastest ? 1 : 0;
            h->stats.walk_ms_base_fastest = a; h->stats.walk_ms_query_fastest = b;
            seen += 2 * m;
            continue;
        }
        RCCHK(run_chunk(h, pl, metric, k, nq_pad, seen, m, false, nullptr, 0, false, false, seen + m >= h->nb_pad));
        seen += m;
    }
    h->walk_auto_q = 0; h->walk_auto_sq = 3;                       // repairs and later searches start from the default again

    return search_finish(h, pl, metric, k, nq, nq_pad, index_offset, out_dist, out_idx, out_location, violations);
}

static int search_common(nwknn_ctx* h, const void* queries, const ColumnSource* qcolumns, int64_t nq, int dtype, int location, int32_t k, int metric,
                         int64_t index_offset, float* out_dist, int64_t* out_idx, int out_location)
{
    long long discarded = 0;
    for (int attempt = 0;; ++attempt) {
        long long bad = 0;
        RCCHK(search_attempt(h, queries, qcolumns, nq, dtype, location, k, metric, index_offset, out_dist, out_idx, out_location, attempt, &bad));
        if (bad == 0) {
            h->stats.prescreen_fallbacks = attempt;
            h->stats.prescreen_violations = discarded;
            return NWKNN_OK;
        }
        discarded += bad;
        fprintf(stderr, "libnwknn: pre-screen invariant failed on %lld of %lld rescored pairs (|estimate - exact| above %lld ppm of the bound): "
                        "attempt %d discarded, searching again %s\n", bad, (long long)h->stats.prescreen_checked, (long long)h->opt_pre_check_ppm,
                attempt, h->f16 ? "under a 16x wider bound" : "on the direct fp32 kernel");                                  // never silent
        if (attempt >= 3) return fail(NWKNN_EHIP, "pre-screen invariant still violated after widening the bound %d times (%lld pairs)", attempt, bad);
    }
}

// ---- a search that rides a base handed over in pieces -----------------------------------------------------------------------------------
// Through files the base arrives at the speed of the parquet decoder (10-18 GB/s on the GPU box's host), the direct kernel walks it at the
// equivalent of ~27 GB/s for 10 000 queries -- and until round 5 it started only after the last piece had landed: 1.4 s of a 4 s stage with
// the decoder idle.  The walk needs nothing from rows it has not reached (a chunk reads its own rows and the running lists), so an attached
// search launches every chunk of its walk -- the ordinary walk with chunks of at most 2^18 rows -- as soon as the pieces that cover it
// have been appended: on a stream of its own, ordered behind the layout kernels by an event, beside the copies and layout
// kernels of the pieces that follow.  nwknn_base_end launches what is left, nwknn_search_collect repairs / sorts / copies out as
// nwknn_search does.  Direct fp32 path only: the pre-screen's scale and exact fp16 mode's finiteness check need the whole base
// (*attached = 0: search after nwknn_base_end as before; the pre-screened search of that shape takes 0.17 s, there is nothing to hide).
static int search_attach_common(nwknn_ctx* h, const void* queries, const ColumnSource* qcolumns, int64_t nq, int dtype, int location, int32_t k,
                                int metric, int64_t index_offset, int* attached)
{
    if (!h || (!queries && !qcolumns) || !attached) return fail(NWKNN_EINVAL, "nwknn_search_attach: null argument");
    *attached = 0;
    if (h->sb_rows <= 0 || h->sb_done != 0) return fail(NWKNN_ESTATE, "nwknn_search_attach: call it between nwknn_base_begin and the first piece");
    if (h->fw.active) return fail(NWKNN_ESTATE, "nwknn_search_attach: a search is attached to this base already");
    if (nq <= 0) return fail(NWKNN_EINVAL, "nwknn_search_attach: nq=%lld must be positive", (long long)nq);
    if (k <= 0 || k > h->sb_rows) return fail(NWKNN_EINVAL, "nwknn_search_attach: k=%d must satisfy 0 < k <= %lld base rows", k, h->sb_rows);
    if (k > NWK_MAX_K) return fail(NWKNN_EINVAL, "nwknn_search_attach: k=%d exceeds the supported maximum %d", k, NWK_MAX_K);
    if (metric < 0 || metric > 2) return fail(NWKNN_EINVAL, "nwknn_search_attach: unknown metric %d", metric);
    if (dtype != NWKNN_DTYPE_F32 && !(qcolumns && dtype == NWKNN_DTYPE_F64)) {
        if (dtype == NWKNN_DTYPE_F16) return NWKNN_OK;                        // fp16 queries belong to the fp16 compute path
        return fail(NWKNN_EINVAL, "nwknn_search_attach: unsupported query dtype %d", dtype);
    }
    if (location != NWKNN_HOST && location != NWKNN_DEVICE) return fail(NWKNN_EINVAL, "nwknn_search_attach: bad location");
    if (h->sb_f16 || h->sb_pre) return NWKNN_OK;
    RCCHK(use_device(h));
    if (!h->fw_stream) HIPCHK(hipStreamCreateWithFlags(&h->fw_stream, hipStreamNonBlocking));
    if (!h->ev_fw_ready) HIPCHK(hipEventCreateWithFlags(&h->ev_fw_ready, hipEventDisableTiming));
    if (!h->ev_fw_done) HIPCHK(hipEventCreateWithFlags(&h->ev_fw_done, hipEventDisableTiming));
    // the geometry the walk reads; nb stays 0 (no ordinary search on half a base) and is lent to the chunk launches by follow_advance
    h->nb_pad = h->sb_rows_pad; h->d = h->sb_d; h->dp = h->sb_dp; h->f16 = 0; h->ms_docs = 0; h->pre_ok = 0; h->f16_finite = 0;
    FollowState f;
    RCCHK(make_plan(h, k, nq, false, &f.pl));
    // What cannot start before the last piece has landed is the walk's last chunk: 2^20 rows are 0.21 s of the direct kernel for 10 000
    // queries of d = 1536, a quarter of that bound costs a few more launches (each with its compaction, ~50 us) and leaves 0.05 s.
    f.pl.max_chunk = std::max<long long>(f.pl.granule, std::min<long long>(f.pl.max_chunk, (1ll << 18) / f.pl.granule * f.pl.granule));
    f.k = k; f.metric = metric; f.nq = nq; f.nq_pad = round_up(nq, NWK_QPAD_ROWS); f.index_offset = index_offset; f.seen = 0; f.rows = h->sb_rows;
    f.growth = h->opt_chunk_growth;
    if (f.growth <= 0) f.growth = std::max<long long>(1, std::min<long long>(3, (long long)((0.8 * f.pl.cap / k - 1.0) / 1.5)));
    const long long nq_pad = f.nq_pad;
    RCCHK(h->qp.ensure((size_t)nq_pad * h->dp * 4));
    RCCHK(h->qn.ensure((size_t)nq_pad * 4));
    RCCHK(h->tau.ensure((size_t)nq_pad * 8));
    RCCHK(h->cnt.ensure((size_t)nq_pad * 4));
    RCCHK(h->ovf.ensure((size_t)nq_pad * 4));
    RCCHK(h->ovf_any.ensure(16));
    RCCHK(h->buf.ensure((size_t)nq_pad * f.pl.cap * 8));
    memset(&h->stats, 0, sizeof h->stats);
    h->stats.nq = nq; h->stats.nb = h->sb_rows; h->stats.d = h->d; h->stats.k = k;
    if (h->opt_profile) HIPCHK(hipEventRecord(h->ev_begin, h->stream));
    if (qcolumns) RCCHK(stage_and_prep_columns(h, *qcolumns, dtype, 0, nq, h->d, nq_pad, h->dp, h->qraw, h->qp.p, h->qn.as<float>()));
    else RCCHK(stage_and_prep(h, queries, location, false, 0, nq, h->d, nq_pad, h->dp, h->qraw, h->qp.p, h->qn.as<float>()));
    HIPCHK(hipMemsetAsync(h->tau.p, 0xFF, (size_t)nq_pad * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->cnt.p, 0, (size_t)nq_pad * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->ovf.p, 0, (size_t)nq_pad * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->ovf_any.p, 0, 4, h->stream));
    h->walk_auto_q = 0; h->walk_auto_sq = 3;
    f.active = 1;
    h->fw = f;
    *attached = 1;
    return NWKNN_OK;
}

// Launch every chunk of the attached search's walk whose rows have been appended (final: the base is complete, padding included).
static int follow_advance(nwknn_ctx* h, bool final)
{
    FollowState& f = h->fw;
    if (!f.active) return NWKNN_OK;
    struct Lend {                                                   // the walk's helpers read h->stream / h->nb: lent for the launches, always handed back
        nwknn_ctx* h; hipStream_t stream; long long nb; bool on = false;
        ~Lend() { if (on) { h->stream = stream; h->nb = nb; } }
    } lend{h, h->stream, h->nb};
    const long long avail = final ? h->nb_pad : h->sb_done;
    int rc = NWKNN_OK;
    while (f.seen < h->nb_pad) {
        const long long m = f.seen == 0 ? std::min(h->nb_pad, f.pl.first) : std::min(std::min(f.pl.max_chunk, f.seen * f.growth), h->nb_pad - f.seen);
        if (f.seen + m > avail) break;
        if (!lend.on) {                                             // everything appended so far has been enqueued on the compute stream
            HIPCHK(hipEventRecord(h->ev_fw_ready, h->stream));
            HIPCHK(hipStreamWaitEvent(h->fw_stream, h->ev_fw_ready, 0));
            lend.on = true;
            h->stream = h->fw_stream;
            h->nb = f.rows;
        }
        rc = run_chunk(h, f.pl, f.metric, f.k, f.nq_pad, f.seen, m, f.seen == 0, nullptr, 0, false, false);
        if (rc != NWKNN_OK) break;
        f.seen += m;
    }
    if (rc != NWKNN_OK) { f.active = 0; return rc; }
    if (final) {
        HIPCHK(hipEventRecord(h->ev_fw_done, h->fw_stream));
        HIPCHK(hipStreamWaitEvent(lend.stream, h->ev_fw_done, 0));
    }
    return NWKNN_OK;
}

static int search_collect_common(nwknn_ctx* h, float* out_dist, int64_t* out_idx, int out_location)
{
    if (!h || !out_dist || !out_idx) return fail(NWKNN_EINVAL, "nwknn_search_collect: null argument");
    if (out_location != NWKNN_HOST && out_location != NWKNN_DEVICE) return fail(NWKNN_EINVAL, "nwknn_search_collect: bad location");
    FollowState& f = h->fw;
    if (!f.active) return fail(NWKNN_ESTATE, "nwknn_search_collect: no search is attached (nwknn_search_attach reported 0, or the base was abandoned)");
    if (h->sb_rows > 0 || h->nb <= 0 || f.seen < h->nb_pad) return fail(NWKNN_ESTATE, "nwknn_search_collect: call nwknn_base_end first");
    RCCHK(use_device(h));
    f.active = 0;
    h->base_prep_ms = 0.0;                                          // the layout passes lie inside this search's span
    long long violations = 0;
    return search_finish(h, f.pl, f.metric, f.k, f.nq, f.nq_pad, f.index_offset, out_dist, out_idx, out_location, &violations);
}

extern "C" int nwknn_search_attach(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, int32_t k, int metric, int64_t index_offset,
                                   int* attached)
{
    if (!queries) return fail(NWKNN_EINVAL, "nwknn_search_attach: null argument");
    return guarded([&] { return search_attach_common(h, queries, nullptr, nq, dtype, location, k, metric, index_offset, attached); });
}

extern "C" int nwknn_search_attach_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t nq,
                                                 int dtype, int32_t k, int metric, int64_t index_offset, int* attached)
{
    if (!chunks || !chunk_rows || n_chunks <= 0) return fail(NWKNN_EINVAL, "nwknn_search_attach_column_chunks: null or empty chunk table");
    const ColumnSource src{chunks, chunk_rows, n_chunks};
    return guarded([&] { return search_attach_common(h, nullptr, &src, nq, dtype, NWKNN_HOST, k, metric, index_offset, attached); });
}

extern "C" int nwknn_search_collect(nwknn_handle h, float* out_dist, int64_t* out_idx, int out_location)
{
    return guarded([&] { return search_collect_common(h, out_dist, out_idx, out_location); });
}

extern "C" int nwknn_search(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, int32_t k, int metric, int64_t index_offset,
                            float* out_dist, int64_t* out_idx, int out_location)
{
    if (!queries) return fail(NWKNN_EINVAL, "nwknn_search: null argument");
    return guarded([&] { return search_common(h, queries, nullptr, nq, dtype, location, k, metric, index_offset, out_dist, out_idx, out_location); });
}

extern "C" int nwknn_search_columns(nwknn_handle h, const void* const* columns, int64_t nq, int dtype, int32_t k, int metric, int64_t index_offset,
                                    float* out_dist, int64_t* out_idx, int out_location)
{
    if (!columns) return fail(NWKNN_EINVAL, "nwknn_search_columns: null column table");
    const int64_t one = nq;
    const ColumnSource src{columns, &one, 1};
    return guarded([&] { return search_common(h, nullptr, &src, nq, dtype, NWKNN_HOST, k, metric, index_offset, out_dist, out_idx, out_location); });
}

extern "C" int nwknn_search_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t nq, int dtype,
                                          int32_t k, int metric, int64_t index_offset, float* out_dist, int64_t* out_idx, int out_location)
{
    if (!chunks || !chunk_rows || n_chunks <= 0) return fail(NWKNN_EINVAL, "nwknn_search_column_chunks: null or empty chunk table");
    const ColumnSource src{chunks, chunk_rows, n_chunks};
    return guarded([&] { return search_common(h, nullptr, &src, nq, dtype, NWKNN_HOST, k, metric, index_offset, out_dist, out_idx, out_location); });
}

// ---- MaxSim late interaction (new: BASELINE config 5; the reference's `ck` is flat token kNN) ----------------
// copies the prepared base and its norms into larger zero-filled buffers; the caller owns nb2 / nn2 (and frees them on failure)
static int grow_padded(nwknn_ctx* h, DevBuf& nb2, DevBuf& nn2, long long pad)
{
    RCCHK(nb2.ensure((size_t)pad * h->dp * 4));
    RCCHK(nn2.ensure((size_t)pad * 4));
    HIPCHK(hipMemsetAsync(nb2.p, 0, (size_t)pad * h->dp * 4, h->stream));
    HIPCHK(hipMemcpyAsync(nb2.p, h->bp.p, (size_t)h->nb_pad * h->dp * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemsetAsync(nn2.p, 0, (size_t)pad * 4, h->stream));
    HIPCHK(hipMemcpyAsync(nn2.p, h->bn.p, (size_t)h->nb_pad * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return NWKNN_OK;
}

static int nwknn_maxsim_set_docs_impl(nwknn_handle h, const void* doc_tokens, int64_t n_docs, int32_t tokens_per_doc, int32_t d, int dtype,
                                     int location)
{
    if (!h || !doc_tokens) return fail(NWKNN_EINVAL, "nwknn_maxsim_set_docs: null argument");
    if (tokens_per_doc != NWK_MS_TOKENS_PER_DOC) return fail(NWKNN_EINVAL, "nwknn_maxsim_set_docs: tokens_per_doc must be %d", NWK_MS_TOKENS_PER_DOC);
    if (dtype != NWKNN_DTYPE_F32) return fail(NWKNN_EINVAL, "nwknn_maxsim_set_docs: float32 tokens only");
    if (n_docs <= 0 || n_docs > (1ll << 25)) return fail(NWKNN_EINVAL, "nwknn_maxsim_set_docs: n_docs out of range");
    const long long save = h->opt_compute_f16, save_pre = h->opt_prescreen;
    h->opt_compute_f16 = 0; h->opt_prescreen = 0;
    // 4 docs = 256 token rows per tile: pad the row count to a multiple of 256 by declaring whole (zero) documents
    const int rc = set_base_common(h, doc_tokens, nullptr, n_docs * tokens_per_doc, d, dtype, location);
    h->opt_compute_f16 = save; h->opt_prescreen = save_pre;
    RCCHK(rc);
    const long long pad = round_up(h->nb, 256);
    if (pad > h->nb_pad) {      // set_base pads to 128 rows; MaxSim tiles need 256: grow and clear the tail
        DevBuf nb2, nn2;
        const int rcg = grow_padded(h, nb2, nn2, pad);
        if (rcg != NWKNN_OK) { nb2.release(); nn2.release(); return rcg; }
        h->bp.release(); h->bn.release();
        h->bp = nb2; h->bn = nn2;
        h->nb_pad = pad;
    }
    h->ms_docs = n_docs;
    h->ms_pre = 0;
    if (save_pre) {             // scaled fp16 copy of the doc tokens for the pre-screen (same conditions as for kNN)
        bool usable = false;
        RCCHK(matrix_maxabs(h, h->bp.as<float>(), h->nb_pad * (long long)h->dp, &h->b_maxabs, &usable));
        if (usable) {
            h->nb_pad_for_copy = h->nb_pad; h->dp_for_copy = h->dp;
            h->dp16 = (int)round_up(d, NWK_K_ALIGN_F16);
            h->b_scale = prescreen_scale(h->b_maxabs);
            h->pre_kind = -1;
            const int rc2 = ensure_base_copy(h, 0);
            if (rc2 == NWKNN_OK) h->ms_pre = 1;
            else if (rc2 != NWKNN_ENOMEM) return rc2;
        }
    }
    return NWKNN_OK;
}

extern "C" int nwknn_maxsim_set_docs(nwknn_handle h, const void* doc_tokens, int64_t n_docs, int32_t tokens_per_doc, int32_t d, int dtype, int location)
{
    return guarded([&] { return nwknn_maxsim_set_docs_impl(h, doc_tokens, n_docs, tokens_per_doc, d, dtype, location); });
}

static int maxsim_chunk(nwknn_ctx* h, int cap, int k, long long nq, long long nq_pad, long long doc_lo, long long docs, bool dense, bool dedupe,
                        bool pre)
{
    NwkTileParams tp;
    memset(&tp, 0, sizeof tp);
    tp.qp = h->qp.as<float>(); tp.bp = h->bp.as<float>(); tp.qn = h->qn.as<float>(); tp.bn = h->bn.as<float>();
    tp.tau = h->tau.as<u64>(); tp.cnt = h->cnt.as<u32>(); tp.buf = h->buf.as<u64>();
    tp.nb_valid = h->ms_docs; tp.dense_col0 = doc_lo;
    tp.n_qtiles = (int)(nq_pad / 4); tp.bt0 = (int)(doc_lo / 4); tp.n_btiles = (int)(docs / 4);
    tp.dp = h->dp; tp.cap = cap; tp.sq_shift = h->opt_sq_shift >= 0 ? (int)h->opt_sq_shift : 3; tp.walk_q_fastest = h->opt_walk_q > 0 ? 1 : 0;
    const bool prescreen = pre && !dense;                // the dense first chunk runs the fp32 kernel (exact keys)
    tp.tiles_per_wg = (int)h->opt_ms_strip;
    tp.f16_shape = h->opt_ms_waves == 4 ? 4 : 8;          // MaxSim ring kernel: waves per workgroup (the field is the kNN kernels' otherwise)
    tp.agg = (int)h->opt_ms_form;                         // ... and which build of the eight-wave tile (option ms_form)
    EvPair* e = ev_begin(h, 0);
    if (prescreen) {
        // the ring kernel streams documents only (the query fragments live in registers): the more query groups of an XCD walk the same
        // strip at the same time, the fewer bytes cross the fabric -- 64 x 1 super-tiles instead of 8 x 8: +3.3 % on C5 (tools/ms_order_sweep.py)
        if (h->opt_sq_shift < 0 && h->dp16 <= 128 && h->opt_ms_strip != 1) {
            int sh = 3;
            const int q_tiles = h->opt_ms_waves == 4 ? tp.n_qtiles : (tp.n_qtiles + 1) / 2;
            while (sh < 6 && (1 << sh) < q_tiles) ++sh;
            tp.sq_shift = sh;
        }
        tp.qp = h->qp16.as<float>(); tp.bp = h->bp16.as<float>(); tp.dp = h->dp16;
        tp.nq_valid = nq; tp.ms_rq = h->ms_rq.as<float>(); tp.ms_cd = h->ms_cd.as<float>();
        tp.pre_scale = h->ms_scale; tp.pre_inv_scale = 1.0f / h->ms_scale;
        HIPCHK(nwk_launch_maxsim_pre(h->stream, tp));
        snprintf(h->tile_kernel, sizeof h->tile_kernel, "%s", (h->dp16 <= 128 && h->opt_ms_strip != 1) ? (h->opt_ms_waves == 4 ? "nwk_maxsim_pre_ring16_kernel<4,1>" : (h->opt_ms_form == 3 ? (h->dp16 == 128 ? "nwk_maxsim_pre_tile8_kernel<4>" : "nwk_maxsim_pre_tile8_kernel<2>") : (h->opt_ms_form == 2 ? "nwk_maxsim_pre_ring16_kernel<8,2>" : "nwk_maxsim_pre_ring16_kernel<8,1>"))) : "nwk_maxsim_pre_kernel");
    } else {
        HIPCHK(nwk_launch_maxsim(h->stream, tp, dense));
        if (!dense) snprintf(h->tile_kernel, sizeof h->tile_kernel, "nwk_maxsim_kernel<false>");
    }
    ev_end(h, e);
    if (prescreen) {
        NwkMsRescoreParams rp;
        rp.qp = h->qp.as<float>(); rp.bp = h->bp.as<float>(); rp.buf = h->buf.as<u64>(); rp.cnt = h->cnt.as<u32>(); rp.cnt0 = h->cnt0.as<u32>();
        rp.rq = h->ms_rq.as<float>(); rp.cd = h->ms_cd.as<float>();
        rp.dp = h->dp; rp.cap = cap; rp.k = k; rp.prune = (h->opt_pre_prune && !dedupe) ? 1 : 0;
        rp.dbg = h->opt_dbg_ptr ? (unsigned long long*)(uintptr_t)h->opt_dbg_ptr : nullptr;
        rp.surv = h->ms_surv.as<u32>(); rp.nsurv = h->ms_nsurv.as<u32>(); rp.rescore_valu = (int)h->opt_ms_rescore_valu;
        rp.chk_scale = (float)((double)h->opt_pre_check_ppm * 1e-6); rp.chk = h->chk.as<u64>();
        // enough (query, slice) waves to fill the chip a few times over
        const int slices = (int)std::max<long long>(1, std::min<long long>(32, (8ll * h->num_cu * 4 + nq_pad - 1) / nq_pad));
        e = ev_begin(h, 3);
        HIPCHK(nwk_launch_maxsim_rescore(h->stream, rp, nq_pad, slices));
        ev_end(h, e);
    }
    h->stats.tile_launches += 1;
    const long long real_docs = std::max<long long>(0, std::min<long long>(doc_lo + docs, h->ms_docs) - doc_lo);
    h->stats.tile_flops += 2.0 * (double)nq * NWK_MS_TOKENS_PER_QUERY * (double)real_docs * NWK_MS_TOKENS_PER_DOC * (double)h->d;
    NwkCompactParams cp;
    memset(&cp, 0, sizeof cp);
    cp.buf = h->buf.as<u64>(); cp.cnt = h->cnt.as<u32>(); cp.tau = h->tau.as<u64>(); cp.ovf = h->ovf.as<u32>(); cp.ovf_any = h->ovf_any.as<u32>();
    cp.cap = cap; cp.k = k; cp.qtile_rows = 4; cp.dense_count = dense ? (int)docs : 0; cp.dedupe = dedupe ? 1 : 0;
    cp.cnt0 = pre ? h->cnt0.as<u32>() : nullptr;
    e = ev_begin(h, 1);
    HIPCHK(nwk_launch_compact(h->stream, cp, nq_pad));
    ev_end(h, e);
    h->stats.compact_launches += 1;
    return NWKNN_OK;
}

static int maxsim_search_attempt(nwknn_handle h, const void* query_tokens, int64_t n_queries, int32_t tokens_per_query, int dtype, int location,
                                 int32_t k, int64_t doc_offset, float* out_score, int64_t* out_doc, int out_location, int attempt, long long* violations)
{
    *violations = 0;
    if (!h || !query_tokens || !out_score || !out_doc) return fail(NWKNN_EINVAL, "nwknn_maxsim_search: null argument");
    if (h->ms_docs <= 0) return fail(NWKNN_ESTATE, "nwknn_maxsim_search: call nwknn_maxsim_set_docs first");
    if (tokens_per_query != NWK_MS_TOKENS_PER_QUERY) return fail(NWKNN_EINVAL, "nwknn_maxsim_search: tokens_per_query must be %d", NWK_MS_TOKENS_PER_QUERY);
    if (dtype != NWKNN_DTYPE_F32) return fail(NWKNN_EINVAL, "nwknn_maxsim_search: float32 tokens only");
    if (n_queries <= 0) return fail(NWKNN_EINVAL, "nwknn_maxsim_search: n_queries must be positive");
    if (k <= 0 || k > h->ms_docs || k > 1024) return fail(NWKNN_EINVAL, "nwknn_maxsim_search: k=%d must satisfy 0 < k <= min(%lld docs, 1024)", k, h->ms_docs);
    if ((location != NWKNN_HOST && location != NWKNN_DEVICE) || (out_location != NWKNN_HOST && out_location != NWKNN_DEVICE))
        return fail(NWKNN_EINVAL, "nwknn_maxsim_search: bad location");
    RCCHK(use_device(h));
    long long cap = h->opt_capacity;
    if (cap == 0) { cap = 1024; while (cap < 4ll * k) cap <<= 1; }
    if (cap < 2ll * k + 4 || cap > 4096) return fail(NWKNN_EINVAL, "nwknn_maxsim_search: capacity %lld unusable for k=%d", cap, k);
    const long long nq = n_queries, nq_pad = round_up(nq, 8);               // 8 queries = 256 query-token rows (NWK_QPAD_ROWS)
    const long long qrows = nq * NWK_MS_TOKENS_PER_QUERY, qrows_pad = nq_pad * NWK_MS_TOKENS_PER_QUERY;
    RCCHK(h->qp.ensure((size_t)qrows_pad * h->dp * 4));
    RCCHK(h->qn.ensure((size_t)qrows_pad * 4));
    RCCHK(h->tau.ensure((size_t)nq_pad * 8)); RCCHK(h->cnt.ensure((size_t)nq_pad * 4)); RCCHK(h->ovf.ensure((size_t)nq_pad * 4));
    RCCHK(h->ovf_any.ensure(16));
    RCCHK(h->buf.ensure((size_t)nq_pad * cap * 8));
    memset(&h->stats, 0, sizeof h->stats);
    h->stats.nq = nq; h->stats.nb = h->ms_docs; h->stats.d = h->d; h->stats.k = k;
    h->ev_used = 0;
    if (h->opt_profile) HIPCHK(hipEventRecord(h->ev_begin, h->stream));
    RCCHK(stage_and_prep(h, query_tokens, location, 0, 0, qrows, h->d, qrows_pad, h->dp, h->qraw, h->qp.p, h->qn.as<float>()));
    bool pre = h->ms_pre != 0 && attempt == 0;                     // a later attempt follows a failed invariant: the direct fp32 MaxSim kernel
    if (pre) {
        RCCHK(h->chk.ensure(32));
        HIPCHK(hipMemsetAsync(h->chk.p, 0, 32, h->stream));
        float q_maxabs = 0.f;
        bool usable = false;
        RCCHK(matrix_maxabs(h, h->qp.as<float>(), qrows_pad * (long long)h->dp, &q_maxabs, &usable));
        if (!usable) pre = false;
        else {
            const float q_scale = prescreen_scale(q_maxabs);
            const long long docs_all = h->nb_pad / NWK_MS_TOKENS_PER_DOC;
            RCCHK(h->qp16.ensure((size_t)qrows_pad * h->dp16 * 2));
            RCCHK(h->cnt0.ensure((size_t)nq_pad * 4));
            RCCHK(h->ms_rq.ensure((size_t)nq_pad * 4));
            RCCHK(h->ms_cd.ensure((size_t)docs_all * 4));
            RCCHK(h->ms_surv.ensure((size_t)nq_pad * cap * 4));
            RCCHK(h->ms_nsurv.ensure((size_t)nq_pad * 4));
            const double u = 0x1p-24, dd = (double)h->dp16;
            const double eps = 1.02 * (0x1p-10 * (1.0 + 0x1p-11) + 5.5 * dd * u);
            const double a_q = 0x1p-14 / (double)q_scale, a_b = 0x1p-14 / (double)h->b_scale, sd = sqrt((double)h->d);
            h->ms_scale = q_scale * h->b_scale;
            h->ms_eps_half = (float)(0.5 * eps);
            h->ms_aq = (float)(1.01 * a_q * sd);        // times sqrt(|doc token|^2)
            h->ms_ab = (float)(1.01 * a_b * sd);        // times sqrt(|query token|^2)
            h->ms_cst = (float)(1.01 * (double)h->d * a_q * a_b);
            EvPair* e = ev_begin(h, 2);
            HIPCHK(nwk_launch_half_copy(h->stream, h->qp.as<float>(), qrows_pad, h->dp, h->dp16, q_scale, nullptr, h->qp16.p));
            HIPCHK(nwk_launch_maxsim_bounds(h->stream, h->qn.as<float>(), nq_pad, NWK_MS_TOKENS_PER_QUERY, h->ms_eps_half, h->ms_ab, h->ms_cst, 0,
                                            h->ms_rq.as<float>()));
            HIPCHK(nwk_launch_maxsim_bounds(h->stream, h->bn.as<float>(), docs_all, NWK_MS_TOKENS_PER_DOC, h->ms_eps_half, h->ms_aq, 0.0f, 1,
                                            h->ms_cd.as<float>()));
            ev_end(h, e);
            HIPCHK(hipMemsetAsync(h->cnt0.p, 0, (size_t)nq_pad * 4, h->stream));
        }
    }
    h->stats.prescreen = pre ? 1 : 0;
    HIPCHK(hipMemsetAsync(h->tau.p, 0xFF, (size_t)nq_pad * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->cnt.p, 0, (size_t)nq_pad * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->ovf.p, 0, (size_t)nq_pad * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->ovf_any.p, 0, 4, h->stream));

    const long long docs_pad = h->nb_pad / NWK_MS_TOKENS_PER_DOC;           // multiple of 4
    // largest chunk: option max_chunk / 4 documents (2^18 by default; round 2a: max_chunk token rows = 2^14 documents, 65 launches per C5
    // search instead of 11: +3 %, tools/ms_chunk_sweep.py), bounded so that document tiles x query groups stays a legal grid
    long long max_chunk = std::max<long long>(4, (h->opt_max_chunk / 4) / 4 * 4);
    {   // HIP refuses launches of 2^32 threads or more: workgroups (256 threads) <= 2^23.  The fp32 and the plain pre-screen kernel take one
        // document tile per workgroup, the strip / ring kernels at least 8 (the launcher's shortest automatic strip)
        const long long tiles_per_wg = (pre && h->dp16 <= 128 && h->opt_ms_strip != 1) ? (h->opt_ms_strip > 1 ? h->opt_ms_strip : 8) : 1;
        const long long wg_budget = (1ll << 23) / std::max<long long>(1, round_up(nq_pad / 4, 64));      // per query group, super-tile padding included
        max_chunk = std::max<long long>(4, std::min<long long>(max_chunk, wg_budget * tiles_per_wg / 2 * 4));
    }
    long long first = h->opt_first_chunk ? round_up(h->opt_first_chunk, 4) : cap / 2;
    first = std::min<long long>(std::max<long long>(first, round_up(k, 4)), cap / 4 * 4);
    long long seen = std::min(docs_pad, first);
    RCCHK(maxsim_chunk(h, (int)cap, k, nq, nq_pad, 0, seen, true, false, pre));
    while (seen < docs_pad) {
        const long long m = std::min(std::min(max_chunk, seen), docs_pad - seen);
        RCCHK(maxsim_chunk(h, (int)cap, k, nq, nq_pad, seen, m, false, false, pre));
        seen += m;
    }
    u32 any = 0;
    HIPCHK(hipMemcpyAsync(&any, h->ovf_any.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (any) {
        // some list overflowed: walk all docs again in ranges that cannot overflow a compacted list (<= cap-k new entries),
        // dropping the duplicates the second walk re-finds
        const long long safe = (cap - k) / 4 * 4;
        std::vector<u32> flags((size_t)nq_pad);
        HIPCHK(hipMemcpy(flags.data(), h->ovf.p, (size_t)nq_pad * 4, hipMemcpyDeviceToHost));
        for (u32 f : flags) h->stats.overflowed_queries += f ? 1 : 0;
        HIPCHK(hipMemsetAsync(h->ovf.p, 0, (size_t)nq_pad * 4, h->stream));
        HIPCHK(hipMemsetAsync(h->ovf_any.p, 0, 4, h->stream));
        for (long long lo = 0; lo < docs_pad; lo += safe) {
            RCCHK(maxsim_chunk(h, (int)cap, k, nq, nq_pad, lo, std::min(safe, docs_pad - lo), false, true, pre));
            h->stats.rescan_launches += 1;
        }
        HIPCHK(hipMemcpyAsync(&any, h->ovf_any.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (any) return fail(NWKNN_EHIP, "internal: MaxSim list overflow on a safe range");
    }
    if (pre) {                                                     // the pre-screen's invariant, counted by every rescoring launch (as in search_attempt)
        u64 chk[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(chk, h->chk.p, sizeof chk, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->stats.prescreen_checked = (int64_t)chk[0];
        float w;
        const u32 wb = (u32)chk[2];
        memcpy(&w, &wb, 4);
        h->stats.prescreen_worst = (double)w;
        if (chk[1] != 0) { *violations = (long long)chk[1]; return NWKNN_OK; }
    }
    float* od = out_score;
    long long* oi = (long long*)out_doc;
    if (out_location == NWKNN_HOST) {
        RCCHK(h->out_d.ensure((size_t)nq * k * 4)); RCCHK(h->out_i.ensure((size_t)nq * k * 8));
        od = h->out_d.as<float>(); oi = h->out_i.as<long long>();
    }
    HIPCHK(nwk_launch_finalize(h->stream, h->buf.as<u64>(), h->cnt.as<u32>(), (int)cap, k, nq, doc_offset, od, oi, 1));
    if (h->opt_profile) HIPCHK(hipEventRecord(h->ev_end, h->stream));
    if (out_location == NWKNN_HOST) {
        HIPCHK(hipMemcpyAsync(out_score, od, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(out_doc, oi, (size_t)nq * k * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->opt_profile) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
        h->stats.total_ms = ms;
        ev_collect(h, &h->stats.tile_ms, &h->stats.compact_ms, &h->stats.prep_ms, &h->stats.rescore_ms);
        h->stats.prep_ms += h->base_prep_ms; h->base_prep_ms = 0.0;
    }
    return NWKNN_OK;
}

extern "C" int nwknn_maxsim_search(nwknn_handle h, const void* query_tokens, int64_t n_queries, int32_t tokens_per_query, int dtype, int location,
                                   int32_t k, int64_t doc_offset, float* out_score, int64_t* out_doc, int out_location)
{
    return guarded([&]() -> int {
        long long discarded = 0;
        for (int attempt = 0;; ++attempt) {
            long long bad = 0;
            RCCHK(maxsim_search_attempt(h, query_tokens, n_queries, tokens_per_query, dtype, location, k, doc_offset, out_score, out_doc, out_location, attempt, &bad));
            if (bad == 0) {
                h->stats.prescreen_fallbacks = attempt;
                h->stats.prescreen_violations = discarded;
                return NWKNN_OK;
            }
            discarded += bad;
            fprintf(stderr, "libnwknn: MaxSim pre-screen invariant failed on %lld of %lld rescored pairs (|estimate - exact| above %lld ppm of the bound): "
                            "attempt discarded, searching again on the direct fp32 kernel\n", bad, (long long)h->stats.prescreen_checked, (long long)h->opt_pre_check_ppm);
            if (attempt >= 1) return fail(NWKNN_EHIP, "MaxSim pre-screen invariant violated on the direct path (%lld pairs): internal error", bad);
        }
    });
}

// ---- merge ------------------------------------------------------------------------------------------
extern "C" int nwknn_merge(nwknn_handle h, const float* dists, const int64_t* idx, int32_t n_lists, int64_t nq, int32_t k_in, int32_t k_out,
                           float* out_dist, int64_t* out_idx, int location)
{
    if (!h || !dists || !idx || !out_dist || !out_idx) return fail(NWKNN_EINVAL, "nwknn_merge: null argument");
    if (n_lists <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0) return fail(NWKNN_EINVAL, "nwknn_merge: sizes must be positive");
    if ((long long)n_lists * k_in > 4096) return fail(NWKNN_EINVAL, "nwknn_merge: n_lists*k_in=%lld exceeds 4096; fold in groups", (long long)n_lists * k_in);
    if (location != NWKNN_HOST && location != NWKNN_DEVICE) return fail(NWKNN_EINVAL, "bad location %d", location);
    RCCHK(use_device(h));
    const size_t nin = (size_t)n_lists * nq * k_in, nout = (size_t)nq * k_out;
    const float* dd = dists; const long long* di = (const long long*)idx;
    float* od = out_dist; long long* oi = (long long*)out_idx;
    if (location == NWKNN_HOST) {
        RCCHK(h->mrg_d.ensure(nin * 4)); RCCHK(h->mrg_i.ensure(nin * 8));
        RCCHK(h->out_d.ensure(nout * 4)); RCCHK(h->out_i.ensure(nout * 8));
        HIPCHK(hipMemcpyAsync(h->mrg_d.p, dists, nin * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->mrg_i.p, idx, nin * 8, hipMemcpyHostToDevice, h->stream));
        dd = h->mrg_d.as<float>(); di = h->mrg_i.as<long long>(); od = h->out_d.as<float>(); oi = h->out_i.as<long long>();
    }
    HIPCHK(nwk_launch_merge(h->stream, dd, di, n_lists, (long long)nq * k_in, nq, k_in, k_out, od, oi));
    if (location == NWKNN_HOST) {
        HIPCHK(hipMemcpyAsync(out_dist, od, nout * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(out_idx, oi, nout * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return NWKNN_OK;
}

// ---- explicit pairs: the canonical distance of (query q, base row idx[q][j]) -- the self-check of validate_files_v0 ----------------------
static int pair_distances_impl(nwknn_ctx* h, const void* queries, int64_t nq, int dtype, int location, const int64_t* idx, int32_t k, int metric,
                               int64_t index_offset, float* out_dist, double* out_dot64, int out_location)
{
    if (!h || !queries || !idx || (!out_dist && !out_dot64)) return fail(NWKNN_EINVAL, "nwknn_pair_distances: null argument");
    if (!h->bp.p || h->nb <= 0) return fail(NWKNN_ESTATE, "nwknn_pair_distances: call nwknn_set_base first");
    if (h->ms_docs > 0) return fail(NWKNN_ESTATE, "nwknn_pair_distances: the handle holds MaxSim documents");
    if (nq <= 0 || k <= 0 || k > 4096) return fail(NWKNN_EINVAL, "nwknn_pair_distances: nq=%lld, k=%d must satisfy nq > 0, 0 < k <= 4096", (long long)nq, k);
    if (metric < 0 || metric > 2) return fail(NWKNN_EINVAL, "nwknn_pair_distances: unknown metric %d", metric);
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16) return fail(NWKNN_EINVAL, "nwknn_pair_distances: unsupported query dtype %d", dtype);
    if ((location != NWKNN_HOST && location != NWKNN_DEVICE) || (out_location != NWKNN_HOST && out_location != NWKNN_DEVICE))
        return fail(NWKNN_EINVAL, "nwknn_pair_distances: bad location");
    RCCHK(use_device(h));
    long long cap = 256;
    while (cap < k) cap <<= 1;
    const long long nq_pad = round_up(nq, NWK_QPAD_ROWS);
    RCCHK(h->qp.ensure((size_t)nq_pad * h->dp * (h->f16 ? 2 : 4)));
    RCCHK(h->qn.ensure((size_t)nq_pad * 4));
    RCCHK(h->cnt.ensure((size_t)nq_pad * 4));
    RCCHK(h->cnt0.ensure((size_t)nq_pad * 4));
    RCCHK(h->buf.ensure((size_t)nq_pad * cap * 8));
    h->ev_used = 0;
    RCCHK(stage_and_prep(h, queries, location, dtype == NWKNN_DTYPE_F16, h->f16, nq, h->d, nq_pad, h->dp, h->qraw, h->qp.p, h->qn.as<float>()));
    const long long* didx = (const long long*)idx;
    const size_t nk = (size_t)nq * k;
    if (location == NWKNN_HOST) {                                   // the index table lives where the queries live
        RCCHK(h->mrg_i.ensure(nk * 8));
        HIPCHK(hipMemcpyAsync(h->mrg_i.p, idx, nk * 8, hipMemcpyHostToDevice, h->stream));
        didx = h->mrg_i.as<long long>();
    }
    if (out_dot64) {                                                // nwknn_pair_dots: float64 sums over the prepared rows
        double* od64 = out_dot64;
        if (out_location == NWKNN_HOST) { RCCHK(h->out_i.ensure(nk * 8)); od64 = h->out_i.as<double>(); }
        HIPCHK(nwk_launch_pair_dots(h->stream, h->f16 ? 1 : 0, h->qp.p, h->bp.p, didx, nq, k, h->dp, index_offset, h->nb, od64));
        if (out_location == NWKNN_HOST) HIPCHK(hipMemcpyAsync(out_dot64, od64, nk * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return NWKNN_OK;
    }
    HIPCHK(nwk_launch_pairs_to_keys(h->stream, didx, nq, nq_pad, k, (int)cap, index_offset, h->nb, h->buf.as<u64>(), h->cnt.as<u32>(), h->cnt0.as<u32>()));
    NwkRescoreParams rp;
    memset(&rp, 0, sizeof rp);
    rp.qp = h->qp.as<float>(); rp.bp = h->bp.as<float>(); rp.qn = h->qn.as<float>(); rp.bn = h->bn.as<float>();
    rp.buf = h->buf.as<u64>(); rp.cnt = h->cnt.as<u32>(); rp.cnt0 = h->cnt0.as<u32>();
    rp.dp = h->dp; rp.cap = (int)cap; rp.k = k; rp.prune = 0; rp.rows16 = h->f16 ? 1 : 0; rp.coop = h->opt_rescore_coop ? 1 : 0;
    rp.chk_scale = 1.0f; rp.chk = nullptr;                          // nothing to hold these keys against: they carry no estimate
    HIPCHK(nwk_launch_rescore(h->stream, rp, metric, nq_pad));      // the canonical chain over the prepared rows, one lane per pair
    float* od = out_dist;
    if (out_location == NWKNN_HOST) { RCCHK(h->out_d.ensure(nk * 4)); od = h->out_d.as<float>(); }
    HIPCHK(nwk_launch_keys_to_distances(h->stream, h->buf.as<u64>(), nq, k, (int)cap, od));
    if (out_location == NWKNN_HOST) HIPCHK(hipMemcpyAsync(out_dist, od, nk * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return NWKNN_OK;
}

extern "C" int nwknn_pair_distances(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, const int64_t* idx, int32_t k, int metric,
                                    int64_t index_offset, float* out_dist, int out_location)
{
    if (!out_dist) return fail(NWKNN_EINVAL, "nwknn_pair_distances: null argument");
    return guarded([&] { return pair_distances_impl(h, queries, nq, dtype, location, idx, k, metric, index_offset, out_dist, nullptr, out_location); });
}

extern "C" int nwknn_pair_dots(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, const int64_t* idx, int32_t k, int64_t index_offset,
                               double* out_dot, int out_location)
{
    if (!out_dot) return fail(NWKNN_EINVAL, "nwknn_pair_dots: null argument");
    return guarded([&] { return pair_distances_impl(h, queries, nq, dtype, location, idx, k, NWKNN_METRIC_ONE_MINUS_DOT, index_offset, nullptr, out_dot, out_location); });
}

// ---- device calibration ---------------------------------------------------------------------------------------------------------------
extern "C" int nwknn_fabric_probe(nwknn_handle h, int64_t bytes, double seconds, double* out_gbs)
{
    if (!h || !out_gbs) return fail(NWKNN_EINVAL, "nwknn_fabric_probe: null argument");
    if (bytes < (1ll << 20) || bytes > (16ll << 30) || bytes % (1ll << 20) != 0) return fail(NWKNN_EINVAL, "nwknn_fabric_probe: bytes must be a multiple of 1 MiB in [1 MiB, 16 GiB]");
    if (!(seconds > 0.0) || seconds > 10.0) return fail(NWKNN_EINVAL, "nwknn_fabric_probe: seconds must be in (0, 10]");
    RCCHK(use_device(h));
    void* buf = nullptr;
    hipError_t me = hipMalloc(&buf, (size_t)bytes + 4096 * 4);
    if (me != hipSuccess) return fail(NWKNN_ENOMEM, "nwknn_fabric_probe: hipMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(me));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = NWKNN_OK;
    auto done = [&](int code) { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); (void)hipFree(buf); return code; };
    if (hipMemsetAsync(buf, 0, (size_t)bytes + 4096 * 4, h->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return done(fail(NWKNN_EHIP, "nwknn_fabric_probe: setup failed: %s", hipGetErrorString(hipGetLastError())));
    const unsigned grid = (unsigned)h->num_cu * 8u;
    float* sink = (float*)((char*)buf + bytes);
    // passes per launch: ~4 ms of work at 4 TB/s, so that a launch is long against its own start-up; launches until `seconds` have passed
    const int passes = (int)std::max<long long>(1, (16ll << 30) / bytes / 1);
    // warm: one launch (brings the working set to where it will live), then launches timed one by one until `seconds` have passed.  The
    // rate reported is the FASTEST launch's: the question is what the fabric can deliver, and a launch that shared the device with
    // something else for a moment (seen once in round 5: one 3 ms launch took 33) must not answer it.
    if (nwk_launch_stream_read(h->stream, buf, bytes, 1, grid, sink) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return done(fail(NWKNN_EHIP, "nwknn_fabric_probe: launch failed: %s", hipGetErrorString(hipGetLastError())));
    const auto t_begin = std::chrono::steady_clock::now();
    double best = 0.0;
    do {
        float ms = 0.f;
        if (hipEventRecord(e0, h->stream) != hipSuccess || nwk_launch_stream_read(h->stream, buf, bytes, passes, grid, sink) != hipSuccess ||
            hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
            rc = fail(NWKNN_EHIP, "nwknn_fabric_probe: launch failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        if (ms > 0.f) best = std::max(best, (double)bytes * passes / 1e9 / (ms * 1e-3));
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < seconds);
    if (rc != NWKNN_OK) return done(rc);
    *out_gbs = best;
    return done(NWKNN_OK);
}

extern "C" int nwknn_mfma_ceiling(nwknn_handle h, int dtype, double seconds, double* out_tflops, double* out_clock_ghz)
{
    if (!h || !out_tflops) return fail(NWKNN_EINVAL, "nwknn_mfma_ceiling: null argument");
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16) return fail(NWKNN_EINVAL, "nwknn_mfma_ceiling: dtype must be f32 or f16");
    if (!(seconds > 0.0) || seconds > 30.0) return fail(NWKNN_EINVAL, "nwknn_mfma_ceiling: seconds must be in (0, 30]");
    RCCHK(use_device(h));
    const int f16 = dtype == NWKNN_DTYPE_F16;
    const size_t rnd_bytes = 4u << 20, out_cap = (size_t)h->num_cu * 16 * 24;
    RCCHK(h->calib.ensure(rnd_bytes + out_cap));
    char* rnd = h->calib.as<char>();
    char* out = rnd + rnd_bytes;
    if (f16) HIPCHK(nwk_launch_gen_f16(h->stream, 0x6E77CA11ull, 0, (long long)(rnd_bytes / 2 / 64), 64, rnd));
    else HIPCHK(nwk_launch_gen(h->stream, 0x6E77CA11ull, 0, (long long)(rnd_bytes / 4 / 64), 64, (float*)rnd));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(NWKNN_EHIP, "hipEventCreate failed"); }
    int rc = NWKNN_OK;
    double best = 0.0;
    unsigned best_grid = 0;
    // The loop is run at every number of MFMA-only waves per SIMD the kernels use or could use (identical waves that do nothing but issue
    // MFMAs contend for the pipe differently than a kernel's waves, which take turns); the ceiling is the best of them.  Each variant:
    // launches back to back until its share of `seconds` has passed (the clock settles under the load), the second half timed.
    const int n_var = f16 ? 2 : 3;
    const bool verbose = getenv("NWKNN_CALIB_VERBOSE") != nullptr;
    for (int wps = 1; wps <= n_var && rc == NWKNN_OK; ++wps) {
        const int waves = f16 ? (wps == 2 ? 8 : 4) : 4;                      // waves per workgroup
        const int per_cu = f16 ? 1 : wps;                                    // workgroups resident per CU
        const unsigned grid = (unsigned)h->num_cu * (unsigned)per_cu * 4u;   // four rounds of resident workgroups per launch
        const int iters = f16 ? (wps == 2 ? 2000 : 4000) : 3000 / wps;       // ~20 ms per launch
        const double flops_per_launch = (double)grid * waves * iters * 64.0 * (f16 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 2);
        const double share = seconds / n_var;
        const auto t_begin = std::chrono::steady_clock::now();
        auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        int timed = 0;
        bool timing = false;
        while (rc == NWKNN_OK) {
            const double t = elapsed();
            if (!timing && t >= 0.5 * share) { if (hipEventRecord(e0, h->stream) != hipSuccess) rc = fail(NWKNN_EHIP, "hipEventRecord failed"); timing = true; }
            if (timing && t >= share && timed > 0) break;
            if (nwk_launch_mfma_bare(h->stream, f16, wps, rnd, iters, grid, out) != hipSuccess) { rc = fail(NWKNN_EHIP, "bare MFMA loop launch failed: %s", hipGetErrorString(hipGetLastError())); break; }
            if (timing) ++timed;
            if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(NWKNN_EHIP, "bare MFMA loop failed: %s", hipGetErrorString(hipGetLastError())); break; }
        }
        if (rc != NWKNN_OK) break;
        float ms = 0.f;
        if (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) {
            rc = fail(NWKNN_EHIP, "timing the bare MFMA loop failed");
            break;
        }
        const double tf = flops_per_launch * timed / (ms * 1e-3) / 1e12;
        if (verbose) fprintf(stderr, "libnwknn: bare %s MFMA loop, %d wave(s) per SIMD: %.1f TFLOP/s\n", f16 ? "f16" : "f32", wps, tf);
        if (tf > best) {
            best = tf; best_grid = grid;
            if (out_clock_ghz) {                                             // median over the workgroups of this variant's last launch: cycles / (100 MHz ticks)
                std::vector<unsigned long long> raw((size_t)grid * 3);
                if (hipMemcpy(raw.data(), out, (size_t)grid * 24, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(NWKNN_EHIP, "reading the loop's clock stamps failed"); break; }
                std::vector<double> clk;
                for (unsigned g = 0; g < grid; ++g) if (raw[(size_t)g * 3 + 1] > 0) clk.push_back((double)raw[(size_t)g * 3] / (double)raw[(size_t)g * 3 + 1] * 0.1);
                std::sort(clk.begin(), clk.end());
                *out_clock_ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
            }
        }
    }
    (void)best_grid;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    RCCHK(rc);
    *out_tflops = best;
    return NWKNN_OK;
}

extern "C" int nwknn_generate(nwknn_handle h, uint64_t seed, int64_t row0, int64_t rows, int32_t d, int dtype, void* out_device)
{
    if (!h || !out_device) return fail(NWKNN_EINVAL, "nwknn_generate: null argument");
    if (rows <= 0 || d <= 0 || row0 < 0) return fail(NWKNN_EINVAL, "nwknn_generate: bad shape");
    if (dtype != NWKNN_DTYPE_F32 && dtype != NWKNN_DTYPE_F16) return fail(NWKNN_EINVAL, "nwknn_generate: unsupported dtype %d", dtype);
    RCCHK(use_device(h));
    if (dtype == NWKNN_DTYPE_F16) HIPCHK(nwk_launch_gen_f16(h->stream, seed, row0, rows, d, out_device));
    else HIPCHK(nwk_launch_gen(h->stream, seed, row0, rows, d, (float*)out_device));
    return NWKNN_OK;
}
