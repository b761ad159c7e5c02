// nwknn_internal.h -- shared between the kernel files nwk_*.hip (device code + launchers) and nwknn_api.cpp (C ABI, orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NWKNN_METRIC_SQEUCLIDEAN 0
#define NWKNN_METRIC_ONE_MINUS_DOT 1
#define NWKNN_METRIC_COSINE 2

#define NWK_TILE_ROWS 128          // base rows per workgroup tile; granule of chunk sizes
#define NWK_QPAD_ROWS 256          // query-side arrays are padded to this (the larger query tile)
#define NWK_K_ALIGN 32             // prepared fp32 rows are zero padded to a multiple of this many floats
#define NWK_K_ALIGN_F16 64         // prepared fp16 rows: multiple of this many halves
#define NWK_MAX_WORKGROUPS_PER_LAUNCH (1ll << 22)   // x 512 threads = 2^31: half of what HIP accepts, super-tile padding has room
#define NWK_F16_TILE_ROWS 256      // fp16 path: query and base tiles are 256 rows
#define NWK_MAX_CAPACITY 8192      // candidate slots per query: the compaction kernel sorts a list in LDS (16 B per slot: 128 KiB of the CU's 160)
#define NWK_MAX_K 2048             // a list holds at least 4 k slots

struct NwkTileParams {
    const float* qp;               // prepared queries [nq_pad][dp]
    const float* bp;               // prepared base    [nb_pad][dp]
    const float* qn;               // squared norms of the queries [nq_pad]
    const float* bn;               // squared norms of the base    [nb_pad]
    const unsigned long long* tau; // current k-th key per query (~0 = "fewer than k seen")
    unsigned int* cnt;             // entries in each query's candidate list
    unsigned long long* buf;       // candidate lists [nq_pad][cap]
    const int* qtile_list;         // optional indirection (rescans); nullptr = tiles 0..n_qtiles-1
    long long nb_valid;            // base rows >= nb_valid are zero padding and never listed
    long long nq_valid;            // query rows >= nq_valid are zero padding: their pre-filter term rejects everything
    long long dense_col0;          // dense mode: list slot = column - dense_col0
    int n_qtiles;
    int bt0;                       // first base tile of this launch
    int n_btiles;
    int n_super_b;                 // filled in by the launcher
    int sq_shift;                  // super-tile = (1 << sq_shift) query tiles x (64 >> sq_shift) base tiles
    int dp;
    int cap;
    unsigned long long* dbg;       // -DNWKNN_DEV builds only (read through NWK_DBG): per-workgroup stamps {t_start, t_loop, t_loop_end, t_end, rt_start, rt_end, xcc, 0}
    // fp16 pre-screen of the exact fp32 search (nwk_launch_tile_prescreen): qp/bp are the SCALED fp16 copies, qn/bn the canonical
    // fp32 norms; a pair passes when  acc >= pre_scale * (row(q) + col(b))  with the error bound of DESIGN.md folded into row/col
    float pre_scale;               // s_q * s_b (power of two): the accumulators hold pre_scale * dot
    float pre_eps_half;            // eps/2: relative error of the fp16 dot product w.r.t. (|q|^2 + |b|^2)/2
    float pre_aq, pre_ab;          // absolute terms: pre_aq * sqrt(|b|^2) and pre_ab * sqrt(|q|^2)
    float pre_cst;
    float pre_inv_scale;           // 1 / pre_scale
    int f16_shape;                 // fp16 tile kernel build (all on v_mfma_f32_16x16x32_f16): 1 = burst form, 3 = streaming form, 4 = ping-pong form, 5 = two-issuing-waves form
    const float* ms_rq;            // MaxSim pre-screen: per-query and per-doc parts of the score bound
    const float* ms_cd;
    int walk_q_fastest;            // 1: inside a super-tile the QUERY tile index runs fastest (consecutive blocks share a base tile), 0: the base tile index
    const float* ctab;             // ping-pong pre-screen kernel: the column term of every base row for THIS search (nwk_launch_col_terms), +inf past nb_valid
    int agg;                       // streaming fp16 kernel: the build whose epilogue reserves list slots per query ROW (loose-threshold chunks; run_chunk)
    int tiles_per_wg;              // fp16 kernels: consecutive base tiles a workgroup walks (strip); the launcher turns n_btiles into strips
    int n_btiles_real;             // filled in by the launcher: base tiles of this launch
    int diag;                      // -DNWKNN_DEV builds only: timing ablations of nwk_tile_kernel_f16x (1 no fragment reads, 2 no LDS-DMA in the loop, 4 no MFMAs, 8 no epilogue); 0 in the product
};

struct NwkRescoreParams {
    const float* qp;               // prepared fp32 queries / base (canonical layout) and norms
    const float* bp;
    const float* qn;
    const float* bn;
    unsigned long long* buf;
    unsigned int* cnt;             // entries now; the rescoring kernel shortens a pruned list to cnt0 + survivors
    const unsigned int* cnt0;      // entries after the last compaction: [cnt0, min(cnt, cap)) are pre-screen candidates
    int dp;
    int cap;
    int k;
    int rows16;                    // 1: qp / bp hold fp16 rows in natural k order (exact fp16 mode); 0: prepared fp32 rows
    int prune;                     // 1: derive bounds from the fp16 estimates and skip pairs that cannot reach the top k
    int coop;                      // 1: the wave fetches the 64 rows of a round together through LDS (nwk_select.hip); 0: a lane fetches its own row
    int defer;                     // 1: prune only -- survivors keep their estimates, the list is cut to them, tau becomes the k-th smallest upper bound (needs prune = 1; nwk_select.hip)
    unsigned long long* tau;       // thresholds (written when defer = 1)
    unsigned int* cnt0_w;          // == cnt0, writable: a deferred pass also drops the exact entries above the bound (defer = 1)
    unsigned int* skip;            // [nq_pad], written when defer = 1: 1 = this query's list is in the deferred state (estimates behind the prefix: the compaction
                                   // that follows must leave it alone), 0 = the query was rescored in this launch after all (see defer_limit) and is to be sorted
    int defer_limit;               // a query whose pruned list would hold more entries than this is rescored now: candidates the bound cannot tell apart must not pile up until the list overflows
    float eps_half, aq, ab, cst;   // the constants of the pre-screen error bound (as in NwkTileParams)
    float chk_scale;               // the invariant |estimate - exact| <= bound is checked against chk_scale x the bound (1: the bound itself; option "pre_check_ppm")
    unsigned long long* chk;       // [0] += pairs checked, [1] += pairs outside the bound, [2] (low word) max= float bits of the worst |estimate - exact| / bound
};

struct NwkMsRescoreParams {
    const float* qp;               // prepared fp32 query tokens [nq_pad*32][dp] / doc tokens [docs_pad*64][dp]
    const float* bp;
    unsigned long long* buf;
    const unsigned int* cnt;
    const unsigned int* cnt0;
    const float* rq;               // score bound: R(query) + C(doc)
    const float* cd;
    int dp, cap, k, prune;
    unsigned int* surv;            // [nq_pad][cap]: list slots of the pairs to rescore, written by the prune kernel
    unsigned int* nsurv;           // [nq_pad]
    unsigned long long* dbg;       // dev: [0] += listed pairs, [1] += rescored pairs
    int q_in_lds;                  // set by the launcher: the query's token rows fit in LDS next to the list arrays
    int rescore_valu;              // 1: the vector-pipe rescoring kernel even where the MFMA one applies (option "ms_rescore_valu")
    float chk_scale;               // the invariant |estimate - exact score| <= R(query) + C(doc), checked as for kNN (NwkRescoreParams::chk)
    unsigned long long* chk;
};

struct NwkCompactParams {
    unsigned long long* buf;
    unsigned int* cnt;
    unsigned long long* tau;
    unsigned int* cnt0;            // optional: receives the post-compaction count (start of the next pre-screen candidates)
    unsigned int* ovf;             // per query: set to 1 when its list overflowed in this launch
    unsigned int* ovf_any;         // one word: OR of all overflows
    const int* qtile_list;         // optional: compact only the queries of these tiles (grid = n_tiles*128)
    int cap;
    int k;
    int qtile_rows;                // rows per entry of qtile_list (128 or 256)
    int dense_count;               // > 0: every list holds exactly this many entries (first chunk)
    int dedupe;
    const unsigned int* skip;      // optional (deferred rescoring): queries flagged 1 are left as they are
};

hipError_t nwk_launch_gen(hipStream_t st, uint64_t seed, long long row0, long long rows, int d, float* out);
hipError_t nwk_launch_gen_f16(hipStream_t st, uint64_t seed, long long row0, long long rows, int d, void* out);
hipError_t nwk_launch_prep_f16(hipStream_t st, const void* raw, int raw_is_f16, long long rows, int d, long long rows_pad, int dp, void* prepared, float* norms);
hipError_t nwk_launch_tile_f16(hipStream_t st, NwkTileParams p, int metric, bool dense, int pre);   // every fp16 tile launch: raw sums (pre = 0) or pre-screen (pre = 1)
hipError_t nwk_launch_tile_f16p(hipStream_t st, const NwkTileParams& p, int metric, int pre, unsigned grid);   // streaming form (f16_mfma = 3)
hipError_t nwk_launch_tile_f16x_pre(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid);          // ping-pong form (f16_mfma = 4), pre-screen
hipError_t nwk_launch_col_terms(hipStream_t st, const NwkTileParams& p, int metric, long long rows_pad, float* out);  // out[b] = the pre-screen's column term of base row b
hipError_t nwk_launch_tile_f16x_raw(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid);          // ... raw fp16 sums
hipError_t nwk_launch_tile_f16y_pre(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid);          // two-issuing-waves form (f16_mfma = 5), pre-screen
hipError_t nwk_launch_tile_f16y_raw(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid);          // ... raw fp16 sums
hipError_t nwk_launch_prep_strided(hipStream_t st, const void* raw, int in_dtype, long long rs, long long ks, long long rows, int d,
                                   long long rows_pad, int dp, int out_f16, void* prepared, float* norms, unsigned int* maxabs_bits = nullptr);
// maxabs_bits (fp32 layouts only, may be null): atomicMax of the float bits of the largest |x| laid out -- the pre-screen's scale without a pass of its own
hipError_t nwk_launch_prep(hipStream_t st, const float* raw, long long rows, int d, long long rows_pad, int dp, float* prepared, float* norms,
                           unsigned int* maxabs_bits = nullptr);
hipError_t nwk_launch_tile(hipStream_t st, NwkTileParams p, int metric, bool dense, int tile_rows, int diag, int occ);   // occ: workgroups per CU of the 256-row shape (2 or 3)
hipError_t nwk_launch_half_copy16(hipStream_t st, const void* rows16, long long rows_pad, int dp, float scale, const float* norms, void* out16);
hipError_t nwk_launch_maxabs_f16(hipStream_t st, const void* x, long long n, unsigned int* out_bits);
hipError_t nwk_launch_rescore(hipStream_t st, const NwkRescoreParams& p, int metric, long long nq_pad);
hipError_t nwk_launch_maxabs(hipStream_t st, const float* x, long long n, unsigned int* out_bits);
hipError_t nwk_launch_half_copy(hipStream_t st, const float* prepared, long long rows_pad, int dp32, int dp16, float scale, const float* norms,
                                void* out16);
hipError_t nwk_launch_compact(hipStream_t st, const NwkCompactParams& p, long long n_queries);
hipError_t nwk_launch_finalize(hipStream_t st, const unsigned long long* buf, const unsigned int* cnt, int cap, int k, long long nq,
                               long long index_offset, float* out_d, long long* out_i, int negate);
hipError_t nwk_launch_maxsim(hipStream_t st, NwkTileParams p, bool dense);
#define NWK_MS_TOKENS_PER_DOC 64
#define NWK_MS_TOKENS_PER_QUERY 32
hipError_t nwk_launch_maxsim_pre(hipStream_t st, NwkTileParams p);
hipError_t nwk_launch_maxsim_bounds(hipStream_t st, const float* norms, long long n_items, int tokens, float eps_half, float a_self, float cst,
                                    int is_doc, float* out);
hipError_t nwk_launch_maxsim_rescore(hipStream_t st, NwkMsRescoreParams p, long long nq_pad, int slices);
hipError_t nwk_launch_merge(hipStream_t st, const float* dists, const long long* idx, int n_lists, long long list_stride, long long nq, int k_in,
                            int k_out, float* out_d, long long* out_i, int descending = 0);
hipError_t nwk_launch_pairs_to_keys(hipStream_t st, const long long* idx, long long nq, long long nq_pad, int k, int cap, long long index_offset,
                                    long long nb, unsigned long long* buf, unsigned int* cnt, unsigned int* cnt0);
hipError_t nwk_launch_pair_dots(hipStream_t st, int rows16, const void* qp, const void* bp, const long long* idx, long long nq, int k, int dp,
                                long long index_offset, long long nb, double* out);
hipError_t nwk_launch_keys_to_distances(hipStream_t st, const unsigned long long* buf, long long nq, int k, int cap, float* out);
hipError_t nwk_launch_stream_read(hipStream_t st, const void* src, long long bytes, int passes, unsigned grid, float* sink);   // nwk_calib.hip: streaming read of `bytes`, `passes` times
hipError_t nwk_launch_mfma_bare(hipStream_t st, int f16, int waves_per_simd, const void* rnd, int iters, unsigned grid, void* out);   // nwk_calib.hip: one launch of the bare MFMA loop; out: grid x {u64 cycles, u64 realtime (100 MHz), float}
int nwknn_internal_fail(int code, const char* msg);        // sets the calling thread's nwknn_last_error() message, returns code
