/*
 * nwknn.h -- C ABI of libnwknn.so: the MI355X (gfx950) brute-force kNN ground-truth engine.
 *
 * The reference (phact/neighborhoodwatch) has NO FFI for this path: its hot path is three Python call
 * sites into CUDA wheels.  Each entry point below names the reference interface it stands in for
 * (paths relative to /root/reference); INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (NWKNN_OK) or a negative NWKNN_E* code; the message of the last failure on
 *     the calling thread is available from nwknn_last_error().  No C++ exception crosses this boundary.
 *     (The reference fails by `assert`: cu_knn.py:136,177,237,281.)
 *   - the caller owns every buffer passed in or out; the library never frees or retains caller memory
 *     after a call returns.  Device memory behind a handle belongs to the library until nwknn_destroy().
 *   - a `location` argument says where a caller buffer lives: NWKNN_HOST or NWKNN_DEVICE (a HIP device
 *     pointer on the handle's device, e.g. torch.Tensor.data_ptr()).
 *   - matrices are dense row-major, C-contiguous.  Distances are float32, indices int64, exactly the tuple
 *     the reference's operators return (cu_knn.py:243-245).
 *   - one search at a time per handle; distinct handles are independent.
 */
#ifndef NWKNN_H
#define NWKNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NWKNN_ABI_VERSION 2        /* 2: nwknn_stats grew the pre-screen invariant fields */

/* error codes */
#define NWKNN_OK 0
#define NWKNN_EINVAL (-1)   /* bad argument: null pointer, k > rows, unknown metric/dtype ... */
#define NWKNN_EHIP (-2)     /* a HIP runtime call failed; see nwknn_last_error() */
#define NWKNN_ENOMEM (-3)   /* device or host allocation failed */
#define NWKNN_ESTATE (-4)   /* call order: e.g. search before set_base */
#define NWKNN_ENODEV (-5)   /* no usable gfx950 device */

/* metric (which reference engine's distance): */
#define NWKNN_METRIC_SQEUCLIDEAN 0   /* engine='raft'  : pylibraft knn default, cu_knn.py:243-245, cu_knn_ds.py:213 */
#define NWKNN_METRIC_ONE_MINUS_DOT 1 /* engine='torch' : 1 - q.b^T + topk,      cu_knn.py:256-257               */
#define NWKNN_METRIC_COSINE 2        /* engine='cuvs'  : brute_force metric="cosine", cu_knn.py:247-248         */

#define NWKNN_DTYPE_F32 0
#define NWKNN_DTYPE_F16 1          /* IEEE half: input element type and/or compute path (BASELINE config 4) */
#define NWKNN_DTYPE_F64 2          /* double: accepted for COLUMN input only (float64 parquet columns, cast once: cu_knn.py:208,243) */

#define NWKNN_HOST 0
#define NWKNN_DEVICE 1

typedef struct nwknn_ctx* nwknn_handle;

/* timing / accounting of the most recent nwknn_search on a handle (all times in milliseconds, measured
 * with HIP events on the handle's stream when option "profile" is 1, else 0) */
typedef struct nwknn_stats {
    int64_t nq, nb;                /* shape of the last search */
    int32_t d, k;
    int32_t tile_launches;         /* launches of the distance-tile kernel */
    int32_t compact_launches;      /* launches of the list compaction kernel */
    int32_t rescan_launches;       /* distance-tile launches spent repairing overflowed lists */
    int32_t overflowed_queries;    /* queries whose candidate list overflowed at least once */
    double tile_ms;                /* sum of distance-tile kernel durations */
    double compact_ms;             /* sum of compaction kernel durations */
    double prep_ms;                /* base + query layout/norm kernels inside the last search/set_base */
    double total_ms;               /* first launch to last launch of the search */
    double tile_flops;             /* 2 * q_rows * b_rows * d summed over the distance-tile launches */
    double rescore_ms;             /* option "prescreen": sum of the fp32 rescoring kernel durations */
    int32_t prescreen;             /* 1 when the RETURNED result came from the fp16 pre-screen + fp32 rescoring, 0 for the direct kernel */
    int32_t prescreen_fallbacks;   /* times the search was run again because the invariant below failed (fp32 data: on the direct kernel;
                                      exact fp16 mode: under a 16x wider bound); the discarded attempts leave no trace in the result */
    /* The pre-screen's invariant, checked on the device for EVERY pair that is rescored: the fp16 estimate a pair was listed under and its
     * canonical fp32 value differ by no more than the error bound the tile kernel pruned with (that bound allows 4 d u for the fp16 MFMA's
     * internal summation order, which the ISA leaves undefined -- this is where the allowance is held against the hardware). */
    int64_t prescreen_checked;     /* pairs compared in the attempt that produced the result */
    int64_t prescreen_violations;  /* pairs outside the bound, summed over the DISCARDED attempts (the returned result has none by construction) */
    double prescreen_worst;        /* max |estimate - exact| / bound over the checked pairs of the returned result (< 1; ~0.05-0.3 measured) */
    /* Walk calibration of the direct fp32 kernel (options super_q_shift / super_walk_q left at auto, at least six full-size chunks): two
     * consecutive full-size chunks are timed, one under each walk order of an XCD's super-tile (8 x 8 tiles base-tile-fastest; 2 x 32
     * query-tile-fastest, which moves ~38 % fewer bytes beyond L2), and the rest of the walk runs the faster one.  Which is faster depends
     * on the device's fabric / Infinity Cache, not on the data; the result is the same bits either way. */
    int32_t walk_calibrated;       /* 1 when the two timed chunks ran in this search */
    int32_t walk_query_fastest;    /* the walk the remaining chunks ran: 0 = 8 x 8 base fastest, 1 = 2 x 32 query fastest */
    double walk_ms_base_fastest, walk_ms_query_fastest;   /* the two timed chunks (equal row counts), milliseconds */
} nwknn_stats;

/* ---- library / device ------------------------------------------------------------------------------- */
int nwknn_version(void);
const char* nwknn_last_error(void);
int nwknn_device_count(void);
/* free/total bytes of HBM on a device.  Replaces the NVML query in tune_memory(), cu_knn.py:62-71 and
 * get_gpu_memory(), cu_knn_ds.py:30-36. */
int nwknn_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* Read rate, GB/s, of a plain streaming kernel over a working set of `bytes` read again and again for `seconds` (a multiple of 1 MiB, at most
 * 16 GiB; launches of ~16 GiB of reads each, the fastest launch's rate is reported): 128 MiB sits beyond the 8 x 4 MiB of L2 and inside the 256 MiB Infinity Cache -- the path the tile kernels' L2 misses take, and the
 * part of a device that differs between boxes when their MFMA pipes do not; 4 GiB measures HBM.  bench.py prints both next to the bare MFMA
 * loop (roofline.fabric_probe_gbs / hbm_probe_gbs) so that a slow line can be told apart from a slow box.  No reference counterpart. */
int nwknn_fabric_probe(nwknn_handle h, int64_t bytes, double seconds, double* out_gbs);

/* ---- handle ------------------------------------------------------------------------------------------ */
/* One handle = one GPU's share of the job (the reference hard-codes device 0, cu_knn.py:65). */
int nwknn_create(nwknn_handle* out, int device);
int nwknn_destroy(nwknn_handle h);
/* Run all work of this handle on an existing HIP stream (hipStream_t, e.g. torch's current stream); NULL
 * restores the handle's own stream. */
int nwknn_set_stream(nwknn_handle h, void* hip_stream);
/* Tunables: "capacity" (candidate slots per query, power of two >= 4k, at most 8192), "first_chunk", "max_chunk"
 * (base rows per launch, multiples of 128; MaxSim: a quarter of it in documents), "profile" (0/1: per-kernel HIP-event timing into nwknn_stats),
 * "compute_dtype" (NWKNN_DTYPE_F32 | NWKNN_DTYPE_F16: with F16 the next nwknn_set_base rounds fp32 input to fp16 and
 * distance tiles run on the fp16 MFMA path with fp32 accumulation), "tile_rows" (0 auto | 128 | 256, fp32 path),
 * "prescreen" (0/1, fp32 compute path, all three metrics: the next nwknn_set_base also keeps a scaled fp16 copy of
 * the base; searches then filter with the fp16 MFMA kernel under a conservative error bound and rescore every listed pair
 * with the canonical fp32 chain -- the same bits out as with 0, several times faster; nwknn_stats.prescreen reports whether a
 * search used it: non-finite or extreme-magnitude inputs fall back to the direct kernel),
 * "f16_exact" (fp16 compute path, default 1: results are the canonical fp32 arithmetic applied to the
 * fp16 values, bit for bit -- the fp16 MFMA kernel pre-screens, listed pairs are rescored; 0: raw fp16-MFMA sums),
 * "f16_mfma" (fp16 tile kernel build, all on v_mfma_f32_16x16x32_f16: -1 = auto, default: 4 from 768 halves per row on, else 3;
 * 4 = ping-pong form: the two waves of a SIMD alternate between loading a slab's fragments and running its MFMAs; 3 = streaming form;
 * 1 = burst form, one burst of LDS-DMA per slab -- the form every dense first chunk runs), "f16_strip" (base tiles a workgroup of the
 * fp16 kernels walks in a row: 0 = auto, default), "f16_agg" (which filtered chunks run the build of the streaming kernel that reserves
 * list slots per query row instead of per pair: -1 = auto, default: the loose-threshold chunks at the start of a walk unless "f16_mfma" names a form; 0 never; 1 all),
 * "tile_occ" (workgroups per CU of the fp32 tile kernel, 3 default | 2), "chunk_growth" (0 = auto), "super_q_shift" / "super_walk_q"
 * (shape and walk order of an XCD's super-tile), "upload_threads" (host threads of the column upload pipeline, default 8),
 * "upload_slab_mib" (slab size of that pipeline, default 128),
 * "pre_eps_pct" (>= 100: percent of the proven pre-screen error bound to apply; larger = more pairs rescored, same result),
 * "rescore_coop" (1 default | 0: how the rescoring kernel fetches the rows of the 64 pairs a wave rescores at a time -- together, 64-byte
 * segments through LDS, or every lane its own row; same bits),
 * "pre_defer" (1 default | 0: in a pre-screened walk only the LAST chunk rescores listed pairs with the canonical chain; the chunks before it
 * keep the fp16 estimates, drop entries whose lower bound exceeds the k-th smallest upper bound and filter the next chunk with that bound --
 * a third of the rescoring traffic; a query whose pruned list grows past a limit sized from the list capacity -- candidates the bound cannot
 * tell apart -- is rescored and sorted in that chunk after all, so nothing piles up until a list overflows; kNN searches only (MaxSim's
 * bound is too loose for it); 0 = every chunk rescores and sorts, as before round 5; same bits either way),
 * "pre_direct_rows" (-1 auto | rows: in a pre-screened search of fp32 data the filtered chunks that start below this many base rows run the
 * direct fp32 kernel -- right behind the dense chunk a quarter of all pairs pass the loose threshold, and rescoring them one by one costs more
 * than computing the exact keys on the MFMA; auto = only the chunk that follows the dense one; same bits whatever the value),
 * "pre_check_ppm" (1..1000000, default 1000000: the invariant |estimate - exact| <= bound of nwknn_stats is checked against this many
 * millionths of the bound ON THE FIRST ATTEMPT of a search; a violation discards the attempt and sends the search down its fallback --
 * fp32 data: the direct fp32 kernel; fp16 data (exact fp16 mode): the same search under a 16x wider bound, held against the whole of that
 * bound -- whose result is the same bits, so a small value only costs time; tests use it to drive the fallback.  A deviation that exceeds
 * even 16^3 x the proven bound -- no summation order of the products can produce one -- ends in NWKNN_EHIP, never in a wrong result),
 * MaxSim pre-screen tile kernel, d <= 128: "ms_strip" (document tiles a workgroup of the ring kernel walks: 0 = auto, 1 = the any-d
 * kernel), "ms_rescore_valu" (1: rescoring on the vector pipes).
 * No option changes a result: every switch that does (timing ablations, in-kernel stamps, a shrunken error bound) exists only in
 * developer builds of the library (-DNWKNN_DEV) and is refused with NWKNN_EINVAL here.  The environment variable NWKNN_DEFAULT_OPTS
 * ("name=value,...") applies options to every new handle through this function and says so on stderr. */
int nwknn_set_option(nwknn_handle h, const char* name, int64_t value);
int nwknn_get_stats(nwknn_handle h, nwknn_stats* out);
/* Name of the distance-tile kernel that carried the filtered launches of the most recent search on the handle, spelled as rocprofv3's
 * kernel trace prints it (e.g. "nwk_tile_kernel<0,false,256,16,0,3>"); "" before the first search.  The string lives in the handle. */
const char* nwknn_tile_kernel_name(nwknn_handle h);
/* Wait for everything queued on the handle's stream. */
int nwknn_synchronize(nwknn_handle h);

/* ---- dataset side: `dataset` argument of knn(dataset, queries, k), cu_knn.py:243; the arrow -> cuDF ->
 *      DLPack -> CuPy row-major conversion of cu_knn.py:205-211 ends here ------------------------------ */
/* Make rows [0, rows) x d of `base` the searchable set of this handle (this GPU's shard).  The library keeps
 * its own MFMA-ready copy plus row norms; the caller's buffer is not referenced after return (for
 * NWKNN_DEVICE the copy is queued on the handle's stream and the buffer may be reused after
 * nwknn_synchronize or any later blocking call).  dtype = element type of `base`; an fp16 base (or option
 * "compute_dtype" = NWKNN_DTYPE_F16) selects the fp16 compute path: results then carry fp16 input rounding; with option
 * "f16_exact" (default) they are the canonical fp32 arithmetic on the fp16 values, bit for bit; inputs holding Inf/NaN run the
 * raw fp16-MFMA sums and are checked against the float64 truth with a tolerance. */
int nwknn_set_base(nwknn_handle h, const void* base, int64_t rows, int32_t d, int dtype, int location);
/* Same, from COLUMN-major host data: columns[j] points at `rows` contiguous values of dimension j -- the layout parquet /
 * arrow / cuDF hold (one column per dimension, generate_dataset.py:219-262).  Replaces the device transpose
 * `cp.from_dlpack(df.to_dlpack()).copy(order='C')` of cu_knn.py:211: no host or device transpose pass is made, the layout
 * kernel reads the columns directly.  dtype: NWKNN_DTYPE_F32 | NWKNN_DTYPE_F64 | NWKNN_DTYPE_F16. */
int nwknn_set_base_columns(nwknn_handle h, const void* const* columns, int64_t rows, int32_t d, int dtype);

/* ---- the operator: distances, indices = knn(dataset, queries, k)  (cu_knn.py:243-245) ------------------- */
/* For each of the nq query rows: the k nearest rows of the base set, ascending by (distance, index),
 * lower index first among equal distances.  out_idx = row number in the base set + index_offset
 * (index_offset plays the role of `+ batch_offset`, cu_knn.py:273: pass the shard's first global row).
 * Requires 0 < k <= base rows (cu_knn.py:177-178,237) and k <= 2048 (a query's candidate list is sorted in LDS: 4 k slots of 16 bytes
 * in a CU's 160 KiB; the reference's operators take any k, its CLI defaults to 100 and `ck` to 1000). */
int nwknn_search(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, int32_t k, int metric,
                 int64_t index_offset, float* out_dist, int64_t* out_idx, int out_location);
/* Same with COLUMN-major host queries (cu_knn.py:229-235 converts the query tile the same way as the base tile). */
int nwknn_search_columns(nwknn_handle h, const void* const* columns, int64_t nq, int dtype, int32_t k, int metric,
                         int64_t index_offset, float* out_dist, int64_t* out_idx, int out_location);
/* The same two with CHUNKED columns, as pq.read_table hands them over (cu_knn.py:112-114: a ChunkedArray per dimension, one
 * chunk per parquet row group): chunk c of column j at chunks[c * d + j], chunk_rows[c] rows each, the same chunking for every
 * column, sum(chunk_rows) == rows.  Nothing is concatenated on the host: the upload pipeline gathers straight from the chunks. */
int nwknn_set_base_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks,
                                 int64_t rows, int32_t d, int dtype);
int nwknn_search_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t nq,
                               int dtype, int32_t k, int metric, int64_t index_offset, float* out_dist, int64_t* out_idx,
                               int out_location);

/* ---- a base tile handed over in pieces: the row groups of the tile loop's `base_table.slice(...)`, cu_knn.py:201-211 ----------------
 * The reference uploads a tile after it has read ALL of it (whole file first: cu_knn.py:112-114).  With 288 GB of HBM a tile is the file,
 * so nothing would reach the GPU before the last row group is decoded.  These three calls take a tile as it is decoded:
 *   nwknn_base_begin(h, rows, d, dtype)   announces a base of `rows` rows (dtype as in nwknn_set_base: it selects the compute path) and
 *                                         allocates it; the previous base is gone from this point on;
 *   nwknn_base_append*(...)               the next n rows, in row order -- row-major (host or device), or per-dimension column chunks
 *                                         exactly as nwknn_set_base_column_chunks takes them (chunk c of column j at chunks[c*d + j]).
 *                                         A column append returns once the caller's arrays have been gathered into the library's pinned
 *                                         slabs: the arrays may be released, and the piece's H2D copy and layout kernel run under whatever
 *                                         the caller does next (decoding the following row group);
 *   nwknn_base_end(h)                     all announced rows must have arrived; pads the last tile, runs the checks and copies of
 *                                         nwknn_set_base and makes the base current.
 * The prepared layout is made row by row, so the base -- and every search on it -- is bit for bit what nwknn_set_base* builds from the
 * concatenation of the pieces.  An error in an append abandons the half-built base (NWKNN_ESTATE from then on until the next begin). */
int nwknn_base_begin(nwknn_handle h, int64_t rows, int32_t d, int dtype);
int nwknn_base_append(nwknn_handle h, const void* rows_ptr, int64_t n, int dtype, int location);
int nwknn_base_append_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t n,
                                    int dtype);
int nwknn_base_end(nwknn_handle h);

/* ---- the tile loop's search, started before the tile has arrived: cu_knn.py:201-245 with the read, the upload and knn() overlapped ------
 * The reference's loop is read -> upload -> knn() per base tile, one after the other.  A search ATTACHED to a base that is being handed
 * over in pieces walks it as it lands: every chunk of the walk is launched as soon as the pieces that cover it have been appended (the
 * ordinary walk with chunks of at most 2^18 rows, so that little is left to do after the last piece; the lists are exact top-k lists under
 * the canonical arithmetic whatever the chunking: the result is bit for bit nwknn_search's), on a stream of its own beside the copies and
 * layout kernels of the pieces that follow.
 *   nwknn_base_begin(...)
 *   nwknn_search_attach*(h, queries..., k, metric, index_offset, &attached)     between begin and the first piece.  *attached = 0 (and
 *                                         NWKNN_OK) when the configured path needs the whole base before its first launch -- the fp16
 *                                         pre-screen (its scale is the base's max |x|), the fp16 compute path (its finiteness check): search
 *                                         after nwknn_base_end as before;
 *   nwknn_base_append*(...) ...           each also launches the chunks it completes;
 *   nwknn_base_end(h)                     launches the rest;
 *   nwknn_search_collect(h, out_dist, out_idx, out_location)                    overflow repair, order, copy-out: what nwknn_search returns.
 * While a search is attached, nwknn_search* on the handle returns NWKNN_ESTATE (the running lists are in use); a new base abandons it. */
int nwknn_search_attach(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, int32_t k, int metric,
                        int64_t index_offset, int* attached);
int nwknn_search_attach_column_chunks(nwknn_handle h, const void* const* chunks, const int64_t* chunk_rows, int32_t n_chunks, int64_t nq,
                                      int dtype, int32_t k, int metric, int64_t index_offset, int* attached);
int nwknn_search_collect(nwknn_handle h, float* out_dist, int64_t* out_idx, int out_location);

/* ---- merge of partial top-k lists: merge_indices_and_distances(), merge.py:56-101 ----------------------- */
/* dists/idx: [n_lists][nq][k_in] (partial list l of query q at ((l*nq)+q)*k_in); idx < 0 marks an empty
 * slot.  Writes the k_out smallest by (distance, index) into out_dist/out_idx [nq][k_out].
 * All four buffers live in `location`. */
int nwknn_merge(nwknn_handle h, const float* dists, const int64_t* idx, int32_t n_lists, int64_t nq, int32_t k_in,
                int32_t k_out, float* out_dist, int64_t* out_idx, int location);

/* ---- one process, N GPUs of one node: base rows sharded over the devices, ONE RCCL all-gather of the [Q,k] partial lists over
 *      xGMI, merge by (distance, index).  The reference is pinned to GPU 0 (cu_knn.py:65) and folds per-base-tile partial tables on
 *      the CPU (merge.py:56-101): this is that fold across GPUs, reachable from any host language (the one-process-per-GPU flavour
 *      of the same exchange is neighborhoodwatch_amd/distributed.py over torch.distributed).  RCCL is bound at run time (librccl.so.1);
 *      a call fails with NWKNN_EHIP when it cannot be loaded. ------------------------------------------------------------------- */
typedef struct nwknn_comm* nwknn_comm_handle;
/* accounting of the most recent nwknn_search_sharded (milliseconds from HIP events; maxima over the devices) */
typedef struct nwknn_comm_stats {
    int32_t n_devices, k;
    int64_t nq;
    double search_ms;                     /* slowest device: its shard search, first to last launch (needs option "profile" = 1) */
    double allgather_ms;                  /* the two ncclAllGather calls (distances, indices) */
    double merge_ms;                      /* merge of a device's slice of the query rows */
    double allgather_bytes_per_device;    /* bytes every device receives: N * Q * k * 12 */
    double tile_flops, tile_ms;           /* sums over the devices (option "profile"): tile_flops / tile_ms = per-GPU kernel rate */
} nwknn_comm_stats;
/* An engine handle + an RCCL communicator per listed device (ncclCommInitAll: all ranks live in this process).  devices: distinct HIP
 * device ordinals, all gfx950 (a device listed twice is refused: one shard per GPU, whatever transport NWKNN_RCCL_LIBRARY names).  n_dev = 1
 * is valid (the exchange is then a copy) and gives the result of nwknn_search. */
int nwknn_comm_create(nwknn_comm_handle* out, const int* devices, int n_dev);
int nwknn_comm_destroy(nwknn_comm_handle c);
int nwknn_comm_device_count(nwknn_comm_handle c);
/* The engine handle of device i (owned by the communicator): for per-device nwknn_get_stats; options are set on all devices at once with */
nwknn_handle nwknn_comm_engine(nwknn_comm_handle c, int i);
int nwknn_comm_set_option(nwknn_comm_handle c, const char* name, int64_t value);
/* Shard rows [0, rows) of a HOST row-major base contiguously over the devices: device g takes rows [rows*g/N, rows*(g+1)/N), uploaded and
 * laid out by one host thread per device.  dtype NWKNN_DTYPE_F32 | NWKNN_DTYPE_F16 as in nwknn_set_base. */
int nwknn_comm_set_base(nwknn_comm_handle c, const void* base_host, int64_t rows, int32_t d, int dtype);
/* The same with seeded synthetic rows generated on the devices (nwknn_generate; row r of the global base on whichever device holds it). */
int nwknn_comm_generate_base(nwknn_comm_handle c, uint64_t seed, int64_t rows, int32_t d, int dtype);
/* distances, indices = knn(whole base, queries, k) with HOST queries and HOST outputs [nq][k]: every device searches its shard with global
 * row ids (index_offset = first row of the shard: "+ batch_offset", cu_knn.py:273), one all-gather per array, device g merges rows
 * [nq*g/N, nq*(g+1)/N) of the result.  Requires k <= rows of the smallest shard (cu_knn.py:177-178,237). */
int nwknn_search_sharded(nwknn_comm_handle c, const void* queries_host, int64_t nq, int dtype, int32_t k, int metric, float* out_dist,
                         int64_t* out_idx);
int nwknn_comm_get_stats(nwknn_comm_handle c, nwknn_comm_stats* out);
/* The exchange's merge step for partial lists the CALLER gathered (another transport, the partial files of merge.py:56-101, a test):
 * dists / idx are HOST arrays [n_lists][nq][k] (idx < 0 marks an empty slot).  The query rows are cut into n_slices contiguous slices the
 * way nwknn_search_sharded cuts them over its devices; slice s is merged on device s mod N of the communicator straight from the lists'
 * own stride, and more than 4096 (list, rank) entries per query are folded in groups of lists, level after level until one merge takes
 * the rest -- the code path of a multi-device search, reachable with any number of lists on any number of devices.  Limit (checked before
 * anything is uploaded or searched, here and in the *_sharded calls): n_lists * k > 4096 needs k <= 2048, because a fold level merges at
 * least two lists of k entries in one 4096-entry sort.  Writes the k best by (distance, index) to HOST out_dist / out_idx [nq][k]. */
int nwknn_comm_merge_lists(nwknn_comm_handle c, const float* dists_host, const int64_t* idx_host, int32_t n_lists, int64_t nq, int32_t k,
                           int32_t n_slices, int descending, float* out_dist, int64_t* out_idx);
/* (descending != 0: the values are SCORES, higher first, ties to the lower id -- the order of nwknn_maxsim_search.) */

/* MaxSim over the devices (BASELINE config 5 names 8 GPUs; like nwknn_maxsim_* an extension with no reference counterpart): documents
 * sharded contiguously over the devices (HOST doc_tokens [n_docs][tokens_per_doc][d] float32), queries replicated, per-shard top-k with
 * global document ids, ONE all-gather per array, device g merges its slice of the queries by (score descending, doc id ascending).
 * With one device: the result of nwknn_maxsim_search. */
int nwknn_comm_maxsim_set_docs(nwknn_comm_handle c, const void* doc_tokens_host, int64_t n_docs, int32_t tokens_per_doc, int32_t d, int dtype);
int nwknn_maxsim_search_sharded(nwknn_comm_handle c, const void* query_tokens_host, int64_t n_queries, int32_t tokens_per_query, int dtype,
                                int32_t k, float* out_score, int64_t* out_doc);

/* ---- MaxSim late interaction (BASELINE.json config 5).  NOT in the reference: its `ck` program runs flat token-level
 *      kNN through process_batches (colbert_knn.py:84-126); this is the ColBERT scoring rule itself, provided as an
 *      extension with its own oracle (oracle nwo_maxsim) and no reference parity claim. ---------------------------- */
/* doc_tokens: [n_docs][tokens_per_doc][d] float32, tokens_per_doc must be 64 (pad short docs by repeating a token). */
int nwknn_maxsim_set_docs(nwknn_handle h, const void* doc_tokens, int64_t n_docs, int32_t tokens_per_doc, int32_t d,
                          int dtype, int location);
/* query_tokens: [n_queries][tokens_per_query][d] float32, tokens_per_query must be 32.
 * score(q, doc) = sum_i max_j <q_i, doc_j>; writes the k highest-scoring docs per query, descending by
 * (score, then lower doc id): out_score float32 [n_queries][k], out_doc int64 [n_queries][k] (+ doc_offset). */
int nwknn_maxsim_search(nwknn_handle h, const void* query_tokens, int64_t n_queries, int32_t tokens_per_query, int dtype,
                        int location, int32_t k, int64_t doc_offset, float* out_score, int64_t* out_doc, int out_location);

/* ---- self-check: the distances of GIVEN pairs (validate_files_v0, parquet_to_format.py:351-383, recomputes every listed neighbour's
 *      distance on the CPU, row by row) ---------------------------------------------------------------------------------------------
 * out_dist[q][j] = the canonical distance (same arithmetic, same bits as nwknn_search) between query row q and base row
 * idx[q][j] - index_offset of the handle's base set; NaN where that row does not exist (idx < index_offset, empty slots of a merge).
 * idx lives where the queries live (`location`); 0 < k <= 4096.  With metric NWKNN_METRIC_ONE_MINUS_DOT on unit vectors this is the
 * validator's `1 - dot(q, b)`. */
int nwknn_pair_distances(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, const int64_t* idx, int32_t k, int metric,
                         int64_t index_offset, float* out_dist, int out_location);

/* The validator's own arithmetic: out_dot[q][j] = the FLOAT64 dot product of query row q and base row idx[q][j] - index_offset (both as the
 * handle holds them: float32 or float16 values; NaN where the row does not exist) -- parquet_to_format.py:368-373 computes exactly this per
 * neighbour in Python.  Float64 sums agree with numpy's to ~1e-13, so `isclose(1 - dot, distance / 2)` can be decided from them. */
int nwknn_pair_dots(nwknn_handle h, const void* queries, int64_t nq, int dtype, int location, const int64_t* idx, int32_t k, int64_t index_offset,
                    double* out_dot, int out_location);

/* ---- device calibration (no reference counterpart) ----------------------------------------------------------------------------
 * TFLOP/s of a BARE MFMA loop on this device, on seeded random operands held in registers -- no LDS, no LDS-DMA, no global memory, no
 * epilogue -- at the register tile and waves per SIMD of the tile kernel of that dtype (f16: v_mfma_f32_16x16x32_f16, 8 x 4 accumulators,
 * two waves per SIMD; f32: v_mfma_f32_32x32x2_f32, 4 x 2, three workgroups per CU), launched back to back for `seconds` and timed over
 * the second half.  Under dense MFMA work the chip holds its clock well below the 2.4 GHz the nominal peaks assume (power limit), by a
 * device-dependent amount: this is what a kernel on THIS device can be held against (bench.py: roofline.frac_of_bare_loop).
 * out_clock_ghz (optional): the in-kernel clock, cycles per 100 MHz tick, median over workgroups. */
int nwknn_mfma_ceiling(nwknn_handle h, int dtype, double seconds, double* out_tflops, double* out_clock_ghz);

/* ---- synthetic inputs (no reference counterpart: generate_dataset.py needs the network) ----------------- */
/* Fill device buffer out[rows][d] with value(seed, row0 + r, c) in [-1,1): float32 on a 2^-23 grid, or fp16 on a
 * 2^-10 grid (the top bits of the same draw); the same integer recipe as oracle/cpu_oracle.gen_numpy /
 * gen_f16_numpy, so hosts can regenerate any row for checking. */
int nwknn_generate(nwknn_handle h, uint64_t seed, int64_t row0, int64_t rows, int32_t d, int dtype, void* out_device);

#ifdef __cplusplus
}
#endif
#endif /* NWKNN_H */
