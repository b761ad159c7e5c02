/*
 * nwknn_oracle.c -- CPU restatement of the brute-force kNN ground-truth hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  The product path (neighborhoodwatch_amd/) never imports, links or falls back to it.
 *
 * What it restates (citations relative to /root/reference):
 *   - the operator contract of neighborhoodwatch/cu_knn.py:243-245
 *         distances, indices = knn(dataset.astype(f32), query.astype(f32), k)
 *     = squared-L2, k smallest per query, ascending, (float32[q,k], int64[q,k]);
 *   - the alternative engines at cu_knn.py:246-248 (cosine) and cu_knn.py:256-257 (1 - q.b^T, topk);
 *   - the tile offset add of cu_knn.py:273 (caller passes row offsets);
 *   - the partial-list fold of neighborhoodwatch/merge.py:73-101.
 *
 * The arithmetic of knn() itself lives in un-vendored CUDA wheels (pylibraft-cu12 24.8.1,
 * cuvs-cu12 24.8.0, torch 2.5.1; pyproject.toml:12,33,47), absent from /root/reference, and the
 * reference's tests hold no golden vectors for it (tests/test_02_cu_knn.py asserts nothing).
 * KNN PARITY IS THEREFORE UNPINNED BY THE REFERENCE; this file defines the canonical arithmetic
 * (see DESIGN.md "Canonical arithmetic") and is cross-checked against an fp64 exact brute force
 * and a literal torch-CPU evaluation of cu_knn.py:256-257 (tests/golden/make_golden.py).
 * The merge half IS pinned: tests/golden/merge_*.npz were produced by running the reference's own
 * merge.py in this container.
 *
 * Canonical arithmetic (bit-exact contract shared with the HIP kernels):
 *   dot(q,b)  = fmaf chain over ascending k, accumulator starts at +0.0f
 *               (this is exactly what a K-loop of v_mfma_f32_32x32x2_f32 on one accumulator does)
 *   norm2(x)  = same chain with q=b=x
 *   sqeuclidean   : d = fmaxf(fmaf(-2.0f, dot, qn + bn), 0.0f)
 *   one_minus_dot : d = 1.0f - dot
 *   cosine        : d = 1.0f - dot / (sqrtf(qn) * sqrtf(bn))
 *   order         : ascending by (d, index); d compared through the usual sortable-uint32 map
 *                   (so -0.0 < +0.0 and every NaN, canonicalised to +qNaN, sorts last), lower index wins ties.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads of every parallel region.  The GPU boxes show 256 CPUs and grant the container a quota of 16: with OpenMP's default (one thread per
 * CPU shown) the suite's oracle calls spent their time in contention -- tests/test_gpu_f16_exact.py 68 s against 2.6 s with 16 threads.
 * cpu_oracle.py sets this from the cgroup quota when it loads the library; 0 = OpenMP's default. */
static int g_nwo_threads = 0;
void nwo_set_threads(int n) { g_nwo_threads = n > 0 ? n : 0; }
static int nwo_threads(void)
{
#ifdef _OPENMP
    return g_nwo_threads > 0 ? g_nwo_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
int nwo_get_threads(void) { return nwo_threads(); }

#define NWO_VERSION 1
#define NWO_METRIC_SQEUCLIDEAN 0
#define NWO_METRIC_ONE_MINUS_DOT 1
#define NWO_METRIC_COSINE 2

int nwo_version(void) { return NWO_VERSION; }

/* ---- synthetic generator: identical integer recipe on host and device ------------ */
static inline uint64_t nwo_fmix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

static inline float nwo_gen_value(uint64_t seed, uint64_t row, uint64_t col)
{
    uint64_t rk = nwo_fmix64(seed + row * 0x9E3779B97F4A7C15ull);
    uint64_t u  = nwo_fmix64(rk + (col + 1) * 0xD1342543DE82EF95ull);
    int32_t  v  = (int32_t)(u >> 40) - (1 << 23);          /* 24-bit signed */
    return (float)v * (1.0f / 8388608.0f);                 /* exact: [-1, 1) on a 2^-23 grid */
}

void nwo_gen(uint64_t seed, int64_t row0, int64_t rows, int32_t d, float* out)
{
#pragma omp parallel for schedule(static) num_threads(nwo_threads())
    for (int64_t r = 0; r < rows; ++r)
        for (int32_t c = 0; c < d; ++c)
            out[r * d + c] = nwo_gen_value(seed, (uint64_t)(row0 + r), (uint64_t)c);
}

/* ---- order-preserving float -> uint32 map -------------------------------------------------- */
static inline uint32_t nwo_sortable(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if (f != f) u = 0x7FC00000u;                /* every NaN ranks as +qNaN (last): x86 and gfx950 generate different NaN signs */
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float nwo_unsortable(uint32_t s)
{
    uint32_t u = (s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s;
    float f; memcpy(&f, &u, 4);
    return f;
}
uint32_t nwo_sortable_u32(float f) { return nwo_sortable(f); }

/* ---- canonical scalar pieces --------------------------------------------------------------- */
static inline float nwo_dot_chain(const float* a, const float* b, int32_t d)
{
    float acc = 0.0f;
    for (int32_t k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

void nwo_row_norms(const float* x, int64_t rows, int32_t d, float* out)
{
#pragma omp parallel for schedule(static) num_threads(nwo_threads())
    for (int64_t r = 0; r < rows; ++r) out[r] = nwo_dot_chain(x + r * d, x + r * d, d);
}

static inline float nwo_epilogue(int metric, float dot, float qn, float bn)
{
    switch (metric) {
    case NWO_METRIC_SQEUCLIDEAN:   return fmaxf(fmaf(-2.0f, dot, qn + bn), 0.0f);
    case NWO_METRIC_ONE_MINUS_DOT: return 1.0f - dot;
    default:                       return 1.0f - dot / (sqrtf(qn) * sqrtf(bn));
    }
}
float nwo_distance(int metric, float dot, float qn, float bn) { return nwo_epilogue(metric, dot, qn, bn); }

/* ---- bounded max-heap of 64-bit keys (sortable(dist) << 32 | local index) ------------------ */
typedef struct { uint64_t* key; int32_t n, cap; } nwo_heap;

static inline void nwo_heap_push(nwo_heap* h, uint64_t key)
{
    if (h->n < h->cap) {
        int32_t i = h->n++;
        h->key[i] = key;
        while (i > 0) {
            int32_t p = (i - 1) >> 1;
            if (h->key[p] >= h->key[i]) break;
            uint64_t t = h->key[p]; h->key[p] = h->key[i]; h->key[i] = t; i = p;
        }
    } else if (key < h->key[0]) {
        int32_t i = 0, n = h->n;
        h->key[0] = key;
        for (;;) {
            int32_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < n && h->key[l] > h->key[m]) m = l;
            if (r < n && h->key[r] > h->key[m]) m = r;
            if (m == i) break;
            uint64_t t = h->key[m]; h->key[m] = h->key[i]; h->key[i] = t; i = m;
        }
    }
}

static int nwo_cmp_u64(const void* a, const void* b)
{
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x > y) - (x < y);
}

static void nwo_heap_emit(nwo_heap* h, int32_t k, int64_t idx_offset, float* od, int64_t* oi)
{
    qsort(h->key, (size_t)h->n, sizeof(uint64_t), nwo_cmp_u64);
    for (int32_t j = 0; j < k; ++j) {
        if (j < h->n) { od[j] = nwo_unsortable((uint32_t)(h->key[j] >> 32)); oi[j] = idx_offset + (int64_t)(uint32_t)h->key[j]; }
        else          { od[j] = INFINITY; oi[j] = -1; }
    }
}

/* ---- scalar reference: the definition, literally ------------------------------------------- */
int nwo_knn_scalar(const float* base, int64_t nb, const float* queries, int64_t nq, int32_t d, int32_t k,
                   int metric, float* out_dist, int64_t* out_idx)
{
    if (!base || !queries || !out_dist || !out_idx || d <= 0 || k <= 0 || k > nb || nb >= (1ll << 32)) return -1;
    float* bn = (float*)malloc((size_t)nb * sizeof(float));
    nwo_row_norms(base, nb, d, bn);
#pragma omp parallel num_threads(nwo_threads())
    {
        uint64_t* keys = (uint64_t*)malloc((size_t)k * sizeof(uint64_t));
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; ++q) {
            const float* qv = queries + q * d;
            float qn = nwo_dot_chain(qv, qv, d);
            nwo_heap h = { keys, 0, k };
            for (int64_t b = 0; b < nb; ++b) {
                float dot = nwo_dot_chain(qv, base + b * d, d);
                float dist = nwo_epilogue(metric, dot, qn, bn[b]);
                nwo_heap_push(&h, ((uint64_t)nwo_sortable(dist) << 32) | (uint64_t)b);
            }
            nwo_heap_emit(&h, k, 0, out_dist + q * k, out_idx + q * k);
        }
        free(keys);
    }
    free(bn);
    return 0;
}

/* ---- fast path: same chains, 16 base rows per AVX2 lane group, 4 queries at a time ---------- */
#define NWO_RB 16   /* base rows per block (two ymm of independent per-row chains) */
#define NWO_QB 4    /* queries per inner group */

static void nwo_transpose_block(const float* base, int64_t b0, int64_t nb, int32_t d, float* T)
{
    /* T[k*16 + r] = base[b0+r][k]; rows past nb are zero (their results are discarded) */
    for (int r = 0; r < NWO_RB; ++r) {
        if (b0 + r < nb) { const float* src = base + (b0 + r) * d; for (int32_t k = 0; k < d; ++k) T[(size_t)k * NWO_RB + r] = src[k]; }
        else             { for (int32_t k = 0; k < d; ++k) T[(size_t)k * NWO_RB + r] = 0.0f; }
    }
}

__attribute__((target("avx2,fma")))
static void nwo_block_dots(const float* T, const float* const* qv, int nqv, int32_t d, float* dots /* [NWO_QB][16] */)
{
    __m256 a00 = _mm256_setzero_ps(), a01 = a00, a10 = a00, a11 = a00, a20 = a00, a21 = a00, a30 = a00, a31 = a00;
    const float* q0 = qv[0]; const float* q1 = qv[nqv > 1 ? 1 : 0];
    const float* q2 = qv[nqv > 2 ? 2 : 0]; const float* q3 = qv[nqv > 3 ? 3 : 0];
    for (int32_t k = 0; k < d; ++k) {
        __m256 t0 = _mm256_loadu_ps(T + (size_t)k * NWO_RB), t1 = _mm256_loadu_ps(T + (size_t)k * NWO_RB + 8);
        __m256 b;
        b = _mm256_broadcast_ss(q0 + k); a00 = _mm256_fmadd_ps(b, t0, a00); a01 = _mm256_fmadd_ps(b, t1, a01);
        b = _mm256_broadcast_ss(q1 + k); a10 = _mm256_fmadd_ps(b, t0, a10); a11 = _mm256_fmadd_ps(b, t1, a11);
        b = _mm256_broadcast_ss(q2 + k); a20 = _mm256_fmadd_ps(b, t0, a20); a21 = _mm256_fmadd_ps(b, t1, a21);
        b = _mm256_broadcast_ss(q3 + k); a30 = _mm256_fmadd_ps(b, t0, a30); a31 = _mm256_fmadd_ps(b, t1, a31);
    }
    _mm256_storeu_ps(dots + 0,  a00); _mm256_storeu_ps(dots + 8,  a01);
    _mm256_storeu_ps(dots + 16, a10); _mm256_storeu_ps(dots + 24, a11);
    _mm256_storeu_ps(dots + 32, a20); _mm256_storeu_ps(dots + 40, a21);
    _mm256_storeu_ps(dots + 48, a30); _mm256_storeu_ps(dots + 56, a31);
}

static void nwo_block_dots_scalar(const float* T, const float* const* qv, int nqv, int32_t d, float* dots)
{
    for (int qi = 0; qi < nqv; ++qi)
        for (int r = 0; r < NWO_RB; ++r) {
            float acc = 0.0f;
            for (int32_t k = 0; k < d; ++k) acc = fmaf(qv[qi][k], T[(size_t)k * NWO_RB + r], acc);
            dots[qi * NWO_RB + r] = acc;
        }
}

int nwo_has_avx2_fma(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

int nwo_knn(const float* base, int64_t nb, const float* queries, int64_t nq, int32_t d, int32_t k, int metric,
            float* out_dist, int64_t* out_idx, int nthreads)
{
    if (!base || !queries || !out_dist || !out_idx || d <= 0 || k <= 0 || k > nb || nb >= (1ll << 32)) return -1;
    if (metric < 0 || metric > 2) return -2;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = nwo_threads();
#else
    nthreads = 1;
#endif
    const int fast = nwo_has_avx2_fma();
    float* bn = (float*)malloc((size_t)nb * sizeof(float));
    float* qn = (float*)malloc((size_t)nq * sizeof(float));
    uint64_t* heaps = (uint64_t*)malloc((size_t)nq * (size_t)k * sizeof(uint64_t));
    int32_t* hn = (int32_t*)calloc((size_t)nq, sizeof(int32_t));
    if (!bn || !qn || !heaps || !hn) { free(bn); free(qn); free(heaps); free(hn); return -3; }
    nwo_row_norms(base, nb, d, bn);
    nwo_row_norms(queries, nq, d, qn);

    /* queries are split into contiguous per-thread slabs; every thread walks the whole base */
    const int64_t slab = 64;                       /* queries per work item */
    const int64_t nslab = (nq + slab - 1) / slab;
#pragma omp parallel num_threads(nthreads)
    {
        float* T = (float*)aligned_alloc(64, (size_t)d * NWO_RB * sizeof(float));
        float dots[NWO_QB * NWO_RB];
#pragma omp for schedule(dynamic, 1)
        for (int64_t s = 0; s < nslab; ++s) {
            const int64_t qa = s * slab, qb = (qa + slab < nq) ? qa + slab : nq;
            for (int64_t b0 = 0; b0 < nb; b0 += NWO_RB) {
                nwo_transpose_block(base, b0, nb, d, T);
                const int rows = (nb - b0 < NWO_RB) ? (int)(nb - b0) : NWO_RB;
                for (int64_t q = qa; q < qb; q += NWO_QB) {
                    const int nqv = (qb - q < NWO_QB) ? (int)(qb - q) : NWO_QB;
                    const float* qv[NWO_QB];
                    for (int i = 0; i < nqv; ++i) qv[i] = queries + (q + i) * d;
                    if (fast) nwo_block_dots(T, qv, nqv, d, dots); else nwo_block_dots_scalar(T, qv, nqv, d, dots);
                    for (int i = 0; i < nqv; ++i) {
                        nwo_heap h = { heaps + (q + i) * k, hn[q + i], k };
                        for (int r = 0; r < rows; ++r) {
                            float dist = nwo_epilogue(metric, dots[i * NWO_RB + r], qn[q + i], bn[b0 + r]);
                            nwo_heap_push(&h, ((uint64_t)nwo_sortable(dist) << 32) | (uint64_t)(b0 + r));
                        }
                        hn[q + i] = h.n;
                    }
                }
            }
            for (int64_t q = qa; q < qb; ++q) {
                nwo_heap h = { heaps + q * k, hn[q], k };
                nwo_heap_emit(&h, k, 0, out_dist + q * k, out_idx + q * k);
            }
        }
        free(T);
    }
    free(bn); free(qn); free(heaps); free(hn);
    return 0;
}

/* ---- merge of partial top-k lists (merge.py:56-101 folds them pairwise; the fold of sorted
 *      [Q,2k] blocks keeping k equals the global k smallest of the concatenation).  Ties between
 *      equal distances are left to numpy's unstable argsort by the reference; the canonical
 *      order here is (distance, index).  Entries with index < 0 are treated as empty. -------- */
typedef struct { uint32_t s; int64_t i; } nwo_pair;
static int nwo_cmp_pair(const void* a, const void* b)
{
    const nwo_pair* x = (const nwo_pair*)a; const nwo_pair* y = (const nwo_pair*)b;
    if (x->s != y->s) return (x->s > y->s) - (x->s < y->s);
    return (x->i > y->i) - (x->i < y->i);
}

int nwo_merge(const float* dists, const int64_t* idx, int32_t n_lists, int64_t nq, int32_t k_in, int32_t k_out,
              float* out_d, int64_t* out_i)
{
    if (!dists || !idx || !out_d || !out_i || n_lists <= 0 || k_in <= 0 || k_out <= 0) return -1;
    const int64_t w = (int64_t)n_lists * k_in;
#pragma omp parallel num_threads(nwo_threads())
    {
        nwo_pair* p = (nwo_pair*)malloc((size_t)w * sizeof(nwo_pair));
#pragma omp for schedule(static)
        for (int64_t q = 0; q < nq; ++q) {
            int64_t n = 0;
            for (int32_t l = 0; l < n_lists; ++l)
                for (int32_t j = 0; j < k_in; ++j) {
                    const int64_t at = ((int64_t)l * nq + q) * k_in + j;   /* lists are [n_lists][Q][k] */
                    if (idx[at] < 0) continue;
                    p[n].s = nwo_sortable(dists[at]); p[n].i = idx[at]; ++n;
                }
            qsort(p, (size_t)n, sizeof(nwo_pair), nwo_cmp_pair);
            for (int32_t j = 0; j < k_out; ++j) {
                if (j < n) { out_d[q * k_out + j] = nwo_unsortable(p[j].s); out_i[q * k_out + j] = p[j].i; }
                else       { out_d[q * k_out + j] = INFINITY; out_i[q * k_out + j] = -1; }
            }
        }
        free(p);
    }
    return 0;
}

/* ---- MaxSim late interaction (BASELINE config 5).  Not in the reference (its `ck` is flat token kNN, colbert_knn.py:84-126);
 *      this restates the ENGINE's canonical arithmetic so the HIP kernel can be checked bit for bit:
 *        dot(q_i, d_j)   = ascending-k fmaf chain
 *        m_i             = max_j dot(q_i, d_j)
 *        score           = balanced pairwise tree sum of m_0..m_{tq-1} in natural order (tq a power of two)
 *      ranked by (-score, doc id).  docs: [nd][td][d], queries: [nq][tq][d]. --------------------------------------- */
/* m[i] = max over the doc's td token rows of the ascending-k fmaf chain dot(q_i, d_j); the doc's tokens arrive transposed in
 * blocks of 16 rows (nwo_transpose_block) so that the AVX2 path runs 16 independent per-row chains per query token -- the chains
 * themselves are the scalar ones (same operations in the same order), so both paths give the same bits. */
static void nwo_maxsim_token_max(const float* T, int nblk, int32_t td, const float* qtok, int32_t tq, int32_t d, int fast, float* m)
{
    float dots[NWO_QB * NWO_RB];
    for (int32_t i0 = 0; i0 < tq; i0 += NWO_QB) {
        const int nqv = (tq - i0 < NWO_QB) ? (int)(tq - i0) : NWO_QB;
        const float* qv[NWO_QB];
        for (int i = 0; i < nqv; ++i) { qv[i] = qtok + (size_t)(i0 + i) * d; m[i0 + i] = -INFINITY; }
        for (int blk = 0; blk < nblk; ++blk) {
            const int rows = (td - blk * NWO_RB < NWO_RB) ? (td - blk * NWO_RB) : NWO_RB;
            if (fast) nwo_block_dots(T + (size_t)blk * d * NWO_RB, qv, nqv, d, dots);
            else nwo_block_dots_scalar(T + (size_t)blk * d * NWO_RB, qv, nqv, d, dots);
            for (int i = 0; i < nqv; ++i)
                for (int r = 0; r < rows; ++r) m[i0 + i] = fmaxf(m[i0 + i], dots[i * NWO_RB + r]);     /* ascending j, as the scalar loop */
        }
    }
}

static float nwo_tree_sum(float* m, int32_t tq)
{
    for (int32_t step = 1; step < tq; step <<= 1)          /* butterfly: every element ends up holding the tree sum */
        for (int32_t i = 0; i < tq; i += 2 * step)
            for (int32_t t = 0; t < step; ++t) {
                const float a = m[i + t] + m[i + t + step];
                m[i + t] = a; m[i + t + step] = a;
            }
    return m[0];
}

int nwo_maxsim(const float* docs, int64_t nd, int32_t td, const float* queries, int64_t nq, int32_t tq, int32_t d, int32_t k,
               float* out_score, int64_t* out_doc)
{
    if (!docs || !queries || !out_score || !out_doc || k <= 0 || k > nd || tq <= 0 || (tq & (tq - 1)) || tq > 64 || nd >= (1ll << 32)) return -1;
    if (td <= 0 || d <= 0) return -1;
    const int fast = nwo_has_avx2_fma();
    const int nblk = (td + NWO_RB - 1) / NWO_RB;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = nwo_threads();
#endif
    /* Work is dealt out in chunks of documents, every thread keeping its own k-heap per query (few queries against many documents:
     * the sampled checks of the full-size tests), merged at the end; the k best by key do not depend on the visiting order.  When
     * that would need too much memory the queries are dealt out instead, one heap each. */
    const int by_docs = (double)nq * (double)k * 8.0 * (double)nthreads <= 512.0 * 1048576.0;
    if (by_docs) {
        uint64_t* heaps = (uint64_t*)malloc((size_t)nthreads * (size_t)nq * (size_t)k * sizeof(uint64_t));
        int32_t* hn = (int32_t*)calloc((size_t)nthreads * (size_t)nq, sizeof(int32_t));
        if (!heaps || !hn) { free(heaps); free(hn); return -3; }
        const int64_t chunk = 256;
        const int64_t nchunk = (nd + chunk - 1) / chunk;
#pragma omp parallel num_threads(nthreads)
        {
            int tnum = 0;
#ifdef _OPENMP
            tnum = omp_get_thread_num();
#endif
            float* T = (float*)aligned_alloc(64, (size_t)nblk * d * NWO_RB * sizeof(float));
            float m[64];
            uint64_t* my = heaps + (size_t)tnum * nq * k;
            int32_t* myn = hn + (size_t)tnum * nq;
#pragma omp for schedule(dynamic, 1)
            for (int64_t c = 0; c < nchunk; ++c) {
                const int64_t d0 = c * chunk, d1 = (d0 + chunk < nd) ? d0 + chunk : nd;
                for (int64_t doc = d0; doc < d1; ++doc) {
                    for (int blk = 0; blk < nblk; ++blk)
                        nwo_transpose_block(docs + (size_t)doc * td * d, (int64_t)blk * NWO_RB, td, d, T + (size_t)blk * d * NWO_RB);
                    for (int64_t q = 0; q < nq; ++q) {
                        nwo_maxsim_token_max(T, nblk, td, queries + (size_t)q * tq * d, tq, d, fast, m);
                        const float score = nwo_tree_sum(m, tq);
                        nwo_heap h = { my + (size_t)q * k, myn[q], k };
                        nwo_heap_push(&h, ((uint64_t)nwo_sortable(-score) << 32) | (uint64_t)doc);
                        myn[q] = h.n;
                    }
                }
            }
            free(T);
        }
        /* merge the per-thread heaps: the k smallest keys of their union */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int64_t q = 0; q < nq; ++q) {
            uint64_t* all = (uint64_t*)malloc((size_t)nthreads * (size_t)k * sizeof(uint64_t));
            size_t n = 0;
            for (int t = 0; t < nthreads; ++t) {
                const int32_t cnt = hn[(size_t)t * nq + q];
                memcpy(all + n, heaps + ((size_t)t * nq + q) * k, (size_t)cnt * sizeof(uint64_t));
                n += (size_t)cnt;
            }
            qsort(all, n, sizeof(uint64_t), nwo_cmp_u64);
            for (int32_t j = 0; j < k; ++j) {
                out_score[q * k + j] = -nwo_unsortable((uint32_t)(all[j] >> 32));
                out_doc[q * k + j] = (int64_t)(uint32_t)all[j];
            }
            free(all);
        }
        free(heaps); free(hn);
        return 0;
    }
#pragma omp parallel num_threads(nthreads)
    {
        uint64_t* keys = (uint64_t*)malloc((size_t)k * sizeof(uint64_t));
        float* T = (float*)aligned_alloc(64, (size_t)nblk * d * NWO_RB * sizeof(float));
        float m[64];
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; ++q) {
            nwo_heap h = { keys, 0, k };
            for (int64_t doc = 0; doc < nd; ++doc) {
                for (int blk = 0; blk < nblk; ++blk)
                    nwo_transpose_block(docs + (size_t)doc * td * d, (int64_t)blk * NWO_RB, td, d, T + (size_t)blk * d * NWO_RB);
                nwo_maxsim_token_max(T, nblk, td, queries + (size_t)q * tq * d, tq, d, fast, m);
                const float score = nwo_tree_sum(m, tq);
                nwo_heap_push(&h, ((uint64_t)nwo_sortable(-score) << 32) | (uint64_t)doc);
            }
            qsort(h.key, (size_t)h.n, sizeof(uint64_t), nwo_cmp_u64);
            for (int32_t j = 0; j < k; ++j) {
                out_score[q * k + j] = -nwo_unsortable((uint32_t)(h.key[j] >> 32));
                out_doc[q * k + j] = (int64_t)(uint32_t)h.key[j];
            }
        }
        free(keys); free(T);
    }
    return 0;
}
