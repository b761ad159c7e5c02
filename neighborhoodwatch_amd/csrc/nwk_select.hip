// nwk_select.hip -- rescoring of pre-screened pairs, per-query list compaction (K3), finalize, partial-list merge (K5)  (gfx950 only; shared pieces and the overall design: nwk_device.h)
#include "nwk_device.h"

// One wave per query (blockDim = 64 * waves, dynamic LDS = waves * cap * 8 bytes).  The keys the pre-screen listed carry the
// fp16 ESTIMATE of the distance; with the error bound e(q,b) the exact value lies in [est - e, est + e].  Phases:
//   1. bounds of every list entry into LDS (entries of the compacted prefix [0, cnt0) are exact: lb = ub = their distance);
//   2. U = k-th smallest UPPER bound (bisection over the sortable 32-bit space, ballot counts): at least k entries are <= U,
//      so the k-th key after this chunk is <= U and every candidate whose LOWER bound is > U can be dropped unrescored;
//   3. the survivors are packed and rescored 64 at a time: ascending-k fmaf chain over the prepared fp32 rows (the chain the
//      fp32 MFMA accumulator performs), canonical epilogue, 64-bit key written back in place; dropped entries become ~0
//      (the sentinel compaction already sorts last).
#define RS_QC 1024                    // query elements staged in LDS at a time (4 KiB per wave)
#define RS_SEG_BYTES 64               // cooperative row fetch: bytes of every row per step (two stages of 64 rows x this per wave)

// |estimate - canonical distance| of a listed pair is at most this: the bound E(q,b) the tile kernel pruned under (DESIGN.md "Bound"),
// taken to the distance domain, plus the roundings of the estimate and of these lines themselves.
template <int METRIC>
__device__ __forceinline__ float rescore_err(const NwkRescoreParams& p, float row_err, float qn, float bn, float est)
{
    float err = row_err + p.eps_half * bn + p.aq * __builtin_amdgcn_sqrtf(bn);          // >= |dot16 - dot32|
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) err = 2.0f * err;
    if (METRIC == NWKNN_METRIC_COSINE) err = err * 1.0001f + 2.0e-6f * (fabsf(est) + 1.0f);
    else err = err * 1.0001f + 2.0e-6f * (fabsf(est) + qn + bn + 1.0f);                  // roundings of est and of these lines
    return err;
}

template <int METRIC, bool ROWS16, bool COOP>
__global__ __launch_bounds__(256) void nwk_rescore_kernel(const NwkRescoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) u32 rs_dyn[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
    const long long q = (long long)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + wv));
    u32* s_ub = rs_dyn + (size_t)wv * 2 * p.cap;          // later: the packed survivor slots
    u32* s_lb = s_ub + p.cap;
    const u32 start = p.cnt0[q];
    u32 end = p.cnt[q];
    const bool over = end > (u32)p.cap;                    // the list overflowed: its length is left alone (the compaction flags the query for repair)
    if (over) end = (u32)p.cap;
    if (start >= end) return;
    const float* __restrict__ qrow = ROWS16 ? nullptr : p.qp + q * p.dp;
    const _Float16* __restrict__ qrow16 = ROWS16 ? (const _Float16*)p.qp + q * p.dp : nullptr;      // exact fp16 mode: the rows ARE the data
    const float qn = p.qn[q];
    u64* list = p.buf + q * p.cap;

    u32 nsurv = 0, U = 0xFFFFFFFFu;
    const float row_err = p.eps_half * qn + p.ab * __builtin_amdgcn_sqrtf(qn) + p.cst;
    if (p.defer && !(p.prune && end >= (u32)p.k)) {         // deferred rescoring, fewer than k entries: nothing can be dropped, the threshold stays open
        if (lane == 0) p.skip[q] = 1u;
        return;
    }
    if (p.prune && end >= (u32)p.k) {
        for (u32 e0 = 0; e0 < end; e0 += 64) {
            const u32 e = e0 + lane;
            if (e < end) {
                const u64 key = list[e];
                if (e < start) { s_ub[e] = (u32)(key >> 32); s_lb[e] = 0u; }
                else if (key == ~0ull) { s_ub[e] = 0xFFFFFFFFu; s_lb[e] = 0xFFFFFFFFu; }       // empty slot of a dense chunk
                else {
                    const float est = __uint_as_float((u32)(key >> 32));
                    const float bn = p.bn[(u32)key];
                    const float err = rescore_err<METRIC>(p, row_err, qn, bn, est);
                    float ub = est + err;
                    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ub = fmaxf(ub, 0.0f);       // the canonical distance is clamped at 0
                    u32 sub = sortable_u32(ub), slb = sortable_u32(est - err);
                    if (METRIC == NWKNN_METRIC_COSINE && !(cosine_norm_ok(qn) && cosine_norm_ok(bn))) {
                        sub = 0xFFC00000u; slb = 0u;                                   // no estimate for degenerate rows: [-inf, NaN rank]
                    }
                    s_ub[e] = sub;
                    s_lb[e] = slb;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        u32 lo = 0u, hi = 0xFFFFFFFFu;
        if (p.cap <= 1024) {
            // lists of up to 1024 entries: a lane keeps its (at most) sixteen upper bounds in registers, so a bisection step is sixteen
            // compares and scalar bit counts with no LDS round trip in it -- the 32 steps x 9 rounds of ds_read + wait were most of a
            // pruning pass, which the deferred walk runs once per chunk (slots past `end` hold the largest value: mid never reaches it)
            u32 r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const u32 e = (u32)(j * 64 + lane);
                r[j] = e < end ? s_ub[e] : 0xFFFFFFFFu;
            }
            while (lo < hi) {                               // smallest x with  #{ub <= x} >= k   (wave-uniform)
                const u32 mid = lo + ((hi - lo) >> 1);
                u32 c = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) c += (u32)__popcll(__ballot(r[j] <= mid));
                if (c >= (u32)p.k) hi = mid; else lo = mid + 1u;
            }
        } else {
            while (lo < hi) {
                const u32 mid = lo + ((hi - lo) >> 1);
                u32 c = 0;
                for (u32 e0 = 0; e0 < end; e0 += 64) {
                    const u32 e = e0 + lane;
                    c += (u32)__popcll(__ballot(e < end && s_ub[e] <= mid));
                }
                if (c >= (u32)p.k) hi = mid; else lo = mid + 1u;
            }
        }
        U = lo;
        // (deferred) the exact prefix is sorted: its entries at or below U are a leading run, the others can no longer be among the k nearest
        // (exact value > U >= the k-th smallest distance) and are dropped with the estimates that cannot -- otherwise the prefix turns into
        // dead weight as the walk displaces its members, and the list fills up with k entries nobody needs
        u32 nstart = start;
        if (p.defer) {
            nstart = 0;
            for (u32 e0 = 0; e0 < start; e0 += 64) {
                const u32 e = e0 + lane;
                nstart += (u32)__popcll(__ballot(e < start && s_ub[e] <= U));
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (u32 e0 = start; e0 < end; e0 += 64) {
            const u32 e = e0 + lane;
            const bool live = e < end;
            const bool keep = live && s_lb[e] <= U;
            const u64 mk = __ballot(keep);
            const u32 pos = nsurv + (u32)__popcll(mk & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (keep) s_ub[pos] = e;                        // s_ub is free after the bisection: it now holds the packed survivor slots
            else if (live && (!p.defer || over)) list[e] = ~0ull;
            nsurv += (u32)__popcll(mk);
        }
        __builtin_amdgcn_wave_barrier();
        // A deferred pass leaves a query alone only while its pruned list stays short.  Where the bound cannot tell many candidates apart
        // (near-duplicates, scores closer than the fp16 error: MaxSim's 32-term bound is such a case and keeps per-chunk rescoring) the
        // estimates would pile up until the list overflows and the query goes through overflow repair -- a re-walk of the whole base.
        // Such a query -- and one whose list is full already -- is rescored NOW, as before round 5: skip[q] = 0 sends it through the
        // compaction that follows every chunk (the others return from it at once), which sorts its exact keys and resets its list to k.
        const bool stay = p.defer && !over && nstart + nsurv <= (u32)p.defer_limit;
        if (p.defer && lane == 0) p.skip[q] = stay ? 1u : 0u;
        if (stay) {
            // Deferred rescoring (every pre-screened chunk but the last of a walk).  Rescoring each chunk's survivors at once costs 4 KiB
            // of row per pair, ~90 pairs per query and chunk, of which only ~k (1 - 1/(1 + growth)) end up among the k nearest of that
            // chunk and far fewer at the end of the walk: on small bases that was 11 % of the search, at the Infinity Cache's / HBM's rate.
            // Here the survivors keep their ESTIMATES: they are packed behind the exact prefix [0, start) as they are, the list is cut to
            // them, and the threshold of the next chunk becomes U -- an upper bound of the k-th smallest canonical distance among the rows
            // seen so far (k listed entries lie at or below it), so every pair that can still be among the k nearest passes the tile
            // kernel's test (conservative for "canonical distance <= threshold") and, its lower bound being <= its distance <= every later
            // U (they only fall: the k entries that defined one U are kept and bound the next), is kept by every later pruning.  The last
            // chunk of the walk rescores what is left -- entries of all chunks, estimates every one -- and the compaction orders exact keys
            // as before: same bits, a third of the rescoring.  No sort is needed in between either (the pruning packs the list).
            // (slot nstart + i <= start + i <= the slot entry i came from, and a round's stores wait for its loads: see the packed write-back below)
            for (u32 i0 = 0; i0 < nsurv; i0 += 64) {
                const u32 i = i0 + lane;
                const bool live = i < nsurv;
                const u64 key = live ? list[s_ub[i]] : 0ull;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (live) list[nstart + i] = key;
            }
            if (lane == 0) {
                p.cnt0_w[q] = nstart;
                p.cnt[q] = nstart + nsurv;                  // >= k: the k entries that define U are all kept
                p.tau[q] = ((u64)U << 32) | 0xFFFFFFFFull;
            }
            return;
        }
    } else {
        nsurv = end - start;
    }
    const bool packed = p.prune && end >= (u32)p.k;

    // Phase 3.  A lane owns one pair and runs its chain in ascending k; what matters is how the rows arrive.  The first version
    // fetched 32 B per lane and iteration and waited for it before its 8 FMAs (128 exposed round trips per pair, hidden only by
    // occupancy; every 128-B line was visited by eight separate loads while 20 waves thrashed the 32 KiB L1: ~3 TB/s of rows).
    // Now a lane takes a whole 128-B line of its row at a time (8 x 16 B back to back: one miss, seven L1 hits), the next line is
    // requested before the current one is consumed (two register sets), and the query row -- the same for all 64 lanes -- is
    // staged through LDS in chunks of RS_QC elements and read back as broadcasts.
    const int ES = ROWS16 ? 2 : 4;                                      // bytes per element
    char* const s_q = (char*)(rs_dyn + (size_t)wpb * 2 * p.cap) + (size_t)wv * RS_QC * 4;
    const char* const qbytes = ROWS16 ? (const char*)qrow16 : (const char*)qrow;
    const int line_elems = 128 / ES;                                    // 32 floats or 64 halves; dp is a multiple of it
    // The invariant behind the pre-screen, checked on every pair that is rescored: the estimate the tile kernel listed the pair under and
    // the canonical value computed here differ by no more than the bound the tile kernel pruned with.  The bound allows 4 d u for the fp16
    // MFMA's internal summation, whose order the ISA does not define; this is where that allowance is held against the hardware.
    u32 n_chk = 0, n_bad = 0;                                           // wave-uniform
    float worst = 0.0f;                                                 // per lane: max |estimate - exact| / bound
    for (u32 i0 = 0; i0 < nsurv; i0 += 64) {
        const u32 i = i0 + lane;
        const bool live = i < nsurv;
        const u32 e = live ? (packed ? s_ub[i] : start + i) : start;
        const u64 old = live ? list[e] : ~0ull;
        const bool valid = old != ~0ull;                               // dense chunks leave ~0 in the slots of padding columns
        const u32 col = valid ? (u32)old : 0u;
        const char* __restrict__ brow = (const char*)p.bp + (long long)col * p.dp * ES;
        float acc = 0.0f;
        if constexpr (COOP) {
            // Round 5.  The lane-per-row loads below ask the CU's address unit for 64 different cache lines per instruction, 16 bytes of
            // each: on chunks whose rows sit in L2 / the Infinity Cache -- the six growth chunks right behind the dense one, where two
            // thirds of all pairs are rescored -- that unit, not memory, was the bound (~6.8 us per round of 64 pairs, one unit per CU:
            // 5.2-5.9 ms per chunk of a C3 shard whatever the chunk's size).  Here the WAVE fetches the 64 rows of a round together:
            // four consecutive lanes take the four 16-byte pieces of one row's 64-byte segment, an instruction covers 16 rows (16 lines,
            // each used whole over two steps), straight into this wave's own LDS stage (global_load_lds, no register round trip); lane i
            // then reads ITS row's segment back from LDS and runs the same ascending-k chain.  Same bits; two stages, the next
            // segment in flight while this one is consumed; no barrier (the stage is private to the wave).
            constexpr int SEG = RS_SEG_BYTES, LPR = SEG / 16, RPI = 64 / LPR, NI = 64 / RPI;
            char* const stage = (char*)(rs_dyn + (size_t)wpb * 2 * p.cap) + (size_t)wpb * RS_QC * 4 + (size_t)wv * 2 * 64 * SEG;
            __builtin_amdgcn_wave_barrier();
            s_lb[lane] = col;                                           // (the lower bounds are not needed after the packing)
            __builtin_amdgcn_wave_barrier();
            const char* src[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = j * RPI + lane / LPR;
                src[j] = (const char*)p.bp + (long long)s_lb[r] * p.dp * ES + (lane % LPR) * 16;
            }
            auto issue = [&](int buf, int kbyte) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kbyte),
                                                     (__attribute__((address_space(3))) void*)(stage + buf * 64 * SEG + j * 1024), 16, 0, 0);
            };
            for (int kc = 0; kc < p.dp; kc += RS_QC) {                  // wave-uniform; one pass for d <= 1024
                const int n = (p.dp - kc < RS_QC) ? p.dp - kc : RS_QC;
                const int steps = n * ES / SEG;                         // n is a multiple of 32 floats / 64 halves
                issue(0, kc * ES);
                f32x4 qt[RS_QC * 4 / 1024];
#pragma unroll
                for (int j = 0; j < RS_QC * 4 / 1024; ++j)
                    if (lane * 16 + j * 1024 < n * ES) qt[j] = *(const f32x4*)(qbytes + (size_t)kc * ES + lane * 16 + j * 1024);
                __builtin_amdgcn_wave_barrier();                        // the previous chunk has been read by every lane
#pragma unroll
                for (int j = 0; j < RS_QC * 4 / 1024; ++j)
                    if (lane * 16 + j * 1024 < n * ES) *(f32x4*)(s_q + lane * 16 + j * 1024) = qt[j];
                __builtin_amdgcn_wave_barrier();
                for (int st = 0; st < steps; ++st) {
                    // stage (st + 1) & 1 was read during step st - 1; those reads fed FMAs that have been issued: it may be overwritten
                    if (st + 1 < steps) {
                        issue((st + 1) & 1, kc * ES + (st + 1) * SEG);
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    const char* mine = stage + (st & 1) * 64 * SEG + lane * SEG;
                    f32x4 ln[LPR];
#pragma unroll
                    for (int c = 0; c < LPR; ++c) ln[c] = *(const f32x4*)(mine + c * 16);
                    const int kq = st * (SEG / ES);                     // element offset inside the staged query chunk
                    if (ROWS16) {
#pragma unroll
                        for (int c = 0; c < LPR; ++c) {
                            const f16x8 bv = __builtin_bit_cast(f16x8, ln[c]);
                            const f16x8 qv = *(const f16x8*)(s_q + (size_t)kq * 2 + c * 16);
#pragma unroll
                            for (int t = 0; t < 8; ++t) acc = __fmaf_rn((float)qv[t], (float)bv[t], acc);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < LPR / 2; ++c) {             // prepared layout: [k k+2 k+4 k+6 | k+1 k+3 k+5 k+7]
                            const f32x4 be = ln[2 * c], bo = ln[2 * c + 1];
                            const f32x4 qe = *(const f32x4*)(s_q + (size_t)kq * 4 + c * 32), qo = *(const f32x4*)(s_q + (size_t)kq * 4 + c * 32 + 16);
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                acc = __fmaf_rn(qe[t], be[t], acc);
                                acc = __fmaf_rn(qo[t], bo[t], acc);
                            }
                        }
                    }
                    asm volatile("" ::: "memory");                      // the reads of this stage stay above the next iteration's issue
                }
            }
        }
        f32x4 la[8], lb[8];                                             // two lines of this lane's row
        auto load_line = [&](f32x4 (&dst)[8], int k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = *(const f32x4*)(brow + (size_t)k * ES + j * 16);
        };
        auto chain_line = [&](const f32x4 (&ln)[8], int kq) {           // kq: element offset inside the staged query chunk
            if (ROWS16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f16x8 bv = __builtin_bit_cast(f16x8, ln[j]);
                    const f16x8 qv = *(const f16x8*)(s_q + (size_t)kq * 2 + j * 16);
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc = __fmaf_rn((float)qv[t], (float)bv[t], acc);      // natural order: ascending k
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {                           // prepared layout: [k k+2 k+4 k+6 | k+1 k+3 k+5 k+7]
                    const f32x4 be = ln[2 * j], bo = ln[2 * j + 1];
                    const f32x4 qe = *(const f32x4*)(s_q + (size_t)kq * 4 + j * 32), qo = *(const f32x4*)(s_q + (size_t)kq * 4 + j * 32 + 16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc = __fmaf_rn(qe[t], be[t], acc);
                        acc = __fmaf_rn(qo[t], bo[t], acc);
                    }
                }
            }
        };
        for (int kc = 0; !COOP && kc < p.dp; kc += RS_QC) {             // wave-uniform; one pass for d <= 1024
            const int n = (p.dp - kc < RS_QC) ? p.dp - kc : RS_QC;
            const int nl = n / line_elems;                              // n is a multiple of line_elems
            load_line(la, kc);
            f32x4 qt[RS_QC * 4 / 1024];                                 // this chunk of the query row: <= 4 x 16 B per lane
#pragma unroll
            for (int j = 0; j < RS_QC * 4 / 1024; ++j)
                if (lane * 16 + j * 1024 < n * ES) qt[j] = *(const f32x4*)(qbytes + (size_t)kc * ES + lane * 16 + j * 1024);
            __builtin_amdgcn_wave_barrier();                            // the previous chunk has been read by every lane
#pragma unroll
            for (int j = 0; j < RS_QC * 4 / 1024; ++j)
                if (lane * 16 + j * 1024 < n * ES) *(f32x4*)(s_q + lane * 16 + j * 1024) = qt[j];
            __builtin_amdgcn_wave_barrier();
            for (int ln = 0; ln + 2 <= nl; ln += 2) {                   // two lines per pass, the loads one line ahead of the chain
                load_line(lb, kc + (ln + 1) * line_elems);
                chain_line(la, ln * line_elems);
                const int nx = (ln + 2 < nl) ? ln + 2 : nl - 1;         // past the end: a line that is there (and is not used)
                load_line(la, kc + nx * line_elems);
                chain_line(lb, (ln + 1) * line_elems);
            }
            if (nl & 1) chain_line(la, (nl - 1) * line_elems);
        }
        const float bnc = p.bn[col];
        const float dist = distance_epilogue<METRIC>(acc, qn, bnc);
        // Survivors go back PACKED behind the compacted prefix (slot start + i; i <= e - start, and the slots a round writes lie below every
        // slot a later round still has to read), and the list's length becomes start + nsurv below: the compaction that follows sorts
        // ~2 k entries instead of every slot the tile kernel filled (a C3 shard: ~520 listed, ~90 rescored per query and chunk).
        const u64 fresh = valid ? (((u64)sortable_u32(dist) << 32) | col) : ~0ull;
        if (packed && !over) { if (live) list[start + i] = fresh; }
        else if (valid) list[e] = fresh;
        {
            const float est = __uint_as_float((u32)(old >> 32));
            const float err = rescore_err<METRIC>(p, row_err, qn, bnc, est);
            bool judged = valid;
            if (METRIC == NWKNN_METRIC_COSINE) judged = judged && cosine_norm_ok(qn) && cosine_norm_ok(bnc);    // degenerate rows carry no estimate
            const bool clamped = METRIC == NWKNN_METRIC_SQEUCLIDEAN && dist == 0.0f && est < 0.0f;              // the canonical distance is clamped at 0
            const float dev = clamped ? 0.0f : fabsf(dist - est);
            const bool bad = judged && !(dev <= err * p.chk_scale);    // NaN counts as a violation
            n_chk += (u32)__popcll(__ballot(judged));
            n_bad += (u32)__popcll(__ballot(bad));
            const float ratio = dev / err;
            if (judged) worst = fmaxf(worst, ratio == ratio ? ratio : __builtin_inff());
        }
    }
    if (packed && !over && lane == 0) p.cnt[q] = start + nsurv;
    if (p.chk) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
        if (lane == 0 && n_chk) {
            atomicAdd(p.chk, (u64)n_chk);
            if (n_bad) atomicAdd(p.chk + 1, (u64)n_bad);
            atomicMax((u32*)(p.chk + 2), __float_as_uint(worst));       // non-negative floats order like their bit patterns
        }
    }
}

// ------------------------------------------------------------------------------------------------
// compaction: sort one query's candidate list, (optionally) drop duplicate keys, keep k, tighten tau
//   dynamic LDS: [cap] u64 keys | [2][cap] u32 scan buffers | 16 B scratch   (one array: keeps the base 16-B aligned)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_compact_kernel(const NwkCompactParams p)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    u64* s = (u64*)dyn;
    u32* scan0 = (u32*)(dyn + (size_t)p.cap * 8);
    u32* scan1 = scan0 + p.cap;
    u32* misc = scan1 + p.cap;

    const int tid = threadIdx.x;
    long long q;
    if (p.qtile_list) q = (long long)p.qtile_list[blockIdx.x / p.qtile_rows] * p.qtile_rows + (blockIdx.x % p.qtile_rows);
    else q = blockIdx.x;

    if (p.skip && p.skip[q]) return;                      // deferred rescoring: estimates behind the prefix, pruned and packed by nwk_rescore_kernel
    u32 n = p.dense_count > 0 ? (u32)p.dense_count : p.cnt[q];
    const bool overflow = n > (u32)p.cap;
    if (overflow) n = p.cap;
    // nothing appended since the last compaction: the list is already the sorted top-k
    // (pre-screened walks: ... and nothing behind the sorted exact prefix -- a deferred pruning pass can leave exactly k entries, some of them
    // rescored a moment ago and in no order)
    if (!overflow && !p.dedupe && p.dense_count == 0 && n <= (u32)p.k && p.tau[q] != ~0ull && (!p.cnt0 || p.cnt0[q] >= n)) return;

    u32 np2 = 2;
    while (np2 < n) np2 <<= 1;
    u64* list = p.buf + q * p.cap;
    for (u32 i = tid; i < np2; i += 256) s[i] = (i < n) ? list[i] : ~0ull;
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    for (u32 k2 = 2; k2 <= np2; k2 <<= 1) {
        for (u32 j = k2 >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < np2; i += 256) {
                const u32 ixj = i ^ j;
                if (ixj > i) {
                    const u64 a = s[i], b = s[ixj];
                    const bool up = (i & k2) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const u32 k = (u32)p.k;
    if (p.dedupe) {
        // a rescan can re-find entries that are already listed: keep the first occurrence of every key
        for (u32 i = tid; i < np2; i += 256) scan0[i] = (s[i] != ~0ull && (i == 0 || s[i] != s[i - 1])) ? 1u : 0u;
        __syncthreads();
        u32* src = scan0;
        u32* dst = scan1;
        for (u32 off = 1; off < np2; off <<= 1) {                  // inclusive Hillis-Steele scan, ping-pong
            for (u32 i = tid; i < np2; i += 256) dst[i] = src[i] + (i >= off ? src[i - off] : 0u);
            __syncthreads();
            u32* t = src; src = dst; dst = t;
        }
        const u32 nvalid = src[np2 - 1];
        for (u32 i = tid; i < np2; i += 256) {
            const bool keep = s[i] != ~0ull && (i == 0 || s[i] != s[i - 1]);
            const u32 at = src[i] - 1;
            if (keep && at < k) {
                list[at] = s[i];
                if (at == k - 1) p.tau[q] = s[i];
            }
        }
        if (tid == 0) {
            p.cnt[q] = nvalid < k ? nvalid : k;
            if (p.cnt0) p.cnt0[q] = nvalid < k ? nvalid : k;
            if (nvalid < k) p.tau[q] = ~0ull;
        }
    } else {
        // sentinels (padding columns of the dense first chunk) sort last: find where they start
        for (u32 i = tid; i < np2; i += 256)
            if (s[i] != ~0ull && (i + 1 == np2 || s[i + 1] == ~0ull)) misc[0] = i + 1;
        __syncthreads();
        const u32 nvalid = misc[0];
        const u32 keep = nvalid < k ? nvalid : k;
        for (u32 i = tid; i < keep; i += 256) list[i] = s[i];
        if (tid == 0) {
            p.cnt[q] = keep;
            if (p.cnt0) p.cnt0[q] = keep;
            p.tau[q] = (nvalid >= k) ? s[k - 1] : ~0ull;
        }
    }
    if (tid == 0 && overflow) { p.ovf[q] = 1u; atomicOr(p.ovf_any, 1u); }
}

// ------------------------------------------------------------------------------------------------
// finalize: keys -> (float32 distance, int64 index + offset)   (cu_knn.py:273 adds the tile offset the same way)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_finalize_kernel(const u64* __restrict__ buf, const u32* __restrict__ cnt, int cap, int k,
                                                            long long nq, long long index_offset, float* __restrict__ out_d,
                                                            long long* __restrict__ out_i, int negate)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * k) return;
    const long long q = i / k;
    const int j = (int)(i - q * k);
    if ((u32)j < cnt[q]) {
        const u64 key = buf[q * cap + j];
        const float v = unsortable_f32((u32)(key >> 32));
        out_d[i] = negate ? -v : v;                          // MaxSim ranks -score
        out_i[i] = (long long)(u32)key + index_offset;
    } else {
        out_d[i] = negate ? -__builtin_inff() : __builtin_inff();
        out_i[i] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// explicit pairs (nwknn_pair_distances): idx [nq][k] -> list slots the rescoring kernel recomputes, and back
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_pairs_to_keys_kernel(const long long* __restrict__ idx, long long nq, long long nq_pad, int k, int cap,
                                                                 long long index_offset, long long nb, u64* __restrict__ buf,
                                                                 u32* __restrict__ cnt, u32* __restrict__ cnt0)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < nq_pad) { cnt[i] = i < nq ? (u32)k : 0u; cnt0[i] = 0u; }
    if (i >= nq * k) return;
    const long long q = i / k;
    const int j = (int)(i - q * k);
    const long long row = idx[i] - index_offset;
    buf[q * cap + j] = (row >= 0 && row < nb) ? (u64)(u32)row : ~0ull;        // ~0: an empty slot, left alone by the rescoring kernel
}

__global__ __launch_bounds__(256) void nwk_keys_to_distances_kernel(const u64* __restrict__ buf, long long nq, int k, int cap, float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * k) return;
    const long long q = i / k;
    const u64 key = buf[q * cap + (int)(i - q * k)];
    out[i] = key == ~0ull ? __builtin_nanf("") : unsortable_f32((u32)(key >> 32));
}

// float64 dot products of given pairs (nwknn_pair_dots): one lane per pair, both prepared rows walked 16 B at a time.  The prepared fp32
// layout permutes k inside groups of 8 -- the same way on both sides, and a sum does not care.  The validator's arithmetic
// (parquet_to_format.py:368-373 works in float64), not the engine's canonical fp32 chain.
template <bool ROWS16>
__global__ __launch_bounds__(256) void nwk_pair_dots_kernel(const void* __restrict__ qp, const void* __restrict__ bp, const long long* __restrict__ idx,
                                                             long long nq, int k, int dp, long long index_offset, long long nb, double* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * k) return;
    const long long q = i / k;
    const long long row = idx[i] - index_offset;
    if (row < 0 || row >= nb) { out[i] = __builtin_nan(""); return; }
    double acc = 0.0;
    if (ROWS16) {
        const f16x8* a = (const f16x8*)((const _Float16*)qp + q * dp);
        const f16x8* b = (const f16x8*)((const _Float16*)bp + row * dp);
        for (int c = 0; c < dp / 8; ++c) {
            const f16x8 x = a[c], y = b[c];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = fma((double)(float)x[t], (double)(float)y[t], acc);
        }
    } else {
        const f32x4* a = (const f32x4*)((const float*)qp + q * dp);
        const f32x4* b = (const f32x4*)((const float*)bp + row * dp);
        for (int c = 0; c < dp / 4; ++c) {
            const f32x4 x = a[c], y = b[c];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = fma((double)x[t], (double)y[t], acc);
        }
    }
    out[i] = acc;
}

// ------------------------------------------------------------------------------------------------
// merge of n_lists partial top-k lists [n_lists][Q][k_in] by (distance, index)   (merge.py:73-101)
// ------------------------------------------------------------------------------------------------
// descending != 0: the values are SCORES, higher first (MaxSim): ranked through their negation, so ties still go to the lower index
__global__ __launch_bounds__(256) void nwk_merge_kernel(const float* __restrict__ dists, const long long* __restrict__ idx, int n_lists,
                                                         long long list_stride, int k_in, int k_out, float* __restrict__ out_d,
                                                         long long* __restrict__ out_i, int np2, int descending)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    u32* sd = (u32*)dyn;                            // sortable distance
    long long* si = (long long*)(dyn + (size_t)np2 * 4);      // np2 is a power of two >= 2: 8-B aligned
    const int tid = threadIdx.x;
    const long long q = blockIdx.x;
    const int w = n_lists * k_in;
    for (int i = tid; i < np2; i += 256) {
        if (i < w) {
            const int lst = i / k_in, j = i - lst * k_in;
            const long long at = (long long)lst * list_stride + q * k_in + j;        // list_stride: entries between two lists (nq * k_in when they are dense)
            const long long id = idx[at];
            sd[i] = id < 0 ? 0xFFFFFFFFu : sortable_u32(descending ? -dists[at] : dists[at]);
            si[i] = id < 0 ? 0x7FFFFFFFFFFFFFFFll : id;
        } else { sd[i] = 0xFFFFFFFFu; si[i] = 0x7FFFFFFFFFFFFFFFll; }
    }
    __syncthreads();
    for (int k2 = 2; k2 <= np2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u32 da = sd[i], db = sd[ixj];
                    const long long ia = si[i], ib = si[ixj];
                    const bool gt = (da > db) || (da == db && ia > ib);
                    const bool up = (i & k2) == 0;
                    if (gt == up) { sd[i] = db; sd[ixj] = da; si[i] = ib; si[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k_out; j += 256) {
        const bool ok = j < np2 && si[j] != 0x7FFFFFFFFFFFFFFFll;
        const float v = ok ? unsortable_f32(sd[j]) : __builtin_inff();
        out_d[q * k_out + j] = descending ? -v : v;
        out_i[q * k_out + j] = ok ? si[j] : -1;
    }
}


// ------------------------------------------------------------------------------------------------
// launchers (host side, called from nwknn_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C++" {

hipError_t nwk_launch_rescore(hipStream_t st, const NwkRescoreParams& p, int metric, long long nq_pad)
{
    if (nq_pad <= 0) return hipSuccess;
    const int waves = p.cap <= 1024 ? 4 : (p.cap <= 2048 ? 2 : 1);         // nq_pad is a multiple of 256
    const unsigned grid = (unsigned)(nq_pad / waves);
    const bool coop = p.coop != 0;
    // lists + staged query chunks + (cooperative fetch) two stages of 64 row segments per wave; the legacy form keeps 512 B of slack
    // behind the lists (predicated-off lanes may still address past `end`), the cooperative form has its stages there
    const size_t lds = (size_t)waves * p.cap * 8 + (size_t)waves * RS_QC * 4 + (coop ? (size_t)waves * 2 * 64 * RS_SEG_BYTES : 512);
#define NWK_RS(M)                                                                                                        \
    do {                                                                                                                  \
        if (p.rows16 && coop) hipLaunchKernelGGL((nwk_rescore_kernel<M, true, true>), dim3(grid), dim3(64 * waves), lds, st, p);        \
        else if (p.rows16) hipLaunchKernelGGL((nwk_rescore_kernel<M, true, false>), dim3(grid), dim3(64 * waves), lds, st, p);          \
        else if (coop) hipLaunchKernelGGL((nwk_rescore_kernel<M, false, true>), dim3(grid), dim3(64 * waves), lds, st, p);              \
        else hipLaunchKernelGGL((nwk_rescore_kernel<M, false, false>), dim3(grid), dim3(64 * waves), lds, st, p);                       \
    } while (0)
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) NWK_RS(NWKNN_METRIC_SQEUCLIDEAN);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) NWK_RS(NWKNN_METRIC_ONE_MINUS_DOT);
    else if (metric == NWKNN_METRIC_COSINE) NWK_RS(NWKNN_METRIC_COSINE);
    else return hipErrorInvalidValue;
#undef NWK_RS
    return hipGetLastError();
}

hipError_t nwk_launch_compact(hipStream_t st, const NwkCompactParams& p, long long n_queries)
{
    if (n_queries <= 0) return hipSuccess;
    const size_t lds = (size_t)p.cap * 16 + 16;
    hipLaunchKernelGGL(nwk_compact_kernel, dim3((unsigned)n_queries), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t nwk_launch_finalize(hipStream_t st, const u64* buf, const u32* cnt, int cap, int k, long long nq, long long index_offset,
                               float* out_d, long long* out_i, int negate)
{
    const long long total = nq * k;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nwk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf, cnt, cap, k, nq, index_offset,
                       out_d, out_i, negate);
    return hipGetLastError();
}

hipError_t nwk_launch_pairs_to_keys(hipStream_t st, const long long* idx, long long nq, long long nq_pad, int k, int cap, long long index_offset,
                                    long long nb, u64* buf, u32* cnt, u32* cnt0)
{
    const long long total = nq * k > nq_pad ? nq * k : nq_pad;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nwk_pairs_to_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, idx, nq, nq_pad, k, cap, index_offset, nb, buf, cnt, cnt0);
    return hipGetLastError();
}

hipError_t nwk_launch_pair_dots(hipStream_t st, int rows16, const void* qp, const void* bp, const long long* idx, long long nq, int k, int dp,
                                long long index_offset, long long nb, double* out)
{
    const long long total = nq * k;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (rows16) hipLaunchKernelGGL(nwk_pair_dots_kernel<true>, dim3(grid), dim3(256), 0, st, qp, bp, idx, nq, k, dp, index_offset, nb, out);
    else hipLaunchKernelGGL(nwk_pair_dots_kernel<false>, dim3(grid), dim3(256), 0, st, qp, bp, idx, nq, k, dp, index_offset, nb, out);
    return hipGetLastError();
}

hipError_t nwk_launch_keys_to_distances(hipStream_t st, const u64* buf, long long nq, int k, int cap, float* out)
{
    const long long total = nq * k;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nwk_keys_to_distances_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf, nq, k, cap, out);
    return hipGetLastError();
}

// lists l = 0..n_lists-1 start list_stride entries apart (dense [n_lists][nq][k_in]: list_stride = nq * k_in; a slice of the query rows of
// gathered lists: the stride of the whole lists)
hipError_t nwk_launch_merge(hipStream_t st, const float* dists, const long long* idx, int n_lists, long long list_stride, long long nq, int k_in,
                            int k_out, float* out_d, long long* out_i, int descending)
{
    if (nq <= 0) return hipSuccess;
    int np2 = 2;
    while (np2 < n_lists * k_in) np2 <<= 1;
    const size_t lds = (size_t)np2 * 12;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nwk_merge_kernel, dim3((unsigned)nq), dim3(256), lds, st, dists, idx, n_lists, list_stride, k_in, k_out, out_d, out_i, np2, descending);
    return hipGetLastError();
}

}  // extern "C++"
