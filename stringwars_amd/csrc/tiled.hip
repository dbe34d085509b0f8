// tiled.hip -- unit-cost Levenshtein without a planning pre-pass: every workgroup plans its own tile.
//
// The globally planned path (prepass.hip -> bitparallel.hip) spends three launches (histogram, scan, scatter), a
// permutation array and a round trip of the plan to the host before the first DP column is computed -- a third of a
// synchronous call on a 1 M-pair batch of tokens, nine tenths of it on 10 K words. Here a workgroup takes a TILE of
// consecutive pairs (<= 1024), and in LDS
//   B. classifies them: pattern side, G = blocks of 32 rows, text length n; finishes the trivial ones;
//   C. moves a class's few left-over pairs up into the next class when that saves a mostly empty work item;
//   D-F. counting-sorts the tile by (G, n bucket) into a u16 index list;
//   G. runs the work items of bp_item.hpp on it, longest first, waves taking items from an LDS ticket.
// Pairs of one item are consecutive in the sorted tile, so the lanes of a wave finish together just as with the global
// sort; what is lost is one partly filled item per class and tile. What is gained: one launch per call, no host
// round trip, no permutation / key arrays in HBM, and results that land in a 4 KB window per tile instead of being
// scattered over the whole output.
//
// The kernel is exact for every pair whose pattern fits 64 blocks (2048 symbols). A longer pair raises the
// `violation` flag (host-mapped memory) and is left alone: the host only picks this kernel when tape statistics or the
// previous call say no such pair exists, checks the flag after the call and falls back to the planned path if it is
// set (api.hip).
#include <cstdio>
#include <cstdlib>

#include "bp_item.hpp"

namespace swh {

// Pairs per tile: 2048, and 4096 for the sixteen-wave workgroup that has a compute unit to itself (kIndexBits of an index-list
// entry, then 6 + 6 bits: common prefix / suffix). ONE tile per workgroup is what the sizes are for: a second round of tiles cost
// C2 12 % (tiles of 652 pairs in three rounds against 977 in two: 25.2 against 28.7 TCUPS), one round of 1954 gave 29.6.
constexpr int tile_index_bits(int waves) { return waves >= 16 ? 12 : 11; }
constexpr int tile_max(int waves) { return 1 << tile_index_bits(waves); }
constexpr uint32_t kAffixCap = 16; // symbols cut off at either end: what one 16-byte window shows
constexpr int kTileBuckets = 32;   // text-length buckets per class
constexpr int kTileClasses = 64;
constexpr int kTileBins = kTileClasses * kTileBuckets;

// Waves per workgroup. The hardware favours a compute unit's oldest workgroups, so the youngest run the last part of
// their tile alone: with four-wave workgroups that is ONE wave per SIMD, and a wave of this serial recurrence issues at
// half rate on its own (C2: the four workgroups of a CU finish at 105 / 150 / 190 / 235 us of a 260 us profiled launch,
// tools/tile_spans.py). Eight-wave workgroups (two per CU, two tiles each) leave two waves per SIMD in that tail: -6 % for
// a launch on its own (C2 synchronous: 22.95 -> 24.1-24.6 TCUPS); overlapping launches of two pipeline lanes, where the
// other launch fills the tail anyway, measure the same with either (27.7-28.2 TCUPS). Round 4 took the step after that: ONE
// sixteen-wave workgroup per compute unit with one tile of up to 4096 pairs (160 KB of LDS: sixteen 8.25 KB tables, the tile's
// 16 KB index list, 8 KB of counters / staged distances) -- sixteen waves on one ticket finish together: 29.6 -> 31.1 TCUPS.
template <int kTileMax>
struct TileLds {
    uint32_t sorted[kTileMax];            // tile-local pair indices (| prefix << kIndexBits | suffix << (kIndexBits + 6)), sorted by (class, text-length bucket)
    uint32_t bins[kTileBins / 2 > kTileMax / 2 ? kTileBins / 2 : kTileMax / 2];   // two u16 counters per word: counts, then exclusive prefixes; later a tile's distances, 16 bits each
    uint32_t class_count[kTileClasses];   // pairs per natural class; after step C: per final class
    uint16_t class_thr[kTileClasses];     // ranks >= thr move up to class_tgt
    uint16_t class_tgt[kTileClasses];
    uint32_t item_prefix[kTileClasses + 1];
    uint32_t wave_sums[16];
    uint32_t ticket;
};
struct TileTail {   // after the last tile the counters' space serves the call summary
    unsigned long long cells, syms;
    uint32_t maxa, maxb, shorts, misfit;
    SummaryLds summary;
};
static_assert(sizeof(TileTail) <= sizeof(uint32_t) * kTileBins / 2, "the tail reuses TileLds::bins");

template <typename Sym, int kWaves> constexpr size_t tiled_lds_bytes() {
    return (size_t)kWaves * (bp_table_words<Sym>() + 64) * 4 + sizeof(TileLds<tile_max(kWaves)>);
}

#ifdef SWH_TILE_PROFILE
// Diagnostic build only (make EXTRA=-DSWH_TILE_PROFILE): wave cycles per phase of k_bitparallel_tiled, summed over waves.
__device__ unsigned long long g_tile_phase[8];
__device__ unsigned long long g_tile_span[2048][4];   // per workgroup: start, end of planning, end of items, end (100 MHz clock)
extern "C" void swh_debug_tile_spans(unsigned long long *out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_span), sizeof(unsigned long long) * 2048 * 4);
}
extern "C" void swh_debug_tile_phases(unsigned long long *out) {
    unsigned long long zero[8] = {};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_phase), sizeof(zero));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_phase), zero, sizeof(zero));
}
#define TILE_STAMP(slot) do { const unsigned long long now__ = __builtin_readcyclecounter(); phase_acc[slot] += now__ - phase_t; phase_t = now__; } while (0)
#else
#define TILE_STAMP(slot) do {} while (0)
#endif

struct TiledArgs {
    KernelArgs k;
    uint32_t tile;          // pairs per tile
    uint32_t tiles;
    uint32_t shift;         // text-length bucket = min(n >> shift, kTileBuckets - 1)
    uint32_t cut_affixes;   // byte strings: cut what a pair shares at both ends before classifying it
    PlanPartial *partials;  // per-workgroup work-unit sums (cells, symbols, maxima, "met a pair that does not fit")
    uint32_t *done_counter;
    CallSummary *summary;   // host-mapped: the last workgroup reports (common.hpp: report_call_summary)
};

template <typename Sym, int kWaves, bool kWide>
__device__ __forceinline__ void tiled_run(const TiledArgs &targs, char *smem, const uint64_t a_total, const uint64_t b_total) {
    constexpr int kThreads = kWaves * 64, kTableWords = bp_table_words<Sym>();
    // pairs per thread and tile, rounded UP: ten-wave workgroups (code points) have 640 threads, and 1024 / 640 = 1 left the
    // pairs 640 .. 1023 of a full tile unclassified (every use below is guarded by idx < count)
    constexpr int kTileMax = tile_max(kWaves), kTileIndexBits = tile_index_bits(kWaves);
    constexpr int kPer = (kTileMax + kThreads - 1) / kThreads;
    const KernelArgs &args = targs.k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    BpWave<Sym> wv;
    wv.init((uint32_t *)smem + (size_t)wave * kTableWords, (uint32_t *)smem + (size_t)kWaves * kTableWords + wave * 64, lane,
            a_total, b_total);
    TileLds<kTileMax> &tl = *(TileLds<kTileMax> *)(smem + (size_t)kWaves * (kTableWords + 64) * 4);
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, misfit = 0;
#ifdef SWH_TILE_PROFILE
    unsigned long long phase_acc[4] = {0, 0, 0, 0}, phase_t = __builtin_readcyclecounter(), items_done = 0;
    if (threadIdx.x == 0) g_tile_span[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif

    for (uint32_t tile = blockIdx.x; tile < targs.tiles; tile += gridDim.x) {
        const uint64_t base = (uint64_t)tile * targs.tile;
        const uint32_t count = (uint32_t)(args.job.pairs - base < targs.tile ? args.job.pairs - base : targs.tile);
        // ---- A: clear the counters ------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(3);   // planning is a chain of round trips and barriers: it goes first, the work items of other workgroups fill the gaps
        for (int i = threadIdx.x; i < kTileBins / 2; i += kThreads) tl.bins[i] = 0;
        if (threadIdx.x < kTileClasses) tl.class_count[threadIdx.x] = 0;
        if (threadIdx.x == 0) tl.ticket = 0;
        __syncthreads();
        // ---- B: classify my pairs ---------------------------------------------------------------------------------
        // Every load a thread needs here is requested in two batches (indices clamped into the batch, no branch between the
        // loads): the extents of its pairs, then -- byte strings -- the first and last sixteen bytes of both strings of every
        // pair. What a pair shares at both ends is cut off before it is classified (the first thing rapidfuzz's Levenshtein
        // does, too): up to kAffixCap symbols each, kept in the upper bits of the pair's entry in the tile's index list.
        uint32_t cls[kPer], txt[kPer], rank[kPer], affix[kPer];
        // (two pairs per thread at a time: a pair's four 16-byte windows are 16 registers while they are in flight)
#ifndef SWH_TILE_BATCH
#define SWH_TILE_BATCH 2
#endif
        constexpr int kBatch = kPer < SWH_TILE_BATCH ? kPer : SWH_TILE_BATCH;
        static_assert(kPer % kBatch == 0, "pairs per thread come in whole batches");
#pragma unroll
        for (int kb = 0; kb < kPer; kb += kBatch) {
        uint64_t a0s[kBatch], b0s[kBatch];
        uint32_t las[kBatch], lbs[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int k = kb + j;
            const uint64_t p = base + (uint32_t)k * kThreads + threadIdx.x;
            const uint64_t pc = p < args.job.pairs ? p : args.job.pairs - 1;
            if (args.off64) pair_extent<uint64_t>(args.job, pc, a0s[j], las[j], b0s[j], lbs[j]);
            else pair_extent<uint32_t>(args.job, pc, a0s[j], las[j], b0s[j], lbs[j]);
            affix[k] = 0;
        }
        if constexpr (sizeof(Sym) == 1 && kWide) {
            if (targs.cut_affixes) {
                uint4 ha[kBatch], hb[kBatch], ta[kBatch], tb[kBatch];
                bool inside[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    // windows that would leave the tapes (the first / last strings) are read at the tapes' start and not used
                    const uint64_t ea = a0s[j] + las[j], eb = b0s[j] + lbs[j];
                    inside[j] = a0s[j] + 16 <= a_total && b0s[j] + 16 <= b_total && ea >= 16 && eb >= 16;
                    const uint8_t *ad = (const uint8_t *)args.job.a.data, *bd = (const uint8_t *)args.job.b.data;
                    __builtin_memcpy(&ha[j], ad + (inside[j] ? a0s[j] : 0), 16);
                    __builtin_memcpy(&hb[j], bd + (inside[j] ? b0s[j] : 0), 16);
                    __builtin_memcpy(&ta[j], ad + (inside[j] ? ea - 16 : 0), 16);
                    __builtin_memcpy(&tb[j], bd + (inside[j] ? eb - 16 : 0), 16);
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const uint32_t mn = las[j] < lbs[j] ? las[j] : lbs[j];
                    const unsigned long long h_lo = (unsigned long long)(ha[j].x ^ hb[j].x) | ((unsigned long long)(ha[j].y ^ hb[j].y) << 32);
                    const unsigned long long h_hi = (unsigned long long)(ha[j].z ^ hb[j].z) | ((unsigned long long)(ha[j].w ^ hb[j].w) << 32);
                    const unsigned long long t_lo = (unsigned long long)(ta[j].x ^ tb[j].x) | ((unsigned long long)(ta[j].y ^ tb[j].y) << 32);
                    const unsigned long long t_hi = (unsigned long long)(ta[j].z ^ tb[j].z) | ((unsigned long long)(ta[j].w ^ tb[j].w) << 32);
                    uint32_t pre = h_lo ? (uint32_t)__builtin_ctzll(h_lo) >> 3 : (h_hi ? 8u + ((uint32_t)__builtin_ctzll(h_hi) >> 3) : kAffixCap);
                    pre = pre < mn ? pre : mn;
                    uint32_t suf = t_hi ? (uint32_t)__builtin_clzll(t_hi) >> 3 : (t_lo ? 8u + ((uint32_t)__builtin_clzll(t_lo) >> 3) : kAffixCap);
                    suf = suf < mn - pre ? suf : mn - pre;
                    affix[kb + j] = inside[j] ? pre | (suf << 6) : 0u;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int k = kb + j;
            const uint32_t idx = (uint32_t)k * kThreads + threadIdx.x;
            cls[k] = 0xFFu; txt[k] = 0; rank[k] = 0;
            if (idx < count) {
                const uint64_t p = base + idx;
                uint32_t la = las[j], lb = lbs[j];
                cells += (unsigned long long)la * lb;
                syms += (unsigned long long)la + lb;
                maxa = la > maxa ? la : maxa;
                maxb = lb > maxb ? lb : maxb;
                shorts += (la <= 32 && lb <= 32) ? 1u : 0u;
                const uint32_t diff = la > lb ? la - lb : lb - la;
                const uint32_t shared = (affix[k] & 63u) + (affix[k] >> 6);
                la -= shared; lb -= shared;
                if (la == 0 || lb == 0) {
                    store_result(args.job, p, (int64_t)clamp_bound(la + lb, args.job.bound));
                } else if (args.job.bound != 0xFFFFFFFFu && diff > args.job.bound) {
                    store_result(args.job, p, (int64_t)args.job.bound + 1);
                } else {
                    const bool pattern_is_a = bp_pattern_is_a(la, lb);
                    const uint32_t m = pattern_is_a ? la : lb, n = pattern_is_a ? lb : la;
                    const uint32_t g = (m + 31) >> 5;
                    if (g > (uint32_t)kTileClasses) {
                        misfit = 1;   // both strings beyond 2048 symbols: not for this kernel (the host redoes the call)
                    } else {
                        cls[k] = g - 1;
                        txt[k] = n;
                        rank[k] = atomicAdd(&tl.class_count[g - 1], 1u);
                    }
                }
            }
        }
        }
        __syncthreads();
        // ---- C: left-overs move up; work items per class (wave 0) -------------------------------------------------
        if (wave == 0) {
            const uint32_t mine = tl.class_count[lane];
            tl.class_thr[lane] = (uint16_t)mine;
            tl.class_tgt[lane] = (uint16_t)lane;
            unsigned long long present = __ballot(mine != 0);
            wave_lds_fence();
            if (lane == 0) {
                uint32_t incoming = 0;
                while (present) {
                    const int c = __builtin_ctzll(present);
                    present &= present - 1;
                    const uint32_t native = tl.class_count[c];
                    uint32_t total = native + incoming;
                    incoming = 0;
                    const uint32_t per = 64u / (uint32_t)(c + 1), rest = total % per;
                    if (rest && present) {
                        const int up = __builtin_ctzll(present);
                        const uint32_t per_up = 64u / (uint32_t)(up + 1);
                        // at most half an item of the class above, instead of a mostly idle item of this one
                        if (rest <= native && 2 * rest <= per_up) {
                            tl.class_thr[c] = (uint16_t)(native - rest);
                            tl.class_tgt[c] = (uint16_t)up;
                            incoming = rest;
                            total -= rest;
                        }
                    }
                    tl.class_count[c] = total;
                }
            }
            wave_lds_fence();
            const uint32_t final_count = tl.class_count[lane], per = 64u / (uint32_t)(lane + 1);
            const uint32_t items = (final_count + per - 1) / per;
            const uint32_t incl = wave_inclusive_sum_u32(items);
            tl.item_prefix[lane] = incl - items;
            if (lane == 63) tl.item_prefix[64] = incl;
        }
        __syncthreads();
        // ---- D: final keys, ranks inside a key ---------------------------------------------------------------------
        uint32_t key[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            key[k] = 0xFFFFFFFFu;
            if (cls[k] != 0xFFu) {
                uint32_t c = cls[k];
                if (rank[k] >= tl.class_thr[c]) c = tl.class_tgt[c];
                const uint32_t bucket = txt[k] >> targs.shift;
                key[k] = c * kTileBuckets + (bucket < (uint32_t)kTileBuckets - 1 ? bucket : (uint32_t)kTileBuckets - 1);
                const uint32_t old = atomicAdd(&tl.bins[key[k] >> 1], (key[k] & 1u) ? 0x10000u : 1u);
                rank[k] = (key[k] & 1u) ? old >> 16 : old & 0xFFFFu;
            }
        }
        __syncthreads();
        // ---- E: exclusive scan of the key counters (two u16 per word, in place) -------------------------------------
        {
            // consecutive words per thread, rounded UP (640 threads: 1024 / 640 = 1 left the keys of classes >= 40 -- patterns of more
            // than 1280 symbols -- out of the scan: wrong distances for code-point strings that long, found by the soak test)
            constexpr int kWords = (kTileBins / 2 + kThreads - 1) / kThreads;
            uint32_t words[kWords], sum = 0;
#pragma unroll
            for (int q = 0; q < kWords; ++q) {
                const int at = threadIdx.x * kWords + q;
                words[q] = at < kTileBins / 2 ? tl.bins[at] : 0u;
                sum += (words[q] & 0xFFFFu) + (words[q] >> 16);
            }
            const uint32_t incl = wave_inclusive_sum_u32(sum);
            if (lane == 63) tl.wave_sums[wave] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (int w = 0; w < wave; ++w) run += tl.wave_sums[w];
#pragma unroll
            for (int q = 0; q < kWords; ++q) {
                const uint32_t lo = words[q] & 0xFFFFu, hi = words[q] >> 16;
                const int at = threadIdx.x * kWords + q;
                if (at < kTileBins / 2) tl.bins[at] = run | ((run + lo) << 16);
                run += lo + hi;
            }
        }
        __syncthreads();
        // ---- F: scatter the tile-local indices ------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (key[k] != 0xFFFFFFFFu) {
                const uint32_t word = tl.bins[key[k] >> 1];
                const uint32_t start = (key[k] & 1u) ? word >> 16 : word & 0xFFFFu;
                tl.sorted[start + rank[k]] = ((uint32_t)k * kThreads + threadIdx.x) | (affix[k] << kTileIndexBits);   // index | prefix << 11 | suffix << 17
            }
        }
        __syncthreads();
        TILE_STAMP(0);   // planning the tile
#ifdef SWH_TILE_PROFILE
        if (threadIdx.x == 0) g_tile_span[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
        // The key counters have done their work: keep the 64 class starts (class_thr is free since step D), then the
        // 4 KB they occupied stage the tile's distances -- one coalesced sweep at the end instead of 4-byte stores
        // scattered over the tile's window (which reached HBM as one 32-byte sector each: 6x the result bytes).
        if (threadIdx.x < kTileClasses) tl.class_thr[threadIdx.x] = (uint16_t)(tl.bins[(threadIdx.x * kTileBuckets) >> 1] & 0xFFFFu);
        __syncthreads();
        static_assert(sizeof(tl.bins) >= (size_t)kTileMax * 2, "the key counters' space holds a tile's distances as 16-bit slots");
        uint16_t *const staged = (uint16_t *)tl.bins;
        for (int i = threadIdx.x; i < kTileMax / 2; i += kThreads) tl.bins[i] = 0xFFFFFFFFu;   // "no distance here" (trivial pairs are stored directly)
        __syncthreads();
        // ---- G: work items, heaviest first (high class, long text), dealt by an LDS ticket --------------------------------
        __builtin_amdgcn_s_setprio(0);
        {
            const uint32_t items_total = tl.item_prefix[64];
            const uint32_t my_prefix = tl.item_prefix[lane];
            for (;;) {
                uint32_t t = 0;
                if (lane == 0) t = __hip_atomic_fetch_add(&tl.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t >= items_total) break;
                const uint32_t item = items_total - 1 - t;
                const uint32_t G = (uint32_t)__popcll(__ballot(my_prefix <= item));   // class = prefix entries <= item
                const uint32_t per = 64 / G;
                const uint32_t chunk = item - tl.item_prefix[G - 1];
                const uint32_t cstart = tl.class_thr[G - 1], ccount = tl.class_count[G - 1];
                const uint32_t slot = (uint32_t)lane / G;
                const uint32_t pidx = chunk * per + slot;
                const bool have = slot < per && pidx < ccount;
                uint64_t p = 0, a0 = 0, b0 = 0;
                uint32_t la = 0, lb = 0;
                if (have) {
                    const uint32_t entry = tl.sorted[cstart + pidx];
                    p = base + (entry & (uint32_t)(kTileMax - 1));
                    if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
                    else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
                    const uint32_t pre = (entry >> kTileIndexBits) & 63u, both = pre + (entry >> (kTileIndexBits + 6));   // what the pair shares at both ends (step B)
                    a0 += pre; b0 += pre; la -= both; lb -= both;
                }
                bp_item<Sym, kWide>(args, wv, G, have, p, a0, la, b0, lb, staged, base);
#ifdef SWH_TILE_PROFILE
                ++items_done;
#endif
            }
        }
        TILE_STAMP(1);   // work items
        __syncthreads();
        TILE_STAMP(2);   // waiting for the workgroup's other waves
#ifdef SWH_TILE_PROFILE
        if (threadIdx.x == 0) g_tile_span[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
#endif
        for (uint32_t i = threadIdx.x; i < count; i += kThreads) {
            const uint32_t d = staged[i];
            if (d != 0xFFFFu) store_result(args.job, base + i, (int64_t)d);
        }
        __syncthreads();   // the next tile rewrites the lists
    }
#ifdef SWH_TILE_PROFILE
    if (threadIdx.x == 0) g_tile_span[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        for (int q = 0; q < 3; ++q) atomicAdd(&g_tile_phase[q], phase_acc[q]);
        atomicAdd(&g_tile_phase[3], 1ull);
        atomicAdd(&g_tile_phase[4], items_done);
    }
#endif
    {
        TileTail &tail = *(TileTail *)tl.bins;   // the tile loop ended with a barrier
        unsigned long long &lcells = tail.cells, &lsyms = tail.syms;
        uint32_t &lmaxa = tail.maxa, &lmaxb = tail.maxb, &lshorts = tail.shorts, &lmisfit = tail.misfit;
        if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lmisfit = 0; }
        __syncthreads();
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            cells += __shfl_xor(cells, off);
            syms += __shfl_xor(syms, off);
            shorts += __shfl_xor(shorts, off);
            misfit |= __shfl_xor(misfit, off);
            const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
            maxa = oa > maxa ? oa : maxa;
            maxb = ob > maxb ? ob : maxb;
        }
        if (lane == 0) {
            atomicAdd(&lcells, cells);
            atomicAdd(&lsyms, syms);
            atomicMax(&lmaxa, maxa);
            atomicMax(&lmaxb, maxb);
            atomicAdd(&lshorts, shorts);
            atomicOr(&lmisfit, misfit);
        }
        __syncthreads();
        report_call_summary(PlanPartial{lcells, lsyms, lmaxa, lmaxb, lshorts, lmisfit}, targs.partials, targs.done_counter, targs.summary, tail.summary);
    }
}

template <typename Sym, int kWaves>
__global__ __launch_bounds__(kWaves * 64, BpTraits<Sym>::kMinWavesPerSimd) void k_bitparallel_tiled(TiledArgs targs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const KernelArgs &args = targs.k;
    const uint64_t a_total = tape_total(args.job.a, args.off64);
    const uint64_t b_total = tape_total(args.job.b, args.off64);
    if constexpr (sizeof(Sym) == 1) {
        if (a_total >= 16 && b_total >= 16) tiled_run<Sym, kWaves, true>(targs, smem, a_total, b_total);
        else tiled_run<Sym, kWaves, false>(targs, smem, a_total, b_total);
    } else {
        tiled_run<Sym, kWaves, false>(targs, smem, a_total, b_total);
    }
}

// Tile size: every workgroup slot should get the same number of tiles (1, 2, ...), tiles stay <= kTileMax pairs, and a
// batch too small to give every slot a useful tile uses fewer workgroups instead of smaller tiles.
TilePlan plan_tiles(uint64_t pairs, uint32_t slots, uint32_t longest_text, uint32_t kTileMax) {
    TilePlan tp{};
    constexpr uint32_t kTileMin = 256;
    const uint64_t rounds = (pairs + (uint64_t)slots * kTileMax - 1) / ((uint64_t)slots * kTileMax);
    uint64_t tile = (pairs + slots * rounds - 1) / (slots * rounds);
    static const uint32_t forced = [] { const char *e = test_hook("STRINGWARS_AMD_TILE"); return e ? (uint32_t)atoi(e) : 0u; }();   // tuning knob
    if (forced) tile = forced;
    if (tile < kTileMin) tile = kTileMin;
    if (tile > (uint64_t)kTileMax) tile = kTileMax;
    tp.tile = (uint32_t)tile;
    tp.tiles = (uint32_t)((pairs + tile - 1) / tile);
    tp.blocks = tp.tiles < slots ? tp.tiles : slots;
    uint32_t shift = 0;
    while (((uint64_t)longest_text >> shift) >= (uint64_t)kTileBuckets) ++shift;
    tp.shift = shift;
    return tp;
}

template <typename Sym, int kWaves>
static void launch_tiled_sym(Scope *scope, const KernelArgs &args, uint64_t pairs, uint32_t longest_text) {
    constexpr size_t lds = tiled_lds_bytes<Sym, kWaves>();
    uint32_t slots = (uint32_t)scope->compute_units * (uint32_t)((160 * 1024) / lds);
    if (slots > (uint32_t)kMaxPartials) slots = kMaxPartials;
    const TilePlan tp = plan_tiles(pairs, slots, longest_text, (uint32_t)tile_max(kWaves));
    TiledArgs t{};
    t.k = args;
    t.k.boundary = nullptr;
    t.tile = tp.tile; t.tiles = tp.tiles; t.shift = tp.shift;
    static const bool no_affix = [] { const char *e = test_hook("STRINGWARS_AMD_AFFIX"); return e && atoi(e) == 0; }();   // comparison knob
    t.cut_affixes = sizeof(Sym) == 1 && !no_affix ? 1u : 0u;
    t.partials = scope->plan_partials;
    t.done_counter = scope->done_counter;
    t.summary = scope->summary_target();
    opt_in_dynamic_lds(scope, (const void *)k_bitparallel_tiled<Sym, kWaves>, lds);
    static const bool debug = test_hook("STRINGWARS_AMD_DEBUG") != nullptr;
    if (debug) {
        int per_cu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bitparallel_tiled<Sym, kWaves>, kWaves * 64, lds);
        fprintf(stderr, "[swh] tiled: pairs %llu tile %u tiles %u blocks %u shift %u lds %zu -> %d workgroups per CU\n",
                (unsigned long long)pairs, tp.tile, tp.tiles, tp.blocks, tp.shift, lds, per_cu);
    }
    StampGuard guard(scope, sizeof(Sym) == 1 ? "bitparallel_tiled" : "bitparallel_tiled_u32");
    hipLaunchKernelGGL((k_bitparallel_tiled<Sym, kWaves>), dim3(tp.blocks), dim3(kWaves * 64), lds, scope->stream, t);
    SWH_HIP_CHECK(hipGetLastError());
}

void launch_bitparallel_tiled(Scope *scope, const KernelArgs &args, uint64_t pairs, uint32_t longest_text) {
    static const int forced = [] { const char *e = test_hook("STRINGWARS_AMD_TILED_WAVES"); return e ? atoi(e) : 0; }();   // comparison knob: 4; code points also 8
    if (args.sym_bytes == 4) {   // code points: 14.25 KB of tables per wave -- one workgroup of ten waves fills a CU next to the tile's lists
                                 // (latency-bound like the planned kernel: 8 -> 10 waves is 8 % on token-sized strings; STRINGWARS_AMD_TILED_WAVES=8 / =4:
                                 // eight waves; two-wave workgroups, five per CU)
        if (forced == 4) launch_tiled_sym<uint32_t, BpTraits<uint32_t>::kWaves>(scope, args, pairs, longest_text);
        else if (forced == 8) launch_tiled_sym<uint32_t, 8>(scope, args, pairs, longest_text);
        else launch_tiled_sym<uint32_t, 10>(scope, args, pairs, longest_text);
    }
    else if (forced == 4) launch_tiled_sym<uint8_t, 4>(scope, args, pairs, longest_text);
    else if (forced == 8) launch_tiled_sym<uint8_t, 8>(scope, args, pairs, longest_text);
    // a batch that fills the device: ONE sixteen-wave workgroup per compute unit, one tile of up to 4096 pairs each. Two eight-wave
    // workgroups share a CU unevenly -- the arbiter favours the older one, it finishes at ~half time and the younger runs the rest
    // of its tile at two waves per SIMD (tools/tile_spans.py) --; sixteen waves on one ticket finish together.
    // (Not in asynchronous scopes: two pipelined launches share the device best as eight-wave workgroups side by side on a CU -- 32.8
    // TCUPS over pipelined steps against 31.2 -- where a 160 KB workgroup keeps the other launch off its CU.)
    else if (forced == 16 || (!forced && !scope->async && pairs >= (uint64_t)scope->compute_units * 2048)) launch_tiled_sym<uint8_t, 16>(scope, args, pairs, longest_text);
    else launch_tiled_sym<uint8_t, 8>(scope, args, pairs, longest_text);
}

}  // namespace swh
