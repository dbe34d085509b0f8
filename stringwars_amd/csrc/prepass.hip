// prepass.hip -- device-side planning for one engine call.
//
//   1. k_plan_hist    : per pair, derive its kernel class + length bucket (a "key"), finish the
//                       trivial pairs (an empty side / cutoff decided by |la-lb|), histogram the keys,
//                       accumulate the reference's work units (cells = len(a)*len(b), bench.rs:413).
//   2. k_plan_scan    : exclusive scan of the key histogram -> write cursors + per-class ranges.
//   3. k_plan_scatter : counting-sort pair ids by key into `perm` (block-local ranks in LDS, one
//                       global atomic per (block, key)).
//
// Sorting by (class, length bucket) is what lets a wave64 hold pairs of one shape: equal block
// count for the bit-parallel kernel, equal columns-per-lane for the wavefront kernel, and nearly
// equal row counts so the lanes of a wave finish together ("load imbalance").
//
// Also here: UTF-8 -> code point staging (k_utf8_count / k_utf8_write) and a small device scan.
#include "common.hpp"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include "bp_window.hpp"

namespace swh {

__device__ __forceinline__ uint32_t ceil_log2_u32(uint32_t x) { return x <= 1 ? 0 : 32 - __clz(x - 1); }

// Key = class * kBuckets + bucket. Must match the decode logic in the kernels (they re-derive
// orientation from the lengths: columns = shorter string when `symmetric`).
__device__ __forceinline__ uint32_t plan_key(uint32_t la, uint32_t lb, uint32_t mode, uint32_t symmetric,
                                             uint32_t sym_bytes, uint32_t banded, uint32_t bound) {
    uint32_t m = la < lb ? la : lb, n = la < lb ? lb : la;
    if (banded) {
        // Banded window: ~(3 (k+1) + 45) wave instructions per column of 64 pairs, against 28 per step of
        // floor(64/G) pairs for the full bit-parallel kernel (bytes) or the u32 wavefront (code points).
        uint32_t g = (m + 31) >> 5;
        bool use = band_cost(bound) < (sym_bytes == 4 ? 40u : 28u) * g;
        if (use) {
            uint32_t bucket = m >> 4;  // text = shorter string = columns walked
            return kClassBanded * kBuckets + (bucket > 63 ? 63 : bucket);
        }
    }
    if (mode == kPlanBitParallel) {
        const bool pattern_is_a = bp_pattern_is_a(la, lb);
        const uint32_t pat_len = pattern_is_a ? la : lb, txt_len = pattern_is_a ? lb : la;
        uint32_t g = (pat_len + 31) >> 5;
        if (g <= 64) {
            uint32_t shift = 2 + ceil_log2_u32(g);
            uint32_t bucket = txt_len >> shift;
            return (kClassBp0 + g - 1) * kBuckets + (bucket > 63 ? 63 : bucket);
        }
        // more than 64 blocks: several passes of 64 blocks over the text (k_bitparallel_long), which draws its pairs from the END of the
        // class: the order is by what a pair costs its wave -- passes x text steps --, heaviest last
        uint64_t bucket = ((uint64_t)((g + 63) >> 6) * txt_len) >> 9;   // (64 bits: passes x text length wraps a u32 from ~16 M x 16 M symbols)
        return kClassBpLong * kBuckets + (bucket > 63 ? 63u : (uint32_t)bucket);
    }
    uint32_t cols = symmetric ? m : lb, rows = symmetric ? n : la;
    uint32_t cls, bucket;
    if (cols <= 128) {
        cls = kClassWf16 + ((cols + 15) >> 4) - 1;
        bucket = rows >> 2;
    } else {
        cls = kClassWfMulti;
        for (int i = 0; i < kNumWideW; ++i)
            if (cols <= 64u * wide_w(i)) { cls = kClassWf64 + i; break; }
        bucket = rows >> 7;
    }
    return cls * kBuckets + (bucket > 63 ? 63 : bucket);
}

struct PairInfo { uint32_t la, lb; bool trivial; int64_t trivial_value; uint64_t a0, b0; };

// Pairs of two short byte strings (both <= 32 symbols) are scored by k_direct_short in tape order, one pair per
// lane; the planning passes then only have to file them under class 0 ("done").
constexpr uint32_t kDirectMax = 32;
__device__ __forceinline__ bool short_pair(const PairInfo &info) {
    return !info.trivial && info.la <= kDirectMax && info.lb <= kDirectMax;
}

// The part of pair_info that needs only the two lengths (already in `info`).
__device__ __forceinline__ void classify_trivial(const PrepassArgs &args, PairInfo &info, int gap_open, int gap_extend,
                                                 bool levenshtein_unit);

template <typename Off>
__device__ __forceinline__ PairInfo pair_info(const PrepassArgs &args, uint64_t p, int gap_open, int gap_extend,
                                              bool levenshtein_unit) {
    PairInfo info;
    pair_extent<Off>(args.job, p, info.a0, info.la, info.b0, info.lb);
    classify_trivial(args, info, gap_open, gap_extend, levenshtein_unit);
    return info;
}

__device__ __forceinline__ void classify_trivial(const PrepassArgs &args, PairInfo &info, int gap_open, int gap_extend,
                                                 bool levenshtein_unit) {
    info.trivial = false;
    info.trivial_value = 0;
    uint32_t la = info.la, lb = info.lb;
    if (la == 0 || lb == 0) {
        uint32_t len = la + lb;
        info.trivial = true;
        // gap(k) = open + (k-1)*extend; a pair of empty strings scores 0.
        info.trivial_value = (len && !args.local) ? (int64_t)gap_open + (int64_t)(len - 1) * gap_extend : 0;
    } else if (levenshtein_unit && args.job.bound != 0xFFFFFFFFu) {
        uint32_t diff = la > lb ? la - lb : lb - la;
        if (diff > args.job.bound) { info.trivial = true; info.trivial_value = -(int64_t)(args.job.bound + 1); }
    }
}

// Sixteen waves share one 24.6 KB counter array, so a CU holds enough waves to cover the serial key computation
// (~120 dependent instructions per pair) and the per-block clear / flush of 6144 counters is spread over four times
// the threads (with 256-thread blocks: 2 waves per SIMD, 14 us per 1 M pairs).
constexpr int kPlanThreads = 1024;
template <typename Off>
__global__ __launch_bounds__(kPlanThreads) void k_plan_hist(PrepassArgs args) {
    // The planning kernels of one pipeline lane run in the shadow of the other lane's bit-parallel kernel, which owns
    // most wave slots and LDS; issue priority lets these short kernels through instead of trickling behind it.
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t lhist[kKeys];
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts;
    // k_direct_short has already finished every pair (and summed the work units): nothing to plan
    if (args.direct_short && *args.leftover == 0) return;
    for (int i = threadIdx.x; i < kKeys; i += blockDim.x) lhist[i] = 0;
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; }
    __syncthreads();
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < args.job.pairs; p += stride) {
        PairInfo info = pair_info<Off>(args, p, args.gap_open, args.gap_extend, args.unit_costs != 0);
        const bool is_short = short_pair(info);
        shorts += is_short ? 1u : 0u;
        cells += (unsigned long long)info.la * info.lb;
        syms += (unsigned long long)info.la + info.lb;
        maxa = info.la > maxa ? info.la : maxa;
        maxb = info.lb > maxb ? info.lb : maxb;
        uint32_t key;
        if (args.redo_filter && (uint64_t)load_result(args.job, p) <= (uint64_t)args.redo_done_upto) {
            key = kClassTrivial * kBuckets;   // settled by the call's first stage (the one-word band)
        } else if (info.trivial) {
            key = kClassTrivial * kBuckets;
            int64_t v = info.trivial_value;
            // distances are stored positive: the max-plus core negates, the trivial path mirrors it
            // the cutoff applies to every distance, whatever the costs (same as store_score in wavefront.hip)
            if (args.job.negate) v = (int64_t)clamp_bound((uint32_t)(-v), args.job.bound);
            store_result(args.job, p, v);
        } else if ((is_short && args.direct_short) || (info.la <= args.skip_upto && info.lb <= args.skip_upto)) {
            key = kClassTrivial * kBuckets;   // scored by k_direct_short (or, skip_upto, by the lane-per-pair alignment kernel)
        } else {
            key = plan_key(info.la, info.lb, args.mode, args.symmetric, args.sym_bytes, args.banded, args.job.bound);
            if (args.stage1 && key / kBuckets != (uint32_t)kClassBanded) {   // not the band's: left to the second stage
                store_result(args.job, p, (int64_t)args.job.bound + 1);
                key = kClassTrivial * kBuckets;
            }
        }
        args.keys[p] = (uint16_t)key;   // k_plan_scatter sorts by the stored keys: no second look at the offsets
        atomicAdd(&lhist[key], 1u);
    }
    // wave-reduce first: 256 lanes adding to ONE LDS word serialise completely (five such atomics per thread were
    // half of this kernel's run time)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cells += __shfl_xor(cells, off);
        syms += __shfl_xor(syms, off);
        shorts += __shfl_xor(shorts, off);
        const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
        maxa = oa > maxa ? oa : maxa;
        maxb = ob > maxb ? ob : maxb;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&lcells, cells);
        atomicAdd(&lsyms, syms);
        atomicMax(&lmaxa, maxa);
        atomicMax(&lmaxb, maxb);
        atomicAdd(&lshorts, shorts);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kKeys; i += blockDim.x)
        if (lhist[i]) atomicAdd(&args.hist[i], lhist[i]);
    if (threadIdx.x == 0) {
        // per-block partial sums; k_plan_scan folds them (same-address global atomics from every block
        // serialise in L2 and used to cost more than the whole histogram). When k_direct_short ran, the sums are its.
        PlanPartial part{lcells, lsyms, lmaxa, lmaxb, lshorts, 0};
        if (args.direct_short) part = PlanPartial{0, 0, 0, 0, 0, 0};
        args.partials[blockIdx.x] = part;
    }
}

// One block of 1024 threads; kKeys = 6144 -> 6 keys per thread. Also folds the per-block partial
// work-unit sums into the plan and re-zeroes the histogram for the next call (no memsets per call).
__global__ __launch_bounds__(1024) void k_plan_scan(uint32_t *hist, uint32_t *cursor, Plan *plan,
                                                    PlanPartial *partials, uint32_t npartials, uint32_t ndirect,
                                                    uint32_t *leftover) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t partial[1024];
    __shared__ unsigned long long rcells[1024], rsyms[1024];
    __shared__ uint32_t rmaxa[1024], rmaxb[1024], rshort[1024];
    constexpr int kPer = (kKeys + 1023) / 1024;
    uint32_t local[kPer];
    uint32_t sum = 0;
    for (int k = 0; k < kPer; ++k) {
        int i = threadIdx.x * kPer + k;
        local[k] = i < kKeys ? hist[i] : 0;
        if (i < kKeys) hist[i] = 0;
        sum += local[k];
    }
    partial[threadIdx.x] = sum;
    unsigned long long c = 0, sy = 0;
    uint32_t ma = 0, mb = 0, sh = 0;
    // k_plan_hist's sums (zero when it returned early: its slots are cleared below for that case) and k_direct_short's
    const bool planned = ndirect == 0 || *leftover != 0;
    for (uint32_t i = threadIdx.x; i < npartials + ndirect; i += 1024) {
        const bool from_hist = i < npartials;
        if (from_hist && !planned) continue;
        PlanPartial pp = partials[from_hist ? i : kMaxPartials + (i - npartials)];
        c += pp.cells; sy += pp.symbols; sh += pp.short_pairs;
        ma = pp.max_la > ma ? pp.max_la : ma;
        mb = pp.max_lb > mb ? pp.max_lb : mb;
    }
    rcells[threadIdx.x] = c; rsyms[threadIdx.x] = sy; rmaxa[threadIdx.x] = ma; rmaxb[threadIdx.x] = mb; rshort[threadIdx.x] = sh;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            rcells[threadIdx.x] += rcells[threadIdx.x + off];
            rsyms[threadIdx.x] += rsyms[threadIdx.x + off];
            rshort[threadIdx.x] += rshort[threadIdx.x + off];
            rmaxa[threadIdx.x] = rmaxa[threadIdx.x + off] > rmaxa[threadIdx.x] ? rmaxa[threadIdx.x + off] : rmaxa[threadIdx.x];
            rmaxb[threadIdx.x] = rmaxb[threadIdx.x + off] > rmaxb[threadIdx.x] ? rmaxb[threadIdx.x + off] : rmaxb[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        plan->cells = rcells[0]; plan->symbols = rsyms[0]; plan->max_la = rmaxa[0]; plan->max_lb = rmaxb[0];
        plan->invalid_utf8 = 0; plan->short_pairs = rshort[0];
        plan->fused_failed = 0;
        *leftover = 0;   // every reader of this call's value is upstream of this kernel or has read it above
    }
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (int)threadIdx.x >= off ? partial[threadIdx.x - off] : 0;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t base = partial[threadIdx.x] - sum;
    for (int k = 0; k < kPer; ++k) {
        int i = threadIdx.x * kPer + k;
        if (i < kKeys) {
            cursor[i] = base;
            if (i % kBuckets == 0) plan->class_start[i / kBuckets] = base;
            base += local[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 1023) plan->class_start[kMaxClasses] = partial[1023];
    __syncthreads();
    if (threadIdx.x < kMaxClasses) {
        uint32_t s0 = plan->class_start[threadIdx.x], s1 = plan->class_start[threadIdx.x + 1];
        plan->class_count[threadIdx.x] = s1 - s0;
    }
}

constexpr int kScatterTile = 2048;  // pairs per block iteration (2 per thread)
template <typename Off>
__global__ __launch_bounds__(kPlanThreads) void k_plan_scatter(PrepassArgs args) {
    __builtin_amdgcn_s_setprio(3);
    // One counter array, reused as the keys' output bases: 24.6 KB, so that a block of this kernel still fits next to
    // the four resident bit-parallel blocks of the scope's other pipeline lane (136 KB of a CU's 160 KB).
    __shared__ uint32_t lcount[kKeys];
    constexpr int kPerThread = kScatterTile / kPlanThreads;
    // nothing left to sort when every pair was finished by the planning pass or the direct kernel
    if (args.plan->class_count[kClassTrivial] == args.plan->class_start[kMaxClasses]) return;
    uint64_t tiles = (args.job.pairs + kScatterTile - 1) / kScatterTile;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < kKeys; i += blockDim.x) lcount[i] = 0;
        __syncthreads();
        uint32_t keys[kPerThread], ranks[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            uint64_t p = tile * kScatterTile + (uint64_t)k * kPlanThreads + threadIdx.x;
            keys[k] = 0xFFFFFFFFu;
            if (p < args.job.pairs) {
                const uint32_t key = args.keys[p];   // as classified by k_plan_hist
                keys[k] = key;
                ranks[k] = atomicAdd(&lcount[key], 1u);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kKeys; i += blockDim.x) {
            const uint32_t c = lcount[i];
            if (c) lcount[i] = atomicAdd(&args.cursor[i], c);   // count -> base of this tile's run of key i
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            uint64_t p = tile * kScatterTile + (uint64_t)k * kPlanThreads + threadIdx.x;
            if (keys[k] != 0xFFFFFFFFu) args.perm[lcount[keys[k]] + ranks[k]] = (uint32_t)p;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// The three planning passes in ONE launch, for batches that fit (<= kFusedPer pairs per thread of a grid that is
// resident as a whole: one 1024-thread workgroup per compute unit). Every workgroup classifies a contiguous slice
// of the batch once, keeping keys and in-key ranks in registers; its per-key counts are added to the global
// histogram with the atomic's RETURN value as the workgroup's offset inside the key; a grid-wide barrier later
// every workgroup scans the 6144 key totals for itself and writes its slice of `perm`. No key array, no second
// look at the offsets, no cursor array, two launch gaps fewer (k_plan_hist / k_plan_scan / k_plan_scatter: 14 + 6 + 12
// us + gaps on 1 M pairs; this: ~15). The histogram is double-buffered: a call zeroes the buffer of the next one.
// The barrier is an ever-growing counter compared against a per-call target (no reset, wrap-safe); it cannot deadlock
// as long as the grid fits the device, because workgroups waiting at it never keep the missing ones from being placed.
// ------------------------------------------------------------------------------------------------------------
constexpr int kFusedPer = 4;
struct FusedArgs {
    uint32_t *ghist;        // kKeys counters, zero on entry
    uint32_t *ghist_next;   // the other buffer: zeroed here for the next call
    uint32_t *barrier;      // monotonic arrival counter
    uint32_t target;        // value it reaches when every workgroup of THIS launch has arrived
    uint32_t chunk;         // pairs per workgroup
    uint32_t dblocks;       // k_direct_short's workgroups (their partial sums sit at partials[kMaxPartials ..])
};

template <typename Off>
__global__ __launch_bounds__(kPlanThreads) void k_plan_fused(PrepassArgs args, FusedArgs fused) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t lhist[kKeys];   // counts -> this workgroup's base inside each key -> final output base
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, wave_sum[kPlanThreads / 64];
    const bool planned = !(args.direct_short && *args.leftover == 0);   // k_direct_short finished every pair: nothing to sort
    for (int i = threadIdx.x; i < kKeys; i += blockDim.x) lhist[i] = 0;
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; }
    __syncthreads();
    uint32_t keys[kFusedPer], ranks[kFusedPer];
    const uint64_t first = (uint64_t)blockIdx.x * fused.chunk;
    const uint64_t last = first + fused.chunk < args.job.pairs ? first + fused.chunk : args.job.pairs;
    if (planned) {
        unsigned long long cells = 0, syms = 0;
        uint32_t maxa = 0, maxb = 0, shorts = 0;
        // every extent this thread needs is requested before the first key is derived (indices clamped into the batch,
        // so no branch sits between the loads): one memory round trip per thread instead of one per pair
        uint32_t las[kFusedPer], lbs[kFusedPer];
#pragma unroll
        for (int k = 0; k < kFusedPer; ++k) {
            const uint64_t p = first + (uint64_t)k * kPlanThreads + threadIdx.x;
            uint64_t a0, b0;
            pair_extent<Off>(args.job, p < args.job.pairs ? p : args.job.pairs - 1, a0, las[k], b0, lbs[k]);
        }
#pragma unroll
        for (int k = 0; k < kFusedPer; ++k) {
            const uint64_t p = first + (uint64_t)k * kPlanThreads + threadIdx.x;
            keys[k] = 0xFFFFFFFFu;
            if (p < last) {
                PairInfo info;
                info.la = las[k]; info.lb = lbs[k]; info.a0 = 0; info.b0 = 0;
                classify_trivial(args, info, args.gap_open, args.gap_extend, args.unit_costs != 0);
                const bool is_short = short_pair(info);
                shorts += is_short ? 1u : 0u;
                cells += (unsigned long long)info.la * info.lb;
                syms += (unsigned long long)info.la + info.lb;
                maxa = info.la > maxa ? info.la : maxa;
                maxb = info.lb > maxb ? info.lb : maxb;
                uint32_t key;
                if (args.redo_filter && (uint64_t)load_result(args.job, p) <= (uint64_t)args.redo_done_upto) {
                    key = kClassTrivial * kBuckets;   // settled by the call's first stage (the one-word band)
                } else if (info.trivial) {
                    key = kClassTrivial * kBuckets;
                    int64_t v = info.trivial_value;
                    if (args.job.negate) v = (int64_t)clamp_bound((uint32_t)(-v), args.job.bound);
                    if (!args.direct_short) store_result(args.job, p, v);   // (k_direct_short has stored it already)
                } else if ((is_short && args.direct_short) || (info.la <= args.skip_upto && info.lb <= args.skip_upto)) {
                    key = kClassTrivial * kBuckets;   // scored by k_direct_short (or, skip_upto, by the lane-per-pair alignment kernel)
                } else {
                    key = plan_key(info.la, info.lb, args.mode, args.symmetric, args.sym_bytes, args.banded, args.job.bound);
                    if (args.stage1 && key / kBuckets != (uint32_t)kClassBanded) {   // not the band's: left to the second stage
                        store_result(args.job, p, (int64_t)args.job.bound + 1);
                        key = kClassTrivial * kBuckets;
                    }
                }
                keys[k] = key;
                ranks[k] = atomicAdd(&lhist[key], 1u);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            cells += __shfl_xor(cells, off);
            syms += __shfl_xor(syms, off);
            shorts += __shfl_xor(shorts, off);
            const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
            maxa = oa > maxa ? oa : maxa;
            maxb = ob > maxb ? ob : maxb;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&lcells, cells);
            atomicAdd(&lsyms, syms);
            atomicMax(&lmaxa, maxa);
            atomicMax(&lmaxb, maxb);
            atomicAdd(&lshorts, shorts);
        }
    }
    __syncthreads();
    // my counts join the global histogram; what was there before is my offset inside the key
    for (int i = threadIdx.x; i < kKeys; i += blockDim.x) {
        const uint32_t c = lhist[i];
        lhist[i] = c ? atomicAdd(&fused.ghist[i], c) : 0u;
        if (blockIdx.x == 0) fused.ghist_next[i] = 0;
    }
    __syncthreads();   // every thread's atomics have returned before the workgroup reports its arrival
    if (threadIdx.x == 0) {
        PlanPartial part{lcells, lsyms, lmaxa, lmaxb, lshorts, 0};
        if (args.direct_short) part = PlanPartial{0, 0, 0, 0, 0, 0};   // the sums are k_direct_short's
        // ordering by completion instead of `__threadfence()` (an L2 write-back per workgroup): see report_call_summary
        unsigned long long *row = (unsigned long long *)&args.partials[blockIdx.x];
        __hip_atomic_store(row + 0, part.cells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + 1, part.symbols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + 2, (unsigned long long)part.max_la | (unsigned long long)part.max_lb << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + 3, (unsigned long long)part.short_pairs | (unsigned long long)part.pad << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Grid barrier with ONE agreed outcome. Arrivals are counted in barrier[0] (monotonic, per-call target); the outcome of
        // this call lives in barrier[1] as (target << 2) | state, state 1 = everybody arrived, 2 = given up, and is set by a
        // compare-and-swap -- by the last workgroup to arrive, or by one that has waited ~2 s (the launch is sized to be
        // resident as a whole, but a device shared with somebody else's long-running kernels could still keep workgroups
        // out). Whoever swaps first decides for all: no workgroup can publish its slice of `perm` while another one skips its
        // own. After a give-up the host falls back to the three-pass planner and never uses this kernel on the scope again.
        const uint32_t tag = fused.target << 2;
        auto decide = [&](uint32_t state) {
            uint32_t seen = __hip_atomic_load(fused.barrier + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((seen & ~3u) != tag || (seen & 3u) == 0) {   // (a value of another call's tag is stale: overwrite it)
                if (__hip_atomic_compare_exchange_strong(fused.barrier + 1, &seen, tag | state, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            }
        };
        if (atomicAdd(fused.barrier, 1u) + 1u == fused.target) decide(1u);
        const unsigned long long spin_start = __builtin_readcyclecounter();
        uint32_t outcome;
        for (;;) {
            outcome = __hip_atomic_load(fused.barrier + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((outcome & ~3u) == tag && (outcome & 3u)) break;
            if (__builtin_readcyclecounter() - spin_start > 5000000000ull) decide(2u);
            else __builtin_amdgcn_s_sleep(2);
        }
        const bool gave_up = (outcome & 3u) != 1u;
        wave_sum[0] = gave_up ? 1u : 0u;   // what crosses the barrier (key totals, partial rows) is read with agent-scope atomic loads
    }
    __syncthreads();
    if (wave_sum[0]) {   // an empty plan with the failure mark: the DP kernels find nothing to do, the host redoes the call
        if (blockIdx.x == 0) {
            for (int c = threadIdx.x; c <= kMaxClasses; c += blockDim.x) {
                args.plan->class_start[c] = 0;
                if (c < kMaxClasses) args.plan->class_count[c] = 0;
            }
            if (threadIdx.x == 0) { args.plan->fused_failed = 1; *args.leftover = 0; }
        }
        return;
    }
    __syncthreads();
    // every workgroup scans the key totals for itself: kKeys / 1024 consecutive keys per thread
    constexpr int kPerThread = (kKeys + kPlanThreads - 1) / kPlanThreads;
    uint32_t totals[kPerThread], sum = 0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const int i = threadIdx.x * kPerThread + q;
        totals[q] = i < kKeys ? __hip_atomic_load(&fused.ghist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        sum += totals[q];
    }
    const uint32_t incl = wave_inclusive_sum_u32(sum);
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += wave_sum[w];
    if (blockIdx.x == 0) {
        // the plan: class ranges from the key prefixes, work units folded from the partial sums
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {
            const int i = threadIdx.x * kPerThread + q;
            if (i < kKeys && i % kBuckets == 0) args.plan->class_start[i / kBuckets] = run + [&] { uint32_t before = 0; for (int r = 0; r < q; ++r) before += totals[r]; return before; }();
        }
        if (threadIdx.x == kPlanThreads - 1) args.plan->class_start[kMaxClasses] = run + sum;
    }
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const int i = threadIdx.x * kPerThread + q;
        if (i < kKeys) lhist[i] += run;
        run += totals[q];
    }
    __syncthreads();
    if (planned) {
#pragma unroll
        for (int k = 0; k < kFusedPer; ++k) {
            const uint64_t p = first + (uint64_t)k * kPlanThreads + threadIdx.x;
            if (keys[k] != 0xFFFFFFFFu) args.perm[lhist[keys[k]] + ranks[k]] = (uint32_t)p;
        }
    }
    if (blockIdx.x != 0) return;
    // ---- workgroup 0 finishes the plan ------------------------------------------------------------------------------
    __syncthreads();
    if (threadIdx.x < kMaxClasses) {
        const uint32_t s0 = args.plan->class_start[threadIdx.x], s1 = args.plan->class_start[threadIdx.x + 1];
        args.plan->class_count[threadIdx.x] = s1 - s0;
    }
    unsigned long long c = 0, sy = 0;
    uint32_t ma = 0, mb = 0, sh = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x + fused.dblocks; i += blockDim.x) {
        const bool from_plan = i < gridDim.x;
        const unsigned long long *q = (const unsigned long long *)&args.partials[from_plan ? i : kMaxPartials + (i - gridDim.x)];
        c += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sy += __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w2 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w3 = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ma = (uint32_t)w2 > ma ? (uint32_t)w2 : ma;
        mb = (uint32_t)(w2 >> 32) > mb ? (uint32_t)(w2 >> 32) : mb;
        sh += (uint32_t)w3;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        c += __shfl_xor(c, off);
        sy += __shfl_xor(sy, off);
        sh += __shfl_xor(sh, off);
        const uint32_t oa = __shfl_xor(ma, off), ob = __shfl_xor(mb, off);
        ma = oa > ma ? oa : ma;
        mb = ob > mb ? ob : mb;
    }
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&lcells, c);
        atomicAdd(&lsyms, sy);
        atomicMax(&lmaxa, ma);
        atomicMax(&lmaxb, mb);
        atomicAdd(&lshorts, sh);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        args.plan->cells = lcells; args.plan->symbols = lsyms; args.plan->max_la = lmaxa; args.plan->max_lb = lmaxb;
        args.plan->invalid_utf8 = 0; args.plan->short_pairs = lshorts;
        args.plan->fused_failed = 0;
        *args.leftover = 0;
    }
}

// One pair per lane in tape order; both strings <= 32 bytes. The recurrence is the single-block case of
// bitparallel.hip (no systolic hand-off): the longer string is the pattern (a table update per byte is cheaper than a
// DP column per byte). Memory is software-pipelined two rounds deep: while round r computes, the strings of round
// r + 1 (two 128-bit loads per string) and the extents of round r + 2 are in flight, so a round never waits for a
// full memory latency; 8 KB of LDS per wave leaves 16 waves per CU on top of that.
// kWide: both tapes hold at least 16 bytes (128-bit loads, bp_window.hpp); compile-time for the reason given there.
template <typename Off, bool kWide>
__device__ __forceinline__ void direct_short_run(const PrepassArgs &args, uint32_t *table, const uint64_t a_total,
                                                 const uint64_t b_total) {
    const int lane = threadIdx.x & 63;
    NibbleTables nib;
    nib.init(table, lane);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (args.job.pairs + stride - 1) / stride;  // wave-uniform trip count
    const uint64_t lane_first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // `safe`: both 32-byte windows of the pair lie inside their tapes, so they can be read as they are; the clamped
    // ByteWindow machinery (64-bit bounds per string) is only built for the first / last strings of a tape.
    struct Short { bool inside, direct, trivial, a_is_pattern, safe; uint64_t p, p0, t0; int64_t trivial_value; uint32_t la, lb, m, n; };
    struct Words { uint32_t pw[8], tw[8]; int moved[4]; };
    const uint8_t *a_data = (const uint8_t *)args.job.a.data, *b_data = (const uint8_t *)args.job.b.data;
    auto open_pair = [&](uint64_t round) -> Short {
        Short sp;
        sp.p = round * stride + lane_first;
        PairInfo info{};
        info.trivial = true;
        const bool inside = round < rounds && sp.p < args.job.pairs;
        if (inside) info = pair_info<Off>(args, sp.p, args.gap_open, args.gap_extend, true);
        sp.inside = inside;
        sp.trivial = inside && info.trivial;
        sp.trivial_value = info.trivial_value;
        sp.la = inside ? info.la : 0;
        sp.lb = inside ? info.lb : 0;
        sp.direct = inside && short_pair(info);
        sp.a_is_pattern = info.la >= info.lb;
        sp.m = sp.direct ? (sp.a_is_pattern ? info.la : info.lb) : 0;
        sp.n = sp.direct ? (sp.a_is_pattern ? info.lb : info.la) : 0;
        sp.p0 = sp.a_is_pattern ? info.a0 : info.b0;
        sp.t0 = sp.a_is_pattern ? info.b0 : info.a0;
        sp.safe = info.a0 + 32 <= a_total && info.b0 + 32 <= b_total;   // lanes without a pair: offsets 0
        return sp;
    };
    auto windows = [&](const Short &sp, ByteWindow &pat, ByteWindow &txt) {
        pat.init(sp.a_is_pattern ? a_data : b_data, sp.p0, sp.a_is_pattern ? a_total : b_total);
        txt.init(sp.a_is_pattern ? b_data : a_data, sp.t0, sp.a_is_pattern ? b_total : a_total);
    };
    auto request = [&](const Short &sp, Words &w) {
        if (kWide && __all(sp.safe)) {
            const uint8_t *pp = (sp.a_is_pattern ? a_data : b_data) + sp.p0, *tp = (sp.a_is_pattern ? b_data : a_data) + sp.t0;
            uint4 v;
            __builtin_memcpy(&v, pp, 16);
            w.pw[0] = v.x; w.pw[1] = v.y; w.pw[2] = v.z; w.pw[3] = v.w;
            __builtin_memcpy(&v, tp, 16);
            w.tw[0] = v.x; w.tw[1] = v.y; w.tw[2] = v.z; w.tw[3] = v.w;
#pragma unroll
            for (int q = 0; q < 4; ++q) { w.pw[4 + q] = 0; w.tw[4 + q] = 0; w.moved[q] = 0; }
            if (__any(sp.m > 16)) {   // bytes 16..31 only when some lane's string is that long (word lists: hardly ever)
                __builtin_memcpy(&v, pp + 16, 16);
                w.pw[4] = v.x; w.pw[5] = v.y; w.pw[6] = v.z; w.pw[7] = v.w;
            }
            if (__any(sp.n > 16)) {
                __builtin_memcpy(&v, tp + 16, 16);
                w.tw[4] = v.x; w.tw[5] = v.y; w.tw[6] = v.z; w.tw[7] = v.w;
            }
            return;
        }
        ByteWindow pat, txt;
        windows(sp, pat, txt);
        if constexpr (kWide) {
            uint32_t half[4];
            w.moved[0] = pat.fetch16_raw(0, half);
#pragma unroll
            for (int q = 0; q < 4; ++q) w.pw[q] = half[q];
            w.moved[1] = pat.fetch16_raw(16, half);
#pragma unroll
            for (int q = 0; q < 4; ++q) w.pw[4 + q] = half[q];
            w.moved[2] = txt.fetch16_raw(0, half);
#pragma unroll
            for (int q = 0; q < 4; ++q) w.tw[q] = half[q];
            w.moved[3] = txt.fetch16_raw(16, half);
#pragma unroll
            for (int q = 0; q < 4; ++q) w.tw[4 + q] = half[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) { w.pw[q] = pat.fetch4(q * 4); w.tw[q] = txt.fetch4(q * 4); }
#pragma unroll
            for (int q = 0; q < 4; ++q) w.moved[q] = 0;
        }
    };
    // this kernel sees every pair, so it also finishes the trivial ones and sums the work units (PrepassArgs::leftover)
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, left = 0;
    Short cur = open_pair(0);
    Words words;
    request(cur, words);
    Short next = open_pair(1);
    for (uint64_t round = 0; round < rounds; ++round) {
        Words words_next;
        request(next, words_next);
        const Short next2 = open_pair(round + 2);
        if (cur.inside) {
            cells += (unsigned long long)cur.la * cur.lb;
            syms += (unsigned long long)cur.la + cur.lb;
            maxa = cur.la > maxa ? cur.la : maxa;
            maxb = cur.lb > maxb ? cur.lb : maxb;
            shorts += cur.direct ? 1u : 0u;
            left += (!cur.trivial && !cur.direct) ? 1u : 0u;
            if (cur.trivial) {
                // unit costs: distances are stored positive, cut off at bound + 1 (same as k_plan_hist)
                int64_t v = cur.trivial_value;
                if (args.job.negate) v = (int64_t)clamp_bound((uint32_t)(-v), args.job.bound);
                store_result(args.job, cur.p, v);
            }
        }
        if (__any(cur.direct)) {
            const uint32_t m = cur.m, n = cur.n;
            // wave maxima of the two lengths (both <= 32) in one reduction, by DPP (no LDS crossbar traffic)
            // wave maxima of the two lengths by DPP (no LDS crossbar traffic)
            const uint32_t m_max = wave_max_u32(m), n_max = wave_max_u32(n);
            if constexpr (kWide) {   // windows the clamp had to move (first / last strings of a tape): re-read by dword
                uint32_t half[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    uint32_t *dst = h < 2 ? words.pw + 4 * h : words.tw + 4 * (h - 2);
                    if (__builtin_expect(words.moved[h] != 0, 0)) {
                        ByteWindow pat, txt;
                        windows(cur, pat, txt);
                        (h < 2 ? pat : txt).fix16(16 * (h & 1), words.moved[h], half);
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[q] = half[q];
                    }
                }
            }
            const uint32_t row_mask = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if ((uint32_t)q * 4 >= m_max) break;
                // rows past the pattern OR in a zero: no branch per byte
                const uint32_t dw = words.pw[q];
                nib.template insert<0>(dw, row_mask & (1u << (q * 4 + 0)));
                nib.template insert<1>(dw, row_mask & (1u << (q * 4 + 1)));
                nib.template insert<2>(dw, row_mask & (1u << (q * 4 + 2)));
                nib.template insert<3>(dw, row_mask & (1u << (q * 4 + 3)));
            }
            uint32_t pv = 0xFFFFFFFFu, mv = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if ((uint32_t)q * 4 >= n_max) break;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if ((uint32_t)(q * 4 + u) < n) {
                        const uint32_t tw = words.tw[q];
                        const uint32_t eq = u == 0 ? nib.template lookup<0>(tw) : (u == 1 ? nib.template lookup<1>(tw) : (u == 2 ? nib.template lookup<2>(tw) : nib.template lookup<3>(tw)));
                        const uint32_t xv = eq | mv;
                        const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
                        uint32_t ph = mv | ~(xh | pv);
                        const uint32_t mh = pv & xh;
                        ph = (ph << 1) | 1u;
                        pv = (mh << 1) | ~(xv | ph);
                        mv = ph & xv;
                    }
                }
            }
            if (cur.direct) {
                const uint32_t mask = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);
                const uint32_t d = n + __popc(pv & mask) - __popc(mv & mask);
                store_result(args.job, cur.p, (int64_t)clamp_bound(d, args.job.bound));
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) table[k * 64 + lane] = 0;
        }
        cur = next;
        words = words_next;
        next = next2;
    }
    // block sums -> this kernel's partial slot; the count of unfinished pairs -> one atomic per block that has any
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, lleft;
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lleft = 0; }
    __syncthreads();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {   // wave-reduce first (see k_plan_hist)
        cells += __shfl_xor(cells, off);
        syms += __shfl_xor(syms, off);
        shorts += __shfl_xor(shorts, off);
        left += __shfl_xor(left, off);
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
        atomicAdd(&lleft, left);
    }
    __syncthreads();
    if (args.summary) {
        // on its own (launch_direct_short_alone): nobody plans the pairs this kernel could not score -- report them
        __shared__ SummaryLds summary_lds;
        report_call_summary(PlanPartial{lcells, lsyms, lmaxa, lmaxb, lshorts, lleft ? 1u : 0u}, args.partials + kMaxPartials,
                            args.done_counter, args.summary, summary_lds);
        return;
    }
    if (threadIdx.x == 0) {
        args.partials[kMaxPartials + blockIdx.x] = PlanPartial{lcells, lsyms, lmaxa, lmaxb, lshorts, 0};
        if (lleft) atomicAdd(args.leftover, lleft);
    }
}

template <typename Off>
__global__ __launch_bounds__(256, 4) void k_direct_short(PrepassArgs args) {
    __shared__ __attribute__((aligned(8192))) uint32_t ltable[4][32 * 64];  // per wave: EqLo[16][64] | EqHi[16][64]; 8 KB aligned for NibbleTables
    const int lane = threadIdx.x & 63;
    uint32_t *table = ltable[threadIdx.x >> 6];
    for (int k = 0; k < 32; ++k) table[k * 64 + lane] = 0;
    const uint64_t a_total = (uint64_t)((const Off *)args.job.a.offsets)[args.job.a.count];
    const uint64_t b_total = (uint64_t)((const Off *)args.job.b.offsets)[args.job.b.count];
    if (a_total >= 16 && b_total >= 16) direct_short_run<Off, true>(args, table, a_total, b_total);
    else direct_short_run<Off, false>(args, table, a_total, b_total);
}

static int direct_short_blocks(const Scope *scope, uint64_t pairs) {
    int dblocks = (int)((pairs + 255) / 256);
    if (dblocks > scope->compute_units * 4) dblocks = scope->compute_units * 4;
    if (dblocks > kMaxPartials) dblocks = kMaxPartials;
    return dblocks < 1 ? 1 : dblocks;
}

// Every pair is known (tape statistics) or believed (previous call; verified by the kernel) to be word-sized: one
// launch, no planning kernels, the summary comes back through host-mapped memory.
void launch_direct_short_alone(Scope *scope, const PrepassArgs &args_in) {
    PrepassArgs args = args_in;
    args.direct_short = 1;
    args.partials = scope->plan_partials;
    args.summary = scope->summary_target();
    args.done_counter = scope->done_counter;
    const int dblocks = direct_short_blocks(scope, args.job.pairs);
    StampGuard guard(scope, "direct_short");
    if (args.off64) hipLaunchKernelGGL(k_direct_short<uint64_t>, dim3(dblocks), dim3(256), 0, scope->stream, args);
    else hipLaunchKernelGGL(k_direct_short<uint32_t>, dim3(dblocks), dim3(256), 0, scope->stream, args);
    SWH_HIP_CHECK(hipGetLastError());
}

// Longest string of a tape, for swh_tape_prepare: grid-stride maximum of offsets[i + 1] - offsets[i].
template <typename Off>
__global__ __launch_bounds__(256) void k_tape_longest(const Off *offsets, uint64_t count, uint32_t *longest) {
    uint64_t best = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t len = (uint64_t)(offsets[i + 1] - offsets[i]);
        best = len > best ? len : best;
    }
    uint32_t v = best > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)best;
    v = wave_max_u32(v);
    // one atomic per workgroup: thousands of waves hitting one word serialise (~88 per microsecond)
    __shared__ uint32_t wave_best[4];
    if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) v = wave_best[w] > v ? wave_best[w] : v;
        if (v) atomicMax(longest, v);
    }
}
void launch_tape_longest(Scope *scope, const void *offsets, uint32_t off64, uint64_t count, uint32_t *longest) {
    if (!count) return;
    uint64_t blocks64 = (count + 1023) / 1024;
    const uint32_t blocks = (uint32_t)(blocks64 > (uint64_t)scope->compute_units * 2 ? (uint64_t)scope->compute_units * 2 : blocks64);
    if (off64) hipLaunchKernelGGL(k_tape_longest<uint64_t>, dim3(blocks), dim3(256), 0, scope->stream, (const uint64_t *)offsets, count, longest);
    else hipLaunchKernelGGL(k_tape_longest<uint32_t>, dim3(blocks), dim3(256), 0, scope->stream, (const uint32_t *)offsets, count, longest);
    SWH_HIP_CHECK(hipGetLastError());
}

void launch_prepass(Scope *scope, const PrepassArgs &args_in) {
    PrepassArgs args = args_in;
    hipStream_t stream = scope->stream;
    uint64_t pairs = args.job.pairs;
    int blocks = (int)((pairs + kPlanThreads * 2 - 1) / (kPlanThreads * 2));
    int max_blocks = scope->compute_units * 4;
    if (max_blocks > kMaxPartials) max_blocks = kMaxPartials;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    int dblocks = 0;
    // the planning passes as one launch when the batch fits a grid that is resident as a whole
    static const bool fused_off = [] { const char *e = test_hook("STRINGWARS_AMD_PLAN"); return e && !strcmp(e, "split"); }();
    if (scope->fused_per_cu < 0) {   // how many 1024-thread planning workgroups one compute unit holds (asked once)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_plan_fused<uint32_t>, kPlanThreads, 0) != hipSuccess) per_cu = 0;
        (void)hipGetLastError();
        scope->fused_per_cu = per_cu > 2 ? 2 : per_cu;
    }
    // A pipeline lane (api.hip: run_pipelined) shares the device with the other lane's calls -- possibly its planner, which spins
    // on a grid barrier of its own: each lane sizes its grid for half the device, so that two planners are resident together.
    const bool is_lane = scope->lane_done != nullptr;
    const uint32_t fused_slots = (uint32_t)scope->fused_per_cu * (uint32_t)scope->compute_units / (is_lane ? 2u : 1u);
    const uint64_t fused_capacity = (uint64_t)fused_slots * kPlanThreads * kFusedPer;
    const bool fused = !fused_off && !scope->fused_disabled && fused_slots > 0 && fused_slots <= (uint32_t)kMaxPartials && pairs <= fused_capacity;
    if (args.direct_short) {
        StampGuard guard(scope, "direct_short");
        dblocks = direct_short_blocks(scope, pairs);
        if (args.off64) hipLaunchKernelGGL(k_direct_short<uint64_t>, dim3(dblocks), dim3(256), 0, stream, args);
        else hipLaunchKernelGGL(k_direct_short<uint32_t>, dim3(dblocks), dim3(256), 0, stream, args);
    }
    if (fused) {
        // two workgroups per compute unit at most (32 waves, 49 KB of LDS: resident together on an idle device; next to
        // another lane's DP kernel they take turns, which the barrier tolerates); small batches use fewer, >= 2048 pairs each
        uint32_t nb = (uint32_t)((pairs + 2047) / 2048);
        if (nb > fused_slots) nb = fused_slots;
        // the outcome word carries the target shifted by two bits: start over long before the shift could lose anything
        if (scope->plan_barrier_target > (1u << 29)) {
            SWH_HIP_CHECK(hipMemsetAsync(scope->plan_barrier, 0, 2 * sizeof(uint32_t), stream));
            scope->plan_barrier_target = 0;
        }
        // test hook: a grid four times larger than the device holds, to exercise the barrier's give-up path
        static const bool oversubscribe = test_hook("STRINGWARS_AMD_FUSED_OVERSUBSCRIBE") != nullptr;
        if (oversubscribe) nb = 4 * fused_slots <= (uint32_t)kMaxPartials ? 4 * fused_slots : (uint32_t)kMaxPartials;
        if (nb < 1) nb = 1;
        FusedArgs f{};
        f.chunk = (uint32_t)((pairs + nb - 1) / nb);
        f.ghist = scope->plan_hist2[scope->plan_parity];
        f.ghist_next = scope->plan_hist2[scope->plan_parity ^ 1];
        scope->plan_parity ^= 1;
        f.barrier = scope->plan_barrier;
        scope->plan_barrier_target += nb;
        f.target = scope->plan_barrier_target;
        f.dblocks = (uint32_t)dblocks;
        StampGuard guard(scope, "plan_fused");
        if (args.off64) hipLaunchKernelGGL(k_plan_fused<uint64_t>, dim3(nb), dim3(kPlanThreads), 0, stream, args, f);
        else hipLaunchKernelGGL(k_plan_fused<uint32_t>, dim3(nb), dim3(kPlanThreads), 0, stream, args, f);
        SWH_HIP_CHECK(hipGetLastError());
        return;
    }
    {
        StampGuard guard(scope, "plan_hist");
        if (args.off64) hipLaunchKernelGGL(k_plan_hist<uint64_t>, dim3(blocks), dim3(kPlanThreads), 0, stream, args);
        else hipLaunchKernelGGL(k_plan_hist<uint32_t>, dim3(blocks), dim3(kPlanThreads), 0, stream, args);
    }
    {
        StampGuard guard(scope, "plan_scan");
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, stream, args.hist, args.cursor, args.plan,
                           args.partials, (uint32_t)blocks, (uint32_t)dblocks, args.leftover);
    }
    {
        StampGuard guard(scope, "plan_scatter");
        uint64_t tiles = (pairs + kScatterTile - 1) / kScatterTile;
        int sblocks = (int)(tiles < (uint64_t)scope->compute_units * 4 ? tiles : (uint64_t)scope->compute_units * 4);
        if (sblocks < 1) sblocks = 1;
        if (args.off64) hipLaunchKernelGGL(k_plan_scatter<uint64_t>, dim3(sblocks), dim3(kPlanThreads), 0, stream, args);
        else hipLaunchKernelGGL(k_plan_scatter<uint32_t>, dim3(sblocks), dim3(kPlanThreads), 0, stream, args);
    }
    SWH_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// UTF-8 staging, flat over the tape bytes (independent of how long the strings are):
//   k_utf8_tile_count : lead bytes (everything but 10xxxxxx) per 8 KB tile and per 256 B sub-tile
//   scan              : exclusive prefix of the tile counts = code-point index of each tile
//   k_utf8_tile_write : every lead byte decodes its sequence to symbols[prefix + rank]; strict validation
//                       (RFC 3629 / Rust `str`: no overlongs, no surrogates, <= U+10FFFF, no stray
//                       continuation bytes)
//   k_utf8_string_offsets : code-point offset of every string = tile prefix + sub-tile prefix + leads in
//                       the < 256 bytes before it; a string may not start with a continuation byte, which
//                       is what catches sequences straddling two strings.
// All loads are coalesced dwords; nothing loops over a string.
// ------------------------------------------------------------------------------------------------
// A block walks its tile in passes of 1 KB (256 threads x one dword, coalesced); sub-tiles of 256 B (one wave of one
// pass) keep the per-string work in k_utf8_string_offsets short. With one pass per block the launch rate of 50K+
// tiny blocks, not memory, set the pace (0.9 TB/s on the count kernel).
// (kUtf8Pass, kUtf8Passes, kUtf8Tile, kUtf8Subs: common.hpp)

__device__ __forceinline__ uint32_t load_tape_dword(const uint8_t *data, int64_t pos, int64_t total) {
    // bytes pos..pos+3, zero where outside [0, total). (A branch-free clamp-shift-mask form was measured: these
    // kernels are instruction-bound, and its 64-bit shifts cost more than the branch.)
    if (pos >= 0 && pos + 4 <= total) { uint32_t dw; __builtin_memcpy(&dw, data + pos, 4); return dw; }
    uint32_t dw = 0;
    for (int u = 0; u < 4; ++u) {
        int64_t q = pos + u;
        if (q >= 0 && q < total) dw |= (uint32_t)data[q] << (8 * u);
    }
    return dw;
}
__device__ __forceinline__ uint32_t lead_mask4(uint32_t dw, int valid) {
    // bit u set when byte u (u < valid) is NOT a continuation byte (10xxxxxx)
    uint32_t m = 0;
    for (int u = 0; u < 4; ++u) {
        uint32_t b = (dw >> (8 * u)) & 0xffu;
        if (u < valid && (b & 0xC0u) != 0x80u) m |= 1u << u;
    }
    return m;
}

#ifdef SWH_TEST_HOOKS   // the count / scan / write staging of rounds 1-2: the tests' second implementation (STRINGWARS_AMD_UTF8_SCAN)
__global__ __launch_bounds__(256) void k_utf8_tile_count(const uint8_t *data, uint64_t total, uint32_t *tile_counts,
                                                         uint32_t *sub_prefix) {
    __shared__ uint32_t wave_sum[kUtf8Passes][4];
    const uint64_t tile = blockIdx.x;
    const int64_t tot = (int64_t)total;
    uint32_t dws[kUtf8Passes];
    int valids[kUtf8Passes];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {   // all loads first: one memory latency per block
        const int64_t pos = (int64_t)(tile * kUtf8Tile + q * kUtf8Pass + threadIdx.x * 4);
        valids[q] = tot - pos >= 4 ? 4 : (tot > pos ? (int)(tot - pos) : 0);
        dws[q] = valids[q] ? load_tape_dword(data, pos, tot) : 0;
    }
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        uint32_t cnt = __popc(lead_mask4(dws[q], valids[q]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if ((threadIdx.x & 63) == 0) wave_sum[q][threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int q = 0; q < kUtf8Passes; ++q)
            for (int w = 0; w < 4; ++w) { sub_prefix[tile * kUtf8Subs + q * 4 + w] = run; run += wave_sum[q][w]; }
        tile_counts[tile] = run;
    }
}

// Validation: every lead byte checks its own sequence (continuation bytes present, no overlong, no
// surrogate, <= U+10FFFF, inside the tape). Stray continuation bytes are caught by a global balance:
// sum over leads of (length - 1) must equal the number of continuation bytes. Claimed ranges are disjoint
// (a claimed byte is a continuation byte, so no lead sits inside another lead's range), hence equality
// means every continuation byte is claimed exactly once.
struct Utf8WriteLds {
    uint32_t wave_tot[kUtf8Passes][4];
    int wave_bal[4];
    union {
        uint32_t stage[2][kUtf8Pass];   // edge tiles: code points of a pass in rank order
        uint16_t leads[kUtf8Tile];      // interior tiles: byte positions of the tile's lead bytes in rank order
    };
    uint32_t raw[kUtf8Tile / 4 + 4];    // interior tiles: the tile's bytes (+ the word after it)
};

// Generic form (bounds-checked loads, branchy decode): the tiles at the end of the tape.
__device__ __noinline__ void utf8_tile_write_edge(Utf8WriteLds &lds, const uint8_t *data, uint64_t total,
                                                   const uint64_t *tile_prefix, uint32_t *symbols, uint32_t *invalid,
                                                   int *balance) {
    auto &wave_tot = lds.wave_tot;
    auto &wave_bal = lds.wave_bal;
    auto &stage = lds.stage;
    const uint64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tot = (int64_t)total;
    // all loads first (the word after mine is somebody's `cur`, so it comes out of the cache)
    uint32_t curs[kUtf8Passes], nexts[kUtf8Passes];
    int valids[kUtf8Passes];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        const int64_t pos = (int64_t)(tile * kUtf8Tile + q * kUtf8Pass + threadIdx.x * 4);
        valids[q] = tot - pos >= 4 ? 4 : (tot > pos ? (int)(tot - pos) : 0);
        curs[q] = valids[q] ? load_tape_dword(data, pos, tot) : 0;
        nexts[q] = valids[q] ? load_tape_dword(data, pos + 4, tot) : 0;
    }
    uint32_t incls[kUtf8Passes];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        uint32_t incl = __popc(lead_mask4(curs[q], valids[q]));
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        incls[q] = incl;
        if (lane == 63) wave_tot[q][wave] = incl;
    }
    __syncthreads();
    int bal = 0;  // expected continuation bytes of my leads minus continuation bytes I hold
    bool bad = false;
    int64_t bad_pos = 0;
    uint64_t pass_base = tile_prefix[tile];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        const int64_t pos = (int64_t)(tile * kUtf8Tile + q * kUtf8Pass + threadIdx.x * 4);
        const uint32_t cur = curs[q], next = nexts[q];
        const int valid = valids[q];
        const bool ascii = (cur & 0x80808080u) == 0;
        const uint32_t leads = lead_mask4(cur, valid);
        const uint32_t mine = __popc(leads);
        uint32_t cps[4] = {0, 0, 0, 0};
        bool bad_here = false;
        if (ascii) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cps[u] = (cur >> (8 * u)) & 0xffu;
        } else {
            const unsigned long long w64 = (unsigned long long)cur | ((unsigned long long)next << 32);
            bal -= valid - (int)mine;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!(leads & (1u << u))) continue;
                const uint32_t seq = (uint32_t)(w64 >> (8 * u));
                const uint32_t c = seq & 0xffu, b1 = (seq >> 8) & 0xffu, b2 = (seq >> 16) & 0xffu, b3 = seq >> 24;
                uint32_t cp = c;
                if (c >= 0x80u) {
                    const int need = c >= 0xF0u ? 3 : (c >= 0xE0u ? 2 : 1);
                    bad_here |= c < 0xC2u || c > 0xF4u;
                    bad_here |= pos + u + need >= tot;
                    bad_here |= (b1 & 0xC0u) != 0x80u;
                    cp = need == 1 ? (c & 0x1Fu) : (need == 2 ? (c & 0x0Fu) : (c & 0x07u));
                    cp = (cp << 6) | (b1 & 0x3Fu);
                    if (need >= 2) { bad_here |= (b2 & 0xC0u) != 0x80u; cp = (cp << 6) | (b2 & 0x3Fu); }
                    if (need == 3) { bad_here |= (b3 & 0xC0u) != 0x80u; cp = (cp << 6) | (b3 & 0x3Fu); }
                    bad_here |= need == 2 && (cp < 0x800u || (cp >= 0xD800u && cp <= 0xDFFFu));
                    bad_here |= need == 3 && (cp < 0x10000u || cp > 0x10FFFFu);
                    bal += need;
                }
                cps[u] = cp;
            }
        }
        if (bad_here && !bad) { bad = true; bad_pos = pos; }
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wave_tot[q][w];
        uint32_t rank = base + incls[q] - mine;   // within the pass
        uint32_t *buf = stage[q & 1];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (leads & (1u << u)) buf[rank++] = cps[u];
        __syncthreads();
        const uint32_t pass_total = wave_tot[q][0] + wave_tot[q][1] + wave_tot[q][2] + wave_tot[q][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = k * 256 + threadIdx.x;
            if (i < pass_total) symbols[pass_base + i] = buf[i];
        }
        pass_base += pass_total;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bal += __shfl_xor(bal, off);
    if (lane == 0) wave_bal[wave] = bal;
    if (bad) atomicCAS(invalid, 0u, (uint32_t)(bad_pos >> 2) + 1u);
    __syncthreads();
    // per-tile balance (sequences straddling a tile edge make it non-zero per tile, zero over the tape);
    // a single hot atomic here serialised ~100K tiles and cost more than the decode itself
    if (threadIdx.x == 0) balance[tile] = wave_bal[0] + wave_bal[1] + wave_bal[2] + wave_bal[3];
}

#endif   // SWH_TEST_HOOKS
// Lead bytes (everything but 10xxxxxx) of four packed bytes as 0x80 flags: !bit7 | bit6.
__device__ __forceinline__ uint32_t lead_flags4(uint32_t dw) { return (~dw | (dw << 1)) & 0x80808080u; }

#ifdef SWH_TEST_HOOKS   // count / scan / write staging, continued
// One UTF-8 sequence starting in the low byte of `seq` (its next three bytes above it), without branches:
// n1 = leading one bits of the first byte (0: ASCII, 2..4: lead of a 2..4 byte sequence), the payload bits of all
// four bytes are packed as if the sequence were four bytes long and shifted down by the bytes it does not have.
// `bad` collects: missing continuation bytes, overlong forms, surrogates, > U+10FFFF, lead bytes F8..FF.
__device__ __forceinline__ uint32_t utf8_decode_one(uint32_t seq, uint32_t &need, bool &bad) {
    const uint32_t c = seq & 0xffu;
    const uint32_t n1 = (uint32_t)__builtin_clz(~(seq << 24) | 0x00800000u);   // <= 8
    need = n1 ? n1 - 1 : 0;                                                        // continuation bytes to follow
    const uint32_t full = ((c & (0x7fu >> n1)) << 18) | ((seq >> 8 & 0x3fu) << 12) | ((seq >> 16 & 0x3fu) << 6) | (seq >> 24 & 0x3fu);
    const uint32_t cp = full >> (6 * (3 - (need > 3 ? 3 : need)));
    const uint32_t cont_wrong = ((seq >> 8 & 0xC0C0C0u) ^ 0x808080u) & ((1u << (8 * (need > 3 ? 3 : need))) - 1u);
    const uint32_t min_cp = (1u << ((0x100B0700u >> (8 * (need > 3 ? 3 : need))) & 0x1fu)) & ~1u;   // 0, 0x80, 0x800, 0x10000
    bad = n1 > 4 || cont_wrong != 0 || cp < min_cp || cp > 0x10FFFFu || (cp - 0xD800u) < 0x800u;
    return cp;
}

// Validation: every lead byte checks its own sequence (continuation bytes present, no overlong, no
// surrogate, <= U+10FFFF, inside the tape). Stray continuation bytes are caught by a global balance:
// sum over leads of (length - 1) must equal the number of continuation bytes. Claimed ranges are disjoint
// (a claimed byte is a continuation byte, so no lead sits inside another lead's range), hence equality
// means every continuation byte is claimed exactly once.
__global__ __launch_bounds__(256) void k_utf8_tile_write(const uint8_t *data, uint64_t total, const uint64_t *tile_prefix,
                                                         uint32_t *symbols, uint32_t *invalid, int *balance) {
    __shared__ Utf8WriteLds lds;
    const uint64_t tile = blockIdx.x;
    // interior tiles: every word this block touches (its own and the one after its last) lies inside the tape
    if ((tile + 1) * kUtf8Tile + 4 > total) {
        utf8_tile_write_edge(lds, data, total, tile_prefix, symbols, invalid, balance);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *src = data + tile * kUtf8Tile + threadIdx.x * 4;
    uint32_t curs[kUtf8Passes];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) __builtin_memcpy(&curs[q], src + q * kUtf8Pass, 4);   // all loads first
    uint32_t edge = 0;   // the word after the tile: look-ahead of its last sequences
    if (threadIdx.x == 0) __builtin_memcpy(&edge, data + (tile + 1) * kUtf8Tile, 4);
    uint32_t incls[kUtf8Passes];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        incls[q] = wave_inclusive_sum_u32((uint32_t)__popc(lead_flags4(curs[q])));
        if (lane == 63) lds.wave_tot[q][wave] = incls[q];
        lds.raw[q * 256 + threadIdx.x] = curs[q];
    }
    if (threadIdx.x == 0) lds.raw[kUtf8Tile / 4] = edge;
    __syncthreads();
    // Work is handed out per SEQUENCE, not per byte: the byte positions of the tile's lead bytes are listed in rank
    // order (LDS), then thread i decodes sequences i, i + 256, ... from the tile's bytes (also in LDS) and stores code
    // points i, i + 256, ... -- coalesced, with no barrier between them, and nothing is decoded for continuation bytes
    // (two of every four bytes in CJK / Cyrillic text).
    int bal = 0;  // continuation bytes the sequences I decode expect, minus continuation bytes I hold
    uint32_t before = 0;   // code points of the passes before q
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        const uint32_t flags = lead_flags4(curs[q]);
        const uint32_t mine = (uint32_t)__popc(flags);
        bal -= 4 - (int)mine;
        uint32_t base = before;
        for (int w = 0; w < wave; ++w) base += lds.wave_tot[q][w];
        uint32_t rank = base + incls[q] - mine;   // within the tile
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((flags >> (8 * u + 7)) & 1u) lds.leads[rank++] = (uint16_t)(q * kUtf8Pass + threadIdx.x * 4 + u);
        before += lds.wave_tot[q][0] + lds.wave_tot[q][1] + lds.wave_tot[q][2] + lds.wave_tot[q][3];
    }
    __syncthreads();
    const uint32_t tile_total = before;
    uint32_t *out = symbols + tile_prefix[tile];
    bool bad = false;
    uint32_t bad_at = 0;
    for (uint32_t i = threadIdx.x; i < tile_total; i += 256) {
        const uint32_t at = lds.leads[i];
        const uint32_t lo = lds.raw[at >> 2], hi = lds.raw[(at >> 2) + 1];
        const uint32_t seq = (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * (at & 3u)));
        uint32_t need;
        bool bad_here;
        const uint32_t cp = utf8_decode_one(seq, need, bad_here);
        bal += (int)need;
        if (bad_here && !bad) { bad = true; bad_at = at; }
        out[i] = cp;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bal += __shfl_xor(bal, off);
    if (lane == 0) lds.wave_bal[wave] = bal;
    if (bad) atomicCAS(invalid, 0u, (uint32_t)((tile * kUtf8Tile + bad_at) >> 2) + 1u);
    __syncthreads();
    if (threadIdx.x == 0) balance[tile] = lds.wave_bal[0] + lds.wave_bal[1] + lds.wave_bal[2] + lds.wave_bal[3];
}

#endif   // SWH_TEST_HOOKS
// ------------------------------------------------------------------------------------------------
// The same staging in ONE pass over the tape (what launch_utf8_decode runs unless STRINGWARS_AMD_UTF8_SCAN=split): count,
// prefix and decode per tile in one kernel, the tile's code-point index found by a decoupled look-back over the tiles before it
// instead of a count kernel + a scan kernel (one read of the tape less, two launches less per tape).
//   status[tile] = epoch | flag | count: flag 1 -- the tile's own code points (published as soon as they are counted),
//                                        flag 2 -- all code points up to and including the tile (published after the look-back).
// Tiles are handed out by a ticket, so every tile before mine belongs to a workgroup that is already running (or done) and
// publishes its flag-1 word without waiting for anybody: the look-back cannot deadlock. Wave 0 looks back 64 tiles at a time.
// tile_prefix[] / sub_prefix[] are still written: k_utf8_string_offsets reads them.
// ------------------------------------------------------------------------------------------------
// A status word: call epoch << 48 | flag << 46 | count. The words live in a buffer of the scope's own that only this kernel
// writes (launch_utf8_decode_pair), so a word of another epoch is simply "not there yet" and nothing has to be zeroed per call.
constexpr unsigned long long kStatusCount = (1ull << 46) - 1;
__device__ __forceinline__ unsigned long long status_word(uint32_t epoch, uint32_t flag, unsigned long long count) {
    return ((unsigned long long)epoch << 48) | ((unsigned long long)flag << 46) | count;
}
__device__ __forceinline__ uint32_t status_flag(unsigned long long word, uint32_t epoch) {
    return (uint32_t)(word >> 48) == epoch ? (uint32_t)(word >> 46) & 3u : 0u;
}

// What the first byte of a sequence says about it, one 16-byte entry per byte value (a workgroup keeps the table in LDS):
//   lead  -- its payload bits where they sit in the scalar value (<< 6 per continuation byte);
//   shape -- bits 0..4: how far the 18 payload bits of the three bytes after it are shifted down (6 per byte the sequence does not
//            have), bits 16..: the sequence's length in bytes (summed per thread: the balance of claimed and held bytes);
//   cont  -- bits 0..4: how many of those payload bits belong to the sequence (the width of a v_bfe), bits 22, 23, 30, 31: the top
//            two bits of its third and fourth byte where it has them -- they must read 10;
//   range -- low half / high half: the range of (second byte << 8 | first byte): the second byte is a continuation byte, E0 / F0
//            are not followed by an overlong form, ED not by a surrogate, F4 by nothing past U+10FFFF; C0, C1, F5..FF are never
//            valid (an empty range: no value is its own median there).
// Continuation bytes get a neutral entry (length 0, any second byte): every byte position is looked up, only lead bytes store.
struct alignas(16) Utf8Lead { uint32_t lead, shape, cont, range; };
constexpr Utf8Lead utf8_lead(uint32_t b) {
    if (b < 0x80u) return Utf8Lead{b, 18u | 1u << 16, 0u, (0xFF00u | b) << 16 | b};
    if (b < 0xC0u) return Utf8Lead{0u, 18u, 0u, 0xFFFF0000u};
    if (b < 0xC2u || b > 0xF4u) return Utf8Lead{0u, 18u | 1u << 16, 0u, (b ^ 0xFFu) << 16 | (b ^ 0xFFu)};
    const uint32_t lo = b == 0xE0u ? 0xA0u : (b == 0xF0u ? 0x90u : 0x80u), hi = b == 0xEDu ? 0x9Fu : (b == 0xF4u ? 0x8Fu : 0xBFu);
    const uint32_t need = b < 0xE0u ? 1u : (b < 0xF0u ? 2u : 3u);
    const uint32_t payload = b & (0x3Fu >> need);
    const uint32_t tops = need == 3u ? 0xC0C00000u : (need == 2u ? 0x00C00000u : 0u);
    return Utf8Lead{payload << (6u * need), (18u - 6u * need) | (need + 1u) << 16, tops | 6u * need, (hi << 8 | b) << 16 | (lo << 8 | b)};
}
struct Utf8LeadTable { Utf8Lead e[256]; };
constexpr Utf8LeadTable make_utf8_leads() {
    Utf8LeadTable t{};
    for (uint32_t b = 0; b < 256; ++b) t.e[b] = utf8_lead(b);
    return t;
}
__device__ const Utf8LeadTable kUtf8Leads = make_utf8_leads();

struct Utf8DecodeLds {
    uint32_t wave_tot[kUtf8Passes][4];
    int wave_bal[4];
    uint16_t leads[kUtf8Tile];          // byte positions of the tile's lead bytes in rank order
    uint32_t raw[kUtf8Tile / 4 + 4];    // the tile's bytes (+ the word after it)
    Utf8Lead table[256];
};

// One tape of a staging launch (both tapes of a call are staged by ONE launch: tickets [0, a.tiles) are tape a's tiles).
struct Utf8TileJob {
    const uint8_t *data; uint64_t total, tiles;
    unsigned long long *status; uint64_t *tile_prefix; uint32_t *sub_prefix; uint32_t *symbols; int *balance;
};

__global__ __launch_bounds__(256) void k_utf8_tile_decode(Utf8TileJob job_a, Utf8TileJob job_b, uint32_t *ticket, uint32_t *invalid, uint32_t epoch) {
    __shared__ Utf8DecodeLds lds;
    __shared__ unsigned long long tile_base;
    __shared__ uint32_t my_tile;
    // One workgroup per tile. (Workgroups that loop over the ticket -- a grid of what the device holds at once -- take the same
    // time for one launch, and make two launches on two streams take turns instead of sharing the device: 0.40 ms for 2 x 100 MB
    // against 0.28. A note for whoever tries again: written as `for (;;) { draw; barrier; if (drawn >= tiles) break; ... }` on
    // the LDS word itself, hipcc wrapped the loop's barriers in per-wave exec-mask bookkeeping and the kernel hung on its first
    // tile; with the drawn ticket passed through readfirstlane and tested in the loop's `while` it ran.)
#ifdef SWH_UTF8_NO_TICKET
    // Diagnostic build only (make EXTRA=-DSWH_UTF8_NO_TICKET): tiles in blockIdx order. What the tickets cost: a 103 MB tape (12.5 K
    // tiles) is staged in 0.134 ms instead of 0.179 -- 3.6 ns per atomic on one address. Safe only if workgroups start in
    // blockIdx order; nothing promises that. (One ticket and one look-back per group of four tiles, the group's code points counted
    // up front and its tiles decoded one after the other by the workgroup, was built: 0.216 ms -- the tiles' words are read twice,
    // the kernel needs 92 registers and a workgroup lives four times as long. Taken out again.)
    const uint32_t drawn = blockIdx.x;
#else
    // Two tapes in one launch draw from TWO tickets, one per tape (words 32 apart: 128 bytes): a ticket is an atomic on one address,
    // 3.6 ns each -- 25 K tiles on one word were 90 us of a 0.25 ms kernel, which is what made one launch over both tapes slower
    // than a launch and a stream each. A workgroup asks its preferred tape first (blockIdx parity) and the other one when that tape
    // has no tile left; every tile before a drawn one belongs to a workgroup that is running or done, per tape, as before.
    if (threadIdx.x == 0) {
        const bool two = job_b.tiles != 0;
        uint32_t pick = two ? (blockIdx.x & 1u) : 0u;
        uint32_t t = atomicAdd(ticket + 32 * pick, 1u);
        if (two && t >= (pick ? job_b.tiles : job_a.tiles)) { pick ^= 1u; t = atomicAdd(ticket + 32 * pick, 1u); }
        my_tile = t | (pick << 31);
    }
    __syncthreads();
    const uint32_t drawn = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_tile);
#endif
    {
    const bool second = (drawn >> 31) != 0;
    const uint64_t tile_in_tape = drawn & 0x7FFFFFFFu;
    if (tile_in_tape >= (second ? job_b.tiles : job_a.tiles)) return;   // (cannot happen: the grid has as many workgroups as the tapes have tiles)
    const uint8_t *data = second ? job_b.data : job_a.data;
    const uint64_t total = second ? job_b.total : job_a.total, tiles = second ? job_b.tiles : job_a.tiles;
    unsigned long long *status = second ? job_b.status : job_a.status;
    uint64_t *tile_prefix = second ? job_b.tile_prefix : job_a.tile_prefix;
    uint32_t *sub_prefix = second ? job_b.sub_prefix : job_a.sub_prefix;
    uint32_t *symbols = second ? job_b.symbols : job_a.symbols;
    int *balance = second ? job_b.balance : job_a.balance;
    const uint64_t tile = tile_in_tape;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tot = (int64_t)total;
    // interior tiles: every word this block touches (its own and the one after its last) lies inside the tape; the others
    // load with bounds checks (bytes past the tape read as zero: never a lead -- masked below -- and never a continuation byte)
    const bool edge = (tile + 1) * kUtf8Tile + 4 > total;
    uint32_t curs[kUtf8Passes], flags[kUtf8Passes];
    uint32_t after = 0;   // the word after the tile: look-ahead of its last sequences
    int held = 0;         // continuation bytes in my words
    *(uint4 *)&lds.table[threadIdx.x] = *(const uint4 *)&kUtf8Leads.e[threadIdx.x];
    if (!edge) {
        const uint8_t *src = data + tile * kUtf8Tile + threadIdx.x * 4;
#pragma unroll
        for (int q = 0; q < kUtf8Passes; ++q) __builtin_memcpy(&curs[q], src + q * kUtf8Pass, 4);   // all loads first
        if (threadIdx.x == 0) __builtin_memcpy(&after, data + (tile + 1) * kUtf8Tile, 4);
#pragma unroll
        for (int q = 0; q < kUtf8Passes; ++q) { flags[q] = lead_flags4(curs[q]); held += 4 - __popc(flags[q]); }
    } else {
#pragma unroll
        for (int q = 0; q < kUtf8Passes; ++q) {
            const int64_t pos = (int64_t)(tile * kUtf8Tile + q * kUtf8Pass + threadIdx.x * 4);
            const int valid = tot - pos >= 4 ? 4 : (tot > pos ? (int)(tot - pos) : 0);
            curs[q] = valid ? load_tape_dword(data, pos, tot) : 0;
            flags[q] = valid ? lead_flags4(curs[q]) & (0x80808080u >> (8 * (4 - valid))) : 0u;
            held += valid - __popc(flags[q]);
        }
        if (threadIdx.x == 0) after = load_tape_dword(data, (int64_t)((tile + 1) * kUtf8Tile), tot);
    }
    {   // a byte above 0x7F anywhere in the tile: the tape is not pure ASCII (a plain store, every such wave writes the same 1)
        uint32_t high = 0;
#pragma unroll
        for (int q = 0; q < kUtf8Passes; ++q) high |= curs[q];
        if (__ballot((high & 0x80808080u) != 0) != 0 && lane == 0) {
            const uint32_t tape_no = (uint32_t)((ticket - invalid) / 32 - 1) + (second ? 1u : 0u);   // the launch's slot, + 1 for its second tape
            invalid[kUtf8AsciiWord + (tape_no & 1u)] = 1u;
        }
    }
    uint32_t incls[kUtf8Passes];
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        incls[q] = wave_inclusive_sum_u32((uint32_t)__popc(flags[q]));
        if (lane == 63) lds.wave_tot[q][wave] = incls[q];
        lds.raw[q * 256 + threadIdx.x] = curs[q];
    }
    if (threadIdx.x == 0) lds.raw[kUtf8Tile / 4] = after;
    __syncthreads();
    // the byte positions of the tile's lead bytes in rank order (see k_utf8_tile_write), and the tile's count
    uint32_t before = 0;   // code points of the passes before q
#pragma unroll
    for (int q = 0; q < kUtf8Passes; ++q) {
        const uint32_t mine = (uint32_t)__popc(flags[q]);
        uint32_t base = before;
        for (int w = 0; w < wave; ++w) base += lds.wave_tot[q][w];
        uint32_t rank = base + incls[q] - mine;   // within the tile
        if (q == 0 && threadIdx.x == 0) {
            // (the count is known to every thread only after this loop: thread 0 sums the 32 wave totals now and publishes)
            uint32_t sum = 0;
            for (int qq = 0; qq < kUtf8Passes; ++qq)
                for (int w = 0; w < 4; ++w) sum += lds.wave_tot[qq][w];
            __hip_atomic_store(status + tile, status_word(epoch, tile == 0 ? 2u : 1u, sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (writing every byte slot unconditionally -- lead bytes at their rank, the others into a dump word -- instead of a
        // store under a condition per slot: fewer SALU instructions, the same time)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((flags[q] >> (8 * u + 7)) & 1u) lds.leads[rank++] = (uint16_t)(q * kUtf8Pass + threadIdx.x * 4 + u);
        before += lds.wave_tot[q][0] + lds.wave_tot[q][1] + lds.wave_tot[q][2] + lds.wave_tot[q][3];
    }
    const uint32_t tile_total = before;
    if (wave == 1 && lane == 0) {   // the sub-tile prefixes k_utf8_string_offsets wants (one wave of one pass = 256 bytes)
        uint32_t run = 0;
        for (int q = 0; q < kUtf8Passes; ++q)
            for (int w = 0; w < 4; ++w) { sub_prefix[tile * kUtf8Subs + q * 4 + w] = run; run += lds.wave_tot[q][w]; }
    }
    if (wave == 0) {
        // look-back: lane l reads the status of tile - 1 - l of the current window
        unsigned long long exclusive = 0;
        for (int64_t first = (int64_t)tile - 1; first >= 0; first -= 64) {
            const int64_t at = first - lane;
            unsigned long long word = status_word(epoch, 2u, 0);   // before the tape: "everything up to here" = 0
            if (at >= 0) {
                while (status_flag(word = __hip_atomic_load(status + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), epoch) == 0) __builtin_amdgcn_s_sleep(1);
            }
            const unsigned long long closed = __ballot(status_flag(word, epoch) == 2u);
            const int stop = closed ? __builtin_ctzll(closed) : 64;   // the nearest tile that knows its inclusive prefix
            unsigned long long part = lane <= stop ? (word & kStatusCount) : 0ull;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
            exclusive += part;
            if (closed) break;
        }
        if (lane == 0) {
            if (tile) __hip_atomic_store(status + tile, status_word(epoch, 2u, exclusive + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tile_base = exclusive;
            tile_prefix[tile] = exclusive;
            if (tile + 1 == tiles) tile_prefix[tiles] = exclusive + tile_total;
        }
    }
    __syncthreads();
    uint32_t *out = symbols + tile_base;
    // One code point per lane and round, in rank order (the stores are full lines): the lead's byte position, the four bytes from
    // it on, one ds_read_b128 of what the lead byte says (table above), the range of the first two bytes (v_med3_u16: a value
    // inside the range is its own median), the third and fourth byte's top bits, the length, the payload of the three bytes after
    // it (three v_bfe, two v_lshl_or) cut to the sequence's own (v_bfe) under the lead's.
    // (27 instructions per code point where the scalar-assembling decode of round 3 took 65 -- and the same 0.25 ms for C3's two
    // tapes side by side: what bounds this kernel is the tiles in flight over a tile's latencies, not its instructions. Measured
    // on the way, all slower or equal, DESIGN 4.4: every byte position decoded in place with the lanes storing their own code
    // points (0.33 ms: the lines arrive in pieces), the same staged through LDS and stored in full lines (0.29), counts published
    // 2048 tiles ahead + the look-back before the scan (0.29), barriers that do not wait for stores (no change), half a tile of
    // lead positions at a time for 7 workgroups per CU instead of 5 (0.28). Timestamps per tile: ticket + loads ~3 us, look-back
    // ~5 us -- a round trip of the status words under the stores of 1.3 K tiles in flight --, decode + stores ~5 us.)
    uint32_t wrong_pair = 0, wrong_tops = 0, shapes = 0;
    for (uint32_t i = threadIdx.x; i < tile_total; i += 256) {
        const uint32_t at = lds.leads[i];
        const uint32_t lo = lds.raw[at >> 2], hi = lds.raw[(at >> 2) + 1];
        const uint32_t sq = (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * (at & 3u)));
        const uint4 e = *(const uint4 *)&lds.table[sq & 0xFFu];   // x lead, y shape, z cont, w range
        uint32_t inside;
        asm volatile("v_med3_u16 %0, %1, %2, %2 op_sel:[0,0,1,0]" : "=v"(inside) : "v"(sq), "v"(e.w));
        wrong_pair |= inside ^ sq;                       // (bits 16.. are noise: the low half is tested)
        wrong_tops |= (sq ^ 0x80800000u) & e.z;          // (bits 0..4 are noise: the top bits are tested)
        shapes += e.y;
        const uint32_t after3 = (((sq >> 8 & 0x3Fu) << 6 | (sq >> 16 & 0x3Fu)) << 6) | (sq >> 24 & 0x3Fu);
        out[i] = e.x | __builtin_amdgcn_ubfe(after3, e.y, e.z);
    }
    // the bytes my sequences claim minus the continuation bytes I hold (and, below, the tile's lead bytes): zero over a valid tape
    int bal = (int)(shapes >> 16) - held;
    const bool bad = ((wrong_pair & 0xFFFFu) | (wrong_tops & 0xC0C00000u)) != 0;
    const uint32_t bad_at = threadIdx.x * 4;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bal += __shfl_xor(bal, off);
    if (lane == 0) lds.wave_bal[wave] = bal;
    if (bad) atomicCAS(invalid, 0u, (uint32_t)((tile * kUtf8Tile + bad_at) >> 2) + 1u);
    __syncthreads();
    if (threadIdx.x == 0) balance[tile] = lds.wave_bal[0] + lds.wave_bal[1] + lds.wave_bal[2] + lds.wave_bal[3] - (int)tile_total;
    }
}

#ifdef SWH_TEST_HOOKS   // count / scan / write staging, continued
__global__ __launch_bounds__(256) void k_utf8_balance(const int *tile_balance, uint64_t tiles, int *balance) {
    __shared__ int red[256];
    int sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < tiles; i += (uint64_t)gridDim.x * 256) sum += tile_balance[i];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] != 0) atomicAdd(balance, red[0]);
}

#endif   // SWH_TEST_HOOKS
// Code-point offset of string i of a tape (i == count: the tape's end) and the check that it starts on a sequence boundary.
template <typename Off>
__device__ __forceinline__ void utf8_string_offset(const Utf8Args &args, const uint64_t *tile_prefix, const uint32_t *sub_prefix, uint64_t i) {
    const Off *offs = (const Off *)args.in.offsets;
    const uint8_t *data = (const uint8_t *)args.in.data;
    const int64_t total = (int64_t)args.total_bytes;
    const int64_t off = (int64_t)offs[i];
    // (the host may have BELIEVED the tape's byte total from its previous call on the same tape: the thread that owns the tape's end checks)
    if (i == args.in.count && off != total) atomicMax(args.invalid, kUtf8SizesChanged);
    if (off >= total) {  // the tape's end (also every trailing empty string)
        uint64_t tiles = ((uint64_t)total + kUtf8Tile - 1) / kUtf8Tile;
        args.offsets[i] = tile_prefix[tiles];
        return;
    }
    const uint64_t tile = (uint64_t)off / kUtf8Tile, sub = ((uint64_t)off % kUtf8Tile) / 256;
    const int64_t sub_start = (int64_t)(tile * kUtf8Tile + sub * 256);
    uint32_t cnt = 0;
    int64_t q = sub_start;
    for (; q + 16 <= off; q += 16) {   // whole 16-byte units below the string: inside the tape, no masks (a dword at a time was a
        uint4 v;                       // chain of up to 63 dependent loads per string: 40 us for 100 K strings)
        __builtin_memcpy(&v, data + q, 16);
        cnt += __popc(lead_flags4(v.x)) + __popc(lead_flags4(v.y)) + __popc(lead_flags4(v.z)) + __popc(lead_flags4(v.w));
    }
    for (; q < off; q += 4) {
        int valid = off - q >= 4 ? 4 : (int)(off - q);
        cnt += __popc(lead_mask4(load_tape_dword(data, q, total), valid));
    }
    args.offsets[i] = tile_prefix[tile] + sub_prefix[tile * kUtf8Subs + sub] + cnt;
    // a non-empty string must start on a sequence boundary
    if (i < args.in.count && (int64_t)offs[i + 1] > off && (data[off] & 0xC0u) == 0x80u)
        atomicCAS(args.invalid, 0u, (uint32_t)i + 1u);
}

#ifdef SWH_TEST_HOOKS   // count / scan / write staging, continued
template <typename Off>
__global__ __launch_bounds__(256) void k_utf8_string_offsets(Utf8Args args, const uint64_t *tile_prefix,
                                                             const uint32_t *sub_prefix, const int *balance) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && *balance != 0) atomicCAS(args.invalid, 0u, 0x7FFFFFFFu);  // stray continuation bytes somewhere
    if (i > args.in.count) return;
    utf8_string_offset<Off>(args, tile_prefix, sub_prefix, i);
}

#endif   // SWH_TEST_HOOKS
// The one-pass staging's second (and last) launch: the string offsets of BOTH tapes (blocks [0, a.blocks) are tape a's), and
// in the first block of each tape's range the balance of its tiles (k_utf8_balance's sum: stray continuation bytes somewhere).
struct Utf8FinishJob {
    Utf8Args args;
    const uint64_t *tile_prefix; const uint32_t *sub_prefix; const int *tile_balance;
    uint64_t tiles; uint32_t blocks;
};
template <typename Off>
__global__ __launch_bounds__(256) void k_utf8_finish(Utf8FinishJob job_a, Utf8FinishJob job_b) {
    __shared__ int red[256];
    const bool second = blockIdx.x >= job_a.blocks;   // (workgroup-uniform: the selections below stay in scalar registers)
    const Utf8Args args = second ? job_b.args : job_a.args;
    const uint64_t *tile_prefix = second ? job_b.tile_prefix : job_a.tile_prefix;
    const uint32_t *sub_prefix = second ? job_b.sub_prefix : job_a.sub_prefix;
    const int *tile_balance = second ? job_b.tile_balance : job_a.tile_balance;
    const uint64_t tiles = second ? job_b.tiles : job_a.tiles;
    const uint32_t block = second ? blockIdx.x - job_a.blocks : blockIdx.x;
    if (block == 0) {
        int sum = 0;
        for (uint64_t t = threadIdx.x; t < tiles; t += 256) sum += tile_balance[t];
        red[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0 && red[0] != 0) atomicCAS(args.invalid, 0u, 0x7FFFFFFFu);
    }
    const uint64_t i = (uint64_t)block * 256 + threadIdx.x;
    if (i > args.in.count) return;
    if (tiles == 0) { args.offsets[i] = 0; return; }   // an empty tape: every string is empty
    utf8_string_offset<Off>(args, tile_prefix, sub_prefix, i);
}

#ifdef SWH_TEST_HOOKS   // count / scan / write staging, continued
// Exclusive scan of u32 counts into u64 offsets (count+1 entries). Three small kernels.
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void k_scan_block_sums(const uint32_t *counts, uint64_t n,
                                                                unsigned long long *block_sums) {
    __shared__ unsigned long long red[kScanBlock];
    uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
    red[threadIdx.x] = i < n ? counts[i] : 0;
    __syncthreads();
    for (int off = kScanBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(kScanBlock) void k_scan_top(unsigned long long *block_sums, uint32_t nblocks) {
    // serial-in-chunks exclusive scan by one block; nblocks is small (count / 1024)
    __shared__ unsigned long long carry;
    __shared__ unsigned long long buf[kScanBlock];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kScanBlock) {
        uint32_t i = base + threadIdx.x;
        unsigned long long v = i < nblocks ? block_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {
            unsigned long long t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry += buf[kScanBlock - 1];
        __syncthreads();
    }
}
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const uint32_t *counts, uint64_t n,
                                                           const unsigned long long *block_sums, uint64_t *offsets) {
    __shared__ unsigned long long buf[kScanBlock];
    uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
    unsigned long long v = i < n ? counts[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        unsigned long long t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long excl = block_sums[blockIdx.x] + buf[threadIdx.x] - v;
    if (i < n) offsets[i] = excl;
    if (i == n - 1) offsets[n] = excl + v;
}

// The same exclusive scan in ONE launch for up to 64 K counts (a 0.5 GB tape): each of 1024 threads sums a run of
// consecutive counts, the run totals are scanned across the block (six DPP adds per wave + 16 wave totals), and each
// thread writes its run's prefixes. Three launches and their gaps cost more than this kernel at these sizes.
constexpr uint64_t kScanOneMax = 64 * 1024;
__global__ __launch_bounds__(kScanBlock) void k_scan_one(const uint32_t *counts, uint64_t n, uint64_t *offsets) {
    __shared__ unsigned long long wave_tot[kScanBlock / 64];
    const uint32_t per = (uint32_t)((n + kScanBlock - 1) / kScanBlock);   // <= 64
    const uint64_t first = (uint64_t)threadIdx.x * per;
    unsigned long long sum = 0;
    for (uint32_t k = 0; k < per; ++k) sum += first + k < n ? counts[first + k] : 0;
    // inclusive scan of the run totals: within the wave by shuffles (64-bit), across waves through LDS
    unsigned long long incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long up = __shfl_up(incl, off);
        if ((int)(threadIdx.x & 63) >= off) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += wave_tot[w];
    unsigned long long run = base + incl - sum;
    for (uint32_t k = 0; k < per; ++k) {
        if (first + k < n) { offsets[first + k] = run; run += counts[first + k]; }
    }
    if (threadIdx.x == kScanBlock - 1) {
        unsigned long long total = 0;
        for (uint32_t w = 0; w < kScanBlock / 64; ++w) total += wave_tot[w];
        offsets[n] = total;
    }
}

#endif   // SWH_TEST_HOOKS
// STRINGWARS_AMD_UTF8_SCAN=split forces the count / scan / write kernels with the three-kernel scan, =scan the same with
// the one-launch scan (what round 2 ran; tests and comparisons use them). 0: the one-pass staging.
static int utf8_scan_mode() {
    static const int mode = [] {
        const char *e = test_hook("STRINGWARS_AMD_UTF8_SCAN");
        return !e ? 0 : (!strcmp(e, "split") ? 2 : (!strcmp(e, "scan") ? 1 : 0));
    }();
    return mode;
}
bool utf8_one_pass() { return utf8_scan_mode() == 0; }

// One-pass staging of one or two tapes (`b` may be null) in TWO launches: k_utf8_tile_decode over the tiles of both, then
// k_utf8_finish over the strings of both. `a.invalid` (= `b->invalid`) points at four zeroed words: [0] the invalid-UTF-8 marker,
// [32 (1 + a.slot)] the tile ticket (kUtf8FlagWords zeroed words in all). Large tapes are staged one launch pair per tape on two streams (api.hip: measured faster than
// one launch over both, whatever the order of the tiles); the caller opens the epoch once (utf8_status_open) and the second launch passes the first's tile count as `first_word`. The look-back words live in a buffer of the scope that nothing else writes, tagged with a per-call
// epoch: no clearing between calls (only when the 16-bit epoch wraps, or the buffer grows).
// Room for `words` look-back words (what a call's tapes have tiles); growing the buffer waits for the device, so a call that
// stages its tapes with two launches on two streams reserves for both before the first.
static void utf8_status_reserve(Scope *scope, uint64_t words) {
    if (words <= scope->utf8_status_cap) return;
    if (scope->utf8_status) SWH_HIP_CHECK(hipFree(scope->utf8_status));   // (hipFree waits for the device)
    scope->utf8_status = nullptr; scope->utf8_status_cap = 0;
    const uint64_t want = words + words / 4 + 1024;
    SWH_HIP_CHECK(hipMalloc((void **)&scope->utf8_status, want * sizeof(unsigned long long)));
    scope->utf8_status_cap = want;
    SWH_HIP_CHECK(hipMemsetAsync(scope->utf8_status, 0, want * sizeof(unsigned long long), scope->stream));
    // test hook STRINGWARS_AMD_UTF8_EPOCH=n: the first epoch of a fresh buffer (default 1), to reach the wrap-around in a few calls
    static const uint32_t first_epoch = [] {
        const char *e = test_hook("STRINGWARS_AMD_UTF8_EPOCH");
        const long v = e ? atol(e) : 1;
        return (uint32_t)(v >= 1 && v <= 0xFFFF ? v : 1);
    }();
    scope->utf8_epoch = first_epoch - 1;
}

// Opens a call's epoch of look-back words on scope->stream (room for `words`; every 65 535 calls, and after the buffer grew, the
// words are cleared). A call that stages its tapes on two streams opens BEFORE it forks: the clearing must not race the fork.
void utf8_status_open(Scope *scope, uint64_t words) {
    utf8_status_reserve(scope, words);
    if (++scope->utf8_epoch > 0xFFFFu) {
        SWH_HIP_CHECK(hipMemsetAsync(scope->utf8_status, 0, scope->utf8_status_cap * sizeof(unsigned long long), scope->stream));
        scope->utf8_epoch = 1;
    }
}

void launch_utf8_decode_pair(Scope *scope, const Utf8Args &a, const Utf8Args *b, uint64_t first_word, bool opened) {
    hipStream_t stream = scope->stream;
    auto carve = [](const Utf8Args &args, Utf8TileJob &tj, Utf8FinishJob &fj) {
        const uint64_t total = args.total_bytes, tiles = (total + kUtf8Tile - 1) / kUtf8Tile;
        uint32_t *tile_counts = args.counts;   // (the carving of launch_utf8_decode: the count / scan kernels' areas stay unused)
        uint32_t *sub_prefix = tile_counts + ((tiles + 2) & ~1ull);
        uint64_t *tile_prefix = (uint64_t *)(sub_prefix + kUtf8Subs * tiles + 2 - ((kUtf8Subs * tiles) & 1));
        unsigned long long *block_sums = (unsigned long long *)(tile_prefix + tiles + 2);
        int *tile_balance = (int *)(block_sums + (tiles + 1023) / 1024 + 4);
        tj = Utf8TileJob{(const uint8_t *)args.in.data, total, tiles, nullptr, tile_prefix, sub_prefix, args.symbols, tile_balance};
        fj = Utf8FinishJob{args, tile_prefix, sub_prefix, tile_balance, tiles, (uint32_t)((args.in.count + 1 + 255) / 256)};
    };
    Utf8TileJob ta{}, tb{};
    Utf8FinishJob fa{}, fb{};
    carve(a, ta, fa);
    if (b) carve(*b, tb, fb);
    const uint64_t words = ta.tiles + tb.tiles;
    if (!opened) utf8_status_open(scope, first_word + words);   // (`opened`: the caller did, for both of its launches)
    ta.status = scope->utf8_status + first_word;
    tb.status = ta.status + ta.tiles;
    if (words) {
        StampGuard guard(scope, "utf8_tile_decode");
        hipLaunchKernelGGL(k_utf8_tile_decode, dim3((uint32_t)words), dim3(256), 0, stream, ta, tb, a.invalid + 32 * (1 + a.slot), a.invalid, scope->utf8_epoch);
    }
    {
        StampGuard guard(scope, "utf8_offsets");
        if (a.off64) hipLaunchKernelGGL(k_utf8_finish<uint64_t>, dim3(fa.blocks + fb.blocks), dim3(256), 0, stream, fa, fb);
        else hipLaunchKernelGGL(k_utf8_finish<uint32_t>, dim3(fa.blocks + fb.blocks), dim3(256), 0, stream, fa, fb);
    }
    SWH_HIP_CHECK(hipGetLastError());
}

void launch_utf8_decode(Scope *scope, const Utf8Args &args) {
    if (utf8_one_pass()) { launch_utf8_decode_pair(scope, args, nullptr, 0, false); return; }
#ifdef SWH_TEST_HOOKS
    hipStream_t stream = scope->stream;
    const uint64_t n = args.in.count, total = args.total_bytes;
    const uint64_t tiles = (total + kUtf8Tile - 1) / kUtf8Tile;
    // scratch carved from `counts`: tile_counts[tiles+1] | sub_prefix[kUtf8Subs*tiles] | tile_prefix u64[tiles+1] | block sums
    uint32_t *tile_counts = args.counts;
    uint32_t *sub_prefix = tile_counts + ((tiles + 2) & ~1ull);
    uint64_t *tile_prefix = (uint64_t *)(sub_prefix + kUtf8Subs * tiles + 2 - ((kUtf8Subs * tiles) & 1));
    unsigned long long *block_sums = (unsigned long long *)(tile_prefix + tiles + 2);
    int *tile_balance = (int *)(block_sums + (tiles + 1023) / 1024 + 4);
    if (tiles) {
        {
            StampGuard guard(scope, "utf8_tile_count");
            hipLaunchKernelGGL(k_utf8_tile_count, dim3((uint32_t)tiles), dim3(256), 0, stream, (const uint8_t *)args.in.data,
                               total, tile_counts, sub_prefix);
        }
        uint32_t nblocks = (uint32_t)((tiles + kScanBlock - 1) / kScanBlock);
        {
            StampGuard guard(scope, "utf8_scan");
            const bool split_scan = utf8_scan_mode() == 2;   // (also what tapes beyond 0.5 GB take on this path)
            if (tiles <= kScanOneMax && !split_scan) {
                hipLaunchKernelGGL(k_scan_one, dim3(1), dim3(kScanBlock), 0, stream, tile_counts, tiles, tile_prefix);
            } else {
                hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(kScanBlock), 0, stream, tile_counts, tiles, block_sums);
                hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, stream, block_sums, nblocks);
                hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(kScanBlock), 0, stream, tile_counts, tiles, block_sums,
                                   tile_prefix);
            }
        }
        {
            StampGuard guard(scope, "utf8_tile_write");
            hipLaunchKernelGGL(k_utf8_tile_write, dim3((uint32_t)tiles), dim3(256), 0, stream, (const uint8_t *)args.in.data,
                               total, tile_prefix, args.symbols, args.invalid, tile_balance);
            hipLaunchKernelGGL(k_utf8_balance, dim3(64), dim3(256), 0, stream, tile_balance, tiles, (int *)(args.invalid + 1 + args.slot));
        }
    } else {
        SWH_HIP_CHECK(hipMemsetAsync(tile_prefix, 0, sizeof(uint64_t), stream));
    }
    {
        StampGuard guard(scope, "utf8_offsets");
        uint32_t blocks = (uint32_t)((n + 1 + 255) / 256);
        if (args.off64) hipLaunchKernelGGL(k_utf8_string_offsets<uint64_t>, dim3(blocks), dim3(256), 0, stream, args, tile_prefix, sub_prefix, (const int *)(args.invalid + 1 + args.slot));
        else hipLaunchKernelGGL(k_utf8_string_offsets<uint32_t>, dim3(blocks), dim3(256), 0, stream, args, tile_prefix, sub_prefix, (const int *)(args.invalid + 1 + args.slot));
    }
    SWH_HIP_CHECK(hipGetLastError());
#endif   // SWH_TEST_HOOKS
}

// ------------------------------------------------------------------------------------------------------------
// Staging STRING BY STRING (round 5): one wave per string, rounds of 1 KB, no ticket, no look-back, no workgroup barrier.
// What made the flat kernels above wait -- a tile needs the code points of every tile before it to know where its own go -- is a
// property of the OUTPUT LAYOUT, not of UTF-8: here string i's code points go to symbols[offsets[i] ...], the place its BYTES had,
// so a string is placed by its own byte offset alone and counted by the wave that decodes it. The tape that results has gaps
// (a string of b bytes holds <= b code points): its `offsets` are (first, end) pairs, TapeRef::gap = 1 (common.hpp), and every
// kernel that reads code points takes its extents through pair_extent / tape_total, which know. Validation is per string too:
// every lead byte checks its own sequence (k_utf8_tile_decode's table), a sequence may not reach past its string's end, and the
// bytes the sequences claim must add up to the string's length -- claimed ranges are disjoint, so equality means every byte is a
// lead byte or claimed once (a string that starts with a continuation byte, or holds a stray one, comes up short).
// For tapes of lines and longer (api.hip: a mean string of >= kUtf8StringsMeanBytes bytes); a string beyond kUtf8StringLongest
// bytes would keep one wave busy for milliseconds: the kernel says so (kUtf8StringTooLong) and the host stages the flat way.
// ------------------------------------------------------------------------------------------------------------
constexpr int kUtf8StrRound = 1024;       // bytes per full round: sixteen per lane
constexpr int kUtf8StrWaves = 4;
struct Utf8StringsLds {
    Utf8Lead table[256];
    uint32_t win[kUtf8StrWaves][kUtf8StrRound];   // the round's sequences in rank order: the four bytes from each lead byte on
};

// D dwords from `pos` on; bytes past the tape read as zero (a branch only the tape's last bytes take)
template <int D>
__device__ __forceinline__ void utf8_load_dwords(const uint8_t *data, uint64_t pos, uint64_t total, uint32_t (&w)[D]) {
    if (pos + 4 * D <= total) { __builtin_memcpy(w, data + pos, 4 * D); return; }
#pragma unroll
    for (int q = 0; q < D; ++q) w[q] = 0;
    for (int j = 0; j < 4 * D; ++j)
        if (pos + (uint64_t)j < total) w[j >> 2] |= (uint32_t)data[pos + j] << (8 * (j & 3));
}

// One round of one string: the 256 D bytes from `pos` on, 4 D per lane (D = 4 for every round but a string's last, whose D is what
// the rest needs: no lane works on bytes the string does not have). `rem`: bytes of the string from `pos` on.
//   1. every lane flags the lead bytes of its dwords (bytes behind the string's end are zeroed first -- a sequence that reaches past
//      the end then fails its continuation check -- and are no lead bytes), a wave scan ranks them;
//   2. every lead byte's WINDOW -- the four bytes from it on, cut from the lane's registers (the dword after its last comes from the
//      next lane by DPP) -- goes to win[rank]: the round's sequences in order, no list of positions, no second look at the bytes;
//   3. lane k decodes sequences k, k + 64, ... (the table of k_utf8_tile_decode: one 16-byte entry per first byte) and stores code
//      points k, k + 64, ...: full lines.
template <int D>
__device__ __forceinline__ uint32_t utf8_string_round(const Utf8Lead *table, uint32_t *win, const uint8_t *data, uint64_t pos, uint64_t total, uint64_t rem,
                                                      uint32_t *out, uint32_t &wrong_pair, uint32_t &wrong_tops, uint32_t &shapes, uint32_t &high) {
    const int lane = threadIdx.x & 63;
    uint32_t dw[D + 1];
    {
        uint32_t got[D];
        utf8_load_dwords<D>(data, pos + (uint64_t)(4 * D) * (uint32_t)lane, total, got);
#pragma unroll
        for (int q = 0; q < D; ++q) dw[q] = got[q];
    }
    uint32_t after[1] = {0u};
    if (lane == 63) utf8_load_dwords<1>(data, pos + 256u * D, total, after);   // look-ahead of the round's last sequences
    uint32_t tail_mask[D];   // (string's last round) which bytes of my dwords are the string's
#pragma unroll
    for (int q = 0; q < D; ++q) tail_mask[q] = 0xFFFFFFFFu;
    if (rem < 256u * D) {   // (wave-uniform) the string ends inside this round: what lies behind its end reads as zero
        const uint32_t mine = 4u * D * (uint32_t)lane;
        const uint32_t have = (uint32_t)rem > mine ? (uint32_t)rem - mine : 0u;   // bytes of the string in my dwords (>= 4 D: all of them)
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const uint32_t nv = have > 4u * q ? (have - 4u * q < 4u ? have - 4u * q : 4u) : 0u;
            tail_mask[q] = (uint32_t)~(~0ull << (8u * nv));
            dw[q] &= tail_mask[q];
        }
    }
    {   // the word behind the round holds bytes of the string only as far as the string goes (wave-uniform arithmetic, lane 63's word)
        const uint32_t beyond = rem > 256u * D ? (rem - 256u * D < 4u ? (uint32_t)(rem - 256u * D) : 4u) : 0u;
        after[0] &= (uint32_t)~(~0ull << (8u * beyond));
    }
    // the string's lead bytes are counted on the masked flags (a zero byte behind the end is no lead byte of the string) ...
    uint32_t mine_count = 0;
#pragma unroll
    for (int q = 0; q < D; ++q) {
        high |= dw[q];
        mine_count += (uint32_t)__popc(lead_flags4(dw[q]) & tail_mask[q]);
    }
    const uint32_t incl = wave_inclusive_sum_u32(mine_count);
    const uint32_t total_leads = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // the dword after my last: the next lane's first (wave_shl:1; lane 63 keeps the word behind the round)
    dw[D] = (uint32_t)__builtin_amdgcn_update_dpp((int)after[0], (int)dw[0], 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    // ... and ranked by position. The zero bytes behind the string's end pass the test below, but they come AFTER every byte of the
    // string: their windows land behind the string's own (win[] holds 1024) and are never decoded. The test itself is one compare on
    // the byte (a lead byte is anything but 10xxxxxx: as a signed byte, >= -64), no flag word to pick bits from.
    // (Written out: hipcc turns the byte test into v_and + v_cmp and moves the rank through a second register -- six vector
    // instructions per byte slot where these are three. s_and_saveexec right behind the v_cmp that feeds it is what hipcc emits itself.)
    uint32_t rank = incl - mine_count;
    const uint32_t win_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)win);
    const int least_lead = -64;
#define SWH_UTF8_SLOT(BYTE, WINDOW)                                                                                          \
    do {                                                                                                                     \
        unsigned long long saved__;                                                                                          \
        uint32_t addr__;                                                                                                     \
        asm volatile("v_cmp_le_i32_sdwa vcc, %[k], sext(%[dw]) src0_sel:DWORD src1_sel:BYTE_" #BYTE "\n\t"                  \
                     "s_and_saveexec_b64 %[sv], vcc\n\t"                                                                    \
                     "v_lshl_add_u32 %[ad], %[rk], 2, %[bs]\n\t"                                                            \
                     "ds_write_b32 %[ad], %[w]\n\t"                                                                         \
                     "v_add_u32 %[rk], 1, %[rk]\n\t"                                                                        \
                     "s_or_b64 exec, exec, %[sv]"                                                                            \
                     : [rk] "+v"(rank), [sv] "=&s"(saved__), [ad] "=&v"(addr__)                                              \
                     : [k] "v"(least_lead), [dw] "v"(dw[q]), [w] "v"(WINDOW), [bs] "s"(win_base)                             \
                     : "vcc", "memory");                                                                                     \
    } while (0)
#pragma unroll
    for (int q = 0; q < D; ++q) {
        const uint32_t w1 = __builtin_amdgcn_alignbit(dw[q + 1], dw[q], 8), w2 = __builtin_amdgcn_alignbit(dw[q + 1], dw[q], 16);
        const uint32_t w3 = __builtin_amdgcn_alignbit(dw[q + 1], dw[q], 24);
        SWH_UTF8_SLOT(0, dw[q]);
        SWH_UTF8_SLOT(1, w1);
        SWH_UTF8_SLOT(2, w2);
        SWH_UTF8_SLOT(3, w3);
    }
#undef SWH_UTF8_SLOT
    wave_lds_fence();   // the lanes read each other's windows
    for (uint32_t k = (uint32_t)lane; k < total_leads; k += 64) {
        const uint32_t sq = win[k];
        const uint4 t = *(const uint4 *)&table[sq & 0xFFu];   // x lead, y shape, z cont, w range (see k_utf8_tile_decode)
        uint32_t inside;
        asm volatile("v_med3_u16 %0, %1, %2, %2 op_sel:[0,0,1,0]" : "=v"(inside) : "v"(sq), "v"(t.w));
        wrong_pair |= inside ^ sq;
        wrong_tops |= (sq ^ 0x80800000u) & t.z;
        shapes += t.y;
        const uint32_t after3 = (((sq >> 8 & 0x3Fu) << 6 | (sq >> 16 & 0x3Fu)) << 6) | (sq >> 24 & 0x3Fu);
        out[k] = t.x | __builtin_amdgcn_ubfe(after3, t.y, t.z);
    }
    wave_lds_fence();   // the next round rewrites the windows
    return total_leads;
}

template <typename Off>
__global__ __launch_bounds__(kUtf8StrWaves * 64) void k_utf8_strings(Utf8StringsJob ja, Utf8StringsJob jb, uint32_t *invalid) {
    __shared__ Utf8StringsLds lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: a string's extent is a scalar load)
    *(uint4 *)&lds.table[threadIdx.x] = *(const uint4 *)&kUtf8Leads.e[threadIdx.x];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        // the host may have BELIEVED the tapes' byte totals (api.hip): the tape's end is compared here; and the capacity slot of the extents
        const Utf8StringsJob &j = threadIdx.x ? jb : ja;
        if (j.offsets) {
            if ((uint64_t)((const Off *)j.offsets)[j.count] != j.total) atomicMax(invalid, kUtf8SizesChanged);
            j.extents[2 * j.count] = j.total;
        }
    }
    uint32_t *win = lds.win[wave];
    const uint64_t strings = ja.count + jb.count;
    const uint64_t waves_total = (uint64_t)gridDim.x * kUtf8StrWaves;
    uint32_t high_a = 0, high_b = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * kUtf8StrWaves + wave; s < strings; s += waves_total) {
        const bool second = s >= ja.count;   // (wave-uniform)
        const uint64_t i = second ? s - ja.count : s;
        const uint8_t *data = second ? jb.data : ja.data;
        const Off *offs = (const Off *)(second ? jb.offsets : ja.offsets);
        const uint64_t total = second ? jb.total : ja.total;
        uint32_t *symbols = second ? jb.symbols : ja.symbols;
        uint64_t *extents = second ? jb.extents : ja.extents;
        const uint64_t o = (uint64_t)offs[i], e = (uint64_t)offs[i + 1];
        if (e > total || o > e) {   // the tape is not what the host believed (or its offsets do not ascend): nothing of it is touched
            if (lane == 0) { atomicMax(invalid, kUtf8SizesChanged); extents[2 * i] = 0; extents[2 * i + 1] = 0; }
            continue;
        }
        const uint64_t len = e - o;
        if (len > kUtf8StringLongest) {
            if (lane == 0) { atomicMax(invalid, kUtf8StringTooLong); extents[2 * i] = o; extents[2 * i + 1] = o; }
            continue;
        }
        uint32_t done = 0;                 // code points of the rounds before this one
        uint32_t wrong_pair = 0, wrong_tops = 0, shapes = 0, high = 0;
        // rounds start AT the string (unaligned loads): no bytes in front of it to mask away; full rounds of 1 KB, then one round of
        // one to four dwords per lane for what is left
        uint64_t at0 = 0;
        for (; len - at0 > (uint64_t)kUtf8StrRound; at0 += kUtf8StrRound)
            done += utf8_string_round<4>(lds.table, win, data, o + at0, total, len - at0, symbols + o + done, wrong_pair, wrong_tops, shapes, high);
        const uint64_t rest = len - at0;
        if (rest > 768) done += utf8_string_round<4>(lds.table, win, data, o + at0, total, rest, symbols + o + done, wrong_pair, wrong_tops, shapes, high);
        else if (rest > 512) done += utf8_string_round<3>(lds.table, win, data, o + at0, total, rest, symbols + o + done, wrong_pair, wrong_tops, shapes, high);
        else if (rest > 256) done += utf8_string_round<2>(lds.table, win, data, o + at0, total, rest, symbols + o + done, wrong_pair, wrong_tops, shapes, high);
        else if (rest > 0) done += utf8_string_round<1>(lds.table, win, data, o + at0, total, rest, symbols + o + done, wrong_pair, wrong_tops, shapes, high);
        if (second) high_b |= high; else high_a |= high;
        // what the sequences claim against what the string holds; any lane's complaint is the string's
        const uint32_t claimed = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum_u32(shapes >> 16), 63);
        const bool bad_lane = ((wrong_pair & 0xFFFFu) | (wrong_tops & 0xC0C00000u)) != 0;
        const bool bad = __ballot(bad_lane) != 0 || claimed != (uint32_t)len;
        if (lane == 0) {
            extents[2 * i] = o;
            extents[2 * i + 1] = o + done;
            if (bad) atomicCAS(invalid, 0u, (uint32_t)i + 1u);   // (the marker: the string's index; k_utf8_tile_decode's is a word of the tape)
        }
    }
    // a byte above 0x7F anywhere: the tape is not pure ASCII (a plain store, every such wave writes the same 1)
    if (__ballot((high_a & 0x80808080u) != 0) != 0 && lane == 0) invalid[kUtf8AsciiWord] = 1u;
    if (__ballot((high_b & 0x80808080u) != 0) != 0 && lane == 0) invalid[kUtf8AsciiWord + 1] = 1u;
}

void launch_utf8_strings(Scope *scope, const Utf8StringsJob &a, const Utf8StringsJob *b, uint32_t off64, uint32_t *invalid) {
    Utf8StringsJob none{};
    const Utf8StringsJob &jb = b ? *b : none;
    const uint64_t strings = a.count + jb.count;
    // eight workgroups per compute unit hold every wave slot; fewer strings than that: a wave each
    const uint64_t most = (uint64_t)scope->compute_units * 8;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(most, (strings + kUtf8StrWaves - 1) / kUtf8StrWaves));
    StampGuard guard(scope, "utf8_strings");
    if (off64) hipLaunchKernelGGL(k_utf8_strings<uint64_t>, dim3(blocks), dim3(kUtf8StrWaves * 64), 0, scope->stream, a, jb, invalid);
    else hipLaunchKernelGGL(k_utf8_strings<uint32_t>, dim3(blocks), dim3(kUtf8StrWaves * 64), 0, scope->stream, a, jb, invalid);
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
