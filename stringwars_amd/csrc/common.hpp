// common.hpp -- shared host/device definitions of the stringwars_amd HIP backend (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/stringwars_amd.h"

namespace swh {

// ------------------------------------------------------------------------------------------------
// Problem description handed to every kernel. One `Job` = one engine call.
// ------------------------------------------------------------------------------------------------
// Symbols are either raw tape bytes (Sym = uint8_t) or decoded Unicode scalar values
// (Sym = uint32_t, produced by the UTF-8 staging kernel into scope scratch). Offsets are the
// tape's own u32/u64 (bytes) or the scratch tape's u64 (code points).
struct TapeRef {
    const void *data;     // device pointer, Sym elements
    const void *offsets;  // device pointer, count+1 entries (gap: 2 count + 1, see below)
    uint64_t count;
    // gap = 1: a code-point tape staged string by string (prepass.hip: k_utf8_strings) -- string i's symbols start where its BYTES
    // started (no prefix over the strings before it is needed to place them), so the strings do not abut and `offsets` holds a
    // (first, end) pair per string: string i = [offsets[2 i], offsets[2 i + 1]), offsets[2 count] = the symbol buffer's capacity.
    uint32_t gap = 0;
};
// offsets[count]: the end of the tape's last string = how many symbols the data buffer holds (gap tapes: its capacity)
template <typename Off>
__device__ __forceinline__ uint64_t tape_total(const TapeRef &t) { return (uint64_t)((const Off *)t.offsets)[t.count << t.gap]; }
__device__ __forceinline__ uint64_t tape_total(const TapeRef &t, uint32_t off64) {
    return off64 ? tape_total<uint64_t>(t) : tape_total<uint32_t>(t);
}

struct Job {
    TapeRef a, b;
    uint64_t pairs;       // number of (a_i, b_j) pairs to score
    uint64_t b_count;     // for cross-product mode: j = p % b_count, i = p / b_count
    uint32_t cross;       // 0 = pairwise (i == j == p), 1 = cross-product
    uint32_t bound;       // SWH_UNBOUNDED or k: out = min(d, k+1)      (Levenshtein only)
    char *out;            // device pointer
    uint64_t out_stride;  // bytes between consecutive pair results (pairwise) / elements in a row
    uint64_t row_stride;  // bytes between rows (cross)
    uint32_t out_elem64;  // 0: 32-bit results, 1: 64-bit results (size_t / ptrdiff_t)
    uint32_t negate;      // results are stored as -score (min-plus distances run on the max-plus core)
    uint32_t div_magic, div_shift;   // cross-product mode: reciprocal of b_count (cross_divider / cross_split)
};

// Cross-product mode: pair p is (query p / b_count, candidate p % b_count). p < 2^32 (api.hip refuses larger batches) and
// b_count < 2^32, so the split is a 32-bit division by an invariant: multiply-high by a precomputed reciprocal (the
// round-up method of Granlund & Montgomery; five instructions against ~50 for the 64-bit division the compiler emits).
__host__ __device__ __forceinline__ void cross_divider(uint32_t d, uint32_t &magic, uint32_t &shift) {
    // d >= 1. shift = ceil(log2 d); magic = floor(2^32 * (2^shift - d) / d) + 1
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) ++l;
    magic = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    shift = l;
}
__device__ __forceinline__ void cross_split(const Job &job, uint64_t p, uint64_t &ia, uint64_t &ib) {
    const uint32_t n = (uint32_t)p;
    const uint32_t t = __umulhi(job.div_magic, n);
    const uint32_t q = job.div_shift ? (uint32_t)((((n - t) >> 1) + t) >> (job.div_shift - 1)) : n;   // shift 0: one candidate
    ia = q;
    ib = n - q * (uint32_t)job.b_count;
}

template <typename Off>
__device__ __forceinline__ void pair_extent(const Job &job, uint64_t p, uint64_t &a0, uint32_t &la, uint64_t &b0,
                                            uint32_t &lb) {
    uint64_t ia = p, ib = p;
    if (job.cross) cross_split(job, p, ia, ib);
    const Off *oa = (const Off *)job.a.offsets + (ia << job.a.gap), *ob = (const Off *)job.b.offsets + (ib << job.b.gap);
    Off x0 = oa[0], x1 = oa[1], y0 = ob[0], y1 = ob[1];
    a0 = (uint64_t)x0; la = (uint32_t)(x1 - x0);
    b0 = (uint64_t)y0; lb = (uint32_t)(y1 - y0);
}

// A result leaves as an agent-scope atomic store (`global_store ... sc1`): written through to where every XCD -- and a copy engine --
// reads it, where a plain store to coarse-grained device memory stays in the writing XCD's L2 until the kernel's end. The kernels
// that report a call summary promise "every result is out" with it (report_call_summary), and a synchronous call returns on that
// word instead of on the stream (api.hip: wait_for_summary); results are written once, so nothing is lost by not caching them.
__device__ __forceinline__ void store_out(char *dst, bool elem64, int64_t value) {
    if (elem64) __hip_atomic_store((long long *)dst, (long long)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store((int *)dst, (int)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_result(const Job &job, uint64_t p, int64_t value) {
    char *dst;
    if (job.cross) {
        uint64_t ia, ib;
        cross_split(job, p, ia, ib);
        dst = job.out + ia * job.row_stride + ib * (job.out_elem64 ? 8 : 4);
    } else {
        dst = job.out + p * job.out_stride;
    }
    store_out(dst, job.out_elem64 != 0, value);
}

// What an earlier kernel of the same call stored for pair p (the second stage of a doubling call asks whether the first one settled it).
__device__ __forceinline__ int64_t load_result(const Job &job, uint64_t p) {
    const char *src;
    if (job.cross) {
        uint64_t ia, ib;
        cross_split(job, p, ia, ib);
        src = job.out + ia * job.row_stride + ib * (job.out_elem64 ? 8 : 4);
    } else {
        src = job.out + p * job.out_stride;
    }
    return job.out_elem64 ? (int64_t)*(const long long *)src : (int64_t)*(const int *)src;
}

// Lanes of one wave that hand data to each other through LDS still need a fence: the compiler reasons per
// thread, so it may sink a lane's ds_write below reads that only OTHER lanes' writes alias (seen on gfx950:
// a staging store moved under the loads that consume it). Wavefront scope costs no cache traffic.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wave that is ahead steps back: priority 3 in the first quarter of its share of the work, 0 in the last. The arbiter's tie-break
// is "oldest first", so with equal shares dealt statically the oldest waves of a SIMD run ahead and leave the youngest to finish
// alone -- a chain of dependent instructions on a SIMD that could issue several times as often. With the priority falling along
// the way the waves behind catch up and the SIMDs stay full to the end (banded.hip: C3 0.319 -> 0.306 ms).
__device__ __forceinline__ void fair_priority(uint64_t done, uint64_t share) {
    const uint64_t quarter = share ? (4u * done) / share : 0u;
    if (quarter == 0) __builtin_amdgcn_s_setprio(3);
    else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
    else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// Maximum over the 64 lanes of a wave, broadcast: four row_shr steps inside each row of 16, row_bcast 15 / 31 across
// rows, then lane 63 holds the maximum.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define SWH_MAX_DPP(CTRL, ROWS, BOUND)                                                              \
    do {                                                                                            \
        const uint32_t o__ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, BOUND); \
        v = o__ > v ? o__ : v;                                                                      \
    } while (0)
    SWH_MAX_DPP(0x111, 0xf, true); SWH_MAX_DPP(0x112, 0xf, true); SWH_MAX_DPP(0x114, 0xf, true); SWH_MAX_DPP(0x118, 0xf, true);
    SWH_MAX_DPP(0x142, 0xa, false); SWH_MAX_DPP(0x143, 0xc, false);
#undef SWH_MAX_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (row_shr 1/2/4/8 inside each row of 16, then
// row_bcast 15 / 31 carry the row totals forward).
__device__ __forceinline__ uint32_t wave_inclusive_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// Bit-parallel orientation: the "pattern" string supplies the 32-row blocks (lanes), the "text" string the steps.
// Work is blocks x (text + blocks - 1) block-steps, so the cheaper assignment wins -- usually the LONGER string
// as pattern when both need the same number of blocks (fewer steps), the shorter one when it saves a block.
// Must be used identically by the plan key (prepass.hip) and the kernel (bitparallel.hip).
__host__ __device__ __forceinline__ bool bp_pattern_is_a(uint32_t la, uint32_t lb) {
    const uint32_t ga = (la + 31) >> 5, gb = (lb + 31) >> 5;
    if (ga > 64 || gb > 64) {
        // Beyond the 64-block systolic array a pattern runs as passes of 64 blocks, each a walk over the whole text, one pair per wave:
        // wave-steps = passes x text (+ the fill). With the same number of passes either way the LONGER string is the cheaper pattern
        // (2600 x 3500 symbols: two passes over 2600 columns instead of over 3500, and 110 of 128 lane-slots busy instead of 82) -- until
        // round 5 the string of fewer blocks was taken, which is right only when it saves a pass or fits the array.
        auto wave_steps = [](uint32_t g, uint32_t text) -> uint64_t {
            if (g > 64) return (uint64_t)((g + 63) >> 6) * text + g;
            return g ? ((uint64_t)text + g - 1) / (64 / g) : 0;   // floor(64 / g) pairs share a wave of k_bitparallel
        };
        return wave_steps(ga, lb) <= wave_steps(gb, la);
    }
    const uint64_t ca = (uint64_t)ga * (lb + ga - 1), cb = (uint64_t)gb * (la + gb - 1);
    return ca <= cb;
}

// What the banded kernel (banded.hip) costs per column of 64 pairs, in wave instructions, against 28 (bytes) / 40 (code points)
// per 32-row block for the bit-parallel kernels: 3 per window symbol + 45 for bounds up to 63 (one 64-bit window, the figures
// plan_key() has always used). A two-word window (bounds 64 .. 127) adds its longer recurrence and its items of 16 pairs:
// + 85, checked on config C3's lines (tools/bench_bounds.py: at 16 blocks, k = 64 / 100 / 127 take 0.70 / 0.80 / 0.87 of the
// unbounded code-point kernel's time; + 170 sent the shorter half of the lines to that kernel at k = 127 and the call -- two
// kernels, the second behind a plan read-back -- took 1.07 ms instead of 0.91). Must be used identically by plan_key() and
// api.hip's route.
constexpr uint32_t kBandMaxBound = 127;
__host__ __device__ __forceinline__ uint32_t band_cost(uint32_t bound) {
    const uint32_t base = 3 * (bound + 1) + 45;
    return bound <= 63 ? base : base + 85;
}

// Levenshtein results honour the cutoff convention out = min(d, bound + 1).
__device__ __forceinline__ uint32_t clamp_bound(uint32_t d, uint32_t bound) {
    return (bound != 0xFFFFFFFFu && d > bound) ? bound + 1 : d;
}

// ------------------------------------------------------------------------------------------------
// Work plan built on the device by the pre-pass (prepass.hip): pairs are counting-sorted by a
// length-class key so that every wave gets pairs of one class and similar row counts.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxClasses = 96;   // kernel classes (a class = one kernel configuration)
constexpr int kBuckets = 64;      // length buckets inside a class (sort granularity)
constexpr int kKeys = kMaxClasses * kBuckets;
static_assert(kKeys <= 65536, "plan keys are stored as u16 between the planning kernels");

struct Plan {
    uint32_t class_start[kMaxClasses + 1];  // exclusive prefix of pairs per class into `perm`
    uint32_t class_count[kMaxClasses];
    uint64_t cells;                          // sum len_s(a)*len_s(b): the reference's CUPS numerator
    uint64_t symbols;                        // sum len_s(a)+len_s(b)
    uint32_t max_la, max_lb;
    uint32_t invalid_utf8;                   // index+1 of the first pair with invalid UTF-8, else 0
    uint32_t short_pairs;                    // pairs with both sides <= 32 symbols (hint: run k_direct_short next time)
    uint32_t fused_failed;                   // k_plan_fused gave up at its grid barrier: the plan is empty, redo with the three passes
};

struct PlanPartial { unsigned long long cells, symbols; uint32_t max_la, max_lb, short_pairs, pad; };
static_assert(sizeof(PlanPartial) == 32, "report_call_summary reads a partial as four 64-bit words");
// What a call that ran without the planning pre-pass (tiled.hip, k_direct_short alone) reports back: written by the last
// workgroup to finish, straight into host-mapped memory, so the host has it after the stream synchronisation it does
// anyway -- no copy, no extra round trip. `violation`: some pair did not fit the kernel (the call must be redone on the
// planned path).
struct CallSummary {
    unsigned long long cells, symbols;
    uint32_t max_la, max_lb, short_pairs, violation;
    // Set with `violation` and cleared only by the host: a summary is overwritten by the next call's, and an asynchronous call's
    // summary is only READ when the host happens to synchronise right behind it (pipelined lanes carry two calls between
    // synchronisations, a sharded call four pieces). Asynchronous calls report into slot 1 of the scope's summary block and
    // swh_scope_synchronize looks at slot 1's `sticky`; synchronous calls (slot 0) handle `violation` on the spot.
    // `landed`: written (1) after everything else, by the last workgroup, once every workgroup's result stores were acknowledged:
    // the host clears it before a synchronous call's launch and returns when it reads 1 (api.hip: wait_for_summary).
    uint32_t sticky, landed;
};
static_assert(sizeof(CallSummary) == 40, "two summary slots share the scope's 256-byte host-mapped block");
// Tail of the kernels that run without the planning pre-pass, called by every thread of the workgroup; thread 0 passes the
// workgroup's sums (`pad` != 0: it met a pair it could not score). The last workgroup to get here folds all partials and
// writes the call's summary into host-mapped memory; the counter resets itself for the next launch.
struct SummaryLds {   // workgroup scratch of report_call_summary (the caller lends it: <= 16 waves per workgroup)
    unsigned long long rcells[16], rsyms[16];
    uint32_t rmaxa[16], rmaxb[16], rshort[16], rviol[16];
    uint32_t is_last;
};
// HARDWARE ASSUMPTION (gfx942 / gfx950): an agent-scope atomic store is `global_store ... sc1` -- written through to where every
// XCD sees it -- and is tracked by `vmcnt` like a load, so `s_waitcnt vmcnt(0)` in front of the counter's bump orders the
// row before the bump by COMPLETION. The HIP / LLVM memory model promises no happens-before edge for this (a relaxed store
// followed by a relaxed RMW); a target with a separate store counter (gfx10+: `vscnt`) would break it silently. The library is
// built for gfx950 only; any other device pass fails here instead of miscounting.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "report_call_summary / k_plan_fused order their hand-over by vmcnt completion: gfx942 / gfx950 only (see the comment above)"
#endif
__device__ __forceinline__ void report_call_summary(const PlanPartial &mine, PlanPartial *partials, uint32_t *done_counter,
                                                    CallSummary *summary, SummaryLds &lds) {
    auto &rcells = lds.rcells; auto &rsyms = lds.rsyms;
    auto &rmaxa = lds.rmaxa; auto &rmaxb = lds.rmaxb; auto &rshort = lds.rshort; auto &rviol = lds.rviol;
    // every wave's result stores are acknowledged before the workgroup counts as done (a workgroup barrier waits for LDS, not for them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // The row goes out as agent-scope atomic stores (written through to where every XCD sees them) and the counter is
        // bumped once they are acknowledged: ordering by completion, without `__threadfence()` -- a release/acquire pair at
        // agent scope writes back and invalidates the XCD's whole L2, here with the call's results freshly dirty in it, once
        // per workgroup. The last workgroup reads the rows with agent-scope atomic loads.
        unsigned long long *row = (unsigned long long *)&partials[blockIdx.x];
        __hip_atomic_store(row + 0, mine.cells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + 1, mine.symbols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + 2, (unsigned long long)mine.max_la | (unsigned long long)mine.max_lb << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + 3, (unsigned long long)mine.short_pairs | (unsigned long long)mine.pad << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds.is_last = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!lds.is_last) return;
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, viol = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += blockDim.x) {
        const unsigned long long *q = (const unsigned long long *)&partials[i];
        cells += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        syms += __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w2 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w3 = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t pa = (uint32_t)w2, pb = (uint32_t)(w2 >> 32);
        maxa = pa > maxa ? pa : maxa;
        maxb = pb > maxb ? pb : maxb;
        shorts += (uint32_t)w3;
        viol |= (uint32_t)(w3 >> 32);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cells += __shfl_xor(cells, off);
        syms += __shfl_xor(syms, off);
        shorts += __shfl_xor(shorts, off);
        viol |= __shfl_xor(viol, off);
        const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
        maxa = oa > maxa ? oa : maxa;
        maxb = ob > maxb ? ob : maxb;
    }
    const uint32_t wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { rcells[wave] = cells; rsyms[wave] = syms; rmaxa[wave] = maxa; rmaxb[wave] = maxb; rshort[wave] = shorts; rviol[wave] = viol; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < waves; ++w) {
            cells += rcells[w]; syms += rsyms[w]; shorts += rshort[w]; viol |= rviol[w];
            maxa = rmaxa[w] > maxa ? rmaxa[w] : maxa;
            maxb = rmaxb[w] > maxb ? rmaxb[w] : maxb;
        }
        __hip_atomic_store(&summary->cells, cells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&summary->symbols, syms, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&summary->max_la, maxa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&summary->max_lb, maxb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&summary->short_pairs, shorts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (non-zero: a pair was not scored. Bit 0: a string did not fit the kernel; bit 1: an item's candidates use more symbol classes
        // than the compacting alignment kernels hold -- the one reason that keeps a scope off those kernels, api.hip)
        __hip_atomic_store(&summary->violation, viol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (viol) __hip_atomic_store(&summary->sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the summary's words (and the counter's reset) before the word that says so
        __hip_atomic_store(&summary->landed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// partials[0, kMaxPartials) belong to k_plan_hist, [kMaxPartials, 2 kMaxPartials) to k_direct_short (see PrepassArgs::leftover)
constexpr int kMaxPartials = 4096;

// Class numbering --------------------------------------------------------------------------------
// 0                : trivial pairs (an empty side, or cutoff decided by lengths) -- finished in the pre-pass
// 1..64            : bit-parallel, G = class = number of 32-row blocks of the shorter string
// 65..72           : wavefront, 16 lanes per pair, W = class-64 columns per lane (cols <= 16*W)
// 73..84           : wavefront, 64 lanes per pair, W = kWideW[class-73]
// 85               : wavefront multi-pass (columns beyond 64*kWideW[last])
// 86               : banded (bounded, k <= 127) Hyyro window of one or two words, <= 64 pairs per wave
constexpr int kClassTrivial = 0;
constexpr int kClassBp0 = 1;
constexpr int kClassWf16 = 65;
constexpr int kClassWf64 = 73;
constexpr int kNumWideW = 12;
constexpr int kClassWfMulti = kClassWf64 + kNumWideW;
constexpr int kClassBanded = kClassWfMulti + 1;  // bounded unit-cost pairs on the sliding 64-bit band (banded.hip)
constexpr int kClassBpLong = kClassBanded + 1;   // bit-parallel, both strings > 2048 symbols: one pair per wave, 64 blocks per pass
__host__ __device__ constexpr int wide_w(int i) {
    constexpr int w[kNumWideW] = {3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 80, 96};
    return w[i];
}

enum PlanMode : uint32_t {
    kPlanBitParallel = 0,  // unit-cost Levenshtein on bytes: bit-parallel where it fits, else wavefront
    kPlanWavefront = 1,    // everything on the wavefront kernels
};

// ------------------------------------------------------------------------------------------------
// Scoring model of the wavefront core (max-plus; distances are negated scores).
// ------------------------------------------------------------------------------------------------
struct Scoring {
    int match, mismatch;   // uniform substitution (used when `matrix` is null)
    int open, extend;      // gap(k) = open + (k-1)*extend
    const int8_t *matrix;  // device pointer to 256x256 i8, row = a symbol, col = b symbol; or null
    const uint8_t *class_table;  // device: 32x32 i8 class costs then 256 B byte->class map, when the matrix has <= 32 classes
    uint32_t classes;            // how many of the 32 classes are in use (0: unknown, treat as 32)
    uint32_t step_span;          // class model: max |class cost| + |open| + |extend| -- neighbouring DP cells differ by no more (0: unknown)
    // 33 .. 128 symbol classes (mixed-case text, IUPAC codes + case: what a rust-bio style scoring closure distinguishes, bench.rs:746-752):
    // too many for the register cost rows of the 32-class model, but the column-profile kernel (nwprofile.hip) only needs a cost row per
    // class while it builds a pass's profile. Device: kWideClasses x kWideClasses i8 costs (biased like class_table), then the 256 B
    // byte -> class map; `classes` says how many are in use. class_table stays null: pairs too narrow for the profile kernel take the
    // 256 x 256 matrix in LDS as before.
    const uint8_t *wide_table;
};
constexpr uint32_t kWideClasses = 128;

// ------------------------------------------------------------------------------------------------
// Host-side scope.
// ------------------------------------------------------------------------------------------------
struct KernelStamp { hipEvent_t start, stop; const char *name; };

struct Scope {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    bool async = false;
    bool profiling = false;
    int compute_units = 0;
    // scratch arena (device), grown on demand, reused across calls
    char *scratch = nullptr;
    size_t scratch_bytes = 0;
    // staging for host-resident inputs
    char *stage = nullptr;
    size_t stage_bytes = 0;
    // right-edge columns of multi-pass wavefront groups
    char *boundary = nullptr;
    size_t boundary_bytes = 0;
    Plan *plan_host = nullptr;  // pinned
    char *plan_area = nullptr;  // device: hist | cursor | partials | plan, zeroed once (the scan kernel re-zeroes hist)
    uint32_t *plan_hist = nullptr, *plan_cursor = nullptr, *plan_leftover = nullptr;   // carved from plan_area at scope creation
    PlanPartial *plan_partials = nullptr;
    Plan *plan_dev = nullptr;
    // fused planning kernel (prepass.hip: k_plan_fused): double-buffered key histogram, grid barrier counter + its target
    uint32_t *plan_hist2[2] = {nullptr, nullptr};
    uint32_t *plan_barrier = nullptr;
    uint32_t plan_barrier_target = 0, plan_parity = 0;
    int fused_per_cu = -1;          // planning workgroups a compute unit holds (occupancy query, once)
    bool fused_disabled = false;    // the fused planner once failed to gather its grid on this scope
    hipStream_t side_stream = nullptr;  // plan read-back overlaps the first DP kernel
    hipEvent_t plan_ready = nullptr;
    unsigned long long *utf8_status = nullptr;   // look-back words of k_utf8_tile_decode: written by nothing else, tagged with utf8_epoch
    uint64_t utf8_status_cap = 0;
    uint32_t utf8_epoch = 0;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;   // second tape's UTF-8 decode runs on side_stream beside the first's
    // wavefront class kernels alternate between the scope's stream and side_stream (launch_wavefront): each is a
    // persistent grid with its own tail, and a 10 K-pair batch is only a few rounds of waves per kernel
    hipStream_t wf_main = nullptr;
    unsigned wf_toggle = 0;
    uint64_t wf_side_boundary = 0;   // int32 elements between the two streams' boundary areas
    bool hint_short = true;     // the previous call saw short pairs: enqueue k_direct_short (first call: assume yes)
    // Longest strings of the previous call on this scope (symbols), the basis for running the NEXT call on raw tapes
    // without a planning pre-pass; the kernels verify it per pair and raise CallSummary::violation if it no longer holds.
    bool hint_lengths = false;
    // What the last call READ as the byte totals of two raw device tapes (offsets[count]): a UTF-8 call on the same tapes believes
    // them instead of fetching them again -- two synchronous round trips before its first launch -- and k_utf8_finish checks
    // the belief on the device (marker kUtf8SizesChanged; the host then reads them afresh and redoes the call).
    // `ascii`: the last staging of that tape met no byte above 0x7F (the one-pass kernel's flag words, kUtf8AsciiWord): the next UTF-8 call on
    // the same tapes runs on their BYTES (code points of ASCII text are its bytes) next to a kernel that checks exactly that (api.hip).
    struct SizeBelief { const void *data = nullptr, *offsets = nullptr; size_t count = 0; int off64 = 0; uint64_t bytes = 0; bool valid = false, ascii = false; } size_belief[2];
    uint32_t early_return_last_us = 0;   // what the previous call that returned on its summary took (api.hip: wait_for_summary bounds its spin by it)
    float doubling_settled = -1.0f;   // share of the pairs the first stage of the scope's previous doubling call settled (< 0: none yet)
    uint32_t doubling_rest = 0;   // calls left before the two-stage (doubling) schedule of api.hip is tried again: its first stage settled too few pairs
    uint32_t utf8_strings_rest = 0;   // raw UTF-8 calls left before the string-by-string staging is tried again (it met a string too long for it)
    // k_align_cross_wide / _long met candidates with more than eight symbol classes: not tried again for THAT engine on THOSE tapes
    // (round 6: the latch used to hold for the whole scope -- DNA after one call on text ran 1.6 x slower for ever; another engine, i.e.
    // another folding of bytes into classes, or other tapes get their own try). A string longer than the believed lengths only drops
    // the belief: the next call measures afresh.
    struct AlignWideOff { uint64_t engine = 0; const void *a = nullptr, *b = nullptr; } align_wide_off;
    uint32_t hint_max_la = 0, hint_max_lb = 0;
    uint32_t hint_mean_x16 = 0;   // mean string length of the previous call, x16 (both tapes together)
    uint32_t hint_mean_string_x16 = 0;   // symbols per STRING of the previous call, x16 (a cross-product counts every string once)
    uint64_t summary_strings = 0;        // strings of the call whose summary is pending
    CallSummary *summary_host = nullptr;   // pinned, mapped: written by kernels, read by the host after a synchronisation
    CallSummary *summary_dev = nullptr;    // the same memory as the device sees it
    uint32_t summary_slot = 0;             // which of the block's two summaries the call in flight reports into (1: asynchronous)
    CallSummary *summary_target() const { return summary_dev + summary_slot; }
    uint32_t *done_counter = nullptr;      // device: workgroups finished (self-resetting), for "last one reports"
    bool summary_pending = false;          // a plan-free call's summary has not been read yet (harvest_timing)
    bool violation_seen = false;           // an asynchronous plan-free call reported a pair that did not fit: swh_scope_synchronize fails
    uint64_t summary_pairs = 0, summary_extra_bytes = 0;   // what the algorithmic byte count of that call needs besides the summary
    uint32_t summary_sym_bytes = 1, summary_ow = 8, summary_elem = 4;
    // Pipelined mode: calls alternate between `lanes` (internal scopes with their own stream, scratch and plan
    // buffers), so the planning pre-pass of call i+1 overlaps the DP kernel of call i. Results are ordered for the
    // caller by swh_scope_join / swh_scope_synchronize.
    bool pipelined = false;
    Scope *lanes[2] = {nullptr, nullptr};
    int next_lane = 0;
    hipEvent_t lane_done = nullptr;   // (on a lane) recorded after the lane's latest call
    hipEvent_t order_ev = nullptr;    // (on the parent) the caller's stream at the time of a pipelined call: the lane waits for it
    // kernels whose dynamic LDS was raised above 64 KB on THIS scope's device (function attributes are per device; a
    // process-wide flag would leave a second device without the opt-in)
    std::unordered_map<const void *, size_t> lds_opt_in;
    Scope *last_lane = nullptr;       // (on the parent) lane that took the latest call
    void *multi = nullptr;             // swh_scope_init_gpus: the per-device member scopes + RCCL communicators (sharded.hip)
    std::vector<KernelStamp> stamps;
    size_t stamps_used = 0;
    bool stamps_pending = false;       // recorded by an asynchronous call, not read yet (api.hip: harvest_timing)
    swh_timing_totals_t totals{};      // sums since profiling was switched on
    swh_timing_t last_timing{};
    std::string error;
};

void free_multi_scope(void *multi);   // sharded.hip
void drop_engine_clones(uint64_t engine_uid);   // sharded.hip: an alignment engine is being freed -- its per-device clones go with it

struct Engine {
    int kind;  // 0 = levenshtein, 1 = needleman-wunsch (global), 2 = smith-waterman (local)
    Scoring scoring;
    bool unit_costs;
    swh_algorithm_t algorithm;
    int8_t *matrix_dev;  // owned
    uint8_t *class_dev;  // owned
    int device;
    int8_t *matrix_host; // owned copy of the 256x256 matrix (alignment engines): a multi-device scope clones the engine per device
    uint64_t uid;        // never reused: keys the per-device clones (sharded.hip)
};
// (api.hip) an alignment engine like `source` -- kind, matrix, gaps -- with its tables on `scope`'s device
swh_status_t clone_alignment_engine(const Engine *source, swh_scope_t scope, void **out, const char **error);

// Kernel launch bookkeeping with optional hipEvent timing.
struct StampGuard {
    Scope *scope; size_t idx; bool on;
    StampGuard(Scope *s, const char *name);
    ~StampGuard();
};

// Launchers implemented in the kernel translation units ------------------------------------------
struct PrepassArgs {
    Job job;
    uint32_t mode;          // PlanMode
    uint32_t off64;         // offsets are u64 (else u32)
    uint32_t sym_bytes;     // 1 or 4
    uint32_t symmetric;     // scoring symmetric in (a,b): columns may be swapped to the shorter string
    int gap_open, gap_extend;  // max-plus gap costs (negative for distances), for the trivial pairs
    uint32_t unit_costs;    // Levenshtein (0,1,1,1): enables the |la-lb| > bound shortcut
    uint32_t banded;        // bound <= 63 with unit costs: pairs may take the banded kernel
    uint32_t local;         // local alignment: pairs with an empty side score 0
    uint32_t direct_short;  // unit-cost byte pairs with both sides <= 32 symbols are scored by k_direct_short
    uint32_t skip_upto;     // pairs with both sides <= this many symbols have been scored already (by k_align_short: the redo of a batch
                            // whose few longer strings did not fit its register row plans only the pairs with such a string); 0: none
    // Doubling (api.hip: a call with a bound beyond one band word, or none, first runs the one-word band at k1 = 63 -- rapidfuzz's own
    // schedule doubles its score hint the same way). stage1: plan the banded class only, every other pair gets job.bound + 1 stored ("not
    // settled") and is filed as done. redo_filter (the second stage): a pair whose stored result is <= redo_done_upto was settled by the
    // first stage and is filed as done untouched; the others are planned with the call's real bound.
    uint32_t stage1, redo_filter, redo_done_upto;
    uint32_t *perm;         // out: pair ids sorted by key
    uint16_t *keys;         // scratch: every pair's plan key, written by k_plan_hist, read back by k_plan_scatter
    uint32_t *hist;         // scratch: kKeys counters
    uint32_t *cursor;       // scratch: kKeys cursors
    PlanPartial *partials;  // scratch: 2 x kMaxPartials per-block work-unit sums
    // When k_direct_short runs it visits every pair anyway, so it also finishes the trivial pairs, sums the work
    // units and counts the pairs it could NOT finish here. Zero left over: k_plan_hist and k_plan_scatter return at
    // once (nothing to plan); k_plan_scan folds the sums and re-zeroes the counter.
    uint32_t *leftover;
    Plan *plan;             // out (device)
    // k_direct_short on its own (launch_direct_short_alone): no planning kernels follow, the last workgroup reports
    CallSummary *summary;
    uint32_t *done_counter;
};
void launch_prepass(Scope *scope, const PrepassArgs &args);

struct KernelArgs {
    Job job;
    const uint32_t *perm;
    const Plan *plan;   // device
    Scoring scoring;
    uint32_t off64, sym_bytes, symmetric, affine;
    uint32_t local;         // Smith-Waterman: local alignment (floors at 0, maximum over all cells)
    uint32_t local_narrow;  // (local) no score of the batch can reach 2^16 (largest cost x the shorter side's longest string): k_nwprofile's plain maxima run as v_max_u16
    int32_t *boundary;      // scratch for multi-pass wavefront
    uint64_t boundary_stride;  // int32 elements per group slot
    uint32_t band_fixed_items;   // k_banded: items of exactly P pairs (comparison knob STRINGWARS_AMD_BAND_ITEMS=fixed)
    uint32_t *ticket;       // k_bitparallel_long: the launch's next pair (zeroed before the launch); null: pairs dealt round-robin
};
void launch_bitparallel(Scope *scope, const KernelArgs &args, uint64_t pairs);
// tiled.hip: bit-parallel Levenshtein with per-workgroup planning (no pre-pass, no host round trip)
struct TilePlan { uint32_t tile, tiles, blocks, shift; };
TilePlan plan_tiles(uint64_t pairs, uint32_t slots, uint32_t longest_text, uint32_t tile_max);
void launch_bitparallel_tiled(Scope *scope, const KernelArgs &args, uint64_t pairs, uint32_t longest_text);
// cross.hip: dense queries x candidates for word-sized strings, the query's match table shared by a wave
void launch_cross_short(Scope *scope, const Job &job, uint32_t off64, uint32_t sym_bytes);   // sym_bytes 4: code points (decoded tapes)
// short.hip: pairwise batches of strings <= 16 bytes: chunks staged in LDS, affixes cut, sorted by what remains
// `mean_bytes`: mean string length of the longer tape, x16 (0: unknown); sizes the chunks so that their segments fit the LDS arrays
void launch_short_tiled(Scope *scope, const Job &job, uint32_t off64, uint32_t mean_bytes_x16);
// prepass.hip: k_direct_short on its own (every pair known to be word-sized)
void launch_direct_short_alone(Scope *scope, const PrepassArgs &args);
// Longest string of a tape (in offsets units) -> *longest (device word, atomicMax; zero it first)
void launch_tape_longest(Scope *scope, const void *offsets, uint32_t off64, uint64_t count, uint32_t *longest);
// pairs of class kClassBpLong (needs the host plan: carry scratch is sized by the longest text)
void launch_bitparallel_long(Scope *scope, KernelArgs args, const Plan &plan_host);
// u32 carry words per wave of k_bitparallel_long: 2 pass parities x (+1 | -1 deltas) x one bit per text column, plus
// slack for the prefetch of the word after the last; the kernel never has more than 4096 waves
inline uint64_t bp_long_carry_words(uint64_t longest_text) { return 4 * (longest_text / 32 + 4); }
constexpr uint64_t kBpLongMaxWaves = 4096;
void launch_wavefront(Scope *scope, const KernelArgs &args, const Plan &plan_host);
void launch_banded(Scope *scope, const KernelArgs &args, uint64_t pairs);
// nwprofile.hip: global alignment on a class table with the substitution scores served from a column profile in LDS; takes
// the pairs of plan classes kClassWf64 + kNwProfileFirstWide .. kClassWfMulti (more than 384 columns) = perm[first, first + count)
constexpr int kNwProfileFirstWide = 3;
uint32_t nwprofile_waves(const Scope *scope, uint32_t classes);   // waves of a launch = boundary areas it needs
void launch_nwprofile(Scope *scope, KernelArgs args, uint32_t first, uint32_t count);
int wavefront_strip_cap();
// alignshort.hip: NW / SW scores on a class table, both strings <= 32 bytes (`longest`: of both tapes), one pair per lane; plan-free
void launch_align_short(Scope *scope, const KernelArgs &args, uint32_t longest, bool wide);   // wide: k_align_cross_wide (33 .. 128 symbols, <= 8 classes per item)
// the same for queries x candidates of any length over a small alphabet (<= 8 classes per work item): columns in passes of 128
// (Gotoh: 64), the boundary column between passes in args.boundary -- align_long_waves() areas of (longest_rows + 8) x 64 ints
// (x 2 for Gotoh's E); queries of up to 4096 symbols
uint32_t align_long_waves(const Scope *scope, uint64_t items, uint32_t longest_rows, bool affine);   // (capped by the boundary budget, 640 MB)
bool align_long_fits(const Scope *scope, uint64_t items, uint32_t longest_rows, bool affine);         // would a full launch stay inside it?
uint32_t align_long_queries(const Scope *scope, uint64_t queries, uint64_t candidates);   // queries per work item (16 .. 1)
void launch_align_long(Scope *scope, const KernelArgs &args, uint32_t longest_rows);

// UTF-8 staging: decodes a byte tape into u32 code points + u64 code-point offsets.
constexpr int kUtf8Pass = 1024, kUtf8Passes = 8;     // a block walks its tile in passes of 256 threads x one dword
constexpr int kUtf8Tile = kUtf8Pass * kUtf8Passes;   // bytes per block
constexpr int kUtf8Subs = kUtf8Tile / 256;           // 256-byte sub-tiles per tile
constexpr int kUtf8AsciiWord = 3;   // flag words [3], [4]: the tape staged by slot 0 / 1 (or the first / second tape of one launch) held a byte above 0x7F
constexpr uint32_t kUtf8SizesChanged = 0xFFFFFFFEu;   // `invalid` marker: a tape's offsets[count] is not the byte total the call believed
constexpr int kUtf8FlagWords = 96;   // the tickets of two concurrent launches sit 128 bytes apart and away from the flag: on one
                                     // cache line their atomics took turns (2 x 100 MB staged in 0.32 ms instead of 0.28)
struct Utf8Args {
    TapeRef in; uint32_t off64;
    uint32_t *symbols;     // out, capacity = total bytes
    uint64_t *offsets;     // out, count+1
    uint32_t *counts;      // scratch, count entries
    uint32_t *invalid;     // out flag (non-zero = invalid UTF-8); words [1], [2] are per-tape balance counters (count / scan / write
                           // path); words [32 (1 + slot)] the one-pass kernel's tile tickets: kUtf8FlagWords zeroed words in all
    uint32_t slot;         // which balance counter this tape uses (0 or 1)
    uint64_t total_bytes;
};
void launch_utf8_decode(Scope *scope, const Utf8Args &args);
// the one-pass staging (default; prepass.hip) takes both tapes of a call in one go -- `invalid` must point at kUtf8FlagWords ZEROED
// words: [0] the invalid-UTF-8 marker, [32 (1 + slot)] the launch's tile ticket (the kernel takes tickets from there: anything
// but zero at launch is a hang or a tile never decoded)
bool utf8_one_pass();
// staging string by string (prepass.hip: k_utf8_strings): the code points of string i from symbols[offsets[i]] on, (first, end)
// pairs in `extents` (2 count + 1 entries: TapeRef::gap = 1). `invalid`: the call's flag words ([0] marker, [3], [4] "not ASCII").
struct Utf8StringsJob { const uint8_t *data; const void *offsets; uint64_t count, total; uint32_t *symbols; uint64_t *extents; };
constexpr uint32_t kUtf8StringTooLong = 0xFFFFFFFDu;   // `invalid` marker: a string beyond kUtf8StringLongest bytes -- stage the flat way
constexpr uint32_t kUtf8StringLongest = 64u * 1024u;
constexpr uint32_t kUtf8StringsMeanBytes = 192;        // tapes whose mean string is shorter keep the flat staging
void launch_utf8_strings(Scope *scope, const Utf8StringsJob &a, const Utf8StringsJob *b, uint32_t off64, uint32_t *invalid);
void launch_utf8_decode_pair(Scope *scope, const Utf8Args &a, const Utf8Args *b, uint64_t first_word, bool opened);
void utf8_status_open(Scope *scope, uint64_t words);

// Environment switches. The shipped library reads four: TRACE, STAMPS, EARLY_RETURN, SHARD_CHECK. Everything else --
// the A / B switches measurements were taken with, fault injection, shapes no input reaches, the UTF-8 staging of rounds 1-2 kept as
// the tests' second implementation -- is a TEST HOOK and exists in the TEST library only (`make test-lib`: -DSWH_TEST_HOOKS -> libstringwars_amd_test.so, which the tests that need a hook
// load in a child process); the shipped library does not look at them and does not carry the code behind them.
inline const char *test_hook(const char *name) {
#ifdef SWH_TEST_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#define SWH_HIP_CHECK_DECLARED 1
#define SWH_HIP_CHECK(expr)                                                                          \
    do {                                                                                              \
        hipError_t err__ = (expr);                                                                    \
        if (err__ != hipSuccess) throw ::swh::HipFailure{err__, #expr};                               \
    } while (0)
struct HipFailure { hipError_t code; const char *what; };

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (scope = device, kernel)
inline void opt_in_dynamic_lds(Scope *scope, const void *func, size_t bytes) {
    auto it = scope->lds_opt_in.find(func);
    if (it != scope->lds_opt_in.end() && it->second >= bytes) return;
    SWH_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    scope->lds_opt_in[func] = bytes;
}

}  // namespace swh
