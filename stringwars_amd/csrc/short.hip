// short.hip -- unit-cost Levenshtein for batches of WORD-SIZED byte strings (both sides <= 16 bytes), pairwise.
//
// BASELINE config C5 (100 M short-word pairs) and C1: a pair is ~43 DP cells, so everything around the cells decides the
// rate. k_direct_short (prepass.hip) scores such pairs one per lane in tape order; a wave then walks to its LONGEST
// pattern and text (15 rows and columns where the mean is 6), builds and clears a match table per pair, and fetches every
// string with a scattered 16-byte load. Here a workgroup takes a CHUNK of <= 1024 consecutive pairs and
//   A. copies the chunk's two tape segments -- consecutive pairs are consecutive bytes -- into LDS with coalesced 16-byte
//      loads (the only global reads besides the offsets);
//   B. per pair: cuts the common prefix and suffix (what rapidfuzz's Levenshtein does first, too; eight bytes of each are
//      compared with two xor + find-first-bit), finishes the pairs with nothing left on one side, and counts the others
//      by (text length, pattern length) of what remains;
//   C-D. counting-sorts them in LDS (256 keys) as 8-byte descriptors;
//   E. runs work items of 64 sorted pairs, heaviest first, one pair per lane: lanes of an item have (nearly) the same row
//      and column counts, so the wave walks ~3 columns instead of ~15. Match tables are 16-bit (patterns <= 16 rows):
//      lanes l and l + 32 share a dword -- they are never in the same LDS lane group -- 4 KB per wave;
//   F. writes the chunk's distances in one coalesced sweep.
// Strings longer than 16 bytes raise `violation` (the host redoes the call on another route), like tiled.hip.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "bp_window.hpp"

namespace swh {

// (make EXTRA=-DSWH_SHORT_THREADS=64 builds the kernel with one-wave workgroups and chunks of 256 pairs -- a wave owning its chunk
// end to end, no barrier that costs anything, no ticket between waves: bit-exact, and 0.386 ms for C5's 20 M pairs against 0.187)
#ifndef SWH_SHORT_THREADS
#define SWH_SHORT_THREADS 256
#endif
// (make EXTRA=-DSWH_SHORT_PER=2: chunks of 512 pairs, 28 KB of LDS, five workgroups per compute unit instead of four -- measured in round 5)
#ifndef SWH_SHORT_PER
#define SWH_SHORT_PER 4
#endif
constexpr int kShortThreads = SWH_SHORT_THREADS, kShortWaves = kShortThreads / 64, kShortPer = SWH_SHORT_PER;
constexpr int kShortChunk = kShortThreads * kShortPer;   // pairs per chunk
constexpr int kShortMaxLen = 16;
constexpr int kShortCap = 6912 * kShortThreads / 256 * kShortPer / 4;      // bytes of one tape's segment a chunk may bring into LDS (432 units of 16; keeps the workgroup below 40 KB)
constexpr int kShortPerCu = kShortThreads == 256 ? (kShortPer == 4 ? 4 : 5) : 13;   // workgroups a compute unit holds (LDS)
constexpr int kShortPad = 16;        // before (tail windows reach back 8 bytes) and after (16-byte windows reach forward)
constexpr int kShortKeys = 256;      // (text length - 1) * 16 + (pattern length - 1)

struct __attribute__((aligned(4096))) ShortLds {
    uint32_t tables[kShortWaves][1024];                 // per wave: Lo[16][32] | Hi[16][32]; 4 KB apart for NibbleTables16
    uint8_t a[kShortPad + kShortCap + kShortPad];
    uint8_t b[kShortPad + kShortCap + kShortPad];
    union {
        uint2 sorted[kShortChunk];                      // descriptors in key order
        struct {                                        // after the last chunk: the workgroup's sums and the call summary
            unsigned long long cells, syms;
            uint32_t maxa, maxb, misfit;
            SummaryLds summary;
        } tail;
    };
    uint32_t hist[kShortKeys];                          // pairs per key, then exclusive starts
    uint8_t staged[kShortChunk];                        // distances of the chunk (<= 16)
    uint32_t total, reserved;
    uint32_t mixed;                                     // some byte of the chunk differs from its wave's first byte in the upper three bits
    uint32_t refs[kShortWaves];                         // those first bytes' upper three bits, broadcast over a dword
};
static_assert(sizeof(ShortLds) <= 163840 / kShortPerCu, "workgroups per compute unit");

// NibbleTables (bp_window.hpp) for 16-row patterns: entry v of Lo sits at tbase + (v << 7), of Hi at tbase + 2048 + (v << 7),
// tbase = the wave's table + 4 * (lane & 31); lanes >= 32 keep their bits in the upper half of the shared dword.
struct NibbleTables16 {
    uint32_t tbase, mask, mask5;
    __device__ __forceinline__ void init(uint32_t *table, int lane) {
        tbase = (uint32_t)(uintptr_t)(lds_u32 *)(table + (lane & 31));
        if (tbase & 0xF80u) __builtin_trap();   // the layout assumptions (4 KB tables on 4 KB boundaries): fail loudly
        mask = 0x780u;
        mask5 = 0xF80u;
        asm volatile("" : "+v"(mask), "+v"(mask5));
    }
    template <int U> __device__ __forceinline__ uint32_t lo_addr(uint32_t x) const {   // x holds the symbol in byte U
        uint32_t s;
        if constexpr (U == 0) s = x << 7;
        else s = x >> (8 * U - 7);
        return (uint32_t)__builtin_amdgcn_bitop3_b32((int)s, (int)mask, (int)tbase, 0xEA);
    }
    template <int U> __device__ __forceinline__ uint32_t hi_addr(uint32_t x) const {
        uint32_t s;
        if constexpr (U == 0) s = x << 3;
        else s = x >> (8 * U - 3);
        return (uint32_t)__builtin_amdgcn_bitop3_b32((int)s, (int)mask, (int)tbase, 0xEA);
    }
    template <int U> __device__ __forceinline__ uint32_t lookup(uint32_t x) const {
        return *(const lds_u32 *)(uintptr_t)lo_addr<U>(x) & *(const lds_u32 *)(uintptr_t)(hi_addr<U>(x) + 2048);
    }
    template <int U> __device__ __forceinline__ void insert(uint32_t x, uint32_t bit) const {
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)lo_addr<U>(x), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(hi_addr<U>(x) + 2048), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // When every byte of a chunk shares its upper three bits (lower-case words, upper-case words, digits ...) a symbol is
    // known by its low five bits: ONE table of 32 entries over the same 4 KB -- entry v at tbase + (v << 7) -- half the
    // ds_or per row, half the look-ups and address arithmetic per column.
    template <int U> __device__ __forceinline__ uint32_t one_addr(uint32_t x) const {   // x holds the symbol in byte U
        uint32_t s;
        if constexpr (U == 0) s = x << 7;
        else s = x >> (8 * U - 7);
        return (uint32_t)__builtin_amdgcn_bitop3_b32((int)s, (int)mask5, (int)tbase, 0xEA);
    }
    template <int U> __device__ __forceinline__ uint32_t lookup_one(uint32_t x) const { return *(const lds_u32 *)(uintptr_t)one_addr<U>(x); }
    template <int U> __device__ __forceinline__ void insert_one(uint32_t x, uint32_t bit) const {
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)one_addr<U>(x), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// one wave's DS instructions execute in issue order: the compiler just must not move them across each other
__device__ __forceinline__ void short_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

#ifdef SWH_SHORT_WG_SPANS
// Diagnostic build only (make EXTRA=-DSWH_SHORT_WG_SPANS): when every workgroup of k_short_tiled starts and ends (100 MHz clock)
__device__ unsigned long long g_short_wg[4096][2];
extern "C" void swh_debug_short_wg(unsigned long long *out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_short_wg), sizeof(unsigned long long) * 4096 * 2);
}
#define SHORT_WG_STAMP(which) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_short_wg[blockIdx.x][which] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SHORT_WG_STAMP(which) do { } while (0)
#endif
#ifdef SWH_SHORT_PROFILE
// Diagnostic build only (make EXTRA=-DSWH_SHORT_PROFILE): wave cycles per phase of k_short_tiled, summed over waves.
__device__ unsigned long long g_short_phase[10];
extern "C" void swh_debug_short_phases(unsigned long long *out) {
    unsigned long long zero[10] = {};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_short_phase), sizeof(zero));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_short_phase), zero, sizeof(zero));
}
// wall-clock (100 MHz) time stamps of the first chunks of every 16th workgroup: g_short_span[workgroup / 16][chunk][stamp]
__device__ unsigned long long g_short_span[64][8][8];
extern "C" void swh_debug_short_spans(unsigned long long *out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_short_span), sizeof(unsigned long long) * 64 * 8 * 8);
}
#define SHORT_STAMP(slot) do { if (span_on && threadIdx.x == 0 && chunk_no < 8) g_short_span[blockIdx.x >> 4][chunk_no][slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SHORT_STAMP(slot) do {} while (0)
#endif

struct ShortArgs {
    Job job;
    uint32_t tile, tiles;   // pairs per workgroup visit (a multiple of 64, <= kShortChunk)
    PlanPartial *partials;
    uint32_t *done_counter;
    CallSummary *summary;
};

// Wide LDS reads are fast only when naturally aligned (a misaligned ds_read_b64 / b128 is served lane by lane: 65 LDS
// cycles against 7-11, tools/lds_ops.hip), so windows at string addresses are read as aligned dwords and realigned with
// v_alignbyte (which takes the byte shift from the low two bits of the address itself).
__device__ __forceinline__ unsigned long long lds_window8(const uint8_t *array, uint32_t at) {
    const uint32_t *p = (const uint32_t *)(array + (at & ~3u));
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    return (unsigned long long)__builtin_amdgcn_alignbyte(d1, d0, at) | ((unsigned long long)__builtin_amdgcn_alignbyte(d2, d1, at) << 32);
}

// What a thread requests from global memory for one chunk: the offsets of its four pairs and its share of the two tape
// segments. Requested for chunk i + 1 before the work items of chunk i run, consumed after them.
template <typename Off> struct ShortRequest {
    Off oa0[kShortPer], oa1[kShortPer], ob0[kShortPer], ob1[kShortPer];
    uint4 va[2], vb[2];
};
struct ShortChunk {
    bool any;
    uint64_t base;        // first pair
    uint32_t len;         // pairs
    uint64_t a_lo, b_lo;  // where its segments start in the tapes
    uint32_t bytes_a, bytes_b;
};

// kWide: both tapes hold at least 16 bytes, so every 16-byte unit of a segment can be requested with ONE unconditional load
// whose address is clamped into the tape (a guard per load would put a memory wait behind each of them); the one unit that
// straddles the end of a tape is repaired byte by byte afterwards. Tapes shorter than that go byte by byte altogether.
template <typename Off, bool kWide>
__device__ __forceinline__ void short_run(const ShortArgs &args, ShortLds &lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Job &job = args.job;
    const Off *off_a = (const Off *)job.a.offsets, *off_b = (const Off *)job.b.offsets;
    const uint8_t *a_data = (const uint8_t *)job.a.data, *b_data = (const uint8_t *)job.b.data;
    const uint64_t a_total = (uint64_t)off_a[job.a.count], b_total = (uint64_t)off_b[job.b.count];
    NibbleTables16 nib;
    nib.init(lds.tables[wave], lane);
    const uint32_t half_shift = lane >= 32 ? 16u : 0u;
    {   // tables start clean
        uint4 *t = (uint4 *)lds.tables[wave];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k * 64 + lane] = make_uint4(0, 0, 0, 0);
    }
    if (threadIdx.x == 0) lds.mixed = 0;
    __syncthreads();
    SHORT_WG_STAMP(0);
    uint32_t cells = 0, syms = 0, maxa = 0, maxb = 0, misfit = 0;
    const uint32_t bound = job.bound;
#ifdef SWH_SHORT_PROFILE
    unsigned long long items_done = 0;
    const bool span_on = (blockIdx.x & 15u) == 0 && (blockIdx.x >> 4) < 64;
    uint32_t chunk_no = 0;
#endif

    // ---- the workgroup's chunks, in order: tiles blockIdx.x, blockIdx.x + gridDim.x, ...; a tile is one chunk when its
    // segments fit the LDS arrays, else it is halved until they do ---------------------------------------------------------------
    uint32_t tile = blockIdx.x, done = 0;   // where the next chunk starts
    auto candidate = [&](uint64_t &base, uint32_t &len) -> bool {   // the rest of the current tile, without taking it
        while (tile < args.tiles) {
            const uint64_t tile_base = (uint64_t)tile * args.tile;
            const uint32_t count = (uint32_t)(job.pairs - tile_base < args.tile ? job.pairs - tile_base : args.tile);
            if (done < count) { base = tile_base + done; len = count - done; return true; }
            tile += gridDim.x; done = 0;
        }
        return false;
    };
    // the four offsets that bound a candidate's segments, as ONE vector load (lanes 0..3): it is consumed much later, and a
    // scalar load would be waited for at the next LDS wait (both count on lgkmcnt)
    auto bounds_request = [&](bool any, uint64_t base, uint32_t len) -> Off {
        const Off *src = ((lane & 2) ? off_b : off_a) + (any ? base + ((lane & 1) ? len : 0u) : 0);
        return *src;
    };
    auto lane_value = [&](Off v, int l) -> uint64_t {
        if constexpr (sizeof(Off) == 8)
            return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l) << 32);
        else
            return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, l);
    };
    // takes the next chunk: `hint` holds the bounds of the candidate (base0, len0) if hinted
    auto next_chunk = [&](bool hinted, uint64_t base0, uint32_t len0, Off hint) -> ShortChunk {
        ShortChunk c{};
        uint64_t base; uint32_t len;
        while (candidate(base, len)) {
            uint64_t a_lo, a_hi, b_lo, b_hi;
            if (hinted && base == base0 && len == len0) {
                a_lo = lane_value(hint, 0); a_hi = lane_value(hint, 1); b_lo = lane_value(hint, 2); b_hi = lane_value(hint, 3);
            } else {
                a_lo = (uint64_t)off_a[base]; a_hi = (uint64_t)off_a[base + len]; b_lo = (uint64_t)off_b[base]; b_hi = (uint64_t)off_b[base + len];
            }
            hinted = false;
            bool fits;
            for (;;) {
                fits = a_hi - a_lo <= (uint64_t)kShortCap && b_hi - b_lo <= (uint64_t)kShortCap;
                if (fits || len <= 64) break;
                len = ((len >> 1) + 63u) & ~63u;
                a_hi = (uint64_t)off_a[base + len]; b_hi = (uint64_t)off_b[base + len];
            }
            done += len;
            if (fits) {
                c.any = true; c.base = base; c.len = len; c.a_lo = a_lo; c.b_lo = b_lo;
                c.bytes_a = (uint32_t)(a_hi - a_lo); c.bytes_b = (uint32_t)(b_hi - b_lo);
                return c;
            }
            misfit = 1;   // 64 pairs beyond the capacity: some string is longer than 16 bytes
        }
        return c;
    };
    auto request = [&](const ShortChunk &c, ShortRequest<Off> &r) {
        // uniform bases, made visibly so (the chunk travels through loop-carried registers): the loads then take an SGPR base
        // and a 32-bit lane offset instead of 64-bit address arithmetic per load
        auto uniform = [](uint64_t v) -> uint64_t {
            return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
                   ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
        };
        const uint64_t first = uniform(c.base);
        const Off *pa = off_a + first, *pb = off_b + first;
#pragma unroll
        for (int k = 0; k < kShortPer; ++k) {
            uint32_t q = (uint32_t)k * kShortThreads + threadIdx.x;
            q = q < c.len ? q : c.len - 1;
            r.oa0[k] = pa[q]; r.oa1[k] = pa[q + 1];
            r.ob0[k] = pb[q]; r.ob1[k] = pb[q + 1];
        }
        if constexpr (kWide) {
            // unit u of a segment starts 16 u bytes into it; units that would reach past the tape are read from its last 16 bytes
            // instead (and repaired below); units past the segment are harmless slack
            const uint8_t *sa = a_data + uniform(c.a_lo), *sb = b_data + uniform(c.b_lo);
            const int64_t room_a = (int64_t)(a_total - c.a_lo) - 16, room_b = (int64_t)(b_total - c.b_lo) - 16;
            const int lim_a = (int)(room_a > 0x10000 ? 0x10000 : room_a), lim_b = (int)(room_b > 0x10000 ? 0x10000 : room_b);
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const int at = 16 * (u2 * kShortThreads + (int)threadIdx.x);
                if (at < kShortCap) {   // (the arrays hold kShortCap / 16 units)
                    __builtin_memcpy(&r.va[u2], sa + (at < lim_a ? at : lim_a), 16);
                    __builtin_memcpy(&r.vb[u2], sb + (at < lim_b ? at : lim_b), 16);
                }
            }
        }
    };

    uint64_t cand_base = 0; uint32_t cand_len = 0;
    ShortChunk cur = next_chunk(false, 0, 0, (Off)0);
    ShortRequest<Off> req;
    if (cur.any) request(cur, req);
    while (cur.any) {
        const uint64_t base = cur.base;
        const uint32_t len = cur.len;
        SHORT_STAMP(7);   // top of the chunk
        // ---- A: the chunk's segments into LDS (requested while the previous chunk's work items ran) ---------------------------
        const bool cand = candidate(cand_base, cand_len);
        const Off cand_bounds = bounds_request(cand, cand_base, cand_len);
        for (int i = threadIdx.x; i < kShortKeys; i += kShortThreads) lds.hist[i] = 0;
        // Do all bytes of the chunk share their upper three bits? (Checked on the 16-byte units as they arrive -- units past
        // the segments hold the tapes' next bytes: a false alarm there only costs the chunk its fast path.) The previous
        // chunk's flag was read before its last barrier; thread 0 of the first wave to get here resets it.
        uint32_t spread = 0;
        if constexpr (kWide) {
            const uint32_t ref = ((uint32_t)__builtin_amdgcn_readfirstlane((int)req.va[0].x) & 0xE0u) * 0x01010101u;
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {   // (units past the segment are harmless slack)
                const uint32_t u = (uint32_t)u2 * kShortThreads + threadIdx.x;
                if (16 * u < (uint32_t)kShortCap) {
                    *(uint4 *)(lds.a + kShortPad + 16 * u) = req.va[u2];
                    *(uint4 *)(lds.b + kShortPad + 16 * u) = req.vb[u2];
                    if (16 * u < cur.bytes_a + 16) spread |= (req.va[u2].x ^ ref) | (req.va[u2].y ^ ref) | (req.va[u2].z ^ ref) | (req.va[u2].w ^ ref);
                    if (16 * u < cur.bytes_b + 16) spread |= (req.vb[u2].x ^ ref) | (req.vb[u2].y ^ ref) | (req.vb[u2].z ^ ref) | (req.vb[u2].w ^ ref);
                }
            }
            // (every wave compares with its own first byte; the four references are compared behind the barrier)
            if (spread & 0xE0E0E0E0u) lds.mixed = 1;
            if (lane == 0) lds.refs[wave] = ref;
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const uint32_t at = 16u * ((uint32_t)u2 * kShortThreads + threadIdx.x);
                if (__builtin_expect(at < cur.bytes_a && cur.a_lo + at + 16 > a_total, 0))
                    for (uint64_t g = cur.a_lo + at; g < a_total; ++g) lds.a[kShortPad + (uint32_t)(g - cur.a_lo)] = a_data[g];
                if (__builtin_expect(at < cur.bytes_b && cur.b_lo + at + 16 > b_total, 0))
                    for (uint64_t g = cur.b_lo + at; g < b_total; ++g) lds.b[kShortPad + (uint32_t)(g - cur.b_lo)] = b_data[g];
            }
        } else {
            for (uint32_t i = threadIdx.x; i < cur.bytes_a; i += kShortThreads) lds.a[kShortPad + i] = a_data[cur.a_lo + i];
            for (uint32_t i = threadIdx.x; i < cur.bytes_b; i += kShortThreads) lds.b[kShortPad + i] = b_data[cur.b_lo + i];
        }
        SHORT_STAMP(0);   // A: copy
        __syncthreads();
        SHORT_STAMP(1);   // barrier after A
        bool one_table = kWide && !lds.mixed;
#pragma unroll
        for (int w = 1; w < kShortWaves; ++w) one_table = one_table && lds.refs[0] == lds.refs[w];
        // ---- B: cut the common affixes, finish what is trivial, count the rest by (text, pattern) length -------------
        // (the LDS windows of two pairs are requested before the first one is used, no branch in between; all four at once
        // hold 48 dwords in flight and spill)
        uint32_t d0[kShortPer], d1[kShortPer], key[kShortPer], rank[kShortPer];
#pragma unroll
        for (int k0 = 0; k0 < kShortPer; k0 += 2) {
            uint32_t ra0s[2], rb0s[2], las[2], lbs[2];
            unsigned long long heads[2], tails[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + j;
                const uint32_t ra0 = (uint32_t)(req.oa0[k] - (Off)cur.a_lo), ra1 = (uint32_t)(req.oa1[k] - (Off)cur.a_lo);
                const uint32_t rb0 = (uint32_t)(req.ob0[k] - (Off)cur.b_lo), rb1 = (uint32_t)(req.ob1[k] - (Off)cur.b_lo);
                const bool valid = (uint32_t)k * kShortThreads + threadIdx.x < len;
                ra0s[j] = ra0; rb0s[j] = rb0;
                las[j] = valid ? ra1 - ra0 : 0u;    // lanes past the chunk repeat its last pair: they count as two empty strings
                lbs[j] = valid ? rb1 - rb0 : 0u;
                heads[j] = lds_window8(lds.a + kShortPad, ra0) ^ lds_window8(lds.b + kShortPad, rb0);
                tails[j] = lds_window8(lds.a + kShortPad - 8, ra1) ^ lds_window8(lds.b + kShortPad - 8, rb1);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + j;
                const uint32_t q = (uint32_t)k * kShortThreads + threadIdx.x;
                uint32_t la = las[j], lb = lbs[j];
                cells += la * lb;
                syms += la + lb;
                maxa = la > maxa ? la : maxa;
                maxb = lb > maxb ? lb : maxb;
                const uint32_t longer = la > lb ? la : lb, mn = la < lb ? la : lb;
                uint32_t pre = heads[j] ? (uint32_t)__builtin_ctzll(heads[j]) >> 3 : 8u;
                pre = pre < mn ? pre : mn;
                uint32_t suf = tails[j] ? (uint32_t)__builtin_clzll(tails[j]) >> 3 : 8u;
                suf = suf < mn - pre ? suf : mn - pre;
                const uint32_t cut = pre + suf;
                const uint32_t m = longer - cut, n = mn - cut;
                const uint32_t pa = (uint32_t)offsetof(ShortLds, a) + kShortPad + ra0s[j] + pre, pb = (uint32_t)offsetof(ShortLds, b) + kShortPad + rb0s[j] + pre;
                const bool a_is_pattern = la >= lb;
                const uint32_t pat = a_is_pattern ? pa : pb, txt = a_is_pattern ? pb : pa;
                const bool cut_off = bound != 0xFFFFFFFFu && m - n > bound;
                key[k] = 0xFFFFFFFFu; rank[k] = 0;
                d0[k] = pat | (txt << 16);
                d1[k] = m | (n << 8) | (q << 16);
                if (q < len && longer <= (uint32_t)kShortMaxLen) {
                    // nothing left on one side: m insertions. One symbol left on the longer side, hence on both (n <= m): they
                    // differ -- equal ones would have gone with the prefix or the suffix, whose windows cover all 16 bytes --, one
                    // substitution, m again. (5.6 % of the synthetic words, beside the 24 % with an empty side.)
                    if (cut_off || n == 0 || m == 1) {
                        lds.staged[q] = (uint8_t)(cut_off ? bound + 1 : clamp_bound(m, bound));
                    } else {
                        key[k] = (n - 1) * 16 + (m - 1);
                        rank[k] = __hip_atomic_fetch_add(&lds.hist[key[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        SHORT_STAMP(2);   // B
        __syncthreads();
        if (threadIdx.x == 0) lds.mixed = 0;   // everybody has read it; the next chunk's step A may raise it again
        // ---- C: exclusive scan of the 256 key counts (wave 0, four keys per lane) -------------------------------------
        if (wave == 0) {
            const uint4 c = ((const uint4 *)lds.hist)[lane];
            const uint32_t sum = c.x + c.y + c.z + c.w;
            const uint32_t incl = wave_inclusive_sum_u32(sum);
            const uint32_t e0 = incl - sum;
            ((uint4 *)lds.hist)[lane] = make_uint4(e0, e0 + c.x, e0 + c.x + c.y, e0 + c.x + c.y + c.z);
            if (lane == 63) lds.total = incl;
        }
        __syncthreads();
        // ---- D: descriptors into key order -----------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < kShortPer; ++k)
            if (key[k] != 0xFFFFFFFFu) lds.sorted[lds.hist[key[k]] + rank[k]] = make_uint2(d0[k], d1[k]);
        __syncthreads();
        SHORT_STAMP(3);   // C, D and their barriers
        // ---- the next chunk's global reads go out now and come back while this chunk's work items run -----------------------
        const ShortChunk nxt = next_chunk(cand, cand_base, cand_len, cand_bounds);
        ShortRequest<Off> req_next;
        // Nothing older may be pending when the requests go out: with the previous chunk's result stores still counted (they
        // finished long ago), hipcc cannot tell what the work-item loop below may overwrite and puts a wait for EVERYTHING --
        // the requests included -- in front of the first item; the whole round trip was exposed.
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        if (nxt.any) request(nxt, req_next);
        // ---- E: work items of 64 sorted pairs, heaviest (last) first ---------------------------------------------------------
        // An item's LDS round trips are taken off its dependency chain (round 6; before, a wave waited for eight of them per item
        // -- ticket, descriptor, strings, one look-up per column -- at four waves per SIMD): items are dealt without a ticket, the
        // NEXT item's descriptor is requested while an item's columns run and its strings while the item's result is
        // staged and the table cleared; the match words of an item's first eight columns are requested together, before the
        // first column is computed. DS instructions of a wave execute in order: the look-ups see the rows OR-ed in before
        // them, the clear comes behind the last look-up.
        __builtin_amdgcn_s_setprio(0);
        auto work_items = [&](auto one_tag) {
            constexpr bool kOne = decltype(one_tag)::value;
            const uint32_t total = lds.total, items = (total + 63u) >> 6;
            // Items are dealt round by round, the waves taking turns in opposite directions (round r: wave w takes item r W + w when
            // r is even, r W + W - 1 - w when odd; heaviest items first): every wave knows its next item without asking. (An LDS
            // ticket costs a round trip at the head of every item -- hipcc waits for a one-lane atomic's result on the spot --, and
            // ds_append, the uniform counter, hung the kernel on its second chunk: tools/c5_probe.py.)
            const uint32_t my_wave = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
            auto ticket_of = [&](uint32_t round) -> uint32_t {
                return round * (uint32_t)kShortWaves + ((round & 1u) ? (uint32_t)kShortWaves - 1u - my_wave : my_wave);
            };
            auto descriptor = [&](uint32_t t) -> uint2 {   // (lanes past the item's pairs read its first pair; their lengths are zeroed when the item starts)
                const uint32_t item = items - 1 - t;
                const uint32_t e = item * 64 + (uint32_t)lane;
                return lds.sorted[e < total ? e : item * 64];
            };
            struct Raw { uint32_t p[5], t[5]; };
            // the strings as aligned dwords (realigned in registers when their item starts)
            auto strings = [&](const uint2 &d) -> Raw {   // bytes 8 .. 15 only when some lane of the item has that many
                Raw r{};
                const uint32_t pat = d.x & 0xFFFFu, txt = d.x >> 16;
                const uint32_t *pp = (const uint32_t *)((const uint8_t *)&lds + (pat & ~3u)), *tp = (const uint32_t *)((const uint8_t *)&lds + (txt & ~3u));
#pragma unroll
                for (int k = 0; k < 3; ++k) { r.p[k] = pp[k]; r.t[k] = tp[k]; }
                if (__builtin_amdgcn_ballot_w64((d.y & 0xFFu) > 8u)) { r.p[3] = pp[3]; r.p[4] = pp[4]; }
                if (__builtin_amdgcn_ballot_w64(((d.y >> 8) & 0xFFu) > 8u)) { r.t[3] = tp[3]; r.t[4] = tp[4]; }
                return r;
            };
            uint32_t round = 0, t = ticket_of(0);
            uint2 d = make_uint2(0u, 0u);
            Raw raw{};
            if (t < items) { d = descriptor(t); raw = strings(d); }
            while (t < items) {
                __builtin_amdgcn_s_setprio(1);
                const uint32_t item = items - 1 - t;
                const bool active = item * 64 + (uint32_t)lane < total;
                const uint32_t pat = d.x & 0xFFFFu, txt = d.x >> 16;
                const uint32_t lens = active ? d.y : 0u;   // no rows, no columns
                const uint32_t m = lens & 0xFFu, n = (lens >> 8) & 0xFFu, q = d.y >> 16;
                const uint32_t last = total - item * 64 - 1 < 63u ? total - item * 64 - 1 : 63u;
                const uint32_t n_max = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)last);   // sorted by text length first
                const uint32_t m_max = wave_max_u32(m);
                uint32_t pw[4], tw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pw[k] = __builtin_amdgcn_alignbyte(raw.p[k + 1], raw.p[k], pat);
                    tw[k] = __builtin_amdgcn_alignbyte(raw.t[k + 1], raw.t[k], txt);
                }
                const uint32_t rows = ((1u << m) - 1u) << half_shift;
                // rows past a lane's pattern OR in a zero: no branch per lane, one scalar test per row
#define SWH_SHORT_ROW(W, U) { if ((uint32_t)(W) * 4 + (U) >= m_max) break; \
                              if constexpr (kOne) nib.template insert_one<U>(pw[W], rows & (0x00010001u << ((W) * 4 + (U)))); \
                              else nib.template insert<U>(pw[W], rows & (0x00010001u << ((W) * 4 + (U)))); }
                do {
                    SWH_SHORT_ROW(0, 0) SWH_SHORT_ROW(0, 1) SWH_SHORT_ROW(0, 2) SWH_SHORT_ROW(0, 3)
                    SWH_SHORT_ROW(1, 0) SWH_SHORT_ROW(1, 1) SWH_SHORT_ROW(1, 2) SWH_SHORT_ROW(1, 3)
                    SWH_SHORT_ROW(2, 0) SWH_SHORT_ROW(2, 1) SWH_SHORT_ROW(2, 2) SWH_SHORT_ROW(2, 3)
                    SWH_SHORT_ROW(3, 0) SWH_SHORT_ROW(3, 1) SWH_SHORT_ROW(3, 2) SWH_SHORT_ROW(3, 3)
                } while (false);
#undef SWH_SHORT_ROW
                short_lds_order();
                // the match words of columns 0..3 (4..7 when some text is that long), requested together: columns past a lane's text
                // look up whatever bytes follow it -- any entry of the table will do, nobody uses the word. (Nibble tables: the two
                // halves are ANDed when their column runs, not here, where it would wait for them.)
                uint32_t eq[8], eh[8];
#define SWH_SHORT_LOOKUP(C, W, U) { if constexpr (kOne) { eq[C] = nib.template lookup_one<U>(tw[W]); eh[C] = 0xFFFFFFFFu; } \
                                    else { eq[C] = *(const lds_u32 *)(uintptr_t)nib.template lo_addr<U>(tw[W]); eh[C] = *(const lds_u32 *)(uintptr_t)(nib.template hi_addr<U>(tw[W]) + 2048); } }
                SWH_SHORT_LOOKUP(0, 0, 0) SWH_SHORT_LOOKUP(1, 0, 1) SWH_SHORT_LOOKUP(2, 0, 2) SWH_SHORT_LOOKUP(3, 0, 3)
                eq[4] = eq[5] = eq[6] = eq[7] = 0; eh[4] = eh[5] = eh[6] = eh[7] = 0;
                if (n_max > 4) { SWH_SHORT_LOOKUP(4, 1, 0) SWH_SHORT_LOOKUP(5, 1, 1) SWH_SHORT_LOOKUP(6, 1, 2) SWH_SHORT_LOOKUP(7, 1, 3) }
#undef SWH_SHORT_LOOKUP
                // the next item's descriptor travels while the columns run
                const uint32_t t_next = ticket_of(++round);
                const bool more = t_next < items;
                uint2 d_next = make_uint2(0u, 0u);
                if (more) d_next = descriptor(t_next);
                // the recurrence runs in the low 16 bits of every lane (the looked-up word is shifted down); what a lane
                // < 32 sees above bit 15 are its partner's rows, and nothing ever moves down across bit 16
                __builtin_amdgcn_s_setprio(0);
                uint32_t pv = 0xFFFFFFFFu, mv = 0;
                // The item's pairs are sorted by text length first: lane 0 holds the shortest text, lane `last` the longest, and in
                // most items the two are equal or one apart. Columns below the shortest length run WITHOUT the per-lane test (a
                // compare, an exec-mask save / restore and a branch per column: a third of a column's issue slots; lanes past the
                // item's pairs compute something nobody reads); only the columns between the two lengths are predicated.
                const uint32_t n_min = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
                auto one_column = [&](uint32_t both) {
                    const uint32_t eqw = both >> half_shift;
                    const uint32_t xv = eqw | mv;
                    const uint32_t xh = (((eqw & pv) + pv) ^ pv) | eqw;
                    uint32_t ph = mv | ~(xh | pv);
                    const uint32_t mh = pv & xh;
                    ph = (ph << 1) | 1u;
                    pv = (mh + mh) | ~(xv | ph);   // (x + x: v_add_u32 issues in 2.5 cycles, a left shift in 4.4)
                    mv = ph & xv;
                };
                auto first_eight = [&]() {
#define SWH_SHORT_COLUMN(C) { if ((uint32_t)(C) >= n_max) return; \
                              const uint32_t both = kOne ? eq[C] : (eq[C] & eh[C]); \
                              if ((uint32_t)(C) < n_min) one_column(both); else if ((uint32_t)(C) < n) one_column(both); }
                    SWH_SHORT_COLUMN(0) SWH_SHORT_COLUMN(1) SWH_SHORT_COLUMN(2) SWH_SHORT_COLUMN(3)
                    SWH_SHORT_COLUMN(4) SWH_SHORT_COLUMN(5) SWH_SHORT_COLUMN(6) SWH_SHORT_COLUMN(7)
#undef SWH_SHORT_COLUMN
                };
                first_eight();
                if (n_max > 8) {   // (3 % of the synthetic words' residues): a look-up per column
                    auto late = [&]() {
                        auto step = [&](auto w_tag, auto u_tag) -> bool {   // false: past the longest text of the item
                            constexpr int w = decltype(w_tag)::value, u = decltype(u_tag)::value;
                            constexpr uint32_t col = (uint32_t)(w * 4 + u);
                            if (col >= n_max) return false;
                            uint32_t both;
                            if constexpr (kOne) both = nib.template lookup_one<u>(tw[w]);
                            else both = nib.template lookup<u>(tw[w]);
                            if (col < n_min) one_column(both);
                            else if (col < n) one_column(both);
                            return true;
                        };
#define SWH_SHORT_COLUMN(W, U) if (!step(std::integral_constant<int, W>{}, std::integral_constant<int, U>{})) return;
                        SWH_SHORT_COLUMN(2, 0) SWH_SHORT_COLUMN(2, 1) SWH_SHORT_COLUMN(2, 2) SWH_SHORT_COLUMN(2, 3)
                        SWH_SHORT_COLUMN(3, 0) SWH_SHORT_COLUMN(3, 1) SWH_SHORT_COLUMN(3, 2) SWH_SHORT_COLUMN(3, 3)
#undef SWH_SHORT_COLUMN
                    };
                    late();
                }
                // the next item's strings travel while this one's result is staged and the table cleared
                Raw raw_next{};
                if (more) raw_next = strings(d_next);
                if (active) {
                    const uint32_t mask = (1u << m) - 1u;
                    const uint32_t dist = n + __popc(pv & mask) - __popc(mv & mask);
                    lds.staged[q] = (uint8_t)clamp_bound(dist, bound);
                }
                short_lds_order();
                {
                    uint4 *tb = (uint4 *)lds.tables[wave];
#pragma unroll
                    for (int k = 0; k < 4; ++k) tb[k * 64 + lane] = make_uint4(0, 0, 0, 0);
                }
                short_lds_order();
                t = t_next; d = d_next; raw = raw_next;
#ifdef SWH_SHORT_PROFILE
                ++items_done;
#endif
            }
        };
        if (one_table) work_items(std::true_type{});
        else work_items(std::false_type{});
        __builtin_amdgcn_s_setprio(3);   // the steps around E are chains of round trips and barriers: they go first, E fills the gaps
        SHORT_STAMP(4);   // E
        __syncthreads();
        SHORT_STAMP(5);   // waiting for the other waves' items
        // ---- F: the chunk's distances, one coalesced sweep ------------------------------------------------------------------
        if (job.out_stride == 4 && !job.out_elem64) {
            uint32_t *dst = (uint32_t *)job.out + base;
#pragma unroll
            for (int k = 0; k < kShortPer; ++k) {
                const uint32_t q = (uint32_t)k * kShortThreads + threadIdx.x;
                if (q < len) store_out((char *)(dst + q), false, (int64_t)lds.staged[q]);
            }
        } else {
            char *chunk_out = job.out + base * job.out_stride;
#pragma unroll
            for (int k = 0; k < kShortPer; ++k) {
                const uint32_t q = (uint32_t)k * kShortThreads + threadIdx.x;
                if (q < len) {
                    char *dst = chunk_out + (uint64_t)q * job.out_stride;
                    const uint32_t v = lds.staged[q];
                    store_out(dst, job.out_elem64 != 0, (int64_t)v);
                }
            }
        }
        SHORT_STAMP(6);   // F
#ifdef SWH_SHORT_PROFILE
        ++chunk_no;
#endif
        // no barrier here: the next chunk rewrites `staged` in its step B, behind its first barrier, and what its step A
        // rewrites (segments, counters, ticket) nobody reads after step E
        cur = nxt;
        req = req_next;
    }
#ifdef SWH_SHORT_PROFILE
    if (lane == 0) {
        atomicAdd(&g_short_phase[7], 1ull);
        atomicAdd(&g_short_phase[8], items_done);
    }
#endif
    SHORT_WG_STAMP(1);
    // ---- the workgroup's work units -> the call summary (common.hpp) -------------------------------------------------------------
    if (threadIdx.x == 0) { lds.tail.cells = 0; lds.tail.syms = 0; lds.tail.maxa = 0; lds.tail.maxb = 0; lds.tail.misfit = 0; }
    __syncthreads();
    unsigned long long wcells = cells, wsyms = syms;
    misfit |= (maxa > (uint32_t)kShortMaxLen || maxb > (uint32_t)kShortMaxLen) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        wcells += __shfl_xor(wcells, off);
        wsyms += __shfl_xor(wsyms, off);
        misfit |= __shfl_xor(misfit, off);
        const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
        maxa = oa > maxa ? oa : maxa;
        maxb = ob > maxb ? ob : maxb;
    }
    if (lane == 0) {
        atomicAdd(&lds.tail.cells, wcells);
        atomicAdd(&lds.tail.syms, wsyms);
        atomicMax(&lds.tail.maxa, maxa);
        atomicMax(&lds.tail.maxb, maxb);
        atomicOr(&lds.tail.misfit, misfit);
    }
    __syncthreads();
    // every pair this kernel accepts is "short" (both sides <= 32): the count is the number of pairs this workgroup visited
    uint32_t visited = 0;
    for (uint32_t t = blockIdx.x; t < args.tiles; t += gridDim.x) {
        const uint64_t tile_base = (uint64_t)t * args.tile;
        visited += (uint32_t)(job.pairs - tile_base < args.tile ? job.pairs - tile_base : args.tile);
    }
    report_call_summary(PlanPartial{lds.tail.cells, lds.tail.syms, lds.tail.maxa, lds.tail.maxb, visited, lds.tail.misfit}, args.partials, args.done_counter,
                        args.summary, lds.tail.summary);
}

template <typename Off>
__global__ __launch_bounds__(kShortThreads, 4) void k_short_tiled(ShortArgs args) {
    __shared__ ShortLds lds;
    const Off *off_a = (const Off *)args.job.a.offsets, *off_b = (const Off *)args.job.b.offsets;
    if ((uint64_t)off_a[args.job.a.count] >= 16 && (uint64_t)off_b[args.job.b.count] >= 16) short_run<Off, true>(args, lds);
    else short_run<Off, false>(args, lds);
}

void launch_short_tiled(Scope *scope, const Job &job, uint32_t off64, uint32_t mean_bytes_x16) {
    int per_cu = kShortPerCu;
    uint32_t slots = (uint32_t)scope->compute_units * (uint32_t)per_cu;
    if (slots > (uint32_t)kMaxPartials) slots = kMaxPartials;
    // A chunk's segments must fit the LDS arrays: with strings of `mean` bytes a chunk holds kShortCap / mean pairs, less a
    // margin (a chunk that does not fit is halved by the kernel: correct, but the sort then works on half as many pairs).
    uint64_t chunk = kShortChunk;
    if (mean_bytes_x16) {
        const uint64_t fit = (uint64_t)kShortCap * 16 * 15 / 16 / mean_bytes_x16;
        if (fit < chunk) chunk = fit;
    }
    chunk &= ~(uint64_t)63;
    if (chunk < 256) chunk = 256;
    // every workgroup slot gets the same number of tiles; small batches use fewer workgroups, not tiles below 256 pairs
    const uint64_t rounds = (job.pairs + (uint64_t)slots * chunk - 1) / ((uint64_t)slots * chunk);
    uint64_t tile = (job.pairs + slots * rounds - 1) / (slots * rounds);
    tile = (tile + 63) & ~(uint64_t)63;
    if (tile < 256) tile = 256;
    if (tile > chunk) tile = chunk;
    ShortArgs args{};
    args.job = job;
    args.tile = (uint32_t)tile;
    args.tiles = (uint32_t)((job.pairs + tile - 1) / tile);
    args.partials = scope->plan_partials;
    args.done_counter = scope->done_counter;
    args.summary = scope->summary_target();
    const uint32_t blocks = args.tiles < slots ? args.tiles : slots;
    StampGuard guard(scope, "short_tiled");
    if (off64) hipLaunchKernelGGL(k_short_tiled<uint64_t>, dim3(blocks), dim3(kShortThreads), 0, scope->stream, args);
    else hipLaunchKernelGGL(k_short_tiled<uint32_t>, dim3(blocks), dim3(kShortThreads), 0, scope->stream, args);
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
