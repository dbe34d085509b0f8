// bp_item.hpp -- one work item of the bit-parallel Levenshtein kernels: a wave64 scores floor(64 / G) pairs whose
// patterns span up to G blocks of 32 rows, G lanes per pair (see bitparallel.hip for the scheme). Shared by the
// globally planned kernel (bitparallel.hip: items come from the device plan) and the tiled kernel (tiled.hip: every
// workgroup plans its own tile of consecutive pairs in LDS).
#pragma once
#include <type_traits>

#include "common.hpp"
#include "bp_window.hpp"

namespace swh {

// Code points (decoded UTF-8, u32, 21 bits) use the same trick with seven groups of three bits:
// Eq(c) = T0[c & 7] & T1[(c >> 3) & 7] & ... & T6[c >> 18], 7 x 8 entries = 14 KB per wave. (Four nibbles + a 32-entry
// plane table, 24 KB per wave, left 1.5 waves per SIMD -- the regime where a serial recurrence issues at half rate;
// two more lookups per column buy 2.5 waves per SIMD.)
struct SymWindow32 {
    const uint32_t *base;
    int lo, hi;
    __device__ __forceinline__ void init(const uint32_t *data, uint64_t start, uint64_t total) {
        auto c31 = [](int64_t v) { return (int)(v < -0x40000000ll ? -0x40000000ll : (v > 0x40000000ll ? 0x40000000ll : v)); };
        base = data + start;
        lo = c31(-(int64_t)start);
        hi = c31((int64_t)total - (int64_t)start - 1);
    }
    __device__ __forceinline__ uint32_t fetch(int idx) const { return base[bp_med3i(idx, lo, hi)]; }
    // Four consecutive symbols with one 128-bit load. The window is clamped into the tape as a whole; a window that had
    // to move (first / last symbols of a tape, tapes shorter than four symbols) is re-read symbol by symbol --
    // positions outside the tape then repeat the edge symbol, which the callers never use.
    __device__ __forceinline__ void fetch4(int idx, uint32_t (&out)[4]) const {
        if (hi - lo >= 3) {
            const int c = bp_med3i(idx, lo, hi - 3);
            uint4 v;
            __builtin_memcpy(&v, base + c, 16);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
            if (__builtin_expect(c == idx, 1)) return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = fetch(idx + q);
    }
};

template <typename Sym> struct BpTraits;
template <> struct BpTraits<uint8_t> {
    static constexpr int kWaves = 4, kEntries = 32, kMinWavesPerSimd = 4;
};
template <> struct BpTraits<uint32_t> {
    static constexpr int kWaves = 2, kEntries = 56, kMinWavesPerSimd = 2;
};
template <typename Sym> constexpr int bp_table_words() { return BpTraits<Sym>::kEntries * 64; }
template <typename Sym> constexpr size_t bp_lds_bytes() {
    return (size_t)BpTraits<Sym>::kWaves * (bp_table_words<Sym>() + 64) * 4 + 80 * 4;
}

// The code-point tables with NibbleTables' address arithmetic (bp_window.hpp): group g's entries start 2048 * g bytes
// into the wave's table (offset field of the ds instruction), the 3-bit value goes to bits 8..10 with one shift and
// one v_bitop3 -- the per-wave tables are 14 KB = 7 x 2 KB apart, so those bits of the base are zero.
struct GroupTables3 {
    uint32_t tbase, mask;
    __device__ __forceinline__ void init(uint32_t *table, int lane) {
        tbase = (uint32_t)(uintptr_t)(lds_u32 *)(table + lane);
        if (tbase & 0x700u) __builtin_trap();   // layout assumption (see NibbleTables)
        mask = 0x700u;
        asm volatile("" : "+v"(mask));
    }
    template <int G> __device__ __forceinline__ uint32_t addr(uint32_t c) const {
        uint32_t s;
        if constexpr (3 * G <= 8) s = c << (8 - 3 * G);
        else s = c >> (3 * G - 8);
        return (uint32_t)__builtin_amdgcn_bitop3_b32((int)s, (int)mask, (int)tbase, 0xEA);
    }
    __device__ __forceinline__ uint32_t lookup(uint32_t c) const { return lookup_n<7>(c); }
    __device__ __forceinline__ void insert(uint32_t c, uint32_t bit) const { insert_n<7>(c, bit); }
    // The first NG groups only -- for an item none of whose PATTERN symbols has a bit at or above 3 NG (bp_item: Latin, Cyrillic, Greek,
    // Arabic, Devanagari ... need four groups, the BMP six). A TEXT symbol with such a bit matches nothing: the last group's index is
    // clamped to 8, which is entry 0 of the group behind it -- a group this item never enters anything into, so its entries are zero.
    template <int NG, int G> __device__ __forceinline__ uint32_t entry_n(uint32_t c) const {
        if constexpr (G + 1 == NG && NG < 7) {
            uint32_t idx = c >> (3 * G);
            idx = idx < 8u ? idx : 8u;
            return *(const lds_u32 *)(uintptr_t)(tbase + (idx << 8) + 2048u * G);
        } else {
            return *(const lds_u32 *)(uintptr_t)(addr<G>(c) + 2048u * G);
        }
    }
    template <int NG> __device__ __forceinline__ uint32_t lookup_n(uint32_t c) const {
        uint32_t e = entry_n<NG, 0>(c) & entry_n<NG, 1>(c);
        e &= entry_n<NG, 2>(c) & entry_n<NG, 3>(c);
        if constexpr (NG > 4) e &= entry_n<NG, 4>(c);
        if constexpr (NG > 5) e &= entry_n<NG, 5>(c);
        if constexpr (NG > 6) e &= entry_n<NG, 6>(c);
        return e;
    }
    template <int NG> __device__ __forceinline__ void insert_n(uint32_t c, uint32_t bit) const {
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)addr<0>(c), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<1>(c) + 2048), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<2>(c) + 4096), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<3>(c) + 6144), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (NG > 4) __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<4>(c) + 8192), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (NG > 5) __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<5>(c) + 10240), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (NG > 6) __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<6>(c) + 12288), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// Per-wave state that outlives an item: the wave's match tables (zeroed between items), its accumulator row.
template <typename Sym>
struct BpWave {
    static constexpr bool kBytes = sizeof(Sym) == 1;
    uint32_t *table;   // [entries][64 lanes]
    uint32_t *acc;     // [64]
    NibbleTables nib;
    GroupTables3 grp;
    int lane;
    uint64_t a_total, b_total;
    __device__ __forceinline__ void init(uint32_t *table_, uint32_t *acc_, int lane_, uint64_t a_total_, uint64_t b_total_) {
        table = table_; acc = acc_; lane = lane_; a_total = a_total_; b_total = b_total_;
        if constexpr (kBytes) nib.init(table, lane);
        else grp.init(table, lane);
#pragma unroll
        for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
    }
};

// kWide: byte tapes of at least 16 bytes each, read with 128-bit loads (bp_window.hpp). A compile-time switch, because
// a run-time one inside the loops makes the two variants' registers merge right after the loads, i.e. puts a full
// memory wait next to every prefetch.
// Lane `lane` serves block `lane % G` of the pair in slot `lane / G`; `have` says whether that slot holds a pair
// (p, a0, la, b0, lb). G may exceed the blocks a pair needs (lanes past the pattern idle).
// `staged` (optional): an LDS array of 16-bit slots the caller drains later -- the distance of pair p goes to staged[p - staged_base]
// instead of straight to the output (tiled.hip writes a tile's results as one coalesced sweep).
template <typename Sym, bool kWide>
__device__ __forceinline__ void bp_item(const KernelArgs &args, BpWave<Sym> &wv, const uint32_t G, const bool have, const uint64_t p,
                                        const uint64_t a0, const uint32_t la, const uint64_t b0, const uint32_t lb,
                                        uint16_t *staged = nullptr, const uint64_t staged_base = 0) {
    constexpr bool kBytes = sizeof(Sym) == 1;
    constexpr bool wide_tapes = kWide;
    const int lane = wv.lane;
    uint32_t *const table = wv.table;
    uint32_t *const acc = wv.acc;
    [[maybe_unused]] const NibbleTables &nib = wv.nib;
    [[maybe_unused]] const GroupTables3 &grp = wv.grp;
    const uint64_t a_total = wv.a_total, b_total = wv.b_total;
    const uint32_t slot = (uint32_t)lane / G, blk = (uint32_t)lane - slot * G;
    // pattern = rows / bits / lanes, text = columns / steps: the cheaper of the two assignments (common.hpp)
    const bool a_is_pattern = bp_pattern_is_a(la, lb);
    const uint32_t m = a_is_pattern ? la : lb, n = a_is_pattern ? lb : la;
    using Window = typename std::conditional<kBytes, ByteWindow, SymWindow32>::type;
    Window pat, txt;
    pat.init((const Sym *)(a_is_pattern ? args.job.a.data : args.job.b.data), a_is_pattern ? a0 : b0,
             a_is_pattern ? a_total : b_total);
    txt.init((const Sym *)(a_is_pattern ? args.job.b.data : args.job.a.data), a_is_pattern ? b0 : a0,
             a_is_pattern ? b_total : a_total);

    // rows of my block
    const uint32_t row0 = blk * 32;
    const uint32_t brows = have ? (m > row0 ? (m - row0 < 32 ? m - row0 : 32) : 0) : 0;

    // text prefetch: 16 symbols per super-step, one super-step ahead (bytes: 4 dwords; code points: 16)
    constexpr int kTextRegs = kBytes ? 4 : 16;
    // byte words arrive unaligned-corrected only when they are consumed (`realign` next to the load would put the
    // memory latency on the critical path of every super-step)
    uint32_t tnxt[kTextRegs];
    int tshift[kBytes ? 4 : 1];
    auto fetch_text = [&](int first) {
        if constexpr (kBytes) {
            if (wide_tapes) {
                tshift[0] = txt.fetch16_raw(first, tnxt);   // [0] = distance the clamp moved the window
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) tnxt[q] = txt.fetch4_raw(first + q * 4, tshift[q]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                uint32_t four[4];
                txt.fetch4(first + q, four);
#pragma unroll
                for (int r = 0; r < 4; ++r) tnxt[q + r] = four[r];
            }
        }
    };
    fetch_text(0 - (int)blk);
    [[maybe_unused]] uint32_t groups = 7;   // (code points) 3-bit groups of the tables this item uses

    // ---- build the match tables of my block -------------------------------------------------
    if constexpr (kBytes) {
        // all eight words of the block are requested at once (clamped addresses are always readable): one memory
        // latency per item instead of one per word
        uint32_t praw[8];
        int pshift[8];
        if (wide_tapes) {
            uint32_t half[2][4];
            const int moved0 = pat.fetch16_raw((int)row0, half[0]), moved1 = pat.fetch16_raw((int)row0 + 16, half[1]);
            pat.fix16((int)row0, moved0, half[0]);
            pat.fix16((int)row0 + 16, moved1, half[1]);
#pragma unroll
            for (int q = 0; q < 8; ++q) { praw[q] = half[q >> 2][q & 3]; pshift[q] = 24; }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) praw[q] = pat.fetch4_raw((int)row0 + q * 4, pshift[q]);
        }
        // rows past the block's end OR in a zero (one predicated branch per word instead of one per byte)
        const uint32_t row_mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (brows > (uint32_t)q * 4) {
                const uint32_t dw = ByteWindow::realign(praw[q], pshift[q]);
                nib.template insert<0>(dw, row_mask & (1u << (q * 4 + 0)));
                nib.template insert<1>(dw, row_mask & (1u << (q * 4 + 1)));
                nib.template insert<2>(dw, row_mask & (1u << (q * 4 + 2)));
                nib.template insert<3>(dw, row_mask & (1u << (q * 4 + 3)));
            }
        }
    } else {
        uint32_t psym[32];   // the block's symbols, eight 128-bit loads in flight before the first table update
#pragma unroll
        for (int q = 0; q < 32; q += 4) {
            uint32_t four[4];
            pat.fetch4((int)row0 + q, four);
#pragma unroll
            for (int r = 0; r < 4; ++r) psym[q + r] = four[r];
        }
        // how many 3-bit groups this ITEM needs: none of its pattern symbols has a bit at or above 3 x groups (wave-uniform)
        uint32_t seen = 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) seen |= (uint32_t)q < brows ? psym[q] : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) seen |= (uint32_t)__shfl_xor((int)seen, off);
        groups = seen < (1u << 12) ? 4u : (seen < (1u << 15) ? 5u : (seen < (1u << 18) ? 6u : 7u));
        auto enter = [&](auto ng_tag) {
            constexpr int NG = decltype(ng_tag)::value;
#pragma unroll
            for (int q = 0; q < 32; ++q)
                if ((uint32_t)q < brows) grp.template insert_n<NG>(psym[q], 1u << q);
        };
        if (groups == 4) enter(std::integral_constant<int, 4>{});
        else if (groups == 5) enter(std::integral_constant<int, 5>{});
        else if (groups == 6) enter(std::integral_constant<int, 6>{});
        else enter(std::integral_constant<int, 7>{});
    }
    acc[lane] = 0;
    wave_lds_fence();  // acc slots are accumulated into by other lanes below

    // wave-uniform step count (lane `blk` of a pair works in steps blk .. n + blk - 1)
    const uint32_t n_eff = wave_max_u32(have ? n + G - 1 : 0);
    const uint32_t steps = (n_eff + 15) & ~15u;

    // Lanes that start a pair take the DP boundary (+1 horizontal delta) instead of a neighbour. The masks are
    // made opaque so that the splice stays two plain bitwise ops (v_bitop3 / v_and issue in ~2.7 cycles); knowing
    // where they come from, the compiler turns it into two v_cndmask_e64 (4.4 cycles each and an SGPR-pair read).
    const bool first_blk = blk == 0;
    uint32_t keep_mask = first_blk ? 0u : 0xFFFFFFFFu, first_ph = first_blk ? 0x80000000u : 0u;
    asm volatile("" : "+v"(keep_mask), "+v"(first_ph));
    uint32_t pv = 0xFFFFFFFFu, mv = 0, ph = 0, mh = 0;
    // One DP column of this lane's block. (A variant without the per-lane range test for groups in which every lane
    // works was measured: fewer VALU instructions, slower kernel -- the test rides on the scalar unit for free.)
    auto column = [&](uint32_t eq, uint32_t s) {
        // bound_ctrl: lane 0 (no source lane) reads 0, so no `old` register has to be re-materialised per step
        uint32_t ph_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ph, 0x138, 0xf, 0xf, true);
        uint32_t mh_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mh, 0x138, 0xf, 0xf, true);
        ph_in = (uint32_t)__builtin_amdgcn_bitop3_b32((int)ph_in, (int)keep_mask, (int)first_ph, 0xEA);  // (a & b) | c
        mh_in = mh_in & keep_mask;
        if (s - blk < n) {
            uint32_t xv = eq | mv;
            eq |= mh_in >> 31;
            uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
            ph = mv | ~(xh | pv);
            mh = pv & xh;
            uint32_t ph_s = __builtin_amdgcn_alignbit(ph, ph_in, 31);  // (ph << 1) | hin(+1)
            uint32_t mh_s = __builtin_amdgcn_alignbit(mh, mh_in, 31);  // (mh << 1) | hin(-1)
            pv = mh_s | ~(xv | ph_s);
            mv = ph_s & xv;
        }
    };
    auto columns = [&](auto ng_tag) {
    [[maybe_unused]] constexpr int NG = decltype(ng_tag)::value;
    for (uint32_t s0 = 0; s0 < steps; s0 += 16) {
        uint32_t tcur[kTextRegs];
        if constexpr (kBytes) {
            if (wide_tapes) {
#pragma unroll
                for (int q = 0; q < 4; ++q) tcur[q] = tnxt[q];
                txt.fix16((int)s0 - (int)blk, tshift[0], tcur);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) tcur[q] = ByteWindow::realign(tnxt[q], tshift[q]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < kTextRegs; ++q) tcur[q] = tnxt[q];
        }
        // unconditional: clamped addresses are always readable, and a branch around the loads would make the
        // compiler wait for them right there (their registers merge with the skipped path's)
        fetch_text((int)s0 + 16 - (int)blk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t gs = s0 + q * 4;
            if (gs >= n_eff) break;  // wave-uniform: no lane has a symbol left in this group
            uint32_t eqs[4];
            if constexpr (kBytes) {
                eqs[0] = nib.template lookup<0>(tcur[q]);
                eqs[1] = nib.template lookup<1>(tcur[q]);
                eqs[2] = nib.template lookup<2>(tcur[q]);
                eqs[3] = nib.template lookup<3>(tcur[q]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) eqs[u] = grp.template lookup_n<NG>(tcur[q * 4 + u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) column(eqs[u], gs + u);
        }
    }
    };
    if constexpr (kBytes) columns(std::integral_constant<int, 7>{});
    else if (groups == 4) columns(std::integral_constant<int, 4>{});
    else if (groups == 5) columns(std::integral_constant<int, 5>{});
    else if (groups == 6) columns(std::integral_constant<int, 6>{});
    else columns(std::integral_constant<int, 7>{});

    // ---- distance = n + sum over blocks popcount(pv) - popcount(mv) -------------------------
    const uint32_t mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
    int part = __popc(pv & mask) - __popc(mv & mask);
    if (have && brows) atomicAdd(&acc[slot * G], (uint32_t)part);
    wave_lds_fence();  // the pair's first lane reads the sum of its blocks' contributions
    if (have && first_blk) {
        uint32_t d = n + acc[lane];
        const uint32_t bounded = clamp_bound(d, args.job.bound);
        if (staged && bounded < 0xFFFFu) staged[p - staged_base] = (uint16_t)bounded;   // (0xFFFF: "no distance here")
        else store_result(args.job, p, (int64_t)bounded);
    }
    // ---- clear my table column (code points: the groups this item entered symbols into) -----
    if constexpr (kBytes) {
#pragma unroll
        for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) table[k * 64 + lane] = 0;
        if (groups > 4) {
#pragma unroll
            for (int k = 32; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
        }
    }
    wave_lds_fence();
}

}  // namespace swh
