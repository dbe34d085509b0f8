// bp_dense.hpp -- a work item of k_bitparallel<u32> (code points) on a PER-PAIR DENSE ALPHABET.
//
// bp_item.hpp serves a code point's match vector from seven 3-bit-group tables: seven dependent LDS look-ups, seven address
// computations and six ANDs per DP column, where a byte costs two look-ups -- C3's lines run at 26 TCUPS as code points and 45 as
// bytes. But a PAIR never sees 2^21 symbols: a line of a thousand code points uses a hundred distinct ones. So a long item first
// gives every distinct symbol of a pair's pattern an 8-bit id and then is the byte item:
//
//   * dictionary: 251 slots of one word per pair in LDS (key = code point + 1, 0 = free), open addressing with double hashing (251
//     is prime: every step walks all slots), at most kDenseTries probes; the G lanes of a pair enter their 32 pattern symbols with
//     `ds_cmpst_rtn` -- THE SLOT THAT TAKES A SYMBOL IS ITS ID, so there is no ranking pass. Id 255 is never a slot: it is what a
//     text symbol that the pattern does not hold translates to, and no row's id has both nibbles 15, so EqLo[15] & EqHi[15] is the
//     empty match vector it needs;
//   * pattern rows enter the two nibble tables under their ids (NibbleTables of bp_window.hpp, 8 KB per wave);
//   * the text is translated ONCE PER PAIR, not once per lane: every sixteen steps sixteen lanes of the pair fetch one symbol each
//     (the generic item has all G lanes fetch all of it), look it up in the dictionary and write the id to a 128-byte ring in LDS;
//     lane `blk` reads the ids of steps s - blk .. s - blk + 3 as one `ds_read2_b32` + `v_alignbyte` (the ring's first word is
//     mirrored behind its last so that the second word never wraps).
//
// A pattern with more distinct symbols than its dictionary takes cheaply (a lane gives up after kDenseBudget failed probes over its
// 32 symbols: ~200 distinct symbols; lines of Chinese, mostly) overflows: the wave stops entering symbols at the next group of four,
// the item clears what it wrote and the caller runs it on the group tables as before (bitparallel.hip: a wave whose items keep
// overflowing tries only every eighth). The first version probed until the dictionary was full: 255 x 32 atomics per lane on a
// line that does not fit -- C3's synthetic four-script lines ran at 7 TCUPS instead of 22. Only items of G >= 16 blocks (patterns of 481+ symbols, at most four pairs
// per wave) come here: four dictionaries, the nibble tables and the rings fit the 14 KB a wave's group tables occupy, whichever
// half of a 4 KB page they start on -- no more LDS, the same eleven waves per compute unit.
//
// Definition matched: `LevenshteinDistancesUtf8` (bench.rs:386-399) == rapidfuzz over chars (bench.rs:416-419); the published rows
// this is for: similarities/README.md:39-40, :56 (XLSum lines through the UTF-8 engines).
#pragma once
#include "bp_item.hpp"

namespace swh {

constexpr uint32_t kDenseMinBlocks = 16;   // items of at least this many blocks: at most four pairs per wave
constexpr uint32_t kDenseSlots = 251;      // dictionary slots = ids 0 .. 250 (prime)
constexpr uint32_t kDenseTries = 48;       // probes before a symbol counts as "does not fit" (entering) / "not there" (looking up)
constexpr uint32_t kDenseBudget = 64;      // failed probes a lane spends on its 32 symbols before the pair counts as "too many symbols"
constexpr uint32_t kDenseSketchBits = 140; // bits of a pattern's 256-bit symbol sketch beyond which it is not tried (~200 distinct symbols)
constexpr uint32_t kDenseAbsent = 255;     // the id of a symbol the pattern does not hold
constexpr uint32_t kDenseRing = 128;       // ids in flight per pair: G - 1 behind the newest step, 32 ahead of it (G <= 64)

__device__ __forceinline__ uint32_t dense_hash(uint32_t c) {
    return __umulhi(c * 0x9E3779B1u, kDenseSlots);   // [0, 251)
}
__device__ __forceinline__ uint32_t dense_step(uint32_t c) {
    return 1u + (c & 127u);   // [1, 128]: symbols that share a first slot rarely share their low bits
}
__device__ __forceinline__ uint32_t dense_next(uint32_t h, uint32_t step) {
    const uint32_t n = h + step;
    return n >= kDenseSlots ? n - kDenseSlots : n;
}

// Where the pieces live in the 14 KB of a wave's group tables (which start on a 2 KB boundary, the odd waves' in the middle of a 4 KB
// page): [nibble tables 8 KB on a 4 KB boundary][four dictionaries of 1 KB][rings 2 KB] or [rings 2 KB][nibble tables][dictionaries].
struct DenseRegion {
    NibbleTables nib;
    uint32_t *nib_words;   // [32][64 lanes]
    uint32_t *dicts;   // [4][256]
    uint8_t *rings;    // [4][kDenseRing + 4]; 2 KB in all
    __device__ __forceinline__ void init(uint32_t *table, int lane) {
        char *const region = (char *)table;
        const bool on_page = ((uint32_t)(uintptr_t)(lds_u32 *)table & 4095u) == 0;
        nib_words = (uint32_t *)(region + (on_page ? 0 : 2048));
        nib.init(nib_words, lane);
        dicts = (uint32_t *)(region + (on_page ? 8192 : 10240));
        rings = (uint8_t *)(region + (on_page ? 12288 : 0));
    }
};

// A 256-bit sketch of a block's symbols (one bit per hashed symbol, `ds_or` without return: 32 in flight) and its population count:
// what sends a line of Chinese away after two LDS round trips -- entering its symbols until the probes run out costs a wave ~40 us of
// serial LDS atomics, a third of such an item.
__device__ __forceinline__ void dense_sketch_add(uint32_t *sketch, const uint32_t (&psym)[32], uint32_t brows) {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const uint32_t bit = __umulhi(psym[q] * 0x9E3779B1u, 256u);
        if ((uint32_t)q < brows) __hip_atomic_fetch_or(&sketch[bit >> 5], 1u << (bit & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ uint32_t dense_sketch_bits(const uint32_t *sketch) {
    uint32_t bits = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) bits += (uint32_t)__popc(sketch[w]);
    return bits;
}

// Enters a block's symbols into the pair's dictionary; pid = the rows' ids, four to a word. Eight symbols' first probes go out
// together: at the loads of real text most of them land, and a round trip of `ds_cmpst_rtn` is what this phase is made of. Returns
// true (wave-uniform) when some lane ran out of probes: the wave stops at the next group of eight.
__device__ __forceinline__ bool dense_enter(uint32_t *dict, const uint32_t (&psym)[32], uint32_t brows, uint32_t (&pid)[8]) {
    bool overflow = false;
    uint32_t budget = kDenseBudget;
#pragma unroll
    for (int q = 0; q < 8; ++q) pid[q] = 0;
#pragma unroll
    for (int q8 = 0; q8 < 4; ++q8) {
        uint32_t slot_of[8], found[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = 8 * q8 + i;
            slot_of[i] = dense_hash(psym[q]);
            found[i] = 0;
            if ((uint32_t)q < brows) found[i] = atomicCAS(&dict[slot_of[i]], 0u, psym[q] + 1);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = 8 * q8 + i;
            if ((uint32_t)q < brows) {
                const uint32_t key = psym[q] + 1, step = dense_step(psym[q]);
                uint32_t h = slot_of[i], old = found[i];
                for (uint32_t tries = 1; old != 0 && old != key; ++tries) {
                    if (tries == kDenseTries || budget == 0) { overflow = true; break; }
                    --budget;
                    h = dense_next(h, step);
                    old = atomicCAS(&dict[h], 0u, key);
                }
                pid[q >> 2] |= h << (8 * (q & 3));
            }
        }
        if (__ballot(overflow)) return true;
    }
    return false;
}

// A text symbol's id: the first probe's answer `k` (slot `h`) is in already -- it was requested a round of columns ago.
__device__ __forceinline__ uint32_t dense_translate(const uint32_t *dict, uint32_t c, uint32_t h, uint32_t k) {
    const uint32_t key = c + 1, step = dense_step(c);
    for (uint32_t tries = 1;; ++tries) {   // (a symbol that was entered sits within its first kDenseTries probes)
        if (k == key) return h;
        if (k == 0 || tries == kDenseTries) return kDenseAbsent;
        h = dense_next(h, step);
        k = dict[h];
    }
}
__device__ __forceinline__ void dense_ring_put(uint8_t *ring, uint32_t j, uint32_t id) {
    const uint32_t pos = j & (kDenseRing - 1);
    ring[pos] = (uint8_t)id;
    if (pos < 4) ring[kDenseRing + pos] = (uint8_t)id;   // the first word again behind the last: a reader's second word never wraps
}
// the ids of text positions j .. j + 3, one to a byte
__device__ __forceinline__ uint32_t dense_ring_ids(const uint8_t *ring, uint32_t j) {
    const uint32_t pos = j & (kDenseRing - 1);
    const uint32_t *const words = (const uint32_t *)ring;
    return __builtin_amdgcn_alignbyte(words[(pos >> 2) + 1], words[pos >> 2], pos & 3u);
}
__device__ __forceinline__ void dense_rows_in(const NibbleTables &nib, const uint32_t (&pid)[8], uint32_t brows) {
    const uint32_t row_mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (brows > (uint32_t)q * 4) {
            nib.template insert<0>(pid[q], row_mask & (1u << (q * 4 + 0)));
            nib.template insert<1>(pid[q], row_mask & (1u << (q * 4 + 1)));
            nib.template insert<2>(pid[q], row_mask & (1u << (q * 4 + 2)));
            nib.template insert<3>(pid[q], row_mask & (1u << (q * 4 + 3)));
        }
    }
}

// Returns false (wave-uniform) when a pair of the item has more distinct pattern symbols than a dictionary takes; the wave's
// tables are all zero again then, as they are after an item that ran.
__device__ __forceinline__ bool bp_item_dense(const KernelArgs &args, BpWave<uint32_t> &wv, const uint32_t G, const bool have, const uint64_t p,
                                              const uint64_t a0, const uint32_t la, const uint64_t b0, const uint32_t lb) {
    const int lane = wv.lane;
    uint32_t *const acc = wv.acc;
    DenseRegion region;
    region.init(wv.table, lane);
    const NibbleTables &nib = region.nib;
    const uint32_t slot = (uint32_t)lane / G, blk = (uint32_t)lane - slot * G;
    uint32_t *const dict = region.dicts + slot * 256;
    uint8_t *const ring = region.rings + slot * (kDenseRing + 4);
    auto give_up = [&]() {
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < BpTraits<uint32_t>::kEntries; ++k) wv.table[k * 64 + lane] = 0;
        wave_lds_fence();
        return false;
    };

    const bool a_is_pattern = bp_pattern_is_a(la, lb);
    const uint32_t m = a_is_pattern ? la : lb, n = a_is_pattern ? lb : la;
    SymWindow32 pat, txt;
    pat.init((const uint32_t *)(a_is_pattern ? args.job.a.data : args.job.b.data), a_is_pattern ? a0 : b0, a_is_pattern ? wv.a_total : wv.b_total);
    txt.init((const uint32_t *)(a_is_pattern ? args.job.b.data : args.job.a.data), a_is_pattern ? b0 : a0, a_is_pattern ? wv.b_total : wv.a_total);
    const uint32_t row0 = blk * 32;
    const uint32_t brows = have ? (m > row0 ? (m - row0 < 32 ? m - row0 : 32) : 0) : 0;
    const bool translator = have && blk < 16;   // the lanes that translate the pair's text, one symbol each per sixteen steps

    // the first two rounds of text and the block's pattern symbols: all requested before anything is waited for
    uint32_t tsym = translator ? txt.fetch((int)blk) : 0u, tsym_next = translator ? txt.fetch(16 + (int)blk) : 0u;
    uint32_t psym[32];
#pragma unroll
    for (int q = 0; q < 32; q += 4) {
        uint32_t four[4];
        pat.fetch4((int)row0 + q, four);
#pragma unroll
        for (int r = 0; r < 4; ++r) psym[q + r] = four[r];
    }
    // ---- is it worth trying? (the sketch borrows the ring's first words) ------------------------------------------------------
    dense_sketch_add((uint32_t *)ring, psym, brows);
    wave_lds_fence();
    if (__ballot(have && dense_sketch_bits((const uint32_t *)ring) > kDenseSketchBits)) return give_up();
    // ---- the pair's dictionary; my rows under their ids in the nibble tables -------------------------------------------------
    uint32_t pid[8];
    if (dense_enter(dict, psym, brows, pid)) return give_up();
    dense_rows_in(nib, pid, brows);
    acc[lane] = 0;
    wave_lds_fence();   // the dictionaries are complete; acc slots are accumulated into by other lanes below
    if (translator) dense_ring_put(ring, blk, dense_translate(dict, tsym, dense_hash(tsym), dict[dense_hash(tsym)]));
    wave_lds_fence();

    const uint32_t n_eff = wave_max_u32(have ? n + G - 1 : 0);
    const uint32_t steps = (n_eff + 15) & ~15u;
    const bool first_blk = blk == 0;
    uint32_t keep_mask = first_blk ? 0u : 0xFFFFFFFFu, first_ph = first_blk ? 0x80000000u : 0u;
    asm volatile("" : "+v"(keep_mask), "+v"(first_ph));
    uint32_t pv = 0xFFFFFFFFu, mv = 0, ph = 0, mh = 0;
    auto column = [&](uint32_t eq, uint32_t s) {   // bp_item.hpp's
        uint32_t ph_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ph, 0x138, 0xf, 0xf, true);
        uint32_t mh_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mh, 0x138, 0xf, 0xf, true);
        ph_in = (uint32_t)__builtin_amdgcn_bitop3_b32((int)ph_in, (int)keep_mask, (int)first_ph, 0xEA);
        mh_in = mh_in & keep_mask;
        if (s - blk < n) {
            uint32_t xv = eq | mv;
            eq |= mh_in >> 31;
            uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
            ph = mv | ~(xh | pv);
            mh = pv & xh;
            uint32_t ph_s = __builtin_amdgcn_alignbit(ph, ph_in, 31);
            uint32_t mh_s = __builtin_amdgcn_alignbit(mh, mh_in, 31);
            pv = mh_s | ~(xv | ph_s);
            mv = ph_s & xv;
        }
    };
    for (uint32_t s0 = 0; s0 < steps; s0 += 16) {
        // the ids of this round's sixteen steps (translated one round ago), requested together
        uint32_t ids[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ids[q] = dense_ring_ids(ring, s0 + 4 * q - blk);
        // the next round's symbols become ids at the END of this round: their first probe goes out now (every lane: the others ask for
        // the slot of symbol 0) and has long answered by then; the round after that is requested from memory
        const uint32_t coming = tsym_next;
        if (translator) tsym_next = txt.fetch((int)s0 + 32 + (int)blk);
        const uint32_t first_slot = dense_hash(coming);
        const uint32_t first_key = dict[first_slot];
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the read next to its use)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t gs = s0 + q * 4;
            if (gs >= n_eff) break;
            uint32_t eqs[4];
            eqs[0] = nib.template lookup<0>(ids[q]);
            eqs[1] = nib.template lookup<1>(ids[q]);
            eqs[2] = nib.template lookup<2>(ids[q]);
            eqs[3] = nib.template lookup<3>(ids[q]);
#pragma unroll
            for (int u = 0; u < 4; ++u) column(eqs[u], gs + u);
        }
        if (translator) dense_ring_put(ring, s0 + 16 + blk, dense_translate(dict, coming, first_slot, first_key));
        wave_lds_fence();   // the ring's new ids are read from the next round on
    }

    const uint32_t mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
    const int part = __popc(pv & mask) - __popc(mv & mask);
    if (have && brows) atomicAdd(&acc[slot * G], (uint32_t)part);
    wave_lds_fence();
    if (have && first_blk) {
        const uint32_t d = n + acc[lane];
        store_result(args.job, p, (int64_t)clamp_bound(d, args.job.bound));
    }
#pragma unroll
    for (int k = 0; k < BpTraits<uint32_t>::kEntries; ++k) wv.table[k * 64 + lane] = 0;
    wave_lds_fence();
    return true;
}

}  // namespace swh
