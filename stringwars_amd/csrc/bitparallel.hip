// bitparallel.hip -- unit-cost Levenshtein as Myers/Hyyro bit-vectors on gfx950.
//
// The shorter string of a pair (the "pattern", m symbols) is cut into G = ceil(m/32) blocks of 32
// DP rows. Each block is owned by ONE lane as two 32-bit vertical-delta words (Pv, Mv); the G lanes
// of a pair form a systolic array: at global step s the lane holding block k consumes text symbol
// t = s - k and hands the horizontal delta of its last row (the top bit of Ph / Mh) to the lane
// above through one `wave_shr:1` DPP move, i.e. the blocks of a pair sit on an anti-diagonal of
// 32x1 tiles. A wave64 therefore carries floor(64/G) pairs at once (64 pairs for words of up to
// 32 symbols, one pair for a 2048-symbol line), and one step costs ~25 VALU instructions for up
// to 64 x 32 DP cells: `v_bitop3_b32` folds Hyyro's boolean recurrences three inputs at a time.
//
// Match vectors. Eq(c) = { j : P[j] == c } is never tabulated per symbol (256 x 4 B per lane would
// cap a CU at ~2 waves). A position matches iff both nibbles match, so
//        Eq(c) = EqLo[c & 15] & EqHi[c >> 4]
// with two 16-entry tables per lane: 2 x 16 x 64 lanes x 4 B = 8 KB of LDS per wave for the FULL
// byte alphabet, lane-interleaved ([nibble][lane]: bank = lane % 32, conflict-free for any symbol
// mix). Built with `ds_or_b32`, read twice per step, cleared by 32 immediate-offset `ds_write_b32`.
// Sixteen waves per CU fit (four per SIMD), which is what hides HBM / LDS latency here.
//
// Distance = n + popcount(Pv) - popcount(Mv) summed over the pair's blocks after the last text
// symbol (D[m][n] = D[0][n] + sum of the vertical deltas of the last column): no per-step score.
// Work items (chunks of floor(64/G) same-class pairs, sorted by text length by the pre-pass) are dealt to
// workgroups by index and inside a workgroup by an LDS ticket (bp_run below). Not by one global ticket: a
// single word saturates near 88 dequeues/us (MI355X_MICROARCH.md "dequeue"), slower than the DP itself on
// short items -- k_bitparallel_long, whose items take milliseconds, does use one.
// Definition matched: `rapidfuzz::distance::levenshtein::distance` (bench.rs:416-419).
#include "bp_dense.hpp"

namespace swh {

#ifdef SWH_TEST_HOOKS
// (test library) [0]: items that ran on a dense alphabet, [1]: items that were sent away / whose dictionary overflowed and went to the
// group tables; [2] / [3]: the same for the passes of k_bitparallel_long<u32>
__device__ uint32_t g_dense_items[4];
#endif

// kBpWaves: waves per workgroup. Byte strings run ONE 16-wave workgroup per compute unit whose waves take the CU's share
// of the items (item = blockIdx.x + turn * gridDim.x) from an LDS ticket: the hardware serves a CU's oldest waves first, so
// with a fixed list per wave the youngest were left to finish theirs alone, one wave per SIMD -- the regime where this
// serial recurrence issues at half rate (see tiled.hip). A ticket in LDS costs nothing next to an item (the global one
// this kernel started with saturated at 88 dequeues per microsecond).
// kDense: code-point items of kDenseMinBlocks blocks and more first try the per-pair dense alphabet (bp_dense.hpp).
template <typename Sym, int kBpWaves, bool kWide, bool kDense>
__device__ __forceinline__ void bp_run(const KernelArgs &args, char *smem, const uint64_t a_total, const uint64_t b_total) {
    constexpr int kBpTableWords = bp_table_words<Sym>();
    constexpr bool kTicket = kBpWaves > BpTraits<Sym>::kWaves;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t *item_prefix = (uint32_t *)smem + (size_t)kBpWaves * (kBpTableWords + 64);  // [65]
    BpWave<Sym> wv;
    wv.init((uint32_t *)smem + (size_t)wave_in_block * kBpTableWords,
            (uint32_t *)smem + (size_t)kBpWaves * kBpTableWords + wave_in_block * 64, lane, a_total, b_total);
    if (threadIdx.x < 64) {
        // work items per class g = lane + 1: ceil(count / floor(64 / g)); exclusive prefix across the wave
        const uint32_t g = threadIdx.x + 1, per = 64 / g;
        const uint32_t cnt = args.plan->class_count[kClassBp0 + threadIdx.x];
        const uint32_t mine = (cnt + per - 1) / per;
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t up = __shfl_up(incl, off);
            if ((int)threadIdx.x >= off) incl += up;
        }
        item_prefix[threadIdx.x] = incl - mine;
        if (threadIdx.x == 63) { item_prefix[64] = incl; item_prefix[66] = 0; item_prefix[67] = 0; item_prefix[68] = 0; }   // [66]: the workgroup's ticket
    }
    __syncthreads();
    const uint32_t items_total = item_prefix[64];
    const uint32_t my_prefix = item_prefix[lane];
    const uint32_t waves_total = gridDim.x * kBpWaves;
    const uint32_t wave_id = blockIdx.x * kBpWaves + wave_in_block;

    for (uint32_t w = wave_id;; w += waves_total) {
        if constexpr (kTicket) {
            uint32_t turn = 0;
            if (lane == 0) turn = __hip_atomic_fetch_add(&item_prefix[66], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            w = blockIdx.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)turn) * gridDim.x;
        }
        if (w >= items_total) break;
        const uint32_t item = items_total - 1 - w;  // heavy classes (many blocks, long texts) first
        // class = number of prefix entries <= item (prefix is non-decreasing, prefix[0] = 0)
        const uint32_t G = (uint32_t)__popcll(__ballot(my_prefix <= item));
        const uint32_t per = 64 / G;  // pairs per wave
        const uint32_t chunk = item - item_prefix[G - 1];
        const uint32_t cls = kClassBp0 + G - 1;
        const uint32_t cstart = args.plan->class_start[cls], ccount = args.plan->class_count[cls];

        const uint32_t slot = (uint32_t)lane / G;
        const uint32_t pidx = chunk * per + slot;
        const bool have = slot < per && pidx < ccount;
        uint64_t p = 0, a0 = 0, b0 = 0;
        uint32_t la = 0, lb = 0;
        if (have) {
            p = args.perm[cstart + pidx];
            if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
            else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
        }
        if constexpr (sizeof(Sym) == 4 && kDense) {
            // (a workgroup whose items keep having too many symbols for their dictionaries -- lines of Chinese -- tries only every eighth: being
            // sent away costs an item its pattern fetch and two LDS round trips before the group tables start over; [67] / [68] of the
            // prefix array count the workgroup's items that ran dense / were sent away)
            if (G >= kDenseMinBlocks) {
                const uint32_t ran = item_prefix[67], failed = item_prefix[68];
                bool dense = false;
                if (failed < 3 || failed <= 2 * ran || (failed & 7u) == 0) {
                    dense = bp_item_dense(args, wv, G, have, p, a0, la, b0, lb);
#ifdef SWH_TEST_HOOKS
                    if (lane == 0) atomicAdd(&g_dense_items[dense ? 0 : 1], 1u);
#endif
                }
                if (lane == 0) __hip_atomic_fetch_add(&item_prefix[dense ? 67 : 68], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (dense) continue;
            }
        }
        bp_item<Sym, kWide>(args, wv, G, have, p, a0, la, b0, lb);
    }
}

template <typename Sym, int kBpWaves, bool kDense = true>
__global__ __launch_bounds__(kBpWaves * 64, BpTraits<Sym>::kMinWavesPerSimd) void k_bitparallel(KernelArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint64_t a_total = tape_total(args.job.a, args.off64);
    const uint64_t b_total = tape_total(args.job.b, args.off64);
    if constexpr (sizeof(Sym) == 1) {
        if (a_total >= 16 && b_total >= 16) bp_run<Sym, kBpWaves, true, false>(args, smem, a_total, b_total);
        else bp_run<Sym, kBpWaves, false, false>(args, smem, a_total, b_total);
    } else {
        bp_run<Sym, kBpWaves, false, kDense>(args, smem, a_total, b_total);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Patterns of more than 64 blocks (> 2048 symbols): ONE pair per wave, the 64 lanes take 64 consecutive blocks and the
// text is walked once per pass of 64 blocks. The horizontal deltas leaving a pass's last block (one +1 and one -1
// bit per text column) are parked in a per-wave carry array and fed to the next pass's first block in place of the
// DP boundary. 32 columns per word, shifted in from the top by lane 63 and shifted out from the bottom by lane 0.
// ------------------------------------------------------------------------------------------------------------
template <typename Window>
__device__ __forceinline__ uint32_t window_symbol(const Window &w, int idx) {   // (code-point windows only; the byte kernel never calls it)
    if constexpr (std::is_same<Window, SymWindow32>::value) return w.fetch(idx);
    else return 0;
}

// Code points (kDense): a pass runs on the pair's dense alphabet (bp_dense.hpp) as long as the pattern's symbols so far fit its
// dictionary -- the dictionary and the sketch stay in LDS from pass to pass (a symbol that an earlier pass entered keeps its id, a
// text symbol that only a LATER pass's blocks hold finds this pass's nibble tables empty for its id); from the pass that does not
// fit on, the pair runs on the group tables.
// Code points: ONE eleven-wave workgroup per compute unit (what 14.25 KB of tables per wave leave room for, as in k_bitparallel<u32, 11>) and at most
// 168 registers, i.e. three waves on a SIMD: the kernel asked for 207 -- 256 with the dense passes -- and ran two waves per SIMD in five workgroups of
// two (eight waves per CU); capped, it spills ~80 registers outside its column loop and runs lines of ~3000 code points 20 % faster.
constexpr int kBpLongWavesU32 = 11;
template <typename Sym, bool kDense = true>
__global__ __launch_bounds__((sizeof(Sym) == 4 ? kBpLongWavesU32 : BpTraits<Sym>::kWaves) * 64, (sizeof(Sym) == 4 ? 3 : BpTraits<Sym>::kMinWavesPerSimd))
void k_bitparallel_long(KernelArgs args) {
    constexpr int kBpWaves = sizeof(Sym) == 4 ? kBpLongWavesU32 : BpTraits<Sym>::kWaves, kBpTableWords = bp_table_words<Sym>();
    constexpr bool kBytes = sizeof(Sym) == 1;
    constexpr bool kTryDense = !kBytes && kDense;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t *table = (uint32_t *)smem + (size_t)wave_in_block * kBpTableWords;  // [entries][64 lanes]
    [[maybe_unused]] NibbleTables nib;
    [[maybe_unused]] GroupTables3 grp;
    [[maybe_unused]] DenseRegion dense;
    if constexpr (kBytes) nib.init(table, lane);
    else grp.init(table, lane);
    if constexpr (kTryDense) dense.init(table, lane);
#pragma unroll
    for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
    const uint32_t cstart = args.plan->class_start[kClassBpLong], ccount = args.plan->class_count[kClassBpLong];
    const uint64_t a_total = tape_total(args.job.a, args.off64);
    const uint64_t b_total = tape_total(args.job.b, args.off64);
    const uint32_t waves_total = gridDim.x * kBpWaves;
    const uint32_t wave_id = blockIdx.x * kBpWaves + wave_in_block;
    // carry words of this wave: [parity of the producing pass][+1 bits | -1 bits][words]
    const uint32_t cwords = (uint32_t)(args.boundary_stride / 4);
    uint32_t *carry = (uint32_t *)args.boundary + (uint64_t)wave_id * args.boundary_stride;

    // lanes other than 0 take their horizontal input from the lane below; lane 0 from the boundary / the carry words
    uint32_t keep_mask = lane == 0 ? 0u : 0xFFFFFFFFu;
    asm volatile("" : "+v"(keep_mask));

    // One pair per wave and a few pairs per wave in all (10 K pairs of 4 KB on 4096 waves: 2.4): dealt round-robin, the last
    // round is half empty. The waves draw the next pair from one global ticket instead -- longest texts first, so the list
    // is scheduled longest-processing-time-first; a pair takes milliseconds, the ticket's 88 dequeues per microsecond
    // (bitparallel.hip, top) are not in the way here.
    for (uint32_t w = wave_id;; w += waves_total) {
        if (args.ticket) {
            uint32_t drawn = 0;
            if (lane == 0) drawn = __hip_atomic_fetch_add(args.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        }
        if (w >= ccount) break;
        const uint64_t p = args.perm[cstart + (ccount - 1 - w)];   // longest texts first
        uint64_t a0, b0;
        uint32_t la, lb;
        if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
        else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
        const bool a_is_pattern = bp_pattern_is_a(la, lb);
        const uint32_t m = a_is_pattern ? la : lb, n = a_is_pattern ? lb : la;
        using Window = typename std::conditional<kBytes, ByteWindow, SymWindow32>::type;
        Window pat, txt;
        pat.init((const Sym *)(a_is_pattern ? args.job.a.data : args.job.b.data), a_is_pattern ? a0 : b0,
                 a_is_pattern ? a_total : b_total);
        txt.init((const Sym *)(a_is_pattern ? args.job.b.data : args.job.a.data), a_is_pattern ? b0 : a0,
                 a_is_pattern ? b_total : a_total);
        const uint32_t blocks_total = (m + 31) >> 5, passes = (blocks_total + 63) >> 6;
        int part = 0;   // my blocks' share of the vertical deltas in the last column
        [[maybe_unused]] bool dense_ok = kTryDense;   // (wave-uniform) the pattern's symbols so far fit the pair's dictionary

        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t blocks_here = blocks_total - pass * 64 < 64 ? blocks_total - pass * 64 : 64;
            const bool my_block = (uint32_t)lane < blocks_here;
            const uint32_t row0 = (pass * 64 + (uint32_t)lane) * 32;
            const uint32_t brows = my_block ? (m - row0 < 32 ? m - row0 : 32) : 0;
            const bool record = pass + 1 < passes;   // a full pass of 64 blocks: lane 63 leaves the carries
            const uint32_t *cin_ph = carry + (size_t)((pass + 1) & 1) * 2 * cwords, *cin_mh = cin_ph + cwords;
            uint32_t *cout_ph = carry + (size_t)(pass & 1) * 2 * cwords, *cout_mh = cout_ph + cwords;
            const uint32_t n_eff = n + blocks_here - 1;
            const uint32_t steps = (n_eff + 15) & ~15u;

            // The pass's columns. kDensePass: the text arrives as ids through the pair's ring (sixteen lanes translate a symbol each per
            // round of sixteen steps), the match vectors come from the nibble tables; else every lane fetches its own text and looks it
            // up in the byte / group tables. `text_ahead` / `text_ahead2`: what was requested before the tables were built.
            // kCarryIn (not the first pass) / kRecord (not the last): compile-time, so that a pass pays only for what it does -- as run-time flags
            // they cost every column a uniform branch, four shifts, and two `v_cndmask` around the recording (36 instructions per step
            // against the plain item's 23). The carries travel oldest column at the TOP of a word: the producer shifts a column in from below
            // with one `v_alignbit` per word, the consumer hands the whole word to lane 0 (only bit 31 of what enters a block is looked at)
            // and shifts it left: two instructions per step on either side.
            auto columns_as = [&](auto dense_tag, auto carry_tag, auto record_tag, uint32_t (&tnxt)[kBytes ? 4 : 16], int (&tshift)[kBytes ? 4 : 1], uint32_t tsym_next) {
                constexpr bool kDensePass = decltype(dense_tag)::value, kCarryIn = decltype(carry_tag)::value, kRecord = decltype(record_tag)::value;
                constexpr int kTextRegs = kBytes ? 4 : 16;
                [[maybe_unused]] const uint32_t *const dict = kTryDense ? dense.dicts : nullptr;
                [[maybe_unused]] uint8_t *const ring = kTryDense ? dense.rings : nullptr;
                [[maybe_unused]] const bool translator = lane < 16;
                auto fetch_text = [&](int first) {
                    if constexpr (kBytes) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) tnxt[q] = txt.fetch4_raw(first + q * 4, tshift[q]);
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
                uint32_t cw_ph = 0, cw_mh = 0, cw_ph_next = 0, cw_mh_next = 0;   // carries entering lane 0, 32 columns per word
                if constexpr (kCarryIn) {
                    cw_ph_next = __hip_atomic_load(cin_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cw_mh_next = __hip_atomic_load(cin_mh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint32_t ow_ph = 0, ow_mh = 0;   // carries leaving lane 63
                uint32_t pv = 0xFFFFFFFFu, mv = 0, ph = 0, mh = 0;
                for (uint32_t s0 = 0; s0 < steps; s0 += 16) {
                    [[maybe_unused]] uint32_t tcur[kTextRegs];
                    [[maybe_unused]] uint32_t ids[4];
                    [[maybe_unused]] uint32_t coming = 0, first_slot = 0, first_key = 0;
                    if constexpr (kDensePass) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) ids[q] = dense_ring_ids(ring, s0 + 4 * q - (uint32_t)lane);
                        coming = tsym_next;
                        if (translator) tsym_next = window_symbol(txt, (int)s0 + 32 + lane);
                        first_slot = dense_hash(coming);
                        first_key = dict[first_slot];
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
#pragma unroll
                        for (int q = 0; q < kTextRegs; ++q) {
                            if constexpr (kBytes) tcur[q] = ByteWindow::realign(tnxt[q], tshift[q]);
                            else tcur[q] = tnxt[q];
                        }
                        fetch_text((int)s0 + 16 - lane);
                    }
                    if (kCarryIn && (s0 & 31u) == 0) {   // lane 0 is at column s0: the word for columns s0 .. s0 + 31
                        cw_ph = cw_ph_next;
                        cw_mh = cw_mh_next;
                        const uint32_t nxt = (s0 >> 5) + 1 < cwords ? (s0 >> 5) + 1 : cwords - 1;
                        cw_ph_next = __hip_atomic_load(cin_ph + nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cw_mh_next = __hip_atomic_load(cin_mh + nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t gs = s0 + q * 4;
                        if (gs >= n_eff) break;  // wave-uniform
                        uint32_t eqs[4];
                        if constexpr (kDensePass) {
                            eqs[0] = dense.nib.template lookup<0>(ids[q]);
                            eqs[1] = dense.nib.template lookup<1>(ids[q]);
                            eqs[2] = dense.nib.template lookup<2>(ids[q]);
                            eqs[3] = dense.nib.template lookup<3>(ids[q]);
                        } else if constexpr (kBytes) {
                            eqs[0] = nib.template lookup<0>(tcur[q]);
                            eqs[1] = nib.template lookup<1>(tcur[q]);
                            eqs[2] = nib.template lookup<2>(tcur[q]);
                            eqs[3] = nib.template lookup<3>(tcur[q]);
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) eqs[u] = grp.lookup(tcur[q * 4 + u]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t s = gs + u;
                            // what enters block 64 * pass in column s: the DP boundary (+1) or the previous pass's carries
                            uint32_t in_ph = 0x80000000u, in_mh = 0;
                            if constexpr (kCarryIn) {   // (bit 31 is this column's; the bits below it are not looked at)
                                in_ph = cw_ph; in_mh = cw_mh;
                                cw_ph <<= 1; cw_mh <<= 1;
                            }
                            uint32_t ph_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ph, 0x138, 0xf, 0xf, true);
                            uint32_t mh_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mh, 0x138, 0xf, 0xf, true);
                            ph_in = (uint32_t)__builtin_amdgcn_bitop3_b32((int)ph_in, (int)keep_mask, (int)in_ph, 0xE2);  // (a & b) | (c & ~b)
                            mh_in = (uint32_t)__builtin_amdgcn_bitop3_b32((int)mh_in, (int)keep_mask, (int)in_mh, 0xE2);
                            if (my_block && s - (uint32_t)lane < n) {
                                uint32_t eq = eqs[u];
                                uint32_t xv = eq | mv;
                                eq |= mh_in >> 31;
                                uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
                                ph = mv | ~(xh | pv);
                                mh = pv & xh;
                                uint32_t ph_s = __builtin_amdgcn_alignbit(ph, ph_in, 31);
                                uint32_t mh_s = __builtin_amdgcn_alignbit(mh, mh_in, 31);
                                pv = mh_s | ~(xv | ph_s);
                                mv = ph_s & xv;
                                if constexpr (kRecord) {   // only lane 63's words are stored
                                    ow_ph = __builtin_amdgcn_alignbit(ow_ph, ph, 31);   // (ow << 1) | (ph >> 31)
                                    ow_mh = __builtin_amdgcn_alignbit(ow_mh, mh, 31);
                                }
                            }
                            // lane 63 has just finished column s - 63: a word is complete every 32 columns
                            if (kRecord && s >= 63 && ((s - 63) & 31u) == 31u && s - 63 < n && lane == 63) {
                                cout_ph[(s - 63) >> 5] = ow_ph;
                                cout_mh[(s - 63) >> 5] = ow_mh;
                            }
                        }
                    }
                    if constexpr (kDensePass) {
                        if (translator) dense_ring_put(ring, s0 + 16 + (uint32_t)lane, dense_translate(dict, coming, first_slot, first_key));
                        wave_lds_fence();   // the ring's new ids are read from the next round on
                    }
                }
                if (kRecord && (n & 31u) && lane == 63) {   // the last, partial word: its columns sit at the bottom, the oldest goes to the top
                    cout_ph[n >> 5] = ow_ph << (32 - (n & 31u));
                    cout_mh[n >> 5] = ow_mh << (32 - (n & 31u));
                }
                const uint32_t mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
                part += __popc(pv & mask) - __popc(mv & mask);
            };
            auto columns = [&](auto dense_tag, uint32_t (&tnxt)[kBytes ? 4 : 16], int (&tshift)[kBytes ? 4 : 1], uint32_t tsym_next) {
                if (pass == 0 && record) columns_as(dense_tag, std::false_type{}, std::true_type{}, tnxt, tshift, tsym_next);
                else if (pass == 0) columns_as(dense_tag, std::false_type{}, std::false_type{}, tnxt, tshift, tsym_next);   // (never: one pass is k_bitparallel's)
                else if (record) columns_as(dense_tag, std::true_type{}, std::true_type{}, tnxt, tshift, tsym_next);
                else columns_as(dense_tag, std::true_type{}, std::false_type{}, tnxt, tshift, tsym_next);
            };

            uint32_t tnxt[kBytes ? 4 : 16];
            int tshift[kBytes ? 4 : 1];
            bool dense_pass = false;
            if constexpr (kBytes) {
                // ---- match tables of my block ------------------------------------------------------------------
#pragma unroll
                for (int q = 0; q < 4; ++q) tnxt[q] = txt.fetch4_raw(0 - lane + q * 4, tshift[q]);
                uint32_t praw[8];
                int pshift[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) praw[q] = pat.fetch4_raw((int)row0 + q * 4, pshift[q]);
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
                columns(std::false_type{}, tnxt, tshift, 0u);
            } else {
                uint32_t psym[32];
#pragma unroll
                for (int q = 0; q < 32; q += 4) {
                    uint32_t four[4];
                    pat.fetch4((int)row0 + q, four);
#pragma unroll
                    for (int r = 0; r < 4; ++r) psym[q + r] = four[r];
                }
                if constexpr (kTryDense) {
                    if (dense_ok) {
                        uint32_t *const dict = dense.dicts, *const sketch = dense.dicts + 256;   // (one pair per wave: the second dictionary's place)
                        const bool translator = lane < 16;
                        const uint32_t tsym = translator ? txt.fetch(lane) : 0u, tsym_next = translator ? txt.fetch(16 + lane) : 0u;
                        dense_sketch_add(sketch, psym, brows);
                        wave_lds_fence();
                        uint32_t pid[8];
                        if (__ballot(dense_sketch_bits(sketch) > kDenseSketchBits) || dense_enter(dict, psym, brows, pid)) {
                            dense_ok = false;   // from here on the group tables: what the dense passes left in LDS goes
                            wave_lds_fence();
#pragma unroll
                            for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
                            wave_lds_fence();
                        } else {
                            dense_pass = true;
                            dense_rows_in(dense.nib, pid, brows);
                            wave_lds_fence();
                            if (translator) dense_ring_put(dense.rings, (uint32_t)lane, dense_translate(dict, tsym, dense_hash(tsym), dict[dense_hash(tsym)]));
                            wave_lds_fence();
                            columns(std::true_type{}, tnxt, tshift, tsym_next);
                        }
                    }
                }
                if (!dense_pass) {
#pragma unroll
                    for (int q = 0; q < 16; q += 4) {
                        uint32_t four[4];
                        txt.fetch4(0 - lane + q, four);
#pragma unroll
                        for (int r = 0; r < 4; ++r) tnxt[q + r] = four[r];
                    }
#pragma unroll
                    for (int q = 0; q < 32; ++q)
                        if ((uint32_t)q < brows) grp.insert(psym[q], 1u << q);
                    columns(std::false_type{}, tnxt, tshift, 0u);
                }
            }
#ifdef SWH_TEST_HOOKS
            if (!kBytes && lane == 0) atomicAdd(&g_dense_items[dense_pass ? 2 : 3], 1u);
#endif
            bool tables_cleared = false;
            if constexpr (kTryDense) {
                if (dense_pass && pass + 1 < passes) {
                    // the next pass keeps the dictionary and the sketch: only the nibble tables go (the ring is rewritten before it is read)
#pragma unroll
                    for (int k = 0; k < 32; ++k) dense.nib_words[k * 64 + lane] = 0;
                    tables_cleared = true;
                }
            }
            if (!tables_cleared) {
#pragma unroll
                for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
            }
            __builtin_amdgcn_s_waitcnt(0);   // carries are in memory before the next pass asks for them
            wave_lds_fence();
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) store_result(args.job, p, (int64_t)clamp_bound(n + (uint32_t)part, args.job.bound));
    }
}

void launch_bitparallel_long(Scope *scope, KernelArgs args, const Plan &plan_host) {
    const uint32_t count = plan_host.class_count[kClassBpLong];
    if (!count) return;
    const bool bytes = args.sym_bytes == 1;
    const int waves = bytes ? BpTraits<uint8_t>::kWaves : kBpLongWavesU32;
    const size_t lds = bytes ? bp_lds_bytes<uint8_t>() : (size_t)kBpLongWavesU32 * (bp_table_words<uint32_t>() + 64) * 4 + 80 * 4;
    uint32_t blocks = (count + waves - 1) / waves;
    const uint32_t max_blocks = (uint32_t)scope->compute_units * (bytes ? 4 : 1);
    if (blocks > max_blocks) blocks = max_blocks;
    // args.boundary / boundary_stride: the carry words, sized by the caller (bp_long_carry_words, <= 4096 waves)
    static const bool round_robin = [] { const char *e = test_hook("STRINGWARS_AMD_LONG_TICKET"); return e && atoi(e) == 0; }();   // comparison knob
    args.ticket = nullptr;
    if (!round_robin && count > blocks * (uint32_t)waves) {   // more pairs than waves: somebody gets a second one
        args.ticket = scope->plan_leftover + 6;
        SWH_HIP_CHECK(hipMemsetAsync(args.ticket, 0, 4, scope->stream));
    }
    StampGuard guard(scope, bytes ? "bitparallel_long" : "bitparallel_long_u32");
    if (bytes) {
        opt_in_dynamic_lds(scope, (const void *)k_bitparallel_long<uint8_t>, lds);
        hipLaunchKernelGGL(k_bitparallel_long<uint8_t>, dim3(blocks), dim3(waves * 64), lds, scope->stream, args);
    } else {
#ifdef SWH_TEST_HOOKS
        if (const char *e = test_hook("STRINGWARS_AMD_BP_DENSE"); e && e[0] == '0') {   // the group tables for every pass: the tests' second implementation
            opt_in_dynamic_lds(scope, (const void *)k_bitparallel_long<uint32_t, false>, lds);
            hipLaunchKernelGGL((k_bitparallel_long<uint32_t, false>), dim3(blocks), dim3(waves * 64), lds, scope->stream, args);
            SWH_HIP_CHECK(hipGetLastError());
            return;
        }
#endif
        opt_in_dynamic_lds(scope, (const void *)k_bitparallel_long<uint32_t>, lds);
        hipLaunchKernelGGL(k_bitparallel_long<uint32_t>, dim3(blocks), dim3(waves * 64), lds, scope->stream, args);
    }
    SWH_HIP_CHECK(hipGetLastError());
}

template <typename Sym, int kWaves, bool kDense = true>
static void launch_bitparallel_sym(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    // The work list lives in the device plan; the host only bounds the grid (an item holds >= 1 pair).
    constexpr size_t lds = (size_t)kWaves * (bp_table_words<Sym>() + 64) * 4 + 80 * 4;
    KernelArgs k = args;
    k.boundary = nullptr;
    uint64_t blocks64 = (pairs + kWaves - 1) / kWaves;
    // as many workgroups as the compute units hold (bytes: one of 16 waves, or four of 4; code points: five of 2)
    const uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
    uint32_t max_blocks = (uint32_t)scope->compute_units * (per_cu ? per_cu : 1u);
    uint32_t blocks = blocks64 > max_blocks ? max_blocks : (uint32_t)blocks64;
    opt_in_dynamic_lds(scope, (const void *)k_bitparallel<Sym, kWaves, kDense>, lds);
    StampGuard guard(scope, sizeof(Sym) == 1 ? "bitparallel" : "bitparallel_u32");
    hipLaunchKernelGGL((k_bitparallel<Sym, kWaves, kDense>), dim3(blocks), dim3(kWaves * 64), lds, scope->stream, k);
    SWH_HIP_CHECK(hipGetLastError());
}

void launch_bitparallel(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    static const int forced = [] { const char *e = test_hook("STRINGWARS_AMD_BP_WAVES"); return e ? atoi(e) : 0; }();   // comparison knob: 4 (both widths), 10 (code points)
    if (args.sym_bytes == 4) {
        if (forced == 4) launch_bitparallel_sym<uint32_t, BpTraits<uint32_t>::kWaves>(scope, args, pairs);
        else if (forced == 10) launch_bitparallel_sym<uint32_t, 10>(scope, args, pairs);
        // 14.25 KB of tables and accumulators per wave: eleven waves are what a CU's 160 KB hold, with 2.9 KB to spare. Code points
        // are latency-bound (seven dependent look-ups per column): 8 / 10 / 11 waves measure 1.31 / 1.13 / 1.05 ms on C3's lines.
        // (Bytes are issue-bound: two workgroups of nine waves lose 8 % against the one of sixteen.)
#ifdef SWH_TEST_HOOKS
        else if (const char *e = test_hook("STRINGWARS_AMD_BP_DENSE"); e && e[0] == '0') launch_bitparallel_sym<uint32_t, 11, false>(scope, args, pairs);   // the group tables for every item: the tests' second implementation
#endif
        else launch_bitparallel_sym<uint32_t, 11>(scope, args, pairs);
    } else if (forced == 4) launch_bitparallel_sym<uint8_t, 4>(scope, args, pairs);
    else launch_bitparallel_sym<uint8_t, 16>(scope, args, pairs);
}

}  // namespace swh

#ifdef SWH_TEST_HOOKS
// (test library) reads and zeroes the current device's dense-alphabet item counters
extern "C" int swh_test_dense_items(uint32_t out[4]) {
    const uint32_t zero[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(swh::g_dense_items), sizeof zero) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(swh::g_dense_items), zero, sizeof zero) != hipSuccess;
}
#endif
