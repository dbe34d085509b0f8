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
// Work items (chunks of floor(64/G) same-class pairs, sorted by text length by the pre-pass) are
// dealt round-robin to waves -- no atomics: one ticket word saturates near 88 dequeues/us
// (MI355X_MICROARCH.md "dequeue"), slower than the DP itself.
// Definition matched: `rapidfuzz::distance::levenshtein::distance` (bench.rs:416-419).
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
    __device__ __forceinline__ uint32_t lookup(uint32_t c) const {
        uint32_t e = *(const lds_u32 *)(uintptr_t)addr<0>(c) & *(const lds_u32 *)(uintptr_t)(addr<1>(c) + 2048) &
                     *(const lds_u32 *)(uintptr_t)(addr<2>(c) + 4096);
        e &= *(const lds_u32 *)(uintptr_t)(addr<3>(c) + 6144) & *(const lds_u32 *)(uintptr_t)(addr<4>(c) + 8192);
        return e & *(const lds_u32 *)(uintptr_t)(addr<5>(c) + 10240) & *(const lds_u32 *)(uintptr_t)(addr<6>(c) + 12288);
    }
    __device__ __forceinline__ void insert(uint32_t c, uint32_t bit) const {
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)addr<0>(c), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<1>(c) + 2048), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<2>(c) + 4096), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<3>(c) + 6144), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<4>(c) + 8192), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<5>(c) + 10240), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(addr<6>(c) + 12288), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

#ifdef SWH_BP_PROFILE
// Diagnostic build only (make EXTRA=-DSWH_BP_PROFILE): summed wave cycles per phase of k_bitparallel.
__device__ unsigned long long g_bp_phase[10];
extern "C" void swh_debug_bp_phases(unsigned long long *out) {
    unsigned long long zero[10] = {};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bp_phase), sizeof(zero));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bp_phase), zero, sizeof(zero));
}
#define BP_STAMP(slot, waitmem)                                                          \
    do {                                                                                 \
        if (waitmem) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long now__ = __builtin_readcyclecounter();                         \
        phase_acc[slot] += now__ - phase_t;                                              \
        phase_t = now__;                                                                 \
    } while (0)
#else
#define BP_STAMP(slot, waitmem) do {} while (0)
#endif

// kWide: byte tapes of at least 16 bytes each, read with 128-bit loads (bp_window.hpp). A compile-time switch, because
// a run-time one inside the loops makes the two variants' registers merge right after the loads, i.e. puts a full
// memory wait next to every prefetch.
template <typename Sym, bool kWide>
__device__ __forceinline__ void bp_run(const KernelArgs &args, char *smem, const uint64_t a_total, const uint64_t b_total) {
    constexpr int kBpWaves = BpTraits<Sym>::kWaves, kBpTableWords = bp_table_words<Sym>();
    constexpr bool kBytes = sizeof(Sym) == 1;
    constexpr bool wide_tapes = kWide;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t *table = (uint32_t *)smem + (size_t)wave_in_block * kBpTableWords;  // [entries][64 lanes]
    uint32_t *acc = (uint32_t *)smem + (size_t)kBpWaves * kBpTableWords + wave_in_block * 64;
    uint32_t *item_prefix = (uint32_t *)smem + (size_t)kBpWaves * (kBpTableWords + 64);  // [65]
    [[maybe_unused]] NibbleTables nib;
    [[maybe_unused]] GroupTables3 grp;
    if constexpr (kBytes) nib.init(table, lane);
    else grp.init(table, lane);

#pragma unroll
    for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
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
        if (threadIdx.x == 63) item_prefix[64] = incl;
    }
    __syncthreads();
    const uint32_t items_total = item_prefix[64];
    const uint32_t my_prefix = item_prefix[lane];
    const uint32_t waves_total = gridDim.x * kBpWaves;
    const uint32_t wave_id = blockIdx.x * kBpWaves + wave_in_block;
#ifdef SWH_BP_PROFILE
    unsigned long long phase_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, phase_t = __builtin_readcyclecounter();
    const unsigned long long phase_t0 = phase_t;
    unsigned long long my_items = 0;
#endif

    for (uint32_t w = wave_id; w < items_total; w += waves_total) {
        const uint32_t item = items_total - 1 - w;  // heavy classes (many blocks, long texts) first
        // class = number of prefix entries <= item (prefix is non-decreasing, prefix[0] = 0)
        const uint32_t G = (uint32_t)__popcll(__ballot(my_prefix <= item));
        const uint32_t per = 64 / G;  // pairs per wave
        const uint32_t chunk = item - item_prefix[G - 1];
        const uint32_t cls = kClassBp0 + G - 1;
        const uint32_t cstart = args.plan->class_start[cls], ccount = args.plan->class_count[cls];

        const uint32_t slot = (uint32_t)lane / G, blk = (uint32_t)lane - slot * G;
        const uint32_t pidx = chunk * per + slot;
        const bool have = slot < per && pidx < ccount;
        uint64_t p = 0, a0 = 0, b0 = 0;
        uint32_t la = 0, lb = 0;
        if (have) {
            p = args.perm[cstart + pidx];
            if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
            else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
        }
        BP_STAMP(0, true);   // locate + perm + extents
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
#pragma unroll
            for (int q = 0; q < 32; ++q)
                if ((uint32_t)q < brows) grp.insert(psym[q], 1u << q);
        }
        acc[lane] = 0;
        wave_lds_fence();  // acc slots are accumulated into by other lanes below
        BP_STAMP(1, true);   // string loads + table build

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
                    for (int u = 0; u < 4; ++u) eqs[u] = grp.lookup(tcur[q * 4 + u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) column(eqs[u], gs + u);
            }
        }

        BP_STAMP(2, false);   // DP steps
        // ---- distance = n + sum over blocks popcount(pv) - popcount(mv) -------------------------
        const uint32_t mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
        int part = __popc(pv & mask) - __popc(mv & mask);
        if (have && brows) atomicAdd(&acc[slot * G], (uint32_t)part);
        wave_lds_fence();  // the pair's first lane reads the sum of its blocks' contributions
        if (have && first_blk) {
            uint32_t d = n + acc[lane];
            store_result(args.job, p, (int64_t)clamp_bound(d, args.job.bound));
        }
        // ---- clear my table column ---------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
        wave_lds_fence();
        BP_STAMP(3, true);   // result + clear
#ifdef SWH_BP_PROFILE
        ++my_items;
#endif
    }
#ifdef SWH_BP_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&g_bp_phase[k], phase_acc[k]);
        atomicAdd(&g_bp_phase[4], __builtin_readcyclecounter() - phase_t0);
        atomicAdd(&g_bp_phase[5], 1ull);
        atomicAdd(&g_bp_phase[6], my_items);
    }
#endif
}

template <typename Sym>
__global__ __launch_bounds__(BpTraits<Sym>::kWaves * 64, BpTraits<Sym>::kMinWavesPerSimd) void k_bitparallel(KernelArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint64_t a_total = args.off64 ? ((const uint64_t *)args.job.a.offsets)[args.job.a.count]
                                        : ((const uint32_t *)args.job.a.offsets)[args.job.a.count];
    const uint64_t b_total = args.off64 ? ((const uint64_t *)args.job.b.offsets)[args.job.b.count]
                                        : ((const uint32_t *)args.job.b.offsets)[args.job.b.count];
    if constexpr (sizeof(Sym) == 1) {
        if (a_total >= 16 && b_total >= 16) bp_run<Sym, true>(args, smem, a_total, b_total);
        else bp_run<Sym, false>(args, smem, a_total, b_total);
    } else {
        bp_run<Sym, false>(args, smem, a_total, b_total);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Patterns of more than 64 blocks (> 2048 symbols): ONE pair per wave, the 64 lanes take 64 consecutive blocks and the
// text is walked once per pass of 64 blocks. The horizontal deltas leaving a pass's last block (one +1 and one -1
// bit per text column) are parked in a per-wave carry array and fed to the next pass's first block in place of the
// DP boundary. 32 columns per word, shifted in from the top by lane 63 and shifted out from the bottom by lane 0.
// ------------------------------------------------------------------------------------------------------------
template <typename Sym>
__global__ __launch_bounds__(BpTraits<Sym>::kWaves * 64, BpTraits<Sym>::kMinWavesPerSimd) void k_bitparallel_long(KernelArgs args) {
    constexpr int kBpWaves = BpTraits<Sym>::kWaves, kBpTableWords = bp_table_words<Sym>();
    constexpr bool kBytes = sizeof(Sym) == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t *table = (uint32_t *)smem + (size_t)wave_in_block * kBpTableWords;  // [entries][64 lanes]
    [[maybe_unused]] NibbleTables nib;
    [[maybe_unused]] GroupTables3 grp;
    if constexpr (kBytes) nib.init(table, lane);
    else grp.init(table, lane);
#pragma unroll
    for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
    const uint32_t cstart = args.plan->class_start[kClassBpLong], ccount = args.plan->class_count[kClassBpLong];
    const uint64_t a_total = args.off64 ? ((const uint64_t *)args.job.a.offsets)[args.job.a.count]
                                        : ((const uint32_t *)args.job.a.offsets)[args.job.a.count];
    const uint64_t b_total = args.off64 ? ((const uint64_t *)args.job.b.offsets)[args.job.b.count]
                                        : ((const uint32_t *)args.job.b.offsets)[args.job.b.count];
    const uint32_t waves_total = gridDim.x * kBpWaves;
    const uint32_t wave_id = blockIdx.x * kBpWaves + wave_in_block;
    // carry words of this wave: [parity of the producing pass][+1 bits | -1 bits][words]
    const uint32_t cwords = (uint32_t)(args.boundary_stride / 4);
    uint32_t *carry = (uint32_t *)args.boundary + (uint64_t)wave_id * args.boundary_stride;

    // lanes other than 0 take their horizontal input from the lane below; lane 0 from the boundary / the carry words
    uint32_t keep_mask = lane == 0 ? 0u : 0xFFFFFFFFu;
    asm volatile("" : "+v"(keep_mask));

    for (uint32_t w = wave_id; w < ccount; w += waves_total) {
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

        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t blocks_here = blocks_total - pass * 64 < 64 ? blocks_total - pass * 64 : 64;
            const bool my_block = (uint32_t)lane < blocks_here;
            const uint32_t row0 = (pass * 64 + (uint32_t)lane) * 32;
            const uint32_t brows = my_block ? (m - row0 < 32 ? m - row0 : 32) : 0;
            const bool record = pass + 1 < passes;   // a full pass of 64 blocks: lane 63 leaves the carries
            const uint32_t *cin_ph = carry + (size_t)((pass + 1) & 1) * 2 * cwords, *cin_mh = cin_ph + cwords;
            uint32_t *cout_ph = carry + (size_t)(pass & 1) * 2 * cwords, *cout_mh = cout_ph + cwords;

            constexpr int kTextRegs = kBytes ? 4 : 16;
            uint32_t tnxt[kTextRegs];
            int tshift[kBytes ? 4 : 1];
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
            fetch_text(0 - lane);
            // ---- match tables of my block ------------------------------------------------------------------
            if constexpr (kBytes) {
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
            } else {
                uint32_t psym[32];
#pragma unroll
                for (int q = 0; q < 32; q += 4) {
                    uint32_t four[4];
                    pat.fetch4((int)row0 + q, four);
#pragma unroll
                    for (int r = 0; r < 4; ++r) psym[q + r] = four[r];
                }
#pragma unroll
                for (int q = 0; q < 32; ++q)
                    if ((uint32_t)q < brows) grp.insert(psym[q], 1u << q);
            }
            const uint32_t n_eff = n + blocks_here - 1;
            const uint32_t steps = (n_eff + 15) & ~15u;
            uint32_t cw_ph = 0, cw_mh = 0, cw_ph_next = 0, cw_mh_next = 0;   // carries entering lane 0, 32 columns per word
            if (pass) {
                cw_ph_next = __hip_atomic_load(cin_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cw_mh_next = __hip_atomic_load(cin_mh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t ow_ph = 0, ow_mh = 0;   // carries leaving lane 63
            uint32_t pv = 0xFFFFFFFFu, mv = 0, ph = 0, mh = 0;
            for (uint32_t s0 = 0; s0 < steps; s0 += 16) {
                uint32_t tcur[kTextRegs];
#pragma unroll
                for (int q = 0; q < kTextRegs; ++q) {
                    if constexpr (kBytes) tcur[q] = ByteWindow::realign(tnxt[q], tshift[q]);
                    else tcur[q] = tnxt[q];
                }
                fetch_text((int)s0 + 16 - lane);
                if (pass && (s0 & 31u) == 0) {   // lane 0 is at column s0: the word for columns s0 .. s0 + 31
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
                    if constexpr (kBytes) {
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
                        if (pass) {   // uniform
                            in_ph = cw_ph << 31; in_mh = cw_mh << 31;
                            cw_ph >>= 1; cw_mh >>= 1;
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
                            if (record) {   // uniform; only lane 63's words are stored
                                ow_ph = (ow_ph >> 1) | (ph & 0x80000000u);
                                ow_mh = (ow_mh >> 1) | (mh & 0x80000000u);
                            }
                        }
                        // lane 63 has just finished column s - 63: a word is complete every 32 columns
                        if (record && s >= 63 && ((s - 63) & 31u) == 31u && s - 63 < n && lane == 63) {
                            cout_ph[(s - 63) >> 5] = ow_ph;
                            cout_mh[(s - 63) >> 5] = ow_mh;
                        }
                    }
                }
            }
            if (record && (n & 31u) && lane == 63) {   // the last, partial word: bits sit at the top
                cout_ph[n >> 5] = ow_ph >> (32 - (n & 31u));
                cout_mh[n >> 5] = ow_mh >> (32 - (n & 31u));
            }
            const uint32_t mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
            part += __popc(pv & mask) - __popc(mv & mask);
#pragma unroll
            for (int k = 0; k < BpTraits<Sym>::kEntries; ++k) table[k * 64 + lane] = 0;
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
    const int waves = bytes ? BpTraits<uint8_t>::kWaves : BpTraits<uint32_t>::kWaves;
    const size_t lds = bytes ? bp_lds_bytes<uint8_t>() : bp_lds_bytes<uint32_t>();
    uint32_t blocks = (count + waves - 1) / waves;
    const uint32_t max_blocks = (uint32_t)scope->compute_units * (bytes ? 4 : 5);
    if (blocks > max_blocks) blocks = max_blocks;
    // args.boundary / boundary_stride: the carry words, sized by the caller (bp_long_carry_words, <= 4096 waves)
    StampGuard guard(scope, bytes ? "bitparallel_long" : "bitparallel_long_u32");
    if (bytes) {
        opt_in_dynamic_lds(scope, (const void *)k_bitparallel_long<uint8_t>, lds);
        hipLaunchKernelGGL(k_bitparallel_long<uint8_t>, dim3(blocks), dim3(waves * 64), lds, scope->stream, args);
    } else {
        opt_in_dynamic_lds(scope, (const void *)k_bitparallel_long<uint32_t>, lds);
        hipLaunchKernelGGL(k_bitparallel_long<uint32_t>, dim3(blocks), dim3(waves * 64), lds, scope->stream, args);
    }
    SWH_HIP_CHECK(hipGetLastError());
}

template <typename Sym>
static void launch_bitparallel_sym(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    // The work list lives in the device plan; the host only bounds the grid (an item holds >= 1 pair).
    constexpr int kWaves = BpTraits<Sym>::kWaves;
    constexpr size_t lds = bp_lds_bytes<Sym>();
    KernelArgs k = args;
    k.boundary = nullptr;
    uint64_t blocks64 = (pairs + kWaves - 1) / kWaves;
    // bytes: 33 KB blocks of 4 waves, 4 per CU; code points: 29 KB blocks of 2 waves, 5 per CU
    uint32_t max_blocks = (uint32_t)scope->compute_units * (sizeof(Sym) == 1 ? 4 : 5);
    uint32_t blocks = blocks64 > max_blocks ? max_blocks : (uint32_t)blocks64;
    opt_in_dynamic_lds(scope, (const void *)k_bitparallel<Sym>, lds);
    StampGuard guard(scope, sizeof(Sym) == 1 ? "bitparallel" : "bitparallel_u32");
    hipLaunchKernelGGL(k_bitparallel<Sym>, dim3(blocks), dim3(kWaves * 64), lds, scope->stream, k);
    SWH_HIP_CHECK(hipGetLastError());
}

void launch_bitparallel(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    if (args.sym_bytes == 4) launch_bitparallel_sym<uint32_t>(scope, args, pairs);
    else launch_bitparallel_sym<uint8_t>(scope, args, pairs);
}

}  // namespace swh
