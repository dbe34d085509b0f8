// bitparallel.hip -- unit-cost Levenshtein as Myers/Hyyro bit-vectors on gfx950.
//
// The shorter string of a pair (the "pattern", m symbols) is cut into G = ceil(m/32) blocks of 32
// DP rows. Each block is owned by ONE lane as two 32-bit vertical-delta words (Pv, Mv); the G lanes
// of a pair form a systolic array: at global step s the lane holding block k consumes text symbol
// t = s - k and hands the horizontal delta of its last row (the top bit of Ph / Mh) to the lane
// above through one `wave_shr:1` DPP move, i.e. the blocks of a pair sit on an anti-diagonal of
// 32x1 tiles. A wave64 therefore carries floor(64/G) pairs at once (64 pairs for words of up to
// 32 symbols, one pair for a 2048-symbol line), and one step costs ~20 VALU instructions for up
// to 64 x 32 DP cells: `v_bitop3_b32` folds Hyyro's boolean recurrences three inputs at a time.
//
// The per-symbol match vectors ("Peq") live in LDS as a lane-interleaved table peq[symbol][lane]
// (bank = lane % 32, conflict-free for any symbol mix); it is built with `ds_or_b32`, read once
// per step with `ds_read_b32` and un-built afterwards by re-walking the pattern. The table is
// either 128 symbols x 64 lanes x 4 B = 32 KB per wave (7-bit fast path, four waves per CU) or
// 256 symbols = 64 KB per wave (any byte, two waves per CU). The 7-bit kernel runs first; a wave
// that meets a byte >= 0x80 appends its chunk to an overflow list that the 8-bit kernel drains.
//
// Distance = n + popcount(Pv) - popcount(Mv) summed over the pair's blocks after the last text
// symbol (D[m][n] = D[0][n] + sum of vertical deltas of the last column), so no per-step score
// bookkeeping is needed. Definition matched: `levenshtein::distance` (bench.rs:416-419).
#include "common.hpp"

namespace swh {

__device__ __forceinline__ int bp_med3i(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Four consecutive tape bytes starting at signed index `idx` relative to `base`, never touching
// memory outside [base+lo, base+hi+4). Bytes that fall outside come back as garbage.
__device__ __forceinline__ uint32_t bp_fetch4(const uint8_t *base, int idx, int lo, int hi, int avail, bool tiny) {
    if (!tiny) {
        int c = bp_med3i(idx, lo, hi);
        uint32_t dw;
        __builtin_memcpy(&dw, base + c, 4);
        int d = bp_med3i(idx - c, -3, 3);
        return d >= 0 ? dw >> (8 * d) : dw << (-8 * d);
    }
    uint32_t dw = 0;
    for (int u = 0; u < 4; ++u) {
        int pos = idx + u;
        if (pos >= lo && pos < avail) dw |= (uint32_t)base[pos] << (8 * u);
    }
    return dw;
}

struct BpOverflow {
    uint32_t *count;   // number of deferred work items
    uint32_t *items;   // deferred work item ids
    uint32_t *ticket;  // work-stealing counter of this launch
};

// NSYM = 128: 7-bit table, defers chunks with bytes >= 0x80 to `overflow`.
// NSYM = 256: drains `overflow` (FROM_LIST) or runs everything (no list).
template <int NSYM, bool FROM_LIST>
__global__ __launch_bounds__(NSYM == 128 ? 256 : 128) void k_bitparallel(KernelArgs args, BpOverflow ovf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kWaves = NSYM == 128 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t *peq = (uint32_t *)smem + (size_t)wave_in_block * NSYM * 64;  // [NSYM][64]
    uint32_t *acc = (uint32_t *)smem + (size_t)kWaves * NSYM * 64 + wave_in_block * 64;
    // exclusive prefix of work items per bit-parallel class (all LDS in the one dynamic region, guide G17)
    uint32_t *item_prefix = (uint32_t *)smem + (size_t)kWaves * NSYM * 64 + kWaves * 64;  // [65]

    for (int i = lane; i < NSYM * 64; i += 64) peq[i] = 0;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int g = 1; g <= 64; ++g) {
            item_prefix[g - 1] = run;
            uint32_t per = 64 / g, cnt = args.plan->class_count[kClassBp0 + g - 1];
            run += (cnt + per - 1) / per;
        }
        item_prefix[64] = run;
    }
    __syncthreads();
    const uint32_t items_total = FROM_LIST ? *ovf.count : item_prefix[64];
    const uint64_t a_total = args.off64 ? ((const uint64_t *)args.job.a.offsets)[args.job.a.count]
                                        : ((const uint32_t *)args.job.a.offsets)[args.job.a.count];
    const uint64_t b_total = args.off64 ? ((const uint64_t *)args.job.b.offsets)[args.job.b.count]
                                        : ((const uint32_t *)args.job.b.offsets)[args.job.b.count];

    for (;;) {
        // grab the next work item (wave-uniform)
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(ovf.ticket, 1u);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= items_total) break;
        uint32_t item = FROM_LIST ? ovf.items[w] : items_total - 1 - w;  // heavy classes first
        // class lookup: largest g with item_prefix[g-1] <= item
        int g = 1;
        for (int k = 1; k <= 64; ++k)
            if (item_prefix[k - 1] <= item) g = k;
        const uint32_t G = (uint32_t)g;
        const uint32_t per = 64 / G;  // pairs per wave
        const uint32_t chunk = item - item_prefix[g - 1];
        const uint32_t cls = kClassBp0 + g - 1;
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
        // pattern = shorter string (rows / bits), text = longer string (columns / steps)
        const bool a_is_pattern = la <= lb;
        const uint32_t m = a_is_pattern ? la : lb, n = a_is_pattern ? lb : la;
        const uint8_t *pat = (const uint8_t *)(a_is_pattern ? args.job.a.data : args.job.b.data) +
                             (a_is_pattern ? a0 : b0);
        const uint8_t *txt = (const uint8_t *)(a_is_pattern ? args.job.b.data : args.job.a.data) +
                             (a_is_pattern ? b0 : a0);
        const uint64_t pat_start = a_is_pattern ? a0 : b0, txt_start = a_is_pattern ? b0 : a0;
        const uint64_t pat_total = a_is_pattern ? a_total : b_total, txt_total = a_is_pattern ? b_total : a_total;
        auto clamp31 = [](int64_t v) { return (int)(v < -0x40000000ll ? -0x40000000ll : (v > 0x40000000ll ? 0x40000000ll : v)); };
        const int pat_lo = clamp31(-(int64_t)pat_start), pat_hi = clamp31((int64_t)pat_total - (int64_t)pat_start - 4);
        const int pat_av = clamp31((int64_t)pat_total - (int64_t)pat_start);
        const int txt_lo = clamp31(-(int64_t)txt_start), txt_hi = clamp31((int64_t)txt_total - (int64_t)txt_start - 4);
        const int txt_av = clamp31((int64_t)txt_total - (int64_t)txt_start);
        const bool pat_tiny = pat_total < 4, txt_tiny = txt_total < 4;

        // rows of my block
        const uint32_t row0 = blk * 32;
        const uint32_t brows = have ? (m > row0 ? (m - row0 < 32 ? m - row0 : 32) : 0) : 0;

        // ---- build Peq for my block ----------------------------------------------------------
        uint32_t pw[8];
        uint32_t high = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            pw[q] = brows > (uint32_t)q * 4 ? bp_fetch4(pat, (int)row0 + q * 4, pat_lo, pat_hi, pat_av, pat_tiny) : 0;
            high |= pw[q] & 0x80808080u;  // over-approximate: stray bytes only cost a deferral, never a wrong answer
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if ((uint32_t)q < brows) {
                uint32_t c = (pw[q >> 2] >> (8 * (q & 3))) & 0xffu;
                if (NSYM == 256 || c < 128) atomicOr(&peq[c * 64 + lane], 1u << q);
            }
        }
        acc[lane] = 0;

        // wave-uniform step count
        uint32_t n_eff = have ? n + G - 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            uint32_t other = __shfl_xor(n_eff, off);
            n_eff = other > n_eff ? other : n_eff;
        }
        const uint32_t steps = (n_eff + 3) & ~3u;

        // lanes that start a pair take the DP boundary (+1 horizontal delta) instead of a neighbour
        const bool first_blk = blk == 0;
        uint32_t pv = 0xFFFFFFFFu, mv = 0, ph = 0, mh = 0;
        uint32_t tcur, tnxt = bp_fetch4(txt, 0 - (int)blk, txt_lo, txt_hi, txt_av, txt_tiny);
        for (uint32_t s0 = 0; s0 < steps; s0 += 4) {
            tcur = tnxt;
            tnxt = bp_fetch4(txt, (int)s0 + 4 - (int)blk, txt_lo, txt_hi, txt_av, txt_tiny);
            high |= tcur & 0x80808080u;
            uint32_t eqs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t c = (tcur >> (8 * u)) & (NSYM - 1);
                eqs[u] = peq[c * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t s = s0 + u;
                uint32_t ph_in = (uint32_t)__builtin_amdgcn_update_dpp((int)0x80000000u, (int)ph, 0x138, 0xf, 0xf, false);
                uint32_t mh_in = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mh, 0x138, 0xf, 0xf, false);
                ph_in = first_blk ? 0x80000000u : ph_in;
                mh_in = first_blk ? 0u : mh_in;
                if (s - blk < n) {
                    uint32_t eq = eqs[u];
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
            }
        }

        // ---- distance = n + sum over blocks popcount(pv) - popcount(mv) -------------------------
        const uint32_t mask = brows >= 32 ? 0xFFFFFFFFu : ((1u << brows) - 1u);
        int part = __popc(pv & mask) - __popc(mv & mask);
        if (have && brows) atomicAdd(&acc[slot * G], (uint32_t)part);
        uint32_t any_high = 0;
        if (NSYM == 128) any_high = __any(have && high) ? 1u : 0u;
        if (NSYM == 128 && any_high) {
            if (lane == 0) {
                uint32_t at = atomicAdd(ovf.count, 1u);
                ovf.items[at] = item;
            }
        } else if (have && first_blk) {
            uint32_t d = n + acc[lane];
            store_result(args.job, p, (int64_t)clamp_bound(d, args.job.bound));
        }

        // ---- un-build Peq ---------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if ((uint32_t)q < brows) {
                uint32_t c = (pw[q >> 2] >> (8 * (q & 3))) & 0xffu;
                if (NSYM == 256 || c < 128) peq[c * 64 + lane] = 0;
            }
        }
    }
}

void launch_bitparallel(Scope *scope, const KernelArgs &args, const Plan &plan) {
    uint64_t items = 0;
    for (int g = 1; g <= 64; ++g) {
        uint32_t per = 64 / g, cnt = plan.class_count[kClassBp0 + g - 1];
        items += (cnt + per - 1) / per;
    }
    if (!items) return;
    // overflow bookkeeping lives at the tail of the boundary scratch handed in by the caller
    uint32_t *ctl = (uint32_t *)args.boundary;  // [0]=ticket7 [1]=ovf count [2]=ticket8, then items
    BpOverflow ovf7{ctl + 1, ctl + 4, ctl + 0};
    BpOverflow ovf8{ctl + 1, ctl + 4, ctl + 2};
    SWH_HIP_CHECK(hipMemsetAsync(ctl, 0, 16, scope->stream));
    KernelArgs k = args;
    k.boundary = nullptr;
    {
        size_t lds = (size_t)4 * 128 * 64 * 4 + 4 * 64 * 4 + 80 * 4;
        uint32_t blocks = (uint32_t)((items + 3) / 4);
        uint32_t max_blocks = (uint32_t)scope->compute_units;  // one 128 KB block per CU
        if (blocks > max_blocks) blocks = max_blocks;
        static bool attr_set = false;
        if (!attr_set) {
            SWH_HIP_CHECK(hipFuncSetAttribute((const void *)k_bitparallel<128, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        StampGuard guard(scope, "bitparallel_7bit");
        hipLaunchKernelGGL((k_bitparallel<128, false>), dim3(blocks), dim3(256), lds, scope->stream, k, ovf7);
    }
    {
        size_t lds = (size_t)2 * 256 * 64 * 4 + 2 * 64 * 4 + 80 * 4;
        uint32_t blocks = (uint32_t)scope->compute_units;
        if (blocks > (items + 1) / 2) blocks = (uint32_t)((items + 1) / 2);
        static bool attr_set = false;
        if (!attr_set) {
            SWH_HIP_CHECK(hipFuncSetAttribute((const void *)k_bitparallel<256, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        StampGuard guard(scope, "bitparallel_8bit");
        hipLaunchKernelGGL((k_bitparallel<256, true>), dim3(blocks), dim3(128), lds, scope->stream, k, ovf8);
    }
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
