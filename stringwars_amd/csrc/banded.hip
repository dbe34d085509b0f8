// banded.hip -- bounded unit-cost Levenshtein (out = min(d, k+1), k <= 127) as a sliding band of one or two 64-bit words.
//
// Ukkonen: an alignment with at most k edits between strings whose lengths differ by delta stays on
// the diagonals [-(k+delta)/2, (k-delta)/2] -- at most k+1 of them. The band is kept as ONE 64-bit
// window per pair that slides down one row per text symbol (Hyyro's diagonal tiling; the scheme of
// rapidfuzz's `levenshtein_hyrroe2003_small_band`, re-anchored on the tight band so k = 32 fits):
//     D0 = (((Eq & VP) + VP) ^ VP) | Eq | VN        HP = VN | ~(D0 | VP)        HN = D0 & VP
//     VP' = HN | ~((D0 >> 1) | HP)                  VN' = (D0 >> 1) & HP
// with bit 63 on the band's bottom diagonal; the distance is tracked along that diagonal while it is
// inside the matrix and along the last row afterwards. Rows above the matrix behave like row 0 on their
// own (VP = VN = Eq = 0 gives HP = 1), rows entering at the bottom start as "+1", which over-estimates
// cells outside the band only -- exact whenever d <= k. Prototype + proof-by-test: DESIGN.md section 4.4.
// Bounds of 64 .. 127 (STRINGWARS_ERROR_BOUND is free-form, README.md:311) run the same scheme on a window of NW = 2 words: bit
// 64 NW - 1 on the bottom diagonal, the addition's carry and the one-bit shift run through the words. (The code is written for
// any NW; three and four words were built and measured on config C3's lines -- k = 128: 1.30 ms against 1.05 for the unbounded
// code-point kernel + the clamp, whose 16 blocks of 32 rows ARE half this matrix's band by then -- and are not instantiated.)
//
// Mapping to a wave64 (no tables, any symbol width -- bytes or decoded code points):
//   phase 1  "match masks": for a chunk of 32 text symbols, lanes = (pair parity, column). Each lane
//            loads the k+1 pattern symbols of its column's window with batched unaligned dword loads,
//            compares them against its text symbol and deposits a 64-bit Eq mask in LDS.
//   phase 2  "recurrence": lanes = pairs (64 per wave); each lane walks the 32 columns of the chunk
//            reading its Eq masks back from LDS (row pitch 33 x 8 B: conflict-free both ways).
// Cost per pair-column is ~(3 (k+1) + 45) / 64 wave instructions instead of 28 * ceil(m/32) / (64/G) for
// the full bit-parallel kernel: 6-10x fewer on ~1 KB lines with k = 32 (config C3).
#include "common.hpp"

namespace swh {

constexpr int kBandChunk = 32;                      // text symbols per chunk
constexpr int kBandPitch = kBandChunk + 1;          // u64 per pair row in LDS
constexpr int kBandWaves = 4;
constexpr int kBandParamWords = 8;
constexpr int band_stage(int wbits) { return wbits <= 64 ? 96 : 32 + wbits; }   // u32 window symbols staged per half-wave
// P = pairs per wave item (64, or 32 when the batch is too small to give every SIMD a few waves at 64; 16 for windows of three and four words)
constexpr size_t band_lds_per_wave(int P, int wbits = 64) {
    return (size_t)((wbits + 63) / 64) * P * kBandPitch * 8 + (size_t)P * kBandParamWords * 4 + 2 * (size_t)band_stage(wbits) * 4;
}

struct BandPair {            // per-pair parameters parked in LDS for phase 1 (uniform reads)
    uint32_t pat_lo, pat_hi; // pattern pointer
    uint32_t txt_lo, txt_hi; // text pointer
    uint32_t len1, len2;     // pattern / text length in symbols
    int32_t start0;          // pattern index of window bit 0 at text index 0 (= dhi - (64 NW - 1))
    uint32_t pad;
};
static_assert(sizeof(BandPair) == kBandParamWords * 4, "BandPair layout");

template <typename Sym>
__device__ __forceinline__ uint32_t band_load_sym(const Sym *base, int idx, int lo, int hi) {
    int c = idx < lo ? lo : (idx > hi ? hi : idx);
    return (uint32_t)base[c];
}

// WBITS = number of live window bits (a multiple of 4 up to 64, of 32 beyond; >= k + 1): bits [64 NW - WBITS, 64 NW - 1], NW = ceil(WBITS / 64).
// P = the most pairs an item can hold (LDS layout). The recurrence is a serial chain per pair and one wave issues a dependent
// instruction only every ~8.5 cycles, so a batch of 100 K pairs (1564 items of 64 = 1.5 waves per SIMD) ran at 11 SIMD-cycles
// per instruction. Items of 32 pairs leave half of phase 2's lanes idle (+30 % instructions) but double the resident waves.
// How many pairs an item really gets is decided here, from the class's size: every wave of the launch the same number of
// items (one, while the class fits) of the same size -- 100 K pairs on 4096 waves are items of 26, four waves on every SIMD,
// where items of 32 left some SIMDs with four waves and some with three.
template <typename Sym, int WBITS, int P>
__global__ __launch_bounds__(256) void k_banded(KernelArgs args, uint32_t cls) {
    constexpr int NW = (WBITS + 63) / 64, kTop = 64 * NW - 1, kBandStage = band_stage(WBITS);
    constexpr size_t kBandLdsPerWave = band_lds_per_wave(P, WBITS);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    char *wave_lds = smem + (size_t)wave_in_block * kBandLdsPerWave;
    unsigned long long *eqbuf = (unsigned long long *)wave_lds;                   // [NW][P][kBandPitch]
    BandPair *params = (BandPair *)(wave_lds + (size_t)NW * P * kBandPitch * 8);       // [P]
    uint32_t *stage = (uint32_t *)(wave_lds + (size_t)NW * P * kBandPitch * 8 + (size_t)P * kBandParamWords * 4);  // [2][kBandStage]

    const uint32_t cstart = args.plan->class_start[cls], ccount = args.plan->class_count[cls];
    const uint32_t waves_total = gridDim.x * kBandWaves;
    uint32_t per_item = P;
    if (!args.band_fixed_items) {
        const uint32_t rounds = (ccount + waves_total * P - 1) / (waves_total * P);
        per_item = rounds ? (ccount + waves_total * rounds - 1) / (waves_total * rounds) : 2;
        per_item = (per_item + 1) & ~1u;
        per_item = per_item < 2 ? 2 : (per_item > (uint32_t)P ? (uint32_t)P : per_item);
    }
    const uint32_t chunks = (ccount + per_item - 1) / per_item;
    const uint32_t wave_id = blockIdx.x * kBandWaves + wave_in_block;
    const uint32_t k = args.job.bound;

    for (uint32_t item_rev = wave_id; item_rev < chunks; item_rev += waves_total) {
        const uint32_t item = chunks - 1 - item_rev;  // longest texts first
        const uint32_t pidx = item * per_item + lane;
        const bool have = (uint32_t)lane < per_item && pidx < ccount;
        uint64_t p = 0, a0 = 0, b0 = 0;
        uint32_t la = 0, lb = 0;
        if (have) {
            p = args.perm[cstart + pidx];
            if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
            else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
        }
        // text = shorter string (columns), pattern = longer string (window rows)
        const bool a_is_text = la <= lb;
        const uint32_t len2 = a_is_text ? la : lb, len1 = a_is_text ? lb : la;
        const Sym *txt = (const Sym *)(a_is_text ? args.job.a.data : args.job.b.data) + (a_is_text ? a0 : b0);
        const Sym *pat = (const Sym *)(a_is_text ? args.job.b.data : args.job.a.data) + (a_is_text ? b0 : a0);
        const int delta = (int)len2 - (int)len1;              // <= 0, |delta| <= k (pre-pass guarantees it)
        const int dhi = ((int)k - delta) / 2;                 // bottom diagonal of the band, 0 <= dhi <= k <= 64 NW - 1
        const int start0 = dhi - kTop;
        {
            BandPair bp;
            bp.pat_lo = (uint32_t)(uintptr_t)pat; bp.pat_hi = (uint32_t)((uintptr_t)pat >> 32);
            bp.txt_lo = (uint32_t)(uintptr_t)txt; bp.txt_hi = (uint32_t)((uintptr_t)txt >> 32);
            bp.len1 = len1; bp.len2 = have ? len2 : 0; bp.start0 = start0; bp.pad = 0;
            if (lane < P) params[lane] = bp;
        }
        wave_lds_fence();  // params are read by other lanes in phase 1
        const uint32_t n_max = wave_max_u32(have ? len2 : 0);
        // recurrence state of my pair (phase 2): rows 0..dhi of column 0 carry vertical +1
        // (32-bit words: the carry chain, the funnel shift and the bit operations are what the machine has; the compiler's 64-bit
        // forms of the same cost the one-word kernel 5 % when the window was first generalised)
        constexpr int NV = 2 * NW;
        uint32_t vp[NV], vn[NV];
        {
            // bits b with start0 + b >= 0  <=>  b >= kTop - dhi
            const int first = kTop - dhi;
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                const int from = first - 32 * w;     // first set bit inside word w
                vp[w] = from <= 0 ? 0xFFFFFFFFu : (from >= 32 ? 0u : (0xFFFFFFFFu << from));
                vn[w] = 0;
            }
        }
        int cur = (int)len1 < dhi ? (int)len1 : dhi;
        const int diag_cols = (int)len1 - dhi;  // text indices i < diag_cols follow the bottom diagonal

        for (uint32_t i0 = 0; i0 < n_max; i0 += kBandChunk) {
            // A wave that is ahead steps back (fair_priority, common.hpp): priority 3 in the first quarter of its item's columns, 0 in the last. Every wave has ONE
            // item here, all start together, and the arbiter's tie-break is "oldest first": the oldest waves of a SIMD ran ahead and
            // left the youngest to finish alone, one dependent instruction every ~8 cycles on a SIMD that could issue four times as
            // often. With the priority falling along the item the waves behind catch up and the SIMD stays full to the end:
            // C3 0.319 -> 0.306 ms (80.7 -> 83.8 TCUPS).
            fair_priority(i0, n_max);
            // ---------------- phase 1: Eq masks, two pairs per iteration ----------------------------
            // The 32 lanes of a half-wave serve one pair: they stage the 32 + WBITS pattern symbols their
            // windows cover into LDS once (rows outside the pattern become a sentinel no symbol equals), then
            // every lane compares its text symbol against its own WBITS-symbol window, one bit per compare.
            const int half = lane >> 5, col = lane & 31;
            uint32_t *win = stage + half * kBandStage;
            constexpr int kStageLoads = (32 + WBITS + 31) / 32;
            // Global loads of pair q+2 are issued before pair q is compared, so their latency overlaps the
            // compare loop instead of heading every iteration.
            uint32_t nxt_sym[kStageLoads], nxt_tsym;
            auto issue_loads = [&](int q) {
                const BandPair bp = params[q];
                const uint32_t i = i0 + (uint32_t)col;
                const Sym *tp = (const Sym *)(((uintptr_t)bp.txt_hi << 32) | bp.txt_lo);
                const Sym *pw = (const Sym *)(((uintptr_t)bp.pat_hi << 32) | bp.pat_lo);
                nxt_tsym = i < bp.len2 ? (uint32_t)tp[i] : 0xFFFFFFFEu;   // past the text: matches nothing
                const int base0 = bp.start0 + (int)i0 + (64 * NW - WBITS);   // pattern index of window symbol 0, column 0
#pragma unroll
                for (int r = 0; r < kStageLoads; ++r) {
                    const int idx = base0 + col + 32 * r;
                    nxt_sym[r] = (idx >= 0 && idx < (int)bp.len1) ? (uint32_t)pw[idx] : 0xFFFFFFFFu;  // outside: sentinel
                }
            };
            issue_loads(half);
#pragma unroll 1
            for (int pp = 0; pp < (int)per_item / 2; ++pp) {
                const int q = pp * 2 + half;               // pair slot served by my half of the wave
                const uint32_t tsym = nxt_tsym;
#pragma unroll
                for (int r = 0; r < kStageLoads; ++r)
                    if (col + 32 * r < 32 + WBITS) win[col + 32 * r] = nxt_sym[r];
                if (pp + 1 < (int)per_item / 2) issue_loads(q + 2);
                wave_lds_fence();  // every lane reads symbols its neighbours staged
                // hits: window symbol j <-> band bit 64 - WBITS + j. Built as kSegs independent accumulators
                // (seg = seg * 2 + (symbol == text symbol), one compare + one add-with-carry per symbol, most
                // significant first) so the carry chains overlap instead of forming one WBITS-deep dependency.
                constexpr int kSegs = WBITS >= 64 ? 8 : 4, kSegLen = WBITS / kSegs;
                static_assert(WBITS % kSegs == 0, "window bits must split evenly");
                uint32_t seg[kSegs];
#pragma unroll
                for (int sgm = 0; sgm < kSegs; ++sgm) seg[sgm] = 0;
#pragma unroll
                for (int t = kSegLen - 1; t >= 0; --t) {
#pragma unroll
                    for (int g4 = 0; g4 < kSegs; g4 += 4) {
                        const uint32_t x0 = win[col + (g4 + 0) * kSegLen + t], x1 = win[col + (g4 + 1) * kSegLen + t];
                        const uint32_t x2 = win[col + (g4 + 2) * kSegLen + t], x3 = win[col + (g4 + 3) * kSegLen + t];
                        unsigned long long m0, m1, m2, m3;
                        // Four compares into four SGPR pairs, then four add-with-carry: every carry is read three
                        // instructions after it was written (gfx950 wants >= 2 wait states between a VALU write of
                        // an SGPR and a VALU read of it; hipcc pads that itself but not inside an asm statement).
                        asm("v_cmp_eq_u32_e64 %4, %8, %12\n\t"
                            "v_cmp_eq_u32_e64 %5, %9, %12\n\t"
                            "v_cmp_eq_u32_e64 %6, %10, %12\n\t"
                            "v_cmp_eq_u32_e64 %7, %11, %12\n\t"
                            "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"
                            "v_addc_co_u32_e64 %1, %5, %1, %1, %5\n\t"
                            "v_addc_co_u32_e64 %2, %6, %2, %2, %6\n\t"
                            "v_addc_co_u32_e64 %3, %7, %3, %3, %7"
                            : "+v"(seg[g4 + 0]), "+v"(seg[g4 + 1]), "+v"(seg[g4 + 2]), "+v"(seg[g4 + 3]), "=&s"(m0), "=&s"(m1),
                              "=&s"(m2), "=&s"(m3)
                            : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(tsym));
                    }
                }
                // segment sgm holds window symbols [sgm kSegLen, (sgm + 1) kSegLen) = band bits from 64 NW - WBITS + sgm kSegLen on
                unsigned long long hits[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) hits[w] = 0;
#pragma unroll
                for (int sgm = 0; sgm < kSegs; ++sgm) {
                    constexpr int kOff = 64 * NW - WBITS;
                    const int pos = kOff + sgm * kSegLen, word = pos >> 6, shift = pos & 63;
                    hits[word] |= (unsigned long long)seg[sgm] << shift;
                    if (shift + kSegLen > 64 && word + 1 < NW) hits[word + 1] |= (unsigned long long)seg[sgm] >> (64 - shift);
                }
#pragma unroll
                for (int w = 0; w < NW; ++w) eqbuf[((size_t)w * P + q) * kBandPitch + col] = hits[w];
                wave_lds_fence();  // the window buffer is rewritten by the next pair
            }
            wave_lds_fence();  // Eq masks cross from (pair parity, column) lanes to pair lanes
            // ---------------- phase 2: recurrence, lane = pair -------------------------------------------
#pragma unroll 4
            for (int c = 0; c < kBandChunk; ++c) {
                const uint32_t i = i0 + (uint32_t)c;
                if (have && i < len2) {
                    uint32_t eq[NV], d0[NV], hp[NV], hn[NV];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const unsigned long long e64 = eqbuf[((size_t)w * P + (lane < P ? lane : 0)) * kBandPitch + c];
                        eq[2 * w] = (uint32_t)e64; eq[2 * w + 1] = (uint32_t)(e64 >> 32);
                    }
                    // D0 = (((Eq & VP) + VP) ^ VP) | Eq | VN   (one addition through all the words)
                    uint32_t carry = 0;
#pragma unroll
                    for (int w = 0; w < NV; ++w) {
                        const uint32_t x = eq[w] & vp[w];
                        const unsigned long long sum = (unsigned long long)x + vp[w] + carry;
                        carry = (uint32_t)(sum >> 32);
                        d0[w] = ((uint32_t)sum ^ vp[w]) | eq[w] | vn[w];
                        hp[w] = vn[w] | ~(d0[w] | vp[w]);
                        hn[w] = d0[w] & vp[w];
                    }
                    if ((int)i < diag_cols) {
                        cur += (d0[NV - 1] >> 31) ? 0 : 1;                // one step down the bottom diagonal
                    } else {
                        const int b = (int)len1 - 1 - (start0 + (int)i);  // last pattern row inside the window
                        uint32_t hp_w = hp[0], hn_w = hn[0];
#pragma unroll
                        for (int w = 1; w < NV; ++w)
                            if ((b >> 5) == w) { hp_w = hp[w]; hn_w = hn[w]; }
                        cur += (int)((hp_w >> (b & 31)) & 1u) - (int)((hn_w >> (b & 31)) & 1u);
                    }
#pragma unroll
                    for (int w = 0; w < NV; ++w) {
                        const uint32_t d1 = w + 1 < NV ? __builtin_amdgcn_alignbit(d0[w + 1 < NV ? w + 1 : w], d0[w], 1) : d0[w] >> 1;   // D0 >> 1
                        vp[w] = hn[w] | ~(d1 | hp[w]);
                        vn[w] = d1 & hp[w];
                    }
                }
            }
            wave_lds_fence();  // the next chunk's phase 1 overwrites the Eq buffer
        }
        if (have) {
            uint32_t d = cur < 0 ? 0u : (uint32_t)cur;
            store_result(args.job, p, (int64_t)(d > k ? k + 1 : d));
        }
        wave_lds_fence();  // the next item rewrites params
    }
}

template <typename Sym, int WBITS, int P>
static void launch_banded_items(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    const size_t lds = kBandWaves * band_lds_per_wave(P, WBITS);
    // the kernel sizes its items from the class's real size (known on the device only); the grid is the whole device
    // unless the batch is too small to give every wave two pairs
    uint64_t items = args.band_fixed_items ? (pairs + P - 1) / P : (pairs + 1) / 2;
    uint64_t blocks64 = (items + kBandWaves - 1) / kBandWaves;
    uint32_t max_blocks = (uint32_t)scope->compute_units * (uint32_t)(160 * 1024 / lds);   // 77 KB (P = 64) / 41 KB per block
    uint32_t blocks = blocks64 > max_blocks ? max_blocks : (uint32_t)blocks64;
    if (!blocks) return;
    opt_in_dynamic_lds(scope, (const void *)k_banded<Sym, WBITS, P>, lds);
    StampGuard guard(scope, "banded");
    hipLaunchKernelGGL((k_banded<Sym, WBITS, P>), dim3(blocks), dim3(256), lds, scope->stream, args, (uint32_t)kClassBanded);
}

template <typename Sym, int WBITS>
static void launch_banded_one(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    // Items of at most 32 pairs: 40 KB of LDS per workgroup, four workgroups = sixteen waves per CU. Measured on C3's 100 K
    // pairs with the items sized by the kernel (DESIGN_HISTORY.md, round-5 text 4.3): capacity 20 / 24 / 32 / 40 / 64 (6 / 5 / 4 / 3 / 2 workgroups
    // per CU): 0.356 / 0.344 / 0.331 / 0.351 / 0.450 ms; at 150 K - 800 K pairs 32 beats 64 by 13-17 % as well.
    // STRINGWARS_AMD_BAND_CAP=64: the large items (comparison knob).
    static const int forced = [] { const char *e = test_hook("STRINGWARS_AMD_BAND_CAP"); return e ? atoi(e) : 0; }();
    // Two-word windows: items of 16 pairs keep the workgroup at 35 KB of LDS = four per CU (items of 32: 69 KB, two per CU, and
    // the serial recurrence at two waves per SIMD: C3's lines at k = 64 / 100 / 127 take 0.81 / 0.99 / 1.05 ms instead of 0.73 / 0.84 / 0.91).
    if constexpr (WBITS > 64) launch_banded_items<Sym, WBITS, 16>(scope, args, pairs);
    else if (forced == 64) launch_banded_items<Sym, WBITS, 64>(scope, args, pairs);
    else launch_banded_items<Sym, WBITS, 32>(scope, args, pairs);
}

void launch_banded(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    const uint32_t k = args.job.bound;
    KernelArgs a = args;
    a.boundary = nullptr;
    static const bool fixed_items = [] { const char *e = test_hook("STRINGWARS_AMD_BAND_ITEMS"); return e && e[0] == 'f'; }();   // "fixed": comparison knob
    a.band_fixed_items = fixed_items ? 1u : 0u;
#define SWH_BAND(SYM)                                                     \
    if (k + 1 <= 8) launch_banded_one<SYM, 8>(scope, a, pairs);           \
    else if (k + 1 <= 16) launch_banded_one<SYM, 16>(scope, a, pairs);    \
    else if (k + 1 <= 36) launch_banded_one<SYM, 36>(scope, a, pairs);    \
    else if (k + 1 <= 64) launch_banded_one<SYM, 64>(scope, a, pairs);    \
    else if (k + 1 <= 96) launch_banded_one<SYM, 96>(scope, a, pairs);    \
    else launch_banded_one<SYM, 128>(scope, a, pairs);
    if (args.sym_bytes == 4) { SWH_BAND(uint32_t) } else { SWH_BAND(uint8_t) }
#undef SWH_BAND
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
