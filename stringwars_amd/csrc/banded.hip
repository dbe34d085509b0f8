// banded.hip -- bounded unit-cost Levenshtein (out = min(d, k+1), k <= 63) as a sliding 64-bit band.
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
constexpr int kBandParamWords = 12;
constexpr size_t kBandLdsPerWave = (size_t)64 * kBandPitch * 8 + 64 * kBandParamWords * 4;

struct BandPair {            // per-pair parameters parked in LDS for phase 1 (uniform reads)
    uint32_t pat_lo, pat_hi; // pattern pointer
    uint32_t txt_lo, txt_hi; // text pointer
    uint32_t len1, len2;     // pattern / text length in symbols
    int32_t start0;          // pattern index of window bit 0 at text index 0 (= dhi - 63)
    int32_t pat_min, pat_max;  // clamp range of symbol-read start indices relative to the pattern pointer
    int32_t txt_avail;       // readable text symbols from the text pointer
    uint32_t pad0, pad1;
};
static_assert(sizeof(BandPair) == kBandParamWords * 4, "BandPair layout");

template <typename Sym>
__device__ __forceinline__ uint32_t band_load_sym(const Sym *base, int idx, int lo, int hi) {
    int c = idx < lo ? lo : (idx > hi ? hi : idx);
    return (uint32_t)base[c];
}

// WBITS = number of live window bits (a multiple of 4, >= k + 1): bits [64 - WBITS, 63].
template <typename Sym, int WBITS>
__global__ __launch_bounds__(256) void k_banded(KernelArgs args, uint32_t cls) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    char *wave_lds = smem + (size_t)wave_in_block * kBandLdsPerWave;
    unsigned long long *eqbuf = (unsigned long long *)wave_lds;                   // [64][kBandPitch]
    BandPair *params = (BandPair *)(wave_lds + (size_t)64 * kBandPitch * 8);      // [64]

    const uint32_t cstart = args.plan->class_start[cls], ccount = args.plan->class_count[cls];
    const uint32_t chunks = (ccount + 63) / 64;
    const uint32_t waves_total = gridDim.x * kBandWaves;
    const uint32_t wave_id = blockIdx.x * kBandWaves + wave_in_block;
    const uint32_t k = args.job.bound;
    const uint64_t a_total = args.off64 ? ((const uint64_t *)args.job.a.offsets)[args.job.a.count]
                                        : ((const uint32_t *)args.job.a.offsets)[args.job.a.count];
    const uint64_t b_total = args.off64 ? ((const uint64_t *)args.job.b.offsets)[args.job.b.count]
                                        : ((const uint32_t *)args.job.b.offsets)[args.job.b.count];

    for (uint32_t item_rev = wave_id; item_rev < chunks; item_rev += waves_total) {
        const uint32_t item = chunks - 1 - item_rev;  // longest texts first
        const uint32_t pidx = item * 64 + lane;
        const bool have = pidx < ccount;
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
        const uint64_t pat_start = a_is_text ? b0 : a0, pat_total = a_is_text ? b_total : a_total;
        const uint64_t txt_start = a_is_text ? a0 : b0, txt_total = a_is_text ? a_total : b_total;
        const int delta = (int)len2 - (int)len1;              // <= 0, |delta| <= k (pre-pass guarantees it)
        const int dhi = ((int)k - delta) / 2;                 // bottom diagonal of the band, 0 <= dhi <= 63
        const int start0 = dhi - 63;
        {
            BandPair bp;
            bp.pat_lo = (uint32_t)(uintptr_t)pat; bp.pat_hi = (uint32_t)((uintptr_t)pat >> 32);
            bp.txt_lo = (uint32_t)(uintptr_t)txt; bp.txt_hi = (uint32_t)((uintptr_t)txt >> 32);
            bp.len1 = len1; bp.len2 = have ? len2 : 0; bp.start0 = start0;
            int64_t lo = -(int64_t)pat_start, hi = (int64_t)pat_total - (int64_t)pat_start - 1;
            bp.pat_min = (int)(lo < -0x40000000ll ? -0x40000000ll : lo);
            bp.pat_max = (int)(hi > 0x40000000ll ? 0x40000000ll : hi);
            int64_t av = (int64_t)txt_total - (int64_t)txt_start;
            bp.txt_avail = (int)(av > 0x40000000ll ? 0x40000000ll : av);
            bp.pad0 = bp.pad1 = 0;
            params[lane] = bp;
        }
        uint32_t n_max = have ? len2 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            uint32_t other = __shfl_xor(n_max, off);
            n_max = other > n_max ? other : n_max;
        }
        // recurrence state of my pair (phase 2): rows 0..dhi of column 0 carry vertical +1
        uint32_t vp_lo, vp_hi, vn_lo = 0, vn_hi = 0;
        {
            // bits b with start0 + b >= 0  <=>  b >= 63 - dhi
            int first = 63 - dhi;
            unsigned long long vp = first <= 0 ? ~0ull : (~0ull << first);
            vp_lo = (uint32_t)vp; vp_hi = (uint32_t)(vp >> 32);
        }
        int cur = (int)len1 < dhi ? (int)len1 : dhi;
        const int diag_cols = (int)len1 - dhi;  // text indices i < diag_cols follow the bottom diagonal

        for (uint32_t i0 = 0; i0 < n_max; i0 += kBandChunk) {
            // ---------------- phase 1: Eq masks, two pairs per iteration ----------------------------
            const int half = lane >> 5, col = lane & 31;
#pragma unroll 1
            for (int pp = 0; pp < 32; ++pp) {
                const int q = pp * 2 + half;               // pair slot served by my half of the wave
                const BandPair bp = params[q];
                const uint32_t i = i0 + (uint32_t)col;
                unsigned long long eq = 0;
                if (i < bp.len2) {
                    const Sym *tp = (const Sym *)(((uintptr_t)bp.txt_hi << 32) | bp.txt_lo);
                    const Sym *pw = (const Sym *)(((uintptr_t)bp.pat_hi << 32) | bp.pat_lo);
                    const uint32_t tsym = (uint32_t)tp[i];
                    const int sp = bp.start0 + (int)i;     // pattern index of window bit 0
                    const int base = sp + (64 - WBITS);    // pattern index of the first live bit
                    uint32_t hits_lo = 0, hits_hi = 0;     // live bits only, bit j <-> window bit 64 - WBITS + j
                    if constexpr (sizeof(Sym) == 1) {
                        // WBITS / 4 unaligned dword loads, clamped into the tape
                        uint32_t dws[WBITS / 4];
                        const int lo = bp.pat_min, hi = bp.pat_max - 3;
                        const bool tiny = hi < lo;
#pragma unroll
                        for (int w = 0; w < WBITS / 4; ++w) {
                            const int idx = base + 4 * w;
                            if (!tiny) {
                                int c = idx < lo ? lo : (idx > hi ? hi : idx);
                                uint32_t dw;
                                __builtin_memcpy(&dw, (const uint8_t *)pw + c, 4);
                                int d = idx - c;
                                d = d < -3 ? -3 : (d > 3 ? 3 : d);
                                dws[w] = d >= 0 ? dw >> (8 * d) : dw << (-8 * d);
                            } else {
                                uint32_t dw = 0;
                                for (int u = 0; u < 4; ++u) {
                                    int pos = idx + u;
                                    if (pos >= bp.pat_min && pos <= bp.pat_max) dw |= (uint32_t)((const uint8_t *)pw)[pos] << (8 * u);
                                }
                                dws[w] = dw;
                            }
                        }
                        const uint32_t splat = tsym * 0x01010101u;
#pragma unroll
                        for (int w = 0; w < WBITS / 4; ++w) {
                            uint32_t x = dws[w] ^ splat;   // zero bytes are matches
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                uint32_t hit = ((x >> (8 * u)) & 0xffu) == 0 ? 1u : 0u;
                                int j = 4 * w + u;
                                if (j < 32) hits_lo |= hit << j; else hits_hi |= hit << (j - 32);
                            }
                        }
                    } else {
                        uint32_t syms[WBITS];
#pragma unroll
                        for (int j = 0; j < WBITS; ++j) syms[j] = band_load_sym<Sym>(pw, base + j, bp.pat_min, bp.pat_max);
#pragma unroll
                        for (int j = 0; j < WBITS; ++j) {
                            uint32_t hit = syms[j] == tsym ? 1u : 0u;
                            if (j < 32) hits_lo |= hit << j; else hits_hi |= hit << (j - 32);
                        }
                    }
                    // keep only window rows that exist: 0 <= sp + b < len1
                    unsigned long long hits = ((unsigned long long)hits_hi << 32) | hits_lo;
                    unsigned long long window = WBITS == 64 ? hits : (hits << (64 - WBITS));
                    int first = -sp, last = (int)bp.len1 - 1 - sp;            // valid bits [first, last]
                    first = first < 0 ? 0 : first;
                    unsigned long long valid = 0;
                    if (last >= first && first <= 63) {
                        last = last > 63 ? 63 : last;
                        unsigned long long upto = last == 63 ? ~0ull : ((1ull << (last + 1)) - 1);
                        valid = upto & (~0ull << first);
                    }
                    eq = window & valid;
                }
                eqbuf[q * kBandPitch + col] = eq;
            }
            // ---------------- phase 2: recurrence, lane = pair -------------------------------------------
#pragma unroll 4
            for (int c = 0; c < kBandChunk; ++c) {
                const uint32_t i = i0 + (uint32_t)c;
                if (have && i < len2) {
                    const unsigned long long eq = eqbuf[lane * kBandPitch + c];
                    const uint32_t eq_lo = (uint32_t)eq, eq_hi = (uint32_t)(eq >> 32);
                    // D0 = (((Eq & VP) + VP) ^ VP) | Eq | VN   (64-bit add with carry)
                    uint32_t x_lo = eq_lo & vp_lo, x_hi = eq_hi & vp_hi;
                    uint32_t s_lo = x_lo + vp_lo;
                    uint32_t carry = s_lo < x_lo ? 1u : 0u;
                    uint32_t s_hi = x_hi + vp_hi + carry;
                    uint32_t d0_lo = (s_lo ^ vp_lo) | eq_lo | vn_lo;
                    uint32_t d0_hi = (s_hi ^ vp_hi) | eq_hi | vn_hi;
                    uint32_t hp_lo = vn_lo | ~(d0_lo | vp_lo), hp_hi = vn_hi | ~(d0_hi | vp_hi);
                    uint32_t hn_lo = d0_lo & vp_lo, hn_hi = d0_hi & vp_hi;
                    if ((int)i < diag_cols) {
                        cur += (d0_hi >> 31) ? 0 : 1;                 // one step down the bottom diagonal
                    } else {
                        const int b = (int)len1 - 1 - (start0 + (int)i);  // last pattern row inside the window
                        const uint32_t hp_w = b >= 32 ? hp_hi : hp_lo, hn_w = b >= 32 ? hn_hi : hn_lo;
                        cur += (int)((hp_w >> (b & 31)) & 1u) - (int)((hn_w >> (b & 31)) & 1u);
                    }
                    const uint32_t d1_lo = __builtin_amdgcn_alignbit(d0_hi, d0_lo, 1), d1_hi = d0_hi >> 1;  // D0 >> 1
                    vp_lo = hn_lo | ~(d1_lo | hp_lo); vp_hi = hn_hi | ~(d1_hi | hp_hi);
                    vn_lo = d1_lo & hp_lo; vn_hi = d1_hi & hp_hi;
                }
            }
        }
        if (have) {
            uint32_t d = cur < 0 ? 0u : (uint32_t)cur;
            store_result(args.job, p, (int64_t)(d > k ? k + 1 : d));
        }
    }
}

template <typename Sym, int WBITS>
static void launch_banded_one(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    const size_t lds = kBandWaves * kBandLdsPerWave;
    uint64_t items = (pairs + 63) / 64;
    uint64_t blocks64 = (items + kBandWaves - 1) / kBandWaves;
    uint32_t max_blocks = (uint32_t)scope->compute_units * 2;  // 77 KB per block -> two blocks per CU
    uint32_t blocks = blocks64 > max_blocks ? max_blocks : (uint32_t)blocks64;
    if (!blocks) return;
    static bool attr_set = false;  // one per instantiation
    if (!attr_set) {
        SWH_HIP_CHECK(hipFuncSetAttribute((const void *)k_banded<Sym, WBITS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    StampGuard guard(scope, "banded");
    hipLaunchKernelGGL((k_banded<Sym, WBITS>), dim3(blocks), dim3(256), lds, scope->stream, args, (uint32_t)kClassBanded);
}

void launch_banded(Scope *scope, const KernelArgs &args, uint64_t pairs) {
    const uint32_t k = args.job.bound;
    KernelArgs a = args;
    a.boundary = nullptr;
#define SWH_BAND(SYM)                                                     \
    if (k + 1 <= 8) launch_banded_one<SYM, 8>(scope, a, pairs);           \
    else if (k + 1 <= 16) launch_banded_one<SYM, 16>(scope, a, pairs);    \
    else if (k + 1 <= 36) launch_banded_one<SYM, 36>(scope, a, pairs);    \
    else launch_banded_one<SYM, 64>(scope, a, pairs);
    if (args.sym_bytes == 4) { SWH_BAND(uint32_t) } else { SWH_BAND(uint8_t) }
#undef SWH_BAND
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
