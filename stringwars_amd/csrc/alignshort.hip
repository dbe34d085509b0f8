// alignshort.hip -- Needleman-Wunsch / Smith-Waterman SCORES for word-sized strings (both sides <= 32 bytes), one pair per lane.
//
// The reference's default token mode is `words` (similarities/bench.rs:271): its NW / SW rows (`perform_linear_benchmarks`,
// `perform_affine_benchmarks`, bench.rs:641-699, :967-1026; README.md:70, the words column) score strings of ~5 bytes. The
// wavefront kernels give such a pair a group of sixteen lanes and a four-step refill cadence: 2048 x 2048 words ran at
// 0.05 TCUPS, thirty times below the Levenshtein rows on the same tapes. Here a LANE owns a pair:
//
//   * the DP row lives in W registers (W = 16 or 32 columns: the b string), the a string supplies the rows; a row is
//     W cells updated in place, its left-to-right dependency is a chain of `v_max3` inside one lane -- no DPP, no LDS;
//   * substitution scores come from the engine's 32 x 32 class table exactly as in the wavefront class models
//     (wavefront.hip): per row ONE 32-byte cost row is read from LDS (the row symbol's class; two `ds_read_b128`, a
//     broadcast when the whole wave scores the same query) and every group of four columns picks its four bytes with one
//     `v_perm_b32` per eight classes in use (PQ) through selectors prepared once per b string; the cell is then
//     `v_add_u32_sdwa (sext byte)` + `v_max3_i32` (Gotoh: two maxima more), with the same baseline-relative forms and the same
//     biased table (global: sub - ext - open, local: sub - open) as the wavefront kernels, so the tables are shared;
//   * pairwise batches run in tape order, 64 consecutive pairs per wave item; the reference's own call shape,
//     `compute_into(queries, candidates, &mut matrix)` (bench.rs:478-486), keeps 64 CANDIDATES in the lanes of a wave (classes
//     and selectors prepared once) and walks a block of queries over them, like k_cross_short.
//
// Exact for strings of up to W symbols; a longer string raises the call summary's `violation` flag and the host redoes the
// call on the planned path (api.hip), as for the other plan-free kernels.
#include <algorithm>

#include "common.hpp"
#include "bp_window.hpp"

namespace swh {

constexpr int kAlignQueries = 16;           // queries per cross-product work item
constexpr int kAlignWaves = 4;
constexpr int kAlignNegInf = -0x20000000;
constexpr size_t kClassLdsBytes = 32 * 32 + 256;   // 32 x 32 i8 class costs, then the byte -> class map (wavefront.hip: kClassLds)

struct AlignShortArgs {
    Job job;
    uint32_t off64;
    int open, extend;
    const uint8_t *class_table;   // 32 x 32 biased class costs, then the byte -> class map (wavefront.hip: kClassLds)
    PlanPartial *partials;
    uint32_t *done_counter;
    CallSummary *summary;
};

__device__ __forceinline__ void align_extent(const void *offsets, uint32_t off64, uint64_t i, uint64_t &start, uint32_t &len) {
    if (off64) { const uint64_t *o = (const uint64_t *)offsets; const uint64_t x0 = o[i], x1 = o[i + 1]; start = x0; len = (uint32_t)(x1 - x0); }
    else { const uint32_t *o = (const uint32_t *)offsets; const uint32_t x0 = o[i], x1 = o[i + 1]; start = x0; len = (uint32_t)(x1 - x0); }
}

// bytes [start, start + 4 * WORDS) of a tape as dwords; what lies past the string is whatever the tape holds there (or a clamped
// window's garbage): the callers never let it decide anything
template <int WORDS>
__device__ __forceinline__ void align_fetch(const uint8_t *data, uint64_t start, uint64_t total, uint32_t (&w)[WORDS]) {
    ByteWindow win;
    win.init(data, start, total);
    if (total >= 16) {
#pragma unroll
        for (int h = 0; h < WORDS / 4; ++h) {
            uint32_t half[4];
            const int moved = win.fetch16_raw(16 * h, half);
            win.fix16(16 * h, moved, half);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[4 * h + q] = half[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < WORDS; ++q) w[q] = win.fetch4(q * 4);
    }
}

// bytes -> classes, four per dword, through the byte -> class map in LDS; only the first `upto` (wave-uniform) symbols matter
template <int WORDS>
__device__ __forceinline__ void align_classes(const uint8_t *lclass_of, const uint32_t (&w)[WORDS], uint32_t upto, uint32_t (&cls)[WORDS]) {
#pragma unroll
    for (int q = 0; q < WORDS; ++q) {
        cls[q] = 0;
        if ((uint32_t)(4 * q) < upto) {
            const uint32_t c0 = lclass_of[w[q] & 0xffu], c1 = lclass_of[(w[q] >> 8) & 0xffu];
            const uint32_t c2 = lclass_of[(w[q] >> 16) & 0xffu], c3 = lclass_of[w[q] >> 24];
            cls[q] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
        }
    }
}

// v_perm selectors of a b string: per group of four columns and per pair of cost-row dwords pq, byte i is (class & 7) when
// the class lives in dwords 2 pq .. 2 pq + 1 of the row, else 0x0C (a zero byte). Packed arithmetic, four columns at a time:
// m = classes ^ (pq << 3) has members in 0..7 and everything else in 8..31; (m + 0x78) sets bit 7 exactly for the others.
// Local alignment: columns right of the string select zero everywhere = a substitution score of `open` <= 0, so a phantom
// cell never exceeds the real cell it descends from and the running maximum needs no column test (wavefront.hip).
template <int W, int PQ, bool kLocal>
__device__ __forceinline__ void align_selectors(const uint32_t (&cls)[W / 4], uint32_t n, uint32_t (&sel)[(W / 4) * PQ]) {
#pragma unroll
    for (int g = 0; g < W / 4; ++g) {
        uint32_t beyond = 0;
        if constexpr (kLocal) beyond = n >= (uint32_t)(4 * g + 4) ? 0u : (n <= (uint32_t)(4 * g) ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (n - 4 * g)));
#pragma unroll
        for (int pq = 0; pq < PQ; ++pq) {
            const uint32_t m = cls[g] ^ (0x08080808u * (uint32_t)pq);
            const uint32_t t = (m + 0x78787878u) & 0x80808080u;
            const uint32_t others = ((t - (t >> 7)) | t) | beyond;           // 0xFF in the bytes that select nothing
            sel[g * PQ + pq] = (m & ~others) | (0x0C0C0C0Cu & others);
        }
    }
}

// SWH_LOCAL_LINEAR_CELL (round 6) -- Smith-Waterman with linear gaps, p = |open| = |extend|. With the true H >= 0 in the strip,
//     H = max(0, Hdiag + s, Hup - p, Hleft - p) = sat( max3(Hdiag + (s + p), Hup, Hleft) - p )
// (the maximum is >= Hup >= 0, so "max with 0 after subtracting p" is one unsigned saturating subtraction, `v_sub_u32 ... clamp`): the
// class table already holds s - open = s + p, and a cell is `v_add_u32_sdwa ; v_max3_i32 ; v_sub_u32 clamp` -- three instructions where
// max(., 0), the running maximum and + open made it five. The running maximum is taken over the max3 values, two per v_max3, and
// loses its p once, at the end. (Gotoh keeps H + open in its strips: sharing sat(H - |open|) between the cell below and the cell to
// the right needs a second register row, which the 128-column kernels do not have.)
__device__ __forceinline__ int align_local_linear_best(int best_of_max3, int open) {
    return (int)__builtin_elementwise_sub_sat((uint32_t)best_of_max3, (uint32_t)-open);
}

// One group of four cells of one DP row, updated in place: `H` holds the row above on entry (the previous row's values) and this
// row's on exit. `c4` = the four substitution scores (biased as the table is), `diag` / `left` / `e` travel along the row. The four
// diagonal sums read the OLD row before any cell of the group is overwritten; the last instruction of a cell has H[k] as a tied
// operand so that the row never moves to other registers (wavefront.hip: in-place strips).
template <int W, bool kAffine, bool kLocal>
__device__ __forceinline__ void align_group(int (&H)[W], int (&F)[kAffine ? W : 1], int g4, uint32_t c4, int &diag, int &left, int &e, int &best,
                                            int open, int ext, int open_minus_ext) {
    constexpr bool kSkew = !kAffine && !kLocal;          // U = H - (r + k) g:  U = max3(U_diag + (s - 2g), U_up, U_left), boundaries 0
    constexpr bool kSkewAffine = kAffine && !kLocal;     // strips hold H^ + (open - ext): see wavefront.hip
    int t[4];
    t[0] = diag + (int)(int8_t)c4;
    t[1] = H[g4] + (int)(int8_t)(c4 >> 8);
    t[2] = H[g4 + 1] + (int)(int8_t)(c4 >> 16);
    t[3] = H[g4 + 2] + (int)(int8_t)(c4 >> 24);
    diag = H[g4 + 3];
    [[maybe_unused]] const int gap = -open;   // (local, linear) |open| = |extend|
    [[maybe_unused]] int mx_even = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = g4 + u;
        if constexpr (kSkew) {
            asm("v_max3_i32 %0, %1, %0, %2" : "+v"(H[k]) : "v"(t[u]), "v"(left));
        } else if constexpr (kSkewAffine) {
            const int f = max(H[k], F[k]);
            F[k] = f;
            e = max(left, e);
            const int h3 = max(max(t[u], e), f);
            asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open_minus_ext));
        } else if constexpr (kAffine) {   // local, Gotoh: strips hold H + open
            const int f = max(H[k], F[k] + ext);
            F[k] = f;
            e = max(left, e + ext);
            const int h3 = max(max(max(t[u], e), f), 0);
            best = max(best, h3);
            asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open));
        } else {                          // local, linear: the strip holds the TRUE H >= 0 (SWH_LOCAL_LINEAR_CELL)
            const int mx = max(max(t[u], H[k]), left);
            if (u & 1) best = max(max(best, mx_even), mx); else mx_even = mx;
            asm("v_sub_u32_e64 %0, %1, %2 clamp" : "+v"(H[k]) : "v"(mx), "v"(gap));
        }
        left = H[k];
    }
}

template <int PQ>
__device__ __forceinline__ uint32_t align_costs4(const uint4 &r_lo, const uint4 &r_hi, const uint32_t *sg) {
    uint32_t c4 = __builtin_amdgcn_perm(r_lo.y, r_lo.x, sg[0]);
    if constexpr (PQ > 1) c4 |= __builtin_amdgcn_perm(r_lo.w, r_lo.z, sg[PQ > 1 ? 1 : 0]);
    if constexpr (PQ > 2) c4 |= __builtin_amdgcn_perm(r_hi.y, r_hi.x, sg[PQ > 2 ? 2 : 0]);
    if constexpr (PQ > 3) c4 |= __builtin_amdgcn_perm(r_hi.w, r_hi.z, sg[PQ > 3 ? 3 : 0]);
    return c4;
}

// The rows of one pair (lane): `acls` the a string's classes, m rows of it, n columns prepared in `sel`; m_max / n_max are the
// wave's maxima (uniform loop bounds). Returns the score of the global alignment or the best local one; pairs with an empty
// side are the caller's.
template <int W, int PQ, bool kAffine, bool kLocal>
__device__ __forceinline__ int align_rows(const uint32_t (&acls)[W / 4], uint32_t m, uint32_t m_max, const uint32_t (&sel)[(W / 4) * PQ],
                                          uint32_t n, uint32_t n_max, const char *ltable, int open, int ext) {
    constexpr bool kSkew = !kAffine && !kLocal;          // U = H - (r + k) g:  U = max3(U_diag + (s - 2g), U_up, U_left), boundaries 0
    constexpr bool kSkewAffine = kAffine && !kLocal;     // strips hold H^ + (open - ext): see wavefront.hip
    const int open_minus_ext = open - ext;
    int H[W];
    [[maybe_unused]] int F[kAffine ? W : 1];
    const int row0 = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : 2 * open_minus_ext);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        H[k] = row0;
        if constexpr (kAffine) F[k] = kAlignNegInf;
    }
    uint32_t rowq[W / 4];
#pragma unroll
    for (int q = 0; q < W / 4; ++q) rowq[q] = acls[q];
    int best = 0;
    uint4 lo_next = *(const uint4 *)(ltable + (rowq[0] & 0xffu) * 32);
    uint4 hi_next{0, 0, 0, 0};
    if constexpr (PQ > 2) hi_next = *(const uint4 *)(ltable + (rowq[0] & 0xffu) * 32 + 16);
    for (uint32_t i = 0; i < m_max; ++i) {
        const uint4 r_lo = lo_next, r_hi = hi_next;
        // the next row's class moves down to byte 0; its cost row is requested before this row's cells
#pragma unroll
        for (int q = 0; q < W / 4; ++q) rowq[q] = q + 1 < W / 4 ? __builtin_amdgcn_alignbyte(rowq[q + 1 < W / 4 ? q + 1 : q], rowq[q], 1) : rowq[q] >> 8;
        lo_next = *(const uint4 *)(ltable + (rowq[0] & 0xffu) * 32);
        if constexpr (PQ > 2) hi_next = *(const uint4 *)(ltable + (rowq[0] & 0xffu) * 32 + 16);
        if (i < m) {
            // boundary column: H of (row i + 1, column 0) on the left, of (row i, column 0) on the diagonal
            int left = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : 2 * open_minus_ext);
            int diag = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : (i == 0 ? open_minus_ext : 2 * open_minus_ext));
            [[maybe_unused]] int e = kAlignNegInf;
#pragma unroll
            for (int g4 = 0; g4 < W; g4 += 4) {
                if ((uint32_t)g4 < n_max) {
                    const uint32_t *sg = sel + (g4 >> 2) * PQ;
                    uint32_t c4 = __builtin_amdgcn_perm(r_lo.y, r_lo.x, sg[0]);
                    if constexpr (PQ > 1) c4 |= __builtin_amdgcn_perm(r_lo.w, r_lo.z, sg[PQ > 1 ? 1 : 0]);
                    if constexpr (PQ > 2) c4 |= __builtin_amdgcn_perm(r_hi.y, r_hi.x, sg[PQ > 2 ? 2 : 0]);
                    if constexpr (PQ > 3) c4 |= __builtin_amdgcn_perm(r_hi.w, r_hi.z, sg[PQ > 3 ? 3 : 0]);
                    // the four diagonal sums read the OLD row before any cell of the group is overwritten
                    int t[4];
                    t[0] = diag + (int)(int8_t)c4;
                    t[1] = H[g4] + (int)(int8_t)(c4 >> 8);
                    t[2] = H[g4 + 1] + (int)(int8_t)(c4 >> 16);
                    t[3] = H[g4 + 2] + (int)(int8_t)(c4 >> 24);
                    diag = H[g4 + 3];
                    [[maybe_unused]] const int gap = -open;   // (local, linear) |open| = |extend|
                    [[maybe_unused]] int mx_even = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = g4 + u;
                        if constexpr (kSkew) {
                            asm("v_max3_i32 %0, %1, %0, %2" : "+v"(H[k]) : "v"(t[u]), "v"(left));
                        } else if constexpr (kSkewAffine) {
                            const int f = max(H[k], F[k]);
                            F[k] = f;
                            e = max(left, e);
                            const int h3 = max(max(t[u], e), f);
                            asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open_minus_ext));
                        } else if constexpr (kAffine) {   // local, Gotoh: strips hold H + open
                            const int f = max(H[k], F[k] + ext);
                            F[k] = f;
                            e = max(left, e + ext);
                            const int h3 = max(max(max(t[u], e), f), 0);
                            best = max(best, h3);
                            asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open));
                        } else {                          // local, linear: the strip holds the TRUE H >= 0 (SWH_LOCAL_LINEAR_CELL)
                            const int mx = max(max(t[u], H[k]), left);
                            if (u & 1) best = max(max(best, mx_even), mx); else mx_even = mx;
                            asm("v_sub_u32_e64 %0, %1, %2 clamp" : "+v"(H[k]) : "v"(mx), "v"(gap));
                        }
                        left = H[k];
                    }
                }
            }
        }
    }
    if constexpr (kLocal) return kAffine ? best : align_local_linear_best(best, open);
    int result = 0;
#pragma unroll
    for (int k = 0; k < W; ++k)
        if ((uint32_t)k + 1 == n) result = H[k];
    if constexpr (kSkew) result += (int)(m + n) * ext;
    else result += (int)(m + n) * ext - open_minus_ext;
    return result;
}

// The same when the WHOLE WAVE walks the same rows (queries x candidates: one query against 64 candidates): `rowcls` points at
// the query's class bytes in LDS, every lane runs all `m` rows (lanes without a candidate compute something nobody looks at), and
// rows are processed TWO AT A TIME, the second one group of four columns behind the first. A row is a chain of W dependent
// maxima inside one lane, and a wave issues a dependent instruction only every ~8 cycles (tools/valu_chain.hip): with two or
// three waves per SIMD (a 128-column row and its selectors fill the register file) the chain, not the issue rate, set the pace.
// Row i + 1's group g - 1 needs row i's values of columns 4g - 5 .. 4g - 1: what row i wrote one step earlier, still in place --
// the pair shares ONE register row, its two chains are independent, and the instruction count is unchanged.
template <int W, int PQ, bool kAffine, bool kLocal>
__device__ __forceinline__ int align_rows_uniform(const uint8_t *rowcls, uint32_t m, const uint32_t (&sel)[(W / 4) * PQ], uint32_t n, uint32_t n_max,
                                                  const char *ltable, uint32_t row_bytes, int open, int ext) {
    constexpr bool kSkew = !kAffine && !kLocal;
    const int open_minus_ext = open - ext;
    int H[W];
    int F[kAffine ? W : 1];
    const int row0 = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : 2 * open_minus_ext);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        H[k] = row0;
        if constexpr (kAffine) F[k] = kAlignNegInf;
    }
    int best = 0;
    // a row's costs: `row_bytes` = 32 (the class table itself) or 8 (a table compacted to the classes in use: PQ == 1, r_lo.x / .y)
    auto fetch = [&](uint32_t row, uint4 &lo, uint4 &hi) {
        const uint32_t rc = row < m ? rowcls[row] : 0u;
        if (row_bytes == 8) { const uint2 v = *(const uint2 *)(ltable + rc * 8); lo = make_uint4(v.x, v.y, 0, 0); }
        else {
            lo = *(const uint4 *)(ltable + rc * 32);
            if constexpr (PQ > 2) hi = *(const uint4 *)(ltable + rc * 32 + 16);
        }
    };
    uint4 a_lo{0, 0, 0, 0}, a_hi{0, 0, 0, 0}, b_lo{0, 0, 0, 0}, b_hi{0, 0, 0, 0};
    fetch(0, a_lo, a_hi);
    fetch(1, b_lo, b_hi);
    const int edge = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : 2 * open_minus_ext);
    uint32_t i = 0;
    for (; i + 1 < m; i += 2) {
        const uint4 ra_lo = a_lo, ra_hi = a_hi, rb_lo = b_lo, rb_hi = b_hi;
        fetch(i + 2, a_lo, a_hi);      // the next pair's costs are requested before this pair's cells
        fetch(i + 3, b_lo, b_hi);
        int left_a = edge, left_b = edge, e_a = kAlignNegInf, e_b = kAlignNegInf;
        int diag_a = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : (i == 0 ? open_minus_ext : 2 * open_minus_ext)), diag_b = edge;
#pragma unroll
        for (int g4 = 0; g4 <= W; g4 += 4) {
            if (g4 < W && (uint32_t)g4 < n_max)
                align_group<W, kAffine, kLocal>(H, F, g4, align_costs4<PQ>(ra_lo, ra_hi, sel + (g4 >> 2) * PQ), diag_a, left_a, e_a, best, open, ext, open_minus_ext);
            if (g4 >= 4 && (uint32_t)(g4 - 4) < n_max)
                align_group<W, kAffine, kLocal>(H, F, g4 - 4, align_costs4<PQ>(rb_lo, rb_hi, sel + ((g4 - 4) >> 2) * PQ), diag_b, left_b, e_b, best, open, ext, open_minus_ext);
        }
    }
    if (i < m) {   // an odd row count: the last row on its own
        int left = edge, e = kAlignNegInf;
        int diag = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : (i == 0 ? open_minus_ext : 2 * open_minus_ext));
#pragma unroll
        for (int g4 = 0; g4 < W; g4 += 4)
            if ((uint32_t)g4 < n_max)
                align_group<W, kAffine, kLocal>(H, F, g4, align_costs4<PQ>(a_lo, a_hi, sel + (g4 >> 2) * PQ), diag, left, e, best, open, ext, open_minus_ext);
    }
    if constexpr (kLocal) return kAffine ? best : align_local_linear_best(best, open);
    int result = 0;
#pragma unroll
    for (int k = 0; k < W; ++k)
        if ((uint32_t)k + 1 == n) result = H[k];
    if constexpr (kSkew) result += (int)(m + n) * ext;
    else result += (int)(m + n) * ext - open_minus_ext;
    return result;
}

__device__ __forceinline__ int align_trivial(uint32_t la, uint32_t lb, bool local, int open, int ext) {
    const uint32_t len = la + lb;
    return (len && !local) ? open + (int)(len - 1) * ext : 0;   // gap(k) = open + (k - 1) extend; two empty strings score 0
}

struct AlignWaveLds {
    uint32_t qlen[kAlignQueries];
    uint32_t qcls[kAlignQueries][16];    // the item's queries as class bytes (<= 64 per query)
};

template <int W, int PQ, bool kAffine, bool kLocal>
__global__ __launch_bounds__(kAlignWaves * 64) void k_align_short(AlignShortArgs args) {
    __shared__ __attribute__((aligned(16))) char ltable[kClassLdsBytes];
    __shared__ AlignWaveLds wave_lds[kAlignWaves];
    __shared__ SummaryLds summary_lds;
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, lmisfit;
    {
        const uint32_t *src = (const uint32_t *)args.class_table;
        for (int i = threadIdx.x; i < (int)kClassLdsBytes / 4; i += blockDim.x) ((uint32_t *)ltable)[i] = src[i];
    }
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lmisfit = 0; }
    __syncthreads();
    const uint8_t *lclass_of = (const uint8_t *)ltable + 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Job &job = args.job;
    const uint8_t *a_data = (const uint8_t *)job.a.data, *b_data = (const uint8_t *)job.b.data;
    const uint64_t na = job.a.count, nb = job.b.count;
    uint64_t a_total, b_total;
    { uint32_t unused; align_extent(job.a.offsets, args.off64, na, a_total, unused); align_extent(job.b.offsets, args.off64, nb, b_total, unused); }
    const uint64_t waves_total = (uint64_t)gridDim.x * kAlignWaves, wave_id = (uint64_t)blockIdx.x * kAlignWaves + wave;
    const int open = args.open, ext = args.extend;
    const size_t elem = job.out_elem64 ? 8 : 4;
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, misfit = 0;

    if (!job.cross) {
        // ---- pairwise: 64 consecutive pairs per item, tape order -------------------------------------------------------------------
        const uint64_t items = (job.pairs + 63) / 64;
        for (uint64_t item = wave_id; item < items; item += waves_total) {
            const uint64_t p = item * 64 + (uint64_t)lane;
            const bool have = p < job.pairs;
            uint64_t a0 = 0, b0 = 0;
            uint32_t la = 0, lb = 0;
            if (have) { align_extent(job.a.offsets, args.off64, p, a0, la); align_extent(job.b.offsets, args.off64, p, b0, lb); }
            uint32_t aw[W / 4], bw[W / 4];
            align_fetch<W / 4>(a_data, a0, a_total, aw);
            align_fetch<W / 4>(b_data, b0, b_total, bw);
            const bool fits = have && la <= (uint32_t)W && lb <= (uint32_t)W;
            if (have && !fits) misfit |= 1;
            const bool runs = fits && la && lb;
            const uint32_t m = runs ? la : 0u, n = runs ? lb : 0u;
            const uint32_t m_max = wave_max_u32(m), n_max = wave_max_u32(n);
            uint32_t acls[W / 4], bcls[W / 4], sel[(W / 4) * PQ];
            align_classes<W / 4>(lclass_of, aw, m_max, acls);
            align_classes<W / 4>(lclass_of, bw, n_max, bcls);
            align_selectors<W, PQ, kLocal>(bcls, n, sel);
            int score = align_rows<W, PQ, kAffine, kLocal>(acls, m, m_max, sel, n, n_max, ltable, open, ext);
            if (fits) {
                if (!runs) score = align_trivial(la, lb, kLocal, open, ext);
                char *dst = job.out + p * job.out_stride;
                store_out(dst, job.out_elem64 != 0, (int64_t)score);
                shorts += 1;
            }
            if (have) {
                cells += (unsigned long long)la * lb;
                syms += (unsigned long long)la + lb;
                maxa = la > maxa ? la : maxa;
                maxb = lb > maxb ? lb : maxb;
            }
        }
    } else {
        // ---- queries x candidates: a wave keeps 64 candidates (columns) and walks a block of queries (rows) over them --------------
        AlignWaveLds &wl = wave_lds[wave];
        const uint64_t chunks = (nb + 63) / 64, qblocks = (na + kAlignQueries - 1) / kAlignQueries;
        const uint64_t items = chunks * qblocks;
        for (uint64_t item = wave_id; item < items; item += waves_total) {
            const uint64_t chunk = item / qblocks, qb = item - chunk * qblocks;
            const uint64_t q_first = qb * kAlignQueries, q_last = q_first + kAlignQueries < na ? q_first + kAlignQueries : na;
            const uint32_t q_count = (uint32_t)(q_last - q_first);
            const uint64_t cand = chunk * 64 + (uint64_t)lane;
            const bool have = cand < nb;
            uint64_t b0 = 0;
            uint32_t lb = 0;
            if (have) align_extent(job.b.offsets, args.off64, cand, b0, lb);
            // queries: lane l stages bytes kLaneBytes (l % 4) .. of query l / 4 as classes (16 queries x 32 bytes = 64 lanes x 8 bytes; rows of
            // 64 cells: 64 bytes per query, 16 per lane)
            constexpr int kLaneBytes = W > 32 ? 16 : 8, kLaneWords = kLaneBytes / 4;
            const uint32_t ql = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
            uint64_t qa0 = 0;
            uint32_t qm = 0;
            if (ql < q_count) align_extent(job.a.offsets, args.off64, q_first + ql, qa0, qm);
            uint32_t staged[kLaneWords];
#pragma unroll
            for (int t = 0; t < kLaneWords; ++t) staged[t] = 0;
#pragma unroll
            for (int t = 0; t < kLaneBytes; ++t) {
                const uint32_t at = part * kLaneBytes + (uint32_t)t;
                const uint32_t byte = (at < qm && qm <= (uint32_t)W) ? a_data[qa0 + at] : 0u;
                staged[t >> 2] |= byte << (8 * (t & 3));
            }
            uint32_t bw[W / 4];
            align_fetch<W / 4>(b_data, b0, b_total, bw);
            const bool fits = have && lb <= (uint32_t)W;
            if (have && !fits) misfit |= 1;
            const uint32_t n = fits ? lb : 0u;
            const uint32_t n_max = wave_max_u32(n);
            uint32_t bcls[W / 4], sel[(W / 4) * PQ];
            align_classes<W / 4>(lclass_of, bw, n_max, bcls);
            align_selectors<W, PQ, kLocal>(bcls, n, sel);
            {
                uint32_t scls[kLaneWords];
                align_classes<kLaneWords>(lclass_of, staged, kLaneBytes, scls);
                wave_lds_fence();                              // the previous item's readers are done with the staging area
#pragma unroll
                for (int t = 0; t < kLaneWords; ++t) wl.qcls[ql][part * kLaneWords + t] = scls[t];
                if (part == 0) wl.qlen[ql] = ql < q_count ? qm : 0u;
                wave_lds_fence();
            }
            unsigned long long sum_m = 0;
            uint32_t item_maxa = 0;
            for (uint32_t q = 0; q < q_count; ++q) {
                const uint32_t qlen = wl.qlen[q];
                sum_m += qlen;
                item_maxa = qlen > item_maxa ? qlen : item_maxa;
                if (qlen > (uint32_t)W) { misfit |= 1; continue; }
                // (rows two at a time -- align_rows_uniform -- pay for strings of ~100 symbols; on words of ~5 the pair's extra step
                // and registers cost more than the second chain returns: 2048 x 2048 words 1.40 -> 1.00 TCUPS, measured)
                uint32_t acls[W / 4];
#pragma unroll
                for (int w4 = 0; w4 < W / 4; ++w4) acls[w4] = wl.qcls[q][w4];
                const uint32_t m = (n && qlen) ? qlen : 0u;
                int score = align_rows<W, PQ, kAffine, kLocal>(acls, m, qlen, sel, n, n_max, ltable, open, ext);
                if (fits) {
                    if (!m) score = align_trivial(qlen, lb, kLocal, open, ext);
                    char *dst = job.out + (q_first + q) * job.row_stride + cand * elem;
                    store_out(dst, job.out_elem64 != 0, (int64_t)score);
                }
            }
            if (have) {
                cells += sum_m * (unsigned long long)lb;
                maxb = lb > maxb ? lb : maxb;
                if (qb == 0) syms += lb;                                  // every candidate once ...
                if (fits) shorts += q_count;
            }
            if (lane == 0) {
                maxa = item_maxa > maxa ? item_maxa : maxa;
                if (chunk == 0) syms += sum_m;                            // ... and every query once (bench.rs:216-224)
            }
        }
    }
    // ---- summary ---------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cells += __shfl_xor(cells, off);
        syms += __shfl_xor(syms, off);
        shorts += __shfl_xor(shorts, off);
        misfit |= __shfl_xor(misfit, off);
        const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
        maxa = oa > maxa ? oa : maxa;
        maxb = ob > maxb ? ob : maxb;
    }
    if (lane == 0) {
        atomicAdd(&lcells, cells);
        atomicAdd(&lsyms, syms);
        atomicAdd(&lshorts, shorts);
        atomicMax(&lmaxa, maxa);
        atomicMax(&lmaxb, maxb);
        atomicOr(&lmisfit, misfit);
    }
    __syncthreads();
    report_call_summary(PlanPartial{lcells, lsyms, lmaxa, lmaxb, lshorts, lmisfit}, args.partials, args.done_counter, args.summary, summary_lds);
}

// ---- queries x candidates on strings of up to 128 symbols over a SMALL alphabet (DNA: the reference's ACGT datasets, README.md:38) ---
// A row of 128 cells still fits a lane's registers when nothing else has to: the v_perm selectors of a 32-class table are four
// registers per four columns, as many as the cells themselves. So the classes are COMPACTED per work item: the (at most eight)
// classes that occur in the item's 64 candidates get the ids 0..7, a 32 x 8 byte cost table for them is built in LDS (256 bytes per
// wave), every row fetches its eight costs with one broadcast `ds_read_b64`, and a group of four columns needs ONE selector and
// ONE v_perm: 2.5 instructions per cell (perm / 4 + sdwa add + max3 + a quarter move), 160 registers for the row and its
// selectors. An item whose candidates use more than eight classes raises `violation` (the call is redone on the planned path and
// the scope stops trying). Linear gaps only (Gotoh's second row would not fit), global and local.
struct AlignWideLds {
    uint32_t qlen[kAlignQueries];
    uint8_t qcls[kAlignQueries][128];    // the item's queries as class bytes
    uint8_t ctab[32][8];                 // cost of (row class, compact column class)
    uint8_t cid[32];                     // class -> compact id
};

template <int W, bool kLocal>
// (two waves per SIMD, i.e. at most 256 registers, stated: with `misfit |= 1` in place of `= 1` below -- round 6 -- hipcc gave the
// W = 128 instantiation 257 registers, one wave per SIMD, and the ACGT-100 cross-product went from 4.0 to 6.4 ms: tools/cross_nw_probe.py)
__global__ __launch_bounds__(kAlignWaves * 64, 2) void k_align_cross_wide(AlignShortArgs args) {
    __shared__ __attribute__((aligned(16))) char ltable[kClassLdsBytes];
    __shared__ __attribute__((aligned(16))) AlignWideLds wave_lds[kAlignWaves];
    __shared__ SummaryLds summary_lds;
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, lmisfit;
    {
        const uint32_t *src = (const uint32_t *)args.class_table;
        for (int i = threadIdx.x; i < (int)kClassLdsBytes / 4; i += blockDim.x) ((uint32_t *)ltable)[i] = src[i];
    }
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lmisfit = 0; }
    __syncthreads();
    const uint8_t *lclass_of = (const uint8_t *)ltable + 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    AlignWideLds &wl = wave_lds[wave];
    const Job &job = args.job;
    const uint8_t *a_data = (const uint8_t *)job.a.data, *b_data = (const uint8_t *)job.b.data;
    const uint64_t na = job.a.count, nb = job.b.count;
    uint64_t a_total, b_total;
    { uint32_t unused; align_extent(job.a.offsets, args.off64, na, a_total, unused); align_extent(job.b.offsets, args.off64, nb, b_total, unused); }
    const uint64_t waves_total = (uint64_t)gridDim.x * kAlignWaves, wave_id = (uint64_t)blockIdx.x * kAlignWaves + wave;
    const int open = args.open, ext = args.extend;
    const size_t elem = job.out_elem64 ? 8 : 4;
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, misfit = 0;
    uint32_t rich = 0;   // the alphabet bit, sticky and joined to `misfit` behind the items (the `misfit = 1` assignments below would drop it)
    const uint64_t chunks = (nb + 63) / 64, qblocks = (na + kAlignQueries - 1) / kAlignQueries;
    const uint64_t items = chunks * qblocks;
    for (uint64_t item = wave_id; item < items; item += waves_total) {
        const uint64_t chunk = item / qblocks, qb = item - chunk * qblocks;
        const uint64_t q_first = qb * kAlignQueries, q_last = q_first + kAlignQueries < na ? q_first + kAlignQueries : na;
        const uint32_t q_count = (uint32_t)(q_last - q_first);
        const uint64_t cand = chunk * 64 + (uint64_t)lane;
        const bool have = cand < nb;
        uint64_t b0 = 0;
        uint32_t lb = 0;
        if (have) align_extent(job.b.offsets, args.off64, cand, b0, lb);
        const bool fits = have && lb <= (uint32_t)W;
        if (have && !fits) misfit = 1;
        const uint32_t n = fits ? lb : 0u;
        const uint32_t n_max = wave_max_u32(n);
        // -- my candidate: bytes -> classes (packed four to a dword), and the set of classes it uses
        uint32_t bcls[W / 4];
        uint32_t used = 0;
        {
            uint32_t bw[W / 4];
            align_fetch<W / 4>(b_data, b0, b_total, bw);
            align_classes<W / 4>(lclass_of, bw, n_max, bcls);
#pragma unroll
            for (int g = 0; g < W / 4; ++g) {
                if ((uint32_t)(4 * g) < n_max) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if ((uint32_t)(4 * g + u) < n) used |= 1u << ((bcls[g] >> (8 * u)) & 31u);
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) used |= (uint32_t)__shfl_xor((int)used, off);
        const bool compact = __popc(used) <= 8;
        if (!compact) rich = 2;   // (bit 1: the alphabet is too rich for the compacting kernels -- CallSummary::violation)
        // -- the item's queries into LDS as class bytes: lane l stages bytes 16 (l % 8) .. + 15 of query l / 8, two rounds of eight queries
        wave_lds_fence();                                  // the previous item's readers are done with the staging area
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const uint32_t ql = (uint32_t)round * 8 + ((uint32_t)lane >> 3), part = (uint32_t)lane & 7u;
            uint64_t qa0 = 0;
            uint32_t qm = 0;
            if (ql < q_count) align_extent(job.a.offsets, args.off64, q_first + ql, qa0, qm);
            uint32_t staged[4] = {0, 0, 0, 0};
            if (16 * part < qm && qm <= (uint32_t)W) {
                ByteWindow win;
                win.init(a_data, qa0 + 16 * part, a_total);
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) staged[w4] = win.fetch4(4 * w4);
            }
            uint32_t scls[4];
            align_classes<4>(lclass_of, staged, 16, scls);
            if (16 * part < (uint32_t)W) *(uint4 *)(wl.qcls[ql] + 16 * part) = make_uint4(scls[0], scls[1], scls[2], scls[3]);
            if (part == 0) wl.qlen[ql] = ql < q_count ? qm : 0u;
        }
        // -- compaction: class -> id for the classes in use, and the 32 x 8 cost table of (row class, id)
        if (lane < 32) {
            wl.cid[lane] = (uint8_t)__popc(used & ((1u << lane) - 1u));
            uint32_t rest = used, lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t cls = rest ? (uint32_t)__builtin_ctz(rest) : 0u;
                const uint32_t cost = rest ? (uint32_t)(uint8_t)ltable[lane * 32 + cls] : 0u;
                rest &= rest - 1u;
                if (j < 4) lo |= cost << (8 * j); else hi |= cost << (8 * (j - 4));
            }
            *(uint2 *)wl.ctab[lane] = make_uint2(lo, hi);
        }
        wave_lds_fence();
        // -- selectors: the compact ids of my candidate's symbols; local alignment: nothing right of the string
        uint32_t sel[W / 4];
#pragma unroll
        for (int g = 0; g < W / 4; ++g) {
            sel[g] = 0x0C0C0C0Cu;
            if ((uint32_t)(4 * g) < n_max) {
                const uint32_t c = bcls[g];
                uint32_t ids = (uint32_t)wl.cid[c & 31u] | ((uint32_t)wl.cid[(c >> 8) & 31u] << 8) | ((uint32_t)wl.cid[(c >> 16) & 31u] << 16) | ((uint32_t)wl.cid[(c >> 24) & 31u] << 24);
                ids &= 0x07070707u;
                if constexpr (kLocal) {
                    const uint32_t beyond = n >= (uint32_t)(4 * g + 4) ? 0u : (n <= (uint32_t)(4 * g) ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (n - 4 * g)));
                    ids = (ids & ~beyond) | (0x0C0C0C0Cu & beyond);
                }
                sel[g] = ids;
            }
        }
        unsigned long long sum_m = 0;
        uint32_t item_maxa = 0;
        for (uint32_t q = 0; q < q_count; ++q) {
            const uint32_t qlen = wl.qlen[q];
            sum_m += qlen;
            item_maxa = qlen > item_maxa ? qlen : item_maxa;
            if (qlen > (uint32_t)W) { misfit = 1; continue; }
            if (!compact) continue;
            int score = align_rows_uniform<W, 1, false, kLocal>(wl.qcls[q], qlen, sel, n, n_max, (const char *)&wl.ctab[0][0], 8u, open, ext);
            if (fits) {
                if (!n || !qlen) score = align_trivial(qlen, lb, kLocal, open, ext);
                char *dst = job.out + (q_first + q) * job.row_stride + cand * elem;
                store_out(dst, job.out_elem64 != 0, (int64_t)score);
            }
        }
        if (have) {
            cells += sum_m * (unsigned long long)lb;
            maxb = lb > maxb ? lb : maxb;
            if (qb == 0) syms += lb;
            if (fits) shorts += q_count;
        }
        if (lane == 0) {
            maxa = item_maxa > maxa ? item_maxa : maxa;
            if (chunk == 0) syms += sum_m;
        }
    }
    misfit |= rich;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cells += __shfl_xor(cells, off);
        syms += __shfl_xor(syms, off);
        shorts += __shfl_xor(shorts, off);
        misfit |= __shfl_xor(misfit, off);
        const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
        maxa = oa > maxa ? oa : maxa;
        maxb = ob > maxb ? ob : maxb;
    }
    if (lane == 0) {
        atomicAdd(&lcells, cells);
        atomicAdd(&lsyms, syms);
        atomicAdd(&lshorts, shorts);
        atomicMax(&lmaxa, maxa);
        atomicMax(&lmaxb, maxb);
        atomicOr(&lmisfit, misfit);
    }
    __syncthreads();
    report_call_summary(PlanPartial{lcells, lsyms, lmaxa, lmaxb, lshorts, lmisfit}, args.partials, args.done_counter, args.summary, summary_lds);
}

// ---- the same for strings of ANY length: the columns run as passes of W, the boundary column between passes travels through
// global memory ------------------------------------------------------------------------------------------------------------------
// A lane still owns a pair; its candidate's columns c0 + 1 .. c0 + W are one pass, the rows of the query stream by two at a time
// (align_rows_uniform's scheme), and what the next pass needs of this one -- H (and Gotoh's E) of the pass's last column, one value
// per row -- is parked in a lane-private column of a per-wave buffer: row r of all 64 lanes is one 256-byte line, written and read
// back coalesced, 8 bytes per 128 cells. The baseline-relative forms carry over unchanged (their recurrences do not mention the
// column index). With the register row bounded by the pass, Gotoh fits too: W = 64 columns of H and F. Queries are staged in LDS
// one at a time (<= kAlignLongRows class bytes); the classes are compacted per work item over ALL of its candidates' columns.
constexpr uint32_t kAlignLongRows = 4096;

struct AlignLongLds {
    uint8_t qcls[kAlignLongRows + 16];   // the current query as class bytes
    uint8_t ctab[32][8];                 // cost of (row class, compact column class)
    uint8_t cid[32];                     // class -> compact id
};

// One pass of one query over my candidate's columns (c0, c0 + W]. `first`: c0 == 0 (the DP's own left edge instead of a parked
// column); `more`: another pass follows (park my right edge). bh / be: my lane's column of the wave's boundary buffer, entry r at
// [64 r]. Returns nothing: H-of-my-last-column is captured into `result` when my string ends inside this pass; `best` runs on.
// kFull: every column of the pass is a column of some candidate of the wave (all passes but a work item's last): no test per group.
template <int W, bool kAffine, bool kLocal, bool kFull>
__device__ __forceinline__ void align_pass(const uint8_t *rowcls, uint32_t m, const uint32_t (&sel)[W / 4], uint32_t n_here, uint32_t cols_here,
                                           const char *ctab, int open, int ext, bool first, bool more, int *bh, int *be, int &result, int &best) {
    constexpr bool kSkew = !kAffine && !kLocal;
    const int open_minus_ext = open - ext;
    int H[W];
    int F[kAffine ? W : 1];
    const int edge = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : 2 * open_minus_ext);   // row 0 right of the corner, and the DP's own left edge below it
    const int corner = kLocal ? (kAffine ? open : 0) : (kSkew ? 0 : (first ? open_minus_ext : 2 * open_minus_ext));
#pragma unroll
    for (int k = 0; k < W; ++k) {
        H[k] = edge;
        if constexpr (kAffine) F[k] = kAlignNegInf;
    }
    auto fetch = [&](uint32_t row, uint2 &lo) {
        const uint32_t rc = row < m ? rowcls[row] : 0u;
        lo = *(const uint2 *)(ctab + rc * 8);
    };
    auto costs4 = [](const uint2 &r, uint32_t selector) -> uint32_t { return __builtin_amdgcn_perm(r.y, r.x, selector); };
    uint2 a_lo, b_lo;
    fetch(0, a_lo);
    fetch(1, b_lo);
    // the parked column: rows i, i + 1 of this iteration, requested an iteration ahead (two iterations ahead measured the same and
    // cost the registers that decide between one and two waves per SIMD); the buffer has slack rows for requests past the query's end
    int h_a = edge, h_b = edge, e_a0 = kAlignNegInf, e_b0 = kAlignNegInf;
    if (!first) {
        h_a = bh[0]; h_b = bh[64];
        if constexpr (kAffine) { e_a0 = be[0]; e_b0 = be[64]; }
    }
    int above = corner;      // H of (row i, column c0): the diagonal of row i + 1's first cell
    uint32_t i = 0;
    for (; i + 1 < m; i += 2) {
        const uint2 ra = a_lo, rb = b_lo;
        fetch(i + 2, a_lo);
        fetch(i + 3, b_lo);
        int left_a = h_a, left_b = h_b, diag_a = above, diag_b = h_a, e_a = e_a0, e_b = e_b0;
        above = h_b;
        if (!first) {
            h_a = bh[64 * (i + 2)]; h_b = bh[64 * (i + 3)];
            if constexpr (kAffine) { e_a0 = be[64 * (i + 2)]; e_b0 = be[64 * (i + 3)]; }
        }
        int out_a = 0, out_b = 0, oute_a = 0, oute_b = 0;
#pragma unroll
        for (int g4 = 0; g4 <= W; g4 += 4) {
            if (g4 < W && (kFull || (uint32_t)g4 < cols_here)) {
                align_group<W, kAffine, kLocal>(H, F, g4, costs4(ra, sel[g4 >> 2]), diag_a, left_a, e_a, best, open, ext, open_minus_ext);
                if (g4 == W - 4) { out_a = left_a; oute_a = e_a; }
            }
            if (g4 >= 4 && (kFull || (uint32_t)(g4 - 4) < cols_here)) {
                align_group<W, kAffine, kLocal>(H, F, g4 - 4, costs4(rb, sel[(g4 - 4) >> 2]), diag_b, left_b, e_b, best, open, ext, open_minus_ext);
                if (g4 == W) { out_b = left_b; oute_b = e_b; }
            }
        }
        if (more) {
            bh[64 * i] = out_a; bh[64 * (i + 1)] = out_b;
            if constexpr (kAffine) { be[64 * i] = oute_a; be[64 * (i + 1)] = oute_b; }
        }
    }
    if (i < m) {   // an odd row count: the last row on its own
        int left = h_a, diag = above, e = e_a0;
#pragma unroll
        for (int g4 = 0; g4 < W; g4 += 4)
            if (kFull || (uint32_t)g4 < cols_here)
                align_group<W, kAffine, kLocal>(H, F, g4, costs4(a_lo, sel[g4 >> 2]), diag, left, e, best, open, ext, open_minus_ext);
        if (more) {
            bh[64 * i] = left;
            if constexpr (kAffine) be[64 * i] = e;
        }
    }
    if constexpr (!kLocal) {
#pragma unroll
        for (int k = 0; k < W; ++k)
            if ((uint32_t)k + 1 == n_here) result = H[k];
    }
}

// (two waves per SIMD: past 256 registers hipcc parks values in the accumulator file and the kernel runs alone on its SIMD, at half the rate)
template <int W, bool kAffine, bool kLocal>
__global__ __launch_bounds__(kAlignWaves * 64, 2) void k_align_cross_long(AlignShortArgs args, int *boundary, uint32_t rows_cap, uint32_t queries_per_item) {
    __shared__ __attribute__((aligned(16))) char ltable[kClassLdsBytes];
    __shared__ __attribute__((aligned(16))) AlignLongLds wave_lds[kAlignWaves];
    __shared__ SummaryLds summary_lds;
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, lmisfit;
    {
        const uint32_t *src = (const uint32_t *)args.class_table;
        for (int i = threadIdx.x; i < (int)kClassLdsBytes / 4; i += blockDim.x) ((uint32_t *)ltable)[i] = src[i];
    }
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lmisfit = 0; }
    __syncthreads();
    const uint8_t *lclass_of = (const uint8_t *)ltable + 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    AlignLongLds &wl = wave_lds[wave];
    const Job &job = args.job;
    const uint8_t *a_data = (const uint8_t *)job.a.data, *b_data = (const uint8_t *)job.b.data;
    const uint64_t na = job.a.count, nb = job.b.count;
    uint64_t a_total, b_total;
    { uint32_t unused; align_extent(job.a.offsets, args.off64, na, a_total, unused); align_extent(job.b.offsets, args.off64, nb, b_total, unused); }
    const uint64_t waves_total = (uint64_t)gridDim.x * kAlignWaves, wave_id = (uint64_t)blockIdx.x * kAlignWaves + wave;
    int *bh = boundary + wave_id * (uint64_t)rows_cap * 64 * (kAffine ? 2 : 1) + lane;
    int *be = bh + (uint64_t)rows_cap * 64;
    const int open = args.open, ext = args.extend, open_minus_ext = open - ext;
    const size_t elem = job.out_elem64 ? 8 : 4;
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, misfit = 0;
    const uint64_t chunks = (nb + 63) / 64, qblocks = (na + queries_per_item - 1) / queries_per_item;
    const uint64_t items = chunks * qblocks;
    for (uint64_t item = wave_id; item < items; item += waves_total) {
        // (a wave that is ahead steps back -- common.hpp: fair_priority --: ACGT-1 K NW linear 12.8 -> 13.5 TCUPS, affine 6.08 -> 6.47, SW
        // 6.93 -> 7.31, ACGT-100 affine 5.25 -> 5.66. The same two lines cost the one-pass kernels: k_align_cross_wide 10.4 -> 6.4 on
        // ACGT-100 NW linear, k_cross_short 2.39 -> 1.81 on words -- their items are too short for a priority to mean anything.)
        fair_priority((item - wave_id) / waves_total, (items - wave_id + waves_total - 1) / waves_total);
        const uint64_t chunk = item / qblocks, qb = item - chunk * qblocks;
        const uint64_t q_first = qb * queries_per_item, q_last = q_first + queries_per_item < na ? q_first + queries_per_item : na;
        const uint32_t q_count = (uint32_t)(q_last - q_first);
        const uint64_t cand = chunk * 64 + (uint64_t)lane;
        const bool have = cand < nb;
        uint64_t b0 = 0;
        uint32_t lb = 0;
        if (have) align_extent(job.b.offsets, args.off64, cand, b0, lb);
        const bool fits = have && lb <= 0x00FFFFFFu;
        if (have && !fits) misfit |= 1;
        const uint32_t n = fits ? lb : 0u;
        const uint32_t n_max = wave_max_u32(n);
        const uint32_t passes = (n_max + W - 1) / W;
        // -- the classes my candidate uses, over all of its columns
        uint32_t used = 0;
        for (uint32_t p = 0; p < passes; ++p) {
            uint32_t bw[W / 4], bcls[W / 4];
            align_fetch<W / 4>(b_data, b0 + (uint64_t)p * W, b_total, bw);
            align_classes<W / 4>(lclass_of, bw, n_max - p * W, bcls);
#pragma unroll
            for (int g = 0; g < W / 4; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p * W + (uint32_t)(4 * g + u) < n) used |= 1u << ((bcls[g] >> (8 * u)) & 31u);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) used |= (uint32_t)__shfl_xor((int)used, off);
        const bool compact = __popc(used) <= 8;
        if (!compact) misfit |= 2;   // (bit 1: the alphabet is too rich for the compacting kernels -- CallSummary::violation)
        wave_lds_fence();                                  // the previous item's readers are done with the tables
        if (lane < 32) {
            wl.cid[lane] = (uint8_t)__popc(used & ((1u << lane) - 1u));
            uint32_t rest = used, lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t cls = rest ? (uint32_t)__builtin_ctz(rest) : 0u;
                const uint32_t cost = rest ? (uint32_t)(uint8_t)ltable[lane * 32 + cls] : 0u;
                rest &= rest - 1u;
                if (j < 4) lo |= cost << (8 * j); else hi |= cost << (8 * (j - 4));
            }
            *(uint2 *)wl.ctab[lane] = make_uint2(lo, hi);
        }
        wave_lds_fence();
        unsigned long long sum_m = 0;
        uint32_t item_maxa = 0;
        for (uint32_t q = 0; q < q_count; ++q) {
            uint64_t qa0 = 0;
            uint32_t qlen = 0;
            align_extent(job.a.offsets, args.off64, q_first + q, qa0, qlen);
            sum_m += qlen;
            item_maxa = qlen > item_maxa ? qlen : item_maxa;
            if (qlen > kAlignLongRows || qlen + 4 > rows_cap) { misfit |= 1; continue; }
            if (!compact) continue;
            // -- the query into LDS as class bytes, 1 KB per round
            wave_lds_fence();
            for (uint32_t at = 16u * (uint32_t)lane; at < qlen; at += 1024u) {
                ByteWindow win;
                win.init(a_data, qa0 + at, a_total);
                uint32_t staged[4], scls[4];
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) staged[w4] = win.fetch4(4 * w4);
                align_classes<4>(lclass_of, staged, 16, scls);
                *(uint4 *)(wl.qcls + at) = make_uint4(scls[0], scls[1], scls[2], scls[3]);
            }
            wave_lds_fence();
            int result = 0, best = 0;
            for (uint32_t p = 0; p < passes; ++p) {
                const uint32_t c0 = p * W;
                // -- selectors of this pass's columns: the compact ids of my candidate's symbols (local: nothing right of the string)
                uint32_t sel[W / 4];
                {
                    uint32_t bw[W / 4], bcls[W / 4];
                    align_fetch<W / 4>(b_data, b0 + c0, b_total, bw);
                    align_classes<W / 4>(lclass_of, bw, n_max - c0, bcls);
                    const uint32_t n_rel = n > c0 ? n - c0 : 0u;
#pragma unroll
                    for (int g = 0; g < W / 4; ++g) {
                        const uint32_t c = bcls[g];
                        uint32_t ids = (uint32_t)wl.cid[c & 31u] | ((uint32_t)wl.cid[(c >> 8) & 31u] << 8) | ((uint32_t)wl.cid[(c >> 16) & 31u] << 16) | ((uint32_t)wl.cid[(c >> 24) & 31u] << 24);
                        ids &= 0x07070707u;
                        if constexpr (kLocal) {
                            const uint32_t beyond = n_rel >= (uint32_t)(4 * g + 4) ? 0u : (n_rel <= (uint32_t)(4 * g) ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (n_rel - 4 * g)));
                            ids = (ids & ~beyond) | (0x0C0C0C0Cu & beyond);
                        }
                        sel[g] = ids;
                    }
                }
                const uint32_t cols_here = n_max - c0 < (uint32_t)W ? n_max - c0 : (uint32_t)W;
                const uint32_t n_here = (n > c0 && n <= c0 + W) ? n - c0 : 0u;     // my string ends in this pass: capture
                if (cols_here == (uint32_t)W)
                    align_pass<W, kAffine, kLocal, true>(wl.qcls, qlen, sel, n_here, cols_here, (const char *)&wl.ctab[0][0], open, ext, p == 0, p + 1 < passes, bh, be, result, best);
                else
                    align_pass<W, kAffine, kLocal, false>(wl.qcls, qlen, sel, n_here, cols_here, (const char *)&wl.ctab[0][0], open, ext, p == 0, p + 1 < passes, bh, be, result, best);
            }
            if (fits) {
                int score;
                if (!n || !qlen) score = align_trivial(qlen, lb, kLocal, open, ext);
                else if constexpr (kLocal) score = kAffine ? best : align_local_linear_best(best, open);
                else if constexpr (kAffine) score = result + (int)(qlen + n) * ext - open_minus_ext;
                else score = result + (int)(qlen + n) * ext;
                char *dst = job.out + (q_first + q) * job.row_stride + cand * elem;
                store_out(dst, job.out_elem64 != 0, (int64_t)score);
            }
        }
        if (have) {
            cells += sum_m * (unsigned long long)lb;
            maxb = lb > maxb ? lb : maxb;
            if (qb == 0) syms += lb;
            if (fits) shorts += q_count;
        }
        if (lane == 0) {
            maxa = item_maxa > maxa ? item_maxa : maxa;
            if (chunk == 0) syms += sum_m;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cells += __shfl_xor(cells, off);
        syms += __shfl_xor(syms, off);
        shorts += __shfl_xor(shorts, off);
        misfit |= __shfl_xor(misfit, off);
        const uint32_t oa = __shfl_xor(maxa, off), ob = __shfl_xor(maxb, off);
        maxa = oa > maxa ? oa : maxa;
        maxb = ob > maxb ? ob : maxb;
    }
    if (lane == 0) {
        atomicAdd(&lcells, cells);
        atomicAdd(&lsyms, syms);
        atomicAdd(&lshorts, shorts);
        atomicMax(&lmaxa, maxa);
        atomicMax(&lmaxb, maxb);
        atomicOr(&lmisfit, misfit);
    }
    __syncthreads();
    report_call_summary(PlanPartial{lcells, lsyms, lmaxa, lmaxb, lshorts, lmisfit}, args.partials, args.done_counter, args.summary, summary_lds);
}

// waves (= boundary areas) a launch of the multi-pass kernel gets: two four-wave workgroups per CU (a 128-column row, or 64
// columns of H and F, leave two waves per SIMD), never more than it has work items -- and never more than kAlignBoundaryBytes of
// boundary columns (an area is (longest_rows + 8) x 64 ints, twice that for Gotoh's E: queries of 4096 symbols on the 2048 waves of
// 256 compute units are 2.2 GB, 4.3 for Gotoh: such calls go to the column-profile kernel instead, align_long_fits below)
constexpr uint64_t kAlignBoundaryBytes = 640ull << 20;
static uint64_t align_long_area(uint32_t longest_rows, bool affine) {   // per workgroup
    return (uint64_t)(longest_rows + 8) * 64 * sizeof(int32_t) * (affine ? 2 : 1) * kAlignWaves;
}
static uint64_t align_long_budget(bool *hooked = nullptr) {
    const char *e = test_hook("STRINGWARS_AMD_ALIGN_BOUNDARY_MB");   // (test library, read per launch: a test lowers it to meet the cap on small inputs)
    if (hooked) *hooked = e != nullptr;
    return e ? (uint64_t)atoll(e) << 20 : kAlignBoundaryBytes;
}
uint32_t align_long_waves(const Scope *scope, uint64_t items, uint32_t longest_rows, bool affine) {
    uint64_t blocks = std::min<uint64_t>((items + kAlignWaves - 1) / kAlignWaves, (uint64_t)scope->compute_units * 2);
    blocks = std::min<uint64_t>(blocks, align_long_budget() / align_long_area(longest_rows, affine));
    return (uint32_t)(blocks ? blocks : 1) * kAlignWaves;
}
// Does a launch with every wave it could use stay inside the budget? If not the call takes the column-profile kernel (api.hip), whose
// boundary rings are a few MB: a cross-product call does not allocate gigabytes of scratch (round 4 let this one grow to 4 GB and
// gave a launch past that fewer waves; with the test hook set that is still what happens, so that the cap itself stays tested).
bool align_long_fits(const Scope *scope, uint64_t items, uint32_t longest_rows, bool affine) {
    bool hooked = false;
    const uint64_t budget = align_long_budget(&hooked);
    if (hooked) return true;
    const uint64_t blocks = std::min<uint64_t>((items + kAlignWaves - 1) / kAlignWaves, (uint64_t)scope->compute_units * 2);
    return blocks * align_long_area(longest_rows, affine) <= budget;
}

// Queries per work item (one item = those queries x 64 candidates, walked by one wave from start to end). An item of 16 queries of
// 1 K symbols is 10^9 cells -- a sixth of a second -- and the items go round-robin over the resident waves, so a cross-product of
// 1774 x 1774 such strings (3108 items on 2048 waves) ran its second round half empty: 9.6 TCUPS where 2048 x 2048 (exactly two
// rounds) ran 12.8. What an item shares between its queries is only the scan of its candidates' classes (one read of 64 strings),
// so the items are cut finer until there are sixteen rounds of them, down to one query each.
uint32_t align_long_queries(const Scope *scope, uint64_t queries, uint64_t candidates) {
    const uint64_t chunks = (candidates + 63) / 64, resident = (uint64_t)scope->compute_units * 2 * kAlignWaves;
    uint32_t per_item = kAlignQueries;
    while (per_item > 1 && chunks * ((queries + per_item - 1) / per_item) < 16 * resident) per_item >>= 1;
    return per_item;
}

void launch_align_long(Scope *scope, const KernelArgs &k, uint32_t longest_rows) {
    AlignShortArgs args{};
    args.job = k.job; args.off64 = k.off64;
    args.open = k.scoring.open; args.extend = k.scoring.extend;
    args.class_table = k.scoring.class_table;
    args.partials = scope->plan_partials; args.done_counter = scope->done_counter; args.summary = scope->summary_target();
    const Job &job = k.job;
    const uint32_t per_item = align_long_queries(scope, job.a.count, job.b.count);
    const uint64_t items = ((job.b.count + 63) / 64) * ((job.a.count + per_item - 1) / per_item);
    const uint32_t waves = align_long_waves(scope, items, longest_rows, k.affine != 0);
    const dim3 grid(waves / kAlignWaves), block(kAlignWaves * 64);
    const bool affine = k.affine != 0, local = k.local != 0;
    const uint32_t rows_cap = longest_rows + 8;
    const char *name = affine ? (local ? "align_long_affine_local" : "align_long_affine") : (local ? "align_long_local" : "align_long");
    StampGuard guard(scope, name);
    int *boundary = (int *)k.boundary;
    // pass widths by what fits 256 registers at two waves per SIMD without spilling: a row of 128 for linear global alignment; the local
    // forms carry the running maximum and their floor, Gotoh a second row (128 columns of local linear spilled 972 bytes per lane to scratch)
    if (!affine && !local) hipLaunchKernelGGL((k_align_cross_long<128, false, false>), grid, block, 0, scope->stream, args, boundary, rows_cap, per_item);
    else if (!affine) hipLaunchKernelGGL((k_align_cross_long<64, false, true>), grid, block, 0, scope->stream, args, boundary, rows_cap, per_item);
    else if (!local) hipLaunchKernelGGL((k_align_cross_long<64, true, false>), grid, block, 0, scope->stream, args, boundary, rows_cap, per_item);
    else hipLaunchKernelGGL((k_align_cross_long<32, true, true>), grid, block, 0, scope->stream, args, boundary, rows_cap, per_item);
    SWH_HIP_CHECK(hipGetLastError());
}

template <int W, int PQ>
static void launch_align_short_model(Scope *scope, const AlignShortArgs &args, bool affine, bool local, uint32_t blocks, const char *&name) {
    const dim3 grid(blocks), block(kAlignWaves * 64);
    if (!affine && !local) { name = W == 16 ? "align_short_w16" : (W == 32 ? "align_short_w32" : "align_short_w64"); }
    else if (affine && !local) { name = W == 16 ? "align_short_affine_w16" : (W == 32 ? "align_short_affine_w32" : "align_short_affine_w64"); }
    else if (!affine) { name = W == 16 ? "align_short_local_w16" : (W == 32 ? "align_short_local_w32" : "align_short_local_w64"); }
    else { name = W == 16 ? "align_short_affine_local_w16" : (W == 32 ? "align_short_affine_local_w32" : "align_short_affine_local_w64"); }
    StampGuard guard(scope, name);
    if (!affine && !local) hipLaunchKernelGGL((k_align_short<W, PQ, false, false>), grid, block, 0, scope->stream, args);
    else if (affine && !local) hipLaunchKernelGGL((k_align_short<W, PQ, true, false>), grid, block, 0, scope->stream, args);
    else if (!affine) hipLaunchKernelGGL((k_align_short<W, PQ, false, true>), grid, block, 0, scope->stream, args);
    else hipLaunchKernelGGL((k_align_short<W, PQ, true, true>), grid, block, 0, scope->stream, args);
    SWH_HIP_CHECK(hipGetLastError());
}

void launch_align_short(Scope *scope, const KernelArgs &k, uint32_t longest, bool wide) {
    AlignShortArgs args{};
    args.job = k.job; args.off64 = k.off64;
    args.open = k.scoring.open; args.extend = k.scoring.extend;
    args.class_table = k.scoring.class_table;
    args.partials = scope->plan_partials; args.done_counter = scope->done_counter; args.summary = scope->summary_target();
    const Job &job = k.job;
    const uint64_t items = job.cross ? ((job.b.count + 63) / 64) * ((job.a.count + kAlignQueries - 1) / kAlignQueries) : (job.pairs + 63) / 64;
    const uint64_t blocks64 = (items + kAlignWaves - 1) / kAlignWaves;
    uint32_t max_blocks = (uint32_t)scope->compute_units * 8;
    if (max_blocks > (uint32_t)kMaxPartials) max_blocks = kMaxPartials;
    const uint32_t blocks = blocks64 > max_blocks ? max_blocks : (uint32_t)(blocks64 ? blocks64 : 1);
    const bool affine = k.affine != 0, local = k.local != 0;
    const bool few = k.scoring.classes && k.scoring.classes <= 8;   // every class in the first two dwords of a cost row
    const char *name = nullptr;
    if (wide) {   // (api.hip picks this only for cross-products with linear gaps)
        const dim3 grid(blocks), block(kAlignWaves * 64);
        name = longest <= 64 ? (local ? "align_wide_local_w64" : "align_wide_w64") : (local ? "align_wide_local_w128" : "align_wide_w128");
        StampGuard guard(scope, name);
        if (longest <= 64) {
            if (local) hipLaunchKernelGGL((k_align_cross_wide<64, true>), grid, block, 0, scope->stream, args);
            else hipLaunchKernelGGL((k_align_cross_wide<64, false>), grid, block, 0, scope->stream, args);
        } else {
            if (local) hipLaunchKernelGGL((k_align_cross_wide<128, true>), grid, block, 0, scope->stream, args);
            else hipLaunchKernelGGL((k_align_cross_wide<128, false>), grid, block, 0, scope->stream, args);
        }
        SWH_HIP_CHECK(hipGetLastError());
        return;
    }
    if (longest <= 16) {
        if (few) launch_align_short_model<16, 1>(scope, args, affine, local, blocks, name);
        else launch_align_short_model<16, 4>(scope, args, affine, local, blocks, name);
    } else if (longest <= 32) {
        if (few) launch_align_short_model<32, 1>(scope, args, affine, local, blocks, name);
        else launch_align_short_model<32, 4>(scope, args, affine, local, blocks, name);
    } else {
        // up to 64 bytes (a register row of 64 cells, 64 selector registers for 32 classes: two waves per SIMD, Gotoh one): the tail of
        // multilingual word tokens, which used to send the whole batch to the planned path -- 2048 x 2048 such words 0.19 -> see DESIGN 4.2c
        if (few) launch_align_short_model<64, 1>(scope, args, affine, local, blocks, name);
        else launch_align_short_model<64, 4>(scope, args, affine, local, blocks, name);
    }
}

}  // namespace swh
