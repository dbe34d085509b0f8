// cross.hip -- the reference's own call shape, `engine.compute_into(queries, candidates, &mut matrix)` (bench.rs:478-486),
// for word-sized strings: dense queries x candidates Levenshtein with the query's match table built ONCE per row block.
//
// The pairwise kernels treat a cross-product as pairs p = (p / nb, p % nb) and pay, per pair, what a pair costs: two
// extent look-ups, two string loads, a match-table build and clear (a third of k_direct_short's instructions). In a
// cross-product a query meets every candidate, so here a wave keeps 64 CANDIDATES in registers (one per lane: length and
// up to 32 bytes) and walks a block of queries over them. The query is the bit-vector side (rows): its table -- Eq(c) =
// Lo[c & 15] & Hi[c >> 4], 32 dwords -- is the same for all lanes, so it lives once per wave in LDS (128 bytes, built by
// 32 lanes with two ds_or, read as broadcasts) instead of once per lane; the table of the next query is built while the
// current one's columns are walked (two buffers), and an item's 16 queries are staged into LDS with one batch of loads.
// (Dealing the candidates to the waves by length, so that the 64 texts of a wave end together, was measured and dropped:
// the matrix is row-major in candidate order, and 64 scattered 8-byte stores per query cost more than the columns saved.)
//
// Exact for queries and candidates of up to 32 bytes; a longer string raises the call summary's `violation` flag and the
// host redoes the call on the general path (api.hip), exactly as for the other plan-free kernels.
#include "common.hpp"
#include "bp_window.hpp"

namespace swh {

constexpr uint32_t kCrossMax = 32;          // longest query / candidate in bytes
constexpr int kCrossQueries = 16;           // queries per work item (2048 x 2048 words: 4096 items, one per wave slot)
constexpr int kCrossWaves = 4;
constexpr uint32_t kCrossCompareRows = 12;  // code points: queries up to here are matched by comparison (k_cross_short_cp), longer ones through the group tables

struct CrossArgs {
    Job job;                 // cross = 1; out is the matrix
    uint32_t off64;
    PlanPartial *partials;
    uint32_t *done_counter;
    CallSummary *summary;
};

template <typename Off>
__device__ __forceinline__ void tape_extent(const void *offsets, uint64_t i, uint64_t &start, uint32_t &len) {
    const Off *o = (const Off *)offsets;
    const Off x0 = o[i], x1 = o[i + 1];
    start = (uint64_t)x0; len = (uint32_t)(x1 - x0);
}

// Between LDS operations of ONE wave only program order has to be kept: a wave's DS instructions are queued and executed
// in issue order, so a table update (ds_or) followed by a lookup (ds_read) of another lane's word needs no wait in
// between -- the compiler just must not move the accesses across each other (it reasons per thread).
__device__ __forceinline__ void lds_program_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

struct CrossWaveLds {
    uint32_t table[2][32];                    // Lo[16] | Hi[16] of the current query and of the next one (built ahead)
    uint32_t qlen[kCrossQueries];
    uint8_t qbytes[kCrossQueries][kCrossMax]; // the item's queries, staged once (one batch of loads instead of one per query)
};

template <typename Off>
__global__ __launch_bounds__(kCrossWaves * 64) void k_cross_short(CrossArgs args) {
    __shared__ CrossWaveLds wave_lds[kCrossWaves];
    __shared__ SummaryLds summary_lds;
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, lmisfit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    CrossWaveLds &wl = wave_lds[wave];
    if (lane < 32) { wl.table[0][lane] = 0; wl.table[1][lane] = 0; }
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lmisfit = 0; }
    __syncthreads();
    const Job &job = args.job;
    const uint64_t na = job.a.count, nb = job.b.count;
    const uint8_t *a_data = (const uint8_t *)job.a.data, *b_data = (const uint8_t *)job.b.data;
    const uint64_t b_total = (uint64_t)((const Off *)job.b.offsets)[nb];
    const uint64_t chunks = (nb + 63) / 64, qblocks = (na + kCrossQueries - 1) / kCrossQueries;
    const uint64_t items = chunks * qblocks;
    const uint64_t waves_total = (uint64_t)gridDim.x * kCrossWaves, wave_id = (uint64_t)blockIdx.x * kCrossWaves + wave;
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, misfit = 0;
    const size_t elem = job.out_elem64 ? 8 : 4;

    for (uint64_t item = wave_id; item < items; item += waves_total) {
        // consecutive waves share a chunk of candidates and take neighbouring query blocks (the candidates' cache lines stay warm)
        const uint64_t chunk = item / qblocks, qb = item - chunk * qblocks;
        const uint64_t q_first = qb * kCrossQueries, q_last = q_first + kCrossQueries < na ? q_first + kCrossQueries : na;
        const uint32_t q_count = (uint32_t)(q_last - q_first);
        // ---- every load of the item is requested here: my candidate (extent, bytes) and the item's queries ---------------
        const uint64_t slot = chunk * 64 + (uint64_t)lane;
        const bool have = slot < nb;
        uint64_t cand = 0, b0 = 0;
        uint32_t n = 0;
        if (have) {
            cand = slot;
            tape_extent<Off>(job.b.offsets, cand, b0, n);
        }
        // queries: lane l stages bytes 8 (l % 4) .. 8 (l % 4) + 7 of query l / 4 (16 queries x 32 bytes = 64 lanes x 8 bytes)
        const uint32_t ql = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
        uint64_t qa0 = 0;
        uint32_t qm = 0;
        if (ql < q_count) tape_extent<Off>(job.a.offsets, q_first + ql, qa0, qm);
        uint32_t staged[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t at = part * 8 + (uint32_t)t;
            staged[t] = (at < qm && qm <= kCrossMax) ? a_data[qa0 + at] : 0u;
        }
        const bool fits = have && n <= kCrossMax;
        if (have && !fits) misfit = 1;
        uint32_t tw[8];
        {
            ByteWindow txt;
            txt.init(b_data, b0, b_total);
            if (b_total >= 16) {
                uint32_t half[4];
                int moved = txt.fetch16_raw(0, half);
                txt.fix16(0, moved, half);
#pragma unroll
                for (int q = 0; q < 4; ++q) tw[q] = half[q];
                moved = txt.fetch16_raw(16, half);
                txt.fix16(16, moved, half);
#pragma unroll
                for (int q = 0; q < 4; ++q) tw[4 + q] = half[q];
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) tw[q] = txt.fetch4(q * 4);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) wl.qbytes[ql][part * 8 + t] = (uint8_t)staged[t];
        if (part == 0) wl.qlen[ql] = ql < q_count ? qm : 0u;
        lds_program_order();
        const uint32_t n_live = fits ? n : 0;
        const uint32_t n_max = wave_max_u32(n_live);
        unsigned long long sum_m = 0;   // over the item's queries, for the work units
        uint32_t item_maxa = 0;
        // ---- the item's queries: the table of query q + 1 is built while query q's columns are walked ----------------------
        auto build = [&](uint32_t q, uint32_t *table) -> uint32_t {
            const uint32_t m = q < q_count ? wl.qlen[q] : 0u;
            if ((uint32_t)lane < m && m <= kCrossMax) {
                const uint32_t byte = wl.qbytes[q][lane];
                atomicOr(&table[byte & 15u], 1u << lane);
                atomicOr(&table[16 + (byte >> 4)], 1u << lane);
            }
            return m;
        };
        uint32_t m_next = build(0, wl.table[0]);
        for (uint32_t q = 0; q < q_count; ++q) {
            uint32_t *table = wl.table[q & 1];
            const uint32_t m = m_next;
            lds_program_order();                                   // table q is complete; table q - 1 (the other buffer) has been cleared
            m_next = build(q + 1, wl.table[(q + 1) & 1]);
            sum_m += m;
            item_maxa = m > item_maxa ? m : item_maxa;
            if (m > kCrossMax) misfit = 1;
            uint32_t pv = 0xFFFFFFFFu, mv = 0;
            if (m <= kCrossMax) {
#pragma unroll
                for (int w4 = 0; w4 < 8; ++w4) {
                    if ((uint32_t)w4 * 4 >= n_max) break;
                    const uint32_t w = tw[w4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // Lo[c & 15] and Hi[c >> 4] as byte offsets: ((w >> 8u) & 15) << 2 and ((w >> (8u + 4)) & 15) << 2
                        const uint32_t lo_at = (u == 0 ? (w << 2) : (w >> (8 * u - 2))) & 0x3Cu;
                        const uint32_t hi_at = (w >> (8 * u + 2)) & 0x3Cu;
                        const uint32_t eq = *(const uint32_t *)((const char *)table + lo_at) & *(const uint32_t *)((const char *)table + 64 + hi_at);
                        if ((uint32_t)(w4 * 4 + u) < n_live) {
                            const uint32_t xv = eq | mv;
                            const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
                            uint32_t ph = mv | ~(xh | pv);
                            const uint32_t mh = pv & xh;
                            ph = (ph << 1) | 1u;
                            pv = (mh << 1) | ~(xv | ph);
                            mv = ph & xv;
                        }
                    }
                }
                if (fits) {
                    const uint32_t mask = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);
                    const uint32_t d = n + __popc(pv & mask) - __popc(mv & mask);
                    char *dst = job.out + (q_first + q) * job.row_stride + cand * elem;
                    store_out(dst, job.out_elem64 != 0, (int64_t)d);
                }
            }
            lds_program_order();   // every lane has read table q
            if (lane < 32) table[lane] = 0;
        }
        lds_program_order();
        if (lane < 32) wl.table[q_count & 1][lane] = 0;   // (the look-ahead build of a query past the block wrote nothing, but keep both clean)
        lds_program_order();
        if (have) {
            cells += sum_m * (unsigned long long)n;
            maxb = n > maxb ? n : maxb;
            if (qb == 0) syms += n;                                   // every candidate once ...
            if (fits) shorts += q_count;
        }
        if (lane == 0) {
            maxa = item_maxa > maxa ? item_maxa : maxa;
            if (chunk == 0) syms += sum_m;                             // ... and every query once (bench.rs:216-224)
        }
    }
    // ---- summary -----------------------------------------------------------------------------------------------------------
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

// ---- the same for CODE POINTS (decoded UTF-8 tapes: 32-bit symbols, 64-bit offsets): `LevenshteinDistancesUtf8` on word-sized tokens,
// the reference's XLSum words column (similarities/README.md:38-63). The query's match table is the code-point model of bp_item.hpp --
// seven groups of three bits, eight entries each, Eq(c) = the AND of the seven entries c's groups select -- 56 dwords per query, shared
// by the wave like the byte kernel's 32; a lane keeps its candidate's up to 32 symbols in registers.
struct CrossCpWaveLds {
    uint32_t table[2][56];
    uint32_t qlen[kCrossQueries];
    uint32_t qsyms[kCrossQueries][kCrossMax];
};

__global__ __launch_bounds__(kCrossWaves * 64) void k_cross_short_cp(CrossArgs args) {
    __shared__ CrossCpWaveLds wave_lds[kCrossWaves];
    __shared__ SummaryLds summary_lds;
    __shared__ unsigned long long lcells, lsyms;
    __shared__ uint32_t lmaxa, lmaxb, lshorts, lmisfit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    CrossCpWaveLds &wl = wave_lds[wave];
    if (lane < 56) { wl.table[0][lane] = 0; wl.table[1][lane] = 0; }
    if (threadIdx.x == 0) { lcells = 0; lsyms = 0; lmaxa = 0; lmaxb = 0; lshorts = 0; lmisfit = 0; }
    __syncthreads();
    const Job &job = args.job;
    const uint64_t na = job.a.count, nb = job.b.count;
    const uint32_t *a_data = (const uint32_t *)job.a.data, *b_data = (const uint32_t *)job.b.data;
    const uint64_t chunks = (nb + 63) / 64, qblocks = (na + kCrossQueries - 1) / kCrossQueries;
    const uint64_t items = chunks * qblocks;
    const uint64_t waves_total = (uint64_t)gridDim.x * kCrossWaves, wave_id = (uint64_t)blockIdx.x * kCrossWaves + wave;
    unsigned long long cells = 0, syms = 0;
    uint32_t maxa = 0, maxb = 0, shorts = 0, misfit = 0;
    const size_t elem = job.out_elem64 ? 8 : 4;
    for (uint64_t item = wave_id; item < items; item += waves_total) {
        const uint64_t chunk = item / qblocks, qb = item - chunk * qblocks;
        const uint64_t q_first = qb * kCrossQueries, q_last = q_first + kCrossQueries < na ? q_first + kCrossQueries : na;
        const uint32_t q_count = (uint32_t)(q_last - q_first);
        const uint64_t slot = chunk * 64 + (uint64_t)lane;
        const bool have = slot < nb;
        uint64_t cand = 0, b0 = 0;
        uint32_t n = 0;
        if (have) {
            cand = slot;
            tape_extent<uint64_t>(job.b.offsets, cand, b0, n);
        }
        // queries: lane l stages symbols 8 (l % 4) .. 8 (l % 4) + 7 of query l / 4
        const uint32_t ql = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
        uint64_t qa0 = 0;
        uint32_t qm = 0;
        if (ql < q_count) tape_extent<uint64_t>(job.a.offsets, q_first + ql, qa0, qm);
        uint32_t staged[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t at = part * 8 + (uint32_t)t;
            staged[t] = (at < qm && qm <= kCrossMax) ? a_data[qa0 + at] : 0u;
        }
        const bool fits = have && n <= kCrossMax;
        if (have && !fits) misfit = 1;
        uint32_t tw[kCrossMax];
#pragma unroll
        for (int t = 0; t < (int)kCrossMax; ++t) tw[t] = (fits && (uint32_t)t < n) ? b_data[b0 + (uint32_t)t] : 0u;
#pragma unroll
        for (int t = 0; t < 8; ++t) wl.qsyms[ql][part * 8 + t] = staged[t];
        if (part == 0) wl.qlen[ql] = ql < q_count ? qm : 0u;
        lds_program_order();
        const uint32_t n_live = fits ? n : 0;
        const uint32_t n_max = wave_max_u32(n_live);
        unsigned long long sum_m = 0;
        uint32_t item_maxa = 0;
        // ---- word-sized queries (every query of the item at most kCrossCompareRows symbols: tokens of ~5 code points): NO match table.
        // The query is the same for all 64 lanes, so its symbols are SCALARS: row i's symbol is compared with the candidates' columns
        // (`v_cmp_eq_u32 vcc, s, v` + `v_addc_co_u32 eq, eq, eq, vcc`: eq = 2 eq + match, rows from the last to the first) -- two
        // instructions per cell of the match matrix, no LDS: against the group tables' seven ds_or per row and, per column, seven
        // look-ups, seven address computations and six ANDs, that is 2 m + 11 instructions per column (m = 5: 21) instead of 31 + 7 LDS.
        uint32_t longest_query = 0;
        for (uint32_t q = 0; q < q_count; ++q) { const uint32_t m = wl.qlen[q]; longest_query = m > longest_query ? m : longest_query; }
        longest_query = (uint32_t)__builtin_amdgcn_readfirstlane((int)longest_query);
        if (longest_query <= kCrossCompareRows) {
            for (uint32_t q = 0; q < q_count; ++q) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)wl.qlen[q]);
                sum_m += m;
                item_maxa = m > item_maxa ? m : item_maxa;
                uint32_t eq[kCrossMax];
#pragma unroll
                for (int t = 0; t < (int)kCrossMax; ++t) eq[t] = 0;
                for (int i = (int)m - 1; i >= 0; --i) {
                    const uint32_t sym = (uint32_t)__builtin_amdgcn_readfirstlane((int)wl.qsyms[q][i]);
#pragma unroll
                    for (int t4 = 0; t4 < (int)kCrossMax; t4 += 4) {
                        if ((uint32_t)t4 >= n_max) break;
                        // eq = 2 eq + (column symbol == row symbol): the comparison's lane mask is the carry-in of the addition
                        // (hipcc builds it from v_cmp + v_cndmask + v_lshl_or and a wait state)
                        asm("v_cmp_eq_u32 vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                            "v_cmp_eq_u32 vcc, %4, %6\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
                            "v_cmp_eq_u32 vcc, %4, %7\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc\n\t"
                            "v_cmp_eq_u32 vcc, %4, %8\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"
                            : "+v"(eq[t4]), "+v"(eq[t4 + 1]), "+v"(eq[t4 + 2]), "+v"(eq[t4 + 3])
                            : "s"(sym), "v"(tw[t4]), "v"(tw[t4 + 1]), "v"(tw[t4 + 2]), "v"(tw[t4 + 3])
                            : "vcc");
                    }
                }
                uint32_t pv = 0xFFFFFFFFu, mv = 0;
#pragma unroll
                for (int t = 0; t < (int)kCrossMax; ++t) {
                    if ((uint32_t)t >= n_max) break;
                    if ((uint32_t)t < n_live) {
                        const uint32_t e = eq[t];
                        const uint32_t xv = e | mv;
                        const uint32_t xh = (((e & pv) + pv) ^ pv) | e;
                        uint32_t ph = mv | ~(xh | pv);
                        const uint32_t mh = pv & xh;
                        ph = (ph << 1) | 1u;
                        pv = (mh + mh) | ~(xv | ph);
                        mv = ph & xv;
                    }
                }
                if (fits) {
                    const uint32_t mask = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);
                    const uint32_t d = n + __popc(pv & mask) - __popc(mv & mask);
                    char *dst = job.out + (q_first + q) * job.row_stride + cand * elem;
                    store_out(dst, job.out_elem64 != 0, (int64_t)d);
                }
            }
        }
        auto build = [&](uint32_t q, uint32_t *table) -> uint32_t {
            const uint32_t m = q < q_count ? wl.qlen[q] : 0u;
            if ((uint32_t)lane < m && m <= kCrossMax) {
                const uint32_t c = wl.qsyms[q][lane];
#pragma unroll
                for (int g = 0; g < 7; ++g) atomicOr(&table[g * 8 + ((c >> (3 * g)) & 7u)], 1u << lane);
            }
            return m;
        };
        const uint32_t q_tabled = longest_query <= kCrossCompareRows ? 0u : q_count;   // (the longer queries' items: the group tables, as before)
        uint32_t m_next = q_tabled ? build(0, wl.table[0]) : 0u;
        for (uint32_t q = 0; q < q_tabled; ++q) {
            uint32_t *table = wl.table[q & 1];
            const uint32_t m = m_next;
            lds_program_order();
            m_next = build(q + 1, wl.table[(q + 1) & 1]);
            sum_m += m;
            item_maxa = m > item_maxa ? m : item_maxa;
            if (m > kCrossMax) misfit = 1;
            uint32_t pv = 0xFFFFFFFFu, mv = 0;
            if (m <= kCrossMax) {
#pragma unroll
                for (int t = 0; t < (int)kCrossMax; ++t) {
                    if ((uint32_t)t >= n_max) break;
                    const uint32_t c = tw[t];
                    uint32_t eq = table[c & 7u] & table[8 + ((c >> 3) & 7u)] & table[16 + ((c >> 6) & 7u)];
                    eq &= table[24 + ((c >> 9) & 7u)] & table[32 + ((c >> 12) & 7u)];
                    eq &= table[40 + ((c >> 15) & 7u)] & table[48 + ((c >> 18) & 7u)];
                    if ((uint32_t)t < n_live) {
                        const uint32_t xv = eq | mv;
                        const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
                        uint32_t ph = mv | ~(xh | pv);
                        const uint32_t mh = pv & xh;
                        ph = (ph << 1) | 1u;
                        pv = (mh << 1) | ~(xv | ph);
                        mv = ph & xv;
                    }
                }
                if (fits) {
                    const uint32_t mask = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);
                    const uint32_t d = n + __popc(pv & mask) - __popc(mv & mask);
                    char *dst = job.out + (q_first + q) * job.row_stride + cand * elem;
                    store_out(dst, job.out_elem64 != 0, (int64_t)d);
                }
            }
            lds_program_order();
            if (lane < 56) table[lane] = 0;
        }
        lds_program_order();
        if (q_tabled && lane < 56) wl.table[q_count & 1][lane] = 0;
        lds_program_order();
        if (have) {
            cells += sum_m * (unsigned long long)n;
            maxb = n > maxb ? n : maxb;
            if (qb == 0) syms += n;
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

void launch_cross_short(Scope *scope, const Job &job, uint32_t off64, uint32_t sym_bytes) {
    CrossArgs args{};
    args.job = job; args.off64 = off64;
    args.partials = scope->plan_partials; args.done_counter = scope->done_counter; args.summary = scope->summary_target();
    const uint64_t items = ((job.b.count + 63) / 64) * ((job.a.count + kCrossQueries - 1) / kCrossQueries);
    uint64_t blocks64 = (items + kCrossWaves - 1) / kCrossWaves;
    uint32_t max_blocks = (uint32_t)scope->compute_units * 8;
    if (max_blocks > (uint32_t)kMaxPartials) max_blocks = kMaxPartials;
    const uint32_t blocks = blocks64 > max_blocks ? max_blocks : (uint32_t)(blocks64 ? blocks64 : 1);
    StampGuard guard(scope, sym_bytes == 4 ? "cross_short_u32" : "cross_short");
    if (sym_bytes == 4) hipLaunchKernelGGL(k_cross_short_cp, dim3(blocks), dim3(kCrossWaves * 64), 0, scope->stream, args);   // (decoded tapes: 64-bit offsets)
    else if (off64) hipLaunchKernelGGL(k_cross_short<uint64_t>, dim3(blocks), dim3(kCrossWaves * 64), 0, scope->stream, args);
    else hipLaunchKernelGGL(k_cross_short<uint32_t>, dim3(blocks), dim3(kCrossWaves * 64), 0, scope->stream, args);
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
