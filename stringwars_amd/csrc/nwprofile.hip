// nwprofile.hip -- Needleman-Wunsch / Gotoh scores for substitution matrices with <= 32 symbol classes: the
// anti-diagonal wavefront of wavefront.hip with the substitution scores served from a COLUMN PROFILE in LDS.
//
// `NeedlemanWunschScores::new(&scope, &byte_to_class, &class_costs, open, extend)` (bench.rs:658-670, :985-997) and any
// 256x256 matrix that distinguishes <= 32 classes (20 amino acids + "other": config C4).
//
// wavefront.hip's class model loads the row symbol's 32-byte cost row once per step and picks each column's byte with
// `v_perm_b32`: 3 perms + 2 ors per four cells for 21 classes -- a third of that kernel's instructions. Here the picking
// is done ONCE PER PASS instead of once per row: one wave scores one pair, lane g owns a strip of `w` <= W consecutive
// columns, and before the rows stream by the wave writes, for every class c, the scores of (c, its columns) to LDS:
//
//     profile[c][plane p][lane]   one dword = the four columns 4p .. 4p+3 of that lane's strip      (classes x 64 W bytes)
//
// A step then reads its row's scores with w/4 conflict-free `ds_read_b32` (lane l -> bank l mod 32, rows are multiples
// of 256 B) and the cell is `v_add_u32_sdwa (sext byte)` + `v_max3_i32` (Gotoh: five instructions) and nothing else.
// LDS holds classes x 64 x W bytes per wave (21 classes, W = 12: 16 KB -> nine single-wave workgroups per CU), which
// caps the strip: a pair's columns run as ceil(cols / 64 W) passes, the right-edge column of a pass parked in global
// memory for the next one (one store and one load per row and wave). The strip width is chosen PER PAIR (w = the
// columns over passes x 64 lanes, rounded up to four): at most 255 phantom columns per pair whatever its length.
//
// The row string reaches the lanes through a ring in LDS: every 64 steps the wave loads the next 64 row symbols with one
// coalesced load, maps them to profile-row offsets (class x 64 W) and stores them as u16; lane g reads entry s - g. No
// per-lane byte extraction, no per-step class look-up.
//
// Pairs come from one global ticket, widest first (perm is sorted by columns class, then rows): a 4 KB x 4 KB pair is
// milliseconds of work, a fixed deal would end in a round that is a third full.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace swh {

namespace {

constexpr int kNegInfP = -0x20000000;
constexpr uint32_t kRingEntries = 256;            // row-symbol ring (u16 profile-row offsets)
constexpr uint32_t kRingBytes = 4 * 2 * kRingEntries;    // four copies, copy c shifted by c entries (aligned 8-byte reads for every lane)
// Boundary columns between passes travel through LDS rings, 64 rows per global transfer: a pass reads the previous pass's
// right edge two 64-step blocks ahead (one coalesced load per block, 128-row ring) and parks its own right edge in a 64-row
// ring that is flushed with one coalesced store per block. (Per-step loads by lane 0 / stores by lane 63 put a memory
// round trip -- stores count on vmcnt, too -- in front of every group of four steps: 500 cycles of work against 2000.)
constexpr uint32_t kBndReadRows = 128, kBndWriteRows = 64;
constexpr uint32_t kBndBytes = 2 * 4 * kBndReadRows + 2 * 4 * kBndWriteRows;   // H and E
constexpr uint32_t kScratchBytes = kRingBytes + kBndBytes;   // row ring + boundary rings; the class costs (1 KB) borrow it while a profile is built
static_assert(kScratchBytes >= 1024, "the class costs must fit the rings' space");

__device__ __forceinline__ int dpp_wave_shr1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

// Narrow strips (kNarrow, Gotoh): a two-operand 32-bit maximum issues at half the rate of a two-operand 16-bit one
// (profiles/r3/valu_ops_issue_rates_3.txt: v_max_i32 4.3 SIMD cycles per wave instruction, v_max_u16 2.5 with two waves per
// SIMD; v_max3_u16 8.3 -- no use), and two of the five operations of a Gotoh cell are such maxima. Everything a wave holds at one time -- 64 W columns,
// 64 rows -- lies within (64 W + 256) x `step_span` of each other (neighbouring cells differ by at most max |cost| + |open| +
// |extend| in the all-gaps-relative form), so the strips keep value - shift with ONE wave-wide `shift`, chosen so that the middle
// of the wave sits at 0x8000 and moved every 64 steps: all stored values are positive 16-bit numbers in 32-bit registers, the
// 32-bit operations (v_add_u32_sdwa, v_max3_i32) see them as they are, and the two plain maxima become v_max_u16. The
// boundary columns in global memory stay true 32-bit values: they are converted where the rings are filled / flushed.
// api.hip sets `Scoring::step_span`; launch_nwprofile() takes this path when (64 W + 256) x step_span <= 30 000.
constexpr int kCenter = 0x8000;
__device__ __forceinline__ int umax16(int a, int b) {
    int r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Four Gotoh cells of a strip (global alignment), the schedule written out -- see run_pass. Per cell, in issue order:
//   e = max(left, e)  |  F[k+1] = max(H[k+1], F[k+1])  |  tmp = max3(t[k], e, F[k])  |  t[k+2] = H[k+1] + score  |  H[k] = tmp + (open - extend)
// (H[k+1] is still the row above when the second and fourth read it). MAX is v_max_i32, or v_max_u16 on narrow strips.
#define SWH_NWP_SCORE_ADD(DST, BASE, WORD, BYTE) \
    "v_add_u32_sdwa " DST ", " BASE ", sext(" WORD ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" BYTE "\n"
#define SWH_NWP_CELL(MAX, LEFT, FNEXT, HNEXT, TCUR, FCUR, TNEXT, WORD, BYTE, HOUT)                                      \
    MAX " %[e], " LEFT ", %[e]\n" MAX " " FNEXT ", " HNEXT ", " FNEXT "\n"                                             \
        "v_max3_i32 %[tmp], " TCUR ", %[e], " FCUR "\n" SWH_NWP_SCORE_ADD(TNEXT, HNEXT, WORD, BYTE) "v_add_u32 " HOUT ", %[tmp], %[c]\n"
#define SWH_NWP_PLANE(MAX)                                                                                              \
    SWH_NWP_CELL(MAX, "%[left]", "%[f1]", "%[h1]", "%[t0]", "%[f0]", "%[t2]", "%[s0]", "2", "%[h0]")                     \
    SWH_NWP_CELL(MAX, "%[h0]", "%[f2]", "%[h2]", "%[t1]", "%[f1]", "%[t3]", "%[s0]", "3", "%[h1]")                       \
    SWH_NWP_CELL(MAX, "%[h1]", "%[f3]", "%[h3]", "%[t2]", "%[f2]", "%[t4]", "%[s1]", "0", "%[h2]")                       \
    SWH_NWP_CELL(MAX, "%[h2]", "%[f4]", "%[hn]", "%[t3]", "%[f3]", "%[t5]", "%[s1]", "1", "%[h3]")
// the strip's last four cells: no column to the right of the fourth
#define SWH_NWP_LAST_PLANE(MAX)                                                                                         \
    SWH_NWP_CELL(MAX, "%[left]", "%[f1]", "%[h1]", "%[t0]", "%[f0]", "%[t2]", "%[s0]", "2", "%[h0]")                     \
    SWH_NWP_CELL(MAX, "%[h0]", "%[f2]", "%[h2]", "%[t1]", "%[f1]", "%[t3]", "%[s0]", "3", "%[h1]")                       \
    MAX " %[e], %[h1], %[e]\n" MAX " %[f3], %[h3], %[f3]\n"                                                              \
        "v_max3_i32 %[tmp], %[t2], %[e], %[f2]\ns_nop 0\nv_add_u32 %[h2], %[tmp], %[c]\n" MAX " %[e], %[h2], %[e]\n"      \
        "s_nop 0\nv_max3_i32 %[tmp], %[t3], %[e], %[f3]\ns_nop 0\nv_add_u32 %[h3], %[tmp], %[c]\n"

// One pass of one pair: the strip is WE columns per lane (a multiple of four, compile-time here: the kernel switches on the
// pass's width), columns c0+1 .. c0+64 WE; `row_bytes` is the profile's row pitch (64 x the kernel's W).
// kRead: the pass takes its left edge from the previous pass's right edge (not the first pass); kWrite: it parks its own
// right edge for the next pass (not the last one).
// kLocal: Smith-Waterman (`SmithWatermanScores`, bench.rs:882-963) -- every cell floored at zero, the result is the maximum over
// all cells (`best`, carried from pass to pass by the caller). Round 6: the floor comes from UNSIGNED SATURATING subtraction
// (`v_sub_u32 ... clamp`, Farrar's trick on 32-bit lanes): the strips hold the true H >= 0 and, beside it, G = sat(H - |open|),
// which is both the `up` term of the cell below and the `left` term of the cell to the right -- one instruction per cell where
// max(., 0) and + open were two. A cell is then  t = Hdiag + sub ; H = max3(t, Gup, Gleft) ; G = sat(H - |open|)  -- max3 of a
// signed t with two non-negative terms needs no floor of its own. Gotoh: E and F are floored at zero as well (they only ever
// enter H through a maximum with 0): f = max(Gup, sat(F - |ext|)), e = max(Gleft, sat(e - |ext|)), H = max3(t, e, f).
// The running maximum takes the diagonal candidates t only, two per v_max3 (the best cell of a local alignment ends a
// substitution, never a gap). Phantom columns score 0: their t is the H of a real cell, nothing new.
template <int WE, bool kAffine, bool kRead, bool kWrite, bool kLocal, bool kNarrow>
__device__ __forceinline__ void run_pass(const KernelArgs &args, char *smem, uint32_t ring_at, const uint8_t *ctab, const uint8_t *cmap,
                                         const uint8_t *table_src, uint32_t cstride,
                                         uint32_t classes, uint32_t row_bytes, const uint8_t *col_data, const uint8_t *row_data,
                                         uint32_t rows, uint32_t cols, uint32_t c0, int32_t *bnd_h, int32_t *bnd_e, uint64_t p, int &best) {
    constexpr int kPlanes = WE / 4;
    const int lane = threadIdx.x;
    const int open = args.scoring.open, ext = args.scoring.extend;
    const int open_minus_ext = open - ext;
    const uint32_t mine = c0 + (uint32_t)lane * WE;      // 0-based index of my first column
    const uint32_t steps = (rows + 63 + 3) & ~3u;
    // ---- the column profile of this pass ----------------------------------------------------------------------------
    // (the class costs are only needed here: they are brought into the space the rings use afterwards -- all 32 x 32 of the register
    // model's table at once, a wide table's rows of kWideClasses bytes as many at a time as fit)
    {
        uint32_t ccls[WE];
#pragma unroll
        for (int k = 0; k < WE; ++k) {
            const uint32_t j = mine + k;
            const bool real = j < cols;
            const uint32_t sym = real ? (uint32_t)col_data[j] : 0u;
            ccls[k] = real ? (uint32_t)cmap[sym] : 0xFFu;
        }
        const uint32_t rows_at_once = kScratchBytes / cstride;   // 88 rows of 32 bytes (every class of the small table), 22 of 128
        for (uint32_t cfirst = 0; cfirst < classes; cfirst += rows_at_once) {
            const uint32_t cn = classes - cfirst < rows_at_once ? classes - cfirst : rows_at_once;
            {
                const uint32_t *src = (const uint32_t *)(table_src + cfirst * cstride);
                for (uint32_t i = (uint32_t)lane; i < cn * cstride / 4; i += 64) ((uint32_t *)(smem + ring_at))[i] = src[i];
            }
            wave_lds_fence();
            for (uint32_t c = cfirst; c < cfirst + cn; ++c) {
                const uint8_t *crow = ctab + (c - cfirst) * cstride;
#pragma unroll
                for (int pl = 0; pl < kPlanes; ++pl) {
                    uint32_t dw = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t cl = ccls[4 * pl + i];
                        // phantom columns score 0: harmless, never read back. Local alignment: the class table holds sub - open (what the lane
                        // kernels of alignshort.hip want); the profile takes the bias back -- its cells work on true scores
                        const uint32_t byte = cl != 0xFFu ? (kLocal ? (uint32_t)(uint8_t)((int)(int8_t)crow[cl & (cstride - 1u)] + open) : (uint32_t)crow[cl & (cstride - 1u)]) : 0u;
                        dw |= byte << (8 * i);
                    }
                    ((uint32_t *)(smem + c * row_bytes))[pl * 64 + lane] = dw;
                }
            }
            wave_lds_fence();   // the next rows of the table (or the rings) overwrite these
        }
    }
    wave_lds_fence();   // everybody is done with the class costs: the rings take their place
    // ---- the row ring ---------------------------------------------------------------------------------------------------
    // Entry e = profile-row offset of row symbol e. A lane reads the four entries s0 - lane .. s0 - lane + 3 of a group of
    // steps with ONE aligned ds_read_b64: the ring is kept in four copies, copy c shifted by c entries, and lane l uses
    // copy l & 3 (a misaligned wide LDS read is served lane by lane: 65 cycles instead of 4).
    uint16_t *ring = (uint16_t *)(smem + ring_at);
    auto row_offset = [&](uint32_t idx) -> uint32_t {
        const uint32_t sym = idx < rows ? (uint32_t)row_data[idx] : 0u;
        return (uint32_t)cmap[sym] * row_bytes;
    };
    auto ring_store = [&](uint32_t entry, uint32_t value) {
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) ring[c * kRingEntries + ((entry + c) & (kRingEntries - 1))] = (uint16_t)value;
    };
    // entries [-128, 0) are never used by an active lane but are read: keep them inside the profile; [0, 128) are loaded now
    ring_store(128 + lane, 0); ring_store(192 + lane, 0);
    ring_store(lane, row_offset(lane));
    ring_store(64 + lane, row_offset(64 + lane));
    wave_lds_fence();

    static_assert(!kNarrow || kAffine, "narrow strips: affine gaps");
    // kNarrow with kLocal: no shifting at all -- Smith-Waterman's H, E and F are non-negative and bounded by the best score, so when the
    // launch knows that no score can reach 2^16 (largest cost x shorter string) the two plain maxima of a cell are v_max_u16 as they are
    constexpr bool kShift = kNarrow && !kLocal;
    int H[WE];
    [[maybe_unused]] int F[kAffine ? WE : 1];
    const int h0_true = kLocal ? 0 : (kAffine ? 2 * open_minus_ext : 0);   // row 0: global -- relative to the all-gaps baseline (wavefront.hip: kSkew / kSkewAffine); local -- 0
    [[maybe_unused]] const uint32_t open_abs = (uint32_t)-open, ext_abs = (uint32_t)-ext;
    [[maybe_unused]] int G[kLocal ? WE : 1];   // (local) sat(H - |open|) of the strip
    [[maybe_unused]] int shift = kShift ? h0_true - kCenter : 0;           // (narrow strips) true value = stored value + shift
    const int h0 = kShift ? kCenter : h0_true;
    // "minus infinity": narrow strips never add to one (F of row 0 may as well be H of row 0: max(H, F) is H either way; E of
    // column 0 is 0, below every stored value)
    constexpr int kNoF = kLocal ? 0 : kNegInfP, kNoE = (kShift || kLocal) ? 0 : kNegInfP;
#pragma unroll
    for (int k = 0; k < WE; ++k) {
        H[k] = h0;
        if constexpr (kAffine) F[k] = kShift ? h0 : kNoF;
        if constexpr (kLocal) G[k] = 0;
    }
    int out_h = h0;
    [[maybe_unused]] int out_e = kShift ? h0 : kNoE;
    int prev_h = kLocal ? 0 : (kAffine ? (mine ? h0 : open_minus_ext - (kShift ? shift : 0)) : 0);   // H[0][my first column - 1]
    // left-edge inputs of lane 0: the DP boundary column (pass 0: a constant, never reloaded) or the previous pass's right edge
    int bnd_next[4] = {h0, h0, h0, h0}, ebnd_next[4] = {kNoE, kNoE, kNoE, kNoE};
    int bnd_cur[4], ebnd_cur[4];
    constexpr bool read_bnd = kRead, write_bnd = kWrite;
    int *rring_h = (int *)(smem + ring_at + kRingBytes), *rring_e = rring_h + kBndReadRows;   // row r at slot (r - 1) & 127
    int *wring_h = rring_e + kBndReadRows, *wring_e = wring_h + kBndWriteRows;                // row r at slot r & 63
    int bload_h = 0;
    [[maybe_unused]] int bload_e = 0;
    auto bnd_request = [&](uint32_t first_row) {   // rows first_row .. first_row + 63 of the previous pass's right edge
        const uint32_t r = first_row + lane < rows ? first_row + lane : rows;   // (rows past the pair are never used)
        bload_h = __hip_atomic_load(bnd_h + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (kAffine) bload_e = __hip_atomic_load(bnd_e + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto bnd_deliver = [&](uint32_t first_row) {
        rring_h[(first_row - 1 + lane) & (kBndReadRows - 1)] = bload_h;
        if constexpr (kAffine) rring_e[(first_row - 1 + lane) & (kBndReadRows - 1)] = bload_e;
    };
    auto bnd_flush = [&](int first_row) {          // rows first_row .. first_row + 63 of my right edge go to global memory
        const int r = first_row + lane;
        if (r >= 1 && (uint32_t)r <= rows) {
            bnd_h[r] = wring_h[r & (kBndWriteRows - 1)] + (kShift ? shift : 0);
            if constexpr (kAffine) bnd_e[r] = wring_e[r & (kBndWriteRows - 1)] + (kShift ? shift : 0);
        }
    };
    if (read_bnd) {
        bnd_request(1); bnd_deliver(1);
        bnd_request(65); bnd_deliver(65);
        wave_lds_fence();
        if constexpr (kShift) {   // rows 1 .. 64 become stored values (rows 65 .. 128: at the end of the first block)
            rring_h[lane] -= shift;
            rring_e[lane] -= shift;
            wave_lds_fence();
        }
        const int4 h4 = *(const int4 *)rring_h;
        bnd_next[0] = h4.x; bnd_next[1] = h4.y; bnd_next[2] = h4.z; bnd_next[3] = h4.w;
        if constexpr (kAffine) {
            const int4 e4 = *(const int4 *)rring_e;
            ebnd_next[0] = e4.x; ebnd_next[1] = e4.y; ebnd_next[2] = e4.z; ebnd_next[3] = e4.w;
        }
    }
    // profile-row offsets of four steps at a time, one group ahead; the scores of a step one step ahead
    const char *my_ring = smem + ring_at + (lane & 3) * (2 * kRingEntries);
    const uint32_t lane4 = (uint32_t)lane & ~3u;
    uint2 off_cur, off_nxt;
    auto read_offsets = [&](uint32_t s0) {
        off_nxt = *(const uint2 *)(my_ring + ((s0 - lane4) & (kRingEntries - 1)) * 2);
    };
    auto offset_of = [&](const uint2 &o, int u) -> uint32_t { return ((u < 2 ? o.x : o.y) >> (16 * (u & 1))) & 0xFFFFu; };
    // the scores of a step are requested TWO steps ahead (an LDS round trip under this load is longer than one step of a
    // narrow strip): four buffers, step u of a group uses buffer u, so every index is static
    // (Measured in round 6 and not kept: one `ds_read_i8` per cell, so that the diagonal sum is a plain v_add_u32 (2.45 SIMD cycles)
    // instead of v_add_u32_sdwa (4.4) -- four times the LDS instructions: C4 linear 11.7 -> 8.7 TCUPS, SW linear 8.2 -> 6.7.)
    uint32_t sc[4][kPlanes];
    auto read_scores = [&](int buffer, uint32_t off) {
        const uint32_t *row = (const uint32_t *)(smem + off);
#pragma unroll
        for (int pl = 0; pl < kPlanes; ++pl) sc[buffer][pl] = row[pl * 64 + lane];
    };
    read_offsets(0);
    read_scores(0, offset_of(off_nxt, 0));
    read_scores(1, offset_of(off_nxt, 1));
    uint32_t refill = 0;   // row symbols of entries [s0 + 128, s0 + 192), requested at the top of a 64-step block
    // Four steps. kAllActive: every lane is inside its rows (steps 63 .. rows - 1: all but the first and the last 63 of a
    // pass), so the per-lane activity test -- an add, a compare and an exec-mask round trip per step -- is left out.
    auto group = [&](const uint32_t s0, auto all_active_tag) {
        constexpr bool kAllActive = decltype(all_active_tag)::value;
        if ((s0 & 63u) == 0) {
            refill = row_offset(s0 + 128 + lane);
            if (read_bnd) bnd_request(s0 + 129);   // delivered at the end of this block, used from step s0 + 128 on
        }
        off_cur = off_nxt;
#pragma unroll
        for (int u = 0; u < 4; ++u) { bnd_cur[u] = bnd_next[u]; ebnd_cur[u] = ebnd_next[u]; }
        read_offsets(s0 + 4);
        if (read_bnd) {   // rows s0 + 5 .. s0 + 8: what lane 0 takes in at the next group of steps
            const int4 h4 = *(const int4 *)(rring_h + ((s0 + 4) & (kBndReadRows - 1)));
            bnd_next[0] = h4.x; bnd_next[1] = h4.y; bnd_next[2] = h4.z; bnd_next[3] = h4.w;
            if constexpr (kAffine) {
                const int4 e4 = *(const int4 *)(rring_e + ((s0 + 4) & (kBndReadRows - 1)));
                ebnd_next[0] = e4.x; ebnd_next[1] = e4.y; ebnd_next[2] = e4.z; ebnd_next[3] = e4.w;
            }
        }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t s = s0 + u;
                read_scores((u + 2) & 3, u < 2 ? offset_of(off_cur, u + 2) : offset_of(off_nxt, u - 2));
                __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks the ds_reads next to their use
                const int recv_h = dpp_wave_shr1(bnd_cur[u], out_h);
                int recv_e = kNoE;
                if constexpr (kAffine) recv_e = dpp_wave_shr1(ebnd_cur[u], out_e);
                if (kAllActive || s - (uint32_t)lane < rows) {   // active: DP row r = s - lane + 1
                    int left = recv_h, e = recv_e;
                    if constexpr (kLocal) left = (int)__builtin_elementwise_sub_sat((uint32_t)recv_h, open_abs);   // what my first cell takes from the left: G
                    int t_pending = 0;
                    // A cell is computed in two halves one column apart (wavefront.hip: in-place strips): `cell(k)` first adds the
                    // substitution score to the diagonal -- H[k - 1] of the previous row, still in its register -- and only then
                    // finishes cell k - 1, whose new value can so be written in place (tied asm operand).
                    auto finish = [&](int k, int t, int after) {
                        if constexpr (kLocal) {
                            if constexpr (kAffine) {
                                const int fdec = (int)__builtin_elementwise_sub_sat((uint32_t)F[k], ext_abs), edec = (int)__builtin_elementwise_sub_sat((uint32_t)e, ext_abs);
                                const int f = kNarrow ? umax16(G[k], fdec) : max(G[k], fdec);
                                F[k] = f;
                                e = kNarrow ? umax16(left, edec) : max(left, edec);
                                asm("v_max3_i32 %0, %1, %2, %3" : "+v"(H[k]) : "v"(t), "v"(e), "v"(f), "v"(after));
                            } else {
                                asm("v_max3_i32 %0, %1, %2, %3" : "+v"(H[k]) : "v"(t), "v"(G[k]), "v"(left), "v"(after));
                            }
                            G[k] = (int)__builtin_elementwise_sub_sat((uint32_t)H[k], open_abs);
                            left = G[k];
                        } else if constexpr (!kAffine) {
                            asm("v_max3_i32 %0, %1, %0, %2" : "+v"(H[k]) : "v"(t), "v"(left), "v"(after));
                            left = H[k];
                        } else {
                            const int f = kNarrow ? umax16(H[k], F[k]) : max(H[k], F[k]);
                            F[k] = f;
                            e = kNarrow ? umax16(left, e) : max(left, e);
                            const int h3 = max(max(t, e), f);
                            asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open_minus_ext), "v"(after));
                            left = H[k];
                        }
                    };
                    [[maybe_unused]] int t_even = 0;
                    auto cell = [&](int k, int sc) {
                        const int t = (k == 0 ? prev_h : H[k > 0 ? k - 1 : 0]) + sc;
                        if (k > 0) finish(k - 1, t_pending, t);
                        t_pending = t;
                        if constexpr (kLocal) {   // the running maximum: two diagonal candidates per v_max3
                            if (k & 1) best = max(max(best, t_even), t);
                            else t_even = t;
                        }
                    };
                    if constexpr (kAffine && !kLocal) {
                        // Gotoh, global: four cells per asm statement, in a fixed order (SWH_NWP_CELL). What this buys is the wait
                        // states: hipcc cannot see into an asm statement and puts an s_nop in front of every reader of what one defines --
                        // 45 per four steps behind the v_max_u16 of narrow strips, 132 when every instruction is its own statement, 18
                        // with one statement per four cells; and an s_nop costs a wave an issue slot like any instruction. (The order
                        // itself is not what matters: tools/gotoh_sched.hip, profiles/r3/gotoh_schedule_cycles.txt -- the five operations
                        // of a cell cost 19.5 SIMD cycles with v_max_u16, 21.3 with v_max_i32 at two waves per SIMD, chain back to back or
                        // interleaved. With the strip's other instructions the narrow kernel runs at 22.8: at its VALU floor.)
                        int t[WE + 2];
                        t[0] = prev_h + (int)(int8_t)sc[u][0];
                        t[1] = H[0] + (int)(int8_t)(sc[u][0] >> 8);
                        F[0] = kNarrow ? umax16(H[0], F[0]) : max(H[0], F[0]);   // the row above, gap opened or extended
#pragma unroll
                        for (int pl = 0; pl < kPlanes; ++pl) {
                            const int k0 = 4 * pl;
                            int h3;
                            if (pl + 1 < kPlanes) {
                                const int kn = 4 * (pl + 1 < kPlanes ? pl + 1 : 0);   // (in range when the branch is not taken)
                                if constexpr (kNarrow)
                                    asm volatile(SWH_NWP_PLANE("v_max_u16")
                                                 : [h0] "+v"(H[k0]), [h1] "+v"(H[k0 + 1]), [h2] "+v"(H[k0 + 2]), [h3] "+v"(H[k0 + 3]), [f1] "+v"(F[k0 + 1]),
                                                   [f2] "+v"(F[k0 + 2]), [f3] "+v"(F[k0 + 3]), [f4] "+v"(F[kn]), [e] "+v"(e), [t2] "=&v"(t[k0 + 2]),
                                                   [t3] "=&v"(t[k0 + 3]), [t4] "=&v"(t[k0 + 4]), [t5] "=&v"(t[k0 + 5]), [tmp] "=&v"(h3)
                                                 : [left] "v"(left), [f0] "v"(F[k0]), [hn] "v"(H[kn]), [t0] "v"(t[k0]), [t1] "v"(t[k0 + 1]), [s0] "v"(sc[u][pl]),
                                                   [s1] "v"(sc[u][pl + 1 < kPlanes ? pl + 1 : 0]), [c] "v"(open_minus_ext));
                                else
                                    asm volatile(SWH_NWP_PLANE("v_max_i32")
                                                 : [h0] "+v"(H[k0]), [h1] "+v"(H[k0 + 1]), [h2] "+v"(H[k0 + 2]), [h3] "+v"(H[k0 + 3]), [f1] "+v"(F[k0 + 1]),
                                                   [f2] "+v"(F[k0 + 2]), [f3] "+v"(F[k0 + 3]), [f4] "+v"(F[kn]), [e] "+v"(e), [t2] "=&v"(t[k0 + 2]),
                                                   [t3] "=&v"(t[k0 + 3]), [t4] "=&v"(t[k0 + 4]), [t5] "=&v"(t[k0 + 5]), [tmp] "=&v"(h3)
                                                 : [left] "v"(left), [f0] "v"(F[k0]), [hn] "v"(H[kn]), [t0] "v"(t[k0]), [t1] "v"(t[k0 + 1]), [s0] "v"(sc[u][pl]),
                                                   [s1] "v"(sc[u][pl + 1 < kPlanes ? pl + 1 : 0]), [c] "v"(open_minus_ext));
                            } else {
                                if constexpr (kNarrow)
                                    asm volatile(SWH_NWP_LAST_PLANE("v_max_u16")
                                                 : [h0] "+v"(H[k0]), [h1] "+v"(H[k0 + 1]), [h2] "+v"(H[k0 + 2]), [h3] "+v"(H[k0 + 3]), [f1] "+v"(F[k0 + 1]),
                                                   [f2] "+v"(F[k0 + 2]), [f3] "+v"(F[k0 + 3]), [e] "+v"(e), [t2] "=&v"(t[k0 + 2]), [t3] "=&v"(t[k0 + 3]), [tmp] "=&v"(h3)
                                                 : [left] "v"(left), [f0] "v"(F[k0]), [t0] "v"(t[k0]), [t1] "v"(t[k0 + 1]), [s0] "v"(sc[u][pl]), [c] "v"(open_minus_ext));
                                else
                                    asm volatile(SWH_NWP_LAST_PLANE("v_max_i32")
                                                 : [h0] "+v"(H[k0]), [h1] "+v"(H[k0 + 1]), [h2] "+v"(H[k0 + 2]), [h3] "+v"(H[k0 + 3]), [f1] "+v"(F[k0 + 1]),
                                                   [f2] "+v"(F[k0 + 2]), [f3] "+v"(F[k0 + 3]), [e] "+v"(e), [t2] "=&v"(t[k0 + 2]), [t3] "=&v"(t[k0 + 3]), [tmp] "=&v"(h3)
                                                 : [left] "v"(left), [f0] "v"(F[k0]), [t0] "v"(t[k0]), [t1] "v"(t[k0 + 1]), [s0] "v"(sc[u][pl]), [c] "v"(open_minus_ext));
                            }
                            left = H[k0 + 3];
                        }
                    } else {
#pragma unroll
                        for (int pl = 0; pl < kPlanes; ++pl) {
                            const uint32_t c4 = sc[u][pl];
#pragma unroll
                            for (int i = 0; i < 4; ++i) cell(4 * pl + i, (int)(int8_t)(c4 >> (8 * i)));
                        }
                        finish(WE - 1, t_pending, t_pending);
                    }
                    out_h = kLocal ? H[WE - 1] : left;   // (local: `left` is G, the neighbour takes the true H and derives its own)
                    if constexpr (kAffine) out_e = e;
                    if constexpr (write_bnd) {
                        if (lane == 63) {   // row s - 62
                            wring_h[(s - 62) & (kBndWriteRows - 1)] = out_h;
                            if constexpr (kAffine) wring_e[(s - 62) & (kBndWriteRows - 1)] = e;
                        }
                    }
                }
                prev_h = recv_h;
            }
        if ((s0 & 63u) == 60) {   // end of a 64-step block: entries [s0 - 60 + 128, + 64) enter the ring
            ring_store(s0 - 60 + 128 + lane, refill);
            if (read_bnd) bnd_deliver(s0 - 60 + 129);
            wave_lds_fence();
            if (write_bnd) bnd_flush((int)(s0 - 60) - 62);   // lane 63 did rows s0 - 60 - 62 .. s0 - 60 + 1 in this block
            if constexpr (kShift) {
                // the middle of the wave goes back to kCenter (after the flush: what the block parked was stored under the old shift)
                const int delta = __builtin_amdgcn_readlane(H[WE / 2], 32) - kCenter;
                shift += delta;
#pragma unroll
                for (int k = 0; k < WE; ++k) { H[k] -= delta; F[k] -= delta; }
                out_h -= delta; out_e -= delta; prev_h -= delta;
                if (read_bnd) {
                    // the left edge of the next block, rows s0 + 5 .. s0 + 68: true values in the ring (delivered two blocks ago) and in
                    // the registers fetched at the top of this group
                    const uint32_t slot = (s0 + 4 + (uint32_t)lane) & (kBndReadRows - 1);
                    rring_h[slot] -= shift;
                    rring_e[slot] -= shift;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { bnd_next[u] -= shift; ebnd_next[u] -= shift; }
                    wave_lds_fence();
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) bnd_next[u] -= delta;   // the DP's own left edge: a constant in true values
                }
            }
        }
    };
    // three loops rather than one with two bodies: where two differently allocated bodies join, hipcc reconciles their
    // registers with a v_mov per strip column and step
    uint32_t at = 0;
    for (; at < 64 && at < steps; at += 4) group(at, std::false_type{});
    for (; at + 3 < rows; at += 4) group(at, std::true_type{});
    for (; at < steps; at += 4) group(at, std::false_type{});
    if (write_bnd && (steps & 63u)) {   // the rows of the last, partial block
        wave_lds_fence();
        bnd_flush((int)(steps & ~63u) - 62);
    }
    // the score lives in the lane / register holding column `cols` (local alignment: the caller reduces `best` after the last pass)
    if (!kLocal && cols > c0 && cols <= c0 + 64 * WE) {
        const uint32_t jj = cols - 1 - c0;
        if ((uint32_t)lane == jj / WE) {
            const uint32_t kk = jj % WE;
            int result = 0;
#pragma unroll
            for (int k = 0; k < WE; ++k)
                if ((uint32_t)k == kk) result = H[k];
            if constexpr (kShift) result += shift;
            result += (int)(rows + cols) * ext;
            if constexpr (kAffine) result -= open_minus_ext;   // the strip holds H^ + (open - ext)
            store_result(args.job, p, (int64_t)result);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);   // boundary stores land before the next pass (or pair) reads / overwrites
    wave_lds_fence();                // and nobody is still reading the profile or the ring
}

template <int W, bool kAffine, bool kLocal, bool kNarrow>
__global__ __launch_bounds__(64) void k_nwprofile(KernelArgs args, uint32_t first, uint32_t count, uint32_t classes) {
    static_assert(W % 4 == 0 && W >= 4 && W <= 16, "strips are handled four columns at a time");
    constexpr uint32_t kRowBytes = 64u * W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t ring_at = classes * kRowBytes;
    const uint32_t cmap_at = ring_at + kScratchBytes;
    const int lane = threadIdx.x;
    // the byte -> class map stays resident; the 32x32 biased class costs (sub - ext - open, api.hip: alignment_init) are
    // reloaded per pass into the rings' space (run_pass)
    const bool wide = args.scoring.class_table == nullptr;   // 33 .. 128 classes: Scoring::wide_table, rows of kWideClasses bytes
    const uint8_t *table_src = wide ? args.scoring.wide_table : args.scoring.class_table;
    const uint32_t cstride = wide ? kWideClasses : 32u;
    ((uint32_t *)(smem + cmap_at))[lane] = ((const uint32_t *)(table_src + cstride * cstride))[lane];
    wave_lds_fence();
    const uint8_t *ctab = (const uint8_t *)(smem + ring_at);
    const uint8_t *cmap = (const uint8_t *)(smem + cmap_at);
    int32_t *bnd_h = args.boundary + (uint64_t)blockIdx.x * args.boundary_stride * 2;
    int32_t *bnd_e = bnd_h + args.boundary_stride;

    for (;;) {
        uint32_t drawn = 0;
        if (lane == 0) drawn = __hip_atomic_fetch_add(args.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drawn = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        if (drawn >= count) break;
        const uint64_t p = args.perm[first + count - 1 - drawn];   // widest pairs first
        uint64_t a0, b0;
        uint32_t la, lb;
        if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
        else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
        // Orientation: columns live across lanes; rows stream. Symmetric scoring puts the shorter string on the columns
        // (as plan_key() and k_wavefront do).
        const bool swapped = args.symmetric && la < lb;
        const uint32_t rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)(swapped ? lb : la));
        const uint32_t cols = (uint32_t)__builtin_amdgcn_readfirstlane((int)(swapped ? la : lb));
        const uint8_t *col_data = (const uint8_t *)(swapped ? args.job.a.data : args.job.b.data) + (swapped ? a0 : b0);
        const uint8_t *row_data = (const uint8_t *)(swapped ? args.job.b.data : args.job.a.data) + (swapped ? b0 : a0);
        // Passes of 64 x W columns, then ONE narrower pass for what is left, rounded up to four columns per lane: at most 255
        // phantom columns per pair whatever its length.
        const uint32_t full = cols / (64 * W);
        const uint32_t rest = cols - full * 64 * W;
        const uint32_t w_last = ((rest + 63) / 64 + 3) & ~3u;           // 0: the full passes cover the pair
        const uint32_t passes = full + (rest ? 1u : 0u);
        int best = 0;   // (local alignment) maximum over the cells of my strips, all passes
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t c0 = pass * 64 * W;
            const uint32_t w = pass < full ? (uint32_t)W : w_last;
#define SWH_PASS2(WE, RD, WR)                                                                                                  \
    run_pass<WE, kAffine, RD, WR, kLocal, kNarrow>(args, smem, ring_at, ctab, cmap, table_src, cstride, classes, kRowBytes, col_data, row_data, rows, cols, c0, bnd_h, bnd_e, p, best)
#define SWH_PASS(WE)                                                                    \
    do {                                                                                \
        if (pass == 0) { if (passes == 1) SWH_PASS2(WE, false, false); else SWH_PASS2(WE, false, true); } \
        else if (pass + 1 < passes) SWH_PASS2(WE, true, true);                          \
        else SWH_PASS2(WE, true, false);                                                \
    } while (0)
            if (w == 4) SWH_PASS(4);
            else if (W >= 8 && w == 8) SWH_PASS(W >= 8 ? 8 : 4);
            else if (W >= 12 && w == 12) SWH_PASS(W >= 12 ? 12 : 4);
            else SWH_PASS(W);
#undef SWH_PASS
#undef SWH_PASS2
        }
        if constexpr (kLocal) {
            // the maximum over the wave's lanes, through the (now idle) ring space -- an LDS atomic, not wave_max_u32(): in this
            // kernel (SGPR spills to VGPR lanes, long branches) the DPP row_bcast reduction never came back
            int *slot = (int *)(smem + ring_at);
            if (lane == 0) *slot = 0;
            wave_lds_fence();
            __hip_atomic_fetch_max(slot, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (scores are >= 0)
            wave_lds_fence();
            if (lane == 0) store_result(args.job, p, (int64_t)*slot);
            wave_lds_fence();
        }
    }
}

template <int W, bool kAffine, bool kLocal, bool kNarrow>
void launch_w(Scope *scope, const KernelArgs &args, uint32_t first, uint32_t count, uint32_t classes, uint32_t blocks, const char *name) {
    const size_t lds = (size_t)classes * 64 * W + kScratchBytes + 256;
    opt_in_dynamic_lds(scope, (const void *)k_nwprofile<W, kAffine, kLocal, kNarrow>, lds);
    StampGuard guard(scope, name);
    hipLaunchKernelGGL((k_nwprofile<W, kAffine, kLocal, kNarrow>), dim3(blocks), dim3(64), lds, scope->stream, args, first, count, classes);
}

template <bool kAffine, bool kLocal, bool kNarrow = false>
void launch_strip(Scope *scope, const KernelArgs &args, uint32_t first, uint32_t count, uint32_t classes, uint32_t blocks, uint32_t strip,
                  const char *n16, const char *n12, const char *n8, const char *n4) {
    if (strip == 16) launch_w<16, kAffine, kLocal, kNarrow>(scope, args, first, count, classes, blocks, n16);
    else if (strip == 12) launch_w<12, kAffine, kLocal, kNarrow>(scope, args, first, count, classes, blocks, n12);
    else if (strip == 8) launch_w<8, kAffine, kLocal, kNarrow>(scope, args, first, count, classes, blocks, n8);
    else launch_w<4, kAffine, kLocal, kNarrow>(scope, args, first, count, classes, blocks, n4);
}

}  // namespace

// Strip width by the classes in use: classes x 64 W bytes of profile per wave, eight or more waves per CU.
uint32_t nwprofile_strip(uint32_t classes) {
    // comparison knob STRINGWARS_AMD_NWP_STRIP=8|12|16 (honoured when the profile still fits a workgroup's 64 KB)
    static const uint32_t forced = [] { const char *e = test_hook("STRINGWARS_AMD_NWP_STRIP"); return e ? (uint32_t)atoi(e) : 0u; }();
    if ((forced == 4 || forced == 8 || forced == 12 || forced == 16) && (size_t)classes * 64 * forced + kScratchBytes + 256 <= 65536 && classes * 64 * forced <= 65535) return forced;
    // beyond the 32 classes of the register model (Scoring::wide_table): the profile's rows are what LDS holds -- strips of eight columns
    // as long as five single-wave workgroups fit a CU (56 classes: 32 KB each), of four beyond (128 classes: 36 KB, four per CU).
    // Measured on 2 K pairs of ~4 KB over 52 letters (53 classes, bench.py c4_letters52): W = 8 on five waves 5.3 TCUPS, W = 4 on nine 4.5 --
    // against 3.8 for one LDS look-up per cell (k_wavefront's matrix model); the 21 classes of config C4 run W = 12 on eight waves at 11.8.
    return classes <= 16 ? 16u : (classes <= 24 ? 12u : (classes <= 56 ? 8u : 4u));
}

// Narrow strips (see kCenter): everything a wave holds at one time must fit 16 bits around its middle.
// Comparison knob: STRINGWARS_AMD_NWP_NARROW=0 keeps the 32-bit maxima.
static bool nwprofile_narrow(const Scoring &scoring, uint32_t strip) {
    static const bool off = [] { const char *e = test_hook("STRINGWARS_AMD_NWP_NARROW"); return e && e[0] == '0'; }();
    return !off && scoring.step_span && (uint64_t)scoring.step_span * (64 * strip + 256) <= 30000;
}

uint32_t nwprofile_waves(const Scope *scope, uint32_t classes) {
    const size_t lds = (size_t)classes * 64 * nwprofile_strip(classes) + kScratchBytes + 256;
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
    if (per_cu > 16) per_cu = 16;
    return (uint32_t)scope->compute_units * (per_cu ? per_cu : 1u);
}

// Pairs perm[first, first + count) of the plan (every one of them with more than kNwProfileMinCols columns).
// args.boundary: nwprofile_waves() x 2 x boundary_stride int32; args.ticket: zeroed here.
void launch_nwprofile(Scope *scope, KernelArgs args, uint32_t first, uint32_t count) {
    if (!count) return;
    const uint32_t classes = args.scoring.classes ? args.scoring.classes : 32;
    uint32_t blocks = nwprofile_waves(scope, classes);
    if (blocks > count) blocks = count;
    args.ticket = scope->plan_leftover + 7;
    SWH_HIP_CHECK(hipMemsetAsync(args.ticket, 0, 4, scope->stream));
    const uint32_t strip = nwprofile_strip(classes);
    if (!args.local) {
        if (!args.affine) launch_strip<false, false>(scope, args, first, count, classes, blocks, strip, "nwprofile_w16", "nwprofile_w12", "nwprofile_w8", "nwprofile_w4");
        else if (nwprofile_narrow(args.scoring, strip))
            launch_strip<true, false, true>(scope, args, first, count, classes, blocks, strip, "nwprofile_affine_narrow_w16", "nwprofile_affine_narrow_w12", "nwprofile_affine_narrow_w8", "nwprofile_affine_narrow_w4");
        else launch_strip<true, false>(scope, args, first, count, classes, blocks, strip, "nwprofile_affine_w16", "nwprofile_affine_w12", "nwprofile_affine_w8", "nwprofile_affine_w4");
    } else {
        if (!args.affine) launch_strip<false, true>(scope, args, first, count, classes, blocks, strip, "nwprofile_local_w16", "nwprofile_local_w12", "nwprofile_local_w8", "nwprofile_local_w4");
        else if (args.local_narrow)   // no score of the batch reaches 2^16: api.hip
            launch_strip<true, true, true>(scope, args, first, count, classes, blocks, strip, "nwprofile_local_affine_narrow_w16", "nwprofile_local_affine_narrow_w12", "nwprofile_local_affine_narrow_w8", "nwprofile_local_affine_narrow_w4");
        else launch_strip<true, true>(scope, args, first, count, classes, blocks, strip, "nwprofile_local_affine_w16", "nwprofile_local_affine_w12", "nwprofile_local_affine_w8", "nwprofile_local_affine_w4");
    }
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
