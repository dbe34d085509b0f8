// wavefront.hip -- anti-diagonal wavefront DP for gfx950, register-tiled, DPP hand-off.
//
// One pair is scored by a group of G lanes (G = 16: four pairs per wave64, `row_shr:1` DPP;
// G = 64: one pair per wave, `wave_shr:1` DPP). Lane g of a group owns W consecutive DP columns
// (the "strip" g*W+1 .. g*W+W) and keeps the previous row of its strip in registers. At global
// step s lane g processes DP row r = s - g + 1, so the lanes of a group sit on one anti-diagonal
// of W-wide tiles; the only cross-lane traffic per step is the right-edge value of the strip
// (plus the horizontal-gap state for affine gaps), shifted one lane up by a single DPP move.
// Row symbols are fetched four steps ahead with one (possibly unaligned) dword load per lane.
//
// The core is max-plus (global alignment score). Distances run on it with negated costs
// (`Job::negate`), so one kernel family serves `LevenshteinDistances` with arbitrary
// (match, mismatch, open, extend) (bench.rs:382), its UTF-8 twin on decoded code points
// (bench.rs:386) and `NeedlemanWunschScores` with a 256x256 matrix (bench.rs:658, config C4).
//
// Strips wider than one pass (columns > G*W of the widest instantiation) run as several passes
// over column chunks; the chunk's right edge column is parked in a per-group global buffer.
#include <cstdlib>

#include "common.hpp"

namespace swh {

enum WfModel : int {
    kUniformLinear = 0, kMatrixLinear = 1, kMatrixAffine = 2, kMatrixLinearLocal = 3, kMatrixAffineLocal = 4,
    kClassLinear = 5, kClassAffine = 6,  // <= 32 symbol classes: cost rows live in registers, bytes picked by v_perm_b32
    kUniformAffine = 7,                  // match / mismatch by comparison (any symbol width), Gotoh gaps: general-cost Levenshtein
    kClassLinearLocal = 8, kClassAffineLocal = 9   // Smith-Waterman on the class tables (cost rows without the global models' bias)
};
constexpr size_t kClassLds = 32 * 32 + 256;  // 32x32 i8 class costs, then the byte -> class map

constexpr int kNegInf = -0x20000000;
constexpr int kMatrixStride = 260;          // bytes per LDS matrix row (256 + one dword of padding)
constexpr size_t kMatrixLds = 256 * kMatrixStride;

template <int G>
__device__ __forceinline__ int dpp_shift_up(int old, int src) {
    if constexpr (G == 16) return __builtin_amdgcn_update_dpp(old, src, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
    else return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

__device__ __forceinline__ int med3i(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Row-symbol stream: symbol for step s is rows[s - g]; four steps are fetched per refill.
template <typename Sym>
struct RowStream;

template <>
struct RowStream<uint8_t> {
    const uint8_t *base;  // rows string start
    int lo, hi;           // clamp range of dword-read start indices relative to `base`
    int avail;            // readable symbols relative to base (total - start)
    bool tiny;            // tape shorter than 4 bytes: bytewise path
    uint32_t cur, nxt;
    int nxt_shift;        // realignment of `nxt`, applied where it is consumed (next to the load it would expose the latency)
    __device__ __forceinline__ void init(const uint8_t *data, uint64_t start, uint64_t total) {
        base = data + start;
        tiny = total < 4;
        int64_t l = -(int64_t)start, h = (int64_t)total - (int64_t)start - 4;
        lo = (int)(l < -0x40000000ll ? -0x40000000ll : l);
        hi = (int)(h > 0x40000000ll ? 0x40000000ll : h);
        int64_t av = (int64_t)total - (int64_t)start;
        avail = (int)(av > 0x40000000ll ? 0x40000000ll : av);
        cur = nxt = 0;
        nxt_shift = 24;
    }
    // bytes idx..idx+3 as loaded (clamped into the tape) plus the shift that puts byte idx first
    __device__ __forceinline__ void prefetch(int idx) {
        if (!tiny) {
            int c = med3i(idx, lo, hi);
            __builtin_memcpy(&nxt, base + c, 4);
            nxt_shift = 8 * med3i(idx - c, -3, 3) + 24;
            return;
        }
        uint32_t dw = 0;
        for (int u = 0; u < 4; ++u) {
            int pos = idx + u;
            if (pos >= lo && pos < avail) dw |= (uint32_t)base[pos] << (8 * u);
        }
        nxt = dw;
        nxt_shift = 24;
    }
    __device__ __forceinline__ uint32_t realigned_next() const {
        return (uint32_t)((((unsigned long long)nxt) << 24) >> (uint32_t)nxt_shift);
    }
    __device__ __forceinline__ void advance() { cur = realigned_next(); }
    __device__ __forceinline__ uint32_t sym(int u) const { return (cur >> (8 * u)) & 0xffu; }
    // row symbol of the step after u (u = 3: first symbol of the prefetched word)
    __device__ __forceinline__ uint32_t sym_after(int u) const { return u < 3 ? sym(u + 1) : (realigned_next() & 0xffu); }
};

template <>
struct RowStream<uint32_t> {
    const uint32_t *base;
    int lo, hi;
    uint32_t cur[4], nxt[4];
    __device__ __forceinline__ void init(const uint32_t *data, uint64_t start, uint64_t total) {
        base = data + start;
        int64_t l = -(int64_t)start, h = (int64_t)total - (int64_t)start - 1;
        lo = (int)(l < -0x40000000ll ? -0x40000000ll : l);
        hi = (int)(h > 0x40000000ll ? 0x40000000ll : h);
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u] = 0;
    }
    __device__ __forceinline__ void prefetch(int idx) {
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = base[med3i(idx + u, lo, hi)];
    }
    __device__ __forceinline__ void advance() {
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
    }
    __device__ __forceinline__ uint32_t sym(int u) const { return cur[u]; }
    __device__ __forceinline__ uint32_t sym_after(int u) const { return u < 3 ? cur[u + 1] : nxt[0]; }
};

// Result plumbing shared with the bit-parallel kernel.
__device__ __forceinline__ void store_score(const Job &job, uint64_t p, int score) {
    if (job.negate) store_result(job, p, (int64_t)clamp_bound((uint32_t)(-score), job.bound));
    else store_result(job, p, (int64_t)score);
}

// PQ (class models only): cost-row dword pairs that can hold a class, ceil(classes / 8) -- one v_perm per pair and
// group of four columns. Up to eight classes need one, 20 amino acids + "other" three, the full 32 classes four.
template <typename Sym, int G, int W, int MODEL, int PQ = 4>
__global__ __launch_bounds__(256) void k_wavefront(KernelArgs args, uint32_t cls) {
    constexpr bool kClass = MODEL == kClassLinear || MODEL == kClassAffine || MODEL == kClassLinearLocal || MODEL == kClassAffineLocal;
    constexpr bool kMatrix = MODEL != kUniformLinear && MODEL != kUniformAffine && !kClass;   // per-cell LDS gather from the 256x256 table
    constexpr bool kAffine = MODEL == kMatrixAffine || MODEL == kMatrixAffineLocal || MODEL == kClassAffine || MODEL == kUniformAffine || MODEL == kClassAffineLocal;
    static_assert(!kClass || W % 4 == 0, "class model handles columns four at a time");
    // Smith-Waterman (local) on the same tiles: boundaries and every cell are floored at 0 and the result is the
    // maximum over all cells (`SmithWatermanScores`, bench.rs:882-963).
    constexpr bool kLocal = MODEL == kMatrixLinearLocal || MODEL == kMatrixAffineLocal || MODEL == kClassLinearLocal || MODEL == kClassAffineLocal;
    // Linear gaps (g = open = extend), global alignment: registers hold scores relative to the all-gaps baseline,
    //     U[r][k] = H[r][k] - (r + k) * g,   so that   U[r][k] = max3( U[r-1][k-1] + (sub - 2g),  U[r-1][k],  U[r][k-1] ):
    // ONE add and one max3 per cell (the up / left terms need no gap add at all), every boundary is zero, lanes
    // exchange U values as they are, and the result is U + (rows + cols) * g. The class model folds -2g into its
    // cost table, so its cell is `v_add_u32_sdwa (sext byte) ; v_max3_i32`.
    constexpr bool kSkew = !kAffine && !kLocal;
    // Smith-Waterman on the class tables keeps H' = H + open in its strips (what the up / left terms need; the diagonal's
    // surplus is folded into the table, which holds sub - open): one addition per cell instead of two.
    constexpr bool kLocalBiased = kLocal && kClass;
    // Gotoh gets the same treatment relative to the extend cost: with X^ = X - (r + k) * ext for H, E and F,
    //     E^ = max(H^left + (open - ext), E^left)   F^ = max(H^up + (open - ext), F^up)   H^ = max3(H^diag + (sub - 2 ext), E^, F^)
    // -- six VALU instead of eight; the boundary row / column become the constant open - ext.
    constexpr bool kSkewAffine = kAffine && !kLocal;
    constexpr int kGroups = 64 / G;  // pairs per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t *lmatrix = (int8_t *)smem;

    if constexpr (kMatrix) {
        // rows padded to kMatrixStride bytes: bank = (row * 65 + col / 4) % 32 depends on the row symbol too,
        // so small alphabets (20 amino acids live in 7 dword columns) do not pile onto a few banks
        const uint32_t *src = (const uint32_t *)args.scoring.matrix;
        uint32_t *dst = (uint32_t *)lmatrix;
        for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) dst[(i >> 6) * (kMatrixStride / 4) + (i & 63)] = src[i];
        __syncthreads();
    }
    if constexpr (kClass) {
        const uint32_t *src = (const uint32_t *)args.scoring.class_table;
        for (int i = threadIdx.x; i < (int)kClassLds / 4; i += blockDim.x) ((uint32_t *)smem)[i] = src[i];
        __syncthreads();
    }
    [[maybe_unused]] const uint8_t *lclass_of = (const uint8_t *)smem + 1024;  // byte -> class (class model)

    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int gl = lane % G;       // lane inside its group
    const int grp = lane / G;      // group inside the wave
    const uint32_t cstart = args.plan->class_start[cls];
    const uint32_t ccount = args.plan->class_count[cls];
    const uint32_t chunks = (ccount + kGroups - 1) / kGroups;
    const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const int open = args.scoring.open, ext = args.scoring.extend;
    const int match = args.scoring.match, mismatch = args.scoring.mismatch;
    const uint64_t a_total = tape_total(args.job.a, args.off64);
    const uint64_t b_total = tape_total(args.job.b, args.off64);
    int32_t *bnd_h = args.boundary ? args.boundary + (uint64_t)(wave_id * kGroups + grp) * args.boundary_stride * 2
                                   : nullptr;
    int32_t *bnd_e = bnd_h ? bnd_h + args.boundary_stride : nullptr;

    // Largest chunks first (perm is sorted ascending by row bucket) for a short tail.
    for (uint32_t chunk_rev = wave_id; chunk_rev < chunks; chunk_rev += waves_total) {
        const uint32_t chunk = chunks - 1 - chunk_rev;
        const uint32_t slot = chunk * kGroups + grp;
        const bool have = slot < ccount;
        uint64_t p = 0, a0 = 0, b0 = 0;
        uint32_t la = 0, lb = 0;
        if (have) {
            p = args.perm[cstart + slot];
            if (args.off64) pair_extent<uint64_t>(args.job, p, a0, la, b0, lb);
            else pair_extent<uint32_t>(args.job, p, a0, la, b0, lb);
        }
        // Orientation: columns live across lanes; rows stream. Symmetric scoring puts the shorter
        // string on the columns (must mirror plan_key()).
        const bool swapped = args.symmetric && la < lb;
        const uint32_t rows = swapped ? lb : la, cols = swapped ? la : lb;
        const Sym *col_data = (const Sym *)(swapped ? args.job.a.data : args.job.b.data);
        const Sym *row_data = (const Sym *)(swapped ? args.job.b.data : args.job.a.data);
        const uint64_t col0 = swapped ? a0 : b0, row0 = swapped ? b0 : a0;
        const uint64_t row_total = swapped ? b_total : a_total;

        // wave-uniform step count: max(rows) + G - 1, rounded up to the 4-step refill cadence
        uint32_t rows_max = rows;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            uint32_t other = __shfl_xor(rows_max, off);
            rows_max = other > rows_max ? other : rows_max;
        }
        uint32_t cols_max = cols;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            uint32_t other = __shfl_xor(cols_max, off);
            cols_max = other > cols_max ? other : cols_max;
        }
        const uint32_t steps = (rows_max + G - 1 + 3) & ~3u;
        const uint32_t passes = cols_max ? (cols_max + G * W - 1) / (G * W) : 0;

        RowStream<Sym> stream;
        stream.init(row_data, row0, row_total);

        int result = 0;
        [[maybe_unused]] int best = 0;  // local alignment: running maximum of my strip
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t c0 = pass * G * W;  // columns c0+1 .. c0+G*W in this pass
            // column symbols of my strip (bytes packed four per register when the strip is wide)
            constexpr bool kPackCols = sizeof(Sym) == 1 && W >= 16 && W % 4 == 0;
            constexpr int kColRegs = kPackCols ? W / 4 : W;
            uint32_t bs[kClass ? 1 : kColRegs];
            // class model: per group of four columns, four v_perm selectors (one per pair of cost-row dwords): byte
            // i of selector p is (class_i & 7) when class_i lives in dwords 2p..2p+1 of the row, else 0x0C (zero)
            [[maybe_unused]] uint32_t sel[kClass ? (W / 4) * PQ : 1];
            int H[W], F[kAffine ? W : 1];
#pragma unroll
            for (int k = 0; k < (kClass ? 1 : kColRegs); ++k) bs[k] = 0;
            if constexpr (kClass) {
#pragma unroll
                for (int k = 0; k < (W / 4) * PQ; ++k) sel[k] = 0;
            }
#pragma unroll
            for (int k = 0; k < W; ++k) {
                uint32_t j = c0 + gl * W + k;  // 0-based column index
                uint32_t sym_k = j < cols ? (uint32_t)col_data[col0 + j] : 0u;
                if constexpr (kClass) {
                    const uint32_t cls_k = lclass_of[sym_k & 0xffu];
                    // (Smith-Waterman: columns right of the pair pick a zero byte everywhere = substitution `open` <= 0, so a
                    // phantom cell never exceeds the real cell it descends from and the running maximum needs no column test)
                    const bool picks = !kLocalBiased || j < cols;
#pragma unroll
                    for (int pq = 0; pq < PQ; ++pq)
                        sel[(k >> 2) * PQ + pq] |= (picks && (cls_k >> 3) == (uint32_t)pq ? (cls_k & 7u) : 0x0Cu) << (8 * (k & 3));
                } else if constexpr (kPackCols) bs[k >> 2] |= sym_k << (8 * (k & 3));
                else bs[k] = sym_k;
                H[k] = kLocalBiased ? open : (kLocal || kSkew) ? 0 : (kSkewAffine ? 2 * (open - ext) : open + (int)j * ext);  // row 0: H[0][j+1] = open + j*ext (skew-affine: H^ + (open - ext), see `finish`)
                if constexpr (kAffine) F[k] = kNegInf;
            }
            auto col_sym = [&](int k) -> uint32_t {
                if constexpr (kPackCols) return (bs[k >> 2] >> (8 * (k & 3))) & 0xffu;
                else return bs[k];
            };
            // my right-edge outputs (what the lane above me consumes), row 0
            int out_h = kLocalBiased ? open : (kLocal || kSkew) ? 0 : (kSkewAffine ? 2 * (open - ext) : open + (int)(c0 + gl * W + W - 1) * ext);
            int out_e = kNegInf;
            // diagonal input for my first active row: H[0][c0 + gl*W]
            int prev_h = kLocalBiased ? open : (!kLocal && !kSkew && (c0 + gl * W)) ? (kSkewAffine ? 2 * (open - ext) : open + (int)(c0 + gl * W - 1) * ext) : (kSkewAffine ? open - ext : 0);
            int bnd_next[4] = {0, 0, 0, 0}, ebnd_next[4] = {kNegInf, kNegInf, kNegInf, kNegInf};
            int bnd_cur[4], ebnd_cur[4];
            const bool read_bnd = pass > 0 && gl == 0;
            const bool write_bnd = pass + 1 < passes && gl == G - 1;

            stream.prefetch(0 - gl);
            if (read_bnd) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bnd_next[u] = __hip_atomic_load(bnd_h + 1 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if constexpr (kAffine)
                        ebnd_next[u] = __hip_atomic_load(bnd_e + 1 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // class model: the cost row of the NEXT step's row symbol is fetched (byte -> class, then 32 bytes of costs:
            // two dependent LDS round trips) while the current step's W cells are computed
            [[maybe_unused]] uint4 row_lo_next{0, 0, 0, 0}, row_hi_next{0, 0, 0, 0};
            [[maybe_unused]] auto fetch_cost_row = [&](uint32_t row_sym) {
                const uint32_t rc = lclass_of[row_sym & 0xffu];
                row_lo_next = *(const uint4 *)(smem + rc * 32);
                if constexpr (PQ > 2) row_hi_next = *(const uint4 *)(smem + rc * 32 + 16);
            };
            if constexpr (kClass) fetch_cost_row(stream.realigned_next());   // step 0's symbol = byte 0 of the first word
            for (uint32_t s0 = 0; s0 < steps; s0 += 4) {
                stream.advance();
                stream.prefetch((int)s0 + 4 - gl);
#pragma unroll
                for (int u = 0; u < 4; ++u) { bnd_cur[u] = bnd_next[u]; ebnd_cur[u] = ebnd_next[u]; }
                if (read_bnd) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bnd_next[u] = __hip_atomic_load(bnd_h + s0 + 5 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if constexpr (kAffine)
                            ebnd_next[u] =
                                __hip_atomic_load(bnd_e + s0 + 5 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t s = s0 + u;
                    // left-edge inputs: group lane 0 takes the DP boundary column, others the lane below
                    int edge_h = pass == 0 ? (kLocalBiased ? open : (kLocal || kSkew) ? 0 : (kSkewAffine ? 2 * (open - ext) : open + (int)s * ext)) : bnd_cur[u];  // H[s+1][c0]
                    int recv_h = dpp_shift_up<G>(edge_h, out_h);
                    int recv_e = kNegInf;
                    if constexpr (kAffine) recv_e = dpp_shift_up<G>(pass == 0 ? kNegInf : ebnd_cur[u], out_e);
                    const uint32_t sym = stream.sym(u);
                    [[maybe_unused]] uint4 row_lo_cur, row_hi_cur;
                    if constexpr (kClass) {
                        row_lo_cur = row_lo_next; row_hi_cur = row_hi_next;
                        fetch_cost_row(stream.sym_after(u));
                    }
                    if (s - (uint32_t)gl < rows) {  // active: DP row r = s - gl + 1
                        int left = recv_h, e = recv_e;
                        [[maybe_unused]] const int bias2 = -(open + ext);   // skewed models use substitution scores as sub - ext - open (= sub - 2g when linear; the affine strip holds H^ + (open - ext))
                        [[maybe_unused]] const int open_minus_ext = open - ext;
                        // A cell is computed in two halves one column apart: `cell(k, sc)` first adds the substitution score to
                        // the diagonal -- H[k - 1] of the previous row, still in its register -- and only then finishes cell
                        // k - 1, whose new value can so be written IN PLACE. Written the obvious way (finish cell k, keep the old
                        // H[k] as the next diagonal) the old and the new H[k] are alive together, the strip moves one register
                        // per row, and because rows are conditional (a lane is active for rows s - gl only) hipcc restored the
                        // canonical registers with W v_mov per row: a quarter of the instructions of a 64-80 column strip.
                        int t_pending = 0;
                        // The last instruction of a cell is written with H[k] as a tied operand: hipcc otherwise puts the new
                        // value into the dying diagonal's register. `after` (the next cell's diagonal sum, which read the OLD
                        // H[k]) is an operand only to keep that addition in front -- scheduled behind, the old H[k] needs a copy.
                        auto finish = [&](int k, int t, int after) {
                            if constexpr (kSkew) {
                                asm("v_max3_i32 %0, %1, %0, %2" : "+v"(H[k]) : "v"(t), "v"(left), "v"(after));   // max3(diag + sc, up, left): the score carries the -2g bias
                                left = H[k];
                                return;
                            }
                            const int up = H[k];
                            int x, y;   // h = max3(t, x, y)
                            if constexpr (kSkewAffine) {
                                // The strip holds H' = H^ + (open - ext): what E^ and F^ take their maxima with. One addition
                                // per cell instead of two (E^ = max(H'left, E^left), F^ = max(H'up, F^up)); the diagonal's
                                // surplus is folded into the substitution scores (sub - ext - open).
                                int f = max(up, F[k]);
                                F[k] = f;
                                e = max(left, e);
                                const int h3 = max(max(t, e), f);
                                asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open_minus_ext), "v"(after));
                                left = H[k];
                                return;
                            } else if constexpr (kAffine) {
                                int f = max(kLocalBiased ? up : up + open, F[k] + ext);
                                F[k] = f;
                                e = max(kLocalBiased ? left : left + open, e + ext);
                                x = e; y = f;
                            } else {
                                x = kLocalBiased ? up : up + open; y = kLocalBiased ? left : left + open;
                            }
                            if constexpr (kLocalBiased) {
                                const int h3 = max(max(max(t, x), y), 0);
                                best = max(best, h3);   // (phantom columns cannot win: see the selectors)
                                asm("v_add_u32 %0, %1, %2" : "+v"(H[k]) : "v"(h3), "v"(open), "v"(after));
                            } else if constexpr (kLocal) {
                                const int h3 = max(max(t, x), y);
                                asm("v_max_i32 %0, 0, %1" : "+v"(H[k]) : "v"(h3), "v"(after));
                                if (c0 + gl * W + k < cols) best = max(best, H[k]);  // phantom columns right of the pair do not count
                            } else {
                                asm("v_max3_i32 %0, %1, %2, %3" : "+v"(H[k]) : "v"(t), "v"(x), "v"(y), "v"(after));
                            }
                            left = H[k];
                        };
                        auto cell = [&](int k, int sc) {
                            const int t = (k == 0 ? prev_h : H[k > 0 ? k - 1 : 0]) + sc;   // diagonal: the previous row's H[k - 1]
                            if (k > 0) finish(k - 1, t_pending, t);
                            t_pending = t;
                        };
                        if constexpr (kClass) {
                            // one 32-byte cost row per step (the row symbol's class), then bytes are picked in
                            // registers: no per-cell LDS traffic at all
                            const uint4 r_lo = row_lo_cur, r_hi = row_hi_cur;
#pragma unroll
                            for (int g4 = 0; g4 < W; g4 += 4) {
                                const uint32_t *sg = sel + (g4 >> 2) * PQ;
                                uint32_t c4 = __builtin_amdgcn_perm(r_lo.y, r_lo.x, sg[0]);
                                if constexpr (PQ > 1) c4 |= __builtin_amdgcn_perm(r_lo.w, r_lo.z, sg[PQ > 1 ? 1 : 0]);
                                if constexpr (PQ > 2) c4 |= __builtin_amdgcn_perm(r_hi.y, r_hi.x, sg[PQ > 2 ? 2 : 0]);
                                if constexpr (PQ > 3) c4 |= __builtin_amdgcn_perm(r_hi.w, r_hi.z, sg[PQ > 3 ? 3 : 0]);
#pragma unroll
                                for (int i4 = 0; i4 < 4; ++i4) cell(g4 + i4, (int)(int8_t)(c4 >> (8 * i4)));
                            }
                        } else if constexpr (kMatrix) {
                            // Substitution scores are gathered from LDS a chunk of columns ahead of the DP chain, so
                            // the ds_read latency overlaps the dependent max/add chain instead of serialising with it.
                            // Columns and rows are only ever swapped for symmetric matrices (api.hip), where
                            // subs[row][col] == subs[col][row]: the lookup needs no orientation fix-up.
                            constexpr int kChunk = W < 8 ? W : (W % 8 == 0 ? 8 : (W % 6 == 0 ? 6 : (W % 5 == 0 ? 5 : (W % 7 == 0 ? 7 : 1))));
                            constexpr int kChunks = W / kChunk;
                            const int8_t *mrow = lmatrix + sym * kMatrixStride;
                            int sc_nxt[kChunk];
#pragma unroll
                            for (int q = 0; q < kChunk; ++q) sc_nxt[q] = mrow[col_sym(q)];
#pragma unroll
                            for (int chunk = 0; chunk < kChunks; ++chunk) {
                                int sc_cur[kChunk];
#pragma unroll
                                for (int q = 0; q < kChunk; ++q) sc_cur[q] = sc_nxt[q];
                                if (chunk + 1 < kChunks) {
#pragma unroll
                                    for (int q = 0; q < kChunk; ++q) sc_nxt[q] = mrow[col_sym((chunk + 1) * kChunk + q)];
                                }
                                // hipcc otherwise sinks every ds_read next to its use (lgkmcnt(0) per cell)
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int q = 0; q < kChunk; ++q) cell(chunk * kChunk + q, (kSkew || kSkewAffine) ? sc_cur[q] + bias2 : sc_cur[q]);
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < W; ++k) cell(k, (sym == col_sym(k)) ? match + bias2 : mismatch + bias2);
                        }
                        finish(W - 1, t_pending, t_pending);
                        out_h = left;
                        out_e = e;
                        if (write_bnd) {
                            bnd_h[s - gl + 1] = left;
                            if constexpr (kAffine) bnd_e[s - gl + 1] = e;
                        }
                    }
                    prev_h = recv_h;
                }
            }
            if constexpr (!kLocal) {
                // result lives in the lane/register holding column `cols`
                if (have && cols > c0 && cols <= c0 + G * W) {
                    uint32_t jj = cols - 1 - c0;
                    if ((uint32_t)gl == jj / W) {
                        uint32_t kk = jj % W;
#pragma unroll
                        for (int k = 0; k < W; ++k)
                            if ((uint32_t)k == kk) result = H[k];
                        if constexpr (kSkew || kSkewAffine) result += (int)(rows + cols) * ext;
                        if constexpr (kSkewAffine) result -= open - ext;   // the strip holds H^ + (open - ext)
                        store_score(args.job, p, result);
                    }
                }
            }
            if (passes > 1) __builtin_amdgcn_s_waitcnt(0);  // boundary stores land before the next pass reads
        }
        if constexpr (kLocal) {
            // columns past `cols` only ever see symbol 0 lookups; mask them out by construction: strips beyond
            // the pair's last column contributed cells too, so restrict the maximum to real columns
            int group_best = best;
#pragma unroll
            for (int off = 1; off < G; off <<= 1) {
                int other = __shfl_xor(group_best, off);
                group_best = other > group_best ? other : group_best;
            }
            if (have && gl == 0) store_result(args.job, p, (int64_t)group_best);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Host dispatch
// ------------------------------------------------------------------------------------------------
template <typename Sym, int G, int W, int MODEL, int PQ = 4>
static void launch_one(Scope *scope, const KernelArgs &args, uint32_t cls, uint32_t count, const char *name) {
    constexpr int kGroups = 64 / G;
    uint32_t chunks = (count + kGroups - 1) / kGroups;
    uint32_t blocks = (chunks + 3) / 4;
    size_t lds = (MODEL == kUniformLinear || MODEL == kUniformAffine) ? 0 : (MODEL == kClassLinear || MODEL == kClassAffine || MODEL == kClassLinearLocal || MODEL == kClassAffineLocal ? kClassLds : kMatrixLds);
    // persistent-ish grid: enough blocks to fill the chip several times over, waves stride over chunks
    uint32_t max_blocks = (uint32_t)scope->compute_units * (lds > 4096 ? 2 : 8);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) return;
    if (lds > 65536) opt_in_dynamic_lds(scope, (const void *)k_wavefront<Sym, G, W, MODEL, PQ>, lds);
    // alternate streams between class kernels (see Scope::wf_main); the side stream's kernels get their own boundary area
    KernelArgs launch_args = args;
    if (scope->wf_main) {
        const bool side = (scope->wf_toggle++ & 1u) != 0;
        scope->stream = side ? scope->side_stream : scope->wf_main;
        if (side && launch_args.boundary) launch_args.boundary += scope->wf_side_boundary;
    }
    StampGuard guard(scope, name);
    hipLaunchKernelGGL((k_wavefront<Sym, G, W, MODEL, PQ>), dim3(blocks), dim3(256), lds, scope->stream, launch_args, cls);
}

// Tuning knob (STRINGWARS_AMD_WF_CAP=<columns per lane>): strips wider than the cap run as several passes of 32
// columns per lane instead of one pass of up to 96 -- fewer registers per wave, more waves per SIMD.
int wavefront_strip_cap() {
    static int cap = [] { const char *e = test_hook("STRINGWARS_AMD_WF_CAP"); return e ? atoi(e) : 0; }();
    return cap;
}

template <typename Sym, int MODEL>
static void launch_model(Scope *scope, const KernelArgs &args, const Plan &plan) {
#define SWH_WF16(WV)                                                                                   \
    if (plan.class_count[kClassWf16 + WV - 1])                                                         \
        launch_one<Sym, 16, WV, MODEL>(scope, args, kClassWf16 + WV - 1, plan.class_count[kClassWf16 + WV - 1], \
                                       "wavefront_g16_w" #WV);
    SWH_WF16(1) SWH_WF16(2) SWH_WF16(3) SWH_WF16(4) SWH_WF16(5) SWH_WF16(6) SWH_WF16(7) SWH_WF16(8)
#undef SWH_WF16
#define SWH_WF64(IDX, WV)                                                                              \
    if (plan.class_count[kClassWf64 + IDX]) {                                                          \
        if (wavefront_strip_cap() && WV > wavefront_strip_cap())                                       \
            launch_one<Sym, 64, 32, MODEL>(scope, args, kClassWf64 + IDX, plan.class_count[kClassWf64 + IDX], \
                                           "wavefront_g64_w32_multipass");                           \
        else                                                                                           \
            launch_one<Sym, 64, WV, MODEL>(scope, args, kClassWf64 + IDX, plan.class_count[kClassWf64 + IDX], \
                                           "wavefront_g64_w" #WV);                                   \
    }
    SWH_WF64(0, 3) SWH_WF64(1, 4) SWH_WF64(2, 6) SWH_WF64(3, 8) SWH_WF64(4, 12) SWH_WF64(5, 16)
    SWH_WF64(6, 24) SWH_WF64(7, 32)
    if constexpr (MODEL != kMatrixAffine && MODEL != kMatrixAffineLocal && MODEL != kUniformAffine) {
        SWH_WF64(8, 48) SWH_WF64(9, 64) SWH_WF64(10, 80) SWH_WF64(11, 96)
    } else {
        // affine keeps two state rows per column: cap the strip at 32 columns, wider pairs go multi-pass
        for (int idx = 8; idx < kNumWideW; ++idx)
            if (plan.class_count[kClassWf64 + idx])
                launch_one<Sym, 64, 32, MODEL>(scope, args, kClassWf64 + idx, plan.class_count[kClassWf64 + idx],
                                               "wavefront_g64_w32_multipass");
    }
#undef SWH_WF64
    if (plan.class_count[kClassWfMulti])
        launch_one<Sym, 64, 32, MODEL>(scope, args, kClassWfMulti, plan.class_count[kClassWfMulti],
                                       "wavefront_g64_w32_multipass");
}

// Class-table models handle columns four at a time: strips are rounded up to a multiple of four columns.
template <int MODEL, int PQ>
static void launch_class_model(Scope *scope, const KernelArgs &args, const Plan &plan) {
    for (int wc = 1; wc <= 8; ++wc) {
        const uint32_t cls = kClassWf16 + wc - 1, count = plan.class_count[cls];
        if (!count) continue;
        if (wc <= 4) launch_one<uint8_t, 16, 4, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g16_w4");
        else launch_one<uint8_t, 16, 8, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g16_w8");
    }
    for (int idx = 0; idx < kNumWideW; ++idx) {
        const uint32_t cls = kClassWf64 + idx, count = plan.class_count[cls];
        if (!count) continue;
        const int w = wide_w(idx);
        if (w <= 4) launch_one<uint8_t, 64, 4, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w4");
        else if (w <= 8) launch_one<uint8_t, 64, 8, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w8");
        else if (w <= 12) launch_one<uint8_t, 64, 12, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w12");
        else if (w <= 16) launch_one<uint8_t, 64, 16, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w16");
        else if (w <= 24) launch_one<uint8_t, 64, 24, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w24");
        else if (w <= 32) launch_one<uint8_t, 64, 32, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w32");
        else if (w <= 48) launch_one<uint8_t, 64, 48, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w48");
        else if (w <= 64) launch_one<uint8_t, 64, 64, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w64");
        else if constexpr (MODEL == kClassLinear || MODEL == kClassLinearLocal) {
            if (w <= 80) launch_one<uint8_t, 64, 80, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w80");
            else launch_one<uint8_t, 64, 96, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w96");
        } else {
            // Gotoh keeps H, F and the selectors per column (3 registers): two exact half-width passes beat one
            // pass that spills or a wider pass whose second half idles
            if (w <= 80) launch_one<uint8_t, 64, 40, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w40_x2");
            else launch_one<uint8_t, 64, 48, MODEL, PQ>(scope, args, cls, count, "wavefront_class_g64_w48_x2");
        }
    }
    if (plan.class_count[kClassWfMulti])
        launch_one<uint8_t, 64, 32, MODEL, PQ>(scope, args, kClassWfMulti, plan.class_count[kClassWfMulti],
                                           "wavefront_class_g64_w32_multipass");
}

static void launch_wavefront_classes(Scope *scope, const KernelArgs &args, const Plan &plan);

void launch_wavefront(Scope *scope, const KernelArgs &args, const Plan &plan) {
    // fork: the side stream starts where the main stream is now; join: the main stream waits for the side stream
    hipStream_t main_stream = scope->stream;
    SWH_HIP_CHECK(hipEventRecord(scope->fork_ev, main_stream));
    SWH_HIP_CHECK(hipStreamWaitEvent(scope->side_stream, scope->fork_ev, 0));
    scope->wf_main = main_stream;
    scope->wf_toggle = 0;
    try {
        launch_wavefront_classes(scope, args, plan);
    } catch (...) {
        scope->stream = main_stream;
        scope->wf_main = nullptr;
        throw;
    }
    scope->stream = main_stream;
    scope->wf_main = nullptr;
    SWH_HIP_CHECK(hipEventRecord(scope->join_ev, scope->side_stream));
    SWH_HIP_CHECK(hipStreamWaitEvent(main_stream, scope->join_ev, 0));
}

static void launch_wavefront_classes(Scope *scope, const KernelArgs &args, const Plan &plan) {
    bool matrix = args.scoring.matrix != nullptr;
    if (args.sym_bytes == 4) {
        // code points: uniform match / mismatch costs only (`LevenshteinDistancesUtf8::new(&scope, m, x, o, e)`, bench.rs:386-389)
        if (!args.affine) launch_model<uint32_t, kUniformLinear>(scope, args, plan);
        else launch_model<uint32_t, kUniformAffine>(scope, args, plan);
    } else if (!matrix) {
        if (!args.affine) launch_model<uint8_t, kUniformLinear>(scope, args, plan);
        else launch_model<uint8_t, kUniformAffine>(scope, args, plan);
    } else if (args.scoring.class_table && args.local) {
        // Smith-Waterman on the class tables (the engine uploaded cost rows without a bias for it)
        const uint32_t classes = args.scoring.classes ? args.scoring.classes : 32;
        if (!args.affine) {
            if (classes <= 8) launch_class_model<kClassLinearLocal, 1>(scope, args, plan);
            else if (classes <= 24) launch_class_model<kClassLinearLocal, 3>(scope, args, plan);
            else launch_class_model<kClassLinearLocal, 4>(scope, args, plan);
        } else {
            if (classes <= 8) launch_class_model<kClassAffineLocal, 1>(scope, args, plan);
            else if (classes <= 24) launch_class_model<kClassAffineLocal, 3>(scope, args, plan);
            else launch_class_model<kClassAffineLocal, 4>(scope, args, plan);
        }
    } else if (args.scoring.class_table) {
        // one v_perm per group of four columns and per 8 classes the matrix distinguishes
        const uint32_t classes = args.scoring.classes ? args.scoring.classes : 32;
        if (!args.affine) {
            if (classes <= 8) launch_class_model<kClassLinear, 1>(scope, args, plan);
            else if (classes <= 24) launch_class_model<kClassLinear, 3>(scope, args, plan);
            else launch_class_model<kClassLinear, 4>(scope, args, plan);
        } else {
            if (classes <= 8) launch_class_model<kClassAffine, 1>(scope, args, plan);
            else if (classes <= 24) launch_class_model<kClassAffine, 3>(scope, args, plan);
            else launch_class_model<kClassAffine, 4>(scope, args, plan);
        }
    } else if (args.local) {
        if (!args.affine) launch_model<uint8_t, kMatrixLinearLocal>(scope, args, plan);
        else launch_model<uint8_t, kMatrixAffineLocal>(scope, args, plan);
    } else if (!args.affine) {
        launch_model<uint8_t, kMatrixLinear>(scope, args, plan);
    } else {
        launch_model<uint8_t, kMatrixAffine>(scope, args, plan);
    }
    SWH_HIP_CHECK(hipGetLastError());
}

}  // namespace swh
