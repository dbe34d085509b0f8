/*
 * oracle.c -- CPU restatement of the edit-distance / alignment-score definitions the
 * StringWars `similarities/` benchmark relies on.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only `tests/`, `__graft_entry__.smoke()`
 * and the `cpu_baseline` leg of `bench.py` may load it. The shipped library
 * (`stringwars_amd/csrc`) never links, includes or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference (`/root/reference`, StringWars v1.8.3) contains no arithmetic
 * for this path and no tests. The numbers are produced by un-vendored crates:
 *   rapidfuzz 0.5.0  (Cargo.lock:3939-3942)  -- called at similarities/bench.rs:416-419, :437
 *   bio 4.0.1        (Cargo.lock:510-513)    -- called at similarities/bench.rs:455, :746-765
 *   stringzilla 5.0.1 (Cargo.lock:4678-4685) -- called at similarities/bench.rs:382-399, :658-670
 * None of them can be built or imported here (no cargo, no wheels, no network). What this file
 * restates is therefore their *published definitions*, anchored on the reference's call sites:
 *   - unit-cost Levenshtein, match 0 / mismatch 1 / open 1 / extend 1 (bench.rs:330, :382);
 *     symbols are bytes (bench.rs:413) or Unicode scalar values (bench.rs:434);
 *   - bounded Levenshtein: out = min(d, k+1), rapidfuzz's score_cutoff convention;
 *   - Needleman-Wunsch global score, max-plus, substitution matrix + gaps with
 *     gap(k) = open + (k-1)*extend (bench.rs:7, :342, :353).
 * The Levenshtein distance of two sequences is a uniquely defined integer, so any correct
 * implementation is bit-exact with rapidfuzz; correctness is pinned by (a) the textbook
 * known-answer table in tests/golden/kat.json, (b) agreement of two independent algorithms in
 * this file (Wagner-Fischer DP vs Hyyro/Myers bit-parallel), (c) metric properties.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* UTF-8 -> Unicode scalar values. Mirrors Rust `str::chars()` (bench.rs:434-437): strict     */
/* UTF-8, no surrogates, no overlongs, max U+10FFFF. Returns the count or -1 when invalid.     */
/* ------------------------------------------------------------------------------------------ */
ORC_API long orc_utf8_decode(const uint8_t *s, size_t n, uint32_t *out) {
    size_t i = 0;
    long count = 0;
    while (i < n) {
        uint32_t c = s[i];
        uint32_t cp;
        size_t need;
        if (c < 0x80) { cp = c; need = 0; }
        else if (c >= 0xC2 && c <= 0xDF) { cp = c & 0x1F; need = 1; }
        else if (c >= 0xE0 && c <= 0xEF) { cp = c & 0x0F; need = 2; }
        else if (c >= 0xF0 && c <= 0xF4) { cp = c & 0x07; need = 3; }
        else return -1;
        for (size_t k = 1; k <= need; ++k) {
            if (i + k >= n) return -1;
            uint32_t cc = s[i + k];
            if ((cc & 0xC0) != 0x80) return -1;
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return -1;
        if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return -1;
        if (out) out[count] = cp;
        ++count;
        i += need + 1;
    }
    return count;
}

/* ------------------------------------------------------------------------------------------ */
/* Wagner-Fischer, two rows. The definition (bench.rs:330: match=0, mismatch=1, gap=1).        */
/* ------------------------------------------------------------------------------------------ */
#define DEFINE_WF(NAME, SYM)                                                                    \
    ORC_API uint32_t NAME(const SYM *a, size_t la, const SYM *b, size_t lb) {                   \
        if (la == 0) return (uint32_t)lb;                                                       \
        if (lb == 0) return (uint32_t)la;                                                       \
        uint32_t *row = (uint32_t *)malloc((lb + 1) * sizeof(uint32_t));                        \
        for (size_t j = 0; j <= lb; ++j) row[j] = (uint32_t)j;                                  \
        for (size_t i = 1; i <= la; ++i) {                                                      \
            uint32_t diag = row[0];                                                             \
            row[0] = (uint32_t)i;                                                               \
            for (size_t j = 1; j <= lb; ++j) {                                                  \
                uint32_t up = row[j];                                                           \
                uint32_t sub = diag + (a[i - 1] != b[j - 1]);                                   \
                uint32_t best = up + 1 < row[j - 1] + 1 ? up + 1 : row[j - 1] + 1;              \
                row[j] = sub < best ? sub : best;                                               \
                diag = up;                                                                      \
            }                                                                                   \
        }                                                                                       \
        uint32_t d = row[lb];                                                                   \
        free(row);                                                                              \
        return d;                                                                               \
    }
DEFINE_WF(orc_lev_bytes, uint8_t)
DEFINE_WF(orc_lev_u32, uint32_t)

/* General-cost edit distance (min-plus Gotoh), the parameterisation of                        */
/* `LevenshteinDistances::new(&scope, match, mismatch, open, extend)` (bench.rs:382).          */
/* gap(k) = open + (k-1)*extend.                                                               */
ORC_API int64_t orc_lev_costs_bytes(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, int match,
                                    int mismatch, int open, int extend) {
    const int64_t INF = INT64_MAX / 4;
    int64_t *H = (int64_t *)malloc((lb + 1) * sizeof(int64_t));
    int64_t *F = (int64_t *)malloc((lb + 1) * sizeof(int64_t)); /* gap in `a` direction (vertical) */
    H[0] = 0;
    for (size_t j = 1; j <= lb; ++j) { H[j] = open + (int64_t)(j - 1) * extend; F[j] = INF; }
    F[0] = INF;
    for (size_t i = 1; i <= la; ++i) {
        int64_t diag = H[0];
        H[0] = open + (int64_t)(i - 1) * extend;
        int64_t E = INF; /* horizontal gap */
        for (size_t j = 1; j <= lb; ++j) {
            int64_t up = H[j];
            int64_t f1 = up + open, f2 = F[j] + extend;
            F[j] = f1 < f2 ? f1 : f2;
            int64_t e1 = H[j - 1] + open, e2 = E + extend;
            E = e1 < e2 ? e1 : e2;
            int64_t s = diag + (a[i - 1] == b[j - 1] ? match : mismatch);
            int64_t best = s < E ? s : E;
            best = best < F[j] ? best : F[j];
            H[j] = best;
            diag = up;
        }
    }
    int64_t r = H[lb];
    free(H); free(F);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Needleman-Wunsch / Gotoh global alignment SCORE (max-plus), 256x256 i8 substitution matrix, */
/* gap(k) = open + (k-1)*extend (both normally negative). Shape of                             */
/* `NeedlemanWunschScores::new(&scope, &byte_to_class, &class_costs, open, extend)`            */
/* (bench.rs:658-662) with the class table expanded to bytes.                                  */
/* ------------------------------------------------------------------------------------------ */
ORC_API int64_t orc_nw_score(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, const int8_t *subs,
                             int open, int extend) {
    const int64_t NINF = INT64_MIN / 4;
    int64_t *H = (int64_t *)malloc((lb + 1) * sizeof(int64_t));
    int64_t *F = (int64_t *)malloc((lb + 1) * sizeof(int64_t));
    H[0] = 0; F[0] = NINF;
    for (size_t j = 1; j <= lb; ++j) { H[j] = open + (int64_t)(j - 1) * extend; F[j] = NINF; }
    for (size_t i = 1; i <= la; ++i) {
        int64_t diag = H[0];
        H[0] = open + (int64_t)(i - 1) * extend;
        int64_t E = NINF;
        const int8_t *srow = subs + (size_t)a[i - 1] * 256;
        for (size_t j = 1; j <= lb; ++j) {
            int64_t up = H[j];
            int64_t f1 = up + open, f2 = F[j] + extend;
            F[j] = f1 > f2 ? f1 : f2;
            int64_t e1 = H[j - 1] + open, e2 = E + extend;
            E = e1 > e2 ? e1 : e2;
            int64_t s = diag + srow[b[j - 1]];
            int64_t best = s > E ? s : E;
            best = best > F[j] ? best : F[j];
            H[j] = best;
            diag = up;
        }
    }
    int64_t r = H[lb];
    free(H); free(F);
    return r;
}

/* Smith-Waterman local score (kept for the SURVEY 8f row; same conventions). */
ORC_API int64_t orc_sw_score(const uint8_t *a, size_t la, const uint8_t *b, size_t lb, const int8_t *subs,
                             int open, int extend) {
    const int64_t NINF = INT64_MIN / 4;
    int64_t *H = (int64_t *)calloc(lb + 1, sizeof(int64_t));
    int64_t *F = (int64_t *)malloc((lb + 1) * sizeof(int64_t));
    for (size_t j = 0; j <= lb; ++j) F[j] = NINF;
    int64_t best_all = 0;
    for (size_t i = 1; i <= la; ++i) {
        int64_t diag = H[0];
        int64_t E = NINF;
        const int8_t *srow = subs + (size_t)a[i - 1] * 256;
        for (size_t j = 1; j <= lb; ++j) {
            int64_t up = H[j];
            int64_t f1 = up + open, f2 = F[j] + extend;
            F[j] = f1 > f2 ? f1 : f2;
            int64_t e1 = H[j - 1] + open, e2 = E + extend;
            E = e1 > e2 ? e1 : e2;
            int64_t s = diag + srow[b[j - 1]];
            int64_t best = s > E ? s : E;
            best = best > F[j] ? best : F[j];
            if (best < 0) best = 0;
            H[j] = best;
            if (best > best_all) best_all = best;
            diag = up;
        }
    }
    free(H); free(F);
    return best_all;
}

/* ------------------------------------------------------------------------------------------ */
/* Second, independent implementation: Hyyro 2003 / Myers 1999 bit-parallel Levenshtein with   */
/* 64-bit blocks (the algorithm family rapidfuzz uses). Also the timed CPU       */
/* baseline of bench.py (`cpu_baseline.kind = "port"`).                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t *peq; size_t nblk; } orc_peq_t;

static uint32_t hyyro_single(const uint8_t *p, size_t m, const uint8_t *t, size_t n) {
    uint64_t peq[256];
    memset(peq, 0, sizeof peq);
    for (size_t i = 0; i < m; ++i) peq[p[i]] |= 1ull << i;
    uint64_t pv = ~0ull, mv = 0, last = 1ull << (m - 1);
    uint32_t score = (uint32_t)m;
    for (size_t j = 0; j < n; ++j) {
        uint64_t eq = peq[t[j]];
        uint64_t xv = eq | mv;
        uint64_t xh = (((eq & pv) + pv) ^ pv) | eq;
        uint64_t ph = mv | ~(xh | pv);
        uint64_t mh = pv & xh;
        score += (ph & last) != 0;
        score -= (mh & last) != 0;
        ph = (ph << 1) | 1;
        mh <<= 1;
        pv = mh | ~(xv | ph);
        mv = ph & xv;
    }
    return score;
}

static uint32_t hyyro_blocks(const uint8_t *p, size_t m, const uint8_t *t, size_t n) {
    size_t nblk = (m + 63) / 64;
    uint64_t *peq = (uint64_t *)calloc(256 * nblk, sizeof(uint64_t));
    uint64_t *pv = (uint64_t *)malloc(nblk * sizeof(uint64_t));
    uint64_t *mv = (uint64_t *)calloc(nblk, sizeof(uint64_t));
    for (size_t i = 0; i < m; ++i) peq[(size_t)p[i] * nblk + i / 64] |= 1ull << (i % 64);
    for (size_t k = 0; k < nblk; ++k) pv[k] = ~0ull;
    uint64_t last = 1ull << ((m - 1) % 64);
    uint32_t score = (uint32_t)m;
    for (size_t j = 0; j < n; ++j) {
        uint64_t ph_carry = 1, mh_carry = 0;
        const uint64_t *eqrow = peq + (size_t)t[j] * nblk;
        for (size_t k = 0; k < nblk; ++k) {
            uint64_t eq = eqrow[k];
            uint64_t xv = eq | mv[k];
            eq |= mh_carry;
            uint64_t xh = (((eq & pv[k]) + pv[k]) ^ pv[k]) | eq;
            uint64_t ph = mv[k] | ~(xh | pv[k]);
            uint64_t mh = pv[k] & xh;
            if (k == nblk - 1) {
                score += (ph & last) != 0;
                score -= (mh & last) != 0;
            }
            uint64_t ph_out = ph >> 63, mh_out = mh >> 63;
            ph = (ph << 1) | ph_carry;
            mh = (mh << 1) | mh_carry;
            ph_carry = ph_out; mh_carry = mh_out;
            pv[k] = mh | ~(xv | ph);
            mv[k] = ph & xv;
        }
    }
    free(peq); free(pv); free(mv);
    return score;
}

ORC_API uint32_t orc_hyyro_bytes(const uint8_t *a, size_t la, const uint8_t *b, size_t lb) {
    if (la == 0) return (uint32_t)lb;
    if (lb == 0) return (uint32_t)la;
    /* pattern = shorter string */
    if (la > lb) { const uint8_t *t = a; a = b; b = t; size_t l = la; la = lb; lb = l; }
    return la <= 64 ? hyyro_single(a, la, b, lb) : hyyro_blocks(a, la, b, lb);
}

/* ------------------------------------------------------------------------------------------ */
/* Second, independent alignment scorer: Waterman-Smith-Beyer 1976 general-gap DP over the FULL */
/* (la+1) x (lb+1) table, O(la*lb*(la+lb)). No E/F state, no rolling rows: every cell looks back  */
/* over every gap length k with gap(k) = open + (k-1)*extend spelled out. Shares nothing with    */
/* orc_nw_score / orc_sw_score but the definition; tests cross-check the two on >= 1e5 cases.    */
/* `local` != 0 gives the Smith-Waterman score (floor 0, maximum over all cells).               */
/* ------------------------------------------------------------------------------------------ */
ORC_API int64_t orc_align_score_general(const uint8_t *a, size_t la, const uint8_t *b, size_t lb,
                                        const int8_t *subs, int open, int extend, int local) {
    const size_t w = lb + 1;
    int64_t *T = (int64_t *)malloc((la + 1) * w * sizeof(int64_t));
    int64_t best_all = 0;
    for (size_t i = 0; i <= la; ++i) {
        for (size_t j = 0; j <= lb; ++j) {
            int64_t v;
            if (i == 0 && j == 0) v = 0;
            else {
                v = INT64_MIN / 4;
                if (i && j) {
                    int64_t d = T[(i - 1) * w + (j - 1)] + subs[(size_t)a[i - 1] * 256 + b[j - 1]];
                    if (d > v) v = d;
                }
                for (size_t k = 1; k <= i; ++k) {   /* k symbols of `a` against a gap */
                    int64_t g = T[(i - k) * w + j] + open + (int64_t)(k - 1) * extend;
                    if (g > v) v = g;
                }
                for (size_t k = 1; k <= j; ++k) {   /* k symbols of `b` against a gap */
                    int64_t g = T[i * w + (j - k)] + open + (int64_t)(k - 1) * extend;
                    if (g > v) v = g;
                }
            }
            if (local && v < 0) v = 0;
            if (v > best_all) best_all = v;
            T[i * w + j] = v;
        }
    }
    int64_t r = local ? best_all : T[la * w + lb];
    free(T);
    return r;
}

/* Third Levenshtein implementation (independent of both above): the full (la+1) x (lb+1) table filled  */
/* anti-diagonal by anti-diagonal -- the GPU kernels' traversal order, on the CPU.                     */
ORC_API uint32_t orc_lev_antidiagonal(const uint8_t *a, size_t la, const uint8_t *b, size_t lb) {
    const size_t w = lb + 1;
    uint32_t *T = (uint32_t *)malloc((la + 1) * w * sizeof(uint32_t));
    for (size_t d = 0; d <= la + lb; ++d) {
        size_t i_lo = d > lb ? d - lb : 0, i_hi = d < la ? d : la;
        for (size_t i = i_lo; i <= i_hi; ++i) {
            size_t j = d - i;
            uint32_t v;
            if (i == 0) v = (uint32_t)j;
            else if (j == 0) v = (uint32_t)i;
            else {
                uint32_t s = T[(i - 1) * w + j - 1] + (a[i - 1] != b[j - 1]);
                uint32_t u = T[(i - 1) * w + j] + 1, l = T[i * w + j - 1] + 1;
                v = s < u ? s : u;
                v = v < l ? v : l;
            }
            T[i * w + j] = v;
        }
    }
    uint32_t r = T[la * w + lb];
    free(T);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Self-checks run by tests/test_oracle.py in a C loop (SURVEY 8c "beyond KATs" (i)): the        */
/* implementations above must agree on seeded random inputs. Return the number of disagreeing   */
/* cases; *first_bad receives the index of the first one (or -1); *cells the DP cells covered.  */
/* ------------------------------------------------------------------------------------------ */
static uint64_t orc_mix(uint64_t *state) {   /* SplitMix64 */
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static size_t orc_pick_length(uint64_t *rng, size_t max_len) {
    static const uint16_t edges[] = {0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300};
    uint64_t r = orc_mix(rng) % 100;
    size_t len;
    if (r < 55) len = orc_mix(rng) % 41;                 /* words */
    else if (r < 80) len = 40 + orc_mix(rng) % 101;      /* tokens */
    else if (r < 92) len = edges[orc_mix(rng) % (sizeof edges / sizeof edges[0])];   /* 64-bit word boundaries */
    else len = orc_mix(rng) % (max_len + 1);
    return len > max_len ? max_len : len;
}
static void orc_random_pair(uint64_t *rng, uint32_t alphabet, size_t max_len, uint8_t *a, size_t *la, uint8_t *b,
                            size_t *lb) {
    *la = orc_pick_length(rng, max_len);
    for (size_t i = 0; i < *la; ++i) a[i] = (uint8_t)(orc_mix(rng) % alphabet);
    if (orc_mix(rng) & 1) {   /* related: a with a few edits */
        size_t n = *la;
        memcpy(b, a, n);
        uint64_t edits = orc_mix(rng) % 9;
        for (uint64_t e = 0; e < edits; ++e) {
            uint64_t op = orc_mix(rng) % 3;
            if (op == 0 && n) b[orc_mix(rng) % n] = (uint8_t)(orc_mix(rng) % alphabet);
            else if (op == 1 && n < max_len) {
                size_t pos = orc_mix(rng) % (n + 1);
                memmove(b + pos + 1, b + pos, n - pos);
                b[pos] = (uint8_t)(orc_mix(rng) % alphabet);
                ++n;
            } else if (op == 2 && n) {
                size_t pos = orc_mix(rng) % n;
                memmove(b + pos, b + pos + 1, n - pos - 1);
                --n;
            }
        }
        *lb = n;
    } else {
        *lb = orc_pick_length(rng, max_len);
        for (size_t i = 0; i < *lb; ++i) b[i] = (uint8_t)(orc_mix(rng) % alphabet);
    }
}

/* Wagner-Fischer vs Hyyro bit-parallel (and, every 16th case, the anti-diagonal table) on `cases` pairs over   */
/* `alphabet` symbols with lengths 0..max_len, incl. the 63/64/65 ... word boundaries.                            */
ORC_API long orc_selfcheck_levenshtein(uint64_t seed, size_t cases, uint32_t alphabet, size_t max_len,
                                       long *first_bad, uint64_t *cells) {
    uint8_t *a = (uint8_t *)malloc(max_len + 1), *b = (uint8_t *)malloc(max_len + 1);
    uint64_t rng = seed * 0xD1342543DE82EF95ull + alphabet;
    long bad = 0;
    *first_bad = -1;
    *cells = 0;
    for (size_t c = 0; c < cases; ++c) {
        size_t la, lb;
        orc_random_pair(&rng, alphabet, max_len, a, &la, b, &lb);
        uint32_t wf = orc_lev_bytes(a, la, b, lb), hy = orc_hyyro_bytes(a, la, b, lb);
        int ok = wf == hy && orc_hyyro_bytes(b, lb, a, la) == wf;
        if ((c & 15) == 0) ok = ok && orc_lev_antidiagonal(a, la, b, lb) == wf;
        *cells += (uint64_t)la * lb;
        if (!ok) { if (!bad) *first_bad = (long)c; ++bad; }
    }
    free(a); free(b);
    return bad;
}

/* Gotoh (orc_nw_score / orc_sw_score) vs the general-gap table on random strings over `alphabet` symbols,     */
/* random i8 matrices (every 3rd asymmetric) and random gaps with |open| >= |extend| (the reference's          */
/* settings, bench.rs:342, :353). With a gap that is cheaper to open than to extend the two differ by          */
/* definition: Gotoh's boundary row / column is ONE gap of length i, the general-gap table may split it.        */
ORC_API long orc_selfcheck_alignment(uint64_t seed, size_t cases, uint32_t alphabet, size_t max_len,
                                     long *first_bad, uint64_t *cells) {
    uint8_t *a = (uint8_t *)malloc(max_len + 1), *b = (uint8_t *)malloc(max_len + 1);
    int8_t *subs = (int8_t *)malloc(65536);
    uint64_t rng = seed * 0xA0761D6478BD642Full + alphabet;
    long bad = 0;
    *first_bad = -1;
    *cells = 0;
    for (size_t c = 0; c < cases; ++c) {
        if (c % 64 == 0) {   /* a fresh matrix every 64 cases */
            int asym = (c / 64) % 3 == 2;
            for (uint32_t i = 0; i < alphabet; ++i)
                for (uint32_t j = 0; j <= i; ++j) {
                    int8_t v = (int8_t)(i == j ? (int)(orc_mix(&rng) % 12) : (int)(orc_mix(&rng) % 12) - 8);
                    subs[i * 256 + j] = v;
                    subs[j * 256 + i] = asym ? (int8_t)((int)(orc_mix(&rng) % 16) - 8) : v;
                }
        }
        size_t la, lb;
        orc_random_pair(&rng, alphabet, max_len, a, &la, b, &lb);
        const int extend = -(int)(orc_mix(&rng) % 6), open = extend - (int)(orc_mix(&rng) % 12);
        int ok = orc_nw_score(a, la, b, lb, subs, open, extend) == orc_align_score_general(a, la, b, lb, subs, open, extend, 0);
        ok = ok && orc_sw_score(a, la, b, lb, subs, open, extend) == orc_align_score_general(a, la, b, lb, subs, open, extend, 1);
        *cells += (uint64_t)la * lb;
        if (!ok) { if (!bad) *first_bad = (long)c; ++bad; }
    }
    free(a); free(b); free(subs);
    return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* Tape (Arrow-style: data + count+1 offsets) batch drivers. `offset_width` is 4 or 8 bytes.   */
/* Return 0, or -(i+1) when pair i holds invalid UTF-8.                                        */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t off_at(const void *offs, int width, size_t i) {
    return width == 8 ? ((const uint64_t *)offs)[i] : ((const uint32_t *)offs)[i];
}

/* algo: 0 = Wagner-Fischer, 1 = Hyyro bit-parallel (bytes only). bound: UINT32_MAX = unbounded */
ORC_API long orc_lev_pairs(const uint8_t *da, const void *oa, const uint8_t *db, const void *ob, int width,
                           size_t first, size_t count, int utf8, int algo, uint32_t bound, uint32_t *out) {
    uint32_t *ca = NULL, *cb = NULL;
    size_t cap_a = 0, cap_b = 0;
    for (size_t i = first; i < first + count; ++i) {
        uint64_t a0 = off_at(oa, width, i), a1 = off_at(oa, width, i + 1);
        uint64_t b0 = off_at(ob, width, i), b1 = off_at(ob, width, i + 1);
        size_t la = (size_t)(a1 - a0), lb = (size_t)(b1 - b0);
        uint32_t d;
        if (utf8) {
            if (la > cap_a) { cap_a = la * 2 + 16; ca = (uint32_t *)realloc(ca, cap_a * 4); }
            if (lb > cap_b) { cap_b = lb * 2 + 16; cb = (uint32_t *)realloc(cb, cap_b * 4); }
            if (!ca) { cap_a = 16; ca = (uint32_t *)malloc(64); }
            if (!cb) { cap_b = 16; cb = (uint32_t *)malloc(64); }
            long na = orc_utf8_decode(da + a0, la, ca);
            long nb = orc_utf8_decode(db + b0, lb, cb);
            if (na < 0 || nb < 0) { free(ca); free(cb); return -(long)(i + 1); }
            d = orc_lev_u32(ca, (size_t)na, cb, (size_t)nb);
        } else {
            d = algo == 1 ? orc_hyyro_bytes(da + a0, la, db + b0, lb) : orc_lev_bytes(da + a0, la, db + b0, lb);
        }
        if (bound != UINT32_MAX && d > bound) d = bound + 1;
        out[i] = d;
    }
    free(ca); free(cb);
    return 0;
}

ORC_API long orc_nw_pairs(const uint8_t *da, const void *oa, const uint8_t *db, const void *ob, int width,
                          size_t first, size_t count, const int8_t *subs, int open, int extend, int64_t *out) {
    for (size_t i = first; i < first + count; ++i) {
        uint64_t a0 = off_at(oa, width, i), a1 = off_at(oa, width, i + 1);
        uint64_t b0 = off_at(ob, width, i), b1 = off_at(ob, width, i + 1);
        out[i] = orc_nw_score(da + a0, (size_t)(a1 - a0), db + b0, (size_t)(b1 - b0), subs, open, extend);
    }
    return 0;
}

ORC_API long orc_lev_costs_pairs(const uint8_t *da, const void *oa, const uint8_t *db, const void *ob, int width,
                                 size_t first, size_t count, int match, int mismatch, int open, int extend,
                                 int64_t *out) {
    for (size_t i = first; i < first + count; ++i) {
        uint64_t a0 = off_at(oa, width, i), a1 = off_at(oa, width, i + 1);
        uint64_t b0 = off_at(ob, width, i), b1 = off_at(ob, width, i + 1);
        out[i] = orc_lev_costs_bytes(da + a0, (size_t)(a1 - a0), db + b0, (size_t)(b1 - b0), match, mismatch,
                                     open, extend);
    }
    return 0;
}

/* Symbol-length products, the reference's CUPS accounting (bench.rs:413, :434, :216-247):     */
/* cells += len_s(a_i) * len_s(b_i). Returns the sum; utf8 counts code points.                 */
ORC_API uint64_t orc_cells(const uint8_t *da, const void *oa, const uint8_t *db, const void *ob, int width,
                           size_t count, int utf8) {
    uint64_t cells = 0;
    for (size_t i = 0; i < count; ++i) {
        uint64_t a0 = off_at(oa, width, i), a1 = off_at(oa, width, i + 1);
        uint64_t b0 = off_at(ob, width, i), b1 = off_at(ob, width, i + 1);
        uint64_t la = a1 - a0, lb = b1 - b0;
        if (utf8) {
            long na = orc_utf8_decode(da + a0, (size_t)la, NULL), nb = orc_utf8_decode(db + b0, (size_t)lb, NULL);
            la = na < 0 ? 0 : (uint64_t)na; lb = nb < 0 ? 0 : (uint64_t)nb;
        }
        cells += la * lb;
    }
    return cells;
}
