// synth.cpp -- deterministic synthetic workloads (BASELINE.json configs C1..C5) and the
// pure helper functions of the reference harness (utils.rs / similarities/bench.rs). Host only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/stringwars_amd_harness.h"

#define SWH_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// SplitMix64 keyed by (seed, pair index): every pair is an independent stream.
struct Rng {
    uint64_t state;
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t index) : state(mix(seed ^ 0x5851F42D4C957F2Dull) ^ mix(index + 0x9E3779B97F4A7C15ull)) {}
    uint64_t next() { state += 0x9E3779B97F4A7C15ull; return mix(state); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }  // [0, n)
    uint32_t range(uint32_t lo, uint32_t hi) { return lo + below(hi - lo + 1); }              // [lo, hi]
};

typedef std::vector<uint32_t> Symbols;

static uint32_t draw_symbol(int workload, Rng &rng) {
    switch (workload) {
    case swh_workload_words16_k:
    case swh_workload_short_words_k: return 'a' + rng.below(26);
    case swh_workload_tokens64_k: return 0x21 + rng.below(0x7E - 0x21 + 1);
    case swh_workload_protein4k_k: return (uint32_t)"ACDEFGHIKLMNPQRSTVWY"[rng.below(20)];
    case swh_workload_bytes4k_k: return rng.below(256);
    case swh_workload_utf8_lines_k: {
        uint32_t r = rng.below(100);
        if (r < 40) return 0x20 + rng.below(0x7E - 0x20 + 1);
        if (r < 70) return 0x0400 + rng.below(0x100);
        if (r < 95) return 0x4E00 + rng.below(0x9FFF - 0x4E00 + 1);
        return 0x1F600 + rng.below(0x50);
    }
    }
    return 'x';
}

static void mutate(int workload, Rng &rng, Symbols &s, uint32_t edits) {
    for (uint32_t e = 0; e < edits; ++e) {
        uint32_t op = rng.below(3);
        if (op == 0 && !s.empty()) s[rng.below((uint32_t)s.size())] = draw_symbol(workload, rng);
        else if (op == 1) s.insert(s.begin() + rng.below((uint32_t)s.size() + 1), draw_symbol(workload, rng));
        else if (s.size() > 1) s.erase(s.begin() + rng.below((uint32_t)s.size()));
    }
}

static void encode(const Symbols &s, bool utf8, std::vector<uint8_t> &out) {
    for (uint32_t cp : s) {
        if (!utf8 || cp < 0x80) out.push_back((uint8_t)cp);
        else if (cp < 0x800) { out.push_back(0xC0 | (cp >> 6)); out.push_back(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) {
            out.push_back(0xE0 | (cp >> 12)); out.push_back(0x80 | ((cp >> 6) & 0x3F)); out.push_back(0x80 | (cp & 0x3F));
        } else {
            out.push_back(0xF0 | (cp >> 18)); out.push_back(0x80 | ((cp >> 12) & 0x3F));
            out.push_back(0x80 | ((cp >> 6) & 0x3F)); out.push_back(0x80 | (cp & 0x3F));
        }
    }
}

static uint32_t utf8_len(uint32_t cp) { return cp < 0x80 ? 1 : cp < 0x800 ? 2 : cp < 0x10000 ? 3 : 4; }

static uint32_t draw_length(int workload, Rng &rng) {
    switch (workload) {
    case swh_workload_words16_k: return rng.range(1, 16);
    case swh_workload_tokens64_k: return rng.range(32, 96);
    case swh_workload_protein4k_k:
    case swh_workload_bytes4k_k: return rng.range(3072, 5120);
    case swh_workload_short_words_k: {
        uint32_t len = 1 + rng.below(6) + rng.below(6);
        if (rng.below(8) == 0) len += rng.below(6);
        return len > 16 ? 16 : len;
    }
    }
    return 1;
}

static void fill(int workload, Rng &rng, Symbols &s, uint32_t len) {
    s.resize(len);
    for (uint32_t i = 0; i < len; ++i) s[i] = draw_symbol(workload, rng);
}

// One line of swh_workload_script_lines_k: letters of one script, ASCII spaces / punctuation / digits between them.
static void script_line(Rng &rng, Symbols &s) {
    static const struct { uint32_t first, count; } scripts[5][2] = {
        {{0x61, 26}, {0x41, 26}},        // Latin
        {{0x430, 32}, {0x410, 32}},      // Cyrillic
        {{0x3B1, 25}, {0x391, 25}},      // Greek
        {{0x621, 42}, {0x621, 42}},      // Arabic
        {{0x905, 53}, {0x93E, 16}},      // Devanagari: letters, vowel signs
    };
    static const char common[] = "            .,;:!?-()\"'0123456789";
    const uint32_t script = rng.below(5), len = rng.range(700, 1300);
    s.resize(len);
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t r = rng.below(100);
        if (r < 20) s[i] = (uint32_t)common[rng.below(sizeof common - 1)];
        else {
            const auto &range = scripts[script][r < 30];   // (an eighth of the letters from the second range: capitals, vowel signs)
            s[i] = range.first + rng.below(range.count);
        }
    }
}

static void make_pair(int workload, uint64_t seed, uint64_t index, Symbols &a, Symbols &b) {
    Rng rng(seed, index);
    a.clear(); b.clear();
    if (workload == swh_workload_script_lines_k) {
        script_line(rng, a);
        script_line(rng, b);
        return;
    }
    if (workload == swh_workload_utf8_lines_k) {
        uint32_t target = rng.range(768, 1280), bytes = 0;
        while (bytes < target) { uint32_t cp = draw_symbol(workload, rng); a.push_back(cp); bytes += utf8_len(cp); }
        b = a;
        mutate(workload, rng, b, rng.range(0, 64));
        return;
    }
    fill(workload, rng, a, draw_length(workload, rng));
    bool related = true;
    if (workload == swh_workload_words16_k || workload == swh_workload_short_words_k ||
        workload == swh_workload_tokens64_k)
        related = rng.below(2) == 1;
    if (!related) { fill(workload, rng, b, draw_length(workload, rng)); return; }
    b = a;
    uint32_t edits;
    if (workload == swh_workload_words16_k || workload == swh_workload_short_words_k) edits = rng.range(0, 3);
    else {
        double rate = workload == swh_workload_tokens64_k ? 0.10 : 0.15;
        double want = rate * (double)a.size();
        edits = (uint32_t)want;
        if (rng.below(1000) < (uint32_t)((want - edits) * 1000.0)) ++edits;
    }
    mutate(workload, rng, b, edits);
    if ((workload == swh_workload_words16_k || workload == swh_workload_short_words_k) && b.size() > 16) b.resize(16);
}

struct Shard { std::vector<uint8_t> da, db; std::vector<uint64_t> la, lb; };

}  // namespace

SWH_EXPORT swh_status_t swh_synth_generate(int workload, uint64_t seed, uint64_t first, size_t count, int threads,
                                           swh_synth_t *out, const char **error) {
    static const char *bad = "unknown synthetic workload id";
    static const char *oom = "host allocation failed";
    if (!out) return swh_invalid_argument_k;
    memset(out, 0, sizeof *out);
    switch (workload) {
    case swh_workload_words16_k: case swh_workload_tokens64_k: case swh_workload_utf8_lines_k:
    case swh_workload_protein4k_k: case swh_workload_short_words_k: case swh_workload_bytes4k_k: case swh_workload_script_lines_k: break;
    default: if (error) *error = bad; return swh_invalid_argument_k;
    }
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if ((size_t)threads > count / 1024 + 1) threads = (int)(count / 1024 + 1);
    const bool utf8 = workload == swh_workload_utf8_lines_k || workload == swh_workload_script_lines_k;
    std::vector<Shard> shards(threads);
    auto work = [&](int t) {
        size_t lo = count * (size_t)t / threads, hi = count * (size_t)(t + 1) / threads;
        Shard &sh = shards[t];
        Symbols a, b;
        sh.la.reserve(hi - lo); sh.lb.reserve(hi - lo);
        for (size_t i = lo; i < hi; ++i) {
            make_pair(workload, seed, first + i, a, b);
            size_t a0 = sh.da.size(), b0 = sh.db.size();
            encode(a, utf8, sh.da); encode(b, utf8, sh.db);
            sh.la.push_back(sh.da.size() - a0); sh.lb.push_back(sh.db.size() - b0);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    size_t bytes_a = 0, bytes_b = 0;
    for (auto &sh : shards) { bytes_a += sh.da.size(); bytes_b += sh.db.size(); }
    out->count = count;
    out->data_a = (uint8_t *)malloc(bytes_a + 16); out->data_b = (uint8_t *)malloc(bytes_b + 16);
    out->offsets_a = (uint64_t *)malloc((count + 1) * 8); out->offsets_b = (uint64_t *)malloc((count + 1) * 8);
    if (!out->data_a || !out->data_b || !out->offsets_a || !out->offsets_b) {
        swh_synth_free(out);
        if (error) *error = oom;
        return swh_bad_alloc_k;
    }
    size_t pa = 0, pb = 0, idx = 0;
    for (auto &sh : shards) {
        if (!sh.da.empty()) memcpy(out->data_a + pa, sh.da.data(), sh.da.size());
        if (!sh.db.empty()) memcpy(out->data_b + pb, sh.db.data(), sh.db.size());
        size_t oa = pa, ob = pb;
        for (size_t i = 0; i < sh.la.size(); ++i, ++idx) {
            out->offsets_a[idx] = oa; out->offsets_b[idx] = ob;
            oa += sh.la[i]; ob += sh.lb[i];
        }
        pa += sh.da.size(); pb += sh.db.size();
        Shard().da.swap(sh.da); Shard().db.swap(sh.db);
    }
    out->offsets_a[count] = pa; out->offsets_b[count] = pb;
    return swh_success_k;
}

SWH_EXPORT void swh_synth_free(swh_synth_t *t) {
    if (!t) return;
    free(t->data_a); free(t->data_b); free(t->offsets_a); free(t->offsets_b);
    memset(t, 0, sizeof *t);
}

SWH_EXPORT void swh_synth_matrix(uint64_t seed, const char *alphabet, int8_t *m) {
    bool used[256];
    for (int i = 0; i < 256; ++i) used[i] = alphabet == nullptr;
    if (alphabet) for (const unsigned char *p = (const unsigned char *)alphabet; *p; ++p) used[*p] = true;
    Rng rng(seed, 0xA11CEull);
    for (int i = 0; i < 256; ++i)
        for (int j = 0; j <= i; ++j) {
            int8_t v;
            if (!used[i] || !used[j]) v = -4;
            else if (i == j) v = (int8_t)(4 + (int)rng.below(8));
            else v = (int8_t)(-4 + (int)rng.below(8));
            m[i * 256 + j] = v; m[j * 256 + i] = v;
        }
}

SWH_EXPORT void swh_unary_class_costs(int8_t match, int8_t mismatch, uint8_t *byte_to_class, int8_t *class_costs) {
    for (int b = 0; b < 256; ++b) byte_to_class[b] = (uint8_t)(b % 32);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) class_costs[i * 32 + j] = i == j ? match : mismatch;
}

SWH_EXPORT size_t swh_crossproduct_side(size_t budget, size_t tape_len) {
    size_t target = (size_t)std::llround(std::sqrt((double)budget));
    if (target < 1) target = 1;
    size_t max_side = tape_len / 2;
    size_t side = target < max_side ? target : max_side;
    return side < 1 ? 1 : side;
}

SWH_EXPORT size_t swh_auto_batch_size(size_t cores, size_t default_base) {
    size_t per_core = default_base;
    if (const char *env = getenv("STRINGWARS_BATCH_PER_CORE")) {
        char *end = nullptr;
        unsigned long long v = strtoull(env, &end, 10);
        if (end != env && *end == 0) per_core = (size_t)v;
    }
    if (per_core < 1) per_core = 1;
    if (cores < 1) cores = 1;
    size_t batch = per_core * cores;
    if (per_core != 0 && batch / per_core != cores) batch = (size_t)-1;  // saturating_mul
    return batch < 1 ? 1 : batch;
}

static double scale_si(double v, const char **prefix) {
    if (v >= 1e9) { *prefix = "G"; return v / 1e9; }
    if (v >= 1e6) { *prefix = "M"; return v / 1e6; }
    if (v >= 1e3) { *prefix = "k"; return v / 1e3; }
    *prefix = "";
    return v;
}

SWH_EXPORT size_t swh_format_si_rate(double rate, const char *unit, int space_before_unit, char *buffer,
                                     size_t capacity) {
    const char *prefix;
    double value = scale_si(rate, &prefix);
    int n;
    if (!*prefix) n = snprintf(buffer, capacity, "%.2f %s", value, unit);
    else if (space_before_unit) n = snprintf(buffer, capacity, "%.2f %s %s", value, prefix, unit);
    else n = snprintf(buffer, capacity, "%.2f %s%s", value, prefix, unit);
    return n < 0 ? 0 : (size_t)n;
}

SWH_EXPORT size_t swh_format_seconds(double value, char *buffer, size_t capacity) {
    int n;
    if (value < 1e-6) n = snprintf(buffer, capacity, "%.2f ns", value * 1e9);
    else if (value < 1e-3) n = snprintf(buffer, capacity, "%.2f \xC2\xB5s", value * 1e6);
    else if (value < 1.0) n = snprintf(buffer, capacity, "%.2f ms", value * 1e3);
    else n = snprintf(buffer, capacity, "%.2f s", value);
    return n < 0 ? 0 : (size_t)n;
}
