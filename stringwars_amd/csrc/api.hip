// api.hip -- the extern "C" boundary declared in include/stringwars_amd.h.
//
// Host-side plumbing only: scope/engine lifetime, residency detection and staging of host tapes,
// scratch carving, the device pre-pass, kernel dispatch and optional hipEvent timing. There is no
// CPU compute path in this library: without a HIP device every entry point fails with
// swh_no_device_k.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "common.hpp"
#include <algorithm>
#include <utility>

namespace swh {

// ---- error text (static storage, one slot per thread) ----------------------------------------
static thread_local char g_error_text[512];
static swh_status_t fail(const char **error, swh_status_t status, const char *fmt, const char *a = "",
                         const char *b = "") {
    snprintf(g_error_text, sizeof g_error_text, fmt, a, b);
    if (error) *error = g_error_text;
    return status;
}
static swh_status_t fail_hip(const char **error, const HipFailure &f) {
    snprintf(g_error_text, sizeof g_error_text, "HIP error '%s' in %s", hipGetErrorString(f.code), f.what);
    if (error) *error = g_error_text;
    (void)hipGetLastError();
    return f.code == hipErrorOutOfMemory ? swh_bad_alloc_k : swh_device_error_k;
}

// ---- kernel stamps -------------------------------------------------------------------------------
// STRINGWARS_AMD_TRACE=1: every stamped launch is announced on stderr and waited for -- the last name printed before a device
// fault is the kernel behind it (diagnostics only: it serialises everything).
static bool trace_launches() {
    static const bool on = [] { const char *e = getenv("STRINGWARS_AMD_TRACE"); return e && atoi(e) != 0; }();
    return on;
}
// STRINGWARS_AMD_STAMPS=1: with profiling on, the start and the length of every stamped launch of a call (event times) on stderr.
static bool trace_stamps() {
    static const bool on = [] { const char *e = getenv("STRINGWARS_AMD_STAMPS"); return e && atoi(e) != 0; }();
    return on;
}
StampGuard::StampGuard(Scope *s, const char *name) : scope(s), idx(0), on(s->profiling) {
    if (trace_launches()) { fprintf(stderr, "[swh] launch %s\n", name); fflush(stderr); }
    if (!on) return;
    if (scope->stamps_used == scope->stamps.size()) {
        KernelStamp st{};
        if (hipEventCreate(&st.start) != hipSuccess || hipEventCreate(&st.stop) != hipSuccess) { on = false; return; }
        scope->stamps.push_back(st);
    }
    idx = scope->stamps_used++;
    scope->stamps[idx].name = name;
    (void)hipEventRecord(scope->stamps[idx].start, scope->stream);
}
StampGuard::~StampGuard() {
    if (on) (void)hipEventRecord(scope->stamps[idx].stop, scope->stream);
    if (trace_launches()) {
        const hipError_t err = hipStreamSynchronize(scope->stream);
        fprintf(stderr, "[swh]   done (%s)\n", hipGetErrorString(err)); fflush(stderr);
    }
}

static void collect_timing(Scope *scope) {
    swh_timing_t &t = scope->last_timing;
    t.total_ms = 0; t.dominant_ms = 0; t.compute_ms = 0; t.dominant_name[0] = 0; t.kernels = (uint32_t)scope->stamps_used;
    if (!scope->stamps_used) return;
    float span = 0;
    for (size_t i = 0; i < scope->stamps_used; ++i) {   // the last kernel to finish need not be the last one launched
        float to_stop = 0;
        (void)hipEventElapsedTime(&to_stop, scope->stamps[0].start, scope->stamps[i].stop);
        if (to_stop > span) span = to_stop;
    }
    t.total_ms = span;
    // DP kernels may run side by side on two streams (wavefront classes): compute_ms is the length of the UNION of
    // their intervals, not the sum of their durations
    std::vector<std::pair<float, float>> dp;
    for (size_t i = 0; i < scope->stamps_used; ++i) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, scope->stamps[i].start, scope->stamps[i].stop);
        if (strncmp(scope->stamps[i].name, "plan_", 5) != 0 && strncmp(scope->stamps[i].name, "utf8_", 5) != 0) {
            float from = 0;
            (void)hipEventElapsedTime(&from, scope->stamps[0].start, scope->stamps[i].start);
            dp.emplace_back(from, from + ms);
        }
        if (trace_launches() || trace_stamps()) {
            float from = 0;
            (void)hipEventElapsedTime(&from, scope->stamps[0].start, scope->stamps[i].start);
            fprintf(stderr, "[swh] stamp %-28s start %8.3f ms  length %8.3f ms\n", scope->stamps[i].name, from, ms);
        }
        if (ms > t.dominant_ms) {
            t.dominant_ms = ms;
            snprintf(t.dominant_name, sizeof t.dominant_name, "%s", scope->stamps[i].name);
        }
    }
    std::sort(dp.begin(), dp.end());
    float covered = 0, reach = -1e30f;
    for (const auto &iv : dp) {
        if (iv.first > reach) { covered += iv.second - iv.first; reach = iv.second; }
        else if (iv.second > reach) { covered += iv.second - reach; reach = iv.second; }
    }
    t.compute_ms = covered;
}

// Reads the events of the scope's last call once they are complete and adds the call to the running totals.
// `complete`: the caller has synchronised with the scope's last call. Otherwise (the start of the next call in
// asynchronous mode) completion is only known when profiling is on, through the call's stop events.
static void harvest_timing(Scope *scope, bool complete) {
    const bool timed = scope->profiling && scope->stamps_pending && scope->stamps_used;
    if (timed) {
        for (size_t i = 0; i < scope->stamps_used; ++i) (void)hipEventSynchronize(scope->stamps[i].stop);   // two streams: no single last event
        complete = true;
    }
    const uint64_t cells = scope->last_timing.cells, bytes = scope->last_timing.bytes;
    if (timed) {
        collect_timing(scope);
        scope->last_timing.cells = cells; scope->last_timing.bytes = bytes;
    }
    // a plan-free call reports its work units through host-mapped memory
    if (scope->summary_pending && complete) {
        const CallSummary sm = scope->summary_host[scope->summary_slot];
        scope->last_timing.cells = sm.cells;
        scope->last_timing.bytes = (scope->summary_extra_bytes ? scope->summary_extra_bytes : sm.symbols * scope->summary_sym_bytes) +
                                   scope->summary_pairs * (2 * scope->summary_ow + scope->summary_elem);
        if (!sm.violation) {
            scope->hint_lengths = true;
            scope->hint_max_la = sm.max_la; scope->hint_max_lb = sm.max_lb;
            scope->hint_mean_x16 = scope->summary_pairs ? (uint32_t)std::min<uint64_t>(sm.symbols * 8 / scope->summary_pairs, 0xFFFFFFu) : 0u;
            scope->hint_mean_string_x16 = scope->summary_strings ? (uint32_t)std::min<uint64_t>(sm.symbols * 16 / scope->summary_strings, 0xFFFFFFu) : 0u;
            scope->hint_short = (uint64_t)sm.short_pairs * 4 >= scope->summary_pairs;
        } else if (scope->async) {
            // An asynchronous plan-free call cannot be redone behind the caller's back (synchronous calls are: run_call_on).
            // It only runs on prepared tapes, whose lengths were measured: a pair that does not fit means the tape's memory
            // changed after swh_tape_prepare_* -- the misfit pairs were NOT scored. Reported by the next synchronisation.
            scope->violation_seen = true;
        }
    }
    scope->summary_pending = false;
    if (timed) {
        scope->totals.total_ms += scope->last_timing.total_ms;
        scope->totals.dominant_ms += scope->last_timing.dominant_ms;
        scope->totals.compute_ms += scope->last_timing.compute_ms;
        scope->totals.calls += 1;
    }
    scope->stamps_pending = false;
}

// ---- scratch -------------------------------------------------------------------------------------
struct Carver {
    char *base; size_t used, cap;
    template <typename T> T *take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        char *p = base ? base + used : nullptr;
        used += bytes;
        return (T *)p;
    }
};

static void ensure(char *&buf, size_t &cap, size_t need) {
    if (need <= cap) return;
    if (buf) SWH_HIP_CHECK(hipFree(buf));
    buf = nullptr; cap = 0;
    size_t want = need + need / 4 + (1 << 20);
    SWH_HIP_CHECK(hipMalloc((void **)&buf, want));
    cap = want;
}

// both tapes of a call up to this size (together) are staged by one launch pair; STRINGWARS_AMD_UTF8_MERGED_MB=n moves it
static uint64_t utf8_merged_bytes() {
    static const uint64_t bytes = [] { const char *e = test_hook("STRINGWARS_AMD_UTF8_MERGED_MB"); return e ? (uint64_t)atol(e) << 20 : ~0ull; }();
    return bytes;
}
// STRINGWARS_AMD_UTF8_STAGING: `strings` -- every raw UTF-8 call on the planned / tiled routes stages string by string (k_utf8_strings: the
// tests send words and empty strings through it), `tiles` -- never (the flat one-pass kernel: the comparison), unset -- tapes whose
// mean string has at least kUtf8StringsMeanBytes bytes.
static int utf8_strings_mode() {
    static const int mode = [] { const char *e = test_hook("STRINGWARS_AMD_UTF8_STAGING"); return !e ? 0 : (!strcmp(e, "strings") ? 1 : (!strcmp(e, "tiles") ? 2 : 0)); }();
    return mode;
}
// u32 words of scratch the flat UTF-8 decoder needs for a tape of `bytes` bytes (see launch_utf8_decode)
static size_t utf8_scratch_words(uint64_t bytes) {
    // mirrors the carving in launch_utf8_decode: tile counts | sub-tile prefixes | u64 tile prefixes | u64 block sums | balances
    // (the look-back words and tickets of the one-pass kernel live in scope->utf8_status / the call's flag words)
    uint64_t tiles = (bytes + kUtf8Tile - 1) / kUtf8Tile;
    return (size_t)((tiles + 4) + (kUtf8Subs * tiles + 4) + 2 * (tiles + 4) + 2 * ((tiles + 1023) / 1024 + 4) + (tiles + 6));
}

static bool is_device_pointer(const void *p) {
    if (!p) return true;
    hipPointerAttribute_t attr;
    hipError_t err = hipPointerGetAttributes(&attr, p);
    if (err != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// hipMalloc'ed memory of a device (not managed, not host-mapped): where a write-through store that was acknowledged can be read by anybody
// -- ON THE SCOPE'S OWN DEVICE: agent-scope write-through is visibility on that agent; a peer device's memory takes the stream's way
static bool is_plain_device_memory(const void *p, int device) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t err = hipPointerGetAttributes(&attr, p);
    if (err != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice && attr.device == device;
}

// ---- one engine call ---------------------------------------------------------------------------
struct HostTape { const uint8_t *data; const void *offsets; size_t count; int off64; };

// swh_prepared_t: a tape made ready once -- resident on the device, measured, and (UTF-8) validated and decoded --
// the counterpart of the `BytesTapeView` / `CharsTapeView` the reference builds ONCE outside its timed closures
// (bench.rs:292-306) and sub-views per iteration (bench.rs:134-139).
struct Prepared {
    int device = 0;
    bool utf8 = false;           // symbols are Unicode scalar values
    bool ascii = false;          // utf8 and every code point is one byte: the byte tape is the code-point tape
    uint32_t off64 = 0;          // width of the byte tape's offsets
    TapeRef bytes{};             // device: u8 data, u32/u64 offsets
    TapeRef symbols{};           // utf8: u32 code points + u64 code-point offsets; otherwise unused
    uint64_t total_bytes = 0, total_symbols = 0;
    uint32_t longest_bytes = 0, longest_symbols = 0;
    std::vector<void *> owned;   // device buffers that go with the handle
};

struct CallSpec {
    HostTape a, b;
    bool cross, utf8;
    uint32_t bound;
    void *out; size_t out_stride, row_stride; bool out64;
    const Prepared *pa = nullptr, *pb = nullptr;   // prepared tapes (both or neither); a.count / b.count = the views' counts
    size_t a_first = 0, b_first = 0;
    bool force_planned = false;                    // redo of a call whose plan-free kernel met a pair it could not score
    uint32_t skip_upto = 0;                        // ... where that kernel HAS scored every pair of two strings of at most this many symbols
    bool flat_staging = false;                     // redo of a raw UTF-8 call whose string-by-string staging met a string too long for it
};

static uint64_t read_offset(const void *offs, int off64, size_t i, bool device, hipStream_t stream) {
    uint64_t v = 0;
    size_t w = off64 ? 8 : 4;
    if (device) {
        SWH_HIP_CHECK(hipMemcpyAsync(&v, (const char *)offs + i * w, w, hipMemcpyDeviceToHost, stream));
        SWH_HIP_CHECK(hipStreamSynchronize(stream));
    } else {
        memcpy(&v, (const char *)offs + i * w, w);
    }
    return v;
}

static swh_status_t run_call_on(Scope *scope, const Engine *engine, const CallSpec &spec, const char **error);

// Does either tape hold a byte above 0x7F? Runs in front of the byte kernels when a UTF-8 call BELIEVES its raw tapes to be ASCII
// (they were, the last time this scope staged them): sixteen bytes per thread and step, the flag in host-mapped memory.
// The tapes' byte totals are read HERE, from offsets[count]: the caller's belief about them may be as stale as the one about their bytes.
__global__ __launch_bounds__(256) void k_ascii_check(const uint8_t *a, const void *a_offsets, uint64_t a_count, const uint8_t *b, const void *b_offsets,
                                                     uint64_t b_count, uint32_t off64, uint32_t *flag) {
    uint32_t high = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 16;
    for (int t = 0; t < 2; ++t) {
        const uint8_t *data = t ? b : a;
        const void *offsets = t ? b_offsets : a_offsets;
        if (!offsets) continue;
        const uint64_t count = t ? b_count : a_count;
        const uint64_t bytes = off64 ? ((const uint64_t *)offsets)[count] : (uint64_t)((const uint32_t *)offsets)[count];
        for (uint64_t at = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; at < bytes; at += stride) {
            if (at + 16 <= bytes) {
                uint4 v;
                __builtin_memcpy(&v, data + at, 16);
                high |= v.x | v.y | v.z | v.w;
            } else {
                for (uint64_t i = at; i < bytes; ++i) high |= data[i];
            }
        }
    }
    if (__ballot((high & 0x80808080u) != 0) != 0 && (threadIdx.x & 63) == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static swh_status_t run_call(Scope *scope, const Engine *engine, const CallSpec &spec, const char **error) {
    if (!scope || !engine) return fail(error, swh_invalid_argument_k, "null scope or engine");
    if (!scope->pipelined) return run_call_on(scope, engine, spec, error);
    Scope *lane = scope->lanes[scope->next_lane];
    scope->next_lane ^= 1;
    scope->last_lane = lane;
    // The lane starts where the caller's stream is now: whatever was enqueued there before this call -- the producer of
    // the inputs, or a consumer still reading the output buffer this call is about to overwrite (an RCCL gather of an
    // earlier step) -- completes first. swh_scope_join gives the other direction.
    if (hipEventRecord(scope->order_ev, scope->stream) != hipSuccess ||
        hipStreamWaitEvent(lane->stream, scope->order_ev, 0) != hipSuccess)
        return fail(error, swh_device_error_k, "could not order a pipeline lane after the scope's stream");
    swh_status_t status = run_call_on(lane, engine, spec, error);
    if (status == swh_success_k && hipEventRecord(lane->lane_done, lane->stream) != hipSuccess)
        return fail(error, swh_device_error_k, "hipEventRecord failed on a pipeline lane");
    scope->last_timing = lane->last_timing;
    return status;
}

// Which kernels a unit-cost Levenshtein call runs on.
enum Route { kRoutePlanned, kRouteTiled, kRouteDirectShort, kRouteShortTiled, kRouteCrossShort, kRouteAlignShort, kRouteAlignLong };

// Strings up to this many symbols (G <= 8 blocks) are scored by the tiled kernel when their lengths are known; beyond it
// a tile holds too few pairs per block count and the global sort of the planned path packs the waves better.
static uint32_t tiled_longest_limit() {
    static const uint32_t limit = [] { const char *e = test_hook("STRINGWARS_AMD_TILED_MAX"); return e ? (uint32_t)atoi(e) : 256u; }();
    return limit;
}
// Word-sized batches: k_short_tiled (<= 16 bytes, batches large enough to give every workgroup a chunk worth sorting) /
// k_direct_short (<= 32 bytes, small batches). Comparison knob STRINGWARS_AMD_SHORT: `direct`
// keeps k_direct_short for all of them, `tiled` sends them to the general tiled kernel.
// (STRINGWARS_AMD_SHORT_MIN_PAIRS, read per call: the tests send small batches to the chunked kernel with it)
static uint64_t short_tiled_min_pairs() {
#ifdef SWH_TEST_HOOKS
    const char *e = test_hook("STRINGWARS_AMD_SHORT_MIN_PAIRS");   // (the test library reads it per call: the tests move it between calls)
    return e ? (uint64_t)atoll(e) : (uint64_t)1 << 16;
#else
    static const uint64_t pairs = [] { const char *e = test_hook("STRINGWARS_AMD_SHORT_MIN_PAIRS"); return e ? (uint64_t)atoll(e) : (uint64_t)1 << 16; }();
    return pairs;
#endif
}
// Bounds up to here may take the banded kernel (STRINGWARS_AMD_BAND_MAX=63: the one-word windows only, the comparison knob).
static uint32_t band_max_bound() {
    static const uint32_t most = [] { const char *e = test_hook("STRINGWARS_AMD_BAND_MAX"); const uint32_t v = e ? (uint32_t)atoi(e) : kBandMaxBound; return v > kBandMaxBound ? kBandMaxBound : v; }();
    return most;
}
// Longest string k_align_cross_long is chosen for: the longest query it takes (STRINGWARS_AMD_ALIGN_LONG_MAX=n lowers it) ...
static uint32_t align_long_limit() {
    static const uint32_t limit = [] { const char *e = test_hook("STRINGWARS_AMD_ALIGN_LONG_MAX"); const uint32_t v = e ? (uint32_t)atoi(e) : 4096u; return v > 4096u ? 4096u : v; }();
    return limit;
}
// ... and per form, the length up to which it measured faster than the column-profile kernel on DNA cross-products (TCUPS, long
// kernel : profile kernel):       1 K symbols        3 K symbols
//      NW linear  (W = 128)       12.8 :  9.9        12.6 : 10.5       -> as far as the kernel goes (a boundary buffer of 2.2 GB there)
//      NW affine  (W =  64)        6.1 :  5.7         5.8 :  6.1       -> 2048
//      SW linear  (W =  64)        6.9 :  6.1         6.6 :  6.6       -> 2048
//      SW affine  (W =  32)        3.7 :  3.7         3.9 :  3.9       -> stays where the wavefront class kernels were the alternative
static uint32_t align_long_pays(bool local, bool affine) { return local && affine ? 384u : (local || affine ? 2048u : 4096u); }
// A synchronous call whose results stay on the device returns when its summary has LANDED -- the last workgroup writes that word
// into host-mapped memory once every workgroup's (write-through) result stores were acknowledged -- instead of when the stream
// reports the kernel complete: the end of a kernel as the runtime sees it (the release that writes the L2s back, the completion
// signal, the wake-up) costs ~5 us that no result waits for (tools/launch_probe.hip: 12.5 us against 7.3 for an empty kernel).
// What the stream still orders is unchanged: the scope's next call, its copies, swh_scope_synchronize. Only for outputs in plain
// device memory (hipMalloc), outside profiling, on the routes whose one kernel reports the summary; after 2 ms of polling the
// call waits for the stream the ordinary way. STRINGWARS_AMD_EARLY_RETURN=0 always does.
static bool early_return_on() {
    static const bool on = [] { const char *e = getenv("STRINGWARS_AMD_EARLY_RETURN"); return !e || atoi(e) != 0; }();
    return on;
}
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
// The poll is bounded by what the scope's previous early-returning call took: a call that ran for more than ~0.4 ms last time waits for
// the stream straight away (5 us are nothing to it, and a core spinning for milliseconds per call is), a shorter one spins for at most
// four times that -- never more than 2 ms -- before it falls back to the stream (which also reports a kernel that died).
static void wait_for_summary(Scope *scope, hipStream_t stream) {
    volatile uint32_t *landed = &scope->summary_host[0].landed;
    const auto begun = std::chrono::steady_clock::now();
    const auto finish = [&]() {
        const auto took = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begun).count();
        scope->early_return_last_us = (uint32_t)std::min<long long>(took, 1000000);
    };
    if (scope->early_return_last_us > 400) {
        SWH_HIP_CHECK(hipStreamSynchronize(stream));
        finish();
        return;
    }
    const auto patience = std::chrono::microseconds(std::min<uint32_t>(2000u, std::max<uint32_t>(100u, 4u * scope->early_return_last_us)));
    for (uint32_t spins = 1;; ++spins) {
        if (__atomic_load_n(landed, __ATOMIC_ACQUIRE)) { finish(); return; }
        if ((spins & 0xFFu) == 0 && std::chrono::steady_clock::now() - begun > patience) {
            SWH_HIP_CHECK(hipStreamSynchronize(stream));
            finish();
            return;
        }
        cpu_relax();
    }
}

static int short_route_choice() {
    static const int choice = [] {
        const char *e = test_hook("STRINGWARS_AMD_SHORT");
        return !e ? 0 : (!strcmp(e, "direct") ? 1 : (!strcmp(e, "tiled") ? 2 : 0));
    }();
    return choice;
}

static swh_status_t run_call_on(Scope *scope, const Engine *engine, const CallSpec &spec, const char **error) {
    if (!scope || !engine) return fail(error, swh_invalid_argument_k, "null scope or engine");
    if (!spec.out && spec.a.count) return fail(error, swh_invalid_argument_k, "null output pointer");
    if (!spec.cross && spec.a.count != spec.b.count)
        return fail(error, swh_invalid_argument_k, "pairwise call needs tapes of equal count");
    const uint64_t pairs = spec.cross ? (uint64_t)spec.a.count * spec.b.count : spec.a.count;
    if (pairs >= 0xFFFFFFF0ull) return fail(error, swh_unsupported_length_k, "more than 2^32 pairs in one call");
    const bool prepared = spec.pa != nullptr;
    if (prepared) {
        if (!spec.pb) return fail(error, swh_invalid_argument_k, "both tapes must be prepared, or neither");
        if (spec.pa->utf8 != spec.pb->utf8) return fail(error, swh_invalid_argument_k, "one tape was prepared as UTF-8, the other as bytes");
        if (spec.pa->device != scope->device || spec.pb->device != scope->device)
            return fail(error, swh_invalid_argument_k, "a prepared tape lives on another device than the scope");
    }
    harvest_timing(scope, false);   // an earlier asynchronous call on this scope / lane
    scope->stamps_used = 0;
    scope->last_timing = swh_timing_t{};
    if (pairs == 0) return swh_success_k;
    // code points of pure-ASCII tapes are their bytes: such a pair of prepared tapes runs on the byte kernels
    // (only when both byte tapes have the same offset width: whether a UTF-8 call is accepted must not depend on what the
    // tapes contain -- a u32 / u64 mix falls back to the decoded tapes, whose offsets are always u64)
    const bool utf8 = prepared ? (spec.pa->utf8 && !(spec.pa->ascii && spec.pb->ascii && spec.pa->off64 == spec.pb->off64)) : spec.utf8;
    if (prepared && !utf8 && spec.pa->off64 != spec.pb->off64)
        return fail(error, swh_invalid_argument_k, "prepared byte tapes must share one offset width");
    if (utf8 && engine->scoring.matrix)
        return fail(error, swh_not_implemented_k, "substitution-matrix scoring over UTF-8 code points (the matrix is indexed by bytes)");
    try {
        SWH_HIP_CHECK(hipSetDevice(scope->device));
        hipStream_t stream = scope->stream;
        const size_t ow = prepared ? (utf8 ? 8 : (spec.pa->off64 ? 8 : 4)) : (spec.a.off64 ? 8 : 4);
        const size_t elem = spec.out64 ? 8 : 4;

        // -- residency --------------------------------------------------------------------------
        bool dev_a_data = true, dev_a_off = true, dev_b_data = true, dev_b_off = true;
        bool same_tape = false;
        if (!prepared) {
            dev_a_data = is_device_pointer(spec.a.data); dev_a_off = is_device_pointer(spec.a.offsets);
            dev_b_data = is_device_pointer(spec.b.data); dev_b_off = is_device_pointer(spec.b.offsets);
            same_tape = spec.b.data == spec.a.data && spec.b.offsets == spec.a.offsets && spec.b.count == spec.a.count;
        }
        const bool dev_out = is_device_pointer(spec.out);
        static const bool believe = [] { const char *e = test_hook("STRINGWARS_AMD_SIZE_BELIEF"); return !e || atoi(e) != 0; }();
        auto same_as_believed = [](const HostTape &t, const Scope::SizeBelief &slot) {
            return slot.valid && slot.data == t.data && slot.offsets == t.offsets && slot.count == t.count && slot.off64 == t.off64;
        };
        // -- raw UTF-8 tapes that were pure ASCII the last time this scope staged them: code points of ASCII text are its bytes, so the
        // call runs on the byte kernels -- no staging, the word-sized and cross-product kernels instead of the code-point ones -- behind
        // a kernel that checks every byte of both tapes; if one is above 0x7F after all, the call is done again the long way.
        // (Synchronous scopes only: an asynchronous call cannot be redone behind the caller's back.)
        if (believe && utf8 && !prepared && !scope->async && !spec.force_planned && dev_a_data && dev_a_off && dev_b_data && dev_b_off &&
            same_as_believed(spec.a, scope->size_belief[0]) && scope->size_belief[0].ascii &&
            (same_tape || (same_as_believed(spec.b, scope->size_belief[1]) && scope->size_belief[1].ascii))) {
            const Scope::SizeBelief &ba = scope->size_belief[0], &bb = same_tape ? scope->size_belief[0] : scope->size_belief[1];
            void *base = nullptr; size_t size = 0;
            const bool covered = hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)spec.a.data) == hipSuccess &&
                                 (const char *)spec.a.data + ba.bytes <= (const char *)base + size &&
                                 hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)spec.b.data) == hipSuccess &&
                                 (const char *)spec.b.data + bb.bytes <= (const char *)base + size;
            if (covered) {
                uint32_t *seen_host = (uint32_t *)((char *)scope->summary_host + 128), *seen_dev = (uint32_t *)((char *)scope->summary_dev + 128);
                *seen_host = 0;
                const uint64_t most = std::max<uint64_t>(ba.bytes, bb.bytes);
                const uint32_t blocks = (uint32_t)std::min<uint64_t>((most + 4095) / 4096 + 1, (uint64_t)scope->compute_units * 8);
                hipLaunchKernelGGL(k_ascii_check, dim3(blocks), dim3(256), 0, stream, (const uint8_t *)spec.a.data, spec.a.offsets, (uint64_t)spec.a.count,
                                   (const uint8_t *)spec.b.data, same_tape ? nullptr : spec.b.offsets, (uint64_t)spec.b.count, (uint32_t)spec.a.off64, seen_dev);
                SWH_HIP_CHECK(hipGetLastError());
                CallSpec as_bytes = spec;
                as_bytes.utf8 = false;
                const swh_status_t status = run_call_on(scope, engine, as_bytes, error);
                if (status != swh_success_k) return status;
                // (the check ran in front of the byte kernels on the same stream: it is complete when their results are)
                if (__atomic_load_n(seen_host, __ATOMIC_ACQUIRE) == 0) return swh_success_k;
                scope->size_belief[0].ascii = scope->size_belief[1].ascii = false;
                scope->summary_pending = false;
                scope->stamps_pending = false;
            } else {
                (void)hipGetLastError();
            }
        }
        uint64_t a_bytes = 0, b_bytes = 0;
        const bool need_sizes = !prepared && (!dev_a_data || !dev_b_data || utf8);
        bool believed_sizes = false;
        if (need_sizes) {
            // A UTF-8 call on raw device tapes needs the tapes' byte totals before its first launch (scratch, grids): two synchronous
            // 4-byte copies, ~25 us of a 0.7 ms call. The same tapes as last time (pointers, count) are believed to hold the same
            // totals if the allocations still cover them; k_utf8_finish compares with offsets[count] and the call is redone if not.
            auto total_of = [&](const HostTape &t, bool dev_data, bool dev_off, Scope::SizeBelief &slot) -> uint64_t {
                if (believe && utf8 && dev_data && dev_off && !spec.force_planned && same_as_believed(t, slot)) {
                    void *base = nullptr; size_t size = 0;
                    const size_t ow_t = t.off64 ? 8 : 4;
                    bool covered = hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)t.data) == hipSuccess &&
                                   (const char *)t.data + slot.bytes <= (const char *)base + size;
                    covered = covered && hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)t.offsets) == hipSuccess &&
                              (const char *)t.offsets + (t.count + 1) * ow_t <= (const char *)base + size;
                    if (covered) { believed_sizes = true; return slot.bytes; }
                    (void)hipGetLastError();
                }
                const uint64_t bytes = read_offset(t.offsets, t.off64, t.count, dev_off, stream);
                slot.data = t.data; slot.offsets = t.offsets; slot.count = t.count; slot.off64 = t.off64; slot.bytes = bytes; slot.valid = dev_data && dev_off;
                slot.ascii = false;
                return bytes;
            };
            // (two device tapes the scope has no belief about: both totals in ONE round trip -- two copies, one wait -- instead of two)
            const bool both_fresh = !same_tape && dev_a_off && dev_b_off && dev_a_data && dev_b_data &&
                                    !(believe && utf8 && !spec.force_planned && (same_as_believed(spec.a, scope->size_belief[0]) || same_as_believed(spec.b, scope->size_belief[1])));
            if (both_fresh) {
                uint64_t *words = (uint64_t *)(scope->plan_host + 1) + 3;   // (pinned: bytes 24 .. 39 behind the plan; the UTF-8 flag words land in its first 20)
                words[0] = words[1] = 0;
                const size_t wa = spec.a.off64 ? 8 : 4, wb = spec.b.off64 ? 8 : 4;
                SWH_HIP_CHECK(hipMemcpyAsync(&words[0], (const char *)spec.a.offsets + spec.a.count * wa, wa, hipMemcpyDeviceToHost, stream));
                SWH_HIP_CHECK(hipMemcpyAsync(&words[1], (const char *)spec.b.offsets + spec.b.count * wb, wb, hipMemcpyDeviceToHost, stream));
                SWH_HIP_CHECK(hipStreamSynchronize(stream));
                a_bytes = words[0]; b_bytes = words[1];
                auto remember = [](Scope::SizeBelief &slot, const HostTape &t, uint64_t bytes) {
                    slot.data = t.data; slot.offsets = t.offsets; slot.count = t.count; slot.off64 = t.off64; slot.bytes = bytes; slot.valid = true; slot.ascii = false;
                };
                remember(scope->size_belief[0], spec.a, a_bytes);
                remember(scope->size_belief[1], spec.b, b_bytes);
            } else {
                a_bytes = total_of(spec.a, dev_a_data, dev_a_off, scope->size_belief[0]);
                b_bytes = same_tape ? a_bytes : total_of(spec.b, dev_b_data, dev_b_off, scope->size_belief[1]);
            }
        }
        // device-resident outputs are written in place with the caller's strides; host outputs are produced
        // compactly in device staging and scattered into the caller's strides by a 2-D copy
        const size_t dev_out_stride = dev_out ? spec.out_stride : elem;
        const size_t dev_row_stride = dev_out ? spec.row_stride : spec.b.count * elem;
        size_t out_bytes = spec.cross ? (spec.a.count ? (spec.a.count - 1) * dev_row_stride + spec.b.count * elem : 0)
                                      : (size_t)(pairs - 1) * dev_out_stride + elem;

        // -- staging of host-resident buffers -----------------------------------------------------
        size_t stage_need = 0;
        auto pad = [](size_t n) { return (n + 255) & ~(size_t)255; };
        if (!dev_a_data) stage_need += pad(a_bytes + 8);
        if (!dev_a_off) stage_need += pad((spec.a.count + 1) * ow + 8);
        if (!same_tape) {
            if (!dev_b_data) stage_need += pad(b_bytes + 8);
            if (!dev_b_off) stage_need += pad((spec.b.count + 1) * ow + 8);
        }
        if (!dev_out) stage_need += pad(out_bytes);
        ensure(scope->stage, scope->stage_bytes, stage_need);
        Carver st{scope->stage, 0, scope->stage_bytes};
        auto stage_in = [&](const void *src, size_t bytes, bool dev) -> const void * {
            if (dev) return src;
            char *dst = st.take<char>(bytes + 8);
            if (bytes) SWH_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
            return dst;
        };
        TapeRef ta, tb;
        uint32_t sym_bytes = 1, off64 = (uint32_t)spec.a.off64;
        if (prepared) {
            auto view = [&](const Prepared *pt, size_t first, size_t count) {
                const TapeRef &whole = utf8 ? pt->symbols : pt->bytes;
                TapeRef t;
                t.data = whole.data;
                t.offsets = (const char *)whole.offsets + first * ow;
                t.count = count;
                return t;
            };
            ta = view(spec.pa, spec.a_first, spec.a.count);
            tb = view(spec.pb, spec.b_first, spec.b.count);
            sym_bytes = utf8 ? 4 : 1;
            off64 = utf8 ? 1 : spec.pa->off64;
        } else {
            ta.data = stage_in(spec.a.data, a_bytes, dev_a_data);
            ta.offsets = stage_in(spec.a.offsets, (spec.a.count + 1) * ow, dev_a_off);
            ta.count = spec.a.count;
            if (same_tape) tb = ta;
            else {
                tb.data = stage_in(spec.b.data, b_bytes, dev_b_data);
                tb.offsets = stage_in(spec.b.offsets, (spec.b.count + 1) * ow, dev_b_off);
                tb.count = spec.b.count;
            }
        }
        char *out_dev = dev_out ? (char *)spec.out : st.take<char>(out_bytes);

        // -- which kernels? ------------------------------------------------------------------------------
        // Unit-cost Levenshtein whose string lengths are known -- from prepared tapes (a guarantee) or from the previous
        // call on this scope (a belief the kernels verify) -- skips the planning pre-pass altogether.
        const bool bitpar_ok = engine->kind == 0 && engine->unit_costs && engine->algorithm != swh_algorithm_wavefront_k;
        Route route = kRoutePlanned;
        uint32_t longest = 0;
        bool guaranteed = false;
        if (bitpar_ok && !spec.force_planned && engine->algorithm != swh_algorithm_bitparallel_k) {
            const bool forced = engine->algorithm == swh_algorithm_tiled_k;
            bool known = false;
            uint32_t la_max = 0, lb_max = 0;
            if (prepared) {
                known = guaranteed = true;
                la_max = utf8 ? spec.pa->longest_symbols : spec.pa->longest_bytes;
                lb_max = utf8 ? spec.pb->longest_symbols : spec.pb->longest_bytes;
            } else if (scope->hint_lengths) {
                known = true;
                la_max = scope->hint_max_la; lb_max = scope->hint_max_lb;
            } else if (forced) {
                known = true;
                la_max = lb_max = 2048;
            }
            // an unverified belief needs the host to look at the outcome: synchronous calls with a device or host output
            const bool can_verify = !scope->async || !dev_out;
            if (known && (guaranteed || can_verify)) {
                longest = la_max > lb_max ? la_max : lb_max;
                const uint32_t shorter_side = la_max < lb_max ? la_max : lb_max;
                // plan_key(): the banded kernel wins from ~6 blocks at k = 32 (bounds of 64 .. 255: where band_cost() says so)
                const bool band_pays = spec.bound <= 63 ? longest > 32
                                                        : spec.bound <= band_max_bound() && band_cost(spec.bound) < (utf8 ? 40u : 28u) * ((longest + 31) >> 5);
                if (forced) route = (!guaranteed || shorter_side <= 2048) ? kRouteTiled : kRoutePlanned;
                else if (!band_pays && longest <= tiled_longest_limit())
                    route = (longest <= 32 && !utf8 && short_route_choice() != 2)
                                ? (spec.cross ? kRouteCrossShort : (longest <= 16 && pairs >= short_tiled_min_pairs() && short_route_choice() == 0 ? kRouteShortTiled : kRouteDirectShort))
                                : ((longest <= 32 && utf8 && spec.cross && short_route_choice() != 2) ? kRouteCrossShort   // word-sized code points: k_cross_short_cp
                                                                                                          : kRouteTiled);
            }
        }

        // Alignment scores on a class table when both tapes hold word-sized strings only (the reference's default `words` token mode,
        // bench.rs:271): one pair per lane, no pre-pass (alignshort.hip). STRINGWARS_AMD_ALIGN_SHORT=0 keeps them on the planned path.
        static const bool align_short_on = [] { const char *e = test_hook("STRINGWARS_AMD_ALIGN_SHORT"); return !e || atoi(e) != 0; }();
        bool align_wide = false;   // kRouteAlignShort on k_align_cross_wide (its alphabet condition is checked by the kernel)
        if (engine->kind != 0 && engine->scoring.class_table && !utf8 && !spec.force_planned && align_short_on) {
            bool known = false;
            uint32_t la_max = 0, lb_max = 0;
            if (prepared) { known = guaranteed = true; la_max = spec.pa->longest_bytes; lb_max = spec.pb->longest_bytes; }
            else if (scope->hint_lengths) { known = true; la_max = scope->hint_max_la; lb_max = scope->hint_max_lb; }
            const bool can_verify = !scope->async || !dev_out;
            const uint32_t both = la_max > lb_max ? la_max : lb_max;
            // up to 128 symbols for cross-products with linear gaps, as long as the candidates of a work item use at most eight symbol
            // classes (DNA; the kernel checks per item, a scope that met richer text stops trying -- `align_wide_off`)
            static const bool wide_on = [] { const char *e = test_hook("STRINGWARS_AMD_ALIGN_WIDE"); return !e || atoi(e) != 0; }();   // comparison knob: 0 = the multi-pass kernel instead
            // (what the latch is keyed by: prepared handles, else the tapes' data pointers -- sub-views of one tape share them)
            const void *key_a = spec.pa ? (const void *)spec.pa : (const void *)spec.a.data, *key_b = spec.pb ? (const void *)spec.pb : (const void *)spec.b.data;
            const bool wide_off = scope->align_wide_off.engine == engine->uid && scope->align_wide_off.a == key_a && scope->align_wide_off.b == key_b;
            const bool wide = wide_on && spec.cross && both <= 128 && engine->scoring.open == engine->scoring.extend && !wide_off && can_verify;
            if (known && (guaranteed || can_verify) && (both <= 32 || wide)) {
                route = kRouteAlignShort;
                longest = both;
                align_wide = both > 32;
            } else if (known && can_verify && spec.cross && !wide_off &&
                       both <= std::min(align_long_limit(), align_long_pays(engine->kind == 2, engine->scoring.open != engine->scoring.extend)) &&
                       align_long_fits(scope, ((uint64_t)(spec.b.count + 63) / 64) * ((uint64_t)(spec.a.count + align_long_queries(scope, spec.a.count, spec.b.count) - 1) /
                                                                                      align_long_queries(scope, spec.a.count, spec.b.count)),
                                       la_max, engine->scoring.open != engine->scoring.extend)) {
                // longer ones on the same small-alphabet condition: columns in passes of 128 (local or Gotoh: 64, both: 32), the boundary
                // column between passes through global memory (alignshort.hip: k_align_cross_long), up to where it beats the
                // column-profile kernel (align_long_pays)
                route = kRouteAlignLong;
                longest = la_max;
            } else if (known && (guaranteed || can_verify) && both <= 64) {
                // tokens of up to 64 bytes over any alphabet (multilingual words: ~5 code points are ~11 bytes, their tail reaches past 32):
                // the lane-per-pair kernel with a register row of 64 cells
                route = kRouteAlignShort;
                longest = both;
            } else if (known && can_verify && !scope->async) {
                // word tokens with a FEW long ones among them (a URL, a sentence of a script that writes no spaces): the lane kernel scores
                // every pair of two strings that fit its 64 cells, reports that some did not, and the redo plans only the pairs with a
                // longer string (`skip_upto`) -- instead of 4 M word pairs on kernels built for long strings (2048 x 2048 multilingual
                // words with one token of 70 bytes: 2.2 ms per call, NW linear). Taken when the mean string is word-sized.
                uint64_t mean_x16 = scope->hint_mean_string_x16;
                if (prepared) {
                    const uint64_t ma = spec.pa->bytes.count ? spec.pa->total_bytes * 16 / spec.pa->bytes.count : 0;
                    const uint64_t mb = spec.pb->bytes.count ? spec.pb->total_bytes * 16 / spec.pb->bytes.count : 0;
                    mean_x16 = std::max(ma, mb);
                }
                if (mean_x16 && mean_x16 <= 24 * 16) {
                    route = kRouteAlignShort;
                    longest = 64;
                }
            }
        }

        // -- scratch carving ------------------------------------------------------------------------
        size_t need = 0;
        {
            Carver probe{nullptr, 0, 0};
            if (route == kRoutePlanned) {
                probe.take<uint32_t>(pairs);            // perm
                probe.take<uint16_t>(pairs);            // plan keys
            }
            if (utf8 && !prepared) {
                probe.take<uint32_t>(a_bytes + 4); probe.take<uint64_t>(spec.a.count + 1);
                probe.take<uint32_t>(utf8_scratch_words(a_bytes));
                probe.take<uint32_t>(b_bytes + 4); probe.take<uint64_t>(spec.b.count + 1);
                probe.take<uint32_t>(utf8_scratch_words(b_bytes));
                probe.take<uint32_t>(kUtf8FlagWords);
                probe.take<uint64_t>(2 * spec.a.count + 2); probe.take<uint64_t>(2 * spec.b.count + 2);   // (first, end) pairs of the string-by-string staging
            }
            need = probe.used;
        }
        ensure(scope->scratch, scope->scratch_bytes, need);
        Carver sc{scope->scratch, 0, scope->scratch_bytes};
        uint32_t *perm = nullptr;
        uint16_t *plan_keys = nullptr;
        if (route == kRoutePlanned) {
            perm = sc.take<uint32_t>(pairs);
            plan_keys = sc.take<uint16_t>(pairs);
        }
        Plan *plan_dev = scope->plan_dev;

        // -- UTF-8 staging ----------------------------------------------------------------------------
        uint32_t *invalid_dev = nullptr;
        bool staged_by_string = false;
        if (utf8 && !prepared) {
            uint32_t decode_slot = 0;
            auto decode = [&](const TapeRef &in, uint64_t bytes, TapeRef &out_tape, uint64_t first_word, bool opened) {
                Utf8Args u{};
                u.slot = decode_slot++;
                u.in = in; u.off64 = off64; u.total_bytes = bytes;
                u.symbols = sc.take<uint32_t>(bytes + 4);
                u.offsets = sc.take<uint64_t>(in.count + 1);
                u.counts = sc.take<uint32_t>(utf8_scratch_words(bytes));
                u.invalid = invalid_dev;
                if (utf8_one_pass()) launch_utf8_decode_pair(scope, u, nullptr, first_word, opened);
                else launch_utf8_decode(scope, u);
                out_tape.data = u.symbols; out_tape.offsets = u.offsets; out_tape.count = in.count;
            };
            TapeRef da, db;
            invalid_dev = sc.take<uint32_t>(kUtf8FlagWords);   // one flag + two balance words + the tile tickets, shared by both decodes
            SWH_HIP_CHECK(hipMemsetAsync(invalid_dev, 0, kUtf8FlagWords * sizeof(uint32_t), stream));
            // Lines and longer strings are staged string by string (prepass.hip: k_utf8_strings -- one launch, no look-back; the code-point
            // tapes it leaves have gaps, TapeRef::gap): the routes whose kernels take their extents through pair_extent.
            const uint64_t all_strings = (uint64_t)spec.a.count + (same_tape ? 0 : spec.b.count), all_bytes = a_bytes + (same_tape ? 0 : b_bytes);
            if (scope->utf8_strings_rest) --scope->utf8_strings_rest;
            // (a too-long string sends the call back to the flat staging: the plan-free route can only do that where the host looks at the outcome)
            staged_by_string = (route == kRoutePlanned || (route == kRouteTiled && (!scope->async || !dev_out))) && utf8_strings_mode() != 2 && !spec.flat_staging &&
                               (utf8_strings_mode() == 1 || (scope->utf8_strings_rest == 0 && all_bytes >= (uint64_t)kUtf8StringsMeanBytes * all_strings));
            if (staged_by_string) {
                auto job_of = [&](const TapeRef &in, uint64_t bytes) {
                    Utf8StringsJob j{};
                    j.data = (const uint8_t *)in.data; j.offsets = in.offsets; j.count = in.count; j.total = bytes;
                    j.symbols = sc.take<uint32_t>(bytes + 4);
                    j.extents = sc.take<uint64_t>(2 * in.count + 2);
                    return j;
                };
                const Utf8StringsJob sa = job_of(ta, a_bytes);
                da.data = sa.symbols; da.offsets = sa.extents; da.count = ta.count; da.gap = 1;
                if (same_tape) {
                    launch_utf8_strings(scope, sa, nullptr, off64, invalid_dev);
                    db = da;
                } else {
                    const Utf8StringsJob sb = job_of(tb, b_bytes);
                    launch_utf8_strings(scope, sa, &sb, off64, invalid_dev);
                    db.data = sb.symbols; db.offsets = sb.extents; db.count = tb.count; db.gap = 1;
                }
            } else if (same_tape) {
                decode(ta, a_bytes, da, 0, false);
                db = da;
            } else if (utf8_one_pass() && a_bytes + b_bytes <= utf8_merged_bytes()) {
                // both tapes in the same two launches (tile decode, then string offsets + balance): what a small call costs is
                // its launches (10 K word pairs: 135 -> 107 us per call). Since the tile kernel draws from one ticket PER TAPE
                // (round 4) this is the path for every size (tools/mid_utf8.py, us per call, one launch pair : a launch pair and
                // a stream per tape -- 4 MB of tapes 186 : 205, 16 MB 216 : 240, 31 MB 256 : 283, 63 MB 330 : 352, 200 MB
                // 684 : 685); with one ticket for both tapes 2 x 100 MB took 0.33 ms in one launch against 0.27 in two.
                // STRINGWARS_AMD_UTF8_MERGED_MB=n sends tapes beyond n MB to the two-stream path below.
                auto prepare = [&](const TapeRef &in, uint64_t bytes, Utf8Args &u) {
                    u.slot = decode_slot++;
                    u.in = in; u.off64 = off64; u.total_bytes = bytes;
                    u.symbols = sc.take<uint32_t>(bytes + 4);
                    u.offsets = sc.take<uint64_t>(in.count + 1);
                    u.counts = sc.take<uint32_t>(utf8_scratch_words(bytes));
                    u.invalid = invalid_dev;
                };
                Utf8Args ua{}, ub{};
                prepare(ta, a_bytes, ua);
                prepare(tb, b_bytes, ub);
                launch_utf8_decode_pair(scope, ua, &ub, 0, false);
                da.data = ua.symbols; da.offsets = ua.offsets; da.count = ta.count;
                db.data = ub.symbols; db.offsets = ub.offsets; db.count = tb.count;
            } else {
                // The two tapes decode side by side: the staging kernels are barrier- and latency-bound (half the issue
                // slots idle), so the second tape's run on the side stream, forked after the inputs are in place.
                struct StreamSwap {   // launch_utf8_decode and its event stamps follow scope->stream
                    Scope *s; hipStream_t keep;
                    StreamSwap(Scope *sc_, hipStream_t to) : s(sc_), keep(sc_->stream) { s->stream = to; }
                    ~StreamSwap() { s->stream = keep; }
                };
                const uint64_t a_tiles = (a_bytes + kUtf8Tile - 1) / kUtf8Tile, b_tiles = (b_bytes + kUtf8Tile - 1) / kUtf8Tile;
                if (utf8_one_pass()) utf8_status_open(scope, a_tiles + b_tiles);   // before the fork: it may clear the words
                SWH_HIP_CHECK(hipEventRecord(scope->fork_ev, stream));
                SWH_HIP_CHECK(hipStreamWaitEvent(scope->side_stream, scope->fork_ev, 0));
                decode(ta, a_bytes, da, 0, true);
                {
                    StreamSwap swap(scope, scope->side_stream);
                    decode(tb, b_bytes, db, a_tiles, true);
                }
                SWH_HIP_CHECK(hipEventRecord(scope->join_ev, scope->side_stream));
                SWH_HIP_CHECK(hipStreamWaitEvent(stream, scope->join_ev, 0));
            }
            ta = da; tb = db;
            sym_bytes = 4; off64 = 1;
        }

        // -- job ----------------------------------------------------------------------------------------------
        Job job{};
        job.a = ta; job.b = tb; job.pairs = pairs; job.b_count = spec.b.count; job.cross = spec.cross ? 1 : 0;
        job.bound = engine->kind == 0 ? spec.bound : SWH_UNBOUNDED;
        job.out = out_dev; job.out_stride = dev_out_stride; job.row_stride = dev_row_stride;
        job.out_elem64 = spec.out64 ? 1 : 0;
        job.negate = engine->kind == 0 ? 1 : 0;
        if (spec.cross) {
            if (spec.b.count >= 0xFFFFFFFFull) return fail(error, swh_unsupported_length_k, "more than 2^32 candidates");
            cross_divider((uint32_t)spec.b.count, job.div_magic, job.div_shift);
        }

        PrepassArgs pre{};
        pre.job = job;
        pre.mode = bitpar_ok ? kPlanBitParallel : kPlanWavefront;
        pre.off64 = off64; pre.sym_bytes = sym_bytes;
        pre.symmetric = engine->kind == 0 ? 1u : (engine->unit_costs ? 1u : 0u);  // nw: set at init when symmetric
        pre.gap_open = engine->scoring.open; pre.gap_extend = engine->scoring.extend;
        pre.unit_costs = engine->kind == 0 && engine->unit_costs ? 1 : 0;
        pre.local = engine->kind == 2 ? 1 : 0;
        pre.direct_short = bitpar_ok && sym_bytes == 1 && engine->algorithm == swh_algorithm_auto_k && scope->hint_short ? 1 : 0;
        pre.skip_upto = spec.skip_upto;
        pre.banded = pre.unit_costs && spec.bound <= band_max_bound() && engine->algorithm == swh_algorithm_auto_k ? 1 : 0;
        pre.perm = perm; pre.keys = plan_keys; pre.hist = scope->plan_hist; pre.cursor = scope->plan_cursor;
        pre.partials = scope->plan_partials; pre.leftover = scope->plan_leftover; pre.plan = plan_dev;

        KernelArgs k{};
        k.job = job; k.perm = perm; k.plan = plan_dev; k.scoring = engine->scoring;
        k.off64 = off64; k.sym_bytes = sym_bytes; k.symmetric = pre.symmetric;
        k.affine = engine->scoring.open != engine->scoring.extend ? 1 : 0;
        k.local = engine->kind == 2 ? 1 : 0;

        uint32_t *invalid_host = (uint32_t *)(scope->plan_host + 1);
        *invalid_host = 0;
        auto copy_results_back = [&]() {
            if (dev_out) return;
            if (spec.cross) {
                SWH_HIP_CHECK(hipMemcpy2DAsync(spec.out, spec.row_stride, out_dev, dev_row_stride, spec.b.count * elem,
                                               spec.a.count, hipMemcpyDeviceToHost, stream));
            } else if (spec.out_stride == elem) {
                SWH_HIP_CHECK(hipMemcpyAsync(spec.out, out_dev, out_bytes, hipMemcpyDeviceToHost, stream));
            } else {
                SWH_HIP_CHECK(hipMemcpy2DAsync(spec.out, spec.out_stride, out_dev, elem, elem, pairs,
                                               hipMemcpyDeviceToHost, stream));
            }
        };
        // what the staging saw of the tapes' bytes (one-pass kernel only): the next call on the same tapes may believe it
        auto learn_ascii = [&]() {
            if (!invalid_dev || !utf8_one_pass()) return;
            if (same_as_believed(spec.a, scope->size_belief[0])) scope->size_belief[0].ascii = invalid_host[kUtf8AsciiWord] == 0;
            if (!same_tape && same_as_believed(spec.b, scope->size_belief[1])) scope->size_belief[1].ascii = invalid_host[kUtf8AsciiWord + 1] == 0;
        };
        auto invalid_utf8 = [&]() -> swh_status_t {
            SWH_HIP_CHECK(hipStreamSynchronize(stream));
            if (*invalid_host == kUtf8SizesChanged && believed_sizes) {
                // the tapes changed behind the belief: read their totals afresh and do the call again
                scope->size_belief[0].valid = scope->size_belief[1].valid = false;
                scope->summary_pending = false;
                scope->stamps_pending = false;
                return run_call_on(scope, engine, spec, error);
            }
            if (*invalid_host == kUtf8StringTooLong && staged_by_string) {
                // a string too long for a wave of its own (k_utf8_strings): this call and the scope's next few stage the flat way
                scope->utf8_strings_rest = 16;
                scope->summary_pending = false;
                scope->stamps_pending = false;
                CallSpec flat = spec;
                flat.flat_staging = true;
                return run_call_on(scope, engine, flat, error);
            }
            snprintf(g_error_text, sizeof g_error_text, "invalid UTF-8 in an input tape (marker %u: the string's index when staged string by string, else a 4-byte word inside the 1 KiB tile that failed)", *invalid_host - 1);
            if (error) *error = g_error_text;
            return swh_invalid_utf8_k;
        };

        if (route != kRoutePlanned) {
            // ---- no pre-pass: one DP launch; its summary (work units, longest strings, "a pair did not fit") arrives in
            // host-mapped memory with the kernel's completion ------------------------------------------------------------
            // (an asynchronous call reports into slot 1, whose `sticky` word outlives the summary: see CallSummary)
            scope->summary_slot = (scope->async && dev_out) ? 1u : 0u;
            const bool early = !scope->async && dev_out && !scope->profiling && !invalid_dev && early_return_on() && is_plain_device_memory(spec.out, scope->device);
            if (early) scope->summary_host[0].landed = 0;
            if (route == kRouteDirectShort) launch_direct_short_alone(scope, pre);
            else if (route == kRouteShortTiled) {
                // mean string length, for the chunk size: exact for prepared tapes (their totals), else what the last call saw
                uint32_t mean_x16 = scope->hint_mean_x16;
                if (prepared) {
                    const uint64_t ma = spec.pa->bytes.count ? spec.pa->total_bytes * 16 / spec.pa->bytes.count : 0;
                    const uint64_t mb = spec.pb->bytes.count ? spec.pb->total_bytes * 16 / spec.pb->bytes.count : 0;
                    mean_x16 = (uint32_t)std::min<uint64_t>(std::max(ma, mb), 0xFFFFFFu);
                }
                launch_short_tiled(scope, job, off64, mean_x16);
            }
            else if (route == kRouteCrossShort) launch_cross_short(scope, job, off64, (uint32_t)sym_bytes);
            else if (route == kRouteAlignShort) launch_align_short(scope, k, longest, align_wide);
            else if (route == kRouteAlignLong) {
                const uint32_t per_item = align_long_queries(scope, spec.a.count, spec.b.count);
                const uint64_t items = ((uint64_t)(spec.b.count + 63) / 64) * ((uint64_t)(spec.a.count + per_item - 1) / per_item);
                const uint64_t ints = (uint64_t)align_long_waves(scope, items, longest, k.affine != 0) * (longest + 8) * 64 * (k.affine ? 2 : 1);
                ensure(scope->boundary, scope->boundary_bytes, ints * sizeof(int32_t));
                k.boundary = (int32_t *)scope->boundary;
                launch_align_long(scope, k, longest);
            }
            else launch_bitparallel_tiled(scope, k, pairs, longest);
            if (invalid_dev) SWH_HIP_CHECK(hipMemcpyAsync(invalid_host, invalid_dev, 4 * (kUtf8AsciiWord + 2), hipMemcpyDeviceToHost, stream));
            copy_results_back();
            scope->summary_sym_bytes = need_sizes ? 0 : sym_bytes;
            scope->summary_pairs = pairs; scope->summary_ow = ow; scope->summary_elem = elem;
            scope->summary_strings = spec.cross ? (uint64_t)spec.a.count + spec.b.count : 2 * (uint64_t)pairs;
            scope->summary_extra_bytes = need_sizes ? a_bytes + b_bytes : 0;
            scope->summary_pending = true;
            scope->stamps_pending = scope->profiling;
            if (!scope->async || !dev_out) {
                if (early) wait_for_summary(scope, stream);
                else SWH_HIP_CHECK(hipStreamSynchronize(stream));
                if (*invalid_host) return invalid_utf8();
                learn_ascii();
                if (scope->summary_host[0].violation) {
                    // the belief about the lengths was wrong (it came from an earlier batch): redo on the planned path
                    scope->hint_lengths = false;
                    // the compacting kernels stay off this scope only when the ALPHABET was the reason (violation bit 1); a string longer
                    // than the believed lengths just drops the belief, and the next batch of the same shape is routed afresh
                    const bool compact_route = (route == kRouteAlignShort && align_wide) || route == kRouteAlignLong;
                    const bool compact_failed = compact_route && (scope->summary_host[0].violation & 2u) != 0;
                    if (compact_failed) {
                        scope->align_wide_off.engine = engine->uid;
                        scope->align_wide_off.a = spec.pa ? (const void *)spec.pa : (const void *)spec.a.data;
                        scope->align_wide_off.b = spec.pb ? (const void *)spec.pb : (const void *)spec.b.data;
                    }
                    scope->summary_pending = false;
                    scope->stamps_pending = false;
                    CallSpec again = spec;
                    // (prepared tapes know their lengths: when only the small-alphabet kernels' condition failed, the redo may still take
                    // the lane-per-pair kernel -- `align_wide_off` keeps it off the compacting ones)
                    again.force_planned = !(compact_failed && prepared);
                    // the lane-per-pair kernel has scored every pair whose two strings fit its register row: the redo plans the others
                    if (route == kRouteAlignShort && !align_wide) again.skip_upto = longest <= 16 ? 16u : (longest <= 32 ? 32u : 64u);
                    return run_call_on(scope, engine, again, error);
                }
                harvest_timing(scope, true);
            }
            return swh_success_k;
        }

        // ---- planned path: classify + counting sort on the device, then the DP kernels the plan's classes call for ----
        // Doubling. A unit-cost call whose bound lies beyond one band word (or that has none) first runs the ONE-WORD band at k1 = 63 over
        // the pairs long enough for it: whatever comes back <= 63 is the distance, under any larger bound, and only the pairs that
        // came back 64 are planned again with the call's own bound (two-word band, bit-parallel blocks). The reference's CPU row does
        // the same inside every call -- rapidfuzz doubles a score hint of 31 until the result fits under it -- and text that is compared
        // for similarity mostly is similar: config C3's lines, unbounded, 26 -> ~50 TCUPS. The first stage costs band_cost(63) per
        // column where the second costs `later`; the share of pairs it has to settle for that to pay (plus a quarter: a second plan, a
        // second tail) is held against what the scope's previous doubling call saw, and a scope that saw less sits eight calls out.
        bool doubling = false;
        double doubling_need = 0;
        constexpr uint32_t kDoublingBound = 63;
        static const bool doubling_on = [] { const char *e = test_hook("STRINGWARS_AMD_DOUBLING"); return !e || atoi(e) != 0; }();
        static const uint64_t doubling_min = [] { const char *e = test_hook("STRINGWARS_AMD_DOUBLING_MIN"); return e ? (uint64_t)atoll(e) : (uint64_t)200000; }();   // tuning knob: pairs x blocks
        if (doubling_on && bitpar_ok && pre.unit_costs && engine->algorithm == swh_algorithm_auto_k && spec.bound > kDoublingBound &&
            (prepared || scope->hint_lengths)) {
            const uint32_t la_max = prepared ? (utf8 ? spec.pa->longest_symbols : spec.pa->longest_bytes) : scope->hint_max_la;
            const uint32_t lb_max = prepared ? (utf8 ? spec.pb->longest_symbols : spec.pb->longest_bytes) : scope->hint_max_lb;
            const uint32_t blocks = (std::min(la_max, lb_max) + 31) >> 5;
            const uint32_t unbounded_cost = (sym_bytes == 4 ? 40u : 28u) * blocks;
            const uint32_t later = spec.bound <= band_max_bound() ? std::min(band_cost(spec.bound), unbounded_cost) : unbounded_cost;
            doubling_need = 1.25 * band_cost(kDoublingBound) / std::max(later, 1u);
            if (scope->doubling_rest) --scope->doubling_rest;
            else doubling = doubling_need <= 0.95 && (uint64_t)pairs * blocks >= doubling_min;   // (a second plan and a second tail: not for small batches)
        }
        if (doubling) {
            PrepassArgs first = pre;
            first.job.bound = kDoublingBound; first.banded = 1; first.stage1 = 1; first.direct_short = 0; first.skip_upto = 0;
            launch_prepass(scope, first);
            KernelArgs kf = k;
            kf.job.bound = kDoublingBound;
            launch_banded(scope, kf, pairs);
            pre.redo_filter = 1; pre.redo_done_upto = kDoublingBound;
            // What is left after a first stage that settles nearly everything is a few thousand pairs: too few to fill the device with
            // the two-word band's items (one wave walks a pair's every column: C3's lines at k = 100, 2 % left over, 0.19 ms), while a
            // bit-parallel item spreads ONE pair over the lanes of its blocks (0.07 ms). Where the scope's previous doubling call left
            // less than a tenth, the second stage plans without the band (the results are clamped to the bound either way).
            if (scope->doubling_settled >= 0.9f) pre.banded = 0;
        }
        launch_prepass(scope, pre);
        // The bit-parallel kernel reads its work list from the device plan, so it is enqueued right away;
        // the host copy of the plan (needed only to pick wavefront kernels) travels on a side stream and
        // overlaps it.
        SWH_HIP_CHECK(hipEventRecord(scope->plan_ready, stream));
        if (pre.banded) launch_banded(scope, k, pairs);
        // Bounded calls: the banded kernel keeps the device busy while the plan travels, so the bit-parallel launch waits until
        // the host knows whether it has any pairs at all (C3: none -- an empty launch and its gap were 8 us of a 0.4 ms call).
        const bool bitpar_deferred = bitpar_ok && pre.banded;
        if (bitpar_ok && !bitpar_deferred) launch_bitparallel(scope, k, pairs);
        Plan &plan = *scope->plan_host;
        SWH_HIP_CHECK(hipStreamWaitEvent(scope->side_stream, scope->plan_ready, 0));
        SWH_HIP_CHECK(hipMemcpyAsync(&plan, plan_dev, sizeof(Plan), hipMemcpyDeviceToHost, scope->side_stream));
        if (invalid_dev)
            SWH_HIP_CHECK(hipMemcpyAsync(invalid_host, invalid_dev, 4 * (kUtf8AsciiWord + 2), hipMemcpyDeviceToHost, scope->side_stream));
        SWH_HIP_CHECK(hipStreamSynchronize(scope->side_stream));
        if (plan.fused_failed) {
            // the one-launch planner could not gather its grid (a device shared with long-running foreign kernels): the DP
            // kernels found an empty plan; plan again with the three passes, now and from here on
            SWH_HIP_CHECK(hipStreamSynchronize(stream));
            scope->fused_disabled = true;
            scope->stamps_pending = false;
            return run_call_on(scope, engine, spec, error);
        }
        // enqueue k_direct_short next time only if short pairs are a real share of the batch (it sweeps all offsets)
        scope->hint_short = (uint64_t)plan.short_pairs * 4 >= pairs;
        scope->hint_lengths = true;
        scope->hint_max_la = plan.max_la; scope->hint_max_lb = plan.max_lb;
        {
            const uint64_t strings = 2 * (uint64_t)pairs;   // (the planner sums la + lb over PAIRS, also for a cross-product)
            scope->hint_mean_string_x16 = strings ? (uint32_t)std::min<uint64_t>(plan.symbols * 16 / strings, 0xFFFFFFu) : 0u;
        }
        if (*invalid_host) return invalid_utf8();
        learn_ascii();
        if (doubling) {
            // what the first stage left over: the pairs the second plan filed under a kernel class
            const uint64_t redo = plan.class_start[kMaxClasses] - plan.class_count[kClassTrivial];
            const double settled = 1.0 - (double)redo / (double)pairs;
            scope->doubling_settled = (float)settled;
            if (settled < doubling_need) scope->doubling_rest = 8;
        }

        if (bitpar_deferred) {
            bool any_bp = false;
            for (int c = kClassBp0; c < kClassBp0 + 64; ++c) any_bp |= plan.class_count[c] != 0;
            if (any_bp) launch_bitparallel(scope, k, pairs);
        }
        // patterns of more than 64 blocks: multi-pass bit-parallel kernel, carries between passes in scratch
        if (bitpar_ok && plan.class_count[kClassBpLong]) {
            const uint64_t stride = bp_long_carry_words(plan.max_la > plan.max_lb ? plan.max_la : plan.max_lb);
            // one carry area per wave the launch can have: ceil(count / waves-per-block) blocks of 4 (bytes) or 11 (code points) waves
            const uint64_t waves = plan.class_count[kClassBpLong] + 16 < kBpLongMaxWaves ? plan.class_count[kClassBpLong] + 16 : kBpLongMaxWaves;
            ensure(scope->boundary, scope->boundary_bytes, waves * stride * sizeof(uint32_t));
            KernelArgs kl = k;
            kl.boundary = (int32_t *)scope->boundary;
            kl.boundary_stride = stride;
            launch_bitparallel_long(scope, kl, plan);
        }

        // Global or local alignment on a class table (<= 32 symbol classes; 33 .. 128 on the wide table), pairs of more than 384 columns: the column-profile kernel
        // (nwprofile.hip) takes them -- perm is sorted by class, so they are one contiguous range -- and the wavefront
        // kernels below see a plan without them. STRINGWARS_AMD_NW=classic keeps everything on the wavefront kernels.
        static const bool nw_classic = [] { const char *e = test_hook("STRINGWARS_AMD_NW"); return e && strcmp(e, "classic") == 0; }();
        uint32_t profile_first = 0, profile_count = 0;
        Plan wf_plan = plan;
        if ((engine->kind == 1 || engine->kind == 2) && (engine->scoring.class_table || engine->scoring.wide_table) && sym_bytes == 1 && !nw_classic) {
            profile_first = plan.class_start[kClassWf64 + kNwProfileFirstWide];
            for (int c = kClassWf64 + kNwProfileFirstWide; c <= kClassWfMulti; ++c) { profile_count += plan.class_count[c]; wf_plan.class_count[c] = 0; }
        }
        // wavefront classes (all of them when the plan is wavefront-only)
        bool any_wf = false, multi = false;
        for (int c = kClassWf16; c <= kClassWfMulti; ++c) {
            if (!wf_plan.class_count[c]) continue;
            any_wf = true;
            if (c == kClassWfMulti || (k.affine && c >= kClassWf64 + 8)) multi = true;  // (affine strips are capped: the widest classes take several passes)
            if (wavefront_strip_cap() && c >= kClassWf64 && wide_w(c - kClassWf64 < kNumWideW ? c - kClassWf64 : kNumWideW - 1) > wavefront_strip_cap()) multi = true;
        }
        if (any_wf || profile_count) {
            // int32 scores with a -2^29 "minus infinity": keep every reachable score well inside it
            const uint64_t worst_step = std::max<uint64_t>({(uint64_t)std::abs(engine->scoring.open), (uint64_t)std::abs(engine->scoring.extend),
                                                            (uint64_t)std::abs(engine->scoring.match), (uint64_t)std::abs(engine->scoring.mismatch),
                                                            engine->scoring.matrix ? 128u : 0u});
            if (worst_step * ((uint64_t)plan.max_la + plan.max_lb + 2) >= 0x10000000ull) {
                SWH_HIP_CHECK(hipStreamSynchronize(stream));
                return fail(error, swh_unsupported_length_k, "scores of this batch could leave the 32-bit range of the wavefront kernels (costs x lengths too large)");
            }
            // one boundary column (H, E) per concurrently resident group: two areas for the wavefront class kernels, which
            // alternate between two streams and run side by side (launch_wavefront), one for the profile kernel's waves
            const uint64_t stride = (uint64_t)(plan.max_la > plan.max_lb ? plan.max_la : plan.max_lb) + 64 + 16;
            const uint64_t groups = multi ? (uint64_t)scope->compute_units * 8 * 4 : 0;  // max blocks * waves (G = 64)
            // (one area per wave the profile launch can have: launch_nwprofile never starts more workgroups than it has pairs)
            const uint64_t profile_waves = profile_count ? std::min<uint64_t>(nwprofile_waves(scope, engine->scoring.classes ? engine->scoring.classes : 32), profile_count) : 0;
            if (multi || profile_count) ensure(scope->boundary, scope->boundary_bytes, (2 * groups + profile_waves) * stride * 2 * sizeof(int32_t));
            if (profile_count) {
                KernelArgs kp = k;
                kp.boundary = (int32_t *)scope->boundary + 2 * groups * stride * 2;
                kp.boundary_stride = stride;
                if (kp.local && engine->scoring.step_span) {
                    // Scoring::step_span = largest |cost| - open - extend (alignment_init): a local score is at most the largest cost x the shorter string
                    const uint64_t largest_cost = (uint64_t)((int64_t)engine->scoring.step_span + engine->scoring.open + engine->scoring.extend);
                    kp.local_narrow = largest_cost * (uint64_t)std::min(plan.max_la, plan.max_lb) < 0xFFFFull ? 1u : 0u;
                }
                launch_nwprofile(scope, kp, profile_first, profile_count);
            }
            if (multi) {
                k.boundary = (int32_t *)scope->boundary;
                k.boundary_stride = stride;
                scope->wf_side_boundary = groups * stride * 2;
            }
            if (any_wf) launch_wavefront(scope, k, wf_plan);
        }

        copy_results_back();
        scope->last_timing.cells = plan.cells;
        scope->last_timing.bytes = (need_sizes ? a_bytes + b_bytes : plan.symbols * sym_bytes) + pairs * (2 * ow + elem);
        scope->stamps_pending = scope->profiling;
        if (!scope->async || !dev_out) {
            SWH_HIP_CHECK(hipStreamSynchronize(stream));
            harvest_timing(scope, true);
        }
        return swh_success_k;
    } catch (const HipFailure &f) {
        return fail_hip(error, f);
    } catch (const std::bad_alloc &) {
        return fail(error, swh_bad_alloc_k, "host allocation failed");
    }
}

}  // namespace swh

using namespace swh;

// =====================================================================================================
// extern "C"
// =====================================================================================================
extern "C" {

const char *swh_version(void) { return "0.1.0"; }
const char *swh_capabilities(void) { return "gfx950,hip,wavefront,bitparallel,tiled,banded,utf8,bounded,nw-linear,nw-affine,sw-linear,sw-affine,cross,prepared,multi-gpu-rccl"; }

static swh_status_t scope_init(int device, void *stream, bool borrow, swh_scope_t *out, const char **error) {
    if (!out) return fail(error, swh_invalid_argument_k, "null scope pointer");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(error, swh_no_device_k, "no HIP device visible (this backend has no CPU path)");
    }
    if (device < 0 || device >= count) return fail(error, swh_no_device_k, "HIP device index out of range");
    try {
        SWH_HIP_CHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        SWH_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(error, swh_no_device_k, "device is %s, this library carries gfx950 code only", prop.gcnArchName);
        Scope *scope = new Scope();
        scope->device = device;
        scope->compute_units = prop.multiProcessorCount;
        if (borrow) { scope->stream = (hipStream_t)stream; scope->owns_stream = false; }
        else { SWH_HIP_CHECK(hipStreamCreateWithFlags(&scope->stream, hipStreamNonBlocking)); scope->owns_stream = true; }
        SWH_HIP_CHECK(hipHostMalloc((void **)&scope->plan_host, sizeof(Plan) + 64, hipHostMallocDefault));
        {   // plan area: hist | cursor | partials | leftover + done counter | plan; zeroed once (the kernels re-zero what they use)
            Carver measure{nullptr, 0, 0};
            measure.take<uint32_t>(kKeys); measure.take<uint32_t>(kKeys); measure.take<PlanPartial>(2 * kMaxPartials);
            measure.take<uint32_t>(8); measure.take<Plan>(1);
            measure.take<uint32_t>(kKeys); measure.take<uint32_t>(kKeys); measure.take<uint32_t>(4);
            SWH_HIP_CHECK(hipMalloc((void **)&scope->plan_area, measure.used));
            SWH_HIP_CHECK(hipMemset(scope->plan_area, 0, measure.used));
            Carver pa{scope->plan_area, 0, measure.used};
            scope->plan_hist = pa.take<uint32_t>(kKeys);
            scope->plan_cursor = pa.take<uint32_t>(kKeys);
            scope->plan_partials = pa.take<PlanPartial>(2 * kMaxPartials);
            scope->plan_leftover = pa.take<uint32_t>(8);
            scope->done_counter = scope->plan_leftover + 4;
            scope->plan_dev = pa.take<Plan>(1);
            scope->plan_hist2[0] = pa.take<uint32_t>(kKeys);
            scope->plan_hist2[1] = pa.take<uint32_t>(kKeys);
            scope->plan_barrier = pa.take<uint32_t>(4);
        }
        // summary of plan-free calls: pinned, mapped, coherent -- the kernels write it, the host reads it after synchronising
        SWH_HIP_CHECK(hipHostMalloc((void **)&scope->summary_host, 256, hipHostMallocMapped | hipHostMallocCoherent));
        memset(scope->summary_host, 0, 256);
        SWH_HIP_CHECK(hipHostGetDevicePointer((void **)&scope->summary_dev, scope->summary_host, 0));
        SWH_HIP_CHECK(hipStreamCreateWithFlags(&scope->side_stream, hipStreamNonBlocking));
        SWH_HIP_CHECK(hipEventCreateWithFlags(&scope->plan_ready, hipEventDisableTiming));
        SWH_HIP_CHECK(hipEventCreateWithFlags(&scope->fork_ev, hipEventDisableTiming));
        SWH_HIP_CHECK(hipEventCreateWithFlags(&scope->join_ev, hipEventDisableTiming));
        *out = (swh_scope_t)scope;
        return swh_success_k;
    } catch (const HipFailure &f) {
        return fail_hip(error, f);
    }
}

swh_status_t swh_scope_init_gpu(int device, swh_scope_t *scope, const char **error) {
    return scope_init(device, nullptr, false, scope, error);
}
swh_status_t swh_scope_init_gpu_stream(int device, void *hip_stream, swh_scope_t *scope, const char **error) {
    return scope_init(device, hip_stream, true, scope, error);
}
swh_status_t swh_scope_init_cpu(size_t, swh_scope_t *scope, const char **error) {
    if (scope) *scope = nullptr;
    return fail(error, swh_not_implemented_k, "stringwars_amd has no CPU backend; use a GPU scope");
}
swh_status_t swh_scope_free(swh_scope_t handle) {
    Scope *scope = (Scope *)handle;
    if (!scope) return swh_success_k;
    for (Scope *&lane : scope->lanes)
        if (lane) { swh_scope_free((swh_scope_t)lane); lane = nullptr; }
    if (scope->multi) { free_multi_scope(scope->multi); scope->multi = nullptr; }
    (void)hipSetDevice(scope->device);
    (void)hipStreamSynchronize(scope->stream);
    if (scope->lane_done) (void)hipEventDestroy(scope->lane_done);
    if (scope->order_ev) (void)hipEventDestroy(scope->order_ev);
    for (auto &st : scope->stamps) { (void)hipEventDestroy(st.start); (void)hipEventDestroy(st.stop); }
    if (scope->scratch) (void)hipFree(scope->scratch);
    if (scope->utf8_status) (void)hipFree(scope->utf8_status);
    if (scope->stage) (void)hipFree(scope->stage);
    if (scope->boundary) (void)hipFree(scope->boundary);
    if (scope->plan_host) (void)hipHostFree(scope->plan_host);
    if (scope->summary_host) (void)hipHostFree(scope->summary_host);
    if (scope->plan_area) (void)hipFree(scope->plan_area);
    if (scope->side_stream) (void)hipStreamDestroy(scope->side_stream);
    if (scope->fork_ev) (void)hipEventDestroy(scope->fork_ev);
    if (scope->join_ev) (void)hipEventDestroy(scope->join_ev);
    if (scope->plan_ready) (void)hipEventDestroy(scope->plan_ready);
    if (scope->owns_stream) (void)hipStreamDestroy(scope->stream);
    delete scope;
    return swh_success_k;
}
swh_status_t swh_scope_compute_units(swh_scope_t handle, size_t *cus) {
    if (!handle || !cus) return swh_invalid_argument_k;
    *cus = (size_t)((Scope *)handle)->compute_units;
    return swh_success_k;
}
swh_status_t swh_scope_set_async(swh_scope_t handle, int async) {
    if (!handle) return swh_invalid_argument_k;
    ((Scope *)handle)->async = async != 0;
    return swh_success_k;
}
swh_status_t swh_scope_synchronize(swh_scope_t handle, const char **error) {
    Scope *scope = (Scope *)handle;
    if (!scope) return fail(error, swh_invalid_argument_k, "null scope");
    bool violated = false;
    for (Scope *lane : scope->lanes)
        if (lane) {
            hipError_t lerr = hipStreamSynchronize(lane->stream);
            if (lerr != hipSuccess) return fail_hip(error, HipFailure{lerr, "hipStreamSynchronize (lane)"});
            harvest_timing(lane, true);
            violated |= lane->violation_seen; lane->violation_seen = false;
        }
    hipError_t err = hipStreamSynchronize(scope->stream);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "hipStreamSynchronize"});
    harvest_timing(scope, true);
    violated |= scope->violation_seen; scope->violation_seen = false;
    // every asynchronous plan-free call since the last synchronisation, not only the one whose summary was still there to read
    for (Scope *each : {scope, scope->lanes[0], scope->lanes[1]})
        if (each && each->summary_host && each->summary_host[1].sticky) { violated = true; each->summary_host[1].sticky = 0; }
    if (scope->pipelined && scope->last_lane) scope->last_timing = scope->last_lane->last_timing;
    if (violated)
        return fail(error, swh_invalid_argument_k, "an asynchronous call met strings longer than its prepared tapes were measured with (was a tape's memory "
                                                   "changed after swh_tape_prepare_*?): those pairs were not scored");
    return swh_success_k;
}
swh_status_t swh_scope_set_pipelined(swh_scope_t handle, int enabled, const char **error) {
    Scope *scope = (Scope *)handle;
    if (!scope) return fail(error, swh_invalid_argument_k, "null scope");
    swh_status_t status = swh_scope_synchronize(handle, error);
    if (status != swh_success_k) return status;
    if (enabled) {
        if (!scope->order_ev && hipEventCreateWithFlags(&scope->order_ev, hipEventDisableTiming) != hipSuccess)
            return fail(error, swh_device_error_k, "hipEventCreate failed for the pipeline");
        for (Scope *&lane : scope->lanes) {
            if (lane) continue;
            swh_scope_t created = nullptr;
            status = swh_scope_init_gpu(scope->device, &created, error);   // own non-blocking stream + buffers
            if (status != swh_success_k) return status;
            lane = (Scope *)created;
            lane->async = true;
            lane->profiling = scope->profiling;
            if (hipEventCreateWithFlags(&lane->lane_done, hipEventDisableTiming) != hipSuccess)
                return fail(error, swh_device_error_k, "hipEventCreate failed for a pipeline lane");
        }
    }
    scope->pipelined = enabled != 0;
    scope->last_lane = nullptr;
    return swh_success_k;
}
swh_status_t swh_scope_join(swh_scope_t handle, const char **error) {
    Scope *scope = (Scope *)handle;
    if (!scope) return fail(error, swh_invalid_argument_k, "null scope");
    if (!scope->pipelined || !scope->last_lane) return swh_success_k;   // calls already ran on the scope's own stream
    hipError_t err = hipStreamWaitEvent(scope->stream, scope->last_lane->lane_done, 0);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "hipStreamWaitEvent"});
    return swh_success_k;
}
swh_status_t swh_scope_forget(swh_scope_t handle) {
    if (!handle) return swh_invalid_argument_k;
    Scope *scope = (Scope *)handle;
    Scope *all[3] = {scope, scope->lanes[0], scope->lanes[1]};
    for (Scope *s : all) {
        if (!s) continue;
        s->hint_short = true;
        s->hint_lengths = false;
        s->hint_max_la = s->hint_max_lb = 0;
        s->hint_mean_x16 = s->hint_mean_string_x16 = 0;
        s->size_belief[0] = Scope::SizeBelief{};
        s->size_belief[1] = Scope::SizeBelief{};
        s->doubling_settled = -1.0f;
        s->doubling_rest = 0;
        s->utf8_strings_rest = 0;
        s->align_wide_off = Scope::AlignWideOff{};
        s->early_return_last_us = 0;
    }
    return swh_success_k;
}
swh_status_t swh_scope_describe(swh_scope_t handle, char *text, size_t capacity) {
    if (!handle || !text || !capacity) return swh_invalid_argument_k;
    const Scope *s = (const Scope *)handle;
    snprintf(text, capacity,
             "lengths_believed=%d longest_a=%u longest_b=%u mean_string_x16=%u short_pairs_expected=%d "
             "utf8_tape0=%s%s utf8_tape1=%s%s doubling_settled=%.3f doubling_rest=%u utf8_strings_rest=%u "
             "align_wide_off_engine=%llu fused_planner=%s",
             s->hint_lengths ? 1 : 0, s->hint_max_la, s->hint_max_lb, s->hint_mean_string_x16, s->hint_short ? 1 : 0,
             s->size_belief[0].valid ? "sized" : "unknown", s->size_belief[0].ascii ? "+ascii" : "",
             s->size_belief[1].valid ? "sized" : "unknown", s->size_belief[1].ascii ? "+ascii" : "",
             (double)s->doubling_settled, s->doubling_rest, s->utf8_strings_rest,
             (unsigned long long)s->align_wide_off.engine, s->fused_disabled ? "off" : "on");
    return swh_success_k;
}
swh_status_t swh_scope_set_profiling(swh_scope_t handle, int enabled) {
    if (!handle) return swh_invalid_argument_k;
    Scope *scope = (Scope *)handle;
    Scope *all[3] = {scope, scope->lanes[0], scope->lanes[1]};
    for (Scope *s : all) {
        if (!s) continue;
        if (enabled && !s->profiling) { s->totals = swh_timing_totals_t{}; s->stamps_pending = false; }
        s->profiling = enabled != 0;
    }
    return swh_success_k;
}
swh_status_t swh_scope_timing_totals(swh_scope_t handle, swh_timing_totals_t *totals) {
    if (!handle || !totals) return swh_invalid_argument_k;
    Scope *scope = (Scope *)handle;
    *totals = swh_timing_totals_t{};
    Scope *all[3] = {scope, scope->lanes[0], scope->lanes[1]};
    for (Scope *s : all) {
        if (!s) continue;
        totals->total_ms += s->totals.total_ms; totals->dominant_ms += s->totals.dominant_ms;
        totals->compute_ms += s->totals.compute_ms; totals->calls += s->totals.calls;
    }
    return swh_success_k;
}
swh_status_t swh_scope_last_timing(swh_scope_t handle, swh_timing_t *timing) {
    if (!handle || !timing) return swh_invalid_argument_k;
    *timing = ((Scope *)handle)->last_timing;
    return swh_success_k;
}

// ---- memory --------------------------------------------------------------------------------------
swh_status_t swh_unified_alloc(swh_scope_t handle, size_t bytes, void **pointer, const char **error) {
    if (!handle || !pointer) return fail(error, swh_invalid_argument_k, "null argument");
    (void)hipSetDevice(((Scope *)handle)->device);
    hipError_t err = hipHostMalloc(pointer, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "hipHostMalloc"});
    return swh_success_k;
}
swh_status_t swh_unified_free(swh_scope_t, void *pointer) {
    if (pointer) (void)hipHostFree(pointer);
    return swh_success_k;
}
swh_status_t swh_device_alloc(swh_scope_t handle, size_t bytes, void **pointer, const char **error) {
    if (!handle || !pointer) return fail(error, swh_invalid_argument_k, "null argument");
    (void)hipSetDevice(((Scope *)handle)->device);
    hipError_t err = hipMalloc(pointer, bytes ? bytes : 1);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "hipMalloc"});
    return swh_success_k;
}
swh_status_t swh_device_free(swh_scope_t, void *pointer) {
    if (pointer) (void)hipFree(pointer);
    return swh_success_k;
}
swh_status_t swh_copy_to_device(swh_scope_t handle, void *dst, const void *src, size_t bytes, const char **error) {
    Scope *scope = (Scope *)handle;
    if (!scope) return fail(error, swh_invalid_argument_k, "null scope");
    hipError_t err = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, scope->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(scope->stream);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "hipMemcpy H2D"});
    return swh_success_k;
}
swh_status_t swh_copy_to_host(swh_scope_t handle, void *dst, const void *src, size_t bytes, const char **error) {
    Scope *scope = (Scope *)handle;
    if (!scope) return fail(error, swh_invalid_argument_k, "null scope");
    hipError_t err = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, scope->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(scope->stream);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "hipMemcpy D2H"});
    return swh_success_k;
}

// ---- prepared tapes ----------------------------------------------------------------------------------
static void free_prepared(Prepared *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void *buffer : p->owned) (void)hipFree(buffer);
    delete p;
}

static swh_status_t prepare_tape(Scope *scope, const HostTape &tape, bool utf8, swh_prepared_t *out, const char **error) {
    if (!scope || !out) return fail(error, swh_invalid_argument_k, "null argument");
    *out = nullptr;
    if (scope->pipelined) return fail(error, swh_invalid_argument_k, "prepare tapes before switching the scope to pipelined mode");
    Prepared *p = new Prepared();
    try {
        SWH_HIP_CHECK(hipSetDevice(scope->device));
        hipStream_t stream = scope->stream;
        p->device = scope->device;
        p->utf8 = utf8;
        p->off64 = (uint32_t)tape.off64;
        const size_t ow = tape.off64 ? 8 : 4;
        const bool dev_data = is_device_pointer(tape.data), dev_off = is_device_pointer(tape.offsets);
        p->total_bytes = tape.count || tape.offsets ? read_offset(tape.offsets, tape.off64, tape.count, dev_off, stream) : 0;
        auto own = [&](size_t bytes) -> void * {
            void *buffer = nullptr;
            SWH_HIP_CHECK(hipMalloc(&buffer, bytes ? bytes : 16));
            p->owned.push_back(buffer);
            return buffer;
        };
        // residency: host tapes are uploaded once (`BytesTape<u64, UnifiedAlloc>::extend`, bench.rs:292-301)
        const void *data = tape.data, *offsets = tape.offsets;
        if (!dev_data) {
            void *buffer = own(p->total_bytes + 16);
            if (p->total_bytes) SWH_HIP_CHECK(hipMemcpyAsync(buffer, tape.data, p->total_bytes, hipMemcpyHostToDevice, stream));
            data = buffer;
        }
        if (!dev_off) {
            void *buffer = own((tape.count + 1) * ow + 16);
            SWH_HIP_CHECK(hipMemcpyAsync(buffer, tape.offsets, (tape.count + 1) * ow, hipMemcpyHostToDevice, stream));
            offsets = buffer;
        }
        p->bytes = TapeRef{data, offsets, tape.count};
        // measurements: longest string in bytes (and in code points below)
        uint32_t *words = (uint32_t *)own((4 + kUtf8FlagWords) * sizeof(uint32_t));   // [0] longest bytes, [1] longest symbols, [4 ...] UTF-8 flag, balances, tickets
        SWH_HIP_CHECK(hipMemsetAsync(words, 0, (4 + kUtf8FlagWords) * sizeof(uint32_t), stream));
        launch_tape_longest(scope, offsets, p->off64, tape.count, words);
        uint64_t total_symbols = p->total_bytes;
        if (utf8) {
            // validate + decode once: `CharsTapeView::try_from` (bench.rs:303-306)
            Utf8Args u{};
            u.in = p->bytes; u.off64 = p->off64; u.total_bytes = p->total_bytes; u.slot = 0;
            u.symbols = (uint32_t *)own((p->total_bytes + 4) * sizeof(uint32_t));
            u.offsets = (uint64_t *)own((tape.count + 1) * sizeof(uint64_t));
            void *scratch = nullptr;
            SWH_HIP_CHECK(hipMalloc(&scratch, utf8_scratch_words(p->total_bytes) * sizeof(uint32_t) + 256));
            u.counts = (uint32_t *)scratch;
            u.invalid = words + 4;
            hipError_t decode_error = hipSuccess;
            try { launch_utf8_decode(scope, u); } catch (const HipFailure &f) { decode_error = f.code; }
            launch_tape_longest(scope, u.offsets, 1, tape.count, words + 1);
            uint32_t host_words[8] = {0};
            SWH_HIP_CHECK(hipMemcpyAsync(host_words, words, sizeof host_words, hipMemcpyDeviceToHost, stream));
            SWH_HIP_CHECK(hipMemcpyAsync(&total_symbols, u.offsets + tape.count, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            SWH_HIP_CHECK(hipStreamSynchronize(stream));
            (void)hipFree(scratch);
            if (decode_error != hipSuccess) throw HipFailure{decode_error, "UTF-8 decode"};
            if (host_words[4]) {
                free_prepared(p);
                snprintf(g_error_text, sizeof g_error_text, "invalid UTF-8 in the tape (marker %u: a 4-byte word inside the 1 KiB tile that failed)", host_words[4] - 1);
                if (error) *error = g_error_text;
                return swh_invalid_utf8_k;
            }
            p->longest_bytes = host_words[0];
            p->longest_symbols = host_words[1];
            p->symbols = TapeRef{u.symbols, u.offsets, tape.count};
            p->ascii = total_symbols == p->total_bytes;
        } else {
            uint32_t host_words[2] = {0, 0};
            SWH_HIP_CHECK(hipMemcpyAsync(host_words, words, sizeof host_words, hipMemcpyDeviceToHost, stream));
            SWH_HIP_CHECK(hipStreamSynchronize(stream));
            p->longest_bytes = p->longest_symbols = host_words[0];
            p->symbols = p->bytes;
        }
        p->total_symbols = total_symbols;
        *out = (swh_prepared_t)p;
        return swh_success_k;
    } catch (const HipFailure &f) {
        free_prepared(p);
        return fail_hip(error, f);
    } catch (const std::bad_alloc &) {
        free_prepared(p);
        return fail(error, swh_bad_alloc_k, "host allocation failed");
    }
}

// ---- engines --------------------------------------------------------------------------------------
static swh_status_t upload_matrix(Engine *engine, const int8_t *matrix, const char **error) {
    hipError_t err = hipMalloc((void **)&engine->matrix_dev, 65536);
    if (err == hipSuccess) err = hipMemcpy(engine->matrix_dev, matrix, 65536, hipMemcpyHostToDevice);
    if (err != hipSuccess) return fail_hip(error, HipFailure{err, "substitution matrix upload"});
    engine->scoring.matrix = engine->matrix_dev;
    return swh_success_k;
}

swh_status_t swh_levenshtein_init(swh_scope_t handle, int match, int mismatch, int open, int extend,
                                  swh_levenshtein_t *out, const char **error) {
    if (!handle || !out) return fail(error, swh_invalid_argument_k, "null argument");
    if (match < 0 || mismatch < 0 || open < 0 || extend < 0 || match > 127 || mismatch > 127 || open > 4096 ||
        extend > 4096)
        return fail(error, swh_invalid_argument_k, "Levenshtein costs must be small non-negative integers");
    Scope *scope = (Scope *)handle;
    Engine *engine = new Engine{};
    engine->kind = 0;
    engine->device = scope->device;
    engine->unit_costs = match == 0 && mismatch == 1 && open == 1 && extend == 1;
    engine->algorithm = swh_algorithm_auto_k;
    // max-plus core: distances are negated scores
    engine->scoring = Scoring{-match, -mismatch, -open, -extend, nullptr, nullptr};
    // affine gaps (open != extend) run on the wavefront kernels' uniform-cost Gotoh model, bytes and code points alike
    *out = (swh_levenshtein_t)engine;
    return swh_success_k;
}
swh_status_t swh_levenshtein_free(swh_levenshtein_t handle) {
    Engine *engine = (Engine *)handle;
    if (!engine) return swh_success_k;
    if (engine->kind != 0 && engine->uid) drop_engine_clones(engine->uid);   // the per-device clones multi-device scopes made of it (sharded.hip)
    if (engine->matrix_dev) { (void)hipSetDevice(engine->device); (void)hipFree(engine->matrix_dev); }
    if (engine->class_dev) { (void)hipSetDevice(engine->device); (void)hipFree(engine->class_dev); }
    delete[] engine->matrix_host;
    delete engine;
    return swh_success_k;
}
swh_status_t swh_levenshtein_set_algorithm(swh_levenshtein_t handle, swh_algorithm_t algorithm) {
    if (!handle) return swh_invalid_argument_k;
    ((Engine *)handle)->algorithm = algorithm;
    return swh_success_k;
}

static uint64_t next_engine_uid() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1);
}
static swh_status_t alignment_init(int kind, swh_scope_t handle, const int8_t *matrix, int open, int extend, void **out,
                                   const char **error) {
    if (!handle || !out || !matrix) return fail(error, swh_invalid_argument_k, "null argument");
    if (open > 0 || extend > 0 || open < -4096 || extend < -4096)
        return fail(error, swh_invalid_argument_k, "gap costs must be in [-4096, 0]");
    Scope *scope = (Scope *)handle;
    Engine *engine = new Engine{};
    engine->kind = kind;
    engine->device = scope->device;
    engine->algorithm = swh_algorithm_wavefront_k;
    engine->scoring = Scoring{0, 0, open, extend, nullptr, nullptr};
    bool symmetric = true;
    for (int i = 0; i < 256 && symmetric; ++i)
        for (int j = 0; j < i; ++j)
            if (matrix[i * 256 + j] != matrix[j * 256 + i]) { symmetric = false; break; }
    engine->unit_costs = symmetric;  // reused as "columns may be swapped" for nw engines
    engine->matrix_host = new int8_t[65536];
    memcpy(engine->matrix_host, matrix, 65536);
    engine->uid = next_engine_uid();
    (void)hipSetDevice(scope->device);
    swh_status_t st = upload_matrix(engine, matrix, error);
    if (st != swh_success_k) { swh_levenshtein_free((swh_levenshtein_t)engine); return st; }   // (releases matrix_host, too)
    // Symbol classes: bytes whose matrix rows AND columns coincide are interchangeable. With <= 32 classes (the
    // reference's own byte_to_class + 32x32 model, bench.rs:95-108; also 20 amino acids + "other") the kernels keep a
    // 32-byte cost row in registers per step instead of one LDS lookup per cell.
    {
        constexpr int kMost = (int)kWideClasses;
        static thread_local uint8_t table[kMost * kMost + 256];
        uint8_t map[256];
        int rep[kMost], classes = 0;
        bool fits = true;
        for (int b = 0; b < 256 && fits; ++b) {
            int found = -1;
            for (int c = 0; c < classes && found < 0; ++c) {
                const int r = rep[c];
                bool same = memcmp(matrix + b * 256, matrix + r * 256, 256) == 0;
                for (int k = 0; k < 256 && same; ++k) same = matrix[k * 256 + b] == matrix[k * 256 + r];
                if (same) found = c;
            }
            if (found < 0) {
                if (classes == kMost) { fits = false; break; }
                rep[classes] = b; found = classes++;
            }
            map[b] = (uint8_t)found;
        }
        // Global alignment runs on scores relative to the all-gaps baseline (wavefront.hip): the table holds
        // cost - extend - open (= cost - 2 g for linear gaps; the affine kernel keeps H + (open - extend) in its strips and
        // takes the surplus back on the diagonal), which must still fit a signed byte (otherwise the 256x256 LDS path is used).
        const int bias = kind == 2 ? -open : -(extend + open);   // local alignment: no baseline to be relative to, but its strips hold H + open
        for (int i = 0; i < classes && fits; ++i)
            for (int j = 0; j < classes; ++j) {
                int v = (int)matrix[rep[i] * 256 + rep[j]] + bias;
                if (v < -128 || v > 127) { fits = false; break; }
            }
        if (fits) {
            // up to 32 classes: the register cost-row model every class kernel reads (rows of 32 bytes); 33 .. 128: rows of kWideClasses
            // bytes for the column-profile kernel alone (Scoring::wide_table)
            const bool wide = classes > 32;
            const int stride = wide ? kMost : 32;
            const size_t bytes = (size_t)stride * stride + 256;
            memset(table, 0, bytes);
            for (int i = 0; i < classes; ++i)
                for (int j = 0; j < classes; ++j) table[i * stride + j] = (uint8_t)(int8_t)((int)matrix[rep[i] * 256 + rep[j]] + bias);
            memcpy(table + (size_t)stride * stride, map, 256);
            hipError_t err = hipMalloc((void **)&engine->class_dev, bytes);
            if (err == hipSuccess) err = hipMemcpy(engine->class_dev, table, bytes, hipMemcpyHostToDevice);
            if (err != hipSuccess) { swh_levenshtein_free((swh_levenshtein_t)engine); return fail_hip(error, HipFailure{err, "class table upload"}); }
            if (wide) engine->scoring.wide_table = engine->class_dev;
            else engine->scoring.class_table = engine->class_dev;
            engine->scoring.classes = (uint32_t)classes;
            int widest = 0;   // of the costs themselves (not the biased table)
            for (int i = 0; i < classes; ++i)
                for (int j = 0; j < classes; ++j) widest = std::max(widest, std::abs((int)matrix[rep[i] * 256 + rep[j]]));
            engine->scoring.step_span = (uint32_t)(widest - open - extend);
        }
    }
    *out = engine;
    return swh_success_k;
}
}  // extern "C"
namespace swh {
swh_status_t clone_alignment_engine(const Engine *source, swh_scope_t scope, void **out, const char **error) {
    if (!source || source->kind == 0 || !source->matrix_host) return fail(error, swh_invalid_argument_k, "not an alignment engine");
    return alignment_init(source->kind, scope, source->matrix_host, source->scoring.open, source->scoring.extend, out, error);
}
}  // namespace swh
extern "C" {
static swh_status_t alignment_init_classes(int kind, swh_scope_t handle, const uint8_t *byte_to_class,
                                           const int8_t *class_costs, int open, int extend, void **out, const char **error) {
    if (!byte_to_class || !class_costs) return fail(error, swh_invalid_argument_k, "null argument");
    static thread_local int8_t table[65536];
    for (int i = 0; i < 256; ++i) {
        if (byte_to_class[i] >= 32) return fail(error, swh_invalid_argument_k, "byte_to_class entries must be < 32");
        for (int j = 0; j < 256; ++j) {
            if (byte_to_class[j] >= 32) return fail(error, swh_invalid_argument_k, "byte_to_class entries must be < 32");
            table[i * 256 + j] = class_costs[byte_to_class[i] * 32 + byte_to_class[j]];
        }
    }
    return alignment_init(kind, handle, table, open, extend, out, error);
}
swh_status_t swh_nw_init(swh_scope_t handle, const int8_t *matrix, int open, int extend, swh_nw_t *out, const char **error) {
    return alignment_init(1, handle, matrix, open, extend, (void **)out, error);
}
swh_status_t swh_nw_init_classes(swh_scope_t handle, const uint8_t *byte_to_class, const int8_t *class_costs, int open,
                                 int extend, swh_nw_t *out, const char **error) {
    return alignment_init_classes(1, handle, byte_to_class, class_costs, open, extend, (void **)out, error);
}
swh_status_t swh_nw_free(swh_nw_t handle) { return swh_levenshtein_free((swh_levenshtein_t)handle); }
swh_status_t swh_sw_init(swh_scope_t handle, const int8_t *matrix, int open, int extend, swh_sw_t *out, const char **error) {
    return alignment_init(2, handle, matrix, open, extend, (void **)out, error);
}
swh_status_t swh_sw_init_classes(swh_scope_t handle, const uint8_t *byte_to_class, const int8_t *class_costs, int open,
                                 int extend, swh_sw_t *out, const char **error) {
    return alignment_init_classes(2, handle, byte_to_class, class_costs, open, extend, (void **)out, error);
}
swh_status_t swh_sw_free(swh_sw_t handle) { return swh_levenshtein_free((swh_levenshtein_t)handle); }

// ---- calls -----------------------------------------------------------------------------------------
#define SWH_TAPE(t, w) HostTape{(t)->data, (const void *)(t)->offsets, (t)->count, (w)}

static swh_status_t lev_pairs(swh_levenshtein_t e, swh_scope_t s, HostTape a, HostTape b, bool utf8, uint32_t bound,
                              uint32_t *out, size_t stride, const char **error) {
    if (e && ((Engine *)e)->kind != 0) return fail(error, swh_invalid_argument_k, "not a Levenshtein engine");
    CallSpec spec{a, b, false, utf8, bound, out, stride ? stride : 4, 0, false};
    if (spec.out_stride < 4) return fail(error, swh_invalid_argument_k, "out_stride_bytes must be >= 4");
    return run_call((Scope *)s, (Engine *)e, spec, error);
}

swh_status_t swh_levenshtein_pairs_u32tape(swh_levenshtein_t e, swh_scope_t s, const swh_tape_u32_t *a,
                                           const swh_tape_u32_t *b, uint32_t bound, uint32_t *out, size_t stride,
                                           const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return lev_pairs(e, s, SWH_TAPE(a, 0), SWH_TAPE(b, 0), false, bound, out, stride, error);
}
swh_status_t swh_levenshtein_pairs_u64tape(swh_levenshtein_t e, swh_scope_t s, const swh_tape_u64_t *a,
                                           const swh_tape_u64_t *b, uint32_t bound, uint32_t *out, size_t stride,
                                           const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return lev_pairs(e, s, SWH_TAPE(a, 1), SWH_TAPE(b, 1), false, bound, out, stride, error);
}
swh_status_t swh_levenshtein_utf8_pairs_u32tape(swh_levenshtein_t e, swh_scope_t s, const swh_tape_u32_t *a,
                                                const swh_tape_u32_t *b, uint32_t bound, uint32_t *out, size_t stride,
                                                const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return lev_pairs(e, s, SWH_TAPE(a, 0), SWH_TAPE(b, 0), true, bound, out, stride, error);
}
swh_status_t swh_levenshtein_utf8_pairs_u64tape(swh_levenshtein_t e, swh_scope_t s, const swh_tape_u64_t *a,
                                                const swh_tape_u64_t *b, uint32_t bound, uint32_t *out, size_t stride,
                                                const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return lev_pairs(e, s, SWH_TAPE(a, 1), SWH_TAPE(b, 1), true, bound, out, stride, error);
}

static swh_status_t cross_call(void *e, int kind, swh_scope_t s, const swh_tape_u64_t *a, const swh_tape_u64_t *b,
                               bool utf8, void *out, size_t row_stride, const char **error) {
    if (!a) return fail(error, swh_invalid_argument_k, "null tape");
    if (e && ((Engine *)e)->kind != kind) return fail(error, swh_invalid_argument_k, "engine kind mismatch");
    const swh_tape_u64_t *bb = b ? b : a;
    CallSpec spec{SWH_TAPE(a, 1), SWH_TAPE(bb, 1), true, utf8, SWH_UNBOUNDED, out, 8,
                  row_stride ? row_stride : bb->count * 8, true};
    if (spec.row_stride < bb->count * 8) return fail(error, swh_invalid_argument_k, "row_stride_bytes too small");
    return run_call((Scope *)s, (Engine *)e, spec, error);
}
swh_status_t swh_levenshtein_cross_u64tape(swh_levenshtein_t e, swh_scope_t s, const swh_tape_u64_t *a,
                                           const swh_tape_u64_t *b, size_t *out, size_t row_stride, const char **error) {
    return cross_call(e, 0, s, a, b, false, out, row_stride, error);
}
swh_status_t swh_levenshtein_utf8_cross_u64tape(swh_levenshtein_t e, swh_scope_t s, const swh_tape_u64_t *a,
                                                const swh_tape_u64_t *b, size_t *out, size_t row_stride,
                                                const char **error) {
    return cross_call(e, 0, s, a, b, true, out, row_stride, error);
}
swh_status_t swh_nw_cross_u64tape(swh_nw_t e, swh_scope_t s, const swh_tape_u64_t *a, const swh_tape_u64_t *b,
                                  ptrdiff_t *out, size_t row_stride, const char **error) {
    return cross_call(e, 1, s, a, b, false, out, row_stride, error);
}

static swh_status_t nw_pairs(void *e, int kind, swh_scope_t s, HostTape a, HostTape b, int32_t *out, size_t stride,
                             const char **error) {
    if (e && ((Engine *)e)->kind != kind) return fail(error, swh_invalid_argument_k, "engine kind mismatch");
    CallSpec spec{a, b, false, false, SWH_UNBOUNDED, out, stride ? stride : 4, 0, false};
    if (spec.out_stride < 4) return fail(error, swh_invalid_argument_k, "out_stride_bytes must be >= 4");
    return run_call((Scope *)s, (Engine *)e, spec, error);
}
swh_status_t swh_nw_pairs_u32tape(swh_nw_t e, swh_scope_t s, const swh_tape_u32_t *a, const swh_tape_u32_t *b,
                                  int32_t *out, size_t stride, const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return nw_pairs(e, 1, s, SWH_TAPE(a, 0), SWH_TAPE(b, 0), out, stride, error);
}
swh_status_t swh_nw_pairs_u64tape(swh_nw_t e, swh_scope_t s, const swh_tape_u64_t *a, const swh_tape_u64_t *b,
                                  int32_t *out, size_t stride, const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return nw_pairs(e, 1, s, SWH_TAPE(a, 1), SWH_TAPE(b, 1), out, stride, error);
}

swh_status_t swh_sw_pairs_u32tape(swh_sw_t e, swh_scope_t s, const swh_tape_u32_t *a, const swh_tape_u32_t *b,
                                  int32_t *out, size_t stride, const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return nw_pairs(e, 2, s, SWH_TAPE(a, 0), SWH_TAPE(b, 0), out, stride, error);
}
swh_status_t swh_sw_pairs_u64tape(swh_sw_t e, swh_scope_t s, const swh_tape_u64_t *a, const swh_tape_u64_t *b,
                                  int32_t *out, size_t stride, const char **error) {
    if (!a || !b) return fail(error, swh_invalid_argument_k, "null tape");
    return nw_pairs(e, 2, s, SWH_TAPE(a, 1), SWH_TAPE(b, 1), out, stride, error);
}
swh_status_t swh_sw_cross_u64tape(swh_sw_t e, swh_scope_t s, const swh_tape_u64_t *a, const swh_tape_u64_t *b,
                                  ptrdiff_t *out, size_t row_stride, const char **error) {
    return cross_call(e, 2, s, a, b, false, out, row_stride, error);
}

// ---- prepared tapes and the calls on them ------------------------------------------------------------------
swh_status_t swh_tape_prepare_u32(swh_scope_t scope, const swh_tape_u32_t *tape, int utf8, swh_prepared_t *prepared, const char **error) {
    if (!tape) return fail(error, swh_invalid_argument_k, "null tape");
    return prepare_tape((Scope *)scope, SWH_TAPE(tape, 0), utf8 != 0, prepared, error);
}
swh_status_t swh_tape_prepare_u64(swh_scope_t scope, const swh_tape_u64_t *tape, int utf8, swh_prepared_t *prepared, const char **error) {
    if (!tape) return fail(error, swh_invalid_argument_k, "null tape");
    return prepare_tape((Scope *)scope, SWH_TAPE(tape, 1), utf8 != 0, prepared, error);
}
swh_status_t swh_prepared_info(swh_prepared_t handle, swh_prepared_info_t *info) {
    const Prepared *p = (const Prepared *)handle;
    if (!p || !info) return swh_invalid_argument_k;
    info->count = (size_t)p->bytes.count;
    info->bytes = p->total_bytes;
    info->symbols = p->total_symbols;
    info->longest = p->utf8 ? p->longest_symbols : p->longest_bytes;
    info->utf8 = p->utf8 ? 1 : 0;
    info->ascii = p->ascii ? 1 : 0;
    return swh_success_k;
}
swh_status_t swh_prepared_free(swh_prepared_t handle) {
    free_prepared((Prepared *)handle);
    return swh_success_k;
}

static swh_status_t prepared_call(void *e, int kind, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b,
                                  bool cross, uint32_t bound, void *out, size_t stride, const char **error) {
    if (!a || !a->tape) return fail(error, swh_invalid_argument_k, "null prepared view");
    if (e && ((Engine *)e)->kind != kind) return fail(error, swh_invalid_argument_k, "engine kind mismatch");
    const swh_prepared_view_t *bb = (b && b->tape) ? b : (cross ? a : nullptr);
    if (!bb) return fail(error, swh_invalid_argument_k, "null prepared view");
    const Prepared *pa = (const Prepared *)a->tape, *pb = (const Prepared *)bb->tape;
    if (a->first > pa->bytes.count || a->count > pa->bytes.count - a->first || bb->first > pb->bytes.count ||
        bb->count > pb->bytes.count - bb->first)
        return fail(error, swh_invalid_argument_k, "view exceeds the prepared tape");
    CallSpec spec{};
    spec.a = HostTape{nullptr, nullptr, a->count, 0};
    spec.b = HostTape{nullptr, nullptr, bb->count, 0};
    spec.cross = cross; spec.utf8 = pa->utf8; spec.bound = bound; spec.out = out;
    if (cross) {
        spec.out_stride = 8; spec.out64 = true;
        spec.row_stride = stride ? stride : bb->count * 8;
        if (spec.row_stride < bb->count * 8) return fail(error, swh_invalid_argument_k, "row_stride_bytes too small");
    } else {
        spec.out_stride = stride ? stride : 4; spec.out64 = false;
        if (spec.out_stride < 4) return fail(error, swh_invalid_argument_k, "out_stride_bytes must be >= 4");
    }
    spec.pa = pa; spec.pb = pb; spec.a_first = a->first; spec.b_first = bb->first;
    return run_call((Scope *)s, (Engine *)e, spec, error);
}
swh_status_t swh_levenshtein_pairs_prepared(swh_levenshtein_t e, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b,
                                            uint32_t bound, uint32_t *out, size_t stride, const char **error) {
    return prepared_call(e, 0, s, a, b, false, bound, out, stride, error);
}
swh_status_t swh_levenshtein_cross_prepared(swh_levenshtein_t e, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b,
                                            size_t *out, size_t row_stride, const char **error) {
    return prepared_call(e, 0, s, a, b, true, SWH_UNBOUNDED, out, row_stride, error);
}
swh_status_t swh_nw_pairs_prepared(swh_nw_t e, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b, int32_t *out,
                                   size_t stride, const char **error) {
    return prepared_call(e, 1, s, a, b, false, SWH_UNBOUNDED, out, stride, error);
}
swh_status_t swh_nw_cross_prepared(swh_nw_t e, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b, ptrdiff_t *out,
                                   size_t row_stride, const char **error) {
    return prepared_call(e, 1, s, a, b, true, SWH_UNBOUNDED, out, row_stride, error);
}
swh_status_t swh_sw_pairs_prepared(swh_sw_t e, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b, int32_t *out,
                                   size_t stride, const char **error) {
    return prepared_call(e, 2, s, a, b, false, SWH_UNBOUNDED, out, stride, error);
}
swh_status_t swh_sw_cross_prepared(swh_sw_t e, swh_scope_t s, const swh_prepared_view_t *a, const swh_prepared_view_t *b, ptrdiff_t *out,
                                   size_t row_stride, const char **error) {
    return prepared_call(e, 2, s, a, b, true, SWH_UNBOUNDED, out, row_stride, error);
}

}  // extern "C"
