// stringwars_amd.hpp -- C++17 host-side mirror of the reference's Rust interfaces for the similarity path,
// header-only, over the C ABI of stringwars_amd.h. Two halves:
//
//   namespace swa            RAII engines with the names and call shapes of `stringzilla::szs` as used by
//                            similarities/bench.rs:79-82 (DeviceScope, LevenshteinDistances,
//                            LevenshteinDistancesUtf8, NeedlemanWunschScores, compute_into) plus `pairs_into`.
//   namespace swa::harness   the shared harness of utils.rs: env helpers (:13-50), should_run (:457-483),
//                            BenchBudget (:566-582), WorkUnits (:529-545), measure_throughput (:721-799),
//                            BenchStats::report (:652-692), dataset loader (:273-433).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <regex>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include <linux/perf_event.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "stringwars_amd.h"
#include "stringwars_amd_harness.h"

namespace swa {

/// `E: Display` of the szs wrappers (bench.rs:480-485 `panic!("{}", error)`, :632-635 SKIPPED).
struct Error : std::runtime_error {
    swh_status_t status;
    Error(swh_status_t s, const char *msg) : std::runtime_error(msg ? msg : ""), status(s) {}
};
inline void check(swh_status_t status, const char *message) {
    if (status != swh_success_k) throw Error(status, message);
}

/// Borrowed Arrow-style tape view, `BytesTapeView<u64>` (bench.rs:298-301); `subview` is O(1) (bench.rs:134-139).
struct BytesTapeView {
    const uint8_t *data = nullptr;
    const uint64_t *offsets = nullptr;
    size_t count = 0;
    size_t size() const { return count; }
    size_t length(size_t i) const { return (size_t)(offsets[i + 1] - offsets[i]); }
    BytesTapeView subview(size_t lo, size_t hi) const { return BytesTapeView{data, offsets + lo, hi - lo}; }
    swh_tape_u64_t c() const { return swh_tape_u64_t{data, offsets, count}; }
};

/// Owning tape, `BytesTape<u64, UnifiedAlloc>` (bench.rs:292-295): host vectors here; pass `view()` to engines.
struct BytesTape {
    std::vector<uint8_t> data;
    std::vector<uint64_t> offsets{0};
    void push(const uint8_t *bytes, size_t n) { data.insert(data.end(), bytes, bytes + n); offsets.push_back(data.size()); }
    size_t size() const { return offsets.size() - 1; }
    BytesTapeView view() const { return BytesTapeView{data.data(), offsets.data(), size()}; }
};

class DeviceScope {
    swh_scope_t handle_ = nullptr;
public:
    /// `DeviceScope::gpu_device(0)` (bench.rs:379). Throws swa::Error when no gfx950 device exists.
    static DeviceScope gpu_device(int index) {
        DeviceScope s; const char *err = nullptr;
        swh_status_t status__ = swh_scope_init_gpu(index, &s.handle_, &err);
        check(status__, err);
        return s;
    }
    /// `DeviceScope::cpu_cores(n)` (bench.rs:376-378): always throws `not_implemented` -- no CPU backend.
    static DeviceScope cpu_cores(size_t cores) {
        DeviceScope s; const char *err = nullptr;
        swh_status_t status__ = swh_scope_init_cpu(cores, &s.handle_, &err);
        check(status__, err);
        return s;
    }
    /// Every listed GPU behind one scope (`swh_scope_init_gpus`): the `<Ngpu>` rows; batches are split by `ShardedPairs`.
    static DeviceScope gpu_devices(const std::vector<int> &devices) {
        DeviceScope s; const char *err = nullptr;
        swh_status_t status__ = swh_scope_init_gpus(devices.data(), (int)devices.size(), &s.handle_, &err);
        check(status__, err);
        return s;
    }
    static int visible_devices() { int n = 0; swh_device_count(&n); return n; }
    size_t device_count() const { size_t n = 1; swh_scope_device_count(handle_, &n); return n; }
    swh_shard_timing_t shard_timing() const { swh_shard_timing_t t{}; swh_scope_shard_timing(handle_, &t); return t; }
    DeviceScope() = default;
    DeviceScope(DeviceScope &&o) noexcept : handle_(o.handle_) { o.handle_ = nullptr; }
    DeviceScope &operator=(DeviceScope &&o) noexcept { std::swap(handle_, o.handle_); return *this; }
    DeviceScope(const DeviceScope &) = delete;
    ~DeviceScope() { if (handle_) swh_scope_free(handle_); }
    swh_scope_t handle() const { return handle_; }
    /// `gpu_multiprocessor_count(0)` (utils.rs:826-836): compute units, one "core" each for batch sizing.
    size_t compute_units() const { size_t n = 0; swh_scope_compute_units(handle_, &n); return n; }
    void set_profiling(bool on) const { swh_scope_set_profiling(handle_, on); }
    swh_timing_t last_timing() const { swh_timing_t t{}; swh_scope_last_timing(handle_, &t); return t; }
    swh_timing_totals_t timing_totals() const { swh_timing_totals_t t{}; swh_scope_timing_totals(handle_, &t); return t; }
};

/// A tape made ready once (`swh_tape_prepare_u64`): resident, measured, UTF-8 validated + decoded -- what the reference does
/// when it builds `BytesTapeView` / `CharsTapeView` once outside its timed closures (bench.rs:292-306). `subview` is O(1).
class PreparedTape {
    swh_prepared_t handle_ = nullptr;
    size_t first_ = 0, count_ = 0;
    bool owner_ = false;
    PreparedTape(swh_prepared_t h, size_t first, size_t count) : handle_(h), first_(first), count_(count) {}
public:
    PreparedTape(const DeviceScope &scope, const BytesTapeView &tape, bool utf8 = false) : count_(tape.count), owner_(true) {
        const char *err = nullptr;
        swh_tape_u64_t t = tape.c();
        swh_status_t status__ = swh_tape_prepare_u64(scope.handle(), &t, utf8 ? 1 : 0, &handle_, &err);
        check(status__, err);   // invalid UTF-8 surfaces here, as `try_into` does (bench.rs:303-306)
    }
    PreparedTape(PreparedTape &&o) noexcept : handle_(o.handle_), first_(o.first_), count_(o.count_), owner_(o.owner_) { o.owner_ = false; }
    PreparedTape(const PreparedTape &) = delete;
    ~PreparedTape() { if (owner_ && handle_) swh_prepared_free(handle_); }
    PreparedTape subview(size_t lo, size_t hi) const { return PreparedTape(handle_, first_ + lo, hi - lo); }
    size_t size() const { return count_; }
    swh_prepared_info_t info() const { swh_prepared_info_t i{}; swh_prepared_info(handle_, &i); return i; }
    swh_prepared_view_t c() const { return swh_prepared_view_t{handle_, first_, count_}; }
};

/// One pairwise batch resident on every GPU of a multi-device scope (`swh_sharded_prepare_u64tape`), cells-balanced shards.
class ShardedPairs {
    swh_sharded_t handle_ = nullptr;
    size_t count_ = 0;
public:
    ShardedPairs(const DeviceScope &scope, const BytesTapeView &a, const BytesTapeView &b, bool utf8 = false) : count_(a.count) {
        const char *err = nullptr;
        swh_tape_u64_t ta = a.c(), tb = b.c();
        swh_status_t status__ = swh_sharded_prepare_u64tape(scope.handle(), &ta, &tb, utf8 ? 1 : 0, &handle_, &err);
        check(status__, err);
    }
    ShardedPairs(const ShardedPairs &) = delete;
    ~ShardedPairs() { if (handle_) swh_sharded_free(handle_); }
    swh_sharded_t handle() const { return handle_; }
    size_t size() const { return count_; }
};

/// A dense queries x candidates product resident on every GPU of a multi-device scope (`swh_sharded_cross_prepare_u64tape`).
class ShardedCross {
    swh_sharded_cross_t handle_ = nullptr;
    size_t rows_ = 0, columns_ = 0;
public:
    ShardedCross(const DeviceScope &scope, const BytesTapeView &queries, const BytesTapeView &candidates, bool utf8 = false)
        : rows_(queries.count), columns_(candidates.count) {
        const char *err = nullptr;
        swh_tape_u64_t tq = queries.c(), tc = candidates.c();
        swh_status_t status__ = swh_sharded_cross_prepare_u64tape(scope.handle(), &tq, &tc, utf8 ? 1 : 0, &handle_, &err);
        check(status__, err);
    }
    ShardedCross(const ShardedCross &) = delete;
    ~ShardedCross() { if (handle_) swh_sharded_cross_free(handle_); }
    swh_sharded_cross_t handle() const { return handle_; }
    size_t rows() const { return rows_; }
    size_t columns() const { return columns_; }
};

class LevenshteinDistances {
protected:
    swh_levenshtein_t handle_ = nullptr;
    bool utf8_ = false;
public:
    /// `LevenshteinDistances::new(&scope, 0, 1, 1, 1)` (bench.rs:382).
    LevenshteinDistances(const DeviceScope &scope, int match = 0, int mismatch = 1, int open = 1, int extend = 1) {
        const char *err = nullptr;
        swh_status_t status__ = swh_levenshtein_init(scope.handle(), match, mismatch, open, extend, &handle_, &err);
        check(status__, err);
    }
    LevenshteinDistances(const LevenshteinDistances &) = delete;
    ~LevenshteinDistances() { if (handle_) swh_levenshtein_free(handle_); }
    void set_algorithm(swh_algorithm_t algorithm) { swh_levenshtein_set_algorithm(handle_, algorithm); }
    /// `engine.compute_into(&scope, View64(q), Some(View64(c)), &mut matrix)` (bench.rs:478-486): dense
    /// q.count x c.count row-major matrix of `size_t`; `candidates == nullptr` is the symmetric self-product.
    void compute_into(const DeviceScope &scope, const BytesTapeView &queries, const BytesTapeView *candidates,
                      size_t *matrix, size_t row_stride_bytes = 0) const {
        const char *err = nullptr;
        swh_tape_u64_t q = queries.c(), c = candidates ? candidates->c() : q;
        auto fn = utf8_ ? swh_levenshtein_utf8_cross_u64tape : swh_levenshtein_cross_u64tape;
        swh_status_t status__ = fn(handle_, scope.handle(), &q, candidates ? &c : nullptr, matrix, row_stride_bytes, &err);
        check(status__, err);
    }
    /// Pairwise batch: out[i] = min(d(a_i, b_i), bound + 1).
    void pairs_into(const DeviceScope &scope, const BytesTapeView &a, const BytesTapeView &b, uint32_t *out,
                    uint32_t bound = SWH_UNBOUNDED) const {
        const char *err = nullptr;
        swh_tape_u64_t ta = a.c(), tb = b.c();
        auto fn = utf8_ ? swh_levenshtein_utf8_pairs_u64tape : swh_levenshtein_pairs_u64tape;
        swh_status_t status__ = fn(handle_, scope.handle(), &ta, &tb, bound, out, 4, &err);
        check(status__, err);
    }
    /// The same on prepared tapes: no per-call decode, no planning pre-pass for word- and token-sized strings.
    void pairs_into(const DeviceScope &scope, const PreparedTape &a, const PreparedTape &b, uint32_t *out,
                    uint32_t bound = SWH_UNBOUNDED) const {
        const char *err = nullptr;
        swh_prepared_view_t va = a.c(), vb = b.c();
        swh_status_t status__ = swh_levenshtein_pairs_prepared(handle_, scope.handle(), &va, &vb, bound, out, 4, &err);
        check(status__, err);
    }
    void compute_into(const DeviceScope &scope, const PreparedTape &queries, const PreparedTape *candidates, size_t *matrix,
                      size_t row_stride_bytes = 0) const {
        const char *err = nullptr;
        swh_prepared_view_t q = queries.c(), c = candidates ? candidates->c() : q;
        swh_status_t status__ = swh_levenshtein_cross_prepared(handle_, scope.handle(), &q, candidates ? &c : nullptr, matrix, row_stride_bytes, &err);
        check(status__, err);
    }
    /// `compute_into` over every GPU of a multi-device scope: each device fills its rows of the matrix.
    void compute_into(const DeviceScope &scope, const ShardedCross &product, size_t *matrix, size_t row_stride_bytes = 0) const {
        const char *err = nullptr;
        swh_status_t status__ = swh_levenshtein_cross_sharded(handle_, scope.handle(), product.handle(), matrix, row_stride_bytes, &err);
        check(status__, err);
    }
    /// One batch over every GPU of a multi-device scope, distances gathered with RCCL inside the library.
    void pairs_into(const DeviceScope &scope, const ShardedPairs &batch, uint32_t *out, uint32_t bound = SWH_UNBOUNDED) const {
        const char *err = nullptr;
        swh_status_t status__ = swh_levenshtein_pairs_sharded(handle_, scope.handle(), batch.handle(), bound, out, &err);
        check(status__, err);
    }
};

/// `LevenshteinDistancesUtf8` (bench.rs:386-399): symbols are Unicode scalar values.
class LevenshteinDistancesUtf8 : public LevenshteinDistances {
public:
    LevenshteinDistancesUtf8(const DeviceScope &scope, int match = 0, int mismatch = 1, int open = 1, int extend = 1)
        : LevenshteinDistances(scope, match, mismatch, open, extend) { utf8_ = true; }
};

class NeedlemanWunschScores {
    swh_nw_t handle_ = nullptr;
public:
    /// `NeedlemanWunschScores::new(&scope, &byte_to_class, &class_costs, open, extend)` (bench.rs:658-662).
    NeedlemanWunschScores(const DeviceScope &scope, const uint8_t (&byte_to_class)[256], const int8_t (&class_costs)[32][32],
                          int open, int extend) {
        const char *err = nullptr;
        swh_status_t status__ = swh_nw_init_classes(scope.handle(), byte_to_class, &class_costs[0][0], open, extend, &handle_, &err);
        check(status__, err);
    }
    /// Full 256x256 substitution matrix (config C4).
    NeedlemanWunschScores(const DeviceScope &scope, const int8_t *matrix_256x256, int open, int extend) {
        const char *err = nullptr;
        swh_status_t status__ = swh_nw_init(scope.handle(), matrix_256x256, open, extend, &handle_, &err);
        check(status__, err);
    }
    NeedlemanWunschScores(const NeedlemanWunschScores &) = delete;
    ~NeedlemanWunschScores() { if (handle_) swh_nw_free(handle_); }
    /// `compute_into(..) -> UnifiedMat<isize>` (bench.rs:814-821).
    void compute_into(const DeviceScope &scope, const BytesTapeView &queries, const BytesTapeView *candidates,
                      ptrdiff_t *matrix, size_t row_stride_bytes = 0) const {
        const char *err = nullptr;
        swh_tape_u64_t q = queries.c(), c = candidates ? candidates->c() : q;
        swh_status_t status__ = swh_nw_cross_u64tape(handle_, scope.handle(), &q, candidates ? &c : nullptr, matrix, row_stride_bytes, &err);
        check(status__, err);
    }
    void pairs_into(const DeviceScope &scope, const BytesTapeView &a, const BytesTapeView &b, int32_t *out) const {
        const char *err = nullptr;
        swh_tape_u64_t ta = a.c(), tb = b.c();
        swh_status_t status__ = swh_nw_pairs_u64tape(handle_, scope.handle(), &ta, &tb, out, 4, &err);
        check(status__, err);
    }
    /// One batch over every GPU of a multi-device scope (the matrix is cloned to each device on first use).
    void pairs_into(const DeviceScope &scope, const ShardedPairs &batch, int32_t *out) const {
        const char *err = nullptr;
        swh_status_t status__ = swh_nw_pairs_sharded(handle_, scope.handle(), batch.handle(), out, &err);
        check(status__, err);
    }
};

/// `SmithWatermanScores` (bench.rs:81, :882-963): local alignment scores, same construction as NeedlemanWunschScores.
class SmithWatermanScores {
    swh_sw_t handle_ = nullptr;
public:
    SmithWatermanScores(const DeviceScope &scope, const uint8_t (&byte_to_class)[256], const int8_t (&class_costs)[32][32],
                        int open, int extend) {
        const char *err = nullptr;
        swh_status_t status__ = swh_sw_init_classes(scope.handle(), byte_to_class, &class_costs[0][0], open, extend, &handle_, &err);
        check(status__, err);
    }
    SmithWatermanScores(const DeviceScope &scope, const int8_t *matrix_256x256, int open, int extend) {
        const char *err = nullptr;
        swh_status_t status__ = swh_sw_init(scope.handle(), matrix_256x256, open, extend, &handle_, &err);
        check(status__, err);
    }
    SmithWatermanScores(const SmithWatermanScores &) = delete;
    ~SmithWatermanScores() { if (handle_) swh_sw_free(handle_); }
    void compute_into(const DeviceScope &scope, const BytesTapeView &queries, const BytesTapeView *candidates,
                      ptrdiff_t *matrix, size_t row_stride_bytes = 0) const {
        const char *err = nullptr;
        swh_tape_u64_t q = queries.c(), c = candidates ? candidates->c() : q;
        swh_status_t status__ = swh_sw_cross_u64tape(handle_, scope.handle(), &q, candidates ? &c : nullptr, matrix, row_stride_bytes, &err);
        check(status__, err);
    }
    void pairs_into(const DeviceScope &scope, const BytesTapeView &a, const BytesTapeView &b, int32_t *out) const {
        const char *err = nullptr;
        swh_tape_u64_t ta = a.c(), tb = b.c();
        swh_status_t status__ = swh_sw_pairs_u64tape(handle_, scope.handle(), &ta, &tb, out, 4, &err);
        check(status__, err);
    }
};

namespace harness {

// ---- env helpers (utils.rs:13-50) --------------------------------------------------------------------
inline bool get_env(const char *name, std::string &out) {
    const char *v = std::getenv(name);
    if (!v) return false;
    out = v;
    return true;
}
inline double get_env_parsed(const char *name, double fallback) {
    std::string v;
    if (!get_env(name, v)) return fallback;
    char *end = nullptr;
    double x = std::strtod(v.c_str(), &end);
    return end && *end == 0 && end != v.c_str() ? x : fallback;
}

// ---- should_run (utils.rs:457-483) ----------------------------------------------------------------------
inline bool should_run(const std::string &name) {
    std::string filter;
    if (!get_env("STRINGWARS_FILTER", filter)) return true;
    static bool announced = false;
    if (!announced) { std::fprintf(stderr, "STRINGWARS_FILTER active: '%s'\n", filter.c_str()); announced = true; }
    try {
        std::regex re(filter);
        bool matches = std::regex_search(name, re);
        if (!matches) std::fprintf(stderr, "  Skipping: %s\n", name.c_str());
        return matches;
    } catch (const std::regex_error &) {
        std::fprintf(stderr, "Warning: Invalid regex pattern '%s', falling back to substring match\n", filter.c_str());
        return name.find(filter) != std::string::npos;
    }
}

// ---- WorkUnits / ReportAs / BenchBudget (utils.rs:529-582) ---------------------------------------------
struct WorkUnits { uint64_t elements = 0, bytes = 0; };
enum class ReportAs { Bytes, Cups, Hashes, Bits, Comparisons };
struct BenchBudget {
    double warm_up_seconds, measure_seconds;
    /// `BenchBudget::from_env(5.0, 30.0)` (bench.rs:1031): STRINGWARS_WARMUP / STRINGWARS_TIME.
    static BenchBudget from_env(double default_warm_up, double default_measure) {
        return BenchBudget{std::max(0.0, get_env_parsed("STRINGWARS_WARMUP", default_warm_up)),
                           std::max(0.0, get_env_parsed("STRINGWARS_TIME", default_measure))};
    }
};

/// Cycle and instruction counters of the calling thread around the measured region (utils.rs:589-620): two
/// perf_event descriptors, user space only; whichever cannot be opened (perf_event_paranoid, containers) is
/// simply absent and its columns are left out, as in the reference. For a GPU engine they describe the HOST side
/// of the call -- the submit, the wait -- which is what the reference's counters see of its `<1gpu>` rows too.
class HardwareCounters {
    int cycles_ = -1, instructions_ = -1;
    static int open_counter(uint64_t config) {
        perf_event_attr attr;
        std::memset(&attr, 0, sizeof attr);
        attr.type = PERF_TYPE_HARDWARE;
        attr.size = sizeof attr;
        attr.config = config;
        attr.disabled = 1;
        attr.exclude_kernel = 1;
        attr.exclude_hv = 1;
        int fd = (int)syscall(SYS_perf_event_open, &attr, 0, -1, -1, 0);
        if (fd < 0) return -1;
        if (ioctl(fd, PERF_EVENT_IOC_RESET, 0) < 0 || ioctl(fd, PERF_EVENT_IOC_ENABLE, 0) < 0) { close(fd); return -1; }
        return fd;
    }
    static bool stop_counter(int &fd, uint64_t &value) {
        if (fd < 0) return false;
        bool ok = ioctl(fd, PERF_EVENT_IOC_DISABLE, 0) == 0 && read(fd, &value, sizeof value) == (ssize_t)sizeof value;
        close(fd);
        fd = -1;
        return ok;
    }

  public:
    HardwareCounters() : cycles_(open_counter(PERF_COUNT_HW_CPU_CYCLES)), instructions_(open_counter(PERF_COUNT_HW_INSTRUCTIONS)) {}
    HardwareCounters(const HardwareCounters &) = delete;
    HardwareCounters &operator=(const HardwareCounters &) = delete;
    ~HardwareCounters() {
        if (cycles_ >= 0) close(cycles_);
        if (instructions_ >= 0) close(instructions_);
    }
    void stop(bool &has_cycles, uint64_t &cycles, bool &has_instructions, uint64_t &instructions) {
        has_cycles = stop_counter(cycles_, cycles);
        has_instructions = stop_counter(instructions_, instructions);
    }
};

struct BenchStats {
    double elapsed_seconds = 0;
    uint64_t calls = 0, elements = 0, bytes = 0;
    bool has_cycles = false, has_instructions = false;
    uint64_t cycles = 0, instructions = 0;
    std::vector<double> latencies_ns;
    /// index round(q * (n-1)) of the sorted samples (utils.rs:639-647)
    bool latency_quantile(double q, double &out) const {
        if (latencies_ns.empty()) return false;
        std::vector<double> sorted = latencies_ns;
        std::sort(sorted.begin(), sorted.end());
        size_t rank = (size_t)std::llround(q * ((double)sorted.size() - 1.0));
        out = sorted[std::min(rank, sorted.size() - 1)];
        return true;
    }
    /// The canonical line `{:<42} {cols.join(" | ")}` (utils.rs:652-692); `cyc/B` and `IPC` appear when the
    /// perf_event counters could be read (utils.rs:672-680) and are left out otherwise.
    std::string line(const std::string &name, ReportAs report) const {
        double seconds = std::max(elapsed_seconds, 1e-12);
        std::vector<std::string> columns;
        char buffer[96];
        double eps = (double)elements / seconds, bps = (double)bytes / seconds;
        switch (report) {
        case ReportAs::Bytes: swh_format_si_rate(bps, "B/s", 0, buffer, sizeof buffer); break;
        case ReportAs::Cups: swh_format_si_rate(eps, "CUPS", 0, buffer, sizeof buffer); break;
        case ReportAs::Hashes: swh_format_si_rate(eps, "hashes/s", 1, buffer, sizeof buffer); break;
        case ReportAs::Bits: swh_format_si_rate(eps, "bits/s", 1, buffer, sizeof buffer); break;
        case ReportAs::Comparisons: swh_format_si_rate(eps, "cmp/s", 1, buffer, sizeof buffer); break;
        }
        columns.push_back(buffer);
        if (report != ReportAs::Bytes && bytes > 0) { swh_format_si_rate(bps, "B/s", 0, buffer, sizeof buffer); columns.push_back(buffer); }
        if (has_cycles && bytes > 0) {
            std::snprintf(buffer, sizeof buffer, "%.2f cyc/B", (double)cycles / (double)bytes);
            columns.push_back(buffer);
        }
        if (has_cycles && has_instructions && cycles > 0) {
            std::snprintf(buffer, sizeof buffer, "IPC %.2f", (double)instructions / (double)cycles);
            columns.push_back(buffer);
        }
        double p50, p99;
        if (latency_quantile(0.5, p50) && latency_quantile(0.99, p99)) {
            char a[48], b[48];
            swh_format_seconds(p50 / 1e9, a, sizeof a); swh_format_seconds(p99 / 1e9, b, sizeof b);
            columns.push_back(std::string("p50 ") + a + " p99 " + b);
        }
        std::string joined;
        for (size_t i = 0; i < columns.size(); ++i) joined += (i ? " | " : "") + columns[i];
        char head[64];
        std::snprintf(head, sizeof head, "%-42s ", name.c_str());
        return std::string(name.size() > 42 ? name + " " : head) + joined;
    }
};

/// Time-budgeted loop (utils.rs:721-799): uncounted warm-up, then cycle `routine` until the deadline with a
/// stride that doubles (cap 1024) while a block of calls takes < 1 ms; at least one call always happens.
inline BenchStats measure_throughput(const std::string &name, ReportAs report, const BenchBudget &budget,
                                     const std::function<WorkUnits()> &routine) {
    using clock = std::chrono::steady_clock;
    BenchStats stats;
    if (!should_run(name)) return stats;
    if (budget.warm_up_seconds > 0) {
        auto start = clock::now();
        while (std::chrono::duration<double>(clock::now() - start).count() < budget.warm_up_seconds) (void)routine();
    }
    const uint64_t kStrideCap = 1024;
    const double kTargetBetweenChecks = 1e-3;
    HardwareCounters counters;
    auto start = clock::now();
    auto deadline = start + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(budget.measure_seconds));
    uint64_t stride = 1, countdown = 1, calls_since_check = 0;
    auto last_check = start;
    for (;;) {
        WorkUnits work = routine();
        stats.elements += work.elements; stats.bytes += work.bytes;
        ++stats.calls; ++calls_since_check;
        if (--countdown != 0) continue;
        auto now = clock::now();
        double block = std::chrono::duration<double>(now - last_check).count();
        if (calls_since_check > 0) stats.latencies_ns.push_back(block * 1e9 / (double)calls_since_check);
        if (now >= deadline) break;
        if (block < kTargetBetweenChecks && stride < kStrideCap) stride = std::min(stride * 2, kStrideCap);
        last_check = now; calls_since_check = 0; countdown = stride;
    }
    stats.elapsed_seconds = std::chrono::duration<double>(clock::now() - start).count();
    counters.stop(stats.has_cycles, stats.cycles, stats.has_instructions, stats.instructions);
    std::printf("%s\n", stats.line(name, report).c_str());
    std::fflush(stdout);
    return stats;
}

// ---- dataset loader (utils.rs:273-433) ---------------------------------------------------------------------
inline std::string format_number(uint64_t n) {
    std::string digits = std::to_string(n), out;
    for (size_t i = 0; i < digits.size(); ++i) {
        if (i && (digits.size() - i) % 3 == 0) out += ',';
        out += digits[i];
    }
    return out;
}

/// `load_dataset_with_default_mode("words")` (bench.rs:271): STRINGWARS_DATASET, STRINGWARS_TOKENS
/// (lines|words|file), STRINGWARS_MAX_TOKENS, STRINGWARS_UNIQUE; words split on ' ' and '\n' only, empties
/// dropped (utils.rs:327-331); statistics + log2 histogram to stderr (utils.rs:367-430).
inline BytesTape load_dataset_with_default_mode(const char *default_mode) {
    std::string path, mode = default_mode, tmp;
    if (!get_env("STRINGWARS_DATASET", path)) throw std::runtime_error("STRINGWARS_DATASET environment variable is not set");
    if (get_env("STRINGWARS_TOKENS", tmp)) mode = tmp;
    size_t limit = (size_t)-1;
    if (get_env("STRINGWARS_MAX_TOKENS", tmp)) {
        limit = (size_t)std::strtoull(tmp.c_str(), nullptr, 10);
        std::fprintf(stderr, "STRINGWARS_MAX_TOKENS: limiting to %zu tokens\n", limit);
    }
    bool unique = get_env("STRINGWARS_UNIQUE", tmp) && (tmp == "1" || tmp == "true" || tmp == "TRUE" || tmp == "yes");
    if (unique) std::fprintf(stderr, "STRINGWARS_UNIQUE: deduplicating tokens\n");
    std::ifstream file(path, std::ios::binary);
    if (!file) throw std::runtime_error("Dataset file not found: " + path);
    std::string content((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    if (content.empty()) throw std::runtime_error("Dataset file is empty: " + path);
    BytesTape tape;
    std::unordered_set<std::string> seen;
    auto emit = [&](const char *p, size_t n) {
        if (n == 0 || tape.size() >= limit) return;
        if (unique && !seen.insert(std::string(p, n)).second) return;
        tape.push((const uint8_t *)p, n);
    };
    if (mode == "file") emit(content.data(), content.size());
    else if (mode == "lines" || mode == "words") {
        size_t start = 0;
        for (size_t i = 0; i <= content.size(); ++i) {
            bool sep = i == content.size() || content[i] == '\n' || (mode == "words" && content[i] == ' ');
            if (sep) { emit(content.data() + start, i - start); start = i + 1; }
        }
    } else throw std::runtime_error("Unknown STRINGWARS_TOKENS mode: " + mode);
    if (tape.size() == 0) throw std::runtime_error("No tokens extracted from " + path + " in mode " + mode);
    size_t count = tape.size(), total = tape.data.size(), min_len = (size_t)-1, max_len = 0;
    double mean = (double)total / (double)count, variance = 0;
    uint64_t buckets[18] = {0};
    for (size_t i = 0; i < count; ++i) {
        size_t len = (size_t)(tape.offsets[i + 1] - tape.offsets[i]);
        min_len = std::min(min_len, len); max_len = std::max(max_len, len);
        variance += ((double)len - mean) * ((double)len - mean);
        size_t bucket = len == 0 ? 0 : len == 1 ? 1 : std::min<size_t>((size_t)std::floor(std::log2((double)len)) + 1, 17);
        ++buckets[bucket];
    }
    std::fprintf(stderr, "Dataset: %s tokens, %s bytes (%.2f GB)\n  Length: min %zu, max %zu, mean %.1f, std %.1f\n",
                 format_number(count).c_str(), format_number(total).c_str(), (double)total / 1e9, min_len, max_len, mean,
                 std::sqrt(variance / (double)count));
    static const char *ranges[18] = {"0", "1", "2-3", "4-7", "8-15", "16-31", "32-63", "64-127", "128-255", "256-511",
                                     "512-1K", "1K-2K", "2K-4K", "4K-8K", "8K-16K", "16K-32K", "32K-64K", "64K+"};
    std::fprintf(stderr, "  Distribution:\n");
    for (int i = 0; i < 18; ++i)
        if (buckets[i]) std::fprintf(stderr, "    %10s bytes: %6.2f%%\n", ranges[i], 100.0 * (double)buckets[i] / (double)count);
    return tape;
}

}  // namespace harness
}  // namespace swa
