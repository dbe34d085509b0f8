// bench_similarities.cpp -- C++ counterpart of the reference's `similarities/bench.rs` for the rows this
// backend provides, on the harness restated in include/stringwars_amd.hpp. Same environment variables
// (STRINGWARS_DATASET/TOKENS/MAX_TOKENS/UNIQUE/BATCH_PER_CORE/TIME/WARMUP/FILTER), same group headers
// (`# uniform`, `# linear`, `# affine`; bench.rs:331/:343/:354), same `group/library::Engine<device>` row names
// and the same output line. Without STRINGWARS_DATASET it runs a synthetic workload
// (STRINGWARS_SYNTHETIC=1|2|3|4|5|40, default 2 = config C2; STRINGWARS_SEED, default 42).
//
//   make -C stringwars_amd/csrc bench && STRINGWARS_DATASET=README.md STRINGWARS_TIME=1 ./stringwars_amd/bench_similarities
#include <cinttypes>
#include <cstdio>
#include <memory>
#include <numeric>

#include "../../include/stringwars_amd.hpp"

using namespace swa;
using namespace swa::harness;

static const size_t kDefaultBatchPerCore = 256;  // bench.rs:93

struct Tokens {
    BytesTape owned;          // dataset mode
    swh_synth_t synth{};      // synthetic mode (a = queries stream, b = candidates stream)
    bool synthetic = false;
    BytesTapeView a() const { return synthetic ? BytesTapeView{synth.data_a, synth.offsets_a, synth.count} : owned.view(); }
    BytesTapeView b() const { return synthetic ? BytesTapeView{synth.data_b, synth.offsets_b, synth.count} : owned.view(); }
};

// (cross_product_cells, total_bytes) of queries [0, side) x candidates (bench.rs:216-224)
static void crossproduct_metrics(const BytesTapeView &q, const BytesTapeView &c, uint64_t &cells, uint64_t &bytes) {
    uint64_t sq = 0, sc = 0;
    for (size_t i = 0; i < q.size(); ++i) sq += q.length(i);
    for (size_t i = 0; i < c.size(); ++i) sc += c.length(i);
    cells = sq * sc; bytes = sq + sc;
}

int main() {
    BenchBudget budget = BenchBudget::from_env(5.0, 30.0);  // bench.rs:1031
    std::printf("stringwars_amd v%s (%s)\n", swh_version(), swh_capabilities());  // log_stringzilla_metadata, utils.rs:78-92
    Tokens tokens;
    std::string tmp;
    try {
        if (get_env("STRINGWARS_DATASET", tmp)) tokens.owned = load_dataset_with_default_mode("words");  // bench.rs:271
        else {
            int workload = (int)get_env_parsed("STRINGWARS_SYNTHETIC", 2);
            size_t count = (size_t)get_env_parsed("STRINGWARS_MAX_TOKENS", workload == 4 || workload == 40 ? 2048 : 1 << 20);
            const char *err = nullptr;
            swh_status_t status = swh_synth_generate(workload, (uint64_t)get_env_parsed("STRINGWARS_SEED", 42), 0, count, 0, &tokens.synth, &err);
            check(status, err);
            tokens.synthetic = true;
            std::fprintf(stderr, "Synthetic workload %d: %zu pairs\n", workload, count);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    const size_t tape_len = tokens.a().size();
    if (tape_len < 2) { std::fprintf(stderr, "Dataset must contain at least two items for comparisons.\n"); return 1; }

    // GPU scope: failure means every row is skipped, like `.ok()` at bench.rs:379/:390-399.
    DeviceScope gpu;
    std::string gpu_error;
    try { gpu = DeviceScope::gpu_device(0); } catch (const Error &e) { gpu_error = e.what(); }
    const bool have_gpu = gpu.handle() != nullptr;
    const size_t cores = have_gpu ? gpu.compute_units() : 64;  // `gpu_multiprocessor_count(0).unwrap_or(64)` bench.rs:287
    const size_t batch_gpu = swh_auto_batch_size(cores, kDefaultBatchPerCore);
    const size_t side = tokens.synthetic ? std::min<size_t>(swh_crossproduct_side(batch_gpu, 2 * tape_len), tape_len)
                                         : swh_crossproduct_side(batch_gpu, tape_len);
    std::printf("Benchmark configuration:\n- GPU batch: %zu (%zux%zu cross-product)\n- Tokens available: %zu\n\n", batch_gpu, side,
                side, tape_len);

    // queries = first `side` tokens, candidates = the next disjoint `side` (bench.rs:10-13); the synthetic
    // streams are already disjoint tapes.
    BytesTapeView q = tokens.a().subview(0, side);
    BytesTapeView c = tokens.synthetic ? tokens.b().subview(0, side) : tokens.a().subview(side, 2 * side);
    uint64_t cells = 0, bytes = 0;
    crossproduct_metrics(q, c, cells, bytes);
    auto skipped = [&](const char *name, const std::string &why) { std::printf("%s: SKIPPED (%s)\n", name, why.c_str()); };

    std::printf("# uniform\n");
    {
        const char *name = "uniform/stringwars_amd::LevenshteinDistances<1gpu>";
        if (!have_gpu) skipped(name, gpu_error);
        else try {
            LevenshteinDistances engine(gpu, 0, 1, 1, 1);
            std::vector<size_t> matrix(side * side);
            engine.compute_into(gpu, q, &c, matrix.data());
            measure_throughput(name, ReportAs::Cups, budget, [&] {
                BytesTapeView qv = tokens.a().subview(0, side);  // views rebuilt per call, bench.rs:134-139
                BytesTapeView cv = tokens.synthetic ? tokens.b().subview(0, side) : tokens.a().subview(side, 2 * side);
                engine.compute_into(gpu, qv, &cv, matrix.data());
                return WorkUnits{cells, bytes};
            });
        } catch (const Error &e) { skipped(name, e.what()); }
    }
    {
        const char *name = "uniform/stringwars_amd::LevenshteinDistancesUtf8<1gpu>";
        if (!have_gpu) skipped(name, gpu_error);
        else try {
            LevenshteinDistancesUtf8 engine(gpu, 0, 1, 1, 1);
            std::vector<size_t> matrix(side * side);
            engine.compute_into(gpu, q, &c, matrix.data());  // invalid UTF-8 -> SKIPPED, bench.rs:615-636
            measure_throughput(name, ReportAs::Cups, budget, [&] {
                engine.compute_into(gpu, q, &c, matrix.data());
                return WorkUnits{cells, bytes};  // for non-ASCII data the reference counts code points (bench.rs:230-247)
            });
        } catch (const Error &e) { skipped(name, e.what()); }
    }
    {
        // Pairwise batch over every available pair: the north-star shape (and `cudf ... edit_distance`, bench.py:602).
        const char *name = "uniform/stringwars_amd::levenshtein_pairs<1gpu>";
        if (!have_gpu) skipped(name, gpu_error);
        else try {
            LevenshteinDistances engine(gpu, 0, 1, 1, 1);
            size_t n = tokens.synthetic ? tape_len : tape_len / 2;
            BytesTapeView pa = tokens.a().subview(0, n);
            BytesTapeView pb = tokens.synthetic ? tokens.b().subview(0, n) : tokens.a().subview(n, 2 * n);
            uint64_t pcells = 0, pbytes = 0;
            for (size_t i = 0; i < n; ++i) { pcells += (uint64_t)pa.length(i) * pb.length(i); pbytes += pa.length(i) + pb.length(i); }
            std::vector<uint32_t> out(n);
            engine.pairs_into(gpu, pa, pb, out.data());
            measure_throughput(name, ReportAs::Cups, budget, [&] {
                engine.pairs_into(gpu, pa, pb, out.data());
                return WorkUnits{pcells, pbytes};
            });
            uint64_t checksum = std::accumulate(out.begin(), out.end(), (uint64_t)0);
            std::fprintf(stderr, "  %s checksum=%" PRIu64 "\n", name, checksum);  // bench.py:272
            // The same pairs on tapes prepared once, the way the reference builds its tape views once above its closures
            // (bench.rs:292-306) and sub-views them inside (bench.rs:134-139).
            const char *prepared_name = "uniform/stringwars_amd::levenshtein_pairs<prepared,1gpu>";
            PreparedTape whole_a(gpu, tokens.a()), whole_b(gpu, tokens.synthetic ? tokens.b() : tokens.a());
            std::vector<uint32_t> again(n);
            // results stay in device memory during the timed calls (the role `UnifiedMat` plays for the reference's GPU rows,
            // bench.rs:466-476) and come back once, for the comparison below
            void *out_device = nullptr;
            const char *alloc_error = nullptr;
            swh_status_t alloc_status = swh_device_alloc(gpu.handle(), n * sizeof(uint32_t) + 16, &out_device, &alloc_error);
            check(alloc_status, alloc_error);
            {   // one call outside the (filterable) measurement, like the allocating `compute` of the other rows
                PreparedTape va = whole_a.subview(0, n), vb = tokens.synthetic ? whole_b.subview(0, n) : whole_b.subview(n, 2 * n);
                engine.pairs_into(gpu, va, vb, (uint32_t *)out_device);
            }
            measure_throughput(prepared_name, ReportAs::Cups, budget, [&] {
                PreparedTape va = whole_a.subview(0, n), vb = tokens.synthetic ? whole_b.subview(0, n) : whole_b.subview(n, 2 * n);
                engine.pairs_into(gpu, va, vb, (uint32_t *)out_device);
                return WorkUnits{pcells, pbytes};
            });
            swh_copy_to_host(gpu.handle(), again.data(), out_device, n * sizeof(uint32_t), nullptr);
            swh_device_free(gpu.handle(), out_device);
            if (again != out) { std::fprintf(stderr, "error: prepared and raw tapes disagree\n"); return 2; }
            // Every visible GPU behind one scope (STRINGWARS_AMD_GPUS=0,0 lists devices by hand -- a device may repeat, a
            // testing arrangement): cells-balanced shards resident per device, RCCL gather inside the library.
            std::vector<int> devices;
            if (get_env("STRINGWARS_AMD_GPUS", tmp)) {
                for (size_t at = 0; at < tmp.size();) {
                    size_t comma = tmp.find(',', at);
                    devices.push_back(std::atoi(tmp.substr(at, comma - at).c_str()));
                    at = comma == std::string::npos ? tmp.size() : comma + 1;
                }
            } else {
                for (int d = 0; d < DeviceScope::visible_devices(); ++d) devices.push_back(d);
            }
            if (devices.size() > 1) {
                char multi_name[96];
                std::snprintf(multi_name, sizeof multi_name, "uniform/stringwars_amd::levenshtein_pairs<%zugpu>", devices.size());
                try {
                    DeviceScope gpus = DeviceScope::gpu_devices(devices);
                    LevenshteinDistances sharded_engine(gpus, 0, 1, 1, 1);
                    ShardedPairs batch(gpus, pa, pb);
                    std::vector<uint32_t> gathered(n);
                    sharded_engine.pairs_into(gpus, batch, gathered.data());
                    measure_throughput(multi_name, ReportAs::Cups, budget, [&] {
                        sharded_engine.pairs_into(gpus, batch, gathered.data());
                        return WorkUnits{pcells, pbytes};
                    });
                    swh_shard_timing_t st = gpus.shard_timing();
                    std::fprintf(stderr, "  %s slowest shard %.3f ms, gather %.3f ms\n", multi_name, st.compute_ms, st.gather_ms);
                    if (gathered != out) { std::fprintf(stderr, "error: sharded and single-GPU results disagree\n"); return 2; }
                } catch (const Error &e) { skipped(multi_name, e.what()); }
            }
        } catch (const Error &e) { skipped(name, e.what()); }
    }

    {
        // Bounded Levenshtein: out[i] = min(d, k + 1). The reference's only trace of it is the
        // STRINGWARS_ERROR_BOUND variable (README.md:311); STRINGWARS_BOUND is accepted as an alias. Bytes and code points.
        const uint32_t bound = (uint32_t)get_env_parsed("STRINGWARS_ERROR_BOUND", get_env_parsed("STRINGWARS_BOUND", 32));
        size_t n = tokens.synthetic ? tape_len : tape_len / 2;
        BytesTapeView pa = tokens.a().subview(0, n);
        BytesTapeView pb = tokens.synthetic ? tokens.b().subview(0, n) : tokens.a().subview(n, 2 * n);
        for (int utf8 = 0; utf8 < 2; ++utf8) {
            char name[96];
            std::snprintf(name, sizeof name, "uniform/stringwars_amd::levenshtein_pairs%s<k=%u,1gpu>", utf8 ? "_utf8" : "", bound);
            if (!have_gpu) { skipped(name, gpu_error); continue; }
            try {
                std::vector<uint32_t> out(n), unbounded(n);
                uint64_t pcells = 0, pbytes = 0;
                // cells counted in code points for the Utf8 row, as the reference does (bench.rs:230-247)
                auto symbols = [&](const BytesTapeView &t, size_t i) {
                    if (!utf8) return (size_t)t.length(i);
                    size_t count = 0;
                    for (uint64_t at = t.offsets[i]; at < t.offsets[i + 1]; ++at) count += (t.data[at] & 0xC0) != 0x80;
                    return count;
                };
                for (size_t i = 0; i < n; ++i) { pcells += (uint64_t)symbols(pa, i) * symbols(pb, i); pbytes += pa.length(i) + pb.length(i); }
                std::unique_ptr<LevenshteinDistances> engine(utf8 ? new LevenshteinDistancesUtf8(gpu, 0, 1, 1, 1) : new LevenshteinDistances(gpu, 0, 1, 1, 1));
                BenchStats stats = measure_throughput(name, ReportAs::Cups, budget, [&] {
                    engine->pairs_into(gpu, pa, pb, out.data(), bound);   // invalid UTF-8 throws on the first call -> SKIPPED (bench.rs:615-636)
                    return WorkUnits{pcells, pbytes};
                });
                if (stats.calls == 0) continue;                            // filtered out: no work at all (utils.rs:727-729)
                engine->pairs_into(gpu, pa, pb, unbounded.data());
                // the definition, checked on every pair of the row: bounded == min(unbounded, k + 1)
                size_t exceeded = 0;
                for (size_t i = 0; i < n; ++i) {
                    uint32_t want = unbounded[i] > bound ? bound + 1 : unbounded[i];
                    if (out[i] != want) { std::fprintf(stderr, "error: %s: pair %zu is %u, min(d, k+1) is %u\n", name, i, out[i], want); return 2; }
                    exceeded += unbounded[i] > bound;
                }
                std::fprintf(stderr, "  %s exceeded=%zu of %zu\n", name, exceeded, n);
            } catch (const Error &e) { skipped(name, e.what()); }
        }
    }

    uint8_t byte_to_class[256];
    int8_t class_costs[32][32];
    swh_unary_class_costs(2, -1, byte_to_class, &class_costs[0][0]);  // bench.rs:655-658
    struct Group { const char *header, *name, *sw_name; int open, extend; };
    const Group groups[2] = {{"# linear", "linear/stringwars_amd::NeedlemanWunschScores<1gpu>", "linear/stringwars_amd::SmithWatermanScores<1gpu>", -2, -2},   // bench.rs:342-351
                             {"# affine", "affine/stringwars_amd::NeedlemanWunschScores<1gpu>", "affine/stringwars_amd::SmithWatermanScores<1gpu>", -5, -1}};  // bench.rs:353-362
    for (const Group &g : groups) {
        std::printf("%s\n", g.header);
        if (!have_gpu) { skipped(g.name, gpu_error); skipped(g.sw_name, gpu_error); continue; }
        try {
            NeedlemanWunschScores engine(gpu, byte_to_class, class_costs, g.open, g.extend);
            std::vector<ptrdiff_t> matrix(side * side);
            engine.compute_into(gpu, q, &c, matrix.data());
            measure_throughput(g.name, ReportAs::Cups, budget, [&] {
                engine.compute_into(gpu, q, &c, matrix.data());
                return WorkUnits{cells, bytes};
            });
        } catch (const Error &e) { skipped(g.name, e.what()); }
        try {  // bench.rs:882-963
            SmithWatermanScores engine(gpu, byte_to_class, class_costs, g.open, g.extend);
            std::vector<ptrdiff_t> matrix(side * side);
            engine.compute_into(gpu, q, &c, matrix.data());
            measure_throughput(g.sw_name, ReportAs::Cups, budget, [&] {
                engine.compute_into(gpu, q, &c, matrix.data());
                return WorkUnits{cells, bytes};
            });
        } catch (const Error &e) { skipped(g.sw_name, e.what()); }
    }
    if (tokens.synthetic) swh_synth_free(&tokens.synth);
    return 0;
}
