/*
 * stringwars_amd.h -- C ABI of the MI355X-native batched edit-distance backend.
 *
 * This is the drop-in boundary for the similarity hot path of ashvardanian/StringWars
 * (`similarities/bench.rs`, `similarities/bench.py`). Every entry point below replaces one call
 * the reference makes into `stringzilla::szs` (whose own C ABI, `stringzillas.h`, is not vendored
 * under /root/reference) or into the per-pair CPU baselines. The `file:line` in each comment is
 * the reference call site that the symbol serves. Plain pointers and sizes only; no C++ or torch
 * types cross this boundary. All functions are `extern "C"`, return a status code and, when
 * `error` is non-NULL, leave a pointer to a static NUL-terminated message in `*error`.
 *
 * Memory: every pointer inside a tape and every `out` pointer may be DEVICE memory (hipMalloc,
 * torch CUDA tensors) -- used in place, the steady state the benchmark times -- or HOST memory
 * (pageable, pinned, managed), in which case the call stages it through PCIe itself. Calls are
 * synchronous (results are visible on return, like `compute_into`, bench.rs:478-486) unless the
 * scope was switched to asynchronous mode with `swh_scope_set_async`. "Visible on return" for an
 * output in device memory: every result has been written through to memory and acknowledged, any
 * stream, device or copy may read it; the scope's own stream may still be retiring the kernel
 * (the call returns on the kernel's summary, not on the stream: DESIGN.md section 3).
 */
#ifndef STRINGWARS_AMD_H_
#define STRINGWARS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWH_VERSION_MAJOR 0
#define SWH_VERSION_MINOR 1
#define SWH_VERSION_PATCH 0

/* Status codes. 0 is success; the rest map onto the `Result<_, E: Display>` errors of the szs
 * wrappers (bench.rs:390-399 `.ok()`, :480-485 `panic!("{}", error)`, :632-635 SKIPPED). */
typedef enum swh_status_t {
    swh_success_k = 0,
    swh_bad_alloc_k = 1,
    swh_invalid_argument_k = 2,
    swh_invalid_utf8_k = 3,        /* C callers can pass bytes that Rust `&str` never could */
    swh_unsupported_length_k = 4,  /* "the engine may decline inputs beyond its supported length" bench.rs:394 */
    swh_no_device_k = 5,           /* no gfx950 device / HIP runtime unavailable */
    swh_device_error_k = 6,        /* a HIP call failed; message carries hipGetErrorString */
    swh_not_implemented_k = 7,
    swh_rccl_error_k = 8           /* RCCL could not be loaded, or a collective failed; message carries ncclGetErrorString */
} swh_status_t;

#define SWH_UNBOUNDED UINT32_MAX

/* Which DP algorithm a Levenshtein engine runs for unit costs (0,1,1,1). */
typedef enum swh_algorithm_t {
    swh_algorithm_auto_k = 0,        /* fastest measured path per length class */
    swh_algorithm_wavefront_k = 1,   /* anti-diagonal wavefront, register-tiled, DPP hand-off */
    swh_algorithm_bitparallel_k = 2, /* Myers/Hyyro bit-vectors, systolic over 32-row blocks; globally planned (device pre-pass) */
    swh_algorithm_tiled_k = 3        /* the same recurrence, every workgroup plans its own tile: no pre-pass, no host round trip */
} swh_algorithm_t;

/* ---- Device scope: replaces `DeviceScope::gpu_device(0)` (bench.rs:379, :652, :978). ------- */
typedef struct swh_scope_s *swh_scope_t;

/* Creates a scope on HIP device `device` with its own stream and scratch arena. */
swh_status_t swh_scope_init_gpu(int device, swh_scope_t *scope, const char **error);
/* Same, but work is enqueued on a caller-owned `hipStream_t` (e.g. torch's current stream). */
swh_status_t swh_scope_init_gpu_stream(int device, void *hip_stream, swh_scope_t *scope, const char **error);
/* `DeviceScope::cpu_cores(n)` (bench.rs:376-378). This backend has no CPU path: always returns
 * swh_not_implemented_k so a harness prints SKIPPED instead of silently running a fallback. */
swh_status_t swh_scope_init_cpu(size_t cores, swh_scope_t *scope, const char **error);
/* Several GPUs of ONE node behind one scope (single process): a member scope per device and an RCCL communicator
 * (`ncclCommInitAll`); the `<Ngpu>` rows beside the reference's `<1gpu>` ones (bench.rs:581-606). Ordinary engine calls on
 * such a scope run on its first device; the `*_sharded` calls below split a batch over all of them. `devices` may name one
 * device several times (members then share it and exchange by copies instead of RCCL -- a testing arrangement). */
swh_status_t swh_device_count(int *count);
swh_status_t swh_scope_init_gpus(const int *devices, int count, swh_scope_t *scope, const char **error);
swh_status_t swh_scope_device_count(swh_scope_t scope, size_t *devices);
swh_status_t swh_scope_free(swh_scope_t scope);
/* Compute-unit count for `auto_batch_size` (utils.rs:815-819, :826-836: one SM == one core). */
swh_status_t swh_scope_compute_units(swh_scope_t scope, size_t *compute_units);
/* async != 0: engine calls only enqueue; `swh_scope_synchronize` makes results visible. */
swh_status_t swh_scope_set_async(swh_scope_t scope, int async);
swh_status_t swh_scope_synchronize(swh_scope_t scope, const char **error);
/* Pipelined mode (implies async): successive engine calls alternate between two internal lanes (stream + scratch), so
 * the host-side planning of one call overlaps the DP kernel of the previous one. Ordering: a call starts after
 * everything that was enqueued on the scope's own stream (the one given to swh_scope_init_gpu_stream) before it --
 * producers of its inputs AND earlier consumers of the output buffer it overwrites. `swh_scope_join` makes that stream
 * wait for the latest call, for consumers ordered on it (e.g. an RCCL gather); `swh_scope_synchronize` waits for
 * everything. Buffers touched by streams the scope does not know about must be idle. */
swh_status_t swh_scope_set_pipelined(swh_scope_t scope, int enabled, const char **error);
swh_status_t swh_scope_join(swh_scope_t scope, const char **error);

/* What a scope REMEMBERS between calls. A scope learns from its calls -- the longest strings of the previous batch (raw tapes
 * then run without a planning pre-pass), the byte totals and ASCII-ness of the raw UTF-8 tapes it last staged, whether the
 * two-stage (band first) schedule paid off, which engine / tapes the small-alphabet alignment kernels could not take -- and
 * every belief is CHECKED on the device: a wrong one costs the call a redo, never a result. The reference's engines keep no
 * such state (`compute_into` is a pure function of its arguments, bench.rs:478-486); a caller who wants that -- or a timing
 * that does not depend on what ran before -- calls `swh_scope_forget` first. `swh_scope_describe` writes the beliefs as one
 * line of text (`key=value` pairs; truncated to `capacity`, always NUL-terminated) for logs and tests. */
swh_status_t swh_scope_forget(swh_scope_t scope);
swh_status_t swh_scope_describe(swh_scope_t scope, char *text, size_t capacity);

/* Kernel timing (hipEvents on the scope's stream around every kernel of the last engine call).
 * Used by bench.py's roofline object; off by default. */
typedef struct swh_timing_t {
    double total_ms;          /* all kernels of the last call, first-start to last-stop */
    double dominant_ms;       /* longest single kernel */
    double compute_ms;        /* time covered by the DP kernels (union of their intervals; everything but the planning / staging pre-pass) */
    char dominant_name[64];   /* its name */
    uint64_t cells;           /* nominal DP cells of the call (sum len_s(a)*len_s(b)) */
    uint64_t bytes;           /* algorithmic HBM bytes: symbols + offsets + results */
    uint32_t kernels;         /* number of kernel launches */
} swh_timing_t;
swh_status_t swh_scope_set_profiling(swh_scope_t scope, int enabled);
swh_status_t swh_scope_last_timing(swh_scope_t scope, swh_timing_t *timing);
/* Sums over every engine call since profiling was last switched on -- also the asynchronous and pipelined ones, whose
 * events are read when their lane is next used or the scope is synchronized. bench.py's roofline uses the mean of
 * `dominant_ms` over a pipelined run, i.e. kernel durations under the same conditions rocprofv3 sees them. */
typedef struct swh_timing_totals_t {
    double total_ms, dominant_ms, compute_ms;   /* sums of the per-call figures of swh_timing_t */
    uint64_t calls;
} swh_timing_totals_t;
swh_status_t swh_scope_timing_totals(swh_scope_t scope, swh_timing_totals_t *totals);

/* ---- Memory: `UnifiedAlloc` / `UnifiedMat` parity (bench.rs:292-295, :466-468). ------------ */
/* Host-visible, device-readable allocation (pinned + mapped). */
swh_status_t swh_unified_alloc(swh_scope_t scope, size_t bytes, void **pointer, const char **error);
swh_status_t swh_unified_free(swh_scope_t scope, void *pointer);
/* Explicit device allocation / copies for device-resident tapes (the timed steady state). */
swh_status_t swh_device_alloc(swh_scope_t scope, size_t bytes, void **pointer, const char **error);
swh_status_t swh_device_free(swh_scope_t scope, void *pointer);
swh_status_t swh_copy_to_device(swh_scope_t scope, void *device_dst, const void *host_src, size_t bytes,
                                const char **error);
swh_status_t swh_copy_to_host(swh_scope_t scope, void *host_dst, const void *device_src, size_t bytes,
                              const char **error);

/* ---- Tapes: `BytesTapeView<u64>` / `AnyBytesTape::View64` (bench.rs:62, :134-143, :292-306). */
/* Arrow-style: `offsets` has `count + 1` entries, string i is data[offsets[i] .. offsets[i+1]). */
typedef struct swh_tape_u32_t { const uint8_t *data; const uint32_t *offsets; size_t count; } swh_tape_u32_t;
typedef struct swh_tape_u64_t { const uint8_t *data; const uint64_t *offsets; size_t count; } swh_tape_u64_t;

/* ---- Prepared tapes: `BytesTape<u64, UnifiedAlloc>` filled once and `CharsTapeView::try_from(bytes_view)` validated
 *      once, both OUTSIDE the timed closures (bench.rs:292-306), then sub-viewed per iteration (bench.rs:134-139).
 * `swh_tape_prepare_*` makes a tape resident on the scope's device (host tapes are uploaded; device tapes are used in
 * place and must outlive the handle), measures it (count, symbols, longest string) and, with `utf8 != 0`, validates
 * and decodes it to code points -- invalid UTF-8 fails HERE with swh_invalid_utf8_k, as `try_into` does, not in every
 * engine call. Engine calls on prepared views skip the per-call decode and, knowing the longest string, the planning
 * pre-pass and its host round trip. */
typedef struct swh_prepared_s *swh_prepared_t;
typedef struct swh_prepared_info_t {
    size_t count;             /* strings */
    uint64_t bytes, symbols;  /* tape bytes; symbols = bytes, or code points for a UTF-8 tape */
    uint32_t longest;         /* longest string in symbols */
    int utf8, ascii;          /* prepared as UTF-8; every code point is a single byte */
} swh_prepared_info_t;
swh_status_t swh_tape_prepare_u32(swh_scope_t scope, const swh_tape_u32_t *tape, int utf8, swh_prepared_t *prepared,
                                  const char **error);
swh_status_t swh_tape_prepare_u64(swh_scope_t scope, const swh_tape_u64_t *tape, int utf8, swh_prepared_t *prepared,
                                  const char **error);
swh_status_t swh_prepared_info(swh_prepared_t prepared, swh_prepared_info_t *info);
swh_status_t swh_prepared_free(swh_prepared_t prepared);
/* Strings [first, first + count) of a prepared tape: `BytesTapeView::subview(lo, hi)` (bench.rs:134-139). */
typedef struct swh_prepared_view_t { swh_prepared_t tape; size_t first, count; } swh_prepared_view_t;

/* ---- Levenshtein: `LevenshteinDistances::new(&scope, 0, 1, 1, 1)` (bench.rs:382-393). ------ */
typedef struct swh_levenshtein_s *swh_levenshtein_t;
/* Costs are non-negative; gap of length k costs open + (k-1)*extend. */
swh_status_t swh_levenshtein_init(swh_scope_t scope, int match, int mismatch, int open, int extend,
                                  swh_levenshtein_t *engine, const char **error);
swh_status_t swh_levenshtein_free(swh_levenshtein_t engine);
swh_status_t swh_levenshtein_set_algorithm(swh_levenshtein_t engine, swh_algorithm_t algorithm);

/* Pairwise batch, bytes as symbols: out[i] = min(d(a_i, b_i), bound + 1); bound == SWH_UNBOUNDED
 * disables the cutoff. Replaces the per-pair loops `rapidfuzz::levenshtein::distance`
 * (bench.rs:404-423) / `bio::levenshtein` (bench.rs:443-459) and the batched pairwise
 * `cudf ... str.edit_distance` (similarities/bench.py:596-604). `a.count == b.count`.
 * `out_stride_bytes` is the distance between consecutive results (>= 4; 0 means 4). */
swh_status_t swh_levenshtein_pairs_u32tape(swh_levenshtein_t engine, swh_scope_t scope, const swh_tape_u32_t *a,
                                           const swh_tape_u32_t *b, uint32_t bound, uint32_t *out,
                                           size_t out_stride_bytes, const char **error);
swh_status_t swh_levenshtein_pairs_u64tape(swh_levenshtein_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                           const swh_tape_u64_t *b, uint32_t bound, uint32_t *out,
                                           size_t out_stride_bytes, const char **error);
/* Same with Unicode scalar values as symbols: `LevenshteinDistancesUtf8` (bench.rs:386-399),
 * `rapidfuzz::levenshtein<Chars>` (bench.rs:425-441). Invalid UTF-8 -> swh_invalid_utf8_k. */
swh_status_t swh_levenshtein_utf8_pairs_u32tape(swh_levenshtein_t engine, swh_scope_t scope,
                                                const swh_tape_u32_t *a, const swh_tape_u32_t *b, uint32_t bound,
                                                uint32_t *out, size_t out_stride_bytes, const char **error);
swh_status_t swh_levenshtein_utf8_pairs_u64tape(swh_levenshtein_t engine, swh_scope_t scope,
                                                const swh_tape_u64_t *a, const swh_tape_u64_t *b, uint32_t bound,
                                                uint32_t *out, size_t out_stride_bytes, const char **error);
/* Dense cross-product `a.count x b.count`, row-major `size_t`, the literal shape of
 * `engine.compute_into(&scope, AnyBytesTape::View64(q), Some(AnyBytesTape::View64(c)), &mut matrix)`
 * (bench.rs:478-486, :599-603). `b == NULL` means the symmetric self-product. */
swh_status_t swh_levenshtein_cross_u64tape(swh_levenshtein_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                           const swh_tape_u64_t *b, size_t *out, size_t row_stride_bytes,
                                           const char **error);
swh_status_t swh_levenshtein_utf8_cross_u64tape(swh_levenshtein_t engine, swh_scope_t scope,
                                                const swh_tape_u64_t *a, const swh_tape_u64_t *b, size_t *out,
                                                size_t row_stride_bytes, const char **error);

/* The same calls on prepared views (both tapes prepared on the scope's device, both as bytes or both as UTF-8 -- the
 * symbols are then code points, as for `LevenshteinDistancesUtf8`). `b == NULL` in the cross-product means a x a. */
swh_status_t swh_levenshtein_pairs_prepared(swh_levenshtein_t engine, swh_scope_t scope, const swh_prepared_view_t *a,
                                            const swh_prepared_view_t *b, uint32_t bound, uint32_t *out,
                                            size_t out_stride_bytes, const char **error);
swh_status_t swh_levenshtein_cross_prepared(swh_levenshtein_t engine, swh_scope_t scope, const swh_prepared_view_t *a,
                                            const swh_prepared_view_t *b, size_t *out, size_t row_stride_bytes,
                                            const char **error);

/* ---- One batch over the GPUs of a multi-device scope (BASELINE config 5). --------------------------------
 * `swh_sharded_prepare_*`: HOST tapes of equal count are cut into contiguous shards balanced on the prefix sum of
 * len(a_i)*len(b_i) (DP cells, not pair counts); shard r is uploaded to and prepared on device r. The handle is the steady
 * state: every `swh_levenshtein_pairs_sharded` call scores each shard on its device (no exchange during the DP), gathers
 * the u32 distances to the first device with one ncclSend / ncclRecv group over xGMI and copies them to `out` (host or
 * first-device memory, pair order). `swh_scope_shard_timing`: slowest shard and the gather of the last call. */
typedef struct swh_sharded_s *swh_sharded_t;
swh_status_t swh_sharded_prepare_u32tape(swh_scope_t scope, const swh_tape_u32_t *a, const swh_tape_u32_t *b, int utf8,
                                         swh_sharded_t *sharded, const char **error);
swh_status_t swh_sharded_prepare_u64tape(swh_scope_t scope, const swh_tape_u64_t *a, const swh_tape_u64_t *b, int utf8,
                                         swh_sharded_t *sharded, const char **error);
swh_status_t swh_sharded_free(swh_sharded_t sharded);
/* the `devices + 1` pair indices where the shards begin / end */
swh_status_t swh_sharded_cuts(swh_sharded_t sharded, size_t *cuts, size_t capacity);
swh_status_t swh_levenshtein_pairs_sharded(swh_levenshtein_t engine, swh_scope_t scope, swh_sharded_t sharded, uint32_t bound,
                                           uint32_t *out, const char **error);
/* one-shot: shard + upload + score + gather + free */
swh_status_t swh_levenshtein_pairs_sharded_u64tape(swh_levenshtein_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                                   const swh_tape_u64_t *b, uint32_t bound, uint32_t *out, const char **error);
typedef struct swh_shard_timing_t {
    double compute_ms;   /* slowest shard, first kernel start to last kernel end on its device */
    double gather_ms;    /* the RCCL gather on the first device's stream */
    uint64_t cells, pairs;
} swh_shard_timing_t;
swh_status_t swh_scope_shard_timing(swh_scope_t scope, swh_shard_timing_t *timing);

/* ---- Needleman-Wunsch: `NeedlemanWunschScores::new(&scope, &byte_to_class, &class_costs,
 *      open, extend)` (bench.rs:658-670, :985-997); scores are max-plus, gaps usually negative. */
typedef struct swh_nw_s *swh_nw_t;
/* Full 256x256 `i8` substitution matrix, row = symbol of a, column = symbol of b (config C4). */
swh_status_t swh_nw_init(swh_scope_t scope, const int8_t *substitution_256x256, int open, int extend,
                         swh_nw_t *engine, const char **error);
/* The reference's 32-class form (`unary_class_costs`, bench.rs:95-108): expands to 256x256. */
swh_status_t swh_nw_init_classes(swh_scope_t scope, const uint8_t *byte_to_class_256,
                                 const int8_t *class_costs_32x32, int open, int extend, swh_nw_t *engine,
                                 const char **error);
swh_status_t swh_nw_free(swh_nw_t engine);
/* Pairwise batch of global alignment scores (`bio ... Aligner::global(a, b).score`,
 * bench.rs:746-765, batched). */
swh_status_t swh_nw_pairs_u32tape(swh_nw_t engine, swh_scope_t scope, const swh_tape_u32_t *a,
                                  const swh_tape_u32_t *b, int32_t *out, size_t out_stride_bytes,
                                  const char **error);
swh_status_t swh_nw_pairs_u64tape(swh_nw_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                  const swh_tape_u64_t *b, int32_t *out, size_t out_stride_bytes,
                                  const char **error);
/* Cross-product into `isize` (`UnifiedMat<isize>`, bench.rs:814-821, :872-876). */
swh_status_t swh_nw_cross_u64tape(swh_nw_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                  const swh_tape_u64_t *b, ptrdiff_t *out, size_t row_stride_bytes,
                                  const char **error);

swh_status_t swh_nw_pairs_prepared(swh_nw_t engine, swh_scope_t scope, const swh_prepared_view_t *a,
                                   const swh_prepared_view_t *b, int32_t *out, size_t out_stride_bytes, const char **error);
swh_status_t swh_nw_cross_prepared(swh_nw_t engine, swh_scope_t scope, const swh_prepared_view_t *a,
                                   const swh_prepared_view_t *b, ptrdiff_t *out, size_t row_stride_bytes, const char **error);

/* ---- Smith-Waterman: `SmithWatermanScores::new(&scope, &byte_to_class, &class_costs, open, extend)`
 *      (bench.rs:882-963). Local alignment score: max over all cells, floored at 0;
 *      same matrix / gap conventions as Needleman-Wunsch. */
typedef struct swh_sw_s *swh_sw_t;
swh_status_t swh_sw_init(swh_scope_t scope, const int8_t *substitution_256x256, int open, int extend,
                         swh_sw_t *engine, const char **error);
swh_status_t swh_sw_init_classes(swh_scope_t scope, const uint8_t *byte_to_class_256,
                                 const int8_t *class_costs_32x32, int open, int extend, swh_sw_t *engine,
                                 const char **error);
swh_status_t swh_sw_free(swh_sw_t engine);
swh_status_t swh_sw_pairs_u32tape(swh_sw_t engine, swh_scope_t scope, const swh_tape_u32_t *a,
                                  const swh_tape_u32_t *b, int32_t *out, size_t out_stride_bytes,
                                  const char **error);
swh_status_t swh_sw_pairs_u64tape(swh_sw_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                  const swh_tape_u64_t *b, int32_t *out, size_t out_stride_bytes,
                                  const char **error);
swh_status_t swh_sw_cross_u64tape(swh_sw_t engine, swh_scope_t scope, const swh_tape_u64_t *a,
                                  const swh_tape_u64_t *b, ptrdiff_t *out, size_t row_stride_bytes,
                                  const char **error);

swh_status_t swh_sw_pairs_prepared(swh_sw_t engine, swh_scope_t scope, const swh_prepared_view_t *a,
                                   const swh_prepared_view_t *b, int32_t *out, size_t out_stride_bytes, const char **error);
swh_status_t swh_sw_cross_prepared(swh_sw_t engine, swh_scope_t scope, const swh_prepared_view_t *a,
                                   const swh_prepared_view_t *b, ptrdiff_t *out, size_t row_stride_bytes, const char **error);

/* The reference's own call shape -- `compute_into(queries, candidates, &mut matrix)`, bench.rs:478-486 -- over every GPU of a
 * multi-device scope: the queries are cut into row blocks of equal symbol counts, block r and all candidates are prepared on
 * device r, every device fills its rows and copies them straight into `matrix` (host memory, or memory of the first device).
 * No collective: the row blocks are disjoint. Levenshtein engines take bytes or (utf8 = 1) code points. */
typedef struct swh_sharded_cross_s *swh_sharded_cross_t;
swh_status_t swh_sharded_cross_prepare_u64tape(swh_scope_t scope, const swh_tape_u64_t *queries, const swh_tape_u64_t *candidates, int utf8,
                                               swh_sharded_cross_t *product, const char **error);
swh_status_t swh_sharded_cross_free(swh_sharded_cross_t product);
swh_status_t swh_levenshtein_cross_sharded(swh_levenshtein_t engine, swh_scope_t scope, swh_sharded_cross_t product, size_t *matrix,
                                           size_t row_stride_bytes, const char **error);
swh_status_t swh_nw_cross_sharded(swh_nw_t engine, swh_scope_t scope, swh_sharded_cross_t product, ptrdiff_t *matrix, size_t row_stride_bytes,
                                  const char **error);
swh_status_t swh_sw_cross_sharded(swh_sw_t engine, swh_scope_t scope, swh_sharded_cross_t product, ptrdiff_t *matrix, size_t row_stride_bytes,
                                  const char **error);

/* Needleman-Wunsch / Smith-Waterman scores of a sharded batch (swh_sharded_prepare_*): the engine's matrix is cloned to every
 * device of the scope on first use -- the `<Ngpu>` twin of the bench.rs:658-670 / :882-963 rows (the engines'
 * read-only state is replicated); scores gathered to the first device like the distances of swh_levenshtein_pairs_sharded */
swh_status_t swh_nw_pairs_sharded(swh_nw_t engine, swh_scope_t scope, swh_sharded_t sharded, int32_t *out, const char **error);
swh_status_t swh_sw_pairs_sharded(swh_sw_t engine, swh_scope_t scope, swh_sharded_t sharded, int32_t *out, const char **error);

/* ---- Introspection: `log_stringzilla_metadata` (utils.rs:78-92). --------------------------- */
const char *swh_version(void);
/* Comma-separated capability string, e.g. "gfx950,hip,wavefront,bitparallel,banded,utf8,...". */
const char *swh_capabilities(void);

#ifdef __cplusplus
}
#endif
#endif /* STRINGWARS_AMD_H_ */
