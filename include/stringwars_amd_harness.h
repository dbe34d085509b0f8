/*
 * stringwars_amd_harness.h -- host-side harness pieces that sit beside the C ABI:
 *   - deterministic synthetic workloads for the BASELINE.json configs,
 *     the stand-in for `load_dataset_with_default_mode("words")` (utils.rs:273-433,
 *     similarities/bench.rs:271) when no dataset file is available;
 *   - the reporter / batch-sizing helpers of the reference harness that are pure functions
 *     (utils.rs:487-514, :695-714, :815-819; bench.rs:113-117).
 * Plain C, no GPU needed.
 */
#ifndef STRINGWARS_AMD_HARNESS_H_
#define STRINGWARS_AMD_HARNESS_H_

#include <stddef.h>
#include <stdint.h>

#include "stringwars_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Workload ids follow BASELINE.json `configs` (C1..C5). */
typedef enum swh_workload_t {
    swh_workload_words16_k = 1,       /* C1: lengths U[1,16], a-z, 50% independent / 50% <=3 edits */
    swh_workload_tokens64_k = 2,      /* C2: lengths U[32,96], printable ASCII, 50% independent / 50% 10% edits */
    swh_workload_utf8_lines_k = 3,    /* C3: ~1 KB UTF-8 lines, script mix, b = a with U[0,64] code-point edits */
    swh_workload_protein4k_k = 4,     /* C4: lengths U[3072,5120], 20 amino-acid letters, 15% edits */
    swh_workload_short_words_k = 5,   /* C5: word lengths 1..16, mean ~6, a-z, as C1 */
    swh_workload_script_lines_k = 6,  /* article lines, ONE script each (Latin / Cyrillic / Greek / Arabic / Devanagari letters 80 %, ASCII spaces,
                                         punctuation, digits 20 %), U[700,1300] code points, a and b INDEPENDENT: what a cross-product of
                                         XLSum lines pairs up (similarities/README.md:18, :39-40) */
    swh_workload_bytes4k_k = 40       /* C4 variant: full 0..255 byte alphabet */
} swh_workload_t;

typedef struct swh_synth_t {
    uint8_t *data_a; uint64_t *offsets_a; /* count + 1 */
    uint8_t *data_b; uint64_t *offsets_b; /* count + 1 */
    size_t count;
} swh_synth_t;

/* Pair i depends only on (workload, seed, i): any [first, first+count) slice of the infinite
 * stream can be generated independently (one shard per GPU rank). `threads` <= 0 picks all cores. */
swh_status_t swh_synth_generate(int workload, uint64_t seed, uint64_t first, size_t count, int threads,
                                swh_synth_t *out, const char **error);
void swh_synth_free(swh_synth_t *tapes);

/* Seeded symmetric 256x256 i8 substitution matrix of config C4: diagonal U[4,11],
 * off-diagonal U[-4,3], rows/columns of bytes outside `alphabet` (NULL = all bytes) = -4. */
void swh_synth_matrix(uint64_t seed, const char *alphabet, int8_t *matrix_256x256);

/* `unary_class_costs(match, mismatch)` (bench.rs:95-108): byte -> class = byte % 32. */
void swh_unary_class_costs(int8_t match, int8_t mismatch, uint8_t *byte_to_class_256, int8_t *class_costs_32x32);

/* Cells-balanced contiguous cuts of a pairwise batch in HOST tapes: `cuts[0..shards]`, shard r = pairs
 * [cuts[r], cuts[r+1]), balanced on the prefix sum of len(a_i)*len(b_i). What `swh_sharded_prepare_*` uses. */
void swh_shard_cuts_u32tape(const swh_tape_u32_t *a, const swh_tape_u32_t *b, size_t shards, size_t *cuts);
void swh_shard_cuts_u64tape(const swh_tape_u64_t *a, const swh_tape_u64_t *b, size_t shards, size_t *cuts);

/* `crossproduct_side(budget, tape_len)` (bench.rs:113-117). */
size_t swh_crossproduct_side(size_t budget, size_t tape_len);
/* `auto_batch_size(cores, default_base)` (utils.rs:815-819) with STRINGWARS_BATCH_PER_CORE. */
size_t swh_auto_batch_size(size_t cores, size_t default_base);
/* `format_si_rate` / `format_byte_rate` / `format_seconds` (utils.rs:487-514, :695-714);
 * each writes a NUL-terminated string into `buffer` and returns its length. */
size_t swh_format_si_rate(double rate, const char *unit, int space_before_unit, char *buffer, size_t capacity);
size_t swh_format_seconds(double seconds, char *buffer, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* STRINGWARS_AMD_HARNESS_H_ */
