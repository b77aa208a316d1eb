/*
 * archon_hip_test.h -- TEST-ONLY entry point of libarchon_hip.so.
 *
 * The forward / inverse pipelines choose between several routes (streaming first stage or 7-pass sort, alphabet
 * compaction, run shortcut, text rounds, pair chains, rank writer, ...), every one of which yields the same a7 order
 * (SURVEY.md 8(a0)).  The tests force each of them in turn through this call; the product never does, and the library
 * reads no routing from the environment of whoever links it.  Not part of the drop-in boundary (include/archon_hip.h).
 *
 *   name      value
 *   RESET     -            every knob back to its default
 *   FORCE_PATH   0 / 1 / -1   7-pass first stage / streaming first stage / the block decides
 *   PASS_RANGES  1..1024 / 0  ranges the streaming passes are cut into (0: one per CU)
 *   SMALL_BLOCK  bytes / -1   blocks below this size take the byte count + LSB passes instead of the streaming stage (-1: 8 MiB)
 *   ALIGNED_MIN  bytes / -1   blocks from this size on may run pass B in bucket mode (-1: 16 MiB)
 *   REL_MIN_SEG  places / -1  bucket mode moves range-relative records when a bucket's segments average at least this many places (-1: 4096)
 *   INV_SLAB, INV_SBITS, INV_WALK_WGS   inverse: slab bytes per chain, log2 rows per chain head, walk workgroups per CU
 *   INV_ROWS                            inverse: 0 = every lane of the walk stores its own 16 bytes, 1 / 2 = slabs written by quads through
 *                                       128- / 64-byte rows of LDS, -1 = the product's rule (rows above 128 MiB)
 *   NO_ALIGNED NO_BREAK_ROUND NO_CHAINS NO_DEEP_HINT NO_PACK NO_PACK_STREAM NO_PAIR_CHAINS NO_PERIOD_HINT NO_PERIOD_PROBE
 *   NO_PERIOD_STREAM NO_PROBE NO_RANK_WRITER NO_TEXT_ROUNDS NO_MID NO_SHALLOW NO_CLOSED_FORM NO_REL_RECORDS      nonzero switches the named step off
 * Returns 0, or ARCHON_E_ARG for an unknown name / a value out of range.  Process-wide; not thread-safe against
 * concurrent transforms (tests run one at a time).
 */
#ifndef ARCHON_HIP_TEST_H
#define ARCHON_HIP_TEST_H
#ifdef __cplusplus
extern "C" {
#endif
int archon_hip_test_route(const char *name, long value);
#ifdef __cplusplus
}
#endif
#endif
