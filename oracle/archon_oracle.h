/*
 * archon_oracle.h -- CPU oracle for the Archon a7 BWT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may include, link or call this.  The product
 * (libarchon_hip.so, the archon CLI) never routes through it.
 *
 * This is a plain-C restatement of the reference's algorithm for the path
 * (kvark/dark-archon bwt/a7): SA-IS induced sorting in a7's ordering convention
 * (items are reversed prefixes; end-of-string sorts above byte 255), the
 * SA->BWT gather, the LF-table build and the LF walk.  Each function cites the
 * reference file:line it follows.  Parity is PINNED: tests/test_oracle.py checks
 * it against the known answers of SURVEY.md 8(a0), a brute-force statement of
 * the definition, and the reference a7 binary compiled from /root/reference
 * (oracle/_ref, see oracle/Makefile) -- fixtures in tests/golden/.
 *
 * Conventions (bwt/a7/src/archon.h:1-4): suffix/t_index = uint32_t, byte = uint8_t.
 * Item s in 1..N names key(s) = x[s-1], x[s-2], ..., x[0], INF   (INF > 255).
 * P[0..N) = items in ascending key order.
 */
#ifndef ARCHON_ORACLE_H
#define ARCHON_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 256-bin histogram + exclusive starts R[0..256], R[256]=N.
 * Follows Constructor::makeBuckets, bwt/a7/src/archon.cpp:118-126. */
void oracle_hist256(const uint8_t *x, size_t n, uint32_t counts[256], uint32_t starts[257]);

/* The definition itself: comparator sort with a7's sufCompare
 * (bwt/a7/src/archon.cpp:83-88).  O(N log N * LCP): small inputs only. */
int oracle_sa_brute(const uint8_t *x, uint32_t n, uint32_t *P);

/* SA-IS in a7 order.  Follows Constructor<T> (bwt/a7/src/archon.cpp:784-819)
 * through its sTracking=false route: makeBuckets 118-126, findLMS 160-172,
 * inducePre 387-434, packTargetIndices 174-182, computeTargetValues 184-205,
 * solve 668-689, derive 720-778, inducePost 518-562.  Returns 0, or <0 on
 * allocation failure.  Requires 1 <= n < 2^31 (bwt/a7/src/main.cpp:27). */
int oracle_sa(const uint8_t *x, uint32_t n, uint32_t *P);

/* BWT[i] = x[P[i]] (x[0] where P[i]==N), base_id = the i with P[i]==N.
 * Follows Archon::enWrite, bwt/a7/src/archon.cpp:887-900. */
void oracle_sa_to_bwt(const uint8_t *x, uint32_t n, const uint32_t *P,
                      uint8_t *bwt, uint32_t *base_id);

/* LF-consistency check of P.  Follows Archon::validate,
 * bwt/a7/src/archon.cpp:862-874.  Returns 1 when consistent. */
int oracle_validate(const uint8_t *x, uint32_t n, const uint32_t *P);

/* Full-definition check: P is a permutation of 1..N and adjacent keys are
 * strictly increasing (a7 debug bruteCheck, archon.cpp:90-94).  O(N*LCP). */
int oracle_check_sorted(const uint8_t *x, uint32_t n, const uint32_t *P);

/* LF table: T[i] = R[bwt[i]]++ taken in order 0..base-1, base+1..N-1, base.
 * Follows Archon::deCompute + roll, bwt/a7/src/archon.cpp:905-908,917-935. */
void oracle_lf_build(const uint8_t *bwt, uint32_t n, uint32_t base_id, uint32_t *T);

/* k=base; repeat N: out(bwt[k]); k=T[k].  Follows Archon::deWrite,
 * bwt/a7/src/archon.cpp:937-943.  Returns 1 if the walk closes (k==base). */
int oracle_lf_walk(const uint8_t *bwt, uint32_t n, uint32_t base_id,
                   const uint32_t *T, uint8_t *out);

/* Convenience: forward = oracle_sa + oracle_sa_to_bwt (enCompute+enWrite);
 * inverse = oracle_lf_build + oracle_lf_walk (deCompute+deWrite).
 * P may be NULL for forward.  Return 0 on success. */
int oracle_forward(const uint8_t *x, uint32_t n, uint32_t *P, uint8_t *bwt, uint32_t *base_id);
int oracle_inverse(const uint8_t *bwt, uint32_t n, uint32_t base_id, uint8_t *out);

/* The 256-bin counting-sort scatter of tool/radix_dir/radix.c:40-44 ("X++" form):
 * dst[R[src[i]]++] = src[i] with R = exclusive starts. */
void oracle_radix_scatter(const uint8_t *src, size_t n, uint8_t *dst);

/* Seconds of process CPU time, clock() as in bwt/a7/src/main.cpp:39-41. */
/* A3: a7 findLMS (archon.cpp:160-172): per-bucket counts and the LMS items as a7 places them; returns n1 */
uint32_t oracle_lms_select(const uint8_t *x, uint32_t n, uint32_t count[256], uint32_t *items);
double oracle_clock_seconds(void);

#ifdef __cplusplus
}
#endif
#endif
