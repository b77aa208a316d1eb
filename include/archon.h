/*
 * archon.h -- the reference's block-coder object as a C API (and, for C++ callers,
 * class Archon in dark-archon_amd/host/archon_host.h with the reference's own
 * method names).  Drop-in for class Archon of kvark/dark-archon
 * (bwt/a7/src/archon.h:8-29): same call order (read -> compute -> [validate] -> write),
 * same in-place contract (after en_compute P[0..N) holds the suffix array, items
 * 1..N), same return conventions (int, 0 / N), same file layout (N BWT bytes then
 * the primary index as uint32 little-endian, archon.cpp:895,898).
 *
 * Host memory stays at the reference's 5N + O(1) (README.md:15): str = N+1 bytes,
 * P = (N + reserve) * 4 bytes; everything else lives in HBM behind libarchon_hip.
 * The compute methods run ONLY on the GPU (include/archon_hip.h); no CPU fallback.
 */
#ifndef ARCHON_H
#define ARCHON_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct archon archon_t;

uint32_t archon_estimate_reserve(uint32_t n);          /* Archon::estimateReserve, archon.cpp:827-838 */
archon_t *archon_create(uint32_t nmax);                /* Archon::Archon,          archon.cpp:840-848 */
void      archon_destroy(archon_t *a);                 /* Archon::~Archon,         archon.cpp:850-853 */
unsigned  archon_count_memory(const archon_t *a);      /* Archon::countMemory,     archon.cpp:855-857 */
int       archon_validate(archon_t *a);                /* Archon::validate,        archon.cpp:862-874; 1 = OK */
int       archon_en_read(archon_t *a, FILE *fx, uint32_t ns);   /* enRead,    archon.cpp:876-880; returns N */
int       archon_en_compute(archon_t *a);                       /* enCompute, archon.cpp:882-885; 0 or <0 */
int       archon_en_write(archon_t *a, FILE *fx);               /* enWrite,   archon.cpp:887-900 */
int       archon_de_read(archon_t *a, FILE *fx, uint32_t ns);   /* deRead,    archon.cpp:910-915; returns N */
int       archon_de_compute(archon_t *a);                       /* deCompute, archon.cpp:917-935 */
int       archon_de_write(archon_t *a, FILE *fx);               /* deWrite,   archon.cpp:937-943 */

/* extras the reference keeps private (archon.h:9-12): read-only views for tests */
const uint32_t *archon_sa(const archon_t *a);          /* P[0..N) after en_compute */
uint32_t  archon_base_id(const archon_t *a);
uint32_t  archon_length(const archon_t *a);
void      archon_set_device(archon_t *a, int dev);     /* default 0 (or $ARCHON_DEVICE); before the first en_compute */
int       archon_last_error(const archon_t *a);        /* the library's code of the object's last compute / validate call */

/* ---- optional post-BWT stage of the container CLI (`archon e -m -b<size>`): move-to-front, zero runs and an
 * order-0 canonical Huffman code over one piece of BWT output.  PARITY UNPINNED: the reference
 * has no such stage (README.md:2 only promises one); self-consistent, round-trip tested, host-side -- not part of
 * the GPU hot path and not a fallback for it. */
size_t    archon_post_bound(size_t n);                                          /* worst-case encoded bytes for n input bytes */
size_t    archon_post_encode(const uint8_t *bwt, size_t n, uint8_t *out);       /* returns encoded bytes */
int       archon_post_decode(const uint8_t *in, size_t in_bytes, uint8_t *bwt, size_t n);   /* 0 ok, -1 malformed */

#ifdef __cplusplus
}
#endif
#endif
