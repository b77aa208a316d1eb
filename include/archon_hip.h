/*
 * archon_hip.h -- C ABI of libarchon_hip.so: the MI355X (gfx950) kernels behind
 * the Archon a7 BWT hot path.  Plain pointers and sizes, no exceptions, no
 * torch/HIP types in any signature (a HIP stream crosses as void*).
 *
 * What each entry point replaces in kvark/dark-archon (paths under bwt/a7/src):
 *
 *   archon_hip_forward      Archon::enCompute (archon.cpp:882-885 -> Constructor<byte>
 *                           784-819) + the gather loop of Archon::enWrite (887-900)
 *   archon_hip_inverse      Archon::deCompute (917-935) + the LF walk of Archon::deWrite (937-943)
 *   archon_hip_hist256      Constructor::makeBuckets (118-126); tool/radix_dir/radix.c:31-36
 *   archon_hip_validate     Archon::validate (862-874)
 *   archon_hip_sa_to_bwt    the gather loop of Archon::enWrite alone (887-900), for a caller's own SA
 *   archon_hip_radix_scatter  the counting-sort scatter of tool/radix_dir/radix.c:40-44
 *   archon_hip_lms_select   Constructor::findLMS (160-172): the subset a7 sorts directly (a4 IT-2: bwt/a4/src/archon.c:163-169)
 *
 * Ordering convention ("a7 order"): item s in 1..N names the
 * reversed prefix x[s-1],x[s-2],...,x[0],INF with INF > 255; sa[0..N) lists the
 * items in ascending key order; bwt[i] = x[sa[i]] (x[0] where sa[i]==N);
 * *base_id = the i with sa[i]==N.
 *
 * Error model: 0 = ok, negative = ARCHON_E_* below (the reference returns int
 * from every Archon method, archon.h:16-28).  The library owns device memory and
 * streams: up to eight compute contexts per device (arena, staging buffers, stream each:
 * about 145 MB of fixed tables plus 27 N .. 96 N of arena for the largest block it has
 * seen, kept until archon_hip_release), created lazily.  A host thread is bound to one
 * context PER DEVICE: the k-th thread that comes to device d takes context k mod 2 of
 * that device (threads never spread over more than two by themselves), or the one it
 * names with archon_hip_bind_context (0..7: the batch entry points and the container's
 * worker pools do).  One thread sees strictly serial behaviour; two
 * threads feeding one GPU overlap one block's copies with the other's kernels;
 * separate devices run concurrently.  A context holds NO results between calls:
 * what outlives a call (the resident block of a block-coder object) belongs to an
 * archon_hip_block handle, and the statistics of a call to the thread that made it.
 * Callers own every buffer they pass.  There is NO CPU fallback: without a HIP
 * device every compute entry point returns ARCHON_E_NODEVICE.
 *
 * Limits: 1 <= n <= ARCHON_HIP_MAX_N (the reference needs n < 2^30 for its
 * default tracking path, archon.cpp:802).
 */
#ifndef ARCHON_HIP_H
#define ARCHON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARCHON_HIP_MAX_N      0x3FFFFF00u

#define ARCHON_OK             0
#define ARCHON_E_ARG        (-1)   /* null pointer / n out of range / bad base_id */
#define ARCHON_E_NODEVICE   (-2)   /* no HIP device, or dev out of range */
#define ARCHON_E_NOMEM      (-3)   /* device or host allocation failed */
#define ARCHON_E_HIP        (-4)   /* a HIP runtime call failed (see archon_hip_last_error) */
#define ARCHON_E_INTERNAL   (-5)   /* device-side consistency flag raised (e.g. look-back spin bound) */
#define ARCHON_E_CORRUPT    (-6)   /* inverse: LF walk does not close (not a BWT of this format) */

/* number of HIP devices visible (0 when none); never fails */
int archon_hip_device_count(void);

/* human-readable text for the last error raised on the calling thread */
const char *archon_hip_last_error(void);

/* ---- host-buffer entry points (what the C host / cgo-style bindings call) ---- */

/* x[n] -> sa (optional, may be NULL), bwt[n], *base_id.  All host pointers. */
int archon_hip_forward(const uint8_t *x, uint32_t n, uint32_t *sa_or_null,
                       uint8_t *bwt, uint32_t *base_id, int dev);

/* ---- resident blocks: the device side of ONE block-coder object (class Archon, bwt/a7/src/archon.h:8-29) ----------
 * The reference's caller runs read -> enCompute -> validate -> enWrite on one object (main.cpp:39-46), and the host
 * must stay within 5N + O(1) bytes.  So enCompute leaves the block, its suffix array and its BWT resident in HBM, in
 * buffers that belong to the handle (6N device bytes): validate then checks what is there (no upload, no second
 * gather of x[sa[i]]), enWrite reads the BWT back in pieces through any O(1) bounce buffer.  Any number of handles,
 * on any threads; a handle is used by one thread at a time (as an Archon object is: a7 is re-entrant per object). */
typedef struct archon_hip_block archon_hip_block;
int  archon_hip_block_create(int dev, archon_hip_block **out);
void archon_hip_block_destroy(archon_hip_block *b);
/* Archon::enCompute (archon.cpp:882-885): x[n] (host) -> sa[n] (host, optional) and *base_id; x, SA, BWT stay resident */
int  archon_hip_block_forward(archon_hip_block *b, const uint8_t *x, uint32_t n, uint32_t *sa_or_null, uint32_t *base_id);
/* the gather loop of Archon::enWrite (archon.cpp:887-900), already done on the device: bytes [offset, offset+len) of the BWT */
int  archon_hip_block_read_bwt(archon_hip_block *b, uint32_t offset, uint32_t len, uint8_t *dst);
/* Archon::validate (archon.cpp:862-874) on the resident block: every sa[i] in 1..n, exactly one n and on the primary
 * row, the LF rule for every other row, the BWT a permutation of the block.  1 = consistent, 0 = not, <0 = error
 * (ARCHON_E_ARG when the last forward on the handle kept no suffix array). */
int  archon_hip_block_validate(archon_hip_block *b);
/* (archon_hip_stats is defined under "measurement" below) */
struct archon_hip_stats;
int  archon_hip_block_stats(archon_hip_block *b, struct archon_hip_stats *out);      /* of the handle's last forward */

/* The same keyed by device, on a default handle that belongs to the CALLING THREAD (so two threads never see each
 * other's BWT): forward_keep = block_forward, read_bwt = block_read_bwt, validate_keep = block_validate. */
int archon_hip_forward_keep(const uint8_t *x, uint32_t n, uint32_t *sa_or_null,
                            uint32_t *base_id, int dev);
int archon_hip_read_bwt(int dev, uint32_t offset, uint32_t len, uint8_t *dst);
int archon_hip_validate_keep(int dev);

/* pinned host memory for block buffers (faster PCIe copies); plain malloc works too */
void *archon_hip_host_alloc(size_t bytes);
void archon_hip_host_free(void *p);

/* bwt[n] + base_id -> x_out[n].  All host pointers. */
int archon_hip_inverse(const uint8_t *bwt, uint32_t n, uint32_t base_id,
                       uint8_t *x_out, int dev);

/* 256-bin histogram of x[n] (host pointers). */
int archon_hip_hist256(const uint8_t *x, size_t n, uint32_t out[256], int dev);

/* LF-consistency of sa against x; returns 1 = consistent, 0 = not, <0 = error. */
int archon_hip_validate(const uint8_t *x, uint32_t n, const uint32_t *sa, int dev);

/* SA -> BWT + primary index for a suffix array the caller already holds (Archon::enWrite, archon.cpp:887-900):
 * bwt[i] = x[sa[i]] (x[0] where sa[i]==n), *base_id = the row holding n.  ARCHON_E_CORRUPT when sa holds a value
 * outside 1..n or not exactly one n. */
int archon_hip_sa_to_bwt(const uint8_t *x, uint32_t n, const uint32_t *sa, uint8_t *bwt, uint32_t *base_id, int dev);

/* The subset the reference sorts directly: a7's LMS items, Constructor::findLMS (archon.cpp:160-172;
 * a4's IT-2 rule, bwt/a4/src/archon.c:163-169, is the same step in a4's convention).  count[c] = LMS items whose first
 * key byte x[i-1] is c; items[] = the buckets' tails one after the other as a7 fills them (P[--RE[c]] = i: ascending
 * slots hold decreasing items); *n1 = their number (<= n/2: items must hold n/2 + 8 words).  The GPU sorter itself
 * sorts all N items -- this is the bucket-setup step as an operator of its own. */
int archon_hip_lms_select(const uint8_t *x, uint32_t n, uint32_t count[256], uint32_t *items, uint32_t *n1, int dev);

/* dst = src stably sorted by byte value (tool/radix_dir scatter), host pointers. */
int archon_hip_radix_scatter(const uint8_t *src, size_t n, uint8_t *dst, int dev);

/* ---- device-resident entry points (inputs already in HBM) ---------------------
 * Every pointer is a device pointer on device `dev`; `stream` is a hipStream_t
 * passed as void* (NULL = the context's own stream).  Work is enqueued on that
 * stream; the forward/inverse pipelines contain host-side decision points
 * (number of unresolved suffix groups), so these calls synchronise the stream
 * internally and the outputs are complete when they return. */
int archon_hip_forward_dev(const uint8_t *d_x, uint32_t n, uint32_t *d_sa_or_null,
                           uint8_t *d_bwt, uint32_t *d_base_id, int dev, void *stream);
int archon_hip_inverse_dev(const uint8_t *d_bwt, uint32_t n, uint32_t base_id,
                           uint8_t *d_x_out, int dev, void *stream);
int archon_hip_hist256_dev(const uint8_t *d_x, size_t n, uint32_t *d_out256, int dev, void *stream);
int archon_hip_validate_dev(const uint8_t *d_x, uint32_t n, const uint32_t *d_sa, int dev, void *stream);
/* Archon::validate for a caller that holds the forward pass's outputs on the device: like archon_hip_validate_dev, but the
 * rows' symbols are taken from d_bwt (checked to be a permutation of the block) instead of being gathered again, and
 * base_id must be the row that holds n.  This is what archon_hip_block_validate runs.  1 / 0 / <0. */
int archon_hip_validate_resident_dev(const uint8_t *d_x, uint32_t n, const uint32_t *d_sa, const uint8_t *d_bwt, uint32_t base_id,
                                     int dev, void *stream);
int archon_hip_sa_to_bwt_dev(const uint8_t *d_x, uint32_t n, const uint32_t *d_sa, uint8_t *d_bwt, uint32_t *d_base_id,
                             int dev, void *stream);
int archon_hip_radix_scatter_dev(const uint8_t *d_src, size_t n, uint8_t *d_dst, int dev, void *stream);
/* d_items: n/2 + 8 words; *n1 is a HOST pointer (the count is needed on the host to size the launches) */
int archon_hip_lms_select_dev(const uint8_t *d_x, uint32_t n, uint32_t *d_count256, uint32_t *d_items, uint32_t *n1, int dev, void *stream);

/* ---- several small blocks per call (x3's block loop, bwt/final/x3/archon.c:120-142, at its default 4 MiB block) ------------
 * One small block cannot fill the chip: a batch call deals blocks x[0..count) of n[i] bytes to `workers` host threads inside
 * the library, each on a compute context of its own, so that the blocks' copies, kernels and launch gaps overlap
 * (workers <= 0: 8 for blocks up to 4 MiB, 4 up to 16 MiB, 2 beyond; at most 8).  Every block is an independent a7
 * transform with its own primary index; the first error stops the batch and is returned.  Host pointers; the _dev
 * forms take device pointers (d_sa_or_null: NULL, or one pointer per block, each NULL or a buffer of n[i] words). */
int archon_hip_forward_batch(const uint8_t *const *x, const uint32_t *n, uint32_t count, uint8_t *const *bwt, uint32_t *base_id, int dev, int workers);
int archon_hip_inverse_batch(const uint8_t *const *bwt, const uint32_t *n, const uint32_t *base_id, uint32_t count, uint8_t *const *x_out, int dev, int workers);
int archon_hip_forward_batch_dev(const uint8_t *const *d_x, const uint32_t *n, uint32_t count, uint32_t *const *d_sa_or_null, uint8_t *const *d_bwt,
                                 uint32_t *const *d_base_id, int dev, int workers);
int archon_hip_inverse_batch_dev(const uint8_t *const *d_bwt, const uint32_t *n, const uint32_t *base_id, uint32_t count, uint8_t *const *d_x_out, int dev, int workers);

/* ---- workspace / lifetime ---------------------------------------------------- */

/* Pre-size the arena of the calling thread's context on `dev` for blocks up to n bytes (optional; the arena grows on
 * demand; the thread that reserves should be the one that transforms, or have bound itself to the same context).
 * Returns bytes reserved via *bytes_or_null. */
int archon_hip_reserve(uint32_t n, int dev, size_t *bytes_or_null);

/* Bind the calling thread to compute context `slot` (0 .. 7; threads that do not ask are dealt to 0 and 1) of `dev` -- what a pool of workers does so that the two
 * workers of one GPU never share a context whatever order they start in (host/archon_container.cpp: worker w of G GPUs
 * drives GPU w mod G on context w / G).  Without it the k-th thread to reach a device gets context k mod 2 of that
 * device.  archon_hip_context_of_thread returns the binding (and makes the default one if there is none yet). */
int archon_hip_bind_context(int dev, int slot);
int archon_hip_context_of_thread(int dev);

/* Product options, per device; read by every transform on that device when it starts.
 *   "pass_ranges"     ranges the two streaming LSB passes are cut into: 0 = one per CU (default), 1..1024.  A pass workgroup
 *                     takes a whole CU: fewer ranges than CUs leave CUs to kernels that run beside the sort (RCCL's copy
 *                     kernels while an exchange overlaps the next blocks: bench.py asks for 224 at N > 1); more, shorter
 *                     ranges shorten the tail instead, at 9 % of the passes' speed.
 *   "pass_b_buckets"  1 (default): LSB pass B deals whole second-byte buckets, one per workgroup, when the block is
 *                     balanced; 0: always by ranges (every workgroup the same work: again for a chip that is shared).
 * Every setting yields the same output. */
int archon_hip_set_option(int dev, const char *name, long value);
int archon_hip_get_option(int dev, const char *name, long *value);

/* Free the device's contexts (arenas, staging buffers, streams, events). */
int archon_hip_release(int dev);

/* ---- the MTF + zero-run + order-0 Huffman stage of the container's `-m` blocks, on the device --------
 * PARITY UNPINNED: the reference has no such stage (kvark/dark-archon README.md:2 only promises "compression schemes
 * eventually"); dark-archon_amd/host/archon_post.cpp states the format, these entry points produce the same bytes.
 * Stream of a block: u32 pieces | u32 bytes of each piece | the pieces; a piece codes 32 KiB of the BWT:
 * u32 n | 258 code lengths | bits (LSB first). */
size_t archon_hip_post_bound(uint32_t n);      /* bytes the stream of an n-byte block can take at most */
/* BWT on the device -> its stream on the device; *out_bytes = the stream's length (returned after a stream sync) */
int archon_hip_post_encode_dev(const uint8_t *d_bwt, uint32_t n, uint8_t *d_out, size_t cap, size_t *out_bytes, int dev, void *stream);
/* host block -> forward BWT -> stream, host buffers: only the packed stream and the primary index come back over the link.
 * cap may be below archon_hip_post_bound(n): a stream longer than cap makes the call fail with ARCHON_E_ARG (nothing is cut). */
int archon_hip_forward_post(const uint8_t *x, uint32_t n, uint8_t *out, size_t cap, size_t *out_bytes, uint32_t *base_id, int dev);

/* the way back (round 4): a block's stream on the device -> its BWT on the device (d_bwt holds cap bytes); *n_out = the block's length
 * (host pointer; returned after a stream sync).  ARCHON_E_CORRUPT for a malformed stream. */
int archon_hip_post_decode_dev(const uint8_t *d_in, size_t in_bytes, uint8_t *d_bwt, uint32_t cap, uint32_t *n_out, int dev, void *stream);
/* host stream + primary index -> the block (x_out holds cap bytes), host buffers: stream decoded and BWT inverted on the device, only the
 * packed stream goes up the link */
int archon_hip_inverse_post(const uint8_t *in, size_t in_bytes, uint32_t base_id, uint8_t *x_out, uint32_t cap, uint32_t *n_out, int dev);

/* ---- measurement ------------------------------------------------------------- */

/* Per-stage device times (HIP events on the stream the kernels ran on) and
 * work counters of the CALLING THREAD's most recent forward/inverse call on `dev`. */
typedef struct archon_hip_stats {
    uint32_t n;                  /* block size of the call */
    uint32_t radix_passes;       /* LSB radix passes executed by the first-stage sort */
    uint32_t doubling_rounds;    /* prefix-doubling rounds executed */
    uint64_t unresolved_initial; /* items left tied by the first stage */
    uint64_t unresolved_total;   /* sum over rounds of items entering a round */
    float ms_total;              /* whole device pipeline */
    float ms_hist;               /* hist256 / bucket setup */
    float ms_sort;               /* first-stage radix bucketing */
    float ms_doubling;           /* prefix-doubling refinement */
    float ms_bwt;                /* sa_to_bwt */
    float ms_lf_build;           /* inverse: LF table */
    float ms_lf_walk;            /* inverse: chain walk */
    uint64_t walk_chains;        /* inverse: number of sub-chains walked in parallel */
    uint32_t kernel_launches;    /* launches issued by the call */
    uint32_t radix_pass_timed;   /* first-stage radix passes bracketed by their own HIP events */
    float ms_radix_pass_sum;     /* device time inside those passes (pass kernels only) */
    float ms_local_sort;         /* streaming path: in-LDS bucket sorts (k_local_sort) */
    float ms_resolve;            /* streaming path: k_resolve_ties */
    uint32_t path;               /* 1 = streaming first stage (2 passes + local sort), 0 = 7-pass LSB */
    uint32_t tie_groups;         /* groups still tied after 5 key bytes */
    uint32_t tie_items;          /* rows flagged as tied by k_local_sort */
    float ms_pass_text;          /* streaming path: LSB pass A (k_pass_text), its own HIP events */
    float ms_pass_rec;           /* streaming path: LSB pass B (k_pass_rec), its own HIP events */
    uint32_t alphabet_bits;      /* 7-pass path: bits per symbol when the alphabet was compacted (0 = bytes) */
    uint32_t period;             /* long-repeat defence: the neighbour gap p it ran with (0 = not run) */
    uint32_t chain_items;        /* rows it settled without doubling */
    uint32_t text_rounds;        /* refinement rounds keyed on the next 4 text bytes (before any doubling round) */
    uint64_t seg_big_items;      /* sum over rounds of entries in groups too long for the in-workgroup sort (sent through the global sort) */
    uint64_t chain_pairs;        /* pairs of rows settled passage by passage (long duplicates) instead of by further doubling rounds */
    uint32_t break_rounds;       /* rounds keyed on the distance to the last defect of the period (groups that straddle defects) */
    uint32_t break_settled;      /* rows such a round settled */
    uint64_t mid_items;          /* sum over rounds of entries in groups of 1025 .. 16384 rows, each sorted by one workgroup in LDS */
    uint64_t arena_bytes;        /* device workspace the call used (bump-allocated from the context's arenas; forward calls) */
    uint32_t host_syncs;         /* times the host waited for the stream inside the call (forward calls: one for a block the streaming stage settles) */
    uint32_t reserved0;
} archon_hip_stats;

int archon_hip_get_stats(int dev, archon_hip_stats *out);

#ifdef __cplusplus
}
#endif
#endif
