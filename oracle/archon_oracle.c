/*
 * archon_oracle.c -- CPU oracle for the Archon a7 BWT hot path (see archon_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py; never by the product path.
 *
 * Parity status: PINNED (tests/test_oracle.py): known answers of SURVEY.md 8(a0),
 * exhaustive small alphabets against the brute-force definition below, and the
 * reference a7 binaries built from /root/reference by oracle/Makefile.
 *
 * How the restatement maps onto bwt/a7/src/archon.cpp
 * ---------------------------------------------------
 * a7 sorts items s=1..N by the reversed prefix x[s-1],x[s-2],...,x[0],INF and
 * runs SA-IS "mirrored": its left-to-right sweep fills bucket starts and extends
 * an item s to s+1 (archon.cpp:392-413, 522-540), its right-to-left sweep fills
 * bucket ends (415-433, 541-561).  Write z[j] = 255 - x[N-1-j]; item s is the
 * suffix j = N-s of z, and "ascending a7 order with INF largest" is exactly
 * "descending textbook suffix order of z with the end marker smallest".  The
 * core below is therefore the textbook orientation of the same algorithm --
 * type classification + LMS seeding (findLMS 160-172), the two induction
 * sweeps (inducePre 387-434 / inducePost 518-562), LMS-substring naming
 * (computeTargetValues 184-205), recursion on the reduced string
 * (solve 668-689) and the final placement + induction (derive 720-778) --
 * reading z on the fly, followed by P[i] = N - SA_z[N-1-i].
 */
#include "archon_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* A2: makeBuckets (archon.cpp:118-126)                                */

void oracle_hist256(const uint8_t *x, size_t n, uint32_t counts[256], uint32_t starts[257])
{
    uint32_t c[256];
    memset(c, 0, sizeof c);
    for (size_t i = 0; i < n; ++i)
        ++c[x[i]];
    uint32_t sum = 0;
    for (int k = 0; k < 256; ++k) {
        if (starts) starts[k] = sum;
        sum += c[k];
    }
    if (starts) starts[256] = sum;
    if (counts) memcpy(counts, c, sizeof c);
}

/* ------------------------------------------------------------------ */
/* The definition: sufCompare (archon.cpp:83-88)                       */

static const uint8_t *g_cmp_x; /* qsort has no context argument */

/* <0 when key(a) < key(b).  Keys are distinct for a != b. */
static int key_compare(const uint8_t *x, uint32_t a, uint32_t b)
{
    uint32_t d = 1;
    while (a >= d && b >= d && x[a - d] == x[b - d])
        ++d;
    if (a < d && b < d) return 0;       /* only when a == b */
    if (a < d) return 1;                /* a ran out: INF, sorts above */
    if (b < d) return -1;
    return (int)x[a - d] - (int)x[b - d];
}

static int qsort_cmp(const void *pa, const void *pb)
{
    return key_compare(g_cmp_x, *(const uint32_t *)pa, *(const uint32_t *)pb);
}

int oracle_sa_brute(const uint8_t *x, uint32_t n, uint32_t *P)
{
    for (uint32_t i = 0; i < n; ++i)    /* directSort seeds P[i]=i+1, archon.cpp:40-43 */
        P[i] = i + 1;
    g_cmp_x = x;
    qsort(P, n, sizeof *P, qsort_cmp);
    return 0;
}

int oracle_check_sorted(const uint8_t *x, uint32_t n, const uint32_t *P)
{
    uint8_t *seen = (uint8_t *)calloc((size_t)n + 1, 1);
    if (!seen) return 0;
    int ok = 1;
    for (uint32_t i = 0; i < n && ok; ++i) {
        uint32_t s = P[i];
        if (s < 1 || s > n || seen[s]) ok = 0; else seen[s] = 1;
    }
    for (uint32_t i = 1; i < n && ok; ++i)
        if (key_compare(x, P[i - 1], P[i]) >= 0) ok = 0;
    free(seen);
    return ok;
}

/* ------------------------------------------------------------------ */
/* SA-IS core (textbook orientation of archon.cpp:118-778, see header)  */

typedef struct {
    const uint8_t *x;   /* level 0: the block; symbol j is 255 - x[n-1-j] */
    const int32_t *t;   /* level >= 1: the reduced string (solve<Q>, archon.cpp:668-689) */
    int32_t n;
} sstr;

static inline int32_t CH(const sstr *s, int32_t j)
{
    return s->x ? 255 - (int32_t)s->x[s->n - 1 - j] : s->t[j];
}

#define TGET(tp, j)   (((tp)[(j) >> 3] >> ((j) & 7)) & 1)       /* 1 = S-type, 0 = L-type */
#define TSET(tp, j)   ((tp)[(j) >> 3] |= (uint8_t)(1u << ((j) & 7)))
#define IS_LMS(tp, j) ((j) > 0 && TGET(tp, j) && !TGET(tp, (j) - 1))

/* bucket starts (end=0) or ends (end=1) from counts; a7 keeps both as R / RE=R+1
 * (archon.cpp:21-23,128-134) */
static void bucket_bounds(const int32_t *C, int32_t *B, int32_t K, int end)
{
    int32_t sum = 0;
    for (int32_t k = 0; k < K; ++k) {
        sum += C[k];
        B[k] = end ? sum : sum - C[k];
    }
}

/* The two induction sweeps.  a7: inducePre 387-434 (LMS-substring sort) and
 * inducePost 518-562 (final) share this shape; the sweep that starts from the
 * item next to the end marker is archon.cpp:416-417 / 543-544. */
static void induce(const sstr *s, int32_t *SA, int32_t n, int32_t K,
                   const int32_t *C, int32_t *B, const uint8_t *tp)
{
    bucket_bounds(C, B, K, 0);
    SA[B[CH(s, n - 1)]++] = n - 1;              /* induced by the end marker */
    for (int32_t i = 0; i < n; ++i) {
        int32_t j = SA[i];
        if (j > 0 && !TGET(tp, j - 1))
            SA[B[CH(s, j - 1)]++] = j - 1;
    }
    bucket_bounds(C, B, K, 1);
    for (int32_t i = n - 1; i >= 0; --i) {
        int32_t j = SA[i];
        if (j > 0 && TGET(tp, j - 1))
            SA[--B[CH(s, j - 1)]] = j - 1;
    }
}

static int sais_core(const sstr *s, int32_t *SA, int32_t n, int32_t K)
{
    int rc = -1;
    uint8_t *tp = (uint8_t *)calloc((size_t)n / 8 + 1, 1);
    int32_t *C = (int32_t *)calloc((size_t)K, sizeof *C);
    int32_t *B = (int32_t *)malloc((size_t)K * sizeof *B);
    int32_t *lms = NULL, *s1 = NULL, *SA1 = NULL, *nm = NULL;
    if (!tp || !C || !B) goto done;

    /* makeBuckets, archon.cpp:118-126 */
    for (int32_t j = 0; j < n; ++j)
        ++C[CH(s, j)];

    /* type classification; position n-1 is L (the end marker is smallest) */
    for (int32_t j = n - 2; j >= 0; --j) {
        int32_t c0 = CH(s, j), c1 = CH(s, j + 1);
        if (c0 < c1 || (c0 == c1 && TGET(tp, j + 1)))
            TSET(tp, j);
    }

    /* findLMS, archon.cpp:160-172: seed LMS items at their bucket ends */
    int32_t n1 = 0;
    for (int32_t i = 0; i < n; ++i) SA[i] = -1;
    bucket_bounds(C, B, K, 1);
    for (int32_t j = 1; j < n; ++j)
        if (IS_LMS(tp, j)) {
            SA[--B[CH(s, j)]] = j;
            ++n1;
        }

    /* inducePre, archon.cpp:387-434: sorts the LMS substrings */
    induce(s, SA, n, K, C, B, tp);

    if (n1 > 0) {
        /* packTargetIndices, archon.cpp:174-182 */
        int32_t m = 0;
        for (int32_t i = 0; i < n; ++i) {
            int32_t j = SA[i];
            if (IS_LMS(tp, j)) SA[m++] = j;
        }
        /* computeTargetValues, archon.cpp:184-205: equal substrings get equal names */
        nm = (int32_t *)malloc(((size_t)n / 2 + 1) * sizeof *nm);
        lms = (int32_t *)malloc((size_t)n1 * sizeof *lms);
        s1 = (int32_t *)malloc((size_t)n1 * sizeof *s1);
        SA1 = (int32_t *)malloc((size_t)n1 * sizeof *SA1);
        if (!nm || !lms || !s1 || !SA1) goto done;
        int32_t names = 0, prev = -1;
        for (int32_t i = 0; i < n1; ++i) {
            int32_t cur = SA[i];
            int diff = (prev < 0);
            for (int32_t d = 0; !diff; ++d) {
                int32_t pc = prev + d, cc = cur + d;
                if (pc == n || cc == n) { diff = 1; break; }
                if (CH(s, pc) != CH(s, cc) || TGET(tp, pc) != TGET(tp, cc)) { diff = 1; break; }
                if (d > 0) {
                    int a = IS_LMS(tp, pc), b = IS_LMS(tp, cc);
                    if (a && b) break;          /* same substring */
                    if (a || b) { diff = 1; break; }
                }
            }
            if (diff) { ++names; prev = cur; }
            nm[cur >> 1] = names - 1;
        }
        /* packTargetValues, archon.cpp:651-666: reduced string in text order */
        m = 0;
        for (int32_t j = 1; j < n; ++j)
            if (IS_LMS(tp, j)) {
                lms[m] = j;
                s1[m] = nm[j >> 1];
                ++m;
            }
        /* solve, archon.cpp:668-689 */
        if (names < n1) {
            sstr sub = { NULL, s1, n1 };
            if (sais_core(&sub, SA1, n1, names) < 0) goto done;
        } else {
            for (int32_t k = 0; k < n1; ++k) SA1[s1[k]] = k;
        }
        /* derive, archon.cpp:720-771: sorted LMS items back to their bucket ends */
        for (int32_t i = 0; i < n; ++i) SA[i] = -1;
        bucket_bounds(C, B, K, 1);
        for (int32_t i = n1 - 1; i >= 0; --i) {
            int32_t j = lms[SA1[i]];
            SA[--B[CH(s, j)]] = j;
        }
        /* inducePost, archon.cpp:518-562 */
        induce(s, SA, n, K, C, B, tp);
    }
    rc = 0;
done:
    free(tp); free(C); free(B); free(lms); free(s1); free(SA1); free(nm);
    return rc;
}

int oracle_sa(const uint8_t *x, uint32_t n, uint32_t *P)
{
    if (n == 0 || n >= 0x80000000u) return -2;
    int32_t *SA = (int32_t *)P;
    sstr top = { x, NULL, (int32_t)n };
    int rc = sais_core(&top, SA, (int32_t)n, 256);
    if (rc < 0) return rc;
    /* P[i] = N - SA_z[N-1-i], in place */
    for (uint32_t i = 0, j = n - 1; i <= j; ++i, --j) {
        uint32_t a = n - (uint32_t)SA[j], b = n - (uint32_t)SA[i];
        P[i] = a;
        P[j] = b;
        if (j == 0) break;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* A7: enWrite (archon.cpp:887-900)                                     */

void oracle_sa_to_bwt(const uint8_t *x, uint32_t n, const uint32_t *P,
                      uint8_t *bwt, uint32_t *base_id)
{
    uint32_t base = n;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t pos = P[i];
        if (pos == n) {
            base = i;
            pos = 0;
        }
        bwt[i] = x[pos];
    }
    if (base_id) *base_id = base;
}

/* A10: validate (archon.cpp:862-874) */
int oracle_validate(const uint8_t *x, uint32_t n, const uint32_t *P)
{
    uint32_t R[256];
    memset(R, 0, sizeof R);
    for (uint32_t i = n; i--;) {
        if (P[i] < 1 || P[i] > n) return 0;
        R[x[P[i] - 1]] = i;
    }
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t s = P[i];
        if (s != n) {
            uint32_t r = R[x[s]]++;
            if (r >= n || P[r] != s + 1) return 0;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* A8: deCompute + roll (archon.cpp:905-908, 917-935)                   */

void oracle_lf_build(const uint8_t *bwt, uint32_t n, uint32_t base_id, uint32_t *T)
{
    uint32_t R[256], k = n;
    memset(R, 0, sizeof R);
    for (uint32_t i = 0; i < n; ++i)
        ++R[bwt[i]];
    for (int c = 256; c--;)
        R[c] = (k -= R[c]);
    for (uint32_t i = 0; i < base_id; ++i) T[i] = R[bwt[i]]++;
    for (uint32_t i = base_id + 1; i < n; ++i) T[i] = R[bwt[i]]++;
    T[base_id] = R[bwt[base_id]]++;
}

/* A9: deWrite (archon.cpp:937-943) */
int oracle_lf_walk(const uint8_t *bwt, uint32_t n, uint32_t base_id,
                   const uint32_t *T, uint8_t *out)
{
    uint32_t k = base_id;
    for (uint32_t i = 0; i < n; ++i, k = T[k])
        out[i] = bwt[k];
    return k == base_id;
}

int oracle_forward(const uint8_t *x, uint32_t n, uint32_t *P, uint8_t *bwt, uint32_t *base_id)
{
    uint32_t *own = NULL;
    if (!P) {
        own = (uint32_t *)malloc((size_t)n * sizeof *own);
        if (!own) return -1;
        P = own;
    }
    int rc = oracle_sa(x, n, P);
    if (rc == 0) oracle_sa_to_bwt(x, n, P, bwt, base_id);
    free(own);
    return rc;
}

int oracle_inverse(const uint8_t *bwt, uint32_t n, uint32_t base_id, uint8_t *out)
{
    if (n == 0 || base_id >= n) return -2;
    uint32_t *T = (uint32_t *)malloc((size_t)n * sizeof *T);
    if (!T) return -1;
    oracle_lf_build(bwt, n, base_id, T);
    int closed = oracle_lf_walk(bwt, n, base_id, T, out);
    free(T);
    return closed ? 0 : -3;
}

/* tool/radix_dir/radix.c:29-36 (bucket starts) + 40-44 (the "X++" scatter) */
void oracle_radix_scatter(const uint8_t *src, size_t n, uint8_t *dst)
{
    uint32_t R[257];
    oracle_hist256(src, n, NULL, R);
    for (size_t i = 0; i < n; ++i)
        dst[R[src[i]]++] = src[i];
}

/* A3: the subset a7 sorts directly -- Constructor::findLMS, bwt/a7/src/archon.cpp:160-172 (a4's IT-2 rule,
 * bwt/a4/src/archon.c:163-169, and a6's IT-1 filter, bwt/a6/src/bwt.c:391-399, pick their subsets the same way in
 * their own conventions).  The scan alternates between a falling phase (skip while x[i-1] >= x[i]) and a rising
 * phase (skip while x[i-1] <= x[i]); the item at which a falling phase ends is an LMS item and is placed at the END of
 * the bucket of its first key byte x[i-1], filling that bucket's tail downwards (P[--RE[x[i-1]]] = i).
 * Output: count[c] = LMS items in bucket c; items[] = the buckets' tails one after the other, each in ascending slot
 * order (= decreasing item).  Returns n1, the number of LMS items. */
uint32_t oracle_lms_select(const uint8_t *x, uint32_t n, uint32_t count[256], uint32_t *items)
{
    uint32_t n1 = 0;
    memset(count, 0, 256 * sizeof(uint32_t));
    for (int pass = 0; pass < 2; ++pass) {          /* pass 0 counts, pass 1 places */
        uint32_t end[256], acc = 0;
        for (int c = 0; c < 256; ++c) { acc += count[c]; end[c] = acc; }
        uint32_t i = 0;
        for (;;) {
            int done = 0;
            do { if (++i >= n) { done = 1; break; } } while (x[i - 1] >= x[i]);      /* archon.cpp:164-167 */
            if (done) break;
            if (pass == 0) { ++count[x[i - 1]]; ++n1; }                              /* archon.cpp:168 ++n1 */
            else items[--end[x[i - 1]]] = i;                                         /* archon.cpp:169 */
            while (++i < n && x[i - 1] <= x[i]) {}                                   /* archon.cpp:170 */
        }
    }
    return n1;
}

double oracle_clock_seconds(void)
{
    return (double)clock() / CLOCKS_PER_SEC;
}
