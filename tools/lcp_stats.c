/*
 * lcp_stats.c -- offline analysis tool (NOT product, NOT a test): what a block costs a
 * group-refinement sorter.  Builds P with the CPU oracle, the LCP array of neighbouring
 * keys in a7 order (key(s) = x[s-1], x[s-2], ...), and prints, for depth schedules
 * h0, 2*h0, 4*h0, ... and h0, h0+step, ...: the tied items m_h at each depth, their
 * sum (item-rounds) and the share of tied items by log2(group length).
 *
 *   gcc -O2 -o /tmp/lcp_stats tools/lcp_stats.c -Ioracle -Loracle -loracle -Wl,-rpath,$PWD/oracle
 *   /tmp/lcp_stats <file> [h0=7]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "archon_oracle.h"

static uint32_t n;
static uint32_t *lcp;   /* lcp[i] = common key bytes of rows i-1 and i (lcp[0] = 0) */

static void at_depth(uint32_t h, int verbose)
{
    /* groups = maximal row runs with lcp >= h between neighbours */
    uint64_t tied = 0, hist[33] = {0};
    uint32_t i = 0;
    while (i < n) {
        uint32_t j = i + 1;
        while (j < n && lcp[j] >= h) ++j;
        uint32_t len = j - i;
        if (len > 1) {
            tied += len;
            int b = 0;
            while ((1u << (b + 1)) <= len) ++b;
            hist[b] += len;
        }
        i = j;
    }
    printf("  h=%-8u tied %10llu (%.3f)", h, (unsigned long long)tied, (double)tied / n);
    if (verbose) {
        printf("  by log2(len):");
        for (int b = 1; b < 33; ++b)
            if (hist[b]) printf(" %d:%.1f%%", b, 100.0 * hist[b] / n);
    }
    printf("\n");
}

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    uint32_t h0 = argc > 2 ? (uint32_t)atoi(argv[2]) : 7;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    n = (uint32_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *x = malloc(n);
    if (fread(x, 1, n, f) != n) return 3;
    fclose(f);
    uint32_t *P = malloc(4ull * n), *rank = malloc(4ull * (n + 1));
    lcp = malloc(4ull * n);
    if (oracle_sa(x, n, P)) return 4;
    for (uint32_t i = 0; i < n; ++i) rank[P[i]] = i;
    /* Kasai over items s = n .. 1: key(s-1) is key(s) without its first byte */
    uint32_t l = 0;
    uint64_t sum = 0;
    uint32_t mx = 0;
    for (uint32_t s = n; s >= 1; --s) {
        uint32_t r = rank[s];
        if (r == 0) { lcp[0] = 0; l = 0; continue; }
        uint32_t t = P[r - 1];
        while (l < s && l < t && x[s - 1 - l] == x[t - 1 - l]) ++l;
        lcp[r] = l;
        sum += l;
        if (l > mx) mx = l;
        if (l) --l;
    }
    printf("n=%u mean lcp %.1f max lcp %u\n", n, (double)sum / n, mx);
    printf("doubling from h0=%u:\n", h0);
    uint64_t total = 0;
    for (uint64_t h = h0; h <= 2ull * mx + 1 && h < (1ull << 31); h *= 2) at_depth((uint32_t)h, 1);
    printf("by depth (text keys):\n");
    for (uint32_t h = 1; h <= 64; h += (h < 16 ? 1 : 8)) at_depth(h, 0);
    (void)total;
    return 0;
}
