// archon_main.cpp -- the a7 command-line surface: `archon [e|d] <in> <out>`.
// Same argv, same call order, same return codes, same timing bracket as
// kvark/dark-archon bwt/a7/src/main.cpp:10-75 (-1 usage, -2 cannot open input /
// too short, -3 empty input or cannot open output); adds -4 for a GPU-side error.
// NO_VALIDATE / NO_WRITE keep their reference meaning (main.cpp:42,47).
// Extension (PARITY UNPINNED -- the reference has no such stage): `archon e -m -b<size> <in> <out>`
// adds move-to-front + zero runs + order-0 Huffman per block (archon_post.cpp); `archon d -b` decodes either kind.
// Extension: `archon e|d -b<size>[k|m] <in> <out>` reads/writes ArchonX3's multi-block
// container (bwt/final/x3/archon.c:100-110: default 4m, 8 <= size <= 256m) and spreads the blocks over the
// GPUs of the node; `d -b` takes the block size from the file header.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "archon_host.h"
#include "../../include/archon_hip.h"

static const char sUsage[] = "Usage: archon [e|d] <in> <out>\n";

int archon_container_encode(FILE *fi, FILE *fo, uint32_t bsize, int ndev, int post);
int archon_container_decode(FILE *fi, FILE *fo, int ndev, uint32_t *bsize_out);

static double now_sec()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(const int argc, const char *const argv[])
{
    FILE *fx;
    bool mode;
    printf("Archon-7 format, MI355X (gfx950) back end\n");
    // `-m` in front of `-b`: blocks additionally go through the MTF + entropy stage (archon_post.cpp; no reference
    // implementation -- parity unpinned); the decoder recognises such containers by their signature
    const bool post = argc == 6 && !strcmp(argv[2], "-m");
    const char *const *av = post ? argv + 1 : argv;
    if ((argc == 5 || post) && av[2][0] == '-' && av[2][1] == 'b' && (!strcmp(argv[1], "e") || !strcmp(argv[1], "d"))) {
        // x3's option parser (final/x3/archon.c:101-110)
        const char *sp = av[2] + 2;
        unsigned fsize = 0;
        for (; sp[0] >= '0' && sp[0] <= '9'; sp++) fsize = fsize * 10 + (unsigned)(sp[0] - '0');
        if (sp[0] == 'm') fsize <<= 20;
        else if (sp[0] == 'k') fsize <<= 10;
        if (fsize < 8 || fsize > 1u << 28) fsize = 1u << 22;
        FILE *fi = fopen(av[3], "rb");
        if (!fi) return -2;
        FILE *fo = fopen(av[4], "wb");
        if (!fo) { fclose(fi); return -3; }
        int ndev = archon_hip_device_count();
        if (const char *e = getenv("ARCHON_DEVICES")) { const int want = atoi(e); if (want > 0 && want < ndev) ndev = want; }
        const double t0c = now_sec();
        const int rc = argv[1][0] == 'e' ? archon_container_encode(fi, fo, fsize, ndev, post ? 1 : 0) : archon_container_decode(fi, fo, ndev, &fsize);
        fclose(fi);
        fclose(fo);
        if (rc) {
            printf("Error %d: %s\n", rc, rc <= -4 || rc == -1 ? archon_hip_last_error() : "bad container");
            return rc == -2 || rc == -3 ? rc : -4;
        }
        printf("Blocks of %u bytes on %d GPU(s), %.2f sec\nDone.\n", fsize, ndev, now_sec() - t0c);
        return 0;
    }
    if (argc != 4) {
        printf("%s", sUsage);
        return -1;
    }
    if (!strcmp(argv[1], "e")) mode = true;
    else if (!strcmp(argv[1], "d")) mode = false;
    else {
        printf("%s", sUsage);
        return -1;
    }
    printf("Initializing...\n");
    fx = fopen(argv[2], "rb");
    if (!fx) return -2;
    fseek(fx, 0, SEEK_END);
    long N = ftell(fx);
    if (!N) return -3;
    if (N < 0 || (unsigned long)N > ARCHON_HIP_MAX_N) return -2;
    Archon ar((t_index)N);
    const unsigned mem = ar.countMemory();
    printf("Allocated %dmb or %.1fn\n", mem >> 20, mem * 1.f / N);
    fseek(fx, 0, SEEK_SET);
    double t0;
    if (mode) {
        printf("Reading raw...\n");
        ar.enRead(fx, (t_index)N);
        fclose(fx);
        printf("Encoding SA...\n");
        t0 = now_sec();                 // wall clock: clock() would not see GPU time
        const int rc = ar.enCompute();
        t0 = now_sec() - t0;
        if (rc) {
            printf("GPU error %d: %s\n", rc, archon_hip_last_error());
            return -4;
        }
#ifndef NO_VALIDATE
        printf("Validating...");
        const bool rez = ar.validate();
        printf("%s\n", rez ? "OK" : "Fail");
#endif
#ifndef NO_WRITE
        printf("Writing BWT...\n");
        fx = fopen(argv[3], "wb");
        if (!fx) return -3;
        if (ar.enWrite(fx)) return -4;
#endif
    } else {
        N -= sizeof(int);
        if (N <= 0) return -2;
        printf("Reading BWT...\n");
        if (ar.deRead(fx, (t_index)N) < 0) {
            fclose(fx);
            return -2;
        }
        fclose(fx);
        printf("Decoding SA...\n");
        t0 = now_sec();
        const int rc = ar.deCompute();
        t0 = now_sec() - t0;
        if (rc) {
            printf("GPU error %d: %s\n", rc, archon_hip_last_error());
            return -4;
        }
        printf("Writing raw...\n");
        fx = fopen(argv[3], "wb");
        if (!fx) return -3;
        if (ar.deWrite(fx)) return -4;
    }
    fclose(fx);
    printf("SA time: %.2f sec\n", t0);
    printf("Done.\n");
    return 0;
}
