// archon_main.cpp -- the a7 command-line surface: `archon [e|d] <in> <out>`.
// Same argv, same call order, same return codes, same timing bracket as
// kvark/dark-archon bwt/a7/src/main.cpp:10-75 (-1 usage, -2 cannot open input /
// too short, -3 empty input or cannot open output); adds -4 for a GPU-side error.
// NO_VALIDATE / NO_WRITE keep their reference meaning (main.cpp:42,47).
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "archon_host.h"
#include "../../include/archon_hip.h"

static const char sUsage[] = "Usage: archon [e|d] <in> <out>\n";

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
