/*
 * a7ref_main.cpp -- harness around the UNMODIFIED reference class Archon
 * (/root/reference/bwt/a7/src/archon.h, archon.cpp), built only into
 * oracle/_ref/ by oracle/Makefile.  TEST INFRASTRUCTURE ONLY.
 *
 * It exists because the reference CLI (bwt/a7/src/main.cpp:10-75) keeps P
 * private and writes only BWT||baseId; the parity tests also want P.  Call
 * order, timing bracket (clock() around *Compute only) and file layout follow
 * main.cpp:30-69 exactly.
 *
 *   a7ref e <in> <out.bwt> [<out.sa>]   -> stdout: "validate=<0|1> sa_time=<sec>"
 *   a7ref d <in.bwt> <out.raw>          -> stdout: "sa_time=<sec>"
 */
#include <stdio.h>
#include <string.h>
#include <time.h>

#define class struct            /* members default to public: read-only peek at Archon::P */
#include "archon.h"
#undef class

int main(int argc, char **argv)
{
    if (argc < 4 || (strcmp(argv[1], "e") && strcmp(argv[1], "d"))) {
        fprintf(stderr, "usage: a7ref e|d <in> <out> [<out.sa>]\n");
        return 1;
    }
    const bool enc = !strcmp(argv[1], "e");
    FILE *fx = fopen(argv[2], "rb");
    if (!fx) return 2;
    fseek(fx, 0, SEEK_END);
    long N = ftell(fx);
    fseek(fx, 0, SEEK_SET);
    if (N <= 0) return 3;
    Archon ar((t_index)N);
    clock_t t0;
    if (enc) {
        ar.enRead(fx, (t_index)N);
        fclose(fx);
        t0 = clock();
        ar.enCompute();
        t0 = clock() - t0;
        const bool ok = ar.validate();
        printf("validate=%d sa_time=%.6f\n", ok ? 1 : 0, (double)t0 / CLOCKS_PER_SEC);
        fflush(stdout);
        fx = fopen(argv[3], "wb");
        if (!fx) return 3;
        ar.enWrite(fx);
        fclose(fx);
        if (argc > 4) {
            fx = fopen(argv[4], "wb");
            if (!fx) return 3;
            fwrite(ar.P, sizeof(suffix), (size_t)N, fx);
            fclose(fx);
        }
    } else {
        N -= (long)sizeof(int);
        if (N <= 0) return 2;
        ar.deRead(fx, (t_index)N);
        fclose(fx);
        t0 = clock();
        ar.deCompute();
        t0 = clock() - t0;
        clock_t t1 = clock();
        fx = fopen(argv[3], "wb");
        if (!fx) return 3;
        ar.deWrite(fx);
        fclose(fx);
        t1 = clock() - t1;
        printf("sa_time=%.6f walk_time=%.6f\n", (double)t0 / CLOCKS_PER_SEC, (double)t1 / CLOCKS_PER_SEC);
    }
    return 0;
}
