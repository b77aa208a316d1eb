/*
 * a7lms_main.cpp -- runs the reference's OWN Constructor<byte>::findLMS
 * (/root/reference/bwt/a7/src/archon.cpp:160-172) on a file and dumps its placement.
 * TEST INFRASTRUCTURE ONLY; built only into oracle/_ref/ by oracle/Makefile from the reference
 * source where it lies (the translation unit includes archon.cpp through -I; nothing is copied).
 *
 * findLMS is a private step of Constructor's constructor, which runs the whole suffix sort.  To
 * see the step on its own, the harness builds a Constructor<byte> in raw storage with the same
 * member values the constructor's initialiser list gives it (archon.cpp:785-788) and calls
 * findLMS() directly -- this one translation unit is compiled with g++ -fno-access-control.
 *
 *   a7lms <in> <out>   -> out: u32 n1, u32 count[256], u32 items[n1] (the buckets' tails one after
 *                         the other, each in ascending slot order: exactly what P holds, FLAG_LMS
 *                         fillers skipped);  stdout: "n1=<n1>"
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "archon.cpp"           /* compiled with -fno-access-control: findLMS and the members are private */

struct Mirror {             /* member for member Constructor<byte> (archon.cpp:19-28) */
    const byte *data; suffix *P; t_index *R, *RE, *R2; t_index N, K, n1, d1, name;
};

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: a7lms <in> <out>\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long N = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (N <= 0) return 3;
    byte *str = (byte *)malloc((size_t)N + 1);
    if (fread(str, 1, (size_t)N, f) != (size_t)N) return 3;
    fclose(f);
    const t_index K = 256, reserve = Archon::estimateReserve((t_index)N);
    suffix *P = (suffix *)malloc(((size_t)N + reserve + 1) * sizeof(suffix));
    typedef Constructor<byte> C;
    static_assert(sizeof(Mirror) == sizeof(C), "layout of Constructor<byte> changed");
    Mirror m;
    m.data = str; m.P = P; m.R = reinterpret_cast<t_index *>(P + N); m.RE = m.R + 1;
    m.R2 = NULL;                                    /* findLMS -> buckets() -> makeBuckets() (archon.cpp:128-134) */
    m.N = (t_index)N; m.K = K; m.n1 = 0; m.d1 = 0; m.name = 0;
    alignas(C) static char raw[sizeof(C)];
    memcpy(raw, &m, sizeof m);
    C *c = reinterpret_cast<C *>(raw);
    c->findLMS();
    const t_index n1 = c->n1;
    printf("n1=%u\n", n1);
    f = fopen(argv[2], "wb");
    if (!f) return 3;
    fwrite(&n1, 4, 1, f);
    /* per-bucket counts: findLMS moves the bucket ends down as it fills (--RE[c], and RE = R + 1 aliases the starts), so
     * they are counted from what it placed: an item i sits in the bucket of its first key byte data[i-1] */
    t_index count[256];
    memset(count, 0, sizeof count);
    for (t_index i = 0; i < (t_index)N; ++i)
        if (!(P[i] & C::FLAG_LMS)) ++count[str[P[i] - 1]];
    fwrite(count, 4, 256, f);
    for (t_index i = 0; i < (t_index)N; ++i)
        if (!(P[i] & C::FLAG_LMS)) fwrite(&P[i], 4, 1, f);
    fclose(f);
    return 0;
}
