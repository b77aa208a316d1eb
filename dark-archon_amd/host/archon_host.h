// archon_host.h -- class Archon with the reference's own interface
// (kvark/dark-archon bwt/a7/src/archon.h:8-29), backed by libarchon_hip (MI355X).
// Same names, argument meaning and return conventions, so the reference's
// main.cpp compiles against this header unchanged.
#pragma once
#include <stdint.h>
#include <stdio.h>

typedef unsigned int  suffix;
typedef unsigned int  t_index;
typedef unsigned char byte;

class Archon {
    const t_index Nmax, Nreserve;
    suffix *const P;      // (Nmax + Nreserve) words: SA after enCompute; decoded bytes after deCompute
    byte *const str;      // Nmax + 1 bytes: the block (encode) / the BWT (decode)
    t_index N, baseId;
    int dev;
    bool pinned;
    int last_rc;
    struct archon_hip_block *blk;   // the object's resident block on the device (x, SA, BWT between enCompute, validate and enWrite)
    bool resident;                  // blk holds what str / P hold: set by enCompute, cleared by every read into str

public:
    static t_index estimateReserve(const t_index);
    Archon(const t_index N);
    ~Archon();
    unsigned countMemory() const;
    bool validate();
    // encoding
    int enRead(FILE *const fx, t_index ns);
    int enCompute();
    int enWrite(FILE *const fx);
    // decoding
    int deRead(FILE *const fx, t_index ns);
    int deCompute();
    int deWrite(FILE *const fx);
    // additions (the reference keeps these private)
    const suffix *sa() const { return P; }
    t_index baseIndex() const { return baseId; }
    t_index length() const { return N; }
    void setDevice(int d);
    int lastError() const { return last_rc; }
};
