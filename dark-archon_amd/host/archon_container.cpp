// archon_container.cpp -- multi-block files: inputs larger than one block stream
// through the tool as a sequence of independent BWT blocks, block b on GPU b mod G.
//
// Layout = ArchonX3's container (kvark/dark-archon bwt/final/x3/archon.c:17,100-110,120-125,133-142):
//   ushort signature 'RA' (bytes 0x41 0x52), uint32 block size, then per block: n transformed bytes followed by
//   the 4-byte primary index; every block but the last has n == block size, and a shorter (possibly empty)
//   block ends the file -- exactly what x3's reader loop `while(n==fsize)` expects.
// The block payload is the a7 transform (BWT || baseId in a7 order), not x3's own sort order.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#include "archon_host.h"
#include "../../include/archon_hip.h"

static const unsigned short kSig = 0x5241;   // 'RA' as x3 writes it

struct Slot {
    std::vector<byte> in, out;
    t_index base = 0;
    int rc = 0;
};

int archon_container_encode(FILE *fi, FILE *fo, uint32_t bsize, int ndev)
{
    if (ndev < 1) return -4;
    fwrite(&kSig, 2, 1, fo);
    fwrite(&bsize, 4, 1, fo);
    std::vector<Slot> slot(ndev);
    bool last_seen = false;
    while (!last_seen) {
        int used = 0;
        for (; used < ndev && !last_seen; ++used) {
            slot[used].in.resize(bsize);
            const size_t n = fread(slot[used].in.data(), 1, bsize, fi);
            slot[used].in.resize(n);
            if (n < bsize) last_seen = true;
        }
        std::vector<std::thread> th;
        for (int k = 0; k < used; ++k)
            th.emplace_back([&, k]() {
                Slot &s = slot[k];
                s.rc = 0;
                s.base = 0;
                s.out.resize(s.in.size());
                if (!s.in.empty())
                    s.rc = archon_hip_forward(s.in.data(), (uint32_t)s.in.size(), NULL, s.out.data(), &s.base, k);
            });
        for (auto &t : th) t.join();
        for (int k = 0; k < used; ++k) {
            if (slot[k].rc) return slot[k].rc;
            if (!slot[k].out.empty()) fwrite(slot[k].out.data(), 1, slot[k].out.size(), fo);
            fwrite(&slot[k].base, 4, 1, fo);
        }
    }
    return 0;
}

int archon_container_decode(FILE *fi, FILE *fo, int ndev)
{
    if (ndev < 1) return -4;
    unsigned short sig = 0;
    uint32_t bsize = 0;
    if (fread(&sig, 2, 1, fi) != 1 || sig != kSig) return -3;
    if (fread(&bsize, 4, 1, fi) != 1 || bsize < 8 || bsize > (1u << 28)) return -3;
    std::vector<Slot> slot(ndev);
    bool last_seen = false;
    while (!last_seen) {
        int used = 0;
        for (; used < ndev && !last_seen; ++used) {
            slot[used].in.resize((size_t)bsize + 4);
            const size_t got = fread(slot[used].in.data(), 1, (size_t)bsize + 4, fi);
            if (got < 4) return -2;
            const size_t n = got - 4;
            memcpy(&slot[used].base, slot[used].in.data() + n, 4);
            slot[used].in.resize(n);
            if (n < bsize) last_seen = true;
        }
        std::vector<std::thread> th;
        for (int k = 0; k < used; ++k)
            th.emplace_back([&, k]() {
                Slot &s = slot[k];
                s.rc = 0;
                s.out.resize(s.in.size());
                if (!s.in.empty())
                    s.rc = archon_hip_inverse(s.in.data(), (uint32_t)s.in.size(), s.base, s.out.data(), k);
            });
        for (auto &t : th) t.join();
        for (int k = 0; k < used; ++k) {
            if (slot[k].rc) return slot[k].rc;
            if (!slot[k].out.empty()) fwrite(slot[k].out.data(), 1, slot[k].out.size(), fo);
        }
    }
    return 0;
}
