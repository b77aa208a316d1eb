// archon_container.cpp -- multi-block files: inputs larger than one block stream
// through the tool as a sequence of independent BWT blocks, block b on GPU b mod G.
//
// Layout = ArchonX3's container (kvark/dark-archon bwt/final/x3/archon.c:17,100-110,120-125,133-142):
//   ushort signature 'RA' (bytes 0x41 0x52), uint32 block size, then per block: n transformed bytes followed by
//   the 4-byte primary index; every block but the last has n == block size, and a shorter (possibly empty)
//   block ends the file -- exactly what x3's reader loop `while(n==fsize)` expects.
// The block payload is the a7 transform (BWT || baseId in a7 order), not x3's own sort order.
//
// x3 reads, transforms and writes one block after the other (archon.c:120-142).  Here the three steps are a
// pipeline over a ring of pinned host slots: a reader thread fills slot b mod S with block b, two worker threads per
// GPU (block b -> GPU b mod G, the sharding rule of dark-archon_amd/archon_shard.py) run the transform
// (H2D + kernels + D2H through the C ABI), a writer thread emits the blocks in order -- so the fread of block k+1
// and the fwrite of block k-1 overlap the GPU's work on block k.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "archon_host.h"
#include "../../include/archon_hip.h"

static const unsigned short kSig = 0x5241;       // 'RA' as x3 writes it
static const unsigned short kSigPost = 0x4E52;   // 'RN': blocks carry the MTF + entropy stage (archon_post.cpp, parity unpinned); the header names the piece size
static const unsigned short kSigPostV1 = 0x4D52; // 'RM': the first revision of that format (pieces of 4 MiB, size not in the header) -- recognised, refused

extern "C" {
size_t archon_post_bound(size_t n);
size_t archon_post_encode(const uint8_t *bwt, size_t n, uint8_t *out);
int archon_post_decode(const uint8_t *in, size_t in_bytes, uint8_t *bwt, size_t n);
}

namespace {

enum SlotState { kFree, kFilled, kDone };

struct Slot {
    byte *in = nullptr, *out = nullptr;
    bool pinned = false;
    size_t n = 0;            // payload bytes of the block in the slot
    size_t packed = 0;       // post stage: bytes of the packed stream in `out`
    t_index base = 0;
    SlotState state = kFree;
    bool last = false;
};

struct Pipe {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slot> slot;
    long nblocks = -1;       // known once the reader has seen the short block
    int rc = 0;              // first error; everybody stops

    bool alloc(int nslots, size_t bytes, size_t out_bytes = 0)
    {
        if (!out_bytes) out_bytes = bytes;
        slot.resize(nslots);
        for (Slot &s : slot) {
            s.in = static_cast<byte *>(archon_hip_host_alloc(bytes));
            s.out = static_cast<byte *>(archon_hip_host_alloc(out_bytes));
            s.pinned = s.in && s.out;
            if (!s.pinned) {
                fprintf(stderr, "archon: %zu + %zu bytes of pinned host memory not available: slot in pageable memory (copies will not overlap)\n", bytes, out_bytes);
                if (s.in) archon_hip_host_free(s.in);
                if (s.out) archon_hip_host_free(s.out);
                s.in = static_cast<byte *>(malloc(bytes));
                s.out = static_cast<byte *>(malloc(out_bytes));
            }
            if (!s.in || !s.out) return false;
        }
        return true;
    }
    ~Pipe()
    {
        for (Slot &s : slot) {
            if (s.pinned) { archon_hip_host_free(s.in); archon_hip_host_free(s.out); }
            else { free(s.in); free(s.out); }
        }
    }
    // waits until slot (b mod S) reaches `want`; false when the pipeline has failed or block b does not exist
    bool wait(long b, SlotState want, Slot **out)
    {
        std::unique_lock<std::mutex> lk(mu);
        Slot &s = slot[(size_t)(b % (long)slot.size())];
        cv.wait(lk, [&] { return rc != 0 || (nblocks >= 0 && b >= nblocks) || s.state == want; });
        if (rc != 0 || (nblocks >= 0 && b >= nblocks)) return false;
        *out = &s;
        return true;
    }
    void set(Slot *s, SlotState st)
    {
        { std::lock_guard<std::mutex> lk(mu); s->state = st; }
        cv.notify_all();
    }
    void fail(int code)
    {
        { std::lock_guard<std::mutex> lk(mu); if (!rc) rc = code; }
        cv.notify_all();
    }
    void total(long n)
    {
        { std::lock_guard<std::mutex> lk(mu); nblocks = n; }
        cv.notify_all();
    }
};

// Several workers per GPU, each bound to a compute context of its own (archon_hip_bind_context): worker w (blocks w, w + WG,
// w + 2WG, ... on GPU w mod G -- still block b on GPU b mod G) moves its block over PCIe while its twins' kernels run.
// Small blocks (x3's default is 4 MiB, final/x3/archon.c:100) cannot fill the chip one or two at a time -- a block is
// thirty launches of a few microseconds -- so they get more workers, each on a context of its own (up to the library's eight).
static int workers_per_gpu(uint32_t bsize) { return bsize <= (4u << 20) ? 8 : bsize <= (16u << 20) ? 4 : 2; }

// worker `first` of `step`: blocks first, first + step, ... on GPU `dev`
template <class Work>
void worker_loop(Pipe &p, int first, int step, int dev, Work work)
{
    for (long b = first;; b += step) {
        Slot *s;
        if (!p.wait(b, kFilled, &s)) return;
        const int rc = s->n ? work(*s, dev) : 0;
        if (rc) { p.fail(rc); return; }
        p.set(s, kDone);
    }
}

// ---- post stage (no reference implementation): a block's BWT is cut into pieces of 32 KiB that are
// coded independently (MTF restarts per piece, archon_post.cpp).  Encoding runs on the GPU behind the transform
// (archon_hip_forward_post: one workgroup per piece, only the packed stream comes back over the link); decoding is a pool
// of host threads.
// Packed block: u32 pieces | u32 packed bytes of each piece | the pieces.
constexpr size_t kPieceEnc = 32u << 10;       // what the encoder (csrc/post.hiph) cuts; the decoder takes the size from the file header
// Host slot of a packed block: an order-0 code over the MTF ranks of n bytes stays under 9/8 n + tables; the worst case the
// format allows (archon_hip_post_bound: 20 bits per symbol) would pin 2.5 n per slot -- 30 GB on an 8-GPU node at 256 MiB
// blocks -- for streams that cannot occur.  A stream that does not fit is an error of the call, not a silent truncation.
static size_t post_slot_bytes(size_t n) { return n + n / 4 + (n / kPieceEnc + 2) * 1100 + 4096; }

// returns the block length n, or -1 on a malformed stream; bwt must hold `cap` bytes
long post_unpack(const byte *in, size_t in_bytes, byte *bwt, size_t cap, size_t kPiece)
{
    if (in_bytes < 4) return -1;
    uint32_t np;
    memcpy(&np, in, 4);
    if ((size_t)np > cap / kPiece + 1 || in_bytes < 4 + 4 * (size_t)np) return -1;
    std::vector<size_t> off(np + 1), len(np);
    off[0] = 4 + 4 * (size_t)np;
    size_t n = 0;
    for (uint32_t k = 0; k < np; ++k) {
        uint32_t sz;
        memcpy(&sz, in + 4 + 4 * k, 4);
        off[k + 1] = off[k] + sz;
        if (off[k + 1] > in_bytes || sz < 4) return -1;
        uint32_t plen;
        memcpy(&plen, in + off[k], 4);
        if (plen > kPiece || (k + 1 < np && plen != kPiece)) return -1;
        len[k] = plen;
        n += plen;
    }
    if (n > cap) return -1;
    std::vector<int> rc(np, 0);
    std::vector<std::thread> th;
    const unsigned nth = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
    for (unsigned t = 0; t < nth; ++t)
        th.emplace_back([&, t] {
            for (size_t k = t; k < np; k += nth)
                rc[k] = archon_post_decode(in + off[k], off[k + 1] - off[k], bwt + k * kPiece, len[k]);
        });
    for (auto &t : th) t.join();
    for (int r : rc)
        if (r) return -1;
    return (long)n;
}

}  // namespace

int archon_container_encode(FILE *fi, FILE *fo, uint32_t bsize, int ndev, int post)
{
    if (ndev < 1) return -4;
    if (fwrite(post ? &kSigPost : &kSig, 2, 1, fo) != 1 || fwrite(&bsize, 4, 1, fo) != 1) return -3;     // (a full disk shows here first)
    const uint32_t piece = (uint32_t)kPieceEnc;
    if (post && fwrite(&piece, 4, 1, fo) != 1) return -3;
    Pipe p;
    const size_t out_cap = post ? post_slot_bytes(bsize) : (size_t)bsize + 4;
    const int kWorkersPerGpu = workers_per_gpu(bsize);
    if (!p.alloc((kWorkersPerGpu + 2) * ndev, (size_t)bsize + 4, out_cap)) return ARCHON_E_NOMEM;

    std::thread reader([&] {
        for (long b = 0;; ++b) {
            Slot *s;
            if (!p.wait(b, kFree, &s)) return;
            s->n = fread(s->in, 1, bsize, fi);
            s->last = s->n < bsize;
            const bool last = s->last;
            p.set(s, kFilled);
            if (last) { p.total(b + 1); return; }
        }
    });
    std::vector<std::thread> workers;
    for (int w = 0; w < kWorkersPerGpu * ndev; ++w)
        workers.emplace_back([&, w] {
            archon_hip_bind_context(w % ndev, w / ndev);        // the two workers of a GPU on its two contexts, whatever order they start in
            worker_loop(p, w, kWorkersPerGpu * ndev, w % ndev, [post, out_cap](Slot &s, int dev) {
                if (post) return archon_hip_forward_post(s.in, (uint32_t)s.n, s.out, out_cap, &s.packed, &s.base, dev);
                return archon_hip_forward(s.in, (uint32_t)s.n, NULL, s.out, &s.base, dev);
            });
        });
    std::thread writer([&] {
        for (long b = 0;; ++b) {
            Slot *s;
            if (!p.wait(b, kDone, &s)) return;
            if (post) {                                  // u32 packed bytes | packed block | index
                static const byte kEmpty[4] = {0, 0, 0, 0};             // an empty last block: zero pieces
                const byte *pk = s->n ? s->out : kEmpty;
                const uint32_t sz = s->n ? (uint32_t)s->packed : 4u;
                if (fwrite(&sz, 4, 1, fo) != 1 || fwrite(pk, 1, sz, fo) != sz) { p.fail(-3); return; }
            } else if (s->n && fwrite(s->out, 1, s->n, fo) != s->n) { p.fail(-3); return; }
            const t_index base = s->n ? s->base : 0;
            if (fwrite(&base, 4, 1, fo) != 1) { p.fail(-3); return; }
            p.set(s, kFree);
        }
    });
    reader.join();
    for (auto &t : workers) t.join();
    writer.join();
    return p.rc;
}

int archon_container_decode(FILE *fi, FILE *fo, int ndev, uint32_t *bsize_out)
{
    if (ndev < 1) return -4;
    unsigned short sig = 0;
    uint32_t bsize = 0;
    if (fread(&sig, 2, 1, fi) != 1) return -3;
    if (sig == kSigPostV1) { fprintf(stderr, "archon: 'RM' container of the first post-stage revision (4 MiB pieces): not readable by this build\n"); return -3; }
    if (sig != kSig && sig != kSigPost) return -3;
    const bool post = sig == kSigPost;
    if (fread(&bsize, 4, 1, fi) != 1 || bsize < 8 || bsize > (1u << 28)) return -3;
    uint32_t piece = 0;
    if (post && (fread(&piece, 4, 1, fi) != 1 || piece < 1024 || piece > (4u << 20))) return -3;
    const size_t kPiece = piece;
    if (bsize_out) *bsize_out = bsize;
    Pipe p;
    const int kWorkersPerGpu = workers_per_gpu(bsize);
    // 'RN' containers are decoded on the GPU when their pieces are the encoder's (32 KiB): the packed stream goes up the link, the
    // stage is undone in front of the inverse transform (archon_hip_inverse_post).  Other piece sizes: host threads (post_unpack).
    const bool gpu_post = post && kPiece == kPieceEnc;
    const size_t in_cap = gpu_post ? post_slot_bytes(bsize) : (size_t)bsize + 4;       // (what the encoder's slots hold: it refuses longer streams)
    if (!p.alloc((kWorkersPerGpu + 2) * ndev, in_cap, (size_t)bsize + 4)) return ARCHON_E_NOMEM;

    std::thread reader([&] {
        for (long b = 0;; ++b) {
            Slot *s;
            if (!p.wait(b, kFree, &s)) return;
            if (post && gpu_post) {
                // the packed stream as it is: the worker's GPU decodes it (csrc/post.hiph) in front of the inverse transform
                uint32_t sz = 0;
                if (fread(&sz, 4, 1, fi) != 1 || sz < 4 || sz > 4 + ((size_t)bsize / kPiece + 1) * (4 + archon_post_bound(kPiece))) { p.fail(-2); return; }
                if (sz > in_cap) {
                    // a stream longer than the pinned slot (the format allows up to 2.5 n, this build's encoder stops at 1.25 n): undone on
                    // the host, the worker then runs the plain inverse on the slot (packed = 0)
                    std::vector<byte> packed(sz);
                    if (fread(packed.data(), 1, sz, fi) != sz || fread(&s->base, 4, 1, fi) != 1) { p.fail(-2); return; }
                    const long n = post_unpack(packed.data(), sz, s->in, bsize, kPiece);
                    if (n < 0) { p.fail(-2); return; }
                    s->n = (size_t)n;
                    s->packed = 0;
                    s->last = s->n < bsize;
                    if (s->n && s->base >= s->n) { p.fail(-2); return; }
                    const bool last = s->last;
                    p.set(s, kFilled);
                    if (last) { p.total(b + 1); return; }
                    continue;
                }
                if (fread(s->in, 1, sz, fi) != sz || fread(&s->base, 4, 1, fi) != 1) { p.fail(-2); return; }
                s->packed = sz;
                uint32_t np = 0;
                memcpy(&np, s->in, 4);
                s->n = np ? 1 : 0;                 // (the block's length comes out of the decoder; an empty last block has no pieces)
                s->last = false;
                if (np == 0) { s->last = true; }
                else {
                    // a block is short -- the last one -- when its last piece is, or when it has fewer pieces than a whole block
                    if ((size_t)np < ((size_t)bsize + kPiece - 1) / kPiece) s->last = true;
                    else {
                        size_t off = 4 + 4 * (size_t)np;
                        bool ok = off <= sz;
                        for (uint32_t k = 0; ok && k + 1 < np; ++k) { uint32_t ps; memcpy(&ps, s->in + 4 + 4 * k, 4); off += ps; ok = off + 4 <= sz; }
                        if (!ok || off + 4 > sz) { p.fail(-2); return; }      // (the last piece's length word lies inside the stream, also when it is the only piece)
                        uint32_t ln; memcpy(&ln, s->in + off, 4);
                        s->last = (size_t)(np - 1) * kPiece + ln < bsize;
                    }
                }
                const bool last = s->last;
                p.set(s, kFilled);
                if (last) { p.total(b + 1); return; }
                continue;
            } else if (post) {
                uint32_t sz = 0;
                if (fread(&sz, 4, 1, fi) != 1 || sz > 4 + ((size_t)bsize / kPiece + 1) * (4 + archon_post_bound(kPiece))) { p.fail(-2); return; }
                std::vector<byte> packed(sz);
                if (fread(packed.data(), 1, sz, fi) != sz || fread(&s->base, 4, 1, fi) != 1) { p.fail(-2); return; }
                const long n = post_unpack(packed.data(), sz, s->in, bsize, kPiece);
                if (n < 0) { p.fail(-2); return; }
                s->n = (size_t)n;
            } else {
                const size_t got = fread(s->in, 1, (size_t)bsize + 4, fi);
                if (got < 4) { p.fail(-2); return; }
                s->n = got - 4;
                memcpy(&s->base, s->in + s->n, 4);
            }
            s->last = s->n < bsize;
            if (s->n && s->base >= s->n) { p.fail(-2); return; }
            const bool last = s->last;
            p.set(s, kFilled);
            if (last) { p.total(b + 1); return; }
        }
    });
    std::vector<std::thread> workers;
    for (int w = 0; w < kWorkersPerGpu * ndev; ++w)
        workers.emplace_back([&, w] {
            archon_hip_bind_context(w % ndev, w / ndev);
            worker_loop(p, w, kWorkersPerGpu * ndev, w % ndev, [gpu_post, bsize](Slot &s, int dev) {
                if (gpu_post && s.packed) {
                    uint32_t n = 0;
                    const int rc = archon_hip_inverse_post(s.in, s.packed, s.base, s.out, bsize, &n, dev);
                    s.n = n;
                    return rc;
                }
                return archon_hip_inverse(s.in, (uint32_t)s.n, s.base, s.out, dev);
            });
        });
    std::thread writer([&] {
        for (long b = 0;; ++b) {
            Slot *s;
            if (!p.wait(b, kDone, &s)) return;
            if (s->n && fwrite(s->out, 1, s->n, fo) != s->n) { p.fail(-3); return; }
            p.set(s, kFree);
        }
    });
    reader.join();
    for (auto &t : workers) t.join();
    writer.join();
    return p.rc;
}
