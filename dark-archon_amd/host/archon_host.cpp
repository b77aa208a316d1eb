// archon_host.cpp -- the reference's block coder object on top of the HIP C ABI.
// Mirrors kvark/dark-archon bwt/a7/src/archon.cpp:823-943 method by method; the
// compute bodies are calls into libarchon_hip.so (no CPU fallback: they return the
// library's negative code when no MI355X is present).
#include "archon_host.h"

#include <stdlib.h>
#include <string.h>

#include "../../include/archon.h"
#include "../../include/archon_hip.h"

// --- INITIALIZATION (archon.cpp:827-857) ---------------------------------------
t_index Archon::estimateReserve(const t_index n)
{
    // The reference reserves room for SA-IS bucket tables (R, R2, D: archon.cpp:828-837).
    // The GPU path needs no host-side tables; the figure is kept so countMemory() and the
    // "Allocated" line of the CLI agree with the reference for the same N.
    t_index total = 0x10000;
    const t_index add = 0x400;
    if (!(n >> 17)) total = n / 4 + 1;
    if (total < add) total = add;
    return total;
}

static void *block_alloc(size_t bytes, bool *pinned)
{
    void *p = archon_hip_host_alloc(bytes);     // pinned when a device is present
    *pinned = p != NULL;
    if (!p) p = malloc(bytes);
    return p;
}

Archon::Archon(const t_index Nx)
    : Nmax(Nx), Nreserve(estimateReserve(Nx)), P(NULL), str(NULL), N(0), baseId(0), dev(0), pinned(false), last_rc(0), blk(NULL), resident(false)
{
    bool p1 = false, p2 = false;
    *const_cast<suffix **>(&P) = static_cast<suffix *>(block_alloc(((size_t)Nmax + Nreserve) * sizeof(suffix), &p1));
    *const_cast<byte **>(&str) = static_cast<byte *>(block_alloc((size_t)Nmax + 1, &p2));
    if (p1 != p2) {   // keep both on the same allocator
        if (p1) { archon_hip_host_free(P); *const_cast<suffix **>(&P) = static_cast<suffix *>(malloc(((size_t)Nmax + Nreserve) * sizeof(suffix))); }
        if (p2) { archon_hip_host_free(str); *const_cast<byte **>(&str) = static_cast<byte *>(malloc((size_t)Nmax + 1)); }
        p1 = p2 = false;
    }
    pinned = p1;
    const char *e = getenv("ARCHON_DEVICE");
    if (e) dev = atoi(e);
}

Archon::~Archon()
{
    archon_hip_block_destroy(blk);
    if (pinned) { archon_hip_host_free(P); archon_hip_host_free(str); }
    else { free(P); free(str); }
}

void Archon::setDevice(int d)
{
    if (d != dev && blk) { archon_hip_block_destroy(blk); blk = NULL; resident = false; }      // what is resident lives on the old device
    dev = d;
}

unsigned Archon::countMemory() const
{
    return Nmax + (Nmax + Nreserve) * sizeof(suffix);
}

// --- ENCODING (archon.cpp:862-900) -------------------------------------------------
bool Archon::validate()
{
    // The reference walks P and str on the host (archon.cpp:862-874).  Here enCompute has left the block, its suffix array
    // and its BWT on the device: the check runs on what is there -- no 5N-byte upload, no second gather of str[P[i]].
    // (An object that has not computed anything -- or has read another block since: the reference would then test the old P
    //  against the new str and fail, archon.cpp:862-874 -- has nothing resident: the host arrays are checked instead.)
    last_rc = (blk && resident) ? archon_hip_block_validate(blk) : ARCHON_E_ARG;
    if (last_rc == ARCHON_E_ARG) last_rc = archon_hip_validate(str, N, P, dev);
    return last_rc == 1;
}

int Archon::enRead(FILE *const fx, t_index ns)
{
    if (ns > Nmax) ns = Nmax;
    resident = false;               // what the device holds is the previous block
    N = (t_index)fread(str, 1, ns, fx);
    return (int)N;
}

int Archon::enCompute()
{
    // P[0..N) <- suffix array; block, suffix array and BWT stay in HBM, in this object's own handle, for validate / enWrite
    if (!blk) {
        last_rc = archon_hip_block_create(dev, &blk);
        if (last_rc != ARCHON_OK) return last_rc;
    }
    last_rc = archon_hip_block_forward(blk, str, N, P, &baseId);
    resident = last_rc == ARCHON_OK;
    return last_rc;
}

int Archon::enWrite(FILE *const fx)
{
    // N BWT bytes, then baseId (archon.cpp:895,898).  The reference gathers
    // str[P[i]] bytewise here; the GPU already did, so stream it out through the
    // reserve area of P (O(1) extra host memory).
    byte *bounce = reinterpret_cast<byte *>(P + Nmax);
    const t_index cap = Nreserve * (t_index)sizeof(suffix);
    for (t_index off = 0; off < N;) {
        const t_index len = N - off < cap ? N - off : cap;
        last_rc = (blk && resident) ? archon_hip_block_read_bwt(blk, off, len, bounce) : ARCHON_E_ARG;
        if (last_rc != ARCHON_OK) return last_rc;
        if (fwrite(bounce, 1, len, fx) != len) return -1;
        off += len;
    }
    fwrite(&baseId, sizeof(t_index), 1, fx);
    return 0;
}

// --- DECODING (archon.cpp:905-943) -------------------------------------------------
int Archon::deRead(FILE *const fx, t_index ns)
{
    enRead(fx, ns);
    const size_t ok = fread(&baseId, sizeof(t_index), 1, fx);
    if (!ok || baseId >= N) return -1;
    return (int)N;
}

int Archon::deCompute()
{
    // LF table + walk on the GPU; the decoded block lands in P's storage (N of its 4N bytes)
    last_rc = archon_hip_inverse(str, N, baseId, reinterpret_cast<byte *>(P), dev);
    return last_rc;
}

int Archon::deWrite(FILE *const fx)
{
    if (last_rc != ARCHON_OK) return last_rc;
    return fwrite(P, 1, N, fx) == N ? 0 : -1;
}

// --- C API (include/archon.h) --------------------------------------------------------
struct archon { Archon impl; explicit archon(t_index n) : impl(n) {} };

extern "C" {
uint32_t archon_estimate_reserve(uint32_t n) { return Archon::estimateReserve(n); }
archon_t *archon_create(uint32_t nmax) { return new archon(nmax); }
void archon_destroy(archon_t *a) { delete a; }
unsigned archon_count_memory(const archon_t *a) { return a->impl.countMemory(); }
int archon_validate(archon_t *a) { return a->impl.validate() ? 1 : 0; }
int archon_en_read(archon_t *a, FILE *fx, uint32_t ns) { return a->impl.enRead(fx, ns); }
int archon_en_compute(archon_t *a) { return a->impl.enCompute(); }
int archon_en_write(archon_t *a, FILE *fx) { return a->impl.enWrite(fx); }
int archon_de_read(archon_t *a, FILE *fx, uint32_t ns) { return a->impl.deRead(fx, ns); }
int archon_de_compute(archon_t *a) { return a->impl.deCompute(); }
int archon_de_write(archon_t *a, FILE *fx) { return a->impl.deWrite(fx); }
const uint32_t *archon_sa(const archon_t *a) { return a->impl.sa(); }
uint32_t archon_base_id(const archon_t *a) { return a->impl.baseIndex(); }
uint32_t archon_length(const archon_t *a) { return a->impl.length(); }
void archon_set_device(archon_t *a, int dev) { a->impl.setDevice(dev); }
int archon_last_error(const archon_t *a) { return a->impl.lastError(); }
}
