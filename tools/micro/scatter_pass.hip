// Micro-benchmark: one LSB pass of rs::k_scatter over 2^28 random (u64 key, u32 item) pairs, and what it costs without
// its parts: MODE 0 = the product kernel's body, 1 = ranks from LDS atomics instead of ballot matching (unstable: timing
// only), 2 = no global stores, 3 = the tiles' prefixes read from a table (what the look-back costs: the table is the status array
// a finished pass leaves behind), 4 = match masks through LDS (ds_or_b64; stable), 5 = 3 + 4.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../dark-archon_amd/csrc/radix_sort.hiph"
namespace archon { namespace rs {
// ---- one LSB pass -------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock, 4) void k_scatter_m(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                    uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, uint32_t n,
                                                    int shift, const uint32_t *__restrict__ gstart /*[256]*/,
                                                    uint32_t *__restrict__ status /*[ntiles][256]*/,
                                                    uint32_t *__restrict__ ticket, uint32_t *__restrict__ err, const uint32_t *__restrict__ ref)
{
    __shared__ uint64_t s_stage[kTile];          // 64 KiB, keys then (as u32) items
    __shared__ uint32_t s_whist[kNW][256];       // per-wave digit counts -> exclusive wave offsets
    __shared__ uint32_t s_dstart[256];           // digit start inside the tile
    __shared__ uint32_t s_gbase[256];            // global address of slot 0 of the digit, minus dstart
    __shared__ uint32_t s_wtot[kNW];
    __shared__ uint32_t s_tile;

    for (int i = threadIdx.x; i < kNW * 256; i += kBlock) (&s_whist[0][0])[i] = 0;
    if (MODE >= 4) for (int i = threadIdx.x; i < kNW * 256; i += kBlock) s_stage[i] = 0ull;      // the match cells live in the staging buffer (free until step 3)
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tile_base = tile * kTile;
    const uint32_t tile_n = (n - tile_base) < (uint32_t)kTile ? (n - tile_base) : (uint32_t)kTile;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t chunk = w * (kWave * kIPT);

    // 1. load (coalesced per wave), rank inside the wave chunk
    uint64_t key[kIPT];
    uint32_t val[kIPT];          // requested with the keys: their latency hides behind the ranking and the look-back
    uint16_t pos[kIPT];
#pragma unroll
    for (int r = 0; r < kIPT; ++r) {
        const uint32_t li = chunk + r * kWave + lane;
        key[r] = li < tile_n ? kin[tile_base + li] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < kIPT; ++r) {
        const uint32_t li = chunk + r * kWave + lane;
        val[r] = li < tile_n ? vin[tile_base + li] : 0u;
    }
    uint32_t *wh = s_whist[w];
#pragma unroll
    for (int r = 0; r < kIPT; ++r) {
        const uint32_t li = chunk + r * kWave + lane;
        const bool valid = li < tile_n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & 0xFFu;
        if (MODE == 1) {
            pos[r] = (uint16_t)atomicAdd(&wh[d], valid ? 1u : 0u);      // unstable, one LDS atomic per item
        } else {
        uint64_t m;
        if (MODE >= 4) {
            unsigned long long *cell = reinterpret_cast<unsigned long long *>(&s_stage[w * 256 + d]);
            if (valid) atomicOr(cell, 1ull << lane);
            m = valid ? *cell : 0ull;
            if (valid && mbcnt64(m) == 0) *cell = 0ull;
        } else m = match_digit8(d, valid);
        const uint32_t rank = mbcnt64(m);
        const uint32_t prev = wh[d];
        pos[r] = (uint16_t)(prev + rank);
        if (valid && rank == 0) wh[d] = prev + (uint32_t)__popcll(m);
        }
    }
    __syncthreads();

    // 2. per digit: exclusive offsets over waves, tile total; then digit starts
    uint32_t total = 0;
    if (threadIdx.x < 256) {
        const uint32_t d = threadIdx.x;
#pragma unroll
        for (int i = 0; i < kNW; ++i) {
            const uint32_t c = s_whist[i][d];
            s_whist[i][d] = total;
            total += c;
        }
    }
    // the tile total goes out at once (later tiles are waiting for it); tile 0's is its inclusive prefix already
    if (threadIdx.x < 256) st_agent(status + (size_t)tile * 256 + threadIdx.x, (tile == 0 ? kFlagPre : kFlagAgg) | total);
    {
        // exclusive scan of `total` over the 256 digit threads (waves 0..3)
        const uint32_t inc = wave_incl_sum(total);
        if (threadIdx.x < 256 && lane == 63) s_wtot[w] = inc;
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t pre = 0;
            for (uint32_t i = 0; i < w; ++i) pre += s_wtot[i];
            s_dstart[threadIdx.x] = pre + inc - total;
        }
    }
    __syncthreads();

    // 3. keys: registers -> LDS in bucket order (their registers are free before the look-back starts)
    uint16_t lp[kIPT];
    uint8_t dg[kIPT];
#pragma unroll
    for (int r = 0; r < kIPT; ++r) {
        const uint32_t li = chunk + r * kWave + lane;
        const uint32_t d = (uint32_t)(key[r] >> shift) & 0xFFu;
        lp[r] = (uint16_t)(s_dstart[d] + s_whist[w][d] + pos[r]);
        if (li < tile_n) s_stage[lp[r]] = key[r];
    }

    // 4. look back for the exclusive prefix over earlier tiles
    if (threadIdx.x < 256) {
        const uint32_t d = threadIdx.x;
        uint32_t excl = 0;
        uint32_t *mine = status + (size_t)tile * 256 + d;
        if ((MODE == 3 || MODE == 5) && tile != 0) {
            excl = ref[(size_t)(tile - 1) * 256 + d] & kValMask;
            st_agent(mine, kFlagPre | ((excl + total) & kValMask));
        } else if (tile != 0) {
            // Look back kLook tiles at a time: the status words of the predecessors are requested together (one latency
            // per kLook tiles instead of one per tile -- with ~500 tiles in flight a tile walks back through dozens of
            // them) and consumed in order, up to the first one that carries an inclusive prefix or is not published yet.
            constexpr uint32_t kLook = 8;
            uint32_t t = tile, spins = 0;          // tiles t-1, t-2, ... are still to be looked at
            for (;;) {
                uint32_t v[kLook];
#pragma unroll
                for (uint32_t j = 0; j < kLook; ++j)
                    v[j] = t > j ? ld_agent(status + (size_t)(t - 1 - j) * 256 + d) : kFlagPre;      // before tile 0: prefix 0
                bool done = false, stall = false;
                uint32_t used = 0;
#pragma unroll
                for (uint32_t j = 0; j < kLook; ++j) {
                    if (done || stall) continue;
                    const uint32_t f = v[j] >> 30;
                    if (f == 0) { stall = true; continue; }
                    excl += v[j] & kValMask;
                    ++used;
                    if (f == 2) done = true;
                }
                t -= used;
                if (done) break;
                if (stall) {
                    if (++spins > kSpinLimit) { atomicOr(err, 1u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            st_agent(mine, kFlagPre | ((excl + total) & kValMask));
        }
        s_gbase[d] = gstart[d] + excl - s_dstart[d];
    }
    __syncthreads();

    // 5. keys: LDS -> global runs
#pragma unroll
    for (int k = 0; k < kIPT; ++k) {
        const uint32_t p = threadIdx.x + k * kBlock;
        if (p < tile_n) {
            const uint64_t kk = s_stage[p];
            dg[k] = (uint8_t)((kk >> shift) & 0xFFu);
            if (MODE != 2) kout[s_gbase[dg[k]] + p] = kk; else if (kk == 0x123456789ull) kout[p] = kk;
        }
    }
    __syncthreads();

    // 6. items, same route
    uint32_t *s_val = reinterpret_cast<uint32_t *>(s_stage);
#pragma unroll
    for (int r = 0; r < kIPT; ++r) {
        const uint32_t li = chunk + r * kWave + lane;
        if (li < tile_n) s_val[lp[r]] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kIPT; ++k) {
        const uint32_t p = threadIdx.x + k * kBlock;
        if (p < tile_n) { if (MODE != 2) vout[s_gbase[dg[k]] + p] = s_val[p]; else if (s_val[p] == 0x12345678u) vout[p] = 1; }
    }
}


}}
using namespace archon;
template <int MODE>
static void run(const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, uint32_t n, const uint32_t *gstart, uint32_t *status, uint32_t *ticket, uint32_t *err, const char *tag, const uint32_t *ref = nullptr)
{
    const uint32_t ntiles = (n + rs::kTile - 1) / rs::kTile;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipMemset(status, 0, (size_t)ntiles * 256 * 4); hipMemset(ticket, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rs::k_scatter_m<MODE>), dim3(ntiles), dim3(rs::kBlock), 0, 0, kin, vin, kout, vout, n, 8, gstart, status, ticket, err, ref);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %5.2f TB/s of 24 B/pair\n", tag, best, (double)n * 24 / best / 1e9);
}
int main()
{
    const uint32_t n = 1u << 28;
    uint64_t *kin, *kout; uint32_t *vin, *vout, *gstart, *status, *ticket;
    hipMalloc(&kin, (size_t)n * 8); hipMalloc(&kout, (size_t)n * 8); hipMalloc(&vin, (size_t)n * 4); hipMalloc(&vout, (size_t)n * 4);
    hipMalloc(&gstart, 1024); hipMalloc(&status, (size_t)(n / rs::kTile + 1) * 1024); hipMalloc(&ticket, 64);
    uint64_t *h = (uint64_t *)malloc((size_t)n * 8);
    uint64_t z = 88172645463325252ull; uint32_t cnt[256] = {0};
    for (uint32_t i = 0; i < n; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = z; ++cnt[(z >> 8) & 255]; }
    hipMemcpy(kin, h, (size_t)n * 8, hipMemcpyHostToDevice); hipMemset(vin, 1, (size_t)n * 4);
    uint32_t st[256], s = 0; for (int d = 0; d < 256; ++d) { st[d] = s; s += cnt[d]; }
    hipMemcpy(gstart, st, 1024, hipMemcpyHostToDevice);
    run<0>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "k_scatter as shipped");
    run<1>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "ranks by LDS atomics (unstable)");
    run<2>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "no global stores");
    const uint32_t ntiles = (n + rs::kTile - 1) / rs::kTile;
    uint32_t *ref; hipMalloc(&ref, (size_t)ntiles * 1024);
    run<0>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "k_scatter as shipped (again)");
    hipMemcpy(ref, status, (size_t)ntiles * 1024, hipMemcpyDeviceToDevice);
    uint64_t *kref; hipMalloc(&kref, (size_t)n * 8); hipMemcpy(kref, kout, (size_t)n * 8, hipMemcpyDeviceToDevice);
    run<3>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "tile prefixes from a table (no look-back)", ref);
    run<4>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "match masks through LDS", ref);
    {
        uint64_t *a = (uint64_t *)malloc((size_t)n * 8);
        hipMemcpy(a, kout, (size_t)n * 8, hipMemcpyDeviceToHost); hipMemcpy(h, kref, (size_t)n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (uint32_t i = 0; i < n; ++i) bad += a[i] != h[i];
        printf("  LDS masks against ballots: %zu keys differ\n", bad);
    }
    run<5>(kin, vin, kout, vout, n, gstart, status, ticket, ticket + 1, "both", ref);
    return 0;
}
