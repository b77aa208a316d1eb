// Micro-benchmark (not product): inv::k_lf_build over 2^28 random bytes and what it costs without its parts.
//   MODE 0 = the product kernel's body; 1 = no look-back (prefix 0: results invalid); 2 = no matching (rank 0: invalid);
//   3 = match masks through LDS (ds_or_b64 of the lane's bit into the digit's cell, read back, cleared by the first lane) -- stable, checked against 0
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lf_build tools/micro/lf_build.hip && /tmp/lf_build
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../dark-archon_amd/csrc/inverse.hiph"
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)
namespace archon { namespace inv {
template <int MODE, int IPT>
__global__ __launch_bounds__(kBlock) void k_lf_m(const uint8_t *__restrict__ bwt, uint32_t n, uint32_t base, const uint32_t *__restrict__ starts,
                                                 uint32_t *__restrict__ T, uint32_t *__restrict__ status, uint32_t *__restrict__ ticket, uint32_t *__restrict__ err)
{
    constexpr int TILE = kBlock * IPT;
    __shared__ uint32_t s_whist[kNW][256];
    __shared__ unsigned long long s_bits[MODE == 3 ? kNW : 1][256];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_tile;
    for (int i = threadIdx.x; i < kNW * 256; i += kBlock) (&s_whist[0][0])[i] = 0;
    if (MODE == 3) for (int i = threadIdx.x; i < kNW * 256; i += kBlock) (&s_bits[0][0])[i] = 0;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tile_base = tile * TILE;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t chunk = tile_base + w * (kWave * IPT);
    uint8_t sym[IPT];
    uint16_t pos[IPT];
    uint32_t *wh = s_whist[w];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = chunk + r * kWave + lane;
        sym[r] = i < n ? bwt[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = chunk + r * kWave + lane;
        const bool valid = i < n && i != base;
        const uint32_t d = sym[r];
        uint64_t m;
        if (MODE == 2) m = 1ull << lane;
        else if (MODE == 3) {
            unsigned long long *cell = &s_bits[w][d];
            if (valid) atomicOr(cell, 1ull << lane);
            m = valid ? *cell : 0ull;
        } else m = match_digit8(d, valid);
        const uint32_t rank = mbcnt64(m);
        const uint32_t prev = wh[d];
        pos[r] = (uint16_t)(prev + rank);
        if (valid && rank == 0) {
            wh[d] = prev + (uint32_t)__popcll(m);
            if (MODE == 3) s_bits[w][d] = 0ull;
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const uint32_t d = threadIdx.x;
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < kNW; ++i) {
            const uint32_t c = s_whist[i][d];
            s_whist[i][d] = total;
            total += c;
        }
        uint32_t excl = 0;
        uint32_t *mine = status + (size_t)tile * 256 + d;
        if (MODE == 1) {
            st_agent(mine, rs::kFlagPre | total);
        } else if (tile == 0) {
            st_agent(mine, rs::kFlagPre | total);
        } else {
            st_agent(mine, rs::kFlagAgg | total);
            excl = rs::look_back(status, tile, d, err);
            st_agent(mine, rs::kFlagPre | ((excl + total) & rs::kValMask));
        }
        s_gbase[d] = starts[d] + excl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = chunk + r * kWave + lane;
        if (i < n) {
            const uint32_t d = sym[r];
            T[i] = (i == base) ? starts[d + 1] - 1u : s_gbase[d] + s_whist[w][d] + pos[r];
        }
    }
}
}}
using namespace archon;
template <int MODE, int IPT>
static float run(const uint8_t *bwt, uint32_t n, uint32_t base, const uint32_t *starts, uint32_t *T, uint32_t *status, uint32_t *small, int reps)
{
    const uint32_t tiles = (n + inv::kBlock * IPT - 1) / (inv::kBlock * IPT);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipMemsetAsync(status, 0, (size_t)tiles * 256 * 4, 0));
        CHECK(hipMemsetAsync(small, 0, 64, 0));
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(inv::k_lf_m<MODE, IPT>), dim3(tiles), dim3(inv::kBlock), 0, 0, bwt, n, base, starts, T, status, small, small + 1);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    const uint32_t n = 1u << 28;
    uint8_t *h = (uint8_t *)malloc(n);
    uint64_t s = 88172645463325252ull;
    uint32_t cnt[257] = {0};
    for (uint32_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint8_t)(s >> 32); }
    const uint32_t base = 12345;
    for (uint32_t i = 0; i < n; ++i) ++cnt[h[i] + 1];
    for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
    uint8_t *bwt; uint32_t *starts, *T, *T2, *status, *small;
    CHECK(hipMalloc(&bwt, n)); CHECK(hipMalloc(&starts, 257 * 4)); CHECK(hipMalloc(&T, (size_t)n * 4)); CHECK(hipMalloc(&T2, (size_t)n * 4));
    CHECK(hipMalloc(&status, (size_t)(n / 4096 + 1) * 256 * 4)); CHECK(hipMalloc(&small, 64));
    CHECK(hipMemcpy(bwt, h, n, hipMemcpyHostToDevice)); CHECK(hipMemcpy(starts, cnt, 257 * 4, hipMemcpyHostToDevice));
    printf("product body (16 per lane)            %.3f ms\n", run<0, 16>(bwt, n, base, starts, T, status, small, 5));
    printf("no look-back                          %.3f ms\n", run<1, 16>(bwt, n, base, starts, T2, status, small, 5));
    printf("no matching                           %.3f ms\n", run<2, 16>(bwt, n, base, starts, T2, status, small, 5));
    printf("match masks through LDS (16 per lane) %.3f ms\n", run<3, 16>(bwt, n, base, starts, T2, status, small, 5));
    uint32_t *a = (uint32_t *)malloc((size_t)n * 4), *b = (uint32_t *)malloc((size_t)n * 4);
    CHECK(hipMemcpy(a, T, (size_t)n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b, T2, (size_t)n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (uint32_t i = 0; i < n; ++i) bad += a[i] != b[i];
    printf("  LDS masks against ballots: %zu differences\n", bad);
    printf("product body, 32 per lane             %.3f ms\n", run<0, 32>(bwt, n, base, starts, T2, status, small, 5));
    printf("LDS masks, 32 per lane                %.3f ms\n", run<3, 32>(bwt, n, base, starts, T2, status, small, 5));
    CHECK(hipMemcpy(b, T2, (size_t)n * 4, hipMemcpyDeviceToHost));
    bad = 0; for (uint32_t i = 0; i < n; ++i) bad += a[i] != b[i];
    printf("  32 per lane against 16: %zu differences\n", bad);
    printf("LDS masks, 8 per lane                 %.3f ms\n", run<3, 8>(bwt, n, base, starts, T2, status, small, 5));
    uint32_t e; CHECK(hipMemcpy(&e, small + 1, 4, hipMemcpyDeviceToHost)); printf("err flag %u\n", e);
    return 0;
}
