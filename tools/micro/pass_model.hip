// Micro-benchmark: a radix pass reduced to its skeleton -- per tile: coalesced loads (1 + 8 bytes per item), an
// LDS-only "compute" phase (ranking atomics, barriers, a staging scatter), scattered 16-byte stores in runs -- to
// measure what co-resident workgroups buy: one 1024-lane workgroup per CU on 12 288-item tiles against two
// 512-lane workgroups on 6 144-item tiles (and other splits).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK) void k_model(const uint2 *__restrict__ in, const uint8_t *__restrict__ inb, uint2 *__restrict__ out,
                                                 uint32_t tiles_per_wg, uint32_t compute, uint64_t out_mask, uint32_t mode)
{
    extern __shared__ uint32_t lds[];                 // size decides how many workgroups share a CU
    constexpr int T = BLOCK * IPT;
    constexpr uint32_t kStage = T <= 4096 ? 4096 : T <= 8192 ? 8192 : 16384;     // slots (power of two: cheap index)
    uint32_t *cnt = lds;                              // 256 counters
    uint2 *stage = reinterpret_cast<uint2 *>(lds + 256);
    for (int i = threadIdx.x; i < 256; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t t = 0; t < tiles_per_wg; ++t) {
        const size_t tile = (size_t)blockIdx.x * tiles_per_wg + t;
        const size_t off = tile * T + w * (64 * IPT) + lane;
        uint32_t dg[IPT];
        uint2 rec[IPT];
#pragma unroll
        for (int r = 0; r < IPT; ++r) dg[r] = inb[off + r * 64];
#pragma unroll
        for (int r = 0; r < IPT; ++r) rec[r] = in[off + r * 64];
        // compute phase: ranking atomics, layout barriers, staging scatter (compute = how many times over)
        for (uint32_t c = 0; c < compute; ++c) {
#pragma unroll
            for (int r = 0; r < IPT; ++r) dg[r] = (dg[r] & 0xFFu) | (atomicAdd(&cnt[(dg[r] + c) & 0xFFu], 1u) << 16);
            __syncthreads();
            if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < IPT; ++r) stage[((dg[r] & 0xFFu) * (T / 256) + ((dg[r] >> 16) & 63u)) & (kStage - 1)] = rec[r];
            __syncthreads();
        }
        // stores: pairs of records, runs of T/256 records per "digit", every run at its own pseudo-random place
        constexpr int kPairs = T / 2 / BLOCK;
        constexpr int kRun = T / 256;                  // records per run
#pragma unroll
        for (int k = 0; k < kPairs; ++k) {
            const uint32_t p = threadIdx.x + k * BLOCK;                 // pair index in the tile
            const uint4 v = *reinterpret_cast<const uint4 *>(&stage[2 * p]);
            const uint32_t run = (2 * p) / kRun, within = (2 * p) % kRun;
            uint64_t base;
            if (mode == 4 || mode == 5) {
                // continuing runs whose pieces are whole lines (mode 4: 128 bytes, mode 5: 64 bytes; alternating piece lengths emulate the carry)
                const uint32_t g = mode == 4 ? 16u : 8u;
                const uint64_t seg = (((uint64_t)run * gridDim.x + blockIdx.x) * (uint64_t)(tiles_per_wg * kRun + 16)) & ~(uint64_t)15;
                base = (seg + (mode == 5 ? 8u : 0u) + (uint64_t)t * kRun) & out_mask;
                (void)g;
            } else if (mode == 2) {
                // what a pass really writes: digit `run` of workgroup b continues where the previous tile ended;
                // the workgroup's segment of the digit starts at an arbitrary multiple of 4 records
                const uint64_t seg = ((uint64_t)run * gridDim.x + blockIdx.x) * (uint64_t)(tiles_per_wg * kRun + 16);
                base = (seg + (((run * 2654435761u + blockIdx.x * 40503u) >> 7) & 12u) + (uint64_t)t * kRun) & out_mask;
            } else {
                base = ((tile * 256 + run) * 0x9E3779B97F4A7C15ull) >> 20;
                base = (base & out_mask) & ~(uint64_t)(mode == 0 ? 15 : mode == 3 ? 7 : 3);  // isolated runs: 128- / 64- / 32-byte aligned start
            }
            *reinterpret_cast<uint4 *>(out + base + within) = v;
        }
        __syncthreads();
    }
}

template <int BLOCK, int IPT>
static void run(const uint2 *in, const uint8_t *inb, uint2 *out, size_t nitems, int wg_per_cu, uint32_t compute, size_t lds_bytes, const char *tag, uint32_t mode = 0)
{
    constexpr int T = BLOCK * IPT;
    const uint32_t grid = 256 * wg_per_cu;
    const uint32_t tiles_per_wg = (uint32_t)(nitems / T / grid);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_model<BLOCK, IPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_model<BLOCK, IPT>), dim3(grid), dim3(BLOCK), lds_bytes, 0, in, inb, out, tiles_per_wg, compute,
                           (uint64_t)((1ull << 28) - 1), mode);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double items = (double)tiles_per_wg * grid * T;
    printf("%-36s mode %u compute x%u  %7.3f ms  %6.2f TB/s (17 B/item)  %5.2f us per 12288 items per CU\n", tag, mode, compute, best, items * 17 / best / 1e9,
           best * 1e3 / (items / 256 / 12288));
    fflush(stdout);
}

int main()
{
    const size_t n = 1ull << 28;
    uint2 *in, *out;
    uint8_t *inb;
    if (hipMalloc(&in, n * 8 + (1 << 20)) != hipSuccess || hipMalloc(&out, (n + (1 << 16)) * 8) != hipSuccess || hipMalloc(&inb, n + (1 << 20)) != hipSuccess) return 1;
    hipMemset(in, 1, n * 8);
    {   // random digits
        uint8_t *h = (uint8_t *)malloc(n);
        uint64_t z = 88172645463325252ull;
        for (size_t i = 0; i < n; i += 8) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; *(uint64_t *)(h + i) = z; }
        hipMemcpy(inb, h, n, hipMemcpyHostToDevice);
        free(h);
    }
    // modes: 0 isolated runs starting on a 128-byte line, 1 isolated runs starting on any 32-byte boundary,
    //        2 runs that continue from tile to tile (what a pass writes)
    for (uint32_t mode : {0u, 3u, 1u, 4u, 5u})
        for (uint32_t compute : {0u, 1u}) {
            run<1024, 12>(in, inb, out, n, 1, compute, 150 * 1024, "1 x 1024 lanes, tile 12288", mode);
            run<512, 12>(in, inb, out, n, 2, compute, 78 * 1024, "2 x 512 lanes, tile 6144", mode);
            run<512, 8>(in, inb, out, n, 3, compute, 52 * 1024, "3 x 512 lanes, tile 4096", mode);
        }
    return 0;
}
