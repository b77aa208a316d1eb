// Micro-benchmark: does the ALIGNMENT of scattered runs matter?  Every wave writes runs of 256 bytes (16 B per lane,
// 16 lanes per run, 4 runs per instruction) into its own slot of `stride` bytes; the run starts `off` bytes into
// the slot, off = 16 * (hash & mask): mask 0 -> every run starts on a 64/128-byte boundary, mask 3 -> starts
// anywhere on a 16-byte grid inside a 64-byte sector, mask 7 -> inside a 128-byte line.  Useful bytes / time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k(uint8_t *buf, uint64_t nslots, uint32_t stride, uint32_t runs_per_wave, uint32_t mask)
{
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t sub = lane >> 4, off = (lane & 15) * 16;
    for (uint32_t i = 0; i < runs_per_wave; i += 4) {
        const uint64_t r = (uint64_t)wave * runs_per_wave + i + sub;
        const uint64_t slot = (r * 0x9E3779B1ull) % nslots;
        const uint32_t sh = (((uint32_t)(r * 2654435761u) >> 13) & mask) * 16;
        *reinterpret_cast<uint4 *>(buf + slot * stride + sh + off) = make_uint4(r, r, r, r);
    }
}

int main()
{
    const size_t bytes = 2ull << 30;
    const uint32_t stride = 512;
    uint8_t *buf;
    if (hipMalloc(&buf, bytes + 4096) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    const uint64_t nslots = bytes / stride;
    const uint32_t nwaves = 256 * 16 * 4, runs_per_wave = (uint32_t)(nslots / nwaves);
    for (uint32_t mask : {0u, 1u, 2u, 4u, 6u, 3u, 7u, 0u}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(nwaves / 4), dim3(256), 0, 0, buf, nslots, stride, runs_per_wave, mask);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it && ms < best) best = ms;
        }
        printf("run 256 B, start on a 16-byte grid with mask %u: %.3f ms  %.2f TB/s of useful bytes\n", mask, best,
               (double)nslots * 256 / best / 1e9);
        fflush(stdout);
    }
    hipFree(buf);
    return 0;
}
