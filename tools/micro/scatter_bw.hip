// Micro-benchmark: HBM write bandwidth for scattered runs of L bytes (what a radix pass emits).
// Every workgroup writes `runs_per_block` runs; run r goes to slot perm(r) of a 2 GiB buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

template <int LANE_BYTES>
__global__ __launch_bounds__(256) void k_scatter(uint8_t *buf, uint64_t nslots, uint32_t run_bytes, uint32_t runs_per_wave, uint32_t mul)
{
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t lanes_per_run = run_bytes / LANE_BYTES;          // <= 64
    const uint32_t runs_per_instr = 64 / lanes_per_run;
    const uint32_t sub = lane / lanes_per_run, off = (lane % lanes_per_run) * LANE_BYTES;
    for (uint32_t i = 0; i < runs_per_wave; i += runs_per_instr) {
        const uint64_t r = (uint64_t)wave * runs_per_wave + i + sub;
        const uint64_t slot = (r * mul) % nslots;                    // pseudo-random permutation (mul odd, nslots power of 2)
        uint8_t *p = buf + slot * run_bytes + off;
        if (LANE_BYTES == 4) *reinterpret_cast<uint32_t *>(p) = (uint32_t)r;
        else if (LANE_BYTES == 16) *reinterpret_cast<uint4 *>(p) = make_uint4(r, r, r, r);
        else if (LANE_BYTES == 1) *p = (uint8_t)r;
        else if (LANE_BYTES == 2) *reinterpret_cast<uint16_t *>(p) = (uint16_t)r;
    }
}

template <int LB>
void run(uint8_t *buf, size_t bytes, uint32_t run_bytes, uint32_t mul, const char *tag)
{
    const uint64_t nslots = bytes / run_bytes;
    const uint32_t nwaves = 256 * 16 * 4;                            // 16 blocks of 4 waves per CU
    const uint32_t runs_per_wave = (uint32_t)(nslots / nwaves);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scatter<LB>), dim3(nwaves / 4), dim3(256), 0, 0, buf, nslots, run_bytes, runs_per_wave, mul);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("%-10s lane=%2dB run=%5u B  %8.3f ms  %7.2f TB/s\n", tag, LB, run_bytes, best, bytes / best / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t bytes = 2ull << 30;
    uint8_t *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    for (uint32_t rb : {32u, 64u, 128u, 256u, 512u, 1024u, 4096u}) {
        if (rb <= 256) run<4>(buf, bytes, rb, 0x9E3779B1u, "random");
        if (rb >= 64 && rb <= 1024) run<16>(buf, bytes, rb, 0x9E3779B1u, "random");
    }
    for (uint32_t rb : {64u, 256u}) run<4>(buf, bytes, rb, 1u, "sequential");
    run<16>(buf, bytes, 1024, 1u, "sequential");
    run<1>(buf, bytes, 32, 0x9E3779B1u, "random");
    run<1>(buf, bytes, 64, 0x9E3779B1u, "random");
    run<2>(buf, bytes, 64, 0x9E3779B1u, "random");
    run<2>(buf, bytes, 128, 0x9E3779B1u, "random");
    hipFree(buf);
    return 0;
}
