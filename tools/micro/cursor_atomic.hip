// cursor_atomic.hip -- micro-benchmark (not product): what one 64-bit atomicAdd on a list cursor costs a workgroup when the
// whole chip shares the cursor, as the S kernel of the refinement rounds does (rounds.hiph, k_round_fused: one atomic per
// window of 3072 entries, a window every ~39 us per workgroup, 512 workgroups in flight) -- against 2, 8 and 64 cursors.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/cursor_atomic tools/micro/cursor_atomic.hip && /tmp/cursor_atomic
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

// every workgroup: `iters` times { busy for ~work_cycles; lane 0 adds to cursor[(blockIdx % ncur) * 32]; everybody waits for it }
__global__ __launch_bounds__(512) void k_cursor(unsigned long long *cursor, uint32_t ncur, uint32_t iters, uint32_t work_cycles,
                                                unsigned long long *wait_sum, uint32_t *sink)
{
    __shared__ unsigned long long s_got;
    unsigned long long waited = 0;
    uint32_t acc = threadIdx.x;
    for (uint32_t it = 0; it < iters; ++it) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < work_cycles) acc = acc * 1664525u + 1013904223u;
        __syncthreads();
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) s_got = atomicAdd(cursor + (size_t)(blockIdx.x % ncur) * 32, 0x100000003ull);
        __syncthreads();
        waited += __builtin_readcyclecounter() - t1;
        acc += (uint32_t)s_got;
    }
    if (threadIdx.x == 0) atomicAdd(wait_sum, waited);
    if (acc == 0x12345u) *sink = acc;
}

int main()
{
    unsigned long long *cursor, *wait_sum;
    uint32_t *sink;
    CHECK(hipMalloc(&cursor, 64 * 32 * 8));
    CHECK(hipMalloc(&wait_sum, 8));
    CHECK(hipMalloc(&sink, 4));
    const uint32_t wgs = 512, iters = 200;
    printf("512 workgroups of 512 lanes, one 64-bit atomicAdd per workgroup every ~W cycles of work; mean wait per atomic (cycles of the shader clock)\n");
    for (uint32_t work : {20000u, 60000u, 90000u}) {
        for (uint32_t ncur : {1u, 2u, 8u, 64u}) {
            CHECK(hipMemset(cursor, 0, 64 * 32 * 8));
            CHECK(hipMemset(wait_sum, 0, 8));
            hipLaunchKernelGGL(k_cursor, dim3(wgs), dim3(512), 0, 0, cursor, ncur, iters, work, wait_sum, sink);
            CHECK(hipDeviceSynchronize());
            unsigned long long w;
            CHECK(hipMemcpy(&w, wait_sum, 8, hipMemcpyDeviceToHost));
            printf("  work %6u cycles, %2u cursor(s): %8.0f cycles per atomic\n", work, ncur, (double)w / ((double)wgs * iters));
        }
    }
    return 0;
}
