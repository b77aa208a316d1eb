// Micro-benchmark: the memory pattern of the in-LDS bucket sort without its sorting -- per workgroup one bucket of ~4096
// records: 8-byte records in (8 or 16 bytes per lane and load), 4-byte SA rows out (16 bytes per lane), 1-byte BWT
// symbols out (4 or 16 bytes per lane) -- to see what the memory system gives that kernel.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int LOADW, int BWTW, int ROUNDS>
__global__ __launch_bounds__(512, 6) void k_ls(const uint2 *__restrict__ in, uint32_t *__restrict__ sa, uint8_t *__restrict__ bwt, uint32_t per)
{
    __shared__ uint64_t s_ic[4608];
    const uint32_t lo = blockIdx.x * per + (blockIdx.x * 2654435761u >> 28);      // buckets start anywhere
    const uint32_t nb = per - 16;
    if (LOADW == 8) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t p = (r * 512 + threadIdx.x) % nb;
            const uint2 e = in[lo + p];
            s_ic[p] = (uint64_t)e.y | ((uint64_t)(e.x >> 24) << 32);
        }
    } else {
#pragma unroll
        for (int r = 0; r < (ROUNDS + 1) / 2; ++r) {
            const uint32_t p = ((r * 512 + threadIdx.x) * 2) % (nb & ~1u);
            const uint4 e = *reinterpret_cast<const uint4 *>(in + ((lo + p) & ~1u));
            s_ic[p] = (uint64_t)e.y | ((uint64_t)(e.x >> 24) << 32);
            s_ic[p + 1] = (uint64_t)e.w | ((uint64_t)(e.z >> 24) << 32);
        }
    }
    __syncthreads();
    const uint32_t g_lo = lo & ~3u, hi = lo + nb;
    for (uint32_t g = g_lo + threadIdx.x * 4; g + 4 <= hi; g += 2048) {
        if (g < lo) continue;
        const uint32_t p = g - lo;
        const uint64_t e0 = s_ic[p], e1 = s_ic[p + 1], e2 = s_ic[p + 2], e3 = s_ic[p + 3];
        *reinterpret_cast<uint4 *>(sa + g) = make_uint4((uint32_t)e0, (uint32_t)e1, (uint32_t)e2, (uint32_t)e3);
        if (BWTW == 4)
            *reinterpret_cast<uint32_t *>(bwt + g) = (uint32_t)(e0 >> 32) | ((uint32_t)(e1 >> 32) << 8) | ((uint32_t)(e2 >> 32) << 16) | ((uint32_t)(e3 >> 32) << 24);
    }
    if (BWTW == 16) {
        const uint32_t g16 = (lo + 15u) & ~15u;
        for (uint32_t g = g16 + threadIdx.x * 16; g + 16 <= hi; g += 512 * 16) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[q] |= (uint32_t)(s_ic[g - lo + q * 4 + j] >> 32) << (8 * j);
            }
            *reinterpret_cast<uint4 *>(bwt + g) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

template <int LOADW, int BWTW, int ROUNDS>
static void run(const uint2 *in, uint32_t *sa, uint8_t *bwt, const char *tag)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ls<LOADW, BWTW, ROUNDS>), dim3(65536), dim3(512), 0, 0, in, sa, bwt, 4096u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    printf("%-50s %7.3f ms  %5.2f TB/s (13 B/item)\n", tag, best, 268435456.0 * 13 / best / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t n = 1ull << 28;
    uint2 *in; uint32_t *sa; uint8_t *bwt;
    if (hipMalloc(&in, n * 8 + (1 << 20)) != hipSuccess || hipMalloc(&sa, n * 4 + (1 << 20)) != hipSuccess || hipMalloc(&bwt, n + (1 << 20)) != hipSuccess) return 1;
    hipMemset(in, 1, n * 8);
    run<8, 4, 9>(in, sa, bwt, "as built: 9 x 8-byte loads, 4-byte BWT stores");
    run<8, 4, 8>(in, sa, bwt, "8 x 8-byte loads, 4-byte BWT stores");
    run<16, 4, 8>(in, sa, bwt, "4 x 16-byte loads, 4-byte BWT stores");
    run<8, 16, 8>(in, sa, bwt, "8 x 8-byte loads, 16-byte BWT stores");
    run<16, 16, 8>(in, sa, bwt, "4 x 16-byte loads, 16-byte BWT stores");
    return 0;
}
