// Micro-benchmark: what the LDS delivers per CU for the operations the radix passes and the bucket sort lean on
// (one 1024-lane workgroup per CU, like the passes): ranking atomics, random staging scatters, ballot matching.
// Prints items per clock per CU at a nominal 2.4 GHz.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

static __device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int kBlock = 1024, kIPT = 12, kRounds = 512;

// mode 0: rtn atomic, 256 bins shared; 1: no-rtn atomic, 256 bins; 2: rtn atomic, per-wave 256 bins;
// 3: rtn atomic, 32768 bins; 4: ds_write_b32 scatter (16K slots); 5: ds_write_b64 scatter (16K slots);
// 6: ballot matching (no LDS); 7: rtn atomic on 256 bins, digits sorted inside the lane first (runs of equal digits);
// 8: rtn atomic 256 bins, all lanes of a wave the same digit (worst case)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_lds(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t s[36864];                      // 144 KiB
    for (int i = threadIdx.x; i < 36864; i += kBlock) s[i] = 0;
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t w = threadIdx.x >> 6;
    uint32_t d0[kIPT];
#pragma unroll
    for (int r = 0; r < kIPT; ++r) d0[r] = mix(seed + (blockIdx.x * kBlock + threadIdx.x) * 131u + r * 7919u);
    for (int it = 0; it < kRounds; ++it) {
        uint32_t d[kIPT];
#pragma unroll
        for (int r = 0; r < kIPT; ++r) d[r] = d0[r] + it * 0x9E3779B1u;      // one VALU op per item: the LDS sets the pace
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) acc += atomicAdd(&s[d[r] & 255u], 1u);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) atomicAdd(&s[d[r] & 255u], 1u);
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) acc += atomicAdd(&s[w * 256 + (d[r] & 255u)], 1u);
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) acc += atomicAdd(&s[d[r] & 32767u], 1u);
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) s[d[r] & 16383u] = d[r];
        } else if (MODE == 5) {
            uint2 *s2 = reinterpret_cast<uint2 *>(s);
#pragma unroll
            for (int r = 0; r < kIPT; ++r) s2[d[r] & 16383u] = make_uint2(d[r], it);
        } else if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) {
                uint64_t m = ~0ull;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool bit = (d[r] >> b) & 1u;
                    const uint64_t bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                acc += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) + (uint32_t)__popcll(m);
            }
        } else if (MODE == 7) {
            // the wave's lanes take their digits from a narrow window: fewer distinct banks per instruction
#pragma unroll
            for (int r = 0; r < kIPT; ++r) acc += atomicAdd(&s[((d[r] & 31u) + (it & 7) * 32u) & 255u], 1u);
        } else if (MODE == 8) {
#pragma unroll
            for (int r = 0; r < kIPT; ++r) acc += atomicAdd(&s[(it * 13 + r) & 255u], 1u);
        }
    }
    __syncthreads();
    acc += s[threadIdx.x];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int MODE>
static void run(uint32_t *out, const char *tag)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lds<MODE>), dim3(256), dim3(kBlock), 0, 0, out, 17u + it);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    const double items_per_cu = (double)kBlock * kIPT * kRounds;
    printf("%-44s %8.3f ms  %6.2f items/clk/CU  (%5.1f clk per wave-instruction)\n", tag, best, items_per_cu / (best * 1e-3 * 2.4e9),
           64.0 / (items_per_cu / (best * 1e-3 * 2.4e9)));
    fflush(stdout);
}

int main()
{
    uint32_t *out;
    if (hipMalloc(&out, 4096) != hipSuccess) return 1;
    run<6>(out, "baseline: digit generation + ballot match");
    run<0>(out, "atomic rtn, 256 bins shared");
    run<1>(out, "atomic no-rtn, 256 bins shared");
    run<2>(out, "atomic rtn, per-wave 256 bins");
    run<3>(out, "atomic rtn, 32768 bins");
    run<7>(out, "atomic rtn, 32-bin window per instruction");
    run<8>(out, "atomic rtn, one bin per instruction");
    run<4>(out, "ds_write_b32 random scatter");
    run<5>(out, "ds_write_b64 random scatter");
    hipFree(out);
    return 0;
}
