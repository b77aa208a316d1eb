// gather_chain.hip -- micro-benchmark (not product): what a random 4-byte gather costs on MI355X, independent against
// dependent (pointer chasing), by table size.  Behind DESIGN.md 3.3 (the LF walk of the inverse BWT is 2^28 dependent
// gathers from a 1 GiB table) and 3.2 (a refinement round is one independent gather per tied entry from the rank table).
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_chain tools/micro/gather_chain.hip && /tmp/gather_chain
//
// Table: T[i] = a full-cycle affine map of i (i * A + C mod n, n a power of two, A = 5 mod 8, C odd), so every chain visits
// every slot once and successive hops land on unrelated lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

__global__ void k_fill(uint32_t *t, uint32_t n, uint32_t a, uint32_t c)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) t[i] = (i * a + c) & (n - 1);
}

// independent gathers: lane j reads T[hash(j + r * total)] for r = 0 .. reps-1, U of them in flight
template <int U>
__global__ __launch_bounds__(256) void k_indep(const uint32_t *__restrict__ t, uint32_t n, uint32_t reps, uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x, total = gridDim.x * 256;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < reps; r += U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t k = (j + (r + u) * total) * 2654435761u;
            k ^= k >> 15;
            v[u] = t[k & (n - 1)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// dependent gathers: every lane follows its own chain k = T[k] for `hops` hops
__global__ __launch_bounds__(256) void k_chase(const uint32_t *__restrict__ t, uint32_t n, uint32_t hops, uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x, total = gridDim.x * 256;
    uint32_t k = (uint32_t)(((uint64_t)j * n) / total);
    for (uint32_t r = 0; r < hops; ++r) k = t[k];
    if (k == 0xFFFFFFFFu) out[0] = k;
}

// two chains per lane (independent of each other): does a second outstanding load per lane help?
__global__ __launch_bounds__(256) void k_chase2(const uint32_t *__restrict__ t, uint32_t n, uint32_t hops, uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x, total = gridDim.x * 256;
    uint32_t k0 = (uint32_t)(((uint64_t)(2 * j) * n) / (2ull * total)), k1 = (uint32_t)(((uint64_t)(2 * j + 1) * n) / (2ull * total));
    for (uint32_t r = 0; r < hops; ++r) { k0 = t[k0]; k1 = t[k1]; }
    if ((k0 ^ k1) == 0xFFFFFFFFu) out[0] = k0;
}

static float run(void (*launch)(void *), void *arg)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(arg);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
        CHECK(hipEventRecord(a));
        launch(arg);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

struct Args { const uint32_t *t; uint32_t n, a, b; uint32_t *out; int variant; };
static void launch(void *p)
{
    Args *g = (Args *)p;
    switch (g->variant) {
    case 1: hipLaunchKernelGGL(k_indep<1>, dim3(g->a), dim3(256), 0, 0, g->t, g->n, g->b, g->out); break;
    case 4: hipLaunchKernelGGL(k_indep<4>, dim3(g->a), dim3(256), 0, 0, g->t, g->n, g->b, g->out); break;
    case 8: hipLaunchKernelGGL(k_indep<8>, dim3(g->a), dim3(256), 0, 0, g->t, g->n, g->b, g->out); break;
    case 100: hipLaunchKernelGGL(k_chase, dim3(g->a), dim3(256), 0, 0, g->t, g->n, g->b, g->out); break;
    case 101: hipLaunchKernelGGL(k_chase2, dim3(g->a), dim3(256), 0, 0, g->t, g->n, g->b, g->out); break;
    }
}

int main()
{
    uint32_t *t, *out;
    const uint32_t nmax = 1u << 28;
    CHECK(hipMalloc(&t, (size_t)nmax * 4));
    CHECK(hipMalloc(&out, 256));
    printf("random 4-byte gathers, MI355X; G/s = 1e9 gathers per second; ps = picoseconds per gather (chip-wide)\n");
    for (uint32_t lg = 22; lg <= 28; lg += 2) {
        const uint32_t n = 1u << lg;
        hipLaunchKernelGGL(k_fill, dim3(n / 256), dim3(256), 0, 0, t, n, 0x9E3779B5u, 0x7F4A7C15u);
        CHECK(hipDeviceSynchronize());
        printf("table %4u MiB\n", n >> 18);
        const uint64_t total = 1ull << 28;              // gathers per launch
        for (int U : {1, 4, 8}) {
            Args g{t, n, 256u * 32u, (uint32_t)(total / (256ull * 32 * 256)), out, U};
            const float ms = run(launch, &g);
            printf("  independent, %d in flight per lane, 2^21 lanes : %7.3f ms  %6.1f G/s  %5.1f ps\n", U, ms, total / ms / 1e6, ms * 1e9 / total);
        }
        for (uint32_t lc = 16; lc <= 23; ++lc) {
            const uint32_t chains = 1u << lc;
            Args g{t, n, chains / 256, (uint32_t)(total / chains), out, 100};
            const float ms = run(launch, &g);
            printf("  dependent, 2^%-2u chains of %8u hops           : %7.3f ms  %6.1f G/s  %5.1f ps\n", lc, (uint32_t)(total / chains), ms, total / ms / 1e6, ms * 1e9 / total);
        }
        for (uint32_t lc = 19; lc <= 22; ++lc) {
            const uint32_t chains = 1u << lc;
            Args g{t, n, chains / 512, (uint32_t)(total / chains), out, 101};
            const float ms = run(launch, &g);
            printf("  dependent, 2^%-2u chains, two per lane            : %7.3f ms  %6.1f G/s  %5.1f ps\n", lc, ms, total / ms / 1e6, ms * 1e9 / total);
        }
    }
    return 0;
}
