// Test-only: occupy `nwg` CUs for about `ms` milliseconds (workgroups of 1024 lanes with 120 KiB of LDS: nothing else
// fits beside one of them on a CU... and a pass workgroup of the library does not fit beside it either), to see what
// co-running kernels (RCCL's send/recv during the overlapped gather) do to the forward pipeline.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void k_hog(unsigned long long cycles, uint32_t *sink)
{
    __shared__ uint32_t s[30 * 1024];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
        acc += s[(threadIdx.x * 7 + acc) & 1023];
        __builtin_amdgcn_s_sleep(64);
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

extern "C" int hog_launch(int nwg, double ms, void **stream_out)
{
    static hipStream_t s = nullptr;
    static uint32_t *sink = nullptr;
    if (!s) { if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1; if (hipMalloc((void **)&sink, 4) != hipSuccess) return -2; }
    const unsigned long long cycles = (unsigned long long)(ms * 1e-3 * 2.1e9);    // s_memtime runs at about the shader clock here
    hipLaunchKernelGGL(k_hog, dim3(nwg), dim3(1024), 0, s, cycles, sink);
    if (stream_out) *stream_out = s;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" int hog_wait() { return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }
