// Infinity Cache (MALL) probe (tools only): streaming read bandwidth of a float4 reduction over working sets of
// 64 .. 512 MB, re-read back to back.  Shows whether a working set under 256 MB is served faster than HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o tools/mall_probe && ./tools/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float* __restrict__ dst, size_t n4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    for (; i < n4; i += stride) { const float4 a = src[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc.x;
}
int main() {
    const size_t maxb = (size_t)768 << 20;
    float* buf; float* dst;
    hipMalloc(&buf, maxb); hipMalloc(&dst, 64);
    hipMemset(buf, 0, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {32, 64, 128, 164, 200, 240, 300, 400, 512, 768}) {
        const size_t n4 = (mb << 20) / 16;
        for (int blocks : {2048, 8192}) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, dst, n4);
            hipEventRecord(e0, 0);
            const int it = 20;
            for (int i = 0; i < it; ++i) hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, dst, n4);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("working set %4zu MB  blocks %5d  %7.1f us per pass  %7.2f TB/s\n", mb, blocks, ms / it * 1e3, (double)(mb << 20) / (ms / it * 1e-3) / 1e12);
        }
    }
    return 0;
}
