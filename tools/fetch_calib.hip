// FETCH_SIZE calibration (tools only).  The guide's gfx950 note -- FETCH_SIZE reports HALF the bytes of a
// wide coalesced streaming read -- is calibrated on 16-byte-per-lane linear streams.  The row-panel kernels read their
// weights differently: a workgroup owns one contiguous 64 KiB panel, its eight waves take every eighth 1 KiB k-step.
// Each kernel below reads a KNOWN number of bytes exactly once, from a part of a 2 GiB buffer nothing has touched since
// it was written (cold in L2); run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- tools/bin/fetch_calib
// and compare the counter (KiB) with the byte count printed here: the ratio is the correction for that access pattern.
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// (a) linear stream: thread i reads float4 i, i + stride, ...
__global__ __launch_bounds__(256) void calib_stream16(const float* __restrict__ src, float* __restrict__ dst, size_t n4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ld4(src + 4 * i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc.x;
}
// (b) the row-panel walk (panel.hip): workgroup c owns tile c = S k-steps x 1 KiB, wave ks takes k-steps ks, ks + 8, ...
__global__ __launch_bounds__(512) void calib_panel_walk(const float* __restrict__ src, float* __restrict__ dst, int S) {
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const float* tile = src + (size_t)blockIdx.x * S * 256 + 4 * lane;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = ks; s < S; s += 8) {
        const float4 v = ld4(tile + (size_t)s * 256);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc.x;
}
// (c) the attention walk (attn.hip spatial2): workgroup (b,t) of 128 threads reads K rows of D floats, two column groups per lane
__global__ __launch_bounds__(128) void calib_slab_walk(const float* __restrict__ src, float* __restrict__ dst, int K, int D) {
    const float* slab = src + (size_t)blockIdx.x * K * D;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int d4 = threadIdx.x; d4 < D / 4; d4 += 128)
        for (int k = 0; k < K; ++k) {
            const float4 v = ld4(slab + (size_t)k * D + 4 * d4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc.x;
}
// (d) 4-byte-per-lane stream (256 B per wave instruction): the "other widths are uncalibrated" case
__global__ __launch_bounds__(256) void calib_stream4(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 12345.678f) dst[0] = acc;
}

int main() {
    const size_t total = (size_t)2 << 30;
    float* buf; float* dst;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&dst, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, total);
    hipDeviceSynchronize();
    size_t off = 0;                                    // floats; every launch reads a fresh region
    auto next = [&](size_t bytes) { const float* p = buf + off; off += bytes / 4; return p; };
    for (int rep = 0; rep < 3; ++rep) {
        {   const size_t bytes = (size_t)64 << 20;     // 64 MiB linear
            hipLaunchKernelGGL(calib_stream16, dim3(4096), dim3(256), 0, 0, next(bytes), dst, bytes / 16);
            printf("calib_stream16    %10zu bytes\n", bytes); }
        {   const int S = 64, tiles = 256;             // lstm_panel_kernel at configs[1]: 256 tiles x 64 k-steps x 1 KiB = 16 MiB
            const size_t bytes = (size_t)tiles * S * 1024;
            hipLaunchKernelGGL(calib_panel_walk, dim3(tiles), dim3(512), 0, 0, next(bytes), dst, S);
            printf("calib_panel_walk  %10zu bytes\n", bytes); }
        {   const int K = 8, D = 1024, items = 1664 * 3;   // spatial2 at configs[1]: 3 slabs of 32 KiB per (b,t)
            const size_t bytes = (size_t)items * K * D * 4;
            hipLaunchKernelGGL(calib_slab_walk, dim3(items), dim3(128), 0, 0, next(bytes), dst, K, D);
            printf("calib_slab_walk   %10zu bytes\n", bytes); }
        {   const size_t bytes = (size_t)64 << 20;
            hipLaunchKernelGGL(calib_stream4, dim3(4096), dim3(256), 0, 0, next(bytes), dst, bytes / 4);
            printf("calib_stream4     %10zu bytes\n", bytes); }
    }
    hipDeviceSynchronize();
    printf("regions used: %.1f MiB of %zu\n", off * 4.0 / (1 << 20), total >> 20);
    return 0;
}
