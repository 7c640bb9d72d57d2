// MFMA issue-rate probe (tools only): v_mfma_f32_32x32x2_f32 / 16x16x4 streams from W waves per SIMD, NACC independent
// accumulators, 256 workgroups; prints cycles per MFMA per SIMD from the event time at an assumed 2.4 GHz and from the
// shader clock counter read in the kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, int iters, long long* clk) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.f;
    const long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    const long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(float* out, int iters, long long* clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.f;
    const long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    const long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <class K>
static void run(const char* name, K kern, int waves, int nacc, int flops_per, int nblk = 256) {
    float* out; long long* clk;
    (void)hipMalloc(&out, (size_t)nblk * 64 * waves * 4); (void)hipMalloc(&clk, 16);
    const int iters = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * waves), 0, 0, out, iters, clk);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, 16, (void)hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * 4 * nacc * (waves / 4.0) * (nblk / 256.0);
    const double tf = (double)nblk * waves * iters * 4 * nacc * flops_per / (ms * 1e-3) / 1e12;
    printf("%-28s %2d waves/WG, %d acc, %d WGs: %7.1f us  %6.1f TFLOP/s | in-kernel: %lld counter ticks over %.2f us wall = %.0f MHz; %.1f ticks per MFMA per SIMD\n",
           name, waves, nacc, nblk, ms * 1e3, tf, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), h[0] / n_per_simd);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    run("32x32x2 1 wave/SIMD", k32<5, 4>, 4, 5, 4096);
    run("32x32x2 1 wave/SIMD", k32<2, 4>, 4, 2, 4096);
    run("32x32x2 1 wave/SIMD", k32<1, 4>, 4, 1, 4096);
    run("32x32x2 2 waves/SIMD", k32<5, 8>, 8, 5, 4096);
    run("32x32x2 2 waves/SIMD", k32<2, 8>, 8, 2, 4096);
    run("32x32x2 2 WGs/CU", k32<5, 4>, 4, 5, 4096, 512);
    run("16x16x4 1 wave/SIMD", k16<8, 4>, 4, 8, 2048);
    run("16x16x4 2 waves/SIMD", k16<8, 8>, 8, 8, 2048);
    run("16x16x4 4 waves/SIMD", k16<4, 16>, 16, 4, 2048);
    return 0;
}
