// Can compute-bound work run BESIDE the latency / HBM-bound reverse scan without slowing it down, when it is confined to a
// few CUs?  Probe for DESIGN.md section 11: a compute-only kernel on a CU-masked stream (hipExtStreamCreateWithCUMask, four
// CUs of every XCD) against a streaming-read kernel of the attention kernels' size on the main stream.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/cumask_probe tools/cumask_probe.hip && tools/bin/cumask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void burn(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        a0 = fmaf(a0, b, 0.5f); a1 = fmaf(a1, b, 0.5f); a2 = fmaf(a2, b, 0.5f); a3 = fmaf(a3, b, 0.5f);
        a4 = fmaf(a4, b, 0.5f); a5 = fmaf(a5, b, 0.5f); a6 = fmaf(a6, b, 0.5f); a7 = fmaf(a7, b, 0.5f);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// one workgroup per 128 KiB slab, like the attention kernels: 1664 workgroups
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ in, float* __restrict__ out, int f4_per_block) {
    const float4* p = in + (size_t)blockIdx.x * f4_per_block;
    float s = 0.f;
    for (int i = threadIdx.x; i < f4_per_block; i += 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int NB = 1664, F4 = 8192;                                  // 1664 x 128 KiB = 218 MB
    float4* in; float *o1, *o2;
    CHECK(hipMalloc(&in, (size_t)NB * F4 * 16)); CHECK(hipMemset(in, 0, (size_t)NB * F4 * 16));
    CHECK(hipMalloc(&o1, (size_t)8192 * 256 * 4)); CHECK(hipMalloc(&o2, (size_t)8192 * 256 * 4));
    hipStream_t s_main, s_masked, s_low;
    CHECK(hipStreamCreate(&s_main));
    // four CUs of every XCD under either enumeration (XCD-major or interleaved): bit i with i % 8 == i / 32
    std::vector<uint32_t> mask(8, 0u);
    int nset = 0;
    for (int i = 0; i < 256; ++i) if ((i % 8) == (i / 32)) { mask[i / 32] |= 1u << (i % 32); ++nset; }
    hipError_t em = hipExtStreamCreateWithCUMask(&s_masked, 8, mask.data());
    printf("hipExtStreamCreateWithCUMask(%d CUs): %s\n", nset, hipGetErrorString(em));
    if (em != hipSuccess) return 1;
    int lo, hi;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithPriority(&s_low, hipStreamNonBlocking, lo));
    printf("priority range: least %d greatest %d\n", lo, hi);
    hipEvent_t e0, e1, b0, b1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
    auto time_stream = [&](int reps, float* ms) -> int {
        CHECK(hipEventRecord(e0, s_main));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream, dim3(NB), dim3(256), 0, s_main, in, o1, F4);
        CHECK(hipEventRecord(e1, s_main));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(ms, e0, e1));
        *ms /= reps;
        return 0;
    };
    float ms;
    if (time_stream(30, &ms)) return 1;
    if (time_stream(30, &ms)) return 1;
    printf("stream kernel alone: %.1f us per launch (%.2f TB/s)\n", ms * 1e3, NB * F4 * 16.0 / (ms * 1e-3) / 1e12);
    // the compute kernel: 2048 workgroups, sized to ~0.25 ms on the whole chip
    const int iters = 6000;
    auto time_burn = [&](hipStream_t s, int nblocks, float* msb) -> int {
        CHECK(hipEventRecord(b0, s));
        hipLaunchKernelGGL(burn, dim3(nblocks), dim3(256), 0, s, o2, iters);
        CHECK(hipEventRecord(b1, s));
        CHECK(hipEventSynchronize(b1));
        CHECK(hipEventElapsedTime(msb, b0, b1));
        return 0;
    };
    float mb;
    if (time_burn(s_main, 2048, &mb)) return 1;
    if (time_burn(s_main, 2048, &mb)) return 1;
    printf("burn, 2048 workgroups, whole chip: %.3f ms\n", mb);
    if (time_burn(s_masked, 2048, &mb)) return 1;
    printf("burn, 2048 workgroups, masked stream: %.3f ms\n", mb);
    // concurrent: burn in the background, 30 stream launches in front
    struct Bg { const char* name; hipStream_t s; } bgs[] = {{"masked stream (32 CUs)", s_masked}, {"low-priority stream (whole chip)", s_low}};
    for (const Bg& bg : bgs) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(b0, bg.s));
        hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, bg.s, o2, iters);
        CHECK(hipEventRecord(b1, bg.s));
        float m1;
        if (time_stream(30, &m1)) return 1;
        const bool still = hipEventQuery(b1) == hipErrorNotReady;
        CHECK(hipEventSynchronize(b1));
        CHECK(hipEventElapsedTime(&mb, b0, b1));
        printf("stream kernel beside burn on the %s: %.1f us per launch (30 launches = %.2f ms; burn took %.3f ms, %s when the 30 were done)\n",
               bg.name, m1 * 1e3, m1 * 30, mb, still ? "still running" : "finished");
    }
    return 0;
}
