// Ablation probe of the split GEMM (tools only):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSTATTN_PROBES [-DGS_VARIANT=n] tools/gemm_split_probe.hip -o /tmp/gs_probe && /tmp/gs_probe
// GS_VARIANT: 0 product kernel; 1 no split arithmetic (all three planes = the upper halves); 2 no LDS stores in the loop;
// 3 no global loads in the loop; 4 no MFMAs; 5 MFMAs only (no operand reads, stores or loads in the loop);
// 6 every global load re-reads the first two k-tiles (cache hits: issue cost without the memory latency);
// 8 the global loads replaced by one multiply per value (split + stores stay live, no memory instructions).
// Results of variants > 0 are wrong by construction; only the time is of interest.
#include "../video-description-with-spatial-temporal-attention_amd/csrc/gemm_split.hip"

#include <cstdio>
#include <vector>

namespace stattn {
hipError_t launch_splitk_reduce(hipStream_t, const float*, float*, int, int, int, int, float, int) { return hipSuccess; }
}
using namespace stattn;

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f);
    }
}
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    struct Shape { const char* name; int M, N, K, tA, tB; };
    const Shape shapes[] = {{"ff_local NN", 13312, 1024, 4096, 0, 0}, {"dW_local TN", 4096, 1024, 13312, 1, 0},
                            {"dL NT", 13312, 1024, 1024, 0, 1}, {"square NN", 4096, 4096, 4096, 0, 0}};
    float *A, *B, *C;
    const size_t nA = (size_t)13312 * 4096, nB = (size_t)13312 * 4096, nC = (size_t)13312 * 4096;
    CK(hipMalloc(&A, nA * 4)); CK(hipMalloc(&B, nB * 4)); CK(hipMalloc(&C, nC * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, nA, 3u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B, nB, 7u);
    CK(hipDeviceSynchronize());
    long long* clk;
    CK(hipMalloc(&clk, 4 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifndef GS_VARIANT
#define GS_VARIANT 0
#endif
    printf("variant %d\n", GS_VARIANT);
    for (const Shape& sh : shapes) {
        GemmArgs g;
        g = GemmArgs{}; g.alpha = 1.f; g.rowgroup = 1;
        g.A = A; g.lda = sh.tA ? sh.M : sh.K; g.B = B; g.ldb = sh.tB ? sh.K : sh.N; g.C = C; g.ldc = sh.N;
        g.M = sh.M; g.N = sh.N; g.K = sh.K; g.split = 1;
        g.clk = clk;
        for (int i = 0; i < 3; ++i) CK(launch_gemm_split(0, g, sh.tA, sh.tB));
        CK(hipEventRecord(e0, 0));
        const int it = 20;
        for (int i = 0; i < it; ++i) CK(launch_gemm_split(0, g, sh.tA, sh.tB));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        long long c[4];
        CK(hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost));
        const double us0 = (c[3] - c[1]) / 100.0;
        printf("  %-12s M=%6d N=%5d K=%6d  %7.3f ms  %6.1f TFLOP/s   block 0: %7.1f us at %4.0f MHz\n", sh.name, sh.M, sh.N, sh.K, ms,
               2.0 * sh.M * sh.N * sh.K / ms / 1e9, us0, us0 > 0 ? (c[2] - c[0]) / us0 : 0.0);
    }
    return 0;
}
