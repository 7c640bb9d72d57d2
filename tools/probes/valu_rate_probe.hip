// Issue rate of the VALU instructions spatial_shared_kernel's score phase is made of (v_fma_f32, v_pk_fma_f32, v_rcp_f32,
// v_exp_f32 and the phase's own mix), per SIMD, at 1 .. 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float v[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{v[i], v[i] + 0.5f}; }
    const float a = 1.0001f, b = 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{a, a}), "v"(f2{b, b}));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 4) {   // the score phase's triple: fma, rcp, fma
                float t;
                asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(t) : "v"(v[i]), "v"(a));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(t), "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
static void run(const char* name, float* d, int per_inst) {
    for (int wps = 1; wps <= 4; ++wps) {
        const int grid = 256 * wps, iters = 4096;      // one 256-thread workgroup = one wave per SIMD of a CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)iters * 8 * per_inst * wps;          // wave instructions per SIMD
        printf("%-10s waves/SIMD %d: %.3f ms, %.2f ns per wave instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, wps, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
    }
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4 * 4);
    run<0>("v_fma", d, 1); run<1>("v_pk_fma", d, 1); run<2>("v_rcp", d, 1); run<3>("v_exp", d, 1); run<4>("fma,rcp,fma", d, 3);
    return 0;
}
