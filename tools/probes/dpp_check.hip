// wave_sum / wave_max of devmath.h against a host loop (build twice: -DSTATTN_DPP_REDUCE=0 / 1).  Every lane must receive the result.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../video-description-with-spatial-temporal-attention_amd/csrc/devmath.h"

__global__ void k(const float* x, float* s, float* m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    s[i] = stattn::wave_sum(x[i]);
    m[i] = stattn::wave_max(x[i]);
}

int main() {
    const int nw = 64, n = nw * 64;
    std::vector<float> x(n), s(n), m(n);
    unsigned r = 12345;
    for (int i = 0; i < n; ++i) { r = r * 1664525u + 1013904223u; x[i] = (float)((int)(r >> 8) % 2001 - 1000) / 64.f; }   // exact in fp32: any order gives the same sum
    float *dx, *ds, *dm;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nw / 4), dim3(256), 0, 0, dx, ds, dm);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < nw; ++w) {
        float es = 0.f, em = -INFINITY;
        for (int l = 0; l < 64; ++l) { es += x[w * 64 + l]; em = fmaxf(em, x[w * 64 + l]); }
        for (int l = 0; l < 64; ++l) if (s[w * 64 + l] != es || m[w * 64 + l] != em) { if (bad++ < 5) printf("wave %d lane %d: sum %g (want %g) max %g (want %g)\n", w, l, s[w * 64 + l], es, m[w * 64 + l], em); }
    }
    printf("STATTN_DPP_REDUCE=%d: %s (%d of %d lanes wrong)\n", STATTN_DPP_REDUCE, bad ? "FAILED" : "ok", bad, n);
    return bad != 0;
}
