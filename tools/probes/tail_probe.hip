// Last-arriver tail vs a second launch (tools only) -- VERDICT r03 item 3(b): "temporal / reduce_T as last-arriver tails of
// spatial2 / spatial_bwd; measure the real cost of 256 threads x 16 B x 8 loads in flight".
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/tail_probe.hip -o tools/bin/tail_probe
// Emulation of one decoder step's attention + temporal fuse at a given shape (rows M, frames T, hidden D, slab bytes per item):
//   stream kernel   one workgroup per (row, frame): streams its slab (HBM), writes a D-vector CL[row, frame] and three scores
//   variant A       a second launch, one workgroup per (row, 256 columns): softmax over the T scores, ctx[row] = sum_t (a_g G + a_m Mo
//                   + a_lt CL)[row, t] -- what temporal_kernel does
//   variant B       the same work done inside the first launch by the LAST workgroup of each row to finish (release fence + ticket,
//                   acquire fence, then 3 T D floats read with 8 x 16 B loads in flight per lane)
// Prints the time of both forms (HIP events, mean of 200 back-to-back step pairs) and checks that they agree.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct Args {
    const float* slab; size_t slab_floats;       // per item
    const float* G; const float* Mo;             // [M, T, D]
    float* CL; float* e;                         // [M, T, D], [3, M, T]
    float* ctx;                                  // [M, D]
    int* ticket;                                 // [M]
    int M, T, D, fused;
};

__device__ void temporal_row(const Args& a, int b, int d0, int dn, int tid, int nthreads, float* s_al) {
    // three softmaxes over T (T <= 64: one wave), then the weighted sums for columns [d0, d0 + dn)
    if (tid < 64) {
        for (int x = 0; x < 3; ++x) {
            const float v = tid < a.T ? (a.fused == 2 ? __hip_atomic_load(a.e + ((size_t)x * a.M + b) * a.T + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                      : a.e[((size_t)x * a.M + b) * a.T + tid]) : -INFINITY;
            float m = v;
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float ex = tid < a.T ? __expf(v - m) : 0.f;
            float sm = ex;
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            if (tid < a.T) s_al[x * 64 + tid] = ex / sm;
        }
    }
    __syncthreads();
    for (int d4 = tid; d4 < dn / 4; d4 += nthreads) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t base = (size_t)b * a.T * a.D + d0 + 4 * d4;
        for (int t0 = 0; t0 < a.T; t0 += 8) {                 // 8 frames x 3 tensors: 24 loads of 16 B in flight per lane
            float4 g[8], m[8], c[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = min(t0 + q, a.T - 1);
                g[q] = ld4(a.G + base + (size_t)t * a.D); m[q] = ld4(a.Mo + base + (size_t)t * a.D);
                if (a.fused == 2) {                            // variant C: what other workgroups of THIS launch wrote, read at agent scope
                    const float* cp = a.CL + base + (size_t)t * a.D;
                    c[q].x = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); c[q].y = __hip_atomic_load(cp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    c[q].z = __hip_atomic_load(cp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); c[q].w = __hip_atomic_load(cp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else c[q] = ld4(a.CL + base + (size_t)t * a.D);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (t0 + q >= a.T) break;
                const float ag = s_al[t0 + q], am = s_al[64 + t0 + q], al = s_al[128 + t0 + q];
                acc.x += ag * g[q].x + am * m[q].x + al * c[q].x; acc.y += ag * g[q].y + am * m[q].y + al * c[q].y;
                acc.z += ag * g[q].z + am * m[q].z + al * c[q].z; acc.w += ag * g[q].w + am * m[q].w + al * c[q].w;
            }
        }
        st4(a.ctx + (size_t)b * a.D + d0 + 4 * d4, acc);
    }
}

__global__ __launch_bounds__(256) void stream_kernel(const Args a) {
    __shared__ float s_al[192];
    __shared__ int s_last;
    const int bt = blockIdx.x, b = bt / a.T, t = bt % a.T, tid = threadIdx.x;
    const float* sl = a.slab + (size_t)bt * a.slab_floats;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = tid; i < a.slab_floats / 4; i += 256 * 4) {       // four 16-byte loads in flight per lane
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld4(sl + 4 * (i + q * 256 < a.slab_floats / 4 ? i + q * 256 : i));
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    if (a.fused == 2) {
        // variant C (end of round 4): no fences.  The hand-off data is written with agent-scope stores (written through to where every XCD
        // reads it), the stores are waited for, the ticket is a relaxed agent-scope add, the last arriver reads with agent-scope loads.
        for (int d4 = tid; d4 < a.D / 4; d4 += 256) {
            float* cp = a.CL + (size_t)bt * a.D + 4 * d4;
            __hip_atomic_store(cp, acc.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(cp + 1, acc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cp + 2, acc.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(cp + 3, acc.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 3) __hip_atomic_store(a.e + ((size_t)tid * a.M + b) * a.T + t, acc.x * 1e-3f + 0.1f * t * (tid + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(a.ticket + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == a.T - 1;
            if (s_last) a.ticket[b] = 0;
        }
        __syncthreads();
        if (s_last) temporal_row(a, b, 0, a.D, tid, 256, s_al);
        return;
    }
    for (int d4 = tid; d4 < a.D / 4; d4 += 256) st4(a.CL + (size_t)bt * a.D + 4 * d4, acc);
    if (tid < 3) a.e[((size_t)tid * a.M + b) * a.T + t] = acc.x * 1e-3f + 0.1f * t * (tid + 1);
    if (!a.fused) return;
    // publish: every wave's stores done -> one lane releases at agent scope -> ticket; the last arriver of the row acquires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(a.ticket + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == a.T - 1;
        if (s_last) { a.ticket[b] = 0; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    }
    __syncthreads();
    if (s_last) temporal_row(a, b, 0, a.D, tid, 256, s_al);
}

__global__ __launch_bounds__(256) void temporal_kernel(const Args a) {
    __shared__ float s_al[192];
    temporal_row(a, blockIdx.x, blockIdx.y * 256, min(256, a.D - (int)blockIdx.y * 256), threadIdx.x, 64, s_al);   // 64 float4 columns per workgroup
}

static void run(int M, int T, int D, size_t slab_bytes) {
    Args a{};
    a.M = M; a.T = T; a.D = D; a.slab_floats = slab_bytes / 4;
    const size_t items = (size_t)M * T;
    float *slab, *G, *Mo, *CL, *e, *ctxA, *ctxB, *ctxC; int* tk;
    CK(hipMalloc(&slab, items * slab_bytes)); CK(hipMalloc(&G, items * D * 4)); CK(hipMalloc(&Mo, items * D * 4)); CK(hipMalloc(&CL, items * D * 4));
    CK(hipMalloc(&e, 3 * items * 4)); CK(hipMalloc(&ctxA, (size_t)M * D * 4)); CK(hipMalloc(&ctxB, (size_t)M * D * 4)); CK(hipMalloc(&ctxC, (size_t)M * D * 4)); CK(hipMalloc(&tk, M * 4));
    std::vector<float> hb(items * D);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    CK(hipMemcpy(G, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Mo, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(slab, 0, items * slab_bytes)); CK(hipMemset(tk, 0, M * 4));
    a.slab = slab; a.G = G; a.Mo = Mo; a.CL = CL; a.e = e; a.ticket = tk;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[3];
    for (int fused = 0; fused < 3; ++fused) {
        a.fused = fused; a.ctx = fused == 2 ? ctxC : fused ? ctxB : ctxA;
        auto step = [&] {
            hipLaunchKernelGGL(stream_kernel, dim3((unsigned)items), dim3(256), 0, 0, a);
            if (!fused) hipLaunchKernelGGL(temporal_kernel, dim3(M, (D + 255) / 256), dim3(64), 0, 0, a);
        };
        for (int i = 0; i < 10; ++i) step();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; ++i) step();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[fused], e0, e1));
    }
    std::vector<float> A((size_t)M * D), B((size_t)M * D), Cc((size_t)M * D);
    CK(hipMemcpy(A.data(), ctxA, A.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(B.data(), ctxB, B.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Cc.data(), ctxC, Cc.size() * 4, hipMemcpyDeviceToHost));
    double err = 0; for (size_t i = 0; i < A.size(); ++i) err = fmax(err, fmax(fabs((double)A[i] - B[i]), fabs((double)A[i] - Cc[i])));
    printf("M=%3d T=%2d D=%4d slab %6.1f KB/item (%.0f MB per launch): two launches %6.2f us | last-arriver tail, fences %6.2f us | tail, agent-scope stores / loads, no fence %6.2f us | max |diff| %.1e\n",
           M, T, D, slab_bytes / 1024.0, items * slab_bytes / 1e6, ms[0] * 5.f, ms[1] * 5.f, ms[2] * 5.f, err);
    hipFree(slab); hipFree(G); hipFree(Mo); hipFree(CL); hipFree(e); hipFree(ctxA); hipFree(ctxB); hipFree(ctxC); hipFree(tk);
}

int main() {
    run(64, 26, 1024, 3 * 8 * 1024 * 4);      // configs[1] training step: 1664 items x 96 KB
    run(64, 26, 1024, 16 * 1024);             // the same grid with a light stream: the tail is exposed
    run(4, 26, 512, 3 * 8 * 512 * 4);         // configs[0], 4 rows
    run(1, 26, 512, 3 * 8 * 512 * 4);         // one video, greedy: 26 items
    return 0;
}
