// Small memory-bound helper kernels of the decoder path (gfx950).
#include "kernels.h"
#include "devmath.h"

namespace stattn {

namespace {

__global__ void fill_kernel(float* __restrict__ p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void iota_kernel(int* __restrict__ p, int n, int mul) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * mul;
}

// mean[b,:] = sum_t G[b,t,:] / sum_t mask[b,t]      (model_attention.py:618+649, 739+766)
__global__ __launch_bounds__(256) void ctx_mean_kernel(const float* __restrict__ G, const float* __restrict__ mask,
                                                       float* __restrict__ mean, int T, int D) {
    const int b = blockIdx.x;
    float cnt = 0.f;
#pragma unroll 8
    for (int t = 0; t < T; ++t) cnt += mask[(size_t)b * T + t];
    const float inv = 1.0f / cnt;
    for (int d = blockIdx.y * 256 + threadIdx.x; d < D; d += gridDim.y * 256) {
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < T; ++t) s += G[((size_t)b * T + t) * D + d];
        mean[(size_t)b * D + d] = s * inv;
    }
}

// word embedding lookup.  Training (:613-617): row r of the (t*m) grid reads word x[r - shift]
// (shift = m: "shift forward in time"), rows < shift are zero.  Sampling (:803-804): shift = 0 and
// a negative index (-1 = first word) gives the zero vector.
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ x, const float* __restrict__ Wemb,
                                                    float* __restrict__ emb, int rows, int E, int V, int shift,
                                                    float* __restrict__ emb_pk) {
    const int r = blockIdx.x;
    int64_t w = -1;
    if (r >= shift) w = x[r - shift];
    if (w >= V) w = V - 1;   // defensive: never read out of bounds (Theano would raise IndexError)
    for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w >= 0) v = ld4(Wemb + (size_t)w * E + 4 * e4);
        st4(emb + (size_t)r * E + 4 * e4, v);
        if (emb_pk) st4(emb_pk + pn_pack_offset(r, 4 * e4, E >> 4), v);      // packed A layout of the row-panel kernels
    }
}

// row softmax (max-subtracted, :708-709 / :840), optional NLL -log(p[x]+1e-8) (:712) and argmax
__global__ __launch_bounds__(256) void softmax_nll_kernel(const float* __restrict__ logits, int ldl,
                                                          float* __restrict__ probs, int ldp,
                                                          const int64_t* __restrict__ x, float* __restrict__ nll,
                                                          int64_t* __restrict__ argmax, int V) {
    __shared__ float s_f[4];
    __shared__ int s_i[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* __restrict__ lg = logits + (size_t)r * ldl;
    float mx = -INFINITY; int mi = 0;
    for (int j = tid; j < V; j += 256) { const float v = lg[j]; if (v > mx) { mx = v; mi = j; } }
    // wave arg-max (ties -> lowest index, like numpy.argmax)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    if (lane == 0) { s_f[w] = mx; s_i[w] = mi; }
    __syncthreads();
    mx = s_f[0]; mi = s_i[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) if (s_f[i] > mx || (s_f[i] == mx && s_i[i] < mi)) { mx = s_f[i]; mi = s_i[i]; }
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < V; j += 256) sum += __expf(lg[j] - mx);
    sum = wave_sum(sum);
    if (lane == 0) s_f[w] = sum;
    __syncthreads();
    sum = s_f[0] + s_f[1] + s_f[2] + s_f[3];
    const float inv = 1.0f / sum;
    float* __restrict__ pr = probs + (size_t)r * ldp;
    for (int j = tid; j < V; j += 256) pr[j] = __expf(lg[j] - mx) * inv;
    if (tid == 0) {
        if (argmax) argmax[r] = mi;
        if (nll) {
            int64_t xi = x[r];
            xi = xi < 0 ? 0 : (xi >= V ? V - 1 : xi);
            nll[r] = -logf(__expf(lg[xi] - mx) * inv + 1e-8f);
        }
    }
}

// Same, with the row held in registers (one pass over the logits instead of three dependent ones) when
// V <= NT * MAXV: 256 threads x 48 values for the (t*m)-row training softmax, 1024 threads x 16 for the few rows of a
// decode step, where the kernel is a chain of load latencies rather than bandwidth.
template <int NT, int MAXV>
__global__ __launch_bounds__(NT) void softmax_nll_reg_kernel(const float* __restrict__ logits, int ldl,
                                                             float* __restrict__ probs, int ldp,
                                                             const int64_t* __restrict__ x, float* __restrict__ nll,
                                                             int64_t* __restrict__ argmax, int V) {
    constexpr int NW = NT / 64;
    __shared__ float s_f[NW];
    __shared__ int s_i[NW];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* __restrict__ lg = logits + (size_t)r * ldl;
    float v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { const int j = tid + i * NT; v[i] = j < V ? lg[j] : -INFINITY; }
    float mx = -INFINITY; int mi = 0;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) if (v[i] > mx) { mx = v[i]; mi = tid + i * NT; }     // ascending j: first maximum wins
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    if (lane == 0) { s_f[w] = mx; s_i[w] = mi; }
    __syncthreads();
    mx = s_f[0]; mi = s_i[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) if (s_f[i] > mx || (s_f[i] == mx && s_i[i] < mi)) { mx = s_f[i]; mi = s_i[i]; }
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { v[i] = __expf(v[i] - mx); sum += v[i]; }            // exp(-inf) = 0 for the padding
    sum = wave_sum(sum);
    if (lane == 0) s_f[w] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) sum += s_f[i];
    const float inv = 1.0f / sum;
    float* __restrict__ pr = probs + (size_t)r * ldp;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { const int j = tid + i * NT; if (j < V) pr[j] = v[i] * inv; }
    if (tid == 0) {
        if (argmax) argmax[r] = mi;
        if (nll) {
            int64_t xi = x[r];
            xi = xi < 0 ? 0 : (xi >= V ? V - 1 : xi);
            nll[r] = -logf(__expf(lg[xi] - mx) * inv + 1e-8f);
        }
    }
}

// cost[b] = sum_t mask[t,b] * nll[t,b]     (:714-715)
__global__ void cost_kernel(const float* __restrict__ nll, const float* __restrict__ mask,
                            float* __restrict__ cost, int t, int m) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= m) return;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < t; ++i) s += mask[(size_t)i * m + b] * nll[(size_t)i * m + b];      // (unrolled: the loads of eight steps in flight)
    cost[b] = s;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Bernoulli(0.5) per element from a counter-based hash: 64 draws per hash
__global__ void bernoulli_kernel(float* __restrict__ p, size_t n, uint64_t seed, uint64_t stream_id) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(mix64(seed ^ (stream_id * 0x9E3779B97F4A7C15ull)) + (i >> 6));
        p[i] = (float)((h >> (i & 63)) & 1ull);
    }
}

// the three dropout masks of a pass (LSTM gates, h, readout) in one launch: segment y draws stream stream0 + y, element
// for element what three bernoulli_kernel launches draw
struct Bern3 { float* p[3]; size_t n[3]; };
__global__ void bernoulli3_kernel(const Bern3 b, uint64_t seed, uint64_t stream0) {
    const int y = blockIdx.y;
    float* __restrict__ p = b.p[y];
    const size_t n = b.n[y];
    const uint64_t key = mix64(seed ^ ((stream0 + y) * 0x9E3779B97F4A7C15ull));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(key + (i >> 6));
        p[i] = (float)((h >> (i & 63)) & 1ull);
    }
}

// uniform in [-1, 1) from the same hash (bench / timing data: full-range signs, never zeros)
__global__ void uniform_kernel(float* __restrict__ p, size_t n, uint64_t seed, uint64_t stream_id) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(mix64(seed ^ (stream_id * 0x9E3779B97F4A7C15ull)) + i);
        p[i] = (float)(h >> 40) * (2.0f / 16777216.0f) - 1.0f;
    }
}

inline int grid_for(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

hipError_t launch_fill(hipStream_t s, float* p, float v, size_t n) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}
hipError_t launch_iota(hipStream_t s, int* p, int n, int mul) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n, mul);
    return hipGetLastError();
}
hipError_t launch_ctx_mean(hipStream_t s, const float* G, const float* mask, float* mean, int B, int T, int D) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(ctx_mean_kernel, dim3(B, (D + 255) / 256), dim3(256), 0, s, G, mask, mean, T, D);
    return hipGetLastError();
}
hipError_t launch_embed(hipStream_t s, const int64_t* x, const float* Wemb, float* emb, int rows, int E, int V, int shift, float* emb_pk) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(256), 0, s, x, Wemb, emb, rows, E, V, shift, emb_pk);
    return hipGetLastError();
}
hipError_t launch_softmax_nll(hipStream_t s, const float* logits, int ldl, float* probs, int ldp,
                              const int64_t* x, float* nll, int64_t* argmax, int rows, int V) {
    if (rows <= 0) return hipSuccess;
    if (rows <= 256 && V <= 1024 * 16)
        hipLaunchKernelGGL((softmax_nll_reg_kernel<1024, 16>), dim3(rows), dim3(1024), 0, s, logits, ldl, probs, ldp, x, nll, argmax, V);
    else if (V <= 256 * 48)
        hipLaunchKernelGGL((softmax_nll_reg_kernel<256, 48>), dim3(rows), dim3(256), 0, s, logits, ldl, probs, ldp, x, nll, argmax, V);
    else
        hipLaunchKernelGGL(softmax_nll_kernel, dim3(rows), dim3(256), 0, s, logits, ldl, probs, ldp, x, nll, argmax, V);
    return hipGetLastError();
}
hipError_t launch_cost(hipStream_t s, const float* nll, const float* mask, float* cost, int t, int m) {
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(cost_kernel, dim3((m + 63) / 64), dim3(64), 0, s, nll, mask, cost, t, m);
    return hipGetLastError();
}
hipError_t launch_bernoulli(hipStream_t s, float* p, size_t n, uint64_t seed, uint64_t stream_id) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(bernoulli_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, n, seed, stream_id);
    return hipGetLastError();
}

hipError_t launch_bernoulli3(hipStream_t s, float* p0, size_t n0, float* p1, size_t n1, float* p2, size_t n2, uint64_t seed, uint64_t stream0) {
    Bern3 b{{p0, p1, p2}, {n0, n1, n2}};
    size_t nmax = n0 > n1 ? n0 : n1; nmax = nmax > n2 ? nmax : n2;
    if (!nmax) return hipSuccess;
    hipLaunchKernelGGL(bernoulli3_kernel, dim3(grid_for(nmax, 256), 3), dim3(256), 0, s, b, seed, stream0);
    return hipGetLastError();
}

hipError_t launch_uniform(hipStream_t s, float* p, size_t n, uint64_t seed, uint64_t stream_id) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(uniform_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, n, seed, stream_id);
    return hipGetLastError();
}

}  // namespace stattn
