// Attention kernels of the decoder step for gfx950 -- the HBM-bound part of the path.
//
// spatial_kernel  (model_attention.py:371-383, plus the score halves of :389-398, :402-411 and,
//                  in lt_mode 1, :415-425):   one workgroup per (row b, frame t)
//     e_k   = Ul . tanh(PL[v,t,k,:] + sl[b,:]) + cl          k = 0..K-1
//     alpha = softmax_k(e)
//     CL[b,t,:] = sum_k alpha_k L[v,t,k,:]
//     eg[b,t] = Ug . tanh(PG[v,t,:] + sg[b,:]) + cg           (same for motion)
//     lt_mode 1:  elt[b,t] = Ult . tanh(sum_k alpha_k LW[v,t,k,:] + blt + slt[b,:]) + clt
//   Every byte of PL / L / LW is read exactly once per step with 16-byte coalesced loads
//   (a K x D slab is contiguous), D is striped over the 256 lanes of the workgroup, the K+2
//   dot products are reduced with wave shuffles + one LDS hop, the K-wide softmax lives in LDS.
//
// temporal_kernel (:398-399, :411-412, :425-435): one workgroup per (row b, 256-wide slice of D)
//     three softmaxes over T (unmasked, Appendix C.5), selector gate, and
//     ctx[b,:] = sel_b * sum_t (ag_t G[v,t,:] + am_t M[v,t,:] + alt_t CL[b,t,:])
#include "kernels.h"

#include <type_traits>

#include "devmath.h"
#include "panel_inl.h"
#include "beam_inl.h"

#include <cstdlib>

namespace stattn {

namespace {

constexpr int KMAX = 64;     // regions per frame supported by the LDS softmax
constexpr int TMAX = 256;    // frames supported by the temporal kernel

__device__ __forceinline__ float dot4_tanh(float4 x, float4 s, float4 u) {
    return fast_tanh(x.x + s.x) * u.x + fast_tanh(x.y + s.y) * u.y +
           fast_tanh(x.z + s.z) * u.z + fast_tanh(x.w + s.w) * u.w;
}

// sum `n` per-thread values over the 256-thread workgroup; result broadcast through out[]
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* s_red /*[4][N]*/, int tid) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float r = wave_sum(v[i]);
        if (lane == 0) s_red[w * N + i] = r;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = s_red[i] + s_red[N + i] + s_red[2 * N + i] + s_red[3 * N + i];
    __syncthreads();
}

__global__ __launch_bounds__(256) void spatial_kernel(const SpatialArgs a) {
    __shared__ float s_red[4 * 10];
    __shared__ float s_e[KMAX];
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt ? a.clt[0] : 0.f;   // scalar loads up front, not inside the one-lane branches that use them
    const int bt = blockIdx.x, b = bt / T, t = bt % T;
    const int v = a.vid ? a.vid[b] : b;
    const int tid = threadIdx.x;
    const size_t slab = ((size_t)v * T + t) * K * D;
    const float* __restrict__ PL = a.PL + slab;
    const float* __restrict__ L = a.L + slab;
    const float* __restrict__ sl = a.sproj + (size_t)b * a.ldsp;
    const int nd4 = D >> 2;

    // ---- scores: K region scores (+ the two frame scores with the first group of 8)
    for (int k0 = 0; k0 < K; k0 += 8) {
        float p[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) p[i] = 0.f;
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            const float4 s4 = ld4(sl + 4 * d4), u4 = ld4(a.Ul + 4 * d4);
            float4 x[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)   // clamp keeps the 8 loads unconditional and in flight together
                x[kk] = ld4(PL + (size_t)min(k0 + kk, K - 1) * D + 4 * d4);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) p[kk] += dot4_tanh(x[kk], s4, u4);
            if (k0 == 0) {
                const size_t fo = ((size_t)v * T + t) * D + 4 * d4;
                p[8] += dot4_tanh(ld4(a.PG + fo), ld4(sl + D + 4 * d4), ld4(a.Ug + 4 * d4));
                p[9] += dot4_tanh(ld4(a.PM + fo), ld4(sl + 2 * D + 4 * d4), ld4(a.Um + 4 * d4));
            }
        }
        block_sum<10>(p, s_red, tid);
        if (tid < 8 && k0 + tid < K) s_e[k0 + tid] = p[tid] + cl0;
        if (k0 == 0 && tid == 8) a.eg[bt] = p[8] + cg0;
        if (k0 == 0 && tid == 9) a.em[bt] = p[9] + cm0;
    }
    __syncthreads();

    // ---- softmax over the K regions (every thread redundantly; K <= 64, LDS broadcast reads)
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_e[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(s_e[k] - mx);
    const float inv = 1.0f / sum;
    __syncthreads();
    if (tid < K) {
        const float al = __expf(s_e[tid] - mx) * inv;
        a.alphal[(size_t)bt * K + tid] = al;
        s_e[tid] = al;
    }
    __syncthreads();

    // ---- attended local feature (and, lt_mode 1, the local-temporal score)
    float pe[1] = {0.f};
    const float* __restrict__ LW = a.LW ? a.LW + slab : nullptr;
    for (int d4 = tid; d4 < nd4; d4 += 256) {
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        // (the loop is written once per case: with `if (LW)` around the second load inside it, hipcc kept the branch and put
        // a full wait behind every LW row -- K dependent round trips per item)
        if (LW) {
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const float al = s_e[k];
                const float4 l4 = ld4(L + (size_t)k * D + 4 * d4);
                const float4 q4 = ld4(LW + (size_t)k * D + 4 * d4);
                c4.x += al * l4.x; c4.y += al * l4.y; c4.z += al * l4.z; c4.w += al * l4.w;
                w4.x += al * q4.x; w4.y += al * q4.y; w4.z += al * q4.z; w4.w += al * q4.w;
            }
        } else {
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const float al = s_e[k];
                const float4 l4 = ld4(L + (size_t)k * D + 4 * d4);
                c4.x += al * l4.x; c4.y += al * l4.y; c4.z += al * l4.z; c4.w += al * l4.w;
            }
        }
        st4(a.CL + (size_t)bt * D + 4 * d4, c4);
        if (LW) {
            const float4 bl = ld4(a.blt + 4 * d4);
            w4.x += bl.x; w4.y += bl.y; w4.z += bl.z; w4.w += bl.w;
            pe[0] += dot4_tanh(w4, ld4(sl + 3 * D + 4 * d4), ld4(a.Ult + 4 * d4));
        }
    }
    if (LW) {
        block_sum<1>(pe, s_red, tid);
        if (tid == 0) a.elt[bt] = pe[0] + clt0;
    }
}

// ---- bf16 storage variant (precision = bf16 handles): PL / L / LW are bf16, 8 values per 16-byte load, all
// arithmetic in fp32.  Halves the HBM traffic of the step's dominant kernel.  NT threads cover D / 8 lanes.
__device__ __forceinline__ void bf8_to_f32(const uint4 v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }

template <int N, int NW>
__device__ __forceinline__ void block_sum_w(float (&v)[N], float* s_red /*[NW][N]*/, int tid) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float r = wave_sum(v[i]);
        if (lane == 0) s_red[w * N + i] = r;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += s_red[q * N + i];
        v[i] = t;
    }
    __syncthreads();
}

// fp32 variant with NT threads per workgroup and TWO d4 columns per thread (both sets of loads in flight together).
// Purpose: spatial_kernel needs 75 VGPRs -> 6 workgroups of 256 per CU -> 1536 resident for the 1664 (b,t) items of
// configs[1]: an 8 % second round.  128-thread workgroups are all resident at once.
template <int NT>
__global__ __launch_bounds__(NT) void spatial2_kernel(const SpatialArgs a) {
    constexpr int NW = NT / 64;
    __shared__ float s_red[NW * 10];
    __shared__ float s_e[KMAX];
    // the first rider.nblocks workgroups compute the rider GEMM (h.U of this step) instead of an attention item
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[NW * 64 * 16];
        rider_tile<NW>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt ? a.clt[0] : 0.f;   // scalar loads up front, not inside the one-lane branches that use them
    const int bt = xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T), b = bt / T, t = bt % T;   // a row's frames on one XCD (shared sproj row)
    const int v = a.vid ? a.vid[b] : b;
    const int tid = threadIdx.x;
    const size_t slab = ((size_t)v * T + t) * K * D;
    const float* __restrict__ PL = a.PL + slab;
    const float* __restrict__ L = a.L + slab;
    const float* __restrict__ sl = a.sproj + (size_t)b * a.ldsp;
    const int nd4 = D >> 2;

    for (int k0 = 0; k0 < K; k0 += 8) {
        float p[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) p[i] = 0.f;
        for (int da = tid; da < nd4; da += 2 * NT) {
            const int db = da + NT < nd4 ? da + NT : da;          // second column (clamped: weight 0 below)
            const float wb = da + NT < nd4 ? 1.f : 0.f;
            float4 xa[8], xb[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const size_t ro = (size_t)min(k0 + kk, K - 1) * D;
                xa[kk] = ld4(PL + ro + 4 * da);
                xb[kk] = ld4(PL + ro + 4 * db);
            }
            const float4 sa = ld4(sl + 4 * da), ua = ld4(a.Ul + 4 * da);
            const float4 sb = ld4(sl + 4 * db), ub = ld4(a.Ul + 4 * db);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) p[kk] += dot4_tanh(xa[kk], sa, ua) + wb * dot4_tanh(xb[kk], sb, ub);
            if (k0 == 0) {
                const size_t fa = ((size_t)v * T + t) * D + 4 * da, fb = ((size_t)v * T + t) * D + 4 * db;
                p[8] += dot4_tanh(ld4(a.PG + fa), ld4(sl + D + 4 * da), ld4(a.Ug + 4 * da)) +
                        wb * dot4_tanh(ld4(a.PG + fb), ld4(sl + D + 4 * db), ld4(a.Ug + 4 * db));
                p[9] += dot4_tanh(ld4(a.PM + fa), ld4(sl + 2 * D + 4 * da), ld4(a.Um + 4 * da)) +
                        wb * dot4_tanh(ld4(a.PM + fb), ld4(sl + 2 * D + 4 * db), ld4(a.Um + 4 * db));
            }
        }
        block_sum_w<10, NW>(p, s_red, tid);
        if (tid < 8 && k0 + tid < K) s_e[k0 + tid] = p[tid] + cl0;
        if (k0 == 0 && tid == 8) a.eg[bt] = p[8] + cg0;
        if (k0 == 0 && tid == 9) a.em[bt] = p[9] + cm0;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_e[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(s_e[k] - mx);
    const float inv = 1.0f / sum;
    __syncthreads();
    if (tid < K) {
        const float al = __expf(s_e[tid] - mx) * inv;
        a.alphal[(size_t)bt * K + tid] = al;
        s_e[tid] = al;
    }
    __syncthreads();

    float pe[1] = {0.f};
    const float* __restrict__ LW = a.LW ? a.LW + slab : nullptr;
    for (int d4 = tid; d4 < nd4; d4 += NT) {
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        // (the loop is written once per case: with `if (LW)` around the second load inside it, hipcc kept the branch and put
        // a full wait behind every LW row -- K dependent round trips per item)
        if (LW) {
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const float al = s_e[k];
                const float4 l4 = ld4(L + (size_t)k * D + 4 * d4);
                const float4 q4 = ld4(LW + (size_t)k * D + 4 * d4);
                c4.x += al * l4.x; c4.y += al * l4.y; c4.z += al * l4.z; c4.w += al * l4.w;
                w4.x += al * q4.x; w4.y += al * q4.y; w4.z += al * q4.z; w4.w += al * q4.w;
            }
        } else {
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const float al = s_e[k];
                const float4 l4 = ld4(L + (size_t)k * D + 4 * d4);
                c4.x += al * l4.x; c4.y += al * l4.y; c4.z += al * l4.z; c4.w += al * l4.w;
            }
        }
        st4(a.CL + (size_t)bt * D + 4 * d4, c4);
        if (LW) {
            const float4 bl = ld4(a.blt + 4 * d4);
            w4.x += bl.x; w4.y += bl.y; w4.z += bl.z; w4.w += bl.w;
            pe[0] += dot4_tanh(w4, ld4(sl + 3 * D + 4 * d4), ld4(a.Ult + 4 * d4));
        }
    }
    if (LW) {
        block_sum_w<1, NW>(pe, s_red, tid);
        if (tid == 0) a.elt[bt] = pe[0] + clt0;
    }
}

// Small grids (decode: a handful of rows; M T workgroups do not even fill the chip): latency is all that counts, so
// a lane requests EVERYTHING it will ever need -- its column of PL, L and LW for all K <= 8 regions, 24 float4 -- in one
// burst and the kernel is a single memory round trip (the kernels above wait for the softmax before they ask for L / LW).
// One float4 column per lane: D <= 4 * blockDim.
__device__ __forceinline__ void spatial_small_body(const SpatialArgs& a) {
    __shared__ float s_red[16 * 10];       // (up to sixteen waves: spatial_small_update_wide_kernel)
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt ? a.clt[0] : 0.f;   // scalar loads up front, not inside the one-lane branches that use them
    const int bt = blockIdx.x, b = bt / T, t = bt % T;
    const int v = a.vid ? a.vid[b] : b;
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    const int d4 = min(tid, (D >> 2) - 1);
    const float on = tid < (D >> 2) ? 1.f : 0.f;
    const size_t slab = ((size_t)v * T + t) * K * D, fo = ((size_t)v * T + t) * D + 4 * d4;
    const float* __restrict__ sl = a.sproj + (size_t)b * a.ldsp;
    float4 pl[8], l[8], lw[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const size_t o = slab + (size_t)min(kk, K - 1) * D + 4 * d4;
        pl[kk] = ld4(a.PL + o); l[kk] = ld4(a.L + o);
        lw[kk] = a.LW ? ld4(a.LW + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 pg = ld4(a.PG + fo), pm = ld4(a.PM + fo);
    const float4 s0 = ld4(sl + 4 * d4), s1 = ld4(sl + D + 4 * d4), s2 = ld4(sl + 2 * D + 4 * d4), s3 = ld4(sl + 3 * D + 4 * d4);
    float p[10];
    {
        const float4 u4 = ld4(a.Ul + 4 * d4);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) p[kk] = on * dot4_tanh(pl[kk], s0, u4);
        p[8] = on * dot4_tanh(pg, s1, ld4(a.Ug + 4 * d4));
        p[9] = on * dot4_tanh(pm, s2, ld4(a.Um + 4 * d4));
    }
    {   // block_sum over nw waves
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int i = 0; i < 10; ++i) { const float r = wave_sum(p[i]); if (lane == 0) s_red[w * 10 + i] = r; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 10; ++i) { float r = 0.f; for (int q = 0; q < nw; ++q) r += s_red[q * 10 + i]; p[i] = r; }
    }
    if (tid == 0) { a.eg[bt] = p[8] + cg0; a.em[bt] = p[9] + cm0; }
    // softmax over the K regions (every lane redundantly, from registers)
    float mx = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { p[kk] += cl0; if (kk < K) mx = fmaxf(mx, p[kk]); }
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { p[kk] = kk < K ? __expf(p[kk] - mx) : 0.f; sum += p[kk]; }
    const float inv = 1.0f / sum;
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = c4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const float al = p[kk] * inv;
        if (tid == kk && kk < K) a.alphal[(size_t)bt * K + kk] = al;
        c4.x += al * l[kk].x; c4.y += al * l[kk].y; c4.z += al * l[kk].z; c4.w += al * l[kk].w;
        w4.x += al * lw[kk].x; w4.y += al * lw[kk].y; w4.z += al * lw[kk].z; w4.w += al * lw[kk].w;
    }
    if (on != 0.f) st4(a.CL + (size_t)bt * D + 4 * d4, c4);
    if (a.LW) {
        const float4 bl = ld4(a.blt + 4 * d4);
        w4.x += bl.x; w4.y += bl.y; w4.z += bl.z; w4.w += bl.w;
        float pe = on * dot4_tanh(w4, s3, ld4(a.Ult + 4 * d4));
        const int lane = tid & 63, w = tid >> 6;
        pe = wave_sum(pe);
        __syncthreads();                       // (s_red is being re-used)
        if (lane == 0) s_red[w] = pe;
        __syncthreads();
        if (tid == 0) { float r = 0.f; for (int q = 0; q < nw; ++q) r += s_red[q]; a.elt[bt] = r + clt0; }
    }
}

__global__ __launch_bounds__(256) void spatial_small_kernel(const SpatialArgs a) { spatial_small_body(a); }

// One-hypothesis decode (greedy, ancestral sampling): the same kernel, plus u.nvid extra workgroups that run the beam bookkeeping
// of the PREVIOUS word (beam_inl.h: arg-max / draw over the vocabulary statistics, log-sum-exp, token, score, <eos>, the chosen
// word's embedding for this word's LSTM launch).  The attention of a word reads the state projections the readout launch left,
// never the chosen word: as a launch of its own the update was 8.9 of the 42.5 us of a configs[0] word.
__global__ __launch_bounds__(256) void spatial_small_update_kernel(const SpatialArgs a, const BeamArgs u) {
    const int items = a.M * a.T;
    if ((int)blockIdx.x >= items) { beam_update_body(u, 0, nullptr, nullptr, (int)blockIdx.x - items, u.nvid); return; }
    spatial_small_body(a);
}

// Beams of 2 .. 8 hypotheses on the small path: the update of such a beam is a 1024-thread job (k * k * tiles candidates to scan), so
// the launch runs 1024-thread workgroups; an attention item only uses its first D / 4 lanes (the others load clamped addresses and
// contribute zeros), and what the 128-register budget of sixteen waves costs it disappears under the update beside it.
__global__ __launch_bounds__(1024) void spatial_small_update_wide_kernel(const SpatialArgs a, const BeamArgs u) {
    const int items = a.M * a.T;
    if ((int)blockIdx.x >= items) { beam_update_body(u, 0, nullptr, nullptr, (int)blockIdx.x - items, u.rw_cost ? u.nvid * u.k : u.nvid); return; }
    spatial_small_body(a);
}

#ifndef STATTN_BF16_FWD_SCHED
#define STATTN_BF16_FWD_SCHED 0
#endif
template <int NT>
__global__ __launch_bounds__(NT) void spatial_bf16_kernel(const SpatialArgs a) {
    constexpr int NW = NT / 64;
    __shared__ float s_red[NW * 10];
    __shared__ float s_e[KMAX];
    // the first rider.nblocks workgroups compute the rider GEMM (h.U of this step) instead of an attention item (see spatial2_kernel)
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[NW * 64 * 16];
        rider_tile<NW>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt ? a.clt[0] : 0.f;   // scalar loads up front, not inside the one-lane branches that use them
    const int bt = xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T), b = bt / T, t = bt % T;
    const int v = a.vid ? a.vid[b] : b;
    const int tid = threadIdx.x;
    const size_t slab = ((size_t)v * T + t) * K * D;
    const uint16_t* __restrict__ PL = reinterpret_cast<const uint16_t*>(a.PL) + slab;
    const uint16_t* __restrict__ L = reinterpret_cast<const uint16_t*>(a.L) + slab;
    const uint16_t* __restrict__ LW = reinterpret_cast<const uint16_t*>(a.LW) + slab;
    const float* __restrict__ sl = a.sproj + (size_t)b * a.ldsp;
    const int nd8 = D >> 3;

    for (int k0 = 0; k0 < K; k0 += 8) {
        float p[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) p[i] = 0.f;
        for (int d8 = tid; d8 < nd8; d8 += NT) {
            uint4 x[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) x[kk] = ld16(PL + (size_t)min(k0 + kk, K - 1) * D + 8 * d8);
            const float4 s0 = ld4(sl + 8 * d8), s1 = ld4(sl + 8 * d8 + 4);
            const float4 u0 = ld4(a.Ul + 8 * d8), u1 = ld4(a.Ul + 8 * d8 + 4);
#if STATTN_BF16_FWD_SCHED
            __builtin_amdgcn_sched_barrier(0);      // (probe build `fwdsched`: every row of the group requested before the first use -- see spatial_bf16v2_kernel)
#endif
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float f[8];
                bf8_to_f32(x[kk], f);
                p[kk] += dot4_tanh(make_float4(f[0], f[1], f[2], f[3]), s0, u0) +
                         dot4_tanh(make_float4(f[4], f[5], f[6], f[7]), s1, u1);
            }
            if (k0 == 0) {      // frame scores: PG / PM stay fp32 (they are K times smaller)
                const size_t fo = ((size_t)v * T + t) * D + 8 * d8;
                p[8] += dot4_tanh(ld4(a.PG + fo), ld4(sl + D + 8 * d8), ld4(a.Ug + 8 * d8)) +
                        dot4_tanh(ld4(a.PG + fo + 4), ld4(sl + D + 8 * d8 + 4), ld4(a.Ug + 8 * d8 + 4));
                p[9] += dot4_tanh(ld4(a.PM + fo), ld4(sl + 2 * D + 8 * d8), ld4(a.Um + 8 * d8)) +
                        dot4_tanh(ld4(a.PM + fo + 4), ld4(sl + 2 * D + 8 * d8 + 4), ld4(a.Um + 8 * d8 + 4));
            }
        }
        block_sum_w<10, NW>(p, s_red, tid);
        if (tid < 8 && k0 + tid < K) s_e[k0 + tid] = p[tid] + cl0;
        if (k0 == 0 && tid == 8) a.eg[bt] = p[8] + cg0;
        if (k0 == 0 && tid == 9) a.em[bt] = p[9] + cm0;
    }
    __syncthreads();

    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_e[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(s_e[k] - mx);
    const float inv = 1.0f / sum;
    __syncthreads();
    if (tid < K) {
        const float al = __expf(s_e[tid] - mx) * inv;
        a.alphal[(size_t)bt * K + tid] = al;
        s_e[tid] = al;
    }
    __syncthreads();

    float pe[1] = {0.f};
    for (int d8 = tid; d8 < nd8; d8 += NT) {
        float c[8], w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { c[q] = 0.f; w[q] = 0.f; }
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float al = s_e[k];
            float f[8], q8[8];
            bf8_to_f32(ld16(L + (size_t)k * D + 8 * d8), f);
            bf8_to_f32(ld16(LW + (size_t)k * D + 8 * d8), q8);
#pragma unroll
            for (int q = 0; q < 8; ++q) { c[q] += al * f[q]; w[q] += al * q8[q]; }
        }
        st4(a.CL + (size_t)bt * D + 8 * d8, make_float4(c[0], c[1], c[2], c[3]));
        st4(a.CL + (size_t)bt * D + 8 * d8 + 4, make_float4(c[4], c[5], c[6], c[7]));
        const float4 b0 = ld4(a.blt + 8 * d8), b1 = ld4(a.blt + 8 * d8 + 4);
        pe[0] += dot4_tanh(make_float4(w[0] + b0.x, w[1] + b0.y, w[2] + b0.z, w[3] + b0.w), ld4(sl + 3 * D + 8 * d8), ld4(a.Ult + 8 * d8)) +
                 dot4_tanh(make_float4(w[4] + b1.x, w[5] + b1.y, w[6] + b1.z, w[7] + b1.w), ld4(sl + 3 * D + 8 * d8 + 4), ld4(a.Ult + 8 * d8 + 4));
    }
    block_sum_w<1, NW>(pe, s_red, tid);
    if (tid == 0) a.elt[bt] = pe[0] + clt0;
}

#if STATTN_EXPERIMENTAL
#include "experimental/attn_bf16v2.inl"       // spatial_bf16v2_kernel: unmeasured, never in the product build
#endif

// ---- beam-search variant: the H hypotheses of a video (rows v*H .. v*H+H-1) attend to the SAME region tensors, so
// one workgroup per (video, frame) streams each K x D slab ONCE and applies it to all H state projections:
// per step 3 slabs x nvid x T instead of 3 x H x that (BASELINE configs[4]: 1.0 GB instead of 5.2 GB through the CUs).
// The slab is walked in groups of 8 regions held in registers (32 VGPRs); per hypothesis only the 8 partial scores,
// the attended feature and its LW twin are carried (H x (8 + 4 + 4) VGPRs), the softmax lives in LDS.
template <int H>
#ifndef STATTN_SHARED_RG
#define STATTN_SHARED_RG 4
#endif
#ifndef STATTN_SH_ABL
#define STATTN_SH_ABL 0       // probe builds: 1 = score phase without its arithmetic, 2 = without its slab loads
#endif
#ifndef STATTN_SH_PHASE
#define STATTN_SH_PHASE 0     // 1 / 2: probe builds that run one phase only
#endif
__device__ __forceinline__ void spatial_shared_body(const SpatialArgs& a, const int vt) {
    __shared__ float s_red[4 * 2 * H];
    __shared__ float s_e[H][KMAX];
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt ? a.clt[0] : 0.f;   // scalar loads up front, not inside the one-lane branches that use them
    const int v = vt / T, t = vt % T;
    const int tid = threadIdx.x;
    const size_t slab = ((size_t)v * T + t) * K * D;
    const float* __restrict__ PL = a.PL + slab;
    const float* __restrict__ L = a.L + slab;
    const float* __restrict__ LW = a.LW ? a.LW + slab : nullptr;
    const int b0 = v * H;                                   // first row (hypothesis) of this video
    const int nd4 = D >> 2;

#if STATTN_SH_PHASE == 2     // probe builds only (tools/shared_probe.sh): the weighted-sum phase alone, uniform weights
    if (tid < H * KMAX) (&s_e[0][0])[tid] = 1.f / K;
    __syncthreads();
#else
    // ---- region scores of all hypotheses.  Wave w owns regions 8w .. 8w+7 (+32, ...) over the WHOLE of D (a row of a
    // slab is 64 lanes x 16 B x D/256 fully coalesced loads), so a wave_sum finishes its scores: no cross-wave reduction
    const int lane = tid & 63, w = tid >> 6;
    constexpr int RP = 8;     // regions in flight per wave and pass (wave w owns regions RP w .. RP w + RP - 1, then + 4 RP)
    for (int k0 = RP * w; k0 < K; k0 += 4 * RP) {
        // This phase is VALU-bound at configs[4] (H K D = 164 k tanh per workgroup, 419 M per word: as long as the
        // 1 GB of slabs takes to stream), so the tanh is split along its sum:
        //     tanh(x + s) = 1 - 2 / (1 + e^{2x} e^{2s})
        // e^{2x} is formed ONCE per slab element and shared by the H hypotheses, e^{2s} once per (hypothesis, column) and
        // shared by the 8 regions in flight: per (hypothesis, region, column) one multiply, one add, one v_rcp and one
        // FMA remain -- ONE transcendental instead of two.  U . 1 is added once at the end (the "1 -" of every term).
        // Exponents are clamped to +-80 (|x|, |s| <= 40: e^{80} e^{80} = inf -> tanh = 1, e^{-80} e^{-80} = 0 -> -1, never
        // 0 x inf); inside that domain the result equals the direct form to 1e-7.
        float p[RP * H];         // p[h * RP + kk] = sum_d (-2 U_d) / (1 + e^{2x} e^{2s})
        float usum = 0.f;
#pragma unroll
        for (int i = 0; i < RP * H; ++i) p[i] = 0.f;
        for (int d4 = lane; d4 < nd4; d4 += 64) {
            float4 x[RP];
#pragma unroll
#if STATTN_SH_ABL == 2
            for (int kk = 0; kk < RP; ++kk) x[kk] = make_float4(1e-3f * (lane + kk), 1e-3f * d4, 0.1f, 0.2f);
#else
            for (int kk = 0; kk < RP; ++kk) x[kk] = ld4_nt(PL + (size_t)min(k0 + kk, K - 1) * D + 4 * d4);
#endif
            // the H state-projection rows are requested HERE, with the slab rows: loaded inside the per-hypothesis blocks below
            // (which a sched_barrier keeps apart) each was an exposed L2 round trip, five per column group
            float4 sp[H];
#pragma unroll
            for (int h = 0; h < H; ++h) sp[h] = ld4(a.sproj + (size_t)(b0 + h) * a.ldsp + 4 * d4);
            const float4 u4 = ld4(a.Ul + 4 * d4);
            usum += (u4.x + u4.y) + (u4.z + u4.w);
            const float4 m2u = make_float4(-2.f * u4.x, -2.f * u4.y, -2.f * u4.z, -2.f * u4.w);
#if STATTN_SH_ABL == 1
#pragma unroll
            for (int kk = 0; kk < RP; ++kk) p[kk] += (x[kk].x + x[kk].y) + (x[kk].z + x[kk].w) + sp[kk % H].x;
            if (false)
#endif
#pragma unroll
            for (int kk = 0; kk < RP; ++kk) x[kk] = exp2x4(x[kk]);
#if STATTN_SH_ABL == 1
            if (false)
#endif
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float4 es = exp2x4(sp[h]);
#pragma unroll
                for (int kk = 0; kk < RP; ++kk) {
                    p[h * RP + kk] += m2u.x * fast_rcp(1.f + x[kk].x * es.x) + m2u.y * fast_rcp(1.f + x[kk].y * es.y) +
                                     m2u.z * fast_rcp(1.f + x[kk].z * es.z) + m2u.w * fast_rcp(1.f + x[kk].w * es.w);
                }
                __builtin_amdgcn_sched_barrier(0);     // one hypothesis at a time: interleaving all 8 H chains spills
            }
        }
        usum = wave_sum(usum);
#pragma unroll
        for (int i = 0; i < RP * H; ++i) {
            const float r = wave_sum(p[i]);
            if (lane == 0 && k0 + (i % RP) < K) s_e[i / RP][k0 + (i % RP)] = r + usum + cl0;
        }
    }
    // ---- the two frame scores per hypothesis (PG / PM rows are shared by the hypotheses as well)
    {
        float q[2 * H];
#pragma unroll
        for (int i = 0; i < 2 * H; ++i) q[i] = 0.f;
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            const size_t fo = ((size_t)v * T + t) * D + 4 * d4;
            const float4 pg = ld4(a.PG + fo), pm = ld4(a.PM + fo), ug = ld4(a.Ug + 4 * d4), um = ld4(a.Um + 4 * d4);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float* sp = a.sproj + (size_t)(b0 + h) * a.ldsp + 4 * d4;
                q[2 * h] += dot4_tanh(pg, ld4(sp + D), ug);
                q[2 * h + 1] += dot4_tanh(pm, ld4(sp + 2 * D), um);
            }
        }
        block_sum<2 * H>(q, s_red, tid);
#pragma unroll
        for (int h = 0; h < H; ++h)
            if (tid == h) { a.eg[(size_t)(b0 + h) * T + t] = q[2 * h] + cg0; a.em[(size_t)(b0 + h) * T + t] = q[2 * h + 1] + cm0; }
    }
    __syncthreads();
    // ---- softmax over the regions, one wave-lane group per hypothesis (K <= 64: one lane per region)
    for (int h = w; h < H; h += 4) {                        // one wave per hypothesis, one lane per region
        const float e = lane < K ? s_e[h][lane] : -INFINITY;
        const float mx = wave_max(e);
        const float ex = lane < K ? __expf(e - mx) : 0.f;
        const float al = ex / wave_sum(ex);
        if (lane < K) { s_e[h][lane] = al; a.alphal[((size_t)(b0 + h) * T + t) * K + lane] = al; }
    }
    __syncthreads();
#endif
#if STATTN_SH_PHASE != 1
    // ---- attended local feature (and the local-temporal score) per hypothesis: L / LW streamed once
    float pe[H];
#pragma unroll
    for (int h = 0; h < H; ++h) pe[h] = 0.f;
    // (written once per lt_mode: with `LW ? load : 0` inside, hipcc kept a branch around every LW load and put a full wait behind
    // each of the first four -- the phase began with four dependent HBM round trips)
    auto sums = [&](auto has_lw) {
        constexpr bool HAS_LW = decltype(has_lw)::value;
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            float4 c4[H], w4[H];
    #pragma unroll
            for (int h = 0; h < H; ++h) { c4[h] = make_float4(0.f, 0.f, 0.f, 0.f); w4[h] = c4[h]; }
            // RG regions per round, the NEXT round's rows requested before this round's FMAs (the rounds used to be a chain of
            // K / RG exposed load latencies: this phase alone was 174 us at configs[4] for 0.67 GB)
            constexpr int RG = STATTN_SHARED_RG;
            float4 l4[RG], q4[RG], ln[RG], qn[RG];
            auto rows = [&](float4 (&l)[RG], float4 (&q)[RG], int k0) {
    #pragma unroll
                for (int kk = 0; kk < RG; ++kk) {
                    const int k = min(k0 + kk, K - 1);
                    l[kk] = ld4_nt(L + (size_t)k * D + 4 * d4);
                    q[kk] = HAS_LW ? ld4_nt(LW + (size_t)k * D + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            rows(l4, q4, 0);
            for (int k0 = 0; k0 < K; k0 += RG) {
                rows(ln, qn, min(k0 + RG, K - 1));
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int kk = 0; kk < RG; ++kk) {
    #pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const float al = k0 + kk < K ? s_e[h][k0 + kk] : 0.f;
                        c4[h].x += al * l4[kk].x; c4[h].y += al * l4[kk].y; c4[h].z += al * l4[kk].z; c4[h].w += al * l4[kk].w;
                        w4[h].x += al * q4[kk].x; w4[h].y += al * q4[kk].y; w4[h].z += al * q4[kk].z; w4[h].w += al * q4[kk].w;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int kk = 0; kk < RG; ++kk) { l4[kk] = ln[kk]; q4[kk] = qn[kk]; }
            }
            const float4 bl = HAS_LW ? ld4(a.blt + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
    #pragma unroll
            for (int h = 0; h < H; ++h) {
                st4(a.CL + ((size_t)(b0 + h) * T + t) * D + 4 * d4, c4[h]);
                if (HAS_LW) {
                    float4 z = w4[h];
                    z.x += bl.x; z.y += bl.y; z.z += bl.z; z.w += bl.w;
                    pe[h] += dot4_tanh(z, ld4(a.sproj + (size_t)(b0 + h) * a.ldsp + 3 * D + 4 * d4), ld4(a.Ult + 4 * d4));
                }
            }
        }
    };
    if (LW) sums(std::true_type{}); else sums(std::false_type{});
    if (LW) {
        block_sum<H>(pe, s_red, tid);
#pragma unroll
        for (int h = 0; h < H; ++h)
            if (tid == h) a.elt[(size_t)(b0 + h) * T + t] = pe[h] + clt0;
    }
#endif
}

template <int H>
__global__ __launch_bounds__(256, 3) void spatial_shared_kernel(const SpatialArgs a) { spatial_shared_body<H>(a, (int)blockIdx.x); }

// Beam search: the same launch with the bookkeeping of the PREVIOUS word in its first u.nvid workgroups (beam_inl.h).  The attention
// of word w + 1 is computed for the hypotheses as they stand BEFORE the beam is re-ordered -- it depends on a hypothesis' state only,
// not on the word chosen for it -- and the temporal kernel picks the parent's scores / region contexts through BeamArgs::rowmap:
// selection, log-sum-exp and the gathers (26 us per configs[4] word as a launch of their own, one workgroup per video) run under
// this HBM-bound launch.
template <int H>
__global__ __launch_bounds__(256, 3) void spatial_shared_update_kernel(const SpatialArgs a, const BeamArgs u) {
    if ((int)blockIdx.x < u.nvid) { beam_update_body(u, 0, nullptr, nullptr, (int)blockIdx.x, u.nvid); return; }
    spatial_shared_body<H>(a, (int)blockIdx.x - u.nvid);
}

#if STATTN_EXPERIMENTAL
#include "experimental/attn_shared_cols.inl"   // spatial_shared_cols_kernel: unmeasured, never in the product build
#endif

// one wave per row: out[r] = dot(P[r,:], U) + c
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ P, int ldp,
                                                     const float* __restrict__ U, const float* __restrict__ c,
                                                     float* __restrict__ out, int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int d4 = lane; d4 < (D >> 2); d4 += 64) {
        const float4 p = ld4(P + (size_t)r * ldp + 4 * d4), u = ld4(U + 4 * d4);
        s += p.x * u.x + p.y * u.y + p.z * u.z + p.w * u.w;
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s + c[0];
}

// One workgroup per (row b, 256-wide slice of D).  The frames a wave will weight (t = w, w+4, ...) are requested
// BEFORE the softmaxes: their addresses do not depend on the attention weights, so the loads fly while the three
// softmaxes are computed (they used to start after the barrier: two dependent latencies per launch).
constexpr int TPF = 8;    // frames per wave held in registers (T <= 32); longer videos loop

__global__ __launch_bounds__(256) void temporal_kernel(const TemporalArgs a) {
    __shared__ float s_al[3][TMAX];
    __shared__ float s_sel;
    __shared__ __attribute__((aligned(16))) float s_part[3][4][256];
    const int T = a.T, D = a.D;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int v = a.vid ? a.vid[b] : b;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = chunk * 256 + 4 * lane;
    const bool on = d < D;
    const float* __restrict__ G = a.G + (size_t)v * T * D + (on ? d : 0);
    const float* __restrict__ Mo = a.Mo + (size_t)v * T * D + (on ? d : 0);
    // beam search with the update riding in the attention launch: scores and region contexts were computed for the hypotheses
    // BEFORE the re-ordering -- row b takes its parent's
    const int bs = a.rowmap ? a.rowmap[b] : b;
    const float* __restrict__ CL = a.CL + (size_t)bs * T * D + (on ? d : 0);
    float4 g4[TPF], m4[TPF], c4[TPF];
#pragma unroll
    for (int i = 0; i < TPF; ++i) {
        const int t = min(w + 4 * i, T - 1);
        g4[i] = ld4(G + (size_t)t * D); m4[i] = ld4(Mo + (size_t)t * D); c4[i] = ld4(CL + (size_t)t * D);
    }

    if (w < 3) {          // three softmaxes over T, one wave each
        // the lane's scores are read ONCE, all of them in flight together (three passes over e[] were three dependent round trips)
        const float* e = (w == 0 ? a.eg : (w == 1 ? a.em : a.elt)) + (size_t)bs * T;
        constexpr int NE = (TMAX + 63) / 64;
        float ev[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) ev[i] = e[min(lane + 64 * i, T - 1)];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NE; ++i) if (lane + 64 * i < T) mx = fmaxf(mx, ev[i]);
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) { ev[i] = lane + 64 * i < T ? __expf(ev[i] - mx) : 0.f; sum += ev[i]; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float* out = (w == 0 ? a.alphag : (w == 1 ? a.alpham : a.alphalt)) + (size_t)b * T;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int t = lane + 64 * i;
            if (t < T) {
                const float al = ev[i] * inv;
                s_al[w][t] = al;
                if (chunk == 0) out[t] = al;
            }
        }
    } else {              // selector gate sigma(h_prev . W_sel + b_sel)   (:433)
        float sel = 1.f;
        if (a.W_sel) {
            const float bsel = a.b_sel[0];
            float s = 0.f;
            for (int d4 = lane; d4 < (D >> 2); d4 += 64) {
                const float4 h4 = ld4(a.h_prev + (size_t)b * D + 4 * d4), w4 = ld4(a.W_sel + 4 * d4);
                s += h4.x * w4.x + h4.y * w4.y + h4.z * w4.z + h4.w * w4.w;
            }
            s = wave_sum(s);
            sel = fast_sigmoid(s + bsel);
        }
        if (lane == 0) {
            s_sel = sel;
            if (chunk == 0 && a.sel) a.sel[b] = sel;
        }
    }
    __syncthreads();

    // weighted sums over frames: wave w takes t = w, w+4, ...; lane owns 4 consecutive d.  cg, cm, clt are kept apart
    // (the backward pass needs each of them) and added at the end
    float4 ac[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) ac[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma3 = [&](int t, const float4& g, const float4& m, const float4& c) {
        const float ag = s_al[0][t], am = s_al[1][t], alt = s_al[2][t];
        ac[0].x += ag * g.x; ac[0].y += ag * g.y; ac[0].z += ag * g.z; ac[0].w += ag * g.w;
        ac[1].x += am * m.x; ac[1].y += am * m.y; ac[1].z += am * m.z; ac[1].w += am * m.w;
        ac[2].x += alt * c.x; ac[2].y += alt * c.y; ac[2].z += alt * c.z; ac[2].w += alt * c.w;
    };
#pragma unroll
    for (int i = 0; i < TPF; ++i) {
        const int t = w + 4 * i;
        if (t < T) fma3(t, g4[i], m4[i], c4[i]);
    }
    for (int t = w + 4 * TPF; t < T; t += 4)            // frames past the register-held ones (T > 32)
        fma3(t, ld4(G + (size_t)t * D), ld4(Mo + (size_t)t * D), ld4(CL + (size_t)t * D));
#pragma unroll
    for (int x = 0; x < 3; ++x) st4(&s_part[x][w][4 * lane], ac[x]);
    __syncthreads();
    if (w == 0 && on) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float4 px = ld4(&s_part[x][0][4 * lane]);
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                const float4 q = ld4(&s_part[x][i][4 * lane]);
                px.x += q.x; px.y += q.y; px.z += q.z; px.w += q.w;
            }
            if (a.cparts) st4(a.cparts + ((size_t)x * a.M + b) * D + d, px);
            r.x += px.x; r.y += px.y; r.z += px.z; r.w += px.w;
        }
        if (a.csum) st4(a.csum + (size_t)b * D + d, r);
        const float sel = s_sel;
        r.x *= sel; r.y *= sel; r.z *= sel; r.w *= sel;
        st4(a.ctx + (size_t)b * D + d, r);
        if (a.ctx_pk) st4(a.ctx_pk + pn_pack_offset(b, d, D >> 4), r);
    }
}

}  // namespace

// only spatial2_kernel carries a rider (its 120-VGPR budget covers the rider's 95; spatial_kernel would drop from six to
// four workgroups per CU, the bf16 and shared-slab kernels are not per-row)
static bool spatial_shared_path(const SpatialArgs& a) {
    static const char* noshare = sw_tool("STATTN_SPATIAL_NOSHARE");     // A/B switch for tools
    static const char* mn = sw_tool("STATTN_SHARED_MIN");               // tools/shared_rounds_probe.py: smallest (video, frame) grid that takes it
    // Smallest (video, frame) grid that takes it, measured against the per-row kernels at beam 5, D = 1024 (tools/shared_rounds_probe.py, attention +
    // state-projection launches together -- up to 64 rows the per-row launch also carries h.U): K = 8, T = 26: 104 / 312 / 520 items lose (45 / 57 / 89 us
    // against 34 / 46 / 74), 832 tie, 1300 win (79 against 92); K = 16, T = 40: 160 items lose (47 against 41), 320 win (64 against 73), 1280: 92 against 188;
    // K = 32, T = 80: 160 lose, 320 win (48 against 59), 640: 65 against 185, 1920: 170 against 535.  (Until the end of round 4 the rule was 2048 items.)
    const int min_items = mn ? atoi(mn) : (a.K <= 8 ? 800 : 320);
    return a.group > 1 && a.group <= 8 && a.M % a.group == 0 && !noshare && (a.M / a.group) * a.T >= min_items;
}
bool spatial_rider_supported(const SpatialArgs& a) {
    static const char* norider = sw_product("STATTN_NO_RIDER");            // A/B switch for tools
    static const char* v1 = sw_tool("STATTN_SPATIAL1");
    return !norider && (a.bf16 || !spatial_shared_path(a)) && a.D % 1024 == 0 && !v1;
}

static bool spatial_small_path(const SpatialArgs& a) {
    static const char* nosm = sw_tool("STATTN_SPATIAL_NOSMALL");   // A/B switch for tools
    return !nosm && !a.bf16 && !spatial_shared_path(a) && !a.rider.nblocks && a.M * a.T <= 512 && a.K <= 8 && a.D <= 1024;
}
bool spatial_update_supported(const SpatialArgs& a) {
    return a.M > 0 && a.K >= 1 && a.K <= KMAX && a.D % 4 == 0 && !a.bf16 && !a.rider.nblocks && (spatial_small_path(a) || spatial_shared_path(a));
}

// does a riding update (launch_spatial with `upd`) run as k workgroups per video?  (the caller's path counter asks what ran)
bool spatial_update_row_workgroups(const SpatialArgs& a, const BeamArgs& u) {
    return u.rw_cost && !spatial_shared_path(a) && u.k >= 2 && u.rw_idx && u.rw_ticket;
}

hipError_t launch_spatial(hipStream_t s, const SpatialArgs& a, const BeamArgs* upd) {
    if (upd) {
        if (!spatial_update_supported(a) || !upd->stats || !upd->ticket || upd->nvid < 1 || upd->ntile < 1 || upd->k > PN_STATS_KB ||
            (upd->stochastic && (upd->k != 1 || upd->tile_cols < 1)) || (upd->proj_next && (!upd->proj_step || upd->nproj % 4)))
            return hipErrorInvalidValue;
        BeamArgs u1;
        if (upd->rw_cost && !spatial_update_row_workgroups(a, *upd)) {     // row workgroups: the 1024-thread small launch only
            u1 = *upd; u1.rw_cost = nullptr; upd = &u1;
        }
        if (spatial_shared_path(a)) {
            const dim3 grid(a.M / a.group * a.T + upd->nvid), block(256);
#if STATTN_EXPERIMENTAL
            if (exp_launch_shared_cols_update(s, a, *upd, grid, block)) return hipGetLastError();
#endif
            switch (a.group) {
                case 2: hipLaunchKernelGGL(spatial_shared_update_kernel<2>, grid, block, 0, s, a, *upd); break;
                case 3: hipLaunchKernelGGL(spatial_shared_update_kernel<3>, grid, block, 0, s, a, *upd); break;
                case 4: hipLaunchKernelGGL(spatial_shared_update_kernel<4>, grid, block, 0, s, a, *upd); break;
                case 5: hipLaunchKernelGGL(spatial_shared_update_kernel<5>, grid, block, 0, s, a, *upd); break;
                case 6: hipLaunchKernelGGL(spatial_shared_update_kernel<6>, grid, block, 0, s, a, *upd); break;
                case 7: hipLaunchKernelGGL(spatial_shared_update_kernel<7>, grid, block, 0, s, a, *upd); break;
                default: hipLaunchKernelGGL(spatial_shared_update_kernel<8>, grid, block, 0, s, a, *upd); break;
            }
            return hipGetLastError();
        }
        const int nt = ((a.D / 4 + 63) / 64) * 64;
        if (upd->k > 1) hipLaunchKernelGGL(spatial_small_update_wide_kernel, dim3(a.M * a.T + (upd->rw_cost ? upd->nvid * upd->k : upd->nvid)), dim3(1024), 0, s, a, *upd);
        else hipLaunchKernelGGL(spatial_small_update_kernel, dim3(a.M * a.T + upd->nvid), dim3(nt), 0, s, a, *upd);
        return hipGetLastError();
    }
    if (a.M <= 0) return hipSuccess;
    if (a.K > KMAX || a.K < 1 || a.D % 4 != 0) return hipErrorInvalidValue;
    if (a.rider.nblocks && (!spatial_rider_supported(a) || !rider_shape_ok(a.rider))) return hipErrorInvalidValue;
    if (a.bf16) {
        if (a.D % 8 != 0 || !a.LW) return hipErrorInvalidValue;
#if STATTN_EXPERIMENTAL
        if (exp_launch_spatial_bf16v2(s, a)) return hipGetLastError();
#endif
        if (a.D <= 1024) hipLaunchKernelGGL(spatial_bf16_kernel<128>, dim3(a.M * a.T + a.rider.nblocks), dim3(128), 0, s, a);
        else hipLaunchKernelGGL(spatial_bf16_kernel<256>, dim3(a.M * a.T + a.rider.nblocks), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    // beam search: the `group` consecutive rows of a video share its region tensors -> one pass over each slab
    // (worth it from a few hundred (video, frame) items on: the rule and its measurements are in spatial_shared_path)
    if (spatial_shared_path(a)) {
        const dim3 grid(a.M / a.group * a.T), block(256);
#if STATTN_EXPERIMENTAL
        if (exp_launch_shared_cols(s, a, grid, block)) return hipGetLastError();
#endif
        switch (a.group) {
            case 2: hipLaunchKernelGGL(spatial_shared_kernel<2>, grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL(spatial_shared_kernel<3>, grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL(spatial_shared_kernel<4>, grid, block, 0, s, a); break;
            case 5: hipLaunchKernelGGL(spatial_shared_kernel<5>, grid, block, 0, s, a); break;
            case 6: hipLaunchKernelGGL(spatial_shared_kernel<6>, grid, block, 0, s, a); break;
            case 7: hipLaunchKernelGGL(spatial_shared_kernel<7>, grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL(spatial_shared_kernel<8>, grid, block, 0, s, a); break;
        }
        return hipGetLastError();
    }
    // a grid that does not fill the chip (decode with a handful of rows): the single-round-trip kernel
    // (on a grid that fills the chip it loses: 38 us against 33 at configs[1] -- three resident workgroups per CU with one
    // round trip each move fewer bytes than eight with three)
    if (spatial_small_path(a)) {
        const int nt = ((a.D / 4 + 63) / 64) * 64;
        hipLaunchKernelGGL(spatial_small_kernel, dim3(a.M * a.T), dim3(nt), 0, s, a);
        return hipGetLastError();
    }
    // D a multiple of 1024: 128-thread workgroups with two columns per thread (all items of configs[1] resident at
    // once: 36 us instead of 41 us per launch there); otherwise the 256-thread kernel, one column per thread
    static const char* v1 = sw_tool("STATTN_SPATIAL1");          // A/B switch for tools
    if (a.D % 1024 == 0 && !v1) hipLaunchKernelGGL(spatial2_kernel<128>, dim3(a.M * a.T + a.rider.nblocks), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(spatial_kernel, dim3(a.M * a.T), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rowdot(hipStream_t s, const float* P, int ldp, const float* U, const float* c,
                         float* out, int rows, int D) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(rowdot_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, P, ldp, U, c, out, rows, D);
    return hipGetLastError();
}

hipError_t launch_temporal(hipStream_t s, const TemporalArgs& a) {
    if (a.M <= 0) return hipSuccess;
    if (a.T > TMAX || a.T < 1 || a.D % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(temporal_kernel, dim3(a.M, (a.D + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace stattn
