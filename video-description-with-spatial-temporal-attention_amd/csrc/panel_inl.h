// Inner loops of the row-panel fp32 MFMA kernels (panel.hip), shared with the kernels that carry a "rider": extra
// workgroups in the grid of an HBM-bound attention launch that compute an independent row-panel GEMM on the idle
// matrix cores (attn.hip spatial kernels: preh = h.U of the same step, model_attention.py:437; bwd.hip
// spatial_bwd_kernel: dhU = dpre.U^T of the reverse scan).  See panel.hip for the operand layouts.
#pragma once
#include "kernels.h"
#include "devmath.h"

namespace stattn {

// Timeline probe (tools/panel_probe.hip builds panel.hip with -DSTATTN_PROBES): wave 0 of every workgroup stamps the
// 100 MHz wall clock at the phase boundaries.  Compiled out of the product.
#ifdef STATTN_PROBES
#ifndef PN_VARIANT
#define PN_VARIANT 0                // compile-time ablations: 1 A read as a packed stream, 2 no MFMAs, 3 no A loads, 4 no B loads
#endif
#endif

namespace {

template <int MT, int NT>
struct PnOps { float4 a[MT]; float4 b[NT]; };

// s = logical k-step; the physical one is rotated by `rot` (see pn_rotation)
template <int MT, int NT>
__device__ __forceinline__ void pn_load(PnOps<MT, NT>& o, const float* const (&Ap)[MT], int astep, const float* __restrict__ Bp,
                                        size_t tile_floats, int s, int rot, int nsteps, bool stream_b = false) {
    s += rot;
    s = s >= nsteps ? s - nsteps : s;
#if !(defined(STATTN_PROBES) && (PN_VARIANT == 4 || PN_VARIANT == 5))
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (stream_b) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(Bp + (size_t)i * tile_floats + (size_t)s * 256));
            o.b[i] = make_float4(t.x, t.y, t.z, t.w);
        } else o.b[i] = ld4(Bp + (size_t)i * tile_floats + (size_t)s * 256);
    }
#endif
#if defined(STATTN_PROBES) && (PN_VARIANT == 3 || PN_VARIANT == 5)
    return;
#endif
#pragma unroll
    for (int i = 0; i < MT; ++i) o.a[i] = ld4(Ap[i] + (size_t)astep * s);
}

template <int MT, int NT>
__device__ __forceinline__ void pn_mfma(f32x4 (&acc)[MT][NT], const PnOps<MT, NT>& o) {
#if defined(STATTN_PROBES) && PN_VARIANT == 2
    asm volatile("" :: "v"(o.a[0].x), "v"(o.b[0].x), "v"(o.a[MT - 1].w), "v"(o.b[NT - 1].w)); return;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][q], o.b[n][q], acc[i][n], 0, 0, 0);
}

// acc += A[rows of this wave, K-steps s0, s0 + stride, ...] . panel
// Ap[i]: row pointer of m-tile i (+ 4 g), Bp: packed panel of the first column tile (+ 4 lane), nsteps = K / 16.
//
// What bounds these kernels is memory-level parallelism, not bandwidth (tools/panel_probe.hip): the weight panel of a
// workgroup comes from HBM / Infinity Cache with ~2 us of loaded latency, so a wave that keeps two or three k-steps in
// flight moves ~8 KB/us per CU and the loop takes three times its MFMA time.  Hence a ring of R k-steps of operands in
// registers: every load of the first R steps is issued before the first MFMA (for K = 1024 and eight K-slice waves
// that is the wave's whole share: the launch is one burst of loads followed by MFMAs), and a slot is refilled R steps
// ahead as soon as its MFMAs have been issued.
// The refills are unconditional: past the last step they re-request it (an L1 hit) -- a branch around the loads
// would make the compiler merge wait counts over both paths and drain the queue.  ONESHOT (the wave has at most R
// steps) compiles the refills out.
template <int MT, int NT, int R, bool ONESHOT>
__device__ __forceinline__ void pn_accumulate(f32x4 (&acc)[MT][NT], const float* const (&Ap)[MT], int astep,
                                              const float* __restrict__ Bp, size_t tile_floats, int nsteps, int s0, int stride, int rot,
                                              bool stream_b = false) {
    if (s0 >= nsteps) return;
    const int n = (nsteps - s0 + stride - 1) / stride;         // steps of this wave
    const int last = s0 + (n - 1) * stride;
    PnOps<MT, NT> ring[R];
#pragma unroll
    for (int u = 0; u < R; ++u) pn_load(ring[u], Ap, astep, Bp, tile_floats, min(s0 + u * stride, last), rot, nsteps, stream_b);
    int base = 0;
    if (!ONESHOT) {
        for (; base + 2 * R <= n; base += R) {   // groups whose refills are all real; no branch between loads and MFMAs
            const int sb = s0 + base * stride;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                __builtin_amdgcn_sched_barrier(0);
                pn_mfma(acc, ring[u]);
                __builtin_amdgcn_sched_barrier(0);
                pn_load(ring[u], Ap, astep, Bp, tile_floats, sb + (u + R) * stride, rot, nsteps, stream_b);
            }
        }
        if (base + R < n) {                      // one more refilling group when a partial group follows (clamped refills)
            const int sb = s0 + base * stride;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                __builtin_amdgcn_sched_barrier(0);
                pn_mfma(acc, ring[u]);
                __builtin_amdgcn_sched_barrier(0);
                pn_load(ring[u], Ap, astep, Bp, tile_floats, min(sb + (u + R) * stride, last), rot, nsteps, stream_b);
            }
            base += R;
        }
    }
    const int rem = n - base;                     // last group: its operands are in the ring, nothing is refilled
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (u >= rem) break;
        __builtin_amdgcn_sched_barrier(0);
        pn_mfma(acc, ring[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Every workgroup walks its panel in the same k order at the same time.  Unrotated, the 256 CUs would request the same
// offset of 256 panels that lie a power of two apart (one HBM / L2 channel at a time: the weight stream ran at 2 TB/s)
// and the same lines of A (32 CUs of an XCD on one L2 channel).  Rotating the k order per workgroup -- by its index
// within the XCD plus 8 per XCD -- spreads both over the channels; it only permutes the summation order.
__device__ __forceinline__ int pn_rotation(int block, int nsteps) { return ((block >> 3) + ((block & 7) << 3)) % nsteps; }

// 16x16 C/D map: col = lane & 15, row = 4 (lane >> 4) + r.  red[ks][row][col], row pitch CB.
template <int MT, int NT>
__device__ __forceinline__ void pn_spill(float* red, int RB, int ks, int mg, const f32x4 (&acc)[MT][NT], int j, int g) {
    constexpr int CB = 16 * NT;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((size_t)ks * RB + (mg * MT + i) * 16 + 4 * g + r) * CB + n * 16 + j] = acc[i][n][r];
}

// ---- rider: C[M x N] = A_pk[M x K] . P (+ add), one 16-column tile and one K-slice per workgroup of NW waves.
// Workgroup `blk` of the rider range owns tile blk % (N / 16) and K-slice blk / (N / 16); kz > 1 writes raw partial
// tiles to C + z * part_stride (summed by the consumer in a fixed order).  M <= 64 (four m-tiles per wave), 95-odd
// VGPRs: below the attention kernels' own budget, so carrying a rider does not change their occupancy.
// `red`: NW * 64 * 16 floats of LDS.
template <int NW, int R = 3>
__device__ __forceinline__ void rider_tile(const RiderArgs& r, int blk, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int ntile = r.N >> 4, S = r.K >> 4;
    const int c = blk % ntile, z = blk / ntile;
    f32x4 acc[4][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* Ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Ap[i] = r.A + ((size_t)min(i, (r.M - 1) >> 4) * S) * 256 + 4 * lane;
    const size_t tile_floats = (size_t)S * 256;
    pn_accumulate<4, 1, R, false>(acc, Ap, 256, r.P + (size_t)c * tile_floats + 4 * lane, tile_floats, S, z * NW + w, NW * r.kz,
                                  pn_rotation(c, S));      // (the same k order for every K-slice of a tile)
    pn_spill<4, 1>(red, 64, w, 0, acc, j, g);
    __syncthreads();
    for (int idx = tid; idx < 64 * 16; idx += NW * 64) {
        const int row = idx >> 4, col = idx & 15;
        if (row >= r.M) continue;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) v += red[((size_t)k * 64 + row) * 16 + col];
        const int n = c * 16 + col;
        if (r.kz > 1) { r.C[(size_t)z * r.part_stride + (size_t)row * r.ldc + n] = v; continue; }
        if (r.add) v += r.add[(size_t)row * r.ldadd + n];
        r.C[(size_t)row * r.ldc + n] = v;
    }
}

}  // namespace

}  // namespace stattn
