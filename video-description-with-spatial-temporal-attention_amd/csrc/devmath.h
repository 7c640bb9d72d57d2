// Device math helpers (gfx950).  Fast transcendental forms built on v_exp_f32 / v_rcp_f32:
// absolute error ~1e-7, far inside the 1e-4 parity bar on attention weights and logits.
#pragma once
#include <hip/hip_runtime.h>

// gfx950 only.  Several hand-offs between workgroups of one launch (beam_inl.h row workgroups, the word counter) carry their data
// in agent-scope (sc1, written-through) stores and loads around a relaxed ticket, with the order fixed by s_waitcnt + s_barrier
// instead of release / acquire fences: measured correct and 2-3 us cheaper per hand-off on this chip, NOT a statement
// about the HIP memory model.  Building for another target must fail here rather than run them.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libstattn is written for gfx950 (MI355X / CDNA4) only"
#endif

namespace stattn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tanh(x) = 1 - 2 / (exp(2x) + 1); saturates cleanly (exp -> inf gives 1, exp -> 0 gives -1)
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * fast_rcp(e + 1.0f);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }

// Wave-wide sum / max, result in every lane.  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): six DEPENDENT LDS round
// trips per value, and hipcc does not interleave the chains of independent values (spatial2_kernel: 66 serial bpermutes per item,
// spatial_bwd_kernel: 144).  STATTN_DPP_REDUCE=1 builds the same reductions from six DPP lane moves on the VALU (quad swaps, half-row
// and row mirrors, row_bcast 15 / 31: the total forms in lane 63) and one v_readlane -- no LDS, no waits.
#ifndef STATTN_DPP_REDUCE
#define STATTN_DPP_REDUCE 0
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
#if STATTN_DPP_REDUCE
    v += dpp_move<0xB1, 0xF>(v, v);            // quad_perm:[1,0,3,2]
    v += dpp_move<0x4E, 0xF>(v, v);            // quad_perm:[2,3,0,1]   every lane: the sum of its quad
    v += dpp_move<0x141, 0xF>(v, v);           // row_half_mirror       ... of its 8 lanes
    v += dpp_move<0x140, 0xF>(v, v);           // row_mirror            ... of its row of 16
    v += dpp_move<0x142, 0xA>(0.f, v);         // row_bcast:15          rows 1, 3 += the row before
    v += dpp_move<0x143, 0xC>(0.f, v);         // row_bcast:31          rows 2, 3 += lane 31: lane 63 holds all 64
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#if STATTN_DPP_REDUCE
    v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));  // (rows 0, 2 keep their own value)
    v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
#endif
}

// Workgroup -> XCD placement.  Block n of a launch runs on XCD n % 8 (observed dispatch order; each XCD has its own 4 MiB
// L2).  Work items that read the same rows / lines should therefore sit on the same XCD, or every XCD fetches its own
// copy through the fabric.  Both maps are bijections of [0, nblocks) for any nblocks: only performance depends on them.
//   xcd_contiguous: XCD x gets a CONTIGUOUS range of items (neighbouring column tiles share cache lines)
__device__ __forceinline__ int xcd_contiguous(int n, int nblocks) {
    const int x = n & 7, i = n >> 3, q = nblocks >> 3, r = nblocks & 7;
    return x * q + (x < r ? x : r) + i;
}
//   xcd_rows: items are (row, sub) pairs with `per_row` sub-items per row (frames of a batch row): XCD x gets whole
//   rows x, x + 8, ... (valid when rows % 8 == 0, else the identity)
__device__ __forceinline__ int xcd_rows(int n, int rows, int per_row) {
    if (rows & 7) return n;
    const int x = n & 7, i = n >> 3;
    return (x + 8 * (i / per_row)) * per_row + i % per_row;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// a stream that must not displace what other kernels re-read from the caches (non-temporal policy)
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

// counter-based generator (splitmix64 finaliser): u in (0, 1) from (seed, a, b), and the Gumbel(0, 1) variate -log(-log u)
__device__ __forceinline__ unsigned long long mix64_dev(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float gumbel01(unsigned long long seed, unsigned long long a, unsigned long long b) {
    const unsigned long long hsh = mix64_dev(mix64_dev(seed ^ (a * 0x9E3779B97F4A7C15ull)) + b);
    const float u = ((float)(hsh >> 40) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1), never 0 or 1
    return -__logf(-__logf(u));
}

// e^{2x} per component, exponent clamped to +-80.  tanh(x + s) = 1 - 2 / (1 + e^{2x} e^{2s}) lets kernels that evaluate
// tanh(x + s_i) for MANY shifts s_i of the same x (the H hypotheses of a video in beam search, the 30 time steps of the
// deferred context gradients) pay one v_exp per x and per shift and only a v_rcp per pair -- one transcendental instead
// of two.  With the clamp e^{80} e^{80} = inf gives tanh = 1 and e^{-80} e^{-80} = 0 gives -1, never 0 x inf; for
// |x|, |s| <= 40 the result equals the direct form to 1e-7.
__device__ __forceinline__ float4 exp2x4(float4 x) {
    return make_float4(__expf(__builtin_amdgcn_fmed3f(2.f * x.x, -80.f, 80.f)), __expf(__builtin_amdgcn_fmed3f(2.f * x.y, -80.f, 80.f)),
                       __expf(__builtin_amdgcn_fmed3f(2.f * x.z, -80.f, 80.f)), __expf(__builtin_amdgcn_fmed3f(2.f * x.w, -80.f, 80.f)));
}
// r = 1 / (1 + ex es) per component:  tanh = 1 - 2 r,  1 - tanh^2 = 4 (r - r^2)
__device__ __forceinline__ float4 rcp1p4(float4 ex, float4 es) {
    return make_float4(fast_rcp(1.f + ex.x * es.x), fast_rcp(1.f + ex.y * es.y), fast_rcp(1.f + ex.z * es.z), fast_rcp(1.f + ex.w * es.w));
}

}  // namespace stattn
