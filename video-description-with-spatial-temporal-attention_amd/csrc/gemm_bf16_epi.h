// Wide epilogue of the bf16-MFMA GEMM kernels (gemm_bf16.hip, gemm_bf16_8ph.hip) for wave tiles of 64 columns.
//
// The 32 x 32 MFMA accumulator layout gives a lane ONE column and sixteen rows: written straight to memory that is 16
// two- or four-byte stores per block, each with its own epilogue arithmetic, row index and -- in the generic form --
// uniform branches and an integer division (`row / rowgroup`) PER ELEMENT: 37-40 us of a 256 x 256 tile whose
// MFMAs take 45 (tools/gemm_8ph_probe.py, round 5).  Here the wave passes its tile through a private 16 KiB LDS
// region 64 rows at a time and comes back with EIGHT CONSECUTIVE COLUMNS of one row per lane: bias / add / rowadd /
// mul arrive as float4 loads, the division is once per row, and the stores are 16 bytes (bf16) or 2 x 16 bytes (fp32)
// per lane, whole 128 / 256-byte rows per 8 lanes.
//   write side: ds_write_b32, lanes 0-31 = 32 consecutive columns of a row (conflict-free);
//   read side : 2 x ds_read_b128 per lane; 16-byte chunk c of row r lives at c ^ (r & 1), which keeps the four rows
//               of a ds_read_b128 lane group on distinct bank quads.
#pragma once
#include "kernels.h"
#include "devmath.h"

namespace stattn {
namespace bf16_epi {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {       // round to nearest even, a in the low half
    const bf16x2_t p = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, p);
}

// per-wave staging region: 64 rows x 64 columns (NB = 2) or 32 rows x 32 columns (NB = 1) of fp32
template <int NB> struct Stage { static constexpr int ROWS = 32 * NB, COLS = 32 * NB, BYTES = ROWS * COLS * 4; };
constexpr int STAGE_BYTES = Stage<2>::BYTES;

// two-output problems (GemmBfArgs::n_split: columns >= n_split go to the second output set with their own bias): a tile
// lies on one side (n_split % tile width == 0); `col0` comes back relative to its output
struct Out { float* C; uint16_t* Cb; const float* bias; };
__device__ __forceinline__ Out select_out(const GemmBfArgs& g, int& col0) {
    if (g.n_split > 0 && col0 >= g.n_split) { col0 -= g.n_split; return Out{g.C2, g.Cb2, g.bias2}; }
    return Out{g.C, g.Cb, g.bias};
}

// acc[i][j][r]: row 32 i + (r & 3) + 8 (r >> 2) + 4 kh, column 32 j + l31 of the wave tile whose first element is
// (row0, col0); `stage` = this wave's private LDS region (Stage<NB>::BYTES, 16-byte aligned).  The caller has made sure
// (barrier) that nobody still reads the region's former content.  16-byte chunk c of staged row r lives at
// c ^ (r & 1) (NB = 2: 256-byte rows) or c ^ ((r >> 1) & 1) (NB = 1: 128-byte rows): the rows of one ds_read_b128 lane
// group then sit on distinct bank quads.
template <int MB, int NB, bool MEDGE>
__device__ __forceinline__ void store_tile(const GemmBfArgs& g, const f32x16 (&acc)[MB][NB], float* stage, int row0, int col0, int lane) {
    static_assert(NB == 1 || (NB == 2 && MB % 2 == 0), "NB = 2: 64 rows per pass");
    constexpr int W = 32 * NB;                   // staged row, floats
    constexpr int RB = NB;                       // 32-row blocks per pass
    constexpr int LPR = 4 * NB;                  // lanes per row on the read side (8 columns each)
    constexpr int RPI = 64 / LPR;                // rows per read iteration
    const Out o = select_out(g, col0);
    const int l31 = lane & 31, kh = lane >> 5;
    const int rl = lane / LPR, cg = lane % LPR;
    const int col = col0 + 8 * cg;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (o.bias) { b0 = ld4(o.bias + col); b1 = ld4(o.bias + col + 4); }
    if (g.bias_b) {
        const float4 c0 = ld4(g.bias_b + col), c1 = ld4(g.bias_b + col + 4);
        b0.x += c0.x; b0.y += c0.y; b0.z += c0.z; b0.w += c0.w; b1.x += c1.x; b1.y += c1.y; b1.z += c1.z; b1.w += c1.w;
    }
    float* w0 = stage + (4 * kh) * W + (((l31 >> 2) << 2) | (l31 & 3));               // rows whose swizzle bit is 0
    float* w1 = stage + (4 * kh) * W + ((((l31 >> 2) ^ 1) << 2) | (l31 & 3));         // ... is 1: chunk ^ 1
#pragma unroll
    for (int hb = 0; hb < MB; hb += RB) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = i * 32 + (r & 3) + 8 * (r >> 2);
                    const int sw = NB == 2 ? (r & 1) : ((r >> 1) & 1);
                    (sw ? w1 : w0)[lr * W + j * 32] = acc[hb + i][j][r];
                }
        __builtin_amdgcn_sched_barrier(0);
        for (int it = 0; it < 32 * RB / RPI; ++it) {
            const int lr = it * RPI + rl;
            const int row = row0 + hb * 32 + lr;
            if (MEDGE && row >= g.M) continue;
            const float* src = stage + lr * W;
            const int sw = NB == 2 ? (lr & 1) : ((lr >> 1) & 1);
            float4 v0 = *reinterpret_cast<const float4*>(src + (((2 * cg) ^ sw) << 2));
            float4 v1 = *reinterpret_cast<const float4*>(src + (((2 * cg + 1) ^ sw) << 2));
            v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w;
            v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
            if (g.add) {
                const float* p = g.add + (size_t)row * g.ldadd + col;
                const float4 a0 = ld4(p), a1 = ld4(p + 4);
                v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
                v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
            }
            if (g.rowadd) {
                const float* p = g.rowadd + (size_t)(row / g.rowgroup) * g.ldrow + col;
                const float4 a0 = ld4(p), a1 = ld4(p + 4);
                v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
                v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
            }
            if (g.act == 1) {
                v0.x = fast_tanh(v0.x); v0.y = fast_tanh(v0.y); v0.z = fast_tanh(v0.z); v0.w = fast_tanh(v0.w);
                v1.x = fast_tanh(v1.x); v1.y = fast_tanh(v1.y); v1.z = fast_tanh(v1.z); v1.w = fast_tanh(v1.w);
            }
            if (g.mul) {
                const float* p = g.mul + (size_t)row * g.ldmul + col;
                const float4 a0 = ld4(p), a1 = ld4(p + 4);
                v0.x *= a0.x; v0.y *= a0.y; v0.z *= a0.z; v0.w *= a0.w;
                v1.x *= a1.x; v1.y *= a1.y; v1.z *= a1.z; v1.w *= a1.w;
            }
            if (o.C) { float* p = o.C + (size_t)row * g.ldc + col; st4(p, v0); st4(p + 4, v1); }
            if (o.Cb) {
                const u32x4_t ov{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
                *reinterpret_cast<u32x4_t*>(o.Cb + (size_t)row * g.ldcb + col) = ov;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// what the wide form needs from the problem: 16-byte rows everywhere it reads or writes vectors
__host__ __device__ inline bool wide_ok(const GemmBfArgs& g) {
    auto al = [](const void* p) { return ((size_t)p & 15) == 0; };
    return (!g.C || (g.ldc % 4 == 0 && al(g.C))) && (!g.Cb || (g.ldcb % 8 == 0 && al(g.Cb))) &&
           (!g.add || (g.ldadd % 4 == 0 && al(g.add))) && (!g.rowadd || (g.ldrow % 4 == 0 && al(g.rowadd))) &&
           (!g.mul || (g.ldmul % 4 == 0 && al(g.mul))) && (!g.bias || al(g.bias)) && (!g.bias_b || al(g.bias_b)) &&
           (g.n_split <= 0 || ((!g.C2 || al(g.C2)) && (!g.Cb2 || al(g.Cb2)) && (!g.bias2 || al(g.bias2))));
}

}  // namespace bf16_epi
}  // namespace stattn
