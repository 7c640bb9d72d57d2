// Pieces shared by the two LDS-tiled GEMMs (gemm.hip: fp32-input MFMA; gemm_split.hip: bf16 MFMA on exactly split
// operands): the XCD-aware tile order, the dealing of a grouped launch's tiles to the XCDs, and the fused epilogue.
// Both kernels hold their accumulators in the 32x32 MFMA C layout.
#pragma once
#include "kernels.h"
#include "devmath.h"

namespace stattn {
namespace gemm_common {

constexpr int NXCD = 8;

// XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous run of tiles
// (consecutive tiles share the A row-panel -> L2 hits).  Bijective for any grid size.
__device__ __forceinline__ int xcd_linear(int bid, int nblk, int remap) {
    const int xcd = bid % NXCD, q8 = nblk / NXCD, r8 = nblk % NXCD;
    return remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / NXCD : bid;
}

// Grouped launch: the tiles of every problem are dealt to the eight XCDs separately (block b runs on XCD b % 8, which walks
// its share of problem 0, then of problem 1, ...), so each XCD gets the same mix of long-K and short-K tiles and the
// short ones fill its tail.  Returns false for a padding block of this XCD.
__device__ __forceinline__ bool group_locate(const GemmGroup& G, int bid, int& p, int& lin) {
    const int xcd = bid % NXCD;
    int j = bid / NXCD;
    for (p = 0; p < G.n; ++p) {
        const int tiles = G.tile_start[p + 1] - G.tile_start[p], q8 = tiles / NXCD, r8 = tiles % NXCD;
        const int mine = q8 + (xcd < r8 ? 1 : 0);
        if (j < mine) { lin = xcd * q8 + (xcd < r8 ? xcd : r8) + j; return true; }
        j -= mine;
    }
    return false;
}

// acc[i][j][r]: row = 32 i + (r & 3) + 8 (r >> 2) + 4 kh, column = 32 j + l31 of the wave tile (the 32x32 MFMA C layout)
template <int MT, int NT>
__device__ __forceinline__ void epilogue(const GemmArgs& g, const f32x16 (&acc)[MT][NT], int m0, int n0, int wm, int wn,
                                         int l31, int kh, float* Cout, int ldc) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * 32 * NT + j * 32 + l31;
            const float bias = (g.bias ? g.bias[col] : 0.f) + (g.bias2 ? g.bias2[col] : 0.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * MT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < g.M && g.kslices > 1) {
                    Cout[(size_t)row * ldc + col] = acc[i][j][r];
                } else if (row < g.M) {
                    float v = g.alpha * acc[i][j][r] + bias;
                    if (g.add) v += g.add[(size_t)row * g.ldadd + col];
                    if (g.rowadd) v += g.rowadd[(size_t)(row / g.rowgroup) * g.ldrow + col];
                    if (g.act == 1) v = fast_tanh(v);
                    if (g.Cact) g.Cact[(size_t)row * g.ldcact + col] = v;
                    if (g.mul) v *= g.mul[(size_t)row * g.ldmul + col];
                    float* c = g.C + (size_t)row * g.ldc + col;
                    if (g.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
}

}  // namespace gemm_common
}  // namespace stattn
