// LDS-tiled fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s peak).
//
// Used for the once-per-batch prologue of the decoder (ff_local / ff_motion F->D projections,
// model_attention.py:664-667; attention pre-projections :322-326; x projection :334-335), the
// per-step local-temporal projection CL.Wclt (:416, lt_mode 0), the batched readout (:687-705)
// and -- transposed variants -- the weight/input gradients of all of them.
//
// Tiling: block = 4 waves (2x2), block tile (64*TM) x (64*TN), BK = 32, two LDS stages.
// Each wave owns a (32*TM) x (32*TN) sub-tile = TM x TN accumulators of 32x32 (16 VGPR each).
//
// Operand trick: v_mfma_f32_32x32x2 contracts k over the two half-waves (lane>>5).  A lane
// reads FOUR consecutive k of its A row with one ds_read_b128 and feeds them to four MFMAs;
// half-wave 0 therefore covers k = 8kk+{0..3}, half-wave 1 k = 8kk+{4..7}.  The B operand uses
// the same k assignment, so the only effect is a permutation of the summation order inside a
// k-block of 8.  The LDS row stride of a k-contiguous tile is BK+4 dwords: ds_read_b128 is
// serviced in 16-lane groups and (36*i) mod 64 hits 16 distinct 4-bank slots -> conflict-free.
#include "gemm_common.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace stattn {

namespace {

using namespace gemm_common;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int KPAD = BK + 4;

// KC: operand stored k-contiguous in memory ([rows][K]); RC: row-contiguous ([K][rows]).
template <int BR, bool KC>
struct Tile {
    static constexpr int ELEMS = KC ? BR * KPAD : BK * BR;
    static constexpr int NF4 = BR * BK / 4 / 256;

    // Addressing: per thread and 16-byte piece a LOOP-INVARIANT 32-bit byte offset from the tile's first row (offsets()),
    // and per k-tile one wave-uniform base pointer, handed to a buffer load as its resource (scalar registers): no 64-bit
    // vector arithmetic in the loop (it was ~60 VALU instructions per k-tile: half of the issue slots the 64 x 64 tile's
    // four MFMAs of a k-block leave).  The resource is rebuilt per k-tile from the 64-bit base, so operands of any size
    // work; only a tile's own extent (128 rows x ld x 4 bytes) has to stay below 4 GB.  Rows beyond rows_total are clamped to the last row (edge rows are never stored).
    __device__ static __forceinline__ void offsets(unsigned (&off)[NF4], int ld, int r0, int rows_total, int tid) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int idx = tid + i * 256;
            if constexpr (KC) {
                const int rr = idx >> 3, kc = idx & 7;
                int row = r0 + rr;
                row = row < rows_total ? row : rows_total - 1;
                off[i] = ((unsigned)(row - r0) * (unsigned)ld + 4u * kc) * 4u;   // bytes
            } else {
                constexpr int RCW = BR / 4;
                const int k = idx / RCW, rc = idx % RCW;
                off[i] = ((unsigned)k * (unsigned)ld + 4u * rc) * 4u;
            }
        }
    }
    // uniform base of k-tile k0: KC rows start at X + r0 * ld + k0, RC rows at X + k0 * ld + r0
    __device__ static __forceinline__ const float* tile_base(const float* X, int ld, int r0, int k0) {
        if constexpr (KC) return X + (size_t)r0 * ld + k0;
        else return X + (size_t)k0 * ld + r0;
    }
    // EDGE = false: K is a multiple of BK and a row-contiguous operand has no partial tile -> no predicates, no
    // exec-mask branches around the loads (those make hipcc serialise the prefetch with vmcnt(0)).
    template <bool EDGE = true>
    __device__ static __forceinline__ void gload(float4 (&r)[NF4], const float* __restrict__ Xt, const unsigned (&off)[NF4],
                                                 int r0, int rows_total, int k0, int K, int tid) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xt), 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            bool ok = true;
            if constexpr (EDGE) {
                const int idx = tid + i * 256;
                if constexpr (KC) ok = k0 + 4 * (idx & 7) < K;
                else { constexpr int RCW = BR / 4; ok = k0 + idx / RCW < K && r0 + 4 * (idx % RCW) < rows_total; }
            }
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (ok) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], 0, 0);
            r[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    }
    __device__ static __forceinline__ void sstore(const float4 (&r)[NF4], float* s, int tid) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int idx = tid + i * 256;
            if constexpr (KC) {
                const int rr = idx >> 3, kc = idx & 7;
                st4(s + rr * KPAD + 4 * kc, r[i]);
            } else {
                constexpr int RCW = BR / 4;
                const int k = idx / RCW, rc = idx % RCW;
                st4(s + k * BR + 4 * rc, r[i]);
            }
        }
    }
    // fragment for row `row` (0..BR) of k-block kk (8 k's): 4 values k = 8kk + 4kh + q
    __device__ static __forceinline__ void frag(float (&f)[4], const float* s, int row, int kk, int kh) {
        if constexpr (KC) {
            const float4 v = ld4(s + row * KPAD + kk * 8 + 4 * kh);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = s[(kk * 8 + 4 * kh + q) * BR + row];
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// Main loop without an MFMA-free phase at the tile boundary.  PMC on the plain "load, barrier, multiply" loop
// (4096^3): matrix pipe busy 79 %, waves stalled on ISSUE 81 % and on waits only 7 % -- the two co-resident waves
// of a SIMD advance in lock step, reach their "wait for the global loads, write LDS, barrier, first fragment
// read" phase together, and the pipe idles.  Here
//   * global loads run TWO tiles ahead (two register sets), so the LDS write of tile kt+1 needs no wait,
//   * that write is placed after the MFMAs of k-block 1, the workgroup barrier after k-block 2,
//   * the first fragments of tile kt+1 are read after the barrier and before the MFMAs of k-block 3,
// so every wave always has 16 MFMAs queued behind whatever else it issues.
// Hazards: stage s^1 is written (k-block 1 of iteration kt) only after barrier(kt-1), and its last readers read it
// before that barrier (their k-block-3 fragments are fetched during k-block 2); stage s^1 is read (k-block 3)
// only after barrier(kt), which follows every wave's write.
// `lin`: linear tile index of this workgroup within the problem (after the XCD remap); `ky`: its K slice (split-K).
// PAIR: the launch carries a second operand pair (C = A.B + A2.B2, K-concatenated); without it the selects between the pairs are compiled out
template <int TM, int TN, bool AT, bool BT, bool EDGE, bool PAIR>
__device__ __forceinline__ void gemm2_body(const GemmArgs& g, const int lin, const int ky) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    using TA = Tile<BM, !AT>;
    using TB = Tile<BN, BT>;
    __shared__ __attribute__((aligned(16))) float smem[2 * (TA::ELEMS + TB::ELEMS)];
    float* sA = smem;
    float* sB = smem + 2 * TA::ELEMS;

    const int tiles_n = g.N / BN;
    const int m0 = (lin / tiles_n) * BM;
    const int n0 = (lin % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
#ifdef STATTN_PROBES
    if (g.clk && lin == 0 && ky == 0 && tid == 0) {
        g.clk[0] = __builtin_readcyclecounter(); g.clk[1] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int Kt = g.K + (PAIR && g.A2 ? g.K2 : 0);  // K-concatenated second operand pair (NN / NT)
    int kb = 0, ke = Kt;
    float* Cout = g.C;
    int ldc = g.ldc;
    if (g.kslices > 1) {
        const int per = ((Kt + g.kslices - 1) / g.kslices + BK - 1) / BK * BK;
        kb = ky * per;
        ke = kb + per < Kt ? kb + per : Kt;
        Cout = g.ws + (size_t)ky * g.M * g.N;
        ldc = g.N;
    }
    const int nk = ke > kb ? (ke - kb + BK - 1) / BK : 0;

    float4 ra0[TA::NF4], rb0[TB::NF4], ra1[TA::NF4], rb1[TB::NF4];
    // tile index clamped to the last one: the prefetch is unconditional (no branch around loads, so hipcc keeps
    // counted vmcnt waits); the redundant tail loads hit L2
    auto ktile = [&](int t) { return kb + (t < nk ? t : (nk > 0 ? nk - 1 : 0)) * BK; };

    // tile t of A / B into a register set; with a second operand pair the tiles at k >= K come from it (uniform selects)
    unsigned offA[TA::NF4], offB[TB::NF4], offA2[TA::NF4], offB2[TB::NF4];
    TA::offsets(offA, g.lda, m0, g.M, tid);
    TB::offsets(offB, g.ldb, n0, g.N, tid);
    if (PAIR && g.A2) { TA::offsets(offA2, g.lda2, m0, g.M, tid); TB::offsets(offB2, g.ldb2, n0, g.N, tid); }
    else {
#pragma unroll
        for (int i = 0; i < TA::NF4; ++i) offA2[i] = offA[i];
#pragma unroll
        for (int i = 0; i < TB::NF4; ++i) offB2[i] = offB[i];
    }
    auto loadA = [&](float4 (&r)[TA::NF4], int t) {
        const int k = ktile(t);
        const bool second = PAIR && g.A2 && k >= g.K;
        const int kk = second ? k - g.K : k;
        const float* Xt = TA::tile_base(second ? g.A2 : g.A, second ? g.lda2 : g.lda, m0, kk);
        unsigned off[TA::NF4];
#pragma unroll
        for (int i = 0; i < TA::NF4; ++i) off[i] = second ? offA2[i] : offA[i];
        TA::template gload<EDGE>(r, Xt, off, m0, g.M, kk, second ? ke - g.K : ke, tid);
    };
    auto loadB = [&](float4 (&r)[TB::NF4], int t) {
        const int k = ktile(t);
        const bool second = PAIR && g.A2 && k >= g.K;
        const int kk = second ? k - g.K : k;
        const float* Xt = TB::tile_base(second ? g.B2 : g.B, second ? g.ldb2 : g.ldb, n0, kk);
        unsigned off[TB::NF4];
#pragma unroll
        for (int i = 0; i < TB::NF4; ++i) off[i] = second ? offB2[i] : offB[i];
        TB::template gload<EDGE>(r, Xt, off, n0, g.N, kk, second ? ke - g.K : ke, tid);
    };
    loadA(ra0, 0);
    loadB(rb0, 0);
    TA::sstore(ra0, sA, tid);
    TB::sstore(rb0, sB, tid);
    loadA(ra1, 1);                                                              // tile 1 -> set 1
    loadB(rb1, 1);
    __syncthreads();

    float a[2][TM][4], b[2][TN][4];
    auto frags = [&](int set, const float* cA, const float* cB, int kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) TA::frag(a[set][i], cA, wm * 32 * TM + i * 32 + l31, kk, kh);
#pragma unroll
        for (int j = 0; j < TN; ++j) TB::frag(b[set][j], cB, wn * 32 * TN + j * 32 + l31, kk, kh);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][i][q], b[set][j][q], acc[i][j], 0, 0, 0);
    };
    frags(0, sA, sB, 0);

    // One tile.  The non-MFMA instructions of every k-block are dealt out ONE BY ONE between its MFMAs
    // (sched_group_barrier pipelines: a compile-time interleave), so each of them issues in the 64-clock shadow of the
    // wave's own previous MFMA.  Issued as blocks ("all fragment reads, then 16 MFMAs, then 8 LDS writes + 8 global loads
    // + their address arithmetic, ...", round 1-3) the wave offers the matrix pipe nothing for the length of a block,
    // and its SIMD partner -- the wave of the co-resident workgroup, same code, same phase -- is in the same block at the
    // same time: MfmaUtil 76-80 %.  Measured (tools/gemm_ab.sh, 128 x 128 tile): 4096^3 128.6 -> 137.7 TFLOP/s, TN
    // 4096 x 1024 x 13312 128.0 -> 138.5; 64 x 64 tile + 2-5 %.  (Starting the co-resident workgroups out of phase with
    // an s_sleep instead: no change.)  Quotas per MFMA: what the k-block has to place, divided by its MFMAs.
    constexpr int NM = 4 * TM * TN;                                        // MFMAs of a k-block
    constexpr int Q_DSR = (TM + 4 * TN + NM - 1) / NM;                     // fragment reads (b128 / 4 x b32 per fragment)
    constexpr int Q_DSW = (4 * (TM + TN) + NM - 1) / NM;                   // LDS writes of the next tile (b128, at worst split in two)
    constexpr int Q_VM = (2 * (TM + TN) + NM - 1) / NM;                    // global loads of tile kt + 3
    constexpr int Q_VALU = (24 * (TM + TN) + NM - 1) / NM;                 // their address arithmetic (VALU + SALU)
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define STATTN_GEMM2_TILE(KT, RA_NEXT, RB_NEXT, RA_FAR, RB_FAR)                                              \
    {                                                                                                         \
        const int st = (KT) & 1;                                                                              \
        const float* cA = sA + st * TA::ELEMS;                                                                \
        const float* cB = sB + st * TB::ELEMS;                                                                \
        float* nA = sA + (st ^ 1) * TA::ELEMS;                                                                \
        float* nB = sB + (st ^ 1) * TB::ELEMS;                                                                \
        /* k-block 0 (fragments of k-block 1 requested under it) */                                           \
        frags(1, cA, cB, 1);                                                                                  \
        mfmas(0);                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < NM; ++i_) { SGB(0x8, 1); SGB(0x100, Q_DSR); }                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        /* k-block 1; tile KT+1 (already in registers: fetched two tiles ahead) goes to the other stage */    \
        frags(0, cA, cB, 2);                                                                                  \
        mfmas(1);                                                                                             \
        TA::sstore(RA_NEXT, nA, tid);                                                                         \
        TB::sstore(RB_NEXT, nB, tid);                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < NM; ++i_) { SGB(0x8, 1); SGB(0x100, Q_DSR); SGB(0x200, Q_DSW); } \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        /* k-block 2; the freed register set starts fetching tile KT+3 (tile KT+2 lives in the FAR set) */    \
        loadA(RA_NEXT, (KT) + 3);                                                                             \
        loadB(RB_NEXT, (KT) + 3);                                                                             \
        frags(1, cA, cB, 3);                                                                                  \
        mfmas(0);                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < NM; ++i_) { SGB(0x8, 1); SGB(0x100, Q_DSR); SGB(0x6, Q_VALU); SGB(0x20, Q_VM); } \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __syncthreads();                                                                                      \
        /* k-block 3, with the first fragments of tile KT+1 requested under it */                             \
        frags(0, nA, nB, 0);                                                                                  \
        mfmas(1);                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < NM; ++i_) { SGB(0x8, 1); SGB(0x100, Q_DSR); }                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }

    // register-set rotation: on entry of tile kt, set (kt+1)&1 ... we keep it simple with a 2-tile unroll:
    //   tile kt   (even): NEXT = set1 (tile kt+1), after its store set1 is refilled with tile kt+3
    //   tile kt+1 (odd) : NEXT = set0 (tile kt+2), ...   set0 must then hold tile kt+2: fetched during tile kt-1.
    // Prologue therefore also fetches tile 2 into set 0 after its LDS store.
    loadA(ra0, 2);
    loadB(rb0, 2);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        STATTN_GEMM2_TILE(kt, ra1, rb1, ra0, rb0)
        STATTN_GEMM2_TILE(kt + 1, ra0, rb0, ra1, rb1)
    }
    if (kt < nk) STATTN_GEMM2_TILE(kt, ra1, rb1, ra0, rb0)
#undef STATTN_GEMM2_TILE
#undef SGB
#ifdef STATTN_PROBES
    if (g.clk && lin == 0 && ky == 0 && tid == 0) {
        g.clk[2] = __builtin_readcyclecounter(); g.clk[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    epilogue<TM, TN>(g, acc, m0, n0, wm, wn, l31, kh, Cout, ldc);
}

template <int TM, int TN, bool AT, bool BT, bool EDGE, bool PAIR>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const GemmArgs g) {
    gemm2_body<TM, TN, AT, BT, EDGE, PAIR>(g, xcd_linear(blockIdx.x, gridDim.x, g.xcd_remap), blockIdx.y);
}

// Several independent problems in one launch (same transposes, no split-K): the tiles of the small ones fill the
// tail of the large one instead of running as under-filled launches of their own (a 1664 x 1024 projection is 416
// tiles for 1024 resident workgroups).  tile_start[p] .. tile_start[p + 1] = the tiles of problem p.
// Work balance: the tiles of every problem are dealt to the eight XCDs separately (block b runs on XCD b % 8, which walks
// its share of problem 0, then of problem 1, ...), so each XCD gets the same mix of long-K and short-K tiles and the
// short ones fill its tail.  (One contiguous run of the concatenated tile list per XCD left the XCDs that drew the
// K = 4096 tiles working 40 % longer than the others.)
template <int TM, int TN, bool AT, bool BT, bool EDGE, bool PAIR>
__global__ __launch_bounds__(256, 2) void gemm2_group_kernel(const GemmGroup G) {
    int p, lin;
    if (!group_locate(G, blockIdx.x, p, lin)) return;                // padding block of this XCD
    gemm2_body<TM, TN, AT, BT, EDGE, PAIR>(G.g[p], lin, 0);
}

// C[i] (+)= alpha * sum_z ws[z][i]   (fixed summation order: deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int ldc, int M, int N,
                                     int slices, float alpha, int accumulate) {
    const size_t n4 = (size_t)M * N / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ld4(ws + 4 * i);
#pragma unroll 4
        for (int z = 1; z < slices; ++z) {
            const float4 b = ld4(ws + (size_t)z * M * N + 4 * i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const size_t e = 4 * i, row = e / N, col = e % N;
        float* c = C + row * ldc + col;
        float4 o = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
        if (accumulate) { const float4 p = ld4(c); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        st4(c, o);
    }
}

// the same with the fused epilogue of gemm_common.h (bias, bias2, add, rowadd, tanh, Cact, mul, accumulate)
__global__ void splitk_reduce_epi_kernel(const GemmArgs g, int slices) {
    const size_t n4 = (size_t)g.M * g.N / 4, MN = (size_t)g.M * g.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ld4(g.ws + 4 * i);
#pragma unroll 4
        for (int z = 1; z < slices; ++z) {
            const float4 b = ld4(g.ws + (size_t)z * MN + 4 * i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const size_t e = 4 * i, row = e / g.N, col = e % g.N;
        float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t c = col + q;
            float x = g.alpha * v[q] + (g.bias ? g.bias[c] : 0.f) + (g.bias2 ? g.bias2[c] : 0.f);
            if (g.add) x += g.add[row * g.ldadd + c];
            if (g.rowadd) x += g.rowadd[(row / g.rowgroup) * g.ldrow + c];
            if (g.act == 1) x = fast_tanh(x);
            if (g.Cact) g.Cact[row * g.ldcact + c] = x;
            if (g.mul) x *= g.mul[row * g.ldmul + c];
            float* o = g.C + row * g.ldc + c;
            if (g.accumulate) x += *o;
            *o = x;
        }
    }
}

template <int TM, int TN>
hipError_t launch_cfg(hipStream_t s, const GemmArgs& g, bool tA, bool tB) {
    const int BM = 64 * TM, BN = 64 * TN;
    const int tiles = ((g.M + BM - 1) / BM) * (g.N / BN);
    dim3 grid(tiles, g.kslices > 1 ? g.kslices : 1), block(256);
    {
        // predicate-free loads when no tile straddles an edge that is NOT handled by clamping
        const int Kt = g.K + (g.A2 ? g.K2 : 0);
        int per = Kt;
        if (g.kslices > 1) per = ((Kt + g.kslices - 1) / g.kslices + BK - 1) / BK * BK;
        const bool edge = (Kt % BK != 0) || (g.kslices > 1 && (size_t)per * (g.kslices - 1) >= (size_t)Kt) ||
                          (tA && g.M % BM != 0);
        if (edge) {
            if (!tA && !tB) { if (g.A2) hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, false, true, true>), grid, block, 0, s, g); else hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, false, true, false>), grid, block, 0, s, g); }
            else if (!tA && tB) { if (g.A2) hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, true, true, true>), grid, block, 0, s, g); else hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, true, true, false>), grid, block, 0, s, g); }
            else if (tA && !tB) hipLaunchKernelGGL((gemm2_kernel<TM, TN, true, false, true, false>), grid, block, 0, s, g);
            else return hipErrorInvalidValue;
        } else {
            if (!tA && !tB) { if (g.A2) hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, false, false, true>), grid, block, 0, s, g); else hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, false, false, false>), grid, block, 0, s, g); }
            else if (!tA && tB) { if (g.A2) hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, true, false, true>), grid, block, 0, s, g); else hipLaunchKernelGGL((gemm2_kernel<TM, TN, false, true, false, false>), grid, block, 0, s, g); }
            else if (tA && !tB) hipLaunchKernelGGL((gemm2_kernel<TM, TN, true, false, false, false>), grid, block, 0, s, g);
            else return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

}  // namespace

void gemm_defaults(GemmArgs& g) {
    g = GemmArgs{};
    g.alpha = 1.f;
    g.rowgroup = 1;
}

// Clock probe (tools only: build with `make PROBES=1`, run with STATTN_GEMM_CLK=1): block 0 of every GEMM launch
// records the shader cycle counter and the 100 MHz wall clock at its start and end; gemm_clock_dump() prints shape,
// block-0 duration and the implied shader clock for every launch since the last dump.  Compiled out of product builds.
#ifndef STATTN_PROBES
void gemm_clock_dump() {}
#else
namespace {
struct ClkRec { int M, N, K, tA, tB; };
constexpr int CLK_SLOTS = 4096;
long long* g_clk_dev = nullptr;
std::vector<ClkRec>* g_clk_rec = nullptr;
}  // namespace

void gemm_clock_dump() {
    if (!g_clk_dev || !g_clk_rec || g_clk_rec->empty()) return;
    (void)hipDeviceSynchronize();
    std::vector<long long> c(4 * g_clk_rec->size());
    if (hipMemcpy(c.data(), g_clk_dev, c.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    for (size_t i = 0; i < g_clk_rec->size(); ++i) {
        const ClkRec& r = (*g_clk_rec)[i];
        const double us = (double)(c[4 * i + 3] - c[4 * i + 1]) / 100.0;
        fprintf(stderr, "[stattn gemm clk] %c%c M=%6d N=%6d K=%6d  block0 %8.1f us  %5.0f MHz\n", r.tA ? 'T' : 'N',
                r.tB ? 'T' : 'N', r.M, r.N, r.K, us, us > 0 ? (double)(c[4 * i + 2] - c[4 * i]) / us : 0.0);
    }
    g_clk_rec->clear();
}
#endif

hipError_t launch_splitk_reduce_epilogue(hipStream_t s, const GemmArgs& g, int slices) {
    const size_t n4 = (size_t)g.M * g.N / 4;
    int nb = (int)((n4 + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_reduce_epi_kernel, dim3(nb), dim3(256), 0, s, g, slices);
    return hipGetLastError();
}

hipError_t launch_splitk_reduce(hipStream_t s, const float* ws, float* C, int ldc, int M, int N, int slices, float alpha, int accumulate) {
    const size_t n4 = (size_t)M * N / 4;
    int nb = (int)((n4 + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, s, ws, C, ldc, M, N, slices, alpha, accumulate);
    return hipGetLastError();
}

template <int TM, int TN>
hipError_t launch_group_cfg(hipStream_t s, const GemmArgs* gs, int n, bool tA, bool tB) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    GemmGroup G{};
    static const char* noremap = sw_tool("STATTN_GEMM_NOREMAP");
    bool edge = false, pair = false;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs g = gs[i];
        if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.N % BN != 0 || (!tA && g.K % 4 != 0) || (tA && g.M % 4 != 0)) return hipErrorInvalidValue;   // (tB: K % 4 == 0 as well, covered)
        if (g.A2 && (tA || g.K % BK != 0 || g.K2 % BK != 0 || g.K2 <= 0)) return hipErrorInvalidValue;   // second operand pair: NN / NT, whole k-tiles
        g.kslices = 1; g.ws = nullptr; g.xcd_remap = noremap ? 0 : 1; g.clk = nullptr;
        edge = edge || g.K % BK != 0 || (tA && g.M % BM != 0);
        pair = pair || g.A2 != nullptr;
        G.g[i] = g;
        G.tile_start[i] = tiles;
        tiles += ((g.M + BM - 1) / BM) * (g.N / BN);
    }
    G.tile_start[n] = tiles; G.n = n;
    int per_xcd = 0;                                            // blocks an XCD may have to walk: sum of its largest shares
    for (int i = 0; i < n; ++i) per_xcd += (G.tile_start[i + 1] - G.tile_start[i] + NXCD - 1) / NXCD;
    const dim3 grid(per_xcd * NXCD);
    if (tA) {
        if (edge) hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, true, false, true, false>), grid, dim3(256), 0, s, G);
        else hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, true, false, false, false>), grid, dim3(256), 0, s, G);
    } else if (tB) {
        if (edge) { if (pair) hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, true, true, true>), grid, dim3(256), 0, s, G); else hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, true, true, false>), grid, dim3(256), 0, s, G); }
        else { if (pair) hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, true, false, true>), grid, dim3(256), 0, s, G); else hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, true, false, false>), grid, dim3(256), 0, s, G); }
    } else {
        if (edge) { if (pair) hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, false, true, true>), grid, dim3(256), 0, s, G); else hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, false, true, false>), grid, dim3(256), 0, s, G); }
        else { if (pair) hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, false, false, true>), grid, dim3(256), 0, s, G); else hipLaunchKernelGGL((gemm2_group_kernel<TM, TN, false, false, false, false>), grid, dim3(256), 0, s, G); }
    }
    return hipGetLastError();
}


hipError_t launch_gemm_group(hipStream_t s, const GemmArgs* gs, int n, bool tA, bool tB) {
    if (n < 1 || n > GEMM_GROUP_MAX || (tA && tB)) return hipErrorInvalidValue;
    if (n == 1) return launch_gemm(s, gs[0], tA, tB);
    {
        bool split = true;
        for (int i = 0; i < n; ++i) split = split && gs[i].split && !gs[i].A2 && gemm_split_supported(gs[i], tA, tB);   // (the split kernels have no second operand pair)
        if (split) return launch_gemm_split_group(s, gs, n, tA, tB);
    }
    // Tile of a grouped launch.  64 x 64 is the default (13 tiles per CU on the biggest projection: no tail).  When every
    // problem has N % 128 == 0 the 128 x 128 tile halves the L2 -> LDS traffic per flop; it is taken when the GROUP's
    // tile count fills whole rounds of the resident workgroups (the small problems are what fills the big one's tail).
    static const char* gt = sw_tool("STATTN_GROUP_TILE");          // "22" / "11": force (tools)
    bool n128 = true;
    long t22 = 0;
    for (int i = 0; i < n; ++i) { n128 = n128 && gs[i].N % 128 == 0 && gs[i].M >= 128; t22 += (long)((gs[i].M + 127) / 128) * (gs[i].N / 128); }
    bool big = false;
    if (n128 && !(gt && gt[0] == '1')) {
        const double rounds = t22 / 512.0;                          // two resident 128 x 128 workgroups per CU
        const double q = rounds / (double)(long)(rounds + 0.999999);
        big = (gt && gt[0] == '2') || (rounds >= 2.0 && q >= 0.93);
    }
    return big ? launch_group_cfg<2, 2>(s, gs, n, tA, tB) : launch_group_cfg<1, 1>(s, gs, n, tA, tB);
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs& gin, bool tA, bool tB) {
    if (gin.split && !gin.A2 && gemm_split_supported(gin, tA, tB)) return launch_gemm_split(s, gin, tA, tB);
    GemmArgs g = gin;
    g.kslices = 1;
#ifdef STATTN_PROBES
    static const char* clk = sw_tool("STATTN_GEMM_CLK");
    if (clk) {
        if (!g_clk_dev) {
            if (hipMalloc(&g_clk_dev, CLK_SLOTS * 4 * sizeof(long long)) != hipSuccess) g_clk_dev = nullptr;
            g_clk_rec = new std::vector<ClkRec>();
        }
        if (g_clk_dev && g_clk_rec->size() < CLK_SLOTS) {
            g.clk = g_clk_dev + 4 * g_clk_rec->size();
            g_clk_rec->push_back(ClkRec{g.M, g.N, g.K, tA, tB});
        }
    }
#endif
    static const char* noremap = sw_tool("STATTN_GEMM_NOREMAP");
    g.xcd_remap = noremap ? 0 : 1;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hipSuccess;
    if (g.N % 64 != 0 || (!tA && g.K % 4 != 0)) return hipErrorInvalidValue;
    if (tB && g.K % 4 != 0) return hipErrorInvalidValue;
    if (tA && (g.M % 4 != 0)) return hipErrorInvalidValue;
    auto blocks = [&](int bm, int bn) { return ((g.M + bm - 1) / bm) * (g.N / bn); };
    const bool n128 = (g.N % 128 == 0);
    if (g.A2 && (tA || g.K % BK != 0 || g.K2 % BK != 0 || g.K2 <= 0)) return hipErrorInvalidValue;   // NN or NT
    const int Kt = g.K + (g.A2 ? g.K2 : 0);
    if (g.ws) {
        // split-K (weight-gradient shapes, and small-output forward GEMMs that would leave most CUs idle): 64x64 tiles,
        // ~4 resident blocks per CU; the partial tiles are summed in a fixed order, by the plain reduction or -- when the
        // launch has a fused epilogue -- by the one that applies it
        const bool epi = g.bias || g.bias2 || g.add || g.rowadd || g.mul || g.act || g.Cact;
        const int t11 = blocks(64, 64);
        // Long-K weight gradients with many output tiles (dff_local_W = ctxl^T.dL: 4096 x 1024 x 13312): 128 x 128 tiles,
        // K cut so that two workgroups per CU are resident -- half the L2 -> LDS traffic per flop of the 64 x 64 tile at the
        // same fill (982 -> 909 us).  Shorter K or fewer tiles lose (dU 1024 x 4096 x 1920: 154 -> 163 us; da NT 221 -> 275).
        static const char* nobig = sw_tool("STATTN_SPLITK_NOBIG");       // A/B switch for tools
        if (!nobig && !epi && tA && n128 && g.M % 128 == 0 && Kt >= 2048 && blocks(128, 128) >= 128) {
            const int t22 = blocks(128, 128);
            int ks = (512 + t22 - 1) / t22;
            if (ks > Kt / 512) ks = Kt / 512;
            while (ks > 1 && (size_t)ks * g.M * g.N > g.ws_floats) --ks;
            if (t22 * ks >= 384 && (g.M * (size_t)g.N) % 4 == 0 && g.ldc % 4 == 0) {
                if (ks > 1) {
                    g.kslices = ks;
                    hipError_t e = launch_cfg<2, 2>(s, g, tA, tB);
                    if (e != hipSuccess) return e;
                    return launch_splitk_reduce(s, g.ws, g.C, g.ldc, g.M, g.N, ks, g.alpha, g.accumulate);
                }
                return launch_cfg<2, 2>(s, g, tA, tB);
            }
        }
        if (t11 < 768 && Kt >= 1024) {
            int ks = (1024 + t11 - 1) / t11;
            // 1024 workgroups are resident at once (four per CU): a slice count whose grid overshoots that by a fraction runs
            // a nearly empty second round (da = dlogit.Wo^T, 240 tiles: 5 slices = 1200 workgroups) -- round down when that
            // still fills the chip
            static const char* ceil_rule = sw_tool("STATTN_SPLITK_CEIL");   // A/B switch for tools: the rule of rounds 1-3
            if (!ceil_rule && ks * t11 > 1024 && (ks - 1) >= 2 && (ks - 1) * t11 >= 832) --ks;
            static const char* fks = sw_tool("STATTN_FWD_KS");            // tools: slice count of epilogue-carrying split-K launches
            if (epi && fks) ks = atoi(fks);
            else if (ks > Kt / 512) ks = Kt / 512;
            // (a launch with an epilogue pays for it once more in the reduction: slices of at least 1024 -- measured on the
            // readout pair 1920 x 512 x 2048: 2 / 3 / 4 / 6 / 8 slices 55.5 / 55.9 / 57.4 / 58.9 / 61.9 us, unsplit 69.0)
            // (a problem of a few dozen tiles -- one video's F -> D projection, 208 x 512 x 4096 -- is a latency chain per
            // workgroup: slices of 512 there)
            if (epi && !fks && ks > Kt / 1024 && t11 > 64) ks = Kt / 1024;
            if (ks > 32) ks = 32;
            while (ks > 1 && (size_t)ks * g.M * g.N > g.ws_floats) --ks;
            if (ks > 1 && (g.M * (size_t)g.N) % 4 == 0 && g.ldc % 4 == 0 && g.N % 4 == 0) {
                g.kslices = ks;
                hipError_t e = launch_cfg<1, 1>(s, g, tA, tB);
                if (e != hipSuccess) return e;
                return epi ? launch_splitk_reduce_epilogue(s, g, ks)
                           : launch_splitk_reduce(s, g.ws, g.C, g.ldc, g.M, g.N, ks, g.alpha, g.accumulate);
            }
        }
    }
    // Measured on MI355X (tools/gemm_probe.py, profiles/r01_gemm_probe.txt): on a 4096^3 problem the 128x128 /
    // 128x64 / 64x64 tiles reach 125 / 118 / 114 TFLOP/s, but what decides the decoder's shapes (N = 1024,
    // M = 13312: 3.25 big tiles per CU) is the tail: whole tiles per CU quantise, so the big tile is only used
    // when its per-CU tile count is (nearly) integral; otherwise the 64x64 tile (4 resident blocks per CU).
    static const char* force = sw_tool("STATTN_GEMM_TILE");       // probing only
    if (force && force[0] == '2' && force[1] == '2' && n128) return launch_cfg<2, 2>(s, g, tA, tB);
    if (force && force[0] == '2' && force[1] == '1') return launch_cfg<2, 1>(s, g, tA, tB);
    if (force && force[0] == '1') return launch_cfg<1, 1>(s, g, tA, tB);
    if (n128 && g.M > 64) {
        const double per_cu = blocks(128, 128) / 256.0;
        const double q = per_cu / (double)(long)(per_cu + 0.999999);
        if (per_cu >= 2.0 && q >= 0.95) return launch_cfg<2, 2>(s, g, tA, tB);
    }
    return launch_cfg<1, 1>(s, g, tA, tB);
}

}  // namespace stattn
