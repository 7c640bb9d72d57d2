// fp32 GEMM on the bf16 matrix cores of gfx950 ("split" GEMM, GemmArgs::split).
//
// The fp32-input MFMA of CDNA4 runs at 1/16 of the bf16 rate (157 against 2516 TFLOP/s dense).  An fp32 number is
// EXACTLY the sum of three bf16 numbers -- a = a0 + a1 + a2 with a0 = bf16(a) rounded to nearest, a1 = bf16(a - a0),
// a2 = a - a0 - a1 (bf16 keeps the fp32 exponent range, so nothing under- or overflows) -- hence
//     a.b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + [a1b2 + a2b1 + a2b2].
// Every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32.  This kernel issues the SIX products
// outside the bracket; with |a1| <= 2^-9 |a| and |a2| <= 2^-17 |a| the bracket is below 2^-25 |a.b| per product, with
// random sign: less than ONE fp32 rounding of that product, which any fp32 dot product commits per term anyway.
// Measured against float64 the result is as close as the fp32-MFMA kernel's (tests/test_gpu_split.py).  Six bf16 MFMAs
// per k-block of 16 cost 6 x 32 cycles for 32x32x16 multiply-adds against 8 x 64 cycles on the fp32 pipe: the roof is
// 2516 / 6 = 419 TFLOP/s of fp32 work.  Not IEEE in two corners that the decoder never visits: an infinite operand gives
// NaN (inf - inf in the split) where fp32 gives inf -- as does a finite one within 0.4 % of FLT_MAX, whose first term rounds
// to infinity -- and below about 1e-33 the third term falls into the bf16 subnormals (lost or flushed).
//
// Same problem description, transposes and epilogue as gemm.hip (model_attention.py:322-335, 416, 664-667, 687-705
// and their gradients); selected per handle (stattn_options.precision = 2; also the backward GEMMs of bf16 handles),
// never silently.  Operand shapes the kernel does not take (N % 128 != 0, unaligned k-contiguous operands, operands
// of 2 GB or more) run on the fp32-MFMA kernel.
//
// Tiling: WM x WN waves of 64 x 64 (32 x 32 in the small configuration) = 2 x 2 accumulators of 32 x 32 each; block
// tile 64 x 64 (2 x 2 waves of 32 x 32), 128 x 128 (2 x 2) or 256 x 128 (4 x 2: eight waves), BK = 16 (one MFMA
// k-block).  Operands are read from global memory as fp32 four tiles ahead, split in registers (v_cvt_pk_bf16_f32,
// shift / and, subtract) while the MFMAs of the current tile run, and written as three bf16 planes to an LDS ring of
// three stages; MFMA operands are fetched one tile ahead; one workgroup barrier per tile.
// LDS image of a plane: [k-group of 8][row][8 bf16]: a lane's MFMA operand (8 consecutive k of one row) is one
// 16-byte read and the 16 lanes of a ds_read_b128 group read consecutive rows = 256 contiguous bytes.  Both operand
// kinds are brought to that image in the write pass: k-contiguous operands with one 16-byte global load per thread and
// row (4 lanes cover 64 contiguous bytes of a row), row-contiguous ones ([K][rows]) with 4-byte loads along the rows
// (a wave reads 256 contiguous bytes per k) so that a thread ends up holding consecutive k of its row.
#include "gemm_common.h"

#include <cstdlib>
#include <cstring>

namespace stattn {

namespace {

using namespace gemm_common;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int SBK = 16;
#ifndef GS_VARIANT
#define GS_VARIANT 0          // tools/gemm_split_probe.hip builds ablations 1..8; the product is 0
#endif

// the upper halves of two fp32 words as one dword (low half = first value)
__device__ __forceinline__ unsigned top16(float first, float second) {
    return __builtin_amdgcn_perm(__float_as_uint(second), __float_as_uint(first), 0x07060302u);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 values rounded to nearest-even bf16, as one dword (low half = first value): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned rne2(float first, float second) {
    bf16x2 v;
    v[0] = (__bf16)first; v[1] = (__bf16)second;
    return __builtin_bit_cast(unsigned, v);
}
// (a0, a1) -> three dwords of bf16 pairs with a = h + m + l EXACTLY: h = bf16(a) to nearest, m = bf16(a - h), l = a - h - m
// (the first remainder has at most 16 significant bits, the second at most 8, so l is exact).  |m| <= 2^-9 |a|,
// |l| <= 2^-17 |a|.
__device__ __forceinline__ void split3_pair(float a0, float a1, unsigned& H, unsigned& M, unsigned& L) {
    if constexpr (GS_VARIANT == 1) { H = M = L = top16(a0, a1); return; }
    H = rne2(a0, a1);
    const float r0 = a0 - __uint_as_float(H << 16), r1 = a1 - __uint_as_float(H & 0xffff0000u);
    M = rne2(r0, r1);
    const float s0 = r0 - __uint_as_float(M << 16), s1 = r1 - __uint_as_float(M & 0xffff0000u);
    L = top16(s0, s1);
}

// KC: operand stored k-contiguous ([rows][K]); otherwise row-contiguous ([K][rows]).
template <int BR, bool KC, int NTH = 256>
struct STile {
    static constexpr int KGSZ = BR * 16 + 64;        // bytes of one k-group; the pad keeps the 8-byte writes of the
                                                     // k-contiguous pass on 32 distinct banks
    static constexpr int PLANE = 2 * KGSZ;
    static constexpr int BYTES = 3 * PLANE;
    static constexpr int NR = BR * 16 / NTH;         // fp32 values per thread and tile
    static constexpr int KPT = NR;                   // row-contiguous pass: consecutive k per thread (16, 8 or 4)

    // Buffer loads: the operand is one resource (SGPRs), the per-thread byte offset is loop-invariant (voff), the k
    // position of a load is a scalar offset -- a tile costs no address arithmetic on the vector ALU, which the split
    // already loads.  Hence: operand extents < 4 GB (gemm_split_supported).
    // `bytes` = the extent of the operand (bounds the per-thread part of the address)
    __device__ static __forceinline__ __amdgpu_buffer_rsrc_t resource(const float* X, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)bytes, 0x00020000);
    }
    // per-thread byte offsets of the loads of a tile (KC: one per 16-byte load; else one)
    static constexpr int NOFF = KC ? NR / 4 : 1;
    __device__ static __forceinline__ void voffsets(unsigned (&off)[NOFF], int ld, int r0, int rows_total, int tid) {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NR / 4; ++i) {
                const int idx = tid + i * NTH, rr = idx >> 2, kq = idx & 3;
                int row = r0 + rr;
                row = row < rows_total ? row : rows_total - 1;          // clamped rows are never stored
                off[i] = (unsigned)row * (unsigned)ld * 4u + 16u * kq;
            }
        } else {
            off[0] = (unsigned)(r0 + tid % BR) * 4u;
        }
    }
    template <bool EDGE>
    __device__ static __forceinline__ void gload(float (&r)[NR], __amdgpu_buffer_rsrc_t X, const unsigned (&off)[NOFF], int ld,
                                                 int r0, int rows_total, int k0, int K, int tid) {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NR / 4; ++i) {
                const int kq = (tid + i * NTH) & 3;
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (!EDGE || k0 + 4 * kq < K) v = __builtin_amdgcn_raw_buffer_load_b128(X, off[i], k0 * 4, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) r[4 * i + q] = __uint_as_float(v[q]);
            }
        } else {
            const int kp = __builtin_amdgcn_readfirstlane(tid / BR);   // BR >= 64: uniform over a wave
            const bool rowok = !EDGE || r0 + tid % BR < rows_total;
            const unsigned stride = (unsigned)ld * 4u;
            unsigned so = (unsigned)(k0 + KPT * kp) * stride;
#pragma unroll
            for (int t = 0; t < KPT; ++t) {
                unsigned v = 0u;
                if (!EDGE || (rowok && k0 + KPT * kp + t < K)) v = __builtin_amdgcn_raw_buffer_load_b32(X, off[0], so, 0);
                r[t] = __uint_as_float(v);
                so += stride;
            }
        }
    }

    // the LDS write pass in NPARTS independent pieces (spread over the MFMAs of a tile by the caller)
    static constexpr int NPARTS = NR / 4;            // four values (8 bytes per plane) per piece
    template <int PART>
    __device__ static __forceinline__ void sstore_part(const float (&r)[NR], unsigned char* s, int tid) {
        if constexpr (PART < NPARTS) {
            unsigned h[2], m[2], l[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) split3_pair(r[4 * PART + 2 * q], r[4 * PART + 2 * q + 1], h[q], m[q], l[q]);
            unsigned char* d;
            if constexpr (KC) {
                const int idx = tid + PART * NTH, rr = idx >> 2, kq = idx & 3;
                d = s + (kq >> 1) * KGSZ + rr * 16 + 8 * (kq & 1);
            } else {
                const int rr = tid % BR, kq = (tid / BR) * NPARTS + PART;      // k = 4 kq .. 4 kq + 3
                d = s + (kq >> 1) * KGSZ + rr * 16 + 8 * (kq & 1);
            }
            *reinterpret_cast<u32x2*>(d) = u32x2{h[0], h[1]};
            *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{m[0], m[1]};
            *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{l[0], l[1]};
        }
    }
    __device__ static __forceinline__ void sstore(const float (&r)[NR], unsigned char* s, int tid) {
        sstore_part<0>(r, s, tid);
        sstore_part<1>(r, s, tid);
        sstore_part<2>(r, s, tid);
        sstore_part<3>(r, s, tid);
    }

    // the three bf16 terms of row `row`, k-group kh (k = 8 kh .. 8 kh + 7)
    __device__ static __forceinline__ void frag(bf16x8 (&f)[3], const unsigned char* s, int row, int kh) {
        const unsigned char* p = s + kh * KGSZ + row * 16;
#pragma unroll
        for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8*>(p + t * PLANE);
    }
};

// WM x WN waves of (32 MT) x (32 NT) each; block tile (32 MT WM) x (32 NT WN)
template <int MT, int NT, int WM, int WN, bool AT, bool BT, bool EDGE>
__device__ __forceinline__ void gemm3_body(const GemmArgs& g, const int lin, const int ky) {
    static_assert(MT <= 2 && NT <= 2, "the tile schedule places two operand fetches per side");
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN, NTH = 64 * WM * WN;
    using TA = STile<BM, !AT, NTH>;
    using TB = STile<BN, BT, NTH>;
    static_assert(TA::NPARTS <= 2 && TB::NPARTS <= 2, "two store pieces per operand and tile are scheduled");
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE];       // stage s: A image, then B image

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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int kb = 0, ke = g.K;
    float* Cout = g.C;
    int ldc = g.ldc;
    if (g.kslices > 1) {
        const int per = ((g.K + g.kslices - 1) / g.kslices + 31) / 32 * 32;
        kb = ky * per;
        ke = kb + per < g.K ? kb + per : g.K;
        Cout = g.ws + (size_t)ky * g.M * g.N;
        ldc = g.N;
    }
    const int nk = ke > kb ? (ke - kb + SBK - 1) / SBK : 0;
    // Prefetches past the last tile re-read it (no branch around the loads; the scalar offset of a buffer load is not
    // bounds-checked, so they must stay inside the operand).  EDGE kernels load zeros for k >= ke instead.
    auto ktile = [&](int t) { return GS_VARIANT == 6 ? kb + (t & 1) * SBK : kb + (EDGE || t < nk ? t : (nk > 0 ? nk - 1 : 0)) * SBK; };

    // Pipeline (one barrier per tile): during the MFMAs of tile kt -- operands already in registers -- a wave reads the
    // operands of tile kt + 1 from stage (kt + 1) % 3, splits the fp32 values of tile kt + 2 (loaded two iterations
    // ago) into stage (kt + 2) % 3 and issues the global loads of tile kt + 4.  Stage (kt + 2) % 3 last held tile
    // kt - 1, whose operands every wave fetched before barrier kt - 2; stage (kt + 1) % 3 was completed before barrier
    // kt - 1.
    float ra0[TA::NR], rb0[TB::NR], ra1[TA::NR], rb1[TB::NR];
    const __amdgpu_buffer_rsrc_t rsA = TA::resource(g.A, (unsigned)(AT ? g.K : g.M) * (unsigned)g.lda * 4u);
    const __amdgpu_buffer_rsrc_t rsB = TB::resource(g.B, (unsigned)(BT ? g.N : g.K) * (unsigned)g.ldb * 4u);
    unsigned offA[TA::NOFF], offB[TB::NOFF];
    TA::voffsets(offA, g.lda, m0, g.M, tid);
    TB::voffsets(offB, g.ldb, n0, g.N, tid);
    TA::template gload<EDGE>(ra0, rsA, offA, g.lda, m0, g.M, ktile(0), ke, tid);
    TB::template gload<EDGE>(rb0, rsB, offB, g.ldb, n0, g.N, ktile(0), ke, tid);
    TA::template gload<EDGE>(ra1, rsA, offA, g.lda, m0, g.M, ktile(1), ke, tid);
    TB::template gload<EDGE>(rb1, rsB, offB, g.ldb, n0, g.N, ktile(1), ke, tid);
    TA::sstore(ra0, smem, tid);
    TB::sstore(rb0, smem + TA::BYTES, tid);
    TA::template gload<EDGE>(ra0, rsA, offA, g.lda, m0, g.M, ktile(2), ke, tid);
    TB::template gload<EDGE>(rb0, rsB, offB, g.ldb, n0, g.N, ktile(2), ke, tid);
    TA::sstore(ra1, smem + STAGE, tid);
    TB::sstore(rb1, smem + STAGE + TA::BYTES, tid);
    TA::template gload<EDGE>(ra1, rsA, offA, g.lda, m0, g.M, ktile(3), ke, tid);
    TB::template gload<EDGE>(rb1, rsB, offB, g.ldb, n0, g.N, ktile(3), ke, tid);
    __syncthreads();

    bf16x8 a0[MT][3], b0[NT][3], a1[MT][3], b1[NT][3];
    auto frags = [&](bf16x8 (&a)[MT][3], bf16x8 (&b)[NT][3], int stage) {
        const unsigned char* cA = smem + stage * STAGE;
        const unsigned char* cB = cA + TA::BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) TA::frag(a[i], cA, wm * 32 * MT + i * 32 + l31, kh);
#pragma unroll
        for (int j = 0; j < NT; ++j) TB::frag(b[j], cB, wn * 32 * NT + j * 32 + l31, kh);
    };
    frags(a0, b0, 0);
    int rd = 1, wr = 2;                    // stages of tiles kt + 1 and kt + 2

    // term pairs, smallest products first; consecutive MFMAs never share an accumulator
#define STATTN_GEMM3_MFMAS(FA, FB, P0, P1)                                                                    \
    if constexpr (GS_VARIANT != 4) _Pragma("unroll") for (int p = P0; p < P1; ++p) {                          \
        const int ta = p == 0 ? 2 : p == 1 ? 0 : p == 2 ? 1 : p == 3 ? 1 : 0;                                 \
        const int tb = p == 0 ? 0 : p == 1 ? 2 : p == 2 ? 1 : p == 3 ? 0 : p == 4 ? 1 : 0;                    \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][ta], FB[j][tb], acc[i][j], 0, 0, 0); \
    }
    // Six scheduling regions per tile, one per term product (MT x NT MFMAs each), so that no wave ever queues a burst of
    // LDS or memory instructions in front of its MFMAs (with all twelve operand reads at the head of the iteration the
    // eight waves of a CU serialised on the LDS right after each barrier: 374 -> 285 TFLOP/s on the MFMA-only loop):
    // regions 0-3 each fetch one of the four operand rows/columns of the NEXT tile (3 reads) and split + store one
    // piece of tile kt + 2 (3 writes); regions 4-5 carry the global loads of tile kt + 4.
// the side work of a region dealt out between its MFMAs instead of issued in front of them (sched_group_barrier pipeline,
// as in gemm.hip; here + 1-3 %: dW_local 203.9 -> 208, dWo 159 -> 164, CL.Wclt 94 -> 101 TFLOP/s -- this kernel is bound by its LDS
// store path and the clock it can hold, section 12 of DESIGN.md).  -DGS_SGB=0 is the A/B build (tools/build_gemm_var.sh).
#ifndef GS_SGB
#define GS_SGB 1
#define GS_SGB_VALU 6
#endif
#if GS_SGB
#define STATTN_GEMM3_DEAL _Pragma("unroll") for (int i_ = 0; i_ < MT * NT; ++i_) {                                \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x2, GS_SGB_VALU, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); \
        __builtin_amdgcn_sched_group_barrier(0x20, 2, 0); }
#else
#define STATTN_GEMM3_DEAL
#endif
#define STATTN_GEMM3_TILE(KT, FA, FB, FA_NEXT, FB_NEXT, RA, RB)                                               \
    {                                                                                                         \
        const unsigned char* cA = smem + rd * STAGE;                                                          \
        const unsigned char* cB = cA + TA::BYTES;                                                             \
        unsigned char* wA = smem + wr * STAGE;                                                                \
        unsigned char* wB = wA + TA::BYTES;                                                                   \
        constexpr bool LD = GS_VARIANT != 5, ST = GS_VARIANT != 2 && GS_VARIANT != 5;                         \
        constexpr bool GL = GS_VARIANT != 3 && GS_VARIANT != 5 && GS_VARIANT != 8;                                               \
        if constexpr (LD) TA::frag(FA_NEXT[0], cA, wm * 32 * MT + l31, kh);                                   \
        if constexpr (ST) TA::template sstore_part<0>(RA, wA, tid);                                           \
        STATTN_GEMM3_MFMAS(FA, FB, 0, 1)                                                                      \
        STATTN_GEMM3_DEAL                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr (LD && MT > 1) TA::frag(FA_NEXT[MT - 1], cA, wm * 32 * MT + (MT - 1) * 32 + l31, kh);    \
        if constexpr (ST) TA::template sstore_part<1>(RA, wA, tid);                                           \
        STATTN_GEMM3_MFMAS(FA, FB, 1, 2)                                                                      \
        STATTN_GEMM3_DEAL                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr (LD) TB::frag(FB_NEXT[0], cB, wn * 32 * NT + l31, kh);                                   \
        if constexpr (ST) TB::template sstore_part<0>(RB, wB, tid);                                           \
        STATTN_GEMM3_MFMAS(FA, FB, 2, 3)                                                                      \
        STATTN_GEMM3_DEAL                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr (LD && NT > 1) TB::frag(FB_NEXT[NT - 1], cB, wn * 32 * NT + (NT - 1) * 32 + l31, kh);    \
        if constexpr (ST) TB::template sstore_part<1>(RB, wB, tid);                                           \
        STATTN_GEMM3_MFMAS(FA, FB, 3, 4)                                                                      \
        STATTN_GEMM3_DEAL                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr (GL) TA::template gload<EDGE>(RA, rsA, offA, g.lda, m0, g.M, ktile((KT) + 4), ke, tid);        \
        if constexpr (GS_VARIANT == 8) { _Pragma("unroll") for (int q = 0; q < TA::NR; ++q) RA[q] *= 1.0001f; }       \
        STATTN_GEMM3_MFMAS(FA, FB, 4, 5)                                                                      \
        STATTN_GEMM3_DEAL                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr (GL) TB::template gload<EDGE>(RB, rsB, offB, g.ldb, n0, g.N, ktile((KT) + 4), ke, tid);        \
        if constexpr (GS_VARIANT == 8) { _Pragma("unroll") for (int q = 0; q < TB::NR; ++q) RB[q] *= 1.0001f; }       \
        STATTN_GEMM3_MFMAS(FA, FB, 5, 6)                                                                      \
        STATTN_GEMM3_DEAL                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        rd = rd == 2 ? 0 : rd + 1;                                                                            \
        wr = wr == 2 ? 0 : wr + 1;                                                                            \
        __syncthreads();                                                                                      \
    }

    // tiles in pairs (the two register sets swap roles): an odd tile count is launched as an EDGE kernel, whose loads
    // deliver zeros for the phantom tile
    for (int kt = 0; kt < nk; kt += 2) {
        STATTN_GEMM3_TILE(kt, a0, b0, a1, b1, ra0, rb0)
        STATTN_GEMM3_TILE(kt + 1, a1, b1, a0, b0, ra1, rb1)
    }
#undef STATTN_GEMM3_TILE
#undef STATTN_GEMM3_DEAL
#undef STATTN_GEMM3_MFMAS
#ifdef STATTN_PROBES
    if (g.clk && lin == 0 && ky == 0 && tid == 0) {
        g.clk[2] = __builtin_readcyclecounter(); g.clk[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    epilogue<MT, NT>(g, acc, m0, n0, wm, wn, l31, kh, Cout, ldc);
}

template <int MT, int NT, int WM, int WN, bool AT, bool BT, bool EDGE>
__global__ __launch_bounds__(64 * WM * WN, MT * NT >= 4 ? 2 : 4) void gemm3_kernel(const GemmArgs g) {
    gemm3_body<MT, NT, WM, WN, AT, BT, EDGE>(g, xcd_linear(blockIdx.x, gridDim.x, g.xcd_remap), blockIdx.y);
}

// several problems in one launch; tiles dealt to the XCDs as in gemm2_group_kernel (gemm.hip)
template <int MT, int NT, int WM, int WN, bool AT, bool BT, bool EDGE>
__global__ __launch_bounds__(64 * WM * WN, MT * NT >= 4 ? 2 : 4) void gemm3_group_kernel(const GemmGroup G) {
    int p, lin;
    if (!group_locate(G, blockIdx.x, p, lin)) return;
    gemm3_body<MT, NT, WM, WN, AT, BT, EDGE>(G.g[p], lin, 0);
}

template <int MT, int NT, int WM, int WN>
hipError_t launch3(hipStream_t s, dim3 grid, const GemmArgs& g, bool tA, bool tB, bool edge) {
    const dim3 block(64 * WM * WN);
    if (edge) {
        if (!tA && !tB) hipLaunchKernelGGL((gemm3_kernel<MT, NT, WM, WN, false, false, true>), grid, block, 0, s, g);
        else if (!tA && tB) hipLaunchKernelGGL((gemm3_kernel<MT, NT, WM, WN, false, true, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm3_kernel<MT, NT, WM, WN, true, false, true>), grid, block, 0, s, g);
    } else {
        if (!tA && !tB) hipLaunchKernelGGL((gemm3_kernel<MT, NT, WM, WN, false, false, false>), grid, block, 0, s, g);
        else if (!tA && tB) hipLaunchKernelGGL((gemm3_kernel<MT, NT, WM, WN, false, true, false>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm3_kernel<MT, NT, WM, WN, true, false, false>), grid, block, 0, s, g);
    }
    return hipGetLastError();
}

template <int MT, int NT, int WM, int WN>
hipError_t launch3_group(hipStream_t s, dim3 grid, const GemmGroup& G, bool tA, bool tB, bool edge) {
    const dim3 block(64 * WM * WN);
    if (edge) {
        if (!tA && !tB) hipLaunchKernelGGL((gemm3_group_kernel<MT, NT, WM, WN, false, false, true>), grid, block, 0, s, G);
        else if (!tA && tB) hipLaunchKernelGGL((gemm3_group_kernel<MT, NT, WM, WN, false, true, true>), grid, block, 0, s, G);
        else hipLaunchKernelGGL((gemm3_group_kernel<MT, NT, WM, WN, true, false, true>), grid, block, 0, s, G);
    } else {
        if (!tA && !tB) hipLaunchKernelGGL((gemm3_group_kernel<MT, NT, WM, WN, false, false, false>), grid, block, 0, s, G);
        else if (!tA && tB) hipLaunchKernelGGL((gemm3_group_kernel<MT, NT, WM, WN, false, true, false>), grid, block, 0, s, G);
        else hipLaunchKernelGGL((gemm3_group_kernel<MT, NT, WM, WN, true, false, false>), grid, block, 0, s, G);
    }
    return hipGetLastError();
}

// loads without predicates need: an even number of whole k-tiles in every K slice, and no partial tile along a
// row-contiguous operand's rows
bool needs_edge(const GemmArgs& g, bool tA, int BM, int per) {
    if (g.kslices > 1) {
        const int last = g.K - per * (g.kslices - 1);
        if (per % (2 * SBK) != 0 || last <= 0 || last % (2 * SBK) != 0) return true;
    }
    return (g.K % (2 * SBK) != 0) || (tA && g.M % BM != 0);
}

}  // namespace

bool gemm_split_supported(const GemmArgs& g, bool tA, bool tB) {
    if (tA && tB) return false;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.N % 128 != 0) return false;
    if (!tA && (g.K % 4 != 0 || g.lda % 4 != 0)) return false;      // 16-byte loads along k
    if (tB && (g.K % 4 != 0 || g.ldb % 4 != 0)) return false;
    // buffer addressing: 32-bit byte offsets into each operand
    const unsigned long long lim = 0x7fffffffull;
    if ((unsigned long long)(tA ? g.K : g.M) * (unsigned)g.lda * 4ull >= lim) return false;
    if ((unsigned long long)(tB ? g.N : g.K) * (unsigned)g.ldb * 4ull >= lim) return false;
    return true;
}

// Tile choice: the 128 x 128 tile does twice the MFMA work per converted operand value, but a problem must offer about
// two of them per CU; below that the 64 x 64 tile (four times the workgroups, 39 KB of LDS each) finishes sooner.
// With at least one 256 x 128 tile per CU the eight-wave workgroup (same 64 x 64 per wave, a quarter less split, store
// and load work per MFMA) is 5-6 % faster still (square 4096: 203 -> 216 TFLOP/s, ff_local 160 -> 168).
static bool big_tiles(long tiles128) { return tiles128 >= 384; }
static bool wide_tiles(long tiles256) { return tiles256 >= 256; }

hipError_t launch_gemm_split(hipStream_t s, const GemmArgs& gin, bool tA, bool tB) {
    GemmArgs g = gin;
    g.kslices = 1;
    static const char* noremap = sw_tool("STATTN_GEMM_NOREMAP");
    g.xcd_remap = noremap ? 0 : 1;
    if (!gemm_split_supported(g, tA, tB)) return hipErrorInvalidValue;
    const int tiles = ((g.M + 127) / 128) * (g.N / 128);
    if (g.ws && !g.bias && !g.add && !g.rowadd && !g.mul && !g.act && !g.Cact && tiles < 384 && g.K >= 1024) {
        // Deterministic split-K for the weight-gradient shapes (few tiles, long K), as in gemm.hip.  The slice count and the
        // tile are chosen together with a small cost model fitted to a sweep over the decoder's shapes (slices 1..16 x both
        // tiles): a CU runs two 128 x 128 workgroups at once, each then taking 2 u per k-step pair, or one alone (1.24 u:
        // a lone workgroup leaves the matrix pipe idle more often), or one 256 x 128 workgroup = the work of two in
        // 1.8 u; per-CU time = that x K / slices (+ a fixed 512 k-steps' worth per workgroup for prologue, epilogue and
        // the reduction).  One slice too many starts another round: da = dlogit.Wo^T ran 9 slices of 60 tiles = 540
        // workgroups for 512 slots (165 us; 8 slices of 32 wide tiles: 128-139 us).
        const int t256 = ((g.M + 255) / 256) * (g.N / 128);
        int kmax = g.K / 512;
        if (kmax > 32) kmax = 32;
        while (kmax > 1 && (size_t)kmax * g.M * g.N > g.ws_floats) --kmax;
        int ks = 1;
        bool wide = false;
        double best = 1e30;
        for (int c = 0; c < 2; ++c) {
            for (int k = 2; k <= kmax; ++k) {
                const int n = ((c ? t256 : tiles) * k + 255) / 256;                      // workgroups on the fullest CU
                const double per_cu = c ? 1.8 * n : (n == 1 ? 1.24 : (double)n);
                const double cost = per_cu * ((double)g.K / k + 512.0);
                if (cost < best * 0.999) { best = cost; ks = k; wide = c != 0; }
            }
        }
        if (const char* fk = sw_tool("STATTN_SPLIT_KS")) {                 // probing only: "<slices>[w]"
            ks = atoi(fk); wide = fk[strlen(fk) - 1] == 'w';
            if (ks > kmax) ks = kmax;
        }
        if (ks > 1 && (g.M * (size_t)g.N) % 4 == 0 && g.ldc % 4 == 0) {
            g.kslices = ks;
            const int per = ((g.K + ks - 1) / ks + 31) / 32 * 32;
            hipError_t e = wide ? launch3<2, 2, 4, 2>(s, dim3(t256, ks), g, tA, tB, needs_edge(g, tA, 256, per))
                                : launch3<2, 2, 2, 2>(s, dim3(tiles, ks), g, tA, tB, needs_edge(g, tA, 128, per));
            if (e != hipSuccess) return e;
            return launch_splitk_reduce(s, g.ws, g.C, g.ldc, g.M, g.N, ks, g.alpha, g.accumulate);
        }
    }
    static const char* force = sw_tool("STATTN_SPLIT_TILE");      // probing only: "1" = 64 x 64, "2" = 128 x 128, "3" = 256 x 128
    const long tiles256 = (long)((g.M + 255) / 256) * (g.N / 128);
    if (force ? force[0] == '3' : wide_tiles(tiles256))
        return launch3<2, 2, 4, 2>(s, dim3((unsigned)tiles256), g, tA, tB, needs_edge(g, tA, 256, g.K));
    const bool big = force ? force[0] == '2' : big_tiles(tiles);
    if (big) return launch3<2, 2, 2, 2>(s, dim3(tiles), g, tA, tB, needs_edge(g, tA, 128, g.K));
    return launch3<1, 1, 2, 2>(s, dim3(((g.M + 63) / 64) * (g.N / 64)), g, tA, tB, needs_edge(g, tA, 64, g.K));
}

hipError_t launch_gemm_split_group(hipStream_t s, const GemmArgs* gs, int n, bool tA, bool tB) {
    if (n < 1 || n > GEMM_GROUP_MAX || (tA && tB)) return hipErrorInvalidValue;
    if (n == 1) return launch_gemm_split(s, gs[0], tA, tB);
    static const char* noremap = sw_tool("STATTN_GEMM_NOREMAP");
    static const char* force = sw_tool("STATTN_SPLIT_TILE");
    long tiles128 = 0, tiles256 = 0;
    for (int i = 0; i < n; ++i) {
        tiles128 += (long)((gs[i].M + 127) / 128) * (gs[i].N / 128);
        tiles256 += (long)((gs[i].M + 255) / 256) * (gs[i].N / 128);
    }
    const bool wide = force ? force[0] == '3' : wide_tiles(tiles256);
    const bool big = wide || (force ? force[0] == '2' : big_tiles(tiles128));
    const int T = wide ? 256 : (big ? 128 : 64), TN_ = big ? 128 : 64;      // tile rows, tile columns
    GemmGroup G{};
    bool edge = false;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs g = gs[i];
        if (!gemm_split_supported(g, tA, tB)) return hipErrorInvalidValue;
        g.kslices = 1; g.ws = nullptr; g.xcd_remap = noremap ? 0 : 1; g.clk = nullptr;
        edge = edge || needs_edge(g, tA, T, g.K);
        G.g[i] = g;
        G.tile_start[i] = tiles;
        tiles += ((g.M + T - 1) / T) * (g.N / TN_);
    }
    G.tile_start[n] = tiles; G.n = n;
    int per_xcd = 0;
    for (int i = 0; i < n; ++i) per_xcd += (G.tile_start[i + 1] - G.tile_start[i] + NXCD - 1) / NXCD;
    if (wide) return launch3_group<2, 2, 4, 2>(s, dim3(per_xcd * NXCD), G, tA, tB, edge);
    if (big) return launch3_group<2, 2, 2, 2>(s, dim3(per_xcd * NXCD), G, tA, tB, edge);
    return launch3_group<1, 1, 2, 2>(s, dim3(per_xcd * NXCD), G, tA, tB, edge);
}

}  // namespace stattn
