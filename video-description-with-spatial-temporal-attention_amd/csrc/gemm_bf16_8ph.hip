// bf16-MFMA GEMM, 256 x 256 x 64 tile, eight-phase ping-pong schedule (gfx950 only) -- the large projections of a
// bf16 handle (BASELINE.json configs[3]; model_attention.py:664-667 ff_local / ff_motion, :322-326 the attention
// pre-projections) and, round 5, every other GEMM of such a handle that offers whole 256-column tiles.
//
//   C[M,N] = epi(A[M,K] . B[N,K]^T)      A and B bf16, k-contiguous; fp32 accumulation (v_mfma_f32_32x32x16_bf16)
//
// Structure (cdna_hip_programming.md section 5, "256^2 8-phase template", rebuilt here from its description):
//   * 8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 8 accumulators of 32 x 32 (128 registers).
//   * LDS = 2 K-tile buffers x 4 parts (A rows 0-127, A rows 128-255, B columns 0-127, B columns 128-255) x 16 KiB.
//     A part is filled by ONE pass of the workgroup: 512 lanes x 2 x `buffer_load_dwordx4 ... lds` (LDS-DMA: no
//     staging registers, no ds_write).  The DMA writes lane-linear, so row r keeps its 16-byte chunk c at position
//     c ^ ((r >> 1) & 7): the permutation is applied to the per-lane SOURCE offset and to the fragment read, and the
//     16 lanes of a ds_read_b128 group land on 16 distinct 4-bank slots.
//   * A K-tile is four phases, one 64 x 32 quadrant of the wave tile each (8 MFMAs): quadrant order (A0,B0) (A0,B1)
//     (A1,B1) (A1,B0), so a phase reads at most one new A sub-tile (8 x ds_read_b128) and one new B sub-tile (4 x);
//     the fourth phase reads nothing (B0 is kept).  Every phase also issues one part of a later K-tile.
//   * Phase = [fragment reads + DMA issue] s_barrier [8 MFMAs] s_barrier.  Waves 4-7 (the second wave of every SIMD)
//     run ONE BARRIER BEHIND waves 0-3: while one wave of a SIMD issues its MFMAs its partner reads / issues, and no
//     barrier ever finds the matrix pipe of a SIMD without a wave that is inside its MFMA block.
//   * The DMA is counted by hand: loads are issued from inline asm (invisible to hipcc's wait-count pass, which would
//     otherwise drain vmcnt to 0 at every LDS read), one `s_waitcnt vmcnt(4)` per K-tile in the fourth phase leaves the
//     two parts of the K-tile after next in flight across the barriers.
//   Hazards: a part is restaged two phases after its last fragment read -- one phase after
//   where the reading phase ends its load half with lgkmcnt(0) (phase 2) --, and read one phase after the wait that
//   retires it; both orders hold for the waves that run a barrier behind.
#include "kernels.h"
#include "devmath.h"
#include "gemm_bf16_epi.h"

#include <cstdlib>
#include <type_traits>

namespace stattn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BK8 = 64;                        // k per tile (bf16 elements): 128-byte rows
constexpr unsigned PART = 128u * BK8 * 2u;     // 16 KiB: 128 rows
constexpr unsigned KBUF = 4u * PART;           // 64 KiB: A_lo, A_hi, B_lo, B_hi of one K-tile
constexpr int NXCD8 = 8;

__device__ __forceinline__ uint16_t f2bf8(float f) {     // round to nearest even (inputs are finite)
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 is written on purpose: it carries the LDS destination of the DMA
// 16 bytes per lane from (resource base + voff + soff) to LDS byte (lds + 16 * lane); lds is wave-uniform
__device__ __forceinline__ void dma16(unsigned voff, i32x4 rs, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ i32x4 make_rsrc(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));      // stride 0: raw buffer
    rs.z = -1;                                                               // no range check (offsets are clamped by hand)
    rs.w = 0x00020000;
    return rs;
}

// VAR bits (tools/gemm_8ph_probe.py; the product runs VAR = 0): 1 = no s_setprio, 2 = no stagger between the wave halves,
// 4 = no epilogue (one word per lane keeps the accumulators live), 8 = no DMA inside the loop, 16 = no fragment reads inside the loop
// GROUP: up to eight independent problems in ONE launch (GemmBfGroup): block b runs on XCD b % 8, which walks its share of problem 0,
// then of problem 1, ... -- the host lists the longest-K problems first, the hardware hands a free CU the next block, and the short
// problems fill the last round of the long one (configs[3]: ff_local alone is 2.5 rounds of 256 tiles).
template <int VAR, bool MEDGE, bool GROUP>
__global__ __launch_bounds__(512, 2) void gemm_bf16_8ph_kernel(const typename std::conditional<GROUP, GemmBfGroup, GemmBfArgs>::type G) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * KBUF];      // 128 KiB
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    int lin;
    GemmBfArgs g;                                        // (a copy: fields of a dynamically indexed kernel argument would be re-loaded at every use)
    if constexpr (GROUP) {
        const int bid = blockIdx.x, xcd = bid % NXCD8;
        int j = bid / NXCD8, p = 0;
        lin = -1;
        for (; p < G.n; ++p) {
            const int tiles = G.tile_start[p + 1] - G.tile_start[p], q8 = tiles / NXCD8, r8 = tiles % NXCD8;
            const int mine = q8 + (xcd < r8 ? 1 : 0);
            if (j < mine) { lin = xcd * q8 + (xcd < r8 ? xcd : r8) + j; break; }
            j -= mine;
        }
        if (lin < 0) return;                             // padding block of this XCD
        g = G.g[p];
    } else {
        g = G;
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid % NXCD8, q8 = nblk / NXCD8, r8 = nblk % NXCD8;
        lin = g.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / NXCD8 : bid;
    }
    const int tiles_n = g.N / 256;
    const int m0 = (lin / tiles_n) * 256;
    const int n0 = (lin % tiles_n) * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                     // 2 x 4 waves, 128 x 64 each
    const int l31 = lane & 31, kh = lane >> 5;

    // ---- DMA source offsets (bytes from the tile's first row), loop invariant: lane -> (row tid >> 3 (+ 64), chunk) ----
    const int srow = tid >> 3;
    const unsigned schunk = (unsigned)(((tid & 7) ^ ((tid >> 4) & 7)) * 16);
    unsigned voa[2][2], vob[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ra = 128 * p + 64 * i + srow;
            if (MEDGE) { const int last = g.M - 1 - m0; ra = ra < last ? ra : last; }      // rows past M: re-read the last one, never stored
            voa[p][i] = (unsigned)ra * (unsigned)g.lda * 2u + schunk;
            vob[p][i] = (unsigned)(128 * p + 64 * i + srow) * (unsigned)g.ldb * 2u + schunk;
        }
    const i32x4 rsA = make_rsrc(g.A + (size_t)m0 * g.lda);
    const i32x4 rsB = make_rsrc(g.B + (size_t)n0 * g.ldb);
    const unsigned ldsw = lds0 + (unsigned)wave * 1024u;
    // deterministic split-K (kslices > 1: weight-gradient shapes, small M x N and a huge K): blockIdx.y owns a run of K-tiles and
    // leaves an fp32 partial tile in g.ws; splitk_reduce_bf_kernel adds the slices in order and applies the epilogue
    const int nk_all = g.K / BK8, kslices = g.kslices > 1 ? g.kslices : 1;
    const int kper = (nk_all + kslices - 1) / kslices;
    const int kbeg = (int)blockIdx.y * kper;
    const int nk = min(kper, nk_all - kbeg);
    const unsigned kbase = (unsigned)kbeg * 128u;

    // part 0 / 1 = A rows 0-127 / 128-255, part 2 / 3 = B columns 0-127 / 128-255 of K-tile kt, into buffer s
#define DMA_PART(s, part, kt)                                                                                 \
    {                                                                                                         \
        const unsigned so_ = kbase + (unsigned)(kt) * 128u;                                                           \
        const unsigned ld_ = ldsw + (unsigned)(s) * KBUF + (unsigned)(part) * PART;                           \
        if (!(VAR & 8) || (kt) < 2) {                                                                         \
        if ((part) < 2) { dma16(voa[(part) & 1][0], rsA, so_, ld_); dma16(voa[(part) & 1][1], rsA, so_, ld_ + 8192u); } \
        else            { dma16(vob[(part) & 1][0], rsB, so_, ld_); dma16(vob[(part) & 1][1], rsB, so_, ld_ + 8192u); } \
        }                                                                                                     \
    }

    // ---- fragment read offsets: row l31 of a 32-row block, 16-byte chunk (2 kk + kh) ^ ((row >> 1) & 7) ----
    unsigned offk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offk[kk] = (unsigned)l31 * 128u + (unsigned)(((2 * kk + kh) ^ ((l31 >> 1) & 7)) * 16);
    const unsigned abase = (unsigned)wr * PART;                                              // this wave's A part
    const unsigned bbase = 2u * PART + (unsigned)(wc >> 1) * PART + (unsigned)(wc & 1) * (64u * 128u);

    bf16x8 fa[2][4], fb0[4], fb1[4];
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define LD_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
#define LOAD_A(s, a)                                                                                          \
    if (!(VAR & 16) || kt < 2)                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                          \
        fa[mi][kk] = LD_FRAG((unsigned)(s) * KBUF + abase + (unsigned)((a) * 64 + mi * 32) * 128u + offk[kk]);
#define LOAD_B(s, b, dst)                                                                                     \
    if (!(VAR & 16) || kt < 2)                                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                          \
        dst[kk] = LD_FRAG((unsigned)(s) * KBUF + bbase + (unsigned)((b) * 32) * 128u + offk[kk]);
#define MFMA_Q(a, bsrc, nb)                                                                                   \
    {                                                                                                         \
        if (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                      \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                      \
            acc[(a) * 2 + mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][kk], bsrc[kk], acc[(a) * 2 + mi][nb], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);                                                        \
    }
#define BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    // one K-tile: kt from buffer s; parts of K-tile kt + 1 go to buffer s ^ 1, parts of kt + 2 to buffer s
#define KTILE(s, kt)                                                                                          \
    {                                                                                                         \
        const bool has1 = (kt) + 1 < nk, has2 = (kt) + 2 < nk;                                                \
        /* phase 1: A0, B0 */                                                                                 \
        LOAD_B(s, 0, fb0)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        LOAD_A(s, 0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (has1) DMA_PART((s) ^ 1, 0, (kt) + 1)                                                              \
        BAR()                                                                                                 \
        MFMA_Q(0, fb0, 0)                                                                                     \
        BAR()                                                                                                 \
        /* phase 2: B1 (its reads are retired before the barrier: phase 3 restages the B parts) */            \
        LOAD_B(s, 1, fb1)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (has1) DMA_PART((s) ^ 1, 1, (kt) + 1)                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        BAR()                                                                                                 \
        MFMA_Q(0, fb1, 1)                                                                                     \
        BAR()                                                                                                 \
        /* phase 3: A1 */                                                                                     \
        LOAD_A(s, 1)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (has2) DMA_PART(s, 2, (kt) + 2)                                                                    \
        BAR()                                                                                                 \
        MFMA_Q(1, fb1, 1)                                                                                     \
        BAR()                                                                                                 \
        /* phase 4: no reads; K-tile kt + 1 has landed once only the two B parts of kt + 2 are in flight */   \
        if (has2) { DMA_PART(s, 3, (kt) + 2) if (!(VAR & 8)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        BAR()                                                                                                 \
        MFMA_Q(1, fb0, 0)                                                                                     \
        BAR()                                                                                                 \
    }

    // ---- prologue: K-tile 0 complete, the B parts of K-tile 1 in flight ----
    DMA_PART(0, 2, 0) DMA_PART(0, 3, 0) DMA_PART(0, 0, 0) DMA_PART(0, 1, 0)
    if (nk > 1) { DMA_PART(1, 2, 1) DMA_PART(1, 3, 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR()
    if (!(VAR & 2) && wr == 1) BAR()                 // the second wave of every SIMD runs one barrier behind

    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        KTILE(0, kt)
        KTILE(1, kt + 1)
    }
    if (kt < nk) KTILE(0, kt)
    if (!(VAR & 2) && wr == 0) BAR()

#undef KTILE
#undef BAR
#undef MFMA_Q
#undef LOAD_B
#undef LOAD_A
#undef LD_FRAG
#undef DMA_PART

    // ---- epilogue (gemm_bf16_epi.h): through LDS, eight consecutive columns of a row per lane ----
    if (VAR & 4) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) v += acc[i][j][r];
        if (g.Cb) g.Cb[(size_t)(m0 + wr * 128 + l31) * g.ldcb + n0 + wc * 64 + kh] = f2bf8(v);
        return;
    }
    __syncthreads();                                  // every wave is done reading the K-tile buffers
    float* stage = reinterpret_cast<float*>(smem + wave * bf16_epi::STAGE_BYTES);
    if (kslices > 1) {
        GemmBfArgs gp{};                              // the raw partial tile of this K-slice
        gp.M = g.M; gp.N = g.N; gp.C = g.ws + (size_t)blockIdx.y * g.M * g.N; gp.ldc = g.N; gp.rowgroup = 1;
        bf16_epi::store_tile<4, 2, MEDGE>(gp, acc, stage, m0 + wr * 128, n0 + wc * 64, lane);
        return;
    }
    bf16_epi::store_tile<4, 2, MEDGE>(g, acc, stage, m0 + wr * 128, n0 + wc * 64, lane);
}

// C = epilogue(sum over the K-slices of the partial tiles), four consecutive columns per thread, slices added in order
__global__ __launch_bounds__(256) void splitk_reduce_bf_kernel(const GemmBfArgs g) {
    const int n4 = g.N >> 2;
    const size_t total = (size_t)g.M * n4, MN = (size_t)g.M * g.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / n4);
        int col = (int)(i % n4) * 4;
        float4 v = ld4(g.ws + (size_t)row * g.N + col);
        for (int sl = 1; sl < g.kslices; ++sl) {
            const float4 p = ld4(g.ws + sl * MN + (size_t)row * g.N + col);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const bf16_epi::Out o = bf16_epi::select_out(g, col);
        if (o.bias) { const float4 b = ld4(o.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (g.bias_b) { const float4 b = ld4(g.bias_b + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (g.add) { const float4 b = ld4(g.add + (size_t)row * g.ldadd + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (g.rowadd) { const float4 b = ld4(g.rowadd + (size_t)(row / g.rowgroup) * g.ldrow + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (g.act == 1) { v.x = fast_tanh(v.x); v.y = fast_tanh(v.y); v.z = fast_tanh(v.z); v.w = fast_tanh(v.w); }
        if (g.mul) { const float4 b = ld4(g.mul + (size_t)row * g.ldmul + col); v.x *= b.x; v.y *= b.y; v.z *= b.z; v.w *= b.w; }
        if (o.C) st4(o.C + (size_t)row * g.ldc + col, v);
        if (o.Cb) *reinterpret_cast<uint2*>(o.Cb + (size_t)row * g.ldcb + col) = make_uint2(bf16_epi::pack_bf16(v.x, v.y), bf16_epi::pack_bf16(v.z, v.w));
    }
}

template <int VAR>
hipError_t launch_var(hipStream_t s, const GemmBfArgs& g) {
    const int tiles = ((g.M + 255) / 256) * (g.N / 256);
    const dim3 grid(tiles, g.kslices > 1 ? g.kslices : 1);
    if (g.M % 256) hipLaunchKernelGGL((gemm_bf16_8ph_kernel<VAR, true, false>), grid, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16_8ph_kernel<VAR, false, false>), grid, dim3(512), 0, s, g);
    if (g.kslices > 1) {
        const size_t total = (size_t)g.M * (g.N / 4);
        hipLaunchKernelGGL(splitk_reduce_bf_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

}  // namespace

bool gemm_bf16_8ph_supported(const GemmBfArgs& g) {
    return g.N % 256 == 0 && g.n_split % 256 == 0 && g.K % BK8 == 0 && g.K >= BK8 && g.lda % 8 == 0 && g.ldb % 8 == 0 &&
           (size_t)256 * g.lda * 2 < 0xffffffffull && (size_t)256 * g.ldb * 2 < 0xffffffffull && bf16_epi::wide_ok(g);
}

// K-slices for a problem of `tiles` 256 x 256 tiles and nk K-tiles: fill the 256 CUs, at least 8 K-tiles per slice
int gemm_bf16_8ph_slices(const GemmBfArgs& g) {
    const long tiles = (long)((g.M + 255) / 256) * (g.N / 256);
    const int nk = g.K / BK8;
    if (tiles >= 192 || nk < 16) return 1;
    int sl = (int)((256 + tiles - 1) / tiles);
    if (sl > nk / 8) sl = nk / 8;
    return sl < 1 ? 1 : sl;
}

hipError_t launch_gemm_bf16_8ph(hipStream_t s, const GemmBfArgs& gin) {
    GemmBfArgs g = gin;
    if (!gemm_bf16_8ph_supported(g)) return hipErrorInvalidValue;
    if (g.kslices > 1) {
        const int nk = g.K / BK8, kper = (nk + g.kslices - 1) / g.kslices;
        g.kslices = (nk + kper - 1) / kper;           // no empty slice
        if (!g.ws || g.ws_floats < (size_t)g.kslices * g.M * g.N || ((size_t)g.ws & 15) || g.N % 4) return hipErrorInvalidValue;
    }
    static const char* var = sw_tool("STATTN_8PH_VAR");       // ablations (tools/gemm_8ph_probe.py); the product runs variant 0
    const int v = var ? atoi(var) : 0;
    if (v == 1) return launch_var<1>(s, g);
    if (v == 2) return launch_var<2>(s, g);
    if (v == 4) return launch_var<4>(s, g);
    if (v == 12) return launch_var<12>(s, g);
    if (v == 20) return launch_var<20>(s, g);
    if (v == 28) return launch_var<28>(s, g);
    return launch_var<0>(s, g);
}

// Up to GEMM_BF_GROUP_MAX independent problems (each gemm_bf16_8ph_supported, no split-K) in one launch; list the longest K first.
hipError_t launch_gemm_bf16_8ph_group(hipStream_t s, const GemmBfArgs* gs, int n) {
    if (n < 1 || n > GEMM_BF_GROUP_MAX) return hipErrorInvalidValue;
    if (n == 1) return launch_gemm_bf16_8ph(s, gs[0]);
    GemmBfGroup G{};
    G.n = n;
    int maxper = 0;
    for (int p = 0; p < n; ++p) {
        if (!gemm_bf16_8ph_supported(gs[p]) || gs[p].kslices > 1 || gs[p].M <= 0) return hipErrorInvalidValue;
        if (gs[p].n_split > 0 && (gs[p].add || gs[p].rowadd || gs[p].mul || gs[p].bias_b)) return hipErrorInvalidValue;     // (see launch_gemm_bf16)
        G.g[p] = gs[p];
        if (G.g[p].rowgroup < 1) G.g[p].rowgroup = 1;
        const int tiles = ((gs[p].M + 255) / 256) * (gs[p].N / 256);
        G.tile_start[p + 1] = G.tile_start[p] + tiles;
    }
    // every XCD gets ceil(tiles_p / 8) or floor blocks of problem p: the grid is 8 x the largest per-XCD total
    for (int x = 0; x < NXCD8; ++x) {
        int per = 0;
        for (int p = 0; p < n; ++p) { const int t = G.tile_start[p + 1] - G.tile_start[p]; per += t / NXCD8 + (x < t % NXCD8 ? 1 : 0); }
        maxper = per > maxper ? per : maxper;
    }
    hipLaunchKernelGGL((gemm_bf16_8ph_kernel<0, true, true>), dim3(maxper * NXCD8), dim3(512), 0, s, G);
    return hipGetLastError();
}

}  // namespace stattn
