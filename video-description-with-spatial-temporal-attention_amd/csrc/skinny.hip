// Register-streaming "skinny" fp32 MFMA GEMM for gfx950, and the fused LSTM cell built on it.
//
// The per-timestep dense work of the decoder has only M = batch rows (64 at the single-GPU
// config, 1..5 when sampling) against 1024-wide weight matrices:
//   state projections h.[Wdl|Wdg|Wdm|Wdlt] and h.U     model_attention.py:371, 389, 402, 415, 437
//   ctx.Wc + gates                                      :439-457
//   readout MLP and vocabulary projection (sampling)    :821-838
// Such a GEMM is bound by streaming the weights once (HBM / Infinity Cache), so B goes straight
// from global memory to VGPRs with 16-byte loads -- no LDS staging -- and A (tiny, L2-resident)
// likewise.  v_mfma_f32_16x16x4_f32 contracts k over the four 16-lane groups g = lane>>4:
// a lane loads FOUR consecutive k of its A row (one dwordx4) and, for each of those k, FOUR
// consecutive n of the matching B row (one dwordx4 per k); the 16 MFMAs of a 16-k step then
// cover a 16 x 64 output tile with columns interleaved 4-per-lane, which makes both the B
// loads and the result stores fully coalesced.
//
// A block is 16 waves = (mt m-tiles of 16 rows) x (16/mt K-slices); K-slices are reduced through
// LDS (deterministic order, no atomics) and the epilogue is applied once.  The grid is
// (#column tiles over all segments) x (row groups): several independent projections of one
// launch fill the 256 CUs together.
#include "kernels.h"
#include "devmath.h"

#include <cstdlib>

namespace stattn {

namespace {

constexpr int NW = 16;          // waves per block
constexpr int TILE_N = 64;

// One wave: acc[nq][r] += sum over its K-slices of A[arow, k] * B[k, bcol + nq]
// steps (16 k each) are dealt round-robin to the `nks` K-slice waves.  The operands of the NEXT step are
// requested before the 16 MFMAs of the current one (two register sets): with only ~8 steps per wave the
// kernel is otherwise a chain of exposed HBM/L2 latencies (measured 18.8 us vs 9 us for h.[Wd*|U]).
struct SkOperands { float4 a; float4 b[4]; };

__device__ __forceinline__ void sk_load(SkOperands& o, const float* __restrict__ Ap, const float* __restrict__ Bp, int ldb, int s) {
    const int k0 = s << 4;
    o.a = ld4(Ap + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) o.b[q] = ld4(Bp + (size_t)(k0 + q) * ldb);
}
__device__ __forceinline__ void sk_mfma(f32x4 (&acc)[4], const SkOperands& o) {
    const float av[4] = {o.a.x, o.a.y, o.a.z, o.a.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], o.b[q].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], o.b[q].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], o.b[q].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], o.b[q].w, acc[3], 0, 0, 0);
    }
}
__device__ __forceinline__ void sk_accumulate(f32x4 (&acc)[4], const SkPair& p, int arow, int bcol,
                                              int ks, int nks, int g) {   // ks / nks: global K-slice index / count
    const int nsteps = p.K >> 4;
    const float* __restrict__ Ap = p.A + (size_t)arow * p.lda + 4 * g;
    const int ldb = p.tile_stride ? 64 : p.ldb;
    const float* __restrict__ Bp = p.tile_stride ? p.B + (size_t)(bcol >> 6) * p.tile_stride + (size_t)(4 * g) * 64 + (bcol & 63)
                                                 : p.B + (size_t)(4 * g) * p.ldb + bcol;
    int s = ks;
    if (s >= nsteps) return;
    SkOperands o0, o1;
    sk_load(o0, Ap, Bp, ldb, s);
    // The prefetch is UNCONDITIONAL (at the tail it re-requests the current step, an L1/L2 hit): a branch
    // around the loads makes hipcc count vmcnt for the no-load path and the wait then drains the prefetch too.
    while (true) {
        int sn = s + nks;
        sk_load(o1, Ap, Bp, ldb, sn < nsteps ? sn : s);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE the MFMAs (hipcc sinks it to its first use)
        sk_mfma(acc, o0);
        __builtin_amdgcn_sched_barrier(0);
        if (sn >= nsteps) break;
        s = sn; sn = s + nks;
        sk_load(o0, Ap, Bp, ldb, sn < nsteps ? sn : s);
        __builtin_amdgcn_sched_barrier(0);
        sk_mfma(acc, o1);
        __builtin_amdgcn_sched_barrier(0);
        if (sn >= nsteps) break;
        s = sn;
    }
}

// 16x16 C/D map: col j = lane & 15, row = 4 * (lane >> 4) + r.  Tile column = 4 j + nq.
__device__ __forceinline__ void sk_spill(float* red, int w, const f32x4 (&acc)[4], int j, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        st4(red + (size_t)w * (16 * TILE_N) + (4 * g + r) * TILE_N + 4 * j,
            make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
}

__global__ __launch_bounds__(1024) void skinny_kernel(const SkArgs a, const int mtb) {
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * TILE_N];
    // locate the segment of this column tile
    int tile = blockIdx.x, si = 0;
    while (si + 1 < a.nseg && tile >= a.seg[si].N / TILE_N) { tile -= a.seg[si].N / TILE_N; ++si; }
    const SkSeg& sg = a.seg[si];
    const int n0 = tile * TILE_N;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int mt = w % mtb, ks = w / mtb, nks = NW / mtb;
    const int rowbase = blockIdx.y * 16 * mtb;
    int arow = rowbase + mt * 16 + j;
    arow = arow < a.M ? arow : a.M - 1;

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kz = a.kz > 1 ? a.kz : 1;
    for (int p = 0; p < sg.npairs; ++p)
        sk_accumulate(acc, sg.p[p], arow, n0 + 4 * j, (int)blockIdx.z * nks + ks, nks * kz, g);
    sk_spill(red, w, acc, j, g);
    __syncthreads();

    for (int idx = tid; idx < 1024 * mtb; idx += 1024) {
        const int mo = idx >> 10, rem = idx & 1023, row = rem >> 6, col = rem & 63;
        const int grow = rowbase + mo * 16 + row;
        if (grow >= a.M) continue;
        float v = 0.f;
        for (int k = 0; k < nks; ++k) v += red[(size_t)(k * mtb + mo) * (16 * TILE_N) + row * TILE_N + col];
        const int n = n0 + col;
        if (kz > 1) { sg.C[(size_t)blockIdx.z * a.part_stride + (size_t)grow * sg.ldc + n] = v; continue; }
        if (sg.bias) v += sg.bias[n];
        if (sg.bias2) v += sg.bias2[n];
        if (sg.add) v += sg.add[(size_t)grow * sg.ldadd + n];
        if (sg.act == 1) v = fast_tanh(v);
        v *= sg.scale;
        if (sg.mul) v *= sg.mul[(size_t)grow * sg.ldmul + n];
        sg.C[(size_t)grow * sg.ldc + n] = v;
    }
}

// Variant for mtb >= 2: the mtb waves of a K-slice group need the SAME 16 x 64 B sub-tile in every step.  Loading it
// once per group (each wave fetches 16/mtb rows) and sharing it through LDS halves / quarters the bytes a CU
// pulls through its vector-memory path -- which is what bounds this kernel (~10 B/clk/CU: tools/skinny_probe.py,
// DESIGN.md section 5) -- at the price of one workgroup barrier per step.  Two LDS stages; the staging area
// aliases the reduction buffer.
__global__ __launch_bounds__(1024) void skinny_shared_kernel(const SkArgs a, const int mtb) {
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * TILE_N];
    int tile = blockIdx.x, si = 0;
    while (si + 1 < a.nseg && tile >= a.seg[si].N / TILE_N) { tile -= a.seg[si].N / TILE_N; ++si; }
    const SkSeg& sg = a.seg[si];
    const int n0 = tile * TILE_N;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int mt = w % mtb, ks = w / mtb, nks = NW / mtb;
    const int rowbase = blockIdx.y * 16 * mtb;
    int arow = rowbase + mt * 16 + j;
    arow = arow < a.M ? arow : a.M - 1;
    const int kz = a.kz > 1 ? a.kz : 1;
    const int slice = (int)blockIdx.z * nks + ks, nslices = nks * kz;
    const int rpw = 16 / mtb;                       // B rows each wave of the group fetches per step (8 or 4)
    float* sB = red + (size_t)ks * (2 * 16 * TILE_N);   // [2 stages][16 k][64 cols] of this K-slice group

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    int stage = 0;
    for (int p = 0; p < sg.npairs; ++p) {
        const SkPair& pr = sg.p[p];
        const int nsteps = pr.K >> 4;
        const int iters = (nsteps + nslices - 1) / nslices;      // uniform over the workgroup (barriers inside)
        const float* __restrict__ Ap = pr.A + (size_t)arow * pr.lda + 4 * g;
        // this lane's part of the shared B sub-tile: rows mt*rpw + {g, g+4 (mtb = 2)}, columns 4j..4j+3
        const float* __restrict__ Bp = pr.B + (size_t)(mt * rpw + g) * pr.ldb + n0 + 4 * j;
        float4 ra, rb0, rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto gload = [&](int it) {
            const int s = slice + it * nslices;
            const bool ok = s < nsteps;
            const int k0 = (ok ? s : 0) << 4;
            ra = ld4(Ap + k0);
            rb0 = ld4(Bp + (size_t)k0 * pr.ldb);
            if (rpw == 8) rb1 = ld4(Bp + (size_t)(k0 + 4) * pr.ldb);
            if (!ok) { ra = make_float4(0.f, 0.f, 0.f, 0.f); }   // a zero A row block contributes nothing
        };
        gload(0);
        for (int it = 0; it < iters; ++it) {
            float* st = sB + stage * (16 * TILE_N);
            st4(st + (mt * rpw + g) * TILE_N + 4 * j, rb0);
            if (rpw == 8) st4(st + (mt * rpw + g + 4) * TILE_N + 4 * j, rb1);
            const float4 acur = ra;
            __syncthreads();
            if (it + 1 < iters) gload(it + 1);                   // next step's operands fly during the MFMAs
            SkOperands o;
            o.a = acur;
#pragma unroll
            for (int q = 0; q < 4; ++q) o.b[q] = ld4(st + (4 * g + q) * TILE_N + 4 * j);
            __builtin_amdgcn_sched_barrier(0);
            sk_mfma(acc, o);
            __builtin_amdgcn_sched_barrier(0);
            stage ^= 1;
        }
    }
    __syncthreads();                                             // staging area is reused by the reduction
    sk_spill(red, w, acc, j, g);
    __syncthreads();
    for (int idx = tid; idx < 1024 * mtb; idx += 1024) {
        const int mo = idx >> 10, rem = idx & 1023, row = rem >> 6, col = rem & 63;
        const int grow = rowbase + mo * 16 + row;
        if (grow >= a.M) continue;
        float v = 0.f;
        for (int k = 0; k < nks; ++k) v += red[(size_t)(k * mtb + mo) * (16 * TILE_N) + row * TILE_N + col];
        const int n = n0 + col;
        if (kz > 1) { sg.C[(size_t)blockIdx.z * a.part_stride + (size_t)grow * sg.ldc + n] = v; continue; }
        if (sg.bias) v += sg.bias[n];
        if (sg.bias2) v += sg.bias2[n];
        if (sg.add) v += sg.add[(size_t)grow * sg.ldadd + n];
        if (sg.act == 1) v = fast_tanh(v);
        v *= sg.scale;
        if (sg.mul) v *= sg.mul[(size_t)grow * sg.ldmul + n];
        sg.C[(size_t)grow * sg.ldc + n] = v;
    }
}

// Fused LSTM cell.  Column tile dt covers units d0 = 16 dt .. +15 of all four gates:
// lane j reads gate (j >> 2), units d0 + 4 (j & 3) .. +3; tile column c = 4 j + nq = gate*16 + dd.
__global__ __launch_bounds__(1024) void lstm_kernel(const LstmArgs a, const int mtb) {
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * TILE_N];
    const int D = a.D;
    const int d0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int mt = w % mtb, ks = w / mtb, nks = NW / mtb;
    const int rowbase = blockIdx.y * 16 * mtb;
    int arow = rowbase + mt * 16 + j;
    arow = arow < a.M ? arow : a.M - 1;
    const int bcol = (j >> 2) * D + d0 + 4 * (j & 3);

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < a.npairs; ++p) sk_accumulate(acc, a.p[p], arow, bcol, ks, nks, g);
    sk_spill(red, w, acc, j, g);
    __syncthreads();

    if (tid < 256 * mtb) {
        const int mo = tid >> 8, rem = tid & 255, row = rem >> 4, dd = rem & 15;
        const int grow = rowbase + mo * 16 + row;
        if (grow < a.M) {
            const int d = d0 + dd;
            float pre[4];
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) {
                float v = 0.f;
                for (int k = 0; k < nks; ++k)
                    v += red[(size_t)(k * mtb + mo) * (16 * TILE_N) + row * TILE_N + gate * 16 + dd];
                if (a.pre_add) v += a.pre_add[(size_t)grow * a.ldpre + gate * D + d];
                if (a.bias) v += a.bias[gate * D + d];
                pre[gate] = v;
            }
            // dropout multiplies the i/f/o PRE-activations (model_attention.py:444-447); g gets none
            const float* dp = a.dp + (size_t)grow * a.lddp;
            const float gi = fast_sigmoid(pre[0] * dp[d]);
            const float gf = fast_sigmoid(pre[1] * dp[D + d]);
            const float go = fast_sigmoid(pre[2] * dp[2 * D + d]);
            const float gg = fast_tanh(pre[3]);
            const float cp = a.c_prev[(size_t)grow * D + d];
            const float hp = a.h_prev[(size_t)grow * D + d];
            const float m = a.mask ? a.mask[grow] : 1.f;
            float c = gf * cp + gi * gg;                 // :453
            c = m * c + (1.f - m) * cp;                  // :454
            float h = go * fast_tanh(c);                 // :456 (uses the masked c)
            h = m * h + (1.f - m) * hp;                  // :457
            a.c_out[(size_t)grow * D + d] = c;
            a.h_out[(size_t)grow * D + d] = h;
            if (a.gates) {
                float* gt = a.gates + (size_t)grow * 4 * D + d;
                gt[0] = gi; gt[D] = gf; gt[2 * D] = go; gt[3 * D] = gg;
            }
            if (a.hd_out) {
                const float d1 = a.d1 ? a.d1[(size_t)grow * a.ldd1 + d] : a.d1_scalar;
                a.hd_out[(size_t)grow * D + d] = h * d1;
            }
        }
    }
}

// m-tiles per block: as many as keep the grid at >= 256 blocks (one per CU), at most ceil(M/16)
int pick_mtb(int ntiles, int M) {
    const int mtmax = (M + 15) / 16;
    for (int mtb = 4; mtb > 1; mtb >>= 1) {
        if (mtb > mtmax && mtb / 2 >= mtmax) continue;
        const int blocks = ntiles * ((M + 16 * mtb - 1) / (16 * mtb));
        if (blocks >= 256) return mtb;
    }
    return 1;
}

}  // namespace

void skinny_seg_defaults(SkSeg& s) {
    s = SkSeg{};
    s.scale = 1.f;
}

hipError_t launch_skinny(hipStream_t s, const SkArgs& a) {
    if (a.M <= 0 || a.nseg <= 0) return hipSuccess;
    int ntiles = 0;
    for (int i = 0; i < a.nseg; ++i) {
        const SkSeg& sg = a.seg[i];
        if (sg.N % TILE_N != 0 || sg.npairs < 1 || sg.npairs > 3) return hipErrorInvalidValue;
        for (int p = 0; p < sg.npairs; ++p)
            if (sg.p[p].K % 16 != 0 || sg.p[p].lda % 4 != 0 || sg.p[p].ldb % 4 != 0) return hipErrorInvalidValue;
        ntiles += sg.N / TILE_N;
    }
    const int kz = a.kz > 1 ? a.kz : 1;
    const int mtb = pick_mtb(ntiles * kz, a.M);
    dim3 grid(ntiles, (a.M + 16 * mtb - 1) / (16 * mtb), kz), block(1024);
    bool packed = false;
    for (int i = 0; i < a.nseg; ++i)
        for (int p = 0; p < a.seg[i].npairs; ++p) packed = packed || a.seg[i].p[p].tile_stride != 0;
    static const char* noshare = sw_tool("STATTN_SKINNY_NOSHARE");
    if (mtb >= 2 && !packed && !noshare) hipLaunchKernelGGL(skinny_shared_kernel, grid, block, 0, s, a, mtb);
    else hipLaunchKernelGGL(skinny_kernel, grid, block, 0, s, a, mtb);
    return hipGetLastError();
}

hipError_t launch_lstm(hipStream_t s, const LstmArgs& a) {
    if (a.M <= 0) return hipSuccess;
    if (a.D % 16 != 0 || a.npairs < 1 || a.npairs > 3) return hipErrorInvalidValue;
    for (int p = 0; p < a.npairs; ++p)
        if (a.p[p].K % 16 != 0) return hipErrorInvalidValue;
    const int ntiles = a.D / 16;
    const int mtb = pick_mtb(ntiles, a.M);
    dim3 grid(ntiles, (a.M + 16 * mtb - 1) / (16 * mtb)), block(1024);
    hipLaunchKernelGGL(lstm_kernel, grid, block, 0, s, a, mtb);
    return hipGetLastError();
}

}  // namespace stattn
