// Wide row-panel fp32 MFMA kernels: the per-timestep dense work of the decoder for 65 .. 256 rows (beam search over many
// videos: videos x beam rows; training batches beyond 64 rows) -- the same problems, operand layouts, segments and fused
// epilogues as panel.hip (state projections h.[Wdl|Wdg|Wdm|Wdlt|U] model_attention.py:371, 389, 402, 415, 437; the LSTM
// cell :437-457; the sampler's readout and vocabulary projection :821-838), tiled for the shape that panel.hip handles badly.
//
// Why another kernel.  panel.hip gives a workgroup EVERY row and 16 output columns on v_mfma_f32_16x16x4_f32.  At 64 rows
// the activations are a small operand; at 160 rows (configs[4]: 32 videos x beam 5) every one of the 512 workgroups of the
// state-projection launch re-reads the whole 640 KB activation panel from L2 for 64 KB of weights -- 360 MB through the
// L2 -> CU path per launch, one 1 KiB operand load per 128 MFMA cycles: 45 us against an MFMA floor of 17 (0.38).
// Here a workgroup owns a row group of up to 8 x 32 rows and 32 output columns on v_mfma_f32_32x32x2_f32: a 1 KiB operand
// load feeds 256 MFMA cycles, the activation bytes per launch halve, and a wave holds MB accumulator blocks so that one
// weight load feeds MB x 4 MFMAs.  Launches whose column blocks alone do not fill the chip split the ROWS over gridDim.y
// (the LSTM at D = 1024 has 128 column blocks: two row groups) -- the weights then cross the fabric once per row group.
//
// Layouts are panel.hip's, unchanged -- the producers of h / ctx / emb / hd keep writing them:
//   activations  A_pk[row / 16][k / 16][(k / 4) % 4 * 16 + row % 16][k % 4]          (pn_pack_offset)
//   weights      P[tile c of 16 columns][k / 16][lane][q] = W[16 s + 4 (lane >> 4) + q][col(c, lane & 15)]
// v_mfma_f32_32x32x2 contracts k over the two half-waves: lane (kh = lane >> 5, l31 = lane & 31) loads the float4 of k =
// 8 s + 4 kh .. + 3 of its row / column, and the four MFMAs of an 8-k step take component q of both operands.  In the
// packed layouts that float4 sits at  base(row or column) + 64 kh + 128 s  floats: per wave and step two 512-byte runs.
//
// Workgroup = KS waves = KS K-slices (4; 8 / 16 for row groups of one or two blocks: wave w walks the 8-k steps w, w + KS, ...)
// with a register ring of R steps of operands in flight; the partial tiles are reduced through LDS in a fixed order
// (deterministic), then the epilogue runs once.
//
// Measured at configs[4] (tools/panelw_probe.hip, 160 rows, one MI355X whose matrix pipe runs 136-142 TFLOP/s = 65 clocks
// per MFMA at 2.18-2.25 GHz in an MFMA-only loop: tools/mfma_rate_probe.hip): state projections 45.5 -> 30.8 us per launch;
// the main loop runs 25 us against 19 us with the loads compiled out (every workgroup re-reads the 640 KB activation panel
// from L2: 19 bytes per clock and CU), K-slice reduction 2.0 us, epilogue 1.2 us (5.4 us before the segment's fields were
// hoisted out of the loops: each was an s_load + s_waitcnt lgkmcnt(0) per iteration).
#include "panel_inl.h"

#include <cstdlib>

namespace stattn {

#ifdef STATTN_PROBES
__device__ long long* pw_probe = nullptr;      // tools/panelw_probe.hip: 8 stamps per workgroup (100 MHz wall clock)
#define PW_STAMP(i) do { if (pw_probe && threadIdx.x == 0) pw_probe[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PW_STAMP(i) do {} while (0)
#endif

namespace {

#if defined(STATTN_PROBES) && !defined(PW_VARIANT)
#define PW_VARIANT 0
#endif
constexpr int WCB = 32;          // columns per workgroup
constexpr int WPITCH = 40;       // LDS row pitch (floats): the two half-waves of a C store hit disjoint banks

template <int MB>
struct WOps { float4 a[MB]; float4 b; };

template <int MB>
__device__ __forceinline__ void pw_load(WOps<MB>& o, const float* const (&Ap)[MB], int astep, const float* __restrict__ Bp, int s, int rot, int nsteps) {
    s += rot;
    s = s >= nsteps ? s - nsteps : s;
#if !(defined(STATTN_PROBES) && (PW_VARIANT == 2 || PW_VARIANT == 4))       // tools/panelw_probe.hip ablations: 2 / 4 = no weight loads
    o.b = ld4(Bp + (size_t)s * 128);
#endif
#if defined(STATTN_PROBES) && (PW_VARIANT == 1 || PW_VARIANT == 4)          // 1 / 4 = no activation loads
    return;
#endif
#pragma unroll
    for (int i = 0; i < MB; ++i) o.a[i] = ld4(Ap[i] + (size_t)astep * s);
}

template <int MB>
__device__ __forceinline__ void pw_mfma(f32x16 (&acc)[MB], const WOps<MB>& o) {
#if defined(STATTN_PROBES) && PW_VARIANT == 3                                // 3 = no MFMAs
    asm volatile("" :: "v"(o.a[0].x), "v"(o.b.x), "v"(o.a[MB - 1].w), "v"(o.b.w)); return;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MB; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i][q], o.b[q], acc[i], 0, 0, 0);
}

// acc += A[row blocks of this workgroup, 8-k steps s0, s0 + 4, ...] . panel  (the ring discipline of pn_accumulate:
// every load of the first R steps is issued before the first MFMA, refills are unconditional and clamped)
template <int MB, int R, int KS>
__device__ __forceinline__ void pw_accumulate(f32x16 (&acc)[MB], const float* const (&Ap)[MB], int astep, const float* __restrict__ Bp,
                                              int nsteps, int s0, int rot) {
    constexpr int stride = KS;
    if (s0 >= nsteps) return;
    const int n = (nsteps - s0 + stride - 1) / stride;
    const int last = s0 + (n - 1) * stride;
    WOps<MB> ring[R];
#pragma unroll
    for (int u = 0; u < R; ++u) pw_load(ring[u], Ap, astep, Bp, min(s0 + u * stride, last), rot, nsteps);
    int base = 0;
    for (; base + R < n; base += R) {
        const int sb = s0 + base * stride;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            pw_mfma(acc, ring[u]);
            __builtin_amdgcn_sched_barrier(0);
            pw_load(ring[u], Ap, astep, Bp, min(sb + (u + R) * stride, last), rot, nsteps);
        }
    }
    const int rem = n - base;
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (u >= rem) break;
        __builtin_amdgcn_sched_barrier(0);
        pw_mfma(acc, ring[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// operand base pointers of this lane: A row block i of the workgroup's row group, B column block cb of a 16-column-tile panel set
template <int MB>
__device__ __forceinline__ void pw_bases(const PnPair& pr, int M, int rb0, int cb, int lane, const float* (&Ap)[MB], int& astep, const float*& Bp) {
    const int kh = lane >> 5, l31 = lane & 31;
    const int S = pr.K >> 4;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int row = min((rb0 + i) * 32 + l31, M - 1);       // rows past M are clamped: their accumulators are never stored
        Ap[i] = pr.apk ? pr.A + ((size_t)(row >> 4) * S) * 256 + (row & 15) * 4 + 64 * kh
                       : pr.A + (size_t)row * pr.lda + 4 * kh;
    }
    astep = pr.apk ? 128 : 8;
    Bp = pr.P + ((size_t)(2 * cb + (l31 >> 4)) * S) * 256 + (l31 & 15) * 4 + 64 * kh;
}

// C layout of the 32x32 MFMA: acc[r] = C[row (r & 3) + 8 (r >> 2) + 4 kh][col l31]
template <int MB>
__device__ __forceinline__ void pw_store(float* buf, const f32x16 (&acc)[MB], int lane) {
    const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * WPITCH + l31] = acc[i][r];
}
template <int MB>
__device__ __forceinline__ void pw_add(const float* buf, f32x16 (&acc)[MB], int lane) {
    const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += buf[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * WPITCH + l31];
}
// KS K-slice partials -> red[0] + red[1], halving the number of live partials per round in a fixed order (deterministic);
// KS / 2 buffers of MB * 32 rows
template <int MB, int KS>
__device__ __forceinline__ void pw_reduce(float* red, f32x16 (&acc)[MB], int w, int lane) {
    constexpr int BUF = MB * 32 * WPITCH;
#pragma unroll
    for (int n = KS; n > 2; n >>= 1) {
        const int hlf = n >> 1;
        if (w >= hlf && w < n) pw_store<MB>(red + (w - hlf) * BUF, acc, lane);
        __syncthreads();
        if (w < hlf) pw_add<MB>(red + w * BUF, acc, lane);
        __syncthreads();
    }
    if (w < 2) pw_store<MB>(red + w * BUF, acc, lane);
    __syncthreads();
}

// ---- general grouped GEMM with the fused epilogue of panel_kernel --------------------------------------------------
// optional epilogue operands are read through pointers that are always valid (panel.hip: a load inside `if (add)` keeps its
// branch and a full wait -- add and mul were two dependent round trips per output row of a thread)
__device__ __attribute__((aligned(16))) const float pw_zero4[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) const float pw_one4[4] = {1.f, 1.f, 1.f, 1.f};
// -DSTATTN_PN_V2=1 (variant `pnv2`, see panel.hip; unmeasured): `p ? p : pw_zero4` mixes a kernel-argument pointer with the address of a
// __device__ constant, so every load through it is a FLAT load -- counted on lgkmcnt like the LDS reads of the K-slice reduction
// beside it, which then wait for a global round trip.  V2 reads these operands through global (address space 1) pointers.
#ifndef STATTN_PN_V2
#define STATTN_PN_V2 0
#endif
#if STATTN_PN_V2
typedef const __attribute__((address_space(1))) float* pw_gptr;
__device__ __forceinline__ float4 pw_ld4(const float* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = *reinterpret_cast<const __attribute__((address_space(1))) v4f*>((pw_gptr)p);
    return make_float4(t.x, t.y, t.z, t.w);
}
#define PW_G(p) ((pw_gptr)(p))
#else
#define pw_ld4 ld4
#define PW_G(p) (p)
#endif

template <int MB, int R, int KS>
__global__ __launch_bounds__(64 * KS) void panelw_kernel(const PnArgs a) {
    constexpr int NTH = 64 * KS, FPITCH = WCB + 1;
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int RB = MB * 32, BUF = RB * WPITCH;
    int cb = a.plain_order ? (int)blockIdx.x : xcd_contiguous((int)blockIdx.x, (int)gridDim.x), si = 0;
    while (si + 1 < a.nseg && cb >= a.seg[si].N / WCB) { cb -= a.seg[si].N / WCB; ++si; }
    const PnSeg& sg = a.seg[si];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb0 = (int)blockIdx.y * MB;                  // first 32-row block of this row group
    const int row0 = rb0 * 32;

    PW_STAMP(0);
    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int p = 0; p < sg.npairs; ++p) {
        const float* Ap[MB]; const float* Bp; int astep;
        pw_bases<MB>(sg.p[p], a.M, rb0, cb, lane, Ap, astep, Bp);
        const int nsteps = sg.p[p].K >> 3;
        pw_accumulate<MB, R, KS>(acc, Ap, astep, Bp, nsteps, w, pn_rotation((int)blockIdx.x, nsteps));
    }
    PW_STAMP(1);
    pw_reduce<MB, KS>(red, acc, w, lane);
    PW_STAMP(2);

    const int n0 = cb * WCB;
    // Every field of the segment the epilogue needs, read ONCE: `a.seg[si]` is indexed dynamically, and left inside the loops
    // each field is an s_load from the kernel-argument segment per iteration, its s_waitcnt lgkmcnt(0) also draining the LDS
    // reads (measured: 5.4 us of epilogue for 20 stores per thread).  A thread owns four consecutive columns (float4 LDS
    // reads, 16-byte global accesses) of every (NTH / 8)-th row.
    const float* const bias = sg.bias ? sg.bias : pw_zero4; const int sbias = sg.bias ? 1 : 0;
    const float* const bias2 = sg.bias2 ? sg.bias2 : pw_zero4; const int sbias2 = sg.bias2 ? 1 : 0;
    const float* const add = sg.add ? sg.add : pw_zero4; const int sadd = sg.add ? 1 : 0;
    const float* const mul = sg.mul ? sg.mul : pw_one4; const int smul = sg.mul ? 1 : 0;
    float* const Cp = sg.C; float* const Cpk = sg.Cpk; float* const stats = sg.stats;
    const int ldadd = sg.ldadd, ldmul = sg.ldmul, ldc = sg.ldc, act = sg.act, Spk = sg.N >> 4;
    const float scale = sg.scale;
    const int c4 = (tid & 7) * 4, n = n0 + c4;
    // every global operand of this thread's epilogue in ONE burst: both biases and add / mul of all its rows
    constexpr int NR = (RB * 8 + NTH - 1) / NTH;            // rows per thread
    float4 b4 = pw_ld4(bias + n * sbias);
    const float4 b2 = pw_ld4(bias2 + n * sbias2);
    float4 ad[NR], ml[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int row = min(row0 + (tid >> 3) + q * (NTH / 8), a.M - 1);
        ad[q] = pw_ld4(add + ((size_t)row * ldadd + n) * sadd);
        ml[q] = pw_ld4(mul + ((size_t)row * ldmul + n) * smul);
    }
    b4.x += b2.x; b4.y += b2.y; b4.z += b2.z; b4.w += b2.w;
    if (stats) {   // vocabulary statistics of this column block (the logits are never stored): see panel_kernel
        const int stats_V = sg.stats_V, stats_kb = sg.stats_kb, skip0 = sg.stats_skip0, ntile = sg.N / WCB;
        float* fin = red + (KS / 2 > 2 ? KS / 2 : 2) * BUF;  // [RB][33] final biased values
        for (int rr = tid >> 3; rr < RB; rr += NTH / 8) {
            const float4 p0 = ld4(red + rr * WPITCH + c4), p1 = ld4(red + BUF + rr * WPITCH + c4);
            float v[4] = {p0.x + p1.x + b4.x, p0.y + p1.y + b4.y, p0.z + p1.z + b4.z, p0.w + p1.w + b4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) fin[rr * FPITCH + c4 + q] = (n + q >= stats_V || (skip0 && n + q == 0)) ? -INFINITY : v[q];
        }
        __syncthreads();
        // one THREAD per row walks its 32 columns (row pitch 33: conflict-free): max, sum exp, and the kb largest values kept
        // as a sorted list (strict comparisons in ascending column order: ties go to the lower column).  No cross-lane
        // traffic -- a wave-per-row reduction costs ten LDS-latency shuffles per row, 10 us at 160 rows.
        for (int rr = tid; rr < RB; rr += NTH) {
            const int row = row0 + rr;
            if (row >= a.M) continue;
            const float* f = fin + rr * FPITCH;
            float fv[WCB];
#pragma unroll
            for (int c = 0; c < WCB; ++c) fv[c] = f[c];
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < WCB; ++c) mx = fmaxf(mx, fv[c]);
            float se = 0.f;
            float lv[PN_STATS_KB]; int lc[PN_STATS_KB];
#pragma unroll
            for (int i = 0; i < PN_STATS_KB; ++i) { lv[i] = -INFINITY; lc[i] = n0; }
#pragma unroll
            for (int c = 0; c < WCB; ++c) {
                const float v = fv[c];
                se += v > -INFINITY ? __expf(v - mx) : 0.f;
                // branch-free sorted insert: the new value enters at the bottom and bubbles up (strict >: ties keep the lower column ahead)
                float cv = v; int cc = n0 + c;
#pragma unroll
                for (int i = PN_STATS_KB - 1; i >= 0; --i) {
                    const bool sw = cv > lv[i];
                    if (i < PN_STATS_KB - 1) { lv[i + 1] = sw ? lv[i] : cv; lc[i + 1] = sw ? lc[i] : cc; }
                    cv = sw ? cv : lv[i]; cc = sw ? cc : lc[i];
                    if (i == 0) { lv[0] = cv; lc[0] = cc; }
                }
            }
            float* rec = stats + ((size_t)row * ntile + cb) * PN_STATS_REC;
            rec[0] = mx; rec[1] = se;
#pragma unroll
            for (int i = 0; i < PN_STATS_KB; ++i)
                if (i < stats_kb) { rec[2 + i] = lv[i]; reinterpret_cast<int*>(rec)[2 + PN_STATS_KB + i] = lc[i]; }
        }
        PW_STAMP(3);
        return;
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int rr = (tid >> 3) + q * (NTH / 8), row = row0 + rr;
        if (rr >= RB || row >= a.M) break;                 // (rows ascend with q)
        const float4 p0 = ld4(red + rr * WPITCH + c4), p1 = ld4(red + BUF + rr * WPITCH + c4);
        float4 v = make_float4(p0.x + p1.x + b4.x, p0.y + p1.y + b4.y, p0.z + p1.z + b4.z, p0.w + p1.w + b4.w);
        v.x += ad[q].x; v.y += ad[q].y; v.z += ad[q].z; v.w += ad[q].w;
        if (act == 1) v = make_float4(fast_tanh(v.x), fast_tanh(v.y), fast_tanh(v.z), fast_tanh(v.w));
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        v.x *= ml[q].x; v.y *= ml[q].y; v.z *= ml[q].z; v.w *= ml[q].w;
        if (Cp) st4(Cp + (size_t)row * ldc + n, v);
        if (Cpk) st4(Cpk + pn_pack_offset(row, n, Spk), v);
    }
    PW_STAMP(3);
}

// ---- LSTM cell with its GEMM (model_attention.py:437-457).  Column block cb = the two PN_COLS_LSTM tiles 2 cb, 2 cb + 1 =
// units 8 cb .. 8 cb + 7 of all four gates: block column t * 16 + gate * 4 + u holds gate `gate` of unit 8 cb + 4 t + u.
template <int MB, int R>
__global__ __launch_bounds__(256) void lstm_panelw_kernel(const LstmPnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int RB = MB * 32, BUF = RB * WPITCH, NITEM = RB * 8 / 256;      // (row, unit) items per thread = MB
    const int D = a.D;
    const int cb = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb0 = (int)blockIdx.y * MB, row0 = rb0 * 32;

    // the epilogue's operands do not depend on the GEMM: requested before it (panel.hip lstm_panel_kernel)
    // (every load unconditional, through pointers that are always valid, and no arithmetic on the loaded values before the
    // main loop: with `a.bias ? ... : 0` hipcc kept the branches and waited for each load where it stood -- the early requests
    // were four dependent round trips per item IN FRONT of the GEMM.)  The bias depends on the unit only: once per thread.
    struct EpiIn { float add[4], dp[3], cp, hp, m, d1; };
    const float* const padd = a.pre_add ? a.pre_add : pw_zero4; const int sadd = a.pre_add ? 1 : 0;
    const float* const pbias = a.bias ? a.bias : pw_zero4; const int sbias = a.bias ? 1 : 0;
    const float* const pmask = a.mask ? a.mask : pw_one4; const int smask = a.mask ? 1 : 0;
    const float* const pd1 = a.d1 ? a.d1 : pw_one4; const int sd1 = a.d1 ? 1 : 0;
    float gbias[4];
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) gbias[gate] = PW_G(pbias)[(gate * D + 8 * cb + (tid & 7)) * sbias];
    auto epi_load = [&](int idx) {
        EpiIn e;
        const int row = min(row0 + (idx >> 3), a.M - 1), d = 8 * cb + (idx & 7);
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) e.add[gate] = PW_G(padd)[((size_t)row * a.ldpre + gate * D + d) * sadd];
#pragma unroll
        for (int q = 0; q < 3; ++q) e.dp[q] = a.dp[(size_t)row * a.lddp + q * D + d];
        e.cp = a.c_prev[(size_t)row * D + d];
        e.hp = a.h_prev[(size_t)row * D + d];
        e.m = PW_G(pmask)[row * smask];
        e.d1 = PW_G(pd1)[((size_t)row * a.ldd1 + d) * sd1];
        return e;
    };
    PW_STAMP(0);
    constexpr int NPRE = NITEM < 3 ? NITEM : 3;
    EpiIn pre[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) pre[q] = epi_load(tid + 256 * q);

    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int p = 0; p < a.npairs; ++p) {
        const float* Ap[MB]; const float* Bp; int astep;
        pw_bases<MB>(a.p[p], a.M, rb0, cb, lane, Ap, astep, Bp);
        const int nsteps = a.p[p].K >> 3;
        pw_accumulate<MB, R, 4>(acc, Ap, astep, Bp, nsteps, w, pn_rotation(cb, nsteps));
    }
    PW_STAMP(1);
    pw_reduce<MB, 4>(red, acc, w, lane);
    PW_STAMP(2);

#pragma unroll
    for (int q = 0; q < NITEM; ++q) {
        const int idx = tid + 256 * q;
        const int rr = idx >> 3, u8 = idx & 7, row = row0 + rr;
        if (row >= a.M) continue;
        const EpiIn e = q < NPRE ? pre[q < NPRE ? q : 0] : epi_load(idx);
        const int d = 8 * cb + u8;
        float pv[4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const int col = (u8 >> 2) * 16 + gate * 4 + (u8 & 3);
            pv[gate] = red[rr * WPITCH + col] + red[BUF + rr * WPITCH + col] + (e.add[gate] + gbias[gate]);
        }
        // dropout multiplies the i/f/o PRE-activations (:444-447); g gets none
        const float gi = fast_sigmoid(pv[0] * e.dp[0]);
        const float gf = fast_sigmoid(pv[1] * e.dp[1]);
        const float go = fast_sigmoid(pv[2] * e.dp[2]);
        const float gg = fast_tanh(pv[3]);
        float cn = gf * e.cp + gi * gg;                // :453
        cn = e.m * cn + (1.f - e.m) * e.cp;            // :454
        float hn = go * fast_tanh(cn);                 // :456 (uses the masked c)
        hn = e.m * hn + (1.f - e.m) * e.hp;            // :457
        a.c_out[(size_t)row * D + d] = cn;
        a.h_out[(size_t)row * D + d] = hn;
        if (a.h_pk) a.h_pk[pn_pack_offset(row, d, D >> 4)] = hn;
        if (a.gates) {
            float* gt = a.gates + (size_t)row * 4 * D + d;
            gt[0] = gi; gt[D] = gf; gt[2 * D] = go; gt[3 * D] = gg;
        }
        const float d1 = a.d1 ? e.d1 : a.d1_scalar;
        if (a.hd_out) a.hd_out[(size_t)row * D + d] = hn * d1;
        if (a.hd_pk) a.hd_pk[pn_pack_offset(row, d, D >> 4)] = hn * d1;
    }
    PW_STAMP(3);
}

template <class F>
hipError_t pw_allow_lds(F f, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// rows -> (row groups, 32-row blocks per group): the fewest row groups whose grid fills the chip, at most 8 blocks each
void pw_geom(int M, int colblocks, int& RG, int& MB) {
    const int nmb = (M + 31) / 32;
    RG = 1;
    while ((nmb + RG - 1) / RG > 8 || (colblocks * RG < 200 && RG < nmb)) ++RG;
    MB = (nmb + RG - 1) / RG;
    if (MB == 7) MB = 8;                                  // instantiated block counts: 1..6, 8
    RG = (nmb + MB - 1) / MB;
}

}  // namespace

bool panel_wide_enabled() {
    static const char* off = sw_tool("STATTN_NO_PANELW");   // A/B switch for tools
    return !off;
}

// the wide kernel takes a launch when it has more than 64 rows, no K split over blocks, and every segment is a multiple of 32 columns
bool panel_wide_supported(const PnArgs& a) {
    if (!panel_wide_enabled() || a.M < (a.wide_from > 0 ? a.wide_from : 65) || a.M > 256 * 8 || a.kz > 1) return false;
    for (int i = 0; i < a.nseg; ++i) {
        if (a.seg[i].N % WCB != 0 || a.seg[i].ldc % 4 != 0 || (a.seg[i].add && a.seg[i].ldadd % 4 != 0) || (a.seg[i].mul && a.seg[i].ldmul % 4 != 0)) return false;
        for (int p = 0; p < a.seg[i].npairs; ++p)
            if (a.seg[i].p[p].K % 32 != 0 || (!a.seg[i].p[p].apk && a.seg[i].p[p].lda % 4 != 0)) return false;
    }
    return true;
}
bool lstm_panel_wide_supported(const LstmPnArgs& a) {
    // (measured at D = 1024 against panel.hip's LSTM kernel: 80 / 96 / 128 rows 31 vs 20.5 / 32 vs 20.5 / 32 vs 24 us, 160 rows 33 vs 37:
    // this kernel's time barely depends on the row count, the 16-column kernel's grows with it -- it takes over from 144 rows)
    if (!panel_wide_enabled() || a.M < 144 || a.D % 8 != 0) return false;
    for (int p = 0; p < a.npairs; ++p)
        if (a.p[p].K % 32 != 0) return false;
    return true;
}

hipError_t launch_panel_wide(hipStream_t s, const PnArgs& a) {
    int colblocks = 0;
    bool stats = false;
    for (int i = 0; i < a.nseg; ++i) { colblocks += a.seg[i].N / WCB; stats = stats || a.seg[i].stats; }
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].stats && (a.seg[i].stats_kb < 1 || a.seg[i].stats_kb > PN_STATS_KB || a.seg[i].stats_seed)) return hipErrorInvalidValue;
    int RG, MB;
    pw_geom(a.M, colblocks, RG, MB);
    // K-slice waves per workgroup: 4, or 8 / 16 for the thin row groups (one or two row blocks per wave leave the matrix pipe
    // waiting for operands: more waves per SIMD cover the ~2 us of loaded latency that a deeper ring cannot)
    const int KS = MB == 1 ? 16 : (MB == 2 ? 8 : 4);
    size_t lds = ((size_t)(KS / 2 > 2 ? KS / 2 : 2) * MB * 32 * WPITCH + (stats ? MB * 32 * (WCB + 1) : 0)) * sizeof(float);
    static const char* ldspad = sw_tool("STATTN_PW_LDS");       // tools: force a dynamic LDS size (bytes) -> workgroups per CU
    if (ldspad && (size_t)atol(ldspad) > lds) lds = (size_t)atol(ldspad);
    const dim3 grid(colblocks, RG), block(64 * KS);
#define STATTN_PW(MB_, R_, KS_)                                                              \
    do {                                                                                     \
        hipError_t e_ = pw_allow_lds(panelw_kernel<MB_, R_, KS_>, lds);                      \
        if (e_ != hipSuccess) return e_;                                                     \
        hipLaunchKernelGGL((panelw_kernel<MB_, R_, KS_>), grid, block, lds, s, a);           \
    } while (0)
    // ring depth: (waves per SIMD) x R x MB x 256 MFMA cycles of operands in flight >= ~6000 cycles (2.5 us)
#ifdef STATTN_PROBES
    if (const char* rr = sw_tool("STATTN_PW_R")) {          // tools/panelw_probe.hip: ring depth sweep
        const int R_ = atoi(rr);
#define STATTN_PW_SWEEP(MB_, KS_) \
        if (MB == MB_) { if (R_ == 3) STATTN_PW(MB_, 3, KS_); else if (R_ == 4) STATTN_PW(MB_, 4, KS_); else if (R_ == 5) STATTN_PW(MB_, 5, KS_); \
                         else if (R_ == 6) STATTN_PW(MB_, 6, KS_); else STATTN_PW(MB_, 8, KS_); return hipGetLastError(); }
        STATTN_PW_SWEEP(1, 16) STATTN_PW_SWEEP(5, 4)
#undef STATTN_PW_SWEEP
    }
#endif
    switch (MB) {
        case 1: STATTN_PW(1, 6, 16); break;
        case 2: STATTN_PW(2, 6, 8); break;
        case 3: STATTN_PW(3, 8, 4); break;
        case 4: STATTN_PW(4, 6, 4); break;
        case 5: STATTN_PW(5, 6, 4); break;
        case 6: STATTN_PW(6, 4, 4); break;
        default: STATTN_PW(8, 3, 4); break;
    }
#undef STATTN_PW
    return hipGetLastError();
}

hipError_t launch_lstm_panel_wide(hipStream_t s, const LstmPnArgs& a) {
    const int colblocks = a.D / 8;
    int RG, MB;
    pw_geom(a.M, colblocks, RG, MB);
    const size_t lds = (size_t)2 * MB * 32 * WPITCH * sizeof(float);
    const dim3 grid(colblocks, RG), block(256);
#define STATTN_LPW(MB_, R_)                                                                  \
    do {                                                                                     \
        hipError_t e_ = pw_allow_lds(lstm_panelw_kernel<MB_, R_>, lds);                      \
        if (e_ != hipSuccess) return e_;                                                     \
        hipLaunchKernelGGL((lstm_panelw_kernel<MB_, R_>), grid, block, lds, s, a);           \
    } while (0)
#ifdef STATTN_PROBES
    if (const char* rr = sw_tool("STATTN_PW_R")) {
        const int R_ = atoi(rr);
#define STATTN_LPW_SWEEP(MB_) \
        if (MB == MB_) { if (R_ == 3) STATTN_LPW(MB_, 3); else if (R_ == 4) STATTN_LPW(MB_, 4); else if (R_ == 5) STATTN_LPW(MB_, 5); \
                         else if (R_ == 6) STATTN_LPW(MB_, 6); else STATTN_LPW(MB_, 8); return hipGetLastError(); }
        STATTN_LPW_SWEEP(2) STATTN_LPW_SWEEP(3)
#undef STATTN_LPW_SWEEP
    }
#endif
    switch (MB) {
        case 1: STATTN_LPW(1, 8); break;      // (rings deeper than ~8 steps are slower: 12 steps at two row blocks, 36 KB per wave in
                                              // flight, doubled the launch time: tools/panelw_probe.hip, STATTN_PW_R)
        case 2: STATTN_LPW(2, 6); break;
        case 3: STATTN_LPW(3, 6); break;
        case 4: STATTN_LPW(4, 6); break;
        case 5: STATTN_LPW(5, 5); break;
        case 6: STATTN_LPW(6, 4); break;
        default: STATTN_LPW(8, 3); break;
    }
#undef STATTN_LPW
    return hipGetLastError();
}

// column-tile width of the wide kernel's statistics records
int panel_wide_tile_cols() { return WCB; }

}  // namespace stattn
