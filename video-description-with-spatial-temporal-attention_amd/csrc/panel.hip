// Row-panel fp32 MFMA kernels for the per-timestep dense work of the decoder at batch sizes of up to 512 rows
// (training: 64 rows; beam search: videos x beam rows):
//   state projections h.[Wdl|Wdg|Wdm|Wdlt] and h.U        model_attention.py:371, 389, 402, 415, 437
//   ctx.Wc (+ emb.W) + gates + cell update                 :437-457
//   readout MLP and vocabulary projection (sampling)       :821-838
//   the three transposed recurrences of the reverse scan   (tensor.grad of the above, :1193)
//
// Partition.  A workgroup owns ALL rows of the batch and a narrow panel of 16 (or 32) output columns, so every
// weight byte crosses the L2 -> CU path exactly once per launch and the activations (<= 1 MB, L2-resident) are the
// operand that is re-read.  (The 64-column "skinny" kernels of skinny.hip split the rows over workgroups for
// occupancy; at 64 rows that streamed every weight panel two to four times and left each wave with 4-8 dependent
// load -> MFMA steps: 17-20 us per launch against an MFMA floor of 3.4-6.8 us.)
//
// Weight layout.  The panels are repacked once per forward / backward pass (the weights are constant across the
// time steps) into the order the MFMA B operand is consumed:
//     P[tile c][k-step s][lane][q] = W[16 s + 4 (lane >> 4) + q][col(c, lane & 15)]
// so one k-step of one column tile is ONE coalesced 1 KiB dwordx4 load per wave, with no LDS staging and no
// shuffles.  col(c, j) = 16 c + j, or for the LSTM the gate-interleaved (j >> 2) D + 4 c + (j & 3), which puts the
// four gates of four hidden units into one tile so the cell update runs in the epilogue.
//
// v_mfma_f32_16x16x4_f32 contracts k over the four 16-lane groups g = lane >> 4: a lane loads FOUR consecutive k of
// its A row (one dwordx4), the matching packed B float4 holds the same four k for its column, and the four MFMAs
// of a 16-k step take component q of both.
//
// Block = MG row groups x KS K-slices waves; a wave holds MT m-tiles x NT n-tiles of accumulators and walks every
// KS-th k-step with a three-deep register ring (two steps of loads in flight behind the MFMAs).  K-slices are
// reduced through LDS in a fixed order (deterministic), then the epilogue runs once.
#include "panel_inl.h"

namespace stattn {

#ifdef STATTN_PROBES
__device__ long long* pn_probe = nullptr;
#define PN_STAMP(i) do { if (pn_probe && threadIdx.x == 0) pn_probe[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PN_STAMP(i) do {} while (0)
#endif

namespace {

// Optional epilogue operands are read through a pointer that is ALWAYS valid (the operand, or one of these with a zero stride):
// a load inside `if (bias)` keeps its branch and hipcc puts a full s_waitcnt behind it -- every optional operand was a dependent
// L2 round trip of its own, in the tail (or, for the LSTM kernel's "early" requests, in front) of a 5 - 12 us launch.
__device__ const float pn_zero = 0.f, pn_one = 1.f;

// -DSTATTN_PN_V2=1 (variant library `pnv2`, tools/build_all_variants.sh; NOT the product build, unmeasured): the prologues of panel_kernel
// and lstm_panel_kernel rewritten against their ISA (round 6, tools/isa_skeleton.py):
//  (1) `p ? p : &pn_zero` makes the early epilogue requests FLAT loads (the select mixes a kernel-argument pointer with the address of
//      a __device__ constant), and flat loads count on lgkmcnt as well as vmcnt: every `s_waitcnt lgkmcnt(0)` behind a later scalar
//      load waited for them, and `s_waitcnt vmcnt(0)` stood in front of the ring's first operand request -- the "early" requests were
//      a fully exposed memory round trip at the head of every launch.  V2 loads them through global (address space 1) pointers;
//  (2) the pair loop (1 - 3 operand pairs) is a real loop: loads pending across its head make LLVM's wait-count pass drain the queue
//      there.  V2 unrolls it (straight-line code for the common single pair): no wait between the early requests and the ring;
//  (3) the fields of the dynamically indexed segment / pair were re-read at every use, each an s_load with its own wait: about ten
//      dependent scalar round trips in front of the first operand request.  V2 reads them once, and finds the segment with
//      independent loads of all segment widths instead of one dependent load per segment passed.
// ISA of panel_kernel<4,1,512,4,false>: 20 -> 8 scalar waits before the first operand request, no vector wait there, the main
// loop's staged vmcnt(15..12) unchanged, 156 -> 152 VGPRs.  Same arithmetic in the same order: results are bit-identical.
#ifndef STATTN_PN_V2
#define STATTN_PN_V2 0
#endif
#if STATTN_PN_V2
typedef const __attribute__((address_space(1))) float* pn_gptr;
#define PN_G(p) ((pn_gptr)(p))
#else
#define PN_G(p) (p)
#endif

// ---- general grouped GEMM with fused epilogue ------------------------------------------------------------
template <int MT, int NT, int MAXT, int R, bool ONESHOT>
__global__ __launch_bounds__(MAXT) void panel_kernel(const PnArgs a, const int MG, const int KS) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int CB = 16 * NT;
    // locate the segment of this group of NT column tiles.  Neighbouring tiles share the cache lines of the row-major
    // epilogue operands and of C (a tile's 64 / 128 bytes per row are a fraction of a line): XCD-contiguous tile ranges
    int tg = xcd_contiguous((int)blockIdx.x, (int)gridDim.x), si = 0;
#if STATTN_PN_V2
    {
        int nt[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) nt[i] = a.seg[i].N / CB;        // (segments past nseg: whatever the argument block holds, never used)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const bool adv = si == i && i + 1 < a.nseg && tg >= nt[i];
            tg -= adv ? nt[i] : 0; si += adv ? 1 : 0;
        }
    }
#else
    while (si + 1 < a.nseg && tg >= a.seg[si].N / CB) { tg -= a.seg[si].N / CB; ++si; }
#endif
    const PnSeg& sg = a.seg[si];
#if STATTN_PN_V2
    // (4) the wave index as a SCALAR: as a vector value, `last = s0 + (n - 1) * stride` of pn_accumulate became a v_mad_u64_u32 whose
    //     64-bit addend pair had a pending load's destination as its unused high half -- and the wait-count pass waited for that load
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#endif
    const int j = lane & 15, g = lane >> 4;
    const int mg = w % MG, ks = w / MG;
    const int RB = MG * MT * 16;
    const int kz = a.kz > 1 ? a.kz : 1;

    PN_STAMP(0);
    // the segment's fields, read once: `a.seg[si]` is indexed dynamically, and left inside the loop every field is an s_load
    // from the kernel-argument segment per iteration whose s_waitcnt lgkmcnt(0) also drains the LDS reads (panelw.hip)
    const int n0 = tg * CB;
    const float* const bias = sg.bias ? sg.bias : &pn_zero; const int sbias = sg.bias ? 1 : 0;
    const float* const bias2 = sg.bias2 ? sg.bias2 : &pn_zero; const int sbias2 = sg.bias2 ? 1 : 0;
    const float* const add = sg.add ? sg.add : &pn_zero; const int sadd = sg.add ? 1 : 0;
    const float* const mul = sg.mul ? sg.mul : &pn_one; const int smul = sg.mul ? 1 : 0;
    const int ldadd = sg.ldadd, ldmul = sg.ldmul, M = a.M;
    // The epilogue's operands do not depend on the GEMM: those of the thread's first (row, column) item -- its only one unless
    // the block has fewer threads than the tile has elements -- are requested before it (four loads in flight behind the main
    // loop instead of up to four dependent round trips after it).
    struct Epi { float b, b2, ad, ml; };
    auto epi_load = [&](int idx) {
        const int row = min(idx / CB, M - 1), n = n0 + idx % CB;
        Epi e;
        e.b = PN_G(bias)[n * sbias]; e.b2 = PN_G(bias2)[n * sbias2];
        e.ad = PN_G(add)[((size_t)row * ldadd + n) * sadd]; e.ml = PN_G(mul)[((size_t)row * ldmul + n) * smul];
        return e;
    };
    const Epi e0 = epi_load(min(tid, RB * CB - 1));
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#if STATTN_PN_V2
    const int npairs = sg.npairs;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (p >= npairs) break;
        struct { const float* A; const float* P; int K, lda, apk; } pr;      // the pair's fields, read once
        pr.A = sg.p[p].A; pr.P = sg.p[p].P; pr.K = sg.p[p].K; pr.lda = sg.p[p].lda; pr.apk = sg.p[p].apk;
        const float* Ap[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = min((mg * MT + i) * 16 + j, M - 1);
            const size_t opk = ((size_t)min(mg * MT + i, (M - 1) >> 4) * (pr.K >> 4)) * 256 + 4 * lane;
            Ap[i] = pr.A + (pr.apk ? opk : (size_t)row * pr.lda + 4 * g);
        }
        const int astep = pr.apk ? 256 : 16;
        const int nsteps = pr.K >> 4;
        const size_t tile_floats = (size_t)nsteps * 256;
        const float* Bp = pr.P + (size_t)tg * NT * tile_floats + 4 * lane;
#else
    for (int p = 0; p < sg.npairs; ++p) {
        const PnPair& pr = sg.p[p];
        const float* Ap[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = min((mg * MT + i) * 16 + j, a.M - 1);
            // packed activations (pn_pack_offset): one coalesced 1 KiB run per m-tile and k-step, like the weights
            // (a row group may reach past the last m-tile: clamped like the rows, those accumulators are never stored)
            Ap[i] = pr.apk ? pr.A + ((size_t)min(mg * MT + i, (a.M - 1) >> 4) * (pr.K >> 4)) * 256 + 4 * lane
                           : pr.A + (size_t)row * pr.lda + 4 * g;
        }
        const int astep = pr.apk ? 256 : 16;
        const int nsteps = pr.K >> 4;
        const size_t tile_floats = (size_t)nsteps * 256;
        const float* Bp = pr.P + (size_t)tg * NT * tile_floats + 4 * lane;
#endif
#if defined(STATTN_PROBES) && PN_VARIANT == 6
        // tools/panel_probe.hip, VERDICT r03 item 3(a): what the main loop costs when the workgroup's weight slice is ALREADY in
        // LDS (a persistent kernel would keep it there across the decoder steps): the slice is copied in before the timed part
        {
            float* wl = red + (size_t)KS * RB * CB;
            for (size_t i = tid; i < tile_floats * NT / 4; i += blockDim.x) st4(wl + 4 * i, ld4(pr.P + (size_t)tg * NT * tile_floats + 4 * i));
            __syncthreads();
            PN_STAMP(0);
            Bp = wl + 4 * lane;
        }
#endif
        pn_accumulate<MT, NT, R, ONESHOT>(acc, Ap, astep, Bp, tile_floats, nsteps, (int)blockIdx.y * KS + ks, KS * kz,
                                          pn_rotation((int)blockIdx.x, nsteps), a.stream_b != 0);
    }
    PN_STAMP(1);
    pn_spill<MT, NT>(red, RB, ks, mg, acc, j, g);
    __syncthreads();
    PN_STAMP(2);

    if (sg.stats) {   // vocabulary statistics of this column tile (small-batch decode: the logits are never stored)
        // final biased values -> the extra RB x CB block behind the K-slice partials (launch_panel sizes it in)
        float* fin = red + (size_t)KS * RB * CB;
        for (int idx = tid; idx < RB * CB; idx += blockDim.x) {
            const int row = idx / CB, col = idx % CB;
            float v = 0.f;
            for (int k = 0; k < KS; ++k) v += red[((size_t)k * RB + row) * CB + col];
            const int n = n0 + col;
            v += idx == tid ? e0.b : bias[n * sbias];
            if (n >= sg.stats_V || (sg.stats_skip0 && n == 0)) v = -INFINITY;
            fin[idx] = v;
        }
        __syncthreads();
        const int nw = blockDim.x >> 6;
        const int ntile = sg.N / CB;
        for (int row = w; row < a.M; row += nw) {        // one wave per row: lane = column of the tile
            float v = lane < CB ? fin[row * CB + lane] : -INFINITY;
            const float mx = wave_max(v);
            const float se = wave_sum(v > -INFINITY ? __expf(v - mx) : 0.f);
            float* rec = sg.stats + ((size_t)row * ntile + tg) * PN_STATS_REC;
            if (lane == 0) { rec[0] = mx; rec[1] = se; }
            if (sg.stats_seed) {                          // ancestral sampling: arg-max of v + Gumbel noise within the tile
                const float pv = v > -INFINITY ? v + gumbel01(*sg.stats_seed, (unsigned long long)(*sg.stats_step) * 65536ull + row, n0 + lane) : -INFINITY;
                const float m = wave_max(pv);
                const unsigned long long hit = __ballot(pv == m);
                const int src = __ffsll((long long)hit) - 1;
                const float vsrc = __shfl(v, src, 64);
                if (lane == 0) { rec[2] = m; rec[3] = vsrc; reinterpret_cast<int*>(rec)[2 + PN_STATS_KB] = n0 + src; }
                continue;
            }
            for (int i = 0; i < sg.stats_kb; ++i) {       // the kb largest, ties to the lower column
                const float m = wave_max(v);
                const unsigned long long hit = __ballot(v == m);
                const int src = __ffsll((long long)hit) - 1;
                if (lane == 0) { rec[2 + i] = m; reinterpret_cast<int*>(rec)[2 + PN_STATS_KB + i] = n0 + src; }
                if (lane == src) v = -INFINITY;
            }
        }
        PN_STAMP(3);
        return;
    }
    float* const Cp = sg.C; float* const Cpk = sg.Cpk;
    const int ldc = sg.ldc, act = sg.act, Spk = sg.N >> 4;
    const float scale = sg.scale;
    const size_t pstride = a.part_stride;
    for (int idx = tid; idx < RB * CB; idx += blockDim.x) {
        const int row = idx / CB, col = idx % CB;
        if (row >= M) continue;
        float v = 0.f;
        for (int k = 0; k < KS; ++k) v += red[((size_t)k * RB + row) * CB + col];
        const int n = n0 + col;
        if (kz > 1) { Cp[(size_t)blockIdx.y * pstride + (size_t)row * ldc + n] = v; continue; }
        const Epi e = idx == tid ? e0 : epi_load(idx);
        v += e.b;
        v += e.b2;
        v += e.ad;
        if (act == 1) v = fast_tanh(v);
        v *= scale;
        v *= e.ml;
        Cp[(size_t)row * ldc + n] = v;
        if (Cpk) Cpk[pn_pack_offset(row, n, Spk)] = v;
    }
    PN_STAMP(3);
}

// ---- LSTM cell with its GEMM (model_attention.py:437-457).  Column tile c = units 4c..4c+3 of all four gates
// (packed with PN_COLS_LSTM: tile column jj = gate * 4 + u).
template <int MT, int MAXT, int R, bool ONESHOT>
__global__ __launch_bounds__(MAXT) void lstm_panel_kernel(const LstmPnArgs a, const int MG, const int KS) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int CB = 16;
    const int D = a.D;
    // XCD-contiguous tile ranges: the epilogue reads 16-byte pieces (four units) of row-major [M, 4D] / [M, D] operands --
    // with tile c on XCD c % 8 the four to eight tiles that share a line sat on as many XCDs and every one of them
    // fetched the line for itself (measured: 40 MB through the fabric per launch for 20 MB of distinct bytes)
    const int c = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
#if STATTN_PN_V2
    // (4) the wave index as a SCALAR: as a vector value, `last = s0 + (n - 1) * stride` of pn_accumulate became a v_mad_u64_u32 whose
    //     64-bit addend pair had a pending load's destination as its unused high half -- and the wait-count pass waited for that load
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#endif
    const int j = lane & 15, g = lane >> 4;
    const int mg = w % MG, ks = w / MG;
    const int RB = MG * MT * 16;

    PN_STAMP(0);
    // The epilogue's operands do not depend on the GEMM: those of the thread's first (row, unit) item -- its only one
    // unless the block has fewer K-slice waves than m-tiles -- are requested before it, so that the cell update at the
    // end is arithmetic only (it was a chain of ten dependent global loads: 4.4 us of a 17 us launch).
    // (Every load unconditional, through pointers that are always valid, and no arithmetic on the loaded values here: with
    // `a.bias ? ... : 0` / `if (a.bias) v += ...` hipcc kept the branches and waited for each load where it stood -- the
    // "early requests" were four to six dependent round trips IN FRONT of the main loop.)
    struct EpiIn { float add[4], bias[4], dp[3], cp, hp, m, d1; };
    const float* const padd = a.pre_add ? a.pre_add : &pn_zero; const int sadd = a.pre_add ? 1 : 0;
    const float* const pbias = a.bias ? a.bias : &pn_zero; const int sbias = a.bias ? 1 : 0;
    const float* const pmask = a.mask ? a.mask : &pn_one; const int smask = a.mask ? 1 : 0;
    const float* const pd1 = a.d1 ? a.d1 : &pn_one; const int sd1 = a.d1 ? 1 : 0;
    auto epi_load = [&](int idx) {
        EpiIn e;
        const int row = min(idx >> 2, a.M - 1), d = 4 * c + (idx & 3);
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            e.add[gate] = PN_G(padd)[((size_t)row * a.ldpre + gate * D + d) * sadd];
            e.bias[gate] = PN_G(pbias)[(gate * D + d) * sbias];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) e.dp[q] = a.dp[(size_t)row * a.lddp + q * D + d];
        e.cp = a.c_prev[(size_t)row * D + d];
        e.hp = a.h_prev[(size_t)row * D + d];
        e.m = PN_G(pmask)[row * smask];
        e.d1 = PN_G(pd1)[((size_t)row * a.ldd1 + d) * sd1];
        return e;
    };
    const EpiIn e0 = epi_load(min(tid, RB * 4 - 1));
    f32x4 acc[MT][1];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#if STATTN_PN_V2
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (p >= a.npairs) break;
#else
    for (int p = 0; p < a.npairs; ++p) {
#endif
        const PnPair& pr = a.p[p];
        const float* Ap[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = min((mg * MT + i) * 16 + j, a.M - 1);
            // packed activations (pn_pack_offset): one coalesced 1 KiB run per m-tile and k-step, like the weights
            // (a row group may reach past the last m-tile: clamped like the rows, those accumulators are never stored)
            Ap[i] = pr.apk ? pr.A + ((size_t)min(mg * MT + i, (a.M - 1) >> 4) * (pr.K >> 4)) * 256 + 4 * lane
                           : pr.A + (size_t)row * pr.lda + 4 * g;
        }
        const int astep = pr.apk ? 256 : 16;
        const int nsteps = pr.K >> 4;
        const size_t tile_floats = (size_t)nsteps * 256;
        pn_accumulate<MT, 1, R, ONESHOT>(acc, Ap, astep, pr.P + (size_t)c * tile_floats + 4 * lane, tile_floats, nsteps, ks, KS,
                                         pn_rotation(c, nsteps));
    }
    PN_STAMP(1);
    pn_spill<MT, 1>(red, RB, ks, mg, acc, j, g);
    __syncthreads();
    PN_STAMP(2);

    for (int idx = tid; idx < RB * 4; idx += blockDim.x) {
        const int row = idx >> 2, u = idx & 3;
        if (row >= a.M) continue;
        const EpiIn e = idx == tid ? e0 : epi_load(idx);
        const int d = 4 * c + u;
        float pre[4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            float v = 0.f;
            for (int k = 0; k < KS; ++k) v += red[((size_t)k * RB + row) * CB + gate * 4 + u];
            pre[gate] = v + (e.add[gate] + e.bias[gate]);
        }
        // dropout multiplies the i/f/o PRE-activations (:444-447); g gets none
        const float gi = fast_sigmoid(pre[0] * e.dp[0]);
        const float gf = fast_sigmoid(pre[1] * e.dp[1]);
        const float go = fast_sigmoid(pre[2] * e.dp[2]);
        const float gg = fast_tanh(pre[3]);
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
    PN_STAMP(3);
}

// ---- repacking.  One thread per packed float4.
// src_t = 0: W is [K][ldw] (k-major rows): the four k of a float4 are four strided reads;
// src_t = 1: the operand is W^T with W stored [N][ldw]: the four k are contiguous in one row of W (the reverse-scan
//            recurrences use the transposed weights without ever materialising a transpose).
__device__ __forceinline__ void pack_one(const PackJob& jb, size_t idx);
__global__ __launch_bounds__(256) void pack_panels_kernel(const PackJob jb) { pack_one(jb, (size_t)blockIdx.x * 256 + threadIdx.x); }
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackBatch b) {
    int q = 0;
    while (q + 1 < b.n && blockIdx.x >= b.blk0[q + 1]) ++q;
    pack_one(b.j[q], (size_t)(blockIdx.x - b.blk0[q]) * 256 + threadIdx.x);
}
__device__ __forceinline__ void pack_one(const PackJob& jb, size_t idx) {
    const int S = jb.K >> 4;
    const size_t total = (size_t)jb.ntiles * S * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int s = (int)((idx >> 6) % S);
    const int c = (int)((idx >> 6) / S);
    const int j = lane & 15, g = lane >> 4;
    const int col = jb.cols == PN_COLS_LSTM ? (j >> 2) * jb.D + 4 * c + (j & 3) : 16 * c + j;
    const int k = 16 * s + 4 * g;
    float4 v;
    if (jb.src_t) {
        v = ld4(jb.W + (size_t)col * jb.ldw + k);
    } else {
        const float* p = jb.W + (size_t)k * jb.ldw + col;
        v = make_float4(p[0], p[jb.ldw], p[2 * (size_t)jb.ldw], p[3 * (size_t)jb.ldw]);
    }
    // dst tile pitch: S_total k-steps (this job may fill only the k-range [s_off, s_off + S) of a taller panel)
    st4(jb.dst + (((size_t)c * jb.S_total + jb.s_off + s) * 64 + lane) * 4, v);
}

// activations [M][ld] -> packed A layout; one thread per packed float4, rows past M are zero
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, int ld, int M, int K, float* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int S = K >> 4, mtiles = (M + 15) >> 4;
    if (idx >= (size_t)mtiles * S * 64) return;
    const int lane = (int)(idx & 63), s = (int)((idx >> 6) % S), mt = (int)((idx >> 6) / S);
    const int row = mt * 16 + (lane & 15), k = 16 * s + 4 * (lane >> 4);
    st4(dst + idx * 4, row < M ? ld4(src + (size_t)row * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f));
}

struct PnGeom { int MT, MG, KS; };

// more than 64 KiB of dynamic LDS has to be granted per kernel, once
template <class F>
hipError_t pn_allow_lds(F f, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#ifndef PN_RING
#define PN_RING 4
#endif
constexpr int PN_R = PN_RING;   // k-steps of operands a wave keeps in flight (sweep: tools/panel_probe.hip -DPN_RING=n)

// rows -> m-tiles per wave and row groups; K-slices fill the block up to 8 (M <= 64) / 12..16 waves, but no more than
// give every wave a full ring of steps (min_steps = k-steps of the shortest pair per K-split block)
PnGeom pn_geom(int M, int max_waves, int min_steps) {
    const int mtiles = (M + 15) / 16;
    PnGeom q;
    // up to 64 rows: one row group, 4 m-tiles per wave (512-thread blocks, 148 VGPRs).  More rows: several row groups
    // of 2 m-tiles (1024-thread blocks leave 128 VGPRs per lane; 4 m-tiles would spill)
    q.MT = mtiles > 4 ? 2 : (mtiles >= 3 ? 4 : (mtiles == 2 ? 2 : 1));
    q.MG = (mtiles + q.MT - 1) / q.MT;
    q.KS = max_waves / q.MG;
    if (q.KS > 8) q.KS = 8;
    while (q.KS > 1 && min_steps / q.KS < 4) q.KS >>= 1;
    if (q.KS < 1) q.KS = 1;
    return q;
}

}  // namespace

bool panel_supported(int M) { return M >= 1 && M <= 512; }    // 16 row groups of two m-tiles in a 1024-thread workgroup

size_t packed_rows_floats(int M, int K) { return (size_t)((M + 15) / 16) * 16 * K; }

hipError_t launch_pack_rows(hipStream_t s, const float* src, int ld, int M, int K, float* dst) {
    if (M <= 0 || K % 16 != 0 || ld % 4 != 0) return hipErrorInvalidValue;
    const size_t total = (size_t)((M + 15) / 16) * (K >> 4) * 64;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ld, M, K, dst);
    return hipGetLastError();
}

hipError_t launch_pack_panels(hipStream_t s, const PackJob& jb) {
    if (jb.K % 16 != 0 || jb.ntiles <= 0 || !jb.W || !jb.dst) return hipErrorInvalidValue;
    if (jb.src_t && (jb.ldw % 4 != 0)) return hipErrorInvalidValue;
    const size_t total = (size_t)jb.ntiles * (jb.K >> 4) * 64;
    hipLaunchKernelGGL(pack_panels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, jb);
    return hipGetLastError();
}

bool pack_batch_add(PackBatch& b, const PackJob& jb) {
    if (b.n >= PACK_BATCH_MAX || jb.K % 16 != 0 || jb.ntiles <= 0 || !jb.W || !jb.dst || (jb.src_t && (jb.ldw % 4 != 0))) return false;
    const size_t total = (size_t)jb.ntiles * (jb.K >> 4) * 64;
    if (b.n == 0) b.blk0[0] = 0;
    b.j[b.n] = jb;
    b.blk0[b.n + 1] = b.blk0[b.n] + (unsigned)((total + 255) / 256);
    ++b.n;
    return true;
}
hipError_t launch_pack_batch(hipStream_t s, const PackBatch& b) {
    if (b.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_batch_kernel, dim3(b.blk0[b.n]), dim3(256), 0, s, b);
    return hipGetLastError();
}

void pn_seg_defaults(PnSeg& s) {
    s = PnSeg{};
    s.scale = 1.f;
}

// column-tile width launch_panel uses for `a` (16, or 32 when that still fills the chip): what PnSeg::stats records are
// counted in (N / width per row)
int panel_tile_cols(const PnArgs& a) {
    if (panel_wide_supported(a)) return panel_wide_tile_cols();
    int tiles = 0, min_steps = 1 << 30;
    const int kz = a.kz > 1 ? a.kz : 1;
    for (int i = 0; i < a.nseg; ++i) {
        tiles += a.seg[i].N / 16;
        for (int p = 0; p < a.seg[i].npairs; ++p) {
            const int st = (a.seg[i].p[p].K / 16 + kz - 1) / kz;
            min_steps = st < min_steps ? st : min_steps;
        }
    }
    const PnGeom q = pn_geom(a.M, 16, min_steps);
    bool nt2 = tiles * kz >= 512 && q.MG == 1;
    for (int i = 0; i < a.nseg; ++i) nt2 = nt2 && (a.seg[i].N % 32 == 0);
    return nt2 ? 32 : 16;
}

hipError_t launch_panel(hipStream_t s, const PnArgs& a) {
    if (a.M <= 0 || a.nseg <= 0) return hipSuccess;
    if (!panel_supported(a.M)) return hipErrorInvalidValue;
    if (panel_wide_supported(a)) return launch_panel_wide(s, a);
    int tiles = 0;
    for (int i = 0; i < a.nseg; ++i) {
        const PnSeg& sg = a.seg[i];
        if (sg.N % 16 != 0 || sg.npairs < 1 || sg.npairs > 3) return hipErrorInvalidValue;
        for (int p = 0; p < sg.npairs; ++p)
            if (sg.p[p].K % 16 != 0 || sg.p[p].lda % 4 != 0) return hipErrorInvalidValue;
        tiles += sg.N / 16;
    }
    const int kz = a.kz > 1 ? a.kz : 1;
    int min_steps = 1 << 30, max_steps = 0;
    for (int i = 0; i < a.nseg; ++i)
        for (int p = 0; p < a.seg[i].npairs; ++p) {
            const int st = (a.seg[i].p[p].K / 16 + kz - 1) / kz;
            min_steps = st < min_steps ? st : min_steps; max_steps = st > max_steps ? st : max_steps;
        }
    // two column tiles per workgroup (A operands loaded once for twice the MFMAs) when that still fills the chip;
    // only for a single row group (512-thread blocks: the register budget of 1024-thread blocks is half)
    const PnGeom q = pn_geom(a.M, 16, min_steps);
    const bool oneshot = (max_steps + q.KS - 1) / q.KS <= PN_R;
    bool nt2 = tiles * kz >= 512 && q.MG == 1;
    for (int i = 0; i < a.nseg; ++i) nt2 = nt2 && (a.seg[i].N % 32 == 0);
    const int CB = nt2 ? 32 : 16;
    dim3 grid(tiles / (nt2 ? 2 : 1), kz), block(64 * q.MG * q.KS);
    bool stats = false;
    for (int i = 0; i < a.nseg; ++i) stats = stats || a.seg[i].stats;
    if (stats && (q.MG != 1 || kz != 1)) return hipErrorInvalidValue;      // every row in one row group, no K split over blocks
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].stats && (a.seg[i].stats_kb < 1 || a.seg[i].stats_kb > PN_STATS_KB)) return hipErrorInvalidValue;
    size_t lds = ((size_t)q.KS + (stats ? 1 : 0)) * q.MG * q.MT * 16 * CB * sizeof(float);
#if defined(STATTN_PROBES) && PN_VARIANT == 6
    lds += (size_t)max_steps * kz * 256 * (nt2 ? 2 : 1) * sizeof(float);       // the workgroup's weight slice (tools/panel_probe.hip)
#endif
#define STATTN_PN_LAUNCH1(MT_, NT_, MAXT_, R_, OS_)                                                         \
    do {                                                                                                    \
        hipError_t e_ = pn_allow_lds(panel_kernel<MT_, NT_, MAXT_, R_, OS_>, lds);                          \
        if (e_ != hipSuccess) return e_;                                                                    \
        hipLaunchKernelGGL((panel_kernel<MT_, NT_, MAXT_, R_, OS_>), grid, block, lds, s, a, q.MG, q.KS);   \
    } while (0)
    // 1024-thread blocks (several row groups): 128 VGPRs per lane -> a ring of 4 steps
#define STATTN_PN_LAUNCH(MT_, NT_, MAXT_)                                                                   \
    do {                                                                                                    \
        if (MAXT_ == 1024) { if (oneshot4) STATTN_PN_LAUNCH1(MT_, NT_, MAXT_, 4, true); else STATTN_PN_LAUNCH1(MT_, NT_, MAXT_, 4, false); } \
        else if (oneshot) STATTN_PN_LAUNCH1(MT_, NT_, MAXT_, PN_R, true);                                   \
        else STATTN_PN_LAUNCH1(MT_, NT_, MAXT_, PN_R, false);                                               \
    } while (0)
    const bool oneshot4 = (max_steps + q.KS - 1) / q.KS <= 4;
    if (q.MG > 1) STATTN_PN_LAUNCH(2, 1, 1024);
    else if (nt2) {
        if (q.MT == 4) STATTN_PN_LAUNCH(4, 2, 512); else if (q.MT == 2) STATTN_PN_LAUNCH(2, 2, 512); else STATTN_PN_LAUNCH(1, 2, 512);
    } else {
        if (q.MT == 4) STATTN_PN_LAUNCH(4, 1, 512); else if (q.MT == 2) STATTN_PN_LAUNCH(2, 1, 512); else STATTN_PN_LAUNCH(1, 1, 512);
    }
#undef STATTN_PN_LAUNCH
#undef STATTN_PN_LAUNCH1
    return hipGetLastError();
}

hipError_t launch_lstm_panel(hipStream_t s, const LstmPnArgs& a) {
    if (a.M <= 0) return hipSuccess;
    if (!panel_supported(a.M) || a.D % 4 != 0 || a.npairs < 1 || a.npairs > 3) return hipErrorInvalidValue;
    for (int p = 0; p < a.npairs; ++p)
        if (a.p[p].K % 16 != 0 || a.p[p].lda % 4 != 0) return hipErrorInvalidValue;
    if (lstm_panel_wide_supported(a)) return launch_lstm_panel_wide(s, a);
    int min_steps = 1 << 30, max_steps = 0;
    for (int p = 0; p < a.npairs; ++p) {
        const int st = a.p[p].K / 16;
        min_steps = st < min_steps ? st : min_steps; max_steps = st > max_steps ? st : max_steps;
    }
    const PnGeom q = pn_geom(a.M, 16, min_steps);
    const bool oneshot = (max_steps + q.KS - 1) / q.KS <= PN_R, oneshot4 = (max_steps + q.KS - 1) / q.KS <= 4;
    dim3 grid(a.D / 4), block(64 * q.MG * q.KS);
    const size_t lds = (size_t)q.KS * q.MG * q.MT * 16 * 16 * sizeof(float);
#define STATTN_LP_LAUNCH1(MT_, MAXT_, R_, OS_)                                                              \
    do {                                                                                                    \
        hipError_t e_ = pn_allow_lds(lstm_panel_kernel<MT_, MAXT_, R_, OS_>, lds);                          \
        if (e_ != hipSuccess) return e_;                                                                    \
        hipLaunchKernelGGL((lstm_panel_kernel<MT_, MAXT_, R_, OS_>), grid, block, lds, s, a, q.MG, q.KS);   \
    } while (0)
#define STATTN_LP_LAUNCH(MT_, MAXT_)                                                                        \
    do {                                                                                                    \
        if (MAXT_ == 1024) { if (oneshot4) STATTN_LP_LAUNCH1(MT_, MAXT_, 4, true); else STATTN_LP_LAUNCH1(MT_, MAXT_, 4, false); } \
        else if (oneshot) STATTN_LP_LAUNCH1(MT_, MAXT_, PN_R, true);                                        \
        else STATTN_LP_LAUNCH1(MT_, MAXT_, PN_R, false);                                                    \
    } while (0)
    if (q.MG > 1) STATTN_LP_LAUNCH(2, 1024);
    else if (q.MT == 4) STATTN_LP_LAUNCH(4, 512);
    else if (q.MT == 2) STATTN_LP_LAUNCH(2, 512);
    else STATTN_LP_LAUNCH(1, 512);
#undef STATTN_LP_LAUNCH
#undef STATTN_LP_LAUNCH1
    return hipGetLastError();
}

}  // namespace stattn
