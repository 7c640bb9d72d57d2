// Device-side beam search bookkeeping for the batched sampler (gfx950).
//
// The reference decodes one video at a time: per word one f_next call, then on the host
// `cand = hyp_scores[:,None] - log(next_p)`, a flat argsort over live_k x V candidates, and Python
// list surgery (model_attention.py:921-985) -- ~0.5 ms of numpy per word at k = 5, V = 12k, more than
// the whole decoder step on the GPU.  Here the same bookkeeping runs on the device for many videos at
// once (one workgroup per video), so a decode step is a fixed kernel sequence with no host round trip:
//   beam_topk_part      the (k - dead_k) smallest candidate costs of a video per vocabulary slice (:921-928)
//   beam_update_kernel  slice winners merged; new hypotheses, finished ones (word 0) retired, h/c gathered    (:923-985)
#include "kernels.h"
#include "devmath.h"

namespace stattn {

#ifdef STATTN_PROBES
// timeline of beam_update_kernel (make PROBES=1; tools/beam_probe.py): workgroup 0, thread 0 stamps the 100 MHz wall clock
__device__ long long* bm_probe = nullptr;
#define BM_STAMP(i) do { if (bm_probe && threadIdx.x == 0 && blockIdx.x == 0) bm_probe[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BM_STAMP(i) do {} while (0)
#endif

}  // namespace stattn
#include "beam_inl.h"

namespace stattn {
namespace {

// n rounds of a workgroup-wide arg-min over per-thread sorted lists (lc/li ascending, KB long): round r's winner is
// reported through res_c/res_i[r] (thread 0) and popped from its owner's list
__device__ __forceinline__ void block_select(float (&lc)[KB], int (&li)[KB], int n, float* s_cost, int* s_idx, int* s_owner,
                                             float* res_c, int* res_i) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int r = 0; r < n; ++r) {
        float c = lc[0]; int idx = li[0]; int owner = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float oc = __shfl_xor(c, o, 64); const int oi = __shfl_xor(idx, o, 64); const int oo = __shfl_xor(owner, o, 64);
            if (cand_less(oc, oi, c, idx)) { c = oc; idx = oi; owner = oo; }
        }
        if (lane == 0) { s_cost[w] = c; s_idx[w] = idx; s_owner[w] = owner; }
        __syncthreads();
        c = s_cost[0]; idx = s_idx[0]; owner = s_owner[0];
        const int nwv = (int)blockDim.x >> 6;
        for (int i = 1; i < nwv; ++i)
            if (cand_less(s_cost[i], s_idx[i], c, idx)) { c = s_cost[i]; idx = s_idx[i]; owner = s_owner[i]; }
        if (tid == 0) { res_c[r] = c; res_i[r] = idx; }
        if (tid == owner) {
#pragma unroll
            for (int i = 0; i < KB - 1; ++i) { lc[i] = lc[i + 1]; li[i] = li[i + 1]; }
            lc[KB - 1] = INFINITY; li[KB - 1] = 0x7fffffff;
        }
        __syncthreads();
    }
}

// Stage 1: workgroup (video v, slice sp) finds the n best candidates among the words of its slice, for all live
// hypotheses of the video.  One workgroup per video walked its 60 k candidates as a chain of dependent loads
// (243 us per word at k = 5, V = 12 k -- two thirds of the whole decode step); slices cut that to a few loads per
// thread.  Any member of the global top n is in the top n of its slice, so stage 2 sees every winner.
__global__ __launch_bounds__(256) void beam_topk_part_kernel(const BeamArgs a, int nsplit, float* __restrict__ pcost,
                                                             int* __restrict__ pidx) {
    __shared__ float s_cost[4];
    __shared__ int s_idx[4];
    __shared__ int s_owner[4];
    __shared__ float res_c[KB];
    __shared__ int res_i[KB];
    const int v = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
    const int k = a.k, V = a.V;
    const int live = a.live_k[v], dead = a.dead_k[v];
    const int n = live > 0 ? k - dead : 0;                    // how many candidates survive (:923)
    float* oc = pcost + ((size_t)v * nsplit + sp) * KB;
    int* oi = pidx + ((size_t)v * nsplit + sp) * KB;
    if (tid < KB) { res_c[tid] = INFINITY; res_i[tid] = 0x7fffffff; }
    __syncthreads();
    if (n > 0) {
        const int chunk = ((V + nsplit - 1) / nsplit + 3) & ~3;
        const int w0 = sp * chunk, w1 = min(V, w0 + chunk);
        float lc[KB]; int li[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) { lc[i] = INFINITY; li[i] = 0x7fffffff; }
        for (int j = 0; j < live; ++j) {
            const float hs = a.hyp_score[v * k + j];
            const float* __restrict__ p = a.probs + (size_t)(v * k + j) * a.ldp;
            for (int wd = w0 + tid; wd < w1; wd += 256) {
                float pr = p[wd];
                if (a.suppress_eos && wd == 0) pr = 0.f;
                float cst = hs - logf(pr);                        // hyp_scores[:,None] - log(next_p)  (:921), float32
                if (cst != cst) cst = INFINITY;                   // NaN sorts last like in numpy's argsort (behind every finite
                                                                  // cost, ties by index) instead of never being selected
                list_insert(lc, li, cst, j * V + wd);
            }
        }
        block_select(lc, li, n, s_cost, s_idx, s_owner, res_c, res_i);
    }
    if (tid < KB) { oc[tid] = res_c[tid]; oi[tid] = res_i[tid]; }
}

// Stage 2 + bookkeeping, one workgroup per video (beam_inl.h)
__global__ __launch_bounds__(1024) void beam_update_kernel(const BeamArgs a, int nsplit, const float* __restrict__ pcost,
                                                          const int* __restrict__ pidx) {
    beam_update_body(a, nsplit, pcost, pidx, (int)blockIdx.x, (int)gridDim.x);
}


__global__ __launch_bounds__(256) void beam_init_kernel(const BeamInitArgs a) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    const int M = a.nvid * a.k, D = a.D;
    if (gid < (size_t)M) {
        const int i = (int)gid;
        a.vid[i] = i / a.k;
        a.next_w[i] = -1;
        a.score0[i] = 0.f;
        if (a.rowmap) a.rowmap[i] = i;
    }
    if (gid < (size_t)a.nvid) { a.live_k[gid] = 1; a.dead_k[gid] = 0; }
    if (gid == 0) { *a.ticket = 0; *a.step = 0; }
    for (size_t i = gid; i < (size_t)M * D; i += gsz) {
        const int row = (int)(i / D), d = (int)(i - (size_t)row * D), v = row / a.k;
        const bool first = row == v * a.k;
        const float hv = first ? a.h0[(size_t)v * D + d] : 0.f, cv = first ? a.c0[(size_t)v * D + d] : 0.f;
        a.hp[i] = hv; a.cp[i] = cv;
        if (a.hp_pk) a.hp_pk[pn_pack_offset(row, d, D >> 4)] = hv;
    }
    if (a.hp_pk) {   // rows M .. roundup(M, 16) of the packed panel are read (and ignored) by the last m-tile: keep them finite
        const int Mp = (M + 15) & ~15;
        for (size_t i = gid; i < (size_t)(Mp - M) * D; i += gsz) {
            const int row = M + (int)(i / D), d = (int)(i % D);
            a.hp_pk[pn_pack_offset(row, d, D >> 4)] = 0.f;
        }
    }
    for (size_t i = gid; i < (size_t)M * 3 * D; i += gsz) a.dp[i] = 0.5f;
    if (a.emb) for (size_t i = gid; i < (size_t)M * a.E; i += gsz) a.emb[i] = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if (a.zero[q]) for (size_t i = gid; i < a.zero_n[q]; i += gsz) a.zero[q][i] = 0.f;
}

}  // namespace

int beam_topk_splits(int nvid) {          // slices of the vocabulary per video: ~512 workgroups in all, <= 256 / KB
    int ns = 512 / (nvid > 0 ? nvid : 1);
    if (ns > 256 / KB) ns = 256 / KB;
    if (ns < 1) ns = 1;
    return ns;
}
hipError_t launch_beam_topk(hipStream_t s, const BeamArgs& a, float* part_cost, int* part_idx) {
    if (a.k > KB || a.k < 1) return hipErrorInvalidValue;
    const int ns = beam_topk_splits(a.nvid);
    hipLaunchKernelGGL(beam_topk_part_kernel, dim3(a.nvid, ns), dim3(256), 0, s, a, ns, part_cost, part_idx);
    return hipGetLastError();
}
// Vocabulary statistics from STORED logits (round 5): the records the row-panel kernels' statistics epilogue leaves -- per (row, tile of
// 32 columns): max, sum exp(v - max), the kb best values and their columns -- for launches whose logits come from the LDS-tiled GEMM
// (beams of more than 64 rows: 160 x 20 096 x 512 runs 41 us there against ~70 in the wide row-panel kernel, whose 628 workgroups
// each re-read the 327 KB activation panel from L2).  One lane per (row, tile): a wave covers 64 consecutive tiles of one row.
__global__ __launch_bounds__(64) void vocab_stats_kernel(const float* __restrict__ lg, int ldl, int M, int V, int ntile, int kb, int skip0,
                                                         float* __restrict__ stats) {
    const int row = blockIdx.y, tile = blockIdx.x * 64 + threadIdx.x;
    if (tile >= ntile) return;
    const int n0 = tile * 32;
    const float* p = lg + (size_t)row * ldl + n0;
    float fv[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = ld4(p + 4 * j);
        fv[4 * j] = v.x; fv[4 * j + 1] = v.y; fv[4 * j + 2] = v.z; fv[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) if (n0 + c >= V || (skip0 && n0 + c == 0)) fv[c] = -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 32; ++c) mx = fmaxf(mx, fv[c]);
    float se = 0.f;
    float lv[PN_STATS_KB]; int lc[PN_STATS_KB];
#pragma unroll
    for (int i = 0; i < PN_STATS_KB; ++i) { lv[i] = -INFINITY; lc[i] = n0; }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const float v = fv[c];
        se += v > -INFINITY ? __expf(v - mx) : 0.f;
        // branch-free sorted insert, strict >: ties keep the lower column ahead (the order of panelw.hip's epilogue)
        float cv = v; int cc = n0 + c;
#pragma unroll
        for (int i = PN_STATS_KB - 1; i >= 0; --i) {
            const bool sw = cv > lv[i];
            if (i < PN_STATS_KB - 1) { lv[i + 1] = sw ? lv[i] : cv; lc[i + 1] = sw ? lc[i] : cc; }
            cv = sw ? cv : lv[i]; cc = sw ? cc : lc[i];
            if (i == 0) { lv[0] = cv; lc[0] = cc; }
        }
    }
    float* rec = stats + ((size_t)row * ntile + tile) * PN_STATS_REC;
    rec[0] = mx; rec[1] = se;
#pragma unroll
    for (int i = 0; i < PN_STATS_KB; ++i)
        if (i < kb) { rec[2 + i] = lv[i]; reinterpret_cast<int*>(rec)[2 + PN_STATS_KB + i] = lc[i]; }
}

hipError_t launch_vocab_stats(hipStream_t s, const float* lg, int ldl, int M, int V, int ntile, int kb, int skip0, float* stats) {
    if (M <= 0 || ntile <= 0) return hipSuccess;
    if (kb < 1 || kb > PN_STATS_KB || ldl % 4 || ntile * 32 > ldl) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vocab_stats_kernel, dim3((ntile + 63) / 64, M), dim3(64), 0, s, lg, ldl, M, V, ntile, kb, skip0, stats);
    return hipGetLastError();
}

hipError_t launch_beam_update(hipStream_t s, const BeamArgs& a, const float* part_cost, const int* part_idx) {
    if (!a.ticket) return hipErrorInvalidValue;
    if (a.stats && (a.ntile < 1 || a.k > PN_STATS_KB || (a.stochastic && (a.k != 1 || a.tile_cols < 1)))) return hipErrorInvalidValue;
    if (a.proj_next && (!a.proj_step || a.nproj % 4)) return hipErrorInvalidValue;
    BeamArgs u = a;
    if (u.rw_cost && !(u.stats && u.k > 1 && u.rw_idx && u.rw_ticket)) u.rw_cost = nullptr;     // row workgroups: statistics mode, k > 1 (1024 threads)
    hipLaunchKernelGGL(beam_update_kernel, dim3(u.rw_cost ? u.nvid * u.k : u.nvid), dim3(u.stats ? 1024 : 256), 0, s, u, beam_topk_splits(u.nvid), part_cost, part_idx);
    return hipGetLastError();
}

hipError_t launch_beam_init(hipStream_t s, const BeamInitArgs& a) {
    if (a.nvid < 1 || a.k < 1 || !a.vid || !a.live_k || !a.dead_k || !a.next_w || !a.score0 || !a.h0 || !a.c0 || !a.hp || !a.cp || !a.dp ||
        !a.ticket || !a.step) return hipErrorInvalidValue;
    size_t n = (size_t)a.nvid * a.k * 3 * a.D;
    for (int q = 0; q < 6; ++q) if (a.zero[q] && a.zero_n[q] > n) n = a.zero_n[q];
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(beam_init_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace stattn

#ifdef STATTN_PROBES
extern "C" int stattn_probe_beam(long long* out8) {      // first call (out8 = null): arm; later: read the eight stamps
    static long long* d = nullptr;
    if (!d) {
        if (hipMalloc(&d, 8 * sizeof(long long)) != hipSuccess) return -1;
        (void)hipMemset(d, 0, 8 * sizeof(long long));
        if (hipMemcpyToSymbol(HIP_SYMBOL(stattn::bm_probe), &d, sizeof d) != hipSuccess) return -2;
    }
    if (out8) { (void)hipDeviceSynchronize(); if (hipMemcpy(out8, d, 8 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return -3; }
    return 0;
}
#endif
