// Device-side pieces of the beam bookkeeping shared by beam.hip (beam_update_kernel) and attn.hip (the single-round-trip
// attention kernel that carries the previous word's update as an extra workgroup: one-hypothesis decode).  See beam.hip.
#pragma once
#include "kernels.h"
#include "devmath.h"

#ifndef STATTN_BEAM_PF
#define STATTN_BEAM_PF 4
#endif
#ifndef BM_STAMP
#define BM_STAMP(i) do {} while (0)
#endif

namespace stattn {
namespace {

constexpr int KB = 8;   // maximum beam width

// n / d for 0 <= n < 2^22 and a workgroup-uniform d >= 1, through the float reciprocal (rd = 1.0f / d) and one correction step:
// exact, six instructions instead of the ~40 of an integer division.  beam_update is ONE workgroup per video -- its 16 waves share
// the issue slots of a single CU, so its time is its instruction count (tools/beam_probe.py), and the flat (row, tile, rank) /
// (row, column) index splits were most of it.
__device__ __forceinline__ int fdiv(int n, int d, float rd) {
    int q = (int)((float)n * rd);
    q -= (q * d > n);
    q += ((q + 1) * d <= n);
    return q;
}

// ordering of candidates: cost ascending, ties by the lower flat index (what a stable argsort of the flat cost
// array yields, :921-923)
__device__ __forceinline__ bool cand_less(float c0, int i0, float c1, int i1) { return c0 < c1 || (c0 == c1 && i0 < i1); }

// (cost, flat index) as ONE 64-bit key whose unsigned order is cand_less's order: the cost's bits made monotone (sign flip),
// the index below them -- a wave-wide arg-min is then a min over one value (two 32-bit shuffles per step instead of three, no
// owner lane to carry: indices are unique, the winner is whoever holds the minimum)
__device__ __forceinline__ unsigned long long cand_key(float c, int i) {
    unsigned u = __float_as_uint(c);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)i;
}
__device__ __forceinline__ float key_cost(unsigned long long key) {
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long wave_min_key(unsigned long long k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(k, o, 64); k = other < k ? other : k; }
    return k;
}

// The same selection with ONE workgroup barrier instead of 2 n: every wave first selects its own n best (n rounds of a
// wave-wide arg-min, no barrier), then wave 0 merges the <= 16 n wave winners.  s_c / s_i: [16 * KB] each.
__device__ __forceinline__ void block_select2(float (&lc)[KB], int (&li)[KB], int n, float* s_c, int* s_i, float* res_c, int* res_i) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nwv = (int)blockDim.x >> 6;
    for (int r = 0; r < n; ++r) {
        const unsigned long long mine = cand_key(lc[0], li[0]), best = wave_min_key(mine);
        if (lane == 0) { s_c[w * KB + r] = key_cost(best); s_i[w * KB + r] = (int)(unsigned)best; }
        if (mine == best) {                                    // (several lanes only when every list is exhausted: sentinels)
#pragma unroll
            for (int i = 0; i < KB - 1; ++i) { lc[i] = lc[i + 1]; li[i] = li[i + 1]; }
            lc[KB - 1] = INFINITY; li[KB - 1] = 0x7fffffff;
        }
    }
    __syncthreads();
    if (w == 0) {
        unsigned long long mk[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;                       // candidate e = (wave e / n, rank e % n)
            const bool in = e < nwv * n;
            mk[u] = in ? cand_key(s_c[(e / n) * KB + e % n], s_i[(e / n) * KB + e % n]) : cand_key(INFINITY, 0x7fffffff);
        }
        if (mk[1] < mk[0]) { const unsigned long long t = mk[0]; mk[0] = mk[1]; mk[1] = t; }
        for (int r = 0; r < n; ++r) {
            const unsigned long long best = wave_min_key(mk[0]);
            if (lane == 0) { res_c[r] = key_cost(best); res_i[r] = (int)(unsigned)best; }
            if (mk[0] == best) { mk[0] = mk[1]; mk[1] = cand_key(INFINITY, 0x7fffffff); }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void list_insert(float (&lc)[KB], int (&li)[KB], float c, int flat) {
    if (cand_less(c, flat, lc[KB - 1], li[KB - 1])) {
        lc[KB - 1] = c; li[KB - 1] = flat;
#pragma unroll
        for (int i = KB - 1; i > 0; --i) {            // one bubble pass keeps the list sorted
            const bool sw = cand_less(lc[i], li[i], lc[i - 1], li[i - 1]);
            if (sw) { const float tc = lc[i]; lc[i] = lc[i - 1]; lc[i - 1] = tc; const int ti = li[i]; li[i] = li[i - 1]; li[i - 1] = ti; }
        }
    }
}

// Stage 2 + bookkeeping, one workgroup per video: merge the nsplit * KB slice winners (<= 256) into the selection
// (:923-928), then build the new hypotheses, retire the finished ones and gather the states (:939-985).
// 256 threads per video, or 1024 on the small-batch path (launch_beam_update): a single video's selection is one workgroup's
// serial work -- lse, candidate scan, selection, gathers -- and at k = 5 it was half of the decoded word
//
// Row workgroups (BeamArgs::rw_cost set: beams of 2 .. 8 hypotheses on the small path, 1024 threads): k workgroups per video instead
// of one.  Workgroup (v, j) forms the log-sum-exp of live row j and the nsel best candidates of THAT row -- the nsel best overall are
// among the rows' nsel best -- and leaves them in rw_cost / rw_idx; the last of a video's k workgroups to arrive (a ticket per video)
// merges the <= 64 row winners and goes on with the bookkeeping and the gathers.  The arithmetic is the single workgroup's, value for
// value: row j's tiles are walked by waves 0 .. wpr - 1 exactly as waves j, j + live, ... walk them there, the partials are combined
// in the same order, a candidate's cost is the same expression; (cost, index) is a total order, so the selection is the same set in
// the same order.  No workgroup waits for another (last arriver, no spin), so the launch cannot deadlock.
__device__ __forceinline__ void beam_update_body(const BeamArgs& a, int nsplit, const float* __restrict__ pcost,
                                                 const int* __restrict__ pidx, const int wg, const int nwg) {
    const bool rw = a.rw_cost != nullptr;
    const int v = rw ? wg / a.k : wg, jrow = rw ? wg - v * a.k : 0;
    __shared__ int s_slot[KB], s_fin[KB], s_ti[KB], s_wi[KB];
    __shared__ int s_n, s_ended, s_rows;
    __shared__ float s_c2[16 * KB];
    __shared__ int s_i2[16 * KB];
    __shared__ float res_c[KB];
    __shared__ int res_i[KB];
    const int tid = threadIdx.x, NT = blockDim.x;
    BM_STAMP(0);
    const int k = a.k, D = a.D, L = a.maxlen, V = a.V, step = *a.step;
    const int live0 = a.live_k[v], dead0 = a.dead_k[v];
    const int nsel = live0 > 0 ? k - dead0 : 0;                // how many candidates survive (:923)
    BM_STAMP(1);
    // the word counter: every workgroup read *a.step when it started; the LAST one to get here (a ticket) knows all of them did, so
    // it may write step + 1 for the next word's kernels (was a one-thread launch of its own)
    auto advance_step = [&]() {
        __syncthreads();
        if (tid == 0) {
            if (nwg == 1) {                            // one video, one workgroup: the last by construction
                __hip_atomic_store(a.step, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // (no fence: the ticket orders READS of *a.step before its one write -- every workgroup has consumed the value it loaded
                // long before its add is issued -- and the write only has to be seen by the next launch.  A __threadfence here was a
                // whole-L2 write-back per update workgroup, on the last arriver's critical path once a video has k of them.)
                if (__hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
                    *a.ticket = 0;
                    __hip_atomic_store(a.step, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (kernels.h BeamArgs::step: no reader inside this launch)
                }
            }
        }
    };
    if (rw) {
        __shared__ int s_last;
        __shared__ float s_m1[16], s_s1[16];
        if (nsel > 0 && jrow < live0) {
            const int live = live0, nt = a.ntile, lane = tid & 63, w = tid >> 6, nwv = NT >> 6;
            const int C = nt * nsel;
            const float r_nsel = 1.0f / (float)nsel;
            const float* recs = a.stats + (size_t)(v * k + jrow) * nt * PN_STATS_REC;
            constexpr int PFR = 6;                             // (752 tiles x 8 candidates over 1024 threads)
            float pv[PFR]; int pi[PFR];
#pragma unroll
            for (int u = 0; u < PFR; ++u) {
                const int e = u * NT + tid;
                pv[u] = -INFINITY; pi[u] = 0x7fffffff;
                if (e < C) {
                    const int t = fdiv(e, nsel, r_nsel), i = e - t * nsel;
                    const float* rec = recs + (size_t)t * PN_STATS_REC;
                    pv[u] = rec[2 + i];
                    pi[u] = jrow * V + reinterpret_cast<const int*>(rec)[2 + PN_STATS_KB + i];
                }
            }
            const float hyp = a.hyp_score[v * k + jrow];
            const int wpr = nwv / live;                        // waves per row of the single-workgroup pass (live <= 8 < 16 waves)
            float rm = -INFINITY, rs = 0.f;
            for (int t = w < wpr ? w * 64 + lane : nt; t < nt; t += wpr * 64) {
                const float tm = recs[(size_t)t * PN_STATS_REC], ts = recs[(size_t)t * PN_STATS_REC + 1];
                if (tm > -INFINITY) {
                    const float nm = fmaxf(rm, tm);
                    rs = rs * __expf(rm - nm) + ts * __expf(tm - nm);
                    rm = nm;
                }
            }
            const float wm = wave_max(rm);
            const float ws = wave_sum(rm > -INFINITY ? rs * __expf(rm - wm) : 0.f);
            if (lane == 0) { s_m1[w] = wm; s_s1[w] = ws; }
            __syncthreads();
            float m = -INFINITY;
            for (int q = 0; q < nwv; ++q) m = fmaxf(m, s_m1[q]);
            float ssum = 0.f;
            for (int q = 0; q < nwv; ++q) if (s_m1[q] > -INFINITY) ssum += s_s1[q] * __expf(s_m1[q] - m);
            const float lse = m + logf(ssum);
            const bool one = live == 1;                        // (one live row: its log-sum-exp is added after the selection, as there)
            const float base = one ? hyp : hyp + lse;
            float lc[KB]; int li[KB];
#pragma unroll
            for (int i = 0; i < KB; ++i) { lc[i] = INFINITY; li[i] = 0x7fffffff; }
#pragma unroll
            for (int u = 0; u < PFR; ++u)
                if (pv[u] > -INFINITY) list_insert(lc, li, base - pv[u], pi[u]);
            for (int e = PFR * NT + tid; e < C; e += NT) {     // (vocabularies beyond 6 candidates per thread)
                const int t = fdiv(e, nsel, r_nsel), i = e - t * nsel;
                const float* rec = recs + (size_t)t * PN_STATS_REC;
                const float val = rec[2 + i];
                if (val > -INFINITY) list_insert(lc, li, base - val, jrow * V + reinterpret_cast<const int*>(rec)[2 + PN_STATS_KB + i]);
            }
            block_select2(lc, li, nsel, s_c2, s_i2, res_c, res_i);
            if (tid < nsel) {
                float c = res_c[tid];
                if (one && res_i[tid] != 0x7fffffff) c += lse;
                // (agent-scope stores: written through to where every XCD reads them, no L2 write-back of the whole cache as a release fence does)
                __hip_atomic_store(a.rw_cost + (size_t)(v * k + jrow) * KB + tid, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.rw_idx + (size_t)(v * k + jrow) * KB + tid, res_i[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);                 // both stores acknowledged before the barrier in front of the ticket
            }
        }
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(a.rw_ticket + v, 1) == k - 1;
        __syncthreads();
        if (!s_last) { advance_step(); return; }
        if (tid == 0) a.rw_ticket[v] = 0;
        if (nsel > 0 && tid < 64) {                            // merge of the <= live * nsel <= 64 row winners, one per lane
            const int nrow = live0 * nsel;
            const bool in = tid < nrow;
            const int row = in ? tid / nsel : 0, r = in ? tid - row * nsel : 0;
            const float c = in ? __hip_atomic_load(a.rw_cost + (size_t)(v * k + row) * KB + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFINITY;
            const int ix = in ? __hip_atomic_load(a.rw_idx + (size_t)(v * k + row) * KB + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
            unsigned long long key = cand_key(c, ix);
            for (int q = 0; q < nsel; ++q) {
                const unsigned long long best = wave_min_key(key);
                if (tid == 0) { res_c[q] = key_cost(best); res_i[q] = (int)(unsigned)best; }
                if (key == best) key = cand_key(INFINITY, 0x7fffffff);
            }
        }
    } else if (nsel > 0 && a.stats) {
        // Small-batch decode: no probabilities were materialised.  The logits launch left, per (row, vocabulary tile),
        // the tile max, sum exp(v - max) and its best values; here: log-sum-exp per live row, then
        // cost = hyp_score - log p = hyp_score + lse - v for the tiles' candidates, merged like the slice winners above.
        __shared__ float s_lse[KB];
        __shared__ float s_m[16][KB], s_s[16][KB];
        const int live = live0, nt = a.ntile, lane = tid & 63, w = tid >> 6;
        // The first four candidates of every thread (all of them up to 4096: every greedy / beam-5 single-video step) are
        // requested BEFORE the log-sum-exp pass: their addresses do not depend on it, only their cost does, so the two passes
        // over the records are one memory round trip instead of two dependent ones.
        const int per = nt * nsel;                             // the nsel best of every tile cover the nsel best overall
        const int C = live * per;
        const float r_per = 1.0f / (float)per, r_nsel = 1.0f / (float)nsel;
        constexpr int PF = STATTN_BEAM_PF;                     // candidates per thread requested up front (-DSTATTN_BEAM_PF=12 / 20 measured: beam-5 word 57.2-57.5 / 57.1-58.9 us against 57.6-58.0, greedy 35.3-35.8 / 35.8-36.4 against 35.0-35.4: within noise, not kept)
        float pv[PF]; int pi[PF], pj[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = u * NT + tid;
            pv[u] = -INFINITY; pi[u] = 0x7fffffff; pj[u] = 0;
            if (e < C) {
                const int j = fdiv(e, per, r_per), rem = e - j * per, t = fdiv(rem, nsel, r_nsel), i = rem - t * nsel;
                const float* rec = a.stats + ((size_t)(v * k + j) * nt + t) * PN_STATS_REC;
                pv[u] = rec[2 + i];                                            // (stochastic: the tile's best PERTURBED value)
                pi[u] = j * V + reinterpret_cast<const int*>(rec)[2 + PN_STATS_KB + i];
                pj[u] = j;
            }
        }
        __shared__ float s_hyp[KB];
        if (tid < live) s_hyp[tid] = a.hyp_score[v * k + tid];
        // Rows in parallel when there are waves enough (1024 threads): wave w takes row w % live and every (nwv / live)-th group
        // of 64 tiles of it -- one load latency for all rows instead of one per row; else row after row over all threads.
        const int nwv_ = NT >> 6, wpr = nwv_ / live;              // waves per row (0: fewer waves than rows)
        for (int j = wpr ? w % live : 0; j < live; j += wpr ? live : 1) {
            const float* rec = a.stats + (size_t)(v * k + j) * nt * PN_STATS_REC;
            float rm = -INFINITY, rs = 0.f;
            const int t0 = wpr ? (w / live) * 64 + lane : tid, tstep = wpr ? wpr * 64 : NT;
            for (int t = (wpr && w / live >= wpr) ? nt : t0; t < nt; t += tstep) {
                const float tm = rec[(size_t)t * PN_STATS_REC], ts = rec[(size_t)t * PN_STATS_REC + 1];
                if (tm > -INFINITY) {
                    const float nm = fmaxf(rm, tm);
                    rs = rs * __expf(rm - nm) + ts * __expf(tm - nm);      // (exp(-inf) = 0 on the first tile)
                    rm = nm;
                }
            }
            const float wm = wave_max(rm);
            const float ws = wave_sum(rm > -INFINITY ? rs * __expf(rm - wm) : 0.f);
            if (lane == 0) { s_m[w][j] = wm; s_s[w][j] = ws; }
            if (wpr && lane == 0)                      // the other rows' slots of this wave: neutral elements
                for (int o = 0; o < live; ++o) if (o != j) { s_m[w][o] = -INFINITY; s_s[w][o] = 0.f; }
        }
        // One live row: its log-sum-exp shifts every candidate's cost by the same amount, so the selection does not wait for it --
        // the waves' partials are combined after the selection (whose barriers publish them) and added to the winners' costs.
        const bool one = live == 1;
        const float hyp0 = a.hyp_score[v * k];
        auto row_lse = [&](int j) {
            const int nwv = NT >> 6;
            float m = -INFINITY;
            for (int q = 0; q < nwv; ++q) m = fmaxf(m, s_m[q][j]);
            float ssum = 0.f;
            for (int q = 0; q < nwv; ++q) if (s_m[q][j] > -INFINITY) ssum += s_s[q][j] * __expf(s_m[q][j] - m);
            return m + logf(ssum);
        };
        if (!one) {
            __syncthreads();
            if (tid < live) s_lse[tid] = row_lse(tid);
            __syncthreads();
        }
        BM_STAMP(2);
        float lc[KB]; int li[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) { lc[i] = INFINITY; li[i] = 0x7fffffff; }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (pv[u] > -INFINITY) list_insert(lc, li, (a.stochastic ? 0.f : (one ? hyp0 : s_hyp[pj[u]] + s_lse[pj[u]])) - pv[u], pi[u]);
        // the rest of the flat index space over (live row, tile, rank), four candidates' loads in flight per thread before they
        // are inserted (row by row and one at a time, the 37 inserts of a thread at k = 5 were 37 exposed L2 latencies)
        for (int c0 = PF * NT; c0 < C; c0 += 4 * NT) {
            // (value AND index of the four candidates requested together, unconditionally -- the last candidate is re-read past the
            // end: with the index load inside `if (val > -inf)` the four were eight dependent round trips)
            float cv[4], val[4]; int ci[4], ix[4], jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(c0 + u * NT + tid, C - 1);
                const int j = fdiv(e, per, r_per), rem = e - j * per, t = fdiv(rem, nsel, r_nsel), i = rem - t * nsel;
                const float* rec = a.stats + ((size_t)(v * k + j) * nt + t) * PN_STATS_REC;
                val[u] = rec[2 + i];                                           // (stochastic: the tile's best PERTURBED value)
                ix[u] = reinterpret_cast<const int*>(rec)[2 + PN_STATS_KB + i];
                jj[u] = j;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cv[u] = INFINITY; ci[u] = 0x7fffffff;
                if (c0 + u * NT + tid < C && val[u] > -INFINITY) {
                    const int j = jj[u];
                    const float base = a.stochastic ? 0.f : (one ? hyp0 : s_hyp[j] + s_lse[j]);
                    cv[u] = base - val[u]; ci[u] = j * V + ix[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ci[u] != 0x7fffffff) list_insert(lc, li, cv[u], ci[u]);
        }
        BM_STAMP(3);
        block_select2(lc, li, nsel, s_c2, s_i2, res_c, res_i);
        if (one && tid == 0) {
            const float lse = row_lse(0);
            s_lse[0] = lse;
            if (!a.stochastic)
                for (int r = 0; r < nsel; ++r) if (res_i[r] != 0x7fffffff) res_c[r] += lse;
        }
        BM_STAMP(4);
        if (a.stochastic && tid == 0 && res_i[0] != 0x7fffffff) {
            // the draw is word res_i[0]; gen_sample's stochastic "score" is the running SUM of the drawn words'
            // probabilities (model_attention.py:916): p = exp(v - lse) with v the unperturbed logit kept by the tile
            const int col = res_i[0] % V;
            const float* rec = a.stats + ((size_t)(v * k) * nt + col / a.tile_cols) * PN_STATS_REC;
            res_c[0] = a.hyp_score[v * k] + __expf(rec[3] - s_lse[0]);
        }
    } else if (nsel > 0) {                                     // (uniform over the workgroup)
        float lc[KB]; int li[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) { lc[i] = INFINITY; li[i] = 0x7fffffff; }
        if (tid < nsplit * KB) { lc[0] = pcost[(size_t)v * nsplit * KB + tid]; li[0] = pidx[(size_t)v * nsplit * KB + tid]; }
        block_select2(lc, li, nsel, s_c2, s_i2, res_c, res_i);
    }
    __syncthreads();
    if (tid < 64) {
        // Bookkeeping of the nsel <= 8 winners, one lane each (it was a serial loop of thread 0: 10 us at k = 5).  Fewer
        // candidates than slots (live * V < k - dead): argsort()[:n] is simply shorter (:923) -- n = the leading run of
        // non-sentinel entries; never index with the sentinel.
        const bool valid = tid < nsel && res_i[tid < KB ? tid : 0] != 0x7fffffff;
        const unsigned long long bv = __ballot(valid);
        const int n = __ffsll((long long)~bv) - 1;
        const bool act = tid < n;
        const int flat = act ? res_i[tid] : 1;
        const int ti = flat / V, wi = flat % V;                // trans_indices = ranks_flat // voc_size, word_indices = % (:926-927)
        const bool fin = act && wi == 0;                       // <eos>: the hypothesis dies (:958-962)
        const unsigned long long bf = __ballot(fin), bl = __ballot(act && !fin), below = (1ull << tid) - 1ull;
        if (act) {
            const float cost = res_c[tid];
            s_ti[tid] = ti; s_wi[tid] = wi; s_fin[tid] = fin ? 1 : 0;
            if (fin) {
                const int slot = dead0 + __popcll(bf & below);
                s_slot[tid] = slot;
                a.fin_score[v * k + slot] = cost;
                a.fin_len[v * k + slot] = step + 1;
            } else {                                           // stays live (:963-970)
                const int slot = __popcll(bl & below);
                s_slot[tid] = slot;
                a.hyp_score_out[v * k + slot] = cost;
                a.next_w[v * k + slot] = wi;
            }
        }
        if (tid == 0) {
            const int dead = dead0 + __popcll(bf), nl = __popcll(bl);
            s_n = n; s_ended = 0; s_rows = live0;
            if (nsel > 0) {
                a.dead_k[v] = dead;
                const int live = (nl < 1 || dead >= k) ? 0 : nl;   // :974-977
                a.live_k[v] = live;
                if (live == 0) { s_ended = 1; if (a.end_rows) a.end_rows[v] = s_rows; }
            }
        }
    }
    __syncthreads();
    BM_STAMP(5);
    const int n = s_n;
    if (a.rowmap && tid < k) {      // parent of the hypothesis that now sits in row tid (unused rows: themselves)
        int src = tid;
        for (int r = 0; r < n; ++r) if (!s_fin[r] && s_slot[r] == tid) src = s_ti[r];
        a.rowmap[v * k + tid] = v * k + src;
    }
    // Copies of the n surviving candidates: tokens, then -- for the ones that stay live -- the parent's state, the next
    // step's state projections, the packed h and the embedding of the chosen word.  One flat index space per field over ALL
    // candidates: the loads of every row are in flight together.  (A loop over the candidates around per-row loops was a
    // chain of n x 5 dependent read -> write round trips: 50 us of a 93 us word at k = 5.)
    const float r_step = 1.0f / (float)(step > 0 ? step : 1), r_D = 1.0f / (float)D;
    for (int i = tid; i < n * step; i += NT) {
        const int r = fdiv(i, step, r_step), j = i - r * step;
        const int* __restrict__ src = a.tok_in + (size_t)(v * k + s_ti[r]) * L;
        int* __restrict__ dst = (s_fin[r] ? a.fin_tok : a.tok_out) + (size_t)(v * k + s_slot[r]) * L;
        dst[j] = src[j];
    }
    if (tid < n) ((s_fin[tid] ? a.fin_tok : a.tok_out) + (size_t)(v * k + s_slot[tid]) * L)[step] = s_wi[tid];
    // (every gather below: four independent loads per thread in flight, then their stores)
    if (a.h_step != a.h_next)                                  // (one hypothesis per video: the LSTM wrote the states in place)
    for (int i0 = tid; i0 < n * D; i0 += 4 * NT) {             // gather the state of the parent hypothesis (:943-945)
        float hv[4], cv[4]; size_t dO[4]; int rr[4], dd[4]; bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT;
            on[u] = i < n * D;
            const int r = on[u] ? fdiv(i, D, r_D) : 0, d = on[u] ? i - r * D : 0;
            on[u] = on[u] && !s_fin[r];
            rr[u] = r; dd[u] = d;
            const size_t so = (size_t)(v * k + s_ti[r]) * D + d;
            dO[u] = (size_t)(v * k + s_slot[r]) * D + d;
            hv[u] = on[u] ? a.h_step[so] : 0.f; cv[u] = on[u] ? a.c_step[so] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!on[u]) continue;
            a.h_next[dO[u]] = hv[u]; a.c_next[dO[u]] = cv[u];
            if (a.h_next_pk) a.h_next_pk[pn_pack_offset(v * k + s_slot[rr[u]], dd[u], D >> 4)] = hv[u];
        }
    }
    if (a.proj_next) {   // the next step's state projections travel with the hypothesis (linear in h: gathered, not recomputed)
        const int np4 = a.nproj >> 2;
        const float r_np4 = 1.0f / (float)np4;
        for (int i0 = tid; i0 < n * np4; i0 += 4 * NT) {
            float4 pv[4]; float4* dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NT;
                dst[u] = nullptr;
                if (i < n * np4) {
                    const int r = fdiv(i, np4, r_np4), d4 = i - r * np4;
                    if (!s_fin[r]) {
                        pv[u] = reinterpret_cast<const float4*>(a.proj_step + (size_t)(v * k + s_ti[r]) * a.nproj)[d4];
                        dst[u] = reinterpret_cast<float4*>(a.proj_next + (size_t)(v * k + s_slot[r]) * a.nproj) + d4;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (dst[u]) *dst[u] = pv[u];
        }
    }
    // the embedding of the word just chosen = the input of the hypothesis' next step (:803-804): written here, so the
    // word loop needs no separate lookup launch
    if (a.emb_next) {
        const float r_E = 1.0f / (float)a.E;
        for (int i0 = tid; i0 < n * a.E; i0 += 4 * NT) {
            float xv[4]; int rr[4], ee[4]; bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NT;
                on[u] = i < n * a.E;
                const int r = on[u] ? fdiv(i, a.E, r_E) : 0, e = on[u] ? i - r * a.E : 0;
                on[u] = on[u] && !s_fin[r];
                rr[u] = r; ee[u] = e;
                xv[u] = on[u] ? a.Wemb[(size_t)s_wi[r] * a.E + e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!on[u]) continue;
                a.emb_next[(size_t)(v * k + s_slot[rr[u]]) * a.E + ee[u]] = xv[u];
                if (a.emb_next_pk) a.emb_next_pk[pn_pack_offset(v * k + s_slot[rr[u]], ee[u], a.E >> 4)] = xv[u];
            }
        }
    }
    // the video's loop ends with this word (:974-977): gen_sample returns f_next's state outputs of this very call,
    // one row per hypothesis that was live going in -- kept aside, later words overwrite h_step
    if (s_ended && a.end_h) {
        const size_t base = (size_t)v * k * D;
        for (int i = tid; i < s_rows * D; i += NT) { a.end_h[base + i] = a.h_step[base + i]; a.end_c[base + i] = a.c_step[base + i]; }
    }
    BM_STAMP(6);
    advance_step();
    BM_STAMP(7);
}

}  // namespace
}  // namespace stattn
