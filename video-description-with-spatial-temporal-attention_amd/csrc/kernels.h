// Internal launcher declarations shared by the .hip translation units and api.cpp.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, fp32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stattn {

// ----------------------------------------------------------------------------
// LDS-tiled fp32 MFMA GEMM (gemm.hip):  C = epi(alpha * op(A).op(B))
//   epi(v)[m,n] = mul[m,n] * act(v + bias[n] + add[m,n] + rowadd[m / rowgroup, n]) (+ C[m,n] if accumulate)
// op(A) = A [M,K] (lda) or, transA, A given as [K,M]; op(B) = B [K,N] (ldb) or, transB, B given as [N,K].
// Constraints: N % 64 == 0; K % 4 == 0; transA additionally M % 4 == 0.
// ----------------------------------------------------------------------------
struct GemmArgs {
    const float* A; const float* B; float* C;
    int lda, ldb, ldc;
    int M, N, K;
    float alpha;
    const float* bias;                 // [N] or null
    const float* add; int ldadd;       // [M,N] or null
    const float* rowadd; int ldrow; int rowgroup;  // [M/rowgroup, N] or null
    const float* mul; int ldmul;       // [M,N] elementwise multiplier applied after act, or null
    int act;                           // 0 none, 1 tanh
    int accumulate;                    // C += result
};
void gemm_defaults(GemmArgs& g);
hipError_t launch_gemm(hipStream_t s, const GemmArgs& g, bool transA, bool transB);

// ----------------------------------------------------------------------------
// Register-streaming "skinny" grouped GEMM (skinny.hip) for M <= a few hundred rows:
// the weight matrix streams HBM/L2 -> VGPR exactly once per 16*mt rows, no LDS staging.
// One launch computes several independent output segments; each segment sums up to three
// (A,B) pairs that share the output tile:   C = epi(sum_p A_p[M,K_p] . B_p[K_p,N])
//   epi(v)[m,n] = scale * act(v + bias[n] + bias2[n] + add[m,n]) * (mul ? mul[m,n] : 1)
// N % 64 == 0, K_p % 16 == 0, rows of A_p 16-byte aligned.
// ----------------------------------------------------------------------------
struct SkPair { const float* A; const float* B; int lda, ldb, K; };
struct SkSeg {
    SkPair p[3]; int npairs;
    float* C; int ldc; int N;          // N = number of output columns of this segment
    const float* bias; const float* bias2;
    const float* add; int ldadd;
    const float* mul; int ldmul;
    float scale; int act;
};
struct SkArgs { SkSeg seg[6]; int nseg; int M; };
void skinny_seg_defaults(SkSeg& s);
hipError_t launch_skinny(hipStream_t s, const SkArgs& a);

// LSTM cell with the ctx.Wc (+ h.U (+ emb.W)) GEMM fused in front of the gate epilogue
// (model_attention.py:437-457).  Column tiles are gate-interleaved (16 units x 4 gates).
struct LstmArgs {
    SkPair p[3]; int npairs;           // pairs accumulate into preact; B matrices are [K, 4D]
    const float* pre_add; int ldpre;   // [M,4D] added to preact (x_ = emb.W + b, and/or h.U), or null
    const float* bias;                 // [4D] or null (decode mode: decoder_b)
    const float* dp; int lddp;         // [M,3D] dropout multipliers on i,f,o pre-activations
    const float* mask;                 // [M] or null (= all ones, one_step mode)
    const float* h_prev; const float* c_prev;  // [M,D]
    float* h_out; float* c_out;        // [M,D]
    float* gates;                      // [M,4D] post-activation i,f,o,g (for backward) or null
    const float* d1; int ldd1;         // [M,D] readout dropout multiplier, or null -> d1_scalar
    float d1_scalar;
    float* hd_out;                     // [M,D] = h_out * d1  (readout input, :684-685) or null
    int M, D;
};
hipError_t launch_lstm(hipStream_t s, const LstmArgs& a);

// ----------------------------------------------------------------------------
// attention kernels (attn.hip)
// ----------------------------------------------------------------------------
struct SpatialArgs {
    // per-video projected context, video index = vid[b] (null => b)
    const float* PL; const float* L; const float* LW;   // [nvid,T,K,D]; LW null in lt_mode 0
    const float* PG; const float* PM;                   // [nvid,T,D]
    const int* vid;
    // state projections of this step: sproj[b] = [sl | sg | sm | slt], row stride ldsp
    const float* sproj; int ldsp;
    const float* Ul; const float* cl;    // [D], [1]
    const float* Ug; const float* cg;
    const float* Um; const float* cm;
    const float* Ult; const float* clt; const float* blt;   // lt_mode 1 only
    float* alphal;     // [M,T,K]
    float* CL;         // [M,T,D]
    float* eg; float* em; float* elt;   // [M,T] raw scores (elt only in lt_mode 1)
    int M, T, K, D;
};
hipError_t launch_spatial(hipStream_t s, const SpatialArgs& a);

// elt[r] = dot(P[r,:], U) + c   (lt_mode 0, after the CL.Wclt GEMM with fused tanh)
hipError_t launch_rowdot(hipStream_t s, const float* P, int ldp, const float* U, const float* c,
                         float* out, int rows, int D);

struct TemporalArgs {
    const float* eg; const float* em; const float* elt;   // [M,T]
    const float* G; const float* Mo;                      // [nvid,T,D]
    const int* vid;
    const float* CL;                                      // [M,T,D]
    const float* h_prev; const float* W_sel; const float* b_sel;  // selector (null W_sel => none)
    float* alphag; float* alpham; float* alphalt;         // [M,T]
    float* csum;       // [M,D] cg+cm+clt before the gate (for backward), or null
    float* sel;        // [M] or null
    float* ctx;        // [M,D]
    int M, T, D;
};
hipError_t launch_temporal(hipStream_t s, const TemporalArgs& a);

// ----------------------------------------------------------------------------
// small kernels (misc.hip)
// ----------------------------------------------------------------------------
hipError_t launch_fill(hipStream_t s, float* p, float v, size_t n);
hipError_t launch_iota(hipStream_t s, int* p, int n, int mul);   // p[i] = i * mul
// mean[b,:] = sum_t G[b,t,:] / sum_t mask[b,t]   (model_attention.py:618, 649 / 739, 766)
hipError_t launch_ctx_mean(hipStream_t s, const float* G, const float* mask, float* mean, int B, int T, int D);
// emb[r,:] = (x[r] < 0) ? 0 : Wemb[x[r],:] ; shift>0: row r reads x[r - shift] and rows < shift are zero
hipError_t launch_embed(hipStream_t s, const int64_t* x, const float* Wemb, float* emb, int rows, int E, int V, int shift);
// row softmax over V (ld = ldl) + optional NLL:  nll[r] = -log(p[r, x[r]] + 1e-8)
hipError_t launch_softmax_nll(hipStream_t s, const float* logits, int ldl, float* probs, int ldp,
                              const int64_t* x, float* nll, int64_t* argmax, int rows, int V);
// cost[b] = sum_t mask[t,b] * nll[t,b]
hipError_t launch_cost(hipStream_t s, const float* nll, const float* mask, float* cost, int t, int m);
// Bernoulli(0.5) in {0,1} from a counter-based hash
hipError_t launch_bernoulli(hipStream_t s, float* p, size_t n, uint64_t seed, uint64_t stream_id);
// uniform in [-1, 1)
hipError_t launch_uniform(hipStream_t s, float* p, size_t n, uint64_t seed, uint64_t stream_id);

}  // namespace stattn
