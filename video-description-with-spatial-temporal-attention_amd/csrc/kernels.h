// Internal launcher declarations shared by the .hip translation units and api.cpp.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, fp32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "switches.h"

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
    const float* bias2;                // second [N] bias (two layers summed into one pre-activation), or null
    // optional second operand pair, K-concatenated: C = A.B + A2.B2 (NN or NT, fp32-MFMA kernel only; K % 32 == 0, K2 % 32 == 0).
    // The readout's a = tanh(hd.Wl1 + ctx.Wl2 + ...) (model_attention.py:687-699) is one launch this way.
    const float* A2; const float* B2; int lda2, ldb2, K2;
    const float* add; int ldadd;       // [M,N] or null
    const float* rowadd; int ldrow; int rowgroup;  // [M/rowgroup, N] or null
    const float* mul; int ldmul;       // [M,N] elementwise multiplier applied after act, or null
    int act;                           // 0 none, 1 tanh
    int accumulate;                    // C += result
    float* Cact; int ldcact;           // optional second output: the activation BEFORE `mul` (tanh(z) for backward)
    // deterministic split-K (weight-gradient shapes: small MxN, huge K): when `ws` is given and the tile grid
    // would leave most CUs idle, K is cut into slices that write partial tiles to ws, and a second kernel
    // sums them in a fixed order (with the fused epilogue, when there is one).
    float* ws; size_t ws_floats;
    int kbeg, kend, kslices;           // internal
    int xcd_remap;                     // internal: XCD-aware tile order on/off
    int split;                         // 1: run on the bf16 matrix cores with three-term operands (gemm_split.hip)
    long long* clk;                    // internal: clock probe slot {cycles0, wall0, cycles1, wall1} written by block 0 (-DSTATTN_PROBES builds only)
};
void gemm_defaults(GemmArgs& g);
hipError_t launch_gemm(hipStream_t s, const GemmArgs& g, bool transA, bool transB);
// up to eight independent problems (64x64 tiles, no split-K; all NN, all with A given as [K,M], or all with B given as
// [N,K]) in ONE launch: small
// problems ride in the tail of a large one instead of under-filling the chip on their own
constexpr int GEMM_GROUP_MAX = 8;
struct GemmGroup { GemmArgs g[GEMM_GROUP_MAX]; int tile_start[GEMM_GROUP_MAX + 1]; int n; };
hipError_t launch_gemm_group(hipStream_t s, const GemmArgs* gs, int n, bool transA = false, bool transB = false);
void gemm_clock_dump();   // tools only (STATTN_GEMM_CLK=1)
hipError_t launch_splitk_reduce(hipStream_t s, const float* ws, float* C, int ldc, int M, int N, int slices, float alpha, int accumulate);
hipError_t launch_splitk_reduce_epilogue(hipStream_t s, const GemmArgs& g, int slices);     // the same with g's fused epilogue

// ----------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores (gemm_split.hip): same problem description and epilogue as above, every operand
// split exactly into three bf16 terms, six MFMA products per k-block, fp32 accumulation.  launch_gemm /
// launch_gemm_group route here when GemmArgs::split is set and gemm_split_supported() (N % 128 == 0, 16-byte
// aligned k-contiguous operands); otherwise they run the fp32-MFMA kernel.
// ----------------------------------------------------------------------------
bool gemm_split_supported(const GemmArgs& g, bool transA, bool transB);
hipError_t launch_gemm_split(hipStream_t s, const GemmArgs& g, bool transA, bool transB);
hipError_t launch_gemm_split_group(hipStream_t s, const GemmArgs* gs, int n, bool transA, bool transB);

// ----------------------------------------------------------------------------
// bf16-MFMA GEMM (gemm_bf16.hip), precision = bf16 handles only:  C = epi(A[M,K] . B[N,K]^T), both operands bf16
// and k-contiguous, fp32 accumulation; epi = act(v + bias[n] + add[m,n] + rowadd[m / rowgroup, n]) written as fp32
// (C) and/or bf16 (Cb).  N % 64 == 0, K % 8 == 0, lda / ldb % 8 == 0.
// ----------------------------------------------------------------------------
struct GemmBfArgs {
    const uint16_t* A; int lda;
    const uint16_t* B; int ldb;
    float* C; int ldc;                 // fp32 output or null
    uint16_t* Cb; int ldcb;            // bf16 output or null
    int M, N, K;
    const float* bias;
    const float* bias_b;               // a second [N] bias added to it (two layers summed into one pre-activation: the K-concatenated readout), or null
    const float* add; int ldadd;
    const float* rowadd; int ldrow; int rowgroup;
    int act;                           // 0 none, 1 tanh
    const float* mul; int ldmul;       // [M,N] multiplier applied after act (dropout mask), or null
    // two problems that share A in one launch (PL = L.Wcl + bl and LW = L.Wclt: B = [Wcl ; Wclt] stacked, N = both): columns
    // >= n_split (a multiple of 256) go to C2 / Cb2 with bias2, same leading dimensions; 0 = one output
    int n_split; float* C2; uint16_t* Cb2; const float* bias2;
    // deterministic split-K of the 256 x 256 kernel (tile 88 only): kslices > 1 cuts K, fp32 partial tiles go to ws
    // (>= kslices * M * N floats) and a second kernel adds them in order and applies the epilogue
    int kslices; float* ws; size_t ws_floats;
    int tile;                          // 0 = choose; 11 / 21 / 22 = (64*TM) x (64*TN) register-staged tile, 84 = 256 x 128 direct-to-LDS, 88 = 256 x 256 eight-phase (sweep tool)
    int xcd_remap;                     // internal
};
hipError_t launch_gemm_bf16(hipStream_t s, const GemmBfArgs& g);
// the 256 x 256 eight-phase kernel (gemm_bf16_8ph.hip): N % 256 == 0, K % 64 == 0; launch_gemm_bf16 routes to it (tile 88)
bool gemm_bf16_8ph_supported(const GemmBfArgs& g);
hipError_t launch_gemm_bf16_8ph(hipStream_t s, const GemmBfArgs& g);
constexpr int GEMM_BF_GROUP_MAX = 6;
struct GemmBfGroup { GemmBfArgs g[GEMM_BF_GROUP_MAX]; int tile_start[GEMM_BF_GROUP_MAX + 1]; int n; };
// several independent problems (each gemm_bf16_8ph_supported, no split-K) in ONE launch, the longest K first
hipError_t launch_gemm_bf16_8ph_group(hipStream_t s, const GemmBfArgs* gs, int n);
int gemm_bf16_8ph_slices(const GemmBfArgs& g);     // K-slices that fill the chip for this shape (1 = none)
// dst[r * ld_dst + c] = bf16(src[r * ld_src + c]), c < cols (cols % 8 == 0; 16-byte aligned rows): K-concatenated operands
hipError_t launch_cvt_bf16_2d(hipStream_t s, const float* src, size_t ld_src, uint16_t* dst, size_t ld_dst, size_t rows, int cols);
// dst[c * ld_dst + r] = bf16(src[r * ld_src + c]): k-contiguous operands of the weight-gradient (TN) GEMMs; src fp32 or bf16
hipError_t launch_transpose_to_bf16(hipStream_t s, const void* src, int src_is_bf16, size_t ld_src, uint16_t* dst, size_t ld_dst, int rows, int cols);
hipError_t launch_cvt_bf16(hipStream_t s, const float* src, uint16_t* dst, size_t n);               // n % 8 == 0
hipError_t launch_cvt_bf16_t(hipStream_t s, const float* src, int ld_src, uint16_t* dst, int ld_dst, int K, int N);  // dst[n][k] = src[k][n]
hipError_t launch_cvt_f32(hipStream_t s, const uint16_t* src, float* dst, size_t n);                 // exact widening, n % 8 == 0

// ----------------------------------------------------------------------------
// Register-streaming "skinny" grouped GEMM (skinny.hip) for M <= a few hundred rows:
// the weight matrix streams HBM/L2 -> VGPR exactly once per 16*mt rows, no LDS staging.
// One launch computes several independent output segments; each segment sums up to three
// (A,B) pairs that share the output tile:   C = epi(sum_p A_p[M,K_p] . B_p[K_p,N])
//   epi(v)[m,n] = scale * act(v + bias[n] + bias2[n] + add[m,n]) * (mul ? mul[m,n] : 1)
// N % 64 == 0, K_p % 16 == 0, rows of A_p 16-byte aligned.
// ----------------------------------------------------------------------------
struct SkPair {
    const float* A; const float* B; int lda, ldb, K;
    // tile-packed B (0 = plain row-major [K][ldb]): column tile t of 64 columns is stored as its own contiguous
    // [K][64] panel at B + t * tile_stride, so a K-slice of a tile is one linear stream instead of 256-byte
    // pieces at a power-of-two stride (which all map to the same few L2 channels)
    size_t tile_stride;
};
struct SkSeg {
    SkPair p[3]; int npairs;
    float* C; int ldc; int N;          // N = number of output columns of this segment
    const float* bias; const float* bias2;
    const float* add; int ldadd;
    const float* mul; int ldmul;
    float scale; int act;
};
struct SkArgs {
    SkSeg seg[6]; int nseg; int M;
    // K-split over blocks (backward recurrences: N = D is narrow, K = 4D is long).  kz > 1: block z owns every
    // kz-th K-slice and stores its raw partial tile to C + z * part_stride; the consumer sums the kz partials
    // in a fixed order.  No epilogue terms on that path.
    int kz; size_t part_stride;
};
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
// Row-panel kernels (panel.hip) for up to 512 rows: a workgroup owns every row and 16 / 32 output columns; the
// weights are read from panels repacked once per pass in MFMA-operand order (see the file header).
// ----------------------------------------------------------------------------
enum { PN_COLS_PLAIN = 0, PN_COLS_LSTM = 1 };
// Packed layout of an activation matrix [rows][K] (the A operand): [row / 16][k / 16][(k / 4) % 4 * 16 + row % 16][k % 4],
// i.e. the floats of one (m-tile, k-step) are the 1 KiB a wave loads with one dwordx4 per lane; four consecutive k
// (k % 4 == 0) stay one float4.  S = K / 16.  Written by the kernels that produce h, ctx, dpre and dsproj.
__host__ __device__ inline size_t pn_pack_offset(int row, int k, int S) {
    return ((((size_t)(row >> 4) * S + (k >> 4)) * 64) + ((k >> 2) & 3) * 16 + (row & 15)) * 4 + (k & 3);
}
struct PackJob {
    const float* W; int ldw;      // source matrix
    int src_t;                    // 0: W is [K][ldw];  1: the operand is W^T, W stored [N][ldw]
    int K;                        // k extent packed by this job (multiple of 16)
    int ntiles;                   // number of 16-column tiles
    int cols; int D;              // PN_COLS_PLAIN, or PN_COLS_LSTM with D hidden units (W has 4D columns)
    float* dst; int S_total; int s_off;   // destination panel set: k-steps per tile, first k-step filled by this job
};
hipError_t launch_pack_panels(hipStream_t s, const PackJob& jb);
// several repacking jobs in ONE launch (a pass repacks six to ten weight matrices: one launch instead of as many)
constexpr int PACK_BATCH_MAX = 12;
struct PackBatch { PackJob j[PACK_BATCH_MAX]; unsigned blk0[PACK_BATCH_MAX + 1]; int n; };
bool pack_batch_add(PackBatch& b, const PackJob& jb);
hipError_t launch_pack_batch(hipStream_t s, const PackBatch& b);
// A: activations [M][lda], or with apk = 1 the packed layout written by the producing kernel (ceil(M / 16) * 16 rows
// allocated); P: packed weight panels of the segment's first tile
struct PnPair { const float* A; int lda; const float* P; int K; int apk; };
struct PnSeg {
    PnPair p[3]; int npairs;
    float* C; int ldc; int N;
    const float* bias; const float* bias2;
    const float* add; int ldadd;
    const float* mul; int ldmul;
    float scale; int act;
    float* Cpk;                        // optional: the result once more in the packed A layout (it feeds another panel GEMM)
    // Vocabulary statistics instead of (not: besides) the plain store -- the small-batch decode step (panel.hip):
    // per (row, column tile) one PnTileStats record of the biased values v[n], n < stats_V (column 0 excluded when
    // stats_skip0): tile max, sum exp(v - max), and the stats_kb largest values with their columns.
    float* stats; int stats_V, stats_kb, stats_skip0;
    // ancestral sampling (gen_sample(stochastic=True), model_attention.py:841, :913-918): with stats_seed given the ONE
    // candidate of a tile is its arg-max of v[n] + Gumbel noise (Gumbel-max: the arg-max over the whole vocabulary is a
    // draw from softmax(v)); record [2] = the perturbed value, [3] = v of that column.  Noise is keyed by
    // (*stats_seed, *stats_step, row, n): reproducible, independent of the tiling.
    const unsigned long long* stats_seed; const int* stats_step;     // device words (a captured graph replays with new draws)
};
// record layout (floats): [0] max, [1] sum of exp(v - max), [2 .. 2+8) values descending, [10 .. 18) their columns (int bits)
constexpr int PN_STATS_KB = 8;
constexpr int PN_STATS_REC = 2 + 2 * PN_STATS_KB;
struct PnArgs {
    PnSeg seg[6]; int nseg; int M;
    int kz; size_t part_stride;        // K split over gridDim.y: raw partial tiles to C + z * part_stride, no epilogue
    int stream_b;                      // weights loaded with the non-temporal policy (a matrix that should not displace the others in L2)
    int wide_from;                     // rows from which the launch takes the wide kernel (panelw.hip); 0 = its default, 65
    int plain_order;                   // wide kernel: column blocks in launch order (segments of different K: the XCD-contiguous order would give whole XCDs the long blocks)
};
void pn_seg_defaults(PnSeg& s);
bool panel_supported(int M);
hipError_t launch_panel(hipStream_t s, const PnArgs& a);
int panel_tile_cols(const PnArgs& a);   // 16 or 32: column-tile width launch_panel picks (unit of PnSeg::stats records)
struct LstmPnArgs {
    PnPair p[3]; int npairs;           // panels packed with PN_COLS_LSTM
    const float* pre_add; int ldpre; const float* bias;
    const float* dp; int lddp; const float* mask;
    const float* h_prev; const float* c_prev; float* h_out; float* c_out; float* gates;
    const float* d1; int ldd1; float d1_scalar; float* hd_out;
    float* h_pk;                       // optional: h_out once more in the packed A layout (next step's state projections)
    float* hd_pk;                      // optional: hd_out in the packed A layout (sampler readout)
    int M, D;
};
hipError_t launch_lstm_panel(hipStream_t s, const LstmPnArgs& a);
// panelw.hip: the same launches for 65 .. 2048 rows on 32-column blocks of v_mfma_f32_32x32x2 (launch_panel / launch_lstm_panel
// route there when *_wide_supported; statistics records are then per 32 columns)
bool panel_wide_supported(const PnArgs& a);
bool lstm_panel_wide_supported(const LstmPnArgs& a);
hipError_t launch_panel_wide(hipStream_t s, const PnArgs& a);
hipError_t launch_lstm_panel_wide(hipStream_t s, const LstmPnArgs& a);
int panel_wide_tile_cols();
// dst (packed A layout, see panel.hip) = src [M][ld]; rows M .. roundup(M, 16) are zero filled
hipError_t launch_pack_rows(hipStream_t s, const float* src, int ld, int M, int K, float* dst);
size_t packed_rows_floats(int M, int K);

// "Rider": an independent row-panel GEMM  C[M x N] = A[M x K] . P (+ add)  computed by `nblocks` EXTRA workgroups at
// the front of the grid of an HBM-bound attention launch, on matrix cores that the attention workgroups leave idle
// (panel_inl.h rider_tile).  A in the packed activation layout (pn_pack_offset), P = 16-column packed panels
// (PN_COLS_PLAIN), M <= 64.  nblocks = (N / 16) * kz; kz > 1: raw K-slice partials at C + z * part_stride.
struct RiderArgs {
    const float* A; const float* P; float* C; int ldc;
    const float* add; int ldadd;       // [M,N] added in the epilogue (kz == 1 only), or null
    int M, N, K, kz; size_t part_stride;
    int nblocks;                       // 0: no rider
};
inline bool rider_shape_ok(const RiderArgs& r) {
    return r.nblocks > 0 && r.M >= 1 && r.M <= 64 && r.N % 16 == 0 && r.K % 16 == 0 && r.kz >= 1 && r.nblocks == (r.N / 16) * r.kz &&
           (r.K / 16) % r.kz == 0;
}

// ----------------------------------------------------------------------------
// attention kernels (attn.hip)
// ----------------------------------------------------------------------------
struct SpatialArgs {
    // per-video projected context, video index = vid[b] (null => b)
    const float* PL; const float* L; const float* LW;   // [nvid,T,K,D]; LW null in lt_mode 0
    int bf16;                                           // PL / L / LW hold bf16 (precision = bf16 handles; lt_mode 1)
    const float* PG; const float* PM;                   // [nvid,T,D]
    const int* vid;
    int group;                                          // > 1: rows b = v * group + h are the hypotheses of video v (beam search);
                                                        // fp32 only, vid ignored
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
    RiderArgs rider;   // optional GEMM computed by extra workgroups of this launch (fp32 per-row kernels only)
};
struct BeamArgs;
// upd (optional; one-hypothesis decode on the single-round-trip kernel only, spatial_update_supported): the beam bookkeeping of the
// PREVIOUS word runs as upd->nvid extra workgroups of this launch -- the attention of a word needs the new state projections but
// neither the chosen word nor its embedding, so the update leaves the critical path of the word loop
hipError_t launch_spatial(hipStream_t s, const SpatialArgs& a, const BeamArgs* upd = nullptr);
bool spatial_update_supported(const SpatialArgs& a);
bool spatial_update_row_workgroups(const SpatialArgs& a, const BeamArgs& u);   // would the riding update run as k workgroups per video?
bool spatial_rider_supported(const SpatialArgs& a);   // would launch_spatial pick a kernel that can carry a rider?

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
    float* cparts;     // [3][M,D] cg, cm, clt on their own (training: the backward pass needs <dcsum, c_x>), or null
    float* sel;        // [M] or null
    float* ctx;        // [M,D]
    float* ctx_pk;     // optional: ctx once more in the packed A layout of the row-panel LSTM kernel (pn_pack_offset)
    int M, T, D;
    const int* rowmap; // optional [M]: row b reads the scores eg / em / elt and the region contexts CL of row rowmap[b] (BeamArgs::rowmap)
};
hipError_t launch_temporal(hipStream_t s, const TemporalArgs& a);

// ----------------------------------------------------------------------------
// small kernels (misc.hip)
// ----------------------------------------------------------------------------
hipError_t launch_fill(hipStream_t s, float* p, float v, size_t n);
// red-zone scan (redzone.hip): *bad must hold ~0 on entry and still does afterwards when every canary byte is intact; see handle.h DevBuf
struct RedzoneRegion { const unsigned char* p; size_t nbytes; };
hipError_t launch_redzone_scan(hipStream_t s, const RedzoneRegion* regs, int nregs, int canary, unsigned long long* bad);
hipError_t launch_iota(hipStream_t s, int* p, int n, int mul);   // p[i] = i * mul
// mean[b,:] = sum_t G[b,t,:] / sum_t mask[b,t]   (model_attention.py:618, 649 / 739, 766)
hipError_t launch_ctx_mean(hipStream_t s, const float* G, const float* mask, float* mean, int B, int T, int D);
// emb[r,:] = (x[r] < 0) ? 0 : Wemb[x[r],:] ; shift>0: row r reads x[r - shift] and rows < shift are zero
hipError_t launch_embed(hipStream_t s, const int64_t* x, const float* Wemb, float* emb, int rows, int E, int V, int shift,
                        float* emb_pk = nullptr);   // emb_pk: optional copy in the packed A layout (pn_pack_offset)
// row softmax over V (ld = ldl) + optional NLL:  nll[r] = -log(p[r, x[r]] + 1e-8)
hipError_t launch_softmax_nll(hipStream_t s, const float* logits, int ldl, float* probs, int ldp,
                              const int64_t* x, float* nll, int64_t* argmax, int rows, int V);
// cost[b] = sum_t mask[t,b] * nll[t,b]
hipError_t launch_cost(hipStream_t s, const float* nll, const float* mask, float* cost, int t, int m);
// Bernoulli(0.5) in {0,1} from a counter-based hash
hipError_t launch_bernoulli(hipStream_t s, float* p, size_t n, uint64_t seed, uint64_t stream_id);
hipError_t launch_bernoulli3(hipStream_t s, float* p0, size_t n0, float* p1, size_t n1, float* p2, size_t n2, uint64_t seed, uint64_t stream0);   // streams stream0, +1, +2 in one launch
// uniform in [-1, 1)
hipError_t launch_uniform(hipStream_t s, float* p, size_t n, uint64_t seed, uint64_t stream_id);

// ----------------------------------------------------------------------------
// backward pass (bwd.hip) -- see the file header for the structure
// ----------------------------------------------------------------------------
struct LstmBwdArgs {
    const float* dh_pass;                 // [M,D] (1-m) dh of the step after, or null at the last step
    const float* dhU; int nU;             // K-slice partials of dpre_{s+1}.U^T   [nU][M,D]
    const float* dhW; int nW;             // partials of dsproj_{s+1}.Wd^T        [nW][M,D]
    const float* dselpre; const float* W_sel;   // rank-1 selector term of step s+1 (null if none)
    const float* dhd; const float* d1;    // readout gradient wrt hd[s] and its dropout multiplier
    const float* gates;                   // [M,4D] i,f,o,g of step s
    const float* c_prev; const float* c_new; const float* mask; const float* dp;   // dp [M,3D]
    float* dc;                            // [M,D] carried dc (in/out); read only if !last
    float* dpre;                          // [M,4D] out
    float* dpre_pk;                       // optional: dpre once more in the packed A layout (pn_pack_offset)
    float* dh_pass_out;                   // [M,D] out
    int M, D, last;
};

struct SpatialBwdArgs {
    const float* PL; const float* L; const float* LW;     // [M,T,K,D]
    int bf16;                                             // the three region tensors are stored as bf16 (bf16 handles)
    const float* sproj; int ldsp;
    // temporal part (was a launch of its own): dctx = readout term + K-slice partials of dpre.Wc^T, selector backward,
    // d alpha of the three temporal attentions and their softmax backward
    const float* dctxP; int nP;                           // partials of dpre.Wc^T  [nP][M,D]
    const float* dctx_r;                                  // [M,D] readout gradient wrt ctx (null if !ctx2out)
    const float* csum; const float* sel; int has_sel;     // forward: [M,D] cg+cm+clt, [M] selector gate
    const float* cparts;                                  // forward: [3][M,D] cg, cm, clt
    const float* G; const float* Mo; const float* CL;     // [M,T,D] (CL of this step)
    const float* rg; const float* rm; const float* rlt;   // [M,T] regulariser terms d/d alpha, or null
    float* dcsum;                                         // [M,D] out (kept for the deferred context gradients)
    float* dselpre;                                       // [M] out
    const float* alphal;                                  // [M,T,K]
    const float* PG; const float* PM;                     // [M,T,D]
    const float* ag; const float* am; const float* alt;   // [M,T] forward temporal attention weights
    const float* Ug; const float* Um;
    float* deg; float* dem; float* delt;                  // [M,T] out: temporal softmax backward
    float* dsgp; float* dsmp;                             // [M,T,D] out: per-frame dsg / dsm
    const float* rl;                                      // [M,T,K] or null
    const float* Ul; const float* Ult; const float* blt;
    float* dplt;                                          // [M,T,D]
    float* del;                                           // [M,T,K]
    float* dslp;                                          // [M,T,D] per-frame dsl
    int M, T, K, D;
    RiderArgs rider;   // optional GEMM computed by extra workgroups of this launch
};

struct CtxGradArgs {
    const float* PL; const float* LW; const float* PG; const float* PM;
    int bf16;                    // PL and LW are stored as bf16 (bf16 handles)
    const float* sproj;          // [S,M,4D]
    const float* dcsum;          // [S,M,D]
    const float* dplt;           // [S,M,T,D]
    const float* alphal; const float* del;     // [S,M,T,K]
    const float* alt; const float* delt; const float* am; const float* deg; const float* dem;   // [S,M,T]
    const float* Ul; const float* Ult; const float* Ug; const float* Um; const float* blt;
    float* dPL; float* dL; float* dLW;         // [M,T,K,D]
    float* dPG; float* dPM; float* dMo;        // [M,T,D]
    float* pUl;                                // [ctxgrad_groups(K)][M*T, D] partials, one block per region group
    float* pUlt; float* pUg; float* pUm;       // [M*T, D] partials
    int S, M, T, K, D;
};

hipError_t launch_dlogit(hipStream_t s, const float* probs, int ldp, const int64_t* x, const float* mask, float nll_scale,
                         float* dl, int ldd, int rows, int V, int Vp);
struct AlphaRegArgs { const float* alpha[4]; float* r[4]; float* sq[4]; size_t n[4]; float coef[4]; int count; };
hipError_t launch_alpha_reg(hipStream_t s, const AlphaRegArgs& a, int steps);    // up to four [steps, n] tensors in one launch
hipError_t launch_lstm_bwd(hipStream_t s, const LstmBwdArgs& a);
hipError_t launch_spatial_bwd(hipStream_t s, const SpatialBwdArgs& a);
hipError_t launch_reduce_T(hipStream_t s, const float* dslp, const float* dsgp, const float* dsmp, const float* dplt,
                           float* dsproj, int lddsp, int M, int T, int D, float* dsproj_pk = nullptr);
hipError_t launch_ctxgrad(hipStream_t s, const CtxGradArgs& a);
int ctxgrad_groups(int K);    // region groups (= workgroups per (row, frame) item, = blocks of pUl)
int colsum_parts(int rows, int N);
hipError_t launch_colsum(hipStream_t s, const float* X, int ldx, int rows, int N, float* part, float* dst, int accumulate,
                         const float* row_weights = nullptr);
// Batched column sums (bias gradients): up to 24 independent jobs dst[n] (+)= sum_r rw[r] * X[r, n] in one launch pair.
// Each job is cut into row slices whose partials land in `part`, a second kernel adds them in slice order
// (deterministic).  N % 4 == 0, ldx % 4 == 0, X 16-byte aligned.  `part` must hold 24 * 131072 floats.
struct ColsumJob { const float* X; const float* rw; float* dst; int ldx, rows, N, rs, accumulate, blk0, fblk0, part0; };
struct ColsumBatch { ColsumJob j[24]; int n; int nblk, nfblk; };
bool colsum_batch_add(ColsumBatch& b, const float* X, int ldx, int rows, int N, float* dst, int accumulate, const float* rw = nullptr);
hipError_t launch_colsum_batch(hipStream_t s, const ColsumBatch& b, float* part);
constexpr size_t COLSUM_BATCH_PART_FLOATS = (size_t)24 * 131072;
struct MultiSumArgs { const float* src[12]; size_t n[12]; float* dst[12]; float scale[12]; int count; };
hipError_t launch_multi_sum(hipStream_t s, const MultiSumArgs& a, float* part);   // dst[i][0] = scale[i] * sum(src[i][0:n[i]]); part: >= 384 floats
hipError_t launch_sum_all(hipStream_t s, const float* x, size_t n, float* dst, float scale, int accumulate);
hipError_t launch_tanh_bwd(hipStream_t s, const float* dy, const float* t, const float* mul, float* out, size_t n, int t_bf16 = 0);
// t[i] = mul[i] != 0 ? a[i] / mul[i] : 0: recovers tanh(z) from a = tanh(z) * mul where only a was kept (bf16 readout)
hipError_t launch_unmul(hipStream_t s, const float* a, const float* mul, float* t, size_t n);
hipError_t launch_add(hipStream_t s, const float* a, const float* b, float* out, size_t n, int accumulate);
// Plan of the deterministic embedding gradient, built on the host from the token ids when a batch is staged
// (api.cpp build_embed_plan): device pointers into one int buffer.
struct EmbedPlan {
    const int* perm;               // [ntok] token indices, grouped by word (ascending inside a word)
    const int* piece_start;        // [npieces + 1] into perm: at most 16 tokens of one word per piece
    const int* piece_word;         // [npieces] index of the piece's word in the distinct-word list
    const int* word_piece_start;   // [nwords + 1] into the piece list
    const int* word_id;            // [nwords] vocabulary index
    const int* multi_word;         // [nmulti] distinct-word indices that have more than one piece
    int npieces, nwords, nmulti;
};
// dWemb must be zero on entry (rows of absent words are not written); part: npieces * E floats
hipError_t launch_embed_bwd(hipStream_t s, const EmbedPlan& pl, const float* demb, float* dWemb, float* part, int E, int shift);
hipError_t launch_transpose(hipStream_t s, const float* in, int ldi, float* out, int ldo, int rows, int cols);
hipError_t launch_state0_bwd(hipStream_t s, const float* dh_pass, const float* dhU, int nU, const float* dhW, int nW,
                             const float* dselpre, const float* W_sel, const float* dc, const float* h0, const float* c0,
                             float* dph0, float* dpc0, int M, int D);
// part[block] = partial sums of (g + two_decay p)^2; nothing is written back
hipError_t launch_decay_sumsq(hipStream_t s, const float* g, const float* p, float two_decay, size_t n, float* part, int nblocks);
// gradient = (g + two_decay p) clipped by the global norm sqrt(g2[0])
hipError_t launch_adadelta(hipStream_t s, float* p, const float* g, float* rg2, float* ru2, size_t n, const float* g2, float clip_c,
                           float two_decay);

// ----------------------------------------------------------------------------
// batched device-side beam search (beam.hip)
// ----------------------------------------------------------------------------
// Everything stattn_beam_search initialises before the first word, in ONE launch (it was ~20 memsets / small copies of
// 4-5 us each plus the gaps between them: a quarter of a per-video decode call): the initial beam (one live, empty, zero-score
// hypothesis per video on row v * k, next word -1, :871-893), its states (row v * k = h0 / c0 of the video, other rows 0), the
// eval dropout multiplier 0.5, zeroed packed activation buffers, the packed initial states, the zero embedding of the first
// word (:803-804), and the word counter / ticket.  Any pointer may be null (buffer not in use).
struct BeamInitArgs {
    int nvid, k, D, E;
    int* vid; int* live_k; int* dead_k; int64_t* next_w; float* score0;
    const float* h0; const float* c0; float* hp; float* cp; float* hp_pk;
    float* dp;                                   // [M, 3D] <- 0.5
    float* zero[6]; size_t zero_n[6];            // buffers to clear (floats)
    float* emb;                                  // [M, E] <- 0
    int* ticket; int* step;
    int* rowmap;                                 // optional [M] <- identity (BeamArgs::rowmap)
};
hipError_t launch_beam_init(hipStream_t s, const BeamInitArgs& a);

struct BeamArgs {
    const float* probs; int ldp;        // [nvid*k, ldp] next-word probabilities of this step
    int V, k, D, maxlen, nvid, suppress_eos;
    // device counter: index of the word being decoded (so a captured graph replays unchanged).  INVARIANT (beam_inl.h advance_step,
    // a relaxed ticket without fences): every workgroup of the update's launch reads *step once, at its start, and nothing inside that
    // launch reads it after the last arriver's store (an agent-scope atomic store); every later consumer -- the statistics epilogue of
    // the next word's logits launch -- is a separate kernel behind a launch boundary.  A rider that read *step late would race.
    int* step;
    int* live_k; int* dead_k;           // [nvid]
    const float* hyp_score; float* hyp_score_out;   // [nvid*k] scores of the live hypotheses (in / out)
    const int* tok_in; int* tok_out;    // [nvid*k, maxlen] words of the live hypotheses (in / out)
    int* fin_tok; float* fin_score; int* fin_len;   // finished hypotheses, in order of death
    int64_t* next_w;                    // [nvid*k] word fed to the next step
    const float* h_step; const float* c_step;       // [nvid*k, D] state after this step
    float* h_next; float* c_next;       // [nvid*k, D] state gathered for the next step
    float* h_next_pk;                   // optional: h_next in the packed A layout of the row-panel kernels
    const float* Wemb; int E;           // embedding table: the update writes the next step's input embedding ...
    float* emb_next; float* emb_next_pk;   // ... [nvid*k, E] (and optionally its packed copy); null = not written
    int* ticket;                        // device int, zero: last workgroup of the update advances *step
    // small-batch decode step: instead of `probs`, per (row, vocabulary tile) statistics written by the logits launch
    // (PnSeg::stats): the update forms log-sum-exp per row and selects among the tiles' best candidates
    const float* stats; int ntile, tile_cols;     // tiles per row and their width (16 / 32 columns)
    int stochastic;                     // stats mode, k = 1: the candidate with the largest PERTURBED value is the draw; its
                                        // "cost" is the running sum of p(word) (:916), word 0 ends the caption (:917)
    // ... and gathers the NEXT step's state projections (computed from h of this step before the beam was re-ordered)
    const float* proj_step; float* proj_next; int nproj;   // [nvid*k, nproj] rows (sproj | preh), or null
    float* end_h; float* end_c; int* end_rows;      // [nvid*k, D], [nvid]: f_next's state outputs of the word that ended a video's loop
    // row workgroups (small path, k > 1; beam_inl.h): k update workgroups per video; null = one workgroup per video
    float* rw_cost; int* rw_idx;        // [nvid*k, 8] the nsel best candidates of every live row
    int* rw_ticket;                     // [nvid], zero: a video's last row workgroup merges them and does the bookkeeping
    int* rowmap;                        // optional [nvid*k]: row of the PARENT of the hypothesis now in each row (identity for unused rows): what the
                                        // temporal kernel needs when the next word's attention ran before the re-ordering (TemporalArgs::rowmap)
};
int beam_topk_splits(int nvid);
// part_cost / part_idx: nvid * beam_topk_splits(nvid) * 8 entries of scratch
hipError_t launch_beam_topk(hipStream_t s, const BeamArgs& a, float* part_cost, int* part_idx);
// the statistics records of PnSeg::stats (32-column tiles) from stored logits [M][ldl]
hipError_t launch_vocab_stats(hipStream_t s, const float* lg, int ldl, int M, int V, int ntile, int kb, int skip0, float* stats);
hipError_t launch_beam_update(hipStream_t s, const BeamArgs& a, const float* part_cost, const int* part_idx);   // also advances *a.step (last workgroup, ticket)

}  // namespace stattn
