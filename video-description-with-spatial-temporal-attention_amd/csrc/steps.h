// Building blocks shared by the translation units behind the C ABI: event profiling, the once-per-batch context
// projections, packed weight panels, one decoder timestep (run_step = _step, model_attention.py:366-459), dropout
// multipliers and the plan of the deterministic embedding gradient.  Defined in steps.cpp.
#pragma once
#include "handle.h"

namespace stattn_detail {

// ---- profiling helpers -------------------------------------------------------------
struct Prof {
    stattn_handle* h; int cls; hipEvent_t a = nullptr, b = nullptr; bool on;
    Prof(stattn_handle* h_, int c, bool enable = true) : h(h_), cls(c), on(h_->profiling && enable) {
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, h->stream);
    }
    ~Prof() {
        if (!on) return;
        (void)hipEventRecord(b, h->stream);
        h->ev_used.push_back({a, b, cls});
    }
};

void prof_collect(stattn_handle* h);
hipError_t gemm_nn(stattn_handle* h, const GemmArgs& g);
int gemm_group(stattn_handle* h, const GemmArgs* gs, int n);

// ---- context tensors of a batch / a video -------------------------------------------
struct CtxPtrs { float *G, *L, *Mo, *PG, *PL, *PM, *LW; };

struct BfWeights { uint16_t *ff_local, *ff_motion, *Wcg, *Wcl, *Wcm, *Wclt, *W, *Wl1, *Wl2, *Wo, *Wl12; };

int bf16_weights(stattn_handle* h, BfWeights* b, bool readout);
hipError_t gemm_bf(stattn_handle* h, const GemmBfArgs& g);
hipError_t gemm_bf_group(stattn_handle* h, const GemmBfArgs* gs, int n);
GemmBfArgs bf_args(const uint16_t* A, int lda, const uint16_t* B, int M, int N, int Kd);
int project_context(stattn_handle* h, int nv, int T, int K, const float* ctxg, const float* ctxl, const float* ctxm,
                    const CtxPtrs& c, const GemmArgs* extra = nullptr, const GemmBfArgs* extra_bf = nullptr);
int init_state(stattn_handle* h, int nv, int T, const float* G, const float* maskG, float* mean, float* h0, float* c0);

// ---- packed weight panels of the per-step kernels (panel.hip) -------------------------
struct FwdPanels { float *Wd, *U, *Wc, *W, *Wl1, *Wl2, *Wo; };
struct BwdPanels { float *WcT, *UT, *WdT; };

bool use_panels(const stattn_handle* h, int M, int min_rows = 17);
int pack(stattn_handle* h, const float* W, int ldw, int src_t, int K, int ntiles, int cols, float* dst, int S_total = 0, int s_off = 0);
int pack_flush(stattn_handle* h);      // run the repacking jobs collected by pack() as one launch
int pack_fwd_panels(stattn_handle* h, FwdPanels* p, bool readout);
int pack_bwd_panels(stattn_handle* h, BwdPanels* p);

// ---- one decoder timestep -------------------------------------------------------------
struct StepIO {
    int M, T, K;
    CtxPtrs c; const int* vid;
    int group;                       // beam search: rows v * group + h share video v (0 / 1: every row has its own video index)
    const float* h_prev; const float* c_prev;
    float *sproj, *preh;             // [M,4D] each (row stride ldproj, 0 = 4D)
    int ldproj;
    bool skip_hproj;                 // sproj / preh of this step are already in place (small-batch decode: they were computed
                                     // from h right after the previous step's LSTM and gathered with the beam)
    const float* xproj;              // [M,4D] (training: emb.W + b) or null
    const float* emb;                // [M,E]  (sampling: third LSTM pair) or null
    const float* dp; const float* mask; const float* d1;
    float *alphal, *CL, *eg, *em, *elt, *plt, *alphag, *alpham, *alphalt, *csum, *sel, *ctx;
    float* cparts;                   // training: [3][M][D] cg, cm, clt of this step (or null)
    float *h_out, *c_out, *gates, *hd;
    const FwdPanels* pn;             // packed weight panels, or null -> the 64-column skinny kernels
    const float* h_prev_pk;          // with pn: h_prev in the packed A layout (or null: plain rows are gathered)
    float *h_out_pk, *ctx_pk;        // with pn: packed copies written by the LSTM / temporal kernels (or null)
    const float* emb_pk;             // with pn, sampling: emb in the packed A layout (or null)
    float* hd_pk;                    // with pn, sampling: packed copy of hd for the readout (or null)
    int phase;                       // 0: the whole step; 1: state projections + attention launch only; 2: temporal fuse + LSTM only
                                     // (one-hypothesis decode runs the attention of word w + 1 in the last launch of word w)
    const int* rowmap;               // phase 2 of a step whose attention ran before the beam was re-ordered: TemporalArgs::rowmap
    const stattn::BeamArgs* upd;     // phase 1 only: beam bookkeeping of the previous word, extra workgroups of the attention launch
};

int run_step(stattn_handle* h, const StepIO& io);
int prepare_masks(stattn_handle* h, int t, int m, float** dp, float** d1, float** d2);

// ---- embedding-gradient plan, input checks ----------------------------------------------
void build_embed_plan(const int64_t* x, int t, int m, std::vector<int>& buf, stattn_handle::EmbPlanHost& ph);
int stage_embed_plan(stattn_handle* h, const int64_t* x, int t, int m, int set, hipStream_t stream);
EmbedPlan device_embed_plan(stattn_handle* h, int set);
int check_words(stattn_handle* h, const int64_t* x, size_t n, const char* who);

}  // namespace stattn_detail
