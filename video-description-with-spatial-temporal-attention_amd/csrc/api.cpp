// C ABI of libstattn.so (see include/stattn.h).  Host-side orchestration only: parameter
// store, batch staging, the per-timestep launch sequence, and result copies.  All arithmetic
// lives in the hand-written gfx950 kernels (gemm.hip, skinny.hip, attn.hip, misc.hip).
#include "handle.h"

namespace stattn_detail { std::string g_create_error; }

namespace {


void add_param(stattn_handle* h, const char* name, int ndim, int64_t d0, int64_t d1, int ld = 0) {
    ParamInfo p;
    p.name = name; p.ndim = ndim; p.dims[0] = d0; p.dims[1] = d1;
    p.count = ndim == 0 ? 1 : (ndim == 1 ? (size_t)d0 : (size_t)d0 * d1);
    p.ld = (ndim == 2) ? (ld ? ld : (int)d1) : (ld ? ld : (ndim == 1 ? (int)d0 : 1));
    const size_t padded = ndim == 2 ? (size_t)d0 * p.ld : (size_t)p.ld;
    p.off = h->nflat;
    h->nflat += align_up(padded, 64);   // every array starts 256-byte aligned
    h->pindex[name] = (int)h->params.size();
    h->params.push_back(p);
}

float* pptr(stattn_handle* h, const char* name) {
    auto it = h->pindex.find(name);
    return it == h->pindex.end() ? nullptr : h->d_params + h->params[it->second].off;
}

// dict order = init_params order (model_attention.py:518-581, 180-282)
void build_param_table(stattn_handle* h) {
    const int D = h->D, E = h->E, V = h->V, Vp = h->Vp;
    add_param(h, "Wemb", 2, V, E);
    add_param(h, "ff_state_W", 2, D, D);  add_param(h, "ff_state_b", 1, D, 0);
    add_param(h, "ff_memory_W", 2, D, D); add_param(h, "ff_memory_b", 1, D, 0);
    add_param(h, "ff_local_W", 2, h->Fl, D);  add_param(h, "ff_local_b", 1, D, 0);
    add_param(h, "ff_motion_W", 2, h->Fm, D); add_param(h, "ff_motion_b", 1, D, 0);
    add_param(h, "decoder_W", 2, E, 4 * D);
    add_param(h, "decoder_U", 2, D, 4 * D);
    add_param(h, "decoder_b", 1, 4 * D, 0);
    add_param(h, "decoder_Wc", 2, D, 4 * D);
    add_param(h, "decoder_Wcg_att", 2, D, D);
    add_param(h, "decoder_Wcm_att", 2, D, D);
    add_param(h, "decoder_Wclt_att", 2, D, D);
    add_param(h, "decoder_Wdg_att", 2, D, D);
    add_param(h, "decoder_Wdm_att", 2, D, D);
    add_param(h, "decoder_Wdlt_att", 2, D, D);
    add_param(h, "decoder_bg_att", 1, D, 0);
    add_param(h, "decoder_bm_att", 1, D, 0);
    add_param(h, "decoder_blt_att", 1, D, 0);
    add_param(h, "decoder_Wcl_att", 2, D, D);
    add_param(h, "decoder_Wdl_att", 2, D, D);
    add_param(h, "decoder_bl_att", 1, D, 0);
    add_param(h, "decoder_Ug_att", 2, D, 1);  add_param(h, "decoder_cg_att", 1, 1, 0);
    add_param(h, "decoder_Um_att", 2, D, 1);  add_param(h, "decoder_cm_att", 1, 1, 0);
    add_param(h, "decoder_Ult_att", 2, D, 1); add_param(h, "decoder_clt_att", 1, 1, 0);
    add_param(h, "decoder_Ul_att", 2, D, 1);  add_param(h, "decoder_cl_att", 1, 1, 0);
    if (h->opt.selector) {
        add_param(h, "decoder_W_sel", 2, D, 1);
        add_param(h, "decoder_b_sel", 0, 0, 0);
    }
    add_param(h, "ff_logit_lstm_W", 2, D, E); add_param(h, "ff_logit_lstm_b", 1, E, 0);
    if (h->opt.ctx2out) {
        add_param(h, "ff_logit_ctxglm_W", 2, D, E); add_param(h, "ff_logit_ctxglm_b", 1, E, 0);
    }
    // vocabulary projection: device layout padded to Vp = roundup(V, 128) columns (zeros)
    add_param(h, "ff_logit_W", 2, E, V, Vp);
    add_param(h, "ff_logit_b", 1, V, 0, Vp);
}

void bind_weights(stattn_handle* h) {
    Weights& w = h->w;
    w.Wemb = pptr(h, "Wemb");
    w.ff_state_W = pptr(h, "ff_state_W"); w.ff_state_b = pptr(h, "ff_state_b");
    w.ff_memory_W = pptr(h, "ff_memory_W"); w.ff_memory_b = pptr(h, "ff_memory_b");
    w.ff_local_W = pptr(h, "ff_local_W"); w.ff_local_b = pptr(h, "ff_local_b");
    w.ff_motion_W = pptr(h, "ff_motion_W"); w.ff_motion_b = pptr(h, "ff_motion_b");
    w.W = pptr(h, "decoder_W"); w.U = pptr(h, "decoder_U"); w.b = pptr(h, "decoder_b"); w.Wc = pptr(h, "decoder_Wc");
    w.Wcg = pptr(h, "decoder_Wcg_att"); w.Wcm = pptr(h, "decoder_Wcm_att"); w.Wclt = pptr(h, "decoder_Wclt_att");
    w.Wdg = pptr(h, "decoder_Wdg_att"); w.Wdm = pptr(h, "decoder_Wdm_att"); w.Wdlt = pptr(h, "decoder_Wdlt_att");
    w.bg = pptr(h, "decoder_bg_att"); w.bm = pptr(h, "decoder_bm_att"); w.blt = pptr(h, "decoder_blt_att");
    w.Wcl = pptr(h, "decoder_Wcl_att"); w.Wdl = pptr(h, "decoder_Wdl_att"); w.bl = pptr(h, "decoder_bl_att");
    w.Ug = pptr(h, "decoder_Ug_att"); w.cg = pptr(h, "decoder_cg_att");
    w.Um = pptr(h, "decoder_Um_att"); w.cm = pptr(h, "decoder_cm_att");
    w.Ult = pptr(h, "decoder_Ult_att"); w.clt = pptr(h, "decoder_clt_att");
    w.Ul = pptr(h, "decoder_Ul_att"); w.cl = pptr(h, "decoder_cl_att");
    w.W_sel = pptr(h, "decoder_W_sel"); w.b_sel = pptr(h, "decoder_b_sel");
    w.Wl1 = pptr(h, "ff_logit_lstm_W"); w.bl1 = pptr(h, "ff_logit_lstm_b");
    w.Wl2 = pptr(h, "ff_logit_ctxglm_W"); w.bl2 = pptr(h, "ff_logit_ctxglm_b");
    w.Wo = pptr(h, "ff_logit_W"); w.bo = pptr(h, "ff_logit_b");
}


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

void prof_collect(stattn_handle* h) {
    if (h->ev_used.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (auto& e : h->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { h->k_ms[e.cls] += ms; h->k_n[e.cls] += 1; }
        h->ev_pool.push_back(e.a); h->ev_pool.push_back(e.b);
    }
    h->ev_used.clear();
}

// every plain (NN) launch of the LDS-tiled GEMM in the forward pass is timed as one class: its average
// duration is what rocprofv3 reports for the symbol gemm_kernel<.., false, false>
hipError_t gemm_nn(stattn_handle* h, const GemmArgs& g) {
    Prof pr(h, KC_GEMM_NN);
    const int seq = h->gemm_seq++;
    Prof one(h, KC_COUNT + (seq < KC_GEMM_SEQ ? seq : 0), seq < KC_GEMM_SEQ);   // the first 16 launches one by one
    return launch_gemm(h->stream, g, false, false);
}

// several independent plain GEMMs in one launch (gemm.hip launch_gemm_group); timed like one launch of the class
int gemm_group(stattn_handle* h, const GemmArgs* gs, int n) {
    static const char* nogroup = getenv("STATTN_GEMM_NOGROUP");     // A/B switch for tools: one launch per problem
    if (nogroup) {
        for (int i = 0; i < n; ++i) HIPCHK(h, gemm_nn(h, gs[i]));
        return STATTN_OK;
    }
    Prof pr(h, KC_GEMM_NN);
    const int seq = h->gemm_seq++;
    Prof one(h, KC_COUNT + (seq < KC_GEMM_SEQ ? seq : 0), seq < KC_GEMM_SEQ);
    HIPCHK(h, launch_gemm_group(h->stream, gs, n));
    return STATTN_OK;
}

// ---- shared building blocks ---------------------------------------------------------
// Project raw features of `nv` videos to the decoder's context tensors (the part f_next
// recomputes on every call in the reference, model_attention.py:782-785 + 322-326).
struct CtxPtrs { float *G, *L, *Mo, *PG, *PL, *PM, *LW; };

// ---- bf16 path (precision = 1) ----------------------------------------------------------------------
// k-contiguous bf16 shadows ([N][K]) of the weight matrices the bf16 GEMMs read.  Rebuilt from the fp32 master
// copy at every use (once per minibatch / once per decoded video: ~35 M elements, tens of microseconds), so they can
// never go stale whichever way the parameters were written (set_param, update, RCCL broadcast into the flat buffer).
struct BfWeights { uint16_t *ff_local, *ff_motion, *Wcg, *Wcl, *Wcm, *Wclt, *W, *Wl1, *Wl2, *Wo; };

int bf16_weights(stattn_handle* h, BfWeights* b, bool readout) {
    const int D = h->D, E = h->E, Vp = h->Vp;
    const Weights& w = h->w;
    hipStream_t s = h->stream;
    struct Item { const char* name; const float* src; int K, N; uint16_t** dst; bool ro; };
    const Item items[] = {
        {"bw_ff_local", w.ff_local_W, h->Fl, D, &b->ff_local, false}, {"bw_ff_motion", w.ff_motion_W, h->Fm, D, &b->ff_motion, false},
        {"bw_Wcg", w.Wcg, D, D, &b->Wcg, false}, {"bw_Wcl", w.Wcl, D, D, &b->Wcl, false},
        {"bw_Wcm", w.Wcm, D, D, &b->Wcm, false}, {"bw_Wclt", w.Wclt, D, D, &b->Wclt, false},
        {"bw_W", w.W, E, 4 * D, &b->W, true}, {"bw_Wl1", w.Wl1, D, E, &b->Wl1, true},
        {"bw_Wl2", w.Wl2, D, E, &b->Wl2, true}, {"bw_Wo", w.Wo, E, Vp, &b->Wo, true},
    };
    for (const Item& it : items) {
        if (it.ro != readout || !it.src) continue;      // absent parameter (ff_logit_ctxglm without ctx2out)
        CHK(getbuf_t(h, it.name, (size_t)it.K * it.N, it.dst));
        HIPCHK(h, launch_cvt_bf16_t(s, it.src, it.N, *it.dst, it.K, it.K, it.N));
    }
    return STATTN_OK;
}

hipError_t gemm_bf(stattn_handle* h, const GemmBfArgs& g) {
    Prof pr(h, KC_GEMM_NN);
    const int seq = h->gemm_seq++;
    Prof one(h, KC_COUNT + (seq < KC_GEMM_SEQ ? seq : 0), seq < KC_GEMM_SEQ);
    return launch_gemm_bf16(h->stream, g);
}
GemmBfArgs bf_args(const uint16_t* A, int lda, const uint16_t* B, int M, int N, int Kd) {
    GemmBfArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = Kd; g.M = M; g.N = N; g.K = Kd; g.rowgroup = 1;
    return g;
}

// project_context with bf16 operands: L / PL / LW are written as bf16 INTO the (fp32-sized) buffers of CtxPtrs
int project_context_bf16(stattn_handle* h, int nv, int T, int K, const float* ctxg, const float* ctxl, const float* ctxm,
                         const CtxPtrs& c) {
    const int D = h->D;
    const Weights& w = h->w;
    hipStream_t s = h->stream;
    Prof pr(h, KC_PROLOGUE);
    BfWeights bw{};
    CHK(bf16_weights(h, &bw, false));
    const size_t nl = (size_t)nv * T * K, nf = (size_t)nv * T;
    uint16_t *xl, *xm, *xg, *mo;
    CHK(getbuf_t(h, "bx_ctxl", nl * h->Fl, &xl));
    CHK(getbuf_t(h, "bx_ctxm", nf * h->Fm, &xm));
    CHK(getbuf_t(h, "bx_ctxg", nf * D, &xg));
    CHK(getbuf_t(h, "bx_Mo", nf * D, &mo));
    HIPCHK(h, launch_cvt_bf16(s, ctxl, xl, nl * h->Fl));
    HIPCHK(h, launch_cvt_bf16(s, ctxm, xm, nf * h->Fm));
    HIPCHK(h, launch_cvt_bf16(s, ctxg, xg, nf * D));
    uint16_t* Lb = reinterpret_cast<uint16_t*>(c.L);
    GemmBfArgs g = bf_args(xl, h->Fl, bw.ff_local, (int)nl, D, h->Fl);       // L = tanh(ctxl . ff_local_W + b)
    g.bias = w.ff_local_b; g.act = 1; g.Cb = Lb; g.ldcb = D;
    HIPCHK(h, gemm_bf(h, g));
    g = bf_args(xm, h->Fm, bw.ff_motion, (int)nf, D, h->Fm);                  // M = tanh(ctxm . ff_motion_W + b)
    g.bias = w.ff_motion_b; g.act = 1; g.C = c.Mo; g.ldc = D; g.Cb = mo; g.ldcb = D;
    HIPCHK(h, gemm_bf(h, g));
    g = bf_args(xg, D, bw.Wcg, (int)nf, D, D);                                // pctxg_
    g.bias = w.bg; g.C = c.PG; g.ldc = D;
    HIPCHK(h, gemm_bf(h, g));
    g = bf_args(Lb, D, bw.Wcl, (int)nl, D, D);                                // pctxl_
    g.bias = w.bl; g.Cb = reinterpret_cast<uint16_t*>(c.PL); g.ldcb = D;
    HIPCHK(h, gemm_bf(h, g));
    g = bf_args(mo, D, bw.Wcm, (int)nf, D, D);                                // pctxm_
    g.bias = w.bm; g.C = c.PM; g.ldc = D;
    HIPCHK(h, gemm_bf(h, g));
    g = bf_args(Lb, D, bw.Wclt, (int)nl, D, D);                               // LW = L . Wclt
    g.Cb = reinterpret_cast<uint16_t*>(c.LW); g.ldcb = D;
    HIPCHK(h, gemm_bf(h, g));
    return STATTN_OK;
}

// `extra`: one more independent plain GEMM that rides in the first launch (training: the x projection), or null.
int project_context(stattn_handle* h, int nv, int T, int K, const float* ctxg, const float* ctxl, const float* ctxm,
                    const CtxPtrs& c, const GemmArgs* extra = nullptr) {
    if (h->opt.precision == 1) return project_context_bf16(h, nv, T, K, ctxg, ctxl, ctxm, c);
    const int D = h->D;
    const Weights& w = h->w;
    Prof pr(h, KC_PROLOGUE);
    // Two grouped launches instead of six (seven) separate ones: the frame-level projections are 416-tile problems
    // that under-fill the chip on their own (77-90 TFLOP/s); as tail fillers of the region-level GEMMs they are
    // nearly free.  Launch 1: everything that reads raw inputs; launch 2: what reads L / M.
    GemmArgs g1[GEMM_GROUP_MAX], g2[GEMM_GROUP_MAX];
    int n1 = 0, n2 = 0;
    {   // L = tanh(ctxl . ff_local_W + b)  (:664-665 / :782-783)
        GemmArgs& g = g1[n1++];
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = ctxl; g.lda = h->Fl; g.B = w.ff_local_W; g.ldb = D; g.C = c.L; g.ldc = D;
        g.M = nv * T * K; g.N = D; g.K = h->Fl; g.bias = w.ff_local_b; g.act = 1;
    }
    {   // M = tanh(ctxm . ff_motion_W + b) (:666-667 / :784-785)
        GemmArgs& g = g1[n1++];
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = ctxm; g.lda = h->Fm; g.B = w.ff_motion_W; g.ldb = D; g.C = c.Mo; g.ldc = D;
        g.M = nv * T; g.N = D; g.K = h->Fm; g.bias = w.ff_motion_b; g.act = 1;
    }
    {   // pctxg_ (:322)
        GemmArgs& g = g1[n1++];
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = ctxg; g.lda = D; g.B = w.Wcg; g.ldb = D; g.C = c.PG; g.ldc = D; g.M = nv * T; g.N = D; g.K = D; g.bias = w.bg;
    }
    if (extra) g1[n1++] = *extra;
    CHK(gemm_group(h, g1, n1));
    {   // pctxl_ (:324)
        GemmArgs& g = g2[n2++];
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = c.L; g.lda = D; g.B = w.Wcl; g.ldb = D; g.C = c.PL; g.ldc = D; g.M = nv * T * K; g.N = D; g.K = D; g.bias = w.bl;
    }
    if (h->opt.lt_mode == 1) {   // LW = L . Wclt  (the :416 projection hoisted out of the time loop)
        GemmArgs& g = g2[n2++];
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = c.L; g.lda = D; g.B = w.Wclt; g.ldb = D; g.C = c.LW; g.ldc = D; g.M = nv * T * K; g.N = D; g.K = D;
    }
    {   // pctxm_ (:326)
        GemmArgs& g = g2[n2++];
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = c.Mo; g.lda = D; g.B = w.Wcm; g.ldb = D; g.C = c.PM; g.ldc = D; g.M = nv * T; g.N = D; g.K = D; g.bias = w.bm;
    }
    CHK(gemm_group(h, g2, n2));
    return STATTN_OK;
}

// mean of the global features + tanh(ff_state), tanh(ff_memory)  (:649, 657-660 / :766, 776-779)
int init_state(stattn_handle* h, int nv, int T, const float* G, const float* maskG, float* mean, float* h0, float* c0) {
    const int D = h->D;
    HIPCHK(h, launch_ctx_mean(h->stream, G, maskG, mean, nv, T, D));
    SkArgs a{};
    a.M = nv; a.nseg = 2;
    for (int i = 0; i < 2; ++i) {
        SkSeg& s = a.seg[i];
        skinny_seg_defaults(s);
        s.npairs = 1;
        s.p[0] = SkPair{mean, i == 0 ? h->w.ff_state_W : h->w.ff_memory_W, D, D, D, 0};
        s.C = i == 0 ? h0 : c0; s.ldc = D; s.N = D;
        s.bias = i == 0 ? h->w.ff_state_b : h->w.ff_memory_b;
        s.act = 1;
    }
    HIPCHK(h, launch_skinny(h->stream, a));
    return STATTN_OK;
}

// ---- packed weight panels of the per-step kernels (panel.hip).  Repacked at the start of every pass that uses them
// (the parameters may have changed through set_param / update / a broadcast): ~50 MB of copies against 30 steps.
struct FwdPanels { float *Wd, *U, *Wc, *W, *Wl1, *Wl2, *Wo; };
struct BwdPanels { float *WcT, *UT, *WdT; };

// min_rows: the training scan repacks the panels every pass and uses them from 17 rows up; beam search packs once per
// call and uses them for any batch (at 4 rows the 16-column panels still give 128+ workgroups where the 64-column
// kernels give 32)
bool use_panels(const stattn_handle* h, int M, int min_rows = 17) {
    static const char* off = getenv("STATTN_NO_PANELS");       // A/B switch for tools
    // (the kernels take up to 512 rows; past 256 the 64-column kernels, which split the rows over workgroups, are as fast)
    return !off && M >= min_rows && M <= 256 && panel_supported(M) && h->D % 16 == 0 && h->E % 16 == 0;
}

int pack(stattn_handle* h, const float* W, int ldw, int src_t, int K, int ntiles, int cols, float* dst, int S_total = 0, int s_off = 0) {
    PackJob jb{};
    jb.W = W; jb.ldw = ldw; jb.src_t = src_t; jb.K = K; jb.ntiles = ntiles; jb.cols = cols; jb.D = h->D;
    jb.dst = dst; jb.S_total = S_total ? S_total : K / 16; jb.s_off = s_off;
    HIPCHK(h, launch_pack_panels(h->stream, jb));
    return STATTN_OK;
}

int pack_fwd_panels(stattn_handle* h, FwdPanels* p, bool readout) {
    const int D = h->D, E = h->E, Vp = h->Vp;
    const Weights& w = h->w;
    CHK(getbuf_t(h, "pn_Wd", (size_t)4 * D * D, &p->Wd));
    CHK(getbuf_t(h, "pn_U", (size_t)4 * D * D, &p->U));
    CHK(getbuf_t(h, "pn_Wc", (size_t)4 * D * D, &p->Wc));
    const float* Wd[4] = {w.Wdl, w.Wdg, w.Wdm, w.Wdlt};
    for (int i = 0; i < 4; ++i) CHK(pack(h, Wd[i], D, 0, D, D / 16, PN_COLS_PLAIN, p->Wd + (size_t)i * D * D));
    CHK(pack(h, w.U, 4 * D, 0, D, 4 * D / 16, PN_COLS_PLAIN, p->U));
    CHK(pack(h, w.Wc, 4 * D, 0, D, D / 4, PN_COLS_LSTM, p->Wc));
    p->W = p->Wl1 = p->Wl2 = p->Wo = nullptr;
    if (readout) {      // sampler: emb.W joins the LSTM GEMM, and the readout MLP runs per step
        CHK(getbuf_t(h, "pn_W", (size_t)E * 4 * D, &p->W));
        CHK(getbuf_t(h, "pn_Wl1", (size_t)D * E, &p->Wl1));
        CHK(getbuf_t(h, "pn_Wl2", (size_t)D * E, &p->Wl2));
        CHK(getbuf_t(h, "pn_Wo", (size_t)E * Vp, &p->Wo));
        CHK(pack(h, w.W, 4 * D, 0, E, D / 4, PN_COLS_LSTM, p->W));
        CHK(pack(h, w.Wl1, E, 0, D, E / 16, PN_COLS_PLAIN, p->Wl1));
        if (h->opt.ctx2out) CHK(pack(h, w.Wl2, E, 0, D, E / 16, PN_COLS_PLAIN, p->Wl2));
        CHK(pack(h, w.Wo, Vp, 0, E, Vp / 16, PN_COLS_PLAIN, p->Wo));
    }
    return STATTN_OK;
}

// transposed recurrent weights of the reverse scan, packed straight from the untransposed parameters
int pack_bwd_panels(stattn_handle* h, BwdPanels* p) {
    const int D = h->D;
    const Weights& w = h->w;
    CHK(getbuf_t(h, "pn_WcT", (size_t)4 * D * D, &p->WcT));
    CHK(getbuf_t(h, "pn_UT", (size_t)4 * D * D, &p->UT));
    CHK(getbuf_t(h, "pn_WdT", (size_t)4 * D * D, &p->WdT));
    CHK(pack(h, w.Wc, 4 * D, 1, 4 * D, D / 16, PN_COLS_PLAIN, p->WcT));
    CHK(pack(h, w.U, 4 * D, 1, 4 * D, D / 16, PN_COLS_PLAIN, p->UT));
    const float* Wd[4] = {w.Wdl, w.Wdg, w.Wdm, w.Wdlt};
    for (int i = 0; i < 4; ++i) CHK(pack(h, Wd[i], D, 1, D, D / 16, PN_COLS_PLAIN, p->WdT, 4 * D / 16, i * D / 16));
    return STATTN_OK;
}

struct StepIO {
    int M, T, K;
    CtxPtrs c; const int* vid;
    int group;                       // beam search: rows v * group + h share video v (0 / 1: every row has its own video index)
    const float* h_prev; const float* c_prev;
    float *sproj, *preh;             // [M,4D] each
    const float* xproj;              // [M,4D] (training: emb.W + b) or null
    const float* emb;                // [M,E]  (sampling: third LSTM pair) or null
    const float* dp; const float* mask; const float* d1;
    float *alphal, *CL, *eg, *em, *elt, *plt, *alphag, *alpham, *alphalt, *csum, *sel, *ctx;
    float *h_out, *c_out, *gates, *hd;
    const FwdPanels* pn;             // packed weight panels, or null -> the 64-column skinny kernels
    const float* h_prev_pk;          // with pn: h_prev in the packed A layout (or null: plain rows are gathered)
    float *h_out_pk, *ctx_pk;        // with pn: packed copies written by the LSTM / temporal kernels (or null)
    const float* emb_pk;             // with pn, sampling: emb in the packed A layout (or null)
    float* hd_pk;                    // with pn, sampling: packed copy of hd for the readout (or null)
};

// one decoder timestep: _step, model_attention.py:366-459
int run_step(stattn_handle* h, const StepIO& io) {
    const int D = h->D, E = h->E;
    const Weights& w = h->w;
    if (io.pn) {   // state projections on the row-panel kernel: one launch, every weight byte streamed once
        Prof pr(h, KC_HPROJ);
        PnArgs a{};
        a.M = io.M; a.nseg = 2;
        PnSeg& s0 = a.seg[0];
        pn_seg_defaults(s0);
        const PnPair hA = io.h_prev_pk ? PnPair{io.h_prev_pk, D, nullptr, D, 1} : PnPair{io.h_prev, D, nullptr, D, 0};
        s0.npairs = 1; s0.p[0] = hA; s0.p[0].P = io.pn->Wd;
        s0.C = io.sproj; s0.ldc = 4 * D; s0.N = 4 * D;         // [Wdl | Wdg | Wdm | Wdlt]: 4 x D/16 consecutive tiles
        PnSeg& s1 = a.seg[1];
        pn_seg_defaults(s1);
        s1.npairs = 1; s1.p[0] = hA; s1.p[0].P = io.pn->U;
        s1.C = io.preh; s1.ldc = 4 * D; s1.N = 4 * D;
        if (io.xproj) { s1.add = io.xproj; s1.ldadd = 4 * D; }
        HIPCHK(h, launch_panel(h->stream, a));
    } else {   // state projections: h.[Wdl | Wdg | Wdm | Wdlt] -> sproj, h.U (+ x_) -> preh   (:371, 389, 402, 415, 437-438)
        Prof pr(h, KC_HPROJ);
        SkArgs a{};
        a.M = io.M; a.nseg = 5;
        const float* Wd[4] = {w.Wdl, w.Wdg, w.Wdm, w.Wdlt};
        for (int i = 0; i < 4; ++i) {
            SkSeg& s = a.seg[i];
            skinny_seg_defaults(s);
            s.npairs = 1; s.p[0] = SkPair{io.h_prev, Wd[i], D, D, D, 0};
            s.C = io.sproj + (size_t)i * D; s.ldc = 4 * D; s.N = D;
        }
        SkSeg& s = a.seg[4];
        skinny_seg_defaults(s);
        s.npairs = 1; s.p[0] = SkPair{io.h_prev, w.U, D, 4 * D, D, 0};
        s.C = io.preh; s.ldc = 4 * D; s.N = 4 * D;
        if (io.xproj) { s.add = io.xproj; s.ldadd = 4 * D; }
        HIPCHK(h, launch_skinny(h->stream, a));
    }
    {   // spatial attention + frame scores (:371-383, 389-397, 402-410, and 415-424 in lt_mode 1)
        Prof pr(h, KC_SPATIAL);
        SpatialArgs a{};
        a.PL = io.c.PL; a.L = io.c.L; a.LW = h->opt.lt_mode == 1 ? io.c.LW : nullptr;
        a.bf16 = h->opt.precision == 1;
        a.PG = io.c.PG; a.PM = io.c.PM; a.vid = io.vid;
        a.group = h->opt.precision != 1 ? io.group : 0;
        a.sproj = io.sproj; a.ldsp = 4 * D;
        a.Ul = w.Ul; a.cl = w.cl; a.Ug = w.Ug; a.cg = w.cg; a.Um = w.Um; a.cm = w.cm;
        a.Ult = w.Ult; a.clt = w.clt; a.blt = w.blt;
        a.alphal = io.alphal; a.CL = io.CL; a.eg = io.eg; a.em = io.em; a.elt = io.elt;
        a.M = io.M; a.T = io.T; a.K = io.K; a.D = D;
        HIPCHK(h, launch_spatial(h->stream, a));
    }
    if (h->opt.lt_mode == 0) {   // pctxlt = CL.Wclt + blt + pstatelt, tanh, . Ult  (:416-422) as one MFMA GEMM
        Prof pr(h, KC_LTGEMM);
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = io.CL; g.lda = D; g.B = w.Wclt; g.ldb = D; g.C = io.plt; g.ldc = D;
        g.M = io.M * io.T; g.N = D; g.K = D; g.bias = w.blt;
        g.rowadd = io.sproj + 3 * (size_t)D; g.ldrow = 4 * D; g.rowgroup = io.T; g.act = 1;
        HIPCHK(h, launch_gemm(h->stream, g, false, false));
        HIPCHK(h, launch_rowdot(h->stream, io.plt, D, w.Ult, w.clt, io.elt, io.M * io.T, D));
    }
    {   // three temporal softmaxes, weighted sums, sum-fusion, selector gate (:398-399, 411-412, 425-435)
        Prof pr(h, KC_TEMPORAL);
        TemporalArgs a{};
        a.eg = io.eg; a.em = io.em; a.elt = io.elt; a.G = io.c.G; a.Mo = io.c.Mo; a.vid = io.vid; a.CL = io.CL;
        a.h_prev = io.h_prev; a.W_sel = h->opt.selector ? w.W_sel : nullptr; a.b_sel = w.b_sel;
        a.alphag = io.alphag; a.alpham = io.alpham; a.alphalt = io.alphalt;
        a.csum = io.csum; a.sel = io.sel; a.ctx = io.ctx; a.ctx_pk = io.pn ? io.ctx_pk : nullptr;
        a.M = io.M; a.T = io.T; a.D = D;
        HIPCHK(h, launch_temporal(h->stream, a));
    }
    if (io.pn) {   // preact = h.U + x_ + ctx.Wc, gates, cell update (:437-457) on the row-panel kernel
        Prof pr(h, KC_LSTM);
        LstmPnArgs a{};
        a.npairs = 1; a.p[0] = io.ctx_pk ? PnPair{io.ctx_pk, D, io.pn->Wc, D, 1} : PnPair{io.ctx, D, io.pn->Wc, D, 0};
        if (io.emb) { a.p[1] = io.emb_pk ? PnPair{io.emb_pk, E, io.pn->W, E, 1} : PnPair{io.emb, E, io.pn->W, E, 0}; a.npairs = 2; a.bias = w.b; }
        a.h_pk = io.h_out_pk; a.hd_pk = io.hd_pk;
        a.pre_add = io.preh; a.ldpre = 4 * D;
        a.dp = io.dp; a.lddp = 3 * D; a.mask = io.mask;
        a.h_prev = io.h_prev; a.c_prev = io.c_prev; a.h_out = io.h_out; a.c_out = io.c_out; a.gates = io.gates;
        a.d1 = io.d1; a.ldd1 = D; a.d1_scalar = 0.5f; a.hd_out = io.hd;
        a.M = io.M; a.D = D;
        HIPCHK(h, launch_lstm_panel(h->stream, a));
    } else {   // preact = h.U + x_ + ctx.Wc, gates, cell update (:437-457)
        Prof pr(h, KC_LSTM);
        LstmArgs a{};
        a.npairs = 1; a.p[0] = SkPair{io.ctx, w.Wc, D, 4 * D, D, 0};
        if (io.emb) { a.p[1] = SkPair{io.emb, w.W, E, 4 * D, E, 0}; a.npairs = 2; a.bias = w.b; }
        a.pre_add = io.preh; a.ldpre = 4 * D;
        a.dp = io.dp; a.lddp = 3 * D; a.mask = io.mask;
        a.h_prev = io.h_prev; a.c_prev = io.c_prev; a.h_out = io.h_out; a.c_out = io.c_out; a.gates = io.gates;
        a.d1 = io.d1; a.ldd1 = D; a.d1_scalar = 0.5f; a.hd_out = io.hd;
        a.M = io.M; a.D = D;
        HIPCHK(h, launch_lstm(h->stream, a));
    }
    return STATTN_OK;
}

// dropout multiplier tensors dp (t,m,3D), d1 (t,m,D), d2 (t,m,E)
int prepare_masks(stattn_handle* h, int t, int m, float** dp, float** d1, float** d2) {
    const size_t n_dp = (size_t)t * m * 3 * h->D, n_d1 = (size_t)t * m * h->D, n_d2 = (size_t)t * m * h->E;
    CHK(getbuf_t(h, "dp", n_dp, dp));
    CHK(getbuf_t(h, "d1", n_d1, d1));
    CHK(getbuf_t(h, "d2", n_d2, d2));
    if (h->masks_user) {
        if (h->masks_t != t || h->masks_m != m)
            return fail(h, STATTN_ESTATE, "dropout masks were supplied for (t=%d,m=%d) but the batch is (t=%d,m=%d)",
                        h->masks_t, h->masks_m, t, m);
        return STATTN_OK;
    }
    if (h->use_noise != 0.f) {   // trng.binomial(p=0.5) (:474-477, common.py:94-99); own counter-based generator
        HIPCHK(h, launch_bernoulli(h->stream, *dp, n_dp, h->seed, 3 * h->draw + 0));
        HIPCHK(h, launch_bernoulli(h->stream, *d1, n_d1, h->seed, 3 * h->draw + 1));
        HIPCHK(h, launch_bernoulli(h->stream, *d2, n_d2, h->seed, 3 * h->draw + 2));
        h->draw++;
        h->masks_state = 2; h->masks_t = t; h->masks_m = m;
    } else if (!(h->masks_state == 1 && h->masks_t >= t && h->masks_m >= m && h->masks_t * h->masks_m >= t * m)) {
        HIPCHK(h, launch_fill(h->stream, *dp, 0.5f, n_dp));   // use_noise = 0: the constant 0.5 (:472, :477)
        HIPCHK(h, launch_fill(h->stream, *d1, 0.5f, n_d1));
        HIPCHK(h, launch_fill(h->stream, *d2, 0.5f, n_d2));
        h->masks_state = 1; h->masks_t = t; h->masks_m = m;
    }
    return STATTN_OK;
}

// Plan of the deterministic embedding gradient (bwd.hip embed_bwd_*): tokens 0 .. (t-1)*m - 1 are the ones whose
// embedding enters the scan (emb is shifted by one step, :613-617).  Layout of the int buffer:
// perm[ntok] | piece_start[np+1] | piece_word[np] | word_piece_start[nw+1] | word_id[nw] | multi_word[nmulti]
void build_embed_plan(const int64_t* x, int t, int m, std::vector<int>& buf, stattn_handle::EmbPlanHost& ph) {
    const int ntok = (t - 1) * m;
    std::vector<std::pair<int64_t, int>> tok(ntok > 0 ? ntok : 0);
    for (int i = 0; i < ntok; ++i) tok[i] = {x[i], i};
    std::sort(tok.begin(), tok.end());                     // by word, then by index (pairs: deterministic)
    std::vector<int> perm(ntok), piece_start, piece_word, word_piece_start, word_id, multi;
    for (int i = 0; i < ntok; ++i) perm[i] = tok[i].second;
    for (int i = 0; i < ntok;) {
        int j = i;
        while (j < ntok && tok[j].first == tok[i].first) ++j;
        const int wi = (int)word_id.size();
        word_id.push_back((int)tok[i].first);
        word_piece_start.push_back((int)piece_word.size());
        for (int p = i; p < j; p += 16) { piece_start.push_back(p); piece_word.push_back(wi); }
        if (j - i > 16) multi.push_back(wi);
        i = j;
    }
    piece_start.push_back(ntok);
    word_piece_start.push_back((int)piece_word.size());
    ph.ntok = ntok; ph.npieces = (int)piece_word.size(); ph.nwords = (int)word_id.size(); ph.nmulti = (int)multi.size();
    buf.clear();
    for (const std::vector<int>* v : {&perm, &piece_start, &piece_word, &word_piece_start, &word_id, &multi}) buf.insert(buf.end(), v->begin(), v->end());
    if (buf.empty()) buf.push_back(0);
}

// stage the plan of batch set `set` (stream-ordered copy; the host vector is kept alive in the handle until then)
int stage_embed_plan(stattn_handle* h, const int64_t* x, int t, int m, int set, hipStream_t stream) {
    static thread_local std::vector<int> buf;
    build_embed_plan(x, t, m, buf, h->emb_plan[set]);
    int* d;
    CHK(getbuf_t(h, bset(h, "embplan", set).c_str(), buf.size(), &d));
    HIPCHK(h, hipMemcpyAsync(d, buf.data(), buf.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(h, hipStreamSynchronize(stream));             // pageable source: the copy must finish before buf is reused
    return STATTN_OK;
}

EmbedPlan device_embed_plan(stattn_handle* h, int set) {
    const stattn_handle::EmbPlanHost& ph = h->emb_plan[set];
    const int* d = reinterpret_cast<const int*>(h->bufs[bset(h, "embplan", set)].p);
    EmbedPlan pl{};
    pl.perm = d; d += ph.ntok;
    pl.piece_start = d; d += ph.npieces + 1;
    pl.piece_word = d; d += ph.npieces;
    pl.word_piece_start = d; d += ph.nwords + 1;
    pl.word_id = d; d += ph.nwords;
    pl.multi_word = d;
    pl.npieces = ph.npieces; pl.nwords = ph.nwords; pl.nmulti = ph.nmulti;
    return pl;
}

// Wemb[x] raises IndexError in the reference for an out-of-range word (:613); the kernels would clamp silently
int check_words(stattn_handle* h, const int64_t* x, size_t n, const char* who) {
    for (size_t i = 0; i < n; ++i)
        if (x[i] < 0 || x[i] >= h->V) return fail(h, STATTN_EINVAL, "%s: word index %lld outside [0, %d)", who, (long long)x[i], h->V);
    return STATTN_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

const char* stattn_version(void) { return "stattn 0.1 (gfx950)"; }

const char* stattn_last_error(const stattn_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int stattn_create(const stattn_options* o, int device, void* stream, stattn_handle** out) {
    if (!o || !out) return fail(nullptr, STATTN_EINVAL, "null argument");
    *out = nullptr;
    if (o->dim <= 0 || o->dim % 64) return fail(nullptr, STATTN_EINVAL, "dim must be a positive multiple of 64 (got %d)", o->dim);
    if (o->dim_word <= 0 || o->dim_word % 64) return fail(nullptr, STATTN_EINVAL, "dim_word must be a positive multiple of 64 (got %d)", o->dim_word);
    if (o->n_words < 2) return fail(nullptr, STATTN_EINVAL, "n_words must be >= 2");
    if (o->ctxg_dim != o->dim)
        return fail(nullptr, STATTN_EINVAL, "ctxg_dim (%d) must equal dim (%d): the reference graph has no ff_global layer "
                    "(model_attention.py:553-554, 661-662)", o->ctxg_dim, o->dim);
    if (o->ctxl_dim <= 0 || o->ctxl_dim % 32 || o->ctxm_dim <= 0 || o->ctxm_dim % 32)
        return fail(nullptr, STATTN_EINVAL, "ctxl_dim and ctxm_dim must be positive multiples of 32");
    if (!o->use_dropout)
        return fail(nullptr, STATTN_EINVAL, "use_dropout must be true: the reference's False branch is broken (model_attention.py:479-481)");
    if (o->lt_mode != 0 && o->lt_mode != 1) return fail(nullptr, STATTN_EINVAL, "lt_mode must be 0 or 1");
    if (o->precision < 0 || o->precision > 2) return fail(nullptr, STATTN_EINVAL, "precision must be 0 (fp32), 1 (bf16) or 2 (fp32 with the large GEMMs on the bf16 matrix cores, three-term operands)");
    if (o->precision == 1 && o->lt_mode != 1) return fail(nullptr, STATTN_EINVAL, "the bf16 path needs lt_mode 1");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, STATTN_EHIP, "no HIP device available (%s): libstattn has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, STATTN_EINVAL, "device %d out of range (0..%d)", device, ndev - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, STATTN_EHIP, "hipSetDevice: %s", hipGetErrorString(e));

    stattn_handle* h = new stattn_handle();
    h->opt = *o;
    h->D = o->dim; h->E = o->dim_word; h->V = o->n_words; h->Vp = (int)align_up((size_t)o->n_words, 128);
    h->Fl = o->ctxl_dim; h->Fm = o->ctxm_dim; h->device = device;
    if (stream) { h->stream = static_cast<hipStream_t>(stream); h->own_stream = false; }
    else {
        e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(nullptr, STATTN_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    build_param_table(h);
    e = hipMalloc((void**)&h->d_params, h->nflat * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_grads, h->nflat * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(h->d_params, 0, h->nflat * sizeof(float), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_grads, 0, h->nflat * sizeof(float), h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        int rc = fail(nullptr, STATTN_EHIP, "parameter allocation (%zu floats): %s", h->nflat, hipGetErrorString(e));
        stattn_destroy(h);
        return rc;
    }
    bind_weights(h);
    *out = h;
    return STATTN_OK;
}

void stattn_destroy(stattn_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& kv : h->bufs) kv.second.release();
    for (auto& e : h->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    comm_release(h);
    if (h->beam_gexec) (void)hipGraphExecDestroy(h->beam_gexec);
    if (h->pin_io) (void)hipHostFree(h->pin_io);
    if (h->pin_plan[0]) (void)hipHostFree(h->pin_plan[0]);
    if (h->pin_plan[1]) (void)hipHostFree(h->pin_plan[1]);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_grads) (void)hipFree(h->d_grads);
    if (h->d_rg2) (void)hipFree(h->d_rg2);
    if (h->d_ru2) (void)hipFree(h->d_ru2);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->staged_ev) (void)hipEventDestroy(h->staged_ev);
    if (h->free_ev[0]) (void)hipEventDestroy(h->free_ev[0]);
    if (h->free_ev[1]) (void)hipEventDestroy(h->free_ev[1]);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int stattn_sync(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    gemm_clock_dump();   // no-op unless STATTN_GEMM_CLK is set (tools)
    return STATTN_OK;
}

int stattn_param_count(const stattn_handle* h) { return h ? (int)h->params.size() : 0; }
const char* stattn_param_name(const stattn_handle* h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr;
}
int stattn_param_shape(const stattn_handle* h, int i, int64_t dims[2], int* ndim) {
    if (!h || i < 0 || i >= (int)h->params.size() || !dims || !ndim) return STATTN_EINVAL;
    dims[0] = h->params[i].dims[0]; dims[1] = h->params[i].dims[1]; *ndim = h->params[i].ndim;
    return STATTN_OK;
}

static int param_copy(stattn_handle* h, float* base, const char* name, float* host_dst, const float* host_src, size_t n) {
    if (!h || !name) return STATTN_EINVAL;
    auto it = h->pindex.find(name);
    if (it == h->pindex.end()) return fail(h, STATTN_ENOTFOUND, "unknown parameter '%s'", name);
    const ParamInfo& p = h->params[it->second];
    if (n != p.count) return fail(h, STATTN_EINVAL, "parameter '%s' has %zu elements, got %zu", name, p.count, n);
    HIPCHK(h, hipSetDevice(h->device));
    float* dev = base + p.off;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (p.ndim == 2 && p.ld != (int)p.dims[1]) {   // padded rows (ff_logit_W)
        const size_t wbytes = (size_t)p.dims[1] * sizeof(float);
        if (host_src) HIPCHK(h, hipMemcpy2D(dev, (size_t)p.ld * sizeof(float), host_src, wbytes, wbytes, (size_t)p.dims[0], hipMemcpyHostToDevice));
        else HIPCHK(h, hipMemcpy2D(host_dst, wbytes, dev, (size_t)p.ld * sizeof(float), wbytes, (size_t)p.dims[0], hipMemcpyDeviceToHost));
    } else {
        if (host_src) HIPCHK(h, hipMemcpy(dev, host_src, n * sizeof(float), hipMemcpyHostToDevice));
        else HIPCHK(h, hipMemcpy(host_dst, dev, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return STATTN_OK;
}

int stattn_set_param(stattn_handle* h, const char* name, const float* src, size_t n) {
    if (!src) return STATTN_EINVAL;
    int rc = param_copy(h, h ? h->d_params : nullptr, name, nullptr, src, n);
    if (rc == STATTN_OK) { h->ck_proj = false; h->have_fwd = false; }
    return rc;
}
int stattn_get_param(stattn_handle* h, const char* name, float* dst, size_t n) {
    if (!dst) return STATTN_EINVAL;
    return param_copy(h, h ? h->d_params : nullptr, name, dst, nullptr, n);
}
int stattn_get_grad(stattn_handle* h, const char* name, float* dst, size_t n) {
    if (!dst) return STATTN_EINVAL;
    return param_copy(h, h ? h->d_grads : nullptr, name, dst, nullptr, n);
}
int stattn_param_buffer_dev(stattn_handle* h, void** p, size_t* n) {
    if (!h || !p || !n) return STATTN_EINVAL;
    *p = h->d_params; *n = h->nflat;
    return STATTN_OK;
}
int stattn_grad_buffer_dev(stattn_handle* h, void** p, size_t* n) {
    if (!h || !p || !n) return STATTN_EINVAL;
    *p = h->d_grads; *n = h->nflat;
    return STATTN_OK;
}

int stattn_set_use_noise(stattn_handle* h, float v) {
    if (!h) return STATTN_EINVAL;
    if ((v != 0.f) != (h->use_noise != 0.f)) h->masks_state = 0;
    h->use_noise = v;
    return STATTN_OK;
}
int stattn_set_seed(stattn_handle* h, uint64_t seed) {
    if (!h) return STATTN_EINVAL;
    h->seed = seed; h->draw = 0; h->host_rng = seed * 0x9E3779B97F4A7C15ull + 0x853c49e6748fea9bull;
    return STATTN_OK;
}

int stattn_set_dropout_masks(stattn_handle* h, const float* dp, const float* d1, const float* d2, int t, int m) {
    if (!h) return STATTN_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (!dp || !d1 || !d2) { h->masks_user = false; h->masks_state = 0; return STATTN_OK; }
    if (t <= 0 || m <= 0) return fail(h, STATTN_EINVAL, "bad mask shape");
    float *b_dp, *b_d1, *b_d2;
    const size_t n_dp = (size_t)t * m * 3 * h->D, n_d1 = (size_t)t * m * h->D, n_d2 = (size_t)t * m * h->E;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(getbuf_t(h, "dp", n_dp, &b_dp));
    CHK(getbuf_t(h, "d1", n_d1, &b_d1));
    CHK(getbuf_t(h, "d2", n_d2, &b_d2));
    HIPCHK(h, hipMemcpy(b_dp, dp, n_dp * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(b_d1, d1, n_d1 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(b_d2, d2, n_d2 * sizeof(float), hipMemcpyHostToDevice));
    h->masks_user = true; h->masks_t = t; h->masks_m = m; h->masks_state = 0;
    return STATTN_OK;
}

// ---- sampler ------------------------------------------------------------------------
int stattn_f_init(stattn_handle* h, const float* ctxg, const float* ctxg_mask, int T, float* out_h0, float* out_c0) {
    if (!h || !ctxg || !ctxg_mask || T <= 0 || !out_h0 || !out_c0) return fail(h, STATTN_EINVAL, "f_init: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    const int D = h->D;
    float *G, *mk, *mean, *h0, *c0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(getbuf_t(h, "fi_G", (size_t)T * D, &G));
    CHK(getbuf_t(h, "fi_mask", (size_t)T, &mk));
    CHK(getbuf_t(h, "fi_mean", (size_t)D, &mean));
    CHK(getbuf_t(h, "fi_h0", (size_t)D, &h0));
    CHK(getbuf_t(h, "fi_c0", (size_t)D, &c0));
    HIPCHK(h, hipMemcpyAsync(G, ctxg, (size_t)T * D * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(mk, ctxg_mask, (size_t)T * sizeof(float), hipMemcpyHostToDevice, h->stream));
    CHK(init_state(h, 1, T, G, mk, mean, h0, c0));
    HIPCHK(h, hipMemcpyAsync(out_h0, h0, D * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(out_c0, c0, D * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return STATTN_OK;
}

// Buffers of the resident video of the sampler: raw features and their projections.
static int video_buffers(stattn_handle* h, int T, int K, CtxPtrs* c, float** rawl, float** rawm) {
    const int D = h->D;
    CHK(getbuf_t(h, "sv_G", (size_t)T * D, &c->G));
    CHK(getbuf_t(h, "sv_rawl", (size_t)T * K * h->Fl, rawl));
    CHK(getbuf_t(h, "sv_rawm", (size_t)T * h->Fm, rawm));
    CHK(getbuf_t(h, "sv_L", (size_t)T * K * D, &c->L));
    CHK(getbuf_t(h, "sv_Mo", (size_t)T * D, &c->Mo));
    CHK(getbuf_t(h, "sv_PG", (size_t)T * D, &c->PG));
    CHK(getbuf_t(h, "sv_PL", (size_t)T * K * D, &c->PL));
    CHK(getbuf_t(h, "sv_PM", (size_t)T * D, &c->PM));
    CHK(getbuf_t(h, "sv_LW", h->opt.lt_mode == 1 ? (size_t)T * K * D : 1, &c->LW));
    return STATTN_OK;
}

// upload (when host features are given) and project the sampler's video
static int stage_video(stattn_handle* h, const float* ctxg, const float* ctxl, const float* ctxm, int T, int K, CtxPtrs* c) {
    float *rawl, *rawm;
    hipStream_t s = h->stream;
    if (ctxg) {
        if (h->ck_valid && (h->ck_T != T || h->ck_K != K)) HIPCHK(h, hipStreamSynchronize(s));   // buffers may be reallocated
        CHK(video_buffers(h, T, K, c, &rawl, &rawm));
        HIPCHK(h, hipMemcpyAsync(c->G, ctxg, (size_t)T * h->D * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(rawl, ctxl, (size_t)T * K * h->Fl * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(rawm, ctxm, (size_t)T * h->Fm * sizeof(float), hipMemcpyHostToDevice, s));
        h->ck_T = T; h->ck_K = K; h->ck_valid = true; h->ck_proj = false;
    } else {
        if (!h->ck_valid) return fail(h, STATTN_ESTATE, "f_next: no resident video (pass the features or call stattn_set_video)");
        if (h->ck_T != T || h->ck_K != K)
            return fail(h, STATTN_EINVAL, "f_next: the resident video is (T=%d,K=%d), the call says (T=%d,K=%d)", h->ck_T, h->ck_K, T, K);
        CHK(video_buffers(h, T, K, c, &rawl, &rawm));
    }
    if (!h->ck_proj) {    // new features, or the parameters changed since the last projection
        CHK(project_context(h, 1, T, K, c->G, rawl, rawm, *c));
        h->ck_proj = true;
    }
    return STATTN_OK;
}

int stattn_set_video(stattn_handle* h, const float* ctxg, const float* ctxl, const float* ctxm, int T, int K) {
    if (!h || !ctxg || !ctxl || !ctxm || T <= 0 || K <= 0) return fail(h, STATTN_EINVAL, "set_video: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    CtxPtrs c{};
    CHK(stage_video(h, ctxg, ctxl, ctxm, T, K, &c));
    HIPCHK(h, hipStreamSynchronize(h->stream));     // the host arrays are borrowed for the call only
    return STATTN_OK;
}

int stattn_f_next(stattn_handle* h, const int64_t* x, int m, const float* ctxg, const float* ctxg_mask,
                  const float* ctxl, const float* ctxl_mask, const float* ctxm, const float* ctxm_mask, int T, int K,
                  const float* h_in, const float* c_in, float* out_probs, int64_t* out_sample, float* out_h, float* out_c,
                  float* out_alphal, float* out_alphag, float* out_alpham, float* out_alphalt, float* out_logits) {
    (void)ctxg_mask; (void)ctxl_mask; (void)ctxm_mask;   // unused by the reference graph too (:848)
    const bool resident = !ctxg && !ctxl && !ctxm;
    if (!h || !x || m <= 0 || (!resident && (!ctxg || !ctxl || !ctxm)) || T <= 0 || K <= 0 || !h_in || !c_in)
        return fail(h, STATTN_EINVAL, "f_next: bad argument");
    for (int r = 0; r < m; ++r)        // Wemb[x] raises IndexError in the reference (:803-804); -1 marks the first word
        if (x[r] < -1 || x[r] >= h->V) return fail(h, STATTN_EINVAL, "f_next: word index %lld outside [-1, %d)", (long long)x[r], h->V);
    HIPCHK(h, hipSetDevice(h->device));
    const int D = h->D, E = h->E, V = h->V, Vp = h->Vp;
    const Weights& w = h->w;
    hipStream_t s = h->stream;
    // (no leading synchronise: every call ends with one, and the buffers below are only touched in stream order)

    // --- the video.  Host features given: uploaded and projected on EVERY call, like the reference graph
    // (:782-788) -- no content guessing.  All three NULL: the video staged by stattn_set_video (or by the last call
    // that passed features) is reused; its projections are redone only if the parameters changed since.
    CtxPtrs c{};
    CHK(stage_video(h, ctxg, ctxl, ctxm, T, K, &c));

    // --- step buffers
    int64_t *dx, *dargmax; int* vid;
    float *hp, *cp, *emb, *sproj, *preh, *dp, *al, *CL, *eg, *em, *elt, *plt, *ag, *am, *alt, *ctx, *ho, *co, *hd, *a1, *lg, *pr;
    // inputs {h | c | x} and outputs {h' | c' | probs} are each one device block mirrored by one pinned host block
    const size_t in_floats = (size_t)2 * m * D + 2 * (size_t)m;                 // x: m int64 = 2m floats, 8-byte aligned
    const size_t out_floats = (size_t)2 * m * D + (size_t)m * Vp;
    float *d_in, *d_out;
    CHK(getbuf_t(h, "sn_in", in_floats, &d_in));
    CHK(getbuf_t(h, "sn_out", out_floats, &d_out));
    hp = d_in; cp = d_in + (size_t)m * D; dx = reinterpret_cast<int64_t*>(d_in + (size_t)2 * m * D);
    if ((in_floats + out_floats) * 4 > h->pin_io_bytes) {
        if (h->pin_io) { (void)hipHostFree(h->pin_io); h->pin_io = nullptr; h->pin_io_bytes = 0; }
        HIPCHK(h, hipHostMalloc(&h->pin_io, (in_floats + out_floats) * 4, hipHostMallocDefault));
        h->pin_io_bytes = (in_floats + out_floats) * 4;
    }
    float* p_in = static_cast<float*>(h->pin_io);
    float* p_out = p_in + in_floats;
    CHK(getbuf_t(h, "sn_argmax", (size_t)m, &dargmax));
    CHK(getbuf_t(h, "sn_vid", (size_t)m, &vid));
    CHK(getbuf_t(h, "sn_emb", (size_t)m * E, &emb));
    CHK(getbuf_t(h, "sn_sproj", (size_t)m * 4 * D, &sproj));
    CHK(getbuf_t(h, "sn_preh", (size_t)m * 4 * D, &preh));
    CHK(getbuf_t(h, "sn_dp", (size_t)m * 3 * D, &dp));
    CHK(getbuf_t(h, "sn_al", (size_t)m * T * K, &al));
    CHK(getbuf_t(h, "sn_CL", (size_t)m * T * D, &CL));
    CHK(getbuf_t(h, "sn_eg", (size_t)m * T, &eg));
    CHK(getbuf_t(h, "sn_em", (size_t)m * T, &em));
    CHK(getbuf_t(h, "sn_elt", (size_t)m * T, &elt));
    CHK(getbuf_t(h, "sn_plt", h->opt.lt_mode == 0 ? (size_t)m * T * D : 1, &plt));
    CHK(getbuf_t(h, "sn_ag", (size_t)m * T, &ag));
    CHK(getbuf_t(h, "sn_am", (size_t)m * T, &am));
    CHK(getbuf_t(h, "sn_alt", (size_t)m * T, &alt));
    CHK(getbuf_t(h, "sn_ctx", (size_t)m * D, &ctx));
    ho = d_out; co = d_out + (size_t)m * D;
    CHK(getbuf_t(h, "sn_hd", (size_t)m * D, &hd));
    CHK(getbuf_t(h, "sn_a1", (size_t)m * E, &a1));
    CHK(getbuf_t(h, "sn_lg", (size_t)m * Vp, &lg));
    pr = d_out + (size_t)2 * m * D;

    memcpy(p_in, h_in, (size_t)m * D * sizeof(float));
    memcpy(p_in + (size_t)m * D, c_in, (size_t)m * D * sizeof(float));
    memcpy(p_in + (size_t)2 * m * D, x, (size_t)m * sizeof(int64_t));
    HIPCHK(h, hipMemcpyAsync(d_in, p_in, in_floats * 4, hipMemcpyHostToDevice, s));
    if (h->sn_m != m || h->sn_dp != dp || h->sn_vid != vid) {   // constant across the calls of a decode loop
        HIPCHK(h, launch_iota(s, vid, m, 0));                   // every hypothesis attends to video 0 (:786-788)
        HIPCHK(h, launch_fill(s, dp, 0.5f, (size_t)m * 3 * D)); // sampler runs with use_noise = 0 (:469-472)
        h->sn_m = m; h->sn_dp = dp; h->sn_vid = vid;
    }
    HIPCHK(h, launch_embed(s, dx, w.Wemb, emb, m, E, V, 0));    // :803-804

    StepIO io{};
    io.M = m; io.T = T; io.K = K; io.c = c; io.vid = vid;
    io.h_prev = hp; io.c_prev = cp; io.sproj = sproj; io.preh = preh; io.xproj = nullptr; io.emb = emb;
    io.dp = dp; io.mask = nullptr; io.d1 = nullptr;
    io.alphal = al; io.CL = CL; io.eg = eg; io.em = em; io.elt = elt; io.plt = plt;
    io.alphag = ag; io.alpham = am; io.alphalt = alt; io.csum = nullptr; io.sel = nullptr; io.ctx = ctx;
    io.h_out = ho; io.c_out = co; io.gates = nullptr; io.hd = hd;
    io.pn = nullptr;              // a handful of rows per call: the 64-column skinny kernels (no repacking per call)
    CHK(run_step(h, io));

    {   // readout (:817-838): a = 0.5 * tanh(0.5h.Wl1 + bl1 [+ emb] [+ ctx.Wl2 + bl2]); logit = a.Wo + bo
        SkArgs a{};
        a.M = m; a.nseg = 1;
        SkSeg& sg = a.seg[0];
        skinny_seg_defaults(sg);
        sg.npairs = 1; sg.p[0] = SkPair{hd, w.Wl1, D, E, D, 0};
        if (h->opt.ctx2out) { sg.p[1] = SkPair{ctx, w.Wl2, D, E, D, 0}; sg.npairs = 2; sg.bias2 = w.bl2; }
        sg.bias = w.bl1;
        if (h->opt.prev2out) { sg.add = emb; sg.ldadd = E; }
        sg.act = 1; sg.scale = 0.5f; sg.C = a1; sg.ldc = E; sg.N = E;
        HIPCHK(h, launch_skinny(s, a));
        SkArgs b{};
        b.M = m; b.nseg = 1;
        SkSeg& so = b.seg[0];
        skinny_seg_defaults(so);
        so.npairs = 1; so.p[0] = SkPair{a1, w.Wo, E, Vp, E, 0};
        so.bias = w.bo; so.C = lg; so.ldc = Vp; so.N = Vp;
        HIPCHK(h, launch_skinny(s, b));
        HIPCHK(h, launch_softmax_nll(s, lg, Vp, pr, Vp, nullptr, nullptr, dargmax, m, V));   // :840
    }

    // {h' | c' | probs} in one transfer to pinned memory; rows of probs are unpadded on the way to the caller
    HIPCHK(h, hipMemcpyAsync(p_out, d_out, (out_probs ? out_floats : (size_t)2 * m * D) * 4, hipMemcpyDeviceToHost, s));
    if (out_logits) HIPCHK(h, hipMemcpy2DAsync(out_logits, (size_t)V * 4, lg, (size_t)Vp * 4, (size_t)V * 4, m, hipMemcpyDeviceToHost, s));
    if (out_alphal) HIPCHK(h, hipMemcpyAsync(out_alphal, al, (size_t)m * T * K * 4, hipMemcpyDeviceToHost, s));
    if (out_alphag) HIPCHK(h, hipMemcpyAsync(out_alphag, ag, (size_t)m * T * 4, hipMemcpyDeviceToHost, s));
    if (out_alpham) HIPCHK(h, hipMemcpyAsync(out_alpham, am, (size_t)m * T * 4, hipMemcpyDeviceToHost, s));
    if (out_alphalt) HIPCHK(h, hipMemcpyAsync(out_alphalt, alt, (size_t)m * T * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (out_h) memcpy(out_h, p_out, (size_t)m * D * 4);
    if (out_c) memcpy(out_c, p_out + (size_t)m * D, (size_t)m * D * 4);
    if (out_probs)
        for (int r = 0; r < m; ++r) memcpy(out_probs + (size_t)r * V, p_out + (size_t)2 * m * D + (size_t)r * Vp, (size_t)V * 4);

    if (out_sample) {
        // next_sample = multinomial(next_probs).argmax(1) (:841): inverse-CDF draw with the library's own
        // generator (bit-parity with Theano's MRG stream is not a goal).  Without probs: arg-max.
        if (out_probs) {
            for (int r = 0; r < m; ++r) {
                h->host_rng ^= h->host_rng << 13; h->host_rng ^= h->host_rng >> 7; h->host_rng ^= h->host_rng << 17;
                const double u = (double)(h->host_rng >> 11) * (1.0 / 9007199254740992.0);
                double acc = 0.0; int64_t pick = V - 1;
                const float* p = out_probs + (size_t)r * V;
                int j = 0;
                for (; j + 64 <= V; j += 64) {          // whole chunks first (the inner sum vectorises), then the hit chunk
                    float cs = 0.f;
                    for (int q = 0; q < 64; ++q) cs += p[j + q];
                    if (u < acc + (double)cs) break;
                    acc += (double)cs;
                }
                for (; j < V; ++j) { acc += p[j]; if (u < acc) { pick = j; break; } }
                out_sample[r] = pick;
            }
        } else {
            HIPCHK(h, hipMemcpy(out_sample, dargmax, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost));
        }
    }
    return STATTN_OK;
}

// ---- batched beam search: gen_sample (model_attention.py:852-994) for many videos at once, on the device ----
// raw features of `nvid` videos -> HBM (shared by stattn_beam_stage and stattn_beam_search)
static int beam_stage_impl(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                           const float* ctxm, int T, int K) {
    hipStream_t s = h->stream;
    const int D = h->D;
    const size_t nG = (size_t)nvid * T * D, nL = (size_t)nvid * T * K * h->Fl, nM = (size_t)nvid * T * h->Fm;
    float *G, *rawl, *rawm, *mG;
    HIPCHK(h, hipStreamSynchronize(s));
    CHK(getbuf_t(h, "bs_G", nG, &G)); CHK(getbuf_t(h, "bs_rawl", nL, &rawl)); CHK(getbuf_t(h, "bs_rawm", nM, &rawm));
    CHK(getbuf_t(h, "bs_mG", (size_t)nvid * T, &mG));
    HIPCHK(h, hipMemcpyAsync(G, ctxg, nG * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(mG, ctxg_mask, (size_t)nvid * T * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(rawl, ctxl, nL * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(rawm, ctxm, nM * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    h->bk_n = nvid; h->bk_T = T; h->bk_K = K; h->bk_valid = true;
    return STATTN_OK;
}

int stattn_beam_stage(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                      const float* ctxm, int T, int K) {
    if (!h || nvid <= 0 || !ctxg || !ctxg_mask || !ctxl || !ctxm || T <= 0 || K <= 0)
        return fail(h, STATTN_EINVAL, "beam_stage: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    return beam_stage_impl(h, nvid, ctxg, ctxg_mask, ctxl, ctxm, T, K);
}

int stattn_beam_search(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                       const float* ctxm, int T, int K, int k, int maxlen, int suppress_eos,
                       int64_t* out_tokens, float* out_scores, int32_t* out_lens, int32_t* out_count) {
    const bool resident = !ctxg && !ctxg_mask && !ctxl && !ctxm;
    if (!h || nvid <= 0 || (!resident && (!ctxg || !ctxg_mask || !ctxl || !ctxm)) || T <= 0 || K <= 0 || k < 1 || k > 8 ||
        maxlen < 1 || !out_tokens || !out_scores || !out_lens || !out_count)
        return fail(h, STATTN_EINVAL, "beam_search: bad argument (1 <= k <= 8)");
    HIPCHK(h, hipSetDevice(h->device));
    const int D = h->D, E = h->E, V = h->V, Vp = h->Vp;
    const Weights& w = h->w;
    hipStream_t s = h->stream;
    HIPCHK(h, hipStreamSynchronize(s));
    const int M = nvid * k, L0 = maxlen;
    const size_t nG = (size_t)nvid * T * D, nL = (size_t)nvid * T * K * h->Fl, nM = (size_t)nvid * T * h->Fm, nLd = (size_t)nvid * T * K * D;

    // host features given: staged on every call (no content guessing); all four NULL: the videos staged by
    // stattn_beam_stage are decoded again (benchmarks, repeated decoding with new parameters)
    if (!resident) CHK(beam_stage_impl(h, nvid, ctxg, ctxg_mask, ctxl, ctxm, T, K));
    else if (!h->bk_valid || h->bk_n != nvid || h->bk_T != T || h->bk_K != K)
        return fail(h, STATTN_ESTATE, "beam_search: no staged videos of this shape (call stattn_beam_stage)");
    CtxPtrs c{};
    float *rawl, *rawm, *mG, *mean, *h0, *c0;
    CHK(getbuf_t(h, "bs_G", nG, &c.G)); CHK(getbuf_t(h, "bs_rawl", nL, &rawl)); CHK(getbuf_t(h, "bs_rawm", nM, &rawm));
    CHK(getbuf_t(h, "bs_mG", (size_t)nvid * T, &mG));
    CHK(getbuf_t(h, "bs_L", nLd, &c.L)); CHK(getbuf_t(h, "bs_Mo", nG, &c.Mo)); CHK(getbuf_t(h, "bs_PG", nG, &c.PG));
    CHK(getbuf_t(h, "bs_PL", nLd, &c.PL)); CHK(getbuf_t(h, "bs_PM", nG, &c.PM));
    CHK(getbuf_t(h, "bs_LW", h->opt.lt_mode == 1 ? nLd : 1, &c.LW));
    CHK(getbuf_t(h, "bs_mean", (size_t)nvid * D, &mean)); CHK(getbuf_t(h, "bs_h0", (size_t)nvid * D, &h0));
    CHK(getbuf_t(h, "bs_c0", (size_t)nvid * D, &c0));
    CHK(project_context(h, nvid, T, K, c.G, rawl, rawm, c));       // once per video, not once per word
    CHK(init_state(h, nvid, T, c.G, mG, mean, h0, c0));            // f_init (:880)

    int *vid, *live_k, *dead_k, *tok[2], *fin_tok, *fin_len;
    int64_t* next_w;
    float *hp, *cp, *ho, *co, *hd, *emb, *sproj, *preh, *dp, *al, *CL, *eg, *em, *elt, *plt, *ag, *am, *alt, *ctx, *a1, *lg, *pr,
          *score[2], *fin_score;
    CHK(getbuf_t(h, "bs_vid", (size_t)M, &vid));
    CHK(getbuf_t(h, "bs_live", (size_t)nvid, &live_k)); CHK(getbuf_t(h, "bs_dead", (size_t)nvid, &dead_k));
    CHK(getbuf_t(h, "bs_tok0", (size_t)M * L0, &tok[0])); CHK(getbuf_t(h, "bs_tok1", (size_t)M * L0, &tok[1]));
    CHK(getbuf_t(h, "bs_fin_tok", (size_t)M * L0, &fin_tok)); CHK(getbuf_t(h, "bs_fin_len", (size_t)M, &fin_len));
    CHK(getbuf_t(h, "bs_fin_score", (size_t)M, &fin_score));
    CHK(getbuf_t(h, "bs_score0", (size_t)M, &score[0])); CHK(getbuf_t(h, "bs_score1", (size_t)M, &score[1]));
    CHK(getbuf_t(h, "bs_next_w", (size_t)M, &next_w));
    int* d_step;
    CHK(getbuf_t(h, "bs_step", (size_t)1, &d_step));
    float *end_h, *end_c; int* end_rows;
    CHK(getbuf_t(h, "bs_end_h", (size_t)M * D, &end_h)); CHK(getbuf_t(h, "bs_end_c", (size_t)M * D, &end_c));
    CHK(getbuf_t(h, "bs_end_rows", (size_t)nvid, &end_rows));
    float* tk_cost; int* tk_idx;
    CHK(getbuf_t(h, "bs_tk_cost", (size_t)nvid * beam_topk_splits(nvid) * 8, &tk_cost));
    CHK(getbuf_t(h, "bs_tk_idx", (size_t)nvid * beam_topk_splits(nvid) * 8, &tk_idx));
    CHK(getbuf_t(h, "bs_hp", (size_t)M * D, &hp)); CHK(getbuf_t(h, "bs_cp", (size_t)M * D, &cp));
    CHK(getbuf_t(h, "bs_ho", (size_t)M * D, &ho)); CHK(getbuf_t(h, "bs_co", (size_t)M * D, &co));
    CHK(getbuf_t(h, "bs_hd", (size_t)M * D, &hd)); CHK(getbuf_t(h, "bs_emb", (size_t)M * E, &emb));
    CHK(getbuf_t(h, "bs_sproj", (size_t)M * 4 * D, &sproj)); CHK(getbuf_t(h, "bs_preh", (size_t)M * 4 * D, &preh));
    CHK(getbuf_t(h, "bs_dp", (size_t)M * 3 * D, &dp));
    CHK(getbuf_t(h, "bs_al", (size_t)M * T * K, &al)); CHK(getbuf_t(h, "bs_CL", (size_t)M * T * D, &CL));
    CHK(getbuf_t(h, "bs_eg", (size_t)M * T, &eg)); CHK(getbuf_t(h, "bs_em", (size_t)M * T, &em)); CHK(getbuf_t(h, "bs_elt", (size_t)M * T, &elt));
    CHK(getbuf_t(h, "bs_plt", h->opt.lt_mode == 0 ? (size_t)M * T * D : 1, &plt));
    CHK(getbuf_t(h, "bs_ag", (size_t)M * T, &ag)); CHK(getbuf_t(h, "bs_am", (size_t)M * T, &am)); CHK(getbuf_t(h, "bs_alt", (size_t)M * T, &alt));
    CHK(getbuf_t(h, "bs_ctx", (size_t)M * D, &ctx)); CHK(getbuf_t(h, "bs_a1", (size_t)M * E, &a1));
    CHK(getbuf_t(h, "bs_lg", (size_t)M * Vp, &lg)); CHK(getbuf_t(h, "bs_pr", (size_t)M * Vp, &pr));

    // initial beam: one live hypothesis per video (row v*k), empty, score 0, next word -1 (:871-893)
    {
        std::vector<int> hv(M), one(nvid, 1);
        std::vector<int64_t> nw(M, -1);
        for (int i = 0; i < M; ++i) hv[i] = i / k;
        HIPCHK(h, hipMemcpyAsync(vid, hv.data(), (size_t)M * 4, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(live_k, one.data(), (size_t)nvid * 4, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(next_w, nw.data(), (size_t)M * 8, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));     // the host vectors go out of scope
    }
    HIPCHK(h, hipMemsetAsync(dead_k, 0, (size_t)nvid * 4, s));
    HIPCHK(h, hipMemsetAsync(score[0], 0, (size_t)M * 4, s));
    HIPCHK(h, hipMemsetAsync(hp, 0, (size_t)M * D * 4, s));
    HIPCHK(h, hipMemsetAsync(cp, 0, (size_t)M * D * 4, s));
    HIPCHK(h, hipMemcpy2DAsync(hp, (size_t)k * D * 4, h0, (size_t)D * 4, (size_t)D * 4, nvid, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipMemcpy2DAsync(cp, (size_t)k * D * 4, c0, (size_t)D * 4, (size_t)D * 4, nvid, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, launch_fill(s, dp, 0.5f, (size_t)M * 3 * D));

    // one decoded word = a fixed sequence of 10 kernel launches whose arguments depend on the word index only through
    // the parity of the ping-pong buffers (the index itself lives in d_step on the device)
    FwdPanels pn{};
    const bool panels = use_panels(h, M, 1) && Vp % 16 == 0;
    float *hp_pk = nullptr, *ctx_pk = nullptr, *emb_pk = nullptr, *hd_pk = nullptr, *a1_pk = nullptr;
    if (panels) {
        CHK(pack_fwd_panels(h, &pn, true));
        // packed-A copies of every activation that feeds a row-panel GEMM, written by the kernel that produces it
        CHK(getbuf_t(h, "bs_hp_pk", packed_rows_floats(M, D), &hp_pk)); CHK(getbuf_t(h, "bs_ctx_pk", packed_rows_floats(M, D), &ctx_pk));
        CHK(getbuf_t(h, "bs_emb_pk", packed_rows_floats(M, E), &emb_pk)); CHK(getbuf_t(h, "bs_hd_pk", packed_rows_floats(M, D), &hd_pk));
        CHK(getbuf_t(h, "bs_a1_pk", packed_rows_floats(M, E), &a1_pk));
        for (float* q : {ctx_pk, hd_pk}) HIPCHK(h, hipMemsetAsync(q, 0, packed_rows_floats(M, D) * sizeof(float), s));
        for (float* q : {emb_pk, a1_pk}) HIPCHK(h, hipMemsetAsync(q, 0, packed_rows_floats(M, E) * sizeof(float), s));
        HIPCHK(h, launch_pack_rows(s, hp, D, M, D, hp_pk));          // initial states; later words: beam_update's gather
    }
    // first word: no previous word, zero embedding (:803-804); afterwards beam_update writes the embedding of the
    // word it selects (no lookup launch inside the loop)
    HIPCHK(h, hipMemsetAsync(emb, 0, (size_t)M * E * sizeof(float), s));
    int* d_ticket;
    CHK(getbuf_t(h, "bs_ticket", (size_t)1, &d_ticket));
    HIPCHK(h, hipMemsetAsync(d_ticket, 0, sizeof(int), s));
    auto enqueue_word = [&](int parity) -> int {
        StepIO io{};
        io.M = M; io.T = T; io.K = K; io.c = c; io.vid = vid; io.group = k;
        io.h_prev = hp; io.c_prev = cp; io.sproj = sproj; io.preh = preh; io.xproj = nullptr; io.emb = emb;
        io.dp = dp; io.mask = nullptr; io.d1 = nullptr;
        io.alphal = al; io.CL = CL; io.eg = eg; io.em = em; io.elt = elt; io.plt = plt;
        io.alphag = ag; io.alpham = am; io.alphalt = alt; io.csum = nullptr; io.sel = nullptr; io.ctx = ctx;
        io.h_out = ho; io.c_out = co; io.gates = nullptr; io.hd = hd;
        io.pn = panels ? &pn : nullptr;
        io.h_prev_pk = hp_pk; io.h_out_pk = nullptr; io.ctx_pk = ctx_pk; io.emb_pk = emb_pk; io.hd_pk = hd_pk;
        CHK(run_step(h, io));
        if (panels) {      // readout (:817-838) on the row-panel kernel
            PnArgs a{};
            a.M = M; a.nseg = 1;
            PnSeg& sg = a.seg[0];
            pn_seg_defaults(sg);
            sg.npairs = 1; sg.p[0] = PnPair{hd_pk, D, pn.Wl1, D, 1};
            if (h->opt.ctx2out) { sg.p[1] = PnPair{ctx_pk, D, pn.Wl2, D, 1}; sg.npairs = 2; sg.bias2 = w.bl2; }
            sg.Cpk = a1_pk;
            sg.bias = w.bl1;
            if (h->opt.prev2out) { sg.add = emb; sg.ldadd = E; }
            sg.act = 1; sg.scale = 0.5f; sg.C = a1; sg.ldc = E; sg.N = E;
            HIPCHK(h, launch_panel(s, a));
            PnArgs b{};
            b.M = M; b.nseg = 1;
            PnSeg& so = b.seg[0];
            pn_seg_defaults(so);
            so.npairs = 1; so.p[0] = PnPair{a1_pk, E, pn.Wo, E, 1};
            so.bias = w.bo; so.C = lg; so.ldc = Vp; so.N = Vp;
            HIPCHK(h, launch_panel(s, b));
            HIPCHK(h, launch_softmax_nll(s, lg, Vp, pr, Vp, nullptr, nullptr, nullptr, M, V));
        } else {
            SkArgs a{};
            a.M = M; a.nseg = 1;
            SkSeg& sg = a.seg[0];
            skinny_seg_defaults(sg);
            sg.npairs = 1; sg.p[0] = SkPair{hd, w.Wl1, D, E, D, 0};
            if (h->opt.ctx2out) { sg.p[1] = SkPair{ctx, w.Wl2, D, E, D, 0}; sg.npairs = 2; sg.bias2 = w.bl2; }
            sg.bias = w.bl1;
            if (h->opt.prev2out) { sg.add = emb; sg.ldadd = E; }
            sg.act = 1; sg.scale = 0.5f; sg.C = a1; sg.ldc = E; sg.N = E;
            HIPCHK(h, launch_skinny(s, a));
            SkArgs b{};
            b.M = M; b.nseg = 1;
            SkSeg& so = b.seg[0];
            skinny_seg_defaults(so);
            so.npairs = 1; so.p[0] = SkPair{a1, w.Wo, E, Vp, E, 0};
            so.bias = w.bo; so.C = lg; so.ldc = Vp; so.N = Vp;
            HIPCHK(h, launch_skinny(s, b));
            HIPCHK(h, launch_softmax_nll(s, lg, Vp, pr, Vp, nullptr, nullptr, nullptr, M, V));
        }
        BeamArgs ba{};
        ba.probs = pr; ba.ldp = Vp; ba.V = V; ba.k = k; ba.D = D; ba.maxlen = L0; ba.nvid = nvid; ba.step = d_step;
        ba.suppress_eos = suppress_eos;
        ba.live_k = live_k; ba.dead_k = dead_k; ba.hyp_score = score[parity]; ba.hyp_score_out = score[parity ^ 1];
        ba.tok_in = tok[parity]; ba.tok_out = tok[parity ^ 1];
        ba.fin_tok = fin_tok; ba.fin_score = fin_score; ba.fin_len = fin_len; ba.next_w = next_w;
        ba.h_step = ho; ba.c_step = co; ba.h_next = hp; ba.c_next = cp;
        ba.end_h = end_h; ba.end_c = end_c; ba.end_rows = end_rows; ba.h_next_pk = hp_pk;
        ba.Wemb = w.Wemb; ba.E = E; ba.emb_next = emb; ba.emb_next_pk = emb_pk; ba.ticket = d_ticket;
        HIPCHK(h, launch_beam_topk(s, ba, tk_cost, tk_idx));
        HIPCHK(h, launch_beam_update(s, ba, tk_cost, tk_idx));
        return STATTN_OK;
    };
    HIPCHK(h, hipMemsetAsync(d_step, 0, sizeof(int), s));

    // The launch-bound inner loop is captured once as a hipGraph of TWO words (even + odd parity) and replayed; the
    // host only comes back every 8 words to see whether every video has finished.  Falls back to eager launches if
    // the capture is refused (or STATTN_BEAM_NOGRAPH is set, for A/B runs).
    hipGraphExec_t gexec = nullptr;
    static const char* nograph = getenv("STATTN_BEAM_NOGRAPH");
    h->beam_graph_replays = 0;
    // everything a captured launch bakes in: shapes, options and every buffer the word sequence touches
    std::vector<uintptr_t> sig = {(uintptr_t)nvid, (uintptr_t)k, (uintptr_t)T, (uintptr_t)K, (uintptr_t)L0, (uintptr_t)suppress_eos,
                                  (uintptr_t)h->opt.lt_mode, (uintptr_t)h->opt.precision, (uintptr_t)s};
    for (const void* q : {(const void*)c.G, (const void*)c.L, (const void*)c.Mo, (const void*)c.PG, (const void*)c.PL, (const void*)c.PM,
                          (const void*)c.LW, (const void*)vid, (const void*)live_k, (const void*)dead_k,
                          (const void*)tok[0], (const void*)tok[1],
                          (const void*)fin_tok, (const void*)fin_len, (const void*)fin_score, (const void*)score[0],
                          (const void*)score[1], (const void*)next_w, (const void*)hp, (const void*)cp, (const void*)ho,
                          (const void*)co, (const void*)hd, (const void*)emb, (const void*)sproj, (const void*)preh, (const void*)dp,
                          (const void*)al, (const void*)CL, (const void*)eg, (const void*)em, (const void*)elt, (const void*)plt,
                          (const void*)ag, (const void*)am, (const void*)alt, (const void*)ctx, (const void*)a1, (const void*)lg,
                          (const void*)pr, (const void*)d_step, (const void*)tk_cost, (const void*)tk_idx, (const void*)pn.Wd,
                          (const void*)pn.U, (const void*)pn.Wc, (const void*)pn.W, (const void*)pn.Wl1, (const void*)pn.Wl2,
                          (const void*)pn.Wo, (const void*)end_h, (const void*)end_c, (const void*)end_rows, (const void*)hp_pk,
                          (const void*)ctx_pk, (const void*)emb_pk, (const void*)hd_pk, (const void*)a1_pk, (const void*)d_ticket})
        sig.push_back((uintptr_t)q);
    if (!nograph && !h->profiling && L0 >= 2 && h->beam_gexec && h->beam_gsig == sig) {
        gexec = h->beam_gexec;                                   // same buffers and shapes as last time: replay as is
    } else if (!nograph && !h->profiling && L0 >= 2) {
        if (h->beam_gexec) { (void)hipGraphExecDestroy(h->beam_gexec); h->beam_gexec = nullptr; }
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int r0 = enqueue_word(0), r1 = r0 == STATTN_OK ? enqueue_word(1) : r0;
            const hipError_t e = hipStreamEndCapture(s, &graph);
            ok = r0 == STATTN_OK && r1 == STATTN_OK && e == hipSuccess && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) { gexec = nullptr; (void)hipGetLastError(); }
        else { h->beam_gexec = gexec; h->beam_gsig = sig; }
    }
    int steps_run = 0;
    int rc_loop = STATTN_OK;
    for (int st = 0; st < L0;) {
        if (gexec && st + 2 <= L0) {
            if (hipGraphLaunch(gexec, s) != hipSuccess) { rc_loop = fail(h, STATTN_EHIP, "beam_search: hipGraphLaunch failed"); break; }
            st += 2;
            ++h->beam_graph_replays;
        } else {
            rc_loop = enqueue_word(st & 1);
            if (rc_loop != STATTN_OK) break;
            st += 1;
        }
        steps_run = st;
        if (!suppress_eos && (st & 7) == 0 && st < L0) {           // early exit once every video has finished
            std::vector<int> lv(nvid);
            if (hipMemcpyAsync(lv.data(), live_k, (size_t)nvid * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) { rc_loop = fail(h, STATTN_EHIP, "beam_search: live-count readback failed"); break; }
            bool any = false;
            for (int x : lv) any = any || x > 0;
            if (!any) break;
        }
    }
    if (rc_loop != STATTN_OK) return rc_loop;
    // results: finished hypotheses in order of death, then the remaining live ones (:987-992)
    {
        std::vector<int> lv(nvid), dv(nvid), ftok((size_t)M * L0), flen(M), ltok((size_t)M * L0);
        std::vector<float> fsc(M), lsc(M);
        const int fb = steps_run & 1;     // buffers written by the last executed step
        HIPCHK(h, hipMemcpyAsync(lv.data(), live_k, (size_t)nvid * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(dv.data(), dead_k, (size_t)nvid * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(ftok.data(), fin_tok, (size_t)M * L0 * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(flen.data(), fin_len, (size_t)M * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(fsc.data(), fin_score, (size_t)M * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(ltok.data(), tok[fb], (size_t)M * L0 * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(lsc.data(), score[fb], (size_t)M * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        for (size_t i = 0; i < (size_t)M * L0; ++i) out_tokens[i] = -1;
        for (int v = 0; v < nvid; ++v) {
            int n = 0;
            for (int j = 0; j < dv[v] && n < k; ++j, ++n) {
                const int ln = flen[v * k + j];
                for (int i = 0; i < ln; ++i) out_tokens[((size_t)v * k + n) * L0 + i] = ftok[((size_t)v * k + j) * L0 + i];
                out_lens[v * k + n] = ln; out_scores[v * k + n] = fsc[v * k + j];
            }
            for (int j = 0; j < lv[v] && n < k; ++j, ++n) {
                for (int i = 0; i < steps_run; ++i) out_tokens[((size_t)v * k + n) * L0 + i] = ltok[((size_t)v * k + j) * L0 + i];
                out_lens[v * k + n] = steps_run; out_scores[v * k + n] = lsc[v * k + j];
            }
            out_count[v] = n;
            for (int j = n; j < k; ++j) { out_lens[v * k + j] = 0; out_scores[v * k + j] = 0.f; }
        }
        // what stattn_beam_final_state hands out: videos whose loop ended early (live count 0) keep the rows saved by
        // beam_update; the others ran to maxlen and return the gathered states of their live hypotheses (:979-985)
        h->bf_nvid = nvid; h->bf_k = k; h->bf_live = lv; h->bf_fb = fb;
    }
    return STATTN_OK;
}

int stattn_beam_final_state(stattn_handle* h, float* out_h, float* out_c, int32_t* out_rows) {
    if (!h || !out_h || !out_c || !out_rows) return fail(h, STATTN_EINVAL, "beam_final_state: bad argument");
    if (h->bf_nvid <= 0) return fail(h, STATTN_ESTATE, "beam_final_state: no beam search has run");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int nvid = h->bf_nvid, k = h->bf_k, D = h->D;
    const size_t n = (size_t)nvid * k * D;
    std::vector<float> eh(n), ec(n), lh(n), lc(n);
    std::vector<int> er(nvid);
    HIPCHK(h, hipMemcpyAsync(eh.data(), findbuf(h, "bs_end_h"), n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(ec.data(), findbuf(h, "bs_end_c"), n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(er.data(), findbuf(h, "bs_end_rows"), (size_t)nvid * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(lh.data(), findbuf(h, "bs_hp"), n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(lc.data(), findbuf(h, "bs_cp"), n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    memset(out_h, 0, n * 4); memset(out_c, 0, n * 4);
    for (int v = 0; v < nvid; ++v) {
        const bool ended = h->bf_live[v] == 0;
        const int rows = ended ? er[v] : h->bf_live[v];
        out_rows[v] = rows;
        const size_t o = (size_t)v * k * D;
        memcpy(out_h + o, (ended ? eh.data() : lh.data()) + o, (size_t)rows * D * 4);
        memcpy(out_c + o, (ended ? ec.data() : lc.data()) + o, (size_t)rows * D * 4);
    }
    return STATTN_OK;
}

// ---- training graph -----------------------------------------------------------------
int stattn_set_batch(stattn_handle* h, const int64_t* x, const float* mask, int t, int m,
                     const float* ctxg, const float* mask_ctxg, const float* ctxl, const float* mask_ctxl,
                     const float* ctxm, const float* mask_ctxm, int T, int K) {
    (void)mask_ctxl; (void)mask_ctxm;   // unused by the reference graph (on_unused_input='ignore', :1127)
    if (!h || !x || !mask || !ctxg || !mask_ctxg || !ctxl || !ctxm || t <= 0 || m <= 0 || T <= 0 || K <= 0)
        return fail(h, STATTN_EINVAL, "set_batch: bad argument");
    CHK(check_words(h, x, (size_t)t * m, "set_batch"));
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    HIPCHK(h, hipStreamSynchronize(s));
    const int D = h->D;
    int64_t* dx; float *dmask, *G, *mG, *rl, *rm;
    CHK(getbuf_t(h, bcur(h, "x").c_str(), (size_t)t * m, &dx));
    CHK(getbuf_t(h, bcur(h, "mask").c_str(), (size_t)t * m, &dmask));
    CHK(getbuf_t(h, bcur(h, "G").c_str(), (size_t)m * T * D, &G));
    CHK(getbuf_t(h, bcur(h, "mG").c_str(), (size_t)m * T, &mG));
    CHK(getbuf_t(h, bcur(h, "rawl").c_str(), (size_t)m * T * K * h->Fl, &rl));
    CHK(getbuf_t(h, bcur(h, "rawm").c_str(), (size_t)m * T * h->Fm, &rm));
    HIPCHK(h, hipMemcpyAsync(dx, x, (size_t)t * m * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dmask, mask, (size_t)t * m * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(G, ctxg, (size_t)m * T * D * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(mG, mask_ctxg, (size_t)m * T * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(rl, ctxl, (size_t)m * T * K * h->Fl * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(rm, ctxm, (size_t)m * T * h->Fm * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    CHK(stage_embed_plan(h, x, t, m, h->cur_set, s));
    h->t = t; h->m = m; h->T = T; h->K = K;
    h->have_batch = true; h->have_fwd = false; h->have_bwd = false;
    return STATTN_OK;
}

// ---- asynchronous staging of the NEXT minibatch (data_engine.prepare_data -> HBM pipeline) ----------------
int stattn_host_alloc(size_t bytes, void** out) {
    if (!out) return STATTN_EINVAL;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 4, hipHostMallocDefault);
    if (e != hipSuccess) { g_create_error = std::string("hipHostMalloc: ") + hipGetErrorString(e); return STATTN_EHIP; }
    return STATTN_OK;
}
int stattn_host_free(void* p) {
    if (!p) return STATTN_OK;
    return hipHostFree(p) == hipSuccess ? STATTN_OK : STATTN_EHIP;
}

int stattn_prefetch_batch(stattn_handle* h, const int64_t* x, const float* mask, int t, int m,
                          const float* ctxg, const float* mask_ctxg, const float* ctxl, const float* mask_ctxl,
                          const float* ctxm, const float* mask_ctxm, int T, int K) {
    (void)mask_ctxl; (void)mask_ctxm;
    if (!h || !x || !mask || !ctxg || !mask_ctxg || !ctxl || !ctxm || t <= 0 || m <= 0 || T <= 0 || K <= 0)
        return fail(h, STATTN_EINVAL, "prefetch_batch: bad argument");
    CHK(check_words(h, x, (size_t)t * m, "prefetch_batch"));
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->copy_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->staged_ev, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->free_ev[0], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->free_ev[1], hipEventDisableTiming));
    }
    const int set = h->cur_set ^ 1;
    // the shadow set may still be read by kernels of the step before the last swap (the host runs ahead of the GPU)
    if (h->free_valid[set]) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->free_ev[set], 0));
    const int D = h->D;
    int64_t* dx; float *dmask, *G, *mG, *rl, *rm;
    // (buffer growth may hipFree/hipMalloc: the shadow set is idle by construction -- it was swapped out before)
    CHK(getbuf_t(h, bset(h, "x", set).c_str(), (size_t)t * m, &dx));
    CHK(getbuf_t(h, bset(h, "mask", set).c_str(), (size_t)t * m, &dmask));
    CHK(getbuf_t(h, bset(h, "G", set).c_str(), (size_t)m * T * D, &G));
    CHK(getbuf_t(h, bset(h, "mG", set).c_str(), (size_t)m * T, &mG));
    CHK(getbuf_t(h, bset(h, "rawl", set).c_str(), (size_t)m * T * K * h->Fl, &rl));
    CHK(getbuf_t(h, bset(h, "rawm", set).c_str(), (size_t)m * T * h->Fm, &rm));
    hipStream_t cs = h->copy_stream;
    HIPCHK(h, hipMemcpyAsync(dx, x, (size_t)t * m * sizeof(int64_t), hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipMemcpyAsync(dmask, mask, (size_t)t * m * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipMemcpyAsync(G, ctxg, (size_t)m * T * D * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipMemcpyAsync(mG, mask_ctxg, (size_t)m * T * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipMemcpyAsync(rl, ctxl, (size_t)m * T * K * h->Fl * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipMemcpyAsync(rm, ctxm, (size_t)m * T * h->Fm * 4, hipMemcpyHostToDevice, cs));
    {   // the embedding-gradient plan of the shadow set: a few KB through a pinned staging block per set (no host wait:
        // the block of this set last fed a copy two prefetches ago, and that copy finished before the swap in between)
        std::vector<int> buf;
        build_embed_plan(x, t, m, buf, h->emb_plan[set]);
        int* d;
        CHK(getbuf_t(h, bset(h, "embplan", set).c_str(), buf.size(), &d));
        if (buf.size() * sizeof(int) > h->pin_plan_bytes[set]) {
            HIPCHK(h, hipStreamSynchronize(cs));
            if (h->pin_plan[set]) { (void)hipHostFree(h->pin_plan[set]); h->pin_plan[set] = nullptr; h->pin_plan_bytes[set] = 0; }
            HIPCHK(h, hipHostMalloc(&h->pin_plan[set], 2 * buf.size() * sizeof(int), hipHostMallocDefault));
            h->pin_plan_bytes[set] = 2 * buf.size() * sizeof(int);
        }
        memcpy(h->pin_plan[set], buf.data(), buf.size() * sizeof(int));
        HIPCHK(h, hipMemcpyAsync(d, h->pin_plan[set], buf.size() * sizeof(int), hipMemcpyHostToDevice, cs));
    }
    HIPCHK(h, hipEventRecord(h->staged_ev, cs));
    h->p_t = t; h->p_m = m; h->p_T = T; h->p_K = K; h->have_pending = true;
    return STATTN_OK;
}

int stattn_swap_batch(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_pending) return fail(h, STATTN_ESTATE, "swap_batch: no prefetched batch (call stattn_prefetch_batch)");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->staged_ev, 0));   // compute stream waits for the copies, the host does not
    HIPCHK(h, hipEventRecord(h->free_ev[h->cur_set], h->stream));  // old set is free once everything enqueued so far ran
    h->free_valid[h->cur_set] = true;
    h->cur_set ^= 1;
    h->t = h->p_t; h->m = h->p_m; h->T = h->p_T; h->K = h->p_K;
    h->have_pending = false; h->have_batch = true; h->have_fwd = false; h->have_bwd = false;
    return STATTN_OK;
}

int stattn_forward_train(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_batch) return fail(h, STATTN_ESTATE, "forward_train: no batch staged (call stattn_set_batch)");
    HIPCHK(h, hipSetDevice(h->device));
    const int t = h->t, m = h->m, T = h->T, K = h->K, D = h->D, E = h->E, V = h->V, Vp = h->Vp;
    const Weights& w = h->w;
    hipStream_t s = h->stream;
    const size_t R = (size_t)t * m;

    int64_t* dx = (int64_t*)h->bufs[bcur(h, "x")].p;
    float* dmask = findbuf(h, bcur(h, "mask").c_str());
    float* mG = findbuf(h, bcur(h, "mG").c_str());
    float* rawl = findbuf(h, bcur(h, "rawl").c_str());
    float* rawm = findbuf(h, bcur(h, "rawm").c_str());
    CtxPtrs c{};
    c.G = findbuf(h, bcur(h, "G").c_str());
    float *mean, *emb, *xproj, *hs, *cs, *hd, *ctx, *csum, *sel, *al, *ag, *am, *alt, *CL, *gates, *sproj, *preh,
          *eg, *em, *elt, *plt, *z1, *a1, *tz, *lg, *pr, *nll, *cost, *dp, *d1, *d2;
    CHK(getbuf_t(h, "L", (size_t)m * T * K * D, &c.L));
    CHK(getbuf_t(h, "Mo", (size_t)m * T * D, &c.Mo));
    CHK(getbuf_t(h, "PG", (size_t)m * T * D, &c.PG));
    CHK(getbuf_t(h, "PL", (size_t)m * T * K * D, &c.PL));
    CHK(getbuf_t(h, "PM", (size_t)m * T * D, &c.PM));
    CHK(getbuf_t(h, "LW", h->opt.lt_mode == 1 ? (size_t)m * T * K * D : 1, &c.LW));
    CHK(getbuf_t(h, "mean", (size_t)m * D, &mean));
    CHK(getbuf_t(h, "emb", R * E, &emb));
    CHK(getbuf_t(h, "xproj", R * 4 * D, &xproj));
    CHK(getbuf_t(h, "hs", (R + m) * D, &hs));          // hs[0] = h0, hs[s+1] = state after step s
    CHK(getbuf_t(h, "cs", (R + m) * D, &cs));
    CHK(getbuf_t(h, "hd", R * D, &hd));
    CHK(getbuf_t(h, "ctx", R * D, &ctx));
    CHK(getbuf_t(h, "csum", R * D, &csum));
    CHK(getbuf_t(h, "sel", R, &sel));
    CHK(getbuf_t(h, "alphal", R * T * K, &al));
    CHK(getbuf_t(h, "alphag", R * T, &ag));
    CHK(getbuf_t(h, "alpham", R * T, &am));
    CHK(getbuf_t(h, "alphalt", R * T, &alt));
    CHK(getbuf_t(h, "CL", R * T * D, &CL));
    CHK(getbuf_t(h, "gates", R * 4 * D, &gates));
    CHK(getbuf_t(h, "sproj", R * 4 * D, &sproj));
    CHK(getbuf_t(h, "preh", R * 4 * D, &preh));
    CHK(getbuf_t(h, "eg", R * T, &eg));
    CHK(getbuf_t(h, "em", R * T, &em));
    CHK(getbuf_t(h, "elt", R * T, &elt));
    CHK(getbuf_t(h, "plt", h->opt.lt_mode == 0 ? R * T * D : 1, &plt));
    CHK(getbuf_t(h, "z1", R * E, &z1));
    CHK(getbuf_t(h, "a1", R * E, &a1));
    CHK(getbuf_t(h, "tz", R * E, &tz));
    CHK(getbuf_t(h, "logits", R * Vp, &lg));
    CHK(getbuf_t(h, "probs", R * Vp, &pr));
    CHK(getbuf_t(h, "nll", R, &nll));
    CHK(getbuf_t(h, "cost", (size_t)m, &cost));
    CHK(prepare_masks(h, t, m, &dp, &d1, &d2));

    // ---- prologue, once per batch
    h->gemm_seq = 0;
    BfWeights bw{};
    uint16_t* bemb = nullptr;
    {
        Prof pp(h, KC_PROLOGUE);
        HIPCHK(h, launch_embed(s, dx, w.Wemb, emb, (int)R, E, V, m));      // emb shifted one step (:613-617)
    }
    if (h->opt.precision == 1) {
        CHK(project_context(h, m, T, K, c.G, rawl, rawm, c));
        Prof pp(h, KC_PROLOGUE);
        CHK(bf16_weights(h, &bw, true));
        CHK(getbuf_t(h, "bx_emb", R * E, &bemb));
        HIPCHK(h, launch_cvt_bf16(s, emb, bemb, R * E));
        GemmBfArgs g = bf_args(bemb, E, bw.W, (int)R, 4 * D, E);       // x_ = emb.W + b (:334-335)
        g.bias = w.b; g.C = xproj; g.ldc = 4 * D;
        HIPCHK(h, gemm_bf(h, g));
    } else {
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;                                               // x_ = emb.W + b (:334-335): rides with the projections
        g.A = emb; g.lda = E; g.B = w.W; g.ldb = 4 * D; g.C = xproj; g.ldc = 4 * D;
        g.M = (int)R; g.N = 4 * D; g.K = E; g.bias = w.b;
        CHK(project_context(h, m, T, K, c.G, rawl, rawm, c, &g));
    }
    {
        Prof pp(h, KC_PROLOGUE);
        CHK(init_state(h, m, T, c.G, mG, mean, hs, cs));
    }

    // ---- the scan over caption positions (:495-512)
    FwdPanels pn{};
    const bool panels = use_panels(h, m);
    float *hpk[2] = {nullptr, nullptr}, *ctxpk = nullptr;
    if (panels) {
        CHK(pack_fwd_panels(h, &pn, false));
        // packed-A copies of the recurrent state (ping-pong: step s reads one and writes the other) and of ctx
        CHK(getbuf_t(h, "pk_h0", packed_rows_floats(m, D), &hpk[0]));
        CHK(getbuf_t(h, "pk_h1", packed_rows_floats(m, D), &hpk[1]));
        CHK(getbuf_t(h, "pk_ctx", packed_rows_floats(m, D), &ctxpk));
        HIPCHK(h, launch_pack_rows(s, hs, D, m, D, hpk[0]));
        if (m % 16) {     // the rows past m of the last m-tile are read (and ignored): keep them finite
            HIPCHK(h, hipMemsetAsync(hpk[1], 0, packed_rows_floats(m, D) * sizeof(float), s));
            HIPCHK(h, hipMemsetAsync(ctxpk, 0, packed_rows_floats(m, D) * sizeof(float), s));
        }
    }
    for (int st = 0; st < t; ++st) {
        const size_t r0 = (size_t)st * m;
        StepIO io{};
        io.M = m; io.T = T; io.K = K; io.c = c; io.vid = nullptr;
        io.h_prev = hs + r0 * D; io.c_prev = cs + r0 * D;
        io.sproj = sproj + r0 * 4 * D; io.preh = preh + r0 * 4 * D;
        io.xproj = xproj + r0 * 4 * D; io.emb = nullptr;
        io.dp = dp + r0 * 3 * D; io.mask = dmask + r0; io.d1 = d1 + r0 * D;
        io.alphal = al + r0 * T * K; io.CL = CL + r0 * T * D;
        io.eg = eg + r0 * T; io.em = em + r0 * T; io.elt = elt + r0 * T; io.plt = plt + (h->opt.lt_mode == 0 ? r0 * T * D : 0);
        io.alphag = ag + r0 * T; io.alpham = am + r0 * T; io.alphalt = alt + r0 * T;
        io.csum = csum + r0 * D; io.sel = sel + r0; io.ctx = ctx + r0 * D;
        io.h_out = hs + (r0 + m) * D; io.c_out = cs + (r0 + m) * D; io.gates = gates + r0 * 4 * D; io.hd = hd + r0 * D;
        io.pn = panels ? &pn : nullptr;
        io.h_prev_pk = hpk[st & 1]; io.h_out_pk = hpk[(st & 1) ^ 1]; io.ctx_pk = ctxpk;
        CHK(run_step(h, io));
    }

    // ---- readout over all (t*m) rows at once (:684-705), softmax and masked NLL (:708-715)
    if (h->opt.precision == 1) {   // same three GEMMs on the bf16 MFMA kernel; activations rounded to bf16 on the way in
        Prof pr_(h, KC_READOUT);
        uint16_t *bhd, *bctx, *ba1;
        CHK(getbuf_t(h, "bx_hd", R * D, &bhd));
        CHK(getbuf_t(h, "bx_ctx", R * D, &bctx));
        CHK(getbuf_t(h, "bx_a1", R * E, &ba1));
        HIPCHK(h, launch_cvt_bf16(s, hd, bhd, R * D));
        GemmBfArgs g = bf_args(bhd, D, bw.Wl1, (int)R, E, D);
        g.bias = w.bl1;
        if (h->opt.prev2out) { g.add = emb; g.ldadd = E; }
        if (h->opt.ctx2out) { g.C = z1; g.ldc = E; }
        else { g.act = 1; g.mul = d2; g.ldmul = E; g.Cb = ba1; g.ldcb = E; g.C = a1; g.ldc = E; }   // (a1 in fp32 too: backward)
        HIPCHK(h, gemm_bf(h, g));
        if (h->opt.ctx2out) {
            HIPCHK(h, launch_cvt_bf16(s, ctx, bctx, R * D));
            g = bf_args(bctx, D, bw.Wl2, (int)R, E, D);
            g.bias = w.bl2; g.add = z1; g.ldadd = E; g.act = 1; g.mul = d2; g.ldmul = E; g.Cb = ba1; g.ldcb = E;
            g.C = a1; g.ldc = E;                                            // (a1 in fp32 too: the backward pass reads it)
            HIPCHK(h, gemm_bf(h, g));
        }
        g = bf_args(ba1, E, bw.Wo, (int)R, Vp, E);
        g.bias = w.bo; g.C = lg; g.ldc = Vp;
        HIPCHK(h, gemm_bf(h, g));
    } else {
        Prof pr_(h, KC_READOUT);
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;      // z1 = (h*d1).Wl1 + bl1 [+ emb]
        g.A = hd; g.lda = D; g.B = w.Wl1; g.ldb = E; g.C = h->opt.ctx2out ? z1 : a1; g.ldc = E;
        g.M = (int)R; g.N = E; g.K = D; g.bias = w.bl1;
        if (h->opt.prev2out) { g.add = emb; g.ldadd = E; }
        if (!h->opt.ctx2out) { g.act = 1; g.mul = d2; g.ldmul = E; g.Cact = tz; g.ldcact = E; }
        HIPCHK(h, gemm_nn(h, g));
        if (h->opt.ctx2out) {  // a = tanh(ctx.Wl2 + bl2 + z1) * d2
            gemm_defaults(g); g.split = h->opt.precision != 0;
            g.A = ctx; g.lda = D; g.B = w.Wl2; g.ldb = E; g.C = a1; g.ldc = E;
            g.M = (int)R; g.N = E; g.K = D; g.bias = w.bl2; g.add = z1; g.ldadd = E; g.act = 1; g.mul = d2; g.ldmul = E;
            g.Cact = tz; g.ldcact = E;
            HIPCHK(h, gemm_nn(h, g));
        }
        gemm_defaults(g); g.split = h->opt.precision != 0;      // logit = a.Wo + bo
        g.A = a1; g.lda = E; g.B = w.Wo; g.ldb = Vp; g.C = lg; g.ldc = Vp;
        g.M = (int)R; g.N = Vp; g.K = E; g.bias = w.bo;
        HIPCHK(h, gemm_nn(h, g));
    }
    HIPCHK(h, launch_softmax_nll(s, lg, Vp, pr, Vp, dx, nll, nullptr, (int)R, V));
    HIPCHK(h, launch_cost(s, nll, dmask, cost, t, m));
    h->have_fwd = true; h->have_bwd = false;      // fresh logits; any earlier gradient belongs to another pass
    return STATTN_OK;
}

int stattn_get_forward(stattn_handle* h, float* cost, float* probs, float* alphal, float* alphag, float* alpham,
                       float* alphalt, float* logits) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_fwd) return fail(h, STATTN_ESTATE, "get_forward: no forward pass has run");
    if (logits && h->have_bwd)
        return fail(h, STATTN_ESTATE, "get_forward: the logits buffer holds d(loss)/d(logit) after stattn_backward; read logits before it");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t R = (size_t)h->t * h->m;
    const int T = h->T, K = h->K, V = h->V, Vp = h->Vp;
    if (cost) HIPCHK(h, hipMemcpyAsync(cost, findbuf(h, "cost"), (size_t)h->m * 4, hipMemcpyDeviceToHost, s));
    if (probs) HIPCHK(h, hipMemcpy2DAsync(probs, (size_t)V * 4, findbuf(h, "probs"), (size_t)Vp * 4, (size_t)V * 4, R, hipMemcpyDeviceToHost, s));
    if (logits) HIPCHK(h, hipMemcpy2DAsync(logits, (size_t)V * 4, findbuf(h, "logits"), (size_t)Vp * 4, (size_t)V * 4, R, hipMemcpyDeviceToHost, s));
    if (alphal) HIPCHK(h, hipMemcpyAsync(alphal, findbuf(h, "alphal"), R * T * K * 4, hipMemcpyDeviceToHost, s));
    if (alphag) HIPCHK(h, hipMemcpyAsync(alphag, findbuf(h, "alphag"), R * T * 4, hipMemcpyDeviceToHost, s));
    if (alpham) HIPCHK(h, hipMemcpyAsync(alpham, findbuf(h, "alpham"), R * T * 4, hipMemcpyDeviceToHost, s));
    if (alphalt) HIPCHK(h, hipMemcpyAsync(alphalt, findbuf(h, "alphalt"), R * T * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return STATTN_OK;
}

int stattn_get_states(stattn_handle* h, float* hs, float* cs, float* ctx) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_fwd) return fail(h, STATTN_ESTATE, "get_states: no forward pass has run");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t R = (size_t)h->t * h->m, D = h->D, m = h->m;
    if (hs) HIPCHK(h, hipMemcpyAsync(hs, findbuf(h, "hs") + m * D, R * D * 4, hipMemcpyDeviceToHost, s));
    if (cs) HIPCHK(h, hipMemcpyAsync(cs, findbuf(h, "cs") + m * D, R * D * 4, hipMemcpyDeviceToHost, s));
    if (ctx) HIPCHK(h, hipMemcpyAsync(ctx, findbuf(h, "ctx"), R * D * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return STATTN_OK;
}

// ---- backward pass, optimizer (model_attention.py:1129-1147, 1193-1203; common.py:178-195) ---------
int stattn_backward(stattn_handle* h, float nll_scale, float alpha_c) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_fwd) return fail(h, STATTN_ESTATE, "backward: no forward pass has run on the staged batch");
    HIPCHK(h, hipSetDevice(h->device));
    const int t = h->t, m = h->m, T = h->T, K = h->K, D = h->D, E = h->E, V = h->V, Vp = h->Vp, Fl = h->Fl, Fm = h->Fm;
    const Weights& w = h->w;
    hipStream_t s = h->stream;
    const size_t R = (size_t)t * m, MT = (size_t)m * T, MTK = MT * K;
    auto G_ = [&](const char* n) { return h->d_grads + h->params[h->pindex[n]].off; };

    // forward tensors
    int64_t* dx = (int64_t*)h->bufs[bcur(h, "x")].p;
    float *dmask = findbuf(h, bcur(h, "mask").c_str()), *Gc = findbuf(h, bcur(h, "G").c_str()), *rawl = findbuf(h, bcur(h, "rawl").c_str()), *rawm = findbuf(h, bcur(h, "rawm").c_str()),
          *L = findbuf(h, "L"), *Mo = findbuf(h, "Mo"), *PG = findbuf(h, "PG"), *PL = findbuf(h, "PL"), *PM = findbuf(h, "PM"),
          *LW = findbuf(h, "LW"), *mean = findbuf(h, "mean"), *emb = findbuf(h, "emb"), *hs = findbuf(h, "hs"),
          *cs = findbuf(h, "cs"), *hd = findbuf(h, "hd"), *ctx = findbuf(h, "ctx"), *csum = findbuf(h, "csum"),
          *sel = findbuf(h, "sel"), *al = findbuf(h, "alphal"), *ag = findbuf(h, "alphag"), *am = findbuf(h, "alpham"),
          *alt = findbuf(h, "alphalt"), *CL = findbuf(h, "CL"), *gates = findbuf(h, "gates"), *sproj = findbuf(h, "sproj"),
          *a1 = findbuf(h, "a1"), *tz = findbuf(h, "tz"), *lg = findbuf(h, "logits"), *pr = findbuf(h, "probs"),
          *dp = findbuf(h, "dp"), *d1 = findbuf(h, "d1"), *d2 = findbuf(h, "d2");

    // backward workspaces
    float *da, *dhd, *dctx_r, *demb, *rg, *rm, *rlt, *rl, *sqg, *sqm, *sqlt, *sql, *UT, *WcT, *WdT, *dpre, *dsproj, *dcsum,
          *dselpre, *deg, *dem, *delt, *del, *dplt, *dslp, *dc, *dhp0, *dhp1, *dctxP, *dhUP, *dhWP, *dPL, *dL, *dLW, *dPG,
          *dPM, *dMo, *pUl, *pUlt, *pUg, *pUm, *cpart, *ws, *dph0, *dpc0, *lossreg, *da_raw, *dsgp, *dsmp;
    const int KZ1 = 8, KZ2 = 16;
    const size_t WS = (size_t)16 << 20;
    CHK(getbuf_t(h, "b_da", R * E, &da));
    CHK(getbuf_t(h, "b_dhd", R * D, &dhd));
    CHK(getbuf_t(h, "b_dctx_r", R * D, &dctx_r));
    CHK(getbuf_t(h, "b_demb", R * E, &demb));
    CHK(getbuf_t(h, "b_rg", MT, &rg)); CHK(getbuf_t(h, "b_rm", MT, &rm)); CHK(getbuf_t(h, "b_rlt", MT, &rlt));
    CHK(getbuf_t(h, "b_rl", MTK, &rl));
    CHK(getbuf_t(h, "b_sqg", MT, &sqg)); CHK(getbuf_t(h, "b_sqm", MT, &sqm)); CHK(getbuf_t(h, "b_sqlt", MT, &sqlt));
    CHK(getbuf_t(h, "b_sql", MTK, &sql));
    CHK(getbuf_t(h, "b_UT", (size_t)4 * D * D, &UT));
    CHK(getbuf_t(h, "b_WcT", (size_t)4 * D * D, &WcT));
    CHK(getbuf_t(h, "b_WdT", (size_t)4 * D * D, &WdT));
    CHK(getbuf_t(h, "b_dpre", R * 4 * D, &dpre));
    CHK(getbuf_t(h, "b_dsproj", R * 4 * D, &dsproj));
    CHK(getbuf_t(h, "b_dcsum", R * D, &dcsum));
    CHK(getbuf_t(h, "b_dselpre", R, &dselpre));
    CHK(getbuf_t(h, "b_deg", R * T, &deg)); CHK(getbuf_t(h, "b_dem", R * T, &dem)); CHK(getbuf_t(h, "b_delt", R * T, &delt));
    CHK(getbuf_t(h, "b_del", R * T * K, &del));
    CHK(getbuf_t(h, "b_dplt", R * T * D, &dplt));
    CHK(getbuf_t(h, "b_dslp", MT * D, &dslp));
    CHK(getbuf_t(h, "b_dsgp", MT * D, &dsgp));
    CHK(getbuf_t(h, "b_dsmp", MT * D, &dsmp));
    CHK(getbuf_t(h, "b_da_raw", 3 * MT, &da_raw));
    CHK(getbuf_t(h, "b_dc", (size_t)m * D, &dc));
    CHK(getbuf_t(h, "b_dhp0", (size_t)m * D, &dhp0)); CHK(getbuf_t(h, "b_dhp1", (size_t)m * D, &dhp1));
    CHK(getbuf_t(h, "b_dctxP", (size_t)KZ1 * m * D, &dctxP));
    CHK(getbuf_t(h, "b_dhUP", (size_t)KZ1 * m * D, &dhUP));
    CHK(getbuf_t(h, "b_dhWP", (size_t)KZ2 * m * D, &dhWP));
    CHK(getbuf_t(h, "b_dPL", MTK * D, &dPL)); CHK(getbuf_t(h, "b_dL", MTK * D, &dL)); CHK(getbuf_t(h, "b_dLW", MTK * D, &dLW));
    CHK(getbuf_t(h, "b_dPG", MT * D, &dPG)); CHK(getbuf_t(h, "b_dPM", MT * D, &dPM)); CHK(getbuf_t(h, "b_dMo", MT * D, &dMo));
    CHK(getbuf_t(h, "b_pUl", MT * D, &pUl)); CHK(getbuf_t(h, "b_pUlt", MT * D, &pUlt));
    CHK(getbuf_t(h, "b_pUg", MT * D, &pUg)); CHK(getbuf_t(h, "b_pUm", MT * D, &pUm));
    CHK(getbuf_t(h, "b_cpart", (size_t)256 * (size_t)(Vp > 4 * D ? Vp : 4 * D), &cpart));
    CHK(getbuf_t(h, "b_ws", WS, &ws));
    CHK(getbuf_t(h, "b_dph0", (size_t)m * D, &dph0)); CHK(getbuf_t(h, "b_dpc0", (size_t)m * D, &dpc0));
    CHK(getbuf_t(h, "b_lossreg", 4, &lossreg));

    auto gemm = [&](bool tA, bool tB, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int Kd,
                    int accumulate, const float* add = nullptr, int ldadd = 0) -> hipError_t {
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = Kd;
        g.accumulate = accumulate; g.add = add; g.ldadd = ldadd;
        if (!add) { g.ws = ws; g.ws_floats = WS; }
        return launch_gemm(s, g, tA, tB);
    };

    // bias gradients (column sums) are collected and run as ONE batched launch pair at the end of the pass: their
    // sources stay untouched until then
    ColsumBatch csb{};
    float* cspart;
    CHK(getbuf_t(h, "b_cspart", COLSUM_BATCH_PART_FLOATS, &cspart));
#define CSADD(X, LD, ROWS, N, DST, ACC, RW)                                                                      \
    do { if (!colsum_batch_add(csb, X, LD, ROWS, N, DST, ACC, RW)) return fail(h, STATTN_EINVAL, "backward: colsum batch overflow"); } while (0)

    // A region [first, before) of the flat gradient buffer is final: run its collected bias sums, then (data
    // parallel with overlap) start summing it over the ranks on the side stream -- comm.cpp
    auto region_done = [&](const char* first, const char* before) -> int {
        if (csb.n) { HIPCHK(h, launch_colsum_batch(s, csb, cspart)); csb = ColsumBatch{}; }
        const size_t lo = h->params[h->pindex[first]].off;
        const size_t hi = before ? h->params[h->pindex[before]].off : h->nflat;
        return comm_reduce_range(h, lo, hi - lo);
    };
    comm_backward_begins(h);
    // every gradient array is written in full by its GEMM / column sum; only Wemb is written row-wise (the rows of the
    // words of this batch), so only that region is cleared (the padding between arrays was zeroed at creation)
    HIPCHK(h, hipMemsetAsync(G_("Wemb"), 0, h->params[h->pindex["ff_state_W"]].off * sizeof(float), s));

    // bf16 handle (mixed precision): the forward pass kept the region tensors L / PL / LW in bf16 and tanh(z) of the readout
    // only as a = tanh(z) * d2.  The backward pass is the fp32 one, evaluated at those stored activations: they are widened
    // (exactly) into fp32 buffers, tanh(z) is recovered from a and the dropout multiplier, everything else was fp32 anyway.
    if (h->opt.precision == 1) {
        float *L32, *PL32, *LW32;
        CHK(getbuf_t(h, "b_L32", MTK * D, &L32)); CHK(getbuf_t(h, "b_PL32", MTK * D, &PL32)); CHK(getbuf_t(h, "b_LW32", MTK * D, &LW32));
        HIPCHK(h, launch_cvt_f32(s, reinterpret_cast<const uint16_t*>(L), L32, MTK * D));
        HIPCHK(h, launch_cvt_f32(s, reinterpret_cast<const uint16_t*>(PL), PL32, MTK * D));
        HIPCHK(h, launch_cvt_f32(s, reinterpret_cast<const uint16_t*>(LW), LW32, MTK * D));
        L = L32; PL = PL32; LW = LW32;
        HIPCHK(h, launch_unmul(s, a1, d2, tz, R * E));
    }

    // lt_mode 0 ran CL.Wclt per step in the forward pass (the reference's summation order, :416).  Its derivative is the
    // same function as lt_mode 1's: <dplt.Wclt^T, L_k> = <dplt, L_k.Wclt> and sum_s CL_s^T.dplt_s = L^T.(sum_s alpha dplt_s),
    // so the backward pass uses the hoisted form for both: LW = L.Wclt is formed here once (27.9 GFLOP at C2).
    if (h->opt.lt_mode == 0) {
        CHK(getbuf_t(h, "LW", MTK * D, &LW));
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = L; g.lda = D; g.B = w.Wclt; g.ldb = D; g.C = LW; g.ldc = D; g.M = (int)MTK; g.N = D; g.K = D;
        HIPCHK(h, launch_gemm(s, g, false, false));
    }

    // ---- regulariser terms d/d alpha (same for every step) and its value (:1138-1147)
    const bool reg = alpha_c > 0.f;
    if (reg) {
        HIPCHK(h, launch_alpha_reg(s, ag, rg, sqg, t, MT, alpha_c / T));
        HIPCHK(h, launch_alpha_reg(s, am, rm, sqm, t, MT, alpha_c / T));
        HIPCHK(h, launch_alpha_reg(s, alt, rlt, sqlt, t, MT, alpha_c / T));
        HIPCHK(h, launch_alpha_reg(s, al, rl, sql, t, MTK, alpha_c / (T * K)));
        MultiSumArgs ms{};
        const float* srcs[4] = {sqg, sqm, sqlt, sql};
        for (int i = 0; i < 4; ++i) {
            ms.src[i] = srcs[i]; ms.n[i] = i < 3 ? MT : MTK; ms.dst[i] = lossreg + i;
            ms.scale[i] = i < 3 ? alpha_c / T : alpha_c / (T * K);
        }
        ms.count = 4;
        HIPCHK(h, launch_multi_sum(s, ms, cpart));
    } else {
        HIPCHK(h, hipMemsetAsync(lossreg, 0, 4 * sizeof(float), s));
    }

    // ---- softmax / NLL and readout (:687-715), all (t*m) rows at once
    HIPCHK(h, launch_dlogit(s, pr, Vp, dx, dmask, nll_scale, lg, Vp, (int)R, V, Vp));           // dlogit overwrites logits
    CSADD(lg, Vp, (int)R, Vp, G_("ff_logit_b"), 0, nullptr);
    HIPCHK(h, gemm(false, true, lg, Vp, w.Wo, Vp, da, E, (int)R, E, Vp, 0));                     // da = dlogit Wo^T
    HIPCHK(h, launch_tanh_bwd(s, da, tz, d2, da, R * E));                                        // dz (in place)
    float* dz = da;
    CSADD(dz, E, (int)R, E, G_("ff_logit_lstm_b"), 0, nullptr);
    if (h->opt.ctx2out) CSADD(dz, E, (int)R, E, G_("ff_logit_ctxglm_b"), 0, nullptr);
    {   // the three readout weight gradients (K = t*m rows) in one grouped launch: dWo = a^T dlogit (1504 tiles) carries
        // dWl1 = hd^T dz and dWl2 = ctx^T dz (128 tiles each, a split-K pass each on their own); likewise the two
        // input gradients dhd = dz Wl1^T, dctx = dz Wl2^T
        static const char* nogroup = getenv("STATTN_GEMM_NOGROUP");
        GemmArgs gw[3], gi[2];
        int nw = 0, ni = 0;
        auto set = [&](GemmArgs& q, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M_, int N_, int K_) {
            gemm_defaults(q); q.split = h->opt.precision != 0;
            q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.C = C; q.ldc = ldc; q.M = M_; q.N = N_; q.K = K_;
        };
        set(gw[nw++], a1, E, lg, Vp, G_("ff_logit_W"), Vp, E, Vp, (int)R);
        set(gw[nw++], hd, D, dz, E, G_("ff_logit_lstm_W"), E, D, E, (int)R);
        set(gi[ni++], dz, E, w.Wl1, E, dhd, D, (int)R, D, E);
        if (h->opt.ctx2out) {
            set(gw[nw++], ctx, D, dz, E, G_("ff_logit_ctxglm_W"), E, D, E, (int)R);
            set(gi[ni++], dz, E, w.Wl2, E, dctx_r, D, (int)R, D, E);
        }
        if (nogroup) {
            for (int i = 0; i < nw; ++i) HIPCHK(h, gemm(true, false, gw[i].A, gw[i].lda, gw[i].B, gw[i].ldb, gw[i].C, gw[i].ldc, gw[i].M, gw[i].N, gw[i].K, 0));
            for (int i = 0; i < ni; ++i) HIPCHK(h, gemm(false, true, gi[i].A, gi[i].lda, gi[i].B, gi[i].ldb, gi[i].C, gi[i].ldc, gi[i].M, gi[i].N, gi[i].K, 0));
        } else {
            HIPCHK(h, launch_gemm_group(s, gw, nw, true, false));
            HIPCHK(h, launch_gemm_group(s, gi, ni, false, true));
        }
    }
    CHK(region_done("ff_logit_lstm_W", nullptr));     // final before the reverse scan even starts

    // ---- transposed recurrent weights of the reverse scan: packed panels (row-panel kernels) or plain transposed
    // copies (skinny kernels)
    BwdPanels bp{};
    const bool panels = use_panels(h, m);
    int kz1 = KZ1, kz2 = KZ2;
    float *dpre_pk = nullptr, *dsproj_pk = nullptr;
    if (panels) {
        CHK(pack_bwd_panels(h, &bp));
        CHK(getbuf_t(h, "pk_dpre", packed_rows_floats(m, 4 * D), &dpre_pk));
        CHK(getbuf_t(h, "pk_dsproj", packed_rows_floats(m, 4 * D), &dsproj_pk));
        if (m % 16) {     // rows past m of the last m-tile are read (and ignored): keep them finite
            HIPCHK(h, hipMemsetAsync(dpre_pk, 0, packed_rows_floats(m, 4 * D) * sizeof(float), s));
            HIPCHK(h, hipMemsetAsync(dsproj_pk, 0, packed_rows_floats(m, 4 * D) * sizeof(float), s));
        }
        // K split so that the launch fills the chip: 2 D / 16 column tiles (dctx | dhU), D / 16 (dhW)
        kz1 = 256 / (2 * D / 16); kz1 = kz1 < 1 ? 1 : (kz1 > KZ1 ? KZ1 : kz1);
        kz2 = 256 / (D / 16); kz2 = kz2 < 1 ? 1 : (kz2 > KZ2 ? KZ2 : kz2);
    } else {
        HIPCHK(h, launch_transpose(s, w.U, 4 * D, UT, D, D, 4 * D));
        HIPCHK(h, launch_transpose(s, w.Wc, 4 * D, WcT, D, D, 4 * D));
        const float* Wd[4] = {w.Wdl, w.Wdg, w.Wdm, w.Wdlt};
        for (int i = 0; i < 4; ++i) HIPCHK(h, launch_transpose(s, Wd[i], D, WdT + (size_t)i * D * D, D, D, D));
    }

    // ---- reverse scan
    float* dhp_in = dhp0; float* dhp_out = dhp1;
    for (int st = t - 1; st >= 0; --st) {
        const size_t r0 = (size_t)st * m;
        const bool last = (st == t - 1);
        {
            LstmBwdArgs a{};
            a.dh_pass = dhp_in; a.dhU = dhUP; a.nU = kz1; a.dhW = dhWP; a.nW = kz2;
            a.dselpre = dselpre + (r0 + m); a.W_sel = h->opt.selector ? w.W_sel : nullptr;
            a.dhd = dhd + r0 * D; a.d1 = d1 + r0 * D; a.gates = gates + r0 * 4 * D;
            a.c_prev = cs + r0 * D; a.c_new = cs + (r0 + m) * D; a.mask = dmask + r0; a.dp = dp + r0 * 3 * D;
            a.dc = dc; a.dpre = dpre + r0 * 4 * D; a.dpre_pk = dpre_pk; a.dh_pass_out = dhp_out; a.M = m; a.D = D; a.last = last ? 1 : 0;
            HIPCHK(h, launch_lstm_bwd(s, a));
        }
        if (panels) {   // dctx = dpre.Wc^T and dhU = dpre.U^T as K-split partials
            PnArgs a{};
            a.M = m; a.nseg = 2; a.kz = kz1; a.part_stride = (size_t)m * D;
            for (int i = 0; i < 2; ++i) {
                PnSeg& sg = a.seg[i];
                pn_seg_defaults(sg);
                sg.npairs = 1; sg.p[0] = PnPair{dpre_pk, 4 * D, i == 0 ? bp.WcT : bp.UT, 4 * D, 1};
                sg.C = i == 0 ? dctxP : dhUP; sg.ldc = D; sg.N = D;
            }
            HIPCHK(h, launch_panel(s, a));
        } else {   // dctx = dpre.Wc^T and dhU = dpre.U^T as K-split partials
            SkArgs a{};
            a.M = m; a.nseg = 2; a.kz = KZ1; a.part_stride = (size_t)m * D;
            for (int i = 0; i < 2; ++i) {
                SkSeg& sg = a.seg[i];
                skinny_seg_defaults(sg);
                sg.npairs = 1; sg.p[0] = SkPair{dpre + r0 * 4 * D, i == 0 ? WcT : UT, 4 * D, D, 4 * D, 0};
                sg.C = i == 0 ? dctxP : dhUP; sg.ldc = D; sg.N = D;
            }
            HIPCHK(h, launch_skinny(s, a));
        }
        {
            TemporalBwdArgs a{};
            a.dctxP = dctxP; a.nP = kz1; a.dctx_r = h->opt.ctx2out ? dctx_r + r0 * D : nullptr;
            a.csum = csum + r0 * D; a.sel = sel + r0; a.G = Gc; a.Mo = Mo; a.CL = CL + r0 * T * D;
            a.rg = reg ? rg : nullptr; a.rm = reg ? rm : nullptr; a.rlt = reg ? rlt : nullptr;
            a.has_sel = h->opt.selector ? 1 : 0;
            a.dcsum = dcsum + r0 * D; a.dselpre = dselpre + r0; a.da_raw = da_raw; a.M = m; a.T = T; a.D = D;
            HIPCHK(h, launch_temporal_bwd(s, a));
        }
        {
            SpatialBwdArgs a{};
            a.PL = PL; a.L = L; a.LW = LW; a.sproj = sproj + r0 * 4 * D; a.ldsp = 4 * D;
            a.dcsum = dcsum + r0 * D; a.alphal = al + r0 * T * K;
            a.PG = PG; a.PM = PM; a.ag = ag + r0 * T; a.am = am + r0 * T; a.alt = alt + r0 * T; a.da_raw = da_raw;
            a.Ug = w.Ug; a.Um = w.Um;
            a.deg = deg + r0 * T; a.dem = dem + r0 * T; a.delt = delt + r0 * T; a.dsgp = dsgp; a.dsmp = dsmp;
            a.rl = reg ? rl : nullptr; a.Ul = w.Ul; a.Ult = w.Ult; a.blt = w.blt;
            a.dplt = dplt + r0 * T * D; a.del = del + r0 * T * K; a.dslp = dslp; a.M = m; a.T = T; a.K = K; a.D = D;
            HIPCHK(h, launch_spatial_bwd(s, a));
        }
        HIPCHK(h, launch_reduce_T(s, dslp, dsgp, dsmp, dplt + r0 * T * D, dsproj + r0 * 4 * D, 4 * D, m, T, D, dsproj_pk));
        if (panels) {   // dhW = [dsl|dsg|dsm|dslt] . [Wdl|Wdg|Wdm|Wdlt]^T
            PnArgs a{};
            a.M = m; a.nseg = 1; a.kz = kz2; a.part_stride = (size_t)m * D;
            PnSeg& sg = a.seg[0];
            pn_seg_defaults(sg);
            sg.npairs = 1; sg.p[0] = PnPair{dsproj_pk, 4 * D, bp.WdT, 4 * D, 1};
            sg.C = dhWP; sg.ldc = D; sg.N = D;
            HIPCHK(h, launch_panel(s, a));
        } else {   // dhW = [dsl|dsg|dsm|dslt] . [Wdl|Wdg|Wdm|Wdlt]^T
            SkArgs a{};
            a.M = m; a.nseg = 1; a.kz = KZ2; a.part_stride = (size_t)m * D;
            SkSeg& sg = a.seg[0];
            skinny_seg_defaults(sg);
            sg.npairs = 1; sg.p[0] = SkPair{dsproj + r0 * 4 * D, WdT, 4 * D, D, 4 * D, 0};
            sg.C = dhWP; sg.ldc = D; sg.N = D;
            HIPCHK(h, launch_skinny(s, a));
        }
        float* tmp = dhp_in; dhp_in = dhp_out; dhp_out = tmp;
    }
    // ---- deferred gradients.  Ordered by region of the flat buffer (= dict order) so that a data-parallel rank can
    // hand each region to the overlapped all-reduce as soon as it is final: decoder_* first (its 76 MB travel while
    // the F->D projection gradients -- the largest GEMM of the pass -- are computed), then ff_*, then Wemb.
    HIPCHK(h, launch_state0_bwd(s, dhp_in, dhUP, kz1, dhWP, kz2, dselpre, h->opt.selector ? w.W_sel : nullptr, dc, hs, cs,
                                dph0, dpc0, m, D));
    {
        CtxGradArgs a{};
        a.PL = PL; a.LW = LW; a.PG = PG; a.PM = PM; a.sproj = sproj; a.dcsum = dcsum; a.dplt = dplt;
        a.alphal = al; a.del = del; a.alt = alt; a.delt = delt; a.am = am; a.deg = deg; a.dem = dem;
        a.Ul = w.Ul; a.Ult = w.Ult; a.Ug = w.Ug; a.Um = w.Um; a.blt = w.blt;
        a.dPL = dPL; a.dL = dL; a.dLW = dLW; a.dPG = dPG; a.dPM = dPM; a.dMo = dMo;
        a.pUl = pUl; a.pUlt = pUlt; a.pUg = pUg; a.pUm = pUm; a.S = t; a.M = m; a.T = T; a.K = K; a.D = D;
        HIPCHK(h, launch_ctxgrad(s, a));
    }
    // -- region decoder_*
    CSADD(pUl, D, (int)MT, D, G_("decoder_Ul_att"), 0, nullptr);
    CSADD(pUlt, D, (int)MT, D, G_("decoder_Ult_att"), 0, nullptr);
    CSADD(pUg, D, (int)MT, D, G_("decoder_Ug_att"), 0, nullptr);
    CSADD(pUm, D, (int)MT, D, G_("decoder_Um_att"), 0, nullptr);
    {   // scalar biases of the four attention scorers (+ the selector bias): full sums, one launch
        MultiSumArgs ms{};
        const float* srcs[5] = {del, delt, deg, dem, dselpre};
        const size_t ns[5] = {R * T * K, R * T, R * T, R * T, R};
        const char* names[5] = {"decoder_cl_att", "decoder_clt_att", "decoder_cg_att", "decoder_cm_att", "decoder_b_sel"};
        ms.count = h->opt.selector ? 5 : 4;
        for (int i = 0; i < ms.count; ++i) { ms.src[i] = srcs[i]; ms.n[i] = ns[i]; ms.dst[i] = G_(names[i]); ms.scale[i] = 1.f; }
        HIPCHK(h, launch_multi_sum(s, ms, cpart));
    }
    CSADD(dsproj + 3 * (size_t)D, 4 * D, (int)R, D, G_("decoder_blt_att"), 0, nullptr);   // sum_{s,b,t} dplt = sum_{s,b} dslt (reduce_T already summed over t)
    if (h->opt.selector) {
        CSADD(hs, D, (int)R, D, G_("decoder_W_sel"), 0, dselpre);   // h_prev^T . dselpre
    }
    // attention pre-projections (:322-326) and the hoisted L.Wclt
    HIPCHK(h, gemm(true, false, L, D, dPL, D, G_("decoder_Wcl_att"), D, D, D, (int)MTK, 0));
    CSADD(dPL, D, (int)MTK, D, G_("decoder_bl_att"), 0, nullptr);
    HIPCHK(h, gemm(true, false, L, D, dLW, D, G_("decoder_Wclt_att"), D, D, D, (int)MTK, 0));
    CSADD(dPG, D, (int)MT, D, G_("decoder_bg_att"), 0, nullptr);
    CSADD(dPM, D, (int)MT, D, G_("decoder_bm_att"), 0, nullptr);
    // recurrent weights: one batched TN GEMM over all (t*m) rows each
    HIPCHK(h, gemm(true, false, hs, D, dpre, 4 * D, G_("decoder_U"), 4 * D, D, 4 * D, (int)R, 0));
    HIPCHK(h, gemm(true, false, ctx, D, dpre, 4 * D, G_("decoder_Wc"), 4 * D, D, 4 * D, (int)R, 0));
    {   // the six D x D weight gradients with a short K (frames or steps x rows): 256 tiles each -- alone they needed a
        // split-K pass each; as ONE grouped launch of 1536 tiles they fill the chip directly
        GemmArgs gq[6];
        auto tn = [&](GemmArgs& q, const float* A, int lda, const float* B, int ldb, float* C, int Kd) {
            gemm_defaults(q); q.split = h->opt.precision != 0;
            q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.C = C; q.ldc = D; q.M = D; q.N = D; q.K = Kd;
        };
        tn(gq[0], Gc, D, dPG, D, G_("decoder_Wcg_att"), (int)MT);
        tn(gq[1], Mo, D, dPM, D, G_("decoder_Wcm_att"), (int)MT);
        const char* names[4] = {"decoder_Wdl_att", "decoder_Wdg_att", "decoder_Wdm_att", "decoder_Wdlt_att"};
        for (int i = 0; i < 4; ++i) tn(gq[2 + i], hs, D, dsproj + (size_t)i * D, 4 * D, G_(names[i]), (int)R);
        static const char* nogroup = getenv("STATTN_GEMM_NOGROUP");
        if (nogroup) { for (const GemmArgs& q : gq) HIPCHK(h, gemm(true, false, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, 0)); }
        else HIPCHK(h, launch_gemm_group(s, gq, 6, true));
    }
    HIPCHK(h, gemm(true, false, emb, E, dpre, 4 * D, G_("decoder_W"), 4 * D, E, 4 * D, (int)R, 0));
    CSADD(dpre, 4 * D, (int)R, 4 * D, G_("decoder_b"), 0, nullptr);
    CHK(region_done("decoder_W", "ff_logit_lstm_W"));
    // -- region ff_*: initial state (:657-660), then back through tanh(ff_local), tanh(ff_motion) (:664-667)
    HIPCHK(h, gemm(true, false, mean, D, dph0, D, G_("ff_state_W"), D, D, D, m, 0));
    CSADD(dph0, D, m, D, G_("ff_state_b"), 0, nullptr);
    HIPCHK(h, gemm(true, false, mean, D, dpc0, D, G_("ff_memory_W"), D, D, D, m, 0));
    CSADD(dpc0, D, m, D, G_("ff_memory_b"), 0, nullptr);
    HIPCHK(h, gemm(false, true, dPL, D, w.Wcl, D, dL, D, (int)MTK, D, D, 1));
    HIPCHK(h, gemm(false, true, dLW, D, w.Wclt, D, dL, D, (int)MTK, D, D, 1));
    HIPCHK(h, launch_tanh_bwd(s, dL, L, nullptr, dL, MTK * D));
    HIPCHK(h, gemm(true, false, rawl, Fl, dL, D, G_("ff_local_W"), D, Fl, D, (int)MTK, 0));
    CSADD(dL, D, (int)MTK, D, G_("ff_local_b"), 0, nullptr);
    HIPCHK(h, gemm(false, true, dPM, D, w.Wcm, D, dMo, D, (int)MT, D, D, 1));
    HIPCHK(h, launch_tanh_bwd(s, dMo, Mo, nullptr, dMo, MT * D));
    HIPCHK(h, gemm(true, false, rawm, Fm, dMo, D, G_("ff_motion_W"), D, Fm, D, (int)MT, 0));
    CSADD(dMo, D, (int)MT, D, G_("ff_motion_b"), 0, nullptr);
    CHK(region_done("ff_state_W", "decoder_W"));
    // -- region Wemb: demb = dpre.W^T (+ dz through prev2out), scattered to the rows of Wemb (:613-617)
    // (dz is copied in first and the product accumulated onto it: without an `add` operand the 1920 x 512 x 4096 problem
    // -- 240 tiles of 64 x 64 -- may be cut along K, which fills the chip)
    if (h->opt.prev2out) HIPCHK(h, hipMemcpyAsync(demb, dz, R * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, gemm(false, true, dpre, 4 * D, w.W, 4 * D, demb, E, (int)R, E, 4 * D, h->opt.prev2out ? 1 : 0));
    {
        const EmbedPlan pl = device_embed_plan(h, h->cur_set);
        float* epart;
        CHK(getbuf_t(h, "b_embpart", (size_t)(pl.npieces > 0 ? pl.npieces : 1) * E, &epart));
        HIPCHK(h, launch_embed_bwd(s, pl, demb, G_("Wemb"), epart, E, m));
    }
    CHK(region_done("Wemb", "ff_state_W"));
#undef CSADD
    h->have_bwd = true;
    return STATTN_OK;
}

int stattn_get_loss(stattn_handle* h, float nll_scale, float decay_c, float* loss) {
    if (!h || !loss) return STATTN_EINVAL;
    if (!h->have_bwd) return fail(h, STATTN_ESTATE, "get_loss: call stattn_backward first (it evaluates the regulariser)");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    std::vector<float> cost(h->m);
    float regv4[4] = {0.f, 0.f, 0.f, 0.f}, p2 = 0.f;
    float *part, *sc;
    CHK(getbuf_t(h, "u_part", (size_t)1024, &part));
    CHK(getbuf_t(h, "u_scalar", (size_t)4, &sc));
    if (decay_c > 0.f) {   // decay_c * sum ||theta||^2 (:1130-1136); two_decay = 0 leaves the buffer unchanged
        HIPCHK(h, launch_decay_sumsq(s, h->d_params, h->d_params, 0.f, h->nflat, part, 1024));
        HIPCHK(h, launch_sum_all(s, part, 1024, sc, 1.f, 0));
        HIPCHK(h, hipMemcpyAsync(&p2, sc, sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipMemcpyAsync(cost.data(), findbuf(h, "cost"), (size_t)h->m * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(regv4, findbuf(h, "b_lossreg"), 4 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    double tot = 0.0;
    for (float c : cost) tot += c;
    const double regv = (double)regv4[0] + regv4[1] + regv4[2] + regv4[3];
    *loss = (float)(nll_scale * tot + regv + (double)decay_c * p2);
    return STATTN_OK;
}

int stattn_update(stattn_handle* h, float decay_c, float clip_c) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_bwd) return fail(h, STATTN_ESTATE, "update: no fresh gradient (call stattn_backward; one update per backward)");
    if (h->comm && h->comm_nranks > 1 && !h->grads_reduced)
        return fail(h, STATTN_ESTATE, "update: this rank belongs to a %d-rank communicator: call stattn_allreduce_grads first", h->comm_nranks);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (!h->d_rg2) {
        HIPCHK(h, hipMalloc((void**)&h->d_rg2, h->nflat * sizeof(float)));
        HIPCHK(h, hipMalloc((void**)&h->d_ru2, h->nflat * sizeof(float)));
        HIPCHK(h, hipMemsetAsync(h->d_rg2, 0, h->nflat * sizeof(float), s));
        HIPCHK(h, hipMemsetAsync(h->d_ru2, 0, h->nflat * sizeof(float), s));
    }
    float *part, *sc;
    CHK(getbuf_t(h, "u_part", (size_t)1024, &part));
    CHK(getbuf_t(h, "u_scalar", (size_t)4, &sc));
    // || g + 2 decay_c theta ||^2 (:1130-1136) in a fixed two-stage order; decay + clip + Adadelta in one pass
    HIPCHK(h, launch_decay_sumsq(s, h->d_grads, h->d_params, 2.f * decay_c, h->nflat, part, 1024));
    HIPCHK(h, launch_sum_all(s, part, 1024, sc, 1.f, 0));
    HIPCHK(h, launch_adadelta(s, h->d_params, h->d_grads, h->d_rg2, h->d_ru2, h->nflat, sc, clip_c, 2.f * decay_c));
    h->ck_proj = false; h->have_fwd = false; h->have_bwd = false;
    return STATTN_OK;
}

int stattn_reset_optimizer(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    if (h->d_rg2) {
        HIPCHK(h, hipMemsetAsync(h->d_rg2, 0, h->nflat * sizeof(float), h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_ru2, 0, h->nflat * sizeof(float), h->stream));
    }
    return STATTN_OK;
}

// ---- kernel-level entry points ------------------------------------------------------
int stattn_dbg_gemm(stattn_handle* h, int kind, int transA, int transB, int M, int N, int K, float alpha,
                    const float* A, const float* B, const float* bias, const float* add, int act, float* C) {
    if (!h || !A || !B || !C || M <= 0 || N <= 0 || K <= 0) return fail(h, STATTN_EINVAL, "dbg_gemm: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    HIPCHK(h, hipStreamSynchronize(s));
    float *dA, *dB, *dC, *dbias, *dadd;
    CHK(getbuf_t(h, "dbg_A", (size_t)M * K, &dA));
    CHK(getbuf_t(h, "dbg_B", (size_t)K * N, &dB));
    CHK(getbuf_t(h, "dbg_C", (size_t)M * N, &dC));
    CHK(getbuf_t(h, "dbg_bias", (size_t)N, &dbias));
    CHK(getbuf_t(h, "dbg_add", (size_t)M * N, &dadd));
    HIPCHK(h, hipMemcpyAsync(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dB, B, (size_t)K * N * 4, hipMemcpyHostToDevice, s));
    if (bias) HIPCHK(h, hipMemcpyAsync(dbias, bias, (size_t)N * 4, hipMemcpyHostToDevice, s));
    if (add) HIPCHK(h, hipMemcpyAsync(dadd, add, (size_t)M * N * 4, hipMemcpyHostToDevice, s));
    if (kind == 0 || kind == 4) {
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.split = kind == 4;
        g.A = dA; g.lda = transA ? M : K; g.B = dB; g.ldb = transB ? K : N; g.C = dC; g.ldc = N;
        g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.bias = bias ? dbias : nullptr;
        if (add) { g.add = dadd; g.ldadd = N; }
        g.act = act;
        if (g.split && !gemm_split_supported(g, transA != 0, transB != 0))
            return fail(h, STATTN_EINVAL, "split kernel: N % 128 == 0, k-contiguous operands 16-byte aligned");
        hipError_t e = launch_gemm(s, g, transA != 0, transB != 0);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_gemm: %s", hipGetErrorString(e));
    } else if (kind == 2) {
        // bf16-MFMA kernel: operands rounded to bf16 on the device, B kept k-contiguous ([N][K])
        if (transA || alpha != 1.f || K % 8 != 0) return fail(h, STATTN_EINVAL, "bf16 kernel: no transA, alpha must be 1, K % 8 == 0");
        uint16_t *bA, *bB;
        CHK(getbuf_t(h, "dbg_bA", (size_t)M * K, &bA));
        CHK(getbuf_t(h, "dbg_bB", (size_t)K * N, &bB));
        HIPCHK(h, launch_cvt_bf16(s, dA, bA, (size_t)M * K));
        if (transB) HIPCHK(h, launch_cvt_bf16(s, dB, bB, (size_t)K * N));
        else HIPCHK(h, launch_cvt_bf16_t(s, dB, N, bB, K, K, N));
        GemmBfArgs g{};
        g.A = bA; g.lda = K; g.B = bB; g.ldb = K; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
        g.bias = bias ? dbias : nullptr;
        if (add) { g.add = dadd; g.ldadd = N; }
        g.act = act; g.rowgroup = 1;
        hipError_t e = launch_gemm_bf16(s, g);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_gemm: %s", hipGetErrorString(e));
    } else if (kind == 3) {
        // row-panel kernel: B repacked on the device (transB: the operand is B^T, packed straight from B [N][K])
        if (transA || alpha != 1.f || !panel_supported(M) || N % 16 || K % 16)
            return fail(h, STATTN_EINVAL, "panel kernel: no transA, alpha must be 1, M <= 512, N and K multiples of 16");
        float* P;
        CHK(getbuf_t(h, "dbg_P", (size_t)K * N, &P));
        CHK(pack(h, dB, transB ? K : N, transB ? 1 : 0, K, N / 16, PN_COLS_PLAIN, P));
        PnArgs a{};
        a.M = M; a.nseg = 1;
        PnSeg& sg = a.seg[0];
        pn_seg_defaults(sg);
        sg.npairs = 1; sg.p[0] = PnPair{dA, K, P, K};
        sg.C = dC; sg.ldc = N; sg.N = N; sg.bias = bias ? dbias : nullptr;
        if (add) { sg.add = dadd; sg.ldadd = N; }
        sg.act = act;
        hipError_t e = launch_panel(s, a);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_gemm: %s", hipGetErrorString(e));
    } else {
        if (transA || transB) return fail(h, STATTN_EINVAL, "skinny kernel has no transposed variants");
        SkArgs a{};
        a.M = M; a.nseg = 1;
        SkSeg& sg = a.seg[0];
        skinny_seg_defaults(sg);
        sg.npairs = 1; sg.p[0] = SkPair{dA, dB, K, N, K, 0};
        sg.C = dC; sg.ldc = N; sg.N = N; sg.bias = bias ? dbias : nullptr;
        if (add) { sg.add = dadd; sg.ldadd = N; }
        sg.act = act; sg.scale = 1.f;
        if (alpha != 1.f) return fail(h, STATTN_EINVAL, "skinny kernel: alpha must be 1");
        hipError_t e = launch_skinny(s, a);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_gemm: %s", hipGetErrorString(e));
    }
    HIPCHK(h, hipMemcpyAsync(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return STATTN_OK;
}

int stattn_dbg_time_gemm(stattn_handle* h, int transA, int transB, int M, int N, int K, int iters, float* ms_per_launch) {
    if (!h || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || !ms_per_launch) return fail(h, STATTN_EINVAL, "dbg_time_gemm: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    float *dA, *dB, *dC;
    CHK(getbuf_t(h, "dbg_A", (size_t)M * K, &dA));
    CHK(getbuf_t(h, "dbg_B", (size_t)K * N, &dB));
    CHK(getbuf_t(h, "dbg_C", (size_t)M * N, &dC));
    // uniform [-1,1) operands: full-range signs (never time a GEMM on zeros -- DVFS inflates the clock)
    HIPCHK(h, launch_uniform(s, dA, (size_t)M * K, 11, 1));
    HIPCHK(h, launch_uniform(s, dB, (size_t)K * N, 11, 2));
    GemmArgs g;
    gemm_defaults(g); g.split = h->opt.precision != 0;
    g.A = dA; g.lda = transA ? M : K; g.B = dB; g.ldb = transB ? K : N; g.C = dC; g.ldc = N;
    g.M = M; g.N = N; g.K = K;
    {   // same split-K workspace the backward pass hands to its weight-gradient GEMMs
        float* ws;
        CHK(getbuf_t(h, "b_ws", (size_t)16 << 20, &ws));
        g.ws = ws; g.ws_floats = (size_t)16 << 20;
    }
    for (int i = 0; i < 2; ++i) HIPCHK(h, launch_gemm(s, g, transA != 0, transB != 0));
    hipEvent_t a, b;
    HIPCHK(h, hipEventCreate(&a)); HIPCHK(h, hipEventCreate(&b));
    HIPCHK(h, hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) HIPCHK(h, launch_gemm(s, g, transA != 0, transB != 0));
    HIPCHK(h, hipEventRecord(b, s));
    HIPCHK(h, hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = ms / iters;
    return STATTN_OK;
}

int stattn_dbg_time_gemm_bf16(stattn_handle* h, int M, int N, int K, int tile, int iters, float* ms_per_launch) {
    if (!h || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || !ms_per_launch) return fail(h, STATTN_EINVAL, "dbg_time_gemm_bf16: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    float *dA, *dB;
    uint16_t *bA, *bB, *bC;
    CHK(getbuf_t(h, "dbg_A", (size_t)M * K, &dA));
    CHK(getbuf_t(h, "dbg_B", (size_t)K * N, &dB));
    CHK(getbuf_t(h, "dbg_bA", (size_t)M * K, &bA));
    CHK(getbuf_t(h, "dbg_bB", (size_t)K * N, &bB));
    CHK(getbuf_t(h, "dbg_bC", (size_t)M * N, &bC));
    HIPCHK(h, launch_uniform(s, dA, (size_t)M * K, 11, 1));
    HIPCHK(h, launch_uniform(s, dB, (size_t)K * N, 11, 2));
    HIPCHK(h, launch_cvt_bf16(s, dA, bA, (size_t)M * K));
    HIPCHK(h, launch_cvt_bf16(s, dB, bB, (size_t)K * N));
    GemmBfArgs g{};
    g.A = bA; g.lda = K; g.B = bB; g.ldb = K; g.Cb = bC; g.ldcb = N; g.M = M; g.N = N; g.K = K; g.rowgroup = 1; g.tile = tile;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = launch_gemm_bf16(s, g);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_time_gemm_bf16: %s", hipGetErrorString(e));
    }
    hipEvent_t a, b;
    HIPCHK(h, hipEventCreate(&a)); HIPCHK(h, hipEventCreate(&b));
    HIPCHK(h, hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) HIPCHK(h, launch_gemm_bf16(s, g));
    HIPCHK(h, hipEventRecord(b, s));
    HIPCHK(h, hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = ms / iters;
    return STATTN_OK;
}

long stattn_dbg_counter(const stattn_handle* h, int which) {
    if (!h) return -1;
    if (which == 0) return h->beam_graph_replays;
    return -1;
}

int stattn_dbg_time_skinny(stattn_handle* h, int M, int N, int K, int nseg, int variant, int iters, float* ms_per_launch) {
    if (!h || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || nseg < 1 || nseg > 6 || !ms_per_launch)
        return fail(h, STATTN_EINVAL, "dbg_time_skinny: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    float *dA, *dB, *dC;
    CHK(getbuf_t(h, "dbg_A", (size_t)M * K, &dA));
    CHK(getbuf_t(h, "dbg_B", (size_t)K * N * nseg, &dB));
    CHK(getbuf_t(h, "dbg_C", (size_t)M * N * nseg, &dC));
    HIPCHK(h, launch_uniform(s, dA, (size_t)M * K, 11, 1));
    HIPCHK(h, launch_uniform(s, dB, (size_t)K * N * nseg, 11, 2));
    SkArgs a{};
    a.M = M; a.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
        SkSeg& sg = a.seg[i];
        skinny_seg_defaults(sg);
        sg.npairs = 1; sg.p[0] = SkPair{dA, dB + (size_t)i * K * N, K, N, K, (variant & 16) ? (size_t)K * 64 : 0};
        sg.C = dC + (size_t)i * N; sg.ldc = N * nseg; sg.N = N;
    }
    for (int i = 0; i < 2; ++i) HIPCHK(h, launch_skinny(s, a));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(h, launch_skinny(s, a));
    HIPCHK(h, hipEventRecord(e1, s));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = ms / iters;
    return STATTN_OK;
}

int stattn_set_profiling(stattn_handle* h, int enable) {
    if (!h) return STATTN_EINVAL;
    prof_collect(h);
    h->profiling = enable != 0;
    for (int i = 0; i < KC_COUNT + KC_GEMM_SEQ; ++i) { h->k_ms[i] = 0; h->k_n[i] = 0; }
    return STATTN_OK;
}
int stattn_get_kernel_ms(stattn_handle* h, int which, float* ms_avg, int* launches) {
    if (!h || which < 0 || which >= KC_COUNT + KC_GEMM_SEQ || !ms_avg) return STATTN_EINVAL;
    prof_collect(h);
    *ms_avg = h->k_n[which] ? (float)(h->k_ms[which] / h->k_n[which]) : 0.f;
    if (launches) *launches = h->k_n[which];
    return STATTN_OK;
}

}  // extern "C"
