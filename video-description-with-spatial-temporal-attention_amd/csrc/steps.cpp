// Shared host-side building blocks of libstattn.so (declared in steps.h).  Orchestration only: every number is computed
// by the hand-written gfx950 kernels (gemm*.hip, panel.hip, skinny.hip, attn.hip, misc.hip).
#include <cstdlib>
#include "steps.h"

namespace stattn_detail {

// ---- profiling helpers -------------------------------------------------------------

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
    static const char* nogroup = sw_product("STATTN_GEMM_NOGROUP");     // A/B switch for tools: one launch per problem
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

// ---- bf16 path (precision = 1) ----------------------------------------------------------------------
// k-contiguous bf16 shadows ([N][K]) of the weight matrices the bf16 GEMMs read.  Rebuilt from the fp32 master
// copy at every use (once per minibatch / once per decoded video: ~35 M elements, tens of microseconds), so they can
// never go stale whichever way the parameters were written (set_param, update, RCCL broadcast into the flat buffer).

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
    // Wcl and Wclt share one buffer, [Wcl^T ; Wclt^T] stacked: PL = L.Wcl + bl and LW = L.Wclt are ONE launch over N = 2 D
    // columns (GemmBfArgs::n_split) -- 5 whole rounds of 256 x 256 tiles at configs[3] instead of twice 2.5
    uint16_t* wcl2 = nullptr;
    if (!readout) CHK(getbuf_t(h, "bw_Wcl_Wclt", (size_t)2 * D * D, &wcl2));
    // readout layer 1 with ctx2out: a = tanh([hd | ctx] . [Wl1 ; Wl2] + bl1 + bl2 ...) is ONE K-concatenated problem (K = 2 D):
    // Wl1^T and Wl2^T side by side in rows of 2 D
    b->Wl12 = nullptr;
    if (readout && w.Wl2) {
        CHK(getbuf_t(h, "bw_Wl12", (size_t)2 * D * E, &b->Wl12));
        HIPCHK(h, launch_cvt_bf16_t(s, w.Wl1, E, b->Wl12, 2 * D, D, E));
        HIPCHK(h, launch_cvt_bf16_t(s, w.Wl2, E, b->Wl12 + D, 2 * D, D, E));
    }
    for (const Item& it : items) {
        if (it.ro != readout || !it.src) continue;      // absent parameter (ff_logit_ctxglm without ctx2out)
        if (it.dst == &b->Wcl) *it.dst = wcl2;
        else if (it.dst == &b->Wclt) *it.dst = wcl2 + (size_t)D * D;
        else CHK(getbuf_t(h, it.name, (size_t)it.K * it.N, it.dst));
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
// several independent bf16 problems: ONE launch of the 256 x 256 kernel when every problem qualifies (one profiling slot, like the
// fp32 path's grouped launches), else one launch each
hipError_t gemm_bf_group(stattn_handle* h, const GemmBfArgs* gs, int n) {
    static const char* nogroup = sw_product("STATTN_GEMM_NOGROUP");           // A/B switch for tools
    bool ok = n > 1 && n <= GEMM_BF_GROUP_MAX && !nogroup;
    long tiles = 0;
    for (int i = 0; i < n && ok; ++i) { ok = gemm_bf16_8ph_supported(gs[i]); tiles += (long)((gs[i].M + 255) / 256) * (gs[i].N / 256); }
    if (!ok || tiles < 128) {
        for (int i = 0; i < n; ++i) { const hipError_t e = gemm_bf(h, gs[i]); if (e != hipSuccess) return e; }
        return hipSuccess;
    }
    Prof pr(h, KC_GEMM_NN);
    const int seq = h->gemm_seq++;
    Prof one(h, KC_COUNT + (seq < KC_GEMM_SEQ ? seq : 0), seq < KC_GEMM_SEQ);
    return launch_gemm_bf16_8ph_group(h->stream, gs, n);
}
GemmBfArgs bf_args(const uint16_t* A, int lda, const uint16_t* B, int M, int N, int Kd) {
    GemmBfArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = Kd; g.M = M; g.N = N; g.K = Kd; g.rowgroup = 1;
    return g;
}

// project_context with bf16 operands: L / PL / LW are written as bf16 INTO the (fp32-sized) buffers of CtxPtrs
static int project_context_bf16(stattn_handle* h, int nv, int T, int K, const float* ctxg, const float* ctxl, const float* ctxm,
                         const CtxPtrs& c, const GemmBfArgs* extra) {
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
    // first launch: everything that needs raw inputs only, the longest K first (gemm_bf_group): L = tanh(ctxl . ff_local_W + b),
    // M = tanh(ctxm . ff_motion_W + b), pctxg_ and -- training -- the x projection the caller hands over
    GemmBfArgs g1[4];
    int n1 = 0;
    g1[n1] = bf_args(xl, h->Fl, bw.ff_local, (int)nl, D, h->Fl);
    g1[n1].bias = w.ff_local_b; g1[n1].act = 1; g1[n1].Cb = Lb; g1[n1].ldcb = D; ++n1;
    g1[n1] = bf_args(xm, h->Fm, bw.ff_motion, (int)nf, D, h->Fm);
    g1[n1].bias = w.ff_motion_b; g1[n1].act = 1; g1[n1].C = c.Mo; g1[n1].ldc = D; g1[n1].Cb = mo; g1[n1].ldcb = D; ++n1;
    g1[n1] = bf_args(xg, D, bw.Wcg, (int)nf, D, D);
    g1[n1].bias = w.bg; g1[n1].C = c.PG; g1[n1].ldc = D; ++n1;
    if (extra) g1[n1++] = *extra;
    HIPCHK(h, gemm_bf_group(h, g1, n1));
    // second launch: what needs L / M.  PL = L.Wcl + bl and LW = L.Wclt are one problem over N = 2 D columns with two outputs
    static const char* nofuse = sw_tool("STATTN_BF16_NOFUSE");                 // A/B switch for tools
    const bool fuse = D % 256 == 0 && !nofuse;
    // (pctxm_ is NOT grouped with them: at configs[3] PL | LW is exactly 5 rounds of 256 tiles and 40 more tiles of the same length
    // would add a sixth for everybody -- 190 against 162 + 19 us measured)
    GemmBfArgs g2[2];
    int n2 = 0;
    if (fuse) {
        g2[n2] = bf_args(Lb, D, bw.Wcl, (int)nl, 2 * D, D);                   // pctxl_ | LW = L . [Wcl | Wclt]  (+ bl on the first half)
        g2[n2].bias = w.bl; g2[n2].Cb = reinterpret_cast<uint16_t*>(c.PL); g2[n2].ldcb = D;
        g2[n2].n_split = D; g2[n2].Cb2 = reinterpret_cast<uint16_t*>(c.LW); ++n2;
    } else {
        g2[n2] = bf_args(Lb, D, bw.Wcl, (int)nl, D, D);                       // pctxl_
        g2[n2].bias = w.bl; g2[n2].Cb = reinterpret_cast<uint16_t*>(c.PL); g2[n2].ldcb = D; ++n2;
        g2[n2] = bf_args(Lb, D, bw.Wclt, (int)nl, D, D);                      // LW = L . Wclt
        g2[n2].Cb = reinterpret_cast<uint16_t*>(c.LW); g2[n2].ldcb = D; ++n2;
    }
    HIPCHK(h, gemm_bf_group(h, g2, n2));
    GemmBfArgs g = bf_args(mo, D, bw.Wcm, (int)nf, D, D);                     // pctxm_
    g.bias = w.bm; g.C = c.PM; g.ldc = D;
    HIPCHK(h, gemm_bf(h, g));
    return STATTN_OK;
}

// `extra`: one more independent plain GEMM that rides in the first launch (training: the x projection), or null.
int project_context(stattn_handle* h, int nv, int T, int K, const float* ctxg, const float* ctxl, const float* ctxm,
                    const CtxPtrs& c, const GemmArgs* extra, const GemmBfArgs* extra_bf) {
    if (h->opt.precision == 1) return project_context_bf16(h, nv, T, K, ctxg, ctxl, ctxm, c, extra_bf);
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
    // A handful of videos (per-video decode, metrics.py:121-135: ONE video = 208 region rows at configs[0]): the F -> D
    // projections are a few dozen tiles with K = 4096 -- in the grouped launch 48 workgroups walk 128 k-tiles each (98 us, a
    // fifth of the set-up of a decode call).  Cut along K instead: the split-K path of launch_gemm (partial tiles in a
    // workspace, summed in a fixed order by the reduction that applies the bias / tanh epilogue).
    const size_t region_rows = (size_t)nv * T * K;
    if (region_rows <= 1024 && h->opt.precision == 0 && !extra) {
        const size_t ws_floats = (size_t)8 * region_rows * D;
        float* ws;
        CHK(getbuf_t(h, "pc_ws", ws_floats, &ws));
        for (int i = 0; i < n1; ++i) {
            if (g1[i].K >= 2048) { g1[i].ws = ws; g1[i].ws_floats = ws_floats; }
            HIPCHK(h, gemm_nn(h, g1[i]));
        }
    } else {
        CHK(gemm_group(h, g1, n1));
    }
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

// min_rows: the training scan repacks the panels every pass and uses them from 17 rows up; beam search packs once per
// call and uses them for any batch (at 4 rows the 16-column panels still give 128+ workgroups where the 64-column
// kernels give 32)
bool use_panels(const stattn_handle* h, int M, int min_rows) {
    static const char* off = sw_product("STATTN_NO_PANELS");       // A/B switch for tools
    // (the kernels take up to 512 rows; past 256 the 64-column kernels, which split the rows over workgroups, were as fast when the rule
    //  was set -- before the wide kernels of panelw.hip existed.  The evaluation workload says otherwise (52 videos = 260 rows: 1609
    //  videos/s against 2597 at 255 rows), so the bound is a tool switch for the A/B: STATTN_PANEL_MAX_ROWS=512 in a tools build)
    static const char* mx = sw_tool("STATTN_PANEL_MAX_ROWS");
    const int max_rows = mx ? atoi(mx) : 256;
    return !off && M >= min_rows && M <= max_rows && panel_supported(M) && h->D % 16 == 0 && h->E % 16 == 0;
}

// repacking jobs are collected per pass and run as one launch (pack_flush)
static thread_local PackBatch g_packs{};
int pack(stattn_handle* h, const float* W, int ldw, int src_t, int K, int ntiles, int cols, float* dst, int S_total, int s_off) {
    PackJob jb{};
    jb.W = W; jb.ldw = ldw; jb.src_t = src_t; jb.K = K; jb.ntiles = ntiles; jb.cols = cols; jb.D = h->D;
    jb.dst = dst; jb.S_total = S_total ? S_total : K / 16; jb.s_off = s_off;
    if (g_packs.n == PACK_BATCH_MAX) CHK(pack_flush(h));
    if (!pack_batch_add(g_packs, jb)) { g_packs = PackBatch{}; return fail(h, STATTN_EINVAL, "pack: bad repacking job"); }
    return STATTN_OK;
}
int pack_flush(stattn_handle* h) {
    const PackBatch b = g_packs;
    g_packs = PackBatch{};
    HIPCHK(h, launch_pack_batch(h->stream, b));
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
    return pack_flush(h);
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
    return pack_flush(h);
}


// one decoder timestep: _step, model_attention.py:366-459
int run_step(stattn_handle* h, const StepIO& io) {
    const int D = h->D, E = h->E;
    const Weights& w = h->w;
    // h.U (+ x_) is not needed before the LSTM kernel: when the attention launch can carry it as a rider (extra
    // workgroups on the idle matrix cores of the HBM-bound kernel), the state-projection launch -- which IS on the
    // critical path -- shrinks to h.[Wdl|Wdg|Wdm|Wdlt]
    SpatialArgs sa{};
    sa.bf16 = h->opt.precision == 1; sa.group = h->opt.precision != 1 ? io.group : 0;
    sa.M = io.M; sa.T = io.T; sa.K = io.K; sa.D = D;
    const bool rider = io.pn && io.h_prev_pk && io.M <= 64 && !io.skip_hproj && spatial_rider_supported(sa);
    const int ldp = io.ldproj ? io.ldproj : 4 * D;
    if (io.phase != 2) { h->path_fwd_rider += rider; h->path_fwd_panel += io.pn != nullptr; }
    if (io.upd && io.phase != 1) return fail(h, STATTN_EINVAL, "run_step: an update rider needs phase 1 of a step");
    if (io.phase == 2 || io.skip_hproj) {
    } else if (io.pn) {   // state projections on the row-panel kernel: one launch, every weight byte streamed once
        Prof pr(h, KC_HPROJ);
        PnArgs a{};
        a.M = io.M; a.nseg = rider ? 1 : 2;
        PnSeg& s0 = a.seg[0];
        pn_seg_defaults(s0);
        const PnPair hA = io.h_prev_pk ? PnPair{io.h_prev_pk, D, nullptr, D, 1} : PnPair{io.h_prev, D, nullptr, D, 0};
        s0.npairs = 1; s0.p[0] = hA; s0.p[0].P = io.pn->Wd;
        s0.C = io.sproj; s0.ldc = ldp; s0.N = 4 * D;           // [Wdl | Wdg | Wdm | Wdlt]: 4 x D/16 consecutive tiles
        PnSeg& s1 = a.seg[1];
        pn_seg_defaults(s1);
        s1.npairs = 1; s1.p[0] = hA; s1.p[0].P = io.pn->U;
        s1.C = io.preh; s1.ldc = ldp; s1.N = 4 * D;
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
            s.C = io.sproj + (size_t)i * D; s.ldc = ldp; s.N = D;
        }
        SkSeg& s = a.seg[4];
        skinny_seg_defaults(s);
        s.npairs = 1; s.p[0] = SkPair{io.h_prev, w.U, D, 4 * D, D, 0};
        s.C = io.preh; s.ldc = ldp; s.N = 4 * D;
        if (io.xproj) { s.add = io.xproj; s.ldadd = 4 * D; }
        HIPCHK(h, launch_skinny(h->stream, a));
    }
    if (io.phase != 2) {   // spatial attention + frame scores (:371-383, 389-397, 402-410, and 415-424 in lt_mode 1)
        Prof pr(h, KC_SPATIAL);
        SpatialArgs a{};
        a.PL = io.c.PL; a.L = io.c.L; a.LW = h->opt.lt_mode == 1 ? io.c.LW : nullptr;
        a.bf16 = h->opt.precision == 1;
        a.PG = io.c.PG; a.PM = io.c.PM; a.vid = io.vid;
        a.group = h->opt.precision != 1 ? io.group : 0;
        a.sproj = io.sproj; a.ldsp = ldp;
        if (rider) {   // preh = h.U (+ x_): 4D / 16 tiles, one K-slice
            RiderArgs& r = a.rider;
            r.A = io.h_prev_pk; r.P = io.pn->U; r.C = io.preh; r.ldc = ldp;
            r.add = io.xproj; r.ldadd = 4 * D;
            r.M = io.M; r.N = 4 * D; r.K = D; r.kz = 1; r.part_stride = 0; r.nblocks = 4 * D / 16;
        }
        a.Ul = w.Ul; a.cl = w.cl; a.Ug = w.Ug; a.cg = w.cg; a.Um = w.Um; a.cm = w.cm;
        a.Ult = w.Ult; a.clt = w.clt; a.blt = w.blt;
        a.alphal = io.alphal; a.CL = io.CL; a.eg = io.eg; a.em = io.em; a.elt = io.elt;
        a.M = io.M; a.T = io.T; a.K = io.K; a.D = D;
        if (io.upd && !spatial_update_supported(a)) return fail(h, STATTN_EINVAL, "run_step: this attention launch cannot carry the update");
        if (io.upd) h->upd_rowwg_last = spatial_update_row_workgroups(a, *io.upd);      // what the path counter reports (api_sampler.cpp)
        HIPCHK(h, launch_spatial(h->stream, a, io.upd));
    }
    if (io.phase == 1) return STATTN_OK;
    if (h->opt.lt_mode == 0) {   // pctxlt = CL.Wclt + blt + pstatelt, tanh, . Ult  (:416-422) as one MFMA GEMM
        Prof pr(h, KC_LTGEMM);
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.A = io.CL; g.lda = D; g.B = w.Wclt; g.ldb = D; g.C = io.plt; g.ldc = D;
        g.M = io.M * io.T; g.N = D; g.K = D; g.bias = w.blt;
        g.rowadd = io.sproj + 3 * (size_t)D; g.ldrow = ldp; g.rowgroup = io.T; g.act = 1;
        HIPCHK(h, launch_gemm(h->stream, g, false, false));
        HIPCHK(h, launch_rowdot(h->stream, io.plt, D, w.Ult, w.clt, io.elt, io.M * io.T, D));
    }
    {   // three temporal softmaxes, weighted sums, sum-fusion, selector gate (:398-399, 411-412, 425-435)
        Prof pr(h, KC_TEMPORAL);
        TemporalArgs a{};
        a.eg = io.eg; a.em = io.em; a.elt = io.elt; a.G = io.c.G; a.Mo = io.c.Mo; a.vid = io.vid; a.CL = io.CL;
        a.h_prev = io.h_prev; a.W_sel = h->opt.selector ? w.W_sel : nullptr; a.b_sel = w.b_sel;
        a.alphag = io.alphag; a.alpham = io.alpham; a.alphalt = io.alphalt;
        a.csum = io.csum; a.cparts = io.cparts; a.sel = io.sel; a.ctx = io.ctx; a.ctx_pk = io.pn ? io.ctx_pk : nullptr;
        a.M = io.M; a.T = io.T; a.D = D; a.rowmap = io.rowmap;
        HIPCHK(h, launch_temporal(h->stream, a));
    }
    if (io.pn) {   // preact = h.U + x_ + ctx.Wc, gates, cell update (:437-457) on the row-panel kernel
        Prof pr(h, KC_LSTM);
        LstmPnArgs a{};
        a.npairs = 1; a.p[0] = io.ctx_pk ? PnPair{io.ctx_pk, D, io.pn->Wc, D, 1} : PnPair{io.ctx, D, io.pn->Wc, D, 0};
        if (io.emb) { a.p[1] = io.emb_pk ? PnPair{io.emb_pk, E, io.pn->W, E, 1} : PnPair{io.emb, E, io.pn->W, E, 0}; a.npairs = 2; a.bias = w.b; }
        a.h_pk = io.h_out_pk; a.hd_pk = io.hd_pk;
        a.pre_add = io.preh; a.ldpre = ldp;
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
        a.pre_add = io.preh; a.ldpre = ldp;
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
        HIPCHK(h, launch_bernoulli3(h->stream, *dp, n_dp, *d1, n_d1, *d2, n_d2, h->seed, 3 * h->draw));
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

}  // namespace stattn_detail
