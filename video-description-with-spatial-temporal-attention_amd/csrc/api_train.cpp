// C ABI of libstattn.so, part 3: minibatch staging (prepare_data -> HBM) and the forward pass of build_model
// (model_attention.py:583-717).
#include "steps.h"

extern "C" {

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
    {   // the embedding-gradient plan of the shadow set: a few KB through a pinned staging block per set
        std::vector<int> buf;
        build_embed_plan(x, t, m, buf, h->emb_plan[set]);
        int* d;
        CHK(getbuf_t(h, bset(h, "embplan", set).c_str(), buf.size(), &d));
        // the pinned block of this set fed an asynchronous copy at the prefetch before last; a host that runs several
        // prefetches ahead of the GPU must not overwrite (or free) it while that copy is still queued
        if (h->plan_ev_valid[set]) HIPCHK(h, hipEventSynchronize(h->plan_ev[set]));
        if (buf.size() * sizeof(int) > h->pin_plan_bytes[set]) {
            if (h->pin_plan[set]) { (void)hipHostFree(h->pin_plan[set]); h->pin_plan[set] = nullptr; h->pin_plan_bytes[set] = 0; }
            HIPCHK(h, hipHostMalloc(&h->pin_plan[set], 2 * buf.size() * sizeof(int), hipHostMallocDefault));
            h->pin_plan_bytes[set] = 2 * buf.size() * sizeof(int);
        }
        memcpy(h->pin_plan[set], buf.data(), buf.size() * sizeof(int));
        HIPCHK(h, hipMemcpyAsync(d, h->pin_plan[set], buf.size() * sizeof(int), hipMemcpyHostToDevice, cs));
        if (!h->plan_ev[set]) HIPCHK(h, hipEventCreateWithFlags(&h->plan_ev[set], hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->plan_ev[set], cs));
        h->plan_ev_valid[set] = true;
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
    h->path_fwd_rider = h->path_fwd_panel = 0;

    int64_t* dx = (int64_t*)h->bufs[bcur(h, "x")].p;
    float* dmask = findbuf(h, bcur(h, "mask").c_str());
    float* mG = findbuf(h, bcur(h, "mG").c_str());
    float* rawl = findbuf(h, bcur(h, "rawl").c_str());
    float* rawm = findbuf(h, bcur(h, "rawm").c_str());
    CtxPtrs c{};
    c.G = findbuf(h, bcur(h, "G").c_str());
    float *mean, *emb, *xproj, *hs, *cs, *hd, *ctx, *csum, *cparts, *sel, *al, *ag, *am, *alt, *CL, *gates, *sproj, *preh,
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
    CHK(getbuf_t(h, "cparts", R * 3 * D, &cparts));
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
        {
            Prof pp(h, KC_PROLOGUE);
            CHK(bf16_weights(h, &bw, true));
            CHK(getbuf_t(h, "bx_emb", R * E, &bemb));
            HIPCHK(h, launch_cvt_bf16(s, emb, bemb, R * E));
        }
        GemmBfArgs g = bf_args(bemb, E, bw.W, (int)R, 4 * D, E);       // x_ = emb.W + b (:334-335): rides with the projections
        g.bias = w.b; g.C = xproj; g.ldc = 4 * D;
        CHK(project_context(h, m, T, K, c.G, rawl, rawm, c, nullptr, &g));
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
        io.csum = csum + r0 * D; io.cparts = cparts + r0 * 3 * D; io.sel = sel + r0; io.ctx = ctx + r0 * D;
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
        GemmBfArgs g;
        static const char* nopair = sw_product("STATTN_READOUT_NOPAIR");       // A/B switch for tools
        if (h->opt.ctx2out && bw.Wl12 && !nopair) {
            // a = tanh((h*d1).Wl1 + ctx.Wl2 + bl1 + bl2 [+ emb]) * d2 as ONE K-concatenated problem: [hd | ctx] rounded into one
            // [R][2 D] operand, [Wl1 ; Wl2] in rows of 2 D (z1 is never formed)
            uint16_t* bcat;
            CHK(getbuf_t(h, "bx_hdctx", R * 2 * D, &bcat));
            HIPCHK(h, launch_cvt_bf16_2d(s, hd, (size_t)D, bcat, (size_t)2 * D, R, D));
            HIPCHK(h, launch_cvt_bf16_2d(s, ctx, (size_t)D, bcat + D, (size_t)2 * D, R, D));
            g = bf_args(bcat, 2 * D, bw.Wl12, (int)R, E, 2 * D);
            g.bias = w.bl1; g.bias_b = w.bl2;
            if (h->opt.prev2out) { g.add = emb; g.ldadd = E; }
            g.act = 1; g.mul = d2; g.ldmul = E; g.Cb = ba1; g.ldcb = E; g.C = a1; g.ldc = E;   // (a1 in fp32 too: the backward pass reads it)
            HIPCHK(h, gemm_bf(h, g));
        } else {
        HIPCHK(h, launch_cvt_bf16(s, hd, bhd, R * D));
        g = bf_args(bhd, D, bw.Wl1, (int)R, E, D);
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
        }
        g = bf_args(ba1, E, bw.Wo, (int)R, Vp, E);
        g.bias = w.bo; g.C = lg; g.ldc = Vp;
        HIPCHK(h, gemm_bf(h, g));
    } else {
        Prof pr_(h, KC_READOUT);
        GemmArgs g;
        static const char* nopair = sw_product("STATTN_READOUT_NOPAIR");       // A/B switch for tools
        const bool pair = h->opt.ctx2out && h->opt.precision == 0 && D % 32 == 0 && !nopair;
        if (pair) {
            // a = tanh((h*d1).Wl1 + ctx.Wl2 + bl1 + bl2 [+ emb]) * d2 as ONE K-concatenated GEMM (K = 2 D): the two 240-tile
            // launches it replaces each left three quarters of the chip's workgroup slots empty; split-K over the joint K
            // fills them, and the epilogue runs in the reduction of the partial tiles
            float* fws;
            const size_t FWS = (size_t)8 * R * E;
            CHK(getbuf_t(h, "f_ws", FWS, &fws));
            gemm_defaults(g);
            g.A = hd; g.lda = D; g.B = w.Wl1; g.ldb = E; g.K = D;
            g.A2 = ctx; g.lda2 = D; g.B2 = w.Wl2; g.ldb2 = E; g.K2 = D;
            g.C = a1; g.ldc = E; g.M = (int)R; g.N = E; g.bias = w.bl1; g.bias2 = w.bl2;
            if (h->opt.prev2out) { g.add = emb; g.ldadd = E; }
            g.act = 1; g.mul = d2; g.ldmul = E; g.Cact = tz; g.ldcact = E;
            g.ws = fws; g.ws_floats = FWS;
            HIPCHK(h, gemm_nn(h, g));
        } else {
        gemm_defaults(g); g.split = h->opt.precision != 0;      // z1 = (h*d1).Wl1 + bl1 [+ emb]
        g.A = hd; g.lda = D; g.B = w.Wl1; g.ldb = E; g.C = h->opt.ctx2out ? z1 : a1; g.ldc = E;
        g.M = (int)R; g.N = E; g.K = D; g.bias = w.bl1;
        if (h->opt.prev2out) { g.add = emb; g.ldadd = E; }
        if (!h->opt.ctx2out) { g.act = 1; g.mul = d2; g.ldmul = E; g.Cact = tz; g.ldcact = E; }
        HIPCHK(h, gemm_nn(h, g));
        }
        if (h->opt.ctx2out && !pair) {  // a = tanh(ctx.Wl2 + bl2 + z1) * d2
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

}  // extern "C"
