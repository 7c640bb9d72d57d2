// C ABI of libstattn.so, part 2: the sampler -- f_init / f_next (model_attention.py:719-850) and gen_sample run on the
// device for many videos at once (stattn_beam_search, :852-994).
#include "steps.h"

#include <memory>

extern "C" {

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

static int beam_search_impl(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                            const float* ctxm, int T, int K, int k, int maxlen, int suppress_eos, int stochastic,
                            int64_t* out_tokens, float* out_scores, int32_t* out_lens, int32_t* out_count);

int stattn_beam_search(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                       const float* ctxm, int T, int K, int k, int maxlen, int suppress_eos,
                       int64_t* out_tokens, float* out_scores, int32_t* out_lens, int32_t* out_count) {
    return beam_search_impl(h, nvid, ctxg, ctxg_mask, ctxl, ctxm, T, K, k, maxlen, suppress_eos, 0, out_tokens, out_scores, out_lens, out_count);
}

// gen_sample(stochastic=True) (model_attention.py:863-918) for up to 16 videos at once, on the device: every word is a
// draw from the next-word distribution (Gumbel-max in the logits launch: no probabilities leave the chip), the caption
// ends with the first <eos> (which is part of the sample, :914-918) or after maxlen words, and the score is the SUM of
// the drawn words' probabilities, as the reference computes it (:916).  Draws follow the handle's seed (stattn_set_seed)
// and a per-call counter: reproducible, different from call to call.
int stattn_sample_search(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                         const float* ctxm, int T, int K, int maxlen, int64_t* out_tokens, float* out_scores, int32_t* out_lens) {
    if (!h || nvid < 1 || nvid > 16) return fail(h, STATTN_EINVAL, "sample_search: 1 <= nvid <= 16");
    if (h->opt.precision == 1) return fail(h, STATTN_EINVAL, "sample_search: not available on a bf16 handle");
    std::vector<int32_t> cnt(nvid);
    return beam_search_impl(h, nvid, ctxg, ctxg_mask, ctxl, ctxm, T, K, 1, maxlen, 0, 1, out_tokens, out_scores, out_lens, cnt.data());
}

static int beam_search_impl(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                            const float* ctxm, int T, int K, int k, int maxlen, int suppress_eos, int stochastic,
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
    // everything the host reads back at the end lives in ONE block (one device -> host copy into pinned memory instead of seven
    // pageable ones of ~20 us each): live_k, dead_k [nvid] | fin_len, fin_score, score0, score1 [M] | fin_tok, tok0, tok1 [M * L0]
    const size_t res_words = (size_t)2 * nvid + (size_t)4 * M + (size_t)3 * M * L0;
    int* res_blk;
    CHK(getbuf_t(h, "bs_result", res_words, &res_blk));
    live_k = res_blk; dead_k = live_k + nvid; fin_len = dead_k + nvid;
    fin_score = reinterpret_cast<float*>(fin_len + M); score[0] = fin_score + M; score[1] = score[0] + M;
    fin_tok = reinterpret_cast<int*>(score[1] + M); tok[0] = fin_tok + (size_t)M * L0; tok[1] = tok[0] + (size_t)M * L0;
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

    // one decoded word = a fixed sequence of kernel launches (five to ten, by path: below) whose arguments depend on the word index only through
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
    }
    // Small batches (<= 16 rows: the reference's own evaluation decodes ONE video at a time, metrics.py:121-135) are
    // launch-latency bound -- each of the ten launches of a word costs 5-11 us however little it computes.  Their word is
    // six launches (five once the update rides in the next word's attention launch, further down): attention, temporal fuse, LSTM,
    // [readout layer 1 | state projections of the NEXT word] in one
    // row-panel launch (both only need the new h; the projections are linear in h, so beam_update gathers their rows
    // with the hypotheses instead of recomputing them), logits with the vocabulary statistics in the epilogue (tile
    // max / sum-exp / best candidates: no logits or probabilities are stored, no softmax or top-k launch), update.
    static const char* nosmall = sw_tool("STATTN_BEAM_NOSMALL");       // A/B switch for tools
    const bool small = panels && M <= 16 && h->opt.precision != 1 && (!nosmall || stochastic);
    if (stochastic && !small) return fail(h, STATTN_EINVAL, "sample_search: needs the row-panel path (at most 16 rows, dim / dim_word multiples of 16)");
    unsigned long long* d_seed = nullptr;
    if (stochastic) {      // seed of this call's draws, in device memory (the captured word graph is reused from call to call)
        const unsigned long long draw_seed = (h->seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull * (++h->draw)) | 1ull;
        CHK(getbuf_t(h, "bs_seed", (size_t)1, &d_seed));
        HIPCHK(h, hipMemcpy(d_seed, &draw_seed, sizeof draw_seed, hipMemcpyHostToDevice));
    }
    float *proj = nullptr, *proj_step = nullptr, *ho_pk = nullptr, *vstats = nullptr;
    int vtile = 0;
    PnArgs lgargs{};
    // The vocabulary launch ends in the statistics epilogue (per row and column tile: max, sum exp, the k best logits) whenever
    // the row-panel kernels run it: the logits are never stored and the softmax / top-k launches disappear.  Up to 16 rows
    // panel.hip's kernel, beyond 64 rows the wide kernel (panelw.hip); in between (one row group of 16-column tiles, but more
    // rows than a statistics pass per wave pays for) the logits are stored and the softmax + top-k launches run.
    bool vocab_stats = small;
    // Beams of 17 .. 64 rows send their vocabulary launch to the wide kernel as well (PnArgs::wide_from = 17; every other launch of
    // such a word stays on the 16-column kernel): no logits store, no softmax launch, no top-k launches, the update reads the records
    // (`--mode beam --config c1 --beam 5`, 20 rows: 205-207 k -> 212 k row-steps/s).  Round 4 tried this by other means and one
    // parity case died with a memory access fault that was never reproduced: with THIS switch the whole GPU suite and 300 random beams
    // (fuzz_parity: 17 .. 64-row grids among them) run clean; STATTN_WIDE_STATS_FROM=65 restores the stored logits.
    static const char* wsf = sw_product("STATTN_WIDE_STATS_FROM");
    const int wide_stats_from = wsf ? atoi(wsf) : 17;
    if (panels && !vocab_stats && !stochastic && h->opt.precision != 1) {
        PnArgs probe{};
        probe.M = M; probe.nseg = 1; probe.wide_from = wide_stats_from;
        pn_seg_defaults(probe.seg[0]);
        probe.seg[0].npairs = 1; probe.seg[0].p[0] = PnPair{a1_pk, E, pn.Wo, E, 1}; probe.seg[0].N = Vp;
        vocab_stats = panel_wide_supported(probe);
    }
    if (vocab_stats) {
        // the logits launch (same arguments for every word)
        lgargs.M = M; lgargs.nseg = 1; lgargs.wide_from = small ? 0 : wide_stats_from;
        // Small-batch decode re-reads the same ~45 MB of weights every word; per XCD that is 5.5 MB through a 4 MB L2, so nothing
        // survives from word to word.  The vocabulary matrix is more than half of it: loaded with the non-temporal policy it no
        // longer displaces the rest, which then hits L2 in the other launches of the next word (configs[0]: 48.0 -> 45.5 us per
        // word).  STATTN_LOGITS_NT=0: default policy (A/B).  (Non-temporal loads on EVERY weight stream were slower: DESIGN.md.)
        { static const char* ntl = sw_tool("STATTN_LOGITS_NT"); lgargs.stream_b = (small && !(ntl && ntl[0] == '0')) ? 1 : 0; }
        PnSeg& so = lgargs.seg[0];
        pn_seg_defaults(so);
        so.npairs = 1; so.p[0] = PnPair{a1_pk, E, pn.Wo, E, 1};
        so.bias = w.bo; so.C = lg; so.ldc = Vp; so.N = Vp;
        so.stats_V = V; so.stats_kb = k; so.stats_skip0 = suppress_eos ? 1 : 0;
        so.stats_seed = d_seed; so.stats_step = d_step;
        vtile = Vp / panel_tile_cols(lgargs);
        CHK(getbuf_t(h, "bs_vstats", (size_t)M * vtile * PN_STATS_REC, &vstats));
        so.stats = vstats;
    }
    // Beams of more than 64 rows: logits on the LDS-tiled GEMM + launch_vocab_stats instead of the wide row-panel kernel with the
    // statistics epilogue (tools/probes/gemm_small_m.py: 160 x 12 032 x 512 31.5 us, 160 x 20 096 x 512 41 us against ~45 / ~70);
    // STATTN_TILED_LOGITS=0 restores the epilogue (A/B)
    static const char* tlg = sw_tool("STATTN_TILED_LOGITS");
    const bool tiled_logits = vocab_stats && !small && M > 64 && !stochastic && vtile * 32 == Vp && E % 4 == 0 && !(tlg && tlg[0] == '0');
    if (small) {
        CHK(getbuf_t(h, "bs_proj", (size_t)M * 8 * D, &proj)); CHK(getbuf_t(h, "bs_proj_step", (size_t)M * 8 * D, &proj_step));
        CHK(getbuf_t(h, "bs_ho_pk", packed_rows_floats(M, D), &ho_pk));
    }
    // Larger beams whose attention launch streams the slabs once per video (spatial_shared_kernel) and whose vocabulary launch leaves
    // statistics: the attention of word w + 1 does not depend on the words chosen at word w, only on the states -- it runs on the
    // hypotheses as they stand BEFORE the re-ordering, right behind the state projections of the new h, and carries the whole update
    // of word w (selection, log-sum-exp, gathers: one workgroup per video, 26 us per configs[4] word as a launch of its own) in its
    // first workgroups.  The temporal kernel reads its parent's scores / region contexts through `rowmap`; h.U travels with the beam.
    bool pre = false;
    int* rowmap = nullptr;
    float* preh_step = nullptr;
    // On the small path (at most 16 rows) the readout launch already leaves the projections of the new h before the re-ordering
    // (`proj_step`): the same order of launches, the attention reading them there.
    if (panels && vocab_stats && !stochastic && k > 1) {
        const char* noride = sw_product("STATTN_NO_UPDATE_RIDER");
        SpatialArgs probe{};
        probe.M = M; probe.T = T; probe.K = K; probe.D = D; probe.group = k;
        const bool hu_rider = !small && M <= 64 && spatial_rider_supported(probe);       // (run_step would let the attention launch carry h.U instead)
        pre = !noride && !hu_rider && spatial_update_supported(probe);
        if (pre) CHK(getbuf_t(h, "bs_rowmap", (size_t)M, &rowmap));
        if (pre && !small) {
            CHK(getbuf_t(h, "bs_preh_step", (size_t)M * 4 * D, &preh_step));
            CHK(getbuf_t(h, "bs_ho_pk", packed_rows_floats(M, D), &ho_pk));
        }
    }
    // Tool switch (STATTN_MERGE_PG=1, measured in DESIGN.md section 6): the state projections of the new h as two more segments of the
    // logits launch -- both only need what the LSTM launch left -- instead of a launch of their own in front of the attention.
    PnArgs lgargs_pg{};
    bool merge_pg = false;
    if (pre && !small) {
        const char* mpg = sw_tool("STATTN_MERGE_PG");
        merge_pg = mpg && mpg[0] == '1';
        if (merge_pg) {
            lgargs_pg.M = M; lgargs_pg.nseg = 3; lgargs_pg.plain_order = 1;
            for (int i = 0; i < 2; ++i) {
                PnSeg& sg = lgargs_pg.seg[i];
                pn_seg_defaults(sg);
                sg.npairs = 1; sg.p[0] = PnPair{ho_pk, D, i == 0 ? pn.Wd : pn.U, D, 1};
                sg.C = i == 0 ? sproj : preh_step; sg.ldc = 4 * D; sg.N = 4 * D;
            }
            lgargs_pg.seg[2] = lgargs.seg[0];
            merge_pg = panel_wide_supported(lgargs_pg);
        }
    }
    int* d_ticket;
    CHK(getbuf_t(h, "bs_ticket", (size_t)1, &d_ticket));
    // Beams of 2 .. 8 hypotheses on the small path: k update workgroups per video (beam_inl.h, "row workgroups"); STATTN_NO_ROW_WG=1: one (A/B)
    float* rw_cost = nullptr; int *rw_idx = nullptr, *rw_ticket = nullptr;
    if (small && vocab_stats && !stochastic && k > 1 && !sw_product("STATTN_NO_ROW_WG")) {
        CHK(getbuf_t(h, "bs_rw_cost", (size_t)M * 8, &rw_cost)); CHK(getbuf_t(h, "bs_rw_idx", (size_t)M * 8, &rw_idx));
        CHK(getbuf_t(h, "bs_rw_ticket", (size_t)nvid, &rw_ticket));
        HIPCHK(h, hipMemsetAsync(rw_ticket, 0, (size_t)nvid * sizeof(int), s));
    }
    {   // the initial beam (one live, empty, zero-score hypothesis per video on row v * k, next word -1, :871-893), its states,
        // the eval dropout multiplier, zeroed packed buffers, the zero embedding of the first word (:803-804), counters:
        // ONE launch (beam.hip beam_init_kernel) instead of twenty memsets and small copies
        BeamInitArgs bi{};
        bi.nvid = nvid; bi.k = k; bi.D = D; bi.E = E;
        bi.vid = vid; bi.live_k = live_k; bi.dead_k = dead_k; bi.next_w = next_w; bi.score0 = score[0];
        bi.h0 = h0; bi.c0 = c0; bi.hp = hp; bi.cp = cp; bi.hp_pk = hp_pk; bi.dp = dp; bi.emb = emb;
        bi.ticket = d_ticket; bi.step = d_step; bi.rowmap = rowmap;
        float* zs[5] = {ctx_pk, hd_pk, emb_pk, a1_pk, ho_pk};
        const size_t zn[5] = {packed_rows_floats(M, D), packed_rows_floats(M, D), packed_rows_floats(M, E), packed_rows_floats(M, E), packed_rows_floats(M, D)};
        for (int q = 0; q < 5; ++q) { bi.zero[q] = zs[q]; bi.zero_n[q] = zs[q] ? zn[q] : 0; }
        HIPCHK(h, launch_beam_init(s, bi));
    }
    if (small) {
        // state projections of the first word from the initial states
        PnArgs a{};
        a.M = M; a.nseg = 2;
        for (int i = 0; i < 2; ++i) {
            PnSeg& sg = a.seg[i];
            pn_seg_defaults(sg);
            sg.npairs = 1; sg.p[0] = PnPair{hp_pk, D, i == 0 ? pn.Wd : pn.U, D, 1};
            sg.C = proj + (size_t)i * 4 * D; sg.ldc = 8 * D; sg.N = 4 * D;
        }
        HIPCHK(h, launch_panel(s, a));
    }
    // (first word: no previous word, zero embedding (:803-804), set by the init launch; afterwards beam_update writes the
    // embedding of the word it selects -- no lookup launch inside the loop)
    // One hypothesis per video (greedy decode, ancestral sampling): the beam is never re-ordered, row v of every state tensor stays
    // row v.  The LSTM then writes the new state straight over the old one (each element is read and written by the same thread)
    // and the readout launch writes the next word's state projections where the attention kernel will read them: beam_update has
    // no state / projection rows to move (its gathers were 2.9 of its 9.6 us at configs[0]).
    const bool direct = small && k == 1;
    // ... and the update itself leaves the critical path: the attention of word w + 1 reads the projections the readout launch of word w
    // wrote, never the chosen word, so the bookkeeping of word w runs as one more workgroup of THAT launch (attn.hip
    // spatial_small_update_kernel) and the word loop is  T L R G [S(w + 1) | U(w)]  behind one S(0): five launches per word.
    // The attention behind the last word is computed for nothing.  STATTN_NO_UPDATE_RIDER=1: the six-launch word (A/B, tests).
    bool ride = pre;
    if (direct) {
        const char* noride = sw_product("STATTN_NO_UPDATE_RIDER");        // (read on every call: tests switch it inside one process)
        SpatialArgs probe{};
        probe.M = M; probe.T = T; probe.K = K; probe.D = D; probe.group = k;
        ride = !noride && spatial_update_supported(probe);
    }
    h->path_upd_rider = 0;
    h->path_upd_rowwg = 0;
    auto step_io = [&]() {
        StepIO io{};
        io.M = M; io.T = T; io.K = K; io.c = c; io.vid = vid; io.group = k;
        io.h_prev = hp; io.c_prev = cp; io.sproj = sproj; io.preh = preh; io.xproj = nullptr; io.emb = emb;
        io.dp = dp; io.mask = nullptr; io.d1 = nullptr;
        io.alphal = al; io.CL = CL; io.eg = eg; io.em = em; io.elt = elt; io.plt = plt;
        io.alphag = ag; io.alpham = am; io.alphalt = alt; io.csum = nullptr; io.sel = nullptr; io.ctx = ctx;
        io.h_out = direct ? hp : ho; io.c_out = direct ? cp : co; io.gates = nullptr; io.hd = hd;
        io.pn = panels ? &pn : nullptr;
        io.h_prev_pk = hp_pk; io.h_out_pk = nullptr; io.ctx_pk = ctx_pk; io.emb_pk = emb_pk; io.hd_pk = hd_pk;
        if (small) {       // projections of this word are in `proj` ([sproj | preh] per row); the LSTM also packs the new h
            io.skip_hproj = true; io.sproj = proj; io.preh = proj + (size_t)4 * D; io.ldproj = 8 * D; io.h_out_pk = ho_pk;
        }
        return io;
    };
    auto enqueue_word = [&](int parity, bool last = false) -> int {      // last: the word at maxlen - 1 (its update is a launch of its own:
                                                                         // there is no next word whose attention could carry it)
        StepIO io = step_io();
        io.phase = ride ? 2 : 0;
        if (pre) { io.rowmap = rowmap; io.h_out_pk = ho_pk; }
        // (lt_mode 0 forms CL.Wclt + slt in this phase, from the CL rows the attention launch left in the order BEFORE the re-ordering:
        //  the slt of those rows, not the gathered ones)
        if (pre && small) io.sproj = proj_step;
        CHK(run_step(h, io));
        std::unique_ptr<Prof> pro(new Prof(h, KC_READOUT));        // readout + vocabulary launch (+ softmax) of this word
        if (small) {       // readout layer 1 + the next word's state projections (before the beam is re-ordered), then logits -> statistics
            PnArgs a{};
            a.M = M; a.nseg = 3;
            PnSeg& sg = a.seg[0];
            pn_seg_defaults(sg);
            sg.npairs = 1; sg.p[0] = PnPair{hd_pk, D, pn.Wl1, D, 1};
            if (h->opt.ctx2out) { sg.p[1] = PnPair{ctx_pk, D, pn.Wl2, D, 1}; sg.npairs = 2; sg.bias2 = w.bl2; }
            sg.Cpk = a1_pk;
            sg.bias = w.bl1;
            if (h->opt.prev2out) { sg.add = emb; sg.ldadd = E; }
            sg.act = 1; sg.scale = 0.5f; sg.C = a1; sg.ldc = E; sg.N = E;
            for (int i = 0; i < 2; ++i) {
                PnSeg& sp = a.seg[1 + i];
                pn_seg_defaults(sp);
                sp.npairs = 1; sp.p[0] = PnPair{ho_pk, D, i == 0 ? pn.Wd : pn.U, D, 1};
                sp.C = (direct ? proj : proj_step) + (size_t)i * 4 * D; sp.ldc = 8 * D; sp.N = 4 * D;
            }
            HIPCHK(h, launch_panel(s, a));
            HIPCHK(h, launch_panel(s, lgargs));
        } else if (panels) {      // readout (:817-838) on the row-panel kernel
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
            if (vocab_stats && tiled_logits && !merge_pg) {
                // more than 64 rows: the logits on the LDS-tiled GEMM (stored), the statistics records by a kernel of their own
                GemmArgs g;
                gemm_defaults(g); g.split = h->opt.precision != 0;
                g.A = a1; g.lda = E; g.B = w.Wo; g.ldb = Vp; g.C = lg; g.ldc = Vp; g.M = M; g.N = Vp; g.K = E; g.bias = w.bo;
                HIPCHK(h, launch_gemm(s, g, false, false));
                HIPCHK(h, launch_vocab_stats(s, lg, Vp, M, V, vtile, k, suppress_eos ? 1 : 0, vstats));
            } else if (vocab_stats) {
                HIPCHK(h, launch_panel(s, (merge_pg && !last) ? lgargs_pg : lgargs));
            } else {
                PnArgs b{};
                b.M = M; b.nseg = 1;
                PnSeg& so = b.seg[0];
                pn_seg_defaults(so);
                so.npairs = 1; so.p[0] = PnPair{a1_pk, E, pn.Wo, E, 1};
                so.bias = w.bo; so.C = lg; so.ldc = Vp; so.N = Vp;
                HIPCHK(h, launch_panel(s, b));
                HIPCHK(h, launch_softmax_nll(s, lg, Vp, pr, Vp, nullptr, nullptr, nullptr, M, V));
            }
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
        pro.reset();
        std::unique_ptr<Prof> prs(new Prof(h, KC_SELECT, !ride));   // candidate selection + beam update of this word (riding: inside the attention launch's scope)
        BeamArgs ba{};
        ba.probs = pr; ba.ldp = Vp; ba.V = V; ba.k = k; ba.D = D; ba.maxlen = L0; ba.nvid = nvid; ba.step = d_step;
        ba.suppress_eos = suppress_eos;
        ba.live_k = live_k; ba.dead_k = dead_k; ba.hyp_score = score[parity]; ba.hyp_score_out = score[parity ^ 1];
        ba.tok_in = tok[parity]; ba.tok_out = tok[parity ^ 1];
        ba.fin_tok = fin_tok; ba.fin_score = fin_score; ba.fin_len = fin_len; ba.next_w = next_w;
        ba.h_step = direct ? hp : ho; ba.c_step = direct ? cp : co; ba.h_next = hp; ba.c_next = cp;
        ba.end_h = end_h; ba.end_c = end_c; ba.end_rows = end_rows; ba.h_next_pk = hp_pk;
        ba.Wemb = w.Wemb; ba.E = E; ba.emb_next = emb; ba.emb_next_pk = emb_pk; ba.ticket = d_ticket;
        if (vocab_stats) {
            ba.probs = nullptr; ba.stats = vstats; ba.ntile = vtile; ba.tile_cols = Vp / vtile; ba.stochastic = stochastic;
            ba.rw_cost = rw_cost; ba.rw_idx = rw_idx; ba.rw_ticket = rw_ticket;
            if (small && !direct) { ba.proj_step = proj_step; ba.proj_next = proj; ba.nproj = 8 * D; }
            if (pre) { ba.rowmap = rowmap; ba.h_next_pk = nullptr; }          // (the packed h of the re-ordered beam has no reader any more)
            if (pre && !small) { ba.proj_step = preh_step; ba.proj_next = preh; ba.nproj = 4 * D; }
        } else {
            HIPCHK(h, launch_beam_topk(s, ba, tk_cost, tk_idx));
        }
        if (ride && !last) {        // the next word's attention launch carries this word's update
            io.phase = 1; io.upd = &ba; io.rowmap = nullptr;
            if (pre && small) io.sproj = proj_step;                                      // where the readout launch of this word left them
            else if (pre && merge_pg) { io.skip_hproj = true; }                            // (they rode in the logits launch)
            else if (pre) { io.h_prev = ho; io.h_prev_pk = ho_pk; io.preh = preh_step; }     // state projections of the new h, before the re-ordering
            CHK(run_step(h, io));
            return STATTN_OK;
        }
        HIPCHK(h, launch_beam_update(s, ba, tk_cost, tk_idx));
        return STATTN_OK;
    };
    if (pre && small)      // (first word: the attention launch and the lt_mode-0 GEMM read the projections of the initial states there)
        HIPCHK(h, hipMemcpyAsync(proj_step, proj, (size_t)M * 8 * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (ride) {            // attention of the first word (with `pre`: behind the state projections of the initial states)
        StepIO io = step_io();
        io.phase = 1;            // (h.U of the initial states goes straight to `preh`: nothing re-orders the beam before the first LSTM launch)
        CHK(run_step(h, io));
    }

    // The launch-bound inner loop is captured once as hipGraphs of EIGHT and of TWO words (even + odd parity alternate)
    // and replayed -- a replay costs 10-16 us of host / front-end time whatever it holds, so the long graph carries the
    // bulk and the short one the remainder; the host only comes back every 8 words to see whether every video has finished.  Falls back to eager launches if
    // the capture is refused (or STATTN_BEAM_NOGRAPH is set, for A/B runs).
    hipGraphExec_t gexec = nullptr;
    static const char* nograph = sw_product("STATTN_BEAM_NOGRAPH");
    h->beam_graph_replays = 0;
    // everything a captured launch bakes in: shapes, options and every buffer the word sequence touches
    std::vector<uintptr_t> sig = {(uintptr_t)nvid, (uintptr_t)k, (uintptr_t)T, (uintptr_t)K, (uintptr_t)L0, (uintptr_t)suppress_eos, (uintptr_t)stochastic, (uintptr_t)d_seed, (uintptr_t)ride, (uintptr_t)pre, (uintptr_t)merge_pg, (uintptr_t)rowmap, (uintptr_t)preh_step,
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
                          (const void*)ctx_pk, (const void*)emb_pk, (const void*)hd_pk, (const void*)a1_pk, (const void*)d_ticket,
                          (const void*)proj, (const void*)proj_step, (const void*)ho_pk, (const void*)vstats, (const void*)rw_cost, (const void*)rw_ticket})
        sig.push_back((uintptr_t)q);
    hipGraphExec_t gexec8 = nullptr, gexec_last = nullptr;
    auto capture = [&](int nwords, bool ends = false) -> hipGraphExec_t {
        hipGraph_t graph = nullptr;
        hipGraphExec_t ge = nullptr;
        bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            int r = STATTN_OK;
            for (int i = 0; i < nwords && r == STATTN_OK; ++i) r = enqueue_word(i & 1, ends && i == nwords - 1);
            const hipError_t e = hipStreamEndCapture(s, &graph);
            ok = r == STATTN_OK && e == hipSuccess && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) { ge = nullptr; (void)hipGetLastError(); }
        return ge;
    };
    if (!nograph && !h->profiling && L0 >= 2 && h->beam_gexec && h->beam_gsig == sig) {
        gexec = h->beam_gexec; gexec8 = h->beam_gexec8; gexec_last = h->beam_gexec_last;          // same buffers and shapes as last time: replay as is
    } else if (!nograph && !h->profiling && L0 >= 2) {
        if (h->beam_gexec) { (void)hipGraphExecDestroy(h->beam_gexec); h->beam_gexec = nullptr; }
        if (h->beam_gexec8) { (void)hipGraphExecDestroy(h->beam_gexec8); h->beam_gexec8 = nullptr; }
        if (h->beam_gexec_last) { (void)hipGraphExecDestroy(h->beam_gexec_last); h->beam_gexec_last = nullptr; }
        gexec = capture(2);
        if (gexec && L0 >= 8) gexec8 = capture(8);
        if (gexec && ride) gexec_last = capture(2, true);
        if (gexec) { h->beam_gexec = gexec; h->beam_gexec8 = gexec8; h->beam_gexec_last = gexec_last; h->beam_gsig = sig; }
    }
    int steps_run = 0;
    int rc_loop = STATTN_OK;
    for (int st = 0; st < L0;) {
        const int tail = gexec_last ? 2 : 0;                       // words kept for the graph that ends the search
        if (gexec_last && (st & 1) == 0 && st + 2 == L0) {
            if (hipGraphLaunch(gexec_last, s) != hipSuccess) { rc_loop = fail(h, STATTN_EHIP, "beam_search: hipGraphLaunch failed"); break; }
            st += 2;
            ++h->beam_graph_replays;
        } else if (gexec8 && (st & 1) == 0 && st + 8 <= L0 - tail) {
            if (hipGraphLaunch(gexec8, s) != hipSuccess) { rc_loop = fail(h, STATTN_EHIP, "beam_search: hipGraphLaunch failed"); break; }
            st += 8;
            ++h->beam_graph_replays;
        } else if (gexec && (st & 1) == 0 && st + 2 <= L0 - tail) {
            if (hipGraphLaunch(gexec, s) != hipSuccess) { rc_loop = fail(h, STATTN_EHIP, "beam_search: hipGraphLaunch failed"); break; }
            st += 2;
            ++h->beam_graph_replays;
        } else {
            rc_loop = enqueue_word(st & 1, ride && st + 1 == L0);
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
    h->path_upd_rider = ride ? steps_run : 0;      // (replayed graphs included)
    // (a riding update falls back to one workgroup per video when its attention launch is the shared-slab kernel: count what ran)
    h->path_upd_rowwg = rw_cost && (!ride || h->upd_rowwg_last) ? steps_run : 0;
    h->path_vocab_stats = vocab_stats ? steps_run : 0;
    // results: finished hypotheses in order of death, then the remaining live ones (:987-992)
    {
        const int fb = steps_run & 1;     // buffers written by the last executed step
        if (res_words * 4 > h->pin_res_bytes) {
            if (h->pin_res) { (void)hipHostFree(h->pin_res); h->pin_res = nullptr; h->pin_res_bytes = 0; }
            HIPCHK(h, hipHostMalloc(&h->pin_res, res_words * 4, hipHostMallocDefault));
            h->pin_res_bytes = res_words * 4;
        }
        HIPCHK(h, hipMemcpyAsync(h->pin_res, res_blk, res_words * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        const int* hr = static_cast<const int*>(h->pin_res);
        std::vector<int> lv(hr, hr + nvid), dv(hr + nvid, hr + 2 * nvid);
        const int* flen = hr + 2 * nvid;
        const float* fsc = reinterpret_cast<const float*>(flen + M);
        const float* lsc = fsc + M + (size_t)fb * M;
        const int* ftok = reinterpret_cast<const int*>(fsc + 3 * (size_t)M);
        const int* ltok = ftok + (size_t)(1 + fb) * M * L0;
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

}  // extern "C"
