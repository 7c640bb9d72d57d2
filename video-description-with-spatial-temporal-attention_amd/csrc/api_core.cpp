// C ABI of libstattn.so (see include/stattn.h), part 1: handle life cycle and the parameter store.
// (api_sampler.cpp: f_init / f_next / beam search; api_train.cpp: batch staging + build_model forward;
// api_backward.cpp: BPTT, loss, Adadelta; api_dbg.cpp: development entry points; comm.cpp: RCCL exchange.)
#include "steps.h"

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
    e = h->fb_params.ensure(h->nflat * sizeof(float));
    if (e == hipSuccess) { h->d_params = static_cast<float*>(h->fb_params.p); e = h->fb_grads.ensure(h->nflat * sizeof(float)); }
    if (e == hipSuccess) h->d_grads = static_cast<float*>(h->fb_grads.p);
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
    if (h->beam_gexec8) (void)hipGraphExecDestroy(h->beam_gexec8);
    if (h->beam_gexec_last) (void)hipGraphExecDestroy(h->beam_gexec_last);
    if (h->pin_io) (void)hipHostFree(h->pin_io);
    if (h->pin_res) (void)hipHostFree(h->pin_res);
    if (h->pin_plan[0]) (void)hipHostFree(h->pin_plan[0]);
    if (h->pin_plan[1]) (void)hipHostFree(h->pin_plan[1]);
    for (hipEvent_t e : h->plan_ev) if (e) (void)hipEventDestroy(e);
    h->fb_params.release(); h->fb_grads.release(); h->fb_rg2.release(); h->fb_ru2.release();
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->staged_ev) (void)hipEventDestroy(h->staged_ev);
    if (h->free_ev[0]) (void)hipEventDestroy(h->free_ev[0]);
    if (h->free_ev[1]) (void)hipEventDestroy(h->free_ev[1]);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// test hook (csrc/stattn_dbg.h): the gradient regions of a configuration, computed on the host -- no device is touched
int stattn_dbg_grad_regions(const stattn_options* o, int max_regions, size_t* offsets, size_t* lengths, int* n_regions, size_t* nflat) {
    if (!o || !offsets || !lengths || !n_regions || !nflat || max_regions < N_GRAD_REGIONS)
        return fail(nullptr, STATTN_EINVAL, "dbg_grad_regions: bad argument");
    if (o->dim <= 0 || o->dim_word <= 0 || o->n_words <= 0 || o->ctxl_dim <= 0 || o->ctxm_dim <= 0)
        return fail(nullptr, STATTN_EINVAL, "dbg_grad_regions: bad options");
    stattn_handle* h = new stattn_handle();
    h->opt = *o;
    h->D = o->dim; h->E = o->dim_word; h->V = o->n_words; h->Vp = (int)align_up((size_t)o->n_words, 128);
    h->Fl = o->ctxl_dim; h->Fm = o->ctxm_dim;
    build_param_table(h);
    int rc = STATTN_OK;
    for (int i = 0; i < N_GRAD_REGIONS && rc == STATTN_OK; ++i) {
        auto a = h->pindex.find(GRAD_REGIONS[i].first);
        auto b = GRAD_REGIONS[i].before ? h->pindex.find(GRAD_REGIONS[i].before) : h->pindex.end();
        if (a == h->pindex.end() || (GRAD_REGIONS[i].before && b == h->pindex.end())) { rc = fail(nullptr, STATTN_ENOTFOUND, "dbg_grad_regions: region %d names an absent parameter", i); break; }
        const size_t lo = h->params[a->second].off, hi = GRAD_REGIONS[i].before ? h->params[b->second].off : h->nflat;
        offsets[i] = lo; lengths[i] = hi - lo;
    }
    *n_regions = N_GRAD_REGIONS; *nflat = h->nflat;
    delete h;
    return rc;
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
    if (h && comm_pending(h))
        return fail(h, STATTN_ESTATE, "get_grad: regions of this gradient are being summed over the ranks on the side stream; call stattn_allreduce_grads first");
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

// ---- measurement hooks -----------------------------------------------------------------
int stattn_set_profiling(stattn_handle* h, int enable) {
    if (!h) return STATTN_EINVAL;
    prof_collect(h);
    h->profiling = enable != 0;
    for (int i = 0; i < KC_TOTAL; ++i) { h->k_ms[i] = 0; h->k_n[i] = 0; }
    return STATTN_OK;
}
int stattn_get_kernel_ms(stattn_handle* h, int which, float* ms_avg, int* launches) {
    if (!h || which < 0 || which >= KC_TOTAL || !ms_avg) return STATTN_EINVAL;
    prof_collect(h);
    *ms_avg = h->k_n[which] ? (float)(h->k_ms[which] / h->k_n[which]) : 0.f;
    if (launches) *launches = h->k_n[which];
    return STATTN_OK;
}

}  // extern "C"
