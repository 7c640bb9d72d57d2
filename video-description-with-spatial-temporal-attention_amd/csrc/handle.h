// Private definitions shared by the translation units behind the C ABI (api.cpp, comm.cpp): the handle, its
// device-buffer cache and the error helpers.  Not installed; include/stattn.h is the public surface.
#pragma once
#include "../../include/stattn.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace stattn;

namespace stattn_detail {

struct ParamInfo {
    std::string name;
    int ndim;
    int64_t dims[2];
    size_t off;     // offset in the flat buffer (floats)
    int ld;         // leading dimension (floats) of the device layout
    size_t count;   // logical element count
};

// Red zones (STATTN_DBG_REDZONE=1, a product switch: csrc/switches.h).  Every device buffer of the library -- all of them come from
// DevBuf -- is then allocated with REDZONE_BYTES of canary bytes in front of it and behind the largest size ever requested for it
// (the allocation's 256-byte rounding slack is canary as well), and stattn_dbg_redzone_check() scans all of them: a kernel that wrote
// outside any buffer of the library is named with the buffer and the byte offset.  HIP AddressSanitizer does not run on this pool
// (no XNACK); this is the bounds check that does.  Off (the default), DevBuf is a plain hipMalloc.
constexpr size_t REDZONE_BYTES = 4096;
constexpr int REDZONE_BYTE = 0xCB;
inline size_t redzone_bytes() {
    static const char* rz = stattn::sw_product("STATTN_DBG_REDZONE");
    return (rz && rz[0] && rz[0] != '0') ? REDZONE_BYTES : 0;
}

struct DevBuf {
    void* p = nullptr;          // what the library uses
    size_t cap = 0;             // usable bytes at p
    void* base = nullptr;       // the allocation (== p without red zones)
    size_t used = 0;            // largest request so far: the tail canary starts at p + used
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) { if (bytes > used) used = bytes; return hipSuccess; }
        if (base) { hipError_t e = hipFree(base); if (e != hipSuccess) return e; base = p = nullptr; cap = used = 0; }
        const size_t rz = redzone_bytes();
        size_t want = (bytes + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&base, want + 2 * rz);
        if (e != hipSuccess) { base = nullptr; return e; }
        p = static_cast<char*>(base) + rz; cap = want; used = bytes;
        if (rz) {
            // (synchronous fills: a debugging mode.  The slack between the request and the rounded size starts as canary too; a later,
            //  larger request that still fits moves the tail's start and simply finds canary bytes where garbage would be otherwise)
            e = hipMemset(base, REDZONE_BYTE, rz);
            if (e == hipSuccess) e = hipMemset(static_cast<char*>(p) + bytes, REDZONE_BYTE, want - bytes + rz);
        }
        return e;
    }
    void release() { if (base) { (void)hipFree(base); base = p = nullptr; cap = used = 0; } }
};

enum KClass { KC_SPATIAL = 0, KC_HPROJ, KC_LTGEMM, KC_TEMPORAL, KC_LSTM, KC_PROLOGUE, KC_READOUT, KC_GEMM_NN, KC_SELECT, KC_COUNT };
constexpr int KC_GEMM_SEQ = 16;      // the first 16 plain GEMM launches of a forward pass are also timed one by one
constexpr int KC_BWD_SEQ = 24;       // every LDS-tiled GEMM launch of a backward pass, in launch order
// kernels of one reverse-scan step and the deferred context-gradient kernel
enum KBwd { KB_LSTM = 0, KB_PANEL1, KB_TBWD, KB_SPATIAL, KB_REDUCE, KB_PANEL2, KB_CTXGRAD, KB_COUNT };
constexpr int KC_BWD0 = KC_COUNT + KC_GEMM_SEQ;            // first backward-GEMM slot
constexpr int KC_KB0 = KC_BWD0 + KC_BWD_SEQ;               // first reverse-scan kernel slot
constexpr int KC_TOTAL = KC_KB0 + KB_COUNT;

struct Weights {   // device pointers into the flat parameter buffer
    float *Wemb, *ff_state_W, *ff_state_b, *ff_memory_W, *ff_memory_b, *ff_local_W, *ff_local_b,
          *ff_motion_W, *ff_motion_b, *W, *U, *b, *Wc, *Wcg, *Wcm, *Wclt, *Wdg, *Wdm, *Wdlt, *bg, *bm, *blt,
          *Wcl, *Wdl, *bl, *Ug, *cg, *Um, *cm, *Ult, *clt, *Ul, *cl, *W_sel, *b_sel,
          *Wl1, *bl1, *Wl2, *bl2, *Wo, *bo;
};

}  // namespace stattn_detail
using namespace stattn_detail;

struct stattn_handle {
    stattn_options opt{};
    int D = 0, E = 0, V = 0, Vp = 0, Fl = 0, Fm = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    std::vector<ParamInfo> params;
    std::map<std::string, int> pindex;
    float* d_params = nullptr;
    float* d_grads = nullptr;
    float* d_rg2 = nullptr;      // Adadelta running averages (common.py:180-181), allocated on first update
    float* d_ru2 = nullptr;
    DevBuf fb_params, fb_grads, fb_rg2, fb_ru2;   // their allocations (DevBuf: red zones under STATTN_DBG_REDZONE like every other buffer)
    bool have_bwd = false;
    size_t nflat = 0;
    Weights w{};

    float use_noise = 0.f;
    uint64_t seed = 1234, draw = 0;

    std::map<std::string, DevBuf> bufs;

    // training batch
    int t = 0, m = 0, T = 0, K = 0;
    bool have_batch = false, have_fwd = false;
    bool masks_user = false;
    int masks_t = 0, masks_m = 0;        // shape the mask buffers currently hold
    int masks_state = 0;                 // 0 invalid, 1 holds eval (0.5), 2 holds a random draw

    // double-buffered batch staging (prepare_data -> HBM pipeline): set 0 / 1, a copy stream and a ready event
    int cur_set = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t staged_ev = nullptr;
    hipEvent_t free_ev[2] = {nullptr, nullptr};   // "every kernel that read set i has been enqueued before this"
    bool free_valid[2] = {false, false};
    bool have_pending = false;
    int p_t = 0, p_m = 0, p_T = 0, p_K = 0;
    // deterministic embedding gradient: per batch set, the plan built from the token ids (kernels.h EmbedPlan)
    struct EmbPlanHost { int npieces = 0, nwords = 0, nmulti = 0, ntok = 0; } emb_plan[2];
    void* pin_plan[2] = {nullptr, nullptr}; size_t pin_plan_bytes[2] = {0, 0};   // pinned staging of the plan (stattn_prefetch_batch)
    hipEvent_t plan_ev[2] = {nullptr, nullptr};   // copy stream: "the H2D copy out of pin_plan[set] has finished"
    bool plan_ev_valid[2] = {false, false};

    // sampler: the resident video (stattn_set_video, or the last f_next call that passed host features).
    // ck_valid: raw features are in HBM; ck_proj: their projections match the current parameters
    int ck_T = 0, ck_K = 0;
    bool ck_proj = false;
    // beam search: the captured two-word graph is kept while every pointer and shape it baked in is unchanged
    hipGraphExec_t beam_gexec = nullptr;    // two words (even + odd ping-pong parity)
    hipGraphExec_t beam_gexec_last = nullptr;   // two words, the second one the LAST of a search (no attention launch for a word that never comes)
    hipGraphExec_t beam_gexec8 = nullptr;   // eight words: a replay costs 10-16 us of launch overhead whatever it holds
    std::vector<uintptr_t> beam_gsig;
    long beam_graph_replays = 0;        // replays in the last stattn_beam_search (0 = eager launches)
    // which code path the decoder steps took (stattn_dbg_counter 1..4; reset by stattn_forward_train / stattn_backward):
    // forward steps whose attention launch carried the h.U rider, forward steps on the row-panel kernels, reverse steps
    // whose attention launch carried the dhU rider, reverse steps on the row-panel kernels
    long path_fwd_rider = 0, path_fwd_panel = 0, path_bwd_rider = 0, path_bwd_panel = 0;
    long path_upd_rider = 0;        // words of the last beam / sample search whose update rode in the next word's attention launch
    bool upd_rowwg_last = false;    // the last riding update launch ran row workgroups (set by run_step)
    long path_vocab_stats = 0;      // words of the last beam search whose vocabulary launch ended in the statistics epilogue (no logits stored)
    long path_upd_rowwg = 0;        // ... whose update ran as k workgroups per video (row workgroups, beam_inl.h)
    bool ck_valid = false;
    // f_next staging: one pinned block for {h, c, x} in and {h, c, probs} out per call (a pageable copy costs
    // ~15 us); sn_m / sn_dp / sn_vid remember that the constant step inputs (video index, eval dropout) are in place
    void* pin_io = nullptr; size_t pin_io_bytes = 0;
    void* pin_res = nullptr; size_t pin_res_bytes = 0;      // pinned landing block of a beam search's results
    int sn_m = -1; const void* sn_dp = nullptr; const void* sn_vid = nullptr;
    uint64_t host_rng = 0x853c49e6748fea9bull;
    // batched beam search: raw features staged by stattn_beam_stage (or the last call that passed host features)
    int bk_n = 0, bk_T = 0, bk_K = 0;
    bool bk_valid = false;
    int bf_nvid = 0, bf_k = 0, bf_fb = 0;          // last beam search: shape, final ping-pong parity, live counts
    std::vector<int> bf_live;

    // profiling
    bool profiling = false;
    struct EvPair { hipEvent_t a, b; int cls; };
    std::vector<EvPair> ev_used;
    std::vector<hipEvent_t> ev_pool;
    double k_ms[KC_TOTAL] = {0};
    int k_n[KC_TOTAL] = {0};
    int bwd_seq = 0;                      // index of the next LDS-tiled GEMM launch within the current backward pass
    int gemm_seq = 0;                     // index of the next plain GEMM launch within the current forward pass

    // data parallel (comm.cpp): RCCL communicator of this rank, a side stream for the bucketed gradient reduce
    void* comm = nullptr;                 // ncclComm_t
    int comm_rank = 0, comm_nranks = 1;
    int comm_overlap = 1;                 // reduce buckets on comm_stream while backward still computes
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ready[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // main stream: "this region's gradients are final" (one per region)
    int comm_regions = 0;                 // regions handed over in the current backward pass
    hipEvent_t comm_done = nullptr;       // comm stream: "every bucket issued so far has been reduced"
    hipEvent_t comm_t0 = nullptr, comm_t1 = nullptr;   // compute stream, timed: around the wait inside stattn_allreduce_grads
    bool comm_timed = false;              // comm_t0 / comm_t1 bracket the last all-reduce
    size_t comm_covered = 0;              // floats of the gradient buffer already handed to the overlapped reduce
    bool grads_reduced = false;           // the gradient buffer holds the SUM over ranks
};

namespace stattn_detail {

extern std::string g_create_error;

inline int fail(stattn_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, STATTN_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                    \
    } while (0)

#define CHK(expr) do { int rc_ = (expr); if (rc_ != STATTN_OK) return rc_; } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int getbuf(stattn_handle* h, const char* name, size_t nbytes, void** out) {
    DevBuf& b = h->bufs[name];
    HIPCHK(h, b.ensure(nbytes ? nbytes : 4));
    *out = b.p;
    return STATTN_OK;
}
template <class T>
inline int getbuf_t(stattn_handle* h, const char* name, size_t n, T** out) {
    void* p = nullptr;
    CHK(getbuf(h, name, n * sizeof(T), &p));
    *out = static_cast<T*>(p);
    return STATTN_OK;
}
// the six input buffers of a minibatch exist twice (sets 0 and 1): one is read by forward / backward while the
// other receives the next minibatch from pinned host memory on the copy stream
inline std::string bset(const stattn_handle* h, const char* name, int set) { return set ? std::string(name) + "#1" : std::string(name); }
inline std::string bcur(const stattn_handle* h, const char* name) { return bset(h, name, h->cur_set); }

inline float* findbuf(stattn_handle* h, const char* name) {
    auto it = h->bufs.find(name);
    return it == h->bufs.end() ? nullptr : static_cast<float*>(it->second.p);
}

// The regions of the flat gradient buffer in the order stattn_backward completes them (api_backward.cpp region_done), each
// [offset of `first`, offset of `before`) in parameter-table order, `before` == nullptr meaning the end of the buffer.  The data-parallel
// all-reduce sums exactly these (comm.cpp); together they must tile [0, nflat) for every option variant: tests/test_dp_gloo.py.
struct GradRegion { const char* first; const char* before; };
constexpr GradRegion GRAD_REGIONS[] = {
    {"ff_logit_lstm_W", nullptr},               // readout + vocabulary projection: final before the reverse scan starts
    {"decoder_W", "decoder_Wcg_att"},           // W, U, b, Wc: need only the reverse scan's per-step factors
    {"decoder_Wcg_att", "ff_logit_lstm_W"},     // the attention weights (after ctxgrad)
    {"ff_state_W", "decoder_W"},                // the context projections and the state initialisers
    {"Wemb", "ff_state_W"},                     // the embedding, last
};
constexpr int N_GRAD_REGIONS = (int)(sizeof(GRAD_REGIONS) / sizeof(GRAD_REGIONS[0]));

// comm.cpp
int comm_reduce_range(stattn_handle* h, size_t off, size_t n);
int comm_backward_begins(stattn_handle* h);
void comm_release(stattn_handle* h);
// region all-reduces of the last backward are (possibly) still running on the side stream and stattn_allreduce_grads
// has not ordered the compute stream behind them: the gradient buffer must not be read, updated from or rewritten
inline bool comm_pending(const stattn_handle* h) { return h->comm && h->comm_covered != 0 && !h->grads_reduced; }

}  // namespace stattn_detail
