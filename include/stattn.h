/*
 * stattn.h -- C ABI of libstattn.so: the MI355X-native (gfx950 / CDNA4, hand-written
 * HIP) spatial-temporal-attention LSTM caption decoder.
 *
 * This is the drop-in boundary for ONE path of
 * tuyunbin/Video-Description-with-Spatial-Temporal-Attention: the decoder graph that
 * model_attention.py builds with Theano (init_params / build_model / build_sampler ->
 * f_init, f_next / f_grad_shared / f_update).  Each entry point names the reference
 * interface it replaces (file:line relative to the reference repo).  The reference
 * binds compiled `theano.function` objects from Python; the matching binding for this
 * library is a ctypes stub (INTEGRATION.md, and stattn/_native.py in this repo).
 *
 * Conventions
 *  - plain pointers and sizes only; every `const float*` / `const int64_t*` argument
 *    is a HOST pointer to C-contiguous data unless the name ends in `_dev`;
 *    dtypes are strict like Theano's: int64 words, float32 everything else.
 *  - every function returns 0 on success or a negative STATTN_E* code and never
 *    aborts; stattn_last_error() gives the message.
 *  - one handle = one GPU + one stream; calls on a handle are serialised by the
 *    caller; distinct handles are independent (one per rank).
 *  - the library owns all device memory; host buffers are borrowed for the call.
 */
#ifndef STATTN_H
#define STATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STATTN_OK 0
#define STATTN_EINVAL (-1)   /* bad argument / unsupported option value        */
#define STATTN_EHIP (-2)     /* a HIP runtime call failed                       */
#define STATTN_ESTATE (-3)   /* call order violated (e.g. backward w/o forward) */
#define STATTN_ENOTFOUND (-4)

typedef struct stattn_handle stattn_handle;

/* The option keys the hot path consumes (model_attention.py:1079-1096 `model_options`,
 * config.py:17-48).  Unsupported-by-the-reference values are rejected exactly where
 * the reference graph is broken: n_layers_init must be 0 (model_attention.py:546-548),
 * use_dropout must be 1 (:479-481), encoder must be 'none' (:524-536),
 * ctxg_dim == dim == ctxglm_dim (ff_global is commented out, :553-554). */
typedef struct stattn_options {
    int32_t dim;        /* LSTM units D            (multiple of 64)              */
    int32_t dim_word;   /* word embedding E        (multiple of 64)              */
    int32_t n_words;    /* vocabulary V            (any)                         */
    int32_t ctxg_dim;   /* global feature dim      (must equal dim)              */
    int32_t ctxl_dim;   /* local (RCNN) feature F  (multiple of 32)              */
    int32_t ctxm_dim;   /* motion (C3D) feature    (multiple of 32)              */
    int32_t selector;   /* model_attention.py:432                                */
    int32_t use_dropout;/* must be 1                                             */
    int32_t prev2out;   /* :689                                                  */
    int32_t ctx2out;    /* :691                                                  */
    int32_t lt_mode;    /* local-temporal projection CL.Wclt (:416):
                           0 = one MFMA GEMM per step, the reference's order;
                           1 = L.Wclt pre-projected once per batch, alpha-weighted
                               sum per step (same maths, different summation order) */
    int32_t precision;  /* 0 = fp32 everywhere (the parity configuration);
                           1 = bf16-MFMA path (BASELINE configs[3]): the once-per-batch context
                               projections, the x projection and the readout GEMMs take bf16
                               operands with fp32 accumulation, and the projected region tensors
                               L / PL / LW are stored in bf16 (the per-step attention kernel reads
                               half the bytes).  Recurrent matmuls, softmaxes and the LSTM stay
                               fp32.  lt_mode 1.  stattn_backward on such a handle is the fp32 backward
                               pass evaluated at the stored (bf16) activations (its GEMMs as in precision 2):
                               mixed-precision training.
                               Accuracy: ~1e-3 on attention weights, ~2e-2 on logits, gradients within
                               a few per cent of their scale.
                           2 = fp32 results with the LDS-tiled GEMMs (context projections, x projection,
                               readout, and all their weight / input gradients) computed on the bf16
                               matrix cores: each fp32 operand is split EXACTLY into three bf16 terms
                               and six of the nine term products are accumulated in fp32 (the dropped
                               three are below one fp32 rounding of the product).  Same accuracy
                               against float64 as precision 0 (tests/test_gpu_split.py), ~2x the GEMM
                               rate.  Everything else is the precision-0 path; both lt_modes.  (Operands must
                               be finite: an infinite value yields NaN in the split.)              */
    int32_t reserved[4];
} stattn_options;

/* ---- lifecycle ---------------------------------------------------------------- */
/* Replaces: Attention.init_tparams / common.init_tparams (model_attention.py:70-78,
 * common.py:103-107): creates the device-resident parameter set (zero-filled).
 * `stream`: a hipStream_t to run on (e.g. torch's current stream) or NULL to create one. */
int stattn_create(const stattn_options* opt, int device, void* stream, stattn_handle** out);
void stattn_destroy(stattn_handle* h);
const char* stattn_last_error(const stattn_handle* h); /* h may be NULL: last create error */
const char* stattn_version(void);
int stattn_sync(stattn_handle* h);                     /* hipStreamSynchronize             */

/* ---- parameters: theano.shared get_value/set_value, zipp/unzip (common.py:78-87) -- */
int stattn_param_count(const stattn_handle* h);
const char* stattn_param_name(const stattn_handle* h, int i);
/* dims[0..1], ndim in {0,1,2}; order = init_params dict order (model_attention.py:518-581) */
int stattn_param_shape(const stattn_handle* h, int i, int64_t dims[2], int* ndim);
int stattn_set_param(stattn_handle* h, const char* name, const float* src, size_t n);
int stattn_get_param(stattn_handle* h, const char* name, float* dst, size_t n);
/* flat device buffers (padded layout, identical on every rank) for the data-parallel
 * all-reduce and for tools: *n = number of floats. */
int stattn_param_buffer_dev(stattn_handle* h, void** ptr_dev, size_t* n);
int stattn_grad_buffer_dev(stattn_handle* h, void** ptr_dev, size_t* n);
int stattn_get_grad(stattn_handle* h, const char* name, float* dst, size_t n);

/* use_noise shared scalar (model_attention.py:585, 1248, 1311): 0 = eval (x0.5), 1 = Bernoulli(.5) */
int stattn_set_use_noise(stattn_handle* h, float use_noise);
int stattn_set_seed(stattn_handle* h, uint64_t seed);
/* Test hook: supply the three dropout multiplier tensors instead of drawing them:
 * dp (t,m,3D) on the i/f/o pre-activations (:444-447), d1 (t,m,D) on proj_h (:685),
 * d2 (t,m,E) on tanh(logit) (:696).  NULL restores the internal generator. */
int stattn_set_dropout_masks(stattn_handle* h, const float* dp, const float* d1, const float* d2,
                             int t, int m);

/* ---- sampler: build_sampler -> f_init, f_next (model_attention.py:719-850) -------- */
/* f_init(ctxg, ctxg_mask) -> [ctxg, h0, c0]  (:791-795).  ctxg (T,D), mask (T,). */
int stattn_f_init(stattn_handle* h, const float* ctxg, const float* ctxg_mask, int T,
                  float* out_h0, float* out_c0);
/* f_next(x, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask, h, c)
 *   -> [next_probs (m,V), next_sample (m,), h' (m,D), c' (m,D)]   (:845-848)
 * x (m,) int64 (-1 = first word), ctxl (T,K,F), ctxm (T,F).  ctxl_mask / ctxm_mask are
 * accepted and ignored exactly like the reference (on_unused_input='ignore').
 * Like the reference graph, a call that passes ctxg / ctxl / ctxm uploads and re-projects them
 * (F->D, :782-785, :322-326) EVERY time -- the library never guesses whether a host array changed.
 * The explicit fast path: stattn_set_video() stages and projects one video once; f_next calls that
 * pass ctxg = ctxl = ctxm = NULL (T, K still given) then run on that resident video.  The
 * projections of the resident video are redone automatically after stattn_set_param / stattn_update.
 * x outside [-1, V) is rejected (the reference raises IndexError).
 * Optional outputs (NULL to skip) for the parity bar: out_alphal (m,T,K), out_alphag/m/lt
 * (m,T), out_logits (m,V). */
int stattn_f_next(stattn_handle* h, const int64_t* x, int m,
                  const float* ctxg, const float* ctxg_mask,
                  const float* ctxl, const float* ctxl_mask,
                  const float* ctxm, const float* ctxm_mask, int T, int K,
                  const float* h_in, const float* c_in,
                  float* out_probs, int64_t* out_sample, float* out_h, float* out_c,
                  float* out_alphal, float* out_alphag, float* out_alpham, float* out_alphalt,
                  float* out_logits);
/* Stage one video for the sampler: ctxg (T,D), ctxl (T,K,F), ctxm (T,F) -> HBM, projected once.
 * Hoists what gen_sample's loop (model_attention.py:896-903) recomputes inside every f_next call. */
int stattn_set_video(stattn_handle* h, const float* ctxg, const float* ctxl, const float* ctxm, int T, int K);

/* Batched beam search: gen_sample (model_attention.py:852-994) for `nvid` videos at once with the whole
 * bookkeeping on the device (candidate costs hyp_score - log p, top (k - dead_k), hypothesis / state gather,
 * :921-985) -- one fixed kernel sequence per word, no host round trip; the videos' F->D projections are done
 * once.  ctxg (nvid,T,D), ctxg_mask (nvid,T), ctxl (nvid,T,K,F), ctxm (nvid,T,F); 1 <= k <= 8.
 * Per video the hypotheses come back in gen_sample's order (finished ones in order of death, then the live
 * ones): out_tokens (nvid,k,maxlen) int64, -1 padded; out_scores, out_lens (nvid,k); out_count (nvid).
 * suppress_eos != 0 forbids word 0 so that every hypothesis runs maxlen steps (benchmarks).
 * Host features are staged on every call; with ctxg = ctxg_mask = ctxl = ctxm = NULL the videos staged by
 * stattn_beam_stage (same nvid, T, K) are decoded (inputs resident in HBM: benchmarks, re-decoding after an
 * update).  k = 1 is gen_sample's greedy mode (arg-max word, :896-918 with stochastic=False). */
int stattn_beam_stage(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                      const float* ctxm, int T, int K);
int stattn_beam_search(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                       const float* ctxm, int T, int K, int k, int maxlen, int suppress_eos,
                       int64_t* out_tokens, float* out_scores, int32_t* out_lens, int32_t* out_count);

/* gen_sample(stochastic=True) (model_attention.py:863-918) for 1 .. 16 videos at once, on the device: every word is a draw
 * from the next-word distribution (the reference draws with Theano's MRG stream at :841; here Gumbel-max on the logits with
 * the library's counter-based generator, seeded by stattn_set_seed and a per-call counter), the caption ends with the first
 * <eos> -- which is part of the sample -- or after maxlen words.  out_tokens (nvid, maxlen) padded with -1, out_lens
 * (nvid,), out_scores (nvid,) = the SUM of the drawn words' probabilities, as the reference computes it (:916).
 * Features as in stattn_beam_search (all NULL: the videos staged by stattn_beam_stage).  stattn_beam_final_state works
 * after it as after a beam search. */
int stattn_sample_search(stattn_handle* h, int nvid, const float* ctxg, const float* ctxg_mask, const float* ctxl,
                         const float* ctxm, int T, int K, int maxlen,
                         int64_t* out_tokens, float* out_scores, int32_t* out_lens);

/* gen_sample's third and fourth return values (next_state, next_memory, :994) for the videos of the last
 * stattn_beam_search: out_h / out_c (nvid,k,D), the first out_rows[v] rows of video v are valid -- the state outputs of
 * the f_next call that ended the video's loop (every live hypothesis at that point), or the gathered states of the
 * hypotheses still live after maxlen words. */
int stattn_beam_final_state(stattn_handle* h, float* out_h, float* out_c, int32_t* out_rows);

/* ---- training graph: build_model / f_log_probs / f_grad_shared (:583-717, 1126, 1207) -- */
/* Stage one minibatch in HBM: prepare_data()'s 8-tuple (data_engine.py:258-337).
 * x (t,m) int64, mask (t,m), ctxg (m,T,D), mask_ctxg (m,T), ctxl (m,T,K,F),
 * mask_ctxl (m,T,K) [ignored], ctxm (m,T,F), mask_ctxm (m,T) [ignored].
 * Words outside [0, V) are rejected (STATTN_EINVAL; the reference raises IndexError at :613). */
int stattn_set_batch(stattn_handle* h, const int64_t* x, const float* mask, int t, int m,
                     const float* ctxg, const float* mask_ctxg,
                     const float* ctxl, const float* mask_ctxl,
                     const float* ctxm, const float* mask_ctxm, int T, int K);
/* Double-buffered staging for a training loop (the H2D of ctxl is 218 MB per batch of 64: SURVEY 8f rank 2).
 * stattn_prefetch_batch copies the NEXT minibatch into the shadow buffer set on a private copy stream and returns
 * immediately when the host arrays are pinned (stattn_host_alloc); stattn_swap_batch makes it current -- the compute
 * stream waits on the copy event, the host never blocks.  The arrays must stay untouched until the swap. */
int stattn_host_alloc(size_t bytes, void** out);   /* pinned host memory for prepare_data's output arrays */
int stattn_host_free(void* p);
int stattn_prefetch_batch(stattn_handle* h, const int64_t* x, const float* mask, int t, int m,
                          const float* ctxg, const float* mask_ctxg,
                          const float* ctxl, const float* mask_ctxl,
                          const float* ctxm, const float* mask_ctxm, int T, int K);
int stattn_swap_batch(stattn_handle* h);
/* Forward of build_model on the staged batch (asynchronous on the handle's stream):
 * prologue (ff_local/ff_motion, attention pre-projections, init state, x projection),
 * t decoder steps, readout, vocabulary softmax, masked NLL. */
int stattn_forward_train(stattn_handle* h);
/* Results of the last forward.  Any pointer may be NULL.  cost (m,) [= -f_log_probs],
 * probs (t*m,V), alphal (t,m,T,K), alphag/alpham/alphalt (t,m,T), logits (t*m,V).
 * logits must be read before stattn_backward (it reuses the buffer for d loss / d logit): STATTN_ESTATE after. */
int stattn_get_forward(stattn_handle* h, float* cost, float* probs,
                       float* alphal, float* alphag, float* alpham, float* alphalt, float* logits);
/* per-step state of the last forward, for tests: hs, cs (t,m,D), ctx (t,m,D) */
int stattn_get_states(stattn_handle* h, float* hs, float* cs, float* ctx);

/* ---- gradients and optimizer: f_grad_shared / f_update (model_attention.py:1129-1147, 1193-1209;
 *      common.py:178-195) ------------------------------------------------------------- */
/* Hand-written BPTT over the last stattn_forward_train: fills the flat gradient buffer (same
 * layout as the parameters) with d/dtheta of
 *     nll_scale * sum_b cost[b]  +  alpha_c * sum_{4 alphas} mean_{T(,K)} sum_b (1 - sum_t alpha)^2 .
 * nll_scale = 1/m reproduces cost.mean() (:1129); a data-parallel rank passes 1/B_global and the
 * all-reduce SUMS the buffers (the regulariser is a batch sum, :1140-1143).  The L2 term
 * (:1130-1136) is batch-independent and is applied once, in stattn_update.  Both lt_modes (the derivative of the
 * per-step CL.Wclt of lt_mode 0 is evaluated in the hoisted form: same function, section 8 of DESIGN.md); on a bf16
 * handle it is the fp32 backward pass evaluated at the stored bf16 activations. */
int stattn_backward(stattn_handle* h, float nll_scale, float alpha_c);
/* value of the loss above + decay_c * sum ||theta||^2, after stattn_backward */
int stattn_get_loss(stattn_handle* h, float nll_scale, float decay_c, float* loss);
/* g += 2 decay_c theta; global-norm clip to clip_c (:1194-1203); Adadelta running averages and
 * parameter update (common.py:183-191; the reference ignores `lr`, Appendix C.8).  One update per
 * stattn_backward: STATTN_ESTATE without a fresh gradient. */
int stattn_update(stattn_handle* h, float decay_c, float clip_c);
int stattn_reset_optimizer(stattn_handle* h);

/* ---- data parallel over the GPUs of one node (the reference is single-process: its
 *      counterpart is the single f_grad_shared / f_update call site, model_attention.py:1259, 1278) ------------
 * One process (or thread) per GPU, one handle each, rows (videos) of the caption batch sharded, weights replicated.
 * Exactness rule: every rank passes nll_scale = 1 / B_global to stattn_backward; the regulariser is a batch SUM;
 * ranks SUM their gradient buffers; the L2 term and the clip are applied once, after the reduce, in stattn_update.
 * RCCL (librccl.so.1) is loaded on the first call; nothing here needs torch. */
#define STATTN_COMM_ID_BYTES 128
/* rank 0 creates the rendezvous token (ncclGetUniqueId) and hands the bytes to the other ranks by any host means */
int stattn_comm_unique_id(void* id_out /* STATTN_COMM_ID_BYTES */);
/* collective: every rank calls it with the same token; binds an RCCL communicator to the handle's GPU */
int stattn_comm_init(stattn_handle* h, int rank, int nranks, const void* id);
int stattn_comm_destroy(stattn_handle* h);
int stattn_comm_info(const stattn_handle* h, int* rank, int* nranks);   /* nranks = 0: no communicator */
/* overlap (default on): stattn_backward starts summing each region of the gradient buffer over the ranks on a side
 * stream as soon as it is final (readout gradients before the reverse scan, decoder_* / ff_* / Wemb while the remaining
 * weight-gradient GEMMs run).  0: one all-reduce of the whole buffer inside stattn_allreduce_grads.  2: overlap even
 * in a one-rank communicator (a sum over one rank: how the side-stream path is tested on a single-GPU box). */
int stattn_comm_set_overlap(stattn_handle* h, int enable);
/* SUM the flat gradient buffer over the ranks (finishes the overlapped reduce); stream-ordered between
 * stattn_backward and stattn_update.  Without a communicator (or with one rank) it only marks the gradient final.
 * stattn_update refuses to run on an un-reduced gradient when the handle belongs to a multi-rank communicator.
 * Rule while regions are in flight (overlap on, between stattn_backward and this call): the gradient buffer belongs
 * to the side stream -- stattn_get_grad and stattn_update return STATTN_ESTATE, a second stattn_backward first waits
 * for the collectives, and a consumer of stattn_grad_buffer_dev must call this function before touching the buffer. */
int stattn_allreduce_grads(stattn_handle* h);
/* What a scaling run needs to prove about itself: ranks of the communicator (0: none), the overlap mode in force,
 * how many regions the last stattn_backward handed to the side stream, and the milliseconds the compute stream spent
 * inside the last stattn_allreduce_grads (HIP events on the compute stream: the whole collective without overlap,
 * only the un-hidden tail with it).  Any pointer may be NULL; exposed_ms synchronises with that all-reduce. */
int stattn_comm_stats(stattn_handle* h, int* nranks, int* overlap, int* regions, float* exposed_ms);
/* path of the RCCL shared object that was bound ("" before the first stattn_comm_* call or when none was found):
 * one already loaded in the process (torch's) is preferred over a second copy from the loader path */
const char* stattn_comm_library_path(void);
/* copy rank `root`'s parameters to every rank (replicas must start identical) */
int stattn_broadcast_params(stattn_handle* h, int root);
/* SUM `n` host floats over the ranks in place (reported cost, counters); n <= 64 */
int stattn_allreduce_scalars(stattn_handle* h, float* vals, int n);

/* ---- measurement hooks (bench.py's per-kernel rooflines) -------------------------------------------- */
/* Average duration (ms) of the named kernel class over the last stattn_forward_train (or the stattn_beam_search /
 * f_next calls since profiling was switched on; a profiled beam search launches its words eagerly instead of replaying
 * its hipGraphs): 0 = spatial attention, 1 = state projections, 2 = local-temporal GEMM, 3 = temporal fuse, 4 = lstm,
 * 5 = prologue scope (sum), 6 = readout scope (training: sum of the batched readout; beam search: readout + vocabulary
 * launch [+ softmax] of one word), 7 = every plain (NN) launch of the LDS-tiled GEMM in the forward pass, 8 = beam search:
 * candidate selection + beam update of one word; 9 + i = the i-th plain GEMM launch alone (i < 16, in launch order:
 * ff_local, ff_motion, pctxg, pctxl, pctxm, L.Wclt [lt_mode 1], x projection, readout 1, readout 2 [ctx2out], logits);
 * over the last stattn_backward: 25 + i = the i-th LDS-tiled GEMM launch of the pass (i < 24, launch order: da, readout
 * weight gradients, readout input gradients, ... ), 49 .. 55 = lstm_bwd, panel dctx (|dhU), [temporal_bwd: part of
 * spatial_bwd since round 3, no launches], spatial_bwd, reduce_T, panel dhW (one launch per reverse-scan step each) and
 * the deferred ctxgrad kernel. */
int stattn_set_profiling(stattn_handle* h, int enable);
int stattn_get_kernel_ms(stattn_handle* h, int which, float* ms_avg, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* STATTN_H */
