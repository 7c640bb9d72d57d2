/* Development entry points of libstattn.so: kernels run and timed in isolation by tests/ and tools/.  NOT part of the
 * drop-in surface (include/stattn.h) and not installed with it. */
#ifndef STATTN_DBG_H
#define STATTN_DBG_H
#include "../../include/stattn.h"
#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] = act(alpha * op(A).op(B) + bias[n] + add[m,n]);  host pointers.
 * transA: A given as [K,M]; transB: B given as [N,K]; act: 0 none, 1 tanh.
 * Runs the LDS-tiled fp32 MFMA kernel (kind=0), the register-streaming skinny
 * kernel (kind=1), the bf16-MFMA kernel (kind=2: no transA, alpha = 1, K % 8 == 0) or the row-panel kernel of the
 * per-step GEMMs (kind=3: M <= 512, N % 16 == 0, K % 16 == 0, no transA, alpha = 1; B is repacked on the device).  Constraints: N % 64 == 0, K % 16 == 0 (kind 1: K % 16 == 0, no trans). */
int stattn_dbg_gemm(stattn_handle* h, int kind, int transA, int transB, int M, int N, int K,
                    float alpha, const float* A, const float* B, const float* bias,
                    const float* add, int act, float* C);
/* Time `iters` launches of the big GEMM on device-resident random data; returns the
 * average milliseconds per launch measured with HIP events on the handle's stream. */
int stattn_dbg_time_gemm(stattn_handle* h, int transA, int transB, int M, int N, int K,
                         int iters, float* ms_per_launch);
/* Debug counters.  which = 0: hipGraph replays (two decoded words each) in the last stattn_beam_search
 * -- 0 means the word sequence was launched eagerly (capture refused, profiling on, STATTN_BEAM_NOGRAPH).
 * which = 1 / 2: decoder steps since the last stattn_forward_train began whose attention launch carried the h.U rider /
 * that ran on the row-panel kernels; which = 3 / 4: the same for the reverse steps of the last stattn_backward (dhU rider);
 * which = 5: words of the last stattn_beam_search / stattn_sample_search whose bookkeeping rode in the next word's attention launch. */
long stattn_dbg_counter(const stattn_handle* h, int which);
/* The bf16-MFMA kernel (stattn_dbg_gemm kind=2 checks it: operands are rounded to bf16 on the device,
 * fp32 accumulation) on device-resident random data, bf16 output.  tile: 0 = the launcher's choice,
 * 11 / 21 / 22 = register-staged workgroup tile (64*TM) x (64*TN), 84 = 256 x 128 with direct global->LDS
 * staging (edge-free shapes only) -- the LDS tile size sweep of BASELINE configs[3]. */
int stattn_dbg_time_gemm_bf16(stattn_handle* h, int M, int N, int K, int tile, int iters, float* ms_per_launch);
/* Same for the register-streaming skinny kernel: `nseg` segments of [M,K].[K,N]; variant 0 = product kernel,
 * other values are reserved. */
int stattn_dbg_time_skinny(stattn_handle* h, int M, int N, int K, int nseg, int variant, int iters,
                           float* ms_per_launch);

/* The regions of the flat gradient buffer that stattn_backward hands to the data-parallel all-reduce, in completion order
 * (csrc/handle.h GRAD_REGIONS), for a configuration: offsets / lengths in floats, *nflat = the buffer's length.  Host only. */
int stattn_dbg_grad_regions(const stattn_options* o, int max_regions, size_t* offsets, size_t* lengths, int* n_regions, size_t* nflat);

/* Red zones (environment STATTN_DBG_REDZONE=1 when the process starts; csrc/handle.h): every device buffer of the library then
 * carries 4 KiB of canary bytes on both sides.  _enabled: 1 when the mode is on.  _buffers: buffers currently guarded.
 * _check: synchronises the device and scans every canary byte; STATTN_OK when all are intact, otherwise STATTN_ESTATE with
 * stattn_last_error naming the buffer and the first damaged byte.  The Python binding calls it after EVERY library call in this mode.
 * _poke: test hook, damages one canary byte of a named buffer (offset >= 0: bytes past its end, < 0: before its start). */
int stattn_dbg_redzone_enabled(void);
long stattn_dbg_redzone_buffers(const stattn_handle* h);
int stattn_dbg_redzone_check(stattn_handle* h);
int stattn_dbg_redzone_poke(stattn_handle* h, const char* name, long offset);

#ifdef __cplusplus
}
#endif
#endif /* STATTN_DBG_H */
