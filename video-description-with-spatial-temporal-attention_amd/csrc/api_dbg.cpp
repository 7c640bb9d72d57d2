// Development entry points (csrc/stattn_dbg.h): kernels run and timed in isolation by tests/ and tools/.
#include <cstdlib>
#include "steps.h"
#include "stattn_dbg.h"

extern "C" {

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
    if (kind == 0 || kind == 4 || kind == 5) {
        GemmArgs g;
        gemm_defaults(g); g.split = h->opt.precision != 0;
        g.split = kind == 4;
        g.A = dA; g.lda = transA ? M : K; g.B = dB; g.ldb = transB ? K : N; g.C = dC; g.ldc = N;
        g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.bias = bias ? dbias : nullptr;
        if (add) { g.add = dadd; g.ldadd = N; }
        g.act = act;
        if (kind == 5) {
            // the same product as TWO K-concatenated operand pairs (k < K / 2 from the first, the rest from the second: the
            // readout's joint launch) with a split-K workspace, so that the epilogue-applying reduction runs when it qualifies
            if (transA || K % 64 != 0) return fail(h, STATTN_EINVAL, "paired GEMM: NN or NT, K % 64 == 0");
            float* dws;
            const size_t WSF = (size_t)8 * M * N;
            CHK(getbuf_t(h, "dbg_ws", WSF, &dws));
            g.K = K / 2;
            g.A2 = dA + K / 2; g.lda2 = K; g.K2 = K / 2;
            if (transB) { g.B2 = dB + K / 2; g.ldb2 = K; } else { g.B2 = dB + (size_t)(K / 2) * N; g.ldb2 = N; }
            g.ws = dws; g.ws_floats = WSF;
        }
        if (g.split && !gemm_split_supported(g, transA != 0, transB != 0))
            return fail(h, STATTN_EINVAL, "split kernel: N % 128 == 0, k-contiguous operands 16-byte aligned");
        hipError_t e = launch_gemm(s, g, transA != 0, transB != 0);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_gemm: %s", hipGetErrorString(e));
    } else if (kind == 2 || kind == 6 || kind == 7) {
        // bf16-MFMA kernel: operands rounded to bf16 on the device and made k-contiguous (A [M][K], B [N][K]): a transposed
        // operand goes through the transposing conversion the weight-gradient GEMMs of a bf16 handle use
        if (alpha != 1.f || K % 8 != 0 || (transA && M % 8 != 0)) return fail(h, STATTN_EINVAL, "bf16 kernel: alpha must be 1, K % 8 == 0 (transA: M % 8 == 0)");
        uint16_t *bA, *bB;
        CHK(getbuf_t(h, "dbg_bA", (size_t)M * K, &bA));
        CHK(getbuf_t(h, "dbg_bB", (size_t)K * N, &bB));
        if (transA) HIPCHK(h, launch_transpose_to_bf16(s, dA, 0, M, bA, K, K, M));
        else HIPCHK(h, launch_cvt_bf16(s, dA, bA, (size_t)M * K));
        if (transB) HIPCHK(h, launch_cvt_bf16(s, dB, bB, (size_t)K * N));
        else if (N % 8 == 0) HIPCHK(h, launch_transpose_to_bf16(s, dB, 0, N, bB, K, K, N));
        else HIPCHK(h, launch_cvt_bf16_t(s, dB, N, bB, K, K, N));
        GemmBfArgs g{};
        g.A = bA; g.lda = K; g.B = bB; g.ldb = K; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
        g.bias = bias ? dbias : nullptr;
        if (add) { g.add = dadd; g.ldadd = N; }
        g.act = act; g.rowgroup = 1;
        float* dC2 = nullptr;
        if (kind == 7) {       // split-K of the 256 x 256 kernel (three slices, or what fills the chip if that is more)
            float* dws;
            g.tile = 88; g.kslices = gemm_bf16_8ph_slices(g) > 3 ? gemm_bf16_8ph_slices(g) : 3;
            g.ws_floats = (size_t)g.kslices * M * N;
            CHK(getbuf_t(h, "dbg_ws", g.ws_floats, &dws));
            g.ws = dws;
        }
        if (kind == 6) {
            // two outputs in one launch (GemmBfArgs::n_split): columns >= N / 2 go to a second [M][N / 2] buffer with the
            // second half of the bias; the halves are put side by side again for the caller
            if (N % 128 != 0 || add) return fail(h, STATTN_EINVAL, "split-output GEMM: N % 128 == 0, no add");
            CHK(getbuf_t(h, "dbg_C2", (size_t)M * N / 2, &dC2));
            g.ldc = N / 2; g.n_split = N / 2; g.C2 = dC2; g.bias2 = bias ? dbias + N / 2 : nullptr;
        }
        hipError_t e = launch_gemm_bf16(s, g);
        if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? STATTN_EINVAL : STATTN_EHIP, "dbg_gemm: %s", hipGetErrorString(e));
        if (kind == 6) {
            HIPCHK(h, hipMemcpy2DAsync(C, (size_t)N * 4, dC, (size_t)N * 2, (size_t)N * 2, M, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpy2DAsync(C + N / 2, (size_t)N * 4, dC2, (size_t)N * 2, (size_t)N * 2, M, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
            return STATTN_OK;
        }
    } else if (kind == 3) {
        // row-panel kernel: B repacked on the device (transB: the operand is B^T, packed straight from B [N][K])
        if (transA || alpha != 1.f || !panel_supported(M) || N % 16 || K % 16)
            return fail(h, STATTN_EINVAL, "panel kernel: no transA, alpha must be 1, M <= 512, N and K multiples of 16");
        float* P;
        CHK(getbuf_t(h, "dbg_P", (size_t)K * N, &P));
        CHK(pack(h, dB, transB ? K : N, transB ? 1 : 0, K, N / 16, PN_COLS_PLAIN, P));
        CHK(pack_flush(h));
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
    float ms = 0.f;
    static const char* cold = sw_tool("STATTN_DBG_COLD");   // tools: every timed launch behind a 1 GB fill (operands out of L2 and the Infinity Cache, dirty lines in both)
    if (cold) {
        float* scr;
        CHK(getbuf_t(h, "dbg_cold", (size_t)256 << 20, &scr));
        for (int i = 0; i < iters; ++i) {
            HIPCHK(h, launch_uniform(s, scr, (size_t)256 << 20, 11, 3 + i));
            HIPCHK(h, hipEventRecord(a, s));
            HIPCHK(h, launch_gemm(s, g, transA != 0, transB != 0));
            HIPCHK(h, hipEventRecord(b, s));
            HIPCHK(h, hipEventSynchronize(b));
            float t = 0.f;
            HIPCHK(h, hipEventElapsedTime(&t, a, b));
            ms += t;
        }
    } else {
        HIPCHK(h, hipEventRecord(a, s));
        for (int i = 0; i < iters; ++i) HIPCHK(h, launch_gemm(s, g, transA != 0, transB != 0));
        HIPCHK(h, hipEventRecord(b, s));
        HIPCHK(h, hipEventSynchronize(b));
        HIPCHK(h, hipEventElapsedTime(&ms, a, b));
    }
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
    if (which == 1) return h->path_fwd_rider;
    if (which == 2) return h->path_fwd_panel;
    if (which == 3) return h->path_bwd_rider;
    if (which == 4) return h->path_bwd_panel;
    if (which == 5) return h->path_upd_rider;
    if (which == 6) return h->path_upd_rowwg;
    if (which == 7) return h->path_vocab_stats;
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

}  // extern "C"

// ---- red zones (STATTN_DBG_REDZONE=1; handle.h DevBuf) -------------------------------------------------------------------
namespace {
struct RzEntry { std::string name; int tail; size_t tail_from; };     // tail: 0 = the zone in front of the buffer, 1 = behind it
void rz_collect(const std::string& name, const stattn_detail::DevBuf& b, std::vector<stattn::RedzoneRegion>& regs, std::vector<RzEntry>& who) {
    const size_t rz = stattn_detail::redzone_bytes();
    if (!b.base || !rz) return;
    regs.push_back(stattn::RedzoneRegion{static_cast<const unsigned char*>(b.base), rz});
    who.push_back(RzEntry{name, 0, 0});
    regs.push_back(stattn::RedzoneRegion{static_cast<const unsigned char*>(b.p) + b.used, b.cap - b.used + rz});
    who.push_back(RzEntry{name, 1, b.used});
}
}  // namespace

int stattn_dbg_redzone_enabled(void) { return stattn_detail::redzone_bytes() ? 1 : 0; }

long stattn_dbg_redzone_buffers(const stattn_handle* h) {
    if (!h || !stattn_detail::redzone_bytes()) return 0;
    long n = 0;
    for (auto& kv : h->bufs) if (kv.second.base) ++n;
    for (const stattn_detail::DevBuf* b : {&h->fb_params, &h->fb_grads, &h->fb_rg2, &h->fb_ru2}) if (b->base) ++n;
    return n;
}

int stattn_dbg_redzone_check(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    if (!stattn_detail::redzone_bytes()) return STATTN_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());            // everything in flight on any stream has written what it will write
    std::vector<stattn::RedzoneRegion> regs;
    std::vector<RzEntry> who;
    for (auto& kv : h->bufs) rz_collect(kv.first, kv.second, regs, who);
    rz_collect("(parameters)", h->fb_params, regs, who); rz_collect("(gradients)", h->fb_grads, regs, who);
    rz_collect("(adadelta rg2)", h->fb_rg2, regs, who); rz_collect("(adadelta ru2)", h->fb_ru2, regs, who);
    if (regs.empty()) return STATTN_OK;
    // scratch of the check itself: plain allocations (not DevBuf: the list being scanned must not change under the scan)
    stattn::RedzoneRegion* dregs = nullptr;
    unsigned long long* dbad = nullptr;
    unsigned long long bad = ~0ull;
    hipError_t e = hipMalloc((void**)&dregs, regs.size() * sizeof(stattn::RedzoneRegion));
    if (e == hipSuccess) e = hipMalloc((void**)&dbad, sizeof bad);
    if (e == hipSuccess) e = hipMemcpy(dregs, regs.data(), regs.size() * sizeof(stattn::RedzoneRegion), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dbad, &bad, sizeof bad, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = stattn::launch_redzone_scan(nullptr, dregs, (int)regs.size(), stattn_detail::REDZONE_BYTE, dbad);
    if (e == hipSuccess) e = hipMemcpy(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost);
    if (dregs) (void)hipFree(dregs);
    if (dbad) (void)hipFree(dbad);
    if (e != hipSuccess) return fail(h, STATTN_EHIP, "red-zone scan: %s", hipGetErrorString(e));
    if (bad == ~0ull) return STATTN_OK;
    const size_t ri = (size_t)(bad >> 40), off = (size_t)((bad >> 8) & 0xffffffffull);
    const RzEntry& w = who[ri < who.size() ? ri : 0];
    if (w.tail)
        return fail(h, STATTN_ESTATE, "red zone damaged: buffer '%s' (%zu bytes requested): byte %zu past its end was overwritten (found 0x%02x)",
                    w.name.c_str(), w.tail_from, off, (unsigned)(bad & 0xff));
    return fail(h, STATTN_ESTATE, "red zone damaged: buffer '%s': byte %zu before its start was overwritten (found 0x%02x)",
                w.name.c_str(), stattn_detail::redzone_bytes() - off, (unsigned)(bad & 0xff));
}

// test hook: write one byte `offset` bytes past the requested end (offset >= 0) or before the start (offset < 0) of a named buffer
int stattn_dbg_redzone_poke(stattn_handle* h, const char* name, long offset) {
    if (!h || !name) return STATTN_EINVAL;
    if (!stattn_detail::redzone_bytes()) return fail(h, STATTN_ESTATE, "redzone_poke: STATTN_DBG_REDZONE is not set");
    auto it = h->bufs.find(name);
    if (it == h->bufs.end() || !it->second.base) return fail(h, STATTN_ENOTFOUND, "redzone_poke: no buffer '%s'", name);
    const stattn_detail::DevBuf& b = it->second;
    if (offset >= (long)(b.cap - b.used + stattn_detail::redzone_bytes()) || -offset > (long)stattn_detail::redzone_bytes())
        return fail(h, STATTN_EINVAL, "redzone_poke: offset outside the red zones");
    unsigned char* p = offset >= 0 ? static_cast<unsigned char*>(b.p) + b.used + offset : static_cast<unsigned char*>(b.p) + offset;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemset(p, 0x5A, 1));
    return STATTN_OK;
}
