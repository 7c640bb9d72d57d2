// Probe of the wide row-panel kernels (tools only):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSTATTN_PROBES [-DPW_VARIANT=n] tools/panelw_probe.hip -o tools/bin/panelw_probe
// Times the configs[4] per-word launches (160 rows): state projections 160 x 8192 x 1024, LSTM 160 x 4096 x 1536, readout
// 160 x 512 x 2048, logits 160 x 12032 x 512 with the statistics epilogue -- back to back, 50 launches each, HIP events.
// PW_VARIANT: 0 product, 1 no activation loads, 2 no weight loads, 3 no MFMAs, 4 no loads at all.
#include "../video-description-with-spatial-temporal-attention_amd/csrc/panelw.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace stattn {
// what the probe uses from panel.hip
size_t packed_rows_floats(int M, int K) { return (size_t)((M + 15) / 16) * 16 * K; }
void pn_seg_defaults(PnSeg& s) { s = PnSeg{}; s.scale = 1.f; }
}
using namespace stattn;

__global__ void fillk(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * 0.05f;
    }
}
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

static float* dev(size_t n, unsigned seed) {
    float* p = nullptr;
    if (hipMalloc(&p, n * 4) != hipSuccess) { printf("alloc failed\n"); exit(1); }
    hipLaunchKernelGGL(fillk, dim3(1024), dim3(256), 0, 0, p, n, seed);
    return p;
}

template <class F>
static float timeit(F f, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

#include <algorithm>
template <class F>
static void timeline(const char* name, F launch, int nblk) {
    long long* d; (void)hipMalloc(&d, (size_t)nblk * 64);
    (void)hipMemset(d, 0, (size_t)nblk * 64);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(stattn::pw_probe), &d, sizeof d);
    launch(); launch();
    (void)hipDeviceSynchronize();
    std::vector<long long> st((size_t)nblk * 8);
    (void)hipMemcpy(st.data(), d, (size_t)nblk * 64, hipMemcpyDeviceToHost);
    long long* z = nullptr; (void)hipMemcpyToSymbol(HIP_SYMBOL(stattn::pw_probe), &z, sizeof z);
    long long t0 = 1LL << 62, t3 = 0;
    std::vector<double> skew, loop, red, epi;
    for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, st[b * 8]); t3 = std::max(t3, st[b * 8 + 3]); }
    for (int b = 0; b < nblk; ++b) { skew.push_back((st[b * 8] - t0) / 100.0); loop.push_back((st[b * 8 + 1] - st[b * 8]) / 100.0);
                                     red.push_back((st[b * 8 + 2] - st[b * 8 + 1]) / 100.0); epi.push_back((st[b * 8 + 3] - st[b * 8 + 2]) / 100.0); }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("  timeline %-12s span %6.2f us | start skew med %5.2f max %5.2f | loop med %5.2f max %5.2f | reduce med %5.2f max %5.2f | epilogue med %5.2f max %5.2f\n",
           name, (t3 - t0) / 100.0, med(skew), mx(skew), med(loop), mx(loop), med(red), mx(red), med(epi), mx(epi));
    (void)hipFree(d);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 160, D = 1024, E = 512, Vp = 12032;
    float* h_pk = dev(packed_rows_floats(M, D), 1); float* ctx_pk = dev(packed_rows_floats(M, D), 2);
    float* emb_pk = dev(packed_rows_floats(M, E), 3); float* a1_pk = dev(packed_rows_floats(M, E), 4);
    float* pWdU = dev((size_t)8 * D * D, 5); float* pWc = dev((size_t)4 * D * D, 6); float* pW = dev((size_t)E * 4 * D, 7);
    float* pWl = dev((size_t)2 * D * E, 8); float* pWo = dev((size_t)E * Vp, 9);
    float* C = dev((size_t)M * 8 * D, 10); float* lg = dev((size_t)M * Vp, 11); float* stats = dev((size_t)M * (Vp / 32) * PN_STATS_REC, 12);
    float* hc = dev((size_t)M * D * 8, 13); float* dp = dev((size_t)M * 3 * D, 14); float* bias = dev(Vp, 15);
    CK(hipDeviceSynchronize());
    PnArgs hp{}; hp.M = M; hp.nseg = 2;
    for (int i = 0; i < 2; ++i) { pn_seg_defaults(hp.seg[i]); hp.seg[i].npairs = 1; hp.seg[i].p[0] = PnPair{h_pk, D, pWdU + (size_t)i * 4 * D * D, D, 1};
                                  hp.seg[i].C = C + (size_t)i * 4 * D; hp.seg[i].ldc = 8 * D; hp.seg[i].N = 4 * D; }
    PnArgs ro{}; ro.M = M; ro.nseg = 1; pn_seg_defaults(ro.seg[0]); ro.seg[0].npairs = 2; ro.seg[0].p[0] = PnPair{h_pk, D, pWl, D, 1};
    ro.seg[0].p[1] = PnPair{ctx_pk, D, pWl + (size_t)D * E, D, 1}; ro.seg[0].C = C; ro.seg[0].ldc = E; ro.seg[0].N = E; ro.seg[0].act = 1; ro.seg[0].Cpk = a1_pk;
    PnArgs lo{}; lo.M = M; lo.nseg = 1; pn_seg_defaults(lo.seg[0]); lo.seg[0].npairs = 1; lo.seg[0].p[0] = PnPair{a1_pk, E, pWo, E, 1};
    lo.seg[0].C = lg; lo.seg[0].ldc = Vp; lo.seg[0].N = Vp; lo.seg[0].bias = bias;
    PnArgs ls = lo; ls.seg[0].stats = stats; ls.seg[0].stats_V = 12000; ls.seg[0].stats_kb = 5; ls.seg[0].stats_skip0 = 1;
    LstmPnArgs la{}; la.npairs = 2; la.p[0] = PnPair{ctx_pk, D, pWc, D, 1}; la.p[1] = PnPair{emb_pk, E, pW, E, 1};
    la.pre_add = C + 4 * D; la.ldpre = 8 * D; la.bias = bias; la.dp = dp; la.lddp = 3 * D; la.h_prev = hc; la.c_prev = hc + (size_t)M * D;
    la.h_out = hc + (size_t)2 * M * D; la.c_out = hc + (size_t)3 * M * D; la.d1_scalar = 0.5f; la.hd_out = hc + (size_t)4 * M * D; la.M = M; la.D = D;
    struct { const char* name; double flops; float us; } r[5];
    r[0] = {"state_proj 8192x1024", 2.0 * M * 8 * D * D, timeit([&] { launch_panel_wide(0, hp); }, 50)};
    r[1] = {"lstm 4096x1536", 2.0 * M * 4 * D * (D + E), timeit([&] { launch_lstm_panel_wide(0, la); }, 50)};
    r[2] = {"readout 512x2048", 2.0 * M * E * 2 * D, timeit([&] { launch_panel_wide(0, ro); }, 50)};
    r[3] = {"logits 12032x512", 2.0 * M * Vp * E, timeit([&] { launch_panel_wide(0, lo); }, 50)};
    r[4] = {"logits + stats", 2.0 * M * Vp * E, timeit([&] { launch_panel_wide(0, ls); }, 50)};
    CK(hipDeviceSynchronize());
    printf("variant %d, %d rows\n", PW_VARIANT, M);
    timeline("state_proj", [&] { launch_panel_wide(0, hp); }, 256); timeline("logits+stats", [&] { launch_panel_wide(0, ls); }, 376);
    timeline("logits", [&] { launch_panel_wide(0, lo); }, 376); timeline("readout", [&] { launch_panel_wide(0, ro); }, 16 * 5);
    timeline("lstm", [&] { launch_lstm_panel_wide(0, la); }, 256);
    for (int Kx = 256; Kx <= 1024; Kx *= 2) {      // fixed cost vs per-step cost of the state-projection launch: K = 256, 512, 1024
        PnArgs q = hp;
        for (int i = 0; i < 2; ++i) q.seg[i].p[0].K = Kx;
        printf("  state_proj K = %4d        %7.1f us\n", Kx, timeit([&] { launch_panel_wide(0, q); }, 50));
    }
    for (int i = 0; i < 5; ++i) printf("  %-22s %7.1f us  %6.1f TFLOP/s (%.2f of 157.3)\n", r[i].name, r[i].us, r[i].flops / r[i].us * 1e-6, r[i].flops / r[i].us * 1e-6 / 157.3);
    return 0;
}
