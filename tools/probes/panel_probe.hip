// Timeline probe of the row-panel kernels (tools only, not part of the product):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSTATTN_PROBES [-DPN_VARIANT=n] tools/panel_probe.hip -o /tmp/panel_probe && /tmp/panel_probe
// Emulates one decoder step at configs[1] (64 rows, D = 1024): [state projections] -> 184 MB stream (stands in for the
// attention kernel: evicts L2 / Infinity Cache like the real step) -> [LSTM GEMM + cell]; prints, over the workgroups
// of a launch, start skew, main-loop, reduction and epilogue durations (100 MHz wall clock) and the event time.
#include "../video-description-with-spatial-temporal-attention_amd/csrc/panel.hip"

namespace stattn {   // the wide-kernel entry points panel.hip links against (panelw.hip is not part of this probe: 64 rows never route there)
bool panel_wide_supported(const PnArgs&) { return false; }
bool lstm_panel_wide_supported(const LstmPnArgs&) { return false; }
hipError_t launch_panel_wide(hipStream_t, const PnArgs&) { return hipErrorInvalidValue; }
hipError_t launch_lstm_panel_wide(hipStream_t, const LstmPnArgs&) { return hipErrorInvalidValue; }
int panel_wide_tile_cols() { return 32; }
}

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

using namespace stattn;

__global__ void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.f) dst[0] = acc;
}
__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 32768.0f - 1.0f) * 0.05f;
    }
}
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

static void report(const char* name, const std::vector<long long>& st, int nblk, float ev_us) {
    long long t0min = 1LL << 62, t3max = 0;
    std::vector<double> skew, loop, red, epi;
    for (int b = 0; b < nblk; ++b) { t0min = std::min(t0min, st[b * 8]); t3max = std::max(t3max, st[b * 8 + 3]); }
    for (int b = 0; b < nblk; ++b) {
        skew.push_back((st[b * 8] - t0min) / 100.0); loop.push_back((st[b * 8 + 1] - st[b * 8]) / 100.0);
        red.push_back((st[b * 8 + 2] - st[b * 8 + 1]) / 100.0); epi.push_back((st[b * 8 + 3] - st[b * 8 + 2]) / 100.0);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("%-10s blocks %4d  event %6.2f us  span %6.2f us | start skew med %5.2f max %5.2f | loop med %5.2f max %5.2f | reduce med %5.2f max %5.2f | epilogue med %5.2f max %5.2f\n",
           name, nblk, ev_us, (t3max - t0min) / 100.0, med(skew), mx(skew), med(loop), mx(loop), med(red), mx(red), med(epi), mx(epi));
}

int main() {
    const int M = 64, D = 1024;
    float *h, *hn, *cn, *c, *ctx, *sproj, *preh, *xproj, *dp, *mask, *gates, *hd, *pWd, *pU, *pWc, *big;
    const size_t nbig = (size_t)46 << 20;   // floats: 184 MB
    CK(hipMalloc(&h, M * D * 4)); CK(hipMalloc(&hn, M * D * 4)); CK(hipMalloc(&cn, M * D * 4)); CK(hipMalloc(&c, M * D * 4));
    CK(hipMalloc(&ctx, M * D * 4)); CK(hipMalloc(&sproj, M * 4 * D * 4)); CK(hipMalloc(&preh, M * 4 * D * 4));
    CK(hipMalloc(&xproj, M * 4 * D * 4)); CK(hipMalloc(&dp, M * 3 * D * 4)); CK(hipMalloc(&mask, M * 4));
    CK(hipMalloc(&gates, M * 4 * D * 4)); CK(hipMalloc(&hd, M * D * 4));
    CK(hipMalloc(&pWd, (size_t)4 * D * D * 4)); CK(hipMalloc(&pU, (size_t)4 * D * D * 4)); CK(hipMalloc(&pWc, (size_t)4 * D * D * 4));
    CK(hipMalloc(&big, nbig * 4));
    float* all[] = {h, c, ctx, xproj, dp, pWd, pU, pWc};
    size_t ns[] = {(size_t)M * D, (size_t)M * D, (size_t)M * D, (size_t)M * 4 * D, (size_t)M * 3 * D, (size_t)4 * D * D, (size_t)4 * D * D, (size_t)4 * D * D};
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, all[i], ns[i], 17u * i + 3);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, mask, (size_t)M, 99u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, big, nbig, 5u);
    long long* probe;
    CK(hipMalloc(&probe, 4096 * 8 * sizeof(long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(pn_probe), &probe, sizeof probe));
    CK(hipDeviceSynchronize());

    float *hpk, *ctxpk, *hnpk;
    CK(hipMalloc(&hpk, M * D * 4)); CK(hipMalloc(&ctxpk, M * D * 4)); CK(hipMalloc(&hnpk, M * D * 4));
    CK(launch_pack_rows(0, h, D, M, D, hpk)); CK(launch_pack_rows(0, ctx, D, M, D, ctxpk));
    const int apk = getenv("PN_UNPACKED_A") ? 0 : 1;
    PnArgs pa{};
    pa.M = M; pa.nseg = getenv("PN_ONE_SEG") ? 1 : 2;     // 1: the product's launch since round 3 (h.U rides in the attention launch)
    pn_seg_defaults(pa.seg[0]); pa.seg[0].npairs = 1; pa.seg[0].p[0] = PnPair{apk ? hpk : h, D, pWd, D, apk}; pa.seg[0].C = sproj; pa.seg[0].ldc = 4 * D; pa.seg[0].N = 4 * D;
    pn_seg_defaults(pa.seg[1]); pa.seg[1].npairs = 1; pa.seg[1].p[0] = PnPair{apk ? hpk : h, D, pU, D, apk}; pa.seg[1].C = preh; pa.seg[1].ldc = 4 * D; pa.seg[1].N = 4 * D;
    pa.seg[1].add = xproj; pa.seg[1].ldadd = 4 * D;
    LstmPnArgs la{};
    la.npairs = 1; la.p[0] = PnPair{apk ? ctxpk : ctx, D, pWc, D, apk}; la.h_pk = hnpk; la.pre_add = preh; la.ldpre = 4 * D; la.dp = dp; la.lddp = 3 * D; la.mask = mask;
    la.h_prev = h; la.c_prev = c; la.h_out = hn; la.c_out = cn; la.gates = gates; la.d1 = nullptr; la.d1_scalar = 0.5f; la.hd_out = hd; la.M = M; la.D = D;

    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    std::vector<long long> st(4096 * 8);
    printf("-- variant %d (0 product, 1 A as a coalesced stream, 2 no MFMAs, 3 no A loads, 4 no B loads)\n", PN_VARIANT);
    {
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)big, (float4*)big, nbig / 4);
        CK(hipEventRecord(e0, 0));
        CK(launch_panel(0, pa));
        CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)big, (float4*)big, nbig / 4);
        CK(hipDeviceSynchronize());
        float us; CK(hipEventElapsedTime(&us, e0, e1)); us *= 1000.f;
        CK(hipMemcpy(st.data(), probe, 4096 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
        if (it >= 3) report("hproj", st, 256, us);
        hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)big, (float4*)big, nbig / 4);
        CK(hipEventRecord(e2, 0));
        CK(launch_lstm_panel(0, la));
        CK(hipEventRecord(e3, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&us, e2, e3)); us *= 1000.f;
        CK(hipMemcpy(st.data(), probe, 4096 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
        if (it >= 3) report("lstm", st, 256, us);
    }
    }
    // back-to-back launches without the cache-evicting stream in between (weights may stay in L2 / Infinity Cache)
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 50; ++i) CK(launch_panel(0, pa));
    CK(hipEventRecord(e1, 0));
    for (int i = 0; i < 50; ++i) CK(launch_lstm_panel(0, la));
    CK(hipEventRecord(e2, 0));
    CK(hipDeviceSynchronize());
    float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
    printf("back-to-back: hproj %.2f us, lstm %.2f us per launch\n", a * 20.f, b * 20.f);
    return 0;
}
