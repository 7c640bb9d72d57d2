// EXPERIMENTAL -- not part of the product build.  Compiled into attn.hip only under -DSTATTN_EXPERIMENTAL=1 (tools/build_variant.sh exp ...).
// Written in round 5 while the GPU pool was closed; has never run on a GPU.  A kernel moves from here into attn.hip when it has passed the
// parity suite AND beaten the shipped kernel on the bench (VERDICT r05 item 2's bars), otherwise this file is deleted.
// Runtime selection inside an experimental build: STATTN_BF16_V2=1.
// Wave sums of N per-thread values -> s_red, and the workgroup total of value `tid` in thread tid < N (the only threads that use
// one: a register array indexed by tid is a scratch array).  Same summation order as block_sum_w.  The caller puts a barrier
// between the last read of a total and the next use of s_red.
template <int N, int NW>
__device__ __forceinline__ float block_sum_keep(const float (&v)[N], float* s_red /*[NW][N]*/, int tid) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float r = wave_sum(v[i]);
        if (lane == 0) s_red[w * N + i] = r;
    }
    __syncthreads();
    float t = 0.f;
    if (tid < N) {
#pragma unroll
        for (int q = 0; q < NW; ++q) t += s_red[q * N + tid];
    }
    return t;
}

// ---- the same item for K <= 16 regions and D <= 1024 (BASELINE configs[3]), written against what hipcc made of spatial_bf16_kernel
// (ISA, round 5): its score loop requests five of a group's eight rows, then the other three ONE AT A TIME behind a full wait each, the
// frame scorers' operands behind the region scores and the weighted sums four regions at a time -- about eighteen dependent memory round
// trips per item at K = 16, which is what its 68 us (4.1 TB/s) at configs[3] are made of.  Here a lane owns one 8-column group and
// every phase's rows are requested together and BEFORE the reduction / softmax in front of the phase:
//     top:                        the first eight PL rows + the region scorer's operands; behind them the frame scorers' operands
//     before the first reduction:  the second eight PL rows (K > 8)
//     before the last reduction:   the first eight L and LW rows
//     after the softmax:           the second eight L and LW rows (K > 8), before the first eight are consumed
// Four exposed round trips (two for K <= 8).  K > 8: two waves per SIMD (the two times sixteen packed rows of the weighted sums are 128
// VGPRs; inside 168 hipcc spills them one row at a time with a full wait each), four workgroups per CU with up to 32 KB in flight each.  Same arithmetic in the same order as spatial_bf16_kernel (the weights of the regions
// past K are zero instead of skipped).  STATTN_BF16_V2=1 selects it: written while the GPU pool was closed, unmeasured.
template <int NT, bool TWO>         // TWO: 8 < K <= 16
__global__ __launch_bounds__(NT, TWO ? 2 : 3) void spatial_bf16v2_kernel(const SpatialArgs a) {
    constexpr int NW = NT / 64;
    __shared__ float s_red[NW * 10];
    __shared__ float s_e[16];
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[NW * 64 * 16];
        rider_tile<NW>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt[0];
    // (integer division runs on the VALU: without readfirstlane the uniform item index -- and every row pointer formed from it -- lives in VGPRs)
    const int bt = __builtin_amdgcn_readfirstlane(xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T));
    const int b = __builtin_amdgcn_readfirstlane(bt / T), t = bt - b * T;
    const int v = a.vid ? a.vid[b] : b;
    const int tid = threadIdx.x;
    const int nd8 = D >> 3, d8 = min(tid, nd8 - 1);
    const bool on = tid < nd8;                         // (D < 8 NT: the lanes past D / 8 load a clamped column group and contribute nothing)
    const float onf = on ? 1.f : 0.f;                  // (a factor, not a branch: `on ? f(x) : 0` became a branch with its own waits per region)
    // every address = a workgroup-uniform row pointer (SGPR pair) + ONE 32-bit lane byte offset: a 64-bit address per lane costs two VGPRs
    // per load in flight, and up to 32 loads are in flight here
    const size_t slab = ((size_t)v * T + t) * K * D;
    const uint16_t* __restrict__ PL = reinterpret_cast<const uint16_t*>(a.PL) + slab;
    const uint16_t* __restrict__ L = reinterpret_cast<const uint16_t*>(a.L) + slab;
    const uint16_t* __restrict__ LW = reinterpret_cast<const uint16_t*>(a.LW) + slab;
    const unsigned lo2 = 16u * (unsigned)d8, lo4 = 32u * (unsigned)d8;     // byte offset of the lane's 8 columns in a bf16 / an fp32 row
    const float* __restrict__ sl = a.sproj + (size_t)b * a.ldsp;
    const size_t fo = ((size_t)v * T + t) * D;
    auto ldr = [&](const uint16_t* base, int row) {      // the lane's 8 bf16 of row `row` (uniform) of a slab
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + (size_t)row * D) + lo2);
    };
    auto ldf = [&](const float* base, int half) {        // the lane's columns 4 half .. 4 half + 3 of an fp32 row
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + lo4 + 16u * half);
    };

    uint4 x[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) x[kk] = ldr(PL, min(kk, K - 1));
    const float4 s0 = ldf(sl, 0), s1 = ldf(sl, 1);
    const float4 u0 = ldf(a.Ul, 0), u1 = ldf(a.Ul, 1);
    __builtin_amdgcn_sched_barrier(0);
    // frame scorers (PG / PM stay fp32: they are K times smaller): in flight under the region scores
    const float4 pg0 = ldf(a.PG + fo, 0), pg1 = ldf(a.PG + fo, 1), sg0 = ldf(sl + D, 0), sg1 = ldf(sl + D, 1);
    const float4 ug0 = ldf(a.Ug, 0), ug1 = ldf(a.Ug, 1);
    const float4 pm0 = ldf(a.PM + fo, 0), pm1 = ldf(a.PM + fo, 1), sm0 = ldf(sl + 2 * D, 0), sm1 = ldf(sl + 2 * D, 1);
    const float4 um0 = ldf(a.Um, 0), um1 = ldf(a.Um, 1);
    __builtin_amdgcn_sched_barrier(0);
    auto score = [&](const uint4 r) {
        float f[8];
        bf8_to_f32(r, f);
        return onf * (dot4_tanh(make_float4(f[0], f[1], f[2], f[3]), s0, u0) + dot4_tanh(make_float4(f[4], f[5], f[6], f[7]), s1, u1));
    };
    float p[10];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) p[kk] = score(x[kk]);
    p[8] = onf * (dot4_tanh(pg0, sg0, ug0) + dot4_tanh(pg1, sg1, ug1));
    p[9] = onf * (dot4_tanh(pm0, sm0, um0) + dot4_tanh(pm1, sm1, um1));
    __builtin_amdgcn_sched_barrier(0);
    // in flight under the first reduction: the second eight PL rows (TWO), or already the rows of the weighted sums
    uint4 y[8], l0[8], lw[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        if (TWO) y[kk] = ldr(PL, min(8 + kk, K - 1));
        else { l0[kk] = ldr(L, min(kk, K - 1)); lw[kk] = ldr(LW, min(kk, K - 1)); }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const float tot = block_sum_keep<10, NW>(p, s_red, tid);
        if (tid < 8 && tid < K) s_e[tid] = tot + cl0;
        if (tid == 8) a.eg[bt] = tot + cg0;
        if (tid == 9) a.em[bt] = tot + cm0;
    }
    if (TWO) {
        float p2[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) p2[kk] = score(y[kk]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { l0[kk] = ldr(L, kk); lw[kk] = ldr(LW, kk); }      // (K > 8: rows 0 .. 7 exist)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                               // every total of the first reduction has been read
        const float tot = block_sum_keep<8, NW>(p2, s_red, tid);
        if (tid < 8 && 8 + tid < K) s_e[8 + tid] = tot + cl0;
    }
    __syncthreads();

    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s_e[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(s_e[k] - mx);
    const float inv = 1.0f / sum;
    __syncthreads();
    if (tid < 16) {
        const float al = tid < K ? __expf(s_e[tid] - mx) * inv : 0.f;      // (zero weights for the clamped rows past K)
        if (tid < K) a.alphal[(size_t)bt * K + tid] = al;
        s_e[tid] = al;
    }
    __syncthreads();

    float c[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { c[q] = 0.f; w[q] = 0.f; }
    auto wsum = [&](const uint4 lr, const uint4 qr, const float al) {
        float f[8], q8[8];
        bf8_to_f32(lr, f);
        bf8_to_f32(qr, q8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { c[q] += al * f[q]; w[q] += al * q8[q]; }
    };
    if (TWO) {      // the second eight rows of both tensors, requested before the first eight are consumed
        uint4 l1[8], q1[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { l1[kk] = ldr(L, min(8 + kk, K - 1)); q1[kk] = ldr(LW, min(8 + kk, K - 1)); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wsum(l0[kk], lw[kk], s_e[kk]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wsum(l1[kk], q1[kk], s_e[8 + kk]);
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wsum(l0[kk], lw[kk], s_e[kk]);
    }
    // the local-temporal scorer's operands (all lanes: clamped column group, weight onf)
    const float4 b0 = ldf(a.blt, 0), b1 = ldf(a.blt, 1);
    if (on) {
        float* cl = reinterpret_cast<float*>(reinterpret_cast<char*>(a.CL + (size_t)bt * D) + lo4);
        st4(cl, make_float4(c[0], c[1], c[2], c[3]));
        st4(cl + 4, make_float4(c[4], c[5], c[6], c[7]));
    }
    float pe[1];
    pe[0] = onf * (dot4_tanh(make_float4(w[0] + b0.x, w[1] + b0.y, w[2] + b0.z, w[3] + b0.w), ldf(sl + 3 * D, 0), ldf(a.Ult, 0)) +
                   dot4_tanh(make_float4(w[4] + b1.x, w[5] + b1.y, w[6] + b1.z, w[7] + b1.w), ldf(sl + 3 * D, 1), ldf(a.Ult, 1)));
    const float tot = block_sum_keep<1, NW>(pe, s_red, tid);
    if (tid == 0) a.elt[bt] = tot + clt0;
}

// launch hook of launch_spatial (bf16 slabs); returns true when it launched
static bool exp_launch_spatial_bf16v2(hipStream_t s, const SpatialArgs& a) {
    static const char* v2 = getenv("STATTN_BF16_V2");
    if (!(v2 && v2[0] == '1' && a.D <= 1024 && a.K <= 16 && a.clt)) return false;
    if (a.K > 8) hipLaunchKernelGGL((spatial_bf16v2_kernel<128, true>), dim3(a.M * a.T + a.rider.nblocks), dim3(128), 0, s, a);
    else hipLaunchKernelGGL((spatial_bf16v2_kernel<128, false>), dim3(a.M * a.T + a.rider.nblocks), dim3(128), 0, s, a);
    return true;
}
