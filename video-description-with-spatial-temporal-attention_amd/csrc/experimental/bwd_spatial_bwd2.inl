// EXPERIMENTAL -- not part of the product build.  Compiled into bwd.hip only under -DSTATTN_EXPERIMENTAL=1 (tools/build_variant.sh exp ...).
// Written in round 5 while the GPU pool was closed; has never run on a GPU.  A kernel moves from here into bwd.hip when it has passed the
// parity suite AND beaten the shipped kernel on the bench (VERDICT r05 item 2's bars), otherwise this file is deleted.
// Runtime selection inside an experimental build: STATTN_BWD2=1 (four workgroups per CU) or 2 (three).
// workgroup-uniform base pointer + per-lane BYTE offset (global_load ... v_off, s[base:base+1]: one VGPR per address instead of two)
__device__ __forceinline__ float4 ld4u(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st4u(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// ---- The same item for K <= 8 regions and D <= 1024 (configs[1]), written against what hipcc made of the kernel above (ISA of
// round 5): inside its 128-VGPR budget the compiler requests the rows of a phase two at a time with a full wait behind each
// pair -- the K <= KR loop alone is SEVEN dependent memory round trips (four LW rows + blt; the state projection; Ult; the L rows in
// four pairs), the dcsum partials four more, and each of the 144 wave-shuffle steps is an LDS round trip.  Here a lane owns one float4
// column of the item, and every phase requests its rows BEFORE the reduction / barrier of the phase in front of it:
//     top:                    the 8 LW rows, the temporal part's operands, the dcsum partials (grouped)
//     before reduction 1:     the 8 L rows, blt / Ult / Ug / Um, the frame rows PG / PM, the state projections
//     before reduction 2:     the 8 PL rows, Ul, the state projection of the local scorer
// Three exposed round trips per item instead of about fifteen.  Same arithmetic in the same order as spatial_bwd_kernel<8, false>
// (bit-equal results); dcsum of the lane's column stays in registers (no LDS copy).  STATTN_BWD2=1 selects it (off until measured).
template <int WPS>       // workgroups per CU the register budget is cut for: 4 (128 VGPRs, as spatial_bwd_kernel<8>) or 3 (168)
__global__ __launch_bounds__(256, WPS) void spatial_bwd2_kernel(const SpatialBwdArgs a) {
    constexpr int KR = 8;
    __shared__ float s_red[4 * KR];
    __shared__ float s_al[KMAX], s_da[KMAX];
    __shared__ float s_de[3];
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[4 * 64 * 16];
        rider_tile<4>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    // (integer division runs on the VALU: without readfirstlane the uniform item index -- and every row pointer formed from it -- lives in VGPRs)
    const int bt = __builtin_amdgcn_readfirstlane(xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T));
    const int b = __builtin_amdgcn_readfirstlane(bt / T), tid = threadIdx.x;
    const int nd4 = D >> 2, d4 = min(tid, nd4 - 1);
    const bool act = tid < nd4;                         // (D < 1024: the lanes past D / 4 load a clamped column and contribute nothing)
    // every address = a workgroup-uniform row pointer (SGPR pair) + ONE 32-bit lane offset: 64-bit per-lane addresses cost two VGPRs
    // per load in flight, and this kernel keeps up to 28 loads in flight inside 128 VGPRs
    const unsigned lob = 16u * (unsigned)d4;              // BYTE offset of the lane (a zero-extended 32-bit byte offset is what the scalar-base addressing mode takes)
    const size_t slab = (size_t)bt * K * D;
    const float* __restrict__ sp = a.sproj + (size_t)b * a.ldsp;
    if (tid < K) s_al[tid] = a.alphal[(size_t)bt * K + tid];
    unsigned ro[KR];                                    // byte offset of (region kk, this lane's column) inside the item's slab: the same for PL, L and LW
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) ro[kk] = lob + (unsigned)min(kk, K - 1) * (unsigned)D * 4u;
    float4 lw[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) lw[kk] = ld4u(a.LW + slab, ro[kk]);
    // ---- temporal part
    const int tf = bt - b * T;
    const size_t MD = (size_t)a.M * D;
    const float sel = a.has_sel ? a.sel[b] : 1.f;
    const size_t ob = (size_t)b * D, o = (size_t)bt * D;        // (uniform; the lane offset is added at the use)
    const float4 blt4 = ld4u(a.blt, lob), s3 = ld4u(sp + 3 * D, lob), ult = ld4u(a.Ult, lob);
    float4 dcs;
    float q[8];
    {
        const float4 xc = ld4u(a.csum + ob, lob), xg = ld4u(a.G + o, lob), xm = ld4u(a.Mo + o, lob), xl = ld4u(a.CL + o, lob);
        const float4 x0 = ld4u(a.cparts + ob, lob), x1 = ld4u(a.cparts + MD + ob, lob), x2 = ld4u(a.cparts + 2 * MD + ob, lob);
        // dcsum[b, column] = sel * (readout term + partials of dpre.Wc^T): the first four partials in flight together
        const int nP = a.nP;
        float4 pp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = ld4u(a.dctxP + (size_t)min(i, nP - 1) * MD + ob, lob);
        float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.dctx_r) dc = ld4u(a.dctx_r + ob, lob);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < nP) add4(dc, pp[i]);
        for (int i = 4; i < nP; ++i) add4(dc, ld4u(a.dctxP + (size_t)i * MD + ob, lob));
        dcs = scale4(dc, sel);
        if (tf == 0 && act) st4u(a.dcsum + ob, lob, dcs);
        q[0] = dot4(dcs, xg); q[1] = dot4(dcs, xm); q[2] = dot4(dcs, xl);
        q[3] = dot4(dcs, x0); q[4] = dot4(dcs, x1); q[5] = dot4(dcs, x2);
        q[6] = dot4(dc, xc); q[7] = 0.f;
        if (!act) {
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = 0.f;
        }
        // (everything of this phase is FINISHED here: left to itself, LLVM sinks each dot product down to the wave reduction that consumes
        //  it -- behind the requests of the next phase, whose rows then arrive next to operands that are still alive: spills)
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(q[i]));
    }
    // plt recomputed (it needs the forward weights and the LW rows only): dplt = dplb * delt once the reduction has produced delt
    __syncthreads();                                    // s_al
    float4 dplb;
    {
        float4 pl = blt4;
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) if (kk < K) fma4(pl, s_al[kk], lw[kk]);
        dplb = mul4(ult, one_minus_sq(tanh4s(pl, s3)));
        asm volatile("" : "+v"(dplb.x), "+v"(dplb.y), "+v"(dplb.z), "+v"(dplb.w));
    }
    // what the spatial part reads, requested before the reduction of the temporal part (and not earlier: with the temporal operands
    // still live the 16 slab rows do not fit the 128-VGPR budget)
    __builtin_amdgcn_sched_barrier(0);
    float4 lr[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) lr[kk] = ld4u(a.L + slab, ro[kk]);
    __builtin_amdgcn_sched_barrier(0);
    block_sum<8>(q, s_red, tid, 4);
    if (tid < 192) {
        const int w = tid >> 6, lane = tid & 63;
        const float* al = (w == 0 ? a.ag : (w == 1 ? a.am : a.alt)) + (size_t)b * T;
        const float* r = w == 0 ? a.rg : (w == 1 ? a.rm : a.rlt);
        float dotr = 0.f;
        if (r) {
            for (int t = lane; t < T; t += 64) dotr += al[t] * r[(size_t)b * T + t];
            dotr = wave_sum(dotr);
        }
        if (lane == 0) {
            const float da = q[w] + (r ? r[bt] : 0.f);
            const float de = al[tf] * (da - (q[3 + w] + dotr));
            s_de[w] = de;
            (w == 0 ? a.deg : (w == 1 ? a.dem : a.delt))[bt] = de;
        }
    } else if (tid == 192 && tf == 0) {
        a.dselpre[b] = a.has_sel ? q[6] * sel * (1.f - sel) : 0.f;
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);      // (nothing that consumes the rows in flight may be scheduled above the reduction: it would wait for them there)
    const float alt = a.alt[bt], delt = s_de[2], deg = s_de[0], dem = s_de[1];
    // dplt, d alpha_k
    float p[KR];
    float4 dpl;
    {
        dpl = scale4(dplb, delt);
        if (act) st4u(a.dplt + o, lob, dpl);
        const float4 dcl = scale4(dcs, alt);
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) p[kk] = act ? dot4(dcl, lr[kk]) + dot4(dpl, lw[kk]) : 0.f;
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) asm volatile("" : "+v"(p[kk]));
    }
    // the rows of the last pass, requested before the reduction of this one
    __builtin_amdgcn_sched_barrier(0);
    float4 plr[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) plr[kk] = ld4u(a.PL + slab, ro[kk]);
    const float4 sl = ld4u(sp, lob), ul = ld4u(a.Ul, lob);
    // ... and the frame scorers' rows (their part, dsg / dsm, comes last: it needs nothing but de of the temporal part)
    const float4 pg = ld4u(a.PG + o, lob), pm = ld4u(a.PM + o, lob), s1 = ld4u(sp + D, lob), s2 = ld4u(sp + 2 * D, lob);
    const float4 ug = ld4u(a.Ug, lob), um = ld4u(a.Um, lob);
    __builtin_amdgcn_sched_barrier(0);
    block_sum<KR>(p, s_red, tid, 4);
    if (tid < KR && tid < K) s_da[tid] = p[tid] + (a.rl ? a.rl[(size_t)bt * K + tid] : 0.f);
    __syncthreads();
    // softmax backward over the K regions
    float dotp = 0.f;
    for (int k = 0; k < K; ++k) dotp += s_al[k] * s_da[k];
    __syncthreads();
    if (tid < K) {
        const float de = s_al[tid] * (s_da[tid] - dotp);
        s_da[tid] = de;
        a.del[(size_t)bt * K + tid] = de;
    }
    __syncthreads();
    // dsl (this frame) = sum_k del_k Ul (1 - tanh^2(PL_k + sl))
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) if (kk < K) fma4(acc, s_da[kk], one_minus_sq(tanh4s(plr[kk], sl)));
        if (act) st4u(a.dslp + o, lob, mul4(acc, ul));
    }
    // per-frame dsg / dsm
    if (act) {
        st4u(a.dsgp + o, lob, scale4(mul4(ug, one_minus_sq(tanh4s(pg, s1))), deg));
        st4u(a.dsmp + o, lob, scale4(mul4(um, one_minus_sq(tanh4s(pm, s2))), dem));
    }
}

// launch hook of launch_spatial_bwd (fp32 slabs, K <= 8); returns true when it launched
static bool exp_launch_spatial_bwd2(hipStream_t s, const SpatialBwdArgs& a, dim3 grid) {
    static const char* bwd2 = getenv("STATTN_BWD2");
    if (!(bwd2 && a.K <= 8 && a.D <= 1024)) return false;
    if (bwd2[0] == '1') hipLaunchKernelGGL(spatial_bwd2_kernel<4>, grid, dim3(256), 0, s, a);
    else if (bwd2[0] == '2') hipLaunchKernelGGL(spatial_bwd2_kernel<3>, grid, dim3(256), 0, s, a);
    else return false;
    return true;
}
