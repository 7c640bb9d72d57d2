// EXPERIMENTAL -- not part of the product build.  Compiled into bwd.hip only under -DSTATTN_EXPERIMENTAL=1 (tools/build_variant.sh exp ...).
// Written in round 5 while the GPU pool was closed; has never run on a GPU.  A kernel moves from here into bwd.hip when it has passed the
// parity suite AND beaten the shipped kernel on the bench (VERDICT r05 item 2's bars), otherwise this file is deleted.
// Runtime selection inside an experimental build: STATTN_BF16_V2=1.
// ---- The same item for K <= 16 and D <= 1024 (BASELINE configs[3]), written against what hipcc made of the kernel above (ISA, round 5):
// at three waves per SIMD its sixteen packed LW rows and everything else do not fit 168 VGPRs -- 43 dwords per lane go to scratch, each
// spill and reload behind a full s_waitcnt vmcnt(0), the `if (kk < K)` of the plt sum is a branch with its own wait per region and the L
// rows arrive four at a time: well over twenty dependent round trips per item (100 us, 3.6 TB/s of its bytes at configs[3]).  Here, at two
// waves per SIMD (no spill), every phase requests its rows before the reduction in front of it, as spatial_bwd2_kernel does for fp32:
//     top:                       the 16 LW rows, the temporal part's operands, the dcsum partials
//     before reduction 1:        the 16 L rows, blt / Ult / Ug / Um, the frame rows PG / PM, the state projections
//     before reduction 2:        the 16 PL rows, Ul, the state projection of the local scorer
// Same arithmetic in the same order (the weights of the regions past K are zero instead of skipped).  STATTN_BF16_V2=1 selects it:
// written while the GPU pool was closed, unmeasured.
__global__ __launch_bounds__(128, 2) void spatial_bwd_bf16v2_kernel(const SpatialBwdArgs a) {
    constexpr int NW = 2, KR = 16;
    __shared__ float s_red[NW * KR];
    __shared__ float s_al[KR], s_da[KR];
    __shared__ float s_de[3];
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[NW * 64 * 16];
        rider_tile<NW>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    const int bt = __builtin_amdgcn_readfirstlane(xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T));
    const int b = __builtin_amdgcn_readfirstlane(bt / T), tid = threadIdx.x;
    const int nd8 = D >> 3, d8 = min(tid, nd8 - 1);
    const bool on = tid < nd8;                          // (D < 1024: the lanes past D / 8 load a clamped column group and contribute nothing)
    const float onf = on ? 1.f : 0.f;
    const unsigned lo2 = 16u * (unsigned)d8, lo4 = 32u * (unsigned)d8;      // byte offset of the lane's 8 columns in a bf16 / an fp32 row
    const size_t slab = (size_t)bt * K * D;
    const uint16_t* __restrict__ PLs = reinterpret_cast<const uint16_t*>(a.PL) + slab;
    const uint16_t* __restrict__ Ls = reinterpret_cast<const uint16_t*>(a.L) + slab;
    const uint16_t* __restrict__ LWs = reinterpret_cast<const uint16_t*>(a.LW) + slab;
    const float* __restrict__ sp = a.sproj + (size_t)b * a.ldsp;
    auto ldr = [&](const uint16_t* base, int row) {       // the lane's 8 bf16 of row `row` (uniform) of a slab
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + (size_t)row * D) + lo2);
    };
    auto ldf = [&](const float* base) {                   // the lane's 8 columns of an fp32 row (uniform pointer)
        const char* q = reinterpret_cast<const char*>(base) + lo4;
        const float4 x = *reinterpret_cast<const float4*>(q), y = *reinterpret_cast<const float4*>(q + 16);
        return F8{{x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w}};
    };
    auto stf = [&](float* base, const F8& x) {
        char* q = reinterpret_cast<char*>(base) + lo4;
        *reinterpret_cast<float4*>(q) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
        *reinterpret_cast<float4*>(q + 16) = make_float4(x.v[4], x.v[5], x.v[6], x.v[7]);
    };
    if (tid < KR) s_al[tid] = tid < K ? a.alphal[(size_t)bt * K + tid] : 0.f;      // (zero weights for the clamped rows past K)
    uint4 lw[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) lw[kk] = ldr(LWs, min(kk, K - 1));
    const F8 blt = ldf(a.blt), slt = ldf(sp + 3 * D), ult = ldf(a.Ult);
    // ---- temporal part (see spatial_bwd_kernel)
    const int tf = bt - b * T;
    const size_t MD = (size_t)a.M * D, ob = (size_t)b * D, o = (size_t)bt * D;
    const float sel = a.has_sel ? a.sel[b] : 1.f;
    F8 dcs;
    float q[8];
    {
        const F8 xc = ldf(a.csum + ob), xg = ldf(a.G + o), xm = ldf(a.Mo + o), xl = ldf(a.CL + o);
        const F8 x0 = ldf(a.cparts + ob), x1 = ldf(a.cparts + MD + ob), x2 = ldf(a.cparts + 2 * MD + ob);
        // dctx[b, columns] = readout term + partials of dpre.Wc^T: the first four partials in flight together
        const int nP = a.nP;
        F8 pp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = ldf(a.dctxP + (size_t)min(i, nP - 1) * MD + ob);
        F8 dc{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        if (a.dctx_r) dc = ldf(a.dctx_r + ob);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dc.v[j] += pp[i].v[j];
            }
        for (int i = 4; i < nP; ++i) {
            const F8 pq = ldf(a.dctxP + (size_t)i * MD + ob);
#pragma unroll
            for (int j = 0; j < 8; ++j) dc.v[j] += pq.v[j];
        }
        q[6] = onf * dot8(dc, xc);
#pragma unroll
        for (int j = 0; j < 8; ++j) dcs.v[j] = dc.v[j] * sel;
        if (tf == 0 && on) stf(a.dcsum + ob, dcs);
        q[0] = onf * dot8(dcs, xg); q[1] = onf * dot8(dcs, xm); q[2] = onf * dot8(dcs, xl);
        q[3] = onf * dot8(dcs, x0); q[4] = onf * dot8(dcs, x1); q[5] = onf * dot8(dcs, x2);
        q[7] = 0.f;
        // (the dot products are FINISHED here: left to itself, LLVM sinks each one down to the wave reduction that consumes it -- behind the
        //  requests of the next phase, whose rows then arrive next to the operands still alive, and the kernel spills)
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(q[i]));
    }
    // plt recomputed (it needs the forward weights and the LW rows only): dplt = dplb * delt once the reduction has produced delt
    __syncthreads();                                    // s_al
    F8 dplb;
    {
        F8 pl = blt;
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) {
            const F8 x = widen8(lw[kk]);
            const float al = s_al[kk];
#pragma unroll
            for (int i = 0; i < 8; ++i) pl.v[i] += al * x.v[i];
            // (four rows at a time: left alone, the scheduler widens all sixteen packed rows up front -- 128 VGPRs for 64)
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float th = fast_tanh(pl.v[i] + slt.v[i]); dplb.v[i] = ult.v[i] * (1.f - th * th); }
    }
    // what the next phase reads, requested before the reduction of this one
    __builtin_amdgcn_sched_barrier(0);
    uint4 lr[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) lr[kk] = ldr(Ls, min(kk, K - 1));
    __builtin_amdgcn_sched_barrier(0);
    block_sum<8>(q, s_red, tid, NW);
    {
        const int lane = tid & 63;
        for (int w = tid >> 6; w < 3; w += NW) {       // softmax backward of the three temporal attentions, one wave each
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
        }
        if (tid == 0 && tf == 0) a.dselpre[b] = a.has_sel ? q[6] * sel * (1.f - sel) : 0.f;
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);      // (nothing that consumes the rows in flight may be scheduled above the reduction: it would wait for them there)
    const float alt = a.alt[bt], delt = s_de[2], deg = s_de[0], dem = s_de[1];
    // plt recomputed, dplt, d alpha_k = <alt dcsum, L_k> + <dplt, LW_k> + r_k
    float p[KR];
    {
        F8 dpl;
#pragma unroll
        for (int i = 0; i < 8; ++i) dpl.v[i] = dplb.v[i] * delt;
        if (on) stf(a.dplt + o, dpl);
        F8 dcl;
#pragma unroll
        for (int i = 0; i < 8; ++i) dcl.v[i] = dcs.v[i] * alt;
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) {
            // The LW rows stay PACKED between their two uses: seen as the same values, hipcc keeps the sixteen rows it widened for plt -- 128
            // VGPRs for 64 -- alive across the reduction (common subexpressions), and the kernel spills
            uint4 t = lw[kk];
            asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
            p[kk] = onf * (dot8(dcl, widen8(lr[kk])) + dot8(dpl, widen8(t)));
            asm volatile("" : "+v"(p[kk]));             // (finished here, not sunk into the reduction: see q above)
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the rows of the last pass, requested before the reduction of this one
    __builtin_amdgcn_sched_barrier(0);
    uint4 plr[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) plr[kk] = ldr(PLs, min(kk, K - 1));
    const F8 sl = ldf(sp), ul = ldf(a.Ul);
    // ... and the frame scorers' rows (their part, dsg / dsm, comes last: next to the 32 packed rows of the pass above they do not fit)
    const F8 pg = ldf(a.PG + o), pm = ldf(a.PM + o), sg = ldf(sp + D), sm = ldf(sp + 2 * D), ug = ldf(a.Ug), um = ldf(a.Um);
    __builtin_amdgcn_sched_barrier(0);
    block_sum<KR>(p, s_red, tid, NW);
    {   // thread k publishes total k: a select chain, not p[tid] (a register array indexed by tid is a scratch array)
        float mine = 0.f;
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) mine = tid == kk ? p[kk] : mine;
        if (tid < K) s_da[tid] = mine + (a.rl ? a.rl[(size_t)bt * K + tid] : 0.f);
    }
    __syncthreads();
    float dotp = 0.f;
    for (int k = 0; k < K; ++k) dotp += s_al[k] * s_da[k];
    __syncthreads();
    if (tid < KR) {
        const float de = tid < K ? s_al[tid] * (s_da[tid] - dotp) : 0.f;        // (zero for the clamped rows past K)
        s_da[tid] = de;
        if (tid < K) a.del[(size_t)bt * K + tid] = de;
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    {   // dsl (this frame) = Ul sum_k del_k (1 - tanh^2(PL_k + sl))
        F8 acc{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) {
            const F8 x = widen8(plr[kk]);
            const float de = s_da[kk];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float th = fast_tanh(x.v[i] + sl.v[i]); acc.v[i] += de * (1.f - th * th); }
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] *= ul.v[i];
        if (on) stf(a.dslp + o, acc);
    }
    {   // per-frame dsg / dsm: de Ug (1 - tanh^2(PG_t + sg))   (:389-397, :402-410)
        F8 og, om;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tg = fast_tanh(pg.v[i] + sg.v[i]), tm = fast_tanh(pm.v[i] + sm.v[i]);
            og.v[i] = ug.v[i] * (1.f - tg * tg) * deg; om.v[i] = um.v[i] * (1.f - tm * tm) * dem;
        }
        if (on) { stf(a.dsgp + o, og); stf(a.dsmp + o, om); }
    }
}

// launch hook of launch_spatial_bwd (bf16 slabs, K <= 16); returns true when it launched
static bool exp_launch_spatial_bwd_bf16v2(hipStream_t s, const SpatialBwdArgs& a, dim3 grid) {
    static const char* v2 = getenv("STATTN_BF16_V2");
    if (!(v2 && v2[0] == '1' && a.K <= 16 && a.D % 8 == 0 && a.D <= 1024)) return false;
    hipLaunchKernelGGL(spatial_bwd_bf16v2_kernel, grid, dim3(128), 0, s, a);
    return true;
}
