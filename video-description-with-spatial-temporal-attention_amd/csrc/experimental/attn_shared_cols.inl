// EXPERIMENTAL -- not part of the product build.  Compiled into attn.hip only under -DSTATTN_EXPERIMENTAL=1 (tools/build_variant.sh exp ...).
// Written in round 5 while the GPU pool was closed; has never run on a GPU.  A kernel moves from here into attn.hip when it has passed the
// parity suite AND beaten the shipped kernel on the bench (VERDICT r05 item 2's bars), otherwise this file is deleted.
// Runtime selection inside an experimental build: STATTN_SHARED_COLS=1.
// ---- the same for K <= 8 regions and D <= 1024: the reference's own evaluation shape (config.py: 8 regions, hidden 1024, beam 5).
// There spatial_shared_body leaves three of its four waves without a region to score (a wave owns EIGHT regions) and walks a slab in
// four dependent bursts of 8 KB: 92 us per word for 137 MB at 51 videos x 28 frames (0.25 of HBM).  Here a lane owns ONE float4 column of
// the item and requests its 8 rows of PL, L and LW, the two frame rows and the H hypotheses' state projections before the first
// wait -- one memory round trip per item, 104 KB in flight per workgroup, all four waves on the H x K x 4 reciprocals -- and the
// H (K + 2) + 1 partial sums cross the workgroup in one LDS hop.  Two workgroups per CU (the 24 slab rows alone are 96 VGPRs).
#ifndef STATTN_COLS_ABL
#define STATTN_COLS_ABL 0     // probe builds (tools/probes/cols_probe.sh): 1 scores without transcendentals, 2 no slab loads, 3 no CL stores
#endif
template <int H, bool HAS_LW>
__device__ __forceinline__ void spatial_shared_cols_body(const SpatialArgs& a, const int vt) {
    constexpr int NV = H * 8 + 2 * H + 1;            // region scores, frame scores, sum of Ul
    __shared__ float s_red[4 * NV];
    __shared__ float s_al[H * 8];
    const int T = a.T, K = a.K, D = a.D;
    const float cl0 = a.cl[0], cg0 = a.cg[0], cm0 = a.cm[0], clt0 = a.clt ? a.clt[0] : 0.f;
    const int v = vt / T, t = vt % T;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nd4 = D >> 2, d4 = min(tid, nd4 - 1);
    const float on = tid < nd4 ? 1.f : 0.f;           // lanes past D / 4 load a clamped column and contribute zeros
    const size_t slab = ((size_t)v * T + t) * K * D + 4 * d4, fo = ((size_t)v * T + t) * D + 4 * d4;
    const int b0 = v * H;                             // first row (hypothesis) of this video

    float4 x[8], l[8], q[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const size_t o = slab + (size_t)min(kk, K - 1) * D;
#if STATTN_COLS_ABL == 2
        x[kk] = make_float4(1e-3f * tid, 1e-3f * kk, 0.1f, 0.2f); l[kk] = x[kk]; q[kk] = x[kk];
#else
        x[kk] = ld4_nt(a.PL + o);
        l[kk] = ld4_nt(a.L + o);
        q[kk] = HAS_LW ? ld4_nt(a.LW + o) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    }
    const float4 pg = ld4(a.PG + fo), pm = ld4(a.PM + fo);
    float4 s0[H], s1[H], s2[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float* sp = a.sproj + (size_t)(b0 + h) * a.ldsp + 4 * d4;
        s0[h] = ld4(sp); s1[h] = ld4(sp + D); s2[h] = ld4(sp + 2 * D);
    }
    const float4 u4 = ld4(a.Ul + 4 * d4), ug = ld4(a.Ug + 4 * d4), um = ld4(a.Um + 4 * d4);

    // ---- scores (tanh split along its sum as in spatial_shared_body: one v_exp per slab element and per state projection element,
    // one v_rcp per (hypothesis, region, column))
    float r[NV];
    {
        const float4 m2u = make_float4(-2.f * on * u4.x, -2.f * on * u4.y, -2.f * on * u4.z, -2.f * on * u4.w);
#if STATTN_COLS_ABL != 1
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) x[kk] = exp2x4(x[kk]);
#endif
#pragma unroll
        for (int h = 0; h < H; ++h) {
#if STATTN_COLS_ABL == 1
            const float4 es = s0[h];
#else
            const float4 es = exp2x4(s0[h]);
#endif
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
#if STATTN_COLS_ABL == 1
                const float4 rc = make_float4(x[kk].x + es.x, x[kk].y + es.y, x[kk].z + es.z, x[kk].w + es.w);
#else
                const float4 rc = rcp1p4(x[kk], es);
#endif
                r[h * 8 + kk] = (m2u.x * rc.x + m2u.y * rc.y) + (m2u.z * rc.z + m2u.w * rc.w);
            }
            r[H * 8 + 2 * h] = on * dot4_tanh(pg, s1[h], ug);
            r[H * 8 + 2 * h + 1] = on * dot4_tanh(pm, s2[h], um);
        }
        r[NV - 1] = on * ((u4.x + u4.y) + (u4.z + u4.w));
    }
    // what the second half needs beside the slabs: requested here, they land under the reduction
    float4 s3[H];
    float4 bl = make_float4(0.f, 0.f, 0.f, 0.f), ult = bl;
    if (HAS_LW) {
#pragma unroll
        for (int h = 0; h < H; ++h) s3[h] = ld4(a.sproj + (size_t)(b0 + h) * a.ldsp + 3 * D + 4 * d4);
        bl = ld4(a.blt + 4 * d4); ult = ld4(a.Ult + 4 * d4);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float rr = wave_sum(r[i]);
        if (lane == 0) s_red[w * NV + i] = rr;
    }
    __syncthreads();
    if (w == 0) {           // lane 8 h + k: softmax of hypothesis h over its 8-lane group
        const int i = min(lane, H * 8 - 1), h = i >> 3, k = i & 7;
        const float usum = s_red[NV - 1] + s_red[2 * NV - 1] + s_red[3 * NV - 1] + s_red[4 * NV - 1];
        const float e = k < K ? s_red[i] + s_red[NV + i] + s_red[2 * NV + i] + s_red[3 * NV + i] + usum + cl0 : -INFINITY;
        float mx = e;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float ex = k < K ? __expf(e - mx) : 0.f;
        float sum = ex;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float al = ex / sum;
        if (lane < H * 8) {
            s_al[lane] = al;
            if (k < K) a.alphal[((size_t)(b0 + h) * T + t) * K + k] = al;
        }
    } else if (w == 1 && lane < 2 * H) {
        const int i = H * 8 + lane, h = lane >> 1;
        const float e = s_red[i] + s_red[NV + i] + s_red[2 * NV + i] + s_red[3 * NV + i];
        if (lane & 1) a.em[(size_t)(b0 + h) * T + t] = e + cm0;
        else a.eg[(size_t)(b0 + h) * T + t] = e + cg0;
    }
    __syncthreads();

    // ---- attended local feature (and the local-temporal score) per hypothesis, from the rows already in registers
    float pe[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = c4;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float al = s_al[h * 8 + kk];          // 0 for the regions past K
            c4.x += al * l[kk].x; c4.y += al * l[kk].y; c4.z += al * l[kk].z; c4.w += al * l[kk].w;
            if (HAS_LW) { w4.x += al * q[kk].x; w4.y += al * q[kk].y; w4.z += al * q[kk].z; w4.w += al * q[kk].w; }
        }
#if STATTN_COLS_ABL == 3
        w4.x += c4.x + c4.y + c4.z + c4.w;
#else
        if (tid < nd4) st4(a.CL + ((size_t)(b0 + h) * T + t) * D + 4 * d4, c4);
#endif
        pe[h] = 0.f;
        if (HAS_LW) {
            w4.x += bl.x; w4.y += bl.y; w4.z += bl.z; w4.w += bl.w;
            pe[h] = on * dot4_tanh(w4, s3[h], ult);
        }
    }
    if (HAS_LW) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float rr = wave_sum(pe[h]);
            if (lane == 0) s_red[w * H + h] = rr;        // (every read of the first use is behind the second barrier above)
        }
        __syncthreads();
        if (tid < H) a.elt[(size_t)(b0 + tid) * T + t] = s_red[tid] + s_red[H + tid] + s_red[2 * H + tid] + s_red[3 * H + tid] + clt0;
    }
}

template <int H, bool HAS_LW>
__global__ __launch_bounds__(256, 2) void spatial_shared_cols_kernel(const SpatialArgs a) { spatial_shared_cols_body<H, HAS_LW>(a, (int)blockIdx.x); }

template <int H, bool HAS_LW>
__global__ __launch_bounds__(256, 2) void spatial_shared_cols_update_kernel(const SpatialArgs a, const BeamArgs u) {
    if ((int)blockIdx.x < u.nvid) { beam_update_body(u, 0, nullptr, nullptr, (int)blockIdx.x, u.nvid); return; }
    spatial_shared_cols_body<H, HAS_LW>(a, (int)blockIdx.x - u.nvid);
}

// K <= 8 regions, D <= 1024: one float4 column per lane, one memory round trip per item (spatial_shared_cols_body)
static bool spatial_shared_cols(const SpatialArgs& a) {
    // OFF unless STATTN_SHARED_COLS=1: measured once at the evaluation shape (51 videos x 28 frames, beam 5) it ran 76 us against the
    // 77 us of spatial_shared_kernel<5> -- not yet the 30 us its bytes allow, and not yet through the parity suite
    static const char* cols = getenv("STATTN_SHARED_COLS");
    return cols && cols[0] == '1' && a.K <= 8 && a.D <= 1024 && a.group <= 6;        // (7 / 8 hypotheses with LW: 256 VGPRs and a spill)
}
// launch hooks of launch_spatial (shared-slab path); return true when they launched
static bool exp_launch_shared_cols_update(hipStream_t s, const SpatialArgs& a, const BeamArgs& upd, dim3 grid, dim3 block) {
    if (!spatial_shared_cols(a)) return false;
#define STATTN_COLS_UPD(HH) case HH: if (a.LW) hipLaunchKernelGGL((spatial_shared_cols_update_kernel<HH, true>), grid, block, 0, s, a, upd); \
                             else hipLaunchKernelGGL((spatial_shared_cols_update_kernel<HH, false>), grid, block, 0, s, a, upd); break;
    switch (a.group) {
        STATTN_COLS_UPD(2) STATTN_COLS_UPD(3) STATTN_COLS_UPD(4) STATTN_COLS_UPD(5)
        default: if (a.LW) hipLaunchKernelGGL((spatial_shared_cols_update_kernel<6, true>), grid, block, 0, s, a, upd);       // (group <= 6: spatial_shared_cols)
                 else hipLaunchKernelGGL((spatial_shared_cols_update_kernel<6, false>), grid, block, 0, s, a, upd); break;
    }
#undef STATTN_COLS_UPD
    return true;
}
static bool exp_launch_shared_cols(hipStream_t s, const SpatialArgs& a, dim3 grid, dim3 block) {
    if (!spatial_shared_cols(a)) return false;
#define STATTN_COLS(HH) case HH: if (a.LW) hipLaunchKernelGGL((spatial_shared_cols_kernel<HH, true>), grid, block, 0, s, a); \
                         else hipLaunchKernelGGL((spatial_shared_cols_kernel<HH, false>), grid, block, 0, s, a); break;
    switch (a.group) {
        STATTN_COLS(2) STATTN_COLS(3) STATTN_COLS(4) STATTN_COLS(5)
        default: if (a.LW) hipLaunchKernelGGL((spatial_shared_cols_kernel<6, true>), grid, block, 0, s, a);       // (group <= 6: spatial_shared_cols)
                 else hipLaunchKernelGGL((spatial_shared_cols_kernel<6, false>), grid, block, 0, s, a); break;
    }
#undef STATTN_COLS
    return true;
}
