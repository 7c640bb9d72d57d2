// Backward pass (BPTT) kernels of the decoder for gfx950 -- what the reference obtains from
// `tensor.grad(cost, wrt=itemlist(tparams))` (model_attention.py:1193) over the graph of
// build_model (:583-717) + the loss terms (:1129-1147).  lt_mode 1 only.
//
// Structure.  Inside the reverse time loop only what the recurrence needs is computed:
//   lstm_bwd  -> dpre (gate pre-activation grads), carried dc, pass-through dh
//   [skinny GEMM: dctx = dpre.Wc^T, dhU = dpre.U^T]
//   spatial_bwd  -> dcsum, selector grad, temporal softmax backwards; dplt, spatial softmax backward (del), per-frame dsl
//                   (+ the riding GEMM dhU = dpre.U^T in extra workgroups)
//   reduce_T     -> dsl, dslt summed over frames
//   [skinny GEMM: dhW = dsproj.[Wdl|Wdg|Wdm|Wdlt]^T]
// Every gradient that accumulates over time WITHOUT feeding the recurrence (dPL, dL, dLW, dPG, dPM,
// dMo, the U*_att vectors and all weight matrices) is deferred: the per-step factors (del, deg, dem,
// dplt, dcsum, dpre, dsproj) are stored -- HBM is 288 GB -- and ctxgrad_kernel / batched MFMA GEMMs
// consume them once after the loop, instead of a read-modify-write of 54 MB tensors in every step.
#include <cstdlib>
#include "kernels.h"
#include "devmath.h"
#include "panel_inl.h"

namespace stattn {

namespace {

constexpr int KMAX = 64;

template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* s_red /*[nwaves][N]*/, int tid, int nwaves) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float r = wave_sum(v[i]);
        if (lane == 0) s_red[w * N + i] = r;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float r = 0.f;
        for (int k = 0; k < nwaves; ++k) r += s_red[k * N + i];
        v[i] = r;
    }
    __syncthreads();
}

// four consecutive values of a region tensor at element offset `off`: fp32, or -- bf16 handles (BF): the tensor is stored
// as bf16 behind the float pointer (steps.cpp project_context_bf16) -- one 8-byte load widened exactly
template <bool BF>
__device__ __forceinline__ float4 lds4(const float* base, size_t off) {
    if constexpr (BF) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                           __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    } else {
        return ld4(base + off);
    }
}

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 tanh4s(float4 x, float4 s) {
    return make_float4(fast_tanh(x.x + s.x), fast_tanh(x.y + s.y), fast_tanh(x.z + s.z), fast_tanh(x.w + s.w));
}
__device__ __forceinline__ float4 one_minus_sq(float4 t) {
    return make_float4(1.f - t.x * t.x, 1.f - t.y * t.y, 1.f - t.z * t.z, 1.f - t.w * t.w);
}
__device__ __forceinline__ float4 r_minus_sq(float4 r) {     // r - r^2 = (1 - tanh^2) / 4 for r = 1 / (1 + e^{2z})
    return make_float4(r.x - r.x * r.x, r.y - r.y * r.y, r.z - r.z * r.z, r.w - r.w * r.w);
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ void fma4(float4& acc, float s, float4 v) { acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w; }
__device__ __forceinline__ void add4(float4& acc, float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }

// ---------------------------------------------------------------------------------------------
// d(loss)/d(logit) for the masked NLL with the +1e-8 inside the log (:712):
//   dlogit[r,j] = w_r (p_j - [j == x_r]),  w_r = nll_scale * mask_r * p_x / (p_x + 1e-8);  pad columns = 0
__global__ __launch_bounds__(256) void dlogit_kernel(const float* __restrict__ probs, int ldp, const int64_t* __restrict__ x,
                                                     const float* __restrict__ mask, float nll_scale,
                                                     float* __restrict__ dl, int ldd, int V, int Vp) {
    const int r = blockIdx.x;
    int64_t xi = x[r];
    xi = xi < 0 ? 0 : (xi >= V ? V - 1 : xi);
    const float* p = probs + (size_t)r * ldp;
    const float px = p[xi];
    const float w = nll_scale * mask[r] * px / (px + 1e-8f);
    float* d = dl + (size_t)r * ldd;
    if (((ldp | ldd | Vp) & 3) == 0) {               // 16-byte loads and stores (the padded vocabulary always qualifies)
        const int x4 = (int)xi >> 2, xc = (int)xi & 3;
        for (int j4 = threadIdx.x; j4 < (Vp >> 2); j4 += 256) {
            float4 q = ld4(p + 4 * j4);
            if (j4 == x4) { if (xc == 0) q.x -= 1.f; else if (xc == 1) q.y -= 1.f; else if (xc == 2) q.z -= 1.f; else q.w -= 1.f; }
            const int j = 4 * j4;
            st4(d + j, make_float4(j < V ? w * q.x : 0.f, j + 1 < V ? w * q.y : 0.f, j + 2 < V ? w * q.z : 0.f, j + 3 < V ? w * q.w : 0.f));
        }
        return;
    }
    for (int j = threadIdx.x; j < Vp; j += 256) {
        float v = 0.f;
        if (j < V) v = w * (p[j] - (j == (int)xi ? 1.f : 0.f));
        d[j] = v;
    }
}

// r tensors of the doubly-stochastic regulariser (:1140-1147): d/d alpha[s,...] = -2 alpha_c / n * (1 - sum_s alpha).
// All four attentions in one launch (blockIdx.y = which): they were four 13 us latency-bound launches of a 30-step loop.
__global__ void alpha_reg_kernel(const AlphaRegArgs a, int steps) {
    const int w = blockIdx.y;
    const size_t n = a.n[w], i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* __restrict__ alpha = a.alpha[w];
    float s = 0.f;
#pragma unroll 6
    for (int t = 0; t < steps; ++t) s += alpha[(size_t)t * n + i];
    a.r[w][i] = -2.f * a.coef[w] * (1.f - s);
    if (a.sq[w]) a.sq[w][i] = (1.f - s) * (1.f - s);
}

// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void lstm_bwd_kernel(const LstmBwdArgs a) {
    const int D = a.D, nd4 = D >> 2;
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one lane = 4 consecutive units of one row
    if (i4 >= (size_t)a.M * nd4) return;
    const int b = (int)(i4 / nd4), d = 4 * (int)(i4 % nd4);
    const size_t idx = (size_t)b * D + d, MD = (size_t)a.M * D;
    float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!a.last) {
        dh = ld4(a.dh_pass + idx);
#pragma unroll 4
        for (int p = 0; p < a.nU; ++p) add4(dh, ld4(a.dhU + (size_t)p * MD + idx));      // (K-slice partials: loads in flight together)
#pragma unroll 4
        for (int p = 0; p < a.nW; ++p) add4(dh, ld4(a.dhW + (size_t)p * MD + idx));
        if (a.W_sel) fma4(dh, a.dselpre[b], ld4(a.W_sel + d));
    }
    add4(dh, mul4(ld4(a.dhd + idx), ld4(a.d1 + idx)));                  // hd = h * d1  (:684-685)
    const float* gt = a.gates + (size_t)b * 4 * D + d;
    const float4 gi = ld4(gt), gf = ld4(gt + D), go = ld4(gt + 2 * D), gg = ld4(gt + 3 * D);
    const float4 cp = ld4(a.c_prev + idx), cn = ld4(a.c_new + idx);
    const float m = a.mask[b];
    const float4 dcin = a.last ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(a.dc + idx);
    const float* dp = a.dp + (size_t)b * 3 * D + d;
    const float4 dpi = ld4(dp), dpf = ld4(dp + D), dpo = ld4(dp + 2 * D);
    float4 o_i, o_f, o_o, o_g, o_dc, o_pass;
#define STATTN_LSTM_BWD_LANE(X)                                                                                  \
    {                                                                                                            \
        const float tc = fast_tanh(cn.X);                                                                        \
        const float dcn = dcin.X + m * dh.X * go.X * (1.f - tc * tc);   /* h = m o tanh(c) + (1-m) h_ (:456-457) */  \
        const float dct = m * dcn;                                      /* c = m (f c_ + i g) + (1-m) c_ (:453-454) */ \
        o_dc.X = (1.f - m) * dcn + dct * gf.X;                                                                   \
        o_i.X = dct * gg.X * gi.X * (1.f - gi.X) * dpi.X;               /* i = sigma(pre_i * dp_i) (:445-448) */ \
        o_f.X = dct * cp.X * gf.X * (1.f - gf.X) * dpf.X;                                                        \
        o_o.X = m * dh.X * tc * go.X * (1.f - go.X) * dpo.X;                                                     \
        o_g.X = dct * gi.X * (1.f - gg.X * gg.X);                                                                \
        o_pass.X = (1.f - m) * dh.X;                                                                             \
    }
    STATTN_LSTM_BWD_LANE(x) STATTN_LSTM_BWD_LANE(y) STATTN_LSTM_BWD_LANE(z) STATTN_LSTM_BWD_LANE(w)
#undef STATTN_LSTM_BWD_LANE
    st4(a.dc + idx, o_dc);
    float* o = a.dpre + (size_t)b * 4 * D + d;
    st4(o, o_i); st4(o + D, o_f); st4(o + 2 * D, o_o); st4(o + 3 * D, o_g);
    if (a.dpre_pk) {    // once more in the packed A layout of the row-panel recurrences (dctx, dhU)
        const int S = (4 * D) >> 4;
        st4(a.dpre_pk + pn_pack_offset(b, d, S), o_i); st4(a.dpre_pk + pn_pack_offset(b, D + d, S), o_f);
        st4(a.dpre_pk + pn_pack_offset(b, 2 * D + d, S), o_o); st4(a.dpre_pk + pn_pack_offset(b, 3 * D + d, S), o_g);
    }
    st4(a.dh_pass_out + idx, o_pass);
}

// ---------------------------------------------------------------------------------------------

// Attention backward of one step, one workgroup per (row b, frame t).
//
// Temporal part (a launch of its own until round 3):
//   dctx = readout term + the K-slice partials of dpre.Wc^T;  selector backward (ctx = sel * csum, :433-435) -> dcsum, dselpre
//   d alpha_x[t] = <dcsum, X_t> + regulariser  for the three temporal attentions (cg = sum_t ag_t G_t :399, cm :412,
//   clt = sum_t alt_t CL_t :426), then the softmax backward  de_x[t] = alpha_x[t] (d alpha_x[t] - <alpha_x, d alpha_x>).
// The softmax backward looked like a dependency on every frame of the row, but
//   <alpha_x, d alpha_x> = <dcsum, sum_t alpha_x[t] X_t> + <alpha_x, r_x> = <dcsum, c_x> + <alpha_x, r_x>
// with c_x the context vector the forward pass formed (TemporalArgs::cparts): one more dot product over D per
// workgroup and a T-long one over the regulariser terms.  Every frame's workgroup re-forms dcsum of its row from the
// partials (a few 4 KB vectors from L2); frame 0 stores it for the deferred context gradients.
//
// Spatial part: dplt, spatial softmax backward (del), per-frame dsl / dsg / dsm.
// KR = regions whose LW rows a lane holds in registers across the two uses (plt, then the d alpha dot products): 8 at four
// waves per SIMD (128 VGPRs), 16 at three (168) -- for 8 < K <= 16 (configs[3]) that saves the second pass over the LW slab
// of the generic path.
template <int KR, bool BF>
__global__ __launch_bounds__(256, KR <= 8 ? 4 : 3) void spatial_bwd_kernel(const SpatialBwdArgs a) {
    __shared__ float s_red[4 * KR];
    __shared__ float s_al[KMAX], s_da[KMAX];
    // the first rider.nblocks workgroups compute the rider GEMM (dhU = dpre.U^T, needed only by the NEXT reverse step's
    // lstm_bwd) on the matrix cores this HBM-bound kernel leaves idle
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[4 * 64 * 16];
        rider_tile<4>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    // all frames of a batch row on one XCD: they share the row's sproj, dcsum partials, csum / cparts
    const int bt = xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T), b = bt / T, tid = threadIdx.x;
    const size_t slab = (size_t)bt * K * D;
    auto ldPL = [&](size_t o) { return lds4<BF>(a.PL, slab + o); };      // BF: the slabs are bf16 (half the stream of this kernel)
    auto ldL = [&](size_t o) { return lds4<BF>(a.L, slab + o); };
    auto ldLW = [&](size_t o) { return lds4<BF>(a.LW, slab + o); };
    const float* __restrict__ sp = a.sproj + (size_t)b * a.ldsp;
    const int nd4 = D >> 2;
    __shared__ float s_de[3];
    if (tid < K) s_al[tid] = a.alphal[(size_t)bt * K + tid];
    // The LW rows of this lane's first column group are requested before anything else: they do not depend on the
    // temporal part below, whose loads / reduction / barrier would otherwise be an exposed latency stage in front of them.
    // (half of them: all eight would not fit the 128-VGPR budget next to the temporal part's own loads)
    constexpr int NPF = 4;
    float4 lw0[NPF];
    if (K <= KR) {
#pragma unroll
        for (int kk = 0; kk < NPF; ++kk) lw0[kk] = ldLW((size_t)min(kk, K - 1) * D + 4 * min(tid, nd4 - 1));
    }
    // ---- temporal part
    const int tf = bt - b * T;
    const size_t MD = (size_t)a.M * D;
    const float sel = a.has_sel ? a.sel[b] : 1.f;
    auto form_dcs = [&](int d4) {          // dcsum[b, 4 d4 ..] = sel * (readout term + partials of dpre.Wc^T)
        const size_t ob = (size_t)b * D + 4 * d4;
        float4 dc = a.dctx_r ? ld4(a.dctx_r + ob) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int q = 0; q < a.nP; ++q) add4(dc, ld4(a.dctxP + (size_t)q * MD + ob));      // (four partials in flight together: not unrolled,
        return dc;                                                                          //  every partial was a round trip of its own)
    };
    // dcsum of the row is needed again by the spatial part: parked in LDS (D <= 2048), else re-formed from the partials
    constexpr int DCS_LDS = 2048;
    __shared__ __attribute__((aligned(16))) float s_dcs[DCS_LDS];
    const bool dcs_lds = D <= DCS_LDS;
    {
        float q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = 0.f;
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            const size_t ob = (size_t)b * D + 4 * d4, o = (size_t)bt * D + 4 * d4;
            // every operand of the seven dot products is requested before dcsum is formed (one round trip, not four)
            const float4 xc = ld4(a.csum + ob), xg = ld4(a.G + o), xm = ld4(a.Mo + o), xl = ld4(a.CL + o);
            const float4 x0 = ld4(a.cparts + ob), x1 = ld4(a.cparts + MD + ob), x2 = ld4(a.cparts + 2 * MD + ob);
            const float4 dc = form_dcs(d4);
            q[6] += dot4(dc, xc);
            const float4 dcs = scale4(dc, sel);
            if (dcs_lds) st4(&s_dcs[4 * d4], dcs);
            if (tf == 0) st4(a.dcsum + ob, dcs);
            q[0] += dot4(dcs, xg);
            q[1] += dot4(dcs, xm);
            q[2] += dot4(dcs, xl);
            q[3] += dot4(dcs, x0);
            q[4] += dot4(dcs, x1);
            q[5] += dot4(dcs, x2);
        }
        block_sum<8>(q, s_red, tid, 4);
        // softmax backward of the three temporal attentions for this frame (wave 0..2; the T-long <alpha, r> only with
        // the regulariser on)
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
    }
    __syncthreads();
    const float alt = a.alt[bt], delt = s_de[2];
    // per-frame dsg / dsm: de Ug (1 - tanh^2(PG_t + sg))   (:389-397, :402-410)
    for (int d4 = tid; d4 < nd4; d4 += 256) {
        const size_t fo = (size_t)bt * D + 4 * d4;
        const float4 tg = tanh4s(ld4(a.PG + fo), ld4(sp + D + 4 * d4));
        const float4 tm = tanh4s(ld4(a.PM + fo), ld4(sp + 2 * D + 4 * d4));
        st4(a.dsgp + fo, scale4(mul4(ld4(a.Ug + 4 * d4), one_minus_sq(tg)), s_de[0]));
        st4(a.dsmp + fo, scale4(mul4(ld4(a.Um + 4 * d4), one_minus_sq(tm)), s_de[1]));
    }

    // pass 1+2 fused per lane: recompute plt = sum_k alpha_k LW_k + blt, dplt = delt Ult (1 - tanh^2(plt + slt)) (:416-422),
    // then dalpha_k = <alt dcsum, L_k> + <dplt, LW_k> + r_k (CL = sum alpha L :383).  dplt of a lane only needs the
    // lane's own columns, so for K <= KR the LW slab is read ONCE and held in registers for both uses.
    if (K <= KR) {
        float p[KR];
#pragma unroll
        for (int i = 0; i < KR; ++i) p[i] = 0.f;
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            float4 lw[KR];
#pragma unroll
            for (int kk = 0; kk < KR; ++kk)
                lw[kk] = (kk < NPF && d4 == tid) ? lw0[kk < NPF ? kk : 0] : ldLW((size_t)min(kk, K - 1) * D + 4 * d4);
            float4 pl = ld4(a.blt + 4 * d4);
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) if (kk < K) fma4(pl, s_al[kk], lw[kk]);
            const float4 th = tanh4s(pl, ld4(sp + 3 * D + 4 * d4));
            const float4 dpl = scale4(mul4(ld4(a.Ult + 4 * d4), one_minus_sq(th)), delt);
            st4(a.dplt + (size_t)bt * D + 4 * d4, dpl);
            const float4 dcl = scale4(dcs_lds ? ld4(&s_dcs[4 * d4]) : scale4(form_dcs(d4), sel), alt);
#pragma unroll
            for (int kk = 0; kk < KR; ++kk)
                p[kk] += dot4(dcl, ldL((size_t)min(kk, K - 1) * D + 4 * d4)) + dot4(dpl, lw[kk]);
        }
        block_sum<KR>(p, s_red, tid, 4);
        if (tid < KR && tid < K) s_da[tid] = p[tid] + (a.rl ? a.rl[(size_t)bt * K + tid] : 0.f);
        __syncthreads();
    } else {
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            float4 pl = ld4(a.blt + 4 * d4);
#pragma unroll 8
            for (int k = 0; k < K; ++k) fma4(pl, s_al[k], ldLW((size_t)k * D + 4 * d4));
            const float4 th = tanh4s(pl, ld4(sp + 3 * D + 4 * d4));
            st4(a.dplt + (size_t)bt * D + 4 * d4, scale4(mul4(ld4(a.Ult + 4 * d4), one_minus_sq(th)), delt));
        }
        __syncthreads();
        for (int k0 = 0; k0 < K; k0 += 8) {
            float p[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = 0.f;
            for (int d4 = tid; d4 < nd4; d4 += 256) {
                const float4 dcl = scale4(dcs_lds ? ld4(&s_dcs[4 * d4]) : scale4(form_dcs(d4), sel), alt);
                const float4 dpl = ld4(a.dplt + (size_t)bt * D + 4 * d4);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const size_t o = (size_t)min(k0 + kk, K - 1) * D + 4 * d4;
                    p[kk] += dot4(dcl, ldL(o)) + dot4(dpl, ldLW(o));
                }
            }
            block_sum<8>(p, s_red, tid, 4);
            if (tid < 8 && k0 + tid < K) s_da[k0 + tid] = p[tid] + (a.rl ? a.rl[(size_t)bt * K + k0 + tid] : 0.f);
            __syncthreads();
        }
    }
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
    // pass 3: dsl (this frame) = sum_k del_k Ul (1 - tanh^2(PL_k + sl))
    for (int d4 = tid; d4 < nd4; d4 += 256) {
        const float4 sl = ld4(sp + 4 * d4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int k = 0; k < K; ++k) fma4(acc, s_da[k], one_minus_sq(tanh4s(ldPL((size_t)k * D + 4 * d4), sl)));
        st4(a.dslp + (size_t)bt * D + 4 * d4, mul4(acc, ld4(a.Ul + 4 * d4)));
    }
}

#if STATTN_EXPERIMENTAL
#include "experimental/bwd_spatial_bwd2.inl"         // spatial_bwd2_kernel: unmeasured, never in the product build
#endif

// ---- bf16 handles, K <= 16: the same item (row b, frame t) with 128 threads and EIGHT columns per lane, so that the bf16 slabs
// are read with 16-byte loads (the 4-column form above reads them 8 bytes at a time: 3.2 TB/s of its bytes at configs[3]).  The LW
// rows of a lane stay in registers as packed bf16 (4 VGPRs per region) between their two uses.  Same arithmetic, fp32 throughout.
struct F8 { float v[8]; };
__device__ __forceinline__ F8 ld8f(const float* p) {
    const float4 a = ld4(p), b = ld4(p + 4);
    return F8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void st8f(float* p, const F8& x) {
    st4(p, make_float4(x.v[0], x.v[1], x.v[2], x.v[3])); st4(p + 4, make_float4(x.v[4], x.v[5], x.v[6], x.v[7]));
}
__device__ __forceinline__ F8 widen8(const uint4 u) {
    return F8{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
               __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
}
__device__ __forceinline__ uint4 ld16b(const float* base, size_t off) {        // 8 bf16 of a region tensor stored behind a float pointer
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off);
}
__device__ __forceinline__ float dot8(const F8& a, const F8& b) {
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a.v[i] * b.v[i];
    return r;
}

// (KR = 16 at three waves per SIMD: 168 VGPRs and 43 spilled dwords per lane, the 16 packed LW rows alone are 64 -- ISA, round 5.
//  -DSTATTN_BWD_BF16_WPS16=2 builds it for two waves per SIMD, 256 VGPRs and no spill, four workgroups per CU instead of six: unmeasured)
#ifndef STATTN_BWD_BF16_WPS16
#define STATTN_BWD_BF16_WPS16 3
#endif
template <int KR>
__global__ __launch_bounds__(128, KR <= 8 ? 4 : STATTN_BWD_BF16_WPS16) void spatial_bwd_bf16_kernel(const SpatialBwdArgs a) {
    constexpr int NT = 128, NW = 2;
    __shared__ float s_red[NW * (KR > 8 ? KR : 8)];
    __shared__ float s_al[KMAX], s_da[KMAX];
    if ((int)blockIdx.x < a.rider.nblocks) {
        __shared__ __attribute__((aligned(16))) float s_rider[NW * 64 * 16];
        rider_tile<NW>(a.rider, (int)blockIdx.x, s_rider);
        return;
    }
    const int T = a.T, K = a.K, D = a.D;
    const int bt = xcd_rows((int)blockIdx.x - a.rider.nblocks, a.M, T), b = bt / T, tid = threadIdx.x;
    const size_t slab = (size_t)bt * K * D;
    const float* __restrict__ sp = a.sproj + (size_t)b * a.ldsp;
    const int nd8 = D >> 3;
    __shared__ float s_de[3];
    if (tid < K) s_al[tid] = a.alphal[(size_t)bt * K + tid];
    const int tf = bt - b * T;
    const size_t MD = (size_t)a.M * D;
    const float sel = a.has_sel ? a.sel[b] : 1.f;
    auto form_dcs = [&](int d8) {          // dctx[b, 8 d8 ..] = readout term + partials of dpre.Wc^T
        const size_t ob = (size_t)b * D + 8 * d8;
        F8 dc = a.dctx_r ? ld8f(a.dctx_r + ob) : F8{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
        for (int q = 0; q < a.nP; ++q) {
            const F8 pq = ld8f(a.dctxP + (size_t)q * MD + ob);
#pragma unroll
            for (int i = 0; i < 8; ++i) dc.v[i] += pq.v[i];
        }
        return dc;
    };
    constexpr int DCS_LDS = 2048;
    __shared__ __attribute__((aligned(16))) float s_dcs[DCS_LDS];
    const bool dcs_lds = D <= DCS_LDS;
    {   // ---- temporal part (see spatial_bwd_kernel)
        float q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = 0.f;
        for (int d8 = tid; d8 < nd8; d8 += NT) {
            const size_t ob = (size_t)b * D + 8 * d8, o = (size_t)bt * D + 8 * d8;
            const F8 xc = ld8f(a.csum + ob), xg = ld8f(a.G + o), xm = ld8f(a.Mo + o), xl = ld8f(a.CL + o);
            const F8 x0 = ld8f(a.cparts + ob), x1 = ld8f(a.cparts + MD + ob), x2 = ld8f(a.cparts + 2 * MD + ob);
            const F8 dc = form_dcs(d8);
            q[6] += dot8(dc, xc);
            F8 dcs;
#pragma unroll
            for (int i = 0; i < 8; ++i) dcs.v[i] = dc.v[i] * sel;
            if (dcs_lds) st8f(&s_dcs[8 * d8], dcs);
            if (tf == 0) st8f(a.dcsum + ob, dcs);
            q[0] += dot8(dcs, xg); q[1] += dot8(dcs, xm); q[2] += dot8(dcs, xl);
            q[3] += dot8(dcs, x0); q[4] += dot8(dcs, x1); q[5] += dot8(dcs, x2);
        }
        block_sum<8>(q, s_red, tid, NW);
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
    const float alt = a.alt[bt], delt = s_de[2], deg = s_de[0], dem = s_de[1];
    auto dcs_of = [&](int d8) {
        if (dcs_lds) return ld8f(&s_dcs[8 * d8]);
        F8 dc = form_dcs(d8);
#pragma unroll
        for (int i = 0; i < 8; ++i) dc.v[i] *= sel;
        return dc;
    };
    // per-frame dsg / dsm
    for (int d8 = tid; d8 < nd8; d8 += NT) {
        const size_t fo = (size_t)bt * D + 8 * d8;
        const F8 pg = ld8f(a.PG + fo), pm = ld8f(a.PM + fo), sg = ld8f(sp + D + 8 * d8), sm = ld8f(sp + 2 * D + 8 * d8);
        const F8 ug = ld8f(a.Ug + 8 * d8), um = ld8f(a.Um + 8 * d8);
        F8 og, om;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tg = fast_tanh(pg.v[i] + sg.v[i]), tm = fast_tanh(pm.v[i] + sm.v[i]);
            og.v[i] = ug.v[i] * (1.f - tg * tg) * deg; om.v[i] = um.v[i] * (1.f - tm * tm) * dem;
        }
        st8f(a.dsgp + fo, og); st8f(a.dsmp + fo, om);
    }
    // plt recomputed, dplt, d alpha_k = <alt dcsum, L_k> + <dplt, LW_k> + r_k; the lane's LW rows stay packed in registers
    {
        float p[KR];
#pragma unroll
        for (int i = 0; i < KR; ++i) p[i] = 0.f;
        for (int d8 = tid; d8 < nd8; d8 += NT) {
            uint4 lw[KR];
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) lw[kk] = ld16b(a.LW, slab + (size_t)min(kk, K - 1) * D + 8 * d8);
            F8 pl = ld8f(a.blt + 8 * d8);
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) if (kk < K) {
                const F8 x = widen8(lw[kk]);
                const float al = s_al[kk];
#pragma unroll
                for (int i = 0; i < 8; ++i) pl.v[i] += al * x.v[i];
            }
            const F8 slt = ld8f(sp + 3 * D + 8 * d8), ult = ld8f(a.Ult + 8 * d8);
            F8 dpl;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float th = fast_tanh(pl.v[i] + slt.v[i]); dpl.v[i] = ult.v[i] * (1.f - th * th) * delt; }
            st8f(a.dplt + (size_t)bt * D + 8 * d8, dpl);
            F8 dcl = dcs_of(d8);
#pragma unroll
            for (int i = 0; i < 8; ++i) dcl.v[i] *= alt;
            // (four L rows in flight at a time: all KR at once spill next to the KR packed LW rows)
#pragma unroll
            for (int k4 = 0; k4 < KR; k4 += 4) {
                uint4 lr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) lr[j] = ld16b(a.L, slab + (size_t)min(k4 + j, K - 1) * D + 8 * d8);
#pragma unroll
                for (int j = 0; j < 4; ++j) p[k4 + j] += dot8(dcl, widen8(lr[j])) + dot8(dpl, widen8(lw[k4 + j]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        block_sum<KR>(p, s_red, tid, NW);
        if (tid < KR && tid < K) s_da[tid] = p[tid] + (a.rl ? a.rl[(size_t)bt * K + tid] : 0.f);
        __syncthreads();
    }
    float dotp = 0.f;
    for (int k = 0; k < K; ++k) dotp += s_al[k] * s_da[k];
    __syncthreads();
    if (tid < K) {
        const float de = s_al[tid] * (s_da[tid] - dotp);
        s_da[tid] = de;
        a.del[(size_t)bt * K + tid] = de;
    }
    __syncthreads();
    // dsl (this frame) = Ul sum_k del_k (1 - tanh^2(PL_k + sl))
    for (int d8 = tid; d8 < nd8; d8 += NT) {
        const F8 sl = ld8f(sp + 8 * d8), ul = ld8f(a.Ul + 8 * d8);
        F8 acc{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
#pragma unroll 8
        for (int k = 0; k < K; ++k) {
            const F8 x = widen8(ld16b(a.PL, slab + (size_t)k * D + 8 * d8));
            const float de = s_da[k];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float th = fast_tanh(x.v[i] + sl.v[i]); acc.v[i] += de * (1.f - th * th); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] *= ul.v[i];
        st8f(a.dslp + (size_t)bt * D + 8 * d8, acc);
    }
}

#if STATTN_EXPERIMENTAL
#include "experimental/bwd_spatial_bwd_bf16v2.inl"   // spatial_bwd_bf16v2_kernel: unmeasured, never in the product build
#endif

// dsproj[b] = [sum_t dslp | sum_t dsgp | sum_t dsmp | sum_t dplt]   (the four state-projection gradients)
__global__ __launch_bounds__(256) void reduce_T_kernel(const float* __restrict__ dslp, const float* __restrict__ dsgp,
                                                       const float* __restrict__ dsmp, const float* __restrict__ dplt,
                                                       float* __restrict__ dsproj, int lddsp, int T, int D,
                                                       float* __restrict__ dsproj_pk) {
    const int b = blockIdx.x, which = blockIdx.y, d4 = blockIdx.z * 256 + threadIdx.x;
    if (d4 >= (D >> 2)) return;
    const float* __restrict__ src = which == 0 ? dslp : (which == 1 ? dsgp : (which == 2 ? dsmp : dplt));
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
    // frames in chunks of 16 whose loads are all in flight together, the last chunk predicated (a plain `unroll 16` left a
    // serial remainder loop: 8.8 us; `unroll 8`: 6.9 us; this: same frame order, T = 26 in two rounds)
    for (int t0 = 0; t0 < T; t0 += 16) {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = t0 + i < T ? ld4(src + ((size_t)b * T + t0 + i) * D + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) add4(s1, v[i]);
    }
    st4(dsproj + (size_t)b * lddsp + which * D + 4 * d4, s1);
    if (dsproj_pk) st4(dsproj_pk + pn_pack_offset(b, which * D + 4 * d4, (4 * D) >> 4), s1);    // packed A layout (dhW)
}

// ---------------------------------------------------------------------------------------------
// Deferred context gradients, one workgroup per (row b, frame t), looping over the time steps:
//   dPL[k,:] = sum_s del[s,k] Ul (1 - tanh^2(PL_k + sl_s))        dL[k,:]  = sum_s alpha_l[s,k] alt_s dcsum_s
//   dLW[k,:] = sum_s alpha_l[s,k] dplt_s                           dPG = sum_s deg_s Ug (1 - tanh^2(PG + sg_s)), dPM likewise
//   dMo      = sum_s am_s dcsum_s
// and the per-(b,t) partials of dUl, dUlt, dUg, dUm (summed over rows by colsum afterwards).

// Work split: NG + 1 workgroups per (b,t) item, NG = ceil(K / 4): region group g < NG owns four regions over ALL steps,
// workgroup NG the frame-level tensors and the dUlt partial.  The workgroups of an item read the same per-step operands
// (sproj, dcsum, dplt rows: 120 KB per item over the 30 steps).  When one workgroup walked the region groups one after
// the other, the second walk came 30 steps after the first and found nothing of it in the 4 MB L2 of its XCD (measured:
// 1.39 GB through the fabric for 0.6 GB of distinct bytes).  Now they are separate workgroups placed on the SAME XCD and
// dispatched together (block n: XCD n % 8 by the round-robin dispatch), so they run side by side and the later readers hit L2.
//
// The step loop is a chain of (load the step's rows -> a few hundred VALU cycles); at three waves per SIMD nothing
// covered the load latency (313 us for a kernel whose VALU and HBM floors are 110 / 150 us).  Every workgroup kind therefore
// requests step s + 1 before it computes step s, and the kinds are split so that each fits 168 VGPRs with the second
// operand set (the dUlt partial needs the frame's whole LW slab in registers: it moved to the frame workgroup).
// KR = 8 (K <= 8): the frame workgroup also forms the dUlt partial, the frame's LW slab in registers (NG + 1 workgroups per
// item).  KR = 16 (K > 8): the dUlt partial has a workgroup of its own with up to 16 LW rows in registers (NG + 2): with the
// rows re-read from L2 every step, configs[3] (K = 16) took 1065 us; split at K <= 8 as well it is slower (291 vs 237 us).
template <int KR, bool BF>
__global__ __launch_bounds__(256, 3) void ctxgrad_kernel(const CtxGradArgs a, const int NG) {
    constexpr bool WITH_ULT = KR <= 8;           // frame workgroup also does the dUlt partial
    const int S = a.S, M = a.M, T = a.T, K = a.K, D = a.D;
    const int NW = NG + (WITH_ULT ? 1 : 2);
    const int n = blockIdx.x;
    int bt, grp;
    if (M & 7) { bt = (n / (8 * NW)) * 8 + (n & 7); grp = (n >> 3) % NW; }
    else {      // whole batch rows per XCD as well: the 30 steps' sproj / dcsum rows of a row are shared by its T frames
        const int x = n & 7, i = n >> 3, it = i / NW;
        grp = i - it * NW;
        bt = (x + 8 * (it / T)) * T + it % T;
    }
    if (bt >= M * T) return;                       // (grid rounded up to whole groups of 8 items)
    const int b = bt / T, tid = threadIdx.x;
    const int nd4 = D >> 2;
    const size_t slab = (size_t)bt * K * D, MT = (size_t)M * T;
    if (grp == NG) {
        // ---- frame-level tensors and the dUlt partial
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            const size_t fo = (size_t)bt * D + 4 * d4;
            const float4 blt = ld4(a.blt + 4 * d4);
            // e^{2 PG}, e^{2 PM} once: tanh(P + s_step) = 1 - 2 r with r = 1 / (1 + e^{2P} e^{2s}) (devmath.h exp2x4); the sums
            // below carry r and r - r^2 and are turned into tanh / 1 - tanh^2 after the loop
            const float4 pg = exp2x4(ld4(a.PG + fo)), pm = exp2x4(ld4(a.PM + fo));
            float4 lw[8];                          // the frame's LW slab (K <= 8): plt = blt + sum_k alpha_k LW_k is recomputed per step
#pragma unroll
            for (int k = 0; k < 8; ++k) lw[k] = (WITH_ULT && K <= 8) ? lds4<BF>(a.LW, slab + (size_t)min(k, K - 1) * D + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 dpg = make_float4(0.f, 0.f, 0.f, 0.f), dpm = dpg, dmo = dpg, ug = dpg, um = dpg, pult = dpg;
            float sdeg = 0.f, sdem = 0.f;
            struct FrameIn { float4 sg, sm, slt, dcs; float deg, dem, am, delt; float al[8]; };
            auto fetch_f = [&](int s_) {
                FrameIn r;
                const float* sp = a.sproj + ((size_t)s_ * M + b) * 4 * D;
                r.sg = ld4(sp + D + 4 * d4); r.sm = ld4(sp + 2 * D + 4 * d4);
                r.slt = WITH_ULT ? ld4(sp + 3 * D + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
                r.dcs = ld4(a.dcsum + ((size_t)s_ * M + b) * D + 4 * d4);
                r.deg = a.deg[s_ * MT + bt]; r.dem = a.dem[s_ * MT + bt]; r.am = a.am[s_ * MT + bt]; r.delt = a.delt[s_ * MT + bt];
                const float* als = a.alphal + (s_ * MT + bt) * K;
#pragma unroll
                for (int k = 0; k < 8; ++k) r.al[k] = WITH_ULT ? als[min(k, K - 1)] : 0.f;
                return r;
            };
            FrameIn nx = fetch_f(0);
            for (int s = 0; s < S; ++s) {
                const FrameIn cf = nx;
                nx = fetch_f(min(s + 1, S - 1));   // (unconditional: a branch around the loads would drain the queue)
                const float4 rg = rcp1p4(pg, exp2x4(cf.sg)), rm = rcp1p4(pm, exp2x4(cf.sm));
                fma4(dpg, cf.deg, r_minus_sq(rg)); fma4(ug, cf.deg, rg); sdeg += cf.deg;
                fma4(dpm, cf.dem, r_minus_sq(rm)); fma4(um, cf.dem, rm); sdem += cf.dem;
                fma4(dmo, cf.am, cf.dcs);
                if (!WITH_ULT) continue;
                // dUlt partial: delt * tanh(plt + slt), plt recomputed from all K regions
                float4 plt = blt;
                if (K <= 8) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k < K) fma4(plt, cf.al[k], lw[k]);
                } else {
                    const float* als = a.alphal + (s * MT + bt) * K;
                    for (int k = 0; k < K; ++k) fma4(plt, als[k], lds4<BF>(a.LW, slab + (size_t)k * D + 4 * d4));
                }
                fma4(pult, cf.delt, tanh4s(plt, cf.slt));
            }
            st4(a.dPG + fo, mul4(scale4(dpg, 4.f), ld4(a.Ug + 4 * d4)));        // sum deg (1 - tanh^2) = 4 sum deg (r - r^2)
            st4(a.dPM + fo, mul4(scale4(dpm, 4.f), ld4(a.Um + 4 * d4)));
            st4(a.dMo + fo, dmo);
            st4(a.pUg + fo, make_float4(sdeg - 2.f * ug.x, sdeg - 2.f * ug.y, sdeg - 2.f * ug.z, sdeg - 2.f * ug.w));   // sum deg tanh = sum deg (1 - 2 r)
            st4(a.pUm + fo, make_float4(sdem - 2.f * um.x, sdem - 2.f * um.y, sdem - 2.f * um.z, sdem - 2.f * um.w));
            if (WITH_ULT) st4(a.pUlt + fo, pult);
        }
        return;
    }
    if (!WITH_ULT && grp == NG + 1) {
        // ---- dUlt partial alone: sum_s delt_s tanh(plt_s + slt_s), plt_s = blt + sum_k alpha_{s,k} LW_k recomputed from the
        // frame's LW slab, in registers for K <= KR (a step's attention weights are uniform over the workgroup: scalar loads,
        // requested a step ahead like the rows)
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            const size_t fo = (size_t)bt * D + 4 * d4;
            const float4 blt = ld4(a.blt + 4 * d4);
            float4 lw[KR];
#pragma unroll
            for (int k = 0; k < KR; ++k) lw[k] = K <= KR ? lds4<BF>(a.LW, slab + (size_t)min(k, K - 1) * D + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 pult = make_float4(0.f, 0.f, 0.f, 0.f);
            struct UltIn { float4 slt; float delt; float al[KR]; };
            auto fetch_u = [&](int s_) {
                UltIn r;
                r.slt = ld4(a.sproj + ((size_t)s_ * M + b) * 4 * D + 3 * D + 4 * d4);
                r.delt = a.delt[s_ * MT + bt];
                const float* als = a.alphal + (s_ * MT + bt) * K;
#pragma unroll
                for (int k = 0; k < KR; ++k) r.al[k] = als[min(k, K - 1)];
                return r;
            };
            UltIn nx = fetch_u(0);
            for (int s = 0; s < S; ++s) {
                const UltIn cu = nx;
                nx = fetch_u(min(s + 1, S - 1));
                float4 plt = blt;
                if (K <= KR) {
#pragma unroll
                    for (int k = 0; k < KR; ++k) if (k < K) fma4(plt, cu.al[k], lw[k]);
                } else {
                    const float* als = a.alphal + (s * MT + bt) * K;
                    for (int k = 0; k < K; ++k) fma4(plt, als[k], lds4<BF>(a.LW, slab + (size_t)k * D + 4 * d4));
                }
                fma4(pult, cu.delt, tanh4s(plt, cu.slt));
            }
            st4(a.pUlt + fo, pult);
        }
        return;
    }
    // ---- region-level tensors, the four regions k0 .. k0 + 3.  (Eight at a time meant 40 float4 accumulators = one wave per
    // SIMD; four at a time re-reads the per-step operands K / 4 times -- L2 hits -- and runs three waves per SIMD.)
    const int k0 = 4 * grp;
    for (int d4 = tid; d4 < nd4; d4 += 256) {
        const float4 ul = ld4(a.Ul + 4 * d4);
        float4 pul = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 pl[4], dpl[4], dl[4], dlw[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            pl[kk] = exp2x4(lds4<BF>(a.PL, slab + (size_t)min(k0 + kk, K - 1) * D + 4 * d4));       // e^{2 PL} (see the frame part)
            dpl[kk] = make_float4(0.f, 0.f, 0.f, 0.f); dl[kk] = dpl[kk]; dlw[kk] = dpl[kk];
        }
        struct StepIn { float4 sl, dcs, dp; float alt; float al[4], de[4]; };
        auto fetch = [&](int s_) {
            StepIn r;
            r.sl = ld4(a.sproj + ((size_t)s_ * M + b) * 4 * D + 4 * d4);
            r.dcs = ld4(a.dcsum + ((size_t)s_ * M + b) * D + 4 * d4);
            r.dp = ld4(a.dplt + (s_ * MT + bt) * D + 4 * d4);
            r.alt = a.alt[s_ * MT + bt];
            const float* als = a.alphal + (s_ * MT + bt) * K;
            const float* des = a.del + (s_ * MT + bt) * K;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = min(k0 + kk, K - 1);
                r.al[kk] = als[k]; r.de[kk] = des[k];
            }
            return r;
        };
        float sde = 0.f;
        StepIn nx = fetch(0);
        for (int s = 0; s < S; ++s) {
            const StepIn cur = nx;
            nx = fetch(min(s + 1, S - 1));
            const float4 dcl = scale4(cur.dcs, cur.alt);
            const float4 esl = exp2x4(cur.sl);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float4 r = rcp1p4(pl[kk], esl);
                fma4(dpl[kk], cur.de[kk], r_minus_sq(r));
                if (k0 + kk < K) { fma4(pul, cur.de[kk], r); sde += cur.de[kk]; }
                fma4(dl[kk], cur.al[kk], dcl);
                fma4(dlw[kk], cur.al[kk], cur.dp);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (k0 + kk < K) {
                const size_t o = slab + (size_t)(k0 + kk) * D + 4 * d4;
                st4(a.dPL + o, mul4(scale4(dpl[kk], 4.f), ul));
                st4(a.dL + o, dl[kk]);
                st4(a.dLW + o, dlw[kk]);
            }
        }
        pul = make_float4(sde - 2.f * pul.x, sde - 2.f * pul.y, sde - 2.f * pul.z, sde - 2.f * pul.w);   // sum de tanh = sum de (1 - 2 r)
        st4(a.pUl + ((size_t)grp * MT + bt) * D + 4 * d4, pul);        // one partial per region group (summed by the column sums)
    }
}

// ---------------------------------------------------------------------------------------------
// column sums with a fixed two-stage order: part[rs][n] = sum over a row range; then dst[n] (+)= sum_rs part
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ X, int ldx, int rows, int N,
                                                          float* __restrict__ part, int rsplit, const float* __restrict__ rw) {
    __shared__ float s[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, rs = blockIdx.y;
    const int per = (rows + rsplit - 1) / rsplit;
    const int r0 = rs * per, r1 = min(rows, r0 + per);
    float acc = 0.f;
    if (n < N)
        for (int r = r0 + w; r < r1; r += 4) acc += (rw ? rw[r] : 1.f) * X[(size_t)r * ldx + n];
    s[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && n < N) part[(size_t)rs * N + n] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int rsplit, int N, float* __restrict__ dst, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc = 0.f;
#pragma unroll 16
    for (int r = 0; r < rsplit; ++r) acc += part[(size_t)r * N + n];
    dst[n] = accumulate ? dst[n] + acc : acc;
}
// the same for many matrices at once: bias gradients are ~17 such sums per backward pass, most of them far too
// small to fill the chip or to amortise a launch on their own.  256 columns (64 lanes x float4) per workgroup.
__global__ __launch_bounds__(256) void colsum_batch_part_kernel(const ColsumBatch b, float* __restrict__ part) {
    __shared__ float4 s[4][64];
    int ji = 0;
    while (ji + 1 < b.n && (int)blockIdx.x >= b.j[ji + 1].blk0) ++ji;
    const ColsumJob& j = b.j[ji];
    const int local = blockIdx.x - j.blk0;
    const int ncb = (j.N + 255) / 256;
    const int cb = local % ncb, slice = local / ncb;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = cb * 256 + 4 * lane;
    const int per = (j.rows + j.rs - 1) / j.rs;
    const int r0 = slice * per, r1 = min(j.rows, r0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < j.N) {
        const float* __restrict__ X = j.X + n;
        if (j.rw) {
            for (int r = r0 + w; r < r1; r += 4) fma4(acc, j.rw[r], ld4(X + (size_t)r * j.ldx));
        } else {
            int r = r0 + w;
            for (; r + 12 < r1; r += 16) {     // four independent loads in flight per lane
                const float4 x0 = ld4(X + (size_t)r * j.ldx), x1 = ld4(X + (size_t)(r + 4) * j.ldx);
                const float4 x2 = ld4(X + (size_t)(r + 8) * j.ldx), x3 = ld4(X + (size_t)(r + 12) * j.ldx);
                add4(acc, x0); add4(acc, x1); add4(acc, x2); add4(acc, x3);
            }
            for (; r < r1; r += 4) add4(acc, ld4(X + (size_t)r * j.ldx));
        }
    }
    s[w][lane] = acc;
    __syncthreads();
    if (w == 0 && n < j.N) {
        float4 t = s[0][lane];
        add4(t, s[1][lane]); add4(t, s[2][lane]); add4(t, s[3][lane]);
        st4(part + j.part0 + (size_t)slice * j.N + n, t);
    }
}
__global__ __launch_bounds__(256) void colsum_batch_final_kernel(const ColsumBatch b, const float* __restrict__ part) {
    int ji = 0;
    while (ji + 1 < b.n && (int)blockIdx.x >= b.j[ji + 1].fblk0) ++ji;
    const ColsumJob& j = b.j[ji];
    const int n = (blockIdx.x - j.fblk0) * 256 + threadIdx.x;
    if (n >= j.N) return;
    const float* __restrict__ p = part + j.part0 + n;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < j.rs; ++r) acc += p[(size_t)r * j.N];
    j.dst[n] = j.accumulate ? j.dst[n] + acc : acc;
}
// several independent full sums in one launch pair: 32 workgroups per job write partials, a second tiny kernel
// adds them in a fixed order (deterministic)
constexpr int MS_BLOCKS = 32;
__global__ __launch_bounds__(256) void multi_sum_part_kernel(const MultiSumArgs a, float* __restrict__ part) {
    __shared__ float s[4];
    const int job = blockIdx.y;
    const float* __restrict__ x = a.src[job];
    const size_t n = a.n[job];
    float acc = 0.f;
#pragma unroll 8
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)MS_BLOCKS * 256) acc += x[i];     // (unrolled: eight loads in flight)
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[job * MS_BLOCKS + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void multi_sum_final_kernel(const MultiSumArgs a, const float* __restrict__ part) {
    const int job = threadIdx.x;
    if (job >= a.count) return;
    float r = 0.f;
    for (int i = 0; i < MS_BLOCKS; ++i) r += part[job * MS_BLOCKS + i];
    a.dst[job][0] = a.scale[job] * r;
}
// dst[0] (+)= scale * sum(x[0:n])   (single block, deterministic)
__global__ __launch_bounds__(1024) void sum_all_kernel(const float* __restrict__ x, size_t n, float* __restrict__ dst, float scale, int accumulate) {
    __shared__ float s[16];
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) acc += x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        for (int i = 0; i < 16; ++i) r += s[i];
        dst[0] = accumulate ? dst[0] + scale * r : scale * r;
    }
}
// y = dy * (1 - t^2) [* mulmat]  elementwise (tanh backward), in place allowed
template <bool TBF>        // TBF: t is stored as bf16 (the region tensor L of a bf16 handle)
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ t, const float* __restrict__ mul,
                                float* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = mul4(ld4(dy + 4 * i), one_minus_sq(lds4<TBF>(t, 4 * i)));
        if (mul) v = mul4(v, ld4(mul + 4 * i));
        st4(out + 4 * i, v);
    }
}
// out[i] (+)= a[i] (+ b[i])
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n4, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = ld4(a + 4 * i);
        if (b) add4(v, ld4(b + 4 * i));
        if (accumulate) add4(v, ld4(out + 4 * i));
        st4(out + 4 * i, v);
    }
}
// dWemb[x[i], :] = sum over the tokens i of one word of demb[i + shift, :]   (embedding lookup backward, :613-617)
// without atomics, so that the gradient is run-to-run reproducible.  The token ids are host data: stattn_set_batch
// sorts them once (stable, by word) into a plan -- perm[] (token indices grouped by word), pieces of at most 16 tokens,
// and for every distinct word its run of pieces.  Stage 1: one workgroup per piece sums its rows in index order (straight
// into dWemb when the word has a single piece); stage 2: one workgroup per multi-piece word adds the piece sums in
// order (the zero-padding word has hundreds of tokens in a batch).  Rows of absent words stay zero (memset).
__global__ __launch_bounds__(128) void embed_bwd_piece_kernel(const EmbedPlan pl, const float* __restrict__ demb, float* __restrict__ dWemb,
                                                              float* __restrict__ part, int E, int shift) {
    const int s = blockIdx.x;
    const int beg = pl.piece_start[s], end = pl.piece_start[s + 1];
    const int wi = pl.piece_word[s];                                    // index into the distinct-word list
    const bool single = pl.word_piece_start[wi + 1] - pl.word_piece_start[wi] == 1;
    float* __restrict__ dst = single ? dWemb + (size_t)pl.word_id[wi] * E : part + (size_t)s * E;
    for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += 128) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int j = beg; j < end; ++j) add4(acc, ld4(demb + (size_t)(pl.perm[j] + shift) * E + 4 * e4));     // (same order; four rows in flight)
        st4(dst + 4 * e4, acc);
    }
}
__global__ __launch_bounds__(128) void embed_bwd_word_kernel(const EmbedPlan pl, const float* __restrict__ part, float* __restrict__ dWemb, int E) {
    const int wi = pl.multi_word[blockIdx.x];
    const int beg = pl.word_piece_start[wi], end = pl.word_piece_start[wi + 1];
    for (int e4 = threadIdx.x; e4 < (E >> 2); e4 += 128) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int s = beg; s < end; ++s) add4(acc, ld4(part + (size_t)s * E + 4 * e4));
        st4(dWemb + (size_t)pl.word_id[wi] * E + 4 * e4, acc);
    }
}
// out[c, r] = in[r, c]   (weight transposes for the backward skinny GEMMs), 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                        int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * ldi + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * ldo + r0 + tx] = tile[tx][i];
}

// ---------------------------------------------------------------------------------------------
// optimizer: || g + 2 decay_c theta ||^2 (two-stage, fixed order), then L2 decay + clip + Adadelta in ONE pass over the
// buffers (common.py:178-195).  The decayed gradient is never written back: both passes form it on the fly, which
// saves a 171 MB store and reload per update.
__global__ __launch_bounds__(256) void decay_sumsq_kernel(const float* __restrict__ g, const float* __restrict__ p, float two_decay,
                                                          size_t n, float* __restrict__ part) {
    __shared__ float s[4];
    float acc = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = ld4(g + 4 * i), b = ld4(p + 4 * i);
        const float x = a.x + two_decay * b.x, y = a.y + two_decay * b.y, z = a.z + two_decay * b.z, w = a.w + two_decay * b.w;
        acc += x * x + y * y + z * z + w * w;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = g[i] + two_decay * p[i];
        acc += v * v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ rg2,
                                                       float* __restrict__ ru2, size_t n, const float* __restrict__ g2, float clip_c,
                                                       float two_decay) {
    const float n2 = g2[0];
    const float scale = (clip_c > 0.f && n2 > clip_c * clip_c) ? clip_c / sqrtf(n2) : 1.f;   // :1194-1203
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = (g[i] + two_decay * p[i]) * scale;               // :1130-1136 (L2 term), then the clip
        const float r = 0.95f * rg2[i] + 0.05f * gi * gi;                 // common.py:184
        const float ud = -sqrtf(ru2[i] + 1e-6f) / sqrtf(r + 1e-6f) * gi;  // :189
        rg2[i] = r;
        ru2[i] = 0.95f * ru2[i] + 0.05f * ud * ud;                        // :190
        p[i] += ud;                                                        // :191
    }
}

// gradient wrt the initial state: h0 = tanh(mean.Ws + bs), c0 = tanh(mean.Wm + bm) (:657-660)
__global__ __launch_bounds__(256) void state0_bwd_kernel(const float* __restrict__ dh_pass, const float* __restrict__ dhU, int nU,
                                                         const float* __restrict__ dhW, int nW, const float* __restrict__ dselpre,
                                                         const float* __restrict__ W_sel, const float* __restrict__ dc,
                                                         const float* __restrict__ h0, const float* __restrict__ c0,
                                                         float* __restrict__ dph0, float* __restrict__ dpc0, int M, int D) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * D) return;
    const int b = (int)(idx / D), d = (int)(idx % D);
    const size_t MD = (size_t)M * D;
    float dh = dh_pass[idx];
    for (int p = 0; p < nU; ++p) dh += dhU[(size_t)p * MD + idx];
    for (int p = 0; p < nW; ++p) dh += dhW[(size_t)p * MD + idx];
    if (W_sel) dh += dselpre[b] * W_sel[d];
    const float h = h0[idx], c = c0[idx];
    dph0[idx] = dh * (1.f - h * h);
    dpc0[idx] = dc[idx] * (1.f - c * c);
}

inline int grid_for(size_t n, int block, int cap = 4096) {
    size_t g = (n + block - 1) / block;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

// ================================ launchers ===================================================
hipError_t launch_dlogit(hipStream_t s, const float* probs, int ldp, const int64_t* x, const float* mask, float nll_scale,
                         float* dl, int ldd, int rows, int V, int Vp) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(dlogit_kernel, dim3(rows), dim3(256), 0, s, probs, ldp, x, mask, nll_scale, dl, ldd, V, Vp);
    return hipGetLastError();
}
hipError_t launch_alpha_reg(hipStream_t s, const AlphaRegArgs& a, int steps) {
    size_t nmax = 0;
    for (int w = 0; w < a.count; ++w) nmax = a.n[w] > nmax ? a.n[w] : nmax;
    if (!nmax) return hipSuccess;
    hipLaunchKernelGGL(alpha_reg_kernel, dim3((unsigned)((nmax + 255) / 256), a.count), dim3(256), 0, s, a, steps);
    return hipGetLastError();
}
hipError_t launch_lstm_bwd(hipStream_t s, const LstmBwdArgs& a) {
    // small batches: one wave per workgroup so that M * D / 4 lanes spread over all CUs (64 x 1024: 256 workgroups)
    const size_t n = (size_t)a.M * (a.D / 4);
    const unsigned bs = n >= (size_t)256 * 1024 ? 256u : 64u;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_spatial_bwd(hipStream_t s, const SpatialBwdArgs& a) {
    if (a.K > KMAX) return hipErrorInvalidValue;
    if (a.rider.nblocks && !rider_shape_ok(a.rider)) return hipErrorInvalidValue;
    if (!a.cparts || !a.csum || !a.dctxP || !a.dcsum || !a.dselpre) return hipErrorInvalidValue;
    const dim3 grid(a.M * a.T + a.rider.nblocks);
    if (a.bf16) {
        if (a.D % 4) return hipErrorInvalidValue;
        static const char* no8 = sw_tool("STATTN_BWD_BF16_4COL");          // A/B switch for tools: the 4-column form for every shape
#if STATTN_EXPERIMENTAL
        if (!no8 && exp_launch_spatial_bwd_bf16v2(s, a, grid)) return hipGetLastError();
#endif
        if (a.K <= 16 && a.D % 8 == 0 && !no8) {     // eight columns per lane, 16-byte slab loads, 128 threads
            if (a.K > 8) hipLaunchKernelGGL(spatial_bwd_bf16_kernel<16>, grid, dim3(128), 0, s, a);
            else hipLaunchKernelGGL(spatial_bwd_bf16_kernel<8>, grid, dim3(128), 0, s, a);
        } else if (a.K > 8 && a.K <= 16) hipLaunchKernelGGL((spatial_bwd_kernel<16, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((spatial_bwd_kernel<8, true>), grid, dim3(256), 0, s, a);
    } else if (a.K > 8 && a.K <= 16) hipLaunchKernelGGL((spatial_bwd_kernel<16, false>), grid, dim3(256), 0, s, a);
    else {
#if STATTN_EXPERIMENTAL
        if (exp_launch_spatial_bwd2(s, a, grid)) return hipGetLastError();
#endif
        hipLaunchKernelGGL((spatial_bwd_kernel<8, false>), grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_reduce_T(hipStream_t s, const float* dslp, const float* dsgp, const float* dsmp, const float* dplt,
                           float* dsproj, int lddsp, int M, int T, int D, float* dsproj_pk) {
    const int nd4 = D >> 2, bx = nd4 < 256 ? ((nd4 + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(reduce_T_kernel, dim3(M, 4, (nd4 + 255) / 256), dim3(bx), 0, s, dslp, dsgp, dsmp, dplt, dsproj, lddsp, T, D, dsproj_pk);
    return hipGetLastError();
}
int ctxgrad_groups(int K) { return (K + 3) / 4; }
hipError_t launch_ctxgrad(hipStream_t s, const CtxGradArgs& a) {
    const int NG = ctxgrad_groups(a.K);
    const int items8 = (a.M * a.T + 7) / 8;
    if (a.bf16) {
        if (a.K <= 8) hipLaunchKernelGGL((ctxgrad_kernel<8, true>), dim3(items8 * 8 * (NG + 1)), dim3(256), 0, s, a, NG);
        else hipLaunchKernelGGL((ctxgrad_kernel<16, true>), dim3(items8 * 8 * (NG + 2)), dim3(256), 0, s, a, NG);
    } else if (a.K <= 8) hipLaunchKernelGGL((ctxgrad_kernel<8, false>), dim3(items8 * 8 * (NG + 1)), dim3(256), 0, s, a, NG);
    else hipLaunchKernelGGL((ctxgrad_kernel<16, false>), dim3(items8 * 8 * (NG + 2)), dim3(256), 0, s, a, NG);
    return hipGetLastError();
}
// dst[n] (+)= sum_r X[r, n]; `part` must hold colsum_parts(rows, N) * N floats
int colsum_parts(int rows, int N) {
    int rs = 1024 / ((N + 63) / 64);
    if (rs > rows / 16) rs = rows / 16;
    if (rs < 1) rs = 1;
    if (rs > 256) rs = 256;
    return rs;
}
hipError_t launch_colsum(hipStream_t s, const float* X, int ldx, int rows, int N, float* part, float* dst, int accumulate,
                         const float* row_weights) {
    if (rows <= 0 || N <= 0) return hipSuccess;
    const int rs = colsum_parts(rows, N);
    hipLaunchKernelGGL(colsum_part_kernel, dim3((N + 63) / 64, rs), dim3(256), 0, s, X, ldx, rows, N, part, rs, row_weights);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, rs, N, dst, accumulate);
    return hipGetLastError();
}
bool colsum_batch_add(ColsumBatch& b, const float* X, int ldx, int rows, int N, float* dst, int accumulate, const float* rw) {
    if (rows <= 0 || N <= 0) return true;
    if (b.n >= 24 || N % 4 != 0 || ldx % 4 != 0) return false;
    ColsumJob& j = b.j[b.n];
    const int ncb = (N + 255) / 256;
    int rs = rows / 32;                       // >= 32 rows per slice (8 per wave)
    if (rs > 512 / ncb) rs = 512 / ncb;
    if (rs < 1) rs = 1;
    j.X = X; j.rw = rw; j.dst = dst; j.ldx = ldx; j.rows = rows; j.N = N; j.rs = rs; j.accumulate = accumulate;
    j.blk0 = b.nblk; j.fblk0 = b.nfblk;
    j.part0 = b.n ? b.j[b.n - 1].part0 + b.j[b.n - 1].rs * b.j[b.n - 1].N : 0;
    if ((size_t)j.part0 + (size_t)rs * N > COLSUM_BATCH_PART_FLOATS) return false;
    b.nblk += ncb * rs; b.nfblk += ncb;
    ++b.n;
    return true;
}
hipError_t launch_colsum_batch(hipStream_t s, const ColsumBatch& b, float* part) {
    if (b.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(colsum_batch_part_kernel, dim3(b.nblk), dim3(256), 0, s, b, part);
    hipLaunchKernelGGL(colsum_batch_final_kernel, dim3(b.nfblk), dim3(256), 0, s, b, part);
    return hipGetLastError();
}
hipError_t launch_multi_sum(hipStream_t s, const MultiSumArgs& a, float* part /* >= 12 * 32 floats */) {
    if (a.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(multi_sum_part_kernel, dim3(MS_BLOCKS, a.count), dim3(256), 0, s, a, part);
    hipLaunchKernelGGL(multi_sum_final_kernel, dim3(1), dim3(64), 0, s, a, part);
    return hipGetLastError();
}
hipError_t launch_sum_all(hipStream_t s, const float* x, size_t n, float* dst, float scale, int accumulate) {
    hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(1024), 0, s, x, n, dst, scale, accumulate);
    return hipGetLastError();
}
hipError_t launch_tanh_bwd(hipStream_t s, const float* dy, const float* t, const float* mul, float* out, size_t n, int t_bf16) {
    if (!n) return hipSuccess;
    if (t_bf16) hipLaunchKernelGGL(tanh_bwd_kernel<true>, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, dy, t, mul, out, n / 4);
    else hipLaunchKernelGGL(tanh_bwd_kernel<false>, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, dy, t, mul, out, n / 4);
    return hipGetLastError();
}
hipError_t launch_add(hipStream_t s, const float* a, const float* b, float* out, size_t n, int accumulate) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, a, b, out, n / 4, accumulate);
    return hipGetLastError();
}
hipError_t launch_embed_bwd(hipStream_t s, const EmbedPlan& pl, const float* demb, float* dWemb, float* part, int E, int shift) {
    if (pl.npieces <= 0) return hipSuccess;
    if (E % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_bwd_piece_kernel, dim3(pl.npieces), dim3(128), 0, s, pl, demb, dWemb, part, E, shift);
    if (pl.nmulti > 0) hipLaunchKernelGGL(embed_bwd_word_kernel, dim3(pl.nmulti), dim3(128), 0, s, pl, part, dWemb, E);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void unmul_kernel(const float* __restrict__ a, const float* __restrict__ mul, float* __restrict__ t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float m = mul[i];
        t[i] = m != 0.f ? a[i] / m : 0.f;
    }
}
hipError_t launch_unmul(hipStream_t s, const float* a, const float* mul, float* t, size_t n) {
    hipLaunchKernelGGL(unmul_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, a, mul, t, n);
    return hipGetLastError();
}
hipError_t launch_transpose(hipStream_t s, const float* in, int ldi, float* out, int ldo, int rows, int cols) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, in, ldi, out, ldo, rows, cols);
    return hipGetLastError();
}
hipError_t launch_state0_bwd(hipStream_t s, const float* dh_pass, const float* dhU, int nU, const float* dhW, int nW,
                             const float* dselpre, const float* W_sel, const float* dc, const float* h0, const float* c0,
                             float* dph0, float* dpc0, int M, int D) {
    hipLaunchKernelGGL(state0_bwd_kernel, dim3((unsigned)(((size_t)M * D + 255) / 256)), dim3(256), 0, s, dh_pass, dhU, nU, dhW, nW,
                       dselpre, W_sel, dc, h0, c0, dph0, dpc0, M, D);
    return hipGetLastError();
}
hipError_t launch_decay_sumsq(hipStream_t s, const float* g, const float* p, float two_decay, size_t n, float* part, int nblocks) {
    hipLaunchKernelGGL(decay_sumsq_kernel, dim3(nblocks), dim3(256), 0, s, g, p, two_decay, n, part);
    return hipGetLastError();
}
hipError_t launch_adadelta(hipStream_t s, float* p, const float* g, float* rg2, float* ru2, size_t n, const float* g2, float clip_c,
                           float two_decay) {
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, p, g, rg2, ru2, n, g2, clip_c, two_decay);
    return hipGetLastError();
}

}  // namespace stattn
