// The second stage of MultiScale_TemporalConv as ONE launch per direction (reference models/ctrgcn.py:52-69 TemporalConv,
// :101-119 the branches, :137-147 forward): every dilated k x 1 branch convolution -- and, in the forward, the pooled
// branch (MaxPool2d((3,1), stride, pad 1), :117) -- reads its slice of the entry convs' pre-BatchNorm output through the
// BatchNorm + ReLU prologue and writes its slice of the concatenated pre-BatchNorm output with that slice's BatchNorm
// moments.  The data gradient is the same kernel on the transposed / flipped weights (zero-upsampled source for a strided
// forward), with the ReLU mask and the entry BatchNorm's backward moments in the epilogue.
//
// Per branch this is a gather-GEMM  Y[m][p] = sum_k sum_tap W[m][k][tap] * X[k][p + tap*dil*V],  p = (t, v), K = M = Cb
// (16 / 32 / 64 channels): small, so a launch of the generic kernel (conv.hip) was a chain of latencies on a few hundred
// workgroups (profiles/r03_roofline_table_nucla.txt: 16-31 % of the roofs).  Here
//   * all branches (x output-channel halves of 32 for Cb = 64) and the pool share one grid: 2-5x the workgroups, ~48 KB
//     of LDS each -> three resident per CU, whose load / MFMA / store phases overlap;
//   * the product is OPERAND-SWAPPED: activation columns are the MFMA rows, output channels the MFMA columns, so a lane's
//     accumulator registers are FOUR CONSECUTIVE columns of one channel: 16-byte stores and BatchNorm moments straight
//     from registers -- no LDS-staged epilogue, none of its barriers;
//   * the line buffer [16 channels][frames + halo][Vp] is filled once per 16-channel chunk (register-staged: the
//     prologue, the temporal zero padding and the zero-upsampling are applied on the way in; each element then feeds
//     KT taps x Cb/16 row tiles of MFMAs), the chunk's weights [tap][k][m] beside it; the next chunk is prefetched into
//     registers under the MFMAs.
// v_mfma_f32_16x16x4_f32 throughout: exact fp32 (the reference's arithmetic).
#include "common.h"

namespace {

constexpr int TC_NT = 256;
constexpr int TC_BK = 16;                                 // input channels per LDS chunk
constexpr int TC_NPF = 8;                                 // float4 prefetch slots per thread and source
constexpr int TC_MAXLB = TC_NPF * TC_NT * 4 / TC_BK;      // floats per line-buffer row: 512
constexpr int TC_MAXB = TAMGCN_TCONV_MAXB;

struct TcArgs {
    SrcDev src;                     // (N, src.ctot, T_src, V); branch b reads Cb channels from src.coff + b*Cb
    int N, T_src, V, Cb, nb, stride, up, pool;
    int dil[TC_MAXB], pad[TC_MAXB];
    const float* w[TC_MAXB];
    const float* bias[TC_MAXB];
    long long ws_m, ws_k, ws_t, w_off;      // weight element strides: W[out channel][contraction channel][tap]
    float* y; int yctot, ycoff, T_out;      // (N, yctot, T_out, V); branch b writes Cb channels at ycoff + b*Cb
    float* stats; int stats_ctot, nparts;   // [2][stats_ctot][nparts] at the output channel
    SrcDev mask; const float* center;       // backward: y *= (mask value > 0); second moment against mask.x1 - center[ch]
    int BT, TIN, LB, pitchX, pitchW, Vs, Vp, nsl, mh;
};

// four consecutive columns col0..col0+3 of one output row: contiguous in HBM (full-width tiles: the row is the flat
// (t, v) run; joint slices: Vs % 4 == 0 keeps the group inside its frame).  Only a full-width tile can end in a partial group.
__device__ __forceinline__ long long tc_off(int V, int Vs, int v0, int t0, int col0, bool flat) {
    if (flat) return (long long)t0 * V + col0;
    const int fr = col0 / Vs;
    return (long long)(t0 + fr) * V + v0 + (col0 - fr * Vs);
}

template <int MT, int CT, int KT, bool BWD>
__global__ __launch_bounds__(TC_NT, BWD ? 2 : 3) void tconv_kernel(const TcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tc_smem[];
    float* Xs = tc_smem;                                  // [16][pitchX]
    float* Ws = Xs + TC_BK * a.pitchX;                    // [KT*16][pitchW]
    float* Ss = Ws + KT * TC_BK * a.pitchW;               // [2][4][MT*16]
    float* cf = Ss + 2 * 4 * MT * 16;                     // [3][Cb]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int tti = blockIdx.x / a.nsl, sl = blockIdx.x - tti * a.nsl;
    const int n = blockIdx.z;
    const int V = a.V, Vs = a.Vs, Vp = a.Vp, v0 = sl * Vs;
    const bool flat = a.nsl == 1;
    const int t0 = tti * a.BT;
    const int bt = min(a.BT, a.T_out - t0);
    const int ncols = bt * Vs;
    const long long cs = (long long)a.T_src * V;          // channel stride of the source
    const long long ocs = (long long)a.T_out * V;         // ... of the output

    if (!BWD && (int)blockIdx.y >= a.nb * a.mh) {
        // ---- pooled branch: max over frames th-1, th, th+1 of the activated source.  The source went through a ReLU
        // (host-checked), so the zeros of the temporal padding never win against the window's always-valid centre: the
        // same value as aten's -inf padding.
        const int br = a.nb;
        const int tin0 = t0 * a.stride - 1;
        const int TINp = (bt - 1) * a.stride + 3;
        const int LB4 = (TINp * Vp) >> 2;
        const long long sbase = ((long long)n * a.src.ctot + a.src.coff + br * a.Cb) * cs;
        for (int kc = 0; kc < a.Cb; kc += TC_BK) {
            __syncthreads();
            for (int e = tid; e < TC_BK * LB4; e += TC_NT) {
                const int kk = e / LB4, pos = (e - kk * LB4) << 2;
                const int slot = pos / Vp, v = pos - slot * Vp;
                const int th = tin0 + slot;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (th >= 0 && th < a.T_src) {
                    const int ch = a.src.coff + br * a.Cb + kc + kk;
                    const float c1 = a.src.coef ? a.src.coef[ch] : 1.f, c0 = a.src.coef ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
                    const float4 x = *reinterpret_cast<const float4*>(a.src.x1 + sbase + (long long)(kc + kk) * cs + (long long)th * V + v0 + v);
                    o.x = fmaxf(fmaf(c1, x.x, c0), 0.f); o.y = fmaxf(fmaf(c1, x.y, c0), 0.f);
                    o.z = fmaxf(fmaf(c1, x.z, c0), 0.f); o.w = fmaxf(fmaf(c1, x.w, c0), 0.f);
                }
                *reinterpret_cast<float4*>(Xs + kk * a.pitchX + pos) = o;
            }
            __syncthreads();
            const int kk = tid >> 4, g = tid & 15;
            const int och = a.ycoff + br * a.Cb + kc + kk;
            float* yrow = a.y + ((long long)n * a.yctot + och) * ocs;
            const float* xr = Xs + kk * a.pitchX;
            float s1 = 0.f, s2 = 0.f;
            for (int c = 4 * g; c < ncols; c += 64) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = c + r;
                    if (cc < ncols) {
                        const int fr = cc / Vs, v = cc - fr * Vs;
                        const float* p = xr + fr * a.stride * Vp + v;
                        const float m = fmaxf(fmaxf(p[0], p[Vp]), p[2 * Vp]);
                        o[r] = m;
                        s1 += m;
                        s2 = fmaf(m, m, s2);
                    }
                }
                const long long off = tc_off(V, Vs, v0, t0, c, flat);
                if (c + 3 < ncols) *reinterpret_cast<f32x4*>(yrow + off) = o;
                else for (int r = 0; r < 4 && c + r < ncols; ++r) yrow[off + r] = o[r];
            }
            if (a.stats) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (g == 0) {
                    const int part = n * gridDim.x + blockIdx.x;
                    a.stats[((long long)0 * a.stats_ctot + och) * a.nparts + part] = s1;
                    a.stats[((long long)1 * a.stats_ctot + och) * a.nparts + part] = s2;
                }
            }
        }
        return;
    }

    const int br = blockIdx.y / a.mh;
    const int m0 = (blockIdx.y - br * a.mh) * (MT * 16);       // first output channel (inside the branch) of this workgroup
    const int dil = a.dil[br];
    const int tin0 = t0 * a.stride - a.pad[br];
    const float* __restrict__ wb = a.w[br];
    const int sch0 = a.src.coff + br * a.Cb;                   // first source channel of the branch

    for (int e = tid; e < a.Cb; e += TC_NT) {
        const int ch = sch0 + e;
        cf[e] = a.src.coef ? a.src.coef[ch] : 1.f;
        cf[a.Cb + e] = (a.src.coef && a.src.x2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cf[2 * a.Cb + e] = a.src.coef ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }

    // LDS offsets of this lane's A-fragment columns: column c = (frame fr, joint v) -> fr*stride*Vp + v (tap adds tap*dil*Vp)
    int boff[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = (wave * CT + c) * 16 + j;
        if (col < ncols) { const int fr = col / Vs; boff[c] = fr * a.stride * Vp + (col - fr * Vs); }
        else boff[c] = 0;                                      // padding tile: reads in-bounds data, never stored
    }
    f32x4 acc[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Line-buffer prefetch: thread (kk = tid / 16, sub = tid % 16) owns the float4 pieces sub + 16*i of channel row kk -- one
    // channel per thread, so the prologue coefficients are three registers per chunk, the LDS address is affine in i, and
    // 16 consecutive lanes fetch 256 contiguous bytes.
    const int LB4 = a.LB >> 2;
    const int pk = tid >> 4, psub = tid & 15;
    const long long sbase = ((long long)n * a.src.ctot + sch0) * cs + (long long)pk * cs;
    int p_off[TC_NPF];
    unsigned okm = 0, inm = 0;                                 // okm: piece inside the source; inm: piece inside the line buffer
#pragma unroll
    for (int i = 0; i < TC_NPF; ++i) {
        const int c4 = psub + 16 * i;
        const int pos = c4 << 2;
        const int slot = pos / Vp, v = pos - slot * Vp;
        int th = tin0 + slot;
        bool ok = c4 < LB4 && th >= 0;
        if (a.up > 1) { ok = ok && (th % a.up == 0); th /= a.up; }
        ok = ok && th < a.T_src;
        if (ok) okm |= 1u << i;
        if (c4 < LB4) inm |= 1u << i;
        p_off[i] = th * V + v0 + v;
    }
    float* const xw = Xs + pk * a.pitchX + (psub << 2);       // piece i goes to xw + 64*i
    // Weight prefetch: thread (r = tid / 16, q0 = tid % 16) owns the elements q0 + 16*i of its row(s) of the chunk, in MEMORY
    // order: forward W[m][k][tap] -> row = output channel (one row of 16*KT contiguous floats per 16-row tile), backward
    // -> row = contraction channel (MT*16*KT contiguous floats: the output channels and their taps, taps flipped).
    constexpr int NW = KT * MT;
    const int wrow = tid >> 4, wq0 = tid & 15;
    const float* const wthr = BWD ? wb + (long long)wrow * a.ws_k + (long long)m0 * KT + wq0
                                  : wb + (long long)(m0 + wrow) * a.ws_m + wq0;
    float4 r1[TC_NPF], r2[BWD ? TC_NPF : 1];
    float wr[NW];
    const bool has2 = BWD && a.src.x2 != nullptr;
    auto prefetch = [&](int kc) {
        const float* x1 = a.src.x1 + sbase + (long long)kc * cs;
#pragma unroll
        for (int i = 0; i < TC_NPF; ++i) {
            r1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (okm & (1u << i)) r1[i] = *reinterpret_cast<const float4*>(x1 + p_off[i]);
        }
        if (BWD) {
            const float* x2 = a.src.x2 + sbase + (long long)kc * cs;
#pragma unroll
            for (int i = 0; i < TC_NPF; ++i) {
                r2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has2 && (okm & (1u << i))) r2[i] = *reinterpret_cast<const float4*>(x2 + p_off[i]);
            }
        }
        const float* wk = wthr + (long long)kc * a.ws_k;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (!BWD) wr[i] = wk[(long long)(i / KT) * 16 * a.ws_m + 16 * (i % KT)];
            else wr[i] = wk[16 * i];
        }
    };
    prefetch(0);
    __syncthreads();                                           // cf table visible

    for (int kc = 0; kc < a.Cb; kc += TC_BK) {
        if (kc) __syncthreads();                               // previous chunk's MFMAs are done with Xs / Ws
        {
            const float c1 = cf[kc + pk], c2 = cf[a.Cb + kc + pk], c0 = cf[2 * a.Cb + kc + pk];
#pragma unroll
            for (int i = 0; i < TC_NPF; ++i) {
                if (inm & (1u << i)) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (okm & (1u << i)) {
                        if (BWD) {
                            o.x = fmaf(c1, r1[i].x, fmaf(c2, r2[i].x, c0)); o.y = fmaf(c1, r1[i].y, fmaf(c2, r2[i].y, c0));
                            o.z = fmaf(c1, r1[i].z, fmaf(c2, r2[i].z, c0)); o.w = fmaf(c1, r1[i].w, fmaf(c2, r2[i].w, c0));
                        } else {
                            o.x = fmaxf(fmaf(c1, r1[i].x, c0), 0.f); o.y = fmaxf(fmaf(c1, r1[i].y, c0), 0.f);
                            o.z = fmaxf(fmaf(c1, r1[i].z, c0), 0.f); o.w = fmaxf(fmaf(c1, r1[i].w, c0), 0.f);
                        }
                    }
                    *reinterpret_cast<float4*>(xw + 64 * i) = o;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            // LDS image Ws[tap][k][m]
            if (!BWD) {
                const int q = wq0 + 16 * (i % KT);              // (k, tap) in memory order
                const int kk = q / KT, tap = q - kk * KT;
                Ws[(tap * TC_BK + kk) * a.pitchW + (i / KT) * 16 + wrow] = wr[i];
            } else {
                const int q = wq0 + 16 * i;                     // (m, flipped tap) in memory order
                const int mi = q / KT, tap = KT - 1 - (q - mi * KT);
                Ws[(tap * TC_BK + wrow) * a.pitchW + mi] = wr[i];
            }
        }
        __syncthreads();
        if (kc + TC_BK < a.Cb) prefetch(kc + TC_BK);           // in flight under the MFMAs below
#pragma unroll 1
        for (int tap = 0; tap < KT; ++tap) {                   // not unrolled: one tap's 4*(CT+MT) LDS addresses live at a time
            const float* xt = Xs + kq * a.pitchX + tap * dil * Vp;
            const float* wt = Ws + (tap * TC_BK + kq) * a.pitchW + j;
#pragma unroll
            for (int k4 = 0; k4 < TC_BK / 4; ++k4) {
                float av[CT], bv[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) bv[mt] = wt[k4 * 4 * a.pitchW + mt * 16];
#pragma unroll
                for (int c = 0; c < CT; ++c) av[c] = xt[k4 * 4 * a.pitchX + boff[c]];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][c] = mfma16(av[c], bv[mt], acc[mt][c]);
            }
        }
    }

    // ---- epilogue straight from the accumulators: lane (j, kq) holds, per (row tile mt, column tile c), the four
    // consecutive columns (wave*CT + c)*16 + 4*kq + r of output channel m0 + mt*16 + j.
    float s1[MT], s2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mch = br * a.Cb + m0 + mt * 16 + j;          // channel inside the launch's output / mask slice
        const int och = a.ycoff + mch;
        float* yrow = a.y + ((long long)n * a.yctot + och) * ocs;
        const float bia = (!BWD && a.bias[br]) ? a.bias[br][m0 + mt * 16 + j] : 0.f;
        float q1 = 0.f, q2 = 0.f;
        if (!BWD) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int col0 = (wave * CT + c) * 16 + 4 * kq;
                if (col0 < ncols) {
                    f32x4 v = acc[mt][c];
                    v[0] += bia; v[1] += bia; v[2] += bia; v[3] += bia;
                    const long long off = tc_off(V, Vs, v0, t0, col0, flat);
                    if (col0 + 3 < ncols) {
                        *reinterpret_cast<f32x4*>(yrow + off) = v;
                        q1 += (v[0] + v[1]) + (v[2] + v[3]);
                        q2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], q2))));
                    } else {
                        for (int r = 0; r < 4 && col0 + r < ncols; ++r) { yrow[off + r] = v[r]; q1 += v[r]; q2 = fmaf(v[r], v[r], q2); }
                    }
                }
            }
        } else {
            // d h_pre = (conv value) where relu(bn(h_pre)) > 0, else 0; moments (sum d, sum d * (h_pre - mean)) for the entry
            // BatchNorm's backward.  All loads first, then the stores (vmcnt counts both, in order).
            const int hch = a.mask.coff + mch;
            const float* hrow = a.mask.x1 + ((long long)n * a.mask.ctot + hch) * ocs;
            const float mc1 = a.mask.coef ? a.mask.coef[hch] : 1.f, mc0 = a.mask.coef ? a.mask.coef[2 * a.mask.ctot + hch] : 0.f;
            const float ctr = a.center ? a.center[hch] : 0.f;
            f32x4 h[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int col0 = (wave * CT + c) * 16 + 4 * kq;
                h[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (col0 < ncols) {
                    const long long off = tc_off(V, Vs, v0, t0, col0, flat);
                    if (col0 + 3 < ncols) h[c] = *reinterpret_cast<const f32x4*>(hrow + off);
                    else for (int r = 0; r < 4 && col0 + r < ncols; ++r) h[c][r] = hrow[off + r];
                }
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int col0 = (wave * CT + c) * 16 + 4 * kq;
                if (col0 < ncols) {
                    f32x4 v = acc[mt][c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!(fmaf(mc1, h[c][r], mc0) > 0.f)) v[r] = 0.f;
                        if (col0 + r < ncols) { q1 += v[r]; q2 = fmaf(v[r], h[c][r] - ctr, q2); }
                    }
                    const long long off = tc_off(V, Vs, v0, t0, col0, flat);
                    if (col0 + 3 < ncols) *reinterpret_cast<f32x4*>(yrow + off) = v;
                    else for (int r = 0; r < 4 && col0 + r < ncols; ++r) yrow[off + r] = v[r];
                }
            }
        }
        s1[mt] = q1; s2[mt] = q2;
    }
    if (a.stats) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float u1 = s1[mt], u2 = s2[mt];
            u1 += __shfl_xor(u1, 16); u1 += __shfl_xor(u1, 32);
            u2 += __shfl_xor(u2, 16); u2 += __shfl_xor(u2, 32);
            if (kq == 0) {
                Ss[(0 * 4 + wave) * (MT * 16) + mt * 16 + j] = u1;
                Ss[(1 * 4 + wave) * (MT * 16) + mt * 16 + j] = u2;
            }
        }
        __syncthreads();
        if (tid < 2 * MT * 16) {
            const int st = tid / (MT * 16), row = tid - st * (MT * 16);
            const float tot = (Ss[(st * 4 + 0) * (MT * 16) + row] + Ss[(st * 4 + 1) * (MT * 16) + row]) +
                              (Ss[(st * 4 + 2) * (MT * 16) + row] + Ss[(st * 4 + 3) * (MT * 16) + row]);
            const int part = n * gridDim.x + blockIdx.x;
            a.stats[((long long)st * a.stats_ctot + a.ycoff + br * a.Cb + m0 + row) * a.nparts + part] = tot;
        }
    }
}

struct TcPlan { int BT, TIN, LB, pitchX, pitchW, Vs, Vp, nsl, ntt, mt, mh, ct; size_t lds; };

// stride: frame step of the product's source per output frame (the forward's stride; 1 for the data gradient, whose source is
// zero-upsampled); span = (KT-1) * largest dilation
static int tc_plan(int V, int Cb, int KT, int span, int stride, int T_out, TcPlan* p) {
    if (V < 1 || Cb < 16 || (Cb != 16 && Cb % 32 != 0) || (KT != 3 && KT != 5) || T_out < 1) return -1;
    p->Vs = V; p->nsl = 1;
    if (V > 32) { if (V % 16) return -1; p->Vs = 16; p->nsl = V / 16; }
    p->Vp = (p->Vs + 3) & ~3;
    int BT = 320 / p->Vs;
    if (BT < 1) return -1;
    if (BT > T_out) BT = T_out;
    for (;; --BT) {
        if (BT < 1) return -1;
        p->TIN = (BT - 1) * stride + span + 1;
        p->LB = p->TIN * p->Vp;
        if (p->LB <= TC_MAXLB) break;
    }
    p->BT = BT;
    p->pitchX = p->LB + (((16 - (p->LB & 31)) + 32) & 31);      // == 16 (mod 32): the two k rows of a 32-lane half sit 16 banks apart
    p->mt = Cb == 16 ? 1 : 2;
    p->mh = Cb == 16 ? 1 : Cb / 32;
    p->pitchW = p->mt == 1 ? 16 : 48;                         // == 16 (mod 32)
    const int tiles = ceil_div(BT * p->Vs, 16);
    p->ct = ceil_div(tiles, 4) <= 3 ? 3 : 5;
    if (ceil_div(tiles, 4) > 5) return -1;
    p->ntt = ceil_div(T_out, BT);
    p->lds = sizeof(float) * ((size_t)TC_BK * p->pitchX + (size_t)KT * TC_BK * p->pitchW + 2 * 4 * p->mt * 16 + 3 * (size_t)Cb);
    return 0;
}

template <bool BWD>
static int tc_launch(const TcArgs& a, const TcPlan& p, int KT, dim3 grid, hipStream_t s) {
#define TC_CASE(MT_, CT_, KT_)                                                                                          \
    if (p.mt == MT_ && p.ct == CT_ && KT == KT_) {                                                                      \
        hipLaunchKernelGGL((tconv_kernel<MT_, CT_, KT_, BWD>), grid, dim3(TC_NT), p.lds, s, a);                         \
        tamgcn_note_kernel("tconv_kernel<%d, %d, %d, %s>", MT_, CT_, KT_, BWD ? "bwd" : "fwd");                         \
        return 0;                                                                                                       \
    }
    TC_CASE(1, 5, 5) TC_CASE(2, 5, 5) TC_CASE(1, 3, 5) TC_CASE(2, 3, 5)
    TC_CASE(1, 5, 3) TC_CASE(2, 5, 3) TC_CASE(1, 3, 3) TC_CASE(2, 3, 3)
#undef TC_CASE
    tamgcn_set_error("tamgcn_tconv: no instantiation mt=%d ct=%d KT=%d", p.mt, p.ct, KT);
    return -1;
}

static int tc_span(const int* dil, int nb, int KT) {
    int d = 1;
    for (int b = 0; b < nb; ++b) d = dil[b] > d ? dil[b] : d;
    return (KT - 1) * d;
}

}  // namespace

extern "C" int tamgcn_tconv_supported(int V, int Cb, int KT, int nb, const int* dil, int stride, int T_in) {
    if (nb < 1 || nb > TC_MAXB || !dil || stride < 1 || T_in < 1) return 0;
    for (int b = 0; b < nb; ++b) if (dil[b] < 1 || ((KT - 1) * dil[b]) % 2) return 0;
    TcPlan p;
    const int T_out = (T_in - 1) / stride + 1;
    return tc_plan(V, Cb, KT, tc_span(dil, nb, KT), stride, T_out, &p) == 0 && tc_plan(V, Cb, KT, tc_span(dil, nb, KT), 1, T_in, &p) == 0;
}

extern "C" int tamgcn_tconv_nparts(const tamgcn_tconv_desc* d, int backward) {
    TcPlan p;
    if (!d || d->nb < 1 || d->nb > TC_MAXB) return -1;
    const int T_out = (d->T_in - 1) / d->stride + 1;
    if (tc_plan(d->V, d->Cb, d->KT, tc_span(d->dil, d->nb, d->KT), backward ? 1 : d->stride, backward ? d->T_in : T_out, &p)) return -1;
    return d->N * p.ntt * p.nsl;
}

static int tc_common_checks(const tamgcn_tconv_desc* d, const char* who) {
    TG_CHECK(d && d->src.x1 && d->y, "%s: null pointer", who);
    TG_CHECK(d->N > 0 && d->N <= 65535 && d->T_in > 0 && d->V > 0 && d->Cb > 0 && d->nb >= 1 && d->nb <= TC_MAXB && d->stride >= 1,
             "%s: bad dims N=%d T_in=%d V=%d Cb=%d nb=%d stride=%d", who, d->N, d->T_in, d->V, d->Cb, d->nb, d->stride);
    for (int b = 0; b < d->nb; ++b) {
        TG_CHECK(d->w[b], "%s: branch %d has no weights", who, b);
        TG_CHECK(d->dil[b] >= 1 && ((d->KT - 1) * d->dil[b]) % 2 == 0, "%s: branch %d: (KT-1)*dil must be even (symmetric padding)", who, b);
        TG_CHECK(((uintptr_t)d->w[b] & 3) == 0, "%s: unaligned weights", who);
    }
    TG_CHECK((long long)d->src.ctot * d->T_in * d->V < (1LL << 31) && (long long)d->yctot * d->T_in * d->V < (1LL << 31),
             "%s: a sample exceeds 2^31 elements", who);
    return 0;
}

/* forward: y[:, ycoff + b*Cb + m] = bias_b[m] + sum_{k,tap} W_b[m][k][tap] * act(src)[:, src.coff + b*Cb + k, t*stride + tap*dil_b - pad_b]
 * for b < nb, and (pool) y[:, ycoff + nb*Cb + c] = max_{-1..1} act(src)[:, src.coff + nb*Cb + c, t*stride + .] */
extern "C" int tamgcn_tconv_fwd(const tamgcn_tconv_desc* d, void* stream) {
    if (tc_common_checks(d, "tamgcn_tconv_fwd")) return -1;
    const int nbr = d->nb + (d->pool ? 1 : 0);
    TG_CHECK(d->src.coff + nbr * d->Cb <= d->src.ctot && d->ycoff + nbr * d->Cb <= d->yctot, "tamgcn_tconv_fwd: channel slice out of range");
    TG_CHECK(!d->src.x2, "tamgcn_tconv_fwd: single-source prologue only");
    TG_CHECK(!d->pool || d->src.act == 1, "tamgcn_tconv_fwd: the pooled branch needs a ReLU source (zero padding stands in for -inf)");
    TG_CHECK(d->src.act == 1, "tamgcn_tconv_fwd: the forward prologue is BatchNorm + ReLU (act = 1)");
    const int T_out = (d->T_in - 1) / d->stride + 1;
    TG_CHECK(d->T_out == T_out, "tamgcn_tconv_fwd: T_out=%d inconsistent with T_in=%d stride=%d", d->T_out, d->T_in, d->stride);
    TcPlan p;
    TG_CHECK(tc_plan(d->V, d->Cb, d->KT, tc_span(d->dil, d->nb, d->KT), d->stride, T_out, &p) == 0,
             "tamgcn_tconv_fwd: no tiling for V=%d Cb=%d KT=%d stride=%d", d->V, d->Cb, d->KT, d->stride);
    TcArgs a;
    a.src = make_src(d->src);
    a.N = d->N; a.T_src = d->T_in; a.V = d->V; a.Cb = d->Cb; a.nb = d->nb; a.stride = d->stride; a.up = 1; a.pool = d->pool;
    for (int b = 0; b < TC_MAXB; ++b) {
        a.dil[b] = b < d->nb ? d->dil[b] : 1;
        a.pad[b] = (d->KT - 1) * a.dil[b] / 2;
        a.w[b] = b < d->nb ? d->w[b] : nullptr;
        a.bias[b] = b < d->nb ? d->bias[b] : nullptr;
    }
    a.ws_m = (long long)d->Cb * d->KT; a.ws_k = d->KT; a.ws_t = 1; a.w_off = 0;
    a.y = d->y; a.yctot = d->yctot; a.ycoff = d->ycoff; a.T_out = T_out;
    a.stats = d->stats_part; a.stats_ctot = d->stats_ctot; a.nparts = d->N * p.ntt * p.nsl;
    a.mask = null_src(); a.center = nullptr;
    a.BT = p.BT; a.TIN = p.TIN; a.LB = p.LB; a.pitchX = p.pitchX; a.pitchW = p.pitchW; a.Vs = p.Vs; a.Vp = p.Vp; a.nsl = p.nsl; a.mh = p.mh;
    dim3 grid(p.ntt * p.nsl, d->nb * p.mh + (d->pool ? 1 : 0), d->N);
    if (tc_launch<false>(a, p, d->KT, grid, (hipStream_t)stream)) return -1;
    TG_LAUNCH_CHECK("tamgcn_tconv_fwd");
    return 0;
}

/* data gradient of the temporal branches: with gy = prologue value of src (N, src.ctot, T_out, V),
 *   y[:, ycoff + b*Cb + k, th] = [mask value > 0] * sum_{m,tap: th = t*stride + tap*dil_b - pad_b} W_b[m][k][tap] * gy[:, src.coff + b*Cb + m, t]
 * (y has T_in frames; mask = the forward's source with its prologue, same channel slice geometry as y) and the partial
 * moments (sum y, sum y * (mask.x1 - center[ch])) at the channels of y.  The pooled branch's backward is tamgcn_maxpool_bwd. */
extern "C" int tamgcn_tconv_bwd(const tamgcn_tconv_desc* d, void* stream) {
    if (tc_common_checks(d, "tamgcn_tconv_bwd")) return -1;
    TG_CHECK(d->src.coff + d->nb * d->Cb <= d->src.ctot && d->ycoff + d->nb * d->Cb <= d->yctot, "tamgcn_tconv_bwd: channel slice out of range");
    TG_CHECK(d->src.act == 0, "tamgcn_tconv_bwd: the gradient prologue is linear (act = 0)");
    TG_CHECK(d->mask && d->mask->x1 && !d->mask->x2, "tamgcn_tconv_bwd: needs the forward source (single-source prologue) as mask");
    TG_CHECK(d->mask->coff + d->nb * d->Cb <= d->mask->ctot, "tamgcn_tconv_bwd: mask channel slice out of range");
    const int T_out = (d->T_in - 1) / d->stride + 1;                // frames of gy
    TG_CHECK(d->T_out == T_out, "tamgcn_tconv_bwd: T_out=%d inconsistent with T_in=%d stride=%d", d->T_out, d->T_in, d->stride);
    TcPlan p;
    TG_CHECK(tc_plan(d->V, d->Cb, d->KT, tc_span(d->dil, d->nb, d->KT), 1, d->T_in, &p) == 0,
             "tamgcn_tconv_bwd: no tiling for V=%d Cb=%d KT=%d", d->V, d->Cb, d->KT);
    TcArgs a;
    a.src = make_src(d->src);
    a.N = d->N; a.T_src = T_out; a.V = d->V; a.Cb = d->Cb; a.nb = d->nb; a.stride = 1; a.up = d->stride; a.pool = 0;
    for (int b = 0; b < TC_MAXB; ++b) {
        a.dil[b] = b < d->nb ? d->dil[b] : 1;
        a.pad[b] = (d->KT - 1) * a.dil[b] - (d->KT - 1) * a.dil[b] / 2;   // flipped taps in the (zero-upsampled) gradient's frame
        a.w[b] = b < d->nb ? d->w[b] : nullptr;
        a.bias[b] = nullptr;
    }
    a.ws_m = d->KT; a.ws_k = (long long)d->Cb * d->KT; a.ws_t = -1; a.w_off = d->KT - 1;
    a.y = d->y; a.yctot = d->yctot; a.ycoff = d->ycoff; a.T_out = d->T_in;
    a.stats = d->stats_part; a.stats_ctot = d->stats_ctot; a.nparts = d->N * p.ntt * p.nsl;
    a.mask = make_src(*d->mask); a.center = d->center;
    a.BT = p.BT; a.TIN = p.TIN; a.LB = p.LB; a.pitchX = p.pitchX; a.pitchW = p.pitchW; a.Vs = p.Vs; a.Vp = p.Vp; a.nsl = p.nsl; a.mh = p.mh;
    dim3 grid(p.ntt * p.nsl, d->nb * p.mh, d->N);
    if (tc_launch<true>(a, p, d->KT, grid, (hipStream_t)stream)) return -1;
    TG_LAUNCH_CHECK("tamgcn_tconv_bwd");
    return 0;
}
