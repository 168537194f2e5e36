// Generic k x 1 convolution over (N,C,T,V) as an fp32 MFMA GEMM, its data
// gradient (same kernel, transposed/flipped weight view) and its weight
// gradient.  Replaces aten::convolution / convolution_backward on the CTR-GCN
// hot path (reference models/ctrgcn.py:56-62, 95-99, 114, 122, 161-164, 183,
// 211-213, 219-221).
//
// GEMM view per sample n:   Y[m][p] = sum_k sum_j W[m][k][j] * X[k][p + j*dil*V]
// with p = (t, v) flattened (V innermost, contiguous in HBM => coalesced along
// t*V+v).  A block owns 64 output channels x (BT output frames x V joints)
// columns; the activation tile (with its temporal halo) is staged once per
// 16-channel K chunk in LDS as a line buffer [k][frame][v]; the dilated taps
// are plain LDS offsets into that line buffer.  BatchNorm-apply(+ReLU) of the
// *input* is fused into the LDS fill, BatchNorm moment accumulation of the
// *output* into the epilogue (per-block partial sums, no atomics).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 64;        // output channels per block
constexpr int BK = 16;        // input channels per LDS chunk
constexpr int BKP = BK + 1;   // padded pitch of the weight tile (odd => conflict-free column reads)
constexpr int MAXCW = 5;      // column tiles (16 wide) per wave  -> 4 waves * 5 * 16 = 320 columns max
constexpr int MAXCOLS = 320;
constexpr int NTHREADS = 256;

TG_TRACE_DEFINE(tamgcn_trace_read)

struct ConvArgs {
    SrcDev src;
    int N, K, T_in, V;
    const float* w; const float* bias;
    int M, KT, dil, stride, pad;
    long long ws_m, ws_k, ws_t, w_off;   // weight element strides for A[i][k][tap]
    int up, wmode;
    float* y; int yctot, ycoff, T_out, T_y, ostride;
    const float* add1; const float* add2; const float* bcast; float bcast_scale;
    SrcDev mask; int has_mask;
    const float* aux; const float* aux_center; int auxctot, auxcoff;
    float* stats_part; int stats_ctot, stats_coff, nparts;
    const float* post_coef; int post_ctot, post_act;   // eval-mode output affine (folded BatchNorm) + ReLU after the residual adds
    // tiling (host chosen)
    int BT;        // output frames per block
    int CW;        // column tiles per wave
    int TIN;       // input frame slots staged per block
    int lstride;   // frame step between staged slots (stride for 1x1, else 1)
    int sB;        // slot step per output frame (1 for 1x1, else stride)
    int LB;        // TIN*Vp
    int pitchB;    // LDS pitch of one channel row (== 16 mod 32)
    // Joint geometry of the vectorised kernels.  A tile covers Vs joints per frame starting at joint v0 = slice*Vs of the
    // V in HBM; its LDS line buffer gives every frame Vp floats.  Vs = Vp = V for the usual skeletons; V % 4 != 0 (NTU's
    // 25 joints): Vp = V rounded up to 4, so that a 16-byte piece never straddles two frames of different validity (its
    // last piece over-reads <= 3 floats of the next frame: callers guarantee 16 readable bytes behind an activation);
    // large V with a temporal halo (V = 64): nsl = V/16 slices of Vs = 16 joints keep halo + tile within the line buffer.
    int Vs, Vp, nsl;
};

__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[4][MAXCW], float* Ss, int n, int m0, int t0,
                                              int ncols, int cw0, int mt_act, int c_act, const int (&tl)[MAXCW],
                                              const int (&vv)[MAXCW], int j, int kq, int wave, int tid) {
    const int V = a.V;
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[mt][r] = 0.f; s2[mt][r] = 0.f; }
    // pass 1: all loads (bias, broadcast, residual adds, mask, aux); results stay in acc
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (mt >= mt_act) continue;
#pragma unroll
        for (int c = 0; c < MAXCW; ++c) {
            if (c >= c_act) continue;
            int col = (cw0 + c) * 16 + j;
            if (col >= ncols) continue;
            int t = (t0 + tl[c]) * a.ostride;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = m0 + mt * 16 + kq * 4 + r;
                if (m >= a.M) continue;
                float val = acc[mt][c][r];
                if (a.bias) val += a.bias[m];
                if (a.post_coef) val = fmaf(a.post_coef[a.ycoff + m], val, a.post_coef[2 * a.post_ctot + a.ycoff + m]);
                long long idx = (((long long)n * a.yctot + a.ycoff + m) * a.T_y + t) * V + vv[c];
                if (a.bcast) val = fmaf(a.bcast[((long long)m * a.N + n) * V + vv[c]], a.bcast_scale, val);
                if (a.add1) val += a.add1[idx];
                if (a.add2) val += a.add2[idx];
                if (a.post_act == 1) val = fmaxf(val, 0.f);
                if (a.has_mask) {
                    int mch = a.mask.coff + m;
                    long long midx = (((long long)n * a.mask.ctot + mch) * a.T_y + t) * V + vv[c];
                    if (!(src_value(a.mask, midx, mch) > 0.f)) val = 0.f;
                }
                acc[mt][c][r] = val;
                if (a.stats_part) {
                    float x2 = val;
                    if (a.aux) x2 = a.aux[(((long long)n * a.auxctot + a.auxcoff + m) * a.T_y + t) * V + vv[c]] - a.aux_center[a.auxcoff + m];
                    s1[mt][r] += val;
                    s2[mt][r] = fmaf(val, x2, s2[mt][r]);
                }
            }
        }
    }
    // pass 2: stores only (vmcnt counts stores in order: never interleave loads with them)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (mt >= mt_act) continue;
#pragma unroll
        for (int c = 0; c < MAXCW; ++c) {
            if (c >= c_act) continue;
            int col = (cw0 + c) * 16 + j;
            if (col >= ncols) continue;
            int t = (t0 + tl[c]) * a.ostride;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = m0 + mt * 16 + kq * 4 + r;
                if (m >= a.M) continue;
                a.y[(((long long)n * a.yctot + a.ycoff + m) * a.T_y + t) * V + vv[c]] = acc[mt][c][r];
            }
        }
    }
    if (a.stats_part) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u1 = wave_sum16(s1[mt][r]);
                float u2 = wave_sum16(s2[mt][r]);
                if (j == 0) {
                    int row = mt * 16 + kq * 4 + r;
                    Ss[(0 * 4 + wave) * BM + row] = u1;
                    Ss[(1 * 4 + wave) * BM + row] = u2;
                }
            }
        __syncthreads();
        if (tid < 2 * BM) {
            int st = tid >> 6, row = tid & 63;
            int m = m0 + row;
            if (m < a.M) {
                float tot = Ss[(st * 4 + 0) * BM + row] + Ss[(st * 4 + 1) * BM + row] +
                            Ss[(st * 4 + 2) * BM + row] + Ss[(st * 4 + 3) * BM + row];
                int part = n * gridDim.x + blockIdx.x;
                a.stats_part[((long long)st * a.stats_ctot + a.stats_coff + m) * a.nparts + part] = tot;
            }
        }
    }
}

__global__ __launch_bounds__(NTHREADS) void conv_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [KT][BM][BKP]
    float* Bs = As + a.KT * BM * BKP;               // [BK][pitchB]
    float* Ss = Bs + BK * a.pitchB;                 // [2][4][BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int n = blockIdx.z, m0 = blockIdx.y * BM, t0 = blockIdx.x * a.BT;
    const int V = a.V;
    const int bt = min(a.BT, a.T_out - t0);
    const int ncols = bt * V;
    const int nct = (ncols + 15) >> 4;
    const int cw0 = wave * a.CW;
    const int mt_act = min(4, (min(BM, a.M - m0) + 15) >> 4);
    int c_act = nct - cw0; c_act = c_act < 0 ? 0 : (c_act > a.CW ? a.CW : c_act);

    int boff[MAXCW], tl[MAXCW], vv[MAXCW];
#pragma unroll
    for (int c = 0; c < MAXCW; ++c) {
        int col = (cw0 + c) * 16 + j;
        if (col < ncols) { tl[c] = col / V; vv[c] = col - tl[c] * V; boff[c] = tl[c] * a.sB * V + vv[c]; }
        else { tl[c] = 0; vv[c] = 0; boff[c] = 0; }
    }

    f32x4 acc[4][MAXCW];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int c = 0; c < MAXCW; ++c) acc[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tin0 = t0 * a.stride - a.pad;
    const long long chan_stride = (long long)a.T_in * V;
    const long long src_n = (long long)n * a.src.ctot * chan_stride;

    for (int k0 = 0; k0 < a.K; k0 += BK) {
        __syncthreads();
        // ---- weight tile A[tap][i][kk]
        const int nA = a.KT * BM * BK;
        for (int e = tid; e < nA; e += NTHREADS) {
            int kk = e & (BK - 1);
            int i = (e >> 4) & (BM - 1);
            int tap = e >> 10;
            int m = m0 + i, k = k0 + kk;
            float wv = 0.f;
            if (m < a.M && k < a.K) wv = a.w[m * a.ws_m + k * a.ws_k + tap * a.ws_t + a.w_off];
            As[(tap * BM + i) * BKP + kk] = wv;
        }
        // ---- activation line buffer B[kk][slot][v] with fused prologue, zero outside [0,T)
        for (int pos = tid; pos < a.LB; pos += NTHREADS) {
            int slot = pos / V;
            int v = pos - slot * V;
            int th = tin0 + slot * a.lstride;
            bool ok = th >= 0;
            if (a.up > 1) { ok = ok && (th % a.up == 0); th /= a.up; }
            ok = ok && th < a.T_in;
            long long goff = src_n + (long long)th * V + v;
#pragma unroll 4
            for (int kk = 0; kk < BK; ++kk) {
                int k = k0 + kk;
                float xv = 0.f;
                if (ok && k < a.K) {
                    int ch = a.src.coff + k;
                    xv = src_value(a.src, goff + ch * chan_stride, ch);
                }
                Bs[kk * a.pitchB + pos] = xv;
            }
        }
        __syncthreads();
        // ---- MFMA
        for (int tap = 0; tap < a.KT; ++tap) {
            const int tapoff = tap * a.dil * V;
#pragma unroll
            for (int k4 = 0; k4 < BK / 4; ++k4) {
                float av[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    av[mt] = As[(tap * BM + mt * 16 + j) * BKP + k4 * 4 + kq];
                const float* brow = Bs + (k4 * 4 + kq) * a.pitchB + tapoff;
#pragma unroll
                for (int c = 0; c < MAXCW; ++c) {
                    if (c < c_act) {
                        float bv = brow[boff[c]];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            if (mt < mt_act) acc[mt][c] = mfma16(av[mt], bv, acc[mt][c]);
                    }
                }
            }
        }
    }

    conv_epilogue(a, acc, Ss, n, m0, t0, ncols, cw0, mt_act, c_act, tl, vv, j, kq, wave, tid);
}

// ---------------------------------------------------------------------------
// Vectorised, software-pipelined variant (V % 4 == 0, K <= 1024): the next K chunk's
// activation rows are fetched with 16-byte global loads into registers while the MFMAs of
// the current chunk run; the BatchNorm-apply prologue is applied on the way into LDS from a
// per-channel coefficient table staged once per workgroup.
// ---------------------------------------------------------------------------
// Second half of the staged epilogue: rows r0..r0+RP-1 of the accumulator tile sit in LDS (Tt, pitch PT);
// TPR threads walk one row in float4 steps: bias / broadcast / residual adds / mask / BatchNorm moments,
// 16-byte coalesced loads and stores.  Row moments land in Ss[0..BMT) / Ss[4*BM..).
// x / d for 0 <= x < 2^20, 4 <= d <= 1024 through the reciprocal (three VALU instead of ~25): (x + 0.5) / d is at least
// 0.5 / d >= 5e-4 away from an integer, float rounding moves it by < 0.07 at these sizes
__device__ __forceinline__ int tc_like_div(int x, float rcp) { return (int)(((float)x + 0.5f) * rcp); }
template <int NTH>
__device__ __forceinline__ void staged_rows(const ConvArgs& a, const float* Tt, int PT, int RP, int r0, int BMT, int m0, int n, int t0,
                                            int ncols, float* Ss, int Vs, int v0) {
    const int tid = threadIdx.x, V = a.V;
    const int TPR = NTH / RP;
    const int row = tid / TPR, seg = tid - row * TPR;
    const int m = m0 + r0 + row;
    // a group of four columns is ONE 16-byte access when its columns are consecutive in HBM: frames contiguous (full-width
    // tile, unit output stride) or the group inside one frame (Vs % 4 == 0).  Row starts need not be 16-byte aligned
    // (V = 25: rows of T*V floats; gfx950 takes dword-aligned dwordx4 accesses at full rate, tools/probes/unaligned_probe.hip).
    const bool flat = Vs == V && a.ostride == 1;
    const bool vecok = flat || (Vs & 3) == 0;
    float p1 = 0.f, p2 = 0.f;
    const bool rowok = r0 + row < BMT && m < a.M;   // RP may overshoot the tile (BMT = 48, RP = 32)
    if (rowok) {
        const float bia = a.bias ? a.bias[m] : 0.f;
        const float ctr = a.aux ? a.aux_center[a.auxcoff + m] : 0.f;
        const float pc1 = a.post_coef ? a.post_coef[a.ycoff + m] : 1.f, pc0 = a.post_coef ? a.post_coef[2 * a.post_ctot + a.ycoff + m] : 0.f;
        const float plo = a.post_act == 1 ? 0.f : -__builtin_inff();
        const long long ybase = ((long long)n * a.yctot + a.ycoff + m) * a.T_y * V;
        const long long mbase = a.has_mask ? ((long long)n * a.mask.ctot + a.mask.coff + m) * a.T_y * V : 0;
        const long long abase = a.aux ? ((long long)n * a.auxctot + a.auxcoff + m) * a.T_y * V : 0;
        const float* bc = a.bcast ? a.bcast + ((long long)m * a.N + n) * V : nullptr;
        float mc1 = 1.f, mc2 = 0.f, mc0 = 0.f;
        if (a.has_mask && a.mask.coef) {
            int mch = a.mask.coff + m;
            mc1 = a.mask.coef[mch]; mc0 = a.mask.coef[2 * a.mask.ctot + mch];
            if (a.mask.x2) mc2 = a.mask.coef[a.mask.ctot + mch];
        }
        const int nc4 = (ncols + 3) >> 2;
        for (int c4 = seg; c4 < nc4; c4 += TPR) {
            const int col = c4 << 2;
            const int fr = col / Vs, v = col - fr * Vs;
            if ((vecok || v + 4 <= Vs) && col + 4 <= ncols) {
                const long long off = (long long)(t0 + fr) * a.ostride * V + v0 + v;
                float4 val = *reinterpret_cast<const float4*>(Tt + row * PT + col);
                val.x = fmaf(pc1, val.x + bia, pc0); val.y = fmaf(pc1, val.y + bia, pc0);
                val.z = fmaf(pc1, val.z + bia, pc0); val.w = fmaf(pc1, val.w + bia, pc0);
                if (bc) {
                    float4 b;
                    if ((V & 3) == 0) b = *reinterpret_cast<const float4*>(bc + v0 + v);
                    else { const int j0 = (v0 + v) % V; b = make_float4(bc[j0], bc[(j0 + 1) % V], bc[(j0 + 2) % V], bc[(j0 + 3) % V]); }
                    val.x = fmaf(b.x, a.bcast_scale, val.x); val.y = fmaf(b.y, a.bcast_scale, val.y);
                    val.z = fmaf(b.z, a.bcast_scale, val.z); val.w = fmaf(b.w, a.bcast_scale, val.w);
                }
                if (a.add1) { float4 t = *reinterpret_cast<const float4*>(a.add1 + ybase + off); val.x += t.x; val.y += t.y; val.z += t.z; val.w += t.w; }
                if (a.add2) { float4 t = *reinterpret_cast<const float4*>(a.add2 + ybase + off); val.x += t.x; val.y += t.y; val.z += t.z; val.w += t.w; }
                val.x = fmaxf(val.x, plo); val.y = fmaxf(val.y, plo); val.z = fmaxf(val.z, plo); val.w = fmaxf(val.w, plo);
                if (a.has_mask) {
                    float4 q = *reinterpret_cast<const float4*>(a.mask.x1 + mbase + off);
                    float4 q2 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.mask.x2) q2 = *reinterpret_cast<const float4*>(a.mask.x2 + mbase + off);
                    if (!(fmaf(mc1, q.x, fmaf(mc2, q2.x, mc0)) > 0.f)) val.x = 0.f;
                    if (!(fmaf(mc1, q.y, fmaf(mc2, q2.y, mc0)) > 0.f)) val.y = 0.f;
                    if (!(fmaf(mc1, q.z, fmaf(mc2, q2.z, mc0)) > 0.f)) val.z = 0.f;
                    if (!(fmaf(mc1, q.w, fmaf(mc2, q2.w, mc0)) > 0.f)) val.w = 0.f;
                }
                if (a.stats_part) {
                    float4 x2 = val;
                    if (a.aux) {
                        x2 = *reinterpret_cast<const float4*>(a.aux + abase + off);
                        x2.x -= ctr; x2.y -= ctr; x2.z -= ctr; x2.w -= ctr;
                    }
                    p1 += (val.x + val.y) + (val.z + val.w);
                    p2 = fmaf(val.x, x2.x, fmaf(val.y, x2.y, fmaf(val.z, x2.z, fmaf(val.w, x2.w, p2))));
                }
                *reinterpret_cast<float4*>(a.y + ybase + off) = val;
            } else {
                // the last, partial group of a tile whose column count is not a multiple of 4, or a group that leaves its
                // frame in a strided / sliced output: one column at a time
                for (int i = 0; i < 4 && col + i < ncols; ++i) {
                    const int ci = col + i, fi = ci / Vs, vi = ci - fi * Vs;
                    const long long off = (long long)(t0 + fi) * a.ostride * V + v0 + vi;
                    float val = fmaf(pc1, Tt[row * PT + ci] + bia, pc0);
                    if (bc) val = fmaf(bc[(v0 + vi) % V], a.bcast_scale, val);
                    if (a.add1) val += a.add1[ybase + off];
                    if (a.add2) val += a.add2[ybase + off];
                    val = fmaxf(val, plo);
                    if (a.has_mask) {
                        const float q = a.mask.x1[mbase + off], q2 = a.mask.x2 ? a.mask.x2[mbase + off] : 0.f;
                        if (!(fmaf(mc1, q, fmaf(mc2, q2, mc0)) > 0.f)) val = 0.f;
                    }
                    if (a.stats_part) {
                        const float x2 = a.aux ? a.aux[abase + off] - ctr : val;
                        p1 += val;
                        p2 = fmaf(val, x2, p2);
                    }
                    a.y[ybase + off] = val;
                }
            }
        }
    }
    if (a.stats_part) {
        for (int o = 1; o < TPR; o <<= 1) { p1 += __shfl_xor(p1, o); p2 += __shfl_xor(p2, o); }
        if (seg == 0 && rowok) { Ss[r0 + row] = p1; Ss[4 * BM + r0 + row] = p2; }
    }
}

template <int BKV, int MT, int CWT>
__global__ __launch_bounds__(NTHREADS) void conv_kernel_vec(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BKVP = BKV + 2;                      // pitch/2 odd: conflict-free A-fragment column reads
    constexpr int BMT = MT * 16;                       // output channels per workgroup
    constexpr int NPF = (BKV == 32) ? 10 : 8;          // float4 prefetch slots per thread (host checks the bound)
    float* As = smem;                                  // [KT][BMT][BKVP]
    float* Bs = As + ((a.KT * BMT * BKVP + 3) & ~3);   // [BKV][pitchB], 16-byte aligned
    float* Ss = Bs + BKV * a.pitchB;                   // [2][4][64]
    float* cf = Ss + 2 * 4 * BM;                       // [3][K]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int tti = blockIdx.x / a.nsl, sl = blockIdx.x - tti * a.nsl;
    const int n = blockIdx.z, m0 = blockIdx.y * BMT, t0 = tti * a.BT;
    const int V = a.V, Vs = a.Vs, Vp = a.Vp, v0 = sl * Vs;
    const int bt = min(a.BT, a.T_out - t0);
    const int ncols = bt * Vs;
    const int cw0 = wave * CWT;

    for (int e = tid; e < a.K; e += NTHREADS) {
        int ch = a.src.coff + e;
        cf[e] = a.src.coef ? a.src.coef[ch] : 1.f;
        cf[a.K + e] = (a.src.coef && a.src.x2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cf[2 * a.K + e] = a.src.coef ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }

    int boff[CWT], tl[CWT], vv[CWT];
#pragma unroll
    for (int c = 0; c < CWT; ++c) {
        int col = (cw0 + c) * 16 + j;
        if (col < ncols) { tl[c] = col / Vs; vv[c] = col - tl[c] * Vs; boff[c] = tl[c] * a.sB * Vp + vv[c]; }
        else { tl[c] = 0; vv[c] = 0; boff[c] = 0; }       // padding tile: reads in-bounds data, never stored
    }
    f32x4 acc[MT][CWT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < CWT; ++c) acc[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-thread prefetch descriptors (identical for every K chunk)
    const int tin0 = t0 * a.stride - a.pad;
    const long long chan_stride = (long long)a.T_in * V;
    const int LB4 = a.LB >> 2;
    const int nvec = BKV * LB4;
    int p_lds[NPF], p_kk[NPF];
    long long p_g[NPF];
    bool p_ok[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        int e = tid + i * NTHREADS;
        int kk = e / LB4, c4 = e - kk * LB4;
        int pos = c4 << 2;
        int slot = pos / Vp, v = pos - slot * Vp;          // Vp == V: a piece may run on into the next frame (contiguous 1x1 form only)
        int th = tin0 + slot * a.lstride;
        bool ok = e < nvec && th >= 0;
        if (a.up > 1) { ok = ok && (th % a.up == 0); th /= a.up; }
        ok = ok && th < a.T_in;
        p_ok[i] = ok;
        p_kk[i] = e < nvec ? kk : -1;
        p_lds[i] = kk * a.pitchB + pos;
        p_g[i] = (long long)n * a.src.ctot * chan_stride + (long long)(a.src.coff + kk) * chan_stride + (long long)th * V + v0 + v;
    }
    float4 r1[NPF], r2[NPF];
    const bool has2 = a.src.x2 != nullptr;
    // the weight tile of a chunk (KT*BMT*BKV floats) is prefetched as well: NAF values per thread.  Round 3: for the k x 1
    // kernels too (up to KTP = 5 taps on the BKV = 16 variants, which is where the temporal branches run): their tile used to
    // be fetched by a load-store loop inside the K loop -- 5 to 20 dependent L2 round trips per chunk, which is what the
    // k = 5 launches spent most of their 33..153 us on (roofs 7..25 us, profiles/r03_roofline_table_nucla.txt)
    constexpr int KTP = BKV == 16 ? 5 : 1;
    constexpr int NAF = (KTP * BMT * BKV + NTHREADS - 1) / NTHREADS;
    const bool apf = a.KT <= KTP;
    float wr[NAF];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            r1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            r2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p_ok[i] && k0 + p_kk[i] < a.K) {
                long long g = p_g[i] + (long long)k0 * chan_stride;
                r1[i] = *reinterpret_cast<const float4*>(a.src.x1 + g);
                if (has2) r2[i] = *reinterpret_cast<const float4*>(a.src.x2 + g);
            }
        }
        if (apf) {
#pragma unroll
            for (int i = 0; i < NAF; ++i) {
                int e = tid + i * NTHREADS;
                const int tap = e / (BMT * BKV);
                e -= tap * (BMT * BKV);
                int ii, kk;
                if (a.wmode == 0) { kk = e % BKV; ii = e / BKV; } else { ii = e % BMT; kk = e / BMT; }
                int m = m0 + ii, k = k0 + kk;
                wr[i] = (tap < a.KT && m < a.M && k < a.K) ? a.w[m * a.ws_m + k * a.ws_k + tap * a.ws_t + a.w_off] : 0.f;
            }
        }
    };
    TG_T(tt0);
    prefetch(0);
    __syncthreads();                                   // cf table visible
    TG_T(tt1); TG_ACC(0, tt1 - tt0);

    const float* arow = As + j * BKVP + kq;
    for (int k0 = 0; k0 < a.K; k0 += BKV) {
        TG_T(ta);
        __syncthreads();                               // previous chunk's MFMAs done with As/Bs
        TG_T(tb); TG_ACC(1, tb - ta);
#ifdef TAMGCN_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TG_T(tw); TG_ACC(2, tw - tb);
        const int nA = a.KT * BMT * BKV;
        if (apf) {
#pragma unroll
            for (int i = 0; i < NAF; ++i) {
                int e = tid + i * NTHREADS;
                const int tap = e / (BMT * BKV);
                e -= tap * (BMT * BKV);
                int ii, kk;
                if (a.wmode == 0) { kk = e % BKV; ii = e / BKV; } else { ii = e % BMT; kk = e / BMT; }
                if (tap < a.KT) As[(tap * BMT + ii) * BKVP + kk] = wr[i];
            }
        } else if (a.wmode == 0) {                     // weight rows contiguous along k
            for (int e = tid; e < nA; e += NTHREADS) {
                int kk = e % BKV;
                int i = (e / BKV) % BMT;
                int tap = e / (BKV * BMT);
                int m = m0 + i, k = k0 + kk;
                float wv = 0.f;
                if (m < a.M && k < a.K) wv = a.w[m * a.ws_m + k * a.ws_k + tap * a.ws_t + a.w_off];
                As[(tap * BMT + i) * BKVP + kk] = wv;
            }
        } else {                                       // transposed view: contiguous along m
            for (int e = tid; e < nA; e += NTHREADS) {
                int i = e % BMT;
                int kk = (e / BMT) % BKV;
                int tap = e / (BKV * BMT);
                int m = m0 + i, k = k0 + kk;
                float wv = 0.f;
                if (m < a.M && k < a.K) wv = a.w[m * a.ws_m + k * a.ws_k + tap * a.ws_t + a.w_off];
                As[(tap * BMT + i) * BKVP + kk] = wv;
            }
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            if (p_kk[i] >= 0) {
                int k = k0 + p_kk[i];
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p_ok[i] && k < a.K) {
                    float c1 = cf[k], c2 = cf[a.K + k], c0 = cf[2 * a.K + k];
                    o.x = fmaf(c1, r1[i].x, fmaf(c2, r2[i].x, c0)); o.y = fmaf(c1, r1[i].y, fmaf(c2, r2[i].y, c0));
                    o.z = fmaf(c1, r1[i].z, fmaf(c2, r2[i].z, c0)); o.w = fmaf(c1, r1[i].w, fmaf(c2, r2[i].w, c0));
                    if (a.src.act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                }
                *reinterpret_cast<float4*>(Bs + p_lds[i]) = o;
            }
        }
        TG_T(tc); TG_ACC(3, tc - tw);
        __syncthreads();
        TG_T(td); TG_ACC(4, td - tc);
        if (k0 + BKV < a.K) prefetch(k0 + BKV);        // in flight under the MFMAs below
        TG_T(te); TG_ACC(5, te - td);
        for (int tap = 0; tap < a.KT; ++tap) {         // branch-free MFMA block: MT x CWT tiles per k-step
            const float* at = arow + tap * BMT * BKVP;
            const float* bt_ = Bs + kq * a.pitchB + tap * a.dil * Vp;
#pragma unroll
            for (int k4 = 0; k4 < BKV / 4; ++k4) {
                float av[MT], bv[CWT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = at[mt * 16 * BKVP + k4 * 4];
#pragma unroll
                for (int c = 0; c < CWT; ++c) bv[c] = bt_[k4 * 4 * a.pitchB + boff[c]];
#pragma unroll
                for (int c = 0; c < CWT; ++c)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][c] = mfma16(av[mt], bv[c], acc[mt][c]);
            }
        }
        TG_T(tf); TG_ACC(6, tf - te);
    }
    TG_T(tg0);

    // ---- epilogue.  The accumulators are staged through LDS (the operand tiles are dead) and
    // a compact loop finishes one float4 per iteration: bias / broadcast / residual adds / mask /
    // BatchNorm moments with 16-byte coalesced loads and stores.  (A fully unrolled register
    // epilogue was 90 KB of code: it thrashed the instruction cache and scaled with MT*CW.)
    __syncthreads();                                   // all MFMA reads of As/Bs are done
    float* Tt = smem;                                  // [rows_per_pass][PT]
    constexpr int PT = CWT * 64 + 4;
    const int avail = ((a.KT * BMT * BKVP + 3) & ~3) + BKV * a.pitchB;
    int RP = (avail / PT) & ~15;                       // rows per pass: 16, 32, 48 or 64 (host checks >= 16)
    if (RP > BMT) RP = BMT;
    if (RP == 48) RP = 32;
    if (a.stats_part) {
        for (int e = tid; e < 2 * 4 * BM; e += NTHREADS) Ss[e] = 0.f;
    }
    for (int r0 = 0; r0 < BMT; r0 += RP) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rl = mt * 16 + kq * 4 - r0;      // local row of this lane's first accumulator row
            if (rl >= 0 && rl < RP) {
#pragma unroll
                for (int c = 0; c < CWT; ++c) {
                    const int col = (cw0 + c) * 16 + j;
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tt[(rl + r) * PT + col] = acc[mt][c][r];
                }
            }
        }
        __syncthreads();
        staged_rows<NTHREADS>(a, Tt, PT, RP, r0, BMT, m0, n, t0, ncols, Ss, Vs, v0);
    }
    if (a.stats_part) {
        __syncthreads();
        if (tid < 2 * BMT) {
            int st = tid / BMT, row = tid - st * BMT;
            int m = m0 + row;
            if (m < a.M) {
                int part = n * gridDim.x + blockIdx.x;
                a.stats_part[((long long)st * a.stats_ctot + a.stats_coff + m) * a.nparts + part] = Ss[st * 4 * BM + row];
            }
        }
    }
}


// ===========================================================================
// 1x1, stride-1 convolution = a GEMM  Y[m][p] = sum_k W[m][k] act(X[k][p])  whose columns p = (t, v)
// are contiguous per channel.  Both operands reach LDS by LDS-DMA (global_load_lds): no VGPR round
// trip, no staging VALU, no ds_write pass -- the register-staged kernel above spent as long issuing
// loads and writing LDS as in its MFMAs (tools/conv_phases.py).  A ring of NST stages keeps two chunks
// in flight across ONE raw s_barrier per chunk (counted vmcnt, never 0 in the loop).
//
//   workgroup   512 threads = 8 waves as 2 (rows) x 4 (columns); tile 64 channels x BT frames (<= 320 cols)
//   B image     [BK][LB] floats per source, lane-linear per 4-row group (5 x 1 KB pieces at LB = 320);
//               odd k rows are rotated by 16 columns ON THE SOURCE ADDRESS, so the two k rows a 32-lane
//               half reads sit 16 banks apart (LB == 0 mod 32: conflict-free)
//   A image     [BK][PA = 80]: one dword LDS-DMA per k row, lane = output channel (pitch == 16 mod 32)
//   prologue    BatchNorm(-backward) apply / two-source combine / ReLU happen at the fragment read:
//               each B element is read by exactly one wave, so this costs 1-3 VALU per element
//   epilogue    the staged float4 epilogue of conv_kernel_vec (staged_rows)
// ===========================================================================

// TG_KO: knock-out side builds for tools/conv_knockout.py (what each part of the kernel costs; results are WRONG by design):
// 1 no epilogue, 2 no prologue arithmetic on the B fragments, 4 no A DMA, 8 no B DMA, 16 no MFMAs, 32 no fragment reads
#ifndef TG_KO
#define TG_KO 0
#endif
constexpr int G_NT = 512, G_BMT = 64, G_PA = 80, G_PBMAX = 320, G_NST = 3, G_CWT = 5;

typedef __attribute__((address_space(1))) const void* tg_gptr;
typedef __attribute__((address_space(3))) void* tg_lptr;

#define TG_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {     // n is wave-uniform
    switch (n) {
        TG_VMCNT_CASE(0) TG_VMCNT_CASE(1) TG_VMCNT_CASE(2) TG_VMCNT_CASE(3) TG_VMCNT_CASE(4)
        TG_VMCNT_CASE(5) TG_VMCNT_CASE(6) TG_VMCNT_CASE(7) TG_VMCNT_CASE(8) TG_VMCNT_CASE(9) TG_VMCNT_CASE(10)
        TG_VMCNT_CASE(11) TG_VMCNT_CASE(12) TG_VMCNT_CASE(13) TG_VMCNT_CASE(14) TG_VMCNT_CASE(15)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// NW = waves per workgroup: 8 (2 x 4 waves of 32 rows x 80 columns, two accumulator row tiles) or 4 (1 x 4 waves of 64 rows x 80
// columns, four row tiles).  On this chip a VALU instruction does not run BESIDE an fp32 MFMA, it runs INSTEAD of one
// (tools/probes/coissue_probe.hip: 16 MFMAs + 64 v_fma per SIMD take 778 cycles where the MFMAs alone take 569; issued from another
// wave of the SIMD it is worse), so what bounds this kernel is VALU instructions per MFMA: the prologue and the address
// arithmetic of a B fragment are paid once per wave that reads it -- four row tiles per wave halve both.
// PRO: the source carries a prologue (coefficients, second source or ReLU).  A plain tensor (the dx <- dx3 GEMMs, K = 3 C) needs neither
// the [3][K] coefficient table -- which alone pushed K >= 384 past 80 KB, i.e. to ONE workgroup per CU -- nor the VALU on the fragments.
template <int NSRC, int BK, bool PRO, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void conv1x1_glds_kernel(const ConvArgs a, int ntt, int nmt) {
    constexpr int G_NT = NW * 64, G_MT = 16 / NW;                 // threads; row tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STG = BK * G_PBMAX * NSRC + BK * G_PA;          // floats per stage
    constexpr int NAI = BK / NW;                                  // A pieces per wave and chunk
    constexpr int MAXB = (NSRC * (BK / 4) * 5 + NW - 1) / NW;     // B pieces per wave and chunk (upper bound)
    float* cf = smem + G_NST * STG;                               // [3][K] (PRO only)
    float* Ss = cf + (PRO ? 3 * a.K : 0);                         // [2][4*BM] row moments

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4, wm = wave >> 2, wn = wave & 3;
    const int V = a.V, K = a.K, LB = a.LB;
    // odd k rows are rotated so that the two rows a 32-lane half reads sit 16 banks apart: a source column c of an odd row
    // sits at position c - rot, LB - rot == 16 (mod 32) (16 columns at LB % 32 == 0; V = 25 tiles have LB = 300: 28 columns).  LB % 4 == 0 keeps rot a whole 16-byte slot.
    const int rot = ((LB & 31) + 16) & 31;
    // workgroup -> (sample, frame tile, channel tile); the channel tiles of one (n, tt) share an XCD's L2
    int g, mtile;
    {
        const int L = blockIdx.x, ng = a.N * ntt;
        if ((ng & 7) == 0) { const int xcd = L & 7, i = L >> 3; mtile = i % nmt; g = (i / nmt) * 8 + xcd; }
        else { g = L / nmt; mtile = L - g * nmt; }
    }
    const int n = g / ntt, tt = g - n * ntt;
    const int m0 = mtile * G_BMT, t0 = tt * a.BT;
    const int bt = min(a.BT, a.T_out - t0);
    const int ncols = bt * V;
    const long long TV = (long long)a.T_in * V;

    // ---- per-lane DMA descriptors (the same for every chunk)
    const int NQ = (LB + 63) >> 6;                                // 1 KB pieces per 4-row group
    const int ls = a.lstride;                                     // frame stride of the source (1; 2: the strided 1x1 convs, host-checked)
    const float rV_ = 1.0f / (float)V;
    const int NI1 = (BK / 4) * NQ;                                // pieces per source and chunk
    int b_rel[MAXB], b_dst[MAXB];                                 // source offset (floats), LDS offset of the piece (floats)
    bool b_ok[MAXB], b_on[MAXB];
    int nissue = 0;                                               // + the A pieces, below
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int id = wave + i * NW;
        const bool idok = id < NSRC * NI1;
        const int src = idok ? id / NI1 : 0;
        const int rem = id - src * NI1;
        const int grp = rem / NQ, q = rem - grp * NQ;
        const int f = q * 256 + lane * 4;                         // float index inside the 4-row group
        const int r = f / LB, off = f - r * LB;
        int col = off + (r & 1) * rot;                            // source column of this LDS slot
        if (col >= LB) col -= LB;
        b_ok[i] = idok && r < 4 && col < ncols;
        // a 1x1 conv with a temporal stride (V % 4 == 0: a 16-byte slot never leaves its frame) reads every ls-th frame
        const int scol = ls > 1 ? col + tc_like_div(col, rV_) * (ls - 1) * V : col;
        b_rel[i] = (int)((grp * 4 + r) * TV) + scol + src * 0x40000000;   // bit 30 tags the second source
        b_dst[i] = src * (BK * G_PBMAX) + grp * 4 * LB + q * 256;
        b_on[i] = __ballot(b_ok[i]) != 0ull;                      // wave-uniform: the piece exists
        nissue += b_on[i] ? 1 : 0;
    }
    const float* xb1 = a.src.x1 + ((long long)n * a.src.ctot + a.src.coff) * TV + (long long)t0 * ls * V;
    const float* xb2 = NSRC == 2 ? a.src.x2 + ((long long)n * a.src.ctot + a.src.coff) * TV + (long long)t0 * ls * V : nullptr;
    // A (weight) pieces.  wmode 1 (w[k][m], data gradient): one dword piece per k row, lane = channel, coalesced.
    // wmode 0 (w[m][k]): a row's BK taps are contiguous, so 64/BK*4.. lanes share a row: dwordx4 pieces of
    // (1024 / (4*BK)) rows, image [m][BK]; the 16-byte quads of a row are XOR-swizzled on the SOURCE address
    // (quad q of row m sits at q ^ f(m)) so the column reads below stay at most 2-way conflicted.
    constexpr int RPP = 256 / BK;                                 // rows per 1 KB piece (wmode 0)
    constexpr int NAP0 = G_BMT / RPP;                             // pieces per chunk (wmode 0): 4 (BK 16) or 2 (BK 8)
    const bool wm0 = a.wmode == 0;
    bool a_ok; const float* wbase; int a_dst0 = 0;
    if (wm0) {
        const int ml = lane / (BK / 4), q = lane - ml * (BK / 4);   // row inside the piece, physical quad
        const int m = wave * RPP + ml;                              // row inside the tile (waves >= NAP0 idle)
        const int f = BK == 16 ? (m >> 1) & 3 : (m >> 2) & 1;
        a_ok = wave < NAP0 && m0 + m < a.M;
        wbase = a.w + (long long)(m0 + (a_ok ? m : 0)) * a.ws_m + a.w_off + 4 * (q ^ f);
        a_dst0 = wave * 256;
    } else {
        a_ok = m0 + lane < a.M;
        wbase = a.w + (long long)(m0 + (a_ok ? lane : 0)) * a.ws_m + a.w_off;
    }
    const bool a_on = __ballot(a_ok) != 0ull;
    nissue += wm0 ? (a_on ? 1 : 0) : NAI;

    auto issue = [&](int c) {
        float* st = smem + (c % G_NST) * STG;
        const int k0 = c * BK;
        if (TG_KO & 4) {
        } else if (wm0) {
            if (a_on) {
                if (a_ok) __builtin_amdgcn_global_load_lds((tg_gptr)(wbase + k0), (tg_lptr)(st + BK * G_PBMAX * NSRC + a_dst0), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const int ka = wave * NAI + i;
                if (a_ok) __builtin_amdgcn_global_load_lds((tg_gptr)(wbase + (long long)(k0 + ka) * a.ws_k),
                                                           (tg_lptr)(st + BK * G_PBMAX * NSRC + ka * G_PA), 4, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < ((TG_KO & 8) ? 0 : MAXB); ++i) {
            if (b_on[i]) {
                const float* base = (NSRC == 2 && (b_rel[i] & 0x40000000)) ? xb2 : xb1;
                const float* gp = base + (long long)k0 * TV + (b_rel[i] & 0x3fffffff);
                if (b_ok[i]) __builtin_amdgcn_global_load_lds((tg_gptr)gp, (tg_lptr)(st + b_dst[i]), 16, 0, 0);
            }
        }
    };

    // fragment addressing: this lane reads k rows of parity kq&1 only
    int bslot[G_CWT];
#pragma unroll
    for (int c = 0; c < G_CWT; ++c) {
        int col = (wn * G_CWT + c) * 16 + j;
        if (col >= ncols) col = 0;                                // padding tile: in-bounds data, never stored
        int sl = col - (kq & 1) * rot;
        bslot[c] = sl < 0 ? sl + LB : sl;
    }
    // Fragment addresses = (per-lane base, 2 + 5 registers) + (a wave-uniform offset per stage and k4 step, recomputed on the
    // scalar unit and hidden from the optimiser: left visible, hipcc hoists all 4 x 7 sums out of the K loop into registers).
    int vb[G_CWT], va[G_MT];
#pragma unroll
    for (int c = 0; c < G_CWT; ++c) vb[c] = kq * LB + bslot[c];
#pragma unroll
    for (int mt = 0; mt < G_MT; ++mt) {
        const int m = wm * (G_MT * 16) + mt * 16 + j;
        const int f = BK == 16 ? (m >> 1) & 3 : (m >> 2) & 1;
        va[mt] = wm0 ? m * BK + 4 * f + kq : kq * G_PA + m;         // wmode 0: quad k4 of row m sits at k4 ^ f, i.e. base ^ (4 k4)
    }
    const int axor = wm0 ? 4 : 0, aadd = wm0 ? 0 : 4 * G_PA;
    f32x4 acc[G_MT][G_CWT];
#pragma unroll
    for (int mt = 0; mt < G_MT; ++mt)
#pragma unroll
        for (int c = 0; c < G_CWT; ++c) acc[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float lo = a.src.act == 1 ? 0.f : -__builtin_inff();
    TG_T(tt0);
    const int nch = K / BK;
    // The prologue coefficients are requested FIRST and the whole ring (three chunks) right behind them: vmcnt retires in
    // order, so waiting for "all but the chunks" is waiting for the coefficients only, and a raw barrier publishes them
    // (__syncthreads would drain the chunks too -- the kernel used to start its K loop with an empty third stage).
    float cfr[1024 / G_NT][3];
#pragma unroll
    for (int i = 0; i < (PRO ? 1024 / G_NT : 0); ++i) {                     // K <= 1024 (host-checked)
        const int e = tid + i * G_NT, ch = a.src.coff + (e < K ? e : 0);
        cfr[i][0] = a.src.coef ? a.src.coef[ch] : 1.f;
        cfr[i][1] = (a.src.coef && a.src.x2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cfr[i][2] = a.src.coef ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }
    const int npre = nch < G_NST ? nch : G_NST;
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 2) issue(2);
    wait_vmcnt(npre * nissue);
#pragma unroll
    for (int i = 0; i < (PRO ? 1024 / G_NT : 0); ++i) {
        const int e = tid + i * G_NT;
        if (e < K) { cf[e] = cfr[i][0]; cf[K + e] = cfr[i][1]; cf[2 * K + e] = cfr[i][2]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TG_T(tt1); TG_ACC(0, tt1 - tt0);
    // Software-pipelined fragment reads.  Left to itself hipcc emits "8 ds_reads, s_waitcnt lgkmcnt(0), VALU, 8 MFMAs" per
    // k4 step; round-robin issue keeps the waves of a SIMD in lockstep, so they all wait for LDS together and the matrix
    // pipe idles for every round trip (tools/conv_knockout.py: reads + MFMAs without any DMA take 350 us where the MFMAs
    // alone take 229, K768 -> M256).  Here the reads of step i + 1 are issued in front of step i's MFMAs (two register
    // sets), across the chunk boundary too: the barrier that publishes chunk c + 1 sits in front of chunk c's LAST step,
    // whose fragments are already in registers -- so stage c is free there and chunk c + 3 is requested into it.
    constexpr int NK = BK / 4;
    constexpr int NRD = G_MT + G_CWT * NSRC + (PRO ? NSRC + 1 : 0);   // LDS reads per step
    float fa[2][G_MT], bvs[2][G_CWT];                                 // the fragments MFMAs read: two sets
    float fb[G_CWT], fb2[NSRC == 2 ? G_CWT : 1], fc[3];               // raw reads of the NEXT step (prologue not applied yet)
    const int cfo = G_NST * STG;                                      // float index of the coefficient table
    auto rd = [&](int set, int stage, int kg, int k4) {                // kg: the chunk's first contraction row
        int sb = stage * STG + k4 * 4 * LB, sa = stage * STG + BK * G_PBMAX * NSRC + k4 * aadd, sc = cfo + kg + k4 * 4, sx = k4 * axor;
        asm volatile("" : "+s"(sb), "+s"(sa), "+s"(sc), "+s"(sx));
#pragma unroll
        for (int mt = 0; mt < G_MT; ++mt) fa[set][mt] = smem[sa + (va[mt] ^ sx)];
        if constexpr (PRO && !(TG_KO & 2)) {
            fc[0] = smem[sc + kq]; fc[2] = smem[sc + 2 * K + kq];
            if constexpr (NSRC == 2) fc[1] = smem[sc + K + kq];
        }
#pragma unroll
        for (int cc = 0; cc < G_CWT; ++cc) {
            fb[cc] = smem[sb + vb[cc]];
            if constexpr (NSRC == 2) fb2[cc] = smem[sb + vb[cc] + BK * G_PBMAX];
        }
    };
    // the prologue in packed fp32 (v_pk_fma_f32: two elements per instruction -- a VALU instruction costs an fp32 MFMA slot here),
    // the ReLU only where the source has one (same operations in the same order as the scalar form: bit-identical)
    typedef float pf2 __attribute__((ext_vector_type(2)));
    const bool relu = a.src.act == 1;
    auto pro = [&](int set) {
        if constexpr (!PRO) {
#pragma unroll
            for (int cc = 0; cc < G_CWT; ++cc) bvs[set][cc] = fb[cc];
        } else if (TG_KO & 2) {
#pragma unroll
            for (int cc = 0; cc < G_CWT; ++cc) bvs[set][cc] = fb[cc] + (NSRC == 2 ? fb2[cc] : 0.f);
        } else {
            const pf2 C1 = {fc[0], fc[0]}, C2 = {fc[1], fc[1]}, C0 = {fc[2], fc[2]};
#pragma unroll
            for (int cc = 0; cc + 1 < G_CWT; cc += 2) {
                pf2 v = {fb[cc], fb[cc + 1]};
                if constexpr (NSRC == 2) v = __builtin_elementwise_fma(C1, v, __builtin_elementwise_fma(C2, (pf2){fb2[cc], fb2[cc + 1]}, C0));
                else v = __builtin_elementwise_fma(C1, v, C0);
                bvs[set][cc] = v[0]; bvs[set][cc + 1] = v[1];
            }
            if constexpr (G_CWT & 1) {
                constexpr int cc = G_CWT - 1;
                if constexpr (NSRC == 2) bvs[set][cc] = fmaf(fc[0], fb[cc], fmaf(fc[1], fb2[cc], fc[2]));
                else bvs[set][cc] = fmaf(fc[0], fb[cc], fc[2]);
            }
            if (relu) {
#pragma unroll
                for (int cc = 0; cc < G_CWT; ++cc) asm("v_max_f32 %0, 0, %0" : "+v"(bvs[set][cc]));   // ONE instruction (fmaxf(x, 0.f) and fmed3 both come with a canonicalising v_max)
            }
        }
    };
    // Pinning the order.  Neither sched_barrier nor sched_group_barrier holds it (instruction selection lets the unchained
    // MFMA nodes float across them).  An empty asm that redefines registers is a real dependence:
    //   mm:  the fragments of this step pass through one AFTER the next step's reads were issued ("memory": the loads stay above)
    //   fin: every accumulator and the raw reads pass through one, the prologue consumes from it -- hipcc puts the
    //        s_waitcnt for the reads in front of it, i.e. behind ten MFMAs' issue; a third keeps the prologue from sinking
    static_assert((G_MT == 2 || G_MT == 4) && G_CWT == 5, "operand lists below");
#define TG_ACC5(mt_) "+v"(acc[mt_][0]), "+v"(acc[mt_][1]), "+v"(acc[mt_][2]), "+v"(acc[mt_][3]), "+v"(acc[mt_][4])
#define TG_V5(x_) "+v"(x_[0]), "+v"(x_[1]), "+v"(x_[2]), "+v"(x_[3]), "+v"(x_[4])
    auto mm = [&](int cur) {
        if constexpr (G_MT == 2) asm volatile("" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), TG_V5(bvs[cur]) :: "memory");
        else asm volatile("" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), "+v"(fa[cur][2]), "+v"(fa[cur][3]), TG_V5(bvs[cur]) :: "memory");
#pragma unroll
        for (int cc = 0; cc < G_CWT; ++cc)
#pragma unroll
            for (int mt = 0; mt < G_MT; ++mt) acc[mt][cc] = mfma16(fa[cur][mt], bvs[cur][cc], acc[mt][cc]);
    };
    auto fin = [&](int nxt) {
        if constexpr (G_MT == 4) asm volatile("" : TG_ACC5(2), TG_ACC5(3) :: "memory");   // (volatile asms keep their order)
        if constexpr (NSRC == 2) asm volatile("" : TG_ACC5(0), TG_ACC5(1), TG_V5(fb), TG_V5(fb2) :: "memory");
        else asm volatile("" : TG_ACC5(0), TG_ACC5(1), TG_V5(fb) :: "memory");
        pro(nxt);
        asm volatile("" : TG_V5(bvs[nxt]));
    };
#undef TG_ACC5
#undef TG_V5
    wait_vmcnt((npre - 1) * nissue);                                  // this wave's pieces of chunk 0 have landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // everyone's have
    rd(0, 0, 0, 0);
    pro(0);
    int stg = 0;
    for (int c = 0; c + 1 < nch; ++c) {                               // every chunk but the last: its last step crosses into the next
        const int k0 = c * BK;
#pragma unroll
        for (int s_ = 0; s_ + 1 < NK; ++s_) {
            rd((s_ & 1) ^ 1, stg, k0, s_ + 1);
            mm(s_ & 1);
            fin((s_ & 1) ^ 1);
        }
        wait_vmcnt(c + 2 < nch ? nissue : 0);                         // this wave's pieces of chunk c + 1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone's have; every fragment of chunk c is in registers
        if (c + 3 < nch) issue(c + 3);
        stg = stg == G_NST - 1 ? 0 : stg + 1;
        rd(NK & 1, stg, k0 + BK, 0);                                  // NK is even: the next chunk starts on set 0 again
        mm((NK - 1) & 1);
        fin(NK & 1);
    }
    {
        const int k0 = (nch - 1) * BK;
#pragma unroll
        for (int s_ = 0; s_ < NK; ++s_) {
            if (s_ + 1 < NK) rd((s_ & 1) ^ 1, stg, k0, s_ + 1);
            mm(s_ & 1);
            if (s_ + 1 < NK) fin((s_ & 1) ^ 1);
        }
    }
    TG_T(tg0);

#if TG_KO & 1
    {
        float ssum = 0.f;
        for (int mt = 0; mt < G_MT; ++mt)
            for (int c = 0; c < G_CWT; ++c) ssum += acc[mt][c][0] + acc[mt][c][1] + acc[mt][c][2] + acc[mt][c][3];
        if (ssum == 123.456f) a.y[tid] = ssum;
        return;
    }
#endif
    // ---- staged epilogue (two passes of 32 rows through the dead stage buffers)
    constexpr int PT = G_CWT * 64 + 4, RP = 32;
    float* Tt = smem;
    if (a.stats_part) {
        for (int e = tid; e < 2 * 4 * BM; e += G_NT) Ss[e] = 0.f;
    }
    for (int r0 = 0; r0 < G_BMT; r0 += RP) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < G_MT; ++mt) {
            const int rl = wm * (G_MT * 16) + mt * 16 - r0;       // first local row of this row tile in the pass
            if (rl >= 0 && rl < RP) {
#pragma unroll
                for (int c = 0; c < G_CWT; ++c) {
                    const int col = (wn * G_CWT + c) * 16 + j;
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tt[(rl + kq * 4 + r) * PT + col] = acc[mt][c][r];
                }
            }
        }
        __syncthreads();
        staged_rows<G_NT>(a, Tt, PT, RP, r0, G_BMT, m0, n, t0, ncols, Ss, V, 0);
    }
    if (a.stats_part) {
        __syncthreads();
        if (tid < 2 * G_BMT) {
            int stt = tid / G_BMT, row = tid - stt * G_BMT;
            int m = m0 + row;
            if (m < a.M) a.stats_part[((long long)stt * a.stats_ctot + a.stats_coff + m) * a.nparts + g] = Ss[stt * 4 * BM + row];
        }
    }
    TG_T(tg1); TG_ACC(7, tg1 - tg0); TG_ACC(8, tg1 - tt0); TG_ACC(9, 1);
}


// ===========================================================================
// Data-gradient form of the pointwise GEMM for wide layers (M >= 128, single source, w[k][m]):
//   tile 128 channels x <= 320 columns, chunk of 32 contraction rows, two stages, split-fp32 MFMA.
// The matrix work of dx <- dx3 at C >= 128 (K = 3C) is what bounds the fp32 kernel (65-75 TFLOP/s of the ~105 the
// chip sustains); activation GRADIENTS never decide a ReLU mask, so the 4e-6 relative error of three
// v_mfma_f32_16x16x32_bf16 stays a linear 1e-5-level perturbation (unlike in the forward, see ctrgc.hip).
//   B image   as conv1x1_glds_kernel ([32][LB], odd rows rotated 16 columns)
//   A image   [32][128]: dwordx4 pieces of 2 k rows, channel quads XOR-swizzled by 4*(k & 1) on the source address
//   fragment  lane (j, kq) owns k = 4e + kq, e = 0..7: ONE K = 32 step per chunk; B of a column tile is split once and
//             meets the four row tiles' pre-split A fragments
// ===========================================================================
constexpr int GS_BK = 32, GS_NST = 2;

// TERMS = 2: the two-term split (backward: weight view w[k][m], wmode 1);  TERMS = 3, WM0: the three-term split for the
// FORWARD 1x1 convolutions into >= 128 channels (weight rows w[m][k]), fp32-exact to rounding, with BatchNorm moments.
// MT = row tiles per wave: 4 -> 128-row workgroup tile (layers into >= 128 channels), 2 -> 64 rows (the C = 64 layers' data gradients).
template <int TERMS, bool WM0, int MT>
__global__ __launch_bounds__(G_NT) void conv1x1_glds_split_kernel(const ConvArgs a, int ntt, int nmt) {
    constexpr int GS_MT = MT, GS_BMT = 32 * MT, WROWS = 16 * MT;   // rows of the workgroup tile / of one wave row-group
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BK = GS_BK;
    constexpr int STG = BK * G_PBMAX + BK * GS_BMT;               // floats per stage
    constexpr int MAXB = ((BK / 4) * 5 + 7) / 8;                  // B pieces per wave and chunk
    constexpr int NAP = BK * GS_BMT / 256 / 8;                    // A pieces per wave and chunk (2 at 128 rows, 1 at 64)
    constexpr int QR = GS_BMT / 4, RPP = 64 / QR;                 // w[k][m] image: channel quads per k row, k rows per piece
    float* cf = smem + GS_NST * STG;                              // [3][K]
    float* Ss = cf + 3 * a.K;                                     // [2][4*BM] row moments
    float* Bc = Ss + 2 * 4 * BM;                                  // [128] per-row constant of the linear prologue

    TG_T(tt0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4, wm = wave >> 2, wn = wave & 3;
    const int V = a.V, K = a.K, LB = a.LB;
    // odd k rows are rotated so that the two rows a 32-lane half reads sit 16 banks apart: a source column c of an odd row
    // sits at position c - rot, LB - rot == 16 (mod 32) (16 columns at LB % 32 == 0; V = 25 tiles have LB = 300: 28 columns).  LB % 4 == 0 keeps rot a whole 16-byte slot.
    const int rot = ((LB & 31) + 16) & 31;
    int g, mtile;
    {
        const int L = blockIdx.x, ng = a.N * ntt;
        if ((ng & 7) == 0) { const int xcd = L & 7, i = L >> 3; mtile = i % nmt; g = (i / nmt) * 8 + xcd; }
        else { g = L / nmt; mtile = L - g * nmt; }
    }
    const int n = g / ntt, tt = g - n * ntt;
    const int m0 = mtile * GS_BMT, t0 = tt * a.BT;
    const int bt = min(a.BT, a.T_out - t0);
    const int ncols = bt * V;
    const long long TV = (long long)a.T_in * V;

    for (int e = tid; e < K; e += G_NT) {
        int ch = a.src.coff + e;
        cf[e] = a.src.coef ? a.src.coef[ch] : 1.f;
        cf[K + e] = (a.src.coef && a.src.x2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cf[2 * K + e] = a.src.coef ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }
    // The operand is LINEAR here (act == 0, host-checked): W^T (c1 x1 + c2 x2 + c0) = (W c1)^T x1 + (W c2)^T x2 + W^T c0.
    // The coefficients scale the A fragment, a second source is just K more contraction rows, and the constant term
    // is a per-channel bias the lanes accumulate from their A values: B is never touched by the prologue.
    const bool plain = !a.src.coef;
    const int nsrc = a.src.x2 ? 2 : 1;

    // ---- B pieces (as in conv1x1_glds_kernel)
    const int NQ = (LB + 63) >> 6, NI1 = (BK / 4) * NQ;
    int b_rel[MAXB], b_dst[MAXB];
    bool b_ok[MAXB], b_on[MAXB];
    int nissue = 0;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int id = wave + i * 8;
        const bool idok = id < NI1;
        const int grp = (idok ? id : 0) / NQ, q = (idok ? id : 0) - grp * NQ;
        const int f = q * 256 + lane * 4;
        const int r = f / LB, off = f - r * LB;
        int col = off + (r & 1) * rot;
        if (col >= LB) col -= LB;
        b_ok[i] = idok && r < 4 && col < ncols;
        b_rel[i] = (int)((grp * 4 + r) * TV) + col;
        b_dst[i] = grp * 4 * LB + q * 256;
        b_on[i] = __ballot(b_ok[i]) != 0ull;
        nissue += b_on[i] ? 1 : 0;
    }
    const long long xoff = ((long long)n * a.src.ctot + a.src.coff) * TV + (long long)t0 * V;
    const float* xb1 = a.src.x1 + xoff;
    const float* xb2 = a.src.x2 ? a.src.x2 + xoff : nullptr;
    const int nchk = K / BK;                                      // chunks per source
    // ---- A pieces: piece p of this wave covers k rows 2p', lanes 0-31 row 2p', lanes 32-63 row 2p'+1; 32 quads of 4 channels
    int a_rel[NAP], a_dst[NAP];
    bool a_ok[NAP];
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int piece = wave * NAP + i;                         // 0..15
        if constexpr (WM0) {
            // weight rows w[m][k]: a piece = 8 rows x 32 taps (128 B each); lane (r, q) fetches source quad q ^ (m & 7), so
            // that the 16 rows a fragment read touches spread over 8 quad positions (2-way instead of 16-way conflicts)
            const int m = piece * 8 + (lane >> 3), q = lane & 7;
            a_ok[i] = m0 + m < a.M;
            a_rel[i] = (m0 + (a_ok[i] ? m : 0)) * (int)a.ws_m + 4 * (q ^ (m & 7));
        } else {
            const int kr = piece * RPP + lane / QR, q = lane % QR;    // k row in the chunk, physical channel quad
            const int qs = q ^ ((kr & 1) << 2);                       // source quad: rows of different parity sit 16 banks apart
            a_ok[i] = m0 + qs * 4 < a.M;                              // M % 4 == 0 (host)
            a_rel[i] = kr * (int)a.ws_k + (m0 + qs * 4) * (int)a.ws_m;
        }
        a_dst[i] = BK * G_PBMAX + piece * 256;
    }
    nissue += NAP;
    auto issue = [&](int c) {
        float* st = smem + (c % GS_NST) * STG;
        const int k0 = (c % nchk) * BK;
        const float* xb = c < nchk ? xb1 : xb2;
#pragma unroll
        for (int i = 0; i < NAP; ++i)
            if (a_ok[i]) __builtin_amdgcn_global_load_lds((tg_gptr)(a.w + a.w_off + (long long)k0 * a.ws_k + a_rel[i]), (tg_lptr)(st + a_dst[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (b_on[i]) {
                const float* gp = xb + (long long)k0 * TV + b_rel[i];
                if (b_ok[i]) __builtin_amdgcn_global_load_lds((tg_gptr)gp, (tg_lptr)(st + b_dst[i]), 16, 0, 0);
            }
        }
    };

    int bslot[G_CWT];
#pragma unroll
    for (int c = 0; c < G_CWT; ++c) {
        int col = (wn * G_CWT + c) * 16 + j;
        if (col >= ncols) col = 0;
        int sl = col - (kq & 1) * rot;
        bslot[c] = sl < 0 ? sl + LB : sl;
    }
    int aoff[GS_MT];                                               // channel of row tile mt, swizzled for this lane's k parity
#pragma unroll
    for (int mt = 0; mt < GS_MT; ++mt) {
        const int m = wm * WROWS + mt * 16 + j;
        aoff[mt] = WM0 ? m : (((m >> 2) ^ ((kq & 1) << 2)) << 2) + (m & 3);
    }
    f32x4 acc[GS_MT][G_CWT];
#pragma unroll
    for (int mt = 0; mt < GS_MT; ++mt)
#pragma unroll
        for (int c = 0; c < G_CWT; ++c) acc[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[GS_MT];                                            // this lane's share of (W^T c0)[m]
#pragma unroll
    for (int mt = 0; mt < GS_MT; ++mt) bsum[mt] = 0.f;

    __syncthreads();
    const int nch = nsrc * nchk;
    issue(0);
    TG_T(tt1); TG_ACC(0, tt1 - tt0);
    for (int c = 0; c < nch; ++c) {
        TG_T(ta);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TG_T(tb); TG_ACC(2, tb - ta);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TG_T(tc); TG_ACC(4, tc - tb);
        if (c + 1 < nch) issue(c + 1);
        TG_T(td); TG_ACC(5, td - tc);
        const float* st = smem + (c % GS_NST) * STG;
        const float* As = st + BK * G_PBMAX;
        const int k0 = (c % nchk) * BK;
        float cs[8], c0[8];                                         // scale of this chunk's rows, constant term (first source only)
        if (!plain) {
            const float* cc1 = cf + (c < nchk ? 0 : K) + k0 + kq;
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs[e] = cc1[4 * e]; c0[e] = c < nchk ? cf[2 * K + k0 + 4 * e + kq] : 0.f; }
        }
        bf16x8_t ah[GS_MT], al[GS_MT];
#pragma unroll
        for (int mt = 0; mt < GS_MT; ++mt) {
            f32x4 v0, v1;
            if constexpr (WM0) {
                const int m = aoff[mt];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = As[m * BK + 4 * (e ^ (m & 7)) + kq]; v1[e] = As[m * BK + 4 * ((e + 4) ^ (m & 7)) + kq]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = As[(4 * e + kq) * GS_BMT + aoff[mt]]; v1[e] = As[(4 * (e + 4) + kq) * GS_BMT + aoff[mt]]; }
            }
            if (!plain) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[mt] = fmaf(v0[e], c0[e], fmaf(v1[e], c0[e + 4], bsum[mt]));
                    v0[e] *= cs[e]; v1[e] *= cs[e + 4];
                }
            }
            split_bf16x8(v0, v1, ah[mt], al[mt]);
        }
        TG_T(te); TG_ACC(3, te - td);
#pragma unroll
        for (int cc = 0; cc < G_CWT; ++cc) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = st[(4 * e + kq) * LB + bslot[cc]]; v1[e] = st[(4 * (e + 4) + kq) * LB + bslot[cc]]; }
            bf16x8_t bh, bl;
            split_bf16x8(v0, v1, bh, bl);
#pragma unroll
            for (int mt = 0; mt < GS_MT; ++mt) acc[mt][cc] = mfma_split(ah[mt], al[mt], bh, bl, acc[mt][cc]);
        }
        TG_T(tf); TG_ACC(6, tf - te);
    }
    TG_T(tg0);

    // ---- staged epilogue: four passes of 32 rows.  The constant term W^T c0: sum over the four kq lanes of a row,
    // one column-wave per row group publishes it (Ss is free: this kernel produces no moments).
    constexpr int PT = G_CWT * 64 + 4, RP = 32;
    float* Tt = smem;
    if (a.stats_part) {
        for (int e = tid; e < 2 * 4 * BM; e += G_NT) Ss[e] = 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < GS_MT; ++mt) {
        float t = bsum[mt];
        t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
        if (wn == 0 && kq == 0) Bc[wm * WROWS + mt * 16 + j] = t;
    }
    for (int r0 = 0; r0 < GS_BMT; r0 += RP) {
        __syncthreads();
        if (wm * WROWS <= r0 && r0 < wm * WROWS + WROWS) {
            const int mtb = (r0 - wm * WROWS) / 16;               // this pass = row tiles mtb, mtb+1 of the wave
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
                for (int c = 0; c < G_CWT; ++c) {
                    const int col = (wn * G_CWT + c) * 16 + j;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x4 v = (MT == 2 || mtb == 0) ? acc[mt2][c] : acc[(MT == 2 ? 0 : 2) + mt2][c];
                        Tt[(mt2 * 16 + kq * 4 + r) * PT + col] = v[r] + Bc[r0 + mt2 * 16 + kq * 4 + r];
                    }
                }
        }
        __syncthreads();
        staged_rows<G_NT>(a, Tt, PT, RP, r0, GS_BMT, m0, n, t0, ncols, Ss + 0, V, 0);
    }
    if (a.stats_part) {
        __syncthreads();
        if (tid < 2 * GS_BMT) {
            int stt = tid / GS_BMT, row = tid - stt * GS_BMT;
            int m = m0 + row;
            if (m < a.M) a.stats_part[((long long)stt * a.stats_ctot + a.stats_coff + m) * a.nparts + g] = Ss[stt * 4 * BM + row];
        }
    }
    TG_T(tg1); TG_ACC(7, tg1 - tg0); TG_ACC(8, tg1 - tt0); TG_ACC(9, 1);
}

template <int NSRC, int BK>
constexpr size_t glds_lds_bytes(int K, bool pro = true) {
    return sizeof(float) * ((size_t)G_NST * (BK * G_PBMAX * NSRC + BK * G_PA) + (pro ? 3 * (size_t)K : 0) + 2 * 4 * BM);
}

struct ConvPlan { int BT, CW, TIN, lstride, sB, LB, pitchB, ntt, bk, mt, cwt, Vs, Vp, nsl; bool vec, flat; size_t lds; };

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static int plan_conv(const tamgcn_conv_desc* d, ConvPlan* p) {
    const int V = d->V;
    if (V < 1 || V > MAXCOLS) return -1;
    // contiguous 1x1 form: a channel row of the tile is one run of BT*V floats in HBM, whatever V is
    // (the INPUT side: an output written to every ostride-th frame -- the data gradient of a strided 1x1 conv -- only changes
    // the epilogue's addresses, staged_rows handles it)
    p->flat = d->KT == 1 && d->stride == 1 && d->up == 1 && d->pad == 0 && d->T_in == d->T_out;
    // joint slices for large skeletons with a temporal halo (V = 64: 8 halo frames are 512 floats per row, the tile 320)
    p->Vs = V; p->nsl = 1;
    if (!p->flat && d->KT > 1 && V > 32 && V % 16 == 0) { p->Vs = 16; p->nsl = V / 16; }
    p->Vp = p->flat ? V : ((p->Vs + 3) & ~3);
    const bool vec_try = d->K <= 1024 && (p->flat || (p->Vp & 3) == 0);
    int BT = MAXCOLS / p->Vs; if (BT < 1) BT = 1;
    if (BT > d->T_out) BT = d->T_out;
    const int q = 4 / gcd_i(V, 4);                          // flat tiles of V % 4 != 0: BT*V must be a multiple of 4
    if (p->flat && q > 1 && BT >= q) BT -= BT % q;
    for (;;) {
        p->BT = BT;
        p->lstride = (d->KT == 1) ? d->stride : 1;
        p->sB = (d->KT == 1) ? 1 : d->stride;
        p->TIN = (d->KT == 1) ? BT : (BT - 1) * d->stride + (d->KT - 1) * d->dil + 1;
        p->LB = p->TIN * p->Vp;
        int pitch = p->LB;
        pitch += ((16 - (pitch & 31)) + 32) & 31;          // pitch == 16 (mod 32)
        p->pitchB = pitch;
        p->CW = ceil_div(ceil_div(BT * p->Vs, 16), 4);
        p->vec = false; p->bk = BK;
        bool slots_short = false;
        if (vec_try && (p->LB & 3) == 0) {
            // BK = 32 when the activation chunk fits 10 float4 per thread, else 16 (8 per thread)
            int lb4 = p->LB / 4;
            int bk = (d->KT == 1 && 32 * lb4 <= 10 * NTHREADS) ? 32 : 16;
            if (bk * lb4 <= ((bk == 32) ? 10 : 8) * NTHREADS) { p->vec = true; p->bk = bk; }
            else slots_short = true;
        }
        // output-channel tile: 16*mt rows, mt chosen so that M splits without a half-empty tile
        p->mt = d->M <= 16 ? 1 : d->M <= 32 ? 2 : (d->M % 64 == 0 ? 4 : (d->M % 48 == 0 ? 3 : 4));
        // k x 1 kernels: a 64-row tile is 1600 dependent MFMAs per wave and one workgroup per CU (the launch is a chain of
        // latencies, not a throughput problem: 52 us for ONE workgroup at C = 64, tools/tconv_scaling.py); two 32-row tiles per
        // 64 channels put two workgroups on a CU: 52 -> 35 us per chain, 57.5 -> 54.5 us at 256 clips, +0.6 % on the step
        if (d->KT > 1 && p->mt == 4) p->mt = 2;
        p->cwt = p->CW <= 3 ? 3 : 5;
        if (p->vec) {                                       // staged epilogue needs >= 16 rows of (cwt*64+4) floats
            size_t avail = (((size_t)d->KT * p->mt * 16 * (p->bk + 2) + 3) & ~(size_t)3) + (size_t)p->bk * pitch;
            if (avail < (size_t)16 * (p->cwt * 64 + 4)) p->vec = false;
        }
        p->lds = p->vec ? sizeof(float) * ((((size_t)d->KT * p->mt * 16 * (p->bk + 2) + 3) & ~(size_t)3) + (size_t)p->bk * pitch +
                                           2 * 4 * BM + 3 * (size_t)d->K)
                        : sizeof(float) * ((size_t)d->KT * BM * (BK + 1) + (size_t)BK * pitch + 2 * 4 * BM);
        // a tile whose line buffer (frames + halo) exceeds the prefetch slots: fewer frames per tile rather than the scalar kernel
        if (slots_short && BT > 1) { BT = BT > 4 ? BT - 2 : BT - 1; if (p->flat && q > 1 && BT >= q) BT -= BT % q; continue; }
        if (p->lds <= 64 * 1024 || BT == 1) break;
        BT = (BT + 1) / 2;
        if (p->flat && q > 1 && BT >= q) BT -= BT % q;
    }
    if (!p->vec) { p->Vs = V; p->nsl = 1; if (p->Vp != V) {           // scalar kernel: whole frames, pitch V
        p->Vp = V; p->LB = p->TIN * V; int pitch = p->LB; pitch += ((16 - (pitch & 31)) + 32) & 31; p->pitchB = pitch;
        p->CW = ceil_div(ceil_div(p->BT * V, 16), 4);
        p->lds = sizeof(float) * ((size_t)d->KT * BM * (BK + 1) + (size_t)BK * pitch + 2 * 4 * BM); } }
    if (p->lds > 160 * 1024 || p->CW > MAXCW) return -1;
    p->ntt = ceil_div(d->T_out, p->BT);
    return 0;
}

}  // namespace


extern "C" int tamgcn_conv_nparts(const tamgcn_conv_desc* d) {
    ConvPlan p;
    if (!d || plan_conv(d, &p)) return -1;
    return d->N * p.ntt * p.nsl;
}

extern "C" int tamgcn_conv(const tamgcn_conv_desc* d, void* stream) {
    TG_CHECK(d && d->src.x1 && d->w && d->y, "tamgcn_conv: null pointer");
    TG_CHECK(d->N > 0 && d->K > 0 && d->M > 0 && d->T_in > 0 && d->T_out > 0 && d->V > 0,
             "tamgcn_conv: bad dims N=%d K=%d M=%d T_in=%d T_out=%d V=%d", d->N, d->K, d->M, d->T_in, d->T_out, d->V);
    TG_CHECK(d->KT >= 1 && d->KT <= 9 && d->dil >= 1 && d->stride >= 1 && d->up >= 1 && d->ostride >= 1,
             "tamgcn_conv: bad taps KT=%d dil=%d stride=%d up=%d", d->KT, d->dil, d->stride, d->up);
    TG_CHECK(d->src.coff + d->K <= d->src.ctot && d->ycoff + d->M <= d->yctot, "tamgcn_conv: channel slice out of range");
    TG_CHECK((d->T_out - 1) * d->ostride < d->T_y, "tamgcn_conv: T_out*ostride exceeds T_y");
    TG_CHECK(!(d->N > 65535), "tamgcn_conv: N too large for grid.z");
    TG_CHECK((long long)d->N * d->yctot * d->T_y * d->V < (1LL << 32), "tamgcn_conv: output tensor exceeds 2^32 elements");
    ConvPlan p;
    TG_CHECK(plan_conv(d, &p) == 0, "tamgcn_conv: no tiling for V=%d KT=%d dil=%d stride=%d", d->V, d->KT, d->dil, d->stride);
    ConvArgs a;
    a.src = make_src(d->src);
    a.N = d->N; a.K = d->K; a.T_in = d->T_in; a.V = d->V;
    a.w = d->w; a.bias = d->bias; a.M = d->M; a.KT = d->KT; a.dil = d->dil; a.stride = d->stride; a.pad = d->pad;
    if (d->wmode == 0) { a.ws_m = (long long)d->K * d->KT; a.ws_k = d->KT; a.ws_t = 1; a.w_off = 0; }
    else { a.ws_m = d->KT; a.ws_k = (long long)d->M * d->KT; a.ws_t = -1; a.w_off = d->KT - 1; }
    a.up = d->up; a.wmode = d->wmode;
    a.y = d->y; a.yctot = d->yctot; a.ycoff = d->ycoff; a.T_out = d->T_out; a.T_y = d->T_y; a.ostride = d->ostride;
    a.add1 = d->add1; a.add2 = d->add2; a.bcast = d->bcast; a.bcast_scale = d->bcast_scale;
    a.has_mask = d->mask != nullptr; a.mask = d->mask ? make_src(*d->mask) : null_src();
    a.aux = d->aux; a.aux_center = d->aux_center; a.auxctot = d->auxctot; a.auxcoff = d->auxcoff;
    TG_CHECK(!d->aux || d->aux_center, "tamgcn_conv: aux needs aux_center");
    a.stats_part = d->stats_part; a.stats_ctot = d->stats_ctot; a.stats_coff = d->stats_coff;
    a.post_coef = d->post_coef; a.post_ctot = d->post_ctot; a.post_act = d->post_act;
    TG_CHECK(!d->post_coef || d->post_ctot >= d->ycoff + d->M, "tamgcn_conv: post_coef has %d channels, need %d", d->post_ctot, d->ycoff + d->M);
    a.nparts = d->N * p.ntt * p.nsl;
    a.BT = p.BT; a.CW = p.CW; a.TIN = p.TIN; a.lstride = p.lstride; a.sB = p.sB; a.LB = p.LB; a.pitchB = p.pitchB;
    a.Vs = p.Vs; a.Vp = p.Vp; a.nsl = p.nsl;
    dim3 grid(p.ntt, ceil_div(d->M, BM), d->N);
    // 1x1 stride-1 convs whose rows are plain contiguous column ranges go to the LDS-DMA GEMM
    const bool aligned16 = (((uintptr_t)d->src.x1 | (uintptr_t)(d->src.x2 ? d->src.x2 : d->src.x1) | (uintptr_t)d->w) & 15) == 0;
    // ... and so do 1x1 convs with a temporal stride when V % 4 == 0 (round 4: they were 0.4 ms/step on the register-staged kernel)
    const bool strided1x1 = d->KT == 1 && d->stride == 2 && d->up == 1 && d->pad == 0 && (d->V & 3) == 0 && d->T_out == (d->T_in - 1) / 2 + 1 &&
                            p.lstride == 2 && p.TIN == p.BT && p.nsl == 1 && p.Vp == d->V && !d->mask && !d->aux;
    const bool glds = p.vec && (p.flat || strided1x1) && d->K % 16 == 0 && p.LB >= 64 && p.LB <= G_PBMAX && (p.LB & 3) == 0 &&
                      aligned16 && (long long)d->K * d->T_in * d->V < (1LL << 30);
    // data gradients into >= 128 channels: 128-row tiles on the bf16 matrix cores, operands split in registers into a
    // two-term bf16 pair (linear prologue only: it folds into the weights); no moments
    const size_t lds_split = sizeof(float) * ((size_t)GS_NST * (GS_BK * G_PBMAX + GS_BK * 128) + 3 * (size_t)d->K + 2 * 4 * BM + 128);
    const bool big = glds && tamgcn_split_mode() >= 1 && d->src.act == 0 && d->M >= 128 && d->M % 4 == 0 && d->K % GS_BK == 0 &&
                     lds_split <= 160 * 1024 && !d->post_coef && d->wmode == 1 && !d->stats_part;
    if (big) {
        const int nmt = ceil_div(d->M, 128);
        const unsigned nblk = (unsigned)(d->N * p.ntt * nmt);
        static tg_devmask fs = 0;
        tg_allow_lds((const void*)conv1x1_glds_split_kernel<2, false, 4>, 160 * 1024, &fs);
        hipLaunchKernelGGL((conv1x1_glds_split_kernel<2, false, 4>), dim3(nblk), dim3(G_NT), lds_split, (hipStream_t)stream, a, p.ntt, nmt);
        tamgcn_note_kernel("conv1x1_glds_split_kernel<2, false, 4>");
    } else if (glds) {
        const int nmt = ceil_div(d->M, G_BMT);
        const unsigned nblk = (unsigned)(d->N * p.ntt * nmt);
        // (Round 4 also built this kernel with an operand-swapped product and a register epilogue, VERDICT r03 item 4: forward
        // convs +-1 %, two-source data gradients 10-17 % slower, K >= 384 dx GEMMs 5 % faster -- profiles/r04_conv_swap_ab.txt.
        // Removed again when the kernel went to four row tiles per wave: 80 accumulator registers leave no room for it.)
        static int nw_env = -1;
        if (nw_env < 0) { const char* e = getenv("TAMGCN_CONV_WAVES"); nw_env = e ? atoi(e) : 8; }
        const bool pro = d->src.coef || d->src.x2 || d->src.act;
#define TG_GLDS_CASE(NS_, BK_, PRO_, NW_)                                                                                 \
        {                                                                                                                   \
            static tg_devmask fl = 0;                                                                                       \
            const size_t lds = glds_lds_bytes<NS_, BK_>(d->K, PRO_);                                                        \
            tg_allow_lds((const void*)conv1x1_glds_kernel<NS_, BK_, PRO_, NW_>, 160 * 1024, &fl);                           \
            hipLaunchKernelGGL((conv1x1_glds_kernel<NS_, BK_, PRO_, NW_>), dim3(nblk), dim3(NW_ * 64), lds, (hipStream_t)stream, a, p.ntt, nmt); \
            tamgcn_note_kernel("conv1x1_glds_kernel<%d, %d, %s, %d>", NS_, BK_, PRO_ ? "true" : "false", NW_);              \
        }
        if (nw_env != 4) {
            if (d->src.x2) TG_GLDS_CASE(2, 8, true, 8) else if (pro) TG_GLDS_CASE(1, 16, true, 8) else TG_GLDS_CASE(1, 16, false, 8)
        } else {
            if (d->src.x2) TG_GLDS_CASE(2, 8, true, 4) else if (pro) TG_GLDS_CASE(1, 16, true, 4) else TG_GLDS_CASE(1, 16, false, 4)
        }
#undef TG_GLDS_CASE
    } else if (p.vec) {
        dim3 gridv(p.ntt * p.nsl, ceil_div(d->M, 16 * p.mt), d->N);
#define TG_CONV_CASE(BKV_, MT_, CW_)                                                                              \
        if (p.bk == BKV_ && p.mt == MT_ && p.cwt == CW_) {                                                         \
            if (p.lds > 64 * 1024)                                                                                 \
                (void)hipFuncSetAttribute((const void*)conv_kernel_vec<BKV_, MT_, CW_>,                            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);                 \
            hipLaunchKernelGGL((conv_kernel_vec<BKV_, MT_, CW_>), gridv, dim3(NTHREADS), p.lds, (hipStream_t)stream, a); \
            tamgcn_note_kernel("conv_kernel_vec<%d, %d, %d>", BKV_, MT_, CW_);                                      \
        } else
        TG_CONV_CASE(32, 4, 5) TG_CONV_CASE(32, 3, 5) TG_CONV_CASE(32, 2, 5) TG_CONV_CASE(32, 1, 5)
        TG_CONV_CASE(32, 4, 3) TG_CONV_CASE(32, 3, 3) TG_CONV_CASE(32, 2, 3) TG_CONV_CASE(32, 1, 3)
        TG_CONV_CASE(16, 4, 5) TG_CONV_CASE(16, 3, 5) TG_CONV_CASE(16, 2, 5) TG_CONV_CASE(16, 1, 5)
        TG_CONV_CASE(16, 4, 3) TG_CONV_CASE(16, 3, 3) TG_CONV_CASE(16, 2, 3) TG_CONV_CASE(16, 1, 3)
        { tamgcn_set_error("tamgcn_conv: no instantiation bk=%d mt=%d cw=%d", p.bk, p.mt, p.cwt); return -1; }
#undef TG_CONV_CASE
    } else {
        if (p.lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        hipLaunchKernelGGL(conv_kernel, grid, dim3(NTHREADS), p.lds, (hipStream_t)stream, a);
        tamgcn_note_kernel("conv_kernel");
    }
    TG_LAUNCH_CHECK("tamgcn_conv");
    return 0;
}

// ===========================================================================
// weight gradient:  dW[m][k][tap] = sum_{n,t,v} gy(n,m,t,v) * x(n,k,t*stride + tap*dil - pad, v)
//
// An "NT" GEMM whose contraction index p = (n,t,v) is the contiguous axis of BOTH operands.
// A workgroup owns a (2*WMT*16) x (2*WKT*16) tile of dW for all taps (2x2 waves, one
// sub-tile per wave, accumulators never leave registers), walks its share of the samples in
// chunks of BT frames, stages both operand tiles in LDS with 16-byte global loads (rows are
// contiguous along t*V+v), the BatchNorm(-backward)-apply prologue fused into the fill, and
// feeds v_mfma_f32_16x16x4_f32 with column reads that are bank-conflict free for pitches
// == 2 (mod 4).  Partial slabs per n-split are summed by reduce_sum (deterministic).
// ===========================================================================
namespace {

struct WgradArgs {
    SrcDev gy, src;
    int N, M, K, T_in, T_out, V, dil, stride, pad;
    float* part; int nsplit;
    int BT, TIN;      // frames per staged chunk (gy side / x side)
    int PY, PX;       // LDS pitches
    int n_per;        // samples per split
    int KTG;          // LDS-DMA kernel: temporal taps (1 = 1x1), one window of the contraction axis per tap
};

__device__ __forceinline__ float wg_apply(float x1, float x2, float c1, float c2, float c0, int act) {
    float v = fmaf(c1, x1, fmaf(c2, x2, c0));
    return act == 1 ? fmaxf(v, 0.f) : v;
}

// fill rows [r0, r0+rows) x [0, len) of an LDS tile from channel-rows of `s`; frames outside
// [0, T) are zero.  f0 = first frame of the tile, V4 = V/4 (VEC) .
template <bool VEC>
__device__ __forceinline__ void wg_fill(float* tile, int pitch, const float* cf, int rows, int nvalid, int ch0,
                                        const SrcDev& s, long long nbase, long long cs, int T, int V, int f0, int frames) {
    const int tid = threadIdx.x;
    const int len = frames * V;
    if constexpr (VEC) {
        const int l4 = len >> 2;
        for (int e = tid; e < rows * l4; e += NTHREADS) {
            int r = e / l4, c4 = e - r * l4;
            int col = c4 << 2;
            int fr = f0 + col / V;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nvalid && fr >= 0 && fr < T) {
                long long g = nbase + (long long)(ch0 + r) * cs + (long long)f0 * V + col;
                float4 a = *reinterpret_cast<const float4*>(s.x1 + g);
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s.x2) b = *reinterpret_cast<const float4*>(s.x2 + g);
                float c1 = cf[r], c2 = cf[rows + r], c0 = cf[2 * rows + r];
                o.x = wg_apply(a.x, b.x, c1, c2, c0, s.act); o.y = wg_apply(a.y, b.y, c1, c2, c0, s.act);
                o.z = wg_apply(a.z, b.z, c1, c2, c0, s.act); o.w = wg_apply(a.w, b.w, c1, c2, c0, s.act);
            }
            float2* d = reinterpret_cast<float2*>(tile + r * pitch + col);
            d[0] = make_float2(o.x, o.y);
            d[1] = make_float2(o.z, o.w);
        }
    } else {
        // V % 4 != 0 (NTU's 25 joints at stride 2): one float per step.  Round 4: the row / frame indices come from reciprocals
        // (two integer divisions per element were ~50 VALU instructions) and four steps' loads are requested together.
        const bool rcp_ok = len >= 4 && len <= 1024 && V >= 4 && rows * len < (1 << 20);
        const float rlen = 1.0f / (float)len, rV = 1.0f / (float)V;
        for (int e0 = tid; e0 < rows * len; e0 += 4 * NTHREADS) {
            float v1[4], v2[4]; int rr[4], cc[4]; bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + i * NTHREADS;
                const bool in = e < rows * len;
                const int ee = in ? e : 0;
                const int r = rcp_ok ? tc_like_div(ee, rlen) : ee / len, col = ee - r * len;
                const int fr = f0 + (rcp_ok ? tc_like_div(col, rV) : col / V);
                rr[i] = r; cc[i] = col;
                ok[i] = in && r < nvalid && fr >= 0 && fr < T;
                const long long g = nbase + (long long)(ch0 + (ok[i] ? r : 0)) * cs + (long long)f0 * V + (ok[i] ? col : 0);
                v1[i] = ok[i] ? s.x1[g] : 0.f;
                v2[i] = (ok[i] && s.x2) ? s.x2[g] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + i * NTHREADS;
                if (e < rows * len) tile[rr[i] * pitch + cc[i]] = ok[i] ? wg_apply(v1[i], v2[i], cf[rr[i]], cf[rows + rr[i]], cf[2 * rows + rr[i]], s.act) : 0.f;
            }
        }
    }
}

constexpr int WG_NPF = 6;     // float4 prefetch slots per thread and operand (1x1 weight-gradient pipeline)

// PS ("p-split", M, K <= 16: the 16-channel temporal branches): the tile is ONE 16x16 MFMA tile; instead of tiling
// (M, K) 2x2 -- three of four waves would idle -- the four waves share it and split the chunk's frames, partial
// accumulators meet in LDS at the end.
template <int KT, int WMT, int WKT, bool VEC, bool PS = false>
__global__ __launch_bounds__(NTHREADS) void wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BMW = PS ? 16 : 2 * WMT * 16, BKW = PS ? 16 : 2 * WKT * 16;
    constexpr bool PF = VEC && (KT == 1 || PS);   // register-prefetch pipeline (host guarantees the slot bound)
    float* Ys = smem;                         // [BMW][PY]
    float* Xs = Ys + BMW * a.PY;              // [BKW][PX]
    float* cfY = Xs + BKW * a.PX;             // [3][BMW]
    float* cfX = cfY + 3 * BMW;               // [3][BKW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wr = PS ? 0 : wave >> 1, wc = PS ? 0 : wave & 1;
    const int k0 = blockIdx.x * BKW, m0 = blockIdx.y * BMW, split = blockIdx.z;
    const int V = a.V, V4 = (V + 3) >> 2;
    // a split owns a contiguous range of the (sample, frame chunk) sequence: splits may be finer than samples
    const int cpt = (a.T_out + a.BT - 1) / a.BT;              // frame chunks per sample
    const int c_begin = split * a.n_per, c_end = min(a.N * cpt, c_begin + a.n_per);
    const int mvalid = min(BMW, a.M - m0), kvalid = min(BKW, a.K - k0);

    for (int e = tid; e < BMW; e += NTHREADS) {
        int ch = a.gy.coff + m0 + e;
        bool ok = e < mvalid && a.gy.coef;
        cfY[e] = ok ? a.gy.coef[ch] : 1.f;
        cfY[BMW + e] = (ok && a.gy.x2) ? a.gy.coef[a.gy.ctot + ch] : 0.f;
        cfY[2 * BMW + e] = ok ? a.gy.coef[2 * a.gy.ctot + ch] : 0.f;
    }
    for (int e = tid; e < BKW; e += NTHREADS) {
        int ch = a.src.coff + k0 + e;
        bool ok = e < kvalid && a.src.coef;
        cfX[e] = ok ? a.src.coef[ch] : 1.f;
        cfX[BKW + e] = (ok && a.src.x2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cfX[2 * BKW + e] = ok ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }

    f32x4 acc[KT][WMT][WKT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int x = 0; x < WMT; ++x)
#pragma unroll
            for (int y = 0; y < WKT; ++y) acc[t][x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long long gy_cs = (long long)a.T_out * V, x_cs = (long long)a.T_in * V;
    const float* yrow = Ys + (wr * WMT * 16 + j) * a.PY;
    const float* xrow = Xs + (wc * WKT * 16 + j) * a.PX;

    // ---- prefetch descriptors (PF only): slot i of this thread -> (row, 4-column group) of each tile
    const int ly4 = (a.BT * V) >> 2, lx4 = (a.TIN * V) >> 2;
    int yr[PF ? WG_NPF : 1], yc[PF ? WG_NPF : 1], xr_[PF ? WG_NPF : 1], xc[PF ? WG_NPF : 1];
    // frame (inside the chunk) of a slot's first element and how many of its 4 elements lie in that frame: for V % 4 != 0
    // (rows only dword aligned -- gfx950 takes such 16-byte loads at full rate) a slot may straddle two frames
    int yf[PF ? WG_NPF : 1], ycn[PF ? WG_NPF : 1], xf[PF ? WG_NPF : 1], xcn[PF ? WG_NPF : 1];
    float4 y1[PF ? WG_NPF : 1], y2[PF ? WG_NPF : 1], x1[PF ? WG_NPF : 1], x2[PF ? WG_NPF : 1];
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < WG_NPF; ++i) {
            int e = tid + i * NTHREADS;
            int r = e / ly4; yr[i] = r < BMW ? r : -1; yc[i] = (e - r * ly4) << 2;
            r = e / lx4; xr_[i] = r < BKW ? r : -1; xc[i] = (e - r * lx4) << 2;
            yf[i] = yc[i] / V; ycn[i] = min(4, V - (yc[i] - yf[i] * V));
            xf[i] = xc[i] / V; xcn[i] = min(4, V - (xc[i] - xf[i] * V));
        }
    }
    const bool gy2 = a.gy.x2 != nullptr, sx2 = a.src.x2 != nullptr;
    // elements [0, cn) of a slot are valid iff va, elements [cn, 4) iff vb: a whole-slot 16-byte load when both hold,
    // element loads for a slot that is cut by the start or the end of the row (never touches memory outside the row)
    auto load_slot = [&](const float* p, int cn, bool va, bool vb) -> float4 {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (va && vb) o = *reinterpret_cast<const float4*>(p);
        else if (va || vb) {
            if (cn > 0 ? va : vb) o.x = p[0];
            if (cn > 1 ? va : vb) o.y = p[1];
            if (cn > 2 ? va : vb) o.z = p[2];
            if (cn > 3 ? va : vb) o.w = p[3];
        }
        return o;
    };
    auto prefetch = [&](int n, int t0) {
        if constexpr (PF) {
            const long long yb = (long long)n * a.gy.ctot * gy_cs + (long long)(a.gy.coff + m0) * gy_cs + (long long)t0 * V;
            const int f0 = t0 * a.stride - a.pad;
            const long long xb = (long long)n * a.src.ctot * x_cs + (long long)(a.src.coff + k0) * x_cs + (long long)f0 * V;
#pragma unroll
            for (int i = 0; i < WG_NPF; ++i) {
                y1[i] = make_float4(0.f, 0.f, 0.f, 0.f); y2[i] = y1[i]; x1[i] = y1[i]; x2[i] = y1[i];
                if (yr[i] >= 0 && yr[i] < mvalid) {
                    const int fa = t0 + yf[i], fb = fa + (ycn[i] < 4 ? 1 : 0);
                    const bool va = fa < a.T_out, vb = fb < a.T_out;
                    long long g = yb + (long long)yr[i] * gy_cs + yc[i];
                    y1[i] = load_slot(a.gy.x1 + g, ycn[i], va, vb);
                    if (gy2) y2[i] = load_slot(a.gy.x2 + g, ycn[i], va, vb);
                }
                if (xr_[i] >= 0 && xr_[i] < kvalid) {
                    const int fa = f0 + xf[i], fb = fa + (xcn[i] < 4 ? 1 : 0);
                    const bool va = fa >= 0 && fa < a.T_in, vb = fb >= 0 && fb < a.T_in;
                    long long g = xb + (long long)xr_[i] * x_cs + xc[i];
                    x1[i] = load_slot(a.src.x1 + g, xcn[i], va, vb);
                    if (sx2) x2[i] = load_slot(a.src.x2 + g, xcn[i], va, vb);
                }
            }
        }
    };
    auto commit = [&](int n, int t0) {        // prologue + LDS store of the prefetched chunk (zeros where invalid)
        if constexpr (PF) {
            const int f0 = t0 * a.stride - a.pad;
#pragma unroll
            for (int i = 0; i < WG_NPF; ++i) {
                if (yr[i] >= 0) {
                    const int r = yr[i];
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int fa = t0 + yf[i], fb = fa + (ycn[i] < 4 ? 1 : 0);
                    const bool va = fa < a.T_out, vb = fb < a.T_out;
                    if (r < mvalid && (va || vb)) {
                        float c1 = cfY[r], c2 = cfY[BMW + r], c0 = cfY[2 * BMW + r];
                        o.x = wg_apply(y1[i].x, y2[i].x, c1, c2, c0, a.gy.act); o.y = wg_apply(y1[i].y, y2[i].y, c1, c2, c0, a.gy.act);
                        o.z = wg_apply(y1[i].z, y2[i].z, c1, c2, c0, a.gy.act); o.w = wg_apply(y1[i].w, y2[i].w, c1, c2, c0, a.gy.act);
                        if (!(va && vb)) {                        // a slot cut by the end of the row: padding is zero AFTER the prologue
                            const int cn = ycn[i];
                            if (!(cn > 0 ? va : vb)) o.x = 0.f;
                            if (!(cn > 1 ? va : vb)) o.y = 0.f;
                            if (!(cn > 2 ? va : vb)) o.z = 0.f;
                            if (!(cn > 3 ? va : vb)) o.w = 0.f;
                        }
                    }
                    float2* d = reinterpret_cast<float2*>(Ys + r * a.PY + yc[i]);
                    d[0] = make_float2(o.x, o.y); d[1] = make_float2(o.z, o.w);
                }
                if (xr_[i] >= 0) {
                    const int r = xr_[i];
                    const int fa = f0 + xf[i], fb = fa + (xcn[i] < 4 ? 1 : 0);
                    const bool va = fa >= 0 && fa < a.T_in, vb = fb >= 0 && fb < a.T_in;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r < kvalid && (va || vb)) {
                        float c1 = cfX[r], c2 = cfX[BKW + r], c0 = cfX[2 * BKW + r];
                        o.x = wg_apply(x1[i].x, x2[i].x, c1, c2, c0, a.src.act); o.y = wg_apply(x1[i].y, x2[i].y, c1, c2, c0, a.src.act);
                        o.z = wg_apply(x1[i].z, x2[i].z, c1, c2, c0, a.src.act); o.w = wg_apply(x1[i].w, x2[i].w, c1, c2, c0, a.src.act);
                        if (!(va && vb)) {
                            const int cn = xcn[i];
                            if (!(cn > 0 ? va : vb)) o.x = 0.f;
                            if (!(cn > 1 ? va : vb)) o.y = 0.f;
                            if (!(cn > 2 ? va : vb)) o.z = 0.f;
                            if (!(cn > 3 ? va : vb)) o.w = 0.f;
                        }
                    }
                    float2* d = reinterpret_cast<float2*>(Xs + r * a.PX + xc[i]);
                    d[0] = make_float2(o.x, o.y); d[1] = make_float2(o.z, o.w);
                }
            }
        }
    };

    __syncthreads();                          // coefficient tables visible
    if (c_begin < c_end) prefetch(c_begin / cpt, (c_begin % cpt) * a.BT);
    for (int ci = c_begin; ci < c_end; ++ci) {
        {
            const int n = ci / cpt, t0 = (ci - n * cpt) * a.BT;
            const int bt = min(a.BT, a.T_out - t0);
            const int tin = (bt - 1) * a.stride + (KT - 1) * a.dil + 1;
            __syncthreads();
            if constexpr (PF) {
                commit(n, t0);
            } else {
                wg_fill<VEC>(Ys, a.PY, cfY, BMW, mvalid, a.gy.coff + m0, a.gy, (long long)n * a.gy.ctot * gy_cs, gy_cs,
                             a.T_out, V, t0, bt);
                wg_fill<VEC>(Xs, a.PX, cfX, BKW, kvalid, a.src.coff + k0, a.src, (long long)n * a.src.ctot * x_cs, x_cs,
                             a.T_in, V, t0 * a.stride - a.pad, tin);
            }
            __syncthreads();
            if constexpr (PF) {               // next chunk's loads fly under this chunk's MFMAs
                if (ci + 1 < c_end) prefetch((ci + 1) / cpt, ((ci + 1) % cpt) * a.BT);
            }
            for (int tl = PS ? wave : 0; tl < bt; tl += PS ? 4 : 1) {
#pragma unroll 5
                for (int v4 = 0; v4 < V4; ++v4) {
                    int v = v4 * 4 + kq;
                    bool vok = v < V;
                    int vc = vok ? v : 0;
                    float av[WMT];
#pragma unroll
                    for (int x = 0; x < WMT; ++x) {
                        float t = yrow[x * 16 * a.PY + tl * V + vc];
                        av[x] = vok ? t : 0.f;
                    }
#pragma unroll
                    for (int tap = 0; tap < KT; ++tap) {
                        const int xo = (tl * a.stride + tap * a.dil) * V + vc;
#pragma unroll
                        for (int y = 0; y < WKT; ++y) {
                            float bv = xrow[y * 16 * a.PX + xo];
#pragma unroll
                            for (int x = 0; x < WMT; ++x) acc[tap][x][y] = mfma16(av[x], bv, acc[tap][x][y]);
                        }
                    }
                }
            }
        }
    }
    float* out = a.part + (long long)split * a.M * a.K * KT;
    if constexpr (PS) {                       // sum the four waves' partial tiles through LDS (operand tiles are dead)
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < KT; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) smem[(wave * KT + tap) * 256 + lane * 4 + r] = acc[tap][0][0][r];
        __syncthreads();
        const int l = tid >> 2, r = tid & 3;
        const int m = m0 + (l >> 4) * 4 + r, k = k0 + (l & 15);
        if (m < a.M && k < a.K) {
#pragma unroll
            for (int tap = 0; tap < KT; ++tap) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) t += smem[(w * KT + tap) * 256 + tid];
                out[((long long)m * a.K + k) * KT + tap] = t;
            }
        }
        return;
    }
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
        for (int x = 0; x < WMT; ++x)
#pragma unroll
            for (int y = 0; y < WKT; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = m0 + (wr * WMT + x) * 16 + kq * 4 + r;
                    int k = k0 + (wc * WKT + y) * 16 + j;
                    if (m < a.M && k < a.K) out[((long long)m * a.K + k) * KT + tap] = acc[tap][x][y][r];
                }
}


// ===========================================================================
// 1x1 stride-1 weight gradient as an LDS-DMA "NT" GEMM:  dW[m][k] = sum_{n,p} gy(n,m,p) x(n,k,p),
// p = (t, v) contiguous in both operands.
//
//   workgroup   512 threads = 8 waves as 2 (m) x 4 (k); tile (32*WMT) x (64*WKT) of dW, one n-split
//   chunk       32 contraction indices; every operand row is 128 B = 8 slots of 16 B; one dwordx4
//               LDS-DMA piece carries 8 rows (1 KB, lane-linear).  Slot u of row r is stored at slot
//               u ^ (r & 7) (swizzle applied to the per-lane SOURCE address), which makes the ds_read_b128
//               fragment reads below bank-conflict free.
//   contraction MFMA step s of a 16-index block takes p = 4*kq + s: each lane covers four steps with one
//               16-byte read per operand tile (the same permutation on both operands).
//   prologue    per-ROW coefficients (BatchNorm(-backward) apply, two-source combine, ReLU) are lane
//               constants here (lane = row), applied to the fragment registers
//   pipeline    3-stage ring, two chunks in flight, counted vmcnt + one raw s_barrier per chunk
//   row tails   a row (T*V floats) that is not a multiple of 32 ends in a chunk that is fetched from [len - 32, len) --
//               no read past the row -- with the elements the previous chunk already covered zeroed in the gy fragment
//   k x 1       (stride 1, "same" padding) tap kt contracts gy[t] with x[t + kt*dil - pad]: the same GEMM over the
//               window of frames both sides have, i.e. two row offsets and a shorter row.  Taps are a grid axis
//               (blockIdx -> (tap, split, tile)); dW is written [m][k][kt].  Offsets are multiples of V floats:
//               gfx950 takes dword-aligned 16-byte DMA pieces at full rate (tools/probes/unaligned_probe.hip).
// ===========================================================================
constexpr int W_PC = 32, W_NST = 3, W_NT = 512;

template <int WMT, int WKT, int NY, int NX, bool SPL, int NST>
#ifndef TG_WKO
#define TG_WKO 0      // knock-out side builds of the weight-gradient kernel (tools/wgrad_knockout.py): 1 one MFMA in eight, 2 no DMA, 4 no fragment reads, 8 no barrier (profiles/r04_wgrad_knockout.txt)
#endif
__global__ __launch_bounds__(W_NT, 2) void wgrad_glds_kernel(const WgradArgs a, int ntk, int ntm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BMW = 2 * WMT * 16, BKW = 4 * WKT * 16;
    constexpr int RY = BMW * NY, RX = BKW * NX, ROWS = RY + RX;
    constexpr int STG = ROWS * W_PC;                              // floats per stage
    constexpr int NPIECE = ROWS / 8;                              // 1 KB pieces per chunk
    constexpr int MAXP = (NPIECE + 7) / 8;                        // per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4, wr = wave >> 2, wc = wave & 3;
    int split, tile, tap;
    {
        const int nt = ntk * ntm, per_tap = nt * a.nsplit;
        int L = blockIdx.x;
        tap = L / per_tap; L -= tap * per_tap;
        if ((a.nsplit & 7) == 0) { const int xcd = L & 7, i = L >> 3; tile = i % nt; split = (i / nt) * 8 + xcd; }
        else { split = L / nt; tile = L - split * nt; }
    }
    const int k0 = (tile % ntk) * BKW, m0 = (tile / ntk) * BMW;
    const long long cs = (long long)a.T_out * a.V;               // channel-row stride of gy (and of x at stride 1)
    const long long csx = (long long)a.T_in * a.V;               // ... of x
    const float rV = 1.0f / (float)a.V;
    // this tap's window: frames t with 0 <= t < T and 0 <= t + dlt < T
    const int dlt = a.KTG > 1 ? tap * a.dil - a.pad : 0;
    const int wlen = (a.T_out - (dlt < 0 ? -dlt : dlt)) * a.V;    // host: >= 2 * W_PC
    const long long ylo = dlt < 0 ? (long long)(-dlt) * a.V : 0, xlo = dlt > 0 ? (long long)dlt * a.V : 0;
    const int cpf = wlen / W_PC, ntail = wlen - cpf * W_PC;       // full chunks, elements of the overlapping last one
    const int cps = cpf + (ntail ? 1 : 0);                        // chunks per sample
    // a split owns a contiguous range of the (n, chunk) sequence: splits may be finer than samples
    const int n_per = (a.N * cps + a.nsplit - 1) / a.nsplit;
    const int c_begin = split * n_per, nch = max(0, min(a.N * cps, c_begin + n_per) - c_begin);

    // ---- DMA descriptors.  Stage rows: [Y1 | Y2 | X1 | X2]; lane -> (row in piece, physical slot)
    const int pr = lane >> 3, ps = lane & 7;
    const int pu = ps ^ pr;                                       // logical 16-byte slot this lane fetches
    const float* p_base[MAXP];
    bool p_ok[MAXP], p_on[MAXP], p_isy[MAXP];
    int p_dst[MAXP];
    int nissue = 0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int q = wave + i * 8;
        const bool qok = q < NPIECE;
        const int row0 = (qok ? q : 0) * 8;                       // first stage row of the piece
        const bool isy = row0 < RY;
        int img, r;                                               // image (0/1) and row inside the tile
        if (isy) { img = row0 / BMW; r = row0 - img * BMW + pr; }
        else { img = (row0 - RY) / BKW; r = row0 - RY - img * BKW + pr; }
        const SrcDev& sd = isy ? a.gy : a.src;
        const int ch = (isy ? m0 : k0) + r;
        p_ok[i] = qok && ch < (isy ? a.M : a.K);
        p_base[i] = (img == 0 ? sd.x1 : sd.x2) + (long long)(sd.coff + (p_ok[i] ? ch : 0)) * (isy ? cs : csx) + pu * 4;
        p_isy[i] = isy;
        p_dst[i] = row0 * W_PC;
        p_on[i] = __ballot(p_ok[i]) != 0ull;
        nissue += p_on[i] ? 1 : 0;
    }
    const long long ystep = (long long)a.gy.ctot * cs, xstep = (long long)a.src.ctot * csx;
    auto issue = [&](int c) {
        if (TG_WKO & 2) return;
        float* st = smem + (c % NST) * STG;
        const int gc = c_begin + c, nn = gc / cps, pc = gc - nn * cps;
        const int po = pc < cpf ? pc * W_PC : wlen - W_PC;        // the row's last, partial chunk: re-fetch the last 32
        const long long oy = (long long)nn * ystep + ylo + po;
        // stride 2: this lane's slot starts at p0 = po + 4 pu of frame p0 / V; its x elements sit one frame further per frame
        const long long ox = (long long)nn * xstep + xlo + po + (a.stride == 2 ? tc_like_div(po + pu * 4, rV) * a.V : 0);
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (p_on[i]) {
                const float* gp = p_base[i] + (p_isy[i] ? oy : ox);
                if (p_ok[i]) __builtin_amdgcn_global_load_lds((tg_gptr)gp, (tg_lptr)(st + p_dst[i]), 16, 0, 0);
            }
        }
    };

    // ---- per-lane row coefficients (lane j <-> row of each fragment tile)
    float cy1[WMT], cy2[WMT], cy0[WMT], cx1[WKT], cx2[WKT], cx0[WKT];
#pragma unroll
    for (int x = 0; x < WMT; ++x) {
        const int m = m0 + (wr * WMT + x) * 16 + j;
        const bool ok = m < a.M && a.gy.coef;
        const int ch = a.gy.coff + (m < a.M ? m : 0);
        cy1[x] = ok ? a.gy.coef[ch] : 1.f;
        cy2[x] = (ok && NY == 2) ? a.gy.coef[a.gy.ctot + ch] : 0.f;
        cy0[x] = ok ? a.gy.coef[2 * a.gy.ctot + ch] : 0.f;
    }
#pragma unroll
    for (int y = 0; y < WKT; ++y) {
        const int k = k0 + (wc * WKT + y) * 16 + j;
        const bool ok = k < a.K && a.src.coef;
        const int ch = a.src.coff + (k < a.K ? k : 0);
        cx1[y] = ok ? a.src.coef[ch] : 1.f;
        cx2[y] = (ok && NX == 2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cx0[y] = ok ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }
    const float loy = a.gy.act == 1 ? 0.f : -__builtin_inff(), lox = a.src.act == 1 ? 0.f : -__builtin_inff();
    const bool yplain = !a.gy.coef && a.gy.act != 1, xplain = !a.src.coef && a.src.act != 1;   // plain tensors skip the prologue

    f32x4 acc[WMT][WKT];
#pragma unroll
    for (int x = 0; x < WMT; ++x)
#pragma unroll
        for (int y = 0; y < WKT; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment offsets inside a stage (floats): row * 32 + 4 * ((4b + kq) ^ (row & 7)), row & 7 == j & 7
    const int yoff = ((wr * WMT) * 16 + j) * W_PC, xoff = (RY + (wc * WKT) * 16 + j) * W_PC;
    const int sl0 = ((kq) ^ (j & 7)) * 4, sl1 = ((4 + kq) ^ (j & 7)) * 4;

    // NST = 3: two chunks in flight; NST = 2 (tiles whose two-stage ring lets a second workgroup share the CU):
    // one chunk in flight per workgroup, the neighbour covers the wait
    if (nch > 0) issue(0);
    if (NST == 3 && nch > 1) issue(1);
    for (int c = 0; c < nch; ++c) {
        wait_vmcnt((NST == 3 && c + 1 < nch) ? nissue : 0);
        if (!(TG_WKO & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + NST - 1 < nch) issue(c + NST - 1);
        const float* st = smem + (c % NST) * STG;
        // one half-chunk's fragments (16 contraction indices): four k4 steps
        auto frag = [&](int b, f32x4 (&avb)[WMT], f32x4 (&bvb)[WKT]) {
            const int sl = b ? sl1 : sl0;
            if (TG_WKO & 4) {
#pragma unroll
                for (int x = 0; x < WMT; ++x) asm volatile("" : "=v"(avb[x]));
#pragma unroll
                for (int y = 0; y < WKT; ++y) asm volatile("" : "=v"(bvb[y]));
                return;
            }
#pragma unroll
            for (int x = 0; x < WMT; ++x) {
                f32x4 v = *reinterpret_cast<const f32x4*>(st + yoff + x * 16 * W_PC + sl);
                if constexpr (NY == 2) {
                    f32x4 v2 = *reinterpret_cast<const f32x4*>(st + BMW * W_PC + yoff + x * 16 * W_PC + sl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(cy1[x], v[e], fmaf(cy2[x], v2[e], cy0[x])), loy);
                } else if (!yplain) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(cy1[x], v[e], cy0[x]), loy);
                }
                avb[x] = v;
            }
#pragma unroll
            for (int y = 0; y < WKT; ++y) {
                f32x4 v = *reinterpret_cast<const f32x4*>(st + xoff + y * 16 * W_PC + sl);
                if constexpr (NX == 2) {
                    f32x4 v2 = *reinterpret_cast<const f32x4*>(st + BKW * W_PC + xoff + y * 16 * W_PC + sl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(cx1[y], v[e], fmaf(cx2[y], v2[e], cx0[y])), lox);
                } else if (!xplain) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(cx1[y], v[e], cx0[y]), lox);
                }
                bvb[y] = v;
            }
            if (ntail && (c_begin + c) % cps == cpf) {           // overlapping last chunk: keep only its last ntail elements
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * (4 * b + kq) + e < W_PC - ntail) {
#pragma unroll
                        for (int x = 0; x < WMT; ++x) avb[x][e] = 0.f;
                    }
            }
        };
        f32x4 av[2][WMT], bv[2][WKT];
        frag(0, av[0], bv[0]);
        frag(1, av[1], bv[1]);
        if constexpr (SPL) {                // the lane's eight contraction indices of this chunk = one K = 32 fragment
            bf16x8_t ah[WMT], al[WMT], bh[WKT], bl[WKT];
#pragma unroll
            for (int x = 0; x < WMT; ++x) split_bf16x8(av[0][x], av[1][x], ah[x], al[x]);
#pragma unroll
            for (int y = 0; y < WKT; ++y) split_bf16x8(bv[0][y], bv[1][y], bh[y], bl[y]);
#pragma unroll
            for (int y = 0; y < WKT; ++y)
#pragma unroll
                for (int x = 0; x < WMT; ++x) acc[x][y] = mfma_split(ah[x], al[x], bh[y], bl[y], acc[x][y]);
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < ((TG_WKO & 1) ? (b ? 0 : 1) : 4); ++e)
#pragma unroll
                    for (int y = 0; y < WKT; ++y)
#pragma unroll
                        for (int x = 0; x < WMT; ++x) acc[x][y] = mfma16(av[b][x][e], bv[b][y][e], acc[x][y]);
        }
    }
    float* out = a.part + (long long)split * a.M * a.K * a.KTG + tap;
#pragma unroll
    for (int x = 0; x < WMT; ++x)
#pragma unroll
        for (int y = 0; y < WKT; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wr * WMT + x) * 16 + kq * 4 + r;
                const int k = k0 + (wc * WKT + y) * 16 + j;
                if (m < a.M && k < a.K) out[((long long)m * a.K + k) * a.KTG] = acc[x][y][r];
            }
}

template <int WMT, int WKT, int NY, int NX, bool SPL>
static int launch_wgrad_glds(WgradArgs& a, hipStream_t s) {
    constexpr int BMW = 2 * WMT * 16, BKW = 4 * WKT * 16;
    constexpr size_t STAGE = sizeof(float) * (size_t)(BMW * NY + BKW * NX) * W_PC;
    // a third stage only where it does not cost the second workgroup per CU
    constexpr int NST = (2 * STAGE <= 80 * 1024 && 3 * STAGE > 80 * 1024) ? 2 : W_NST;
    const size_t lds = NST * STAGE;
    static tg_devmask flag = 0;
    tg_allow_lds((const void*)wgrad_glds_kernel<WMT, WKT, NY, NX, SPL, NST>, 160 * 1024, &flag);
    const int ntk = ceil_div(a.K, BKW), ntm = ceil_div(a.M, BMW);
    hipLaunchKernelGGL((wgrad_glds_kernel<WMT, WKT, NY, NX, SPL, NST>), dim3((unsigned)(ntk * ntm * a.nsplit * a.KTG)), dim3(W_NT), lds, s, a, ntk, ntm);
    tamgcn_note_kernel("wgrad_glds_kernel<%d, %d, %d, %d, %s, %d>%s", WMT, WKT, NY, NX, SPL ? "split" : "f32", NST, a.KTG > 1 ? " taps" : "");
    return 0;
}

template <int WMT, int WKT, bool SPL>
static int launch_wgrad_glds_spl(WgradArgs& a, hipStream_t s) {
    const bool y2 = a.gy.x2 != nullptr, x2 = a.src.x2 != nullptr;
    if (y2 && x2) return launch_wgrad_glds<WMT, WKT, 2, 2, SPL>(a, s);
    if (y2) return launch_wgrad_glds<WMT, WKT, 2, 1, SPL>(a, s);
    if (x2) return launch_wgrad_glds<WMT, WKT, 1, 2, SPL>(a, s);
    return launch_wgrad_glds<WMT, WKT, 1, 1, SPL>(a, s);
}

// split-fp32 MFMA (TAMGCN_SPLIT_BF16 >= 1, the default) or the exact fp32-input MFMA (0)
template <int WMT, int WKT>
static int launch_wgrad_glds_src(WgradArgs& a, hipStream_t s) {
    const int mode = tamgcn_split_mode();
    const bool spl = mode >= 1;            // measured faster at every layer shape, HBM-bound ones included (fewer MFMA cycles per byte)
    return spl ? launch_wgrad_glds_spl<WMT, WKT, true>(a, s) : launch_wgrad_glds_spl<WMT, WKT, false>(a, s);
}

static inline int even_pitch(int n) {      // smallest p >= n with p == 2 (mod 4): conflict-free column reads, 8-byte rows
    int p = (n + 3) & ~3;
    return p + 2;
}

template <int KT, int WMT, int WKT, bool PS = false>
static int launch_wgrad(WgradArgs& a, hipStream_t s) {
    constexpr int BMW = PS ? 16 : 2 * WMT * 16, BKW = PS ? 16 : 2 * WKT * 16;
    const int V = a.V;
    // 16-byte slots: always for V % 4 == 0; the p-split kernel also takes rows that are only dword aligned (V = 25), its
    // slots then may straddle two frames -- the chunk must still be whole slots on both sides
    const bool ragged = (V % 4) != 0;
    const bool vec = !ragged || PS;
    int BT = 8;
    if (BT > a.T_out && !ragged) BT = a.T_out;
    size_t lds;
    for (;;) {
        a.BT = BT;
        a.TIN = (BT - 1) * a.stride + (KT - 1) * a.dil + 1;
        a.PY = even_pitch(BT * V);
        a.PX = even_pitch(a.TIN * V);
        lds = sizeof(float) * ((size_t)BMW * a.PY + (size_t)BKW * a.PX + 3 * (BMW + BKW));
        bool slots_ok = !(vec && (KT == 1 || PS)) ||
                        (BMW * (BT * V / 4) <= WG_NPF * NTHREADS && BKW * (a.TIN * V / 4) <= WG_NPF * NTHREADS);
        if (ragged && PS && ((BT * V) % 4 != 0 || (a.TIN * V) % 4 != 0)) slots_ok = false;
        if (PS && lds < sizeof(float) * 4 * KT * 256) lds = sizeof(float) * 4 * KT * 256;   // the final cross-wave reduction
        if ((lds <= 48 * 1024 && slots_ok) || BT == 1) {
            if (!slots_ok) { tamgcn_set_error("tamgcn_wgrad: prefetch slots exceeded (V=%d stride=%d)", V, a.stride); return -1; }
            break;
        }
        BT = BT / 2;
    }
    if (lds > 160 * 1024) { tamgcn_set_error("tamgcn_wgrad: tile does not fit LDS (V=%d)", V); return -1; }
    a.n_per = ceil_div(a.N * ceil_div(a.T_out, a.BT), a.nsplit);          // frame chunks per split
    tamgcn_note_kernel("wgrad_kernel<%d, %d, %d, %s%s>", KT, WMT, WKT, vec ? "true" : "false", PS ? ", p-split" : "");
    dim3 grid(ceil_div(a.K, BKW), ceil_div(a.M, BMW), a.nsplit);
    if (vec) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)wgrad_kernel<KT, WMT, WKT, true, PS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((wgrad_kernel<KT, WMT, WKT, true, PS>), grid, dim3(NTHREADS), lds, s, a);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)wgrad_kernel<KT, WMT, WKT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((wgrad_kernel<KT, WMT, WKT, false>), grid, dim3(NTHREADS), lds, s, a);
    }
    return 0;
}

// tile shape chosen from (M, K, KT); the host wrapper uses the same rule to size nsplit
static void wgrad_tile(int M, int K, int KT, int* wmt, int* wkt) {
    if (KT == 1) { *wmt = M <= 64 ? 2 : 4; *wkt = K <= 64 ? 2 : 4; }
    else if (KT == 9) { *wmt = 1; *wkt = 1; }
    else { *wmt = M <= 32 ? 1 : 2; *wkt = K <= 32 ? 1 : 2; if (*wmt != *wkt) { *wmt = 2; *wkt = 2; } }
}

// out[e] = sum_s part[s][e]: 64 consecutive e per block (coalesced), the split axis spread over
// the 4 waves of the block, fp64 accumulation, fixed order => deterministic.
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* part, int nsplit, long long stride_s, long long count,
                                                         float scale, int accumulate, float* out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    double s = 0.0;
    if (e < count)
        for (int k = w; k < nsplit; k += 4) s += (double)part[k * stride_s + e];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < count) {
        double t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float r = (float)(t * (double)scale);
        out[e] = accumulate ? out[e] + r : r;
    }
}

}  // namespace

// the LDS-DMA form applies (and with which tile) -- shared by tamgcn_wgrad and tamgcn_wgrad_max_split
static bool wgrad_glds_plan(const tamgcn_wgrad_desc* d, int* wmt, int* wkt) {
    wgrad_tile(d->M, d->K, d->KT, wmt, wkt);
    const bool al16 = (((uintptr_t)d->gy.x1 | (uintptr_t)d->src.x1 | (uintptr_t)(d->gy.x2 ? d->gy.x2 : d->gy.x1) |
                        (uintptr_t)(d->src.x2 ? d->src.x2 : d->src.x1)) & 15) == 0;
    // 1x1, or k x 1 with "same" padding (one window per tap); every window holds at least two 32-element chunks.
    // The 16-channel temporal branches stay on the register-staged kernels (one 16x16 tile: nothing for 8 waves to share;
    // measured r02: N-UCLA step 31.1 vs 30.2 ms, NTU 1.33 vs 1.0 ms per launch), and so do the 32-channel ones where that
    // kernel has its 16-byte form (V % 4 == 0: 51 vs 114 us per launch at N-UCLA; at V = 25 the tap form wins 672 vs 896 us); TAMGCN_WGRAD_TAPS=2 sends them here too,
    // =0 disables the tap form.
    const bool taps = d->KT > 1 && d->pad == d->dil * (d->KT - 1) / 2 && (d->dil * (d->KT - 1)) % 2 == 0 && tamgcn_wgrad_taps() &&
                      (d->M > 32 || d->K > 32 || (d->V % 4 != 0 && (d->M > 16 || d->K > 16)) || tamgcn_wgrad_taps() == 2);
    // a 1x1 conv with temporal stride 2 (V % 4 == 0): gy rows are contiguous, the x slot of contraction index p = t*V + v
    // sits at (2 t) V + v -- a per-lane source offset of the DMA piece, recomputed per chunk (round 4)
    const bool strided = d->KT == 1 && d->pad == 0 && d->stride == 2 && (d->V & 3) == 0 && d->T_out == (d->T_in - 1) / 2 + 1;
    bool glds = (taps || (d->KT == 1 && d->pad == 0)) && ((d->stride == 1 && d->T_in == d->T_out) || strided) && al16 &&
                (long long)(d->T_out - d->pad) * d->V >= 2 * W_PC;
    if (!glds) return false;
    if (d->KT > 1) { *wmt = d->M <= 64 ? 2 : 4; *wkt = d->K <= 64 ? 2 : 4; }      // the 1x1 rule (units of 32)
    // three stages of both operands (every source) must fit the CU's LDS: shrink the tile
    auto fits = [&](int tm, int tk) {
        const size_t rows = (size_t)tm * 32 * (d->gy.x2 ? 2 : 1) + (size_t)tk * 32 * (d->src.x2 ? 2 : 1);
        return sizeof(float) * W_NST * rows * W_PC <= 160 * 1024;
    };
    if (!fits(*wmt, *wkt) && *wmt == 4) *wmt = 2;
    if (!fits(*wmt, *wkt) && *wkt == 4) *wkt = 2;
    if (fits(*wmt, *wkt)) return true;
    wgrad_tile(d->M, d->K, d->KT, wmt, wkt);
    return false;
}

extern "C" int tamgcn_wgrad_max_split(const tamgcn_wgrad_desc* d) {
    if (!d || d->N <= 0 || d->T_out <= 0 || d->V <= 0) return -1;
    int wmt, wkt;
    if (!wgrad_glds_plan(d, &wmt, &wkt)) {
        // register-staged kernel: frame chunks hold at most 8 frames; at least one of those chunks per workgroup
        const long long m = (long long)d->N * ((d->T_out + 7) / 8);
        return (int)(m < d->N ? d->N : (m > 65535 ? 65535 : m));
    }
    const long long chunks = (long long)d->N * (((long long)(d->T_out - d->pad) * d->V) / W_PC);
    const long long m = chunks / 8;                      // at least 8 chunks of 32 per workgroup
    return (int)(m < d->N ? d->N : (m > 65535 ? 65535 : m));
}

extern "C" int tamgcn_wgrad(const tamgcn_wgrad_desc* d, void* stream) {
    TG_CHECK(d && d->gy.x1 && d->src.x1 && d->part, "tamgcn_wgrad: null pointer");
    TG_CHECK(d->N > 0 && d->M > 0 && d->K > 0 && d->T_in > 0 && d->T_out > 0 && d->V > 0 && d->nsplit > 0,
             "tamgcn_wgrad: bad dims");
    TG_CHECK(d->gy.coff + d->M <= d->gy.ctot && d->src.coff + d->K <= d->src.ctot, "tamgcn_wgrad: channel slice out of range");
    TG_CHECK(d->nsplit <= tamgcn_wgrad_max_split(d), "tamgcn_wgrad: nsplit=%d exceeds tamgcn_wgrad_max_split=%d", d->nsplit,
             tamgcn_wgrad_max_split(d));
    WgradArgs a;
    a.gy = make_src(d->gy); a.src = make_src(d->src);
    a.N = d->N; a.M = d->M; a.K = d->K; a.T_in = d->T_in; a.T_out = d->T_out; a.V = d->V;
    a.dil = d->dil; a.stride = d->stride; a.pad = d->pad; a.part = d->part; a.nsplit = d->nsplit;
    hipStream_t s = (hipStream_t)stream;
    int rc, wmt, wkt;
    const bool glds = wgrad_glds_plan(d, &wmt, &wkt);
    a.KTG = glds ? d->KT : 1;
    if (glds) {              // tile = 64 or 128 per side by the same rule as wgrad_tile (wmt: rows/32, wkt: cols/64)
        if (wmt == 2 && wkt == 2) rc = launch_wgrad_glds_src<2, 1>(a, s);
        else if (wmt == 4 && wkt == 2) rc = launch_wgrad_glds_src<4, 1>(a, s);
        else if (wmt == 2 && wkt == 4) rc = launch_wgrad_glds_src<2, 2>(a, s);
        else rc = launch_wgrad_glds_src<4, 2>(a, s);
        if (rc) return rc;
        TG_LAUNCH_CHECK("tamgcn_wgrad");
        return 0;
    }
    if (d->KT == 1) {        // register-staged 1x1 form: one frame of a 128-row tile must fit the prefetch slots (V = 64: 64-row tiles)
        const int per_row = (d->V + 3) / 4;
        if (wmt == 4 && 128 * per_row > WG_NPF * NTHREADS) wmt = 2;
        if (wkt == 4 && 128 * per_row > WG_NPF * NTHREADS) wkt = 2;
    }
    switch (d->KT) {
        case 1:
            if (wmt == 2 && wkt == 2) rc = launch_wgrad<1, 2, 2>(a, s);
            else if (wmt == 4 && wkt == 2) rc = launch_wgrad<1, 4, 2>(a, s);
            else if (wmt == 2 && wkt == 4) rc = launch_wgrad<1, 2, 4>(a, s);
            else rc = launch_wgrad<1, 4, 4>(a, s);
            break;
        // k x 1 kernels: when the preferred tile's line buffer (frames + temporal halo, V joints each) does not fit --
        // V = 64 -- fall back to the next smaller tile instead of refusing
        case 3:
            rc = wmt == 1 ? -1 : launch_wgrad<3, 2, 2>(a, s);
            if (rc == -1) rc = launch_wgrad<3, 1, 1>(a, s);
            break;
        case 5:
            rc = (d->M <= 16 && d->K <= 16 && (d->V % 4 == 0 || d->stride == 1)) ? launch_wgrad<5, 1, 1, true>(a, s) : -1;
            if (rc == -1 && wmt != 1) rc = launch_wgrad<5, 2, 2>(a, s);
            if (rc == -1) rc = launch_wgrad<5, 1, 1>(a, s);
            break;
        case 9: rc = launch_wgrad<9, 1, 1>(a, s); break;
        default: tamgcn_set_error("tamgcn_wgrad: kernel size %d not instantiated (1,3,5,9)", d->KT); return -1;
    }
    if (rc) return rc;
    TG_LAUNCH_CHECK("tamgcn_wgrad");
    return 0;
}

// first stage for many splits: group g (64 splits) is summed in place into its first slab.  Only
// the block owning (64 elements, group g) touches those slabs for those elements: race-free.
__global__ __launch_bounds__(256) void reduce_group_kernel(float* part, int nsplit, long long stride_s, long long count) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    const int s0 = blockIdx.y * 64, s1 = min(nsplit, s0 + 64);
    double s = 0.0;
    if (e < count)
        for (int k = s0 + w; k < s1; k += 4) s += (double)part[k * stride_s + e];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < count)
        part[(long long)s0 * stride_s + e] = (float)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// Many slab reductions in one launch: blockIdx.y = descriptor, blockIdx.x = 64-element group (the longest descriptor
// sets the grid, the others leave early).  A layer's backward produces ~25 partial-slab sets; reducing each with its own
// one or two 6-us launches was 1.4 ms of a 31 ms step.  Same arithmetic as reduce_sum_kernel: fp64, fixed order.
constexpr int RM_MAX = 24;
struct ReduceMulti { int n; tamgcn_reduce_desc d[RM_MAX]; };

__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceMulti md) {
    __shared__ double red[4][64];
    const tamgcn_reduce_desc& d = md.d[blockIdx.y];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    if ((long long)blockIdx.x * 64 >= d.count) return;          // whole workgroup: uniform
    double s = 0.0;
    if (e < d.count) {
        const float* p = d.part + e;
        int k = w;
        for (; k + 28 < d.nsplit; k += 32) {                      // eight independent loads in flight, summed as two groups of four
            float a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = p[(long long)(k + 4 * i) * d.stride_s];
            s += ((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3]);
            s += ((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]);
        }
        for (; k + 12 < d.nsplit; k += 16) {                      // four independent loads in flight
            float a0 = p[(long long)k * d.stride_s], a1 = p[(long long)(k + 4) * d.stride_s];
            float a2 = p[(long long)(k + 8) * d.stride_s], a3 = p[(long long)(k + 12) * d.stride_s];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; k < d.nsplit; k += 4) s += (double)p[(long long)k * d.stride_s];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < d.count) {
        double t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float r = (float)(t * (double)d.scale);
        d.out[e] = d.accumulate ? d.out[e] + r : r;
    }
}

extern "C" int tamgcn_reduce_multi(const tamgcn_reduce_desc* descs, int n, void* stream) {
    TG_CHECK(descs && n > 0, "tamgcn_reduce_multi: bad args");
    for (int i0 = 0; i0 < n; i0 += RM_MAX) {
        ReduceMulti md;
        md.n = n - i0 < RM_MAX ? n - i0 : RM_MAX;
        long long maxc = 0;
        for (int i = 0; i < md.n; ++i) {
            md.d[i] = descs[i0 + i];
            TG_CHECK(md.d[i].part && md.d[i].out && md.d[i].nsplit > 0 && md.d[i].count > 0, "tamgcn_reduce_multi: bad descriptor %d", i0 + i);
            if (md.d[i].count > maxc) maxc = md.d[i].count;
        }
        hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)((maxc + 63) / 64), md.n), dim3(256), 0, (hipStream_t)stream, md);
    }
    tamgcn_note_kernel("reduce_multi_kernel");
    TG_LAUNCH_CHECK("tamgcn_reduce_multi");
    return 0;
}

extern "C" int tamgcn_reduce_sum(float* part, int nsplit, long long stride_s, long long count,
                                 float scale, int accumulate, float* out, void* stream) {
    TG_CHECK(part && out && nsplit > 0 && count > 0, "tamgcn_reduce_sum: bad args");
    long long blocks = (count + 63) / 64;
    if (nsplit > 128) {                                 // two stages (clobbers `part`, which is scratch)
        int groups = (nsplit + 63) / 64;
        TG_CHECK(groups <= 65535, "tamgcn_reduce_sum: too many splits");
        hipLaunchKernelGGL(reduce_group_kernel, dim3((unsigned)blocks, groups), dim3(256), 0, (hipStream_t)stream,
                           part, nsplit, stride_s, count);
        nsplit = groups;
        stride_s *= 64;
    }
    hipLaunchKernelGGL(reduce_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       part, nsplit, stride_s, count, scale, accumulate, out);
    tamgcn_note_kernel("reduce_sum_kernel");
    TG_LAUNCH_CHECK("tamgcn_reduce_sum");
    return 0;
}
