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

namespace {

constexpr int BM = 64;        // output channels per block
constexpr int BK = 16;        // input channels per LDS chunk
constexpr int BKP = BK + 1;   // padded pitch of the weight tile (odd => conflict-free column reads)
constexpr int MAXCW = 5;      // column tiles (16 wide) per wave  -> 4 waves * 5 * 16 = 320 columns max
constexpr int MAXCOLS = 320;
constexpr int NTHREADS = 256;

struct ConvArgs {
    SrcDev src;
    int N, K, T_in, V;
    const float* w; const float* bias;
    int M, KT, dil, stride, pad;
    long long ws_m, ws_k, ws_t, w_off;   // weight element strides for A[i][k][tap]
    int up;
    float* y; int yctot, ycoff, T_out, T_y, ostride;
    const float* add1; const float* add2; const float* bcast; float bcast_scale;
    SrcDev mask; int has_mask;
    const float* aux; const float* aux_center; int auxctot, auxcoff;
    float* stats_part; int stats_ctot, stats_coff, nparts;
    // tiling (host chosen)
    int BT;        // output frames per block
    int CW;        // column tiles per wave
    int TIN;       // input frame slots staged per block
    int lstride;   // frame step between staged slots (stride for 1x1, else 1)
    int sB;        // slot step per output frame (1 for 1x1, else stride)
    int LB;        // TIN*V
    int pitchB;    // LDS pitch of one channel row (== 16 mod 32)
};

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

    // ---- epilogue
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[mt][r] = 0.f; s2[mt][r] = 0.f; }

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
                long long idx = (((long long)n * a.yctot + a.ycoff + m) * a.T_y + t) * V + vv[c];
                if (a.bcast) val = fmaf(a.bcast[((long long)m * a.N + n) * V + vv[c]], a.bcast_scale, val);
                if (a.add1) val += a.add1[idx];
                if (a.add2) val += a.add2[idx];
                if (a.has_mask) {
                    int mch = a.mask.coff + m;
                    long long midx = (((long long)n * a.mask.ctot + mch) * a.T_y + t) * V + vv[c];
                    if (!(src_value(a.mask, midx, mch) > 0.f)) val = 0.f;
                }
                a.y[idx] = val;
                if (a.stats_part) {
                    float x2 = val;
                    if (a.aux) x2 = a.aux[(((long long)n * a.auxctot + a.auxcoff + m) * a.T_y + t) * V + vv[c]] - a.aux_center[a.auxcoff + m];
                    s1[mt][r] += val;
                    s2[mt][r] = fmaf(val, x2, s2[mt][r]);
                }
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

struct ConvPlan { int BT, CW, TIN, lstride, sB, LB, pitchB, ntt; size_t lds; };

static int plan_conv(const tamgcn_conv_desc* d, ConvPlan* p) {
    int V = d->V;
    if (V < 1 || V > MAXCOLS) return -1;
    int BT = MAXCOLS / V; if (BT < 1) BT = 1;
    if (BT > d->T_out) BT = d->T_out;
    for (;;) {
        p->BT = BT;
        p->lstride = (d->KT == 1) ? d->stride : 1;
        p->sB = (d->KT == 1) ? 1 : d->stride;
        p->TIN = (d->KT == 1) ? BT : (BT - 1) * d->stride + (d->KT - 1) * d->dil + 1;
        p->LB = p->TIN * V;
        int pitch = p->LB;
        pitch += ((16 - (pitch & 31)) + 32) & 31;          // pitch == 16 (mod 32)
        p->pitchB = pitch;
        p->CW = ceil_div(ceil_div(BT * V, 16), 4);
        p->lds = sizeof(float) * ((size_t)d->KT * BM * BKP + (size_t)BK * pitch + 2 * 4 * BM);
        if (p->lds <= 64 * 1024 || BT == 1) break;
        BT = (BT + 1) / 2;
    }
    if (p->lds > 160 * 1024 || p->CW > MAXCW) return -1;
    p->ntt = ceil_div(d->T_out, p->BT);
    return 0;
}

}  // namespace

extern "C" int tamgcn_conv_nparts(const tamgcn_conv_desc* d) {
    ConvPlan p;
    if (!d || plan_conv(d, &p)) return -1;
    return d->N * p.ntt;
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
    ConvPlan p;
    TG_CHECK(plan_conv(d, &p) == 0, "tamgcn_conv: no tiling for V=%d KT=%d dil=%d stride=%d", d->V, d->KT, d->dil, d->stride);
    ConvArgs a;
    a.src = make_src(d->src);
    a.N = d->N; a.K = d->K; a.T_in = d->T_in; a.V = d->V;
    a.w = d->w; a.bias = d->bias; a.M = d->M; a.KT = d->KT; a.dil = d->dil; a.stride = d->stride; a.pad = d->pad;
    if (d->wmode == 0) { a.ws_m = (long long)d->K * d->KT; a.ws_k = d->KT; a.ws_t = 1; a.w_off = 0; }
    else { a.ws_m = d->KT; a.ws_k = (long long)d->M * d->KT; a.ws_t = -1; a.w_off = d->KT - 1; }
    a.up = d->up;
    a.y = d->y; a.yctot = d->yctot; a.ycoff = d->ycoff; a.T_out = d->T_out; a.T_y = d->T_y; a.ostride = d->ostride;
    a.add1 = d->add1; a.add2 = d->add2; a.bcast = d->bcast; a.bcast_scale = d->bcast_scale;
    a.has_mask = d->mask != nullptr; a.mask = d->mask ? make_src(*d->mask) : null_src();
    a.aux = d->aux; a.aux_center = d->aux_center; a.auxctot = d->auxctot; a.auxcoff = d->auxcoff;
    TG_CHECK(!d->aux || d->aux_center, "tamgcn_conv: aux needs aux_center");
    a.stats_part = d->stats_part; a.stats_ctot = d->stats_ctot; a.stats_coff = d->stats_coff;
    a.nparts = d->N * p.ntt;
    a.BT = p.BT; a.CW = p.CW; a.TIN = p.TIN; a.lstride = p.lstride; a.sB = p.sB; a.LB = p.LB; a.pitchB = p.pitchB;
    dim3 grid(p.ntt, ceil_div(d->M, BM), d->N);
    if (p.lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(conv_kernel, grid, dim3(NTHREADS), p.lds, (hipStream_t)stream, a);
    TG_LAUNCH_CHECK("tamgcn_conv");
    return 0;
}

// ===========================================================================
// weight gradient
// ===========================================================================
namespace {

struct WgradArgs {
    SrcDev gy, src;
    int N, M, K, T_in, T_out, V, dil, stride, pad;
    float* part; int nsplit;
    int BMW, BKW;     // tile of dW (multiples of 16, <= 64)
    int BT, TIN;      // frames per staged tile
    int PY, PX;       // LDS pitches (odd)
    int n_per;        // samples per split
};

template <int KT, int MTW, int KTW>
__global__ __launch_bounds__(NTHREADS) void wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BMW = MTW * 16, BKW = KTW * 16;
    float* Ys = smem;                         // [BMW][PY]
    float* Xs = Ys + BMW * a.PY;              // [BKW][PX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int k0 = blockIdx.x * BKW, m0 = blockIdx.y * BMW, split = blockIdx.z;
    const int V = a.V, V4 = (V + 3) >> 2;
    const int n_begin = split * a.n_per, n_end = min(a.N, n_begin + a.n_per);

    f32x4 acc[KT][MTW][KTW];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int x = 0; x < MTW; ++x)
#pragma unroll
            for (int y = 0; y < KTW; ++y) acc[t][x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long long gy_cs = (long long)a.T_out * V, x_cs = (long long)a.T_in * V;
    for (int n = n_begin; n < n_end; ++n) {
        for (int t0 = 0; t0 < a.T_out; t0 += a.BT) {
            const int bt = min(a.BT, a.T_out - t0);
            const int ncols = bt * V;
            const int tin0 = t0 * a.stride - a.pad;
            const int tin = (bt - 1) * a.stride + (KT - 1) * a.dil + 1;
            const int LX = tin * V;
            __syncthreads();
            for (int pos = tid; pos < ncols; pos += NTHREADS) {
                long long goff = (long long)n * a.gy.ctot * gy_cs + (long long)t0 * V + pos;
                for (int i = 0; i < BMW; ++i) {
                    int m = m0 + i;
                    float v = 0.f;
                    if (m < a.M) { int ch = a.gy.coff + m; v = src_value(a.gy, goff + ch * gy_cs, ch); }
                    Ys[i * a.PY + pos] = v;
                }
            }
            for (int pos = tid; pos < LX; pos += NTHREADS) {
                int slot = pos / V;
                int v = pos - slot * V;
                int th = tin0 + slot;
                bool ok = th >= 0 && th < a.T_in;
                long long goff = (long long)n * a.src.ctot * x_cs + (long long)th * V + v;
                for (int i = 0; i < BKW; ++i) {
                    int k = k0 + i;
                    float xv = 0.f;
                    if (ok && k < a.K) { int ch = a.src.coff + k; xv = src_value(a.src, goff + ch * x_cs, ch); }
                    Xs[i * a.PX + pos] = xv;
                }
            }
            __syncthreads();
            const int nsteps = bt * V4;
            for (int st = wave; st < nsteps; st += 4) {
                int tloc = st / V4;
                int v = (st - tloc * V4) * 4 + kq;
                bool vok = v < V;
                int vc = vok ? v : 0;
                float av[MTW];
#pragma unroll
                for (int x = 0; x < MTW; ++x) {
                    float t = Ys[(x * 16 + j) * a.PY + tloc * V + vc];
                    av[x] = vok ? t : 0.f;
                }
#pragma unroll
                for (int tap = 0; tap < KT; ++tap) {
                    int xo = (tloc * a.stride + tap * a.dil) * V + vc;
#pragma unroll
                    for (int y = 0; y < KTW; ++y) {
                        float bv = Xs[(y * 16 + j) * a.PX + xo];
#pragma unroll
                        for (int x = 0; x < MTW; ++x) acc[tap][x][y] = mfma16(av[x], bv, acc[tap][x][y]);
                    }
                }
            }
        }
    }
    // cross-wave reduction through LDS, then one plain store per element
    __syncthreads();
    float* Rs = smem;                          // [KT][BMW][BKW]
    for (int e = tid; e < KT * BMW * BKW; e += NTHREADS) Rs[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
        for (int x = 0; x < MTW; ++x)
#pragma unroll
            for (int y = 0; y < KTW; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(&Rs[(tap * BMW + x * 16 + kq * 4 + r) * BKW + y * 16 + j], acc[tap][x][y][r]);
    __syncthreads();
    float* out = a.part + (long long)split * a.M * a.K * KT;
    for (int e = tid; e < KT * BMW * BKW; e += NTHREADS) {
        int kk = e % BKW;
        int i = (e / BKW) % BMW;
        int tap = e / (BKW * BMW);
        int m = m0 + i, k = k0 + kk;
        if (m < a.M && k < a.K) out[((long long)m * a.K + k) * KT + tap] = Rs[e];
    }
}

template <int KT, int MTW, int KTW>
static int launch_wgrad(WgradArgs& a, hipStream_t s) {
    constexpr int BMW = MTW * 16, BKW = KTW * 16;
    a.BMW = BMW; a.BKW = BKW;
    int V = a.V;
    int BT = 160 / V; if (BT < 1) BT = 1; if (BT > a.T_out) BT = a.T_out;
    size_t lds;
    for (;;) {
        a.BT = BT;
        a.TIN = (BT - 1) * a.stride + (KT - 1) * a.dil + 1;
        a.PY = (BT * V) | 1;
        a.PX = (a.TIN * V) | 1;
        lds = sizeof(float) * ((size_t)BMW * a.PY + (size_t)BKW * a.PX);
        size_t red = sizeof(float) * (size_t)KT * BMW * BKW;
        if (lds < red) lds = red;
        if (lds <= 64 * 1024 || BT == 1) break;
        BT = (BT + 1) / 2;
    }
    if (lds > 160 * 1024) { tamgcn_set_error("tamgcn_wgrad: tile does not fit LDS (V=%d)", V); return -1; }
    int mt = ceil_div(a.M, BMW), kt = ceil_div(a.K, BKW);
    a.n_per = ceil_div(a.N, a.nsplit);
    dim3 grid(kt, mt, a.nsplit);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<KT, MTW, KTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((wgrad_kernel<KT, MTW, KTW>), grid, dim3(NTHREADS), lds, s, a);
    return 0;
}

__global__ void reduce_sum_kernel(const float* part, int nsplit, long long stride_s, long long count,
                                  float scale, int accumulate, float* out) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += (double)part[k * stride_s + e];
    float r = (float)(s * (double)scale);
    out[e] = accumulate ? out[e] + r : r;
}

}  // namespace

extern "C" int tamgcn_wgrad(const tamgcn_wgrad_desc* d, void* stream) {
    TG_CHECK(d && d->gy.x1 && d->src.x1 && d->part, "tamgcn_wgrad: null pointer");
    TG_CHECK(d->N > 0 && d->M > 0 && d->K > 0 && d->T_in > 0 && d->T_out > 0 && d->V > 0 && d->nsplit > 0,
             "tamgcn_wgrad: bad dims");
    TG_CHECK(d->gy.coff + d->M <= d->gy.ctot && d->src.coff + d->K <= d->src.ctot, "tamgcn_wgrad: channel slice out of range");
    TG_CHECK(d->nsplit <= d->N && d->nsplit <= 65535, "tamgcn_wgrad: nsplit=%d out of range", d->nsplit);
    WgradArgs a;
    a.gy = make_src(d->gy); a.src = make_src(d->src);
    a.N = d->N; a.M = d->M; a.K = d->K; a.T_in = d->T_in; a.T_out = d->T_out; a.V = d->V;
    a.dil = d->dil; a.stride = d->stride; a.pad = d->pad; a.part = d->part; a.nsplit = d->nsplit;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    bool small = d->M <= 32 && d->K <= 32;
    switch (d->KT) {
        case 1: rc = small ? launch_wgrad<1, 2, 2>(a, s) : launch_wgrad<1, 4, 4>(a, s); break;
        case 3: rc = launch_wgrad<3, 2, 2>(a, s); break;
        case 5: rc = launch_wgrad<5, 2, 2>(a, s); break;
        case 9: rc = launch_wgrad<9, 2, 2>(a, s); break;
        default: tamgcn_set_error("tamgcn_wgrad: kernel size %d not instantiated (1,3,5,9)", d->KT); return -1;
    }
    if (rc) return rc;
    TG_LAUNCH_CHECK("tamgcn_wgrad");
    return 0;
}

extern "C" int tamgcn_reduce_sum(const float* part, int nsplit, long long stride_s, long long count,
                                 float scale, int accumulate, float* out, void* stream) {
    TG_CHECK(part && out && nsplit > 0 && count > 0, "tamgcn_reduce_sum: bad args");
    long long blocks = (count + 255) / 256;
    hipLaunchKernelGGL(reduce_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       part, nsplit, stride_s, count, scale, accumulate, out);
    TG_LAUNCH_CHECK("tamgcn_reduce_sum");
    return 0;
}
