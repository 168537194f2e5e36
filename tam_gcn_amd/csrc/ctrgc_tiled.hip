// CTRGC for LARGE skeletons (V a multiple of 32; built for V = 64, BASELINE.json configs[4] "synthetic V = 64, T = 512,
// C = 256: LDS-tiling / MFMA crossover").  Reference arithmetic: models/ctrgcn.py:172-177 (V-generic) and the 3-subset
// sum of unit_gcn.forward, :252-254.
//
// Why a second kernel family.  At V = 20 / 25 (ctrgc.hip) a workgroup keeps the per-channel topology E of 16 channels x 3
// subsets in LDS (77 KB) and fuses the dense x3 = W3.x GEMM in front of a VALU aggregation.  At V = 64 one channel's E is
// 3 x 64 x 64 floats = 48 KB: an E-stationary workgroup can hold at most 3 channels, which leaves the x3 GEMM an M of 9
// rows (MFMA needs 16), while an x3-stationary workgroup would have to stream 786 KB of E per 8-frame chunk.  So the
// crossover is taken the other way: x3 = W3.x + b3 is the plain pointwise GEMM (tamgcn_conv, LDS-DMA ring, M = 3*Cout)
// and leaves through HBM once (the backward keeps it anyway), and the aggregation, which at V = 64 pads exactly into
// 16x16 MFMA tiles, becomes a matrix kernel of its own:
//
//   ctrgc_E_tiled      E[n,s,c,u,v] = alpha*(W4_s tanh(p_u - q_v) + b4_s)[c] + A_s[u,v]; D no longer fits LDS for all u
//                      (R*V*V*4 = 512 KB at R = 32): workgroup = (n, s, chunk of 512/V rows u), W4.D on MFMA
//   ctrgc_agg_fwd      y[n,c,t,u]    = sum_s sum_v x3_s[n,c,t,v] E_s[n,c,u,v]       workgroup = (n, c): E_c (48 KB) stationary,
//                      x3 streamed in 32-frame chunks; (32 x 64) = (32 x 192).(192 x 64) per chunk on v_mfma_f32_16x16x4_f32
//   ctrgc_agg_bwd      dx3_s[n,c,t,v] = sum_u dy[n,c,t,u] E_s[n,c,u,v]              the same with E transposed in LDS, 3 outputs
//   ctrgc_de_acc_mfma  dE_s[n,c,u,v]  = sum_t dy[n,c,t,u] x3_s[n,c,t,v]             (64 x 64) = (64 x T).(T x 64) per subset
//   ctrgc_de_tail_tiled dE -> dA, db4, dW4, dalpha, dp, dq per (n, s, u chunk)       (ctrgc_de.hip's tail with a column window)
//
// Per (n, c) the aggregation moves 3*T*V*4 (x3) + T*V*4 (y) + 48 KB (E) bytes for 2*T*V*3V flops: 24 FLOP/B at V = 64,
// right at the fp32 ridge (157 TFLOP/s / 8 TB/s = 20): both roofs ~5 ms for 256 clips x 256 channels x T = 512.
//
// LDS images.  A-type operands (rows = MFMA row index i, contraction along the row) use pitch V + 2 (16 rows x 2 k hit 32
// distinct banks with ds_read_b32); B-type operands of [k][j] form use pitch == 16 (mod 32).
#include "common.h"

namespace {

constexpr int TL_BT = 32;            // frames per chunk of the streaming kernels

// ---------------------------------------------------------------------------------------------------------------
// E for (n, s, rows u0..u0+UT-1), every channel
// ---------------------------------------------------------------------------------------------------------------
struct ETArgs {
    int N, Cout, S, R;
    const float* pq; const float* w4; const float* b4; const float* A; const float* alpha;
    float* E;
};

template <int V>
__global__ __launch_bounds__(512) void ctrgc_E_tiled_kernel(const ETArgs a) {
    constexpr int VV = V * V, UT = 512 / V, COLS = UT * V, NUC = V / UT, PD = COLS + 16, NW = 8;
    constexpr int NTILE = COLS / 16, TPW = NTILE / NW;
    static_assert(COLS == 512 && TPW == 4, "512 columns per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                         // [R][PD]
    float* Pq = Ds + a.R * PD;                // p [R][UT] then q [R][V]
    const int uc = blockIdx.x % NUC, ns = blockIdx.x / NUC;
    const int n = ns / a.S, s = ns - n * a.S, u0 = uc * UT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const long long NV = (long long)a.N * V;
    const float alpha = a.alpha[0];
    const float* pb = a.pq + ((long long)(s * 2 + 0) * a.R) * NV + (long long)n * V;
    const float* qb = a.pq + ((long long)(s * 2 + 1) * a.R) * NV + (long long)n * V;
    for (int e = tid; e < a.R * UT; e += 512) { const int r = e / UT, ul = e - r * UT; Pq[e] = pb[r * NV + u0 + ul]; }
    for (int e = tid; e < a.R * V; e += 512) { const int r = e / V, v = e - r * V; Pq[a.R * UT + e] = qb[r * NV + v]; }
    float Ar[TPW];
#pragma unroll
    for (int it = 0; it < TPW; ++it) Ar[it] = a.A[s * VV + u0 * V + (wave * TPW + it) * 16 + j];
    __syncthreads();
    for (int e = tid; e < a.R * COLS; e += 512) {
        const int r = e / COLS, col = e - r * COLS;
        const int ul = col / V, v = col - ul * V;
        Ds[r * PD + col] = fast_tanh(Pq[r * UT + ul] - Pq[a.R * UT + r * V + v]);
    }
    __syncthreads();
    float* Eg = a.E + ((long long)n * a.S + s) * a.Cout * VV + (long long)u0 * V;
    for (int c0 = 0; c0 < a.Cout; c0 += 16) {
        float aw[8], b4r[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) aw[k] = (k * 4 + kq < a.R) ? a.w4[((long long)s * a.Cout + c0 + j) * a.R + k * 4 + kq] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) b4r[r] = a.b4[s * a.Cout + c0 + kq * 4 + r];
#pragma unroll
        for (int it = 0; it < TPW; ++it) {
            const int col = (wave * TPW + it) * 16 + j;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k * 4 < a.R) acc = mfma16(aw[k], Ds[(k * 4 + kq < a.R ? k * 4 + kq : 0) * PD + col], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) Eg[(long long)(c0 + kq * 4 + r) * VV + col] = alpha * (acc[r] + b4r[r]) + Ar[it];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The three streaming kernels below are generic in (V, VP): V joints in HBM, VP = V rounded up to a multiple of 16 in LDS
// (zero padding), so that skeletons that do not tile into 16x16 MFMA shapes run on the same code: NTU's V = 25 as VP = 32
// (78 % of the matrix work is real; the kernels are HBM-bound).  Rows of V floats need no alignment (16-byte accesses from
// dword-aligned addresses run at full rate on gfx950, tools/probes/unaligned_probe.hip); a row's last 16-byte piece may read
// <= 12 bytes into the next row: x3 / E / dy come from the ops' allocator, which keeps that slack behind every tensor.
// ---------------------------------------------------------------------------------------------------------------
template <int V> struct TlGeo {
    static constexpr int VP = (V + 15) & ~15;          // joints in LDS
    static constexpr int PE = VP + 2;                  // A-type pitch
    static constexpr int K4 = (V + 3) / 4;             // contraction steps over joints (pads are zero)
    static constexpr int NUT = VP / 16;                // 16-wide joint tiles
    static constexpr bool AL = (V % 4) == 0;
};

// float4 of a contiguous [rows][V] run starting at flat index f -> LDS image with row pitch P (rows of V real + zero pads)
template <int V>
__device__ __forceinline__ void scatter4(float* img, int P, int f, const float4& t) {
    if constexpr ((V & 3) == 0) {
        const int r = f / V, c = f - r * V;
        float2* d = reinterpret_cast<float2*>(img + r * P + c);
        d[0] = make_float2(t.x, t.y);
        d[1] = make_float2(t.z, t.w);
    } else {
        const float vals[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int r = (f + k) / V, c = (f + k) - r * V; img[r * P + c] = vals[k]; }
    }
}

// E_c of every subset -> LDS, rows u (TRANSPOSE = false: Es[s][u][v]) or rows v (true: Es[s][v][u]), pitch VP + 2, pads zero
template <int V, int ST, bool TRANSPOSE>
__device__ __forceinline__ void stage_E(const float* __restrict__ Eg, int Cout, int n, int c, float* Es) {
    using G = TlGeo<V>;
    constexpr int VV = V * V, PE = G::PE, VP = G::VP;
    if constexpr (VP != V) {
        for (int e = threadIdx.x; e < ST * VP * PE; e += 256) Es[e] = 0.f;
        __syncthreads();
    }
    if constexpr (G::AL) {
        constexpr int NV4 = ST * VV / 4, NL = (NV4 + 255) / 256;
        float4 t[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = threadIdx.x + i * 256, ec = e < NV4 ? e : 0, s = ec / (VV / 4), r = ec - s * (VV / 4);
            t[i] = reinterpret_cast<const float4*>(Eg + (((long long)n * ST + s) * Cout + c) * VV)[r];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = threadIdx.x + i * 256, s = e / (VV / 4), r = e - s * (VV / 4);
            if (e < NV4) {
                const int u = (r * 4) / V, v = (r * 4) - u * V;
                if constexpr (!TRANSPOSE) {
                    float2* d = reinterpret_cast<float2*>(Es + (s * VP + u) * PE + v);
                    d[0] = make_float2(t[i].x, t[i].y);
                    d[1] = make_float2(t[i].z, t[i].w);
                } else {
                    float* d = Es + (s * VP + v) * PE + u;
                    d[0] = t[i].x; d[PE] = t[i].y; d[2 * PE] = t[i].z; d[3 * PE] = t[i].w;
                }
            }
        }
    } else {
        constexpr int NE = ST * VV, NL = (NE + 255) / 256;
        float t[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = threadIdx.x + i * 256, ec = e < NE ? e : 0, s = ec / VV, r = ec - s * VV;
            t[i] = Eg[(((long long)n * ST + s) * Cout + c) * VV + r];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = threadIdx.x + i * 256, s = e / VV, r = e - s * VV;
            if (e < NE) {
                const int u = r / V, v = r - u * V;
                Es[TRANSPOSE ? (s * VP + v) * PE + u : (s * VP + u) * PE + v] = t[i];
            }
        }
    }
}

// aggregation, forward: workgroup = (n, c)
template <int V, int ST>
__global__ __launch_bounds__(256, 2) void ctrgc_agg_fwd_kernel(int N, int Cout, int T, const float* __restrict__ x3, const float* __restrict__ E,
                                                               float* __restrict__ y, float* __restrict__ stats_part) {
    using G = TlGeo<V>;
    constexpr int PE = G::PE, VP = G::VP, UW = G::NUT / 2, BT = TL_BT, CH4 = BT * V / 4, NPF = (ST * CH4 + 255) / 256;
    static_assert(G::NUT % 2 == 0 && (BT * V) % 4 == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[2][4];
    float* Es = smem;                         // [ST][VP][PE]
    float* Xs = Es + ST * VP * PE;            // [ST][BT][PE]
    const int n = blockIdx.x / Cout, c = blockIdx.x - n * Cout;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int tt = wave >> 1, ub = (wave & 1) * UW;
    const long long TV = (long long)T * V;
    if constexpr (VP != V) for (int e = tid; e < ST * BT * PE; e += 256) Xs[e] = 0.f;     // joint pads stay zero
    stage_E<V, ST, false>(E, Cout, n, c, Es);

    float4 pre[NPF];
    auto prefetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + i * 256, ec = e < ST * CH4 ? e : 0, s = ec / CH4, r = ec - s * CH4;
            const bool ok = e < ST * CH4 && t0 + (r * 4) / V < T;
            pre[i] = ok ? reinterpret_cast<const float4*>(x3 + (((long long)n * ST + s) * Cout + c) * TV + (long long)t0 * V)[r]
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(0);
    float s1 = 0.f, s2 = 0.f;
    float* yrow = y + ((long long)n * Cout + c) * TV;
    for (int t0 = 0; t0 < T; t0 += BT) {
        __syncthreads();                      // previous chunk's MFMAs are done with Xs (first pass: E staged, pads zeroed)
        const int valid = min(BT, T - t0) * V;   // a piece that starts inside the clip may run past its end: those floats belong to
#pragma unroll                                   // the next row and must not enter the sum: cut per element
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + i * 256, s = e / CH4, r = e - s * CH4;
            if (e < ST * CH4) {
                float4 t = pre[i];
                if (r * 4 + 3 >= valid) { if (r * 4 + 0 >= valid) t.x = 0.f; if (r * 4 + 1 >= valid) t.y = 0.f; if (r * 4 + 2 >= valid) t.z = 0.f; t.w = 0.f; }
                scatter4<V>(Xs + s * BT * PE, PE, r * 4, t);
            }
        }
        __syncthreads();
        if (t0 + BT < T) prefetch(t0 + BT);   // in flight under the MFMAs
        f32x4 acc[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            const float* ar = Xs + (s * BT + tt * 16 + j) * PE + kq;
            const float* br = Es + (s * VP + ub * 16 + j) * PE + kq;
#pragma unroll
            for (int k4 = 0; k4 < G::K4; ++k4) {
                const float av = ar[k4 * 4];
#pragma unroll
                for (int u = 0; u < UW; ++u) acc[u] = mfma16(av, br[u * 16 * PE + k4 * 4], acc[u]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + tt * 16 + kq * 4 + r;
            if (t < T) {
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int uu = (ub + u) * 16 + j;
                    if (VP == V || uu < V) {
                        const float v = acc[u][r];
                        yrow[(long long)t * V + uu] = v;
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
            }
        }
    }
    if (stats_part) {
        s1 = wave_sum64(s1); s2 = wave_sum64(s2);
        if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
        __syncthreads();
        if (tid < 2) stats_part[((long long)tid * Cout + c) * N + n] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    }
}

// aggregation, backward w.r.t. x3: dx3_s[t][v] = sum_u dy[t][u] E_s[u][v]
template <int V, int ST>
__global__ __launch_bounds__(256, 2) void ctrgc_agg_bwd_kernel(int N, int Cout, int T, const SrcDev dy, const float* __restrict__ E,
                                                               float* __restrict__ dx3, float* __restrict__ db3_part) {
    using G = TlGeo<V>;
    constexpr int PE = G::PE, VP = G::VP, VW = G::NUT / 2, BT = TL_BT, CH4 = BT * V / 4, NPF = (CH4 + 255) / 256;
    static_assert(G::NUT % 2 == 0 && (BT * V) % 4 == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[ST][4];
    float* Es = smem;                         // [ST][v][PE] (transposed)
    float* Zs = Es + ST * VP * PE;            // [BT][PE]
    const int n = blockIdx.x / Cout, c = blockIdx.x - n * Cout;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int tt = wave >> 1, vb = (wave & 1) * VW;
    const long long TV = (long long)T * V;
    if constexpr (VP != V) for (int e = tid; e < BT * PE; e += 256) Zs[e] = 0.f;
    stage_E<V, ST, true>(E, Cout, n, c, Es);
    const int ch = dy.coff + c;
    const float c1 = dy.coef ? dy.coef[ch] : 1.f;
    const float c2 = (dy.coef && dy.x2) ? dy.coef[dy.ctot + ch] : 0.f;
    const float c0 = dy.coef ? dy.coef[2 * dy.ctot + ch] : 0.f;
    const long long dyb = ((long long)n * dy.ctot + ch) * TV;

    float4 p1[NPF], p2[NPF];
    auto prefetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int r = tid + i * 256;
            const bool ok = r < CH4 && t0 + (r * 4) / V < T;
            p1[i] = ok ? reinterpret_cast<const float4*>(dy.x1 + dyb + (long long)t0 * V)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            p2[i] = (ok && dy.x2) ? reinterpret_cast<const float4*>(dy.x2 + dyb + (long long)t0 * V)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(0);
    float sb[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) sb[s] = 0.f;
    for (int t0 = 0; t0 < T; t0 += BT) {
        __syncthreads();
        const int valid = min(BT, T - t0) * V;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int r = tid + i * 256;
            if (r < CH4) {
                float o[4] = {fmaf(c1, p1[i].x, fmaf(c2, p2[i].x, c0)), fmaf(c1, p1[i].y, fmaf(c2, p2[i].y, c0)),
                              fmaf(c1, p1[i].z, fmaf(c2, p2[i].z, c0)), fmaf(c1, p1[i].w, fmaf(c2, p2[i].w, c0))};
#pragma unroll
                for (int k = 0; k < 4; ++k) {       // rows past T are zero (the prologue's constant must not leak in)
                    if (dy.act == 1) o[k] = fmaxf(o[k], 0.f);
                    if (r * 4 + k >= valid) o[k] = 0.f;
                }
                scatter4<V>(Zs, PE, r * 4, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
        __syncthreads();
        if (t0 + BT < T) prefetch(t0 + BT);
        f32x4 acc[ST][VW];
#pragma unroll
        for (int s = 0; s < ST; ++s)
#pragma unroll
            for (int v = 0; v < VW; ++v) acc[s][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* ar = Zs + (tt * 16 + j) * PE + kq;
#pragma unroll
        for (int k4 = 0; k4 < G::K4; ++k4) {
            const float av = ar[k4 * 4];
#pragma unroll
            for (int s = 0; s < ST; ++s)
#pragma unroll
                for (int v = 0; v < VW; ++v) acc[s][v] = mfma16(av, Es[(s * VP + (vb + v) * 16 + j) * PE + k4 * 4 + kq], acc[s][v]);
        }
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            float* orow = dx3 + (((long long)n * ST + s) * Cout + c) * TV;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + tt * 16 + kq * 4 + r;
                if (t < T) {
#pragma unroll
                    for (int v = 0; v < VW; ++v) {
                        const int vv = (vb + v) * 16 + j;
                        if (VP == V || vv < V) {
                            orow[(long long)t * V + vv] = acc[s][v][r];
                            sb[s] += acc[s][v][r];
                        }
                    }
                }
            }
        }
    }
    if (db3_part) {
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            const float t = wave_sum64(sb[s]);
            if (lane == 0) red[s][wave] = t;
        }
        __syncthreads();
        if (tid < ST) db3_part[(long long)n * ST * Cout + tid * Cout + c] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dE_s[u][v] = sum_t dy[t][u] x3_s[t][v]: workgroup = (n, c)
// ---------------------------------------------------------------------------------------------------------------
template <int V, int ST>
__global__ __launch_bounds__(256) void ctrgc_de_acc_mfma_kernel(int N, int Cout, int T, const float* __restrict__ x3, const SrcDev dy,
                                                                float* __restrict__ dE) {
    using G = TlGeo<V>;
    constexpr int VP = G::VP, P = VP + 16, NUT = G::NUT, UW = NUT < 4 ? NUT : 4, VWAVES = 4 / UW, VPW = NUT / VWAVES, BT = TL_BT;
    constexpr int CH4 = BT * V / 4, NPX = (ST * CH4 + 255) / 256, NPY = (CH4 + 255) / 256, VV = V * V;
    static_assert((BT * V) % 4 == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Zs = smem;                         // [BT][P]      dy chunk (prologue applied)
    float* Xs = Zs + BT * P;                  // [ST][BT][P]
    const int n = blockIdx.x / Cout, c = blockIdx.x - n * Cout;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int ut = wave % UW, vb = (wave / UW) * VPW;
    const long long TV = (long long)T * V;
    const int ch = dy.coff + c;
    const float c1 = dy.coef ? dy.coef[ch] : 1.f;
    const float c2 = (dy.coef && dy.x2) ? dy.coef[dy.ctot + ch] : 0.f;
    const float c0 = dy.coef ? dy.coef[2 * dy.ctot + ch] : 0.f;
    const long long dyb = ((long long)n * dy.ctot + ch) * TV;
    if constexpr (VP != V) for (int e = tid; e < (ST + 1) * BT * P; e += 256) smem[e] = 0.f;   // joint pads stay zero

    float4 px[NPX], p1[NPY], p2[NPY];
    auto prefetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int e = tid + i * 256, ec = e < ST * CH4 ? e : 0, s = ec / CH4, r = ec - s * CH4;
            const bool ok = e < ST * CH4 && t0 + (r * 4) / V < T;
            px[i] = ok ? reinterpret_cast<const float4*>(x3 + (((long long)n * ST + s) * Cout + c) * TV + (long long)t0 * V)[r]
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
            const int r = tid + i * 256;
            const bool ok = r < CH4 && t0 + (r * 4) / V < T;
            p1[i] = ok ? reinterpret_cast<const float4*>(dy.x1 + dyb + (long long)t0 * V)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            p2[i] = (ok && dy.x2) ? reinterpret_cast<const float4*>(dy.x2 + dyb + (long long)t0 * V)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(0);
    f32x4 acc[ST][VPW];
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
        for (int v = 0; v < VPW; ++v) acc[s][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += BT) {
        __syncthreads();
        const int valid = min(BT, T - t0) * V;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int e = tid + i * 256, s = e / CH4, r = e - s * CH4;
            if (e < ST * CH4) {
                float4 t = px[i];
                if (r * 4 + 3 >= valid) { if (r * 4 + 0 >= valid) t.x = 0.f; if (r * 4 + 1 >= valid) t.y = 0.f; if (r * 4 + 2 >= valid) t.z = 0.f; t.w = 0.f; }
                if constexpr (G::AL) { const int tl = (r * 4) / V, v = (r * 4) - tl * V; *reinterpret_cast<float4*>(Xs + (s * BT + tl) * P + v) = t; }
                else scatter4<V>(Xs + s * BT * P, P, r * 4, t);
            }
        }
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
            const int r = tid + i * 256;
            if (r < CH4) {
                float o[4] = {fmaf(c1, p1[i].x, fmaf(c2, p2[i].x, c0)), fmaf(c1, p1[i].y, fmaf(c2, p2[i].y, c0)),
                              fmaf(c1, p1[i].z, fmaf(c2, p2[i].z, c0)), fmaf(c1, p1[i].w, fmaf(c2, p2[i].w, c0))};
#pragma unroll
                for (int k = 0; k < 4; ++k) { if (dy.act == 1) o[k] = fmaxf(o[k], 0.f); if (r * 4 + k >= valid) o[k] = 0.f; }
                const float4 t = make_float4(o[0], o[1], o[2], o[3]);
                if constexpr (G::AL) { const int tl = (r * 4) / V, u = (r * 4) - tl * V; *reinterpret_cast<float4*>(Zs + tl * P + u) = t; }
                else scatter4<V>(Zs, P, r * 4, t);
            }
        }
        __syncthreads();
        if (t0 + BT < T) prefetch(t0 + BT);
#pragma unroll
        for (int k4 = 0; k4 < BT / 4; ++k4) {
            const float av = Zs[(k4 * 4 + kq) * P + ut * 16 + j];          // A[i = u][k = t]
#pragma unroll
            for (int s = 0; s < ST; ++s)
#pragma unroll
                for (int v = 0; v < VPW; ++v)
                    acc[s][v] = mfma16(av, Xs[(s * BT + k4 * 4 + kq) * P + (vb + v) * 16 + j], acc[s][v]);   // B[k = t][j = v]
        }
    }
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        float* o = dE + (((long long)n * ST + s) * Cout + c) * VV;
#pragma unroll
        for (int v = 0; v < VPW; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int uu = ut * 16 + kq * 4 + r, vv = (vb + v) * 16 + j;
                if (VP == V || (uu < V && vv < V)) o[uu * V + vv] = acc[s][v][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dE -> dA, db4, dW4, dalpha, dp, dq for (n, s, rows u0..u0+UT-1): ctrgc_de.hip's tail kernel on a 512-column window of
// (u, v).  dp of the chunk's own rows is complete (the sum over v is inside the window), dq is a partial sum over the
// chunk's rows; every chunk writes its whole [2R][V] slab (zeros for foreign dp rows), the host sums the NUC slabs.
// ---------------------------------------------------------------------------------------------------------------
struct TTArgs {
    int N, Cout, S, R;
    const float* dE; const float* pq; const float* w4; const float* b4; const float* alpha;
    float* dA_part;        // [N][S][V][V]            (each chunk owns its rows)
    float* dw4_part;       // [N*NUC][S][Cout][R]
    float* db4_part;       // [N*NUC][S][Cout]
    float* dalpha_part;    // [N*S*NUC]
    float* dpq;            // [NUC][S*2*R][N][V]
};

template <int V, int RT>
__global__ __launch_bounds__(512) void ctrgc_de_tail_tiled_kernel(const TTArgs a) {
    constexpr int VV = V * V, UT = 512 / V, COLS = 512, NUC = V / UT, PD = COLS + 2, NT = 512, NW = 8;
    constexpr int NCT = COLS / 16, TPW = NCT / NW, KST = COLS / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red_alpha[NW];
    float* Ds = smem;                        // [R][PD]    D, later dS in place
    float* DEs = Ds + a.R * PD;              // [16][PD]   dE chunk
    float* red = DEs + 16 * PD;              // [NW][16][RT*16]
    float* Pq = red + NW * 16 * RT * 16;     // p [R][UT], q [R][V]
    const int uc = blockIdx.x % NUC, ns = blockIdx.x / NUC;
    const int n = ns / a.S, s = ns - n * a.S, u0 = uc * UT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mj = lane & 15, mkq = lane >> 4;
    const long long NV = (long long)a.N * V;
    const float alpha = a.alpha[0];
    const long long slab = (long long)n * NUC + uc;

    float4 pre[4];                           // 16 channels x 512 columns = 2048 float4 / 512 threads
    auto load = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * NT, cl = e / (COLS / 4), r = e - cl * (COLS / 4);
            pre[i] = reinterpret_cast<const float4*>(a.dE + (((long long)n * a.S + s) * a.Cout + c0 + cl) * VV + (long long)u0 * V)[r];
        }
    };
    load(0);
    {
        const float* pb = a.pq + ((long long)(s * 2 + 0) * a.R) * NV + (long long)n * V;
        const float* qb = a.pq + ((long long)(s * 2 + 1) * a.R) * NV + (long long)n * V;
        for (int e = tid; e < a.R * UT; e += NT) { const int r = e / UT, ul = e - r * UT; Pq[e] = pb[r * NV + u0 + ul]; }
        for (int e = tid; e < a.R * V; e += NT) { const int r = e / V, v = e - r * V; Pq[a.R * UT + e] = qb[r * NV + v]; }
        __syncthreads();
        for (int e = tid; e < a.R * COLS; e += NT) {
            const int r = e / COLS, col = e - r * COLS;
            const int ul = col / V, v = col - ul * V;
            Ds[r * PD + col] = fast_tanh(Pq[r * UT + ul] - Pq[a.R * UT + r * V + v]);
        }
    }
    f32x4 accG[TPW][RT];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) accG[q][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accA = 0.f, dalpha_acc = 0.f;

    for (int c0 = 0; c0 < a.Cout; c0 += 16) {
        float aw[RT][4];                     // W4^T fragment of this chunk: A[i = r][k = c]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                aw[rt][k4] = (rt * 16 + mj < a.R) ? a.w4[((long long)s * a.Cout + c0 + k4 * 4 + mkq) * a.R + rt * 16 + mj] : 0.f;
        __syncthreads();                     // previous chunk (and the D fill) done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * NT, cl = e / (COLS / 4), r = e - cl * (COLS / 4);
            const float4 t = pre[i];
            float2* d = reinterpret_cast<float2*>(DEs + cl * PD + r * 4);
            d[0] = make_float2(t.x, t.y);
            d[1] = make_float2(t.z, t.w);
        }
        __syncthreads();
        if (c0 + 16 < a.Cout) load(c0 + 16);
        // dG[r][col] += sum_c W4[c][r] dE[c][col]
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int col = (wave + q * NW) * 16 + mj;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float b = DEs[(k4 * 4 + mkq) * PD + col];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) accG[q][rt] = mfma16(aw[rt][k4], b, accG[q][rt]);
            }
        }
        {   // dA: sum over channels, one column per thread
            float t = 0.f;
#pragma unroll
            for (int cl = 0; cl < 16; ++cl) t += DEs[cl * PD + tid];
            accA += t;
        }
        {   // db4raw[c] = sum_col dE[c][col]
            const int cl = (tid >> 4) & 15, l16 = tid & 15;          // threads 256.. shadow 0..255 (no second write)
            float t = 0.f;
            for (int col = l16; col < COLS; col += 16) t += DEs[cl * PD + col];
            t = wave_sum16(t);
            if (l16 == 0 && tid < 256) {
                a.db4_part[(slab * a.S + s) * a.Cout + c0 + cl] = alpha * t;
                dalpha_acc = fmaf(a.b4[s * a.Cout + c0 + cl], t, dalpha_acc);
            }
        }
        {   // dW4raw[c][r] = sum_col dE[c][col] D[r][col]: K = 512 columns split over the waves
            f32x4 accW[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) accW[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int st = wave; st < KST; st += NW) {
                const int k = st * 4 + mkq;
                const float av = DEs[mj * PD + k];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = rt * 16 + mj;
                    const float bv = r < a.R ? Ds[r * PD + k] : 0.f;
                    accW[rt] = mfma16(av, bv, accW[rt]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[(wave * 16 + mkq * 4 + rr) * (RT * 16) + rt * 16 + mj] = accW[rt][rr];
        }
        __syncthreads();
        for (int e = tid; e < 16 * RT * 16; e += NT) {
            const int cl = e / (RT * 16), r = e - cl * (RT * 16);
            if (r < a.R) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += red[(w * 16 + cl) * (RT * 16) + r];
                const long long wi = ((long long)s * a.Cout + c0 + cl) * a.R + r;
                a.dw4_part[slab * a.S * a.Cout * a.R + wi] = alpha * t;
                dalpha_acc = fmaf(a.w4[wi], t, dalpha_acc);
            }
        }
    }
    a.dA_part[((long long)n * a.S + s) * VV + (long long)u0 * V + tid] = accA;
    __syncthreads();                         // every wave is done reading D
#pragma unroll
    for (int q = 0; q < TPW; ++q) {          // dS[r][col] = alpha * dG * (1 - D^2), in place over D
        const int col = (wave + q * NW) * 16 + mj;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rt * 16 + mkq * 4 + rr;
                if (r < a.R) {
                    const float d = Ds[r * PD + col];
                    Ds[r * PD + col] = alpha * accG[q][rt][rr] * (1.f - d * d);
                }
            }
    }
    __syncthreads();
    // dp[r][u] = sum_v dS[r][u][v] (u of this chunk, 0 elsewhere);  dq[r][v] = -sum_{u in chunk} dS[r][u][v]
    for (int e = tid; e < a.R * V * 2; e += NT) {
        const int which = e / (a.R * V);
        const int rem = e - which * a.R * V;
        const int r = rem / V, k = rem - r * V;
        float t = 0.f;
        if (which == 0) {
            const int ul = k - u0;
            if (ul >= 0 && ul < UT) {
#pragma unroll 8
                for (int v = 0; v < V; ++v) t += Ds[r * PD + ul * V + v];
            }
        } else {
#pragma unroll
            for (int ul = 0; ul < UT; ++ul) t -= Ds[r * PD + ul * V + k];
        }
        a.dpq[(((long long)uc * a.S * 2 + s * 2 + which) * a.R + r) * NV + (long long)n * V + k] = t;
    }
    dalpha_acc = wave_sum64(dalpha_acc);
    if (lane == 0) red_alpha[wave] = dalpha_acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < NW; ++w) t += red_alpha[w];
        a.dalpha_part[(n * a.S + s) * NUC + uc] = t;
    }
}

template <int V>
constexpr size_t agg_lds(int ST, bool bwd) {
    return sizeof(float) * (size_t)(ST * TlGeo<V>::VP * TlGeo<V>::PE + (bwd ? 1 : ST) * TL_BT * TlGeo<V>::PE);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
static bool tiled_v_ok(int V) { return V == 64 || V == 32; }          // the whole family (E / tail tiles included)
static bool stream_v_ok(int V) { return V == 64 || V == 32 || V == 25 || V == 20; }   // the three streaming kernels

// 1: the whole tiled family; 2: only the streaming kernels (aggregation fwd / bwd, dE accumulation): E and the dE tail
// then come from the LDS-resident family's tamgcn_ctrgc_build_e / tamgcn_ctrgc_bwd_de_tail; 0: neither
extern "C" int tamgcn_ctrgc_tiled_supported(int V) { return tiled_v_ok(V) ? 1 : (stream_v_ok(V) ? 2 : 0); }

// largest dynamic-LDS request of the tiled family for (S, V, R): what tamgcn_ctrgc_lds_bytes reports for these V
int tamgcn_ctrgc_tiled_lds_bytes(int S, int V, int R) {
    if (!tiled_v_ok(V) || !(S == 1 || S == 3) || R < 4 || R > 32 || R % 4) return -1;
    const int UT = 512 / V, RT = R <= 16 ? 1 : 2;
    const size_t agg = sizeof(float) * (size_t)(S * V * (V + 2) + S * TL_BT * (V + 2));
    const size_t tail = sizeof(float) * ((size_t)(R + 16) * (512 + 2) + 8 * 16 * RT * 16 + (size_t)R * (UT + V));
    const size_t et = sizeof(float) * ((size_t)R * (512 + 16) + (size_t)R * (UT + V));
    size_t m = agg > tail ? agg : tail;
    return (int)(m > et ? m : et);
}
extern "C" int tamgcn_ctrgc_tiled_chunks(int V) { return tiled_v_ok(V) ? V / (512 / V) : -1; }

#define TL_DISPATCH_V(CALL64, CALL32) do { if (d->V == 64) { CALL64; } else { CALL32; } } while (0)

extern "C" int tamgcn_ctrgc_tiled_build_e(const tamgcn_ctrgc_desc* d, float* E, void* stream) {
    TG_CHECK(d && E && d->pq && d->w4 && d->b4 && d->A && d->alpha, "tamgcn_ctrgc_tiled_build_e: null pointer");
    TG_CHECK(tiled_v_ok(d->V), "tamgcn_ctrgc_tiled_build_e: V=%d (the tiled CTRGC kernels are built for V in {32, 64})", d->V);
    TG_CHECK(d->N > 0 && d->S > 0 && d->Cout > 0 && d->Cout % 16 == 0, "tamgcn_ctrgc_tiled_build_e: bad shape N=%d S=%d Cout=%d", d->N, d->S, d->Cout);
    TG_CHECK(d->R >= 4 && d->R <= 32 && d->R % 4 == 0, "tamgcn_ctrgc_tiled_build_e: R=%d outside 4..32 (multiples of 4)", d->R);
    ETArgs a;
    a.N = d->N; a.Cout = d->Cout; a.S = d->S; a.R = d->R;
    a.pq = d->pq; a.w4 = d->w4; a.b4 = d->b4; a.A = d->A; a.alpha = d->alpha; a.E = E;
    const int UT = 512 / d->V, NUC = d->V / UT;
    const size_t lds = sizeof(float) * ((size_t)d->R * (512 + 16) + (size_t)d->R * (UT + d->V));
    static tg_devmask f64 = 0, f32 = 0;
    TL_DISPATCH_V(tg_allow_lds((const void*)ctrgc_E_tiled_kernel<64>, 80 * 1024, &f64);
                  hipLaunchKernelGGL((ctrgc_E_tiled_kernel<64>), dim3(d->N * d->S * NUC), dim3(512), lds, (hipStream_t)stream, a),
                  tg_allow_lds((const void*)ctrgc_E_tiled_kernel<32>, 80 * 1024, &f32);
                  hipLaunchKernelGGL((ctrgc_E_tiled_kernel<32>), dim3(d->N * d->S * NUC), dim3(512), lds, (hipStream_t)stream, a));
    tamgcn_note_kernel("ctrgc_E_tiled_kernel<%d>", d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_tiled_build_e");
    return 0;
}

#define TL_CASE(KERNEL, VV_, SS_, LDS_, ...)                                                                          \
    if (d->V == VV_ && d->S == SS_) {                                                                                 \
        static tg_devmask flag = 0;                                                                                   \
        const size_t lds_ = (LDS_);                                                                                   \
        tg_allow_lds((const void*)KERNEL<VV_, SS_>, lds_, &flag);                                                     \
        hipLaunchKernelGGL((KERNEL<VV_, SS_>), dim3((unsigned)(d->N * d->Cout)), dim3(256), lds_, (hipStream_t)stream, __VA_ARGS__); \
        tamgcn_note_kernel(#KERNEL "<%d, %d>", VV_, SS_);                                                             \
        launched = true;                                                                                              \
    }

static int tiled_common_check(const tamgcn_ctrgc_desc* d, const char* who) {
    if (!d) { tamgcn_set_error("%s: null descriptor", who); return -1; }
    if (!stream_v_ok(d->V)) { tamgcn_set_error("%s: V=%d (the streaming CTRGC kernels are built for V in {20, 25, 32, 64})", who, d->V); return -1; }
    if (!(d->S == 1 || d->S == 3)) { tamgcn_set_error("%s: S=%d (1 or 3 subsets)", who, d->S); return -1; }
    if (!(d->N > 0 && d->Cout > 0 && d->T > 0)) { tamgcn_set_error("%s: bad dims N=%d Cout=%d T=%d", who, d->N, d->Cout, d->T); return -1; }
    if ((long long)d->N * d->Cout >= (1LL << 31)) { tamgcn_set_error("%s: N*Cout too large for the grid", who); return -1; }
    return 0;
}

extern "C" int tamgcn_ctrgc_tiled_agg_fwd(const tamgcn_ctrgc_desc* d, const float* x3, const float* E, float* y, float* stats_part, void* stream) {
    if (tiled_common_check(d, "tamgcn_ctrgc_tiled_agg_fwd")) return -1;
    TG_CHECK(x3 && E && y, "tamgcn_ctrgc_tiled_agg_fwd: null pointer");
    bool launched = false;
    TL_CASE(ctrgc_agg_fwd_kernel, 64, 3, agg_lds<64>(3, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 64, 1, agg_lds<64>(1, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 32, 3, agg_lds<32>(3, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 32, 1, agg_lds<32>(1, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 25, 3, agg_lds<25>(3, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 25, 1, agg_lds<25>(1, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 20, 3, agg_lds<20>(3, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    else TL_CASE(ctrgc_agg_fwd_kernel, 20, 1, agg_lds<20>(1, false), d->N, d->Cout, d->T, x3, E, y, stats_part)
    TG_CHECK(launched, "tamgcn_ctrgc_tiled_agg_fwd: no instantiation for S=%d V=%d", d->S, d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_tiled_agg_fwd");
    return 0;
}

extern "C" int tamgcn_ctrgc_tiled_agg_bwd(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, const float* E, float* dx3, float* db3_part, void* stream) {
    if (tiled_common_check(d, "tamgcn_ctrgc_tiled_agg_bwd")) return -1;
    TG_CHECK(dy && dy->x1 && E && dx3, "tamgcn_ctrgc_tiled_agg_bwd: null pointer");
    TG_CHECK(dy->ctot >= dy->coff + d->Cout, "tamgcn_ctrgc_tiled_agg_bwd: dy has %d channels from %d, need %d", dy->ctot, dy->coff, d->Cout);
    const SrcDev dys = make_src(*dy);
    bool launched = false;
    TL_CASE(ctrgc_agg_bwd_kernel, 64, 3, agg_lds<64>(3, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 64, 1, agg_lds<64>(1, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 32, 3, agg_lds<32>(3, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 32, 1, agg_lds<32>(1, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 25, 3, agg_lds<25>(3, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 25, 1, agg_lds<25>(1, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 20, 3, agg_lds<20>(3, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    else TL_CASE(ctrgc_agg_bwd_kernel, 20, 1, agg_lds<20>(1, true), d->N, d->Cout, d->T, dys, E, dx3, db3_part)
    TG_CHECK(launched, "tamgcn_ctrgc_tiled_agg_bwd: no instantiation for S=%d V=%d", d->S, d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_tiled_agg_bwd");
    return 0;
}

extern "C" int tamgcn_ctrgc_tiled_de_acc(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, const float* x3, float* dE, void* stream) {
    if (tiled_common_check(d, "tamgcn_ctrgc_tiled_de_acc")) return -1;
    TG_CHECK(dy && dy->x1 && x3 && dE, "tamgcn_ctrgc_tiled_de_acc: null pointer");
    TG_CHECK(dy->ctot >= dy->coff + d->Cout, "tamgcn_ctrgc_tiled_de_acc: dy has %d channels from %d, need %d", dy->ctot, dy->coff, d->Cout);
    const SrcDev dys = make_src(*dy);
    bool launched = false;
#define TL_DE_LDS(VV_, SS_) (sizeof(float) * (size_t)((SS_ + 1) * TL_BT * (TlGeo<VV_>::VP + 16)))
    TL_CASE(ctrgc_de_acc_mfma_kernel, 64, 3, TL_DE_LDS(64, 3), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 64, 1, TL_DE_LDS(64, 1), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 32, 3, TL_DE_LDS(32, 3), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 32, 1, TL_DE_LDS(32, 1), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 25, 3, TL_DE_LDS(25, 3), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 25, 1, TL_DE_LDS(25, 1), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 20, 3, TL_DE_LDS(20, 3), d->N, d->Cout, d->T, x3, dys, dE)
    else TL_CASE(ctrgc_de_acc_mfma_kernel, 20, 1, TL_DE_LDS(20, 1), d->N, d->Cout, d->T, x3, dys, dE)
    TG_CHECK(launched, "tamgcn_ctrgc_tiled_de_acc: no instantiation for S=%d V=%d", d->S, d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_tiled_de_acc");
    return 0;
}

#define TL_TAIL_CASE(VV_, RT_)                                                                                        \
    if (d->V == VV_ && rt == RT_) {                                                                                   \
        static tg_devmask flag = 0;                                                                                   \
        const int UT = 512 / VV_;                                                                                     \
        const size_t lds = sizeof(float) * ((size_t)(d->R + 16) * (512 + 2) + 8 * 16 * RT_ * 16 + (size_t)d->R * (UT + VV_)); \
        tg_allow_lds((const void*)ctrgc_de_tail_tiled_kernel<VV_, RT_>, 136 * 1024, &flag);   /* R = 32: 124 KB; static LDS on top */ \
        hipLaunchKernelGGL((ctrgc_de_tail_tiled_kernel<VV_, RT_>), dim3(d->N * d->S * (VV_ / UT)), dim3(512), lds, (hipStream_t)stream, a); \
        tamgcn_note_kernel("ctrgc_de_tail_tiled_kernel<%d, %d>", VV_, RT_);                                           \
        launched = true;                                                                                              \
    }

extern "C" int tamgcn_ctrgc_tiled_de_tail(const tamgcn_ctrgc_desc* d, const float* dE, float* dA_part, float* dw4_part, float* db4_part,
                                          float* dalpha_part, float* dpq, void* stream) {
    TG_CHECK(d && dE && dA_part && dw4_part && db4_part && dalpha_part && dpq, "tamgcn_ctrgc_tiled_de_tail: null pointer");
    TG_CHECK(d->pq && d->w4 && d->b4 && d->alpha, "tamgcn_ctrgc_tiled_de_tail: null parameter pointer");
    TG_CHECK(tiled_v_ok(d->V), "tamgcn_ctrgc_tiled_de_tail: V=%d (built for V in {32, 64})", d->V);
    TG_CHECK(d->N > 0 && d->S > 0 && d->Cout > 0 && d->Cout % 16 == 0, "tamgcn_ctrgc_tiled_de_tail: bad shape N=%d S=%d Cout=%d", d->N, d->S, d->Cout);
    TG_CHECK(d->R >= 1 && d->R <= 32, "tamgcn_ctrgc_tiled_de_tail: R=%d outside 1..32", d->R);
    TTArgs a;
    a.N = d->N; a.Cout = d->Cout; a.S = d->S; a.R = d->R;
    a.dE = dE; a.pq = d->pq; a.w4 = d->w4; a.b4 = d->b4; a.alpha = d->alpha;
    a.dA_part = dA_part; a.dw4_part = dw4_part; a.db4_part = db4_part; a.dalpha_part = dalpha_part; a.dpq = dpq;
    const int rt = d->R <= 16 ? 1 : 2;
    bool launched = false;
    TL_TAIL_CASE(64, 1) else TL_TAIL_CASE(64, 2) else TL_TAIL_CASE(32, 1) else TL_TAIL_CASE(32, 2)
    TG_CHECK(launched, "tamgcn_ctrgc_tiled_de_tail: no instantiation");
    TG_LAUNCH_CHECK("tamgcn_ctrgc_tiled_de_tail");
    return 0;
}
