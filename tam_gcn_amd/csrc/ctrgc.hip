// Fused CTRGC kernels (reference models/ctrgcn.py:172-177 and the 3-subset sum
// of unit_gcn.forward, :252-254).
//
// One workgroup owns one sample n and a tile of CT=16 output channels, for all
// S subsets and all T frames:
//   1. the channel-wise topology  E_s[c,u,v] = alpha*(W4_s[c,:].tanh(p_s[:,u]-q_s[:,v]) + b4_s[c]) + A_s[u,v]
//      is built once per workgroup into LDS (S*16*V*V floats) and never touches HBM;
//   2. per chunk of BT frames, x3 = W3 x + b3 for the S*16 rows is computed by
//      v_mfma_f32_16x16x4_f32 from an LDS-staged x tile (coalesced along t*V+v)
//      straight into an LDS tile [s*16+c][t][v];
//   3. the V-aggregation  z[c,t,u] = sum_s sum_v E_s[c,u,v]*x3_s[c,t,v]  runs on
//      the VALU with a (TB frames x UB joints) register block per thread, E read
//      as 16-byte LDS vectors;
//   4. z is staged through LDS and written as whole contiguous rows; the train-mode
//      BatchNorm moments of z are accumulated on the way out (per-sample partials).
// The backward kernels reuse the same building blocks:
//   bwd_dx3: dx3_s[c,t,v] = sum_u E_s[c,u,v] dy[c,t,u]          (E^T tiles in LDS)
//   bwd_de : dE_s[c,u,v]  = sum_t dy[c,t,u] x3_s[c,t,v]  (x3 recomputed by MFMA)
//            and the chain through E's definition down to dA, dalpha, dW4, db4, dp, dq.
#include "common.h"

namespace {

constexpr int CT = 16;          // channels per workgroup
constexpr int NT = 256;         // threads
constexpr int SBK = 16;         // K chunk of the x3 GEMM
constexpr int SBKP = SBK + 1;
constexpr int MAXCW = 5;

struct CtrgcArgs {
    int N, Cin, Cout, S, R, T;
    SrcDev x;
    const float* pq; const float* w3; const float* b3; const float* w4; const float* b4;
    const float* A; const float* alpha;
    int nct;                    // Cout / CT
    int pitchB;                 // LDS pitch of the staged x chunk
    int regionB;                // floats of the shared "B" region (x3 tile / stage / D scratch)
};

template <int V, int TB>
struct Geo {
    static constexpr int BT = 4 * TB;               // frames per chunk
    static constexpr int NCOLS = BT * V;            // <= 320
    static constexpr int VV = V * V;
    static constexpr int UB = (V + 3) / 4;          // joints per thread in the aggregation
    static constexpr int UB5 = (V + 4) / 5;         // joints per thread in the dE accumulation
    static constexpr int PX3 = NCOLS;               // pitch of the x3 tile
};

// blockIdx -> (n, channel tile); blocks that share n are b, b+8, ... => same XCD / L2
__device__ __forceinline__ bool block_coords(const CtrgcArgs& a, int& n, int& c0) {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    n = (q / a.nct) * 8 + xcd;
    c0 = (q % a.nct) * CT;
    return n < a.N;
}

// ---------------------------------------------------------------------------
// E tiles.  Es[s][c][u*V+v] (or transposed [v*V+u]).  Dbuf is scratch of `region` floats.
// ---------------------------------------------------------------------------
template <int V>
__device__ void build_E(const CtrgcArgs& a, int n, int c0, float* Es, float* Dbuf, int region, bool transpose) {
    constexpr int VV = V * V;
    const int tid = threadIdx.x;
    const float alpha = a.alpha[0];
    const int RC = min(a.R, region / VV);
    const long long NV = (long long)a.N * V;
    for (int s = 0; s < a.S; ++s) {
        for (int r0 = 0; r0 < a.R; r0 += RC) {
            const int rc = min(RC, a.R - r0);
            __syncthreads();
            for (int e = tid; e < rc * VV; e += NT) {
                int r = e / VV, uv = e - r * VV;
                int u = uv / V, v = uv - u * V;
                float p = a.pq[((long long)(s * 2 + 0) * a.R + r0 + r) * NV + (long long)n * V + u];
                float q = a.pq[((long long)(s * 2 + 1) * a.R + r0 + r) * NV + (long long)n * V + v];
                Dbuf[e] = tanhf(p - q);
            }
            __syncthreads();
            for (int e = tid; e < CT * VV; e += NT) {
                int c = e / VV, uv = e - c * VV;
                const float* w4 = a.w4 + ((long long)s * a.Cout + c0 + c) * a.R + r0;
                float acc = 0.f;
                for (int r = 0; r < rc; ++r) acc = fmaf(w4[r], Dbuf[r * VV + uv], acc);
                int u = uv / V, v = uv - u * V;
                int dst = (s * CT + c) * VV + (transpose ? v * V + u : uv);
                float prev = (r0 == 0) ? 0.f : Es[dst];
                float tot = prev + acc;
                if (r0 + rc >= a.R) tot = alpha * (tot + a.b4[s * a.Cout + c0 + c]) + a.A[s * VV + uv];
                Es[dst] = tot;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// x3 tile for frames [t0, t0+bt): X3[(s*16+c)*PX3 + tl*V + v] = (W3_s x)[c0+c] + b3
// `stage` aliases the X3 tile (it is dead before the tile is written).
// ---------------------------------------------------------------------------
template <int V, int TB>
__device__ void x3_chunk(const CtrgcArgs& a, int n, int c0, int t0, int bt, float* X3) {
    using G = Geo<V, TB>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int ncols = bt * V;
    const int nctile = (ncols + 15) >> 4;
    constexpr int CW = (((G::NCOLS + 15) >> 4) + 3) / 4;
    static_assert(CW <= MAXCW, "chunk too wide");
    const int cw0 = wave * CW;
    int c_act = nctile - cw0; c_act = c_act < 0 ? 0 : (c_act > CW ? CW : c_act);
    const int M3 = a.S * CT;
    float* Bs = X3;                                   // [SBK][pitchB]
    float* As = X3 + SBK * a.pitchB;                  // [M3][SBKP]

    f32x4 acc[TAMGCN_MAX_SUBSETS][CW];
#pragma unroll
    for (int s = 0; s < TAMGCN_MAX_SUBSETS; ++s)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[s][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long long cs = (long long)a.T * V;
    const long long xb = (long long)n * a.x.ctot * cs + (long long)t0 * V;
    for (int k0 = 0; k0 < a.Cin; k0 += SBK) {
        __syncthreads();
        for (int e = tid; e < M3 * SBK; e += NT) {
            int kk = e & (SBK - 1), i = e >> 4;
            int s = i >> 4, c = i & 15, k = k0 + kk;
            As[i * SBKP + kk] = (k < a.Cin) ? a.w3[((long long)s * a.Cout + c0 + c) * a.Cin + k] : 0.f;
        }
        for (int pos = tid; pos < ncols; pos += NT) {
#pragma unroll 4
            for (int kk = 0; kk < SBK; ++kk) {
                int k = k0 + kk;
                float xv = 0.f;
                if (k < a.Cin) { int ch = a.x.coff + k; xv = src_value(a.x, xb + ch * cs + pos, ch); }
                Bs[kk * a.pitchB + pos] = xv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < SBK / 4; ++k4) {
            float av[TAMGCN_MAX_SUBSETS];
#pragma unroll
            for (int s = 0; s < TAMGCN_MAX_SUBSETS; ++s)
                av[s] = (s < a.S) ? As[(s * 16 + j) * SBKP + k4 * 4 + kq] : 0.f;
            const float* brow = Bs + (k4 * 4 + kq) * a.pitchB;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                if (c < c_act) {
                    int col = (cw0 + c) * 16 + j;
                    float bv = brow[col < ncols ? col : 0];
#pragma unroll
                    for (int s = 0; s < TAMGCN_MAX_SUBSETS; ++s)
                        if (s < a.S) acc[s][c] = mfma16(av[s], bv, acc[s][c]);
                }
            }
        }
    }
    __syncthreads();                                   // stage dead; X3 may be overwritten
#pragma unroll
    for (int s = 0; s < TAMGCN_MAX_SUBSETS; ++s) {
        if (s >= a.S) continue;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            if (c >= c_act) continue;
            int col = (cw0 + c) * 16 + j;
            if (col >= ncols) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int cl = kq * 4 + r;
                X3[(s * 16 + cl) * G::PX3 + col] = acc[s][c][r] + a.b3[s * a.Cout + c0 + cl];
            }
        }
    }
    __syncthreads();
}

// out[tt][ub] (+)= sum_b M[c][a0+ub][b] * in[(tq*TB+tt)*V + b]   (a0 = uq*UB)
template <int V, int TB>
__device__ __forceinline__ void aggregate(const float* Mc, const float* inrow, int a0, float (&out)[TB][(V + 3) / 4]) {
    constexpr int UB = (V + 3) / 4;
    float xin[TB][V];
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
        if constexpr (V % 4 == 0) {
#pragma unroll
            for (int b = 0; b < V; b += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(inrow + tt * V + b);
                xin[tt][b] = t[0]; xin[tt][b + 1] = t[1]; xin[tt][b + 2] = t[2]; xin[tt][b + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int b = 0; b < V; ++b) xin[tt][b] = inrow[tt * V + b];
        }
    }
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
        int aidx = a0 + ub;
        if (aidx >= V) break;
        const float* mrow = Mc + aidx * V;
        if constexpr (V % 4 == 0) {
#pragma unroll
            for (int b = 0; b < V; b += 4) {
                f32x4 m = *reinterpret_cast<const f32x4*>(mrow + b);
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    out[tt][ub] = fmaf(m[0], xin[tt][b], out[tt][ub]);
                    out[tt][ub] = fmaf(m[1], xin[tt][b + 1], out[tt][ub]);
                    out[tt][ub] = fmaf(m[2], xin[tt][b + 2], out[tt][ub]);
                    out[tt][ub] = fmaf(m[3], xin[tt][b + 3], out[tt][ub]);
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < V; ++b) {
                float m = mrow[b];
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) out[tt][ub] = fmaf(m, xin[tt][b], out[tt][ub]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int V, int TB>
__global__ __launch_bounds__(NT) void ctrgc_fwd_kernel(const CtrgcArgs a, float* y, float* stats_part) {
    using G = Geo<V, TB>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords(a, n, c0)) return;
    float* Es = smem;                                  // [S][CT][VV]
    float* X3 = Es + a.S * CT * G::VV;                 // regionB floats
    float* Zs = X3 + a.regionB;                        // [CT][NCOLS]
    const int tid = threadIdx.x;
    const int c = tid >> 4, tq = (tid >> 2) & 3, uq = tid & 3;
    const int l16 = tid & 15;

    build_E<V>(a, n, c0, Es, X3, a.regionB, false);

    float st1 = 0.f, st2 = 0.f;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        x3_chunk<V, TB>(a, n, c0, t0, bt, X3);
        float z[TB][G::UB];
#pragma unroll
        for (int tt = 0; tt < TB; ++tt)
#pragma unroll
            for (int ub = 0; ub < G::UB; ++ub) z[tt][ub] = 0.f;
        if (tq * TB < bt) {       // rows beyond bt hold stale data: results are discarded below
            for (int s = 0; s < a.S; ++s)
                aggregate<V, TB>(Es + (s * CT + c) * G::VV, X3 + (s * 16 + c) * G::PX3 + tq * TB * V, uq * G::UB, z);
        }
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            int tl = tq * TB + tt;
            if (tl < bt) {
#pragma unroll
                for (int ub = 0; ub < G::UB; ++ub) {
                    int u = uq * G::UB + ub;
                    if (u < V) Zs[c * G::NCOLS + tl * V + u] = z[tt][ub];
                }
            }
        }
        __syncthreads();
        float* yrow = y + (((long long)n * a.Cout + c0 + c) * a.T + t0) * V;
        for (int p = l16; p < ncols; p += 16) {
            float v = Zs[c * G::NCOLS + p];
            yrow[p] = v;
            st1 += v;
            st2 = fmaf(v, v, st2);
        }
        // next chunk's first barrier (inside x3_chunk) protects Zs / X3 reuse
    }
    if (stats_part) {
        st1 = wave_sum16(st1);
        st2 = wave_sum16(st2);
        if (l16 == 0) {
            stats_part[((long long)0 * a.Cout + c0 + c) * a.N + n] = st1;
            stats_part[((long long)1 * a.Cout + c0 + c) * a.N + n] = st2;
        }
    }
}

// ---------------------------------------------------------------------------
// backward 1: dx3
// ---------------------------------------------------------------------------
template <int V, int TB>
__global__ __launch_bounds__(NT) void ctrgc_bwd_dx3_kernel(const CtrgcArgs a, const SrcDev dy, float* dx3, float* db3_part) {
    using G = Geo<V, TB>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords(a, n, c0)) return;
    float* Es = smem;                                  // transposed tiles [S][CT][v][u]
    float* X3 = Es + a.S * CT * G::VV;                 // output staging [S*16][PX3]
    float* Zs = X3 + a.regionB;                        // dy chunk [CT][NCOLS]
    const int tid = threadIdx.x;
    const int c = tid >> 4, tq = (tid >> 2) & 3, vq = tid & 3;
    const int l16 = tid & 15;

    build_E<V>(a, n, c0, Es, X3, a.regionB, true);

    float sb[TAMGCN_MAX_SUBSETS] = {0.f, 0.f, 0.f};
    const long long dcs = (long long)a.T * V;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        __syncthreads();
        {
            int ch = dy.coff + c0 + c;
            long long base = ((long long)n * dy.ctot + ch) * dcs + (long long)t0 * V;
            for (int p = l16; p < ncols; p += 16) Zs[c * G::NCOLS + p] = src_value(dy, base + p, ch);
        }
        __syncthreads();
        if (tq * TB < bt) {
            for (int s = 0; s < a.S; ++s) {
                float o[TB][G::UB];
#pragma unroll
                for (int tt = 0; tt < TB; ++tt)
#pragma unroll
                    for (int ub = 0; ub < G::UB; ++ub) o[tt][ub] = 0.f;
                aggregate<V, TB>(Es + (s * CT + c) * G::VV, Zs + c * G::NCOLS + tq * TB * V, vq * G::UB, o);
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    int tl = tq * TB + tt;
                    if (tl < bt) {
#pragma unroll
                        for (int ub = 0; ub < G::UB; ++ub) {
                            int v = vq * G::UB + ub;
                            if (v < V) { X3[(s * 16 + c) * G::PX3 + tl * V + v] = o[tt][ub]; sb[s] += o[tt][ub]; }
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int s = 0; s < a.S; ++s) {
            float* orow = dx3 + (((long long)n * a.S * a.Cout + s * a.Cout + c0 + c) * a.T + t0) * V;
            for (int p = l16; p < ncols; p += 16) orow[p] = X3[(s * 16 + c) * G::PX3 + p];
        }
    }
    if (db3_part) {
        for (int s = 0; s < a.S; ++s) {
            float v = wave_sum16(sb[s]);
            if (l16 == 0) db3_part[(long long)n * a.S * a.Cout + s * a.Cout + c0 + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// backward 2: dE and everything behind it
// ---------------------------------------------------------------------------
template <int V, int TB>
__global__ __launch_bounds__(NT) void ctrgc_bwd_de_kernel(const CtrgcArgs a, const SrcDev dy, float* dA_part, float* dw4_part,
                                                          float* db4_part, float* dalpha_part, float* dpq) {
    using G = Geo<V, TB>;
    constexpr int VV = G::VV, UB5 = G::UB5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red_alpha[4];
    int n, c0;
    if (!block_coords(a, n, c0)) return;
    float* DE = smem;                                  // [S][CT][VV]
    float* X3 = DE + a.S * CT * VV;                    // x3 tile / later D scratch
    float* Zs = X3 + a.regionB;                        // dy chunk [CT][NCOLS]
    const int tid = threadIdx.x;
    const int l16 = tid & 15;
    // dE ownership: thread -> (s, c, u-group of UB5 joints)
    const int own_s = tid / (CT * 5), own_c = (tid / 5) % CT, own_g = tid % 5;
    const bool owner = own_s < a.S;
    float dE[UB5][V];
#pragma unroll
    for (int i = 0; i < UB5; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) dE[i][v] = 0.f;

    const long long dcs = (long long)a.T * V;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        x3_chunk<V, TB>(a, n, c0, t0, bt, X3);         // begins with a barrier: previous chunk fully consumed
        {
            int c = tid >> 4;
            int ch = dy.coff + c0 + c;
            long long base = ((long long)n * dy.ctot + ch) * dcs + (long long)t0 * V;
            for (int p = l16; p < ncols; p += 16) Zs[c * G::NCOLS + p] = src_value(dy, base + p, ch);
        }
        __syncthreads();
        if (owner) {
            const float* xr = X3 + (own_s * 16 + own_c) * G::PX3;
            const float* dr = Zs + own_c * G::NCOLS;
            for (int tl = 0; tl < bt; ++tl) {
                float xv[V];
#pragma unroll
                for (int v = 0; v < V; ++v) xv[v] = xr[tl * V + v];
#pragma unroll
                for (int i = 0; i < UB5; ++i) {
                    int u = own_g * UB5 + i;
                    float d = (u < V) ? dr[tl * V + u] : 0.f;
#pragma unroll
                    for (int v = 0; v < V; ++v) dE[i][v] = fmaf(d, xv[v], dE[i][v]);
                }
            }
        }
    }
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int i = 0; i < UB5; ++i) {
            int u = own_g * UB5 + i;
            if (u < V) {
#pragma unroll
                for (int v = 0; v < V; ++v) DE[(own_s * CT + own_c) * VV + u * V + v] = dE[i][v];
            }
        }
    }
    __syncthreads();
    // (a) dA partial: sum over this block's channels
    {
        const int blk = n * a.nct + c0 / CT;
        for (int e = tid; e < a.S * VV; e += NT) {
            int s = e / VV, uv = e - s * VV;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc += DE[(s * CT + c) * VV + uv];
            dA_part[(long long)blk * a.S * VV + e] = acc;
        }
    }
    // (b) chain through conv4 / tanh, subset by subset, rel-channel chunk by chunk
    const float alpha = a.alpha[0];
    const int RC = min(min(a.R, 8), a.regionB / VV);
    const long long NV = (long long)a.N * V;
    float dalpha_acc = 0.f;
    const int c = tid >> 4;
    for (int s = 0; s < a.S; ++s) {
        float db4raw = 0.f;
        for (int uv = l16; uv < VV; uv += 16) db4raw += DE[(s * CT + c) * VV + uv];
        db4raw = wave_sum16(db4raw);
        if (l16 == 0) {
            db4_part[((long long)n * a.S + s) * a.Cout + c0 + c] = alpha * db4raw;
            dalpha_acc = fmaf(a.b4[s * a.Cout + c0 + c], db4raw, dalpha_acc);
        }
        for (int r0 = 0; r0 < a.R; r0 += RC) {
            const int rc = min(RC, a.R - r0);
            __syncthreads();
            for (int e = tid; e < rc * VV; e += NT) {
                int r = e / VV, uv = e - r * VV;
                int u = uv / V, v = uv - u * V;
                float p = a.pq[((long long)(s * 2 + 0) * a.R + r0 + r) * NV + (long long)n * V + u];
                float q = a.pq[((long long)(s * 2 + 1) * a.R + r0 + r) * NV + (long long)n * V + v];
                X3[e] = tanhf(p - q);
            }
            __syncthreads();
            // dW4raw[c][r] = sum_uv dE[c][uv] * D[r][uv]
            float wacc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) wacc[r] = 0.f;
            for (int uv = l16; uv < VV; uv += 16) {
                float de = DE[(s * CT + c) * VV + uv];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < rc) wacc[r] = fmaf(de, X3[r * VV + uv], wacc[r]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float wsum = wave_sum16(wacc[r]);
                if (l16 == 0 && r < rc) {
                    long long wi = ((long long)s * a.Cout + c0 + c) * a.R + r0 + r;
                    dw4_part[(long long)n * a.S * a.Cout * a.R + wi] = alpha * wsum;
                    dalpha_acc = fmaf(a.w4[wi], wsum, dalpha_acc);
                }
            }
            __syncthreads();
            // dS[r][uv] = alpha * (sum_c W4[c][r] dE[c][uv]) * (1 - D^2), in place over D
            for (int e = tid; e < rc * VV; e += NT) {
                int r = e / VV, uv = e - r * VV;
                float acc = 0.f;
#pragma unroll
                for (int cc = 0; cc < CT; ++cc)
                    acc = fmaf(a.w4[((long long)s * a.Cout + c0 + cc) * a.R + r0 + r], DE[(s * CT + cc) * VV + uv], acc);
                float d = X3[e];
                X3[e] = alpha * acc * (1.f - d * d);
            }
            __syncthreads();
            // dp[r][u] = sum_v dS ; dq[r][v] = -sum_u dS   (other channel tiles add to the same slots)
            for (int e = tid; e < rc * V * 2; e += NT) {
                int which = e / (rc * V);
                int rem = e - which * rc * V;
                int r = rem / V, k = rem - r * V;
                float acc = 0.f;
                if (which == 0) { for (int v = 0; v < V; ++v) acc += X3[r * VV + k * V + v]; }
                else { for (int u = 0; u < V; ++u) acc -= X3[r * VV + u * V + k]; }
                atomicAdd(&dpq[((long long)(s * 2 + which) * a.R + r0 + r) * NV + (long long)n * V + k], acc);
            }
        }
    }
    // dalpha partial of this block
    dalpha_acc = wave_sum64(dalpha_acc);
    if ((tid & 63) == 0) red_alpha[tid >> 6] = dalpha_acc;
    __syncthreads();
    if (tid == 0) dalpha_part[n * a.nct + c0 / CT] = red_alpha[0] + red_alpha[1] + red_alpha[2] + red_alpha[3];
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
struct CtrgcPlan { int TB, pitchB, regionB; size_t lds; };

template <int V, int TB>
static bool plan_for(int S, CtrgcPlan* p) {
    using G = Geo<V, TB>;
    int pitch = G::NCOLS;
    pitch += ((16 - (pitch & 31)) + 32) & 31;
    int stage = SBK * pitch + S * CT * SBKP;
    int x3 = S * 16 * G::PX3;
    int region = stage > x3 ? stage : x3;
    region = (region + 3) & ~3;
    if (region < G::VV) return false;
    size_t lds = sizeof(float) * ((size_t)S * CT * G::VV + region + (size_t)CT * G::NCOLS);
    p->TB = TB; p->pitchB = pitch; p->regionB = region; p->lds = lds;
    return lds <= 160 * 1024;
}

static int plan_ctrgc(int S, int V, CtrgcPlan* p) {
    if (S < 1 || S > TAMGCN_MAX_SUBSETS) return -1;
    switch (V) {
        case 20: return plan_for<20, 4>(S, p) ? 0 : -1;
        case 25: return plan_for<25, 1>(S, p) ? 0 : -1;
        default: return -1;
    }
}

static int fill_args(const tamgcn_ctrgc_desc* d, const CtrgcPlan& p, CtrgcArgs* a, const char* who) {
    if (!(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->R > 0 && d->T > 0)) { tamgcn_set_error("%s: bad dims", who); return -1; }
    if (d->Cout % CT) { tamgcn_set_error("%s: Cout=%d must be a multiple of %d", who, d->Cout, CT); return -1; }
    if (!(d->x.x1 && d->pq && d->w3 && d->b3 && d->w4 && d->b4 && d->A && d->alpha)) { tamgcn_set_error("%s: null pointer", who); return -1; }
    if (d->x.coff + d->Cin > d->x.ctot) { tamgcn_set_error("%s: x channel slice out of range", who); return -1; }
    a->N = d->N; a->Cin = d->Cin; a->Cout = d->Cout; a->S = d->S; a->R = d->R; a->T = d->T;
    a->x = make_src(d->x);
    a->pq = d->pq; a->w3 = d->w3; a->b3 = d->b3; a->w4 = d->w4; a->b4 = d->b4; a->A = d->A; a->alpha = d->alpha;
    a->nct = d->Cout / CT; a->pitchB = p.pitchB; a->regionB = p.regionB;
    return 0;
}

static unsigned grid_blocks(const CtrgcArgs& a) { return 8u * (unsigned)ceil_div(a.N, 8) * (unsigned)a.nct; }

template <typename K>
static void allow_lds(K kernel, size_t lds, bool* done) {
    if (!*done) {   // once per instantiation; not a stream operation
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        *done = true;
    }
}

#define CTRGC_DISPATCH(KERNEL, ...)                                                                          \
    do {                                                                                                     \
        static bool set20 = false, set25 = false;                                                            \
        if (d->V == 20) {                                                                                    \
            allow_lds(KERNEL<20, 4>, p.lds, &set20);                                                         \
            hipLaunchKernelGGL((KERNEL<20, 4>), dim3(grid_blocks(a)), dim3(NT), p.lds, (hipStream_t)stream, __VA_ARGS__); \
        } else {                                                                                             \
            allow_lds(KERNEL<25, 1>, p.lds, &set25);                                                         \
            hipLaunchKernelGGL((KERNEL<25, 1>), dim3(grid_blocks(a)), dim3(NT), p.lds, (hipStream_t)stream, __VA_ARGS__); \
        }                                                                                                    \
    } while (0)

}  // namespace

extern "C" int tamgcn_ctrgc_lds_bytes(int S, int V, int R) {
    (void)R;
    CtrgcPlan p;
    if (plan_ctrgc(S, V, &p)) return -1;
    return (int)p.lds;
}

extern "C" int tamgcn_ctrgc_fwd(const tamgcn_ctrgc_desc* d, float* y, float* stats_part, void* stream) {
    TG_CHECK(d && y, "tamgcn_ctrgc_fwd: null pointer");
    CtrgcPlan p;
    TG_CHECK(plan_ctrgc(d->S, d->V, &p) == 0, "tamgcn_ctrgc_fwd: unsupported S=%d V=%d (LDS-resident tiles exist for V in {20,25})", d->S, d->V);
    CtrgcArgs a;
    if (fill_args(d, p, &a, "tamgcn_ctrgc_fwd")) return -1;
    CTRGC_DISPATCH(ctrgc_fwd_kernel, a, y, stats_part);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_fwd");
    return 0;
}

extern "C" int tamgcn_ctrgc_bwd_dx3(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, float* dx3, float* db3_part, void* stream) {
    TG_CHECK(d && dy && dy->x1 && dx3, "tamgcn_ctrgc_bwd_dx3: null pointer");
    CtrgcPlan p;
    TG_CHECK(plan_ctrgc(d->S, d->V, &p) == 0, "tamgcn_ctrgc_bwd_dx3: unsupported S=%d V=%d", d->S, d->V);
    CtrgcArgs a;
    if (fill_args(d, p, &a, "tamgcn_ctrgc_bwd_dx3")) return -1;
    CTRGC_DISPATCH(ctrgc_bwd_dx3_kernel, a, make_src(*dy), dx3, db3_part);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_bwd_dx3");
    return 0;
}

extern "C" int tamgcn_ctrgc_bwd_de(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, float* dA_part, float* dw4_part,
                                   float* db4_part, float* dalpha_part, float* dpq, void* stream) {
    TG_CHECK(d && dy && dy->x1 && dA_part && dw4_part && db4_part && dalpha_part && dpq, "tamgcn_ctrgc_bwd_de: null pointer");
    CtrgcPlan p;
    TG_CHECK(plan_ctrgc(d->S, d->V, &p) == 0, "tamgcn_ctrgc_bwd_de: unsupported S=%d V=%d", d->S, d->V);
    CtrgcArgs a;
    if (fill_args(d, p, &a, "tamgcn_ctrgc_bwd_de")) return -1;
    CTRGC_DISPATCH(ctrgc_bwd_de_kernel, a, make_src(*dy), dA_part, dw4_part, db4_part, dalpha_part, dpq);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_bwd_de");
    return 0;
}
