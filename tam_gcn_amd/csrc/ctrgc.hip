// Fused CTRGC kernels (reference models/ctrgcn.py:172-177 and the 3-subset sum
// of unit_gcn.forward, :252-254).
//
// One workgroup owns one sample n and a tile of CT=16 output channels, for all
// S subsets and all T frames:
//   1. the channel-wise topology  E_s[c,u,v] = alpha*(W4_s[c,:].tanh(p_s[:,u]-q_s[:,v]) + b4_s[c]) + A_s[u,v]
//      is built once per workgroup into LDS (S*16*V*V floats) and never touches HBM:
//      D = tanh(p_u - q_v) goes to an LDS tile, W4.D runs on v_mfma_f32_16x16x4_f32;
//   2. per chunk of BT frames, x3 = W3 x + b3 for the S*16 rows is an MFMA GEMM over an
//      LDS-staged x tile: 16-byte global loads (coalesced along t*V+v) are prefetched into
//      registers one K chunk (32 channels) ahead, so HBM/L2 latency hides under the MFMAs;
//      the result lands in an LDS tile [s*16+c][t][v];
//   3. the V-aggregation  z[c,t,u] = sum_s sum_v E_s[c,u,v]*x3_s[c,t,v]  runs on the VALU with
//      a (TB frames x UB joints) register block per thread, E read as 16-byte LDS vectors;
//   4. z is staged through LDS and written as whole contiguous rows; the train-mode
//      BatchNorm moments of z are accumulated on the way out (per-sample partials).
// The backward kernels reuse the same building blocks:
//   bwd_dx3: dx3_s[c,t,v] = sum_u E_s[c,u,v] dy[c,t,u]          (E^T tiles in LDS)
//   bwd_de : dE_s[c,u,v]  = sum_t dy[c,t,u] x3_s[c,t,v]  (x3 recomputed by MFMA)
//            and the chain through E's definition down to dA, dalpha, dW4, db4, dp, dq.
// All MFMA blocks are branch-free with compile-time tile counts (padding tiles are computed
// and discarded): per-MFMA guards made hipcc serialise every ds_read/MFMA pair.
#include "common.h"

TG_TRACE_DEFINE(tamgcn_trace_read_ctrgc)

namespace {

constexpr int CT = 16;          // channels per workgroup
constexpr int SBK = 32;         // K chunk of the x3 GEMM
constexpr int SBKP = SBK + 2;   // pitch/2 odd => the 16x4 A-fragment column reads hit 32 distinct banks

struct CtrgcArgs {
    int N, Cin, Cout, S, R, T;
    const float* x; int x_ctot, x_coff;
    const float* pq; const float* w3; const float* b3; const float* w4; const float* b4;
    const float* A; const float* alpha;
    const float* E;             // (N, S, Cout, V*V) from tamgcn_ctrgc_build_e, or null: build the tiles on chip
    int nct;                    // Cout / CT
    int pitchB;                 // LDS pitch of the staged x chunk
    int regionB;                // floats of the shared "B" region (x3 tile / stage / D scratch)
};

// V joints; TB frames per thread in the aggregation; NTQ frame groups => NT = 16*NTQ*4 threads
template <int V_, int TB_, int NTQ_>
struct Geo {
    static constexpr int V = V_, TB = TB_, NTQ = NTQ_;
    static constexpr int NT = CT * NTQ * 4;         // threads
    static constexpr int NW = NT / 64;              // waves
    static constexpr int BT = NTQ * TB;             // frames per chunk
    static constexpr int NCOLS = BT * V;            // <= 320
    static constexpr int NCT = (NCOLS + 15) / 16;   // 16-wide column tiles of the x3 GEMM
    static constexpr int CW = (NCT + NW - 1) / NW;  // column tiles per wave
    static constexpr int VV = V * V;
    static constexpr int UB = (V + 3) / 4;          // joints per thread in the aggregation
    static constexpr int UB5 = (V + 4) / 5;         // joints per thread in the dE accumulation
    static constexpr int PX3 = NCOLS;               // pitch of the x3 tile
    static constexpr bool VEC = (V % 4) == 0;
    static constexpr int NPF = VEC ? (SBK * (NCOLS / 4) + NT - 1) / NT : (SBK * NCOLS + NT - 1) / NT;
};

// blockIdx -> (n, channel tile); blocks that share n are b, b+8, ... => same XCD / L2
__device__ __forceinline__ bool block_coords(const CtrgcArgs& a, int& n, int& c0) {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    n = (q / a.nct) * 8 + xcd;
    c0 = (q % a.nct) * CT;
    return n < a.N;
}



// D[r][u*V+v] = tanh(p[r][u] - q[r][v]) for rel-channels r0..r0+rc-1 of subset s, sample n.
// Four independent (p, q) pairs are in flight per thread so the L2 latency is paid once per
// batch, not once per element.
template <class G>
__device__ __forceinline__ void fill_D(const CtrgcArgs& a, int n, int s, int r0, int rc, float* Dbuf) {
    constexpr int V = G::V, VV = G::VV, NT = G::NT;
    const long long NV = (long long)a.N * V;
    const float* pb = a.pq + ((long long)(s * 2 + 0) * a.R + r0) * NV + (long long)n * V;
    const float* qb = a.pq + ((long long)(s * 2 + 1) * a.R + r0) * NV + (long long)n * V;
    const int total = rc * VV;
    for (int e0 = threadIdx.x; e0 < total; e0 += 4 * NT) {
        float pv[4], qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = e0 + i * NT;
            int ec = e < total ? e : 0;
            int r = ec / VV, uv = ec - r * VV;
            int u = uv / V, v = uv - u * V;
            pv[i] = pb[r * NV + u];
            qv[i] = qb[r * NV + v];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = e0 + i * NT;
            if (e < total) Dbuf[e] = fast_tanh(pv[i] - qv[i]);
        }
    }
}

// ---------------------------------------------------------------------------
// E tiles.  Es[s][c][u*V+v] (or transposed [v*V+u]).  Dbuf is scratch of `region` floats.
// D chunk [rc][VV] -> LDS, then E(16 x VV) += W4(16 x rc) . D  on MFMA (rows = channels).
// ---------------------------------------------------------------------------
template <class G>
__device__ void build_E(const CtrgcArgs& a, int n, int c0, float* Es, float* Dbuf, int region, bool transpose) {
    constexpr int V = G::V, VV = G::VV, NT = G::NT, NW = G::NW;
    constexpr int NTILE = (VV + 15) / 16, NIT = (NTILE + NW - 1) / NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const float alpha = a.alpha[0];
    const long long NV = (long long)a.N * V;
    // p / q of this sample for every subset go to LDS in ONE batch of loads (the D fill used to pay an
    // L2 round trip per four elements: 27 % of the forward kernel at C = 64, 60 % at C = 256).
    const int per = 2 * a.R * V;                      // [p|q][r][v] of one subset
    const bool all = region - ((a.S * per + 3) & ~3) >= 4 * VV;   // every subset at once, else one at a time
    const int nst = ((all ? a.S * per : per) + 3) & ~3;
    float* PQ = Dbuf + region - nst;
    auto stage = [&](int s0, int cnt) {
        constexpr int MAXL = 8;
        for (int e0 = tid; e0 < cnt; e0 += MAXL * NT) {
            float t[MAXL];
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int e = e0 + i * NT;
                const int row = (e < cnt ? e : 0) / V, v = (e < cnt ? e : 0) - row * V;
                t[i] = a.pq[((long long)s0 * 2 * a.R + row) * NV + (long long)n * V + v];
            }
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int e = e0 + i * NT;
                if (e < cnt) PQ[e] = t[i];
            }
        }
    };
    if (all) stage(0, a.S * per);
    int RC = min(min(a.R, 16), (region - nst) / VV) & ~3;   // rel-channels per pass (<= 16: four A registers)
    if (RC < 4) RC = 4;
    for (int s = 0; s < a.S; ++s) {
        if (!all) { __syncthreads(); stage(s, per); }  // previous subset's passes are done with PQ
        const float* Pp = PQ + (all ? s * per : 0);
        const float* Qp = Pp + a.R * V;
        for (int r0 = 0; r0 < a.R; r0 += RC) {
            const int rc = min(RC, a.R - r0);
            // Everything this pass needs from global memory is requested BEFORE the tanh fill and consumed
            // after it: the W4 fragment, and on the last pass b4 and the A entries of this wave's tiles.
            const bool last = r0 + rc >= a.R;
            float aw[4], b4r[4], Ar[NIT];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                aw[k] = (k * 4 + kq < rc) ? a.w4[((long long)s * a.Cout + c0 + j) * a.R + r0 + k * 4 + kq] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) b4r[r] = last ? a.b4[s * a.Cout + c0 + kq * 4 + r] : 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int col = (wave + it * NW) * 16 + j;
                Ar[it] = (last && col < VV) ? a.A[s * VV + col] : 0.f;
            }
            __syncthreads();                           // PQ staged / previous pass done with Dbuf
            for (int e = tid; e < rc * VV; e += NT) {
                const int r = e / VV, uv = e - r * VV;
                const int u = uv / V, v = uv - u * V;
                Dbuf[e] = fast_tanh(Pp[(r0 + r) * V + u] - Qp[(r0 + r) * V + v]);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int ct = wave + it * NW;
                if (ct < NTILE) {
                    const int col = ct * 16 + j;
                    const int colc = col < VV ? col : 0;
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k * 4 < rc) acc = mfma16(aw[k], Dbuf[(k * 4 + kq) * VV + colc], acc);
                    if (col < VV) {
                        const int u = col / V, v = col - u * V;
                        const int off = transpose ? v * V + u : col;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int dst = (s * CT + kq * 4 + r) * VV + off;
                            float tot = acc[r] + (r0 == 0 ? 0.f : Es[dst]);
                            if (last) tot = alpha * (tot + b4r[r]) + Ar[it];
                            Es[dst] = tot;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
}

// E tiles of channels c0..c0+15 from the tensor tamgcn_ctrgc_build_e wrote, (N, S, Cout, V*V): per subset the
// 16 rows are one contiguous run, fetched with 16-byte loads in a single batch (the on-chip builder above costs
// 14 % of the forward kernel at C = 64 and 45 % at C = 256, where 16 channel tiles repeat the same tanh work).
template <class G, int ST>
__device__ __forceinline__ void load_E(const float* __restrict__ Eg, int N_unused, int Cout, int n, int c0, float* Es, bool transpose) {
    constexpr int V = G::V, VV = G::VV, NT = G::NT;
    constexpr int PER = CT * VV / 4;                   // float4 per subset
    constexpr int NL = (ST * PER + NT - 1) / NT;
    (void)N_unused;
    float4 t[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = threadIdx.x + i * NT;
        const int sidx = (e < ST * PER ? e : 0) / PER, r = (e < ST * PER ? e : 0) - sidx * PER;
        t[i] = reinterpret_cast<const float4*>(Eg + (((long long)n * ST + sidx) * Cout + c0) * VV)[r];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = threadIdx.x + i * NT;
        if (e < ST * PER) {
            if (!transpose) {
                reinterpret_cast<float4*>(Es)[e] = t[i];          // same linear order: [s][c][uv]
            } else {
                const float vals[4] = {t[i].x, t[i].y, t[i].z, t[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = e * 4 + k;
                    const int row = f / VV, uv = f - row * VV;   // row = s*16 + c
                    const int u = uv / V, v = uv - u * V;
                    Es[row * VV + v * V + u] = vals[k];
                }
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// E for every channel of one (sample, subset) -- tamgcn_ctrgc_build_e.  D = tanh(p_u - q_v) is built once
// in LDS and reused by all Cout/16 channel tiles; tiles leave through LDS as 16-byte coalesced rows.
// ---------------------------------------------------------------------------
struct EArgs {
    int N, Cout, S, R;
    const float* pq; const float* w4; const float* b4; const float* A; const float* alpha;
    float* E;
};

template <int V>
__global__ __launch_bounds__(512) void ctrgc_E_kernel(const EArgs a) {
    constexpr int VV = V * V, NT = 512, NW = 8, NTILE = (VV + 15) / 16, NIT = (NTILE + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Tt = smem;                         // [16][VV] finished tile
    float* Ds = Tt + 16 * VV;                 // [R][VV]
    float* PQ = Ds + a.R * VV;                // [p|q][R][V]
    const int n = blockIdx.x / a.S, s = blockIdx.x - n * a.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const long long NV = (long long)a.N * V;
    const float alpha = a.alpha[0];
    {
        constexpr int MAXL = 4;
        const int cnt = 2 * a.R * V;
        for (int e0 = tid; e0 < cnt; e0 += MAXL * NT) {
            float t[MAXL];
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int e = e0 + i * NT;
                const int row = (e < cnt ? e : 0) / V, v = (e < cnt ? e : 0) - row * V;
                t[i] = a.pq[((long long)s * 2 * a.R + row) * NV + (long long)n * V + v];
            }
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int e = e0 + i * NT;
                if (e < cnt) PQ[e] = t[i];
            }
        }
    }
    float Ar[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (wave + it * NW) * 16 + j;
        Ar[it] = col < VV ? a.A[s * VV + col] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < a.R * VV; e += NT) {
        const int r = e / VV, uv = e - r * VV;
        const int u = uv / V, v = uv - u * V;
        Ds[e] = fast_tanh(PQ[r * V + u] - PQ[(a.R + r) * V + v]);
    }
    float aw[8], b4r[4];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            aw[k] = (k * 4 + kq < a.R) ? a.w4[((long long)s * a.Cout + c0 + j) * a.R + k * 4 + kq] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) b4r[r] = a.b4[s * a.Cout + c0 + kq * 4 + r];
    };
    fetch(0);
    __syncthreads();
    float* Eg = a.E + ((long long)n * a.S + s) * a.Cout * VV;
    for (int c0 = 0; c0 < a.Cout; c0 += 16) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int ct = wave + it * NW;
            if (ct < NTILE) {
                const int col = ct * 16 + j;
                const int colc = col < VV ? col : 0;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k * 4 < a.R) acc = mfma16(aw[k], Ds[(k * 4 + kq) * VV + colc], acc);
                if (col < VV) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tt[(kq * 4 + r) * VV + col] = alpha * (acc[r] + b4r[r]) + Ar[it];
                }
            }
        }
        if (c0 + 16 < a.Cout) fetch(c0 + 16);           // in flight under the store pass
        __syncthreads();
        for (int e = tid; e < 16 * VV / 4; e += NT)
            reinterpret_cast<float4*>(Eg + (long long)c0 * VV)[e] = reinterpret_cast<const float4*>(Tt)[e];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// x3 tile for frames [t0, t0+bt): X3[(s*16+c)*PX3 + tl*V + v] = (W3_s x)[c0+c] + b3
// The staging buffers alias the X3 tile (they are dead before the tile is written).
// ---------------------------------------------------------------------------
template <class G, int ST, int SPL>   // SPL: 0 exact fp32-input MFMA, 2 / 3 = two- / three-term bf16 split
__device__ void x3_chunk(const CtrgcArgs& a, int n, int c0, int t0, int bt, float* X3) {
    constexpr int V = G::V, NT = G::NT, CW = G::CW, NPF = G::NPF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int ncols = bt * V;
    // Column tiles per wave.  Waves w and w + 4 share a SIMD (its MFMA pipe): with 20 tiles on 8 waves, "3 per wave in
    // order" loads the four SIMDs 6/6/5/3; give every SIMD 5 instead -- wave w < 4 takes three, wave w + 4 two.
    constexpr bool BAL = (G::NW == 8 && G::NCT == 20);
    const int cw0 = BAL ? 5 * (wave & 3) + (wave < 4 ? 0 : 3) : wave * CW;
    const bool third = !BAL || wave < 4;               // wave-uniform: does tile c = 2 exist for this wave
    float* Bs = X3;                                   // [SBK][pitchB]
    float* As = X3 + SBK * a.pitchB;                  // [ST*16][SBKP]

    f32x4 acc[ST][CW];
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[s][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int bcol[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { int col = (cw0 + c) * 16 + j; bcol[c] = (col < ncols && (c < 2 || third)) ? col : 0; }

    const long long cs = (long long)a.T * V;
    const long long xb = ((long long)n * a.x_ctot + a.x_coff) * cs + (long long)t0 * V;
    // prefetch descriptors: element e -> (row kk, position pos) of the [SBK][NCOLS] chunk
    int p_kk[NPF], p_pos[NPF];
    constexpr int ROWV = G::VEC ? G::NCOLS / 4 : G::NCOLS;        // vectors per row (full chunk geometry)
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        int e = tid + i * NT;
        int kk = e / ROWV, pv = e - kk * ROWV;
        p_kk[i] = kk < SBK ? kk : -1;
        p_pos[i] = G::VEC ? pv * 4 : pv;
    }
    float4 rv[G::VEC ? NPF : 1];
    float rs[G::VEC ? 1 : NPF];
    constexpr int NAF = (ST * 16 * SBK + NT - 1) / NT;            // weight-tile values per thread
    float wv[NAF];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NAF; ++i) {
            int e = tid + i * NT;
            int kk = e & (SBK - 1), row = e >> 5;
            int sidx = row >> 4, c = row & 15, k = k0 + kk;
            wv[i] = (row < ST * 16 && k < a.Cin) ? a.w3[((long long)sidx * a.Cout + c0 + c) * a.Cin + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int k = k0 + p_kk[i];
            const bool ok = p_kk[i] >= 0 && k < a.Cin && p_pos[i] < ncols;
            if constexpr (G::VEC) {
                rv[i] = ok ? *reinterpret_cast<const float4*>(a.x + xb + (long long)k * cs + p_pos[i])
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                rs[i] = ok ? a.x[xb + (long long)k * cs + p_pos[i]] : 0.f;
            }
        }
    };
    float b3r[ST][4];                                  // fetched here: in flight under the K loop, not exposed after it
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) b3r[s][r] = a.b3[s * a.Cout + c0 + kq * 4 + r];
    prefetch(0);
    for (int k0 = 0; k0 < a.Cin; k0 += SBK) {
        __syncthreads();                               // previous users of the region are done
#pragma unroll
        for (int i = 0; i < NAF; ++i) {
            int e = tid + i * NT;
            int kk = e & (SBK - 1), row = e >> 5;
            if (row < ST * 16) As[row * SBKP + kk] = wv[i];
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            if (p_kk[i] >= 0) {
                if constexpr (G::VEC) *reinterpret_cast<float4*>(Bs + p_kk[i] * a.pitchB + p_pos[i]) = rv[i];
                else Bs[p_kk[i] * a.pitchB + p_pos[i]] = rs[i];
            }
        }
        __syncthreads();
        if (k0 + SBK < a.Cin) prefetch(k0 + SBK);      // in flight under the MFMAs
        const float* at = As + j * SBKP + kq;
        const float* bt_ = Bs + kq * a.pitchB;
        if constexpr (SPL != 0) {
            // split-fp32: the eight k = 4*k4 + kq this lane reads across the chunk form ONE K = 32 bf16 fragment
            static_assert(SBK == 32, "one K = 32 step per chunk");
            f32x4 a0[ST], a1[ST], b0[CW], b1[CW];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
                for (int s = 0; s < ST; ++s) { a0[s][k4] = at[s * 16 * SBKP + k4 * 4]; a1[s][k4] = at[s * 16 * SBKP + (k4 + 4) * 4]; }
#pragma unroll
                for (int c = 0; c < CW; ++c) { b0[c][k4] = bt_[k4 * 4 * a.pitchB + bcol[c]]; b1[c][k4] = bt_[(k4 + 4) * 4 * a.pitchB + bcol[c]]; }
            }
            if constexpr (SPL == 3) {                   // hh, hm, mh, hl, lh, mm: the fp32 product to ~1.2e-7 relative
                bf16x8_t ah[ST], am[ST], al[ST];
#pragma unroll
                for (int s = 0; s < ST; ++s) split3_bf16x8(a0[s], a1[s], ah[s], am[s], al[s]);
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if (BAL && c == 2 && !third) continue;  // wave-uniform
                    bf16x8_t bh, bm, bl;
                    split3_bf16x8(b0[c], b1[c], bh, bm, bl);
#pragma unroll
                    for (int s = 0; s < ST; ++s) acc[s][c] = mfma_split3(ah[s], am[s], al[s], bh, bm, bl, acc[s][c]);
                }
            } else {
                bf16x8_t ah[ST], al[ST], bh[CW], bl[CW];
#pragma unroll
                for (int s = 0; s < ST; ++s) split_bf16x8(a0[s], a1[s], ah[s], al[s]);
#pragma unroll
                for (int c = 0; c < CW; ++c) split_bf16x8(b0[c], b1[c], bh[c], bl[c]);
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if (BAL && c == 2 && !third) continue;  // wave-uniform
#pragma unroll
                    for (int s = 0; s < ST; ++s) acc[s][c] = mfma_split(ah[s], al[s], bh[c], bl[c], acc[s][c]);
                }
            }
        } else {
#pragma unroll
            for (int k4 = 0; k4 < SBK / 4; ++k4) {
                float av[ST], bv[CW];
#pragma unroll
                for (int s = 0; s < ST; ++s) av[s] = at[s * 16 * SBKP + k4 * 4];
#pragma unroll
                for (int c = 0; c < CW; ++c) bv[c] = bt_[k4 * 4 * a.pitchB + bcol[c]];
#pragma unroll
                for (int c = 0; c < (BAL ? 2 : CW); ++c)
#pragma unroll
                    for (int s = 0; s < ST; ++s) acc[s][c] = mfma16(av[s], bv[c], acc[s][c]);
                if (BAL && third) {                     // wave-uniform branch around the third tile's MFMAs
#pragma unroll
                    for (int s = 0; s < ST; ++s) acc[s][2] = mfma16(av[s], bv[2], acc[s][2]);
                }
            }
        }
    }
    __syncthreads();                                   // stage dead; X3 may be overwritten
#pragma unroll
    for (int s = 0; s < ST; ++s) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            int col = (cw0 + c) * 16 + j;
            if (col >= ncols || (BAL && c == 2 && !third)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int cl = kq * 4 + r;
                X3[(s * 16 + cl) * G::PX3 + col] = acc[s][c][r] + b3r[s][r];
            }
        }
    }
    __syncthreads();
}

// out[tt][ub] (+)= sum_b M[a0+ub][b] * in[tt*V + b]
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


// dy chunk [CT][ncols] of frames [t0, t0+bt): loads (with the BatchNorm-backward prologue operands)
// go to registers first, commit() applies the prologue and stores to LDS.  NDY vectors per thread.
template <class G>
struct DyTile {
    static constexpr int VECW = G::VEC ? 4 : 1;
    static constexpr int ROWV = G::NCOLS / VECW;
    static constexpr int NDY = (CT * ROWV + G::NT - 1) / G::NT;
    float v1[NDY][VECW], v2[NDY][VECW], c1[NDY], c2[NDY], c0[NDY];

    __device__ __forceinline__ void load(const SrcDev& dy, int n, int c0ch, int T, int t0, int bt) {
        constexpr int V = G::V;
        const int ncols = bt * V;
        const long long cs = (long long)T * V;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            int e = threadIdx.x + i * G::NT;
            int row = e / ROWV, pos = (e - row * ROWV) * VECW;
            bool ok = row < CT && pos < ncols;
            int ch = dy.coff + c0ch + (ok ? row : 0);
            long long g = ((long long)n * dy.ctot + ch) * cs + (long long)t0 * V + (ok ? pos : 0);
            c1[i] = dy.coef ? dy.coef[ch] : 1.f;
            c2[i] = (dy.coef && dy.x2) ? dy.coef[dy.ctot + ch] : 0.f;
            c0[i] = dy.coef ? dy.coef[2 * dy.ctot + ch] : 0.f;
            if constexpr (G::VEC) {
                float4 a4 = ok ? *reinterpret_cast<const float4*>(dy.x1 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 b4 = (ok && dy.x2) ? *reinterpret_cast<const float4*>(dy.x2 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                v1[i][0] = a4.x; v1[i][1] = a4.y; v1[i][2] = a4.z; v1[i][3] = a4.w;
                v2[i][0] = b4.x; v2[i][1] = b4.y; v2[i][2] = b4.z; v2[i][3] = b4.w;
            } else {
                v1[i][0] = ok ? dy.x1[g] : 0.f;
                v2[i][0] = (ok && dy.x2) ? dy.x2[g] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void commit(const SrcDev& dy, float* Zs) {
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            int e = threadIdx.x + i * G::NT;
            int row = e / ROWV, pos = (e - row * ROWV) * VECW;
            if (row < CT) {
#pragma unroll
                for (int k = 0; k < VECW; ++k) {
                    float v = fmaf(c1[i], v1[i][k], fmaf(c2[i], v2[i][k], c0[i]));
                    if (dy.act == 1) v = fmaxf(v, 0.f);
                    Zs[row * G::NCOLS + pos + k] = v;
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <class G, int ST, int SPL>   // SPL: 0 exact fp32-input MFMA, 2 / 3 = two- / three-term bf16 split
__device__ __forceinline__ void ctrgc_fwd_body(const CtrgcArgs& a, float* y, float* stats_part, float* x3_out) {
    constexpr int V = G::V, TB = G::TB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords(a, n, c0)) return;
    float* Es = smem;                                  // [S][CT][VV]
    float* X3 = Es + ST * CT * G::VV;                  // regionB floats
    float* Zs = X3 + a.regionB;                        // [CT][NCOLS]
    const int tid = threadIdx.x;
    const int c = tid / (G::NTQ * 4), tq = (tid >> 2) % G::NTQ, uq = tid & 3;
    const int lrow = tid % (G::NTQ * 4);               // lane index inside the channel row (copy-out)

    TG_T(tt0);
    if (a.E) load_E<G, ST>(a.E, a.N, a.Cout, n, c0, Es, false);
    else build_E<G>(a, n, c0, Es, X3, a.regionB, false);
    TG_T(tt1); TG_ACC(0, tt1 - tt0);

    float st1 = 0.f, st2 = 0.f;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        TG_T(ta);
        x3_chunk<G, ST, SPL>(a, n, c0, t0, bt, X3);
        TG_T(tb); TG_ACC(1, tb - ta);
        float z[TB][G::UB];
#pragma unroll
        for (int tt = 0; tt < TB; ++tt)
#pragma unroll
            for (int ub = 0; ub < G::UB; ++ub) z[tt][ub] = 0.f;
        if (tq * TB < bt) {       // rows beyond bt hold stale data: results are discarded below
#pragma unroll
            for (int s = 0; s < ST; ++s)
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
        TG_T(tc); TG_ACC(2, tc - tb);
        __syncthreads();
        TG_T(td); TG_ACC(3, td - tc);
        float* yrow = y + (((long long)n * a.Cout + c0 + c) * a.T + t0) * V;
        for (int p = lrow; p < ncols; p += G::NTQ * 4) {
            float v = Zs[c * G::NCOLS + p];
            yrow[p] = v;
            st1 += v;
            st2 = fmaf(v, v, st2);
        }
        if (x3_out) {             // keep x3 for the backward (saves recomputing the GEMM there)
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                float* xo = x3_out + (((long long)n * ST * a.Cout + s * a.Cout + c0 + c) * a.T + t0) * V;
                const float* xi = X3 + (s * 16 + c) * G::PX3;
                if constexpr (G::VEC) {
                    for (int p4 = lrow; p4 < (ncols >> 2); p4 += G::NTQ * 4)
                        reinterpret_cast<float4*>(xo)[p4] = reinterpret_cast<const float4*>(xi)[p4];
                } else {
                    for (int p = lrow; p < ncols; p += G::NTQ * 4) xo[p] = xi[p];
                }
            }
        }
        // next chunk's first barrier (inside x3_chunk) protects Zs / X3 reuse
        TG_T(te); TG_ACC(4, te - td);
    }
    TG_T(tt2); TG_ACC(8, tt2 - tt0); TG_ACC(9, 1);
    if (stats_part) {
        // reduce over the NTQ*4 threads of the channel row (16 or 32 consecutive lanes)
#pragma unroll
        for (int o = 1; o < G::NTQ * 4; o <<= 1) { st1 += __shfl_xor(st1, o); st2 += __shfl_xor(st2, o); }
        if (lrow == 0) {
            stats_part[((long long)0 * a.Cout + c0 + c) * a.N + n] = st1;
            stats_part[((long long)1 * a.Cout + c0 + c) * a.N + n] = st2;
        }
    }
}

template <class G, int ST>
__global__ __launch_bounds__(G::NT) void ctrgc_fwd_kernel(const CtrgcArgs a, float* y, float* stats_part, float* x3_out) {
    ctrgc_fwd_body<G, ST, 0>(a, y, stats_part, x3_out);          // exact fp32-input MFMA
}
template <class G, int ST>
__global__ __launch_bounds__(G::NT) void ctrgc_fwd_split_kernel(const CtrgcArgs a, float* y, float* stats_part, float* x3_out) {
    ctrgc_fwd_body<G, ST, 2>(a, y, stats_part, x3_out);          // x3 GEMM as two-term split-fp32 on the bf16 matrix cores
}

template <class G, int ST>
__global__ __launch_bounds__(G::NT) void ctrgc_fwd_split3_kernel(const CtrgcArgs a, float* y, float* stats_part, float* x3_out) {
    ctrgc_fwd_body<G, ST, 3>(a, y, stats_part, x3_out);          // three-term split: fp32-exact to rounding (opt-in, TAMGCN_SPLIT3_FWD)
}

// ---------------------------------------------------------------------------
// backward 1: dx3
// ---------------------------------------------------------------------------
template <class G, int ST>
__global__ __launch_bounds__(G::NT) void ctrgc_bwd_dx3_kernel(const CtrgcArgs a, const SrcDev dy, float* dx3, float* db3_part) {
    constexpr int V = G::V, TB = G::TB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords(a, n, c0)) return;
    float* Es = smem;                                  // transposed tiles [S][CT][v][u]
    float* X3 = Es + ST * CT * G::VV;                  // output staging [S*16][PX3]
    float* Zs = X3 + a.regionB;                        // dy chunk [CT][NCOLS]
    const int tid = threadIdx.x;
    const int c = tid / (G::NTQ * 4), tq = (tid >> 2) % G::NTQ, vq = tid & 3;
    const int lrow = tid % (G::NTQ * 4);

    if (a.E) load_E<G, ST>(a.E, a.N, a.Cout, n, c0, Es, true);
    else build_E<G>(a, n, c0, Es, X3, a.regionB, true);

    float sb[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) sb[s] = 0.f;
    DyTile<G> dyt;
    dyt.load(dy, n, c0, a.T, 0, min(G::BT, a.T));
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        __syncthreads();
        dyt.commit(dy, Zs);
        __syncthreads();
        if (t0 + G::BT < a.T) dyt.load(dy, n, c0, a.T, t0 + G::BT, min(G::BT, a.T - t0 - G::BT));   // next chunk in flight
        if (tq * TB < bt) {
#pragma unroll
            for (int s = 0; s < ST; ++s) {
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
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            float* orow = dx3 + (((long long)n * ST * a.Cout + s * a.Cout + c0 + c) * a.T + t0) * V;
            for (int p = lrow; p < ncols; p += G::NTQ * 4) orow[p] = X3[(s * 16 + c) * G::PX3 + p];
        }
    }
    if (db3_part) {
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            float v = sb[s];
#pragma unroll
            for (int o = 1; o < G::NTQ * 4; o <<= 1) v += __shfl_xor(v, o);
            if (lrow == 0) db3_part[(long long)n * ST * a.Cout + s * a.Cout + c0 + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// backward 2: dE and everything behind it
// ---------------------------------------------------------------------------
template <class G, int ST>
__global__ __launch_bounds__(G::NT) void ctrgc_bwd_de_kernel(const CtrgcArgs a, const SrcDev dy, float* dA_part, float* dw4_part,
                                                             float* db4_part, float* dalpha_part, float* dpq) {
    constexpr int V = G::V, VV = G::VV, NT = G::NT;
    // dE ownership: thread -> (s, c, group of UBG joints); every owner walks all frames of a chunk
    constexpr int NUG0 = (NT / (ST * CT)) < V ? (NT / (ST * CT)) : V;
    constexpr int UBG = (V + NUG0 - 1) / NUG0;         // joints per owner (2 for V=20,S=3 on 512 threads)
    constexpr int NUG = (V + UBG - 1) / UBG;           // joint groups
    constexpr int NOWN = ST * CT * NUG;
    static_assert(NOWN <= NT, "not enough threads for the dE owners");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red_alpha[16];
    int n, c0;
    if (!block_coords(a, n, c0)) return;
    float* DE = smem;                                  // [S][CT][VV]
    float* X3 = DE + ST * CT * VV;                     // x3 tile / later D scratch
    float* Zs = X3 + a.regionB;                        // dy chunk [CT][NCOLS]
    const int tid = threadIdx.x;
    const int own_s = tid / (CT * NUG), own_c = (tid / NUG) % CT, own_g = tid % NUG;
    const bool owner = tid < NOWN;
    float dE[UBG][V];
#pragma unroll
    for (int i = 0; i < UBG; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) dE[i][v] = 0.f;

    DyTile<G> dyt;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        dyt.load(dy, n, c0, a.T, t0, bt);              // in flight under the x3 GEMM below
        x3_chunk<G, ST, 0>(a, n, c0, t0, bt, X3);  // begins with a barrier: previous chunk fully consumed
        dyt.commit(dy, Zs);
        __syncthreads();
        if (owner) {
            const float* xr = X3 + (own_s * 16 + own_c) * G::PX3;
            const float* dr = Zs + own_c * G::NCOLS;
            for (int tl = 0; tl < bt; ++tl) {
                float xv[V];
                if constexpr (V % 4 == 0) {
#pragma unroll
                    for (int v = 0; v < V; v += 4) {
                        f32x4 t = *reinterpret_cast<const f32x4*>(xr + tl * V + v);
                        xv[v] = t[0]; xv[v + 1] = t[1]; xv[v + 2] = t[2]; xv[v + 3] = t[3];
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) xv[v] = xr[tl * V + v];
                }
#pragma unroll
                for (int i = 0; i < UBG; ++i) {
                    int u = own_g * UBG + i;
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
        for (int i = 0; i < UBG; ++i) {
            int u = own_g * UBG + i;
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
        for (int e = tid; e < ST * VV; e += NT) {
            int s = e / VV, uv = e - s * VV;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc += DE[(s * CT + c) * VV + uv];
            dA_part[(long long)blk * ST * VV + e] = acc;
        }
    }
    // (b) chain through conv4 / tanh, subset by subset, rel-channel chunk by chunk.
    // Row work (per channel) is done by 16 threads per channel: tid < 256.
    const float alpha = a.alpha[0];
    const int RC = min(min(a.R, 16), a.regionB / VV) & ~3;
    const long long NV = (long long)a.N * V;
    float dalpha_acc = 0.f;
    const int c = tid >> 4, l16 = tid & 15;
    const bool rowthr = tid < 256;
    const int lane = tid & 63, wave = tid >> 6, mj = lane & 15, mkq = lane >> 4;
    constexpr int NTILE = (VV + 15) / 16;
    for (int s = 0; s < ST; ++s) {
        if (rowthr) {
            float db4raw = 0.f;
            for (int uv = l16; uv < VV; uv += 16) db4raw += DE[(s * CT + c) * VV + uv];
            db4raw = wave_sum16(db4raw);
            if (l16 == 0) {
                db4_part[((long long)n * ST + s) * a.Cout + c0 + c] = alpha * db4raw;
                dalpha_acc = fmaf(a.b4[s * a.Cout + c0 + c], db4raw, dalpha_acc);
            }
        }
        for (int r0 = 0; r0 < a.R; r0 += RC) {
            const int rc = min(RC, a.R - r0);
            __syncthreads();
            fill_D<G>(a, n, s, r0, rc, X3);
            __syncthreads();
            if (rowthr) {
                // dW4raw[c][r] = sum_uv dE[c][uv] * D[r][uv]
                float wacc[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
                for (int uv = l16; uv < VV; uv += 16) {
                    float de = DE[(s * CT + c) * VV + uv];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r < rc) wacc[r] = fmaf(de, X3[r * VV + uv], wacc[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float wsum = wave_sum16(wacc[r]);
                    if (l16 == 0 && r < rc) {
                        long long wi = ((long long)s * a.Cout + c0 + c) * a.R + r0 + r;
                        dw4_part[(long long)n * ST * a.Cout * a.R + wi] = alpha * wsum;
                        dalpha_acc = fmaf(a.w4[wi], wsum, dalpha_acc);
                    }
                }
            }
            __syncthreads();
            // dS[r][uv] = alpha * (sum_c W4[c][r] dE[c][uv]) * (1 - D^2), in place over D.
            // MFMA: rows = rel-channels (<= 16), K = the 16 channels of the tile, cols = (u,v).
            {
                float aw[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
                    aw[k4] = (mj < rc) ? a.w4[((long long)s * a.Cout + c0 + k4 * 4 + mkq) * a.R + r0 + mj] : 0.f;
                for (int ct = wave; ct < NTILE; ct += G::NW) {
                    const int col = ct * 16 + mj;
                    const int colc = col < VV ? col : 0;
                    f32x4 dd = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        dd = mfma16(aw[k4], DE[(s * CT + k4 * 4 + mkq) * VV + colc], dd);
                    if (col < VV) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int row = mkq * 4 + rr;
                            if (row < rc) {
                                float d = X3[row * VV + col];
                                X3[row * VV + col] = alpha * dd[rr] * (1.f - d * d);
                            }
                        }
                    }
                }
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
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < G::NW; ++w) t += red_alpha[w];
        dalpha_part[n * a.nct + c0 / CT] = t;
    }
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
using G20 = Geo<20, 2, 8>;       // 512 threads, 16 frames per chunk
using G25 = Geo<25, 1, 4>;       // 256 threads, 4 frames per chunk (E tiles take 120 KB)

struct CtrgcPlan { int pitchB, regionB; size_t lds; };

template <class G>
static bool plan_for(int S, CtrgcPlan* p) {
    int pitch = G::NCOLS;
    pitch += ((16 - (pitch & 31)) + 32) & 31;           // == 16 (mod 32): conflict-free B reads, 16-byte rows
    int stage = SBK * pitch + S * CT * SBKP;
    int x3 = S * 16 * G::PX3;
    int region = stage > x3 ? stage : x3;
    region = (region + 3) & ~3;
    if (region < 4 * G::VV) return false;               // E/D builders need >= 4 rel-channels of scratch (+ p/q: fill_args)
    size_t lds = sizeof(float) * ((size_t)S * CT * G::VV + region + (size_t)CT * G::NCOLS);
    p->pitchB = pitch; p->regionB = region; p->lds = lds;
    return lds <= 160 * 1024;
}

static int plan_ctrgc(int S, int V, CtrgcPlan* p) {
    if (S != 1 && S != 3) return -1;
    switch (V) {
        case 20: return plan_for<G20>(S, p) ? 0 : -1;
        case 25: return plan_for<G25>(S, p) ? 0 : -1;
        default: return -1;
    }
}

static int fill_args(const tamgcn_ctrgc_desc* d, const CtrgcPlan& p, CtrgcArgs* a, const char* who) {
    if (!(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->R > 0 && d->T > 0)) { tamgcn_set_error("%s: bad dims", who); return -1; }
    if (d->Cout % CT) { tamgcn_set_error("%s: Cout=%d must be a multiple of %d", who, d->Cout, CT); return -1; }
    if (d->R % 4) { tamgcn_set_error("%s: R=%d must be a multiple of 4", who, d->R); return -1; }
    if (p.regionB - ((2 * d->R * d->V + 3) & ~3) < 4 * d->V * d->V) {
        tamgcn_set_error("%s: R=%d too large for the E builder's LDS scratch (V=%d)", who, d->R, d->V); return -1;
    }
    if (!(d->x.x1 && d->pq && d->w3 && d->b3 && d->w4 && d->b4 && d->A && d->alpha)) { tamgcn_set_error("%s: null pointer", who); return -1; }
    if (d->x.x2 || d->x.coef || d->x.act) { tamgcn_set_error("%s: x must be a plain tensor (no fused prologue)", who); return -1; }
    if (d->x.coff + d->Cin > d->x.ctot) { tamgcn_set_error("%s: x channel slice out of range", who); return -1; }
    a->N = d->N; a->Cin = d->Cin; a->Cout = d->Cout; a->S = d->S; a->R = d->R; a->T = d->T;
    a->x = d->x.x1; a->x_ctot = d->x.ctot; a->x_coff = d->x.coff;
    a->pq = d->pq; a->w3 = d->w3; a->b3 = d->b3; a->w4 = d->w4; a->b4 = d->b4; a->A = d->A; a->alpha = d->alpha;
    a->E = d->E;
    a->nct = d->Cout / CT; a->pitchB = p.pitchB; a->regionB = p.regionB;
    return 0;
}

static unsigned grid_blocks(const CtrgcArgs& a) { return 8u * (unsigned)ceil_div(a.N, 8) * (unsigned)a.nct; }

template <typename K>
static void allow_lds(K kernel, size_t lds, tg_devmask* done) {   // once per instantiation and device
    tg_allow_lds((const void*)kernel, lds, done);
}

#define CTRGC_LAUNCH(KERNEL, GEO, ST_, FLAG, ...)                                                              \
    do {                                                                                                       \
        static tg_devmask FLAG = 0;                                                                            \
        allow_lds(KERNEL<GEO, ST_>, p.lds, &FLAG);   /* exact size: static LDS comes on top */                                                        \
        hipLaunchKernelGGL((KERNEL<GEO, ST_>), dim3(grid_blocks(a)), dim3(GEO::NT), p.lds, (hipStream_t)stream, __VA_ARGS__); \
        tamgcn_note_kernel(#KERNEL "<Geo<%d, %d, %d>, %d>", GEO::V, GEO::TB, GEO::NTQ, ST_);                           \
    } while (0)

#define CTRGC_DISPATCH(KERNEL, ...)                                                                            \
    do {                                                                                                       \
        if (d->V == 20 && d->S == 3) CTRGC_LAUNCH(KERNEL, G20, 3, f203, __VA_ARGS__);                          \
        else if (d->V == 20) CTRGC_LAUNCH(KERNEL, G20, 1, f201, __VA_ARGS__);                                  \
        else if (d->S == 3) CTRGC_LAUNCH(KERNEL, G25, 3, f253, __VA_ARGS__);                                   \
        else CTRGC_LAUNCH(KERNEL, G25, 1, f251, __VA_ARGS__);                                                  \
    } while (0)

}  // namespace

int tamgcn_ctrgc_tiled_lds_bytes(int S, int V, int R);      // ctrgc_tiled.hip: the large-skeleton family (V in {32, 64})

extern "C" int tamgcn_ctrgc_lds_bytes(int S, int V, int R) {
    CtrgcPlan p;
    if (plan_ctrgc(S, V, &p)) return tamgcn_ctrgc_tiled_lds_bytes(S, V, R);
    return (int)p.lds;
}

extern "C" int tamgcn_ctrgc_build_e(const tamgcn_ctrgc_desc* d, float* E, void* stream) {
    TG_CHECK(d && E && d->pq && d->w4 && d->b4 && d->A && d->alpha, "tamgcn_ctrgc_build_e: null pointer");
    TG_CHECK(d->N > 0 && d->S > 0 && d->Cout > 0 && d->Cout % 16 == 0, "tamgcn_ctrgc_build_e: bad shape N=%d S=%d Cout=%d", d->N, d->S, d->Cout);
    TG_CHECK(d->R >= 4 && d->R <= 32 && d->R % 4 == 0, "tamgcn_ctrgc_build_e: R=%d outside 4..32 (leave E null: tiles are then built on chip)", d->R);
    TG_CHECK(d->V == 20 || d->V == 25, "tamgcn_ctrgc_build_e: unsupported V=%d (V in {20,25})", d->V);
    EArgs a;
    a.N = d->N; a.Cout = d->Cout; a.S = d->S; a.R = d->R;
    a.pq = d->pq; a.w4 = d->w4; a.b4 = d->b4; a.A = d->A; a.alpha = d->alpha; a.E = E;
    const size_t lds = sizeof(float) * ((size_t)(16 + d->R) * d->V * d->V + 2 * (size_t)d->R * d->V);
    if (d->V == 20) {
        static tg_devmask f = 0;
        allow_lds(ctrgc_E_kernel<20>, 160 * 1024, &f);
        hipLaunchKernelGGL((ctrgc_E_kernel<20>), dim3(d->N * d->S), dim3(512), lds, (hipStream_t)stream, a);
    } else {
        static tg_devmask f = 0;
        allow_lds(ctrgc_E_kernel<25>, 160 * 1024, &f);
        hipLaunchKernelGGL((ctrgc_E_kernel<25>), dim3(d->N * d->S), dim3(512), lds, (hipStream_t)stream, a);
    }
    tamgcn_note_kernel("ctrgc_E_kernel<%d>", d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_build_e");
    return 0;
}

extern "C" int tamgcn_ctrgc_fwd(const tamgcn_ctrgc_desc* d, float* y, float* stats_part, float* x3_out, void* stream) {
    TG_CHECK(d && y, "tamgcn_ctrgc_fwd: null pointer");
    CtrgcPlan p;
    TG_CHECK(plan_ctrgc(d->S, d->V, &p) == 0, "tamgcn_ctrgc_fwd: unsupported S=%d V=%d (LDS-resident tiles exist for S in {1,3}, V in {20,25})", d->S, d->V);
    CtrgcArgs a;
    if (fill_args(d, p, &a, "tamgcn_ctrgc_fwd")) return -1;
    // Split-fp32 in an ACTIVATION-producing GEMM is opt-in (mode 2): its 4e-6 relative error is twenty times the fp32
    // rounding noise, flips correspondingly more ReLU masks, and end-to-end gradients then differ from the reference
    // by ~1 % in places (tests/test_gpu_model.py strict case) although every tensor of the forward stays within 5e-6.
    if (tamgcn_split_mode() >= 2) CTRGC_DISPATCH(ctrgc_fwd_split_kernel, a, y, stats_part, x3_out);
    else if (tamgcn_split3_fwd() && d->Cin % 32 == 0) CTRGC_DISPATCH(ctrgc_fwd_split3_kernel, a, y, stats_part, x3_out);
    else CTRGC_DISPATCH(ctrgc_fwd_kernel, a, y, stats_part, x3_out);
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
