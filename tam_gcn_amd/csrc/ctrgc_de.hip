// The dE chain of CTRGC's backward when the forward kept x3 (reference: autograd of
// models/ctrgcn.py:172-177 and :252-254), as two kernels instead of one LDS-bound monolith:
//
//   ctrgc_de_acc_kernel   dE[n,s,c,u,v] = sum_t dy(n,c,t,u) * x3[n,s*Cout+c,t,v]
//                         streaming, HBM-bound: reads 4 activations' worth, writes N*S*Cout*V*V
//   ctrgc_de_tail_kernel  one workgroup per (n, s) pushes dE through E = alpha*(W4 D + b4) + A,
//                         D = tanh(p_u - q_v): dA, db4, dW4 (MFMA, K = V*V), dD = W4^T dE (MFMA,
//                         K = Cout), dp / dq row / column sums.  Every (n, s) owns its slice of
//                         dpq, so there are no atomics and nothing to zero.
//
// Splitting costs one round trip of dE through HBM (<= 315 MB per layer at batch 256) and removes a
// per-channel-tile tail that was latency-bound with a single 159 KB workgroup per CU.
#include "common.h"

TG_TRACE_DEFINE(tamgcn_trace_read_de)

namespace {

// ---------------------------------------------------------------------------------------------
// kernel A: accumulate dE
// ---------------------------------------------------------------------------------------------
template <int V_, int ST_>
struct AccGeo {
    static constexpr int V = V_, ST = ST_, VV = V * V;
    static constexpr int CA = 8;                       // channels per workgroup
    static constexpr int NT = 256;
    static constexpr bool VEC = (V % 4 == 0);
    static constexpr int BT = V <= 20 ? 16 : 8;        // frames per chunk
    static constexpr int NCOLS = BT * V;
    // x3 rows: three consecutive rows meet in one ds_read_b128 lane group -> offset rows by one 16-B slot
    static constexpr int PX = NCOLS + 4;
    // dy rows: owners read 2 joints (8 B) each, 10 owners per row: pitch = 20 banks (mod 64) packs the
    // rows of a 32-lane group into 64 distinct banks
    static constexpr int PZ = NCOLS + 20;
    static constexpr int NUG0 = (NT / (ST * CA)) < V ? (NT / (ST * CA)) : V;
    static constexpr int UBG = (V + NUG0 - 1) / NUG0;  // joints per owner
    static constexpr int NUG = (V + UBG - 1) / UBG;
    static constexpr int NOWN = ST * CA * NUG;
    static constexpr int VECW = VEC ? 4 : 1;
    static constexpr int ROWV = NCOLS / VECW;
    static constexpr int NX = (ST * CA * ROWV + NT - 1) / NT;
    static constexpr int NY = (CA * ROWV + NT - 1) / NT;
    static constexpr size_t LDS = sizeof(float) * (size_t)(ST * CA * PX + CA * PZ);
    static_assert(NOWN <= NT, "not enough threads for the dE owners");
};

template <class G>
__global__ __launch_bounds__(G::NT) void ctrgc_de_acc_kernel(int N, int Cout, int T, const float* __restrict__ x3, const SrcDev dy,
                                                             float* __restrict__ dE) {
    constexpr int V = G::V, VV = G::VV, ST = G::ST, NT = G::NT, CA = G::CA, UBG = G::UBG, VECW = G::VECW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X3 = smem;                    // [ST*CA][PX]
    float* Zs = smem + ST * CA * G::PX;  // [CA][PZ]
    const int nca = Cout / CA;
    const int n = blockIdx.x / nca, c0 = (blockIdx.x - n * nca) * CA;
    const int tid = threadIdx.x;
    const int own_s = tid / (CA * G::NUG), own_c = (tid / G::NUG) % CA, own_g = tid % G::NUG;
    const bool owner = tid < G::NOWN;
    const long long cs = (long long)T * V;

    float xq[G::NX][VECW], y1[G::NY][VECW], y2[G::NY][VECW], k1[G::NY], k2[G::NY], k0[G::NY];
    auto load = [&](int t0, int bt) {
        const int ncols = bt * V;
#pragma unroll
        for (int i = 0; i < G::NX; ++i) {
            int e = tid + i * NT;
            int row = e / G::ROWV, pos = (e - row * G::ROWV) * VECW;
            bool ok = row < ST * CA && pos < ncols;
            int s = row / CA, c = row - s * CA;
            long long g = (((long long)n * ST + s) * Cout + c0 + c) * cs + (long long)t0 * V + pos;
            if constexpr (G::VEC) {
                float4 t = ok ? *reinterpret_cast<const float4*>(x3 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                xq[i][0] = t.x; xq[i][1] = t.y; xq[i][2] = t.z; xq[i][3] = t.w;
            } else {
                xq[i][0] = ok ? x3[g] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < G::NY; ++i) {
            int e = tid + i * NT;
            int row = e / G::ROWV, pos = (e - row * G::ROWV) * VECW;
            bool ok = row < CA && pos < ncols;
            int ch = dy.coff + c0 + (ok ? row : 0);
            long long g = ((long long)n * dy.ctot + ch) * cs + (long long)t0 * V + (ok ? pos : 0);
            k1[i] = dy.coef ? dy.coef[ch] : 1.f;
            k2[i] = (dy.coef && dy.x2) ? dy.coef[dy.ctot + ch] : 0.f;
            k0[i] = dy.coef ? dy.coef[2 * dy.ctot + ch] : 0.f;
            if constexpr (G::VEC) {
                float4 t = ok ? *reinterpret_cast<const float4*>(dy.x1 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 u = (ok && dy.x2) ? *reinterpret_cast<const float4*>(dy.x2 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
                y1[i][0] = t.x; y1[i][1] = t.y; y1[i][2] = t.z; y1[i][3] = t.w;
                y2[i][0] = u.x; y2[i][1] = u.y; y2[i][2] = u.z; y2[i][3] = u.w;
            } else {
                y1[i][0] = ok ? dy.x1[g] : 0.f;
                y2[i][0] = (ok && dy.x2) ? dy.x2[g] : 0.f;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < G::NX; ++i) {
            int e = tid + i * NT;
            int row = e / G::ROWV, pos = (e - row * G::ROWV) * VECW;
            if (row < ST * CA) {
                if constexpr (G::VEC) *reinterpret_cast<float4*>(X3 + row * G::PX + pos) = make_float4(xq[i][0], xq[i][1], xq[i][2], xq[i][3]);
                else X3[row * G::PX + pos] = xq[i][0];
            }
        }
#pragma unroll
        for (int i = 0; i < G::NY; ++i) {
            int e = tid + i * NT;
            int row = e / G::ROWV, pos = (e - row * G::ROWV) * VECW;
            if (row < CA) {
                float o[VECW];
#pragma unroll
                for (int k = 0; k < VECW; ++k) {
                    float v = fmaf(k1[i], y1[i][k], fmaf(k2[i], y2[i][k], k0[i]));
                    o[k] = dy.act == 1 ? fmaxf(v, 0.f) : v;
                }
                if constexpr (G::VEC) *reinterpret_cast<float4*>(Zs + row * G::PZ + pos) = make_float4(o[0], o[1], o[2], o[3]);
                else Zs[row * G::PZ + pos] = o[0];
            }
        }
    };

    float acc[UBG][V];
#pragma unroll
    for (int i = 0; i < UBG; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[i][v] = 0.f;

    // No register prefetch across the FMAs: three workgroups share a CU (42 KB of LDS, <= 168 VGPRs each)
    // and cover each other's load phases; holding the next chunk in registers would cost the third one.
    for (int t0 = 0; t0 < T; t0 += G::BT) {
        const int bt = min(G::BT, T - t0);
        load(t0, bt);
        __syncthreads();                               // owners are done with the previous chunk
        commit();
        __syncthreads();
        if (owner) {
            const float* xr = X3 + (own_s * CA + own_c) * G::PX;
            const float* dr = Zs + own_c * G::PZ + own_g * UBG;
            for (int tl = 0; tl < bt; ++tl) {
                float xv[V], d[UBG];
                if constexpr (G::VEC) {
#pragma unroll
                    for (int v = 0; v < V; v += 4) {
                        f32x4 t = *reinterpret_cast<const f32x4*>(xr + tl * V + v);
                        xv[v] = t[0]; xv[v + 1] = t[1]; xv[v + 2] = t[2]; xv[v + 3] = t[3];
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) xv[v] = xr[tl * V + v];
                }
                if constexpr (UBG == 2 && V % 2 == 0) {
                    float2 t = *reinterpret_cast<const float2*>(dr + tl * V);
                    d[0] = t.x; d[1] = t.y;
                } else {
#pragma unroll
                    for (int i = 0; i < UBG; ++i) d[i] = (own_g * UBG + i < V) ? dr[tl * V + i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < UBG; ++i)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[i][v] = fmaf(d[i], xv[v], acc[i][v]);
            }
        }
    }
    if (owner) {
        float* o = dE + (((long long)n * ST + own_s) * Cout + c0 + own_c) * VV;
#pragma unroll
        for (int i = 0; i < UBG; ++i) {
            const int u = own_g * UBG + i;
            if (u < V) {
                if constexpr (G::VEC) {
#pragma unroll
                    for (int v = 0; v < V; v += 4)
                        *reinterpret_cast<float4*>(o + u * V + v) = make_float4(acc[i][v], acc[i][v + 1], acc[i][v + 2], acc[i][v + 3]);
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) o[u * V + v] = acc[i][v];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel B: dE -> dA, db4, dW4, dalpha, dp, dq     (one workgroup per (n, s))
// ---------------------------------------------------------------------------------------------
struct TailArgs {
    int N, Cout, S, R, G;        // G channel groups: a workgroup handles Cout/G channels of one (n, subset)
    const float* dE; const float* pq; const float* w4; const float* b4; const float* alpha;
    float* dA_part; float* dw4_part; float* db4_part; float* dalpha_part; float* dpq;
};

constexpr int tail_pitch(int vv) {   // smallest pitch >= vv with pitch % 16 == 2: 16 rows x 2 k hit 32 banks
    int p = vv;
    while (p % 16 != 2) ++p;
    return p;
}

typedef __attribute__((address_space(1))) const void* de_gptr;
typedef __attribute__((address_space(3))) void* de_lptr;

template <int V, int RT, bool DBR>
__global__ __launch_bounds__(512) void ctrgc_de_tail_kernel(const TailArgs a) {
    // 512 threads: the kernel holds one workgroup per CU at R = 32 (D alone is 51 KB), so parallelism inside the
    // workgroup is what hides its latencies (256 threads = one wave per SIMD measured 400 us at C = 256).
    // Round 3: the dE chunk (16 channels x V*V floats = 25 whole 1 KB pieces when V = 20) reaches LDS by LDS-DMA into one of two
    // buffers while the previous chunk is worked on -- no register staging, no commit pass with its index arithmetic -- and
    // the per-wave partial dW4 tiles are double-buffered too (DBR), so a chunk costs ONE barrier instead of three; at R <= 8 the
    // kernel keeps one partial buffer and 73 KB instead, so that two workgroups share a CU.
    constexpr int VV = V * V, PD = tail_pitch(VV), NT = 512, NW = 8;
    constexpr int PE = VV;                       // dE chunk rows are contiguous (the DMA image of a contiguous run); 400 = 16 (mod 64):
    //                                              the four k rows of a dG fragment read sit 16 banks apart; the dW4 reads (13 per wave and
    //                                              chunk) take a 4-way conflict
    constexpr int CH = 16 * PE, CHP = (CH + 255) & ~255;           // chunk floats, padded to whole pieces (V = 25: 40 pieces, the last partial)
    constexpr int NPIECE = CHP / 256, PPW = (NPIECE + NW - 1) / NW;
    constexpr int NCT = (VV + 15) / 16, TPW = (NCT + NW - 1) / NW;
    constexpr int NA = (VV + NT - 1) / NT;
    constexpr int KST = (VV + 3) / 4;
    constexpr int NRED = DBR ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red_alpha[NW];
    float* DEs = smem;                           // [2][CHP]         dE chunks
    float* red = DEs + 2 * CHP;                  // [NRED][NW][16][RT*16]
    float* Ds = red + NRED * NW * 16 * RT * 16;  // [R][PD]          D, later dS in place
    float* PQ = Ds + a.R * PD;                   // [p | q][R][V]
    const int WQ = (16 * a.R + 255) & ~255;      // a chunk's W4 rows (16 channels x R, contiguous in global memory), whole pieces
    float* W4s = PQ + ((2 * a.R * V + 3) & ~3);  // [2][WQ + 256]: W4 rows, then one piece slot whose first 16 floats are b4
    const int grp = blockIdx.x % a.G, ns = blockIdx.x / a.G;
    const int n = ns / a.S, s = ns - n * a.S;
    const int cg = a.Cout / a.G, cbeg = grp * cg, cend = cbeg + cg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mj = lane & 15, mkq = lane >> 4;
    const long long NV = (long long)a.N * V;
    const float alpha = a.alpha[0];

    const float* dEg = a.dE + (((long long)n * a.S + s) * a.Cout) * VV;
    auto issue = [&](int c0, int buf) {          // wave-uniform pieces; lanes past the chunk re-read its last slot into the padding
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave * PPW + i;
            if (piece < NPIECE) {
                int f = piece * 256 + lane * 4;
                if (f > CH - 4) f = CH - 4;
                __builtin_amdgcn_global_load_lds((de_gptr)(dEg + (long long)c0 * VV + f), (de_lptr)(DEs + buf * CHP + piece * 256), 16, 0, 0);
            }
        }
    };
    TG_T(tt0);
    issue(cbeg, 0);
    {   // p, q of this (n, subset) -> LDS in one batch of loads, then D[r][uv] = tanh(p[r][u] - q[r][v]) from LDS
        const int cnt = 2 * a.R * V;
        constexpr int MAXL = 8;
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
        __syncthreads();
        const int total = a.R * VV;
        for (int e = tid; e < total; e += NT) {
            const int r = e / VV, uv = e - r * VV;
            const int u = uv / V, v = uv - u * V;
            Ds[r * PD + uv] = fast_tanh(PQ[r * V + u] - PQ[(a.R + r) * V + v]);
        }
    }
    f32x4 accG[TPW][RT];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) accG[j][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) accA[i] = 0.f;
    float dalpha_acc = 0.f;

    // Small operands of a chunk -- the W4 rows of its 16 channels (the W4^T fragment of the dG product, and w4[c][r] for dalpha)
    // and b4[c] -- travel by LDS-DMA beside the dE chunk, into the same double-buffered scheme.  vmcnt retires IN ORDER, so a
    // plain load consumed inside the loop body would wait for the whole DMA chunk issued in front of it (and hipcc sinks plain
    // loads below the DMA at will): that serialised the prefetch (tools/de_tail_phases.py: 70 % of a chunk's time).  A first fix
    // -- inline-asm loads one chunk ahead -- was WRONG: the compiler does not know such registers are pending and may copy them
    // (loop-carried values, allocation) before the wait; it showed as garbage dp / dq of single (n, subset) pairs once four models'
    // kernels ran side by side under one HIP graph.  Through LDS every consumer is an ordinary ds_read after the barrier.
    auto issue_small = [&](int c0, int buf) {
        if (wave == NW - 1) {                                       // the wave with no dE piece of its own (25 pieces over 8 waves)
            float* dst = W4s + buf * (WQ + 256);
            const float* g4 = a.w4 + ((long long)s * a.Cout + c0) * a.R;
            for (int piece = 0; piece < WQ / 256; ++piece) {
                int f = piece * 256 + lane * 4;
                if (f > 16 * a.R - 4) f = 16 * a.R - 4;
                __builtin_amdgcn_global_load_lds((de_gptr)(g4 + f), (de_lptr)(dst + piece * 256), 16, 0, 0);
            }
            const int fb = lane < 4 ? lane * 4 : 12;
            __builtin_amdgcn_global_load_lds((de_gptr)(a.b4 + s * a.Cout + c0 + fb), (de_lptr)(dst + WQ), 16, 0, 0);
        }
    };
    issue_small(cbeg, 0);
    constexpr int NFL = (16 * RT * 16 + NT - 1) / NT;               // flush items per thread (1)
    static_assert(NFL == 1, "one flush item per thread");
    const int dbc = (tid >> 4) & 15;                                // this thread's channel in the db4 pass
    float aw[RT][4];
    // the partial dW4 tiles of one chunk -> global (fixed order over the waves)
    const int fe_c = tid / (RT * 16), fe_r = tid - fe_c * (RT * 16);
    const bool fe_ok = tid < 16 * RT * 16 && fe_r < a.R;            // this thread's entry (channel, r) of a chunk's dW4 tile
    float w4own = 0.f, w4prev = 0.f;                                // w4[c][r] of that entry: this chunk's, the previous chunk's
    auto flush_red = [&](int c0, const float* rb, float w4v) {
        if (fe_ok) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) t += rb[(wv * 16 + fe_c) * (RT * 16) + fe_r];
            const long long wi = ((long long)s * a.Cout + c0 + fe_c) * a.R + fe_r;
            a.dw4_part[(long long)n * a.S * a.Cout * a.R + wi] = alpha * t;
            dalpha_acc = fmaf(w4v, t, dalpha_acc);
        }
    };

    int ci = 0;
    TG_T(tt1); TG_ACC(0, tt1 - tt0);
    for (int c0 = cbeg; c0 < cend; c0 += 16, ++ci) {
        const float* DE = DEs + (ci & 1) * CHP;
        float* rb = red + (NRED == 2 ? (ci & 1) : 0) * NW * 16 * RT * 16;
        TG_T(ta);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of chunk c0 (dE and small operands) have landed
        TG_T(tb); TG_ACC(1, tb - ta);
        __syncthreads();                                           // everyone's have; chunk c0 - 16 is consumed (and D is filled)
        TG_T(tc); TG_ACC(2, tc - tb);
        if (c0 + 16 < cend) { issue_small(c0 + 16, (ci + 1) & 1); issue(c0 + 16, (ci + 1) & 1); }
        const float* W4c = W4s + (ci & 1) * (WQ + 256);           // this chunk's W4 rows [16][R] and b4 [16]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) aw[rt][k4] = rt * 16 + mj < a.R ? W4c[(k4 * 4 + mkq) * a.R + rt * 16 + mj] : 0.f;
        const float b4c = W4c[WQ + dbc];
        // (the previous chunk's W4 rows sit in the buffer the DMA just issued is overwriting: its entry for the dalpha sum was taken
        // at that chunk's own iteration)
        w4prev = w4own;
        w4own = fe_ok ? W4c[fe_c * a.R + fe_r] : 0.f;
        if (NRED == 2 && ci > 0) flush_red(c0 - 16, red + ((ci - 1) & 1) * NW * 16 * RT * 16, w4prev);
        TG_T(td); TG_ACC(3, td - tc);
        // dG[r][uv] += sum_c W4[c][r] dE[c][uv]
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int ct = wave + j * NW;
            if (ct < NCT) {
                const int col = ct * 16 + mj;
                const int colc = col < VV ? col : VV - 1;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    float bv = DE[(k4 * 4 + mkq) * PE + colc];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accG[j][rt] = mfma16(aw[rt][k4], bv, accG[j][rt]);
                }
            }
        }
        TG_T(te); TG_ACC(4, te - td);
        // dA partial: sum over channels
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int uv = tid + i * NT;
            if (uv < VV) {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) t += DE[c * PE + uv];
                accA[i] += t;
            }
        }
        TG_T(tf); TG_ACC(5, tf - te);
        // db4raw[c] = sum_uv dE[c][uv]
        {
            const int c = (tid >> 4) & 15, l16 = tid & 15;          // threads 256.. shadow 0..255 (no second write)
            float t = 0.f;
            for (int uv = l16; uv < VV; uv += 16) t += DE[c * PE + uv];
            t = wave_sum16(t);
            if (l16 == 0 && tid < 256) {
                a.db4_part[((long long)n * a.S + s) * a.Cout + c0 + c] = alpha * t;
                dalpha_acc = fmaf(b4c, t, dalpha_acc);
            }
        }
        TG_T(tg); TG_ACC(6, tg - tf);
        // dW4raw[c][r] = sum_uv dE[c][uv] D[r][uv]: K = VV split over the waves
        {
            f32x4 accW[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) accW[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int st = wave; st < KST; st += NW) {
                const int k = st * 4 + mkq;
                const bool kok = k < VV;
                float av = kok ? DE[mj * PE + k] : 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = rt * 16 + mj;
                    float bv = (kok && r < a.R) ? Ds[r * PD + k] : 0.f;
                    accW[rt] = mfma16(av, bv, accW[rt]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) rb[(wave * 16 + mkq * 4 + rr) * (RT * 16) + rt * 16 + mj] = accW[rt][rr];
        }
        TG_T(th); TG_ACC(7, th - tg);
        if (NRED == 1) {
            __syncthreads();
            flush_red(c0, rb, w4own);
        }
        TG_T(ti); TG_ACC(10, ti - th);
    }
    if (NRED == 2) {
        __syncthreads();
        flush_red(cend - 16, red + ((ci - 1) & 1) * NW * 16 * RT * 16, w4own);
    }
    TG_T(tz0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int uv = tid + i * NT;
        if (uv < VV) a.dA_part[(((long long)n * a.G + grp) * a.S + s) * VV + uv] = accA[i];
    }
    __syncthreads();                         // every wave is done reading D
    // dS[r][uv] = alpha * dG * (1 - D^2), in place over D
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int ct = wave + j * NW;
        const int col = ct * 16 + mj;
        if (ct < NCT && col < VV) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = rt * 16 + mkq * 4 + rr;
                    if (r < a.R) {
                        float d = Ds[r * PD + col];
                        Ds[r * PD + col] = alpha * accG[j][rt][rr] * (1.f - d * d);
                    }
                }
        }
    }
    __syncthreads();
    // dp[r][u] = sum_v dS[r][u][v];  dq[r][v] = -sum_u dS[r][u][v]
    for (int e = tid; e < a.R * V * 2; e += NT) {
        const int which = e / (a.R * V);
        const int rem = e - which * a.R * V;
        const int r = rem / V, k = rem - r * V;
        float t = 0.f;
        if (which == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) t += Ds[r * PD + k * V + v];
        } else {
#pragma unroll
            for (int u = 0; u < V; ++u) t -= Ds[r * PD + u * V + k];
        }
        a.dpq[(((long long)grp * a.S * 2 + s * 2 + which) * a.R + r) * NV + (long long)n * V + k] = t;
    }
    dalpha_acc = wave_sum64(dalpha_acc);
    if (lane == 0) red_alpha[wave] = dalpha_acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < NW; ++w) t += red_alpha[w];
        a.dalpha_part[(n * a.S + s) * a.G + grp] = t;
    }
    TG_T(tz1); TG_ACC(11, tz1 - tz0); TG_ACC(8, tz1 - tt0); TG_ACC(9, 1);
}

// The register-staged form of the tail (one dE buffer, the next chunk prefetched into registers): V = 25, where two DMA buffers
// (2 x 41 KB) do not fit beside D (80 KB at R = 32).
template <int V, int RT>
__global__ __launch_bounds__(512) void ctrgc_de_tail_reg_kernel(const TailArgs a) {
    // 512 threads: the kernel holds one workgroup per CU at R = 32 (D alone is 51 KB), so parallelism inside the
    // workgroup is what hides its latencies (256 threads = one wave per SIMD measured 400 us at C = 256)
    constexpr int VV = V * V, PD = tail_pitch(VV), NT = 512, NW = 8;
    constexpr int NCT = (VV + 15) / 16, TPW = (NCT + NW - 1) / NW;
    constexpr int NA = (VV + NT - 1) / NT;
    constexpr int NCH = (16 * VV / 4 + NT - 1) / NT;
    constexpr int KST = (VV + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red_alpha[NW];
    float* Ds = smem;                        // [R][PD]    D, later dS in place
    float* DEs = Ds + a.R * PD;              // [16][PD]   dE chunk
    float* red = DEs + 16 * PD;              // [NW][16][RT*16]
    const int grp = blockIdx.x % a.G, ns = blockIdx.x / a.G;
    const int n = ns / a.S, s = ns - n * a.S;
    const int cg = a.Cout / a.G, cbeg = grp * cg, cend = cbeg + cg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mj = lane & 15, mkq = lane >> 4;
    const long long NV = (long long)a.N * V;
    const float alpha = a.alpha[0];

    float4 pre[NCH];
    auto load = [&](int c0) {
        const float4* g = reinterpret_cast<const float4*>(a.dE + (((long long)n * a.S + s) * a.Cout + c0) * VV);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int e = tid + i * NT;
            pre[i] = e < 16 * VV / 4 ? g[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load(cbeg);
    {   // p, q of this (n, subset) -> LDS in one batch of loads, then D[r][uv] = tanh(p[r][u] - q[r][v]) from LDS
        // (reading p, q per element from L2 was a dozen dependent round trips per thread)
        float* PQ = red + NW * 16 * RT * 16;                       // [p | q][R][V]
        const int cnt = 2 * a.R * V;
        constexpr int MAXL = 8;
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
        __syncthreads();
        const int total = a.R * VV;
        for (int e = tid; e < total; e += NT) {
            const int r = e / VV, uv = e - r * VV;
            const int u = uv / V, v = uv - u * V;
            Ds[r * PD + uv] = fast_tanh(PQ[r * V + u] - PQ[(a.R + r) * V + v]);
        }
    }
    f32x4 accG[TPW][RT];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) accG[j][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) accA[i] = 0.f;
    float dalpha_acc = 0.f;

    for (int c0 = cbeg; c0 < cend; c0 += 16) {
        // W4^T fragment of this chunk: A[i = r][k = c]
        float aw[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                aw[rt][k4] = (rt * 16 + mj < a.R) ? a.w4[((long long)s * a.Cout + c0 + k4 * 4 + mkq) * a.R + rt * 16 + mj] : 0.f;
        __syncthreads();                     // previous chunk (and the D fill) done
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int e4 = (tid + i * NT) * 4;
            if (e4 < 16 * VV) {
                float vals[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int e = e4 + k;
                    int c = e / VV, uv = e - c * VV;
                    DEs[c * PD + uv] = vals[k];
                }
            }
        }
        __syncthreads();
        if (c0 + 16 < cend) load(c0 + 16);
        // dG[r][uv] += sum_c W4[c][r] dE[c][uv]
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int ct = wave + j * NW;
            if (ct < NCT) {
                const int col = ct * 16 + mj;
                const int colc = col < VV ? col : VV - 1;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    float b = DEs[(k4 * 4 + mkq) * PD + colc];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accG[j][rt] = mfma16(aw[rt][k4], b, accG[j][rt]);
                }
            }
        }
        // dA partial: sum over channels
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int uv = tid + i * NT;
            if (uv < VV) {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) t += DEs[c * PD + uv];
                accA[i] += t;
            }
        }
        // db4raw[c] = sum_uv dE[c][uv]
        {
            const int c = (tid >> 4) & 15, l16 = tid & 15;          // threads 256.. shadow 0..255 (no second write)
            float t = 0.f;
            for (int uv = l16; uv < VV; uv += 16) t += DEs[c * PD + uv];
            t = wave_sum16(t);
            if (l16 == 0 && tid < 256) {
                a.db4_part[((long long)n * a.S + s) * a.Cout + c0 + c] = alpha * t;
                dalpha_acc = fmaf(a.b4[s * a.Cout + c0 + c], t, dalpha_acc);
            }
        }
        // dW4raw[c][r] = sum_uv dE[c][uv] D[r][uv]: K = VV split over the waves
        {
            f32x4 accW[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) accW[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int st = wave; st < KST; st += NW) {
                const int k = st * 4 + mkq;
                const bool kok = k < VV;
                float av = kok ? DEs[mj * PD + k] : 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = rt * 16 + mj;
                    float bv = (kok && r < a.R) ? Ds[r * PD + k] : 0.f;
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
            const int c = e / (RT * 16), r = e - c * (RT * 16);
            if (r < a.R) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += red[(w * 16 + c) * (RT * 16) + r];
                const long long wi = ((long long)s * a.Cout + c0 + c) * a.R + r;
                a.dw4_part[(long long)n * a.S * a.Cout * a.R + wi] = alpha * t;
                dalpha_acc = fmaf(a.w4[wi], t, dalpha_acc);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int uv = tid + i * NT;
        if (uv < VV) a.dA_part[(((long long)n * a.G + grp) * a.S + s) * VV + uv] = accA[i];
    }
    __syncthreads();                         // every wave is done reading D
    // dS[r][uv] = alpha * dG * (1 - D^2), in place over D
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int ct = wave + j * NW;
        const int col = ct * 16 + mj;
        if (ct < NCT && col < VV) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = rt * 16 + mkq * 4 + rr;
                    if (r < a.R) {
                        float d = Ds[r * PD + col];
                        Ds[r * PD + col] = alpha * accG[j][rt][rr] * (1.f - d * d);
                    }
                }
        }
    }
    __syncthreads();
    // dp[r][u] = sum_v dS[r][u][v];  dq[r][v] = -sum_u dS[r][u][v]
    for (int e = tid; e < a.R * V * 2; e += NT) {
        const int which = e / (a.R * V);
        const int rem = e - which * a.R * V;
        const int r = rem / V, k = rem - r * V;
        float t = 0.f;
        if (which == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) t += Ds[r * PD + k * V + v];
        } else {
#pragma unroll
            for (int u = 0; u < V; ++u) t -= Ds[r * PD + u * V + k];
        }
        a.dpq[(((long long)grp * a.S * 2 + s * 2 + which) * a.R + r) * NV + (long long)n * V + k] = t;
    }
    dalpha_acc = wave_sum64(dalpha_acc);
    if (lane == 0) red_alpha[wave] = dalpha_acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < NW; ++w) t += red_alpha[w];
        a.dalpha_part[(n * a.S + s) * a.G + grp] = t;
    }
}

template <typename K>
void allow_lds(K kernel, size_t lds, tg_devmask* done) {       // once per instantiation and device
    if (lds > 48 * 1024) tg_allow_lds(reinterpret_cast<const void*>(kernel), lds, done);
}

template <int V>
size_t tail_lds(int R, int RT, bool dbr) {
    const size_t chp = ((size_t)16 * V * V + 255) & ~(size_t)255;
    const size_t wq = ((size_t)16 * R + 255) & ~(size_t)255;
    return sizeof(float) * (2 * chp + (size_t)(dbr ? 2 : 1) * 8 * 16 * RT * 16 + (size_t)R * tail_pitch(V * V) + ((2 * (size_t)R * V + 3) & ~(size_t)3) + 2 * (wq + 256));
}

template <int V>
size_t tail_reg_lds(int R, int RT) { return sizeof(float) * ((size_t)(R + 16) * tail_pitch(V * V) + 8 * 16 * RT * 16 + 2 * (size_t)R * V); }

}  // namespace

#define DE_ACC_CASE(VV_, SS_)                                                                                          \
    if (d->V == VV_ && d->S == SS_) {                                                                                  \
        using G = AccGeo<VV_, SS_>;                                                                                    \
        static tg_devmask flag = 0;                                                                                    \
        allow_lds(ctrgc_de_acc_kernel<G>, G::LDS, &flag);                                                              \
        hipLaunchKernelGGL((ctrgc_de_acc_kernel<G>), dim3(d->N * (d->Cout / G::CA)), dim3(G::NT), G::LDS, (hipStream_t)stream, \
                           d->N, d->Cout, d->T, x3, make_src(*dy), dE);                                                \
        tamgcn_note_kernel("ctrgc_de_acc_kernel<AccGeo<%d, %d>>", VV_, SS_);                                           \
        launched = true;                                                                                               \
    }

extern "C" int tamgcn_ctrgc_bwd_de_acc(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, const float* x3, float* dE, void* stream) {
    TG_CHECK(d && dy && dy->x1 && x3 && dE, "tamgcn_ctrgc_bwd_de_acc: null pointer");
    TG_CHECK(d->N > 0 && d->T > 0 && d->Cout > 0 && d->Cout % 16 == 0, "tamgcn_ctrgc_bwd_de_acc: bad shape N=%d T=%d Cout=%d", d->N, d->T, d->Cout);
    TG_CHECK(dy->ctot >= dy->coff + d->Cout, "tamgcn_ctrgc_bwd_de_acc: dy has %d channels from %d, need %d", dy->ctot, dy->coff, d->Cout);
    bool launched = false;
    DE_ACC_CASE(20, 3) else DE_ACC_CASE(20, 1) else DE_ACC_CASE(25, 3) else DE_ACC_CASE(25, 1)
    TG_CHECK(launched, "tamgcn_ctrgc_bwd_de_acc: unsupported S=%d V=%d (S in {1,3}, V in {20,25})", d->S, d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_bwd_de_acc");
    return 0;
}

#define DE_TAIL_REG_CASE(VV_, RT_)                                                                                     \
    if (d->V == VV_ && rt == RT_) {                                                                                    \
        static tg_devmask flag = 0;                                                                                    \
        const size_t lds = tail_reg_lds<VV_>(d->R, RT_);                                                               \
        allow_lds(ctrgc_de_tail_reg_kernel<VV_, RT_>, tail_reg_lds<VV_>(32, 2), &flag);                                \
        hipLaunchKernelGGL((ctrgc_de_tail_reg_kernel<VV_, RT_>), dim3(d->N * d->S * groups), dim3(512), lds, (hipStream_t)stream, a); \
        tamgcn_note_kernel("ctrgc_de_tail_reg_kernel<%d, %d>", VV_, RT_);                                              \
        launched = true;                                                                                               \
    }
#define DE_TAIL_CASE(VV_, RT_, DBR_)                                                                                   \
    if (d->V == VV_ && rt == RT_ && dbr == DBR_) {                                                                     \
        static tg_devmask flag = 0;                                                                                    \
        const size_t lds = tail_lds<VV_>(d->R, RT_, DBR_);                                                             \
        allow_lds(ctrgc_de_tail_kernel<VV_, RT_, DBR_>, tail_lds<VV_>(RT_ == 2 ? 32 : 16, RT_, DBR_), &flag);          \
        hipLaunchKernelGGL((ctrgc_de_tail_kernel<VV_, RT_, DBR_>), dim3(d->N * d->S * groups), dim3(512), lds, (hipStream_t)stream, a); \
        tamgcn_note_kernel("ctrgc_de_tail_kernel<%d, %d, %d>", VV_, RT_, (int)DBR_);                                   \
        launched = true;                                                                                               \
    }

extern "C" int tamgcn_ctrgc_bwd_de_tail(const tamgcn_ctrgc_desc* d, const float* dE, float* dA_part, float* dw4_part, float* db4_part,
                                        float* dalpha_part, float* dpq, int groups, void* stream) {
    TG_CHECK(d && dE && dA_part && dw4_part && db4_part && dalpha_part && dpq, "tamgcn_ctrgc_bwd_de_tail: null pointer");
    TG_CHECK(d->pq && d->w4 && d->b4 && d->alpha, "tamgcn_ctrgc_bwd_de_tail: null parameter pointer");
    TG_CHECK(d->N > 0 && d->S > 0 && d->Cout > 0 && d->Cout % 16 == 0, "tamgcn_ctrgc_bwd_de_tail: bad shape N=%d S=%d Cout=%d", d->N, d->S, d->Cout);
    TG_CHECK(d->R >= 1 && d->R <= 32, "tamgcn_ctrgc_bwd_de_tail: R=%d outside 1..32 (the built range: see CTRGC.__init__ / INTEGRATION.md)", d->R);
    TailArgs a;
    TG_CHECK(groups >= 1 && d->Cout % (16 * groups) == 0, "tamgcn_ctrgc_bwd_de_tail: groups=%d must divide Cout/16=%d", groups, d->Cout / 16);
    a.N = d->N; a.Cout = d->Cout; a.S = d->S; a.R = d->R; a.G = groups;
    a.dE = dE; a.pq = d->pq; a.w4 = d->w4; a.b4 = d->b4; a.alpha = d->alpha;
    a.dA_part = dA_part; a.dw4_part = dw4_part; a.db4_part = db4_part; a.dalpha_part = dalpha_part; a.dpq = dpq;
    const int rt = d->R <= 16 ? 1 : 2;
    bool launched = false;
    // V = 20: the LDS-DMA form at R <= 8 (75 KB, one partial buffer: two workgroups per CU) and at R > 16 (145 KB, everything
    // double-buffered, one per CU either way); 8 < R <= 16 stays on the register-staged form, whose 62 KB put two workgroups
    // on a CU where the DMA form's 95 KB put one (measured: 238 vs 254 us per dE chain at 128 channels).  DMA pieces are 16 bytes
    // per lane: operands that are not 16-byte aligned take the register-staged form too.
    const bool al16 = (((uintptr_t)dE | (uintptr_t)d->w4 | (uintptr_t)d->b4) & 15) == 0;
    const bool dbr = d->R > 16, reg = d->V != 20 || !al16 || (d->R > 8 && d->R <= 16);
    if (reg) {
        DE_TAIL_REG_CASE(20, 1) else DE_TAIL_REG_CASE(20, 2) else DE_TAIL_REG_CASE(25, 1) else DE_TAIL_REG_CASE(25, 2)
    } else {
        DE_TAIL_CASE(20, 1, false) else DE_TAIL_CASE(20, 2, true)
    }
    TG_CHECK(launched, "tamgcn_ctrgc_bwd_de_tail: unsupported V=%d (V in {20,25})", d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_bwd_de_tail");
    return 0;
}
