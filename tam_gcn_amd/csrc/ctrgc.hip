// Fused CTRGC kernels for V = 20 (reference models/ctrgcn.py:172-177 and the 3-subset sum
// of unit_gcn.forward, :252-254).
//
// The channel-wise topology  E_s[n,c,u,v] = alpha*(W4_s[c,:].tanh(p_s[:,u]-q_s[:,v]) + b4_s[c]) + A_s[u,v]
// is built ONCE per layer by ctrgc_E_kernel (one workgroup per (sample, subset): D = tanh(p_u - q_v) in LDS,
// W4.D on v_mfma_f32_16x16x4_f32 for every 16-channel tile).  The forward and the dx3 backward load their
// tiles of it.
//
// ctrgc_fwd_kernel: one workgroup owns one sample n and a tile of CT output channels, for all S subsets and
// all T frames:
//   1. the E tiles of its channels (S*CT*V*V floats) are loaded into LDS and stay there;
//   2. per chunk of BT = 16 frames, x3 = W3 x + b3 for the S*CT rows is an MFMA GEMM over an LDS-staged x tile:
//      16-byte global loads (coalesced along t*V+v) are prefetched into registers one K chunk ahead -- across the
//      chunk boundary too: the next chunk's first operands are requested before the copy-out stores of this one;
//   3. the V-aggregation  z[c,t,u] = sum_s sum_v E_s[c,u,v]*x3_s[c,t,v]  is a per-channel (16 frames) x (V joints) x
//      (K = S*V) product on the matrix cores: both operands are rows of K contiguous floats in LDS (the x3 tile is
//      written as [c][frame][s*V+v], E as [c][u][s*V+v]) read as 16-byte vectors; rounds 1-2 ran it on the VALU with
//      E rows as 16-byte LDS reads: 3.4 of the 10 k LDS clocks per chunk, and as many VALU cycles as the GEMM takes
//      MFMA cycles (the fp32-input MFMA runs at the vector rate);
//   4. z is staged through LDS and written as whole contiguous rows, the x3 tile is written out for the
//      backward, and the train-mode BatchNorm moments of z are accumulated on the way (per-sample partials).
// Geometries.  Forward: CT = 16 channels, 512 threads, E 77 KB + x3 tile 61 KB + z tile 20 KB = 155 KB (one workgroup
// per CU): the x3 GEMM has S*CT = 48 rows = three full 16-row MFMA tiles.  An 8-channel form (78 KB, 256 threads, two
// workgroups per CU, 24 rows in two padded tiles) is kept for Cout % 16 == 8; it measured 8-17 % slower at every layer
// shape: two workgroups of four waves put the same two waves on a SIMD as one workgroup of eight, and the phases that
// bound the kernel -- waiting for the staged operands and for the copy-out stores to drain (vmcnt retires in order) --
// do not overlap better for it.  dx3 (no GEMM, no staging): CT = 8, two workgroups per CU.
//
// ctrgc_bwd_dx3_kernel:  dx3_s[c,t,v] = sum_u E_s[c,u,v] dy[c,t,u], on the matrix cores from the same E layout.
//
// All MFMA blocks are branch-free with compile-time tile counts (padding tiles are computed
// and discarded): per-MFMA guards made hipcc serialise every ds_read/MFMA pair.
#include "common.h"
#include <stdlib.h>

TG_TRACE_DEFINE(tamgcn_trace_read_ctrgc)

#ifndef TG_CKO
#define TG_CKO 0      // knock-out side builds of ctrgc_fwd_kernel (tools/ctrgc_knockout.py; results wrong by design): 1 no GEMM MFMAs,
#endif                // 2 no operand loads, 4 no stage commit, 8 no x3 tile write, 16 no aggregation, 32 no copy-out stores, 64 no GEMM fragment reads

namespace {

struct CtrgcArgs {
    int N, Cin, Cout, S, T;
    const float* x; int x_ctot, x_coff;
    const float* w3; const float* b3;
    const float* E;             // (N, S, Cout, V*V) from tamgcn_ctrgc_build_e
    int nct;                    // Cout / CT
};

// V joints; CT channels per workgroup; TB frames per thread in the aggregation; NTQ frame groups
// => NT = CT*NTQ*4 threads; SBK = K chunk of the x3 GEMM
template <int V_, int CT_, int TB_, int NTQ_, int SBK_>
struct Geo {
    static constexpr int V = V_, CT = CT_, TB = TB_, NTQ = NTQ_, SBK = SBK_;
    static constexpr int SBKP = SBK + 2;            // pitch/2 odd => the 16x4 A-fragment column reads hit 32 distinct banks
    static constexpr int NT = CT * NTQ * 4;         // threads
    static constexpr int NW = NT / 64;              // waves
    static constexpr int BT = NTQ * TB;             // frames per chunk
    static constexpr int NCOLS = BT * V;            // <= 320
    static constexpr int NCT = (NCOLS + 15) / 16;   // 16-wide column tiles of the x3 GEMM
    static constexpr int CW = (NCT + NW - 1) / NW;  // column tiles per wave
    static constexpr int VV = V * V;
    static constexpr int UB = (V + 3) / 4;          // joints per thread in the aggregation
    static constexpr int PX3 = NCOLS;               // pitch of the x3 tile
    static constexpr int PITCHB = NCOLS + (((16 - (NCOLS & 31)) + 32) & 31);   // == 16 (mod 32): conflict-free B reads, 16-byte rows
    static constexpr int NPF = (SBK * (NCOLS / 4) + NT - 1) / NT;              // float4 of the x chunk per thread
    static_assert(V % 4 == 0, "rows of V joints are moved as 16-byte vectors");
    static_assert(NT % 64 == 0 && (NTQ * 4) <= 64 && 64 % (NTQ * 4) == 0, "a channel row's lanes sit in one wave");
};

template <class G, int ST>
struct Plan {
    static constexpr int NR = ST * G::CT;           // rows of the x3 GEMM
    static constexpr int NRT = (NR + 15) / 16;      // 16-row MFMA tiles (rows >= NR are zero weights)
    static constexpr int STAGE = G::SBK * G::PITCHB + NRT * 16 * G::SBKP;
    static constexpr int X3T = NR * G::PX3;
    static constexpr int REGION = ((STAGE > X3T ? STAGE : X3T) + 3) & ~3;      // the stage aliases the x3 tile
    static constexpr int ETILE = (NR * G::VV + 255) & ~255;                    // E tiles, whole 1 KB DMA pieces
    static constexpr size_t LDS = sizeof(float) * ((size_t)ETILE + REGION + (size_t)G::CT * G::NCOLS);
};

// blockIdx -> (n, channel tile); blocks that share n are b, b+8, ... => same XCD / L2
template <class G>
__device__ __forceinline__ bool block_coords(const CtrgcArgs& a, int& n, int& c0) {
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    n = (q / a.nct) * 8 + xcd;
    c0 = (q % a.nct) * G::CT;
    return n < a.N;
}

// E tiles of channels c0..c0+CT-1 from the tensor tamgcn_ctrgc_build_e wrote, (N, S, Cout, V*V): per subset the CT rows
// are one contiguous run, fetched with 16-byte loads in a single batch, into the layout of the MFMA aggregation:
// Ek[c][u][s*V + v] (one row of KP = S*V floats per (channel, joint u)):
// row u of a channel is the B operand's k axis, contiguous for 16-byte fragment reads; rows are 240 B apart at S = 3, V = 20,
// i.e. 15 sixteen-byte slots: the sixteen rows a fragment read touches sit in sixteen different slots of the bank row.
// The tiles travel by LDS-DMA (global_load_lds, 16 bytes per lane, 1 KB per wave instruction): the LDS image of a piece
// is linear, so the layout change sits in the per-lane SOURCE address; no registers, no LDS store instructions, and the
// transfer overlaps whatever the workgroup requests next (the first operand chunk).  It is complete at the first
// __syncthreads() after the call (hipcc drains outstanding LDS-DMA there) -- the caller's first barrier.
typedef __attribute__((address_space(1))) const void* cg_gptr;
typedef __attribute__((address_space(3))) void* cg_lptr;

template <class G, int ST>
__device__ __forceinline__ void load_E_k(const float* __restrict__ Eg, int Cout, int n, int c0, float* Ek) {
    constexpr int V = G::V, VV = G::VV, CT = G::CT, KP = ST * V;
    constexpr int TOT4 = CT * V * KP / 4;              // float4 of the image
    constexpr int NPIECE = (TOT4 + 63) / 64, NPW = (NPIECE + G::NW - 1) / G::NW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int piece = wave * NPW + i;                          // wave-uniform
        if (piece < NPIECE) {
            int q = piece * 64 + lane;                             // float4 index inside the image [c][u][s*V + v]
            if (q >= TOT4) q = TOT4 - 1;                           // tail lanes of the last piece land in the region's padding
            const int f = 4 * q;
            const int cu = f / KP, k = f - cu * KP;                // cu = c*V + u
            const int sidx = k / V, v = k - sidx * V;
            const int c = cu / V, u = cu - c * V;
            const float* gp = Eg + (((long long)n * ST + sidx) * Cout + c0 + c) * VV + u * V + v;
            __builtin_amdgcn_global_load_lds((cg_gptr)gp, (cg_lptr)(Ek + piece * 256), 16, 0, 0);
        }
    }
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
    float* Ds = smem;                         // [R][VV]
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
    // Tiles leave straight from the accumulators: a lane holds four channels of one (u, v) column, so a store instruction
    // writes four 64-byte row segments (round 3: the staging tile with its two barriers per 16 channels kept the waves in
    // step for a kernel that only has to stream 173 MB out)
    float awn[8], b4n[4];
    for (int c0 = 0; c0 < a.Cout; c0 += 16) {
        if (c0 + 16 < a.Cout) {                        // next tile's fragment in flight under this tile's MFMAs and stores
#pragma unroll
            for (int k = 0; k < 8; ++k)
                awn[k] = (k * 4 + kq < a.R) ? a.w4[((long long)s * a.Cout + c0 + 16 + j) * a.R + k * 4 + kq] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) b4n[r] = a.b4[s * a.Cout + c0 + 16 + kq * 4 + r];
        }
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
                    for (int r = 0; r < 4; ++r) Eg[(long long)(c0 + kq * 4 + r) * VV + col] = alpha * (acc[r] + b4r[r]) + Ar[it];
                }
            }
        }
        if (c0 + 16 < a.Cout) {
#pragma unroll
            for (int k = 0; k < 8; ++k) aw[k] = awn[k];
#pragma unroll
            for (int r = 0; r < 4; ++r) b4r[r] = b4n[r];
        }
    }
}

// ---------------------------------------------------------------------------
// One K chunk (SBK input channels) of the x3 GEMM's operands on its way from global memory to the LDS stage: SBK rows of
// the x tile and the matching weight columns, as registers.  load() issues the reads; pin() makes the compiler wait for
// them at THAT point (vmcnt retires in order and counts stores: a wait placed after the copy-out stores of a chunk would
// also wait for those stores to drain to HBM -- the loads of the next chunk are therefore issued before the stores and
// pinned before the first store leaves).
// ---------------------------------------------------------------------------
template <class G, int ST>
struct X3Pref {
    using P = Plan<G, ST>;
    static constexpr int NPF = G::NPF, NT = G::NT, SBK = G::SBK, CT = G::CT;
    static constexpr int ROWV = G::NCOLS / 4;                     // vectors per row (full chunk geometry)
    static constexpr int NAF = (P::NRT * 16 * SBK + NT - 1) / NT; // weight-tile values per thread
    float4 rv[NPF];
    float wv[NAF];

    __device__ __forceinline__ void load(const CtrgcArgs& a, int n, int c0, int t0, int bt, int k0) {
        if (TG_CKO & 2) return;
        constexpr int V = G::V;
        const int tid = threadIdx.x, ncols = bt * V;
        const long long cs = (long long)a.T * V;
        const long long xb = ((long long)n * a.x_ctot + a.x_coff) * cs + (long long)t0 * V;
#pragma unroll
        for (int i = 0; i < NAF; ++i) {
            const int e = tid + i * NT;
            const int kk = e % SBK, row = e / SBK;
            const int sidx = row / CT, c = row - sidx * CT, k = k0 + kk;
            wv[i] = (row < P::NR && k < a.Cin) ? a.w3[((long long)sidx * a.Cout + c0 + c) * a.Cin + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + i * NT;
            const int kk = e / ROWV, pos = (e - kk * ROWV) * 4;
            const int k = k0 + kk;
            const bool ok = kk < SBK && k < a.Cin && pos < ncols;
            rv[i] = ok ? *reinterpret_cast<const float4*>(a.x + xb + (long long)k * cs + pos) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) asm volatile("" : "+v"(rv[i].x), "+v"(rv[i].y), "+v"(rv[i].z), "+v"(rv[i].w));
#pragma unroll
        for (int i = 0; i < NAF; ++i) asm volatile("" : "+v"(wv[i]));
    }
    __device__ __forceinline__ void commit(float* As, float* Bs) const {    // registers -> LDS stage
        if (TG_CKO & 4) return;
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NAF; ++i) {
            const int e = tid + i * NT;
            const int kk = e % SBK, row = e / SBK;
            if (row < P::NRT * 16) As[row * G::SBKP + kk] = wv[i];
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + i * NT;
            const int kk = e / ROWV, pos = (e - kk * ROWV) * 4;
            if (kk < SBK) *reinterpret_cast<float4*>(Bs + kk * G::PITCHB + pos) = rv[i];
        }
    }
};

// ---------------------------------------------------------------------------
// x3 tile for frames [t0, t0+bt): X3[(s*CT+c)*PX3 + tl*V + v] = (W3_s x)[c0+c] + b3
// The staging buffers alias the X3 tile (they are dead before the tile is written).  pf holds the first K chunk of THIS
// frame chunk (loaded by the caller, or by the previous call); when a next frame chunk exists its first K chunk is
// requested right after the K loop, so that it travels under the tile write, the aggregation and the copy-out.
// ---------------------------------------------------------------------------
template <class G, int ST>
__device__ void x3_chunk(const CtrgcArgs& a, int n, int c0, int t0, int bt, float* X3, X3Pref<G, ST>& pf, int next_t0, int next_bt) {
    // the tile leaves as X3[(c*BT + frame)*KP + s*V + v], KP = S*V: a (channel, frame) row holds the aggregation's k axis
    using P = Plan<G, ST>;
    constexpr int V = G::V, CW = G::CW, SBK = G::SBK, SBKP = G::SBKP, CT = G::CT;
    constexpr int NR = P::NR, NRT = P::NRT, PB = G::PITCHB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int ncols = bt * V;
    // Column tiles per wave.  Waves w and w + 4 share a SIMD (its MFMA pipe): with 20 tiles on 8 waves, "3 per wave in
    // order" would load the four SIMDs 6/6/5/3; every SIMD gets 5 instead -- wave w < 4 takes three, wave w + 4 two.
    constexpr bool BAL = (G::NW == 8 && G::NCT == 20);
    const int cw0 = BAL ? 5 * (wave & 3) + (wave < 4 ? 0 : 3) : wave * CW;
    const bool third = !BAL || wave < 4;               // wave-uniform: does tile c = 2 exist for this wave
    float* Bs = X3;                                   // [SBK][PB]
    float* As = X3 + SBK * PB;                        // [NRT*16][SBKP]

    f32x4 acc[NRT][CW];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int bcol[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { int col = (cw0 + c) * 16 + j; bcol[c] = (col < ncols && (c < 2 || third)) ? col : 0; }

    float b3r[NRT][4];                                 // fetched here: in flight under the K loop, not exposed after it
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kq * 4 + r;
            const int sidx = row / CT, c = row - sidx * CT;
            b3r[rt][r] = row < NR ? a.b3[sidx * a.Cout + c0 + c] : 0.f;
        }
    for (int k0 = 0; k0 < a.Cin; k0 += SBK) {
        __syncthreads();                               // previous users of the region are done
        pf.commit(As, Bs);
        __syncthreads();
        if (k0 + SBK < a.Cin) pf.load(a, n, c0, t0, bt, k0 + SBK);      // in flight under the MFMAs
        else if (next_bt > 0) pf.load(a, n, c0, next_t0, next_bt, 0);   // the next frame chunk's first K chunk
        const float* at = As + j * SBKP + kq;
        const float* bt_ = Bs + kq * PB;
#pragma unroll
        for (int k4 = 0; k4 < SBK / 4; ++k4) {
            float av[NRT], bv[CW];
            if (TG_CKO & 64) {
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) asm volatile("" : "=v"(av[rt]));
#pragma unroll
                for (int c = 0; c < CW; ++c) asm volatile("" : "=v"(bv[c]));
            } else {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) av[rt] = at[rt * 16 * SBKP + k4 * 4];
#pragma unroll
            for (int c = 0; c < CW; ++c) bv[c] = bt_[k4 * 4 * PB + bcol[c]];
            }
            if ((TG_CKO & 1) && k4) continue;
#pragma unroll
            for (int c = 0; c < (BAL ? 2 : CW); ++c)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) acc[rt][c] = mfma16(av[rt], bv[c], acc[rt][c]);
            if (BAL && third) {                         // wave-uniform branch around the third tile's MFMAs
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) acc[rt][2] = mfma16(av[rt], bv[2], acc[rt][2]);
            }
        }
    }
    __syncthreads();                                   // stage dead; X3 may be overwritten
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            int col = (cw0 + c) * 16 + j;
            if ((TG_CKO & 8) && (rt || c)) continue;
            if (col >= ncols || (BAL && c == 2 && !third)) continue;
            const int fr = col / V, v = col - fr * V;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + kq * 4 + r;
                const int sidx = row / CT, ch = row - sidx * CT;
                if (row < NR) X3[(ch * G::BT + fr) * (ST * V) + sidx * V + v] = acc[rt][c][r] + b3r[rt][r];
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// V-aggregation on the matrix cores:  z[c][t][u] = sum_k x3[c][t][k] * E[c][u][k],  k = (s, v), K = S*V (60, padded to 64).
// Per channel a (16 frames) x (V joints, two 16-column tiles) x K product; lane (i, kq) reads its operands as 16-byte
// vectors: k = 16*m + 4*kq + r is element r of vector m -- any assignment of k to (step, lane) is valid as long as both
// operands use the same one.  The four k beyond K (m = 3, kq = 3 at K = 60) are zeroed in BOTH operands (what lies
// behind a row is the next row, or stale bytes behind the tile).  Wave w owns channels w, w + NW, ...; the two column
// tiles of a channel share its A fragments and alternate, so consecutive MFMAs never wait on each other's accumulator.
// ---------------------------------------------------------------------------
template <class G, int ST>
__device__ __forceinline__ void aggregate_mfma(const float* Ek, const float* X3, float* Zs, int bt) {
    constexpr int V = G::V, CT = G::CT, KP = ST * V, NM = (KP + 15) / 16, NUT = (V + 15) / 16;
    static_assert(G::BT == 16, "one 16-frame MFMA row tile per chunk");
    static_assert(KP % 4 == 0, "rows are read as 16-byte vectors");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    for (int c = wave; c < CT; c += G::NW) {
        f32x4 av[NM], bv[NUT][NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const bool in = 16 * m + 4 * kq < KP;                 // whole vector inside the row (KP % 4 == 0)
            const int ko = in ? 16 * m + 4 * kq : 0;
            av[m] = *reinterpret_cast<const f32x4*>(X3 + (c * G::BT + j) * KP + ko);
#pragma unroll
            for (int ut = 0; ut < NUT; ++ut) {
                const int u = ut * 16 + j < V ? ut * 16 + j : V - 1;   // padding columns repeat the last joint: discarded below
                bv[ut][m] = *reinterpret_cast<const f32x4*>(Ek + (c * V + u) * KP + ko);
            }
            if (!in) {
                av[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ut = 0; ut < NUT; ++ut) bv[ut][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 acc[NUT];
#pragma unroll
        for (int ut = 0; ut < NUT; ++ut) acc[ut] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ut = 0; ut < NUT; ++ut) acc[ut] = mfma16(av[m][r], bv[ut][m][r], acc[ut]);
#pragma unroll
        for (int ut = 0; ut < NUT; ++ut) {
            const int u = ut * 16 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = kq * 4 + r;
                if (u < V && fr < bt) Zs[c * G::NCOLS + fr * V + u] = acc[ut][r];
            }
        }
    }
}

// dy chunk [CT][ncols] of frames [t0, t0+bt): loads (with the BatchNorm-backward prologue operands)
// go to registers first, commit() applies the prologue and stores to LDS.  NDY vectors per thread.
template <class G>
struct DyTile {
    static constexpr int ROWV = G::NCOLS / 4;
    static constexpr int NDY = (G::CT * ROWV + G::NT - 1) / G::NT;
    float v1[NDY][4], v2[NDY][4], c1[NDY], c2[NDY], c0[NDY];

    __device__ __forceinline__ void load(const SrcDev& dy, int n, int c0ch, int T, int t0, int bt) {
        constexpr int V = G::V;
        const int ncols = bt * V;
        const long long cs = (long long)T * V;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            int e = threadIdx.x + i * G::NT;
            int row = e / ROWV, pos = (e - row * ROWV) * 4;
            bool ok = row < G::CT && pos < ncols;
            int ch = dy.coff + c0ch + (ok ? row : 0);
            long long g = ((long long)n * dy.ctot + ch) * cs + (long long)t0 * V + (ok ? pos : 0);
            c1[i] = dy.coef ? dy.coef[ch] : 1.f;
            c2[i] = (dy.coef && dy.x2) ? dy.coef[dy.ctot + ch] : 0.f;
            c0[i] = dy.coef ? dy.coef[2 * dy.ctot + ch] : 0.f;
            float4 a4 = ok ? *reinterpret_cast<const float4*>(dy.x1 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b4 = (ok && dy.x2) ? *reinterpret_cast<const float4*>(dy.x2 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            v1[i][0] = a4.x; v1[i][1] = a4.y; v1[i][2] = a4.z; v1[i][3] = a4.w;
            v2[i][0] = b4.x; v2[i][1] = b4.y; v2[i][2] = b4.z; v2[i][3] = b4.w;
        }
    }
    __device__ __forceinline__ void pin() {             // make the compiler wait for the loads HERE (see X3Pref)
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            asm volatile("" : "+v"(v1[i][0]), "+v"(v1[i][1]), "+v"(v1[i][2]), "+v"(v1[i][3]));
            asm volatile("" : "+v"(v2[i][0]), "+v"(v2[i][1]), "+v"(v2[i][2]), "+v"(v2[i][3]));
            asm volatile("" : "+v"(c1[i]), "+v"(c2[i]), "+v"(c0[i]));
        }
    }
    __device__ __forceinline__ void commit(const SrcDev& dy, float* Zs) {
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            int e = threadIdx.x + i * G::NT;
            int row = e / ROWV, pos = (e - row * ROWV) * 4;
            if (row < G::CT) {
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = fmaf(c1[i], v1[i][k], fmaf(c2[i], v2[i][k], c0[i]));
                    o[k] = dy.act == 1 ? fmaxf(v, 0.f) : v;
                }
                *reinterpret_cast<float4*>(Zs + row * G::NCOLS + pos) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <class G, int ST>
__global__ __launch_bounds__(G::NT, 2) void ctrgc_fwd_kernel(const CtrgcArgs a, float* y, float* stats_part, float* x3_out) {
    using P = Plan<G, ST>;
    constexpr int V = G::V, CT = G::CT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords<G>(a, n, c0)) return;
    float* Es = smem;                                  // [CT][V][S*V]: row (c, u) = E_s[c][u][v] over (s, v)
    float* X3 = Es + P::ETILE;                         // REGION floats: GEMM stage, then the x3 tile [CT][BT][S*V]
    float* Zs = X3 + P::REGION;                        // [CT][NCOLS]
    const int tid = threadIdx.x;
    const int c = tid / (G::NTQ * 4);
    const int lrow = tid % (G::NTQ * 4);               // lane index inside the channel row (copy-out)

    TG_T(tt0);
    X3Pref<G, ST> pf;
    pf.load(a, n, c0, 0, min(G::BT, a.T), 0);          // first operands of the first chunk: in flight under the E load
    load_E_k<G, ST>(a.E, a.Cout, n, c0, Es);
    TG_T(tt1); TG_ACC(0, tt1 - tt0);

    float st1 = 0.f, st2 = 0.f;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        const int nt0 = t0 + G::BT, nbt = nt0 < a.T ? min(G::BT, a.T - nt0) : 0;
        TG_T(ta);
        x3_chunk<G, ST>(a, n, c0, t0, bt, X3, pf, nt0, nbt);
        TG_T(tb); TG_ACC(1, tb - ta);
        if (!(TG_CKO & 16)) aggregate_mfma<G, ST>(Es, X3, Zs, bt);      // frames beyond bt hold stale data: their rows are not stored
        TG_T(tc); TG_ACC(2, tc - tb);
        __syncthreads();
        TG_T(td); TG_ACC(3, td - tc);
        // copy-out: every LDS read of the pass is issued before the first store leaves (a store waits for its own
        // read only; vmcnt retires in order, so interleaving reads and stores serialises them)
        constexpr int RL = G::NTQ * 4;                              // lanes per channel row
        constexpr int NV4 = (G::NCOLS / 4 + RL - 1) / RL;           // float4 per lane and row
        float* yrow = y + (((long long)n * a.Cout + c0 + c) * a.T + t0) * V;
        float4 zv[NV4];
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int p4 = lrow + i * RL;
            zv[i] = p4 < (ncols >> 2) ? reinterpret_cast<const float4*>(Zs + c * G::NCOLS)[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (nbt > 0) pf.pin();    // the next chunk's operands have landed: no load is left in front of the stores below
        if (x3_out) {             // keep x3 for the backward (saves recomputing the GEMM there)
            float4 xv[ST][NV4];
#pragma unroll
            for (int s = 0; s < ST; ++s)
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const int p4 = lrow + i * RL;                  // float4 p4 of the output row = (frame p4 / (V/4), joints 4*(p4 % (V/4)) ..)
                    const int fr = p4 / (V / 4), q = p4 - fr * (V / 4);
                    xv[s][i] = p4 < (ncols >> 2) ? *reinterpret_cast<const float4*>(X3 + (c * G::BT + fr) * (ST * V) + s * V + 4 * q)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                float* xo = x3_out + (((long long)n * ST * a.Cout + s * a.Cout + c0 + c) * a.T + t0) * V;
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const int p4 = lrow + i * RL;
                    if (p4 < (ncols >> 2) && (!(TG_CKO & 32) || xv[s][i].x == 1.2345f)) reinterpret_cast<float4*>(xo)[p4] = xv[s][i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int p4 = lrow + i * RL;
            if (p4 < (ncols >> 2)) {
                if (!(TG_CKO & 32) || zv[i].x == 1.2345f) reinterpret_cast<float4*>(yrow)[p4] = zv[i];
                st1 += (zv[i].x + zv[i].y) + (zv[i].z + zv[i].w);
                st2 = fmaf(zv[i].x, zv[i].x, fmaf(zv[i].y, zv[i].y, fmaf(zv[i].z, zv[i].z, fmaf(zv[i].w, zv[i].w, st2))));
            }
        }
        // next chunk's first barrier (inside x3_chunk) protects Zs / X3 reuse
        TG_T(te); TG_ACC(4, te - td);
    }
    TG_T(tt2); TG_ACC(8, tt2 - tt0); TG_ACC(9, 1);
    if (stats_part) {
        // reduce over the NTQ*4 threads of the channel row (consecutive lanes of one wave)
#pragma unroll
        for (int o = 1; o < G::NTQ * 4; o <<= 1) { st1 += __shfl_xor(st1, o); st2 += __shfl_xor(st2, o); }
        if (lrow == 0) {
            stats_part[((long long)0 * a.Cout + c0 + c) * a.N + n] = st1;
            stats_part[((long long)1 * a.Cout + c0 + c) * a.N + n] = st2;
        }
    }
}

// ---------------------------------------------------------------------------
// backward: dx3_s[c][t][v] = sum_u dy[c][t][u] * E_s[c][u][v]  -- per channel a (16 frames) x (S*V columns) x (V joints)
// product on the matrix cores: A = the dy tile row (c, frame), k = u; B = the E rows of the forward's layout
// Ek[c][u][s*V + v] read along (s, v); k = 4*ks + kq.  Wave w owns channels w, w + NW, ...
// ---------------------------------------------------------------------------
template <class G, int ST>
__device__ __forceinline__ void dx3_mfma(const float* Ek, const float* Zs, float* X3, int bt) {
    constexpr int V = G::V, CT = G::CT, KP = ST * V, NKS = (V + 3) / 4, NCT = (KP + 15) / 16;
    static_assert(G::BT == 16, "one 16-frame MFMA row tile per chunk");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    for (int c = wave; c < CT; c += G::NW) {
        float av[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int u = 4 * ks + kq;
            av[ks] = u < V ? Zs[c * G::NCOLS + j * V + u] : 0.f;
        }
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int u = 4 * ks + kq < V ? 4 * ks + kq : V - 1;      // rows beyond V meet a zero A value
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int col = ct * 16 + j < KP ? ct * 16 + j : KP - 1;   // padding columns: discarded below
                acc[ct] = mfma16(av[ks], Ek[(c * V + u) * KP + col], acc[ct]);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = ct * 16 + j;
            const int sidx = col / V, v = col - sidx * V;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = kq * 4 + r;
                if (col < KP && fr < bt) X3[(sidx * CT + c) * G::PX3 + fr * V + v] = acc[ct][r];
            }
        }
    }
}

template <class G, int ST>
__global__ __launch_bounds__(G::NT, 2) void ctrgc_bwd_dx3_kernel(const CtrgcArgs a, const SrcDev dy, float* dx3, float* db3_part) {
    using P = Plan<G, ST>;
    constexpr int V = G::V, CT = G::CT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords<G>(a, n, c0)) return;
    float* Es = smem;                                  // [CT][V][S*V], the forward's layout
    float* X3 = Es + P::ETILE;                         // output staging [S*CT][PX3]
    float* Zs = X3 + P::REGION;                        // dy chunk [CT][NCOLS]
    const int tid = threadIdx.x;
    const int c = tid / (G::NTQ * 4);
    const int lrow = tid % (G::NTQ * 4);

    DyTile<G> dyt;
    dyt.load(dy, n, c0, a.T, 0, min(G::BT, a.T));      // in flight under the E load
    load_E_k<G, ST>(a.E, a.Cout, n, c0, Es);

    float sb[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) sb[s] = 0.f;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        __syncthreads();
        dyt.commit(dy, Zs);
        __syncthreads();
        if (t0 + G::BT < a.T) dyt.load(dy, n, c0, a.T, t0 + G::BT, min(G::BT, a.T - t0 - G::BT));   // next chunk in flight
        dx3_mfma<G, ST>(Es, Zs, X3, bt);
        __syncthreads();
        if (t0 + G::BT < a.T) dyt.pin();               // the next dy chunk has landed before the first store below leaves
        constexpr int RL = G::NTQ * 4;
        constexpr int NV4 = (G::NCOLS / 4 + RL - 1) / RL;
        float4 xv[ST][NV4];
#pragma unroll
        for (int s = 0; s < ST; ++s)
#pragma unroll
            for (int i = 0; i < NV4; ++i) {
                const int p4 = lrow + i * RL;
                xv[s][i] = p4 < (ncols >> 2) ? reinterpret_cast<const float4*>(X3 + (s * CT + c) * G::PX3)[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            float* orow = dx3 + (((long long)n * ST * a.Cout + s * a.Cout + c0 + c) * a.T + t0) * V;
#pragma unroll
            for (int i = 0; i < NV4; ++i) {
                const int p4 = lrow + i * RL;
                if (p4 < (ncols >> 2)) {
                    reinterpret_cast<float4*>(orow)[p4] = xv[s][i];
                    sb[s] += (xv[s][i].x + xv[s][i].y) + (xv[s][i].z + xv[s][i].w);
                }
            }
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
// host
// ---------------------------------------------------------------------------
using G20 = Geo<20, 8, 2, 8, 16>;      // 8 channels x 16 frames per chunk, 256 threads, 78 KB: two workgroups per CU
using G20W = Geo<20, 16, 2, 8, 32>;    // 16 channels, 512 threads, 155 KB: one workgroup per CU, 48 GEMM rows = three full MFMA row tiles

static int fill_args(const tamgcn_ctrgc_desc* d, CtrgcArgs* a, const char* who, int ct) {
    if (!(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->T > 0)) { tamgcn_set_error("%s: bad dims", who); return -1; }
    if (d->V != 20 || (d->S != 1 && d->S != 3)) {
        tamgcn_set_error("%s: unsupported S=%d V=%d (the LDS-resident kernels exist for S in {1,3}, V = 20; V in {25, 32, 64}: tamgcn_ctrgc_tiled_*)", who, d->S, d->V);
        return -1;
    }
    if (d->Cout % ct) { tamgcn_set_error("%s: Cout=%d must be a multiple of %d", who, d->Cout, ct); return -1; }
    if (!(d->x.x1 && d->w3 && d->b3)) { tamgcn_set_error("%s: null pointer", who); return -1; }
    if (!d->E) { tamgcn_set_error("%s: d->E is NULL (build it with tamgcn_ctrgc_build_e)", who); return -1; }
    if (d->x.x2 || d->x.coef || d->x.act) { tamgcn_set_error("%s: x must be a plain tensor (no fused prologue)", who); return -1; }
    if (d->x.coff + d->Cin > d->x.ctot) { tamgcn_set_error("%s: x channel slice out of range", who); return -1; }
    if ((long long)d->x.ctot * d->T * d->V >= (1LL << 31) || (long long)d->S * d->Cout * d->T * d->V >= (1LL << 31)) {
        tamgcn_set_error("%s: a sample block of >= 2^31 elements", who); return -1;
    }
    a->N = d->N; a->Cin = d->Cin; a->Cout = d->Cout; a->S = d->S; a->T = d->T;
    a->x = d->x.x1; a->x_ctot = d->x.ctot; a->x_coff = d->x.coff;
    a->w3 = d->w3; a->b3 = d->b3; a->E = d->E;
    a->nct = d->Cout / ct;
    return 0;
}

static unsigned grid_blocks(const CtrgcArgs& a) { return 8u * (unsigned)ceil_div(a.N, 8) * (unsigned)a.nct; }

#define CTRGC_LAUNCH(KERNEL, GEO, ST_, FLAG, ...)                                                               \
    do {                                                                                                        \
        static tg_devmask FLAG = 0;                                                                             \
        constexpr size_t lds_ = Plan<GEO, ST_>::LDS;                                                            \
        tg_allow_lds((const void*)KERNEL<GEO, ST_>, lds_, &FLAG);   /* exact size */                            \
        if (getenv("TAMGCN_DEBUG_OCC")) {                                                                       \
            int nb_ = -1;                                                                                       \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, (const void*)KERNEL<GEO, ST_>, GEO::NT, lds_); \
            fprintf(stderr, "[tamgcn] %s<CT %d, %d>: %d workgroups per CU (%zu B LDS, %d threads)\n", #KERNEL, GEO::CT, ST_, nb_, lds_, GEO::NT); \
        }                                                                                                       \
        hipLaunchKernelGGL((KERNEL<GEO, ST_>), dim3(grid_blocks(a)), dim3(GEO::NT), lds_, (hipStream_t)stream, __VA_ARGS__); \
        tamgcn_note_kernel(#KERNEL "<Geo<%d, %d, %d, %d, %d>, %d>", GEO::V, GEO::CT, GEO::TB, GEO::NTQ, GEO::SBK, ST_); \
    } while (0)

// The forward's channel tile: 16 (48 GEMM rows = three full MFMA row tiles, x read by half as many workgroups) wherever
// Cout allows; measured on one box against the 8-channel form (two workgroups per CU, 24 rows in two padded tiles):
// 216 / 429 / 317 / 627 / 521 us against 243 / 462 / 357 / 696 / 608 us at the five layer shapes (profiles/r03_ctrgc_ab.txt).
static int fwd_ct(int Cout) { return Cout % 16 ? 8 : 16; }

}  // namespace

int tamgcn_ctrgc_tiled_lds_bytes(int S, int V, int R);      // ctrgc_tiled.hip: the streaming family (V in {25, 32, 64})

extern "C" int tamgcn_ctrgc_lds_bytes(int S, int V, int R) {
    if (S != 1 && S != 3) return -1;
    if (V == 20) return S == 3 ? (int)Plan<G20W, 3>::LDS : (int)Plan<G20W, 1>::LDS;   // the fused forward's workgroup (the largest)
    if (V == 25) {                                      // streaming route: the E builder is its largest request
        if (R < 4 || R > 32 || R % 4) return -1;
        return (int)(sizeof(float) * ((size_t)(16 + R) * V * V + 2 * (size_t)R * V));
    }
    return tamgcn_ctrgc_tiled_lds_bytes(S, V, R);
}

extern "C" int tamgcn_ctrgc_build_e(const tamgcn_ctrgc_desc* d, float* E, void* stream) {
    TG_CHECK(d && E && d->pq && d->w4 && d->b4 && d->A && d->alpha, "tamgcn_ctrgc_build_e: null pointer");
    TG_CHECK(d->N > 0 && d->S > 0 && d->Cout > 0 && d->Cout % 16 == 0, "tamgcn_ctrgc_build_e: bad shape N=%d S=%d Cout=%d", d->N, d->S, d->Cout);
    TG_CHECK(d->R >= 4 && d->R <= 32 && d->R % 4 == 0, "tamgcn_ctrgc_build_e: R=%d outside 4..32 (multiples of 4)", d->R);
    TG_CHECK(d->V == 20 || d->V == 25, "tamgcn_ctrgc_build_e: unsupported V=%d (V in {20,25})", d->V);
    EArgs a;
    a.N = d->N; a.Cout = d->Cout; a.S = d->S; a.R = d->R;
    a.pq = d->pq; a.w4 = d->w4; a.b4 = d->b4; a.A = d->A; a.alpha = d->alpha; a.E = E;
    const size_t lds = sizeof(float) * ((size_t)d->R * d->V * d->V + 2 * (size_t)d->R * d->V);
    if (d->V == 20) {
        static tg_devmask f = 0;
        tg_allow_lds((const void*)ctrgc_E_kernel<20>, 160 * 1024, &f);
        hipLaunchKernelGGL((ctrgc_E_kernel<20>), dim3(d->N * d->S), dim3(512), lds, (hipStream_t)stream, a);
    } else {
        static tg_devmask f = 0;
        tg_allow_lds((const void*)ctrgc_E_kernel<25>, 160 * 1024, &f);
        hipLaunchKernelGGL((ctrgc_E_kernel<25>), dim3(d->N * d->S), dim3(512), lds, (hipStream_t)stream, a);
    }
    tamgcn_note_kernel("ctrgc_E_kernel<%d>", d->V);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_build_e");
    return 0;
}

extern "C" int tamgcn_ctrgc_fwd(const tamgcn_ctrgc_desc* d, float* y, float* stats_part, float* x3_out, void* stream) {
    TG_CHECK(d && y, "tamgcn_ctrgc_fwd: null pointer");
    CtrgcArgs a;
    const int ct = fwd_ct(d->Cout);
    if (fill_args(d, &a, "tamgcn_ctrgc_fwd", ct)) return -1;
    if (ct == 16) {
        if (d->S == 3) CTRGC_LAUNCH(ctrgc_fwd_kernel, G20W, 3, fw3, a, y, stats_part, x3_out);
        else CTRGC_LAUNCH(ctrgc_fwd_kernel, G20W, 1, fw1, a, y, stats_part, x3_out);
    } else {
        if (d->S == 3) CTRGC_LAUNCH(ctrgc_fwd_kernel, G20, 3, f3, a, y, stats_part, x3_out);
        else CTRGC_LAUNCH(ctrgc_fwd_kernel, G20, 1, f1, a, y, stats_part, x3_out);
    }
    TG_LAUNCH_CHECK("tamgcn_ctrgc_fwd");
    return 0;
}

extern "C" int tamgcn_ctrgc_bwd_dx3(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, float* dx3, float* db3_part, void* stream) {
    TG_CHECK(d && dy && dy->x1 && dx3, "tamgcn_ctrgc_bwd_dx3: null pointer");
    TG_CHECK(dy->ctot >= dy->coff + d->Cout, "tamgcn_ctrgc_bwd_dx3: dy has %d channels from %d, need %d", dy->ctot, dy->coff, d->Cout);
    CtrgcArgs a;
    tamgcn_ctrgc_desc dd = *d;                           // x, w3, b3 are not read by this kernel: only their presence is checked
    if (!dd.w3) dd.w3 = (const float*)d->E;
    if (!dd.b3) dd.b3 = (const float*)d->E;
    if (fill_args(&dd, &a, "tamgcn_ctrgc_bwd_dx3", G20::CT)) return -1;
    if (d->S == 3) CTRGC_LAUNCH(ctrgc_bwd_dx3_kernel, G20, 3, f3, a, make_src(*dy), dx3, db3_part);
    else CTRGC_LAUNCH(ctrgc_bwd_dx3_kernel, G20, 1, f1, a, make_src(*dy), dx3, db3_part);
    TG_LAUNCH_CHECK("tamgcn_ctrgc_bwd_dx3");
    return 0;
}
