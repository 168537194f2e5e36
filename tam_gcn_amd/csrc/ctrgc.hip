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
// Round 4 added two forms of the 16-channel forward WITHOUT the resident E tile (80 KB of LDS, two workgroups per CU, 128 registers
// per wave): ctrgc_fwd_kernel<G, ST, false> (this kernel with the aggregation's E fragments read from global memory) and
// ctrgc_fwd2_kernel (its operands by LDS-DMA as well); tamgcn_ctrgc_fwd picks by Cin -- see the comment above ctrgc_fwd2_kernel.
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
    static constexpr size_t LDS_NOE = sizeof(float) * ((size_t)REGION + (size_t)G::CT * G::NCOLS);   // E read from L2 per channel (forward, ER = false)
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
// wave-uniform base + 32-bit byte offset: one SGPR pair and one VGPR per address (the 64-bit per-lane pointers the compiler
// otherwise builds -- and hoists out of the frame loop -- cost two registers each)
template <class T>
__device__ __forceinline__ const T* cg_at(const void* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ T* cg_at_w(void* base, unsigned byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}

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

    __device__ __forceinline__ void load(const CtrgcArgs& a, int n, int c0, int t0, int bt, int k0, int tid) {
        if (TG_CKO & 2) return;
        constexpr int V = G::V;
        const int ncols = bt * V;
        const int cs = a.T * V;                                     // host: 4 * SBK * T * V < 2^32
        const float* xk = a.x + ((long long)n * a.x_ctot + a.x_coff + k0) * cs + (long long)t0 * V;    // wave-uniform
        const float* wk = a.w3 + (long long)c0 * a.Cin + k0;                                            // wave-uniform
#pragma unroll
        for (int i = 0; i < NAF; ++i) {
            const int e = tid + i * NT;
            const int kk = e % SBK, row = e / SBK;
            const int sidx = row / CT, c = row - sidx * CT;
            const bool ok = row < P::NR && k0 + kk < a.Cin;
            wv[i] = ok ? *cg_at<float>(wk, 4u * (unsigned)((sidx * a.Cout + c) * a.Cin + kk)) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + i * NT;
            const int kk = e / ROWV, pos = (e - kk * ROWV) * 4;
            const bool ok = kk < SBK && k0 + kk < a.Cin && pos < ncols;
            rv[i] = ok ? *cg_at<float4>(xk, 4u * (unsigned)(kk * cs + pos)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) asm volatile("" : "+v"(rv[i].x), "+v"(rv[i].y), "+v"(rv[i].z), "+v"(rv[i].w));
#pragma unroll
        for (int i = 0; i < NAF; ++i) asm volatile("" : "+v"(wv[i]));
    }
    __device__ __forceinline__ void commit(float* As, float* Bs, int tid) const {    // registers -> LDS stage
        if (TG_CKO & 4) return;
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
template <class G, int ST, class Pre>
__device__ __forceinline__ void x3_chunk(const CtrgcArgs& a, int n, int c0, int t0, int bt, float* X3, X3Pref<G, ST>& pf, int next_t0, int next_bt, int tid, Pre&& pre) {
    // the tile leaves as X3[(c*BT + frame)*KP + s*V + v], KP = S*V: a (channel, frame) row holds the aggregation's k axis
    using P = Plan<G, ST>;
    constexpr int V = G::V, CW = G::CW, SBK = G::SBK, SBKP = G::SBKP, CT = G::CT;
    constexpr int NR = P::NR, NRT = P::NRT, PB = G::PITCHB;
    const int lane = tid & 63, wave = tid >> 6;
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
        pf.commit(As, Bs, tid);
        __syncthreads();
        if (k0 + SBK < a.Cin) pf.load(a, n, c0, t0, bt, k0 + SBK, tid);      // in flight under the MFMAs
        else if (next_bt > 0) pf.load(a, n, c0, next_t0, next_bt, 0, tid);   // the next frame chunk's first K chunk
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
    pre();                                             // the caller's requests that should travel under the tile write
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

// The same product with the B fragments (rows of E) taken from global memory instead of an LDS-resident tile: lane (j, kq)'s vector m
// of row u is E[n][s][c][u][v0 .. v0+3] with 16 m + 4 kq = s V + v0 (V % 4 == 0: a vector never leaves its subset) -- the layout
// tamgcn_ctrgc_build_e writes, read as it lies.  A workgroup re-reads its 77 KB per frame chunk from L2; without the tile two
// 16-channel workgroups share a CU.  ld() requests one channel's fragments; the caller requests the first channel's before the
// x3 tile write so that they travel under it.
template <class G, int ST>
struct EFrag {
    static constexpr int V = G::V, KP = ST * V, NM = (KP + 15) / 16, NUT = (V + 15) / 16;
    f32x4 bv[NUT][NM];
    __device__ __forceinline__ void ld(const float* __restrict__ En, int Cout, int ch, int tid) {   // En = E + n*S*Cout*VV, ch = absolute channel (wave-uniform)
        const int lane = tid & 63, j = lane & 15, kq = lane >> 4;
        const float* Ec = En + ch * G::VV;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int k = 16 * m + 4 * kq;
            const bool in = k < KP;
            const int kk = in ? k : 0, sidx = kk / V, v0 = kk - sidx * V;
#pragma unroll
            for (int ut = 0; ut < NUT; ++ut) {
                const int u = ut * 16 + j < V ? ut * 16 + j : V - 1;
                const f32x4 v = *cg_at<f32x4>(Ec, 4u * (unsigned)(sidx * Cout * G::VV + u * V + v0));
                bv[ut][m] = in ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
};

template <class G, int ST, bool DB = false>      // DB: the next channel's fragments in a second register set
__device__ __forceinline__ void aggregate_mfma_g(const float* __restrict__ En, int Cout, int c0, EFrag<G, ST>& e0, const float* X3, float* Zs, int bt, int tid) {
    constexpr int V = G::V, CT = G::CT, KP = ST * V, NM = (KP + 15) / 16, NUT = (V + 15) / 16;
    static_assert(G::BT == 16, "one 16-frame MFMA row tile per chunk");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int ci = 0; ci < CT / G::NW; ++ci) {
        const int c = wave + ci * G::NW;
        EFrag<G, ST> en;
        if (DB && ci + 1 < CT / G::NW) en.ld(En, Cout, c0 + c + G::NW, tid);
        f32x4 av[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const bool in = 16 * m + 4 * kq < KP;
            av[m] = *reinterpret_cast<const f32x4*>(X3 + (c * G::BT + j) * KP + (in ? 16 * m + 4 * kq : 0));
            if (!in) av[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc[NUT];
#pragma unroll
        for (int ut = 0; ut < NUT; ++ut) acc[ut] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ut = 0; ut < NUT; ++ut) acc[ut] = mfma16(av[m][r], e0.bv[ut][m][r], acc[ut]);
#pragma unroll
        for (int ut = 0; ut < NUT; ++ut) {
            const int u = ut * 16 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = kq * 4 + r;
                if (u < V && fr < bt) Zs[c * G::NCOLS + fr * V + u] = acc[ut][r];
            }
        }
        if (ci + 1 < CT / G::NW) {
            if (DB) e0 = en;
            else e0.ld(En, Cout, c0 + c + G::NW, tid);                        // no registers for a second set
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
template <class G, int ST, bool ER = true>       // ER: the E tile resident in LDS (one 16-channel workgroup per CU) or read from L2 per channel (two)
__global__ __launch_bounds__(G::NT, ER ? 2 : 4) void ctrgc_fwd_kernel(const CtrgcArgs a, float* y, float* stats_part, float* x3_out) {
    using P = Plan<G, ST>;
    constexpr int V = G::V, CT = G::CT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords<G>(a, n, c0)) return;
    float* Es = smem;                                  // [CT][V][S*V]: row (c, u) = E_s[c][u][v] over (s, v)
    float* X3 = ER ? Es + P::ETILE : smem;             // REGION floats: GEMM stage, then the x3 tile [CT][BT][S*V]
    float* Zs = X3 + P::REGION;                        // [CT][NCOLS]
    const int tid0 = threadIdx.x;
    const float* En = a.E + (long long)n * ST * a.Cout * G::VV;
    static_assert(ER || CT % G::NW == 0, "whole channels per wave");

    TG_T(tt0);
    X3Pref<G, ST> pf;
    pf.load(a, n, c0, 0, min(G::BT, a.T), 0, tid0);          // first operands of the first chunk: in flight under the E load
    if constexpr (ER) load_E_k<G, ST>(a.E, a.Cout, n, c0, Es);
    TG_T(tt1); TG_ACC(0, tt1 - tt0);

    float st1 = 0.f, st2 = 0.f;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        const int nt0 = t0 + G::BT, nbt = nt0 < a.T ? min(G::BT, a.T - nt0) : 0;
        // 128 registers per wave (ER = false): everything derived from the thread index is recomputed per frame chunk instead of
        // being hoisted out of this loop and spilled (a scratch reload is a vmcnt wait behind every prefetch in flight)
        int tid = tid0;
        if constexpr (!ER) asm volatile("" : "+v"(tid));
        const int c = tid / (G::NTQ * 4);
        const int lrow = tid % (G::NTQ * 4);           // lane index inside the channel row (copy-out)
        TG_T(ta);
        EFrag<G, ST> ef;
        x3_chunk<G, ST>(a, n, c0, t0, bt, X3, pf, nt0, nbt, tid, [&] { if constexpr (!ER) ef.ld(En, a.Cout, c0 + __builtin_amdgcn_readfirstlane(tid >> 6), tid); });
        TG_T(tb); TG_ACC(1, tb - ta);
        if constexpr (!ER) aggregate_mfma_g<G, ST>(En, a.Cout, c0, ef, X3, Zs, bt, tid);
        else if (!(TG_CKO & 16)) aggregate_mfma<G, ST>(Es, X3, Zs, bt);      // frames beyond bt hold stale data: their rows are not stored
        TG_T(tc); TG_ACC(2, tc - tb);
        __syncthreads();
        TG_T(td); TG_ACC(3, td - tc);
        // copy-out: every LDS read of the pass is issued before the first store leaves (a store waits for its own
        // read only; vmcnt retires in order, so interleaving reads and stores serialises them)
        constexpr int RL = G::NTQ * 4;                              // lanes per channel row
        constexpr int NV4 = (G::NCOLS / 4 + RL - 1) / RL;           // float4 per lane and row
        float* ybase = y + (((long long)n * a.Cout + c0) * a.T + t0) * V;          // wave-uniform; rows by 32-bit byte offsets
        const unsigned rowb = 4u * (unsigned)(c * a.T * V);
        float4 zv[NV4];
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int p4 = lrow + i * RL;
            zv[i] = p4 < (ncols >> 2) ? reinterpret_cast<const float4*>(Zs + c * G::NCOLS)[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (nbt > 0) pf.pin();    // the next chunk's operands have landed: no load is left in front of the stores below
        if constexpr (!ER) {      // 128 registers per wave: one subset's vectors at a time
            if (x3_out) {
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    float4 xs[NV4];
#pragma unroll
                    for (int i = 0; i < NV4; ++i) {
                        const int p4 = lrow + i * RL;
                        const int fr = p4 / (V / 4), q = p4 - fr * (V / 4);
                        xs[i] = p4 < (ncols >> 2) ? *reinterpret_cast<const float4*>(X3 + (c * G::BT + fr) * (ST * V) + s * V + 4 * q)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    float* xo = x3_out + (((long long)n * ST * a.Cout + s * a.Cout + c0) * a.T + t0) * V;
#pragma unroll
                    for (int i = 0; i < NV4; ++i) {
                        const int p4 = lrow + i * RL;
                        if (p4 < (ncols >> 2)) *cg_at_w<float4>(xo, rowb + 16u * (unsigned)p4) = xs[i];
                    }
                }
            }
        } else
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
                float* xo = x3_out + (((long long)n * ST * a.Cout + s * a.Cout + c0) * a.T + t0) * V;
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const int p4 = lrow + i * RL;
                    if (p4 < (ncols >> 2) && (!(TG_CKO & 32) || xv[s][i].x == 1.2345f)) *cg_at_w<float4>(xo, rowb + 16u * (unsigned)p4) = xv[s][i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int p4 = lrow + i * RL;
            if (p4 < (ncols >> 2)) {
                if (!(TG_CKO & 32) || zv[i].x == 1.2345f) *cg_at_w<float4>(ybase, rowb + 16u * (unsigned)p4) = zv[i];
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
        const int c = tid0 / (G::NTQ * 4), lrow = tid0 % (G::NTQ * 4);
        if (lrow == 0) {
            stats_part[((long long)0 * a.Cout + c0 + c) * a.N + n] = st1;
            stats_part[((long long)1 * a.Cout + c0 + c) * a.N + n] = st2;
        }
    }
}

// ---------------------------------------------------------------------------
// forward, second form (round 4): TWO 16-channel workgroups per CU.  profiles/r04_ctrgc_fwd_knockout.txt: the x3 GEMM of the form
// above already runs at the rate the matrix cores sustain; the aggregation, the tile write, the copy-out and the start-up are
// serial to it, and with E resident (77 KB) nothing else fits the CU.  Here
//   * E is NOT resident: the aggregation's B fragments are 16-byte rows of E as tamgcn_ctrgc_build_e wrote them, loaded into
//     registers one channel ahead (EFrag); the workgroup's 77 KB stay in L2 between frame chunks.  LDS = x3 tile + z tile = 80 KB;
//   * the x3 GEMM's operands travel by LDS-DMA into two 25 KB stages inside the x3 tile's region (16 input channels each):
//     no staging registers, no commit pass, one barrier per K chunk -- the register-staged form spent 28 % of the launch on the
//     loads and their commit once two workgroups shared the CU;
//   * 128 registers per wave: everything derived from the thread index is recomputed per frame chunk (an opaque copy of it) --
//     hoisted out of the loop the compiler spilled 143 registers, and a scratch reload is a vmcnt wait behind every request in flight;
//   * the first stage of the NEXT frame chunk is requested after the copy-out's LDS reads and BEFORE its global stores (vmcnt
//     retires in order): the next chunk's first wait allows exactly those stores to be still in flight.
// Applies to S = 3, Cout % 16 == 0, Cin % 16 == 0, 16-byte aligned x and w3; everything else stays on ctrgc_fwd_kernel.
// ---------------------------------------------------------------------------
#define CG_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void cg_wait_vmcnt(int n) {     // n is wave-uniform
    switch (n) {
        CG_VMCNT_CASE(0) CG_VMCNT_CASE(1) CG_VMCNT_CASE(2) CG_VMCNT_CASE(3) CG_VMCNT_CASE(4) CG_VMCNT_CASE(5) CG_VMCNT_CASE(6)
        CG_VMCNT_CASE(7) CG_VMCNT_CASE(8) CG_VMCNT_CASE(9) CG_VMCNT_CASE(10) CG_VMCNT_CASE(11) CG_VMCNT_CASE(12)
        CG_VMCNT_CASE(13) CG_VMCNT_CASE(14) CG_VMCNT_CASE(15) CG_VMCNT_CASE(16) CG_VMCNT_CASE(17) CG_VMCNT_CASE(18)
        CG_VMCNT_CASE(19) CG_VMCNT_CASE(20) CG_VMCNT_CASE(21) CG_VMCNT_CASE(22) CG_VMCNT_CASE(23) CG_VMCNT_CASE(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <class G, int ST>
struct Fwd2 {
    using P = Plan<G, ST>;
    static constexpr int SBK = 16, PB = G::PITCHB, AP = 20;           // input channels per stage; x row pitch, weight row pitch (floats)
    static constexpr int RS = PB / 4;                                 // 16-byte slots per x row (NCOLS / 4 of them used)
    static constexpr int XSL = SBK * RS;                              // slots of the x image
    static constexpr int ASL = P::NR * (AP / 4);                      // slots of the weight image (AP / 4 per row, SBK / 4 used)
    static constexpr int NPIECE = (XSL + ASL + 63) / 64, NPW = (NPIECE + G::NW - 1) / G::NW;
    static constexpr int STG = NPIECE * 256;                          // floats per stage: whole 1 KB pieces
    static_assert(P::NR == P::NRT * 16, "full weight row tiles");
    static_assert(2 * STG <= P::REGION, "two stages inside the x3 tile's region");
    static_assert(3 * STG <= P::REGION + G::CT * G::NCOLS, "a third stage may reach into the z tile (dead while the K loop runs)");
    static_assert(XSL % 64 == 0, "a piece holds x slots or weight slots, not both");
};

template <class G, int ST, int NSTG>     // NSTG stages of 16 input channels: NSTG - 1 in flight while one is consumed
__global__ __launch_bounds__(G::NT, 4) void ctrgc_fwd2_kernel(const CtrgcArgs a, float* y, float* stats_part, float* x3_out) {
    using P = Plan<G, ST>;
    using F = Fwd2<G, ST>;
    constexpr int V = G::V, CT = G::CT, NRT = P::NRT, CW = G::CW, PB = F::PB, AP = F::AP, SBK = F::SBK;
    static_assert(G::NW == 8 && G::NCT == 20 && CW == 3 && CT % G::NW == 0, "the 16-channel geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int n, c0;
    if (!block_coords<G>(a, n, c0)) return;
    float* X3 = smem;                                  // REGION floats: GEMM stages, then the x3 tile [CT][BT][S*V]
    float* Zs = X3 + P::REGION;                        // [CT][NCOLS]; a third stage reaches into it -- only while the K loop runs, when nobody holds a z tile
    const int tid0 = threadIdx.x;
    const float* En = a.E + (long long)n * ST * a.Cout * G::VV;
    const int cs = a.T * V;                            // host: 64 * T * V < 2^32
    const int nk = a.Cin / SBK;                        // host: Cin % 16 == 0
    const float* xn = a.x + ((long long)n * a.x_ctot + a.x_coff) * cs;    // wave-uniform
    const float* wk0 = a.w3 + (long long)c0 * a.Cin;                      // wave-uniform
    constexpr int RL = G::NTQ * 4;                                        // lanes per channel row (copy-out)
    constexpr int NV4 = (G::NCOLS / 4 + RL - 1) / RL;                     // float4 per lane and row
    const int nst = NV4 * (x3_out ? ST + 1 : 1);                          // store instructions of one copy-out

    // one K chunk (16 input channels from k0) of the frames from t0 into stage `stage`: per lane a 16-byte slot of each of the
    // wave's pieces -- the image is linear in LDS, the layout sits in the source address
    auto issue = [&](int stage, int t0, int ncols, int k0, int tid) -> int {   // returns the number of requests this wave made (wave-uniform)
        if (TG_CKO & 2) return 0;
        int cnt = 0;
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        float* st = X3 + stage * F::STG;
        const float* xk = xn + (long long)k0 * cs + (long long)t0 * V;
        const float* wk = wk0 + k0;
#pragma unroll
        for (int i = 0; i < F::NPW; ++i) {
            const int p = wave + i * G::NW;                                // wave-uniform
            if (p < F::NPIECE) {
                const int L = p * 64 + lane;
                if (p * 64 < F::XSL) {
                    const int row = L / F::RS, c4 = L - row * F::RS;
                    const bool ok = c4 * 4 < ncols;
                    if (__ballot(ok) != 0ull) {                              // the request exists or not for the whole wave: it is counted
                        if (ok) __builtin_amdgcn_global_load_lds((cg_gptr)cg_at<float>(xk, 4u * (unsigned)(row * cs + c4 * 4)), (cg_lptr)(st + p * 256), 16, 0, 0);
                        ++cnt;
                    }
                } else {
                    const int La = L - F::XSL;
                    const int row = La / (AP / 4), q = La - row * (AP / 4);
                    const int sidx = row / CT, c = row - sidx * CT;
                    const bool ok = q < SBK / 4 && La < F::ASL;
                    if (__ballot(ok) != 0ull) {
                        if (ok) __builtin_amdgcn_global_load_lds((cg_gptr)cg_at<float>(wk, 4u * (unsigned)((sidx * a.Cout + c) * a.Cin + q * 4)), (cg_lptr)(st + p * 256), 16, 0, 0);
                        ++cnt;
                    }
                }
            }
        }
        return cnt;
    };

    int cnt = 0;                                       // requests per stage of the current frame chunk (this wave)
#pragma unroll
    for (int sg = 0; sg < NSTG - 1; ++sg)
        if (sg < nk) cnt = issue(sg, 0, min(G::BT, a.T) * V, sg * SBK, tid0);
    int after = 0;                                     // requests this wave made after the chunk's first stage (the previous copy-out's stores)
    float st1 = 0.f, st2 = 0.f;
    for (int t0 = 0; t0 < a.T; t0 += G::BT) {
        const int bt = min(G::BT, a.T - t0);
        const int ncols = bt * V;
        const int nt0 = t0 + G::BT, nbt = nt0 < a.T ? min(G::BT, a.T - nt0) : 0;
        int tid = tid0;
        asm volatile("" : "+v"(tid));                  // see above: nothing derived from it leaves this iteration
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int j = lane & 15, kq = lane >> 4;

        // ---- x3 = W3 x + b3 for the 48 rows: column tiles as in x3_chunk (every SIMD five: wave w < 4 three, wave w + 4 two)
        const int cw0 = 5 * (wave & 3) + (wave < 4 ? 0 : 3);
        const bool third = wave < 4;
        f32x4 acc[NRT][CW];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int bcol[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) { const int col = (cw0 + c) * 16 + j; bcol[c] = (col < ncols && (c < 2 || third)) ? col : 0; }
        for (int kc = 0; kc < nk; ++kc) {
            // this wave's pieces of stage kc have landed: the stages requested after it (and, for the stages requested before the
            // copy-out's stores, those stores) may still be in flight
            cg_wait_vmcnt((kc < NSTG - 1 ? after : 0) + (min(kc + NSTG - 2, nk - 1) - kc) * cnt);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everyone's have; nobody reads stage kc - 1 any more
            if (kc + NSTG - 1 < nk) issue((kc + NSTG - 1) % NSTG, t0, ncols, (kc + NSTG - 1) * SBK, tid);
            const int slot = kc % NSTG;
            const float* As = X3 + slot * F::STG + F::XSL * 4 + j * AP + kq;
            const float* Bs = X3 + slot * F::STG + kq * PB;
#pragma unroll
            for (int k4 = 0; k4 < SBK / 4; ++k4) {
                float av[NRT], bv[CW];
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) av[rt] = As[rt * 16 * AP + k4 * 4];
#pragma unroll
                for (int c = 0; c < CW; ++c) bv[c] = Bs[k4 * 4 * PB + bcol[c]];
                if ((TG_CKO & 1) && k4) continue;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) acc[rt][c] = mfma16(av[rt], bv[c], acc[rt][c]);
                if (third) {
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) acc[rt][2] = mfma16(av[rt], bv[2], acc[rt][2]);
                }
            }
        }
        float b3r[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) b3r[rt][r] = a.b3[rt * a.Cout + c0 + kq * 4 + r];       // row = rt*16 + kq*4 + r = (subset rt, channel kq*4 + r)
        EFrag<G, ST> ef;
        ef.ld(En, a.Cout, c0 + wave, tid);             // the first channel's E rows travel under the tile write
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // stages dead: the x3 tile may be written
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int col = (cw0 + c) * 16 + j;
                if (col >= ncols || (c == 2 && !third)) continue;
                const int fr = col / V, v = col - fr * V;
#pragma unroll
                for (int r = 0; r < 4; ++r) X3[((kq * 4 + r) * G::BT + fr) * (ST * V) + rt * V + v] = acc[rt][c][r] + b3r[rt][r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        aggregate_mfma_g<G, ST, true>(En, a.Cout, c0, ef, X3, Zs, bt, tid);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // ---- copy-out: LDS -> registers, then (everybody done with the region) the next chunk's first stage, then the stores
        const int c = tid / RL, lrow = tid % RL;
        float4 zv[NV4], xv[ST][NV4];
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int p4 = lrow + i * RL;
            zv[i] = p4 < (ncols >> 2) ? reinterpret_cast<const float4*>(Zs + c * G::NCOLS)[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (x3_out) {
#pragma unroll
            for (int s = 0; s < ST; ++s)
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const int p4 = lrow + i * RL;
                    const int fr = p4 / (V / 4), q = p4 - fr * (V / 4);
                    xv[s][i] = p4 < (ncols >> 2) ? *reinterpret_cast<const float4*>(X3 + (c * G::BT + fr) * (ST * V) + s * V + 4 * q)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        if (nbt > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int sg = 0; sg < NSTG - 1; ++sg)      // slots 0 .. NSTG - 2, as the next K loop expects them (the z tile is dead too: barrier above)
                if (sg < nk) cnt = issue(sg, nt0, nbt * V, sg * SBK, tid);
            asm volatile("" ::: "memory");             // the stores below stay below: `after` counts them
            after = nst;
        }
        const unsigned rowb = 4u * (unsigned)(c * a.T * V);
        if (x3_out) {
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                float* xo = x3_out + (((long long)n * ST * a.Cout + s * a.Cout + c0) * a.T + t0) * V;
#pragma unroll
                for (int i = 0; i < NV4; ++i) {
                    const int p4 = lrow + i * RL;
                    if (p4 < (ncols >> 2)) *cg_at_w<float4>(xo, rowb + 16u * (unsigned)p4) = xv[s][i];
                }
            }
        }
        float* ybase = y + (((long long)n * a.Cout + c0) * a.T + t0) * V;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int p4 = lrow + i * RL;
            if (p4 < (ncols >> 2)) {
                *cg_at_w<float4>(ybase, rowb + 16u * (unsigned)p4) = zv[i];
                st1 += (zv[i].x + zv[i].y) + (zv[i].z + zv[i].w);
                st2 = fmaf(zv[i].x, zv[i].x, fmaf(zv[i].y, zv[i].y, fmaf(zv[i].z, zv[i].z, fmaf(zv[i].w, zv[i].w, st2))));
            }
        }
    }
    if (stats_part) {
#pragma unroll
        for (int o = 1; o < RL; o <<= 1) { st1 += __shfl_xor(st1, o); st2 += __shfl_xor(st2, o); }
        const int c = tid0 / RL;
        if (tid0 % RL == 0) {
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
    if ((long long)d->T * d->V >= (1LL << 24)) { tamgcn_set_error("%s: T * V >= 2^24 (32-bit byte offsets inside a 32-row operand chunk)", who); return -1; }
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
    // 16-channel tiles, S = 3: two workgroups per CU without a resident E tile (round 4).  Measured per layer shape (256 clips, us):
    //   resident E (ctrgc_fwd_kernel<.., true>)    158 / 206 / 400 / 297 / 590 / 488   (l1 l2 l5 l6 l8 l9)
    //   E from L2, register-staged operands        151 / 200 / 385 / 277 / 545 / 442
    //   E from L2, LDS-DMA operands (fwd2)           - / 218 / 421 / 287 / 553 / 420
    // TAMGCN_CTRGC_FWD2: 0 resident E everywhere, 1 (default) the faster of the other two by Cin, 2 fwd2 wherever it applies, 3 never fwd2.
    static const int mode = [] { const char* e = getenv("TAMGCN_CTRGC_FWD2"); return e ? atoi(e) : 1; }();
    const bool al16 = (((uintptr_t)d->x.x1 | (uintptr_t)d->w3) & 15) == 0;
    const bool two = ct == 16 && d->S == 3 && mode != 0 && (d->Cin >= 64 || mode >= 2);   // the stem layer (Cin = 3) is faster with E resident inside the step (144 vs 173 us)
    if (two && d->Cin % 16 == 0 && al16 && (mode == 2 || (mode == 1 && d->Cin >= 256))) {
        static tg_devmask fw3g = 0;
        constexpr size_t lds_ = Plan<G20W, 3>::LDS_NOE;
        static_assert(lds_ <= 80 * 1024, "two workgroups per CU");
        // two stages: a third one (it fits, reaching into the z tile while the K loop runs) was measured SLOWER at every shape --
        // 231 / 450 / 316 / 607 / 487 us against 223 / 422 / 292 / 564 / 440 (l2 l5 l6 l8 l9, same box): more requests in flight
        // load the L2 -> LDS path further, they do not hide its latency
        tg_allow_lds((const void*)ctrgc_fwd2_kernel<G20W, 3, 2>, lds_, &fw3g);
        hipLaunchKernelGGL((ctrgc_fwd2_kernel<G20W, 3, 2>), dim3(grid_blocks(a)), dim3(G20W::NT), lds_, (hipStream_t)stream, a, y, stats_part, x3_out);
        tamgcn_note_kernel("ctrgc_fwd2_kernel<Geo<%d, %d, %d, %d, %d>, 3>", G20W::V, G20W::CT, G20W::TB, G20W::NTQ, G20W::SBK);
    } else if (two) {
        static tg_devmask fw3e = 0;
        constexpr size_t lds_ = Plan<G20W, 3>::LDS_NOE;
        tg_allow_lds((const void*)ctrgc_fwd_kernel<G20W, 3, false>, lds_, &fw3e);
        if (getenv("TAMGCN_DEBUG_OCC")) {
            int nb_ = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, (const void*)ctrgc_fwd_kernel<G20W, 3, false>, G20W::NT, lds_);
            fprintf(stderr, "[tamgcn] ctrgc_fwd_kernel<CT 16, 3, E from L2>: %d workgroups per CU (%zu B LDS)\n", nb_, lds_);
        }
        hipLaunchKernelGGL((ctrgc_fwd_kernel<G20W, 3, false>), dim3(grid_blocks(a)), dim3(G20W::NT), lds_, (hipStream_t)stream, a, y, stats_part, x3_out);
        tamgcn_note_kernel("ctrgc_fwd_kernel<Geo<%d, %d, %d, %d, %d>, 3, E from L2>", G20W::V, G20W::CT, G20W::TB, G20W::NTQ, G20W::SBK);
    } else if (ct == 16) {
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
