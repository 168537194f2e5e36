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
#include <stdlib.h>

namespace {

TG_TRACE_DEFINE(tamgcn_trace_read_tconv)
// stamps are summed in registers and leave through ONE batch of atomics at the end of the workgroup (an atomic per stamp
// would sit in the vmcnt queue the kernel's own waits count)
#ifdef TAMGCN_TRACE
#define TC_ACC(slot, expr) tacc[slot] += (unsigned long long)(expr)
#else
#define TC_ACC(slot, expr)
#endif

constexpr int TC_NT = 256;
constexpr int TC_BK = 16;                                 // input channels per LDS chunk
constexpr int TC_NPF = 8;                                 // float4 prefetch slots per thread and source
constexpr int TC_MAXLB = TC_NPF * TC_NT * 4 / TC_BK;      // floats per line-buffer row: 512
constexpr int TC_MAXB = TAMGCN_TCONV_MAXB;
// LDS pitches are compile-time constants: the four k rows of a fragment and the taps' row blocks are then IMMEDIATE offsets
// of the ds_reads (with run-time pitches hipcc kept 24 address registers per tap and 24 v_adds to step them).  528 >= TC_MAXLB
// and == 16 (mod 32): the two k rows a 32-lane half reads sit 16 banks apart.
constexpr int TC_PX = 528;
template <int MT> struct TcPitchW { static constexpr int v = MT == 1 ? 16 : MT == 2 ? 48 : 80; };      // >= 16*MT and == 16 (mod 32)

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
    int BT, TIN, LB, Vs, Vp, nsl, mh, ntiles, nby, ntt, tpw;    // ntiles = ceil(ntt / tpw) * nsl workgroup units per (sample, branch)
};

// x / d for 0 <= x < 2^20 and 4 <= d <= 64 through the reciprocal (rcp = 1.0f / d): three VALU instead of the ~25 of an
// integer division.  (x + 0.5) / d lies at least 0.5 / d >= 0.0078 from an integer, float rounding moves it by < 0.07.
// The kernel's address arithmetic had ~30 divisions per tile and lane: more issue slots than its MFMAs
// (tools/tconv_phases.py: 3.7 k of 13.6 k clocks per tile went into "issuing 13 loads").
__device__ __forceinline__ int tc_div(int x, float rcp) { return (int)(((float)x + 0.5f) * rcp); }

// four consecutive columns col0..col0+3 of one output row: contiguous in HBM (full-width tiles: the row is the flat
// (t, v) run; joint slices: Vs % 4 == 0 keeps the group inside its frame).  Only a full-width tile can end in a partial group.
__device__ __forceinline__ long long tc_off(int V, int Vs, float rVs, int v0, int t0, int col0, bool flat) {
    if (flat) return (long long)t0 * V + col0;
    const int fr = tc_div(col0, rVs);
    return (long long)(t0 + fr) * V + v0 + (col0 - fr * Vs);
}

// The pooled branch (MaxPool2d((3,1), stride, pad 1) of relu(bn(src)), reference models/ctrgcn.py:113-118) as its own
// LDS-free launch: workgroup = (unit of `tpw` frame tiles, sample), 16 lanes per channel row, four consecutive columns per
// lane, the three rows of a window straight from global memory (L1 / L2 serve the re-reads), 12 independent 16-byte loads in
// flight per lane.  Its moment partials use the convolution launch's slots (same units).  (Inside the convolution launch's
// grid a pool workgroup held that launch's 78-119 KB of LDS for nothing: +30 us at 64 channels.)
__global__ __launch_bounds__(TC_NT) void tpool_kernel(const TcArgs a) {
    const int tid = threadIdx.x;
    const int pk = tid >> 4, psub = tid & 15;
    const int unit = blockIdx.x % a.ntiles, n = blockIdx.x / a.ntiles;
    const int grp = unit / a.nsl, sl = unit - grp * a.nsl;
    const int tl0 = grp * a.tpw, tl1 = min(a.ntt, tl0 + a.tpw);
    const int V = a.V, Vs = a.Vs, v0 = sl * Vs;
    const float rVs = 1.0f / (float)Vs;
    const long long cs = (long long)a.T_src * V, ocs = (long long)a.T_out * V;
    const int part = n * a.ntiles + unit;
    const int br = a.nb;
    const int c_lo = tl0 * a.BT * Vs, c_hi = min(a.T_out, tl1 * a.BT) * Vs;      // this unit's columns (frame-major within the slice)
    for (int kc = 0; kc < a.Cb; kc += TC_BK) {
        const int sch = a.src.coff + br * a.Cb + kc + pk;
        const float c1 = a.src.coef ? a.src.coef[sch] : 1.f, c0 = a.src.coef ? a.src.coef[2 * a.src.ctot + sch] : 0.f;
        const float* xrow = a.src.x1 + ((long long)n * a.src.ctot + sch) * cs;
        const int och = a.ycoff + br * a.Cb + kc + pk;
        float* yrow = a.y + ((long long)n * a.yctot + och) * ocs;
        float s1 = 0.f, s2 = 0.f;
        constexpr int PB = 4;                              // groups per batch: 12 independent 16-byte loads in flight per lane
        for (int cb = c_lo + 4 * psub; cb < c_hi; cb += 64 * PB) {
            f32x4 q0[PB], q1[PB], q2[PB];
            bool vec[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {                 // every load unconditional (clamped to the centre row / a valid group)
                const int c = cb + 64 * u;
                const int cc = c < c_hi ? c : c_lo;
                const int fr = tc_div(cc, rVs), v = cc - fr * Vs;
                vec[u] = c < c_hi && v + 4 <= Vs && c + 4 <= c_hi;
                const int vv = v + 4 <= Vs ? v : 0;         // a group that leaves its frame is redone element-wise below
                const int th = fr * a.stride;
                const float* p = xrow + (long long)th * V + v0 + vv;
                q1[u] = *reinterpret_cast<const f32x4*>(p);
                q0[u] = *reinterpret_cast<const f32x4*>(th > 0 ? p - V : p);
                q2[u] = *reinterpret_cast<const f32x4*>(th + 1 < a.T_src ? p + V : p);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int c = cb + 64 * u;
                if (vec[u]) {
                    const int fr = tc_div(c, rVs), v = c - fr * Vs;
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = fmaxf(fmaxf(fmaxf(fmaf(c1, q0[u][r], c0), fmaf(c1, q1[u][r], c0)), fmaf(c1, q2[u][r], c0)), 0.f);
                    *reinterpret_cast<f32x4*>(yrow + (long long)fr * V + v0 + v) = o;
                    s1 += (o[0] + o[1]) + (o[2] + o[3]);
                    s2 = fmaf(o[0], o[0], fmaf(o[1], o[1], fmaf(o[2], o[2], fmaf(o[3], o[3], s2))));
                } else if (c < c_hi) {
                    for (int r = 0; r < 4 && c + r < c_hi; ++r) {
                        const int cc = c + r, f2 = tc_div(cc, rVs), v2 = cc - f2 * Vs, th = f2 * a.stride;
                        const float* p = xrow + (long long)th * V + v0 + v2;
                        float m = fmaf(c1, p[0], c0);
                        if (th > 0) m = fmaxf(m, fmaf(c1, p[-V], c0));
                        if (th + 1 < a.T_src) m = fmaxf(m, fmaf(c1, p[V], c0));
                        m = fmaxf(m, 0.f);
                        yrow[(long long)f2 * V + v0 + v2] = m;
                        s1 += m;
                        s2 = fmaf(m, m, s2);
                    }
                }
            }
        }
        if (a.stats) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (psub == 0) {
                a.stats[((long long)0 * a.stats_ctot + och) * a.nparts + part] = s1;
                a.stats[((long long)1 * a.stats_ctot + och) * a.nparts + part] = s2;
            }
        }
    }
}

// MT = 16-row output tiles of the workgroup = its wave groups: 256*MT threads, wave (wm = wave / 4, wc = wave % 4) owns row
// tile wm and the column tiles wc*CT .. wc*CT + CT-1.  Cb = 16 -> 256 threads (78 KB of LDS, two workgroups per CU),
// Cb = 32 -> 512 (98 KB), Cb = 64 -> 1024 (119 KB): one line buffer serves every output channel of the branch.
template <int MT, int CT, int KT, bool BWD>
__global__ __launch_bounds__(TC_NT * MT, MT == 4 ? 4 : 2) void tconv_kernel(const TcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tc_smem[];
    constexpr int PW = TcPitchW<MT>::v;
    constexpr int XSZ = TC_BK * TC_PX, WSZ = KT * TC_BK * PW;
    float* Xs = tc_smem;                                  // [2][16][TC_PX]   double-buffered line buffer
    float* Ws = Xs + 2 * XSZ;                             // [2][KT*16][PW]   ... and weight image [tap][k][m]
    float* Ss = Ws + 2 * WSZ;                             // [2][4][MT*16]
    float* cf = Ss + 2 * 4 * MT * 16;                     // [3][Cb]
    constexpr int NTH = TC_NT * MT;                       // threads
    constexpr int NPFW = TC_NPF / MT;                     // line-buffer pieces per thread

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wm = tid >> 8;
    const int j0_ = lane & 15, kq0_ = lane >> 4;
    int j = j0_, kq = kq0_;
    // 1-D grid, XCD-aware: consecutive block ids go round-robin over the 8 XCDs, so block id L is workgroup
    // (L % 8) * (G / 8) + L / 8 of the (unit fastest, branch, sample) order: the workgroups of one (sample, branch) -- which
    // share temporal halos -- and the two output halves of a 64-channel branch -- which share their whole source tile --
    // follow each other on ONE XCD's L2 (speed only; any mapping is correct).
    int L = blockIdx.x;
    { const int G = gridDim.x, G8 = G >> 3; if (L < (G8 << 3)) L = (L & 7) * G8 + (L >> 3); }
    // a workgroup owns `tpw` consecutive frame tiles of one (sample, branch [, output half], joint slice): unit = (group, slice)
    const int unit = L % a.ntiles, by = (L / a.ntiles) % a.nby, n = L / (a.ntiles * a.nby);
    const int grp = unit / a.nsl, sl = unit - grp * a.nsl;
    const int tl0 = grp * a.tpw, tl1 = min(a.ntt, tl0 + a.tpw);
    const int V = a.V, Vs = a.Vs, Vp = a.Vp, v0 = sl * Vs;
    const bool flat = a.nsl == 1;
    const float rVs = 1.0f / (float)Vs, rVp = 1.0f / (float)Vp;
    const long long cs = (long long)a.T_src * V;          // channel stride of the source
    const long long ocs = (long long)a.T_out * V;         // ... of the output
    const int part = n * a.ntiles + unit;                 // this workgroup's slot of the moment partials
    int pk = (tid >> 4) & 15, psub = (tid & 15) + 16 * NPFW * wm;   // staging: thread (channel row pk, float4 pieces psub + 16*i)

    const int br = by / a.mh;
    const int m0 = (by - br * a.mh) * (MT * 16);               // first output channel (inside the branch) of this workgroup
    const int mw = m0 + wm * 16;                               // ... of this wave
    const int dil = a.dil[br];
    const int padb = a.pad[br];
    const float* __restrict__ wb = a.w[br];
    const int sch0 = a.src.coff + br * a.Cb;                   // first source channel of the branch

    for (int e = tid; e < a.Cb; e += NTH) {
        const int ch = sch0 + e;
        cf[e] = a.src.coef ? a.src.coef[ch] : 1.f;
        cf[a.Cb + e] = (a.src.coef && a.src.x2) ? a.src.coef[a.src.ctot + ch] : 0.f;
        cf[2 * a.Cb + e] = a.src.coef ? a.src.coef[2 * a.src.ctot + ch] : 0.f;
    }

    // Line-buffer staging: thread (pk, psub) owns the float4 pieces psub + 16*i of channel row pk -- one channel per thread,
    // so the prologue coefficients are three registers per item, the LDS address is affine in i, and 16 consecutive lanes
    // fetch 256 contiguous bytes.  EVERY piece is loaded, unconditionally (a piece outside the source reads the row's
    // first bytes instead and is zeroed on its way into LDS): no branch around a load, so the number of vector-memory
    // operations in flight is a compile-time constant and the waits are counted, never vmcnt(0) behind the stores.
    const int LB4 = a.LB >> 2;
    const long long sbase = ((long long)n * a.src.ctot + sch0) * cs + (long long)pk * cs;
    int p_off[NPFW];
    unsigned okm = 0;                                          // piece i lies inside the source (of the tile being staged)
    auto setup = [&](int tti) {
        const int tin0 = tti * a.BT * a.stride - padb;
        okm = 0;
#pragma unroll
        for (int i = 0; i < NPFW; ++i) {
            const int pos = (psub + 16 * i) << 2;
            const int slot = tc_div(pos, rVp), v = pos - slot * Vp;
            int th = tin0 + slot;
            bool ok = (psub + 16 * i) < LB4 && th >= 0;
            if (a.up == 2) { ok = ok && !(th & 1); th >>= 1; }      // zero-upsampled source of a stride-2 forward (host: up <= 2)
            ok = ok && th < a.T_src;
            if (ok) okm |= 1u << i;
            p_off[i] = ok ? th * V + v0 + v : 0;
        }
    };
    // Weight staging: thread (r = tid / 16, q0 = tid % 16) owns the elements q0 + 16*i of its row(s) of the chunk, in MEMORY
    // order: forward W[m][k][tap] -> row = output channel (one row of 16*KT contiguous floats per 16-row tile), backward
    // -> row = contraction channel (MT*16*KT contiguous floats: the output channels and their taps, taps flipped).
    constexpr int NW = KT;                                     // KT*16*MT*16 elements per chunk over 256*MT threads
    int wq0 = tid & 15;
    const float* const wthr = BWD ? wb + (long long)pk * a.ws_k + (long long)m0 * KT + wq0 + 16 * KT * wm
                                  : wb + (long long)(mw + pk) * a.ws_m + wq0;
    float4 r1[NPFW], r2[BWD ? NPFW : 1];
    float wr[NW];
    const bool has2 = BWD && a.src.x2 != nullptr;
    const int nch = a.Cb / TC_BK;                              // chunks per tile
    const int nitems = (tl1 - tl0) * nch;
    auto prefetch = [&](int kc) {
        const float* x1 = a.src.x1 + sbase + (long long)kc * cs;
#pragma unroll
        for (int i = 0; i < NPFW; ++i) r1[i] = *reinterpret_cast<const float4*>(x1 + p_off[i]);
        if (BWD) {
            const float* x2 = (has2 ? a.src.x2 : a.src.x1) + sbase + (long long)kc * cs;
#pragma unroll
            for (int i = 0; i < NPFW; ++i) r2[i] = *reinterpret_cast<const float4*>(x2 + p_off[i]);
        }
        const float* wk = wthr + (long long)kc * a.ws_k;
#pragma unroll
        for (int i = 0; i < NW; ++i) wr[i] = wk[16 * i];
    };
    auto stage = [&](int buf, int kc) {                         // registers -> LDS image `buf` (prologue, zero padding)
        float* xs = Xs + buf * XSZ + pk * TC_PX + (psub << 2);
        const float c1 = cf[kc + pk], c2 = has2 ? cf[a.Cb + kc + pk] : 0.f, c0 = cf[2 * a.Cb + kc + pk];
#pragma unroll
        for (int i = 0; i < NPFW; ++i) {
            {                                                   // pieces 0..127 of the row: all the A fragments can reach (LB <= 512)
                float4 o;
                if (BWD) {
                    o.x = fmaf(c1, r1[i].x, fmaf(c2, r2[i].x, c0)); o.y = fmaf(c1, r1[i].y, fmaf(c2, r2[i].y, c0));
                    o.z = fmaf(c1, r1[i].z, fmaf(c2, r2[i].z, c0)); o.w = fmaf(c1, r1[i].w, fmaf(c2, r2[i].w, c0));
                } else {
                    o.x = fmaxf(fmaf(c1, r1[i].x, c0), 0.f); o.y = fmaxf(fmaf(c1, r1[i].y, c0), 0.f);
                    o.z = fmaxf(fmaf(c1, r1[i].z, c0), 0.f); o.w = fmaxf(fmaf(c1, r1[i].w, c0), 0.f);
                }
                if (!(okm & (1u << i))) o = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(xs + 64 * i) = o;
            }
        }
        {
            float* ws = Ws + buf * WSZ;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if (!BWD) {
                    const int q = wq0 + 16 * i;                 // (k, tap) in memory order
                    const int kk = q / KT, tap = q - kk * KT;
                    ws[(tap * TC_BK + kk) * PW + wm * 16 + pk] = wr[i];
                } else {
                    const int q = wq0 + 16 * (i + KT * wm);     // (m, flipped tap) in memory order
                    const int mi = q / KT, tap = KT - 1 - (q - mi * KT);
                    ws[(tap * TC_BK + pk) * PW + mi] = wr[i];
                }
            }
        }
    };

    float s1 = 0.f, s2 = 0.f;
    // per-lane epilogue constants, fetched once: a load inside the epilogue puts a vmcnt(0) -- i.e. a wait for the previous
    // column tile's STORES -- in front of every use
    float bia = 0.f, mc1 = 1.f, mc0 = 0.f, ctr = 0.f;
    if (!BWD) { if (a.bias[br]) bia = a.bias[br][mw + j]; }
    else {
        const int hch = a.mask.coff + br * a.Cb + mw + j;
        if (a.mask.coef) { mc1 = a.mask.coef[hch]; mc0 = a.mask.coef[2 * a.mask.ctot + hch]; }
        if (a.center) ctr = a.center[hch];
    }

    // ---- software pipeline over the items (tile, chunk) of this workgroup, two LDS images:
    //   item it:  [loads of item it+1 issued] -> MFMAs on image it&1 -> loads landed: stage item it+1 into image (it+1)&1
    //             -> (last chunk of a tile) epilogue: stores -> ONE barrier.
    // The loads travel under the MFMAs; the stores under the barrier and the next item's MFMAs (nothing waits for them
    // until the next staging, a whole MFMA phase later).
    // No branch encloses a load or a staging pass (the item after the last one re-stages the workgroup's first item into
    // the idle image): two call sites of the prefetch merging in a phi made hipcc COPY the just-requested registers at the
    // join, i.e. wait for the loads in front of the MFMAs they were meant to travel under.
#ifdef TAMGCN_TRACE
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TG_T(tt0);
    setup(tl0);
    prefetch(0);
    __syncthreads();                                           // cf table visible
    stage(0, 0);
    __syncthreads();
    TG_T(tt1); TC_ACC(0, tt1 - tt0);

    f32x4 acc[CT];
    int boff[CT];
    int it = 0;
    for (int tti = tl0; tti < tl1; ++tti) {
        const int t0 = tti * a.BT;
        const int bt = min(a.BT, a.T_out - t0);
        const int ncols = bt * Vs;
        // LDS offsets of this lane's A-fragment columns: column c = (frame fr, joint v) -> fr*stride*Vp + v (tap adds tap*dil*Vp)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = (wave * CT + c) * 16 + j;
            if (col < ncols) { const int fr = tc_div(col, rVs); boff[c] = fr * a.stride * Vp + (col - fr * Vs); }
            else boff[c] = 0;                                  // padding tile: reads in-bounds data, never stored
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // backward: the mask / centring operand h_pre of this tile's outputs, requested ahead of the tile's MFMAs
        f32x4 h[BWD ? CT : 1];
        if (BWD) {
            const int hch = a.mask.coff + br * a.Cb + mw + j;
            const float* hrow = a.mask.x1 + ((long long)n * a.mask.ctot + hch) * ocs;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                int col0 = (wave * CT + c) * 16 + 4 * kq;
                if (col0 + 4 > ncols) col0 = 0;                 // padding / partial group: a safe address, re-read element-wise below
                h[c] = *reinterpret_cast<const f32x4*>(hrow + tc_off(V, Vs, rVs, v0, t0, col0, flat));
            }
        }
        for (int kc = 0; kc < a.Cb; kc += TC_BK, ++it) {
            // hipcc hoists every item-invariant address (weight image, staging rows, epilogue rows) out of these loops and
            // then runs out of registers: opaque copies of the lane coordinates keep that arithmetic inside the loop
            asm volatile("" : "+v"(j), "+v"(kq), "+v"(pk), "+v"(psub), "+v"(wq0));
            const int buf = it & 1;
            const bool lastc = kc + TC_BK >= a.Cb;
            const int kcn = lastc ? 0 : kc + TC_BK;
            TG_T(ta);
            if (lastc) setup(tti + 1 < tl1 ? tti + 1 : tl0);   // integer work only
            prefetch(kcn);
            TG_T(tb); TC_ACC(1, tb - ta);
            {
                const float* xb = Xs + buf * XSZ;
                const float* wbuf = Ws + buf * WSZ;
#pragma unroll 1
                for (int tap = 0; tap < KT; ++tap) {           // not unrolled: one tap's LDS addresses live at a time
                    const float* xt = xb + kq * TC_PX + tap * dil * Vp;
                    const float* wt = wbuf + (tap * TC_BK + kq) * PW + wm * 16 + j;
#pragma unroll
                    for (int k4 = 0; k4 < TC_BK / 4; ++k4) {
                        float av[CT];
                        const float bv = wt[k4 * 4 * PW];
#pragma unroll
                        for (int c = 0; c < CT; ++c) av[c] = xt[k4 * 4 * TC_PX + boff[c]];
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[c] = mfma16(av[c], bv, acc[c]);
                    }
                }
            }
            TG_T(tc); TC_ACC(2, tc - tb);
#ifdef TAMGCN_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            TG_T(tw); TC_ACC(3, tw - tc);
            stage(buf ^ 1, kcn);
            TG_T(td); TC_ACC(4, td - tw);
            if (lastc) {
                // ---- epilogue straight from the accumulators: lane (j, kq) holds, per column tile c, the four consecutive
                // columns (wave*CT + c)*16 + 4*kq + r of output channel mw + j.
                const int mch = br * a.Cb + mw + j;              // channel inside the launch's output / mask slice
                float* yrow = a.y + ((long long)n * a.yctot + a.ycoff + mch) * ocs;
                float q1 = 0.f, q2 = 0.f;
                if (!BWD) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const int col0 = (wave * CT + c) * 16 + 4 * kq;
                        if (col0 < ncols) {
                            f32x4 v = acc[c];
                            v[0] += bia; v[1] += bia; v[2] += bia; v[3] += bia;
                            const long long off = tc_off(V, Vs, rVs, v0, t0, col0, flat);
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
                    // d h_pre = (conv value) where relu(bn(h_pre)) > 0, else 0; moments (sum d, sum d * (h_pre - mean)) for the
                    // entry BatchNorm's backward
                    const int hch = a.mask.coff + mch;
                    const float* hrow = a.mask.x1 + ((long long)n * a.mask.ctot + hch) * ocs;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const int col0 = (wave * CT + c) * 16 + 4 * kq;
                        if (col0 < ncols) {
                            f32x4 v = acc[c];
                            f32x4 hv = h[c];
                            const long long off = tc_off(V, Vs, rVs, v0, t0, col0, flat);
                            if (col0 + 3 >= ncols) for (int r = 0; r < 4 && col0 + r < ncols; ++r) hv[r] = hrow[off + r];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (!(fmaf(mc1, hv[r], mc0) > 0.f)) v[r] = 0.f;
                                if (col0 + r < ncols) { q1 += v[r]; q2 = fmaf(v[r], hv[r] - ctr, q2); }
                            }
                            if (col0 + 3 < ncols) *reinterpret_cast<f32x4*>(yrow + off) = v;
                            else for (int r = 0; r < 4 && col0 + r < ncols; ++r) yrow[off + r] = v[r];
                        }
                    }
                }
                s1 += q1; s2 += q2;
            }
            TG_T(te); TC_ACC(5, te - td);
            __syncthreads();                                   // image (it+1)&1 complete; everyone is done reading image it&1
            TG_T(tf); TC_ACC(6, tf - te); TC_ACC(7, 1);
        }
    }
    TG_T(tz); TC_ACC(8, tz - tt0); TC_ACC(9, 1);
#ifdef TAMGCN_TRACE
    if (threadIdx.x == 0) for (int i = 0; i < 10; ++i) atomicAdd(&tg_trace[i], tacc[i]);
#endif
    if (a.stats) {
        float u1 = s1, u2 = s2;
        u1 += __shfl_xor(u1, 16); u1 += __shfl_xor(u1, 32);
        u2 += __shfl_xor(u2, 16); u2 += __shfl_xor(u2, 32);
        if (kq == 0) {
            Ss[(0 * 4 + wave) * (MT * 16) + wm * 16 + j] = u1;
            Ss[(1 * 4 + wave) * (MT * 16) + wm * 16 + j] = u2;
        }
        __syncthreads();
        if (tid < 2 * MT * 16) {
            const int st = tid / (MT * 16), row = tid - st * (MT * 16);
            const float tot = (Ss[(st * 4 + 0) * (MT * 16) + row] + Ss[(st * 4 + 1) * (MT * 16) + row]) +
                              (Ss[(st * 4 + 2) * (MT * 16) + row] + Ss[(st * 4 + 3) * (MT * 16) + row]);
            a.stats[((long long)st * a.stats_ctot + a.ycoff + br * a.Cb + m0 + row) * a.nparts + part] = tot;
        }
    }
}

struct TcPlan { int BT, TIN, LB, Vs, Vp, nsl, ntt, mt, mh, ct, tpw, ngrp; size_t lds; };

// stride: frame step of the product's source per output frame (the forward's stride; 1 for the data gradient, whose source is
// zero-upsampled); span = (KT-1) * largest dilation
static int tc_plan(int V, int Cb, int KT, int span, int stride, int T_out, TcPlan* p) {
    if (V < 1 || Cb < 16 || (Cb != 16 && Cb != 32 && Cb % 64 != 0) || (KT != 3 && KT != 5) || T_out < 1) return -1;
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
    p->mt = Cb == 16 ? 1 : Cb == 32 ? 2 : 4;                  // 16-row tiles = wave groups of a workgroup
    p->mh = Cb / (16 * p->mt);
    const int tiles = ceil_div(BT * p->Vs, 16);
    p->ct = ceil_div(tiles, 4) <= 3 ? 3 : 5;
    if (ceil_div(tiles, 4) > 5) return -1;
    p->ntt = ceil_div(T_out, BT);
    p->lds = sizeof(float) * (2 * (size_t)TC_BK * TC_PX + 2 * (size_t)KT * TC_BK * (p->mt == 1 ? 16 : p->mt == 2 ? 48 : 80) + 2 * 4 * p->mt * 16 + 3 * (size_t)Cb);
    return 0;
}

template <bool BWD>
static int tc_launch(const TcArgs& a, const TcPlan& p, int KT, dim3 grid, hipStream_t s) {
#define TC_CASE(MT_, CT_, KT_)                                                                                          \
    if (p.mt == MT_ && p.ct == CT_ && KT == KT_) {                                                                      \
        static tg_devmask done = 0;                                                                                     \
        tg_allow_lds((const void*)tconv_kernel<MT_, CT_, KT_, BWD>, 160 * 1024, &done);                                 \
        hipLaunchKernelGGL((tconv_kernel<MT_, CT_, KT_, BWD>), grid, dim3(TC_NT * MT_), p.lds, s, a);                   \
        tamgcn_note_kernel("tconv_kernel<%d, %d, %d, %s>", MT_, CT_, KT_, BWD ? "true" : "false");                         \
        return 0;                                                                                                       \
    }
    TC_CASE(1, 5, 5) TC_CASE(2, 5, 5) TC_CASE(4, 5, 5) TC_CASE(1, 3, 5) TC_CASE(2, 3, 5) TC_CASE(4, 3, 5)
    TC_CASE(1, 5, 3) TC_CASE(2, 5, 3) TC_CASE(4, 5, 3) TC_CASE(1, 3, 3) TC_CASE(2, 3, 3) TC_CASE(4, 3, 3)
#undef TC_CASE
    tamgcn_set_error("tamgcn_tconv: no instantiation mt=%d ct=%d KT=%d", p.mt, p.ct, KT);
    return -1;
}

// frame tiles per workgroup: enough workgroups to fill the chip about six deep (the tile loop hides a tile's load latency under
// the previous tile's MFMAs and stores; too few workgroups leave CUs idle at the tail).  TAMGCN_TC_TPW overrides (A/B runs).
static void tc_split(TcPlan* p, int N, int nby) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("TAMGCN_TC_TPW"); env = e ? atoi(e) : 0; }
    const long long tiles = (long long)N * nby * p->ntt * p->nsl;
    // measured (tools/tconv_bench.py, 256 clips): N-UCLA shapes are fastest with every tile of a (sample, branch) in one
    // workgroup (4 / 2 / 1 tiles), NTU's 30 tiles per sample in groups of 8-10
    int tpw = env > 0 ? env : (int)((tiles + 256) / 512);
    if (tpw < 1) tpw = 1;
    if (tpw > 8 && env <= 0) tpw = 8;
    if (tpw > p->ntt) tpw = p->ntt;
    p->ngrp = ceil_div(p->ntt, tpw);
    p->tpw = ceil_div(p->ntt, p->ngrp);                      // even groups
    p->ngrp = ceil_div(p->ntt, p->tpw);
}

static int tc_span(const int* dil, int nb, int KT) {
    int d = 1;
    for (int b = 0; b < nb; ++b) d = dil[b] > d ? dil[b] : d;
    return (KT - 1) * d;
}

}  // namespace

extern "C" int tamgcn_tconv_supported(int V, int Cb, int KT, int nb, const int* dil, int stride, int T_in) {
    if (nb < 1 || nb > TC_MAXB || !dil || stride < 1 || stride > 2 || T_in < 1) return 0;
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
    tc_split(&p, d->N, d->nb * p.mh);
    return d->N * p.ngrp * p.nsl;
}

static int tc_common_checks(const tamgcn_tconv_desc* d, const char* who, bool need_w = true) {
    TG_CHECK(d && d->src.x1 && d->y, "%s: null pointer", who);
    TG_CHECK(d->N > 0 && d->N <= (1 << 20) && d->T_in > 0 && d->V > 0 && d->Cb > 0 && d->nb >= 1 && d->nb <= TC_MAXB && d->stride >= 1 && d->stride <= 2,
             "%s: bad dims N=%d T_in=%d V=%d Cb=%d nb=%d stride=%d", who, d->N, d->T_in, d->V, d->Cb, d->nb, d->stride);
    for (int b = 0; b < d->nb; ++b) {
        TG_CHECK(!need_w || d->w[b], "%s: branch %d has no weights", who, b);
        TG_CHECK(d->dil[b] >= 1 && ((d->KT - 1) * d->dil[b]) % 2 == 0, "%s: branch %d: (KT-1)*dil must be even (symmetric padding)", who, b);
        TG_CHECK(((uintptr_t)d->w[b] & 3) == 0, "%s: unaligned weights", who);
    }
    TG_CHECK((long long)d->src.ctot * d->T_in * d->V < (1LL << 31) && (!need_w || (long long)d->yctot * d->T_in * d->V < (1LL << 31)),
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
    tc_split(&p, d->N, d->nb * p.mh);
    a.stats = d->stats_part; a.stats_ctot = d->stats_ctot; a.nparts = d->N * p.ngrp * p.nsl;
    a.mask = null_src(); a.center = nullptr;
    a.BT = p.BT; a.TIN = p.TIN; a.LB = p.LB; a.Vs = p.Vs; a.Vp = p.Vp; a.nsl = p.nsl; a.mh = p.mh;
    a.ntiles = p.ngrp * p.nsl; a.nby = d->nb * p.mh; a.ntt = p.ntt; a.tpw = p.tpw;
    dim3 grid((unsigned)(a.ntiles * a.nby * d->N));
    if (d->pool) {                                             // first: short, and it leaves the source's rows in L2 for the convolutions
        hipLaunchKernelGGL(tpool_kernel, dim3((unsigned)(a.ntiles * d->N)), dim3(TC_NT), 0, (hipStream_t)stream, a);
        TG_LAUNCH_CHECK("tamgcn_tconv_fwd (pooled branch)");
    }
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
    tc_split(&p, d->N, d->nb * p.mh);
    a.stats = d->stats_part; a.stats_ctot = d->stats_ctot; a.nparts = d->N * p.ngrp * p.nsl;
    a.mask = make_src(*d->mask); a.center = d->center;
    a.BT = p.BT; a.TIN = p.TIN; a.LB = p.LB; a.Vs = p.Vs; a.Vp = p.Vp; a.nsl = p.nsl; a.mh = p.mh;
    a.ntiles = p.ngrp * p.nsl; a.nby = d->nb * p.mh; a.ntt = p.ntt; a.tpw = p.tpw;
    dim3 grid((unsigned)(a.ntiles * a.nby * d->N));
    if (tc_launch<true>(a, p, d->KT, grid, (hipStream_t)stream)) return -1;
    TG_LAUNCH_CHECK("tamgcn_tconv_bwd");
    return 0;
}

// ===========================================================================
// Weight gradient of the temporal branches (aten::convolution_backward's weight part for every branch's k x 1 convolution,
// reference models/ctrgcn.py:52-69 as used at :101-111):
//   dW_b[m][k][tap] = sum_{n,t,v} gy(n, b*Cb + m, t, v) * act(src)(n, b*Cb + k, t*stride + tap*dil_b - pad_b, v)
// for all branches in ONE launch.  The contraction runs over the columns p = (n, t, v): a workgroup owns a 16- or 32-channel
// block of k, a 16- or 32-channel block of m (64-channel branches: four such blocks) and every tap, loops over its share of
// the (sample, frame tile) items and keeps the KTL x MTL x KT accumulator tiles in registers; per item it stages
//   Xs  the activated source rows of its k block with their temporal halo -- the forward's line buffer [k][slot][Vp], so a
//       strided forward and the taps are LDS offsets -- and
//   Gs  the gradient rows of its m block, two-source BatchNorm-backward prologue applied, [m][column],
// and issues, per group of four columns, MFMAs  D[k][m] += X[k][4 cols + tap shift] . G[4 cols][m]  (v_mfma_f32_16x16x4_f32:
// exact fp32).  Row pitches == 2 (mod 32): the 16 rows x 2 columns a 32-lane half reads hit 32 banks.  The four waves split
// the column groups; their tiles meet in LDS once, at the end, and leave as one partial slab per workgroup (reduced in fp64
// by tamgcn_reduce_*: deterministic, no atomics).
// Replaces wgrad_kernel<5, ...> (register-staged, p-split), the tap-window form of wgrad_glds_kernel and the scalar stride-2
// path for these shapes (profiles/r03_roofline_table_nucla.txt: 21-35 % of their roofs).
// ===========================================================================
namespace {

constexpr int TW_NT = 256;

struct TwArgs {
    SrcDev gy, src;                 // gy (N, gy.ctot, T_out, V): branch b = channels gy.coff + b*Cb; src (N, src.ctot, T_in, V) likewise
    int N, T_in, T_out, V, Cb, nb, stride;
    int dil[TC_MAXB], pad[TC_MAXB];
    float* part;                    // [nsplit][nb][Cb][Cb][KT]
    int nsplit, nblk;               // nblk: (k block, m block) pairs per branch
    int BT, TIN, LB, PX, PG, Vs, Vp, nsl, ntt;
};

template <int KTL, int MTL, int KT>
__global__ __launch_bounds__(TW_NT, 2) void tconv_wgrad_kernel(const TwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tw_smem[];
    const int PX = a.PX, PG = a.PG;
    float* Xs = tw_smem;                                   // [KTL*16][PX]
    float* Gs = Xs + KTL * 16 * PX;                        // [MTL*16][PG]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int sp = blockIdx.x;                             // split: items sp, sp + nsplit, ...
    const int br = blockIdx.y / a.nblk, blk = blockIdx.y - br * a.nblk;
    const int nmb = a.Cb / (MTL * 16);                     // m blocks per branch
    const int kb = blk / nmb, mb = blk - kb * nmb;
    const int k0 = kb * KTL * 16, m0 = mb * MTL * 16;
    const int V = a.V, Vs = a.Vs, Vp = a.Vp;
    const float rVs = 1.0f / (float)Vs, rVp = 1.0f / (float)Vp;
    const int dil = a.dil[br], padb = a.pad[br];
    const long long cs = (long long)a.T_in * V, gcs = (long long)a.T_out * V;
    const int pk = tid >> 4, psub = tid & 15;
    const int LB4 = a.LB >> 2;

    f32x4 acc[KTL][MTL][KT];
#pragma unroll
    for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int tp = 0; tp < KT; ++tp) acc[kt][mt][tp] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nitems = a.N * a.ntt * a.nsl;
    for (int item = sp; item < nitems; item += a.nsplit) {
        const int n = item / (a.ntt * a.nsl), rem = item - n * (a.ntt * a.nsl);
        const int tti = rem / a.nsl, sl = rem - tti * a.nsl;
        const int v0 = sl * Vs;
        const int t0 = tti * a.BT;
        const int bt = min(a.BT, a.T_out - t0);
        const int ncols = bt * Vs;
        const int tin0 = t0 * a.stride - padb;
        const bool flatg = a.nsl == 1;
        __syncthreads();                                   // the previous item's MFMAs are done with Xs / Gs
        // ---- stage the source rows (prologue + ReLU, zero padding), thread (row pk [+16], pieces psub + 16*i).  All loads of a
        // row block are requested before the first is used (unconditional: a piece outside the source re-reads the row's start)
#pragma unroll
        for (int kt = 0; kt < KTL; ++kt) {
            const int sch = a.src.coff + br * a.Cb + k0 + kt * 16 + pk;
            const float c1 = a.src.coef ? a.src.coef[sch] : 1.f, c0 = a.src.coef ? a.src.coef[2 * a.src.ctot + sch] : 0.f;
            const float* xrow = a.src.x1 + ((long long)n * a.src.ctot + sch) * cs;
            float* xs = Xs + (kt * 16 + pk) * PX;
            float4 xr[TC_NPF];
            unsigned okm = 0;
#pragma unroll
            for (int i = 0; i < TC_NPF; ++i) {
                const int pos = (psub + 16 * i) << 2;
                const int slot = tc_div(pos, rVp), v = pos - slot * Vp;
                const int th = tin0 + slot;
                const bool ok = psub + 16 * i < LB4 && th >= 0 && th < a.T_in;
                if (ok) okm |= 1u << i;
                xr[i] = *reinterpret_cast<const float4*>(xrow + (ok ? th * V + v0 + v : 0));
            }
#pragma unroll
            for (int i = 0; i < TC_NPF; ++i) {
                if (psub + 16 * i < LB4) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (okm & (1u << i)) {
                        o.x = fmaf(c1, xr[i].x, c0); o.y = fmaf(c1, xr[i].y, c0); o.z = fmaf(c1, xr[i].z, c0); o.w = fmaf(c1, xr[i].w, c0);
                        if (a.src.act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    }
                    const int pos = (psub + 16 * i) << 2;
                    *reinterpret_cast<float2*>(xs + pos) = make_float2(o.x, o.y);     // rows are 8-byte aligned (pitch == 2 mod 32)
                    *reinterpret_cast<float2*>(xs + pos + 2) = make_float2(o.z, o.w);
                }
            }
        }
        // ---- stage the gradient rows (linear two-source prologue), [m][column]; columns >= ncols are zero
        const int NG4 = (PG - 2) >> 2;
        const float* g2p = a.gy.x2 ? a.gy.x2 : a.gy.x1;
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) {
            const int gch = a.gy.coff + br * a.Cb + m0 + mt * 16 + pk;
            const float c1 = a.gy.coef ? a.gy.coef[gch] : 1.f, c2 = (a.gy.coef && a.gy.x2) ? a.gy.coef[a.gy.ctot + gch] : 0.f,
                        c0 = a.gy.coef ? a.gy.coef[2 * a.gy.ctot + gch] : 0.f;
            const long long gb = ((long long)n * a.gy.ctot + gch) * gcs;
            float* gs = Gs + (mt * 16 + pk) * PG;
            float4 g1[4], g2[4];
            int goff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                  // NG4 <= 64 (host)
                const int col = (psub + 16 * i) << 2;
                const bool full = col + 4 <= ncols;
                const int cc = full ? col : 0;
                const int fr = tc_div(cc, rVs), v = cc - fr * Vs;
                goff[i] = flatg ? t0 * V + cc : (t0 + fr) * V + v0 + v;
                g1[i] = *reinterpret_cast<const float4*>(a.gy.x1 + gb + goff[i]);
                g2[i] = *reinterpret_cast<const float4*>(g2p + gb + goff[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c4 = psub + 16 * i, col = c4 << 2;
                if (c4 < NG4) {
                    float o[4] = {0.f, 0.f, 0.f, 0.f};
                    if (col + 4 <= ncols) {
                        o[0] = fmaf(c1, g1[i].x, fmaf(c2, g2[i].x, c0)); o[1] = fmaf(c1, g1[i].y, fmaf(c2, g2[i].y, c0));
                        o[2] = fmaf(c1, g1[i].z, fmaf(c2, g2[i].z, c0)); o[3] = fmaf(c1, g1[i].w, fmaf(c2, g2[i].w, c0));
                    } else if (col < ncols) {              // the tile's last, partial group (V % 4 != 0): element-wise
                        const int fr = tc_div(col, rVs), v = col - fr * Vs;
                        const long long off = flatg ? (long long)t0 * V + col : (long long)(t0 + fr) * V + v0 + v;
                        for (int r = 0; r < 4 && col + r < ncols; ++r)
                            o[r] = fmaf(c1, a.gy.x1[gb + off + r], fmaf(c2, g2p[gb + off + r], c0));
                    }
                    *reinterpret_cast<float2*>(gs + col) = make_float2(o[0], o[1]);
                    *reinterpret_cast<float2*>(gs + col + 2) = make_float2(o[2], o[3]);
                }
            }
        }
        __syncthreads();
        // ---- MFMAs: this wave's groups of four columns
        const int ngrp = (ncols + 3) >> 2;
        for (int g = wave; g < ngrp; g += 4) {
            const int col = 4 * g + kq;                    // this lane's contraction column (a column >= ncols has a zero in Gs)
            const int cc = col < ncols ? col : 0;
            const int fr = tc_div(cc, rVs);
            const int boff = fr * a.stride * Vp + (cc - fr * Vs);
            float bv[MTL];
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) bv[mt] = Gs[(mt * 16 + j) * PG + col];
#pragma unroll
            for (int tp = 0; tp < KT; ++tp) {
                float av[KTL];
#pragma unroll
                for (int kt = 0; kt < KTL; ++kt) av[kt] = Xs[(kt * 16 + j) * PX + boff + tp * dil * Vp];
#pragma unroll
                for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
                    for (int mt = 0; mt < MTL; ++mt) acc[kt][mt][tp] = mfma16(av[kt], bv[mt], acc[kt][mt][tp]);
            }
        }
    }
    // ---- the four waves' tiles meet in LDS (two waves at a time: the tile set of 32 x 32 x 5 is 20 KB per wave), then leave as
    // this workgroup's slab: lane (j, kq) holds dW[m = m0 + mt*16 + j][k = k0 + kt*16 + 4*kq + r][tap]
    __syncthreads();
    float* Rs = tw_smem;                                   // [2][KTL*MTL*KT*256]
    constexpr int NTILE = KTL * MTL * KT;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if ((wave >> 1) == pass) {
            float* rs = Rs + (wave & 1) * NTILE * 256;
            int ti = 0;
#pragma unroll
            for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                    for (int tp = 0; tp < KT; ++tp, ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* q = rs + ti * 256 + r * 64 + lane;
                            *q = pass == 0 ? acc[kt][mt][tp][r] : *q + acc[kt][mt][tp][r];
                        }
        }
        __syncthreads();
    }
    float* slab = a.part + (((long long)sp * a.nb + br) * a.Cb) * a.Cb * KT;
    for (int e = tid; e < NTILE * 256; e += TW_NT) {
        const int ti = e >> 8, r = (e >> 6) & 3, ln = e & 63;
        const int tp = ti % KT, mt = (ti / KT) % MTL, kt = ti / (KT * MTL);
        const int m = m0 + mt * 16 + (ln & 15), k = k0 + kt * 16 + 4 * (ln >> 4) + r;
        slab[((long long)m * a.Cb + k) * KT + tp] = Rs[e] + Rs[NTILE * 256 + e];
    }
}

struct TwPlan { int BT, TIN, LB, PX, PG, Vs, Vp, nsl, ntt, ktl, mtl, nblk; size_t lds; };

static int tw_plan(int V, int Cb, int KT, int span, int stride, int T_out, TwPlan* p) {
    if (V < 1 || (Cb != 16 && Cb % 32 != 0) || (KT != 3 && KT != 5) || T_out < 1) return -1;
    p->Vs = V; p->nsl = 1;
    if (V > 32) { if (V % 16) return -1; p->Vs = 16; p->nsl = V / 16; }
    p->Vp = (p->Vs + 3) & ~3;
    p->ktl = p->mtl = Cb == 16 ? 1 : 2;
    p->nblk = (Cb / (16 * p->ktl)) * (Cb / (16 * p->mtl));
    // tiles of <= 160 columns whose line buffer stays within 320 floats (two workgroups per CU at 32 channels); shapes whose
    // halo would leave fewer than six frames (stride 2, V = 25) take 512-float line buffers instead
    for (int cap = 320; ; cap = 512) {
        int BT = 160 / p->Vs;
        if (BT < 1) BT = 1;
        if (BT > T_out) BT = T_out;
        for (; BT >= 1; --BT) {
            p->TIN = (BT - 1) * stride + span + 1;
            p->LB = p->TIN * p->Vp;
            if (p->LB <= cap) break;
        }
        if (BT >= 1 && (BT >= 6 || BT == T_out || cap == 512)) { p->BT = BT; break; }
        if (cap == 512) return -1;
    }
    const int ncmax = p->BT * p->Vs;
    p->PX = ((p->LB + 31) & ~31) + 2;                        // == 2 (mod 32)
    p->PG = ((((ncmax + 3) & ~3) + 31) & ~31) + 2;
    if ((p->PG - 2) / 4 > 64) return -1;
    p->ntt = ceil_div(T_out, p->BT);
    const size_t stage = sizeof(float) * ((size_t)p->ktl * 16 * p->PX + (size_t)p->mtl * 16 * p->PG);
    const size_t red = sizeof(float) * 2 * (size_t)p->ktl * p->mtl * KT * 256;
    p->lds = stage > red ? stage : red;
    return p->lds <= 160 * 1024 ? 0 : -1;
}

}  // namespace

/* partial slabs tamgcn_tconv_wgrad writes: choose nsplit <= this, size `part` [nsplit][nb][Cb][Cb][KT] */
extern "C" int tamgcn_tconv_wgrad_max_split(const tamgcn_tconv_desc* d) {
    TwPlan p;
    if (!d || d->nb < 1 || d->nb > TC_MAXB) return -1;
    const int T_out = (d->T_in - 1) / d->stride + 1;
    if (tw_plan(d->V, d->Cb, d->KT, tc_span(d->dil, d->nb, d->KT), d->stride, T_out, &p)) return -1;
    return d->N * p.ntt * p.nsl;
}

/* d->src = the gradient w.r.t. the branches' outputs (two-source prologue, T_out frames), d->mask = the forward's source with its
 * prologue (T_in frames); d->y = part [nsplit][nb][Cb][Cb][KT], d->yctot = nsplit.  Everything else as tamgcn_tconv_bwd. */
extern "C" int tamgcn_tconv_wgrad(const tamgcn_tconv_desc* d, void* stream) {
    if (tc_common_checks(d, "tamgcn_tconv_wgrad", false)) return -1;
    TG_CHECK((long long)d->mask->ctot * d->T_in * d->V < (1LL << 31), "tamgcn_tconv_wgrad: a sample exceeds 2^31 elements");
    TG_CHECK(d->mask && d->mask->x1 && !d->mask->x2, "tamgcn_tconv_wgrad: needs the forward source (single-source prologue) in `mask`");
    TG_CHECK(d->src.coff + d->nb * d->Cb <= d->src.ctot && d->mask->coff + d->nb * d->Cb <= d->mask->ctot, "tamgcn_tconv_wgrad: channel slice out of range");
    const int T_out = (d->T_in - 1) / d->stride + 1;
    TG_CHECK(d->T_out == T_out, "tamgcn_tconv_wgrad: T_out=%d inconsistent with T_in=%d stride=%d", d->T_out, d->T_in, d->stride);
    TwPlan p;
    TG_CHECK(tw_plan(d->V, d->Cb, d->KT, tc_span(d->dil, d->nb, d->KT), d->stride, T_out, &p) == 0,
             "tamgcn_tconv_wgrad: no tiling for V=%d Cb=%d KT=%d stride=%d", d->V, d->Cb, d->KT, d->stride);
    const int nsplit = d->yctot;
    TG_CHECK(nsplit >= 1 && nsplit <= d->N * p.ntt * p.nsl, "tamgcn_tconv_wgrad: nsplit %d outside 1..%d", nsplit, d->N * p.ntt * p.nsl);
    const bool al = ((((uintptr_t)d->src.x1 | (uintptr_t)(d->src.x2 ? d->src.x2 : d->src.x1) | (uintptr_t)d->mask->x1) & 3) == 0);
    TG_CHECK(al, "tamgcn_tconv_wgrad: unaligned operands");
    TwArgs a;
    a.gy = make_src(d->src); a.src = make_src(*d->mask);
    a.N = d->N; a.T_in = d->T_in; a.T_out = T_out; a.V = d->V; a.Cb = d->Cb; a.nb = d->nb; a.stride = d->stride;
    for (int b = 0; b < TC_MAXB; ++b) { a.dil[b] = b < d->nb ? d->dil[b] : 1; a.pad[b] = (d->KT - 1) * a.dil[b] / 2; }
    a.part = d->y; a.nsplit = nsplit; a.nblk = p.nblk;
    a.BT = p.BT; a.TIN = p.TIN; a.LB = p.LB; a.PX = p.PX; a.PG = p.PG; a.Vs = p.Vs; a.Vp = p.Vp; a.nsl = p.nsl; a.ntt = p.ntt;
    dim3 grid((unsigned)nsplit, (unsigned)(d->nb * p.nblk));
#define TW_CASE(KTL_, MTL_, KT_)                                                                                        \
    if (p.ktl == KTL_ && p.mtl == MTL_ && d->KT == KT_) {                                                               \
        static tg_devmask done = 0;                                                                                     \
        tg_allow_lds((const void*)tconv_wgrad_kernel<KTL_, MTL_, KT_>, 160 * 1024, &done);                              \
        hipLaunchKernelGGL((tconv_wgrad_kernel<KTL_, MTL_, KT_>), grid, dim3(TW_NT), p.lds, (hipStream_t)stream, a);    \
        tamgcn_note_kernel("tconv_wgrad_kernel<%d, %d, %d>", KTL_, MTL_, KT_);                                          \
        TG_LAUNCH_CHECK("tamgcn_tconv_wgrad");                                                                          \
        return 0;                                                                                                       \
    }
    TW_CASE(1, 1, 5) TW_CASE(2, 2, 5) TW_CASE(1, 1, 3) TW_CASE(2, 2, 3)
#undef TW_CASE
    tamgcn_set_error("tamgcn_tconv_wgrad: no instantiation");
    return -1;
}
