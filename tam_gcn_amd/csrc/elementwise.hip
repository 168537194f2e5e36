// HBM-bound element-wise epilogues of the CTR-GCN block and the per-channel
// reductions their backward needs.  A group of TPR lanes (16, 32 or 64: about five
// 16-byte steps per lane, all loads independent) streams one (n, c) row of T*V
// contiguous floats; a 256-thread workgroup carries 256/TPR rows and never
// synchronises: row sums are wave shuffles.  Per-channel partial sums go to
// [stat][C][N] slabs (deterministic, finalised in fp64 by bn.hip).
// Reference: models/ctrgcn.py:117 (max-pool), :145-146, :256-261, :283.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int EW_THREADS = 256;

// row geometry of a launch: lanes per row (power of two, 16..256) -> rows per workgroup.  Round 4: long rows get the WHOLE workgroup
// (the four waves walk one row together: 5.0-5.6 against 4.8-5.4 TB/s at HBM-streaming sizes, tools/probes/ew_layout_probe.hip;
// in the product the N-UCLA rows of 320 steps gained nothing); their row sums then meet in LDS.
struct RowGeo { int tpr, rows; };               // rows = N * C
__device__ __forceinline__ bool row_coords(const RowGeo& g, int C, int& c, int& n, int& li) {
    const int rpb = EW_THREADS / g.tpr;
    const int row = blockIdx.x * rpb + threadIdx.x / g.tpr;
    li = threadIdx.x & (g.tpr - 1);
    const bool ok = row < g.rows;
    const int r = ok ? row : 0;                    // idle lanes shadow row 0 (loads only) and take part in the shuffles
    n = r / C; c = r - n * C;
    return ok;
}
// sum over the lanes of a row (every thread of the workgroup calls it; fixed order: lanes by xor-shuffle, then the row's waves in turn)
__device__ __forceinline__ float row_sum(float v, const RowGeo& g) {
    const int w = g.tpr < 64 ? g.tpr : 64;
    for (int o = 1; o < w; o <<= 1) v += __shfl_xor(v, o);
    if (g.tpr > 64) {                               // wave-uniform (kernel argument): the row spans tpr / 64 waves
        __shared__ float red[EW_THREADS / 64];
        const int wave = threadIdx.x >> 6, wpr = g.tpr >> 6, first = wave & ~(wpr - 1);
        __syncthreads();                            // a previous call's reads of red
        if ((threadIdx.x & 63) == 0) red[wave] = v;
        __syncthreads();
        float t = red[first];
        for (int k = 1; k < wpr; ++k) t += red[first + k];
        v = t;
    }
    return v;
}
__device__ __forceinline__ void row_store_sums(const float* vals, int nst, float* part, int C, int N, int c, int n,
                                               const RowGeo& g, int li, bool ok) {
    for (int s = 0; s < nst; ++s) {
        const float v = row_sum(vals[s], g);
        if (li == 0 && ok) part[((long long)s * C + c) * N + n] = v;
    }
}

// float4 view of the fused prologue for one channel row (coefficients are per channel)
struct RowSrc {
    const float* x1; const float* x2; float c1, c2, c0; int act;
};
__device__ __forceinline__ RowSrc row_src(const SrcDev& s, long long base, int ch) {
    RowSrc r;
    r.x1 = s.x1 + base; r.x2 = s.x2 ? s.x2 + base : nullptr;
    r.c1 = s.coef ? s.coef[ch] : 1.f;
    r.c2 = (s.coef && s.x2) ? s.coef[s.ctot + ch] : 0.f;
    r.c0 = s.coef ? s.coef[2 * s.ctot + ch] : 0.f;
    r.act = s.act;
    return r;
}
__device__ __forceinline__ float row_val(const RowSrc& r, int i) {
    float v = fmaf(r.c1, r.x1[i], fmaf(r.c2, r.x2 ? r.x2[i] : 0.f, r.c0));
    return r.act == 1 ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float4 row_val4(const RowSrc& r, int i4) {
    float4 a = reinterpret_cast<const float4*>(r.x1)[i4];
    float4 b = r.x2 ? reinterpret_cast<const float4*>(r.x2)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = fmaf(r.c1, a.x, fmaf(r.c2, b.x, r.c0)); o.y = fmaf(r.c1, a.y, fmaf(r.c2, b.y, r.c0));
    o.z = fmaf(r.c1, a.z, fmaf(r.c2, b.z, r.c0)); o.w = fmaf(r.c1, a.w, fmaf(r.c2, b.w, r.c0));
    if (r.act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    return o;
}
// Every row-wise kernel below walks a row as L / 4 groups of four floats plus a scalar tail of L % 4 elements.  Rows are 16-byte
// aligned only if L % 4 == 0; gfx950 takes dword-aligned 16-byte accesses at full rate (tools/probes/unaligned_probe.hip), and the
// scalar loops NTU's T = 150 / 75 layers (L = 3750 / 1875) used to take ran at under half the streaming rate.

// ---- unit_gcn tail -------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void gcn_tail_fwd_kernel(RowGeo geo, SrcDev y, SrcDev o, SrcDev res, int has_res,
                                                                  int C, int L, float* g) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) L = 0;                     // idle lanes only take part in the shuffles
    const RowSrc ry = row_src(y, ((long long)n * y.ctot + y.coff + c) * L, y.coff + c);
    const RowSrc ro = row_src(o, ((long long)n * o.ctot + o.coff + c) * L, o.coff + c);
    RowSrc rr = ry;
    if (has_res) rr = row_src(res, ((long long)n * res.ctot + res.coff + c) * L, res.coff + c);
    float* gp = g + ((long long)n * C + c) * L;
    {                                      // groups of four floats (rows need only be 4-byte aligned: V = 25), then the row's tail
        for (int i = li; i < (L >> 2); i += geo.tpr) {
            float4 a = row_val4(ry, i), b = row_val4(ro, i);
            float4 r = has_res ? row_val4(rr, i) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v;
            v.x = fmaxf(a.x + tanhf(b.x) + r.x, 0.f); v.y = fmaxf(a.y + tanhf(b.y) + r.y, 0.f);
            v.z = fmaxf(a.z + tanhf(b.z) + r.z, 0.f); v.w = fmaxf(a.w + tanhf(b.w) + r.w, 0.f);
            reinterpret_cast<float4*>(gp)[i] = v;
        }
    }
    {
        for (int i = (L & ~3) + li; i < L; i += geo.tpr) {
            float v = row_val(ry, i) + tanhf(row_val(ro, i));
            if (has_res) v += row_val(rr, i);
            gp[i] = fmaxf(v, 0.f);
        }
    }
}

__global__ __launch_bounds__(EW_THREADS) void gcn_tail_bwd_kernel(RowGeo geo, const float* dg, const float* g, SrcDev o, const float* o_save,
                                                                  int C, int L, int N, float* dsum, float* doz, float* part) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) L = 0;                     // idle lanes only take part in the shuffles
    const long long b = ((long long)n * C + c) * L;
    const long long bo = ((long long)n * o.ctot + o.coff + c) * L;
    const RowSrc ro = row_src(o, bo, o.coff + c);
    float s[2] = {0.f, 0.f};
    const float mu = o_save[o.coff + c];
    {                                      // groups of four floats (rows need only be 4-byte aligned: V = 25), then the row's tail
        for (int i = li; i < (L >> 2); i += geo.tpr) {
            float4 gg = reinterpret_cast<const float4*>(g + b)[i], dd = reinterpret_cast<const float4*>(dg + b)[i];
            float4 ob = row_val4(ro, i), op = reinterpret_cast<const float4*>(o.x1 + bo)[i];
            float4 d, z;
            float t;
            d.x = gg.x > 0.f ? dd.x : 0.f; t = tanhf(ob.x); z.x = d.x * (1.f - t * t);
            d.y = gg.y > 0.f ? dd.y : 0.f; t = tanhf(ob.y); z.y = d.y * (1.f - t * t);
            d.z = gg.z > 0.f ? dd.z : 0.f; t = tanhf(ob.z); z.z = d.z * (1.f - t * t);
            d.w = gg.w > 0.f ? dd.w : 0.f; t = tanhf(ob.w); z.w = d.w * (1.f - t * t);
            reinterpret_cast<float4*>(dsum + b)[i] = d;
            reinterpret_cast<float4*>(doz + b)[i] = z;
            s[0] += (z.x + z.y) + (z.z + z.w);
            s[1] = fmaf(z.x, op.x - mu, fmaf(z.y, op.y - mu, fmaf(z.z, op.z - mu, fmaf(z.w, op.w - mu, s[1]))));
        }
    }
    {
        for (int i = (L & ~3) + li; i < L; i += geo.tpr) {
            float d = g[b + i] > 0.f ? dg[b + i] : 0.f;
            float off = tanhf(row_val(ro, i));
            float dz = d * (1.f - off * off);
            dsum[b + i] = d;
            doz[b + i] = dz;
            s[0] += dz;
            s[1] = fmaf(dz, o.x1[bo + i] - mu, s[1]);
        }
    }
    row_store_sums(s, 2, part, C, N, c, n, geo, li, rowok);
}

__global__ __launch_bounds__(EW_THREADS) void gcn_mid_bwd_kernel(RowGeo geo, const float* dsum, const float* ddiff, const float* y_pre, const float* y_save,
                                                                 const float* r_pre, const float* r_save, int C, int L, int N,
                                                                 float* dyb, float* dres, float* part) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) L = 0;                     // idle lanes only take part in the shuffles
    const long long b = ((long long)n * C + c) * L;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float muy = y_save[c], mur = r_pre ? r_save[c] : 0.f;
    {                                      // groups of four floats (rows need only be 4-byte aligned: V = 25), then the row's tail
        for (int i = li; i < (L >> 2); i += geo.tpr) {
            float4 d = reinterpret_cast<const float4*>(dsum + b)[i], dd = reinterpret_cast<const float4*>(ddiff + b)[i];
            float4 yp = reinterpret_cast<const float4*>(y_pre + b)[i];
            float4 rp = r_pre ? reinterpret_cast<const float4*>(r_pre + b)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 a = make_float4(d.x - dd.x, d.y - dd.y, d.z - dd.z, d.w - dd.w);
            float4 r = make_float4(d.x + dd.x, d.y + dd.y, d.z + dd.z, d.w + dd.w);
            reinterpret_cast<float4*>(dyb + b)[i] = a;
            if (dres) reinterpret_cast<float4*>(dres + b)[i] = r;
            s[0] += (a.x + a.y) + (a.z + a.w);
            s[1] = fmaf(a.x, yp.x - muy, fmaf(a.y, yp.y - muy, fmaf(a.z, yp.z - muy, fmaf(a.w, yp.w - muy, s[1]))));
            if (r_pre) {
                s[2] += (r.x + r.y) + (r.z + r.w);
                s[3] = fmaf(r.x, rp.x - mur, fmaf(r.y, rp.y - mur, fmaf(r.z, rp.z - mur, fmaf(r.w, rp.w - mur, s[3]))));
            }
        }
    }
    {
        for (int i = (L & ~3) + li; i < L; i += geo.tpr) {
            float d = dsum[b + i], dd = ddiff[b + i];
            float a = d - dd, r = d + dd;
            dyb[b + i] = a;
            s[0] += a;
            s[1] = fmaf(a, y_pre[b + i] - muy, s[1]);
            if (dres) dres[b + i] = r;
            if (r_pre) { s[2] += r; s[3] = fmaf(r, r_pre[b + i] - mur, s[3]); }
        }
    }
    row_store_sums(s, r_pre ? 4 : 2, part, C, N, c, n, geo, li, rowok);
}

// ---- max-pool branch ------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void maxpool_fwd_kernel(RowGeo geo, SrcDev src, int C, int T_in, int V, int stride,
                                                                 float* y, int yctot, int ycoff, int T_out, int N, float* part) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) T_out = 0;                     // idle lanes only take part in the shuffles
    const long long bs = ((long long)n * src.ctot + src.coff + c) * T_in * V;
    float* yp = y + ((long long)n * yctot + ycoff + c) * T_out * V;
    float s[2] = {0.f, 0.f};
    for (int i = li; i < T_out * V; i += geo.tpr) {
        int t = i / V, v = i - t * V;
        float best = -INFINITY;
        for (int k = -1; k <= 1; ++k) {
            int th = t * stride + k;
            if (th >= 0 && th < T_in) best = fmaxf(best, src_value(src, bs + (long long)th * V + v, src.coff + c));
        }
        yp[i] = best;
        s[0] += best;
        s[1] = fmaf(best, best, s[1]);
    }
    if (part) row_store_sums(s, 2, part, yctot, N, ycoff + c, n, geo, li, rowok);
}

__global__ __launch_bounds__(EW_THREADS) void maxpool_post_fwd_kernel(RowGeo geo, SrcDev src, int C, int T_in, int V, int stride,
                                                                      float* y, int yctot, int ycoff, int T_out, const float* coef, const float* add, int relu) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) return;
    const long long bs = ((long long)n * src.ctot + src.coff + c) * T_in * V;
    const long long yb = ((long long)n * yctot + ycoff + c) * T_out * V;
    const float c1 = coef[ycoff + c], c0 = coef[2 * yctot + ycoff + c];
    for (int i = li; i < T_out * V; i += geo.tpr) {
        int t = i / V, v = i - t * V;
        float best = -INFINITY;
        for (int k = -1; k <= 1; ++k) {
            int th = t * stride + k;
            if (th >= 0 && th < T_in) best = fmaxf(best, src_value(src, bs + (long long)th * V + v, src.coff + c));
        }
        float o = fmaf(c1, best, c0);
        if (add) o += add[yb + i];
        y[yb + i] = relu ? fmaxf(o, 0.f) : o;
    }
}

// The window logic needs up to nine neighbouring activations per element; read straight from memory that is a
// chain of dependent L2 round trips inside data-dependent control flow (3400 cycles per element measured).  With
// LDS = 1 the row's activations (prologue applied) and its upstream gradients are staged in LDS first -- one
// batch of coalesced loads -- and the window logic runs out of LDS.
template <int LDS>
__global__ __launch_bounds__(EW_THREADS) void maxpool_bwd_kernel(RowGeo geo, SrcDev gy, SrcDev src, const float* src_save, int C, int T_in, int T_out, int V,
                                                                 int stride, float* d, int dctot, int dcoff, int N, float* part) {
    extern __shared__ __attribute__((aligned(16))) float ew_smem[];
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    const int Lin = T_in * V, Lout = T_out * V;
    const int lin = rowok ? Lin : 0;                   // idle lanes only take part in the barrier and the shuffles
    const long long bs = ((long long)n * src.ctot + src.coff + c) * Lin;
    const long long bg = ((long long)n * gy.ctot + gy.coff + c) * Lout;
    float* dp = d + ((long long)n * dctot + dcoff + c) * Lin;
    float* xs = ew_smem + (threadIdx.x / geo.tpr) * (Lin + Lout);
    float* gs = xs + Lin;
    if (LDS) {
        for (int i = li; i < lin; i += geo.tpr) xs[i] = src_value(src, bs + i, src.coff + c);
        for (int i = li; i < (rowok ? Lout : 0); i += geo.tpr) gs[i] = src_value(gy, bg + i, gy.coff + c);
        __syncthreads();
    }
    auto xval = [&](int tt, int v) { return LDS ? xs[tt * V + v] : src_value(src, bs + (long long)tt * V + v, src.coff + c); };
    auto gval = [&](int t, int v) { return LDS ? gs[t * V + v] : src_value(gy, bg + (long long)t * V + v, gy.coff + c); };
    float s[2] = {0.f, 0.f};
    const float mu = src_save[src.coff + c];
    for (int i = li; i < lin; i += geo.tpr) {
        int th = i / V, v = i - th * V;
        float x0 = xval(th, v);
        float grad = 0.f;
        if (x0 > 0.f) {
            // windows t with |t*stride - th| <= 1
            int lo = th - 1; lo = lo < 0 ? 0 : (lo + stride - 1) / stride;
            int hi = (th + 1) / stride; if (hi > T_out - 1) hi = T_out - 1;
            for (int t = lo; t <= hi; ++t) {
                // first arg-max of the window (aten max_pool2d keeps the first maximal index)
                int arg = -1; float best = -INFINITY;
                for (int k = -1; k <= 1; ++k) {
                    int tt = t * stride + k;
                    if (tt < 0 || tt >= T_in) continue;
                    float xv = (tt == th) ? x0 : xval(tt, v);
                    if (xv > best) { best = xv; arg = tt; }
                }
                if (arg == th) grad += gval(t, v);
            }
        }
        dp[i] = grad;
        s[0] += grad;
        s[1] = fmaf(grad, src.x1[bs + i] - mu, s[1]);
    }
    if (part) row_store_sums(s, 2, part, dctot, N, dcoff + c, n, geo, li, rowok);
}

// The same gradient for V % 4 == 0 and stride 1 or 2 without LDS and without a barrier: a lane owns groups of four
// consecutive joints of one frame th and requests, in ONE batch, the five source rows th-2 .. th+2 a decision can depend on
// (an element is the first arg-max of window t only if it beats t's other two candidates) and the gradient rows of the
// windows that contain th (t = th-1, th, th+1 at stride 1; th/2 or (th -+ 1)/2 at stride 2), both operands through their
// prologues; every load is unconditional (rows outside the tensor re-read row th and are ignored).  L1 / L2 serve the
// overlapping rows, HBM sees every tensor once.  (The staged kernel above ran at a quarter of the streaming rate:
// 58-70 us for 84 MB at 256 clips, profiles/r04a_step_breakdown.txt.)
template <int STRIDE>
__global__ __launch_bounds__(EW_THREADS) void maxpool_bwd_vec_kernel(RowGeo geo, SrcDev gy, SrcDev src, const float* src_save, int C, int T_in, int T_out,
                                                                     int V, float* d, int dctot, int dcoff, int N, float* part) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    const int V4 = V >> 2;
    const int ngrp = rowok ? T_in * V4 : 0;
    const long long bs = ((long long)n * src.ctot + src.coff + c) * T_in * V;
    const long long bg = ((long long)n * gy.ctot + gy.coff + c) * T_out * V;
    float* dp = d + ((long long)n * dctot + dcoff + c) * T_in * V;
    const int sch = src.coff + c, gch = gy.coff + c;
    const float sc1 = src.coef ? src.coef[sch] : 1.f, sc0 = src.coef ? src.coef[2 * src.ctot + sch] : 0.f;
    const float gc1 = gy.coef ? gy.coef[gch] : 1.f, gc2 = (gy.coef && gy.x2) ? gy.coef[gy.ctot + gch] : 0.f,
                gc0 = gy.coef ? gy.coef[2 * gy.ctot + gch] : 0.f;
    const float mu = src_save[sch];
    const float* g2p = gy.x2 ? gy.x2 : gy.x1;
    const float rV4 = 1.0f / (float)V4;
    float s[2] = {0.f, 0.f};
    for (int g = li; g < ngrp; g += geo.tpr) {
        const int th = (int)(((float)g + 0.5f) * rV4), v = (g - th * V4) << 2;
        // source rows th-2 .. th+2 (activated), raw centre row for the centred moment
        float4 xr[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int tt = th + k - 2;
            const int tc = (tt >= 0 && tt < T_in) ? tt : th;
            xr[k] = *reinterpret_cast<const float4*>(src.x1 + bs + (long long)tc * V + v);
        }
        // gradient rows of the windows that contain th
        constexpr int NWIN = STRIDE == 1 ? 3 : 2;
        int wt[NWIN];
        if (STRIDE == 1) { wt[0] = th - 1; wt[1] = th; wt[2] = th + 1; }
        else if (th & 1) { wt[0] = (th - 1) >> 1; wt[1] = (th + 1) >> 1; }
        else { wt[0] = th >> 1; wt[1] = -1; }
        float4 g1[NWIN], g2[NWIN];
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            const int tc = (wt[w] >= 0 && wt[w] < T_out) ? wt[w] : 0;
            g1[w] = *reinterpret_cast<const float4*>(gy.x1 + bg + (long long)tc * V + v);
            g2[w] = *reinterpret_cast<const float4*>(g2p + bg + (long long)tc * V + v);
        }
        float xa[5][4];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int tt = th + k - 2;
            const bool ok = tt >= 0 && tt < T_in;
            const float raw[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float val = fmaf(sc1, raw[e], sc0);
                if (src.act == 1) val = fmaxf(val, 0.f);
                xa[k][e] = ok ? val : -INFINITY;                       // outside the tensor: never a candidate
            }
        }
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            const bool wok = wt[w] >= 0 && wt[w] < T_out;
            const int ctr = wt[w] * STRIDE - th + 2;                   // index of the window's centre row in xa (1, 2 or 3)
            const float ga[4] = {g1[w].x, g1[w].y, g1[w].z, g1[w].w}, gb[4] = {g2[w].x, g2[w].y, g2[w].z, g2[w].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = xa[2][e];
                // window t's candidates are rows ctr-1, ctr, ctr+1 of xa in frame order, th is row 2.  aten keeps the FIRST
                // maximum: th wins iff every earlier candidate is strictly smaller and no later one is greater (a candidate
                // outside the tensor is -inf and satisfies both).
                bool win;
                if (ctr == 1) win = xa[0][e] < x0 && xa[1][e] < x0;
                else if (ctr == 2) win = xa[1][e] < x0 && !(xa[3][e] > x0);
                else win = !(xa[3][e] > x0) && !(xa[4][e] > x0);
                if (wok && win && x0 > 0.f) o[e] += fmaf(gc1, ga[e], fmaf(gc2, gb[e], gc0));
            }
        }
        *reinterpret_cast<float4*>(dp + (long long)th * V + v) = make_float4(o[0], o[1], o[2], o[3]);
        const float rawc[4] = {xr[2].x, xr[2].y, xr[2].z, xr[2].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[0] += o[e]; s[1] = fmaf(o[e], rawc[e] - mu, s[1]); }
    }
    if (part) row_store_sums(s, 2, part, dctot, N, dcoff + c, n, geo, li, rowok);
}

// Stride 1, ANY joint count (NTU's V = 25: 6.8 of config 3's 176 ms went through the staged kernel): the same decisions on
// groups of four consecutive FLAT positions p = t*V + v of a (n, c) row of L = T*V floats.  The source rows th-2 .. th+2 of
// an element are the positions p + (k-2)*V and the gradient rows of its windows p + (w-1)*V -- the same shift for all four
// elements of a group -- and "row inside the tensor" is 0 <= position < L: no frame index, no division.  A shifted group that
// sticks out of the row (two per row and shift) is fetched element by element; rows need only be 4-byte aligned.
__global__ __launch_bounds__(EW_THREADS) void maxpool_bwd_flat_kernel(RowGeo geo, SrcDev gy, SrcDev src, const float* src_save, int C, int T, int V,
                                                                      float* d, int dctot, int dcoff, int N, float* part) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    const int L = T * V;
    const int ngrp = rowok ? (L + 3) >> 2 : 0;
    const float* sp = src.x1 + ((long long)n * src.ctot + src.coff + c) * L;
    const float* g1p = gy.x1 + ((long long)n * gy.ctot + gy.coff + c) * L;
    const float* g2p = (gy.x2 ? gy.x2 : gy.x1) + ((long long)n * gy.ctot + gy.coff + c) * L;
    float* dp = d + ((long long)n * dctot + dcoff + c) * L;
    const int sch = src.coff + c, gch = gy.coff + c;
    const float sc1 = src.coef ? src.coef[sch] : 1.f, sc0 = src.coef ? src.coef[2 * src.ctot + sch] : 0.f;
    const float gc1 = gy.coef ? gy.coef[gch] : 1.f, gc2 = (gy.coef && gy.x2) ? gy.coef[gy.ctot + gch] : 0.f,
                gc0 = gy.coef ? gy.coef[2 * gy.ctot + gch] : 0.f;
    const float mu = src_save[sch];
    auto ld4 = [&](const float* base, int start, float (&o)[4]) {
        if (start >= 0 && start + 4 <= L) {
            const float4 t = *reinterpret_cast<const float4*>(base + start);
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (start + e >= 0 && start + e < L) ? base[start + e] : 0.f;
        }
    };
    float s[2] = {0.f, 0.f};
    for (int g = li; g < ngrp; g += geo.tpr) {
        const int p = g << 2;
        float xr[5][4], ga[3][4], gb[3][4];
#pragma unroll
        for (int k = 0; k < 5; ++k) ld4(sp, p + (k - 2) * V, xr[k]);
#pragma unroll
        for (int w = 0; w < 3; ++w) { ld4(g1p, p + (w - 1) * V, ga[w]); ld4(g2p, p + (w - 1) * V, gb[w]); }
        float xa[5][4];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = p + e + (k - 2) * V;
                float val = fmaf(sc1, xr[k][e], sc0);
                if (src.act == 1) val = fmaxf(val, 0.f);
                xa[k][e] = (q >= 0 && q < L) ? val : -INFINITY;        // outside the tensor: never a candidate
            }
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = p + e + (w - 1) * V;                     // the window's centre position
                const float x0 = xa[2][e];
                bool win;                                              // as in maxpool_bwd_vec_kernel<1>: ctr = w + 1
                if (w == 0) win = xa[0][e] < x0 && xa[1][e] < x0;
                else if (w == 1) win = xa[1][e] < x0 && !(xa[3][e] > x0);
                else win = !(xa[3][e] > x0) && !(xa[4][e] > x0);
                if (q >= 0 && q < L && win && x0 > 0.f) o[e] += fmaf(gc1, ga[w][e], fmaf(gc2, gb[w][e], gc0));
            }
        if (p + 4 <= L) *reinterpret_cast<float4*>(dp + p) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (p + e < L) {
                if (p + 4 > L) dp[p + e] = o[e];
                s[0] += o[e];
                s[1] = fmaf(o[e], xr[2][e] - mu, s[1]);
            }
    }
    if (part) row_store_sums(s, 2, part, dctot, N, dcoff + c, n, geo, li, rowok);
}

// ---- residual add (+ReLU) -------------------------------------------------
// rowmean != nullptr: also the mean of every output row (the last block's contribution to the model head's pooling,
// models/ctrgcn.py:343-345, taken while the row is in registers)
__global__ __launch_bounds__(EW_THREADS) void add_act_fwd_kernel(RowGeo geo, SrcDev a, SrcDev res, int has_res, int relu,
                                                                 int C, int L, float* out, float* rowmean) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) L = 0;                     // idle lanes only take part in the shuffles
    const RowSrc ra = row_src(a, ((long long)n * a.ctot + a.coff + c) * L, a.coff + c);
    RowSrc rr = ra;
    if (has_res) rr = row_src(res, ((long long)n * res.ctot + res.coff + c) * L, res.coff + c);
    float* op = out + ((long long)n * C + c) * L;
    const int Lrow = L;
    float acc = 0.f;
    {                                      // groups of four floats (rows need only be 4-byte aligned: V = 25), then the row's tail
        for (int i = li; i < (L >> 2); i += geo.tpr) {
            float4 v = row_val4(ra, i);
            if (has_res) { float4 r = row_val4(rr, i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4*>(op)[i] = v;
            acc += (v.x + v.y) + (v.z + v.w);
        }
    }
    {
        for (int i = (L & ~3) + li; i < L; i += geo.tpr) {
            float v = row_val(ra, i);
            if (has_res) v += row_val(rr, i);
            v = relu ? fmaxf(v, 0.f) : v;
            op[i] = v;
            acc += v;
        }
    }
    if (rowmean) {                          // uniform over the launch: every lane of the wave reaches the shuffles
        acc = row_sum(acc, geo);
        if (li == 0 && rowok) rowmean[(long long)n * C + c] = acc / (float)Lrow;
    }
}

__global__ __launch_bounds__(EW_THREADS) void add_act_bwd_kernel(RowGeo geo, const float* dout, const float* out, int relu,
                                                                 const float* a_pre, const float* a_save,
                                                                 const float* r_pre, const float* r_save,
                                                                 int C, int L, int N, float* dz, float* part) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) L = 0;                     // idle lanes only take part in the shuffles
    const long long b = ((long long)n * C + c) * L;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float mua = a_pre ? a_save[c] : 0.f, mur = r_pre ? r_save[c] : 0.f;
    {                                      // groups of four floats (rows need only be 4-byte aligned: V = 25), then the row's tail
        for (int i = li; i < (L >> 2); i += geo.tpr) {
            float4 d = reinterpret_cast<const float4*>(dout + b)[i];
            if (relu) {
                float4 o = reinterpret_cast<const float4*>(out + b)[i];
                if (!(o.x > 0.f)) d.x = 0.f;
                if (!(o.y > 0.f)) d.y = 0.f;
                if (!(o.z > 0.f)) d.z = 0.f;
                if (!(o.w > 0.f)) d.w = 0.f;
            }
            if (dz) reinterpret_cast<float4*>(dz + b)[i] = d;
            s[0] += (d.x + d.y) + (d.z + d.w);
            if (a_pre) {
                float4 p = reinterpret_cast<const float4*>(a_pre + b)[i];
                s[1] = fmaf(d.x, p.x - mua, fmaf(d.y, p.y - mua, fmaf(d.z, p.z - mua, fmaf(d.w, p.w - mua, s[1]))));
            }
            if (r_pre) {
                float4 p = reinterpret_cast<const float4*>(r_pre + b)[i];
                s[2] += (d.x + d.y) + (d.z + d.w);
                s[3] = fmaf(d.x, p.x - mur, fmaf(d.y, p.y - mur, fmaf(d.z, p.z - mur, fmaf(d.w, p.w - mur, s[3]))));
            }
        }
    }
    {
        for (int i = (L & ~3) + li; i < L; i += geo.tpr) {
            float d = dout[b + i];
            if (relu && !(out[b + i] > 0.f)) d = 0.f;
            if (dz) dz[b + i] = d;
            s[0] += d;
            if (a_pre) s[1] = fmaf(d, a_pre[b + i] - mua, s[1]);
            if (r_pre) { s[2] += d; s[3] = fmaf(d, r_pre[b + i] - mur, s[3]); }
        }
    }
    row_store_sums(s, r_pre ? 4 : 2, part, C, N, c, n, geo, li, rowok);
}

__global__ __launch_bounds__(EW_THREADS) void apply_kernel(RowGeo geo, SrcDev src, int C, int L, float* y, int yctot, int ycoff) {
    int c, n, li;
    const bool rowok = row_coords(geo, C, c, n, li);
    if (!rowok) L = 0;                     // idle lanes only take part in the shuffles
    const long long bs = ((long long)n * src.ctot + src.coff + c) * L;
    float* yp = y + ((long long)n * yctot + ycoff + c) * L;
    for (int i = li; i < L; i += geo.tpr) yp[i] = src_value(src, bs + i, src.coff + c);
}

// out = act(a + res) AND xbar[c][n][v] = mean_t out(n,c,t,v), the next block's pooled joint embedding input (reference
// models/ctrgcn.py:172-174: conv1 / conv2 commute with the mean over T), in the pass that writes `out`: V % 4 == 0, V <= 64.
// One wave per (n, c) row, as four 16-lane groups; lane (sub, l = f * V4 + g) owns the 16-byte joint group g of the frames
// f + F * (sub + 4k) (F = 16 / V4 whole frames per step of a group: 240 contiguous bytes at V = 20), so its partial sum is ONE
// float4; the sums over the F frame phases and then over the four groups are fixed-order shuffle chains: deterministic, graph
// replay = eager bit for bit.
__global__ __launch_bounds__(EW_THREADS) void add_act_fwd_tmean_kernel(SrcDev a, SrcDev res, int has_res, int relu, int N, int C, int T, int V,
                                                                       float* out, float* xbar) {
    const int row = blockIdx.x * (EW_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int sub = lane >> 4, l = lane & 15;
    const int V4 = V >> 2, F = 16 / V4;
    const bool rowok = row < N * C;
    const int r = rowok ? row : 0;
    const int n = r / C, c = r - n * C;
    const int f = l / V4, g = l - f * V4;
    const bool lane_on = rowok && f < F;
    const int L = T * V;
    const RowSrc ra = row_src(a, ((long long)n * a.ctot + a.coff + c) * L, a.coff + c);
    RowSrc rr = ra;
    if (has_res) rr = row_src(res, ((long long)n * res.ctot + res.coff + c) * L, res.coff + c);
    float* op = out + ((long long)n * C + c) * L;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_on) {
        for (int t = f + F * sub; t < T; t += 4 * F) {
            const int i4 = t * V4 + g;
            float4 v = row_val4(ra, i4);
            if (has_res) { const float4 q = row_val4(rr, i4); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4*>(op)[i4] = v;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    // frame phases 1 .. F-1 into phase 0, in order, inside every 16-lane group (all lanes of the wave take part in the shuffles)
    float4 tot = acc;
    for (int ff = 1; ff < F; ++ff) {
        const int srcl = (lane & 48) + ff * V4 + g;
        const float px = __shfl(acc.x, srcl), py = __shfl(acc.y, srcl), pz = __shfl(acc.z, srcl), pw = __shfl(acc.w, srcl);
        tot.x += px; tot.y += py; tot.z += pz; tot.w += pw;
    }
    // ... then the four groups: (0 + 1) + (2 + 3)
    tot.x += __shfl_xor(tot.x, 16); tot.y += __shfl_xor(tot.y, 16); tot.z += __shfl_xor(tot.z, 16); tot.w += __shfl_xor(tot.w, 16);
    tot.x += __shfl_xor(tot.x, 32); tot.y += __shfl_xor(tot.y, 32); tot.z += __shfl_xor(tot.z, 32); tot.w += __shfl_xor(tot.w, 32);
    if (lane_on && f == 0 && sub == 0) {
        const float inv = 1.0f / (float)T;
        *reinterpret_cast<float4*>(xbar + ((long long)c * N + n) * V + 4 * g) = make_float4(tot.x * inv, tot.y * inv, tot.z * inv, tot.w * inv);
    }
}

// xbar[c][n][v] = mean_t value(n,c,t,v)
__global__ __launch_bounds__(EW_THREADS) void tmean_kernel(SrcDev src, int N, int C, int T, int V, float* xbar) {
    const int cpb = EW_THREADS / V;                   // channels per block
    const int cl = threadIdx.x / V, v = threadIdx.x - cl * V;
    const int c = blockIdx.x * cpb + cl, n = blockIdx.y;
    if (cl >= cpb || c >= C) return;
    const long long b = ((long long)n * src.ctot + src.coff + c) * T * V + v;
    float s = 0.f;
    int t = 0;
    for (; t + 8 <= T; t += 8) {                      // eight frames per round trip (the additions keep their order)
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = src_value(src, b + (long long)(t + k) * V, src.coff + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += x[k];
    }
    for (; t < T; ++t) s += src_value(src, b + (long long)t * V, src.coff + c);
    xbar[((long long)c * N + n) * V + v] = s / (float)T;
}

static bool grid_ok(int N, int C) { return N > 0 && C > 0 && (long long)N * C < (1LL << 31); }

// lanes per row: the whole workgroup for rows of >= 448 steps, else about five steps per lane (16..64);
// a step is 16 bytes where the kernel walks the row in groups of four floats
static RowGeo row_geo(int N, int C, int L, bool vec_capable) {
    const int steps = vec_capable ? (L + 3) >> 2 : L;
    int tpr = 16;
    while (tpr < 64 && tpr * 5 < steps) tpr <<= 1;
    static int wide = -1;
    if (wide < 0) { const char* e = getenv("TAMGCN_EW_WIDE"); wide = e ? atoi(e) : 1; }     // 0: at most one wave per row (A/B)
    if (wide && steps >= 448) tpr = 256;            // measured: NTU (1875 / 937 / 468 steps) 168.4 -> 165.9 ms; N-UCLA's 320-step rows no faster
    RowGeo g; g.tpr = tpr; g.rows = N * C;
    return g;
}
static dim3 row_grid(const RowGeo& g) { return dim3((unsigned)ceil_div(g.rows, EW_THREADS / g.tpr)); }

}  // namespace

extern "C" int tamgcn_ew_nparts(int N, int C, int T, int V) { (void)C; (void)T; (void)V; return N; }

extern "C" int tamgcn_gcn_tail_fwd(const tamgcn_src* y, const tamgcn_src* o, const tamgcn_src* res,
                                   int N, int C, int T, int V, float* g, void* stream) {
    TG_CHECK(y && o && g && y->x1 && o->x1 && grid_ok(N, C), "tamgcn_gcn_tail_fwd: bad args");
    const RowGeo geo = row_geo(N, C, T * V, true);
    hipLaunchKernelGGL(gcn_tail_fwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, make_src(*y), make_src(*o), res ? make_src(*res) : null_src(), res ? 1 : 0, C, T * V, g);
    tamgcn_note_kernel("gcn_tail_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_gcn_tail_fwd");
    return 0;
}

extern "C" int tamgcn_gcn_tail_bwd(const float* dg, const float* g, const tamgcn_src* o, const float* o_save,
                                   int N, int C, int T, int V, float* dsum, float* doz, float* part, void* stream) {
    TG_CHECK(dg && g && o && o->x1 && o_save && dsum && doz && part && grid_ok(N, C), "tamgcn_gcn_tail_bwd: bad args");
    const RowGeo geo = row_geo(N, C, T * V, true);
    hipLaunchKernelGGL(gcn_tail_bwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, dg, g, make_src(*o), o_save, C, T * V, N, dsum, doz, part);
    tamgcn_note_kernel("gcn_tail_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_gcn_tail_bwd");
    return 0;
}

extern "C" int tamgcn_gcn_mid_bwd(const float* dsum, const float* ddiff, const float* y_pre, const float* y_save,
                                  const float* r_pre, const float* r_save,
                                  int N, int C, int T, int V, float* dyb, float* dres, float* part, void* stream) {
    TG_CHECK(dsum && ddiff && y_pre && y_save && dyb && part && grid_ok(N, C) && (!r_pre || r_save), "tamgcn_gcn_mid_bwd: bad args");
    const RowGeo geo = row_geo(N, C, T * V, true);
    hipLaunchKernelGGL(gcn_mid_bwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, dsum, ddiff, y_pre, y_save, r_pre, r_save, C, T * V, N, dyb, dres, part);
    tamgcn_note_kernel("gcn_mid_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_gcn_mid_bwd");
    return 0;
}

extern "C" int tamgcn_maxpool_fwd(const tamgcn_src* src, int N, int C, int T_in, int V, int stride,
                                  float* y, int yctot, int ycoff, int T_out, float* stats_part, void* stream) {
    TG_CHECK(src && src->x1 && y && grid_ok(N, C) && stride >= 1, "tamgcn_maxpool_fwd: bad args");
    TG_CHECK(T_out == (T_in + 2 - 3) / stride + 1, "tamgcn_maxpool_fwd: T_out=%d inconsistent with T_in=%d stride=%d", T_out, T_in, stride);
    const RowGeo geo = row_geo(N, C, T_out * V, false);
    hipLaunchKernelGGL(maxpool_fwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, make_src(*src), C, T_in, V, stride, y, yctot, ycoff, T_out, N, stats_part);
    tamgcn_note_kernel("maxpool_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_maxpool_fwd");
    return 0;
}

extern "C" int tamgcn_maxpool_post_fwd(const tamgcn_src* src, int N, int C, int T_in, int V, int stride,
                                       float* y, int yctot, int ycoff, int T_out, const float* coef, const float* add, int relu, void* stream) {
    TG_CHECK(src && src->x1 && y && coef && grid_ok(N, C) && stride >= 1, "tamgcn_maxpool_post_fwd: bad args");
    TG_CHECK(T_out == (T_in + 2 - 3) / stride + 1, "tamgcn_maxpool_post_fwd: T_out=%d inconsistent with T_in=%d stride=%d", T_out, T_in, stride);
    const RowGeo geo = row_geo(N, C, T_out * V, false);
    hipLaunchKernelGGL(maxpool_post_fwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, make_src(*src), C, T_in, V, stride, y, yctot, ycoff, T_out, coef, add, relu);
    tamgcn_note_kernel("maxpool_post_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_maxpool_post_fwd");
    return 0;
}

extern "C" int tamgcn_maxpool_bwd(const tamgcn_src* gy, const tamgcn_src* src, const float* src_save, int N, int C, int T_in, int T_out, int V,
                                  int stride, float* d, int dctot, int dcoff, float* part, void* stream) {
    TG_CHECK(gy && src && gy->x1 && src->x1 && src_save && d && grid_ok(N, C) && stride >= 1, "tamgcn_maxpool_bwd: bad args");
    const RowGeo geo = row_geo(N, C, T_in * V, false);
    const size_t lds = sizeof(float) * (size_t)(EW_THREADS / geo.tpr) * ((size_t)T_in * V + (size_t)T_out * V);
    const bool al16 = ((((uintptr_t)gy->x1 | (uintptr_t)(gy->x2 ? gy->x2 : gy->x1) | (uintptr_t)src->x1 | (uintptr_t)d) & 15) == 0);
    static int vec_env = -1;
    if (vec_env < 0) { const char* e = getenv("TAMGCN_POOL_VEC"); vec_env = e ? atoi(e) : 1; }     // 0: the staged kernel (A/B)
    if (vec_env && (V & 3) == 0 && (stride == 1 || stride == 2) && al16 && !src->x2 && T_out == (T_in - 1) / stride + 1) {
        if (stride == 1)
            hipLaunchKernelGGL(maxpool_bwd_vec_kernel<1>, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                               geo, make_src(*gy), make_src(*src), src_save, C, T_in, T_out, V, d, dctot, dcoff, N, part);
        else
            hipLaunchKernelGGL(maxpool_bwd_vec_kernel<2>, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                               geo, make_src(*gy), make_src(*src), src_save, C, T_in, T_out, V, d, dctot, dcoff, N, part);
        tamgcn_note_kernel("maxpool_bwd_vec_kernel<%d>", stride);
        TG_LAUNCH_CHECK("tamgcn_maxpool_bwd");
        return 0;
    }
    if (vec_env && stride == 1 && !src->x2 && T_out == T_in && (long long)T_in * V < (1LL << 30)) {
        hipLaunchKernelGGL(maxpool_bwd_flat_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           geo, make_src(*gy), make_src(*src), src_save, C, T_in, V, d, dctot, dcoff, N, part);
        tamgcn_note_kernel("maxpool_bwd_flat_kernel");
        TG_LAUNCH_CHECK("tamgcn_maxpool_bwd");
        return 0;
    }
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(maxpool_bwd_kernel<1>, row_grid(geo), dim3(EW_THREADS), lds, (hipStream_t)stream,
                           geo, make_src(*gy), make_src(*src), src_save, C, T_in, T_out, V, stride, d, dctot, dcoff, N, part);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<0>, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           geo, make_src(*gy), make_src(*src), src_save, C, T_in, T_out, V, stride, d, dctot, dcoff, N, part);
    tamgcn_note_kernel("maxpool_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_maxpool_bwd");
    return 0;
}

extern "C" int tamgcn_add_act_fwd(const tamgcn_src* a, const tamgcn_src* res, int relu,
                                  int N, int C, int T, int V, float* out, float* rowmean, float* xbar, void* stream) {
    TG_CHECK(a && a->x1 && out && grid_ok(N, C), "tamgcn_add_act_fwd: bad args");
    if (xbar) {
        TG_CHECK(!rowmean && (V & 3) == 0 && V <= 64, "tamgcn_add_act_fwd: the frame-mean output needs V %% 4 == 0, V <= 64 and no row-mean output");
        const bool al16 = (((uintptr_t)a->x1 | (uintptr_t)(a->x2 ? a->x2 : a->x1) | (uintptr_t)out | (uintptr_t)xbar |
                            (uintptr_t)(res ? res->x1 : a->x1) | (uintptr_t)((res && res->x2) ? res->x2 : a->x1)) & 15) == 0;
        TG_CHECK(al16, "tamgcn_add_act_fwd: the frame-mean form needs 16-byte aligned operands");
        hipLaunchKernelGGL(add_act_fwd_tmean_kernel, dim3((unsigned)ceil_div(N * C, EW_THREADS / 64)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           make_src(*a), res ? make_src(*res) : null_src(), res ? 1 : 0, relu, N, C, T, V, out, xbar);
        tamgcn_note_kernel("add_act_fwd_tmean_kernel");
        TG_LAUNCH_CHECK("tamgcn_add_act_fwd");
        return 0;
    }
    const RowGeo geo = row_geo(N, C, T * V, true);
    hipLaunchKernelGGL(add_act_fwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, make_src(*a), res ? make_src(*res) : null_src(), res ? 1 : 0, relu, C, T * V, out, rowmean);
    tamgcn_note_kernel("add_act_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_add_act_fwd");
    return 0;
}

extern "C" int tamgcn_add_act_bwd(const float* dout, const float* out, int relu, const float* a_pre, const float* a_save,
                                  const float* r_pre, const float* r_save,
                                  int N, int C, int T, int V, float* dz, float* part, void* stream) {
    TG_CHECK(dout && part && grid_ok(N, C) && (!relu || out) && (!a_pre || a_save) && (!r_pre || r_save), "tamgcn_add_act_bwd: bad args");
    const RowGeo geo = row_geo(N, C, T * V, true);
    hipLaunchKernelGGL(add_act_bwd_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, dout, out, relu, a_pre, a_save, r_pre, r_save, C, T * V, N, dz, part);
    tamgcn_note_kernel("add_act_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_add_act_bwd");
    return 0;
}

extern "C" int tamgcn_apply(const tamgcn_src* src, int N, int C, int T, int V, float* y, int yctot, int ycoff, void* stream) {
    TG_CHECK(src && src->x1 && y && grid_ok(N, C), "tamgcn_apply: bad args");
    const RowGeo geo = row_geo(N, C, T * V, false);
    hipLaunchKernelGGL(apply_kernel, row_grid(geo), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       geo, make_src(*src), C, T * V, y, yctot, ycoff);
    tamgcn_note_kernel("apply_kernel");
    TG_LAUNCH_CHECK("tamgcn_apply");
    return 0;
}

extern "C" int tamgcn_tmean(const tamgcn_src* src, int N, int C, int T, int V, float* xbar, void* stream) {
    TG_CHECK(src && src->x1 && xbar && grid_ok(N, C) && V <= EW_THREADS, "tamgcn_tmean: bad args");
    int cpb = EW_THREADS / V;
    hipLaunchKernelGGL(tmean_kernel, dim3(ceil_div(C, cpb), N), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       make_src(*src), N, C, T, V, xbar);
    tamgcn_note_kernel("tmean_kernel");
    TG_LAUNCH_CHECK("tamgcn_tmean");
    return 0;
}
