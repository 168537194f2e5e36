// f2: the eval-mode TCN_GCN_unit for SMALL batches (the inference callers -- ensemble evaluation
// ensemble/ensemble_ctrgcn_resnet_eval.py:147-183, the frozen backbone of models/resnet_gcn_attention.py:82-85,
// visual.py:53-55 -- run model(data) on 1..16 clips).  At batch 1 an activation is 266 KB at every depth (C*T is constant
// down the network) and a block is 65 MFLOP: the training-size kernels (one workgroup per (sample, 64 channels, 16 frames),
// K loops with one DMA round trip per chunk) put 4..16 workgroups on a 256-CU chip and take 12..60 us each, 118 launches per
// forward.  This family cuts a block along its all-channel dependencies instead -- five launches, every one of them
// 50..130 workgroups per sample, operands staged ONCE per workgroup (an activation tile is <= 86 KB: it fits LDS whole) and
// no K-loop pipeline to fill:
//
//   f2_e     (n, subset, 16 channels)   xbar = mean_t x; p, q = W1/W2 xbar + b; D = tanh(p_u - q_v); E = alpha (W4 D + b4) + A
//                                       (models/ctrgcn.py:172-175 with conv1/conv2 commuted with the mean)
//   f2_gcn   (n, 8 channels, 4 frames)  x3 = W3 x + b3 for the tile's 24 rows and down(x) for its 8 (one 32-row MFMA product),
//                                       z = sum_s E_s x3_s, y = bn(z); writes y + res and res - y   (:176, :252-257)
//   f2_gemm  (n, 16 rows, 4 frames)     mode 0: g = relu(y + res + tanh(bn(Wo diff)))               (:219-223, :258-261)
//                                       mode 1: h = relu(bn(W_in g)) rows, plain-branch rows unclamped (:95-99, :121-124)
//   f2_tcn   (n, 16 channels, 4 frames) temporal branches (k x 1, dilated, strided), pooled branch, plain branch, residual
//                                       (identity / 1x1 strided conv), ReLU                           (:101-119, :145-146, :281-283)
//
// BatchNorm is folded into the weights by the caller (eval mode: a per-channel affine of the running statistics).
// All GEMMs: v_mfma_f32_16x16x4_f32 (exact fp32), A rows straight from global memory (L2-resident weights, 16 bytes per lane
// and 16-k block, prefetched one block ahead), B = the LDS tile, the K blocks of a product dealt round-robin to the waves and
// the partial tiles summed through LDS in a fixed order (deterministic).  V = 20 only.
#include "common.h"

namespace {

constexpr int F2_NT = 256, F2_V = 20, F2_VV = 400, F2_BT = 4, F2_NCT = 5, F2_PB = 84;   // 4 frames x 20 joints = 80 columns = 5 MFMA tiles
// pitch 84: the four k rows (4*kq + i) a fragment read touches sit 4*84 = 16 (mod 64) banks apart: conflict-free

typedef __attribute__((address_space(1))) const void* f2_gptr;
typedef __attribute__((address_space(3))) void* f2_lptr;

// A fragment of one 16-k block: lane (j, kq) holds A[row j][k0 + 4*kq + i], i = 0..3 (MFMA step i of the block contracts
// k = k0 + 4*kq + i on BOTH operands: any bijection between (step, kq) and k is a valid K order)
__device__ __forceinline__ void f2_load_a(const float* arow, int K, int k0, int kq, bool vec, float (&a)[4]) {
    const int k = k0 + 4 * kq;
    if (vec) {
        const float4 t = arow ? *reinterpret_cast<const float4*>(arow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (arow && k + i < K) ? arow[k + i] : 0.f;
    }
}

// acc[ct] += A[16 rows][k blocks kb0, kb0 + kbstep, ...] * B, B value for this lane's column of tile ct at row k = bf(k, ct).
// A fragments are fetched four blocks at a time, one group ahead: with one block in flight the loop was one L2 round trip
// (~0.8 us) per 20 MFMAs (~0.1 us).
constexpr int F2_G = 4;
__device__ __forceinline__ void f2_load_group(const float* arow, int K, bool vec, int kb, int kbstep, int kq, float (&dst)[F2_G][4]) {
    const int nkb = (K + 15) >> 4;
#pragma unroll
    for (int g = 0; g < F2_G; ++g) {
        const int b = kb + g * kbstep;
        if (b < nkb) f2_load_a(arow, K, b * 16, kq, vec, dst[g]);
        else { dst[g][0] = dst[g][1] = dst[g][2] = dst[g][3] = 0.f; }
    }
}

// a: the first group (blocks kb0, kb0 + kbstep, ...), already requested by the caller -- before whatever B waits for
template <int NCT, class BF>
__device__ __forceinline__ void f2_gemm16_pre(f32x4 (&acc)[NCT], float (&a)[F2_G][4], const float* arow, int K, bool vec, int kb0, int kbstep, int kq, BF bf) {
    constexpr int G = F2_G;
    const int nkb = (K + 15) >> 4;
    float an[G][4];
    for (int kb = kb0; kb < nkb; kb += G * kbstep) {
        f2_load_group(arow, K, vec, kb + G * kbstep, kbstep, kq, an);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int b = kb + g * kbstep;
            if (b < nkb) {                                         // wave-uniform: a whole block of 4 x NCT MFMAs
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = b * 16 + 4 * kq + i;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma16(a[g][i], bf(k, ct), acc[ct]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[g][i] = an[g][i];
    }
}

template <int NCT, class BF>
__device__ __forceinline__ void f2_gemm16(f32x4 (&acc)[NCT], const float* arow, int K, bool vec, int kb0, int kbstep, int kq, BF bf) {
    float a[F2_G][4];
    f2_load_group(arow, K, vec, kb0, kbstep, kq, a);
    f2_gemm16_pre<NCT>(acc, a, arow, K, vec, kb0, kbstep, kq, bf);
}

// rows [0, K) of a (rows, T, V) activation, frames t0 + tl*fstep (tl < bt), as an LDS tile [Kp][F2_PB]; everything else zero
// (a zero A element times an uninitialised LDS word could be NaN)
__device__ __forceinline__ void f2_stage(float* Bs, const float* src, long long rs, int K, int Kp, int bt, int fstep, int tid) {
    constexpr int U = 8;                                           // loads in flight per thread (one L2 round trip per batch)
    for (int e0 = tid; e0 < Kp * 20; e0 += U * F2_NT) {
        float4 t[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int e = e0 + i * F2_NT;
            const int k = e / 20, c4 = (e - k * 20) * 4;
            const int tl = c4 / F2_V, v = c4 - tl * F2_V;
            t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Kp * 20 && k < K && tl < bt) t[i] = *reinterpret_cast<const float4*>(src + k * rs + (long long)tl * fstep * F2_V + v);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int e = e0 + i * F2_NT;
            const int k = e / 20, c4 = (e - k * 20) * 4;
            if (e < Kp * 20) *reinterpret_cast<float4*>(Bs + k * F2_PB + c4) = t[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct F2GcnArgs {
    int N, Cin, Cout, T, S, R, res_mode;
    const float *x, *w12, *b12, *w4, *b4, *A, *alpha, *w3, *b3, *sy, *ty, *wd, *bd, *xpart;
    float *E, *sum, *diff;
    int vec12, vec4, vec3, vecd;
};

// ---- E for 16 channels of one (sample, subset)
constexpr int F2_PX = 36, F2_PD = 404;     // xbar / pq pitch (20 columns in two tiles), D pitch (4*404 = 16 mod 64)

__global__ __launch_bounds__(F2_NT) void f2_e_kernel(const F2GcnArgs a) {
    constexpr int V = F2_V, VV = F2_VV, NT = F2_NT, PX = F2_PX, PD = F2_PD, NTP = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = (a.Cin + 15) & ~15, R2 = 2 * a.R, R2p = R2 < 16 ? 16 : R2, Rp = (a.R + 15) & ~15;
    float* XB = smem;                    // [Kp][PX]   xbar, columns >= 20 zero
    float* PQ = XB + Kp * PX;            // [R2p][PX]  p rows 0..R-1, q rows R..2R-1
    float* Pp = PQ + R2p * PX;           // [64][PX]   partial pq tiles
    float* Ds = Pp + 64 * PX;            // [Rp][PD]   D; before that: [NTP][Kp][V] partial frame sums
    const int nct = a.Cout / 16;
    const int s = blockIdx.x / nct, c0 = (blockIdx.x - s * nct) * 16, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const long long TV = (long long)a.T * V;
    // Everything that does not depend on x is requested now and used after the xbar phase: the kernel is a chain of dependent
    // round trips on 12..48 workgroups, not a bandwidth problem
    const int nrt = R2p / 16, nparts = 4 / nrt;                // pq product: 1 x 4, 2 x 2 or 4 x 1 (row tiles x K parts) over the waves
    const int prt = wave % nrt, ppart = wave / nrt;
    const float* a12 = prt * 16 + j < R2 ? a.w12 + ((long long)s * R2 + prt * 16 + j) * a.Cin : nullptr;
    float a12g[F2_G][4];
    f2_load_group(a12, a.Cin, a.vec12 != 0, ppart, nparts, kq, a12g);
    float b12r[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) b12r[i] = tid + i * NT < R2 * V ? a.b12[s * R2 + (tid + i * NT) / V] : 0.f;
    const float alpha = a.alpha[0];
    float aw[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, b4r[4], Avr[7];
    {
        const float* arow = a.w4 + ((long long)s * a.Cout + c0 + j) * a.R;
        f2_load_a(arow, a.R, 0, kq, a.vec4 != 0, aw[0]);
        if (a.R > 16) f2_load_a(arow, a.R, 16, kq, a.vec4 != 0, aw[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) b4r[r] = a.b4[s * a.Cout + c0 + kq * 4 + r];
#pragma unroll
        for (int i = 0; i < 7; ++i) Avr[i] = wave + 4 * i < VV / 16 ? a.A[s * VV + (wave + 4 * i) * 16 + j] : 0.f;
    }
    // 1. xbar.  From the producer's per-tile column sums when it left them (f2_tcn: [N][tiles][Cin][V]; reading all of x
    // again in each of these workgroups was 10-20 us of dependent round trips), else from x: four frame phases in parallel.
    // Either way summed in a fixed order.
    {
        float* XP = Ds;
        const float inv = 1.f / (float)a.T;
        if (a.xpart) {
            // (tile group g, channel, joint quad): NTP groups of tpg tiles each; eight such sums advance together, one tile (= one
            // load each, eight in flight) per step
            const int ntt = (a.T + F2_BT - 1) / F2_BT, tpg = (ntt + NTP - 1) / NTP;
            const float* xp = a.xpart + (long long)n * ntt * a.Cin * V;
            const int cnt = NTP * a.Cin * 5;
            for (int e0 = tid; e0 < cnt; e0 += 8 * NT) {
                float4 acc[8];
                int off[8], g8[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int e = e0 + p * NT;
                    const int g = e / (a.Cin * 5), rem = e - g * a.Cin * 5;
                    const int ci = rem / 5, v4 = (rem - ci * 5) * 4;
                    acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                    off[p] = ci * V + v4;
                    g8[p] = e < cnt ? g : NTP;                      // NTP * tpg >= ntt: an out-of-range pair never loads
                }
                for (int i = 0; i < tpg; ++i) {
                    float4 q[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const int tile = g8[p] * tpg + i;
                        q[p] = tile < ntt ? *reinterpret_cast<const float4*>(xp + (long long)tile * a.Cin * V + off[p]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int p = 0; p < 8; ++p) { acc[p].x += q[p].x; acc[p].y += q[p].y; acc[p].z += q[p].z; acc[p].w += q[p].w; }
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int e = e0 + p * NT;
                    if (e < cnt) *reinterpret_cast<float4*>(XP + (g8[p] * Kp) * V + off[p]) = acc[p];
                }
            }
        } else {
            const float* xb = a.x + (long long)n * a.Cin * TV;
            for (int e = tid; e < NTP * a.Cin * 5; e += NT) {
                const int tp = e / (a.Cin * 5), rem = e - tp * a.Cin * 5;
                const int ci = rem / 5, v4 = (rem - ci * 5) * 4;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* p = xb + ci * TV + v4;
                for (int t = tp; t < a.T; t += 8 * NTP) {              // eight loads in flight, summed in frame order
                    float4 q[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        q[i] = t + i * NTP < a.T ? *reinterpret_cast<const float4*>(p + (long long)(t + i * NTP) * V) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { acc.x += q[i].x; acc.y += q[i].y; acc.z += q[i].z; acc.w += q[i].w; }
                }
                *reinterpret_cast<float4*>(XP + (tp * Kp + ci) * V + v4) = acc;
            }
        }
        __syncthreads();
        {
            for (int e = tid; e < Kp * (PX / 4); e += NT) {
                const int ci = e / (PX / 4), v4 = (e - ci * (PX / 4)) * 4;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < a.Cin && v4 < V) {
#pragma unroll
                    for (int tp = 0; tp < NTP; ++tp) {
                        const float4 q = *reinterpret_cast<const float4*>(XP + (tp * Kp + ci) * V + v4);
                        o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                    }
                    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
                }
                *reinterpret_cast<float4*>(XB + ci * PX + v4) = o;
            }
        }
        __syncthreads();
    }
    // 2. p, q = W12_s xbar + b12_s: (2R x Cin) x (Cin x 20)
    {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f2_gemm16_pre<2>(acc, a12g, a12, a.Cin, a.vec12 != 0, ppart, nparts, kq, [&](int k, int ct) { return XB[k * PX + ct * 16 + j]; });
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pp[((ppart * nrt + prt) * 16 + kq * 4 + r) * PX + ct * 16 + j] = acc[ct][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e = tid + i * NT;
            if (e < R2 * V) {
                const int row2 = e / V, v = e - row2 * V;
                const int rt2 = row2 >> 4, rr = row2 & 15;
                float t = b12r[i];
                for (int p = 0; p < nparts; ++p) t += Pp[((p * nrt + rt2) * 16 + rr) * PX + v];
                PQ[row2 * PX + v] = t;
            }
        }
        __syncthreads();
    }
    // 3. D[r][u*V + v] = tanh(p[r][u] - q[r][v]); rows R..Rp zero
    for (int e = tid; e < Rp * VV; e += NT) {
        const int r = e / VV, uv = e - r * VV;
        const int u = uv / V, v = uv - u * V;
        Ds[r * PD + uv] = r < a.R ? fast_tanh(PQ[r * PX + u] - PQ[(a.R + r) * PX + v]) : 0.f;
    }
    __syncthreads();
    // 4. E tile = alpha (W4 D + b4) + A: 16 channels x 400, 25 column tiles over the four waves, K = R <= 32
    {
        float* Eg = a.E + (((long long)n * a.S + s) * a.Cout + c0) * VV;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int ct = wave + 4 * it;
            if (ct < VV / 16) {
                const int col = ct * 16 + j;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = mfma16(aw[0][i], Ds[(4 * kq + i) * PD + col], acc);
                if (a.R > 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = mfma16(aw[1][i], Ds[(16 + 4 * kq + i) * PD + col], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Eg[(long long)(kq * 4 + r) * VV + col] = alpha * (acc[r] + b4r[r]) + Avr[it];
            }
        }
    }
}

// ---- x3 GEMM + aggregation + BatchNorm + residual for 8 channels x 4 frames
constexpr int F2_CT = 8, F2_ES = 3328;     // E run of one subset: 8 * 400 floats = 12.5 DMA pieces of 1 KB, padded to 13

__global__ __launch_bounds__(F2_NT) void f2_gcn_kernel(const F2GcnArgs a) {
    constexpr int V = F2_V, VV = F2_VV, NT = F2_NT, PB = F2_PB, CT = F2_CT, ES = F2_ES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = (a.Cin + 15) & ~15;
    float* Xs = smem;                    // [Kp][PB]      x tile
    float* X3 = Xs + Kp * PB;            // [2][32][PB]   partial products: rows s*8 + c (s < 3), 24 + c = down(x)
    float* Es = X3 + 2 * 32 * PB;        // [3][ES]       E_s[c][u][v]
    const int nct = a.Cout / CT;
    const int ctile = blockIdx.x % nct, tt = blockIdx.x / nct, n = blockIdx.y;
    const int c0 = ctile * CT, t0 = tt * F2_BT, bt = min(F2_BT, a.T - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const long long TV = (long long)a.T * V;
    // E tiles travel by LDS-DMA under the x tile's staging and the GEMM (the barrier below drains them)
    for (int p = wave; p < 3 * 13; p += 4) {
        const int s = p / 13, q = p - s * 13;
        int f = q * 256 + lane * 4;
        if (f > CT * VV - 4) f = CT * VV - 4;                     // lanes past the run re-read its last slot into the padding
        const float* g = a.E + (((long long)n * a.S + s) * a.Cout + c0) * VV + f;
        __builtin_amdgcn_global_load_lds((f2_gptr)g, (f2_lptr)(Es + s * ES + q * 256), 16, 0, 0);
    }
    const int rt = wave & 1, kh = wave >> 1;
    const float* arow;
    bool vec;
    {
        const int row = rt * 16 + j, sidx = row >> 3, c = row & 7;
        arow = sidx < 3 ? a.w3 + ((long long)sidx * a.Cout + c0 + c) * a.Cin
                        : (a.res_mode == 2 ? a.wd + (long long)(c0 + c) * a.Cin : nullptr);
        vec = sidx < 3 ? a.vec3 != 0 : a.vecd != 0;               // (uniform per 8 lanes, both paths are branch-safe)
    }
    float ag[F2_G][4];
    f2_load_group(arow, a.Cin, vec, kh, 2, kq, ag);                // weights travel under the staging of the tile
    f2_stage(Xs, a.x + (long long)n * a.Cin * TV + (long long)t0 * V, TV, a.Cin, Kp, bt, 1, tid);
    __syncthreads();
    {
        f32x4 acc[F2_NCT];
#pragma unroll
        for (int ct = 0; ct < F2_NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f2_gemm16_pre<F2_NCT>(acc, ag, arow, a.Cin, vec, kh, 2, kq, [&](int k, int ct) { return Xs[k * PB + ct * 16 + j]; });
#pragma unroll
        for (int ct = 0; ct < F2_NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) X3[((kh * 32) + rt * 16 + kq * 4 + r) * PB + ct * 16 + j] = acc[ct][r];
    }
    __syncthreads();
    // aggregation: thread = (channel c, frame t, joint group ug): u = ug, ug + 8, ug + 16
    {
        const int c = tid >> 5, t = (tid >> 3) & 3, ug = tid & 7;
        float x3v[3 * V];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float b = a.b3[s * a.Cout + c0 + c];
#pragma unroll
            for (int v4 = 0; v4 < V; v4 += 4) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(X3 + (s * 8 + c) * PB + t * V + v4);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(X3 + (32 + s * 8 + c) * PB + t * V + v4);
#pragma unroll
                for (int i = 0; i < 4; ++i) x3v[s * V + v4 + i] = p0[i] + p1[i] + b;
            }
        }
        const float sy = a.sy[c0 + c], ty = a.ty[c0 + c];
        const float bd = a.res_mode == 2 ? a.bd[c0 + c] : 0.f;
        if (t < bt) {
            for (int u = ug; u < V; u += 8) {
                float z = 0.f;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float* er = Es + s * ES + (c * V + u) * V;
#pragma unroll
                    for (int v4 = 0; v4 < V; v4 += 4) {
                        const f32x4 e = *reinterpret_cast<const f32x4*>(er + v4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) z = fmaf(e[i], x3v[s * V + v4 + i], z);
                    }
                }
                const float y = fmaf(sy, z, ty);
                float res = 0.f;
                const int col = t * V + u;
                if (a.res_mode == 1) res = Xs[(c0 + c) * PB + col];
                else if (a.res_mode == 2) res = X3[(24 + c) * PB + col] + X3[(32 + 24 + c) * PB + col] + bd;
                const long long o = ((long long)n * a.Cout + c0 + c) * TV + (long long)(t0 + t) * V + u;
                a.sum[o] = y + res;
                a.diff[o] = res - y;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16 rows x 4 frames of a pointwise product with the block's epilogues
// ---------------------------------------------------------------------------------------------------------------------
struct F2GemmArgs {
    int N, K, M, T, mode, relu_rows, vec;
    const float *x, *w, *b, *add;
    float* out;
};

__global__ __launch_bounds__(F2_NT) void f2_gemm_kernel(const F2GemmArgs a) {
    constexpr int V = F2_V, NT = F2_NT, PB = F2_PB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = (a.K + 15) & ~15;
    float* Bs = smem;                    // [Kp][PB]
    float* Pp = Bs + Kp * PB;            // [4][16][PB]
    const int nmt = a.M / 16;
    const int mtile = blockIdx.x % nmt, tt = blockIdx.x / nmt, n = blockIdx.y;
    const int m0 = mtile * 16, t0 = tt * F2_BT, bt = min(F2_BT, a.T - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const long long TV = (long long)a.T * V;
    const float* arow = a.w + (long long)(m0 + j) * a.K;
    float ag[F2_G][4];
    f2_load_group(arow, a.K, a.vec != 0, wave, 4, kq, ag);         // weights travel under the staging of the tile
    f2_stage(Bs, a.x + (long long)n * a.K * TV + (long long)t0 * V, TV, a.K, Kp, bt, 1, tid);
    __syncthreads();
    {
        f32x4 acc[F2_NCT];
#pragma unroll
        for (int ct = 0; ct < F2_NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f2_gemm16_pre<F2_NCT>(acc, ag, arow, a.K, a.vec != 0, wave, 4, kq, [&](int k, int ct) { return Bs[k * PB + ct * 16 + j]; });
#pragma unroll
        for (int ct = 0; ct < F2_NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pp[(wave * 16 + kq * 4 + r) * PB + ct * 16 + j] = acc[ct][r];
    }
    __syncthreads();
    for (int e = tid; e < 16 * 20; e += NT) {
        const int row = e / 20, c4 = (e - row * 20) * 4;
        if (c4 >= bt * V) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(Pp + row * PB + c4);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(Pp + (w * 16 + row) * PB + c4);
            v += p;
        }
        const float b = a.b[m0 + row];
        const long long o = ((long long)n * a.M + m0 + row) * TV + (long long)t0 * V + c4;
        float4 r;
        if (a.mode == 0) {
            const float4 ad = *reinterpret_cast<const float4*>(a.add + o);
            r.x = fmaxf(ad.x + fast_tanh(v[0] + b), 0.f); r.y = fmaxf(ad.y + fast_tanh(v[1] + b), 0.f);
            r.z = fmaxf(ad.z + fast_tanh(v[2] + b), 0.f); r.w = fmaxf(ad.w + fast_tanh(v[3] + b), 0.f);
        } else {
            const float lo = (m0 + row < a.relu_rows) ? 0.f : -__builtin_inff();
            r.x = fmaxf(v[0] + b, lo); r.y = fmaxf(v[1] + b, lo); r.z = fmaxf(v[2] + b, lo); r.w = fmaxf(v[3] + b, lo);
        }
        *reinterpret_cast<float4*>(a.out + o) = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// MS-TCN after its entry convs + the block's residual and ReLU: 16 output channels x 4 output frames
// ---------------------------------------------------------------------------------------------------------------------
struct F2TcnArgs {
    int N, Cin, Cout, T, T2, stride, Cb, nb, ks, res_mode, vect, vecr;
    int dil[4];
    const float* h;                      // (N, Cout, T, V): entry outputs (ReLU applied) of branches 0..nb, then the plain branch
    const float* wt[4]; const float* bt[4];   // [Cb][Cb*ks] folded, [Cb]
    const float *sp, *tp;                // pooled branch's BatchNorm as an affine [Cb]
    const float *x, *wr, *br;            // block input (N, Cin, T, V); folded residual conv [Cout][Cin], [Cout]
    float* out;                          // (N, Cout, T2, V)
    float* xpart;                        // NULL | (N, ceil(T2 / 4), Cout, V): sum over each tile's frames of out (the next block's xbar)
};

constexpr int F2_HF = 15, F2_PH = F2_HF * F2_V + 4;    // halo tile: 3*stride + (ks-1)*dil + 1 <= 15 frames

__global__ __launch_bounds__(F2_NT) void f2_tcn_kernel(const F2TcnArgs a) {
    constexpr int V = F2_V, NT = F2_NT, PB = F2_PB, PH = F2_PH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = a.res_mode == 2 ? (a.Cin + 15) & ~15 : 0;
    float* Pp = smem;                    // [4][16][PB]
    float* Ot = Pp + 4 * 16 * PB;        // [16][PB]      finished tile (for the column sums)
    float* Xs = Ot + 16 * PB;            // [Kp][PB]      strided frames of the block input (convolutional residual)
    float* Hs = Xs + Kp * PB;            // [Cb][PH]      this branch's entry output with the temporal halo
    const int nmt = a.Cout / 16;
    const int mtile = blockIdx.x % nmt, tt = blockIdx.x / nmt, n = blockIdx.y;
    const int c0 = mtile * 16, t0 = tt * F2_BT, bt = min(F2_BT, a.T2 - t0);
    const int branch = c0 / a.Cb, cb0 = c0 - branch * a.Cb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const long long TV = (long long)a.T * V, TV2 = (long long)a.T2 * V;
    const bool temporal = branch < a.nb;
    int d = 1, pad = 0, tlo = 0;
    if (temporal) {
        d = a.dil[branch];
        pad = ((a.ks - 1) * d) / 2;
        tlo = t0 * a.stride - pad;
        const int nfr = (F2_BT - 1) * a.stride + (a.ks - 1) * d + 1;
        const float* hb = a.h + ((long long)n * a.Cout + branch * a.Cb) * TV;
        const int cnt = a.Cb * nfr * 5;
        for (int e0 = tid; e0 < cnt; e0 += 8 * NT) {
            float4 q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * NT;
                const int ci = e / (nfr * 5), rem = e - ci * nfr * 5;
                const int f = rem / 5, v4 = (rem - f * 5) * 4;
                const int t = tlo + f;
                q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < cnt && t >= 0 && t < a.T) q[i] = *reinterpret_cast<const float4*>(hb + ci * TV + (long long)t * V + v4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * NT;
                const int ci = e / (nfr * 5), rem = e - ci * nfr * 5;
                const int f = rem / 5, v4 = (rem - f * 5) * 4;
                if (e < cnt) *reinterpret_cast<float4*>(Hs + ci * PH + f * V + v4) = q[i];
            }
        }
    }
    if (a.res_mode == 2)
        f2_stage(Xs, a.x + (long long)n * a.Cin * TV + (long long)t0 * a.stride * V, TV, a.Cin, Kp, bt, a.stride, tid);
    __syncthreads();
    {
        f32x4 acc[F2_NCT];
#pragma unroll
        for (int ct = 0; ct < F2_NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (temporal) {
            // this lane's columns: (frame tl, joint v) of tile ct -> offset of tap 0 inside a halo row
            int boff[F2_NCT];
#pragma unroll
            for (int ct = 0; ct < F2_NCT; ++ct) {
                const int col = ct * 16 + j, tl = col / V, v = col - tl * V;
                boff[ct] = tl * a.stride * V + v;
            }
            const int K = a.Cb * a.ks;
            f2_gemm16<F2_NCT>(acc, a.wt[branch] + (long long)(cb0 + j) * K, K, a.vect != 0, wave, 4, kq, [&](int k, int ct) {
                const int ci = k / a.ks, tap = k - ci * a.ks;
                return Hs[ci * PH + tap * d * V + boff[ct]];
            });
        }
        if (a.res_mode == 2)
            f2_gemm16<F2_NCT>(acc, a.wr + (long long)(c0 + j) * a.Cin, a.Cin, a.vecr != 0, wave, 4, kq,
                              [&](int k, int ct) { return Xs[k * PB + ct * 16 + j]; });
#pragma unroll
        for (int ct = 0; ct < F2_NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pp[(wave * 16 + kq * 4 + r) * PB + ct * 16 + j] = acc[ct][r];
    }
    __syncthreads();
    const int Ch = (a.nb + 1) * a.Cb;
    for (int e = tid; e < 16 * 20; e += NT) {
        const int row = e / 20, c4 = (e - row * 20) * 4;
        const int tl = c4 / V, v = c4 - tl * V;
        if (tl >= bt) continue;
        f32x4 acc = *reinterpret_cast<const f32x4*>(Pp + row * PB + c4);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(Pp + (w * 16 + row) * PB + c4);
            acc += p;
        }
        const int c = c0 + row, cb = cb0 + row, tq = t0 + tl, ts = tq * a.stride;
        f32x4 val;
        if (temporal) {
            const float b = a.bt[branch][cb];
            val = (f32x4){b, b, b, b};
        } else if (branch == a.nb) {                               // MaxPool2d((3,1), stride, pad 1) of the ReLU'd entry output, then its BatchNorm
            const float* hp = a.h + ((long long)n * a.Cout + c) * TV + v;
            f32x4 m = *reinterpret_cast<const f32x4*>(hp + (long long)ts * V);
            if (ts - 1 >= 0) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(hp + (long long)(ts - 1) * V);
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], q[i]);
            }
            if (ts + 1 < a.T) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(hp + (long long)(ts + 1) * V);
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], q[i]);
            }
            const float sp = a.sp[cb], tp = a.tp[cb];
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] = fmaf(sp, m[i], tp);
        } else {                                                   // plain branch: computed with the entry convs (rows >= Ch of h)
            val = *reinterpret_cast<const f32x4*>(a.h + ((long long)n * a.Cout + Ch + cb) * TV + (long long)ts * V + v);
        }
        val += acc;
        if (a.res_mode == 1) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(a.x + ((long long)n * a.Cin + c) * TV + (long long)tq * V + v);
            val += q;
        } else if (a.res_mode == 2) {
            const float b = a.br[c];
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] += b;
        }
        float4 r = make_float4(fmaxf(val[0], 0.f), fmaxf(val[1], 0.f), fmaxf(val[2], 0.f), fmaxf(val[3], 0.f));
        *reinterpret_cast<float4*>(a.out + ((long long)n * a.Cout + c) * TV2 + (long long)tq * V + v) = r;
        *reinterpret_cast<float4*>(Ot + row * PB + c4) = r;
    }
    if (a.xpart) {
        __syncthreads();
        if (tid < 16 * 5) {
            const int row = tid / 5, v4 = (tid - row * 5) * 4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tl = 0; tl < bt; ++tl) {
                const float4 q = *reinterpret_cast<const float4*>(Ot + row * PB + tl * V + v4);
                o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
            }
            const int ntt = (a.T2 + F2_BT - 1) / F2_BT;
            *reinterpret_cast<float4*>(a.xpart + (((long long)n * ntt + tt) * a.Cout + c0 + row) * V + v4) = o;
        }
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t f2_e_lds(int Cin, int R) {
    const int Kp = (Cin + 15) & ~15, R2p = 2 * R < 16 ? 16 : 2 * R, Rp = (R + 15) & ~15;
    const size_t tail = (size_t)Rp * F2_PD > (size_t)4 * Kp * F2_V ? (size_t)Rp * F2_PD : (size_t)4 * Kp * F2_V;
    return sizeof(float) * ((size_t)Kp * F2_PX + (size_t)R2p * F2_PX + 64 * F2_PX + tail);
}
size_t f2_gcn_lds(int Cin) { return sizeof(float) * ((size_t)((Cin + 15) & ~15) * F2_PB + 2 * 32 * F2_PB + 3 * F2_ES); }
size_t f2_gemm_lds(int K) { return sizeof(float) * ((size_t)((K + 15) & ~15) * F2_PB + 4 * 16 * F2_PB); }
size_t f2_tcn_lds(int Cin, int Cb, int res_mode) {
    return sizeof(float) * ((size_t)5 * 16 * F2_PB + (size_t)(res_mode == 2 ? (Cin + 15) & ~15 : 0) * F2_PB + (size_t)Cb * F2_PH);
}

int f2_fill(const tamgcn_f2_gcn_desc* d, F2GcnArgs* a, const char* who) {
    TG_CHECK(d && d->x && d->w12 && d->b12 && d->w4 && d->b4 && d->A && d->alpha && d->w3 && d->b3 && d->sy && d->ty && d->E,
             "%s: null pointer", who);
    TG_CHECK(d->V == F2_V && d->S == 3, "%s: V=%d S=%d (this family is built for V = 20, S = 3)", who, d->V, d->S);
    TG_CHECK(d->N > 0 && d->T > 0 && d->Cin > 0 && d->Cin <= 256 && d->Cout > 0 && d->Cout % 16 == 0,
             "%s: bad shape N=%d T=%d Cin=%d Cout=%d (Cin <= 256, Cout %% 16 == 0)", who, d->N, d->T, d->Cin, d->Cout);
    TG_CHECK(d->R >= 1 && d->R <= 32, "%s: R=%d outside 1..32", who, d->R);
    TG_CHECK(d->res_mode >= 0 && d->res_mode <= 2, "%s: res_mode=%d", who, d->res_mode);
    TG_CHECK(d->res_mode != 1 || d->Cin == d->Cout, "%s: identity residual needs Cin == Cout", who);
    TG_CHECK(d->res_mode != 2 || (d->wd && d->bd), "%s: convolutional residual without weights", who);
    TG_CHECK(al16(d->x) && al16(d->E) && (!d->xpart || al16(d->xpart)), "%s: x, E and xpart must be 16-byte aligned", who);
    a->N = d->N; a->Cin = d->Cin; a->Cout = d->Cout; a->T = d->T; a->S = d->S; a->R = d->R; a->res_mode = d->res_mode;
    a->x = d->x; a->w12 = d->w12; a->b12 = d->b12; a->w4 = d->w4; a->b4 = d->b4; a->A = d->A; a->alpha = d->alpha;
    a->w3 = d->w3; a->b3 = d->b3; a->sy = d->sy; a->ty = d->ty; a->wd = d->wd; a->bd = d->bd;
    a->E = d->E; a->sum = d->sum; a->diff = d->diff; a->xpart = d->xpart;
    a->vec12 = d->Cin % 16 == 0 && al16(d->w12);
    a->vec3 = d->Cin % 16 == 0 && al16(d->w3);
    a->vecd = d->res_mode == 2 && d->Cin % 16 == 0 && al16(d->wd);
    a->vec4 = d->R % 16 == 0 && al16(d->w4);
    return 0;
}

}  // namespace

extern "C" int tamgcn_f2_e(const tamgcn_f2_gcn_desc* d, void* stream) {
    F2GcnArgs a;
    if (f2_fill(d, &a, "tamgcn_f2_e")) return -1;
    static tg_devmask flag = 0;
    const size_t lds = f2_e_lds(d->Cin, d->R);
    tg_allow_lds((const void*)f2_e_kernel, f2_e_lds(256, 32), &flag);
    hipLaunchKernelGGL(f2_e_kernel, dim3(d->S * (d->Cout / 16), d->N), dim3(F2_NT), lds, (hipStream_t)stream, a);
    tamgcn_note_kernel("f2_e_kernel");
    TG_LAUNCH_CHECK("tamgcn_f2_e");
    return 0;
}

extern "C" int tamgcn_f2_gcn(const tamgcn_f2_gcn_desc* d, void* stream) {
    F2GcnArgs a;
    if (f2_fill(d, &a, "tamgcn_f2_gcn")) return -1;
    TG_CHECK(d->sum && d->diff && al16(d->sum) && al16(d->diff), "tamgcn_f2_gcn: null or misaligned output");
    static tg_devmask flag = 0;
    const size_t lds = f2_gcn_lds(d->Cin);
    tg_allow_lds((const void*)f2_gcn_kernel, f2_gcn_lds(256), &flag);
    hipLaunchKernelGGL(f2_gcn_kernel, dim3(ceil_div(d->T, F2_BT) * (d->Cout / F2_CT), d->N), dim3(F2_NT), lds, (hipStream_t)stream, a);
    tamgcn_note_kernel("f2_gcn_kernel");
    TG_LAUNCH_CHECK("tamgcn_f2_gcn");
    return 0;
}

extern "C" int tamgcn_f2_gemm(const tamgcn_f2_gemm_desc* d, void* stream) {
    TG_CHECK(d && d->x && d->w && d->b && d->out, "tamgcn_f2_gemm: null pointer");
    TG_CHECK(d->V == F2_V, "tamgcn_f2_gemm: V=%d (built for V = 20)", d->V);
    TG_CHECK(d->N > 0 && d->T > 0 && d->K > 0 && d->K <= 256 && d->M > 0 && d->M % 16 == 0,
             "tamgcn_f2_gemm: bad shape N=%d T=%d K=%d M=%d (K <= 256, M %% 16 == 0)", d->N, d->T, d->K, d->M);
    TG_CHECK(d->mode == 0 || d->mode == 1, "tamgcn_f2_gemm: mode=%d", d->mode);
    TG_CHECK(d->mode != 0 || d->add, "tamgcn_f2_gemm: mode 0 needs the addend");
    TG_CHECK(al16(d->x) && al16(d->out) && (!d->add || al16(d->add)), "tamgcn_f2_gemm: activations must be 16-byte aligned");
    F2GemmArgs a;
    a.N = d->N; a.K = d->K; a.M = d->M; a.T = d->T; a.mode = d->mode; a.relu_rows = d->relu_rows;
    a.vec = d->K % 16 == 0 && al16(d->w);
    a.x = d->x; a.w = d->w; a.b = d->b; a.add = d->add; a.out = d->out;
    static tg_devmask flag = 0;
    tg_allow_lds((const void*)f2_gemm_kernel, f2_gemm_lds(256), &flag);
    hipLaunchKernelGGL(f2_gemm_kernel, dim3(ceil_div(d->T, F2_BT) * (d->M / 16), d->N), dim3(F2_NT), f2_gemm_lds(d->K), (hipStream_t)stream, a);
    tamgcn_note_kernel("f2_gemm_kernel");
    TG_LAUNCH_CHECK("tamgcn_f2_gemm");
    return 0;
}

extern "C" int tamgcn_f2_tcn(const tamgcn_f2_tcn_desc* d, void* stream) {
    TG_CHECK(d && d->h && d->out && d->sp && d->tp, "tamgcn_f2_tcn: null pointer");
    TG_CHECK(d->V == F2_V, "tamgcn_f2_tcn: V=%d (built for V = 20)", d->V);
    TG_CHECK(d->N > 0 && d->T > 0 && d->Cout > 0 && d->Cout % 16 == 0 && d->stride >= 1 && d->stride <= 2,
             "tamgcn_f2_tcn: bad shape N=%d T=%d Cout=%d stride=%d", d->N, d->T, d->Cout, d->stride);
    TG_CHECK(d->nb >= 1 && d->nb <= 4 && d->Cb % 16 == 0 && d->Cb <= 64 && (d->nb + 2) * d->Cb == d->Cout,
             "tamgcn_f2_tcn: nb=%d Cb=%d Cout=%d (Cb %% 16 == 0, Cb <= 64, (nb + 2) Cb == Cout)", d->nb, d->Cb, d->Cout);
    TG_CHECK(d->ks >= 1 && d->ks % 2 == 1, "tamgcn_f2_tcn: kernel size %d", d->ks);
    for (int b = 0; b < d->nb; ++b) {
        TG_CHECK(d->wt[b] && d->bt[b] && d->dil[b] >= 1, "tamgcn_f2_tcn: branch %d: null weights or dilation %d", b, d->dil[b]);
        TG_CHECK((F2_BT - 1) * d->stride + (d->ks - 1) * d->dil[b] + 1 <= F2_HF, "tamgcn_f2_tcn: branch %d: halo of k=%d dilation %d stride %d exceeds %d frames",
                 b, d->ks, d->dil[b], d->stride, F2_HF);
    }
    TG_CHECK(d->res_mode >= 0 && d->res_mode <= 2, "tamgcn_f2_tcn: res_mode=%d", d->res_mode);
    TG_CHECK(d->res_mode == 0 || d->x, "tamgcn_f2_tcn: residual without the block input");
    TG_CHECK(d->res_mode != 1 || (d->Cin == d->Cout && d->stride == 1), "tamgcn_f2_tcn: identity residual needs Cin == Cout, stride 1");
    TG_CHECK(d->res_mode != 2 || (d->wr && d->br && d->Cin > 0 && d->Cin <= 256), "tamgcn_f2_tcn: convolutional residual: weights / Cin=%d", d->Cin);
    TG_CHECK(al16(d->h) && al16(d->out) && (!d->x || al16(d->x)) && (!d->xpart || al16(d->xpart)), "tamgcn_f2_tcn: activations must be 16-byte aligned");
    F2TcnArgs a;
    a.N = d->N; a.Cin = d->Cin; a.Cout = d->Cout; a.T = d->T; a.stride = d->stride; a.T2 = (d->T - 1) / d->stride + 1;
    a.Cb = d->Cb; a.nb = d->nb; a.ks = d->ks; a.res_mode = d->res_mode;
    bool vt = (d->Cb * d->ks) % 16 == 0;
    for (int b = 0; b < 4; ++b) {
        a.dil[b] = b < d->nb ? d->dil[b] : 1;
        a.wt[b] = b < d->nb ? d->wt[b] : nullptr;
        a.bt[b] = b < d->nb ? d->bt[b] : nullptr;
        if (b < d->nb) vt = vt && al16(d->wt[b]);
    }
    a.vect = vt;
    a.vecr = d->res_mode == 2 && d->Cin % 16 == 0 && al16(d->wr);
    a.h = d->h; a.sp = d->sp; a.tp = d->tp; a.x = d->x; a.wr = d->wr; a.br = d->br; a.out = d->out; a.xpart = d->xpart;
    const size_t lds = f2_tcn_lds(d->Cin, d->Cb, d->res_mode);
    TG_CHECK(lds <= 160 * 1024, "tamgcn_f2_tcn: %zu bytes of LDS", lds);
    static tg_devmask flag = 0;
    tg_allow_lds((const void*)f2_tcn_kernel, 160 * 1024, &flag);
    hipLaunchKernelGGL(f2_tcn_kernel, dim3(ceil_div(a.T2, F2_BT) * (d->Cout / 16), d->N), dim3(F2_NT), lds, (hipStream_t)stream, a);
    tamgcn_note_kernel("f2_tcn_kernel");
    TG_LAUNCH_CHECK("tamgcn_f2_tcn");
    return 0;
}
