// Stem and head of models.ctrgcn.Model: the input BatchNorm1d over (person, joint, channel)
// with its two permutes (reference models/ctrgcn.py:328-332) and global mean-pool + fc (:343-348).
// Tiny tensors (3-6 channels in, 256 x num_class out): the point is not bandwidth but that the model's forward /
// backward no longer contains stock framework kernels or full-tensor permute copies.
//
//   stem   x (N, C, T, V, M) --data_bn over j = (m*V + v)*C + c, statistics over (n, t)--> (N*M, C, T, V)
//   head   x10 (N*M, C, T, V) --mean over (m, t, v)--> pooled (N, C) --fc--> logits (N, K)
#include "common.h"

namespace {

constexpr int SH_NT = 256;

// part[stat][j][n]: stat 0 = sum a, stat 1 = sum a * (b - center[j]) over t   (forward: a = b = x, center = 0;
// backward: a = dout, b = x, center = saved mean)
__global__ __launch_bounds__(SH_NT) void stem_stats_kernel(const float* x, const float* dout, const float* center,
                                                           int N, int C, int T, int V, int M, float* part) {
    const int n = blockIdx.x, J = C * V * M, VM = V * M;
    for (int i = threadIdx.x; i < J; i += SH_NT) {         // i in memory order (c, v, m)
        const int c = i / VM, vm = i - c * VM, v = vm / M, m = vm - v * M;
        const int j = (m * V + v) * C + c;
        const float mu = center ? center[j] : 0.f;
        const float* xp = x + ((long long)n * C + c) * T * VM + vm;
        const float* dp = dout ? dout + (((long long)n * M + m) * C + c) * T * V + v : nullptr;
        float s1 = 0.f, s2 = 0.f;
        int t = 0;
        for (; t + 8 <= T; t += 8) {                        // eight frames per round trip; the additions keep their order
            float bb[8], aa[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { bb[k] = xp[(long long)(t + k) * VM]; aa[k] = dp ? dp[(long long)(t + k) * V] : bb[k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1 += aa[k]; s2 = fmaf(aa[k], bb[k] - mu, s2); }
        }
        for (; t < T; ++t) {
            const float b = xp[(long long)t * VM];
            const float a = dp ? dp[(long long)t * V] : b;
            s1 += a;
            s2 = fmaf(a, b - mu, s2);
        }
        part[((long long)0 * J + j) * N + n] = s1;
        part[((long long)1 * J + j) * N + n] = s2;
    }
}

// forward: out[(n*M+m), c, t, v] = c1[j] x + c0[j];  backward: dx[n,c,t,v,m] = c1[j] dout + c2[j] x + c0[j]
__global__ __launch_bounds__(SH_NT) void stem_apply_kernel(const float* x, const float* dout, const float* coef,
                                                           int N, int C, int T, int V, int M, float* out, int backward) {
    // one workgroup per (n, c) row of x: T*V*M contiguous floats.  (Round 4: the flat form of this kernel took an element index
    // apart with five 64-bit divisions -- 118 us for 4 MB; per row everything is 32-bit and the frame index comes from a reciprocal.)
    const int row = blockIdx.x, n = row / C, c = row - n * C;
    const int TV = T * V, len = TV * M, J = C * V * M;
    const float rV = 1.0f / (float)V;
    const long long xb = (long long)row * len;
    for (int i = threadIdx.x; i < len; i += SH_NT) {
        const int tv = M == 1 ? i : (M == 2 ? i >> 1 : i / M), m = i - tv * M;
        const int t = (int)(((float)tv + 0.5f) * rV), v = tv - t * V;          // tv < 2^20 (sh_dims_ok): exact
        const int j = (m * V + v) * C + c;
        const long long o = (((long long)n * M + m) * C + c) * TV + tv;         // (n*M+m, c, t, v)
        if (!backward) out[o] = fmaf(coef[j], x[xb + i], coef[2 * J + j]);
        else out[xb + i] = fmaf(coef[j], dout[o], fmaf(coef[J + j], x[xb + i], coef[2 * J + j]));
    }
}

// pooled[n][c] = mean over (m, t, v) of x[(n*M+m), c, t, v]; one wave per (n, c)
__global__ __launch_bounds__(SH_NT) void pool_fwd_kernel(const float* x, int N, int C, int L, int M, float* pooled) {
    const int row = blockIdx.x * (SH_NT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N * C) return;
    const int n = row / C, c = row - n * C;
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
        const float* p = x + (((long long)n * M + m) * C + c) * L;
        for (int i = lane; i < L; i += 64) s += p[i];
    }
    s = wave_sum64(s);
    if (lane == 0) pooled[row] = s / (float)(M * L);
}

// dx[(n*M+m), c, :, :] = dpooled[n][c] / (M*L)
__global__ __launch_bounds__(SH_NT) void pool_bwd_kernel(const float* dpooled, int N, int C, int L, int M, float* dx) {
    const int row = blockIdx.x, n = row / (M * C), c = row % C;       // row = (n*M+m)*C + c
    const float v = dpooled[n * C + c] / (float)(M * L);
    float* p = dx + (long long)row * L;
    for (int i = threadIdx.x; i < L; i += SH_NT) p[i] = v;
}

// logits[n][k] = b[k] + sum_c pooled[n][c] W[k][c]; one workgroup per sample, a wave per class (round robin)
__global__ __launch_bounds__(SH_NT) void fc_fwd_kernel(const float* pooled, const float* W, const float* b, int N, int C, int K, float* logits) {
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < K; k += SH_NT / 64) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(pooled[(long long)n * C + c], W[(long long)k * C + c], s);
        s = wave_sum64(s);
        if (lane == 0) logits[(long long)n * K + k] = s + (b ? b[k] : 0.f);
    }
}

// blockIdx.x < K: dW[k][:] = sum_n dl[n][k] pooled[n][:], db[k] = sum_n dl[n][k];  blockIdx.x >= K: dpooled[n][:] = dl[n][:] W
__global__ __launch_bounds__(SH_NT) void fc_bwd_kernel(const float* dl, const float* pooled, const float* W, int N, int C, int K,
                                                       float* dW, float* db, float* dpooled) {
    if ((int)blockIdx.x < K) {
        const int k = blockIdx.x;
        for (int c = threadIdx.x; c < C; c += SH_NT) {
            // 16 samples' operands are requested together (the loop was a chain of N dependent round trips: 77 us at N = 256);
            // the additions keep their order
            float s = 0.f;
            int n = 0;
            for (; n + 16 <= N; n += 16) {
                float a[16], b[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { a[i] = dl[(long long)(n + i) * K + k]; b[i] = pooled[(long long)(n + i) * C + c]; }
#pragma unroll
                for (int i = 0; i < 16; ++i) s = fmaf(a[i], b[i], s);
            }
            for (; n < N; ++n) s = fmaf(dl[(long long)n * K + k], pooled[(long long)n * C + c], s);
            dW[(long long)k * C + c] = s;
        }
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += dl[(long long)n * K + k];
            db[k] = s;
        }
    } else {
        const int n = blockIdx.x - K;
        for (int c = threadIdx.x; c < C; c += SH_NT) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(dl[(long long)n * K + k], W[(long long)k * C + c], s);
            dpooled[(long long)n * C + c] = s;
        }
    }
}

// ---- cross-entropy over the logits (nn.CrossEntropyLoss with its default arguments, reduction 'mean', ignore_index -100:
// reference processor/recognition_rgb.py:19, :62) ----
// ONE workgroup: thread -> samples n = tid, tid + 256, ...; per sample lse by max-shift, the loss term, and
// g[n][k] = (softmax_k - [k == y_n]) / kept (the backward only scales it by dloss).  Rows labelled ignore_index are
// skipped (zero gradient, not counted: the mean is over the kept rows, NaN when none is kept, as torch gives).  Any
// other label outside [0, K) -- torch raises a device assert there -- makes the loss NaN and zeroes that row's gradient;
// no memory outside the row is touched.  Terms are summed in fp64 in a fixed order.
constexpr long long CE_IGNORE_INDEX = -100;
__global__ __launch_bounds__(SH_NT) void ce_fwd_kernel(const float* logits, const long long* labels, int N, int K, float* loss, float* g) {
    __shared__ double red[SH_NT];
    __shared__ int cnt[SH_NT];
    __shared__ int bad[SH_NT];
    int kept = 0, nbad = 0;
    for (int n = threadIdx.x; n < N; n += SH_NT) {
        const long long y = labels[n];
        if (y >= 0 && y < K) ++kept;
        else if (y != CE_IGNORE_INDEX) ++nbad;
    }
    cnt[threadIdx.x] = kept;
    bad[threadIdx.x] = nbad;
    __syncthreads();
    for (int o = SH_NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { cnt[threadIdx.x] += cnt[threadIdx.x + o]; bad[threadIdx.x] += bad[threadIdx.x + o]; }
        __syncthreads();
    }
    kept = cnt[0];
    nbad = bad[0];
    const float invk = kept > 0 ? 1.f / (float)kept : 0.f;
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += SH_NT) {
        const float* l = logits + (long long)n * K;
        const long long y = labels[n];
        if (y < 0 || y >= K) {
            for (int k = 0; k < K; ++k) g[(long long)n * K + k] = 0.f;
            continue;
        }
        float m = l[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, l[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(l[k] - m);
        const float lse = m + logf(s);
        const float inv = 1.f / s;
        for (int k = 0; k < K; ++k) g[(long long)n * K + k] = (expf(l[k] - m) * inv - (k == y ? 1.f : 0.f)) * invk;
        acc += (double)(lse - l[y]);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = SH_NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (nbad || kept == 0) ? __builtin_nanf("") : (float)(red[0] / (double)kept);
}

__global__ __launch_bounds__(SH_NT) void ce_bwd_kernel(const float* g, const float* dloss, int total, float scale, float* dlogits) {
    const int e = blockIdx.x * SH_NT + threadIdx.x;
    if (e < total) dlogits[e] = g[e] * (dloss[0] * scale);
}

// score-level ensemble of the reference's evaluation scripts: fused[n][k] = sum_s w[s] * (softmax_k? scores[s][n][:]) ,
// pred[n] = first arg max_k (numpy.argmax), per-class (correct, total) counts.  One lane per sample, K is 10..60.
// Reference: ensemble/ensemble_resnet_ctrgcn.py:50-54 (raw scores), ensemble/ensemble_ctrgcn_resnet_eval.py:99-108, :217-234.
__global__ __launch_bounds__(SH_NT) void score_fuse_kernel(const float* scores, const float* w, int S, int N, int K, int softmax,
                                                           const long long* labels, float* fused, long long* pred, int* stats) {
    const int n = blockIdx.x * SH_NT + threadIdx.x;
    if (n >= N) return;
    float* f = fused + (long long)n * K;
    for (int k = 0; k < K; ++k) f[k] = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* x = scores + ((long long)s * N + n) * K;
        float mx = 0.f, inv = 1.f;
        if (softmax) {
            mx = x[0];
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
            float den = 0.f;
            for (int k = 0; k < K; ++k) den += expf(x[k] - mx);
            inv = 1.f / den;
        }
        const float ws = w[s];
        // separate multiply and add (no fma contraction): numpy evaluates score_a + (alpha * score_b) with both roundings
        for (int k = 0; k < K; ++k) f[k] = __fadd_rn(f[k], __fmul_rn(ws, softmax ? expf(x[k] - mx) * inv : x[k]));
    }
    int best = 0;
    for (int k = 1; k < K; ++k) if (f[k] > f[best]) best = k;
    pred[n] = best;
    if (labels && stats) {
        const int l = (int)labels[n];
        if (l >= 0 && l < K) {
            atomicAdd(&stats[2 * l + 1], 1);
            if (l == best) atomicAdd(&stats[2 * l], 1);
        }
    }
}

}  // namespace

static bool sh_dims_ok(int N, int C, int T, int V, int M) {
    return N > 0 && C > 0 && T > 0 && V > 0 && M > 0 && (long long)N * C * T * V * M < (1LL << 40);
}

extern "C" int tamgcn_stem_stats(const float* x, const float* dout, const float* center, int N, int C, int T, int V, int M,
                                 float* part, void* stream) {
    TG_CHECK(x && part && sh_dims_ok(N, C, T, V, M) && (!dout || center), "tamgcn_stem_stats: bad args");
    hipLaunchKernelGGL(stem_stats_kernel, dim3(N), dim3(SH_NT), 0, (hipStream_t)stream, x, dout, center, N, C, T, V, M, part);
    tamgcn_note_kernel("stem_stats_kernel");
    TG_LAUNCH_CHECK("tamgcn_stem_stats");
    return 0;
}

extern "C" int tamgcn_stem_apply(const float* x, const float* dout, const float* coef, int N, int C, int T, int V, int M,
                                 float* out, void* stream) {
    TG_CHECK(x && coef && out && sh_dims_ok(N, C, T, V, M), "tamgcn_stem_apply: bad args");
    TG_CHECK((long long)T * V * M < (1LL << 20) && (long long)N * C < (1LL << 31), "tamgcn_stem_apply: a row of T*V*M = %lld floats (limit 2^20)", (long long)T * V * M);
    hipLaunchKernelGGL(stem_apply_kernel, dim3((unsigned)(N * C)), dim3(SH_NT), 0, (hipStream_t)stream, x, dout, coef, N, C, T, V, M, out,
                       dout ? 1 : 0);
    tamgcn_note_kernel("stem_apply_kernel");
    TG_LAUNCH_CHECK("tamgcn_stem_apply");
    return 0;
}

extern "C" int tamgcn_head_pool_fwd(const float* x, int N, int C, int T, int V, int M, float* pooled, void* stream) {
    TG_CHECK(x && pooled && sh_dims_ok(N, C, T, V, M), "tamgcn_head_pool_fwd: bad args");
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)ceil_div(N * C, SH_NT / 64)), dim3(SH_NT), 0, (hipStream_t)stream, x, N, C, T * V, M, pooled);
    tamgcn_note_kernel("pool_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_head_pool_fwd");
    return 0;
}

extern "C" int tamgcn_head_pool_bwd(const float* dpooled, int N, int C, int T, int V, int M, float* dx, void* stream) {
    TG_CHECK(dpooled && dx && sh_dims_ok(N, C, T, V, M), "tamgcn_head_pool_bwd: bad args");
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)(N * M * C)), dim3(SH_NT), 0, (hipStream_t)stream, dpooled, N, C, T * V, M, dx);
    tamgcn_note_kernel("pool_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_head_pool_bwd");
    return 0;
}

extern "C" int tamgcn_head_fc_fwd(const float* pooled, const float* W, const float* b, int N, int C, int K, float* logits, void* stream) {
    TG_CHECK(pooled && W && logits && N > 0 && C > 0 && K > 0, "tamgcn_head_fc_fwd: bad args");
    hipLaunchKernelGGL(fc_fwd_kernel, dim3(N), dim3(SH_NT), 0, (hipStream_t)stream, pooled, W, b, N, C, K, logits);
    tamgcn_note_kernel("fc_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_head_fc_fwd");
    return 0;
}

extern "C" int tamgcn_head_fc_bwd(const float* dlogits, const float* pooled, const float* W, int N, int C, int K,
                                  float* dW, float* db, float* dpooled, void* stream) {
    TG_CHECK(dlogits && pooled && W && dW && db && dpooled && N > 0 && C > 0 && K > 0, "tamgcn_head_fc_bwd: bad args");
    hipLaunchKernelGGL(fc_bwd_kernel, dim3(K + N), dim3(SH_NT), 0, (hipStream_t)stream, dlogits, pooled, W, N, C, K, dW, db, dpooled);
    tamgcn_note_kernel("fc_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_head_fc_bwd");
    return 0;
}

extern "C" int tamgcn_ce_fwd(const float* logits, const long long* labels, int N, int K, float* loss, float* g, void* stream) {
    TG_CHECK(logits && labels && loss && g && N > 0 && K > 0, "tamgcn_ce_fwd: bad args");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(SH_NT), 0, (hipStream_t)stream, logits, labels, N, K, loss, g);
    tamgcn_note_kernel("ce_fwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_ce_fwd");
    return 0;
}

extern "C" int tamgcn_ce_bwd(const float* g, const float* dloss, int N, int K, float* dlogits, void* stream) {
    TG_CHECK(g && dloss && dlogits && N > 0 && K > 0, "tamgcn_ce_bwd: bad args");
    const int total = N * K;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)ceil_div(total, SH_NT)), dim3(SH_NT), 0, (hipStream_t)stream, g, dloss, total, 1.f, dlogits);   // g carries 1 / kept already
    tamgcn_note_kernel("ce_bwd_kernel");
    TG_LAUNCH_CHECK("tamgcn_ce_bwd");
    return 0;
}

extern "C" int tamgcn_score_fuse(const float* scores, const float* weights, int S, int N, int K, int softmax,
                                 const long long* labels, float* fused, long long* pred, int* class_stats, void* stream) {
    TG_CHECK(scores && weights && fused && pred && S > 0 && N > 0 && K > 0 && (!class_stats || labels), "tamgcn_score_fuse: bad args");
    if (class_stats) {
        hipError_t e = hipMemsetAsync(class_stats, 0, sizeof(int) * 2 * K, (hipStream_t)stream);
        TG_CHECK(e == hipSuccess, "tamgcn_score_fuse: memset failed");
    }
    hipLaunchKernelGGL(score_fuse_kernel, dim3((unsigned)ceil_div(N, SH_NT)), dim3(SH_NT), 0, (hipStream_t)stream,
                       scores, weights, S, N, K, softmax, labels, fused, pred, class_stats);
    tamgcn_note_kernel("score_fuse_kernel");
    TG_LAUNCH_CHECK("tamgcn_score_fuse");
    return 0;
}
