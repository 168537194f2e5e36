/*
 * tamgcn.h — C ABI of the MI355X-native CTR-GCN hot path (libtamgcn.so).
 *
 * The reference (Tamnemng/TAM-GCN) has no FFI / operator registry: its hot
 * path is stock ATen ops issued from models/ctrgcn.py.  The
 * entry points below are therefore what a binding for that path binds
 * instead of those ATen calls; each one cites the reference lines whose
 * arithmetic it replaces.  Plain pointers and sizes only: no torch types.
 *
 * Conventions
 *   - every tensor is fp32, dense, NCHW-contiguous (N, C, T, V), V innermost,
 *     and lives in device (HBM) memory owned by the caller;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     allocates nothing, never synchronises the device and is safe inside
 *     hipGraph stream capture; scratch/partial buffers are caller-provided;
 *   - return value 0 = launched; <0 = rejected before any launch
 *     (tamgcn_last_error() gives the reason for the calling thread);
 *   - calls are re-entrant per stream: no global mutable state;
 *   - skeletons whose joint count V is not a multiple of 4 (NTU: 25): the 16-byte kernels read the last piece of a frame
 *     whole, i.e. up to 12 bytes past the last element of a SOURCE activation -- such buffers must be readable for 16 bytes
 *     behind their end (the Python layer allocates that slack and copies caller tensors that lack it).
 *
 * "src" operands.  Most kernels read their activation operand through a fused
 * per-channel affine/mix prologue so that train-mode BatchNorm apply (forward)
 * and BatchNorm-backward apply never cost a separate pass over HBM:
 *      value(n,c,t,v) = act( c1[c]*x1 + c2[c]*x2 + c0[c] )
 * with coef = [3][ctot] = (c1, c2, c0) indexed by absolute channel, x2 and
 * coef optional (NULL: value = x1), act 0 = identity, 1 = ReLU.
 * Out-of-range taps (temporal zero padding) are 0 *after* the prologue, as in
 * the reference where padding follows BN+ReLU (models/ctrgcn.py:95-107).
 */
#ifndef TAMGCN_H
#define TAMGCN_H

#ifdef __cplusplus
extern "C" {
#endif

#define TAMGCN_VERSION 401          /* round 4: bumped on every change of a struct layout, signature or documented semantics */
#define TAMGCN_MAX_SUBSETS 3
#define TAMGCN_MAX_V 32             /* joints supported by the LDS-resident CTRGC tiles (V in {20, 25}); V in {32, 64}: tamgcn_ctrgc_tiled_* */

typedef struct tamgcn_src {
    const float* x1;
    const float* x2;                /* optional */
    const float* coef;              /* optional, [3][ctot] */
    int ctot;                       /* channels of the x1/x2 allocations */
    int coff;                       /* first channel used */
    int act;                        /* 0 none, 1 relu */
} tamgcn_src;

/* ------------------------------------------------------------------------
 * Library
 * --------------------------------------------------------------------- */
int         tamgcn_version(void);
const char* tamgcn_last_error(void);
/* symbol (template arguments included) of the kernel the calling thread's last ABI call launched;
 * lets a profiler attribute HIP-event timings to the rows of a rocprofv3 kernel trace */
const char* tamgcn_last_kernel(void);
/* GEMM arithmetic policy (process-wide; initial value from the environment variable TAMGCN_SPLIT_BF16, default 0 =
 * the reference's own arithmetic; 1 is an opt-in):
 *   0  exact fp32-input MFMA (v_mfma_f32_16x16x4_f32) in every GEMM;
 *   1  as 0 in the forward; weight-gradient GEMMs and the data-gradient GEMMs into >= 128 channels run as a 2-term
 *      bf16 split (three v_mfma_f32_16x16x32_bf16, ~4.5e-6 relative error): their results never feed an activation.
 * Takes effect for launches issued after the call; not a stream operation. */
int         tamgcn_get_split_mode(void);
int         tamgcn_set_split_mode(int mode);

/* bytes of LDS the CTRGC kernels need for (S subsets, V joints, R rel-channels): the fused LDS-resident workgroup for
 * V = 20, the streaming family (tamgcn_ctrgc_tiled_*) for V in {25, 32, 64};
 * <0 if the shape is unsupported.  Lets the host fail early and loudly. */
int         tamgcn_ctrgc_lds_bytes(int S, int V, int R);

/* ------------------------------------------------------------------------
 * k x 1 convolution over (N,C,T,V) as an MFMA GEMM (fp32 in / fp32 acc,
 * v_mfma_f32_16x16x4_f32).  Replaces aten::convolution for
 *   - 1x1 channel mixing: CTRGC conv1..4 on pooled joints, unit_gcn.down,
 *     unit_gcn.offset_conv, the 4 MS-TCN branch entry convs, unit_tcn
 *     (reference models/ctrgcn.py:161-164, 211-213, 219-221, 95-99, 114, 122, 183)
 *   - dilated temporal convs (TemporalConv, models/ctrgcn.py:56-62)
 * and, with wmode = 1, for their data gradients (aten::convolution_backward,
 * input part): the weight is then read transposed with taps flipped and the
 * src is zero-upsampled by `up` (the forward stride).
 *
 *   y[n, ycoff+m, (t*ostride), v] = bias[m]
 *        + sum_{k,j} W(m,k,j) * src(n, k, t*stride + j*dil - pad, v)
 *        + bcast[m, n, v]*bcast_scale + add1[...] + add2[...]
 *   then  y *= (mask_src > 0)  if mask is given,
 *   and per-channel partial sums  (sum y, sum y*(aux - aux_center))  -> stats_part
 *   (aux = y itself, uncentred, when aux is NULL: the train-mode BatchNorm moments).
 * ---------------------------------------------------------------------- */
typedef struct tamgcn_conv_desc {
    tamgcn_src src;                 /* (N, src.ctot, T_in, V); uses K channels from src.coff */
    int N, K, T_in, V;
    const float* w;                 /* wmode 0: [M][K][KT]   wmode 1: [K][M][KT] (read transposed+flipped) */
    const float* bias;              /* [M] or NULL */
    int M, KT, dil, stride, pad;
    int wmode;                      /* 0 forward, 1 data-gradient */
    int up;                         /* src zero-upsampling factor (1 = none); wmode 1 with strided fwd */
    float* y;                       /* (N, yctot, T_y, V) */
    int yctot, ycoff;
    int T_out;                      /* number of output t computed */
    int T_y;                        /* T extent of the y allocation */
    int ostride;                    /* output t stride (1; 2 scatters a strided 1x1 data-gradient) */
    const float* add1;              /* optional, same geometry as y (may alias y) */
    const float* add2;              /* optional, same geometry as y */
    const float* bcast;             /* optional [M][N][V], broadcast over t */
    float bcast_scale;
    const tamgcn_src* mask;         /* optional, geometry of y: y *= (value > 0) */
    const float* aux;               /* optional, geometry (N, auxctot, T_y, V) at auxcoff */
    const float* aux_center;        /* [auxctot] per-channel value subtracted from aux (the BN batch mean) */
    int auxctot, auxcoff;
    float* stats_part;              /* optional [2][stats_ctot][nparts] written at channel stats_coff+m */
    int stats_ctot, stats_coff;
    /* eval-mode fusion: with BatchNorm folded to a per-channel affine the convolution can finish
     * its consumer's work:  y = act( c1[ch]*(conv + bias) + c0[ch] + bcast + add1 + add2 ),  ch = ycoff + m,
     * c1 = post_coef[ch], c0 = post_coef[2*post_ctot + ch] (the [3][post_ctot] layout tamgcn_bn_fwd_finalize writes),
     * post_act 1 = ReLU after the adds.  NULL / 0: the plain form above. */
    const float* post_coef;
    int post_ctot, post_act;
} tamgcn_conv_desc;

/* number of stats partials per channel the call writes (= N * t-tiles) */
int tamgcn_conv_nparts(const tamgcn_conv_desc* d);
int tamgcn_conv(const tamgcn_conv_desc* d, void* stream);

/* Weight gradient of the same convolution (aten::convolution_backward, weight part):
 *   dW[m][k][j] = sum_{n,t,v} gy(n,m,t,v) * src(n,k,t*stride + j*dil - pad, v)
 * Both operands go through the prologue.  Split over n into `nsplit` partial
 * slabs (deterministic: no float atomics), reduced by tamgcn_reduce_sum. */
typedef struct tamgcn_wgrad_desc {
    tamgcn_src gy;                  /* (N, gy.ctot, T_out, V), M channels from gy.coff */
    tamgcn_src src;                 /* (N, src.ctot, T_in, V), K channels from src.coff */
    int N, M, K, T_in, T_out, V, KT, dil, stride, pad;
    float* part;                    /* [nsplit][M][K][KT] */
    int nsplit;
} tamgcn_wgrad_desc;
/* Largest nsplit tamgcn_wgrad accepts for this descriptor: N, or more when the contraction can also be
 * split inside a sample (1x1 stride-1 form).  Size `part` and nsplit from it. */
int tamgcn_wgrad_max_split(const tamgcn_wgrad_desc* d);
int tamgcn_wgrad(const tamgcn_wgrad_desc* d, void* stream);

/* Several slab reductions in ONE launch (a layer's backward produces ~25 of them; the sums over (n, t, v) of aten's
 * convolution_backward / native_batch_norm_backward for the modules of reference models/ctrgcn.py:53-284): for each descriptor
 * out[e] (+)= scale * sum_s part[s*stride_s + e], fp64 accumulation in a fixed order. */
typedef struct tamgcn_reduce_desc {
    const float* part; float* out;
    int nsplit, accumulate;
    long long stride_s, count;
    float scale;
} tamgcn_reduce_desc;
int tamgcn_reduce_multi(const tamgcn_reduce_desc* descs, int n, void* stream);

/* out[e] = (accumulate ? out[e] : 0) + scale * sum_{s<nsplit} part[s*stride_s + e]
 * (fp64 accumulation, fixed order).  `part` is scratch: for nsplit > 128 it is reduced in place first. */
int tamgcn_reduce_sum(float* part, int nsplit, long long stride_s, long long count,
                      float scale, int accumulate, float* out, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm bookkeeping (aten::native_batch_norm / _backward,
 * reference nn.BatchNorm2d at models/ctrgcn.py:64,100,115,118,123,186,213,221,230)
 * ---------------------------------------------------------------------- */
/* training=1: mean/var from partial sums [2][part_ctot][nparts] (channels part_coff..+C),
 *             running stats updated (momentum, unbiased var), *nbt += 1;
 * training=0: running stats used.
 * Writes coef (c1 = gamma*invstd, c2 = 0, c0 = beta - mean*c1) at channels coef_coff..+C of
 * a [3][coef_ctot] array and save = (mean, invstd) at the same channels of a [2][coef_ctot] array. */
int tamgcn_bn_fwd_finalize(const float* part, int part_ctot, int part_coff, int nparts, double count,
                           const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* num_batches_tracked,
                           float momentum, float eps, int training,
                           float* coef, float* save, int coef_ctot, int coef_coff, int C, void* stream);

/* Coefficients of unit_gcn's offset_conv input diff = down(x) - bn(y) (reference models/ctrgcn.py:256-258) as a two-source
 * prologue c1*x1 + c2*x2 + c0 from the [3][C] sets tamgcn_bn_fwd_finalize wrote for down's BatchNorm (coef_d) and bn (coef_y):
 *   mode 0 (x1 = down's conv output, x2 = y):  (cd[0], -cy[0], cd[2] - cy[2])
 *   mode 1 (down = identity, x1 = x, x2 = y):  (1, -cy[0], -cy[2])          mode 2 (no residual, x1 = y):  (-cy[0], 0, -cy[2]) */
int tamgcn_coef_diff(const float* coef_d, const float* coef_y, float* out, int C, int mode, void* stream);

/* part = [2][part_ctot][nparts] partial sums of (dz, dz*(x_pre - mean)) per channel; every backward
 * reducer below centres by the saved batch mean (save[0]) so that dgamma has no cancellation.
 * Produces dgamma, dbeta, the bias gradient of the conv that fed the BN (optional) and the
 * backward-apply coefficients  d x_pre = c1*dz + c2*x_pre + c0  (train: full BN backward;
 * eval: c1 = gamma*invstd, c2 = c0 = 0). */
int tamgcn_bn_bwd_finalize(const float* part, int part_ctot, int part_coff, int nparts, double count,
                           const float* gamma, const float* save, int save_ctot, int save_coff,
                           int training, float* dgamma, float* dbeta, float* dbias_conv,
                           float* coef, int coef_ctot, int coef_coff, int C, void* stream);

/* Several BatchNorms in one launch: the same arithmetic as the two entry points above, one descriptor per BatchNorm
 * (fields = their arguments).  A block's forward finalises ~11 BatchNorms and its backward ~9; the ones whose partial sums
 * are ready together share a launch. */
typedef struct tamgcn_bn_fwd_desc {
    const float* part; int part_ctot, part_coff, nparts; double count;
    const float* gamma; const float* beta; float* running_mean; float* running_var; long long* num_batches_tracked;
    float momentum, eps; int training;
    float* coef; float* save; int coef_ctot, coef_coff, C;
} tamgcn_bn_fwd_desc;
typedef struct tamgcn_bn_bwd_desc {
    const float* part; int part_ctot, part_coff, nparts; double count;
    const float* gamma; const float* save; int save_ctot, save_coff, training;
    float* dgamma; float* dbeta; float* dbias_conv; float* coef; int coef_ctot, coef_coff, C;
} tamgcn_bn_bwd_desc;
int tamgcn_bn_fwd_finalize_multi(const tamgcn_bn_fwd_desc* descs, int n, void* stream);
int tamgcn_bn_bwd_finalize_multi(const tamgcn_bn_bwd_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------------
 * CTRGC — channel-wise topology refinement graph convolution
 * (reference models/ctrgcn.py:172-177, and the 3-subset sum :252-254).
 * ---------------------------------------------------------------------- */
/* xbar[c][n][v] = mean_t src(n,c,t,v)   (layout (C, N, V): one "sample" with T = N, so the
 * pooled 1x1 convs conv1/conv2 run through tamgcn_conv with N=1, T=N). */
int tamgcn_tmean(const tamgcn_src* src, int N, int C, int T, int V, float* xbar, void* stream);

typedef struct tamgcn_ctrgc_desc {
    int N, Cin, Cout, S, R, T, V;
    tamgcn_src x;                   /* block input (N, Cin, T, V) */
    const float* pq;                /* [S*2*R][N][V]: row (s*2+0)*R+r = p, (s*2+1)*R+r = q  (conv1/conv2 of xbar) */
    const float* w3;                /* [S*Cout][Cin]  conv3 weights, subsets stacked */
    const float* b3;                /* [S*Cout] */
    const float* w4;                /* [S][Cout][R]   conv4 */
    const float* b4;                /* [S][Cout] */
    const float* A;                 /* [S][V][V]      PA (or the A given to CTRGC.forward) */
    const float* alpha;             /* [1] device scalar */
    const float* E;                 /* (N, S, Cout, V, V) from tamgcn_ctrgc_build_e: fwd / bwd_dx3 load their E tiles from it */
} tamgcn_ctrgc_desc;

/* E[n,s,c,u,v] = alpha*(W4_s tanh(p_s[n,:,u] - q_s[n,:,v]) + b4_s)[c] + A_s[u,v] for every channel, once per
 * layer and sample (R <= 32): reference models/ctrgcn.py:174-176 (tanh of the pairwise difference, conv4, * alpha + A); pass it as d->E to tamgcn_ctrgc_fwd and tamgcn_ctrgc_bwd_dx3. */
int tamgcn_ctrgc_build_e(const tamgcn_ctrgc_desc* d, float* E, void* stream);

/* y[n,c,t,u] = sum_s sum_v E_s[n,c,u,v] * (W3_s x + b3_s)[n,c,t,v],  E_s = d->E (V = 20, S in {1, 3}, Cout % 8 == 0;
 * other joint counts: the tamgcn_ctrgc_tiled_* family below).  d->pq, w4, b4, A, alpha are not read.
 * stats_part (optional): [2][Cout][N] partial (sum y, sum y^2) per sample.
 * x3_out (optional): (N, S*Cout, T, V) receives x3 = W3 x + b3 (the tile is in LDS anyway): the backward's dE
 * accumulation and the conv3 weight gradient read it; NULL (inference) keeps the forward write-minimal. */
int tamgcn_ctrgc_fwd(const tamgcn_ctrgc_desc* d, float* y, float* stats_part, float* x3_out, void* stream);

/* dx3[n, s*Cout+c, t, v] = sum_u E_s[n,c,u,v] * dy(n,c,t,u);  db3_part [N][S*Cout];  E_s = d->E, d->w3 / b3 not read */
int tamgcn_ctrgc_bwd_dx3(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy,
                         float* dx3, float* db3_part, void* stream);

/* dE_s[n,c,u,v] = sum_t dy(n,c,t,u) * x3_s[n,c,t,v] pushed through E's definition (autograd of reference
 * models/ctrgcn.py:172-177 w.r.t. PA, alpha, conv4, conv1/conv2 outputs), from the x3 tamgcn_ctrgc_fwd kept (x3_out), as two launches:
 *   _de_acc   dE (N, S, Cout, V, V) = sum_t dy(n,c,t,u) * x3[n, s*Cout+c, t, v]     (streaming, HBM-bound)
 *   _de_tail  one workgroup per (n, s, channel group g of `groups`): dA_part [N*groups][S][V][V],
 *             dw4_part [N][S][Cout][R], db4_part [N][S][Cout], dalpha_part [N*S*groups],
 *             dpq [groups][S*2*R][N][V] (plain stores: every workgroup owns its slice, nothing to zero;
 *             sum the slabs over `groups`).  R <= 32, Cout % (16*groups) == 0.
 * d->x and d->w3/b3 are not read by either. */
int tamgcn_ctrgc_bwd_de_acc(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, const float* x3,
                            float* dE, void* stream);
int tamgcn_ctrgc_bwd_de_tail(const tamgcn_ctrgc_desc* d, const float* dE,
                             float* dA_part, float* dw4_part, float* db4_part,
                             float* dalpha_part, float* dpq, int groups, void* stream);

/* ---- CTRGC for large skeletons (V in {32, 64}; BASELINE.json configs[4]: V = 64, T = 512, C = 256) ---------------
 * One channel's topology E is 3*V*V floats (48 KB at V = 64): the fused LDS-resident kernels above do not apply
 * (tamgcn_ctrgc_lds_bytes < 0).  The same arithmetic (reference models/ctrgcn.py:172-177, V-generic) then runs as
 *   x3 = W3 x + b3                  tamgcn_conv (M = S*Cout), kept in HBM: (N, S*Cout, T, V)
 *   _tiled_build_e                  E (N, S, Cout, V, V), workgroup = (n, s, 512/V rows u)
 *   _tiled_agg_fwd                  y[n,c,t,u] = sum_s sum_v x3_s[n,c,t,v] E_s[n,c,u,v] on MFMA; stats_part [2][Cout][N] optional
 *   _tiled_agg_bwd                  dx3[n,s*Cout+c,t,v] = sum_u dy(n,c,t,u) E_s[n,c,u,v]; db3_part [N][S*Cout] optional
 *   _tiled_de_acc                   dE (N, S, Cout, V, V) = sum_t dy(n,c,t,u) x3[n,s*Cout+c,t,v]
 *   _tiled_de_tail                  dE -> dA_part [N][S][V][V], dw4_part [N*NUC][S][Cout][R], db4_part [N*NUC][S][Cout],
 *                                   dalpha_part [N*S*NUC], dpq [NUC][S*2*R][N][V] (sum the slabs over NUC);
 *                                   NUC = tamgcn_ctrgc_tiled_chunks(V) row chunks of E per (n, s).  R <= 32.
 * d->x, d->w3, d->b3, d->E are not read by these entry points (E and x3 are explicit arguments). */
int tamgcn_ctrgc_tiled_supported(int V);
int tamgcn_ctrgc_tiled_chunks(int V);
int tamgcn_ctrgc_tiled_build_e(const tamgcn_ctrgc_desc* d, float* E, void* stream);
int tamgcn_ctrgc_tiled_agg_fwd(const tamgcn_ctrgc_desc* d, const float* x3, const float* E, float* y, float* stats_part, void* stream);
int tamgcn_ctrgc_tiled_agg_bwd(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, const float* E, float* dx3, float* db3_part, void* stream);
int tamgcn_ctrgc_tiled_de_acc(const tamgcn_ctrgc_desc* d, const tamgcn_src* dy, const float* x3, float* dE, void* stream);
int tamgcn_ctrgc_tiled_de_tail(const tamgcn_ctrgc_desc* d, const float* dE, float* dA_part, float* dw4_part, float* db4_part,
                               float* dalpha_part, float* dpq, void* stream);

/* ------------------------------------------------------------------------
 * Element-wise block epilogues and their backward reductions.
 * All stats partial buffers are [nstat][C][nparts]; nparts is returned by
 * tamgcn_ew_nparts(N, T, V) (one partial per (n, chunk of t)).
 * ---------------------------------------------------------------------- */
int tamgcn_ew_nparts(int N, int C, int T, int V);

/* unit_gcn tail, models/ctrgcn.py:256-261:
 *   g = relu( ybn + tanh(obn) + res ),  ybn = y.coef-applied y_pre, obn likewise,
 *   res = 0 | x | coef-applied d_pre  (res may be NULL). */
int tamgcn_gcn_tail_fwd(const tamgcn_src* y, const tamgcn_src* o, const tamgcn_src* res,
                        int N, int C, int T, int V, float* g, void* stream);
/* dsum = dg*(g>0); doz = dsum*(1-tanh(obn)^2); partials (sum doz, sum doz*(o_pre - o_save[c])) */
int tamgcn_gcn_tail_bwd(const float* dg, const float* g, const tamgcn_src* o, const float* o_save,
                        int N, int C, int T, int V, float* dsum, float* doz, float* part, void* stream);
/* dyb = dsum - ddiff ; dres = dsum + ddiff (dres optional);
 * part[0..1] = (sum dyb, sum dyb*y_pre); part[2..3] = (sum dres, sum dres*r_pre) if r_pre given */
int tamgcn_gcn_mid_bwd(const float* dsum, const float* ddiff, const float* y_pre, const float* y_save,
                       const float* r_pre, const float* r_save,
                       int N, int C, int T, int V, float* dyb, float* dres, float* part, void* stream);

/* ------------------------------------------------------------------------
 * The second stage of MultiScale_TemporalConv in one launch per direction (reference models/ctrgcn.py:52-69, 101-119,
 * 137-147; backward: aten::convolution_backward's input gradient of every branch's k x 1 convolution).
 *   forward   for b < nb:  y[:, ycoff + b*Cb + m, t] = bias_b[m] + sum_{k, tap} W_b[m][k][tap] *
 *                              act(src)[:, src.coff + b*Cb + k, t*stride + tap*dil_b - (KT-1)*dil_b/2]      (zero padded)
 *             pool = 1:    y[:, ycoff + nb*Cb + c, t] = max_{i in -1..1} act(src)[:, src.coff + nb*Cb + c, t*stride + i]
 *             stats_part (optional) [2][stats_ctot][nparts]: (sum y, sum y^2) per workgroup at the channels of y.
 *             The prologue must be BatchNorm + ReLU (src.act = 1, no second source).
 *   backward  src = the gradient w.r.t. y, (N, src.ctot, T_out, V), linear two-source prologue (act = 0);
 *             for b < nb:  y[:, ycoff + b*Cb + k, th] = [mask value > 0] * sum_{m, tap, t: t*stride + tap*dil_b - pad_b = th}
 *                              W_b[m][k][tap] * src value[:, src.coff + b*Cb + m, t],        y has T_in frames;
 *             mask = the forward's source with its prologue (channel mask.coff + b*Cb + k); stats_part: (sum y,
 *             sum y * (mask.x1 - center[channel])), the entry BatchNorm's backward moments.  pool is ignored (the pooled
 *             branch's gradient is tamgcn_maxpool_bwd).
 * Built for Cb = 16 or a multiple of 32, KT in {3, 5}, (KT-1)*dil even, V <= 32 or V % 16 == 0: tamgcn_tconv_supported()
 * says whether a shape is; w_b = [Cb][Cb][KT] as nn.Conv2d stores it. */
#define TAMGCN_TCONV_MAXB 6
typedef struct tamgcn_tconv_desc {
    tamgcn_src src;
    int N, T_in, T_out, V, Cb, nb, KT, stride, pool;     /* T_in: frames of the forward's input, T_out = (T_in-1)/stride + 1 */
    int dil[TAMGCN_TCONV_MAXB];
    const float* w[TAMGCN_TCONV_MAXB];
    const float* bias[TAMGCN_TCONV_MAXB];                /* forward only; entries may be NULL */
    float* y; int yctot, ycoff;
    float* stats_part; int stats_ctot;
    const tamgcn_src* mask; const float* center;         /* backward only */
} tamgcn_tconv_desc;
int tamgcn_tconv_supported(int V, int Cb, int KT, int nb, const int* dil, int stride, int T_in);
int tamgcn_tconv_nparts(const tamgcn_tconv_desc* d, int backward);
int tamgcn_tconv_fwd(const tamgcn_tconv_desc* d, void* stream);
int tamgcn_tconv_bwd(const tamgcn_tconv_desc* d, void* stream);
/* Weight gradient of every temporal branch in one launch (aten::convolution_backward, weight part):
 *   dW_b[m][k][tap] = sum_{n,t,v} gy(n, src.coff + b*Cb + m, t, v) * act(mask)(n, mask.coff + b*Cb + k, t*stride + tap*dil_b - pad_b, v)
 * d->src = the gradient w.r.t. the branches' outputs (two-source prologue, T_out frames), d->mask = the forward's source with
 * its prologue (T_in frames).  The contraction is split over (sample, frame tile) items into nsplit = d->yctot partial slabs
 * d->y = part [nsplit][nb][Cb][Cb][KT] (deterministic: no float atomics; reduce with tamgcn_reduce_*);
 * 1 <= nsplit <= tamgcn_tconv_wgrad_max_split(d).  Shapes: as tamgcn_tconv_supported(). */
int tamgcn_tconv_wgrad_max_split(const tamgcn_tconv_desc* d);
int tamgcn_tconv_wgrad(const tamgcn_tconv_desc* d, void* stream);

/* MaxPool2d((3,1), stride (s,1), pad (1,0)) over the prologue value (models/ctrgcn.py:117),
 * written at channel ycoff of y (N, yctot, T_out, V) + (sum, sum^2) partials [2][yctot][nparts]. */
int tamgcn_maxpool_fwd(const tamgcn_src* src, int N, int C, int T_in, int V, int stride,
                       float* y, int yctot, int ycoff, int T_out, float* stats_part, void* stream);
/* the same pooled value finished for an eval-mode block (row f2):  y = act( c1[ch]*maxpool + c0[ch] + add ),
 * ch = ycoff + c, coef [3][yctot] as tamgcn_bn_fwd_finalize writes it, add optional with the geometry of y. */
int tamgcn_maxpool_post_fwd(const tamgcn_src* src, int N, int C, int T_in, int V, int stride,
                            float* y, int yctot, int ycoff, int T_out, const float* coef, const float* add, int relu, void* stream);
/* d src_value routed to the first arg-max of each window, times relu mask (value > 0);
 * gy goes through its own prologue; result written at channel dcoff of d (N, dctot, T_in, V),
 * partials (sum d, sum d*src.x1) at the same channel of [2][dctot][nparts]. */
int tamgcn_maxpool_bwd(const tamgcn_src* gy, const tamgcn_src* src, const float* src_save, int N, int C, int T_in, int T_out, int V,
                       int stride, float* d, int dctot, int dcoff, float* part, void* stream);

/* out = act( a + res ), a = prologue value of `a`, res = NULL | src (identity or coef-applied
 * conv residual); models/ctrgcn.py:145-146 and :283.  rowmean: NULL | [N][C], the mean over (t, v) of every
 * output row -- the last block hands the model head its pooling input (models/ctrgcn.py:343-345) this way.
 * xbar: NULL | [C][N][V], the mean over t of `out` -- the NEXT block's pooled joint-embedding input (what tamgcn_tmean
 * computes, models/ctrgcn.py:172-174), taken while the row is in registers; needs V % 4 == 0, V <= 64, 16-byte aligned
 * operands and rowmean == NULL. */
int tamgcn_add_act_fwd(const tamgcn_src* a, const tamgcn_src* res, int relu,
                       int N, int C, int T, int V, float* out, float* rowmean, float* xbar, void* stream);
/* dz = dout * (out>0) (relu=1; dz may be NULL when relu=0 and only sums are wanted);
 * part[0..1] = (sum dz, sum dz*a_pre), part[2..3] = (sum dz, sum dz*r_pre) if r_pre given. */
int tamgcn_add_act_bwd(const float* dout, const float* out, int relu, const float* a_pre, const float* a_save,
                       const float* r_pre, const float* r_save,
                       int N, int C, int T, int V, float* dz, float* part, void* stream);

/* y = prologue value (materialise a src; used by stand-alone modules and tests) */
int tamgcn_apply(const tamgcn_src* src, int N, int C, int T, int V, float* y, int yctot, int ycoff, void* stream);

/* ---- stem and head of models.ctrgcn.Model (reference models/ctrgcn.py:328-332, 343-348) -------------------------
 * stem: x (N, C, T, V, M) -> data_bn over channel j = (m*V + v)*C + c with statistics over (n, t) -> (N*M, C, T, V).
 *   _stem_stats  part [2][J][N], J = C*V*M: (sum a, sum a*(b - center[j])) over t; forward a = b = x, center NULL (moments
 *                for tamgcn_bn_fwd_finalize); backward a = dout (N*M, C, T, V), b = x, center = saved mean (for _bn_bwd_finalize)
 *   _stem_apply  dout NULL: out (N*M, C, T, V) = c1[j]*x + c0[j];  dout given: out = dx (N, C, T, V, M) = c1*dout + c2*x + c0
 * head: pooled (N, C) = mean over (m, t, v) of x10 (N*M, C, T, V); logits (N, K) = pooled W^T + b and their gradients. */
int tamgcn_stem_stats(const float* x, const float* dout, const float* center, int N, int C, int T, int V, int M,
                      float* part, void* stream);
int tamgcn_stem_apply(const float* x, const float* dout, const float* coef, int N, int C, int T, int V, int M,
                      float* out, void* stream);
int tamgcn_head_pool_fwd(const float* x, int N, int C, int T, int V, int M, float* pooled, void* stream);
int tamgcn_head_pool_bwd(const float* dpooled, int N, int C, int T, int V, int M, float* dx, void* stream);
int tamgcn_head_fc_fwd(const float* pooled, const float* W, const float* b, int N, int C, int K, float* logits, void* stream);
int tamgcn_head_fc_bwd(const float* dlogits, const float* pooled, const float* W, int N, int C, int K,
                       float* dW, float* db, float* dpooled, void* stream);

/* ---- loss of the harness step (nn.CrossEntropyLoss(), reduction 'mean', ignore_index -100: reference
 *      processor/recognition_rgb.py:19, :62) ----
 * _ce_fwd  loss[0] = mean over the KEPT rows of (logsumexp(logits[n]) - logits[n][labels[n]]); g (N, K) = (softmax - onehot) / kept
 *          (kept for the backward).  labels int64 on the device: a row labelled -100 (torch's default ignore_index) is skipped
 *          -- zero gradient, not counted; the loss is NaN when no row is kept, as torch's is.  Any other label outside
 *          [0, K), where torch raises a device assert, makes loss[0] NaN and zeroes that row of g; nothing outside the
 *          row is read.  One launch (aten: log_softmax + nll_loss).
 * _ce_bwd  dlogits = g * dloss[0]   (aten: nll_loss_backward + log_softmax_backward). */
int tamgcn_ce_fwd(const float* logits, const long long* labels, int N, int K, float* loss, float* g, void* stream);
int tamgcn_ce_bwd(const float* g, const float* dloss, int N, int K, float* dlogits, void* stream);

/* ---- score-level ensemble of the evaluation scripts ----
 * fused (N, K) = sum_s weights[s] * scores[s]  (softmax = 0: reference ensemble/ensemble_resnet_ctrgcn.py:50-54, score_a + alpha * score_b)
 *             or sum_s weights[s] * softmax_k(scores[s])  (softmax = 1: ensemble/ensemble_ctrgcn_resnet_eval.py:99-106);
 * pred (N) int64 = first arg max over k (numpy.argmax); class_stats NULL | int32 [K][2] = (correct, total) per true class
 * (compute_accuracy, ensemble_ctrgcn_resnet_eval.py:217-234), needs labels (N) int64; a label outside [0, K) belongs to no
 * class row and is left out of every (correct, total) pair (the scripts' labels come from the split lists and are always
 * in range; their overall accuracy divides by len(labels) -- the Python layer does the same and raises on such a label).
 * scores is [S][N][K] contiguous. */
int tamgcn_score_fuse(const float* scores, const float* weights, int S, int N, int K, int softmax,
                      const long long* labels, float* fused, long long* pred, int* class_stats, void* stream);

/* ---- input side: skeleton streams and the feeder's per-sample transform ----------------------
 * _stream_derive  the other three inputs of the 4-stream recipe from a joint batch x (N, C, T, V, M) resident in HBM:
 *                 mode 1 bone        out[.., v, m] = x[.., v, m] - x[.., parent[v], m]   (reference feeder/feeder_nucla_gcn.py:27-28,
 *                                    119-123: pair (v1, v2) of self.bone = (v + 1, parent[v] + 1); the pair (3, 3) gives 0)
 *                 mode 2 motion      out[:, :, t] = x[:, :, t + 1] - x[:, :, t], last frame 0          (:124-127)
 *                 mode 3 bone-motion motion of bone (upstream CTR-GCN's fourth stream; the reference feeder's `elif`
 *                                    collapses a 'bone_motion' label path to bone -- stated in DESIGN.md)
 * _feeder_transform  reference feeder/feeder_nucla_gcn.py:85-130 for N ragged clips at once, fp64 arithmetic as numpy's:
 *                 raw (sum L_n, V, 3) fp64 with frame offsets [N + 1]; centre on joint `center_joint` of frame 0 (:98-99),
 *                 p' = p . rot[n] (3 x 3 row-major view matrix Ry.Rx.S of :75-83; identity on the val path), per-coordinate
 *                 min-max to [-1, 1] over the whole clip (:102-105), gather frames idx[n][time_steps] (:108-117: the host
 *                 draws / computes the indices exactly as the reference does), stream `mode` (0 joint, 1..3 as above),
 *                 out (N, 3, time_steps, V, 1) fp32 (:129-130, :154). */
int tamgcn_stream_derive(const float* x, int N, int C, int T, int V, int M, const int* parent, int mode, float* out, void* stream);
int tamgcn_feeder_transform(const double* raw, const long long* offsets, const double* rot, const int* idx, const int* parent,
                            int N, int V, int time_steps, int center_joint, int mode, float* out, void* stream);

/* ---- f2: the eval-mode TCN_GCN_unit for small batches (callers: reference
 * ensemble/ensemble_ctrgcn_resnet_eval.py:147-183, models/resnet_gcn_attention.py:82-85, visual.py:53-55 -- model(data)
 * on 1..16 clips in eval mode).  Five launches per block, each 50..130 workgroups per clip; V = 20, S = 3.  The CALLER folds
 * every eval-mode BatchNorm into the weights it passes (bn(W x + b) = (s W) x + (s b + t), s = gamma / sqrt(var + eps),
 * t = beta - mean s) -- tam_gcn_amd/f2.py does.  Reference arithmetic: models/ctrgcn.py:172-177, :252-261 (unit_gcn with
 * the offset_conv branch), :93-146 (MultiScale_TemporalConv), :281-283 (residual + ReLU of TCN_GCN_unit).
 *   _f2_e     E (N, S, Cout, V, V) = alpha (W4 tanh(p_u - q_v) + b4) + A with p, q = W12 mean_t(x) + b12  (conv1 / conv2 commute
 *             with the mean over T: SURVEY.md §8a);  reads x, w12, b12, w4, b4, A, alpha; writes d->E
 *   _f2_gcn   z = sum_s E_s (W3_s x + b3_s);  y = sy z + ty;  res = 0 | x | wd x + bd (res_mode 0 | 1 | 2);
 *             writes sum = y + res and diff = res - y, both (N, Cout, T, V); reads d->E
 *   _f2_gemm  out (N, M, T, V) = epilogue(W x + b), W [M][K] row-major, x (N, K, T, V):
 *             mode 0: relu(add + tanh(.))            (offset_conv + the block tail :258-261; add = sum, x = diff)
 *             mode 1: rows < relu_rows: relu(.), others plain  (entry convs of the temporal / pooled branches stacked over the
 *                     plain 1x1 branch: all read g)
 *   _f2_tcn   h = _f2_gemm mode 1's output (N, Cout, T, V) with Cout = (nb + 2) Cb: branches b < nb: k x 1 conv (dilation dil[b],
 *             stride, "same" padding) of rows [b Cb, (b+1) Cb) with wt[b] [Cb][Cb*ks] (tap innermost), + bt[b]; branch nb:
 *             MaxPool2d((3,1), stride, pad 1) of rows [nb Cb, ..) then sp . + tp; branch nb + 1: rows [(nb+1) Cb, ..) at the
 *             strided frames; + residual (res_mode 0 none | 1 the block input x | 2 wr x(strided frames) + br); ReLU;
 *             out (N, Cout, (T - 1) / stride + 1, V). */
typedef struct tamgcn_f2_gcn_desc {
    int N, Cin, Cout, T, V, S, R, res_mode;
    const float* x;                                  /* (N, Cin, T, V) */
    const float* w12; const float* b12;              /* [S*2R][Cin], [S*2R]: rows s*2R + r = conv1, s*2R + R + r = conv2 */
    const float* w4; const float* b4;                /* [S][Cout][R], [S][Cout] */
    const float* A; const float* alpha;              /* [S][V][V], [1] */
    const float* w3; const float* b3;                /* [S*Cout][Cin], [S*Cout] */
    const float* sy; const float* ty;                /* [Cout] folded unit_gcn.bn */
    const float* wd; const float* bd;                /* [Cout][Cin], [Cout] folded down conv + BatchNorm (res_mode 2) or NULL */
    float* E;                                        /* (N, S, Cout, V, V) */
    float* sum; float* diff;                         /* (N, Cout, T, V) each (not touched by _f2_e) */
    const float* xpart;                              /* NULL | (N, ceil(T/4), Cin, V): the sums over each 4-frame tile of x that the
                                                        previous block's _f2_tcn left (xpart there); _f2_e then takes mean_t(x) from
                                                        them instead of reading x again */
} tamgcn_f2_gcn_desc;
int tamgcn_f2_e(const tamgcn_f2_gcn_desc* d, void* stream);
int tamgcn_f2_gcn(const tamgcn_f2_gcn_desc* d, void* stream);

typedef struct tamgcn_f2_gemm_desc {
    int N, K, M, T, V, mode, relu_rows;
    const float* x; const float* w; const float* b; const float* add;
    float* out;
} tamgcn_f2_gemm_desc;
int tamgcn_f2_gemm(const tamgcn_f2_gemm_desc* d, void* stream);

typedef struct tamgcn_f2_tcn_desc {
    int N, Cin, Cout, T, V, stride, Cb, nb, ks, res_mode;
    int dil[4];
    const float* h;
    const float* wt[4]; const float* bt[4];
    const float* sp; const float* tp;
    const float* x; const float* wr; const float* br;
    float* out;
    float* xpart;                                    /* NULL | (N, ceil(T_out/4), Cout, V): per-tile frame sums of out */
} tamgcn_f2_tcn_desc;
int tamgcn_f2_tcn(const tamgcn_f2_tcn_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TAMGCN_H */
