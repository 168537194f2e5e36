// BatchNorm bookkeeping kernels.  The heavy part of train-mode BatchNorm (the
// two per-channel moments and the normalise/affine pass) is fused into the
// producing / consuming kernels; these tiny kernels only turn the per-block
// partial sums into per-channel coefficients (fp64 finalisation) and keep the
// running statistics exactly as nn.BatchNorm2d does (momentum update with the
// unbiased variance, num_batches_tracked += 1).
// Reference call sites: models/ctrgcn.py:64,100,115,118,123,186,213,221,230.
#include "common.h"

namespace {

__device__ __forceinline__ double wave_sum64d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void bn_fwd_finalize_body(int c,
        const float* part, int part_ctot, int part_coff, int nparts, double count,
        const float* gamma, const float* beta, float* rmean, float* rvar, long long* nbt,
        float momentum, float eps, int training, float* coef, float* save, int coef_ctot, int coef_coff) {
    const int lane = threadIdx.x;
    double mean, var;
    if (training) {
        const float* p1 = part + ((long long)0 * part_ctot + part_coff + c) * nparts;
        const float* p2 = part + ((long long)1 * part_ctot + part_coff + c) * nparts;
        double s1 = 0.0, s2 = 0.0;
        for (int i = lane; i < nparts; i += 64) { s1 += (double)p1[i]; s2 += (double)p2[i]; }
        s1 = wave_sum64d(s1); s2 = wave_sum64d(s2);
        mean = s1 / count;
        var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)rmean[c];
        var = (double)rvar[c];
    }
    if (lane == 0) {
        double invstd = 1.0 / sqrt(var + (double)eps);
        float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        float c1 = (float)((double)g * invstd);
        int cc = coef_coff + c;
        coef[cc] = c1;
        coef[coef_ctot + cc] = 0.f;
        coef[2 * coef_ctot + cc] = (float)((double)b - mean * (double)g * invstd);
        save[cc] = (float)mean;
        save[coef_ctot + cc] = (float)invstd;
        if (training) {
            double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            if (rmean) rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
            if (rvar) rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
            if (nbt && c == 0) *nbt += 1;
        }
    }
}

__global__ __launch_bounds__(64) void bn_fwd_finalize_kernel(
        const float* part, int part_ctot, int part_coff, int nparts, double count,
        const float* gamma, const float* beta, float* rmean, float* rvar, long long* nbt,
        float momentum, float eps, int training, float* coef, float* save, int coef_ctot, int coef_coff) {
    bn_fwd_finalize_body(blockIdx.x, part, part_ctot, part_coff, nparts, count, gamma, beta, rmean, rvar, nbt, momentum, eps, training,
                         coef, save, coef_ctot, coef_coff);
}

__device__ __forceinline__ void bn_bwd_finalize_body(int c,
        const float* part, int part_ctot, int part_coff, int nparts, double count,
        const float* gamma, const float* save, int save_ctot, int save_coff, int training,
        float* dgamma, float* dbeta, float* dbias_conv, float* coef, int coef_ctot, int coef_coff) {
    const int lane = threadIdx.x;
    const float* p1 = part + ((long long)0 * part_ctot + part_coff + c) * nparts;
    const float* p2 = part + ((long long)1 * part_ctot + part_coff + c) * nparts;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < nparts; i += 64) { s1 += (double)p1[i]; s2 += (double)p2[i]; }
    s1 = wave_sum64d(s1); s2 = wave_sum64d(s2);          // s1 = sum dz, s2 = sum dz * (x_pre - mean)
    if (lane == 0) {
        double mean = (double)save[save_coff + c], invstd = (double)save[save_ctot + save_coff + c];
        double g = gamma ? (double)gamma[c] : 1.0;
        double dg = s2 * invstd;                          // sum dz * xhat (producers centre by the saved mean)
        double a = g * invstd;
        double c1 = a, c2 = 0.0, c0 = 0.0;
        if (training) {
            c2 = -a * invstd * dg / count;
            c0 = -a * s1 / count - c2 * mean;
        }
        if (dgamma) dgamma[c] = (float)dg;
        if (dbeta) dbeta[c] = (float)s1;
        // bias gradient of the conv in front of the BN = sum over (n,t,v) of d x_pre
        if (dbias_conv) dbias_conv[c] = (float)(c1 * s1 + c2 * (mean * count) + c0 * count);
        int cc = coef_coff + c;
        coef[cc] = (float)c1;
        coef[coef_ctot + cc] = (float)c2;
        coef[2 * coef_ctot + cc] = (float)c0;
    }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(
        const float* part, int part_ctot, int part_coff, int nparts, double count,
        const float* gamma, const float* save, int save_ctot, int save_coff, int training,
        float* dgamma, float* dbeta, float* dbias_conv, float* coef, int coef_ctot, int coef_coff) {
    bn_bwd_finalize_body(blockIdx.x, part, part_ctot, part_coff, nparts, count, gamma, save, save_ctot, save_coff, training,
                         dgamma, dbeta, dbias_conv, coef, coef_ctot, coef_coff);
}

// Several BatchNorms in ONE launch (blockIdx.y = descriptor): a block's forward finalises ~11 of them and its backward ~9,
// each a 6-us launch for arithmetic on <= 256 channels (1.6 ms of a 31 ms step); the ones whose inputs are ready together
// (the three entry norms of MS-TCN, its five output norms after the branch join, bn + down of unit_gcn) now share a launch.
constexpr int BNM_MAX = 8;
struct BnFwdMulti { tamgcn_bn_fwd_desc d[BNM_MAX]; };
struct BnBwdMulti { tamgcn_bn_bwd_desc d[BNM_MAX]; };

__global__ __launch_bounds__(64) void bn_fwd_finalize_multi_kernel(const BnFwdMulti m) {
    const tamgcn_bn_fwd_desc& d = m.d[blockIdx.y];
    if ((int)blockIdx.x >= d.C) return;
    bn_fwd_finalize_body(blockIdx.x, d.part, d.part_ctot, d.part_coff, d.nparts, d.count, d.gamma, d.beta, d.running_mean, d.running_var,
                         d.num_batches_tracked, d.momentum, d.eps, d.training, d.coef, d.save, d.coef_ctot, d.coef_coff);
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_multi_kernel(const BnBwdMulti m) {
    const tamgcn_bn_bwd_desc& d = m.d[blockIdx.y];
    if ((int)blockIdx.x >= d.C) return;
    bn_bwd_finalize_body(blockIdx.x, d.part, d.part_ctot, d.part_coff, d.nparts, d.count, d.gamma, d.save, d.save_ctot, d.save_coff, d.training,
                         d.dgamma, d.dbeta, d.dbias_conv, d.coef, d.coef_ctot, d.coef_coff);
}

}  // namespace

extern "C" int tamgcn_bn_fwd_finalize_multi(const tamgcn_bn_fwd_desc* descs, int n, void* stream) {
    TG_CHECK(descs && n > 0, "tamgcn_bn_fwd_finalize_multi: bad args");
    for (int i0 = 0; i0 < n; i0 += BNM_MAX) {
        BnFwdMulti m;
        const int k = n - i0 < BNM_MAX ? n - i0 : BNM_MAX;
        int maxc = 0;
        for (int i = 0; i < k; ++i) {
            const tamgcn_bn_fwd_desc& d = descs[i0 + i];
            TG_CHECK(d.coef && d.save && d.C > 0, "tamgcn_bn_fwd_finalize_multi: descriptor %d: bad args", i0 + i);
            TG_CHECK(!d.training || (d.part && d.nparts > 0 && d.count > 0), "tamgcn_bn_fwd_finalize_multi: descriptor %d: training needs partial sums", i0 + i);
            TG_CHECK(d.training || (d.running_mean && d.running_var), "tamgcn_bn_fwd_finalize_multi: descriptor %d: eval needs running stats", i0 + i);
            TG_CHECK(d.coef_coff + d.C <= d.coef_ctot, "tamgcn_bn_fwd_finalize_multi: descriptor %d: coef slice out of range", i0 + i);
            m.d[i] = d;
            if (d.C > maxc) maxc = d.C;
        }
        hipLaunchKernelGGL(bn_fwd_finalize_multi_kernel, dim3(maxc, k), dim3(64), 0, (hipStream_t)stream, m);
    }
    tamgcn_note_kernel("bn_fwd_finalize_multi_kernel");
    TG_LAUNCH_CHECK("tamgcn_bn_fwd_finalize_multi");
    return 0;
}

extern "C" int tamgcn_bn_bwd_finalize_multi(const tamgcn_bn_bwd_desc* descs, int n, void* stream) {
    TG_CHECK(descs && n > 0, "tamgcn_bn_bwd_finalize_multi: bad args");
    for (int i0 = 0; i0 < n; i0 += BNM_MAX) {
        BnBwdMulti m;
        const int k = n - i0 < BNM_MAX ? n - i0 : BNM_MAX;
        int maxc = 0;
        for (int i = 0; i < k; ++i) {
            const tamgcn_bn_bwd_desc& d = descs[i0 + i];
            TG_CHECK(d.part && d.save && d.coef && d.C > 0 && d.nparts > 0 && d.count > 0, "tamgcn_bn_bwd_finalize_multi: descriptor %d: bad args", i0 + i);
            TG_CHECK(d.coef_coff + d.C <= d.coef_ctot && d.save_coff + d.C <= d.save_ctot && d.part_coff + d.C <= d.part_ctot,
                     "tamgcn_bn_bwd_finalize_multi: descriptor %d: slice out of range", i0 + i);
            m.d[i] = d;
            if (d.C > maxc) maxc = d.C;
        }
        hipLaunchKernelGGL(bn_bwd_finalize_multi_kernel, dim3(maxc, k), dim3(64), 0, (hipStream_t)stream, m);
    }
    tamgcn_note_kernel("bn_bwd_finalize_multi_kernel");
    TG_LAUNCH_CHECK("tamgcn_bn_bwd_finalize_multi");
    return 0;
}

extern "C" int tamgcn_bn_fwd_finalize(const float* part, int part_ctot, int part_coff, int nparts, double count,
                                      const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, long long* num_batches_tracked,
                                      float momentum, float eps, int training,
                                      float* coef, float* save, int coef_ctot, int coef_coff, int C, void* stream) {
    TG_CHECK(coef && save && C > 0, "tamgcn_bn_fwd_finalize: bad args");
    TG_CHECK(!training || (part && nparts > 0 && count > 0), "tamgcn_bn_fwd_finalize: training needs partial sums");
    TG_CHECK(training || (running_mean && running_var), "tamgcn_bn_fwd_finalize: eval needs running stats");
    TG_CHECK(coef_coff + C <= coef_ctot, "tamgcn_bn_fwd_finalize: coef slice out of range");
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream,
                       part, part_ctot, part_coff, nparts, count, gamma, beta, running_mean, running_var,
                       num_batches_tracked, momentum, eps, training, coef, save, coef_ctot, coef_coff);
    tamgcn_note_kernel("bn_fwd_finalize_kernel");
    TG_LAUNCH_CHECK("tamgcn_bn_fwd_finalize");
    return 0;
}

// The two-source prologue of unit_gcn's offset_conv input (models/ctrgcn.py:256-258, diff = down(x) - bn(y)) as ONE launch
// instead of torch's neg / sub / ones_like / stack (42 tiny launches per training step):  out [3][C] =
//   mode 0 (down = conv + bn):  (cd[0], -cy[0], cd[2] - cy[2])      mode 1 (down = identity):  (1, -cy[0], -cy[2])
//   mode 2 (no residual):       (-cy[0], 0, -cy[2])
__global__ void coef_diff_kernel(const float* cd, const float* cy, float* out, int C, int mode) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float y0 = cy[c], y2 = cy[2 * C + c];
    // (written with selects: hipcc 7.2 compiled the if / else-if / else form of this into code whose third branch stored
    // r0 = 1 and an unset register for r2 -- found by tests/test_gpu_primitives.py::test_coef_diff_kernel; mode 2 has no caller
    // in the CTR-GCN models, whose unit_gcn always carries a residual)
    const bool m0 = mode == 0, m1 = mode == 1;
    const float d0 = m0 ? cd[c] : 0.f, d2 = m0 ? cd[2 * C + c] : 0.f;
    const float r0 = m0 ? d0 : (m1 ? 1.f : -y0);
    const float r1 = (m0 || m1) ? -y0 : 0.f;
    const float r2 = m0 ? d2 - y2 : -y2;
    out[c] = r0; out[C + c] = r1; out[2 * C + c] = r2;
}

extern "C" int tamgcn_coef_diff(const float* coef_d, const float* coef_y, float* out, int C, int mode, void* stream) {
    TG_CHECK(coef_y && out && C > 0 && mode >= 0 && mode <= 2 && (mode != 0 || coef_d), "tamgcn_coef_diff: bad args");
    hipLaunchKernelGGL(coef_diff_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, coef_d, coef_y, out, C, mode);
    tamgcn_note_kernel("coef_diff_kernel");
    TG_LAUNCH_CHECK("tamgcn_coef_diff");
    return 0;
}

extern "C" int tamgcn_bn_bwd_finalize(const float* part, int part_ctot, int part_coff, int nparts, double count,
                                      const float* gamma, const float* save, int save_ctot, int save_coff,
                                      int training, float* dgamma, float* dbeta, float* dbias_conv,
                                      float* coef, int coef_ctot, int coef_coff, int C, void* stream) {
    TG_CHECK(part && save && coef && C > 0 && nparts > 0 && count > 0, "tamgcn_bn_bwd_finalize: bad args");
    TG_CHECK(coef_coff + C <= coef_ctot && save_coff + C <= save_ctot && part_coff + C <= part_ctot,
             "tamgcn_bn_bwd_finalize: slice out of range");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream,
                       part, part_ctot, part_coff, nparts, count, gamma, save, save_ctot, save_coff, training,
                       dgamma, dbeta, dbias_conv, coef, coef_ctot, coef_coff);
    tamgcn_note_kernel("bn_bwd_finalize_kernel");
    TG_LAUNCH_CHECK("tamgcn_bn_bwd_finalize");
    return 0;
}
