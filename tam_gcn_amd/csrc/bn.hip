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

__global__ __launch_bounds__(64) void bn_fwd_finalize_kernel(
        const float* part, int part_ctot, int part_coff, int nparts, double count,
        const float* gamma, const float* beta, float* rmean, float* rvar, long long* nbt,
        float momentum, float eps, int training, float* coef, float* save, int coef_ctot, int coef_coff) {
    const int c = blockIdx.x, lane = threadIdx.x;
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

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(
        const float* part, int part_ctot, int part_coff, int nparts, double count,
        const float* gamma, const float* save, int save_ctot, int save_coff, int training,
        float* dgamma, float* dbeta, float* dbias_conv, float* coef, int coef_ctot, int coef_coff) {
    const int c = blockIdx.x, lane = threadIdx.x;
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

}  // namespace

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
