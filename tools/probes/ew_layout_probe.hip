// GPU-box probe: why do the row-wise element-wise kernels stream at 4.8-5.3 TB/s where torch.add reaches 6.0 (492 MB tensors)?
// out = relu(c1[ch] * a + c0[ch] + b) over (N*C rows) x L floats, three ways:
//   rowwise   the product's mapping: 64 lanes per row, a lane walks its row in 1 KB strides (4 rows per 256-thread workgroup)
//   flat      torch's mapping: a workgroup owns one contiguous 16 KB chunk of the flat array, a thread 4 x 16 bytes of it, all
//             loads of a thread requested up front; the channel index comes from a division per 16-byte group
//   rowchunk  rows, but a workgroup owns a contiguous 16 KB piece of ONE row (a thread 4 x 16 bytes, loads up front)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/ew_layout_probe.hip -o tools/probes/ew_layout_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void rowwise(const float* a, const float* b, const float* coef, float* out, int rows, int C, int L) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), li = threadIdx.x & 63;
    if (row >= rows) return;
    const int c = row % C;
    const float c1 = coef[c], c0 = coef[2 * C + c];
    const float4* ap = reinterpret_cast<const float4*>(a + (long long)row * L);
    const float4* bp = reinterpret_cast<const float4*>(b + (long long)row * L);
    float4* op = reinterpret_cast<float4*>(out + (long long)row * L);
    for (int i = li; i < (L >> 2); i += 64) {
        float4 x = ap[i], y = bp[i], v;
        v.x = fmaxf(fmaf(c1, x.x, c0) + y.x, 0.f); v.y = fmaxf(fmaf(c1, x.y, c0) + y.y, 0.f);
        v.z = fmaxf(fmaf(c1, x.z, c0) + y.z, 0.f); v.w = fmaxf(fmaf(c1, x.w, c0) + y.w, 0.f);
        op[i] = v;
    }
}
__global__ __launch_bounds__(256) void flat(const float* a, const float* b, const float* coef, float* out, long long n4, int C, int L4) {
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
    float4 x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256;
        if (i < n4) { x[k] = reinterpret_cast<const float4*>(a)[i]; y[k] = reinterpret_cast<const float4*>(b)[i]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256;
        if (i < n4) {
            const int c = (int)((i / L4) % C);
            const float c1 = coef[c], c0 = coef[2 * C + c];
            float4 v;
            v.x = fmaxf(fmaf(c1, x[k].x, c0) + y[k].x, 0.f); v.y = fmaxf(fmaf(c1, x[k].y, c0) + y[k].y, 0.f);
            v.z = fmaxf(fmaf(c1, x[k].z, c0) + y[k].z, 0.f); v.w = fmaxf(fmaf(c1, x[k].w, c0) + y[k].w, 0.f);
            reinterpret_cast<float4*>(out)[i] = v;
        }
    }
}
__global__ __launch_bounds__(256) void rowchunk(const float* a, const float* b, const float* coef, float* out, int rows, int C, int L, int cpr) {
    const int row = blockIdx.x / cpr, ch = blockIdx.x - row * cpr;      // cpr = 16 KB chunks per row
    const int c = row % C;
    const float c1 = coef[c], c0 = coef[2 * C + c];
    const int L4 = L >> 2;
    const float4* ap = reinterpret_cast<const float4*>(a + (long long)row * L);
    const float4* bp = reinterpret_cast<const float4*>(b + (long long)row * L);
    float4* op = reinterpret_cast<float4*>(out + (long long)row * L);
    float4 x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = ch * 1024 + k * 256 + threadIdx.x; if (i < L4) { x[k] = ap[i]; y[k] = bp[i]; } }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = ch * 1024 + k * 256 + threadIdx.x;
        if (i < L4) {
            float4 v;
            v.x = fmaxf(fmaf(c1, x[k].x, c0) + y[k].x, 0.f); v.y = fmaxf(fmaf(c1, x[k].y, c0) + y[k].y, 0.f);
            v.z = fmaxf(fmaf(c1, x[k].z, c0) + y[k].z, 0.f); v.w = fmaxf(fmaf(c1, x[k].w, c0) + y[k].w, 0.f);
            op[i] = v;
        }
    }
}
// rowwg: one 256-thread workgroup per row, the four waves walk it together in 4 KB strides (loop kept, loads not batched)
__global__ __launch_bounds__(256) void rowwg(const float* a, const float* b, const float* coef, float* out, int rows, int C, int L) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float c1 = coef[c], c0 = coef[2 * C + c];
    const float4* ap = reinterpret_cast<const float4*>(a + (long long)row * L);
    const float4* bp = reinterpret_cast<const float4*>(b + (long long)row * L);
    float4* op = reinterpret_cast<float4*>(out + (long long)row * L);
    for (int i = threadIdx.x; i < (L >> 2); i += 256) {
        float4 x = ap[i], y = bp[i], v;
        v.x = fmaxf(fmaf(c1, x.x, c0) + y.x, 0.f); v.y = fmaxf(fmaf(c1, x.y, c0) + y.y, 0.f);
        v.z = fmaxf(fmaf(c1, x.z, c0) + y.z, 0.f); v.w = fmaxf(fmaf(c1, x.w, c0) + y.w, 0.f);
        op[i] = v;
    }
}
// rowwave: one 64-thread workgroup per row (the product's lane mapping, but a workgroup = one row)
__global__ __launch_bounds__(64) void rowwave(const float* a, const float* b, const float* coef, float* out, int rows, int C, int L) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float c1 = coef[c], c0 = coef[2 * C + c];
    const float4* ap = reinterpret_cast<const float4*>(a + (long long)row * L);
    const float4* bp = reinterpret_cast<const float4*>(b + (long long)row * L);
    float4* op = reinterpret_cast<float4*>(out + (long long)row * L);
    for (int i = threadIdx.x; i < (L >> 2); i += 64) {
        float4 x = ap[i], y = bp[i], v;
        v.x = fmaxf(fmaf(c1, x.x, c0) + y.x, 0.f); v.y = fmaxf(fmaf(c1, x.y, c0) + y.y, 0.f);
        v.z = fmaxf(fmaf(c1, x.z, c0) + y.z, 0.f); v.w = fmaxf(fmaf(c1, x.w, c0) + y.w, 0.f);
        op[i] = v;
    }
}
template <typename F> static double timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e-3;
}
int main() {
    const int shapes[4][3] = {{256 * 64, 64, 1280}, {256 * 64, 64, 7500}, {256 * 256, 256, 1875 * 4 / 4 * 1}, {128 * 256, 256, 32768}};
    for (auto& sh : shapes) {
        const int rows = sh[0], C = sh[1]; int L = sh[2]; L &= ~3;
        const long long n = (long long)rows * L;
        float *a, *b, *o, *coef;
        (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMalloc(&o, n * 4); (void)hipMalloc(&coef, 3 * C * 4);
        (void)hipMemset(a, 0, n * 4); (void)hipMemset(b, 0, n * 4); (void)hipMemset(coef, 0, 3 * C * 4);
        const double bytes = 3.0 * n * 4;
        const double t1 = timeit([&] { hipLaunchKernelGGL(rowwise, dim3((rows + 3) / 4), dim3(256), 0, 0, a, b, coef, o, rows, C, L); });
        const long long n4 = n / 4;
        const double t2 = timeit([&] { hipLaunchKernelGGL(flat, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, a, b, coef, o, n4, C, L / 4); });
        const int cpr = (L / 4 + 1023) / 1024;
        const double t3 = timeit([&] { hipLaunchKernelGGL(rowchunk, dim3((unsigned)(rows * cpr)), dim3(256), 0, 0, a, b, coef, o, rows, C, L, cpr); });
        const double t4 = timeit([&] { hipLaunchKernelGGL(rowwg, dim3(rows), dim3(256), 0, 0, a, b, coef, o, rows, C, L); });
        const double t5 = timeit([&] { hipLaunchKernelGGL(rowwave, dim3(rows), dim3(64), 0, 0, a, b, coef, o, rows, C, L); });
        printf("rows %6d x L %6d (%5.0f MB/tensor): rowwise %5.2f TB/s   flat %5.2f   rowchunk %5.2f   rowwg %5.2f   rowwave %5.2f\n", rows, L, n * 4 / 1e6, bytes / t1 / 1e12, bytes / t2 / 1e12, bytes / t3 / 1e12, bytes / t4 / 1e12, bytes / t5 / 1e12);
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(o); (void)hipFree(coef);
    }
    return 0;
}
