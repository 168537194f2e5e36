// GPU-box probe: what an LDS-DMA ring delivers when nothing consumes the data -- the ceiling of the load path
// the pointwise GEMM kernels share.  Each workgroup (512 threads) streams `rows` rows of 1280 B per chunk
// (global_load_lds_dwordx4, 1 KB pieces) through an NST-stage ring with one raw barrier per chunk.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

typedef float f32x4 __attribute__((ext_vector_type(4)));
// WORK: k4-steps of (7 LDS reads + 10 MFMAs) per wave and chunk, the shape of the pointwise GEMM's inner loop
template <int NST, int PIECES, int WORK = 0>   // PIECES = 1 KB pieces per wave and chunk
__global__ __launch_bounds__(512) void probe(const float* x, float* out, int nchunk, long long block_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int STG = 8 * PIECES * 256;                   // floats per stage
    const float* base = x + (long long)blockIdx.x * block_stride + (wave * PIECES) * 256 + lane * 4;
    auto issue = [&](int c) {
        float* st = smem + (c % NST) * STG + wave * PIECES * 256;
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            __builtin_amdgcn_global_load_lds((gptr)(base + (long long)c * STG + i * 256), (lptr)(st + i * 256), 16, 0, 0);
    };
    float acc = 0.f;
    f32x4 ac[10];
    for (int i = 0; i < 10; ++i) ac[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NST - 1 && c < nchunk; ++c) issue(c);
    for (int c = 0; c < nchunk; ++c) {
        if (c + NST - 1 <= nchunk) {
            if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (c + 1 < nchunk) {
                if (PIECES * (NST - 2) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (PIECES * (NST - 2) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (PIECES * (NST - 2) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (PIECES * (NST - 2) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (PIECES * (NST - 2) == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if (PIECES * (NST - 2) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + NST - 1 < nchunk) issue(c + NST - 1);
        acc += smem[(c % NST) * STG + threadIdx.x];         // token consumer
        const float* st = smem + (c % NST) * STG;
#pragma unroll
        for (int w = 0; w < WORK; ++w) {
            float a0 = st[(w * 64 + lane) & (STG - 1)], a1 = st[(w * 64 + lane + 512) & (STG - 1)];
            float b[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) b[i] = st[(lane + 64 * i + 256 * wave + w * 32) & (STG - 1)];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                ac[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[i], ac[2 * i], 0, 0, 0);
                ac[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[i], ac[2 * i + 1], 0, 0, 0);
            }
        }
    }
    for (int i = 0; i < 10; ++i) acc += ac[i][0] + ac[i][1] + ac[i][2] + ac[i][3];
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <int NST, int PIECES, int WORK = 0>
void run(const float* x, float* out, size_t total_floats, size_t pad_lds) {
    constexpr int STG = 8 * PIECES * 256;
    const int nchunk = 24;
    const long long bs = (long long)STG * nchunk;
    const int nblk = (int)(total_floats / bs);
    const size_t lds = sizeof(float) * NST * STG + pad_lds;
    hipFuncSetAttribute((const void*)probe<NST, PIECES, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<NST, PIECES, WORK>), dim3(nblk), dim3(512), lds, 0, x, out, nchunk, bs);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((probe<NST, PIECES, WORK>), dim3(nblk), dim3(512), lds, 0, x, out, nchunk, bs);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nblk * bs * 4;
    printf("work %d  stages %d  chunk %3d KB  lds/block %6.1f KB (%d blocks/CU)  blocks %5d : %7.1f us  %5.2f TB/s\n", WORK, NST, STG * 4 / 1024,
           lds / 1024.0, (int)(160 * 1024 / lds), nblk, ms * 1e3 / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}

int main() {
    const size_t n = (size_t)1 << 28;                       // 1 GiB of floats
    float *x, *out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 1 << 20);
    hipMemset(x, 0, n * 4);
    const size_t quarter = n / 4;                           // 256 MB working set, like one layer's operand
    run<3, 3>(x, out, quarter, 0);                          // ring only
    run<3, 3, 2>(x, out, quarter, 0);   run<3, 3, 4>(x, out, quarter, 0);   run<3, 3, 8>(x, out, quarter, 0);
    run<3, 5, 2>(x, out, quarter, 0);   run<3, 5, 4>(x, out, quarter, 0);
    run<3, 3, 2>(x, out, quarter, 45 * 1024);               // one workgroup per CU
    run<3, 3, 4>(x, out, n, 0);                             // 1 GB working set
    // the split data-gradient GEMM's shape: two stages of 57 KB (seven pieces per wave), one workgroup per CU
    run<2, 7>(x, out, quarter, 0);      run<2, 7, 4>(x, out, quarter, 0);   run<2, 7, 8>(x, out, quarter, 0);   run<2, 7, 16>(x, out, quarter, 0);
    run<2, 5>(x, out, quarter, 0);      run<2, 5, 8>(x, out, quarter, 0);   run<2, 3>(x, out, quarter, 0);      run<2, 3, 8>(x, out, quarter, 0);
    return 0;
}
