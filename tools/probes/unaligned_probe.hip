// GPU-box probe: do 16-byte vector loads (global_load_dwordx4 to registers, and global_load_lds_dwordx4 = LDS-DMA) work
// from source addresses that are only 4-byte aligned?  V = 25 rows (T*V*4 bytes, T*V % 4 != 0) start on arbitrary dword
// boundaries; if the hardware takes them, the NTU path can use the 16-byte kernels unchanged.  Reports correctness for
// each misalignment (0, 4, 8, 12 bytes) and the streaming rate of a misaligned DMA read relative to the aligned one.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/unaligned_probe.hip -o gpurun_out/unaligned_probe && gpurun_out/unaligned_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void check(const float* x, int off, float* out_reg, float* out_dma) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    const float* p = x + off + lane * 4;                       // off floats of misalignment
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    for (int k = 0; k < 4; ++k) out_reg[lane * 4 + k] = v[k];
    __builtin_amdgcn_global_load_lds((gptr)p, (lptr)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 4; ++k) out_dma[lane * 4 + k] = lds[lane * 4 + k];
}

__global__ __launch_bounds__(512) void stream(const float* x, int off, int nchunk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = x + off + ((long long)blockIdx.x * nchunk) * 8192 + wave * 1024 + lane * 4;
    float acc = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        float* st = smem + (c & 1) * 8192 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr)(base + (long long)c * 8192 + i * 256), (lptr)(st + i * 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += smem[(c & 1) * 8192 + threadIdx.x];
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

int main() {
    const long long n = 1ll << 28;                              // 1 GiB of floats
    float *x, *o1, *o2, *sink;
    hipMalloc(&x, n * 4 + 64); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024); hipMalloc(&sink, 4096 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMemcpy(x, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, x, off, o1, o2);
        std::vector<float> a(256), b(256);
        hipMemcpy(a.data(), o1, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), o2, 1024, hipMemcpyDeviceToHost);
        int bad1 = 0, bad2 = 0;
        for (int i = 0; i < 256; ++i) { bad1 += a[i] != (float)(i + off); bad2 += b[i] != (float)(i + off); }
        printf("misalign %2d B: dwordx4->reg %s (%d bad)   dwordx4->LDS (DMA) %s (%d bad)\n", off * 4, bad1 ? "WRONG" : "ok", bad1, bad2 ? "WRONG" : "ok", bad2);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, nchunk = (int)(n / 8192 / blocks);
    for (int off = 0; off < 4; ++off) {
        hipLaunchKernelGGL(stream, dim3(blocks), dim3(512), 65536, 0, x, off, nchunk, sink);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(stream, dim3(blocks), dim3(512), 65536, 0, x, off, nchunk, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("DMA stream, misalign %2d B: %.2f TB/s\n", off * 4, 5.0 * blocks * nchunk * 8192 * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
