// GPU-box probe: what the matrix pipe sustains for the instructions the kernels use, with nothing else in the loop
// (operands in registers, independent accumulators, 1-4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o tools/probes/mfma_rate_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void f32_16x16x4(int iters, float* sink) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[blockIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void bf16_16x16x32(int iters, float* sink) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(threadIdx.x * 2e-3f - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[blockIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %8.1f TFLOP/s\n", name, 3.0 * flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* sink; (void)hipMalloc(&sink, 1 << 20);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {             // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        const int blocks = 256 * wps;
        char nm[128];
        snprintf(nm, sizeof nm, "v_mfma_f32_16x16x4_f32, 8 accumulators, %d wave(s)/SIMD", wps);
        run(nm, [&] { hipLaunchKernelGGL(f32_16x16x4<8>, dim3(blocks), dim3(256), 0, 0, iters, sink); }, (double)blocks * 4 * iters * 8 * 2048.0);
        snprintf(nm, sizeof nm, "v_mfma_f32_16x16x32_bf16, 8 accumulators, %d wave(s)/SIMD", wps);
        run(nm, [&] { hipLaunchKernelGGL(bf16_16x16x32<8>, dim3(blocks), dim3(256), 0, 0, iters, sink); }, (double)blocks * 4 * iters * 8 * 16384.0);
    }
    run("v_mfma_f32_16x16x4_f32, 2 accumulators, 1 wave/SIMD", [&] { hipLaunchKernelGGL(f32_16x16x4<2>, dim3(256), dim3(256), 0, 0, iters, sink); }, 256.0 * 4 * iters * 2 * 2048.0);
    return 0;
}
