// GPU-box probe: do vector-ALU instructions run beside the matrix instructions, or in their place?
// Each wave loops over (8 independent MFMAs + NV independent VALU instructions); MODE picks the VALU kind; SPLIT = 1 gives the
// MFMAs to even waves and the VALU work to odd waves of the same SIMD (2 waves/SIMD) instead of mixing them in one wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coissue_probe.hip -o tools/probes/coissue_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND: 0 v_mfma_f32_16x16x4_f32, 1 v_mfma_f32_16x16x32_bf16, 2 v_mfma_f32_32x32x2_f32      MODE: 0 v_fma_f32, 1 v_add_u32, 2 v_max_f32
template <int KIND, int NV, int MODE, int SPLIT>
__global__ __launch_bounds__(512) void probe(int iters, float* sink) {
    f32x4 acc[8];
    f32x16 acc32[4];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(threadIdx.x * 1e-3f + i); bh[i] = (__bf16)(threadIdx.x * 2e-3f - i); }
    float v[16]; int u[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x + i; }
    const int wave = threadIdx.x >> 6;
    // waves 0..3 sit on SIMDs 0..3, waves 4..7 again: SPLIT gives waves 0-3 the matrix work and waves 4-7 the vector work
    const bool do_m = !SPLIT || wave < 4, do_v = !SPLIT || wave >= 4;
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
                else if (i < 4) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[i], 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i & 15]) : "v"(a));
                else if (MODE == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i & 15]) : "v"(wave));
                else asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(a));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc32[i][e];
    for (int i = 0; i < 16; ++i) s += v[i] + (float)u[i];
    if (s == 123.456f) sink[blockIdx.x] = s;
}

template <int KIND, int NV, int MODE, int SPLIT>
static void run(const char* what, float* sink) {
    const int iters = 4000, blocks = 256;                  // one 512-thread workgroup per CU: 2 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, NV, MODE, SPLIT>), dim3(blocks), dim3(512), 0, 0, iters, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<KIND, NV, MODE, SPLIT>), dim3(blocks), dim3(512), 0, 0, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us_iter = ms * 1e3 / 3 / iters;           // per loop iteration
    const double cyc = us_iter * 2400.0;                   // shader cycles at 2.4 GHz
    printf("%-28s NV=%2d %-9s %-18s %7.1f cycles / iteration\n", KIND == 0 ? "mfma_f32_16x16x4_f32 x8" : KIND == 1 ? "mfma_f32_16x16x32_bf16 x8" : "mfma_f32_32x32x2_f32 x4", NV,
           MODE == 0 ? "v_fma_f32" : MODE == 1 ? "v_add_u32" : "v_max_f32", SPLIT ? "separate waves" : "same wave", cyc);
    (void)what;
}

int main() {
    float* sink; (void)hipMalloc(&sink, 1 << 20);
    printf("# one 512-thread workgroup per CU (2 waves/SIMD).  Same-wave rows: each wave issues the MFMAs and NV VALU per iteration, so a SIMD sees 2x(MFMAs) + 2x(NV VALU).\n");
    printf("# Separate-wave rows: one wave per SIMD issues the MFMAs, the other the VALU: a SIMD sees 1x + 1x.\n");
    run<0, 0, 0, 0>("", sink); run<0, 8, 0, 0>("", sink); run<0, 16, 0, 0>("", sink); run<0, 32, 0, 0>("", sink);
    run<0, 16, 1, 0>("", sink); run<0, 32, 1, 0>("", sink); run<0, 32, 2, 0>("", sink);
    run<0, 0, 0, 1>("", sink); run<0, 16, 0, 1>("", sink); run<0, 32, 0, 1>("", sink); run<0, 64, 0, 1>("", sink); run<0, 64, 1, 1>("", sink);
    run<1, 0, 0, 0>("", sink); run<1, 16, 0, 0>("", sink); run<1, 32, 0, 0>("", sink); run<1, 32, 1, 0>("", sink);
    run<1, 0, 0, 1>("", sink); run<1, 32, 0, 1>("", sink); run<1, 64, 0, 1>("", sink);
    run<2, 0, 0, 0>("", sink); run<2, 16, 0, 0>("", sink); run<2, 32, 0, 0>("", sink); run<2, 32, 1, 0>("", sink);
    run<2, 0, 0, 1>("", sink); run<2, 32, 0, 1>("", sink); run<2, 64, 0, 1>("", sink);
    return 0;
}
