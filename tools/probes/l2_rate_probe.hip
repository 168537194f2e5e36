// GPU-box probe: at what rate does L2-RESIDENT data reach a CU (a) by LDS-DMA (global_load_lds_dwordx4) and (b) by
// global_load_dwordx4 into registers (optionally written on to LDS)?  Every workgroup walks the same 2 MB window again and
// again, so after the first pass everything is an L2 hit; the GEMM kernels re-read their activation tile once per row tile
// (4-16x), i.e. they live on this path.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l2_rate_probe.hip -o tools/probes/l2_rate_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WIN = 512 * 1024;                    // floats in the shared window (2 MB)

// 8 KB per wave per step, NST-deep ring per wave (no cross-wave sharing: pure transport test)
template <int DEPTH>
__global__ __launch_bounds__(512) void dma(const float* x, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* my = smem + wave * (DEPTH * 1024);
    const int start = ((blockIdx.x * 8 + wave) * 1024) % WIN;
    float acc = 0.f;
    auto issue = [&](int s) {
        const float* p = x + (start + (long long)s * 1024 * 37) % WIN + lane * 4;
        float* st = my + (s % DEPTH) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr)(p + i * 256), (lptr)(st + i * 256), 16, 0, 0);
    };
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    for (int s = 0; s < steps; ++s) {
        if (s + DEPTH - 1 < steps) issue(s + DEPTH - 1);
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        acc += my[(s % DEPTH) * 1024 + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <int DEPTH, bool TOLDS>
__global__ __launch_bounds__(512) void regs(const float* x, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* my = smem + wave * 1024;
    const int start = ((blockIdx.x * 8 + wave) * 1024) % WIN;
    f32x4 buf[DEPTH][4];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load = [&](int s, f32x4 (&b)[4]) {
        const float* p = x + (start + (long long)s * 1024 * 37) % WIN + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4*>(p + i * 256);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load(d, buf[d]);
    for (int s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = s0 + d;
            if (s < steps) {
                if (s + DEPTH - 1 < steps) load(s + DEPTH - 1, buf[(d + DEPTH - 1) % DEPTH]);
                if (TOLDS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(my + i * 256 + lane * 4) = buf[d][i];
                    acc[0] += my[lane];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc += buf[d][i];
                }
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x] = acc[0];
}

template <typename F>
static void run(const char* name, F launch, int blocks, int steps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * blocks * 8.0 * steps * 4096.0;
    printf("%-44s %6.2f TB/s  (%5.1f B/clk/CU at 2.4 GHz)\n", name, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    float *x, *sink;
    hipMalloc(&x, (size_t)WIN * 4 * 2 + 65536); hipMalloc(&sink, 65536);
    hipMemset(x, 0, (size_t)WIN * 4 * 2);
    const int steps = 2000;
    for (int wg = 1; wg <= 2; ++wg) {
        const int blocks = 256 * wg;
        printf("-- %d workgroup(s) of 8 waves per CU\n", wg);
        run("LDS-DMA, 1 step (4 KB/wave) in flight", [&] { hipLaunchKernelGGL(dma<2>, dim3(blocks), dim3(512), 8 * 2 * 4096, 0, x, steps, sink); }, blocks, steps);
        run("LDS-DMA, 2 steps in flight", [&] { hipLaunchKernelGGL(dma<3>, dim3(blocks), dim3(512), 8 * 3 * 4096, 0, x, steps, sink); }, blocks, steps);
        run("LDS-DMA, 3 steps in flight", [&] { hipLaunchKernelGGL(dma<4>, dim3(blocks), dim3(512), 8 * 4 * 4096, 0, x, steps, sink); }, blocks, steps);
        run("dwordx4 -> registers, 1 step in flight", [&] { hipLaunchKernelGGL((regs<2, false>), dim3(blocks), dim3(512), 8 * 4096, 0, x, steps, sink); }, blocks, steps);
        run("dwordx4 -> registers, 2 steps in flight", [&] { hipLaunchKernelGGL((regs<3, false>), dim3(blocks), dim3(512), 8 * 4096, 0, x, steps, sink); }, blocks, steps);
        run("dwordx4 -> registers -> ds_write_b128, 2 steps", [&] { hipLaunchKernelGGL((regs<3, true>), dim3(blocks), dim3(512), 8 * 4096, 0, x, steps, sink); }, blocks, steps);
    }
    return 0;
}
