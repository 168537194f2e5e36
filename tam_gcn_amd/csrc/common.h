// Shared device/host helpers for libtamgcn (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tamgcn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// host side: error reporting
// ---------------------------------------------------------------------------
void tamgcn_set_error(const char* fmt, ...);
void tamgcn_note_kernel(const char* fmt, ...);   // symbol of the kernel the last ABI call launched (per thread)
int tamgcn_wgrad_taps(void);     // k x 1 weight gradients on the LDS-DMA kernel (TAMGCN_WGRAD_TAPS, default 1)
int tamgcn_split_mode(void);     // TAMGCN_SPLIT_BF16: 0 (default) = exact fp32-input MFMA everywhere; 1 = two-term split-fp32 on the bf16 matrix
                                 // cores in the BACKWARD GEMMs (weight gradients, data gradients into >= 128 channels): their
                                 // results never decide a ReLU mask, the 4.5e-6 relative error stays a linear perturbation

#define TG_CHECK(cond, ...)                         \
    do {                                            \
        if (!(cond)) {                              \
            tamgcn_set_error(__VA_ARGS__);          \
            return -1;                              \
        }                                           \
    } while (0)

#define TG_LAUNCH_CHECK(name)                                                         \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            tamgcn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return -2;                                                                \
        }                                                                             \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: the guard is a bit per device
// ordinal (nn.DataParallel drives several devices from one process), not a process-wide flag.  Not a stream operation.
typedef unsigned long long tg_devmask;
static inline void tg_allow_lds(const void* fn, size_t bytes, tg_devmask* done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const tg_devmask bit = 1ull << (dev & 63);
    if (!(__atomic_load_n(done, __ATOMIC_RELAXED) & bit)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
    }
}

// ---------------------------------------------------------------------------
// device side: fused operand prologue  value = act(c1*x1 + c2*x2 + c0)
// ---------------------------------------------------------------------------
struct SrcDev {
    const float* x1;
    const float* x2;
    const float* coef;   // [3][ctot] or null
    int ctot, coff, act;
};

static inline SrcDev make_src(const tamgcn_src& s) {
    SrcDev d;
    d.x1 = s.x1; d.x2 = s.x2; d.coef = s.coef; d.ctot = s.ctot; d.coff = s.coff; d.act = s.act;
    return d;
}
static inline SrcDev null_src() {
    SrcDev d; d.x1 = nullptr; d.x2 = nullptr; d.coef = nullptr; d.ctot = 0; d.coff = 0; d.act = 0;
    return d;
}

// idx = element offset inside the (N, ctot, T, V) allocation, ch = absolute channel
__device__ __forceinline__ float src_value(const SrcDev& s, long long idx, int ch) {
    float v = s.x1[idx];
    if (s.coef) {
        float c1 = s.coef[ch], c0 = s.coef[2 * s.ctot + ch];
        v = fmaf(c1, v, c0);
        if (s.x2) v = fmaf(s.coef[s.ctot + ch], s.x2[idx], v);
    }
    if (s.act == 1) v = fmaxf(v, 0.f);
    return v;
}

__device__ __forceinline__ float wave_sum16(float v) {   // sum over the 16 lanes sharing lane>>4
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// v_mfma_f32_16x16x4_f32: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D[row = 4*(lane>>4) + reg][col = lane&15]   (cdna_hip_programming.md §3)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// split-fp32 operands for the bf16 matrix cores: x = hi + lo with hi = bf16(x) (RNE), lo = bf16(x - hi);
// a.b ~= ah.bh + ah.bl + al.bh drops only al.bl and lo's rounding: ~2^-17 relative per product (4.5e-6 measured on
// the layer GEMMs, tools/split_accuracy.py), at three v_mfma_f32_16x16x32_bf16 (K = 32, 16 cycles each) instead of
// eight v_mfma_f32_16x16x4_f32 (32 cycles each): the fp32-input MFMA runs at 1/16 of the bf16 rate.  (The legacy
// K = 16 bf16 form takes 32 cycles: measured, no faster than 1.37x.)  Fragment layout: lane (i = lane & 15,
// kq = lane >> 4) supplies eight contraction indices; which eight is free as long as both operands agree.
// The split costs 12 VALU per four elements, which is what bounds these loops.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// eight fp32 values (two 16-byte fragments) -> hi and lo bf16x8 operands
__device__ __forceinline__ void split_bf16x8(const f32x4& v0, const f32x4& v1, bf16x8_t& hi, bf16x8_t& lo) {
    u32x4_t h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2_t x = {p < 2 ? v0[2 * p] : v1[2 * p - 4], p < 2 ? v0[2 * p + 1] : v1[2 * p - 3]};
        h[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));          // v_cvt_pk_bf16_f32
        f32x2_t r = {x[0] - __uint_as_float(h[p] << 16), x[1] - __uint_as_float(h[p] & 0xffff0000u)};
        l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    }
    hi = __builtin_bit_cast(bf16x8_t, h);
    lo = __builtin_bit_cast(bf16x8_t, l);
}
__device__ __forceinline__ f32x4 mfma_split(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, const bf16x8_t& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}

// tanh(x) = 1 - 2/(exp(2x)+1): absolute error ~1e-7 (D is O(1) and enters E linearly)
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.f * x);
    return 1.f - __fdividef(2.f, e + 1.f);
}

// ---------------------------------------------------------------------------
// phase tracing for tools/*_phases.py (side builds with -DTAMGCN_TRACE only): shader-clock stamps of
// wave 0 of every workgroup, summed per slot; one table and one reader per translation unit.
// ---------------------------------------------------------------------------
#ifdef TAMGCN_TRACE
#define TG_TRACE_DEFINE(READER)                                                                        \
    __device__ unsigned long long tg_trace[16];                                                        \
    extern "C" int READER(unsigned long long* out16, int reset) {                                      \
        if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(tg_trace), sizeof(unsigned long long) * 16) != hipSuccess) return -1; \
        if (reset) {                                                                                   \
            unsigned long long z[16] = {0};                                                            \
            if (hipMemcpyToSymbol(HIP_SYMBOL(tg_trace), z, sizeof(z)) != hipSuccess) return -1;        \
        }                                                                                              \
        return 0;                                                                                      \
    }
#define TG_T(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define TG_ACC(slot, expr) do { if (threadIdx.x == 0) atomicAdd(&tg_trace[slot], (unsigned long long)(expr)); } while (0)
#else
#define TG_TRACE_DEFINE(READER)
#define TG_T(var)
#define TG_ACC(slot, expr)
#endif
