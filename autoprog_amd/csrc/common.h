// Shared device helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/autoprog_hip.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define AP_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(bf16_t, b); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi; return __builtin_bit_cast(unsigned, v);
}
// 8 bf16 (one 16-byte chunk) <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(v[i]); f[2 * i + 1] = bf_hi(v[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
// the same for a GLOBAL output that this kernel does not read again: nontemporal (no L2 allocation).  The consumer is the next kernel,
// behind an L2 write-back either way; on the GEMM epilogues alone this was 13.36 -> 12.96 ms per step (AP_NT_STORES=0 at build time: plain)
#ifndef AP_NT_STORES
#define AP_NT_STORES 1
#endif
__device__ __forceinline__ void st16_nt(void* p, const u32x4& v) {
    if (AP_NT_STORES) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = v;
}
__device__ __forceinline__ void st16f_nt(float* p, const f32x4& v) {
    if (AP_NT_STORES) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = v;
}

// exact-erf GELU (nn.GELU default) and its derivative.  erf is evaluated with Abramowitz-Stegun
// 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-level) on z = |x|/sqrt(2): one v_exp + one v_rcp + 5 FMAs;
// the same exponential exp(-x^2/2) also gives the Gaussian pdf needed by the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __expf(-0.5f * x * x);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_tail = 0.5f * poly * t * e;          // 0.5*erfc(z)
    cdf = x >= 0.f ? 1.0f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_erf(float x) { float c, e; gelu_parts(x, c, e); return x * c; }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float c, e; gelu_parts(x, c, e);
    return fmaf(x * 0.39894228040143268f, e, c);
}

template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a linear workgroup id (MI355X deals workgroups round-robin over its 8 XCDs,
// each with a private L2): ids that share an XCD (id % 8) are mapped to a CONTIGUOUS range of work items, so
// neighbouring items (same image / same operand panel) hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + k;
}

static inline int ap_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AP_OK : AP_ERR_LAUNCH;
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
