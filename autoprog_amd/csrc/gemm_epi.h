// Epilogue shared by every bf16 MFMA GEMM of the hot path (gemm.hip, gemm_dma.h): bias, GELU (+ stored
// pre-activation), gelu'(h) factor, DropPath row scale, residual add, bf16 store -- on 8 consecutive output columns.
#pragma once
#include "common.h"

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// Patch addressing (non-overlapping k x k / stride k convolutions as GEMMs without a gathered copy: PatchEmbed.proj and
// Downsample, models/volo.py:368-372,383-396).  GEMM row m = (image row group, output column) and GEMM column kk = (kernel row,
// kernel column x channel) of an NHWC feature map sit at element offset
//     (m / group) * gstride + (m % group) * rstride  +  (kk / kseg) * kstride + (kk % kseg)
// group = output columns per image row, gstride = k image rows, rstride = k pixels, kseg = k pixels, kstride = one image row.
// The divisions are multiplications by ceil(2^32 / d) (exact for m < 2^32 / d).
struct PatchMap {
    int group, gstride, rstride, kseg, kstride;
    unsigned gmagic, kmagic;
};
__device__ __forceinline__ int64_t patch_row(const PatchMap& p, int m) {
    const int q = (int)__umulhi((unsigned)m, p.gmagic);
    return (int64_t)q * p.gstride + (int64_t)(m - q * p.group) * p.rstride;
}
__device__ __forceinline__ int patch_col(const PatchMap& p, int kk) {
    const int q = (int)__umulhi((unsigned)kk, p.kmagic);
    return q * p.kstride + (kk - q * p.kseg);
}

// relu(bn(.)) applied to an operand while it is staged (conv.hip PRE_BN; round 5: the A rows of the patch GEMM and the B rows of its weight
// gradient -- PatchEmbed.proj reads the PRE-BatchNorm output of the last stem convolution, models/volo.py:364-372)
struct BnIn { const float* mean; const float* rstd; const float* gamma; const float* beta; };
__device__ __forceinline__ void bn_in_consts(const BnIn& bn, int c8, float* sc, float* sh) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = bn.rstd[c8 + k] * bn.gamma[c8 + k];
        sh[k] = bn.beta[c8 + k] - bn.mean[c8 + k] * sc[k];
    }
}
__device__ __forceinline__ u32x4 bn_in_apply(const u32x4& v, const float* sc, const float* sh) {
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = fmaxf(fmaf(f[k], sc[k], sh[k]), 0.f);
    return pack8(f);
}


struct EpiArgs {
    const float* bias; int gelu; bf16_t* preact; const bf16_t* dgelu_of; const float* row_scale;
    int rows_per_scale; const bf16_t* residual; int ldr; int dbg; unsigned long long* stamps;
    PatchMap pm;                // used by the PATCH instantiations of k_gemm_nt only
    const float* dq_a; const float* dq_b;     // fp8 instantiation: device scalars, accumulators are multiplied by dq_a[0] * dq_b[0] first
    const bf16_t* mul_by;       // out = v * mul_by[m,n] (ld = ldc): the stored activation derivative of a gelu = 2 forward
    unsigned char* q8; const float* q8_scale; float* q8_amax;     // fp8 GELU launches of the 8-phase kernel: the output a second time as e4m3
                                                                  // bytes [M, ldc], q8 = sat(out * q8_scale[0]), q8_amax[0] raised to max |out|
    const unsigned char* mul8;  // out = v * gq_decode(mul8[m,n]) (ld = ldc bytes): the 8-bit derivative codes of a gelu = 3 forward
    const unsigned* gelu_tab;   // gelu = 3 launches of the 8-phase kernel: the 4096-entry table of gq_tab_entry() below (global memory), or nullptr
    BnIn abn;                   // PATCH = 1 only: abn.mean != nullptr -> the A rows (patches of an NHWC map with 64 channels) are relu(bn(.)) of what is read
};

// 8-bit fixed-point code of gelu'(h) in [-0.1290, 1.1290] (gelu = 3 / mul_by8, include/autoprog_hip.h): code = clamp(rint(202 g) + 26, 0, 255),
// g = (code - 26) / 202.  0, 1/2 and 1 are exact; |error| <= 1 / 404 = the spacing of bf16 numbers in [1/2, 1).
#define GQ_SCALE 202.0f
#define GQ_ZERO 26.0f
__device__ __forceinline__ unsigned gq_code(float g) {
    return (unsigned)fminf(fmaxf(__builtin_rintf(fmaf(g, GQ_SCALE, GQ_ZERO)), 0.f), 255.f);
}
// four codes into a dword: v_cvt_pk_u8_f32 rounds to nearest even, saturates to [0, 255] and places the byte (tools/probe/cvt_u8.hip on
// gfx950: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2, 254.5 -> 254, 300 -> 255, -3 -> 0) -- rint + clamp + pack in one instruction: fma + cvt_pk per element
__device__ __forceinline__ unsigned gq_pack4(const float* g) {
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[0], GQ_SCALE, GQ_ZERO), 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[1], GQ_SCALE, GQ_ZERO), 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[2], GQ_SCALE, GQ_ZERO), 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[3], GQ_SCALE, GQ_ZERO), 3, w);
    return w;
}
__device__ __forceinline__ float gq_decode(unsigned code) { return fmaf((float)code, 1.0f / GQ_SCALE, -GQ_ZERO / GQ_SCALE); }
// the four codes of a dword (v_cvt_f32_ubyte0..3 + one fma each)
__device__ __forceinline__ void gq_unpack4(unsigned w, float* f) {
    f[0] = gq_decode(w & 0xffu); f[1] = gq_decode((w >> 8) & 0xffu); f[2] = gq_decode((w >> 16) & 0xffu); f[3] = gq_decode(w >> 24);
}

// The GELU of a bf16 value is a function of 16 bits.  For |h| in [2^-12, 16) -- 16 exponents x 128 mantissas x 2 signs = 4096 values, 16 KB --
// the table holds, per value, Phi(h) (so that gelu(h) = h * Phi(h): one multiply) in the top 24 bits of an fp32 and the 8-bit code of
// gelu'(h) in the low byte.  Below 2^-12 the entry of 2^-12 serves (Phi = 0.5 + 1e-4, the code of 0.5), from 16 on the entry of 15.94
// (Phi = 1 or 0, codes of 1 and 0): a clamp of the index, no branch.  The row phase of the GELU launches (gemm8p.h) costs ~30 VALU slots
// per element in arithmetic (one v_rcp, one v_exp, the A&S polynomial, the derivative) and ~11 with the table in LDS.
#define GQ_TAB_LO (115 << 7)                  // bf16 bits of 2^-12
#define GQ_TAB_N 2048                         // entries per sign
__device__ __forceinline__ unsigned gq_tab_entry(unsigned idx) {
    const unsigned bits = ((GQ_TAB_LO + (idx & (GQ_TAB_N - 1))) | ((idx >> 11) << 15)) << 16;
    const float h = __uint_as_float(bits);
    float c, e;
    gelu_parts(h, c, e);
    const float gp = fmaf(h * 0.39894228040143268f, e, c);
    return ((__float_as_uint(c) + 0x80u) & 0xffffff00u) | gq_code(gp);
}
// the table in global memory (built on first use, csrc/gemm.hip); nullptr: AP_GELU_TABLE=0, or `st` is being captured before a first build
const unsigned* g8_gelu_table_ptr(hipStream_t st);
// table index of a bf16 bit pattern held in bits [SH, SH + 16) of w
template <int SH>
__device__ __forceinline__ int gq_tab_index(unsigned w) {
    const int mag = (int)((w >> SH) & 0x7fffu) - GQ_TAB_LO;
    return min(max(mag, 0), GQ_TAB_N - 1) | (int)((w >> (SH + 4)) & 0x800u);
}


// epilogue of 8 consecutive output columns [n, n+8) of row m held in v[] (fp32 accumulators):
// +bias; GELU (storing the pre-activation); * gelu'(h); * DropPath row scale; + residual; bf16 store.
// All global accesses are 16-byte when the chunk is complete and the leading dimensions allow it.
// `pre` (used when has_pre): the chunk's 16 bytes of ep.dgelu_of -- or, when that is null, of ep.residual -- already loaded by the
// caller (k_gemm_nt issues these loads under its last K step so that their latency is hidden behind the MFMAs).
__device__ __forceinline__ void epi_chunk(float* v, int m, int n, int N, int ldc, bool vec_ok, const EpiArgs& ep, bf16_t* __restrict__ C,
                                          u32x4 pre = u32x4{0u, 0u, 0u, 0u}, bool has_pre = false) {
    const int nval = min(8, N - n);
    const bool full = (nval == 8) && vec_ok;
    if (ep.bias) {
        if (nval == 8) {
            const float4 b0 = *reinterpret_cast<const float4*>(ep.bias + n), b1 = *reinterpret_cast<const float4*>(ep.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < nval) v[q] += ep.bias[n + q];
        }
    }
    if (ep.gelu) {
        if (ep.preact) {
            bf16_t* p = ep.preact + (int64_t)m * ldc + n;
            // the activation is applied to the ROUNDED pre-activation so that backward (which reads the
            // stored bf16 h, or the derivative stored here) differentiates exactly what forward computed
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = bf2f(f2bf(v[q]));
            float sv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) sv[q] = ep.gelu >= 2 ? gelu_erf_grad(v[q]) : v[q];
            if (ep.gelu == 3) {                                       // 8-bit derivative codes, [M, ldc] bytes
                unsigned char* p8 = reinterpret_cast<unsigned char*>(ep.preact) + (int64_t)m * ldc + n;
                if (full) { u32x2 c; c[0] = gq_pack4(sv); c[1] = gq_pack4(sv + 4); *reinterpret_cast<u32x2*>(p8) = c; }
                else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q < nval) p8[q] = (unsigned char)gq_code(sv[q]);
                }
            } else if (full) st16(p, pack8(sv));
            else {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (q < nval) p[q] = f2bf(sv[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = gelu_erf(v[q]);
    }
    if (ep.dgelu_of || ep.mul_by) {
        const bf16_t* hp = (ep.dgelu_of ? ep.dgelu_of : ep.mul_by) + (int64_t)m * ldc + n;
        float h[8];
        if (has_pre) unpack8(pre, h);
        else if (full) unpack8(ld16(hp), h);
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = (q < nval) ? bf2f(hp[q]) : 0.f;
        }
        if (ep.dgelu_of) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= gelu_erf_grad(h[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= h[q];
        }
    }
    if (ep.mul8) {
        const unsigned char* hp = ep.mul8 + (int64_t)m * ldc + n;
        float h[8];
        if (full) { const u32x2 c = *reinterpret_cast<const u32x2*>(hp); gq_unpack4(c[0], h); gq_unpack4(c[1], h + 4); }
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = (q < nval) ? gq_decode(hp[q]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= h[q];
    }
    if (ep.row_scale) {
        const float rs = ep.row_scale[m / ep.rows_per_scale];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= rs;
    }
    if (ep.residual) {
        const bf16_t* rp = ep.residual + (int64_t)m * ep.ldr + n;
        float h[8];
        if (has_pre && !ep.dgelu_of && !ep.mul_by) unpack8(pre, h);
        else if (full) unpack8(ld16(rp), h);
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = (q < nval) ? bf2f(rp[q]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += h[q];
    }
    bf16_t* cp = C + (int64_t)m * ldc + n;
    if (full) st16(cp, pack8(v));
    else {
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < nval) cp[q] = f2bf(v[q]);
    }
}

__device__ __forceinline__ int key_a(int r) { return r & 7; }
// B-tile swizzle key for the N-permuted fragment rows (see "direct epilogue" below): the 16 rows one
// ds_read_b128 lane group touches are 8q + 4b + p (q = 0..3, p = 0..3) -> keys p | (q&1)<<2 are distinct
__device__ __forceinline__ int key_b(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

