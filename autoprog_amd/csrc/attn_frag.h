// LDS tile addressing and MFMA fragment fetches shared by the attention kernels (mhsa.hip: sequences that fit one
// workgroup's LDS; mhsa_flash.hip: key/query-blocked kernels for long sequences and head_dim 48).
// Tiles are [tokens][HD] bf16 with HD = 32 (64-byte rows) or 64 (128-byte rows).
#pragma once
// Backward: p = exp2(s * c2 - lse * log2 e) is recomputed per (query, key).  A PADDED key (zero K row, s = 0) gets p = exp(-lse), which
// is harmless -- its dS meets zero K rows, its dK / dV rows are not stored -- until every real logit of the row is so negative that
// lse < -88: then exp(-lse) is inf and inf * 0 = NaN lands in dQ (seen at step ~150-250 of a run on one synthetic batch, once a head's
// logits had drifted to -100).  The exponent is clamped instead of masking by key index: real keys have s * c2 - lse * log2 e <= ~0.
#define ATT_PCAP 64.0f

#include "common.h"

// element offset of 16-B chunk `chunk` of row `row` in a [tokens][HD] LDS tile.
// HD=32 (64-B rows): chunk ^= f((row>>2)&3), f={0,2,3,1}; HD=64 (128-B rows): chunk ^= row&7.
// Both are conflict-free for the ds_read_b128 row reads and for the transposed reads used here.
template <int HD>
__device__ __forceinline__ int att_off(int row, int chunk) {
    if (HD == 32) {
        const int f = (0x78 >> (((row >> 2) & 3) << 1)) & 3;
        return row * 32 + ((chunk ^ f) << 3);
    }
    return row * 64 + ((chunk ^ (row & 7)) << 3);
}
template <int HD>
__device__ __forceinline__ bf16x8 att_row_frag(const bf16_t* tile, int row0, int lane, int kc = 0) {
    // operand whose k axis is the head dim: tile row row0+(lane&15), k chunk kc*4 + lane>>4
    return __builtin_bit_cast(bf16x8, ld16(tile + att_off<HD>(row0 + (lane & 15), kc * 4 + (lane >> 4))));
}
template <int HD>
__device__ __forceinline__ bf16x8 att_tr_frag(const bf16_t* tile, int row0, int dt, int lane) {
    // operand whose k axis is the TOKEN axis in the accumulator-permuted order:
    // k = 8g+j  <->  token row0 + 16*(j>>2) + 4g + (j&3); column = dt*16 + (lane&15)
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r1 = row0 + 4 * g + q, r2 = r1 + 16;
    const int ch = 2 * dt + (p >> 1), e = (p & 1) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + att_off<HD>(r1, ch) + e));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + att_off<HD>(r2, ch) + e));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// Lane-constant parts of the two fragment address patterns, computed ONCE per kernel: the swizzle term
// only depends on lane bits (row offsets used in the loops are multiples of 16 resp. 32 tokens, which do
// not touch the swizzled row bits), so every read in the loops is base + compile-time/loop-linear offset.
// (rocprof: ~930 VALU instructions per 16-query tile before hoisting, most of them address arithmetic.)
template <int HD>
__device__ __forceinline__ int att_row_base(int lane, int kc) { return att_off<HD>(lane & 15, kc * 4 + (lane >> 4)); }
template <int HD>
__device__ __forceinline__ bf16x8 att_row_at(const bf16_t* tile, int base, int row0) {
    return __builtin_bit_cast(bf16x8, ld16(tile + base + row0 * HD));
}
template <int HD>
__device__ __forceinline__ int att_tr_base(int lane, int dt) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    return att_off<HD>(4 * g + q, 2 * dt + (p >> 1)) + (p & 1) * 4;
}
template <int HD>
__device__ __forceinline__ bf16x8 att_tr_at(const bf16_t* tile, int base, int row0) {
    const bf16_t* a1 = tile + base + row0 * HD;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1 + 16 * HD));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
    u32x4 v;
    v[0] = pack_bf2(a[0], a[1]); v[1] = pack_bf2(a[2], a[3]); v[2] = pack_bf2(b[0], b[1]); v[3] = pack_bf2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, v);
}


// key/query-blocked kernels for long sequences and head_dim 48 (mhsa_flash.hip)
int ap_mhsa_flash_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int B, int N, int heads, int hd, float scale, const float* out_row_scale,
                      hipStream_t s, unsigned char* out8 = nullptr, const float* q_scale = nullptr, float* q_amax = nullptr);
size_t ap_mhsa_flash_bwd_ws(int B, int N, int heads);
int ap_mhsa_flash_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int B, int N, int heads, int hd,
                      float scale, float* delta, hipStream_t s);
