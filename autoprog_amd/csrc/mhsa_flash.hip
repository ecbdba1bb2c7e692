// Key/query-blocked ("flash") multi-head self-attention for sequences that do not fit one workgroup's LDS and for
// head_dim 48:  softmax(q k^T * scale) v  (models/volo.py:188-197) and its backward.  This is the path of
// VOLO-D4/D5 (16 heads of 48, models/volo.py:776-821) and of 448-px inputs (N = 784 tokens); the LDS-resident kernels of
// mhsa.hip stay the path of N <= 256 with head_dim 32 / 64.
//
// Same MFMA formulation as mhsa.hip (v_mfma_f32_16x16x32_bf16, S^T = K.Q^T so a query's scores sit in 4 lanes, P feeds the
// PV product as the A operand, V / Q / dO fetched with ds_read_b64_tr_b16), but the other operand is STREAMED through a
// double-buffered LDS block of 64 tokens while the wave's own 16 tokens stay in registers:
//   forward   workgroup = 128 queries of one (image, head); keys/values in blocks of 64; online softmax (running max and sum,
//             accumulator rescaled per block)
//   backward  k_attn_delta   delta = rowsum(dO * O)
//             k_..._bwd_kv   workgroup = 128 keys; queries / dO / lse / delta streamed in blocks of 64 -> dK, dV
//             k_..._bwd_q    workgroup = 128 queries; keys / values streamed                       -> dQ
//             (two reduction-free kernels instead of atomics: every output row is owned by exactly one wave)
// head_dim 48 uses 128-byte LDS rows whose last 16 columns are zero (the QK^T reduction runs over 64) and 3 output fragments.
#include "common.h"
#include "attn_frag.h"

#define FB 64            // streamed tokens per block
#define FW 128           // tokens owned by a workgroup (8 waves x 16)

// global -> registers -> LDS staging of one 64-token block of NT_ token-major operands (row stride ld[i] elements, `hd` valid
// columns, rows >= N and columns >= hd read as zero).  NLD chunks per thread and operand.
template <int HDP, int NT_>
struct BlockStage {
    static constexpr int CPRW = HDP / 8;                       // 16-byte chunks per LDS row
    static constexpr int CH_T = FB * CPRW;                     // chunks per tile
    static constexpr int NLD = (NT_ * CH_T + 511) / 512;       // chunks per thread
    u32x4 v[NLD];
    __device__ __forceinline__ void load(const bf16_t* const* src, const int64_t* ld, int row0, int N, int hd) {
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 512;
            const int t = idx / CH_T, rem = idx - t * CH_T;
            const int row = rem / CPRW, c = rem - row * CPRW;
            const bool ok = (idx < NT_ * CH_T) && (row0 + row < N) && (c * 8 < hd);
            v[i] = ok ? ld16(src[t < NT_ ? t : 0] + (int64_t)(row0 + row) * ld[t < NT_ ? t : 0] + c * 8) : zero4;
        }
    }
    __device__ __forceinline__ void store(bf16_t* const* tiles) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 512;
            if (idx < NT_ * CH_T) {
                const int t = idx / CH_T, rem = idx - t * CH_T;
                const int row = rem / CPRW, c = rem - row * CPRW;
                st16(tiles[t] + att_off<HDP>(row, c), v[i]);
            }
        }
    }
};

// the wave's own 16 tokens as an MFMA operand straight from global memory (k = head dim, zero beyond hd)
template <int KC>
__device__ __forceinline__ void own_frag(bf16x8* f, const bf16_t* base, int64_t ld, int row, int hd, int g) {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const int d = kc * 32 + g * 8;
        f[kc] = __builtin_bit_cast(bf16x8, d < hd ? ld16(base + (int64_t)row * ld + d) : zero4);
    }
}

// (launch bounds: six waves per SIMD = THREE workgroups per CU -- the forward fits 79 registers without a spill (88 unhinted, two
// workgroups): 93 -> 88 us on the D5 layer (B = 16, 784 tokens, 16 heads of 48).  The kernels are bound by each wave's own chain
// S = K Q^T -> softmax -> P V, which only other waves fill: the opposite direction -- FOUR waves of two 16-token tiles each, every
// streamed fragment read once for both tiles, half the LDS reads per FLOP -- measured 93 -> 116 us forward and 256 -> 293 us backward
// (151 - 168 registers, two to three waves per SIMD); the same hint on the dQ kernel (100 -> 80 registers, 6 spilled) gave nothing.)
template <int HDP, int DT>
__global__ void __launch_bounds__(512, 6)
k_mhsa_flash_fwd(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int N, int heads, int hd, float scale, int nqb,
                 const float* __restrict__ out_row_scale, unsigned char* __restrict__ out8 = nullptr, const float* __restrict__ q_scale = nullptr,
                 float* __restrict__ q_amax = nullptr) {
    // out8 (nullable): the output a second time as OCP e4m3 bytes, out8 = sat(bf16(out) * q_scale[0]), q_amax[0] raised to max |bf16(out)|:
    // the operand of the fp8 output projection without a quantisation pass (ap_mhsa_fwd_fp8)
    extern __shared__ __attribute__((aligned(16))) bf16_t fsm[];
    constexpr int KC = HDP / 32;
    const int wg = blockIdx.x;
    const int bh = wg / nqb, qb = wg - bh * nqb;
    const int b = bh / heads, h = bh - b * heads;
    const int C = heads * hd;
    const int64_t ld = 3 * C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const int q0 = qb * FW + wave * 16;
    const bool active = q0 < N;                                      // wave-uniform; inactive waves only help staging
    const float c2 = scale * 1.4426950408889634f;
    int kb[KC], vb[DT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kb[kc] = att_row_base<HDP>(lane, kc);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vb[dt] = att_tr_base<HDP>(lane, dt);
    bf16x8 qf[KC];
    own_frag<KC>(qf, base, ld, min(q0 + fr, N - 1), hd, g);
    const bf16_t* srcs[2] = {base + C, base + 2 * C};
    const int64_t lds_[2] = {ld, ld};
    BlockStage<HDP, 2> st;
    st.load(srcs, lds_, 0, N, hd);
    float m = -1.0e30f, l = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nkb = (N + FB - 1) / FB;
    for (int blk = 0; blk < nkb; ++blk) {
        bf16_t* tiles[2] = {fsm + (blk & 1) * 2 * FB * HDP, fsm + (blk & 1) * 2 * FB * HDP + FB * HDP};   // K | V block
        st.store(tiles);
        __syncthreads();                                             // block `blk` visible; the other buffer is free again
        if (blk + 1 < nkb) st.load(srcs, lds_, (blk + 1) * FB, N, hd);
        if (!active) continue;
        f32x4 s[4];
        float bm = -1.0e30f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HDP>(tiles[0], kb[kc], t * 16), qf[kc], s[t], 0, 0, 0);
            if (blk == nkb - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (blk * FB + t * 16 + 4 * g + r >= N) s[t][r] = -1.0e30f;
            }
            bm = fmaxf(bm, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * c2);
        const float nmx = -mn * c2;
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, nmx)); ps += s[t][r]; }
        l = fmaf(l, alpha, ps);                                      // per-lane partial sum; the 4 lanes of a query meet at the end
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, 4 * g + r, 64);           // accumulator rows are queries 4g+r
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt][r] *= ar;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack_frag(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HDP>(tiles[1], vb[dt], 32 * s2), o[dt], 0, 0, 0);
        }
    }
    if (!active) return;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = (out_row_scale ? out_row_scale[b] : 1.0f) / l;
    if (g == 0 && q0 + fr < N) lse[((int64_t)b * heads + h) * N + q0 + fr] = (m * c2 + log2f(l)) * 0.6931471805599453f;
    const float qs = out8 ? q_scale[0] : 1.f;
    float qmx = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float ir = __shfl(inv, 4 * g + r, 64);
        const int q = q0 + 4 * g + r;
        if (q < N) {
            bf16_t* op = out + ((int64_t)b * N + q) * C + h * hd + fr;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16_t ob = f2bf(o[dt][r] * ir);
                op[dt * 16] = ob;
                if (out8) {
                    const float v = bf2f(ob);
                    qmx = fmaxf(qmx, fabsf(v));
                    const float c = fminf(fmaxf(v * qs, -448.f), 448.f);
                    out8[((int64_t)b * N + q) * C + h * hd + fr + dt * 16] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(c, c, 0, false) & 0xff);
                }
            }
        }
    }
    if (out8 && q_amax) {
        qmx = group_max<64>(qmx);
        if ((threadIdx.x & 63) == 0 && __float_as_int(qmx) > *reinterpret_cast<volatile int*>(q_amax))
            atomicMax(reinterpret_cast<int*>(q_amax), __float_as_int(qmx));
    }
}

// delta[b, h, n] = sum_d O[b, n, h*hd + d] * dO[b, n, h*hd + d]
__global__ void __launch_bounds__(256)
k_attn_delta(const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, float* __restrict__ delta, int64_t rows, int N, int heads, int hd) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (row, head)
    if (idx >= rows * heads) return;
    const int64_t row = idx / heads;
    const int h = (int)(idx - row * heads);
    const bf16_t* a = out + row * (int64_t)(heads * hd) + h * hd;
    const bf16_t* d = dout + row * (int64_t)(heads * hd) + h * hd;
    float acc = 0.f;
    for (int c = 0; c < hd; c += 8) {
        float x[8], y[8];
        unpack8(ld16(a + c), x);
        unpack8(ld16(d + c), y);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += x[k] * y[k];
    }
    const int64_t b = row / N, n = row - b * N;
    delta[(b * heads + h) * N + n] = acc;
}

template <int HDP, int DT>
__global__ void __launch_bounds__(512)
k_mhsa_flash_bwd_kv(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
                    bf16_t* __restrict__ dqkv, int N, int heads, int hd, float scale, int nkb) {
    extern __shared__ __attribute__((aligned(16))) bf16_t fsm[];
    constexpr int KC = HDP / 32;
    float* stats = reinterpret_cast<float*>(fsm + 4 * FB * HDP);       // [2 buffers][lse | delta][FB]
    const int wg = blockIdx.x;
    const int bh = wg / nkb, kblk = wg - bh * nkb;
    const int b = bh / heads, h = bh - b * heads;
    const int C = heads * hd;
    const int64_t ld = 3 * C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * hd;
    const bf16_t* gbase = dout + (int64_t)b * N * C + h * hd;
    const float* lrow = lse + ((int64_t)b * heads + h) * N;
    const float* drow = delta + ((int64_t)b * heads + h) * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const int k0 = kblk * FW + wave * 16;
    const bool active = k0 < N;
    const float c2 = scale * 1.4426950408889634f;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    int rb[KC], tb[DT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) rb[kc] = att_row_base<HDP>(lane, kc);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) tb[dt] = att_tr_base<HDP>(lane, dt);
    bf16x8 kf[KC], vf[KC];
    own_frag<KC>(kf, base + C, ld, min(k0 + fr, N - 1), hd, g);
    own_frag<KC>(vf, base + 2 * C, ld, min(k0 + fr, N - 1), hd, g);
    const bf16_t* srcs[2] = {base, gbase};
    const int64_t lds_[2] = {ld, (int64_t)C};
    BlockStage<HDP, 2> st;
    st.load(srcs, lds_, 0, N, hd);
    float nl = 0.f, nd = 0.f;                                         // staged lse / delta of this thread's row (threads 0..63)
    auto load_stats = [&](int row0) {
        if (threadIdx.x < FB) {
            const int row = row0 + threadIdx.x;
            nl = row < N ? lrow[row] * 1.4426950408889634f : 1.0e30f;    // padded queries: p = exp2(s - huge) = 0
            nd = row < N ? drow[row] : 0.f;
        }
    };
    load_stats(0);
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt] = z; dv[dt] = z; }
    const int nqb = (N + FB - 1) / FB;
    for (int blk = 0; blk < nqb; ++blk) {
        bf16_t* tiles[2] = {fsm + (blk & 1) * 2 * FB * HDP, fsm + (blk & 1) * 2 * FB * HDP + FB * HDP};   // Q | dO block
        float* flb = stats + (blk & 1) * 2 * FB;
        float* fdb = flb + FB;
        st.store(tiles);
        if (threadIdx.x < FB) { flb[threadIdx.x] = nl; fdb[threadIdx.x] = nd; }
        __syncthreads();
        if (blk + 1 < nqb) { st.load(srcs, lds_, (blk + 1) * FB, N, hd); load_stats((blk + 1) * FB); }
        if (!active) continue;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int q0 = (2 * qs + hf) * 16;
                f32x4 sc = z, dp = z;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HDP>(tiles[0], rb[kc], q0), kf[kc], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HDP>(tiles[1], rb[kc], q0), vf[kc], dp, 0, 0, 0);
                }
                const f32x4 fl4 = *reinterpret_cast<const f32x4*>(flb + q0 + 4 * g);
                const f32x4 fd4 = *reinterpret_cast<const f32x4*>(fdb + q0 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fminf(fmaf(sc[r], c2, -fl4[r]), ATT_PCAP));
                    p[hf][r] = pv;
                    ds[hf][r] = pv * (dp[r] - fd4[r]);
                }
            }
            const bf16x8 pf = pack_frag(p[0], p[1]);
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HDP>(tiles[1], tb[dt], 32 * qs), dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, att_tr_at<HDP>(tiles[0], tb[dt], 32 * qs), dk[dt], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    bf16_t* dbase = dqkv + (int64_t)b * N * ld + h * hd;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = k0 + 4 * g + r;
        if (key < N) {
            bf16_t* kp = dbase + (int64_t)key * ld + C + fr;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { kp[dt * 16] = f2bf(dk[dt][r] * scale); kp[C + dt * 16] = f2bf(dv[dt][r]); }
        }
    }
}

template <int HDP, int DT>
__global__ void __launch_bounds__(512)
k_mhsa_flash_bwd_q(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
                   bf16_t* __restrict__ dqkv, int N, int heads, int hd, float scale, int nqb) {
    extern __shared__ __attribute__((aligned(16))) bf16_t fsm[];
    constexpr int KC = HDP / 32;
    const int wg = blockIdx.x;
    const int bh = wg / nqb, qb = wg - bh * nqb;
    const int b = bh / heads, h = bh - b * heads;
    const int C = heads * hd;
    const int64_t ld = 3 * C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * hd;
    const bf16_t* gbase = dout + (int64_t)b * N * C + h * hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const int q0 = qb * FW + wave * 16;
    const bool active = q0 < N;
    const float c2 = scale * 1.4426950408889634f;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    int rb[KC], tb[DT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) rb[kc] = att_row_base<HDP>(lane, kc);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) tb[dt] = att_tr_base<HDP>(lane, dt);
    bf16x8 qf[KC], gf[KC];
    const int qrow = min(q0 + fr, N - 1);
    own_frag<KC>(qf, base, ld, qrow, hd, g);
    own_frag<KC>(gf, gbase, (int64_t)C, qrow, hd, g);
    const int64_t sidx = ((int64_t)b * heads + h) * N + qrow;
    const float flq = (q0 + fr < N) ? lse[sidx] * 1.4426950408889634f : 1.0e30f;
    const float fdq = (q0 + fr < N) ? delta[sidx] : 0.f;
    const bf16_t* srcs[2] = {base + C, base + 2 * C};
    const int64_t lds_[2] = {ld, ld};
    BlockStage<HDP, 2> st;
    st.load(srcs, lds_, 0, N, hd);
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = z;
    const int nkb = (N + FB - 1) / FB;
    for (int blk = 0; blk < nkb; ++blk) {
        bf16_t* tiles[2] = {fsm + (blk & 1) * 2 * FB * HDP, fsm + (blk & 1) * 2 * FB * HDP + FB * HDP};   // K | V block
        st.store(tiles);
        __syncthreads();
        if (blk + 1 < nkb) st.load(srcs, lds_, (blk + 1) * FB, N, hd);
        if (!active) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 ds[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kk = (2 * ks + hf) * 16;
                f32x4 sc = z, dp = z;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HDP>(tiles[0], rb[kc], kk), qf[kc], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HDP>(tiles[1], rb[kc], kk), gf[kc], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded keys need no mask: their K rows are zero in LDS, so they add nothing to dQ
                    const float pv = __builtin_amdgcn_exp2f(fminf(fmaf(sc[r], c2, -flq), ATT_PCAP));
                    ds[hf][r] = pv * (dp[r] - fdq);
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, att_tr_at<HDP>(tiles[0], tb[dt], 32 * ks), dq[dt], 0, 0, 0);
        }
    }
    if (!active) return;
    bf16_t* dbase = dqkv + (int64_t)b * N * ld + h * hd;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        if (q < N) {
            bf16_t* qp = dbase + (int64_t)q * ld + fr;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) qp[dt * 16] = f2bf(dq[dt][r] * scale);
        }
    }
}

#define FLASH_DISPATCH(KERNEL, ...)                                                                                  \
    if (hd == 32) hipLaunchKernelGGL((KERNEL<32, 2>), grid, dim3(512), lds, s, __VA_ARGS__);                           \
    else if (hd == 48) hipLaunchKernelGGL((KERNEL<64, 3>), grid, dim3(512), lds, s, __VA_ARGS__);                      \
    else hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(512), lds, s, __VA_ARGS__);

int ap_mhsa_flash_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int B, int N, int heads, int hd, float scale, const float* out_row_scale,
                      hipStream_t s, unsigned char* out8, const float* q_scale, float* q_amax) {
    const int nqb = (N + FW - 1) / FW;
    const int hdp = hd == 32 ? 32 : 64;
    const dim3 grid((unsigned)((int64_t)B * heads * nqb));
    const size_t lds = (size_t)4 * FB * hdp * sizeof(bf16_t);
    (void)hipGetLastError();
    FLASH_DISPATCH(k_mhsa_flash_fwd, qkv, out, lse, N, heads, hd, scale, nqb, out_row_scale, out8, q_scale, q_amax)
    return ap_check_launch();
}

size_t ap_mhsa_flash_bwd_ws(int B, int N, int heads) { return (size_t)B * heads * N * sizeof(float); }

int ap_mhsa_flash_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int B, int N, int heads, int hd,
                      float scale, float* delta, hipStream_t s) {
    const int64_t rows = (int64_t)B * N;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_attn_delta, dim3((unsigned)((rows * heads + 255) / 256)), dim3(256), 0, s, out, dout, delta, rows, N, heads, hd);
    const int nb = (N + FW - 1) / FW;
    const int hdp = hd == 32 ? 32 : 64;
    const dim3 grid((unsigned)((int64_t)B * heads * nb));
    {
        const size_t lds = (size_t)4 * FB * hdp * sizeof(bf16_t) + (size_t)4 * FB * sizeof(float);
        FLASH_DISPATCH(k_mhsa_flash_bwd_kv, qkv, dout, lse, delta, dqkv, N, heads, hd, scale, nb)
    }
    {
        const size_t lds = (size_t)4 * FB * hdp * sizeof(bf16_t);
        FLASH_DISPATCH(k_mhsa_flash_bwd_q, qkv, dout, lse, delta, dqkv, N, heads, hd, scale, nb)
    }
    return ap_check_launch();
}
