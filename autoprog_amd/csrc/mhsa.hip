// Fused multi-head self-attention core for short sequences (N <= 256 tokens, head_dim 32):
//   softmax(q k^T * scale) v   (models/volo.py:188-197) and its backward.
// One workgroup (4 waves) per (image, head); K/V (and Q/dO in backward) of that head live in LDS
// for the whole kernel, so qkv is read once from HBM (HBM-bound: ~3*N*32*2 B in, N*32*2 B out).
//
// MFMA v_mfma_f32_16x16x32_bf16 (k = head_dim = 32: one instruction per 16x16 score tile):
//   forward : S^T tile = K_tile . Q_tile^T  -> lane holds 4 keys x 1 query per tile, the whole
//             score row of a query is spread over the 4 lanes (l, l+16, l+32, l+48): softmax
//             max/sum = in-register + 2 shuffles.  P then feeds the PV product directly as the
//             MFMA A operand (k order permuted, hardware-probe-verified), V is fetched with
//             ds_read_b64_tr_b16 from the row-major LDS image (no transposed copy).
//   backward: two passes over the LDS-resident head, both reduction-free across waves:
//             pass A, wave owns 16 keys  : dV += P^T dO, dK += dS^T Q   (S,dP recomputed)
//             pass B, wave owns 16 queries: dQ += dS K                  (S^T,dP^T recomputed)
// LDS tiles are [tokens][32] bf16 (64-byte rows) with the 16-byte chunk index XORed by
// f((row>>2)&3), f = {0,2,3,1}: conflict-free for both the ds_read_b128 row reads and the
// transposed reads.
#include "common.h"
#include "attn_frag.h"
#include <cstdlib>


// Stage NTILES token-major operands ([N, HD] slices with row stride ld[i]) into their LDS tiles.  ALL global loads of a thread
// (up to MAXIT iterations x NTILES tiles) are issued before the first LDS store: the former one-load-then-store loop, called
// once per tile, exposed a full memory latency per iteration (8 of them in the backward kernel).
template <int HD, int NTILES, int MAXIT>
__device__ __forceinline__ void att_stage_n(bf16_t* const* tiles, const bf16_t* const* srcs, const int64_t* lds, int N, int Npad) {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    constexpr int CH = HD / 8;
    const int total = Npad * CH;
    for (int idx0 = threadIdx.x; idx0 < total; idx0 += MAXIT * blockDim.x) {
        u32x4 v[MAXIT][NTILES];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = idx0 + it * blockDim.x;
            const int row = idx / CH, c = idx % CH;
#pragma unroll
            for (int t = 0; t < NTILES; ++t) v[it][t] = (idx < total && row < N) ? ld16(srcs[t] + (int64_t)row * lds[t] + c * 8) : zero4;
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = idx0 + it * blockDim.x;
            if (idx < total) {
                const int row = idx / CH, c = idx % CH;
#pragma unroll
                for (int t = 0; t < NTILES; ++t) st16(tiles[t] + att_off<HD>(row, c), v[it][t]);
            }
        }
    }
}

template <int NT, int HD>
__global__ void __launch_bounds__(256)
k_mhsa_fwd(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int N, int heads, float scale,
           const float* __restrict__ out_row_scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    constexpr int Npad = NT * 16;
    constexpr int KC = HD / 32, DT = HD / 16;
    bf16_t* Ks = smem;
    bf16_t* Vs = smem + Npad * HD;
    // heads of one image share 128-B lines of qkv (64 B per token and head at head_dim 32): keep them on one XCD
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / heads, h = wg % heads;
    const int C = heads * HD;
    const int64_t ld = 3 * C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    {
        bf16_t* tiles[2] = {Ks, Vs};
        const bf16_t* srcs[2] = {base + C, base + 2 * C};
        const int64_t strides[2] = {ld, ld};
        att_stage_n<HD, 2, 4>(tiles, srcs, strides, N, Npad);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const float c2 = scale * 1.4426950408889634f;
    const int nq = (N + 15) >> 4;
    int kb[KC], vb[DT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kb[kc] = att_row_base<HD>(lane, kc);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vb[dt] = att_tr_base<HD>(lane, dt);
    // the Q fragment of the NEXT query tile is fetched from global memory while the current tile is computed
    bf16x8 qnext[KC];
    {
        const int qrow0 = min(wave * 16 + fr, N - 1);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) qnext[kc] = __builtin_bit_cast(bf16x8, ld16(base + (int64_t)qrow0 * ld + kc * 32 + g * 8));
    }
    for (int qt = wave; qt < nq; qt += 4) {
        bf16x8 qf[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) qf[kc] = qnext[kc];
        {
            const int qrow = min((qt + 4) * 16 + fr, N - 1);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) qnext[kc] = __builtin_bit_cast(bf16x8, ld16(base + (int64_t)qrow * ld + kc * 32 + g * 8));
        }
        f32x4 s[NT];
        float mx = -1.0e30f;                         // max of the RAW scores (scale > 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Ks, kb[kc], t * 16), qf[kc], s[t], 0, 0, 0);
            if (t >= NT - 2) {                       // N > 16*(NT-2): only the last two key tiles can hold padding
#pragma unroll
                for (int r = 0; r < 4; ++r) if (t * 16 + 4 * g + r >= N) s[t][r] = -1.0e30f;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmx = -mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, nmx)); sum += s[t][r]; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            const bf16x8 pf = pack_frag(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HD>(Vs, vb[dt], 32 * s2), o[dt], 0, 0, 0);
        }
        const float inv = (out_row_scale ? out_row_scale[b] : 1.0f) / sum;      // 0/1 DropPath keep mask of the projection that follows
        if (g == 0 && qt * 16 + fr < N) lse[((int64_t)b * heads + h) * N + qt * 16 + fr] = (mx * c2 + log2f(sum)) * 0.6931471805599453f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ir = __shfl(inv, 4 * g + r, 64);
            const int q = qt * 16 + 4 * g + r;
            if (q < N) {
                bf16_t* op = out + ((int64_t)b * N + q) * C + h * HD + fr;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) op[dt * 16] = f2bf(o[dt][r] * ir);
            }
        }
    }
}

// The same forward as a PERSISTENT kernel (head_dim 32, 7 .. 13 query tiles): 16 waves, one workgroup per CU walking the (image, head)
// items.  Wave w < nq owns query tile w of every item (one tile per wave: no 13-tiles-on-4-waves rounding); waves 13 .. 15 are
// loaders: they fetch the NEXT item's K and V rows and write them to the other K/V buffer in LDS while the others compute, so an item
// starts on operands that are already there (k_mhsa_fwd stages K/V per workgroup and relies on co-resident workgroups for overlap).
// One LDS-only barrier per item (s_waitcnt lgkmcnt(0); s_barrier): nothing waits for the Q prefetch or the output stores.
#define FP_LOADERS 3
template <int NT>
__global__ void __launch_bounds__(1024)
k_mhsa_fwd_p(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse, int N, int heads, float scale,
             const float* __restrict__ out_row_scale, int nitems) {
    constexpr int HD = 32, Npad = NT * 16, LT = 64 * FP_LOADERS, NPRE = (Npad * 4 + LT - 1) / LT;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];             // [2 buffers][K | V][Npad][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const int C = heads * HD;
    const int64_t ld = 3 * C;
    const int nq = (N + 15) >> 4;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define FP_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // heads of one image share 128-B lines of qkv (64 B per token and head): the items an XCD works on at a time are consecutive
    auto item_of = [&](int v) { return xcd_remap(v, nitems); };
    const int nround = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // items of this workgroup
    if (wave >= 16 - FP_LOADERS) {
        const int ltid = threadIdx.x - 64 * (16 - FP_LOADERS);
        auto fetch = [&](int item, bf16_t* Kd, bf16_t* Vd) {
            const int b = item / heads, h = item % heads;
            const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
            int lt = ltid;
            asm volatile("" : "+v"(lt));                                      // keeps the per-chunk offsets out of long-lived registers
            u32x4 pk[NPRE], pv[NPRE];
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int id = i * LT + lt, row = min(id >> 2, N - 1), c = id & 3;
                const unsigned rq = (unsigned)(row * 3 * C + c * 8);           // uniform base + 32-bit lane offset
                pk[i] = ld16(base + C + rq);
                pv[i] = ld16(base + 2 * C + rq);
            }
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int id = i * LT + lt, row = id >> 2, c = id & 3;
                if (id < Npad * 4) {
                    const bool ok = row < N;
                    const int off = att_off<HD>(row, c);
                    *reinterpret_cast<u32x4*>(Kd + off) = ok ? pk[i] : zero4;
                    *reinterpret_cast<u32x4*>(Vd + off) = ok ? pv[i] : zero4;
                }
            }
        };
        if (nround > 0) fetch(item_of(blockIdx.x), smem, smem + Npad * HD);
        for (int k = 0; k < nround; ++k) {
            FP_BAR();                                                          // buffer k & 1 is complete; everyone is done with the other one
            if (k + 1 < nround) {
                bf16_t* nb = smem + ((k + 1) & 1) * 2 * Npad * HD;
                fetch(item_of(blockIdx.x + (k + 1) * gridDim.x), nb, nb + Npad * HD);
            }
        }
        return;
    }
    const float c2 = scale * 1.4426950408889634f;
    const int kb = att_row_base<HD>(lane, 0);
    int vb[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vb[dt] = att_tr_base<HD>(lane, dt);
    const int qt = wave;
    const unsigned qoff = (unsigned)(min(qt * 16 + fr, N - 1) * 3 * C + g * 8);
    bf16x8 qnext = __builtin_bit_cast(bf16x8, zero4);
    if (qt < nq && nround > 0) {
        const int item = item_of(blockIdx.x);
        qnext = __builtin_bit_cast(bf16x8, ld16(qkv + (int64_t)(item / heads) * N * ld + (item % heads) * HD + qoff));
    }
    for (int k = 0; k < nround; ++k) {
        FP_BAR();
        if (qt >= nq) continue;
        const int item = item_of(blockIdx.x + k * gridDim.x);
        const int b = item / heads, h = item % heads;
        const bf16_t* Ks = smem + (k & 1) * 2 * Npad * HD;
        const bf16_t* Vs = Ks + Npad * HD;
        const bf16x8 qf = qnext;
        if (k + 1 < nround) {                                                  // the Q fragment of the next item: in flight during this one
            const int nx = item_of(blockIdx.x + (k + 1) * gridDim.x);
            qnext = __builtin_bit_cast(bf16x8, ld16(qkv + (int64_t)(nx / heads) * N * ld + (nx % heads) * HD + qoff));
        }
        f32x4 s[NT];
        float mx = -1.0e30f;                         // max of the RAW scores (scale > 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Ks, kb, t * 16), qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if (t >= NT - 2) {                       // N > 16*(NT-2): only the last two key tiles can hold padding
#pragma unroll
                for (int r = 0; r < 4; ++r) if (t * 16 + 4 * g + r >= N) s[t][r] = -1.0e30f;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmx = -mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, nmx)); sum += s[t][r]; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            const bf16x8 pf = pack_frag(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HD>(Vs, vb[dt], 32 * s2), o[dt], 0, 0, 0);
        }
        const float inv = (out_row_scale ? out_row_scale[b] : 1.0f) / sum;      // 0/1 DropPath keep mask of the projection that follows
        if (g == 0 && qt * 16 + fr < N) lse[((int64_t)b * heads + h) * N + qt * 16 + fr] = (mx * c2 + log2f(sum)) * 0.6931471805599453f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ir = __shfl(inv, 4 * g + r, 64);
            const int q = qt * 16 + 4 * g + r;
            if (q < N) {
                bf16_t* op = out + ((int64_t)b * N + q) * C + h * HD + fr;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) op[dt * 16] = f2bf(o[dt][r] * ir);
            }
        }
    }
#undef FP_BAR
}

template <int HD>
__global__ void __launch_bounds__(512)
k_mhsa_bwd(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
           const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int N, int heads, float scale, int NT) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    constexpr int KC = HD / 32, DT = HD / 16, CH = HD / 8;
    const int Npad = NT * 16;
    bf16_t* Qs = smem;
    bf16_t* Ks = Qs + Npad * HD;
    bf16_t* Vs = Ks + Npad * HD;
    bf16_t* Gs = Vs + Npad * HD;                                   // dO
    float* fl = reinterpret_cast<float*>(Gs + Npad * HD);          // lse * log2(e)   (+huge for padded rows)
    float* fd = fl + Npad;                                         // delta = rowsum(dO * O)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / heads, h = wg % heads;
    const int C = heads * HD;
    const int64_t ld = 3 * C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* obase = out + (int64_t)b * N * C + h * HD;
    const bf16_t* gbase = dout + (int64_t)b * N * C + h * HD;
    // delta = rowsum(dO * O) and the scaled lse.  The loads of the first sweep (the only one at head_dim 32) are issued BEFORE the
    // operand staging, so that the workgroup exposes ONE memory latency instead of two
    constexpr int DIT = 2;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 va0[DIT], vd0[DIT];
    float ls0[DIT];
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
        const int idx = threadIdx.x + it * 512;
        const int row = idx / CH, c = idx % CH;
        const bool ok = idx < Npad * CH && row < N;
        va0[it] = ok ? ld16(obase + (int64_t)row * C + c * 8) : zero4;
        vd0[it] = ok ? ld16(gbase + (int64_t)row * C + c * 8) : zero4;
        ls0[it] = (ok && c == 0) ? lse[((int64_t)b * heads + h) * N + row] : 0.f;
    }
    {
        bf16_t* tiles[4] = {Qs, Ks, Vs, Gs};
        const bf16_t* srcs[4] = {base, base + C, base + 2 * C, gbase};
        const int64_t strides[4] = {ld, ld, ld, (int64_t)C};
        att_stage_n<HD, 4, 2>(tiles, srcs, strides, N, Npad);
    }
    {
        auto reduce_rows = [&](int idx0, const u32x4* va, const u32x4* vd, const float* ls) {
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int idx = idx0 + it * 512;
                if (idx >= Npad * CH) continue;                                   // wave-uniform (whole waves)
                const int row = idx / CH, c = idx % CH;
                float a[8], d[8];
                unpack8(va[it], a);
                unpack8(vd[it], d);
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) part += a[k] * d[k];
#pragma unroll
                for (int o = 1; o < CH; o <<= 1) part += __shfl_xor(part, o, 64);
                if (c == 0) {
                    fd[row] = part;
                    fl[row] = (row < N) ? ls[it] * 1.4426950408889634f : 1.0e30f;
                }
            }
        };
        reduce_rows(threadIdx.x, va0, vd0, ls0);
        for (int idx0 = threadIdx.x + DIT * 512; idx0 < Npad * CH; idx0 += DIT * 512) {     // head_dim 64: a second sweep (Npad*CH is a multiple of 128: whole waves)
            u32x4 va[DIT], vd[DIT];
            float ls[DIT];
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int idx = idx0 + it * 512;
                const int row = idx / CH, c = idx % CH;
                const bool ok = idx < Npad * CH && row < N;
                va[it] = ok ? ld16(obase + (int64_t)row * C + c * 8) : zero4;
                vd[it] = ok ? ld16(gbase + (int64_t)row * C + c * 8) : zero4;
                ls[it] = (ok && c == 0) ? lse[((int64_t)b * heads + h) * N + row] : 0.f;
            }
            reduce_rows(idx0, va, vd, ls);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const float c2 = scale * 1.4426950408889634f;
    const int ntile = (N + 15) >> 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    bf16_t* dbase = dqkv + (int64_t)b * N * ld + h * HD;
    int rb[KC], tb[DT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) rb[kc] = att_row_base<HD>(lane, kc);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) tb[dt] = att_tr_base<HD>(lane, dt);

    // ---- pass A: this wave owns key tile jt -> dK, dV
    for (int jt = wave; jt < ntile; jt += 8) {
        bf16x8 kf[KC], vf[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) { kf[kc] = att_row_at<HD>(Ks, rb[kc], jt * 16); vf[kc] = att_row_at<HD>(Vs, rb[kc], jt * 16); }
        f32x4 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[dt] = z; dv[dt] = z; }
#pragma unroll 1
        for (int qs = 0; qs < NT / 2; ++qs) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int q0 = (2 * qs + hf) * 16;
                f32x4 sc = z, dp = z;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Qs, rb[kc], q0), kf[kc], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Gs, rb[kc], q0), vf[kc], dp, 0, 0, 0);
                }
                // the lane's 4 query rows q0+4g..+3 are consecutive: one 16-byte LDS read each for lse and delta
                const f32x4 fl4 = *reinterpret_cast<const f32x4*>(fl + q0 + 4 * g);
                const f32x4 fd4 = *reinterpret_cast<const f32x4*>(fd + q0 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fminf(fmaf(sc[r], c2, -fl4[r]), ATT_PCAP));
                    p[hf][r] = pv;
                    ds[hf][r] = pv * (dp[r] - fd4[r]);                // the softmax scale is applied once to dK / dQ
                }
            }
            const bf16x8 pf = pack_frag(p[0], p[1]);
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HD>(Gs, tb[dt], 32 * qs), dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, att_tr_at<HD>(Qs, tb[dt], 32 * qs), dk[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = jt * 16 + 4 * g + r;
            if (key < N) {
                bf16_t* kp = dbase + (int64_t)key * ld + C + fr;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { kp[dt * 16] = f2bf(dk[dt][r] * scale); kp[C + dt * 16] = f2bf(dv[dt][r]); }
            }
        }
    }
    // ---- pass B: this wave owns query tile qt -> dQ
    for (int qt = wave; qt < ntile; qt += 8) {
        bf16x8 qf[KC], gf[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) { qf[kc] = att_row_at<HD>(Qs, rb[kc], qt * 16); gf[kc] = att_row_at<HD>(Gs, rb[kc], qt * 16); }
        const float flq = fl[qt * 16 + fr], fdq = fd[qt * 16 + fr];
        f32x4 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = z;
#pragma unroll 1
        for (int ks = 0; ks < NT / 2; ++ks) {
            f32x4 ds[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int k0 = (2 * ks + hf) * 16;
                f32x4 sc = z, dp = z;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Ks, rb[kc], k0), qf[kc], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Vs, rb[kc], k0), gf[kc], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded keys: their K rows are zero, so they add nothing to dQ -- as long as p stays finite (ATT_PCAP, attn_frag.h)
                    const float pv = __builtin_amdgcn_exp2f(fminf(fmaf(sc[r], c2, -flq), ATT_PCAP));
                    ds[hf][r] = pv * (dp[r] - fdq);
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, att_tr_at<HD>(Ks, tb[dt], 32 * ks), dq[dt], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * g + r;
            if (q < N) {
                bf16_t* qp = dbase + (int64_t)q * ld + fr;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) qp[dt * 16] = f2bf(dq[dt][r] * scale);
            }
        }
    }
}

// ---- backward with dS handed over through LDS (head_dim 32, 7 .. 13 token tiles): the default for the 196 tokens of a 224 px image -----
// k_mhsa_bwd computes S, dP and the exponentials twice (key-owned for dK / dV, query-owned for dQ) and stages its operands once per
// workgroup with nothing of its own to overlap.  Here: 16 waves, one workgroup per CU, persistent over the (image, head) items.
//   waves 0 .. ntile-1 : phase 1 -- the key-owned pass of k_mhsa_bwd (wave = one key tile, all query pairs, fully unrolled: dK, dV),
//                        with -delta as the initial accumulator of dP, plus its dS block written to LDS as [key][query] (8-byte
//                        stores); barrier; phase 2 -- wave = one QUERY tile: dQ = dS K with dS read back transposed
//                        (ds_read_b64_tr_b16) in the key order of the K^T fragment.  No second softmax, no partial dQ sums across waves
//                        (a first one-pass version kept dQ partials of all query tiles in every wave: 112 registers and a
//                        cross-wave reduction, 117 us).
//   waves 13 .. 15     : loaders -- fetch the NEXT item's O, dO, lse (-> delta, into the other lse / delta buffer) and Q, K, V rows into
//                        registers while the others compute; Q, V, dO go to their LDS tiles behind the phase-1 barrier, K behind the
//                        item's last one: the operand latency of an item hides behind the previous item's arithmetic.
// LDS: four operand tiles (57 KB at N = 196) + 2 x (lse, delta) + dS^T [keys][232] (94 KB) = 155 KB.  The barriers are LDS-only
// (s_waitcnt lgkmcnt(0); s_barrier): nothing waits for the prefetch or for the result stores.
// B = 128, N = 196, 12 heads: 52 us against 66 us (k_mhsa_bwd); by phase: barriers + launch 10, arithmetic 31 (VALU-bound: 56 exp, fma,
// mul, cvt each per wave and item, on the SIMD that holds 4 of the 13 compute waves), exposed loader work + result stores 11.
#define DS_STR 232               // dS^T row stride (bf16): 224 queries + 8 of padding (464 B: conflict-free transposed reads)
#define DS_LOADERS 3
template <int NP>                // query-tile pairs (NT == 2 NP): the phase-1 loop is unrolled, every LDS address is a base + constant
__global__ void __launch_bounds__(1024)
k_mhsa_bwd_ds(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
              const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int N, int heads, float scale, int NT, int nitems) {
    constexpr int HD = 32, LT = 64 * DS_LOADERS, NPRE = 5;        // NPRE: 16-byte chunks per loader thread and tile (Npad * 4 <= 5 * 192)
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int Npad = NT * 16;
    bf16_t* Qs = smem;
    bf16_t* Ks = Qs + Npad * HD;
    bf16_t* Vs = Ks + Npad * HD;
    bf16_t* Gs = Vs + Npad * HD;
    float* flb = reinterpret_cast<float*>(Gs + Npad * HD);                    // [2 buffers][lse * log2 e | delta][Npad]
    bf16_t* dSt = reinterpret_cast<bf16_t*>(flb + 4 * Npad);                  // [ntile * 16 keys][DS_STR]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4, q4 = fr >> 2, p4 = fr & 3;
    const int C = heads * HD;
    const int64_t ld = 3 * C;
    const float c2 = scale * 1.4426950408889634f;
    const int ntile = (N + 15) >> 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const bool loader = wave >= 16 - DS_LOADERS;
    const int ltid = threadIdx.x - 64 * (16 - DS_LOADERS);

    // loader state: chunk i of tile t is (row, c) = ((i * LT + ltid) >> 2, ltid & 3); rows beyond N read row N-1 and become zeros in LDS
    // loader state: chunk i of a tile is (row, c) = ((i * LT + ltid) >> 2, ltid & 3); rows beyond N read row N-1 and become zeros in LDS.
    // Registers decide the shape: every spilled load destination makes the loader wait for that load before it can issue the next
    // batch (one memory latency per batch: 8 - 11 us per item with 150 - 220 B of scratch).  So: O, dO and lse first, delta and lse
    // go straight to the OTHER lse / delta buffer (the current one is being read), O is dropped, then Q, K, V follow: 80 registers.
    u32x4 pre[4][NPRE];
    auto issue = [&](int item, float* fl_n, float* fd_n) {
        const int b = item / heads, h = item % heads;
        const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
        const bf16_t* obase = out + (int64_t)b * N * C + h * HD;
        const bf16_t* gbase = dout + (int64_t)b * N * C + h * HD;
        const float* lbase = lse + ((int64_t)b * heads + h) * N;
        int lt = ltid;
        asm volatile("" : "+v"(lt));        // as in deposit(): keeps the per-chunk offsets out of long-lived registers
        {
            u32x4 po[NPRE];
            float pl[NPRE];
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int id = i * LT + lt, row = min(id >> 2, N - 1), c = id & 3;
                const unsigned ro = (unsigned)(row * C + c * 8);               // uniform base + 32-bit lane offset
                po[i] = ld16(obase + ro);
                pre[3][i] = ld16(gbase + ro);
                pl[i] = lbase[(unsigned)row];
            }
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int id = i * LT + lt, row = id >> 2, c = id & 3;
                float a8[8], d8[8];
                unpack8(po[i], a8);
                unpack8(pre[3][i], d8);
                float pt = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) pt += a8[k] * d8[k];
                pt += __shfl_xor(pt, 1, 64);                                   // the 4 chunks of a row sit in 4 consecutive lanes
                pt += __shfl_xor(pt, 2, 64);
                if (c == 0 && id < Npad * 4) {
                    const bool ok = row < N;
                    fd_n[row] = ok ? -pt : 0.f;                                // negated: the initial accumulator of dP
                    fl_n[row] = ok ? pl[i] * 1.4426950408889634f : 1.0e30f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int id = i * LT + lt, row = min(id >> 2, N - 1), c = id & 3;
            const unsigned rq = (unsigned)(row * 3 * C + c * 8);
            pre[0][i] = ld16(base + rq);
            pre[1][i] = ld16(base + C + rq);
            pre[2][i] = ld16(base + 2 * C + rq);
        }
    };
    auto deposit = [&](bool keys) {         // keys = false: Q, V, dO (free once phase 1 is over); true: K (read by phase 2 as well)
        int lt = ltid;
        asm volatile("" : "+v"(lt));        // laundered: otherwise the 20 LDS addresses are hoisted out of the item loop, held for the whole kernel and spilled
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int id = i * LT + lt, row = id >> 2, c = id & 3;
            if (id < Npad * 4) {
                const bool ok = row < N;
                const int off = att_off<HD>(row, c);
                if (keys) *reinterpret_cast<u32x4*>(Ks + off) = ok ? pre[1][i] : zero4;
                else {
                    *reinterpret_cast<u32x4*>(Qs + off) = ok ? pre[0][i] : zero4;
                    *reinterpret_cast<u32x4*>(Vs + off) = ok ? pre[2][i] : zero4;
                    *reinterpret_cast<u32x4*>(Gs + off) = ok ? pre[3][i] : zero4;
                }
            }
        }
    };

    const int rb0 = att_row_base<HD>(lane, 0);
    int tb[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) tb[dt] = att_tr_base<HD>(lane, dt);
    const int sw_base = fr * DS_STR + 4 * g;                                   // write: row = key fr of the tile, columns 4 g .. (+16: second query tile)
    const int sr_base = (4 * g + q4) * DS_STR + 4 * p4;                         // transposed read: rows 4 g + q4 (+16), columns 4 p4 ..

    // LDS-only barrier: the waves wait for their LDS traffic, not for global stores / prefetches in flight (a __syncthreads would also
    // drain vmcnt -- the loaders' prefetch and the dK / dV stores)
#define DS_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if (loader) {
        // the two roles are separate loops with the same three barriers per item: in one loop the loader's 90 registers of prefetched rows
        // would be live across the other waves' arithmetic (the allocator does not know the branch is per wave) and spill
        // item v of the launch is xcd_remap(v): the items an XCD works on at a time are consecutive (heads of one image share 128-B lines)
        int item = blockIdx.x, par = 0;
        if (item < nitems) { issue(xcd_remap(item, nitems), flb, flb + Npad); deposit(false); deposit(true); }
        for (; item < nitems; item += gridDim.x) {
            DS_BAR();                                                          // A: tiles, lse, delta of `item` are in LDS
            const int nxt = item + gridDim.x;
            par ^= 1;
            if (nxt < nitems) issue(xcd_remap(nxt, nitems), flb + 2 * par * Npad, flb + (2 * par + 1) * Npad);
            DS_BAR();                                                          // B: phase 1 is over, Q / V / dO are free
            if (nxt < nitems) deposit(false);
            DS_BAR();                                                          // C: K and dS^T are free
            if (nxt < nitems) deposit(true);
        }
        return;
    }
    int par = 0;
    for (int v = blockIdx.x; v < nitems; v += gridDim.x, par ^= 1) {
        DS_BAR();                                                              // A
        const float* fl = flb + 2 * par * Npad;
        const float* fd = fl + Npad;
        const int item = xcd_remap(v, nitems);
        const int b = item / heads, h = item % heads;
        bf16_t* dbase = dqkv + (int64_t)b * N * ld + h * HD;
        f32x4 dk[2] = {z, z}, dv[2] = {z, z};
        if (wave < ntile) {
            // ---- phase 1: this wave owns key tile `wave` -> dK, dV, and its dS block to LDS
            const int jt = wave;
            const bf16x8 kf = att_row_at<HD>(Ks, rb0, jt * 16), vf = att_row_at<HD>(Vs, rb0, jt * 16);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { dk[dt] = z; dv[dt] = z; }
            bf16_t* Sj = dSt + 16 * jt * DS_STR + sw_base;
#pragma unroll
            for (int qs = 0; qs < NP; ++qs) {
                f32x4 p[2], ds[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int q0 = (2 * qs + hf) * 16;
                    // the accumulator of dP starts at -delta (stored negated): dP - delta costs no VALU.  (A v_mul_f32 in inline asm, to keep
                    // hipcc from pairing the multiplies into v_pk_mul_f32, is not an option: the hazard recognizer does not see it and the
                    // multiply read dP before the MFMA had written it.)
                    const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Qs, rb0, q0), kf, z, 0, 0, 0);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Gs, rb0, q0), vf, *reinterpret_cast<const f32x4*>(fd + q0 + 4 * g), 0, 0, 0);
                    const f32x4 fl4 = *reinterpret_cast<const f32x4*>(fl + q0 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fminf(fmaf(sc[r], c2, -fl4[r]), ATT_PCAP));
                        p[hf][r] = pv;
                        ds[hf][r] = pv * dp[r];                                   // the softmax scale is applied once to dK / dQ
                    }
                }
                const bf16x8 pf = pack_frag(p[0], p[1]);
                const u32x4 dsu = __builtin_bit_cast(u32x4, pack_frag(ds[0], ds[1]));
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, dsu);
                u32x2 w0, w1;
                w0[0] = dsu[0]; w0[1] = dsu[1]; w1[0] = dsu[2]; w1[1] = dsu[3];
                *reinterpret_cast<u32x2*>(Sj + 32 * qs) = w0;             // queries 32 qs + 4 g .. +3 of key fr
                *reinterpret_cast<u32x2*>(Sj + 32 * qs + 16) = w1;        // second query tile of the pair
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HD>(Gs, tb[dt], 32 * qs), dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, att_tr_at<HD>(Qs, tb[dt], 32 * qs), dk[dt], 0, 0, 0);
                }
            }
        }
        DS_BAR();                                                              // B: dS^T is complete
        if (wave < ntile) {
            // ---- phase 2: this wave owns query tile `wave` -> dQ = dS K (padded keys: zero K rows)
            const int qt = wave;
            f32x4 dq[2] = {z, z};
            const bf16_t* a0 = dSt + sr_base + 16 * qt;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x4 z4 = {0, 0, 0, 0};
            for (int kp = 0; 2 * kp < ntile; ++kp) {
                const bf16_t* a1 = a0 + 32 * kp * DS_STR;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
                s16x4 hi = z4;
                if (2 * kp + 1 < ntile) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1 + 16 * DS_STR));
                const bf16x8 dsq = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsq, att_tr_at<HD>(Ks, tb[dt], 32 * kp), dq[dt], 0, 0, 0);
            }
            // results leave after the arithmetic: nothing waits for these stores (the barriers do not count them)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qt * 16 + 4 * g + r;
                if (q < N) {
                    bf16_t* qp = dbase + (int64_t)q * ld + fr;
                    bf16_t* kp = qp + C;                                 // key index == query index: tile `wave` on both sides
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        qp[dt * 16] = f2bf(dq[dt][r] * scale);
                        kp[dt * 16] = f2bf(dk[dt][r] * scale);
                        kp[C + dt * 16] = f2bf(dv[dt][r]);
                    }
                }
            }
        }
        DS_BAR();                                                              // C: tiles and dS^T are free
    }
#undef DS_BAR
}

// ------------------------------------------------------------------------- class attention
// one query (token 0) per image and head: HBM-bound VALU kernel, PARTS lanes x 16 B per key row (PARTS = 4: head_dim 32;
// PARTS = 8: head_dim 48 / 64 -- lanes whose 8 columns lie beyond head_dim contribute zeros).
template <int PARTS>
__global__ void __launch_bounds__(256)
k_class_attn_fwd(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv, bf16_t* __restrict__ out,
                 float* __restrict__ probs, int N, int heads, int hd, float scale, const bf16_t* __restrict__ kv0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // scores[N] | red[KS*HDP]
    constexpr int KS = 256 / PARTS, HDP = PARTS * 8;               // keys per pass, padded head dim
    float* sc = sm;
    float* red = sm + ((N + 63) & ~63);
    __shared__ float wred[8];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C = heads * hd;
    const int part = threadIdx.x % PARTS, kslot = threadIdx.x / PARTS;
    const bool pok = part * 8 < hd;
    float qv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pok) unpack8(ld16(q + (int64_t)b * C + h * hd + part * 8), qv);
    // split layout (kv0 != nullptr): key 0 (the class token) lives in kv0 [B, 2C], keys 1..N-1 are the N-1 token rows of kv -- the
    // caller never concatenates the class token with the tokens (models/volo.py:304-308 does, a 19 MB copy per block at B = 128)
    const int shift = kv0 ? 1 : 0;
    const bf16_t* kb = kv + ((int64_t)b * (N - shift) - shift) * 2 * C + h * hd + part * 8;      // kb + key*2C is row `key` for key >= shift
    const bf16_t* krow0 = kv0 ? kv0 + (int64_t)b * 2 * C + h * hd + part * 8 : kb;
    float mx = -1.0e30f;
    for (int k0 = 0; k0 < N; k0 += KS) {
        const int key = k0 + kslot;
        float d = 0.f;
        if (key < N && pok) {
            float kk[8];
            unpack8(ld16((kv0 && key == 0) ? krow0 : kb + (int64_t)key * 2 * C), kk);
#pragma unroll
            for (int i = 0; i < 8; ++i) d += qv[i] * kk[i];
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if (PARTS == 8) d += __shfl_xor(d, 4, 64);
        d *= scale;
        if (key < N) { if (part == 0) sc[key] = d; mx = fmaxf(mx, d); }
    }
    mx = group_max<64>(mx);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    float sum = 0.f;
    for (int key = threadIdx.x; key < N; key += 256) { const float e = __expf(sc[key] - mx); sc[key] = e; sum += e; }
    sum = group_sum<64>(sum);
    if ((threadIdx.x & 63) == 0) wred[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (wred[4] + wred[5] + wred[6] + wred[7]);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < N; k0 += KS) {
        const int key = k0 + kslot;
        if (key < N) {
            const float p = sc[key] * inv;
            if (part == 0) probs[((int64_t)b * heads + h) * N + key] = p;
            if (pok) {
                float vv[8];
                unpack8(ld16(((kv0 && key == 0) ? krow0 : kb + (int64_t)key * 2 * C) + C), vv);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += p * vv[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[kslot * HDP + part * 8 + i] = acc[i];
    __syncthreads();
    if ((int)threadIdx.x < hd) {
        float s = 0.f;
        for (int k = 0; k < KS; ++k) s += red[k * HDP + threadIdx.x];
        out[(int64_t)b * C + h * hd + threadIdx.x] = f2bf(s);
    }
}

template <int PARTS>
__global__ void __launch_bounds__(256)
k_class_attn_bwd(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv, const float* __restrict__ probs,
                 const bf16_t* __restrict__ dout, bf16_t* __restrict__ dq, bf16_t* __restrict__ dkv,
                 int N, int heads, int hd, float scale, const bf16_t* __restrict__ kv0, bf16_t* __restrict__ dkv0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // dp[N] | red[KS*HDP]
    constexpr int KS = 256 / PARTS, HDP = PARTS * 8;
    float* dps = sm;
    float* red = sm + ((N + 63) & ~63);
    __shared__ float wred[4];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C = heads * hd;
    const int part = threadIdx.x % PARTS, kslot = threadIdx.x / PARTS;
    const bool pok = part * 8 < hd;
    float qv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pok) {
        unpack8(ld16(q + (int64_t)b * C + h * hd + part * 8), qv);
        unpack8(ld16(dout + (int64_t)b * C + h * hd + part * 8), gv);
    }
    const int shift = kv0 ? 1 : 0;                                  // split layout, see k_class_attn_fwd
    const bf16_t* kb = kv + ((int64_t)b * (N - shift) - shift) * 2 * C + h * hd + part * 8;
    bf16_t* db = dkv + ((int64_t)b * (N - shift) - shift) * 2 * C + h * hd + part * 8;
    const bf16_t* krow0 = kv0 ? kv0 + (int64_t)b * 2 * C + h * hd + part * 8 : kb;
    bf16_t* drow0 = kv0 ? dkv0 + (int64_t)b * 2 * C + h * hd + part * 8 : db;
    const float* pr = probs + ((int64_t)b * heads + h) * N;
    // dp_k = <dout, V_k>;  dV_k = p_k * dout;  dot = sum_k p_k dp_k
    float dot = 0.f;
    for (int k0 = 0; k0 < N; k0 += KS) {
        const int key = k0 + kslot;
        float d = 0.f;
        if (key < N && pok) {
            float vv[8], o8[8];
            const bool first = kv0 && key == 0;
            unpack8(ld16((first ? krow0 : kb + (int64_t)key * 2 * C) + C), vv);
            const float p = pr[key];
#pragma unroll
            for (int i = 0; i < 8; ++i) { d += gv[i] * vv[i]; o8[i] = p * gv[i]; }
            st16((first ? drow0 : db + (int64_t)key * 2 * C) + C, pack8(o8));
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if (PARTS == 8) d += __shfl_xor(d, 4, 64);
        if (key < N && part == 0) { dps[key] = d; dot += pr[key] * d; }
    }
    dot = group_sum<64>(dot);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = dot;
    __syncthreads();
    dot = wred[0] + wred[1] + wred[2] + wred[3];
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < N; k0 += KS) {
        const int key = k0 + kslot;
        if (key < N && pok) {
            const float ds = pr[key] * (dps[key] - dot) * scale;       // d(score)/d(q.k)
            float kk[8], o8[8];
            const bool first = kv0 && key == 0;
            unpack8(ld16(first ? krow0 : kb + (int64_t)key * 2 * C), kk);
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc[i] += ds * kk[i]; o8[i] = ds * qv[i]; }
            st16(first ? drow0 : db + (int64_t)key * 2 * C, pack8(o8));
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[kslot * HDP + part * 8 + i] = acc[i];
    __syncthreads();
    if ((int)threadIdx.x < hd) {
        float s = 0.f;
        for (int k = 0; k < KS; ++k) s += red[k * HDP + threadIdx.x];
        dq[(int64_t)b * C + h * hd + threadIdx.x] = f2bf(s);
    }
}

#define MHSA_FWD_CASE(NTV, HDV) case NTV: hipLaunchKernelGGL((k_mhsa_fwd<NTV, HDV>), grid, dim3(256), lds, s, qkv, out, lse, N, heads, scale, out_row_scale); break;
#define MHSA_FWD_SWITCH(HDV)                                                                     \
    switch (nt) { MHSA_FWD_CASE(2, HDV) MHSA_FWD_CASE(4, HDV) MHSA_FWD_CASE(6, HDV) MHSA_FWD_CASE(8, HDV)  \
                  MHSA_FWD_CASE(10, HDV) MHSA_FWD_CASE(12, HDV) MHSA_FWD_CASE(14, HDV)             \
                  default: hipLaunchKernelGGL((k_mhsa_fwd<16, HDV>), grid, dim3(256), lds, s, qkv, out, lse, N, heads, scale, out_row_scale); break; }

// N <= 256 with head_dim 32 / 64: one workgroup holds the whole head in LDS (kernels above); anything else (448-px inputs,
// head_dim 48 of VOLO-D4/D5) goes to the key/query-blocked kernels of mhsa_flash.hip.  AP_MHSA_FLASH=1 forces the blocked path
// (parity tests run both on the same inputs).
static bool mhsa_use_flash(int N, int hd) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("AP_MHSA_FLASH"); force = (e && e[0] == '1') ? 1 : 0; }
    return force || N > 256 || hd == 48;
}

extern "C" {

int ap_mhsa_fwd_fp8(const ap_bf16* qkv, ap_bf16* out, unsigned char* out8, const float* q_scale, float* q_amax, float* lse, int B, int N,
                    int heads, int hd, float scale, const float* out_row_scale, ap_stream_t stream) {
    if (!qkv || !out || !lse || !out8 || !q_scale) return AP_ERR_NULL;
    if (B <= 0 || N <= 0 || heads <= 0) return AP_ERR_SHAPE;
    if (hd != 32 && hd != 48 && hd != 64) return AP_ERR_UNSUPPORTED;
    if (!mhsa_use_flash(N, hd)) return AP_ERR_UNSUPPORTED;          // the e4m3 side output exists in the blocked kernel only
    return ap_mhsa_flash_fwd(qkv, out, lse, B, N, heads, hd, scale, out_row_scale, (hipStream_t)stream, out8, q_scale, q_amax);
}

int ap_mhsa_fwd(const ap_bf16* qkv, ap_bf16* out, float* lse, int B, int N, int heads, int hd, float scale, const float* out_row_scale,
                ap_stream_t stream) {
    if (!qkv || !out || !lse) return AP_ERR_NULL;
    if (B <= 0 || N <= 0 || heads <= 0) return AP_ERR_SHAPE;
    if (hd != 32 && hd != 48 && hd != 64) return AP_ERR_UNSUPPORTED;
    if (mhsa_use_flash(N, hd)) return ap_mhsa_flash_fwd(qkv, out, lse, B, N, heads, hd, scale, out_row_scale, (hipStream_t)stream);
    const int nt = 2 * ((N + 31) / 32);
    const dim3 grid(B * heads);
    const size_t lds = (size_t)2 * nt * 16 * hd * sizeof(bf16_t);
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_mhsa_fwd<14, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void*)k_mhsa_fwd<16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_done = true;
    }
    (void)hipGetLastError();
    // head_dim 32 and 7 .. 13 query tiles: the persistent kernel with loader waves (AP_MHSA_FWD_P=0: one workgroup per (image, head))
    static int use_p = -1, p_cu = 0;
    if (use_p < 0) {
        const char* e = getenv("AP_MHSA_FWD_P"); use_p = e ? atoi(e) : 1;
        int dev = 0; hipDeviceProp_t pr;
        p_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
        p_cu &= ~7;                                                            // whole XCD groups (xcd_remap)
        (void)hipGetLastError();
    }
    const int nq = (N + 15) / 16;
    if (hd == 32 && use_p && nq >= 7 && nq <= 16 - FP_LOADERS && p_cu >= 8) {
        const int items = B * heads;
        const dim3 gp(items < p_cu ? items : p_cu);
        const size_t lds_p = 2 * lds;
#define FP_LAUNCH(NTV) hipLaunchKernelGGL((k_mhsa_fwd_p<NTV>), gp, dim3(1024), lds_p, s, qkv, out, lse, N, heads, scale, out_row_scale, items)
        switch (nt) { case 8: FP_LAUNCH(8); break; case 10: FP_LAUNCH(10); break; case 12: FP_LAUNCH(12); break; default: FP_LAUNCH(14); break; }
#undef FP_LAUNCH
        return ap_check_launch();
    }
    if (hd == 32) { MHSA_FWD_SWITCH(32) } else { MHSA_FWD_SWITCH(64) }
    return ap_check_launch();
}

size_t ap_mhsa_bwd_workspace(int B, int N, int heads, int hd) {
    if (B <= 0 || N <= 0 || heads <= 0) return 0;
    return mhsa_use_flash(N, hd) ? ap_mhsa_flash_bwd_ws(B, N, heads) : 0;
}

int ap_mhsa_bwd(const ap_bf16* qkv, const ap_bf16* out, const ap_bf16* dout, const float* lse, ap_bf16* dqkv,
                int B, int N, int heads, int hd, float scale, void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !dqkv) return AP_ERR_NULL;
    if (B <= 0 || N <= 0 || heads <= 0) return AP_ERR_SHAPE;
    if (hd != 32 && hd != 48 && hd != 64) return AP_ERR_UNSUPPORTED;
    if (mhsa_use_flash(N, hd)) {
        if (!workspace) return AP_ERR_NULL;
        if (ws_bytes < ap_mhsa_flash_bwd_ws(B, N, heads)) return AP_ERR_SHAPE;
        return ap_mhsa_flash_bwd(qkv, out, dout, lse, dqkv, B, N, heads, hd, scale, (float*)workspace, (hipStream_t)stream);
    }
    const int nt = 2 * ((N + 31) / 32);
    const dim3 grid(B * heads);
    const size_t lds = (size_t)4 * nt * 16 * hd * sizeof(bf16_t) + (size_t)2 * nt * 16 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_mhsa_bwd<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_mhsa_bwd<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    (void)hipGetLastError();
    // default for head_dim 32 and 8 .. 13 token tiles (N = 113 .. 208): dS through LDS, loader waves (AP_MHSA_BWD_DS=0: the two-pass kernel)
    static int use_ds = -1, ds_cu = 0;
    if (use_ds < 0) {
        const char* e = getenv("AP_MHSA_BWD_DS"); use_ds = e ? atoi(e) : 1;
        int dev = 0; hipDeviceProp_t pr;
        ds_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
        if (ds_cu >= 8) ds_cu &= ~7;                                           // whole XCD groups (xcd_remap)
        (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_ds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_ds<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_ds<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_ds<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    {
        const int ntile = (N + 15) / 16;
        const size_t lds_ds = lds + (size_t)2 * nt * 16 * sizeof(float) + (size_t)ntile * 16 * DS_STR * sizeof(bf16_t);
        if (hd == 32 && use_ds && ntile <= 16 - DS_LOADERS && ntile >= 7 && nt * 16 * 4 <= 5 * 64 * DS_LOADERS && nt * 16 <= DS_STR - 8 &&
            lds_ds <= (size_t)160 * 1024) {
            const int items = B * heads;
            const dim3 gds(items < ds_cu ? items : ds_cu);
#define DS_LAUNCH(NP) hipLaunchKernelGGL(k_mhsa_bwd_ds<NP>, gds, dim3(1024), lds_ds, s, qkv, out, dout, lse, dqkv, N, heads, scale, nt, items)
            switch (nt / 2) { case 4: DS_LAUNCH(4); break; case 5: DS_LAUNCH(5); break; case 6: DS_LAUNCH(6); break; default: DS_LAUNCH(7); break; }
#undef DS_LAUNCH
            return ap_check_launch();
        }
    }
    if (hd == 32) hipLaunchKernelGGL(k_mhsa_bwd<32>, grid, dim3(512), lds, s, qkv, out, dout, lse, dqkv, N, heads, scale, nt);
    else hipLaunchKernelGGL(k_mhsa_bwd<64>, grid, dim3(512), lds, s, qkv, out, dout, lse, dqkv, N, heads, scale, nt);
    return ap_check_launch();
}

int ap_class_attn_fwd(const ap_bf16* q, const ap_bf16* kv, const ap_bf16* kv_cls, ap_bf16* out, float* probs, int B, int N, int heads, int hd,
                      float scale, ap_stream_t stream) {
    if (!q || !kv || !out || !probs) return AP_ERR_NULL;
    if (B <= 0 || N <= 0 || heads <= 0) return AP_ERR_SHAPE;
    if (hd != 32 && hd != 48 && hd != 64) return AP_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)((N + 63) & ~63) + 64 * 32) * sizeof(float);          // KS * HDP = 2048 floats for both PARTS
    if (lds > 64 * 1024) return AP_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    if (hd == 32) hipLaunchKernelGGL(k_class_attn_fwd<4>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, q, kv, out, probs, N, heads, hd, scale, kv_cls);
    else hipLaunchKernelGGL(k_class_attn_fwd<8>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, q, kv, out, probs, N, heads, hd, scale, kv_cls);
    return ap_check_launch();
}

int ap_class_attn_bwd(const ap_bf16* q, const ap_bf16* kv, const ap_bf16* kv_cls, const float* probs, const ap_bf16* dout, ap_bf16* dq, ap_bf16* dkv,
                      ap_bf16* dkv_cls, int B, int N, int heads, int hd, float scale, ap_stream_t stream) {
    if (!q || !kv || !probs || !dout || !dq || !dkv || ((kv_cls == nullptr) != (dkv_cls == nullptr))) return AP_ERR_NULL;
    if (B <= 0 || N <= 0 || heads <= 0) return AP_ERR_SHAPE;
    if (hd != 32 && hd != 48 && hd != 64) return AP_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)((N + 63) & ~63) + 64 * 32) * sizeof(float);
    if (lds > 64 * 1024) return AP_ERR_UNSUPPORTED;
    (void)hipGetLastError();
    if (hd == 32) hipLaunchKernelGGL(k_class_attn_bwd<4>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, q, kv, probs, dout, dq, dkv, N, heads, hd, scale, kv_cls, dkv_cls);
    else hipLaunchKernelGGL(k_class_attn_bwd<8>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, q, kv, probs, dout, dq, dkv, N, heads, hd, scale, kv_cls, dkv_cls);
    return ap_check_launch();
}

}  // extern "C"
