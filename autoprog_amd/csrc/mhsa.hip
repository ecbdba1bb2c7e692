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
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -fl4[r]));
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
                    // padded keys need no mask here: their K rows are zero, so they add nothing to dQ
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -flq));
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

// ---- backward in ONE pass (head_dim 32, N <= 256) ---------------------------------------------------------------------------------
// k_mhsa_bwd recomputes S and dP (and the exponentials) twice: once key-owned for dK/dV, once query-owned for dQ.  Here a wave owns a
// PAIR of key tiles (32 keys) and walks the query-tile pairs once: S, dP, P, dS for the 32 x 32 block, dV += P^T dO and dK += dS^T Q as
// before -- and dQ, which needs dS with the QUERY as the fragment row: the wave writes its dS block to a private LDS scratch as
// [key][q] (8-byte stores) and reads it back transposed (ds_read_b64_tr_b16) in the key order of the K^T fragment.  Every wave keeps
// dQ partial sums of ALL query tiles in registers (16 x 2 accumulator tiles); after the loop the waves' partials are added through LDS
// in a fixed order (bit-reproducible) and stored.  Half the exponentials / VALU work and 30 % fewer MFMAs than the two-pass kernel; one
// workgroup per CU (registers), persistent over the (image, head) items.
// STATUS (round 2): parity-green, selected with AP_MHSA_ONEPASS=1, NOT the default: 117 us against 64 us at B = 128, N = 196.  By
// phase (AP_MHSA_DBG bits): operand staging alone 33 us, + main loop 49, + dQ reduction 35 -- with one workgroup per CU the three run
// one after the other, and the main loop is a dependent MFMA -> exp -> pack -> MFMA -> LDS -> MFMA chain per wave at two waves per SIMD.
// What it needs to win: the next item's tiles staged during the main loop, a tree reduction through 16-byte LDS accesses, and two
// query-pair chains interleaved by hand.
#define OP_SSTR 40               // scratch row stride (bf16): 32 q + padding
template <int NP, bool EXACT>    // key / query tile PAIRS the registers are sized for (7: N <= 224, i.e. the 196 tokens of a 224 px image);
                                 // EXACT: NT == 2 NP, so the unrolled loops carry no guards and stay one basic block the scheduler can interleave
__global__ void __launch_bounds__(512)
k_mhsa_bwd_onepass(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                   const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int N, int heads, float scale, int NT, int nitems, int dbg = 0) {
    constexpr int HD = 32, CH = 4;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int Npad = NT * 16;
    bf16_t* Qs = smem;
    bf16_t* Ks = Qs + Npad * HD;
    bf16_t* Vs = Ks + Npad * HD;
    bf16_t* Gs = Vs + Npad * HD;
    float* fl = reinterpret_cast<float*>(Gs + Npad * HD);
    float* fd = fl + Npad;
    bf16_t* scratch = reinterpret_cast<bf16_t*>(fd + Npad);                   // [8 waves][4 slots][32 keys][OP_SSTR]
    float* red = reinterpret_cast<float*>(smem);                              // after the main loop: [8 waves][32 q][32 d] fp32 over Q/K/V
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4, q4 = fr >> 2, p4 = fr & 3;
    const int C = heads * HD;
    const int64_t ld = 3 * C;
    const float c2 = scale * 1.4426950408889634f;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int rb0 = att_row_base<HD>(lane, 0);
    int tb[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) tb[dt] = att_tr_base<HD>(lane, dt);
    bf16_t* Sw = scratch + wave * 4 * 32 * OP_SSTR;
    const int sw_base = fr * OP_SSTR + 4 * g;                                  // write: row = key (fr) (+16 jt), columns 4 g .. (+16 hf)
    const int sr_base = (4 * g + q4) * OP_SSTR + 4 * p4;                        // transposed read: rows 4 g + q4 (+16), columns 4 p4 .. (+16 hf)
    const int npair = NT >> 1;
    typedef __attribute__((ext_vector_type(8))) short s16x8;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int wg = item;
        const int b = wg / heads, h = wg % heads;
        const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
        const bf16_t* obase = out + (int64_t)b * N * C + h * HD;
        const bf16_t* gbase = dout + (int64_t)b * N * C + h * HD;
        bf16_t* dbase = dqkv + (int64_t)b * N * ld + h * HD;
        __syncthreads();                                     // the previous item's reduction reads are done
        {   // delta loads, then the operand tiles: one exposed latency
            u32x4 va0[2], vd0[2];
            float ls0[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = threadIdx.x + it * 512;
                const int row = idx / CH, c = idx % CH;
                const bool ok = idx < Npad * CH && row < N;
                va0[it] = ok ? ld16(obase + (int64_t)row * C + c * 8) : zero4;
                vd0[it] = ok ? ld16(gbase + (int64_t)row * C + c * 8) : zero4;
                ls0[it] = (ok && c == 0) ? lse[((int64_t)b * heads + h) * N + row] : 0.f;
            }
            bf16_t* tiles[4] = {Qs, Ks, Vs, Gs};
            const bf16_t* srcs[4] = {base, base + C, base + 2 * C, gbase};
            const int64_t strides[4] = {ld, ld, ld, (int64_t)C};
            att_stage_n<HD, 4, 2>(tiles, srcs, strides, N, Npad);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = threadIdx.x + it * 512;
                if (idx >= Npad * CH) continue;
                const int row = idx / CH, c = idx % CH;
                float a[8], d[8];
                unpack8(va0[it], a);
                unpack8(vd0[it], d);
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) part += a[k] * d[k];
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                if (c == 0) { fd[row] = part; fl[row] = (row < N) ? ls0[it] * 1.4426950408889634f : 1.0e30f; }
            }
        }
        __syncthreads();
        f32x4 dq[2 * NP][2];
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t) { dq[t][0] = z; dq[t][1] = z; }
        if (wave < npair && !(dbg & 1)) {
            const int k0 = 32 * wave;
            bf16x8 kf[2], vf[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) { kf[jt] = att_row_at<HD>(Ks, rb0, k0 + 16 * jt); vf[jt] = att_row_at<HD>(Vs, rb0, k0 + 16 * jt); }
            f32x4 dk[2][2], dv[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { dk[a][0] = z; dk[a][1] = z; dv[a][0] = z; dv[a][1] = z; }
#pragma unroll
            for (int qs = 0; qs < NP; ++qs) {
                if (EXACT || qs < npair) {
                    bf16_t* S = Sw + (qs & 3) * 32 * OP_SSTR;          // four scratch slots per wave: consecutive query pairs do not wait for each other
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) {         // one key tile at a time: P / dS of 16 keys x 32 queries live
                        f32x4 p[2], ds[2];                   // [hf]
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const int q0 = (2 * qs + hf) * 16;
                            const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Qs, rb0, q0), kf[jt], z, 0, 0, 0);
                            const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_at<HD>(Gs, rb0, q0), vf[jt], z, 0, 0, 0);
                            const f32x4 fl4 = *reinterpret_cast<const f32x4*>(fl + q0 + 4 * g);
                            const f32x4 fd4 = *reinterpret_cast<const f32x4*>(fd + q0 + 4 * g);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float pv = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -fl4[r]));
                                p[hf][r] = pv;
                                ds[hf][r] = pv * (dp[r] - fd4[r]);
                            }
                        }
                        const bf16x8 pf = pack_frag(p[0], p[1]);
                        const u32x4 dsu = __builtin_bit_cast(u32x4, pack_frag(ds[0], ds[1]));
                        const bf16x8 dsf = __builtin_bit_cast(bf16x8, dsu);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            dv[jt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, att_tr_at<HD>(Gs, tb[dt], 32 * qs), dv[jt][dt], 0, 0, 0);
                            dk[jt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, att_tr_at<HD>(Qs, tb[dt], 32 * qs), dk[jt][dt], 0, 0, 0);
                        }
                        // dS block -> scratch [key][q]: this lane's 4 queries (4 g ..) of key fr, both query tiles
                        u32x2 w0, w1;
                        w0[0] = dsu[0]; w0[1] = dsu[1]; w1[0] = dsu[2]; w1[1] = dsu[3];
                        *reinterpret_cast<u32x2*>(S + (16 * jt) * OP_SSTR + sw_base) = w0;            // hf = 0
                        *reinterpret_cast<u32x2*>(S + (16 * jt) * OP_SSTR + sw_base + 16) = w1;       // hf = 1
                    }
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const bf16_t* a1 = S + sr_base + 16 * hf;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1 + 16 * OP_SSTR));
                        const bf16x8 dsq = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            dq[2 * qs + hf][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsq, att_tr_at<HD>(Ks, tb[dt], k0), dq[2 * qs + hf][dt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + 16 * jt + 4 * g + r;
                    if (key < N) {
                        bf16_t* kp = dbase + (int64_t)key * ld + C + fr;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) { kp[dt * 16] = f2bf(dk[jt][dt][r] * scale); kp[C + dt * 16] = f2bf(dv[jt][dt][r]); }
                    }
                }
        }
        __syncthreads();                                     // every wave is done with the operand tiles: their LDS becomes the reduction buffer
        // dQ = sum over the key-pair waves, two query tiles per round, fixed order
#pragma unroll
        for (int rr = 0; rr < NP; ++rr) {
            if ((EXACT || rr < npair) && !(dbg & 2)) {
                if (wave < npair) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                red[(wave * 32 + 16 * hf + 4 * g + r) * 32 + 16 * dt + fr] = dq[2 * rr + hf][dt][r];
                }
                __syncthreads();
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int e = threadIdx.x + 512 * e2, ql = e >> 5, d = e & 31;
                    const int q = 32 * rr + ql;
                    float acc = 0.f;
                    for (int w = 0; w < npair; ++w) acc += red[(w * 32 + ql) * 32 + d];
                    if (q < N) dbase[(int64_t)q * ld + d] = f2bf(acc * scale);
                }
                __syncthreads();
            }
        }
    }
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
    static int onepass = -1;
    if (onepass < 0) { const char* e = getenv("AP_MHSA_ONEPASS"); onepass = e ? atoi(e) : 0; }      // default off: 117 us against 64 (phases below)
    if (hd == 32 && onepass && nt <= 14) {            // up to 7 tile pairs (N <= 224): beyond that the dQ accumulators spill
        const size_t lds1 = lds + (size_t)8 * 4 * 32 * OP_SSTR * sizeof(bf16_t);
        static bool a1 = false;
        if (!a1) {
            (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_onepass<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_onepass<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_mhsa_bwd_onepass<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            a1 = true; (void)hipGetLastError();
        }
        static int n_cu = 0;
        if (n_cu == 0) { int dev = 0; hipGetDevice(&dev); hipDeviceProp_t pr; n_cu = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
        const int items = B * heads;
        int g1 = onepass > 1 ? onepass : n_cu;               // AP_MHSA_ONEPASS=<n> > 1: that many persistent workgroups
        if (g1 > items) g1 = items;
        if ((size_t)8 * 32 * 32 * sizeof(float) <= (size_t)3 * nt * 16 * hd * sizeof(bf16_t)) {      // the reduction buffer fits the Q/K/V tiles
            const int np = nt / 2;
            if (np <= 4) hipLaunchKernelGGL((k_mhsa_bwd_onepass<4, false>), dim3(g1), dim3(512), lds1, s, qkv, out, dout, lse, dqkv, N, heads, scale, nt, items);
            else if (np == 7) { static int dbg = -1; if (dbg < 0) { const char* e = getenv("AP_MHSA_DBG"); dbg = e ? atoi(e) : 0; }
                hipLaunchKernelGGL((k_mhsa_bwd_onepass<7, true>), dim3(g1), dim3(512), lds1, s, qkv, out, dout, lse, dqkv, N, heads, scale, nt, items, dbg); }
            else hipLaunchKernelGGL((k_mhsa_bwd_onepass<7, false>), dim3(g1), dim3(512), lds1, s, qkv, out, dout, lse, dqkv, N, heads, scale, nt, items);
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
