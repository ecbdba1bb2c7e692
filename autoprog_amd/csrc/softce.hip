// Dense soft-target cross entropy, forward + gradient in one pass over (logits, target):
//   row_loss = lse*sum_c t - sum_c t*x ;  dx = gscale*(softmax(x)*sum_c t - t)
// (loss/cross_entropy.py:35-36; token-label targets are CLASS-major [B,C,2+N] while logits are
// token-major [B*N,C], loss/cross_entropy.py:147-148 -- the transpose is done through LDS).
// HBM-bound: logits 2 B + target 4 B + dlogits 2 B per (row, class).
//
// Block = one (batch, tile of TN=16 tokens).  Phase A: the target tile is read coalesced along
// the token axis (16 consecutive fp32 = 64 B per class) and stored token-major in LDS with an
// odd row stride.  Phase B: each wave owns 4 rows; lanes stride over classes (coalesced logits),
// keep the row in registers, reduce max / sum-exp / sum t / sum t*x with wavefront shuffles.
#include "common.h"

#define CE_TN 16
#define CE_MAXV 8          // classes per lane pairs: supports C <= 64*2*CE_MAXV = 1024

__global__ void __launch_bounds__(256)
k_soft_ce(const bf16_t* __restrict__ logits, int ldx, const float* __restrict__ target, int64_t t_sb, int64_t t_sc,
          int64_t t_sn, int rows_per_batch, float* __restrict__ row_loss, bf16_t* __restrict__ dlogits,
          float gscale, int64_t M, int C, int tiles_per_batch, float mix_lam, int mix_batches, const float* __restrict__ lam_dev = nullptr) {
    if (lam_dev) mix_lam = lam_dev[0];                   // the step's lam from device memory (graph replay; lam = 1: lam t + 0 t' = t exactly)
    extern __shared__ __attribute__((aligned(16))) float tt[];      // [CE_TN][Cp] Cp odd
    const int Cp = C | 1;
    const int64_t b = blockIdx.x / tiles_per_batch;
    const int n0 = (blockIdx.x % tiles_per_batch) * CE_TN;
    const int ntok = min(CE_TN, rows_per_batch - n0);
    // the logits of this wave's (up to) four rows are requested FIRST: they do not depend on the target tile, so their latency hides
    // behind phase A (clamped, unconditional 4-byte loads; columns beyond C are masked where they are used)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned lraw[CE_TN / 4][CE_MAXV];
#pragma unroll
    for (int r = 0; r < CE_TN / 4; ++r) {
        const int64_t rowc = min(b * rows_per_batch + n0 + min(wave + 4 * r, ntok - 1), M - 1);
        const bf16_t* xr = logits + rowc * ldx;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) lraw[r][i] = *reinterpret_cast<const unsigned*>(xr + min(2 * (lane + 64 * i), ldx - 2));
    }
    // ---- phase A: target tile -> LDS (token-major)
    {
        const int tn = threadIdx.x & (CE_TN - 1), cl = threadIdx.x / CE_TN;     // 16 classes per pass
        const int tnc = min(tn, ntok - 1);                                       // lanes beyond the tile's tokens read a valid token (never stored)
        const float* tb = target + b * t_sb + (int64_t)(n0 + tnc) * t_sn;
        // mix-token: the image-level label of sample b is lam * t[b] + (1 - lam) * t[B-1-b] (loss/cross_entropy.py:151-152)
        const float* tb2 = mix_batches > 0 ? target + (int64_t)(mix_batches - 1 - b) * t_sb + (int64_t)(n0 + tnc) * t_sn : tb;
        const float lam2 = mix_batches > 0 ? 1.0f - mix_lam : 0.f, lam1 = mix_batches > 0 ? mix_lam : 1.0f;
        // CE_INFLIGHT independent loads in flight per thread: the tile is 63 four-byte loads per thread, and every batch exposes one memory
        // latency (one load per iteration: 63 latencies; 8 per batch: 8; 32 per batch: 2)
        constexpr int CSTEP = 256 / CE_TN;
        constexpr int CE_INFLIGHT = 32;
        const bool mixed = mix_batches > 0;
        for (int c0 = cl; c0 < C; c0 += CE_INFLIGHT * CSTEP) {
            float v[CE_INFLIGHT], v2[CE_INFLIGHT];
#pragma unroll
            for (int u = 0; u < CE_INFLIGHT; ++u) {
                const int c = min(c0 + u * CSTEP, C - 1);                       // clamped: unconditional loads (no exec-masked blocks)
                v[u] = tb[(int64_t)c * t_sc];
                v2[u] = mixed ? tb2[(int64_t)c * t_sc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < CE_INFLIGHT; ++u) {
                const int c = c0 + u * CSTEP;
                if (tn < ntok && c < C) tt[tn * Cp + c] = lam1 * v[u] + lam2 * v2[u];
            }
        }
    }
    __syncthreads();
    // ---- phase B
#pragma unroll
    for (int r = 0; r < CE_TN / 4; ++r) {
        const int tn = wave + 4 * r;
        const int64_t row = b * rows_per_batch + n0 + tn;
        if (tn >= ntok || row >= M) break;
        const float* tr = tt + tn * Cp;
        float xv[CE_MAXV][2];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) {
            const int c = 2 * (lane + 64 * i);
            const unsigned u = lraw[r][i];
            xv[i][0] = c < C ? bf_lo(u) : -3.0e38f;
            xv[i][1] = c + 1 < C ? bf_hi(u) : -3.0e38f;
            mx = fmaxf(mx, fmaxf(xv[i][0], xv[i][1]));
        }
        mx = group_max<64>(mx);
        float se = 0.f, st = 0.f, stx = 0.f;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) {
            const int c = 2 * (lane + 64 * i);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (c + k < C) {
                    const float t = tr[c + k];
                    se += __expf(xv[i][k] - mx);
                    st += t;
                    stx += t * xv[i][k];
                }
            }
        }
        se = group_sum<64>(se); st = group_sum<64>(st); stx = group_sum<64>(stx);
        const float lse = mx + __logf(se);
        if (lane == 0) row_loss[row] = lse * st - stx;
        bf16_t* dr = dlogits + row * ldx;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) {
            const int c = 2 * (lane + 64 * i);
            if (c < ldx) {      // ldx is even (multiple of 8): pairs never straddle the row end
                float d0 = 0.f, d1 = 0.f;
                if (c < C) d0 = gscale * (__expf(xv[i][0] - lse) * st - tr[c]);
                if (c + 1 < C) d1 = gscale * (__expf(xv[i][1] - lse) * st - tr[c + 1]);
                *reinterpret_cast<unsigned*>(dr + c) = pack_bf2(d0, d1);
            }
        }
    }
}

// The same loss on the token-label target in its SOURCE form.  The reference builds the dense class-major [B,C,2+N] tensor from
// top-K (class, score) label maps plus label smoothing on the GPU every step (main_prog.py:994-1004, tlt create_token_label_target)
// and the dense kernel above then reads 4 B per (row, class) of it -- 101 MB at B = 128.  Here a row's target is
//     t[c] = (1 - s) * sum_k [idx_k == c] * val_k + s / C
// formed in registers from its K pairs: logits in, dlogits out, nothing else.  One wave per row, a lane owns class pairs
// 2 (lane + 64 i).  Pairs of row r = (b, n), b = r / rows_per_batch, sit at pairs + b * p_sb + n * p_sn (K entries each).
// mix_batches = B > 0 (the mix-token class target, loss/cross_entropy.py:150-152: lam * t[b] + (1 - lam) * t[B-1-b]): lanes K .. 2K-1
// hold the pairs of the same slot of image B-1-b weighted 1 - lam, the row's own are weighted lam -- 2K <= CE_MAXK pairs, no
// concatenated / flipped / scaled copies of the label maps on the host side.
#define CE_MAXK 16
#define CE_SR 4            // rows per wave, all of their loads issued before the first is reduced
__global__ void __launch_bounds__(256)
k_soft_ce_sparse(const bf16_t* __restrict__ logits, int ldx, const int* __restrict__ idx, const float* __restrict__ val, int K,
                 int64_t p_sb, int64_t p_sn, int rows_per_batch, float smoothing, float* __restrict__ row_loss,
                 bf16_t* __restrict__ dlogits, float gscale, int64_t M, int C, float mix_lam, int mix_batches, const float* __restrict__ lam_dev = nullptr) {
    if (lam_dev) mix_lam = lam_dev[0];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * CE_SR;
    if (row0 >= M) return;
    const int KK = mix_batches > 0 ? 2 * K : K;              // pairs per row
    unsigned lraw[CE_SR][CE_MAXV];
    int my_i[CE_SR];
    float my_v[CE_SR];
#pragma unroll
    for (int r = 0; r < CE_SR; ++r) {
        const int64_t row = min(row0 + r, M - 1);
        const bf16_t* xr = logits + row * ldx;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) lraw[r][i] = *reinterpret_cast<const unsigned*>(xr + min(2 * (lane + 64 * i), ldx - 2));
        const int64_t b = row / rows_per_batch, n = row - b * rows_per_batch;
        const bool other = lane >= K;                                // (mix) the partner image's pairs
        const int64_t po = ((mix_batches > 0 && other) ? (int64_t)(mix_batches - 1) - b : b) * p_sb + n * p_sn + (other ? lane - K : lane);
        const float wgt = mix_batches > 0 ? (other ? 1.0f - mix_lam : mix_lam) : 1.0f;
        my_i[r] = lane < KK ? idx[po] : -1;                          // lane k holds pair k of the row
        my_v[r] = lane < KK ? val[po] * wgt * (1.0f - smoothing) : 0.f;
    }
    const float base = smoothing / (float)C;
#pragma unroll
    for (int r = 0; r < CE_SR; ++r) {
        const int64_t row = row0 + r;
        if (row >= M) break;
        float xv[CE_MAXV][2], tv[CE_MAXV][2];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) {
            const int c = 2 * (lane + 64 * i);
            xv[i][0] = c < C ? bf_lo(lraw[r][i]) : -3.0e38f;
            xv[i][1] = c + 1 < C ? bf_hi(lraw[r][i]) : -3.0e38f;
            tv[i][0] = c < C ? base : 0.f;
            tv[i][1] = c + 1 < C ? base : 0.f;
            mx = fmaxf(mx, fmaxf(xv[i][0], xv[i][1]));
        }
        // the K pairs are wave-uniform once read with v_readlane: class ci = 2 * (owner + 64 * slot) + (ci & 1) lives in ONE lane
        for (int k = 0; k < KK; ++k) {
            const int ci = __builtin_amdgcn_readlane(my_i[r], k);
            const float cv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_v[r]), k));
            if (ci < 0 || ci >= C) continue;
            const int owner = (ci >> 1) & 63, slot = ci >> 7, odd = ci & 1;
            const float add = lane == owner ? cv : 0.f;
#pragma unroll
            for (int i = 0; i < CE_MAXV; ++i) { tv[i][0] += (i == slot && !odd) ? add : 0.f; tv[i][1] += (i == slot && odd) ? add : 0.f; }
        }
        mx = group_max<64>(mx);
        float se = 0.f, st = 0.f, stx = 0.f;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (2 * (lane + 64 * i) + k < C) {
                    se += __expf(xv[i][k] - mx);
                    st += tv[i][k];
                    stx += tv[i][k] * xv[i][k];
                }
            }
        se = group_sum<64>(se); st = group_sum<64>(st); stx = group_sum<64>(stx);
        const float lse = mx + __logf(se);
        if (lane == 0) row_loss[row] = lse * st - stx;
        bf16_t* dr = dlogits + row * ldx;
#pragma unroll
        for (int i = 0; i < CE_MAXV; ++i) {
            const int c = 2 * (lane + 64 * i);
            if (c < ldx) {
                float d0 = 0.f, d1 = 0.f;
                if (c < C) d0 = gscale * (__expf(xv[i][0] - lse) * st - tv[i][0]);
                if (c + 1 < C) d1 = gscale * (__expf(xv[i][1] - lse) * st - tv[i][1]);
                *reinterpret_cast<unsigned*>(dr + c) = pack_bf2(d0, d1);
            }
        }
    }
}

// loss = wa * sum(a[0:na]) + wb * sum(b[0:nb]) in one workgroup (the two CE terms of the token-label loss, loss/cross_entropy.py:154-156)
__global__ void __launch_bounds__(1024)
k_loss_combine(const float* __restrict__ a, int64_t na, float wa, const float* __restrict__ b, int64_t nb, float wb, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < na; i += 1024) s += wa * a[i];
    for (int64_t i = threadIdx.x; i < nb; i += 1024) s += wb * b[i];
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = t;
    }
}

extern "C" int ap_loss_combine(const float* a, int64_t na, float wa, const float* b, int64_t nb, float wb, float* out, ap_stream_t stream) {
    if (!a || !out || (nb > 0 && !b)) return AP_ERR_NULL;
    if (na < 0 || nb < 0) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_loss_combine, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, na, wa, b, nb, wb, out);
    return ap_check_launch();
}

extern "C" int ap_soft_ce_fwd_bwd(const ap_bf16* logits, int ldx, const float* target, int64_t t_sb, int64_t t_sc,
                                  int64_t t_sn, int rows_per_batch, float* row_loss, ap_bf16* dlogits,
                                  float grad_scale, int64_t M, int C, float mix_lam, int mix_batches, ap_stream_t stream) {
    return ap_soft_ce_fwd_bwd_dev(logits, ldx, target, t_sb, t_sc, t_sn, rows_per_batch, row_loss, dlogits, grad_scale, M, C, mix_lam, mix_batches, nullptr, stream);
}

extern "C" int ap_soft_ce_fwd_bwd_dev(const ap_bf16* logits, int ldx, const float* target, int64_t t_sb, int64_t t_sc,
                                      int64_t t_sn, int rows_per_batch, float* row_loss, ap_bf16* dlogits,
                                      float grad_scale, int64_t M, int C, float mix_lam, int mix_batches, const float* mix_lam_dev, ap_stream_t stream) {
    if (!logits || !target || !row_loss || !dlogits) return AP_ERR_NULL;
    if (mix_batches != 0 && (mix_batches < 0 || (int64_t)mix_batches * rows_per_batch != M)) return AP_ERR_SHAPE;
    if (C <= 0 || ldx < C || (ldx & 7) || rows_per_batch <= 0 || M % rows_per_batch) return AP_ERR_SHAPE;
    if (ldx > 64 * 2 * CE_MAXV) return AP_ERR_UNSUPPORTED;
    if (M == 0) return AP_OK;
    const int tiles = (rows_per_batch + CE_TN - 1) / CE_TN;
    const int64_t blocks = (M / rows_per_batch) * tiles;
    const size_t lds = (size_t)CE_TN * (C | 1) * sizeof(float);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_soft_ce, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, logits, ldx, target, t_sb, t_sc,
                       t_sn, rows_per_batch, row_loss, dlogits, grad_scale, M, C, tiles, mix_lam, mix_batches, mix_lam_dev);
    return ap_check_launch();
}

extern "C" int ap_soft_ce_sparse_fwd_bwd(const ap_bf16* logits, int ldx, const int* idx, const float* val, int K, int64_t p_sb, int64_t p_sn,
                                         int rows_per_batch, float smoothing, float* row_loss, ap_bf16* dlogits, float grad_scale,
                                         int64_t M, int C, float mix_lam, int mix_batches, ap_stream_t stream) {
    return ap_soft_ce_sparse_fwd_bwd_dev(logits, ldx, idx, val, K, p_sb, p_sn, rows_per_batch, smoothing, row_loss, dlogits, grad_scale, M, C, mix_lam, mix_batches,
                                         nullptr, stream);
}

extern "C" int ap_soft_ce_sparse_fwd_bwd_dev(const ap_bf16* logits, int ldx, const int* idx, const float* val, int K, int64_t p_sb, int64_t p_sn,
                                             int rows_per_batch, float smoothing, float* row_loss, ap_bf16* dlogits, float grad_scale,
                                             int64_t M, int C, float mix_lam, int mix_batches, const float* mix_lam_dev, ap_stream_t stream) {
    if (!logits || !idx || !val || !row_loss || !dlogits) return AP_ERR_NULL;
    if (C <= 0 || ldx < C || (ldx & 7) || rows_per_batch <= 0 || M < 0 || K <= 0 || K > CE_MAXK || smoothing < 0.f || smoothing >= 1.f) return AP_ERR_SHAPE;
    if (mix_batches != 0 && (mix_batches < 0 || (int64_t)mix_batches * rows_per_batch != M || 2 * K > CE_MAXK)) return AP_ERR_SHAPE;
    if (ldx > 64 * 2 * CE_MAXV) return AP_ERR_UNSUPPORTED;
    if (M == 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_soft_ce_sparse, dim3((unsigned)((M + 4 * CE_SR - 1) / (4 * CE_SR))), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(logits), ldx,
                       idx, val, K, p_sb, p_sn, rows_per_batch, smoothing, row_loss, reinterpret_cast<bf16_t*>(dlogits), grad_scale, M, C, mix_lam, mix_batches, mix_lam_dev);
    return ap_check_launch();
}
