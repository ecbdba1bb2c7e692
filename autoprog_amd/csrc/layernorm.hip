// LayerNorm forward / backward for token-major bf16 activations (nn.LayerNorm semantics:
// biased variance, fp32 statistics, affine) -- reference call sites models/volo.py:122,131,
// 213,221,290,297,550.  HBM-bound: one pass over x (fwd), one pass over (dy, x[, dres]) (bwd).
//
// Mapping: a row is owned by a group of G lanes (G = 16/32/64, power of two >= C/8), every lane
// holds V 16-byte chunks (8 channels) of the row; reductions are wavefront shuffles (xor < G).
// Rows are grid-strided so every lane keeps the same channels and accumulates dgamma/dbeta in
// registers; those are reduced through LDS per block and added with fp32 atomics.
#include "common.h"
#include <cstdlib>

template <int V, int U>
__global__ void __launch_bounds__(256)
k_ln_fwd(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
         bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
         int64_t rows, int C, int G, float eps,
         unsigned char* __restrict__ y8 = nullptr, const float* __restrict__ q_scale = nullptr, float* __restrict__ q_amax = nullptr) {
    // y8 (nullable): the same output a second time as OCP e4m3 bytes, y8 = sat(bf16(y) * q_scale[0]), and q_amax[0] = max(q_amax[0],
    // max |bf16(y)|) -- the operand of an fp8 GEMM without a quantisation pass of its own (ap_layernorm_fwd_fp8)
    const float qs = y8 ? q_scale[0] : 1.f;
    float qmx = 0.f;
    // U row-iterations are loaded before any is reduced (the per-row chain load -> shuffles -> store is latency bound)
    const int lane_in_group = threadIdx.x & (G - 1);
    const int groups_per_block = 256 / G;
    const int group = threadIdx.x / G;
    const int nchunks = C >> 3;
    const float invC = 1.0f / (float)C;
    // gamma/beta: one coalesced pass per block into LDS, then each lane keeps its 8*V channels in registers
    // (per-lane global loads of the affine parameters were as many instructions as the row data itself)
    extern __shared__ __attribute__((aligned(16))) float sgb[];     // [C] gamma | [C] beta
    for (int c = threadIdx.x; c < C; c += 256) { sgb[c] = gamma[c]; sgb[C + c] = beta[c]; }
    __syncthreads();
    float gam[V][8], bet[V][8];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = lane_in_group + i * G;
        if (ch < nchunks) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sgb + 8 * ch), g1 = *reinterpret_cast<const f32x4*>(sgb + 8 * ch + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sgb + C + 8 * ch), b1 = *reinterpret_cast<const f32x4*>(sgb + C + 8 * ch + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { gam[i][k] = g0[k]; gam[i][4 + k] = g1[k]; bet[i][k] = b0[k]; bet[i][4 + k] = b1[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { gam[i][k] = 0.f; bet[i][k] = 0.f; }
        }
    }
    const int64_t row_stride = (int64_t)gridDim.x * groups_per_block;
    for (int64_t row0 = (int64_t)blockIdx.x * groups_per_block + group; row0 < rows; row0 += row_stride * U) {
        u32x4 rx[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + u * row_stride;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                const u32x4 z = {0u, 0u, 0u, 0u};
                rx[u][i] = (row < rows && ch < nchunks) ? ld16(x + row * C + 8 * ch) : z;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + u * row_stride;
            if (row >= rows) continue;
            float v[V][8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                unpack8(rx[u][i], v[i]);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[i][k];
            }
            for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mu = s * invC;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                if (ch < nchunks) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
                }
            }
            for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            const float rs = rsqrtf(q * invC + eps);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                if (ch < nchunks) {
                    float o8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) o8[k] = (v[i][k] - mu) * rs * gam[i][k] + bet[i][k];
                    const u32x4 ob = pack8(o8);
                    st16_nt(y + row * C + 8 * ch, ob);
                    if (y8) {
                        float r8[8];
                        unpack8(ob, r8);
                        u32x2 o;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            float c4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { qmx = fmaxf(qmx, fabsf(r8[4 * h2 + e])); c4[e] = fminf(fmaxf(r8[4 * h2 + e] * qs, -448.f), 448.f); }
                            int w = 0;
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[0], c4[1], w, false);
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[2], c4[3], w, true);
                            o[h2] = (unsigned)w;
                        }
                        *reinterpret_cast<u32x2*>(y8 + row * C + 8 * ch) = o;
                    }
                }
            }
            if (lane_in_group == 0) { mean[row] = mu; rstd[row] = rs; }
        }
    }
    if (y8 && q_amax) {                       // one atomic per workgroup, and only when it raises the value (see k_quantize_fp8)
        __syncthreads();
        qmx = group_max<64>(qmx);
        if ((threadIdx.x & 63) == 0) sgb[threadIdx.x >> 6] = qmx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(sgb[0], sgb[1]), fmaxf(sgb[2], sgb[3]));
            if (__float_as_int(m) > *reinterpret_cast<volatile int*>(q_amax)) atomicMax(reinterpret_cast<int*>(q_amax), __float_as_int(m));
        }
    }
}

// Round 4: the forward at LOW register pressure.  k_ln_fwd<3, 2> holds the lane's 48 affine parameters and two rows in registers: 116
// VGPRs, four workgroups per CU -- and the training step launches it with 6.1 workgroups per CU (one row per lane group), i.e. a
// second, half-empty round that costs a whole load -> reduce -> store chain again.  Here the parameters stay in LDS (as float4 planes
// [gamma lo | gamma hi | beta lo | beta hi][chunk]: consecutive lanes read consecutive 16-byte words, no bank conflicts) and are read
// where they are used; the lane's row chunks are requested FIRST and the parameter loads behind them wait on a counted vmcnt, so the
// parameters' trip through LDS runs under the rows' memory latency.  <= 72 VGPRs: seven workgroups per CU, one round.
template <int V>
__global__ void __launch_bounds__(256, 7)
k_ln_fwd_lp(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
            bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int C, int G, float eps,
            unsigned char* __restrict__ y8 = nullptr, const float* __restrict__ q_scale = nullptr, float* __restrict__ q_amax = nullptr) {
    // y8 / q_scale / q_amax: the e4m3 side output of ap_layernorm_fwd_fp8, as in k_ln_fwd
    const float qs = y8 ? q_scale[0] : 1.f;
    float qmx = 0.f;
    extern __shared__ __attribute__((aligned(16))) float sgb[];     // 4 planes x [nchunks] float4
    const int lane_in_group = threadIdx.x & (G - 1);
    const int groups_per_block = 256 / G;
    const int group = threadIdx.x / G;
    const int nchunks = C >> 3;
    const float invC = 1.0f / (float)C;
    const int64_t row_stride = (int64_t)gridDim.x * groups_per_block;
    const u32x4 z = {0u, 0u, 0u, 0u};
    // parameter loads first in program order (they are waited for first), the row right behind them
    float gt[8], bt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int c = threadIdx.x + k * 256; gt[k] = c < C ? gamma[c] : 0.f; bt[k] = c < C ? beta[c] : 0.f; }
    int64_t row = (int64_t)blockIdx.x * groups_per_block + group;
    u32x4 rx[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = lane_in_group + i * G;
        rx[i] = (row < rows && ch < nchunks) ? ld16(x + row * C + 8 * ch) : z;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = threadIdx.x + k * 256;
        if (c < C) {
            const int idx = (((c >> 2) & 1) * nchunks + (c >> 3)) * 4 + (c & 3);
            sgb[idx] = gt[k];
            sgb[8 * nchunks + idx] = bt[k];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const f32x4* pl = reinterpret_cast<const f32x4*>(sgb);
    while (row < rows) {
        {
            float v[V][8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                unpack8(rx[i], v[i]);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[i][k];
            }
            for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mu = s * invC;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                if (ch < nchunks) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q = __builtin_fmaf(d, d, q); }     // (an explicit fma: mlp_fused.hip's in-kernel LayerNorm repeats this sum bit for bit)
                }
            }
            for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            const float rs = rsqrtf(q * invC + eps);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                if (ch < nchunks) {
                    const f32x4 g0 = pl[ch], g1 = pl[nchunks + ch], b0 = pl[2 * nchunks + ch], b1 = pl[3 * nchunks + ch];
                    float o8[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o8[k] = (v[i][k] - mu) * rs * g0[k] + b0[k];
                        o8[4 + k] = (v[i][4 + k] - mu) * rs * g1[k] + b1[k];
                    }
                    const u32x4 ob = pack8(o8);
                    st16_nt(y + row * C + 8 * ch, ob);
                    if (y8) {
                        float r8[8];
                        unpack8(ob, r8);
                        u32x2 o;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            float c4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { qmx = fmaxf(qmx, fabsf(r8[4 * h2 + e])); c4[e] = fminf(fmaxf(r8[4 * h2 + e] * qs, -448.f), 448.f); }
                            int w = 0;
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[0], c4[1], w, false);
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[2], c4[3], w, true);
                            o[h2] = (unsigned)w;
                        }
                        *reinterpret_cast<u32x2*>(y8 + row * C + 8 * ch) = o;
                    }
                }
            }
            if (lane_in_group == 0) { mean[row] = mu; rstd[row] = rs; }
        }
        row += row_stride;                       // (a lane group's next row, if the grid was capped: all lanes of a group stay together)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int ch = lane_in_group + i * G;
            rx[i] = (row < rows && ch < nchunks) ? ld16(x + row * C + 8 * ch) : z;
        }
    }    if (y8 && q_amax) {                       // one atomic per workgroup, and only when it raises the value (see k_quantize_fp8)
        __syncthreads();
        qmx = group_max<64>(qmx);
        if ((threadIdx.x & 63) == 0) sgb[threadIdx.x >> 6] = qmx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(sgb[0], sgb[1]), fmaxf(sgb[2], sgb[3]));
            if (__float_as_int(m) > *reinterpret_cast<volatile int*>(q_amax)) atomicMax(reinterpret_cast<int*>(q_amax), __float_as_int(m));
        }
    }
}

template <int V, int U>
__global__ void __launch_bounds__(256)
k_ln_bwd(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
         const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
         bf16_t* __restrict__ dx, float* __restrict__ partial,
         int64_t rows, int C, int G) {
    // U rows per lane group are in flight per iteration: the row loop is a dependent chain
    // load -> shuffle reduce -> store, so memory-level parallelism has to come from unrolling rows
    // (the U=1 version ran at ~40 % of the HBM rate with 12 sequential iterations per wave)
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves][C] x 2
    const int lane_in_group = threadIdx.x & (G - 1);
    const int groups_per_block = 256 / G;
    const int group = threadIdx.x / G;
    const int nchunks = C >> 3;
    const float invC = 1.0f / (float)C;
    float gam[V][8], ag[V][8], ab[V][8];
    for (int c = threadIdx.x; c < C; c += 256) red[c] = gamma[c];      // staged through LDS (see k_ln_fwd); `red` is reused below
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = lane_in_group + i * G;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ag[i][k] = 0.f; ab[i][k] = 0.f; gam[i][k] = (ch < nchunks) ? red[8 * ch + k] : 0.f; }
    }
    __syncthreads();
    const int64_t row_stride = (int64_t)gridDim.x * groups_per_block;
    for (int64_t row0 = (int64_t)blockIdx.x * groups_per_block + group; row0 < rows; row0 += row_stride * U) {
        u32x4 rdy[U][V], rx[U][V], rres[U][V];
        float mu[U], rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + u * row_stride;
            const bool rok = row < rows;
            mu[u] = rok ? mean[row] : 0.f;
            rs[u] = rok ? rstd[row] : 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                const bool ok = rok && ch < nchunks;
                const u32x4 z = {0u, 0u, 0u, 0u};
                rdy[u][i] = ok ? ld16(dy + row * C + 8 * ch) : z;
                rx[u][i] = ok ? ld16(x + row * C + 8 * ch) : z;
                rres[u][i] = (ok && dres != nullptr) ? ld16(dres + row * C + 8 * ch) : z;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + u * row_stride;
            if (row >= rows) continue;
            float g[V][8], xh[V][8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                float d8[8], x8[8];
                unpack8(rdy[u][i], d8);
                unpack8(rx[u][i], x8);
                const bool ok = ch < nchunks;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    xh[i][k] = ok ? (x8[k] - mu[u]) * rs[u] : 0.f;
                    g[i][k] = d8[k] * gam[i][k];
                    s1 += g[i][k];
                    s2 += g[i][k] * xh[i][k];
                    ag[i][k] += d8[k] * xh[i][k];
                    ab[i][k] += d8[k];
                }
            }
            for (int o = G >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            s1 *= invC; s2 *= invC;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int ch = lane_in_group + i * G;
                if (ch < nchunks) {
                    float o8[8];
                    unpack8(rres[u][i], o8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o8[k] += rs[u] * (g[i][k] - s1 - xh[i][k] * s2);
                    st16_nt(dx + row * C + 8 * ch, pack8(o8));
                }
            }
        }
    }
    // dgamma/dbeta partials: lanes with the same lane_in_group hold the same channels -> butterfly over the 64/G lane
    // groups of the wave, then the 4 waves of the block meet in LDS and the block writes ONE partial row
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k)
            for (int o = G; o < 64; o <<= 1) { ag[i][k] += __shfl_xor(ag[i][k], o, 64); ab[i][k] += __shfl_xor(ab[i][k], o, 64); }
    float* rg = red;                       // [4 waves][C]
    float* rb = red + 4 * C;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < G) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int ch = lane_in_group + i * G;
            if (ch < nchunks) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { rg[wave * C + 8 * ch + k] = ag[i][k]; rb[wave * C + 8 * ch + k] = ab[i][k]; }
            }
        }
    }
    __syncthreads();
    // per-block partials go to a workspace with plain stores; k_ln_bwd_reduce sums them.  (Atomics from
    // ~1000 workgroups into the same 2*C floats ran at the contended-atomic rate and were 2/3 of the kernel.)
    float* prow = partial + (int64_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        prow[c] = rg[c] + rg[C + c] + rg[2 * C + c] + rg[3 * C + c];
        prow[C + c] = rb[c] + rb[C + c] + rb[2 * C + c] + rb[3 * C + c];
    }
}

// Round 5: the same backward as a software pipeline.  k_ln_bwd runs "request U rows -> wait -> reduce -> store" trips one after the other,
// and the three workgroups of a CU start together, so the CU alternates between a burst of requests and a phase with nothing in flight
// (4.5 TB/s at 25088 x 384).  Here a lane group requests trip t+1 BEFORE it reduces trip t (two register sets of U rows, the loop unrolled
// by two so that both are static), and every load is unconditional at a clamped address (a guarded load compiles to an exec-masked
// block that zeroes its destination: DESIGN section 3, round 1) -- rows past the end are skipped in the reduce, idle chunk lanes
// (C/8 < G) are masked to zero after the load.  gamma comes straight from global memory (two 16-byte loads per chunk, L2 hits, under
// the first trip's latency): no LDS trip and no barrier in front of the rows.
// POOL: the incoming gradient of row (b, y, x) of a [B, H, W, C] token grid is dy + pool[b, y / 2, x / 2] / count -- the backward of the
// 2 x 2 ceil-mode average pool that reads the SAME LayerNorm output (OutlookAttention: models/volo.py:75,87; count = the number of
// pixels the pooled cell covers: 4, or 2 / 1 on the last odd row / column).  Was a pass of its own over the 77 MB gradient
// (ap_avgpool2_bwd_acc, 19 us per outlooker block); here it is one more 16-byte load per chunk from a tensor a quarter of the size.
struct LnPool { const bf16_t* grad; int H, W, h, w; unsigned magic_hw, magic_w; };
template <int V, int U, int OCC = 1, bool POOL = false>
__global__ void __launch_bounds__(256, OCC)
k_ln_bwd_pf(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
            const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
            bf16_t* __restrict__ dx, float* __restrict__ partial,
            int64_t rows, int C, int G, LnPool pool = LnPool{nullptr, 0, 0, 0, 0, 0u, 0u}) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves][C] x 2 (the end of the kernel only)
    const int lane_in_group = threadIdx.x & (G - 1);
    const int groups_per_block = 256 / G;
    const int group = threadIdx.x / G;
    const int nchunks = C >> 3;
    const float invC = 1.0f / (float)C;
    const bool has_res = dres != nullptr;
    float gam[V][8], ag[V][8], ab[V][8];
    int choff[V];                                  // element offset of the lane's chunk i inside a row (clamped into the row)
    unsigned chmask[V];                            // all ones where the chunk exists
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = lane_in_group + i * G;
        const bool ok = ch < nchunks;
        choff[i] = 8 * (ok ? ch : nchunks - 1);
        chmask[i] = ok ? 0xffffffffu : 0u;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + choff[i]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + choff[i] + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { gam[i][k] = ok ? g0[k] : 0.f; gam[i][4 + k] = ok ? g1[k] : 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { ag[i][k] = 0.f; ab[i][k] = 0.f; }
    }
    const int64_t row_stride = (int64_t)gridDim.x * groups_per_block;
    const int64_t trip = row_stride * U;
    const int64_t last = rows - 1;

#define LN_PF_LOAD(R0, DY, X, RES, MU, RS, PG, PI)                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                           \
        int64_t row = (R0) + u * row_stride;                                                                  \
        row = row < rows ? row : last;                                                                        \
        MU[u] = mean[row]; RS[u] = rstd[row];                                                                 \
        int64_t prow = 0;                                                                                     \
        if constexpr (POOL) {                                                                                 \
            const unsigned r32 = (unsigned)row;                                                               \
            const unsigned b = __umulhi(r32, pool.magic_hw), rem = r32 - b * (unsigned)(pool.H * pool.W);     \
            const unsigned yy = __umulhi(rem, pool.magic_w), xx = rem - yy * (unsigned)pool.W;                \
            const int cy = min(2, pool.H - (int)(yy & ~1u)), cx = min(2, pool.W - (int)(xx & ~1u));           \
            PI[u] = 1.0f / (float)(cy * cx);                                                                  \
            prow = ((int64_t)b * pool.h + (yy >> 1)) * pool.w + (xx >> 1);                                    \
        }                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < V; ++i) {                                                       \
            DY[u][i] = ld16(dy + row * C + choff[i]);                                                         \
            X[u][i] = ld16(x + row * C + choff[i]);                                                           \
            if (has_res) RES[u][i] = ld16(dres + row * C + choff[i]);                                         \
            if constexpr (POOL) PG[u][i] = ld16(pool.grad + prow * C + choff[i]);                             \
        }                                                                                                     \
    }

#define LN_PF_COMPUTE(R0, DY, X, RES, MU, RS, PG, PI)                                                        \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                           \
        const int64_t row = (R0) + u * row_stride;                                                            \
        if (row < rows) {                                                                                     \
            float g[V][8], xh[V][8];                                                                          \
            float s1 = 0.f, s2 = 0.f;                                                                         \
            _Pragma("unroll") for (int i = 0; i < V; ++i) {                                                   \
                float d8[8], x8[8];                                                                           \
                u32x4 dm = DY[u][i];                                                                          \
                dm[0] &= chmask[i]; dm[1] &= chmask[i]; dm[2] &= chmask[i]; dm[3] &= chmask[i];               \
                unpack8(dm, d8);                                                                              \
                if constexpr (POOL) {                                                                         \
                    float p8[8];                                                                              \
                    u32x4 pm_ = PG[u][i];                                                                     \
                    pm_[0] &= chmask[i]; pm_[1] &= chmask[i]; pm_[2] &= chmask[i]; pm_[3] &= chmask[i];       \
                    unpack8(pm_, p8);                                                                         \
                    _Pragma("unroll") for (int k = 0; k < 8; ++k) d8[k] = fmaf(p8[k], PI[u], d8[k]);          \
                }                                                                                             \
                unpack8(X[u][i], x8);                                                                         \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                               \
                    xh[i][k] = (x8[k] - MU[u]) * RS[u];                                                       \
                    g[i][k] = d8[k] * gam[i][k];                                                              \
                    s1 += g[i][k];                                                                            \
                    s2 += g[i][k] * xh[i][k];                                                                 \
                    ag[i][k] += d8[k] * xh[i][k];                                                             \
                    ab[i][k] += d8[k];                                                                        \
                }                                                                                             \
            }                                                                                                 \
            for (int o = G >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); } \
            s1 *= invC; s2 *= invC;                                                                           \
            _Pragma("unroll") for (int i = 0; i < V; ++i) {                                                   \
                if (chmask[i]) {                                                                              \
                    float o8[8];                                                                              \
                    if (has_res) unpack8(RES[u][i], o8);                                                      \
                    else { _Pragma("unroll") for (int k = 0; k < 8; ++k) o8[k] = 0.f; }                       \
                    _Pragma("unroll") for (int k = 0; k < 8; ++k) o8[k] += RS[u] * (g[i][k] - s1 - xh[i][k] * s2); \
                    st16_nt(dx + row * C + choff[i], pack8(o8));                                              \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
    }

    u32x4 a_dy[U][V], a_x[U][V], a_res[U][V], b_dy[U][V], b_x[U][V], b_res[U][V];
    u32x4 a_pg[POOL ? U : 1][POOL ? V : 1], b_pg[POOL ? U : 1][POOL ? V : 1];
    float a_mu[U], a_rs[U], b_mu[U], b_rs[U], a_pi[U], b_pi[U];
    int64_t row0 = (int64_t)blockIdx.x * groups_per_block + group;
    LN_PF_LOAD(row0, a_dy, a_x, a_res, a_mu, a_rs, a_pg, a_pi)
    for (; row0 < rows; row0 += 2 * trip) {
        const int64_t row1 = row0 + trip;
        LN_PF_LOAD(row1, b_dy, b_x, b_res, b_mu, b_rs, b_pg, b_pi)
        LN_PF_COMPUTE(row0, a_dy, a_x, a_res, a_mu, a_rs, a_pg, a_pi)
        if (row1 >= rows) break;
        LN_PF_LOAD(row1 + trip, a_dy, a_x, a_res, a_mu, a_rs, a_pg, a_pi)
        LN_PF_COMPUTE(row1, b_dy, b_x, b_res, b_mu, b_rs, b_pg, b_pi)
    }
#undef LN_PF_LOAD
#undef LN_PF_COMPUTE
    // dgamma/dbeta partials: as k_ln_bwd (butterfly over the wave's lane groups, the four waves meet in LDS, one partial row per block)
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k)
            for (int o = G; o < 64; o <<= 1) { ag[i][k] += __shfl_xor(ag[i][k], o, 64); ab[i][k] += __shfl_xor(ab[i][k], o, 64); }
    float* rg = red;                       // [4 waves][C]
    float* rb = red + 4 * C;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < G) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if (chmask[i]) {
                f32x4 w0, w1, w2, w3;
#pragma unroll
                for (int k = 0; k < 4; ++k) { w0[k] = ag[i][k]; w1[k] = ag[i][4 + k]; w2[k] = ab[i][k]; w3[k] = ab[i][4 + k]; }
                *reinterpret_cast<f32x4*>(rg + wave * C + choff[i]) = w0;
                *reinterpret_cast<f32x4*>(rg + wave * C + choff[i] + 4) = w1;
                *reinterpret_cast<f32x4*>(rb + wave * C + choff[i]) = w2;
                *reinterpret_cast<f32x4*>(rb + wave * C + choff[i] + 4) = w3;
            }
        }
    }
    __syncthreads();
    float* prow = partial + (int64_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        prow[c] = rg[c] + rg[C + c] + rg[2 * C + c] + rg[3 * C + c];
        prow[C + c] = rb[c] + rb[C + c] + rb[2 * C + c] + rb[3 * C + c];
    }
}

// out[c] += sum_b partial[b][c] for the 2*C columns (dgamma | dbeta); 32 columns x 32 row slices per block,
// independent loads unrolled x8 so the column sums are not a serial chain of L2 round trips
__global__ void __launch_bounds__(1024)
k_ln_bwd_reduce(const float* __restrict__ partial, int nblocks, int C2, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
    __shared__ float red[32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (c < C2) {
        int b = ry;
        for (; b + 7 * 32 < nblocks; b += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * 32) * C2 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 32) s += partial[(int64_t)b * C2 + c];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C2) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][cx];
        if (c < C) dgamma[c] += t; else dbeta[c - C] += t;
    }
}

// lane-group width G (8..64) and chunks per lane V (1..4) with G*V >= C/8 and the fewest idle lanes; ties go to the
// narrower group (more rows per wave in flight).  C = 384 -> (16,3), 192 -> (8,3), 768 -> (32,3), 256 -> (8,4), 1000 -> (32,4)
static int pick_group(int C, int* V) {
    const int nch = C / 8;
    int bestG = 64, bestV = 4, best_waste = 1 << 30;
    for (int G = 8; G <= 64; G <<= 1)
        for (int v = 1; v <= 4; ++v) {
            if (G * v < nch) continue;
            const int waste = G * v - nch;
            if (waste < best_waste) { best_waste = waste; bestG = G; bestV = v; }
        }
    *V = bestV;
    return bestG;
}

// backward keeps 3 operand rows per row in registers next to the dgamma/dbeta accumulators: one chunk per lane and a
// wide group (G >= C/8) with 4 rows unrolled measured faster there (22.7 vs 27.4 us at 25088 x 384) than the
// no-idle-lane mapping above, whose register footprint halves the occupancy
static int pick_group_wide(int C, int* V) {
    const int nch = C / 8;
    int G = 16;
    while (G < 64 && G < nch) G <<= 1;
    const int v = (nch + G - 1) / G;
    *V = v <= 1 ? 1 : (v <= 2 ? 2 : 4);
    return G;
}

extern "C" {

int ap_layernorm_fwd(const ap_bf16* x, const float* gamma, const float* beta, ap_bf16* y, float* mean, float* rstd,
                     int64_t rows, int C, float eps, ap_stream_t stream) {
    return ap_layernorm_fwd_fp8(x, gamma, beta, y, nullptr, nullptr, nullptr, mean, rstd, rows, C, eps, stream);
}

int ap_layernorm_fwd_fp8(const ap_bf16* x, const float* gamma, const float* beta, ap_bf16* y, unsigned char* y8, const float* q_scale,
                         float* q_amax, float* mean, float* rstd, int64_t rows, int C, float eps, ap_stream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return AP_ERR_NULL;
    if (y8 && !q_scale) return AP_ERR_NULL;
    if (C <= 0 || (C & 7)) return AP_ERR_SHAPE;
    if (C > 2048) return AP_ERR_UNSUPPORTED;
    if (rows <= 0) return AP_OK;
    static int rpg = 0, wide = -1;
    if (rpg == 0) { const char* e = getenv("AP_LN_FWD_RPG"); rpg = e ? atoi(e) : 1; if (rpg < 1) rpg = 1;       // one row per lane group and workgroup: 10.3 vs 10.7 us at 25088 x 384 (4: 11.4, 8: 15.3)
                    const char* w = getenv("AP_LN_FWD_WIDE"); wide = w ? atoi(w) : 0; }
    int V; const int G = wide ? pick_group_wide(C, &V) : pick_group(C, &V);
    const int gpb = 256 / G;
    int64_t grid = ceil_div64(rows, (int64_t)gpb * rpg);
    if (grid > 256 * 8) grid = 256 * 8;
    if (grid < 1) grid = 1;
    const size_t lds = (size_t)2 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    static int lp = -1;
    if (lp < 0) { const char* e = getenv("AP_LN_FWD_LP"); lp = e ? atoi(e) : 1; }
    if (lp && V <= 3) {          // the low-register forward (the widths it was measured on: 192, 384, 768)
        if (V == 1) hipLaunchKernelGGL((k_ln_fwd_lp<1>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
        else if (V == 2) hipLaunchKernelGGL((k_ln_fwd_lp<2>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
        else hipLaunchKernelGGL((k_ln_fwd_lp<3>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
        return ap_check_launch();
    }
    if (V == 1) hipLaunchKernelGGL((k_ln_fwd<1, 4>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
    else if (V == 2) hipLaunchKernelGGL((k_ln_fwd<2, 4>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
    else if (V == 3) hipLaunchKernelGGL((k_ln_fwd<3, 2>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
    else hipLaunchKernelGGL((k_ln_fwd<4, 2>), dim3((int)grid), dim3(256), lds, s, x, gamma, beta, y, mean, rstd, rows, C, G, eps, y8, q_scale, q_amax);
    return ap_check_launch();
}

// up to AP_LN_MAX_BATCH dgamma/dbeta reductions in one launch (blockIdx.y = reduction): the LayerNorms of one block
struct LnReduceBatch { const float* partial[AP_LN_MAX_BATCH]; float* dgamma[AP_LN_MAX_BATCH]; float* dbeta[AP_LN_MAX_BATCH];
                       int nblocks[AP_LN_MAX_BATCH]; int C[AP_LN_MAX_BATCH]; };
__global__ void __launch_bounds__(1024)
k_ln_bwd_reduce_batched(LnReduceBatch bt) {
    __shared__ float red[32][33];
    const int y = blockIdx.y;
    const float* __restrict__ partial = bt.partial[y];
    const int nblocks = bt.nblocks[y], C = bt.C[y], C2 = 2 * C;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    if (blockIdx.x * 32 >= C2) return;
    float s = 0.f;
    if (c < C2) {
        int b = ry;
        for (; b + 7 * 32 < nblocks; b += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * 32) * C2 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 32) s += partial[(int64_t)b * C2 + c];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C2) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][cx];
        if (c < C) bt.dgamma[y][c] += t; else bt.dbeta[y][c - C] += t;
    }
}

static thread_local int* g_ln_defer_blocks = nullptr;       // set by ap_layernorm_bwd_partial around its call of ap_layernorm_bwd
static thread_local LnPool g_ln_pool = {nullptr, 0, 0, 0, 0, 0u, 0u};     // set by ap_layernorm_bwd_partial_pool likewise

size_t ap_layernorm_bwd_workspace(int64_t rows, int C) {
    (void)rows;
    return (size_t)1024 * 2 * (size_t)C * sizeof(float);
}

int ap_layernorm_bwd(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* mean, const float* rstd,
                     const ap_bf16* dres, ap_bf16* dx, float* dgamma, float* dbeta, int64_t rows, int C,
                     void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return AP_ERR_NULL;
    if (ws_bytes < ap_layernorm_bwd_workspace(rows, C)) return AP_ERR_SHAPE;
    float* partial = static_cast<float*>(workspace);
    if (C <= 0 || (C & 7)) return AP_ERR_SHAPE;
    if (C > 2048) return AP_ERR_UNSUPPORTED;
    if (rows <= 0) return AP_OK;
    int V; const int G = pick_group_wide(C, &V);
    const int gpb = 256 / G;
    int64_t grid = ceil_div64(rows, gpb);
    static int grid_cap = 0;
    if (grid_cap == 0) { const char* e = getenv("AP_LN_BWD_GRID"); grid_cap = e ? atoi(e) : 768; if (grid_cap < 1 || grid_cap > 1024) grid_cap = 768; }   // 3 blocks per CU measured best (20.7 vs 23.5 us at 1024)
    if (grid > grid_cap) grid = grid_cap;   // bounds the dgamma/dbeta partial rows (workspace holds 1024)
    if (V >= 2 && grid > 512 && !getenv("AP_LN_BWD_GRID")) grid = 512;      // 768-wide rows (DeiT-Base, VOLO-D5): 19.6 us at two workgroups per CU, 24.3 at three
    const size_t lds = (size_t)2 * 4 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    // (5 rows per trip instead of 4 -- every lane group of the VOLO-D1 launch then finishes its 8 or 9 rows in two trips instead of three for
    // a sixth of them -- measured equal, 17.5 vs 17.7 us: the trips are not what the launch waits for)
    static int pf = -1;             // AP_LN_BWD_PF: 0 = the trip-by-trip kernel of rounds 1 - 4; 1 = pipelined, half the rows per trip (the same registers in flight);
    if (pf < 0) { const char* e = getenv("AP_LN_BWD_PF"); pf = e ? atoi(e) : 1; }        //               2 = pipelined with the old rows per trip (twice the registers)
    if (g_ln_pool.grad) {           // the average pool's gradient rides in (pipelined kernel, one chunk per lane: C <= 512)
        if (!pf || V != 1) return AP_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((k_ln_bwd_pf<1, 2, 1, true>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G, g_ln_pool);
    } else
    if (pf && V <= 2) {             // (four chunks per lane: two register sets do not fit 256 registers)
        if (V == 1 && pf == 3) hipLaunchKernelGGL((k_ln_bwd_pf<1, 2, 4>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
        else if (V == 1) { if (pf == 2) hipLaunchKernelGGL((k_ln_bwd_pf<1, 4>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
                      else hipLaunchKernelGGL((k_ln_bwd_pf<1, 2>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G); }
        else { if (pf == 2) hipLaunchKernelGGL((k_ln_bwd_pf<2, 2>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
               else hipLaunchKernelGGL((k_ln_bwd_pf<2, 1>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G); }
    } else
    if (V == 1) hipLaunchKernelGGL((k_ln_bwd<1, 4>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
    else if (V == 2) hipLaunchKernelGGL((k_ln_bwd<2, 2>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
    else if (V == 3) hipLaunchKernelGGL((k_ln_bwd<3, 2>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
    else hipLaunchKernelGGL((k_ln_bwd<4, 1>), dim3((int)grid), dim3(256), lds, s, dy, x, gamma, mean, rstd, dres, dx, partial, rows, C, G);
    int rc = ap_check_launch();
    if (rc != AP_OK) return rc;
    if (g_ln_defer_blocks) { *g_ln_defer_blocks = (int)grid; return AP_OK; }      // ap_layernorm_bwd_partial: the caller batches the reduction
    hipLaunchKernelGGL(k_ln_bwd_reduce, dim3((2 * C + 31) / 32), dim3(1024), 0, s, partial, (int)grid, 2 * C, dgamma, dbeta, C);
    return ap_check_launch();
}

int ap_layernorm_bwd_partial(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* mean, const float* rstd,
                             const ap_bf16* dres, ap_bf16* dx, int64_t rows, int C, void* workspace, size_t ws_bytes, int* n_partial,
                             ap_stream_t stream) {
    if (!n_partial) return AP_ERR_NULL;
    float dummy = 0.f;                                   // dgamma / dbeta are not touched on this path
    g_ln_defer_blocks = n_partial;
    *n_partial = 0;
    const int rc = ap_layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, &dummy, &dummy, rows, C, workspace, ws_bytes, stream);
    g_ln_defer_blocks = nullptr;
    return rc;
}

int ap_layernorm_bwd_partial_pool(const ap_bf16* dy, const ap_bf16* pool_grad, int B, int H, int W, const ap_bf16* x, const float* gamma,
                                  const float* mean, const float* rstd, const ap_bf16* dres, ap_bf16* dx, int C, void* workspace, size_t ws_bytes,
                                  int* n_partial, ap_stream_t stream) {
    if (!pool_grad) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || (int64_t)B * H * W > 0x7fffffff) return AP_ERR_SHAPE;
    const unsigned hw = (unsigned)(H * W);
    if (hw < 2 || W < 2) return AP_ERR_UNSUPPORTED;                  // (no exact 32-bit magic for a division by 1; no caller needs it)
    LnPool p;
    p.grad = pool_grad; p.H = H; p.W = W; p.h = (H + 1) / 2; p.w = (W + 1) / 2;
    p.magic_hw = (unsigned)(0xFFFFFFFFu / hw) + 1u;                  // exact for rows < 2^32 / (H W)
    p.magic_w = (unsigned)(0xFFFFFFFFu / (unsigned)W) + 1u;
    if ((uint64_t)B * hw >= (uint64_t)(0xFFFFFFFFu / hw)) return AP_ERR_UNSUPPORTED;
    g_ln_pool = p;
    const int rc = ap_layernorm_bwd_partial(dy, x, gamma, mean, rstd, dres, dx, (int64_t)B * H * W, C, workspace, ws_bytes, n_partial, stream);
    g_ln_pool.grad = nullptr;
    return rc;
}

int ap_layernorm_bwd_reduce_batched(const ap_ln_reduce* items, int count, ap_stream_t stream) {
    if (!items) return AP_ERR_NULL;
    if (count <= 0 || count > AP_LN_MAX_BATCH) return AP_ERR_SHAPE;
    LnReduceBatch bt;
    int cmax = 0;
    for (int i = 0; i < count; ++i) {
        if (!items[i].partial || !items[i].dgamma || !items[i].dbeta) return AP_ERR_NULL;
        if (items[i].n_partial <= 0 || items[i].C <= 0) return AP_ERR_SHAPE;
        bt.partial[i] = items[i].partial; bt.dgamma[i] = items[i].dgamma; bt.dbeta[i] = items[i].dbeta;
        bt.nblocks[i] = items[i].n_partial; bt.C[i] = items[i].C;
        if (items[i].C > cmax) cmax = items[i].C;
    }
    for (int i = count; i < AP_LN_MAX_BATCH; ++i) { bt.partial[i] = bt.partial[0]; bt.dgamma[i] = bt.dgamma[0]; bt.dbeta[i] = bt.dbeta[0]; bt.nblocks[i] = 0; bt.C[i] = 0; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_ln_bwd_reduce_batched, dim3((2 * cmax + 31) / 32, count), dim3(1024), 0, (hipStream_t)stream, bt);
    return ap_check_launch();
}

}  // extern "C"
