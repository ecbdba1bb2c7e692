// First convolution of the VOLO stem: 7x7 / stride 2 / pad 3, 3 -> 64 channels, no bias (reference models/volo.py:355-357,
// nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False)), forward and weight gradient (the input gradient is never needed: the input is the image).
//
// Space-to-depth turns the strided convolution into a stride-1 one the MFMA path of conv.hip can walk with constant offsets:
//   xs[b][Y][X][(sy,sx,c)] = x[b][2Y+sy][2X+sx][c]        12 channels, stored as 16 (bf16, 32 B per pixel; written by
//                                                          ap_resize_bilinear_s2d16, which also does the per-step input resize)
//   y[b][oy][ox][co] = sum_{dy,dx in 0..3} sum_{c16} xs[b][oy-2+dy][ox-2+dx][c16] * W8[co][dy][dx][c16]
// with W8[co][dy][dx][(sy,sx,c)] = w[co][c][2dy+sy-1][2dx+sx-1] (zero where an index is -1): the 7x7 kernel padded to 8x8.
// K = 16 taps x 16 channels = 256 (147 of them real): one MFMA K step (32) is two horizontally adjacent taps.
//
// Forward: persistent 256-thread workgroups, 32 x 16 output tiles, LDS = packed weights (32 KB) + 35 x 19 pixel patch (21 KB):
// two workgroups per CU overlap each other's load / compute / store phases.  HBM-bound by the 205 MB of output (B = 128, 224 px).
// Weight gradient: 16 x 16 pixel tiles, wave w owns kernel row dy = w (4 taps x 64 output channels = 16 accumulator tiles); the
// pixel axis is the reduction axis, fragments come from transposed LDS reads; per-workgroup slabs, ordered reduction (no atomics).
#include "common.h"
#include "gemm_epi.h"
#include <cstdlib>

#define S_CO 64
#define S_CI 16
#define S_TR 32
#define S_TW 16
#define S_PH (S_TR + 3)
#define S_PW (S_TW + 3)
#define S_NPIX (S_PH * S_PW)                           // 665
#define S_WELEMS (16 * S_CO * S_CI)                    // packed weights: [8 K steps][64 co][32]
#define S_LDS_BYTES ((S_WELEMS + S_NPIX * S_CI) * 2)
#define S_NPRE ((S_NPIX * 2 + 255) / 256)              // 16-byte chunks of the patch per thread (2 per pixel): 6

// w fp32 [64][3][7][7] -> bf16 [s = 2 dy + dxp][co][t2 * 16 + c16]  (tap (dy, dx = 2 dxp + t2), c16 = (sy * 2 + sx) * 3 + c)
__global__ void __launch_bounds__(256)
k_conv7_pack(const float* __restrict__ w, bf16_t* __restrict__ wp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S_WELEMS) return;
    const int k32 = idx & 31, co = (idx >> 5) & 63, s = idx >> 11;
    const int dy = s >> 1, dx = 2 * (s & 1) + (k32 >> 4), c16 = k32 & 15;
    float v = 0.f;
    if (c16 < 12) {
        const int sy = c16 / 6, sx = (c16 / 3) & 1, c = c16 % 3;
        const int ky = 2 * dy + sy - 1, kx = 2 * dx + sx - 1;
        if (ky >= 0 && kx >= 0) v = w[((co * 3 + c) * 7 + ky) * 7 + kx];
    }
    wp[idx] = f2bf(v);
}

template <bool STATS>
__global__ void __launch_bounds__(256, 2)
k_conv7_s2d(const bf16_t* __restrict__ xs, const bf16_t* __restrict__ wp, bf16_t* __restrict__ y, int H, int W,
            int tiles_x, int tiles_y, int ntiles, float* __restrict__ stats, int ldy = S_CO) {
    // ldy: pixel stride of y in elements (64; 128 when y is one 64-channel half of the 128-wide stem's tensor, ap_conv7_s2d_ld)
    extern __shared__ __attribute__((aligned(16))) bf16_t s7_smem[];
    bf16_t* Wl = s7_smem;                       // [8][64 co][32]
    bf16_t* P = s7_smem + S_WELEMS;             // [665 px][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (int idx = tid; idx < S_WELEMS / 8; idx += 256) st16(Wl + idx * 8, ld16(wp + (int64_t)idx * 8));
    int wbase[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 32 * (q >> 1) + 8 * (fr >> 2) + 4 * (q & 1) + (fr & 3);       // N-permuted rows: 16-byte output chunks (gemm.hip "direct epilogue")
        wbase[q] = r * 32 + g * 8;
    }
    const int abase = (wave * S_PW + fr + (g >> 1)) * S_CI + (g & 1) * 8;
    float ssum[2][8], ssq[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) { ssum[pr][k] = 0.f; ssq[pr][k] = 0.f; }

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int ty0 = ty * S_TR, tx0 = tx * S_TW;
        const bf16_t* img = xs + (int64_t)b * H * W * S_CI;
        u32x4 pre[S_NPRE];
#pragma unroll
        for (int i = 0; i < S_NPRE; ++i) {
            const int ch = tid + 256 * i, pix = min(ch >> 1, S_NPIX - 1);
            const int py = pix / S_PW, px = pix - py * S_PW;
            const int gy = min(max(ty0 - 2 + py, 0), H - 1), gx = min(max(tx0 - 2 + px, 0), W - 1);
            pre[i] = ld16(img + (unsigned)((gy * W + gx) * S_CI + (ch & 1) * 8));
        }
        __syncthreads();                         // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < S_NPRE; ++i) {
            const int ch = tid + 256 * i, pix = ch >> 1;
            const int py = pix / S_PW, px = pix - py * S_PW;
            const unsigned gy = (unsigned)(ty0 - 2 + py), gx = (unsigned)(tx0 - 2 + px);
            if (pix < S_NPIX) st16(P + pix * S_CI + (ch & 1) * 8, (gy < (unsigned)H && gx < (unsigned)W) ? pre[i] : zero4);
        }
        __syncthreads();
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nrow = (H - ty0 - wave + 3) >> 2;                   // rows wave + 4 i of the tile that lie inside the image
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int dy = s >> 1, dx0 = 2 * (s & 1);
            bf16x8 bfr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bfr[q] = as_bf16x8(ld16(Wl + s * S_CO * 32 + wbase[q]));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < nrow) {
                    const bf16x8 a = as_bf16x8(ld16(P + abase + ((4 * i + dy) * S_PW + dx0) * S_CI));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[q], a, acc[i][q], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        bf16_t* out = y + (((int64_t)b * H + ty0 + wave) * W + tx0 + fr) * ldy + 8 * g;
        const bool col_ok = tx0 + fr < W;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nrow && col_ok) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
                    v[0] = acc[i][2 * pr][0]; v[1] = acc[i][2 * pr][1]; v[2] = acc[i][2 * pr][2]; v[3] = acc[i][2 * pr][3];
                    v[4] = acc[i][2 * pr + 1][0]; v[5] = acc[i][2 * pr + 1][1]; v[6] = acc[i][2 * pr + 1][2]; v[7] = acc[i][2 * pr + 1][3];
                    const u32x4 pk = pack8(v);
                    st16(out + (int64_t)(4 * i) * W * ldy + 32 * pr, pk);
                    if constexpr (STATS) {
                        float r8[8];
                        unpack8(pk, r8);
#pragma unroll
                        for (int k = 0; k < 8; ++k) { ssum[pr][k] += r8[k]; ssq[pr][k] += r8[k] * r8[k]; }
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(P);                    // [4 waves][2][64]
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { ssum[pr][k] += __shfl_xor(ssum[pr][k], o, 64); ssq[pr][k] += __shfl_xor(ssq[pr][k], o, 64); }
            }
        if (fr == 0) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    red[(wave * 2 + 0) * S_CO + 32 * pr + 8 * g + k] = ssum[pr][k];
                    red[(wave * 2 + 1) * S_CO + 32 * pr + 8 * g + k] = ssq[pr][k];
                }
        }
        __syncthreads();
        if (tid < 2 * S_CO)
            stats[(int64_t)blockIdx.x * 2 * S_CO + tid] = red[tid] + red[2 * S_CO + tid] + red[4 * S_CO + tid] + red[6 * S_CO + tid];
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
#define SW_T 16
#define SW_PW (SW_T + 3)
#define SW_DPIX (SW_T * SW_T)
#define SW_APIX (SW_PW * SW_PW)                        // 361
#define SW_LDS_BYTES (SW_DPIX * S_CO * 2 + SW_APIX * S_CI * 2)
#define SW_SLAB (16 * S_CO * S_CI)                     // [tap][co][c16] floats per workgroup
__device__ __forceinline__ int sw_key(int col) { return (((col >> 1) & 1) << 1) | (((col >> 3) & 1) << 2); }      // as cw_key (conv.hip)

__global__ void __launch_bounds__(256, 2)
k_conv7_s2d_wgrad(const bf16_t* __restrict__ xs, const bf16_t* __restrict__ dz, float* __restrict__ slab, int H, int W,
                  int tiles_x, int tiles_y, int ntiles, int lddz = S_CO) {
    extern __shared__ __attribute__((aligned(16))) bf16_t sw_smem[];
    bf16_t* D = sw_smem;                        // [256 px][64 co], chunk ^ sw_key(column)
    bf16_t* A = sw_smem + SW_DPIX * S_CO;       // [361 px][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4, q = fr >> 2, p = fr & 3;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    int dbase[4][2], abase[4][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int col = 8 * (g & 1) + q + 4 * hf;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            dbase[t][hf] = ((g >> 1) * SW_T + col) * S_CO + (((2 * t + (p >> 1)) ^ sw_key(col)) << 3) + (p & 1) * 4;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
            abase[dx][hf] = (((g >> 1) + wave) * SW_PW + col + dx) * S_CI + p * 4;       // kernel row dy = wave
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto trfrag = [&](const bf16_t* base0, const bf16_t* base1) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base1));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f32x4 acc[4][4];                            // [dx][co tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c8 = (tid & 7) * 8, cidx = tid & 7;
    // the next tile's loads are in flight during the MFMA steps of the current one, and the fragments of K step k+1 are read while
    // the MFMAs of step k issue (as k_conv3x3_c64_wgrad_p)
    u32x4 rd[8], ra[3];
    auto origin = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        b = t / (tiles_x * tiles_y); ty0 = ty * SW_T; tx0 = tx * SW_T;
    };
    auto gload = [&](int t) {
        int b, ty0, tx0;
        origin(t, b, ty0, tx0);
        const bf16_t* ximg = xs + (int64_t)b * H * W * S_CI;
        const bf16_t* dimg = dz + (int64_t)b * H * W * lddz;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int px = (tid >> 3) + 32 * i, r = px >> 4, c = px & 15;
            const int gy = min(ty0 + r, H - 1), gx = min(tx0 + c, W - 1);
            rd[i] = ld16(dimg + (unsigned)((gy * W + gx) * lddz + c8));
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ch = tid + 256 * i, px = min(ch >> 1, SW_APIX - 1), r = px / SW_PW, c = px - r * SW_PW;
            const int gy = min(max(ty0 - 2 + r, 0), H - 1), gx = min(max(tx0 - 2 + c, 0), W - 1);
            ra[i] = ld16(ximg + (unsigned)((gy * W + gx) * S_CI + (ch & 1) * 8));
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) gload(t);
    for (; t < ntiles; t += gridDim.x) {
        int b, ty0, tx0;
        origin(t, b, ty0, tx0);
        __syncthreads();                         // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int px = (tid >> 3) + 32 * i, r = px >> 4, c = px & 15;
            st16(D + px * S_CO + ((cidx ^ sw_key(c)) << 3), ((ty0 + r < H) && (tx0 + c < W)) ? rd[i] : zero4);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ch = tid + 256 * i, px = ch >> 1, r = px / SW_PW, c = px - r * SW_PW;
            const unsigned gy = (unsigned)(ty0 - 2 + r), gx = (unsigned)(tx0 - 2 + c);
            if (px < SW_APIX) st16(A + px * S_CI + (ch & 1) * 8, (gy < (unsigned)H && gx < (unsigned)W) ? ra[i] : zero4);
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) gload(t + gridDim.x);
        bf16x8 df[2][4], af[2][4];
        auto fload = [&](int k, bf16x8* dfr, bf16x8* afr) {
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) dfr[tq] = trfrag(D + dbase[tq][0] + 2 * k * SW_T * S_CO, D + dbase[tq][1] + 2 * k * SW_T * S_CO);
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) afr[dx] = trfrag(A + abase[dx][0] + 2 * k * SW_PW * S_CI, A + abase[dx][1] + 2 * k * SW_PW * S_CI);
        };
        fload(0, df[0], af[0]);
#pragma unroll
        for (int k = 0; k < SW_T / 2; ++k) {
            if (k + 1 < SW_T / 2) fload(k + 1, df[(k + 1) & 1], af[(k + 1) & 1]);
#pragma unroll
            for (int dx = 0; dx < 4; ++dx)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) acc[dx][tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[k & 1][tq], af[k & 1][dx], acc[dx][tq], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // slab[wg][tap = 4 dy + dx][co][c16]; lane holds co = 16 tq + 4 g + r, c16 = fr
    float* mine = slab + (int64_t)blockIdx.x * SW_SLAB;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mine[((wave * 4 + dx) * S_CO + 16 * tq + 4 * g + r) * S_CI + fr] = acc[dx][tq][r];
}

// w.grad[co][c][ky][kx] (fp32 [64][3][7][7]) += sum over the slabs, in a fixed order.  Block = 256 consecutive slab elements x 16 waves;
// wave v adds slabs v, v + 16, ... with 16-byte coalesced loads, eight in flight; the 16 partial sums meet in LDS and are added in wave
// order, then each slab element (tap, co, c16) that is a real weight goes to its OIHW place.  (First version: one thread per real
// weight walking all slabs with scattered 4-byte loads, 60 us; the same scheme as k_conv3x3_wgrad_reduce.)
__global__ void __launch_bounds__(1024)
k_conv7_wgrad_reduce(const float* __restrict__ slab, int nslab, float* __restrict__ dw) {
    __shared__ float4 part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = blockIdx.x * 256 + lane * 4;              // SW_SLAB = 64 * 256
    float4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int w = wave;
    for (; w + 7 * 16 < nslab; w += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(slab + (int64_t)(w + 16 * u) * SW_SLAB + e0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s[u & 1].x += v[u].x; s[u & 1].y += v[u].y; s[u & 1].z += v[u].z; s[u & 1].w += v[u].w; }
    }
    for (; w < nslab; w += 16) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (int64_t)w * SW_SLAB + e0);
        s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
    part[wave][lane] = make_float4(s[0].x + s[1].x, s[0].y + s[1].y, s[0].z + s[1].z, s[0].w + s[1].w);
    __syncthreads();
    if (threadIdx.x < 256) {
        const int e = blockIdx.x * 256 + threadIdx.x;        // slab element (tap, co, c16)
        const float* pf = reinterpret_cast<const float*>(&part[0][0]);
        float t = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) t += pf[v * 256 + threadIdx.x];
        const int c16 = e % S_CI, co = (e / S_CI) % S_CO, tap = e / (S_CI * S_CO);
        // tap = (ky8 >> 1) * 4 + (kx8 >> 1), c16 = ((ky8 & 1) * 2 + (kx8 & 1)) * 3 + c with (ky8, kx8) = (ky + 1, kx + 1)
        const int par = c16 / 3, c = c16 - 3 * par;
        const int ky = 2 * (tap >> 2) + (par >> 1) - 1, kx = 2 * (tap & 3) + (par & 1) - 1;
        if (c16 < 12 && ky >= 0 && ky < 7 && kx >= 0 && kx < 7) dw[((co * 3 + c) * 7 + ky) * 7 + kx] += t;
    }
}

extern "C" {

static int s7_grid(int ntiles, const char* env, int dflt) {
    const char* e = getenv(env);
    int cap = e ? atoi(e) : dflt;
    if (cap < 1) cap = dflt;
    return ntiles < cap ? ntiles : cap;
}
static int s7_tiles(int B, int H, int W, int tr, int tw, int* tx, int* ty) {
    *tx = (W + tw - 1) / tw; *ty = (H + tr - 1) / tr;
    const int64_t nt = (int64_t)B * *tx * *ty;
    return nt > 0x7fffffff ? -1 : (int)nt;
}

int ap_conv7_pack(const float* w_oihw, ap_bf16* w_packed, ap_stream_t stream) {
    if (!w_oihw || !w_packed) return AP_ERR_NULL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_conv7_pack, dim3(S_WELEMS / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, w_packed);
    return ap_check_launch();
}

int ap_conv7_s2d_stat_rows(int B, int H, int W) {
    int tx, ty;
    const int nt = (B > 0 && H > 0 && W > 0) ? s7_tiles(B, H, W, S_TR, S_TW, &tx, &ty) : 0;
    return nt > 0 ? s7_grid(nt, "AP_CONV7_GRID", 512) : 0;
}

int ap_conv7_s2d(const ap_bf16* xs, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats, ap_stream_t stream) {
    return ap_conv7_s2d_ld(xs, w_packed, y, S_CO, B, H, W, stats, stream);
}

int ap_conv7_s2d_ld(const ap_bf16* xs, const ap_bf16* w_packed, ap_bf16* y, int ldy, int B, int H, int W, float* stats, ap_stream_t stream) {
    if (!xs || !w_packed || !y) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || ldy < S_CO || (ldy & 7) || (int64_t)H * W * ldy > 0x7fffffff) return AP_ERR_SHAPE;
    int tx, ty;
    const int nt = s7_tiles(B, H, W, S_TR, S_TW, &tx, &ty);
    if (nt < 0) return AP_ERR_SHAPE;
    const int grid = s7_grid(nt, "AP_CONV7_GRID", 512);
    static int attr_done = 0;
    (void)hipGetLastError();
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv7_s2d<false>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv7_s2d<true>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        attr_done = 1;
    }
    if (stats) hipLaunchKernelGGL((k_conv7_s2d<true>), dim3(grid), dim3(256), S_LDS_BYTES, (hipStream_t)stream, xs, w_packed, y, H, W, tx, ty, nt, stats, ldy);
    else hipLaunchKernelGGL((k_conv7_s2d<false>), dim3(grid), dim3(256), S_LDS_BYTES, (hipStream_t)stream, xs, w_packed, y, H, W, tx, ty, nt, stats, ldy);
    return ap_check_launch();
}

size_t ap_conv7_s2d_wgrad_workspace(int B, int H, int W) {
    int tx, ty;
    const int nt = (B > 0 && H > 0 && W > 0) ? s7_tiles(B, H, W, SW_T, SW_T, &tx, &ty) : 0;
    return nt > 0 ? (size_t)s7_grid(nt, "AP_CONV7_WGRAD_GRID", 512) * SW_SLAB * sizeof(float) : 0;
}

int ap_conv7_s2d_wgrad(const ap_bf16* xs, const ap_bf16* dz, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                       ap_stream_t stream) {
    return ap_conv7_s2d_wgrad_ld(xs, dz, S_CO, dw_oihw, B, H, W, workspace, ws_bytes, stream);
}

int ap_conv7_s2d_wgrad_ld(const ap_bf16* xs, const ap_bf16* dz, int lddz, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                          ap_stream_t stream) {
    if (!xs || !dz || !dw_oihw || !workspace) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || lddz < S_CO || (lddz & 7) || (int64_t)H * W * lddz > 0x7fffffff) return AP_ERR_SHAPE;
    int tx, ty;
    const int nt = s7_tiles(B, H, W, SW_T, SW_T, &tx, &ty);
    if (nt < 0 || ws_bytes < ap_conv7_s2d_wgrad_workspace(B, H, W)) return AP_ERR_SHAPE;
    const int grid = s7_grid(nt, "AP_CONV7_WGRAD_GRID", 512);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_conv7_s2d_wgrad, dim3(grid), dim3(256), SW_LDS_BYTES, (hipStream_t)stream, xs, dz, static_cast<float*>(workspace), H, W, tx, ty, nt, lddz);
    int rc = ap_check_launch();
    if (rc != AP_OK) return rc;
    static_assert(SW_SLAB % 256 == 0, "whole reduction blocks");
    hipLaunchKernelGGL(k_conv7_wgrad_reduce, dim3(SW_SLAB / 256), dim3(1024), 0, (hipStream_t)stream, static_cast<const float*>(workspace), grid, dw_oihw);
    return ap_check_launch();
}

}  // extern "C"
