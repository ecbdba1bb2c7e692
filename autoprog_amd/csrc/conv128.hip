// 3x3 / stride 1 / pad 1 convolution on NHWC bf16 feature maps with 128 channels in and out: the two
// `nn.Conv2d(hidden, hidden, 3, 1, 1, bias=False)` of PatchEmbed at stem_hidden_dim = 128 -- VOLO-D5, BASELINE configs[4]
// (reference models/volo.py:355-367, 799-821) -- forward and input gradient (the same kernel on flipped / transposed weights).
// The weight gradient is csrc/conv.hip's 64-channel kernel on the four (output half, input half) quadrants (ap_conv3x3_c128_wgrad).
//
// At 64 channels (csrc/conv.hip) all 9 x 64 x 64 weights AND a 34 x 18 patch sit in LDS; at 128 channels the weights alone are
// 288 KB.  Here the PATCH stays and the weights stream: an implicit GEMM, M = pixels, N = 128 output channels, K = 9 taps x 128 input
// channels, cut into 18 K-slabs of (tap, 64 input channels) = a [128 co][64 ci] weight panel of 16 KB each.
//   workgroup : 512 threads (8 waves, two per SIMD), one per CU, persistent over 16 x 16-pixel output tiles
//   LDS       : the 18 x 18-pixel input patch of the tile (pixel stride 128 + 8 elements: consecutive pixels start 4 banks apart, a tap
//               is a CONSTANT byte offset) 88 KB + two weight slabs (16-byte chunks XORed with row & 7) 32 KB
//   wave      : (pg, ch) = (wave & 3, wave >> 2): tile rows 4 pg .. 4 pg + 3 (four 16-pixel fragments) x output channels 64 ch .. + 63
//               (four 16-channel fragments): 16 accumulator tiles; per slab 8 weight + 8 pixel fragment reads for 32 MFMAs
//   stream    : step s requests slab s + 1 (of the same 18: the ring runs across tiles) into registers and ONE 16-byte chunk per thread
//               of the NEXT tile's patch, computes slab s, drops the weights into the other buffer, one barrier.  The weights come from
//               L2 (every workgroup streams the same 288 KB per tile: 37 GB/s per CU at the MFMA rate), the patch from HBM (1.27x halo).
//   epilogue  : accumulators -> bf16 -> the patch region as a [256 px][128] tile -> 16 bytes per lane along pixel rows (whole 256-byte
//               pixel rows leave the CU); STATS: per-channel sum / sum of squares of the ROUNDED outputs, one partial row per workgroup
//               (the layout ap_bn_relu_fwd_partials reads), as csrc/conv.hip.
// 2 * 9 * 128 * 128 FLOP per pixel: 947 GFLOP per call at B = 64, 224 x 224 (the D5 stem at 448 px).
#include "common.h"
#include "gemm_epi.h"
#include <cstdlib>

#define C8_C 128
#define C8_T 16
#define C8_PW (C8_T + 2)
#define C8_NPIX (C8_PW * C8_PW)                       // 324 patch pixels
#define C8_PSTR 136                                   // patch pixel stride in elements
#define C8_NPRE 11                                    // 16-byte patch chunks per thread: ceil(324 * 16 / 512)
#define C8_SLAB (C8_C * 64)                           // elements of a weight slab [128 rows][64 k]
#define C8_NSLAB 18
#define C8_WELEMS (C8_NSLAB * C8_SLAB)                // 9 * 128 * 128
#define C8_LDS_BYTES ((C8_NPIX * C8_PSTR + 2 * C8_SLAB) * 2)
#define C8_LDS_STATS (32 * 2 * C8_C * 4)              // STATS: the per-thread partial sums live in LDS (16 registers would spill next to 64 + 44 + 32)

// fp32 OIHW [128][128][3][3] -> bf16 slabs: forward [tap * 2 + ci / 64][co][ci % 64]; input gradient (dx = conv3x3(dy, W^T, taps
// flipped)) [(8 - tap) * 2 + co / 64][ci][co % 64]; one thread per weight
__global__ void __launch_bounds__(256)
k_conv3x3_c128_pack(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C8_WELEMS) return;
    const int tap = idx % 9, ci = (idx / 9) % C8_C, co = idx / (9 * C8_C);
    const bf16_t v = f2bf(w[idx]);
    wf[((tap * 2 + (ci >> 6)) * C8_C + co) * 64 + (ci & 63)] = v;
    wb[(((8 - tap) * 2 + (co >> 6)) * C8_C + ci) * 64 + (co & 63)] = v;
}

template <bool STATS>
__global__ void __launch_bounds__(512)
k_conv3x3_c128(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, bf16_t* __restrict__ y, int H, int W,
               int tiles_x, int tiles_y, int ntiles, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) bf16_t c8_smem[];
    bf16_t* const P = c8_smem;                                  // patch [324][136]; the output tile [256][136] in the epilogue
    bf16_t* const Wl = c8_smem + C8_NPIX * C8_PSTR;             // two weight slabs
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, ch = wave >> 2;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // fragment bases (elements): pixel (tile row 4 pg + pt + ky, column fr + kx), K chunk g of the slab's 64 channels; weight row 64 ch + 16 cot + fr
    const int pbase = (4 * pg * C8_PW + fr) * C8_PSTR + g * 8;
    const int wrow = ch * 64 + fr;
    // (swizzle key (row >> 1) & 7: rows of 128 bytes alternate between the two halves of the banks, so the eight rows of one parity that a
    // 16-lane ds_read_b128 group reads need eight different chunks; row & 7 gave rows fr and fr + 8 the same chunk AND parity: two-way conflicts)
    const int wb0 = wrow * 64 + ((g ^ ((fr >> 1) & 7)) << 3), wb1 = wrow * 64 + (((4 + g) ^ ((fr >> 1) & 7)) << 3);
    // staging assignment: 16-byte chunk c16 of pixel p0 + 32 i (patch: i < 11; output tile: i < 8)
    const int c16 = tid & 15;
    int p0 = tid >> 4;                           // laundered once per tile: the 11 per-chunk (row, column) pairs are 3 VALU each; hoisted out of
                                                 // the tile loop they would be 22 registers held for the whole kernel
    // weight slab: chunks tid and tid + 512 of the slab's 1024
    int wdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 512 * j, row = idx >> 3, c = idx & 7;
        wdst[j] = row * 64 + ((c ^ ((row >> 1) & 7)) << 3);
    }
    auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        b = t / (tiles_x * tiles_y); ty0 = ty * C8_T; tx0 = tx * C8_T;
    };
    u32x4 pre[C8_NPRE], wreg[2];
    // patch chunk i of this thread.  The load is UNCONDITIONAL at a clamped address; the zero padding is applied when the chunk goes to LDS
    auto gload1 = [&](int i, int b, int ty0, int tx0) {
        const int pix = min(p0 + 32 * i, C8_NPIX - 1);
        const int py = (pix * 3641) >> 16, px = pix - py * C8_PW;            // pix / 18 (exact below 1170)
        const int gy = min(max(ty0 - 1 + py, 0), H - 1), gx = min(max(tx0 - 1 + px, 0), W - 1);
        pre[i] = ld16(x + ((int64_t)b * H + gy) * W * C8_C + (unsigned)(gx * C8_C + c16 * 8));
    };
    auto pstore = [&](int b, int ty0, int tx0) {
#pragma unroll
        for (int i = 0; i < C8_NPRE; ++i) {
            const int pix = p0 + 32 * i;
            if (pix < C8_NPIX) {
                const int py = (pix * 3641) >> 16, px = pix - py * C8_PW;
                const unsigned gy = (unsigned)(ty0 - 1 + py), gx = (unsigned)(tx0 - 1 + px);
                st16(P + pix * C8_PSTR + c16 * 8, (gy < (unsigned)H && gx < (unsigned)W) ? pre[i] : zero4);
            }
        }
    };
    auto wload = [&](int s) {
#pragma unroll
        for (int j = 0; j < 2; ++j) wreg[j] = ld16(wp + (int64_t)s * C8_SLAB + (tid + 512 * j) * 8);
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) st16(Wl + buf * C8_SLAB + wdst[j], wreg[j]);
    };
    float* const sred = reinterpret_cast<float*>(c8_smem + C8_NPIX * C8_PSTR + 2 * C8_SLAB);     // [32 p0][2][128]: this thread owns (p0, 8 c16 .. + 7)
    if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sred[((tid >> 4) * 2 + 0) * C8_C + c16 * 8 + k] = 0.f; sred[((tid >> 4) * 2 + 1) * C8_C + c16 * 8 + k] = 0.f; }
    }

    int t = blockIdx.x;
    if (t >= ntiles) return;                                     // (whole workgroup; before any barrier)
    {   // prologue: the first tile's patch and slab 0
        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
#pragma unroll
        for (int i = 0; i < C8_NPRE; ++i) gload1(i, b, ty0, tx0);
        wload(0);
        pstore(b, ty0, tx0);
        wstore(0);
        __syncthreads();
    }
    int cur = 0;
    for (; t < ntiles; t += gridDim.x) {
        int b, ty0, tx0, nb = 0, nty0 = 0, ntx0 = 0;
        tile_origin(t, b, ty0, tx0);
        const int tn = t + gridDim.x;
        const bool has_next = tn < ntiles;                       // uniform
        asm volatile("" : "+v"(p0));
        if (has_next) tile_origin(tn, nb, nty0, ntx0);
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < C8_NSLAB; ++s) {
            const int tap = s >> 1, cih = s & 1, ky = tap / 3, kx = tap - 3 * ky;
            wload(s + 1 == C8_NSLAB ? 0 : s + 1);                // the ring runs across tiles: slab 0 of the next tile
            if (s < C8_NPRE && has_next) gload1(s, nb, nty0, ntx0);
            const bf16_t* wl = Wl + cur * C8_SLAB;
            // one 32-deep K half at a time: 4 + 4 fragments (32 registers) live next to the 64 accumulators and the 44 of the prefetched patch
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 af[4], bfr[4];
#pragma unroll
                for (int cot = 0; cot < 4; ++cot) af[cot] = ld16(wl + (kb ? wb1 : wb0) + cot * 16 * 64);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) bfr[pt] = ld16(P + pbase + ((pt + ky) * C8_PW + kx) * C8_PSTR + cih * 64 + kb * 32);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int cot = 0; cot < 4; ++cot)
                        acc[pt][cot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(af[cot]), as_bf16x8(bfr[pt]), acc[pt][cot], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);               // (keeps the second half's reads from being hoisted above the first half's MFMAs)
            }
            wstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        // ---- epilogue: every wave has read its last fragments (the barrier above); the patch region becomes the output tile.
        // lane (fr, g) of tile (pt, cot) holds pixel (row 4 pg + pt, column fr), channels 64 ch + 16 cot + 4 g .. + 3
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int cot = 0; cot < 4; ++cot) {
                u32x2 pk;
                pk[0] = pack_bf2(acc[pt][cot][0], acc[pt][cot][1]); pk[1] = pack_bf2(acc[pt][cot][2], acc[pt][cot][3]);
                *reinterpret_cast<u32x2*>(P + ((4 * pg + pt) * 16 + fr) * C8_PSTR + ch * 64 + cot * 16 + 4 * g) = pk;
            }
        __syncthreads();
        float ssum[8], ssq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = p0 + 32 * i, prow = p >> 4, pcol = p & 15;
            const int gy = ty0 + prow, gx = tx0 + pcol;
            if (gy < H && gx < W) {
                const u32x4 v = ld16(P + p * C8_PSTR + c16 * 8);
                st16_nt(y + ((int64_t)b * H + gy) * W * C8_C + (unsigned)(gx * C8_C + c16 * 8), v);
                if constexpr (STATS) {
                    float r8[8];
                    unpack8(v, r8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { ssum[k] += r8[k]; ssq[k] = fmaf(r8[k], r8[k], ssq[k]); }
                }
            }
        }
        if constexpr (STATS) {
            float* mine = sred + (tid >> 4) * 2 * C8_C + c16 * 8;
            f32x4 a0 = *reinterpret_cast<f32x4*>(mine), a1 = *reinterpret_cast<f32x4*>(mine + 4);
            f32x4 q0 = *reinterpret_cast<f32x4*>(mine + C8_C), q1 = *reinterpret_cast<f32x4*>(mine + C8_C + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += ssum[k]; a1[k] += ssum[4 + k]; q0[k] += ssq[k]; q1[k] += ssq[4 + k]; }
            *reinterpret_cast<f32x4*>(mine) = a0; *reinterpret_cast<f32x4*>(mine + 4) = a1;
            *reinterpret_cast<f32x4*>(mine + C8_C) = q0; *reinterpret_cast<f32x4*>(mine + C8_C + 4) = q1;
        }
        __syncthreads();
        if (has_next) {
            pstore(nb, nty0, ntx0);
            __syncthreads();
        }
    }
    if constexpr (STATS) {
        // 32 threads (tid >> 4 = 0 .. 31) hold partial sums of the same 8 channels: one partial row [2][128] per workgroup
        __syncthreads();
        if (tid < 2 * C8_C) {
            float tsum = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) tsum += sred[r * 2 * C8_C + tid];
            stats[(int64_t)blockIdx.x * 2 * C8_C + tid] = tsum;
        }
    }
}

extern "C" {

static int c8_grid(int ntiles) {
    static int cap = 0;
    if (cap == 0) {
        const char* e = getenv("AP_CONV128_GRID");
        cap = e ? atoi(e) : 0;
        if (cap < 1) {
            int dev = 0; hipDeviceProp_t pr;
            cap = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        }
    }
    return ntiles < cap ? ntiles : cap;
}

int ap_conv3x3_c128_pack(const float* w_oihw, ap_bf16* w_fwd, ap_bf16* w_bwd, ap_stream_t stream) {
    if (!w_oihw || !w_fwd || !w_bwd) return AP_ERR_NULL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_conv3x3_c128_pack, dim3((C8_WELEMS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, w_fwd, w_bwd);
    return ap_check_launch();
}

int ap_conv3x3_c128_stat_rows(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t nt = (int64_t)B * ((W + C8_T - 1) / C8_T) * ((H + C8_T - 1) / C8_T);
    return c8_grid((int)(nt > 0x7fffffff ? 0x7fffffff : nt));
}

int ap_conv3x3_c128(const ap_bf16* x, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats, ap_stream_t stream) {
    if (!x || !w_packed || !y) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    const int tiles_x = (W + C8_T - 1) / C8_T, tiles_y = (H + C8_T - 1) / C8_T;
    const int64_t nt64 = (int64_t)B * tiles_x * tiles_y;
    if (nt64 > 0x7fffffff || (int64_t)W * C8_C > 0x7fffffff) return AP_ERR_SHAPE;
    const int ntiles = (int)nt64;
    static int attr_done = 0;
    (void)hipGetLastError();
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS_BYTES + C8_LDS_STATS) != hipSuccess) return AP_ERR_LAUNCH;
        attr_done = 1;
    }
    const int grid = c8_grid(ntiles);
    if (stats) hipLaunchKernelGGL((k_conv3x3_c128<true>), dim3(grid), dim3(512), C8_LDS_BYTES + C8_LDS_STATS, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats);
    else hipLaunchKernelGGL((k_conv3x3_c128<false>), dim3(grid), dim3(512), C8_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats);
    return ap_check_launch();
}

}  // extern "C"
