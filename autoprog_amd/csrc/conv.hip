// 3x3 / stride 1 / pad 1 convolution of the VOLO stem on NHWC bf16 feature maps with 64 channels in and out
// (reference models/volo.py:355-367: the two `nn.Conv2d(hidden, hidden, 3, 1, 1, bias=False)` of PatchEmbed), forward,
// input gradient (the same kernel on flipped / transposed weights) and weight gradient.
//
// Forward is an implicit GEMM: M = B*H*W pixels, N = 64 output channels, K = 9 taps x 64 input channels, with NO im2col
// buffer -- a tap is an address offset into the input patch held in LDS.
//   workgroup   : 512 threads (8 waves: two per SIMD; the 4-wave shape of rounds 2 - 3 described below is NW = 4, AP_CONV_WAVES=4),
//                 PERSISTENT over the 32 x 16 pixel output tiles of the batch
//   LDS         : all 9 x 64 x 64 weights (72 KB, loaded once per workgroup) + the 34 x 18 pixel input patch of the
//                 current tile (76.5 KB); 16-byte chunks XOR-swizzled so that the MFMA fragment reads are conflict free
//   wave        : (NW = 4) tile rows w, w+4, ... (8 rows of 16 pixels) x all 64 output channels: 8 x 4 accumulator tiles; per
//                 (tap, 32-channel K step) it reads 4 weight fragments + 8 pixel fragments for 32 MFMAs (96 B/clk of LDS
//                 traffic per CU at full MFMA rate, against a 128 B/clk LDS)
//   pipelining  : the next tile's patch is loaded into registers (20 x 16 B per thread) before the current tile is
//                 computed and written to LDS after it -- one workgroup per CU, so the overlap is within the workgroup
// HBM traffic: input 1.22x (tile halo), output 1x; 118 GFLOP per D1 call (B = 128, 112 x 112) = 47 us at the dense bf16 peak,
// 70 us at 6 TB/s: the kernel is bound by neither until the MFMA issue rate of 4 waves per CU is reached.
#include "common.h"
#include "gemm_epi.h"
#include <cstdlib>
#include <type_traits>

#ifndef AP_EXPERIMENTS
#define AP_EXPERIMENTS 0
#endif
#define CV_C 64
#define CV_TR 32                  // tile rows
#define CV_TW 16                  // tile columns (one MFMA fragment of 16 pixels per row)
#define CV_PH (CV_TR + 2)
#define CV_PW (CV_TW + 2)
#define CV_NPIX (CV_PH * CV_PW)                       // 612 patch pixels
#define CV_NCHUNK (CV_NPIX * 8)                       // 16-byte chunks of the patch
#define CV_NPRE ((CV_NCHUNK + 255) / 256)             // chunks per thread (20)
#define CV_WELEMS (9 * CV_C * CV_C)
#define CV_PSTR 72                                    // patch pixel stride in elements: 64 channels + 16 B of padding -- consecutive pixels start
                                                      // 4 banks apart, so the 8 lanes of a ds_read_b128 group cover all 32 banks, and a tap is a
                                                      // CONSTANT byte offset (an XOR swizzle keyed on the pixel would need address math per tap)
#define CV_LDS_BYTES ((CV_WELEMS + CV_NPIX * CV_PSTR) * 2)

// fp32 OIHW [64][64][3][3] -> bf16 [tap][co][ci] (forward) and [tap'][ci][co] with the taps flipped (input gradient:
// dx = conv3x3(dy, W^T flipped)); one thread per weight
__global__ void __launch_bounds__(256)
k_conv3x3_pack(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CV_WELEMS) return;
    const int tap = idx % 9, ci = (idx / 9) % CV_C, co = idx / (9 * CV_C);
    const bf16_t v = f2bf(w[idx]);
    wf[(tap * CV_C + co) * CV_C + ci] = v;
    wb[((8 - tap) * CV_C + ci) * CV_C + co] = v;
}

// Swizzle key of the weight image (rows of 128 bytes: a row's bank base is (row & 1) * 32).  The 16 rows a ds_read_b128 lane group reads are
// 8 q + 4 b + p (q, p = 0 .. 3): their parity is p & 1, so the eight rows of one parity need eight different chunks -- key = (p >> 1) | q << 1.
// (Rounds 2 - 4 used the GEMM's key_b = p | (q & 1) << 2: rows q and q + 2 shared chunk AND parity, a two-way conflict on every weight
// fragment read -- the "27 % LDS-conflict share" of these kernels in the round-4 counters.)
__device__ __forceinline__ int key_cv(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

// Input transform (PRE_BN): the kernel reads the PRE-BatchNorm output z of the previous convolution and applies
// relu((z - mean) * rstd * gamma + beta) -- in the arithmetic of k_bn_relu_apply, rounded to bf16 -- to every chunk on its way from the
// staging registers to LDS; pixels outside the image stay zero (the padding is of the ACTIVATION).  The activation tensor between two
// convolutions of the stem is then never written or read (models/volo.py:355-366): 2 x 411 MB per step and the pass that made them.
// STATS: every workgroup also stores the per-channel sum / sum of squares of its (bf16-rounded) outputs to stats[blockIdx.x][2][64]
// -- the partial sums of the BatchNorm that follows (models/volo.py:356-366), in the layout k_bn_finalize reads.
//
// Per tile the wave runs 18 fenced steps (9 taps x 2 K halves) of 32 MFMAs.  Everything else rides in their shadow: step s
// reads the fragments of step s+1 from LDS, issues the global load of chunk s of the NEXT tile's patch (20 chunks per thread)
// and stores chunk s of the PREVIOUS tile's output (16 chunks per thread, kept packed in registers) -- with one workgroup per
// CU, a load phase, a compute phase and a store phase of their own would run one after the other on every CU at once
// (measured: 151 us with the phases apart, 84 us for the MFMAs alone).
//
// NW = waves per workgroup (4 or 8; AP_CONV_WAVES).  With four waves a SIMD holds ONE wave: every LDS wait, every address computation
// and every barrier of that wave leaves the SIMD's matrix pipe idle.  Eight waves (two per SIMD, 256 registers each -- the budget of
// a wave does not shrink) take tile rows w, w + 8, ...: 4 x 4 accumulator tiles, half the staging and output registers per wave, the
// weight fragments read by twice as many waves (0.5 instead of 0.375 KB of LDS reads per MFMA).
//
// BSTATS (round 5; the input-gradient launch of a stem layer, NW = 8): y is dL/da of the layer BELOW -- a = relu(bn(zprev)) -- and the
// first pass of that BatchNorm's backward (k_bn_relu_bwd_reduce: sum dz, sum dz * xhat per channel, dz = dy where the ReLU passed) needs
// exactly the values this kernel holds in registers when a tile is done.  The tile's zprev chunks are requested behind the next tile's
// patch loads (they land under the remaining MFMA steps), the sums are formed from the bf16-ROUNDED outputs in the arithmetic of
// k_bn_relu_bwd_reduce, and every workgroup stores its partial row to stats[blockIdx.x][2][64] -- the layout k_bn_bwd_finalize reads.
// `bn` then holds the statistics / affine parameters of the layer below (in LDS: 1 KB behind the patch).  Saves one pass over
// dL/da and zprev (2 x 205 MB per layer at B = 128, 224 px).
template <bool STATS, int ABL = 0, bool PRE_BN = false, int NW = 4, bool BSTATS = false>
__global__ void __launch_bounds__(64 * NW)
k_conv3x3_c64(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, bf16_t* __restrict__ y, int H, int W,
              int tiles_x, int tiles_y, int ntiles, float* __restrict__ stats, BnIn bn = BnIn{nullptr, nullptr, nullptr, nullptr},
              const bf16_t* __restrict__ zprev = nullptr) {
    static_assert(!(BSTATS && (STATS || PRE_BN)), "BSTATS is the input-gradient launch: no forward statistics, no input transform");
    constexpr int NTH = 64 * NW, RPW = CV_TR / NW, NPRE = (CV_NCHUNK + NTH - 1) / NTH, PSTEP = NTH / 8;     // threads, tile rows per wave, patch chunks per thread, pixels per staging sweep
    constexpr int NOUT = 2 * RPW, SL = NOUT / 4;                                                        // output chunks per lane; first step of the next tile's loads
    extern __shared__ __attribute__((aligned(16))) bf16_t cv_smem[];
    bf16_t* Wl = cv_smem;                       // [9][64 co][64 ci], chunk ^ key_cv(co)
    bf16_t* P = cv_smem + CV_WELEMS;            // [612 px][64 ci + 8 pad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // per-lane element offsets of the MFMA fragments: pixel (row wave, column fr) chunk g of the patch; weight row r(q, fr) chunk g / 4 + g
    const int abase = (wave * CV_PW + fr) * CV_PSTR + g * 8;
    int wbase[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 32 * (q >> 1) + 8 * (fr >> 2) + 4 * (q & 1) + (fr & 3);       // N-permuted rows: see gemm.hip "direct epilogue"
        wbase[q][0] = r * CV_C + ((g ^ key_cv(r)) << 3);
        wbase[q][1] = r * CV_C + (((4 + g) ^ key_cv(r)) << 3);
    }
    for (int idx = tid; idx < CV_WELEMS / 8; idx += NTH) {
        const int row = idx >> 3, c = idx & 7;
        st16(Wl + row * CV_C + ((c ^ key_cv(row & (CV_C - 1))) << 3), ld16(wp + (int64_t)idx * 8));
    }
    float* const bpar = reinterpret_cast<float*>(P + CV_NPIX * CV_PSTR);     // BSTATS: [mean | rstd | gamma | beta][64] of the layer below
    if constexpr (BSTATS) {
        if (tid < 4 * CV_C) {
            const float* src = tid < CV_C ? bn.mean : tid < 2 * CV_C ? bn.rstd : tid < 3 * CV_C ? bn.gamma : bn.beta;
            bpar[tid] = src[tid & (CV_C - 1)];
        }
    }

    auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        b = t / (tiles_x * tiles_y); ty0 = ty * CV_TR; tx0 = tx * CV_TW;
    };
    // patch chunk i of this thread: pixel (tid >> 3) + 32 i, chunk tid & 7 -- its position in the patch is the same for every
    // tile: one packed (py, px) word per chunk
    u32x4 pre[NPRE];
    const int c8 = (tid & 7) * 8;
    float bsc[8], bsh[8];                        // PRE_BN: scale / shift of this thread's 8 channels (its chunk index is the same for every chunk)
    if constexpr (PRE_BN) bn_in_consts(bn, c8, bsc, bsh);
    int tl = tid >> 3;                           // laundered once per tile: keeps the 20 per-chunk offsets from being hoisted out of the
                                                 // tile loop into 20 registers held for the whole kernel (they are 3 VALU each)
    // The load is UNCONDITIONAL (halo pixels outside the image read a clamped, valid address) and the zero padding is applied
    // when the chunk is written to LDS: a predicated load (zero-initialise, then an exec-masked load into the same registers)
    // made the compiler drain vmcnt(0) in every step.
    auto pvalid = [&](int i, int ty0, int tx0) {
        const int pix = tl + PSTEP * i;                              // < 640
        const int py = (pix * 3641) >> 16, px = pix - py * CV_PW;    // pix / 18 (exact below 1170)
        const unsigned gy = (unsigned)(ty0 - 1 + py), gx = (unsigned)(tx0 - 1 + px);
        return pix < CV_NPIX && gy < (unsigned)H && gx < (unsigned)W;
    };
    auto gload1 = [&](int i, const bf16_t* img, int ty0, int tx0) {
        const int pix = tl + PSTEP * i;
        const int py = (pix * 3641) >> 16, px = pix - py * CV_PW;
        const int gy = min(max(ty0 - 1 + py, 0), H - 1), gx = min(max(tx0 - 1 + px, 0), W - 1);
        pre[i] = ld16(img + (unsigned)((gy * W + gx) * CV_C + c8));  // uniform base + 32-bit lane offset
    };
    auto patch_org = [&](int b, int ty0, int tx0) { return x + (int64_t)b * H * W * CV_C; };
    float ssum[2][8], ssq[2][8];                 // STATS / BSTATS: this lane's channels 32 pr + 8 g .. +7
    u32x4 zq[BSTATS ? NOUT : 1];                 // BSTATS: zprev at this lane's output positions (chunk 2 i + pr)
    float bsum[2] = {0.f, 0.f}, bsq[2] = {0.f, 0.f};        // BSTATS: running sums of the channels 32 pr + 8 g + (fr & 7), over half of this lane's 16-lane group
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) { ssum[pr][k] = 0.f; ssq[pr][k] = 0.f; }

    u32x4 outp[NOUT];                            // the previous tile's output, packed bf16: chunk 2 i + pr
    const bf16_t* out_prev = nullptr;            // UNIFORM: the previous tile's first pixel; null: nothing pending
    int nrow_prev = 0;
    bool col_ok_prev = false;
    const unsigned out_lane = (unsigned)((wave * W + fr) * CV_C + 8 * g);       // this lane's pixel (row wave, column fr), channel 8 g
    auto store1 = [&](int s) {                   // chunk s = 2 i + pr of the pending output
        const int i = s >> 1, pr = s & 1;
        if (out_prev != nullptr && i < nrow_prev && col_ok_prev && (!(ABL & 1) || outp[s][0] == 0x12345678u))
            st16(const_cast<bf16_t*>(out_prev) + ((int64_t)(NW * i) * W * CV_C + 32 * pr) + out_lane, outp[s]);
    };

    int t = blockIdx.x;
    if (t < ntiles) {
        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        const bf16_t* org = patch_org(b, ty0, tx0);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) gload1(i, org, ty0, tx0);
    }
    for (; t < ntiles; t += gridDim.x) {
        int bc, ty0c, tx0c;
        tile_origin(t, bc, ty0c, tx0c);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int idx = tid + NTH * i;
            if (idx < CV_NCHUNK && (!(ABL & 2) || t == (int)blockIdx.x))
                st16(P + ((tid >> 3) + PSTEP * i) * CV_PSTR + c8, pvalid(i, ty0c, tx0c) ? (PRE_BN ? bn_in_apply(pre[i], bsc, bsh) : pre[i]) : zero4);
        }
        __syncthreads();
        asm volatile("" : "+v"(tl));
        const int tn = t + gridDim.x;
        const bool have_next = !(ABL & 2) && tn < ntiles;
        int bn = 0, ty0n = 0, tx0n = 0;
        if (have_next) tile_origin(tn, bn, ty0n, tx0n);
        const bf16_t* orgn = patch_org(bn, ty0n, tx0n);

        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        f32x4 acc[RPW][4];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // rows of this wave: r = wave + NW i; rows below the image are skipped (wave-uniform)
        const int nrow = (H - ty0 - wave + NW - 1) / NW;              // number of i with ty0 + wave + NW i < H
        // fragments of one step: 4 weight fragments (double-buffered by hand) + 8 pixel fragments, each refilled for the next step
        // right behind the 4 MFMAs that read it (one wave per SIMD -- nothing else hides the LDS latency).  Kept as u32x4: bf16x8
        // values crossing the pipeline stages get legalised element-wise.  Every LDS address is a per-lane base + a constant.
        auto bload = [&](u32x4* fb, int s) {
            if ((ABL & 4) && s) return;
#pragma unroll
            for (int q = 0; q < 4; ++q) fb[q] = ld16(Wl + wbase[q][s & 1] + (s >> 1) * CV_C * CV_C);
        };
        auto aload = [&](u32x4& fa, int i, int s) {
            if ((ABL & 4) && s) return;
            const int tap = s >> 1, dy = tap / 3, dx = tap - 3 * dy;
            fa = ld16(P + abase + (NW * i + dy) * CV_PW * CV_PSTR + dx * CV_PSTR + (s & 1) * 32);
        };
        // the memory work of step s.  Program order: the previous tile's 16 output chunks first (steps 0-3), then the 20 chunk
        // loads of the next tile (steps 4-8): vmcnt counts loads and stores in one in-order counter, so the wait for the loads at
        // the top of the next tile also covers everything issued before them -- stores issued AFTER a load would sit in front of
        // that wait -- and the last load still has ten steps (~3 us) to land.  (One load + one store in every step: 200 us.)
        const bf16_t* zimg = BSTATS ? zprev + (int64_t)bc * H * W * CV_C : nullptr;             // UNIFORM
        const int zcol = min(tx0c + fr, W - 1);
        auto zload1 = [&](int c2) {              // rows / columns beyond the image read a clamped, valid address and are not counted
            const int row = min(ty0c + wave + NW * (c2 >> 1), H - 1);
            zq[BSTATS ? c2 : 0] = ld16(zimg + (unsigned)((row * W + zcol) * CV_C + 32 * (c2 & 1) + 8 * g));
        };
        constexpr int SZ = SL + (NPRE + 3) / 4;                    // BSTATS: first step of the zprev loads (behind the patch loads)
        auto side = [&](int s) {
            if (s < SL) {
#pragma unroll
                for (int k = 0; k < 4; ++k) store1(4 * s + k);
            } else if (s < SZ) {
                if (have_next) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (4 * (s - SL) + k < NPRE) gload1(4 * (s - SL) + k, orgn, ty0n, tx0n);
                }
            } else if (BSTATS && s < SZ + NOUT / 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) zload1(4 * (s - SZ) + k);
            }
        };
        auto compute = [&](auto full) {
            u32x4 fb[2][4], fa[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) fa[i] = zero4;
            bload(fb[0], 0);
#pragma unroll
            for (int i = 0; i < RPW; ++i) if (full.value || i < nrow) aload(fa[i], i, 0);
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s + 1 < 18) bload(fb[(s + 1) & 1], s + 1);
                side(s);
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    if (full.value || i < nrow) {
                        if (!(ABL & 8)) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(fb[s & 1][q]), as_bf16x8(fa[i]), acc[i][q], 0, 0, 0);
                        }
                        if (s + 1 < 18) aload(fa[i], i, s + 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);            // keep the hand-made pipeline: without fences the scheduler hoists
            }                                                 // dozens of LDS reads and spills (516 us instead of 170)
        };
        if (nrow >= RPW) compute(std::true_type{}); else compute(std::false_type{});
        // pack: lane (fr, g) holds, per fragment pair pr, channels 32 pr + 8 g .. +7 of pixel (row ty0 + wave + 4 i, column tx0 + fr)
        out_prev = y + (((int64_t)b * H + ty0) * W + tx0) * CV_C;
        nrow_prev = nrow;
        col_ok_prev = tx0 + fr < W;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float v[8];
                v[0] = acc[i][2 * pr][0]; v[1] = acc[i][2 * pr][1]; v[2] = acc[i][2 * pr][2]; v[3] = acc[i][2 * pr][3];
                v[4] = acc[i][2 * pr + 1][0]; v[5] = acc[i][2 * pr + 1][1]; v[6] = acc[i][2 * pr + 1][2]; v[7] = acc[i][2 * pr + 1][3];
                outp[2 * i + pr] = pack8(v);
                if constexpr (STATS) {
                    if (i < nrow && col_ok_prev) {
                        float r8[8];
                        unpack8(outp[2 * i + pr], r8);        // statistics of what is stored (the BatchNorm input is the bf16 tensor)
#pragma unroll
                        for (int k = 0; k < 8; ++k) { ssum[pr][k] += r8[k]; ssq[pr][k] += r8[k] * r8[k]; }
                    }
                }
            }
        }
        if constexpr (BSTATS) {
            // 32 running sums per lane do not fit beside the pipeline's registers: a channel half's sums are temporaries, a reduce-scatter
            // over 8 of the 16 lanes of a group (same g = same channels) leaves each lane ONE channel's pair per half to accumulate
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 mu[2], rs[2], ga[2], be[2];
                const float* bp = bpar + 32 * pr + 8 * g;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    mu[k] = *reinterpret_cast<const f32x4*>(bp + 4 * k);
                    rs[k] = *reinterpret_cast<const f32x4*>(bp + CV_C + 4 * k);
                    ga[k] = *reinterpret_cast<const f32x4*>(bp + 2 * CV_C + 4 * k);
                    be[k] = *reinterpret_cast<const f32x4*>(bp + 3 * CV_C + 4 * k);
                }
                float ts[8], tq[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { ts[k] = 0.f; tq[k] = 0.f; }
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    if (i < nrow && col_ok_prev) {
                        float d8[8], z8[8];
                        unpack8(outp[2 * i + pr], d8);        // what is stored: the BatchNorm backward reads the bf16 tensor
                        unpack8(zq[2 * i + pr], z8);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float xh = (z8[k] - mu[k >> 2][k & 3]) * rs[k >> 2][k & 3];
                            const float dz = (fmaf(xh, ga[k >> 2][k & 3], be[k >> 2][k & 3]) > 0.f) ? d8[k] : 0.f;
                            ts[k] += dz; tq[k] += dz * xh;
                        }
                    }
                }
#pragma unroll
                for (int n = 4; n >= 1; n >>= 1) {   // keep the half of the index range that matches this lane's bit, add the partner's
                    const bool up = fr & n;
#pragma unroll
                    for (int j = 0; j < n; ++j) {
                        const float ks = up ? ts[n + j] : ts[j], gs = up ? ts[j] : ts[n + j];
                        const float kq = up ? tq[n + j] : tq[j], gq = up ? tq[j] : tq[n + j];
                        ts[j] = ks + __shfl_xor(gs, n, 64);
                        tq[j] = kq + __shfl_xor(gq, n, 64);
                    }
                }
                bsum[pr] += ts[0]; bsq[pr] += tq[0];          // channel 32 pr + 8 g + (fr & 7), the lanes fr and fr ^ 8 hold two halves of the group's sum
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                       // every wave is done with the patch before the next one overwrites it
    }
#pragma unroll
    for (int s = 0; s < NOUT; ++s) store1(s);  // the last tile's output
    if constexpr (STATS || BSTATS) {
        // lanes with the same g hold the same channels: butterfly over fr, then the 4 waves meet in LDS (the patch is free)
        float* red = reinterpret_cast<float*>(P);                    // [NW waves][2][64]
        if constexpr (BSTATS) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const float ts = bsum[pr] + __shfl_xor(bsum[pr], 8, 64), tq = bsq[pr] + __shfl_xor(bsq[pr], 8, 64);
                if (fr < 8) {
                    red[(wave * 2 + 0) * CV_C + 32 * pr + 8 * g + fr] = ts;
                    red[(wave * 2 + 1) * CV_C + 32 * pr + 8 * g + fr] = tq;
                }
            }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (BSTATS) break;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { ssum[pr][k] += __shfl_xor(ssum[pr][k], o, 64); ssq[pr][k] += __shfl_xor(ssq[pr][k], o, 64); }
            }
        if (fr == 0 && !BSTATS) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    red[(wave * 2 + 0) * CV_C + 32 * pr + 8 * g + k] = ssum[pr][k];
                    red[(wave * 2 + 1) * CV_C + 32 * pr + 8 * g + k] = ssq[pr][k];
                }
        }
        __syncthreads();
        if (tid < 2 * CV_C) {                  // this workgroup's partial row [2][64]: plain store, summed (in fp64) by the BatchNorm finalize
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w * 2 * CV_C + tid];
            stats[(int64_t)blockIdx.x * 2 * CV_C + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][ci][ky][kx] = sum over pixels of dy[p][co] * x[p + (ky-1, kx-1)][ci]  (zero padding).  Per tap a 64 x 64 matrix product
// whose reduction axis is the PIXEL axis -- the slow axis of both NHWC operands -- so the MFMA fragments are fetched with
// ds_read_b64_tr_b16 (hardware transpose) from pixel-major LDS images, as in k_gemm_tn.
//   workgroup : 256 threads, persistent over 16 x 16 pixel tiles; LDS = dy tile (32 KB) + 18 x 18 input patch (40.5 KB): two
//               workgroups per CU, whose load and compute phases overlap each other
//   wave w    : input channels 16 w .. 16 w + 15, ALL 64 output channels, ALL 9 taps: 9 x 4 accumulator tiles (144 registers).
//               Per 32-pixel K step: 4 dy^T fragments (shared by the taps) + 9 shifted input fragments for 36 MFMAs.
//   swizzle   : 16-byte chunk ^ key(column) -- a function of the pixel COLUMN only, so that a tap (dy: a row shift, dx: three
//               columns per lane) and a K step (two rows) are constant byte offsets from six per-lane bases
//   output    : every workgroup STORES its partial 9 x 64 x 64 sums to its own slab; k_conv3x3_wgrad_reduce adds the slabs in
//               order into dW (fp32 OIHW, +=): no atomics, bit-reproducible
#define CW_T 16
#ifndef CW_PTR
#define CW_PTR 16
#endif
#define CW_PW (CW_T + 2)
#define CW_DPIX (CW_T * CW_T)
#define CW_APIX (CW_PW * CW_PW)
#define CW_LDS_BYTES ((CW_DPIX + CW_APIX) * CV_C * 2)
#define CW_ND (CW_DPIX * 8 / 256)                     // dy chunks per thread (8)
#define CW_NA ((CW_APIX * 8 + 255) / 256)             // patch chunks per thread (11)
__device__ __forceinline__ int cw_key(int col) { return (((col >> 1) & 1) << 1) | (((col >> 3) & 1) << 2); }

__global__ void __launch_bounds__(256, 2)
k_conv3x3_c64_wgrad(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ slab, int H, int W,
                    int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) bf16_t cw_smem[];
    bf16_t* D = cw_smem;                        // [256 px][64 co]
    bf16_t* A = cw_smem + CW_DPIX * CV_C;       // [324 px][64 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4, q = fr >> 2, p = fr & 3;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int c8 = (tid & 7) * 8, cidx = tid & 7;

    // per-lane fragment bases (elements).  Lane (g, q, p) addresses 4 channels (8 B) of pixel row g >> 1, column 8 (g & 1) + q (+4 for
    // the second read of a fragment) of the current two-row K step; the transpose hands lane i = 4 q + p channel i of the tile.
    int dbase[4][2], abase[3][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int col = 8 * (g & 1) + q + 4 * hf;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            dbase[t][hf] = ((g >> 1) * CW_T + col) * CV_C + (((2 * t + (p >> 1)) ^ cw_key(col)) << 3) + (p & 1) * 4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int pc = col + dx;
            abase[dx][hf] = ((g >> 1) * CW_PW + pc) * CV_C + (((2 * wave + (p >> 1)) ^ cw_key(pc)) << 3) + (p & 1) * 4;
        }
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto trfrag = [&](const bf16_t* base0, const bf16_t* base1) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base1));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int ty0 = ty * CW_T, tx0 = tx * CW_T;
        const bf16_t* ximg = x + (int64_t)b * H * W * CV_C;
        const bf16_t* dimg = dy + (int64_t)b * H * W * CV_C;
        // two batches (dy tile, then input patch) of unconditional loads at clamped addresses; pixels outside the image become
        // zeros on the way into LDS.  (One batch of all 19 chunks on top of the 144 accumulators spills.)
        {
            u32x4 rd[CW_ND];
#pragma unroll
            for (int i = 0; i < CW_ND; ++i) {
                const int px = (tid >> 3) + 32 * i, r = px >> 4, c = px & 15;
                const int gy = min(ty0 + r, H - 1), gx = min(tx0 + c, W - 1);
                rd[i] = ld16(dimg + (unsigned)((gy * W + gx) * CV_C + c8));
            }
#pragma unroll
            for (int i = 0; i < CW_ND; ++i) {
                const int px = (tid >> 3) + 32 * i, r = px >> 4, c = px & 15;
                const bool ok = (ty0 + r < H) && (tx0 + c < W);
                st16(D + px * CV_C + ((cidx ^ cw_key(c)) << 3), ok ? rd[i] : zero4);
            }
        }
        {
            u32x4 ra[CW_NA];
#pragma unroll
            for (int i = 0; i < CW_NA; ++i) {
                const int px = min((tid >> 3) + 32 * i, CW_APIX - 1), r = px / CW_PW, c = px - r * CW_PW;
                const int gy = min(max(ty0 - 1 + r, 0), H - 1), gx = min(max(tx0 - 1 + c, 0), W - 1);
                ra[i] = ld16(ximg + (unsigned)((gy * W + gx) * CV_C + c8));
            }
#pragma unroll
            for (int i = 0; i < CW_NA; ++i) {
                const int px = (tid >> 3) + 32 * i, r = px / CW_PW, c = px - r * CW_PW;
                const unsigned gy = (unsigned)(ty0 - 1 + r), gx = (unsigned)(tx0 - 1 + c);
                if (px < CW_APIX) st16(A + px * CV_C + ((cidx ^ cw_key(c)) << 3), (gy < (unsigned)H && gx < (unsigned)W) ? ra[i] : zero4);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CW_T / 2; ++k) {                 // K step: tile rows 2k, 2k+1 (32 pixels)
            bf16x8 df[4];
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) df[tq] = trfrag(D + dbase[tq][0] + 2 * k * CW_T * CV_C, D + dbase[tq][1] + 2 * k * CW_T * CV_C);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dyy = tap / 3, dxx = tap - 3 * dyy;
                const bf16x8 af = trfrag(A + abase[dxx][0] + (2 * k + dyy) * CW_PW * CV_C, A + abase[dxx][1] + (2 * k + dyy) * CW_PW * CV_C);
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) acc[tap][tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[tq], af, acc[tap][tq], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // partial sums: slab[wg][tap][co][ci]; lane holds co = 16 tq + 4 g + r, ci = 16 wave + fr
    float* mine = slab + (int64_t)blockIdx.x * CV_WELEMS;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mine[(tap * CV_C + 16 * tq + 4 * g + r) * CV_C + 16 * wave + fr] = acc[tap][tq][r];
}

// The same reduction with ONE workgroup per CU and the whole register file behind it (the default; AP_CONV_WGRAD_P=0 selects the kernel
// above): the next tile's dy tile and input patch are loaded into registers (19 x 16 B per thread) while the current one is computed,
// and the fragments of K step k+1 are read from LDS while the MFMAs of step k issue.  The two-workgroup kernel cannot afford either
// next to its 144 accumulators and spends 74 % of its wave time waiting.  208 -> 151 us at B = 128, 112 x 112 (16-row tiles; 32-row
// tiles spill and take 177); half the slabs, too.
// NW = 8 (AP_CONV_WAVES, the default): two waves per SIMD.  Wave (cg, ch) = (wave & 3, wave >> 2) takes input channels 16 cg .. +15 and
// output channels 32 ch .. +31: 9 x 2 accumulator tiles, its 2 dy^T fragments + the 9 shifted input fragments per K step for 18 MFMAs.
template <int TR, bool PRE_BN = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW)
k_conv3x3_c64_wgrad_p(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ slab, int H, int W,
                      int tiles_x, int tiles_y, int ntiles, BnIn bn = BnIn{nullptr, nullptr, nullptr, nullptr}, int xs = CV_C, int dys = CV_C) {
    // xs / dys: pixel strides (elements) of x and dy -- 64 for the 64-channel tensors; 128 when the operands are 64-channel halves of
    // 128-channel feature maps (ap_conv3x3_c128_wgrad: the pointers then carry the channel offset)
    constexpr int NTH = 64 * NW, PSTEP = NTH / 8, COT = NW == 8 ? 2 : 4;      // threads, pixels per staging sweep, 16-channel output tiles per wave
    constexpr int DPIX = TR * CW_T, APIX = (TR + 2) * CW_PW;
    constexpr int ND = DPIX * 8 / NTH, NA = (APIX * 8 + NTH - 1) / NTH;
    static_assert(DPIX * 8 % NTH == 0, "dy tile chunks per thread");
    extern __shared__ __attribute__((aligned(16))) bf16_t cw_smem[];
    bf16_t* D = cw_smem;                        // [TR*16 px][64 co]
    bf16_t* A = cw_smem + DPIX * CV_C;          // [(TR+2)*18 px][64 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4, q = fr >> 2, p = fr & 3;
    const int cg = wave & 3, ch = NW == 8 ? (wave >> 2) : 0;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int c8 = (tid & 7) * 8, cidx = tid & 7;
    float bsc[8], bsh[8];                       // PRE_BN: x is the pre-BatchNorm tensor, the operand is relu(bn(x)) (see BnIn)
    if constexpr (PRE_BN) bn_in_consts(bn, c8, bsc, bsh);
    int dbase[COT][2], abase[3][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int col = 8 * (g & 1) + q + 4 * hf;
#pragma unroll
        for (int t = 0; t < COT; ++t)
            dbase[t][hf] = ((g >> 1) * CW_T + col) * CV_C + (((2 * (COT * ch + t) + (p >> 1)) ^ cw_key(col)) << 3) + (p & 1) * 4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int pc = col + dx;
            abase[dx][hf] = ((g >> 1) * CW_PW + pc) * CV_C + (((2 * cg + (p >> 1)) ^ cw_key(pc)) << 3) + (p & 1) * 4;
        }
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto trfrag = [&](const bf16_t* base0, const bf16_t* base1) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base1));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f32x4 acc[9][COT];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int b = 0; b < COT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 rd[ND], ra[NA];
    auto origin = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        b = t / (tiles_x * tiles_y); ty0 = ty * TR; tx0 = tx * CW_T;
    };
    int tl = tid >> 3;                               // laundered per tile (see k_conv3x3_c64): keeps per-chunk offsets out of long-lived registers
    auto gload = [&](int t) {
        int b, ty0, tx0;
        origin(t, b, ty0, tx0);
        const bf16_t* ximg = x + (int64_t)b * H * W * xs;
        const bf16_t* dimg = dy + (int64_t)b * H * W * dys;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int px = tl + PSTEP * i, r = px >> 4, c = px & 15;
            const int gy = min(ty0 + r, H - 1), gx = min(tx0 + c, W - 1);
            rd[i] = ld16(dimg + (unsigned)((gy * W + gx) * dys + c8));
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int px = min(tl + PSTEP * i, APIX - 1), r = (px * 3641) >> 16, c = px - r * CW_PW;     // px / 18 (exact below 1170)
            const int gy = min(max(ty0 - 1 + r, 0), H - 1), gx = min(max(tx0 - 1 + c, 0), W - 1);
            ra[i] = ld16(ximg + (unsigned)((gy * W + gx) * xs + c8));
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) gload(t);
    for (; t < ntiles; t += gridDim.x) {
        int b, ty0, tx0;
        origin(t, b, ty0, tx0);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int px = (tid >> 3) + PSTEP * i, r = px >> 4, c = px & 15;
            st16(D + px * CV_C + ((cidx ^ cw_key(c)) << 3), ((ty0 + r < H) && (tx0 + c < W)) ? rd[i] : zero4);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int px = (tid >> 3) + PSTEP * i, r = (px * 3641) >> 16, c = px - r * CW_PW;
            const unsigned gy = (unsigned)(ty0 - 1 + r), gx = (unsigned)(tx0 - 1 + c);
            if (px < APIX) st16(A + px * CV_C + ((cidx ^ cw_key(c)) << 3), (gy < (unsigned)H && gx < (unsigned)W) ? (PRE_BN ? bn_in_apply(ra[i], bsc, bsh) : ra[i]) : zero4);
        }
        __syncthreads();
        asm volatile("" : "+v"(tl));
        if (t + (int)gridDim.x < ntiles) gload(t + gridDim.x);          // in flight during the MFMA steps below
        // K steps (tile rows 2k, 2k+1 = 32 pixels), fully unrolled with the fragments of step k+1 read from LDS while the MFMAs of
        // step k issue; fences keep the two-stage pipeline as written
        bf16x8 df[2][COT], af[2][9];
        auto fload = [&](int k, bf16x8* dfr, bf16x8* afr) {
#pragma unroll
            for (int tq = 0; tq < COT; ++tq) dfr[tq] = trfrag(D + dbase[tq][0] + 2 * k * CW_T * CV_C, D + dbase[tq][1] + 2 * k * CW_T * CV_C);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dyy = tap / 3, dxx = tap - 3 * dyy;
                afr[tap] = trfrag(A + abase[dxx][0] + (2 * k + dyy) * CW_PW * CV_C, A + abase[dxx][1] + (2 * k + dyy) * CW_PW * CV_C);
            }
        };
        fload(0, df[0], af[0]);
#pragma unroll
        for (int k = 0; k < TR / 2; ++k) {
            if (k + 1 < TR / 2) fload(k + 1, df[(k + 1) & 1], af[(k + 1) & 1]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int tq = 0; tq < COT; ++tq) acc[tap][tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[k & 1][tq], af[k & 1][tap], acc[tap][tq], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float* mine = slab + (int64_t)blockIdx.x * CV_WELEMS;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tq = 0; tq < COT; ++tq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mine[(tap * CV_C + 16 * (COT * ch + tq) + 4 * g + r) * CV_C + 16 * cg + fr] = acc[tap][tq][r];
}

// dW[co][ci][tap] (fp32 OIHW) += sum over the workgroup slabs, in a fixed order.  Block = 256 consecutive slab elements x 16 waves;
// wave v adds slabs v, v + 16, ... (eight 1-KB loads in flight per wave: the first version, one thread per element walking all slabs
// with four loads in flight, ran at 1.9 TB/s), the 16 partial sums meet in LDS and are added in wave order.
__global__ void __launch_bounds__(1024)
k_conv3x3_wgrad_reduce(const float* __restrict__ slab, int nslab, float* __restrict__ dw, int cin = CV_C) {
    __shared__ float4 part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = blockIdx.x * 256 + lane * 4;              // CV_WELEMS = 144 * 256
    float4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int w = wave;
    for (; w + 7 * 16 < nslab; w += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(slab + (int64_t)(w + 16 * u) * CV_WELEMS + e0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s[u & 1].x += v[u].x; s[u & 1].y += v[u].y; s[u & 1].z += v[u].z; s[u & 1].w += v[u].w; }
    }
    for (; w < nslab; w += 16) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (int64_t)w * CV_WELEMS + e0);
        s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
    part[wave][lane] = make_float4(s[0].x + s[1].x, s[0].y + s[1].y, s[0].z + s[1].z, s[0].w + s[1].w);
    __syncthreads();
    if (threadIdx.x < 256) {
        const int idx = blockIdx.x * 256 + threadIdx.x;      // slab element (tap, co, ci)
        const float* pf = reinterpret_cast<const float*>(&part[0][0]);
        float t = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) t += pf[v * 256 + threadIdx.x];
        const int ci = idx % CV_C, co = (idx / CV_C) % CV_C, tap = idx / (CV_C * CV_C);
        dw[(co * cin + ci) * 9 + tap] += t;           // cin: input channels of the WHOLE weight tensor (64; 128 for a quadrant of a 128 x 128 layer)
    }
}

extern "C" {

int ap_conv3x3_c64_pack(const float* w_oihw, ap_bf16* w_fwd, ap_bf16* w_bwd, ap_stream_t stream) {
    if (!w_oihw || !w_fwd || !w_bwd) return AP_ERR_NULL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_conv3x3_pack, dim3((CV_WELEMS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, w_fwd, w_bwd);
    return ap_check_launch();
}

static int cv_grid(int ntiles) {
    static int grid_cap = 0;
    if (grid_cap == 0) { const char* e = getenv("AP_CONV_GRID"); grid_cap = e ? atoi(e) : 256; if (grid_cap < 1) grid_cap = 256; }
    return ntiles < grid_cap ? ntiles : grid_cap;
}

int ap_conv3x3_c64_stat_rows(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t nt = (int64_t)B * ((W + CV_TW - 1) / CV_TW) * ((H + CV_TR - 1) / CV_TR);
    return cv_grid((int)(nt > 0x7fffffff ? 0x7fffffff : nt));
}

int ap_conv3x3_c64(const ap_bf16* x, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats, ap_stream_t stream) {
    return ap_conv3x3_c64_bn(x, nullptr, w_packed, y, B, H, W, stats, stream);
}

int ap_conv3x3_c64_bn(const ap_bf16* x, const ap_bn_input* bn_in, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats,
                      ap_stream_t stream) {
    if (!x || !w_packed || !y) return AP_ERR_NULL;
    if (bn_in && (!bn_in->mean || !bn_in->rstd || !bn_in->gamma || !bn_in->beta)) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    const int tiles_x = (W + CV_TW - 1) / CV_TW, tiles_y = (H + CV_TR - 1) / CV_TR;
    const int64_t nt64 = (int64_t)B * tiles_x * tiles_y;
    if (nt64 > 0x7fffffff) return AP_ERR_SHAPE;
    const int ntiles = (int)nt64;
    static int attr_done = 0;
    (void)hipGetLastError();
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        attr_done = 1;
    }
    const int grid = cv_grid(ntiles);
    static int waves = 0;
    if (waves == 0) { const char* e = getenv("AP_CONV_WAVES"); waves = (e && atoi(e) == 4) ? 4 : 8; }
    if (waves == 8) {
        static int attr8 = 0;
        if (!attr8) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<false, 0, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<true, 0, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<false, 0, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<true, 0, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
            attr8 = 1;
        }
        const BnIn bn = bn_in ? BnIn{bn_in->mean, bn_in->rstd, bn_in->gamma, bn_in->beta} : BnIn{nullptr, nullptr, nullptr, nullptr};
        if (bn_in && stats) hipLaunchKernelGGL((k_conv3x3_c64<true, 0, true, 8>), dim3(grid), dim3(512), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats, bn);
        else if (bn_in) hipLaunchKernelGGL((k_conv3x3_c64<false, 0, true, 8>), dim3(grid), dim3(512), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats, bn);
        else if (stats) hipLaunchKernelGGL((k_conv3x3_c64<true, 0, false, 8>), dim3(grid), dim3(512), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats, bn);
        else hipLaunchKernelGGL((k_conv3x3_c64<false, 0, false, 8>), dim3(grid), dim3(512), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats, bn);
        return ap_check_launch();
    }
    if (bn_in) {
        const BnIn bn = {bn_in->mean, bn_in->rstd, bn_in->gamma, bn_in->beta};
        if (stats) hipLaunchKernelGGL((k_conv3x3_c64<true, 0, true>), dim3(grid), dim3(256), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats, bn);
        else hipLaunchKernelGGL((k_conv3x3_c64<false, 0, true>), dim3(grid), dim3(256), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats, bn);
        return ap_check_launch();
    }
#if AP_EXPERIMENTS             // ablation instantiations (DESIGN.md "Convolution kernels, by ablation"): make EXTRA=-DAP_EXPERIMENTS=1, AP_CONV_ABL=bits
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("AP_CONV_ABL"); abl = e ? atoi(e) : 0; }
#define CV_ABL_LAUNCH(A) case A: hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<false, A>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES); \
        hipLaunchKernelGGL((k_conv3x3_c64<false, A>), dim3(grid), dim3(256), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats); return ap_check_launch();
    switch (abl) { CV_ABL_LAUNCH(1) CV_ABL_LAUNCH(2) CV_ABL_LAUNCH(3) CV_ABL_LAUNCH(4) CV_ABL_LAUNCH(7) CV_ABL_LAUNCH(8) CV_ABL_LAUNCH(11) default: break; }
#endif
    if (stats) hipLaunchKernelGGL((k_conv3x3_c64<true>), dim3(grid), dim3(256), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats);
    else hipLaunchKernelGGL((k_conv3x3_c64<false>), dim3(grid), dim3(256), CV_LDS_BYTES, (hipStream_t)stream, x, w_packed, y, H, W, tiles_x, tiles_y, ntiles, stats);
    return ap_check_launch();
}

int ap_conv3x3_c64_bwd_stats(const ap_bf16* dz, const ap_bf16* w_packed_bwd, ap_bf16* da, int B, int H, int W, const ap_bf16* z_below,
                             const ap_bn_input* bn_below, float* stats, ap_stream_t stream) {
    if (!dz || !w_packed_bwd || !da || !z_below || !bn_below || !stats) return AP_ERR_NULL;
    if (!bn_below->mean || !bn_below->rstd || !bn_below->gamma || !bn_below->beta) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    const int tiles_x = (W + CV_TW - 1) / CV_TW, tiles_y = (H + CV_TR - 1) / CV_TR;
    const int64_t nt64 = (int64_t)B * tiles_x * tiles_y;
    if (nt64 > 0x7fffffff || (int64_t)H * W * CV_C > 0x7fffffff) return AP_ERR_SHAPE;
    const int ntiles = (int)nt64;
    static int attr_done = 0;
    (void)hipGetLastError();
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<false, 0, false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES + 1024) != hipSuccess) return AP_ERR_LAUNCH;
        attr_done = 1;
    }
    const BnIn bn = {bn_below->mean, bn_below->rstd, bn_below->gamma, bn_below->beta};
    hipLaunchKernelGGL((k_conv3x3_c64<false, 0, false, 8, true>), dim3(cv_grid(ntiles)), dim3(512), CV_LDS_BYTES + 1024, (hipStream_t)stream, dz, w_packed_bwd,
                       da, H, W, tiles_x, tiles_y, ntiles, stats, bn, z_below);
    return ap_check_launch();
}

static int cw_grid(int ntiles) {
    static int cap = 0;
    if (cap == 0) { const char* e = getenv("AP_CONV_WGRAD_GRID"); cap = e ? atoi(e) : 512; if (cap < 1) cap = 512; }
    return ntiles < cap ? ntiles : cap;
}

size_t ap_conv3x3_c64_wgrad_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t nt = (int64_t)B * ((W + CW_T - 1) / CW_T) * ((H + CW_T - 1) / CW_T);
    return (size_t)cw_grid((int)(nt > 0x7fffffff ? 0x7fffffff : nt)) * CV_WELEMS * sizeof(float);
}

int ap_conv3x3_c64_wgrad(const ap_bf16* x, const ap_bf16* dy, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                         ap_stream_t stream) {
    return ap_conv3x3_c64_wgrad_bn(x, nullptr, dy, dw_oihw, B, H, W, workspace, ws_bytes, stream);
}

int ap_conv3x3_c64_wgrad_bn(const ap_bf16* x, const ap_bn_input* bn_in, const ap_bf16* dy, float* dw_oihw, int B, int H, int W, void* workspace,
                            size_t ws_bytes, ap_stream_t stream) {
    if (!x || !dy || !dw_oihw || !workspace) return AP_ERR_NULL;
    if (bn_in && (!bn_in->mean || !bn_in->rstd || !bn_in->gamma || !bn_in->beta)) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    const int tiles_x = (W + CW_T - 1) / CW_T, tiles_y = (H + CW_T - 1) / CW_T;
    const int64_t nt64 = (int64_t)B * tiles_x * tiles_y;
    if (nt64 > 0x7fffffff) return AP_ERR_SHAPE;
    if (ws_bytes < ap_conv3x3_c64_wgrad_workspace(B, H, W)) return AP_ERR_SHAPE;
    int ntiles = (int)nt64, grid = cw_grid(ntiles);
    static int attr_done = 0;
    (void)hipGetLastError();
    constexpr int PTR = CW_PTR;
    constexpr int P_LDS = (PTR * CW_T + (PTR + 2) * CW_PW) * CV_C * 2;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS_BYTES) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64_wgrad_p<PTR>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return AP_ERR_LAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64_wgrad_p<PTR, true>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return AP_ERR_LAUNCH;
        attr_done = 1;
    }
    static int use_p = -1;
    if (use_p < 0) { const char* e = getenv("AP_CONV_WGRAD_P"); use_p = e ? atoi(e) : 1; }
    if (use_p || bn_in) {       // one prefetching workgroup per CU, 32 x 16 tiles (the only kernel with the BatchNorm input transform)
        const int tyy = (H + PTR - 1) / PTR;
        const int64_t np = (int64_t)B * tiles_x * tyy;
        ntiles = (int)np;
        const int gp = ntiles < 256 ? ntiles : 256;
        if ((size_t)gp * CV_WELEMS * sizeof(float) > ws_bytes) return AP_ERR_SHAPE;
        grid = gp;
        static int waves = 0;
        if (waves == 0) { const char* e = getenv("AP_CONV_WAVES"); waves = (e && atoi(e) == 4) ? 4 : 8; }
        const BnIn bn = bn_in ? BnIn{bn_in->mean, bn_in->rstd, bn_in->gamma, bn_in->beta} : BnIn{nullptr, nullptr, nullptr, nullptr};
        if (waves == 8) {
            static int attr8 = 0;
            if (!attr8) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64_wgrad_p<PTR, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return AP_ERR_LAUNCH;
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64_wgrad_p<PTR, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return AP_ERR_LAUNCH;
                attr8 = 1;
            }
            if (bn_in) hipLaunchKernelGGL((k_conv3x3_c64_wgrad_p<PTR, true, 8>), dim3(grid), dim3(512), P_LDS, (hipStream_t)stream, x, dy, static_cast<float*>(workspace), H, W, tiles_x, tyy, ntiles, bn);
            else hipLaunchKernelGGL((k_conv3x3_c64_wgrad_p<PTR, false, 8>), dim3(grid), dim3(512), P_LDS, (hipStream_t)stream, x, dy, static_cast<float*>(workspace), H, W, tiles_x, tyy, ntiles, bn);
        } else if (bn_in) {
            hipLaunchKernelGGL((k_conv3x3_c64_wgrad_p<PTR, true>), dim3(grid), dim3(256), P_LDS, (hipStream_t)stream, x, dy, static_cast<float*>(workspace), H, W, tiles_x, tyy, ntiles, bn);
        } else
        hipLaunchKernelGGL((k_conv3x3_c64_wgrad_p<PTR>), dim3(grid), dim3(256), P_LDS, (hipStream_t)stream, x, dy, static_cast<float*>(workspace), H, W, tiles_x, tyy, ntiles);
    } else
    hipLaunchKernelGGL(k_conv3x3_c64_wgrad, dim3(grid), dim3(256), CW_LDS_BYTES, (hipStream_t)stream, x, dy, static_cast<float*>(workspace), H, W, tiles_x, tiles_y, ntiles);
    int rc = ap_check_launch();
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(k_conv3x3_wgrad_reduce, dim3(CV_WELEMS / 256), dim3(1024), 0, (hipStream_t)stream, static_cast<const float*>(workspace), grid, dw_oihw);
    return ap_check_launch();
}

// ---- weight gradient of the 128 -> 128 layer (VOLO-D4 / D5 stem, csrc/conv128.hip): dW[co][ci] splits into four 64 x 64 quadrants
// (output half a, input half b), each exactly the 64-channel problem on the channel halves of x and dy -- the 64-channel kernel with a
// pixel stride of 128 elements and the channel offset in the pointers; its slab reduction writes the quadrant into the [128][128][3][3]
// tensor.  Every operand half is read twice; nothing else is new.
size_t ap_conv3x3_c128_wgrad_workspace(int B, int H, int W) { return ap_conv3x3_c64_wgrad_workspace(B, H, W); }

int ap_conv3x3_c128_wgrad(const ap_bf16* x, const ap_bf16* dy, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                          ap_stream_t stream) {
    if (!x || !dy || !dw_oihw || !workspace) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    if (ws_bytes < ap_conv3x3_c128_wgrad_workspace(B, H, W) || (int64_t)H * W * 128 > 0x7fffffff) return AP_ERR_SHAPE;
    constexpr int PTR = CW_PTR;
    constexpr int P_LDS = (PTR * CW_T + (PTR + 2) * CW_PW) * CV_C * 2;
    const int tiles_x = (W + CW_T - 1) / CW_T, tyy = (H + PTR - 1) / PTR;
    const int64_t np = (int64_t)B * tiles_x * tyy;
    if (np > 0x7fffffff) return AP_ERR_SHAPE;
    const int ntiles = (int)np, grid = ntiles < 256 ? ntiles : 256;
    if ((size_t)grid * CV_WELEMS * sizeof(float) > ws_bytes) return AP_ERR_SHAPE;
    static int attr_done = 0;
    (void)hipGetLastError();
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64_wgrad_p<PTR, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return AP_ERR_LAUNCH;
        attr_done = 1;
    }
    const BnIn bn = BnIn{nullptr, nullptr, nullptr, nullptr};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            hipLaunchKernelGGL((k_conv3x3_c64_wgrad_p<PTR, false, 8>), dim3(grid), dim3(512), P_LDS, (hipStream_t)stream, x + 64 * b, dy + 64 * a,
                               static_cast<float*>(workspace), H, W, tiles_x, tyy, ntiles, bn, 128, 128);
            hipLaunchKernelGGL(k_conv3x3_wgrad_reduce, dim3(CV_WELEMS / 256), dim3(1024), 0, (hipStream_t)stream, static_cast<const float*>(workspace), grid,
                               dw_oihw + ((int64_t)(64 * a) * 128 + 64 * b) * 9, 128);
        }
    return ap_check_launch();
}

}  // extern "C"
