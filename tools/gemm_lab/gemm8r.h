// Row-complete NT GEMM for N = 384 with LayerNorm inside the epilogue (gfx950).
//
//   LNF (forward,  models/volo.py:230-234):  x1 = (A . W^T + bias) * rs + res ;  xn = LayerNorm(x1; gamma, beta)   -> x1, xn, mean, rstd
//   LNB (backward, autograd of the same):    dxn = A . W^T ;  dx = LayerNorm_backward(dxn; x, mean, rstd, gamma) + dres  -> dx, dgamma / dbeta partials
//
// A LayerNorm needs whole rows.  The 256 x 192 tiles of gemm8p.h cut a 384-wide row in two, so the transformer block ran
// GEMM -> (x1 | dxn to HBM) -> k_ln_fwd / k_ln_bwd (read it back, 11 / 19 us per launch, 42 launches per step).  Here a
// workgroup owns a 128 x 384 tile -- every row complete -- and the normalisation runs in the row phase of the staged epilogue:
// the intermediate never leaves the CU.  Same machinery as gemm8p.h: one 512-thread workgroup per CU, LDS-DMA in whole lines,
// counted vmcnt, raw barriers, two wave groups a barrier apart.
//
// Geometry.  Wave (wr, wc) = (wave >> 2, wave & 3): rows wr*64 .. +63 (4 MFMA tiles), columns wc*96 .. +95 = three PAIRS of
// 16-column tiles (a lane ends with 8 consecutive columns per pair, as in gemm8p.h).  K-tile buffer (64 KB): A [128 x 64] at 0,
// B part j (j = 0, 1, 2: for every wc the 32 columns wc*96 + j*32 ..) at 16384 * (1 + j); three phases of 16 MFMAs per K-tile,
// one B part each, the A fragments stay in registers for the K-tile.  Restaging: B1, B2 of K-tile T+1 in phases 1, 2; B0, A of
// K-tile T+2 in phases 2, 3; EVERY phase ends with a counted wait for what the next phase reads (four parts stay in flight).
//
// Epilogue.  The whole tile (+ bias, row scale, residual in fp32) -> bf16 -> the K-tile buffers (128 rows x 768 B, chunks XORed
// with row & 7) behind one barrier; row phase: 16 lanes own a row (3 chunks of 8 columns each), four rows of a wave in flight, four
// steps; row statistics by shuffles inside the 16 lanes; the outputs leave as whole lines.
#pragma once
#include "../../autoprog_amd/csrc/common.h"
#include "../../autoprog_amd/csrc/gemm_epi.h"
#include "../../autoprog_amd/csrc/gemm8p.h"

#ifndef G8R_ABL
#define G8R_ABL 0           // lab ablations: 1 no MFMA, 2 no fragment reads, 4 no DMA
#endif
#define G8R_LDS_BYTES 163840

struct G8RArgs {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;           // [384, K]
    bf16_t* C; int ldc;                 // LNF: x1; LNB: dx
    int M, K, ntiles;
    // LayerNorm side
    const float* gamma; const float* beta; float eps;
    bf16_t* xn; int ldxn;               // LNF out
    float* mean; float* rstd;           // LNF out / LNB in
    const bf16_t* x; int ldx;           // LNB: the LayerNorm's input
    const bf16_t* dres; int lddres;     // LNB: gradient arriving over the residual path (nullable)
    float* partial;                     // LNB: [ntiles][768] = per-tile column sums (dgamma | dbeta)
};

// MODE 1: LNF, 2: LNB
template <int MODE>
__global__ void __launch_bounds__(512, 2) k_gemm_nt_8r(G8RArgs ga, EpiArgs ep) {
    constexpr int NC = 384, KS = 65536, STG = 98304, RS = NC * 2, GRP = 16 * RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char g8r_smem[];
    unsigned char* const smem = g8r_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, g = lane >> 4;
    const int nk = ga.K >> 6;

    // ---- DMA source geometry: a wave-instruction fills one 1 KB piece = 8 image rows x 64 k (whole 128-byte lines); wave w fills
    // pieces 2w, 2w+1 (image rows 16w .. 16w+15) of every part
    const int sr8 = lane >> 3;
    const int scol = ((lane & 7) ^ sr8) * 8;
    const int irow = wave * 16 + sr8;                                       // image row of the first piece (+8: second)
    auto bcol_of = [&](int r, int j) { const int i = r & 15; return (r >> 5) * 96 + j * 32 + (i >> 2) * 8 + ((r >> 4) & 1) * 4 + (i & 3); };
    const bf16_t* qb[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) qb[j][h] = ga.B + (int64_t)bcol_of(irow + 8 * h, j) * ga.ldb + scol;
    unsigned char* const dma_dst = smem + wave * 2048;                       // + buffer + part * 16384 (+ 1024: second piece)
    // ---- fragment read addresses
    const int lane_row = (fr >> 3) * 1024 + (fr & 7) * 128;
    const int lane_off0 = lane_row + ((g ^ (fr & 7)) << 4);
    const int lane_off1 = lane_row + (((4 + g) ^ (fr & 7)) << 4);
    const unsigned char* const rdA = smem + wr * 8192;                       // + bo + mt*2048 + lane_off{kb}
    const unsigned char* const rdB = smem + 16384 + wc * 4096;               // + bo + j*16384 + nt*2048 + lane_off{kb}

#define G8R_BAR() __builtin_amdgcn_s_barrier()
#define G8R_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define G8R_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#define G8R_FENCE() __builtin_amdgcn_sched_barrier(0)

    for (int tile = blockIdx.x; tile < ga.ntiles; tile += gridDim.x) {
        const int m0 = tile * 128;
        const bf16_t* qa[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) qa[h] = ga.A + (int64_t)min(m0 + irow + 8 * h, ga.M - 1) * ga.lda + scol;
        auto dmaA = [&](int bo, int kt) {
            if (G8R_ABL & 4) return;
            __builtin_amdgcn_global_load_lds(G8_GLB(qa[0] + kt * 64), G8_LDS(dma_dst + bo), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(G8_GLB(qa[1] + kt * 64), G8_LDS(dma_dst + bo + 1024), 16, 0, 0);
        };
        auto dmaB = [&](int j, int bo, int kt) {
            if (G8R_ABL & 4) return;
            __builtin_amdgcn_global_load_lds(G8_GLB(qb[j][0] + kt * 64), G8_LDS(dma_dst + bo + 16384 * (1 + j)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(G8_GLB(qb[j][1] + kt * 64), G8_LDS(dma_dst + bo + 16384 * (1 + j) + 1024), 16, 0, 0);
        };
        f32x4 acc[3][4][2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[j][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32x4 af[4][2], bf[2][2];
        auto readA = [&](int ao) {
            if (G8R_ABL & 2) return;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) af[mt][kb] = ld16(rdA + ao + mt * 2048 + (kb ? lane_off1 : lane_off0));
        };
        auto readB = [&](int bo, int j) {
            if (G8R_ABL & 2) return;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) bf[nt][kb] = ld16(rdB + bo + j * 16384 + nt * 2048 + (kb ? lane_off1 : lane_off0));
        };
        auto mma = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (G8R_ABL & 1) return;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[j][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(bf[nt][kb]), as_bf16x8(af[mt][kb]), acc[j][mt][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // ---- A has a ring of FOUR slots (the A part of the two K-tile buffers + two more behind them), B two buffers: the activation
        // rows come from HBM / the Infinity Cache (2-3 us away, 24-33 GB/s per CU at 72 KB in flight), the weights from this XCD's L2
        // (66-73 GB/s): A(T+3) is issued during K-tile T, three K-tiles ahead, B parts one to two K-tiles ahead.
        auto aoff = [](int t) { const int q = t & 3; return q < 2 ? q * 65536 : 131072 + (q - 2) * 16384; };
        // ---- prologue.  Issue order B0(0) A(0) | A(1) B1(0) B2(0) B0(1) A(2): the state every K-tile starts in (B0, A of it landed; A of
        // the next, its own B1, B2, B0 of the next and A of the one after that in flight)
        dmaB(0, 0, 0); dmaA(aoff(0), 0);
        if (nk > 1) dmaA(aoff(1), 1);
        dmaB(1, 0, 0); dmaB(2, 0, 0);
        if (nk > 1) dmaB(0, KS, 1);
        if (nk > 2) dmaA(aoff(2), 2);
        if (nk > 2) G8R_VM(10); else if (nk > 1) G8R_VM(8); else G8R_VM(4);
        G8R_FENCE();
        G8R_BAR();
        if (wr == 1) G8R_BAR();                // the second wave group runs one barrier behind the first
        // one K-tile: B parts on the buffer at bo, A in ring slot ao; n1 / n2 / n3: K-tiles kt+1 / kt+2 / kt+3 exist (compile-time: the
        // steady loop has no branches).  Issue order of the stream: ph1 B1(T+1) | ph2 B2(T+1), B0(T+2) | ph3 A(T+3); every phase ends
        // with a counted wait for exactly the part(s) the NEXT phase reads; four to five parts (64-80 KB) stay in flight.
        // (First version: one wait per K-tile in phase 3 with B2(T+1) issued a single phase earlier -- 1.6 us per K-tile.)
        auto ktile = [&](int bo, int kt, auto n1c, auto n2c, auto n3c) {
            constexpr bool n1 = decltype(n1c)::value, n2 = decltype(n2c)::value, n3 = decltype(n3c)::value;
            // phase 1: pair 0.  B1(Y) was last read two phases ago.  Retire B1(T) for phase 2.
            readB(bo, 0); G8R_FENCE(); readA(aoff(kt)); G8R_FENCE();
            if (n1) dmaB(1, KS - bo, kt + 1);
            G8R_LGKM(8);                        // the B0 reads have landed: phase 2 of the other wave group restages B0
            G8R_VM(2 * (1 + 2 * n1 + n2)); G8R_FENCE();
            G8R_BAR(); G8R_LGKM(0); G8R_FENCE();
            mma(std::integral_constant<int, 0>{}); G8R_FENCE();
            G8R_BAR();
            // phase 2: pair 1.  Retire B2(T) for phase 3.
            readB(bo, 1); G8R_FENCE();
            if (n1) dmaB(2, KS - bo, kt + 1);
            if (n2) dmaB(0, bo, kt + 2);
            G8R_VM(2 * (3 * n1 + 2 * n2)); G8R_FENCE();
            G8R_BAR(); G8R_LGKM(0); G8R_FENCE();
            mma(std::integral_constant<int, 1>{}); G8R_FENCE();
            G8R_BAR();
            // phase 3: pair 2.  Retire B0(T+1) (and with it the older A(T+1)) for phase 1 of the next K-tile.
            readB(bo, 2); G8R_FENCE();
            if (n3) dmaA(aoff(kt + 3), kt + 3);
            if (n1) G8R_VM(2 * (2 * n1 + 2 * n2 + n3));
            G8R_FENCE();
            G8R_BAR(); G8R_LGKM(0); G8R_FENCE();
            mma(std::integral_constant<int, 2>{}); G8R_FENCE();
            G8R_BAR();
        };
        int bo = 0, kt = 0;
        const std::true_type T_{}; const std::false_type F_{};
        for (; kt + 3 < nk; ++kt) { ktile(bo, kt, T_, T_, T_); bo = KS - bo; }
        if (kt + 2 < nk) { ktile(bo, kt, T_, T_, F_); bo = KS - bo; ++kt; }
        if (kt + 1 < nk) { ktile(bo, kt, T_, F_, F_); bo = KS - bo; ++kt; }
        ktile(bo, kt, F_, F_, F_);
        if (wr == 0) G8R_BAR();

        // ================================================================ epilogue
        // Every wave is past its last fragment read here (the resync barrier above), so the K-tile buffers are free: the WHOLE tile is
        // staged as bf16 (128 rows x 768 B = 96 KB at offset 0) behind ONE barrier, and the row phase after it has no barriers -- its
        // global loads (LNB: x, dres, mean, rstd of all 16 rows a wave owns) are all in flight together.  (First version: four passes
        // through a 24 KB staging area, a barrier pair and an exposed load latency per pass -- 29 us against 31 for the two launches.)
        // (the lane index is laundered through an empty asm: everything below is a function of it, and the compiler otherwise computes
        // the epilogue's addresses BEFORE the K loop, spills them -- 56 / 124 registers -- and reloads them one by one down here)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int fr = lane_e & 15, g = lane_e >> 4;
        const unsigned stg_a = g8_lds_addr(smem);
        const int rr = lane_e >> 4, l16 = lane_e & 15;                // row phase: rows wave*16 + step*4 + rr, chunks l16 + 16 i
        const float invC = 1.0f / (float)NC;
        // ---- accumulators (+ bias, row scale, residual in fp32) -> bf16 -> staging, MFMA layout: row fr of the 16-row tile, 8 columns per pair
        {
            u32x4 rin[4][3];
            if (MODE == 1 && ep.residual) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf16_t* rp = ep.residual + (int64_t)min(m0 + wr * 64 + mt * 16 + fr, ga.M - 1) * ep.ldr + wc * 96 + g * 8;
#pragma unroll
                    for (int j = 0; j < 3; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rin[mt][j]) : "v"(rp + j * 32) : "memory");
                }
            }
            float bias[3][8];
            if (MODE == 1 && ep.bias) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int c = wc * 96 + j * 32 + g * 8;
                    *reinterpret_cast<float4*>(bias[j]) = *reinterpret_cast<const float4*>(ep.bias + c);
                    *reinterpret_cast<float4*>(bias[j] + 4) = *reinterpret_cast<const float4*>(ep.bias + c + 4);
                }
            }
            float rsv[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                rsv[mt] = (MODE == 1 && ep.row_scale) ? ep.row_scale[min(m0 + wr * 64 + mt * 16 + fr, ga.M - 1) / ep.rows_per_scale] : 1.f;
            if (MODE == 1 && ep.residual) {
                G8R_VM(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(rin[mt][j]));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wr * 64 + mt * 16 + fr;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[j][mt][0][r]; v[4 + r] = acc[j][mt][1][r]; }
                    if (MODE == 1) {
                        if (ep.bias) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += bias[j][q];
                        }
                        if (ep.row_scale) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] *= rsv[mt];
                        }
                        if (ep.residual) {
                            float h[8];
                            unpack8(rin[mt][j], h);
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += h[q];
                        }
                    }
                    const int chunk = wc * 12 + j * 4 + g;
                    g8_lds_st16(stg_a + row * RS + ((chunk ^ (row & 7)) << 4), pack8(v));
                }
            }
        }
        // ---- row-phase operands of LNB, issued behind the staging writes (the accumulators are dead by then), in front of the barrier
        // (two steps ahead, not all four: 96 registers of operands on top of the 48 dgamma / dbeta sums spilled)
        u32x4 xin[4][3], din[4][3];
        float mu_in[4], rs_in[4];
        auto row_in = [&](int sp) {
            const int mc = min(m0 + wave * 16 + sp * 4 + rr, ga.M - 1);
            mu_in[sp] = ga.mean[mc]; rs_in[sp] = ga.rstd[mc];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                xin[sp][i] = ld16(ga.x + (int64_t)mc * ga.ldx + (l16 + 16 * i) * 8);
                din[sp][i] = ga.dres ? ld16(ga.dres + (int64_t)mc * ga.lddres + (l16 + 16 * i) * 8) : u32x4{0u, 0u, 0u, 0u};
            }
        };
        if (MODE == 2) { row_in(0); row_in(1); }
        float gam[24], bet[24];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = (l16 + 16 * i) * 8;
            *reinterpret_cast<float4*>(gam + 8 * i) = *reinterpret_cast<const float4*>(ga.gamma + c);
            *reinterpret_cast<float4*>(gam + 8 * i + 4) = *reinterpret_cast<const float4*>(ga.gamma + c + 4);
            if (MODE == 1) {
                *reinterpret_cast<float4*>(bet + 8 * i) = *reinterpret_cast<const float4*>(ga.beta + c);
                *reinterpret_cast<float4*>(bet + 8 * i + 4) = *reinterpret_cast<const float4*>(ga.beta + c + 4);
            }
        }
        float dg[24], db[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) { dg[i] = 0.f; db[i] = 0.f; }
        G8R_LGKM(0);
        G8R_BAR();
        // ---- row phase: 16 lanes own a row (3 chunks of 8 columns), four rows of the wave in flight, four steps
        u32x4 xs[4][3];
        if (MODE == 1) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int row = wave * 16 + sp * 4 + rr;
#pragma unroll
                for (int i = 0; i < 3; ++i) xs[sp][i] = g8_lds_ld16(stg_a + row * RS + (((l16 + 16 * i) ^ (row & 7)) << 4));
            }
            G8R_LGKM(0);
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(xs[sp][i]));
        }
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const int m = m0 + wave * 16 + sp * 4 + rr;
            const bool ok = m < ga.M;
            if (MODE == 2) {
                G8R_FENCE();
                if (sp + 2 < 4) row_in(sp + 2);
                const int row = wave * 16 + sp * 4 + rr;
#pragma unroll
                for (int i = 0; i < 3; ++i) xs[sp][i] = g8_lds_ld16(stg_a + row * RS + (((l16 + 16 * i) ^ (row & 7)) << 4));
                G8R_LGKM(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(xs[sp][i]));
            }
            float v[24];
#pragma unroll
            for (int i = 0; i < 3; ++i) unpack8(xs[sp][i], v + 8 * i);
            if (MODE == 1) {
                // x1 out, LayerNorm of the ROUNDED row (what the separate kernel read back), xn / mean / rstd out
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) st16(ga.C + (int64_t)m * ga.ldc + (l16 + 16 * i) * 8, xs[sp][i]);
                }
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 24; ++q) s += v[q];
                s = group_sum<16>(s);
                const float mu = s * invC;
                float qv = 0.f;
#pragma unroll
                for (int q = 0; q < 24; ++q) { const float d = v[q] - mu; qv += d * d; }
                qv = group_sum<16>(qv);
                const float rstd = rsqrtf(qv * invC + ga.eps);
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        float o[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) o[q] = (v[8 * i + q] - mu) * rstd * gam[8 * i + q] + bet[8 * i + q];
                        st16(ga.xn + (int64_t)m * ga.ldxn + (l16 + 16 * i) * 8, pack8(o));
                    }
                    if (l16 == 0) { ga.mean[m] = mu; ga.rstd[m] = rstd; }
                }
            } else {
                // LayerNorm backward of the row: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) + dres, g = dxn * gamma
                const float mu = mu_in[sp], rstd = rs_in[sp];
                float xh[24];
#pragma unroll
                for (int i = 0; i < 3; ++i) unpack8(xin[sp][i], xh + 8 * i);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    xh[q] = (xh[q] - mu) * rstd;
                    const float gq = v[q] * gam[q];
                    s1 += gq; s2 += gq * xh[q];
                }
                s1 = group_sum<16>(s1) * invC; s2 = group_sum<16>(s2) * invC;
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        float o[8], dr[8];
                        unpack8(din[sp][i], dr);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int k = 8 * i + q;
                            o[q] = rstd * (v[k] * gam[k] - s1 - xh[k] * s2) + dr[q];
                            dg[k] += v[k] * xh[k]; db[k] += v[k];
                        }
                        st16(ga.C + (int64_t)m * ga.ldc + (l16 + 16 * i) * 8, pack8(o));
                    }
                }
            }
        }
        G8R_BAR();                             // the staging area = the K-tile buffers of the next tile
        if (MODE == 2) {
            // ---- column sums of the tile: 4 rows per wave (lanes 16 apart), 8 waves -> through the staging region -> one partial row
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                dg[q] += __shfl_xor(dg[q], 16, 64); dg[q] += __shfl_xor(dg[q], 32, 64);
                db[q] += __shfl_xor(db[q], 16, 64); db[q] += __shfl_xor(db[q], 32, 64);
            }
            float* red = reinterpret_cast<float*>(smem + STG);          // [8 waves][768], behind the staged tile
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = (l16 + 16 * i) * 8 + q;
                        red[wave * 768 + c] = dg[8 * i + q];
                        red[wave * 768 + NC + c] = db[8 * i + q];
                    }
            }
            __syncthreads();
            for (int c = tid; c < 768; c += 512) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) t += red[w * 768 + c];
                ga.partial[(int64_t)tile * 768 + c] = t;
            }
            __syncthreads();
        }
    }
    G8R_VM(0);
}
