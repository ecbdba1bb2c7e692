// 256 x 256 x 64 bf16 NT GEMM for gfx950 with the 8-phase, two-wave-group schedule of the CDNA4 guide
// (cdna_hip_programming.md "The 256^2 8-phase template"): ONE persistent 512-thread workgroup per CU,
// operands global -> LDS by LDS-DMA (global_load_lds_dwordx4) into two K-tile buffers of four 16 KB
// half-tiles each, counted vmcnt (never 0 in the loop), raw s_barrier, the two wave groups (waves 0-3 /
// 4-7, one of each per SIMD) one barrier apart so that one group's 16-MFMA cluster runs beside the other
// group's LDS reads and DMA issue.
//
//   C[M,N] = epi(A[M,K] . B[N,K]^T)       (the Linear layers of models/volo.py:67,68,71,156,158,180,182 and
//                                          their input gradients; same contract as k_gemm_nt in gemm.hip)
//
// Geometry.  Wave (wr, wc) = (wave >> 2, wave & 3).  The block tile is cut into half-tiles that a PHASE
// consumes whole, so that a half-tile is dead (restageable) one or two phases after it was read:
//   A half h  = block rows h*128 .. h*128+127; wave (wr, .) reads rows wr*64 .. wr*64+63 of it
//   B half h  = for every wc the 32 columns wc*64 + h*32 .. +31 (image rows wc*32 .. wc*32+31)
// A wave therefore owns output rows {h*128 + wr*64 + [0,64)} x columns {wc*64 + [0,64)}: four 64 x 32
// quadrants (mh, nh), one per phase: (0,0) (0,1) (1,1) (1,0).  B fragments of both halves stay in
// registers for the whole K-tile, A fragments of one half at a time.
//
// LDS image of a half-tile: 16 subtiles [16 rows][32 k] of 1024 B (= one wave-instruction of the DMA),
// subtile (rb, kb) at ((rb * 2 + kb) * 1024); inside a subtile byte (r * 64 + c * 2) ^ (((r >> 3) & 1) << 5)
// (the guide's st_16x32 swizzle: a 16-lane ds_read_b128 group then covers all 64 banks once).  The DMA writes
// lane-linear, so the swizzle is applied to the per-lane SOURCE address and again to the read address.
//
// The MFMA is issued as D = Bfrag (A operand) x Afrag (B operand): a lane ends with 4 consecutive output
// columns of one row; the B image rows are permuted (row wc*32 + nt*16 + i  <->  column wc*64 + h*32 + (i>>2)*8
// + nt*4 + (i&3)) so that the two 16-column tiles of a quadrant give a lane 8 CONSECUTIVE columns: 16-byte
// stores / epilogue-operand loads, 64 contiguous bytes per row and instruction.
//
// Stream.  A workgroup walks its output tiles (tile = first + j * gridDim.x) and their K-tiles as ONE stream
// of half-tile loads, 7 half-tiles ahead of the reads at the prologue and 3 in flight behind every counted
// wait: the next output tile's first K-tiles are already landing while the current tile's epilogue runs.
#pragma once
#include "common.h"
#include "gemm_epi.h"

#define G8_LDS_BYTES (2 * 4 * 16384)

#define G8_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define G8_LDS(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef G8_ABL
#define G8_ABL 0          // timing-only ablations (lab builds): 1 no MFMA, 2 no LDS reads, 4 no DMA, 8 no epilogue stores
#endif

struct G8Args {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;
    bf16_t* C; int ldc;
    int M, N, K;
    int tiles_n, ntiles;
};

template <int EPI_MODE>
__global__ void __launch_bounds__(512, 2) k_gemm_nt_8p(G8Args ga, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char g8_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, g = lane >> 4;
    const int nk = ga.K >> 6;

    // ---- DMA source geometry (per lane): image row srow of a half-tile, 16-byte chunk scol of a 32-wide k block
    const int lp = lane ^ ((lane >> 5) << 1);
    const int srow = wave * 16 + (lp >> 2);
    const int scol = (lp & 3) * 8;
    const int bi = srow & 15;
    const int bcol = (srow >> 5) * 64 + (bi >> 2) * 8 + ((srow >> 4) & 1) * 4 + (bi & 3);      // + h * 32
    unsigned char* const dma_dst = g8_smem + wave * 2048;                                       // + slot + kb * 1024

    // ---- fragment read addresses
    const int lane_off = fr * 64 + ((g ^ ((fr >> 3) << 1)) << 4);
    const unsigned char* const rdA = g8_smem + wr * 8192 + lane_off;          // + buf*65536 + h*16384 + (mt*2+kb)*1024
    const unsigned char* const rdB = g8_smem + 32768 + wc * 4096 + lane_off;  // + buf*65536 + h*16384 + (nt*2+kb)*1024

    // ---- issue side of the stream
    int q_tile = blockIdx.x, q_kt = 0;
    const bf16_t *qa0, *qa1, *qb0, *qb1;
    auto set_q = [&]() {
        const int t = q_tile < ga.ntiles ? q_tile : ga.ntiles - 1;
        const int m0 = (t / ga.tiles_n) * 256, n0 = (t % ga.tiles_n) * 256;
        qa0 = ga.A + (int64_t)min(m0 + srow, ga.M - 1) * ga.lda + scol;
        qa1 = ga.A + (int64_t)min(m0 + 128 + srow, ga.M - 1) * ga.lda + scol;
        qb0 = ga.B + (int64_t)min(n0 + bcol, ga.N - 1) * ga.ldb + scol;
        qb1 = ga.B + (int64_t)min(n0 + bcol + 32, ga.N - 1) * ga.ldb + scol;
    };
    set_q();
    // half-tile slots inside a K-tile buffer: 0 = A h0, 1 = A h1, 2 = B h0, 3 = B h1
    auto dma = [&](const bf16_t* src, int bufoff, int slot) {           // bufoff = 0 / 65536: the K-tile buffer
        if (!(G8_ABL & 4)) {
            __builtin_amdgcn_global_load_lds(G8_GLB(src + q_kt * 64), G8_LDS(dma_dst + bufoff + slot * 16384), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(G8_GLB(src + q_kt * 64 + 32), G8_LDS(dma_dst + bufoff + slot * 16384 + 1024), 16, 0, 0);
        }
    };
    auto q_advance = [&]() {
        if (++q_kt == nk) { q_kt = 0; q_tile += gridDim.x; set_q(); }
    };

    f32x4 acc[2][2][4][2];
    u32x4 af[4][2], bf[2][2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto readA = [&](const unsigned char* base, int h) {
        if (!(G8_ABL & 2)) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) af[mt][kb] = ld16(base + h * 16384 + (mt * 2 + kb) * 1024);
        }
    };
    auto readB = [&](const unsigned char* base, int h) {
        if (!(G8_ABL & 2)) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) bf[h][nt][kb] = ld16(base + h * 16384 + (nt * 2 + kb) * 1024);
        }
    };
    auto mma = [&](int mh, int nh) {
        if (!(G8_ABL & 1)) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mh][nh][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(bf[nh][nt][kb]), as_bf16x8(af[mt][kb]),
                                                                                     acc[mh][nh][mt][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };
#define G8_BAR() __builtin_amdgcn_s_barrier()
#define G8_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define G8_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G8_FENCE() __builtin_amdgcn_sched_barrier(0)

    // One K-tile = four phases on buffer BUF.  q_live: the half-tiles issued here are real (more stream left).
    auto ktile = [&](int bo) {               // bo = 0 / 65536: byte offset of this K-tile's buffer
        const unsigned char* const ra = rdA + bo;
        const unsigned char* const rb = rdB + bo;
        // phase 1: quadrant (0,0); completes K-tile T+1 (A h1 into the other buffer)
        readB(rb, 0); G8_FENCE(); readA(ra, 0); G8_FENCE();
        const bool live1 = q_tile < ga.ntiles;
        if (live1) dma(qa1, bo ^ 65536, 1);
        q_advance();
        G8_LGKM(8); G8_FENCE();
        G8_BAR(); G8_LGKM(0); G8_FENCE();
        mma(0, 0); G8_FENCE();
        G8_BAR();
        // phase 2: quadrant (0,1); B h0 of K-tile T+2 into this buffer (its reads were retired by the lgkmcnt(8) above)
        const bool live = q_tile < ga.ntiles;
        readB(rb, 1); G8_FENCE();
        if (live) dma(qb0, bo, 2);
        G8_BAR(); G8_LGKM(0); G8_FENCE();
        mma(0, 1); G8_FENCE();
        G8_BAR();
        // phase 3: quadrant (1,1); A h0 of K-tile T+2
        readA(ra, 1); G8_FENCE();
        if (live) dma(qa0, bo, 0);
        G8_BAR(); G8_LGKM(0); G8_FENCE();
        mma(1, 1); G8_FENCE();
        G8_BAR();
        // phase 4: quadrant (1,0); B h1 of K-tile T+2; the counted wait retires the OTHER buffer (K-tile T+1)
        if (live) { dma(qb1, bo, 3); G8_VM(6); } else { G8_VM(0); }
        G8_FENCE();
        G8_BAR();
        mma(1, 0); G8_FENCE();
        G8_BAR();
    };

    // ---- prologue: K-tile 0 whole, K-tile 1 without its A h1
    {
        dma(qb0, 0, 2); dma(qa0, 0, 0); dma(qb1, 0, 3); dma(qa1, 0, 1);
        q_advance();
        if (q_tile < ga.ntiles) { dma(qb0, 65536, 2); dma(qa0, 65536, 0); dma(qb1, 65536, 3); G8_VM(6); } else { G8_VM(0); }
        G8_FENCE();
        G8_BAR();
    }
    if (wr == 1) G8_BAR();            // the second wave group runs one barrier behind the first

    int bo = 0;
    for (int tile = blockIdx.x; tile < ga.ntiles; tile += gridDim.x) {
        zero_acc();
        for (int kt = 0; kt < nk; ++kt) { ktile(bo); bo ^= 65536; }
        // ---- epilogue, straight from the accumulators: lane (fr, g) holds row fr of each 16-row tile, 8 consecutive columns per quadrant
        const int m0 = (tile / ga.tiles_n) * 256, n0 = (tile % ga.tiles_n) * 256;
        const bool vec_ok = (ga.ldc & 7) == 0;
        if (!(G8_ABL & 8)) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int m = m0 + mh * 128 + wr * 64 + mt * 16 + fr;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const int n = n0 + wc * 64 + nh * 32 + g * 8;
                        if (m < ga.M && n < ga.N) {
                            float v[8];
#pragma unroll
                            for (int r = 0; r < 4; ++r) { v[r] = acc[mh][nh][mt][0][r]; v[4 + r] = acc[mh][nh][mt][1][r]; }
                            epi_chunk(v, m, n, ga.N, ga.ldc, vec_ok, ep, ga.C);
                        }
                    }
                }
        }
    }
    if (wr == 0) G8_BAR();
    G8_VM(0);
}
