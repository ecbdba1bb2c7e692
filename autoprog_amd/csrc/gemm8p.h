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
// Geometry.  Block tile 256 x BN, BN = 4 * WN, WN = 32 + 16 * NT1 (NT1 = 2: 256 x 256; NT1 = 1: 256 x 192 -- every Linear
// width of VOLO-D1 except the heads is a multiple of 192).  Wave (wr, wc) = (wave >> 2, wave & 3).  The block tile is cut
// into parts that a PHASE consumes whole, so that a part is dead (restageable) one or two phases after it was read:
//   A half h  = block rows h*128 .. h*128+127; wave (wr, .) reads rows wr*64 .. wr*64+63 of it
//   B part 0  = for every wc the 32 columns wc*WN .. +31          (image rows wc*32 .. wc*32+31)
//   B part 1  = for every wc the 16*NT1 columns wc*WN + 32 .. (image rows wc*16*NT1 ..)
// A wave therefore owns output rows {h*128 + wr*64 + [0,64)} x columns {wc*WN + [0,WN)}: four quadrants (mh, part), one per
// phase: (0,0) (0,1) (1,1) (1,0) with 16 / 8*NT1 / 8*NT1 / 16 MFMAs.  B fragments of both parts stay in registers for the
// whole K-tile, A fragments of one half at a time.
//
// LDS image of a half-tile: 16 subtiles [16 rows][32 k] of 1024 B (= one wave-instruction of the DMA),
// subtile (rb, kb) at ((rb * 2 + kb) * 1024); inside a subtile byte (r * 64 + c * 2) ^ (((r >> 3) & 1) << 5)
// (the guide's st_16x32 swizzle: a 16-lane ds_read_b128 group then covers all 64 banks once).  The DMA writes
// lane-linear, so the swizzle is applied to the per-lane SOURCE address and again to the read address.
//
// The MFMA is issued as D = Bfrag (A operand) x Afrag (B operand): a lane ends with 4 consecutive output
// columns of one row; the B image rows are permuted (row wc*32 + nt*16 + i  <->  column wc*64 + h*32 + (i>>2)*8
// + nt*4 + (i&3)) so that the two 16-column tiles of a quadrant give a lane 8 CONSECUTIVE columns: 16-byte
// stores / epilogue-operand loads, 64 contiguous bytes per row and instruction (the single 16-column tile of part 1 at
// NT1 = 1 keeps its natural order: 4 consecutive columns, 8-byte accesses).
//
// Tile end.  The first wave group waits one barrier for the second (both then hold their results), the epilogues of all eight
// waves run together, straight from the accumulators, and the second group falls one barrier behind again.  The load stream is
// not interrupted: the next tile's first K-tiles land under the epilogue.  (Measured and dropped: the epilogue cut into quadrants
// and run inside the load sections of the following four phases -- bias / row scale by LDS-DMA, second operand by hand-counted asm
// loads; tools/gemm_lab/gemm8p_distributed_epilogue.h.txt.  A CU stores ~10 B/clk, so a quadrant's stores hold its wave group's
// load section for ~1600 cycles against the ~256 of the other group's MFMA cluster, and the two groups' epilogues, which run side
// by side here, then run one after the other: qkv shape 38.9 -> 43.1 us, fc1 + GELU 68 -> 84 us.)
//
// Stream.  A workgroup walks its output tiles (tile = first + j * gridDim.x) and their K-tiles as ONE stream
// of half-tile loads, 7 half-tiles ahead of the reads at the prologue and 3 in flight behind every counted
// wait: the next output tile's first K-tiles are already landing while the current tile's epilogue runs.
#pragma once
#include "common.h"
#include "gemm_epi.h"
#include <type_traits>

#ifndef G8_NT_STORE
#define G8_NT_STORE 1          // the epilogue's row stores bypass L2 allocation (nontemporal): the outputs are read by the NEXT kernel, behind an L2 write-back anyway
#endif
#define G8_LDS_BYTES 163840                    // all of the CU's LDS: two K-tile buffers + the epilogue staging region

#define G8_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define G8_LDS(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef G8_ABL
#define G8_ABL 0          // timing-only ablations (lab builds): 1 no MFMA, 2 no LDS reads, 4 no DMA, 8 no epilogue (accumulators kept alive)
#endif

// LDS accesses of the epilogue in inline asm: behind a pending LDS-DMA (the stream's prefetch of the next tile) hipcc guards every
// ds_read / ds_write it can see with s_waitcnt vmcnt(0), which also waits for the epilogue's own stores, one by one.  These are
// counted by hand (lgkmcnt) and never alias the K-tile buffers the DMA writes.
__device__ __forceinline__ unsigned g8_lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void g8_lds_st16(unsigned a, const u32x4& v) { asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void g8_lds_st8(unsigned a, const u32x2& v) { asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 g8_lds_ld16(unsigned a) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ u32x2 g8_lds_ld8(unsigned a) { u32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ unsigned g8_lds_ld4(unsigned a) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }

struct G8Args {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;
    bf16_t* C; int ldc;
    int M, N, K;
    int tiles_n, ntiles;
};

// epilogue flavour of an instantiation (EF >= 0: bits known at compile time; EF < 0: read from the arguments at run time).  The
// generic epilogue is ~25 KB of code that a CU runs once or twice per launch, cold: an instantiation per flavour of the training
// step keeps what is fetched to what is used (2.8 us -> see DESIGN.md on the 25088 x 384 x 1152 launch).
enum { G8_BIAS = 1, G8_GELU = 2, G8_DGELU = 4, G8_RS = 8, G8_RES = 16, G8_MUL = 32, G8_MUL8 = 64, G8_GTAB = 128, G8_Q8 = 256 };
__host__ __device__ inline int g8_flavour(const EpiArgs& ep) {
    return (ep.bias ? G8_BIAS : 0) | (ep.gelu ? G8_GELU : 0) | (ep.dgelu_of ? G8_DGELU : 0) | (ep.row_scale ? G8_RS : 0) | (ep.residual ? G8_RES : 0) |
           (ep.mul_by ? G8_MUL : 0) | (ep.mul8 ? G8_MUL8 : 0) | ((ep.gelu == 3 && ep.gelu_tab) ? G8_GTAB : 0) |
           ((ep.q8 && !ep.gelu) ? G8_Q8 : 0);        // (the fp8 GELU launches emit their e4m3 side output without a flavour bit of their own)
}

// NT1: B part 1 holds NT1 16-column tiles per wave (1: 256 x 192 block tile, 2: 256 x 256).
// FP8: A and B hold OCP e4m3 bytes (configs[4]: "mixed MFMA fp8 GEMM"); the kernel is launched on byte PAIRS (lda, ldb, K in 2-byte units:
// a K-tile is 128 bytes of a row either way), the two 16-byte fragments of a row feed one K = 128 MFMA (the same K set on both
// operands), and the accumulators are multiplied by dq_a[0] * dq_b[0] before the epilogue.  Half the operand bytes per FLOP.
// (one v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales on the two 16-byte fragments of a row's 128-byte K-tile: the double-rate
// fp8 instruction of gfx950; the pair of v_mfma_f32_16x16x32_fp8_fp8 per fragment of the first version ran at the bf16 rate)
__device__ __forceinline__ void g8_store(bf16_t* p, const u32x4& v) {
    if (G8_NT_STORE) st16_nt(p, v); else st16(p, v);
}
typedef int __attribute__((ext_vector_type(8))) g8_i32x8;
__device__ __forceinline__ f32x4 g8_mma_bf16(const u32x4& b, const u32x4& a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b), as_bf16x8(a), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 g8_mma_fp8(const u32x4& b0, const u32x4& b1, const u32x4& a0, const u32x4& a1, f32x4 c) {
    const g8_i32x8 b = __builtin_bit_cast(g8_i32x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
    const g8_i32x8 a = __builtin_bit_cast(g8_i32x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
// BM = 224 (NT1 = 1 only; round 5): a 224-row block tile -- the second A half contributes 96 rows, three 16-row tiles per wave instead of four.
// 25088 rows (VOLO-D1, B = 128, 196 tokens) are 98 tiles of 256 rows: the N = 384 products of a transformer block are 196 tiles on 256 CUs,
// and tools/tile_rounds_probe.py shows that such a launch takes (nearly) the time of ONE tile whatever the number of idle CUs
// (T(256 tiles) / T(196 tiles) = 1.06 - 1.16): 112 tiles of 224 rows x 2 = 224 tiles do 7 / 8 of the work per CU.  The DMA stream is
// unchanged (the second half still stages 128 rows: its last 32 are the next tile's and are not read).
template <int NT1, int EF, bool FP8 = false, int BM = 256>
__global__ void __launch_bounds__(512, 2) k_gemm_nt_8p(G8Args ga, EpiArgs ep) {
    static_assert(BM == 256 || (BM == 224 && NT1 == 1 && !FP8), "block rows");
    constexpr int MT1 = BM == 224 ? 3 : 4;       // 16-row tiles per wave in the second A half
    constexpr int R1 = MT1 * 16;                 // ... = rows per wave there
    const bool has_bias = EF < 0 ? ep.bias != nullptr : (EF & G8_BIAS) != 0;
    const bool has_gelu = EF < 0 ? ep.gelu != 0 : (EF & G8_GELU) != 0;
    const bool has_dgelu = EF < 0 ? ep.dgelu_of != nullptr : (EF & G8_DGELU) != 0;
    const bool has_rs = EF < 0 ? ep.row_scale != nullptr : (EF & G8_RS) != 0;
    const bool has_res = EF < 0 ? ep.residual != nullptr : (EF & G8_RES) != 0;
    const bool has_mul = EF < 0 ? ep.mul_by != nullptr : (EF & G8_MUL) != 0;
    const bool has_mul8 = EF < 0 ? ep.mul8 != nullptr : (EF & G8_MUL8) != 0;       // the 8-bit gelu' codes of a gelu = 3 forward
    // Q8C: this instantiation can write its output a second time as e4m3 bytes (ep.q8): the fp8 GELU launches (the operand of fc2), and the
    // G8_Q8 flavours of the bf16 kernel (round 5: the input gradient of fc2, dL/dh -- the operand of fc1's fp8 input-gradient product)
    constexpr bool Q8C = FP8 || (EF >= 0 && (EF & G8_Q8) != 0);
    float qmx = 0.f;                              // Q8C with ep.q8: running max |output| of this lane, and the quantisation scale
    const float qsc = (Q8C && ep.q8) ? ep.q8_scale[0] : 1.f;
    constexpr int WN = 32 + 16 * NT1, BN = 4 * WN;
    constexpr int VMN = 4 + NT1;                 // DMA instructions of the three parts in flight behind a counted wait
    constexpr int KS = 49152 + 8192 * NT1;       // bytes of a K-tile buffer (A h0 | A h1 | B part 0 | B part 1); the two buffers are adjacent
    constexpr int STG = 2 * KS;                  // epilogue staging region: the rest of the LDS (48 KB at BN = 192, 32 KB at BN = 256)
    // GTAB (the GELU flavours): 16 KB of the staging region hold the GELU table of gemm_epi.h (gelu = 3 launches: Phi(h) and the 8-bit
    // derivative code per bf16 value); the staging passes are half as tall
    constexpr bool GTAB = EF >= 0 && (EF & G8_GELU) != 0 && (EF & G8_GTAB) != 0;        // (the gelu = 1 / 2 launches keep the tall passes: 55.9 against 61.4 us)
    constexpr int PASS_MT = (NT1 == 1 ? 4 : 2) / (GTAB ? 2 : 1);    // 16-row tiles per wave group and staging pass
    constexpr int NPASS = 8 / PASS_MT;
    constexpr int RS = BN * 2;                   // bytes of a staged row
    constexpr int GRP = 16 * PASS_MT * RS;       // bytes of a wave group's staging area (24 KB / 16 KB)
    constexpr int TAB = STG + 2 * GRP;           // byte offset of the GELU table
    static_assert(TAB + (GTAB ? 16384 : 0) <= G8_LDS_BYTES, "staging region");
    extern __shared__ __attribute__((aligned(16))) unsigned char g8_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, g = lane >> 4;
    const int nk = ga.K >> 6;

    // ---- DMA source geometry (per lane).  A wave-instruction fills one 1 KB piece: 8 image rows x 64 k, i.e. it fetches eight
    // whole 128-byte lines (16 rows x 32 k, half lines, measured slower: twice the L2 requests).  Lane l writes LDS bytes
    // [16 l, 16 l + 16) of the piece, which hold 16-byte chunk (l & 7) ^ (row & 7) of row l >> 3: the swizzle is applied to the
    // SOURCE address and again to the read address.
    // K-tile buffer (KS bytes, the second one right behind the first): A h0 at 0, A h1 at 16384, B part 0 at 32768, B part 1 at
    // 49152; image row r of a part at (r >> 3) * 1024 + (r & 7) * 128.
    const int sr8 = lane >> 3;                                   // row inside the piece
    const int scol = ((lane & 7) ^ sr8) * 8;                     // element column inside the 64-wide k tile
    // 128-row parts: wave w fills pieces 2w, 2w + 1 = image rows 16 w .. 16 w + 15
    const int srow = wave * 16 + sr8;                            // (+ 8 for the second piece)
    // B part 0: image row wc*32 + nt*16 + i  <->  column wc*WN + (i >> 2)*8 + nt*4 + (i & 3)
    auto bcol0_of = [&](int r) { const int i = r & 15; return (r >> 5) * WN + (i >> 2) * 8 + ((r >> 4) & 1) * 4 + (i & 3); };
    // part 1: NT1 = 2 -> 128 image rows, same pairing; NT1 = 1 -> 64 image rows in natural order, ONE piece per wave (rows 8 w ..)
    const int srow1 = NT1 == 2 ? srow : wave * 8 + sr8;
    auto bcol1_of = [&](int r) { return NT1 == 2 ? bcol0_of(r) + 32 : (r >> 4) * WN + 32 + (r & 15); };
    unsigned char* const dma_dst = g8_smem + wave * 2048;
    unsigned char* const dma_dst1 = g8_smem + 49152 + (NT1 == 2 ? wave * 2048 : wave * 1024);

    // ---- fragment read addresses: row fr of a 16-row tile, chunk kb * 4 + g
    const int lane_row = (fr >> 3) * 1024 + (fr & 7) * 128;
    const int lane_off0 = lane_row + ((g ^ (fr & 7)) << 4);              // kb = 0
    const int lane_off1 = lane_row + (((4 + g) ^ (fr & 7)) << 4);        // kb = 1
    const unsigned char* const rdA = g8_smem + wr * 8192;                          // + bo + h*16384 + mt*2048 + lane_off{kb}
    const unsigned char* const rdA1 = g8_smem + wr * (R1 * 128);                   // the second half's rows of this wave (BM = 224: 48 per wave)
    const unsigned char* const rdB0 = g8_smem + 32768 + wc * 4096;                 // + bo + nt*2048 + lane_off{kb}
    const unsigned char* const rdB1 = g8_smem + 49152 + wc * (2048 * NT1);         // + bo + nt*2048 + lane_off{kb}

    // ---- tile walk: the workgroups that share an XCD (blockIdx % 8, dealt round-robin) take a CONTIGUOUS range of tile ids
    // (n fastest), so the tiles in flight on one XCD share their A row panels and the weight panel in that XCD's L2
    const int nxw = (gridDim.x + 7) >> 3;                          // workgroups per XCD label (the grid is a multiple of 8)
    const int xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int xq = ga.ntiles >> 3, xr = ga.ntiles & 7;
    const int t_begin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int t_end = (xcd < xr ? (xcd + 1) * (xq + 1) : xr * (xq + 1) + (xcd + 1 - xr) * xq);
    if (t_begin >= t_end) return;                                  // (whole workgroup; before any barrier)
    const bool use_tab = GTAB && ep.gelu == 3 && ep.gelu_tab != nullptr;
    if constexpr (GTAB) {
        // the table by LDS-DMA, two 1 KB pieces per wave, in FRONT of the operand stream: every counted wait of the stream retires them
        // first (in-order vmcnt), the K loop's barriers publish them long before the first epilogue reads
        if (use_tab) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds(G8_GLB(ep.gelu_tab + (wave * 2 + j) * 256 + lane * 4), G8_LDS(g8_smem + TAB + (wave * 2 + j) * 1024), 16, 0, 0);
        }
    }

    // ---- issue side of the stream
    int q_tile = t_begin, q_kt = 0;
    // per-lane source pointers of the two pieces of each part (row pointers: rows 8 apart are NOT a constant apart at the matrix edge)
    const bf16_t *qa0[2], *qa1[2], *qb0[2], *qb1[NT1];
    auto set_q = [&]() {
        const int t = q_tile < t_end ? q_tile : t_end - 1;
        const int m0 = (t / ga.tiles_n) * BM, n0 = (t % ga.tiles_n) * BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            qa0[j] = ga.A + (int64_t)min(m0 + srow + 8 * j, ga.M - 1) * ga.lda + scol;
            qa1[j] = ga.A + (int64_t)min(m0 + 128 + srow + 8 * j, ga.M - 1) * ga.lda + scol;
            qb0[j] = ga.B + (int64_t)min(n0 + bcol0_of(srow + 8 * j), ga.N - 1) * ga.ldb + scol;
        }
#pragma unroll
        for (int j = 0; j < NT1; ++j) qb1[j] = ga.B + (int64_t)min(n0 + bcol1_of(srow1 + 8 * j), ga.N - 1) * ga.ldb + scol;
    };
    set_q();
    auto dma = [&](const bf16_t* const* src, int bufoff, int part) {   // parts 0 / 1 / 2: A h0, A h1, B part 0 (two pieces per wave)
        if (!(G8_ABL & 4)) {
            __builtin_amdgcn_global_load_lds(G8_GLB(src[0] + q_kt * 64), G8_LDS(dma_dst + bufoff + part * 16384), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(G8_GLB(src[1] + q_kt * 64), G8_LDS(dma_dst + bufoff + part * 16384 + 1024), 16, 0, 0);
        }
    };
    auto dma1 = [&](int bufoff) {                                        // B part 1
        if (!(G8_ABL & 4)) {
            __builtin_amdgcn_global_load_lds(G8_GLB(qb1[0] + q_kt * 64), G8_LDS(dma_dst1 + bufoff), 16, 0, 0);
            if constexpr (NT1 == 2) __builtin_amdgcn_global_load_lds(G8_GLB(qb1[NT1 - 1] + q_kt * 64), G8_LDS(dma_dst1 + bufoff + 1024), 16, 0, 0);
        }
    };
    auto q_advance = [&]() {
        if (++q_kt == nk) { q_kt = 0; q_tile += nxw; set_q(); }
    };

    f32x4 acc0[2][4][2], acc1[2][4][NT1];        // [mh][mt][nt] of part 0 / part 1
    u32x4 af[4][2], bf0[2][2], bf1[NT1][2];
    u32x4 af1[NT1 == 1 ? 4 : 1][2];              // BN = 192: the second A half has its own registers (three phases per K-tile)
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int d = 0; d < 2; ++d) acc0[a][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < NT1; ++d) acc1[a][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    };
    auto readA_to = [&](u32x4 (&a)[4][2], const unsigned char* base, int h, int nmt = 4) {
        if (!(G8_ABL & 2)) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    if (mt < nmt) a[mt][kb] = ld16(base + h * 16384 + mt * 2048 + (kb ? lane_off1 : lane_off0));
        }
    };
    auto readA = [&](const unsigned char* base, int h) { readA_to(af, base, h); };
    auto readB0 = [&](const unsigned char* base) {
        if (!(G8_ABL & 2)) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) bf0[nt][kb] = ld16(base + nt * 2048 + (kb ? lane_off1 : lane_off0));
        }
    };
    auto readB1 = [&](const unsigned char* base) {
        if (!(G8_ABL & 2)) {
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) bf1[nt][kb] = ld16(base + nt * 2048 + (kb ? lane_off1 : lane_off0));
        }
    };
    auto mma0_of = [&](int mh, const u32x4 (&a)[4][2], int nmt = 4) {
        if (!(G8_ABL & 1)) {
            if constexpr (FP8) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc0[mh][mt][nt] = g8_mma_fp8(bf0[nt][0], bf0[nt][1], a[mt][0], a[mt][1], acc0[mh][mt][nt]);
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            if (mt < nmt) acc0[mh][mt][nt] = g8_mma_bf16(bf0[nt][kb], a[mt][kb], acc0[mh][mt][nt]);
            }
        }
    };
    auto mma1_of = [&](int mh, const u32x4 (&a)[4][2], int nmt = 4) {
        if (!(G8_ABL & 1)) {
            if constexpr (FP8) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt)
                        acc1[mh][mt][nt] = g8_mma_fp8(bf1[nt][0], bf1[nt][1], a[mt][0], a[mt][1], acc1[mh][mt][nt]);
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT1; ++nt)
                            if (mt < nmt) acc1[mh][mt][nt] = g8_mma_bf16(bf1[nt][kb], a[mt][kb], acc1[mh][mt][nt]);
            }
        }
    };
    auto mma0 = [&](int mh) { __builtin_amdgcn_s_setprio(1); mma0_of(mh, af); __builtin_amdgcn_s_setprio(0); };
    auto mma1 = [&](int mh) { __builtin_amdgcn_s_setprio(1); mma1_of(mh, af); __builtin_amdgcn_s_setprio(0); };
#define G8_BAR() __builtin_amdgcn_s_barrier()
#define G8_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define G8_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#define G8_FENCE() __builtin_amdgcn_sched_barrier(0)

    // One K-tile T = four phases on the buffer at byte offset bo.  A part is restaged (with K-tile T+2) two phases after the phase
    // that read it -- one phase after for B part 0, whose reads are retired by the lgkmcnt(8) in front of phase 1's first barrier --
    // and read one phase after the counted wait that retires it (phase 4's, for the whole other buffer).
    // BN = 192 (16 / 8 / 8 / 16 MFMAs in four phases, and a half-step costs ~400 cycles whatever it holds): THREE phases of 16 MFMAs
    // -- (0, part 0) | (0, part 1) + (1, part 1) | (1, part 0) -- with the second A half in registers of its own.  Parts of K-tile
    // T+1 still missing go out in phase 1 (A h1, B part 1: their last reads were phase 2 of T-1), of T+2 in phases 2 (B part 0) and
    // 3 (A h0); the counted wait of phase 3 leaves those two (4 pieces) in flight.
    // STEADY (compile time): the K-tiles this one issues for belong to the SAME output tile (kt + 2 < nk), so every issue is
    // unconditional and the loop body has no branch and no tile bookkeeping
    auto ktile3 = [&](int bo, auto steadyc) {
        constexpr bool STEADY = decltype(steadyc)::value;
        if constexpr (NT1 == 1) {
            readB0(rdB0 + bo); G8_FENCE(); readA(rdA + bo, 0); G8_FENCE();
            if (STEADY || q_tile < t_end) { dma(qa1, KS - bo, 1); dma1(KS - bo); }
            if constexpr (STEADY) ++q_kt; else q_advance();
            G8_LGKM(8); G8_FENCE();
            G8_BAR(); G8_LGKM(0); G8_FENCE();
            mma0(0); G8_FENCE();
            G8_BAR();
            const bool live = STEADY || q_tile < t_end;
            readB1(rdB1 + bo); G8_FENCE(); readA_to(af1, rdA1 + bo, 1, MT1); G8_FENCE();
            if (live) dma(qb0, bo, 2);
            G8_BAR(); G8_LGKM(0); G8_FENCE();
            __builtin_amdgcn_s_setprio(1); mma1_of(0, af); mma1_of(1, af1, MT1); __builtin_amdgcn_s_setprio(0); G8_FENCE();
            G8_BAR();
            if (live) { dma(qa0, bo, 0); G8_VM(4); } else { G8_VM(0); }
            G8_FENCE();
            G8_BAR();
            __builtin_amdgcn_s_setprio(1); mma0_of(1, af1, MT1); __builtin_amdgcn_s_setprio(0); G8_FENCE();
            G8_BAR();
        }
    };
    auto ktile = [&](int bo, auto steadyc) {
        constexpr bool STEADY = decltype(steadyc)::value;
        if constexpr (NT1 == 1) { ktile3(bo, steadyc); return; }
        // phase 1: quadrant (0, part 0); completes K-tile T+1 (A h1 into the other buffer)
        readB0(rdB0 + bo); G8_FENCE(); readA(rdA + bo, 0); G8_FENCE();
        if (STEADY || q_tile < t_end) dma(qa1, KS - bo, 1);
        if constexpr (STEADY) ++q_kt; else q_advance();
        G8_LGKM(8); G8_FENCE();
        G8_BAR(); G8_LGKM(0); G8_FENCE();
        mma0(0); G8_FENCE();
        G8_BAR();
        // phase 2: quadrant (0, part 1); B part 0 of K-tile T+2
        const bool live = STEADY || q_tile < t_end;
        readB1(rdB1 + bo); G8_FENCE();
        if (live) dma(qb0, bo, 2);
        G8_BAR(); G8_LGKM(0); G8_FENCE();
        mma1(0); G8_FENCE();
        G8_BAR();
        // phase 3: quadrant (1, part 1); A h0 of K-tile T+2
        readA(rdA + bo, 1); G8_FENCE();
        if (live) dma(qa0, bo, 0);
        G8_BAR(); G8_LGKM(0); G8_FENCE();
        mma1(1); G8_FENCE();
        G8_BAR();
        // phase 4: quadrant (1, part 0); B part 1 of K-tile T+2; the counted wait retires the OTHER buffer (K-tile T+1)
        if (live) { dma1(bo); G8_VM(VMN); } else { G8_VM(0); }
        G8_FENCE();
        G8_BAR();
        mma0(1); G8_FENCE();
        G8_BAR();
    };

    // ---- prologue: K-tile 0 whole, K-tile 1 without its A h1
    {
        dma(qb0, 0, 2); dma(qa0, 0, 0); dma1(0); dma(qa1, 0, 1);
        q_advance();
        if constexpr (NT1 == 1) {
            if (q_tile < t_end) { dma(qb0, KS, 2); dma(qa0, KS, 0); G8_VM(4); } else { G8_VM(0); }
        } else {
            if (q_tile < t_end) { dma(qb0, KS, 2); dma(qa0, KS, 0); dma1(KS); G8_VM(VMN); } else { G8_VM(0); }
        }
        G8_FENCE();
        G8_BAR();
    }

    int bo = 0;
    for (int tile = t_begin; tile < t_end; tile += nxw) {
        if (wr == 1) G8_BAR();            // the second wave group runs one barrier behind the first
        zero_acc();
        int kt = 0;
        for (; kt + 2 < nk; ++kt) { ktile(bo, std::true_type{}); bo = KS - bo; }
        for (; kt < nk; ++kt) { ktile(bo, std::false_type{}); bo = KS - bo; }
        if (wr == 0) G8_BAR();            // ... and is waited for here: the eight epilogues run together
        // ---- epilogue, straight from the accumulators: lane (fr, g) holds row fr of each 16-row tile
        const int m0 = (tile / ga.tiles_n) * BM, n0 = (tile % ga.tiles_n) * BN;
        if (G8_ABL & 8) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) asm volatile("" :: "v"(acc0[a][c][d]));
#pragma unroll
                    for (int d = 0; d < NT1; ++d) asm volatile("" :: "v"(acc1[a][c][d]));
                }
        } else if (G8_ABL & 16) {
            // timing probe (results WRONG): the same bytes stored as whole 1 KB pieces (full 128-byte lines)
            bf16_t* base = ga.C + (int64_t)m0 * ga.ldc + (int64_t)n0 * 256 + wave * (16 * (2 + NT1) * 512) + lane * 8;
            int k = 0;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc0[mh][mt][0][r]; v[4 + r] = acc0[mh][mt][1][r]; }
                    st16(base + (k++) * 512, pack8(v));
                    if constexpr (NT1 == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = acc1[mh][mt][0][r]; v[4 + r] = acc1[mh][mt][1][r]; }
                        st16(base + (k++) * 512, pack8(v));
                    } else if ((mt & 1) == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = acc1[mh][mt][0][r]; v[4 + r] = acc1[mh][mt + 1][0][r]; }
                        st16(base + (k++) * 512, pack8(v));
                    }
                }
        } else {
            // ---- epilogue through the staging region: whole 128-byte lines leave the CU (straight from the accumulators a lane
            // owns 8 consecutive columns of a row, i.e. 64-byte row segments of lines that three waves share: measured 1.3 - 1.5x
            // the time of line-sized stores on the store phase).  Per pass (16 * PASS_MT rows per wave group):
            //   [second operand of the pass -> staging by LDS-DMA, whole lines]  barrier
            //   accumulators (+ bias, gelu' / row scale / residual in fp32) -> bf16 -> staging, in place      barrier
            //   staging -> global, 16 bytes per lane along rows (GELU runs here, on the stored pre-activation)   barrier
            // A staged row keeps its 16-byte chunks XORed with (row & 7) inside each group of eight: the 8-lane groups of a
            // ds_write_b128 then hit 8 different bank quads, and a DMA / row read still covers whole 128-byte lines.
            unsigned char* const stg = g8_smem + STG + wr * GRP;
            const unsigned stg_a = g8_lds_addr(stg);
            const unsigned tab_a = g8_lds_addr(g8_smem + TAB);
            const bf16_t* const in_src = has_dgelu ? ep.dgelu_of : has_mul ? ep.mul_by : (has_res ? ep.residual : nullptr);
            const int in_ld = (has_dgelu || has_mul) ? ga.ldc : ep.ldr;
            const bool rowgelu = has_gelu;                        // GELU (and what follows it) is applied in the row phase
            constexpr int CPR = BN / 8;                               // 16-byte chunks per staged row
            constexpr int NIT = 16 * PASS_MT * CPR / 256;             // chunks per lane and pass (6 / 4)
            const int nb = n0 + wc * WN;
            float bias0[8], bias1[4 * NT1];
            if (has_bias) {
                *reinterpret_cast<float4*>(bias0) = *reinterpret_cast<const float4*>(ep.bias + min(nb + g * 8, ga.N - 8));
                *reinterpret_cast<float4*>(bias0 + 4) = *reinterpret_cast<const float4*>(ep.bias + min(nb + g * 8, ga.N - 8) + 4);
#pragma unroll
                for (int q = 0; q < NT1; ++q)
                    *reinterpret_cast<float4*>(bias1 + 4 * q) = *reinterpret_cast<const float4*>(ep.bias + min(nb + 32 + g * 4 * NT1, ga.N - 4 * NT1) + 4 * q);
            }
            // MFMA-layout addresses inside a staged row: part 0 = 16 bytes at chunk wc*WN/8 + g, part 1 = 8 * NT1 bytes behind it
            const int c0 = (wc * WN) / 8 + g;
            const int c1 = NT1 == 2 ? (wc * WN + 32) / 8 + g : (wc * WN + 32) / 8 + (g >> 1);
            const int off0 = ((c0 ^ (fr & 7)) << 4);
            const int off1 = ((c1 ^ (fr & 7)) << 4) + (NT1 == 2 ? 0 : (g & 1) * 8);
            // the second operand of the whole tile (residual, or the stored pre-activation of gelu') in MFMA layout, by loads hipcc does
            // not see (it would guard every later LDS access and store with vmcnt(0)): one exposed latency per tile, under which the
            // stream's prefetch of the next tile keeps landing.  (Staging it through LDS by DMA, pass by pass, exposed one latency per
            // pass: gelu' at N = 1152 was 62 us against 60.6 for the 128 x 128-tile kernel.)
            typedef typename std::conditional<NT1 == 2, u32x4, u32x2>::type in1_t;
            // (the instantiations that spill registers take ordinary loads: a destination of an asm load may be spilled before it lands)
            constexpr bool ASM_IN = EF >= 0 && !(NT1 == 2 && EF == (G8_BIAS | G8_RS | G8_RES));
            u32x4 in0[2][4];
            in1_t in1[2][4];
            // the 8-bit derivative codes (gelu = 3 / mul_by8): 8 bytes per lane for the 8 columns of part 0, 8 / 4 bytes for part 1
            typedef typename std::conditional<NT1 == 2, u32x2, unsigned>::type j1_t;
            u32x2 j0[2][4];
            j1_t j1[2][4];
            if (has_mul8) {
                const unsigned char* jbase = ep.mul8 + min(nb + g * 8, ga.N - 8);
                const unsigned char* jbase1 = ep.mul8 + min(nb + 32 + g * 4 * NT1, ga.N - 4 * NT1);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int64_t roff = (int64_t)min(m0 + mh * 128 + wr * (mh ? R1 : 64) + mt * 16 + fr, ga.M - 1) * ga.ldc;
                        if constexpr (EF >= 0) {
                            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(j0[mh][mt]) : "v"(jbase + roff) : "memory");
                            if constexpr (NT1 == 2) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(j1[mh][mt]) : "v"(jbase1 + roff) : "memory");
                            else asm volatile("global_load_dword %0, %1, off" : "=v"(j1[mh][mt]) : "v"(jbase1 + roff) : "memory");
                        } else {
                            j0[mh][mt] = *reinterpret_cast<const u32x2*>(jbase + roff);
                            j1[mh][mt] = *reinterpret_cast<const j1_t*>(jbase1 + roff);
                        }
                    }
                G8_VM(0);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) { asm volatile("" : "+v"(j0[mh][mt])); asm volatile("" : "+v"(j1[mh][mt])); }
            }
            if (has_dgelu || has_mul || has_res) {
                const bf16_t* ibase = in_src + min(nb + g * 8, ga.N - 8);
                const bf16_t* ibase1 = in_src + min(nb + 32 + g * 4 * NT1, ga.N - 4 * NT1);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int64_t roff = (int64_t)min(m0 + mh * 128 + wr * (mh ? R1 : 64) + mt * 16 + fr, ga.M - 1) * in_ld;
                        if constexpr (ASM_IN) {
                            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(in0[mh][mt]) : "v"(ibase + roff) : "memory");
                            if constexpr (NT1 == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(in1[mh][mt]) : "v"(ibase1 + roff) : "memory");
                            else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(in1[mh][mt]) : "v"(ibase1 + roff) : "memory");
                        } else {
                            in0[mh][mt] = *reinterpret_cast<const u32x4*>(ibase + roff);
                            in1[mh][mt] = *reinterpret_cast<const in1_t*>(ibase1 + roff);
                        }
                    }
                G8_VM(0);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) { asm volatile("" : "+v"(in0[mh][mt])); asm volatile("" : "+v"(in1[mh][mt])); }
            }
            const float dq = FP8 ? ep.dq_a[0] * ep.dq_b[0] : 1.f;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int mh = pass / (NPASS / 2), mtb = (pass % (NPASS / 2)) * PASS_MT;    // this pass: tiles mt = mtb .. mtb + PASS_MT - 1 of half mh
                const int rbase = m0 + mh * 128 + wr * (mh ? R1 : 64) + mtb * 16;            // first matrix row of the wave group's pass
                const int vrows = mh ? max(0, min(16 * PASS_MT, R1 - mtb * 16)) : 16 * PASS_MT; // rows of this pass that exist (BM = 224: the second half has 3 tiles per wave)
#pragma unroll
                for (int t = 0; t < PASS_MT; ++t) {
                    const int mt = mtb + t;
                    if (mh == 1 && mt >= MT1) continue;
                    const int m = min(rbase + t * 16 + fr, ga.M - 1);
                    const unsigned rowp = stg_a + (t * 16 + fr) * RS;
                    const float rs = has_rs ? ep.row_scale[m / ep.rows_per_scale] : 1.f;
                    float v[8], w[4 * NT1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc0[mh][mt][0][r]; v[4 + r] = acc0[mh][mt][1][r]; }
#pragma unroll
                    for (int d = 0; d < NT1; ++d)
#pragma unroll
                        for (int r = 0; r < 4; ++r) w[4 * d + r] = acc1[mh][mt][d][r];
                    if constexpr (FP8) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] *= dq;
#pragma unroll
                        for (int q = 0; q < 4 * NT1; ++q) w[q] *= dq;
                    }
                    if (has_bias) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += bias0[q];
#pragma unroll
                        for (int q = 0; q < 4 * NT1; ++q) w[q] += bias1[q];
                    }
                    if (!rowgelu) {
                        const u32x4 i0 = in0[mh][mt];
                        u32x4 i1 = {0u, 0u, 0u, 0u};
                        if constexpr (NT1 == 2) i1 = in1[mh][mt]; else { i1[0] = in1[mh][mt][0]; i1[1] = in1[mh][mt][1]; }
                        if (has_dgelu) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[2 * q] *= gelu_erf_grad(bf_lo(i0[q])); v[2 * q + 1] *= gelu_erf_grad(bf_hi(i0[q])); }
#pragma unroll
                            for (int q = 0; q < 2 * NT1; ++q) { w[2 * q] *= gelu_erf_grad(bf_lo(i1[q])); w[2 * q + 1] *= gelu_erf_grad(bf_hi(i1[q])); }
                        }
                        if (has_mul) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[2 * q] *= bf_lo(i0[q]); v[2 * q + 1] *= bf_hi(i0[q]); }
#pragma unroll
                            for (int q = 0; q < 2 * NT1; ++q) { w[2 * q] *= bf_lo(i1[q]); w[2 * q + 1] *= bf_hi(i1[q]); }
                        }
                        if (has_mul8) {
                            float d[8];
                            gq_unpack4(j0[mh][mt][0], d); gq_unpack4(j0[mh][mt][1], d + 4);
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] *= d[q];
                            if constexpr (NT1 == 2) { gq_unpack4(j1[mh][mt][0], d); gq_unpack4(j1[mh][mt][1], d + 4); }
                            else gq_unpack4(j1[mh][mt], d);
#pragma unroll
                            for (int q = 0; q < 4 * NT1; ++q) w[q] *= d[q];
                        }
                        if (has_rs) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] *= rs;
#pragma unroll
                            for (int q = 0; q < 4 * NT1; ++q) w[q] *= rs;
                        }
                        if (has_res && !has_dgelu && !has_mul) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[2 * q] += bf_lo(i0[q]); v[2 * q + 1] += bf_hi(i0[q]); }
#pragma unroll
                            for (int q = 0; q < 2 * NT1; ++q) { w[2 * q] += bf_lo(i1[q]); w[2 * q + 1] += bf_hi(i1[q]); }
                        }
                    }
                    if (G8_ABL & 128) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) asm volatile("" :: "v"(v[q]));
#pragma unroll
                        for (int q = 0; q < 4 * NT1; ++q) asm volatile("" :: "v"(w[q]));
                    } else {
                    g8_lds_st16(rowp + off0, pack8(v));
                    if constexpr (NT1 == 2) g8_lds_st16(rowp + off1, pack8(w));
                    else { u32x2 t2; t2[0] = pack_bf2(w[0], w[1]); t2[1] = pack_bf2(w[2], w[3]); g8_lds_st8(rowp + off1, t2); }
                    }
                }
                G8_LGKM(0);
                G8_BAR();
                // rows out: chunk id -> (row, position p); position p of a row holds the row's 16-byte chunk p ^ (row & 7)
                u32x4 xs[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) xs[it] = (G8_ABL & 64) ? u32x4{0u, 0u, 0u, 0u} : g8_lds_ld16(stg_a + (it * 256 + wc * 64 + lane) * 16);
                G8_LGKM(0);
#pragma unroll
                for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(xs[it]));
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int id = it * 256 + wc * 64 + lane, row = id / CPR, p = id % CPR;
                    const int m = rbase + row, n = n0 + ((p ^ (row & 7)) << 3);
                    u32x4 x = xs[it];
                    u32x2 q8v = {0u, 0u}; bool q8ok = false;
                    u32x2 gqv = {0u, 0u}; bool gqok = false;          // gelu = 3: the 8 derivative codes of this chunk
                    if (m < ga.M && n < ga.N && row < vrows) {
                        if (GTAB && use_tab) {
                            // x = 8 bf16-rounded pre-activations: gelu(h) = h * Phi(h) and the derivative code from the table (see gemm_epi.h)
                            const float rs = has_rs ? ep.row_scale[m / ep.rows_per_scale] : 1.f;
                            unsigned e8[8];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                e8[2 * q] = g8_lds_ld4(tab_a + 4 * gq_tab_index<0>(x[q]));
                                e8[2 * q + 1] = g8_lds_ld4(tab_a + 4 * gq_tab_index<16>(x[q]));
                            }
                            G8_LGKM(0);
#pragma unroll
                            for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(e8[q]));
                            u32x4 o;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                o[q] = pack_bf2(bf_lo(x[q]) * __uint_as_float(e8[2 * q]) * rs, bf_hi(x[q]) * __uint_as_float(e8[2 * q + 1]) * rs);
                            // codes: byte 0 of each entry
                            gqv[0] = __builtin_amdgcn_perm(e8[1], e8[0], 0x0c0c0400u) | (__builtin_amdgcn_perm(e8[3], e8[2], 0x0c0c0400u) << 16);
                            gqv[1] = __builtin_amdgcn_perm(e8[5], e8[4], 0x0c0c0400u) | (__builtin_amdgcn_perm(e8[7], e8[6], 0x0c0c0400u) << 16);
                            gqok = true;
                            x = o;
                        } else
                        if (rowgelu) {
                            // x = the bf16-rounded pre-activation (bias included): the activation is applied to the rounded value; what is stored
                            // beside the output is x itself, or (gelu = 2) its activation derivative -- all the backward needs of it
                            float f[8];
                            unpack8(x, f);
                            const float rs = has_rs ? ep.row_scale[m / ep.rows_per_scale] : 1.f;
                            if (ep.gelu >= 2) {
                                float gp[8];
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    float c, e; gelu_parts(f[q], c, e);
                                    gp[q] = fmaf(f[q] * 0.39894228040143268f, e, c);
                                    f[q] = f[q] * c * rs;
                                }
                                if (ep.gelu == 3) { gqv[0] = gq_pack4(gp); gqv[1] = gq_pack4(gp + 4); gqok = true; }
                                else g8_store(ep.preact + (int64_t)m * ga.ldc + n, pack8(gp));
                            } else {
                                if (ep.preact) g8_store(ep.preact + (int64_t)m * ga.ldc + n, x);
#pragma unroll
                                for (int q = 0; q < 8; ++q) f[q] = gelu_erf(f[q]) * rs;
                            }
                            x = pack8(f);
                        }
                        if (rowgelu || (EF >= 0 && (EF & G8_Q8) != 0)) {
                            if constexpr (Q8C) {
                                if (ep.q8) {           // the operand of the fp8 GEMM that consumes this tensor, without a pass of its own
                                    float r8[8];
                                    unpack8(x, r8);
                                    u32x2 o;
#pragma unroll
                                    for (int h2 = 0; h2 < 2; ++h2) {
                                        float c4[4];
#pragma unroll
                                        for (int e = 0; e < 4; ++e) { qmx = fmaxf(qmx, fabsf(r8[4 * h2 + e])); c4[e] = fminf(fmaxf(r8[4 * h2 + e] * qsc, -448.f), 448.f); }
                                        int w = 0;
                                        w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[0], c4[1], w, false);
                                        w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[2], c4[3], w, true);
                                        o[h2] = (unsigned)w;
                                    }
                                    q8v = o; q8ok = true;
                                }
                            }
                        }
                        if (!(G8_ABL & 32)) g8_store(ga.C + (int64_t)m * ga.ldc + n, x);
                        else asm volatile("" :: "v"(x));
                    }
                    if (rowgelu && ep.gelu == 3) {
                        // 8 code bytes per lane: the lane with the even chunk of a pair takes its neighbour's 8 bytes (the XOR swizzle keeps chunk
                        // pairs on lane pairs) and stores 16 -- whole 128-byte lines per 16 lanes where the tile's column offset allows it
                        const unsigned n0lo = (unsigned)__shfl_xor((int)gqv[0], 1, 64), n0hi = (unsigned)__shfl_xor((int)gqv[1], 1, 64);
                        const bool nok = __shfl_xor((int)gqok, 1, 64) != 0;
                        if (gqok) {
                            unsigned char* gp8 = reinterpret_cast<unsigned char*>(ep.preact) + (int64_t)m * ga.ldc + n;
                            if (nok && !(ga.ldc & 15)) {
                                if (!((n >> 3) & 1)) { u32x4 o4; o4[0] = gqv[0]; o4[1] = gqv[1]; o4[2] = n0lo; o4[3] = n0hi; g8_store(reinterpret_cast<bf16_t*>(gp8), o4); }
                            } else *reinterpret_cast<u32x2*>(gp8) = gqv;
                        }
                    }
                    if constexpr (Q8C) {
                        if (ep.q8) {
                            // 8 bytes per lane would leave as half lines: the lane with the even chunk of a pair takes its neighbour's 8 bytes
                            // (the XOR swizzle keeps chunk pairs on lane pairs) and stores 16
                            const unsigned n0lo = (unsigned)__shfl_xor((int)q8v[0], 1, 64), n0hi = (unsigned)__shfl_xor((int)q8v[1], 1, 64);
                            const bool nok = __shfl_xor((int)q8ok, 1, 64) != 0;
                            if (q8ok) {
                                unsigned char* qp = ep.q8 + (int64_t)m * ga.ldc + n;
                                if (nok && !(ga.ldc & 15)) {
                                    if (!((n >> 3) & 1)) { u32x4 o4; o4[0] = q8v[0]; o4[1] = q8v[1]; o4[2] = n0lo; o4[3] = n0hi; st16(qp, o4); }
                                } else *reinterpret_cast<u32x2*>(qp) = q8v;
                            }
                        }
                    }
                }
                if (pass + 1 < NPASS) G8_BAR();        // (the row reads were retired above)
            }
            G8_BAR();                     // (the staging region is free again; keeps the two groups' barrier counts equal per tile)
        }
    }
    G8_VM(0);
    if constexpr (Q8C) {
        if (ep.q8 && ep.q8_amax) {
            qmx = group_max<64>(qmx);
            if (lane == 0 && __float_as_int(qmx) > *reinterpret_cast<volatile int*>(ep.q8_amax)) atomicMax(reinterpret_cast<int*>(ep.q8_amax), __float_as_int(qmx));
        }
    }
}
