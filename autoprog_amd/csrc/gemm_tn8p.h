// Weight gradients C[N1,N2] += alpha * A[M,N1]^T . B[M,N2] (fp32 accumulate; autograd of the Linear layers of models/volo.py:67,68,71,
// 156,158,180,182) in the structure of gemm8p.h: ONE 512-thread workgroup per CU, operands global -> LDS by LDS-DMA in whole
// 128-byte lines, counted vmcnt, raw barriers, two wave groups one barrier apart.
//
// Why: the 128 x 128 tile kernel (k_gemm_tn_grouped*) moves (128 + 128) columns x 2 B per token and tile -- 1.16 GB of L2 -> CU
// traffic for the four problems of a transformer block against 270 MB of operands -- and runs at that traffic's rate (120 us).
// A 192 x 192 tile (every Linear width of VOLO-D1 / D5 is a multiple of 192) moves 2/3 of it, and a ring of three 64-token K-tiles
// keeps up to 72 KB of loads in flight per CU.
//
// Work item = (problem, 192 x 192 output tile, token range); the host cuts the token axis so that a launch has about one item per
// CU and places all tiles of one (problem, token range) on one XCD (the placement table of tn_place).
//
// Geometry.  Wave (w1, w2) = (wave >> 2, wave & 3) owns output rows w1*96 .. +95 (6 MFMA tiles) and columns w2*48 .. +47 (3 tiles):
// 18 accumulator tiles, 36 MFMAs (16x16x32) per 64-token K-tile.  The reduction axis (tokens) is the SLOW axis of both operands,
// so fragments come from transposed LDS reads (ds_read_b64_tr_b16) of token-major tiles.
//
// LDS.  K-tile slot (48 KB) = A tile [64 tok][192] | B tile [64 tok][192]; an operand tile = 3 column blocks (64 columns = one
// 128-byte line per token) x 8 pieces of [8 tok][128 B] = one wave-instruction of the DMA.  Inside a piece, 16-byte position s of
// token row t holds the row's chunk s ^ f(t), f(t) = 2 * (((t >> 1) & 1) | (((t >> 3) & 1) << 1)): the four token rows that share a
// bank window in one half of a transposed read then sit in four different 32-byte slots (conflict free), and chunk pairs stay
// adjacent (f is even), so a fragment is base + constants.  The DMA writes lane-linear: the swizzle is on the SOURCE address.
//
// Schedule.  A K-tile is two parts (tokens 0-31 / 32-63 = one MFMA k-step each), a part is a phase:
//     tr-reads of part P (18)  |  DMA of part P+4 (3 pieces per wave: waves 0-3 the A pieces, waves 4-7 the B pieces)  |
//     s_waitcnt vmcnt(9)  -> part P+1 has landed, P+2 .. P+4 (72 KB) stay in flight  |  barrier | 18 MFMAs | barrier
// The ring holds six parts; part P+4 overwrites part P-2, read two phases earlier.
#pragma once
#include "common.h"
#include "gemm_epi.h"
#include <type_traits>

#ifndef T8_ABL
#define T8_ABL 0          // timing-only ablations (tools/abl_tn.sh): 1 no MFMA, 2 no transposed reads, 4 no DMA, 8 no atomics
#endif
#define T8_RING (3 * 49152)
#define T8_LDS_BYTES (T8_RING + 2 * 1536)

#define T8_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define T8_LDS(p) ((__attribute__((address_space(3))) void*)(p))

struct T8Item {            // one problem of a grouped launch
    const bf16_t* A; const bf16_t* B; float* C; float* colsum; const bf16_t* cs_weight;
    int lda, ldb, ldc, M, t2, ksteps, ksps;          // t2: column tiles; ksteps: 64-token K-tiles of the problem; ksps: K-tiles per work item
    float alpha, cs_scale;
    int shared_out;                                  // another problem of the launch adds to the same C / column sum: atomics even when unsplit
};

// The transposed reads are inline asm: behind a pending LDS-DMA hipcc guards every LDS read it can see with s_waitcnt vmcnt(0)
// (it did so in front of the first ds_read_b64_tr_b16 of every phase), which drains the ring.  lgkmcnt is counted by hand.
// (a + 512: token rows +4 of the same piece)
__device__ __forceinline__ bf16x8 t8_join(const u32x2& lo, const u32x2& hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// it: the problem; tile, split: which 192 x 192 tile and which token range
__device__ __forceinline__ void t8_item(const T8Item& it, int tile, int split, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w1 = wave >> 2, w2 = wave & 3;
    const int i1 = tile / it.t2, i2 = tile - i1 * it.t2;
    const int n1_0 = i1 * 192, n2_0 = i2 * 192;
    const int kt_begin = split * it.ksps;
    const int nkt = min(it.ksps, it.ksteps - kt_begin);
    if (nkt <= 0) return;
    const int nparts = 2 * nkt;

    // ---- DMA source: this wave's pieces are token rows 8 * (wave & 3) + rr of a part, all three column blocks of ONE operand
    const int rr = lane >> 3, pt = wave & 3;
    const int f_src = ((((rr >> 1) & 1) | ((pt & 1) << 1)) << 1);
    const int chunk = (lane & 7) ^ f_src;
    const bool opB = w1 != 0;
    const int64_t ld = opB ? it.ldb : it.lda;
    const bf16_t* src = (opB ? it.B + n2_0 : it.A + n1_0) + ((int64_t)kt_begin * 64 + pt * 8 + rr) * ld + chunk * 8;
    const int64_t part_stride = 32 * ld;
    unsigned char* const dst0 = smem + (opB ? 24576 : 0) + pt * 1024;             // + slot*49152 + kb*4096 + cb*8192
    // per-token weights of a fused, masked column sum: the waves that own a column sum bring their part's 32 weights into a corner
    const bool cs_wave = it.colsum != nullptr && i2 == 0 && w2 == 0;
    const bool cs_w = cs_wave && it.cs_weight != nullptr;
    unsigned char* const wcorner = smem + T8_RING + w1 * 1536;
    const unsigned smem_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned wcorner_a = smem_a + T8_RING + w1 * 1536;
    const bf16_t* wsrc = it.cs_weight ? it.cs_weight + (int64_t)kt_begin * 64 + min(2 * lane, 30) : nullptr;
    // the running source pointer of the part that goes out next; ring slot of part P = P % 6 (K-tile slot (P % 6) >> 1, k-step P & 1)
    const bf16_t* nsrc = src;
    const bf16_t* nwsrc = wsrc;
    int issued = 0;
    auto issue_to = [&](int slot, bool with_w) {
        if (!(T8_ABL & 4)) {
            unsigned char* d = dst0 + (slot >> 1) * 49152 + (slot & 1) * 4096;
            __builtin_amdgcn_global_load_lds(T8_GLB(nsrc), T8_LDS(d), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(T8_GLB(nsrc + 64), T8_LDS(d + 8192), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(T8_GLB(nsrc + 128), T8_LDS(d + 16384), 16, 0, 0);
            if (with_w) __builtin_amdgcn_global_load_lds(T8_GLB(nwsrc), T8_LDS(wcorner + slot * 256), 4, 0, 0);
        }
        nsrc += part_stride; nwsrc += 32;
        ++issued;
    };

    // ---- fragment read addresses: lane = 16 g + 4 q + p reads 8 bytes of token row 8 g + q (+4), columns 4 p .. 4 p + 3 of a 16-column tile
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int f_rd = ((((q >> 1) & 1) | ((g & 1) << 1)) << 1);
    const int lane_rd = g * 1024 + q * 128 + p * 8;
    // two address registers per fragment: K-tile slots 0 / 1 are immediates away from the first (ds offsets are 16 bits), slot 2 from the second
    unsigned offA[6], offB[3], offA2[6], offB2[3];
#pragma unroll
    for (int t = 0; t < 6; ++t) { const int T = w1 * 6 + t; offA[t] = smem_a + (T >> 2) * 8192 + ((((T & 3) * 2) ^ f_rd) << 4) + lane_rd; offA2[t] = offA[t] + 98304; }
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int T = w2 * 3 + t; offB[t] = smem_a + 24576 + (T >> 2) * 8192 + ((((T & 3) * 2) ^ f_rd) << 4) + lane_rd; offB2[t] = offB[t] + 98304; }

    f32x4 acc[6][3], csum[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        csum[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

#define T8_BAR() __builtin_amdgcn_s_barrier()
#define T8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define T8_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#define T8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define T8_READ(lo, hi, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(addr), "i"(off), "i"((off) + 512) : "memory")

    // one phase on ring slot S (compile time): transposed reads of the part | counted wait | barrier | 18 MFMAs with the DMA of the part
    // four ahead issued among them (a piece costs ~60 cycles of issue there, 100 - 185 in a load section) | barrier.
    // STEADY: at least four more parts follow (every issue and wait is unconditional: the loop body has no branch);
    // CS / CSW: this wave owns a column sum / a weighted one (decided once per wave, outside the loop).
    auto phase = [&](auto slotc, auto steadyc, auto csc, auto cswc, int P) {
        constexpr int S = decltype(slotc)::value;
        constexpr bool STEADY = decltype(steadyc)::value, CS = decltype(csc)::value, CSW = decltype(cswc)::value;
        constexpr int OFF = ((S >> 1) == 1 ? 49152 : 0) + (S & 1) * 4096;
        u32x2 alo[6], ahi[6], blo[3], bhi[3];
        if (T8_ABL & 2) {
#pragma unroll
            for (int t = 0; t < 3; ++t) { blo[t] = u32x2{offB[t], 1u}; bhi[t] = u32x2{2u, offB[t]}; }
#pragma unroll
            for (int t = 0; t < 6; ++t) { alo[t] = u32x2{offA[t], 3u}; ahi[t] = u32x2{4u, offA[t]}; }
        } else {
#pragma unroll
            for (int t = 0; t < 3; ++t) T8_READ(blo[t], bhi[t], (S >> 1) == 2 ? offB2[t] : offB[t], OFF);
#pragma unroll
            for (int t = 0; t < 6; ++t) T8_READ(alo[t], ahi[t], (S >> 1) == 2 ? offA2[t] : offA[t], OFF);
        }
        u32x4 wv = ones_u;
        if constexpr (CSW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wv) : "v"(wcorner_a + g * 16), "i"(S * 256) : "memory");
        T8_FENCE();
        // part P+1 has landed (P+2, P+3 stay in flight; P+4 goes out below)
        if constexpr (STEADY) { if constexpr (CSW) T8_VM(8); else T8_VM(6); }
        else { if (issued - (P + 1) >= 3) { if constexpr (CSW) T8_VM(8); else T8_VM(6); } else T8_VM(0); }
        T8_FENCE();
        T8_BAR(); T8_LGKM0();
#pragma unroll
        for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(blo[t]), "+v"(bhi[t]));
#pragma unroll
        for (int t = 0; t < 6; ++t) asm volatile("" : "+v"(alo[t]), "+v"(ahi[t]));
        asm volatile("" : "+v"(wv));
        T8_FENCE();
        bf16x8 af[6], bfr[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bfr[t] = t8_join(blo[t], bhi[t]);
#pragma unroll
        for (int t = 0; t < 6; ++t) af[t] = t8_join(alo[t], ahi[t]);
        __builtin_amdgcn_s_setprio(1);
        if (T8_ABL & 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t) asm volatile("" :: "v"(bfr[t]));
#pragma unroll
            for (int t = 0; t < 6; ++t) asm volatile("" :: "v"(af[t]));
        }
#pragma unroll
        for (int a = 0; a < ((T8_ABL & 1) ? 0 : 2); ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        T8_FENCE();
        if constexpr (STEADY) issue_to((S + 4) % 6, CSW);
        else { if (issued < nparts) issue_to((S + 4) % 6, CSW); }
        T8_FENCE();
#pragma unroll
        for (int a = 2; a < ((T8_ABL & 1) ? 0 : 6); ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        if constexpr (CS) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
            for (int a = 0; a < 6; ++a) csum[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], wf, csum[a], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        T8_FENCE();
        T8_BAR();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using Yes = std::true_type; using No = std::false_type;
    // six phases = three K-tiles per trip while at least four more parts follow the trip; the general form for the rest
    auto run = [&](auto csc, auto cswc) {
        int P = 0;
        for (; P + 10 <= nparts; P += 6) {
            phase(I0{}, Yes{}, csc, cswc, P); phase(I1{}, Yes{}, csc, cswc, P + 1); phase(I2{}, Yes{}, csc, cswc, P + 2);
            phase(I3{}, Yes{}, csc, cswc, P + 3); phase(I4{}, Yes{}, csc, cswc, P + 4); phase(I5{}, Yes{}, csc, cswc, P + 5);
        }
        for (; P < nparts; ++P) {               // (P is a multiple of 6 here: the slots stay compile-time)
            switch (P % 6) {
                case 0: phase(I0{}, No{}, csc, cswc, P); break;
                case 1: phase(I1{}, No{}, csc, cswc, P); break;
                case 2: phase(I2{}, No{}, csc, cswc, P); break;
                case 3: phase(I3{}, No{}, csc, cswc, P); break;
                case 4: phase(I4{}, No{}, csc, cswc, P); break;
                default: phase(I5{}, No{}, csc, cswc, P); break;
            }
        }
    };

    // prologue: parts 0 .. 3 out, part 0 landed
    for (int i = 0; i < 4 && i < nparts; ++i) issue_to(i, cs_w);
    if (nparts > 3) { if (cs_w) T8_VM(12); else T8_VM(9); } else T8_VM(0);
    T8_FENCE();
    T8_BAR();
    if (w1 == 1) T8_BAR();                    // the second wave group runs one barrier behind the first
    if (cs_w) run(Yes{}, Yes{}); else if (cs_wave) run(Yes{}, No{}); else run(No{}, No{});
    if (w1 == 0) T8_BAR();

    // ---- partial tile -> C (4 rows x 64 bytes per wave-instruction)
    const int fr = lane & 15;
    if (T8_ABL & 8) {
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) asm volatile("" :: "v"(acc[a][b]));
        return;
    }
    float* const crow = it.C + (int64_t)(n1_0 + w1 * 96 + 4 * g) * it.ldc + n2_0 + w2 * 48 + fr;
    const float sc = it.cs_weight ? it.cs_scale : 1.0f;
    if (it.ksps >= it.ksteps && !it.shared_out) {
        // the item is the problem's whole token axis: nobody else adds to this tile, so C += is a plain read-add-store (the chip adds
        // 1.3 TB/s of fp32 atomics against ~6 TB/s of stores, and a launch holds one 144 KB tile per CU whatever it is cut into)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float old[3][4];
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[b][r] = crow[(int64_t)(a * 16 + r) * it.ldc + b * 16];
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(fmaf(acc[a][b][r], it.alpha, old[b][r]), crow + (int64_t)(a * 16 + r) * it.ldc + b * 16);
        }
        if (cs_wave && fr == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) { float* q = it.colsum + n1_0 + w1 * 96 + a * 16 + 4 * g + r; *q = fmaf(csum[a][r], sc, *q); }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(crow + (int64_t)(a * 16 + r) * it.ldc + b * 16, acc[a][b][r] * it.alpha);
    if (cs_wave && fr == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(it.colsum + n1_0 + w1 * 96 + a * 16 + 4 * g + r, csum[a][r] * sc);
    }
}
