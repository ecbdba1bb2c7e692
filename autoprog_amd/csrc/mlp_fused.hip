// Fused MLP for gfx950: fc1 -> GELU -> fc2 (+ DropPath scale + residual) in ONE launch, and the same kernel for the two input-gradient
// products of the backward pass (models/volo.py:147-167 `Mlp.forward`, called at :143 / :233).
//
//   out[M, C] = epi2( sum_j  f_j( X[M, C] . Wa[64 j .. 64 j + 63, :]^T ) . Wb[:, 64 j .. 64 j + 63]^T )          Hd = 3 C hidden units, 64 per chunk j
//
//   forward  : X = LN2(x), Wa = fc1.weight [Hd, C], Wb = fc2.weight [C, Hd];  f: t = acc + b1, h = bf16(t), a = bf16(h Phi(h) rs1) and the
//              8-bit code of gelu'(h) from the 16-KB table (gemm_epi.h); a and the codes also leave for the backward pass;
//              epi2: out = bf16((acc + b2) rs2 + residual)
//   backward : X = dL/dy, Wa = fc2.weight^T copy [Hd, C], Wb = fc1.weight^T copy [C, Hd];  f: dh = bf16(acc g'(code) rs1), stored for the weight
//              gradients;  epi2: out = bf16(acc) = dL/d(LN2 x)
//
// STATUS (round 6): bit-identical to the two launches in both directions and ON by default (functional.FUSED_MLP / AP_FUSED_MLP=0 switches it off).  Version 2
// below takes 74 - 76 / 68 - 69 us between Python calls and 77 / 70 us inside the training step, against 51 + 35 / 43 + 27 for the launches it replaces (step
// 12.27 -> 12.09 ms on one box).  It was SLOWER in the step (109 / 76 us) until the weights were prefetched into L2 at kernel start: every workgroup streams all
// 1.77 MB of weights through a ring that covers an L2 hit, not a miss, and in the step the weights are cold (tools/check_mlp_cold.py reproduces that outside the
// step).  DESIGN.md section 3 "Round 6" and profiles/r06_mlp_fused.txt have the measurements; tools/mlp_lab.py / mlp_stamps.py are the instruments.
//
// The idea.  The two launches write the hidden activation (57.8 MB), drain, fill again and read it back; each runs its K loop at what one CU pulls from L2 and
// its store phase at what the chip writes, one after the other.  Here a workgroup keeps its 128 rows of X as MFMA fragments in REGISTERS for the whole kernel, the
// hidden chunk never leaves the registers either -- the accumulator layout of phase 1 (lane (fr, g) holds 8 consecutive hidden units of row fr) IS the operand
// layout of phase 2 -- and the only thing that streams through LDS is the weights: 2 * 3 C * C * 2 bytes per workgroup, in 8-KB pieces by LDS-DMA.
//
// Version 1 (this kernel, AP_MLP_FUSED_V=1; 111 / 81 us: abandoned for version 2 further down):
//   workgroup : 256 threads = 4 waves, ONE per SIMD (the kernel is a 512-register kernel: 96 registers of X fragments + 192 of output accumulators
//               at C = 384); wave w owns rows 32 w .. 32 w + 31 of the block in BOTH phases, so nothing is exchanged between waves -- the four
//               waves share only the weight stream
//   piece     : 64 weight rows x 64 k (bf16, 128-byte rows, 16-byte chunks XOR-swizzled by row & 7 on the DMA's SOURCE address and on the read
//               address).  Phase 1 of chunk j: C / 64 pieces of Wa rows 64 j .. (k tiles of X's width); phase 2: C / 64 pieces of Wb (64 output
//               columns each, k = the chunk's 64 hidden units).  A piece's rows are permuted (row 16 t + i  <->  unit 32 (t >> 1) + 8 (i >> 2)
//               + 4 (t & 1) + (i & 3)) so that a lane's results of two 16-row tiles are 8 CONSECUTIVE units / output columns
//   ring      : 5 slots of 3 pieces (24 KB); slot s is waited for (counted vmcnt + one barrier) in front of the LAST piece of slot s - 1 -- the
//               first fragments of slot s are read while that piece multiplies -- and the same barrier frees slot s - 2, into which the DMA of
//               slot s + 3 goes: three slots (72 KB) in flight.  vmcnt is ONE in-order counter for DMA, the stores of the hidden chunk and the
//               loads of the codes: the count in front of every barrier is computed at compile time from the chunk's event sequence (mf_nwait)
//   MFMA      : D = W fragment (A operand) x X fragment (B operand), v_mfma_f32_16x16x32_bf16, fp32 accumulation over ascending 32-deep K steps --
//               the K order and the rounding points of the unfused launches (gemm8p.h): results are bit-identical to them
//   LDS       : ring 120 KB | GELU table 16 KB | fc1 bias (Hd floats)
#include "common.h"
#include "gemm_epi.h"
#include <type_traits>
#include <cstdlib>

#define MF_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define MF_LDS(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef MF_EPI_PRIO
#define MF_EPI_PRIO 2
#endif
#ifndef MF_ABL
#define MF_ABL 0          // timing-only ablations (lab builds): 1 no MFMA, 2 no fragment reads, 4 no DMA, 8 no epilogue-1 arithmetic, 16 no hidden stores
#endif

struct MlpArgs {
    const bf16_t* X; int ldx;
    const bf16_t* Wa; int ldwa;        // [Hd, C]
    const bf16_t* Wb; int ldwb;        // [C, Hd]
    bf16_t* Out; int ldo;              // [M, C]
    bf16_t* Hout; int ldh;             // [M, Hd]: forward gelu(h) * rs1, backward dL/dh
    unsigned char* G;                  // [M, Hd] bytes (row stride ldh): the gelu' codes, written by the forward, read by the backward
    const float* bias1; const float* bias2;
    const float* rs1; const float* rs2; int rows_per_scale;
    const bf16_t* res; int ldr;
    const unsigned* gelu_tab;
    int M, Hd;
    // forward, optional (k_mlp_fused2<C, false, true>): X holds the rows BEFORE the LayerNorm in front of fc1; the producer waves normalise
    // them in registers, write them to LnOut (row stride ldlo; the fc1 weight gradient reads them) and the row statistics to LnMean / LnRstd
    bf16_t* LnOut; int ldlo; const float* LnG; const float* LnB; float ln_eps; float* LnMean; float* LnRstd;
};

constexpr int MF_BM = 128, MF_PIECE = 8192, MF_SLOT = 3 * MF_PIECE, MF_NSLOT = 5, MF_LA = 3, MF_DMA = 6;
constexpr int MF_RING = MF_NSLOT * MF_SLOT, MF_TAB = MF_RING, MF_BIAS = MF_TAB + 16384;
constexpr int MF_LDS_BYTES = MF_BIAS + 8192;

// vector-memory operations issued after the DMA of slot s = tc * SPC + tb and before the wait in front of barrier s (see "ring" above).
// Event sequence: prologue DMA(0 .. LA - 1); barrier 0 (wait, DMA(LA)); then per chunk: segment 0 | barrier 1 | segment 1 | ... | barrier SPC |
// segment SPC, a barrier s issuing DMA(s + LA) while that slot exists; L loads at the start of segment lseg, S stores in segment sseg.
constexpr int mf_nwait(int SPC, int NCH, int L, int S, int lseg, int sseg, int tc, int tb) {
    const int NSL = SPC * NCH, ts = tc * SPC + tb;
    int count = 0, mark = -1;
    for (int s = 0; s < MF_LA; ++s) if (s < NSL) { count += MF_DMA; if (s == ts) mark = count; }
    if (ts == 0) return count - mark;
    if (MF_LA < NSL) { count += MF_DMA; if (MF_LA == ts) mark = count; }
    for (int c = 0; c < NCH; ++c)
        for (int seg = 0; seg <= SPC; ++seg) {
            if (seg >= 1) {
                const int s = c * SPC + seg;
                if (s == ts) return count - mark;
                if (s + MF_LA < NSL) { count += MF_DMA; if (s + MF_LA == ts) mark = count; }
            }
            if (seg == lseg) count += L;
            if (seg == sseg) count += S;
        }
    return 0;
}

// barriers b in (after, SPC] of chunk tc that still issue a DMA (slot tc * SPC + b + LA exists)
constexpr int mf_issuing_after(int SPC, int NSL, int tc, int after) {
    int n = 0;
    for (int b = after + 1; b <= SPC; ++b) if (tc * SPC + b + MF_LA < NSL) ++n;
    return n;
}

template <int N> __device__ __forceinline__ void mf_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void mf_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
// ds_read_b128 with the constant part of the address in the instruction's offset field (one address register for all fragments of a K block)
template <int OFF> __device__ __forceinline__ u32x4 mf_lds_ld16(unsigned adr) {
    u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(adr), "n"(OFF) : "memory"); return v;
}
template <int I, int N, class F> __device__ __forceinline__ void mf_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); mf_for<I + 1, N>(f); }
}
template <int V> using mf_ic = std::integral_constant<int, V>;

template <int C, bool BWD>
__global__ void __launch_bounds__(256, 1) k_mlp_fused(MlpArgs a) {
    static_assert(C % 192 == 0, "C = 192 or 384");
    constexpr int PP = C / 64;                 // pieces per phase
    constexpr int NP = 2 * PP;                 // pieces per chunk
    constexpr int SPC = NP / 3;                // slots per chunk
    constexpr int NCH = 3 * C / 64;            // chunks (Hd = 3 C)
    constexpr int NSL = SPC * NCH;             // slots in all
    constexpr int KS = C / 32;                 // 32-deep K steps of phase 1
    constexpr int NT2 = C / 16;                // output tiles per 16-row tile
    constexpr int L_OPS = BWD ? 4 : 0, S_OPS = BWD ? 4 : 8;
    constexpr int LSEG = 1, SSEG = PP / 3;     // piece p is computed in segment (p + 1) / 3; the hidden chunk is stored behind piece PP - 1
    static_assert(NCH >= 5 && NP % 3 == 0 && PP % 3 == 0, "chunk kinds / slots never straddle the two phases");
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * MF_BM + wave * 32;

    // ---- prologue 1 (before any DMA is in flight): the fc1 bias into LDS
    if constexpr (!BWD) {
        float* lb = reinterpret_cast<float*>(mf_smem + MF_BIAS);
        for (int i = tid; i < a.Hd; i += 256) lb[i] = a.bias1 ? a.bias1[i] : 0.f;
        __syncthreads();
    }
    // ---- prologue 2: this wave's 32 rows of X as MFMA B-operand fragments (row fr of tile mt, k = 32 ks + 8 g ..), by loads hipcc does not see
    u32x4 xf[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const bf16_t* xp = a.X + (int64_t)(m0 + mt * 16 + fr) * a.ldx + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xf[mt][ks]) : "v"(xp + ks * 32) : "memory");
    }
    // the gelu' codes (backward): cj[.] this chunk's, cn[.] the next one's, requested a chunk ahead
    u32x2 cj[2][2], cn[2][2];
    const int64_t hrow[2] = {(int64_t)(m0 + fr) * a.ldh + g * 8, (int64_t)(m0 + 16 + fr) * a.ldh + g * 8};
    auto load_codes = [&](u32x2 (&c)[2][2], int j) {
        if constexpr (BWD) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(c[mt][b]) : "v"(a.G + hrow[mt] + j * 64 + b * 32) : "memory");
        }
    };
    load_codes(cj, 0);
    float rs1v[2], rs2v[2];                     // per-row DropPath factors of this lane's two rows
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = m0 + mt * 16 + fr;
        rs1v[mt] = a.rs1 ? a.rs1[row / a.rows_per_scale] : 1.f;
        rs2v[mt] = a.rs2 ? a.rs2[row / a.rows_per_scale] : 1.f;
    }

    // ---- DMA geometry: a slot = 3 pieces = 24 wave-instructions of 1 KB, six per wave; instruction id = 6 wave + i covers rows 8 sub .. 8 sub + 7
    // of piece pc.  Lane l writes LDS bytes [16 l, 16 l + 16) of the instruction's KB = chunk position l & 7 of row l >> 3, which holds the row's
    // 16-byte chunk (l & 7) ^ (row & 7).
    // The source address of instruction i = a wave-uniform part (the piece, the eight rows' position in it) + ONE per-lane byte offset per weight
    // matrix (the lane's row among the eight and its chunk): the row of id's rows 8 sub + sr is unit 32 (sub >> 2) + 16 (sub & 1) + 4 ((sub >> 1) & 1)
    // + 8 (sr >> 2) + (sr & 3) under the permutation above.
    const int sr = lane >> 3;
    const int srcchunk = ((lane & 7) ^ sr) * 8;
    const int lunit = (sr >> 2) * 8 + (sr & 3);
    const unsigned voffA = (unsigned)(lunit * a.ldwa + srcchunk) * 2u, voffB = (unsigned)(lunit * a.ldwb + srcchunk) * 2u;
    int dpc[MF_DMA], ddst[MF_DMA], dunit[MF_DMA];          // wave-uniform: piece of the slot, byte offset inside the slot, first unit of the eight rows
#pragma unroll
    for (int i = 0; i < MF_DMA; ++i) {
        const int id = wave * MF_DMA + i;
        dpc[i] = id >> 3;
        const int sub = id & 7;
        ddst[i] = dpc[i] * MF_PIECE + sub * 1024;
        dunit[i] = (sub >> 2) * 32 + (sub & 1) * 16 + ((sub >> 1) & 1) * 4;
    }
    // slot q of chunk j (q compile time, < SPC): phase 1 pieces kt = 3 q + pc, phase 2 pieces nb = 3 q - PP + pc
    auto dma_slot = [&](int j, auto qc, int ring) {
        constexpr int q = decltype(qc)::value;
        if (MF_ABL & 4) return;
        unsigned char* dst = mf_smem + ring;
        if constexpr (3 * q < PP) {
            const bf16_t* base = a.Wa + (int64_t)j * 64 * a.ldwa + 3 * q * 64;
#pragma unroll
            for (int i = 0; i < MF_DMA; ++i) {
                const unsigned char* src = reinterpret_cast<const unsigned char*>(base + dunit[i] * a.ldwa + dpc[i] * 64);
                __builtin_amdgcn_global_load_lds(MF_GLB(src + voffA), MF_LDS(dst + ddst[i]), 16, 0, 0);
            }
        } else {
            const bf16_t* base = a.Wb + (int64_t)(3 * q - PP) * 64 * a.ldwb + j * 64;
#pragma unroll
            for (int i = 0; i < MF_DMA; ++i) {
                const unsigned char* src = reinterpret_cast<const unsigned char*>(base + (int64_t)(dpc[i] * 64 + dunit[i]) * a.ldwb);
                __builtin_amdgcn_global_load_lds(MF_GLB(src + voffB), MF_LDS(dst + ddst[i]), 16, 0, 0);
            }
        }
    };
    auto ring_of = [](int r0, int k) { int x = r0 + k; x = x >= 2 * MF_NSLOT ? x - 2 * MF_NSLOT : x; return (x >= MF_NSLOT ? x - MF_NSLOT : x) * MF_SLOT; };
    // DMA of global slot SPC j + q for any q >= 0 (q >= SPC: a later chunk)
    auto dma_global = [&](int j, auto qc, int r0) {
        constexpr int q = decltype(qc)::value;
        dma_slot(j + q / SPC, mf_ic<q % SPC>{}, ring_of(r0, q));
    };

    // ---- fragment reads: tile nt (16 weight rows), K block kb of the piece at byte offset pb
    const int lane_off0 = fr * 128 + ((g ^ (fr & 7)) << 4), lane_off1 = fr * 128 + (((4 + g) ^ (fr & 7)) << 4);
    u32x4 wf[4][2], wn[4][2];
    if (MF_ABL & 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) { wf[nt][kb] = (u32x4){lane * 3u + 1u, 5u, 7u, 11u}; wn[nt][kb] = wf[nt][kb]; asm volatile("" : "+v"(wf[nt][kb]), "+v"(wn[nt][kb])); }
    }
    auto read_piece = [&](u32x4 (&w)[4][2], int pb) {
        if (MF_ABL & 2) return;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            w[nt][0] = ld16(mf_smem + pb + nt * 2048 + lane_off0);
            w[nt][1] = ld16(mf_smem + pb + nt * 2048 + lane_off1);
        }
    };

    f32x4 oacc[2][NT2], hacc[2][4];
    u32x4 af[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int t = 0; t < NT2; ++t) oacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) hacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) af[mt][b] = (u32x4){0u, 0u, 0u, 0u};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // X fragments, codes of chunk 0 (the row factors are hipcc's own loads)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[mt][ks]));
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(cj[mt][b]));
        asm volatile("" : "+v"(rs1v[mt]), "+v"(rs2v[mt]));
    }
    // the GELU table (16 KB: four 1-KB pieces per wave) in front of the weight stream; the first barrier's counted wait retires it
    if constexpr (!BWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(MF_GLB(a.gelu_tab + (wave * 4 + i) * 256 + lane * 4), MF_LDS(mf_smem + MF_TAB + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    // ---- ring prologue: slots 0 .. LA - 1, barrier 0, slot LA
    dma_global(0, mf_ic<0>{}, 0);
    dma_global(0, mf_ic<1>{}, 0);
    dma_global(0, mf_ic<2>{}, 0);
    mf_vmcnt<mf_nwait(SPC, NCH, L_OPS, S_OPS, LSEG, SSEG, 0, 0)>();
    __builtin_amdgcn_s_barrier();
    dma_global(0, mf_ic<3>{}, 0);
    read_piece(wf, 0);

    auto mma = [&](const u32x4& w, const u32x4& x, f32x4 c) -> f32x4 {
        if (MF_ABL & 1) { asm volatile("" :: "v"(w), "v"(x)); return c; }
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w), as_bf16x8(x), c, 0, 0, 0);
    };

    // ---- the hidden chunk behind phase 1: lane (fr, g) holds units 32 b + 8 g .. + 7 of row fr (tiles 2 b, 2 b + 1) -- the fragment of phase 2.
    // The LDS reads of the bias and of the table are inline asm with hand-counted waits: behind a pending LDS-DMA hipcc guards the reads it can
    // see (these; not the fragment reads) with s_waitcnt vmcnt(0), which would drain the ring once per chunk.
    const unsigned tab_a = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)(mf_smem + MF_TAB);
    const unsigned bias_a = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)(mf_smem + MF_BIAS) + g * 32;
    auto lds_ld16 = [](unsigned adr) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(adr) : "memory"); return v; };
    auto lds_ld4 = [](unsigned adr) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(adr) : "memory"); return v; };
    auto epi1 = [&](int j) {
        u32x4 bq[2][2];                         // fc1 bias of the lane's 2 x 8 units
        if constexpr (!BWD) {
#pragma unroll
            for (int b = 0; b < 2; ++b) { bq[b][0] = lds_ld16(bias_a + (j * 64 + b * 32) * 4); bq[b][1] = lds_ld16(bias_a + (j * 64 + b * 32) * 4 + 16); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < 2; ++b) { asm volatile("" : "+v"(bq[b][0])); asm volatile("" : "+v"(bq[b][1])); }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float rs = rs1v[mt];
            u32x4 hb[2];
            unsigned e[2][8];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = hacc[mt][2 * b][r]; v[4 + r] = hacc[mt][2 * b + 1][r]; }
                hacc[mt][2 * b] = (f32x4){0.f, 0.f, 0.f, 0.f}; hacc[mt][2 * b + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (!BWD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] += __uint_as_float(bq[b][0][q]); v[4 + q] += __uint_as_float(bq[b][1][q]); }
                    hb[b] = pack8(v);                    // the bf16-rounded pre-activation: gelu(h) = h Phi(h), and the code of gelu'(h), from the table
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        e[b][2 * q] = lds_ld4(tab_a + 4 * gq_tab_index<0>(hb[b][q]));
                        e[b][2 * q + 1] = lds_ld4(tab_a + 4 * gq_tab_index<16>(hb[b][q]));
                    }
                } else {
                    float d[8];
                    gq_unpack4(cj[mt][b][0], d); gq_unpack4(cj[mt][b][1], d + 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= d[q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= rs;
                    af[mt][b] = pack8(v);
                    if (!(MF_ABL & 16)) st16_nt(a.Hout + hrow[mt] + j * 64 + b * 32, af[mt][b]);
                }
            }
            if constexpr (!BWD) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(e[b][q]));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack_bf2(bf_lo(hb[b][q]) * __uint_as_float(e[b][2 * q]) * rs, bf_hi(hb[b][q]) * __uint_as_float(e[b][2 * q + 1]) * rs);
                    u32x2 gq;
                    gq[0] = __builtin_amdgcn_perm(e[b][1], e[b][0], 0x0c0c0400u) | (__builtin_amdgcn_perm(e[b][3], e[b][2], 0x0c0c0400u) << 16);
                    gq[1] = __builtin_amdgcn_perm(e[b][5], e[b][4], 0x0c0c0400u) | (__builtin_amdgcn_perm(e[b][7], e[b][6], 0x0c0c0400u) << 16);
                    af[mt][b] = o;
                    if (!(MF_ABL & 16)) {
                        st16_nt(a.Hout + hrow[mt] + j * 64 + b * 32, o);
                        *reinterpret_cast<u32x2*>(a.G + hrow[mt] + j * 64 + b * 32) = gq;
                    } else asm volatile("" :: "v"(gq));
                }
            }
        }
    };

    // ---- one chunk.  KIND: 0 / 1 the first two chunks, 2 steady, 3 / 4 the last two: they differ in the counted waits and in which barriers
    // still issue a DMA, all at compile time.  r0 = ring slot of the chunk's first slot.
    auto chunk = [&](int j, int r0, auto kindc) {
        constexpr int KIND = decltype(kindc)::value;
        constexpr int TC = KIND == 0 ? 0 : KIND == 1 ? 1 : KIND == 2 ? 2 : KIND == 3 ? NCH - 2 : NCH - 1;
        mf_for<0, NP>([&](auto pic) {
            constexpr int pi = decltype(pic)::value;
            if constexpr (pi % 3 == 2) {
                // barrier b of the chunk, in front of the last piece of slot b - 1: slot SPC j + b is complete (every wave's pieces), slot
                // SPC j + b - 2 has been read by everyone -> the DMA of slot SPC j + b + LA goes there
                constexpr int b = (pi + 1) / 3, s = TC * SPC + b;
                if constexpr (s < NSL) {
                    mf_vmcnt<mf_nwait(SPC, NCH, L_OPS, S_OPS, LSEG, SSEG, TC, b)>();
                    __builtin_amdgcn_s_barrier();
                    if constexpr (s + MF_LA < NSL) dma_global(j, mf_ic<b + MF_LA>{}, r0);
                }
                if constexpr (b == LSEG) load_codes(cn, min(j + 1, NCH - 1));
            }
            // the next piece's fragments (piece 0 of the next chunk behind the last one) while this piece multiplies
            constexpr bool has_next = (pi + 1 < NP) || KIND != 4;
            if constexpr (has_next) {
                const int pb = ring_of(r0, (pi + 1) / 3) + ((pi + 1) % 3) * MF_PIECE;
                if constexpr (pi & 1) read_piece(wf, pb); else read_piece(wn, pb);
            }
            u32x4 (&w)[4][2] = *((pi & 1) ? &wn : &wf);
            if constexpr (pi < PP) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) hacc[mt][nt] = mma(w[nt][kb], xf[mt][2 * pi + kb], hacc[mt][nt]);
                if constexpr (pi == PP - 1) epi1(j);
            } else {
                constexpr int nb = pi - PP;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) oacc[mt][nb * 4 + nt] = mma(w[nt][kb], af[mt][kb], oacc[mt][nb * 4 + nt]);
            }
        });
        if constexpr (BWD) {
            // the next chunk's codes were requested behind barrier LSEG's DMA by loads hipcc does not see: they must have LANDED before their
            // registers are read (or, once hipcc considers them dead, reused).  Younger than them: this chunk's stores and the DMAs of the
            // barriers behind LSEG that still issue one.
            constexpr int younger = S_OPS + MF_DMA * mf_issuing_after(SPC, NSL, TC, LSEG);
            mf_vmcnt<younger>();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int b = 0; b < 2; ++b) { asm volatile("" : "+v"(cn[mt][b])); cj[mt][b] = cn[mt][b]; }
        }
    };
    auto adv = [](int r0) { const int x = r0 + SPC; return x >= MF_NSLOT ? x - MF_NSLOT : x; };

    int r0 = 0;
    chunk(0, r0, mf_ic<0>{}); r0 = adv(r0);
    chunk(1, r0, mf_ic<1>{}); r0 = adv(r0);
    for (int j = 2; j < NCH - 2; ++j) { chunk(j, r0, mf_ic<2>{}); r0 = adv(r0); }
    chunk(NCH - 2, r0, mf_ic<3>{}); r0 = adv(r0);
    chunk(NCH - 1, r0, mf_ic<4>{});

    // ---- out: lane (fr, g) holds columns 32 tp + 8 g .. + 7 of row fr (tiles 2 tp, 2 tp + 1)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int64_t row = m0 + mt * 16 + fr;
#pragma unroll
        for (int tp = 0; tp < C / 32; ++tp) {
            const int col = tp * 32 + g * 8;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = oacc[mt][2 * tp][r]; v[4 + r] = oacc[mt][2 * tp + 1][r]; }
            if constexpr (!BWD) {
                if (a.bias2) {
                    const float4 b0 = *reinterpret_cast<const float4*>(a.bias2 + col), b1 = *reinterpret_cast<const float4*>(a.bias2 + col + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                const float rs = rs2v[mt];
                if (a.res) {
                    // (v rs + residual) as ONE fused multiply-add, as hipcc contracts it in the unfused launch's epilogue (gemm8p.h)
                    const u32x4 r8 = ld16(a.res + row * a.ldr + col);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[2 * q] = __builtin_fmaf(v[2 * q], rs, bf_lo(r8[q])); v[2 * q + 1] = __builtin_fmaf(v[2 * q + 1], rs, bf_hi(r8[q])); }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= rs;
                }
            }
            st16_nt(a.Out + row * a.ldo + col, pack8(v));
        }
    }
}

#ifdef MF_LAB
// lab builds: s_memtime stamps of workgroup MF_STAMP_WG, waves 0 (role A) and 4 (role B): [role][period][slot][0: behind the barrier, 1: slot's work done]
__device__ unsigned long long mf_stamps[2 * 32 * 4 * 2];
#ifndef MF_STAMP_WG
#define MF_STAMP_WG 3
#endif
#define MF_STAMP(role, t, q, k) do { if (blockIdx.x == MF_STAMP_WG && lane == 0 && (wave & 3) == 0) mf_stamps[(((role) * 32 + (t)) * 4 + (q)) * 2 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MF_STAMP(role, t, q, k) do {} while (0)
#endif

// =====================================================================================================================================
// Version 2: the same product with TWO waves per SIMD in two roles (round 6, second structure).  What version 1 measures (tools/mlp_lab.py,
// profiles/r06_mlp_fused.txt): one wave per SIMD cannot overlap anything -- its MFMAs, its fragment-read latencies, the GELU arithmetic and
// the DMA issue add up (78 us for a lone workgroup whose MFMAs take 26), and at 512 registers hipcc folds the fragment double buffer away.
//
//   workgroup : 512 threads = 8 waves; waves 0-3 are PRODUCERS (role A), waves 4-7 CONSUMERS (role B); waves w and w + 4 share a SIMD and the
//               32 rows 32 (w & 3) .. of the block
//   role A    : keeps its rows of X as fragments (96 registers), runs phase 1 of chunk t (the hidden chunk's 32 x 64 accumulators), then the
//               chunk's epilogue (bias, GELU + code from the table / times the stored codes, DropPath factor, bf16), stores the chunk for the
//               backward pass and hands it to its partner through 4 KB of LDS in phase-2 fragment order
//   role B    : keeps the 32 x C output accumulators (192 registers), runs phase 2 of chunk t - 1 while A is in the epilogue of chunk t --
//               the matrix pipe of the SIMD alternates between the two waves, the vector pipe works beside it -- and issues EVERY LDS-DMA of
//               the weight stream (it idles while A multiplies), so that only B counts vmcnt
//   period t  : four ring slots of three 8-KB pieces in consumption order: Wa(t) pieces 0-2 | Wa(t) 3-5 | Wb(t - 1) 0-2 | Wb(t - 1) 3-5; one
//               barrier in front of each slot (B: s_waitcnt vmcnt(12) first -- its six DMA instructions of each of the two younger slots), the
//               DMA of slot s + 3 behind it into the ring slot that slot s - 2 left; NCH + 1 periods (B has nothing in period 0, A nothing in
//               the last one; the slots without a chunk are loaded from the neighbouring chunk so that every count stays constant)
//   LDS       : ring 5 x 24 KB | hidden chunk hand-off 16 KB | GELU table 16 KB | fc1 bias
// (the table sits at LDS address 0: a table entry's byte offset is then the whole address of its ds_read_b32)
constexpr int M2_TAB = 0, M2_BIAS = 16384, M2_A = 21504, M2_RING = M2_A + 16384, M2_BIAS2 = M2_RING + MF_RING, M2_SCRATCH = M2_BIAS2 + 1536, M2_LDS_BYTES = M2_SCRATCH + 1024;
static_assert(M2_LDS_BYTES <= 163840 && M2_BIAS + 4608 <= M2_A, "LDS");

// LDS byte offsets of the table entries of the two bf16 values packed in w (gq_tab_index<0> / <16> times four), with packed 16-bit arithmetic:
// eight vector instructions per pair instead of twelve
__device__ __forceinline__ void mf_tab_addr2(unsigned w, unsigned& a0, unsigned& a1) {
    unsigned m = w & 0x7fff7fffu;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(m) : "v"(m), "s"((unsigned)((GQ_TAB_LO << 16) | GQ_TAB_LO)));      // max(mag - LO, 0)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "s"((unsigned)(((GQ_TAB_N - 1) << 16) | (GQ_TAB_N - 1))));
    const unsigned idx4 = (m << 2) | ((w >> 2) & 0x20002000u);        // sign (bit 15) -> entry 2048 + ., times four
    a0 = idx4 & 0xffffu;
    a1 = idx4 >> 16;
}

// RES: the forward launch adds a residual (a.res != nullptr).  A template parameter, not a branch: with both epilogues in one kernel hipcc spilled
// accumulator quadruples where the two paths join and reloaded them -- each reload behind an s_waitcnt vmcnt(0) -- in the middle of the pipelined epilogue.
template <int C, bool BWD, bool LN = false, bool RES = false>
__global__ void __launch_bounds__(512, 2) k_mlp_fused2(MlpArgs a) {
    static_assert(!(BWD && (LN || RES)), "the LayerNorm in front of fc1 and the residual belong to the forward launch");
    static_assert(C == 384, "version 2 is written for C = 384 (six pieces per phase, two slots per phase)");
    constexpr int PP = C / 64, NCH = 3 * C / 64, KS = C / 32, NT2 = C / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool roleB = wave >= 4;
    const int rg = wave & 3;
    const int fr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * MF_BM + rg * 32;
    const int64_t hrow[2] = {(int64_t)(m0 + fr) * a.ldh + g * 8, (int64_t)(m0 + 16 + fr) * a.ldh + g * 8};

    if constexpr (!BWD) {                        // the fc1 bias into LDS, before any DMA is in flight
        float* lb = reinterpret_cast<float*>(mf_smem + M2_BIAS);
        for (int i = tid; i < a.Hd; i += 512) lb[i] = a.bias1 ? a.bias1[i] : 0.f;
        float* lb2 = reinterpret_cast<float*>(mf_smem + M2_BIAS2);
        for (int i = tid; i < C; i += 512) lb2[i] = a.bias2 ? a.bias2[i] : 0.f;
        if constexpr (LN) {                      // gamma | beta in the hand-off buffer, which nobody writes before the second slot barrier
            // column c = 32 (4 i + 2 p + e) + 8 g + k  ->  float ((((g * KS / 4 + i) * 2 + p) * 8 + k) * 2 + e: the lanes of group g read their
            // 2 x 8 x 2 values of (i, p) as four 16-byte words {(k, e = 0), (k, 1), (k + 1, 0), (k + 1, 1)}
            float* lg = reinterpret_cast<float*>(mf_smem + M2_A);
            for (int c = tid; c < 2 * C; c += 512) {
                const int cc = c < C ? c : c - C, sidx = cc >> 5, gq = (cc >> 3) & 3, k = cc & 7;
                const int idx = ((((gq * (KS / 4) + (sidx >> 2)) * 2 + ((sidx >> 1) & 1)) * 8 + k) * 2 + (sidx & 1));
                lg[(c < C ? 0 : C) + idx] = c < C ? a.LnG[cc] : a.LnB[cc];
            }
        }
        __syncthreads();
    }
    const int lane_off0 = fr * 128 + ((g ^ (fr & 7)) << 4), lane_off1 = fr * 128 + (((4 + g) ^ (fr & 7)) << 4);
    auto ring_of = [](int s) { return M2_RING + (s % MF_NSLOT) * MF_SLOT; };
    auto lds_ld16 = [](unsigned adr) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(adr) : "memory"); return v; };
    auto lds_ld4 = [](unsigned adr) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(adr) : "memory"); return v; };
    auto lds_st16 = [](unsigned adr, const u32x4& v) { asm volatile("ds_write_b128 %0, %1" :: "v"(adr), "v"(v) : "memory"); };
    const unsigned smem_a = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)mf_smem;
    if (smem_a != 0) __builtin_trap();           // (no static LDS in this kernel: the dynamic block starts at 0, and mf_tab_addr2 relies on it)
    const unsigned abuf_a = smem_a + M2_A + rg * 4096 + lane * 16;        // + (mt * 2 + kb) * 1024
    auto mma = [&](const u32x4& w, const u32x4& x, f32x4 c) -> f32x4 {
        if (MF_ABL & 1) { asm volatile("" :: "v"(w), "v"(x)); return c; }
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w), as_bf16x8(x), c, 0, 0, 0);
    };
    // fragment (nt, kb) of piece pc of the ring slot at byte offset rb
    auto frag = [&](int rb, int pc, int nt, int kb) -> u32x4 {
        if (MF_ABL & 2) return (u32x4){lane * 3u + 1u, 5u, 7u, 11u};
        return ld16(mf_smem + rb + pc * MF_PIECE + nt * 2048 + (kb ? lane_off1 : lane_off0));
    };

    if (!roleB) {
        // ================================================================ role A: phase 1 + the chunk's epilogue
        // The weights into THIS XCD's L2, ahead of the ring.  In the training step every block's weights come from memory (tools/check_mlp.py re-reads
        // one set from L2: 69 us; with cold caches the same launch took 93, the two launches it replaces 84): the ring keeps three slots in flight, which
        // covers an L2 hit, not a miss.  One 4-byte LDS-DMA per 128-byte line is enough to bring the line in -- a wave instruction touches 64 lines = 8 KB --
        // so the workgroups that share an XCD (blockIdx % 8) split the 216 8-KB units of the two matrices between them: two or three instructions per
        // producer wave, into a 256-byte scratch nobody reads, never waited for.
        if (!(MF_ABL & 4)) {
            const int xi = blockIdx.x >> 3, nx = (gridDim.x + 7) >> 3;
            const int unitsA = a.Hd * a.ldwa * 2 / 8192, units = unitsA + C * a.ldwb * 2 / 8192;
            for (int u = xi + nx * wave; u < units; u += nx * 4) {
                const unsigned char* base = u < unitsA ? reinterpret_cast<const unsigned char*>(a.Wa) + (size_t)u * 8192
                                                       : reinterpret_cast<const unsigned char*>(a.Wb) + (size_t)(u - unitsA) * 8192;
                __builtin_amdgcn_global_load_lds(MF_GLB(base + lane * 128), MF_LDS(mf_smem + M2_SCRATCH + wave * 256), 4, 0, 0);
            }
        }
        u32x4 xf[2][KS];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bf16_t* xp = a.X + (int64_t)(m0 + mt * 16 + fr) * a.ldx + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xf[mt][ks]) : "v"(xp + ks * 32) : "memory");
        }
        u32x2 c0[2][2], c1[2][2];                 // the gelu' codes (backward) of the even / odd chunks, requested a period ahead: they come from memory
                                                  // (28.9 MB per launch, read once) and a slot or two do not cover that latency under load.  Two
                                                  // buffers that trade places (the period loop is unrolled by two), never a copy: see load_codes
        // ("+v", and no copy of a buffer anywhere: the destination IS the loop-carried register.  With "=v", or with `cj = cn` behind the wait, hipcc
        // loaded into a fresh register and copied it into the carried one right behind the load -- before the data had landed: an asm's result is
        // "there" for the compiler the moment the asm is issued)
        auto load_codes = [&](u32x2 (&c)[2][2], int j) {
            if constexpr (BWD) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(c[mt][b]) : "v"(a.G + hrow[mt] + j * 64 + b * 32) : "memory");
            }
        };
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int b = 0; b < 2; ++b) { c0[mt][b] = (u32x2){0u, 0u}; c1[mt][b] = (u32x2){0u, 0u}; }
        load_codes(c0, 0);
        float rs1v[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) rs1v[mt] = a.rs1 ? a.rs1[(m0 + mt * 16 + fr) / a.rows_per_scale] : 1.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[mt][ks]));
            asm volatile("" : "+v"(rs1v[mt]));
        }
        if constexpr (LN) {
            // LayerNorm of the wave's 32 rows, in the registers they were loaded into.  The arithmetic AND its order are k_ln_fwd_lp<3>'s (layernorm.hip:
            // 16 lanes per row, lane q owns the 8-element chunks q, q + 16, q + 32, sums them in that order and the 16 partial sums meet in an xor
            // butterfly 8, 4, 2, 1), so that the result is bit-identical to the separate launch: this lane (g = lane >> 4) holds chunks 4 i + j of its
            // row, i.e. ALL chunks of the four "lanes" q = g + 4 j -- their partial sums are formed here in the same order, the butterfly's first two
            // levels (q ^ 8, q ^ 4) pair chain j with j ^ 2 and j ^ 1 inside the lane, the last two (q ^ 2, q ^ 1) are lanes 32 and 16 apart.
            // Chains 0 / 1 and 2 / 3 run as the two halves of packed fp32 instructions (v_pk_add / v_pk_fma / v_pk_mul: the same IEEE results per half).
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const float invC = 1.0f / (float)C;
            const unsigned char* gb = mf_smem + M2_A + g * (KS / 4 * 2 * 64);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x2 v[KS / 4][2][8];                    // [i][chain pair][k]: .x = chunk 4 i + 2 p, .y = chunk 4 i + 2 p + 1
#pragma unroll
                for (int i = 0; i < KS / 4; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned w0 = xf[mt][4 * i + 2 * p][w], w1 = xf[mt][4 * i + 2 * p + 1][w];
                            v[i][p][2 * w] = (f32x2){bf_lo(w0), bf_lo(w1)};
                            v[i][p][2 * w + 1] = (f32x2){bf_hi(w0), bf_hi(w1)};
                        }
                f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KS / 4; ++i)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { s01 += v[i][0][k]; s23 += v[i][1][k]; }
                float sm = (s01.x + s23.x) + (s01.y + s23.y);
                sm += __shfl_xor(sm, 32, 64);
                sm += __shfl_xor(sm, 16, 64);
                const float mu = sm * invC;
                const f32x2 mu2 = {mu, mu};
                f32x2 q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KS / 4; ++i)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        v[i][0][k] -= mu2; v[i][1][k] -= mu2;
                        q01 = __builtin_elementwise_fma(v[i][0][k], v[i][0][k], q01);
                        q23 = __builtin_elementwise_fma(v[i][1][k], v[i][1][k], q23);
                    }
                float q = (q01.x + q23.x) + (q01.y + q23.y);
                q += __shfl_xor(q, 32, 64);
                q += __shfl_xor(q, 16, 64);
                const float rsd = rsqrtf(q * invC + a.ln_eps);
                if (g == 0) { a.LnMean[m0 + mt * 16 + fr] = mu; a.LnRstd[m0 + mt * 16 + fr] = rsd; }
                const f32x2 rs2v = {rsd, rsd};
#pragma unroll
                for (int i = 0; i < KS / 4; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        unsigned o0[4], o1[4];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            // gamma / beta of (k = 2 w, 2 w + 1) x (chunk 4 i + 2 p, + 1): one 16-byte read each (the staging loop interleaved them)
                            const f32x4 gg = *reinterpret_cast<const f32x4*>(gb + ((i * 2 + p) * 4 + w) * 16);
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(gb + C * 4 + ((i * 2 + p) * 4 + w) * 16);
                            const f32x2 e = __builtin_elementwise_fma(v[i][p][2 * w] * rs2v, (f32x2){gg[0], gg[1]}, (f32x2){bb[0], bb[1]});
                            const f32x2 o = __builtin_elementwise_fma(v[i][p][2 * w + 1] * rs2v, (f32x2){gg[2], gg[3]}, (f32x2){bb[2], bb[3]});
                            o0[w] = pack_bf2(e.x, o.x);
                            o1[w] = pack_bf2(e.y, o.y);
                        }
                        xf[mt][4 * i + 2 * p] = (u32x4){o0[0], o0[1], o0[2], o0[3]};
                        xf[mt][4 * i + 2 * p + 1] = (u32x4){o1[0], o1[1], o1[2], o1[3]};
                        st16_nt(a.LnOut + (int64_t)(m0 + mt * 16 + fr) * a.ldlo + (4 * i + 2 * p) * 32 + g * 8, xf[mt][4 * i + 2 * p]);
                        st16_nt(a.LnOut + (int64_t)(m0 + mt * 16 + fr) * a.ldlo + (4 * i + 2 * p + 1) * 32 + g * 8, xf[mt][4 * i + 2 * p + 1]);
                    }
            }
        }
        f32x4 hacc[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) hacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned bias_a = smem_a + M2_BIAS + g * 32;
        u32x4 bq[2][2];                          // fc1 bias of the lane's 2 x 8 units of the chunk: requested at the start of the chunk's first slot

        // the epilogue of m-tile mt of chunk j: -> the two phase-2 fragments o[kb] of the tile (rows fr, units 32 kb + 8 g ..)
        auto epi = [&](int j, int mt, u32x4 (&o)[2], const u32x2 (&cj)[2][2]) {
            const float rs = rs1v[mt];
            if constexpr (!BWD) {
                u32x4 hb[2];
                unsigned e[2][8];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = hacc[mt][2 * b][r]; v[4 + r] = hacc[mt][2 * b + 1][r]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] += __uint_as_float(bq[b][0][q]); v[4 + q] += __uint_as_float(bq[b][1][q]); }
                    hb[b] = pack8(v);            // the bf16-rounded pre-activation: gelu(h) = h Phi(h), and the code of gelu'(h), from the table
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned a0, a1;
                        mf_tab_addr2(hb[b][q], a0, a1);
                        e[b][2 * q] = lds_ld4(a0);
                        e[b][2 * q + 1] = lds_ld4(a1);
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // the first half's eight entries have landed once at most the second half's eight reads are outstanding
                    if (b == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(e[b][q]));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[b][q] = pack_bf2(bf_lo(hb[b][q]) * __uint_as_float(e[b][2 * q]) * rs, bf_hi(hb[b][q]) * __uint_as_float(e[b][2 * q + 1]) * rs);
                    u32x2 gq;
                    gq[0] = __builtin_amdgcn_perm(e[b][1], e[b][0], 0x0c0c0400u) | (__builtin_amdgcn_perm(e[b][3], e[b][2], 0x0c0c0400u) << 16);
                    gq[1] = __builtin_amdgcn_perm(e[b][5], e[b][4], 0x0c0c0400u) | (__builtin_amdgcn_perm(e[b][7], e[b][6], 0x0c0c0400u) << 16);
                    if (!(MF_ABL & 16)) {
                        st16_nt(a.Hout + hrow[mt] + j * 64 + b * 32, o[b]);
                        *reinterpret_cast<u32x2*>(a.G + hrow[mt] + j * 64 + b * 32) = gq;
                    } else asm volatile("" :: "v"(gq));
                }
            } else {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float v[8], d[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = hacc[mt][2 * b][r]; v[4 + r] = hacc[mt][2 * b + 1][r]; }
                    gq_unpack4(cj[mt][b][0], d); gq_unpack4(cj[mt][b][1], d + 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= d[q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= rs;
                    o[b] = pack8(v);
                    if (!(MF_ABL & 16)) st16_nt(a.Hout + hrow[mt] + j * 64 + b * 32, o[b]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) hacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        };

        auto period = [&](int t, u32x2 (&cj)[2][2], u32x2 (&cn)[2][2]) {
            mf_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int rb = ring_of(4 * t + q);
                __builtin_amdgcn_s_barrier();
                MF_STAMP(0, t, q, 0);
                if constexpr (q < 2) {
                    // phase 1, pieces 3 q .. 3 q + 2 of Wa(t): 24 fragments in (piece, kb, nt) order through a rolling buffer of eight.  The
                    // reads are inline asm with counted waits: left to itself hipcc sinks every read to its use (two fragments in flight, a
                    // full LDS latency per four MFMAs -- and this wave is the only one of its SIMD that multiplies during this slot)
                    u32x4 w[8];
                    const unsigned fa = smem_a + rb;
                    const unsigned fk[2] = {fa + (unsigned)lane_off0, fa + (unsigned)lane_off1};
                    if constexpr (!BWD && q == 0) {
                        // this chunk's bias: older than every fragment read below, so the first counted wait retires it
#pragma unroll
                        for (int b = 0; b < 2; ++b) { bq[b][0] = lds_ld16(bias_a + (t * 64 + b * 32) * 4); bq[b][1] = lds_ld16(bias_a + (t * 64 + b * 32) * 4 + 16); }
                        if (MF_ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (!(MF_ABL & 2)) {
                        mf_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; w[i] = mf_lds_ld16<(i >> 3) * MF_PIECE + (i & 3) * 2048>(fk[(i >> 2) & 1]); });
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) w[i] = (u32x4){lane * 3u + 1u, 5u, 7u, 11u};
                    }
                    mf_for<0, 24>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        constexpr int pc = i >> 3, kb = (i >> 2) & 1, nt = i & 3;
                        if (!(MF_ABL & 2)) { mf_lgkmcnt<(23 - i < 7 ? 23 - i : 7)>(); __builtin_amdgcn_sched_barrier(0); }
                        asm volatile("" : "+v"(w[i & 7]));
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) hacc[mt][nt] = mma(w[i & 7], xf[mt][2 * (3 * q + pc) + kb], hacc[mt][nt]);
                        if constexpr (i + 8 < 24) {
                            constexpr int n = i + 8;
                            if (!(MF_ABL & 2)) w[i & 7] = mf_lds_ld16<(n >> 3) * MF_PIECE + (n & 3) * 2048>(fk[(n >> 2) & 1]);
                        }
                    });
                    if constexpr (BWD && q == 0) {
                        // chunk t's codes were requested a period ago (younger: the last chunk's four stores): take them over, then request chunk
                        // t + 1's -- HERE, behind this slot's MFMAs: at the start of a slot the consumer waves issue their DMA, and these strided
                        // 8-byte loads in front of it cost them a third of the slot
                        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // (younger: the last chunk's four stores)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(cj[mt][b]));
                        load_codes(cn, min(t + 1, NCH - 1));
                    }
                } else {
                    // The chunk's epilogue, and the chunk into the hand-off buffer: B took chunk t - 1 out of it while this wave multiplied
                    // (second slot of this period).  Forward: tile 0 in this period's third slot, tile 1 in the fourth (the GELU is as long as
                    // B's MFMAs of a slot); backward: both tiles in the third (a fraction of B's slot), nothing in the fourth.
                    constexpr bool both = BWD;
                    if constexpr (q == 2 || !both) {
                        if constexpr (!BWD) __builtin_amdgcn_s_setprio(MF_EPI_PRIO);   // the GELU's vector instructions in front of the partner's MFMAs in the SIMD's issue arbitration
                        u32x4 o[2];
                        if constexpr (q == 2) {
                            epi(t, 0, o, cj);
                            lds_st16(abuf_a, o[0]); lds_st16(abuf_a + 1024, o[1]);
                        }
                        if constexpr (q == 3 || both) {
                            epi(t, 1, o, cj);
                            lds_st16(abuf_a + 2048, o[0]); lds_st16(abuf_a + 3072, o[1]);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_setprio(0);
                    }
                }
                MF_STAMP(0, t, q, 1);
            });
        };
        static_assert(NCH % 2 == 0, "the period loop is unrolled by two");
        for (int t = 0; t < NCH; t += 2) { period(t, c0, c1); period(t + 1, c1, c0); }
        for (int q = 0; q < 4; ++q) { __builtin_amdgcn_s_barrier(); MF_STAMP(0, NCH, q, 0); MF_STAMP(0, NCH, q, 1); }      // the last period is B's alone
        return;
    }

    // ==================================================================== role B: the weight stream and phase 2
    const int wb = wave - 4;
    const int sr = lane >> 3;
    const int srcchunk = ((lane & 7) ^ sr) * 8;
    const int lunit = (sr >> 2) * 8 + (sr & 3);
    const unsigned voffA = (unsigned)(lunit * a.ldwa + srcchunk) * 2u, voffB = (unsigned)(lunit * a.ldwb + srcchunk) * 2u;
    int dpc[MF_DMA], ddst[MF_DMA], dunit[MF_DMA];
#pragma unroll
    for (int i = 0; i < MF_DMA; ++i) {
        const int id = wb * MF_DMA + i;
        dpc[i] = id >> 3;
        const int sub = id & 7;
        ddst[i] = dpc[i] * MF_PIECE + sub * 1024;
        dunit[i] = (sub >> 2) * 32 + (sub & 1) * 16 + ((sub >> 1) & 1) * 4;
    }
    // slot q of period t: q < 2 pieces 3 q .. of Wa(t); q >= 2 pieces 3 (q - 2) .. of Wb(t - 1); chunks outside [0, NCH) are taken from the nearest one.
    // buffer_load ... lds through one descriptor per weight matrix: the instruction's address is a scalar offset (piece, rows) + this lane's
    // constant 32-bit offset -- no per-instruction vector arithmetic, half the address bits of the global_load form
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wa), 0, a.Hd * a.ldwa * 2, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Wb), 0, C * a.ldwb * 2, 0x00020000);
    auto dma_slot = [&](int s) {
        if (MF_ABL & 4) return;
        const int t = s >> 2, q = s & 3;
        unsigned char* dst = mf_smem + ring_of(s);
        if (q < 2) {
            const int j = min(t, NCH - 1);
            const int base = (j * 64 * a.ldwa + 3 * q * 64) * 2;
#pragma unroll
            for (int i = 0; i < MF_DMA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, MF_LDS(dst + ddst[i]), 16, voffA, base + (dunit[i] * a.ldwa + dpc[i] * 64) * 2, 0, 0);
        } else {
            const int j = max(t - 1, 0);
            const int base = (3 * (q - 2) * 64 * a.ldwb + j * 64) * 2;
#pragma unroll
            for (int i = 0; i < MF_DMA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, MF_LDS(dst + ddst[i]), 16, voffB, base + (dpc[i] * 64 + dunit[i]) * a.ldwb * 2, 0, 0);
        }
    };
    float rs2v[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) rs2v[mt] = a.rs2 ? a.rs2[(m0 + mt * 16 + fr) / a.rows_per_scale] : 1.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+v"(rs2v[mt]));
    if constexpr (!BWD) {                        // the GELU table in front of the weight stream (16 one-KB pieces, four per B wave)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(MF_GLB(a.gelu_tab + (wb * 4 + i) * 256 + lane * 4), MF_LDS(mf_smem + M2_TAB + (wb * 4 + i) * 1024), 16, 0, 0);
    }
    dma_slot(0); dma_slot(1); dma_slot(2);
    f32x4 oacc[2][NT2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < NT2; ++t) oacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 af[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) af[mt][kb] = (u32x4){0u, 0u, 0u, 0u};
    constexpr int NSLOTS = 4 * (NCH + 1);
    for (int t = 0; t <= NCH; ++t) {
        mf_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int s = 4 * t + q;
            const int rb = ring_of(s);
            // DMA schedule: slots s + 3 AND s + 4 go out behind the barrier of the period's first slot (the two ring slots the last period's
            // phase 2 left; this wave idles while A multiplies), s + 3 behind the second and the third, none behind the fourth: the wave issues
            // six instead of twelve DMA instructions in front of its own MFMAs.  In flight behind the wait for slot s: two younger slots at
            // the period's first barrier, three at the others (all of them landed in the last two periods, where nothing is issued any more).
            if (t == NCH) mf_vmcnt<0>();
            else if constexpr (q == 0) mf_vmcnt<2 * MF_DMA>();
            else mf_vmcnt<3 * MF_DMA>();
            __builtin_amdgcn_s_barrier();
            MF_STAMP(1, t, q, 0);
            if constexpr (q == 0) { if (s + 3 < NSLOTS) dma_slot(s + 3); if (s + 4 < NSLOTS) dma_slot(s + 4); }
            if constexpr (q == 1) {
                if (s + 4 < NSLOTS) dma_slot(s + 4);
                if (t >= 1) {
                    // chunk t - 1 out of the hand-off buffer, while A multiplies (A wrote it in the last period's third and fourth slots and
                    // writes the next chunk from this period's third slot on: barriers on both sides)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) af[mt][kb] = lds_ld16(abuf_a + (mt * 2 + kb) * 1024);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+v"(af[mt][kb]));
                    // (measured and dropped: B also STORING the chunk from these fragments, to take four store instructions per period off A --
                    // this slot became B's longest, 1200 - 2100 cycles against A's 1050: forward 77.4 -> 80.2 us, backward 66.4 -> 67.3)
                }
            }
            if constexpr (q == 2) { if (t < 1 && s + 4 < NSLOTS) dma_slot(s + 4); }          // (period 0 has no phase 2; otherwise see below)
            if constexpr (q >= 2) {
                if (t >= 1) {
                    // phase 2, output pieces 3 (q - 2) .. + 2: 24 fragments in (piece, kb, nt) order through a rolling buffer of six (asm reads
                    // with counted waits, as in role A)
                    constexpr int RB = 6;
                    u32x4 w[RB];
                    const unsigned fa = smem_a + rb;
                    const unsigned fk[2] = {fa + (unsigned)lane_off0, fa + (unsigned)lane_off1};
                    if (!(MF_ABL & 2)) {
                        mf_for<0, RB>([&](auto ic) { constexpr int i = decltype(ic)::value; w[i] = mf_lds_ld16<(i >> 3) * MF_PIECE + (i & 3) * 2048>(fk[(i >> 2) & 1]); });
                        // (the slot's one DMA batch goes out HERE, under the latency of the first fragment reads)
                        if constexpr (q == 2) { if (s + 4 < NSLOTS) dma_slot(s + 4); }
                    } else {
#pragma unroll
                        for (int i = 0; i < RB; ++i) w[i] = (u32x4){lane * 3u + 1u, 5u, 7u, 11u};
                        if constexpr (q == 2) { if (s + 4 < NSLOTS) dma_slot(s + 4); }
                    }
                    constexpr int nb0 = 3 * (q - 2);
                    mf_for<0, 24>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        constexpr int pc = i >> 3, kb = (i >> 2) & 1, nt = i & 3;
                        if (!(MF_ABL & 2)) { mf_lgkmcnt<(23 - i < RB - 1 ? 23 - i : RB - 1)>(); __builtin_amdgcn_sched_barrier(0); }
                        asm volatile("" : "+v"(w[i % RB]));
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) oacc[mt][(nb0 + pc) * 4 + nt] = mma(w[i % RB], af[mt][kb], oacc[mt][(nb0 + pc) * 4 + nt]);
                        if constexpr (i + RB < 24) {
                            constexpr int n = i + RB;
                            if (!(MF_ABL & 2)) w[i % RB] = mf_lds_ld16<(n >> 3) * MF_PIECE + (n & 3) * 2048>(fk[(n >> 2) & 1]);
                        }
                    });
                }
            }
            MF_STAMP(1, t, q, 1);
        });
    }
    // ---- out: lane (fr, g) holds columns 32 tp + 8 g .. + 7 of row fr (tiles 2 tp, 2 tp + 1).  Forward with a residual: its 24 chunks per lane through a
    // three-deep pipeline of loads hipcc does not see, with counted waits -- written naively (a load, its use and the store per chunk, under the run-time
    // `if (a.res)`) every chunk paid a memory latency: 26 000 cycles, a fifth of the launch (tools/mlp_stamps.py)
    auto out_chunk = [&](int mt, int tp, const u32x4* r8, const u32x4* bb = nullptr) {
        const int64_t row = m0 + mt * 16 + fr;
        const int col = tp * 32 + g * 8;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = oacc[mt][2 * tp][r]; v[4 + r] = oacc[mt][2 * tp + 1][r]; }
        if constexpr (!BWD) {
            if (bb) {                                   // (fc2's bias from LDS, zeros when there is none: adding +0 changes no bit of a finite sum)
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += __uint_as_float(bb[0][q]); v[4 + q] += __uint_as_float(bb[1][q]); }
            } else if (a.bias2) {
                const float4 b0 = *reinterpret_cast<const float4*>(a.bias2 + col), b1 = *reinterpret_cast<const float4*>(a.bias2 + col + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            const float rs = rs2v[mt];
            if (r8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[2 * q] = __builtin_fmaf(v[2 * q], rs, bf_lo((*r8)[q])); v[2 * q + 1] = __builtin_fmaf(v[2 * q + 1], rs, bf_hi((*r8)[q])); }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= rs;
            }
        }
        st16_nt(a.Out + row * a.ldo + col, pack8(v));
    };
    constexpr int NCK = 2 * (C / 32);
    constexpr bool piped = !BWD && RES;
    if constexpr (piped) {
        {
            mf_vmcnt<0>();                         // (what is left of this wave's DMA: nothing is read from the ring any more, but the counts below start at zero)
            u32x4 rb[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            auto rload = [&](u32x4& d, int c) {
                const bf16_t* rp = a.res + (int64_t)(m0 + (c / (C / 32)) * 16 + fr) * a.ldr + (c % (C / 32)) * 32 + g * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(d) : "v"(rp) : "memory");
            };
            rload(rb[0], 0); rload(rb[1], 1); rload(rb[2], 2);
            // fc2's bias of a chunk's eight columns out of LDS, one chunk ahead (inline asm: a bias LOAD from memory inside this loop would
            // make hipcc wait for every load in flight)
            const unsigned b2a = smem_a + M2_BIAS2 + g * 32;
            u32x4 bb[2][2];
            bb[0][0] = lds_ld16(b2a); bb[0][1] = lds_ld16(b2a + 16);
            mf_for<0, NCK>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c + 1 < NCK) {
                    bb[(c + 1) & 1][0] = lds_ld16(b2a + ((c + 1) % (C / 32)) * 128); bb[(c + 1) & 1][1] = lds_ld16(b2a + ((c + 1) % (C / 32)) * 128 + 16);
                }
                // younger than the load of chunk c: the loads of c + 1, c + 2 (while they exist) and the stores of c - 2, c - 1 (from chunk 2 on)
                constexpr int younger = (c + 1 < NCK) + (c + 2 < NCK) + (c >= 1) + (c >= 2);
                mf_vmcnt<younger>();
                if constexpr (c + 1 < NCK) mf_lgkmcnt<2>(); else mf_lgkmcnt<0>();       // this chunk's bias (the next one's two reads may be in flight)
                asm volatile("" : "+v"(rb[c % 3]), "+v"(bb[c & 1][0]), "+v"(bb[c & 1][1]));
                const u32x4 r8 = rb[c % 3];
                const u32x4 b8[2] = {bb[c & 1][0], bb[c & 1][1]};
                out_chunk(c / (C / 32), c % (C / 32), &r8, b8);
                if constexpr (c + 3 < NCK) rload(rb[c % 3], c + 3);
            });
        }
    }
    if constexpr (!piped) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int tp = 0; tp < C / 32; ++tp) out_chunk(mt, tp, nullptr);
    }
    MF_STAMP(1, NCH + 1, 0, 0);
}

#ifdef MF_LAB
extern "C" int mf_lab_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mf_stamps), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

#ifdef MF_LAB
// lab builds (tools/mlp_lab.py: this file alone as a shared object, one per MF_ABL value): a table of their own
__global__ void k_mf_lab_table(unsigned* tab) { const unsigned idx = blockIdx.x * 256 + threadIdx.x; if (idx < 2 * GQ_TAB_N) tab[idx] = gq_tab_entry(idx); }
const unsigned* g8_gelu_table_ptr(hipStream_t st) {
    static unsigned* t = nullptr;
    if (!t) {
        if (hipMalloc(&t, 2 * GQ_TAB_N * sizeof(unsigned)) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(k_mf_lab_table, dim3(2 * GQ_TAB_N / 256), dim3(256), 0, st, t);
        (void)hipStreamSynchronize(st);
    }
    return t;
}
#endif

template <int C, bool BWD, bool LN = false, bool RES = false>
static int mf_launch(const MlpArgs& a, hipStream_t st) {
    static int version = 0;                  // AP_MLP_FUSED_V = 1: the one-wave-per-SIMD kernel; default 2: producer / consumer waves
    if (!version) { const char* e = getenv("AP_MLP_FUSED_V"); version = (e && e[0] == '1') ? 1 : 2; }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_mlp_fused<C, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)k_mlp_fused2<C, BWD, LN, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, M2_LDS_BYTES);
        attr = true; (void)hipGetLastError();
    }
    if (version == 1 && LN) return AP_ERR_UNSUPPORTED;
    if (version == 1) hipLaunchKernelGGL((k_mlp_fused<C, BWD>), dim3(a.M / MF_BM), dim3(256), MF_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_mlp_fused2<C, BWD, LN, RES>), dim3(a.M / MF_BM), dim3(512), M2_LDS_BYTES, st, a);
    return ap_check_launch();
}

extern "C" {

int ap_mlp_fused(const ap_mlp_fused_args* p, ap_stream_t stream) {
    if (!p || !p->wa || !p->wb || !p->out || !p->hidden_out || !p->codes) return AP_ERR_NULL;
    const bool ln = p->ln_in != nullptr;
    if (!ln && !p->x) return AP_ERR_NULL;
    if (ln && (!p->ln_out || !p->ln_gamma || !p->ln_beta || !p->ln_mean || !p->ln_rstd)) return AP_ERR_NULL;
    if (ln && (p->backward || (p->ld_ln & 7) || p->ld_ln < p->c || (p->ld_lno & 7) || p->ld_lno < p->c)) return AP_ERR_SHAPE;
    const int C = p->c, Hd = p->hidden, M = p->m;
    if (M <= 0 || C <= 0 || Hd <= 0) return AP_ERR_SHAPE;
    if (!ln && ((p->ldx & 7) || p->ldx < C)) return AP_ERR_SHAPE;
    if ((p->ldwa & 7) || (p->ldwb & 7) || (p->ldo & 7) || (p->ldh & 7) || p->ldwa < C || p->ldwb < Hd || p->ldo < C || p->ldh < Hd)
        return AP_ERR_SHAPE;
    if (p->residual && ((p->ldr & 7) || p->ldr < C)) return AP_ERR_SHAPE;
    if (p->backward && (p->bias1 || p->bias2 || p->residual || p->row_scale_out)) return AP_ERR_SHAPE;
    // what the kernel is built for: the MLPs of VOLO-D1's transformer stages (C = 384, hidden 3 C), whole 128-row blocks
    if (C != 384 || Hd != 3 * C || (M % MF_BM)) return AP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    MlpArgs a;
    a.X = p->x; a.ldx = p->ldx; a.Wa = p->wa; a.ldwa = p->ldwa; a.Wb = p->wb; a.ldwb = p->ldwb; a.Out = p->out; a.ldo = p->ldo;
    a.Hout = p->hidden_out; a.ldh = p->ldh; a.G = p->codes; a.bias1 = p->bias1; a.bias2 = p->bias2;
    a.rs1 = p->row_scale_hidden; a.rs2 = p->row_scale_out; a.rows_per_scale = p->rows_per_scale > 0 ? p->rows_per_scale : 1;
    a.res = p->residual; a.ldr = p->ldr; a.gelu_tab = nullptr; a.M = M; a.Hd = Hd;
    a.LnOut = nullptr; a.ldlo = 0; a.LnG = a.LnB = nullptr; a.ln_eps = 0.f; a.LnMean = a.LnRstd = nullptr;
    if (ln) {
        a.X = p->ln_in; a.ldx = p->ld_ln; a.LnOut = p->ln_out; a.ldlo = p->ld_lno; a.LnG = p->ln_gamma; a.LnB = p->ln_beta; a.ln_eps = p->ln_eps;
        a.LnMean = p->ln_mean; a.LnRstd = p->ln_rstd;
    }
    (void)hipGetLastError();
    if (!p->backward) {
        a.gelu_tab = g8_gelu_table_ptr(st);
        if (!a.gelu_tab) return AP_ERR_UNSUPPORTED;            // (AP_GELU_TABLE=0, or a capture in front of the table's first build)
        if (a.res) return ln ? mf_launch<384, false, true, true>(a, st) : mf_launch<384, false, false, true>(a, st);
        return ln ? mf_launch<384, false, true, false>(a, st) : mf_launch<384, false, false, false>(a, st);
    }
    return mf_launch<384, true>(a, st);
}

}  // extern "C"
