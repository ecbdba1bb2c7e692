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
// The two launches this replaces (fc1 + GELU 47 - 52 us, fc2 + residual 37 us at 25088 x 384 x 1152) write the hidden activation (57.8 MB), drain,
// fill again and read it back; each runs its K loop at what one CU pulls from L2 and its store phase at what the chip writes, one after the
// other.  Here a workgroup keeps its 128 rows of X as MFMA fragments in REGISTERS for the whole kernel, the hidden chunk never leaves the
// registers either -- the accumulator layout of phase 1 (lane (fr, g) holds 8 consecutive hidden units of row fr) IS the operand layout of
// phase 2 -- and the only thing that streams through LDS is the weights: 2 * 3 C * C * 2 bytes per workgroup, in 8-KB pieces by LDS-DMA.
//
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

#define MF_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define MF_LDS(p) ((__attribute__((address_space(3))) void*)(p))

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

template <int C, bool BWD>
static int mf_launch(const MlpArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_mlp_fused<C, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS_BYTES); attr = true; (void)hipGetLastError(); }
    hipLaunchKernelGGL((k_mlp_fused<C, BWD>), dim3(a.M / MF_BM), dim3(256), MF_LDS_BYTES, st, a);
    return ap_check_launch();
}

extern "C" {

int ap_mlp_fused(const ap_mlp_fused_args* p, ap_stream_t stream) {
    if (!p || !p->x || !p->wa || !p->wb || !p->out || !p->hidden_out || !p->codes) return AP_ERR_NULL;
    const int C = p->c, Hd = p->hidden, M = p->m;
    if (M <= 0 || C <= 0 || Hd <= 0) return AP_ERR_SHAPE;
    if ((p->ldx & 7) || (p->ldwa & 7) || (p->ldwb & 7) || (p->ldo & 7) || (p->ldh & 7) || p->ldx < C || p->ldwa < C || p->ldwb < Hd || p->ldo < C || p->ldh < Hd)
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
    (void)hipGetLastError();
    if (!p->backward) {
        a.gelu_tab = g8_gelu_table_ptr(st);
        if (!a.gelu_tab) return AP_ERR_UNSUPPORTED;            // (AP_GELU_TABLE=0, or a capture in front of the table's first build)
        return mf_launch<384, false>(a, st);
    }
    return mf_launch<384, true>(a, st);
}

}  // extern "C"
