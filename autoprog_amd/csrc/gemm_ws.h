// Weight-stationary NT GEMM for K = 192 (round 5): the Outlooker's MLP products at 100352 rows (models/volo.py:156-163 at stage 1 of VOLO-D1:
// fc1 192 -> 576 with GELU, and the input gradient of fc2, 192 -> 576 times the stored gelu' codes).
//
//   C[M, N] = epi(A[M, 192] . W[N, 192]^T),   N a multiple of 192, M a multiple of 64
//
// Why another kernel.  With K = 192 a 256 x 192 tile of the 8-phase kernel (gemm8p.h) has three K-tiles of work and 144 KB of output: the
// launch is its epilogue, and a persistent launch runs K loop and epilogue one after the other in every CU at once.  Here the WEIGHTS stay:
// a wave keeps its 48 columns of a 192-column slice of W as MFMA fragments in registers for the whole kernel and the workgroup streams
// 64-row tiles of A through them -- rows two tiles ahead in flight, the finished tile leaving through an LDS staging pass as whole 16-byte
// row chunks.  Measured with the three workgroups of a 64-row tile on ONE XCD (tools/check_ws.py, rotating buffers, bit-identical results;
// DESIGN.md section 3 "Round 5"): 100352 x 576 fc1 + GELU 66.0 -> 62.1 us, times the codes 63.9 -> 52.7; 16384 x 192 14.1 -> 10.7 / 11.4 -> 8.5;
// 32768 and 73728 x 576 (the early AutoProg stages) 26.0 -> 25.8 / 22.5 -> 18.6 and 50.5 -> 48.6 / 48.1 -> 41.2; 204800 x 576 145.8 -> 133.4 /
// 113.2 -> 96.3; in the D1 step 72.0 -> 59.4 and 61.5 -> 50.8 us (this kernel needs 64 - 72 KB of LDS, not all 160: the next launch starts
// on a CU before the last workgroup has left it).  What it did NOT do is reach the 35 us the 212 MB of the large shape need: with loads,
// stores, MFMAs and table lookups ablated one by one (WS_ABL) no single one is worth more than 15 us and the bare skeleton is 35 -- the
// row phase and the staging pass through LDS cost what they cost in the 8-phase kernel.
//
//   workgroup : 512 threads, persistent, ONE column slice and a stream of 64-row tiles; the slices' workgroups of a tile share an XCD
//   two wave groups (waves 0-3 / 4-7), each with its OWN 32-row half of the tile, A buffer and staging buffer, ONE barrier apart: while a group
//               multiplies (LDS reads, MFMAs, staging writes) the other is in its row phase (table lookups, global stores, whose issue stalls
//               on the chip's write rate) -- a single group of eight waves did the two one after the other: 65 us for the 212 MB of fc1 + GELU
//   wave wn = wave & 3 of a group: all 32 rows (two 16-row tiles), columns 48 wn .. + 47 of the slice (three 16-column tiles)
//   registers : a wave's 48 columns x 192 k of W as 18 MFMA fragments (72 VGPRs), loaded once
//   LDS       : A tile [64][192] (bf16, rows of 384 bytes) | staging [64][192] (bf16 + the 16 KB GELU table, or fp32)
//   swizzle   : rows of 384 bytes start 32 banks apart: chunk c of row r sits at chunk (c & ~7) | ((c & 7) ^ key(r)),
//               key(r) = ((r >> 1) & 1) | (((r >> 2) & 3) << 1) -- conflict-free for the lane groups ds_read_b128 really uses
//               ({0-3, 12-15} of one 16-lane quarter with {4-11} of the next: MI355X_MICROARCH.md, LDS table)
//   MFMA      : D = W fragment (A operand) x A fragment (B operand): a lane ends with 4 consecutive columns of one row, fp32 accumulation
//               over ascending 32-deep K steps -- the order of the 8-phase kernel: results are bit-identical to it
//   EPI = 0   : + bias, GELU and its 8-bit derivative code from the table (ap_gemm_epilogue.gelu = 3), the codes to preact_out
//   EPI = 1   : * the stored codes (mul_by8)
#pragma once
#include "common.h"
#include "gemm_epi.h"

#ifndef WS_ABL
#define WS_ABL 0          // timing-only ablations (results WRONG): 1 no global stores, 2 no MFMAs, 4 no table lookups, 8 no A loads after the first
#endif
#define WS_K 192
#define WS_BN 192
#define WS_BM 64
#define WS_ROWB 384                               // bytes of an LDS row (192 bf16)
#define WS_LDS_W (WS_BN * WS_ROWB)                // 73728
#define WS_LDS_A (WS_BM * WS_ROWB)                // 24576: the two groups' 32-row halves
// EPI 0: bf16 staging + the 16 KB table; EPI 1: fp32 staging (the multiply by the stored derivative happens on the fp32 accumulator value,
// rounded once -- as in the 8-phase kernel)
#define WS_LDS_BYTES(EPI) (WS_LDS_A + ((EPI) == 0 ? WS_LDS_A + 16384 : 2 * WS_LDS_A))

__device__ __forceinline__ int ws_key(int r) { return ((r >> 1) & 1) | (((r >> 2) & 3) << 1); }
// byte offset of 16-byte chunk c (0 .. 23) of row r
__device__ __forceinline__ int ws_off(int r, int c) { return r * WS_ROWB + (((c & ~7) | ((c & 7) ^ ws_key(r & 15))) << 4); }

// fp32 staging (EPI 1): rows of 768 bytes (all start at bank 0): 16-byte chunk q of row r at chunk (q & ~7) | ((q & 7) ^ (r & 7))
__device__ __forceinline__ int ws_off32(int r, int q) { return r * (2 * WS_ROWB) + (((q & ~7) | ((q & 7) ^ (r & 7))) << 4); }

struct WsArgs {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldb;
    bf16_t* C; int ldc;
    int M, N, n_slices, n_items, per_xcd;     // n_items: 64-row tiles; per_xcd: tile streams per XCD (each served by n_slices workgroups)
};

template <int EPI>
__global__ void __launch_bounds__(512, 2) k_gemm_nt_ws(WsArgs a, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    unsigned char* const Al = ws_smem;
    unsigned char* const St = Al + WS_LDS_A;
    const unsigned* const Tab = reinterpret_cast<const unsigned*>(St + WS_LDS_A);
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3, ltid = tid & 255;
    unsigned char* const Ag = Al + grp * (WS_LDS_A / 2);
    unsigned char* const Sg = St + grp * (EPI == 0 ? WS_LDS_A / 2 : WS_LDS_A);
#define WS_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // Placement: workgroups are dealt to the 8 XCDs round-robin (blockIdx & 7).  The n_slices workgroups that share a 64-row tile -- one per column
    // slice, all reading the same rows of A at about the same time -- are taken from ONE XCD, so those rows come from memory once and through one L2
    // (dealt by blockIdx alone they sat on three XCDs: 310 MB of counter traffic for 212 MB of operands, and the launch was at the HBM roof with it)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= a.per_xcd * a.n_slices) return;
    const int slice = idx % a.n_slices;
    int item = xcd * a.per_xcd + idx / a.n_slices;                // this workgroup's first tile; it walks tiles item, item + 8 per_xcd, ...
    if (item >= a.n_items) return;
    const int n0 = slice * WS_BN;

    // ---- once: this wave's 48 columns of the weight slice as MFMA fragments, in REGISTERS for the whole kernel (72 of them): the weights are what
    // every tile shares, and as registers they cost a tile no LDS read at all -- the first version kept the slice in LDS and spent its time on
    // 30 fragment reads per wave and half tile, one LDS latency per K step (35 of its 65 us with loads, stores, MFMAs and table lookups ablated)
    u32x4 wfr[3][WS_K / 32];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int ks = 0; ks < WS_K / 32; ++ks) wfr[nt][ks] = ld16(a.W + (int64_t)(n0 + 48 * wn + 16 * nt + fr) * a.ldb + (4 * ks + g) * 8);
    if constexpr (EPI == 0) {
        for (int id = tid; id < 1024; id += 512) st16(St + WS_LDS_A + id * 16, ld16(ep.gelu_tab + id * 4));
    }
    float bias[3][4];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[nt][r] = (EPI == 0 && ep.bias) ? ep.bias[n0 + 48 * wn + 16 * nt + 4 * g + r] : 0.f;

    // ---- per-thread pieces of a tile: three 16-byte chunks of the A tile (rows / chunk fixed per thread), three of the staged output
    int arow[3], achk[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int id = ltid + 256 * i; arow[i] = id / 24; achk[i] = id - arow[i] * 24; }      // rows 0 .. 31 of this group's half
    u32x4 areg[3];
    auto request = [&](int it) {
        const int64_t m0 = (int64_t)it * WS_BM + 32 * grp;
#pragma unroll
        for (int i = 0; i < 3; ++i) areg[i] = ld16(a.A + (m0 + arow[i]) * a.lda + achk[i] * 8);
    };
    u32x2 cq[3];                                                  // EPI 1: a tile's derivative codes, 8 per staged chunk
    auto request_codes = [&](int it) {
        if constexpr (EPI == 1) {
            const int64_t m0 = (int64_t)it * WS_BM + 32 * grp;
#pragma unroll
            for (int i = 0; i < 3; ++i) cq[i] = *reinterpret_cast<const u32x2*>(ep.mul8 + (m0 + arow[i]) * a.ldc + n0 + achk[i] * 8);
        }
    };
    const int step = 8 * a.per_xcd;
    request(item);
#pragma unroll
    for (int i = 0; i < 3; ++i) st16(Ag + ws_off(arow[i], achk[i]), areg[i]);
    // Memory schedule (vmcnt is ONE in-order counter for loads and stores, and hipcc waits for a load with everything older): the rows of tile
    // t + 2 and the codes of tile t + 1 are requested at the END of tile t's row phase, behind its stores, and first used a whole barrier
    // interval later (the rows after tile t + 1's B2, dropped into the A buffer that is free by then; the codes in tile t + 1's row phase),
    // BEFORE that tile's own stores are issued -- no wait in the loop stands in front of a round trip it did not have an interval to make.
    // (Requested at the top of a tile and awaited at its B2, one MFMA phase later, the loop ran at 1.8 us per interval for ~0.8 of work.)
    request(item + step < a.n_items ? item + step : item);
    request_codes(item);
    __syncthreads();                                              // the weight slice, the table and the first half tiles are in LDS
    if (grp == 1) WS_BAR();                                       // the second group runs one barrier behind the first
    while (item < a.n_items) {
        const int64_t m0 = (int64_t)item * WS_BM + 32 * grp;
        WS_BAR();                                                 // B1: this group's half tile is in LDS and its staging is free (the other group: B2)
        const int next = item + step, next2 = next + step;
        f32x4 acc[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < WS_K / 32; ++ks) {
            u32x4 af[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = ld16(Ag + ws_off(16 * mt + fr, 4 * ks + g));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    if (!(WS_ABL & 2)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wfr[nt][ks]), as_bf16x8(af[mt]), acc[mt][nt], 0, 0, 0);
                    else { acc[mt][nt][0] += __uint_as_float(wfr[nt][ks][0] ^ af[mt][0]); }
        }
        // ---- accumulators -> staging: row 16 mt + fr of the half, columns 48 wn + 16 nt + 4 g .. + 3
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int r = 16 * mt + fr, n = 48 * wn + 16 * nt + 4 * g;
                if constexpr (EPI == 0) {
                    u32x2 pk;
                    pk[0] = pack_bf2(acc[mt][nt][0] + bias[nt][0], acc[mt][nt][1] + bias[nt][1]);
                    pk[1] = pack_bf2(acc[mt][nt][2] + bias[nt][2], acc[mt][nt][3] + bias[nt][3]);
                    *reinterpret_cast<u32x2*>(Sg + ws_off(r, n >> 3) + ((n & 4) << 1)) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(Sg + ws_off32(r, n >> 2)) = acc[mt][nt];
                }
            }
        WS_BAR();                                                 // B2: the staged half tile is complete, its A buffer free (the other group: B1)
#pragma unroll
        for (int i = 0; i < 3; ++i) st16(Ag + ws_off(arow[i], achk[i]), areg[i]);
        // ---- rows out: three 16-byte chunks per thread (the positions its A chunks had)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r = arow[i], p = achk[i];
            const int64_t m = m0 + r;
            const int n = n0 + p * 8;
            u32x4 o;
            if constexpr (EPI == 0) {
                const u32x4 x = ld16(Sg + ws_off(r, p));
                // x = 8 bf16-rounded pre-activations: gelu(h) = h * Phi(h) and the derivative code from the table (gemm_epi.h)
                unsigned e8[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (WS_ABL & 4) { e8[2 * q] = x[q]; e8[2 * q + 1] = x[q] >> 3; continue; }
                    e8[2 * q] = Tab[gq_tab_index<0>(x[q])]; e8[2 * q + 1] = Tab[gq_tab_index<16>(x[q])];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pack_bf2(bf_lo(x[q]) * __uint_as_float(e8[2 * q]), bf_hi(x[q]) * __uint_as_float(e8[2 * q + 1]));
                u32x2 gq;
                gq[0] = __builtin_amdgcn_perm(e8[1], e8[0], 0x0c0c0400u) | (__builtin_amdgcn_perm(e8[3], e8[2], 0x0c0c0400u) << 16);
                gq[1] = __builtin_amdgcn_perm(e8[5], e8[4], 0x0c0c0400u) | (__builtin_amdgcn_perm(e8[7], e8[6], 0x0c0c0400u) << 16);
                // 8 code bytes per lane: the lane with the even chunk of a pair takes its neighbour's and stores 16 (24 chunks per row: pairs never straddle rows)
                const unsigned n0lo = (unsigned)__shfl_xor((int)gq[0], 1, 64), n0hi = (unsigned)__shfl_xor((int)gq[1], 1, 64);
                if (!(p & 1)) {
                    u32x4 o4; o4[0] = gq[0]; o4[1] = gq[1]; o4[2] = n0lo; o4[3] = n0hi;
                    if (!(WS_ABL & 1) || o4[0] == 0x12345678u) st16_nt(reinterpret_cast<unsigned char*>(ep.preact) + m * a.ldc + n, o4);
                }
            } else {
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(Sg + ws_off32(r, 2 * p)), f1 = *reinterpret_cast<const f32x4*>(Sg + ws_off32(r, 2 * p + 1));
                float d[8];
                gq_unpack4(cq[i][0], d); gq_unpack4(cq[i][1], d + 4);
                o[0] = pack_bf2(f0[0] * d[0], f0[1] * d[1]); o[1] = pack_bf2(f0[2] * d[2], f0[3] * d[3]);
                o[2] = pack_bf2(f1[0] * d[4], f1[1] * d[5]); o[3] = pack_bf2(f1[2] * d[6], f1[3] * d[7]);
            }
            if (!(WS_ABL & 1) || o[0] == 0x12345678u) st16_nt(a.C + m * a.ldc + n, o);
        }
        if (!(WS_ABL & 8)) {
            request(next2 < a.n_items ? next2 : item);
            request_codes(next < a.n_items ? next : item);
        }
        item = next;
    }
    if (grp == 0) WS_BAR();                                       // (equal barrier counts for the two groups)
#undef WS_BAR
}
