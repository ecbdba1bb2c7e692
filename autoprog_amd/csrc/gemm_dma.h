// LDS-DMA pipelined NT GEMM  C[M,N] = epi(A[M,K] . B[N,K]^T)  for K % 64 == 0 (every Linear of the VOLO / DeiT
// models except the 1000-class heads' input gradients): models/volo.py:67,68,71,156,158,180,182 and their backward.
//
// Why a second main loop next to k_gemm_nt (gemm.hip): that kernel stages global -> VGPR -> ds_write_b128 -> LDS with ONE
// K step of prefetch; round-1 ablations put 16.7 of its 36 us into the LDS write/read/barrier phases.  Here
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write, and a ring of ST
//     stages keeps ST-1 K steps of loads in flight behind a COUNTED s_waitcnt vmcnt + one raw s_barrier per K step;
//   * the (tile, k) stream of a workgroup is flattened: launched with fewer workgroups than tiles it is persistent and
//     the DMA of the next tile's first K steps flies while the current tile's epilogue runs;
//   * the epilogue is DIRECT from the accumulators (weight fragments are N-permuted so a lane ends with 8 consecutive
//     output columns: 16-byte loads of residual / gelu input, 16-byte stores) -- no LDS round trip, no barrier.
// LDS image of a stage: [TM activation rows | TN weight rows] x 64 bf16 (128 B per row); the DMA writes LDS linearly
// (wave-uniform base + lane * 16 B), so the bank swizzle (16-byte chunk index ^ key(row)) is applied to the per-lane
// SOURCE address and again on the fragment reads.
#pragma once
#include "gemm_epi.h"

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int TM, int TN, int WGM, int WGN, int ST>
__global__ void __launch_bounds__(WGM * WGN * 64)
k_gemm_nt_dma(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
              int M, int N, int K, int tiles_n, int ntiles, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dma_raw[];
    constexpr int NW = WGM * WGN;
    constexpr int ROWS = TM + TN;
    constexpr int PIECES = ROWS / 8;                 // one DMA wave-instruction fills 8 rows (1 KiB)
    constexpr int NI = PIECES / NW;                  // DMA instructions per wave per K step
    constexpr int MT = TM / WGM / 16, NT = TN / WGN / 16;
    constexpr int STAGE = ROWS * 64;                 // bf16 elements per stage
    static_assert(TM % 8 == 0 && PIECES % NW == 0, "stage rows split evenly over the waves");
    static_assert((TM / WGM) % 16 == 0 && (TN / WGN) % 32 == 0, "wave tile: 16-row fragments, N-permuted fragment pairs");
    static_assert(ST >= 2 && ST <= 4, "ring depth");
    bf16_t* ring = reinterpret_cast<bf16_t*>(dma_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, g = lane >> 4;
    const int nk = K >> 6;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;

    // ---- DMA issue cursor over the flattened (tile, k) stream
    uint32_t soff[NI];                               // per-lane source offsets in elements (launcher checks < 2^32)
    auto set_tile_src = [&](int ti) {
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = (wave + NW * i) * 8 + (lane >> 3);
            if (r < TM) {
                const int c = (lane & 7) ^ key_a(r);
                soff[i] = (uint32_t)min(m0 + r, M - 1) * (uint32_t)lda + c * 8;
            } else {
                const int rb = r - TM;
                const int c = (lane & 7) ^ key_b(rb);
                soff[i] = (uint32_t)min(n0 + rb, N - 1) * (uint32_t)ldb + c * 8;
            }
        }
    };
    int q_tile = 0, q_k = 0, q_stage = 0;
    auto issue_next = [&]() {
        if (q_k == 0) set_tile_src(q_tile);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r0 = (wave + NW * i) * 8;      // wave-uniform first row of this piece
            const bf16_t* base = (r0 < TM) ? A : B;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + soff[i] + q_k * 64),
                                             (__attribute__((address_space(3))) void*)(ring + q_stage * STAGE + r0 * 64), 16, 0, 0);
        }
        if (++q_k == nk) { q_k = 0; ++q_tile; }
        q_stage = (q_stage == ST - 1) ? 0 : q_stage + 1;
    };

    f32x4 acc[NT][MT];
    int issued = 0;
    for (; issued < ST - 1 && issued < total; ++issued) issue_next();
    int stage = 0, kt = 0, ti = 0;
    const bool vec_ok = ((ldc & 7) == 0) && (ep.residual == nullptr || (ep.ldr & 7) == 0);
    for (int s = 0; s < total; ++s) {
        if (kt == 0) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // K steps issued beyond s: issued - s - 1 in [0, ST-2].  Wait until only those are outstanding: step s has landed
        // (this wave's pieces; the barrier extends it to every wave's).  Over-waiting (epilogue stores of the previous tile are
        // younger than some of these DMAs and get drained too) is always safe.
        const int ahead = issued - s - 1;
        if (ST >= 4 && ahead >= 2) wait_vmcnt<2 * NI>();
        else if (ST >= 3 && ahead >= 1) wait_vmcnt<NI>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // every wave has finished reading the slot of step s-1: refill it with step s + ST - 1
        if (issued < total) { issue_next(); ++issued; }
        const bf16_t* sA = ring + stage * STAGE;
        const bf16_t* sB = sA + TM * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int r = wm * (TM / WGM) + t * 16 + fr;
                xf[t] = as_bf16x8(ld16(sA + r * 64 + (((ks * 4 + g) ^ key_a(r)) << 3)));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // fragment t of pair P = t>>1 holds weight rows 32P + 8q + 4(t&1) + p for MFMA row 4q + p: a lane's 4 + 4
                // accumulator registers of the pair are 8 CONSECUTIVE output columns
                const int r = wn * (TN / WGN) + 32 * (t >> 1) + 8 * (fr >> 2) + 4 * (t & 1) + (fr & 3);
                wf[t] = as_bf16x8(ld16(sB + r * 64 + (((ks * 4 + g) ^ key_b(r)) << 3)));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        stage = (stage == ST - 1) ? 0 : stage + 1;
        if (++kt < nk) continue;
        kt = 0;
        // ------------------------------------------------------------ direct epilogue of tile `ti`
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        ++ti;
        const int m0 = (tile / tiles_n) * TM + wm * (TM / WGM), n0 = (tile % tiles_n) * TN + wn * (TN / WGN);
        if (ep.dbg & 1) {
            float sacc = 0.f;
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
            if (sacc == 12345.678f) C[0] = 1;
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + fr;
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {
                const int n = n0 + 32 * pr + 8 * g;
                if (m >= M || n >= N) continue;
                float v[8];
                v[0] = acc[2 * pr][mt][0]; v[1] = acc[2 * pr][mt][1]; v[2] = acc[2 * pr][mt][2]; v[3] = acc[2 * pr][mt][3];
                v[4] = acc[2 * pr + 1][mt][0]; v[5] = acc[2 * pr + 1][mt][1]; v[6] = acc[2 * pr + 1][mt][2]; v[7] = acc[2 * pr + 1][mt][3];
                epi_chunk(v, m, n, N, ldc, vec_ok, ep, C);
            }
        }
    }
}
