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
#include "../../../autoprog_amd/csrc/gemm_epi.h"

#ifndef AP_DMA_ABL
#define AP_DMA_ABL 0      // timing-only ablation builds (tools/gemm_lab): 2 = no fragment reads / MFMAs, 4 = no DMA
#endif

// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the instruction takes an immediate, so dispatch over the values the ring
// can ask for (n <= 48); larger requests wait for 48 (waiting for MORE operations than necessary is always safe)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
#define AP_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        AP_VM_CASE(0) AP_VM_CASE(1) AP_VM_CASE(2) AP_VM_CASE(3) AP_VM_CASE(4) AP_VM_CASE(5) AP_VM_CASE(6) AP_VM_CASE(7) 
        AP_VM_CASE(8) AP_VM_CASE(9) AP_VM_CASE(10) AP_VM_CASE(11) AP_VM_CASE(12) AP_VM_CASE(13) AP_VM_CASE(14) AP_VM_CASE(15) 
        AP_VM_CASE(16) AP_VM_CASE(17) AP_VM_CASE(18) AP_VM_CASE(19) AP_VM_CASE(20) AP_VM_CASE(21) AP_VM_CASE(22) AP_VM_CASE(23) 
        AP_VM_CASE(24) AP_VM_CASE(25) AP_VM_CASE(26) AP_VM_CASE(27) AP_VM_CASE(28) AP_VM_CASE(29) AP_VM_CASE(30) AP_VM_CASE(31) 
        AP_VM_CASE(32) AP_VM_CASE(33) AP_VM_CASE(34) AP_VM_CASE(35) AP_VM_CASE(36) AP_VM_CASE(37) AP_VM_CASE(38) AP_VM_CASE(39) 
        AP_VM_CASE(40) AP_VM_CASE(41) AP_VM_CASE(42) AP_VM_CASE(43) AP_VM_CASE(44) AP_VM_CASE(45) AP_VM_CASE(46) AP_VM_CASE(47) 
        
        default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    }
#undef AP_VM_CASE
}

// bank-swizzle keys (XORed into the 16-byte chunk index of a row).  BK = 64: 128-byte rows, 8 chunks, key_a / key_b of
// gemm_epi.h.  BK = 32: 64-byte rows, 4 chunks: the 16 lanes of one ds_read_b128 group read rows {0-3,12-15} chunk c and rows
// {4-11} chunk c+1 (or the mirrored set); four rows share a 256-byte bank line, so rows 4j..4j+3 get key (-j)&3 = {0,3,2,1}:
// every group then touches 16 distinct 16-byte slots.  For the N-permuted weight rows (fragment row 4q+p lives at tile row
// 8q + 4(t&1) + p) the row group is q = (r>>3)&3.
template <int BK> __device__ __forceinline__ int dkey_a(int r) { return BK == 64 ? (r & 7) : ((0 - (r >> 2)) & 3); }
template <int BK> __device__ __forceinline__ int dkey_b(int r) { return BK == 64 ? ((r & 3) | (((r >> 3) & 1) << 2)) : ((0 - (r >> 3)) & 3); }

template <int TM, int TN, int WGM, int WGN, int ST, int BK = 64>
__global__ void __launch_bounds__(WGM * WGN * 64)
k_gemm_nt_dma(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
              int M, int N, int K, int tiles_n, int ntiles, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dma_raw[];
    constexpr int NW = WGM * WGN;
    constexpr int ROWS = TM + TN;
    constexpr int RPP = 512 / BK;                    // rows per DMA wave-instruction (1 KiB): 8 at BK 64, 16 at BK 32
    constexpr int CPR = BK / 8;                      // 16-byte chunks per row
    constexpr int PIECES = ROWS / RPP;
    constexpr int NI = (PIECES + NW - 1) / NW;       // DMA instructions per wave per K step (the last one may be absent)
    constexpr int MT = TM / WGM / 16, NT = TN / WGN / 16;
    constexpr int STAGE = ROWS * BK;                 // bf16 elements per stage
    constexpr int KS = BK / 32;                      // MFMA k-steps per stage
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(TM % RPP == 0 && ROWS % RPP == 0, "a DMA piece lies entirely in the activation or in the weight rows");
    static_assert((TM / WGM) % 16 == 0 && (TN / WGN) % 32 == 0, "wave tile: 16-row fragments, N-permuted fragment pairs");
    static_assert(ST >= 2 && ST <= 8, "ring depth");
    bf16_t* ring = reinterpret_cast<bf16_t*>(dma_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, g = lane >> 4;
    const int nk = K / BK;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;
    const int my_ni = (PIECES % NW == 0 || wave < PIECES % NW) ? NI : NI - 1;      // DMA instructions this wave issues per K step

    // ---- DMA issue cursor over the flattened (tile, k) stream
    uint32_t soff[NI];                               // per-lane source offsets in elements (launcher checks < 2^32)
    auto set_tile_src = [&](int ti) {
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = (wave + NW * i) * RPP + lane / CPR;
            if (r < TM) {
                const int c = (lane % CPR) ^ dkey_a<BK>(r);
                soff[i] = (uint32_t)min(m0 + r, M - 1) * (uint32_t)lda + c * 8;
            } else {
                const int rb = min(r - TM, TN - 1);
                const int c = (lane % CPR) ^ dkey_b<BK>(rb);
                soff[i] = (uint32_t)min(n0 + rb, N - 1) * (uint32_t)ldb + c * 8;
            }
        }
    };
    int q_tile = 0, q_k = 0, q_stage = 0;
    auto issue_next = [&]() {
        if (q_k == 0) set_tile_src(q_tile);
        if (!(AP_DMA_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int p = wave + NW * i;             // wave-uniform piece index
                if (i == NI - 1 && PIECES % NW != 0 && p >= PIECES) break;
                const int r0 = p * RPP;
                const bf16_t* base = (r0 < TM) ? A : B;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + soff[i] + q_k * BK),
                                                 (__attribute__((address_space(3))) void*)(ring + q_stage * STAGE + r0 * BK), 16, 0, 0);
            }
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
        wait_vmcnt_dyn((issued - s - 1) * my_ni);
        __builtin_amdgcn_s_barrier();
        // every wave has finished reading the slot of step s-1: refill it with step s + ST - 1
        if (issued < total) { issue_next(); ++issued; }
        const bf16_t* sA = ring + stage * STAGE;
        const bf16_t* sB = sA + TM * BK;
        if (!(AP_DMA_ABL & 2)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 xf[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int r = wm * (TM / WGM) + t * 16 + fr;
                xf[t] = as_bf16x8(ld16(sA + r * BK + (((ks * 4 + g) ^ dkey_a<BK>(r)) << 3)));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // fragment t of pair P = t>>1 holds weight rows 32P + 8q + 4(t&1) + p for MFMA row 4q + p: a lane's 4 + 4
                // accumulator registers of the pair are 8 CONSECUTIVE output columns
                const int r = wn * (TN / WGN) + 32 * (t >> 1) + 8 * (fr >> 2) + 4 * (t & 1) + (fr & 3);
                wf[t] = as_bf16x8(ld16(sB + r * BK + (((ks * 4 + g) ^ dkey_b<BK>(r)) << 3)));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        }
        stage = (stage == ST - 1) ? 0 : stage + 1;
        if (++kt < nk) continue;
        kt = 0;
        // ------------------------------------------------------------ direct epilogue of tile `ti`
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        ++ti;
        const int m0 = (tile / tiles_n) * TM + wm * (TM / WGM), n0 = (tile % tiles_n) * TN + wn * (TN / WGN);
        if ((AP_DMA_ABL & 1) || (ep.dbg & 1)) {
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


// ---------------------------------------------------------------------------------------------------------------------
// Second structure (lab measurements, profiles/r02_gemm_lab.txt): in k_gemm_nt_dma the three phases of a tile ADD UP
// (256x192 tile on the qkv shape: DMA only 19.7 us, MFMAs + fragment reads ~16 us, stores only 20.6 us, everything 49.8 us):
//   * all DMA instructions of a step were issued in one burst right behind the barrier; a VMEM instruction that finds the
//     CU's address pipeline full blocks its wave, so every wave sat in the burst before its first MFMA and nothing was sent
//     while the MFMAs ran.  Here the DMA instructions of a step are spread BETWEEN its MFMAs (ILV).
//   * the counted wait at the top of a step also drained the previous tile's epilogue stores (they are younger than the DMA
//     it waits for).  Here their number is added to the count for the ST-1 steps in which they are inside the window, so the
//     stores of a full tile retire behind the next tile's MFMAs.
template <int TM, int TN, int WGM, int WGN, int ST, int BK, int ILV>
__global__ void __launch_bounds__(WGM * WGN * 64)
k_gemm_nt_dma2(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
               int M, int N, int K, int tiles_n, int ntiles, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dma_raw[];
    constexpr int NW = WGM * WGN;
    constexpr int ROWS = TM + TN;
    constexpr int RPP = 512 / BK;
    constexpr int CPR = BK / 8;
    constexpr int PIECES = ROWS / RPP;
    constexpr int NI = (PIECES + NW - 1) / NW;
    constexpr int MT = TM / WGM / 16, NT = TN / WGN / 16;
    constexpr int STAGE = ROWS * BK;
    constexpr int KS = BK / 32;
    constexpr int NMFMA = KS * NT * MT;              // MFMAs of one K step
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(TM % RPP == 0 && ROWS % RPP == 0, "a DMA piece lies entirely in the activation or in the weight rows");
    static_assert((TM / WGM) % 16 == 0 && (TN / WGN) % 32 == 0, "wave tile: 16-row fragments, N-permuted fragment pairs");
    static_assert(ST >= 2 && ST <= 8, "ring depth");
    bf16_t* ring = reinterpret_cast<bf16_t*>(dma_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, g = lane >> 4;
    const int nk = K / BK;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;
    const int my_ni = (PIECES % NW == 0 || wave < PIECES % NW) ? NI : NI - 1;

    uint32_t soff[NI];
    auto set_tile_src = [&](int ti) {
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = (wave + NW * i) * RPP + lane / CPR;
            if (r < TM) {
                const int c = (lane % CPR) ^ dkey_a<BK>(r);
                soff[i] = (uint32_t)min(m0 + r, M - 1) * (uint32_t)lda + c * 8;
            } else {
                const int rb = min(r - TM, TN - 1);
                const int c = (lane % CPR) ^ dkey_b<BK>(rb);
                soff[i] = (uint32_t)min(n0 + rb, N - 1) * (uint32_t)ldb + c * 8;
            }
        }
    };
    int q_tile = 0, q_k = 0, q_stage = 0;            // cursor: the K step the NEXT DMA group belongs to
    auto issue_piece = [&](int i) {                  // one DMA instruction of the cursor's K step
        const int p = wave + NW * i;
        if (PIECES % NW != 0 && i == NI - 1 && p >= PIECES) return;
        const int r0 = p * RPP;
        const bf16_t* base = (r0 < TM) ? A : B;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + soff[i] + q_k * BK),
                                         (__attribute__((address_space(3))) void*)(ring + q_stage * STAGE + r0 * BK), 16, 0, 0);
    };
    auto advance = [&]() {
        if (++q_k == nk) { q_k = 0; ++q_tile; }
        q_stage = (q_stage == ST - 1) ? 0 : q_stage + 1;
    };

    f32x4 acc[NT][MT];
    int issued = 0;
    for (; issued < ST - 1 && issued < total; ++issued) {
        if (q_k == 0) set_tile_src(q_tile);
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_piece(i);
        advance();
    }
    int stage = 0, kt = 0, ti = 0;
    int epi_stores = 0, since_epi = 1 << 20;         // stores of the last full-tile epilogue, K steps since it ran
    const bool vec_ok = ((ldc & 7) == 0) && (ep.residual == nullptr || (ep.ldr & 7) == 0);
    for (int s = 0; s < total; ++s) {
        if (kt == 0) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // step s has landed once only the YOUNGER operations of this wave are outstanding: the DMA groups of steps s+1 .. issued-1
        // and, for ST-1 steps after a full-tile epilogue, its stores (issued after the DMA group of step s)
        ++since_epi;                                  // = s - (last step of the previous tile)
        wait_vmcnt_dyn((issued - s - 1) * my_ni + ((since_epi <= ST - 1) ? epi_stores : 0));
        __builtin_amdgcn_s_barrier();
        const bool do_issue = issued < total;         // the slot of step s-1 is free now: refill it with step s + ST - 1
        if (do_issue && q_k == 0) set_tile_src(q_tile);
        if (!ILV && do_issue && !(AP_DMA_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue_piece(i);
        }
        const bf16_t* sA = ring + stage * STAGE;
        const bf16_t* sB = sA + TM * BK;
        if (!(AP_DMA_ABL & 2)) {
            int j = 0, di = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 xf[MT], wf[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int r = wm * (TM / WGM) + t * 16 + fr;
                    xf[t] = as_bf16x8(ld16(sA + r * BK + (((ks * 4 + g) ^ dkey_a<BK>(r)) << 3)));
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int r = wn * (TN / WGN) + 32 * (t >> 1) + 8 * (fr >> 2) + 4 * (t & 1) + (fr & 3);
                    wf[t] = as_bf16x8(ld16(sB + r * BK + (((ks * 4 + g) ^ dkey_b<BK>(r)) << 3)));
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
                        ++j;
                        if (ILV && !(AP_DMA_ABL & 4) && di < NI && j == ((di + 1) * NMFMA) / (NI + 1)) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (do_issue) issue_piece(di);
                            __builtin_amdgcn_sched_barrier(0);
                            ++di;
                        }
                    }
            }
        } else if (ILV && do_issue && !(AP_DMA_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue_piece(i);
        }
        if (do_issue) { advance(); ++issued; }
        stage = (stage == ST - 1) ? 0 : stage + 1;
        if (++kt < nk) continue;
        kt = 0;
        // ------------------------------------------------------------ direct epilogue of tile `ti`
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        ++ti;
        const int m0 = (tile / tiles_n) * TM + wm * (TM / WGM), n0 = (tile % tiles_n) * TN + wn * (TN / WGN);
        if ((AP_DMA_ABL & 1) || (ep.dbg & 1)) {
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
        // every lane of this wave stored all its chunks (interior tile): exactly MT * NT/2 store instructions (twice with a stored
        // pre-activation) are in flight behind the DMA groups issued so far
        const bool interior = vec_ok && (m0 + TM / WGM <= M) && (n0 + TN / WGN <= N);
        epi_stores = interior ? MT * (NT / 2) * ((ep.gelu && ep.preact) ? 2 : 1) : 0;
        since_epi = 0;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Third structure: the MFMA loop itself.  Lab ablation of k_gemm_nt_dma2<256,192> on the qkv shape: the loop WITHOUT any global
// traffic (AP_DMA_ABL=5) takes 25 us where 588 tiles x 37.7 MFLOP need 13.2 us of MFMA issue on the busiest CU -- every K step
// opens with all 8 waves reading their fragments at once (nobody computes until the LDS queue has drained) and ends in a barrier.
// Here (BK = 32 so that more ring stages fit):
//   * the fragments of step s+1 are read WHILE the MFMAs of step s issue (two fragment sets, reads slotted between the MFMAs with
//     sched_group_barrier), so a step opens with MFMAs whose operands are already in registers;
//   * that needs stage s+1 landed one step early: the wait at the top of step s covers stage s+1, the ring (ST stages) keeps
//     ST-2 further stages in flight;
//   * K % 64 == 0 makes the number of 32-deep steps of a tile even, so the two fragment sets alternate in an unrolled pair of
//     steps with compile-time register indices.
template <int TM, int TN, int WGM, int WGN, int ST>
__global__ void __launch_bounds__(WGM * WGN * 64)
k_gemm_nt_dma3(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
               int M, int N, int K, int tiles_n, int ntiles, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dma_raw[];
    constexpr int BK = 32;
    constexpr int NW = WGM * WGN;
    constexpr int ROWS = TM + TN;
    constexpr int RPP = 16, CPR = 4;                 // rows per DMA instruction, 16-byte chunks per row
    constexpr int PIECES = ROWS / RPP;
    constexpr int NI = (PIECES + NW - 1) / NW;
    constexpr int MT = TM / WGM / 16, NT = TN / WGN / 16;
    constexpr int STAGE = ROWS * BK;
    constexpr int NMFMA = NT * MT, NRD = MT + NT;
    static_assert(TM % RPP == 0 && ROWS % RPP == 0, "a DMA piece lies entirely in the activation or in the weight rows");
    static_assert((TM / WGM) % 16 == 0 && (TN / WGN) % 32 == 0, "wave tile");
    static_assert(ST >= 3 && ST <= 8, "ring depth: computing + landed + >= 1 in flight");
    bf16_t* ring = reinterpret_cast<bf16_t*>(dma_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, g = lane >> 4;
    const int nk = K / BK;                           // even (K % 64 == 0)
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;
    const int my_ni = (PIECES % NW == 0 || wave < PIECES % NW) ? NI : NI - 1;

    uint32_t soff[NI];
    auto set_tile_src = [&](int ti) {
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = (wave + NW * i) * RPP + lane / CPR;
            if (r < TM) soff[i] = (uint32_t)min(m0 + r, M - 1) * (uint32_t)lda + ((lane % CPR) ^ dkey_a<BK>(r)) * 8;
            else { const int rb = min(r - TM, TN - 1); soff[i] = (uint32_t)min(n0 + rb, N - 1) * (uint32_t)ldb + ((lane % CPR) ^ dkey_b<BK>(rb)) * 8; }
        }
    };
    int q_tile = 0, q_k = 0, q_stage = 0;
    auto issue_piece = [&](int i) {
        const int p = wave + NW * i;
        if (PIECES % NW != 0 && i == NI - 1 && p >= PIECES) return;
        const int r0 = p * RPP;
        const bf16_t* base = (r0 < TM) ? A : B;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + soff[i] + q_k * BK),
                                         (__attribute__((address_space(3))) void*)(ring + q_stage * STAGE + r0 * BK), 16, 0, 0);
    };
    auto advance = [&]() {
        if (++q_k == nk) { q_k = 0; ++q_tile; }
        q_stage = (q_stage == ST - 1) ? 0 : q_stage + 1;
    };
    // lane-constant fragment addresses inside a stage
    int xoff[MT], woff[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { const int r = wm * (TM / WGM) + t * 16 + fr; xoff[t] = r * BK + ((g ^ dkey_a<BK>(r)) << 3); }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = wn * (TN / WGN) + 32 * (t >> 1) + 8 * (fr >> 2) + 4 * (t & 1) + (fr & 3);
        woff[t] = TM * BK + r * BK + ((g ^ dkey_b<BK>(r)) << 3);
    }
    // fragments are kept as plain 4 x u32 registers and bit-cast at the MFMA: a <8 x bf16> value that crosses control flow is
    // legalised element by element (16 shifts + 16 v_perm per fragment set and step in the first version of this loop)
    auto read_frags = [&](u32x4* xf, u32x4* wf, int stage) {
        const bf16_t* st = ring + stage * STAGE;
#pragma unroll
        for (int t = 0; t < MT; ++t) xf[t] = ld16(st + xoff[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t] = ld16(st + woff[t]);
    };

    f32x4 acc[NT][MT];
    u32x4 xf0[MT], wf0[NT], xf1[MT], wf1[NT];
    int issued = 0;
    for (; issued < ST - 1 && issued < total; ++issued) {
        if (q_k == 0) set_tile_src(q_tile);
#pragma unroll
        for (int i = 0; i < NI; ++i) issue_piece(i);
        advance();
    }
    // stage 0 landed -> first fragment set
    wait_vmcnt_dyn((issued - 1) * my_ni);
    __builtin_amdgcn_s_barrier();
    if (total > 0) read_frags(xf0, wf0, 0);
    int kt = 0, ti = 0;
    int epi_stores = 0, since_epi = 1 << 20;
    const bool vec_ok = ((ldc & 7) == 0) && (ep.residual == nullptr || (ep.ldr & 7) == 0);

    // one 32-deep step: MFMAs on (xc, wc) = stage s, fragment reads of stage s+1 into (xn, wn_) and the DMA of stage s+ST-1 between them
    auto step = [&](int s, const u32x4* xc, const u32x4* wc, u32x4* xn, u32x4* wn_) {
        if (kt == 0) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        ++since_epi;
        // stage s+1 must have landed (it is read during this step); younger operations of this wave: the DMA groups of stages
        // s+2 .. issued-1 and, for ST-2 steps after an epilogue, its stores
        const bool more = s + 1 < total;
        if (more) wait_vmcnt_dyn(max(issued - s - 2, 0) * my_ni + ((since_epi <= ST - 2) ? epi_stores : 0));
        __builtin_amdgcn_s_barrier();                 // stage s+1 readable by every wave; every wave is done reading stage s-1
        const bool do_issue = issued < total;
        if (do_issue && q_k == 0) set_tile_src(q_tile);
        const int nstage = (s + 1) % ST;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this step's fragments (read during the previous step)
        __builtin_amdgcn_sched_barrier(0);
        int j = 0, di = 0, ri = 0;
        const bf16_t* st = ring + nstage * STAGE;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wc[nt]), as_bf16x8(xc[mt]), acc[nt][mt], 0, 0, 0);
                ++j;
                // next step's fragment reads, one per two MFMAs from the start of the step
                // (unconditional: past the last step they fetch a stage nobody uses)
                if (!(AP_DMA_ABL & 2) && (j & 1) && ri < NRD) {
                    if (ri < MT) xn[ri] = ld16(st + xoff[ri]);
                    else wn_[ri - MT] = ld16(st + woff[ri - MT]);
                    ++ri;
                }
                if (!(AP_DMA_ABL & 4) && di < NI && j == ((di + 1) * NMFMA) / (NI + 1)) {
                    if (do_issue) issue_piece(di);
                    ++di;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (!(AP_DMA_ABL & 2)) {                      // wave tiles with more fragments than MFMA pairs: the rest of the reads
#pragma unroll
            for (; ri < NRD; ++ri) {
                if (ri < MT) xn[ri] = ld16(st + xoff[ri]);
                else wn_[ri - MT] = ld16(st + woff[ri - MT]);
            }
        }
        if (do_issue) { advance(); ++issued; }
    };

    for (int s = 0; s < total; s += 2) {
        step(s, xf0, wf0, xf1, wf1);
        kt += 1;
        step(s + 1, xf1, wf1, xf0, wf0);
        kt += 1;
        if (kt < nk) continue;
        kt = 0;
        // ------------------------------------------------------------ direct epilogue of tile `ti`
        const int tile = xcd_remap(blockIdx.x + ti * G, ntiles);
        ++ti;
        const int m0 = (tile / tiles_n) * TM + wm * (TM / WGM), n0 = (tile % tiles_n) * TN + wn * (TN / WGN);
        if ((AP_DMA_ABL & 1) || (ep.dbg & 1)) {
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
        const bool interior = vec_ok && (m0 + TM / WGM <= M) && (n0 + TN / WGN <= N);
        epi_stores = interior ? MT * (NT / 2) * ((ep.gelu && ep.preact) ? 2 : 1) : 0;
        since_epi = 0;
    }
}
