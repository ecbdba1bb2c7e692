// bf16 MFMA GEMMs for the Linear layers of the VOLO/DeiT hot path (models/volo.py:67,68,71,
// 156,158,180,182,253,256,258,547,553 and their autograd backward).
//
//   ap_gemm_nt     C[M,N]  = epi(A[M,K] . B[N,K]^T)      forward (B = weight) and input-gradient
//                                                        (B = pre-transposed weight) GEMMs
//   ap_gemm_tn_acc C[N1,N2] += A[M,N1]^T . B[M,N2]       weight gradients, fp32 accumulate
//
// Both use v_mfma_f32_16x16x32_bf16 (lane maps pinned on hardware by tools/probe, profiles/
// r01_hw_probe.txt): A operand lane l holds A[row l&15][k 8*(l>>4)+j], B operand B[k][col l&15],
// accumulator D[row 4*(l>>4)+r][col l&15].
//
// gemm_nt: 128x128x64 tile, 4 waves (2x2) x 64x64 per wave.  Operands are staged global ->
// registers -> LDS (16-byte chunks) with the 16-B chunk index XORed by (row&7): ds_write_b128 of a
// row and ds_read_b128 of an MFMA fragment are both bank-conflict free.  The MFMA is issued as
// D = W_frag (A operand) x X_frag (B operand) so that a lane ends with 4 CONSECUTIVE output columns
// of one output row -> 8-byte bf16 stores and vector loads of bias / residual in the epilogue.
// The next K tile's global loads are issued before the MFMAs of the current one.
//
// gemm_tn: the reduction runs over the token axis, which is the SLOW axis of both operands, so
// fragments are fetched with ds_read_b64_tr_b16 (hardware transpose) from row-major [64 tok][128]
// LDS tiles; the chunk swizzle f(row) keeps those transposed reads conflict free.  The token range
// is split across blockIdx.z and partial tiles are added with fp32 atomics (few MB per call).
#include <mutex>
#include "common.h"
#include <type_traits>
#include "gemm_epi.h"
#include "gemm8p.h"
#include "gemm_ws.h"
#include "gemm_tn8p.h"
// (the measured-and-rejected kernels of rounds 1 and 2 -- LDS-DMA rings, persistent 256-row tiles, the ring weight-gradient kernel -- and
// their AP_GEMM_NT_P / _RING / _DMA, AP_GEMM_TN_RING switches live under tools/gemm_lab/rejected/, outside the product library)
#include <cstdlib>
#include <cstdio>

#ifndef AP_ABL
#define AP_ABL 0
#endif
#define SBM 128
#define SBN 128
#define SBK 64

// Register-staged multi-workgroup kernel, templated on the block tile TM x TN and the wave grid WGM x WGN
// (wave tile (TM/WGM) x (TN/WGN), BK = 64).  Variants are selected per shape by ap_gemm_nt.
// PRE = true: the instantiation used when the epilogue READS a second tile (residual, or the stored pre-activation of gelu'): that
// tile is prefetched under the last K step (see "epilogue input prefetch" below).  PRE = false is the plain kernel, unchanged.
// amdgpu_waves_per_eu: the PRE kernel keeps the register budget of the plain one (three workgroups of the 128x128 tile per CU);
// without the hint the scheduler spends a wave of occupancy on interleaving the erf evaluations of the epilogue chunks.
// PATCH = 1: the rows of A are patches of an NHWC feature map (PatchMap, gemm_epi.h); PATCH = 2: the rows of C are (the input gradient
// of such a patch convolution, written straight into the feature-map layout; plain epilogue only).
// FP8 = true: A and B hold OCP e4m3 bytes (configs[4] "mixed MFMA fp8 GEMM / bf16 accum" -> fp32 accumulate here).  The kernel is
// launched with K, lda, ldb counted in PAIRS of bytes, so staging, swizzle and addressing are the bf16 kernel's; a 16-byte fragment is
// 16 K positions = two v_mfma_f32_16x16x32_fp8_fp8 on its 8-byte halves (the same K set on both operands), and the accumulators are
// scaled by the two dequantisation factors in the epilogue.
template <int TM, int TN, int WGM, int WGN, bool DEPI = true, int PF = 1, bool PRE = false, int PATCH = 0, bool FP8 = false>
__global__ void __launch_bounds__(WGM * WGN * 64) __attribute__((amdgpu_waves_per_eu((PRE && TM * TN == 128 * 128 && WGM * WGN == 4) ? 3 : 1)))
k_gemm_nt(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
          int M, int N, int K, int tiles_n, int ntiles, EpiArgs ep) {
    constexpr int NTH = WGM * WGN * 64;
    constexpr int MT = TM / WGM / 16, NT = TN / WGN / 16;
    constexpr int NA = TM * 8 / NTH, NB = TN * 8 / NTH;              // 16-B chunks per thread per K step
    constexpr int EROWS_RAW = ((TM + TN) * 32 / TN) / 16 * 16;        // rows of the fp32 staging tile that fit the LDS
    constexpr int EROWS = EROWS_RAW < TM ? EROWS_RAW : TM;
    constexpr int PASSES = (TM + EROWS - 1) / EROWS;
    static_assert(TM * 8 % NTH == 0 && TN * 8 % NTH == 0, "staging split");
    static_assert(PATCH != 2 || (!DEPI && !PRE), "patch-addressed C rows: staged plain epilogue only");
    static_assert((TM / WGM) % 16 == 0 && (TN / WGN) % 16 == 0, "wave tile");
    static_assert(EROWS >= 16 && EROWS % 16 == 0, "epilogue pass split");
    __shared__ __attribute__((aligned(16))) bf16_t smem_nt[(TM + TN) * SBK];       // A tile | B tile; reused by the epilogue
    bf16_t* sA = smem_nt;
    bf16_t* sB = smem_nt + TM * SBK;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, g = lane >> 4;

    // staging assignment: rows (tid>>3) + (NTH/8)*i, chunk kc = tid&7
    const int srow = tid >> 3, kc = tid & 7;
    const bf16_t* ga[NA];
    const bf16_t* gb[NB];
    int soffa[NA], soffb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = srow + (NTH / 8) * i;
        ga[i] = A + (PATCH == 1 ? patch_row(ep.pm, min(m0 + r, M - 1)) : (int64_t)min(m0 + r, M - 1) * lda) + kc * 8;
        soffa[i] = r * SBK + ((kc ^ (r & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = srow + (NTH / 8) * i;
        gb[i] = B + (int64_t)min(n0 + r, N - 1) * ldb + kc * 8;
        soffb[i] = r * SBK + ((kc ^ (DEPI ? key_b(r) : key_a(r))) << 3);
    }
    // PF register sets: the global loads of the next PF K steps are in flight while one step is computed.  PF = 2 was tried on
    // the small tiles (K = 192..576, 3..9 steps) and LOST inside the training step (fc1+gelu of the outlooker 96 -> 111 us,
    // 192x576 plain 40.6 -> 45.3), and on the 128x128 tile (38.9 -> 64.6 us on the dfc1 shape): the extra 16-32 VGPRs cost a
    // wave (or two) per SIMD, and three co-resident workgroups already keep three K steps of loads in flight.  Default 1.
    u32x4 ra0[NA], rb0[NB], ra1[PF > 1 ? NA : 1], rb1[PF > 1 ? NB : 1];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // K % 64 == 0 (every Linear of the models except the 1000-class heads): unconditional loads.  The per-lane `ok ? load : 0`
    // of the ragged-K path compiles to one exec-masked branch block with 8 register zeroings per load pair.
    const bool kfull = (K & (SBK - 1)) == 0;
    auto gload = [&](u32x4* ra, u32x4* rb, int k0) {
        if (kfull) {
            const int ka = PATCH == 1 ? patch_col(ep.pm, k0) : k0;      // uniform: a 64-wide K step never straddles a kernel row
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = ld16(ga[i] + ka);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = ld16(gb[i] + k0);
        } else {
            const bool ok = (k0 + kc * 8) < K;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = ok ? ld16(ga[i] + k0) : zero4;
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = ok ? ld16(gb[i] + k0) : zero4;
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Epilogue input prefetch.  The residual tile (or the stored pre-activation of the gelu' epilogue) is TM x TN bf16 = NPRE
    // 16-byte chunks per thread, no more than the A/B staging registers hold -- and those are dead once the last K step has
    // been written to the LDS.  Loading the epilogue's input there, under the last step's MFMAs, takes its HBM latency out of
    // the epilogue (384x384 + residual: 32.7 us against 19.6 us plain before this, for 19 MB more traffic).  Full tiles only.
    constexpr int NPRE = TM * TN / 8 / NTH;
    constexpr int CPR_E = TN / 8;
    const bf16_t* emul = ep.dgelu_of ? ep.dgelu_of : ep.mul_by;          // the factor tile (gelu' input or stored derivative), else the residual
    const bf16_t* esrc = emul ? emul : ep.residual;
    const int elds = emul ? ldc : ep.ldr;
    // (the staged epilogue's (pass, iteration) -> chunk map is the row-major one only when a pass is a whole number of thread sweeps)
    // The chunks live in the staging registers themselves (chunk c in ra0[c] / rb0[c - NA]): no register beyond the main loop's.
    constexpr bool PFOK = PRE && !DEPI && PF == 1 && (NPRE <= NA + NB) && ((EROWS * (TN / 8)) % NTH == 0) && (TM % EROWS == 0) && (NTH % (TN / 8) == 0);
    static_assert(PFOK || !PRE, "PRE is instantiated only for tiles whose staged epilogue sweeps whole rows of chunks");
    const bool pf = PFOK && esrc != nullptr && (emul == nullptr || ep.residual == nullptr) && (ldc & 7) == 0 && (elds & 7) == 0 &&
                    m0 + TM <= M && n0 + TN <= N && !(ep.dbg & 2);
    auto pre = [&](int c) -> u32x4& { return c < NA ? ra0[c < NA ? c : 0] : rb0[c >= NA && c < NA + NB ? c - NA : 0]; };
    // addresses = one uniform tile base (+ a uniform per-chunk step) + ONE 32-bit per-thread offset: a single VGPR across the main loop
    auto eload = [&]() {
        // the row-major chunk order of the staged epilogue: chunk c of thread tid is (row tid / CPR_E + c * NTH / CPR_E, column chunk
        // tid % CPR_E) -- NTH % CPR_E == 0, a thread keeps its column chunk across the sweeps
        const bf16_t* tbase = esrc + (int64_t)m0 * elds + n0;
        const unsigned toff = (unsigned)((tid / CPR_E) * elds + 8 * (tid % CPR_E));
#pragma unroll
        for (int c = 0; c < NPRE; ++c) pre(c) = ld16(tbase + c * (NTH / CPR_E) * elds + toff);
    };

    // one K step: registers -> LDS, refill the same register set with step `refill_k`, MFMAs
    // (mode 0: a refill always follows; 1: the last step, the epilogue prefetch takes the refill's place; 2: decided at run time)
    // PATCH = 1 with ep.abn: this thread's chunk kc holds channels 8 kc .. + 7 of a 64-channel pixel in EVERY K step (a 64-deep step is one pixel)
    float bsc[PATCH == 1 ? 8 : 1], bsh[PATCH == 1 ? 8 : 1];
    bool a_bn = false;
    if constexpr (PATCH == 1) {
        a_bn = ep.abn.mean != nullptr;
        if (a_bn) bn_in_consts(ep.abn, kc * 8, bsc, bsh);
    }
    auto step = [&](u32x4* ra, u32x4* rb, int refill_k, auto mode) {
        if constexpr (PATCH == 1) {
            if (a_bn) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = bn_in_apply(ra[i], bsc, bsh);
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) st16(sA + soffa[i], ra[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) st16(sB + soffb[i], rb[i]);
        __syncthreads();
        if constexpr (mode.value == 0) gload(ra, rb, refill_k);
        else if constexpr (mode.value == 1) { if (pf) eload(); }
        else if (refill_k < K) gload(ra, rb, refill_k);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int r = wm * (TM / WGM) + t * 16 + fr;
                xf[t] = as_bf16x8(ld16(sA + r * SBK + (((ks * 4 + g) ^ (r & 7)) << 3)));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // DEPI: fragment t of pair P = t>>1 holds weight rows 32P + 8q + 4(t&1) + p for MFMA row 4q + p, so a
                // lane's 4 + 4 accumulator registers of the pair are 8 CONSECUTIVE output columns
                const int r = DEPI ? wn * (TN / WGN) + 32 * (t >> 1) + 8 * (fr >> 2) + 4 * (t & 1) + (fr & 3)
                                   : wn * (TN / WGN) + t * 16 + fr;
                wf[t] = as_bf16x8(ld16(sB + r * SBK + (((ks * 4 + g) ^ (DEPI ? key_b(r) : key_a(r))) << 3)));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (FP8) {
                        typedef __attribute__((ext_vector_type(2))) long l64x2;
                        const l64x2 wv = __builtin_bit_cast(l64x2, wf[nt]), xv = __builtin_bit_cast(l64x2, xf[mt]);
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wv[0], xv[0], acc[nt][mt], 0, 0, 0);
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wv[1], xv[1], acc[nt][mt], 0, 0, 0);
                    } else acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
                }
        }
        __syncthreads();
    };
    gload(ra0, rb0, 0);
    if constexpr (PF > 1) {
        if (SBK < K) gload(ra1, rb1, SBK);
        for (int k0 = 0; k0 < K; k0 += 2 * SBK) {
            step(ra0, rb0, k0 + 2 * SBK, std::integral_constant<int, 2>{});
            if (k0 + SBK < K) step(ra1, rb1, k0 + 3 * SBK, std::integral_constant<int, 2>{});
        }
    } else {
        if constexpr (PFOK) {
            for (int k0 = SBK; k0 < K; k0 += SBK) step(ra0, rb0, k0, std::integral_constant<int, 0>{});
            step(ra0, rb0, K, std::integral_constant<int, 1>{});
        } else {
            for (int k0 = 0; k0 < K; k0 += SBK) step(ra0, rb0, k0 + SBK, std::integral_constant<int, 2>{});
        }
    }
#if (AP_ABL & 128)
    if (ep.dbg & 1) {                     // ablation build only (AP_GEMM_DBG=1): main loop without the epilogue
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (sacc == 12345.678f) C[0] = 1;
        return;
    }
#endif

    if constexpr (FP8) {
        const float dq = ep.dq_a[0] * ep.dq_b[0];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] *= dq;
    }
    // ---------------------------------------------------------------- epilogue
    // PASSES passes of EROWS rows: accumulators -> fp32 [EROWS][TN] tile in LDS (16-B chunk swizzle) -> every
    // thread finishes 8 consecutive columns of a row with 16-byte coalesced global accesses
    const bool vec_ok = ((ldc & 7) == 0) && (ep.residual == nullptr || (ep.ldr & 7) == 0);
    if constexpr (DEPI) {
        // direct epilogue: no LDS round trip.  Lane (fr, g) owns, per fragment pair, columns 8g..8g+7 of row fr:
        // one wave instruction stores 16 rows x 64 contiguous bytes.
        static_assert(NT % 2 == 0, "N-permuted fragments come in pairs");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wm * (TM / WGM) + mt * 16 + fr;
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {
                const int n = n0 + wn * (TN / WGN) + 32 * pr + 8 * g;
                if (m >= M || n >= N) continue;
                float v[8];
                v[0] = acc[2 * pr][mt][0]; v[1] = acc[2 * pr][mt][1]; v[2] = acc[2 * pr][mt][2]; v[3] = acc[2 * pr][mt][3];
                v[4] = acc[2 * pr + 1][mt][0]; v[5] = acc[2 * pr + 1][mt][1]; v[6] = acc[2 * pr + 1][mt][2]; v[7] = acc[2 * pr + 1][mt][3];
                epi_chunk(v, m, n, N, ldc, vec_ok, ep, C);
            }
        }
        return;
    }
    float* ctile = reinterpret_cast<float*>(smem_nt);
    constexpr int CPR = TN / 8;
    constexpr int IPP = (EROWS * CPR + NTH - 1) / NTH;                // chunks per thread per pass
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const int prow0 = pass * EROWS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rt = wm * (TM / WGM) + mt * 16;                 // first row of this 16-row fragment within the tile
            if (rt >= prow0 && rt < prow0 + EROWS) {
                const int r = rt - prow0 + fr;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int c16 = (wn * (TN / WGN) + nt * 16) / 4 + g;
                    *reinterpret_cast<f32x4*>(ctile + r * TN + ((c16 ^ (r & 7)) << 2)) = acc[nt][mt];
                }
            }
        }
        __syncthreads();
        if constexpr (!PFOK) {
#pragma unroll 2
            for (int id = tid; id < EROWS * CPR; id += NTH) {
                const int r = id / CPR, j = id - r * CPR;
                const int m = m0 + prow0 + r, n = n0 + 8 * j;
                if (prow0 + r >= TM || m >= M || n >= N) continue;
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + r * TN + (((2 * j) ^ (r & 7)) << 2));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + r * TN + (((2 * j + 1) ^ (r & 7)) << 2));
                v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                if constexpr (PATCH == 2) st16(C + patch_row(ep.pm, m) + patch_col(ep.pm, n), pack8(v));      // N % 8 == 0, plain epilogue (host-checked)
                else epi_chunk(v, m, n, N, ldc, vec_ok, ep, C);
            }
        } else {
            // UNR chunks at a time in a ROLLED loop (a fully unrolled sweep holds 4 chunks of erf / gelu' temporaries); the
            // prefetched chunks are consumed in order, so after each group the rest move down by UNR registers
            constexpr int UNR = (IPP % 2 == 0) ? 2 : 1;
#pragma unroll 1
            for (int it0 = 0; it0 < IPP; it0 += UNR) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int id = tid + NTH * (it0 + u);
                    const int r = id / CPR, j = id - r * CPR;
                    const int m = m0 + prow0 + r, n = n0 + 8 * j;
                    if (m >= M || n >= N) continue;
                    float v[8];
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + r * TN + (((2 * j) ^ (r & 7)) << 2));
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + r * TN + (((2 * j + 1) ^ (r & 7)) << 2));
                    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                    epi_chunk(v, m, n, N, ldc, vec_ok, ep, C, pre(u), pf);
                }
#pragma unroll
                for (int c = 0; c + UNR < NPRE; ++c) pre(c) = pre(c + UNR);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ gemm_nt for very few rows (M <= 256)
// The class-attention blocks and the heads work on ONE token per image (M = batch = 128).  As 128x128 tiles these are
// 3-9 workgroups walking K serially, one exposed memory latency per 64-deep step: 19-51 us inside the training step for
// 0.04-0.1 GFLOP.  Here: 64x32 output tile per workgroup (24-72 workgroups), the 8 waves split K into 32-deep chunks, each
// wave loads its MFMA fragments straight from global memory with up to three chunks in flight, partial sums meet in LDS.
#define SK_STRIDE 36        // floats per LDS row of a wave's 64x32 partial tile (16-B aligned, spreads the banks)
__global__ void __launch_bounds__(512)
k_gemm_nt_skinny(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
                 int M, int N, int K, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) float sk_part[];          // [8 waves][64][SK_STRIDE]
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int nch = (K + 31) >> 5;            // the last chunk may be partial (K % 8 == 0: whole 16-byte pieces; lanes beyond K load zeros)
    const bf16_t* arow[4];
    const bf16_t* brow[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) arow[t] = A + (int64_t)min(m0 + t * 16 + fr, M - 1) * lda + g * 8;
#pragma unroll
    for (int t = 0; t < 2; ++t) brow[t] = B + (int64_t)min(n0 + t * 16 + fr, N - 1) * ldb + g * 8;
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (int c0 = wave; c0 < nch; c0 += 24) {          // rounds of up to 3 chunks per wave, chunks interleaved over the waves
        u32x4 xa[3][4], wb[3][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ch = c0 + 8 * r;
            const bool ok = ch < nch && ch * 32 + g * 8 < K;
#pragma unroll
            for (int t = 0; t < 4; ++t) xa[r][t] = ok ? ld16(arow[t] + ch * 32) : zero4;
#pragma unroll
            for (int t = 0; t < 2; ++t) wb[r][t] = ok ? ld16(brow[t] + ch * 32) : zero4;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wb[r][nt]), as_bf16x8(xa[r][mt]), acc[nt][mt], 0, 0, 0);
    }
    float* mine = sk_part + wave * 64 * SK_STRIDE;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            *reinterpret_cast<f32x4*>(mine + (mt * 16 + fr) * SK_STRIDE + nt * 16 + 4 * g) = acc[nt][mt];
    __syncthreads();
    if (tid < 256) {
        const int r = tid >> 2, j = tid & 3;
        const int m = m0 + r, n = n0 + 8 * j;
        if (m < M && n < N) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float* p = sk_part + (w * 64 + r) * SK_STRIDE + 8 * j;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
                v[0] += lo[0]; v[1] += lo[1]; v[2] += lo[2]; v[3] += lo[3]; v[4] += hi[0]; v[5] += hi[1]; v[6] += hi[2]; v[7] += hi[3];
            }
            const bool vec_ok = ((ldc & 7) == 0) && (ep.residual == nullptr || (ep.ldr & 7) == 0);
            epi_chunk(v, m, n, N, ldc, vec_ok, ep, C);
        }
    }
}


// ------------------------------------------------------------------------------------ wgrad
#define TM 64          // tokens per step (MFMA reduction)
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int row0, int col0, int lane) {
    // fragment for MFMA 16x16x32 whose k axis is the tile ROW axis: rows row0+8g..+7, cols col0..+15
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r1 = row0 + 8 * g + q, r2 = r1 + 4;
    const int j = (col0 >> 3) + (p >> 1);
    const bf16_t* a1 = tile + r1 * 128 + ((j ^ tn_swz(r1)) << 3) + (p & 1) * 4;
    const bf16_t* a2 = tile + r2 * 128 + ((j ^ tn_swz(r2)) << 3) + (p & 1) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a2));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// same fragment from a per-lane base offset computed once outside the token loop: the swizzle term is
// lane-constant (it depends on q = (lane&15)>>2 and g&1 only), so every read of the loop is
// base[t] + compile-time immediate (ks*32 rows, +4 rows for the second half)
__device__ __forceinline__ int tr_base(int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r1 = 8 * g + q;
    const int j = (col0 >> 3) + (p >> 1);
    return r1 * 128 + ((j ^ tn_swz(r1)) << 3) + (p & 1) * 4;
}
__device__ __forceinline__ bf16x8 tr_frag_at(const bf16_t* tile, int base, int ks) {
    const bf16_t* a1 = tile + base + ks * 32 * 128;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1 + 4 * 128));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// one weight-gradient problem of a (possibly grouped) launch
struct TnArgs {
    const bf16_t* A; const bf16_t* B; float* C; float* colsum;
    int lda, ldb, ldc, M, N1, N2, steps_per_split, t1, t2, nblocks, start;
    const bf16_t* cs_weight;    // per-token weights of the fused column sum (bf16 [M], nullptr = ones)
    float cs_scale;             // factor applied to the column sums
    float alpha;                // factor applied to the product
    float* slab;                // deterministic mode: partial tiles [splits][N1][N2] (+ [splits][N1] column sums behind them), else nullptr
    int splits;
    PatchMap pb;                // pb.group != 0: the rows of B are patches of an NHWC feature map (gemm_epi.h)
};
// deterministic mode: partial results are STORED per token split and summed in split order by k_tn_reduce
struct TnDet { float* slab; float* cs_slab; };
#define TN_MAX_GROUP 8
struct TnGroup { TnArgs p[TN_MAX_GROUP]; int count; };
// Placement of a grouped launch: entry b = (problem << 12 | split * tiles + tile) of workgroup b, TN_MAP_IDLE = none.  Workgroup
// b runs on XCD b % 8, so the host puts ALL output tiles of one (problem, token split) on one XCD: that split's operand slab is
// then fetched through ONE L2 instead of the 1.5-2 that a contiguous id range per XCD gives when 5 splits meet 8 XCDs
// (transformer block: 486 MB of beyond-L2 traffic per launch for 270 MB of operands before).
#define TN_MAP_MAX 1024
#define TN_MAP_IDLE 0xFFFFu
struct TnMap { unsigned short e[TN_MAP_MAX]; };

template <bool BPATCH = false, bool BBN = false>
__device__ __forceinline__ void tn_tile(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                        int M, int N1, int N2, int steps_per_split, float* __restrict__ colsum, int t1, int t2, int nblocks, int block,
                                        bf16_t* sA, bf16_t* sB, const bf16_t* __restrict__ cs_weight = nullptr, float cs_scale = 1.0f,
                                        float* __restrict__ slab = nullptr, int splits = 1, float alpha = 1.0f,
                                        const PatchMap pb = PatchMap{0, 0, 0, 0, 0, 0u, 0u}, const BnIn bbn = BnIn{nullptr, nullptr, nullptr, nullptr}) {
    // XCD-aware decode: workgroups that share an XCD (and its L2) get consecutive ids, i.e. all output
    // tiles of the SAME token split, so each token range is fetched from HBM by one L2 only
    // (before: 347 MB of beyond-L2 traffic for 77 MB of operands on the qkv shape)
    const int id = nblocks > 0 ? xcd_remap(block, nblocks) : block;        // nblocks <= 0: `block` is already split * tiles + tile (TnMap)
    const int bz = id / (t1 * t2), tl = id - bz * (t1 * t2);
    const int by = tl / t1, bx = tl - by * t1;
    const int n0 = bx * 128, k0 = by * 128;
    const int step_begin = bz * steps_per_split;
    const int total_steps = (M + TM - 1) / TM;
    const int step_end = min(total_steps, step_begin + steps_per_split);
    if (step_begin >= step_end) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int srow = tid >> 4, j = tid & 15;            // 4 chunks: rows srow + 16*i, chunk j
    const bool a_ok = (n0 + j * 8) < N1, b_ok = (k0 + j * 8) < N2;
    // two register sets: the loads of steps t+1 and t+2 are in flight while step t is computed
    // (rocprof: 1.4 waves/SIMD resident, 42 % of wave cycles parked waiting for the single prefetch)
    u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // full token steps: unconditional loads; the guarded form compiles to an exec-masked branch block with register zeroing
    // around every load
    // (columns beyond N1 / N2 only feed output rows / columns that are never stored: their chunk index is clamped into the
    // row instead of being guarded, so ragged tiles -- N = 192, 486, 576 -- take the fast path too)
    const int ca = min(n0 + j * 8, lda - 8), cb = min(k0 + j * 8, ldb - 8);
    // per-token weights of the fused column sum travel with the operands (same prefetch distance): loading them inside the MFMA loop
    // made the loop wait for ALL older loads -- the operand prefetch included -- and cost 0.8 ms per training step
    // (branch-free: a guarded load compiles to an exec-masked block, and the wait-count pass then drains the operand prefetch at its join)
    const bool cs_w = (colsum != nullptr) && (cs_weight != nullptr) && (by == 0) && (wk == 0);
    const bf16_t* wsrc = (cs_weight != nullptr) ? cs_weight : A;                  // any valid address when there are no weights
    const int wmax = (cs_weight != nullptr) ? ((M + 7) & ~7) - 8 : 0;            // the weight vector is padded to a multiple of 8 tokens
    // only the few waves that own a column-sum (wk == 0 of the by == 0 tiles) of a weighted problem load them: a PROVABLY wave-uniform
    // branch (readfirstlane), so the other waves neither issue the two extra loads per step nor wait for them
    const bool cs_w_u = __builtin_amdgcn_readfirstlane((int)cs_w) != 0;
    auto wload = [&](u32x4* rw, int step) {
        if (cs_w_u) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) rw[ks] = ld16(wsrc + min(step * TM + ks * 32 + 8 * (lane >> 4), wmax));
        }
    };
    // patch-addressed B (the weight gradient of a k x k / stride k convolution: B rows are patches of the layer input): the column part
    // of the address is fixed per thread, the row part costs one multiply-high per load
    // (a separate instantiation: the extra address registers cost the grouped kernel its second workgroup per CU)
    constexpr bool b_patch = BPATCH;
    const int cbp = b_patch ? patch_col(pb, min(k0 + j * 8, N2 - 8)) : 0;
    auto brow = [&](int m) { return b_patch ? patch_row(pb, m) + cbp : (int64_t)m * ldb + cb; };
    auto gload = [&](u32x4* ra, u32x4* rb, int step) {
        if (step < step_end && (step + 1) * TM <= M) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = step * TM + srow + 16 * i;
                ra[i] = ld16(A + (int64_t)m * lda + ca);
                rb[i] = ld16(B + brow(m));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = step * TM + srow + 16 * i;
            const bool ok = (m < M) && (step < step_end);
            ra[i] = (ok && a_ok) ? ld16(A + (int64_t)m * lda + n0 + j * 8) : zero4;
            rb[i] = (ok && b_ok) ? ld16(B + (b_patch ? brow(min(m, M - 1)) : (int64_t)m * ldb + k0 + j * 8)) : zero4;
        }
    };
    f32x4 acc[4][4];     // [nt][kt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // optional fused bias gradient: column sums of A = A^T . 1, one extra MFMA per A fragment with an
    // all-ones B fragment, done only by the wk==0 waves of the blockIdx.y==0 tiles
    const bool do_colsum = (colsum != nullptr) && (by == 0) && (wk == 0);
    f32x4 csum[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) csum[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

    int abase[4], bbase[4], woff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        abase[t] = tr_base(wn * 64 + t * 16, lane);
        bbase[t] = tr_base(wk * 64 + t * 16, lane);
        const int r = srow + 16 * t;
        woff[t] = r * 128 + ((j ^ tn_swz(r)) << 3);
    }
    // BBN (a patch instantiation of its own: as a run-time branch its 16 registers doubled the time of EVERY patch launch, 67 -> 130 us):
    // the B rows are relu(bn(.)) of the 64-channel map that is read; chunk j of a 128-column tile is channels 8 (j & 7) .. + 7
    float tsc[BBN ? 8 : 1], tsh[BBN ? 8 : 1];
    if constexpr (BBN) bn_in_consts(bbn, (j & 7) * 8, tsc, tsh);
    auto stage_and_compute = [&](u32x4* ra, u32x4* rb, u32x4* rw, int refill_step) {
        const u32x4 wcur[2] = {rw[0], rw[1]};          // this step's weights (the refill below overwrites the set)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st16(sA + woff[i], ra[i]);
            if constexpr (BBN) rb[i] = bn_in_apply(rb[i], tsc, tsh);
            st16(sB + woff[i], rb[i]);
        }
        __syncthreads();
        gload(ra, rb, refill_step);
        wload(rw, refill_step);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = tr_frag_at(sA, abase[t], ks);
                bfr[t] = tr_frag_at(sB, bbase[t], ks);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
            if (do_colsum) {
                const bf16x8 wfrag = cs_w ? __builtin_bit_cast(bf16x8, wcur[ks]) : ones;       // element j <-> token 8g + j of this k-step
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) csum[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], wfrag, csum[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    u32x4 rw0[2] = {zero4, zero4}, rw1[2] = {zero4, zero4};
    gload(ra0, rb0, step_begin);
    wload(rw0, step_begin);
    gload(ra1, rb1, step_begin + 1);
    wload(rw1, step_begin + 1);
    for (int step = step_begin; step < step_end; step += 2) {
        stage_and_compute(ra0, rb0, rw0, step + 2);
        if (step + 1 < step_end) stage_and_compute(ra1, rb1, rw1, step + 3);
    }
    const int fr = lane & 15, g = lane >> 4;
    // deterministic mode: this split's partial tile / column sums are STORED (k_tn_reduce adds the splits in order)
    float* cs_out = slab ? slab + (int64_t)splits * N1 * N2 + (int64_t)bz * N1 : colsum;
    if (do_colsum && fr == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + nt * 16 + 4 * g + r;
                if (n < N1) { if (slab) cs_out[n] = csum[nt][r] * cs_scale; else atomicAdd(cs_out + n, csum[nt][r] * cs_scale); }
            }
    }
    if (slab) {
        float* part = slab + (int64_t)bz * N1 * N2;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int kk = k0 + wk * 64 + kt * 16 + fr;
                if (kk >= N2) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + nt * 16 + 4 * g + r;
                    if (n < N1) part[(int64_t)n * N2 + kk] = acc[nt][kt][r] * alpha;
                }
            }
        return;
    }
#if (AP_ABL & 64)
    {   // ablation: no atomics (keep the accumulators live)
        float sacc = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) sacc += acc[nt][kt][0] + acc[nt][kt][1] + acc[nt][kt][2] + acc[nt][kt][3];
        if (sacc == 12345.678f) C[0] = 1.f;
        return;
    }
#endif
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int kk = k0 + wk * 64 + kt * 16 + fr;
            if (kk >= N2) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + nt * 16 + 4 * g + r;
                if (n < N1) atomicAdd(C + (int64_t)n * ldc + kk, acc[nt][kt][r] * alpha);
            }
        }
}

__global__ void __launch_bounds__(256)
k_gemm_tn(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
          int M, int N1, int N2, int steps_per_split, float* __restrict__ colsum, int t1, int t2, int nblocks) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[TM * 128];
    __shared__ __attribute__((aligned(16))) bf16_t sB[TM * 128];
    tn_tile(A, lda, B, ldb, C, ldc, M, N1, N2, steps_per_split, colsum, t1, t2, nblocks, blockIdx.x, sA, sB);
}

// Several weight gradients in ONE launch (all Linear layers of a block): the workgroups of the launch are shared
// between the problems, so each problem is split over fewer token ranges -> longer reduction loops and
// (#problems)x fewer fp32 atomics than one launch per problem (atomics were 12-17 us of a 34-53 us launch).
__global__ void __launch_bounds__(256)
k_gemm_tn_grouped(TnGroup grp) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[TM * 128];
    __shared__ __attribute__((aligned(16))) bf16_t sB[TM * 128];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TN_MAX_GROUP; ++i)
        if (i < grp.count && (int)blockIdx.x >= grp.p[i].start) pi = i;
    const TnArgs& a = grp.p[pi];
    tn_tile(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N1, a.N2, a.steps_per_split, a.colsum, a.t1, a.t2, a.nblocks,
            (int)blockIdx.x - a.start, sA, sB, a.cs_weight, a.cs_scale, a.slab, a.splits, a.alpha);
}
// LayerNorm dgamma / dbeta reductions riding in the grouped launch (ap_gemm_tn_acc_grouped_ln): the workgroups behind the placement
// table add the partial rows k_ln_bwd left (column sums of [nblocks][2 C]) -- the block's own reduction launch (5.8 us, 19 per step,
// nothing but latency) disappears into a launch that has idle workgroup slots anyway.
struct TnLnItem { const float* partial; float* dgamma; float* dbeta; int nblocks; int C; };
struct TnLn { TnLnItem it[AP_LN_MAX_BATCH]; int count; int first; };            // first: blockIdx.x of the first reduction workgroup
__device__ __forceinline__ void tn_ln_role(const TnLn& ln, int blk, float* red) {          // red: >= (blockDim.x / 32) * 33 floats of LDS
    int y = 0, b0 = 0;
    for (int i = 0; i + 1 < ln.count; ++i) {                                     // which reduction this workgroup belongs to
        const int nb = (2 * ln.it[i].C + 31) / 32;
        if (y == i && blk >= b0 + nb) { b0 += nb; y = i + 1; }
    }
    const float* __restrict__ partial = ln.it[y].partial;
    const int nblocks = ln.it[y].nblocks, C = ln.it[y].C, C2 = 2 * C;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, RG = (int)blockDim.x >> 5;      // 32 columns x RG row groups
    const int c = (blk - b0) * 32 + cx;
    if ((blk - b0) * 32 >= C2) return;
    float s = 0.f;
    if (c < C2) {
        int b = ry;
        for (; b + 15 * RG < nblocks; b += 16 * RG) {                            // 16 loads in flight per thread
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(int64_t)(b + u * RG) * C2 + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; b < nblocks; b += RG) s += partial[(int64_t)b * C2 + c];
    }
    red[ry * 33 + cx] = s;
    __syncthreads();
    if (ry == 0 && c < C2) {
        float t = 0.f;
        for (int r = 0; r < RG; ++r) t += red[r * 33 + cx];
        if (c < C) ln.it[y].dgamma[c] += t; else ln.it[y].dbeta[c - C] += t;
    }
}

// the same with the host-made placement table (one (problem, split) per XCD)
__global__ void __launch_bounds__(256)
k_gemm_tn_grouped_map(TnGroup grp, TnMap map, TnLn ln) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[TM * 128];
    __shared__ __attribute__((aligned(16))) bf16_t sB[TM * 128];
    if ((int)blockIdx.x >= ln.first) { tn_ln_role(ln, (int)blockIdx.x - ln.first, reinterpret_cast<float*>(sA)); return; }
    const unsigned e = map.e[blockIdx.x];
    if (e == TN_MAP_IDLE) return;
    const TnArgs& a = grp.p[e >> 12];
    tn_tile(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N1, a.N2, a.steps_per_split, a.colsum, a.t1, a.t2, 0,
            (int)(e & 0xFFFu), sA, sB, a.cs_weight, a.cs_scale, a.slab, a.splits, a.alpha);
}

// the 192 x 192-tile, LDS-DMA weight-gradient kernel (gemm_tn8p.h) with the same placement table and LayerNorm riders
// (up to AP_TN_MAX_GROUP = 32 problems -- the Linear layers of six transformer blocks, or of two and four outlooker blocks -- and one work item per CU: its own, smaller
// placement table, entry = problem << 11 | split * tiles + tile; the three arguments stay under 4 KB)
#define T8_MAP_MAX 320
#define T8_MAP_SHIFT 11
struct T8Map { unsigned short e[T8_MAP_MAX]; };
struct T8Group { T8Item p[AP_TN_MAX_GROUP]; int tiles[AP_TN_MAX_GROUP]; };
static_assert(sizeof(T8Group) + sizeof(T8Map) + sizeof(TnLn) <= 4096, "kernel arguments of k_gemm_tn_8p");
static_assert(sizeof(TnGroup) + sizeof(TnMap) + sizeof(TnLn) <= 4096, "kernel arguments of k_gemm_tn_grouped_map");
__global__ void __launch_bounds__(512, 2)
k_gemm_tn_8p(T8Group grp, T8Map map, TnLn ln) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t8_smem[];
    if ((int)blockIdx.x >= ln.first) { tn_ln_role(ln, (int)blockIdx.x - ln.first, reinterpret_cast<float*>(t8_smem)); return; }
    const unsigned e = map.e[blockIdx.x];
    if (e == TN_MAP_IDLE) return;
    const int pi = e >> T8_MAP_SHIFT, idx = e & ((1u << T8_MAP_SHIFT) - 1u), tiles = grp.tiles[pi];
    t8_item(grp.p[pi], idx % tiles, idx / tiles, t8_smem);
}

// one weight gradient whose B rows are patches of an NHWC feature map (PatchMap): a k x k / stride k convolution's dW
template <bool BBN>
__global__ void __launch_bounds__(256)
k_gemm_tn_patch(TnArgs a, BnIn bbn) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[TM * 128];
    __shared__ __attribute__((aligned(16))) bf16_t sB[TM * 128];
    tn_tile<true, BBN>(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N1, a.N2, a.steps_per_split, a.colsum, a.t1, a.t2, a.nblocks,
                  (int)blockIdx.x, sA, sB, a.cs_weight, a.cs_scale, a.slab, a.splits, a.alpha, a.pb, bbn);
}

// deterministic mode, second pass: C[n][k] += sum over splits (in split order) of the stored partial tiles; same for the column sums
__global__ void __launch_bounds__(256)
k_tn_reduce(TnGroup grp) {
    for (int pi = 0; pi < grp.count; ++pi) {
        const TnArgs& a = grp.p[pi];
        const int64_t nmat = (int64_t)a.N1 * a.N2, ntot = nmat + (a.colsum ? a.N1 : 0);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ntot; i += (int64_t)gridDim.x * 256) {
            if (i < nmat) {
                float acc = 0.f;
                for (int z = 0; z < a.splits; ++z) acc += a.slab[(int64_t)z * nmat + i];
                const int64_t n = i / a.N2, k = i - n * a.N2;
                a.C[n * a.ldc + k] += acc;
            } else {
                const int64_t n = i - nmat;
                float acc = 0.f;
                for (int z = 0; z < a.splits; ++z) acc += a.slab[(int64_t)a.splits * nmat + (int64_t)z * a.N1 + n];
                a.colsum[n] += acc;
            }
        }
    }
}


// the GELU table of gemm_epi.h (gelu = 3 launches of the 8-phase kernel): a static device array, filled once per process on the stream of
// its first use (no allocation inside the library; AP_GELU_TABLE=0: the row phase evaluates the functions instead)
__device__ unsigned g8_gelu_table[2 * GQ_TAB_N];
__global__ void __launch_bounds__(256) k_build_gelu_table(unsigned* tab) {
    const unsigned idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 2 * GQ_TAB_N) tab[idx] = gq_tab_entry(idx);
}
// (not static: csrc/mlp_fused.hip reads the same table; declared in gemm_epi.h)
// Built once per device, synchronously (the build used to be enqueued on the stream of its first use -- if that first use sat inside
// a stream capture the kernel was only RECORDED and later eager launches read an unbuilt table; launches on another stream had no dependency
// on it).  While `st` is capturing nothing can be built or waited for: the caller gets nullptr and takes the arithmetic path for that launch.
const unsigned* g8_gelu_table_ptr(hipStream_t st) {
    static std::mutex mu;
    static int state[16] = {0};                 // per device -- 0: not built, 1: built, -1: switched off
    static unsigned* ptr[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev] == 0) {
        const char* e = getenv("AP_GELU_TABLE");
        if (e && e[0] == '0') { state[dev] = -1; return nullptr; }
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr[dev]), HIP_SYMBOL(g8_gelu_table)) != hipSuccess || !ptr[dev]) { (void)hipGetLastError(); state[dev] = -1; return nullptr; }
        hipLaunchKernelGGL(k_build_gelu_table, dim3(2 * GQ_TAB_N / 256), dim3(256), 0, st, ptr[dev]);
        if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); state[dev] = -1; return nullptr; }
        state[dev] = 1;
    }
    return state[dev] == 1 ? ptr[dev] : nullptr;
}

// launch of the persistent 8-phase kernel (gemm8p.h): one instantiation per epilogue flavour of the training step, a generic one
// for anything else
template <int NT1, int EF>
static void g8_go(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_gemm_nt_8p<NT1, EF>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_BYTES); attr = true; (void)hipGetLastError(); }
    hipLaunchKernelGGL((k_gemm_nt_8p<NT1, EF>), dim3(grid), dim3(512), G8_LDS_BYTES, st, ga, ep);
}
// the 224-row instantiations (256 x 192 tiles only): the flavours of the N = 384 products of a transformer block -- plain, row scale,
// bias (+ row scale) + residual; -> false for any other flavour (the caller then launches 256-row tiles)
template <int EF>
static void g8_go224(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_gemm_nt_8p<1, EF, false, 224>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_BYTES); attr = true; (void)hipGetLastError(); }
    hipLaunchKernelGGL((k_gemm_nt_8p<1, EF, false, 224>), dim3(grid), dim3(512), G8_LDS_BYTES, st, ga, ep);
}
static bool g8_has224(const EpiArgs& ep) {
    const int f = g8_flavour(ep);
    return f == 0 || f == G8_RS || f == (G8_BIAS | G8_RES) || f == (G8_BIAS | G8_RS | G8_RES);
}
static void g8_pick224(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    switch (g8_flavour(ep)) {
        case 0: g8_go224<0>(ga, ep, grid, st); break;
        case G8_RS: g8_go224<G8_RS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_RES: g8_go224<G8_BIAS | G8_RES>(ga, ep, grid, st); break;
        default: g8_go224<G8_BIAS | G8_RS | G8_RES>(ga, ep, grid, st); break;
    }
}
template <int NT1>
static void g8_pick(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    switch (g8_flavour(ep)) {
        case 0: g8_go<NT1, 0>(ga, ep, grid, st); break;
        case G8_BIAS: g8_go<NT1, G8_BIAS>(ga, ep, grid, st); break;
        case G8_RS: g8_go<NT1, G8_RS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU: g8_go<NT1, G8_BIAS | G8_GELU>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_RS: g8_go<NT1, G8_BIAS | G8_GELU | G8_RS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_GTAB: g8_go<NT1, G8_BIAS | G8_GELU | G8_GTAB>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_RS | G8_GTAB: g8_go<NT1, G8_BIAS | G8_GELU | G8_RS | G8_GTAB>(ga, ep, grid, st); break;
        case G8_DGELU: g8_go<NT1, G8_DGELU>(ga, ep, grid, st); break;
        case G8_DGELU | G8_RS: g8_go<NT1, G8_DGELU | G8_RS>(ga, ep, grid, st); break;
        case G8_MUL: g8_go<NT1, G8_MUL>(ga, ep, grid, st); break;
        case G8_MUL | G8_RS: g8_go<NT1, G8_MUL | G8_RS>(ga, ep, grid, st); break;
        case G8_MUL8: g8_go<NT1, G8_MUL8>(ga, ep, grid, st); break;
        case G8_MUL8 | G8_RS: g8_go<NT1, G8_MUL8 | G8_RS>(ga, ep, grid, st); break;
        case G8_MUL8 | G8_Q8: g8_go<NT1, G8_MUL8 | G8_Q8>(ga, ep, grid, st); break;
        case G8_MUL8 | G8_RS | G8_Q8: g8_go<NT1, G8_MUL8 | G8_RS | G8_Q8>(ga, ep, grid, st); break;
        case G8_BIAS | G8_RES: g8_go<NT1, G8_BIAS | G8_RES>(ga, ep, grid, st); break;
        case G8_BIAS | G8_RS | G8_RES: g8_go<NT1, G8_BIAS | G8_RS | G8_RES>(ga, ep, grid, st); break;
        default: g8_go<NT1, -1>(ga, ep, grid, st); break;
    }
}
// the fp8 instantiations (ap_gemm_nt_fp8: the forward Linear layers of the configs[4] step): the flavours a transformer block's forward uses
template <int NT1, int EF>
static void g8_go_fp8(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_gemm_nt_8p<NT1, EF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_BYTES); attr = true; (void)hipGetLastError(); }
    hipLaunchKernelGGL((k_gemm_nt_8p<NT1, EF, true>), dim3(grid), dim3(512), G8_LDS_BYTES, st, ga, ep);
}
template <int NT1>
static void g8_pick_fp8(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    switch (g8_flavour(ep)) {
        case 0: g8_go_fp8<NT1, 0>(ga, ep, grid, st); break;
        case G8_BIAS: g8_go_fp8<NT1, G8_BIAS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU: g8_go_fp8<NT1, G8_BIAS | G8_GELU>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_RS: g8_go_fp8<NT1, G8_BIAS | G8_GELU | G8_RS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_GTAB: g8_go_fp8<NT1, G8_BIAS | G8_GELU | G8_GTAB>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_RS | G8_GTAB: g8_go_fp8<NT1, G8_BIAS | G8_GELU | G8_RS | G8_GTAB>(ga, ep, grid, st); break;
        case G8_BIAS | G8_RES: g8_go_fp8<NT1, G8_BIAS | G8_RES>(ga, ep, grid, st); break;
        case G8_BIAS | G8_RS | G8_RES: g8_go_fp8<NT1, G8_BIAS | G8_RS | G8_RES>(ga, ep, grid, st); break;
        default: g8_go_fp8<NT1, -1>(ga, ep, grid, st); break;
    }
}
static int g8_launch(int bn, const bf16_t* A, int lda, const bf16_t* B, int ldb, bf16_t* C, int ldc, int M, int N, int K, const EpiArgs& ep, hipStream_t st, bool fp8 = false) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr;
        n_cu = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    }
    G8Args ga;
    ga.A = A; ga.lda = lda; ga.B = B; ga.ldb = ldb; ga.C = C; ga.ldc = ldc; ga.M = M; ga.N = N; ga.K = K;
    ga.tiles_n = (N + bn - 1) / bn; ga.ntiles = ((M + 255) / 256) * ga.tiles_n;
    const int cap = n_cu & ~7;
    // 224-row tiles where they put more CUs to work inside ONE round of the persistent grid (a launch of fewer tiles than CUs takes
    // about one tile's time whatever the number of idle CUs: tools/tile_rounds_probe.py): VOLO-D1's N = 384 products at 25088 rows,
    // 98 x 2 = 196 tiles of 256 rows -> 112 x 2 = 224 tiles of 224.  AP_GEMM_BM224=0: 256-row tiles everywhere.
    static int bm224 = -1;
    if (bm224 < 0) { const char* e = getenv("AP_GEMM_BM224"); bm224 = e ? atoi(e) : 1; }
    if (bm224 && !fp8 && bn == 192 && g8_has224(ep)) {
        const int t224 = ((M + 223) / 224) * ga.tiles_n;
        if (ga.ntiles < cap && t224 <= cap && t224 > ga.ntiles) {
            ga.ntiles = t224;
            const int want4 = (t224 + 7) & ~7;
            g8_pick224(ga, ep, want4 < cap ? want4 : cap, st);
            return ap_check_launch();
        }
    }
    const int want = (ga.ntiles + 7) & ~7;        // a multiple of 8: the kernel deals tiles per XCD label
    const int grid = want < cap ? want : cap;
    if (fp8) { if (bn == 192) g8_pick_fp8<1>(ga, ep, grid, st); else g8_pick_fp8<2>(ga, ep, grid, st); }
    else if (bn == 192) g8_pick<1>(ga, ep, grid, st); else g8_pick<2>(ga, ep, grid, st);
    return ap_check_launch();
}

// Where the persistent 8-phase kernel is picked (measured INSIDE the training step, tools/instep_8p.sh; profiles/r03_gemm_instep_*):
// every launch with K % 64 == 0, M >= 4096 and N a multiple of 192 (256 x 192 tiles) or N >= 1024 (256 x 256 tiles, last column tile
// masked) -- 384 x 1152 plain 38.3 -> 30.7 us, + residual 45.3 -> 41.0, fc1 + GELU 61.5 -> 56.0, qkv 39.0 -> 35.0, the K = 192
// outlooker shapes 5 - 15 % -- except the gelu' epilogue, whose staged second operand still costs more than it saves (60.6 -> 62.3).
// AP_GEMM_8P = 0: never; 1 (default): as above; 2: the gelu' launches too.
static int use_8p(int M, int N, int K, int ldc, const EpiArgs& ep) {          // -> 0 (no), 192 or 256 (block tile width)
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("AP_GEMM_8P"); mode = e ? atoi(e) : 1; }
    if (mode == 0 || (K & 63) || K < 128 || M < 4096 || (N & 7) || (ldc & 7) || (ep.residual && (ep.ldr & 7))) return 0;
    if (ep.dgelu_of && (mode < 2 || ep.residual)) return 0;
    if ((ep.mul_by || ep.mul8) && ep.residual) return 0;
    int bn = N >= 1024 ? 256 : (N % 192 == 0 ? 192 : (N % 256 == 0 ? 256 : 0));
    if (N >= 384 && N % 192 == 0) {
        // both widths tile N: the persistent grid walks ceil(tiles / #CU) rounds of tiles, a 256-wide tile costs ~1.3 of a 192-wide one.
        // N = 1152 (4.5 -> 5 column tiles of 256): 25088 rows 2 x 1.3 against 3 rounds -> 256; the AutoProg stages' 8192 / 18432 rows
        // (128 / 192 px) one round against one, two against two -> 192 (12.2 against 13.8 us, 21.3 against 25.5); 12800 rows (160 px) one
        // round of 250 tiles against two of 300 -> 256 (16.3 against 21.5): tools/sweep_nt.py, AP_GEMM_NT_TILE = 20 / 21.
        static int ncu = 0;
        if (ncu == 0) { hipDeviceProp_t pr; int dev = 0; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
        const int64_t mt = (M + 255) / 256;
        const int64_t r192 = (mt * (N / 192) + ncu - 1) / ncu, r256 = (mt * ((N + 255) / 256) + ncu - 1) / ncu;
        bn = (r192 * 10 < r256 * 13) ? 192 : 256;
        // (N = 768, the D5 widths, at 50176 rows = batch 64: 784 tiles of 192 are 3.06 -> FOUR rounds, 588 of 256 three: 64.2 against 67.1 us
        // at K = 768, 231 against 241 - 257 at K = 3072; at 48608 rows three rounds of 192: 54.0 against 63.1.  N = 384 / 576 stay on 192.)
    }
    return N < 192 ? 0 : bn;
}

extern "C" {

// -> 1: not a launch of the weight-stationary kernel (the caller goes on); otherwise the launch's status.  AP_GEMM_WS=0: never
static int ws_try(const bf16_t* A, int lda, const bf16_t* B, int ldb, bf16_t* C, int ldc, int M, int N, int K, EpiArgs& ep, hipStream_t st) {
    static int on = -1, ncu = 0;
    if (on < 0) {
        const char* e = getenv("AP_GEMM_WS"); on = (e && e[0] == '0') ? 0 : 1;
        int dev = 0; hipDeviceProp_t pr;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    if (!on || K != WS_K || N % WS_BN || M % WS_BM || M < 16384 || (ldc & 15) || (lda & 7) || (ldb & 7)) return 1;
    if (ep.dgelu_of || ep.mul_by || ep.row_scale || ep.residual || ep.q8) return 1;
    int epi = -1;
    if (ep.gelu == 3 && ep.preact && !ep.mul8) {
        ep.gelu_tab = g8_gelu_table_ptr(st);
        if (ep.gelu_tab) epi = 0;
    } else if (!ep.gelu && ep.mul8 && !ep.bias) epi = 1;
    if (epi < 0) return 1;
    WsArgs a;
    a.A = A; a.lda = lda; a.W = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N;
    a.n_slices = N / WS_BN; a.n_items = M / WS_BM;
    if (a.n_slices > ncu / 8) return 1;
    a.per_xcd = (ncu / 8) / a.n_slices;                           // tile streams per XCD: 10 for three slices on 32 CUs (two CUs per XCD stay idle)
    const int grid = ncu & ~7;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_gemm_nt_ws<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES(0));
        (void)hipFuncSetAttribute((const void*)k_gemm_nt_ws<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES(1));
        attr = true; (void)hipGetLastError();
    }
    if (epi == 0) hipLaunchKernelGGL(k_gemm_nt_ws<0>, dim3(grid), dim3(512), WS_LDS_BYTES(0), st, a, ep);
    else hipLaunchKernelGGL(k_gemm_nt_ws<1>, dim3(grid), dim3(512), WS_LDS_BYTES(1), st, a, ep);
    return ap_check_launch();
}

int ap_gemm_nt(const ap_bf16* A, int lda, const ap_bf16* B, int ldb, ap_bf16* C, int ldc, int M, int N, int K,
               const ap_gemm_epilogue* epi, ap_stream_t stream) {
    if (!A || !B || !C) return AP_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return AP_ERR_SHAPE;
    if ((K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K || ldc < N) return AP_ERR_SHAPE;
    EpiArgs ep = {nullptr, 0, nullptr, nullptr, nullptr, 1, nullptr, 0, 0, nullptr, {0, 0, 0, 0, 0, 0u, 0u}, nullptr, nullptr, nullptr};
    { const char* e = getenv("AP_GEMM_DBG"); if (e) ep.dbg = atoi(e); }
    if (epi) {
        ep.bias = epi->bias; ep.gelu = epi->gelu; ep.preact = epi->preact_out; ep.dgelu_of = epi->dgelu_of; ep.mul_by = epi->mul_by;
        ep.row_scale = epi->row_scale; ep.rows_per_scale = epi->rows_per_scale > 0 ? epi->rows_per_scale : 1;
        ep.residual = epi->residual; ep.ldr = epi->ldr; ep.mul8 = epi->mul_by8;
        if (ep.residual && ep.ldr < N) return AP_ERR_SHAPE;
        if ((ep.mul_by != nullptr) + (ep.dgelu_of != nullptr) + (ep.mul8 != nullptr) > 1) return AP_ERR_SHAPE;
        if (ep.gelu < 0 || ep.gelu > 3 || (ep.gelu >= 2 && !ep.preact)) return AP_ERR_SHAPE;
        // q8_out of a bf16 launch (round 5): the output a second time as e4m3 bytes, for the fp8 input-gradient product that consumes it.
        // Exists in the 8-phase kernel's mul_by8 flavours only (the input gradient of fc2): anything else is refused, not silently skipped
        if (epi->q8_out) {
            ep.q8 = epi->q8_out; ep.q8_scale = epi->q8_scale; ep.q8_amax = epi->q8_amax;
            if (!ep.q8_scale || !ep.mul8 || ep.bias || ep.gelu || ep.residual || (ldc & 15)) return AP_ERR_UNSUPPORTED;
            if (!use_8p(M, N, K, ldc, ep)) return AP_ERR_UNSUPPORTED;
        }
    }
    (void)hipGetLastError();
    static int skinny = -1;
    if (skinny < 0) { const char* e = getenv("AP_GEMM_NT_NO_SKINNY"); skinny = (e && e[0] == '1') ? 0 : 1; }
    if (skinny && M <= 256 && (K & 7) == 0) {          // (K = 1000: the head's input gradient ran on three 128 x 128 workgroups, 31 us, until round 4)
        const size_t lds = (size_t)8 * 64 * SK_STRIDE * sizeof(float);
        static bool sk_attr = false;
        if (!sk_attr) { (void)hipFuncSetAttribute((const void*)k_gemm_nt_skinny, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); sk_attr = true; (void)hipGetLastError(); }
        hipLaunchKernelGGL(k_gemm_nt_skinny, dim3((N + 31) / 32, (M + 63) / 64), dim3(512), lds, (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, ep);
        return ap_check_launch();
    }
    // K = 192 with a GELU-table or a stored-derivative epilogue at many rows (the Outlooker's MLP): the weight-stationary kernel (gemm_ws.h)
    if (const int rc = ws_try(A, lda, B, ldb, C, ldc, M, N, K, ep, (hipStream_t)stream); rc != 1) return rc;
    // tile selection (measured on the VOLO-D1 shape list, tools/bench_gemm.py): AP_GEMM_NT_TILE forces a variant
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("AP_GEMM_NT_TILE"); forced = e ? atoi(e) : 0; }
    int variant = forced;
    if (variant == 0) {
        // measured on the D1 shape list (tools/bench_gemm.py, AP_GEMM_NT_TILE sweep): narrow/short problems want
        // small tiles (more workgroups, 4-6 waves/SIMD, no wasted columns at N = 192/486/576), the rest 128x128
        static int allow192 = -1;
        if (allow192 < 0) { const char* e = getenv("AP_GEMM_NT_192"); allow192 = (e && e[0] == '1') ? 1 : 0; }
        // 128x192 tiles (wave tile 64x96: 17 % fewer LDS bytes per FLOP, direct epilogue), AP_GEMM_NT_192=1: faster with
        // cache-warm operands (33.9 vs 37.2 us on the dfc1 shape when the same buffers are reused back to back) but the training
        // step, whose operands come from HBM, is 0.3 ms SLOWER with them -> off by default
        if (allow192 && N % 192 == 0 && N >= 384 && M >= 4096 && !ep.gelu && !ep.dgelu_of && !ep.residual && K >= 384) variant = 10;
        else if (const int bn8 = use_8p(M, N, K, ldc, ep)) variant = bn8 == 192 ? 20 : 21;   // persistent 8-phase kernel (gemm8p.h)
        else if (N <= 512 && K <= 256) variant = 4;                              // 64x64
        else if (N <= 256 || ((N % 128 != 0) && (N % 64 == 0))) variant = 2;     // 128x64
        else variant = 1;                                                        // 128x128
    }
    // epilogue flavour: the LDS-staged one (fully coalesced 16-B rows) wins on 128x128 / 64x64 tiles, the direct one
    // (64-B row segments, no LDS round trip) on the 128x64 tiles of the N = 576 shapes; AP_GEMM_LDS_EPI=0/1 forces one
    static int lds_epi_env = -2;
    if (lds_epi_env == -2) { const char* e = getenv("AP_GEMM_LDS_EPI"); lds_epi_env = e ? (e[0] == '1') : -1; }
    // tuning aid: AP_GEMM_NT_RULE="N,K,variant[,lds_epi];N,K,variant;..." overrides the choice for exact (N, K) pairs, so that a
    // candidate can be timed INSIDE the training step (bench.py AP_GEMM_TABLE=1) instead of in a cache-warm microbenchmark
    int rule_epi = -1;
    {
        static int nrules = -1;
        static int rules[16][4];
        if (nrules < 0) {
            nrules = 0;
            const char* e = getenv("AP_GEMM_NT_RULE");
            while (e && *e && nrules < 16) {
                int n_ = 0, k_ = 0, v_ = 0, l_ = -1, used = 0;
                const int got = sscanf(e, "%d,%d,%d%n", &n_, &k_, &v_, &used);
                if (got < 3) break;
                e += used;
                if (*e == ',') { int u2 = 0; if (sscanf(e, ",%d%n", &l_, &u2) == 1) e += u2; }
                rules[nrules][0] = n_; rules[nrules][1] = k_; rules[nrules][2] = v_; rules[nrules][3] = l_;
                ++nrules;
                if (*e == ';') ++e;
            }
        }
        for (int i = 0; i < nrules; ++i)
            if (rules[i][0] == N && rules[i][1] == K && forced == 0) { variant = rules[i][2]; rule_epi = rules[i][3]; }
    }
    // variants 20 / 21: the persistent 8-phase kernel of gemm8p.h with 256 x 192 / 256 x 256 tiles (whole 8-column chunks only)
    if ((variant == 20 || variant == 21) && ((K & 63) || (N & 7) || (ldc & 7) || (ep.residual && (ep.ldr & 7)))) variant = 1;
    // q8_out exists in the 8-phase kernel's G8_Q8 flavours only: a forced tile (AP_GEMM_NT_TILE), a rule (AP_GEMM_NT_RULE), AP_GEMM_NT_192 or
    // the fallback above may have picked a kernel that never writes it -- refused here, AFTER the choice, not silently skipped
    if (ep.q8 && variant != 20 && variant != 21) return AP_ERR_UNSUPPORTED;
    if (variant == 20 || variant == 21) {
        if (ep.gelu == 3) ep.gelu_tab = g8_gelu_table_ptr((hipStream_t)stream);
        return g8_launch(variant == 20 ? 192 : 256, A, lda, B, ldb, C, ldc, M, N, K, ep, (hipStream_t)stream);
    }
    const int lds_epi = rule_epi >= 0 ? rule_epi : (lds_epi_env >= 0 ? lds_epi_env : (variant != 2 && variant != 10));
#define NT_LAUNCH(TMv, TNv, WMv, WNv)                                                                        \
    {                                                                                                          \
        const int tm_ = (M + TMv - 1) / TMv, tn_ = (N + TNv - 1) / TNv, nt_ = tm_ * tn_;                       \
        if (lds_epi) hipLaunchKernelGGL((k_gemm_nt<TMv, TNv, WMv, WNv, false>), dim3(nt_), dim3(WMv * WNv * 64), 0, (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, tn_, nt_, ep); \
        else hipLaunchKernelGGL((k_gemm_nt<TMv, TNv, WMv, WNv, true>), dim3(nt_), dim3(WMv * WNv * 64), 0, (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, tn_, nt_, ep); \
    }
    switch (variant) {
        case 2: NT_LAUNCH(128, 64, 2, 2) break;
        case 3: NT_LAUNCH(64, 128, 2, 2) break;
        case 4: NT_LAUNCH(64, 64, 2, 2) break;
        case 5: NT_LAUNCH(256, 128, 4, 2) break;
        case 6: NT_LAUNCH(128, 128, 2, 4) break;
        case 7: NT_LAUNCH(256, 128, 2, 2) break;
        case 8: NT_LAUNCH(128, 256, 2, 2) break;
        case 9: NT_LAUNCH(256, 256, 2, 4) break;
        case 10: NT_LAUNCH(128, 192, 2, 2) break;
        case 11: NT_LAUNCH(64, 192, 2, 2) break;
        default:
            if (lds_epi && (ep.residual != nullptr) != (ep.dgelu_of != nullptr || ep.mul_by != nullptr) && !(ep.dbg & 2)) {    // epilogue reads ONE more tile: prefetching instantiation
                const int tm_ = (M + 127) / 128, tn_ = (N + 127) / 128, nt_ = tm_ * tn_;
                hipLaunchKernelGGL((k_gemm_nt<128, 128, 2, 2, false, 1, true>), dim3(nt_), dim3(256), 0, (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, tn_, nt_, ep);
            } else NT_LAUNCH(128, 128, 2, 2)
            break;
    }
#undef NT_LAUNCH
    return ap_check_launch();
}

int ap_gemm_nt_fp8(const unsigned char* A, int lda, const unsigned char* B, int ldb, ap_bf16* C, int ldc, int M, int N, int K,
                   const float* dq_a, const float* dq_b, const ap_gemm_epilogue* epi, ap_stream_t stream) {
    if (!A || !B || !C || !dq_a || !dq_b) return AP_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return AP_ERR_SHAPE;
    if ((K & 15) || (lda & 15) || (ldb & 15) || lda < K || ldb < K || ldc < N) return AP_ERR_SHAPE;       // 16-byte chunks of e4m3
    EpiArgs ep = {nullptr, 0, nullptr, nullptr, nullptr, 1, nullptr, 0, 0, nullptr, {0, 0, 0, 0, 0, 0u, 0u}, dq_a, dq_b, nullptr};
    if (epi) {
        ep.bias = epi->bias; ep.gelu = epi->gelu; ep.preact = epi->preact_out; ep.dgelu_of = epi->dgelu_of;
        ep.row_scale = epi->row_scale; ep.rows_per_scale = epi->rows_per_scale > 0 ? epi->rows_per_scale : 1;
        ep.residual = epi->residual; ep.ldr = epi->ldr;
        if (ep.residual && ep.ldr < N) return AP_ERR_SHAPE;
        if (ep.gelu < 0 || ep.gelu > 3 || (ep.gelu >= 2 && !ep.preact)) return AP_ERR_SHAPE;
        ep.q8 = epi->q8_out; ep.q8_scale = epi->q8_scale; ep.q8_amax = epi->q8_amax;
        if (ep.q8 && (!ep.q8_scale || !ep.gelu)) return AP_ERR_SHAPE;
    }
    (void)hipGetLastError();
    // byte pairs: the kernels' element is 2 bytes.  The persistent 8-phase kernel where the bf16 launch of these dimensions would take it
    // (K % 128 == 0 here: whole 128-byte K-tiles), the 128 x 128-tile kernel otherwise
    if ((K & 127) == 0 && !ep.dgelu_of) {
        const int bn8 = use_8p(M, N, K / 2, ldc, ep);
        if (bn8 && ep.gelu == 3) ep.gelu_tab = g8_gelu_table_ptr((hipStream_t)stream);
        if (bn8) return g8_launch(bn8, reinterpret_cast<const bf16_t*>(A), lda / 2, reinterpret_cast<const bf16_t*>(B), ldb / 2, C, ldc, M, N, K / 2, ep,
                                  (hipStream_t)stream, true);
    }
    if (ep.q8) return AP_ERR_UNSUPPORTED;            // the e4m3 side output exists in the 8-phase kernel's row phase only
    const int tm = (M + 127) / 128, tn = (N + 127) / 128, nt = tm * tn;
    hipLaunchKernelGGL((k_gemm_nt<128, 128, 2, 2, false, 1, false, 0, true>), dim3(nt), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const bf16_t*>(A), lda / 2, reinterpret_cast<const bf16_t*>(B), ldb / 2, C, ldc, M, N, K / 2, tn, nt, ep);
    return ap_check_launch();
}

static bool patch_map_device(const ap_patch_map* map, PatchMap& pm) {
    if (!map || map->group < 1 || map->kseg < 1) return false;
    pm.group = map->group; pm.gstride = map->group_stride; pm.rstride = map->row_stride; pm.kseg = map->kseg; pm.kstride = map->kseg_stride;
    pm.gmagic = map->group == 1 ? 0xFFFFFFFFu : (unsigned)(0xFFFFFFFFu / (unsigned)map->group) + 1u;      // group 1: q = m - (m != 0 ? 0 : 0) handled below
    pm.kmagic = map->kseg == 1 ? 0xFFFFFFFFu : (unsigned)(0xFFFFFFFFu / (unsigned)map->kseg) + 1u;
    return map->group > 1 && map->kseg > 1;         // d = 1 has no exact 32-bit magic; no caller needs it
}

int ap_gemm_nt_patch(const ap_bf16* A, const ap_bf16* B, int ldb, ap_bf16* C, int ld, int M, int N, int K,
                     const float* bias, const ap_patch_map* map, int side, ap_stream_t stream) {
    return ap_gemm_nt_patch_bn(A, nullptr, B, ldb, C, ld, M, N, K, bias, map, side, stream);
}

int ap_gemm_nt_patch_bn(const ap_bf16* A, const ap_bn_input* a_bn, const ap_bf16* B, int ldb, ap_bf16* C, int ld, int M, int N, int K,
                        const float* bias, const ap_patch_map* map, int side, ap_stream_t stream) {
    if (!A || !B || !C || !map) return AP_ERR_NULL;
    if (a_bn && (side != 1 || !a_bn->mean || !a_bn->rstd || !a_bn->gamma || !a_bn->beta)) return AP_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (ldb & 7) || ldb < K) return AP_ERR_SHAPE;
    EpiArgs ep = {nullptr, 0, nullptr, nullptr, nullptr, 1, nullptr, 0, 0, nullptr, {0, 0, 0, 0, 0, 0u, 0u}, nullptr, nullptr, nullptr};
    if (!patch_map_device(map, ep.pm)) return AP_ERR_SHAPE;
    if ((int64_t)M >= (int64_t)(0xFFFFFFFFu / (unsigned)map->group)) return AP_ERR_SHAPE;       // exactness bound of the magic division
    (void)hipGetLastError();
    if (side == 1) {            // A rows are patches; C plain [M, ld]
        if ((K & 63) || (map->kseg & 63) || K % map->kseg || ld < N) return AP_ERR_SHAPE;
        ep.bias = bias;
        if (a_bn) ep.abn = BnIn{a_bn->mean, a_bn->rstd, a_bn->gamma, a_bn->beta};       // (64-channel pixels: kseg % 64 == 0 is checked above)
        const int tm = (M + 127) / 128, tn = (N + 63) / 64, nt = tm * tn;
        hipLaunchKernelGGL((k_gemm_nt<128, 64, 2, 2, true, 1, false, 1>), dim3(nt), dim3(256), 0, (hipStream_t)stream, A, K, B, ldb, C, ld, M, N, K, tn, nt, ep);
    } else if (side == 2) {     // C rows are patches (input gradient); A plain [M, ld]
        if (bias || (N & 7) || (map->kseg & 7) || N % map->kseg || (ld & 7) || ld < K) return AP_ERR_SHAPE;
        const int tm = (M + 127) / 128, tn = (N + 127) / 128, nt = tm * tn;
        hipLaunchKernelGGL((k_gemm_nt<128, 128, 2, 2, false, 1, false, 2>), dim3(nt), dim3(256), 0, (hipStream_t)stream, A, ld, B, ldb, C, N, M, N, K, tn, nt, ep);
    } else return AP_ERR_SHAPE;
    return ap_check_launch();
}

int ap_gemm_tn_acc(const ap_bf16* A, int lda, const ap_bf16* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                   float* colsum_A, ap_stream_t stream) {
    if (!A || !B || !C) return AP_ERR_NULL;
    if (M <= 0 || N1 <= 0 || N2 <= 0) return AP_ERR_SHAPE;
    if ((lda & 7) || (ldb & 7) || lda < N1 || ldb < N2 || ldc < N2) return AP_ERR_SHAPE;
    const int t1 = (N1 + 127) / 128, t2 = (N2 + 127) / 128;
    int full_steps = 0;
    const int done = full_steps * TM;
    if (done < M) {                                   // token tail (or everything when the ring is disabled)
        const int Mt = M - done;
        const int total_steps = (Mt + TM - 1) / TM;
        static int target_blocks = 0;
        if (target_blocks == 0) { const char* e = getenv("AP_GEMM_TN_BLOCKS"); target_blocks = e ? atoi(e) : 384; /* measured optimum: atomics vs parallelism */ }
        int splits = (target_blocks + t1 * t2 - 1) / (t1 * t2);
        if (splits > total_steps / 4) splits = total_steps / 4;
        if (splits < 1) splits = 1;
        const int sps = (total_steps + splits - 1) / splits;
        splits = (total_steps + sps - 1) / sps;
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_gemm_tn, dim3(t1 * t2 * splits), dim3(256), 0, (hipStream_t)stream, A + (int64_t)done * lda, lda, B + (int64_t)done * ldb, ldb,
                           C, ldc, Mt, N1, N2, sps, colsum_A, t1, t2, t1 * t2 * splits);
        return ap_check_launch();
    }
    return AP_OK;
}

static int tn_grouped_128(const ap_tn_problem* problems, int count, const ap_ln_reduce* ln_items, int ln_count,
                          void* workspace, size_t ws_bytes, ap_stream_t stream);
// plan of a grouped launch: token splits per problem such that the launch fills the resident workgroup capacity once
static int tn_validate(const ap_tn_problem* problems, int count) {
    if (!problems) return AP_ERR_NULL;
    if (count <= 0 || count > AP_TN_MAX_GROUP) return AP_ERR_SHAPE;
    for (int i = 0; i < count; ++i) {
        const ap_tn_problem& q = problems[i];
        if (!q.A || !q.B || !q.C) return AP_ERR_NULL;
        if (q.M <= 0 || q.N1 <= 0 || q.N2 <= 0) return AP_ERR_SHAPE;
        if ((q.lda & 7) || q.lda < q.N1 || q.ldc < q.N2) return AP_ERR_SHAPE;
        if (!q.b_patch && ((q.ldb & 7) || q.ldb < q.N2)) return AP_ERR_SHAPE;         // patch-addressed B has no leading dimension
    }
    return AP_OK;
}
static int tn_plan(const ap_tn_problem* problems, int count, TnGroup& grp, int& total_blocks, size_t& slab_floats) {
    if (!problems) return AP_ERR_NULL;
    if (count <= 0 || count > TN_MAX_GROUP) return AP_ERR_SHAPE;
    // Two 256-thread workgroups are resident per CU: the launch should fill that single wave of workgroups as fully as
    // possible but never spill into a second one (measured on the D1 blocks: 450 workgroups 125 us, 540 -> 175 us).
    static int capacity = 0;
    if (capacity == 0) {
        const char* e = getenv("AP_GEMM_TN_GROUP_BLOCKS");
        if (e) capacity = atoi(e);
        else { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; capacity = 2 * ((hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256); }
    }
    int max_steps = 1;
    for (int i = 0; i < count; ++i) {
        const ap_tn_problem& q = problems[i];
        if (!q.A || !q.B || !q.C) return AP_ERR_NULL;
        if (q.M <= 0 || q.N1 <= 0 || q.N2 <= 0) return AP_ERR_SHAPE;
        if ((q.lda & 7) || q.lda < q.N1 || q.ldc < q.N2) return AP_ERR_SHAPE;
        if (!q.b_patch && ((q.ldb & 7) || q.ldb < q.N2)) return AP_ERR_SHAPE;         // patch-addressed B has no leading dimension
        const int steps = (q.M + TM - 1) / TM;
        if (steps > max_steps) max_steps = steps;
    }
    auto blocks_at = [&](int sps) {
        int64_t nb = 0;
        for (int i = 0; i < count; ++i) {
            const ap_tn_problem& q = problems[i];
            const int steps = (q.M + TM - 1) / TM;
            nb += (int64_t)((q.N1 + 127) / 128) * ((q.N2 + 127) / 128) * ((steps + sps - 1) / sps);
        }
        return nb;
    };
    // smallest steps-per-workgroup (>= 8) whose block count fits the resident capacity (binary search: blocks_at is monotone)
    int lo = 8, hi = max_steps > 8 ? max_steps : 8;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (blocks_at(mid) <= capacity) hi = mid; else lo = mid + 1;
    }
    const int sps_target = lo;
    grp.count = count;
    int start = 0;
    slab_floats = 0;
    for (int i = 0; i < count; ++i) {
        const ap_tn_problem& q = problems[i];
        TnArgs& a = grp.p[i];
        const int t1 = (q.N1 + 127) / 128, t2 = (q.N2 + 127) / 128, steps = (q.M + TM - 1) / TM;
        int splits = (steps + sps_target - 1) / sps_target;
        if (splits < 1) splits = 1;
        const int sps = (steps + splits - 1) / splits;
        splits = (steps + sps - 1) / sps;
        a.A = reinterpret_cast<const bf16_t*>(q.A); a.B = reinterpret_cast<const bf16_t*>(q.B); a.C = q.C; a.colsum = q.colsum_A;
        a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.M = q.M; a.N1 = q.N1; a.N2 = q.N2;
        a.steps_per_split = sps; a.t1 = t1; a.t2 = t2; a.nblocks = t1 * t2 * splits; a.start = start;
        a.cs_weight = reinterpret_cast<const bf16_t*>(q.colsum_weight);
        a.cs_scale = q.colsum_weight ? q.colsum_scale : 1.0f;
        a.alpha = q.alpha != 0.0f ? q.alpha : 1.0f;
        a.slab = nullptr; a.splits = splits;
        a.pb = PatchMap{0, 0, 0, 0, 0, 0u, 0u};
        if (q.b_patch) {
            if (!patch_map_device(q.b_patch, a.pb) || (q.N2 & 7) || (q.b_patch->kseg & 7) || q.N2 % q.b_patch->kseg ||
                (int64_t)q.M >= (int64_t)(0xFFFFFFFFu / (unsigned)q.b_patch->group)) return AP_ERR_SHAPE;
        }
        slab_floats += (size_t)splits * ((size_t)q.N1 * q.N2 + (q.colsum_A ? (size_t)q.N1 : 0));
        start += a.nblocks;
    }
    for (int i = count; i < TN_MAX_GROUP; ++i) { grp.p[i] = grp.p[0]; grp.p[i].start = 0x7fffffff; grp.p[i].nblocks = 0; }
    total_blocks = start;
    return AP_OK;
}

// One (problem, token split) = one XCD: greedy packing of the groups, largest first, onto the least loaded of the 8 XCDs; the
// launch is 8 * (largest load) workgroups, the unused ids return at once.  false = does not fit the table or the resident capacity
// (more than `cap` slots per XCD would start a second round of workgroups): the caller keeps the contiguous-range placement.
static bool tn_place(const TnGroup& grp, int cap, TnMap& map, int& blocks) {
    struct G { int p, s, n; };
    G g[TN_MAP_MAX]; int ng = 0;
    for (int i = 0; i < grp.count; ++i) {
        const int tiles = grp.p[i].t1 * grp.p[i].t2;
        if (tiles * grp.p[i].splits > 0x1000) return false;
        for (int s = 0; s < grp.p[i].splits; ++s) { if (ng == TN_MAP_MAX) return false; g[ng++] = G{i, s, tiles}; }
    }
    for (int i = 1; i < ng; ++i) {                       // insertion sort by size, descending, stable
        const G v = g[i]; int j = i - 1;
        while (j >= 0 && g[j].n < v.n) { g[j + 1] = g[j]; --j; }
        g[j + 1] = v;
    }
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < TN_MAP_MAX; ++i) map.e[i] = TN_MAP_IDLE;
    for (int i = 0; i < ng; ++i) {
        int x = 0;
        for (int k = 1; k < 8; ++k) if (load[k] < load[x]) x = k;
        if (load[x] + g[i].n > cap || (load[x] + g[i].n) * 8 > TN_MAP_MAX) return false;
        for (int t = 0; t < g[i].n; ++t) map.e[(load[x] + t) * 8 + x] = (unsigned short)((g[i].p << 12) | (g[i].s * g[i].n + t));
        load[x] += g[i].n;
    }
    int mx = 0;
    for (int k = 0; k < 8; ++k) if (load[k] > mx) mx = load[k];
    blocks = mx * 8;
    return true;
}

// Plan of the 192 x 192-tile kernel: every width a multiple of 192, whole 64-token K-tiles, plain row-major operands.  The token
// axis is cut into ranges of `ksps` K-tiles, the smallest that leaves at most one work item per CU and fits the per-XCD placement.
static bool tn8_fits(const ap_tn_problem& q) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("AP_GEMM_TN_8P"); enabled = e ? atoi(e) : 1; }
    if (!enabled || !q.A || !q.B || !q.C || q.b_patch || q.N1 <= 0 || q.N2 <= 0) return false;
    if (q.N1 % 192 || q.N2 % 192 || q.M % 64 || q.M < 4096 || (q.lda & 7) || (q.ldb & 7) || q.lda < q.N1 || q.ldb < q.N2 || q.ldc < q.N2) return false;
    return !(q.colsum_weight && (reinterpret_cast<uintptr_t>(q.colsum_weight) & 3));
}
// placement for the 8p kernel: as tn_place, but a (problem, token range) that does not fit the least loaded XCD any more is CUT over
// XCDs (consecutive tiles share their A column block): with six blocks in a launch the table is nearly full -- 240 of 256 slots
struct T8Plan { int tiles, splits; };
static bool t8_place(const T8Plan* pl, int count, int cap, T8Map& map, int& blocks) {
    struct G { int p, s, n; };
    G g[T8_MAP_MAX]; int ng = 0;
    for (int i = 0; i < count; ++i) {
        if (pl[i].tiles * pl[i].splits > (1 << T8_MAP_SHIFT)) return false;
        for (int s = 0; s < pl[i].splits; ++s) { if (ng == T8_MAP_MAX) return false; g[ng++] = G{i, s, pl[i].tiles}; }
    }
    for (int i = 1; i < ng; ++i) {                       // insertion sort by size, descending, stable
        const G v = g[i]; int j = i - 1;
        while (j >= 0 && g[j].n < v.n) { g[j + 1] = g[j]; --j; }
        g[j + 1] = v;
    }
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < T8_MAP_MAX; ++i) map.e[i] = TN_MAP_IDLE;
    for (int i = 0; i < ng; ++i) {
        int t = 0;
        while (t < g[i].n) {
            int x = 0;
            for (int k = 1; k < 8; ++k) if (load[k] < load[x]) x = k;
            int take = g[i].n - t;
            if (take > cap - load[x]) take = cap - load[x];
            if (take <= 0 || (load[x] + take) * 8 > T8_MAP_MAX) return false;
            for (int u = 0; u < take; ++u) map.e[(load[x] + u) * 8 + x] = (unsigned short)((g[i].p << T8_MAP_SHIFT) | (g[i].s * g[i].n + t + u));
            load[x] += take; t += take;
        }
    }
    int mx = 0;
    for (int k = 0; k < 8; ++k) if (load[k] > mx) mx = load[k];
    blocks = mx * 8;
    return true;
}
static bool tn8_plan(const ap_tn_problem* problems, int count, int n_cu, T8Group& g8, T8Map& map, int& mblocks) {
    if (count <= 0 || count > AP_TN_MAX_GROUP) return false;
    int64_t work = 0; int max_k = 1;
    for (int i = 0; i < count; ++i) {
        const ap_tn_problem& q = problems[i];
        if (!tn8_fits(q)) return false;
        work += (int64_t)(q.N1 / 192) * (q.N2 / 192) * (q.M / 64);
        if (q.M / 64 > max_k) max_k = q.M / 64;
    }
    int ksps = (int)((work + n_cu - 1) / n_cu);
    if (ksps < 8) ksps = 8;
    T8Plan pl[AP_TN_MAX_GROUP];
    for (; ksps <= max_k; ++ksps) {
        int64_t items = 0;
        for (int i = 0; i < count; ++i) {
            const ap_tn_problem& q = problems[i];
            const int ksteps = q.M / 64;
            pl[i].tiles = (q.N1 / 192) * (q.N2 / 192); pl[i].splits = (ksteps + ksps - 1) / ksps;
            items += (int64_t)pl[i].tiles * pl[i].splits;
        }
        if (items <= n_cu && t8_place(pl, count, n_cu / 8, map, mblocks)) break;
    }
    if (ksps > max_k) return false;
    for (int i = 0; i < count; ++i) {
        const ap_tn_problem& q = problems[i];
        T8Item& t = g8.p[i];
        t.A = reinterpret_cast<const bf16_t*>(q.A); t.B = reinterpret_cast<const bf16_t*>(q.B); t.C = q.C; t.colsum = q.colsum_A;
        t.cs_weight = reinterpret_cast<const bf16_t*>(q.colsum_weight);
        t.lda = q.lda; t.ldb = q.ldb; t.ldc = q.ldc; t.M = q.M; t.t2 = q.N2 / 192; t.ksteps = q.M / 64; t.ksps = ksps;
        t.alpha = q.alpha != 0.0f ? q.alpha : 1.0f; t.cs_scale = q.colsum_weight ? q.colsum_scale : 1.0f;
        g8.tiles[i] = (q.N1 / 192) * (q.N2 / 192);
        t.shared_out = 0;
        for (int j = 0; j < count; ++j)
            if (j != i && (problems[j].C == q.C || (q.colsum_A && problems[j].colsum_A == q.colsum_A))) t.shared_out = 1;
    }
    for (int i = count; i < AP_TN_MAX_GROUP; ++i) { g8.p[i] = g8.p[0]; g8.tiles[i] = 1; }
    return true;
}

size_t ap_gemm_tn_grouped_workspace(const ap_tn_problem* problems, int count) {
    TnGroup grp; int blocks = 0; size_t fl = 0;
    if (tn_plan(problems, count, grp, blocks, fl) != AP_OK) return 0;
    return fl * sizeof(float);
}

int ap_gemm_tn_acc_grouped(const ap_tn_problem* problems, int count, void* workspace, size_t ws_bytes, ap_stream_t stream) {
    return ap_gemm_tn_acc_grouped_ln(problems, count, nullptr, 0, workspace, ws_bytes, stream);
}

int ap_gemm_tn_acc_grouped_ln(const ap_tn_problem* problems, int count, const ap_ln_reduce* ln_items, int ln_count,
                              void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (ln_count < 0 || ln_count > AP_LN_MAX_BATCH || (ln_count > 0 && !ln_items)) return AP_ERR_SHAPE;
    for (int i = 0; i < ln_count; ++i) {
        if (!ln_items[i].partial || !ln_items[i].dgamma || !ln_items[i].dbeta) return AP_ERR_NULL;
        if (ln_items[i].n_partial <= 0 || ln_items[i].C <= 0) return AP_ERR_SHAPE;
    }
    {
        const int rcv = tn_validate(problems, count);
        if (rcv != AP_OK) return rcv;
    }
    if (!workspace) {
        // the 192 x 192-tile LDS-DMA kernel takes the problems that fit it (with the LayerNorm riders); the rest of the group -- e.g. the
        // 486-wide attention-weight projection of an outlooker block -- goes on in launches of the 128 x 128-tile kernel, 8 problems each
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr;
            n_cu = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
            (void)hipFuncSetAttribute((const void*)k_gemm_tn_8p, hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES);
            (void)hipGetLastError();
        }
        ap_tn_problem fit[AP_TN_MAX_GROUP], rest[AP_TN_MAX_GROUP];
        int nfit = 0, nrest = 0;
        for (int i = 0; i < count; ++i) { if (tn8_fits(problems[i])) fit[nfit++] = problems[i]; else rest[nrest++] = problems[i]; }
        T8Group g8; T8Map map8; int mb8 = 0;
        bool riders_done = false;
        if (nfit > 0 && tn8_plan(fit, nfit, n_cu, g8, map8, mb8)) {
            TnLn ln;
            int lblocks = 0;
            for (int i = 0; i < AP_LN_MAX_BATCH; ++i) {
                if (i < ln_count) {
                    const ap_ln_reduce& q = ln_items[i];
                    ln.it[i] = TnLnItem{q.partial, q.dgamma, q.dbeta, q.n_partial, q.C}; lblocks += (2 * q.C + 31) / 32;
                } else ln.it[i] = TnLnItem{nullptr, nullptr, nullptr, 0, 0};
            }
            ln.count = ln_count; ln.first = mb8;
            (void)hipGetLastError();
            hipLaunchKernelGGL(k_gemm_tn_8p, dim3(mb8 + lblocks), dim3(512), T8_LDS_BYTES, (hipStream_t)stream, g8, map8, ln);
            const int rc8 = ap_check_launch();
            if (rc8 != AP_OK) return rc8;
            riders_done = true;
        } else { for (int i = 0; i < nfit; ++i) rest[nrest++] = fit[i]; }
        if (nrest == 0 && riders_done) return AP_OK;
        if (nrest > TN_MAX_GROUP || (nrest > 0 && riders_done)) {
            for (int i0 = 0; i0 < nrest; i0 += TN_MAX_GROUP) {
                const int n = nrest - i0 < TN_MAX_GROUP ? nrest - i0 : TN_MAX_GROUP;
                const bool with_ln = !riders_done;
                const int rcc = tn_grouped_128(rest + i0, n, with_ln ? ln_items : nullptr, with_ln ? ln_count : 0, nullptr, 0, stream);
                if (rcc != AP_OK) return rcc;
                riders_done = true;
            }
            return AP_OK;
        }
    }
    return tn_grouped_128(problems, count, ln_items, ln_count, workspace, ws_bytes, stream);
}

// the 128 x 128-tile kernels: at most TN_MAX_GROUP problems, optional deterministic mode
static int tn_grouped_128(const ap_tn_problem* problems, int count, const ap_ln_reduce* ln_items, int ln_count,
                          void* workspace, size_t ws_bytes, ap_stream_t stream) {
    bool ln_done = ln_count == 0;
    TnGroup grp; int blocks = 0; size_t fl = 0;
    const int rc = tn_plan(problems, count, grp, blocks, fl);
    if (rc != AP_OK) return rc;
    if (workspace) {                                  // deterministic: stored partial tiles + ordered reduce instead of fp32 atomics
        if (ws_bytes < fl * sizeof(float)) return AP_ERR_SHAPE;
        float* w = reinterpret_cast<float*>(workspace);
        for (int i = 0; i < count; ++i) {
            grp.p[i].slab = w;
            w += (size_t)grp.p[i].splits * ((size_t)grp.p[i].N1 * grp.p[i].N2 + (grp.p[i].colsum ? (size_t)grp.p[i].N1 : 0));
        }
    }
    (void)hipGetLastError();
    bool any_patch = false;
    for (int i = 0; i < count; ++i) any_patch = any_patch || problems[i].b_patch != nullptr;
    if (any_patch) {
        if (count != 1) return AP_ERR_UNSUPPORTED;                  // a patch-addressed problem is launched on its own
        BnIn bbn = BnIn{nullptr, nullptr, nullptr, nullptr};
        if (problems[0].b_bn) {
            const ap_bn_input* q = problems[0].b_bn;
            if (!q->mean || !q->rstd || !q->gamma || !q->beta) return AP_ERR_NULL;
            if (problems[0].b_patch->kseg % 64 || problems[0].N2 % 64) return AP_ERR_SHAPE;         // 64-channel pixels, whole pixels per patch row
            bbn = BnIn{q->mean, q->rstd, q->gamma, q->beta};
        }
        if (problems[0].b_bn) hipLaunchKernelGGL(k_gemm_tn_patch<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp.p[0], bbn);
        else hipLaunchKernelGGL(k_gemm_tn_patch<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp.p[0], bbn);
    } else {
        static int place = -1, cap = 0;
        if (place < 0) {
            const char* e = getenv("AP_GEMM_TN_PLACE"); place = e ? atoi(e) : 1;
            int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr;
            cap = 2 * ((hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256) / 8;     // resident workgroups per XCD
        }
        TnMap map; int mblocks = 0;
        if (place && tn_place(grp, cap, map, mblocks)) {
            TnLn ln;
            int lblocks = 0;
            for (int i = 0; i < AP_LN_MAX_BATCH; ++i) {
                if (i < ln_count) {
                    const ap_ln_reduce& q = ln_items[i];
                    ln.it[i] = TnLnItem{q.partial, q.dgamma, q.dbeta, q.n_partial, q.C}; lblocks += (2 * q.C + 31) / 32;
                } else ln.it[i] = TnLnItem{nullptr, nullptr, nullptr, 0, 0};
            }
            ln.count = ln_count; ln.first = mblocks;
            hipLaunchKernelGGL(k_gemm_tn_grouped_map, dim3(mblocks + lblocks), dim3(256), 0, (hipStream_t)stream, grp, map, ln);
            ln_done = true;
        } else hipLaunchKernelGGL(k_gemm_tn_grouped, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    }
    if (!ln_done) {                                    // no placement table (or a patch-addressed problem): the reductions get their own launch
        const int rc2 = ap_layernorm_bwd_reduce_batched(ln_items, ln_count, stream);
        if (rc2 != AP_OK) return rc2;
    }
    if (workspace) hipLaunchKernelGGL(k_tn_reduce, dim3(1024), dim3(256), 0, (hipStream_t)stream, grp);
    return ap_check_launch();
}

}  // extern "C"
