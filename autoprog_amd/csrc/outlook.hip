// Outlook attention core (models/volo.py:83-98): unfold(3x3, pad 1, stride 2) -> softmax over the
// 9 source slots -> (9x9)@(9xhd) per window and head -> fold.  Closed form:
//   Y[t] = sum_{(w,p): src(w,p)=t} sum_q softmax_q(s*A[w,p,:])[q] * V[src(w,q)]
//   src(w=(i,j), slot) = (2i-1+slot/3, 2j-1+slot%3)
// written as a GATHER over output pixels so there is no atomics and no [B,9C,hw] unfold tensor:
// a 2x2 pixel quad (2i..2i+1, 2j..2j+1) receives exactly 9 (window, slot) contributions from the
// windows (i,j),(i,j+1),(i+1,j),(i+1,j+1).  HBM-bound: V + logits + Y, each touched once.
//
// Work decomposition: workgroup = (image, head, strip of window rows); the head's V strip (+halo, zero
// padded) is staged once in LDS (chunk-major bf16) and the strip's softmax matrices as fp32
// [window][9][9]; one lane per (output pixel, 8-channel chunk), pixels taken by parity class.  The same
// kernel run with TRANSPOSED probabilities on dY yields dV (the fold backward is an unfold of dY and vice
// versa).  dlogits uses one lane per (window, row p): dP = <dY[src p], V[src q]>,
// dA = s*P*(dP - sum_q P*dP).
#include "common.h"
#include <cstdlib>
#include <algorithm>

// acc += a.lo*b.lo + a.hi*b.hi on packed bf16 pairs, fp32 accumulate.  Inline asm: hipcc (ROCm 7.2) miscompiles
// __builtin_amdgcn_fdot2_f32_bf16 on elements of a 4 x u32 vector (every call reads element 0; tools/probe/dot2.hip)
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float acc) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
// hipcc's hazard recognizer does not look inside inline asm: a DOT result read by another VALU instruction needs 3 wait states on
// gfx940+ and gets none (seen in round 4: a v_pk_mul_f32 scheduled right behind the last v_dot2c of a chain read the accumulator
// one term short -- dlogits 4.5 % off).  Close every chain with this before its sum is used.
__device__ __forceinline__ void dot2_done(float& acc) { asm volatile("s_nop 2" : "+v"(acc)); }
#define OHD 32          // head dim handled by these kernels
#define OKK 9           // 3x3 slots
#define OPP 81

// stage pixel rows [y0, y0+rows) x cols [-1, pw-1) of one head (32 channels) into LDS, zero outside the image
// LDS layout is CHUNK-major: 16-B chunk c (8 channels) of pixel pix lives at (c*npix + pix)*16 B, so lanes
// that own neighbouring quads/windows (pixel stride 2) read 32 B apart (2-way bank conflict) instead of
// 128 B apart (8-way with a pixel-major [pix][32] image).
__device__ __forceinline__ void stage_patch(bf16_t* patch, const bf16_t* src, int H, int W, int C, int y0, int rows, int pw, int npix, int nthreads) {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int total = rows * pw * 4;
    constexpr int SU = 4;               // loads in flight per thread (one load -> one LDS store per iteration exposed a latency each)
    for (int idx0 = threadIdx.x; idx0 < total; idx0 += SU * nthreads) {
        u32x4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = idx0 + u * nthreads;
            const int c = idx & 3, pix = idx >> 2;
            const int pr = pix / pw, pc = pix - pr * pw;
            const int y = y0 + pr, x = pc - 1;
            const bool in = (idx < total) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
            v[u] = in ? ld16(src + ((int64_t)y * W + x) * C + c * 8) : zero4;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = idx0 + u * nthreads;
            if (idx < total) st16(patch + ((idx & 3) * npix + (idx >> 2)) * 8, v[u]);
        }
    }
}

// softmax rows of the windows [wy0, wy0+nwr) x [0,w) of one head -> fp32 P[win][9][9]
__device__ __forceinline__ void stage_probs(float* P, const bf16_t* logits, int ldl, int64_t win_base, int w, int nwin,
                                            int head, float scale, int nthreads) {
    // the 9 logits of up to PU rows per thread are loaded before the first softmax (rows are independent)
    constexpr int PU = 3;
    for (int r0 = threadIdx.x; r0 < nwin * OKK; r0 += PU * nthreads) {
        unsigned short raw[PU][OKK];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int r = r0 + u * nthreads;
            const int rc = min(r, nwin * OKK - 1);
            const int wl = rc / OKK, p = rc - wl * OKK;
            const bf16_t* a = logits + (win_base + wl) * ldl + head * OPP + p * OKK;
#pragma unroll
            for (int q = 0; q < OKK; ++q) raw[u][q] = a[q];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
        const int r = r0 + u * nthreads;
        if (r >= nwin * OKK) continue;
        float s[OKK];
        float mx = -3.0e38f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) { s[q] = bf2f(raw[u][q]) * scale; mx = fmaxf(mx, s[q]); }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) { s[q] = __expf(s[q] - mx); sum += s[q]; }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int q = 0; q < OKK; ++q) P[r * OKK + q] = s[q] * inv;
        }
    }
}

// chunk stride (in pixels) of the CHUNK-major patch image, padded so the four chunks of a pixel start 16 banks apart
__device__ __host__ __forceinline__ int pad_npix(int npix) { return npix + ((4 - (npix & 15)) & 15); }
// row stride (pixels) of the persistent kernels' LDS patch: the smallest value >= pw that is 3 (mod 4) -- see k_outlook_p
__device__ __host__ __forceinline__ int olk_pws(int pw) { return pw + ((3 - pw) & 3); }

#define OGT 256         // threads per workgroup of the two kernels below
// One lane per (output pixel, 8-channel chunk): 16x the lanes of the former one-lane-per-quad mapping (which left
// ~2 waves per SIMD on the whole chip with a 2600-FMA serial loop each).  Pixels are processed by PARITY CLASS
// (y&1, x&1): a class has a uniform number of contributing windows (1, 2, 2, 4), so a wave never diverges.
template <bool TP>
__global__ void __launch_bounds__(OGT)
k_outlook_gather(const bf16_t* __restrict__ in, const bf16_t* __restrict__ logits, int ldl, bf16_t* __restrict__ out,
                 int H, int W, int heads, float scale, int SR, int nstrips) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int h = (H + 1) >> 1, w = (W + 1) >> 1;
    const int C = heads * OHD;
    int bid = xcd_remap(blockIdx.x, gridDim.x);     // the heads of one (image, strip) share cache lines: same XCD
    const int head = bid % heads; bid /= heads;
    const int strip = bid % nstrips;
    const int b = bid / nstrips;
    const int I0 = strip * SR;
    const int nq = min(SR, h - I0);
    const int pw = 2 * w + 1, ph = 2 * nq + 3;
    const int y0 = 2 * I0 - 1;
    const int nwr = min(nq + 1, h - I0);
    const int npix = pad_npix((2 * SR + 3) * pw);
    bf16_t* patch = reinterpret_cast<bf16_t*>(smem_raw);
    float* P = reinterpret_cast<float*>(smem_raw + (size_t)npix * OHD * 2);
    stage_patch(patch, in + (int64_t)b * H * W * C + head * OHD, H, W, C, y0, ph, pw, npix, OGT);
    stage_probs(P, logits, ldl, ((int64_t)b * h + I0) * w, w, nwr * w, head, scale, OGT);
    __syncthreads();
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
        const int dy = cls >> 1, dx = cls & 1;
        // pixels (2i+dy, 2j+dx), i in [I0, I0+nq), j in [0, w), inside the image
        const int ncols = (W - dx + 1) >> 1;
        const int nrows = min(nq, (H - dy + 1) / 2 - I0);
        const int nro = dy ? 2 : 1, nco = dx ? 2 : 1;
        for (int item = threadIdx.x; item < nrows * ncols * 4; item += OGT) {
            const int c = item & 3, pq = item >> 2;
            const int qi = pq / ncols, j = pq - qi * ncols;
            const int i = I0 + qi;
            const int y = 2 * i + dy, x = 2 * j + dx;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.f;
            for (int ro = 0; ro < nro; ++ro) {
                const int wi = dy ? (ro ? i + 1 : i) : i;
                const int ar = dy ? (ro ? 0 : 2) : 1;
                if (wi >= h) continue;
                for (int co = 0; co < nco; ++co) {
                    const int wj = dx ? (co ? j + 1 : j) : j;
                    const int ac = dx ? (co ? 0 : 2) : 1;
                    if (wj >= w) continue;
                    const int a = ar * 3 + ac;
                    const float* Pw = P + ((wi - I0) * w + wj) * OPP;
                    const bf16_t* px0 = patch + (c * npix + (2 * wi - 1 - y0) * pw + 2 * wj) * 8;
#pragma unroll
                    for (int br = 0; br < 3; ++br) {
#pragma unroll
                        for (int bc = 0; bc < 3; ++bc) {
                            const int bs = br * 3 + bc;
                            const float wgt = TP ? Pw[bs * OKK + a] : Pw[a * OKK + bs];
                            float f[8];
                            unpack8(ld16(px0 + (br * pw + bc) * 8), f);
#pragma unroll
                            for (int k = 0; k < 8; ++k) acc[k] += wgt * f[k];
                        }
                    }
                }
            }
            st16(out + (((int64_t)b * H + y) * W + x) * C + head * OHD + c * 8, pack8(acc));
        }
    }
}

// dlogits: one lane per (window, row p of its 9x9 matrix): dP[q] = <dY[src p], V[src q]>, dA[p,:] = s*P[p,:]*(dP - <P[p,:], dP>)
__global__ void __launch_bounds__(OGT)
k_outlook_dlogits(const bf16_t* __restrict__ v, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ logits, int ldl,
                  bf16_t* __restrict__ dlogits, int H, int W, int heads, float scale, int SRW, int nstrips) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int h = (H + 1) >> 1, w = (W + 1) >> 1;
    const int C = heads * OHD;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = bid % heads; bid /= heads;
    const int strip = bid % nstrips;
    const int b = bid / nstrips;
    const int I0 = strip * SRW;
    const int nwr = min(SRW, h - I0);
    const int pw = 2 * w + 1, ph = 2 * nwr + 1;
    const int y0 = 2 * I0 - 1;
    const int npix = pad_npix((2 * SRW + 1) * pw);
    const size_t patch_bytes = (size_t)npix * OHD * 2;
    bf16_t* pv = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* pg = reinterpret_cast<bf16_t*>(smem_raw + patch_bytes);
    float* P = reinterpret_cast<float*>(smem_raw + 2 * patch_bytes);
    const int64_t img = (int64_t)b * H * W * C + head * OHD;
    stage_patch(pv, v + img, H, W, C, y0, ph, pw, npix, OGT);
    stage_patch(pg, dy + img, H, W, C, y0, ph, pw, npix, OGT);
    const int nwin = nwr * w;
    const int64_t win_base = ((int64_t)b * h + I0) * w;
    stage_probs(P, logits, ldl, win_base, w, nwin, head, scale, OGT);
    __syncthreads();
    for (int item = threadIdx.x; item < nwin * OKK; item += OGT) {
        const int wl = item / OKK, p = item - wl * OKK;
        const int wi = wl / w, wj = wl - wi * w;
        const int pr0 = 2 * wi, pc0 = 2 * wj;        // patch coords of slot (0,0): y = 2(I0+wi)-1 -> row 2wi
        float* Prow = P + wl * OPP + p * OKK;
        // <dY[src p], V[src q]> over 32 channels as 16 v_dot2c_f32_bf16 on the packed bf16 pairs (no unpacking)
        u32x4 g[4];
        const bf16_t* gp = pg + ((pr0 + p / 3) * pw + pc0 + p % 3) * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = ld16(gp + c * npix * 8);
        float dP[OKK], pr[OKK];
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) {
            const bf16_t* vp = pv + ((pr0 + q / 3) * pw + pc0 + q % 3) * 8;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 f = ld16(vp + c * npix * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) s = dot2_bf16(g[c][k], f[k], s);
            }
            dot2_done(s);
            dP[q] = s;
            pr[q] = Prow[q];
            dot += pr[q] * s;
        }
#pragma unroll
        for (int q = 0; q < OKK; ++q) Prow[q] = scale * pr[q] * (dP[q] - dot);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < nwin * OPP; r += OGT) {
        const int wl2 = r / OPP, e = r - wl2 * OPP;
        dlogits[(win_base + wl2) * ldl + head * OPP + e] = f2bf(P[r]);
    }
    if (head == 0) {                                 // zero the padding columns once per window
        const int padc = ldl - heads * OPP;
        for (int r = threadIdx.x; r < nwin * padc; r += OGT) {
            const int wl2 = r / padc, e = r - wl2 * padc;
            dlogits[(win_base + wl2) * ldl + heads * OPP + e] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ MFMA formulation
// The gather above spends 16 VALU operations per (contribution, source slot, 8 channels) -- it is VALU-bound (2 TB/s).
// Here the 9 x 9 by 9 x 32 product of every window runs on the matrix pipe and only the fold stays on the VALU:
//   Z_w[p][ch] = sum_q P_w[p][q] * V[src(w, q)][ch]      two v_mfma_f32_16x16x16_bf16 per window (9 of 16 rows / K slots used)
//   Y[t][ch]   = sum_{(w, p): src(w, p) = t} Z_w[p][ch]   1, 2, 2 or 4 terms per pixel (parity classes as above)
// P_w goes to LDS as bf16 rows of 16 (slots 9..15 zero); the V fragment (rows = channels, K = the window's 9 source pixels) is one
// ds_read_b64_tr_b16 per 16 channels from the pixel-major patch; Z_w (bf16, 9 x 32) overwrites P_w's slot -- the wave that owns the
// window reads P before it writes Z.  TP = true folds dY with the transposed probabilities (dV).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
#define OPZ 288                  // bf16 elements of a window's P / Z slot (9 x 32)
#define OPR 24                   // k_outlook_p: stride (bf16 elements) of the 16-element P rows inside a slot.  48 bytes: consecutive rows start 12 banks
                                 // apart, so the 16 rows a 16-byte store (or the 9 an 8-byte operand read) touches per cycle are disjoint; at 32 bytes
                                 // rows r and r + 8 shared their banks.  9 (+ 1 zero) rows x 48 bytes fit the 576-byte slot.
template <bool TP>
__global__ void __launch_bounds__(OGT)
k_outlook_gather_mfma(const bf16_t* __restrict__ in, const bf16_t* __restrict__ logits, int ldl, bf16_t* __restrict__ out,
                      int H, int W, int heads, float scale, int SR, int nstrips) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int h = (H + 1) >> 1, w = (W + 1) >> 1;
    const int C = heads * OHD;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = bid % heads; bid /= heads;
    const int strip = bid % nstrips;
    const int b = bid / nstrips;
    const int I0 = strip * SR;
    const int nq = min(SR, h - I0);
    const int pw = 2 * w + 1, ph = 2 * nq + 3;
    const int y0 = 2 * I0 - 1;
    const int nwr = min(nq + 1, h - I0);
    const int nwin = nwr * w;
    const int npix = (2 * SR + 3) * pw;
    bf16_t* patch = reinterpret_cast<bf16_t*>(smem_raw);                       // [npix][32], pixel-major
    bf16_t* PZ = patch + (size_t)npix * OHD;                                   // [nwin][9][32]: P rows (16 used per row) then Z
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // ONE exposed memory latency per workgroup: the logits of the first PU softmax rows of every thread are requested before
    // the patch, and consumed after it
    constexpr int PU = 3;
    const int64_t win_base = ((int64_t)b * h + I0) * w;
    const int nrows_p = nwin * OKK;
    unsigned short raw[PU][OKK];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int rc = min((int)threadIdx.x + u * OGT, nrows_p - 1);
        const int wl = rc / OKK, p = rc - wl * OKK;
        // the row's 9 bf16 (18 bytes at a 2-byte aligned address) as five aligned dwords instead of nine 2-byte gathers; the dword before
        // / after the row lies inside the logits matrix (ldl >= heads * 81 + pad, 16-byte aligned rows)
        const bf16_t* a = logits + (win_base + wl) * ldl + head * OPP + p * OKK;
        const int odd = (int)((reinterpret_cast<uintptr_t>(a) >> 1) & 1);
        const unsigned* a4 = reinterpret_cast<const unsigned*>(a - odd);
        unsigned dw[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) dw[k] = a4[k];
#pragma unroll
        for (int q = 0; q < OKK; ++q) {
            const int e = q + odd;                                   // halfword index within the 5 dwords (odd is uniform per row only)
            const unsigned lo = dw[q >> 1], hi = dw[(q + 1) >> 1];
            raw[u][q] = odd ? (unsigned short)((q & 1) ? (hi & 0xffffu) : (lo >> 16)) : (unsigned short)((q & 1) ? (lo >> 16) : (lo & 0xffffu));
            (void)e;
        }
    }
    {   // patch: rows [y0, y0+ph) x cols [-1, pw-1), zero outside the image; all loads of a thread first
        const bf16_t* src = in + (int64_t)b * H * W * C + head * OHD;
        const int total = ph * pw * 4;
        constexpr int SU = 5;
        for (int idx0 = threadIdx.x; idx0 < total; idx0 += SU * OGT) {
            u32x4 v[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int idx = idx0 + u * OGT;
                const int c = idx & 3, pix = idx >> 2;
                const int pr = pix / pw, pc = pix - pr * pw;
                const int y = y0 + pr, x = pc - 1;
                const bool ok = (idx < total) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
                v[u] = ok ? ld16(src + ((int64_t)y * W + x) * C + c * 8) : zero4;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int idx = idx0 + u * OGT;
                if (idx < total) st16(patch + (idx >> 2) * OHD + (idx & 3) * 8, v[u]);
            }
        }
    }
    // softmax rows -> bf16 P (TP: transposed) in rows of 16 (slots 9..15 zero), first 9 x 16 elements of the window's slot
    auto softmax_row = [&](int r, const unsigned short* lg) {
        const int wl = r / OKK, p = r - wl * OKK;
        float sv[OKK];
        float mx = -3.0e38f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) { sv[q] = bf2f(lg[q]) * scale; mx = fmaxf(mx, sv[q]); }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) { sv[q] = __expf(sv[q] - mx); sum += sv[q]; }
        const float inv = 1.0f / sum;
        bf16_t* slot = PZ + wl * OPZ;
        if (TP) {
#pragma unroll
            for (int q = 0; q < OKK; ++q) slot[q * 16 + p] = f2bf(sv[q] * inv);          // A[q][p] = P[p][q]
            if (p < 7) {
#pragma unroll
                for (int q = 0; q < OKK; ++q) slot[q * 16 + 9 + p] = 0;                     // K slots 9..15 of every row
            }
        } else {
            float o8[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) o8[q] = q < OKK ? sv[q] * inv : 0.f;
            st16(slot + p * 16, pack8(o8));
            st16(slot + p * 16 + 8, pack8(o8 + 8));
        }
    };
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int r = threadIdx.x + u * OGT;
        if (r < nrows_p) softmax_row(r, raw[u]);
    }
    for (int r = threadIdx.x + PU * OGT; r < nrows_p; r += OGT) {                           // taller strips than 3 rows per thread
        const int wl = r / OKK, p = r - wl * OKK;
        const bf16_t* a = logits + (win_base + wl) * ldl + head * OPP + p * OKK;
        unsigned short lg[OKK];
#pragma unroll
        for (int q = 0; q < OKK; ++q) lg[q] = a[q];
        softmax_row(r, lg);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4, q4 = fr >> 2, p4 = fr & 3;
    for (int wl = wave; wl < nwin; wl += OGT / 64) {
        const int wi = wl / w, wj = wl - wi * w;                      // window row within the strip, column
        bf16_t* slot = PZ + wl * OPZ;
        // second operand: rows = p (fr), K = q: 4 bf16 at q = 4 g .. 4 g + 3 (rows 9..15 read the slot's tail: their output columns are dropped)
        const s16x4_t pfrag = *reinterpret_cast<const s16x4_t*>(slot + min(fr, OKK - 1) * 16 + g * 4);
        // first operand: rows = channels, K = source pixels: lane (q4, p4) of group g addresses pixel src(w, 4 g + q4), channels 4 p4 .. + 3
        const int kq = min(4 * g + q4, OKK - 1);                      // K slots 9..15 meet zeros of P: any finite pixel will do
        const bf16_t* vpix = patch + ((2 * wi + kq / 3) * pw + 2 * wj + kq % 3) * OHD + p4 * 4;
        f32x4 z[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const s16x4_t vfrag = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vpix + t * 16));
            z[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vfrag, pfrag, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // lane (fr = p, g) holds channels 16 t + 4 g .. + 3 of Z_w[p]
        if (fr < OKK) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x2 pk;
                pk[0] = pack_bf2(z[t][0], z[t][1]); pk[1] = pack_bf2(z[t][2], z[t][3]);
                *reinterpret_cast<u32x2*>(slot + fr * OHD + 16 * t + 4 * g) = pk;
            }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
        const int dy = cls >> 1, dx = cls & 1;
        const int ncols = (W - dx + 1) >> 1;
        const int nrows = min(nq, (H - dy + 1) / 2 - I0);
        const int nro = dy ? 2 : 1, nco = dx ? 2 : 1;
        for (int item = threadIdx.x; item < nrows * ncols * 4; item += OGT) {
            const int c = item & 3, pq = item >> 2;
            const int qi = pq / ncols, j = pq - qi * ncols;
            const int i = I0 + qi;
            const int y = 2 * i + dy, x = 2 * j + dx;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.f;
            for (int ro = 0; ro < nro; ++ro) {
                const int wi = dy ? (ro ? i + 1 : i) : i;
                const int ar = dy ? (ro ? 0 : 2) : 1;
                if (wi >= h) continue;
                for (int co = 0; co < nco; ++co) {
                    const int wj = dx ? (co ? j + 1 : j) : j;
                    const int ac = dx ? (co ? 0 : 2) : 1;
                    if (wj >= w) continue;
                    float f[8];
                    unpack8(ld16(PZ + ((wi - I0) * w + wj) * OPZ + (ar * 3 + ac) * OHD + c * 8), f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += f[k];
                }
            }
            st16(out + (((int64_t)b * H + y) * W + x) * C + head * OHD + c * 8, pack8(acc));
        }
    }
}

// ------------------------------------------------------------------------------------------------ persistent kernels (round 4)
// The kernels above run one (image, strip, head) per workgroup: global loads -> LDS -> barrier -> MFMA -> barrier -> fold, three
// workgroups per CU, ~15 workgroups per CU one after the other -- every one of them exposes a memory latency, and every one of them
// derives its indices (five non-constant integer divisions per thread and phase) again.  Here a workgroup is PERSISTENT: it walks items
// it = first + k * grid, keeps the index tables of its threads in registers (they depend on the thread, not on the item), and requests
// item k+1's pixels and logits into registers right after it has dropped item k's into LDS -- the loads land under the MFMA and fold
// phases of item k behind LDS-only barriers (nobody waits for them or for the result stores).
//   BWD = false:  Y = fold(softmax(A) V)                                   (models/volo.py:83-98)
//   BWD = true :  dV = fold'(softmax(A)^T dY)  AND  dA = s P (dP - <P, dP>), dP = <dY[src p], V[src q]>  in ONE kernel: dY, the logits and
//                 the softmax are read / computed once for both (two launches before: 227 MB -> 165 MB of traffic per call)
// P goes to LDS row-major as bf16 rows of 16 (slots 9..15 zero) for both directions; the transposed operand of the backward fold is read
// with ds_read_b64_tr_b16 (rows 9..15 of the transposed matrix come from a zeroed row behind the nine).
#ifndef OLK_ABL
#define OLK_ABL 0      // timing-only ablations (tools/abl.sh outlook OLK_ABL n; results WRONG): 1 no MFMA phase, 2 no fold, 4 no output stores, 8 no
#endif                 // loads after the first item, 16 no softmax, 32 no dlogits rows, 64 no patch stores to LDS
#define OLK_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
struct OlkArgs {
    const bf16_t* v; const bf16_t* dy; const bf16_t* logits; bf16_t* out; bf16_t* dlogits;
    int ldl, B, H, W, heads, SR, nstrips, nitems, wp_shift;
    float scale;
};

template <int T, int NSU, int NPU, bool BWD>
__global__ void __launch_bounds__(T, (BWD ? 2 : 3) * T / 256)          // waves per SIMD: three (forward) / two (backward) workgroups per CU
k_outlook_p(OlkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int H = a.H, W = a.W, heads = a.heads, SR = a.SR, ldl = a.ldl;
    const int h = (H + 1) >> 1, w = (W + 1) >> 1, C = heads * OHD, pw = 2 * w + 1;
    const int PH = 2 * SR + 3, npix = PH * pw, NWR = SR + 1;
    // LDS row stride of the patch, in pixels: = 3 (mod 4).  A pixel row of one head is 64 bytes = 16 banks, so the bank group of pixel
    // (r, c) is (r * pws + c) mod 4 -- with pws = pw = 2 w + 1 = 29 the pixels (r, c + 1) and (r + 1, c) of a 3 x 3 window met in one
    // group (two- and three-way conflicts in every transposed operand read: 30 % of the forward's time in the r04 counters); with
    // pws = 3 (mod 4) the nine pixels of a window fall in groups 0 1 2 | 3 0 1 | 2 3 0: the four a 16-lane read touches are distinct
    const int pws = olk_pws(pw), dpw = pws - pw, npixl = PH * pws;
    const int npixc = pad_npix(npixl);
    bf16_t* const pm = reinterpret_cast<bf16_t*>(smem_raw);           // pixel-major patch [PH][pws][32]: V (forward) / dY (backward): the MFMA operand
    bf16_t* const pvc = pm + (size_t)npixl * OHD;                       // BWD: chunk-major V patch [4][npixc][8] (16-byte reads of neighbouring pixels)
    bf16_t* const PZ = BWD ? pvc + (size_t)npixc * OHD : pvc;           // [NWR * w][9][32]: P rows (16 used per row, + a zero row), then Z
    bf16_t* const dA = PZ + (size_t)NWR * w * OPZ;                      // BWD: [SR * w][81] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // ---- per-thread tables (functions of the thread, not of the item)
    int s_goff[NSU], s_pr[NSU];                  // patch chunk u of this thread: element offset from (row y0, column 0, this head); patch row (or -1)
    bool s_xok[NSU];
#pragma unroll
    for (int u = 0; u < NSU; ++u) {
        const int idx = tid + u * T;
        const int c = idx & 3, pix = idx >> 2;
        const int pr = pix / pw, pc = pix - pr * pw, x = pc - 1;
        s_pr[u] = idx < npix * 4 ? pr : -1;
        s_xok[u] = x >= 0 && x < W;
        s_goff[u] = (pr * W + x) * C + c * 8;
    }
    int r_loff[NPU], r_wl[NPU], r_p[NPU], r_pix[NPU];   // softmax row u: logits offset from the strip's first window, local window, row p, patch pixel of slot (0,0)
#pragma unroll
    for (int u = 0; u < NPU; ++u) {
        const int r = tid + u * T;
        const int wl = r / OKK, p = r - wl * OKK;
        const int wi = wl / w, wj = wl - wi * w;
        r_wl[u] = wl; r_p[u] = p;
        r_loff[u] = wl * ldl + p * OKK;
        r_pix[u] = (2 * wi) * pws + 2 * wj;
    }

    // ---- item walk: workgroups of one XCD (blockIdx % 8) take neighbouring items in every round, so the heads of an (image, strip)
    // -- which share cache lines -- meet in one L2
    const int G = gridDim.x;
    int item = xcd_remap(blockIdx.x, G);
    u32x4 fm[NSU], fv[BWD ? NSU : 1];
    unsigned lg[NPU][5];
    int I0, nq, nwr, head, b;
    auto decode = [&](int it, int& b_, int& head_, int& I0_, int& nq_, int& nwr_) {
        head_ = it % heads; it /= heads;
        const int strip = it % a.nstrips;
        b_ = it / a.nstrips;
        I0_ = strip * SR; nq_ = min(SR, h - I0_); nwr_ = min(nq_ + 1, h - I0_);
    };
    auto request = [&](int it) {                 // global -> registers
        int b_, head_, I0_, nq_, nwr_;
        decode(it, b_, head_, I0_, nq_, nwr_);
        const int y0 = 2 * I0_ - 1, ph = 2 * nq_ + 3;
        const int64_t base = ((int64_t)b_ * H + y0) * W * C + head_ * OHD;
#pragma unroll
        for (int u = 0; u < NSU; ++u) {
            const int y = y0 + s_pr[u];
            const bool ok = s_pr[u] >= 0 && s_pr[u] < ph && s_xok[u] && y >= 0 && y < H;
            fm[u] = ok ? ld16((BWD ? a.dy : a.v) + base + s_goff[u]) : zero4;
            if constexpr (BWD) fv[u] = ok ? ld16(a.v + base + s_goff[u]) : zero4;
        }
        const bf16_t* lbase = a.logits + ((int64_t)b_ * h + I0_) * w * ldl + head_ * OPP;
        const int nrows = nwr_ * w * OKK;
#pragma unroll
        for (int u = 0; u < NPU; ++u) {
            // the row's 9 bf16 (18 bytes, 2-byte aligned) as five aligned dwords; the halfword in front of / behind the row lies inside the matrix
            const int r = tid + u * T;
            const int off = r < nrows ? r_loff[u] : 0;
            const int odd = (head_ + (r < nrows ? r_p[u] : 0)) & 1;
            const unsigned* a4 = reinterpret_cast<const unsigned*>(lbase + off - odd);
#pragma unroll
            for (int k = 0; k < 5; ++k) lg[u][k] = a4[k];
        }
    };
    if (item < a.nitems) request(item);
    while (item < a.nitems) {
        decode(item, b, head, I0, nq, nwr);
        const int nwin = nwr * w;
        // ---- (A) registers -> LDS: patches; softmax rows -> P
#pragma unroll
        for (int u = 0; u < NSU; ++u) {
            const int idx = tid + u * T;
            if (s_pr[u] >= 0 && !(OLK_ABL & 64)) {
                const int lpix = (idx >> 2) + s_pr[u] * dpw;           // pixel index with the padded row stride
                st16(pm + (size_t)(lpix * 4 + (idx & 3)) * 8, fm[u]);
                if constexpr (BWD) st16(pvc + ((size_t)(idx & 3) * npixc + lpix) * 8, fv[u]);
            }
        }
        float prow[NPU][OKK];
#pragma unroll
        for (int u = 0; u < NPU; ++u) {
            if (OLK_ABL & 16) {
#pragma unroll
                for (int q = 0; q < OKK; ++q) prow[u][q] = __uint_as_float(lg[u][q >> 1]);
                continue;
            }
            const int r = tid + u * T;
            const int p = r_p[u];
            const unsigned sh = ((head + p) & 1) << 4;         // the row starts at the high half of its first dword
            unsigned al[5];
#pragma unroll
            for (int k = 0; k < 4; ++k) al[k] = __builtin_amdgcn_alignbit(lg[u][k + 1], lg[u][k], sh);
            al[4] = lg[u][4] >> sh;
            float sv[OKK];
#pragma unroll
            for (int q = 0; q < OKK; ++q) sv[q] = (q & 1) ? bf_hi(al[q >> 1]) : bf_lo(al[q >> 1]);
            float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(fmaxf(sv[6], sv[7]), sv[8])));
            // softmax(s x): exp2((x - max x) s log2 e)  (s > 0), one fma per element in front of the v_exp
            const float c2 = a.scale * 1.4426950408889634f, nm = -mx * c2;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < OKK; ++q) { sv[q] = __builtin_amdgcn_exp2f(fmaf(sv[q], c2, nm)); sum += sv[q]; }
            const float inv = __builtin_amdgcn_rcpf(sum);
            float o8[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) o8[q] = q < OKK ? sv[q] * inv : 0.f;
#pragma unroll
            for (int q = 0; q < OKK; ++q) prow[u][q] = o8[q];
            if (r < nwin * OKK) {
                bf16_t* slot = PZ + (size_t)r_wl[u] * OPZ;
                st16(slot + p * OPR, pack8(o8));
                st16(slot + p * OPR + 8, pack8(o8 + 8));
                if (BWD && p == 0) { st16(slot + 9 * OPR, zero4); st16(slot + 9 * OPR + 8, zero4); }      // the zero row of the transposed read
            }
        }
        const int next = item + G;
        if (next < a.nitems && !(OLK_ABL & 8)) request(next);
        OLK_BAR();
        // ---- (B) Z_w = P_w (or its transpose) x pixels of the window, on the matrix pipe; wave-uniform window walk, TWO windows per trip:
        // a window is one dependent chain (LDS reads -> MFMA pair -> pack -> LDS write, ~250 clocks) and a wave has 7 of them per item --
        // with the chains of two windows interleaved the reads of one run under the MFMAs and packs of the other
        {
            const int fr = lane & 15, g = lane >> 4, q4 = fr >> 2, p4 = fr & 3;
            constexpr int NWV = T / 64;
            const int kq = min(4 * g + q4, OKK - 1);
            const int kr = (kq * 11) >> 5;                            // kq / 3 for kq < 9
            // this lane's source pixel of the window, relative to the window's first; channel offset 8 p4 (+ 4 t): row 4 g + r of MFMA t is then
            // channel 8 g + 4 t + r, so a lane's two results are 8 CONSECUTIVE channels of its P row -- one 16-byte LDS write instead of two
            // 8-byte ones whose rows 0 / 4 / 8 met in the same banks (r04 counters: two thirds of the LDS pipe's cycles were conflicts)
            const int koff = (kr * pws + (kq - 3 * kr)) * OHD + p4 * 8;
            const int poff = BWD ? min(4 * g + q4, OKK) * OPR + p4 * 4 : min(fr, OKK - 1) * OPR + g * 4;
            int wi = 0, wj = wave;
            while (wj >= w) { wj -= w; ++wi; }
            const int nwin_e = (OLK_ABL & 1) ? 0 : nwin;
            for (int wl = wave; wl < nwin_e; wl += 2 * NWV) {
                int wi2 = wi, wj2 = wj + NWV;
                while (wj2 >= w) { wj2 -= w; ++wi2; }
                const bool two = wl + NWV < nwin_e;                   // wave-uniform
                bf16_t* slot0 = PZ + (size_t)wl * OPZ;
                bf16_t* slot1 = PZ + (size_t)(two ? wl + NWV : wl) * OPZ;
                const bf16_t* vp0 = pm + ((2 * wi) * pws + 2 * wj) * OHD + koff;
                const bf16_t* vp1 = pm + ((2 * (two ? wi2 : wi)) * pws + 2 * (two ? wj2 : wj)) * OHD + koff;
                s16x4_t pf0, pf1;
                if constexpr (BWD) {
                    pf0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(slot0 + poff));
                    pf1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(slot1 + poff));
                } else {
                    pf0 = *reinterpret_cast<const s16x4_t*>(slot0 + poff);
                    pf1 = *reinterpret_cast<const s16x4_t*>(slot1 + poff);
                }
                s16x4_t vf0[2], vf1[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    vf0[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vp0 + t * 4));
                    vf1[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vp1 + t * 4));
                }
                f32x4 z0[2], z1[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    z0[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf0[t], pf0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    z1[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf1[t], pf1, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
                if (fr < OKK) {
                    // Z row fr, channels 8 g .. 8 g + 7, at chunk g ^ (fr >> 2): the rows 0, 4, 8 (64 bytes each: the same 16 banks) take different chunks
                    const int zoff = fr * OHD + ((g ^ (fr >> 2)) << 3);
                    u32x4 pk;
                    pk[0] = pack_bf2(z0[0][0], z0[0][1]); pk[1] = pack_bf2(z0[0][2], z0[0][3]);
                    pk[2] = pack_bf2(z0[1][0], z0[1][1]); pk[3] = pack_bf2(z0[1][2], z0[1][3]);
                    st16(slot0 + zoff, pk);
                    if (two) {
                        pk[0] = pack_bf2(z1[0][0], z1[0][1]); pk[1] = pack_bf2(z1[0][2], z1[0][3]);
                        pk[2] = pack_bf2(z1[1][0], z1[1][1]); pk[3] = pack_bf2(z1[1][2], z1[1][3]);
                        st16(slot1 + zoff, pk);
                    }
                }
                wi = wi2; wj = wj2 + NWV;
                while (wj >= w) { wj -= w; ++wi; }
            }
        }
        if constexpr (BWD) {
            // dA rows of the windows this strip owns: one lane per (window, row p) -- the lane that holds the row's probabilities
#pragma unroll
            for (int u = 0; u < NPU; ++u) {
                const int r = tid + u * T;
                if (r < nq * w * OKK && !(OLK_ABL & 32)) {
                    const int p = r_p[u];
                    const int pr3 = (p * 11) >> 5, pc3 = p - 3 * pr3;
                    u32x4 gch[4];
                    const bf16_t* gp = pm + (size_t)(r_pix[u] + pr3 * pws + pc3) * OHD;
#pragma unroll
                    for (int c = 0; c < 4; ++c) gch[c] = ld16(gp + c * 8);
                    float dP[OKK];
                    float dot = 0.f;
#pragma unroll
                    for (int q = 0; q < OKK; ++q) {
                        const bf16_t* vp = pvc + (size_t)(r_pix[u] + (q / 3) * pws + (q % 3)) * 8;
                        float sacc = 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const u32x4 f = ld16(vp + (size_t)c * npixc * 8);
#pragma unroll
                            for (int k = 0; k < 4; ++k) sacc = dot2_bf16(gch[c][k], f[k], sacc);
                        }
                        dot2_done(sacc);
                        dP[q] = sacc;
                        dot += prow[u][q] * sacc;
                    }
                    bf16_t* drow = dA + (size_t)r_wl[u] * OPP + p * OKK;
#pragma unroll
                    for (int q = 0; q < OKK; ++q) drow[q] = f2bf(a.scale * prow[u][q] * (dP[q] - dot));
                }
            }
        }
        OLK_BAR();
        // ---- (C) fold: one lane per (output pixel, 8-channel chunk); a group of 16 lanes-pixels shares (row, column parity), so no wave diverges
        {
            const int wp = 1 << a.wp_shift;
            const int nfold = 4 * nq * wp * 4;                     // 2 nq pixel rows x 2 column parities x wp columns x 4 chunks
            for (int id = tid; id < ((OLK_ABL & 2) ? 0 : nfold); id += T) {
                const int c = id & 3, pid = id >> 2;
                const int jj = pid & (wp - 1), t = pid >> a.wp_shift;
                const int dx = t & 1, yy = t >> 1;
                const int dy = yy & 1, wi0 = yy >> 1;
                const int y = 2 * I0 + yy, x = 2 * jj + dx;
                if (x >= W || y >= H) continue;
                float acc[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = 0.f;
                auto add = [&](int wi, int wj, int slot9) {
                    float f[8];
                    unpack8(ld16(PZ + ((size_t)(wi * w + wj)) * OPZ + slot9 * OHD + ((c ^ (slot9 >> 2)) << 3)), f);      // (chunk swizzle of the Z rows: phase B)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += f[k];
                };
                const bool row2 = dy && (I0 + wi0 + 1 < h), col2 = dx && (jj + 1 < w);
                const int ar = dy ? 2 : 1, ac = dx ? 2 : 1;
                add(wi0, jj, ar * 3 + ac);
                if (col2) add(wi0, jj + 1, ar * 3);
                if (row2) {
                    add(wi0 + 1, jj, ac);
                    if (col2) add(wi0 + 1, jj + 1, 0);
                }
                if (!(OLK_ABL & 4)) st16(a.out + (((int64_t)b * H + y) * W + x) * C + head * OHD + c * 8, pack8(acc));
                else asm volatile("" :: "v"(acc[0]), "v"(acc[7]));
            }
        }
        if constexpr (BWD) {
            const int64_t win_base = ((int64_t)b * h + I0) * w;
            const int nown = nq * w;
            for (int r = tid; r < nown * OPP; r += T) {
                const int wl2 = r / OPP, e = r - wl2 * OPP;
                a.dlogits[(win_base + wl2) * ldl + head * OPP + e] = dA[r];
            }
            if (head == 0) {                                 // zero the padding columns once per window
                const int padc = ldl - heads * OPP;
                for (int r = tid; r < nown * padc; r += T) {
                    const int wl2 = r / padc, e = r - wl2 * padc;
                    a.dlogits[(win_base + wl2) * ldl + heads * OPP + e] = 0;
                }
            }
        }
        OLK_BAR();
        item = next;
    }
}

// geometry of a persistent launch: the tallest strip (<= 3 window rows) whose tables fit the instantiation and whose LDS image leaves
// `wgs` workgroups per CU; -> SR (0: this shape stays on the one-item-per-workgroup kernels)
static int olk_pick(int H, int W, bool bwd, int T, int NSU, int NPU, size_t& lds) {
    const int h = (H + 1) / 2, w = (W + 1) / 2, pw = 2 * w + 1;
    if (w > 32) return 0;
    static int sr_env = -1;
    if (sr_env < 0) { const char* e = getenv("AP_OUTLOOK_SR"); sr_env = e ? atoi(e) : 0; }
    for (int SR = std::min(sr_env > 0 ? sr_env : 3, h); SR >= 1; --SR) {
        const int npix = (2 * SR + 3) * pw;
        if (npix * 4 > NSU * T || (SR + 1) * w * OKK > NPU * T) continue;
        const int npixl = (2 * SR + 3) * olk_pws(pw);
        lds = (size_t)npixl * OHD * 2 + (size_t)(SR + 1) * w * OPZ * 2;
        if (bwd) lds += (size_t)pad_npix(npixl) * OHD * 2 + (((size_t)SR * w * OPP * 2 + 15) & ~(size_t)15);
        if (lds <= 80 * 1024) return SR;
    }
    return 0;
}

template <int T, int NSU, int NPU, bool BWD>
static int olk_launch(const OlkArgs& a0, int SR, size_t lds, hipStream_t s) {
    OlkArgs a = a0;
    const int h = (a.H + 1) / 2, w = (a.W + 1) / 2;
    a.SR = SR; a.nstrips = (h + SR - 1) / SR; a.nitems = a.B * a.nstrips * a.heads;
    a.wp_shift = w <= 16 ? 4 : 5;
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); ncu = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_outlook_p<T, NSU, NPU, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; (void)hipGetLastError(); }
    const int per_cu = std::max(1, std::min((int)(160 * 1024 / lds), 2048 / T));
    int grid = std::min(a.nitems, ncu * per_cu);
    grid = std::max(8, grid & ~7);                      // a multiple of 8: the XCD-contiguous item walk
    if (grid > a.nitems) grid = a.nitems;
    hipLaunchKernelGGL((k_outlook_p<T, NSU, NPU, BWD>), dim3((unsigned)grid), dim3(T), lds, s, a);
    return ap_check_launch();
}

static int gather_launch(bool tp, const bf16_t* in, const bf16_t* logits, int ldl, bf16_t* out, int B, int H, int W, int heads,
                         float scale, hipStream_t s) {
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    const int pw = 2 * w + 1;
    // strips of SR window rows: a strip re-stages 3 halo pixel rows and one halo window row, so taller strips waste
    // less; 4 rows keep ~3000 workgroups x 256 lanes in flight at 28x28 / B = 128 and 3 workgroups per CU (LDS)
    auto lds_of = [&](int sr) { return (size_t)pad_npix((2 * sr + 3) * pw) * OHD * 2 + (size_t)(sr + 1) * w * OPP * 4; };
    static int sr_env = -1;
    if (sr_env < 0) { const char* e = getenv("AP_OUTLOOK_SR"); sr_env = e ? atoi(e) : 0; }
    int SR = sr_env > 0 ? sr_env : 4; if (SR > h) SR = h;
    while (SR > 1 && lds_of(SR) > 78 * 1024) --SR;
    if (lds_of(SR) > 160 * 1024) return AP_ERR_UNSUPPORTED;
    const int nstrips = (h + SR - 1) / SR;
    const dim3 grid((unsigned)(B * nstrips * heads));
    (void)hipGetLastError();
    static int use_mfma = -1;
    if (use_mfma < 0) { const char* e = getenv("AP_OUTLOOK_MFMA"); use_mfma = e ? atoi(e) : 1; }
    if (use_mfma) {
        auto lds_m = [&](int sr) { return (size_t)(2 * sr + 3) * pw * OHD * 2 + (size_t)(sr + 1) * w * OPZ * 2; };
        int SRm = sr_env > 0 ? sr_env : 3; if (SRm > h) SRm = h;       // 3 window rows: 50.4 us forward at 28 x 28 / B = 128 (4: 58.6, 2: 50.3)
        while (SRm > 1 && lds_m(SRm) > 78 * 1024) --SRm;
        if (lds_m(SRm) <= 160 * 1024) {
            const int ns = (h + SRm - 1) / SRm;
            const dim3 gm((unsigned)(B * ns * heads));
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute((const void*)k_outlook_gather_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)k_outlook_gather_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_done = true; (void)hipGetLastError();
            }
            if (tp) hipLaunchKernelGGL(k_outlook_gather_mfma<true>, gm, dim3(OGT), lds_m(SRm), s, in, logits, ldl, out, H, W, heads, scale, SRm, ns);
            else hipLaunchKernelGGL(k_outlook_gather_mfma<false>, gm, dim3(OGT), lds_m(SRm), s, in, logits, ldl, out, H, W, heads, scale, SRm, ns);
            return ap_check_launch();
        }
    }
    if (tp) hipLaunchKernelGGL(k_outlook_gather<true>, grid, dim3(OGT), lds_of(SR), s, in, logits, ldl, out, H, W, heads, scale, SR, nstrips);
    else hipLaunchKernelGGL(k_outlook_gather<false>, grid, dim3(OGT), lds_of(SR), s, in, logits, ldl, out, H, W, heads, scale, SR, nstrips);
    return ap_check_launch();
}

// AP_OUTLOOK_P: 1 (default) the persistent kernels with 512-thread workgroups (256 where a shape's tables do not fit 512 threads);
// 2 the 256-thread instantiations; 0 the one-item-per-workgroup kernels of rounds 1-3
static int olk_mode() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("AP_OUTLOOK_P"); m = e ? atoi(e) : 1; }
    return m;
}

extern "C" {

int ap_outlook_fwd(const ap_bf16* v, const ap_bf16* logits, int ldl, ap_bf16* y, int B, int H, int W, int heads, int hd,
                   float scale, ap_stream_t stream) {
    if (!v || !logits || !y) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || ldl < heads * OPP) return AP_ERR_SHAPE;
    if (hd != OHD) return AP_ERR_UNSUPPORTED;
    if (olk_mode()) {
        size_t lds = 0;
        OlkArgs a = {v, nullptr, logits, y, nullptr, ldl, B, H, W, heads, 0, 0, 0, 0, scale};
        if (olk_mode() == 1) { if (const int SR = olk_pick(H, W, false, 512, 3, 1, lds)) return olk_launch<512, 3, 1, false>(a, SR, lds, (hipStream_t)stream); }
        if (const int SR = olk_pick(H, W, false, 256, 5, 2, lds)) return olk_launch<256, 5, 2, false>(a, SR, lds, (hipStream_t)stream);
    }
    return gather_launch(false, v, logits, ldl, y, B, H, W, heads, scale, (hipStream_t)stream);
}

int ap_outlook_bwd(const ap_bf16* v, const ap_bf16* logits, int ldl, const ap_bf16* dy, ap_bf16* dv, ap_bf16* dlogits,
                   int B, int H, int W, int heads, int hd, float scale, ap_stream_t stream) {
    if (!v || !logits || !dy || !dv || !dlogits) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || ldl < heads * OPP) return AP_ERR_SHAPE;
    if (hd != OHD) return AP_ERR_UNSUPPORTED;
    if (olk_mode()) {
        size_t lds = 0;
        OlkArgs a = {v, dy, logits, dv, dlogits, ldl, B, H, W, heads, 0, 0, 0, 0, scale};
        if (olk_mode() == 1) { if (const int SR = olk_pick(H, W, true, 512, 3, 1, lds)) return olk_launch<512, 3, 1, true>(a, SR, lds, (hipStream_t)stream); }
        if (const int SR = olk_pick(H, W, true, 256, 5, 2, lds)) return olk_launch<256, 5, 2, true>(a, SR, lds, (hipStream_t)stream);
    }
    int rc = gather_launch(true, dy, logits, ldl, dv, B, H, W, heads, scale, (hipStream_t)stream);
    if (rc != AP_OK) return rc;
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    const int pw = 2 * w + 1;
    auto lds_of = [&](int sr) { return 2 * ((size_t)pad_npix((2 * sr + 1) * pw) * OHD * 2) + (size_t)sr * w * OPP * 4; };
    static int srw_env = -1;
    if (srw_env < 0) { const char* e = getenv("AP_OUTLOOK_SRW"); srw_env = e ? atoi(e) : 0; }
    int SRW = srw_env > 0 ? srw_env : 4; if (SRW > h) SRW = h;
    while (SRW > 1 && lds_of(SRW) > 78 * 1024) --SRW;
    if (lds_of(SRW) > 160 * 1024) return AP_ERR_UNSUPPORTED;
    const int nstrips = (h + SRW - 1) / SRW;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_outlook_dlogits, dim3((unsigned)(B * nstrips * heads)), dim3(OGT), lds_of(SRW), (hipStream_t)stream,
                       v, dy, logits, ldl, dlogits, H, W, heads, scale, SRW, nstrips);
    return ap_check_launch();
}

}  // extern "C"
