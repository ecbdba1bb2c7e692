// Outlook attention core (models/volo.py:83-98): unfold(3x3, pad 1, stride 2) -> softmax over the
// 9 source slots -> (9x9)@(9xhd) per window and head -> fold.  Closed form:
//   Y[t] = sum_{(w,p): src(w,p)=t} sum_q softmax_q(s*A[w,p,:])[q] * V[src(w,q)]
//   src(w=(i,j), slot) = (2i-1+slot/3, 2j-1+slot%3)
// written as a GATHER over output pixels so there is no atomics and no [B,9C,hw] unfold tensor:
// a 2x2 pixel quad (2i..2i+1, 2j..2j+1) receives exactly 9 (window, slot) contributions from the
// windows (i,j),(i,j+1),(i+1,j),(i+1,j+1).  HBM-bound: V + logits + Y, each touched once.
//
// Work decomposition: workgroup = (image, head, strip of quad rows); one lane per quad, 32 fp32
// accumulators per output pixel; the head's V strip (+halo, zero padded) is staged once in LDS as
// [rows][2w+1][32] bf16 and the strip's softmax matrices as fp32 [window][9][9] (row stride 81 dwords
// = odd -> conflict-free across lanes).  The same kernel run with TRANSPOSED probabilities on dY
// yields dV (the fold backward is an unfold of dY and vice versa).
// dlogits uses one lane per window: dP = <dY[src p], V[src q]>, dA = s*P*(dP - sum_q P*dP).
#include "common.h"

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
    for (int idx = threadIdx.x; idx < total; idx += nthreads) {
        const int c = idx & 3, pix = idx >> 2;
        const int pr = pix / pw, pc = pix - pr * pw;
        const int y = y0 + pr, x = pc - 1;
        const bool in = (y >= 0) & (y < H) & (x >= 0) & (x < W);
        const u32x4 v = in ? ld16(src + ((int64_t)y * W + x) * C + c * 8) : zero4;
        st16(patch + (c * npix + pix) * 8, v);
    }
}

// softmax rows of the windows [wy0, wy0+nwr) x [0,w) of one head -> fp32 P[win][9][9]
__device__ __forceinline__ void stage_probs(float* P, const bf16_t* logits, int ldl, int64_t win_base, int w, int nwin,
                                            int head, float scale, int nthreads) {
    for (int r = threadIdx.x; r < nwin * OKK; r += nthreads) {
        const int wl = r / OKK, p = r - wl * OKK;
        const bf16_t* a = logits + (win_base + wl) * ldl + head * OPP + p * OKK;
        float s[OKK];
        float mx = -3.0e38f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) { s[q] = bf2f(a[q]) * scale; mx = fmaxf(mx, s[q]); }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < OKK; ++q) { s[q] = __expf(s[q] - mx); sum += s[q]; }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int q = 0; q < OKK; ++q) P[r * OKK + q] = s[q] * inv;
    }
}

template <bool TP>
__global__ void __launch_bounds__(128)
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
    bf16_t* patch = reinterpret_cast<bf16_t*>(smem_raw);
    float* P = reinterpret_cast<float*>(smem_raw + (((size_t)(2 * SR + 3) * pw * OHD * 2 + 15) & ~(size_t)15));
    const int npix = (2 * SR + 3) * pw;
    stage_patch(patch, in + (int64_t)b * H * W * C + head * OHD, H, W, C, y0, ph, pw, npix, 128);
    stage_probs(P, logits, ldl, ((int64_t)b * h + I0) * w, w, nwr * w, head, scale, 128);
    __syncthreads();
    const int ql = threadIdx.x;
    if (ql >= nq * w) return;
    const int qi = ql / w, j = ql - qi * w;
    const int i = I0 + qi;
#pragma unroll 1
    for (int pix = 0; pix < 4; ++pix) {
        const int dy = pix >> 1, dx = pix & 1;
        const int y = 2 * i + dy, x = 2 * j + dx;
        if (y >= H || x >= W) continue;
        float acc[OHD];
#pragma unroll
        for (int k = 0; k < OHD; ++k) acc[k] = 0.f;
        // (window row, slot row) options for this pixel row; same for columns
        const int nro = dy ? 2 : 1, nco = dx ? 2 : 1;
#pragma unroll 1
        for (int ro = 0; ro < nro; ++ro) {
            const int wi = dy ? (ro ? i + 1 : i) : i;
            const int ar = dy ? (ro ? 0 : 2) : 1;
            if (wi >= h) continue;
#pragma unroll 1
            for (int co = 0; co < nco; ++co) {
                const int wj = dx ? (co ? j + 1 : j) : j;
                const int ac = dx ? (co ? 0 : 2) : 1;
                if (wj >= w) continue;
                const int a = ar * 3 + ac;
                const float* Pw = P + ((wi - I0) * w + wj) * OPP;
                const int pr0 = 2 * wi - 1 - y0, pc0 = 2 * wj;      // patch coords of the window's slot (0,0)
#pragma unroll 1
                for (int br = 0; br < 3; ++br) {
#pragma unroll
                    for (int bc = 0; bc < 3; ++bc) {
                        const int bs = br * 3 + bc;
                        const float wgt = TP ? Pw[bs * OKK + a] : Pw[a * OKK + bs];
                        const bf16_t* px = patch + ((pr0 + br) * pw + pc0 + bc) * 8;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float f[8];
                            unpack8(ld16(px + c * npix * 8), f);
#pragma unroll
                            for (int k = 0; k < 8; ++k) acc[c * 8 + k] += wgt * f[k];
                        }
                    }
                }
            }
        }
        bf16_t* op = out + (((int64_t)b * H + y) * W + x) * C + head * OHD;
#pragma unroll
        for (int c = 0; c < 4; ++c) st16(op + c * 8, pack8(acc + c * 8));
    }
}

__global__ void __launch_bounds__(64)
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
    const size_t patch_bytes = ((size_t)(2 * SRW + 1) * pw * OHD * 2 + 15) & ~(size_t)15;
    bf16_t* pv = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* pg = reinterpret_cast<bf16_t*>(smem_raw + patch_bytes);
    float* P = reinterpret_cast<float*>(smem_raw + 2 * patch_bytes);
    const int64_t img = (int64_t)b * H * W * C + head * OHD;
    const int npix = (2 * SRW + 1) * pw;
    stage_patch(pv, v + img, H, W, C, y0, ph, pw, npix, 64);
    stage_patch(pg, dy + img, H, W, C, y0, ph, pw, npix, 64);
    const int nwin = nwr * w;
    const int64_t win_base = ((int64_t)b * h + I0) * w;
    stage_probs(P, logits, ldl, win_base, w, nwin, head, scale, 64);
    __syncthreads();
    const int wl = threadIdx.x;
    if (wl < nwin) {
        const int wi = wl / w, wj = wl - wi * w;
        const int pr0 = 2 * wi, pc0 = 2 * wj;        // patch coords of slot (0,0): y = 2(I0+wi)-1 -> row 2wi
        float* Pw = P + wl * OPP;
#pragma unroll 1
        for (int p = 0; p < OKK; ++p) {
            float g[OHD];
            const bf16_t* gp = pg + ((pr0 + p / 3) * pw + pc0 + p % 3) * 8;
#pragma unroll
            for (int c = 0; c < 4; ++c) unpack8(ld16(gp + c * npix * 8), g + c * 8);
            float dP[OKK];
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < OKK; ++q) {
                const bf16_t* vp = pv + ((pr0 + q / 3) * pw + pc0 + q % 3) * 8;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float f[8];
                    unpack8(ld16(vp + c * npix * 8), f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += g[c * 8 + k] * f[k];
                }
                dP[q] = s;
                dot += Pw[p * OKK + q] * s;
            }
#pragma unroll
            for (int q = 0; q < OKK; ++q) Pw[p * OKK + q] = scale * Pw[p * OKK + q] * (dP[q] - dot);
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < nwin * OPP; r += 64) {
        const int wl2 = r / OPP, e = r - wl2 * OPP;
        dlogits[(win_base + wl2) * ldl + head * OPP + e] = f2bf(P[r]);
    }
    if (head == 0) {                                 // zero the padding columns once per window
        const int padc = ldl - heads * OPP;
        for (int r = threadIdx.x; r < nwin * padc; r += 64) {
            const int wl2 = r / padc, e = r - wl2 * padc;
            dlogits[(win_base + wl2) * ldl + heads * OPP + e] = 0;
        }
    }
}

static int gather_launch(bool tp, const bf16_t* in, const bf16_t* logits, int ldl, bf16_t* out, int B, int H, int W, int heads,
                         float scale, hipStream_t s) {
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    if (w > 128) return AP_ERR_UNSUPPORTED;
    int SR = 128 / w; if (SR > h) SR = h; if (SR < 1) SR = 1;
    const int pw = 2 * w + 1;
    // keep >= 2 workgroups per CU resident: shrink the strip until LDS <= 72 KB
    auto lds_of = [&](int sr) { return (((size_t)(2 * sr + 3) * pw * OHD * 2 + 15) & ~(size_t)15) + (size_t)(sr + 1) * w * OPP * 4; };
    while (SR > 1 && lds_of(SR) > 72 * 1024) --SR;
    if (lds_of(SR) > 160 * 1024) return AP_ERR_UNSUPPORTED;
    const int nstrips = (h + SR - 1) / SR;
    const dim3 grid((unsigned)(B * nstrips * heads));
    (void)hipGetLastError();
    if (tp) hipLaunchKernelGGL(k_outlook_gather<true>, grid, dim3(128), lds_of(SR), s, in, logits, ldl, out, H, W, heads, scale, SR, nstrips);
    else hipLaunchKernelGGL(k_outlook_gather<false>, grid, dim3(128), lds_of(SR), s, in, logits, ldl, out, H, W, heads, scale, SR, nstrips);
    return ap_check_launch();
}

extern "C" {

int ap_outlook_fwd(const ap_bf16* v, const ap_bf16* logits, int ldl, ap_bf16* y, int B, int H, int W, int heads, int hd,
                   float scale, ap_stream_t stream) {
    if (!v || !logits || !y) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || ldl < heads * OPP) return AP_ERR_SHAPE;
    if (hd != OHD) return AP_ERR_UNSUPPORTED;
    return gather_launch(false, v, logits, ldl, y, B, H, W, heads, scale, (hipStream_t)stream);
}

int ap_outlook_bwd(const ap_bf16* v, const ap_bf16* logits, int ldl, const ap_bf16* dy, ap_bf16* dv, ap_bf16* dlogits,
                   int B, int H, int W, int heads, int hd, float scale, ap_stream_t stream) {
    if (!v || !logits || !dy || !dv || !dlogits) return AP_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || ldl < heads * OPP) return AP_ERR_SHAPE;
    if (hd != OHD) return AP_ERR_UNSUPPORTED;
    int rc = gather_launch(true, dy, logits, ldl, dv, B, H, W, heads, scale, (hipStream_t)stream);
    if (rc != AP_OK) return rc;
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    if (w > 64) return AP_ERR_UNSUPPORTED;
    int SRW = 64 / w; if (SRW > h) SRW = h; if (SRW < 1) SRW = 1;
    const int pw = 2 * w + 1;
    auto lds_of = [&](int sr) { return 2 * (((size_t)(2 * sr + 1) * pw * OHD * 2 + 15) & ~(size_t)15) + (size_t)sr * w * OPP * 4; };
    while (SRW > 1 && lds_of(SRW) > 52 * 1024) --SRW;
    if (lds_of(SRW) > 160 * 1024) return AP_ERR_UNSUPPORTED;
    const int nstrips = (h + SRW - 1) / SRW;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_outlook_dlogits, dim3((unsigned)(B * nstrips * heads)), dim3(64), lds_of(SRW), (hipStream_t)stream,
                       v, dy, logits, ldl, dlogits, H, W, heads, scale, SRW, nstrips);
    return ap_check_launch();
}

}  // extern "C"
