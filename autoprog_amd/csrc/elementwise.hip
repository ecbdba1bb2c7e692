// HBM-bound helper kernels: precision casts, DropPath row scaling, broadcast add and its
// gradient, mix-token region swap, 2x2 ceil-mode average pooling, column sums (bias grads).
// All bf16 traffic is 16 bytes per lane (8 elements); grids are capped and grid-strided.
#include "common.h"

static inline int grid_for(int64_t work_items, int block = 256, int cap = 256 * 8) {
    int64_t g = ceil_div64(work_items, block);
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ------------------------------------------------------------------------------------ casts
__global__ void k_cast_f32_bf16(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
    const int64_t nv = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(s)[2 * i];
        const float4 b = reinterpret_cast<const float4*>(s)[2 * i + 1];
        u32x4 o;
        o[0] = pack_bf2(a.x, a.y); o[1] = pack_bf2(a.z, a.w); o[2] = pack_bf2(b.x, b.y); o[3] = pack_bf2(b.z, b.w);
        st16(d + 8 * i, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) d[(nv << 3) + threadIdx.x] = f2bf(s[(nv << 3) + threadIdx.x]);
}
__global__ void k_cast_bf16_f32(const bf16_t* __restrict__ s, float* __restrict__ d, int64_t n) {
    const int64_t nv = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        float f[8];
        unpack8(ld16(s + 8 * i), f);
        reinterpret_cast<float4*>(d)[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
        reinterpret_cast<float4*>(d)[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) d[(nv << 3) + threadIdx.x] = bf2f(s[(nv << 3) + threadIdx.x]);
}
// 32x32 tile transpose through LDS: dst[c][r] = src[r][c]
__global__ void k_cast_transpose(const float* __restrict__ s, bf16_t* __restrict__ d, int rows, int cols, int ld) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? s[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;          // dst row = c, dst col = r
        if (c < cols && r < ld) d[(int64_t)c * ld + r] = f2bf(tile[tx][i]);
    }
}

// ------------------------------------------------------------------------------ row scaling
__global__ void k_row_scale(const bf16_t* __restrict__ x, const float* __restrict__ sc, bf16_t* __restrict__ y,
                            int64_t M, int Cv, int rps) {
    const int64_t total = M * Cv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / Cv;
        const float s = sc[m / rps];
        float f[8];
        unpack8(ld16(x + 8 * i), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= s;
        st16(y + 8 * i, pack8(f));
    }
}
// out[oy, ox, c] (+)= sum_iy wy[oy, iy] * sum_ix wx[ox, ix] * in[iy, ix, c]: a separable resampling of a small fp32 NHWC grid by dense tap
// matrices -- the bicubic interpolation of the position embedding (reference models/volo.py:580-596: F.interpolate(mode="bicubic") of
// a [1, C, h, w] grid on EVERY forward at a resolution other than the model's) and, with the transposed matrices and acc = 1, its backward
// into the embedding's gradient.  A tap row has at most four non-zeros (clamped border taps coincide): zeros are skipped, so a thread
// does <= 16 multiply-adds.  (torch's upsample_bicubic2d and its backward took 80 us EACH on these 8 x 8 ... 14 x 14 grids: 10 % of an
// AutoProg stage-1 step.)
__global__ void __launch_bounds__(256)
k_resample_grid(const float* __restrict__ in, int hi, int wi, const float* __restrict__ wy, const float* __restrict__ wx,
                float* __restrict__ out, int ho, int wo, int C, int acc) {
    // one workgroup per output pixel, threads over the channels; the pixel's two tap rows go through LDS first (read from global memory
    // inside the zero-skipping loops they were a chain of dependent L2 round trips: 12.8 us per launch on an 8 x 8 x 384 grid)
    __shared__ float sy[64], sx[64];
    const int ox = blockIdx.x % wo, oy = blockIdx.x / wo;
    for (int i = threadIdx.x; i < hi; i += 256) sy[i] = wy[oy * hi + i];
    for (int i = threadIdx.x; i < wi; i += 256) sx[i] = wx[ox * wi + i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int iy = 0; iy < hi; ++iy) {
            const float a = sy[iy];
            if (a == 0.f) continue;
            float r = 0.f;
            for (int ix = 0; ix < wi; ++ix) {
                const float b = sx[ix];
                if (b != 0.f) r = fmaf(b, in[((int64_t)iy * wi + ix) * C + c], r);
            }
            s = fmaf(a, r, s);
        }
        const int64_t idx = ((int64_t)oy * wo + ox) * C + c;
        out[idx] = acc ? out[idx] + s : s;
    }
}
__global__ void k_add_bcast(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                            int64_t nv, int64_t bv) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        float fa[8], fb[8];
        unpack8(ld16(a + 8 * i), fa);
        unpack8(ld16(b + 8 * (i % bv)), fb);
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] += fb[j];
        st16(y + 8 * i, pack8(fa));
    }
}
// out[i] += sum_r x[r][i]: the repetitions are split over blockIdx.y (16 per workgroup, 8 loads in flight per thread) and the
// partial sums meet in fp32 atomics -- one thread walking all 128 repetitions serially took 61 us for 19 MB
#define SR_CHUNK 16
__global__ void k_sum_reps_acc(const bf16_t* __restrict__ x, float* __restrict__ out, int64_t nv, int reps) {
    const int r0 = blockIdx.y * SR_CHUNK, r1 = min(reps, r0 + SR_CHUNK);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = r0; r < r1; r += 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                v[u] = (r + u < r1) ? ld16(x + 8 * ((int64_t)(r + u) * nv + i)) : z;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(out + 8 * i + j, acc[j]);
    }
}

// ------------------------------------------------------------------------------- mix token
__global__ void k_mix_swap(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int Cv,
                           int r0, int r1, int c0, int c1, const int* __restrict__ box_dev = nullptr, int scale = 1) {
    if (box_dev) { r0 = box_dev[0] * scale; r1 = box_dev[1] * scale; c0 = box_dev[2] * scale; c1 = box_dev[3] * scale; }    // the step's box from device memory (graph replay)
    const int64_t total = (int64_t)B * H * W * Cv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t img = (int64_t)H * W * Cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t b = i / img, rem = i - b * img;
        const int pix = (int)(rem / Cv);
        const int r = pix / W, c = pix - r * W;
        const bool in = (r >= r0) & (r < r1) & (c >= c0) & (c < c1);
        const int64_t src = in ? ((int64_t)(B - 1 - b) * img + rem) : i;
        st16(y + 8 * i, ld16(x + 8 * src));
    }
}

// --------------------------------------------------------------------------------- pooling
__global__ void k_avgpool2_fwd(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int Cv) {
    const int h = (H + 1) >> 1, w = (W + 1) >> 1;
    const int64_t total = (int64_t)B * h * w * Cv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cv = (int)(i % Cv);
        int64_t t = i / Cv;
        const int j = (int)(t % w); t /= w;
        const int ii = (int)(t % h);
        const int64_t b = t / h;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = 2 * ii + dy, xx = 2 * j + dx;
                if (yy < H && xx < W) {
                    float f[8];
                    unpack8(ld16(x + 8 * (((b * H + yy) * W + xx) * Cv + cv)), f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += f[k];
                    ++cnt;
                }
            }
        const float inv = 1.0f / (float)cnt;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= inv;
        st16(y + 8 * i, pack8(acc));
    }
}
__global__ void k_avgpool2_bwd_acc(const bf16_t* __restrict__ dp, bf16_t* __restrict__ dx, int B, int H, int W, int Cv) {
    const int h = (H + 1) >> 1, w = (W + 1) >> 1;
    const int64_t total = (int64_t)B * H * W * Cv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cv = (int)(i % Cv);
        int64_t t = i / Cv;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int64_t b = t / H;
        const int ii = yy >> 1, j = xx >> 1;
        const int cnt = ((2 * ii + 1 < H) ? 2 : 1) * ((2 * j + 1 < W) ? 2 : 1);
        const float inv = 1.0f / (float)cnt;
        float g[8], f[8];
        unpack8(ld16(dp + 8 * (((b * h + ii) * w + j) * Cv + cv)), g);
        unpack8(ld16(dx + 8 * i), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += g[k] * inv;
        st16(dx + 8 * i, pack8(f));
    }
}

// --------------------------------------------------------------------------------- colsum
// out[n] += sum_m A[m,n].  Block = 256 threads = 32 column-chunks(8 cols) x 8 row lanes.
__global__ void k_colsum_acc(const bf16_t* __restrict__ A, int lda, float* __restrict__ out, int M, int N, int rows_per_block) {
    __shared__ float red[8][256 + 8];
    const int cchunk = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n0 = (blockIdx.x * 32 + cchunk) * 8;
    const int m_begin = blockIdx.y * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n0 < N) {
        for (int m = m_begin + rl; m < m_end; m += 8) {
            float f[8];
            unpack8(ld16(A + (int64_t)m * lda + n0), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += f[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl][cchunk * 8 + k] = acc[k];
    __syncthreads();
    const int col = threadIdx.x;          // 256 columns per block
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][col];
    const int n = blockIdx.x * 256 + col;
    if (n < N) atomicAdd(out + n, s);
}

// ===================================================================================== C ABI
// ------------------------------------------------------------- input resize (main_prog.py:973,1908)
// F.interpolate(input, size=(r, r), mode='bilinear', align_corners=False) of the fp32 NCHW batch, fused with the stem's layout
// change and bf16 cast: one pass writes the NHWC bf16 image the stem convolution reads.  PyTorch's source-index rule:
// src = max(0, (dst + 0.5) * in/out - 0.5), i0 = floor(src), i1 = i0 + (i0 < in - 1), weight of i1 = src - i0.
__global__ void __launch_bounds__(256)
k_resize_bilinear_nhwc(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int C, int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int64_t total = (int64_t)B * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const int64_t t = i / Wo;
        const int oy = (int)(t % Ho), b = (int)(t / Ho);
        const float fy = fmaxf(((float)oy + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sw - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hi - 1), x1 = x0 + (x0 < Wi - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = x + (int64_t)b * C * Hi * Wi;
        bf16_t* o = y + i * C;
        for (int c = 0; c < C; ++c) {
            const float* pc = p + (int64_t)c * Hi * Wi;
            const float top = (1.f - lx) * pc[(int64_t)y0 * Wi + x0] + lx * pc[(int64_t)y0 * Wi + x1];
            const float bot = (1.f - lx) * pc[(int64_t)y1 * Wi + x0] + lx * pc[(int64_t)y1 * Wi + x1];
            o[c] = f2bf((1.f - ly) * top + ly * bot);
        }
    }
}

// the same resize, written in the space-to-depth layout the 7x7 / stride 2 stem convolution reads (csrc/conv7.hip):
// y[b][oy >> 1][ox >> 1][((oy & 1) * 2 + (ox & 1)) * 3 + c], 16 bf16 per 2x2 block (12 used, 4 zero); C = 3, Ho and Wo even
__global__ void __launch_bounds__(256)
k_resize_bilinear_s2d16(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int64_t total = (int64_t)B * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const int64_t t = i / Wo;
        const int oy = (int)(t % Ho), b = (int)(t / Ho);
        const float fy = fmaxf(((float)oy + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sw - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hi - 1), x1 = x0 + (x0 < Wi - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = x + (int64_t)b * 3 * Hi * Wi;
        const int sub = (oy & 1) * 2 + (ox & 1);
        bf16_t* o = y + ((((int64_t)b * (Ho >> 1) + (oy >> 1)) * (Wo >> 1) + (ox >> 1)) << 4) + sub * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pc = p + (int64_t)c * Hi * Wi;
            const float top = (1.f - lx) * pc[(int64_t)y0 * Wi + x0] + lx * pc[(int64_t)y0 * Wi + x1];
            const float bot = (1.f - lx) * pc[(int64_t)y1 * Wi + x0] + lx * pc[(int64_t)y1 * Wi + x1];
            o[c] = f2bf((1.f - ly) * top + ly * bot);
        }
        if (sub == 3) { o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0; }           // channels 12..15 of the block
    }
}

// ---------------------------------------------------------------------------- fp8 (OCP e4m3) quantisation
// y = sat(x * scale[0]) as e4m3 (|.| <= 448), 16 values per thread; amax[0] = max(amax[0], max |x|) (fp32 bits compare as ints)
__global__ void __launch_bounds__(256)
k_quantize_fp8(const bf16_t* __restrict__ x, unsigned char* __restrict__ y, int64_t n16, const float* __restrict__ scale, float* __restrict__ amax) {
    const float sc = scale[0];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        float f[16];
        unpack8(ld16(x + i * 16), f);
        unpack8(ld16(x + i * 16 + 8), f + 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { mx = fmaxf(mx, fabsf(f[4 * k + e])); v[e] = fminf(fmaxf(f[4 * k + e] * sc, -448.f), 448.f); }
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
            o[k] = (unsigned)w;
        }
        st16(y + i * 16, o);
    }
    if (amax) {
        // one atomic per workgroup, and only when it would raise the value: a wave-level atomicMax from every wave of 2048 workgroups
        // onto one address took 60 of the kernel's 70 us (same-address atomics retire one after the other)
        __shared__ float smx[4];
        mx = group_max<64>(mx);
        if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
            if (__float_as_int(m) > *reinterpret_cast<volatile int*>(amax)) atomicMax(reinterpret_cast<int*>(amax), __float_as_int(m));
        }
    }
}

// several tensors in one launch (blockIdx.y = tensor): the Linear weights of a model, re-quantised once per optimizer step -- 144 launches
// of 4 us of work each for VOLO-D5 otherwise.  scale / amax of job j: scales[jobs[j].slot], amax[jobs[j].slot].
__global__ void __launch_bounds__(256)
k_quantize_fp8_multi(const ap_fp8_job* __restrict__ jobs, const float* __restrict__ scales, float* __restrict__ amax) {
    const ap_fp8_job jb = jobs[blockIdx.y];
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(jb.x);
    unsigned char* __restrict__ y = jb.y;
    const int64_t n16 = jb.n >> 4;
    const float sc = scales[jb.slot];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        float f[16];
        unpack8(ld16(x + i * 16), f);
        unpack8(ld16(x + i * 16 + 8), f + 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { mx = fmaxf(mx, fabsf(f[4 * k + e])); v[e] = fminf(fmaxf(f[4 * k + e] * sc, -448.f), 448.f); }
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
            o[k] = (unsigned)w;
        }
        st16(y + i * 16, o);
    }
    if (amax) {
        __shared__ float smx[4];
        mx = group_max<64>(mx);
        if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
            float* a = amax + jb.slot;
            if (__float_as_int(m) > *reinterpret_cast<volatile int*>(a)) atomicMax(reinterpret_cast<int*>(a), __float_as_int(m));
        }
    }
}

// ---------------------------------------------------------------------------- test aid: poison the LDS of every CU
// every workgroup fills all 160 KB of its CU's LDS with `pattern` (e.g. 0x7FC07FC0 = bf16 NaN pairs, 0xFFFFFFFF = fp32 NaN) and spins
// until `min_wgs` workgroups have arrived, so that the fill lands on many CUs: a kernel that reads LDS words it never wrote (padded
// rows of a tile, a tail chunk) then computes on NaNs instead of on whatever the previous kernel left there
__global__ void __launch_bounds__(1024)
k_poison_lds(unsigned pattern, unsigned* counter, int min_wgs) {
    extern __shared__ unsigned lds_all[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds_all[i] = pattern;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(counter, 1u);
        for (int spin = 0; spin < 20000 && atomicAdd(counter, 0u) < (unsigned)min_wgs; ++spin) __builtin_amdgcn_s_sleep(20);
    }
    __syncthreads();
    if (lds_all[threadIdx.x] != pattern) counter[1] = 1;        // (keeps the fill alive)
}

// ---------------------------------------------------------------------------- DropPath masks
// timm DropPath: mask = floor(keep + U[0,1)), factor = mask / keep.  One launch produces, for every DropPath site of
// a forward pass: the per-sample factors, the 0/1 masks and the per-token bf16 masks the bias gradients read (16-byte aligned rows).
__global__ void __launch_bounds__(256)
k_droppath_masks(const float* __restrict__ u, const float* __restrict__ keep, float* __restrict__ factor, float* __restrict__ mask,
                 bf16_t* __restrict__ tokmask, int sites, int B, int tokens, int row) {
    const int64_t total = (int64_t)sites * row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int s = (int)(i / row), t = (int)(i - (int64_t)s * row);
        const int b = tokens > 0 ? t / tokens : t;
        const bool valid = b < B && (tokens > 0 ? t < B * tokens : true);
        const float k = keep[s];
        const float m = valid ? floorf(k + u[s * B + b]) : 0.f;
        if (tokmask) tokmask[i] = f2bf(m);
        if (valid && (tokens > 0 ? t == b * tokens : true)) { mask[s * B + b] = m; factor[s * B + b] = m / k; }
    }
}

__device__ __forceinline__ bf16x8 as_bf16x8_calib(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
// ---- in-process calibration of the box a benchmark runs on (bench.py `calibration`): the boxes of a pool differ by
// +-2-4 % in step time, so a bench line carries what THIS device delivers on two fixed probes -- a float4 copy (HBM) and a register-only
// MFMA loop on random operands (matrix pipe at the clock the chip holds under load) -- and round-over-round changes are quoted against them.
__global__ void __launch_bounds__(256) k_calib_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n16) {
    // a workgroup per 16 KB: four 16-byte loads per lane in flight, then the four stores
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (base + j * 256 < n16) v[j] = __builtin_nontemporal_load(src + base + j * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (base + j * 256 < n16) __builtin_nontemporal_store(v[j], dst + base + j * 256);
}
// one wave per SIMD, 16 independent accumulators, operands in registers: 16 x v_mfma_f32_16x16x32_bf16 per trip
__global__ void __launch_bounds__(256, 1) k_calib_mfma(const bf16_t* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const u32x4*>(seed + (i * 64 + lane) * 8);
        b[i] = *reinterpret_cast<const u32x4*>(seed + ((2 + i) * 64 + lane) * 8);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8_calib(a[i & 1]), as_bf16x8_calib(b[(i >> 1) & 1]), acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 123456.789f) sink[0] = s[0];        // (keeps the loop alive; never true on the seeds bench.py passes)
}

extern "C" {

int ap_abi_version(void) { return 7; }

const char* ap_error_string(int code) {
    switch (code) {
        case AP_OK: return "ok";
        case AP_ERR_SHAPE: return "shape/stride constraint violated";
        case AP_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        case AP_ERR_LAUNCH: return "kernel launch failed";
        case AP_ERR_NULL: return "null pointer";
        default: return "unknown error";
    }
}

int ap_cast_f32_bf16(const float* src, ap_bf16* dst, int64_t n, ap_stream_t stream) {
    if (!src || !dst) return AP_ERR_NULL;
    if (n <= 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_cast_f32_bf16, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return ap_check_launch();
}
int ap_cast_bf16_f32(const ap_bf16* src, float* dst, int64_t n, ap_stream_t stream) {
    if (!src || !dst) return AP_ERR_NULL;
    if (n <= 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_cast_bf16_f32, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return ap_check_launch();
}
int ap_cast_transpose_f32_bf16(const float* src, ap_bf16* dst, int rows, int cols, int ld_dst, ap_stream_t stream) {
    if (!src || !dst) return AP_ERR_NULL;
    if (rows <= 0 || cols <= 0 || ld_dst < rows) return AP_ERR_SHAPE;
    dim3 grid((cols + 31) / 32, (ld_dst + 31) / 32);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_cast_transpose, grid, dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, ld_dst);
    return ap_check_launch();
}
int ap_resize_bilinear_nhwc(const float* x, ap_bf16* y, int B, int C, int Hi, int Wi, int Ho, int Wo, ap_stream_t stream) {
    if (!x || !y) return AP_ERR_NULL;
    if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_resize_bilinear_nhwc, dim3(grid_for((int64_t)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, y, B, C, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return ap_check_launch();
}

int ap_resize_bilinear_s2d16(const float* x, ap_bf16* y, int B, int Hi, int Wi, int Ho, int Wo, ap_stream_t stream) {
    if (!x || !y) return AP_ERR_NULL;
    if (B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (Ho & 1) || (Wo & 1)) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_resize_bilinear_s2d16, dim3(grid_for((int64_t)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, y, B, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return ap_check_launch();
}

int ap_quantize_fp8(const ap_bf16* x, unsigned char* y, int64_t n, const float* scale, float* amax, ap_stream_t stream) {
    if (!x || !y || !scale) return AP_ERR_NULL;
    if (n < 0 || (n & 15)) return AP_ERR_SHAPE;
    if (n == 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_quantize_fp8, dim3(grid_for(n / 16)), dim3(256), 0, (hipStream_t)stream, x, y, n / 16, scale, amax);
    return ap_check_launch();
}

int ap_quantize_fp8_multi(const ap_fp8_job* jobs_device, int njobs, const float* scales, float* amax, ap_stream_t stream) {
    if (!jobs_device || !scales) return AP_ERR_NULL;
    if (njobs < 0 || njobs > 65535) return AP_ERR_SHAPE;
    if (njobs == 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_quantize_fp8_multi, dim3(32, njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, scales, amax);
    return ap_check_launch();
}

int ap_debug_poison_lds(unsigned pattern, unsigned* scratch2, ap_stream_t stream) {
    if (!scratch2) return AP_ERR_NULL;
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    (void)hipGetLastError();
    if (hipMemsetAsync(scratch2, 0, 8, (hipStream_t)stream) != hipSuccess) return AP_ERR_LAUNCH;
    hipLaunchKernelGGL(k_poison_lds, dim3(ncu), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, scratch2, ncu);
    return ap_check_launch();
}

int ap_droppath_masks(const float* uniform, const float* keep, float* factor, float* mask, ap_bf16* token_mask, int sites, int B,
                      int tokens, int token_row, ap_stream_t stream) {
    if (!uniform || !keep || !factor || !mask) return AP_ERR_NULL;
    if (sites <= 0 || B <= 0 || tokens < 0 || (tokens > 0 && (!token_mask || token_row < B * tokens || (token_row & 7)))) return AP_ERR_SHAPE;
    const int row = tokens > 0 ? token_row : B;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_droppath_masks, dim3(grid_for((int64_t)sites * row)), dim3(256), 0, (hipStream_t)stream, uniform, keep, factor, mask,
                       tokens > 0 ? token_mask : nullptr, sites, B, tokens, row);
    return ap_check_launch();
}

int ap_row_scale(const ap_bf16* x, const float* scale, ap_bf16* y, int64_t M, int C, int rows_per_scale, ap_stream_t stream) {
    if (!x || !scale || !y) return AP_ERR_NULL;
    if ((C & 7) || rows_per_scale <= 0) return AP_ERR_SHAPE;
    if (M <= 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_row_scale, dim3(grid_for(M * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, scale, y, M, C / 8, rows_per_scale);
    return ap_check_launch();
}
int ap_add_bcast(const ap_bf16* a, const ap_bf16* b, ap_bf16* y, int64_t n, int64_t b_elems, ap_stream_t stream) {
    if (!a || !b || !y) return AP_ERR_NULL;
    if ((n & 7) || (b_elems & 7) || b_elems <= 0 || n % b_elems) return AP_ERR_SHAPE;
    if (n == 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_add_bcast, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, a, b, y, n / 8, b_elems / 8);
    return ap_check_launch();
}
int ap_resample_grid(const float* in, int hi, int wi, const float* wy, const float* wx, float* out, int ho, int wo, int C, int accumulate,
                     ap_stream_t stream) {
    if (!in || !wy || !wx || !out) return AP_ERR_NULL;
    if (hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || C <= 0 || in == out) return AP_ERR_SHAPE;
    if (hi > 64 || wi > 64 || (int64_t)ho * wo > 0x7fffffff) return AP_ERR_UNSUPPORTED;       // tap rows are staged in 2 x 64 floats of LDS
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_resample_grid, dim3((unsigned)(ho * wo)), dim3(256), 0, (hipStream_t)stream, in, hi, wi, wy, wx, out, ho, wo, C,
                       accumulate ? 1 : 0);
    return ap_check_launch();
}
int ap_sum_reps_acc(const ap_bf16* x, float* out, int64_t n, int reps, ap_stream_t stream) {
    if (!x || !out) return AP_ERR_NULL;
    if ((n & 7) || reps <= 0) return AP_ERR_SHAPE;
    if (n == 0) return AP_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_sum_reps_acc, dim3(grid_for(n / 8), (reps + SR_CHUNK - 1) / SR_CHUNK), dim3(256), 0, (hipStream_t)stream, x, out, n / 8, reps);
    return ap_check_launch();
}
int ap_mix_token_swap(const ap_bf16* x, ap_bf16* y, int B, int H, int W, int C, int r0, int r1, int c0, int c1, ap_stream_t stream) {
    if (!x || !y) return AP_ERR_NULL;
    if ((C & 7) || B <= 0 || H <= 0 || W <= 0 || x == y) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_mix_swap, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, H, W, C / 8, r0, r1, c0, c1, nullptr, 1);
    return ap_check_launch();
}
int ap_mix_token_swap_dev(const ap_bf16* x, ap_bf16* y, int B, int H, int W, int C, const int* box_dev, int scale, ap_stream_t stream) {
    if (!x || !y || !box_dev) return AP_ERR_NULL;
    if ((C & 7) || B <= 0 || H <= 0 || W <= 0 || x == y || scale < 1) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_mix_swap, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, H, W, C / 8, 0, 0, 0, 0, box_dev, scale);
    return ap_check_launch();
}
int ap_avgpool2_fwd(const ap_bf16* x, ap_bf16* y, int B, int H, int W, int C, ap_stream_t stream) {
    if (!x || !y) return AP_ERR_NULL;
    if ((C & 7) || B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_avgpool2_fwd, dim3(grid_for((int64_t)B * h * w * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 8);
    return ap_check_launch();
}
int ap_avgpool2_bwd_acc(const ap_bf16* dpooled, ap_bf16* dx, int B, int H, int W, int C, ap_stream_t stream) {
    if (!dpooled || !dx) return AP_ERR_NULL;
    if ((C & 7) || B <= 0 || H <= 0 || W <= 0) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_avgpool2_bwd_acc, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, dpooled, dx, B, H, W, C / 8);
    return ap_check_launch();
}
int ap_colsum_acc(const ap_bf16* A, int lda, float* out, int M, int N, ap_stream_t stream) {
    if (!A || !out) return AP_ERR_NULL;
    if ((lda & 7) || M <= 0 || N <= 0 || lda < N) return AP_ERR_SHAPE;
    const int gx = (N + 255) / 256;
    int gy = (M + 255) / 256;                         // >= 256 rows per block
    const int cap = (2048 + gx - 1) / gx;
    if (gy > cap) gy = cap;
    const int rows_per_block = (M + gy - 1) / gy;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_colsum_acc, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, A, lda, out, M, N, rows_per_block);
    return ap_check_launch();
}

int ap_calib_copy(const void* src, void* dst, int64_t bytes, ap_stream_t stream) {
    if (!src || !dst) return AP_ERR_NULL;
    if (bytes <= 0 || (bytes & 15)) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_calib_copy, dim3((unsigned)((bytes / 16 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, bytes / 16);
    return ap_check_launch();
}
int ap_calib_mfma(const ap_bf16* seed, float* sink, int iters, ap_stream_t stream) {
    if (!seed || !sink) return AP_ERR_NULL;
    if (iters <= 0) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_calib_mfma, dim3(256), dim3(256), 0, (hipStream_t)stream, seed, sink, iters);
    return ap_check_launch();
}

}  // extern "C"
