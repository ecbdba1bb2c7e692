// Fused multi-tensor AdamW + up to 4 EMA updates over ONE flat fp32 parameter slab (SURVEY.md row N4;
// reference: timm create_optimizer -> torch.optim.AdamW at main_prog.py:484, ModelEmaV2.update x4 at
// main_prog.py:1030-1033 with decays 0.998 0.9986 0.999 0.9996, scripts/train_autoprog.sh:5).
// One pass: reads p,g,m,v,ema_0..3 and a 1-byte weight-decay mask, writes p,m,v,ema_0..3 (60 B per
// parameter instead of ~100 B and ~70 launches for foreach AdamW + 4 foreach lerps).  HBM-bound.
#include "common.h"

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;   // bc1 = 1-beta1^t, bc2_sqrt = sqrt(1-beta2^t)
    float gscale;                                      // gradient pre-scale (1/world_size: the data-parallel mean)
    const float* gnorm_sq;                             // device scalar: sum of squares of the UNSCALED slab (ap_sumsq_f32), or nullptr
    float max_norm;                                    // clip_grad_norm_ bound on the scaled gradient (with gnorm_sq)
    float clip_value;                                  // clip_grad_value_ bound on the scaled gradient elements (> 0), else 0
    const float* step_dev;                             // device [lr, 1 - beta1^t, sqrt(1 - beta2^t)] (graph replay), or nullptr: the host values above
    int n_ema;
    float decay[4];
    float* ema[4];
};

__global__ void __launch_bounds__(256)
k_adamw_ema(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            const unsigned char* __restrict__ wd_mask, int64_t n, AdamArgs a, bf16_t* __restrict__ p16) {
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (a.step_dev) { a.lr = a.step_dev[0]; a.bc1 = a.step_dev[1]; a.bc2_sqrt = a.step_dev[2]; }
    const float step_size = a.lr / a.bc1;
    // gradient clipping folded into the update (prog/scaler.py:60-68 -> timm dispatch_clip_grad, main_prog.py:1019-1027):
    // mode 'norm' = torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (||g|| + 1e-6)) with ||g|| the norm of the MEAN gradient
    // (gscale * the norm of the slab, which holds the all-reduced SUM under a deferred mean); mode 'value': clamp every element
    float gs = a.gscale;
    if (a.gnorm_sq) {
        const float total = sqrtf(a.gnorm_sq[0]) * a.gscale;
        gs *= fminf(1.0f, a.max_norm / (total + 1e-6f));
    }
    const float cv = a.clip_value;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        gg.x *= gs; gg.y *= gs; gg.z *= gs; gg.w *= gs;
        if (cv > 0.f) { gg.x = fminf(fmaxf(gg.x, -cv), cv); gg.y = fminf(fmaxf(gg.y, -cv), cv); gg.z = fminf(fmaxf(gg.z, -cv), cv); gg.w = fminf(fmaxf(gg.w, -cv), cv); }
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        const uchar4 wm = reinterpret_cast<const uchar4*>(wd_mask)[i];
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
        const unsigned char W[4] = {wm.x, wm.y, wm.z, wm.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = P[k];
            if (W[k]) x *= (1.0f - a.lr * a.wd);                         // decoupled weight decay (torch AdamW order)
            M[k] = a.beta1 * M[k] + (1.0f - a.beta1) * G[k];
            V[k] = a.beta2 * V[k] + (1.0f - a.beta2) * G[k] * G[k];
            const float denom = sqrtf(V[k]) / a.bc2_sqrt + a.eps;
            P[k] = x - step_size * (M[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        if (p16) { u32x2 o; o[0] = pack_bf2(pp.x, pp.y); o[1] = pack_bf2(pp.z, pp.w); reinterpret_cast<u32x2*>(p16)[i] = o; }
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e < a.n_ema) {
                float4 ee = reinterpret_cast<float4*>(a.ema[e])[i];
                const float d = a.decay[e];
                ee.x = d * ee.x + (1.0f - d) * pp.x; ee.y = d * ee.y + (1.0f - d) * pp.y;
                ee.z = d * ee.z + (1.0f - d) * pp.z; ee.w = d * ee.w + (1.0f - d) * pp.w;
                reinterpret_cast<float4*>(a.ema[e])[i] = ee;
            }
        }
    }
}

// sum of squares of a flat fp32 slab in two deterministic passes: 1024 per-workgroup partials (fp64 inside a workgroup's tree),
// then one workgroup adds them in order.  16 bytes per lane, every load of a thread's sweep independent.
__global__ void __launch_bounds__(256)
k_sumsq_partial(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    __shared__ double red[4];
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
    }
    double s = (double)s0 + (double)s1 + (double)s2 + (double)s3;
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int64_t i = nv << 2; i < n; ++i) s += (double)x[i] * (double)x[i];     // (n % 4 tail)
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256)
k_sumsq_final(const double* __restrict__ partial, int count, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

extern "C" size_t ap_sumsq_workspace(void) { return (size_t)1024 * sizeof(double); }

extern "C" int ap_sumsq_f32(const float* x, int64_t n, float* out, void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!x || !out || !workspace) return AP_ERR_NULL;
    if (n <= 0 || ws_bytes < ap_sumsq_workspace() || ((uintptr_t)x & 15)) return AP_ERR_SHAPE;
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_sumsq_partial, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n, static_cast<double*>(workspace));
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const double*>(workspace), (int)grid, out);
    return ap_check_launch();
}

extern "C" int ap_adamw_ema_step(float* p, const float* g, float* m, float* v, const unsigned char* wd_mask, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                 const float* gnorm_sq, float max_norm, float clip_value, const float* step_scalars_dev,
                                 float* const* ema, const float* ema_decay, int n_ema, ap_bf16* p_bf16, ap_stream_t stream) {
    if (!p || !g || !m || !v || !wd_mask) return AP_ERR_NULL;
    if (n <= 0 || (n & 3) || n_ema < 0 || n_ema > 4 || step < 1) return AP_ERR_SHAPE;
    if (gnorm_sq && !(max_norm > 0.f)) return AP_ERR_SHAPE;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
    a.gnorm_sq = gnorm_sq; a.max_norm = max_norm; a.clip_value = clip_value > 0.f ? clip_value : 0.f; a.step_dev = step_scalars_dev;
    // (in double from the float arguments, rounded once: graph.StepScalars.set_adam forms the SAME two numbers on the host, so a step
    // replayed from a graph -- which reads them from device memory -- is bit-identical to the eager step)
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.n_ema = n_ema;
    for (int e = 0; e < 4; ++e) { a.ema[e] = (e < n_ema) ? ema[e] : nullptr; a.decay[e] = (e < n_ema) ? ema_decay[e] : 0.f; if (e < n_ema && !ema[e]) return AP_ERR_NULL; }
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_adamw_ema, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, n, a, p_bf16);
    return ap_check_launch();
}

// ---- batched transpose of many bf16 matrices living in one slab (the [K, ld(N)] weight copies used by
// the input-gradient GEMMs): one launch instead of one per Linear.  desc[i] = {src_off, dst_off, rows, cols,
// ld_dst, first_tile}; tiles are 32x32, a workgroup finds its matrix by binary search over first_tile.
struct TrDesc { long long src_off, dst_off; int rows, cols, ld_dst, first_tile; };

__global__ void __launch_bounds__(256)
k_batched_transpose(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const TrDesc* __restrict__ desc, int count) {
    __shared__ bf16_t tile[32][34];
    int lo = 0, hi = count - 1;
    const int t = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].first_tile <= t) lo = mid; else hi = mid - 1; }
    const TrDesc d = desc[lo];
    const int tiles_c = (d.cols + 31) / 32;
    const int lt = t - d.first_tile;
    const int r0 = (lt / tiles_c) * 32, c0 = (lt % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    bf16_t v[4];                                  // all four loads of the thread before the first LDS store
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = r0 + ty + 8 * u, c = c0 + tx;
        v[u] = (r < d.rows && c < d.cols) ? src[d.src_off + (long long)r * d.cols + c] : (bf16_t)0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[ty + 8 * u][tx] = v[u];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < d.cols && r < d.ld_dst) dst[d.dst_off + (long long)c * d.ld_dst + r] = tile[tx][i];
    }
}

extern "C" int ap_batched_transpose_bf16(const ap_bf16* src, ap_bf16* dst, const void* desc_dev, int count, int total_tiles, ap_stream_t stream) {
    if (!src || !dst || !desc_dev) return AP_ERR_NULL;
    if (count <= 0 || total_tiles <= 0) return AP_ERR_SHAPE;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_batched_transpose, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, src, dst, (const TrDesc*)desc_dev, count);
    return ap_check_launch();
}
