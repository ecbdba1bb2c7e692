// Fused BatchNorm2d (training / eval) + ReLU for the conv stem on NHWC bf16 activations
// (reference: nn.BatchNorm2d + nn.ReLU after each stem conv, models/volo.py:355-367).
// MIOpen needs 3 kernels + a separate ReLU pass per layer in each direction; here forward = one reduction
// pass + one apply pass, backward = one reduction pass + one dx pass (ReLU mask recomputed from x).
// Layout: x [T, C] rows (T = B*H*W, C in {8,...,512}, C/8 a power of two <= 64): a lane owns one 16-byte
// channel chunk (lane % (C/8)) for all its rows, so per-channel sums stay in registers.
#include "common.h"

#define BN_BLOCK 256

// partial[blockIdx][0..C) = sum x, [C..2C) = sum x^2   (fp32 per block, combined in fp64 by the finalize kernel)
__global__ void __launch_bounds__(BN_BLOCK)
k_bn_stats(const bf16_t* __restrict__ x, float* __restrict__ partial, int64_t T, int C) {
    __shared__ float red[2][BN_BLOCK / 64][512];
    const int cpr = C >> 3;                                   // chunks per row
    const int chunk = threadIdx.x % cpr;
    const int rows_per_pass = BN_BLOCK / cpr;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t row = (int64_t)blockIdx.x * rows_per_pass + threadIdx.x / cpr; row < T; row += (int64_t)gridDim.x * rows_per_pass) {
        float f[8];
        unpack8(ld16(x + row * C + chunk * 8), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += f[k]; q[k] += f[k] * f[k]; }
    }
    // lanes with the same chunk inside a wave: lane ids differ by multiples of cpr
    for (int o = 32; o >= cpr; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += __shfl_xor(s[k], o, 64); q[k] += __shfl_xor(q[k], o, 64); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < cpr || cpr > 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[0][wave][chunk * 8 + k] = s[k]; red[1][wave][chunk * 8 + k] = q[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += BN_BLOCK) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < BN_BLOCK / 64; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
        partial[(int64_t)blockIdx.x * 2 * C + c] = a;
        partial[(int64_t)blockIdx.x * 2 * C + C + c] = b;
    }
}

// block-wide fp64 sum of the per-workgroup partials of ONE channel (blockIdx.x = channel): the partial rows are
// read by 256 threads in parallel (a single thread walking 2048 rows is a ~0.5 ms chain of L2 round trips)
__device__ __forceinline__ void bn_channel_sums(const float* __restrict__ partial, int nblocks, int C, int c, double& s, double& q) {
    __shared__ double rs[BN_BLOCK], rq[BN_BLOCK];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += BN_BLOCK) { a += partial[(int64_t)i * 2 * C + c]; b += partial[(int64_t)i * 2 * C + C + c]; }
    rs[threadIdx.x] = a; rq[threadIdx.x] = b;
    __syncthreads();
    for (int o = BN_BLOCK / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; }
        __syncthreads();
    }
    s = rs[0]; q = rq[0];
}

// mean/rstd from the partials (+ running-stat update with momentum, unbiased variance as nn.BatchNorm2d)
__global__ void __launch_bounds__(BN_BLOCK)
k_bn_finalize(const float* __restrict__ partial, int nblocks, int64_t T, int C, float eps, float momentum,
              float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
              float* __restrict__ running_var) {
    const int c = blockIdx.x;
    double s, q;
    bn_channel_sums(partial, nblocks, C, c, s, q);
    if (threadIdx.x != 0) return;
    const double m = s / (double)T;
    double var = q / (double)T - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = T > 1 ? var * (double)T / (double)(T - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// y = relu((x - mean) * rstd * gamma + beta)
__global__ void __launch_bounds__(BN_BLOCK)
k_bn_relu_apply(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t T, int C) {
    const int cpr = C >> 3;
    const int chunk = threadIdx.x % cpr;
    const int rows_per_pass = BN_BLOCK / cpr;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = chunk * 8 + k;
        sc[k] = rstd[c] * gamma[c];
        sh[k] = beta[c] - mean[c] * sc[k];
    }
    for (int64_t row = (int64_t)blockIdx.x * rows_per_pass + threadIdx.x / cpr; row < T; row += (int64_t)gridDim.x * rows_per_pass) {
        float f[8];
        unpack8(ld16(x + row * C + chunk * 8), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = fmaxf(fmaf(f[k], sc[k], sh[k]), 0.f);
        st16(y + row * C + chunk * 8, pack8(f));
    }
}

// backward pass 1: partial[b][0..C) = sum dz, [C..2C) = sum dz*xhat with dz = dy * (z > 0), z = xhat*gamma+beta
__global__ void __launch_bounds__(BN_BLOCK)
k_bn_relu_bwd_reduce(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                     const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float* __restrict__ partial, int64_t T, int C) {
    __shared__ float red[2][BN_BLOCK / 64][512];
    const int cpr = C >> 3;
    const int chunk = threadIdx.x % cpr;
    const int rows_per_pass = BN_BLOCK / cpr;
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int c = chunk * 8 + k; mu[k] = mean[c]; rs[k] = rstd[c]; ga[k] = gamma[c]; be[k] = beta[c]; }
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t row = (int64_t)blockIdx.x * rows_per_pass + threadIdx.x / cpr; row < T; row += (int64_t)gridDim.x * rows_per_pass) {
        float f[8], d[8];
        unpack8(ld16(x + row * C + chunk * 8), f);
        unpack8(ld16(dy + row * C + chunk * 8), d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (f[k] - mu[k]) * rs[k];
            const float dz = (fmaf(xh, ga[k], be[k]) > 0.f) ? d[k] : 0.f;
            s[k] += dz; q[k] += dz * xh;
        }
    }
    for (int o = 32; o >= cpr; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += __shfl_xor(s[k], o, 64); q[k] += __shfl_xor(q[k], o, 64); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < cpr || cpr > 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[0][wave][chunk * 8 + k] = s[k]; red[1][wave][chunk * 8 + k] = q[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += BN_BLOCK) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < BN_BLOCK / 64; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
        partial[(int64_t)blockIdx.x * 2 * C + c] = a;
        partial[(int64_t)blockIdx.x * 2 * C + C + c] = b;
    }
}

// dbeta/dgamma (+=) from the partials; sums[0..C) = dbeta total, sums[C..2C) = dgamma total (for pass 2)
__global__ void __launch_bounds__(BN_BLOCK)
k_bn_bwd_finalize(const float* __restrict__ partial, int nblocks, int C, float* __restrict__ sums,
                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x;
    double s, q;
    bn_channel_sums(partial, nblocks, C, c, s, q);
    if (threadIdx.x != 0) return;
    sums[c] = (float)s; sums[C + c] = (float)q;
    dbeta[c] += (float)s; dgamma[c] += (float)q;
}

// backward pass 2: dx = gamma*rstd*(dz - dbeta/T - xhat*dgamma/T)
// ACT: also write act = relu(bn(x)) in the arithmetic of k_bn_relu_apply / bn_in_apply (the forward that never stored it, gemm_epi.h): the
// pass holds x anyway, and the weight gradient of the layer above then reads a plain tensor instead of transforming every chunk it stages
template <bool ACT>
__global__ void __launch_bounds__(BN_BLOCK)
k_bn_relu_bwd_dx(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                 const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                 const float* __restrict__ sums, bf16_t* __restrict__ dx, int64_t T, int C, bf16_t* __restrict__ act = nullptr) {
    const int cpr = C >> 3;
    const int chunk = threadIdx.x % cpr;
    const int rows_per_pass = BN_BLOCK / cpr;
    const float invT = 1.0f / (float)T;
    float mu[8], rs[8], ga[8], be[8], sb[8], sg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = chunk * 8 + k;
        mu[k] = mean[c]; rs[k] = rstd[c]; ga[k] = gamma[c]; be[k] = beta[c]; sb[k] = sums[c] * invT; sg[k] = sums[C + c] * invT;
    }
    for (int64_t row = (int64_t)blockIdx.x * rows_per_pass + threadIdx.x / cpr; row < T; row += (int64_t)gridDim.x * rows_per_pass) {
        float f[8], d[8];
        unpack8(ld16(x + row * C + chunk * 8), f);
        unpack8(ld16(dy + row * C + chunk * 8), d);
        if constexpr (ACT) {
            float a8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float sc = rs[k] * ga[k]; a8[k] = fmaxf(fmaf(f[k], sc, be[k] - mu[k] * sc), 0.f); }
            st16(act + row * C + chunk * 8, pack8(a8));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (f[k] - mu[k]) * rs[k];
            const float dz = (fmaf(xh, ga[k], be[k]) > 0.f) ? d[k] : 0.f;
            f[k] = ga[k] * rs[k] * (dz - sb[k] - xh * sg[k]);
        }
        st16(dx + row * C + chunk * 8, pack8(f));
    }
}

static inline bool bn_shape_ok(int C) {
    if (C < 8 || C > 512 || (C & 7)) return false;
    const int cpr = C >> 3;
    return (cpr & (cpr - 1)) == 0;
}
static inline int bn_grid(int64_t T, int C) {
    const int rows_per_pass = BN_BLOCK / (C >> 3);
    int64_t g = (T + rows_per_pass - 1) / rows_per_pass;
    if (g > 2048) g = 2048;
    return (int)(g < 1 ? 1 : g);
}

extern "C" {

size_t ap_bn_relu_workspace(int64_t T, int C) { (void)T; return (size_t)(2048 * 2 + 2) * (size_t)C * sizeof(float); }

int ap_bn_relu_fwd(const ap_bf16* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   int training, float momentum, float eps, ap_bf16* y, float* mean, float* rstd, int64_t T, int C,
                   void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return AP_ERR_NULL;
    if (!bn_shape_ok(C) || T <= 0) return AP_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int grid = bn_grid(T, C);
    (void)hipGetLastError();
    if (training) {
        if (!workspace || ws_bytes < ap_bn_relu_workspace(T, C)) return AP_ERR_SHAPE;
        float* partial = static_cast<float*>(workspace);
        hipLaunchKernelGGL(k_bn_stats, dim3(grid), dim3(BN_BLOCK), 0, s, x, partial, T, C);
        hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(BN_BLOCK), 0, s, partial, grid, T, C, eps, momentum, mean, rstd, running_mean, running_var);
    }
    // eval mode: the caller passes mean = running_mean and rstd = 1/sqrt(running_var + eps)
    hipLaunchKernelGGL(k_bn_relu_apply, dim3(grid), dim3(BN_BLOCK), 0, s, x, mean, rstd, gamma, beta, y, T, C);
    return ap_check_launch();
}

int ap_bn_relu_fwd_partials(const ap_bf16* x, const float* partial, int n_partial, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, ap_bf16* y, float* mean, float* rstd,
                            int64_t T, int C, ap_stream_t stream) {
    if (!partial || !mean || !rstd || (y && (!x || !gamma || !beta))) return AP_ERR_NULL;
    if (!bn_shape_ok(C) || T <= 0 || n_partial <= 0) return AP_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(BN_BLOCK), 0, s, partial, n_partial, T, C, eps, momentum, mean, rstd, running_mean, running_var);
    if (y) hipLaunchKernelGGL(k_bn_relu_apply, dim3(bn_grid(T, C)), dim3(BN_BLOCK), 0, s, x, mean, rstd, gamma, beta, y, T, C);
    return ap_check_launch();
}

int ap_bn_relu_bwd(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* beta, const float* mean,
                   const float* rstd, ap_bf16* dx, float* dgamma, float* dbeta, int64_t T, int C,
                   void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return AP_ERR_NULL;
    if (!bn_shape_ok(C) || T <= 0 || ws_bytes < ap_bn_relu_workspace(T, C)) return AP_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int grid = bn_grid(T, C);
    float* partial = static_cast<float*>(workspace);
    float* sums = partial + (size_t)2048 * 2 * C;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_bn_relu_bwd_reduce, dim3(grid), dim3(BN_BLOCK), 0, s, dy, x, mean, rstd, gamma, beta, partial, T, C);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C), dim3(BN_BLOCK), 0, s, partial, grid, C, sums, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn_relu_bwd_dx<false>, dim3(grid), dim3(BN_BLOCK), 0, s, dy, x, mean, rstd, gamma, beta, sums, dx, T, C, nullptr);
    return ap_check_launch();
}

int ap_bn_relu_bwd_act(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, ap_bf16* dx, ap_bf16* act, float* dgamma, float* dbeta, int64_t T, int C,
                       void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || !act || !dgamma || !dbeta || !workspace) return AP_ERR_NULL;
    if (!bn_shape_ok(C) || T <= 0 || ws_bytes < ap_bn_relu_workspace(T, C)) return AP_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int grid = bn_grid(T, C);
    float* partial = static_cast<float*>(workspace);
    float* sums = partial + (size_t)2048 * 2 * C;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_bn_relu_bwd_reduce, dim3(grid), dim3(BN_BLOCK), 0, s, dy, x, mean, rstd, gamma, beta, partial, T, C);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C), dim3(BN_BLOCK), 0, s, partial, grid, C, sums, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn_relu_bwd_dx<true>, dim3(grid), dim3(BN_BLOCK), 0, s, dy, x, mean, rstd, gamma, beta, sums, dx, T, C, act);
    return ap_check_launch();
}

int ap_bn_relu_bwd_partials(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, const float* partial, int n_partial, ap_bf16* dx, float* dgamma, float* dbeta,
                            int64_t T, int C, void* workspace, size_t ws_bytes, ap_stream_t stream) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !partial || !dx || !dgamma || !dbeta || !workspace) return AP_ERR_NULL;
    if (!bn_shape_ok(C) || T <= 0 || n_partial <= 0 || ws_bytes < ap_bn_relu_workspace(T, C)) return AP_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    float* sums = static_cast<float*>(workspace) + (size_t)2048 * 2 * C;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C), dim3(BN_BLOCK), 0, s, partial, n_partial, C, sums, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn_relu_bwd_dx<false>, dim3(bn_grid(T, C)), dim3(BN_BLOCK), 0, s, dy, x, mean, rstd, gamma, beta, sums, dx, T, C, nullptr);
    return ap_check_launch();
}

}  // extern "C"
