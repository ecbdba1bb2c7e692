// Bench + check of the row-complete GEMM with LayerNorm in its epilogue (gemm8r.h) against the shipped pair of launches
// (ap_gemm_nt then ap_layernorm_fwd / ap_layernorm_bwd_partial).  GPU box only.  Build: make -C tools/gemm_lab lab8r
// usage: lab8r [M K mode] ...   mode = lnf | lnb     (N is 384)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <type_traits>
#define k_gemm_nt_8r k_gemm_nt_8r_lab
#include "gemm8r.h"
#include "../../include/autoprog_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = f2bf(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void k_fillf(float* p, size_t n, unsigned seed, float scale, float add) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) / 32768.0f - 1.0f) * scale + add;
    }
}
__global__ void k_diff(const bf16_t* a, const bf16_t* b, size_t n, float* out) {
    float mx = 0.f, ref = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = bf2f(a[i]), y = bf2f(b[i]);
        mx = fmaxf(mx, fabsf(x - y)); ref = fmaxf(ref, fabsf(y));
    }
    atomicMax((unsigned*)out, __float_as_uint(mx));
    atomicMax((unsigned*)out + 1, __float_as_uint(ref));
}
__global__ void k_diff_f(const float* a, const float* b, size_t n, float* out) {
    float mx = 0.f, ref = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        mx = fmaxf(mx, fabsf(a[i] - b[i])); ref = fmaxf(ref, fabsf(b[i]));
    }
    atomicMax((unsigned*)out, __float_as_uint(mx));
    atomicMax((unsigned*)out + 1, __float_as_uint(ref));
}
__global__ void k_colsum(const float* p, int rows, int cols, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += p[(size_t)r * cols + c];
    out[c] = t;
}

struct Shape { int M, K; std::string mode; };

template <int MODE>
static void go(const G8RArgs& ga, const EpiArgs& ep, hipStream_t st) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)k_gemm_nt_8r<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, G8R_LDS_BYTES)); attr = true; }
    hipLaunchKernelGGL((k_gemm_nt_8r<MODE>), dim3(std::min(ga.ntiles, 256)), dim3(512), G8R_LDS_BYTES, st, ga, ep);
}

int main(int argc, char** argv) {
    std::vector<Shape> shapes;
    for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), argv[i + 2]});
    if (shapes.empty()) shapes = {{25088, 384, "lnf"}, {25088, 1152, "lnf"}, {25088, 1152, "lnb"}, {25000, 384, "lnf"}, {25000, 1152, "lnb"}, {4100, 128, "lnb"}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* dd; CK(hipMalloc(&dd, 8));
    const int N = 384;
    printf("%-22s %10s %10s   %s\n", "shape", "fused us", "pair us", "maxdiff/ref (out0, out1, stats)");
    for (const Shape& s : shapes) {
        const bool fwd = s.mode == "lnf";
        const size_t nA = (size_t)s.M * s.K, nC = (size_t)s.M * N;
        const int nset = 4;
        const int ntiles = (s.M + 127) / 128;
        std::vector<bf16_t*> A(nset), B(nset), R(nset), X(nset), O0(nset), O1(nset), P0(nset), P1(nset);
        std::vector<float*> mean(nset), rstd(nset), mean2(nset), rstd2(nset), part(nset);
        size_t wsb = ap_layernorm_bwd_workspace(s.M, N);
        std::vector<void*> ws(nset);
        for (int i = 0; i < nset; ++i) {
            CK(hipMalloc(&A[i], nA * 2)); CK(hipMalloc(&B[i], (size_t)N * s.K * 2)); CK(hipMalloc(&R[i], nC * 2)); CK(hipMalloc(&X[i], nC * 2));
            CK(hipMalloc(&O0[i], nC * 2)); CK(hipMalloc(&O1[i], nC * 2)); CK(hipMalloc(&P0[i], nC * 2)); CK(hipMalloc(&P1[i], nC * 2));
            CK(hipMalloc(&mean[i], s.M * 4)); CK(hipMalloc(&rstd[i], s.M * 4)); CK(hipMalloc(&mean2[i], s.M * 4)); CK(hipMalloc(&rstd2[i], s.M * 4));
            CK(hipMalloc(&part[i], (size_t)ntiles * 768 * 4)); CK(hipMalloc(&ws[i], wsb));
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, A[i], nA, 17u + i, 1.0f);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, B[i], (size_t)N * s.K, 91u + i, 0.125f);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, R[i], nC, 5u + i, 1.0f);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, X[i], nC, 7u + i, 2.0f);
            hipLaunchKernelGGL(k_fillf, dim3(64), dim3(256), 0, st, mean[i], (size_t)s.M, 23u, 0.1f, 0.f);
            hipLaunchKernelGGL(k_fillf, dim3(64), dim3(256), 0, st, rstd[i], (size_t)s.M, 29u, 0.2f, 0.9f);
            CK(hipMemsetAsync(O0[i], 0xff, nC * 2, st)); CK(hipMemsetAsync(P0[i], 0x7f, nC * 2, st));
        }
        float *bias, *gam, *bet, *rsc, *cs0, *cs1;
        CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&gam, N * 4)); CK(hipMalloc(&bet, N * 4)); CK(hipMalloc(&rsc, (s.M / 196 + 2) * 4));
        CK(hipMalloc(&cs0, 768 * 4)); CK(hipMalloc(&cs1, 768 * 4));
        hipLaunchKernelGGL(k_fillf, dim3(4), dim3(256), 0, st, bias, (size_t)N, 3u, 0.5f, 0.f);
        hipLaunchKernelGGL(k_fillf, dim3(4), dim3(256), 0, st, gam, (size_t)N, 31u, 0.3f, 1.0f);
        hipLaunchKernelGGL(k_fillf, dim3(4), dim3(256), 0, st, bet, (size_t)N, 37u, 0.3f, 0.f);
        hipLaunchKernelGGL(k_fillf, dim3(4), dim3(256), 0, st, rsc, (size_t)(s.M / 196 + 2), 11u, 1.0f, 0.f);
        auto fused = [&](int i) {
            G8RArgs ga; memset(&ga, 0, sizeof(ga));
            EpiArgs ep; memset(&ep, 0, sizeof(ep));
            ga.A = A[i]; ga.lda = s.K; ga.B = B[i]; ga.ldb = s.K; ga.C = O0[i]; ga.ldc = N; ga.M = s.M; ga.K = s.K; ga.ntiles = ntiles;
            ga.gamma = gam; ga.beta = bet; ga.eps = 1e-5f;
            if (fwd) {
                ga.xn = O1[i]; ga.ldxn = N; ga.mean = mean2[i]; ga.rstd = rstd2[i];
                ep.bias = bias; ep.row_scale = rsc; ep.rows_per_scale = 196; ep.residual = R[i]; ep.ldr = N;
                go<1>(ga, ep, st);
            } else {
                ga.mean = mean[i]; ga.rstd = rstd[i]; ga.x = X[i]; ga.ldx = N; ga.dres = R[i]; ga.lddres = N; ga.partial = part[i];
                go<2>(ga, ep, st);
            }
        };
        int npart = 0;
        auto pair = [&](int i) {
            ap_gemm_epilogue e; memset(&e, 0, sizeof(e));
            if (fwd) {
                e.bias = bias; e.row_scale = rsc; e.rows_per_scale = 196; e.residual = R[i]; e.ldr = N;
                ap_gemm_nt(A[i], s.K, B[i], s.K, P0[i], N, s.M, N, s.K, &e, st);
                ap_layernorm_fwd(P0[i], gam, bet, P1[i], mean[i], rstd[i], s.M, N, 1e-5f, st);
            } else {
                ap_gemm_nt(A[i], s.K, B[i], s.K, P1[i], N, s.M, N, s.K, &e, st);
                ap_layernorm_bwd_partial(P1[i], X[i], gam, mean[i], rstd[i], R[i], P0[i], s.M, N, ws[i], wsb, &npart, st);
            }
        };
        fused(0); CK(hipGetLastError()); pair(0); CK(hipStreamSynchronize(st));
        auto diff = [&](const bf16_t* a, const bf16_t* b, size_t n, float* hd) {
            CK(hipMemsetAsync(dd, 0, 8, st));
            hipLaunchKernelGGL(k_diff, dim3(512), dim3(256), 0, st, a, b, n, dd);
            CK(hipMemcpyAsync(hd, dd, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        };
        auto diff_f = [&](const float* a, const float* b, size_t n, float* hd) {
            CK(hipMemsetAsync(dd, 0, 8, st));
            hipLaunchKernelGGL(k_diff_f, dim3(64), dim3(256), 0, st, a, b, n, dd);
            CK(hipMemcpyAsync(hd, dd, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        };
        float d0[2], d1[2] = {0, 0}, d2[2] = {0, 0}, d3[2] = {0, 0};
        diff(O0[0], P0[0], nC, d0);
        if (fwd) { diff(O1[0], P1[0], nC, d1); diff_f(mean2[0], mean[0], s.M, d2); diff_f(rstd2[0], rstd[0], s.M, d3); }
        else {
            hipLaunchKernelGGL(k_colsum, dim3(3), dim3(256), 0, st, part[0], ntiles, 768, cs0);
            hipLaunchKernelGGL(k_colsum, dim3(3), dim3(256), 0, st, (const float*)ws[0], npart, 768, cs1);
            diff_f(cs0, cs1, 768, d2);
        }
        auto timeit = [&](auto fn) {
            const int reps = 40;
            for (int i = 0; i < nset; ++i) fn(i);
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) fn(r % nset);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.f / reps;
        };
        float tf = 1e9, tp = 1e9;
        for (int round = 0; round < 3; ++round) { tf = std::min(tf, timeit(fused)); tp = std::min(tp, timeit(pair)); }
        char nm[64]; snprintf(nm, sizeof nm, "%dx384x%d %s", s.M, s.K, s.mode.c_str());
        printf("%-22s %10.1f %10.1f   %.4g/%.3g  %.4g/%.3g  %.4g/%.3g  %.4g/%.3g\n", nm, tf, tp, d0[0], d0[1], d1[0], d1[1], d2[0], d2[1], d3[0], d3[1]);
        fflush(stdout);
        for (int i = 0; i < nset; ++i) {
            hipFree(A[i]); hipFree(B[i]); hipFree(R[i]); hipFree(X[i]); hipFree(O0[i]); hipFree(O1[i]); hipFree(P0[i]); hipFree(P1[i]);
            hipFree(mean[i]); hipFree(rstd[i]); hipFree(mean2[i]); hipFree(rstd2[i]); hipFree(part[i]); hipFree(ws[i]);
        }
        hipFree(bias); hipFree(gam); hipFree(bet); hipFree(rsc); hipFree(cs0); hipFree(cs1);
    }
    return 0;
}
