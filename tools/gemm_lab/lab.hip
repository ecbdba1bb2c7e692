// GEMM laboratory (GPU box only): times NT-GEMM kernel variants on the VOLO-D1 shapes with operands ROTATED over more memory
// than the 256 MiB Infinity Cache (cold operands, as inside the training step), checks each variant against the shipped
// ap_gemm_nt, prints one table.  Build: make -C tools/gemm_lab ; run: tools/gemm_lab/lab [filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include <algorithm>
#include "rejected/gemm_dma.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = f2bf(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void k_fillf(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) / 32768.0f - 1.0f) * scale;
    }
}
__global__ void k_diff(const bf16_t* a, const bf16_t* b, int M, int N, int ld, float* out) {
    float mx = 0.f, ref = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * N; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / N, c = i % N;
        const float x = bf2f(a[r * ld + c]), y = bf2f(b[r * ld + c]);
        mx = fmaxf(mx, fabsf(x - y)); ref = fmaxf(ref, fabsf(y));
    }
    atomicMax((unsigned*)out, __float_as_uint(mx));
    atomicMax((unsigned*)out + 1, __float_as_uint(ref));
}

struct Shape { int M, N, K; const char* epi; };
struct Set { bf16_t *A, *B, *C, *R, *H, *P; };
struct Variant { std::string name; std::function<bool(const Shape&, const Set&, int ldc, EpiArgs ep, float* bias, float* rs, hipStream_t)> run; };

static int g_ncu = 256;

static int g_dbg = 0;
template <int TM, int TN, int WGM, int WGN, int ST, int BK = 64>
static bool launch_dma(const Shape& s, const Set& b, int ldc, EpiArgs ep, int wg_per_cu /*0 = one workgroup per tile*/, hipStream_t st) {
    if (s.K % BK) return false;
    ep.dbg = g_dbg;
    const size_t lds = (size_t)ST * (TM + TN) * BK * 2;
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)k_gemm_nt_dma<TM, TN, WGM, WGN, ST, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    const int tm = (s.M + TM - 1) / TM, tn = (s.N + TN - 1) / TN, nt = tm * tn;
    int grid = nt;
    if (wg_per_cu > 0) grid = std::min(nt, wg_per_cu * g_ncu);
    hipLaunchKernelGGL((k_gemm_nt_dma<TM, TN, WGM, WGN, ST, BK>), dim3(grid), dim3(WGM * WGN * 64), lds, st, b.A, s.K, b.B, s.K, b.C, ldc, s.M, s.N, s.K, tn, nt, ep);
    return hipGetLastError() == hipSuccess;
}

template <int TM, int TN, int WGM, int WGN, int ST, int BK, int ILV>
static bool launch_dma2(const Shape& s, const Set& b, int ldc, EpiArgs ep, int wg_per_cu, hipStream_t st) {
    if (s.K % BK) return false;
    ep.dbg = g_dbg;
    const size_t lds = (size_t)ST * (TM + TN) * BK * 2;
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)k_gemm_nt_dma2<TM, TN, WGM, WGN, ST, BK, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    const int tm = (s.M + TM - 1) / TM, tn = (s.N + TN - 1) / TN, nt = tm * tn;
    int grid = nt;
    if (wg_per_cu > 0) grid = std::min(nt, wg_per_cu * g_ncu);
    hipLaunchKernelGGL((k_gemm_nt_dma2<TM, TN, WGM, WGN, ST, BK, ILV>), dim3(grid), dim3(WGM * WGN * 64), lds, st, b.A, s.K, b.B, s.K, b.C, ldc, s.M, s.N, s.K, tn, nt, ep);
    return hipGetLastError() == hipSuccess;
}

template <int TM, int TN, int WGM, int WGN, int ST>
static bool launch_dma3(const Shape& s, const Set& b, int ldc, EpiArgs ep, int wg_per_cu, hipStream_t st) {
    if (s.K % 64) return false;
    ep.dbg = g_dbg;
    const size_t lds = (size_t)ST * (TM + TN) * 64;
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)k_gemm_nt_dma3<TM, TN, WGM, WGN, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    const int tm = (s.M + TM - 1) / TM, tn = (s.N + TN - 1) / TN, nt = tm * tn;
    int grid = nt;
    if (wg_per_cu > 0) grid = std::min(nt, wg_per_cu * g_ncu);
    hipLaunchKernelGGL((k_gemm_nt_dma3<TM, TN, WGM, WGN, ST>), dim3(grid), dim3(WGM * WGN * 64), lds, st, b.A, s.K, b.B, s.K, b.C, ldc, s.M, s.N, s.K, tn, nt, ep);
    return hipGetLastError() == hipSuccess;
}

int main(int argc, char** argv) {
    const char* filter = argc > 1 ? argv[1] : "";
    const char* vfilter = argc > 2 ? argv[2] : "";
    if (getenv("LAB_DBG")) g_dbg = atoi(getenv("LAB_DBG"));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); g_ncu = pr.multiProcessorCount;
    printf("# device %s, %d CUs\n", pr.name, g_ncu);
    std::vector<Shape> shapes = {
        {25088, 1152, 384, "plain"}, {25088, 1152, 384, "gelu"}, {25088, 1152, 384, "dgelu"}, {25088, 384, 1152, "plain"},
        {25088, 384, 1152, "res"}, {25088, 384, 384, "plain"}, {25088, 384, 384, "res"},
        {100352, 576, 192, "gelu"}, {100352, 576, 192, "dgelu"}, {100352, 192, 576, "plain"}, {100352, 192, 576, "res"},
        {100352, 192, 192, "plain"}, {100352, 192, 192, "res"}, {25088, 486, 192, "bias"}, {25088, 192, 512, "plain"},
        {25216, 768, 384, "plain"}, {25088, 1000, 384, "bias"},
    };
    std::vector<Variant> vars;
    vars.push_back({"base(ap_gemm_nt)", [](const Shape& s, const Set& b, int ldc, EpiArgs ep, float* bias, float* rs, hipStream_t st) {
        ap_gemm_epilogue e; memset(&e, 0, sizeof(e));
        e.bias = ep.bias; e.gelu = ep.gelu; e.preact_out = ep.preact; e.dgelu_of = ep.dgelu_of; e.row_scale = ep.row_scale; e.rows_per_scale = ep.rows_per_scale;
        e.residual = ep.residual; e.ldr = ep.ldr;
        return ap_gemm_nt(b.A, s.K, b.B, s.K, b.C, ldc, s.M, s.N, s.K, &e, st) == 0; }});
#define DMA(NAME, TM, TN, WGM, WGN, ST, BK, WPC) vars.push_back({NAME, [](const Shape& s, const Set& b, int ldc, EpiArgs ep, float*, float*, hipStream_t st) { return launch_dma<TM, TN, WGM, WGN, ST, BK>(s, b, ldc, ep, WPC, st); }});
    DMA("d256x192w8k64s2p1", 256, 192, 4, 2, 2, 64, 1)
    DMA("d256x256w8k32s4p1", 256, 256, 4, 2, 4, 32, 1)
#define DMA2(NAME, TM, TN, WGM, WGN, ST, BK, ILV, WPC) vars.push_back({NAME, [](const Shape& s, const Set& b, int ldc, EpiArgs ep, float*, float*, hipStream_t st) { return launch_dma2<TM, TN, WGM, WGN, ST, BK, ILV>(s, b, ldc, ep, WPC, st); }});
    DMA2("e256x192k64s2i1", 256, 192, 4, 2, 2, 64, 1, 1)
    DMA2("e256x192k32s5i1", 256, 192, 4, 2, 5, 32, 1, 1)
#define DMA3(NAME, TM, TN, WGM, WGN, ST, WPC) vars.push_back({NAME, [](const Shape& s, const Set& b, int ldc, EpiArgs ep, float*, float*, hipStream_t st) { return launch_dma3<TM, TN, WGM, WGN, ST>(s, b, ldc, ep, WPC, st); }});
    DMA3("f256x192s5", 256, 192, 4, 2, 5, 1)
    DMA3("f256x192s4", 256, 192, 4, 2, 4, 1)
    DMA3("f256x192s3", 256, 192, 4, 2, 3, 1)
    DMA3("f320x192s4", 320, 192, 4, 2, 4, 1)
    DMA3("f256x128s6", 256, 128, 4, 2, 6, 1)
    DMA3("f128x192s4p2", 128, 192, 2, 2, 4, 2)
    DMA3("f128x192w8s7", 128, 192, 4, 2, 7, 1)
    DMA3("f256x256s4", 256, 256, 4, 2, 4, 1)
    const size_t ROT_BYTES = (size_t)640 << 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    float* d_diff; CK(hipMalloc(&d_diff, 8));
    for (const Shape& s : shapes) {
        char tag[96]; snprintf(tag, sizeof(tag), "%dx%dx%d:%s", s.M, s.N, s.K, s.epi);
        if (filter[0] && !strstr(tag, filter)) continue;
        const int ldc = (s.N + 7) / 8 * 8;
        const bool gelu = !strcmp(s.epi, "gelu"), dgelu = !strcmp(s.epi, "dgelu"), res = !strcmp(s.epi, "res"), bias = gelu || res || !strcmp(s.epi, "bias");
        const size_t bytes_set = 2 * ((size_t)s.M * s.K + (size_t)s.N * s.K + (size_t)s.M * ldc * (1 + (gelu || dgelu) + res));
        const int nset = (int)std::max<size_t>(2, (ROT_BYTES + bytes_set - 1) / bytes_set);
        std::vector<Set> sets(nset);
        for (int i = 0; i < nset; ++i) {
            Set& b = sets[i];
            CK(hipMalloc(&b.A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&b.B, (size_t)s.N * s.K * 2)); CK(hipMalloc(&b.C, (size_t)s.M * ldc * 2));
            b.R = b.H = b.P = nullptr;
            if (res) CK(hipMalloc(&b.R, (size_t)s.M * ldc * 2));
            if (gelu || dgelu) CK(hipMalloc(&b.H, (size_t)s.M * ldc * 2));
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, b.A, (size_t)s.M * s.K, 17u + i, 1.0f);
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, b.B, (size_t)s.N * s.K, 91u + i, 0.05f);
            if (res) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, b.R, (size_t)s.M * ldc, 33u + i, 1.0f);
            if (dgelu) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, b.H, (size_t)s.M * ldc, 55u + i, 2.0f);
        }
        float *d_bias, *d_rs; CK(hipMalloc(&d_bias, ldc * 4)); CK(hipMalloc(&d_rs, 4 * (s.M / 196 + 1)));
        hipLaunchKernelGGL(k_fillf, dim3(4), dim3(256), 0, st, d_bias, (size_t)ldc, 5u, 0.5f);
        hipLaunchKernelGGL(k_fillf, dim3(4), dim3(256), 0, st, d_rs, (size_t)(s.M / 196 + 1), 7u, 1.0f);
        bf16_t* Cref; CK(hipMalloc(&Cref, (size_t)s.M * ldc * 2));
        CK(hipStreamSynchronize(st));
        auto epi_for = [&](const Set& b) {
            EpiArgs ep = {nullptr, 0, nullptr, nullptr, nullptr, 1, nullptr, 0, 0, nullptr, {0, 0, 0, 0, 0, 0u, 0u}, nullptr, nullptr};
            if (bias) ep.bias = d_bias;
            if (gelu) { ep.gelu = 1; ep.preact = b.H; }
            if (dgelu) ep.dgelu_of = b.H;
            if (res) { ep.residual = b.R; ep.ldr = ldc; ep.row_scale = d_rs; ep.rows_per_scale = 196; }
            return ep;
        };
        const double flops = 2.0 * s.M * s.N * s.K;
        const double bytes = (double)bytes_set;
        printf("%-28s  %6.1f MB  hbm-floor %5.1f us (6 TB/s)  mfma-floor %5.1f us\n", tag, bytes / 1e6, bytes / 6e12 * 1e6, flops / 2.5e15 * 1e6);
        for (size_t vi = 0; vi < vars.size(); ++vi) {
            Variant& v = vars[vi];
            if (vfilter[0] && vi > 0 && !strstr(v.name.c_str(), vfilter)) continue;
            // correctness on set 0 against the baseline result
            CK(hipMemsetAsync(sets[0].C, 0, (size_t)s.M * ldc * 2, st));
            if (!v.run(s, sets[0], ldc, epi_for(sets[0]), d_bias, d_rs, st)) { (void)hipGetLastError(); continue; }
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { printf("    %-18s FAILED: %s\n", v.name.c_str(), hipGetErrorString(e)); return 1; }
            float hd[2] = {0, 0};
            if (vi == 0) CK(hipMemcpyAsync(Cref, sets[0].C, (size_t)s.M * ldc * 2, hipMemcpyDeviceToDevice, st));
            else {
                CK(hipMemsetAsync(d_diff, 0, 8, st));
                hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, st, sets[0].C, Cref, s.M, s.N, ldc, d_diff);
                CK(hipMemcpyAsync(hd, d_diff, 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            }
            // timing: rotate over the sets
            const int reps = 3 * nset;
            for (int i = 0; i < nset; ++i) v.run(s, sets[i % nset], ldc, epi_for(sets[i % nset]), d_bias, d_rs, st);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e30f, tot = 0.f;
            for (int round = 0; round < 3; ++round) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < nset; ++i) v.run(s, sets[i], ldc, epi_for(sets[i]), d_bias, d_rs, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / nset); tot += ms / nset;
            }
            (void)reps;
            const float us = tot / 3 * 1e3f;
            printf("    %-18s %7.1f us (best %6.1f)  %6.0f TF  %5.2f TB/s   maxdiff %.3g (ref max %.3g)%s\n", v.name.c_str(), us, best * 1e3f, flops / us / 1e6, bytes / us / 1e6,
                   hd[0], hd[1], (vi > 0 && !g_dbg && !AP_DMA_ABL && hd[0] > 0.02f * hd[1] + 1e-3f) ? "  <-- MISMATCH" : "");
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        for (Set& b : sets) { hipFree(b.A); hipFree(b.B); hipFree(b.C); if (b.R) hipFree(b.R); if (b.H) hipFree(b.H); }
        hipFree(d_bias); hipFree(d_rs); hipFree(Cref);
        fflush(stdout);
    }
    return 0;
}
