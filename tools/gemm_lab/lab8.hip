// Calibration bench for the 8-phase GEMM (GPU box only): k_gemm_nt_8p against the shipped ap_gemm_nt, uniform random
// [-1, 1) operands, operand sets rotated over > 256 MiB where the shape is small.  Build: make -C tools/gemm_lab lab8
// usage: lab8 [M N K [epi]] ...   (no arguments: 4096^3, 8192^3 and the VOLO-D1 shapes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <type_traits>
// the library exports the same kernel template: give this translation unit's instantiations their own symbol (with equal names the
// loader resolves the host stub to ONE of the two definitions and an ablation build silently times the library's kernel)
#define k_gemm_nt_8p k_gemm_nt_8p_lab
#include "../../autoprog_amd/csrc/gemm8p.h"

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

struct Shape { int M, N, K; std::string epi; };
static int g_ncu = 256;

static int g_generic = 0;  // G8_GENERIC=1: the run-time-flavour instantiation for everything
template <int NT1, int EF>
static void lab_go(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)k_gemm_nt_8p<NT1, EF>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_BYTES)); attr = true; }
    hipLaunchKernelGGL((k_gemm_nt_8p<NT1, EF>), dim3(grid), dim3(512), G8_LDS_BYTES, st, ga, ep);
}
template <int NT1>
static void lab_pick(const G8Args& ga, const EpiArgs& ep, int grid, hipStream_t st) {
    if (g_generic) { lab_go<NT1, -1>(ga, ep, grid, st); return; }
    switch (g8_flavour(ep)) {
        case 0: lab_go<NT1, 0>(ga, ep, grid, st); break;
        case G8_BIAS: lab_go<NT1, G8_BIAS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU: lab_go<NT1, G8_BIAS | G8_GELU>(ga, ep, grid, st); break;
        case G8_BIAS | G8_GELU | G8_RS: lab_go<NT1, G8_BIAS | G8_GELU | G8_RS>(ga, ep, grid, st); break;
        case G8_DGELU: lab_go<NT1, G8_DGELU>(ga, ep, grid, st); break;
        case G8_DGELU | G8_RS: lab_go<NT1, G8_DGELU | G8_RS>(ga, ep, grid, st); break;
        case G8_MUL: lab_go<NT1, G8_MUL>(ga, ep, grid, st); break;
        case G8_MUL | G8_RS: lab_go<NT1, G8_MUL | G8_RS>(ga, ep, grid, st); break;
        case G8_BIAS | G8_RS | G8_RES: lab_go<NT1, G8_BIAS | G8_RS | G8_RES>(ga, ep, grid, st); break;
        default: lab_go<NT1, -1>(ga, ep, grid, st); break;
    }
}
static int g_bn = 0;       // 0: 192 when it divides N, else 256
static void launch8(const bf16_t* A, const bf16_t* B, bf16_t* C, int ldc, const Shape& s, const EpiArgs& ep, int grid_cap, hipStream_t st) {
    G8Args ga;
    ga.A = A; ga.lda = s.K; ga.B = B; ga.ldb = s.K; ga.C = C; ga.ldc = ldc; ga.M = s.M; ga.N = s.N; ga.K = s.K;
    const int bn = g_bn ? g_bn : (s.N % 192 == 0 ? 192 : 256);
    ga.tiles_n = (s.N + bn - 1) / bn; ga.ntiles = ((s.M + 255) / 256) * ga.tiles_n;
    const int grid = std::min((ga.ntiles + 7) / 8 * 8, grid_cap);      // a multiple of 8: the kernel deals tiles per XCD label
    if (bn == 192) lab_pick<1>(ga, ep, grid, st); else lab_pick<2>(ga, ep, grid, st);
}

int main(int argc, char** argv) {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); g_ncu = pr.multiProcessorCount;
    printf("# device %s, %d CUs\n", pr.name, g_ncu);
    if (getenv("G8_GENERIC")) g_generic = atoi(getenv("G8_GENERIC"));
    if (getenv("G8_BN")) g_bn = atoi(getenv("G8_BN"));
    std::vector<Shape> shapes;
    for (int i = 1; i + 2 < argc; i += 4) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), i + 3 < argc ? argv[i + 3] : "plain"});
    if (shapes.empty()) shapes = {
        {4096, 4096, 4096, "plain"}, {8192, 8192, 8192, "plain"},
        {25088, 1152, 384, "plain"}, {25088, 1152, 384, "gelurs"}, {25088, 1152, 384, "dgrs"}, {25088, 384, 1152, "plain"},
        {25088, 384, 1152, "brr"}, {25088, 384, 384, "bias"}, {25088, 384, 384, "brr"},
        {100352, 576, 192, "gelu"}, {100352, 576, 192, "dgelu"}, {100352, 192, 576, "brr"}, {100352, 192, 192, "plain"}, {100352, 192, 192, "brr"},
        {25216, 768, 384, "plain"}, {300, 192, 192, "res"}, {257, 1000, 384, "bias"}, {1000, 576, 192, "gelurs"}, {25216, 384, 768, "brr"},
    };
    const int grid_cap = getenv("G8_GRID") ? atoi(getenv("G8_GRID")) : g_ncu;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* dd; CK(hipMalloc(&dd, 8));
    printf("%-28s %10s %10s %9s %9s %10s\n", "shape", "8p us", "base us", "8p TF", "base TF", "maxdiff/ref");
    for (const Shape& s : shapes) {
        const int ldc = (s.N + 7) / 8 * 8;
        const size_t bytesA = (size_t)s.M * s.K * 2, bytesB = (size_t)s.N * s.K * 2, bytesC = (size_t)s.M * ldc * 2;
        const size_t per = bytesA + bytesB + 4 * bytesC;
        int nset = (int)std::min<size_t>(8, std::max<size_t>(1, ((size_t)640 << 20) / per + 1));
        std::vector<bf16_t*> As(nset), Bs(nset), Cs(nset), Rs(nset), Hs(nset), C2(nset);
        for (int i = 0; i < nset; ++i) {
            CK(hipMalloc(&As[i], bytesA)); CK(hipMalloc(&Bs[i], bytesB)); CK(hipMalloc(&Cs[i], bytesC)); CK(hipMalloc(&C2[i], bytesC));
            CK(hipMalloc(&Rs[i], bytesC)); CK(hipMalloc(&Hs[i], bytesC));
            const float sc = 1.0f;
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, As[i], (size_t)s.M * s.K, 17u + i, sc);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, Bs[i], (size_t)s.N * s.K, 91u + i, sc * 0.125f);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, Rs[i], (size_t)s.M * ldc, 5u + i, 1.0f);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, Hs[i], (size_t)s.M * ldc, 7u + i, 2.0f);
            CK(hipMemsetAsync(Cs[i], 0xff, bytesC, st)); CK(hipMemsetAsync(C2[i], 0x7f, bytesC, st));
        }
        float* bias; CK(hipMalloc(&bias, (size_t)ldc * 4));
        hipLaunchKernelGGL(k_fillf, dim3(64), dim3(256), 0, st, bias, (size_t)ldc, 3u, 0.5f);
        float* rsc; CK(hipMalloc(&rsc, (size_t)(s.M / 196 + 2) * 4));
        hipLaunchKernelGGL(k_fillf, dim3(64), dim3(256), 0, st, rsc, (size_t)(s.M / 196 + 2), 11u, 1.0f);
        auto mk = [&](int i, bf16_t* pre) {
            EpiArgs ep; memset(&ep, 0, sizeof(ep));
            if (s.epi == "bias" || s.epi == "gelu") ep.bias = bias;
            if (s.epi == "gelu") { ep.gelu = 1; ep.preact = pre; }
            if (s.epi == "dgelu") ep.dgelu_of = Hs[i];
            if (s.epi == "res") { ep.residual = Rs[i]; ep.ldr = ldc; }
            if (s.epi == "brr") { ep.bias = bias; ep.row_scale = rsc; ep.rows_per_scale = 196; ep.residual = Rs[i]; ep.ldr = ldc; }
            if (s.epi == "dgrs") { ep.dgelu_of = Hs[i]; ep.row_scale = rsc; ep.rows_per_scale = 196; }
            if (s.epi == "mulrs") { ep.mul_by = Hs[i]; ep.row_scale = rsc; ep.rows_per_scale = 196; }
            if (s.epi == "gelu2rs") { ep.bias = bias; ep.gelu = 2; ep.preact = pre; ep.row_scale = rsc; ep.rows_per_scale = 196; }
            if (s.epi == "gelurs") { ep.bias = bias; ep.gelu = 1; ep.preact = pre; ep.row_scale = rsc; ep.rows_per_scale = 196; }
            return ep;
        };
        auto base = [&](int i) {
            EpiArgs ep = mk(i, Hs[i]);
            ap_gemm_epilogue e; memset(&e, 0, sizeof(e));
            e.bias = ep.bias; e.gelu = ep.gelu; e.preact_out = ep.preact; e.dgelu_of = ep.dgelu_of; e.residual = ep.residual; e.ldr = ep.ldr;
            e.row_scale = ep.row_scale; e.rows_per_scale = ep.rows_per_scale; e.mul_by = ep.mul_by;
            return ap_gemm_nt(As[i], s.K, Bs[i], s.K, C2[i], ldc, s.M, s.N, s.K, &e, st) == 0;
        };
        // correctness on set 0 (the gelu case writes its pre-activation to R so that H stays the dgelu input of other runs)
        launch8(As[0], Bs[0], Cs[0], ldc, s, mk(0, Rs[0]), grid_cap, st);
        CK(hipGetLastError());
        bool okb = base(0);
        CK(hipMemsetAsync(dd, 0, 8, st));
        hipLaunchKernelGGL(k_diff, dim3(512), dim3(256), 0, st, Cs[0], C2[0], s.M, s.N, ldc, dd);
        float hd[2]; CK(hipMemcpyAsync(hd, dd, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        const int reps = std::max(3, (int)(2e10 / (2.0 * s.M * s.N * s.K)) * 2 + 3);
        auto timeit = [&](auto fn) {
            for (int i = 0; i < nset; ++i) fn(i);
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) fn(r % nset);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.f / reps;
        };
        float t8 = 0, tb = 0;
        for (int round = 0; round < 3; ++round) {
            const float a = timeit([&](int i) { launch8(As[i], Bs[i], Cs[i], ldc, s, mk(i, Rs[i]), grid_cap, st); });
            const float b = okb ? timeit([&](int i) { base(i); }) : 0.f;
            t8 = round ? std::min(t8, a) : a; tb = round ? std::min(tb, b) : b;
        }
        const double fl = 2.0 * s.M * s.N * s.K;
        char nm[64]; snprintf(nm, sizeof nm, "%dx%dx%d %s", s.M, s.N, s.K, s.epi.c_str());
        printf("%-28s %10.1f %10.1f %9.0f %9.0f %10.4g/%.3g\n", nm, t8, tb, fl / t8 * 1e-6, tb > 0 ? fl / tb * 1e-6 : 0.0, hd[0], hd[1]);
        fflush(stdout);
        for (int i = 0; i < nset; ++i) { hipFree(As[i]); hipFree(Bs[i]); hipFree(Cs[i]); hipFree(C2[i]); hipFree(Rs[i]); hipFree(Hs[i]); }
        hipFree(bias); hipFree(rsc);
    }
    return 0;
}
