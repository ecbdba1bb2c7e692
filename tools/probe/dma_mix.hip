// Probe: how fast can one CU pull a stream into LDS -- by LDS-DMA (global_load_lds_dwordx4), by register staging
// (global_load_dwordx4 + ds_write_b128), or by both at once?  One 512-thread workgroup per CU, every wave keeps DEPTH 1-KB
// instructions (8 rows x 128 B, rows 4 KB apart: the shape of a K-tile piece of a row-major operand) in flight.
// window 2 MB: every CU reads the same L2-resident bytes; 512 MB: beyond the Infinity Cache.
// build: hipcc -O3 --offload-arch=gfx950 dma_mix.hip -o dma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// MODE 0: DMA only; 1: registers only; 2: alternate (even instruction DMA, odd instruction registers)
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512, 2) k_pull(const unsigned char* __restrict__ src, size_t window, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* dst = smem + wave * (DEPTH * 1024);
    const size_t lane_off = (size_t)(lane >> 3) * 4096 + (lane & 7) * 16;
    const size_t base = ((size_t)blockIdx.x * 8 + wave) * 32768 * 4;
    u32x4 r[DEPTH];
    unsigned acc = 0;
    size_t j = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d, ++j) {
            const unsigned char* p = src + (base + (j >> 5) * 32768 + (j & 31) * 128) % window + lane_off;
            const bool dma = MODE == 0 || (MODE == 2 && !(d & 1));
            if (dma) __builtin_amdgcn_global_load_lds(GLB(p), LDSP(dst + d * 1024), 16, 0, 0);
            else r[d] = *reinterpret_cast<const u32x4*>(p);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool dma = MODE == 0 || (MODE == 2 && !(d & 1));
            if (!dma) *reinterpret_cast<u32x4*>(dst + d * 1024 + lane * 16) = r[d];
        }
        acc += *reinterpret_cast<volatile unsigned*>(dst + lane * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int DEPTH>
static void run(const unsigned char* src, size_t window, int ncu, unsigned* sink, const char* name) {
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)k_pull<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_pull<MODE, DEPTH>), dim3(ncu), dim3(512), 8 * DEPTH * 1024, 0, src, window, iters, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)ncu * 8 * DEPTH * 1024.0 * iters;
        if (rep) printf("%-24s depth %2d  window %6.1f MB  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, DEPTH, window / 1048576.0, bytes / ms * 1e-6 / ncu, bytes / ms * 1e-9);
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    const size_t total = (size_t)1 << 30;
    unsigned char* src; CK(hipMalloc(&src, total + (1 << 20))); CK(hipMemset(src, 1, total + (1 << 20)));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    for (size_t window : {(size_t)2 << 20, (size_t)512 << 20}) {
        run<0, 4>(src, window, ncu, sink, "LDS-DMA");
        run<0, 8>(src, window, ncu, sink, "LDS-DMA");
        run<0, 16>(src, window, ncu, sink, "LDS-DMA");
        run<1, 8>(src, window, ncu, sink, "registers + ds_write");
        run<1, 16>(src, window, ncu, sink, "registers + ds_write");
        run<2, 8>(src, window, ncu, sink, "alternating");
        run<2, 16>(src, window, ncu, sink, "alternating");
    }
    return 0;
}
