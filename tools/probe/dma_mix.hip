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
// hbm_every: workgroup b pulls from the large window when b % hbm_every == 0, else from the 2 MB L2-resident one (0: all from `window`)
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512, 2) k_pull(const unsigned char* __restrict__ src, size_t window, int iters, unsigned* sink, int hbm_every = 0,
                                                  unsigned long long* clocks = nullptr) {
    if (hbm_every) window = (blockIdx.x % hbm_every == 0) ? window : ((size_t)2 << 20);
    const unsigned long long t0 = wall_clock64();
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
    if (clocks && threadIdx.x == 0) clocks[blockIdx.x] = wall_clock64() - t0;
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

// One CU, a quarter of its bytes from memory and three quarters from L2 (a GEMM's stream): MIXED -- every wave issues one memory-side
// instruction in four -- against SEGREGATED -- waves 0, 1 pull only from memory, waves 2..7 only from L2.  A wave's loads retire in
// order, so in the mixed form every L2 hit queues behind a memory access.
template <int DEPTH, bool SEG>
__global__ void __launch_bounds__(512, 2) k_mix(const unsigned char* __restrict__ src, size_t big, int iters, unsigned* sink, unsigned long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* dst = smem + wave * (DEPTH * 1024);
    const size_t lane_off = (size_t)(lane >> 3) * 4096 + (lane & 7) * 16;
    const size_t base = ((size_t)blockIdx.x * 8 + wave) * 32768 * 4;
    const size_t small = (size_t)2 << 20;
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    size_t j = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d, ++j) {
            const bool mem = SEG ? (wave < 2) : ((d & 3) == 0);
            const size_t w = mem ? big : small;
            const unsigned char* p = src + (mem ? small : 0) + (base + (j >> 5) * 32768 + (j & 31) * 128) % w + lane_off;
            __builtin_amdgcn_global_load_lds(GLB(p), LDSP(dst + d * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<volatile unsigned*>(dst + lane * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) clocks[blockIdx.x] = wall_clock64() - t0;
}
template <int DEPTH, bool SEG>
static void run_mix(const unsigned char* src, int ncu, unsigned* sink) {
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)k_mix<DEPTH, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024));
    unsigned long long* clk; CK(hipMalloc(&clk, ncu * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_mix<DEPTH, SEG>), dim3(ncu), dim3(512), 8 * DEPTH * 1024, 0, src, (size_t)512 << 20, iters, sink, clk);
        CK(hipDeviceSynchronize());
    }
    unsigned long long* h = (unsigned long long*)malloc(ncu * 8); CK(hipMemcpy(h, clk, ncu * 8, hipMemcpyDeviceToHost));
    double t = 0; for (int b = 0; b < ncu; ++b) t += h[b];
    printf("%-11s 1/4 memory + 3/4 L2, depth %2d: %6.1f GB/s per CU\n", SEG ? "segregated" : "mixed", DEPTH, 8.0 * DEPTH * 1024.0 * iters / (t / ncu * 10e-9) * 1e-9);
    hipFree(clk); free(h);
}

// The same mixed stream the way a pipelined GEMM issues it: PER instructions per wave and phase, then a counted wait that leaves INFL
// instructions in flight, then NBAR workgroup barriers -- against the free-running loop above.
template <int PER, int INFL, int NBAR, int MEMEVERY = 4>
__global__ void __launch_bounds__(512, 2) k_phased(const unsigned char* __restrict__ src, size_t big, int phases, unsigned* sink, unsigned long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int SLOTS = INFL + PER;
    unsigned char* dst = smem + wave * (SLOTS * 1024);
    const size_t lane_off = (size_t)(lane >> 3) * 4096 + (lane & 7) * 16;
    const size_t base = ((size_t)blockIdx.x * 8 + wave) * 32768 * 4;
    const size_t small = (size_t)2 << 20;
    const unsigned long long t0 = wall_clock64();
    size_t j = 0;
    int slot = 0;
    for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
        for (int d = 0; d < PER; ++d, ++j) {
            const bool mem = MEMEVERY && (j % MEMEVERY) == 0;
            const size_t w = mem ? big : small;
            const unsigned char* p = src + (mem ? small : 0) + (base + (j >> 5) * 32768 + (j & 31) * 128) % w + lane_off;
            __builtin_amdgcn_global_load_lds(GLB(p), LDSP(dst + slot * 1024), 16, 0, 0);
            slot = slot + 1 == SLOTS ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFL) : "memory");
#pragma unroll
        for (int b = 0; b < NBAR; ++b) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*reinterpret_cast<volatile unsigned*>(dst + lane * 4) == 0x12345678u) sink[0] = 1;
    if (threadIdx.x == 0) clocks[blockIdx.x] = wall_clock64() - t0;
}
template <int PER, int INFL, int NBAR, int MEMEVERY = 4>
static void run_phased(const unsigned char* src, int ncu, unsigned* sink) {
    const int phases = 16000 / PER;
    const size_t lds = 8 * (INFL + PER) * 1024;
    CK(hipFuncSetAttribute((const void*)k_phased<PER, INFL, NBAR, MEMEVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned long long* clk; CK(hipMalloc(&clk, ncu * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_phased<PER, INFL, NBAR, MEMEVERY>), dim3(ncu), dim3(512), lds, 0, src, (size_t)512 << 20, phases, sink, clk);
        CK(hipDeviceSynchronize());
    }
    unsigned long long* h = (unsigned long long*)malloc(ncu * 8); CK(hipMemcpy(h, clk, ncu * 8, hipMemcpyDeviceToHost));
    double t = 0; for (int b = 0; b < ncu; ++b) t += h[b];
    printf("phased: %d instr per wave and phase, %2d left in flight, %d barriers: %6.1f GB/s per CU\n", PER, INFL, NBAR, 8.0 * PER * 1024.0 * phases / (t / ncu * 10e-9) * 1e-9);
    hipFree(clk); free(h);
}

// half / a quarter / an eighth of the workgroups pull from memory, the others from L2: what does ONE CU get from memory when the chip's
// memory system is not saturated?  (per-workgroup wall clocks, 100 MHz)
template <int DEPTH>
static void run_split(const unsigned char* src, size_t window, int ncu, unsigned* sink, int every) {
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)k_pull<0, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024));
    unsigned long long* clk; CK(hipMalloc(&clk, ncu * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_pull<0, DEPTH>), dim3(ncu), dim3(512), 8 * DEPTH * 1024, 0, src, window, iters, sink, every, clk);
        CK(hipDeviceSynchronize());
    }
    unsigned long long* h = (unsigned long long*)malloc(ncu * 8); CK(hipMemcpy(h, clk, ncu * 8, hipMemcpyDeviceToHost));
    double tm = 0, tl = 0; int nm = 0, nl = 0;
    for (int b = 0; b < ncu; ++b) { if (b % every == 0) { tm += h[b]; ++nm; } else { tl += h[b]; ++nl; } }
    const double bytes = 8.0 * DEPTH * 1024.0 * iters;
    printf("1 of %d workgroups from memory, depth %2d: memory-side CU %6.1f GB/s (%d CUs, %.2f TB/s together), L2-side CU %6.1f GB/s\n", every, DEPTH,
           bytes / (tm / nm * 10e-9) * 1e-9, nm, bytes / (tm / nm * 10e-9) * 1e-12 * nm, bytes / (tl / nl * 10e-9) * 1e-9);
    hipFree(clk); free(h);
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
    run_mix<8, false>(src, ncu, sink); run_mix<8, true>(src, ncu, sink); run_mix<16, false>(src, ncu, sink); run_mix<16, true>(src, ncu, sink);
    run_phased<2, 8, 0, 0>(src, ncu, sink); run_phased<2, 8, 2, 0>(src, ncu, sink); run_phased<3, 9, 2, 0>(src, ncu, sink); run_phased<4, 8, 2, 0>(src, ncu, sink);
    run_phased<6, 6, 2, 0>(src, ncu, sink); run_phased<7, 7, 2, 0>(src, ncu, sink); run_phased<8, 8, 2, 0>(src, ncu, sink);
    run_phased<2, 8, 0, 4>(src, ncu, sink); run_phased<2, 8, 2, 4>(src, ncu, sink); run_phased<2, 8, 2, 2>(src, ncu, sink); run_phased<3, 9, 2, 3>(src, ncu, sink);
    run_phased<4, 8, 2, 4>(src, ncu, sink); run_phased<6, 6, 2, 3>(src, ncu, sink); run_phased<7, 7, 2, 7>(src, ncu, sink); run_phased<8, 8, 2, 4>(src, ncu, sink);
    run_phased<4, 8, 2, 2>(src, ncu, sink); run_phased<8, 8, 2, 2>(src, ncu, sink);
    for (int every : {2, 4, 8, 32}) { run_split<8>(src, (size_t)512 << 20, ncu, sink, every); run_split<16>(src, (size_t)512 << 20, ncu, sink, every); }
    return 0;
}
