// probe: semantics of __builtin_amdgcn_fdot2_f32_bf16 (v_dot2c_f32_bf16) on gfx950: single instruction and a
// dependent chain of 16 (one 32-channel dot product) against an fp64 host reference
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
__device__ __forceinline__ float dot2_asm(unsigned a, unsigned b, float c) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
__global__ void k3(const u4* a, const u4* b, float* o) {
    const int t = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u4 x = a[t * 4 + c], y = b[t * 4 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) s = dot2_asm(x[j], y[j], s);
    }
    o[t] = s;
}
template <int NACC>
__global__ void k2(const u4* a, const u4* b, float* o) {
    const int t = threadIdx.x;
    float s[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) s[i] = 0.f;
    u4 x[4], y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = a[t * 4 + c]; y[c] = b[t * 4 + c]; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        s[i % NACC] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[i >> 2][i & 3]), __builtin_bit_cast(bf2, y[i >> 2][i & 3]), s[i % NACC], false);
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += s[i];
    o[t] = r;
}
__global__ void k(const u4* a, const u4* b, float* o) {
    const int t = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u4 x = a[t * 4 + c], y = b[t * 4 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[j]), __builtin_bit_cast(bf2, y[j]), s, false);
    }
    o[t] = s;
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int T = 64;
    unsigned short ha[T * 32], hb[T * 32]; float ho[T];
    srand(1);
    for (int i = 0; i < T * 32; ++i) { ha[i] = f2bf((rand() / (float)RAND_MAX - 0.5f) * 4.f); hb[i] = f2bf((rand() / (float)RAND_MAX - 0.5f) * 4.f); }
    void *da, *db; float* d_o;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&d_o, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(T), 0, 0, (const u4*)da, (const u4*)db, d_o);
    hipMemcpy(ho, d_o, sizeof(ho), hipMemcpyDeviceToHost);
    double worst = 0, ref_norm = 0, err_norm = 0;
    for (int t = 0; t < T; ++t) {
        double r = 0; for (int i = 0; i < 32; ++i) r += (double)bf2f(ha[t * 32 + i]) * bf2f(hb[t * 32 + i]);
        worst = fmax(worst, fabs(ho[t] - r)); ref_norm += r * r; err_norm += (ho[t] - r) * (ho[t] - r);
        if (t < 4) printf("lane %d: gpu %.6f  ref %.6f\n", t, ho[t], r);
    }
    printf("chain of 16 dot2c: rel-L2 error %.3e, worst abs %.3e\n", sqrt(err_norm / ref_norm), worst);
    for (int v = 0; v < 4; ++v) {
        if (v == 3) hipLaunchKernelGGL(k3, dim3(1), dim3(T), 0, 0, (const u4*)da, (const u4*)db, d_o);
        if (v == 0) hipLaunchKernelGGL(k2<2>, dim3(1), dim3(T), 0, 0, (const u4*)da, (const u4*)db, d_o);
        if (v == 1) hipLaunchKernelGGL(k2<4>, dim3(1), dim3(T), 0, 0, (const u4*)da, (const u4*)db, d_o);
        if (v == 2) hipLaunchKernelGGL(k2<8>, dim3(1), dim3(T), 0, 0, (const u4*)da, (const u4*)db, d_o);
        hipMemcpy(ho, d_o, sizeof(ho), hipMemcpyDeviceToHost);
        ref_norm = err_norm = 0;
        for (int t = 0; t < T; ++t) {
            double r = 0; for (int i = 0; i < 32; ++i) r += (double)bf2f(ha[t * 32 + i]) * bf2f(hb[t * 32 + i]);
            ref_norm += r * r; err_norm += (ho[t] - r) * (ho[t] - r);
        }
        printf("%s: rel-L2 error %.3e\n", v == 0 ? "2 interleaved accumulators" : (v == 1 ? "4 interleaved accumulators" : (v == 2 ? "8 interleaved accumulators" : "inline-asm chain of 16")), sqrt(err_norm / ref_norm));
    }
    return 0;
}
