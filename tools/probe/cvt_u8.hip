// how does v_cvt_pk_u8_f32 round?  (the 8-bit gelu' codes of csrc/gemm_epi.h: with round-to-nearest the v_rndne_f32 in front of it can go)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* y, int n) {
    const int i = threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main() {
    const float h[] = {0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, -0.3f, -3.f, 254.4f, 254.5f, 255.4f, 255.7f, 300.f, 127.49f, 127.51f, 0.999f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* dx; unsigned* dy; unsigned out[32];
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, n);
    hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], out[i]);
    return 0;
}
