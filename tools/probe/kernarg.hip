// does a kernel take more than 4 KB of by-value arguments on this runtime?  (the grouped weight-gradient launches carry their problem
// table as an argument; 48 problems need ~5.5 KB)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(Big<N> b, int* out) { int s = 0; for (int i = threadIdx.x; i < N; i += 64) s += b.v[i]; atomicAdd(out, s); }
template <int N> void run() {
    Big<N> b; long want = 0;
    for (int i = 0; i < N; ++i) { b.v[i] = i % 7; want += i % 7; }
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    int got = 0; hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
    printf("kernarg %5zu bytes: launch %s, sync %s, sum %d (want %ld)\n", sizeof(b) + 8, hipGetErrorString(e), hipGetErrorString(e2), got, want);
}
int main() { run<1000>(); run<1500>(); run<2000>(); run<4000>(); return 0; }
