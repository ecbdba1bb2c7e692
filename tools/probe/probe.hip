// Hardware probe for gfx950: dumps the lane maps this repo's kernels rely on
// (MFMA operand/accumulator layouts, ds_read_tr16_b64, global_load_lds) so that
// kernel index math is pinned against real hardware, not documentation.
// Build: hipcc --offload-arch=gfx950 -O2 probe.hip -o probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} }while(0)

__device__ inline __bf16 tobf(float f){ return (__bf16)f; }

// ---- 1. MFMA 16x16x32: A[16][32], B[32][16] row-major float inputs (exact small ints) ----
__global__ void mfma16(const float* A, const float* B, float* C){
  int l = threadIdx.x; int r = l & 15, g = l >> 4;
  b8 a, b;
  for(int j=0;j<8;j++){ a[j] = tobf(A[r*32 + 8*g + j]); b[j] = tobf(B[(8*g+j)*16 + r]); }
  f4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int i=0;i<4;i++) C[(4*g+i)*16 + r] = c[i];   // assumed: col=lane&15,row=4*(lane>>4)+i
}
// ---- 2. MFMA 32x32x16: A[32][16], B[16][32] ----
__global__ void mfma32(const float* A, const float* B, float* C){
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  b8 a, b;
  for(int j=0;j<8;j++){ a[j] = tobf(A[r*16 + 8*h + j]); b[j] = tobf(B[(8*h+j)*32 + r]); }
  f16v c; for(int i=0;i<16;i++) c[i]=0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,c,0,0,0);
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; C[row*32 + r] = c[i]; }
}
// ---- 3. ds_read_tr16_b64: LDS holds T[64 rows][64 cols] of short = row*64+col ----
// assumed: within each 16-lane group, lane i=4q+p supplies &T[r0+q][c0+4p]; lane i receives
// T[r0+0..3][c0+i].
__global__ void trprobe(short* out){
  __shared__ __attribute__((aligned(16))) short T[64*64];
  for(int i=threadIdx.x;i<64*64;i+=64) T[i]=(short)i;
  __syncthreads();
  int l = threadIdx.x; int g = l>>4, i = l&15, q=i>>2, p=i&3;
  int r0 = 8*g, c0 = 16;     // group g reads block rows 8g..8g+3, cols 16..31
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(&T[(r0+q)*64 + c0 + 4*p]));
  for(int k=0;k<4;k++) out[l*4+k]=v[k];
}
// ---- 4. global_load_lds 16B: each lane supplies its own global src; LDS dest = base + lane*16 ? ----
__global__ void gldsprobe(const int* src, int* out){
  __shared__ __attribute__((aligned(16))) int L[64*4*2];
  for(int i=threadIdx.x;i<64*4*2;i+=64) L[i]=-1;
  __syncthreads();
  int l = threadIdx.x;
  // lane l reads 16B from src + ((63-l)*4) ints  (reversed) ; dest passed as wave-uniform base
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (63-l)*4),
                                   (__attribute__((address_space(3))) void*)(&L[0]), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for(int i=threadIdx.x;i<64*4*2;i+=64) out[i]=L[i];
}
// ---- 5. streaming copy bandwidth ----
__global__ void copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x*blockDim.x;
  for(; i<n; i+=st) b[i]=a[i];
}
// ---- 6. XCC id per block ----
__global__ void xccprobe(int* out){
  if(threadIdx.x==0){ unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); out[blockIdx.x]=(int)(v&0xf); }
}

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s arch %s CUs %d clock %d kHz memclk %d kHz L2 %d bytes smem/block %zu regs/block %d\n",
         p.name,p.gcnArchName,p.multiProcessorCount,p.clockRate,p.memoryClockRate,p.l2CacheSize,p.sharedMemPerBlock,p.regsPerBlock);
  // 1
  { std::vector<float> A(16*32),B(32*16),C(16*16),R(16*16,0.f);
    for(int i=0;i<16;i++)for(int k=0;k<32;k++)A[i*32+k]=(float)((i*3+k*5)%7-3);
    for(int k=0;k<32;k++)for(int j=0;j<16;j++)B[k*16+j]=(float)((k*2+j*7)%5-2);
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){float s=0;for(int k=0;k<32;k++)s+=A[i*32+k]*B[k*16+j];R[i*16+j]=s;}
    float *dA,*dB,*dC; CK(hipMalloc(&dA,A.size()*4));CK(hipMalloc(&dB,B.size()*4));CK(hipMalloc(&dC,C.size()*4));
    CK(hipMemcpy(dA,A.data(),A.size()*4,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),B.size()*4,hipMemcpyHostToDevice));
    mfma16<<<1,64>>>(dA,dB,dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(),dC,C.size()*4,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<256;i++) if(C[i]!=R[i]) bad++;
    printf("[mfma16x16x32] mismatches=%d (0 => assumed A/B/C lane maps correct)\n",bad); }
  // 2
  { std::vector<float> A(32*16),B(16*32),C(32*32),R(32*32,0.f);
    for(int i=0;i<32;i++)for(int k=0;k<16;k++)A[i*16+k]=(float)((i*3+k*5)%7-3);
    for(int k=0;k<16;k++)for(int j=0;j<32;j++)B[k*32+j]=(float)((k*2+j*7)%5-2);
    for(int i=0;i<32;i++)for(int j=0;j<32;j++){float s=0;for(int k=0;k<16;k++)s+=A[i*16+k]*B[k*32+j];R[i*32+j]=s;}
    float *dA,*dB,*dC; CK(hipMalloc(&dA,A.size()*4));CK(hipMalloc(&dB,B.size()*4));CK(hipMalloc(&dC,C.size()*4));
    CK(hipMemcpy(dA,A.data(),A.size()*4,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),B.size()*4,hipMemcpyHostToDevice));
    mfma32<<<1,64>>>(dA,dB,dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(),dC,C.size()*4,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<1024;i++) if(C[i]!=R[i]) bad++;
    printf("[mfma32x32x16] mismatches=%d\n",bad); }
  // 3
  { short* d; CK(hipMalloc(&d,64*4*2)); std::vector<short> h(256);
    trprobe<<<1,64>>>(d); CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(),d,512,hipMemcpyDeviceToHost));
    int bad=0;
    for(int l=0;l<64;l++){ int g=l>>4,i=l&15; for(int k=0;k<4;k++){ int exp=(8*g+k)*64+16+i; if(h[l*4+k]!=exp) bad++; } }
    printf("[ds_read_tr16_b64] mismatches=%d vs assumed (lane i gets T[r0+k][c0+i])\n",bad);
    for(int l=0;l<64;l++){ printf("  lane %2d:",l); for(int k=0;k<4;k++) printf(" (r%d,c%d)",h[l*4+k]/64,h[l*4+k]%64); printf("\n"); } }
  // 4
  { int *s,*o; CK(hipMalloc(&s,64*16)); CK(hipMalloc(&o,64*4*2*4)); std::vector<int> hs(256),ho(512);
    for(int i=0;i<256;i++)hs[i]=i; CK(hipMemcpy(s,hs.data(),1024,hipMemcpyHostToDevice));
    gldsprobe<<<1,64>>>(s,o); CK(hipDeviceSynchronize()); CK(hipMemcpy(ho.data(),o,2048,hipMemcpyDeviceToHost));
    int bad=0; for(int l=0;l<64;l++)for(int k=0;k<4;k++) if(ho[l*4+k]!=(63-l)*4+k) bad++;
    printf("[global_load_lds x16] mismatches=%d vs assumed (LDS[base+16*lane] <- lane's src)\n",bad);
    printf("  first 16 ints:"); for(int i=0;i<16;i++)printf(" %d",ho[i]); printf("\n"); }
  // 5
  { size_t n = (size_t)1<<30; float4 *a,*b; CK(hipMalloc(&a,n)); CK(hipMalloc(&b,n)); CK(hipMemset(a,1,n));
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for(int it=0;it<2;it++) copy16<<<2048,256>>>(a,b,n/16);
    CK(hipEventRecord(e0)); for(int it=0;it<10;it++) copy16<<<2048,256>>>(a,b,n/16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("[copy 1GiB r+w] %.3f ms/iter => %.2f TB/s (read+write)\n",ms/10, 2.0*n/(ms/10*1e-3)/1e12); }
  // 6
  { int* d; CK(hipMalloc(&d,64*4)); std::vector<int> h(64); xccprobe<<<64,64>>>(d); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(),d,256,hipMemcpyDeviceToHost)); printf("[xcc id of blocks 0..31]"); for(int i=0;i<32;i++)printf(" %d",h[i]); printf("\n"); }
  return 0;
}
