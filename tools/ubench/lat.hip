// micro-benchmarks that calibrate the latency model used for the update kernels (run on the MI355X box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <thread>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void k_empty(float* o){ if(threadIdx.x==9999) o[0]=1; }
__global__ void k_fma(float* o, int n){ float a=threadIdx.x*1e-3f, b=1.0001f; for(int i=0;i<n;++i) a=a*b+0.5f; o[threadIdx.x]=a; }
__global__ void k_clock(unsigned long long* o, int n){ float a=threadIdx.x*1e-3f; unsigned long long t0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime(); for(int i=0;i<n;++i) a=a*1.0001f+0.5f; unsigned long long t1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime(); if(threadIdx.x==0){o[0]=t1-t0;o[1]=r1-r0;} if(a==123.f) o[2]=1; }
__global__ void k_chase(const int* p, int* o, int n){ int i=threadIdx.x; for(int k=0;k<n;++k) i=p[i]; o[threadIdx.x]=i; }
__global__ void k_chase_clock(const int* p, unsigned long long* o, int n){ int i=0; unsigned long long t0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime(); for(int k=0;k<n;++k) i=p[i]; unsigned long long t1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime(); o[0]=t1-t0;o[1]=r1-r0; o[2]=i; }
static float tm(hipEvent_t a, hipEvent_t b){ float ms; hipEventElapsedTime(&ms,a,b); return ms*1e3f; }
int main(){
  float* d; CK(hipMalloc(&d, 1<<20)); unsigned long long* dc; CK(hipMalloc(&dc, 64)); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // 1. back-to-back empty kernels
  for(int w=0;w<100;++w) hipLaunchKernelGGL(k_empty,dim3(1),dim3(64),0,0,d); hipDeviceSynchronize();
  hipEventRecord(e0); for(int i=0;i<1000;++i) hipLaunchKernelGGL(k_empty,dim3(256),dim3(256),0,0,d); hipEventRecord(e1); hipDeviceSynchronize();
  printf("empty kernel back-to-back: %.2f us each\n", tm(e0,e1)/1000);
  // 2. clock: busy vs after idle
  unsigned long long h[3];
  for(int rep=0;rep<3;++rep){ hipLaunchKernelGGL(k_clock,dim3(1),dim3(64),0,0,dc,20000); hipMemcpy(h,dc,24,hipMemcpyDeviceToHost); printf("clock (1 wave, 20k dependent fma): %llu cycles, %.2f us -> %.0f MHz, %.1f cyc/fma\n", h[0], h[1]/100.0, h[0]/(h[1]/100.0), h[0]/20000.0); }
  std::this_thread::sleep_for(std::chrono::milliseconds(200));
  hipLaunchKernelGGL(k_clock,dim3(1),dim3(64),0,0,dc,2000); hipMemcpy(h,dc,24,hipMemcpyDeviceToHost); printf("after 200 ms idle, 2k fma: %llu cycles %.2f us -> %.0f MHz\n", h[0], h[1]/100.0, h[0]/(h[1]/100.0));
  // 3. sparse short kernels (like the bench loop): 1-wave kernels with 50 us host gaps
  for(int i=0;i<200;++i){ hipLaunchKernelGGL(k_clock,dim3(1),dim3(64),0,0,dc,2000); hipDeviceSynchronize(); }
  hipMemcpy(h,dc,24,hipMemcpyDeviceToHost); printf("sparse launches, 2k fma: %llu cycles %.2f us -> %.0f MHz\n", h[0], h[1]/100.0, h[0]/(h[1]/100.0));
  // 4. dependent global loads (pointer chase): L2-resident (64 KB) and HBM-ish (256 MB)
  for(size_t bytes : {size_t(64)<<10, size_t(256)<<20}){ size_t n=bytes/4; std::vector<int> hp(n); size_t stride= (n>65536)? 1000003 : 257; for(size_t i=0;i<n;++i) hp[i]=(int)((i+stride)%n); int* dp; CK(hipMalloc(&dp,bytes)); hipMemcpy(dp,hp.data(),bytes,hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chase_clock,dim3(1),dim3(1),0,0,dp,dc,2000); hipMemcpy(h,dc,24,hipMemcpyDeviceToHost); hipLaunchKernelGGL(k_chase_clock,dim3(1),dim3(1),0,0,dp,dc,2000); hipMemcpy(h,dc,24,hipMemcpyDeviceToHost);
    printf("pointer chase over %zu KB: %.0f cycles / %.0f ns per dependent load (clock %.0f MHz)\n", bytes>>10, h[0]/2000.0, h[1]*10.0/2000.0, h[0]/(h[1]/100.0)); hipFree(dp); }
  // 5. kernel duration of a 1-block kernel with ~3k dependent FMA (compare with rocprof durations)
  hipEventRecord(e0); for(int i=0;i<200;++i) hipLaunchKernelGGL(k_fma,dim3(1),dim3(128),0,0,d,3000); hipEventRecord(e1); hipDeviceSynchronize(); printf("1-block 3k-fma kernel back-to-back: %.2f us each\n", tm(e0,e1)/200);
  hipEventRecord(e0); for(int i=0;i<200;++i) hipLaunchKernelGGL(k_fma,dim3(256),dim3(256),0,0,d,3000); hipEventRecord(e1); hipDeviceSynchronize(); printf("256-block 3k-fma kernel back-to-back: %.2f us each\n", tm(e0,e1)/200);
  return 0; }
