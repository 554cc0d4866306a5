// Microbenchmark: fp32 atomic scatter-add throughput on gfx950 under different placements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ uint32_t hash32(uint32_t x){ x ^= x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

// mode 0: shared table; 1: table copy selected by blockIdx%8; 2: table copy selected by HW XCC id
template<int MODE, int PER_THREAD>
__global__ void scatter(float* table, uint32_t rows_mask, uint32_t rows, int nfeat){
  uint32_t gid = blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t copy = 0;
  if (MODE==1) copy = blockIdx.x & 7;
  if (MODE==2) { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); copy = x & 7; }
  float* t = table + (size_t)copy*rows*nfeat;
  #pragma unroll
  for(int i=0;i<PER_THREAD;i++){
    uint32_t r = hash32(gid*PER_THREAD+i) & rows_mask;
    for(int f=0; f<nfeat; f++) atomicAdd(t + (size_t)r*nfeat + f, 1.0f);
  }
}
template<int PER_THREAD>
__global__ void scatter_lds(float* out, uint32_t rows_mask){
  extern __shared__ float lds[];
  for(int i=threadIdx.x;i<=rows_mask;i+=blockDim.x) lds[i]=0;
  __syncthreads();
  uint32_t gid = blockIdx.x*blockDim.x + threadIdx.x;
  #pragma unroll 4
  for(int i=0;i<PER_THREAD;i++){
    uint32_t r = hash32(gid*PER_THREAD+i) & rows_mask;
    atomicAdd(&lds[r], 1.0f);
  }
  __syncthreads();
  if(threadIdx.x==0) out[blockIdx.x]=lds[0];
}
template<int PER_THREAD, int MODE>
__global__ void lds_ops(float* out){
  extern __shared__ float lds[];
  for(int i=threadIdx.x;i<16384;i+=blockDim.x) lds[i]=0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0;
  #pragma unroll 8
  for(int i=0;i<PER_THREAD;i++){
    // conflict-free: lane l -> distinct consecutive banks; a different 64-float row per iteration and wave
    const int a = ((i*4 + wave) * 64 + lane) & 16383;
    if (MODE==0) atomicAdd(&lds[a], 1.0f);                 // ds_add_f32
    if (MODE==1) lds[a] = lds[a] + 1.0f;                   // plain read-modify-write
    if (MODE==2) acc += lds[a];                            // read only
    if (MODE==3) atomicAdd(&lds[(lane*65 + i) & 16383], 1.0f);   // stride-65: conflict free, scattered rows
    if (MODE==4) atomicAdd(&lds[(lane*2 + (i&1)*128) & 16383], 1.0f);   // 2-way bank conflict
    if (MODE==5) atomicAdd((unsigned*)&lds[a], 1u);                                     // ds_add_u32
    if (MODE==6) acc += __uint_as_float(atomicAdd((unsigned*)&lds[a], 1u));             // ds_add_rtn_u32
    if (MODE==7) acc += __uint_as_float(atomicCAS((unsigned*)&lds[a], 0u, (unsigned)lane)); // ds_cmpst_rtn_b32
    if (MODE==8) atomicAdd((unsigned long long*)&lds[a & 16382], 1ull);                 // ds_add_u64
    if (MODE==9) atomicMax((unsigned*)&lds[a], (unsigned)i);                            // ds_max_u32
    if (MODE==10) acc += atomicAdd(&lds[a], 1.0f);                                      // ds_add_rtn_f32
    if (MODE==11) { float2* q = (float2*)&lds[a & 16382]; float2 v = *q; v.x += 1.f; v.y += 2.f; *q = v; }  // b64 RMW
  }
  __syncthreads();
  if(threadIdx.x==0) out[blockIdx.x]=lds[0]+acc;
}
__global__ void gather(const float2* table, uint32_t rows_mask, float* out){
  uint32_t gid = blockIdx.x*blockDim.x + threadIdx.x;
  float acc=0;
  #pragma unroll
  for(int i=0;i<8;i++){ float2 v = table[hash32(gid*8+i)&rows_mask]; acc+=v.x+v.y; }
  out[gid]=acc;
}
template<typename F> float timeit(F f,int n=5){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize(); hipEventRecord(a); for(int i=0;i<n;i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); return ms/n; }

int main(){
  const uint32_t rows = 1u<<18; const int nfeat=2;
  float* table; CK(hipMalloc(&table, (size_t)8*rows*nfeat*4*4)); CK(hipMemset(table,0,(size_t)8*rows*nfeat*4*4));
  float* out; CK(hipMalloc(&out, 64u<<20));
  const int threads=256; const int per=4; const int64_t total = 8ll*1024*1024; // contributions
  int blocks = total/per/threads;
  for(int nf=1; nf<=2; nf++){
    float t0 = timeit([&]{ hipLaunchKernelGGL((scatter<0,per>), dim3(blocks), dim3(threads),0,0, table, rows-1, rows, nf); });
    float t1 = timeit([&]{ hipLaunchKernelGGL((scatter<1,per>), dim3(blocks), dim3(threads),0,0, table, rows-1, rows, nf); });
    float t2 = timeit([&]{ hipLaunchKernelGGL((scatter<2,per>), dim3(blocks), dim3(threads),0,0, table, rows-1, rows, nf); });
    printf("nfeat=%d  shared %.3f ms (%.1f Gatom/s) | copy=blk%%8 %.3f ms (%.1f) | copy=xcc %.3f ms (%.1f)\n", nf, t0, total*nf/t0/1e6, t1, total*nf/t1/1e6, t2, total*nf/t2/1e6);
  }
  // table size sweep (shared), nfeat=1
  for(uint32_t lg=10; lg<=24; lg+=2){
    uint32_t r = 1u<<lg; if((size_t)r*4 > (size_t)8*rows*nfeat*4*4) break;
    float t0 = timeit([&]{ hipLaunchKernelGGL((scatter<0,per>), dim3(blocks), dim3(threads),0,0, table, r-1, r, 1); });
    printf("rows=2^%u shared nfeat=1: %.3f ms (%.1f Gatom/s)\n", lg, t0, total/t0/1e6);
  }
  // LDS atomics: 16K-entry table per block
  { const int perl=64; int b = total/perl/threads; 
    float t = timeit([&]{ hipLaunchKernelGGL((scatter_lds<perl>), dim3(b), dim3(threads), 16384*4, 0, out, 16383u); });
    printf("LDS atomics (16K floats/block, random): %.3f ms (%.1f Gatom/s)\n", t, total/t/1e6); }
  { const int perl=256; int b = 1024;
    #define LDSRUN(M, name) { float t = timeit([&]{ hipLaunchKernelGGL((lds_ops<perl,M>), dim3(b), dim3(threads), 16384*4, 0, out); }); \
      printf("LDS %s: %.3f ms -> %.1f G lane-ops/s, %.1f cycles per wave-instr per CU (2.4GHz, 256 CU)\n", name, t, (double)b*threads*perl/t/1e6, t*1e-3*2.4e9*256/((double)b*threads*perl/64)); }
    LDSRUN(0, "ds_add_f32 conflict-free"); LDSRUN(1, "plain RMW conflict-free"); LDSRUN(2, "read conflict-free");
    LDSRUN(3, "ds_add_f32 stride-65"); LDSRUN(4, "ds_add_f32 2-way conflict");
    LDSRUN(5, "ds_add_u32"); LDSRUN(6, "ds_add_rtn_u32"); LDSRUN(7, "ds_cmpst_rtn_b32"); LDSRUN(8, "ds_add_u64");
    LDSRUN(9, "ds_max_u32"); LDSRUN(10, "ds_add_rtn_f32"); LDSRUN(11, "plain b64 RMW"); }
  { float t = timeit([&]{ hipLaunchKernelGGL(gather, dim3(total/8/threads), dim3(threads),0,0,(const float2*)table, rows-1, out); });
    printf("gather float2 from 2MiB table: %.3f ms (%.1f Ggather/s)\n", t, total/t/1e6); }
  return 0;
}
