// What does one VALU instruction cost on gfx950, and does it depend on how many waves share the SIMD?
// Workgroups of 256*W threads (W waves per SIMD, one workgroup per CU forced by a 100 KB LDS allocation), every wave runs
// ITERS iterations of an unrolled body of 64 INDEPENDENT instructions on 32 distinct registers (no dependent chains
// shorter than 32 instructions).  Reported: shader cycles (s_memtime) per instruction per SIMD, wall ns, derived clock.
// Second part: roles split by wave -- waves 0..3 bf16 MFMAs only, waves 4..7 (same SIMDs) VALU only, separate loops.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_issue_bench.hip -o tools/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { K_FMA, K_MUL, K_AND, K_PERM, K_EXP, K_PKFMA, K_PKMUL, K_CVTPK, K_MOV, K_ADD3, K_LSHLOR, K_BFE };

template <int KIND>
__device__ __forceinline__ void op(float& x, float a, float b) {
  if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if constexpr (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  if constexpr (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
  if constexpr (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if constexpr (KIND == K_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  if constexpr (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
  if constexpr (KIND == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if constexpr (KIND == K_LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(a));
  if constexpr (KIND == K_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x));
}
template <int KIND>
__device__ __forceinline__ void op2(f32x2& x, f32x2 a, f32x2 b) {
  if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if constexpr (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
}

template <int KIND>
__device__ __forceinline__ float valu_loop(int iters, float a, float b) {
  float s = 0.f;
  if constexpr (KIND == K_PKFMA || KIND == K_PKMUL) {
    f32x2 x[16];
    for (int k = 0; k < 16; k++) x[k] = f32x2{a * (float)(k + threadIdx.x), b};
    const f32x2 aa = {a, a}, bb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 64; k++) op2<KIND>(x[k & 15], aa, bb);
    }
    for (int k = 0; k < 16; k++) s += x[k].x + x[k].y;
  } else {
    float x[32];
    for (int k = 0; k < 32; k++) x[k] = a * (float)(k + threadIdx.x);
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 64; k++) op<KIND>(x[k & 31], a, b);
    }
    for (int k = 0; k < 32; k++) s += x[k];
  }
  return s;
}

template <int MK>  // 0: 32x32x16 bf16 on 4 accumulators, 1: 16x16x32 bf16 on 8 accumulators
__device__ __forceinline__ float mfma_loop(int iters, float a, float b) {
  bf16x8 ba, bb;
  for (int k = 0; k < 8; k++) { ba[k] = (__bf16)a; bb[k] = (__bf16)b; }
  float s = 0.f;
  if constexpr (MK == 0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int m = 0; m < 8; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[m & 3], 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  } else {
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int m = 0; m < 16; m++) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[m & 7], 0, 0, 0);
    }
    for (int i = 0; i < 8; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  }
  return s;
}

template <int KIND>
__global__ void valu_only(int iters, float a, float b, float* out, unsigned long long* cyc) {
  extern __shared__ char lds[];
  const unsigned long long t0 = __builtin_readcyclecounter();
  const float s = valu_loop<KIND>(iters, a, b);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// mode 0: both roles, 1: MFMA waves only, 2: VALU waves only.  VALU iterations scaled by vscale so both take similar time.
template <int KIND, int MK>
__global__ void __launch_bounds__(512) roles(int iters, int viters, float a, float b, float* out, unsigned long long* cyc, int mode) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (mode != 2) s = mfma_loop<MK>(iters, a, b);
  } else {
    if (mode != 1) s = valu_loop<KIND>(viters, a, b);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

// ONE wave per SIMD interleaving bf16 MFMAs with NV independent VALU each
template <int KIND, int NV>
__global__ void interleaved(int iters, float a, float b, float* out, unsigned long long* cyc) {
  extern __shared__ char lds[];
  bf16x8 ba, bb;
  for (int k = 0; k < 8; k++) { ba[k] = (__bf16)a; bb[k] = (__bf16)b; }
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float x[32];
  for (int k = 0; k < 32; k++) x[k] = a * (float)(k + threadIdx.x);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 4; m++) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; k++) op<KIND>(x[(m * NV + k) & 31], a, b);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  for (int k = 0; k < 32; k++) s += x[k];
  if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;
static const int ITERS = 4000;

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

template <int KIND>
static void valu_row(const char* name) {
  for (int w : {1, 2, 4}) {
    const float ms = time_ms([w] {
      hipLaunchKernelGGL((valu_only<KIND>), dim3(256), dim3(256 * w), 100 * 1024, 0, ITERS, 1.0001f, 0.5f, g_out, g_cyc);
    });
    unsigned long long c;
    hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * ITERS;  // instructions per wave
    printf("%-18s %d wave/SIMD: %6.2f ns per instr per SIMD (wall), %6.2f counter ticks per instr per wave, tick = %.3f ns\n",
           name, w, ms * 1e6 / (n * w), (double)c / n, ms * 1e6 / (double)c);
  }
}

template <int KIND, int MK>
static void roles_row(const char* name, int viters) {
  float t[3];
  unsigned long long c[3][8];
  for (int mode = 0; mode < 3; mode++) {
    t[mode] = time_ms([mode, viters] {
      hipLaunchKernelGGL((roles<KIND, MK>), dim3(256), dim3(512), 100 * 1024, 0, ITERS, viters, 1.0001f, 0.5f, g_out, g_cyc, mode);
    });
    hipMemcpy(c[mode], g_cyc, 64, hipMemcpyDeviceToHost);
  }
  const int nm = MK == 0 ? 8 : 16;
  printf("roles %-14s mfma %s: both %.3f ms | mfma waves alone %.3f ms (%.1f ns per mfma) | valu waves alone %.3f ms (%.2f ns per valu)\n",
         name, MK == 0 ? "32x32x16" : "16x16x32", t[0], t[1], t[1] * 1e6 / (ITERS * nm), t[2], t[2] * 1e6 / (64.0 * viters));
}

template <int KIND, int NV>
static void inter_row(const char* name) {
  const float ms = time_ms([] {
    hipLaunchKernelGGL((interleaved<KIND, NV>), dim3(256), dim3(256), 100 * 1024, 0, ITERS, 1.0001f, 0.5f, g_out, g_cyc);
  });
  printf("interleaved 1 wave/SIMD: 4 x (mfma 32x32x16 bf16 + %2d x %-10s): %7.1f ns per group\n", NV, name, ms * 1e6 / ITERS);
}

int main() {
  hipMalloc(&g_out, 4);
  hipMalloc(&g_cyc, 64);
  hipFuncSetAttribute((const void*)valu_only<K_FMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
#define ATTR(k) hipFuncSetAttribute((const void*)valu_only<k>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)
  ATTR(K_MUL); ATTR(K_AND); ATTR(K_PERM); ATTR(K_EXP); ATTR(K_PKFMA); ATTR(K_PKMUL); ATTR(K_CVTPK); ATTR(K_MOV); ATTR(K_ADD3);
  ATTR(K_LSHLOR); ATTR(K_BFE);
  valu_row<K_FMA>("v_fma_f32");
  valu_row<K_MUL>("v_mul_f32");
  valu_row<K_AND>("v_and_b32");
  valu_row<K_PERM>("v_perm_b32");
  valu_row<K_EXP>("v_exp_f32");
  valu_row<K_PKFMA>("v_pk_fma_f32");
  valu_row<K_PKMUL>("v_pk_mul_f32");
  valu_row<K_CVTPK>("v_cvt_pk_bf16_f32");
  valu_row<K_MOV>("v_mov_b32");
  valu_row<K_ADD3>("v_add3_u32");
  valu_row<K_LSHLOR>("v_lshl_or_b32");
  valu_row<K_BFE>("v_bfe_u32");
  hipFuncSetAttribute((const void*)roles<K_FMA, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)roles<K_FMA, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)roles<K_PKFMA, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)roles<K_EXP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  roles_row<K_FMA, 0>("v_fma_f32", ITERS);
  roles_row<K_FMA, 0>("v_fma_f32 x2", 2 * ITERS);
  roles_row<K_FMA, 1>("v_fma_f32", ITERS);
  roles_row<K_PKFMA, 0>("v_pk_fma_f32", ITERS);
  roles_row<K_EXP, 0>("v_exp_f32", ITERS);
#define ATTRI(k, n) hipFuncSetAttribute((const void*)interleaved<k, n>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)
  ATTRI(K_FMA, 0); ATTRI(K_FMA, 4); ATTRI(K_FMA, 8); ATTRI(K_FMA, 12); ATTRI(K_FMA, 16); ATTRI(K_FMA, 24); ATTRI(K_PERM, 8); ATTRI(K_EXP, 8);
  inter_row<K_FMA, 0>("v_fma_f32");
  inter_row<K_FMA, 4>("v_fma_f32");
  inter_row<K_FMA, 8>("v_fma_f32");
  inter_row<K_FMA, 12>("v_fma_f32");
  inter_row<K_FMA, 16>("v_fma_f32");
  inter_row<K_FMA, 24>("v_fma_f32");
  inter_row<K_PERM, 8>("v_perm_b32");
  inter_row<K_EXP, 8>("v_exp_f32");
  return 0;
}
