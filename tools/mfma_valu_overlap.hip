// Does VALU work hide in the shadow of the matrix pipe?  One wave per SIMD issues, per loop iteration, 4 MFMAs on 4
// independent accumulators, each followed by NV independent VALU FMAs.  If the two pipes overlap the iteration time
// stays at the MFMA time until the issue slots run out; if they share a datapath the times add.
// Second experiment: 8-wave workgroups whose waves 0-3 issue only MFMAs and waves 4-7 (same SIMDs, tools/simd_map.hip)
// only VALU.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap && tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F32_32 = 0, F32_16 = 1, BF16_32 = 2, NONE = 3 };
enum { V_FMA = 0, V_PKFMA = 1, V_EXP = 2 };

template <int VK>
__device__ __forceinline__ void valu(float (&x)[8], f32x2 (&xp)[4], int k, float a, float b) {
  if constexpr (VK == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(a), "v"(b));
  if constexpr (VK == V_PKFMA) {
    f32x2 aa = {a, a}, bb = {b, b};
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[k & 3]) : "v"(aa), "v"(bb));
  }
  if constexpr (VK == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
}

template <int MK, int VK, int NV>
__device__ __forceinline__ void body(int iters, float a, float b, float* out, bool do_mfma, bool do_valu) {
  f32x16 acc[4];
  f32x4 acc4[4];
  for (int i = 0; i < 4; i++) {
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int r = 0; r < 4; r++) acc4[i][r] = 0.f;
  }
  float x[8];
  f32x2 xp[4];
  for (int k = 0; k < 8; k++) x[k] = a * (float)(k + threadIdx.x);
  for (int k = 0; k < 4; k++) xp[k] = f32x2{x[k], x[k + 4]};
  bf16x8 ba, bb;
  for (int k = 0; k < 8; k++) { ba[k] = (__bf16)a; bb[k] = (__bf16)b; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 4; m++) {
      if (do_mfma) {
        if constexpr (MK == F32_32) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
        if constexpr (MK == F32_16) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m], 0, 0, 0);
        if constexpr (MK == BF16_32) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[m], 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int k = 0; k < NV; k++) valu<VK>(x, xp, k, a, b);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; i++) {
    for (int r = 0; r < 16; r++) s += acc[i][r];
    for (int r = 0; r < 4; r++) s += acc4[i][r];
  }
  for (int k = 0; k < 8; k++) s += x[k];
  for (int k = 0; k < 4; k++) s += xp[k].x + xp[k].y;
  if (s == 12345.678f) out[0] = s;
}

template <int MK, int VK, int NV>
__global__ void __launch_bounds__(256) one_wave(int iters, float a, float b, float* out) {
  body<MK, VK, NV>(iters, a, b, out, MK != NONE, true);
}

// waves 0-3: MFMA only; waves 4-7: VALU only (mode 0 = both, 1 = only the MFMA waves work, 2 = only the VALU waves)
template <int MK, int VK, int NV>
__global__ void __launch_bounds__(512) two_waves(int iters, float a, float b, float* out, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  if (mf && mode == 2) return;
  if (!mf && mode == 1) return;
  body<MK, VK, NV>(iters, a, b, out, mf, !mf);
}

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

static float* g_out;
static const int ITERS = 20000;

template <int MK, int VK, int NV>
static void row(const char* mname, const char* vname) {
  const float ms = time_ms([] { hipLaunchKernelGGL((one_wave<MK, VK, NV>), dim3(256), dim3(256), 0, 0, ITERS, 1.0001f, 0.5f, g_out); });
  printf("one wave/SIMD  %-14s + %2d x %-12s per MFMA: %8.1f ns / 4-MFMA group\n", mname, NV, vname, ms * 1e6f / ITERS);
}

template <int MK, int VK, int NV>
static void pair(const char* mname, const char* vname) {
  float t[3];
  for (int mode = 0; mode < 3; mode++)
    t[mode] = time_ms([mode] {
      hipLaunchKernelGGL((two_waves<MK, VK, NV>), dim3(256), dim3(512), 0, 0, ITERS, 1.0001f, 0.5f, g_out, mode);
    });
  printf("two waves/SIMD %-14s | %2d x %-12s: both %8.1f  mfma-only %8.1f  valu-only %8.1f ns / group\n", mname, NV, vname,
         t[0] * 1e6f / ITERS, t[1] * 1e6f / ITERS, t[2] * 1e6f / ITERS);
}

int main() {
  hipMalloc(&g_out, 4);
  row<NONE, V_FMA, 8>("none", "v_fma_f32");
  row<NONE, V_PKFMA, 8>("none", "v_pk_fma_f32");
  row<NONE, V_EXP, 8>("none", "v_exp_f32");
  row<F32_32, V_FMA, 0>("f32 32x32x2", "v_fma_f32");
  row<F32_32, V_FMA, 4>("f32 32x32x2", "v_fma_f32");
  row<F32_32, V_FMA, 8>("f32 32x32x2", "v_fma_f32");
  row<F32_32, V_FMA, 12>("f32 32x32x2", "v_fma_f32");
  row<F32_32, V_FMA, 16>("f32 32x32x2", "v_fma_f32");
  row<F32_32, V_PKFMA, 8>("f32 32x32x2", "v_pk_fma_f32");
  row<F32_32, V_EXP, 8>("f32 32x32x2", "v_exp_f32");
  row<F32_16, V_FMA, 0>("f32 16x16x4", "v_fma_f32");
  row<F32_16, V_FMA, 4>("f32 16x16x4", "v_fma_f32");
  row<F32_16, V_FMA, 8>("f32 16x16x4", "v_fma_f32");
  row<F32_16, V_PKFMA, 4>("f32 16x16x4", "v_pk_fma_f32");
  row<BF16_32, V_FMA, 0>("bf16 32x32x16", "v_fma_f32");
  row<BF16_32, V_FMA, 4>("bf16 32x32x16", "v_fma_f32");
  row<BF16_32, V_FMA, 8>("bf16 32x32x16", "v_fma_f32");
  row<BF16_32, V_PKFMA, 4>("bf16 32x32x16", "v_pk_fma_f32");
  pair<F32_32, V_FMA, 8>("f32 32x32x2", "v_fma_f32");
  pair<F32_32, V_FMA, 16>("f32 32x32x2", "v_fma_f32");
  pair<F32_32, V_PKFMA, 8>("f32 32x32x2", "v_pk_fma_f32");
  pair<F32_16, V_FMA, 8>("f32 16x16x4", "v_fma_f32");
  pair<BF16_32, V_FMA, 8>("bf16 32x32x16", "v_fma_f32");
  return 0;
}
