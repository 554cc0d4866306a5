// Which SIMD does wave i of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (15 << 11));  // HW_ID bits [15:0]
    out[blockIdx.x * 16 + wave] = (int)hw;
  }
}
int main() {
  int* d; hipMalloc(&d, 64 * 16 * 4); hipMemset(d, 0, 64 * 16 * 4);
  for (int threads : {256, 512, 768}) {
    hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; b++) {
      printf("threads %d block %d: ", threads, b);
      for (int w = 0; w < threads / 64; w++) printf("w%d->simd%d(slot%d,cu%d) ", w, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15, (h[b * 16 + w] >> 8) & 15);
      printf("\n");
    }
  }
  return 0;
}
