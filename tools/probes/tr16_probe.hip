#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void probe(const int* addr_bytes, float* out) {
  __shared__ __align__(16) _Float16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)(i & 2047);
  __syncthreads();
  const int a = addr_bytes[threadIdx.x];
  auto p = (__attribute__((address_space(3))) s4*)((__attribute__((address_space(3))) char*)lds + a);
  s4 vi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p); h4 v = __builtin_bit_cast(h4, vi);
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (float)v[j];
}
int main() {
  int h_addr[64]; float h_out[256];
  int* d_addr; float* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int test = 0; test < 2; test++) {
    for (int l = 0; l < 64; l++) {
      if (test == 0) h_addr[l] = 8 * (l & 15) + 128 * (l >> 4);        // canonical: consecutive 8-byte chunks
      else { int i = l & 15, kg = l >> 4; int row = 4 * kg + (i >> 2), q = i & 3; h_addr[l] = row * 128 + q * 16 + 2048 * 0; }  // rows of 64 halfs, chunks at 16-byte pitch
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("test %d\n", test);
    for (int l = 0; l < 64; l++) {
      if (l < 20 || l % 16 < 2) printf("lane %2d (addr elem %4d): %5.0f %5.0f %5.0f %5.0f\n", l, h_addr[l] / 2, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    }
  }
  return 0;
}
