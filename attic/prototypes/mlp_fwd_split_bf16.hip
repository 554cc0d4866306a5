// Prototype: the 36-64-64-64-1 SDF MLP forward with fp32 operands split into bf16 pieces and multiplied on the bf16
// matrix pipe (v_mfma_f32_32x32x16_bf16), fp32 accumulation.  Motivation (tools/mfma_valu_overlap.hip): fp32 MFMAs
// and VALU work do NOT overlap on gfx950, bf16 MFMAs do, and per k they are 16x faster -- so six bf16 products
//   a*b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),   a = a1 + a2 + a3 (8 mantissa bits each, truncation)
// cost 6/16 of the fp32 MFMA time, keep ~2^-22 relative accuracy per product, and leave the GELU in the shadow of the
// matrix pipe.  TERMS = 3 keeps only the first three products (~2^-16).
// Same chained-register design as csrc/mlp.hip: everything is computed transposed (Z^T = W H^T), the D tile of layer l
// is the B operand of layer l+1, only the weight images are permuted.
//   hipcc -O3 --offload-arch=gfx950 tools/prototypes/mlp_fwd_split_bf16.hip -o tools/mlp_fwd_split_bf16 && tools/mlp_fwd_split_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int K0 = 36, HID = 64, S0 = 3 /* k-steps of layer 0 (48 >= 36) */, SH = 4 /* k-steps of a chain layer */;
__host__ __device__ inline int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------ LDS image (units: 16-byte lane records)
// layer 0 : [to 2][s 3][piece 3][lane 64]      chain: [to 2][s 4][piece 3][lane 64]
static constexpr int REC0 = 2 * S0 * 3 * 64, RECH = 2 * SH * 3 * 64;
static constexpr int OFF_W0 = 0, OFF_W1 = REC0, OFF_W2 = REC0 + RECH, OFF_F32 = REC0 + 2 * RECH;  // then fp32 tail
static constexpr int TAIL_FLOATS = 3 * HID + HID + 1;  // biases of the three hidden layers, final weights, final bias
static constexpr size_t IMG_BYTES = (size_t)OFF_F32 * 16 + TAIL_FLOATS * 4;

__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float hi = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float lo = fmaf(q, a, a);
  return t > 0.927734375f ? hi : lo;
}
#ifdef GELU_EXP2_POLY
// gelu(x) = max(x, 0) - t Phi(-t), t = min(|x|, 5.75), Phi(-t) = exp2(P8(t)): 12 instructions; fit and error report in
// tools/gelu_fit.py (max error / |x| 8.6e-8 against float64, the fp32 erf formula itself has 1.06e-7)
__device__ __forceinline__ float gelu(float x) {
  const float t = fminf(fabsf(x), 5.75f);
  float p = -2.772052994e-06f;
  p = fmaf(p, t, 3.862077210e-05f);
  p = fmaf(p, t, -1.825476502e-04f);
  p = fmaf(p, t, -1.458701736e-04f);
  p = fmaf(p, t, 7.075471804e-03f);
  p = fmaf(p, t, -5.250502750e-02f);
  p = fmaf(p, t, -4.592049122e-01f);
  p = fmaf(p, t, -1.151105762e+00f);
  p = fmaf(p, t, -1.000000000e+00f);
  return fmaf(-t, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
#else
__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
#endif

// eight fp32 -> three bf16x8 pieces by truncation (each piece = the top 16 bits of the running remainder)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  uint32_t a[8], b[8], c[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    a[j] = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(a[j] & 0xFFFF0000u);
    b[j] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b[j] & 0xFFFF0000u);
    c[j] = __float_as_uint(r2);
  }
  u32x4 q1, q2, q3;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    q1[i] = __builtin_amdgcn_perm(a[2 * i + 1], a[2 * i], 0x07060302u);
    q2[i] = __builtin_amdgcn_perm(b[2 * i + 1], b[2 * i], 0x07060302u);
    q3[i] = __builtin_amdgcn_perm(c[2 * i + 1], c[2 * i], 0x07060302u);
  }
  p1 = __builtin_bit_cast(bf16x8, q1);
  p2 = __builtin_bit_cast(bf16x8, q2);
  p3 = __builtin_bit_cast(bf16x8, q3);
}

template <int TERMS, int NS>
__device__ __forceinline__ void mac(f32x16 (&out)[2], const float (&x)[8], const u32x4* __restrict__ w_s, int lane) {
  // w_s -> record [to = 0][s][piece 0][lane 0]; stride between `to` images = NS*3*64 records
  bf16x8 b1, b2, b3;
  split8(x, b1, b2, b3);
#pragma unroll
  for (int to = 0; to < 2; to++) {
    const u32x4* wt = w_s + (size_t)to * NS * 3 * 64 + lane;
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, wt[0]);
    const bf16x8 a2 = __builtin_bit_cast(bf16x8, wt[64]);
    if constexpr (TERMS == 6) {
      const bf16x8 a3 = __builtin_bit_cast(bf16x8, wt[128]);
      out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, out[to], 0, 0, 0);
      out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, out[to], 0, 0, 0);
      out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, out[to], 0, 0, 0);
    }
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, out[to], 0, 0, 0);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, out[to], 0, 0, 0);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, out[to], 0, 0, 0);
  }
}

__device__ __forceinline__ void bias_init(f32x16 (&acc)[2], const float* __restrict__ b, int h) {
#pragma unroll
  for (int to = 0; to < 2; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[to][r] = b[32 * to + row_of(r, h)];
}
__device__ __forceinline__ void gelu_all(f32x16 (&acc)[2]) {
#pragma unroll
  for (int to = 0; to < 2; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[to][r] = gelu(acc[to][r]);
}

template <int TERMS>
__global__ void __launch_bounds__(256, 2) fwd(int64_t N, const float* __restrict__ X, const u32x4* __restrict__ img, float* __restrict__ Y) {
  extern __shared__ __align__(16) u32x4 lds[];
  constexpr int NREC = (int)((IMG_BYTES + 15) / 16);
  for (int i = threadIdx.x; i < NREC; i += 256) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, sl = lane & 31;
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t n = tile * 32 + sl, nc = n < N ? n : N - 1;
    f32x16 h1[2], h2[2];
    bias_init(h1, tail, h);
    float xs[S0][8];
#pragma unroll
    for (int s = 0; s < S0; s++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = 16 * s + 8 * h + j;
        xs[s][j] = k < K0 ? X[(int64_t)k * N + nc] : 0.f;
      }
#pragma unroll
    for (int s = 0; s < S0; s++) mac<TERMS, S0>(h1, xs[s], lds + OFF_W0 + s * 3 * 64, lane);
    gelu_all(h1);
    bias_init(h2, tail + HID, h);
#pragma unroll
    for (int s = 0; s < SH; s++) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = h1[s >> 1][8 * (s & 1) + j];
      mac<TERMS, SH>(h2, x, lds + OFF_W1 + s * 3 * 64, lane);
    }
    gelu_all(h2);
    bias_init(h1, tail + 2 * HID, h);
#pragma unroll
    for (int s = 0; s < SH; s++) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = h2[s >> 1][8 * (s & 1) + j];
      mac<TERMS, SH>(h1, x, lds + OFF_W2 + s * 3 * 64, lane);
    }
    gelu_all(h1);
    const float* wf = tail + 3 * HID;
    float acc = 0.f;
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc = fmaf(wf[32 * to + row_of(r, h)], h1[to][r], acc);
    acc += __shfl_xor(acc, 32, 64);
    acc += wf[HID];
    if (h == 0 && n < N) Y[n] = acc;
  }
}

// ------------------------------------------------------------------ host
static void split3(float x, uint16_t (&p)[3]) {
  float r = x;
  for (int i = 0; i < 3; i++) {
    uint32_t u;
    memcpy(&u, &r, 4);
    p[i] = (uint16_t)(u >> 16);
    uint32_t t = u & 0xFFFF0000u;
    float tf;
    memcpy(&tf, &t, 4);
    r -= tf;
  }
}

int main() {
  const int64_t N = 1 << 21;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  const int dims[5] = {K0, HID, HID, HID, 1};
  std::vector<std::vector<float>> W(4), B(4);
  for (int l = 0; l < 4; l++) {
    W[l].resize((size_t)dims[l + 1] * dims[l]);
    B[l].resize(dims[l + 1]);
    const float sc = std::sqrt(2.0f / dims[l]);
    for (auto& w : W[l]) w = nd(rng) * sc;
    for (auto& b : B[l]) b = nd(rng) * 0.1f;
  }
  std::vector<float> X((size_t)K0 * N);
  for (auto& x : X) x = nd(rng);

  std::vector<uint8_t> img(((IMG_BYTES + 15) / 16) * 16, 0);
  uint16_t* rec = reinterpret_cast<uint16_t*>(img.data());
  auto put = [&](int off_rec, int NS, int to, int s, int lane, int j, float w) {
    uint16_t p[3];
    split3(w, p);
    for (int piece = 0; piece < 3; piece++)
      rec[((size_t)(off_rec + ((to * NS + s) * 3 + piece) * 64 + lane)) * 8 + j] = p[piece];
  };
  for (int to = 0; to < 2; to++)
    for (int lane = 0; lane < 64; lane++) {
      const int m = lane & 31, hh = lane >> 5;
      for (int j = 0; j < 8; j++) {
        for (int s = 0; s < S0; s++) {
          const int k = 16 * s + 8 * hh + j;
          put(OFF_W0, S0, to, s, lane, j, k < K0 ? W[0][(size_t)(32 * to + m) * K0 + k] : 0.f);
        }
        for (int s = 0; s < SH; s++) {
          const int feat = 32 * (s >> 1) + row_of(8 * (s & 1) + j, hh);
          put(OFF_W1, SH, to, s, lane, j, W[1][(size_t)(32 * to + m) * HID + feat]);
          put(OFF_W2, SH, to, s, lane, j, W[2][(size_t)(32 * to + m) * HID + feat]);
        }
      }
    }
  float* tail = reinterpret_cast<float*>(img.data() + (size_t)OFF_F32 * 16);
  for (int l = 0; l < 3; l++) memcpy(tail + l * HID, B[l].data(), HID * 4);
  memcpy(tail + 3 * HID, W[3].data(), HID * 4);
  tail[4 * HID] = B[3][0];

  float *dX, *dY;
  u32x4* dI;
  hipMalloc(&dX, X.size() * 4);
  hipMalloc(&dY, N * 4);
  hipMalloc(&dI, img.size());
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dI, img.data(), img.size(), hipMemcpyHostToDevice);

  // double-precision reference on a sample of points
  const int NCHK = 4096;
  std::vector<double> ref(NCHK);
  for (int i = 0; i < NCHK; i++) {
    const int64_t n = (int64_t)i * (N / NCHK) + (i % 31);
    std::vector<double> a(K0), z;
    for (int k = 0; k < K0; k++) a[k] = X[(size_t)k * N + n];
    for (int l = 0; l < 4; l++) {
      z.assign(dims[l + 1], 0.0);
      for (int o = 0; o < dims[l + 1]; o++) {
        double acc = B[l][o];
        for (int k = 0; k < dims[l]; k++) acc += (double)W[l][(size_t)o * dims[l] + k] * a[k];
        z[o] = l < 3 ? 0.5 * acc * (1.0 + std::erf(acc * 0.70710678118654752440)) : acc;
      }
      a = z;
    }
    ref[i] = a[0];
  }

  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> Y(N);
  for (int terms : {6, 3}) {
    auto launch = [&] {
      if (terms == 6) hipLaunchKernelGGL(fwd<6>, dim3(1024), dim3(256), IMG_BYTES, 0, N, dX, dI, dY);
      else hipLaunchKernelGGL(fwd<3>, dim3(1024), dim3(256), IMG_BYTES, 0, N, dX, dI, dY);
    };
    if (IMG_BYTES > 64 * 1024) {
      hipFuncSetAttribute((const void*)fwd<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES);
      hipFuncSetAttribute((const void*)fwd<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES);
    }
    hipMemset(dY, 0, N * 4);
    launch();
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; i++) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = std::fmin(best, ms / 20);
    }
    hipMemcpy(Y.data(), dY, N * 4, hipMemcpyDeviceToHost);
    double maxabs = 0, maxref = 0, sumsq = 0;
    for (int i = 0; i < NCHK; i++) {
      const int64_t n = (int64_t)i * (N / NCHK) + (i % 31);
      const double d = std::fabs((double)Y[n] - ref[i]);
      maxabs = std::fmax(maxabs, d);
      maxref = std::fmax(maxref, std::fabs(ref[i]));
      sumsq += d * d;
    }
    printf("bf16 x%d: %.4f ms at N = %lld (%.1f TF fp32-equivalent)   max |err| %.3e  rms %.3e  (max |y| %.3f; fp32 eps*|y| = %.1e)\n",
           terms, best, (long long)N, 21120.0 * N / (best * 1e-3) / 1e12, maxabs, std::sqrt(sumsq / NCHK), maxref,
           maxref * 1.19e-7);
  }
  return 0;
}
