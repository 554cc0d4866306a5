// Prototype, second step of the plan in DESIGN.md ("fp32 MFMAs do not overlap with VALU work"): the DATA-GRADIENT-ONLY
// backward of the 36-64-64-64-1 SDF MLP with split-bf16 operands on v_mfma_f32_32x32x16_bf16 -- forward recompute
// (keeping gelu' of every hidden layer in registers) followed by the dH chain dH_l^T = W_{l+1}^T dZ_{l+1}^T, which in the
// transposed formulation is the same chained-register product as the forward with the weight image transposed: no
// cross-lane movement, no accumulators.  This is the first half of the full backward (the dW products and their
// sample<->feature transposes are the second) and what the sphere tracer / normal renderer call.
// 8 waves per workgroup share one 143 KB LDS image (three forward + three transposed layers of bf16 pieces).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/prototypes/mlp_dx_split_bf16.hip -o tools/mlp_dx_split_bf16
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
static constexpr int OFF_W0 = 0, OFF_W1 = REC0, OFF_W2 = REC0 + RECH;                      // forward images
static constexpr int OFF_T2 = REC0 + 2 * RECH, OFF_T1 = REC0 + 3 * RECH, OFF_T0 = REC0 + 4 * RECH;  // transposed images
static constexpr int OFF_F32 = REC0 + 5 * RECH;                                                     // then fp32 tail
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
__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
// gelu and its derivative Phi(z) + z phi(z) from one erf and one exp
__device__ __forceinline__ void gelu_both(float z, float& hval, float& gprime) {
  const float cdf = fmaf(0.5f, erf_fast(z * 0.70710678118654752440f), 0.5f);
  const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
  hval = z * cdf;
  gprime = fmaf(z, pdf, cdf);
}

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

__device__ __forceinline__ void zero_init(f32x16 (&acc)[2]) {
#pragma unroll
  for (int to = 0; to < 2; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[to][r] = 0.f;
}
// in-place: acc <- gelu(acc), g <- gelu'(acc)
__device__ __forceinline__ void act_both(f32x16 (&acc)[2], f32x16 (&g)[2]) {
#pragma unroll
  for (int to = 0; to < 2; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float hv, gp;
      gelu_both(acc[to][r], hv, gp);
      acc[to][r] = hv;
      g[to][r] = gp;
    }
}
template <int TERMS>
__device__ __forceinline__ void chain(const f32x16 (&in)[2], f32x16 (&out)[2], const u32x4* __restrict__ w, int lane) {
#pragma unroll
  for (int s = 0; s < SH; s++) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = in[s >> 1][8 * (s & 1) + j];
    mac<TERMS, SH>(out, x, w + s * 3 * 64, lane);
  }
}

#ifndef NWAVES_PER_WG
#define NWAVES_PER_WG 8
#endif
constexpr int NWAVES = NWAVES_PER_WG;  // 8: two waves per SIMD, 256 registers each; 4: one wave with 512
template <int TERMS>
__global__ void __launch_bounds__(NWAVES * 64, 1)
    dxk(int64_t N, const float* __restrict__ X, const float* __restrict__ dY, const u32x4* __restrict__ img,
        float* __restrict__ dX) {
  extern __shared__ __align__(16) u32x4 lds[];
  constexpr int NREC = (int)((IMG_BYTES + 15) / 16);
  for (int i = threadIdx.x; i < NREC; i += NWAVES * 64) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, sl = lane & 31;
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * NWAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * NWAVES) {
    asm volatile("" ::: "memory");
    const int64_t n = tile * 32 + sl, nc = n < N ? n : N - 1;
    // ---------------- forward recompute, gelu' kept
    f32x16 a[2], b[2], g1[2], g2[2];
    bias_init(a, tail, h);
    {
      float xs[S0][8];
#pragma unroll
      for (int s = 0; s < S0; s++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int k = 16 * s + 8 * h + j;
          xs[s][j] = k < K0 ? X[(int64_t)k * N + nc] : 0.f;
        }
#pragma unroll
      for (int s = 0; s < S0; s++) mac<TERMS, S0>(a, xs[s], lds + OFF_W0 + s * 3 * 64, lane);
    }
    act_both(a, g1);
    bias_init(b, tail + HID, h);
    chain<TERMS>(a, b, lds + OFF_W1, lane);
    act_both(b, g2);
    bias_init(a, tail + 2 * HID, h);
    chain<TERMS>(b, a, lds + OFF_W2, lane);
    // ---------------- backward chain; dZ3^T = w4 dy gelu'(z3) in place
    const float dy = dY[nc];
    const float* wf = tail + 3 * HID;
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float hv, gp;
        gelu_both(a[to][r], hv, gp);
        a[to][r] = wf[32 * to + row_of(r, h)] * dy * gp;
      }
    zero_init(b);
    chain<TERMS>(a, b, lds + OFF_T2, lane);                                                    // dH2^T = W3^T dZ3^T
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) b[to][r] *= g2[to][r];
    zero_init(a);
    chain<TERMS>(b, a, lds + OFF_T1, lane);                                                    // dH1^T
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) a[to][r] *= g1[to][r];
    zero_init(b);
    chain<TERMS>(a, b, lds + OFF_T0, lane);                                                    // dX^T (rows >= K0 are zero)
    if (n < N) {
#pragma unroll
      for (int to = 0; to < 2; to++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int k = 32 * to + row_of(r, h);
          if (k < K0) dX[(int64_t)k * N + n] = b[to][r];
        }
    }
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
          // transposed images: row (32 to + m) is an INPUT neuron of the layer, k runs over its OUTPUT neurons
          put(OFF_T2, SH, to, s, lane, j, W[2][(size_t)feat * HID + (32 * to + m)]);
          put(OFF_T1, SH, to, s, lane, j, W[1][(size_t)feat * HID + (32 * to + m)]);
          put(OFF_T0, SH, to, s, lane, j, (32 * to + m) < K0 ? W[0][(size_t)feat * K0 + (32 * to + m)] : 0.f);
        }
      }
    }
  float* tail = reinterpret_cast<float*>(img.data() + (size_t)OFF_F32 * 16);
  for (int l = 0; l < 3; l++) memcpy(tail + l * HID, B[l].data(), HID * 4);
  memcpy(tail + 3 * HID, W[3].data(), HID * 4);
  tail[4 * HID] = B[3][0];

  std::vector<float> dYh(N);
  for (auto& v : dYh) v = nd(rng);
  float *dXin, *dYd, *dXout;
  u32x4* dI;
  hipMalloc(&dXin, X.size() * 4);
  hipMalloc(&dYd, N * 4);
  hipMalloc(&dXout, X.size() * 4);
  hipMalloc(&dI, img.size());
  hipMemcpy(dXin, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dYd, dYh.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(dI, img.data(), img.size(), hipMemcpyHostToDevice);

  // double-precision reference of dX on a sample of points
  const int NCHK = 2048;
  std::vector<double> ref((size_t)NCHK * K0);
  auto pick = [&](int i) { return (int64_t)i * (N / NCHK) + (i % 31); };
  for (int i = 0; i < NCHK; i++) {
    const int64_t n = pick(i);
    std::vector<std::vector<double>> z(3), hh(3);
    std::vector<double> a(K0);
    for (int k = 0; k < K0; k++) a[k] = X[(size_t)k * N + n];
    for (int l = 0; l < 3; l++) {
      z[l].assign(HID, 0.0);
      hh[l].assign(HID, 0.0);
      for (int o = 0; o < HID; o++) {
        double acc = B[l][o];
        for (int k = 0; k < dims[l]; k++) acc += (double)W[l][(size_t)o * dims[l] + k] * a[k];
        z[l][o] = acc;
        hh[l][o] = 0.5 * acc * (1.0 + std::erf(acc * 0.70710678118654752440));
      }
      a = hh[l];
    }
    auto gp = [](double v) { return 0.5 * (1.0 + std::erf(v * 0.70710678118654752440)) + v * 0.3989422804014327 * std::exp(-0.5 * v * v); };
    std::vector<double> dz(HID), dh;
    for (int o = 0; o < HID; o++) dz[o] = (double)W[3][o] * dYh[n] * gp(z[2][o]);
    for (int l = 2; l >= 1; l--) {
      dh.assign(HID, 0.0);
      for (int o = 0; o < HID; o++)
        for (int k = 0; k < HID; k++) dh[k] += (double)W[l][(size_t)o * HID + k] * dz[o];
      for (int k = 0; k < HID; k++) dz[k] = dh[k] * gp(z[l - 1][k]);
    }
    for (int k = 0; k < K0; k++) {
      double acc = 0;
      for (int o = 0; o < HID; o++) acc += (double)W[0][(size_t)o * K0 + k] * dz[o];
      ref[(size_t)i * K0 + k] = acc;
    }
  }

  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> out(X.size());
  hipFuncSetAttribute((const void*)dxk<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES);
  auto launch = [&] { hipLaunchKernelGGL(dxk<6>, dim3(256), dim3(NWAVES * 64), IMG_BYTES, 0, N, dXin, dYd, dI, dXout); };
  hipMemset(dXout, 0, X.size() * 4);
  launch();
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::fmin(best, ms / 10);
  }
  hipMemcpy(out.data(), dXout, X.size() * 4, hipMemcpyDeviceToHost);
  double maxabs = 0, maxref = 0, sumsq = 0;
  int worst_k = -1;
  for (int i = 0; i < NCHK; i++)
    for (int k = 0; k < K0; k++) {
      const double d = std::fabs((double)out[(size_t)k * N + pick(i)] - ref[(size_t)i * K0 + k]);
      if (d > maxabs) { maxabs = d; worst_k = k; }
      maxref = std::fmax(maxref, std::fabs(ref[(size_t)i * K0 + k]));
      sumsq += d * d;
    }
  printf("dX-only backward, bf16 x6: %.4f ms at N = %lld (LDS image %zu B)   max |err| %.3e (input %d)  rms %.3e  max |dX| %.3f\n",
         best, (long long)N, IMG_BYTES, maxabs, worst_k, std::sqrt(sumsq / ((double)NCHK * K0)), maxref);
  return 0;
}
