// Fused backward (with forward recompute) of the WIDE colour network: LipshitzMLP 111 -> 128 -> 128 -> 64 -> 3 of the reference
// (permuto_sdf_py/models/models.py:54-129, 349-350; the largest GEMM of the training step: 77 952 FLOP/sample forward).
//
// The narrow nets keep their whole weight-gradient accumulator in one wave's registers (mlp_bwd.hip); 39 303 parameters do
// not fit (608 registers), and the 157-KB fp32 weight set does not fit LDS beside the activations either.  So here a
// WORKGROUP of 8 waves (two per SIMD) owns a tile of 32 samples and the waves split every layer by OUTPUT tile (16 rows):
//   * activations, their GELU derivatives (overwritten in place by dZ) and the upstream gradient live in LDS as
//     [feature][32 samples] (row stride 36 floats): the SAME buffer is the B operand of the chain (lane = sample, 4 features
//     per read) and, read the other way (lane = feature, one 128-bit read = 4 samples), both operands of dW = dZ H^T --
//     no transposes anywhere;
//   * weights are streamed from L2 as MFMA A operands, 128 bits per lane = 4 MFMAs (W for the forward, W^T -- written once
//     per call by the pack kernel, which also applies the Lipschitz normalisation -- for dH); every weight byte is used for
//     32 samples, ~10 KB of L2 traffic per sample;
//   * each wave keeps the dW rows of ITS output tile in registers for the whole kernel (<= 108 registers) and they leave as
//     one gradient image per workgroup, summed by a second launch;
//   * fp32 MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products, no operand splitting; the batch is ~49 k samples per step, the
//     kernel is latency / launch bound, not matrix bound.
// One barrier per layer and direction (8 per tile).
#include "psdf_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int WN = 8;        // waves per workgroup
constexpr int TS = 32;       // samples per workgroup tile (two 16-sample MFMA column blocks)
constexpr int RS = 36;       // LDS row stride in floats (144 B: 128-bit reads of 16 consecutive rows spread over the banks)

__device__ __forceinline__ float erf_w(float a) {   // < 1 ulp (same polynomial as mlp_device.h)
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
__device__ __forceinline__ void gelu_both_w(float z, float& h, float& gp) {
  const float cdf = fmaf(0.5f, erf_w(z * 0.70710678118654752440f), 0.5f);
  const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
  h = z * cdf;
  gp = fmaf(z, pdf, cdf);
}

struct WideArgs {
  const float* W[4];    // normalised weights, row major [out][in_pad]  (in_pad = 16 * tiles of the input)
  const float* WT[4];   // their transposes [in_pad][out_pad]
  const float* b[4];
  int dims[5];          // true widths
};

// One layer forward for the wave's own output tile `t`: Z[16 rows][32 samples] = W[rows][:] * In + bias, GELU (unless last),
// H and gelu' go to LDS.  In/H/G are LDS [feature][RS].
template <int TIN, bool ACT>
__device__ __forceinline__ void layer_fwd(const float* __restrict__ W, int in_pad, const float* __restrict__ bias, int out_true,
                                          int t, const float* __restrict__ In, float* __restrict__ H, float* __restrict__ G,
                                          int c, int g) {
  f32x4 acc[2];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = 16 * t + 4 * g + r;
    const float bv = row < out_true ? bias[row] : 0.f;
    acc[0][r] = bv;
    acc[1][r] = bv;
  }
  const float* wrow = W + (size_t)(16 * t + c) * in_pad + 4 * g;
#pragma unroll 2
  for (int kg = 0; kg < TIN; kg++) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(wrow + 16 * kg);      // k = 16 kg + 4 g + j
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float* in = In + (16 * kg + 4 * g + j) * RS + c;
      acc[0] = MFMA4(a4[j], in[0], acc[0]);
      acc[1] = MFMA4(a4[j], in[16], acc[1]);
    }
  }
#pragma unroll
  for (int sb = 0; sb < 2; sb++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * t + 4 * g + r;
      float h = acc[sb][r], gp = 1.f;
      if (ACT) gelu_both_w(acc[sb][r], h, gp);
      H[row * RS + 16 * sb + c] = h;
      if (G) G[row * RS + 16 * sb + c] = gp;
    }
}

// dW[own tile `to`][all input tiles] += dZ_l(own rows) * H_{l-1}^T over the 32 samples; db += row sums
template <int TIN>
__device__ __forceinline__ void layer_dw(const float* __restrict__ D, const float* __restrict__ Hin, int to, f32x4 (&dW)[TIN],
                                         float& db, int c, int g) {
#pragma unroll
  for (int sb = 0; sb < 2; sb++) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(D + (16 * to + c) * RS + 16 * sb + 4 * g);   // samples 16 sb + 4 g + j
    db += (a4[0] + a4[1]) + (a4[2] + a4[3]);
#pragma unroll
    for (int ti = 0; ti < TIN; ti++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(Hin + (16 * ti + c) * RS + 16 * sb + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; j++) dW[ti] = MFMA4(a4[j], b4[j], dW[ti]);
    }
  }
}

// dH_{l-1}[own input tile `ti`] = W_l^T dZ_l, then * gelu'(z_{l-1}) in place (G -> dZ), or to global dX for the first layer
template <int TOUT>
__device__ __forceinline__ void layer_dh(const float* __restrict__ WT, int out_pad, int ti, const float* __restrict__ D,
                                         f32x4 (&acc)[2], int c, int g) {
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wrow = WT + (size_t)(16 * ti + c) * out_pad + 4 * g;
#pragma unroll 2
  for (int kg = 0; kg < TOUT; kg++) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(wrow + 16 * kg);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float* d = D + (16 * kg + 4 * g + j) * RS + c;
      acc[0] = MFMA4(a4[j], d[0], acc[0]);
      acc[1] = MFMA4(a4[j], d[16], acc[1]);
    }
  }
}

// gradient image of one workgroup (floats): dW1 [T1*16][TI0*16], dW2 [T2*16][T1*16], dW3 [T3*16][T2*16], dW4 [T4*16][T3*16],
// then db1, db2, db3, db4 (padded widths)
template <int TI0, int T1, int T2, int T3, int T4>
struct GImg {
  static constexpr int TL = T3 > 0 ? T3 : T2;   // tiles of the last hidden layer (T3 == 0: a net with two hidden layers, round 6)
  static constexpr int W1 = 0, W2 = W1 + T1 * 16 * TI0 * 16, W3 = W2 + T2 * 16 * T1 * 16, W4 = W3 + T3 * 16 * T2 * 16,
                       B1 = W4 + T4 * 16 * TL * 16, B2 = B1 + T1 * 16, B3 = B2 + T2 * 16, B4 = B3 + T3 * 16,
                       TOTAL = B4 + T4 * 16;
};

// T4 = output tiles of the (linear) last layer: 1 for the colour network (3 outputs), 5 for the background density / feature
// net 52 -> 64 x 3 -> 65 (models.py:451-459), whose single-wave fp32 kernel ran out of the register file (213-227 spilled
// registers, 209 us per call at 23 k samples: the second most expensive kernel of the training step until round 3).
template <int TI0, int T1, int T2, int T3, int T4>
__global__ void __launch_bounds__(WN * 64, 1)
    mlp_wide_bwd_kernel(WideArgs a, int64_t N, const float* __restrict__ X, const float* __restrict__ dY,
                        float* __restrict__ dX, float* __restrict__ partial) {
  static_assert(TI0 <= WN && T1 <= WN && T2 <= WN && T3 <= WN && T4 <= WN, "one output tile per wave and layer");
  extern __shared__ __align__(16) float lds[];
  float* H0 = lds;                         // [TI0*16][RS]   inputs
  float* H1 = H0 + TI0 * 16 * RS;          // activations
  float* H2 = H1 + T1 * 16 * RS;
  float* H3 = H2 + T2 * 16 * RS;
  float* D1 = H3 + T3 * 16 * RS;           // gelu' of the layer, overwritten by dZ
  float* D2 = D1 + T1 * 16 * RS;
  float* D3 = D2 + T2 * 16 * RS;
  float* D4 = D3 + T3 * 16 * RS;           // [T4*16][RS]: upstream gradient of the (linear) last layer, rows >= out are zero
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int K0 = a.dims[0], OUT = a.dims[4];
  const int in_pad[4] = {TI0 * 16, T1 * 16, T2 * 16, T3 * 16};
  f32x4 dW1[TI0], dW2[T1], dW3[T2], dW4[T3];
#pragma unroll
  for (int i = 0; i < TI0; i++) dW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < T1; i++) dW2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < T2; i++) dW3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < T3; i++) dW4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f, db3 = 0.f, db4 = 0.f;
  const int64_t ntiles = (N + TS - 1) / TS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();   // the previous tile's last readers are done with the buffers
    // ---- stage X rows (zero beyond K0 / N) and dY
    for (int e = threadIdx.x; e < TI0 * 16 * TS; e += WN * 64) {
      const int row = e / TS, s = e % TS;
      const int64_t n = n0 + s;
      H0[row * RS + s] = (row < K0 && n < N) ? X[(int64_t)row * N + n] : 0.f;
    }
    for (int e = threadIdx.x; e < T4 * 16 * TS; e += WN * 64) {
      const int row = e / TS, s = e % TS;
      const int64_t n = n0 + s;
      D4[row * RS + s] = (row < OUT && n < N) ? dY[(int64_t)row * N + n] : 0.f;
    }
    __syncthreads();
    // ---- forward
    if (wave < T1) layer_fwd<TI0, true>(a.W[0], in_pad[0], a.b[0], a.dims[1], wave, H0, H1, D1, c, g);
    __syncthreads();
    if (wave < T2) layer_fwd<T1, true>(a.W[1], in_pad[1], a.b[1], a.dims[2], wave, H1, H2, D2, c, g);
    __syncthreads();
    if (wave < T3) layer_fwd<T2, true>(a.W[2], in_pad[2], a.b[2], a.dims[3], wave, H2, H3, D3, c, g);
    __syncthreads();
    // (the last layer's output is not needed: the upstream gradient is given)
    // ---- backward, layer 4 (linear): dW4, db4 by the owners of its output tiles; dH3 -> dZ3 by the owners of H3's tiles
    if (wave < T4) layer_dw<T3>(D4, H3, wave, dW4, db4, c, g);
    if (wave < T3) {
      f32x4 acc[2];
      layer_dh<T4>(a.WT[3], T4 * 16, wave, D4, acc, c, g);
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float* p = D3 + (16 * wave + 4 * g + r) * RS + 16 * sb + c;
          *p = acc[sb][r] * *p;
        }
    }
    __syncthreads();
    // ---- layer 3
    if (wave < T3) layer_dw<T2>(D3, H2, wave, dW3, db3, c, g);
    if (wave < T2) {
      f32x4 acc[2];
      layer_dh<T3>(a.WT[2], in_pad[3], wave, D3, acc, c, g);
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float* p = D2 + (16 * wave + 4 * g + r) * RS + 16 * sb + c;
          *p = acc[sb][r] * *p;
        }
    }
    __syncthreads();
    // ---- layer 2
    if (wave < T2) layer_dw<T1>(D2, H1, wave, dW2, db2, c, g);
    if (wave < T1) {
      f32x4 acc[2];
      layer_dh<T2>(a.WT[1], in_pad[2], wave, D2, acc, c, g);
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float* p = D1 + (16 * wave + 4 * g + r) * RS + 16 * sb + c;
          *p = acc[sb][r] * *p;
        }
    }
    __syncthreads();
    // ---- layer 1
    if (wave < T1) layer_dw<TI0>(D1, H0, wave, dW1, db1, c, g);
    if (dX && wave < TI0) {
      f32x4 acc[2];
      layer_dh<T1>(a.WT[0], in_pad[1], wave, D1, acc, c, g);
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * wave + 4 * g + r;
          const int64_t n = n0 + 16 * sb + c;
          if (row < K0 && n < N) dX[(int64_t)row * N + n] = acc[sb][r];
        }
    }
  }
  // ---- the wave's accumulators -> this workgroup's gradient image.  D layout of an MFMA result: lane (col = c, g), register
  // r = row 4 g + r; the dW tiles have rows = output feature (own tile), cols = input feature.
  using GI = GImg<TI0, T1, T2, T3, T4>;
  float* img = partial + (size_t)blockIdx.x * GI::TOTAL;
  auto put = [&](int base, int ncols_pad, int to, int ntiles_in, const f32x4* acc) {
    for (int ti = 0; ti < ntiles_in; ti++)
#pragma unroll
      for (int r = 0; r < 4; r++) img[base + (16 * to + 4 * g + r) * ncols_pad + 16 * ti + c] = acc[ti][r];
  };
  auto put_db = [&](int base, int to, float v) {   // lane (feature c, g) holds the sum over its 2 x 4 samples: add the 4 groups
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) img[base + 16 * to + c] = v;
  };
  if (wave < T1) { put(GI::W1, TI0 * 16, wave, TI0, dW1); put_db(GI::B1, wave, db1); }
  if (wave < T2) { put(GI::W2, T1 * 16, wave, T1, dW2); put_db(GI::B2, wave, db2); }
  if (wave < T3) { put(GI::W3, T2 * 16, wave, T2, dW3); put_db(GI::B3, wave, db3); }
  if (wave < T4) { put(GI::W4, T3 * 16, wave, T3, dW4); put_db(GI::B4, wave, db4); }
}

// sum of the workgroup images -> ACCUMULATED into the torch-layout gradients of the (normalised) weights
template <int TI0, int T1, int T2, int T3, int T4>
__global__ void mlp_wide_reduce_kernel(const float* __restrict__ partial, int nimg, WideArgs a, float* dW0, float* dW1,
                                       float* dW2, float* dW3, float* db0, float* db1, float* db2, float* db3) {
  using GI = GImg<TI0, T1, T2, T3, T4>;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= GI::TOTAL) return;
  float s = 0.f;
  for (int b = blockIdx.y; b < nimg; b += gridDim.y) s += partial[(size_t)b * GI::TOTAL + e];   // slices meet by atomics
  auto mat = [&](int off, int cols_pad, int rows_true, int cols_true, float* dst) {
    const int row = off / cols_pad, col = off % cols_pad;
    if (row < rows_true && col < cols_true) atomicAdd(&dst[row * cols_true + col], s);
  };
  if (e < GI::W2) mat(e - GI::W1, TI0 * 16, a.dims[1], a.dims[0], dW0);
  else if (e < GI::W3) mat(e - GI::W2, T1 * 16, a.dims[2], a.dims[1], dW1);
  else if (e < GI::W4) mat(e - GI::W3, T2 * 16, a.dims[3], a.dims[2], dW2);
  else if (e < GI::B1) mat(e - GI::W4, GI::TL * 16, a.dims[4], a.dims[3], dW3);
  else if (e < GI::B2) { if (e - GI::B1 < a.dims[1]) atomicAdd(&db0[e - GI::B1], s); }
  else if (e < GI::B3) { if (e - GI::B2 < a.dims[2]) atomicAdd(&db1[e - GI::B2], s); }
  else if (e < GI::B4) { if (e - GI::B3 < a.dims[3]) atomicAdd(&db2[e - GI::B3], s); }
  else { if (e - GI::B4 < a.dims[4]) atomicAdd(&db3[e - GI::B4], s); }
}

// zero-padded copies of the weights in the two orientations the kernel streams: Wp [out_pad][in_pad], WTp [in_pad][out_pad];
// all four layers in one launch (blockIdx.y = layer)
struct PackLayers {
  int out[4], in[4], out_pad[4], in_pad[4];
  const float* W[4];
  float* Wp[4];
  float* WTp[4];
};
__global__ void mlp_wide_pack_kernel(PackLayers p) {
  const int l = blockIdx.y;
  const int out = p.out[l], in = p.in[l], out_pad = p.out_pad[l], in_pad = p.in_pad[l];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= out_pad * in_pad) return;
  const int o = e / in_pad, i = e % in_pad;
  const float v = (o < out && i < in) ? p.W[l][o * in + i] : 0.f;
  p.Wp[l][e] = v;
  p.WTp[l][i * out_pad + o] = v;
}

// Lipschitz weight normalisation of one layer (models.py:98-104): Wn[r][:] = W[r][:] * min(1, softplus(c) / sum_j |W[r][j]|).
// One wave per row.  The reference evaluates it with six torch launches per layer and direction.
__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch's threshold

__global__ void __launch_bounds__(64)
    lipshitz_norm_fwd_kernel(int in, const float* __restrict__ W, const float* __restrict__ c, float* __restrict__ Wn) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float a = 0.f;
  for (int j = lane; j < in; j += 64) a += fabsf(W[r * in + j]);
  a = psdf::wave_sum(a);
  const float scale = fminf(softplus_t(c[0]) / a, 1.0f);
  for (int j = lane; j < in; j += 64) Wn[r * in + j] = W[r * in + j] * scale;
}

// G = dL/dWn -> dW (written) and dc[0] (ACCUMULATED).  Active rows (scale < 1): Wn = W sp / A with A = sum |W|:
//   dW_j = G_j sp / A - (sum_k G_k W_k) sp / A^2 sign(W_j);  dsp += (sum_k G_k W_k) / A;  dc = dsp sigmoid(c)
__global__ void __launch_bounds__(64)
    lipshitz_norm_bwd_kernel(int in, const float* __restrict__ W, const float* __restrict__ c, const float* __restrict__ G,
                             float* __restrict__ dW, float* __restrict__ dc) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float a = 0.f, gw = 0.f;
  for (int j = lane; j < in; j += 64) {
    const float w = W[r * in + j];
    a += fabsf(w);
    gw += G[r * in + j] * w;
  }
  a = psdf::wave_sum(a);
  gw = psdf::wave_sum(gw);
  const float sp = softplus_t(c[0]);
  const float ratio = sp / a;
  const bool active = ratio < 1.0f;     // torch.clamp(max=1): the gradient passes where the input is below the bound
  for (int j = lane; j < in; j += 64) {
    const float w = W[r * in + j], g = G[r * in + j];
    const float sgn = w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f);
    dW[r * in + j] = active ? g * ratio - gw * sp / (a * a) * sgn : g;
  }
  if (active && lane == 0) {
    const float x = c[0];
    const float sig = 1.0f / (1.0f + expf(-x));
    atomicAdd(dc, gw / a * (x > 20.f ? 1.0f : sig));
  }
}

// All layers of a LipshitzMLP in one launch (blockIdx.y = layer, blockIdx.x = row): what the training step needs every
// iteration -- four forward and four backward launches otherwise.
constexpr int LIP_MAX_LAYERS = 8;
struct LipLayers {
  int out[LIP_MAX_LAYERS], in[LIP_MAX_LAYERS];
  const float* W[LIP_MAX_LAYERS];
  const float* c[LIP_MAX_LAYERS];
  const float* G[LIP_MAX_LAYERS];   // backward: dL/dWn
  float* Wn[LIP_MAX_LAYERS];        // forward output / backward: dW
  float* dc[LIP_MAX_LAYERS];
};
__global__ void __launch_bounds__(64) lipshitz_norm_multi_kernel(LipLayers p, int backward) {
  const int l = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
  if (r >= p.out[l]) return;
  const int in = p.in[l];
  const float* __restrict__ W = p.W[l];
  if (!backward) {
    float a = 0.f;
    for (int j = lane; j < in; j += 64) a += fabsf(W[r * in + j]);
    a = psdf::wave_sum(a);
    const float scale = fminf(softplus_t(p.c[l][0]) / a, 1.0f);
    for (int j = lane; j < in; j += 64) p.Wn[l][r * in + j] = W[r * in + j] * scale;
    return;
  }
  const float* __restrict__ G = p.G[l];
  float a = 0.f, gw = 0.f;
  for (int j = lane; j < in; j += 64) {
    const float w = W[r * in + j];
    a += fabsf(w);
    gw += G[r * in + j] * w;
  }
  a = psdf::wave_sum(a);
  gw = psdf::wave_sum(gw);
  const float x = p.c[l][0];
  const float sp = softplus_t(x);
  const float ratio = sp / a;
  const bool active = ratio < 1.0f;
  for (int j = lane; j < in; j += 64) {
    const float w = W[r * in + j], g = G[r * in + j];
    const float sgn = w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f);
    p.Wn[l][r * in + j] = active ? g * ratio - gw * sp / (a * a) * sgn : g;
  }
  if (active && lane == 0) {
    const float sig = 1.0f / (1.0f + expf(-x));
    atomicAdd(p.dc[l], gw / a * (x > 20.f ? 1.0f : sig));
  }
}

// pack (both weight orientations, zero padded), main launch, summing launch
template <int TI0, int T1, int T2, int T3, int T4>
int wide_launch(const int* dims, int64_t N, const float* X, const float* const* weights, const float* const* biases,
                const float* dY, float* dX, float* const* dW, float* const* db, hipStream_t st) {
  using GI = GImg<TI0, T1, T2, T3, T4>;
  const int pads[5] = {TI0 * 16, T1 * 16, T2 * 16, T3 * 16, T4 * 16};
  size_t wfloats = 0;
  for (int l = 0; l < 4; l++) wfloats += 2 * (size_t)pads[l] * pads[l + 1];
  const int64_t ntiles = (N + TS - 1) / TS;
  int64_t blocks = ntiles < 256 ? ntiles : 256;
  char* scratch = (char*)psdf::stream_scratch((wfloats + (size_t)blocks * GI::TOTAL) * sizeof(float), st);  // NULL while capturing
  if (!scratch) return PSDF_ERR_UNSUPPORTED;
  WideArgs a;
  PackLayers pk;
  float* wp = reinterpret_cast<float*>(scratch);
  int nmax = 0;
  for (int l = 0; l < 4; l++) {
    const int n = pads[l] * pads[l + 1];
    float* Wp = wp;
    float* WTp = wp + n;
    wp += 2 * n;
    pk.out[l] = dims[l + 1], pk.in[l] = dims[l], pk.out_pad[l] = pads[l + 1], pk.in_pad[l] = pads[l];
    pk.W[l] = weights[l], pk.Wp[l] = Wp, pk.WTp[l] = WTp;
    nmax = n > nmax ? n : nmax;
    a.W[l] = Wp;
    a.WT[l] = WTp;
    a.b[l] = biases[l];
  }
  hipLaunchKernelGGL(mlp_wide_pack_kernel, dim3((nmax + 255) / 256, 4), dim3(256), 0, st, pk);
  for (int i = 0; i < 5; i++) a.dims[i] = dims[i];
  float* partial = wp;
  const size_t lds_bytes = (size_t)((TI0 + 2 * T1 + 2 * T2 + 2 * T3 + T4) * 16) * RS * sizeof(float);
  auto kern = mlp_wide_bwd_kernel<TI0, T1, T2, T3, T4>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WN * 64), lds_bytes, st, a, N, X, dY, dX, partial);
  // (the small nets' images: 32 slices of the workgroup images instead of 8 -- their summing launch waits on loads, not on bytes)
  hipLaunchKernelGGL((mlp_wide_reduce_kernel<TI0, T1, T2, T3, T4>), dim3((GI::TOTAL + 255) / 256, GI::TOTAL < 20000 ? 32 : 8), dim3(256), 0, st, partial,
                     (int)blocks, a, dW[0], dW[1], dW[2], dW[3], db[0], db[1], db[2], db[3]);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}


// ====================================================================================================================
// The same workgroup-cooperative backward on the fp16 MATRIX PIPE with two pieces per fp32 operand (round 6).
// The fp32 kernel above is bound by its matrix instructions: a 32-sample tile of the colour network is 15 MFLOP with the forward
// recompute, 24 us at a CU's fp32-MFMA rate, six tiles per workgroup at a training step's ~49 k samples = the kernel's 170 us.
// Here every fp32 operand is a = a0 + a1 (a0 = fp16(a) to nearest, a1 = fp16(a - a0): 11 + sign + 11 bits, csrc/
// mlp_bwd_split_f16.hip) and a product keeps a0 b0 + a0 b1 + a1 b0 on v_mfma_f32_16x16x32_f16: one instruction covers 32 k
// values instead of 4, ~5x less matrix time.  What changes with it:
//   * LDS holds the activations as ready-made operand RECORDS (16 bytes per lane and piece), in two orientations:
//       B records  [k-step][sample block][piece][lane (c = sample, g)]: slot j = feature kf(s, g, j)   -- the chain products
//       T records  [feature tile][piece][lane (c = feature, g)]:       slot j = sample 16 (j >> 2) + 4 g + (j & 3) -- dW's H side
//     written once by the wave that produces the tile (B: two 8-byte stores into the lane's own record; T: eight 2-byte stores);
//     dZ overwrites the B records of the activation it belongs to; the dZ side of a wave's own dW rows goes through a private
//     2-KB scratch (the same eight 2-byte stores, read back as two records);
//   * gelu' stays in the REGISTERS of the wave that computed it (the same wave applies it on the way back);
//   * weights arrive pre-split as A records (both orientations) from a pack launch, two 16-byte loads per k-step;
//   * fp16 has five exponent bits and the upstream gradient of a radiance spans many decades (NeuS weights): the chain of sample
//     n runs on dY[:, n] * 2^k(n) (k from the largest of its entries: magnitudes in [2^4, 2^5)), dX[:, n] and the bias sums take
//     the factor out again (exact); for the parameter gradients, sums over samples, the factor goes to the other operand
//     (H[:, n] * 2^(kmin - k(n)), kmin = the tile's smallest k) and the tile's accumulators are folded into the running sums with
//     2^-kmin -- no second pass over dY, no extra launch.
// Same gradient image, same summing launch as the fp32 kernel.  Accuracy: tests/test_gpu_mlp.py::test_wide_net_backward_*.
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wh2 __attribute__((ext_vector_type(2)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t wu32x2 __attribute__((ext_vector_type(2)));
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
__host__ __device__ inline int wkf(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }
// two fp32 -> packed high pieces, packed low pieces (element 0 in the low half)
__device__ __forceinline__ void wsplit2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const wh2 h = __builtin_convertvector(wf32x2{x0, x1}, wh2);      // v_cvt_pk_f16_f32: nearest even
  const wf32x2 r = wf32x2{x0, x1} - wf32x2{(float)h[0], (float)h[1]};   // exact
  const wh2 l = __builtin_convertvector(r, wh2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
struct WRec {   // the two pieces of one 8-slot operand
  wf16x8 p[2];
};
__device__ __forceinline__ WRec wload(const wu32x4* __restrict__ r) {   // r -> piece 0 of the lane's record; piece 1 is 64 records on
  WRec o;
  o.p[0] = __builtin_bit_cast(wf16x8, r[0]);
  o.p[1] = __builtin_bit_cast(wf16x8, r[64]);
  return o;
}
__device__ __forceinline__ f32x4 wmac3(const WRec& a, const WRec& b, f32x4 acc) {   // a0 b0 + a0 b1 + a1 b0
  acc = MFMA16H(a.p[1], b.p[0], acc);
  acc = MFMA16H(a.p[0], b.p[1], acc);
  acc = MFMA16H(a.p[0], b.p[0], acc);
  return acc;
}

// gelu and gelu' of four values from ONE exponential and ONE reciprocal each (the fit of csrc/mlp_bwd_split_f16.hip: E = exp(-z^2/2),
// t = 1 / (1 + 0.39 |z|), Phi(-|z|) = t P6(t) E; errors 1.8e-7 |z| and 1.9e-7 against float64), two packed pairs side by side
__device__ __forceinline__ wf32x2 wfma2(wf32x2 a, wf32x2 b, wf32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ wf32x2 wsp2(float v) { return wf32x2{v, v}; }
__device__ __forceinline__ void wgelu4(const f32x4& z, f32x4& h, f32x4& gp) {
  const wf32x2 za = {z[0], z[1]}, zb = {z[2], z[3]};
  const wf32x2 ea = (za * za) * wsp2(-0.72134752044448170368f), eb = (zb * zb) * wsp2(-0.72134752044448170368f);
  const wf32x2 Ea = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const wf32x2 Eb = {__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)};
  const wf32x2 da = {__builtin_fmaf(__builtin_fabsf(za.x), 0.39f, 1.0f), __builtin_fmaf(__builtin_fabsf(za.y), 0.39f, 1.0f)};
  const wf32x2 db = {__builtin_fmaf(__builtin_fabsf(zb.x), 0.39f, 1.0f), __builtin_fmaf(__builtin_fabsf(zb.y), 0.39f, 1.0f)};
  const wf32x2 ta = {__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
  const wf32x2 tb = {__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
  wf32x2 qa = wsp2(5.384693295e-02f), qb = wsp2(5.384693295e-02f);
#define WHORNER(C) qa = wfma2(qa, ta, wsp2(C)); qb = wfma2(qb, tb, wsp2(C));
  WHORNER(-2.582434118e-01f) WHORNER(3.751679361e-01f) WHORNER(-1.663514599e-02f) WHORNER(1.944366544e-01f) WHORNER(1.514270604e-01f)
#undef WHORNER
  const wf32x2 la = (qa * ta) * Ea, lb = (qb * tb) * Eb;
  const wf32x2 ma = wsp2(0.5f) - la, mb = wsp2(0.5f) - lb;
  const wf32x2 ca = wf32x2{__builtin_copysignf(ma.x, za.x), __builtin_copysignf(ma.y, za.y)} + wsp2(0.5f);
  const wf32x2 cb = wf32x2{__builtin_copysignf(mb.x, zb.x), __builtin_copysignf(mb.y, zb.y)} + wsp2(0.5f);
  const wf32x2 ha = za * ca, hb = zb * cb;
  const wf32x2 ga = wfma2(za, Ea * wsp2(0.3989422804014327f), ca), gb = wfma2(zb, Eb * wsp2(0.3989422804014327f), cb);
  h = f32x4{ha.x, ha.y, hb.x, hb.y};
  gp = f32x4{ga.x, ga.y, gb.x, gb.y};
}

struct WideArgsH {
  const wu32x4* A[4];    // forward weight records of layer l: [out tile][k-step][piece][lane]
  const wu32x4* AT[4];   // transposed: [in tile][k-step over the outputs][piece][lane]
  const float* b[4];
  int dims[5];
  uint32_t* overflow;    // host-mapped word: some |value| left the fp16 range (the launcher then falls back to the fp32 kernel)
};
constexpr int wns(int tiles) { return (tiles + 1) / 2; }   // k-steps (32 features) covering `tiles` 16-feature tiles

// weight records of all four layers, both orientations: blockIdx.y = layer * 2 + orientation, thread = (tile, k-step, lane)
struct PackH {
  int out[4], in[4], out_tiles[4], in_tiles[4];
  const float* W[4];
  wu32x4* A[4];
  wu32x4* AT[4];
};
__global__ void mlp_wide_f16_pack_kernel(PackH p) {
  const int l = blockIdx.y >> 1, tr = blockIdx.y & 1;
  if ((tr && !p.AT[l]) || !p.W[l]) return;                           // (forward only: no transposed records; an empty layer slot)
  const int rows_t = tr ? p.in_tiles[l] : p.out_tiles[l];            // tiles of the records' rows
  const int ks = wns(tr ? p.out_tiles[l] : p.in_tiles[l]);           // k-steps
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows_t * ks * 64) return;
  const int lane = e & 63, s = (e >> 6) % ks, t = (e >> 6) / ks;
  const int c = lane & 15, g = lane >> 4, row = 16 * t + c;
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = wkf(s, g, j);
    const int o = tr ? k : row, i = tr ? row : k;
    w[j] = (o < p.out[l] && i < p.in[l]) ? p.W[l][o * p.in[l] + i] : 0.f;
  }
  wu32x4 hi, lo;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t h, l2;
    wsplit2(w[2 * q], w[2 * q + 1], h, l2);
    hi[q] = h;
    lo[q] = l2;
  }
  wu32x4* dst = (tr ? p.AT[l] : p.A[l]) + ((size_t)(t * ks + s) * 2) * 64 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

#if defined(PSDF_WIDE_DEBUG)
__device__ unsigned long long g_wide_dbg[64];
#define WDBG if (blockIdx.x == 0 && lane == 0 && tile == (int64_t)blockIdx.x + gridDim.x && (wave == 0 || wave == 7)) g_wide_dbg[(wave == 7 ? 32 : 0) + wdbg_i++] = __builtin_readcyclecounter();
#else
#define WDBG
#endif
template <int TI0, int T1, int T2, int T3, int T4>
__global__ void __launch_bounds__(WN * 64, 1)
    mlp_wide_bwd_f16_kernel(WideArgsH a, int64_t N, const float* __restrict__ X, const float* __restrict__ dY,
                            float* __restrict__ dX, float* __restrict__ partial) {
  static_assert(TI0 <= WN && T1 <= WN && T2 <= WN && T3 <= WN && T4 <= WN, "one output tile per wave and layer");
  constexpr int NS0 = wns(TI0), NS1 = wns(T1), NS2 = wns(T2), NS3 = wns(T3), NS4 = wns(T4);
  extern __shared__ __align__(16) wu32x4 wl[];
  // B records: [k-step][sample block 2][piece 2][lane 64] = 256 records per k-step
  wu32x4* B0 = wl;                       // inputs
  wu32x4* B1 = B0 + NS0 * 256;           // h1, later dZ1
  wu32x4* B2 = B1 + NS1 * 256;
  wu32x4* B3 = B2 + NS2 * 256;
  wu32x4* B4 = B3 + NS3 * 256;           // the (scaled) upstream gradient
  // T records: [tile][piece 2][lane 64] = 128 records per tile
  wu32x4* X0T = B4 + NS4 * 256;
  wu32x4* H1T = X0T + TI0 * 128;
  wu32x4* H2T = H1T + T1 * 128;
  wu32x4* H3T = H2T + T2 * 128;
  wu32x4* SCR = H3T + T3 * 128;          // per wave: the dZ side of its own dW rows, [piece 2][lane 64]
  wu32x4* YST = SCR + 2 * WN * 128;      // [T4][8][64] floats: the staged upstream gradient of the next tile
  float* MSC = reinterpret_cast<float*>(YST + T4 * 128);        // [T4][32]: per output tile, the largest |dY| of a sample
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int K0 = a.dims[0], OUT = a.dims[4];
  // scratch of an output tile: the dZ side (transposed) of that tile's dW rows; TWO sets, alternating layer by layer (the owners
  // of a slice of a tile's dW rows read set A while the tile's chain wave already writes the next layer's dZ into set B)
  // Ownership of the parameter-gradient blocks (16 x 16, a row tile of dZ x a column tile of H): a layer with T output tiles
  // gives wave w the row tile w % T and every (8 / T)-th ... precisely: the column tiles [cb, cb + CN) with CN = ceil(cols * T / 8)
  // ... so that all eight waves carry accumulators for every layer (a layer with 4 output tiles would otherwise leave half of the
  // waves' registers unused and the other half short: the kernel lives at 256 registers per wave).
  constexpr bool L3 = T3 > 0;            // three hidden layers (the colour / density nets) or two (80 -> 64 -> 64 -> 3, the background
                                         // colour head, models.py:463-469: layer 3 does not exist, the linear layer reads h2)
  constexpr int TL = L3 ? T3 : T2;
  constexpr int S1 = WN / T1 > 0 ? WN / T1 : 1, S2 = WN / T2 > 0 ? WN / T2 : 1, S3 = (L3 && WN / (L3 ? T3 : 1) > 0) ? WN / (L3 ? T3 : 1) : 1,
                S4 = WN / T4 > 0 ? WN / T4 : 1;
  constexpr int C1 = (TI0 + S1 - 1) / S1, C2 = (T1 + S2 - 1) / S2, C3 = L3 ? (T2 + S3 - 1) / S3 : 1, C4 = (TL + S4 - 1) / S4;   // column tiles per wave
  f32x4 dW1[C1], dW2[C2], dW3[C3], dW4[C4];
#pragma unroll
  for (int i = 0; i < C1; i++) dW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < C2; i++) dW2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < C3; i++) dW3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < C4; i++) dW4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 db1 = {0.f, 0.f, 0.f, 0.f}, db2 = db1, db3 = db1, db4 = db1;   // D layout: row 4 g + r of the own tile, this lane's samples
  float vmax = 0.f;
  const int64_t ntiles = (N + TS - 1) / TS;

  // a tile of values in D layout (v[sb][r] = feature 16 t + 4 g + r, sample 16 sb + c) -> its slots of the B records of `Breg`
  auto put_b = [&](wu32x4* Breg, int t, const f32x4 (&v)[2]) {
#pragma unroll
    for (int sb = 0; sb < 2; sb++) {
      uint32_t h0, l0, h1, l1;
      wsplit2(v[sb][0], v[sb][1], h0, l0);
      wsplit2(v[sb][2], v[sb][3], h1, l1);
      wu32x2* rec = reinterpret_cast<wu32x2*>(Breg + ((t >> 1) * 2 + sb) * 128 + lane) + (t & 1);
      rec[0] = wu32x2{h0, h1};
      rec[128] = wu32x2{l0, l1};      // piece 1: 64 records = 128 half records on
    }
  };
  // four features (4 g + r of a tile) of the lane's sample 16 sb + c, transposed, into the T records at `Treg` (2 x 64 records)
  auto put_t1 = [&](wu32x4* Treg, int sb, const f32x4& v) {
    _Float16* base = reinterpret_cast<_Float16*>(Treg);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const _Float16 h = (_Float16)v[r];
      const _Float16 l = (_Float16)(v[r] - (float)h);
      const int idx = (((c >> 2) * 16 + 4 * g + r) * 8) + 4 * sb + (c & 3);     // record lane' * 8 halves + slot
      base[idx] = h;
      base[64 * 8 + idx] = l;
    }
  };
  auto put_t = [&](wu32x4* Treg, const f32x4 (&v)[2]) {
    put_t1(Treg, 0, v[0]);
    put_t1(Treg, 1, v[1]);
  };
  static_assert(wns(TI0) * 2 <= WN, "one (k-step, sample block) of the inputs per wave");
  // The inputs of a tile arrive by LDS-DMA (global_load_lds: memory -> LDS without passing through registers), requested during
  // the tile BEFORE: a tile's ~14 KB come straight from HBM and nothing else would hide that latency.  Wave w < 2 NS0 fetches the
  // 8 features kf(s, g, j) of sample 16 sb + c, (s, sb) = (w / 2, w % 2), instruction j landing at [j][lane] of the 2 KB that
  // hold the wave's own B records (k-step s, sample block sb) afterwards -- nobody else touches them between the first forward
  // layer of a tile and the staging of the next; wave w < T4 fetches rows 4 g + r of output tile w of the upstream gradient into
  // YST.  Addresses are clamped; what lies outside the batch / the net is zeroed where the values are read (`settle`).
  float px[8];
  f32x4 pdy[2];
  auto request = [&](int64_t t2) {
    const int64_t m0 = (t2 < ntiles ? t2 : ntiles - 1) * TS;
    // (lane arithmetic redone HERE from a value the optimiser cannot see through: hoisted out of the tile loop, the sixteen row
    //  offsets end up in scratch, and a scratch reload between two requests waits (vmcnt) for the request in front of it, i.e.
    //  for HBM -- eight times per tile, 10 000 cycles; the lesson of csrc/mlp_bwd_split.hip)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int c = lane_o & 15, g = lane_o >> 4;
    if (wave < NS0 * 2) {
      const int s = wave >> 1, sb = wave & 1;
      int64_t n = m0 + 16 * sb + c;
      n = n < N ? n : N - 1;
      float* dst = reinterpret_cast<float*>(B0 + wave * 128);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int row = wkf(s, g, j);
        row = row < K0 ? row : K0 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)row * N + n),
                                         (__attribute__((address_space(3))) void*)(dst + j * 64), 4, 0, 0);
      }
    }
    if (wave < T4) {
      float* dst = reinterpret_cast<float*>(YST + wave * 128);
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int row = 16 * wave + 4 * g + r;
          row = row < OUT ? row : OUT - 1;
          int64_t n = m0 + 16 * sb + c;
          n = n < N ? n : N - 1;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dY + (int64_t)row * N + n),
                                           (__attribute__((address_space(3))) void*)(dst + (sb * 4 + r) * 64), 4, 0, 0);
        }
    }
  };
  auto settle = [&](int64_t t2) {      // the staged values into registers; zero in place of values outside the batch / the net
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA completion is not tracked by the compiler
    const int64_t m0 = t2 * TS;
    const int s = (wave < NS0 * 2 ? wave : 0) >> 1, sbx = wave & 1;
    const float* srcx = reinterpret_cast<const float*>(B0 + (wave < NS0 * 2 ? wave : 0) * 128);
#pragma unroll
    for (int j = 0; j < 8; j++) px[j] = (wkf(s, g, j) < K0 && m0 + 16 * sbx + c < N) ? srcx[j * 64 + lane] : 0.f;
    const float* srcy = reinterpret_cast<const float*>(YST + (wave < T4 ? wave : 0) * 128);
#pragma unroll
    for (int sb = 0; sb < 2; sb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * wave + 4 * g + r;
        pdy[sb][r] = (wave < T4 && row < OUT && m0 + 16 * sb + c < N) ? srcy[(sb * 4 + r) * 64 + lane] : 0.f;
      }
  };
  request(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    int wdbg_i = 0;
    WDBG
    __syncthreads();   // the previous tile's last readers are done with the buffers
    WDBG
    settle(tile);
    // ---- the lane's two samples (16 sb + c): scale 2^k of the upstream gradient.  pdy = the upstream gradient of rows 4 g + r
    // of tile `wave` (D layout, requested during the previous tile's backward); the largest entry of a sample's column is the
    // maximum over the rows of a lane, over the four lanes (g) that hold the sample and -- more than one output tile -- over
    // the waves (through LDS)
    float up[2], dn[2];      // 2^k(n), 2^-k(n)
    int kk[2];
    {
      float m2[2];
#pragma unroll
      for (int sb = 0; sb < 2; sb++) {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) m = fmaxf(m, fabsf(pdy[sb][r]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        m2[sb] = m;
      }
      {
        if (wave < T4 && g == 0) {
          MSC[wave * 32 + c] = m2[0];
          MSC[wave * 32 + 16 + c] = m2[1];
        }
        __syncthreads();
#pragma unroll
        for (int w2 = 0; w2 < T4; w2++) {
          m2[0] = w2 == 0 ? MSC[c] : fmaxf(m2[0], MSC[w2 * 32 + c]);
          m2[1] = w2 == 0 ? MSC[16 + c] : fmaxf(m2[1], MSC[w2 * 32 + 16 + c]);
        }
      }
#pragma unroll
      for (int sb = 0; sb < 2; sb++) {
        const int ex = (int)((__float_as_uint(m2[sb]) >> 23) & 255u);
        // m * 2^k in [2^4, 2^5).  A sample without upstream gradient (zero, denormal; the padding beyond N) gets the LARGEST k: it
        // contributes nothing whatever its factor and must not set the tile's kmin (which would push the H side of every other
        // sample's parameter-gradient products towards fp16's subnormals); non-finite: 1
        int k = ex == 255 ? 0 : (ex == 0 ? 100 : (127 + 4) - ex);
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        kk[sb] = k;
        up[sb] = __uint_as_float((uint32_t)(127 + k) << 23);
        dn[sb] = __uint_as_float((uint32_t)(127 - k) << 23);
      }
    }
    int kmin = kk[0] < kk[1] ? kk[0] : kk[1];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int other = __shfl_xor(kmin, o, 64);
      kmin = other < kmin ? other : kmin;
    }
    float hs[2];             // 2^(kmin - k(n)) <= 1: the H side of the parameter-gradient products
#pragma unroll
    for (int sb = 0; sb < 2; sb++) hs[sb] = (kmin - kk[sb] < -120) ? 0.f : __uint_as_float((uint32_t)(127 + kmin - kk[sb]) << 23);
    const float fold = __uint_as_float((uint32_t)(127 - kmin) << 23);   // 2^-kmin
    // ---- stage the inputs (wave = (k-step, sample block)): B records, and the same values scaled and transposed as T records
    if (wave < NS0 * 2) {
      const int s = wave >> 1, sb = wave & 1;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x[j] = px[j];
        vmax = fmaxf(vmax, fabsf(x[j]));
      }
      wu32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t h, l;
        wsplit2(x[2 * j], x[2 * j + 1], h, l);
        hi[j] = h;
        lo[j] = l;
      }
      wu32x4* rec = B0 + (s * 2 + sb) * 128 + lane;
      rec[0] = hi;
      rec[64] = lo;
      const float f = hs[sb];
      put_t1(X0T + (2 * s) * 128, sb, f32x4{x[0] * f, x[1] * f, x[2] * f, x[3] * f});
      if (2 * s + 1 < TI0) put_t1(X0T + (2 * s + 1) * 128, sb, f32x4{x[4] * f, x[5] * f, x[6] * f, x[7] * f});
    }
    // ... and the upstream gradient of the last layer, scaled per sample: B records + own scratch (for dW4) + bias sums
    f32x4 g1[2], g2[2], g3[2];     // gelu' of the own tile of each layer
    if (wave < T4) {
      f32x4 v[2];
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float d = pdy[sb][r];
          db4[r] += d;
          v[sb][r] = d * up[sb];
        }
      put_b(B4, wave, v);
      put_t(SCR + wave * 128, v);                      // set A (layer 4)
    } else if (T4 & 1) {
      // (an odd tile count leaves the upper half of the last k-step's records unwritten: zero them once per tile)
      if (wave == WN - 1) {
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
          wu32x2* rec = reinterpret_cast<wu32x2*>(B4 + ((NS4 - 1) * 2 + sb) * 128 + lane) + 1;
          rec[0] = wu32x2{0u, 0u};
          rec[128] = wu32x2{0u, 0u};
        }
      }
    }
    // weight records of the wave's tile for a phase, requested a phase AHEAD (their L2 latency hides behind the activation work
    // and the barrier in between; with two waves per SIMD nothing else would hide it)
    static_assert(NS0 <= 4 && NS1 <= 4 && NS2 <= 4 && NS3 <= 4 && NS4 <= 4, "four k-steps of weight records in flight");
    // (the first two k-steps: 16 registers; the other two are requested when the phase starts and arrive under the first MFMAs --
    //  all four ahead cost 32 registers that the kernel, at 256 per wave, does not have)
    WRec wq[2];
    const wu32x4* wnext = nullptr;     // the records the current prefetch belongs to (k-steps 2, 3 follow from it)
    auto prefetch = [&](const wu32x4* Aw, int ns, int tiles) {
      // (the lane index behind an empty asm: left visible, the eight record addresses of a tile are hoisted out of the tile loop as
      //  64-bit register pairs, parked in scratch, and their reloads wait -- vmcnt counts in order -- for the LDS-DMA prefetch of
      //  the next tile that is in flight by then; recomputed here they are two instructions each)
      int lane_p = lane;
      asm volatile("" : "+v"(lane_p));
      wnext = reinterpret_cast<const wu32x4*>(reinterpret_cast<const char*>(Aw) + (uint32_t)((wave * ns * 128 + lane_p) * 16));
#pragma unroll
      for (int s = 0; s < 2; s++)
        if (s < ns && wave < tiles) wq[s] = wload(wnext + (size_t)s * 128);
    };
    // acc = (bias +) sum over the k-steps of  weight records x B records
    auto mma = [&](int ns, const wu32x4* Bin, f32x4 (&acc)[2]) {
      const wu32x4* wcur = wnext;
      WRec w23[2];
#pragma unroll
      for (int s = 2; s < 4; s++)
        if (s < ns) w23[s - 2] = wload(wcur + (size_t)s * 128);
#pragma unroll
      for (int s = 0; s < 4; s++)
        if (s < ns) {
          const WRec& wa = s < 2 ? wq[s] : w23[s - 2];
#pragma unroll
          for (int sb = 0; sb < 2; sb++) acc[sb] = wmac3(wa, wload(Bin + (s * 2 + sb) * 128 + lane), acc[sb]);
        }
    };
    auto bias_init = [&](const float* bias, int out_true, f32x4 (&acc)[2]) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * wave + 4 * g + r;
        const float bv = row < out_true ? bias[row] : 0.f;
        acc[0][r] = bv;
        acc[1][r] = bv;
      }
    };
    const f32x4 zero2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // pre-activations -> activation (B records, scaled T records) and gelu' (kept)
    auto act = [&](f32x4 (&acc)[2], wu32x4* Bout, wu32x4* Tout, f32x4 (&gp)[2], bool odd_pad, int tiles_out) {
      f32x4 h[2], hsc[2];
#pragma unroll
      for (int sb = 0; sb < 2; sb++) {
        wgelu4(acc[sb], h[sb], gp[sb]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          hsc[sb][r] = h[sb][r] * hs[sb];
          vmax = fmaxf(vmax, fabsf(h[sb][r]));
        }
      }
      put_b(Bout, wave, h);
      put_t(Tout + wave * 128, hsc);
      if (odd_pad && wave == tiles_out - 1) put_b(Bout, wave + 1, zero2);   // an odd number of tiles: the partner half is zero
    };
    // dW rows of the own tile: dZ^T (own scratch) x H^T (T records), folded into the running sums
    // (the scratch was written before the last barrier, by the wave that owns the tile in the chain)
    auto dw = [&](const wu32x4* Sset, const wu32x4* Tin, int tout, int tin, int cn, f32x4* run) {
      const int sl = wave / tout, to = wave - sl * tout;       // slice of the columns, row tile
      const WRec za = wload(Sset + to * 128 + lane);
      for (int i = 0; i < cn; i++) {
        const int ti = sl * cn + i;
        if (ti < tin) {
          const f32x4 t = wmac3(za, wload(Tin + ti * 128 + lane), f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int r = 0; r < 4; r++) run[i][r] = fmaf(t[r], fold, run[i][r]);
        }
      }
    };
    // dH (of the own tile of the layer below) times gelu' -> dZ: B records (in place of the activation's), scratch, bias sums
    auto to_dz = [&](f32x4 (&acc)[2], const f32x4 (&gp)[2], f32x4& db, wu32x4* Bout, wu32x4* Sset, bool odd_pad, int tiles) {
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          acc[sb][r] *= gp[sb][r];
          db[r] = fmaf(acc[sb][r], dn[sb], db[r]);
        }
      put_b(Bout, wave, acc);
      put_t(Sset + wave * 128, acc);
      if (odd_pad && wave == tiles - 1) put_b(Bout, wave + 1, zero2);
    };
    f32x4 acc[2];
    prefetch(a.A[0], NS0, T1);
    WDBG
    __syncthreads();
    WDBG
    // ---- forward: the wave's own output tile of each layer
    if (wave < T1) {
      bias_init(a.b[0], a.dims[1], acc);
      mma(NS0, B0, acc);
    }
    prefetch(a.A[1], NS1, T2);
    if (wave < T1) act(acc, B1, H1T, g1, (T1 & 1) != 0, T1);
    WDBG
    __syncthreads();
    WDBG
    if (wave < T2) {
      bias_init(a.b[1], a.dims[2], acc);
      mma(NS1, B1, acc);
    }
    if (L3) prefetch(a.A[2], NS2, T3);
    else prefetch(a.AT[3], NS4, T2);
    // the next tile's inputs (their landing zone, the input records, has had its last reader before the barrier above); behind
    // this phase's weight requests: loads return in order, an earlier place would make those wait for HBM
    request(tile + gridDim.x);
    if (wave < T2) act(acc, B2, H2T, g2, (T2 & 1) != 0, T2);
    WDBG
    __syncthreads();
    WDBG
    // (the two scratch sets alternate layer by layer, starting with set A for the linear layer's upstream gradient)
    wu32x4* const SA = SCR;
    wu32x4* const SB = SCR + WN * 128;
    wu32x4* const S2set = L3 ? SA : SB;       // where dZ2 (transposed) goes; dZ1 takes the other set
    wu32x4* const S1set = L3 ? SB : SA;
    if constexpr (L3) {
      if (wave < T3) {
        bias_init(a.b[2], a.dims[3], acc);
        mma(NS2, B2, acc);
      }
      prefetch(a.AT[3], NS4, T3);
      if (wave < T3) act(acc, B3, H3T, g3, (T3 & 1) != 0, T3);
      WDBG
      __syncthreads();
      WDBG
    }
    // ---- backward.  The linear last layer: its upstream gradient is in B4 / set A of the scratch
    if (wave < S4 * T4) dw(SA, L3 ? H3T : H2T, T4, TL, C4, dW4);
    if constexpr (L3) {
      if (wave < T3) {
        acc[0] = zero2[0], acc[1] = zero2[1];
        mma(NS4, B4, acc);
      }
      prefetch(a.AT[2], NS3, T2);
      if (wave < T3) to_dz(acc, g3, db3, B3, SB, (T3 & 1) != 0, T3);
      WDBG
      __syncthreads();
      WDBG
      if (wave < S3 * T3) dw(SB, H2T, T3, T2, C3, dW3);
    }
    if (wave < T2) {
      acc[0] = zero2[0], acc[1] = zero2[1];
      mma(L3 ? NS3 : NS4, L3 ? B3 : B4, acc);
    }
    prefetch(a.AT[1], NS2, T1);
    if (wave < T2) to_dz(acc, g2, db2, B2, S2set, (T2 & 1) != 0, T2);
    WDBG
    __syncthreads();
    WDBG
    if (wave < S2 * T2) dw(S2set, H1T, T2, T1, C2, dW2);
    if (wave < T1) {
      acc[0] = zero2[0], acc[1] = zero2[1];
      mma(NS2, B2, acc);
    }
    prefetch(a.AT[0], NS1, TI0);
    if (wave < T1) to_dz(acc, g1, db1, B1, S1set, (T1 & 1) != 0, T1);
    WDBG
    __syncthreads();
    WDBG
    if (wave < S1 * T1) dw(S1set, X0T, T1, TI0, C1, dW1);
    if (dX && wave < TI0) {
      acc[0] = zero2[0], acc[1] = zero2[1];
      mma(NS1, B1, acc);
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * wave + 4 * g + r;
          const int64_t n = n0 + 16 * sb + c;
          if (row < K0 && n < N) dX[(int64_t)row * N + n] = acc[sb][r] * dn[sb];
        }
    }
  }
  if (vmax >= 32768.f && a.overflow) atomicOr(a.overflow, 1u);
  // ---- the wave's accumulators -> this workgroup's gradient image (the layout of the fp32 kernel)
  using GI = GImg<TI0, T1, T2, T3, T4>;
  float* img = partial + (size_t)blockIdx.x * GI::TOTAL;
  auto put = [&](int base, int ncols_pad, int tout, int ntiles_in, int cn, const f32x4* accw) {
    const int sl = wave / tout, to = wave - sl * tout;
    for (int i = 0; i < cn; i++) {
      const int ti = sl * cn + i;
      if (ti < ntiles_in)
#pragma unroll
        for (int r = 0; r < 4; r++) img[base + (16 * to + 4 * g + r) * ncols_pad + 16 * ti + c] = accw[i][r];
    }
  };
  auto put_db = [&](int base, int to, const f32x4& v) {   // lane (sample c, g) holds the sums of rows 4 g + r over its samples
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float x = v[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) x += __shfl_xor(x, o, 64);
      if (c == 0) img[base + 16 * to + 4 * g + r] = x;
    }
  };
  if (wave < S1 * T1) put(GI::W1, TI0 * 16, T1, TI0, C1, dW1);
  if (wave < S2 * T2) put(GI::W2, T1 * 16, T2, T1, C2, dW2);
  if (L3 && wave < S3 * T3) put(GI::W3, T2 * 16, L3 ? T3 : 1, T2, C3, dW3);
  if (wave < S4 * T4) put(GI::W4, TL * 16, T4, TL, C4, dW4);
  if (wave < T1) put_db(GI::B1, wave, db1);
  if (wave < T2) put_db(GI::B2, wave, db2);
  if (L3 && wave < T3) put_db(GI::B3, wave, db3);
  if (wave < T4) put_db(GI::B4, wave, db4);
}

// ---- forward only (round 6): the colour network's evaluation on the same records.  No parameter gradients, so no accumulators,
// no T records, no gelu': 56 KB of LDS, two workgroups per CU.  Y [OUT, N] feature-major.
template <int TI0, int T1, int T2, int T3, int T4>
__global__ void __launch_bounds__(WN * 64, 2)
    mlp_wide_fwd_f16_kernel(WideArgsH a, int64_t N, const float* __restrict__ X, float* __restrict__ Y) {
  static_assert(TI0 <= WN && T1 <= WN && T2 <= WN && T3 <= WN && T4 <= WN && wns(TI0) * 2 <= WN, "one output tile per wave and layer");
  constexpr int NS0 = wns(TI0), NS1 = wns(T1), NS2 = wns(T2), NS3 = wns(T3);
  static_assert(NS0 <= 4 && NS1 <= 4 && NS2 <= 4 && NS3 <= 4, "four k-steps of weight records per phase");
  extern __shared__ __align__(16) wu32x4 wl[];
  wu32x4* B0 = wl;
  wu32x4* B1 = B0 + NS0 * 256;
  wu32x4* B2 = B1 + NS1 * 256;
  wu32x4* B3 = B2 + NS2 * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int K0 = a.dims[0], OUT = a.dims[4];
  const int64_t ntiles = (N + TS - 1) / TS;
  float vmax = 0.f;
  auto put_b = [&](wu32x4* Breg, int t, const f32x4 (&v)[2]) {
#pragma unroll
    for (int sb = 0; sb < 2; sb++) {
      uint32_t h0, l0, h1, l1;
      wsplit2(v[sb][0], v[sb][1], h0, l0);
      wsplit2(v[sb][2], v[sb][3], h1, l1);
      wu32x2* rec = reinterpret_cast<wu32x2*>(Breg + ((t >> 1) * 2 + sb) * 128 + lane) + (t & 1);
      rec[0] = wu32x2{h0, h1};
      rec[128] = wu32x2{l0, l1};
    }
  };
  auto request = [&](int64_t t2) {      // the next tile's inputs by LDS-DMA into the wave's own 2 KB of the input records (see the backward)
    const int64_t m0 = (t2 < ntiles ? t2 : ntiles - 1) * TS;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int c2 = lane_o & 15, g2 = lane_o >> 4;
    if (wave < NS0 * 2) {
      const int s = wave >> 1, sb = wave & 1;
      int64_t n = m0 + 16 * sb + c2;
      n = n < N ? n : N - 1;
      float* dst = reinterpret_cast<float*>(B0 + wave * 128);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int row = wkf(s, g2, j);
        row = row < K0 ? row : K0 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)row * N + n),
                                         (__attribute__((address_space(3))) void*)(dst + j * 64), 4, 0, 0);
      }
    }
  };
  const f32x4 zero2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  request(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave < NS0 * 2) {
      const int s = wave >> 1, sb = wave & 1;
      const float* src = reinterpret_cast<const float*>(B0 + wave * 128);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x[j] = (wkf(s, g, j) < K0 && n0 + 16 * sb + c < N) ? src[j * 64 + lane] : 0.f;
        vmax = fmaxf(vmax, fabsf(x[j]));
      }
      wu32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t h, l;
        wsplit2(x[2 * j], x[2 * j + 1], h, l);
        hi[j] = h;
        lo[j] = l;
      }
      wu32x4* rec = B0 + (s * 2 + sb) * 128 + lane;
      rec[0] = hi;
      rec[64] = lo;
    }
    auto layer = [&](const wu32x4* Aw, const float* bias, int out_true, int ns, int tiles, const wu32x4* Bin, f32x4 (&acc)[2]) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * wave + 4 * g + r;
        const float bv = (wave < tiles && row < out_true) ? bias[row] : 0.f;
        acc[0][r] = bv;
        acc[1][r] = bv;
      }
      if (wave < tiles) {
        WRec w[4];
#pragma unroll
        for (int s = 0; s < 4; s++)
          if (s < ns) w[s] = wload(Aw + ((size_t)(wave * ns + s) * 2) * 64 + lane);
#pragma unroll
        for (int s = 0; s < 4; s++)
          if (s < ns) {
#pragma unroll
            for (int sb = 0; sb < 2; sb++) acc[sb] = wmac3(w[s], wload(Bin + (s * 2 + sb) * 128 + lane), acc[sb]);
          }
      }
    };
    auto act = [&](f32x4 (&acc)[2], wu32x4* Bout, int tiles) {
      if (wave < tiles) {
        f32x4 h[2], gp;
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
          wgelu4(acc[sb], h[sb], gp);
#pragma unroll
          for (int r = 0; r < 4; r++) vmax = fmaxf(vmax, fabsf(h[sb][r]));
        }
        put_b(Bout, wave, h);
        if ((tiles & 1) && wave == tiles - 1) put_b(Bout, wave + 1, zero2);
      }
    };
    f32x4 acc[2];
    __syncthreads();
    layer(a.A[0], a.b[0], a.dims[1], NS0, T1, B0, acc);
    act(acc, B1, T1);
    __syncthreads();
    request(tile + gridDim.x);       // (the input records have had their last reader)
    layer(a.A[1], a.b[1], a.dims[2], NS1, T2, B1, acc);
    act(acc, B2, T2);
    __syncthreads();
    layer(a.A[2], a.b[2], a.dims[3], NS2, T3, B2, acc);
    act(acc, B3, T3);
    __syncthreads();
    layer(a.A[3], a.b[3], a.dims[4], NS3, T4, B3, acc);
    if (wave < T4) {
#pragma unroll
      for (int sb = 0; sb < 2; sb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * wave + 4 * g + r;
          const int64_t n = n0 + 16 * sb + c;
          if (row < OUT && n < N) Y[(int64_t)row * N + n] = acc[sb][r];
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (vmax >= 32768.f && a.overflow) atomicOr(a.overflow, 1u);
}

static uint32_t* wide_overflow_word() {
  static uint32_t* w = [] {
    uint32_t* h = nullptr;
    if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) != hipSuccess || !h) {
      (void)hipGetLastError();
      return (uint32_t*)nullptr;
    }
    h[0] = 0u;
    return h;
  }();
  return w;
}
int g_wide_form = 0;    // 1 = fp32 MFMA kernel, 2 = split-fp16 kernel (last launch)

// n_layers = 4: dims {in, h1, h2, h3, out}; n_layers = 3 (T3 == 0): dims {in, h1, h2, out} -- the linear last layer then sits in slot 3
// of the kernel's arrays and slot 2 stays empty
template <int TI0, int T1, int T2, int T3, int T4>
int wide_launch_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights, const float* const* biases,
                    const float* dY, float* dX, float* const* dW, float* const* db, hipStream_t st) {
  using GI = GImg<TI0, T1, T2, T3, T4>;
  constexpr int TL = T3 > 0 ? T3 : T2;
  const int in_t[4] = {TI0, T1, T2, TL}, out_t[4] = {T1, T2, T3, T4};       // tiles of every layer's input / output (slot 2: maybe none)
  const int src[4] = {0, 1, n_layers == 4 ? 2 : -1, n_layers - 1};          // which of the caller's layers a slot holds
  int kd[5] = {dims[0], dims[1], dims[2], n_layers == 4 ? dims[3] : dims[2], dims[n_layers]};
  size_t nrec = 0;      // 16-byte records of the weight images
  for (int l = 0; l < 4; l++) nrec += (size_t)out_t[l] * wns(in_t[l]) * 128 + (size_t)in_t[l] * wns(out_t[l]) * 128;
  const int64_t ntiles = (N + TS - 1) / TS;
  int64_t blocks = ntiles < 256 ? ntiles : 256;
  char* scratch = (char*)psdf::stream_scratch(nrec * 16 + (size_t)blocks * GI::TOTAL * sizeof(float), st);  // NULL while capturing
  if (!scratch) return PSDF_ERR_UNSUPPORTED;
  WideArgsH a;
  PackH pk;
  wu32x4* wp = reinterpret_cast<wu32x4*>(scratch);
  int nmax = 1;
  for (int l = 0; l < 4; l++) {
    const bool have = src[l] >= 0;
    pk.out[l] = have ? (l == 3 ? kd[4] : kd[l + 1]) : 0;
    pk.in[l] = have ? kd[l] : 0;
    pk.out_tiles[l] = have ? out_t[l] : 0;
    pk.in_tiles[l] = have ? in_t[l] : 0;
    pk.W[l] = have ? weights[src[l]] : nullptr;
    pk.A[l] = wp;
    wp += (size_t)out_t[l] * wns(in_t[l]) * 128;
    pk.AT[l] = wp;
    wp += (size_t)in_t[l] * wns(out_t[l]) * 128;
    a.A[l] = pk.A[l], a.AT[l] = pk.AT[l], a.b[l] = have ? biases[src[l]] : nullptr;
    const int n1 = out_t[l] * wns(in_t[l]) * 64, n2 = in_t[l] * wns(out_t[l]) * 64;
    nmax = n1 > nmax ? n1 : nmax;
    nmax = n2 > nmax ? n2 : nmax;
  }
  for (int i = 0; i < 5; i++) a.dims[i] = kd[i];
  a.overflow = wide_overflow_word();
  hipLaunchKernelGGL(mlp_wide_f16_pack_kernel, dim3((nmax + 255) / 256, 8), dim3(256), 0, st, pk);
  float* partial = reinterpret_cast<float*>(wp);
  const size_t lds_bytes = (size_t)((wns(TI0) + wns(T1) + wns(T2) + wns(T3) + wns(T4)) * 256 + (TI0 + T1 + T2 + T3) * 128 + 2 * WN * 128 + T4 * 128) * 16 + (size_t)T4 * 32 * 4;
  if (lds_bytes > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  auto kern = mlp_wide_bwd_f16_kernel<TI0, T1, T2, T3, T4>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WN * 64), lds_bytes, st, a, N, X, dY, dX, partial);
  WideArgs ar;
  for (int i = 0; i < 5; i++) ar.dims[i] = kd[i];
  for (int l = 0; l < 4; l++) ar.W[l] = ar.WT[l] = ar.b[l] = nullptr;
  float* gw[4];
  float* gb[4];
  for (int l = 0; l < 4; l++) {
    gw[l] = src[l] >= 0 ? dW[src[l]] : nullptr;
    gb[l] = src[l] >= 0 ? db[src[l]] : nullptr;
  }
  // (the small nets' images: 32 slices of the workgroup images instead of 8 -- their summing launch waits on loads, not on bytes)
  hipLaunchKernelGGL((mlp_wide_reduce_kernel<TI0, T1, T2, T3, T4>), dim3((GI::TOTAL + 255) / 256, GI::TOTAL < 20000 ? 32 : 8), dim3(256), 0, st, partial,
                     (int)blocks, ar, gw[0], gw[1], gw[2], gw[3], gb[0], gb[1], gb[2], gb[3]);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

template <int TI0, int T1, int T2, int T3, int T4>
int wide_forward_f16(const int* dims, int64_t N, const float* X, const float* const* weights, const float* const* biases, float* Y,
                     hipStream_t st) {
  const int tiles[5] = {TI0, T1, T2, T3, T4};
  size_t nrec = 0;
  for (int l = 0; l < 4; l++) nrec += (size_t)tiles[l + 1] * wns(tiles[l]) * 128;
  char* scratch = (char*)psdf::stream_scratch(nrec * 16, st);  // NULL while capturing
  if (!scratch) return PSDF_ERR_UNSUPPORTED;
  WideArgsH a;
  PackH pk;
  wu32x4* wp = reinterpret_cast<wu32x4*>(scratch);
  int nmax = 0;
  for (int l = 0; l < 4; l++) {
    pk.out[l] = dims[l + 1], pk.in[l] = dims[l], pk.out_tiles[l] = tiles[l + 1], pk.in_tiles[l] = tiles[l];
    pk.W[l] = weights[l];
    pk.A[l] = wp;
    pk.AT[l] = nullptr;
    wp += (size_t)tiles[l + 1] * wns(tiles[l]) * 128;
    a.A[l] = pk.A[l], a.AT[l] = nullptr, a.b[l] = biases[l];
    const int n1 = tiles[l + 1] * wns(tiles[l]) * 64;
    nmax = n1 > nmax ? n1 : nmax;
  }
  for (int i = 0; i < 5; i++) a.dims[i] = dims[i];
  a.overflow = wide_overflow_word();
  hipLaunchKernelGGL(mlp_wide_f16_pack_kernel, dim3((nmax + 255) / 256, 8), dim3(256), 0, st, pk);
  const int64_t ntiles = (N + TS - 1) / TS;
  int64_t blocks = ntiles < 512 ? ntiles : 512;
  const size_t lds_bytes = (size_t)((wns(TI0) + wns(T1) + wns(T2) + wns(T3)) * 256) * 16;
  auto kern = mlp_wide_fwd_f16_kernel<TI0, T1, T2, T3, T4>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WN * 64), lds_bytes, st, a, N, X, Y);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // namespace

extern "C" {

// Forward of the colour network's shape (dims[0] <= 112, dims[1], dims[2] <= 128, dims[3] <= 64, dims[4] <= 16, not both hidden widths
// <= 64; GELU between the layers, the last one linear) on the fp16 matrix pipe with two pieces per fp32 operand (round 6): X
// [dims[0], N] and Y [dims[4], N] feature-major, weights[l] / biases[l] the torch-layout parameters (for a LipshitzMLP: the NORMALISED
// weights).  -2 for other shapes, while a stream is being captured, and after a value beyond the fp16 range was met
// (psdf_mlp_forward evaluates every shape in fp32).  Replaces the torch.nn / LipshitzMLP forward of models.py:54-129,349-350.
int psdf_mlp_forward_wide_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                              const float* const* biases, float* Y, void* stream) {
  if (n_layers != 4 || !dims) return PSDF_ERR_UNSUPPORTED;
  if (N <= 0 || !X || !weights || !biases || !Y) return PSDF_ERR_ARG;
  for (int l = 0; l < 4; l++)
    if (!weights[l] || !biases[l]) return PSDF_ERR_ARG;
  const char* sp = getenv("PSDF_MLP_WIDE_SPLIT");
  if (sp && sp[0] == 'f' && sp[1] == '3') return PSDF_ERR_UNSUPPORTED;
  uint32_t* ov = wide_overflow_word();
  if (ov && *(volatile uint32_t*)ov) return PSDF_ERR_UNSUPPORTED;
  if (dims[0] <= 112 && dims[1] <= 128 && dims[2] <= 128 && dims[3] <= 64 && dims[4] <= 16 && !(dims[1] <= 64 && dims[2] <= 64))
    return wide_forward_f16<7, 8, 8, 4, 1>(dims, N, X, weights, biases, Y, (hipStream_t)stream);
  // the background density / feature net 52 -> 64 x 3 -> 65 (models.py:451-459)
  if (dims[0] <= 64 && dims[1] > 32 && dims[1] <= 64 && dims[2] > 32 && dims[2] <= 64 && dims[3] > 32 && dims[3] <= 64 &&
      dims[4] > 16 && dims[4] <= 80)
    return wide_forward_f16<4, 4, 4, 4, 5>(dims, N, X, weights, biases, Y, (hipStream_t)stream);
  return PSDF_ERR_UNSUPPORTED;
}

// 1 = the last psdf_mlp_backward_wide ran the fp32-MFMA kernel, 2 = the split-fp16 kernel; 0 = none yet (debug query, host only)
int psdf_mlp_backward_wide_form(void) { return g_wide_form; }
#if defined(PSDF_WIDE_DEBUG)
int psdf_wide_debug(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_dbg), 64 * 8); }
#endif

// replaces: LipshitzMLP.normalization, permuto_sdf_py/models/models.py:98-104 (W [out, in] row major, c [1] on the device)
int psdf_lipshitz_normalize_forward(int out, int in, const float* W, const float* c, float* Wn, void* stream) {
  if (out <= 0 || in <= 0) return PSDF_OK;
  if (!W || !c || !Wn) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(lipshitz_norm_fwd_kernel, dim3(out), dim3(64), 0, (hipStream_t)stream, in, W, c, Wn);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// replaces: torch autograd of the same; grad_W [out, in] is written, grad_c [1] is ACCUMULATED into
int psdf_lipshitz_normalize_backward(int out, int in, const float* W, const float* c, const float* grad_Wn, float* grad_W,
                                     float* grad_c, void* stream) {
  if (out <= 0 || in <= 0) return PSDF_OK;
  if (!W || !c || !grad_Wn || !grad_W || !grad_c) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(lipshitz_norm_bwd_kernel, dim3(out), dim3(64), 0, (hipStream_t)stream, in, W, c, grad_Wn, grad_W, grad_c);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// every layer of the net in one launch: W[l] [out[l], in[l]], c[l] [1], Wn[l] like W[l]; n_layers <= 8
int psdf_lipshitz_normalize_forward_multi(int n_layers, const int* out, const int* in, const float* const* W, const float* const* c,
                                          float* const* Wn, void* stream) {
  if (n_layers <= 0) return PSDF_OK;
  if (n_layers > LIP_MAX_LAYERS || !out || !in || !W || !c || !Wn) return PSDF_ERR_ARG;
  LipLayers p{};
  int rows = 0;
  for (int l = 0; l < n_layers; l++) {
    if (out[l] <= 0 || in[l] <= 0 || !W[l] || !c[l] || !Wn[l]) return PSDF_ERR_ARG;
    p.out[l] = out[l], p.in[l] = in[l], p.W[l] = W[l], p.c[l] = c[l], p.Wn[l] = Wn[l];
    rows = out[l] > rows ? out[l] : rows;
  }
  hipLaunchKernelGGL(lipshitz_norm_multi_kernel, dim3(rows, n_layers), dim3(64), 0, (hipStream_t)stream, p, 0);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// grad_W[l] is written, grad_c[l] [1] is ACCUMULATED into
int psdf_lipshitz_normalize_backward_multi(int n_layers, const int* out, const int* in, const float* const* W, const float* const* c,
                                           const float* const* grad_Wn, float* const* grad_W, float* const* grad_c, void* stream) {
  if (n_layers <= 0) return PSDF_OK;
  if (n_layers > LIP_MAX_LAYERS || !out || !in || !W || !c || !grad_Wn || !grad_W || !grad_c) return PSDF_ERR_ARG;
  LipLayers p{};
  int rows = 0;
  for (int l = 0; l < n_layers; l++) {
    if (out[l] <= 0 || in[l] <= 0 || !W[l] || !c[l] || !grad_Wn[l] || !grad_W[l] || !grad_c[l]) return PSDF_ERR_ARG;
    p.out[l] = out[l], p.in[l] = in[l], p.W[l] = W[l], p.c[l] = c[l], p.G[l] = grad_Wn[l], p.Wn[l] = grad_W[l], p.dc[l] = grad_c[l];
    rows = out[l] > rows ? out[l] : rows;
  }
  hipLaunchKernelGGL(lipshitz_norm_multi_kernel, dim3(rows, n_layers), dim3(64), 0, (hipStream_t)stream, p, 1);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// Same contract as psdf_mlp_backward (include/psdf.h) for 4-layer nets whose dW does not fit one wave's registers:
//   * dims[0] <= 112, dims[1], dims[2] <= 128, dims[3] <= 64, dims[4] <= 16 (not both hidden widths <= 64): the reference's colour
//     network, LipshitzMLP 111 -> 128 -> 128 -> 64 -> 3 (models.py:349-350);
//   * dims[0..3] <= 64, 16 < dims[4] <= 80: the background density / feature net 52 -> 64 x 3 -> 65 (models.py:451-459) and
//     64 x 3 -> 33 (round 3);
// -2 otherwise.
int psdf_mlp_backward_wide(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                           const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                           void* stream) {
  if ((n_layers != 4 && n_layers != 3) || !dims || !dW || !db) return PSDF_ERR_UNSUPPORTED;
  if (N <= 0 || !X || !weights || !biases || !dY) return PSDF_ERR_ARG;
  for (int l = 0; l < n_layers; l++)
    if (!weights[l] || !biases[l] || !dW[l] || !db[l]) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n_layers == 3) {
    // two hidden layers: the background colour head 80 -> 64 -> 64 -> 3 (models.py:463-469), split-fp16 kernel only (round 6)
    const char* sp3 = getenv("PSDF_MLP_WIDE_SPLIT");
    uint32_t* ov3 = wide_overflow_word();
    if ((sp3 && sp3[0] == 'f' && sp3[1] == '3') || (ov3 && *(volatile uint32_t*)ov3)) return PSDF_ERR_UNSUPPORTED;
    if (dims[0] > 64 && dims[0] <= 80 && dims[1] > 32 && dims[1] <= 64 && dims[2] > 32 && dims[2] <= 64 && dims[3] <= 16) {
      g_wide_form = 2;
      return wide_launch_f16<5, 4, 4, 0, 1>(3, dims, N, X, weights, biases, dY, dX, dW, db, st);
    }
    return PSDF_ERR_UNSUPPORTED;
  }
  // PSDF_MLP_WIDE_SPLIT = f16 (default: two fp16 pieces per operand on the fp16 matrix pipe) | f32 (fp32 MFMAs); a value that left
  // the fp16 range in an earlier launch (host-mapped flag) switches the process to the fp32 kernel
  const char* sp = getenv("PSDF_MLP_WIDE_SPLIT");
  bool f16 = !(sp && sp[0] == 'f' && sp[1] == '3');
  uint32_t* ov = wide_overflow_word();
  if (ov && *(volatile uint32_t*)ov) {
    static bool warned = false;
    if (!warned) {
      warned = true;
      fprintf(stderr, "psdf: a value of the wide MLP backward left the fp16 range (|value| >= 32768): the fp32-MFMA kernel is used from "
                      "here on (PSDF_MLP_WIDE_SPLIT=f32 selects it outright)\n");
    }
    f16 = false;
  }
  g_wide_form = f16 ? 2 : 1;
  // the colour network (111 -> 128 -> 128 -> 64 -> 3) and anything that fits its tile counts with one output tile
  if (dims[0] <= 112 && dims[1] <= 128 && dims[2] <= 128 && dims[3] <= 64 && dims[4] <= 16 && !(dims[1] <= 64 && dims[2] <= 64)) {
    if (f16) {
      const int r = wide_launch_f16<7, 8, 8, 4, 1>(4, dims, N, X, weights, biases, dY, dX, dW, db, st);
      if (r != PSDF_ERR_UNSUPPORTED) return r;
      g_wide_form = 1;
    }
    return wide_launch<7, 8, 8, 4, 1>(dims, N, X, weights, biases, dY, dX, dW, db, st);
  }
  // the background density / feature net (52 -> 64 x 3 -> 65) and its 33-output sibling: up to 80 outputs, 64-wide hidden layers
  if (dims[0] <= 64 && dims[1] <= 64 && dims[2] <= 64 && dims[3] <= 64 && dims[4] > 16 && dims[4] <= 80) {
    if (f16) {
      const int r = wide_launch_f16<4, 4, 4, 4, 5>(4, dims, N, X, weights, biases, dY, dX, dW, db, st);
      if (r != PSDF_ERR_UNSUPPORTED) return r;
      g_wide_form = 1;
    }
    return wide_launch<4, 4, 4, 4, 5>(dims, N, X, weights, biases, dY, dX, dW, db, st);
  }
  return PSDF_ERR_UNSUPPORTED;
}

}  // extern "C"
