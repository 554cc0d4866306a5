// Prototype v3 of the full split-bf16 MLP backward (36-64-64-64-1): 16-SAMPLE tiles on v_mfma_f32_16x16x32_bf16.
// v1 (tools/prototypes/mlp_bwd_split_bf16.hip, verified on the GPU) and v2 (transposes on the matrix pipe) carry ~500 registers of
// live state per wave with 32-sample tiles and spill; halving the tile halves the per-sample state (activations and
// derivatives: 16 registers per layer instead of 32) while the 176 accumulator registers stay -- about 230 VGPRs + 176
// AGPRs.  Price: the dW products run with half-filled k (16 samples in the 32 k-slots), matrix time that is otherwise idle.
// Lane maps (c = lane & 15, g = lane >> 4): A[m][k]: lane (m = c, g), k-slot (g, j); B[k][n]: lane (n = c, g), k-slot (g, j);
// D[m][n]: lane (n = c, g), register r = row 4 g + r.  Chained k order of a k-step s (= D tiles 2s, 2s+1 of the previous
// layer): slot (g, j) <-> feature kf = 32 s + 16 (j >> 2) + 4 g + (j & 3).  Transposes = products with a 0/1 operand, two per
// k-step (one per 16-feature tile); a feature-lane tile holds samples 4 g + r, which fill k-slots (g, 0..3) of a dW operand.
// The tile loop is emulated lane by lane in numpy (tools/prototypes/emulate_bwd_v3.py: all nine gradients match float64 to 4e-7).
// *** NOT YET RUN ON THE GPU (the round's GPU budget was spent): treat every number it prints as unverified. ***
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 tools/prototypes/mlp_bwd_split_bf16_v3.hip \
//         -o tools/mlp_bwd_split_bf16_v3
// (by default every MFMA writes AGPRs and each chain / transpose result is copied out with v_accvgpr_read: 948 copies and
//  80 spilled dwords per tile; with the VGPR form allowed the compiler keeps only accumulators in AGPRs: 583 and 24)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr int K0 = 36, HID = 64, NT = 4 /* 16-feature tiles of a hidden layer */, NT0 = 3 /* tiles covering K0 */;
__host__ __device__ inline int kf(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

// ------------------------------------------------------------------ LDS image (units: 16-byte lane records)
// every layer: [tile][k-step 2][piece 3][lane 64]
static constexpr int RECL = NT * 2 * 3 * 64, RECT0 = NT0 * 2 * 3 * 64;
static constexpr int OFF_W0 = 0, OFF_W1 = RECL, OFF_W2 = 2 * RECL, OFF_T2 = 3 * RECL, OFF_T1 = 4 * RECL, OFF_T0 = 5 * RECL;
static constexpr int OFF_F32 = 5 * RECL + RECT0;
static constexpr int TAIL_FLOATS = 3 * HID + HID + 1;  // biases of the three hidden layers, final weights, final bias
static constexpr size_t IMG_BYTES = (size_t)OFF_F32 * 16 + TAIL_FLOATS * 4;
static constexpr int NWAVES = 4;
static constexpr size_t IMG_ALIGNED = (IMG_BYTES + 15) / 16 * 16;
// per-wave staging of the next tile's inputs (LDS-DMA, double buffered): X rows [K0][16 samples] + 64 floats of dY
static constexpr int STAGE_FLOATS = K0 * 16 + 64;
#ifdef PREFETCH_LDS
static constexpr size_t LDS_BYTES = IMG_ALIGNED + (size_t)NWAVES * 2 * STAGE_FLOATS * 4;
#else
static constexpr size_t LDS_BYTES = IMG_ALIGNED;
#endif
// gradient image (floats): dW1 [64][64 (36 used)], dW2 [64][64], dW3 [64][64], db1, db2, db3 [64], dW4 [64], db4
static constexpr int G_W1 = 0, G_W2 = 4096, G_W3 = 8192, G_B1 = 12288, G_B2 = 12352, G_B3 = 12416, G_W4 = 12480, G_B4 = 12544,
                     G_TOTAL = 12545;

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
// gelu and its derivative Phi(z) + z phi(z) from one erf and one exp
#if !defined(GELU_EXP2_POLY)
__device__ __forceinline__ void gelu_both(float z, float& hval, float& gprime) {
  const float cdf = fmaf(0.5f, erf_fast(z * 0.70710678118654752440f), 0.5f);
  const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
  hval = z * cdf;
  gprime = fmaf(z, pdf, cdf);
}
#else
// tools/gelu_fit.py: e = Phi(-t) = exp2(P8(t)), t = min(|z|, 5.75); gelu = max(z, 0) - t e; gelu' = (z < 0 ? e : 1 - e) + zc phi(t)
__device__ __forceinline__ void gelu_both(float z, float& hval, float& gprime) {
  const float t = fminf(fabsf(z), 5.75f);
  float p = -2.772052994e-06f;
  p = fmaf(p, t, 3.862077210e-05f);
  p = fmaf(p, t, -1.825476502e-04f);
  p = fmaf(p, t, -1.458701736e-04f);
  p = fmaf(p, t, 7.075471804e-03f);
  p = fmaf(p, t, -5.250502750e-02f);
  p = fmaf(p, t, -4.592049122e-01f);
  p = fmaf(p, t, -1.151105762e+00f);
  p = fmaf(p, t, -1.000000000e+00f);
  const float e = __builtin_amdgcn_exp2f(p);
  hval = fmaf(-t, e, fmaxf(z, 0.f));
  const float cdf = z < 0.f ? e : 1.0f - e;
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(t * t * -0.72134752044448170368f);
  gprime = fmaf(copysignf(t, z), pdf, cdf);
}
#endif
#if defined(GELU_PACKED)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void gelu_both2(f32x2 z, f32x2& hval, f32x2& gprime) {
  const f32x2 t = {fminf(fabsf(z.x), 5.75f), fminf(fabsf(z.y), 5.75f)};
  f32x2 p = {-2.772052994e-06f, -2.772052994e-06f};
  p = pkfma(p, t, f32x2{3.862077210e-05f, 3.862077210e-05f});
  p = pkfma(p, t, f32x2{-1.825476502e-04f, -1.825476502e-04f});
  p = pkfma(p, t, f32x2{-1.458701736e-04f, -1.458701736e-04f});
  p = pkfma(p, t, f32x2{7.075471804e-03f, 7.075471804e-03f});
  p = pkfma(p, t, f32x2{-5.250502750e-02f, -5.250502750e-02f});
  p = pkfma(p, t, f32x2{-4.592049122e-01f, -4.592049122e-01f});
  p = pkfma(p, t, f32x2{-1.151105762e+00f, -1.151105762e+00f});
  p = pkfma(p, t, f32x2{-1.000000000e+00f, -1.000000000e+00f});
  const f32x2 e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  hval = pkfma(-t, e, f32x2{fmaxf(z.x, 0.f), fmaxf(z.y, 0.f)});
  const f32x2 cdf = {z.x < 0.f ? e.x : 1.0f - e.x, z.y < 0.f ? e.y : 1.0f - e.y};
  const f32x2 q = t * t * f32x2{-0.72134752044448170368f, -0.72134752044448170368f};
  const f32x2 pdf = f32x2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)} * f32x2{0.3989422804014327f, 0.3989422804014327f};
  gprime = pkfma(f32x2{copysignf(t.x, z.x), copysignf(t.y, z.y)}, pdf, cdf);
}
#endif

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct BP {  // the three bf16 pieces of one 8-element operand
  bf16x8 p[3];
};
__device__ __forceinline__ uint32_t top_pair(float hi, float lo) {  // {top half of hi, top half of lo}
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// eight fp32 -> three bf16x8 pieces by truncation of the running remainder
__device__ __forceinline__ void split8(const float (&x)[8], BP& o) {
  float r1[8], r2[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    r1[j] = x[j] - __uint_as_float(__float_as_uint(x[j]) & 0xFFFF0000u);
    r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xFFFF0000u);
  }
  u32x4 q0, q1, q2;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    q0[i] = top_pair(x[2 * i + 1], x[2 * i]);
    q1[i] = top_pair(r1[2 * i + 1], r1[2 * i]);
    q2[i] = top_pair(r2[2 * i + 1], r2[2 * i]);
  }
  o.p[0] = __builtin_bit_cast(bf16x8, q0);
  o.p[1] = __builtin_bit_cast(bf16x8, q1);
  o.p[2] = __builtin_bit_cast(bf16x8, q2);
}
// four fp32 (a feature-lane tile: samples 4 g + r) -> dW operand pieces: k-slots (g, 0..3), slots (g, 4..7) zero
__device__ __forceinline__ void split4(const f32x4& t, BP& o) {
  float r1[4], r2[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    r1[j] = t[j] - __uint_as_float(__float_as_uint(t[j]) & 0xFFFF0000u);
    r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xFFFF0000u);
  }
  const u32x4 q0 = {top_pair(t[1], t[0]), top_pair(t[3], t[2]), 0u, 0u};
  const u32x4 q1 = {top_pair(r1[1], r1[0]), top_pair(r1[3], r1[2]), 0u, 0u};
  const u32x4 q2 = {top_pair(r2[1], r2[0]), top_pair(r2[3], r2[2]), 0u, 0u};
  o.p[0] = __builtin_bit_cast(bf16x8, q0);
  o.p[1] = __builtin_bit_cast(bf16x8, q1);
  o.p[2] = __builtin_bit_cast(bf16x8, q2);
}
// bf16-valued registers of a transposed piece -> dW operand
__device__ __forceinline__ bf16x8 pack4(const f32x4& v) {
  const u32x4 q = {top_pair(v[1], v[0]), top_pair(v[3], v[2]), 0u, 0u};
  return __builtin_bit_cast(bf16x8, q);
}

// out[t] += W(tile t, k-step s) x operand pieces: six products, smallest first; two tiles at a time so that consecutive
// MFMAs go to different accumulators.  w_s -> record [t = 0][s][piece 0][lane]; tile stride = 2*3*64 records.
template <int NTILE>
__device__ __forceinline__ void mac16(f32x4 (&out)[NTILE], const BP& b, const u32x4* __restrict__ w_s) {
#pragma unroll
  for (int t0 = 0; t0 < NTILE; t0 += 2) {
    bf16x8 a[2][3];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int p = 0; p < 3; p++)
        if (t0 + dt < NTILE) a[dt][p] = __builtin_bit_cast(bf16x8, w_s[(t0 + dt) * 384 + p * 64]);
#define PROD(PA, PB)                                                                  \
  _Pragma("unroll") for (int dt = 0; dt < 2; dt++) if (t0 + dt < NTILE) out[t0 + dt] = \
      MFMA16(a[dt][PA], b.p[PB], out[t0 + dt]);
    PROD(2, 0) PROD(1, 1) PROD(0, 2) PROD(1, 0) PROD(0, 1) PROD(0, 0)
#undef PROD
  }
}
// B operand of k-step s from the D tiles 2s, 2s+1 of an activation
__device__ __forceinline__ void step_operand(const f32x4 (&act)[NT], int s, float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    x[j] = act[2 * s][j];
    x[4 + j] = act[2 * s + 1][j];
  }
}
// 0/1 operand that selects the 16 features of tile 2s+u out of a k-step (the same for every s)
__device__ __forceinline__ bf16x8 ident_op(int u, int lane) {
  const int c = lane & 15, g = lane >> 4;
  u32x4 q;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j0 = 2 * i, j1 = 2 * i + 1;
    const uint32_t lo = ((j0 >> 2) == u && 4 * g + (j0 & 3) == c) ? 0x3F80u : 0u;
    const uint32_t hi = ((j1 >> 2) == u && 4 * g + (j1 & 3) == c) ? 0x3F80u : 0u;
    q[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(bf16x8, q);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// fp32 feature-lane tile (exact: the three pieces sum to the value): register r of lane (f, g) = feature f, sample 4 g + r
__device__ __forceinline__ f32x4 transpose_f32(const BP& b, bf16x8 id) {
  f32x4 o = zero4();
  o = MFMA16(b.p[2], id, o);
  o = MFMA16(b.p[1], id, o);
  o = MFMA16(b.p[0], id, o);
  return o;
}
// piece-wise transpose: dW operand pieces of a 16-feature tile, plus this lane's fp32 sum for the bias gradient
__device__ __forceinline__ void transpose_pieces(const BP& b, bf16x8 id, BP& out, float& sum) {
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const f32x4 o = MFMA16(b.p[p], id, zero4());
    sum += (o[0] + o[1]) + (o[2] + o[3]);
    out.p[p] = pack4(o);
  }
}
__device__ __forceinline__ f32x4 dw_mac(f32x4 acc, const BP& A, const BP& B) {
  acc = MFMA16(A.p[2], B.p[0], acc);
  acc = MFMA16(A.p[1], B.p[1], acc);
  acc = MFMA16(A.p[0], B.p[2], acc);
  acc = MFMA16(A.p[1], B.p[0], acc);
  acc = MFMA16(A.p[0], B.p[1], acc);
  acc = MFMA16(A.p[0], B.p[0], acc);
  return acc;
}
template <int NTILE>
__device__ __forceinline__ void bias_init(f32x4 (&acc)[NTILE], const float* __restrict__ b, int g) {
#pragma unroll
  for (int t = 0; t < NTILE; t++) acc[t] = *reinterpret_cast<const f32x4*>(b + 16 * t + 4 * g);
}
template <int NTILE>
__device__ __forceinline__ void zero_init(f32x4 (&acc)[NTILE]) {
#pragma unroll
  for (int t = 0; t < NTILE; t++) acc[t] = zero4();
}
// in place: acc <- gelu(acc), gp <- gelu'(acc)
__device__ __forceinline__ void act_both(f32x4 (&acc)[NT], f32x4 (&gp)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; t++) {
#if defined(GELU_PACKED)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      f32x2 hv, d;
      gelu_both2(f32x2{acc[t][r], acc[t][r + 1]}, hv, d);
      acc[t][r] = hv.x; acc[t][r + 1] = hv.y;
      gp[t][r] = d.x; gp[t][r + 1] = d.y;
    }
#else
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float hv, d;
      gelu_both(acc[t][r], hv, d);
      acc[t][r] = hv;
      gp[t][r] = d;
    }
#endif
  }
}
// chain layer over the two k-steps of `in`; per_step(s, pieces) sees the operand pieces of each k-step
template <int NTILE, typename F>
__device__ __forceinline__ void chain(const f32x4 (&in)[NT], f32x4 (&out)[NTILE], const u32x4* __restrict__ w, int lane,
                                      F&& per_step) {
#pragma unroll
  for (int s = 0; s < 2; s++) {
    float x[8];
    step_operand(in, s, x);
    BP b;
    split8(x, b);
    mac16<NTILE>(out, b, w + s * 192 + lane);
    per_step(s, b);
  }
}
// backward of one layer: dH chain (hands the pieces of dZ to the transposes), then dW[to][ti] += dZ(to) x H(ti)
template <int NTO, int NTI>
__device__ __forceinline__ void layer_bwd(const f32x4 (&dz)[NT], f32x4 (&dh)[NTO], const u32x4* __restrict__ wT, int lane,
                                          const bf16x8 (&id)[2], const f32x4 (&hT)[NTI], f32x4 (&dW)[NT][NTI], float (&db)[NT]) {
#ifdef DW_PER_STEP
  // the transposed pieces of a k-step's two dZ tiles are consumed at once (24 registers live instead of 48); the price is
  // that the pieces of H are cut twice
  chain<NTO>(dz, dh, wT, lane, [&](int s, const BP& b) {
    BP A0, A1;
    transpose_pieces(b, id[0], A0, db[2 * s]);
    transpose_pieces(b, id[1], A1, db[2 * s + 1]);
#pragma unroll
    for (int ti = 0; ti < NTI; ti++) {
      BP B;
      split4(hT[ti], B);
      dW[2 * s][ti] = dw_mac(dW[2 * s][ti], A0, B);
      dW[2 * s + 1][ti] = dw_mac(dW[2 * s + 1][ti], A1, B);
    }
  });
#else
  BP A[NT];
  chain<NTO>(dz, dh, wT, lane, [&](int s, const BP& b) {
    transpose_pieces(b, id[0], A[2 * s], db[2 * s]);
    transpose_pieces(b, id[1], A[2 * s + 1], db[2 * s + 1]);
  });
#pragma unroll
  for (int ti = 0; ti < NTI; ti++) {
    BP B;
    split4(hT[ti], B);
#pragma unroll
    for (int to = 0; to < NT; to++) dW[to][ti] = dw_mac(dW[to][ti], A[to], B);
  }
#endif
}

__global__ void __launch_bounds__(NWAVES * 64, 1)
    bwdk(int64_t N, const float* __restrict__ X, const float* __restrict__ dY, const u32x4* __restrict__ img,
         float* __restrict__ dX, float* __restrict__ partial) {
  extern __shared__ __align__(16) u32x4 lds[];
  constexpr int NREC = (int)(IMG_ALIGNED / 16);
  for (int i = threadIdx.x; i < NREC; i += NWAVES * 64) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int lane_k = lane;
  const bf16x8 id[2] = {ident_op(0, lane), ident_op(1, lane)};
  f32x4 dW1[NT][NT0], dW2[NT][NT], dW3[NT][NT];
#pragma unroll
  for (int to = 0; to < NT; to++) {
#pragma unroll
    for (int ti = 0; ti < NT; ti++) dW2[to][ti] = dW3[to][ti] = zero4();
#pragma unroll
    for (int ti = 0; ti < NT0; ti++) dW1[to][ti] = zero4();
  }
  float db1[NT] = {0.f, 0.f, 0.f, 0.f}, db2[NT] = {0.f, 0.f, 0.f, 0.f}, db3[NT] = {0.f, 0.f, 0.f, 0.f},
        dw4[NT] = {0.f, 0.f, 0.f, 0.f}, db4 = 0.f;
  const int64_t ntiles = N / 16;  // prototype: N is a multiple of 16
#ifdef PREFETCH_LDS
  // One wave per SIMD: nothing hides an HBM round trip (SQ counters of the version without this: 39 % of the wave's time
  // in s_waitcnt).  The inputs of the NEXT tile are requested at the top of the current one with global_load_lds.
  float* stage = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + IMG_ALIGNED) + wave * 2 * STAGE_FLOATS;
  auto prefetch = [&](int64_t t, float* buf) {
    const int64_t nn = t * 16 + c;
#pragma unroll
    for (int i = 0; i < K0 / 4; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)(4 * i + g) * N + nn),
                                       (__attribute__((address_space(3))) void*)(buf + i * 64), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dY + nn),
                                     (__attribute__((address_space(3))) void*)(buf + K0 * 16), 4, 0, 0);
  };
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave, tstride = (int64_t)gridDim.x * NWAVES;
  if (tile0 < ntiles) prefetch(tile0, stage);
  int cur = 0;
  for (int64_t tile = tile0; tile < ntiles; tile += tstride, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA completion is not tracked by the compiler
    const float* xb = stage + cur * STAGE_FLOATS;
#ifndef PREFETCH_POS
#define PREFETCH_POS 0
#endif
#define ISSUE_PREFETCH() if (tile + tstride < ntiles) prefetch(tile + tstride, stage + (cur ^ 1) * STAGE_FLOATS)
    if (PREFETCH_POS == 0) ISSUE_PREFETCH();
#else
#define PREFETCH_POS -1
#define ISSUE_PREFETCH()
  for (int64_t tile = (int64_t)blockIdx.x * NWAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * NWAVES) {
    asm volatile("" ::: "memory");
#endif
#ifdef REMAT_LANE
    // loop-invariant lane arithmetic (addresses, masks) is cheap to redo and expensive to keep: hoisted out of the loop it
    // ends up in scratch, and every scratch reload waits (vmcnt) for the LDS-DMA prefetch in flight
    int lane_l = lane_k;
    asm volatile("" : "+v"(lane_l));
    const int lane = lane_l, c = lane & 15, g = lane >> 4;
#endif
    const int64_t n0 = tile * 16, n = n0 + c;
    // ---------------- forward recompute; h1, h2 leave the sweep as fp32 feature-lane tiles
    f32x4 a[NT], g1[NT], b[NT], g2[NT], h1T[NT], h2T[NT];
    bias_init<NT>(a, tail, g);
    {
      float xs[2][8];
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int k = 32 * s + 8 * g + j;  // layer 0: natural k order (the image is packed to match)
#ifdef PREFETCH_LDS
          xs[s][j] = k < K0 ? xb[k * 16 + c] : 0.f;
#else
          xs[s][j] = k < K0 ? X[(int64_t)k * N + n] : 0.f;
#endif
        }
#pragma unroll
      for (int s = 0; s < 2; s++) {
        BP bx;
        split8(xs[s], bx);
        mac16<NT>(a, bx, lds + OFF_W0 + s * 192 + lane);
      }
    }
    act_both(a, g1);  // a = h1
    bias_init<NT>(b, tail + HID, g);
    chain<NT>(a, b, lds + OFF_W1, lane, [&](int s, const BP& p) {
      h1T[2 * s] = transpose_f32(p, id[0]);
      h1T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    act_both(b, g2);  // b = h2
    bias_init<NT>(a, tail + 2 * HID, g);
    chain<NT>(b, a, lds + OFF_W2, lane, [&](int s, const BP& p) {
      h2T[2 * s] = transpose_f32(p, id[0]);
      h2T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    f32x4 dz[NT];
    act_both(a, dz);  // a = h3, dz = gelu'(z3) for now
    // ---------------- output layer: dW4 = sum dy h3, db4 = sum dy, dZ3 = w4 dy gelu'(z3)
    {
#ifdef PREFETCH_LDS
      const f32x4 dyT = *reinterpret_cast<const f32x4*>(xb + K0 * 16 + 4 * g);  // samples 4 g + r
#else
      const f32x4 dyT = *reinterpret_cast<const f32x4*>(dY + n0 + 4 * g);  // samples 4 g + r
#endif
      db4 += (dyT[0] + dyT[1]) + (dyT[2] + dyT[3]);
#pragma unroll
      for (int s = 0; s < 2; s++) {
        float x[8];
        step_operand(a, s, x);
        BP p;
        split8(x, p);
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const f32x4 h3T = transpose_f32(p, id[u]);
          dw4[2 * s + u] += fmaf(h3T[0], dyT[0], fmaf(h3T[1], dyT[1], fmaf(h3T[2], dyT[2], h3T[3] * dyT[3])));
        }
      }
    }
#ifdef PREFETCH_LDS
    const float dy = xb[K0 * 16 + c];
#else
    const float dy = dY[n];
#endif
    const float* wf = tail + 3 * HID;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wf + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; r++) dz[t][r] *= w4[r] * dy;
    }
    // ---------------- layer 3
    if (PREFETCH_POS == 1) ISSUE_PREFETCH();
    zero_init<NT>(a);
    layer_bwd<NT, NT>(dz, a, lds + OFF_T2, lane, id, h2T, dW3, db3);  // a = dH2^T
#pragma unroll
    for (int t = 0; t < NT; t++) a[t] *= g2[t];                       // dZ2^T
    // ---------------- layer 2
    if (PREFETCH_POS == 2) ISSUE_PREFETCH();
    zero_init<NT>(dz);
    layer_bwd<NT, NT>(a, dz, lds + OFF_T1, lane, id, h1T, dW2, db2);  // dz = dH1^T
#pragma unroll
    for (int t = 0; t < NT; t++) dz[t] *= g1[t];                      // dZ1^T
    // ---------------- layer 1: H = X, read in feature-lane order straight from the feature-major rows
    if (PREFETCH_POS == 3) ISSUE_PREFETCH();
    f32x4 xT[NT0], dx[NT0];
#pragma unroll
    for (int u = 0; u < NT0; u++) {
      const int feat = 16 * u + c;
      xT[u] = zero4();
#ifdef PREFETCH_LDS
      if (feat < K0) xT[u] = *reinterpret_cast<const f32x4*>(xb + feat * 16 + 4 * g);
#else
      if (feat < K0) xT[u] = *reinterpret_cast<const f32x4*>(X + (int64_t)feat * N + n0 + 4 * g);
#endif
    }
    zero_init<NT0>(dx);
    layer_bwd<NT0, NT0>(dz, dx, lds + OFF_T0, lane, id, xT, dW1, db1);  // dx = dX^T
#pragma unroll
    for (int t = 0; t < NT0; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = 16 * t + 4 * g + r;
        if (k < K0) dX[(int64_t)k * N + n] = dx[t][r];
      }
  }
  // ---------------- wave accumulators -> workgroup image (the weight images are dead) -> this workgroup's slot
  __syncthreads();
  float* G = reinterpret_cast<float*>(lds);
  for (int e = threadIdx.x; e < G_TOTAL; e += NWAVES * 64) G[e] = 0.f;
  __syncthreads();
  for (int w = 0; w < NWAVES; w++) {  // one wave at a time: plain read-modify-write, no LDS float atomics
    if (wave == w) {
#pragma unroll
      for (int to = 0; to < NT; to++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = (16 * to + 4 * g + r) * 64;  // [out][in]
#pragma unroll
          for (int ti = 0; ti < NT; ti++) {
            G[G_W2 + row + 16 * ti + c] += dW2[to][ti][r];
            G[G_W3 + row + 16 * ti + c] += dW3[to][ti][r];
          }
#pragma unroll
          for (int ti = 0; ti < NT0; ti++) G[G_W1 + row + 16 * ti + c] += dW1[to][ti][r];
        }
#pragma unroll
      for (int t = 0; t < NT; t++) {  // lane (f = c, g) holds the partial of its four samples: add the four groups
        float v1 = db1[t], v2 = db2[t], v3 = db3[t], v4 = dw4[t];
        v1 += __shfl_xor(v1, 16, 64); v2 += __shfl_xor(v2, 16, 64); v3 += __shfl_xor(v3, 16, 64); v4 += __shfl_xor(v4, 16, 64);
        v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64); v3 += __shfl_xor(v3, 32, 64); v4 += __shfl_xor(v4, 32, 64);
        if (g == 0) {
          G[G_B1 + 16 * t + c] += v1;
          G[G_B2 + 16 * t + c] += v2;
          G[G_B3 + 16 * t + c] += v3;
          G[G_W4 + 16 * t + c] += v4;
        }
      }
      float b4 = db4;
      b4 += __shfl_xor(b4, 16, 64);
      b4 += __shfl_xor(b4, 32, 64);
      if (lane == 0) G[G_B4] += b4;
    }
    __syncthreads();
  }
  float* dst = partial + (size_t)blockIdx.x * G_TOTAL;
  for (int e = threadIdx.x; e < G_TOTAL; e += NWAVES * 64) dst[e] = G[e];
}

__global__ void reduce_images(const float* __restrict__ partial, int nimg, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= G_TOTAL) return;
  float s = 0.f;
  for (int b = 0; b < nimg; b++) s += partial[(size_t)b * G_TOTAL + e];
  out[e] = s;
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
  for (int lane = 0; lane < 64; lane++) {
    const int c = lane & 15, g = lane >> 4;
    for (int j = 0; j < 8; j++)
      for (int s = 0; s < 2; s++) {
        const int k0 = 32 * s + 8 * g + j, kc = kf(s, g, j);
        for (int t = 0; t < NT; t++) {
          const int row = 16 * t + c;
          put(OFF_W0, 2, t, s, lane, j, k0 < K0 ? W[0][(size_t)row * K0 + k0] : 0.f);
          put(OFF_W1, 2, t, s, lane, j, W[1][(size_t)row * HID + kc]);
          put(OFF_W2, 2, t, s, lane, j, W[2][(size_t)row * HID + kc]);
          // transposed images: row is an INPUT neuron of the layer, k runs over its OUTPUT neurons
          put(OFF_T2, 2, t, s, lane, j, W[2][(size_t)kc * HID + row]);
          put(OFF_T1, 2, t, s, lane, j, W[1][(size_t)kc * HID + row]);
          if (t < NT0) put(OFF_T0, 2, t, s, lane, j, row < K0 ? W[0][(size_t)kc * K0 + row] : 0.f);
        }
      }
  }
  float* tail = reinterpret_cast<float*>(img.data() + (size_t)OFF_F32 * 16);
  for (int l = 0; l < 3; l++) memcpy(tail + l * HID, B[l].data(), HID * 4);
  memcpy(tail + 3 * HID, W[3].data(), HID * 4);
  tail[4 * HID] = B[3][0];

  std::vector<float> dYh(N);
  for (auto& v : dYh) v = nd(rng);
  float *dXin, *dYd, *dXout, *dPart, *dGrad;
  u32x4* dI;
  hipMalloc(&dXin, X.size() * 4);
  hipMalloc(&dYd, N * 4);
  hipMalloc(&dXout, X.size() * 4);
  hipMalloc(&dPart, (size_t)256 * G_TOTAL * 4);
  hipMalloc(&dGrad, G_TOTAL * 4);
  hipMalloc(&dI, img.size());
  hipMemcpy(dI, img.data(), img.size(), hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)bwdk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
  printf("LDS per workgroup %zu B, gradient image %d floats\n", LDS_BYTES, G_TOTAL);

  bool lean = false;
  auto kernel = [&](int64_t blocks, int64_t n) {
    hipLaunchKernelGGL(bwdk, dim3((unsigned)blocks), dim3(NWAVES * 64), LDS_BYTES, 0, n, dXin, dYd, dI, dXout, dPart);
  };
  auto run = [&](int64_t n, const float* xh, const float* dyh) {  // xh: [K0][n] feature-major
    hipMemcpy(dXin, xh, (size_t)K0 * n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dYd, dyh, n * 4, hipMemcpyHostToDevice);
    int64_t blocks = n / 16 / NWAVES;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    kernel(blocks, n);
    hipLaunchKernelGGL(reduce_images, dim3((G_TOTAL + 255) / 256), dim3(256), 0, 0, dPart, (int)blocks, dGrad);
    return (int)blocks;
  };

  // ---- correctness at NC samples against float64
  const int64_t NC = 65536;
  std::vector<float> Xc((size_t)K0 * NC), dYc(NC);
  for (int k = 0; k < K0; k++)
    for (int64_t n = 0; n < NC; n++) Xc[(size_t)k * NC + n] = X[(size_t)k * N + n];
  for (int64_t n = 0; n < NC; n++) dYc[n] = dYh[n];
  std::vector<double> rW1((size_t)HID * K0, 0.0), rW2((size_t)HID * HID, 0.0), rW3((size_t)HID * HID, 0.0), rB1(HID, 0.0),
      rB2(HID, 0.0), rB3(HID, 0.0), rW4(HID, 0.0), rdX((size_t)(NC / 64) * K0, 0.0);
  double rB4 = 0;
  {
    auto gp = [](double v) { return 0.5 * (1.0 + std::erf(v * 0.70710678118654752440)) + v * 0.3989422804014327 * std::exp(-0.5 * v * v); };
    std::vector<double> x(K0), z[3], hh[3], dzl[3];
    for (int l = 0; l < 3; l++) { z[l].resize(HID); hh[l].resize(HID); dzl[l].resize(HID); }
    for (int64_t n = 0; n < NC; n++) {
      for (int k = 0; k < K0; k++) x[k] = Xc[(size_t)k * NC + n];
      const std::vector<double>* in = &x;
      for (int l = 0; l < 3; l++) {
        for (int o = 0; o < HID; o++) {
          double acc = B[l][o];
          for (int k = 0; k < dims[l]; k++) acc += (double)W[l][(size_t)o * dims[l] + k] * (*in)[k];
          z[l][o] = acc;
          hh[l][o] = 0.5 * acc * (1.0 + std::erf(acc * 0.70710678118654752440));
        }
        in = &hh[l];
      }
      const double dy = dYc[n];
      rB4 += dy;
      for (int o = 0; o < HID; o++) {
        rW4[o] += dy * hh[2][o];
        dzl[2][o] = (double)W[3][o] * dy * gp(z[2][o]);
      }
      for (int l = 2; l >= 1; l--)
        for (int k = 0; k < HID; k++) {
          double acc = 0;
          for (int o = 0; o < HID; o++) acc += (double)W[l][(size_t)o * HID + k] * dzl[l][o];
          dzl[l - 1][k] = acc * gp(z[l - 1][k]);
        }
      for (int o = 0; o < HID; o++) {
        rB1[o] += dzl[0][o];
        rB2[o] += dzl[1][o];
        rB3[o] += dzl[2][o];
        for (int k = 0; k < K0; k++) rW1[(size_t)o * K0 + k] += dzl[0][o] * x[k];
        for (int k = 0; k < HID; k++) {
          rW2[(size_t)o * HID + k] += dzl[1][o] * hh[0][k];
          rW3[(size_t)o * HID + k] += dzl[2][o] * hh[1][k];
        }
      }
      if (n % 64 == 0)
        for (int k = 0; k < K0; k++) {
          double acc = 0;
          for (int o = 0; o < HID; o++) acc += (double)W[0][(size_t)o * K0 + k] * dzl[0][o];
          rdX[(size_t)(n / 64) * K0 + k] = acc;
        }
    }
  }
  std::vector<float> G(G_TOTAL), dXg((size_t)K0 * NC);
  auto check = [&]() {
    run(NC, Xc.data(), dYc.data());
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return false; }
    hipMemcpy(G.data(), dGrad, G_TOTAL * 4, hipMemcpyDeviceToHost);
    hipMemcpy(dXg.data(), dXout, dXg.size() * 4, hipMemcpyDeviceToHost);
    double exmax = 0, exref = 0;
    for (int64_t n = 0; n < NC; n += 64)
      for (int k = 0; k < K0; k++) {
        exmax = std::fmax(exmax, std::fabs(rdX[(size_t)(n / 64) * K0 + k] - (double)dXg[(size_t)k * NC + n]));
        exref = std::fmax(exref, std::fabs(rdX[(size_t)(n / 64) * K0 + k]));
      }
    auto report = [&](const char* name, const double* ref, int rows, int cols, int goff, int gstride) {
      double e = 0, m = 0;
      for (int o = 0; o < rows; o++)
        for (int k = 0; k < cols; k++) {
          e = std::fmax(e, std::fabs(ref[(size_t)o * cols + k] - (double)G[goff + o * gstride + k]));
          m = std::fmax(m, std::fabs(ref[(size_t)o * cols + k]));
        }
      printf("  %-4s max |err| %.3e   max |ref| %.3e   rel %.2e\n", name, e, m, e / m);
    };
    printf("%s variant, check at N = %lld against float64:\n  dX   max |err| %.3e   max |ref| %.3e   rel %.2e\n",
           "16-sample tiles", (long long)NC, exmax, exref, exmax / exref);
    report("dW1", rW1.data(), HID, K0, G_W1, 64);
    report("dW2", rW2.data(), HID, HID, G_W2, 64);
    report("dW3", rW3.data(), HID, HID, G_W3, 64);
    report("db1", rB1.data(), 1, HID, G_B1, 0);
    report("db2", rB2.data(), 1, HID, G_B2, 0);
    report("db3", rB3.data(), 1, HID, G_B3, 0);
    report("dW4", rW4.data(), 1, HID, G_W4, 0);
    report("db4", &rB4, 1, 1, G_B4, 0);
    return true;
  };
  auto timing = [&]() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    run(N, X.data(), dYh.data());
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 8; i++) {
        kernel(256, N);
        hipLaunchKernelGGL(reduce_images, dim3((G_TOTAL + 255) / 256), dim3(256), 0, 0, dPart, 256, dGrad);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = std::fmin(best, ms / 8);
    }
    printf("%s variant, full backward bf16 x6: %.4f ms at N = %lld (%.1f TF algorithmic at 42240 FLOP/sample)\n",
           "16-sample tiles", best, (long long)N, 42240.0 * N / (best * 1e-3) / 1e12);
  };
  if (!check()) return 1;
  timing();
  return 0;
}
