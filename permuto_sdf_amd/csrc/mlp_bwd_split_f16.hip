// Backward of the 64x3 -> 1 SDF net on the fp16 MATRIX PIPE with TWO pieces per fp32 operand (round 3; the sibling of
// mlp_bwd_split.hip, which uses three bf16 pieces and six products).  a = a0 + a1, a0 = fp16(a) rounded toward zero (so that
// a - a0 is exact in fp32), a1 = fp16(a - a0): 11 + 11 mantissa bits; chains keep a0 b0 + a0 b1 + a1 b0 (error ~2^-22 |a b|),
// the dW products all four (the fourth rides in an otherwise empty K half).  Against the bf16 scheme: 274 instead of 480 MFMAs
// per 16-sample tile, operand splitting 5 instead of 9 VALU instructions per pair, a 95-KB instead of a 142-KB weight image
// (so every input width up to 64 gets double-buffered staging).  What makes it legitimate on gfx950 (measured with
// tools/prototypes/mlp_fwd_split_f16.hip): the matrix pipe HONOURS fp16 subnormal inputs, so a low piece below 2^-14 keeps an absolute
// precision of 2^-24 instead of being flushed.  fp16 has 5 exponent bits, hence two guards:
//   * the gradient chain of sample n is linear in dY[n], so it is evaluated on the MANTISSA of dY[n] (sign kept, magnitude
//     scaled into [2^4, 2^5)) and dX[n] is multiplied by 2^(e(n) - 4) at the store (exact): every sample keeps 22 bits RELATIVE TO ITSELF, whatever
//     the spread of dY over the batch (NeuS weights span many decades, and the lattice gradient is a sparse sum of dX rows);
//   * the parameter gradients sum over all samples, so there the factor goes to the other operand: H[n] * 2^(e(n) - e_max),
//     e_max from max|dY| of the launch (mlp_absmax_kernel): contributions of samples far below the largest dY lose relative
//     precision but keep an absolute one of 2^-28 of the largest contribution; the images are multiplied by 2^(e_max - 4) in the
//     summing launch (exact).  The forward recompute needs no scaling for activations and weights of ordinary size;
//     |values| >= 65504 would saturate -- such nets belong to the bf16 kernel (PSDF_MLP_BWD_SPLIT=bf16).
// Everything else -- 16-sample tiles on v_mfma_f32_16x16x32, one wave per SIMD, persistent dW accumulators, transposes as
// MFMAs against a 0/1 operand, LDS-DMA staging of the next tile, one gradient image per workgroup + a summing launch -- is the
// design of mlp_bwd_split.hip; see there for the measurements that led to it.
// Accuracy against float64: tests/test_gpu_mlp.py::test_split_f16_backward_matches_float64.  Built with
// -mllvm -amdgpu-mfma-vgpr-form=1.
#include "psdf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 64, NT = 4 /* 16-feature tiles of a hidden layer */;   // NT0 (template) = tiles covering the input: 3 (<= 48) or 4 (<= 64)
__host__ __device__ inline int kf(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

// ------------------------------------------------------------------ LDS image (units: 16-byte lane records)
// every layer: [tile][k-step 2][piece 2][lane 64]
constexpr int NP = 2;   // pieces per operand
constexpr int RECL = NT * 2 * NP * 64;
constexpr int OFF_W0 = 0, OFF_W1 = RECL, OFF_W2 = 2 * RECL, OFF_T2 = 3 * RECL, OFF_T1 = 4 * RECL, OFF_T0 = 5 * RECL;
constexpr int off_f32(int nt0) { return 5 * RECL + nt0 * 2 * NP * 64; }
constexpr int TAIL_FLOATS = 3 * HID + HID + 1;  // biases of the three hidden layers, final weights, final bias
#if defined(PSDF_F16_PROTO_OCC)
// MEASUREMENT BUILD ONLY (round 4, tools/r04_mlp_occupancy_proto.sh; never the shipped library): what would TWO waves per SIMD
// buy?  The workgroup-cooperative design of DESIGN.md "Next" keeps 1/4 of the dW accumulators per wave (44 registers instead of
// 176) so that eight waves share one CU.  This build emulates its per-wave resources without its LDS exchange: every dW product
// of a layer lands in one of FOUR accumulators (wrong sums, same MFMA and VALU instruction mix), eight waves per workgroup.
constexpr int NWAVES = 8;
#else
constexpr int NWAVES = 4;
#endif
constexpr size_t img_aligned(int nt0) { return ((size_t)off_f32(nt0) * 16 + TAIL_FLOATS * 4 + 15) / 16 * 16; }
// gradient image (floats): dW1 [64][64 (K0 used)], dW2 [64][64], dW3 [64][64], db1, db2, db3 [64], dW4 [64], db4
constexpr int G_W1 = 0, G_W2 = 4096, G_W3 = 8192, G_B1 = 12288, G_B2 = 12352, G_B3 = 12416, G_W4 = 12480, G_B4 = 12544,
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
// Three interchangeable evaluators; the kernel picks per instantiation (see gelu_both below).
// tools/gelu_fit_rational.py: gelu AND gelu' from ONE exponential and ONE reciprocal (the recompute needs both):
//   E = exp(-z^2/2), t = 1/(1 + p|z|), Phi(-|z|) = t P6(t) E, cdf = z < 0 ? Phi(-|z|) : 1 - Phi(-|z|),
//   gelu = z cdf, gelu' = cdf + z E / sqrt(2 pi).  17 instructions against ~30; error against float64: gelu 1.8e-7 |z|
//   (the fp32 formula 0.5 z (1 + erf(z / sqrt 2)) itself: 1.1e-7 |z|), gelu' 1.9e-7.
__device__ __forceinline__ void gelu_rational(float z, float& hval, float& gprime) {
  const float E = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(z), 0.39f, 1.0f));
  float q = 5.384693295e-02f;
  q = fmaf(q, t, -2.582434118e-01f);
  q = fmaf(q, t, 3.751679361e-01f);
  q = fmaf(q, t, -1.663514599e-02f);
  q = fmaf(q, t, 1.944366544e-01f);
  q = fmaf(q, t, 1.514270604e-01f);
  const float tail = q * t * E;
  const float cdf = z < 0.f ? tail : 1.0f - tail;
  hval = z * cdf;
  gprime = fmaf(z, E * 0.3989422804014327f, cdf);
}
// torch's formula 0.5 z (1 + erf(z / sqrt 2)): one erf (itself one exp) and one more exp
__device__ __forceinline__ void gelu_erf(float z, float& hval, float& gprime) {
  const float cdf = fmaf(0.5f, erf_fast(z * 0.70710678118654752440f), 0.5f);
  const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
  hval = z * cdf;
  gprime = fmaf(z, pdf, cdf);
}
// tools/gelu_fit.py: e = Phi(-t) = exp2(P8(t)), t = min(|z|, 5.75); gelu = max(z, 0) - t e (error 8.6e-8 |z| against float64, the
// fp32 erf formula itself has 1.06e-7 |z|); gelu' = (z < 0 ? e : 1 - e) + z phi(t) from the same e (1.5e-7)
__device__ __forceinline__ void gelu_poly(float z, float& hval, float& gprime) {
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
// Measured on the headline batch (profiles/r02_mlp_bwd_prototype_timings.txt): rational 1.37 ms, erf 1.47 ms, poly 1.47 ms
// for the double-staged instantiation (zero scratch in all three).  The widest instantiation (K0 > 48) is at the register
// limit and the rational form's extra live values spill there (44 B scratch; a spill reload waits for the LDS-DMA in
// flight), so it keeps erf.
template <bool RATIONAL>
__device__ __forceinline__ void gelu_both(float z, float& hval, float& gprime) {
  if constexpr (RATIONAL) gelu_rational(z, hval, gprime);
  else gelu_erf(z, hval, gprime);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

struct BP {  // the two fp16 pieces of one 8-element operand: p[0] = high, p[1] = low
  f16x8 p[NP];
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> {high pieces, low pieces}, each a packed pair (element 0 in the low half).  v_cvt_pkrtz rounds toward zero, so
// the remainder v - high is exact in fp32 (24 - 11 = 13 significant bits); the low piece keeps its top 11.
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const auto h2 = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  const f32x2 r = f32x2{x0, x1} - f32x2{(float)h2[0], (float)h2[1]};
  const auto l2 = __builtin_amdgcn_cvt_pkrtz(r.x, r.y);
  hi = __builtin_bit_cast(uint32_t, h2);
  lo = __builtin_bit_cast(uint32_t, l2);
}
__device__ __forceinline__ void split8(const float (&x)[8], BP& o) {
  u32x4 q0, q1;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t h, l;
    split2(x[2 * i], x[2 * i + 1], h, l);
    q0[i] = h;
    q1[i] = l;
  }
  o.p[0] = __builtin_bit_cast(f16x8, q0);
  o.p[1] = __builtin_bit_cast(f16x8, q1);
}
// dW operands.  A transposed tile gives a lane only four samples (k-slots (g, 0..3) of the 32-deep MFMA); slots (g, 4..7)
// carry ANOTHER PIECE of the same four samples, so that one MFMA sums two piece products:
//   [a0|a0] x [b0|b1] = a0 b0 + a0 b1,   [a1|a1] x [b0|b1] = a1 b0 + a1 b1   (the last one is free: its K half was empty)
// Two MFMAs per 16x16 block of dW (88 per tile; the bf16 scheme: 132).
struct AT {  // dZ side
  f16x8 t00, t11;
};
struct BT {  // H side
  f16x8 t01;
};
__device__ __forceinline__ f16x8 halves(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
  const u32x4 q = {a0, a1, b0, b1};
  return __builtin_bit_cast(f16x8, q);
}
// four fp32 (a feature-lane tile: samples 4 g + r) -> H-side operand
__device__ __forceinline__ void split4(const f32x4& t, BT& o) {
  uint32_t ha, la, hb, lb;
  split2(t[0], t[1], ha, la);
  split2(t[2], t[3], hb, lb);
  o.t01 = halves(ha, hb, la, lb);
}

// out[t] += W(tile t, k-step s) x operand pieces: three products, smallest first; two tiles at a time so that consecutive
// MFMAs go to different accumulators.  w_s -> record [t = 0][s][piece 0][lane]; tile stride = 2*NP*64 records.
template <int NTILE>
__device__ __forceinline__ void mac16(f32x4 (&out)[NTILE], const BP& b, const u32x4* __restrict__ w_s) {
#pragma unroll
  for (int t0 = 0; t0 < NTILE; t0 += 2) {
    f16x8 a[2][NP];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int p = 0; p < NP; p++)
        if (t0 + dt < NTILE) a[dt][p] = __builtin_bit_cast(f16x8, w_s[(t0 + dt) * (2 * NP * 64) + p * 64]);
#define PROD(PA, PB)                                                                  \
  _Pragma("unroll") for (int dt = 0; dt < 2; dt++) if (t0 + dt < NTILE) out[t0 + dt] = \
      MFMA16(a[dt][PA], b.p[PB], out[t0 + dt]);
    PROD(1, 0) PROD(0, 1) PROD(0, 0)
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
__device__ __forceinline__ f16x8 ident_op(int u, int lane) {
  const int c = lane & 15, g = lane >> 4;
  u32x4 q;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j0 = 2 * i, j1 = 2 * i + 1;
    const uint32_t lo = ((j0 >> 2) == u && 4 * g + (j0 & 3) == c) ? 0x3C00u : 0u;
    const uint32_t hi = ((j1 >> 2) == u && 4 * g + (j1 & 3) == c) ? 0x3C00u : 0u;
    q[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(f16x8, q);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// fp32 feature-lane tile (the two pieces sum exactly in fp32): register r of lane (f, g) = feature f, sample 4 g + r
__device__ __forceinline__ f32x4 transpose_f32(const BP& b, f16x8 id) {
  f32x4 o = zero4();
  o = MFMA16(b.p[1], id, o);
  o = MFMA16(b.p[0], id, o);
  return o;
}
// piece-wise transpose: dZ-side dW operands of a 16-feature tile, plus this lane's fp32 sum for the bias gradient
__device__ __forceinline__ void transpose_pieces(const BP& b, f16x8 id, AT& out, float& sum, const f32x4& rT) {
  uint32_t q[NP][2];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const f32x4 o = MFMA16(b.p[p], id, zero4());   // fp16-valued: the conversion back is exact
    // bias gradient: dZ of sample 4 g + r was evaluated on the mantissa of its dY; rT[r] restores the magnitude (globally scaled)
    sum += fmaf(o[0], rT[0], o[1] * rT[1]) + fmaf(o[2], rT[2], o[3] * rT[3]);
    q[p][0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(o[0], o[1]));
    q[p][1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(o[2], o[3]));
  }
  out.t00 = halves(q[0][0], q[0][1], q[0][0], q[0][1]);
  out.t11 = halves(q[1][0], q[1][1], q[1][0], q[1][1]);
}
__device__ __forceinline__ f32x4 dw_mac(f32x4 acc, const AT& A, const BT& B) {
  acc = MFMA16(A.t11, B.t01, acc);   // smallest first
  acc = MFMA16(A.t00, B.t01, acc);
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
// gelu_rational on a PAIR of values with packed fp32 arithmetic (v_pk_mul / v_pk_fma / v_pk_add: one issue slot for two
// elements; with one wave per SIMD an instruction costs ~5 cycles whatever it is).  Same operations in the same order as the
// scalar form (bit-identical); exp2, rcp, abs and the sign select stay per element: ~24 instructions per pair instead of 34.
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 sp2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ void gelu_rational2(f32x2 z, f32x2& hval, f32x2& gprime) {
  const f32x2 ea = (z * z) * sp2(-0.72134752044448170368f);
  const f32x2 E = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const f32x2 az = {fabsf(z.x), fabsf(z.y)};
  const f32x2 den = pk_fma2(az, sp2(0.39f), sp2(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f32x2 q = sp2(5.384693295e-02f);
  q = pk_fma2(q, t, sp2(-2.582434118e-01f));
  q = pk_fma2(q, t, sp2(3.751679361e-01f));
  q = pk_fma2(q, t, sp2(-1.663514599e-02f));
  q = pk_fma2(q, t, sp2(1.944366544e-01f));
  q = pk_fma2(q, t, sp2(1.514270604e-01f));
  const f32x2 tail = (q * t) * E;
  const f32x2 om = sp2(1.0f) - tail;
  const f32x2 cdf = {z.x < 0.f ? tail.x : om.x, z.y < 0.f ? tail.y : om.y};
  hval = z * cdf;
  gprime = pk_fma2(z, E * sp2(0.3989422804014327f), cdf);
}
// in place: acc <- gelu(acc), gp <- gelu'(acc)
template <bool RATIONAL>
__device__ __forceinline__ void act_both(f32x4 (&acc)[NT], f32x4 (&gp)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; t++) {
#if defined(PSDF_F16_PACKED_GELU)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      f32x2 hv, d;
      gelu_rational2(f32x2{acc[t][r], acc[t][r + 1]}, hv, d);
      acc[t][r] = hv.x;
      acc[t][r + 1] = hv.y;
      gp[t][r] = d.x;
      gp[t][r + 1] = d.y;
    }
#else
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float hv, d;
      gelu_both<RATIONAL>(acc[t][r], hv, d);
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
    mac16<NTILE>(out, b, w + s * (NP * 64) + lane);
    per_step(s, b);
  }
}
// backward of one layer: dH chain (hands the pieces of dZ to the transposes), then dW[to][ti] += dZ(to) x H(ti)
// backward of one layer: dH chain (hands the pieces of dZ to the transposes), then dW[to][ti] += dZ(to) x H(ti)
#if defined(PSDF_F16_PROTO_OCC)
#define PSDF_DW_COLS 1
#else
#define PSDF_DW_COLS NTI
#endif
template <int NTO, int NTI>
__device__ __forceinline__ void layer_bwd(const f32x4 (&dz)[NT], f32x4 (&dh)[NTO], const u32x4* __restrict__ wT, int lane,
                                          const f16x8 (&id)[2], const f32x4 (&hT)[NTI], f32x4 (&dW)[NT][PSDF_DW_COLS], float (&db)[NT],
                                          const f32x4& rT) {
  AT A[NT];
  chain<NTO>(dz, dh, wT, lane, [&](int s, const BP& b) {
    transpose_pieces(b, id[0], A[2 * s], db[2 * s], rT);
    transpose_pieces(b, id[1], A[2 * s + 1], db[2 * s + 1], rT);
  });
#pragma unroll
  for (int ti = 0; ti < NTI; ti++) {
    BT B;
    split4(hT[ti] * rT, B);      // H of sample 4 g + r carries that sample's dY magnitude (see the header)
#if defined(PSDF_F16_PROTO_OCC)
#pragma unroll
    for (int to = 0; to < NT; to++) dW[(to + ti) & 3][0] = dw_mac(dW[(to + ti) & 3][0], A[to], B);
#else
#pragma unroll
    for (int to = 0; to < NT; to++) dW[to][ti] = dw_mac(dW[to][ti], A[to], B);
#endif
  }
}

// 2^k with max|dY| * 2^k in [2^4, 2^5) (k clamped to +-100; 1 when dY is all zero / not finite): bits = max over the batch of
// the bit pattern of |dY| (mlp_absmax_kernel)
__device__ __forceinline__ int dy_scale(uint32_t bits, float& sc, float& isc) {
  const int e = (int)(bits >> 23) & 255;
  int k = (e == 0 || e == 255) ? 0 : 4 - (e - 127);
  k = k < -100 ? -100 : (k > 100 ? 100 : k);
  sc = __uint_as_float((uint32_t)(127 + k) << 23);
  isc = __uint_as_float((uint32_t)(127 - k) << 23);
  return k;
}
// dY of one sample -> (mantissa with sign scaled to a magnitude in [2^4, 2^5): the chain's intermediate values then sit where
// their low fp16 pieces are still normal numbers; 2^(e - 4), the factor that restores dX).  Zero and fp32-subnormal values
// give (0, 0): the sample contributes nothing.  Inf / NaN pass through as they are (and poison what they touch, as in any
// fp32 evaluation).  (Values below 2^-122 lose their factor to the exponent clamp: treated as zero.)
constexpr int CHAIN_EXP = 4;
// The H-side operand of the parameter-gradient products is H[n] * 2^(e(n) - e_max) split into two fp16 pieces, and fp16 ends at
// 2^-24: with activations of 1e-2 (the encoding features that feed layer 1 are that small) the LOW piece of even the largest-dY
// sample is already subnormal, and samples a few decades below the largest dY lose their contribution's precision altogether --
// measured with all 2 097 152 samples carrying gradient, dY spread over six decades plus one outlier: dW errors of 1.2e-3 ..
// 1.7e-3 of the largest entry (tests/test_gpu_hotpath_parity.py::test_cfg2_full_batch_dense_gradient_against_float64; reproduced
// by a numpy emulation of this arithmetic).  fp16 has as much room ABOVE as it lacks below: the operand is pre-scaled by 2^8
// (exact) and the factor is taken out again by the summing launch.  Cost: none.  Limit: |activation| * 2^8 must stay below
// 65504, i.e. inputs and hidden activations up to 255 (beyond: the piece saturates at the largest fp16 number -- a silently
// clipped contribution, not an inf; such nets belong to PSDF_MLP_BWD_SPLIT=bf16 like those whose forward exceeds 65504).
constexpr int H_PRESCALE_EXP = 8;
__device__ __forceinline__ void dy_parts(float dy, float& mant, float& pow2) {
  const uint32_t b = __float_as_uint(dy);
  const int ex = (int)(b >> 23) & 255;
  const bool special = ex == 255, zero = ex <= CHAIN_EXP;
  mant = zero ? 0.f : (special ? dy : __uint_as_float((b & 0x807FFFFFu) | ((uint32_t)(127 + CHAIN_EXP) << 23)));
  pow2 = zero ? 0.f : (special ? 1.f : __uint_as_float((uint32_t)(ex - CHAIN_EXP) << 23));
}

// X [K0, N], dY [1, N], dX [K0, N] (optional) feature-major; img = the LDS image (mlp_split_pack_kernel); partial
// [gridDim.x][G_TOTAL] receives this workgroup's gradient image (of dY * 2^k: mlp_split_reduce_kernel takes the factor out
// again).  rows4 = K0 rounded up to a multiple of 4.
// DOUBLE: the staged inputs are double buffered and serve the whole tile (K0 <= 36: it fits beside the image in 160 KB of
// LDS).  Otherwise ONE staging buffer per wave: it is read at the top of the tile and refilled at once for the next tile;
// the feature-lane copy of X that the last layer's dW needs comes from global memory (an L2 hit: the tile was just staged).
template <int NT0, bool DOUBLE>
__global__ void __launch_bounds__(NWAVES * 64, 1)
    mlp_bwd_split_f16_kernel(int64_t N, int K0, int rows4, const float* __restrict__ X, const float* __restrict__ dY,
                             const u32x4* __restrict__ img, const uint32_t* __restrict__ absmax, float* __restrict__ dX,
                             float* __restrict__ partial) {
  extern __shared__ __align__(16) u32x4 lds[];
  float sc, isc;
  const int kscale = dy_scale(absmax[0], sc, isc);      // 2^kscale * max|dY| in [2^4, 2^5)
  constexpr size_t IMG_ALIGNED = img_aligned(NT0);
  constexpr int OFF_F32 = off_f32(NT0);
  constexpr int NREC = (int)(IMG_ALIGNED / 16);
  for (int i = threadIdx.x; i < NREC; i += NWAVES * 64) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  const int lane_k = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f16x8 id[2] = {ident_op(0, lane_k), ident_op(1, lane_k)};
#if defined(PSDF_F16_PROTO_OCC)
  f32x4 dW1[NT][1], dW2[NT][1], dW3[NT][1];
#pragma unroll
  for (int to = 0; to < NT; to++) dW1[to][0] = dW2[to][0] = dW3[to][0] = zero4();
#else
  f32x4 dW1[NT][NT0], dW2[NT][NT], dW3[NT][NT];
#pragma unroll
  for (int to = 0; to < NT; to++) {
#pragma unroll
    for (int ti = 0; ti < NT; ti++) dW2[to][ti] = dW3[to][ti] = zero4();
#pragma unroll
    for (int ti = 0; ti < NT0; ti++) dW1[to][ti] = zero4();
  }
#endif
  float db1[NT] = {0.f, 0.f, 0.f, 0.f}, db2[NT] = {0.f, 0.f, 0.f, 0.f}, db3[NT] = {0.f, 0.f, 0.f, 0.f},
        dw4[NT] = {0.f, 0.f, 0.f, 0.f}, db4 = 0.f;
  const int64_t ntiles = (N + 15) / 16;
  const int stage_floats = rows4 * 16 + 64;
  float* stage = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + IMG_ALIGNED) + wave * (DOUBLE ? 2 : 1) * stage_floats;
  // The inputs of the NEXT tile are requested with global_load_lds while this one is computed.  16-byte form (N % 4 == 0):
  // one instruction brings 16 rows x 16 samples (lane = row 16 j + (lane >> 2), samples 4 (lane & 3) .. + 3) to
  // buf[row * 16 + sample]; the rows past the last multiple of 16 and dY come with the 4-byte form (lane = row 4 i + g,
  // sample c).  For K0 = 36 that is 2 + 1 + 1 instructions instead of 10 (each LDS-DMA instruction costs the wave ~100
  // cycles of issue).  Rows are clamped to the last real row, samples to the end of the batch.
  const bool wide_dma = (N & 3) == 0 && N >= 4;
  auto prefetch = [&](int64_t t, float* buf) {
    const int c = lane_k & 15, g = lane_k >> 4;
    int64_t nn = t * 16 + c;
    nn = nn < N ? nn : N - 1;
    int i0 = 0;
    if (wide_dma) {
      int64_t n4 = t * 16 + 4 * (lane_k & 3);
      n4 = n4 + 3 < N ? n4 : N - 4;
      const int n16 = rows4 >> 4;
      for (int j = 0; j < n16; j++) {
        const int k = 16 * j + (lane_k >> 2);   // < rows4; rows4 - K0 < 4 of them are padding
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)(k < K0 ? k : K0 - 1) * N + n4),
                                         (__attribute__((address_space(3))) void*)(buf + j * 256), 16, 0, 0);
      }
      i0 = n16 * 4;
    }
    for (int i = i0; i < (rows4 >> 2); i++) {
      int k = 4 * i + g;
      k = k < K0 ? k : K0 - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)k * N + nn),
                                       (__attribute__((address_space(3))) void*)(buf + i * 64), 4, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dY + nn),
                                     (__attribute__((address_space(3))) void*)(buf + rows4 * 16), 4, 0, 0);
  };
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave, tstride = (int64_t)gridDim.x * NWAVES;
  if (tile0 < ntiles) prefetch(tile0, stage);
  int cur = 0;
  for (int64_t tile = tile0; tile < ntiles; tile += tstride, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA completion is not tracked by the compiler
    const float* xb = stage + (DOUBLE ? cur : 0) * stage_floats;
    // loop-invariant lane arithmetic (addresses, masks) is cheap to redo and expensive to keep: hoisted out of the loop it
    // ends up in scratch, and every scratch reload waits (vmcnt) for the LDS-DMA prefetch in flight
    int lane_l = lane_k;
    asm volatile("" : "+v"(lane_l));
    const int lane = lane_l, c = lane & 15, g = lane >> 4;
    const int64_t n0 = tile * 16, n = n0 + c;
    const bool live = n < N;
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
          xs[s][j] = k < K0 ? xb[k * 16 + c] : 0.f;
        }
#pragma unroll
      for (int s = 0; s < 2; s++) {
        BP bx;
        split8(xs[s], bx);
        mac16<NT>(a, bx, lds + OFF_W0 + s * (NP * 64) + lane);
      }
    }
    // single staging buffer: take the two dY operands now and refill the buffer for the next tile at once
    f32x4 dyT_early = zero4();
    float dy_early = 0.f;
    if (!DOUBLE) {
      dyT_early = *reinterpret_cast<const f32x4*>(xb + rows4 * 16 + 4 * g);
      dy_early = xb[rows4 * 16 + c];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the LDS reads above have returned before the DMA overwrites
      if (tile + tstride < ntiles) prefetch(tile + tstride, stage);
    }
    act_both<true>(a, g1);  // a = h1
    bias_init<NT>(b, tail + HID, g);
    chain<NT>(a, b, lds + OFF_W1, lane, [&](int s, const BP& p) {
      h1T[2 * s] = transpose_f32(p, id[0]);
      h1T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    act_both<true>(b, g2);  // b = h2
    bias_init<NT>(a, tail + 2 * HID, g);
    chain<NT>(b, a, lds + OFF_W2, lane, [&](int s, const BP& p) {
      h2T[2 * s] = transpose_f32(p, id[0]);
      h2T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    f32x4 dz[NT];
    act_both<true>(a, dz);  // a = h3, dz = gelu'(z3) for now
    // ---------------- output layer: dW4 = sum dy h3, db4 = sum dy, dZ3 = w4 dy gelu'(z3); samples past N carry dy = 0,
    // which zeroes every contribution of theirs below
    f32x4 rT;   // 2^(e(n) + kscale) of the samples 4 g + r: what their H / dZ carry into the parameter gradients
    {
      f32x4 dyT = DOUBLE ? *reinterpret_cast<const f32x4*>(xb + rows4 * 16 + 4 * g) : dyT_early;  // samples 4 g + r
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const bool in = n0 + 4 * g + r < N;
        const int ex = (int)(__float_as_uint(dyT[r]) >> 23) & 255;
        int er = ex + kscale - CHAIN_EXP + H_PRESCALE_EXP;  // dZ of the chain = true dZ * 2^(CHAIN_EXP - e(n)); + the H pre-scale
        er = er < 1 ? 0 : (er > 254 ? 254 : er);            // below 2^-126 after scaling: the contribution is dropped
        rT[r] = (in && ex > CHAIN_EXP && ex != 255) ? __uint_as_float((uint32_t)er << 23) : ((in && ex == 255) ? 1.f : 0.f);
        dyT[r] = in ? dyT[r] * sc : 0.f;
      }
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
    // the chain of sample n runs on the mantissa of its dY (magnitude in [2^4, 2^5)); dX is multiplied by 2^(e(n) - 4) at the store
    float dy, dy_pow2;
    dy_parts(live ? (DOUBLE ? xb[rows4 * 16 + c] : dy_early) : 0.f, dy, dy_pow2);
    const float* wf = tail + 3 * HID;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wf + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; r++) dz[t][r] *= w4[r] * dy;
    }
    // ---------------- layer 3
    zero_init<NT>(a);
    layer_bwd<NT, NT>(dz, a, lds + OFF_T2, lane, id, h2T, dW3, db3, rT);  // a = dH2^T
#pragma unroll
    for (int t = 0; t < NT; t++) a[t] *= g2[t];                       // dZ2^T
    // ---------------- layer 2 (the prefetch goes out here: late enough that the early part of the tile does not wait on
    // it, early enough for an HBM round trip before the next tile)
    if (DOUBLE && tile + tstride < ntiles) prefetch(tile + tstride, stage + (cur ^ 1) * stage_floats);
    zero_init<NT>(dz);
    layer_bwd<NT, NT>(a, dz, lds + OFF_T1, lane, id, h1T, dW2, db2, rT);  // dz = dH1^T
#pragma unroll
    for (int t = 0; t < NT; t++) dz[t] *= g1[t];                      // dZ1^T
    // ---------------- layer 1: H = X in feature-lane order, straight from the staged rows
    f32x4 xT[NT0], dx[NT0];
#pragma unroll
    for (int u = 0; u < NT0; u++) {
      const int feat = 16 * u + c;
      xT[u] = zero4();
      if (feat < K0) {
        if (DOUBLE) {
          xT[u] = *reinterpret_cast<const f32x4*>(xb + feat * 16 + 4 * g);
        } else {   // samples n0 + 4 g + r of feature `feat` (clamped at the end of the batch: their dZ is zero)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            int64_t nn = n0 + 4 * g + r;
            nn = nn < N ? nn : N - 1;
            xT[u][r] = X[(int64_t)feat * N + nn];
          }
        }
      }
    }
    zero_init<NT0>(dx);
    layer_bwd<NT0, NT0>(dz, dx, lds + OFF_T0, lane, id, xT, dW1, db1, rT);  // dx = dX^T
    if (dX) {
#pragma unroll
      for (int t = 0; t < NT0; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 16 * t + 4 * g + r;
          if (k < K0 && live) dX[(int64_t)k * N + n] = dx[t][r] * dy_pow2;
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---------------- wave accumulators -> workgroup image (the weight images are dead) -> this workgroup's slot
  const int lane = lane_k, c = lane & 15, g = lane >> 4;
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
#if defined(PSDF_F16_PROTO_OCC)
          G[G_W2 + row + c] += dW2[to][0][r];
          G[G_W3 + row + c] += dW3[to][0][r];
          G[G_W1 + row + c] += dW1[to][0][r];
#else
#pragma unroll
          for (int ti = 0; ti < NT; ti++) {
            G[G_W2 + row + 16 * ti + c] += dW2[to][ti][r];
            G[G_W3 + row + 16 * ti + c] += dW3[to][ti][r];
          }
#pragma unroll
          for (int ti = 0; ti < NT0; ti++) G[G_W1 + row + 16 * ti + c] += dW1[to][ti][r];
#endif
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

// Sum of the workgroup images, accumulated into the torch-layout gradients (dW_l [out, in], db_l)
__global__ void mlp_split_reduce_kernel(const float* __restrict__ partial, const uint32_t* __restrict__ absmax, int nimg, int K0, float* __restrict__ dW0,
                                        float* __restrict__ dW1, float* __restrict__ dW2, float* __restrict__ dW3,
                                        float* __restrict__ db0, float* __restrict__ db1, float* __restrict__ db2,
                                        float* __restrict__ db3) {
  // blockIdx.y = a slice of the images (a serial loop over 256 images per element left the chip idle: 63 us); the slices
  // meet in the destination with one float atomic each (the destinations are accumulated into anyway)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= G_TOTAL) return;
  float s = 0.f;
  for (int b = blockIdx.y; b < nimg; b += gridDim.y) s += partial[(size_t)b * G_TOTAL + e];
  float sc, isc;
  dy_scale(absmax[0], sc, isc);
  s *= isc;                        // the images are gradients of dY * 2^k: exact power-of-two scaling
  if (e < G_W4) s *= __uint_as_float((uint32_t)(127 - H_PRESCALE_EXP) << 23);   // dW1..3, db1..3 carry the H pre-scale; dW4, db4 do not
  if (e < G_W2) {
    const int o = e >> 6, k = e & 63;
    if (k < K0) atomicAdd(&dW0[o * K0 + k], s);
  } else if (e < G_W3) {
    atomicAdd(&dW1[e - G_W2], s);
  } else if (e < G_B1) {
    atomicAdd(&dW2[e - G_W3], s);
  } else if (e < G_B2) {
    atomicAdd(&db0[e - G_B1], s);
  } else if (e < G_B3) {
    atomicAdd(&db1[e - G_B2], s);
  } else if (e < G_W4) {
    atomicAdd(&db2[e - G_B3], s);
  } else if (e < G_B4) {
    atomicAdd(&dW3[e - G_W4], s);
  } else {
    atomicAdd(&db3[0], s);
  }
}

// max over the batch of the bit pattern of |dY| (non-negative floats order like their bit patterns).  16-byte loads, many
// short threads: the first version (4-byte loads, 65 536 threads walking 32 elements each) took 17 us for 8 MB.
__global__ void __launch_bounds__(256) mlp_absmax_kernel(int64_t N, const float* __restrict__ dY, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  const int64_t n4 = (((uintptr_t)dY & 15) == 0) ? (N >> 2) : 0;     // an unaligned view takes the scalar loop below
  const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(dY);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = p4[i];
    const uint32_t a = v.x & 0x7FFFFFFFu, b = v.y & 0x7FFFFFFFu, c = v.z & 0x7FFFFFFFu, d = v.w & 0x7FFFFFFFu;
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
    m = q > m ? q : m;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t b = __float_as_uint(dY[i]) & 0x7FFFFFFFu;      // ragged tail (or everything, unaligned)
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)m, o, 64);
    m = t > m ? t : m;
  }
  // ONE atomic per workgroup: atomics to a single address serialise (8192 of them, one per wave, cost 80 us)
  __shared__ uint32_t wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    const uint32_t q = a > b ? a : b;
    if (q) atomicMax(out, q);
  }
}

// The LDS image from the torch-layout parameters: thread = (image 0..5, tile, k-step, lane) writes its two 16-byte
// records (one per piece); the tail threads copy biases / final weights; the last thread clears the |dY| maximum slot.
template <int NT0>
__global__ void mlp_split_pack_kernel(int K0, const float* __restrict__ W0, const float* __restrict__ W1,
                                      const float* __restrict__ W2, const float* __restrict__ W3,
                                      const float* __restrict__ b0, const float* __restrict__ b1,
                                      const float* __restrict__ b2, const float* __restrict__ b3, uint32_t* __restrict__ rec,
                                      uint32_t* __restrict__ absmax) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int PER_IMG = NT * 2 * 64, PER_T0 = NT0 * 2 * 64, NTHR = 5 * PER_IMG + PER_T0;
  if (t < NTHR) {
    const int im = t < 5 * PER_IMG ? t / PER_IMG : 5;
    const int q = t - im * PER_IMG;
    const int lane = q & 63, s = (q >> 6) & 1, tile = q >> 7;
    const int c = lane & 15, g = lane >> 4, row = 16 * tile + c;
    const int off[6] = {OFF_W0, OFF_W1, OFF_W2, OFF_T2, OFF_T1, OFF_T0};
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k0 = 32 * s + 8 * g + j, kc = kf(s, g, j);
      switch (im) {
        case 0: w[j] = k0 < K0 ? W0[row * K0 + k0] : 0.f; break;
        case 1: w[j] = W1[row * HID + kc]; break;
        case 2: w[j] = W2[row * HID + kc]; break;
        case 3: w[j] = W2[kc * HID + row]; break;                 // transposed images: row is an INPUT neuron of the layer
        case 4: w[j] = W1[kc * HID + row]; break;
        default: w[j] = row < K0 ? W0[kc * K0 + row] : 0.f; break;
      }
    }
    u32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t h, l;
      split2(w[2 * i], w[2 * i + 1], h, l);
      hi[i] = h;
      lo[i] = l;
    }
    u32x4* dst = reinterpret_cast<u32x4*>(rec) + (off[im] + ((tile * 2 + s) * NP) * 64 + lane);
    dst[0] = hi;
    dst[64] = lo;
  } else {
    const int e = t - NTHR;
    float* tail = reinterpret_cast<float*>(rec + (size_t)off_f32(NT0) * 4);
    if (e < HID) tail[e] = b0[e];
    else if (e < 2 * HID) tail[e] = b1[e - HID];
    else if (e < 3 * HID) tail[e] = b2[e - 2 * HID];
    else if (e < 4 * HID) tail[e] = W3[e - 3 * HID];
    else if (e == 4 * HID) tail[e] = b3[0];
    else if (e == 4 * HID + 1) absmax[0] = 0u;
  }
}

}  // namespace

extern "C" {

// Same contract as psdf_mlp_backward (include/psdf.h) for dims = {K0 <= 64, 64, 64, 64, 1} with dW / db requested; returns
// PSDF_ERR_UNSUPPORTED (-2) for everything else and when the library's per-stream scratch is unavailable (stream capture).
int psdf_mlp_backward_split_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                                const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                                void* stream) {
  if (n_layers != 4 || !dims || dims[1] != HID || dims[2] != HID || dims[3] != HID || dims[4] != 1 || !dW || !db)
    return PSDF_ERR_UNSUPPORTED;
  const int K0 = dims[0];
  if (K0 < 1 || K0 > 64) return PSDF_ERR_UNSUPPORTED;
  const int rows4 = (K0 + 3) & ~3;
  const int nt0 = K0 <= 48 ? 3 : 4;
  const size_t stage_bytes = (size_t)NWAVES * (rows4 * 16 + 64) * 4;
  const size_t img_bytes = img_aligned(nt0);
  const size_t lds_bytes = img_bytes + 2 * stage_bytes;          // always double buffered: the image is 95 KB
  if (lds_bytes > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  if (N <= 0 || !X || !weights || !biases || !dY) return PSDF_ERR_ARG;
  for (int l = 0; l < 4; l++)
    if (!weights[l] || !biases[l] || !dW[l] || !db[l]) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = (N + 15) / 16;
  int64_t blocks = (ntiles + NWAVES - 1) / NWAVES;
  if (blocks > 256) blocks = 256;  // one workgroup per CU; each wave walks many tiles
  const size_t part_bytes = (size_t)blocks * G_TOTAL * sizeof(float);
  char* scratch = (char*)psdf::stream_scratch(img_bytes + 16 + part_bytes, st);   // NULL while capturing
  if (!scratch) return PSDF_ERR_UNSUPPORTED;
  uint32_t* rec = reinterpret_cast<uint32_t*>(scratch);
  uint32_t* absmax = reinterpret_cast<uint32_t*>(scratch + img_bytes);
  float* partial = reinterpret_cast<float*>(scratch + img_bytes + 16);
  const int pack_threads = (5 * NT + nt0) * 2 * 64 + TAIL_FLOATS + 1;
#define PACK(NT0_)                                                                                                         \
  hipLaunchKernelGGL(mlp_split_pack_kernel<NT0_>, dim3((pack_threads + 255) / 256), dim3(256), 0, st, K0, weights[0],        \
                     weights[1], weights[2], weights[3], biases[0], biases[1], biases[2], biases[3], rec, absmax)
#define MAIN(NT0_)                                                                                                          \
  do {                                                                                                                      \
    auto kern = mlp_bwd_split_f16_kernel<NT0_, true>;                                                                        \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);       \
    if (e != hipSuccess) return (int)e;                                                                                     \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NWAVES * 64), lds_bytes, st, N, K0, rows4, X, dY,                  \
                       reinterpret_cast<const u32x4*>(rec), absmax, dX, partial);                                           \
  } while (0)
  if (nt0 == 3) PACK(3); else PACK(4);
  {
    int64_t ab = ((N >> 2) + 1023) / 1024;     // four 16-byte loads per thread, at most 512 workgroups (= 512 atomics)
    ab = ab < 1 ? 1 : (ab > 512 ? 512 : ab);
    hipLaunchKernelGGL(mlp_absmax_kernel, dim3((unsigned)ab), dim3(256), 0, st, N, dY, absmax);
  }
  if (nt0 == 3) MAIN(3); else MAIN(4);
#undef PACK
#undef MAIN
  hipLaunchKernelGGL(mlp_split_reduce_kernel, dim3((G_TOTAL + 255) / 256, 16), dim3(256), 0, st, partial, absmax, (int)blocks, K0,
                     dW[0], dW[1], dW[2], dW[3], db[0], db[1], db[2], db[3]);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // extern "C"
