// Backward of the 64x3 -> 1 SDF net (K0 <= 36 inputs: the BASELINE net on a 16-level encoding) on the bf16 MATRIX PIPE with
// fp32 accuracy: every fp32 operand is cut into three bf16 pieces and six of the nine piece products are kept (error
// 2^-24-ish per product, the level of an fp32 evaluation; same scheme as the forward, csrc/mlp.hip).  Why: fp32 MFMAs and
// VALU work do not overlap on gfx950 (profiles/r01_mfma_valu_overlap.txt), and the fp32 kernel (mlp_bwd.hip) spends 49 %
// of its time in them.  Structure (attic/prototypes/mlp_bwd_split_bf16_v3.hip is the standalone prototype with its history):
//   * 16-sample tiles on v_mfma_f32_16x16x32_bf16, one wave per SIMD, dW accumulators persistent in registers (176);
//   * forward recomputed from X; gelu and gelu' from one exponential and one reciprocal (gelu_rational below);
//   * the sample<->feature transposes that the dW products need are MFMAs against a 0/1 operand (no LDS, no VALU);
//   * a dW MFMA (K = samples, only 16 of 32 slots filled by a tile) carries two piece products in its two K halves
//     (480 instead of 612 MFMAs per tile);
//   * both weight orientations as pre-split pieces in LDS (142 KB), built per call by mlp_split_pack_kernel;
//   * the next tile's inputs arrive by LDS-DMA (global_load_lds, 16-byte form) while the current tile is computed (SQ
//     counters of the version without it: 39 % of the wave's time in s_waitcnt), and loop-invariant lane arithmetic is
//     re-materialised per tile because a scratch reload would wait (vmcnt) for the DMA in flight;
//   * one gradient image per workgroup, summed by a second launch into the torch-layout dW / db.
// Measured (2 M samples, 36-64-64-64-1): 1.20 ms against 1.83 ms for the fp32 kernel (history: DESIGN.md section 9,
// profiles/r02_mlp_bwd_prototype_timings.txt); gradients within 1e-6 relative of a float64 evaluation.  Built with
// -mllvm -amdgpu-mfma-vgpr-form=1 (only the accumulators live in AGPRs); the K0 <= 36 instantiation in its own translation
// unit (mlp_bwd_split_double.hip).
#include "psdf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 64, NT = 4 /* 16-feature tiles of a hidden layer */;   // NT0 (template) = tiles covering the input: 3 (<= 48) or 4 (<= 64)
__host__ __device__ inline int kf(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

// ------------------------------------------------------------------ LDS image (units: 16-byte lane records)
// every layer: [tile][k-step 2][piece 3][lane 64]
constexpr int RECL = NT * 2 * 3 * 64;
constexpr int OFF_W0 = 0, OFF_W1 = RECL, OFF_W2 = 2 * RECL, OFF_T2 = 3 * RECL, OFF_T1 = 4 * RECL, OFF_T0 = 5 * RECL;
constexpr int off_f32(int nt0) { return 5 * RECL + nt0 * 2 * 3 * 64; }
constexpr int TAIL_FLOATS = 3 * HID + HID + 1;  // biases of the three hidden layers, final weights, final bias
constexpr int NWAVES = 4;
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

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct BP {  // the three bf16 pieces of one 8-element operand
  bf16x8 p[3];
};
__device__ __forceinline__ uint32_t top_pair(float hi, float lo) {  // {top half of hi, top half of lo}
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// eight fp32 -> three bf16x8 pieces by truncation of the running remainder.  The remainders are formed on PAIRS
// (v_pk_add_f32: one issue slot for two subtractions -- with one wave per SIMD a packed instruction costs what a plain one does)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 trunc_pair(f32x2 v) {
  return f32x2{__uint_as_float(__float_as_uint(v.x) & 0xFFFF0000u), __uint_as_float(__float_as_uint(v.y) & 0xFFFF0000u)};
}
__device__ __forceinline__ void split8(const float (&x)[8], BP& o) {
  u32x4 q0, q1, q2;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    const f32x2 r1 = v - trunc_pair(v);
    const f32x2 r2 = r1 - trunc_pair(r1);
    q0[i] = top_pair(v.y, v.x);
    q1[i] = top_pair(r1.y, r1.x);
    q2[i] = top_pair(r2.y, r2.x);
  }
  o.p[0] = __builtin_bit_cast(bf16x8, q0);
  o.p[1] = __builtin_bit_cast(bf16x8, q1);
  o.p[2] = __builtin_bit_cast(bf16x8, q2);
}
// dW operands.  A transposed tile gives a lane only four samples (k-slots (g, 0..3) of the 32-deep MFMA); instead of
// leaving slots (g, 4..7) zero they carry ANOTHER PIECE of the same four samples, so that one MFMA sums two of the six
// piece products:  [a0|a2] x [b2|b0] = a0 b2 + a2 b0,  [a1|a1] x [b0|b1] = a1 b0 + a1 b1,  [a0|a0] x [b0|b1] = a0 b0 + a0 b1.
// Three MFMAs per 16x16 block of dW instead of six (132 instead of 264 per tile).
struct AT {  // dZ side
  bf16x8 t02, t11, t00;
};
struct BT {  // H side
  bf16x8 t20, t01;
};
__device__ __forceinline__ bf16x8 halves(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
  const u32x4 q = {a0, a1, b0, b1};
  return __builtin_bit_cast(bf16x8, q);
}
// four fp32 (a feature-lane tile: samples 4 g + r) -> H-side operands
__device__ __forceinline__ void split4(const f32x4& t, BT& o) {
  const f32x2 va = {t[0], t[1]}, vb = {t[2], t[3]};
  const f32x2 r1a = va - trunc_pair(va), r1b = vb - trunc_pair(vb);
  const f32x2 r2a = r1a - trunc_pair(r1a), r2b = r1b - trunc_pair(r1b);
  const uint32_t p0a = top_pair(va.y, va.x), p0b = top_pair(vb.y, vb.x);
  const uint32_t p1a = top_pair(r1a.y, r1a.x), p1b = top_pair(r1b.y, r1b.x);
  const uint32_t p2a = top_pair(r2a.y, r2a.x), p2b = top_pair(r2b.y, r2b.x);
  o.t20 = halves(p2a, p2b, p0a, p0b);
  o.t01 = halves(p0a, p0b, p1a, p1b);
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
// piece-wise transpose: dZ-side dW operands of a 16-feature tile, plus this lane's fp32 sum for the bias gradient
__device__ __forceinline__ void transpose_pieces(const BP& b, bf16x8 id, AT& out, float& sum) {
  uint32_t q[3][2];
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const f32x4 o = MFMA16(b.p[p], id, zero4());   // bf16-valued: the top halves are the piece
    sum += (o[0] + o[1]) + (o[2] + o[3]);
    q[p][0] = top_pair(o[1], o[0]);
    q[p][1] = top_pair(o[3], o[2]);
  }
  out.t02 = halves(q[0][0], q[0][1], q[2][0], q[2][1]);
  out.t11 = halves(q[1][0], q[1][1], q[1][0], q[1][1]);
  out.t00 = halves(q[0][0], q[0][1], q[0][0], q[0][1]);
}
__device__ __forceinline__ f32x4 dw_mac(f32x4 acc, const AT& A, const BT& B) {
  acc = MFMA16(A.t02, B.t20, acc);   // smallest first
  acc = MFMA16(A.t11, B.t01, acc);
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
// in place: acc <- gelu(acc), gp <- gelu'(acc)
template <bool RATIONAL>
__device__ __forceinline__ void act_both(f32x4 (&acc)[NT], f32x4 (&gp)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; t++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float hv, d;
      gelu_both<RATIONAL>(acc[t][r], hv, d);
      acc[t][r] = hv;
      gp[t][r] = d;
    }
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
// backward of one layer: dH chain (hands the pieces of dZ to the transposes), then dW[to][ti] += dZ(to) x H(ti)
template <int NTO, int NTI>
__device__ __forceinline__ void layer_bwd(const f32x4 (&dz)[NT], f32x4 (&dh)[NTO], const u32x4* __restrict__ wT, int lane,
                                          const bf16x8 (&id)[2], const f32x4 (&hT)[NTI], f32x4 (&dW)[NT][NTI], float (&db)[NT]) {
  AT A[NT];
  chain<NTO>(dz, dh, wT, lane, [&](int s, const BP& b) {
    transpose_pieces(b, id[0], A[2 * s], db[2 * s]);
    transpose_pieces(b, id[1], A[2 * s + 1], db[2 * s + 1]);
  });
#pragma unroll
  for (int ti = 0; ti < NTI; ti++) {
    BT B;
    split4(hT[ti], B);
#pragma unroll
    for (int to = 0; to < NT; to++) dW[to][ti] = dw_mac(dW[to][ti], A[to], B);
  }
}

// X [K0, N], dY [1, N], dX [K0, N] (optional) feature-major; img = the LDS image (mlp_split_pack_kernel); partial
// [gridDim.x][G_TOTAL] receives this workgroup's gradient image.  rows4 = K0 rounded up to a multiple of 4.
// DOUBLE: the staged inputs are double buffered and serve the whole tile (K0 <= 36: it fits beside the image in 160 KB of
// LDS).  Otherwise ONE staging buffer per wave: it is read at the top of the tile and refilled at once for the next tile;
// the feature-lane copy of X that the last layer's dW needs comes from global memory (an L2 hit: the tile was just staged).
template <int NT0, bool DOUBLE>
__global__ void __launch_bounds__(NWAVES * 64, 1)
    mlp_bwd_split_kernel(int64_t N, int K0, int rows4, const float* __restrict__ X, const float* __restrict__ dY,
                         const u32x4* __restrict__ img, float* __restrict__ dX, float* __restrict__ partial,
                         const uint32_t* __restrict__ only_if) {
  // only_if: NULL, or a device word that must be non-zero for this launch to do anything -- the split-fp16 kernels queue
  // this one behind themselves for the batches that leave their range (mlp_bwd_split_f16.hip, "range guard")
  if (only_if && only_if[0] == 0u) return;
  extern __shared__ __align__(16) u32x4 lds[];
  constexpr size_t IMG_ALIGNED = img_aligned(NT0);
  constexpr int OFF_F32 = off_f32(NT0);
  constexpr int NREC = (int)(IMG_ALIGNED / 16);
  for (int i = threadIdx.x; i < NREC; i += NWAVES * 64) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  const int lane_k = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16x8 id[2] = {ident_op(0, lane_k), ident_op(1, lane_k)};
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
        mac16<NT>(a, bx, lds + OFF_W0 + s * 192 + lane);
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
    act_both<NT0 == 3>(a, g1);  // a = h1
    bias_init<NT>(b, tail + HID, g);
    chain<NT>(a, b, lds + OFF_W1, lane, [&](int s, const BP& p) {
      h1T[2 * s] = transpose_f32(p, id[0]);
      h1T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    act_both<NT0 == 3>(b, g2);  // b = h2
    bias_init<NT>(a, tail + 2 * HID, g);
    chain<NT>(b, a, lds + OFF_W2, lane, [&](int s, const BP& p) {
      h2T[2 * s] = transpose_f32(p, id[0]);
      h2T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    f32x4 dz[NT];
    act_both<NT0 == 3>(a, dz);  // a = h3, dz = gelu'(z3) for now
    // ---------------- output layer: dW4 = sum dy h3, db4 = sum dy, dZ3 = w4 dy gelu'(z3); samples past N carry dy = 0,
    // which zeroes every contribution of theirs below
    {
      f32x4 dyT = DOUBLE ? *reinterpret_cast<const f32x4*>(xb + rows4 * 16 + 4 * g) : dyT_early;  // samples 4 g + r
#pragma unroll
      for (int r = 0; r < 4; r++) dyT[r] = (n0 + 4 * g + r < N) ? dyT[r] : 0.f;
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
    const float dy = live ? (DOUBLE ? xb[rows4 * 16 + c] : dy_early) : 0.f;
    const float* wf = tail + 3 * HID;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wf + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; r++) dz[t][r] *= w4[r] * dy;
    }
    // ---------------- layer 3
    zero_init<NT>(a);
    layer_bwd<NT, NT>(dz, a, lds + OFF_T2, lane, id, h2T, dW3, db3);  // a = dH2^T
#pragma unroll
    for (int t = 0; t < NT; t++) a[t] *= g2[t];                       // dZ2^T
    // ---------------- layer 2 (the prefetch goes out here: late enough that the early part of the tile does not wait on
    // it, early enough for an HBM round trip before the next tile)
    if (DOUBLE && tile + tstride < ntiles) prefetch(tile + tstride, stage + (cur ^ 1) * stage_floats);
    zero_init<NT>(dz);
    layer_bwd<NT, NT>(a, dz, lds + OFF_T1, lane, id, h1T, dW2, db2);  // dz = dH1^T
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
    layer_bwd<NT0, NT0>(dz, dx, lds + OFF_T0, lane, id, xT, dW1, db1);  // dx = dX^T
    if (dX) {
#pragma unroll
      for (int t = 0; t < NT0; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 16 * t + 4 * g + r;
          if (k < K0 && live) dX[(int64_t)k * N + n] = dx[t][r];
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

// Sum of the workgroup images, accumulated into the torch-layout gradients (dW_l [out, in], db_l)
__global__ void mlp_split_reduce_kernel(const float* __restrict__ partial, int nimg, int K0, float* __restrict__ dW0,
                                        float* __restrict__ dW1, float* __restrict__ dW2, float* __restrict__ dW3,
                                        float* __restrict__ db0, float* __restrict__ db1, float* __restrict__ db2,
                                        float* __restrict__ db3, const uint32_t* __restrict__ only_if) {
  if (only_if && only_if[0] == 0u) return;
  // blockIdx.y = a slice of the images (a serial loop over 256 images per element left the chip idle: 63 us); the slices
  // meet in the destination with one float atomic each (the destinations are accumulated into anyway)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= G_TOTAL) return;
  float s = 0.f;
  for (int b = blockIdx.y; b < nimg; b += gridDim.y) s += partial[(size_t)b * G_TOTAL + e];
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

// fp32 -> three bf16 pieces by truncation of the running remainder (the pieces sum to the value exactly)
__device__ __forceinline__ void split3(float x, uint16_t (&p)[3]) {
  float r = x;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const uint32_t u = __float_as_uint(r);
    p[i] = (uint16_t)(u >> 16);
    r -= __uint_as_float(u & 0xFFFF0000u);
  }
}

// The LDS image from the torch-layout parameters: thread = (image 0..5, tile, k-step, lane) writes its three 16-byte
// records (one per piece); the tail threads copy biases / final weights.
template <int NT0>
__global__ void mlp_split_pack_kernel(int K0, const float* __restrict__ W0, const float* __restrict__ W1,
                                      const float* __restrict__ W2, const float* __restrict__ W3,
                                      const float* __restrict__ b0, const float* __restrict__ b1,
                                      const float* __restrict__ b2, const float* __restrict__ b3, uint16_t* __restrict__ rec,
                                      const uint32_t* __restrict__ only_if) {
  if (only_if && only_if[0] == 0u) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int PER_IMG = NT * 2 * 64, PER_T0 = NT0 * 2 * 64, NTHR = 5 * PER_IMG + PER_T0;
  if (t < NTHR) {
    const int im = t < 5 * PER_IMG ? t / PER_IMG : 5;
    const int q = t - im * PER_IMG;
    const int lane = q & 63, s = (q >> 6) & 1, tile = q >> 7;
    const int c = lane & 15, g = lane >> 4, row = 16 * tile + c;
    const int off[6] = {OFF_W0, OFF_W1, OFF_W2, OFF_T2, OFF_T1, OFF_T0};
    uint16_t out[3][8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k0 = 32 * s + 8 * g + j, kc = kf(s, g, j);
      float w;
      switch (im) {
        case 0: w = k0 < K0 ? W0[row * K0 + k0] : 0.f; break;
        case 1: w = W1[row * HID + kc]; break;
        case 2: w = W2[row * HID + kc]; break;
        case 3: w = W2[kc * HID + row]; break;                 // transposed images: row is an INPUT neuron of the layer
        case 4: w = W1[kc * HID + row]; break;
        default: w = row < K0 ? W0[kc * K0 + row] : 0.f; break;
      }
      uint16_t p[3];
      split3(w, p);
      out[0][j] = p[0]; out[1][j] = p[1]; out[2][j] = p[2];
    }
#pragma unroll
    for (int piece = 0; piece < 3; piece++) {
      uint16_t* dst = rec + ((size_t)(off[im] + ((tile * 2 + s) * 3 + piece) * 64 + lane)) * 8;
#pragma unroll
      for (int j = 0; j < 8; j++) dst[j] = out[piece][j];
    }
  } else {
    const int e = t - NTHR;
    float* tail = reinterpret_cast<float*>(rec + (size_t)off_f32(NT0) * 8);
    if (e < HID) tail[e] = b0[e];
    else if (e < 2 * HID) tail[e] = b1[e - HID];
    else if (e < 3 * HID) tail[e] = b2[e - 2 * HID];
    else if (e < 4 * HID) tail[e] = W3[e - 3 * HID];
    else if (e == 4 * HID) tail[e] = b3[0];
  }
}

}  // namespace

// The double-staged instantiation (K0 <= 36: the BASELINE net) is compiled in its own translation unit,
// mlp_bwd_split_double.hip, with -mllvm -amdgpu-sched-strategy=max-ilp: 1.7 % faster there (1.217 -> 1.195 ms), while the
// single-staged instantiations spill under that strategy (108 / 64 B of scratch) and stay with the default one.
namespace psdf {
int mlp_bwd_split_launch_double(unsigned blocks, size_t lds_bytes, hipStream_t st, int64_t N, int K0, int rows4, const float* X,
                                const float* dY, const void* rec, float* dX, float* partial, const uint32_t* only_if);
// psdf_mlp_backward_split with the caller's scratch (mlp_backward_split_scratch_bytes of it, 16-byte aligned; NULL = the
// library's per-stream scratch) and an optional device-side condition (see mlp_bwd_split_kernel)
size_t mlp_backward_split_scratch_bytes(int K0, int64_t N);
int mlp_backward_split_impl(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                            const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                            hipStream_t st, char* scratch, const uint32_t* only_if);
}
#if defined(PSDF_SPLIT_TU_DOUBLE)
int psdf::mlp_bwd_split_launch_double(unsigned blocks, size_t lds_bytes, hipStream_t st, int64_t N, int K0, int rows4,
                                      const float* X, const float* dY, const void* rec, float* dX, float* partial,
                                      const uint32_t* only_if) {
  auto kern = mlp_bwd_split_kernel<3, true>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), lds_bytes, st, N, K0, rows4, X, dY,
                     reinterpret_cast<const u32x4*>(rec), dX, partial, only_if);
  return PSDF_OK;
}
#else

extern "C" {

// Same contract as psdf_mlp_backward (include/psdf.h) for the nets this kernel covers: dims = {K0 <= 52, 64, 64, 64, 1} (what fits 160 KB of LDS),
// dW / db requested; returns PSDF_ERR_UNSUPPORTED (-2) for everything else (the caller then takes the fp32 kernel).
// Needs the library's per-stream scratch (psdf::stream_scratch: the 142-KB operand image and one gradient image per workgroup); when that
// is not available (stream capture, allocation failure) it also returns -2.
int psdf_mlp_backward_split(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                            const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                            void* stream) {
  return psdf::mlp_backward_split_impl(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, (hipStream_t)stream, nullptr, nullptr);
}

}  // extern "C"

static int64_t split_blocks(int64_t N) {
  const int64_t ntiles = (N + 15) / 16;
  int64_t blocks = (ntiles + NWAVES - 1) / NWAVES;
  return blocks > 256 ? 256 : blocks;  // one workgroup per CU; each wave walks many tiles
}
size_t psdf::mlp_backward_split_scratch_bytes(int K0, int64_t N) {
  return img_aligned(K0 <= 48 ? 3 : 4) + (size_t)split_blocks(N) * G_TOTAL * sizeof(float);
}
int psdf::mlp_backward_split_impl(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                                  const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                                  hipStream_t st, char* scratch, const uint32_t* only_if) {
  if (n_layers != 4 || !dims || dims[1] != HID || dims[2] != HID || dims[3] != HID || dims[4] != 1 || !dW || !db)
    return PSDF_ERR_UNSUPPORTED;
  const int K0 = dims[0];
  if (K0 < 1 || K0 > 64) return PSDF_ERR_UNSUPPORTED;
  const int rows4 = (K0 + 3) & ~3;
  const int nt0 = K0 <= 48 ? 3 : 4;
  const size_t stage_bytes = (size_t)NWAVES * (rows4 * 16 + 64) * 4;
  const bool dbl = nt0 == 3 && img_aligned(3) + 2 * stage_bytes <= 160 * 1024;      // K0 <= 36
  const size_t img_bytes = img_aligned(nt0);
  const size_t lds_bytes = img_bytes + (dbl ? 2 : 1) * stage_bytes;
  if (lds_bytes > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  if (N <= 0 || !X || !weights || !biases || !dY) return PSDF_ERR_ARG;
  for (int l = 0; l < 4; l++)
    if (!weights[l] || !biases[l] || !dW[l] || !db[l]) return PSDF_ERR_ARG;
  const int64_t blocks = split_blocks(N);
  const size_t part_bytes = (size_t)blocks * G_TOTAL * sizeof(float);
  if (!scratch) scratch = (char*)psdf::stream_scratch(img_bytes + part_bytes, st);   // NULL while capturing
  if (!scratch) return PSDF_ERR_UNSUPPORTED;
  uint16_t* rec = reinterpret_cast<uint16_t*>(scratch);
  float* partial = reinterpret_cast<float*>(scratch + img_bytes);
  const int pack_threads = (5 * NT + nt0) * 2 * 64 + TAIL_FLOATS;
#define PACK(NT0_)                                                                                                         \
  hipLaunchKernelGGL(mlp_split_pack_kernel<NT0_>, dim3((pack_threads + 255) / 256), dim3(256), 0, st, K0, weights[0],        \
                     weights[1], weights[2], weights[3], biases[0], biases[1], biases[2], biases[3], rec, only_if)
#define MAIN(NT0_, DBL_)                                                                                                    \
  do {                                                                                                                      \
    auto kern = mlp_bwd_split_kernel<NT0_, DBL_>;                                                                            \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);       \
    if (e != hipSuccess) return (int)e;                                                                                     \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NWAVES * 64), lds_bytes, st, N, K0, rows4, X, dY,                  \
                       reinterpret_cast<const u32x4*>(rec), dX, partial, only_if);                                          \
  } while (0)
  if (nt0 == 3) {
    PACK(3);
    if (dbl) {
      const int rc = psdf::mlp_bwd_split_launch_double((unsigned)blocks, lds_bytes, st, N, K0, rows4, X, dY, rec, dX, partial, only_if);
      if (rc != PSDF_OK) return rc;
    } else {
      MAIN(3, false);
    }
  } else {
    PACK(4);
    MAIN(4, false);
  }
#undef PACK
#undef MAIN
  hipLaunchKernelGGL(mlp_split_reduce_kernel, dim3((G_TOTAL + 255) / 256, 16), dim3(256), 0, st, partial, (int)blocks, K0, dW[0],
                     dW[1], dW[2], dW[3], db[0], db[1], db[2], db[3], only_if);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}
#endif  // PSDF_SPLIT_TU_DOUBLE
