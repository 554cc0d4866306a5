// Backward of the 64x3 -> 1 SDF net on the fp16 MATRIX PIPE with TWO pieces per fp32 operand (round 3; the sibling of
// mlp_bwd_split.hip, which uses three bf16 pieces and six products).  a = a0 + a1, a0 = fp16(a) rounded to nearest (a - a0
// is exact in fp32), a1 = fp16(a - a0): 11 + 1 (sign of a1) + 11 mantissa bits; chains keep a0 b0 + a0 b1 + a1 b0 (error ~2^-22 |a b|),
// the dW products all four (the fourth rides in an otherwise empty K half).  Against the bf16 scheme: 274 instead of 480 MFMAs
// per 16-sample tile, operand splitting 5 instead of 9 VALU instructions per pair, a 95-KB instead of a 142-KB weight image
// (so every input width up to 64 gets double-buffered staging).  What makes it legitimate on gfx950 (measured with
// attic/prototypes/mlp_fwd_split_f16.hip): the matrix pipe HONOURS fp16 subnormal inputs, so a low piece below 2^-14 keeps an absolute
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
// Round 4, second half (1.08 -> 0.92 ms on 2 M samples, tools/r04_mlp_sched_ab.sh / profiles/r04_mlp_sched_ab.txt): the kernel
// is bound by what ONE wave can issue (2 670 instructions per tile at ~7 cycles each; the matrix pipe is busy a quarter of the
// time and hardly ever beside the VALU), so the round went into the instruction stream:
//   * the 176 dW accumulators are PINNED in accumulation registers (an empty asm with an "a" constraint on either side of
//     their MFMA pair: the compiler then emits the AGPR form itself and still sees the MFMAs for its hazard bookkeeping) -- left
//     alone, a changing subset was parked in AGPRs and copied to VGPRs and back around every use (100 - 230 v_accvgpr moves
//     per tile, depending on the build);
//   * an operand is split with v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (3 instead of 5 instructions per pair, see split2);
//   * weight records are requested one k-step ahead of the MFMAs that need them, biases and final weights ahead of the
//     activation blocks (38 -> 15 s_waitcnt per tile);
//   * two GELU pairs are evaluated side by side (a dependent v_pk_fma_f32 needs a wait state: 117 -> 32 s_nop per tile);
//   * the staging buffer keeps 64 zero-padded rows, so the layer-0 operand is read without predicates (-110 instructions);
//   * dX is stored through one lane base + uniform row offsets (-100), bias sums run as register pairs (-110, 86 of them moves);
//   * gelu' of the inner layers waits in LDS instead of in 32 registers, the dZ-side dW operand takes 4 registers instead of 8.
// 2 000 instructions per tile now.  Measured and not kept: requesting weights without the sched_barrier pins (the compiler sinks
// them back), a software-pipelined H-side split with sched_group_barrier patterns (the pattern solver rearranged the rest of the
// region: slower), other scheduler strategies (max-ilp, iterative-*: more spills or longer).
// Accuracy against float64: tests/test_gpu_mlp.py::test_split_f16_backward_matches_float64.  Built with
// -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize (build.py).
#include "psdf_common.h"
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>

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
constexpr int NWAVES = 4;
constexpr size_t img_aligned(int nt0) { return ((size_t)off_f32(nt0) * 16 + TAIL_FLOATS * 4 + 15) / 16 * 16; }
// gradient image (floats): dW1 [64][64 (K0 used)], dW2 [64][64], dW3 [64][64], db1, db2, db3 [64], dW4 [64], db4
constexpr int G_W1 = 0, G_W2 = 4096, G_W3 = 8192, G_B1 = 12288, G_B2 = 12352, G_B3 = 12416, G_W4 = 12480, G_B4 = 12544,
              G_TOTAL = 12545;

// gelu AND gelu' = Phi(z) + z phi(z) from ONE exponential and ONE reciprocal (tools/gelu_fit_rational.py; the recompute needs
// both):  E = exp(-z^2/2), t = 1/(1 + p|z|), Phi(-|z|) = t P6(t) E, cdf = z < 0 ? Phi(-|z|) : 1 - Phi(-|z|),
//   gelu = z cdf, gelu' = cdf + z E / sqrt(2 pi).  Error against float64: gelu 1.8e-7 |z| (the fp32 formula
//   0.5 z (1 + erf(z / sqrt 2)) itself: 1.1e-7 |z|), gelu' 1.9e-7.  (An erf-based and a pure-polynomial evaluator were measured
//   beside it in round 2 -- 1.47 ms against 1.37 ms, profiles/r02_mlp_bwd_prototype_timings.txt -- and are gone.)
// Written for TWO PAIRS at a time in packed fp32 arithmetic: gelu_rational4 below.  (The forward kernels evaluate the same
// fit as max(z, 0) - |z| Phi(-|z|), mlp_device.h: equal up to the last bit or two of an fp32 evaluation.)

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

struct BP {  // the two fp16 pieces of one 8-element operand: p[0] = high, p[1] = low
  f16x8 p[NP];
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> {high pieces, low pieces}, each a packed pair (element 0 in the low half).  The high piece is fp16(x) rounded to
// NEAREST (v_cvt_pk_f16_f32, new on gfx950; rounds 3-5 truncated with v_cvt_pkrtz): |x - high| <= 2^-11 of x's binade and a
// multiple of its fp32 ulp, so the remainder is exact in fp32 and has at most 12 significant bits; the low piece keeps 11 of
// them: x - (high + low) is 0 for three operands in four and 2^-23 of the binade otherwise (truncation left up to 2^-22 and a
// ONE-SIDED error that grows with the depth of a dot product instead of its square root), and |low| <= 2^-11 |x| makes the
// dropped low x low product <= 2^-22 of the largest and ~2^-25 of a typical product.  Same three instructions per pair.
// `one` is 1.0f the optimiser cannot see through (an empty asm on a scalar register): fma(x, one, -high) must reach instruction
// selection as an fma with an fp16 source and an fp16 result, which is v_fma_mixlo_f16 / v_fma_mixhi_f16 -- x * 1 - high formed
// exactly and rounded once to fp16 (nearest even; subnormal results kept: the kernel runs with fp16 denormals on) -- ONE
// instruction per element instead of two conversions back, a packed subtraction and a second packed conversion.  Needs
// -fno-slp-vectorize (the SLP vectoriser packs the two fmas into v_pk_fma_f32 + conversions otherwise).  NOT inline asm: the
// hazard recogniser must see the VALU write -- an MFMA that reads a register needs two wait states after a VALU wrote it (the
// compiler puts an s_nop 1 there), and a build with the instruction in an asm statement computed garbage whenever the
// scheduler happened to place one directly in front of an MFMA.
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  float one = 1.0f;
  asm("" : "+s"(one));
  const h2_t h2 = __builtin_convertvector(f32x2{x0, x1}, h2_t);   // v_cvt_pk_f16_f32 (round to nearest even)
  h2_t l;
  l[0] = (_Float16)__builtin_fmaf(x0, one, -(float)h2[0]);
  l[1] = (_Float16)__builtin_fmaf(x1, one, -(float)h2[1]);
  hi = __builtin_bit_cast(uint32_t, h2);
  lo = __builtin_bit_cast(uint32_t, l);
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
//   [a0|a1] x [b1|b1] = a0 b1 + a1 b1,   [a0|a1] x [b0|b0] = a0 b0 + a1 b0   (the fourth product is free: its K half was empty)
// Two MFMAs per 16x16 block of dW (88 per tile; the bf16 scheme: 132).  The duplicated halves sit on the H side: its operand
// lives for one column of blocks, the four dZ operands for the whole layer (4 registers each instead of 8).
struct AT {  // dZ side
  f16x8 t01;
};
struct BT {  // H side
  f16x8 t00, t11;
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
  o.t00 = halves(ha, hb, ha, hb);
  o.t11 = halves(la, lb, la, lb);
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
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 sp2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 lo2(const f32x4& v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(const f32x4& v) { return __builtin_shufflevector(v, v, 2, 3); }
// elementwise products of register quadruples as TWO packed multiplies (the file is built without the SLP vectoriser, which is
// what would otherwise pair the four scalar multiplies of `a * b`): 117 -> ~60 multiply instructions per tile
__device__ __forceinline__ f32x4 mul4(const f32x4& a, const f32x4& b) {
  const f32x2 l = lo2(a) * lo2(b), h = hi2(a) * hi2(b);
  return f32x4{l.x, l.y, h.x, h.y};
}
__device__ __forceinline__ f32x4 mul4s(const f32x4& a, float s) { return mul4(a, f32x4{s, s, s, s}); }
// running sum of products over the samples of a lane.  PAIR: two partial sums (samples 4 g + {0, 1} and 4 g + {2, 3}: register
// pairs as an MFMA leaves them, so the packed multiply-adds need no moves -- the scalar form cost 86 v_mov per tile), added in the
// epilogue; the widest instantiation (K0 > 48) has no registers for the second half and keeps one sum
template <bool PAIR> struct Sum;
template <> struct Sum<true> {
  f32x2 v;
  __device__ __forceinline__ void clear() { v = f32x2{0.f, 0.f}; }
  __device__ __forceinline__ void add(const f32x4& o, const f32x4& w) {
    v = pk_fma2(lo2(o), lo2(w), v);
    v = pk_fma2(hi2(o), hi2(w), v);
  }
  __device__ __forceinline__ float total() const { return v.x + v.y; }
};
template <> struct Sum<false> {
  float v;
  __device__ __forceinline__ void clear() { v = 0.f; }
  __device__ __forceinline__ void add(const f32x4& o, const f32x4& w) { v += fmaf(o[0], w[0], o[1] * w[1]) + fmaf(o[2], w[2], o[3] * w[3]); }
  __device__ __forceinline__ float total() const { return v; }
};
// piece-wise transpose: dZ-side dW operand of a 16-feature tile, plus this lane's fp32 sum for the bias gradient
template <bool PAIR>
__device__ __forceinline__ void transpose_pieces(const BP& b, f16x8 id, AT& out, Sum<PAIR>& sum, const f32x4& rT) {
  uint32_t q[NP][2];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const f32x4 o = MFMA16(b.p[p], id, zero4());   // fp16-valued: the conversion back is exact
    // bias gradient: dZ of sample 4 g + r was evaluated on the mantissa of its dY; rT[r] restores the magnitude (globally scaled)
    sum.add(o, rT);
    q[p][0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(o[0], o[1]));
    q[p][1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(o[2], o[3]));
  }
  out.t01 = halves(q[0][0], q[0][1], q[1][0], q[1][1]);
}
// The 176 dW accumulators live in ACCUMULATION registers for the whole kernel: the empty asm statements pin the value to an
// AGPR on either side of its MFMA pair, and the compiler then emits the AGPR form of the two MFMAs itself (and keeps them in
// its hazard bookkeeping -- an MFMA written in asm is invisible to it, see split2).  Left to the register allocator, a changing
// subset of the accumulators was parked in AGPRs and moved to VGPRs and back around its pair (v_accvgpr_read x 4, two MFMAs,
// v_accvgpr_write x 4: 100 - 230 moves per tile, depending on the build).
template <bool PIN = true>
__device__ __forceinline__ f32x4 dw_mac(f32x4 acc, const AT& A, const BT& B) {
  if constexpr (PIN) asm("" : "+a"(acc));
  acc = MFMA16(A.t01, B.t11, acc);   // smallest first
  acc = MFMA16(A.t01, B.t00, acc);
  if constexpr (PIN) asm("" : "+a"(acc));
  return acc;
}
// (PIN = false: the wave-pair kernel below.  A kernel whose budget is 256 registers gets 128 + 128 as soon as anything in it asks
// for an accumulation register; without such a request the whole budget is VGPRs and the MFMAs use their VGPR form throughout.)
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
// two PAIRS at once, statement by statement: a dependent v_pk_fma_f32 needs a wait state after the one that feeds it, and the
// Horner chain of a single pair is nothing but such dependences (117 s_nop per tile) -- two chains side by side fill them
__device__ __forceinline__ void gelu_rational4(f32x2 za, f32x2 zb, f32x2& ha, f32x2& hb, f32x2& ga, f32x2& gb) {
  const f32x2 ea = (za * za) * sp2(-0.72134752044448170368f), eb = (zb * zb) * sp2(-0.72134752044448170368f);
  const f32x2 Ea = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const f32x2 Eb = {__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)};
#if !defined(PSDF_F16_GELU_ABS_MOD)
#define PSDF_F16_GELU_ABS_MOD 1
#endif
#if PSDF_F16_GELU_ABS_MOD
  // 1 + p |z|: two scalar fmas whose |.| is a source modifier (no instruction), instead of two v_and + one packed fma -- the
  // results only feed v_rcp_f32, which is scalar anyway; same fused arithmetic, bit-identical
  const f32x2 da = {__builtin_fmaf(__builtin_fabsf(za.x), 0.39f, 1.0f), __builtin_fmaf(__builtin_fabsf(za.y), 0.39f, 1.0f)};
  const f32x2 db = {__builtin_fmaf(__builtin_fabsf(zb.x), 0.39f, 1.0f), __builtin_fmaf(__builtin_fabsf(zb.y), 0.39f, 1.0f)};
#else
  const f32x2 da = pk_fma2(f32x2{fabsf(za.x), fabsf(za.y)}, sp2(0.39f), sp2(1.0f));
  const f32x2 db = pk_fma2(f32x2{fabsf(zb.x), fabsf(zb.y)}, sp2(0.39f), sp2(1.0f));
#endif
  const f32x2 ta = {__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
  const f32x2 tb = {__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
  f32x2 qa = sp2(5.384693295e-02f), qb = sp2(5.384693295e-02f);
#define HORNER(C) qa = pk_fma2(qa, ta, sp2(C)); qb = pk_fma2(qb, tb, sp2(C));
  HORNER(-2.582434118e-01f) HORNER(3.751679361e-01f) HORNER(-1.663514599e-02f) HORNER(1.944366544e-01f) HORNER(1.514270604e-01f)
#undef HORNER
  const f32x2 la = (qa * ta) * Ea, lb = (qb * tb) * Eb;
#if !defined(PSDF_F16_GELU_COPYSIGN)
#define PSDF_F16_GELU_COPYSIGN 1
#endif
#if PSDF_F16_GELU_COPYSIGN
  // cdf = 1/2 + sign(z) (1/2 - Phi(-|z|)): one v_bfi_b32 per value instead of a compare and a select (1/2 - l >= 0 always);
  // differs from `z < 0 ? l : 1 - l` by at most one rounding of the sum (6e-8)
  const f32x2 ma = sp2(0.5f) - la, mb = sp2(0.5f) - lb;
  const f32x2 ca = f32x2{__builtin_copysignf(ma.x, za.x), __builtin_copysignf(ma.y, za.y)} + sp2(0.5f);
  const f32x2 cb = f32x2{__builtin_copysignf(mb.x, zb.x), __builtin_copysignf(mb.y, zb.y)} + sp2(0.5f);
#else
  const f32x2 oa = sp2(1.0f) - la, ob = sp2(1.0f) - lb;
  const f32x2 ca = {za.x < 0.f ? la.x : oa.x, za.y < 0.f ? la.y : oa.y};
  const f32x2 cb = {zb.x < 0.f ? lb.x : ob.x, zb.y < 0.f ? lb.y : ob.y};
#endif
  ha = za * ca;
  hb = zb * cb;
  ga = pk_fma2(za, Ea * sp2(0.3989422804014327f), ca);
  gb = pk_fma2(zb, Eb * sp2(0.3989422804014327f), cb);
}
// in place: acc <- gelu(acc), gp <- gelu'(acc)
__device__ __forceinline__ void act_both(f32x4 (&acc)[NT], f32x4 (&gp)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; t++) {
    f32x2 ha, hb, ga, gb;
    gelu_rational4(f32x2{acc[t][0], acc[t][1]}, f32x2{acc[t][2], acc[t][3]}, ha, hb, ga, gb);
    acc[t] = f32x4{ha.x, ha.y, hb.x, hb.y};
    gp[t] = f32x4{ga.x, ga.y, gb.x, gb.y};
  }
}
// chain layer over the two k-steps of `in`; per_step(s, pieces) sees the operand pieces of each k-step.
// The weight records of a k-step (NTILE tiles x 2 pieces) are requested BEFORE the operand of that step is split, and those of
// step 1 before the MFMAs of step 0 are issued: the LDS round trip runs under the VALU work instead of in front of the MFMAs
template <int NTILE>
__device__ __forceinline__ void load_w(f16x8 (&a)[NTILE][NP], const u32x4* __restrict__ w_s) {
#pragma unroll
  for (int t = 0; t < NTILE; t++)
#pragma unroll
    for (int p = 0; p < NP; p++) a[t][p] = __builtin_bit_cast(f16x8, w_s[t * (2 * NP * 64) + p * 64]);
}
// out[t] += W(tile t) x operand pieces: three products, smallest first; two tiles at a time so that consecutive MFMAs go to
// different accumulators
template <int NTILE>
__device__ __forceinline__ void mac16r(f32x4 (&out)[NTILE], const BP& b, const f16x8 (&a)[NTILE][NP]) {
#pragma unroll
  for (int t0 = 0; t0 < NTILE; t0 += 2) {
#define PROD(PA, PB)                                                                  \
  _Pragma("unroll") for (int dt = 0; dt < 2; dt++) if (t0 + dt < NTILE) out[t0 + dt] = \
      MFMA16(a[t0 + dt][PA], b.p[PB], out[t0 + dt]);
    PROD(1, 0) PROD(0, 1) PROD(0, 0)
#undef PROD
  }
}
template <int NTILE, typename F>
__device__ __forceinline__ void chain(const f32x4 (&in)[NT], f32x4 (&out)[NTILE], const u32x4* __restrict__ w, int lane,
                                      F&& per_step) {
  f16x8 w0[NTILE][NP], w1[NTILE][NP];
  load_w<NTILE>(w0, w + lane);
  __builtin_amdgcn_sched_barrier(0);
  {
    float x[8];
    step_operand(in, 0, x);
    BP b;
    split8(x, b);
    load_w<NTILE>(w1, w + (NP * 64) + lane);
    __builtin_amdgcn_sched_barrier(0);
    mac16r<NTILE>(out, b, w0);
    per_step(0, b);
  }
  {
    float x[8];
    step_operand(in, 1, x);
    BP b;
    split8(x, b);
    mac16r<NTILE>(out, b, w1);
    per_step(1, b);
  }
}
// backward of one layer: dH chain (hands the pieces of dZ to the transposes), then dW[to][ti] += dZ(to) x H(ti)
template <int NTO, int NTI, bool PAIR>
__device__ __forceinline__ void layer_bwd(const f32x4 (&dz)[NT], f32x4 (&dh)[NTO], const u32x4* __restrict__ wT, int lane,
                                          const f16x8 (&id)[2], const f32x4 (&hT)[NTI], f32x4 (&dW)[NT][NTI], Sum<PAIR> (&db)[NT],
                                          const f32x4& rT) {
  AT A[NT];
  chain<NTO>(dz, dh, wT, lane, [&](int s, const BP& b) {
    transpose_pieces(b, id[0], A[2 * s], db[2 * s], rT);
    transpose_pieces(b, id[1], A[2 * s + 1], db[2 * s + 1], rT);
  });
#pragma unroll
  for (int ti = 0; ti < NTI; ti++) {
    BT B;
    split4(mul4(hT[ti], rT), B);      // H of sample 4 g + r carries that sample's dY magnitude (see the header)
#pragma unroll
    for (int to = 0; to < NT; to++) dW[to][ti] = dw_mac(dW[to][ti], A[to], B);
  }
}

// Range guard.  The two-piece scheme holds while every input and hidden activation stays below 2^8 in magnitude (the H-side
// dW operand is pre-scaled by 2^8 and fp16 ends at 65504) and every weight below 65504.  The kernels keep the largest |x|, |h|
// they meet (one v_max3_f32 per two values, 32 per tile) and raise the launch's guard word (absmax[1]) when it reaches
// RANGE_LIMIT; the pack kernel does the same for the weights.  A raised word makes the summing launch drop the images and lets
// the three-piece bf16 kernel -- queued behind every launch, a no-op while the word is zero -- redo the batch (dX is
// overwritten): psdf_mlp_backward_split_f16 below.
constexpr float RANGE_LIMIT = 255.0f;
__device__ __forceinline__ float amax_of(float m, const f32x4& v) {
  m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
  return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
}
__device__ __forceinline__ float amax_of8(float m, const float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(x[j])), __builtin_fabsf(x[j + 1]));
  return m;
}
// end of a kernel: one atomic per wave that saw a value out of range (NaN compares false: a NaN batch is NaN in every kernel)
__device__ __forceinline__ void range_report(float m, uint32_t* guard) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m >= RANGE_LIMIT) atomicOr(guard, 1u);
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
  // selects on the bit patterns (the nested conditional on floats compiled to an exec-masked block per tile)
  uint32_t m = (b & 0x807FFFFFu) | ((uint32_t)(127 + CHAIN_EXP) << 23), p = (uint32_t)(ex - CHAIN_EXP) << 23;
  m = special ? b : m;
  p = special ? 0x3F800000u : p;
  mant = __uint_as_float(zero ? 0u : m);
  pow2 = __uint_as_float(zero ? 0u : p);
}

// X [K0, N], dY [1, N], dX [K0, N] (optional) feature-major; img = the LDS image (mlp_split_pack_kernel); partial
// [gridDim.x][G_TOTAL] receives this workgroup's gradient image (of dY * 2^k: mlp_split_reduce_kernel takes the factor out
// again).  rows4 = K0 rounded up to a multiple of 4.
// The staged inputs are double buffered (two buffers per wave beside the image: 130 KB of LDS for K0 <= 48, 134 KB beyond; the
// single-buffer form of round 3 went with the image that needed it).
// GLDS (the three-tile instantiation, K0 <= 48): gelu' of the two inner layers waits in LDS (8 KB per wave) between the forward
// sweep and the backward chain instead of in 32 registers -- own-lane records, no synchronisation; with four input tiles the
// 32 KB do not fit beside the larger image (167 KB) and the values stay in registers
template <int NT0>
__global__ void __launch_bounds__(NWAVES * 64, 1)
    mlp_bwd_split_f16_kernel(int64_t N, int K0, int rows4, const float* __restrict__ X, const float* __restrict__ dY,
                             const u32x4* __restrict__ img, uint32_t* __restrict__ absmax, float* __restrict__ dX,
                             float* __restrict__ partial) {
  extern __shared__ __align__(16) u32x4 lds[];
  constexpr bool GLDS = NT0 == 3;
  uint32_t out_of_range = 0u;   // range guard: some lane of this wave met |input| or |hidden activation| >= RANGE_LIMIT (uniform)
  float sc, isc;
  const int kscale = dy_scale(absmax[0], sc, isc);      // 2^kscale * max|dY| in [2^4, 2^5)
  constexpr size_t IMG_ALIGNED = img_aligned(NT0);
  constexpr int OFF_F32 = off_f32(NT0);
  constexpr int NREC = (int)(IMG_ALIGNED / 16);
  for (int i = threadIdx.x; i < NREC; i += NWAVES * 64) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  // (the wave index as a scalar: tile numbers, loop bounds and the prefetch condition are then uniform to the compiler -- the
  //  prefetch block was exec-masked with vector compares before)
  const int lane_k = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const f16x8 id[2] = {ident_op(0, lane_k), ident_op(1, lane_k)};
  f32x4 dW1[NT][NT0], dW2[NT][NT], dW3[NT][NT];
#pragma unroll
  for (int to = 0; to < NT; to++) {
#pragma unroll
    for (int ti = 0; ti < NT; ti++) dW2[to][ti] = dW3[to][ti] = zero4();
#pragma unroll
    for (int ti = 0; ti < NT0; ti++) dW1[to][ti] = zero4();
  }
  constexpr bool PAIR = NT0 == 3;
  Sum<PAIR> db1[NT], db2[NT], db3[NT], dw4[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    db1[t].clear(); db2[t].clear(); db3[t].clear(); dw4[t].clear();
  }
  float db4 = 0.f;
  const int64_t ntiles = (N + 15) / 16;
  // staging buffer of a tile: 64 rows of 16 samples (rows < rows4 arrive by DMA, the rest are ZERO, written once below: the
  // layer-0 operand and the feature-lane copy of X are then read without a predicate or an index clamp -- 16 predicated
  // ds_read_b32 with their own address arithmetic were 110 instructions per tile), then dY (16 values in four copies: the
  // request writes one value per lane)
  constexpr int STAGE_ROWS = 64, stage_floats = STAGE_ROWS * 16 + 64, OFF_DY = STAGE_ROWS * 16;
  float* stage = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + IMG_ALIGNED) + wave * 2 * stage_floats;
  for (int i = rows4 * 16 + lane_k; i < STAGE_ROWS * 16; i += 64) {
    stage[i] = 0.f;
    stage[stage_floats + i] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
                                     (__attribute__((address_space(3))) void*)(buf + OFF_DY), 4, 0, 0);
  };
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave, tstride = (int64_t)gridDim.x * NWAVES;
  if (tile0 < ntiles) prefetch(tile0, stage);
  int cur = 0;
  for (int64_t tile = tile0; tile < ntiles; tile += tstride, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA completion is not tracked by the compiler
    const float* xb = stage + cur * stage_floats;
    // loop-invariant lane arithmetic (addresses, masks) is cheap to redo and expensive to keep: hoisted out of the loop it
    // ends up in scratch, and every scratch reload waits (vmcnt) for the LDS-DMA prefetch in flight
    int lane_l = lane_k;
    asm volatile("" : "+v"(lane_l));
    const int lane = lane_l, c = lane & 15, g = lane >> 4;
    const int64_t n0 = tile * 16, n = n0 + c;
    const bool live = n < N;
    // ---------------- forward recompute; h1, h2 leave the sweep as fp32 feature-lane tiles
    f32x4 a[NT], g1[NT], b[NT], g2[NT], h1T[NT], h2T[NT];
    float seen;            // largest |input|, |hidden activation| of this lane in this tile: lives through the forward sweep only
    f32x4* gl = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds) + IMG_ALIGNED +
                                         (size_t)NWAVES * 2 * stage_floats * 4) + wave * (2 * NT * 64) + lane;
    bias_init<NT>(a, tail, g);
    {
      f16x8 w00[NT][NP], w01[NT][NP];
      load_w<NT>(w00, lds + OFF_W0 + lane);
      float xs[2][8];
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int k = 32 * s + 8 * g + j;  // layer 0: natural k order (the image is packed to match); rows >= K0 are zero
          xs[s][j] = xb[k * 16 + c];   // (their weights are zero too)
        }
      __builtin_amdgcn_sched_barrier(0);
      BP bx;
      split8(xs[0], bx);
      load_w<NT>(w01, lds + OFF_W0 + (NP * 64) + lane);
      __builtin_amdgcn_sched_barrier(0);
      mac16r<NT>(a, bx, w00);
      split8(xs[1], bx);
      mac16r<NT>(a, bx, w01);
      seen = amax_of8(amax_of8(0.f, xs[0]), xs[1]);
    }
    bias_init<NT>(b, tail + HID, g);      // (requested ahead of the activation block: b is dead until the chain)
    act_both(a, g1);  // a = h1
#pragma unroll
    for (int t = 0; t < NT; t++) seen = amax_of(seen, a[t]);
    if constexpr (GLDS) {
#pragma unroll
      for (int t = 0; t < NT; t++) gl[t * 64] = g1[t];
    }
    chain<NT>(a, b, lds + OFF_W1, lane, [&](int s, const BP& p) {
      h1T[2 * s] = transpose_f32(p, id[0]);
      h1T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    bias_init<NT>(a, tail + 2 * HID, g);
    act_both(b, g2);  // b = h2
#pragma unroll
    for (int t = 0; t < NT; t++) seen = amax_of(seen, b[t]);
    if constexpr (GLDS) {
#pragma unroll
      for (int t = 0; t < NT; t++) gl[(NT + t) * 64] = g2[t];
    }
    chain<NT>(b, a, lds + OFF_W2, lane, [&](int s, const BP& p) {
      h2T[2 * s] = transpose_f32(p, id[0]);
      h2T[2 * s + 1] = transpose_f32(p, id[1]);
    });
    f32x4 dz[NT], w4[NT];
    {
      const float* wf0 = tail + 3 * HID;
#pragma unroll
      for (int t = 0; t < NT; t++) w4[t] = *reinterpret_cast<const f32x4*>(wf0 + 16 * t + 4 * g);
    }
    act_both(a, dz);  // a = h3, dz = gelu'(z3) for now
#pragma unroll
    for (int t = 0; t < NT; t++) seen = amax_of(seen, a[t]);
    out_of_range |= __builtin_amdgcn_ballot_w64(seen >= RANGE_LIMIT) != 0ull ? 1u : 0u;
    // ---------------- output layer: dW4 = sum dy h3, db4 = sum dy, dZ3 = w4 dy gelu'(z3); samples past N carry dy = 0,
    // which zeroes every contribution of theirs below
    f32x4 rT;   // 2^(e(n) + kscale) of the samples 4 g + r: what their H / dZ carry into the parameter gradients
    {
      f32x4 dyT = *reinterpret_cast<const f32x4*>(xb + OFF_DY + 4 * g);  // samples 4 g + r
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const bool in = n0 + 4 * g + r < N;
        const int ex = (int)(__float_as_uint(dyT[r]) >> 23) & 255;
        int er = ex + kscale - CHAIN_EXP + H_PRESCALE_EXP;  // dZ of the chain = true dZ * 2^(CHAIN_EXP - e(n)); + the H pre-scale
        er = er < 1 ? 0 : (er > 254 ? 254 : er);            // below 2^-126 after scaling: the contribution is dropped
        uint32_t bits = (uint32_t)er << 23;                 // (selects, not branches: the nested conditional compiled to four
        bits = ex > CHAIN_EXP ? bits : 0u;                  //  exec-masked blocks per tile)
        bits = ex == 255 ? 0x3F800000u : bits;
        rT[r] = __uint_as_float(in ? bits : 0u);
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
          dw4[2 * s + u].add(h3T, dyT);
        }
      }
    }
    // the chain of sample n runs on the mantissa of its dY (magnitude in [2^4, 2^5)); dX is multiplied by 2^(e(n) - 4) at the store
    float dy, dy_pow2;
    {
      const float dy_read = xb[OFF_DY + c];         // (read first, then select: `live ? read : 0` became an exec-masked load)
      dy_parts(live ? dy_read : 0.f, dy, dy_pow2);
    }
#pragma unroll
    for (int t = 0; t < NT; t++) dz[t] = mul4(dz[t], mul4s(w4[t], dy));
    // ---------------- layer 3
    zero_init<NT>(a);
    layer_bwd<NT, NT, PAIR>(dz, a, lds + OFF_T2, lane, id, h2T, dW3, db3, rT);  // a = dH2^T
#pragma unroll
    for (int t = 0; t < NT; t++) a[t] = mul4(a[t], GLDS ? gl[(NT + t) * 64] : g2[t]);   // dZ2^T
    // ---------------- layer 2 (the prefetch goes out here: late enough that the early part of the tile does not wait on
    // it, early enough for an HBM round trip before the next tile)
    if (tile + tstride < ntiles) prefetch(tile + tstride, stage + (cur ^ 1) * stage_floats);
    zero_init<NT>(dz);
    layer_bwd<NT, NT, PAIR>(a, dz, lds + OFF_T1, lane, id, h1T, dW2, db2, rT);  // dz = dH1^T
#pragma unroll
    for (int t = 0; t < NT; t++) dz[t] = mul4(dz[t], GLDS ? gl[t * 64] : g1[t]);         // dZ1^T
    // ---------------- layer 1: H = X in feature-lane order, straight from the staged rows
    f32x4 xT[NT0], dx[NT0];
#pragma unroll
    for (int u = 0; u < NT0; u++) {
      const int feat = 16 * u + c;
      xT[u] = *reinterpret_cast<const f32x4*>(xb + feat * 16 + 4 * g);       // (zero rows past K0)
    }
    zero_init<NT0>(dx);
    layer_bwd<NT0, NT0, PAIR>(dz, dx, lds + OFF_T0, lane, id, xT, dW1, db1, rT);  // dx = dX^T
    if (dX && live) {
      // row 16 t + 4 g + r: the lane's base (rows 4 g, sample n) once per tile, then a uniform row offset per store (it was a
      // 64-bit multiply-add per lane and store); tiles wholly inside K0 need no lane predicate
      float* p0 = dX + (int64_t)(4 * g) * N + n;
#pragma unroll
      for (int t = 0; t < NT0; t++) {
        if (16 * (t + 1) <= K0) {
#pragma unroll
          for (int r = 0; r < 4; r++) p0[(int64_t)(16 * t + r) * N] = dx[t][r] * dy_pow2;
        } else if (16 * t < K0) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (16 * t + 4 * g + r < K0) p0[(int64_t)(16 * t + r) * N] = dx[t][r] * dy_pow2;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (out_of_range && lane_k == 0) atomicOr(absmax + 1, 1u);
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
        float v1 = db1[t].total(), v2 = db2[t].total(), v3 = db3[t].total(), v4 = dw4[t].total();
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

// guard_drops: the launch has a bf16 launch queued behind it that redoes the batch when the range guard is raised (absmax[1]):
// the images are then dropped here.  events (host-mapped, may be NULL): count of raised guards, for the one-time warning.
__global__ void mlp_split_reduce_kernel(const float* __restrict__ partial, const uint32_t* __restrict__ absmax, int nimg, int K0, float* __restrict__ dW0,
                                        float* __restrict__ dW1, float* __restrict__ dW2, float* __restrict__ dW3,
                                        float* __restrict__ db0, float* __restrict__ db1, float* __restrict__ db2,
                                        float* __restrict__ db3, int guard_drops, volatile uint32_t* events) {
  // blockIdx.y = a slice of the images (a serial loop over 256 images per element left the chip idle: 63 us); the slices
  // meet in the destination with one float atomic each (the destinations are accumulated into anyway)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (absmax[1]) {
    if (events && e == 0 && blockIdx.y == 0) events[0] = events[0] + 1u;
    if (guard_drops) return;
  }
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
// records (one per piece); the tail threads copy biases / final weights.
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
    float wmax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t h, l;
      split2(w[2 * i], w[2 * i + 1], h, l);
      hi[i] = h;
      lo[i] = l;
      wmax = __builtin_fmaxf(__builtin_fmaxf(wmax, __builtin_fabsf(w[2 * i])), __builtin_fabsf(w[2 * i + 1]));
    }
    if (wmax >= 65504.f) atomicOr(absmax + 1, 1u);    // range guard (the words are cleared by a memset in front of this launch)
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
  }
}

}  // namespace

static int g_f16_form = 0;   // 1 once a split-fp16 backward has been launched (the wave-pair forms 2 / 3 of round 5 live in attic/rejected/)
namespace psdf {
size_t mlp_backward_split_scratch_bytes(int K0, int64_t N);      // mlp_bwd_split.hip
int mlp_backward_split_impl(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                            const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                            hipStream_t st, char* scratch, const uint32_t* only_if);
}
// one host-mapped word the summing launches count raised range guards into (NULL when the allocation fails: no warning then)
static uint32_t g_range_warned = 0;
static volatile uint32_t* range_events() {
  static uint32_t* dev_ptr = [] {
    uint32_t* h = nullptr;
    if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) != hipSuccess || !h) {
      (void)hipGetLastError();
      return (uint32_t*)nullptr;
    }
    h[0] = 0u;
    return h;
  }();
  return dev_ptr;
}

extern "C" {

// 0 = no split-fp16 backward yet, 1 = the last one ran mlp_bwd_split_f16_kernel (the only form built since round 6)
int psdf_mlp_backward_split_f16_form(void) { return g_f16_form; }

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
  const size_t stage_bytes = (size_t)NWAVES * (64 * 16 + 64) * 4;
  const size_t img_bytes = img_aligned(nt0);
  const size_t g_bytes = nt0 == 3 ? (size_t)NWAVES * 2 * NT * 64 * 16 : 0;     // gelu' of the inner layers (see GLDS)
  const size_t lds_bytes = img_bytes + 2 * stage_bytes + g_bytes;              // 159.0 KB (K0 <= 48) / 131.0 KB
  if (lds_bytes > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  if (N <= 0 || !X || !weights || !biases || !dY) return PSDF_ERR_ARG;
  for (int l = 0; l < 4; l++)
    if (!weights[l] || !biases[l] || !dW[l] || !db[l]) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = (N + 15) / 16;
  g_f16_form = 1;
  int64_t blocks = (ntiles + NWAVES - 1) / NWAVES;   // four tiles in flight per workgroup
  if (blocks > 256) blocks = 256;  // one workgroup per CU; each wave walks many tiles
  const size_t part_bytes = ((size_t)blocks * G_TOTAL * sizeof(float) + 15) & ~(size_t)15;
  // range guard (see RANGE_LIMIT): nets the three-piece bf16 kernel covers (K0 <= 52) get that kernel queued behind this one,
  // conditional on the guard word; wider inputs (53 .. 64: no such kernel) keep the saturating arithmetic and only count the event
  const bool guarded = K0 <= 52 && !getenv("PSDF_MLP_F16_NO_GUARD");
  const size_t fb_bytes = guarded ? psdf::mlp_backward_split_scratch_bytes(K0, N) : 0;
  char* scratch = (char*)psdf::stream_scratch(img_bytes + 16 + part_bytes + fb_bytes, st);   // NULL while capturing
  if (!scratch) return PSDF_ERR_UNSUPPORTED;
  uint32_t* rec = reinterpret_cast<uint32_t*>(scratch);
  uint32_t* absmax = reinterpret_cast<uint32_t*>(scratch + img_bytes);      // [0] max |dY| bits, [1] range guard
  float* partial = reinterpret_cast<float*>(scratch + img_bytes + 16);
  volatile uint32_t* events = range_events();
  if (events && events[0] != g_range_warned) {      // raised by an EARLIER call (the word is written by the device)
    if (!g_range_warned)
      fprintf(stderr, "psdf: MLP inputs / activations beyond the split-fp16 range (|value| >= 255): such batches are redone by the "
                      "three-piece bf16 kernel (slower); PSDF_MLP_BWD_SPLIT=bf16 selects it outright\n");
    g_range_warned = events[0];
  }
  if (hipMemsetAsync(absmax, 0, 16, st) != hipSuccess) return PSDF_ERR_UNSUPPORTED;
  const int pack_threads = (5 * NT + nt0) * 2 * 64 + TAIL_FLOATS;
#define PACK(NT0_)                                                                                                         \
  hipLaunchKernelGGL(mlp_split_pack_kernel<NT0_>, dim3((pack_threads + 255) / 256), dim3(256), 0, st, K0, weights[0],        \
                     weights[1], weights[2], weights[3], biases[0], biases[1], biases[2], biases[3], rec, absmax)
#define MAIN(NT0_)                                                                                                          \
  do {                                                                                                                      \
    auto kern = mlp_bwd_split_f16_kernel<NT0_>;                                                                              \
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
                     dW[0], dW[1], dW[2], dW[3], db[0], db[1], db[2], db[3], guarded ? 1 : 0, events);
  PSDF_LAUNCH_CHECK();
  if (guarded) {
    const int rc = psdf::mlp_backward_split_impl(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, st,
                                                 scratch + img_bytes + 16 + part_bytes, absmax + 1);
    if (rc != PSDF_OK) return rc;
  }
  return PSDF_OK;
}

// how many launches raised the range guard so far (the device counts into host-mapped memory: exact after a synchronisation)
unsigned psdf_mlp_f16_range_events(void) {
  volatile uint32_t* e = range_events();
  return e ? e[0] : 0u;
}

}  // extern "C"
