// Device-side building blocks of the fused MLP evaluators (shared by mlp.hip and fused.hip); see mlp.hip for the
// formulation (transposed products, chained D->B operand layout, packed LDS weight image).
#pragma once
#include "psdf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int MAXL = 5;  // max number of linear layers
constexpr int WS = 65;   // LDS row stride of a 64-lane weight row (64 + 1 pad float)

struct MlpPlan {
  int n_layers;        // number of linear layers (hidden layers + 1)
  int dims[MAXL + 1];  // true widths: dims[0] = input, dims[n_layers] = output
  int in_steps0;       // ceil(dims[0]/2)
  int tiles[MAXL + 1]; // tiles[i] = ceil(dims[i]/32) for i>=1
  int w_off[MAXL];     // float offsets into the packed buffer
  int b_off[MAXL];
  int total;           // packed floats
  int final_dot;       // 1 when the last layer is evaluated with VALU dot products (out <= 4)
};

__host__ __device__ inline int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

static int make_plan(int n_layers, const int* dims, MlpPlan& p) {
  if (n_layers < 2 || n_layers > MAXL) return PSDF_ERR_ARG;
  p.n_layers = n_layers;
  for (int i = 0; i <= n_layers; i++) {
    if (dims[i] <= 0) return PSDF_ERR_ARG;
    p.dims[i] = dims[i];
    p.tiles[i] = (dims[i] + 31) / 32;
  }
  p.in_steps0 = (dims[0] + 1) / 2;
  p.final_dot = dims[n_layers] <= 4;
  int off = 0;
  for (int l = 0; l < n_layers; l++) {
    p.w_off[l] = off;
    const bool last = (l == n_layers - 1);
    if (l == 0)
      off += p.tiles[1] * p.in_steps0 * WS;
    else if (last && p.final_dot)
      off += dims[n_layers] * p.tiles[l] * 32;
    else
      off += p.tiles[l + 1] * p.tiles[l] * 16 * 64;  // [to][ti][r/4][lane][r%4]: one ds_read_b128 feeds 4 MFMAs
    p.b_off[l] = off;
    off += (last && p.final_dot) ? 4 : p.tiles[l + 1] * 32;
  }
  p.total = off;
  return PSDF_OK;
}

// -------------------------------------------------------------------------------------- device math
// erf with < 1 ulp error, branch-free (both ranges evaluated, then selected): a ~20-instruction VALU
// sequence instead of the two-branch library erff, which matters because 96 GELUs per lane sit between
// the MFMA chains of every tile.  Polynomials: the widely used single-precision minimax pair
// (|x| <= 0.927734375: odd polynomial in x; above: 1 - exp(p(|x|))).
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a);
  const float s = a * a;
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

// The same erf on PAIRS of values with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of FMA
// per instruction on gfx950) -- identical operations in identical order, so the results are bit-identical to erf_fast;
// only the Horner chains (13 of the ~20 instructions) are paired, abs / exp / sign / select stay per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 erf_fast2(f32x2 a) {
  const f32x2 t = {fabsf(a.x), fabsf(a.y)};
  const f32x2 s = a * a;
  f32x2 r = pk_fma(splat2(-1.72853470e-5f), t, splat2(3.83197126e-4f));
  const f32x2 u = pk_fma(splat2(-3.88396438e-3f), t, splat2(2.42546219e-2f));
  r = pk_fma(r, s, u);
  r = pk_fma(r, t, splat2(-1.06777877e-1f));
  r = pk_fma(r, t, splat2(-6.34846687e-1f));
  r = pk_fma(r, t, splat2(-1.28717512e-1f));
  r = pk_fma(r, t, -t);
  const f32x2 hi = {copysignf(1.0f - __expf(r.x), a.x), copysignf(1.0f - __expf(r.y), a.y)};
  f32x2 q = splat2(-5.96761703e-4f);
  q = pk_fma(q, s, splat2(4.99119423e-3f));
  q = pk_fma(q, s, splat2(-2.67681349e-2f));
  q = pk_fma(q, s, splat2(1.12819925e-1f));
  q = pk_fma(q, s, splat2(-3.76125336e-1f));
  q = pk_fma(q, s, splat2(1.28379166e-1f));
  const f32x2 lo = pk_fma(q, a, a);
  return f32x2{t.x > 0.927734375f ? hi.x : lo.x, t.y > 0.927734375f ? hi.y : lo.y};
}
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 x) {
  return (splat2(0.5f) * x) * (splat2(1.0f) + erf_fast2(x * splat2(0.70710678118654752440f)));
}

__device__ __forceinline__ float gelu_exact(float x) {
  // torch.nn.GELU() default (erf form): 0.5*x*(1+erf(x/sqrt(2)))
  return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return fmaf(x, pdf, cdf);
}

template <int T>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[T], const float* __restrict__ bias_lds, int h) {
#pragma unroll
  for (int to = 0; to < T; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[to][r] = bias_lds[32 * to + row_of(r, h)];
}

template <int T>
__device__ __forceinline__ void apply_gelu(f32x16 (&acc)[T]) {
#pragma unroll
  for (int to = 0; to < T; to++)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 y = gelu_exact2(f32x2{acc[to][r], acc[to][r + 1]});
      acc[to][r] = y.x;
      acc[to][r + 1] = y.y;
    }
}

// gelu from ONE exponential and ONE reciprocal (tools/gelu_fit_rational.py): Phi(-|z|) = t P6(t) exp(-z^2/2) with
// t = 1 / (1 + 0.39 |z|), gelu(z) = max(z, 0) - |z| Phi(-|z|).  14 instructions against ~26 for the erf form; error
// against float64 1.8e-7 |z| (torch's fp32 formula 0.5 z (1 + erf(z / sqrt 2)) itself: 1.1e-7 |z|) -- both are rounding
// noise of an fp32 evaluation, and the GPU tests hold the kernels to a multiple of torch's own fp32 error.
__device__ __forceinline__ float gelu_rational(float z) {
  const float E = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(z), 0.39f, 1.0f));
  float q = 5.384693295e-02f;
  q = fmaf(q, t, -2.582434118e-01f);
  q = fmaf(q, t, 3.751679361e-01f);
  q = fmaf(q, t, -1.663514599e-02f);
  q = fmaf(q, t, 1.944366544e-01f);
  q = fmaf(q, t, 1.514270604e-01f);
  const float tail = q * t * E;
  return fmaf(-fabsf(z), tail, fmaxf(z, 0.f));
}

// One element per instruction: the form to use beside bf16 MFMAs (packed fp32 arithmetic does not hide in the shadow of
// the matrix pipe, plain VALU does: tools/mfma_valu_overlap.hip).  The split-bf16 forward is VALU bound (gelu + operand
// splitting against 156 MFMAs per tile), so it takes the cheaper evaluator: 0.799 -> 0.759 ms for the forward of the bench.
template <int T>
__device__ __forceinline__ void apply_gelu_scalar(f32x16 (&acc)[T]) {
#pragma unroll
  for (int to = 0; to < T; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[to][r] = gelu_rational(acc[to][r]);
}

// Two elements per instruction where the instruction set has a packed form (round 5): the two-piece fp16 forward issues 66
// MFMAs per tile where the bf16 one issues 156, so it is bound by the NUMBER of VALU instructions rather than by what hides
// beside the matrix pipe -- 9.5 instead of 14 instructions per element.  Same operations in the same order, fused where the
// scalar form is fused: bit-identical results.  |z| rides as a source modifier of the scalar fmas that need it.
__device__ __forceinline__ f32x2 gelu_rational2(f32x2 z) {
  const f32x2 e = (z * z) * f32x2{-0.72134752044448170368f, -0.72134752044448170368f};
  const f32x2 E = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
  const f32x2 t = {__builtin_amdgcn_rcpf(fmaf(fabsf(z.x), 0.39f, 1.0f)), __builtin_amdgcn_rcpf(fmaf(fabsf(z.y), 0.39f, 1.0f))};
  f32x2 q = {5.384693295e-02f, 5.384693295e-02f};
#define PSDF_H2(C) q = __builtin_elementwise_fma(q, t, f32x2{C, C});
  PSDF_H2(-2.582434118e-01f) PSDF_H2(3.751679361e-01f) PSDF_H2(-1.663514599e-02f) PSDF_H2(1.944366544e-01f) PSDF_H2(1.514270604e-01f)
#undef PSDF_H2
  const f32x2 tail = (q * t) * E;
  return f32x2{fmaf(-fabsf(z.x), tail.x, fmaxf(z.x, 0.f)), fmaf(-fabsf(z.y), tail.y, fmaxf(z.y, 0.f))};
}
template <int T>
__device__ __forceinline__ void apply_gelu_packed(f32x16 (&acc)[T]) {
#pragma unroll
  for (int to = 0; to < T; to++)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 y = gelu_rational2(f32x2{acc[to][r], acc[to][r + 1]});
      acc[to][r] = y.x;
      acc[to][r + 1] = y.y;
    }
}

// out^T = W * in^T for register-resident activations (chained layout, see header).
// Weight image of a chain layer: [(to,ti)][rq][lane][j] = A operand of k-step r = 4 rq + j, so one 128-bit LDS read
// per lane (conflict free: consecutive lanes, 16 bytes each) serves four MFMAs.
typedef float f32x4w __attribute__((ext_vector_type(4)));
template <int TI, int TO>
__device__ __forceinline__ void dense_chain(const f32x16 (&in)[TI], f32x16 (&out)[TO], const float* __restrict__ w_lds,
                                            int lane) {
#pragma unroll
  for (int ti = 0; ti < TI; ti++)
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      f32x4w a[TO];
#pragma unroll
      for (int to = 0; to < TO; to++)
        a[to] = *reinterpret_cast<const f32x4w*>(w_lds + (((to * TI + ti) * 4 + rq) * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float b = in[ti][4 * rq + j];
#pragma unroll
        for (int to = 0; to < TO; to++) out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[to][j], b, out[to], 0, 0, 0);
      }
    }
}

// ------------------------------------------------------------------------------ split-bf16 operand path
// fp32 MFMAs and VALU work do not overlap on gfx950 (their times add), bf16 MFMAs do and are 16x faster per k
// (tools/mfma_valu_overlap.hip, profiles/r01_mfma_valu_overlap.txt).  So the forward evaluator multiplies fp32
// operands as three bf16 pieces each, a = a1 + a2 + a3 (8 mantissa bits per piece, by truncation, so the sum is
// exact for |a| >= 2^-100 and within 2^-132 below: tests/test_split_bf16_numerics.py), and keeps the six products down to 2^-16 relative size:
//     a b ~= a3 b1 + a2 b2 + a1 b3 + a2 b1 + a1 b2 + a1 b1          (error ~2^-22 |a b|, fp32 accumulation)
// on v_mfma_f32_32x32x16_bf16: 6/16 of the fp32 matrix time, hidden under the GELU.  Same chained-register design:
// D tile of layer l = B operand of layer l+1.  Lane (sample n = lane & 31, half h = lane >> 5) supplies, for k-step s
// of input tile ti = s >> 1, its registers r = 8 (s & 1) + j, j = 0..7, i.e. the neurons 32 ti + row_of(r, h); the
// weight image is permuted to match.  Layer 0 reads features k = 16 s + 8 h + j of the feature-major input.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SPLIT_LDS_MAX = 80 * 1024;  // two workgroups per CU

struct SplitPlan {
  int ok;            // the image fits SPLIT_LDS_MAX (else the fp32 MFMA kernel is used)
  int base;          // float offset of the split image inside the packed buffer (multiple of 4: 16-byte records)
  int ns[MAXL];      // k-steps (16 inputs each) of layer l
  int w_rec[MAXL];   // record offset of layer l's weight image [to][s][piece][lane] (8 bf16 per record)
  int tail_rec;      // record offset of the fp32 tail (biases, final dot weights)
  int b_off[MAXL];   // float offsets inside the tail
  int wf_off;        // final dot weights [o][ti][r][h] (final_dot nets)
  int total_rec;
};

static void make_split_plan(const MlpPlan& p, SplitPlan& sp) {
  sp.base = (p.total + 3) & ~3;
  int rec = 0;
  const int nl = p.n_layers;
  for (int l = 0; l < MAXL; l++) sp.ns[l] = sp.w_rec[l] = sp.b_off[l] = 0;
  for (int l = 0; l < nl; l++) {
    const bool dot = (l == nl - 1) && p.final_dot;
    sp.ns[l] = l == 0 ? (p.dims[0] + 15) / 16 : 2 * p.tiles[l];
    sp.w_rec[l] = rec;
    if (!dot) rec += p.tiles[l + 1] * sp.ns[l] * 3 * 64;
  }
  sp.tail_rec = rec;
  int f = 0;
  for (int l = 0; l < nl; l++) {
    const bool dot = (l == nl - 1) && p.final_dot;
    if (dot) {
      sp.wf_off = f;
      f += p.dims[nl] * p.tiles[l] * 32;
    }
    sp.b_off[l] = f;
    f += dot ? 4 : p.tiles[l + 1] * 32;
  }
  if (!p.final_dot) sp.wf_off = 0;
  sp.total_rec = rec + (f + 3) / 4;
  sp.ok = (nl == 3 || nl == 4) && (size_t)sp.total_rec * 16 <= (size_t)SPLIT_LDS_MAX;
}

// three bf16 pieces of a float (top 16 bits of the running remainder); host + device, used by the pack kernel
__host__ __device__ inline void split3(float x, uint32_t (&piece)[3]) {
  float r = x;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    uint32_t u;
    __builtin_memcpy(&u, &r, 4);
    piece[i] = u >> 16;
    const uint32_t t = u & 0xFFFF0000u;
    float tf;
    __builtin_memcpy(&tf, &t, 4);
    r -= tf;
  }
}

// eight fp32 registers -> three bf16x8 B operands
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
  for (int i = 0; i < 4; i++) {  // v_perm_b32: {high half of odd element, high half of even element}
    q1[i] = __builtin_amdgcn_perm(a[2 * i + 1], a[2 * i], 0x07060302u);
    q2[i] = __builtin_amdgcn_perm(b[2 * i + 1], b[2 * i], 0x07060302u);
    q3[i] = __builtin_amdgcn_perm(c[2 * i + 1], c[2 * i], 0x07060302u);
  }
  p1 = __builtin_bit_cast(bf16x8, q1);
  p2 = __builtin_bit_cast(bf16x8, q2);
  p3 = __builtin_bit_cast(bf16x8, q3);
}

// bias rows of a D tile: registers 4q..4q+3 of lane half h are the consecutive rows 8q + 4h .. +3 -> one 128-bit read
// (every bias block of the split image starts on a 16-byte boundary)
template <int T>
__device__ __forceinline__ void init_bias4(f32x16 (&acc)[T], const float* __restrict__ bias_lds, int h) {
  const f32x4w* __restrict__ b4 = reinterpret_cast<const f32x4w*>(bias_lds);
#pragma unroll
  for (int to = 0; to < T; to++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4w v = b4[8 * to + 2 * q + h];
#pragma unroll
      for (int c = 0; c < 4; c++) acc[to][4 * q + c] = v[c];
    }
}

// ---- the same with TWO fp16 pieces per operand (round 3; opt-in, see psdf_mlp_forward_f16): a = a0 + a1, a0 = fp16(a) rounded
// to nearest (v_cvt_pk_f16_f32; the remainder is exact in fp32 and fits 12 bits), a1 = fp16(a - a0) to nearest (round 6; rounds
// 3-5 truncated both pieces, a one-sided error of up to 3 * 2^-23 per operand); products a1 b0 + a0 b1 + a0 b0 on v_mfma_f32_32x32x16_f16.
// Half the MFMAs and about half the splitting work of the three-piece bf16 scheme; gfx950's matrix pipe honours fp16 subnormals
// (attic/prototypes/mlp_fwd_split_f16.hip), so small low pieces keep an absolute precision of 2^-24; values must stay below 65504.
// The image keeps the three-slot record layout (slot 2 unused), so SplitPlan is shared.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8h(const float (&x)[8], f16x8& hi, f16x8& lo) {
  u32x4 qh, ql;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const h2v_t h2 = __builtin_convertvector(f32x2{x[2 * i], x[2 * i + 1]}, h2v_t);   // v_cvt_pk_f16_f32, nearest even
    const f32x2 r = f32x2{x[2 * i], x[2 * i + 1]} - f32x2{(float)h2[0], (float)h2[1]};      // exact
    const h2v_t l2 = __builtin_convertvector(r, h2v_t);
    qh[i] = __builtin_bit_cast(uint32_t, h2);
    ql[i] = __builtin_bit_cast(uint32_t, l2);
  }
  hi = __builtin_bit_cast(f16x8, qh);
  lo = __builtin_bit_cast(f16x8, ql);
}
// two pieces of a float for the pack kernel: {hi, lo} as 16-bit patterns
__device__ __forceinline__ void split2h_bits(float x, uint32_t (&piece)[3]) {
  const h2v_t h2 = __builtin_convertvector(f32x2{x, 0.f}, h2v_t);
  const float r = x - (float)h2[0];
  const h2v_t l2 = __builtin_convertvector(f32x2{r, 0.f}, h2v_t);
  piece[0] = __builtin_bit_cast(uint32_t, h2) & 0xFFFFu;
  piece[1] = __builtin_bit_cast(uint32_t, l2) & 0xFFFFu;
  piece[2] = 0u;
}

// one k-step (16 inputs) into TO output tiles.  w_s -> record [to = 0][s][piece 0][lane 0]; `to` stride = ns*192 records.
template <int TO, bool F16 = false>
__device__ __forceinline__ void split_mac(f32x16 (&out)[TO], const float (&x)[8], const u32x4* __restrict__ w_s, int ns,
                                          int lane) {
  if constexpr (F16) {
    f16x8 bh, bl;
    split8h(x, bh, bl);
#pragma unroll
    for (int to = 0; to < TO; to++) {
      const u32x4* wt = w_s + to * ns * 192 + lane;
      const f16x8 ah = __builtin_bit_cast(f16x8, wt[0]);
      const f16x8 al = __builtin_bit_cast(f16x8, wt[64]);
      out[to] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, out[to], 0, 0, 0);  // smallest terms first
      out[to] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, out[to], 0, 0, 0);
      out[to] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, out[to], 0, 0, 0);
    }
    return;
  }
  bf16x8 b1, b2, b3;
  split8(x, b1, b2, b3);
#pragma unroll
  for (int to = 0; to < TO; to++) {
    const u32x4* wt = w_s + to * ns * 192 + lane;
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, wt[0]);
    const bf16x8 a2 = __builtin_bit_cast(bf16x8, wt[64]);
    const bf16x8 a3 = __builtin_bit_cast(bf16x8, wt[128]);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, out[to], 0, 0, 0);  // smallest terms first
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, out[to], 0, 0, 0);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, out[to], 0, 0, 0);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, out[to], 0, 0, 0);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, out[to], 0, 0, 0);
    out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, out[to], 0, 0, 0);
  }
}

template <int TI, int TO, bool F16 = false>
__device__ __forceinline__ void split_chain(const f32x16 (&in)[TI], f32x16 (&out)[TO], const u32x4* __restrict__ w,
                                            int lane) {
#pragma unroll
  for (int s = 0; s < 2 * TI; s++) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = in[s >> 1][8 * (s & 1) + j];
    split_mac<TO, F16>(out, x, w + s * 192, 2 * TI, lane);
  }
}


}  // namespace
