// Prototype of the FULL backward (dX, dW1..4, db1..4) of the 36-64-64-64-1 SDF MLP with split-bf16 operands on
// v_mfma_f32_32x32x16_bf16 -- the plan of DESIGN.md ("fp32 MFMAs do not overlap with VALU work").  Per 32-sample tile:
//   forward recompute (h_l, gelu'(z_l) kept in registers)                       tools/prototypes/mlp_fwd_split_bf16.hip
//   dH chain with transposed weight images                                      tools/prototypes/mlp_dx_split_bf16.hip
//   dW_l += dZ_l^T-tile x H_{l-1}-tile with k = the 32 samples: both operands need sample<->feature transposes, done one
//     32x32 fp32 tile at a time through a 4.6 KB per-wave LDS scratch (row stride 36 floats), then split into pieces;
//     the transposed tile of dZ is the A operand, that of H the B operand (same lane mapping); X is read transposed
//     straight from its feature-major rows.  db_l and dW4 = sum dy h3 fall out of the transposed tiles (one register each).
//   dW accumulators (12 tiles x 16 registers) persist for the life of the wave; at the end the four waves add them into an
//   fp32 image in LDS (the weight images are dead by then), the workgroup stores the image, a second launch sums the images.
// One wave per SIMD (the register budget), 160.8 KB of LDS.  Checked against float64 at N = 65536, timed at N = 2 M.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/prototypes/mlp_bwd_split_bf16.hip -o tools/mlp_bwd_split_bf16
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
static constexpr int NWAVES = 4, TS = 36;                                  // transposition scratch: 32 rows x 36 floats per wave
static constexpr size_t IMG_ALIGNED = (IMG_BYTES + 15) / 16 * 16;
static constexpr size_t LDS_BYTES = IMG_ALIGNED + (size_t)NWAVES * 32 * TS * 4;
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

#define NT_FENCE()                                              \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D-layout tile (lane = sample, registers = 16 of the 32 feature rows) -> lane = feature row (lane & 31), o[ks][j] =
// sample 16 ks + 8 (lane >> 5) + j: the A / B operand order of a k-step over samples.
__device__ __forceinline__ void tr_tile(const f32x16& t, float* sc, int lane, float (&o)[2][8]) {
  const int h = lane >> 5, sl = lane & 31;
  NT_FENCE();  // every lane is done reading the previous tile
#pragma unroll
  for (int r = 0; r < 16; r++) sc[row_of(r, h) * TS + sl] = t[r];
  NT_FENCE();
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const f32x4* p = reinterpret_cast<const f32x4*>(sc + sl * TS + 16 * ks + 8 * h);
    const f32x4 v0 = p[0], v1 = p[1];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      o[ks][j] = v0[j];
      o[ks][4 + j] = v1[j];
    }
  }
}

struct Pieces {  // one transposed 32x32 tile as MFMA operands: [k-step over samples][piece]
  bf16x8 p[2][3];
};
__device__ __forceinline__ void to_pieces(const float (&o)[2][8], Pieces& q) {
#pragma unroll
  for (int ks = 0; ks < 2; ks++) split8(o[ks], q.p[ks][0], q.p[ks][1], q.p[ks][2]);
}
__device__ __forceinline__ float sum16(const float (&o)[2][8]) {
  float s = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int j = 0; j < 8; j++) s += o[ks][j];
  return s;
}
// acc (rows = features of A's tile, cols = features of B's tile) += A B over the 32 samples, six products per k-step
__device__ __forceinline__ void dw_mac(f32x16& acc, const Pieces& A, const Pieces& B) {
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[ks][2], B.p[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[ks][1], B.p[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[ks][0], B.p[ks][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[ks][1], B.p[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[ks][0], B.p[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[ks][0], B.p[ks][0], acc, 0, 0, 0);
  }
}
// dW[to][ti] += dZ(tile to) x H(tile ti) for one layer; db partials from the transposed dZ tiles.
// LEAN: one A and one B tile of pieces live at a time (H tiles transposed and split twice) -- 24 registers less.
template <bool LEAN>
__device__ __forceinline__ void dw_layer(f32x16 (&dW)[2][2], float (&db)[2], const f32x16 (&dz)[2], const f32x16 (&hin)[2],
                                         float* sc, int lane) {
  if constexpr (LEAN) {
#pragma unroll
    for (int to = 0; to < 2; to++) {
      Pieces A;
      {
        float o[2][8];
        tr_tile(dz[to], sc, lane, o);
        db[to] += sum16(o);
        to_pieces(o, A);
      }
#pragma unroll
      for (int ti = 0; ti < 2; ti++) {
        float o[2][8];
        tr_tile(hin[ti], sc, lane, o);
        Pieces B;
        to_pieces(o, B);
        dw_mac(dW[to][ti], A, B);
      }
    }
  } else {
    Pieces A[2];
#pragma unroll
    for (int to = 0; to < 2; to++) {
      float o[2][8];
      tr_tile(dz[to], sc, lane, o);
      db[to] += sum16(o);
      to_pieces(o, A[to]);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ti++) {
      float o[2][8];
      tr_tile(hin[ti], sc, lane, o);
      Pieces B;
      to_pieces(o, B);
#pragma unroll
      for (int to = 0; to < 2; to++) dw_mac(dW[to][ti], A[to], B);
    }
  }
}

template <int TERMS, bool LEAN>
__global__ void __launch_bounds__(NWAVES * 64, 1)
    bwdk(int64_t N, const float* __restrict__ X, const float* __restrict__ dY, const u32x4* __restrict__ img,
         float* __restrict__ dX, float* __restrict__ partial) {
  extern __shared__ __align__(16) u32x4 lds[];
  constexpr int NREC = (int)(IMG_ALIGNED / 16);
  for (int i = threadIdx.x; i < NREC; i += NWAVES * 64) lds[i] = img[i];
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(lds + OFF_F32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, sl = lane & 31;
  float* sc = reinterpret_cast<float*>(lds) + IMG_ALIGNED / 4 + wave * 32 * TS;  // this wave's transposition scratch
  f32x16 dW1[2][2], dW2[2][2], dW3[2][2];
#pragma unroll
  for (int to = 0; to < 2; to++)
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
      for (int r = 0; r < 16; r++) dW1[to][ti][r] = dW2[to][ti][r] = dW3[to][ti][r] = 0.f;
  float db1[2] = {0.f, 0.f}, db2[2] = {0.f, 0.f}, db3[2] = {0.f, 0.f}, dw4[2] = {0.f, 0.f}, db4 = 0.f;
  const int64_t ntiles = N / 32;  // prototype: N is a multiple of 32
  for (int64_t tile = (int64_t)blockIdx.x * NWAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * NWAVES) {
    asm volatile("" ::: "memory");
    const int64_t n0 = tile * 32, n = n0 + sl;
    // ---------------- forward recompute
    f32x16 h1[2], g1[2], h2[2], g2[2], c[2];
    bias_init(h1, tail, h);
    {
      float xs[S0][8];
#pragma unroll
      for (int s = 0; s < S0; s++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int k = 16 * s + 8 * h + j;
          xs[s][j] = k < K0 ? X[(int64_t)k * N + n] : 0.f;
        }
#pragma unroll
      for (int s = 0; s < S0; s++) mac<TERMS, S0>(h1, xs[s], lds + OFF_W0 + s * 3 * 64, lane);
    }
    act_both(h1, g1);
    bias_init(h2, tail + HID, h);
    chain<TERMS>(h1, h2, lds + OFF_W1, lane);
    act_both(h2, g2);
    bias_init(c, tail + 2 * HID, h);
    chain<TERMS>(h2, c, lds + OFF_W2, lane);
    f32x16 dz[2];
    act_both(c, dz);  // c = h3, dz = gelu'(z3) for now
    // ---------------- output layer: dW4 = sum dy h3, db4 = sum dy, dZ3 = w4 dy gelu'(z3)
    float dyt[2][8];  // dy of the samples this lane owns in the transposed tiles
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const f32x4* p = reinterpret_cast<const f32x4*>(dY + n0 + 16 * ks + 8 * h);
      const f32x4 v0 = p[0], v1 = p[1];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        dyt[ks][j] = v0[j];
        dyt[ks][4 + j] = v1[j];
      }
    }
    db4 += sum16(dyt);
#pragma unroll
    for (int to = 0; to < 2; to++) {
      float o[2][8];
      tr_tile(c[to], sc, lane, o);
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) s = fmaf(o[ks][j], dyt[ks][j], s);
      dw4[to] += s;
    }
    const float dy = dY[n];
    const float* wf = tail + 3 * HID;
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) dz[to][r] *= wf[32 * to + row_of(r, h)] * dy;
    // ---------------- layer 3
    dw_layer<LEAN>(dW3, db3, dz, h2, sc, lane);
    zero_init(c);
    chain<TERMS>(dz, c, lds + OFF_T2, lane);  // dH2^T
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) c[to][r] *= g2[to][r];  // dZ2^T
    // ---------------- layer 2
    dw_layer<LEAN>(dW2, db2, c, h1, sc, lane);
    zero_init(dz);
    chain<TERMS>(c, dz, lds + OFF_T1, lane);  // dH1^T
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) dz[to][r] *= g1[to][r];  // dZ1^T
    // ---------------- layer 1: B operand = X transposed, read straight from the feature-major rows
    {
      Pieces A[2];
#pragma unroll
      for (int to = 0; to < 2; to++) {
        float o[2][8];
        tr_tile(dz[to], sc, lane, o);
        db1[to] += sum16(o);
        to_pieces(o, A[to]);
      }
#pragma unroll
      for (int ti = 0; ti < 2; ti++) {
        const int feat = 32 * ti + sl;
        float o[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
          if (feat < K0) {
            const f32x4* p = reinterpret_cast<const f32x4*>(X + (int64_t)feat * N + n0 + 16 * ks + 8 * h);
            v0 = p[0];
            v1 = p[1];
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            o[ks][j] = v0[j];
            o[ks][4 + j] = v1[j];
          }
        }
        Pieces B;
        to_pieces(o, B);
#pragma unroll
        for (int to = 0; to < 2; to++) dw_mac(dW1[to][ti], A[to], B);
      }
    }
    zero_init(c);
    chain<TERMS>(dz, c, lds + OFF_T0, lane);  // dX^T
#pragma unroll
    for (int to = 0; to < 2; to++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 32 * to + row_of(r, h);
        if (k < K0) dX[(int64_t)k * N + n] = c[to][r];
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
      for (int to = 0; to < 2; to++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int idx = (32 * to + row_of(r, h)) * 64 + 32 * ti + sl;  // [out][in]
            G[G_W1 + idx] += dW1[to][ti][r];
            G[G_W2 + idx] += dW2[to][ti][r];
            G[G_W3 + idx] += dW3[to][ti][r];
          }
#pragma unroll
      for (int to = 0; to < 2; to++) {
        const float b1 = db1[to] + __shfl_xor(db1[to], 32, 64), b2 = db2[to] + __shfl_xor(db2[to], 32, 64),
                    b3 = db3[to] + __shfl_xor(db3[to], 32, 64), w4 = dw4[to] + __shfl_xor(dw4[to], 32, 64);
        if (h == 0) {
          G[G_B1 + 32 * to + sl] += b1;
          G[G_B2 + 32 * to + sl] += b2;
          G[G_B3 + 32 * to + sl] += b3;
          G[G_W4 + 32 * to + sl] += w4;
        }
      }
      const float b4 = db4 + __shfl_xor(db4, 32, 64);
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
  float *dXin, *dYd, *dXout, *dPart, *dGrad;
  u32x4* dI;
  hipMalloc(&dXin, X.size() * 4);
  hipMalloc(&dYd, N * 4);
  hipMalloc(&dXout, X.size() * 4);
  hipMalloc(&dPart, (size_t)256 * G_TOTAL * 4);
  hipMalloc(&dGrad, G_TOTAL * 4);
  hipMalloc(&dI, img.size());
  hipMemcpy(dI, img.data(), img.size(), hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)bwdk<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
  hipFuncSetAttribute((const void*)bwdk<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
  printf("LDS per workgroup %zu B, gradient image %d floats\n", LDS_BYTES, G_TOTAL);

  bool lean = false;
  auto kernel = [&](int64_t blocks, int64_t n) {
    if (lean) hipLaunchKernelGGL((bwdk<6, true>), dim3((unsigned)blocks), dim3(NWAVES * 64), LDS_BYTES, 0, n, dXin, dYd, dI, dXout, dPart);
    else hipLaunchKernelGGL((bwdk<6, false>), dim3((unsigned)blocks), dim3(NWAVES * 64), LDS_BYTES, 0, n, dXin, dYd, dI, dXout, dPart);
  };
  auto run = [&](int64_t n, const float* xh, const float* dyh) {  // xh: [K0][n] feature-major
    hipMemcpy(dXin, xh, (size_t)K0 * n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dYd, dyh, n * 4, hipMemcpyHostToDevice);
    int64_t blocks = n / 32 / NWAVES;
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
           lean ? "lean" : "wide", (long long)NC, exmax, exref, exmax / exref);
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
           lean ? "lean" : "wide", best, (long long)N, 42240.0 * N / (best * 1e-3) / 1e12);
  };
  for (int v = 0; v < 2; v++) {
    lean = v == 1;
    if (!check()) return 1;
    timing();
    fflush(stdout);
  }
  return 0;
}
