// Fused permutohedral encoding -> MLP evaluation for gfx950 (pos_dim 3, 2 features per level).
//
// Replaces, in ONE launch, the pair the reference evaluates for every SDF query
//   feat = self.encoding(points, window); out = self.mlp_sdf(feat)          (permuto_sdf_py/models/models.py:186-192)
// without ever writing the [N, 2L+3] feature tensor (256-384 MiB at the BASELINE sizes, SURVEY.md section 8 cfg 2).
//
// Why it fuses without any data movement: the MLP kernels (mlp.hip) compute Z^T = W * H^T with
// v_mfma_f32_32x32x2_f32; for the first layer the B operand of k-step s is, at lane (h = lane>>5, sl = lane&31),
// input feature 2s+h of sample sl.  With F = 2 that is "feature h of LEVEL s of sample sl": an encoding level is
// exactly one MFMA k-step.  So lane (h, sl) evaluates the levels of parity h for sample sl (simplex, 4 hashed
// 8-byte gathers, barycentric blend -- the same expressions in the same order as encode.hip, so the results are
// bit-identical to the unfused pair), and one cross-half exchange per level PAIR (v_permlane32_swap-shaped:
// lower half gives feature 1 of its level, upper half gives feature 0 of its level) yields the two B operands.
// The concatenated, scaled input point (pseudo-levels L, L+1) drops out of the same scheme.
//
// The per-sample `skip` mask lets fixed-shape callers (the sphere tracer, sphere_trace.py) keep one slot per ray:
// a wave whose 32 samples are all masked does nothing.
//
// Launch shape: 256 threads = 4 waves, each wave walks 32-sample tiles; <= 1024 workgroups (weights are staged in
// LDS once per workgroup).  Gather latency is covered inside the wave: 4 levels x 4 gathers are issued per batch
// before they are consumed, and the MFMA chain of the previous batch overlaps with them.
#include "encode_device.h"
#include "mlp_device.h"

namespace {

struct EncArgs {
  int L;               // hashed levels
  int Lt;              // L + pseudo-levels (== in_steps0 of the net)
  uint32_t capacity;
  EncConv conv;            // runtime conventions (encode_device.h)
  const float* positions;  // [N,3]
  const float* lattice;    // [L,T,2]
  const float* scale_factor;  // [L,3]
  const float* shifts;        // [L,3]
  const float* window;        // [L]
  float points_scaling;
  int C;               // rows of `feat`: 2*Lt (padded pseudo-levels) or 2*L + 3 (appended points), encode_conventions.h
};

constexpr int JB = 1;  // level pairs per gather batch

// features (f0,f1) of level `lv` (hashed if lv < L, scaled point if L <= lv < Lt, zero beyond) for one sample
struct LevelLoad {
  float2 v[4];
  float bw[4];
};

// `j` is wave-uniform: the constants of both levels of the pair come in through scalar loads and each half-wave
// selects its own (a lane-indexed read would cost 7 vector loads per level on top of the 4 gathers).
__device__ __forceinline__ void level_issue(const EncArgs& e, const float (&pos)[3], int j, int h, LevelLoad& ld) {
  const int l0 = (2 * j < e.L) ? 2 * j : 0;          // clamped: out-of-range levels gather (and discard) level 0
  const int l1 = (2 * j + 1 < e.L) ? 2 * j + 1 : 0;
  float sh[3], sf[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float a = e.shifts[l0 * 3 + i], b = e.shifts[l1 * 3 + i];
    const float c = e.scale_factor[l0 * 3 + i], d = e.scale_factor[l1 * 3 + i];
    sh[i] = h ? b : a;
    sf[i] = h ? d : c;
  }
  const float w0 = e.window[l0], w1 = e.window[l1];
  const float w = h ? w1 : w0;
  const int lvc = h ? l1 : l0;
  Simplex<3> s;
  compute_simplex<3>(pos, sh, sf, s, e.conv.tie_later);
  const float* __restrict__ table = e.lattice + (int64_t)lvc * e.capacity * 2;
  uint32_t rows[4];
  vertex_rows<3>(s, e.capacity, rows, e.conv.hash_c);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t row = rows[r];
    ld.v[r] = *reinterpret_cast<const float2*>(table + (int64_t)row * 2);
    ld.bw[r] = s.bary[r] * w;
  }
}

__device__ __forceinline__ float2 level_finish(const EncArgs& e, const float (&pos)[3], int lv, const LevelLoad& ld) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    a0 = a0 + ld.v[r].x * ld.bw[r];
    a1 = a1 + ld.v[r].y * ld.bw[r];
  }
  if (lv >= e.L) {
    const int d = (lv - e.L) * 2;
    a0 = (d == 0) ? pos[0] * e.points_scaling : (d == 2) ? pos[2] * e.points_scaling : 0.f;
    a1 = (d == 0) ? pos[1] * e.points_scaling : 0.f;
    if (lv >= e.Lt) a0 = 0.f, a1 = 0.f;
  }
  return make_float2(a0, a1);
}

// Layer 0 of the net with the encoding as its B operand.  `feat` (optional) receives the feature-major [2*Lt, N]
// tensor as a by-product (training forward: the backward kernels read it).
template <int T1>
__device__ __forceinline__ void encode_layer0(const MlpPlan& p, const float* __restrict__ lds, const EncArgs& e,
                                              const float (&pos)[3], int lane, int h, f32x16 (&h1)[T1],
                                              float* __restrict__ feat, int64_t N, int64_t n, bool n_ok) {
  const float* __restrict__ w0 = lds + p.w_off[0];
  const int steps = p.in_steps0;
  const int npairs = (steps + 1) >> 1;
  for (int j0 = 0; j0 < npairs; j0 += JB) {
    LevelLoad ld[JB];
#pragma unroll
    for (int jj = 0; jj < JB; jj++) level_issue(e, pos, j0 + jj, h, ld[jj]);
#pragma unroll
    for (int jj = 0; jj < JB; jj++) {
      const int j = j0 + jj;
      if (j < npairs) {  // wave-uniform
        const int lv = 2 * j + h;
        const float2 f = level_finish(e, pos, lv, ld[jj]);
        if (feat && n_ok && lv < e.Lt) {
          feat[(int64_t)(2 * lv) * N + n] = f.x;
          if (2 * lv + 1 < e.C) feat[(int64_t)(2 * lv + 1) * N + n] = f.y;
        }
        const float give = h ? f.x : f.y;
        const float got = __shfl_xor(give, 32, 64);
        const float b_even = h ? got : f.x;   // k-step 2j   : feature h of level 2j
        const float b_odd = h ? f.y : got;    // k-step 2j+1 : feature h of level 2j+1
#pragma unroll
        for (int to = 0; to < T1; to++)
          h1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[(to * steps + 2 * j) * WS + lane], b_even, h1[to], 0, 0, 0);
        if (2 * j + 1 < steps) {
#pragma unroll
          for (int to = 0; to < T1; to++)
            h1[to] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(w0[(to * steps + 2 * j + 1) * WS + lane], b_odd, h1[to], 0, 0, 0);
        }
      }
    }
  }
}

template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT>
__global__ void __launch_bounds__(PSDF_BLOCK, 3)
    fused_fwd_kernel(MlpPlan p, EncArgs e, int64_t N, const unsigned char* __restrict__ skip,
                     const float* __restrict__ packed, float* __restrict__ feat, float* __restrict__ Y) {
  extern __shared__ __align__(16) float lds[];
  for (int i = threadIdx.x; i < p.total; i += PSDF_BLOCK) lds[i] = packed[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int OUT = p.dims[p.n_layers];
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");  // keep the LDS weight reads inside the tile loop (see mlp.hip)
    const int64_t n = tile * 32 + sl;
    const bool n_ok = n < N;
    const int64_t nc = n_ok ? n : N - 1;
    if (skip) {
      const bool active = n_ok && !skip[nc];
      if (__ballot(active) == 0) continue;
    }
    float pos[3];
    load_pos<3>(e.positions, nc, pos);
    f32x16 h1[T1];
    init_bias<T1>(h1, lds + p.b_off[0], h);
    encode_layer0<T1>(p, lds, e, pos, lane, h, h1, feat, N, n, n_ok);
    apply_gelu<T1>(h1);
    f32x16 h2[T2];
    init_bias<T2>(h2, lds + p.b_off[1], h);
    dense_chain<T1, T2>(h1, h2, lds + p.w_off[1], lane);
    apply_gelu<T2>(h2);
    constexpr int TL = (T3 > 0) ? T3 : T2;
    f32x16 hl[TL];
    if constexpr (T3 > 0) {
      init_bias<T3>(hl, lds + p.b_off[2], h);
      dense_chain<T2, T3>(h2, hl, lds + p.w_off[2], lane);
      apply_gelu<T3>(hl);
    } else {
#pragma unroll
      for (int t = 0; t < T2; t++) hl[t] = h2[t];
    }
    const int lf = p.n_layers - 1;
    if constexpr (FINAL_DOT) {
      const float* __restrict__ wf = lds + p.w_off[lf];
      for (int o = 0; o < OUT; o++) {
        float acc = 0.f;
#pragma unroll
        for (int ti = 0; ti < TL; ti++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc = fmaf(wf[((o * TL + ti) * 16 + r) * 2 + h], hl[ti][r], acc);
        acc += __shfl_xor(acc, 32, 64);
        acc += lds[p.b_off[lf] + o];
        if (h == 0 && n_ok) Y[(int64_t)o * N + n] = acc;
      }
    } else {
      f32x16 y[OUT_T];
      init_bias<OUT_T>(y, lds + p.b_off[lf], h);
      dense_chain<TL, OUT_T>(hl, y, lds + p.w_off[lf], lane);
      if (n_ok) {
#pragma unroll
        for (int to = 0; to < OUT_T; to++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = 32 * to + row_of(r, h);
            if (row < OUT) Y[(int64_t)row * N + n] = y[to][r];
          }
      }
    }
  }
}

template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT>
int launch_fused(const MlpPlan& p, const EncArgs& e, int64_t N, const unsigned char* skip, const float* packed,
                 float* feat, float* Y, hipStream_t st) {
  const size_t shmem = (size_t)p.total * sizeof(float);
  auto kern = fused_fwd_kernel<T1, T2, T3, OUT_T, FINAL_DOT>;
  if (shmem > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  if (shmem > 64 * 1024) {
    hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (er != hipSuccess) return (int)er;
  }
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  if (blocks > 256 * 4) blocks = 256 * 4;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(PSDF_BLOCK), shmem, st, p, e, N, skip, packed, feat, Y);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // namespace

extern "C" {

// Y[dims[n_layers], N] = MLP(encode(positions)) in one launch (pos_dim 3, 2 features per level only).
//   dims[0] must equal the encoding's channel count: 2*(nr_levels + 2) for padded pseudo-levels (concat_points = 1),
//   2*nr_levels + 3 for appended points (2), 2*nr_levels without; `packed` comes from psdf_mlp_pack.
//   skip    optional [N] bytes: samples with skip != 0 may be left unevaluated (their Y entries are then untouched)
//   feat    optional [dims[0], N] feature-major: receives the encoding as a by-product (what psdf_encode_forward
//           would have written; rows of fully skipped tiles are untouched)
int psdf_encode_mlp_forward(int64_t N, int nr_levels, int capacity, const float* positions, const float* lattice,
                            const float* scale_factor, const float* shifts, const float* window, int concat_points,
                            float points_scaling, int n_layers, const int* dims, const float* packed,
                            const unsigned char* skip, float* feat, float* Y, void* stream) {
  MlpPlan p;
  int rc = make_plan(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (nr_levels <= 0 || capacity <= 0) return PSDF_ERR_ARG;
  const int Lt = nr_levels + (concat_points ? 2 : 0);
  const int C = concat_points == PSDF_ENC_CONCAT_APPEND ? 2 * nr_levels + 3 : 2 * Lt;   // the net's first layer has C inputs
  if (dims[0] != C || concat_points < 0 || concat_points > PSDF_ENC_CONCAT_APPEND) return PSDF_ERR_ARG;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !positions || !lattice || !scale_factor || !shifts || !window || !packed || !Y) return PSDF_ERR_ARG;
  if (n_layers != 3 && n_layers != 4) return PSDF_ERR_UNSUPPORTED;
  EncArgs e{nr_levels, Lt, (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window, points_scaling, C};
  hipStream_t st = (hipStream_t)stream;
  const int t1 = p.tiles[1], t2 = p.tiles[2], t3 = (n_layers == 4) ? p.tiles[3] : 0, to = p.tiles[n_layers];
#define CASE(A, B, C, O, D)                                          \
  if (t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) \
    return launch_fused<A, B, C, O, D>(p, e, N, skip, packed, feat, Y, st);
  CASE(2, 2, 2, 1, true)   // 64x3 -> 1..4   (BASELINE SDF net)
  CASE(1, 1, 1, 1, true)   // 32x3 -> 1..4   (reference SDF net, SDF channel only: sphere tracing)
  CASE(1, 1, 1, 2, false)  // 32x3 -> 33     (reference SDF net, models.py:153-161)
  CASE(2, 2, 2, 2, false)  // 64x3 -> 33
#undef CASE
  return PSDF_ERR_UNSUPPORTED;
}

}  // extern "C"
