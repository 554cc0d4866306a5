// Fused small-MLP evaluator for gfx950 (fp32 MFMA, LDS-staged weight tiles).
// Replaces the torch.nn.Sequential(Linear, GELU, ...) evaluators of the reference
// (permuto_sdf_py/models/models.py:153-161 SDF net, :451-470 background nets, :54-129,350 colour net,
// and the BASELINE 64x3 SDF variant).  The reference has no native MLP code; the oracle is torch.nn fp32.
//
// Formulation: everything is computed TRANSPOSED, Z^T[out x samples] = W[out x in] * H^T[in x samples],
// with v_mfma_f32_32x32x2_f32.  One wave owns a tile of 32 samples.
//   A operand (1 VGPR): lane l holds W[32*to + (l&31)][k(l>>5)]         -> read from LDS, pre-packed
//   B operand (1 VGPR): lane l holds H^T[k(l>>5)][sample (l&31)]
//   D (16 VGPRs):       lane l, reg r holds Z^T[32*to + (r&3)+8*(r>>2)+4*(l>>5)][sample (l&31)]
// Because a dot product may visit k in any order as long as A and B agree, the k-step "(tile ti, reg r)"
// is defined as the neuron pair { 32*ti + (r&3)+8*(r>>2) + 4*h : h = 0,1 }: then register r of the previous
// layer's D tile IS the B operand of that k-step.  Activations never leave registers and need no
// cross-lane movement between layers; only the weights are permuted (once, by psdf_mlp_pack).
// The first layer reads its B operand straight from the feature-major encoding output [K0, N]
// (two 128-B segments per wave load); the result is stored feature-major [OUT, N].
#include "psdf_common.h"

#include "mlp_device.h"

namespace {

// ---------------------------------------------------------------------------------------- packing
// One thread per packed float.  W_l is torch layout [dims[l+1]][dims[l]] row major.
struct PackArgs {
  MlpPlan plan;
  const float* W[MAXL];
  const float* b[MAXL];
};

__global__ void mlp_pack_kernel(PackArgs a, float* __restrict__ packed) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const MlpPlan& p = a.plan;
  if (e >= p.total) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < MAXL; i++)
    if (i < p.n_layers && e >= p.w_off[i]) l = i;
  const int out_d = p.dims[l + 1], in_d = p.dims[l];
  const bool last = (l == p.n_layers - 1);
  float v = 0.f;
  if (e >= p.b_off[l]) {
    const int row = e - p.b_off[l];
    if (row < out_d) v = a.b[l][row];
  } else {
    const int q = e - p.w_off[l];
    int row, col;
    if (l == 0) {  // [to][s][lane], row stride WS
      const int lane = q % WS, s = (q / WS) % p.in_steps0, to = (q / WS) / p.in_steps0;
      row = (lane < 64) ? 32 * to + (lane & 31) : (1 << 20);
      col = 2 * s + (lane >> 5);
    } else if (last && p.final_dot) {  // [o][ti][r][h]
      const int h = q & 1, r = (q >> 1) & 15, ti = (q >> 5) % p.tiles[l], o = (q >> 5) / p.tiles[l];
      row = o;
      col = 32 * ti + row_of(r, h);
    } else {  // [to][ti][r][lane], row stride WS
      const int lane = q % WS, r = (q / WS) & 15, ti = (q / (WS * 16)) % p.tiles[l], to = (q / (WS * 16)) / p.tiles[l];
      row = (lane < 64) ? 32 * to + (lane & 31) : (1 << 20);
      col = 32 * ti + row_of(r, lane >> 5);
    }
    if (row < out_d && col < in_d) v = a.W[l][(int64_t)row * in_d + col];
  }
  packed[e] = v;
}

// Forward of the whole net for one 32-sample tile; fills the hidden pre-activation free result in `hid`.
// T1,T2,T3: hidden widths in tiles of 32 (T3 == 0: only two hidden layers).
template <int T1, int T2, int T3>
struct Net {
  static constexpr int NH = (T3 > 0) ? 3 : 2;
  static constexpr int TL = (T3 > 0) ? T3 : T2;  // tiles of the last hidden layer
};

template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT>
__global__ void __launch_bounds__(PSDF_BLOCK, 2)
    mlp_fwd_kernel(MlpPlan p, int64_t N, const float* __restrict__ X, const float* __restrict__ packed,
                   float* __restrict__ Y) {
  extern __shared__ __align__(16) float lds[];
  for (int i = threadIdx.x; i < p.total; i += PSDF_BLOCK) lds[i] = packed[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, sl = lane & 31;
  const int K0 = p.dims[0];
  const int OUT = p.dims[p.n_layers];
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    // compiler-only barrier: keeps the (loop-invariant) LDS weight reads inside the tile loop; without it
    // LICM hoists hundreds of them into VGPRs and the kernel spills
    asm volatile("" ::: "memory");
    const int64_t n = tile * 32 + sl;
    const int64_t nc = n < N ? n : N - 1;
    // ---- layer 0: B operand from global (feature-major input)
    f32x16 h1[T1];
    init_bias<T1>(h1, lds + p.b_off[0], h);
    {
      const float* __restrict__ w0 = lds + p.w_off[0];
      const int steps = p.in_steps0;
      for (int s = 0; s < steps; s++) {
        const int k = 2 * s + h;
        const float b = (k < K0) ? X[(int64_t)k * N + nc] : 0.f;
#pragma unroll
        for (int to = 0; to < T1; to++)
          h1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[(to * steps + s) * WS + lane], b, h1[to], 0, 0, 0);
      }
    }
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
        if (h == 0 && n < N) Y[(int64_t)o * N + n] = acc;
      }
    } else {
      f32x16 y[OUT_T];
      init_bias<OUT_T>(y, lds + p.b_off[lf], h);
      dense_chain<TL, OUT_T>(hl, y, lds + p.w_off[lf], lane);
      if (n < N) {
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
int launch_fwd(const MlpPlan& p, int64_t N, const float* X, const float* packed, float* Y, hipStream_t st) {
  const size_t shmem = (size_t)p.total * sizeof(float);
  auto kern = mlp_fwd_kernel<T1, T2, T3, OUT_T, FINAL_DOT>;
  if (shmem > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = 256 * 4;  // persistent-ish: the weight staging is amortised over many tiles
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(PSDF_BLOCK), shmem, st, p, N, X, packed, Y);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}


// ======================================================================================== backward
// One kernel does the whole backward of the net for a 32-sample tile per wave:
//   1. recompute the forward, keeping the pre-activations z_l in registers (transposed D layout);
//   2. data-gradient chain dH_{l-1}^T = W_l^T dZ_l^T with the SAME LDS weight image (transposed read,
//      2-way bank conflict thanks to the 65-float row stride) -- again no cross-lane movement;
//   3. weight gradients dW_l = dZ_l^T H_{l-1}: an MFMA whose k dimension is the SAMPLE index.  In the
//      non-transposed D layout (lane = neuron, register = sample) a tile is directly a valid A or B operand
//      for that product, so dZ_l and H_{l-1} are transposed once through a 32x33-float per-wave LDS buffer;
//   4. dW/db are accumulated with LDS atomics into a workgroup-private image that mirrors the packed
//      parameter layout, and flushed with one global atomic per parameter per workgroup at the end.
struct GradPtrs {
  float* dW[MAXL];
  float* db[MAXL];
};

__device__ __forceinline__ int reg_of(int c) { return (c & 3) + 4 * (c >> 3); }  // inverse of row_of: c = row_of(r,h)
__device__ __forceinline__ int half_of(int c) { return (c >> 2) & 1; }

// in (lane = sample, reg = neuron row_of(r,h))  ->  out (lane = neuron sl, reg = sample row_of(r,h))
__device__ __forceinline__ void transpose_tile(const f32x16& in, f32x16& out, float* __restrict__ buf, int sl, int h) {
#pragma unroll
  for (int r = 0; r < 16; r++) buf[row_of(r, h) * 33 + sl] = in[r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; r++) out[r] = buf[sl * 33 + row_of(r, h)];
  __builtin_amdgcn_wave_barrier();
}

// dh^T[in x s] += W^T dz^T  for a chain layer (weights [to][ti][r][lane], stride WS)
template <int TO, int TI>
__device__ __forceinline__ void dense_chain_T(const f32x16 (&dz)[TO], f32x16 (&dh)[TI], const float* __restrict__ w_lds,
                                              int sl, int hl) {
  const int lane_off = reg_of(sl) * WS + half_of(sl) * 32 + 4 * hl;
#pragma unroll
  for (int to = 0; to < TO; to++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float b = dz[to][r];
#pragma unroll
      for (int ti = 0; ti < TI; ti++) {
        const float a = w_lds[(to * TI + ti) * 16 * WS + lane_off + row_of(r, 0)];
        dh[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, dh[ti], 0, 0, 0);
      }
    }
}

// dW[to][ti] += dz_nt[to] (A: lane = out neuron, k = sample) x hin_nt[ti] (B: k = sample, lane = in neuron).
// ds_add_f32 costs ~196 cycles per wave instruction on this chip whatever the access pattern (tools/atomic_bench.hip:
// LDS float atomics retire ~1 lane per 3 cycles), against 8-15 cycles for a plain LDS read or read-modify-write, and
// a 64x3 net needs >200 of them per tile: the first version of this kernel spent 90 % of its time there.  So the
// partial tiles of the workgroup's waves are exchanged through a staging area instead: every wave writes its 32x32
// partial (16 registers) to stage[wave], and after a barrier ONE owner wave (pair index mod #waves) sums the staged
// partials and adds them to the accumulator image with plain read-modify-writes (it is the only writer of that pair).
// All waves of the workgroup call this function together (the tile loop is workgroup-uniform).
template <int TO, int TI, bool LAYER0>
__device__ __forceinline__ void weight_grads(const f32x16 (&dz_nt)[TO], const f32x16 (&hin_nt)[TI],
                                             float* __restrict__ acc_w, float* __restrict__ acc_b, int steps0,
                                             int sl, int hl, float* __restrict__ stage, int wave, int nwaves) {
  const int lane = hl * 32 + sl;
  __syncthreads();  // the staging area aliases every wave's transpose buffer: wait until all of them are done with it
#pragma unroll
  for (int to = 0; to < TO; to++) {
#pragma unroll
    for (int ti = 0; ti < TI; ti++) {
      f32x16 d;
#pragma unroll
      for (int q = 0; q < 16; q++) d[q] = 0.f;
#pragma unroll
      for (int q = 0; q < 16; q++) d = __builtin_amdgcn_mfma_f32_32x32x2f32(dz_nt[to][q], hin_nt[ti][q], d, 0, 0, 0);
      // d[q] at lane (sl,hl) = dW[32to + row_of(q,hl)][32ti + sl]
#pragma unroll
      for (int q = 0; q < 16; q++) stage[(wave * 16 + q) * 64 + lane] = d[q];
      __syncthreads();
      if (wave == (to * TI + ti) % nwaves) {
        float* base;
        bool ok = true;
        if (LAYER0) {
          const int col = 32 * ti + sl;
          ok = (col >> 1) < steps0;
          base = acc_w + (to * steps0 + (col >> 1)) * WS + (col & 1) * 32 + 4 * hl;
        } else {
          base = acc_w + ((to * TI + ti) * 16 + reg_of(sl)) * WS + half_of(sl) * 32 + 4 * hl;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
          float sum = stage[q * 64 + lane];
          for (int w = 1; w < nwaves; w++) sum += stage[(w * 16 + q) * 64 + lane];
          if (ok) base[row_of(q, 0)] += sum;
        }
      }
      __syncthreads();
    }
    // bias: sum over the 32 samples of this tile (16 regs here + the other half-wave); 2 atomics per layer and tile
    float sb = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) sb += dz_nt[to][q];
    sb += __shfl_xor(sb, 32, 64);
    if (hl == 0) atomicAdd(acc_b + 32 * to + sl, sb);
  }
}

template <int T>
__device__ __forceinline__ void mul_gelu_grad(f32x16 (&dh)[T], const f32x16 (&z)[T]) {
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) dh[t][r] = dh[t][r] * gelu_grad(z[t][r]);
}
template <int T>
__device__ __forceinline__ void zero_tiles(f32x16 (&a)[T]) {
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) a[t][r] = 0.f;
}
template <int T>
__device__ __forceinline__ void gelu_and_transpose(const f32x16 (&z)[T], f32x16 (&out)[T], float* buf, int sl, int hl) {
#pragma unroll
  for (int t = 0; t < T; t++) {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = gelu_exact(z[t][r]);
    transpose_tile(a, out[t], buf, sl, hl);
  }
}
template <int T>
__device__ __forceinline__ void transpose_tiles(const f32x16 (&in)[T], f32x16 (&out)[T], float* buf, int sl, int hl) {
#pragma unroll
  for (int t = 0; t < T; t++) transpose_tile(in[t], out[t], buf, sl, hl);
}

constexpr int BWD_WAVES = 4;

template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, bool NEED_DX>
__global__ void __launch_bounds__(BWD_WAVES * 64)
    mlp_bwd_kernel(MlpPlan p, int64_t N, const float* __restrict__ X, const float* __restrict__ packed,
                   const float* __restrict__ dY, float* __restrict__ dX, GradPtrs gp) {
  extern __shared__ __align__(16) float lds[];
  float* __restrict__ W = lds;
  float* __restrict__ ACC = lds + p.total;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* __restrict__ tbuf = lds + 2 * p.total + wave * (32 * 33 + 32);
  float* __restrict__ dyb = tbuf + 32 * 33;
  // [wave][16][64] partial dW tiles; aliases the transpose buffers, which are dead while weight_grads runs (its
  // operands are in registers and it ends on a barrier)
  float* __restrict__ stage = lds + 2 * p.total;
  static_assert(BWD_WAVES * (32 * 33 + 32) >= BWD_WAVES * 16 * 64, "stage must fit in the transpose buffers");
  for (int i = threadIdx.x; i < p.total; i += BWD_WAVES * 64) {
    W[i] = packed[i];
    ACC[i] = 0.f;
  }
  __syncthreads();
  const int hl = lane >> 5, sl = lane & 31;
  const int K0 = p.dims[0], OUT = p.dims[p.n_layers], S0 = p.in_steps0;
  constexpr int TL = (T3 > 0) ? T3 : T2;
  const int lf = p.n_layers - 1;
  const int64_t ntiles = (N + 31) / 32;
  // workgroup-uniform trip count: the waves meet at barriers inside weight_grads (a wave without a tile runs on zeros)
  for (int64_t tbase_ = (int64_t)blockIdx.x * BWD_WAVES; tbase_ < ntiles; tbase_ += (int64_t)gridDim.x * BWD_WAVES) {
    asm volatile("" ::: "memory");
    const int64_t tile = tbase_ + wave;
    const int64_t n = tile * 32 + sl;
    const bool live = n < N;
    const int64_t nc = live ? n : N - 1;
    // ------------------------------------------------ forward (keep pre-activations)
    f32x16 xb[TI0];
#pragma unroll
    for (int t = 0; t < TI0; t++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int k = 32 * t + 2 * q + hl;
        xb[t][q] = (k < K0 && live) ? X[(int64_t)k * N + nc] : 0.f;
      }
    f32x16 z1[T1];
    init_bias<T1>(z1, W + p.b_off[0], hl);
#pragma unroll
    for (int t = 0; t < TI0; t++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int s = 16 * t + q;
        if (s < S0) {
#pragma unroll
          for (int to = 0; to < T1; to++)
            z1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[p.w_off[0] + (to * S0 + s) * WS + lane], xb[t][q], z1[to], 0, 0, 0);
        }
      }
    f32x16 z2[T2];
    {
      f32x16 a1[T1];
#pragma unroll
      for (int t = 0; t < T1; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) a1[t][r] = gelu_exact(z1[t][r]);
      init_bias<T2>(z2, W + p.b_off[1], hl);
      dense_chain<T1, T2>(a1, z2, W + p.w_off[1], lane);
    }
    f32x16 z3[TL];
    if constexpr (T3 > 0) {
      f32x16 a2[T2];
#pragma unroll
      for (int t = 0; t < T2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) a2[t][r] = gelu_exact(z2[t][r]);
      init_bias<T3>(z3, W + p.b_off[2], hl);
      dense_chain<T2, T3>(a2, z3, W + p.w_off[2], lane);
    }
    // ------------------------------------------------ output layer
    f32x16 dhl[TL];  // gradient wrt the last hidden ACTIVATION
    zero_tiles<TL>(dhl);
    {
      f32x16 hl_nt[TL];
      if constexpr (T3 > 0)
        gelu_and_transpose<TL>(z3, hl_nt, tbuf, sl, hl);
      else
        gelu_and_transpose<TL>(z2, hl_nt, tbuf, sl, hl);
      if constexpr (FINAL_DOT) {
        const float* __restrict__ wf = W + p.w_off[lf];
        for (int o = 0; o < OUT; o++) {
          const float dy = live ? dY[(int64_t)o * N + n] : 0.f;
#pragma unroll
          for (int ti = 0; ti < TL; ti++)
#pragma unroll
            for (int r = 0; r < 16; r++) dhl[ti][r] = fmaf(wf[((o * TL + ti) * 16 + r) * 2 + hl], dy, dhl[ti][r]);
          if (hl == 0) dyb[sl] = dy;
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int ti = 0; ti < TL; ti++) {
            float pr = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) pr = fmaf(hl_nt[ti][r], dyb[row_of(r, hl)], pr);
            pr += __shfl_xor(pr, 32, 64);
            if (hl == 0) atomicAdd(ACC + p.w_off[lf] + ((o * TL + ti) * 16 + reg_of(sl)) * 2 + half_of(sl), pr);
          }
          float sb = (hl == 0) ? dy : 0.f;
          sb = psdf::wave_sum(sb);
          if (lane == 0) atomicAdd(ACC + p.b_off[lf] + o, sb);
          __builtin_amdgcn_wave_barrier();
        }
      } else {
        f32x16 dyT[OUT_T];
#pragma unroll
        for (int to = 0; to < OUT_T; to++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = 32 * to + row_of(r, hl);
            dyT[to][r] = (row < OUT && live) ? dY[(int64_t)row * N + n] : 0.f;
          }
        dense_chain_T<OUT_T, TL>(dyT, dhl, W + p.w_off[lf], sl, hl);
        f32x16 dy_nt[OUT_T];
        transpose_tiles<OUT_T>(dyT, dy_nt, tbuf, sl, hl);
        weight_grads<OUT_T, TL, false>(dy_nt, hl_nt, ACC + p.w_off[lf], ACC + p.b_off[lf], S0, sl, hl, stage, wave, BWD_WAVES);
      }
    }
    // ------------------------------------------------ hidden layers, last to first
    f32x16 dh2[T2];
    if constexpr (T3 > 0) {
      mul_gelu_grad<T3>(dhl, z3);  // now dZ3
      {
        f32x16 dz_nt[T3], hin_nt[T2];
        transpose_tiles<T3>(dhl, dz_nt, tbuf, sl, hl);
        gelu_and_transpose<T2>(z2, hin_nt, tbuf, sl, hl);
        weight_grads<T3, T2, false>(dz_nt, hin_nt, ACC + p.w_off[2], ACC + p.b_off[2], S0, sl, hl, stage, wave, BWD_WAVES);
      }
      zero_tiles<T2>(dh2);
      dense_chain_T<T3, T2>(dhl, dh2, W + p.w_off[2], sl, hl);
    } else {
#pragma unroll
      for (int t = 0; t < T2; t++) dh2[t] = dhl[t];
    }
    mul_gelu_grad<T2>(dh2, z2);  // dZ2
    {
      f32x16 dz_nt[T2], hin_nt[T1];
      transpose_tiles<T2>(dh2, dz_nt, tbuf, sl, hl);
      gelu_and_transpose<T1>(z1, hin_nt, tbuf, sl, hl);
      weight_grads<T2, T1, false>(dz_nt, hin_nt, ACC + p.w_off[1], ACC + p.b_off[1], S0, sl, hl, stage, wave, BWD_WAVES);
    }
    f32x16 dh1[T1];
    zero_tiles<T1>(dh1);
    dense_chain_T<T2, T1>(dh2, dh1, W + p.w_off[1], sl, hl);
    mul_gelu_grad<T1>(dh1, z1);  // dZ1
    {
      f32x16 dz_nt[T1], x_nt[TI0];
      transpose_tiles<T1>(dh1, dz_nt, tbuf, sl, hl);
      // xb[t][q] holds feature 32t + 2q + hl of sample sl: write as [feature][sample], read [sample regs] per feature lane
#pragma unroll
      for (int t = 0; t < TI0; t++) {
#pragma unroll
        for (int q = 0; q < 16; q++) tbuf[(2 * q + hl) * 33 + sl] = xb[t][q];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; r++) x_nt[t][r] = tbuf[sl * 33 + row_of(r, hl)];
        __builtin_amdgcn_wave_barrier();
      }
      weight_grads<T1, TI0, true>(dz_nt, x_nt, ACC + p.w_off[0], ACC + p.b_off[0], S0, sl, hl, stage, wave, BWD_WAVES);
    }
    if constexpr (NEED_DX) {
      // dX^T[k][s] = sum_out W0[out][k] dZ1[out][s];  A operand read transposed from the layer-0 image
#pragma unroll
      for (int t = 0; t < TI0; t++) {
        f32x16 dx;
#pragma unroll
        for (int q = 0; q < 16; q++) dx[q] = 0.f;
        const int col = 32 * t + sl;
        const bool colok = (col >> 1) < S0;
#pragma unroll
        for (int to = 0; to < T1; to++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float a = colok ? W[p.w_off[0] + (to * S0 + (col >> 1)) * WS + (col & 1) * 32 + row_of(r, hl)] : 0.f;
            dx = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dh1[to][r], dx, 0, 0, 0);
          }
        if (live) {
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const int k = 32 * t + row_of(q, hl);
            if (k < K0) dX[(int64_t)k * N + n] = dx[q];
          }
        }
      }
    }
  }
  // ------------------------------------------------ flush the workgroup's gradient image
  __syncthreads();
  for (int e = threadIdx.x; e < p.total; e += BWD_WAVES * 64) {
    const float v = ACC[e];
    if (v == 0.f) continue;
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAXL; i++)
      if (i < p.n_layers && e >= p.w_off[i]) l = i;
    const int out_d = p.dims[l + 1], in_d = p.dims[l];
    const bool last = (l == p.n_layers - 1);
    if (e >= p.b_off[l]) {
      const int row = e - p.b_off[l];
      if (row < out_d) atomicAdd(gp.db[l] + row, v);
    } else {
      const int q = e - p.w_off[l];
      int row, col;
      if (l == 0) {
        const int ln = q % WS, s = (q / WS) % p.in_steps0, to = (q / WS) / p.in_steps0;
        row = (ln < 64) ? 32 * to + (ln & 31) : (1 << 20);
        col = 2 * s + (ln >> 5);
      } else if (last && p.final_dot) {
        const int h = q & 1, r = (q >> 1) & 15, ti = (q >> 5) % p.tiles[l], o = (q >> 5) / p.tiles[l];
        row = o;
        col = 32 * ti + row_of(r, h);
      } else {
        const int ln = q % WS, r = (q / WS) & 15, ti = (q / (WS * 16)) % p.tiles[l], to = (q / (WS * 16)) / p.tiles[l];
        row = (ln < 64) ? 32 * to + (ln & 31) : (1 << 20);
        col = 32 * ti + row_of(r, ln >> 5);
      }
      if (row < out_d && col < in_d) atomicAdd(gp.dW[l] + (int64_t)row * in_d + col, v);
    }
  }
}

template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT>
int launch_bwd(const MlpPlan& p, int64_t N, const float* X, const float* packed, const float* dY, float* dX,
               const GradPtrs& gp, hipStream_t st) {
  const size_t shmem = ((size_t)2 * p.total + BWD_WAVES * (32 * 33 + 32)) * sizeof(float);
  if (shmem > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + BWD_WAVES - 1) / BWD_WAVES;
  if (blocks > 256) blocks = 256;  // one resident workgroup per CU (LDS bound); each walks many tiles
#define GO(DX)                                                                                                    \
  do {                                                                                                            \
    auto kern = mlp_bwd_kernel<TI0, T1, T2, T3, OUT_T, FINAL_DOT, DX>;                                             \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    if (e != hipSuccess) return (int)e;                                                                           \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BWD_WAVES * 64), shmem, st, p, N, X, packed, dY, dX, gp);  \
  } while (0)
  if (dX)
    GO(true);
  else
    GO(false);
#undef GO
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // namespace

extern "C" {

// Number of floats of the packed (MFMA-operand ordered) parameter buffer for a net with the given
// layer widths: dims[0]=input .. dims[n_layers]=output.  Returns <0 on argument error.
int64_t psdf_mlp_packed_size(int n_layers, const int* dims) {
  MlpPlan p;
  if (make_plan(n_layers, dims, p) != PSDF_OK) return PSDF_ERR_ARG;
  return p.total;
}

// weights[l]: device pointer to torch-layout W_l [dims[l+1], dims[l]]; biases[l]: [dims[l+1]].
int psdf_mlp_pack(int n_layers, const int* dims, const float* const* weights, const float* const* biases,
                  float* packed, void* stream) {
  PackArgs a;
  int rc = make_plan(n_layers, dims, a.plan);
  if (rc != PSDF_OK) return rc;
  for (int l = 0; l < MAXL; l++) {
    a.W[l] = l < n_layers ? weights[l] : nullptr;
    a.b[l] = l < n_layers ? biases[l] : nullptr;
  }
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(psdf_blocks(a.plan.total, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     packed);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// X: [dims[0], N] feature-major; Y: [dims[n_layers], N] feature-major.  GELU (erf) after every layer but the last.
int psdf_mlp_forward(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, float* Y,
                     void* stream) {
  MlpPlan p;
  int rc = make_plan(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !X || !packed || !Y) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int t1 = p.tiles[1], t2 = p.tiles[2], t3 = (n_layers == 4) ? p.tiles[3] : 0, to = p.tiles[n_layers];
  if (n_layers != 3 && n_layers != 4) return PSDF_ERR_UNSUPPORTED;
#define CASE(A, B, C, O, D)                                          \
  if (t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) \
    return launch_fwd<A, B, C, O, D>(p, N, X, packed, Y, st);
  CASE(2, 2, 2, 1, true)   // 64x3 -> 1..4      (BASELINE SDF net)
  CASE(1, 1, 1, 1, true)   // 32x3 -> 1..4
  CASE(1, 1, 1, 2, false)  // 32x3 -> 33        (reference SDF net, models.py:153-161)
  CASE(2, 2, 2, 3, false)  // 64x3 -> 65        (background density+feature net, models.py:451-459)
  CASE(2, 2, 2, 2, false)  // 64x3 -> 33
  CASE(2, 2, 0, 1, true)   // 64x2 -> 3         (background colour head, models.py:463-469)
  CASE(4, 4, 2, 1, true)   // 128,128,64 -> 3   (colour net, models.py:350)
#undef CASE
  return PSDF_ERR_UNSUPPORTED;
}

// Backward of psdf_mlp_forward.  dY [dims[n_layers], N] feature-major; dX [dims[0], N] feature-major or NULL;
// dW[l] (torch layout [dims[l+1], dims[l]]) and db[l] are ACCUMULATED INTO (caller zero-fills).
int psdf_mlp_backward(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, const float* dY,
                      float* dX, float* const* dW, float* const* db, void* stream) {
  MlpPlan p;
  int rc = make_plan(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !X || !packed || !dY || !dW || !db) return PSDF_ERR_ARG;
  if (n_layers != 3 && n_layers != 4) return PSDF_ERR_UNSUPPORTED;
  GradPtrs gp;
  for (int l = 0; l < MAXL; l++) {
    gp.dW[l] = l < n_layers ? dW[l] : nullptr;
    gp.db[l] = l < n_layers ? db[l] : nullptr;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ti0 = (dims[0] + 31) / 32;
  const int t1 = p.tiles[1], t2 = p.tiles[2], t3 = (n_layers == 4) ? p.tiles[3] : 0, to = p.tiles[n_layers];
#define CASE(I, A, B, C, O, D)                                                   \
  if (ti0 == I && t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) \
    return launch_bwd<I, A, B, C, O, D>(p, N, X, packed, dY, dX, gp, st);
  CASE(2, 2, 2, 2, 1, true)
  CASE(2, 1, 1, 1, 1, true)
  CASE(2, 1, 1, 1, 2, false)
  CASE(2, 2, 2, 2, 3, false)
  CASE(2, 2, 2, 2, 2, false)
  CASE(3, 2, 2, 0, 1, true)
  CASE(1, 2, 2, 2, 1, true)   // <=32 input channels (small encodings, e.g. 8 levels + points)
  CASE(1, 1, 1, 1, 1, true)
#undef CASE
  return PSDF_ERR_UNSUPPORTED;
}

}  // extern "C"
