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
    } else {  // [to][ti][r/4][lane][r%4]: one 128-bit read per lane = the A operands of 4 consecutive k-steps
      const int j = q & 3, lane = (q >> 2) & 63, rq = (q >> 8) & 3, pr = q >> 10;
      const int r = 4 * rq + j, ti = pr % p.tiles[l], to = pr / p.tiles[l];
      row = 32 * to + (lane & 31);
      col = 32 * ti + row_of(r, lane >> 5);
    }
    if (row < out_d && col < in_d) v = a.W[l][(int64_t)row * in_d + col];
  }
  packed[e] = v;
}

// Split-bf16 image (mlp_device.h): one thread per 32-bit word = two bf16 of one piece of one lane record, or one float
// of the tail.
__global__ void mlp_pack_split_kernel(PackArgs a, SplitPlan sp, float* __restrict__ packed, int f16) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const MlpPlan& p = a.plan;
  if (w >= sp.total_rec * 4) return;
  uint32_t* dst = reinterpret_cast<uint32_t*>(packed + sp.base);
  const int nl = p.n_layers;
  const int rec = w >> 2;
  if (rec >= sp.tail_rec) {
    const int f = w - sp.tail_rec * 4;
    float v = 0.f;
    for (int l = 0; l < nl; l++) {
      const bool dot = (l == nl - 1) && p.final_dot;
      const int nb = dot ? 4 : p.tiles[l + 1] * 32;
      if (f >= sp.b_off[l] && f < sp.b_off[l] + nb) {
        const int row = f - sp.b_off[l];
        if (row < p.dims[l + 1]) v = a.b[l][row];
      }
      if (dot && f >= sp.wf_off && f < sp.wf_off + p.dims[nl] * p.tiles[l] * 32) {
        const int q = f - sp.wf_off;  // [o][ti][r][h]
        const int h = q & 1, r = (q >> 1) & 15, ti = (q >> 5) % p.tiles[l], o = (q >> 5) / p.tiles[l];
        const int col = 32 * ti + row_of(r, h);
        if (col < p.dims[l]) v = a.W[l][(int64_t)o * p.dims[l] + col];
      }
    }
    dst[w] = __float_as_uint(v);
    return;
  }
  int l = 0;
  for (int i = 1; i < nl; i++) {
    const bool dot = (i == nl - 1) && p.final_dot;
    if (!dot && rec >= sp.w_rec[i]) l = i;
  }
  const int local = rec - sp.w_rec[l];
  const int lane = local & 63, piece = (local >> 6) % 3, s = (local / 192) % sp.ns[l], to = (local / 192) / sp.ns[l];
  const int row = 32 * to + (lane & 31), hh = lane >> 5;
  uint32_t out = 0;
  for (int e = 0; e < 2; e++) {
    const int j = 2 * (w & 3) + e;
    const int col = l == 0 ? 16 * s + 8 * hh + j : 32 * (s >> 1) + row_of(8 * (s & 1) + j, hh);
    float v = 0.f;
    if (row < p.dims[l + 1] && col < p.dims[l]) v = a.W[l][(int64_t)row * p.dims[l] + col];
    uint32_t pc[3];
    if (f16) split2h_bits(v, pc);      // two fp16 pieces (slot 2 unused)
    else split3(v, pc);
    out |= pc[piece] << (16 * e);
  }
  dst[w] = out;
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
                   const unsigned char* __restrict__ skip, float* __restrict__ Y) {
  extern __shared__ __align__(16) float lds[];
  for (int i = threadIdx.x; i < p.total; i += PSDF_BLOCK) lds[i] = packed[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: the tile loop is uniform)
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
    if (skip && __ballot(n < N && !skip[nc]) == 0) continue;  // every sample of the tile is masked
    // ---- layer 0: B operand from global (feature-major input)
    f32x16 h1[T1];
    init_bias<T1>(h1, lds + p.b_off[0], h);
    {
      const float* __restrict__ w0 = lds + p.w_off[0];
      const int steps = p.in_steps0;
      // all operand loads of a chunk are issued before the first MFMA consumes one (a load -> MFMA -> load chain
      // would expose one HBM latency per k-step)
      constexpr int XC = 16;
      for (int s0 = 0; s0 < steps; s0 += XC) {
        float xb[XC];
#pragma unroll
        for (int i = 0; i < XC; i++) {
          const int k = 2 * (s0 + i) + h;
          xb[i] = (s0 + i < steps && k < K0) ? X[(int64_t)k * N + nc] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < XC; i++) {
          if (s0 + i < steps) {
#pragma unroll
            for (int to = 0; to < T1; to++)
              h1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[(to * steps + s0 + i) * WS + lane], xb[i], h1[to], 0, 0, 0);
          }
        }
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

// The same evaluator on the bf16 matrix pipe with split fp32 operands (mlp_device.h, "split-bf16 operand path").
// CH: k-steps of layer 0 whose loads are issued together (= all of them when the input has <= 64 features).
// activation of the split forwards: packed arithmetic for the two-piece fp16 form (VALU-count bound), one element per
// instruction for the three-piece bf16 form (see apply_gelu_scalar / apply_gelu_packed in mlp_device.h)
#if !defined(PSDF_FWD_F16_GELU_PACKED)
#define PSDF_FWD_F16_GELU_PACKED 1
#endif
#define GELU_SPLIT(T_, H_)                                           \
  do {                                                               \
    if constexpr (F16 && PSDF_FWD_F16_GELU_PACKED) apply_gelu_packed<T_>(H_); \
    else apply_gelu_scalar<T_>(H_);                                  \
  } while (0)
template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, int CH, bool F16 = false>
__global__ void __launch_bounds__(PSDF_BLOCK, 2)
    mlp_fwd_split_kernel(MlpPlan p, SplitPlan sp, int64_t N, const float* __restrict__ X,
                         const float* __restrict__ packed, const unsigned char* __restrict__ skip,
                         float* __restrict__ Y) {
  extern __shared__ __align__(16) u32x4 simg[];
  {
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(packed + sp.base);
    for (int i = threadIdx.x; i < sp.total_rec; i += PSDF_BLOCK) simg[i] = src[i];
  }
  __syncthreads();
  const float* tail = reinterpret_cast<const float*>(simg + sp.tail_rec);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: the tile loop is uniform)
  const int h = lane >> 5, sl = lane & 31;
  const int K0 = p.dims[0];
  const int OUT = p.dims[p.n_layers];
  const int ns0 = sp.ns[0];
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");  // keeps the loop-invariant LDS reads inside the tile loop (see mlp_fwd_kernel)
    const int64_t n = tile * 32 + sl;
    const int64_t nc = n < N ? n : N - 1;
    if (skip && __ballot(n < N && !skip[nc]) == 0) continue;
    // ---- layer 0: lane (n, h) reads features 16 s + 8 h + j; CH k-steps of loads in flight before their MFMAs
    f32x16 h1[T1];
    init_bias4<T1>(h1, tail + sp.b_off[0], h);
    const float* __restrict__ xl = X + nc + (h ? (int64_t)8 * N : (int64_t)0);  // this lane's column, rows 8 h + ...
    for (int s0 = 0; s0 < ns0; s0 += CH) {
      float xs[CH][8];
#pragma unroll
      for (int i = 0; i < CH; i++) {
        const int sb = 16 * (s0 + i);  // wave-uniform
        if (sb + 16 <= K0) {           // whole k-step inside the input: row offsets are scalar, no guards
#pragma unroll
          for (int j = 0; j < 8; j++) xs[i][j] = xl[(int64_t)(sb + j) * N];
        } else if (sb < K0) {          // the partial last k-step: clamped address + select (no exec-mask branch per load)
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const int k = sb + 8 * h + j;
            const float v = X[(int64_t)(k < K0 ? k : K0 - 1) * N + nc];
            xs[i][j] = k < K0 ? v : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) xs[i][j] = 0.f;
        }
      }
      // branch-free: a k-step past the end multiplies zeros with the (finite) last weight record -- a guarded MFMA would
      // make the accumulators live across control flow (32 register copies per branch)
#pragma unroll
      for (int i = 0; i < CH; i++) {
        const int s = s0 + i < ns0 ? s0 + i : ns0 - 1;
        split_mac<T1, F16>(h1, xs[i], simg + sp.w_rec[0] + s * 192, ns0, lane);
      }
    }
    GELU_SPLIT(T1, h1);
    f32x16 h2[T2];
    init_bias4<T2>(h2, tail + sp.b_off[1], h);
    split_chain<T1, T2, F16>(h1, h2, simg + sp.w_rec[1], lane);
    GELU_SPLIT(T2, h2);
    constexpr int TL = (T3 > 0) ? T3 : T2;
    f32x16 hl[TL];
    if constexpr (T3 > 0) {
      init_bias4<T3>(hl, tail + sp.b_off[2], h);
      split_chain<T2, T3, F16>(h2, hl, simg + sp.w_rec[2], lane);
      GELU_SPLIT(T3, hl);
    } else {
#pragma unroll
      for (int t = 0; t < T2; t++) hl[t] = h2[t];
    }
    const int lf = p.n_layers - 1;
    if constexpr (FINAL_DOT) {
      const float* __restrict__ wf = tail + sp.wf_off;
      for (int o = 0; o < OUT; o++) {
        float acc = 0.f;
#pragma unroll
        for (int ti = 0; ti < TL; ti++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc = fmaf(wf[((o * TL + ti) * 16 + r) * 2 + h], hl[ti][r], acc);
        acc += __shfl_xor(acc, 32, 64);
        acc += tail[sp.b_off[lf] + o];
        if (h == 0 && n < N) Y[(int64_t)o * N + n] = acc;
      }
    } else {
      f32x16 y[OUT_T];
      init_bias4<OUT_T>(y, tail + sp.b_off[lf], h);
      split_chain<TL, OUT_T, F16>(hl, y, simg + sp.w_rec[lf], lane);
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

template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, int CH, bool F16 = false>
int launch_fwd_split_ch(const MlpPlan& p, const SplitPlan& sp, int64_t N, const float* X, const float* packed,
                     const unsigned char* skip, float* Y, hipStream_t st) {
  const size_t shmem = (size_t)sp.total_rec * 16;
  auto kern = mlp_fwd_split_kernel<T1, T2, T3, OUT_T, FINAL_DOT, CH, F16>;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = 256 * 4;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(PSDF_BLOCK), shmem, st, p, sp, N, X, packed, skip, Y);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT>
int launch_fwd_split(const MlpPlan& p, const SplitPlan& sp, int64_t N, const float* X, const float* packed,
                     const unsigned char* skip, float* Y, hipStream_t st, bool f16 = false) {
  if constexpr (T1 == 2 && T2 == 2 && T3 == 2 && OUT_T == 1 && FINAL_DOT) {
    // the two-piece fp16 arithmetic is instantiated for the BASELINE net only (64x3 -> 1..4 outputs, <= 64 inputs)
    if (f16) {
      switch (sp.ns[0]) {
        case 1: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 1, true>(p, sp, N, X, packed, skip, Y, st);
        case 2: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 2, true>(p, sp, N, X, packed, skip, Y, st);
        case 3: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 3, true>(p, sp, N, X, packed, skip, Y, st);
        default: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 4, true>(p, sp, N, X, packed, skip, Y, st);
      }
    }
  } else if (f16) {
    return PSDF_ERR_UNSUPPORTED;
  }
  switch (sp.ns[0]) {  // up to 64 input features: one chunk, no padded k-step
    case 1: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 1>(p, sp, N, X, packed, skip, Y, st);
    case 2: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 2>(p, sp, N, X, packed, skip, Y, st);
    case 3: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 3>(p, sp, N, X, packed, skip, Y, st);
    default: return launch_fwd_split_ch<T1, T2, T3, OUT_T, FINAL_DOT, 4>(p, sp, N, X, packed, skip, Y, st);
  }
}

template <int T1, int T2, int T3, int OUT_T, bool FINAL_DOT>
int launch_fwd(const MlpPlan& p, int64_t N, const float* X, const float* packed, const unsigned char* skip, float* Y,
               hipStream_t st) {
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
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(PSDF_BLOCK), shmem, st, p, N, X, packed, skip, Y);
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
  SplitPlan sp;
  make_split_plan(p, sp);
  return sp.ok ? (int64_t)sp.base + (int64_t)sp.total_rec * 4 : (int64_t)p.total;
}

// weights[l]: device pointer to torch-layout W_l [dims[l+1], dims[l]]; biases[l]: [dims[l+1]].
static int mlp_pack_impl(int n_layers, const int* dims, const float* const* weights, const float* const* biases,
                         float* packed, void* stream, int f16) {
  PackArgs a;
  int rc = make_plan(n_layers, dims, a.plan);
  if (rc != PSDF_OK) return rc;
  // the two-piece fp16 image exists for the net psdf_mlp_forward_f16 is instantiated for
  if (f16 && !(n_layers == 4 && dims[0] <= 64 && dims[1] == 64 && dims[2] == 64 && dims[3] == 64 && dims[4] <= 4))
    return PSDF_ERR_UNSUPPORTED;
  for (int l = 0; l < MAXL; l++) {
    a.W[l] = l < n_layers ? weights[l] : nullptr;
    a.b[l] = l < n_layers ? biases[l] : nullptr;
  }
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(psdf_blocks(a.plan.total, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     packed);
  PSDF_LAUNCH_CHECK();
  SplitPlan sp;
  make_split_plan(a.plan, sp);
  if (sp.ok) {  // second image of the same parameters: bf16 pieces in the operand order of mlp_fwd_split_kernel
    hipLaunchKernelGGL(mlp_pack_split_kernel, dim3(psdf_blocks(sp.total_rec * 4, 256)), dim3(256), 0,
                       (hipStream_t)stream, a, sp, packed, f16);
    PSDF_LAUNCH_CHECK();
  } else if (f16) {
    return PSDF_ERR_UNSUPPORTED;
  }
  return PSDF_OK;
}

int psdf_mlp_pack(int n_layers, const int* dims, const float* const* weights, const float* const* biases,
                  float* packed, void* stream) {
  return mlp_pack_impl(n_layers, dims, weights, biases, packed, stream, 0);
}

// The same buffer with the split image holding TWO fp16 pieces per weight: what psdf_mlp_forward_f16 consumes (and only it).
int psdf_mlp_pack_f16(int n_layers, const int* dims, const float* const* weights, const float* const* biases,
                      float* packed, void* stream) {
  return mlp_pack_impl(n_layers, dims, weights, biases, packed, stream, 1);
}

// X: [dims[0], N] feature-major; Y: [dims[n_layers], N] feature-major.  GELU (erf) after every layer but the last.
static int mlp_forward_impl(int n_layers, const int* dims, int64_t N, const float* X, const float* packed,
                            const unsigned char* skip, float* Y, void* stream, bool f16 = false) {
  MlpPlan p;
  int rc = make_plan(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !X || !packed || !Y) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int t1 = p.tiles[1], t2 = p.tiles[2], t3 = (n_layers == 4) ? p.tiles[3] : 0, to = p.tiles[n_layers];
  if (n_layers != 3 && n_layers != 4) return PSDF_ERR_UNSUPPORTED;
  SplitPlan sp;
  make_split_plan(p, sp);
  // S: shapes whose split-bf16 image can fit SPLIT_LDS_MAX (the wider nets never do, so that kernel is not built for them)
#define CASE(A, B, C, O, D, S)                                                   \
  if (t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) {           \
    if constexpr (S) {                                                          \
      if (sp.ok) {                                                              \
        psdf::g_last_path[psdf::PATH_MLP_FWD] = f16 ? 3 : 2;                    \
        return launch_fwd_split<A, B, C, O, D>(p, sp, N, X, packed, skip, Y, st, f16); \
      }                                                                         \
    }                                                                           \
    if (f16) return PSDF_ERR_UNSUPPORTED;                                       \
    psdf::g_last_path[psdf::PATH_MLP_FWD] = 1;                                  \
    return launch_fwd<A, B, C, O, D>(p, N, X, packed, skip, Y, st);             \
  }
  CASE(2, 2, 2, 1, true, true)    // 64x3 -> 1..4      (BASELINE SDF net)
  CASE(1, 1, 1, 1, true, true)    // 32x3 -> 1..4
  CASE(1, 1, 1, 2, false, true)   // 32x3 -> 33        (reference SDF net, models.py:153-161)
  CASE(2, 2, 2, 3, false, false)  // 64x3 -> 65        (background density+feature net, models.py:451-459)
  CASE(2, 2, 2, 2, false, false)  // 64x3 -> 33
  CASE(2, 2, 0, 1, true, true)    // 64x2 -> 3         (background colour head, models.py:463-469)
  CASE(4, 4, 2, 1, true, false)   // 128,128,64 -> 3   (colour net, models.py:350)
#undef CASE
  return PSDF_ERR_UNSUPPORTED;
}


int psdf_mlp_forward(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, float* Y,
                     void* stream) {
  return mlp_forward_impl(n_layers, dims, N, X, packed, nullptr, Y, stream);
}

// The same evaluation with TWO fp16 pieces per fp32 operand (three products) for a buffer made by psdf_mlp_pack_f16: the
// BASELINE net only (dims = {<= 64, 64, 64, 64, <= 4}; -2 otherwise).  Opt-in: max error ~3e-6 of the largest output instead of
// ~1e-6, and inputs / activations / weights must stay below 65504 in magnitude (the three-piece bf16 form has no such limit).
int psdf_mlp_forward_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, float* Y,
                         void* stream) {
  return mlp_forward_impl(n_layers, dims, N, X, packed, nullptr, Y, stream, true);
}

// Same with a per-sample mask: 32-sample tiles whose samples all have skip[n] != 0 are not evaluated (their Y entries
// are left as they are); partially masked tiles are evaluated in full.
int psdf_mlp_forward_masked(int n_layers, const int* dims, int64_t N, const float* X, const float* packed,
                            const unsigned char* skip, float* Y, void* stream) {
  return mlp_forward_impl(n_layers, dims, N, X, packed, skip, Y, stream);
}

}  // extern "C"
