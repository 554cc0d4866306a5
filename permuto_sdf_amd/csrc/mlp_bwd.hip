// Fused small-MLP BACKWARD for gfx950: dX, dW_l, db_l of Linear/GELU stacks in ONE kernel.
// Autograd backward of the reference's torch.nn.Sequential(Linear, GELU, ...) evaluators
// (permuto_sdf_py/models/models.py:153-161 SDF net, :451-470 background nets) and of the BASELINE 64x3 net.
//
// Design (third version; the two earlier ones are kept under attic/rejected/ with their measurements):
//   * 16-sample tiles, v_mfma_f32_16x16x4_f32.  Same FLOP/cycle as the 32x32x2 instruction, but the per-wave state of
//     a tile (activations and their derivatives of three 64-wide layers) is 96 registers instead of 192.
//   * The weight gradient dW_l = dZ_l^T H_{l-1} is a product whose k dimension is the SAMPLE index, so its MFMA
//     accumulators can simply stay in registers for the whole life of the wave: every tile adds into them with the
//     accumulate operand of the instruction.  176 registers for the 36-64-64-64-1 net; with the tile state that fits
//     the 512-register file at one wave per SIMD without scratch.  Nothing is exchanged between waves in the tile
//     loop: no barriers, no LDS atomics (ds_add_f32 costs ~196 cycles per wave instruction on this chip,
//     tools/atomic_bench.hip -- the first version spent 90 % of its time there), no staging.
//   * Each activation is visited ONCE: gelu(z) and gelu'(z) come out of one erf + one exp evaluation and are kept;
//     z itself is dropped.
//   * The forward is recomputed from X (the encoding output) -- nothing but X is saved by the forward pass.
//   * The layer-0 operand of the next tile is prefetched with global_load_lds; weight operands are read 128 bits at a
//     time (one ds_read_b128 = the A operands of four MFMAs).
//   * At the end every wave adds its accumulators into a workgroup image in LDS (once per launch, so the slow LDS
//     atomics do not matter) and the workgroup flushes that image with one global atomic per parameter.
//
// Layouts (lane l: g = l>>4, c = l&15):
//   MFMA 16x16x4:  A[i=c][k=g],  B[k=g][j=c],  D reg r = D[row 4g+r][col c]
//   "T" tile  (chain layout):  lane (g, c=sample), reg r = value(neuron 16t+4g+r, sample c)
//             -> register r of a T tile IS the B operand of k-step (t, r) of the next layer: no data movement;
//   "NT" tile (lane = neuron): lane (g, c=neuron), reg r = value(neuron 16t+c, sample 4g+r)
//             -> A (dZ) and B (H) operands of dW; obtained from a T tile by one 16x17-float LDS transpose per wave.
//   weights in LDS: see struct Img below (separate forward / transposed images read 128 bits at a time); the images
//   are built by the kernel itself from the torch-layout parameters (no separate packing pass).  The packed layout of
//   Plan16 (row stride 65) is only the layout of the GRADIENT image the workgroup flushes at the end.
#include <cstdlib>
#include "mlp_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#include <mutex>
#include <vector>

#define PSDF_MLP_BWD_SPLIT_DEFAULT 2   // 1 = three bf16 pieces, 2 = two fp16 pieces (see psdf_mlp_backward)
namespace psdf {
int g_last_path[PATH_FAMILIES] = {};
void* stream_scratch(size_t bytes, hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  struct Entry {
    hipStream_t st;
    int dev;
    void* ptr;
    size_t cap;
  };
  static std::mutex mu;
  static std::vector<Entry> entries;
  // Buffers that were outgrown are RETIRED, not freed: a second host thread may still hold the old pointer between this call
  // and its launch, and work already queued on the stream may still be using it; a hipFree here would also need a blocking
  // hipStreamSynchronize on the hot path.  Growth is geometric (x1.5, 4 MiB floor), so the retired total stays below twice the
  // live buffer; everything is released with the process.
  static std::vector<void*> retired;
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (auto& x : entries)
    if (x.st == st && x.dev == dev) e = &x;
  if (e && e->cap >= bytes) return e->ptr;
  size_t cap = bytes + bytes / 2;
  if (cap < ((size_t)1 << 22)) cap = (size_t)1 << 22;
  void* np = nullptr;
  if (hipMalloc(&np, cap) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (e) {
    retired.push_back(e->ptr);
    e->ptr = np;
    e->cap = cap;
  } else {
    entries.push_back(Entry{st, dev, np, cap});
  }
  return np;
}
}  // namespace psdf

namespace {

struct Plan16 {
  int n_layers;
  int dims[MAXL + 1];
  int tiles[MAXL + 1];  // ceil(dims/16); tiles[0] = input tiles
  int steps0;           // ceil(dims[0]/4)
  int w_off[MAXL], b_off[MAXL];
  int total;
  int final_dot;
};

int make_plan16(int n_layers, const int* dims, Plan16& p) {
  if (n_layers < 2 || n_layers > MAXL) return PSDF_ERR_ARG;
  p.n_layers = n_layers;
  for (int i = 0; i <= n_layers; i++) {
    if (dims[i] <= 0) return PSDF_ERR_ARG;
    p.dims[i] = dims[i];
    p.tiles[i] = (dims[i] + 15) / 16;
  }
  p.steps0 = (dims[0] + 3) / 4;
  p.final_dot = dims[n_layers] <= 4;
  int off = 0;
  for (int l = 0; l < n_layers; l++) {
    const bool last = l == n_layers - 1;
    p.w_off[l] = off;
    if (l == 0)
      off += p.tiles[1] * p.steps0 * WS;
    else if (last && p.final_dot)
      off += dims[n_layers] * p.tiles[l] * 16;
    else
      off += p.tiles[l + 1] * p.tiles[l] * 4 * WS;
    p.b_off[l] = off;
    off += (last && p.final_dot) ? 4 : p.tiles[l + 1] * 16;
  }
  p.total = off;
  return PSDF_OK;
}

// image index -> (layer, row, col | bias row); row/col may be out of range (padding)
__device__ __forceinline__ void unpack_index(const Plan16& p, int e, int& l, int& row, int& col, bool& is_bias) {
  l = 0;
#pragma unroll
  for (int i = 1; i < MAXL; i++)
    if (i < p.n_layers && e >= p.w_off[i]) l = i;
  const bool last = l == p.n_layers - 1;
  is_bias = e >= p.b_off[l];
  if (is_bias) {
    row = e - p.b_off[l];
    col = 0;
    return;
  }
  const int q = e - p.w_off[l];
  if (l == 0) {
    const int lane = q % WS, s = (q / WS) % p.steps0, to = (q / WS) / p.steps0;
    row = lane < 64 ? 16 * to + (lane & 15) : (1 << 20);
    col = 4 * s + (lane >> 4);
  } else if (last && p.final_dot) {  // [o][ti][r][g]
    const int g = q & 3, r = (q >> 2) & 3, ti = (q >> 4) % p.tiles[l], o = (q >> 4) / p.tiles[l];
    row = o;
    col = 16 * ti + 4 * g + r;
  } else {
    const int lane = q % WS, r = (q / WS) & 3, ti = (q / (WS * 4)) % p.tiles[l], to = (q / (WS * 4)) / p.tiles[l];
    row = lane < 64 ? 16 * to + (lane & 15) : (1 << 20);
    col = 16 * ti + 4 * (lane >> 4) + r;
  }
}

// upstream gradient of output `row` at sample n; dY == NULL means "the unit gradient of output 0" (d y_0 / d . : what the SDF
// normal needs, models.py:236-251) -- callers then neither allocate nor fill a [rows, N] tensor that is 1 in one row (round 4)
__device__ __forceinline__ float ld_dy(const float* __restrict__ dY, int row, int64_t N, int64_t n) {
  return dY ? dY[(int64_t)row * N + n] : (row == 0 ? 1.f : 0.f);
}

struct BwdPtrs {
  const float* W[MAXL];
  const float* b[MAXL];
  float* dW[MAXL];
  float* db[MAXL];
  float* partial;  // [gridDim.x][p.total] workgroup gradient images (summed by mlp_grad_reduce_kernel), or NULL: atomics
  const unsigned char* skip = nullptr;  // [N] or NULL: 16-sample tiles whose samples are all masked are not evaluated
};

// Workgroup gradient image -> global.  With a scratch buffer every workgroup stores its image (coalesced, no atomics:
// 256 workgroups adding into the same ~11 K addresses serialise in the L2) and a small second launch sums them.
template <int NTHREADS>
__device__ __forceinline__ void flush_image(const Plan16& p, const BwdPtrs& a, const float* __restrict__ ACC) {
  if (a.partial) {
    float* __restrict__ dst = a.partial + (size_t)blockIdx.x * p.total;
    for (int e = threadIdx.x; e < p.total; e += NTHREADS) dst[e] = ACC[e];
    return;
  }
  for (int e = threadIdx.x; e < p.total; e += NTHREADS) {
    const float v = ACC[e];
    if (v == 0.f) continue;
    int l, row, col;
    bool is_bias;
    unpack_index(p, e, l, row, col, is_bias);
    if (is_bias) {
      if (row < p.dims[l + 1]) atomicAdd(a.db[l] + row, v);
    } else if (row < p.dims[l + 1] && col < p.dims[l]) {
      atomicAdd(a.dW[l] + (int64_t)row * p.dims[l] + col, v);
    }
  }
}

// 64 image entries per workgroup; wave w sums the workgroup images w, w + 16, ... (coalesced), LDS combines the sixteen.
// (Four waves walking 64 images each took 16 us per call, five calls per training step: the loop is a chain of dependent
// loads on 77 workgroups; sixteen waves shorten it four-fold -- round 4.)
constexpr int GRW = 16;
__global__ void __launch_bounds__(GRW * 64) mlp_grad_reduce_kernel(Plan16 p, BwdPtrs a, int nimages) {
  __shared__ float part[GRW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < p.total)
    for (int b = wave; b < nimages; b += GRW) s += a.partial[(size_t)b * p.total + e];
  part[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || e >= p.total) return;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < GRW; w += 4) v += (part[w][lane] + part[w + 1][lane]) + (part[w + 2][lane] + part[w + 3][lane]);
  if (v == 0.f) return;
  int l, row, col;
  bool is_bias;
  unpack_index(p, e, l, row, col, is_bias);
  if (is_bias) {
    if (row < p.dims[l + 1]) a.db[l][row] += v;
  } else if (row < p.dims[l + 1] && col < p.dims[l]) {
    a.dW[l][(int64_t)row * p.dims[l] + col] += v;
  }
}

// Workgroup gradient images live in the library's per-stream scratch (psdf::stream_scratch); NULL (-> float atomics into
// dW / db: ~21 G/s on this chip, 0.2 ms for the 64-wide nets) only while the stream is being captured.
static float* grad_scratch_alloc(size_t floats, hipStream_t st) {
  static const bool force_atomics = getenv("PSDF_MLP_GRAD_ATOMICS") != nullptr;  // A/B switch for measurements
  if (force_atomics) return nullptr;
  return (float*)psdf::stream_scratch(floats * sizeof(float), st);
}

static void grad_scratch_reduce(const Plan16& p, const BwdPtrs& a, int nimages, hipStream_t st) {
  if (!a.partial) return;
  hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3((unsigned)((p.total + 63) / 64)), dim3(GRW * 64), 0, st, p, a, nimages);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int T>
__device__ __forceinline__ void init_bias16(f32x4 (&acc)[T], const float* __restrict__ b_lds, int g) {
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) acc[t][r] = b_lds[16 * t + 4 * g + r];
}
template <int T>
__device__ __forceinline__ void zero16(f32x4 (&a)[T]) {
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) a[t][r] = 0.f;
}

// z -> (gelu(z), gelu'(z)) with one erf and one exp; same formulas as gelu_exact / gelu_grad (mlp_device.h)
template <int T>
__device__ __forceinline__ void gelu_both(const f32x4 (&z)[T], f32x4 (&h)[T], f32x4 (&dg)[T]) {
  // pairs of values in packed fp32 arithmetic (same operations and order as the scalar formulas: bit-identical)
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2 x = {z[t][r], z[t][r + 1]};
      const f32x2 one_erf = splat2(1.0f) + erf_fast2(x * splat2(0.70710678118654752440f));
      const f32x2 hh = (splat2(0.5f) * x) * one_erf;
      const f32x2 e = (splat2(-0.5f) * x) * x;
      const f32x2 pdf = splat2(0.3989422804014327f) * f32x2{__expf(e.x), __expf(e.y)};
      const f32x2 d = pk_fma(x, pdf, splat2(0.5f) * one_erf);
      h[t][r] = hh.x;
      h[t][r + 1] = hh.y;
      dg[t][r] = d.x;
      dg[t][r + 1] = d.y;
    }
}

// Cross-LANE exchange through LDS: the compiler proves that, for ONE thread, the write of register r and the read of
// register r' != r never alias and interleaves them freely across __builtin_amdgcn_wave_barrier() (which is not a
// memory barrier for LLVM).  A wavefront-scope fence pair is: it costs no instruction (LDS is in order within a wave).
#define NT_FENCE()                                         \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// T tile -> NT tile through a 16x17 LDS buffer private to the wave
// (`buf` must NOT be __restrict__: that would tell LLVM nothing else -- not even a fence -- touches it.)
__device__ __forceinline__ void to_nt(const f32x4& in, f32x4& out, float* buf, int g, int c) {
#pragma unroll
  for (int r = 0; r < 4; r++) buf[(4 * g + r) * 17 + c] = in[r];
  NT_FENCE();
#pragma unroll
  for (int r = 0; r < 4; r++) out[r] = buf[c * 17 + 4 * g + r];
  NT_FENCE();
}

// Persistent accumulators of one chain layer: aw[to][ti] (reg q) = dW[16to+4g+q][16ti+c], ab[to] = per-lane partial of
// db[16to+c] (summed over g at the end).
template <int TO, int TI>
struct LayerAcc {
  f32x4 aw[TO][TI];
  float ab[TO];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int to = 0; to < TO; to++) {
      ab[to] = 0.f;
#pragma unroll
      for (int ti = 0; ti < TI; ti++)
#pragma unroll
        for (int q = 0; q < 4; q++) aw[to][ti][q] = 0.f;
    }
  }
  // aw += dz_nt x hin_nt over the 16 samples of the tile
  __device__ __forceinline__ void add(const f32x4 (&dz_nt)[TO], const f32x4 (&hin_nt)[TI]) {
#pragma unroll
    for (int to = 0; to < TO; to++) {
#pragma unroll
      for (int ti = 0; ti < TI; ti++)
#pragma unroll
        for (int s = 0; s < 4; s++) aw[to][ti] = MFMA16(dz_nt[to][s], hin_nt[ti][s], aw[to][ti]);
      ab[to] += (dz_nt[to][0] + dz_nt[to][1]) + (dz_nt[to][2] + dz_nt[to][3]);
    }
  }
  // same product without the bias term (contributions that are not d/dz of the forward layer)
  __device__ __forceinline__ void add_nb(const f32x4 (&a_nt)[TO], const f32x4 (&b_nt)[TI]) {
#pragma unroll
    for (int to = 0; to < TO; to++)
#pragma unroll
      for (int ti = 0; ti < TI; ti++)
#pragma unroll
        for (int s = 0; s < 4; s++) aw[to][ti] = MFMA16(a_nt[to][s], b_nt[ti][s], aw[to][ti]);
  }
  // Wave accumulators -> workgroup image, WITHOUT LDS float atomics: a wave-wide ds_add_f32 costs ~197 cycles on this chip
  // whatever the addresses (tools/atomic_bench.hip), and the ~90 of them per wave x 8 waves were 54 us of a kernel whose tile
  // loop takes 10 (measured with stage stamps, tools/bwd_stage_timing.py: cfg-4 training step sizes).  Instead the image is cut
  // into blocks (one per 16x16 accumulator tile, one per bias tile); the flush runs in NWV rounds separated by barriers, and
  // in round r wave w adds -- plain read, add, write -- the blocks b with b mod NWV == (w + r) mod NWV.  Within a round the
  // waves own disjoint blocks, after NWV rounds every wave has added every block.  `sel` = (w + r) mod NWV, `blk0` = index
  // of this layer's first block (compile-time after inlining); returns the number of blocks of the layer.
  template <int NWV>
  __device__ __forceinline__ int flush_chain(float* __restrict__ acc_w, float* __restrict__ acc_b, int g, int c, int sel, int blk0) {
    const int lane_off = (c & 3) * WS + (c >> 2) * 16 + 4 * g;
#pragma unroll
    for (int to = 0; to < TO; to++) {
#pragma unroll
      for (int ti = 0; ti < TI; ti++)
        if (((blk0 + to * TI + ti) & (NWV - 1)) == sel) {
          float* dst = acc_w + (to * TI + ti) * 4 * WS + lane_off;
#pragma unroll
          for (int q = 0; q < 4; q++) dst[q] += aw[to][ti][q];
        }
      if (((blk0 + TO * TI + to) & (NWV - 1)) == sel) {
        float sb = ab[to];
        sb += __shfl_xor(sb, 16, 64);
        sb += __shfl_xor(sb, 32, 64);
        if (g == 0) acc_b[16 * to + c] += sb;
      }
    }
    return TO * TI + TO;
  }
  // layer 0 image layout: [(to*S0 + 4t + c>>2)][(c&3)*16 + 4g + q]
  template <int NWV>
  __device__ __forceinline__ int flush_layer0(float* __restrict__ acc_w, float* __restrict__ acc_b, int S0, int g, int c, int sel,
                                              int blk0) {
#pragma unroll
    for (int to = 0; to < TO; to++) {
#pragma unroll
      for (int t = 0; t < TI; t++)
        if (((blk0 + to * TI + t) & (NWV - 1)) == sel && (4 * t + (c >> 2)) < S0) {
          float* dst = acc_w + (to * S0 + 4 * t + (c >> 2)) * WS + (c & 3) * 16 + 4 * g;
#pragma unroll
          for (int q = 0; q < 4; q++) dst[q] += aw[to][t][q];
        }
      if (((blk0 + TO * TI + to) & (NWV - 1)) == sel) {
        float sb = ab[to];
        sb += __shfl_xor(sb, 16, 64);
        sb += __shfl_xor(sb, 32, 64);
        if (g == 0) acc_b[16 * to + c] += sb;
      }
    }
    return TO * TI + TO;
  }
};

// Weight images of the kernel, laid out so that ONE ds_read_b128 per lane delivers the A operands of four
// consecutive MFMAs (the per-MFMA ds_read_b32 of the first version of this kernel was a third of its LDS traffic in
// instructions and the main thing a lone wave waited on).  Separate images for the forward product (A = W) and the
// transposed one (A = W^T), lane = (g, c), 4 floats per lane:
//   FWD chain layer (to,ti):  [lane][r] = W[16to + c][16ti + 4g + r]
//   BWD chain layer (to,ti):  [lane][r] = W[16to + 4g + r][16ti + c]
//   layer 0 forward (to,s4):  [lane][j] = W0[16to + c][4(4 s4 + j) + g]          (k-steps grouped by four)
//   layer 0 transposed (t,to):[lane][r] = W0[16to + 4g + r][16t + c]             (dX)
// followed by the hidden biases and the dot-head rows ([o][ti][r][g]).
template <int TI0, int T1, int T2, int T3, int OTS, bool FINAL_DOT>
struct Img {
  static constexpr int TL = (T3 > 0) ? T3 : T2;
  static constexpr int F0 = 0;
  static constexpr int B0 = F0 + T1 * TI0 * 256;
  static constexpr int F1 = B0 + TI0 * T1 * 256;
  static constexpr int B1 = F1 + T2 * T1 * 256;
  static constexpr int F2 = B1 + T2 * T1 * 256;
  static constexpr int B2 = F2 + T3 * T2 * 256;
  static constexpr int BO = B2 + T3 * T2 * 256;                 // transposed image of an MFMA output layer
  static constexpr int BIAS1 = BO + (FINAL_DOT ? 0 : OTS * TL * 256);
  static constexpr int BIAS2 = BIAS1 + T1 * 16;
  static constexpr int BIAS3 = BIAS2 + T2 * 16;
  static constexpr int WF = BIAS3 + T3 * 16;                    // dot head [4][TL][4][4]
  static constexpr int TOTAL = WF + (FINAL_DOT ? 4 * TL * 16 : 0);
};

// (clamped address + select, not a guarded load: a guarded load is a branch on the exec mask, and the staging loops below then
//  run one load -> wait -> store per iteration -- 5.5 us per workgroup for the SDF net's 9.7 K image entries, more than the tiles
//  of a training step's batch take; unconditional loads are issued together, round 6)
__device__ __forceinline__ float ld_w(const float* __restrict__ W, int rows, int cols, int row, int col) {
  const int rr = row < rows ? row : rows - 1, cc = col < cols ? col : cols - 1;
  const float v = W[rr * cols + cc];
  return (row < rows && col < cols) ? v : 0.f;
}
// stage one chain-layer pair of images (forward + transposed) of W [rows x cols]
__device__ __forceinline__ void stage_chain(float* fwd, float* bwd, const float* W, int rows, int cols, int TO, int TI,
                                            int tid, int nthreads) {
  const int n = TO * TI * 256;
#pragma unroll 4
  for (int e = tid; e < n; e += nthreads) {
    const int r = e & 3, lane = (e >> 2) & 63, pr = e >> 8, ti = pr % TI, to = pr / TI;
    const int g = lane >> 4, c = lane & 15;
    if (fwd) fwd[e] = ld_w(W, rows, cols, 16 * to + c, 16 * ti + 4 * g + r);
    if (bwd) bwd[e] = ld_w(W, rows, cols, 16 * to + 4 * g + r, 16 * ti + c);
  }
}

// out^T = W * in^T  (chain layout), A operands by 128-bit reads
template <int TI, int TO>
__device__ __forceinline__ void chain_fwd4(const f32x4 (&in)[TI], f32x4 (&out)[TO], const float* __restrict__ w, int lane) {
#pragma unroll
  for (int ti = 0; ti < TI; ti++) {
    f32x4 wv[TO];
#pragma unroll
    for (int to = 0; to < TO; to++) wv[to] = *reinterpret_cast<const f32x4*>(w + ((to * TI + ti) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int to = 0; to < TO; to++) out[to] = MFMA16(wv[to][r], in[ti][r], out[to]);
  }
}
// dh^T[in] += W^T dz^T[out]
template <int TO, int TI>
__device__ __forceinline__ void chain_bwd4(const f32x4 (&dz)[TO], f32x4 (&dh)[TI], const float* __restrict__ w, int lane) {
#pragma unroll
  for (int to = 0; to < TO; to++) {
    f32x4 wv[TI];
#pragma unroll
    for (int ti = 0; ti < TI; ti++) wv[ti] = *reinterpret_cast<const f32x4*>(w + ((to * TI + ti) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int ti = 0; ti < TI; ti++) dh[ti] = MFMA16(wv[ti][r], dz[to][r], dh[ti]);
  }
}

constexpr int BW = 4;  // waves per workgroup (one per SIMD: the kernel wants the whole 512-register file)

#ifdef PSDF_BWD_TIMING   // measurement build (tools/small_batch_bench.py): stage stamps of workgroup 0, 100 MHz clock
__device__ unsigned long long g_bwd_t[8];
#define BWD_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_bwd_t[k] = wall_clock64(); } while (0)
#else
#define BWD_STAMP(k) do { } while (0)
#endif

// NEED_DW = false: data gradient only (dX of a fixed net: analytic normals at inference); no accumulators, no transposes
// NW = waves per workgroup: 4 (one per SIMD, the whole register file) for 64-wide nets, 8 for 32-wide nets whose
// accumulators + state fit 256 registers -- two waves per SIMD cover each other's GELU / LDS / MFMA shadows.
template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, bool NEED_DX, bool NEED_DW = true, int NW = BW>
__global__ void __launch_bounds__(NW * 64)
    mlp_bwd_kernel(Plan16 p, int64_t N, const float* __restrict__ X, const float* __restrict__ dY,
                   float* __restrict__ dX, BwdPtrs a) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int SX = 4 * TI0;                       // k-steps of layer 0 (upper bound of p.steps0)
  constexpr int WAVE_LDS = 16 * 17 + 16 + 2 * SX * 64;
  constexpr int OTS_ = FINAL_DOT ? 1 : OUT_T;
  using IM = Img<TI0, T1, T2, T3, OTS_, FINAL_DOT>;
  const int img = IM::TOTAL > p.total ? IM::TOTAL : ((p.total + 3) & ~3);  // the gradient image re-uses this region later
  float* tbuf = lds + img + wave * WAVE_LDS;        // 16x17 transpose buffer
  float* dyb = tbuf + 16 * 17;                      // 16 floats
  float* xbuf = dyb + 16;                           // 2 x [SX][64]: layer-0 operand tiles landed by LDS-DMA
  BWD_STAMP(0);
  {
    const int tid = threadIdx.x, nt = NW * 64;
    const int d0 = p.dims[0], d1 = p.dims[1], d2 = p.dims[2], d3 = p.dims[3];
#pragma unroll 4
    for (int e = tid; e < T1 * TI0 * 256; e += nt) {  // layer 0: forward (k-steps by four) and transposed (dX)
      const int j = e & 3, ln = (e >> 2) & 63, gg = ln >> 4, cc = ln & 15;
      {
        const int pr = e >> 8, s4 = pr % TI0, to = pr / TI0;
        lds[IM::F0 + e] = ld_w(a.W[0], d1, d0, 16 * to + cc, 4 * (4 * s4 + j) + gg);
      }
      {
        const int pr = e >> 8, to = pr % T1, t = pr / T1;
        lds[IM::B0 + e] = ld_w(a.W[0], d1, d0, 16 * to + 4 * gg + j, 16 * t + cc);
      }
    }
    stage_chain(lds + IM::F1, lds + IM::B1, a.W[1], d2, d1, T2, T1, tid, nt);
    if constexpr (T3 > 0) stage_chain(lds + IM::F2, lds + IM::B2, a.W[2], d3, d2, T3, T2, tid, nt);
    const int lfi = p.n_layers - 1;
    if constexpr (!FINAL_DOT)
      stage_chain(nullptr, lds + IM::BO, a.W[lfi], p.dims[lfi + 1], p.dims[lfi], OTS_, IM::TL, tid, nt);
    for (int e = tid; e < T1 * 16; e += nt) lds[IM::BIAS1 + e] = e < d1 ? a.b[0][e] : 0.f;
    for (int e = tid; e < T2 * 16; e += nt) lds[IM::BIAS2 + e] = e < d2 ? a.b[1][e] : 0.f;
    if constexpr (T3 > 0)
      for (int e = tid; e < T3 * 16; e += nt) lds[IM::BIAS3 + e] = e < d3 ? a.b[2][e] : 0.f;
    if constexpr (FINAL_DOT)
      for (int e = tid; e < 4 * IM::TL * 16; e += nt) {  // [o][ti][r][g] = W[o][16ti + 4g + r]
        const int gg = e & 3, r = (e >> 2) & 3, ti = (e >> 4) % IM::TL, o = (e >> 4) / IM::TL;
        lds[IM::WF + e] = ld_w(a.W[lfi], p.dims[lfi + 1], p.dims[lfi], o, 16 * ti + 4 * gg + r);
      }
  }
  __syncthreads();
  BWD_STAMP(1);
  const int g = lane >> 4, c = lane & 15;
  const int K0 = p.dims[0], OUT = p.dims[p.n_layers], S0 = p.steps0;
  constexpr int TL = (T3 > 0) ? T3 : T2;
  constexpr int T3S = (T3 > 0) ? T3 : 1;      // storage sizes for the (absent) third hidden layer
  constexpr int OTS = FINAL_DOT ? 1 : OUT_T;
  const int lf = p.n_layers - 1;

  LayerAcc<T1, TI0> acc0;
  LayerAcc<T2, T1> acc1;
  LayerAcc<T3S, T2> acc2;
  LayerAcc<OTS, TL> acco;      // MFMA output layer
  float accf[4][TL];           // VALU (dot) output layer: per-lane partial of dW[o][16t+c]
  float accfb[4];
  acc0.zero();
  acc1.zero();
  acc2.zero();
  acco.zero();
#pragma unroll
  for (int o = 0; o < 4; o++) {
    accfb[o] = 0.f;
#pragma unroll
    for (int t = 0; t < TL; t++) accf[o][t] = 0.f;
  }

  {  // ---- scope of the read-only weight image
  const float* __restrict__ W = lds;
  // The wave runs alone on its SIMD (register budget), so nothing hides a memory latency for it.  The layer-0 operand
  // of the NEXT tile is therefore requested at the top of the current one with global_load_lds (memory -> LDS without
  // passing through registers: lane (g,c) of k-step s fetches X[4s+g][sample c] to xbuf[s][lane]); a tile later it is
  // read back both as the forward B operand and, transposed, as the B operand of dW0 (so X is read from HBM once).
  auto prefetch = [&](int64_t t, float* buf) {
    int64_t n = t * 16 + c;
    n = n < N ? n : N - 1;
    for (int s = 0; s < S0; s++) {
      int k = 4 * s + g;
      k = k < K0 ? k : K0 - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)k * N + n),
                                       (__attribute__((address_space(3))) void*)(buf + s * 64), 4, 0, 0);
    }
  };
  const int64_t ntiles = (N + 15) / 16;
  const int64_t tile0 = (int64_t)blockIdx.x * NW + wave, tstride = (int64_t)gridDim.x * NW;
  if (tile0 < ntiles) prefetch(tile0, xbuf);
  int cur = 0;
  for (int64_t tile = tile0; tile < ntiles; tile += tstride, cur ^= 1) {
    // the compiler does not track LDS-DMA completion: wait for the tile requested one iteration ago (this is also a
    // compiler barrier that keeps the LDS weight reads inside the tile loop)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* xb = xbuf + cur * (SX * 64);
    if (tile + tstride < ntiles) prefetch(tile + tstride, xbuf + (cur ^ 1) * (SX * 64));
    const int64_t n = tile * 16 + c;
    const bool live = n < N;
    if (a.skip && __ballot(live && !a.skip[n]) == 0) continue;   // a fully masked tile (its dX stays as it is)
    // ---------------------------------------------------------------- forward: activations and their derivatives
    f32x4 h1[T1], d1[T1], h2[T2], d2[T2], h3[T3S], d3[T3S];
    {
      f32x4 z[T1];
      init_bias16<T1>(z, W + IM::BIAS1, g);
      const int S4 = (S0 + 3) >> 2;
      for (int s4 = 0; s4 < S4; s4++) {
        f32x4 wv[T1];
#pragma unroll
        for (int to = 0; to < T1; to++)
          wv[to] = *reinterpret_cast<const f32x4*>(W + IM::F0 + ((to * TI0 + s4) * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int sj = 4 * s4 + j;
          if (sj < S0) {  // wave-uniform
            const float xv = xb[sj * 64 + lane];
            const float b = (4 * sj + g < K0 && live) ? xv : 0.f;
#pragma unroll
            for (int to = 0; to < T1; to++) z[to] = MFMA16(wv[to][j], b, z[to]);
          }
        }
      }
      gelu_both<T1>(z, h1, d1);
    }
    {
      f32x4 z[T2];
      init_bias16<T2>(z, W + IM::BIAS2, g);
      chain_fwd4<T1, T2>(h1, z, W + IM::F1, lane);
      gelu_both<T2>(z, h2, d2);
    }
    if constexpr (T3 > 0) {
      f32x4 z[T3S];
      init_bias16<T3S>(z, W + IM::BIAS3, g);
      chain_fwd4<T2, T3S>(h2, z, W + IM::F2, lane);
      gelu_both<T3S>(z, h3, d3);
    }
    // ---------------------------------------------------------------- output layer
    f32x4 dhl[TL];
    zero16<TL>(dhl);
    {
      f32x4 hl_nt[TL];
      if constexpr (NEED_DW) {
#pragma unroll
        for (int t = 0; t < TL; t++) {
          if constexpr (T3 > 0)
            to_nt(h3[t], hl_nt[t], tbuf, g, c);
          else
            to_nt(h2[t], hl_nt[t], tbuf, g, c);
        }
      }
      if constexpr (FINAL_DOT) {
        const float* __restrict__ wf = W + IM::WF;
#pragma unroll
        for (int o = 0; o < 4; o++) {
          if (o < OUT) {
            const float dy = live ? ld_dy(dY, o, N, n) : 0.f;
#pragma unroll
            for (int t = 0; t < TL; t++)
#pragma unroll
              for (int r = 0; r < 4; r++) dhl[t][r] = fmaf(wf[((o * TL + t) * 4 + r) * 4 + g], dy, dhl[t][r]);
            if constexpr (NEED_DW) {
              if (g == 0) dyb[c] = dy;
              NT_FENCE();
#pragma unroll
              for (int t = 0; t < TL; t++) {
                float pr = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) pr = fmaf(hl_nt[t][r], dyb[4 * g + r], pr);
                accf[o][t] += pr;
              }
              accfb[o] += (g == 0) ? dy : 0.f;
              NT_FENCE();
            }
          }
        }
      } else {
        f32x4 dyT[OTS];
#pragma unroll
        for (int to = 0; to < OTS; to++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * to + 4 * g + r;
            dyT[to][r] = (row < OUT && live) ? ld_dy(dY, row, N, n) : 0.f;
          }
        chain_bwd4<OTS, TL>(dyT, dhl, W + IM::BO, lane);
        if constexpr (NEED_DW) {
          f32x4 dy_nt[OTS];
#pragma unroll
          for (int to = 0; to < OTS; to++) to_nt(dyT[to], dy_nt[to], tbuf, g, c);
          acco.add(dy_nt, hl_nt);
        }
      }
    }
    // ---------------------------------------------------------------- hidden layers, last to first
    f32x4 dh2[T2];
    if constexpr (T3 > 0) {
#pragma unroll
      for (int t = 0; t < T3S; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) dhl[t][r] = dhl[t][r] * d3[t][r];  // dZ3
      if constexpr (NEED_DW) {
        f32x4 dz_nt[T3S], hin_nt[T2];
#pragma unroll
        for (int t = 0; t < T3S; t++) to_nt(dhl[t], dz_nt[t], tbuf, g, c);
#pragma unroll
        for (int t = 0; t < T2; t++) to_nt(h2[t], hin_nt[t], tbuf, g, c);
        acc2.add(dz_nt, hin_nt);
      }
      zero16<T2>(dh2);
      chain_bwd4<T3S, T2>(dhl, dh2, W + IM::B2, lane);
    } else {
#pragma unroll
      for (int t = 0; t < T2; t++) dh2[t] = dhl[t];
    }
#pragma unroll
    for (int t = 0; t < T2; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) dh2[t][r] = dh2[t][r] * d2[t][r];  // dZ2
    if constexpr (NEED_DW) {
      f32x4 dz_nt[T2], hin_nt[T1];
#pragma unroll
      for (int t = 0; t < T2; t++) to_nt(dh2[t], dz_nt[t], tbuf, g, c);
#pragma unroll
      for (int t = 0; t < T1; t++) to_nt(h1[t], hin_nt[t], tbuf, g, c);
      acc1.add(dz_nt, hin_nt);
    }
    f32x4 dh1[T1];
    zero16<T1>(dh1);
    chain_bwd4<T2, T1>(dh2, dh1, W + IM::B1, lane);
#pragma unroll
    for (int t = 0; t < T1; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) dh1[t][r] = dh1[t][r] * d1[t][r];  // dZ1
    if constexpr (NEED_DW) {
      // dW0[out][k] = sum_s dZ1[out][s] X[k][s]:  A = dZ1 NT, B = X NT = transposed read of the staged tile
      // (lane (g, c): feature k = 16t+c lives at k-step 4t + c>>2, row c&3; its samples 4g..4g+3 are contiguous)
      f32x4 dz_nt[T1], x_nt[TI0];
#pragma unroll
      for (int t = 0; t < T1; t++) to_nt(dh1[t], dz_nt[t], tbuf, g, c);
      const int64_t n0 = tile * 16 + 4 * g;
#pragma unroll
      for (int t = 0; t < TI0; t++) {
        const int k = 16 * t + c;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (4 * t + (c >> 2)) * 64 + (c & 3) * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; r++) x_nt[t][r] = (k < K0 && n0 + r < N) ? v[r] : 0.f;
      }
      acc0.add(dz_nt, x_nt);
    }
    if constexpr (NEED_DX) {
#pragma unroll
      for (int t = 0; t < TI0; t++) {
        f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int to = 0; to < T1; to++) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(W + IM::B0 + ((t * T1 + to) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; r++) dx = MFMA16(wv[r], dh1[to][r], dx);  // W0[16to + 4g + r][16t + c]
        }
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int k = 16 * t + 4 * g + q;
            if (k < K0) dX[(int64_t)k * N + n] = dx[q];
          }
        }
      }
    }
  }
  }  // ---- end of the weight image scope
  BWD_STAMP(2);
  if constexpr (!NEED_DW) return;
  // ---------------------------------------------------------------- wave accumulators -> workgroup image -> global
  // (the weight image is dead once every wave has left the tile loop: the gradient image takes its place)
  __syncthreads();
  float* ACC = lds;
  for (int e = threadIdx.x; e < p.total; e += NW * 64) ACC[e] = 0.f;
  __syncthreads();
  BWD_STAMP(3);
#pragma unroll 1
  for (int r = 0; r < NW; r++) {      // rotating rounds of plain read-add-write, see LayerAcc::flush_chain
    const int sel = (wave + r) & (NW - 1);
    int blk = 0;
    blk += acc0.template flush_layer0<NW>(ACC + p.w_off[0], ACC + p.b_off[0], S0, g, c, sel, blk);
    blk += acc1.template flush_chain<NW>(ACC + p.w_off[1], ACC + p.b_off[1], g, c, sel, blk);
    if constexpr (T3 > 0) blk += acc2.template flush_chain<NW>(ACC + p.w_off[2], ACC + p.b_off[2], g, c, sel, blk);
    if constexpr (FINAL_DOT) {
#pragma unroll
      for (int o = 0; o < 4; o++) {
        if (o < OUT) {
#pragma unroll
          for (int t = 0; t < TL; t++)
            if (((blk + o * TL + t) & (NW - 1)) == sel) {
              float pr = accf[o][t];
              pr += __shfl_xor(pr, 16, 64);
              pr += __shfl_xor(pr, 32, 64);
              // neuron 16t + c sits at [o][t][r = c&3][g = c>>2]
              if (g == 0) ACC[p.w_off[lf] + ((o * TL + t) * 4 + (c & 3)) * 4 + (c >> 2)] += pr;
            }
          if (((blk + 4 * TL + o) & (NW - 1)) == sel) {
            const float sb = psdf::wave_sum(accfb[o]);
            if (lane == 0) ACC[p.b_off[lf] + o] += sb;
          }
        }
      }
    } else {
      acco.template flush_chain<NW>(ACC + p.w_off[lf], ACC + p.b_off[lf], g, c, sel, blk);
    }
    __syncthreads();
  }
  BWD_STAMP(4);
  flush_image<NW * 64>(p, a, ACC);
  BWD_STAMP(5);
}

// ======================================================================================================
// DOUBLE backward: the vector-Jacobian product of the map (X, params) -> dX = J_X^T gy with an upstream gradient V,
// i.e. what differentiating THROUGH the analytic input gradient of the net needs (the reference's eikonal and
// curvature losses: get_sdf_and_gradient with create_graph=True, permuto_sdf_py/models/models.py:236-251, followed by
// loss.backward()).  With a_l = gelu'(z_l), c_l = gelu''(z_l) and the first-order chain
//   q3 = W3^T gy, u3 = a3*q3, q2 = W2^T u3, u2 = a2*q2, q1 = W1^T u2, u1 = a1*q1, dX = W0^T u1
// the adjoint is one forward-like sweep
//   du1 = W0 V,  dW0 += u1 V^T,  da1 = du1*q1, dq1 = du1*a1,  du2 = W1 dq1,  dW1 += u2 dq1^T,  ... dW3 += gy dq3^T
// followed by an ordinary backward sweep of the forward net with the injected pre-activation gradients
//   dz3 = da3*c3,  dz2 = (W2^T dz3)*a2 + da2*c2,  dz1 = (W1^T dz2)*a1 + da1*c1,  dX2 = W0^T dz1,  dW_l += dz_{l+1} h_l^T.
// Same tile shape, weight images, persistent dW accumulators and LDS-DMA staging (of X and V) as mlp_bwd_kernel.
template <int T>
__device__ __forceinline__ void gelu3(const f32x4 (&z)[T], f32x4 (&h)[T], f32x4 (&a1)[T], f32x4 (&c2)[T]) {
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2 x = {z[t][r], z[t][r + 1]};
      const f32x2 one_erf = splat2(1.0f) + erf_fast2(x * splat2(0.70710678118654752440f));
      const f32x2 e = (splat2(-0.5f) * x) * x;
      const f32x2 pdf = splat2(0.3989422804014327f) * f32x2{__expf(e.x), __expf(e.y)};
      const f32x2 hh = (splat2(0.5f) * x) * one_erf;
      const f32x2 aa = pk_fma(x, pdf, splat2(0.5f) * one_erf);   // Phi + x phi
      const f32x2 cc = pdf * (splat2(2.0f) - x * x);              // 2 phi + x phi' = phi (2 - x^2)
      h[t][r] = hh.x;
      h[t][r + 1] = hh.y;
      a1[t][r] = aa.x;
      a1[t][r + 1] = aa.y;
      c2[t][r] = cc.x;
      c2[t][r + 1] = cc.y;
    }
}
template <int T>
__device__ __forceinline__ void mul16(const f32x4 (&a)[T], const f32x4 (&b)[T], f32x4 (&out)[T]) {
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) out[t][r] = a[t][r] * b[t][r];
}
template <int T>
__device__ __forceinline__ void to_nt_all(const f32x4 (&in)[T], f32x4 (&out)[T], float* buf, int g, int c) {
#pragma unroll
  for (int t = 0; t < T; t++) to_nt(in[t], out[t], buf, g, c);
}

// Y2 (round 6): the same pass also takes an upstream gradient dY2 [OUT, N] of the net's OUTPUTS -- the plain backward that the
// training step runs right beside this one on the same samples (g_y of (sdf, geometry features) next to g_n of the normals).
// The ordinary backward sweep is linear in what is injected into it, so dY2 enters as one more chain product and one more dW
// block: dz3 += (W3^T dY2) * a3, dW3 += dY2 h3^T, db3 += dY2 -- one forward recomputation, one sweep, one staging and one flush
// instead of two of each (216 + 352 -> 400 matrix instructions per tile); dX2 comes out as the SUM of both data gradients.
template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, bool Y2 = false>
__global__ void __launch_bounds__(BW * 64)
    mlp_dbl_bwd_kernel(Plan16 p, int64_t N, const float* __restrict__ X, const float* __restrict__ V,
                       const float* __restrict__ dY, const float* __restrict__ dY2, float* __restrict__ dX2, BwdPtrs a) {
  static_assert(T3 > 0, "three hidden layers");
  static_assert(!(Y2 && FINAL_DOT), "the fused plain backward exists for nets with a matrix output layer");
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int SX = 4 * TI0;
  constexpr int WAVE_LDS = 16 * 17 + 16 + 4 * SX * 64;   // transpose buffer, dy, 2 x X tile, 2 x V tile
  constexpr int OTS = FINAL_DOT ? 1 : OUT_T;
  using IM = Img<TI0, T1, T2, T3, OTS, FINAL_DOT>;
  const int img = IM::TOTAL > p.total ? IM::TOTAL : ((p.total + 3) & ~3);
  float* tbuf = lds + img + wave * WAVE_LDS;
  float* dyb = tbuf + 16 * 17;
  float* xbuf = dyb + 16;
  float* vbuf = xbuf + 2 * SX * 64;
  BWD_STAMP(0);
  {
    const int tid = threadIdx.x, nt = BW * 64;
    const int d0 = p.dims[0], d1 = p.dims[1], d2 = p.dims[2], d3 = p.dims[3];
#pragma unroll 4
    for (int e = tid; e < T1 * TI0 * 256; e += nt) {
      const int j = e & 3, ln = (e >> 2) & 63, gg = ln >> 4, cc = ln & 15;
      {
        const int pr = e >> 8, s4 = pr % TI0, to = pr / TI0;
        lds[IM::F0 + e] = ld_w(a.W[0], d1, d0, 16 * to + cc, 4 * (4 * s4 + j) + gg);
      }
      {
        const int pr = e >> 8, to = pr % T1, t = pr / T1;
        lds[IM::B0 + e] = ld_w(a.W[0], d1, d0, 16 * to + 4 * gg + j, 16 * t + cc);
      }
    }
    stage_chain(lds + IM::F1, lds + IM::B1, a.W[1], d2, d1, T2, T1, tid, nt);
    stage_chain(lds + IM::F2, lds + IM::B2, a.W[2], d3, d2, T3, T2, tid, nt);
    const int lfi = p.n_layers - 1;
    if constexpr (!FINAL_DOT)
      stage_chain(nullptr, lds + IM::BO, a.W[lfi], p.dims[lfi + 1], p.dims[lfi], OTS, IM::TL, tid, nt);
    for (int e = tid; e < T1 * 16; e += nt) lds[IM::BIAS1 + e] = e < d1 ? a.b[0][e] : 0.f;
    for (int e = tid; e < T2 * 16; e += nt) lds[IM::BIAS2 + e] = e < d2 ? a.b[1][e] : 0.f;
    for (int e = tid; e < T3 * 16; e += nt) lds[IM::BIAS3 + e] = e < d3 ? a.b[2][e] : 0.f;
    if constexpr (FINAL_DOT)
      for (int e = tid; e < 4 * IM::TL * 16; e += nt) {
        const int gg = e & 3, r = (e >> 2) & 3, ti = (e >> 4) % IM::TL, o = (e >> 4) / IM::TL;
        lds[IM::WF + e] = ld_w(a.W[lfi], p.dims[lfi + 1], p.dims[lfi], o, 16 * ti + 4 * gg + r);
      }
  }
  __syncthreads();
  BWD_STAMP(1);
  const int g = lane >> 4, c = lane & 15;
  const int K0 = p.dims[0], OUT = p.dims[p.n_layers], S0 = p.steps0;
  const int lf = p.n_layers - 1;

  LayerAcc<T1, TI0> acc0;
  LayerAcc<T2, T1> acc1;
  LayerAcc<T3, T2> acc2;
  LayerAcc<OTS, T3> acco;
  float accf[4][T3];
  acc0.zero();
  acc1.zero();
  acc2.zero();
  acco.zero();
#pragma unroll
  for (int o = 0; o < 4; o++)
#pragma unroll
    for (int t = 0; t < T3; t++) accf[o][t] = 0.f;

  {
  const float* __restrict__ W = lds;
  auto prefetch = [&](int64_t t, float* xb_, float* vb_) {
    int64_t n = t * 16 + c;
    n = n < N ? n : N - 1;
    for (int s = 0; s < S0; s++) {
      int k = 4 * s + g;
      k = k < K0 ? k : K0 - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (int64_t)k * N + n),
                                       (__attribute__((address_space(3))) void*)(xb_ + s * 64), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(V + (int64_t)k * N + n),
                                       (__attribute__((address_space(3))) void*)(vb_ + s * 64), 4, 0, 0);
    }
  };
  // layer-0 product W0 * (operand tile staged in LDS), optional bias
  auto layer0_fwd = [&](const float* ob, bool live, f32x4 (&z)[T1]) {
    const int S4 = (S0 + 3) >> 2;
    for (int s4 = 0; s4 < S4; s4++) {
      f32x4 wv[T1];
#pragma unroll
      for (int to = 0; to < T1; to++)
        wv[to] = *reinterpret_cast<const f32x4*>(W + IM::F0 + ((to * TI0 + s4) * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int sj = 4 * s4 + j;
        if (sj < S0) {
          const float xv = ob[sj * 64 + lane];
          const float b = (4 * sj + g < K0 && live) ? xv : 0.f;
#pragma unroll
          for (int to = 0; to < T1; to++) z[to] = MFMA16(wv[to][j], b, z[to]);
        }
      }
    }
  };
  auto staged_nt = [&](const float* ob, int64_t tile, f32x4 (&o_nt)[TI0]) {
    const int64_t n0 = tile * 16 + 4 * g;
#pragma unroll
    for (int t = 0; t < TI0; t++) {
      const int k = 16 * t + c;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ob + (4 * t + (c >> 2)) * 64 + (c & 3) * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; r++) o_nt[t][r] = (k < K0 && n0 + r < N) ? v[r] : 0.f;
    }
  };
  const int64_t ntiles = (N + 15) / 16;
  const int64_t tile0 = (int64_t)blockIdx.x * BW + wave, tstride = (int64_t)gridDim.x * BW;
  if (tile0 < ntiles) prefetch(tile0, xbuf, vbuf);
  int cur = 0;
  for (int64_t tile = tile0; tile < ntiles; tile += tstride, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* xb = xbuf + cur * (SX * 64);
    const float* vb = vbuf + cur * (SX * 64);
    if (tile + tstride < ntiles) prefetch(tile + tstride, xbuf + (cur ^ 1) * (SX * 64), vbuf + (cur ^ 1) * (SX * 64));
    const int64_t n = tile * 16 + c;
    const bool live = n < N;
    // ---- forward: h, a = gelu', c = gelu''
    f32x4 h1[T1], a1[T1], c1[T1], h2[T2], a2[T2], c2[T2], a3[T3], c3[T3];
    f32x4 h3k[Y2 ? T3 : 1];      // h3: the output layer's input, needed again for dW3 += dY2 h3^T
    {
      f32x4 z[T1];
      init_bias16<T1>(z, W + IM::BIAS1, g);
      layer0_fwd(xb, live, z);
      gelu3<T1>(z, h1, a1, c1);
    }
    {
      f32x4 z[T2];
      init_bias16<T2>(z, W + IM::BIAS2, g);
      chain_fwd4<T1, T2>(h1, z, W + IM::F1, lane);
      gelu3<T2>(z, h2, a2, c2);
    }
    {
      f32x4 z[T3], h3[T3];
      init_bias16<T3>(z, W + IM::BIAS3, g);
      chain_fwd4<T2, T3>(h2, z, W + IM::F2, lane);
      gelu3<T3>(z, h3, a3, c3);
      if constexpr (Y2) {
#pragma unroll
        for (int t = 0; t < T3; t++) h3k[t] = h3[t];
      }
    }
    // ---- first-order chain: q3 = W3^T gy ... u1
    f32x4 q3[T3], q2[T2], q1[T1];
    zero16<T3>(q3);
    zero16<T2>(q2);
    zero16<T1>(q1);
    float dyo[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 dyT[OTS];
    if constexpr (FINAL_DOT) {
      const float* __restrict__ wf = W + IM::WF;
#pragma unroll
      for (int o = 0; o < 4; o++) {
        if (o < OUT) {
          dyo[o] = live ? ld_dy(dY, o, N, n) : 0.f;
#pragma unroll
          for (int t = 0; t < T3; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) q3[t][r] = fmaf(wf[((o * T3 + t) * 4 + r) * 4 + g], dyo[o], q3[t][r]);
        }
      }
    } else {
#pragma unroll
      for (int to = 0; to < OTS; to++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * to + 4 * g + r;
          dyT[to][r] = (row < OUT && live) ? ld_dy(dY, row, N, n) : 0.f;
        }
      chain_bwd4<OTS, T3>(dyT, q3, W + IM::BO, lane);
    }
    f32x4 u3[T3], u2[T2], u1[T1];
    mul16<T3>(a3, q3, u3);
    chain_bwd4<T3, T2>(u3, q2, W + IM::B2, lane);
    mul16<T2>(a2, q2, u2);
    chain_bwd4<T2, T1>(u2, q1, W + IM::B1, lane);
    mul16<T1>(a1, q1, u1);
    // ---- adjoint, forward-like sweep
    f32x4 da1[T1], da2[T2], da3[T3];
    {
      f32x4 du1[T1];
      zero16<T1>(du1);
      layer0_fwd(vb, live, du1);                                   // du1 = W0 V
      f32x4 u_nt[T1], v_nt[TI0];
      to_nt_all<T1>(u1, u_nt, tbuf, g, c);
      staged_nt(vb, tile, v_nt);
      acc0.add_nb(u_nt, v_nt);                                     // dW0 += u1 V^T
      f32x4 dq1[T1];
      mul16<T1>(du1, q1, da1);
      mul16<T1>(du1, a1, dq1);
      f32x4 du2[T2];
      zero16<T2>(du2);
      chain_fwd4<T1, T2>(dq1, du2, W + IM::F1, lane);              // du2 = W1 dq1
      f32x4 u2_nt[T2], dq1_nt[T1];
      to_nt_all<T2>(u2, u2_nt, tbuf, g, c);
      to_nt_all<T1>(dq1, dq1_nt, tbuf, g, c);
      acc1.add_nb(u2_nt, dq1_nt);                                  // dW1 += u2 dq1^T
      f32x4 dq2[T2];
      mul16<T2>(du2, q2, da2);
      mul16<T2>(du2, a2, dq2);
      f32x4 du3[T3];
      zero16<T3>(du3);
      chain_fwd4<T2, T3>(dq2, du3, W + IM::F2, lane);              // du3 = W2 dq2
      f32x4 u3_nt[T3], dq2_nt[T2];
      to_nt_all<T3>(u3, u3_nt, tbuf, g, c);
      to_nt_all<T2>(dq2, dq2_nt, tbuf, g, c);
      acc2.add_nb(u3_nt, dq2_nt);                                  // dW2 += u3 dq2^T
      f32x4 dq3[T3], dq3_nt[T3];
      mul16<T3>(du3, q3, da3);
      mul16<T3>(du3, a3, dq3);
      to_nt_all<T3>(dq3, dq3_nt, tbuf, g, c);
      if constexpr (FINAL_DOT) {                                   // dW3 += gy dq3^T
#pragma unroll
        for (int o = 0; o < 4; o++) {
          if (o < OUT) {
            if (g == 0) dyb[c] = dyo[o];
            NT_FENCE();
#pragma unroll
            for (int t = 0; t < T3; t++) {
              float pr = 0.f;
#pragma unroll
              for (int r = 0; r < 4; r++) pr = fmaf(dq3_nt[t][r], dyb[4 * g + r], pr);
              accf[o][t] += pr;
            }
            NT_FENCE();
          }
        }
      } else {
        f32x4 dy_nt[OTS];
        to_nt_all<OTS>(dyT, dy_nt, tbuf, g, c);
        acco.add_nb(dy_nt, dq3_nt);
      }
    }
    // ---- ordinary backward sweep with the injected pre-activation gradients
    f32x4 dz3[T3];
    mul16<T3>(da3, c3, dz3);
    if constexpr (Y2) {
      f32x4 gyT[OTS];
#pragma unroll
      for (int to = 0; to < OTS; to++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * to + 4 * g + r;
          gyT[to][r] = (row < OUT && live) ? dY2[(int64_t)row * N + n] : 0.f;
        }
      f32x4 dh3[T3];
      zero16<T3>(dh3);
      chain_bwd4<OTS, T3>(gyT, dh3, W + IM::BO, lane);                // W3^T dY2
#pragma unroll
      for (int t = 0; t < T3; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) dz3[t][r] = dz3[t][r] + dh3[t][r] * a3[t][r];
      f32x4 gy_nt[OTS], h3_nt[T3];
      to_nt_all<OTS>(gyT, gy_nt, tbuf, g, c);
      to_nt_all<T3>(h3k, h3_nt, tbuf, g, c);
      acco.add(gy_nt, h3_nt);                                          // dW3 += dY2 h3^T, db3 += dY2
    }
    {
      f32x4 dz_nt[T3], h_nt[T2];
      to_nt_all<T3>(dz3, dz_nt, tbuf, g, c);
      to_nt_all<T2>(h2, h_nt, tbuf, g, c);
      acc2.add(dz_nt, h_nt);
    }
    f32x4 dz2[T2];
    zero16<T2>(dz2);
    chain_bwd4<T3, T2>(dz3, dz2, W + IM::B2, lane);
#pragma unroll
    for (int t = 0; t < T2; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) dz2[t][r] = dz2[t][r] * a2[t][r] + da2[t][r] * c2[t][r];
    {
      f32x4 dz_nt[T2], h_nt[T1];
      to_nt_all<T2>(dz2, dz_nt, tbuf, g, c);
      to_nt_all<T1>(h1, h_nt, tbuf, g, c);
      acc1.add(dz_nt, h_nt);
    }
    f32x4 dz1[T1];
    zero16<T1>(dz1);
    chain_bwd4<T2, T1>(dz2, dz1, W + IM::B1, lane);
#pragma unroll
    for (int t = 0; t < T1; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) dz1[t][r] = dz1[t][r] * a1[t][r] + da1[t][r] * c1[t][r];
    {
      f32x4 dz_nt[T1], x_nt[TI0];
      to_nt_all<T1>(dz1, dz_nt, tbuf, g, c);
      staged_nt(xb, tile, x_nt);
      acc0.add(dz_nt, x_nt);
    }
#pragma unroll
    for (int t = 0; t < TI0; t++) {
      f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int to = 0; to < T1; to++) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(W + IM::B0 + ((t * T1 + to) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; r++) dx = MFMA16(wv[r], dz1[to][r], dx);
      }
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k = 16 * t + 4 * g + q;
          if (k < K0) dX2[(int64_t)k * N + n] = dx[q];
        }
      }
    }
  }
  }
  BWD_STAMP(2);
  __syncthreads();
  float* ACC = lds;
  for (int e = threadIdx.x; e < p.total; e += BW * 64) ACC[e] = 0.f;
  __syncthreads();
  BWD_STAMP(3);
#pragma unroll 1
  for (int r = 0; r < BW; r++) {      // rotating rounds of plain read-add-write, see LayerAcc::flush_chain
    const int sel = (wave + r) & (BW - 1);
    int blk = 0;
    blk += acc0.template flush_layer0<BW>(ACC + p.w_off[0], ACC + p.b_off[0], S0, g, c, sel, blk);
    blk += acc1.template flush_chain<BW>(ACC + p.w_off[1], ACC + p.b_off[1], g, c, sel, blk);
    blk += acc2.template flush_chain<BW>(ACC + p.w_off[2], ACC + p.b_off[2], g, c, sel, blk);
    if constexpr (FINAL_DOT) {
#pragma unroll
      for (int o = 0; o < 4; o++) {
        if (o < OUT) {
#pragma unroll
          for (int t = 0; t < T3; t++)
            if (((blk + o * T3 + t) & (BW - 1)) == sel) {
              float pr = accf[o][t];
              pr += __shfl_xor(pr, 16, 64);
              pr += __shfl_xor(pr, 32, 64);
              if (g == 0) ACC[p.w_off[lf] + ((o * T3 + t) * 4 + (c & 3)) * 4 + (c >> 2)] += pr;
            }
        }
      }
    } else {
      acco.template flush_chain<BW>(ACC + p.w_off[lf], ACC + p.b_off[lf], g, c, sel, blk);  // its bias partials are zero (add_nb only)
    }
    __syncthreads();
  }
  BWD_STAMP(4);
  flush_image<BW * 64>(p, a, ACC);
  BWD_STAMP(5);
}

template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, bool Y2 = false>
int launch_dbl_bwd(const Plan16& p, int64_t N, const float* X, const float* V, const float* dY, float* dX2,
                   const BwdPtrs& a, hipStream_t st, const float* dY2 = nullptr) {
  using IM = Img<TI0, T1, T2, T3, FINAL_DOT ? 1 : OUT_T, FINAL_DOT>;
  const int img = IM::TOTAL > p.total ? IM::TOTAL : ((p.total + 3) & ~3);
  const size_t shmem = ((size_t)img + BW * (16 * 17 + 16 + 4 * 4 * TI0 * 64)) * sizeof(float);
  if (shmem > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
  const int64_t ntiles = (N + 15) / 16;
  int64_t blocks = (ntiles + BW - 1) / BW;
  if (blocks > 256) blocks = 256;
  auto kern = mlp_dbl_bwd_kernel<TI0, T1, T2, T3, OUT_T, FINAL_DOT, Y2>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return (int)e;
  BwdPtrs ap = a;
  ap.partial = grad_scratch_alloc((size_t)blocks * p.total, st);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BW * 64), shmem, st, p, N, X, V, dY, dY2, dX2, ap);
  grad_scratch_reduce(p, ap, (int)blocks, st);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// WITH_DW = false: only the data-gradient instantiation exists for this shape.  The 64-wide nets with MANY outputs take their
// parameter gradients from the workgroup-cooperative kernel (mlp_wide.hip); their single-wave dW instantiations spilled 128 - 253
// registers and were reachable only while a stream was being captured -- they are not built any more (round 4): such a call
// returns PSDF_ERR_UNSUPPORTED (tests/test_dispatch_tables.py lists every instantiation that still spills, with its route).
template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, bool WITH_DW, int NW_>
int launch_bwd_nw(const Plan16& p, int64_t N, const float* X, const float* dY, float* dX, const BwdPtrs& a, hipStream_t st) {
  constexpr int NW = NW_;   // waves per workgroup (launch_bwd below picks it)
  const int64_t ntiles = (N + 15) / 16;
  int64_t blocks = (ntiles + NW - 1) / NW;
  if (blocks > 256) blocks = 256;  // one workgroup per CU; each wave walks many tiles
  using IM = Img<TI0, T1, T2, T3, FINAL_DOT ? 1 : OUT_T, FINAL_DOT>;
  const int img = IM::TOTAL > p.total ? IM::TOTAL : ((p.total + 3) & ~3);
  if (!a.dW[0]) {  // data gradient only: no accumulators -> two workgroups per CU
    if (!dX) return PSDF_OK;
    const size_t shmem = ((size_t)img + BW * (16 * 17 + 16 + 2 * 4 * TI0 * 64)) * sizeof(float);
    if (shmem > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
    auto kern = mlp_bwd_kernel<TI0, T1, T2, T3, OUT_T, FINAL_DOT, true, false, BW>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return (int)e;
    // ONE resident round of workgroups (two per CU): staging the weight image is 5.6 us of index arithmetic per workgroup, as
    // long as two tiles of the SDF net, so a second round costs more than a longer walk (round 6, tools/small_batch_bench.py
    // 52-32-32-32-33 at 49 152 / 262 144 samples: 1024 workgroups 30.0 / 92.8 us, 512: 25.4 / 86.6, 256: 28.2 / 108.7; eight waves
    // per workgroup sharing one image: 25.1 / 83.2 at 256 workgroups -- no better, not kept)
    int64_t nb = (ntiles + BW - 1) / BW;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(BW * 64), shmem, st, p, N, X, dY, dX, a);
    PSDF_LAUNCH_CHECK();
    return PSDF_OK;
  }
  if constexpr (!WITH_DW) {
    return PSDF_ERR_UNSUPPORTED;
  } else {
  const size_t shmem = ((size_t)img + NW * (16 * 17 + 16 + 2 * 4 * TI0 * 64)) * sizeof(float);
  if (shmem > 160 * 1024) return PSDF_ERR_UNSUPPORTED;
#define GO(DX)                                                                                                     \
  do {                                                                                                             \
    auto kern = mlp_bwd_kernel<TI0, T1, T2, T3, OUT_T, FINAL_DOT, DX, true, NW>;                                    \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);  \
    if (e != hipSuccess) return (int)e;                                                                            \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), shmem, st, p, N, X, dY, dX, ap);                 \
  } while (0)
  BwdPtrs ap = a;
  ap.partial = grad_scratch_alloc((size_t)blocks * p.total, st);
  if (dX)
    GO(true);
  else
    GO(false);
#undef GO
  grad_scratch_reduce(p, ap, (int)blocks, st);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
  }
}
// 32-wide nets: two waves per SIMD (8 per workgroup) pay from ~2^17 samples on -- 131.8 against 148.6 us at 262 144 samples --
// but at a training step's ~49 K samples one wave per SIMD with the whole register file is faster AND does not spill (the
// 8-wave instantiation of the reference's SDF net parks 24 registers in scratch): 49.7 against 52.9 us, 25.4 against 30.9 us at
// 1 024 samples (round 5, tools/small_batch_bench.py 52-32-32-32-33).  Wider nets: one wave per SIMD always.
template <int TI0, int T1, int T2, int T3, int OUT_T, bool FINAL_DOT, bool WITH_DW = true>
int launch_bwd(const Plan16& p, int64_t N, const float* X, const float* dY, float* dX, const BwdPtrs& a, hipStream_t st) {
  if constexpr (T1 <= 2 && T2 <= 2 && T3 <= 2) {
    if (N >= ((int64_t)1 << 17)) return launch_bwd_nw<TI0, T1, T2, T3, OUT_T, FINAL_DOT, WITH_DW, 8>(p, N, X, dY, dX, a, st);
  }
  return launch_bwd_nw<TI0, T1, T2, T3, OUT_T, FINAL_DOT, WITH_DW, BW>(p, N, X, dY, dX, a, st);
}

}  // namespace

#ifdef PSDF_BWD_TIMING
extern "C" int psdf_debug_bwd_timing(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_t), sizeof(unsigned long long) * 8);
}
#endif

extern "C" {

int psdf_mlp_backward_split(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                            const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                            void* stream);   // mlp_bwd_split.hip
int psdf_mlp_backward_wide(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                           const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                           void* stream);    // mlp_wide.hip
int psdf_mlp_backward_split_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                                const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                                void* stream);   // mlp_bwd_split_f16.hip

// Backward of psdf_mlp_forward.  weights[l] / biases[l]: the torch-layout parameters (W_l [dims[l+1], dims[l]]);
// X [dims[0], N], dY [dims[n_layers], N] and dX [dims[0], N] (or NULL) are feature-major; dW[l] (torch layout) and
// db[l] are ACCUMULATED INTO (caller zero-fills); pass dW = db = NULL for the data gradient only (a lighter kernel).
int psdf_mlp_backward(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                      const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db,
                      void* stream) {
  Plan16 p;
  int rc = make_plan16(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !X || !weights || !biases || ((dW == nullptr) != (db == nullptr))) return PSDF_ERR_ARG;
  if (!dY && dW) return PSDF_ERR_ARG;      // dY == NULL (unit gradient of output 0) is a data-gradient-only request
  if (n_layers != 3 && n_layers != 4) return PSDF_ERR_UNSUPPORTED;
  // Large batches of the BASELINE net (<= 64 inputs, 64x3, 1 output) with parameter gradients: the split-operand kernels on
  // the 16-bit matrix pipes -- by default two fp16 pieces per operand (mlp_bwd_split_f16.hip: errors of a few 1e-6 of the largest
  // entry), PSDF_MLP_BWD_SPLIT=bf16 three bf16 pieces (mlp_bwd_split.hip: fp32-level accuracy, no range restriction, 1.3x
  // slower), =0 the fp32-MFMA kernel below; -2 from a split entry (other widths, no stream-ordered scratch) falls through.
  if (dW && N >= (1 << 18)) {
    // PSDF_MLP_BWD_SPLIT: "0" fp32-MFMA kernel, "bf16" three bf16 pieces (mlp_bwd_split.hip), "f16" two fp16 pieces
    // (mlp_bwd_split_f16.hip); read at every call (a getenv is nanoseconds) so that tests can A/B the variants in one process
    const char* v = getenv("PSDF_MLP_BWD_SPLIT");
    const int which = !v ? PSDF_MLP_BWD_SPLIT_DEFAULT : (v[0] == '0' ? 0 : (v[0] == 'f' ? 2 : 1));
    if (which == 2) {
      const int r = psdf_mlp_backward_split_f16(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, stream);
      if (r != PSDF_ERR_UNSUPPORTED) {
        psdf::g_last_path[psdf::PATH_MLP_BWD] = 4;
        return r;
      }
    }
    if (which != 0) {
      const int r = psdf_mlp_backward_split(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, stream);
      if (r != PSDF_ERR_UNSUPPORTED) {
        psdf::g_last_path[psdf::PATH_MLP_BWD] = 2;
        return r;
      }
    }
  }
  BwdPtrs a;
  a.partial = nullptr;
  for (int l = 0; l < MAXL; l++) {
    a.W[l] = l < n_layers ? weights[l] : nullptr;
    a.b[l] = l < n_layers ? biases[l] : nullptr;
    a.dW[l] = (dW && l < n_layers) ? dW[l] : nullptr;
    a.db[l] = (db && l < n_layers) ? db[l] : nullptr;
    if (l < n_layers && (!a.W[l] || !a.b[l] || (dW && (!a.dW[l] || !a.db[l])))) return PSDF_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ti0 = p.tiles[0], t1 = p.tiles[1], t2 = p.tiles[2], t3 = (n_layers == 4) ? p.tiles[3] : 0,
            to = p.tiles[n_layers];
  // 64-wide nets with MANY outputs (the background density / feature net 52 -> 64 x 3 -> 65, and 64 x 3 -> 33): their dW does
  // not fit one wave's registers (the single-wave instantiations below spill 213-227 registers) -- the workgroup-cooperative
  // kernel of mlp_wide.hip splits the dW rows over 8 waves.  -2 (no stream-ordered scratch: capture) falls through.
  if (dW && n_layers == 4 && to >= 2 && t1 == 4 && t2 == 4 && t3 == 4) {
    const int r = psdf_mlp_backward_wide(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, stream);
    if (r != PSDF_ERR_UNSUPPORTED) {
      psdf::g_last_path[psdf::PATH_MLP_BWD] = 3;
      return r;
    }
  }
  // the background colour head 80 -> 64 x 2 -> 3: the split-fp16 workgroup kernel, two hidden layers (round 6; the fp32 single-wave
  // instantiation below stays for PSDF_MLP_WIDE_SPLIT=f32, capture, and after an overflow of the fp16 range guard)
  if (dW && dX && n_layers == 3 && ti0 == 5 && t1 == 4 && t2 == 4 && to == 1 && p.final_dot) {
    const int r = psdf_mlp_backward_wide(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, stream);
    if (r != PSDF_ERR_UNSUPPORTED) {
      psdf::g_last_path[psdf::PATH_MLP_BWD] = 3;
      return r;
    }
  }
  psdf::g_last_path[psdf::PATH_MLP_BWD] = 1;
#define CASE_(I, A, B, C, O, D, W)                                               \
  if (ti0 == I && t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) \
    return launch_bwd<I, A, B, C, O, D, W>(p, N, X, dY, dX, a, st);
#define CASE(I, A, B, C, O, D) CASE_(I, A, B, C, O, D, true)
#define CASE_DX_ONLY(I, A, B, C, O, D) CASE_(I, A, B, C, O, D, false)
  CASE(3, 4, 4, 4, 1, true)   // 33..48 -> 64x3 -> 1..4   (BASELINE SDF net on a 16-level encoding)
  CASE(4, 4, 4, 4, 1, true)   // 49..64 -> 64x3 -> 1..4   (24-level encoding)
  CASE(2, 4, 4, 4, 1, true)   // 17..32 -> 64x3 -> 1..4   (small encodings)
  CASE(4, 2, 2, 2, 1, true)   // 49..64 -> 32x3 -> 1..4
  CASE(3, 2, 2, 2, 1, true)   // 33..48 -> 32x3 -> 1..4
  CASE(2, 2, 2, 2, 1, true)   // 17..32 -> 32x3 -> 1..4
  CASE(4, 2, 2, 2, 3, false)  // 52 -> 32x3 -> 33         (reference SDF net, models.py:153-161)
  CASE(3, 2, 2, 2, 3, false)  // 36 -> 32x3 -> 33         (same net on a 16-level encoding)
  CASE_DX_ONLY(4, 4, 4, 4, 5, false)  // 52 -> 64x3 -> 65  (background density net, models.py:451-459): dX here, dW in mlp_wide.hip
  CASE_DX_ONLY(4, 4, 4, 4, 3, false)  // 52 -> 64x3 -> 33
  CASE_DX_ONLY(3, 4, 4, 4, 3, false)  // 36 -> 64x3 -> 33
  CASE(5, 4, 4, 0, 1, true)   // 80 -> 64x2 -> 3          (background colour head, models.py:463-469)
#undef CASE
#undef CASE_DX_ONLY
#undef CASE_
  // nets too wide for one wave's registers (the 128-wide colour network): workgroup-cooperative kernel, mlp_wide.hip
  psdf::g_last_path[psdf::PATH_MLP_BWD] = 3;
  if (dW) return psdf_mlp_backward_wide(n_layers, dims, N, X, weights, biases, dY, dX, dW, db, stream);
  return PSDF_ERR_UNSUPPORTED;
}

// Which kernel variant the last call of an operator family dispatched to (debug query for the parity tests; host only).
//   family 0, encode backward: 1 = LDS scatter cache + float atomics (small batches), 2 = queue mode (binning launch +
//             encode_bwd_reduce_kernel), 3 = position gradient only
//   family 1, MLP backward   : 1 = fp32-MFMA kernel (mlp_bwd_kernel), 2 = split-bf16 kernel (mlp_bwd_split_kernel),
//             3 = workgroup-cooperative wide kernel (mlp_wide_bwd_kernel), 4 = split-fp16 kernel (mlp_bwd_split_f16_kernel)
//   family 2, MLP forward    : 1 = fp32-MFMA kernel (mlp_fwd_kernel), 2 = split-bf16 kernel (mlp_fwd_split_kernel),
//             3 = split-fp16 kernel (psdf_mlp_forward_f16)
// 0 = no call yet; -1 = unknown family.
int psdf_last_path(int family) {
  if (family < 0 || family >= psdf::PATH_FAMILIES) return -1;
  return psdf::g_last_path[family];
}

// psdf_mlp_backward restricted to the data gradient (dW = db = NULL) with a per-sample mask: 16-sample tiles whose samples
// are all masked are skipped (their columns of dX keep their contents); masked samples inside a live tile are evaluated.
int psdf_mlp_backward_data_masked(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                                  const float* const* biases, const float* dY, const unsigned char* skip, float* dX,
                                  void* stream) {
  Plan16 p;
  int rc = make_plan16(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !X || !weights || !biases || !dX) return PSDF_ERR_ARG;      // dY == NULL: unit gradient of output 0
  if (n_layers != 3 && n_layers != 4) return PSDF_ERR_UNSUPPORTED;
  BwdPtrs a;
  a.partial = nullptr;
  a.skip = skip;
  for (int l = 0; l < MAXL; l++) {
    a.W[l] = l < n_layers ? weights[l] : nullptr;
    a.b[l] = l < n_layers ? biases[l] : nullptr;
    a.dW[l] = nullptr;
    a.db[l] = nullptr;
    if (l < n_layers && (!a.W[l] || !a.b[l])) return PSDF_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ti0 = p.tiles[0], t1 = p.tiles[1], t2 = p.tiles[2], t3 = (n_layers == 4) ? p.tiles[3] : 0,
            to = p.tiles[n_layers];
#define CASE(I, A, B, C, O, D)                                                   \
  if (ti0 == I && t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) \
    return launch_bwd<I, A, B, C, O, D>(p, N, X, dY, dX, a, st);
  CASE(3, 4, 4, 4, 1, true)
  CASE(4, 4, 4, 4, 1, true)
  CASE(2, 4, 4, 4, 1, true)
  CASE(4, 2, 2, 2, 1, true)
  CASE(3, 2, 2, 2, 1, true)
  CASE(2, 2, 2, 2, 1, true)
#undef CASE
  return PSDF_ERR_UNSUPPORTED;
}

// Double backward (see mlp_dbl_bwd_kernel): V [dims[0], N] is the upstream gradient of the dX that psdf_mlp_backward
// produced for the same X / parameters / dY.  dX2 [dims[0], N] receives the gradient wrt X; dW[l] / db[l] are
// ACCUMULATED INTO.  Three hidden layers only; returns PSDF_ERR_UNSUPPORTED (-2) for widths without an instantiation.
static int mlp_double_backward_impl(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                                    const float* const* biases, const float* dY, const float* V, const float* dY2, float* dX2,
                                    float* const* dW, float* const* db, void* stream) {
  Plan16 p;
  int rc = make_plan16(n_layers, dims, p);
  if (rc != PSDF_OK) return rc;
  if (n_layers != 4) return PSDF_ERR_UNSUPPORTED;
  if (N == 0) return PSDF_OK;
  if (N < 0 || !X || !weights || !biases || !V || !dX2 || !dW || !db) return PSDF_ERR_ARG;   // dY == NULL: unit gradient of output 0
  BwdPtrs a;
  a.partial = nullptr;
  for (int l = 0; l < MAXL; l++) {
    a.W[l] = l < n_layers ? weights[l] : nullptr;
    a.b[l] = l < n_layers ? biases[l] : nullptr;
    a.dW[l] = l < n_layers ? dW[l] : nullptr;
    a.db[l] = l < n_layers ? db[l] : nullptr;
    if (l < n_layers && (!a.W[l] || !a.b[l] || !a.dW[l] || !a.db[l])) return PSDF_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ti0 = p.tiles[0], t1 = p.tiles[1], t2 = p.tiles[2], t3 = p.tiles[3], to = p.tiles[n_layers];
#define CASE(I, A, B, C, O, D)                                                   \
  if (ti0 == I && t1 == A && t2 == B && t3 == C && to == O && p.final_dot == D) \
    return launch_dbl_bwd<I, A, B, C, O, D>(p, N, X, V, dY, dX2, a, st);
  if (dY2) {     // with the plain backward of an upstream gradient of the outputs folded in: the reference's SDF net only
    if (ti0 == 4 && t1 == 2 && t2 == 2 && t3 == 2 && to == 3 && !p.final_dot)
      return launch_dbl_bwd<4, 2, 2, 2, 3, false, true>(p, N, X, V, dY, dX2, a, st, dY2);
    if (ti0 == 3 && t1 == 2 && t2 == 2 && t3 == 2 && to == 3 && !p.final_dot)
      return launch_dbl_bwd<3, 2, 2, 2, 3, false, true>(p, N, X, V, dY, dX2, a, st, dY2);
    return PSDF_ERR_UNSUPPORTED;
  }
  CASE(4, 2, 2, 2, 3, false)  // 52 -> 32x3 -> 33   (reference SDF net, models.py:153-161)
  CASE(3, 2, 2, 2, 3, false)  // 36 -> 32x3 -> 33
  CASE(4, 2, 2, 2, 1, true)   // 49..64 -> 32x3 -> 1..4
  CASE(3, 2, 2, 2, 1, true)
  CASE(2, 2, 2, 2, 1, true)
  CASE(3, 4, 4, 4, 1, true)   // 36 -> 64x3 -> 1..4 (BASELINE net)
  CASE(4, 4, 4, 4, 1, true)
#undef CASE
  return PSDF_ERR_UNSUPPORTED;
}

int psdf_mlp_double_backward(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                             const float* const* biases, const float* dY, const float* V, float* dX2,
                             float* const* dW, float* const* db, void* stream) {
  return mlp_double_backward_impl(n_layers, dims, N, X, weights, biases, dY, V, nullptr, dX2, dW, db, stream);
}

// The double backward AND the plain backward of an upstream gradient dY2 [dims[n_layers], N] of the net's outputs in one pass
// (mlp_dbl_bwd_kernel<..., Y2>): dX2 = (data gradient of the double backward) + (dX of psdf_mlp_backward for dY2), dW / db
// accumulate both.  The reference's SDF net shapes (<= 64 inputs, 32 x 3 hidden, 5..48 outputs); -2 otherwise.
int psdf_mlp_double_backward_plus(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
                                  const float* const* biases, const float* dY, const float* V, const float* dY2, float* dX2,
                                  float* const* dW, float* const* db, void* stream) {
  if (!dY2) return PSDF_ERR_ARG;
  return mlp_double_backward_impl(n_layers, dims, N, X, weights, biases, dY, V, dY2, dX2, dW, db, stream);
}

}  // extern "C"
