// Permutohedral-lattice hash encoding for gfx950: forward, backward (lattice + positions) and
// double backward (from positions).  Replaces the CUDA op of the un-vendored package
// `permutohedral_encoding` that the reference imports at permuto_sdf_py/models/models.py:20 and calls
// at models.py:186,370,500,542.  Algorithm and frozen conventions: SURVEY.md App. A / oracle/permuto_oracle.py.
//
// Data layout in HBM
//   positions      [N, P]      fp32 row major (what the callers hand over)
//   lattice_values [L, T, F]   fp32 ("monolithic"; per-level table = T*F*4 B = 2 MiB at T=2^18, F=2)
//   sliced         [Lt*F, N]   fp32 feature-major (Lt = L + ceil(P/F) when points are concatenated):
//                              every store of a wave is one 256-B line, and the consumer (the fused MLP,
//                              mlp.hip) reads its MFMA B-operand rows straight from this layout.
// Launch shape: grid (ceil(N/256), Lt), one thread per (point, level).  blockIdx.x is the fast dispatch
// dimension, so the chip sweeps one level's 2-MiB table at a time and that table stays resident in every
// XCD's 4-MiB L2 (the gathers are L2 hits, HBM sees positions + outputs only).
#include "psdf_common.h"

namespace {

template <int P>
struct Simplex {
  int rem0[P + 1];
  int rank[P + 1];
  float bary[P + 2];
};

// elevate -> closest 0-colour point -> rank -> barycentric.  All loops are fully unrolled and every
// array index is a compile-time constant after unrolling (runtime-indexed arrays would go to scratch).
template <int P>
__device__ __forceinline__ void compute_simplex(const float* __restrict__ pos, const float* __restrict__ shift,
                                                const float* __restrict__ sf, Simplex<P>& s) {
  float E[P + 1];
  float sm = 0.f;
#pragma unroll
  for (int i = P; i > 0; i--) {
    float cf = (pos[i - 1] + shift[i - 1]) * sf[i - 1];
    E[i] = sm - (float)i * cf;
    sm = sm + cf;
  }
  E[0] = sm;

  const double inv = 1.0 / (P + 1);
  int sum = 0;
#pragma unroll
  for (int i = 0; i <= P; i++) {
    float v = (float)((double)E[i] * inv);
    float up = ceilf(v) * (float)(P + 1);
    float down = floorf(v) * (float)(P + 1);
    s.rem0[i] = ((up - E[i]) < (E[i] - down)) ? (int)up : (int)down;
    sum += s.rem0[i];
  }
  sum /= (P + 1);

  float d[P + 1];
#pragma unroll
  for (int i = 0; i <= P; i++) {
    d[i] = E[i] - (float)s.rem0[i];
    s.rank[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < P; i++) {
#pragma unroll
    for (int j = i + 1; j <= P; j++) {
      if (d[i] < d[j])
        s.rank[i]++;
      else
        s.rank[j]++;
    }
  }
#pragma unroll
  for (int i = 0; i <= P; i++) {
    s.rank[i] += sum;
    if (s.rank[i] < 0) {
      s.rank[i] += P + 1;
      s.rem0[i] += P + 1;
    } else if (s.rank[i] > P) {
      s.rank[i] -= P + 1;
      s.rem0[i] -= P + 1;
    }
  }
  // recompute d after the fix-up (rem0 may have moved by +-(P+1)); same expression as the oracle
#pragma unroll
  for (int k = 0; k <= P + 1; k++) s.bary[k] = 0.f;
#pragma unroll
  for (int i = 0; i <= P; i++) {
    float delta = (float)((double)(E[i] - (float)s.rem0[i]) * inv);
#pragma unroll
    for (int k = 0; k <= P + 1; k++) {
      if (k == P - s.rank[i]) s.bary[k] = s.bary[k] + delta;
      if (k == P + 1 - s.rank[i]) s.bary[k] = s.bary[k] - delta;
    }
  }
  s.bary[0] = (float)((double)s.bary[0] + (1.0 + (double)s.bary[P + 1]));
}

template <int P>
__device__ __forceinline__ uint32_t vertex_row(const Simplex<P>& s, int remainder, uint32_t capacity) {
  uint32_t h = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
    int k = s.rem0[i] + remainder;
    if (s.rank[i] > P - remainder) k -= (P + 1);
    h += (uint32_t)k;
    h *= 2531011u;
  }
  return h % capacity;
}

template <int P>
__device__ __forceinline__ void load_pos(const float* __restrict__ positions, int64_t n, float* pos) {
#pragma unroll
  for (int i = 0; i < P; i++) pos[i] = positions[n * P + i];
}

// ------------------------------------------------------------------------------------------ forward
template <int P, int F>
__global__ void __launch_bounds__(PSDF_BLOCK)
    encode_fwd_kernel(int64_t N, int L, uint32_t capacity, const float* __restrict__ positions,
                      const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                      const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                      float* __restrict__ sliced) {
  const int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (n >= N) return;
  const int level = blockIdx.y;
  float pos[P];
  load_pos<P>(positions, n, pos);
  if (level >= L) {  // pseudo-levels carrying the scaled input point (zero padded)
    const int e = level - L;
#pragma unroll
    for (int f = 0; f < F; f++) {
      const int d = e * F + f;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < P; i++)
        if (i == d) v = pos[i] * points_scaling;
      sliced[((int64_t)level * F + f) * N + n] = v;
    }
    return;
  }
  Simplex<P> s;
  compute_simplex<P>(pos, shifts + level * P, scale_factor + level * P, s);
  const float w = window[level];
  const float* __restrict__ table = lattice + (int64_t)level * capacity * F;
  // issue all P+1 gathers before consuming them (independent 8-B loads in flight)
  uint32_t row[P + 1];
#pragma unroll
  for (int r = 0; r <= P; r++) row[r] = vertex_row<P>(s, r, capacity);
  float fv[P + 1][F];
#pragma unroll
  for (int r = 0; r <= P; r++) {
    if (F == 2) {
      float2 t = *reinterpret_cast<const float2*>(table + (int64_t)row[r] * 2);
      fv[r][0] = t.x;
      fv[r][F - 1] = t.y;
    } else {
#pragma unroll
      for (int f = 0; f < F; f++) fv[r][f] = table[(int64_t)row[r] * F + f];
    }
  }
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; f++) acc[f] = 0.f;
#pragma unroll
  for (int r = 0; r <= P; r++) {
    const float bw = s.bary[r] * w;
#pragma unroll
    for (int f = 0; f < F; f++) acc[f] = acc[f] + fv[r][f] * bw;
  }
#pragma unroll
  for (int f = 0; f < F; f++) sliced[((int64_t)level * F + f) * N + n] = acc[f];
}


// ------------------------------------------------------------------- LDS-privatised scatter-add
// The lattice gradient is a scatter-add of (P+1)*F floats per (point, level).  At coarse levels the whole
// batch lands on a few dozen table rows, and fp32 atomics to one address serialise (measured: 135 ms for
// 2M points x 16 levels with plain global atomics).  Each workgroup therefore owns a direct-mapped cache
// in LDS (tag = table row, F partial sums) that it fills with LDS atomics while it walks its share of the
// points of one level, and flushes with one global atomic per live entry at the end.  A row whose slot is
// taken by another row bypasses the cache (plain global atomic), so the structure is exact for any
// distribution; a workgroup whose hit rate is poor after its first tiles (fine, fully hashed levels)
// switches the cache off for the rest of its walk.
constexpr uint32_t SC_EMPTY = 0xFFFFFFFFu;

template <int F>
struct ScatterCache {
  static constexpr int SC_SLOTS = 8192 / F;  // F=2: 16 KiB tags + 32 KiB sums
  uint32_t* tags;
  float* sums;
  int* stats;  // [0] = hits, [1] = tries, [2] = enabled
  __device__ __forceinline__ void init(float* lds) {
    tags = reinterpret_cast<uint32_t*>(lds);
    sums = lds + SC_SLOTS;
    stats = reinterpret_cast<int*>(lds + SC_SLOTS + SC_SLOTS * F);
    for (int i = threadIdx.x; i < SC_SLOTS; i += blockDim.x) tags[i] = SC_EMPTY;
    for (int i = threadIdx.x; i < SC_SLOTS * F; i += blockDim.x) sums[i] = 0.f;
    if (threadIdx.x == 0) {
      stats[0] = 0;
      stats[1] = 0;
      stats[2] = 1;
    }
    __syncthreads();
  }
  // returns true when the contribution was absorbed by the cache
  __device__ __forceinline__ bool add(uint32_t row, const float* v) {
    const uint32_t slot = (row ^ (row >> 12)) & (SC_SLOTS - 1);
    uint32_t old = tags[slot];
    if (old == SC_EMPTY) old = atomicCAS(&tags[slot], SC_EMPTY, row);
    if (old == SC_EMPTY || old == row) {
#pragma unroll
      for (int f = 0; f < F; f++) atomicAdd(&sums[slot * F + f], v[f]);
      return true;
    }
    return false;
  }
  __device__ __forceinline__ void flush(float* __restrict__ table_grad) {
    __syncthreads();
    for (int i = threadIdx.x; i < SC_SLOTS; i += blockDim.x) {
      const uint32_t row = tags[i];
      if (row != SC_EMPTY) {
#pragma unroll
        for (int f = 0; f < F; f++) {
          const float v = sums[i * F + f];
          if (v != 0.f) atomicAdd(table_grad + (int64_t)row * F + f, v);
        }
      }
    }
  }
  static constexpr size_t bytes() { return (size_t)(SC_SLOTS + SC_SLOTS * F + 4) * 4; }
};

// ----------------------------------------------------------------------------------------- backward
// grad_lattice[l][row][f] += bary_r * w_l * g[l][f][n]            (LDS-privatised, then fp32 L2 atomics)
// grad_pos[n][i]          += dL/dpos_i  (chain through barycentric -> elevated -> position)
// Launch: grid (B, Lt), workgroup b of level l walks point tiles b, b+B, ...
template <int F>
__device__ __forceinline__ bool cache_vote(ScatterCache<F>& sc, int hits, int tries) {
  // called by every thread of the workgroup after its second tile; returns the workgroup-uniform decision
  hits = (int)psdf::wave_sum((float)hits);
  tries = (int)psdf::wave_sum((float)tries);
  if (psdf::lane_id() == 0) {
    atomicAdd(&sc.stats[0], hits);
    atomicAdd(&sc.stats[1], tries);
  }
  __syncthreads();
  return sc.stats[0] * 2 >= sc.stats[1];
}

template <int P, int F, bool LATTICE, bool POS>
__global__ void __launch_bounds__(PSDF_BLOCK)
    encode_bwd_kernel(int64_t N, int L, uint32_t capacity, const float* __restrict__ positions,
                      const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                      const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                      const float* __restrict__ grad_sliced, float* __restrict__ grad_lattice,
                      float* __restrict__ grad_positions) {
  extern __shared__ __align__(16) float lds[];
  const int level = blockIdx.y;
  const int64_t ntiles = (N + PSDF_BLOCK - 1) / PSDF_BLOCK;
  if (level >= L) {
    if (POS) {
      const int e = level - L;
      for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
        if (n >= N) continue;
#pragma unroll
        for (int f = 0; f < F; f++) {
          const int d = e * F + f;
          if (d < P)
            atomicAdd(grad_positions + n * P + d, grad_sliced[((int64_t)level * F + f) * N + n] * points_scaling);
        }
      }
    }
    return;
  }
  ScatterCache<F> sc;
  if (LATTICE) sc.init(lds);
  bool use_cache = LATTICE;
  int hits = 0, tries = 0, iter = 0;
  const float w = window[level];
  const int64_t tbase = (int64_t)level * capacity * F;
  float sfl[P], shl[P];
#pragma unroll
  for (int i = 0; i < P; i++) {
    sfl[i] = scale_factor[level * P + i];
    shl[i] = shifts[level * P + i];
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, iter++) {
    const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
    if (n < N) {
      float g[F];
#pragma unroll
      for (int f = 0; f < F; f++) g[f] = grad_sliced[((int64_t)level * F + f) * N + n];
      float pos[P];
      load_pos<P>(positions, n, pos);
      Simplex<P> s;
      compute_simplex<P>(pos, shl, sfl, s);
      float dbary[P + 2];
#pragma unroll
      for (int k = 0; k <= P + 1; k++) dbary[k] = 0.f;
#pragma unroll
      for (int r = 0; r <= P; r++) {
        const uint32_t row = vertex_row<P>(s, r, capacity);
        if (LATTICE) {
          const float bw = s.bary[r] * w;
          float v[F];
#pragma unroll
          for (int f = 0; f < F; f++) v[f] = g[f] * bw;
          bool absorbed = false;
          if (use_cache) {
            absorbed = sc.add(row, v);
            hits += absorbed;
            tries++;
          }
          if (!absorbed) {
#pragma unroll
            for (int f = 0; f < F; f++) atomicAdd(grad_lattice + tbase + (int64_t)row * F + f, v[f]);
          }
        }
        if (POS) {
#pragma unroll
          for (int f = 0; f < F; f++) dbary[r] = dbary[r] + lattice[tbase + (int64_t)row * F + f] * w * g[f];
        }
      }
      if (POS) {
        dbary[P + 1] = dbary[P + 1] + dbary[0];  // adjoint of bary[0] += 1 + bary[P+1]
        float dE[P + 1];
        const float invp = 1.0f / (P + 1);
#pragma unroll
        for (int i = 0; i <= P; i++) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int k = 0; k <= P + 1; k++) {
            if (k == P - s.rank[i]) a = dbary[k];
            if (k == P + 1 - s.rank[i]) b = dbary[k];
          }
          dE[i] = (a - b) * invp;
        }
#pragma unroll
        for (int i = 0; i < P; i++) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j <= i; j++) acc = acc + dE[j];
          acc = acc - dE[i + 1] * (float)(i + 1);
          atomicAdd(grad_positions + n * P + i, acc * sfl[i]);
        }
      }
    }
    if (LATTICE && iter == 1) use_cache = cache_vote<F>(sc, hits, tries);
  }
  if (LATTICE) sc.flush(grad_lattice + tbase);
}

// ---------------------------------------------------------------------------------- double backward
// Inputs: u = dL/d(grad_positions) [N,P], g = grad_sliced [Lt*F,N].  grad_positions is bilinear in
// (g, lattice) for a fixed simplex, so
//   q_r = sum_i u_i * d bary_r / d pos_i        (directional derivative of the barycentrics along u)
//   grad_lattice[l][row_r][f] += q_r * w_l * g[l][f][n]
//   grad_g[l][f][n]            = sum_r q_r * w_l * lattice[l][row_r][f]
// and for the concatenated-point channels grad_g = u_d * points_scaling.
template <int P, int F, bool LATTICE>
__global__ void __launch_bounds__(PSDF_BLOCK)
    encode_dbl_bwd_kernel(int64_t N, int L, uint32_t capacity, const float* __restrict__ positions,
                          const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                          const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                          const float* __restrict__ dd_positions, const float* __restrict__ grad_sliced,
                          float* __restrict__ grad_lattice, float* __restrict__ grad_grad_sliced) {
  extern __shared__ __align__(16) float lds[];
  const int level = blockIdx.y;
  const int64_t ntiles = (N + PSDF_BLOCK - 1) / PSDF_BLOCK;
  if (level >= L) {
    const int e = level - L;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
      if (n >= N) continue;
      float u[P];
      load_pos<P>(dd_positions, n, u);
#pragma unroll
      for (int f = 0; f < F; f++) {
        const int d = e * F + f;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < P; i++)
          if (i == d) v = u[i] * points_scaling;
        grad_grad_sliced[((int64_t)level * F + f) * N + n] = v;
      }
    }
    return;
  }
  ScatterCache<F> sc;
  if (LATTICE) sc.init(lds);
  bool use_cache = LATTICE;
  int hits = 0, tries = 0, iter = 0;
  const float w = window[level];
  const int64_t tbase = (int64_t)level * capacity * F;
  float sfl[P], shl[P];
#pragma unroll
  for (int i = 0; i < P; i++) {
    sfl[i] = scale_factor[level * P + i];
    shl[i] = shifts[level * P + i];
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, iter++) {
    const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
    if (n < N) {
      float u[P], pos[P];
      load_pos<P>(dd_positions, n, u);
      load_pos<P>(positions, n, pos);
      Simplex<P> s;
      compute_simplex<P>(pos, shl, sfl, s);
      // adjoint of pos -> elevated
      float aE[P + 1];
#pragma unroll
      for (int j = 0; j <= P; j++) aE[j] = 0.f;
#pragma unroll
      for (int k = 0; k < P; k++) {
        const float us = u[k] * sfl[k];
#pragma unroll
        for (int j = 0; j <= k; j++) aE[j] = aE[j] + us;
        aE[k + 1] = aE[k + 1] - us * (float)(k + 1);
      }
      // adjoint of elevated -> barycentric slots
      float q[P + 2];
#pragma unroll
      for (int k = 0; k <= P + 1; k++) q[k] = 0.f;
      const float invp = 1.0f / (P + 1);
#pragma unroll
      for (int i = 0; i <= P; i++) {
        const float t = aE[i] * invp;
#pragma unroll
        for (int k = 0; k <= P + 1; k++) {
          if (k == P - s.rank[i]) q[k] = q[k] + t;
          if (k == P + 1 - s.rank[i]) q[k] = q[k] - t;
        }
      }
      q[0] = q[0] + q[P + 1];
      float g[F], gg[F];
#pragma unroll
      for (int f = 0; f < F; f++) {
        g[f] = grad_sliced[((int64_t)level * F + f) * N + n];
        gg[f] = 0.f;
      }
#pragma unroll
      for (int r = 0; r <= P; r++) {
        const uint32_t row = vertex_row<P>(s, r, capacity);
        const float qw = q[r] * w;
        if (LATTICE) {
          float v[F];
#pragma unroll
          for (int f = 0; f < F; f++) v[f] = qw * g[f];
          bool absorbed = false;
          if (use_cache) {
            absorbed = sc.add(row, v);
            hits += absorbed;
            tries++;
          }
          if (!absorbed) {
#pragma unroll
            for (int f = 0; f < F; f++) atomicAdd(grad_lattice + tbase + (int64_t)row * F + f, v[f]);
          }
        }
#pragma unroll
        for (int f = 0; f < F; f++) gg[f] = gg[f] + qw * lattice[tbase + (int64_t)row * F + f];
      }
#pragma unroll
      for (int f = 0; f < F; f++) grad_grad_sliced[((int64_t)level * F + f) * N + n] = gg[f];
    }
    if (LATTICE && iter == 1) use_cache = cache_vote<F>(sc, hits, tries);
  }
  if (LATTICE) sc.flush(grad_lattice + tbase);
}

inline int extra_levels(int P, int F, int concat) { return concat ? (P + F - 1) / F : 0; }

}  // namespace

// ================================================================================== C ABI
extern "C" {

int psdf_encode_forward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                        const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                        int concat_points, float points_scaling, float* sliced, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !sliced) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + extra_levels(pos_dim, nr_feat, concat_points);
  dim3 grid(psdf_blocks(N, PSDF_BLOCK), Lt);
#define FWD(P_, F_)                                                                                            \
  hipLaunchKernelGGL((encode_fwd_kernel<P_, F_>), grid, dim3(PSDF_BLOCK), 0, st, N, nr_levels, (uint32_t)capacity, \
                     positions, lattice, scale_factor, shifts, window, points_scaling, sliced)
  if (pos_dim == 3 && nr_feat == 2)
    FWD(3, 2);
  else if (pos_dim == 4 && nr_feat == 2)
    FWD(4, 2);
  else if (pos_dim == 2 && nr_feat == 2)
    FWD(2, 2);
  else if (pos_dim == 3 && nr_feat == 4)
    FWD(3, 4);
  else
    return PSDF_ERR_UNSUPPORTED;
#undef FWD
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// grad_lattice / grad_positions must be zero-initialised by the caller (or hold a running sum to add to);
// either may be NULL to skip that gradient.
int psdf_encode_backward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                         const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                         int concat_points, float points_scaling, const float* grad_sliced, float* grad_lattice,
                         float* grad_positions, void* stream) {
  if (N == 0 || (!grad_lattice && !grad_positions)) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !grad_sliced) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + ((grad_positions != nullptr) ? extra_levels(pos_dim, nr_feat, concat_points) : 0);
  const unsigned nb = psdf_blocks(N, PSDF_BLOCK);
  dim3 grid(nb < 512u ? nb : 512u, Lt);
#define BWD(P_, F_, A_, B_)                                                                                     \
  hipLaunchKernelGGL((encode_bwd_kernel<P_, F_, A_, B_>), grid, dim3(PSDF_BLOCK),                                \
                     (A_) ? ScatterCache<F_>::bytes() : 0, st, N, nr_levels, (uint32_t)capacity, positions,      \
                     lattice, scale_factor, shifts, window, points_scaling, grad_sliced, grad_lattice,           \
                     grad_positions)
#define BWD_PF(P_, F_)                  \
  do {                                  \
    if (grad_lattice && grad_positions) \
      BWD(P_, F_, true, true);          \
    else if (grad_lattice)              \
      BWD(P_, F_, true, false);         \
    else                                \
      BWD(P_, F_, false, true);         \
  } while (0)
  if (pos_dim == 3 && nr_feat == 2)
    BWD_PF(3, 2);
  else if (pos_dim == 4 && nr_feat == 2)
    BWD_PF(4, 2);
  else if (pos_dim == 2 && nr_feat == 2)
    BWD_PF(2, 2);
  else if (pos_dim == 3 && nr_feat == 4)
    BWD_PF(3, 4);
  else
    return PSDF_ERR_UNSUPPORTED;
#undef BWD_PF
#undef BWD
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// grad_lattice must be zero-initialised (or NULL to skip); grad_grad_sliced is fully overwritten.
int psdf_encode_double_backward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity,
                                const float* positions, const float* lattice, const float* scale_factor,
                                const float* shifts, const float* window, int concat_points, float points_scaling,
                                const float* dd_positions, const float* grad_sliced, float* grad_lattice,
                                float* grad_grad_sliced, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !dd_positions || !grad_sliced ||
      !grad_grad_sliced)
    return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + extra_levels(pos_dim, nr_feat, concat_points);
  const unsigned nb = psdf_blocks(N, PSDF_BLOCK);
  dim3 grid(nb < 512u ? nb : 512u, Lt);
#define DBL(P_, F_)                                                                                              \
  do {                                                                                                           \
    if (grad_lattice)                                                                                            \
      hipLaunchKernelGGL((encode_dbl_bwd_kernel<P_, F_, true>), grid, dim3(PSDF_BLOCK), ScatterCache<F_>::bytes(), \
                         st, N, nr_levels, (uint32_t)capacity, positions, lattice, scale_factor, shifts, window,   \
                         points_scaling, dd_positions, grad_sliced, grad_lattice, grad_grad_sliced);               \
    else                                                                                                         \
      hipLaunchKernelGGL((encode_dbl_bwd_kernel<P_, F_, false>), grid, dim3(PSDF_BLOCK), 0, st, N, nr_levels,      \
                         (uint32_t)capacity, positions, lattice, scale_factor, shifts, window, points_scaling,     \
                         dd_positions, grad_sliced, grad_lattice, grad_grad_sliced);                               \
  } while (0)
  if (pos_dim == 3 && nr_feat == 2)
    DBL(3, 2);
  else if (pos_dim == 4 && nr_feat == 2)
    DBL(4, 2);
  else if (pos_dim == 2 && nr_feat == 2)
    DBL(2, 2);
  else if (pos_dim == 3 && nr_feat == 4)
    DBL(3, 4);
  else
    return PSDF_ERR_UNSUPPORTED;
#undef DBL
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // extern "C"
