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
#include <cstdlib>
#include <cstdio>
#include "encode_device.h"
#include <vector>
#include <mutex>
#include <cstring>

namespace {

// ------------------------------------------------------------------------------------------ forward
template <int P, int F, bool PLAIN = false>   // PLAIN: no skip mask, no touched-block map (their arguments are ignored)
__global__ void __launch_bounds__(PSDF_BLOCK)
    encode_fwd_kernel(int64_t N, int L, uint32_t capacity, EncConv conv, const float* __restrict__ positions,
                      const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                      const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                      int pad_points, const unsigned char* __restrict__ skip, float* __restrict__ sliced,
                      unsigned char* __restrict__ touched, int touch_shift, int blocks_per_level) {
  // (Round 4 measured an XCD-pinned shape -- a 1-D grid whose workgroup id i lands on XCD i % 8, each XCD taking two or three
  // levels for ALL points so that a table is fetched into one L2 instead of eight: 0.358 -> 0.631 ms at 2 M points, 16 levels
  // (profiles/r04_enc_ab.jsonl).  Eight XCDs streaming eight different tables each touch every position and write every
  // output row with 1/8 of the chip; the level-at-a-time sweep below keeps all 256 CUs on one 2-MiB table.)
  const int level = blockIdx.y;
  const int64_t ntiles = (N + PSDF_BLOCK - 1) / PSDF_BLOCK;
  // a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its level (gridDim.x = the number of tiles unless the
  // launcher caps it: then the level's constants are fetched once per workgroup instead of once per tile)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
  if (n >= N) continue;
  if (!PLAIN && skip && skip[n]) continue;  // masked point (fixed-shape callers, e.g. converged rays): its columns stay untouched
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
      // channel L*F + d; d >= P exists only in the zero-padded pseudo-level layout (encode_conventions.h)
      if (d < P || pad_points) sliced[((int64_t)level * F + f) * N + n] = v;
    }
    continue;
  }
  const float w = window[level];
  if (w == 0.f) {  // a level the coarse-to-fine window keeps closed (workgroup-uniform): its channels are 0 * (finite rows) = 0,
                   // no simplex, no gathers, nothing to mark (its backward contributes nothing either)
#pragma unroll
    for (int f = 0; f < F; f++) sliced[((int64_t)level * F + f) * N + n] = 0.f;
    continue;
  }
  Simplex<P> s;
  compute_simplex<P>(pos, shifts + level * P, scale_factor + level * P, s, conv.tie_later);
  const float* __restrict__ table = lattice + (int64_t)level * capacity * F;
  // issue all P+1 gathers before consuming them (independent 8-B loads in flight)
  uint32_t row[P + 1];
  vertex_rows<P>(s, capacity, row, conv.hash_c);
  // Training forward: remember which blocks of table rows this batch reads.  Every lattice-gradient contribution of the
  // backward / double backward at these positions lands on exactly these rows, so the optimiser can skip blocks whose
  // gradient and moments are still exactly zero (optim.hip: adamw_blocks_kernel) -- a superset is all it needs.
  if (!PLAIN && touched) {
#pragma unroll
    for (int r = 0; r <= P; r++) touched[(int64_t)level * blocks_per_level + (row[r] >> touch_shift)] = 1;
  }
  float fv[P + 1][F];
#pragma unroll
  for (int r = 0; r <= P; r++) {
    // 32-bit byte offset from the level's table (a wave-uniform base): one shift per gather instead of a 64-bit shift-add
    // (a level of the table is below 4 GiB: checked by the launcher)
    const char* tb = reinterpret_cast<const char*>(table);
    if (F == 2) {
      float2 t = *reinterpret_cast<const float2*>(tb + (row[r] * 8u));
      fv[r][0] = t.x;
      fv[r][F - 1] = t.y;
    } else {
#pragma unroll
      for (int f = 0; f < F; f++) fv[r][f] = *reinterpret_cast<const float*>(tb + (row[r] * (uint32_t)(F * 4) + (uint32_t)(f * 4)));
    }
  }
  float acc[F];
#pragma unroll
  for (int r = 0; r <= P; r++) {
    const float bw = s.bary[r] * w;
#pragma unroll
    for (int f = 0; f < F; f++) acc[f] = r == 0 ? fv[r][f] * bw : acc[f] + fv[r][f] * bw;   // (0 + x: the sign of a zero only)
  }
#pragma unroll
  for (int f = 0; f < F; f++) sliced[((int64_t)level * F + f) * N + n] = acc[f];
  }
}


// ------------------------------------------------------------------- LDS-privatised scatter-add
// The lattice gradient is a scatter-add of (P+1)*F floats per (point, level).  At coarse levels the whole
// batch lands on a few dozen table rows, and fp32 atomics to one address serialise (measured: 135 ms for
// 2M points x 16 levels with plain global atomics).  Each workgroup therefore owns a direct-mapped cache
// in LDS (tag = table row, F partial sums) that it fills with LDS atomics while it walks its share of the
// points of one level, and flushes with one global atomic per live entry at the end.  A row whose slot is
// taken by another row bypasses the cache (plain global atomic), so the structure is exact for any
// distribution; a workgroup whose hit rate is poor after its first tiles (fine, fully hashed levels)
// switches the cache off for the rest of its walk (vote after its first tile: share of contributions that hit an
// entry which already existed).
constexpr uint32_t SC_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t Q_NONE = 0xFFFFFFFFu;   // "no contribution" in a thread's list of rows (rows are < capacity <= 2^31)
#if !defined(PSDF_ENC_QSPT)
#define PSDF_ENC_QSPT 2
#endif
constexpr int QUEUE_SPT = PSDF_ENC_QSPT;   // points per thread and super-tile in queue mode (see ScatterCache)

// LDS float accumulation without ds_add_f32.  Measured on this chip (tools/atomic_bench.hip): a wave-wide ds_add_f32
// costs ~197 cycles whatever the access pattern (1 lane per ~3 cycles), while INTEGER LDS atomics run at LDS speed
// (ds_add_u32 10 cycles, ds_cmpst_rtn_b32 12, ds_add_u64 16, plain 8-byte read-modify-write 18).  So a pair of
// floats is added with one 64-bit compare-and-swap: read, add in registers, ds_cmpst_rtn_b64; lanes that lose the
// race (another lane or wave changed the pair) retry up to 3 times, and whatever is still pending after that -- heavy
// same-address contention, where the constant-cost float atomic is the better tool -- falls back to ds_add_f32.
// `hot` remembers that fallback per thread for a few adds so that a contended stream stops paying for doomed attempts.
__device__ __forceinline__ void lds_add_pair(float* p, float a, float b, int& hot) {
  if (hot == 0) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    unsigned long long old = *q;
#pragma unroll
    for (int it = 0; it < 4; it++) {  // k lanes of a wave on one pair need k rounds: covers multiplicity <= 4
      const float x = __uint_as_float((uint32_t)old) + a;
      const float y = __uint_as_float((uint32_t)(old >> 32)) + b;
      const unsigned long long want = (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32);
      const unsigned long long prev = atomicCAS(q, old, want);
      if (prev == old) return;
      old = prev;
    }
    hot = 8;  // heavily shared pair: use the float atomic for the next few adds, then probe again
  } else {
    hot--;
  }
  atomicAdd(p, a);
  atomicAdd(p + 1, b);
}

// Adjacent lanes are adjacent samples of a ray, and at coarse and medium levels they sit in the same simplex: their
// contributions go to the same rows.  Such runs of equal rows are summed in registers (segmented inclusive scan
// inside each 16-lane DPP row, Hillis-Steele with head flags) and only the LAST lane of a run hands the sum on, so
// that what reaches LDS has a multiplicity of at most 4 per wave -- within reach of the retries of lds_add_pair.
// Returns true on the lanes that own a (partial) run sum.
template <int F>
__device__ __forceinline__ bool combine_runs16(uint32_t key, float (&v)[F]) {
  constexpr int ROW_SHR = 0x110, ROW_SHL = 0x100;
  // key Q_NONE: a lane without a contribution (v = 0).  Such lanes may join each other -- they add zeros and never own a run.
  const int k = (int)key;
  // Every cross-lane move uses bound_ctrl: a lane that would read across the edge of its 16-lane row receives 0 from the
  // instruction itself (otherwise the compiler preloads each destination with a v_mov: 10 per contribution).  The flag is
  // therefore kept INVERTED -- j = 1: "joined to the run of the lane before" -- so that the 0 shifted in means "a run starts
  // here".  (The first lane of a row may read j = 1 against the shifted-in key 0; everything it then adds is a shifted-in 0.)
  const int kprev = __builtin_amdgcn_update_dpp(0, k, ROW_SHR | 1, 0xf, 0xf, true);
  const float joined = (kprev == k) ? 1.0f : 0.0f;
  // The flag is a float 1.0 / 0.0 and the conditional add is ONE fused multiply-add per feature whose shifted operand comes
  // through DPP (v_fmac_f32_dpp: v = shifted(v) * j + v; exact -- a product by 1 or 0 -- instead of v_add_dpp + v_cmp + v_cndmask),
  // the flag update one v_mul_f32_dpp: 3 instructions per step for F = 2, was 7.  (A non-finite contribution now also spoils the
  // lanes of its row that are NOT in its run, 0 x inf; such a step is lost either way.)
  float j = joined;
#define PSDF_SCAN_STEP(D)                                                                                        \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < F; i++) v[i] = __builtin_fmaf(                                         \
        __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), ROW_SHR | D, 0xf, 0xf, true)), j, v[i]); \
    j = j * __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(j), ROW_SHR | D, 0xf, 0xf, true));      \
  }
  PSDF_SCAN_STEP(1)
  PSDF_SCAN_STEP(2)
  PSDF_SCAN_STEP(4)
  PSDF_SCAN_STEP(8)
#undef PSDF_SCAN_STEP
  // the lane after: joined to this one (0 shifted in past the end of the row: a run ends with its row)
  const int next_joined = __builtin_amdgcn_update_dpp(0, __float_as_int(joined), ROW_SHL | 1, 0xf, 0xf, true);
  return key != Q_NONE && next_joined == 0;
}

// TOTAL = slots x F.  8192 (48 KiB for F = 2) where the cache is all there is between the contributions and global float
// atomics (small batches); 4096 (24 KiB) in queue mode, where what the cache does not absorb goes to the queues anyway:
// there the binning kernel was held at 3 waves per SIMD by both its LDS and its registers while it spent 48 % of its wave
// cycles waiting (profiles/r02_pmc_sq_encode_bwd.txt) -- half the cache and 2 instead of 4 points per thread and super-tile
// (half the per-thread contribution registers) let more waves in: binning + reduce 0.894 -> 0.828 ms on the bench batch.
template <int F, int TOTAL = 8192>
struct ScatterCache {
  static constexpr int SC_SLOTS = TOTAL / F;  // F=2, TOTAL 8192: 16 KiB tags + 32 KiB sums
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
  // 0: slot taken by another row (caller adds to HBM / queues); 1: absorbed, claimed an empty slot;
  // 2: absorbed into an entry that already existed (a genuine re-use: what the adaptive vote counts)
  __device__ __forceinline__ int add(uint32_t row, const float* v, int& hot) {
    const uint32_t slot = (row ^ (row >> 12)) & (SC_SLOTS - 1);
    uint32_t old = tags[slot];
    if (old == SC_EMPTY) old = atomicCAS(&tags[slot], SC_EMPTY, row);
    if (old == SC_EMPTY || old == row) {
      static_assert(F % 2 == 0, "pairs");
#pragma unroll
      for (int f = 0; f < F; f += 2) lds_add_pair(&sums[slot * F + f], v[f], v[f + 1], hot);
      return old == row ? 2 : 1;
    }
    return 0;
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
template <typename SC>
__device__ __forceinline__ bool cache_vote(SC& sc, int hits, int tries) {
  // called by every thread of the workgroup after its first tile; returns the workgroup-uniform decision
  hits = (int)psdf::wave_sum((float)hits);
  tries = (int)psdf::wave_sum((float)tries);
  if (psdf::lane_id() == 0) {
    atomicAdd(&sc.stats[0], hits);
    atomicAdd(&sc.stats[1], tries);
  }
  __syncthreads();
  return sc.stats[0] * 8 >= sc.stats[1];  // keep the cache when >= 1/8 of the first tile hit an existing entry
}

// Binned hand-off to the LDS reduction (large batches, fully hashed levels).  fp32 global atomics are capped at
// ~21 G/s on this chip whatever the placement (tools/atomic_bench.hip) while LDS atomics sustain >= 185 G/s, so
// the contributions of a workgroup that is NOT served by its scatter cache are appended to per-(level, partition)
// queues instead of being added to HBM one atomic at a time: partition p owns table rows [p*RPP, (p+1)*RPP),
// RPP*F floats = 128 KiB = one workgroup's LDS.  Slots are reserved per tile with LDS counters and ONE global
// atomic per (tile, partition); encode_bwd_reduce_kernel then folds every queue into LDS and adds the slice to
// the gradient with plain stores.  A queue that is full falls back to the global atomic (exact either way).
struct Queues {
  uint16_t* rows;  // [L, NP, cap]      row index inside the partition
  float* vals;     // [L, NP, cap, F]
  int* tails;      // [L, NP]           zeroed per call
  int cap, np, shift;  // rows per partition = 1 << shift
};
constexpr int Q_MAX_PARTS = 64;
constexpr int Q_LDS_INTS = 4 * Q_MAX_PARTS + 4;   // q_cnt, q_base, q_off (+1), q_gd; 3 vote words

// Workgroup-collective append of up to NC contributions per thread to the level's partition queues.
// Slots are reserved with an LDS counter per partition and ONE global atomic per (call, partition).
// Addressing (round 5: the binning kernel is VALU bound, profiles/r05_pmc_sq_encode_bwd.txt, and the 64-bit entry offset
// ((level * np + part) * cap + idx) took 12 vector instructions per contribution): the level's slice of the queues is a
// wave-uniform base pointer; the entry offset inside it is 32 bits (queue_plan refuses plans whose level slice exceeds 2^32
// bytes) and comes out of LDS ready made -- next to a partition's reserved tail (q_base[part], for the "queue full" test) sits
// part * cap + tail (q_aux[part], 64 ints further: one ds_read2_b32 fetches both) -- so that a contribution needs two adds.
struct LevelQueues {
  char* rows;   // Q.rows + level * np * cap                [np, cap] uint16
  char* vals;   // Q.vals + level * np * cap * F            [np, cap, F] float
  uint32_t mask;   // rows per partition - 1
};
template <int F>
__device__ __forceinline__ LevelQueues level_queues(const Queues& Q, int level) {
  const int64_t e = (int64_t)level * Q.np * Q.cap;
  return LevelQueues{reinterpret_cast<char*>(Q.rows + e), reinterpret_cast<char*>(Q.vals + e * F), (1u << Q.shift) - 1u};
}
template <int F>
__device__ __forceinline__ void queue_store(const LevelQueues& LQ, uint32_t o, uint32_t row, const float* v) {
  *reinterpret_cast<uint16_t*>(LQ.rows + (o * 2u)) = (uint16_t)(row & LQ.mask);
  if (F == 2) {
    *reinterpret_cast<float2*>(LQ.vals + (o * 8u)) = make_float2(v[0], v[F - 1]);
  } else {
#pragma unroll
    for (int f = 0; f < F; f++) *reinterpret_cast<float*>(LQ.vals + (o * (uint32_t)(F * 4) + (uint32_t)(f * 4))) = v[f];
  }
}

template <int NC, int F>
__device__ __forceinline__ void queue_push(const Queues& Q, int level, int* q_cnt, int* q_base,
                                           const uint32_t (&crow)[NC], const float (&cval)[NC][F],
                                           float* __restrict__ table_grad) {
  int* q_aux = q_base + Q_MAX_PARTS;
  const LevelQueues LQ = level_queues<F>(Q, level);
  if (threadIdx.x < Q.np) q_cnt[threadIdx.x] = 0;
  __syncthreads();
  int slot[NC];
#pragma unroll
  for (int r = 0; r < NC; r++)
    if (crow[r] != Q_NONE) slot[r] = atomicAdd(&q_cnt[crow[r] >> Q.shift], 1);
  __syncthreads();
  if (threadIdx.x < Q.np) {
    const int c = q_cnt[threadIdx.x];
    const int base = c ? atomicAdd(&Q.tails[level * Q.np + threadIdx.x], c) : 0;
    q_base[threadIdx.x] = base;
    q_aux[threadIdx.x] = (int)threadIdx.x * Q.cap + base;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NC; r++) {
    if (crow[r] == Q_NONE) continue;
    const int part = crow[r] >> Q.shift;
    const int idx = q_base[part] + slot[r];
    if (idx < Q.cap) {
      queue_store<F>(LQ, (uint32_t)(q_aux[part] + slot[r]), crow[r], cval[r]);
    } else {  // queue full: exact fallback
#pragma unroll
      for (int f = 0; f < F; f++) atomicAdd(table_grad + (int64_t)crow[r] * F + f, cval[r][f]);
    }
  }
}

// Same hand-off with COALESCED stores, for workgroups whose scatter cache is switched off (its 48 KiB of LDS are free):
// a direct append makes every lane of a store instruction hit a different partition's queue, and the address
// processing of such scattered 2-/8-byte stores (one lane per cycle per CU) was the limiter of the binning kernel.
// Here the contributions of the super-tile are first laid out in LDS grouped by partition (counting sort: the slot
// inside the group comes from the same LDS counter that reserves the queue segment), then written out linearly, so a
// wave writes 64 consecutive queue entries.
template <int NC>
__device__ __forceinline__ void queue_push_staged(const Queues& Q, int level, int* q_cnt, int* q_base, int* q_off, int* q_gd,
                                                  float* stage, const uint32_t (&crow)[NC],
                                                  const float (&cval)[NC][2], float* __restrict__ table_grad) {
  uint32_t* st_row = reinterpret_cast<uint32_t*>(stage);                 // [NC * PSDF_BLOCK]
  float2* st_val = reinterpret_cast<float2*>(stage + NC * PSDF_BLOCK);   // [NC * PSDF_BLOCK]
  const LevelQueues LQ = level_queues<2>(Q, level);
  if (threadIdx.x < Q.np) q_cnt[threadIdx.x] = 0;
  __syncthreads();
  int slot[NC];
#pragma unroll
  for (int r = 0; r < NC; r++)
    if (crow[r] != Q_NONE) slot[r] = atomicAdd(&q_cnt[crow[r] >> Q.shift], 1);
  __syncthreads();
  if (threadIdx.x < 64) {   // the first wave: segment offsets by a wave scan (np <= 64).  A serial walk over the counters by
    // thread p (p LDS reads in a dependent chain, up to 31 of them, inside a barrier-separated phase) cost 4.5 % of the pair.
    const int lane = threadIdx.x;
    const int c = lane < Q.np ? q_cnt[lane] : 0;
    const int incl = psdf::wave_incl_scan_add_i(c);
    if (lane < Q.np) {
      // staged entry i of partition p goes to queue slot tail_p + (i - off_p): what the write-out loop needs per entry is
      // i + (tail_p - off_p) for the "queue full" test and i + (p * cap + tail_p - off_p) as the offset in the level's slice
      // (q_gd: written here, behind two barriers of this call, and read by the write-out loop of this call only -- so the
      // next call may start while slower waves are still in that loop, no barrier at the end)
      const int base = c ? atomicAdd(&Q.tails[level * Q.np + lane], c) : 0;
      const int off = incl - c;
      q_off[lane] = off;
      q_base[lane] = base - off;
      q_gd[lane] = lane * Q.cap + base - off;
      if (lane == Q.np - 1) q_off[Q.np] = incl;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NC; r++)
    if (crow[r] != Q_NONE) {
      const int i = q_off[crow[r] >> Q.shift] + slot[r];
      st_row[i] = crow[r];
      st_val[i] = make_float2(cval[r][0], cval[r][1]);
    }
  __syncthreads();
  const int total = q_off[Q.np];
  for (int i = threadIdx.x; i < total; i += PSDF_BLOCK) {
    const uint32_t row = st_row[i];
    const float2 v = st_val[i];
    const int part = row >> Q.shift;
    const int idx = i + q_base[part];
    if (idx < Q.cap) {
      const float vv[2] = {v.x, v.y};
      queue_store<2>(LQ, (uint32_t)(i + q_gd[part]), row, vv);
    } else {  // queue full: exact fallback
      atomicAdd(table_grad + (int64_t)row * 2, v.x);
      atomicAdd(table_grad + (int64_t)row * 2 + 1, v.y);
    }
  }
}

// The scatter cache was switched off: hand its live entries to the queues too (instead of flushing them with one
// global atomic each at the end) and empty it.
template <int F, typename SC>
__device__ __forceinline__ void cache_drain_to_queue(SC& sc, const Queues& Q, int level, int* q_cnt,
                                                     int* q_base, float* __restrict__ table_grad) {
  for (int base = 0; base < SC::SC_SLOTS; base += PSDF_BLOCK) {
    const int i = base + threadIdx.x;
    uint32_t crow[1];
    float cval[1][F];
    const uint32_t tag = sc.tags[i];
    static_assert(SC_EMPTY == Q_NONE, "an empty slot is 'no contribution'");
    crow[0] = tag;
#pragma unroll
    for (int f = 0; f < F; f++) cval[0][f] = sc.sums[i * F + f];
    queue_push<1, F>(Q, level, q_cnt, q_base, crow, cval, table_grad);
    sc.tags[i] = SC_EMPTY;
  }
  __syncthreads();
}

// Queue mode asks for 5 waves per SIMD: left alone the compiler takes 162 VGPRs (3 waves), told so it needs 96 without a
// vector spill, and the kernel waits on LDS / gathers for half of its wave cycles (profiles/r02_pmc_sq_encode_bwd.txt), so the
// extra residents pay: bench batch 0.835 -> 0.79 ms (4 waves: no change; 6 waves spill 13 VGPRs).  The 24-KiB cache of queue
// mode lets 5 workgroups share a CU's LDS.  (P = 4 and F = 4 would spill a few VGPRs at 5 waves: they ask for 4.)
// DBL = true: the DOUBLE backward's lattice scatter through the same machinery (see encode_dbl_bwd_kernel for the maths): the
// coefficient of vertex r is q_r (the directional derivative of its barycentric along u = dd_positions) instead of bary_r, and
// the kernel also writes grad_grad_sliced = sum_r q_r w lattice[row_r] (a gather of the rows it has just computed; skipped
// when the pointer is NULL).  grad_sliced2 (optional): the PLAIN backward's scatter of a second upstream gradient rides along,
// row_r += w (q_r g + bary_r g2) -- a training step scatters both onto the same rows of the same simplices (the double backward
// of the normals and the backward of the features), so one simplex, one run combine, one queue pass and one reduce serve both.
// Workgroups per level (round 5).  The launch is ONE resident round of workgroups, and with an equal share per level the
// coarse levels (everything absorbed by the LDS cache) were done in half the time of the fine ones (everything queued): level
// durations 0.29 .. 0.55 ms inside a 0.57-ms kernel (profiles/r03_enc_profile_levels.txt), closed levels of a coarse-to-fine
// window holding their share for nothing.  With a plan (n > 0) the grid is one-dimensional, workgroup i serves the level whose
// range [first[l], first[l + 1]) holds it, and it reports its duration (100-MHz ticks, tagged with the launch's generation)
// into host-mapped memory; the host re-deals the round in proportion to count x duration at the next call (encode_balance).
struct LevelPlan {
  int n;                    // 0: rectangular grid (blockIdx.y = level)
  uint32_t gen;             // low 8 bits tag the entries of `times`
  uint32_t* times;          // [workgroups] or NULL
  uint16_t first[42];
};
template <int P, int F, bool LATTICE, bool POS, bool QUEUE, bool DBL = false>
#if !defined(PSDF_ENC_QWAVES)
#define PSDF_ENC_QWAVES 5
#endif
__global__ void __launch_bounds__(PSDF_BLOCK, QUEUE ? ((P <= 3 && F == 2) ? PSDF_ENC_QWAVES : 4) : 1)
    encode_bwd_kernel(int64_t N, int L, uint32_t capacity, EncConv conv, const float* __restrict__ positions,
                      const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                      const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                      const float* __restrict__ grad_sliced, float* __restrict__ grad_lattice,
                      float* __restrict__ grad_positions, Queues Q, const float* __restrict__ dd_positions = nullptr,
                      float* __restrict__ grad_grad_sliced = nullptr, int pad_points = 0,
                      const float* __restrict__ grad_sliced2 = nullptr, LevelPlan plan = LevelPlan{}) {
  static_assert(!(DBL && POS), "the double backward has no position output");
  extern __shared__ __align__(16) float lds[];
#if defined(PSDF_ENC_PROFILE)
  const long long psdf_prof_t0 = (long long)wall_clock64();
#endif
  int level = blockIdx.y, bx = blockIdx.x, gx = gridDim.x;
  long long plan_t0 = 0;
  if (plan.n) {
    level = 0;
    for (int l = 1; l < plan.n; l++)
      if ((int)blockIdx.x >= (int)plan.first[l]) level = l;
    bx = (int)blockIdx.x - (int)plan.first[level];
    gx = (int)plan.first[level + 1] - (int)plan.first[level];
    plan_t0 = (long long)wall_clock64();
  }
  // duration of this workgroup, for the host's next deal (at least 1 tick: 0 means "not written")
  auto plan_report = [&]() {
    if (plan.n && plan.times && threadIdx.x == 0) {
      long long dt = (long long)wall_clock64() - plan_t0;
      dt = dt < 1 ? 1 : (dt > 0xFFFFFF ? 0xFFFFFF : dt);
      __hip_atomic_store(plan.times + blockIdx.x, ((plan.gen & 255u) << 24) | (uint32_t)dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  const int64_t ntiles = (N + PSDF_BLOCK - 1) / PSDF_BLOCK;
  if (level >= L) {
    if (DBL) {   // concatenated-point channels: grad_g = u_d * points_scaling
      const int e = level - L;
      for (int64_t tile = bx; tile < ntiles; tile += gx) {
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
          if ((d < P || pad_points) && grad_grad_sliced) grad_grad_sliced[((int64_t)level * F + f) * N + n] = v;
        }
      }
    }
    if (POS) {
      const int e = level - L;
      for (int64_t tile = bx; tile < ntiles; tile += gx) {
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
    plan_report();
    return;
  }
  if (window[level] == 0.f) {   // closed level (workgroup-uniform, before any barrier): every contribution carries the factor 0
    if (DBL && grad_grad_sliced) {
      for (int64_t tile = bx; tile < ntiles; tile += gx) {
        const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
        if (n >= N) continue;
#pragma unroll
        for (int f = 0; f < F; f++) grad_grad_sliced[((int64_t)level * F + f) * N + n] = 0.f;
      }
    }
    plan_report();
    return;
  }
  using SCache = ScatterCache<F, QUEUE ? 4096 : 8192>;
  SCache sc;
  int* q_cnt = reinterpret_cast<int*>(lds + SCache::bytes() / 4);  // [Q_MAX_PARTS]
  int* q_base = q_cnt + Q_MAX_PARTS;                                          // [Q_MAX_PARTS]
  int* q_off = q_base + Q_MAX_PARTS;                                          // [Q_MAX_PARTS + 1]
  int* q_gd = q_off + Q_MAX_PARTS + 1;                                        // [Q_MAX_PARTS]
  int* q_vote = q_gd + Q_MAX_PARTS;                                           // [3]
  if (QUEUE && threadIdx.x < 3) q_vote[threadIdx.x] = 0;   // (a barrier follows in sc.init)
  if (LATTICE) sc.init(lds);
  bool use_cache = LATTICE;
  int hot = 0;  // lds_add_pair: >0 while this thread's adds are contended (go straight to the float atomic)
  int hits = 0, tries = 0, iter = 0;
  const float w = window[level];
  const int64_t tbase = (int64_t)level * capacity * F;
  float sfl[P], shl[P];
#pragma unroll
  for (int i = 0; i < P; i++) {
    sfl[i] = scale_factor[level * P + i];
    shl[i] = shifts[level * P + i];
  }
  // In queue mode a thread handles SPT points per tile, so that one round of slot reservation (3 barriers and a
  // returning global atomic per partition) is amortised over SPT*256 points.
  constexpr int SPT = QUEUE ? QUEUE_SPT : 1;
  constexpr int NC = SPT * (P + 1);
  const int64_t ntiles_w = (N + (int64_t)PSDF_BLOCK * SPT - 1) / ((int64_t)PSDF_BLOCK * SPT);
  for (int64_t tile = bx; tile < ntiles_w; tile += gx, iter++) {
    uint32_t crow[NC];
    float cval[NC][F];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      crow[c] = Q_NONE;   // until the thread's point fills it in; again after the combine where another lane / the cache took it
#pragma unroll
      for (int f = 0; f < F; f++) cval[c][f] = 0.f;
    }
#pragma unroll
    for (int sp = 0; sp < SPT; sp++) {
      const int64_t n = (tile * SPT + sp) * PSDF_BLOCK + threadIdx.x;
      if (n < N) {
        float g[F], g2[F];
#pragma unroll
        for (int f = 0; f < F; f++) {
          g[f] = grad_sliced[((int64_t)level * F + f) * N + n];
          g2[f] = (DBL && grad_sliced2) ? grad_sliced2[((int64_t)level * F + f) * N + n] : 0.f;
        }
        float pos[P];
        load_pos<P>(positions, n, pos);
        Simplex<P> s;
        compute_simplex<P>(pos, shl, sfl, s, conv.tie_later);
        float dbary[P + 2];
#pragma unroll
        for (int k = 0; k <= P + 1; k++) dbary[k] = 0.f;
        float q[P + 2], gg[F];
        if (DBL) {   // q_r = sum_i u_i d bary_r / d pos_i: adjoint of pos -> elevated -> barycentric slots
          float u[P];
          load_pos<P>(dd_positions, n, u);
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
#pragma unroll
          for (int f = 0; f < F; f++) gg[f] = 0.f;
        }
        uint32_t rows[P + 1];
        vertex_rows<P>(s, capacity, rows, conv.hash_c);
#pragma unroll
        for (int r = 0; r <= P; r++) {
          const uint32_t row = rows[r];
          if (DBL && grad_grad_sliced) {
            const float qw = q[r] * w;
#pragma unroll
            for (int f = 0; f < F; f++) gg[f] = gg[f] + qw * lattice[tbase + (int64_t)row * F + f];
          }
          if (LATTICE) {
            const int c = sp * (P + 1) + r;
            const float bw = (DBL ? q[r] : s.bary[r]) * w;
            crow[c] = row;
#pragma unroll
            for (int f = 0; f < F; f++) cval[c][f] = g[f] * bw;
            if (DBL && grad_sliced2) {
              const float bw2 = s.bary[r] * w;
#pragma unroll
              for (int f = 0; f < F; f++) cval[c][f] = cval[c][f] + g2[f] * bw2;
            }
          }
          if (POS) {
#pragma unroll
            for (int f = 0; f < F; f++) dbary[r] = dbary[r] + lattice[tbase + (int64_t)row * F + f] * w * g[f];
          }
        }
        if (DBL && grad_grad_sliced) {
#pragma unroll
          for (int f = 0; f < F; f++) grad_grad_sliced[((int64_t)level * F + f) * N + n] = gg[f];
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
    }
    if (LATTICE) {
      // all lanes (also those past the end of the batch) take part in the DPP run combine
#pragma unroll
      for (int c = 0; c < NC; c++) {
        // (also when the cache has been voted off: skipping the combine there was measured, round 3 -- encode backward pair
        // 0.78 -> 0.92 ms, the runs of the mid levels are what keeps their queue traffic down)
        // (also where neighbouring samples do not share rows -- the finest levels: a per-workgroup vote that skips the combine
        // when the first tile merged < 1/16 of its contributions was measured in round 4 and made the pair SLOWER, 0.798 ->
        // 0.835 ms on the bench step, profiles/r04_bench_combine_vote_ab.txt: the scan is cheaper than the vote's bookkeeping)
        const bool own = combine_runs16<F>(crow[c], cval[c]);
        bool absorbed = false;
        if (own && use_cache) {
          const int rc = sc.add(crow[c], cval[c], hot);
          absorbed = rc != 0;
          hits += (rc == 2);
          tries++;
        }
        if (!own || absorbed) crow[c] = Q_NONE;
      }
    }
    if (LATTICE) {
      if (QUEUE) {
        // everything the cache did not absorb goes to the queues (also while the cache is on: rays are spatially
        // coherent, so a mid level can re-use entries AND overflow the 4096 slots)
        uint32_t all = Q_NONE;   // all ones: the AND of the rows is Q_NONE iff every one of them is
#pragma unroll
        for (int c = 0; c < NC; c++) all &= crow[c];
        const bool any = all != Q_NONE;
        // Does any thread of the workgroup hold something for the queues?  With the cache off: yes, no vote.  Otherwise one
        // barrier (__syncthreads_or takes three): a wave that has something sets word iter % 3, everybody reads it behind the
        // barrier; the word of the NEXT super-tile is cleared in front of that barrier -- its last readers (super-tile iter - 2)
        // have arrived at the barrier of iter - 1 by then.
        bool go = !use_cache && iter > 0;
        if (!go) {
          const int vw = iter % 3;
          if (threadIdx.x == 0) q_vote[vw == 2 ? 0 : vw + 1] = 0;
          if (__builtin_amdgcn_ballot_w64(any) != 0ull && psdf::lane_id() == 0) q_vote[vw] = 1;
          __syncthreads();
          go = q_vote[vw] != 0;
        }
        if (go) {
          // cache off (and drained): its LDS is the staging area of the coalesced hand-off
          if constexpr (F == 2 && NC * PSDF_BLOCK * 3 * 4 <= SCache::SC_SLOTS * (1 + F) * 4) {
            if (!use_cache && iter > 0)
              queue_push_staged<NC>(Q, level, q_cnt, q_base, q_off, q_gd, lds, crow, cval, grad_lattice + tbase);
            else
              queue_push<NC, F>(Q, level, q_cnt, q_base, crow, cval, grad_lattice + tbase);
          } else {
            queue_push<NC, F>(Q, level, q_cnt, q_base, crow, cval, grad_lattice + tbase);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < NC; c++)
          if (crow[c] != Q_NONE) {
#pragma unroll
            for (int f = 0; f < F; f++) atomicAdd(grad_lattice + tbase + (int64_t)crow[c] * F + f, cval[c][f]);
          }
      }
    }
    if (LATTICE && iter == 0) {
      use_cache = cache_vote(sc, hits, tries);  // re-use rate of the first tile
      if (QUEUE && !use_cache) cache_drain_to_queue<F>(sc, Q, level, q_cnt, q_base, grad_lattice + tbase);
    }
  }
  // (a cache that was switched off in queue mode has been drained and its LDS re-used: nothing to flush)
  if (LATTICE && !(QUEUE && !use_cache)) sc.flush(grad_lattice + tbase);
  plan_report();
#if defined(PSDF_ENC_PROFILE)
  if (QUEUE && threadIdx.x == 0) {   // per-level duration of the workgroups (wall clock ticks), into the tail of the queue counters
    const long long dt = (long long)wall_clock64() - psdf_prof_t0;
    int* prof = Q.tails + L * Q.np;
    atomicMax(&prof[level], (int)dt);
    atomicAdd(&prof[64 + level], (int)(dt >> 6));
    atomicAdd(&prof[128 + level], use_cache ? 1 : 0);
  }
#endif
}

// Position gradient ONLY (no lattice gradient: inference normals, sphere_trace.py; the reference gets them from
// autograd, models.py:236-251).  Same level-major launch shape as the forward, but a thread handles POS_LPB consecutive
// levels and sums their contributions in registers before it touches grad_positions: a quarter of the atomics of the
// general kernel (which also carries the scatter-cache / queue machinery of the lattice gradient) while the 4 tables
// of a group (8 MiB) still mostly live in the L2s.
// PARTIAL (round 4): the level groups do not meet in grad_positions with float atomics; each group stores its sum to its own
// [N, P] slab of a scratch buffer (plain coalesced stores) and encode_bwd_pos_reduce_kernel adds the slabs in group order.
// What it buys is DETERMINISM, not time: the position gradient (the normals of every training step, whose last bits the
// curvature term amplifies) no longer depends on the order in which atomics land.  Measured (profiles/r04_enc_ab.jsonl, 2 M
// points, 16 levels): 0.634 ms against 0.644 ms with atomics -- the 30 M atomics were NOT what makes this kernel 1.8x the
// forward (0.358 ms); 49 152 points x 24 levels: 30.8 against 28.8 us (one more launch).
// LPB = levels per thread: 2 for large batches, 8 for small ones (round 4, profiles/r04_enc_ab_pos_lpb.jsonl).  At 2 M points the
// kernel waits for gathers (70 % of its wave cycles) and the tables of 4 levels (8 MiB) do not fit an XCD's 4-MiB L2 (hit rate
// 48 %, profiles/r04_pmc_encode_bwd_pos.txt): 16 levels 0.623 ms with 4 levels per thread, **0.503 ms with 2** (two tables = one
// L2), 0.524 with 1 (nine more slabs to write and read), 0.749 with 8.  At a training step's 49 152 points x 24 levels nothing
// thrashes and fewer, longer threads win: 28.8 us with 8, 31.0 with 4, 32.9 with 2.
template <int P, int F, bool PARTIAL, int POS_LPB>
__global__ void __launch_bounds__(PSDF_BLOCK)
    encode_bwd_pos_kernel(int64_t N, int L, int Lt, uint32_t capacity, EncConv conv, const float* __restrict__ positions,
                          const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                          const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                          int pad_points, const float* __restrict__ grad_sliced, const unsigned char* __restrict__ skip,
                          float* __restrict__ grad_positions) {
  const int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (n >= N) return;
  if (skip && skip[n]) return;     // masked sample: its gradient stays as it is
  float pos[P];
  load_pos<P>(positions, n, pos);
  float gp[P];
#pragma unroll
  for (int i = 0; i < P; i++) gp[i] = 0.f;
#pragma unroll
  for (int li = 0; li < POS_LPB; li++) {
    const int level = blockIdx.y * POS_LPB + li;
    if (level >= Lt) break;
    float g[F];
#pragma unroll
    for (int f = 0; f < F; f++)   // rows beyond L*F + P exist only in the padded layout
      g[f] = (level < L || (level - L) * F + f < P || pad_points) ? grad_sliced[((int64_t)level * F + f) * N + n] : 0.f;
    if (level >= L) {  // pseudo-levels: the concatenated, scaled point
      const int e = level - L;
#pragma unroll
      for (int f = 0; f < F; f++) {
        const int d = e * F + f;
#pragma unroll
        for (int i = 0; i < P; i++)
          if (i == d) gp[i] = gp[i] + g[f] * points_scaling;
      }
      continue;
    }
    float sfl[P], shl[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
      sfl[i] = scale_factor[level * P + i];
      shl[i] = shifts[level * P + i];
    }
    const float w = window[level];
    if (w == 0.f) continue;   // closed level: contributes 0
    const int64_t tbase = (int64_t)level * capacity * F;
    Simplex<P> s;
    compute_simplex<P>(pos, shl, sfl, s, conv.tie_later);
    float dbary[P + 2];
#pragma unroll
    for (int k = 0; k <= P + 1; k++) dbary[k] = 0.f;
    uint32_t rows[P + 1];
    vertex_rows<P>(s, capacity, rows, conv.hash_c);
#pragma unroll
    for (int r = 0; r <= P; r++) {
      const uint32_t row = rows[r];
#pragma unroll
      for (int f = 0; f < F; f++) dbary[r] = dbary[r] + lattice[tbase + (int64_t)row * F + f] * w * g[f];
    }
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
      gp[i] = gp[i] + acc * sfl[i];
    }
  }
  if (PARTIAL) {   // grad_positions = the scratch slabs [groups][N][P]
#pragma unroll
    for (int i = 0; i < P; i++) grad_positions[((int64_t)blockIdx.y * N + n) * P + i] = gp[i];
  } else {
#pragma unroll
    for (int i = 0; i < P; i++) atomicAdd(grad_positions + n * P + i, gp[i]);
  }
}

// grad_positions[n][i] += sum over the level groups of partial[g][n][i] (in group order: deterministic); one thread per float
__global__ void __launch_bounds__(PSDF_BLOCK)
    encode_bwd_pos_reduce_kernel(int64_t NP, int P, int groups, const float* __restrict__ partial,
                                 const unsigned char* __restrict__ skip, float* __restrict__ grad_positions) {
  const int64_t e = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (e >= NP) return;
  if (skip && skip[e / P]) return;     // masked sample: its slabs were not written, its gradient stays as it is
  float s = partial[e];
  for (int g = 1; g < groups; g++) s = s + partial[(int64_t)g * NP + e];
  grad_positions[e] = grad_positions[e] + s;
}

// One workgroup per (partition, level): fold the queue into an LDS image of the partition's table slice (64-bit
// compare-and-swap per feature pair, lds_add_pair), then add the slice to the gradient table (plain read-modify-write:
// this workgroup is the only writer of these rows after the binning kernel has finished).
template <int F>
__global__ void __launch_bounds__(1024)
    encode_bwd_reduce_kernel(uint32_t capacity, Queues Q, float* __restrict__ grad_lattice) {
  extern __shared__ __align__(16) float tab[];  // [rows per partition][F]
  const int part = blockIdx.x, level = blockIdx.y;
  int n = Q.tails[level * Q.np + part];
  if (n == 0) return;
  if (n > Q.cap) n = Q.cap;
  const int rpp = 1 << Q.shift;
  for (int i = threadIdx.x; i < rpp * F; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  const int64_t qb = ((int64_t)level * Q.np + part) * Q.cap;
  const uint16_t* __restrict__ rows = Q.rows + qb;
  const float* __restrict__ vals = Q.vals + qb * F;
#if !defined(PSDF_ENC_REDUCE_U)
#define PSDF_ENC_REDUCE_U 8
#endif
  // queue entries in flight per thread (the loop is HBM-latency bound otherwise); measured on the bench step, encode backward
  // pair: U = 4 0.847 ms, 8 0.742, 12 0.742, 16 0.750
  constexpr int U = PSDF_ENC_REDUCE_U;
  int hot = 0;
  for (int base = 0; base < n; base += U * (int)blockDim.x) {
    int row[U];
    float v[U][F];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
      row[u] = (i < n) ? (int)rows[i] : -1;
      if (F == 2) {
        const float2 t = (i < n) ? *reinterpret_cast<const float2*>(vals + (int64_t)i * 2) : make_float2(0.f, 0.f);
        v[u][0] = t.x;
        v[u][F - 1] = t.y;
      } else {
#pragma unroll
        for (int f = 0; f < F; f++) v[u][f] = (i < n) ? vals[(int64_t)i * F + f] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (row[u] >= 0) {
#pragma unroll
        for (int f = 0; f < F; f += 2) lds_add_pair(&tab[row[u] * F + f], v[u][f], v[u][f + 1], hot);
      }
  }
  __syncthreads();
  const int64_t row0 = (int64_t)part << Q.shift;
  int64_t nrows = (int64_t)capacity - row0;
  if (nrows > rpp) nrows = rpp;
  float* __restrict__ out = grad_lattice + ((int64_t)level * capacity + row0) * F;
  for (int64_t i = threadIdx.x; i < nrows * F; i += blockDim.x) {
    const float v = tab[i];
    if (v != 0.f) out[i] = out[i] + v;
  }
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
    encode_dbl_bwd_kernel(int64_t N, int L, uint32_t capacity, EncConv conv, const float* __restrict__ positions,
                          const float* __restrict__ lattice, const float* __restrict__ scale_factor,
                          const float* __restrict__ shifts, const float* __restrict__ window, float points_scaling,
                          int pad_points, const float* __restrict__ dd_positions, const float* __restrict__ grad_sliced,
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
        if (d < P || pad_points) grad_grad_sliced[((int64_t)level * F + f) * N + n] = v;
      }
    }
    return;
  }
  const float w = window[level];
  if (w == 0.f) {   // closed level: no lattice contribution, gradient w.r.t. the feature gradient 0
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t n = tile * PSDF_BLOCK + threadIdx.x;
      if (n >= N) continue;
#pragma unroll
      for (int f = 0; f < F; f++) grad_grad_sliced[((int64_t)level * F + f) * N + n] = 0.f;
    }
    return;
  }
  ScatterCache<F> sc;
  if (LATTICE) sc.init(lds);
  bool use_cache = LATTICE;
  int hot = 0;  // lds_add_pair: >0 while this thread's adds are contended (go straight to the float atomic)
  int hits = 0, tries = 0, iter = 0;
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
      compute_simplex<P>(pos, shl, sfl, s, conv.tie_later);
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
      uint32_t rows[P + 1];
      vertex_rows<P>(s, capacity, rows, conv.hash_c);
#pragma unroll
      for (int r = 0; r <= P; r++) {
        const uint32_t row = rows[r];
        const float qw = q[r] * w;
        if (LATTICE) {
          float v[F];
#pragma unroll
          for (int f = 0; f < F; f++) v[f] = qw * g[f];
          bool absorbed = false;
          if (use_cache) {
            const int rc = sc.add(row, v, hot);
            absorbed = rc != 0;
            hits += (rc == 2);
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
    if (LATTICE && iter == 0) use_cache = cache_vote(sc, hits, tries);  // re-use rate of the first tile
  }
  if (LATTICE) sc.flush(grad_lattice + tbase);
}

// `concat` = PSDF_ENC_CONCAT_* (encode_conventions.h): both concatenation layouts sweep ceil(P/F) pseudo-levels; the padded
// one also owns the channels L*F + d with d >= P (zeros), the appended one does not have them.
inline int extra_levels(int P, int F, int concat) { return concat ? (P + F - 1) / F : 0; }
inline int pad_points(int concat) { return concat == PSDF_ENC_CONCAT_PSEUDO_LEVELS; }
inline bool concat_ok(int concat) { return concat >= PSDF_ENC_CONCAT_NONE && concat <= PSDF_ENC_CONCAT_APPEND; }

}  // namespace

namespace psdf {
EncConv& enc_conv_state() {
  static EncConv c{(uint32_t)PSDF_ENC_HASH_MULTIPLIER, (uint32_t)PSDF_ENC_RANK_TIE_RAISES_LATER};
  return c;
}
}  // namespace psdf

// ================================================================================== C ABI
// Position gradient only: the level-group kernel + the slab reduction when stream-ordered scratch is available (not while a
// graph is being captured: there the float-atomic form runs), PSDF_ENC_POS_ATOMICS=1 forces the atomic form (A/B).
template <int P_, int F_, int POS_LPB>
static void launch_bwd_pos_lpb(int64_t N, int nr_levels, int Lt, int capacity, const float* positions, const float* lattice,
                               const float* scale_factor, const float* shifts, const float* window, float points_scaling, int pad,
                               const float* grad_sliced, const unsigned char* skip, float* grad_positions, hipStream_t st,
                               bool capturing) {
  static const bool force_atomics = getenv("PSDF_ENC_POS_ATOMICS") && atoi(getenv("PSDF_ENC_POS_ATOMICS")) != 0;
  const unsigned nb = psdf_blocks(N, PSDF_BLOCK);
  const int groups = (Lt + POS_LPB - 1) / POS_LPB;
  // The slabs are per-stream scratch that is never handed back: capped (PSDF_ENC_POS_SLAB_MAX_MB, default 256 MiB = 1.7 M points
  // at 24 + 2 levels) -- a multi-million-sample render pass takes the float-atomic form (same time, profiles/r04_enc_ab.jsonl:
  // what the slabs buy is run-to-run identical normals, which matters to the training step's curvature term at its 49 K samples)
  static const size_t slab_cap = (size_t)(getenv("PSDF_ENC_POS_SLAB_MAX_MB") ? atoll(getenv("PSDF_ENC_POS_SLAB_MAX_MB")) : 256) << 20;
  const size_t slab_bytes = (size_t)groups * N * P_ * sizeof(float);
  float* slabs = (groups > 1 && !force_atomics && !capturing && slab_bytes <= slab_cap) ? (float*)psdf::stream_scratch(slab_bytes, st) : nullptr;
  if (slabs) {
    hipLaunchKernelGGL((encode_bwd_pos_kernel<P_, F_, true, POS_LPB>), dim3(nb, groups), dim3(PSDF_BLOCK), 0, st, N, nr_levels, Lt,
                       (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window,
                       points_scaling, pad, grad_sliced, skip, slabs);
    hipLaunchKernelGGL(encode_bwd_pos_reduce_kernel, dim3(psdf_blocks(N * P_, PSDF_BLOCK)), dim3(PSDF_BLOCK), 0, st, N * P_, P_,
                       groups, slabs, skip, grad_positions);
  } else {
    hipLaunchKernelGGL((encode_bwd_pos_kernel<P_, F_, false, POS_LPB>), dim3(nb, groups), dim3(PSDF_BLOCK), 0, st, N, nr_levels, Lt,
                       (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window,
                       points_scaling, pad, grad_sliced, skip, grad_positions);
  }
}

template <int P_, int F_>
static void launch_bwd_pos(int64_t N, int nr_levels, int Lt, int capacity, const float* positions, const float* lattice,
                           const float* scale_factor, const float* shifts, const float* window, float points_scaling, int pad,
                           const float* grad_sliced, const unsigned char* skip, float* grad_positions, hipStream_t st) {
  // (the capture state is probed once per launch: while a graph is being captured -- the sphere tracer's -- there is no
  // stream-ordered scratch and the float-atomic form runs, as measured in round 2)
  const bool capturing = psdf::stream_scratch(16, st) == nullptr;
  if (capturing)
    launch_bwd_pos_lpb<P_, F_, 4>(N, nr_levels, Lt, capacity, positions, lattice, scale_factor, shifts, window, points_scaling,
                                  pad, grad_sliced, skip, grad_positions, st, true);
  else if (N >= ((int64_t)1 << 17))
    launch_bwd_pos_lpb<P_, F_, 2>(N, nr_levels, Lt, capacity, positions, lattice, scale_factor, shifts, window, points_scaling,
                                  pad, grad_sliced, skip, grad_positions, st, false);
  else
    launch_bwd_pos_lpb<P_, F_, 8>(N, nr_levels, Lt, capacity, positions, lattice, scale_factor, shifts, window, points_scaling,
                                  pad, grad_sliced, skip, grad_positions, st, false);
}

extern "C" {

// The conventions in force, by index: 0 hash multiplier, 1 rank tie rule (both RUNTIME values: the defaults of
// encode_conventions.h until psdf_encode_set_conventions() replaces them), 2 sqrt term in scale_factor, 3 inverse-std-dev
// term, 4 default concatenation layout (host-side conventions: scale_factor and the layout are ARGUMENTS of every entry point,
// the values here are the defaults the Python mirror and the oracle read).  Host only (no device needed).
int64_t psdf_encode_convention(int which) {
  switch (which) {
    case 0: return (int64_t)psdf::enc_conv_state().hash_c;
    case 1: return (int64_t)psdf::enc_conv_state().tie_later;
    case 2: return PSDF_ENC_SCALE_SQRT_TERM;
    case 3: return PSDF_ENC_SCALE_INV_STDDEV;
    case 4: return PSDF_ENC_CONCAT_DEFAULT_LAYOUT;
    default: return -1;
  }
}

// Replace the two device-side conventions for this process (every later launch uses them): hash multiplier (any odd or even
// 32-bit value; 0 is rejected) and the rank tie rule (0 / 1).  Not thread-safe against concurrent launches.
int psdf_encode_set_conventions(uint32_t hash_multiplier, int rank_tie_raises_later) {
  if (hash_multiplier == 0u || (rank_tie_raises_later != 0 && rank_tie_raises_later != 1)) return PSDF_ERR_ARG;
  psdf::enc_conv_state() = EncConv{hash_multiplier, (uint32_t)rank_tie_raises_later};
  return PSDF_OK;
}

static int device_cus();
static int encode_forward_impl(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                               const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                               int concat_points, float points_scaling, const unsigned char* skip, float* sliced,
                               void* stream, unsigned char* touched = nullptr, int touch_shift = 0) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !sliced || !concat_ok(concat_points))
    return PSDF_ERR_ARG;
  if ((int64_t)capacity * nr_feat * 4 > 0xffffffffll) return PSDF_ERR_UNSUPPORTED;   // the kernel's 32-bit offsets inside a level
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + extra_levels(pos_dim, nr_feat, concat_points);
  dim3 grid(psdf_blocks(N, PSDF_BLOCK), Lt);
  // one resident round of workgroups per level (8 per CU) that walk the tiles, instead of one workgroup per tile: the level's
  // constants are fetched once per workgroup.  Round 5, 2 M points x 16 levels, forward bracket of the bench step: 0.650 ms
  // uncapped, 0.646 with 1024, 0.636 with 2048 or 4096 (profiles/r05_mlp_pair_ab.txt).  PSDF_ENC_FWD_WGS=0: one tile each.
  static const int fwd_env = getenv("PSDF_ENC_FWD_WGS") ? atoi(getenv("PSDF_ENC_FWD_WGS")) : -1;
  const int fwd_wgs = fwd_env >= 0 ? fwd_env : device_cus() * 8;
  if (fwd_wgs > 0 && (int)grid.x > fwd_wgs) grid.x = fwd_wgs;
#define FWD_(P_, F_, PL_)                                                                                      \
  hipLaunchKernelGGL((encode_fwd_kernel<P_, F_, PL_>), grid, dim3(PSDF_BLOCK), 0, st, N, nr_levels, (uint32_t)capacity, psdf::enc_conv_state(), \
                     positions, lattice, scale_factor, shifts, window, points_scaling, pad_points(concat_points), skip, sliced,  \
                     touched, touch_shift, (capacity + (1 << touch_shift) - 1) >> touch_shift)
  // (without a mask and a touched-block map the kernel needs fewer scalar registers: 8 instead of 7 waves per SIMD)
#define FWD(P_, F_)                          \
  do {                                       \
    if (!skip && !touched) FWD_(P_, F_, true); \
    else FWD_(P_, F_, false);                \
  } while (0)
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
#undef FWD_
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_encode_forward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                        const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                        int concat_points, float points_scaling, float* sliced, void* stream) {
  return encode_forward_impl(pos_dim, nr_feat, N, nr_levels, capacity, positions, lattice, scale_factor, shifts, window,
                             concat_points, points_scaling, nullptr, sliced, stream);
}

// Same with a per-point mask: points with skip[n] != 0 are not evaluated and their columns of `sliced` are left as
// they are (fixed-shape callers that keep one slot per ray, e.g. the sphere tracer's converged rays).
int psdf_encode_forward_masked(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                               const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                               int concat_points, float points_scaling, const unsigned char* skip, float* sliced,
                               void* stream) {
  return encode_forward_impl(pos_dim, nr_feat, N, nr_levels, capacity, positions, lattice, scale_factor, shifts, window,
                             concat_points, points_scaling, skip, sliced, stream);
}

// Workspace for the binned (queue + LDS reduction) lattice-gradient path; 0 = the path does not apply (small batch,
// table too large for 64 partitions) and psdf_encode_backward_ws behaves exactly like psdf_encode_backward.
static bool queue_plan(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, Queues& Q, int64_t& bytes) {
  bytes = 0;
  // Below ~8 K points the fixed cost of the plan (counter memset, one resident round of binning workgroups, a reduce pass over
  // the whole table: ~27 us for 24 levels x 2^18 rows) exceeds what the plain path (LDS cache + float atomics, ~4.7 ns per point
  // when every level carries a gradient) spends.  Measured, tools/small_batch_enc_bench.py, lattice backward: 1 056 points
  // 10.8 us plain / 27.1 us queued; 49 152 points 232 / 71.5 us; 262 080 points 1206 / 174.5 us.  (Until round 3 the
  // threshold was 2^18 points: every backward of a training step -- ~49 K ray samples -- took the plain path.)
  static const int64_t min_n = getenv("PSDF_ENC_QUEUE_MIN_N") ? atoll(getenv("PSDF_ENC_QUEUE_MIN_N")) : ((int64_t)1 << 13);
  if (N < min_n) return false;
  // rows per partition: the reduce kernel holds a partition's slice of the table in LDS.  64-KiB slices (two reduce
  // workgroups per CU) instead of the 128 KiB that fit: the coarse levels' queues are nearly empty, so with one workgroup per
  // (level, partition) and 16 partitions only half the CUs had work (16 levels: reduce + binning 0.966 -> 0.905 ms with 32).
  const int base = (nr_feat <= 2) ? 14 : (nr_feat <= 4 ? 13 : 12);   // rows/partition * F * 4 B <= 128 KiB
  static const int delta = getenv("PSDF_ENC_QUEUE_SLICE_LOG2_DELTA") ? atoi(getenv("PSDF_ENC_QUEUE_SLICE_LOG2_DELTA")) : 1;   // A/B switch
  int shift = base - (delta < 0 ? 0 : (delta > 3 ? 3 : delta));
  while (shift < base && ((capacity + (1 << shift) - 1) >> shift) > Q_MAX_PARTS) shift++;   // large tables: keep the partition count
  const int rpp = 1 << shift;
  const int np = (capacity + rpp - 1) / rpp;
  if (np > Q_MAX_PARTS || rpp * nr_feat * 4 > 128 * 1024) return false;
  const int64_t contrib = (int64_t)(pos_dim + 1) * N;
  const int64_t cap = (contrib / np) + (contrib / np) / 4 + 4096;
  if (cap > 0x7fffffff) return false;
  // the kernels address a level's slice of the queues with 32-bit byte offsets (queue_store)
  if ((int64_t)np * cap * (nr_feat * 4 > 2 ? nr_feat * 4 : 2) > 0xffffffffll) return false;
  Q.cap = (int)cap;
  Q.np = np;
  Q.shift = shift;
  const int64_t entries = (int64_t)nr_levels * np * cap;
  const int64_t rows_b = (entries * 2 + 255) & ~(int64_t)255, vals_b = (entries * nr_feat * 4 + 255) & ~(int64_t)255;
  bytes = rows_b + vals_b + (((int64_t)nr_levels * np * 4 + 255) & ~(int64_t)255) + 1024;   // + profile slots (variant builds)
  return true;
}
static void queue_carve(void* ws, int nr_feat, int nr_levels, Queues& Q) {
  const int64_t entries = (int64_t)nr_levels * Q.np * Q.cap;
  const int64_t rows_b = (entries * 2 + 255) & ~(int64_t)255, vals_b = (entries * nr_feat * 4 + 255) & ~(int64_t)255;
  char* p = (char*)ws;
  Q.rows = (uint16_t*)p;
  Q.vals = (float*)(p + rows_b);
  Q.tails = (int*)(p + rows_b + vals_b);
}

// Forward that also records the blocks of 2^block_rows_log2 table rows it reads: touched_blocks [nr_levels, ceil(capacity /
// 2^block_rows_log2)] bytes, set to 1 (never cleared here).  See psdf_adamw_step_blocks.
int psdf_encode_forward_mark(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                             const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                             int concat_points, float points_scaling, float* sliced, unsigned char* touched_blocks,
                             int block_rows_log2, void* stream) {
  if (!touched_blocks || block_rows_log2 < 0 || block_rows_log2 > 20) return PSDF_ERR_ARG;
  return encode_forward_impl(pos_dim, nr_feat, N, nr_levels, capacity, positions, lattice, scale_factor, shifts, window,
                             concat_points, points_scaling, nullptr, sliced, stream, touched_blocks, block_rows_log2);
}

int64_t psdf_encode_backward_workspace_bytes(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity) {
  Queues Q{};
  int64_t bytes = 0;
  queue_plan(pos_dim, nr_feat, N, nr_levels, capacity, Q, bytes);
  return bytes;
}

// grad_lattice / grad_positions must be zero-initialised by the caller (or hold a running sum to add to);
// either may be NULL to skip that gradient.  workspace: device scratch of at least
// psdf_encode_backward_workspace_bytes() bytes (contents undefined on entry and exit) or NULL.
static int device_cus() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

}  // extern "C"

// The deal of one resident round of binning workgroups over the levels (see LevelPlan).  State of the previous call per process:
// the plan it launched with and the host-mapped array its workgroups report their durations into (two arrays, used alternately,
// entries tagged with the call's generation: a call never reads what an older or an unfinished launch wrote as if it were the
// previous one's).  share_l <- 1/2 share_l + 1/2 (count_l x mean duration_l) / sum: a level whose workgroups ran longer gets
// more of them next time; a closed level (its workgroups return at once) falls to the minimum.  PSDF_ENC_BWD_BALANCE=0: equal shares.
namespace {
struct EncBalance {
  int levels = 0, total = 0, kind = 0, bucket = -1;
  const void* ident = nullptr;   // the lattice the launches belong to (encodings of one shape have different cost profiles)
  uint32_t gen = 0;
  uint64_t last_use = 0;
  std::vector<int> counts;
  uint32_t* times[2] = {nullptr, nullptr};
  int device = -1;
};
constexpr int BAL_MAX_WG = 8192, BAL_MIN_PER_LEVEL = 4, BAL_STATES = 16;
std::mutex g_shares_mu;
std::vector<int> g_last_shares;

// returns false when the plan cannot be used (equal shares through the rectangular grid then).  kind: 0 backward, 1 double
// backward.  One state per (lattice, kind, levels, size class of N: quarter octaves, round size) -- a training step alternates between
// several shapes and its sample count moves a little from step to step --, at most BAL_STATES of them (least recently used
// one is re-used).
bool encode_balance(int nr_levels, int64_t N, int total, int64_t super_tiles, LevelPlan& plan, const void* ident, int kind = 0) {
  static const bool off = getenv("PSDF_ENC_BWD_BALANCE") && atoi(getenv("PSDF_ENC_BWD_BALANCE")) == 0;
  static std::mutex mu;
  static EncBalance states[BAL_STATES];
  static uint64_t tick = 0;
  // (batches below 2^18 points: a training step's 49 152 samples have fewer super-tiles per level than a level's equal share of
  // the round, and the measured effect is inside the run-to-run noise, 449 / 447 it/s with against 471 / 438 without: off)
  if (off || nr_levels > 40 || total > BAL_MAX_WG || total < nr_levels * BAL_MIN_PER_LEVEL || N < ((int64_t)1 << 18)) return false;
  std::lock_guard<std::mutex> lock(mu);
  int bucket = 0;
  for (int64_t v = N; v > 1; v >>= 1) bucket += 4;
  {
    const int64_t top = (int64_t)1 << (bucket / 4);
    bucket += (int)(((N - top) * 4) / top);
  }
  // (the key holds the device as well: an address re-used on another GPU -- or after a free -- must not inherit a deal measured
  //  elsewhere)
  int device = -1;
  (void)hipGetDevice(&device);
  EncBalance* Bp = nullptr;
  for (auto& st_ : states)
    if (st_.times[0] && st_.ident == ident && st_.device == device && st_.kind == kind && st_.levels == nr_levels && st_.bucket == bucket &&
        st_.total == total)
      Bp = &st_;
  if (!Bp) {
    for (auto& st_ : states)
      if (!Bp || st_.last_use < Bp->last_use) Bp = &st_;
    if (!Bp->times[0]) {
      for (int i = 0; i < 2; i++) {
        void* h = nullptr;
        if (hipHostMalloc(&h, BAL_MAX_WG * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess || !h) {
          (void)hipGetLastError();
          if (i == 1) (void)hipHostFree(Bp->times[0]);
          Bp->times[0] = Bp->times[1] = nullptr;
          return false;
        }
        Bp->times[i] = (uint32_t*)h;
      }
    }
    memset(Bp->times[0], 0, BAL_MAX_WG * sizeof(uint32_t));     // (a launch of the state's previous owner may still write: its
    memset(Bp->times[1], 0, BAL_MAX_WG * sizeof(uint32_t));     // entries carry an older generation tag and are ignored)
    Bp->levels = 0;
  }
  EncBalance& B = *Bp;
  B.last_use = ++tick;
  if (!B.times[0] || !B.times[1]) return false;
  const int cap = (int)(super_tiles < total ? super_tiles : total);       // more workgroups than super-tiles of a level: idle ones
  if (B.levels != nr_levels || (int)B.counts.size() != nr_levels) {
    B.levels = nr_levels;
    B.kind = kind;
    B.ident = ident;
    B.device = device;
    B.bucket = bucket;
    B.total = total;
    B.counts.assign(nr_levels, total / nr_levels);
  } else {
    // what the previous call's workgroups reported (all of them, with its tag -- otherwise the launch is still running, or
    // another shape ran in between: keep the deal)
    const volatile uint32_t* t = B.times[B.gen & 1];
    std::vector<double> work(nr_levels, 0.0);
    bool complete = true;
    int id = 0;
    double sum = 0.0;
    for (int l = 0; l < nr_levels && complete; l++) {
      double acc = 0.0;
      for (int b = 0; b < B.counts[l]; b++, id++) {
        const uint32_t v = t[id];
        if ((v >> 24) != (B.gen & 255u) || (v & 0xFFFFFFu) == 0u) {
          complete = false;
          break;
        }
        acc += (double)(v & 0xFFFFFFu);
      }
      work[l] = acc;       // = count x mean duration
      sum += acc;
    }
    if (complete && sum > 0.0) {
      std::vector<double> want(nr_levels);
      for (int l = 0; l < nr_levels; l++) want[l] = 0.5 * B.counts[l] + 0.5 * total * work[l] / sum;
      int used = 0;
      for (int l = 0; l < nr_levels; l++) {
        int c = (int)(want[l] + 0.5);
        c = c < BAL_MIN_PER_LEVEL ? BAL_MIN_PER_LEVEL : (c > cap ? cap : c);
        B.counts[l] = c;
        used += c;
      }
      // never more than one resident round: take the excess from the largest shares
      while (used > total) {
        int big = 0;
        for (int l = 1; l < nr_levels; l++)
          if (B.counts[l] > B.counts[big]) big = l;
        if (B.counts[big] <= BAL_MIN_PER_LEVEL) break;
        B.counts[big]--;
        used--;
      }
    }
  }
  B.gen++;
  plan.n = nr_levels;
  plan.gen = B.gen;
  plan.times = B.times[B.gen & 1];
  int at = 0;
  for (int l = 0; l < nr_levels; l++) {
    if (B.counts[l] > cap) B.counts[l] = cap;
    if (B.counts[l] < 1) B.counts[l] = 1;
    plan.first[l] = (uint16_t)at;
    at += B.counts[l];
  }
  plan.first[nr_levels] = (uint16_t)at;
  {
    std::lock_guard<std::mutex> l2(g_shares_mu);
    g_last_shares = B.counts;
  }
  return at <= BAL_MAX_WG;
}
}  // namespace

extern "C" {

// the deal of the last balanced launch (debug query): counts[0 .. nr_levels), returns nr_levels (0: none yet)
int psdf_encode_backward_level_shares(int* counts, int max_levels) {
  std::lock_guard<std::mutex> lock(g_shares_mu);
  const int n = (int)g_last_shares.size();
  for (int l = 0; l < n && l < max_levels; l++) counts[l] = g_last_shares[l];
  return n;
}

int psdf_encode_backward_ws(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                            const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                            int concat_points, float points_scaling, const float* grad_sliced, float* grad_lattice,
                            float* grad_positions, void* workspace, int64_t workspace_bytes, void* stream) {
  if (N == 0 || (!grad_lattice && !grad_positions)) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !grad_sliced || !concat_ok(concat_points))
    return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + ((grad_positions != nullptr) ? extra_levels(pos_dim, nr_feat, concat_points) : 0);
  const unsigned nb = psdf_blocks(N, PSDF_BLOCK);
  // Plain (cache + atomics) mode: a workgroup sets up and flushes a 48-KiB LDS cache, so it should walk several tiles;
  // PSDF_ENC_BWD_WG_PER_LEVEL overrides the per-level workgroup cap (measurement switch).
  static const int wg_cap_env = getenv("PSDF_ENC_BWD_WG_PER_LEVEL") ? atoi(getenv("PSDF_ENC_BWD_WG_PER_LEVEL")) : 0;
  const unsigned wg_cap = wg_cap_env > 0 ? (unsigned)wg_cap_env : 512u;
  dim3 grid(nb < wg_cap ? nb : wg_cap, Lt);
  static const bool pos_fused = getenv("PSDF_ENC_POS_FUSED") && atoi(getenv("PSDF_ENC_POS_FUSED")) != 0;   // A/B: the round-3 form
  Queues Q{};
  int64_t need = 0;
  const bool use_queue = grad_lattice && workspace && queue_plan(pos_dim, nr_feat, N, nr_levels, capacity, Q, need) &&
                         workspace_bytes >= need;
  if (use_queue) {
    grid.x = 128;  // provisional: BWD_PF sizes it to ONE resident round of workgroups (see there)
    queue_carve(workspace, nr_feat, nr_levels, Q);
#if defined(PSDF_ENC_PROFILE)
    hipError_t e = hipMemsetAsync(Q.tails, 0, (size_t)nr_levels * Q.np * sizeof(int) + 1024, st);
#else
    hipError_t e = hipMemsetAsync(Q.tails, 0, (size_t)nr_levels * Q.np * sizeof(int), st);
#endif
    if (e != hipSuccess) return (int)e;
  }
#define BWD(P_, F_, A_, B_, Q_)                                                                                  \
  hipLaunchKernelGGL((encode_bwd_kernel<P_, F_, A_, B_, Q_>), grid, dim3(PSDF_BLOCK),                             \
                     ((A_) ? ScatterCache<F_, ((Q_) ? 4096 : 8192)>::bytes() + Q_LDS_INTS * sizeof(int) : 0), st, N, \
                     nr_levels,                                                                                    \
                     (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window, points_scaling,        \
                     grad_sliced, grad_lattice, grad_positions, Q)
// Queue mode: every workgroup of the launch resident at once and none left over.  The binning kernel is a long walk per
// workgroup (LDS cache warm-up, then tens of super-tiles), so a second, partly filled round of workgroups is a tail the
// size of a whole walk: measured on the bench batch (2 M points; 3 workgroups fit a CU, 768 in all) 16 levels x 48 = 768
// workgroups 0.96 ms, x 40 1.09, x 56 1.26 (just over one round), x 96 (two rounds) 1.03, x 128 1.10; 24 levels x 32 best.
// (Those figures are from the 3-workgroups-per-CU state of the kernel; 5 fit now -- 80 per level at 16 levels -- by the same rule.)
#define BWD_PLAN(P_, F_)                                                                                          \
  hipLaunchKernelGGL((encode_bwd_kernel<P_, F_, true, false, true>), dim3(plan.first[nr_levels]), dim3(PSDF_BLOCK), \
                     (ScatterCache<F_, 4096>::bytes() + Q_LDS_INTS * sizeof(int)), st, N, nr_levels,    \
                     (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window, \
                     points_scaling, grad_sliced, grad_lattice, (float*)nullptr, Q, (const float*)nullptr,          \
                     (float*)nullptr, 0, (const float*)nullptr, plan)
#define BWD_PF(P_, F_)                                                                                           \
  do {                                                                                                           \
    LevelPlan plan{};                                                                                            \
    bool balanced = false;                                                                                       \
    if (use_queue) {                                                                                             \
      int per_cu = 0;                                                                                            \
      const size_t shm = ScatterCache<F_, 4096>::bytes() + Q_LDS_INTS * sizeof(int);                  \
      const hipError_t eo =                                                                                      \
          (grad_positions && pos_fused) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(                                        \
                               &per_cu, encode_bwd_kernel<P_, F_, true, true, true>, PSDF_BLOCK, shm)            \
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(                                        \
                               &per_cu, encode_bwd_kernel<P_, F_, true, false, true>, PSDF_BLOCK, shm);          \
      if (eo == hipSuccess && per_cu > 0) {                                                                      \
        int64_t gx = (int64_t)per_cu * device_cus() / nr_levels;                                                 \
        const int64_t super_tiles = (N + (int64_t)PSDF_BLOCK * QUEUE_SPT - 1) / ((int64_t)PSDF_BLOCK * QUEUE_SPT); \
        if (gx > super_tiles) gx = super_tiles;                                                                  \
        grid.x = (unsigned)(gx < 1 ? 1 : gx);                                                                    \
        /* lattice-only binning launches: the round is dealt over the levels by their measured cost */           \
        if (!(grad_positions && pos_fused))                                                                      \
          balanced = encode_balance(nr_levels, N, per_cu * device_cus(), super_tiles, plan, lattice);                     \
      } else {                                                                                                   \
        (void)hipGetLastError();                                                                                 \
      }                                                                                                          \
    }                                                                                                            \
    if (use_queue && grad_positions && !pos_fused) {                                                             \
      /* large batch, both gradients: the binning kernel for the lattice (its own levels only) and the level-group */ \
      /* kernel for the positions -- the fused form adds the position gradient with float atomics from inside the */ \
      /* binning kernel: +0.98 ms at 2 M points x 16 levels against 0.63 ms for the separate kernel (round 4) */   \
      grid.y = nr_levels;                                                                                        \
      if (balanced) BWD_PLAN(P_, F_); else BWD(P_, F_, true, false, true);                                       \
      launch_bwd_pos<P_, F_>(N, nr_levels, Lt, capacity, positions, lattice, scale_factor, shifts, window,        \
                             points_scaling, pad_points(concat_points), grad_sliced, (const unsigned char*)nullptr, \
                             grad_positions, st);                                                                 \
    } else if (use_queue && grad_positions)                                                                      \
      BWD(P_, F_, true, true, true);                                                                             \
    else if (use_queue) {                                                                                        \
      if (balanced) BWD_PLAN(P_, F_); else BWD(P_, F_, true, false, true);                                       \
    }                                                                                                            \
    else if (grad_lattice && grad_positions)                                                                     \
      BWD(P_, F_, true, true, false);                                                                            \
    else if (grad_lattice)                                                                                       \
      BWD(P_, F_, true, false, false);                                                                           \
    else                                                                                                         \
      launch_bwd_pos<P_, F_>(N, nr_levels, Lt, capacity, positions, lattice, scale_factor, shifts, window,        \
                             points_scaling, pad_points(concat_points), grad_sliced, (const unsigned char*)nullptr, \
                             grad_positions, st);                                                                 \
    if (use_queue) {                                                                                             \
      const size_t lds_b = (size_t)(1 << Q.shift) * F_ * sizeof(float);                                          \
      hipError_t e2 = hipFuncSetAttribute((const void*)encode_bwd_reduce_kernel<F_>,                              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);                \
      if (e2 != hipSuccess) return (int)e2;                                                                      \
      hipLaunchKernelGGL((encode_bwd_reduce_kernel<F_>), dim3(Q.np, nr_levels), dim3(1024), lds_b, st,            \
                         (uint32_t)capacity, Q, grad_lattice);                                                   \
    }                                                                                                            \
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
#undef BWD_PLAN
#undef BWD
  psdf::g_last_path[psdf::PATH_ENCODE_BWD] = use_queue ? 2 : (grad_lattice ? 1 : 3);
#if defined(PSDF_ENC_PROFILE)
  if (use_queue) {
    static int calls = 0;
    if (++calls == 8) {
      (void)hipStreamSynchronize(st);
      int prof[192];
      (void)hipMemcpy(prof, Q.tails + nr_levels * Q.np, sizeof(prof), hipMemcpyDeviceToHost);
      std::vector<int> tails((size_t)nr_levels * Q.np);
      (void)hipMemcpy(tails.data(), Q.tails, tails.size() * sizeof(int), hipMemcpyDeviceToHost);
      for (int l = 0; l < nr_levels; l++) {
        long long queued = 0;
        for (int q = 0; q < Q.np; q++) queued += tails[(size_t)l * Q.np + q];
        fprintf(stderr, "[enc-profile] level %2d: max %8d ticks, mean %8.0f ticks, cache kept by %d of %u workgroups, queued %lld of %lld "
                "contributions (%.1f %%)\n", l, prof[l], 64.0 * prof[64 + l] / grid.x, prof[128 + l], grid.x, queued,
                (long long)N * (pos_dim + 1), 100.0 * queued / ((double)N * (pos_dim + 1)));
      }
    }
  }
#endif
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_encode_backward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                         const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                         int concat_points, float points_scaling, const float* grad_sliced, float* grad_lattice,
                         float* grad_positions, void* stream) {
  return psdf_encode_backward_ws(pos_dim, nr_feat, N, nr_levels, capacity, positions, lattice, scale_factor, shifts,
                                 window, concat_points, points_scaling, grad_sliced, grad_lattice, grad_positions,
                                 nullptr, 0, stream);
}

// Position gradient only, with a per-sample mask (masked samples: no gathers, their rows of grad_positions keep their
// contents).  grad_positions is ACCUMULATED INTO, as in psdf_encode_backward.
int psdf_encode_backward_positions_masked(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity,
                                          const float* positions, const float* lattice, const float* scale_factor,
                                          const float* shifts, const float* window, int concat_points, float points_scaling,
                                          const float* grad_sliced, const unsigned char* skip, float* grad_positions,
                                          void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !grad_sliced || !grad_positions ||
      !concat_ok(concat_points))
    return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + extra_levels(pos_dim, nr_feat, concat_points);
#define POS(P_, F_)                                                                                                  \
  launch_bwd_pos<P_, F_>(N, nr_levels, Lt, capacity, positions, lattice, scale_factor, shifts, window, points_scaling, \
                         pad_points(concat_points), grad_sliced, skip, grad_positions, st)
  if (pos_dim == 3 && nr_feat == 2)
    POS(3, 2);
  else if (pos_dim == 4 && nr_feat == 2)
    POS(4, 2);
  else if (pos_dim == 2 && nr_feat == 2)
    POS(2, 2);
  else if (pos_dim == 3 && nr_feat == 4)
    POS(3, 4);
  else
    return PSDF_ERR_UNSUPPORTED;
#undef POS
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// grad_lattice must be zero-initialised (or NULL to skip); grad_grad_sliced is fully overwritten.  workspace: device scratch of
// psdf_encode_backward_workspace_bytes() bytes or NULL -- with it, batches large enough for the queue plan send the lattice
// scatter through the binning + reduce kernels of the backward (encode_bwd_kernel<.., DBL = true>) instead of float atomics:
// measured at 49 152 ray samples, 24 levels (a training step) 269 -> ~75 us, at 262 080 samples 1.36 ms -> ~0.2 ms.
int psdf_encode_double_backward_ws(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity,
                                   const float* positions, const float* lattice, const float* scale_factor,
                                   const float* shifts, const float* window, int concat_points, float points_scaling,
                                   const float* dd_positions, const float* grad_sliced, float* grad_lattice,
                                   float* grad_grad_sliced, const float* grad_sliced_direct, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || nr_levels <= 0 || capacity <= 0 || !positions || !lattice || !dd_positions || !grad_sliced ||
      (!grad_grad_sliced && !grad_lattice) || (grad_sliced_direct && !grad_lattice) || !concat_ok(concat_points))
    return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Lt = nr_levels + extra_levels(pos_dim, nr_feat, concat_points);
  const unsigned nb = psdf_blocks(N, PSDF_BLOCK);
  {
    Queues Q{};
    int64_t need = 0;
    if (grad_lattice && workspace && queue_plan(pos_dim, nr_feat, N, nr_levels, capacity, Q, need) && workspace_bytes >= need) {
      queue_carve(workspace, nr_feat, nr_levels, Q);
      hipError_t e = hipMemsetAsync(Q.tails, 0, (size_t)nr_levels * Q.np * sizeof(int), st);
      if (e != hipSuccess) return (int)e;
#define DBLQ(P_, F_)                                                                                                     \
  do {                                                                                                                   \
    auto kern = encode_bwd_kernel<P_, F_, true, false, true, true>;                                                      \
    const size_t shm = ScatterCache<F_, 4096>::bytes() + Q_LDS_INTS * sizeof(int);                            \
    int per_cu = 0;                                                                                                      \
    int64_t gx = 128;                                                                                                    \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, PSDF_BLOCK, shm) == hipSuccess && per_cu > 0)        \
      gx = (int64_t)per_cu * device_cus() / nr_levels;   /* one resident round, as in psdf_encode_backward_ws */         \
    else                                                                                                                 \
      (void)hipGetLastError();                                                                                           \
    const int64_t super_tiles = (N + (int64_t)PSDF_BLOCK * QUEUE_SPT - 1) / ((int64_t)PSDF_BLOCK * QUEUE_SPT);           \
    if (gx > super_tiles) gx = super_tiles;                                                                              \
    /* the round dealt over the levels by their measured cost (the concatenated-point channels are levels of the deal): */ \
    /* a training step's coarse-to-fine window leaves levels closed, whose workgroups return at once */                   \
    LevelPlan plan{};                                                                                                    \
    dim3 g2((unsigned)(gx < 1 ? 1 : gx), Lt);                                                                            \
    if (per_cu > 0 && encode_balance(Lt, N, per_cu * device_cus(), super_tiles, plan, lattice, 1)) g2 = dim3(plan.first[Lt]);       \
    else plan = LevelPlan{};                                                                                             \
    hipLaunchKernelGGL(kern, g2, dim3(PSDF_BLOCK), shm, st, N, nr_levels,                                                \
                       (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window,     \
                       points_scaling, grad_sliced, grad_lattice, (float*)nullptr, Q, dd_positions, grad_grad_sliced,    \
                       pad_points(concat_points), grad_sliced_direct, plan);                                             \
    const size_t lds_b = (size_t)(1 << Q.shift) * F_ * sizeof(float);                                                    \
    hipError_t e2 = hipFuncSetAttribute((const void*)encode_bwd_reduce_kernel<F_>,                                       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);                         \
    if (e2 != hipSuccess) return (int)e2;                                                                                \
    hipLaunchKernelGGL((encode_bwd_reduce_kernel<F_>), dim3(Q.np, nr_levels), dim3(1024), lds_b, st, (uint32_t)capacity, \
                       Q, grad_lattice);                                                                                 \
  } while (0)
      if (pos_dim == 3 && nr_feat == 2)
        DBLQ(3, 2);
      else if (pos_dim == 4 && nr_feat == 2)
        DBLQ(4, 2);
      else if (pos_dim == 2 && nr_feat == 2)
        DBLQ(2, 2);
      else if (pos_dim == 3 && nr_feat == 4)
        DBLQ(3, 4);
      else
        return PSDF_ERR_UNSUPPORTED;
#undef DBLQ
      PSDF_LAUNCH_CHECK();
      return PSDF_OK;
    }
  }
  // small batches: the plain double-backward kernel (it always writes the gathered output: give it scratch when the caller
  // wants the lattice gradient only), and the direct part as an ordinary backward
  if (grad_sliced_direct) {
    const int rc = psdf_encode_backward_ws(pos_dim, nr_feat, N, nr_levels, capacity, positions, lattice, scale_factor, shifts,
                                           window, concat_points, points_scaling, grad_sliced_direct, grad_lattice, nullptr,
                                           nullptr, 0, stream);
    if (rc != PSDF_OK) return rc;
  }
  if (!grad_grad_sliced) {
    grad_grad_sliced = (float*)psdf::stream_scratch((size_t)Lt * nr_feat * N * sizeof(float), st);
    if (!grad_grad_sliced) return PSDF_ERR_UNSUPPORTED;
  }
  // Plain (cache + atomics) mode: a workgroup sets up and flushes a 48-KiB LDS cache, so it should walk several tiles;
  // PSDF_ENC_BWD_WG_PER_LEVEL overrides the per-level workgroup cap (measurement switch).
  static const int wg_cap_env = getenv("PSDF_ENC_BWD_WG_PER_LEVEL") ? atoi(getenv("PSDF_ENC_BWD_WG_PER_LEVEL")) : 0;
  const unsigned wg_cap = wg_cap_env > 0 ? (unsigned)wg_cap_env : 512u;
  dim3 grid(nb < wg_cap ? nb : wg_cap, Lt);
#define DBL(P_, F_)                                                                                              \
  do {                                                                                                           \
    if (grad_lattice)                                                                                            \
      hipLaunchKernelGGL((encode_dbl_bwd_kernel<P_, F_, true>), grid, dim3(PSDF_BLOCK), ScatterCache<F_>::bytes(), \
                         st, N, nr_levels, (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window,   \
                         points_scaling, pad_points(concat_points), dd_positions, grad_sliced, grad_lattice,       \
                         grad_grad_sliced);                                                                        \
    else                                                                                                         \
      hipLaunchKernelGGL((encode_dbl_bwd_kernel<P_, F_, false>), grid, dim3(PSDF_BLOCK), 0, st, N, nr_levels,      \
                         (uint32_t)capacity, psdf::enc_conv_state(), positions, lattice, scale_factor, shifts, window, points_scaling,     \
                         pad_points(concat_points), dd_positions, grad_sliced, grad_lattice, grad_grad_sliced);    \
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

int psdf_encode_double_backward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity,
                                const float* positions, const float* lattice, const float* scale_factor,
                                const float* shifts, const float* window, int concat_points, float points_scaling,
                                const float* dd_positions, const float* grad_sliced, float* grad_lattice,
                                float* grad_grad_sliced, void* stream) {
  return psdf_encode_double_backward_ws(pos_dim, nr_feat, N, nr_levels, capacity, positions, lattice, scale_factor, shifts,
                                        window, concat_points, points_scaling, dd_positions, grad_sliced, grad_lattice,
                                        grad_grad_sliced, nullptr, nullptr, 0, stream);
}

}  // extern "C"
