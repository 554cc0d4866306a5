// Device-side simplex location / hashing of the permutohedral encoding (shared by encode.hip and fused.hip).
// Conventions: SURVEY.md App. A / oracle/permuto_oracle.py.
#pragma once
#include "encode_conventions.h"
#include "psdf_common.h"

// The two conventions that live in device code (encode_conventions.h) as RUNTIME values: a kernel argument, wave-uniform.
// Defaults are the #defines; psdf_encode_set_conventions() (C ABI) replaces them for the process, so that matching an upstream
// build that disagrees is a flag flip and not a rebuild.
struct EncConv {
  uint32_t hash_c;      // PSDF_ENC_HASH_MULTIPLIER
  uint32_t tie_later;   // PSDF_ENC_RANK_TIE_RAISES_LATER
};
namespace psdf {
EncConv& enc_conv_state();   // defined in encode.hip (host)
}

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
                                                const float* __restrict__ sf, Simplex<P>& s, uint32_t tie_later) {
  float E[P + 1];
  float sm = 0.f;
#pragma unroll
  for (int i = P; i > 0; i--) {
    float cf = (pos[i - 1] + shift[i - 1]) * sf[i - 1];
    E[i] = sm - (float)i * cf;
    sm = sm + cf;
  }
  E[0] = sm;

  // The frozen convention (oracle/permuto_oracle.py) divides by P+1 in double and rounds once.  When P+1 is a power of
  // two (pos_dim 3: the SDF / colour lattices) that is an exact scaling, so the float product is bit-identical and the
  // f64 converts and multiplies (half / quarter rate) are skipped.
  constexpr bool POW2 = ((P + 1) & P) == 0;
  const double inv = 1.0 / (P + 1);
  const float invf = 1.0f / (P + 1);
  int sum = 0;
  float remf[P + 1];
#pragma unroll
  for (int i = 0; i <= P; i++) {
    float v = POW2 ? E[i] * invf : (float)((double)E[i] * inv);
    float down = floorf(v) * (float)(P + 1);
    if (POW2) {
      // The nearer of the two multiples of P+1 around E, the lower one on a tie: E > down + (P+1)/2 -- the midpoint is exact, no
      // rounded differences to compare.  Same answer as the reference's (up - E) < (E - down) on every float (those differences
      // are exact or round monotonically towards the midpoint; tests/test_dpp_scan_emulation.py sweeps the neighbourhoods of all
      // multiples of 1/2 and 5e7 random values): floor, two packed products / sums, a compare, a select -- was ceil, floor, two
      // products, two differences, a compare, a select.
      const float mid = down + 0.5f * (float)(P + 1), up = down + (float)(P + 1);
      remf[i] = (E[i] > mid) ? up : down;
    } else {
      float up = ceilf(v) * (float)(P + 1);
      remf[i] = ((up - E[i]) < (E[i] - down)) ? up : down;
    }
    // (integer valued: (float)(int)remf == remf, one convert instead of two)
    s.rem0[i] = (int)remf[i];
    sum += s.rem0[i];
  }
  // the remainders add up to a multiple of P+1 (the elevated coordinates add up to 0): for a power of two an arithmetic shift
  sum = POW2 ? (sum >> __builtin_ctz(P + 1)) : (sum / (P + 1));

  float d[P + 1];
#pragma unroll
  for (int i = 0; i <= P; i++) {
    d[i] = E[i] - remf[i];
    s.rank[i] = 0;
  }
  // tie rule (encode_conventions.h): a wave-uniform value, so ONE scalar branch picks the comparison for all pairs
  if (tie_later) {
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
  } else {
#pragma unroll
    for (int i = 0; i < P; i++) {
#pragma unroll
      for (int j = i + 1; j <= P; j++) {
        if (d[i] <= d[j])
          s.rank[i]++;
        else
          s.rank[j]++;
      }
    }
  }
#pragma unroll
  for (int i = 0; i <= P; i++) {   // wrap into 0..P; as selects: written as if / else-if the compiler emits a divergent branch per i
    const int r = s.rank[i] + sum;
    // P + 1 a power of two: the wrap by +-(P+1) is the two's-complement remainder (r lies in [-(P+1), 2P+1]) -- add, and, sub
    // instead of two compares and two selects, and the compiler now KNOWS 0 <= rank <= P (vertex_rows drops its r = 0 terms)
    const int adj = POW2 ? ((r & P) - r) : ((r < 0) ? (P + 1) : ((r > P) ? -(P + 1) : 0));
    s.rank[i] = r + adj;
    s.rem0[i] += adj;
  }
  // recompute d after the fix-up (rem0 may have moved by +-(P+1)); same expression as the oracle.
  // The oracle scatters: bary[P - rank_i] += delta_i; bary[P + 1 - rank_i] -= delta_i.  The ranks are a permutation,
  // so every bary[k] receives exactly one "+" (from the element of rank P-k) and one "-" (rank P+1-k): gathering the
  // deltas by rank first and forming bary[k] = ds[P-k] - ds[P+1-k] is the same single rounding with (P+1)^2 selects
  // instead of 2(P+1)(P+2).
  float ds[P + 1];
#pragma unroll
  for (int r = 0; r <= P; r++) ds[r] = 0.f;
#pragma unroll
  for (int i = 0; i <= P; i++) {
    float delta = POW2 ? (E[i] - (float)s.rem0[i]) * invf : (float)((double)(E[i] - (float)s.rem0[i]) * inv);
#pragma unroll
    for (int r = 0; r <= P; r++)
      if (s.rank[i] == r) ds[r] = delta;
  }
  // (the oracle accumulates into zero-initialised slots, 0 + ds: that differs from ds only in the sign of a zero, which no
  // consumer sees -- a barycentric coordinate is a factor of products that are added to sums)
  s.bary[0] = ds[P];
#pragma unroll
  for (int k = 1; k <= P; k++) s.bary[k] = ds[P - k] - ds[P + 1 - k];
  s.bary[P + 1] = 0.f - ds[0];
  s.bary[0] = (float)((double)s.bary[0] + (1.0 + (double)s.bary[P + 1]));
}

// Hash of the vertex with remainder r: key_i = rem0_i + r - (P+1)*[rank_i > P - r],  h = (..((key_0)*c + key_1)*c ..)*c.
// In the ring of 32-bit integers that is  sum_i key_i c^(P-i), so with H0 = sum_i rem0_i c^(P-i) (P multiplies, ONCE per
// simplex) every vertex is  H0 + r*(c^P + .. + c) - sum_i [rank_i > P - r] * (P+1) c^(P-i):  adds and selects only
// (integer multiplies are quarter rate and the direct form needs P of them per vertex).  The multiplier c is a kernel argument
// (wave-uniform): its powers and their sum are scalar-unit arithmetic, computed once per wave.

// All P+1 rows of a simplex.  `capacity` is wave-uniform (a kernel argument): ONE scalar branch on the kind of modulo for all
// vertices (a power of two is a mask; the general case a ~35-instruction runtime modulo per vertex).
template <int P>
__device__ __forceinline__ void vertex_rows(const Simplex<P>& s, uint32_t capacity, uint32_t (&rows)[P + 1], uint32_t hash_c) {
  uint32_t pw[P + 1];   // pw[e] = c^e
  pw[0] = 1u;
#pragma unroll
  for (int e = 1; e <= P; e++) pw[e] = pw[e - 1] * hash_c;
  uint32_t geom = 0;    // c^P + .. + c
#pragma unroll
  for (int e = 1; e <= P; e++) geom += pw[e];
  uint32_t h0 = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
    h0 += (uint32_t)s.rem0[i];
    h0 *= hash_c;
  }
  // sum_i [rank_i > P - r] t_i with t_i = (P+1) c^(P-i): the ranks are a permutation, so gather T_k = t_i of the coordinate of
  // rank k (the same rank_i == k compares as the barycentric gather of compute_simplex: the compiler shares them) and run the
  // sum up from k = P: vertex r subtracts T_P + .. + T_(P+1-r).
  uint32_t T[P + 1];
#pragma unroll
  for (int k = 1; k <= P; k++) {
    T[k] = 0u;
#pragma unroll
    for (int i = 0; i < P; i++)
      if (s.rank[i] == k) T[k] = (uint32_t)(P + 1) * pw[P - i];
  }
  rows[0] = h0;
  uint32_t S = 0u;
#pragma unroll
  for (int r = 1; r <= P; r++) {
    S += T[P + 1 - r];
    rows[r] = h0 + ((uint32_t)r * geom - S);
  }
  if ((capacity & (capacity - 1u)) == 0u) {
#pragma unroll
    for (int r = 0; r <= P; r++) rows[r] &= capacity - 1u;
  } else {
#pragma unroll
    for (int r = 0; r <= P; r++) rows[r] %= capacity;
  }
}

template <int P>
__device__ __forceinline__ void load_pos(const float* __restrict__ positions, int64_t n, float* pos) {
#pragma unroll
  for (int i = 0; i < P; i++) pos[i] = positions[n * P + i];
}

}  // namespace
