// Shared device helpers for the gfx950 kernels of the PermutoSDF hot path.
// Everything here is written for CDNA4 only (wave64, no CUDA fallbacks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// All kernels are built with -ffp-contract=off: every a*b+c below is two roundings unless it is an
// explicit fmaf(), so results do not depend on the compiler's fusion choices and match the CPU oracle.

#define PSDF_OK 0
#define PSDF_ERR_ARG (-1)
#define PSDF_ERR_UNSUPPORTED (-2)

#define PSDF_BLOCK 256
#define PSDF_WAVE 64

#define PSDF_LAUNCH_CHECK()                        \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline unsigned psdf_blocks(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

namespace psdf {

struct v3 {
  float x, y, z;
};
__device__ __forceinline__ v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 ld3(const float* p) { return v3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, v3 a) {
  p[0] = a.x;
  p[1] = a.y;
  p[2] = a.z;
}
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return v3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// point on a ray: o + t*d, one rounding per op (the reference's `ray_origin+t*ray_dir`)
__device__ __forceinline__ v3 along(v3 o, float t, v3 d) { return o + t * d; }

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- PCG32 (PCG-XSH-RR 64/32), the generator family the reference uses for jitter
//      (reference kernels/permuto_sdf/pcg32.h:45-206).  State is passed by value to kernels.
#define PSDF_PCG_DEFAULT_STATE 0x853c49e6748fea9bULL
#define PSDF_PCG_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define PSDF_PCG_MULT 0x5851f42d4c957f2dULL
struct Pcg {
  uint64_t state, inc;
  __host__ __device__ uint32_t next_uint() {
    uint64_t old = state;
    state = old * PSDF_PCG_MULT + inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
  }
  // uniform in [0,1): 23 mantissa bits in [1,2) minus 1
  __host__ __device__ float next_float() {
    union {
      uint32_t u;
      float f;
    } c;
    c.u = (next_uint() >> 9) | 0x3f800000u;
    return c.f - 1.0f;
  }
  // O(log delta) jump ahead (Brown 1994)
  __host__ __device__ void advance(uint64_t delta) {
    uint64_t cur_mult = PSDF_PCG_MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
      if (delta & 1) {
        acc_mult *= cur_mult;
        acc_plus = acc_plus * cur_mult + cur_plus;
      }
      cur_plus = (cur_mult + 1) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1;
    }
    state = acc_mult * state + acc_plus;
  }
};

// ---- wave64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive sum scan over the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}
// Integer inclusive scan over the wave on the DPP path: four shifted adds inside each 16-lane row (0 shifted in at the row's
// edge: bound_ctrl), then the last lane of row 0 / 2 added to row 1 / 3 (row_bcast:15, rows 1 and 3 enabled) and lane 31 to rows
// 2 and 3 (row_bcast:31).  Six VALU instructions; the __shfl_up form was six ds_bpermute round trips in a dependent chain plus
// a lane compare and a select each (and six 64-bit lane masks: SGPR spills in the encode binning kernel).  Integer adds: the
// result is the same whatever the order.
__device__ __forceinline__ int wave_incl_scan_add_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
// inclusive product scan
__device__ __forceinline__ float wave_incl_scan_mul(float v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (l >= o) v *= t;
  }
  return v;
}

// Per-stream scratch owned by the library (defined in mlp_bwd.hip): grows on demand, is never handed back, and is safe
// to re-use from call to call because the calls of one stream execute in order.  A stream-ordered allocation per call
// (hipMallocAsync) cost ~0.2 ms of HOST time each, which the host-bound training step could not hide.
// Returns NULL while the stream is being captured (a later growth would move the buffer under the graph) or when the
// allocation fails: callers then take their allocation-free path.
void* stream_scratch(size_t bytes, hipStream_t st);

// Debug record of the kernel variant the LAST call of an operator family dispatched to (host side, no device work; defined in
// mlp_bwd.hip, read through psdf_last_path()).  The parity tests use it to assert that the configuration they compare with
// the oracle really ran the kernels the benchmark times (the split-bf16 MLP kernels, the queue-mode encode backward).
enum { PATH_ENCODE_BWD = 0, PATH_MLP_BWD = 1, PATH_MLP_FWD = 2, PATH_FAMILIES = 8 };
extern int g_last_path[PATH_FAMILIES];
}  // namespace psdf
