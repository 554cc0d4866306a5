// Sample generation for gfx950: Morton occupancy grid (update, DDA ray marching, queries), ray samplers
// (foreground uniform, NeRF++ style background), bounding sphere, packed-sample container utilities,
// spherical harmonics and random ray generation from an image stack.
// Replaces, with the same per-item arithmetic (fp32, one rounding per operation, see psdf_common.h):
//   src/OccupancyGrid.cu + kernels/permuto_sdf/OccupancyGridGPU.cuh      (all kernels)
//   src/RaySampler.cu    + kernels/permuto_sdf/RaySamplerGPU.cuh         (:37 bg, :162 fg)
//   src/Sphere.cu        + kernels/permuto_sdf/SphereGPU.cuh             (:21, :96)
//   src/RaySamplesPacked.cu + kernels/permuto_sdf/RaySamplesPackedGPU.cuh (:15 compact, :84 per-sample ray idx)
//   src/PermutoSDF.cu    + kernels/permuto_sdf/PermutoSDFGPU.cuh         (:24 random_rays_from_reel, :275 SH)
//
// Differences in STRUCTURE (not arithmetic): every kernel that reserved output slots with atomicAdd on a
// global counter (nondeterministic ray order, OccupancyGridGPU.cuh:599, RaySamplerGPU.cuh:228,
// RaySamplesPackedGPU.cuh:51) is split into count -> exclusive scan over rays -> fill, so packed samples are
// ray-ordered, reproducible and exactly sized (no holes, no compaction copy).
#include <cstdlib>
#include "psdf_common.h"
#include <stdlib.h>

using namespace psdf;

namespace {

// ---------------------------------------------------------------------------------- Morton helpers
__host__ __device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
// the same spreading for v < 1024 (every in-grid coordinate) with shifts instead of the four quarter-rate integer multiplies:
// v * (2^k + 1) = v | (v << k) when the two terms share no bit, which the masks guarantee from 10 bits on
__host__ __device__ __forceinline__ uint32_t expand_bits10_small(uint32_t v) {
  v = (v | (v << 16)) & 0xFF0000FFu;
  v = (v | (v << 8)) & 0x0F00F00Fu;
  v = (v | (v << 4)) & 0xC30C30C3u;
  v = (v | (v << 2)) & 0x49249249u;
  return v;
}
__host__ __device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  if ((x | y | z) < 1024u) return expand_bits10_small(x) | (expand_bits10_small(y) << 1) | (expand_bits10_small(z) << 2);
  return expand_bits10(x) | (expand_bits10(y) << 1) | (expand_bits10(z) << 2);   // out-of-grid coordinates: the reference's arithmetic
}
__host__ __device__ __forceinline__ uint32_t compact_bits10(uint32_t x) {
  x &= 0x49249249u;
  x = (x | (x >> 2)) & 0xc30c30c3u;
  x = (x | (x >> 4)) & 0x0f00f00fu;
  x = (x | (x >> 8)) & 0xff0000ffu;
  x = (x | (x >> 16)) & 0x0000ffffu;
  return x;
}
// float -> uint32 with the device semantics the reference relies on (negative / NaN -> 0, huge / +inf -> 2^32-1, truncation
// otherwise): exactly what v_cvt_u32_f32 does in hardware (it saturates, NaN converts to 0), so ONE instruction instead of
// two compares, two branches and the conversion -- three times per step of every DDA loop.  (A C cast would be undefined
// for the out-of-range values; the instruction is named explicitly.)  tests/test_gpu_sampling.py feeds NaN / inf / huge /
// negative coordinates through it against the oracle's explicit form.
__device__ __forceinline__ uint32_t sat_u32(float f) {
  uint32_t r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}

struct Grid {
  int n;         // voxels per dimension (power of two)
  float extent;  // edge length of the cube
  float tx, ty, tz;
  // The reference divides by `extent` and by `n` in every DDA step (two correctly rounded fp32 divisions, ~10 instructions
  // each, on the dependent chain that bounds these loops).  Dividing by 1.0 is the identity and dividing by a power of two is
  // an exact scaling, so for the grids the reference builds (extent 1, n = 256) both become bit-identical cheap forms.
  float inv_n;   // 1 / n when n is a power of two (exact), else 0: use the division
  int unit;      // extent == 1
  __device__ __forceinline__ float div_n(float x) const { return inv_n != 0.f ? x * inv_n : x / n; }
  __device__ __forceinline__ float div_extent(float x) const { return unit ? x : x / extent; }
  __device__ __forceinline__ int nr_voxels() const { return n * n * n; }
  // voxel centre (or corner) of a Morton index (OccupancyGridGPU.cuh:112-155)
  __device__ __forceinline__ v3 idx_to_pos(uint32_t idx, bool centre) const {
    float x = (float)compact_bits10(idx), y = (float)compact_bits10(idx >> 1), z = (float)compact_bits10(idx >> 2);
    x = div_n(x);
    y = div_n(y);
    z = div_n(z);
    x = x - 0.5f;
    y = y - 0.5f;
    z = z - 0.5f;
    if (centre) {
      const float half = (float)((double)(float)(1.0 / n) / 2);
      x += half;
      y += half;
      z += half;
    }
    return mk3(x * extent + tx, y * extent + ty, z * extent + tz);
  }
  // world position -> Morton index (OccupancyGridGPU.cuh:158-193, get_center_of_voxel=false)
  __device__ __forceinline__ int pos_to_idx(v3 p) const {
    float x = div_extent(p.x - tx), y = div_extent(p.y - ty), z = div_extent(p.z - tz);
    x = (x + 0.5f) * n;
    y = (y + 0.5f) * n;
    z = (z + 0.5f) * n;
    return (int)morton3(sat_u32(x), sat_u32(y), sat_u32(z));
  }
  __device__ __forceinline__ bool in_range(int idx) const { return !(idx >= nr_voxels() || idx < 0); }
};

__device__ __forceinline__ int sgn(float x) { return x > 0 ? 1 : (x < 0 ? -1 : 0); }

// DDA step to the next voxel face, in world units (OccupancyGridGPU.cuh:95-109)
__device__ __forceinline__ float dist_to_next_voxel(v3 pos, v3 dir, v3 idir, const Grid& g) {
  const int n = g.n;
  pos = (float)n * pos;
  const float tx = (floorf(pos.x + 0.5f + 0.5f * sgn(dir.x)) - pos.x) * idir.x;
  const float ty = (floorf(pos.y + 0.5f + 0.5f * sgn(dir.y)) - pos.y) * idir.y;
  const float tz = (floorf(pos.z + 0.5f + 0.5f * sgn(dir.z)) - pos.z) * idir.z;
  const float t = fminf(fminf(fabsf(tx), fabsf(ty)), fabsf(tz));
  return fmaxf(g.div_n(t), 0.0f);
}
// The same grid with what the reference's grids always have folded in at compile time: extent 1 (dividing by it is the
// identity) and a power-of-two n (dividing by it is an exact scaling).  The DDA loops are latency chains -- a step is ~150
// instructions in the generic form, a third of them the run-time selection between the exact shortcuts and the divisions,
// scalar bookkeeping and the re-derivation of n^3 -- so the kernels are instantiated for both types and the host picks
// (mk_grid: `unit` and `inv_n`).  Same expressions in the same order: bit-identical results.
struct GridFast {
  int n;
  float nf;       // (float)n, exact
  int nvox;       // n^3
  float tx, ty, tz;
  float inv_n;    // 1 / n, exact
  __device__ __forceinline__ int nr_voxels() const { return nvox; }
  __device__ __forceinline__ float div_n(float x) const { return x * inv_n; }
  __device__ __forceinline__ int pos_to_idx(v3 p) const {
    float x = p.x - tx, y = p.y - ty, z = p.z - tz;
    x = (x + 0.5f) * nf;
    y = (y + 0.5f) * nf;
    z = (z + 0.5f) * nf;
    return (int)morton3(sat_u32(x), sat_u32(y), sat_u32(z));
  }
  __device__ __forceinline__ bool in_range(int idx) const { return !(idx >= nvox || idx < 0); }
};
__device__ __forceinline__ float dist_to_next_voxel(v3 pos, v3 dir, v3 idir, const GridFast& g) {
  pos = g.nf * pos;
  const float tx = (floorf(pos.x + 0.5f + 0.5f * sgn(dir.x)) - pos.x) * idir.x;
  const float ty = (floorf(pos.y + 0.5f + 0.5f * sgn(dir.y)) - pos.y) * idir.y;
  const float tz = (floorf(pos.z + 0.5f + 0.5f * sgn(dir.z)) - pos.z) * idir.z;
  const float t = fminf(fminf(fabsf(tx), fabsf(ty)), fabsf(tz));
  return fmaxf(t * g.inv_n, 0.0f);
}
__device__ __forceinline__ v3 safe_inverse(v3 d) {
  v3 r;
  r.x = fabsf(d.x) < 1e-16f ? 0.f : (float)(1.0 / (double)d.x);
  r.y = fabsf(d.y) < 1e-16f ? 0.f : (float)(1.0 / (double)d.y);
  r.z = fabsf(d.z) < 1e-16f ? 0.f : (float)(1.0 / (double)d.z);
  return r;
}
constexpr int MAX_DDA_STEPS = 4096;
#if !defined(PSDF_MARCH_AHEAD)
#define PSDF_MARCH_AHEAD 4     // steps the first march walks ahead of its occupancy probes (march_kernel)
#endif
constexpr float DDA_EPS = 1e-6f;

// Occupancy as the DDA loops see it: the reference's one byte per voxel (Morton order) plus an OPTIONAL coarse mask, one
// bit per 8x8x8 block (512 consecutive Morton indices) = "some voxel of the block is occupied" (occupancy_coarse_kernel).
// The mask of a 256^3 grid is 4 KiB: every workgroup copies it into LDS, and a probe of a voxel whose block is empty is
// answered there.  Why: the loops are chains of dependent, uncoalesced 1-byte probes -- latency bound for the few
// hundred rays of a training step (~0.7 us per step of the march), texture-addresser bound for the 2 M rays of a
// render; rays spend most of their steps in empty space.  The answer is the same bit for bit (the arithmetic of the march
// is untouched), it only arrives from LDS.
struct Occ {
  const uint8_t* bytes;
  const uint32_t* coarse;  // NULL: none
  int words;               // 32-bit words of the mask
};
constexpr int COARSE_SHIFT = 9;
// workgroup-collective; call before any thread returns
__device__ __forceinline__ const uint32_t* stage_coarse(const Occ& o, uint32_t* lds) {
  if (!o.coarse) return nullptr;
  for (int i = threadIdx.x; i < o.words; i += blockDim.x) lds[i] = o.coarse[i];
  __syncthreads();
  return lds;
}
__device__ __forceinline__ bool probe(const Occ& o, const uint32_t* cm, int vox) {
  if (cm) {
    const uint32_t c = (uint32_t)vox >> COARSE_SHIFT;
    if (!((cm[c >> 5] >> (c & 31u)) & 1u)) return false;
  }
  return o.bytes[vox] != 0;
}
// one wave per mask word: 32 blocks x 512 bytes, read 1 KiB (two blocks) at a time
__global__ void __launch_bounds__(PSDF_BLOCK)
    occupancy_coarse_kernel(int words, const uint8_t* __restrict__ occ, uint32_t* __restrict__ coarse) {
  const int w = blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6);
  if (w >= words) return;
  const int lane = threadIdx.x & 63;
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(occ + ((size_t)w << (COARSE_SHIFT + 5)));
  uint32_t bits = 0u;
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const uint4 v = src[it * 64 + lane];
    const unsigned long long any = __ballot((v.x | v.y | v.z | v.w) != 0u);
    if ((uint32_t)any) bits |= 1u << (2 * it);
    if ((uint32_t)(any >> 32)) bits |= 1u << (2 * it + 1);
  }
  if (lane == 0) coarse[w] = bits;
}

// ---------------------------------------------------------------------------------- grid points
__global__ void __launch_bounds__(PSDF_BLOCK)
    grid_points_kernel(int count, Grid g, const int* __restrict__ indices, Pcg rng, int randomize,
                       float* __restrict__ out) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const uint32_t vox = indices ? (uint32_t)indices[i] : (uint32_t)i;
  v3 p = g.idx_to_pos(vox, true);
  if (randomize) {
    const float voxel = g.extent / g.n;
    const float half = (float)((double)voxel / 2.0);
    rng.advance((uint64_t)(int64_t)(i * 3));
    p.x += voxel * rng.next_float() - half;
    p.y += voxel * rng.next_float() - half;
    p.z += voxel * rng.next_float() - half;
  }
  st3(out + 3 * (int64_t)i, p);
}

// ---------------------------------------------------------------------------------- grid updates
__global__ void __launch_bounds__(PSDF_BLOCK)
    update_density_kernel(int count, const int* __restrict__ indices, const float* __restrict__ density, float decay,
                          float thresh, float* __restrict__ values, uint8_t* __restrict__ occ) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const int vox = indices ? indices[i] : i;
  const float v = fmaxf(density[i], values[vox] * decay);
  values[vox] = v;
  occ[vox] = v > thresh;
}
// NeuS logistic density: s e^{-sx} / (1 + e^{-sx})^2 (OccupancyGridGPU.cuh:381-384)
__device__ __forceinline__ float logistic_density(float x, float s) {
  const float e = expf(-s * x);
  return s * e / powf(1 + e, 2);
}
__global__ void __launch_bounds__(PSDF_BLOCK)
    update_sdf_kernel(int count, const int* __restrict__ indices, const float* __restrict__ sdf, Grid g, float inv_s,
                      const float* __restrict__ inv_s_tensor, int full_update, float thresh,
                      float* __restrict__ values, uint8_t* __restrict__ occ) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const int vox = indices ? indices[i] : i;
  const float voxel = g.extent / g.n;
  const float half = (float)((double)voxel / 2.0);
  const float half_diag = sqrtf(3.0f) * half;
  const float v = sdf[i];
  values[vox] = v;
  const float err = (float)((full_update ? 1.3 : 1.0) * (double)half_diag);
  const float x = fmaxf(0.f, fminf(fabsf(v) - err, 1e10f));
  const float s = inv_s_tensor ? inv_s_tensor[0] : inv_s;
  occ[vox] = logistic_density(x, s) > thresh;
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    check_occupancy_kernel(int count, Grid g, const uint8_t* __restrict__ occ, const float* __restrict__ pts,
                           uint8_t* __restrict__ out) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const int vox = g.pos_to_idx(ld3(pts + 3 * (int64_t)i));
  out[i] = g.in_range(vox) ? occ[vox] : 0;
}

// ---------------------------------------------------------------------------------- DDA samplers
// One thread per ray (the march is inherently serial and its float sequence is an index-exact contract).
// WRITE=false: count the samples this ray will emit.  WRITE=true: emit them at offsets[ray].
// USE_GRID=false gives the reference's compute_samples_fg (RaySamplerGPU.cuh:162) = same march without occupancy.
// Pass 1 of 3 (thread per ray): the reference's two marches (occupied length, then sample placement) run ONCE; the
// distances t of the samples it places go to a per-ray scratch row, the count and the spacing to per-ray scratch.
// Pass 2 is an exclusive scan of the counts; pass 3 (march_fill_kernel, wave per ray) turns the t values into samples
// at the ray's exact offset with coalesced stores.  (The first version of this file ran the whole march twice, once
// to count and once to write: the DDA loop is a chain of dependent 1-byte grid probes, the most expensive thing in a
// volume render after the network itself.)
template <bool USE_GRID, typename G>
__device__ __forceinline__ void march_ray(int ray, const G& g, const Occ& o, const uint32_t* cm, const float* __restrict__ origins,
                                          const float* __restrict__ dirs, const float* __restrict__ t_entry,
                                          const float* __restrict__ t_exit_p, float min_dist, int max_per_ray, Pcg rng,
                                          int jitter, int* __restrict__ counts, float* __restrict__ spacings,
                                          float* __restrict__ ztemp) {
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const v3 idir = safe_inverse(dir);
  const float t_start = t_entry[ray], t_exit = t_exit_p[ray];
  float occupied = 0.f;
  if (USE_GRID) {
    // The walk of this first march does not depend on what it finds (only `occupied` does), so it runs AHEAD steps ahead of
    // its probes: AHEAD probes in flight instead of one (a training step marches a few hundred rays: nothing else hides
    // the latency).  Same float sequence, same order of the additions into `occupied`.
    constexpr int AHEAD = PSDF_MARCH_AHEAD;
    float t = t_start;
    int steps = 0;
    bool walking = true;
    while (walking) {
      int vox[AHEAD];
#pragma unroll
      for (int k = 0; k < AHEAD; k++) vox[k] = 0;
      float dd[AHEAD], tt[AHEAD];
      int m = 0;
#pragma unroll
      for (int k = 0; k < AHEAD; k++) {
        if (walking) {
          if (!(t < t_exit && steps < MAX_DDA_STEPS)) {
            walking = false;
          } else {
            const v3 pos = along(org, t, dir);
            const int v = g.pos_to_idx(pos);
            if (!g.in_range(v)) {
              walking = false;
            } else {
              const float d = dist_to_next_voxel(pos, dir, idir, g);
              t += d;
              t += DDA_EPS;
              vox[k] = v;
              dd[k] = d;
              tt[k] = t;
              m = k + 1;
              steps++;
            }
          }
        }
      }
      // unconditional reads (slots past m re-read voxel 0): a guarded read is a branch, and branches serialise the probes
      uint8_t byte[AHEAD];
#pragma unroll
      for (int k = 0; k < AHEAD; k++) byte[k] = o.bytes[k < m ? vox[k] : 0];
#pragma unroll
      for (int k = 0; k < AHEAD; k++)
        if (k < m && byte[k]) {
          occupied += dd[k];
          if ((tt[k] - DDA_EPS) > t_exit) occupied -= (tt[k] - DDA_EPS) - t_exit;
        }
    }
  } else {
    occupied = t_exit - t_start;
  }
  int to_create = (int)(occupied / min_dist);
  to_create = clampi(to_create, 0, max_per_ray);
  const float spacing = occupied / to_create;
  int created = 0;
  float* __restrict__ zrow = ztemp + (int64_t)ray * max_per_ray;
  const bool go = USE_GRID ? (to_create > 1) : (to_create > 1 && occupied > DDA_EPS);
  if (go) {
    float t = t_start;
    int steps = 0;
    if (jitter) {
      rng.advance((uint64_t)(int64_t)ray);
      t = t + spacing * rng.next_float();
    }
    while (t < t_exit && steps < MAX_DDA_STEPS) {
      t = fmaxf(t_start, fminf(t, t_exit));
      bool occupied_here = true;
      v3 pos = along(org, t, dir);
      if (USE_GRID) {
        const int vox = g.pos_to_idx(pos);
        if (!g.in_range(vox)) break;
        occupied_here = probe(o, cm, vox);
      }
      if (occupied_here && created < to_create) {
        zrow[created] = t;
        t += spacing;
        created++;
      } else if (USE_GRID) {
        float delta = dist_to_next_voxel(pos, dir, idir, g);
        if (jitter) delta = delta + spacing * rng.next_float();
        t += delta;
        t += DDA_EPS;
      } else {
        break;  // all samples of a grid-less ray are out; the reference idles here until its step bound
      }
      steps++;
    }
    if (created <= 2) created = 0;  // rays with <= 2 samples are dropped (OccupancyGridGPU.cuh:685-689)
  }
  counts[ray] = created;
  spacings[ray] = spacing;
}
template <bool USE_GRID, typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    march_kernel(int nr_rays, G g, Occ o, const float* __restrict__ origins,
                 const float* __restrict__ dirs, const float* __restrict__ t_entry, const float* __restrict__ t_exit_p,
                 float min_dist, int max_per_ray, Pcg rng, int jitter, int* __restrict__ counts,
                 float* __restrict__ spacings, float* __restrict__ ztemp) {
  extern __shared__ uint32_t cm_lds[];
  const uint32_t* cm = stage_coarse(o, cm_lds);
  const int ray = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (ray >= nr_rays) return;
  march_ray<USE_GRID, G>(ray, g, o, cm, origins, dirs, t_entry, t_exit_p, min_dist, max_per_ray, rng, jitter, counts, spacings,
                         ztemp);
}

// ---------------------------------------------------------------------------------- the march of a FEW rays (round 6)
// A training step marches a few hundred rays: march_kernel then runs on a dozen waves, one per SIMD, and its duration is the
// length of ONE ray's dependent chain -- ~900 DDA steps of ~100 instructions (4 cycles of issue each for the one wave of a SIMD)
// plus a global probe's latency per step or group of steps: 207 us for 727 rays (profiles/r05_cfg4_manual_kernel_stats.txt).
// The float sequence of a ray is a contract (sample counts and positions are bit-exact against the reference), so the walk
// itself cannot be split over lanes by distance; what CAN be split is the work of one step:
//   * FOUR LANES PER RAY (a quad): lane c < 3 owns axis c (lane 3 repeats axis 2), so position, voxel coordinate and face
//     distance of the three axes are ONE instruction stream of scalar-sized work; the three-way minimum and the voxel key are
//     combined with quad_perm DPP (no LDS, no extra latency).  Same expressions, same order, per axis: bit-identical.
//   * the first march (occupied length) never waits for memory: the walk does not depend on what it finds, so it only RECORDS
//     (voxel key, step length) per step in LDS (512 steps per ray: every ray through a 256^3 grid); a second phase reads the
//     occupancy bytes of all recorded voxels, four per ray and instruction, and adds the lengths up in step order;
//   * the second march (sample placement) finds its occupancy in that record instead of in memory: it visits the same
//     voxels in the same order, so a four-entry window (one per lane of the quad) that moves forward answers every probe
//     whose voxel KEY matches a recorded one -- the answer is the byte of exactly the voxel the reference would read; a voxel
//     that is not in the window (a corner the first walk skipped, a jump past the window) is probed in memory as before;
//   * the voxel key is the packed coordinate triple (10 bits each) instead of the Morton index: spreading the bits is only
//     needed where a byte is read, not on the dependent chain (coordinates >= 1024 keep the reference's multiply form, tagged).
// Used for fast grids (extent 1, power-of-two n) and nr_rays <= MARCH_QUAD_MAX_RAYS; PSDF_MARCH_FORM=thread|quad overrides.
constexpr int QCAP = 512, QSTRIDE = QCAP + 1, QRAYS = 16;    // steps recorded per ray; row stride (bank spread); rays per wave
constexpr int MARCH_QUAD_MAX_RAYS = 6144;    // one workgroup (16 rays, 100 KB of LDS) per CU and round: 4096 rays are one round; measured
                                             // 200 against 378 us at 4096 rays, 391 against 397 at 8192, 751 against 406 at 16 384
constexpr uint32_t KEY_OCC = 0x80000000u, KEY_MASK = 0x7fffffffu;   // a key = the voxel's coordinates, 10 bits each; bit 31 = occupied
template <int J> __device__ __forceinline__ uint32_t qb(uint32_t v) {       // lane J of the quad, to all four
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xf, 0xf, true);
}
template <int J> __device__ __forceinline__ float qbf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {                   // OR over the quad, in all four lanes
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
  return v;
}
// minimum of |x| over the quad.  The operands are magnitudes (sign bit clear), so the order of the floats is the order of
// their bit patterns as unsigned integers and a NaN is larger than every number: the unsigned minimum IS fminf(fminf(|a|, |b|),
// |c|) -- "a NaN is a missing value" included -- and v_min_u32 takes a DPP operand directly (fminf's lowering puts a
// canonicalising v_max in front of every minimum: three instructions instead of one)
__device__ __forceinline__ float quad_min_abs(float x) {
  uint32_t v = __builtin_bit_cast(uint32_t, x) & 0x7fffffffu;
  const uint32_t v1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
  v = v < v1 ? v : v1;
  const uint32_t v2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
  v = v < v2 ? v : v2;
  return __builtin_bit_cast(float, v);
}
struct QAxis {     // what a lane knows of its ray: one axis
  float org, dir, idir, hs, tr;
  int ax;
};
// Key of the voxel of a position (GridFast::pos_to_idx, one axis per lane): the three coordinates side by side -- spreading
// their bits into the Morton index is only needed where a byte is read (key_to_vox), not on the chain of a step.  Inside the grid
// <=> no coordinate has a bit at or above log2 n (`hi`).  umax collects the lane's largest coordinate: beyond 1023 the fields
// overlap and the reference's multiply form of the bit spreading differs from the shift form -- such a ray (it would have to be
// four grid widths outside) is redone by march_ray at the end.
__device__ __forceinline__ uint32_t quad_key(const QAxis& a, const GridFast& g, float pos, uint32_t& umax) {
  float x = pos - a.tr;
  x = (x + 0.5f) * g.nf;
  const uint32_t u = sat_u32(x);
  umax = umax > u ? umax : u;
  return quad_or(u << (10 * a.ax));
}
__device__ __forceinline__ int key_to_vox(uint32_t key) {
  return (int)(expand_bits10_small(key & 1023u) | (expand_bits10_small((key >> 10) & 1023u) << 1) |
               (expand_bits10_small((key >> 20) & 1023u) << 2));
}
__device__ __forceinline__ float quad_dist(const QAxis& a, const GridFast& g, float pos) {   // dist_to_next_voxel
  const float p = g.nf * pos;
  const float tx = (floorf(p + 0.5f + a.hs) - p) * a.idir;
  return fmaxf(quad_min_abs(tx) * g.inv_n, 0.0f);
}
#if defined(PSDF_MARCH_DEBUG)
__device__ unsigned long long g_march_dbg[16];
#define DBG_T(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_march_dbg[i] = __builtin_readcyclecounter();
#else
#define DBG_T(i)
#endif
// PCG32 output function of a state, as the [0, 1) float Pcg::next_float makes of it
__device__ __forceinline__ float pcg_float_of(uint64_t old) {
  const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
  const uint32_t u = (xs >> rot) | (xs << ((~rot + 1u) & 31u));
  return __builtin_bit_cast(float, (u >> 9) | 0x3f800000u) - 1.0f;
}
// Every loop body below is ONE basic block with its exit test and its rare cases at the END: a step is a chain of ~20 dependent
// vector instructions at ~10 cycles each for the single wave of a SIMD, and a branch in the middle of it makes the chains on
// either side run one after the other instead of interleaved (the first build of this kernel, with the break between the voxel
// and the face distance: 450 cycles per step of the first walk).
__global__ void __launch_bounds__(64)
    march_quad_kernel(int nr_rays, GridFast g, Occ o, const float* __restrict__ origins, const float* __restrict__ dirs,
                      const float* __restrict__ t_entry, const float* __restrict__ t_exit_p, float min_dist, int max_per_ray,
                      Pcg rng, int jitter, int* __restrict__ counts, float* __restrict__ spacings, float* __restrict__ ztemp) {
  extern __shared__ uint32_t q_lds[];
  uint32_t* keyA = q_lds;                                                   // [QRAYS][QSTRIDE] voxel keys (+ occupancy bit)
  float* ddA = reinterpret_cast<float*>(q_lds + QRAYS * QSTRIDE);           // [QRAYS][QSTRIDE] step lengths; later the samples
  float* rnA = reinterpret_cast<float*>(q_lds + 2 * QRAYS * QSTRIDE);       // [QRAYS][QSTRIDE] the ray's jitter numbers
  uint32_t* cm_l = q_lds + 3 * QRAYS * QSTRIDE;
  const uint32_t* cm = stage_coarse(o, cm_l);
  const int lane = threadIdx.x, q = lane >> 2, c = lane & 3;
  const int ray_raw = blockIdx.x * QRAYS + q;
  const bool valid = ray_raw < nr_rays;
  const int ray = valid ? ray_raw : nr_rays - 1;           // (spare quads repeat the last ray and write nothing)
  QAxis a;
  a.ax = c < 3 ? c : 2;
  a.org = origins[3 * (int64_t)ray + a.ax];
  a.dir = dirs[3 * (int64_t)ray + a.ax];
  a.idir = fabsf(a.dir) < 1e-16f ? 0.f : (float)(1.0 / (double)a.dir);
  a.hs = 0.5f * sgn(a.dir);
  a.tr = a.ax == 0 ? g.tx : (a.ax == 1 ? g.ty : g.tz);
  const float t_start = t_entry[ray], t_exit = t_exit_p[ray];
  uint32_t* krow = keyA + q * QSTRIDE;
  float* drow = ddA + q * QSTRIDE;
  float* rrow = rnA + q * QSTRIDE;
  // lane 0 records the key, lane 1 the step length; lanes 2 / 3 (and steps that are not taken) write the row's spare word
  uint32_t* rec = c == 0 ? krow : reinterpret_cast<uint32_t*>(drow);
  uint32_t* spare = reinterpret_cast<uint32_t*>(drow) + QCAP;
  uint32_t umax = 0u;
  const uint32_t hi = (1023u & ~(uint32_t)(g.n - 1)) * 0x00100401u;     // bits >= log2 n of the three coordinate fields
  const Pcg rng0 = rng;      // (the stream as the launch received it: march_ray below positions it itself)
  DBG_T(0)
  // ---- first march: occupied length
  float occupied = 0.f;
  float t = t_start;
  int steps = 0;            // all steps of the walk
  int kk = 0;               // steps recorded in the current chunk
  bool walking = true, cached = true;
  uint32_t last_key = 0u;
  while (true) {
    // -- walk up to QCAP steps, recording (voxel, length).  kcap = the steps this ray may still record in the chunk: QCAP, less if
    // the walk's step bound comes first, 0 once the walk has ended (one compare per step instead of three)
    const int steps0 = steps;
    int kcap = walking ? (MAX_DDA_STEPS - steps0 < QCAP ? MAX_DDA_STEPS - steps0 : QCAP) : 0;
    const int kcap0 = kcap;
    kk = 0;
    while (true) {
      const float pos = a.org + t * a.dir;
      const uint32_t key = quad_key(a, g, pos, umax);
      const float d = quad_dist(a, g, pos);
      const bool cont = kk < kcap;
      const bool go = cont && (t < t_exit) && ((key & hi) == 0u);
      kcap = (cont && !go) ? kk : kcap;                    // the walk of this ray ends here
      uint32_t* dst = (go && c < 2) ? rec + kk : spare;
      *dst = c == 0 ? key : __builtin_bit_cast(uint32_t, d);
      t = go ? (t + d) + DDA_EPS : t;
      kk += go ? 1 : 0;
      if (!__any(go)) break;
    }
    steps = steps0 + kk;
    walking = walking && kk == kcap0 && steps < MAX_DDA_STEPS;      // a full chunk: the walk goes on (or ends with an empty one)
    DBG_T(1)
    // -- occupancy of the recorded voxels: lane c of the quad takes entries 4 j + c; the byte loads run PD entries ahead; the
    // lengths of the occupied steps are compacted in place (a compacted slot never lies ahead of the entry being read)
    const int jmax = __builtin_amdgcn_readfirstlane(__reduce_max_sync(~0ull, (kk + 3) >> 2));
    constexpr int PD = 8;
    uint32_t kq[PD];
    float dq[PD];
    uint8_t bq[PD];
#pragma unroll
    for (int s = 0; s < PD; s++) {
      const int k = 4 * s + c;
      const bool act = k < kk;
      kq[s] = krow[act ? k : 0];
      dq[s] = drow[act ? k : 0];
      bq[s] = o.bytes[act ? key_to_vox(kq[s]) : 0];
    }
    int nocc = 0;
    for (int j0 = 0; j0 < jmax; j0 += PD) {
#pragma unroll
      for (int s = 0; s < PD; s++) {
        const int k = 4 * (j0 + s) + c;
        const bool act = k < kk;
        const uint32_t f = (act && bq[s]) ? 1u : 0u;
        const uint32_t f0 = qb<0>(f), f1 = qb<1>(f), f2 = qb<2>(f), f3 = qb<3>(f);
        const int pre = (c > 0 ? f0 : 0u) + (c > 1 ? f1 : 0u) + (c > 2 ? f2 : 0u);
        if (act) krow[k] = kq[s] | (f ? KEY_OCC : 0u);
        if (f) drow[nocc + pre] = dq[s];
        nocc += (int)(f0 + f1 + f2 + f3);
        // refill the slot with the entry PD groups ahead
        const int k2 = k + 4 * PD;
        const bool act2 = k2 < kk;
        kq[s] = krow[act2 ? k2 : 0];
        dq[s] = drow[act2 ? k2 : 0];     // (k2 > every slot written so far: nocc + pre <= k)
        bq[s] = o.bytes[act2 ? key_to_vox(kq[s]) : 0];
      }
    }
    // -- the occupied lengths, added in step order (x + 0 = x: the padding of the last group adds nothing)
    const int imax = __builtin_amdgcn_readfirstlane(__reduce_max_sync(~0ull, (nocc + 3) >> 2));
    for (int i = 0; i < imax; i++) {
      const int e = 4 * i + c;
      const float v = e < nocc ? drow[e] : 0.f;
      occupied += qbf<0>(v);
      occupied += qbf<1>(v);
      occupied += qbf<2>(v);
      occupied += qbf<3>(v);
    }
    if (kk > 0) last_key = krow[kk - 1];
    DBG_T(2)
    if (!__any(walking)) break;
    if (walking) cached = false;          // the record no longer starts at the ray's first step
  }
  // only the LAST step of a walk can end beyond the exit (the walk stops there): the reference's clip of an occupied step
  if (steps > 0 && (last_key & KEY_OCC) && (t - DDA_EPS) > t_exit) occupied -= (t - DDA_EPS) - t_exit;
  const int krec = cached ? kk : 0;       // valid entries of the record for the second march

  int to_create = (int)(occupied / min_dist);
  to_create = clampi(to_create, 0, max_per_ray);
  const float spacing = occupied / to_create;
  int created = 0;
  // ---- second march: sample placement
  if (__any(to_create > 1)) {
    bool alive = to_create > 1;
    t = t_start;
    steps = 0;
    // the ray's jitter numbers: entry i = the i-th next_float() of the ray's stream.  Lane c of the quad makes entries 4 j + c
    // with the generator stepped four at a time (state' = M^4 state + (M^3 + M^2 + M + 1) inc); the second march then
    // reads instead of running the 64-bit generator on its dependent chain.  As many as the first walk took steps (+ 8): an
    // empty step of the second march leaves a voxel as well; a ray that needs more computes them by jumping ahead (exact).
    uint64_t base_state = 0;
    int npre = 0;
    if (jitter) {
      rng.advance((uint64_t)(int64_t)ray);
      base_state = rng.state;
      const uint64_t M = PSDF_PCG_MULT, M2 = M * M, M3 = M2 * M, M4 = M2 * M2, P4 = (M3 + M2 + M + 1ull) * rng.inc;
      uint64_t st = base_state;
      if (c > 0) st = st * M + rng.inc;
      if (c > 1) st = st * M + rng.inc;
      if (c > 2) st = st * M + rng.inc;
      npre = (cached && krec + 8 < QCAP) ? krec + 8 : QCAP;     // (a walk longer than the record: all of them)
      const int gmax = __builtin_amdgcn_readfirstlane(__reduce_max_sync(~0ull, (npre + 3) >> 2));
      for (int j = 0; j < gmax; j++) {
        const int e = 4 * j + c;
        if (e < QCAP) rrow[e] = pcg_float_of(st);
        st = st * M4 + P4;
      }
      npre = (npre + 3) & ~3;
      npre = npre < QCAP ? npre : QCAP;
    }
    auto jitter_at = [&](int i) -> float {           // the i-th number of the ray's stream
      if (i < npre) return rrow[i];
      Pcg r2{base_state, rng.inc};
      r2.advance((uint64_t)i);
      return r2.next_float();
    };
    int ri = 0;
    if (jitter && alive) {
      t = t + spacing * jitter_at(0);
      ri = 1;
    }
    float r_cur = jitter ? jitter_at(ri) : 0.f;
    int ptr = 0;
    uint32_t win = krow[c];                 // window: entry ptr + c
    const int lim = krec - c;                     // entry ptr + c of the window is a recorded one <=> ptr < lim
    const uint32_t cval = 8u | (uint32_t)c;       // "a match, in lane c"
    if (!jitter) r_cur = 0.f;                     // dist + spacing * 0 = dist
    // steps <= iterations: the reference's bound on the steps of a ray cannot bind before the wave has made that many iterations;
    // a wave that gets there (degenerate rays) hands its unfinished rays to march_ray
    int iter = 0;
    bool overrun = false;
    while (true) {
      const bool al = alive && (t < t_exit);
      const float tc = fmaxf(t_start, fminf(t, t_exit));
      const float pos = a.org + tc * a.dir;
      const uint32_t key = quad_key(a, g, pos, umax);
      const float dist = quad_dist(a, g, pos);
      const bool al2 = al && ((key & hi) == 0u);
      // occupancy of the voxel: from the record where it is in the window (entries ptr .. ptr + 3, one per lane; consecutive
      // entries are different voxels, so one lane matches -- should an entry repeat, the lanes' indices OR to a later one and
      // the following lookups go through the rare path below: slower, the same answers)
      const uint32_t x = win ^ key;                                      // 0 or KEY_OCC: this entry is the voxel
      const bool m = ((x << 1) == 0u) && (ptr < lim);
      const uint32_t w = quad_or(m ? (cval | ((win >> 31) << 2)) : 0u);  // bit 3: found, bits 0-1: where, bit 2: occupied
      const bool hit = w >= 8u;
      const bool commit = al2 && hit;
      const bool miss = al2 && !hit;                        // handled at the end of the iteration
      const bool place = commit && (w & 4u) != 0u && created < to_create;
      // move the window to the matching entry (a later sample may sit in the same voxel); the read is used a step later
      ptr += commit ? (int)(w & 3u) : 0;
      win = krow[ptr + c < QCAP ? ptr + c : QCAP - 1];
      drow[created] = tc;                                   // the sample, should one be placed (else overwritten by the next)
      const float delta = dist + spacing * r_cur;
      const float s1 = tc + (place ? spacing : delta);
      const float s2 = s1 + DDA_EPS;
      t = commit ? (place ? s1 : s2) : t;
      created += place ? 1 : 0;
      ri += (commit && !place) ? 1 : 0;
      if (jitter) r_cur = rrow[ri < QCAP ? ri : QCAP - 1];
      alive = al2;
      iter++;
      if (!__any(al2)) break;
      if (__builtin_expect(iter >= MAX_DDA_STEPS, 0)) {
        overrun = al2;
        break;
      }
      if (__builtin_expect(__any(miss || (jitter && al2 && ri >= npre)), 0)) {
        if (miss) {
          // not in the window: look further along the record (the walk only moves forward), at most 8 windows; a voxel the
          // first march did not record at all is read from memory (the window stays: later probes may match again)
          int p2 = ptr;
          uint32_t w2 = 0u, win2 = win;
          for (int it = 0; it < 8 && w2 < 8u; it++) {
            p2 += 4;
            if (p2 >= krec) break;
            win2 = krow[p2 + c < QCAP ? p2 + c : QCAP - 1];
            const bool mm = (((win2 ^ key) << 1) == 0u) && (p2 + c < krec);
            w2 = quad_or(mm ? (cval | ((win2 >> 31) << 2)) : 0u);
          }
          bool occ;
          if (w2 >= 8u) {
            ptr = p2 + (int)(w2 & 3u);
            win = krow[ptr + c < QCAP ? ptr + c : QCAP - 1];
            occ = (w2 & 4u) != 0u;
          } else {
            occ = probe(o, cm, key_to_vox(key));
          }
          // the step itself, as above
          const bool pl = occ && created < to_create;
          const float s1b = tc + (pl ? spacing : delta);
          t = pl ? s1b : s1b + DDA_EPS;
          created += pl ? 1 : 0;
          ri += pl ? 0 : 1;
          if (jitter) r_cur = rrow[ri < QCAP ? ri : QCAP - 1];
        }
        if (jitter && al2 && ri >= npre) r_cur = jitter_at(ri);
      }
    }
    umax = overrun ? 0xffffffffu : umax;          // (hand the ray to march_ray)
    if (created <= 2) created = 0;
  }
  DBG_T(3)
#if defined(PSDF_MARCH_DEBUG)
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_march_dbg[4] = steps; g_march_dbg[5] = kk; g_march_dbg[6] = created; }
#endif
  const bool bail = quad_or(umax) >= 1024u;
  if (!bail && valid) {
    float* __restrict__ zrow = ztemp + (int64_t)ray * max_per_ray;
    for (int i = c; i < created; i += 4) zrow[i] = drow[i];
    if (c == 0) {
      counts[ray] = created;
      spacings[ray] = spacing;
    }
  }
  if (__builtin_expect(__any(bail), 0)) {     // coordinates beyond 1023: the reference's arithmetic, one lane per ray
    if (bail && valid && c == 0)
      march_ray<true, GridFast>(ray, g, o, cm, origins, dirs, t_entry, t_exit_p, min_dist, max_per_ray, rng0, jitter, counts,
                                spacings, ztemp);
  }
}

// Pass 3 (wave per ray): samples from the stored distances, at the ray's exact offset.
__global__ void __launch_bounds__(PSDF_BLOCK)
    march_fill_kernel(int nr_rays, const float* __restrict__ origins, const float* __restrict__ dirs,
                      const float* __restrict__ t_exit_p, int max_per_ray, int max_nr_samples,
                      const int* __restrict__ offsets, const int* __restrict__ counts, const float* __restrict__ spacings,
                      const float* __restrict__ ztemp, float* __restrict__ s_pos, float* __restrict__ s_dirs,
                      float* __restrict__ s_z, float* __restrict__ s_dt, float* __restrict__ ray_fixed_dt,
                      int* __restrict__ start_end) {
  const int ray = blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6);
  if (ray >= nr_rays) return;
  const int lane = threadIdx.x & 63;
  const int cnt = counts[ray], base = offsets[ray];
  const float spacing = spacings[ray];
  if (cnt == 0) {  // empty range at the running offset: the layout the reference has after its compaction pass
    if (lane == 0) {
      start_end[2 * ray] = base;
      start_end[2 * ray + 1] = base;
      ray_fixed_dt[ray] = 0.f;
    }
    return;
  }
  if (lane == 0) {
    start_end[2 * ray] = base;
    start_end[2 * ray + 1] = base + cnt;
    ray_fixed_dt[ray] = spacing;
  }
  if (base + cnt > max_nr_samples) return;  // reservation overflows the pool: the range stays so consumers skip the ray
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const float t_exit = t_exit_p[ray];
  const float* __restrict__ zrow = ztemp + (int64_t)ray * max_per_ray;
  for (int i = lane; i < cnt; i += 64) {
    const float t = zrow[i];
    const int64_t o = base + i;
    st3(s_pos + 3 * o, along(org, t, dir));
    st3(s_dirs + 3 * o, dir);
    s_z[o] = t;
    // the last sample may sit closer than `spacing` to the exit
    s_dt[o] = (i == cnt - 1) ? fmaxf(0.0f, fminf(t_exit - t, spacing)) : spacing;
  }
}

// first sample at the entry of the first occupied voxel (sphere-tracing start), OccupancyGridGPU.cuh:707-814
template <bool WRITE, typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    first_hit_kernel(int nr_rays, G g, Occ o, const float* __restrict__ origins,
                     const float* __restrict__ dirs, const float* __restrict__ t_entry, const float* __restrict__ t_exit_p,
                     int max_nr_samples, const int* __restrict__ offsets, int* __restrict__ counts,
                     float* __restrict__ s_pos, float* __restrict__ s_dirs, float* __restrict__ s_z,
                     float* __restrict__ s_dt, float* __restrict__ ray_fixed_dt, int* __restrict__ start_end) {
  extern __shared__ uint32_t cm_lds[];
  const uint32_t* cm = stage_coarse(o, cm_lds);
  const int ray = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (ray >= nr_rays) return;
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const v3 idir = safe_inverse(dir);
  float t = t_entry[ray];
  const float t_exit = t_exit_p[ray];
  int steps = 0;
  bool hit = false;
  v3 hit_pos = mk3(0, 0, 0);
  float hit_t = 0.f;
  while (t < t_exit && steps < MAX_DDA_STEPS) {
    const v3 pos = along(org, t, dir);
    const int vox = g.pos_to_idx(pos);
    if (!g.in_range(vox)) break;
    const float d = dist_to_next_voxel(pos, dir, idir, g);
    t += d;
    t += DDA_EPS;
    if (probe(o, cm, vox)) {  // the stored z is the t AFTER the step, the position the one before it (as in the reference)
      hit = true;
      hit_pos = pos;
      hit_t = t;
      break;
    }
    // note: the reference does not count steps in this loop; the bound only guards degenerate directions
    steps++;
  }
  if (!WRITE) {
    counts[ray] = hit ? 1 : 0;
    return;
  }
  ray_fixed_dt[ray] = 0.f;
  if (hit) {
    const int o = offsets[ray];
    start_end[2 * ray] = o;
    start_end[2 * ray + 1] = o + 1;
    if (o + 1 <= max_nr_samples) {
      st3(s_pos + 3 * (int64_t)o, hit_pos);
      st3(s_dirs + 3 * (int64_t)o, dir);
      s_z[o] = hit_t;
      s_dt[o] = 0.f;
    }
  } else {
    const int o = offsets[ray];
    start_end[2 * ray] = o;
    start_end[2 * ray + 1] = o;
  }
}

// march a point along its direction to the next occupied voxel (OccupancyGridGPU.cuh:817-895); in place
template <typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    advance_kernel(int count, G g, Occ o, const float* __restrict__ dirs,
                   float* __restrict__ pts, uint8_t* __restrict__ within) {
  extern __shared__ uint32_t cm_lds[];
  const uint32_t* cm = stage_coarse(o, cm_lds);
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const v3 org = ld3(pts + 3 * (int64_t)i), dir = ld3(dirs + 3 * (int64_t)i);
  const v3 idir = safe_inverse(dir);
  float t = 0.f;
  int steps = 0;
  bool inside = true;
  // reference bound: steps < n*sqrt(3) in double (an irrational number: same as the integer bound floor(..)+1)
  const int limit = (int)((double)g.n * sqrt(3.0)) + 1;
  while (inside && steps < limit) {
    const v3 pos = along(org, t, dir);
    const int vox = g.pos_to_idx(pos);
    if (!g.in_range(vox)) {
      inside = false;
      st3(pts + 3 * (int64_t)i, pos);
      break;
    }
    const float d = dist_to_next_voxel(pos, dir, idir, g);
    t += d;
    t += DDA_EPS;
    if (probe(o, cm, vox)) {
      st3(pts + 3 * (int64_t)i, pos);
      break;
    }
    steps++;
  }
  within[i] = inside;
}

// ---------------------------------------------------------------------------------- fixed-shape sphere tracing
// The reference's sphere tracer (permuto_sdf_py/utils/sdf_utils.py:120-218) compacts the unconverged rays with boolean
// masks every iteration (dynamic shapes, implicit host syncs).  These two kernels keep ONE slot per ray for the whole
// trace, so the 15-iteration loop is a fixed sequence of launches that a hipGraph can replay.
// first hit, dense: the per-ray result of compute_first_sample_start_of_occupied_regions without packing
template <typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    first_hit_dense_kernel(int nr_rays, G g, Occ o, const float* __restrict__ origins,
                           const float* __restrict__ dirs, const float* __restrict__ t_entry,
                           const float* __restrict__ t_exit_p, float push, float* __restrict__ pos,
                           uint8_t* __restrict__ converged) {
  extern __shared__ uint32_t cm_lds[];
  const uint32_t* cm = stage_coarse(o, cm_lds);
  const int ray = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (ray >= nr_rays) return;
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const v3 idir = safe_inverse(dir);
  float t = t_entry[ray];
  const float t_exit = t_exit_p[ray];
  int steps = 0;
  bool hit = false;
  v3 hp = org;
  while (t < t_exit && steps < MAX_DDA_STEPS) {
    const v3 pos_ = along(org, t, dir);
    const int vox = g.pos_to_idx(pos_);
    if (!g.in_range(vox)) break;
    const float d = dist_to_next_voxel(pos_, dir, idir, g);
    t += d;
    t += DDA_EPS;
    if (probe(o, cm, vox)) {
      hit = true;
      hp = pos_;
      break;
    }
    steps++;
  }
  // move slightly inside the voxel: pos + dirs*voxel_size*0.5 (sdf_utils.py:133), same operation order
  if (hit) hp = hp + (dir * push) * 0.5f;
  st3(pos + 3 * (int64_t)ray, hp);
  converged[ray] = !hit;  // a ray that never meets an occupied voxel takes no part in the trace
}

// one trace iteration for every ray that has not converged (sdf_utils.py:167-185): step along the ray by
// sdf*multiplier, mark converged when |sdf| < threshold, march to the next occupied voxel, mark converged when the
// march leaves the grid.
template <typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    sphere_trace_step_kernel(int count, G g, Occ o, const float* __restrict__ dirs,
                             const float* __restrict__ sdf, float multiplier, float thresh, float* __restrict__ pts,
                             uint8_t* __restrict__ converged) {
  extern __shared__ uint32_t cm_lds[];
  const uint32_t* cm = stage_coarse(o, cm_lds);
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  if (converged[i]) return;
  const v3 dir = ld3(dirs + 3 * (int64_t)i);
  const float s = sdf[i];
  v3 p = ld3(pts + 3 * (int64_t)i) + (dir * s) * multiplier;
  bool done = fabsf(s) < thresh;
  const v3 idir = safe_inverse(dir);
  float t = 0.f;
  int steps = 0;
  bool inside = true;
  v3 out = p;
  // reference bound: steps < n*sqrt(3) in double (an irrational number: same as the integer bound floor(..)+1)
  const int limit = (int)((double)g.n * sqrt(3.0)) + 1;
  while (inside && steps < limit) {
    const v3 q = along(p, t, dir);
    const int vox = g.pos_to_idx(q);
    if (!g.in_range(vox)) {
      inside = false;
      out = q;
      break;
    }
    const float d = dist_to_next_voxel(q, dir, idir, g);
    t += d;
    t += DDA_EPS;
    if (probe(o, cm, vox)) {
      out = q;
      break;
    }
    steps++;
  }
  st3(pts + 3 * (int64_t)i, out);
  converged[i] = done || !inside;
}

// The same iteration in two launches, for fields where most rays stay on an occupied voxel after their step and only a few
// (silhouette rays leaving the band) march far: the long marches of a few lanes otherwise hold their whole waves (measured
// on a sphere-initialised field: 258 us per iteration, whatever the number of live rays).  Phase A does the step, the
// convergence test and the FIRST probe of the march for every live ray; a ray whose first voxel is empty is appended to a
// list (one atomic per wave).  Phase B marches the listed rays, densely packed, over a fixed grid (count read from device
// memory: the launch sequence stays fixed-shape and graph-capturable).  Per ray the arithmetic is the one of
// sphere_trace_step_kernel, so the end points are the same bit for bit.
template <typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    sphere_trace_step_a_kernel(int count, G g, Occ o, const float* __restrict__ dirs, const float* __restrict__ sdf,
                               float multiplier, float thresh, float* __restrict__ pts, uint8_t* __restrict__ converged,
                               uint8_t* __restrict__ done_flag, int* __restrict__ list, int* __restrict__ list_count) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  bool defer = false;
  if (i < count && !converged[i]) {
    const v3 dir = ld3(dirs + 3 * (int64_t)i);
    const float s = sdf[i];
    const v3 p = ld3(pts + 3 * (int64_t)i) + (dir * s) * multiplier;
    const bool done = fabsf(s) < thresh;
    const v3 q = along(p, 0.f, dir);           // the march's first position, as the one-launch kernel forms it
    const int vox = g.pos_to_idx(q);
    if (!g.in_range(vox)) {
      st3(pts + 3 * (int64_t)i, q);
      converged[i] = true;                     // left the grid
    } else if (o.bytes[vox]) {
      st3(pts + 3 * (int64_t)i, q);
      converged[i] = done;
    } else {                                   // empty voxel: the march continues in phase B, from the stepped point
      st3(pts + 3 * (int64_t)i, p);
      done_flag[i] = done;
      defer = true;
    }
  }
  const unsigned long long m = __ballot(defer);
  if (m) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(list_count, __popcll(m));
    base = __shfl(base, 0, 64);
    if (defer) list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
  }
}

template <typename G>
__global__ void __launch_bounds__(PSDF_BLOCK)
    sphere_trace_step_b_kernel(G g, Occ o, const float* __restrict__ dirs, float* __restrict__ pts,
                               uint8_t* __restrict__ converged, const uint8_t* __restrict__ done_flag,
                               const int* __restrict__ list, const int* __restrict__ list_count) {
  extern __shared__ uint32_t cm_lds[];
  const uint32_t* cm = stage_coarse(o, cm_lds);
  const int n = list_count[0];
  const int limit = (int)((double)g.n * sqrt(3.0)) + 1;
  // A handful of rays, each a chain of a few hundred dependent steps: the launch lasts as long as ONE march, so what counts
  // is the latency of a step.  Empty 8x8x8 blocks are answered by the LDS mask when there is one (these ARE the long marches
  // through empty space).  Walking 2 or 4 steps ahead of the probes measured 1-2 % slower: the arithmetic of a step, not
  // the probe, is the latency that is left.
  for (int k0 = blockIdx.x * PSDF_BLOCK + threadIdx.x; k0 < n; k0 += gridDim.x * PSDF_BLOCK) {
    const int i = list[k0];
    const v3 dir = ld3(dirs + 3 * (int64_t)i);
    const v3 p = ld3(pts + 3 * (int64_t)i);
    const v3 idir = safe_inverse(dir);
    float t = 0.f;
    int steps = 0;
    bool inside = true;
    v3 out = p;
    while (inside && steps < limit) {
      const v3 q = along(p, t, dir);
      const int vox = g.pos_to_idx(q);
      if (!g.in_range(vox)) {
        inside = false;
        out = q;
        break;
      }
      const float d = dist_to_next_voxel(q, dir, idir, g);
      t += d;
      t += DDA_EPS;
      if (probe(o, cm, vox)) {
        out = q;
        break;
      }
      steps++;
    }
    st3(pts + 3 * (int64_t)i, out);
    converged[i] = done_flag[i] || !inside;
  }
}

// ---------------------------------------------------------------------------------- background sampler
// inverse-depth samples outside the bounding sphere, 3-D point (optionally contracted) + 4-D NeRF++ point
// (thread per ray: the reference's loop; launched only for more than PSDF_BLOCK samples per ray)
__global__ void __launch_bounds__(PSDF_BLOCK)
    samples_bg_serial_kernel(int nr_rays, int per_ray, const float* __restrict__ origins, const float* __restrict__ dirs,
                      const float* __restrict__ t_exit_p, float radius, float cx, float cy, float cz, Pcg rng,
                      int randomize, int contract, float* __restrict__ p3, float* __restrict__ p4,
                      float* __restrict__ s_dirs, float* __restrict__ s_z, float* __restrict__ s_dt,
                      float* __restrict__ ray_fixed_dt, int* __restrict__ start_end) {
  const int ray = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (ray >= nr_rays) return;
  const float t_exit = t_exit_p[ray];
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const v3 centre = mk3(cx, cy, cz);
  const float min_t = 1e-3f;
  const float step = (float)((1.0 - (double)min_t) / (per_ray - 1));
  const int64_t base = (int64_t)ray * per_ray;
  float prev_z = 0.f;
  for (int i = 0; i < per_ray; i++) {
    float ts = (float)(1.0 - (double)(i * step));
    if (randomize) {
      rng.advance((uint64_t)(int64_t)(ray * per_ray));
      const float rnd = rng.next_float();
      ts += (float)((double)(step * rnd) - (double)step / 2.0);
    }
    ts = fmaxf(min_t, fminf(ts, 1.0f));
    const float z = t_exit / ts;
    s_z[base + i] = z;
    v3 p = along(org, z, dir);
    if (contract) {
      const float tr = ts * radius;
      const float len = sqrtf(dot3(p, p));
      const v3 u = mk3(p.x / len, p.y / len, p.z / len);
      p = (2 * radius - tr) * u;
    }
    st3(p3 + 3 * (base + i), p);
    const v3 q = p - centre;
    const float inv = rsqrtf(dot3(q, q));
    const float dist = sqrtf(dot3(q, q));
    const v3 u = q * inv;
    float* o4 = p4 + 4 * (base + i);
    o4[0] = u.x;
    o4[1] = u.y;
    o4[2] = u.z;
    o4[3] = radius / fmaxf(1e-6f, dist);
    st3(s_dirs + 3 * (base + i), dir);
    if (i > 0) s_dt[base + i - 1] = z - prev_z;
    prev_z = z;
  }
  s_dt[base + per_ray - 1] = 1e10f;
  ray_fixed_dt[ray] = 0.f;
  start_end[2 * ray] = ray * per_ray;
  start_end[2 * ray + 1] = ray * per_ray + per_ray;
}

// The same samples with ONE THREAD PER SAMPLE.  What made the loop above serial is the generator: every iteration advances it
// by ray * per_ray and draws once, so sample i sees the state (i + 1) * ray * per_ray + i steps from the launch state -- a
// jump the generator can make directly (advance() is exact arithmetic mod 2^64).  The thread-per-ray loop spent its time in
// 32 such jumps per ray (54 us for the ~700 rays of a training step); here a thread makes one, and the spacing to the next
// sample comes from the neighbour through LDS.  A workgroup holds PSDF_BLOCK / per_ray whole rays.
__global__ void __launch_bounds__(PSDF_BLOCK)
    samples_bg_kernel(int nr_rays, int per_ray, const float* __restrict__ origins, const float* __restrict__ dirs,
                      const float* __restrict__ t_exit_p, float radius, float cx, float cy, float cz, Pcg rng,
                      int randomize, int contract, float* __restrict__ p3, float* __restrict__ p4,
                      float* __restrict__ s_dirs, float* __restrict__ s_z, float* __restrict__ s_dt,
                      float* __restrict__ ray_fixed_dt, int* __restrict__ start_end) {
  __shared__ float zs[PSDF_BLOCK + 1];
  const int rpb = PSDF_BLOCK / per_ray;
  const int lr = threadIdx.x / per_ray, i = threadIdx.x - lr * per_ray;
  const int ray = blockIdx.x * rpb + lr;
  const bool live = lr < rpb && ray < nr_rays;
  const float min_t = 1e-3f;
  const float step = (float)((1.0 - (double)min_t) / (per_ray - 1));
  const int64_t base = (int64_t)ray * per_ray;
  float z = 0.f;
  if (live) {
    const float t_exit = t_exit_p[ray];
    const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
    const v3 centre = mk3(cx, cy, cz);
    float ts = (float)(1.0 - (double)(i * step));
    if (randomize) {
      const uint64_t delta = (uint64_t)(int64_t)(ray * per_ray);
      rng.advance((uint64_t)(i + 1) * delta + (uint64_t)i);
      const float rnd = rng.next_float();
      ts += (float)((double)(step * rnd) - (double)step / 2.0);
    }
    ts = fmaxf(min_t, fminf(ts, 1.0f));
    z = t_exit / ts;
    s_z[base + i] = z;
    v3 p = along(org, z, dir);
    if (contract) {
      const float tr = ts * radius;
      const float len = sqrtf(dot3(p, p));
      const v3 u = mk3(p.x / len, p.y / len, p.z / len);
      p = (2 * radius - tr) * u;
    }
    st3(p3 + 3 * (base + i), p);
    const v3 q = p - centre;
    const float inv = rsqrtf(dot3(q, q));
    const float dist = sqrtf(dot3(q, q));
    const v3 u = q * inv;
    float* o4 = p4 + 4 * (base + i);
    o4[0] = u.x;
    o4[1] = u.y;
    o4[2] = u.z;
    o4[3] = radius / fmaxf(1e-6f, dist);
    st3(s_dirs + 3 * (base + i), dir);
    if (i == 0) {
      ray_fixed_dt[ray] = 0.f;
      start_end[2 * ray] = ray * per_ray;
      start_end[2 * ray + 1] = ray * per_ray + per_ray;
    }
  }
  zs[threadIdx.x] = z;
  __syncthreads();
  if (live) s_dt[base + i] = (i == per_ray - 1) ? 1e10f : zs[threadIdx.x + 1] - z;
}

// ---------------------------------------------------------------------------------- sphere
__global__ void __launch_bounds__(PSDF_BLOCK)
    sphere_intersect_kernel(int nr_rays, float radius, float cx, float cy, float cz, const float* __restrict__ origins,
                            const float* __restrict__ dirs, float* __restrict__ p_entry, float* __restrict__ t_entry,
                            float* __restrict__ p_exit, float* __restrict__ t_exit, uint8_t* __restrict__ hit) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= nr_rays) return;
  const v3 o = ld3(origins + 3 * (int64_t)i), d = ld3(dirs + 3 * (int64_t)i);
  const v3 oc = o - mk3(cx, cy, cz);
  const float a = dot3(d, d);
  const float b = (float)(2.0 * (double)dot3(oc, d));
  const float c = dot3(oc, oc) - radius * radius;
  const float disc = b * b - 4 * a * c;
  const float sq = sqrtf(fabsf(disc));
  float t0 = (float)((double)(-b - sq) / (2.0 * (double)a));
  float t1 = (float)((double)(-b + sq) / (2.0 * (double)a));
  const bool miss = disc < 0;
  if (miss) {
    t0 = 0.f;
    t1 = 0.f;
  }
  t0 = fmaxf(0.0f, t0);
  st3(p_entry + 3 * (int64_t)i, along(o, t0, d));
  st3(p_exit + 3 * (int64_t)i, along(o, t1, d));
  t_entry[i] = t0;
  t_exit[i] = t1;
  hit[i] = !miss;
}

// spherical coordinates from three uniform tensors; like the reference (SphereGPU.cuh:121-127) the centre is ignored
__global__ void __launch_bounds__(PSDF_BLOCK)
    sphere_rand_points_kernel(int count, float radius, const float* __restrict__ phi, const float* __restrict__ costheta,
                              const float* __restrict__ u, float* __restrict__ out) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const float theta = acosf(costheta[i]);
  const float r = (float)((double)radius * pow((double)u[i], 1.0 / 3));
  const float st = sinf(theta);
  out[3 * (int64_t)i] = r * st * cosf(phi[i]);
  out[3 * (int64_t)i + 1] = r * st * sinf(phi[i]);
  out[3 * (int64_t)i + 2] = r * cosf(theta);
}

// ---------------------------------------------------------------------------------- packed samples
__global__ void __launch_bounds__(PSDF_BLOCK)
    ray_counts_kernel(int nr_rays, const int* __restrict__ start_end, int* __restrict__ counts) {
  const int r = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (r < nr_rays) counts[r] = start_end[2 * r + 1] - start_end[2 * r];
}
// wave per ray: copy the ray's sample range to its compacted position (13 floats per sample)
__global__ void __launch_bounds__(PSDF_BLOCK)
    compact_copy_kernel(int nr_rays, const int* __restrict__ start_end, const int* __restrict__ offsets,
                        const float* __restrict__ pos, const float* __restrict__ pos4, const float* __restrict__ dirs,
                        const float* __restrict__ z, const float* __restrict__ dt, const float* __restrict__ sdf,
                        const float* __restrict__ fixed_dt, float* __restrict__ o_pos, float* __restrict__ o_pos4,
                        float* __restrict__ o_dirs, float* __restrict__ o_z, float* __restrict__ o_dt,
                        float* __restrict__ o_sdf, float* __restrict__ o_fixed_dt, int* __restrict__ o_start_end) {
  const int lane = lane_id();
  for (int ray = blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6); ray < nr_rays; ray += gridDim.x * (PSDF_BLOCK / 64)) {
    const int s = start_end[2 * ray], n = start_end[2 * ray + 1] - s;
    const int64_t o = offsets[ray];
    for (int j = lane; j < 3 * n; j += 64) {
      o_pos[3 * o + j] = pos[3 * (int64_t)s + j];
      o_dirs[3 * o + j] = dirs[3 * (int64_t)s + j];
    }
    for (int j = lane; j < 4 * n; j += 64) o_pos4[4 * o + j] = pos4[4 * (int64_t)s + j];
    for (int j = lane; j < n; j += 64) {
      o_z[o + j] = z[s + j];
      o_dt[o + j] = dt[s + j];
      o_sdf[o + j] = sdf[s + j];
    }
    if (lane == 0) {
      o_fixed_dt[ray] = fixed_dt[ray];
      o_start_end[2 * ray] = (int)o;
      o_start_end[2 * ray + 1] = (int)o + n;
    }
  }
}
__global__ void __launch_bounds__(PSDF_BLOCK)
    per_sample_ray_idx_kernel(int nr_rays, int nr_samples, const int* __restrict__ start_end, int* __restrict__ out) {
  const int lane = lane_id();
  for (int ray = blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6); ray < nr_rays; ray += gridDim.x * (PSDF_BLOCK / 64)) {
    const int s = start_end[2 * ray], e = start_end[2 * ray + 1];
    for (int i = s + lane; i < e; i += 64)
      if (i < nr_samples) out[i] = ray;
  }
}

// ---------------------------------------------------------------------------------- spherical harmonics
// real SH basis, degree <= 7 (49 channels), polynomial form (PermutoSDFGPU.cuh:275-365)
__global__ void __launch_bounds__(PSDF_BLOCK)
    sh_kernel(int count, int degree, int channels, const float* __restrict__ dirs, float* __restrict__ out) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= count) return;
  const float x = dirs[3 * (int64_t)i], y = dirs[3 * (int64_t)i + 1], z = dirs[3 * (int64_t)i + 2];
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
  const float x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
  float* o = out + (int64_t)i * channels;
  o[0] = 0.28209479177387814f;
  if (degree <= 1) return;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  if (degree <= 2) return;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  if (degree <= 3) return;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  if (degree <= 4) return;
  o[16] = 2.5033429417967046f * xy * (x2 - y2);
  o[17] = 1.7701307697799304f * yz * (-3.0f * x2 + y2);
  o[18] = 0.94617469575756008f * xy * (7.0f * z2 - 1.0f);
  o[19] = 0.66904654355728921f * yz * (3.0f - 7.0f * z2);
  o[20] = -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f;
  o[21] = 0.66904654355728921f * xz * (3.0f - 7.0f * z2);
  o[22] = 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f);
  o[23] = 1.7701307697799304f * xz * (-x2 + 3.0f * y2);
  o[24] = -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4;
  if (degree <= 5) return;
  o[25] = 0.65638205684017015f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4);
  o[26] = 8.3026492595241645f * xy * z * (x2 - y2);
  o[27] = -0.48923829943525038f * y * (3.0f * x2 - y2) * (9.0f * z2 - 1.0f);
  o[28] = 4.7935367849733241f * xy * z * (3.0f * z2 - 1.0f);
  o[29] = 0.45294665119569694f * y * (14.0f * z2 - 21.0f * z4 - 1.0f);
  o[30] = 0.1169503224534236f * z * (-70.0f * z2 + 63.0f * z4 + 15.0f);
  o[31] = 0.45294665119569694f * x * (14.0f * z2 - 21.0f * z4 - 1.0f);
  o[32] = 2.3967683924866621f * z * (x2 - y2) * (3.0f * z2 - 1.0f);
  o[33] = -0.48923829943525038f * x * (x2 - 3.0f * y2) * (9.0f * z2 - 1.0f);
  o[34] = 2.0756623148810411f * z * (-6.0f * x2 * y2 + x4 + y4);
  o[35] = 0.65638205684017015f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4);
  if (degree <= 6) return;
  o[36] = 1.3663682103838286f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
  o[37] = 2.3666191622317521f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4);
  o[38] = 2.0182596029148963f * xy * (x2 - y2) * (11.0f * z2 - 1.0f);
  o[39] = -0.92120525951492349f * yz * (3.0f * x2 - y2) * (11.0f * z2 - 3.0f);
  o[40] = 0.92120525951492349f * xy * (-18.0f * z2 + 33.0f * z4 + 1.0f);
  o[41] = 0.58262136251873131f * yz * (30.0f * z2 - 33.0f * z4 - 5.0f);
  o[42] = 6.6747662381009842f * z2 - 20.024298714302954f * z4 + 14.684485723822165f * z6 - 0.31784601133814211f;
  o[43] = 0.58262136251873131f * xz * (30.0f * z2 - 33.0f * z4 - 5.0f);
  o[44] = 0.46060262975746175f * (x2 - y2) * (11.0f * z2 * (3.0f * z2 - 1.0f) - 7.0f * z2 + 1.0f);
  o[45] = -0.92120525951492349f * xz * (x2 - 3.0f * y2) * (11.0f * z2 - 3.0f);
  o[46] = 0.50456490072872406f * (11.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4);
  o[47] = 2.3666191622317521f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4);
  o[48] = 10.247761577878714f * x2 * y4 - 10.247761577878714f * x4 * y2 + 0.6831841051919143f * x6 - 0.6831841051919143f * y6;
}

// ---------------------------------------------------------------------------------- rays from the image reel
__global__ void __launch_bounds__(PSDF_BLOCK)
    rays_from_reel_kernel(int nr_rays, int H, int W, const float* __restrict__ rgb, const float* __restrict__ mask,
                          const float* __restrict__ K, const float* __restrict__ tf_world_cam,
                          const int* __restrict__ pixel_idx, const int* __restrict__ img_idx, int has_mask,
                          float* __restrict__ origins, float* __restrict__ dirs, float* __restrict__ gt_rgb,
                          float* __restrict__ gt_mask) {
  const int i = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (i >= nr_rays) return;
  const int im = img_idx[i], pix = pixel_idx[i];
  const float px = (float)((double)(float)(pix % W) + 0.5), py = (float)((double)(float)(pix / W) + 0.5);
  const float* Ki = K + 9 * (int64_t)im;
  const float fx = Ki[0], fy = Ki[4], cx = Ki[2], cy = Ki[5];
  const v3 pc = mk3((px - cx) / fx, (py - cy) / fy, 1.0f);
  const float* T = tf_world_cam + 16 * (int64_t)im;  // row major [R|t]
  const v3 t = mk3(T[3], T[7], T[11]);
  // R * pc as the sum of scaled columns, in column order (mat3 * float3 of the reference)
  v3 pw = mk3(T[0] * pc.x + T[1] * pc.y + T[2] * pc.z, T[4] * pc.x + T[5] * pc.y + T[6] * pc.z,
              T[8] * pc.x + T[9] * pc.y + T[10] * pc.z);
  pw = pw + t;
  const v3 d0 = pw - t;
  const v3 d = d0 * rsqrtf(dot3(d0, d0));
  const int x = (int)floorf(px), y = (int)floorf(py);
  const int64_t plane = (int64_t)H * W;
  const float* img = rgb + (int64_t)im * 3 * plane + (int64_t)y * W + x;
  const float m = has_mask ? mask[(int64_t)im * plane + (int64_t)y * W + x] : 1.0f;
  st3(origins + 3 * (int64_t)i, t);
  st3(dirs + 3 * (int64_t)i, d);
  gt_rgb[3 * (int64_t)i] = img[0] * m;
  gt_rgb[3 * (int64_t)i + 1] = img[plane] * m;
  gt_rgb[3 * (int64_t)i + 2] = img[2 * plane] * m;
  gt_mask[i] = m;
}

__global__ void __launch_bounds__(1024)
    scan_i32_kernel(int n, const int* __restrict__ in, int* __restrict__ out, int* __restrict__ total_out) {
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < n) ? in[i] : 0;
    const int incl = wave_incl_scan_add_i(v);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wave_tot[w];
    const int carry = carry_s;
    if (i < n) out[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

inline Grid mk_grid(int n, float extent, const float* tr) {
  const bool pow2 = n > 0 && (n & (n - 1)) == 0;
  return Grid{n, extent, tr[0], tr[1], tr[2], pow2 ? 1.0f / (float)n : 0.f, extent == 1.0f ? 1 : 0};
}
inline bool grid_is_fast(const Grid& g) { return g.unit && g.inv_n != 0.f; }
inline GridFast mk_fast(const Grid& g) { return GridFast{g.n, (float)g.n, g.n * g.n * g.n, g.tx, g.ty, g.tz, g.inv_n}; }
#define GRID1(n) dim3(psdf_blocks((n), PSDF_BLOCK)), dim3(PSDF_BLOCK), 0, st
#define GRID1C(n, oc) dim3(psdf_blocks((n), PSDF_BLOCK)), dim3(PSDF_BLOCK), (oc).coarse ? (size_t)(oc).words * 4 : 0, st
// words of the coarse mask of an n^3 grid (0: grid too small or too large for the LDS copy -> no mask)
inline int coarse_words(int n) {
  const long long v = (long long)n * n * n;
  if (v % (1ll << (COARSE_SHIFT + 5)) != 0) return 0;
  const long long w = v >> (COARSE_SHIFT + 5);
  return (w >= 1 && w <= 16384) ? (int)w : 0;   // <= 64 KiB of LDS (n = 512: 32 KiB)
}
inline Occ mk_occ(int n, const uint8_t* bytes, const uint32_t* coarse) {
  const int w = coarse ? coarse_words(n) : 0;
  return Occ{bytes, w ? coarse : nullptr, w};
}
inline unsigned wave_ray_grid(int nr_rays) {
  unsigned b = psdf_blocks(nr_rays, PSDF_BLOCK / 64);
  return b < 16384u ? (b ? b : 1u) : 16384u;
}

int g_march_form = 0;
}  // namespace

// ================================================================================== C ABI
// grid_translation: HOST pointer to 3 floats.  Bool tensors are 1 byte per element (torch.bool).
extern "C" {

int psdf_grid_points(int count, int nr_voxels_per_dim, float extent, const float* grid_translation,
                     const int* voxel_indices /*NULL = all voxels in Morton order*/, uint64_t rng_state, uint64_t rng_inc,
                     int randomize, float* out_points, void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grid_points_kernel, GRID1(count), count, mk_grid(nr_voxels_per_dim, extent, grid_translation),
                     voxel_indices, Pcg{rng_state, rng_inc}, randomize, out_points);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_grid_update_with_density(int count, const int* voxel_indices, const float* density, float decay, float thresh,
                                  float* grid_values, uint8_t* grid_occupancy, void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(update_density_kernel, GRID1(count), count, voxel_indices, density, decay, thresh, grid_values,
                     grid_occupancy);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// inv_s_tensor (device, 1 float) overrides inv_s when non-NULL.  full_update selects the sdf error range:
// 1.3 half-diagonals for the full update, 1.0 for the random-subset update (OccupancyGridGPU.cuh:437 / :497).
int psdf_grid_update_with_sdf(int count, const int* voxel_indices, const float* sdf, int nr_voxels_per_dim, float extent,
                              const float* grid_translation, float inv_s, const float* inv_s_tensor, int full_update,
                              float thresh, float* grid_values, uint8_t* grid_occupancy, void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(update_sdf_kernel, GRID1(count), count, voxel_indices, sdf,
                     mk_grid(nr_voxels_per_dim, extent, grid_translation), inv_s, inv_s_tensor, full_update, thresh,
                     grid_values, grid_occupancy);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_grid_check_occupancy(int count, int nr_voxels_per_dim, float extent, const float* grid_translation,
                              const uint8_t* grid_occupancy, const float* points, uint8_t* out, void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(check_occupancy_kernel, GRID1(count), count, mk_grid(nr_voxels_per_dim, extent, grid_translation),
                     grid_occupancy, points, out);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// Coarse mask of the occupancy (see struct Occ): psdf_occupancy_coarse_words(n) 32-bit words (0: not available for this
// grid size, pass NULL to the marches).  Rebuild it whenever the occupancy bytes may have changed.
int psdf_occupancy_coarse_words(int nr_voxels_per_dim) { return coarse_words(nr_voxels_per_dim); }
int psdf_occupancy_coarse_mask(int nr_voxels_per_dim, const uint8_t* grid_occupancy, uint32_t* coarse_mask, void* stream) {
  const int words = coarse_words(nr_voxels_per_dim);
  if (words == 0 || !grid_occupancy || !coarse_mask) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(occupancy_coarse_kernel, dim3((words + 3) / 4), dim3(PSDF_BLOCK), 0, st, words, grid_occupancy,
                     coarse_mask);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// 1 = the last psdf_march_samples ran march_kernel (a thread per ray), 2 = march_quad_kernel (four lanes per ray); debug query
int psdf_march_form(void) { return g_march_form; }
#if defined(PSDF_MARCH_DEBUG)
int psdf_march_debug(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_dbg), 16 * 8); }
#endif

// use_grid=1: OccupancyGrid::compute_samples_in_occupied_regions; use_grid=0: RaySampler::compute_samples_fg.
// scratch: nr_rays * (3 + max_nr_samples_per_ray) 4-byte words.  cur_nr_samples (device int) receives the exact total.
int psdf_march_samples(int use_grid, int nr_rays, int nr_voxels_per_dim, float extent, const float* grid_translation,
                       const uint8_t* grid_occupancy, const float* ray_origins, const float* ray_dirs,
                       const float* ray_t_entry, const float* ray_t_exit, float min_dist_between_samples,
                       int max_nr_samples_per_ray, int max_nr_samples, uint64_t rng_state, uint64_t rng_inc, int jitter,
                       float* samples_pos, float* samples_dirs, float* samples_z, float* samples_dt, float* ray_fixed_dt,
                       int* ray_start_end_idx, int* cur_nr_samples, int* scratch, const uint32_t* coarse_mask,
                       void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (max_nr_samples_per_ray < 0 || !scratch) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const Occ oc = mk_occ(nr_voxels_per_dim, grid_occupancy, use_grid ? coarse_mask : nullptr);
  Grid g = use_grid ? mk_grid(nr_voxels_per_dim, extent, grid_translation) : Grid{1, 1.f, 0.f, 0.f, 0.f, 1.f, 1};
  Pcg rng{rng_state, rng_inc};
  int* counts = scratch;
  int* offsets = scratch + nr_rays;
  float* spacings = reinterpret_cast<float*>(scratch + 2 * (int64_t)nr_rays);
  float* ztemp = reinterpret_cast<float*>(scratch + 3 * (int64_t)nr_rays);
#define MARCH(G_, T_, g_)                                                                                                \
  hipLaunchKernelGGL((march_kernel<G_, T_>), GRID1C(nr_rays, oc), nr_rays, g_, oc, ray_origins, ray_dirs, ray_t_entry,    \
                     ray_t_exit, min_dist_between_samples, max_nr_samples_per_ray, rng, jitter, counts, spacings, ztemp)
  const char* form = getenv("PSDF_MARCH_FORM");
  const bool quad = use_grid && grid_is_fast(g) && nr_voxels_per_dim <= 1024 && max_nr_samples_per_ray <= QCAP &&
                    (form ? form[0] == 'q' : nr_rays <= MARCH_QUAD_MAX_RAYS);
  g_march_form = quad ? 2 : 1;
  if (quad) {
    const size_t lds = (size_t)(3 * QRAYS * QSTRIDE + (oc.coarse ? oc.words : 0)) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)march_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(march_quad_kernel, dim3((nr_rays + QRAYS - 1) / QRAYS), dim3(64), lds, st, nr_rays, mk_fast(g), oc,
                       ray_origins, ray_dirs, ray_t_entry, ray_t_exit, min_dist_between_samples, max_nr_samples_per_ray, rng,
                       jitter, counts, spacings, ztemp);
  } else if (use_grid && grid_is_fast(g))
    MARCH(true, GridFast, mk_fast(g));
  else if (use_grid)
    MARCH(true, Grid, g);
  else
    MARCH(false, Grid, g);
#undef MARCH
  hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, nr_rays, counts, offsets, cur_nr_samples);
  hipLaunchKernelGGL(march_fill_kernel, dim3((nr_rays + 3) / 4), dim3(PSDF_BLOCK), 0, st, nr_rays, ray_origins, ray_dirs,
                     ray_t_exit, max_nr_samples_per_ray, max_nr_samples, offsets, counts, spacings, ztemp, samples_pos,
                     samples_dirs, samples_z, samples_dt, ray_fixed_dt, ray_start_end_idx);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_first_hit_samples(int nr_rays, int nr_voxels_per_dim, float extent, const float* grid_translation,
                           const uint8_t* grid_occupancy, const float* ray_origins, const float* ray_dirs,
                           const float* ray_t_entry, const float* ray_t_exit, int max_nr_samples, float* samples_pos,
                           float* samples_dirs, float* samples_z, float* samples_dt, float* ray_fixed_dt,
                           int* ray_start_end_idx, int* cur_nr_samples, int* scratch, const uint32_t* coarse_mask,
                           void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  Grid g = mk_grid(nr_voxels_per_dim, extent, grid_translation);
  const Occ oc = mk_occ(nr_voxels_per_dim, grid_occupancy, coarse_mask);
  int* counts = scratch;
  int* offsets = scratch + nr_rays;
#define FIRST_HIT(W_, T_, g_)                                                                                          \
  hipLaunchKernelGGL((first_hit_kernel<W_, T_>), GRID1C(nr_rays, oc), nr_rays, g_, oc, ray_origins, ray_dirs, ray_t_entry, \
                     ray_t_exit, max_nr_samples, offsets, counts, samples_pos, samples_dirs, samples_z, samples_dt,       \
                     ray_fixed_dt, ray_start_end_idx)
  const bool fast = grid_is_fast(g);
  if (fast)
    FIRST_HIT(false, GridFast, mk_fast(g));
  else
    FIRST_HIT(false, Grid, g);
  hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, nr_rays, counts, offsets, cur_nr_samples);
  if (fast)
    FIRST_HIT(true, GridFast, mk_fast(g));
  else
    FIRST_HIT(true, Grid, g);
#undef FIRST_HIT
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// samples_pos is updated IN PLACE (the reference aliases input and output, src/OccupancyGrid.cu:311)
int psdf_advance_to_next_occupied_voxel(int count, int nr_voxels_per_dim, float extent, const float* grid_translation,
                                        const uint8_t* grid_occupancy, const float* samples_dirs, float* samples_pos,
                                        uint8_t* is_within_bounds, const uint32_t* coarse_mask, void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const Occ oc = mk_occ(nr_voxels_per_dim, grid_occupancy, coarse_mask);
  const Grid g = mk_grid(nr_voxels_per_dim, extent, grid_translation);
  if (grid_is_fast(g))
    hipLaunchKernelGGL(advance_kernel<GridFast>, GRID1C(count, oc), count, mk_fast(g), oc, samples_dirs, samples_pos,
                       is_within_bounds);
  else
    hipLaunchKernelGGL(advance_kernel<Grid>, GRID1C(count, oc), count, g, oc, samples_dirs, samples_pos, is_within_bounds);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// Dense (one slot per ray) first hit for the fixed-shape sphere tracer: pos[ray] = entry point of the first occupied
// voxel pushed half a voxel inside, converged[ray] = 1 for rays that meet no occupied voxel.
int psdf_first_hit_dense(int nr_rays, int nr_voxels_per_dim, float extent, const float* grid_translation,
                         const uint8_t* grid_occupancy, const float* ray_origins, const float* ray_dirs,
                         const float* ray_t_entry, const float* ray_t_exit, float* pos, uint8_t* converged,
                         const uint32_t* coarse_mask, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const float voxel = (float)(1.0 / nr_voxels_per_dim);
  const Occ oc = mk_occ(nr_voxels_per_dim, grid_occupancy, coarse_mask);
  const Grid g = mk_grid(nr_voxels_per_dim, extent, grid_translation);
  if (grid_is_fast(g))
    hipLaunchKernelGGL(first_hit_dense_kernel<GridFast>, GRID1C(nr_rays, oc), nr_rays, mk_fast(g), oc, ray_origins, ray_dirs,
                       ray_t_entry, ray_t_exit, voxel, pos, converged);
  else
    hipLaunchKernelGGL(first_hit_dense_kernel<Grid>, GRID1C(nr_rays, oc), nr_rays, g, oc, ray_origins, ray_dirs, ray_t_entry,
                       ray_t_exit, voxel, pos, converged);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// One iteration of the fixed-shape sphere tracer; sdf[count] is the SDF at pts before the step.
int psdf_sphere_trace_step(int count, int nr_voxels_per_dim, float extent, const float* grid_translation,
                           const uint8_t* grid_occupancy, const float* dirs, const float* sdf, float sdf_multiplier,
                           float sdf_converged_thresh, float* pts, uint8_t* converged, const uint32_t* coarse_mask,
                           void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const Occ oc = mk_occ(nr_voxels_per_dim, grid_occupancy, coarse_mask);
  const Grid g = mk_grid(nr_voxels_per_dim, extent, grid_translation);
  if (grid_is_fast(g))
    hipLaunchKernelGGL(sphere_trace_step_kernel<GridFast>, GRID1C(count, oc), count, mk_fast(g), oc, dirs, sdf, sdf_multiplier,
                       sdf_converged_thresh, pts, converged);
  else
    hipLaunchKernelGGL(sphere_trace_step_kernel<Grid>, GRID1C(count, oc), count, g, oc, dirs, sdf, sdf_multiplier,
                       sdf_converged_thresh, pts, converged);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// psdf_sphere_trace_step as two launches with the long marches compacted (see sphere_trace_step_a_kernel): work = `count`
// bytes + `count` ints + 1 int (the counter, which the CALLER zeroes before the call, e.g. one memset for all the iterations
// of a trace with one counter each).  Same results as psdf_sphere_trace_step.
int psdf_sphere_trace_step_compacted(int count, int nr_voxels_per_dim, float extent, const float* grid_translation,
                                     const uint8_t* grid_occupancy, const float* dirs, const float* sdf, float sdf_multiplier,
                                     float sdf_converged_thresh, float* pts, uint8_t* converged, uint8_t* work_flags,
                                     int* work_list, int* work_count, const uint32_t* coarse_mask, void* stream) {
  if (count <= 0) return PSDF_OK;
  if (!work_flags || !work_list || !work_count) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const Grid g = mk_grid(nr_voxels_per_dim, extent, grid_translation);
  const Occ oc = mk_occ(nr_voxels_per_dim, grid_occupancy, nullptr);
  const Occ ocb = mk_occ(nr_voxels_per_dim, grid_occupancy, coarse_mask);   // the mask serves the long marches only
  const bool fast = grid_is_fast(g);
  if (fast)
    hipLaunchKernelGGL(sphere_trace_step_a_kernel<GridFast>, GRID1(count), count, mk_fast(g), oc, dirs, sdf, sdf_multiplier,
                       sdf_converged_thresh, pts, converged, work_flags, work_list, work_count);
  else
    hipLaunchKernelGGL(sphere_trace_step_a_kernel<Grid>, GRID1(count), count, g, oc, dirs, sdf, sdf_multiplier,
                       sdf_converged_thresh, pts, converged, work_flags, work_list, work_count);
  const int blocks = count < 512 * PSDF_BLOCK ? (count + PSDF_BLOCK - 1) / PSDF_BLOCK : 512;
  // the LDS mask is worth 6 % of the frame here (8.11 -> 7.64 ms on the sphere-initialised field)
  const size_t lds_b = ocb.coarse ? (size_t)ocb.words * 4 : 0;
  if (fast)
    hipLaunchKernelGGL((sphere_trace_step_b_kernel<GridFast>), dim3(blocks), dim3(PSDF_BLOCK), lds_b, st, mk_fast(g), ocb, dirs,
                       pts, converged, work_flags, work_list, work_count);
  else
    hipLaunchKernelGGL((sphere_trace_step_b_kernel<Grid>), dim3(blocks), dim3(PSDF_BLOCK), lds_b, st, g, ocb, dirs, pts,
                       converged, work_flags, work_list, work_count);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_samples_bg(int nr_rays, int nr_samples_per_ray, const float* ray_origins, const float* ray_dirs,
                    const float* ray_t_exit, float sphere_radius, const float* sphere_center /*host, 3 floats*/,
                    uint64_t rng_state, uint64_t rng_inc, int randomize, int contract_3d_samples, float* samples_3d,
                    float* samples_4d, float* samples_dirs, float* samples_z, float* samples_dt, float* ray_fixed_dt,
                    int* ray_start_end_idx, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (nr_samples_per_ray < 2) return PSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (nr_samples_per_ray <= PSDF_BLOCK) {   // thread per sample, whole rays per workgroup
    const int rpb = PSDF_BLOCK / nr_samples_per_ray;
    hipLaunchKernelGGL(samples_bg_kernel, dim3(psdf_blocks(nr_rays, rpb)), dim3(PSDF_BLOCK), 0, st, nr_rays,
                       nr_samples_per_ray, ray_origins, ray_dirs, ray_t_exit, sphere_radius, sphere_center[0],
                       sphere_center[1], sphere_center[2], Pcg{rng_state, rng_inc}, randomize, contract_3d_samples,
                       samples_3d, samples_4d, samples_dirs, samples_z, samples_dt, ray_fixed_dt, ray_start_end_idx);
  } else {
    hipLaunchKernelGGL(samples_bg_serial_kernel, GRID1(nr_rays), nr_rays, nr_samples_per_ray, ray_origins, ray_dirs,
                       ray_t_exit, sphere_radius, sphere_center[0], sphere_center[1], sphere_center[2],
                       Pcg{rng_state, rng_inc}, randomize, contract_3d_samples, samples_3d, samples_4d, samples_dirs,
                       samples_z, samples_dt, ray_fixed_dt, ray_start_end_idx);
  }
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sphere_ray_intersection(int nr_rays, float radius, const float* center /*host*/, const float* ray_origins,
                                 const float* ray_dirs, float* points_entry, float* t_entry, float* points_exit,
                                 float* t_exit, uint8_t* does_intersect, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sphere_intersect_kernel, GRID1(nr_rays), nr_rays, radius, center[0], center[1], center[2],
                     ray_origins, ray_dirs, points_entry, t_entry, points_exit, t_exit, does_intersect);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sphere_rand_points_inside(int count, float radius, const float* phi, const float* costheta, const float* u,
                                   float* points, void* stream) {
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sphere_rand_points_kernel, GRID1(count), count, radius, phi, costheta, u, points);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// generic compaction of a packed container with holes: counts -> scan -> wave-per-ray copy.
// step 1 (this call) computes offsets + total; the caller reads the total (the one host sync the API
// semantics require: the result tensors are exactly sized) and calls psdf_compact_copy.
int psdf_compact_offsets(int nr_rays, const int* ray_start_end_idx, int* scratch /*2*nr_rays*/, int* total,
                         void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ray_counts_kernel, GRID1(nr_rays), nr_rays, ray_start_end_idx, scratch);
  hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, nr_rays, scratch, scratch + nr_rays, total);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}
int psdf_compact_copy(int nr_rays, const int* ray_start_end_idx, const int* offsets, const float* pos, const float* pos4,
                      const float* dirs, const float* z, const float* dt, const float* sdf, const float* fixed_dt,
                      float* o_pos, float* o_pos4, float* o_dirs, float* o_z, float* o_dt, float* o_sdf,
                      float* o_fixed_dt, int* o_start_end, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(compact_copy_kernel, dim3(wave_ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, st, nr_rays,
                     ray_start_end_idx, offsets, pos, pos4, dirs, z, dt, sdf, fixed_dt, o_pos, o_pos4, o_dirs, o_z, o_dt,
                     o_sdf, o_fixed_dt, o_start_end);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_per_sample_ray_idx(int nr_rays, int nr_samples, const int* ray_start_end_idx, int* out, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(per_sample_ray_idx_kernel, dim3(wave_ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, st, nr_rays,
                     nr_samples, ray_start_end_idx, out);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_spherical_harmonics(int count, int degree, const float* dirs, float* out, void* stream) {
  if (degree < 1 || degree > 7) return PSDF_ERR_ARG;
  if (count <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sh_kernel, GRID1(count), count, degree, degree * degree, dirs, out);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_random_rays_from_reel(int nr_rays, int nr_images, int height, int width, const float* rgb_reel,
                               const float* mask_reel, const float* K_reel, const float* tf_world_cam_reel,
                               const int* pixel_indices, const int* img_indices, int has_mask, float* ray_origins,
                               float* ray_dirs, float* gt_rgb, float* gt_mask, void* stream) {
  (void)nr_images;
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rays_from_reel_kernel, GRID1(nr_rays), nr_rays, height, width, rgb_reel, mask_reel, K_reel,
                     tf_world_cam_reel, pixel_indices, img_indices, has_mask, ray_origins, ray_dirs, gt_rgb, gt_mask);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // extern "C"
