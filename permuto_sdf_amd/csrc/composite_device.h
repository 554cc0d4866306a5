// Device helpers shared by the compositing kernels (volume_rendering.hip), the NeuS opacity kernels (neus.hip) and their fusion
// (composite_fused.hip): the ray-range accessor of a packed sample container and the section-point opacity of
// VolumeRenderingNeus.compute_weights (permuto_sdf_py/volume_rendering/volume_rendering_modules.py:129-163).
#pragma once
#include "psdf_common.h"

namespace {
using namespace psdf;

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Section {   // everything the backward needs again
  float tc, pre_a, pre_b, ic, en, ep, pc, nc, p, c, q;
};

__device__ __forceinline__ Section section(float sdf, v3 dir, v3 grad, float dt, float inv_s, float r) {
  Section s;
  s.tc = (dir.x * grad.x + dir.y * grad.y) + dir.z * grad.z;            // (dirs * gradients).sum(-1)
  s.pre_a = -s.tc * 0.5f + 0.5f;
  s.pre_b = -s.tc;
  s.ic = -(fmaxf(s.pre_a, 0.f) * (1.0f - r) + fmaxf(s.pre_b, 0.f) * r); // always non-positive
  const float half = s.ic * dt * 0.5f;
  s.en = sdf + half;
  s.ep = sdf - half;
  s.pc = sigm(s.ep * inv_s);
  s.nc = sigm(s.en * inv_s);
  s.p = s.pc - s.nc;
  s.c = s.pc;
  s.q = (s.p + 1e-5f) / (s.c + 1e-5f);
  return s;
}

struct RayIndex {
  const int* __restrict__ start_end;  // [R,2]
  int equal;                          // rays_have_equal_nr_of_samples
  int fixed;                          // fixed_nr_of_samples_per_ray
  int max_nr_samples;
  __device__ __forceinline__ void get(int ray, int& s, int& e) const {
    if (equal) {
      s = ray * fixed;
      e = s + fixed;
    } else {
      s = start_end[2 * ray];
      e = start_end[2 * ray + 1];
    }
  }
  // the reference skips rays whose reservation overflowed the pool, and empty rays
  __device__ __forceinline__ bool valid(int s, int e) const { return !(e > max_nr_samples || e == s); }
};

#define RAY_LOOP(ray, nr_rays) \
  for (int ray = blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6); ray < nr_rays; ray += gridDim.x * (PSDF_BLOCK / 64))

static inline unsigned ray_grid(int nr_rays) {
  unsigned b = psdf_blocks(nr_rays, PSDF_BLOCK / 64);
  return b < 16384u ? (b ? b : 1u) : 16384u;
}


}  // namespace
