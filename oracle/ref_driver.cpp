// TEST INFRASTRUCTURE.  Runs the REFERENCE's own CUDA kernel headers on the CPU, one "thread" at a time.
// The headers are compiled where they lie (-I /root/reference/kernels); nothing of them is copied here.
// Output: oracle/_ref/libpsdf_ref.so (git-ignored), used to pin oracle/psdf_oracle.c and, on request, as checker.
// Sequential execution makes the reference's atomicAdd slot reservation ray-ordered, i.e. deterministic.
#include "cuda_runtime.h"
thread_local uint3 threadIdx = {0, 0, 0};
thread_local uint3 blockIdx = {0, 0, 0};
thread_local dim3 blockDim;

#include <torch/torch.h>
// Device float->uint32 conversion saturates (negative -> 0); a host conversion wraps.  The reference relies on
// the device behaviour when it passes float voxel coordinates to morton3D(uint32_t...) (OccupancyGridGPU.cuh:190),
// so the float call is routed through a saturating overload; the reference's own uint32 version does the work.
namespace OccupancyGridGPU {
inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z);
static inline uint32_t sat_u32(float f) { return !(f > 0.f) ? 0u : (f >= 4294967296.f ? 0xFFFFFFFFu : (uint32_t)f); }
inline uint32_t morton3D(float x, float y, float z) { return morton3D(sat_u32(x), sat_u32(y), sat_u32(z)); }
inline uint32_t morton3D(int x, int y, int z) { return morton3D((uint32_t)x, (uint32_t)y, (uint32_t)z); }
}  // namespace OccupancyGridGPU
#include "permuto_sdf/OccupancyGridGPU.cuh"
#include "permuto_sdf/RaySamplerGPU.cuh"
#include "permuto_sdf/RaySamplesPackedGPU.cuh"
#include "permuto_sdf/VolumeRenderingGPU.cuh"
#include "permuto_sdf/SphereGPU.cuh"
#include "permuto_sdf/PermutoSDFGPU.cuh"

template <typename T, size_t N>
using Acc = torch::PackedTensorAccessor32<T, N, torch::RestrictPtrTraits>;
typedef Acc<float, 1> F1;
typedef Acc<float, 2> F2;
typedef Acc<float, 3> F3;
typedef Acc<float, 4> F4;
typedef Acc<int, 1> I1;
typedef Acc<int, 2> I2;
typedef Acc<bool, 1> B1;
typedef Acc<bool, 2> B2;

#define FOR_THREADS(n) for (int i__ = 0; i__ < (n); i__++) if ((blockIdx.x = (unsigned)i__, blockDim.x = 1, threadIdx.x = 0, true))
static pcg32 mk_rng(uint64_t state, uint64_t inc) {
  pcg32 r;
  r.state = state;
  r.inc = inc;
  return r;
}

extern "C" {

uint32_t ref_morton3D(uint32_t x, uint32_t y, uint32_t z) { return OccupancyGridGPU::morton3D(x, y, z); }
uint32_t ref_morton3D_invert(uint32_t x) { return OccupancyGridGPU::morton3D_invert(x); }
void ref_pcg32(uint64_t* state, uint64_t* inc, int64_t advance, int n, uint32_t* out_u, float* out_f) {
  pcg32 r = mk_rng(*state, *inc);
  if (advance) r.advance(advance);
  for (int i = 0; i < n; i++) {
    if (out_u) out_u[i] = r.next_uint();
    if (out_f) out_f[i] = r.next_float();
  }
  *state = r.state;
  *inc = r.inc;
}

void ref_compute_grid_points(int count, int n, float extent, float* tr, int* indices, uint64_t st, uint64_t inc,
                             int randomize, float* out) {
  FOR_THREADS(count) {
    if (indices)
      OccupancyGridGPU::compute_random_sample_of_grid_points_gpu(count, n, extent, F1(tr, {3}), I1(indices, {count}),
                                                                 mk_rng(st, inc), randomize, F2(out, {count, 3}));
    else
      OccupancyGridGPU::compute_grid_points_gpu(count, n, extent, F1(tr, {3}), mk_rng(st, inc), randomize,
                                                F2(out, {count, 3}));
  }
}
void ref_update_with_density(int count, int nvox, int* indices, float* density, int n, float decay, float thresh,
                             float* values, bool* occ) {
  FOR_THREADS(count) {
    if (indices)
      OccupancyGridGPU::update_with_density_random_sample_gpu(count, F2(density, {count, 1}), n, I1(indices, {count}),
                                                              decay, thresh, F1(values, {nvox}), B1(occ, {nvox}));
    else
      OccupancyGridGPU::update_with_density_gpu(count, F2(density, {count, 1}), n, decay, thresh, F1(values, {nvox}),
                                                B1(occ, {nvox}));
  }
}
void ref_update_with_sdf(int count, int nvox, int* indices, float* sdf, float extent, int n, float inv_s,
                         float* inv_s_tensor, float thresh, float* values, bool* occ) {
  FOR_THREADS(count) {
    if (indices)
      OccupancyGridGPU::update_with_sdf_random_sample_gpu(count, F2(sdf, {count, 1}), extent, n, I1(indices, {count}),
                                                          F1(inv_s_tensor, {1}), thresh, F1(values, {nvox}),
                                                          B1(occ, {nvox}));
    else
      OccupancyGridGPU::update_with_sdf_gpu(count, F2(sdf, {count, 1}), extent, n, inv_s, 0.f, thresh,
                                            F1(values, {nvox}), B1(occ, {nvox}));
  }
}
void ref_check_occupancy(int count, int n, float extent, float* tr, bool* occ, float* pts, bool* out) {
  const int nvox = n * n * n;
  FOR_THREADS(count) {
    OccupancyGridGPU::check_occupancy_gpu(count, n, extent, F1(tr, {3}), B1(occ, {nvox}), F2(pts, {count, 3}),
                                          B2(out, {count, 1}));
  }
}
void ref_compute_samples_in_occupied_regions(int R, int n, float extent, float* tr, float* o, float* d, float* te,
                                             float* tx, bool* occ, float min_dist, int max_per_ray, int M, uint64_t st,
                                             uint64_t inc, int jitter, float* pos, float* dirs, float* z, float* dt,
                                             float* fixed_dt, int* start_end, int* cur) {
  const int nvox = n * n * n;
  FOR_THREADS(R) {
    OccupancyGridGPU::compute_samples_in_occupied_regions_gpu(
        R, n, extent, F1(tr, {3}), F2(o, {R, 3}), F2(d, {R, 3}), F2(te, {R, 1}), F2(tx, {R, 1}), B1(occ, {nvox}),
        min_dist, max_per_ray, M, mk_rng(st, inc), jitter, F2(pos, {M, 3}), F2(dirs, {M, 3}), F2(z, {M, 1}),
        F2(dt, {M, 1}), F2(fixed_dt, {R, 1}), I2(start_end, {R, 2}), I1(cur, {1}));
  }
}
void ref_compute_first_sample(int R, int n, float extent, float* tr, float* o, float* d, float* te, float* tx,
                              bool* occ, int M, float* pos, float* dirs, float* z, float* dt, float* fixed_dt,
                              int* start_end, int* cur) {
  const int nvox = n * n * n;
  FOR_THREADS(R) {
    OccupancyGridGPU::compute_first_sample_start_of_occupied_regions_gpu(
        R, n, extent, F1(tr, {3}), F2(o, {R, 3}), F2(d, {R, 3}), F2(te, {R, 1}), F2(tx, {R, 1}), B1(occ, {nvox}), M,
        F2(pos, {M, 3}), F2(dirs, {M, 3}), F2(z, {M, 1}), F2(dt, {M, 1}), F2(fixed_dt, {R, 1}), I2(start_end, {R, 2}),
        I1(cur, {1}));
  }
}
void ref_advance_sample(int count, int n, float extent, float* tr, float* dirs, float* pos, bool* occ, float* new_pos,
                        bool* within) {
  const int nvox = n * n * n;
  FOR_THREADS(count) {
    OccupancyGridGPU::advance_sample_to_next_occupied_voxel_gpu(count, n, extent, F1(tr, {3}), F2(dirs, {count, 3}),
                                                                F2(pos, {count, 3}), B1(occ, {nvox}),
                                                                F2(new_pos, {count, 3}), B2(within, {count, 1}));
  }
}
void ref_compact(int R, int M_in, int M_out, float* pos, float* pos4, float* dirs, float* z, float* dt, float* sdf,
                 float* fdt, int* se, float* o_pos, float* o_pos4, float* o_dirs, float* o_z, float* o_dt, float* o_sdf,
                 float* o_fdt, int* o_se, int* o_cur) {
  FOR_THREADS(R) {
    RaySamplesPackedGPU::compact_to_valid_samples_gpu(
        R, F2(pos, {M_in, 3}), F2(pos4, {M_in, 4}), F2(dirs, {M_in, 3}), F2(z, {M_in, 1}), F2(dt, {M_in, 1}),
        F2(sdf, {M_in, 1}), F2(fdt, {R, 1}), I2(se, {R, 2}), F2(o_pos, {M_out, 3}), F2(o_pos4, {M_out, 4}),
        F2(o_dirs, {M_out, 3}), F2(o_z, {M_out, 1}), F2(o_dt, {M_out, 1}), F2(o_sdf, {M_out, 1}), F2(o_fdt, {R, 1}),
        I2(o_se, {R, 2}), I1(o_cur, {1}));
  }
}
void ref_per_sample_ray_idx(int R, int M, int* se, int* out) {
  FOR_THREADS(R) { RaySamplesPackedGPU::compute_per_sample_ray_idx_gpu(R, M, I2(se, {R, 2}), I1(out, {M})); }
}
void ref_samples_bg(int R, int per_ray, float* o, float* d, float* tx, float radius, float* center, uint64_t st,
                    uint64_t inc, int randomize, int contract, float* p3, float* p4, float* dirs, float* z, float* dt,
                    float* fdt, int* se) {
  FOR_THREADS(R) {
    RaySamplerGPU::compute_samples_bg_gpu(R, per_ray, F2(o, {R, 3}), F2(d, {R, 3}), F2(tx, {R, 1}), radius,
                                          F1(center, {3}), mk_rng(st, inc), randomize, contract,
                                          F3(p3, {R, per_ray, 3}), F3(p4, {R, per_ray, 4}), F3(dirs, {R, per_ray, 3}),
                                          F2(z, {R, per_ray}), F2(dt, {R, per_ray}), F2(fdt, {R, 1}), I2(se, {R, 2}));
  }
}
void ref_samples_fg(int R, float* o, float* d, float* te, float* tx, float radius, float* center, float min_dist,
                    int max_per_ray, int M, uint64_t st, uint64_t inc, int jitter, float* pos, float* dirs, float* z,
                    float* dt, float* fdt, int* se, int* cur) {
  FOR_THREADS(R) {
    RaySamplerGPU::compute_samples_fg_gpu(R, F2(o, {R, 3}), F2(d, {R, 3}), F2(te, {R, 1}), F2(tx, {R, 1}), radius,
                                          F1(center, {3}), min_dist, max_per_ray, M, mk_rng(st, inc), jitter,
                                          F2(pos, {M, 3}), F2(dirs, {M, 3}), F2(z, {M, 1}), F2(dt, {M, 1}),
                                          F2(fdt, {R, 1}), I2(se, {R, 2}), I1(cur, {1}));
  }
}
void ref_sphere_intersect(int R, float radius, float* center, float* o, float* d, float* p0, float* t0, float* p1,
                          float* t1, bool* hit) {
  FOR_THREADS(R) {
    ray_intersection_gpu(R, radius, F1(center, {3}), F2(o, {R, 3}), F2(d, {R, 3}), F2(p0, {R, 3}), F2(t0, {R, 1}),
                         F2(p1, {R, 3}), F2(t1, {R, 1}), B2(hit, {R, 1}));
  }
}
void ref_rand_points_inside(int n, float radius, float* center, float* phi, float* ct, float* u, float* out) {
  FOR_THREADS(n) {
    rand_points_inside_gpu(n, radius, F1(center, {3}), F1(phi, {n}), F1(ct, {n}), F1(u, {n}), F2(out, {n, 3}));
  }
}
void ref_spherical_harmonics(int n, int degree, float* dirs, float* out) {
  const int ch = degree * degree;
  FOR_THREADS(n) { spherical_harmonics_gpu(n, degree, F2(dirs, {n, 3}), F2(out, {n, ch})); }
}
void ref_random_rays_from_reel(int R, int I, int H, int W, float* rgb, float* mask, float* K, float* tf, int* pix,
                               int* img, int has_mask, float* o, float* d, float* gt, float* gm) {
  FOR_THREADS(R) {
    random_rays_from_reel_gpu(R, I, H, W, F4(rgb, {I, 3, H, W}), F4(mask, {I, 1, H, W}), F3(K, {I, 3, 3}),
                              F3(tf, {I, 4, 4}), I1(pix, {R}), I1(img, {R}), has_mask, F2(o, {R, 3}), F2(d, {R, 3}),
                              F2(gt, {R, 3}), F2(gm, {R, 1}));
  }
}

// ---- volume rendering (ray index args: R, M, start_end, equal, fixed, max_nr_samples)
#define RI_ARGS int R, int M, int* se, int equal, int fixed, int maxn
void ref_volume_render_nerf(RI_ARGS, float* rgb, float* sigma, float* z, float* dt, float* pred, float* depth, float* bg,
                            float* w) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::volume_render_nerf(R, false, F2(bg, {R, 1}), F2(rgb, {M, 3}), F2(sigma, {M, 1}), maxn,
                                           F2(z, {M, 1}), F2(dt, {M, 1}), I2(se, {R, 2}), equal, fixed, F2(pred, {R, 3}),
                                           F2(depth, {R, 1}), F2(bg, {R, 1}), F2(w, {M, 1}));
  }
}
void ref_volume_render_nerf_backward(RI_ARGS, float* g_pred, float* g_bg, float* g_w, float* pred, float* bg, float* rgb,
                                     float* sigma, float* dt, float* g_rgb, float* g_sigma) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::volume_render_nerf_backward(R, false, F2(g_pred, {R, 3}), F2(g_bg, {R, 1}), F2(g_w, {M, 1}),
                                                    F2(pred, {R, 3}), F2(bg, {R, 1}), F2(bg, {R, 1}), F2(rgb, {M, 3}),
                                                    F2(sigma, {M, 1}), maxn, F2(dt, {M, 1}), I2(se, {R, 2}), equal,
                                                    fixed, F2(g_rgb, {M, 3}), F2(g_sigma, {M, 1}));
  }
}
void ref_compute_dt(RI_ARGS, float* z, float* t_exit, int use_t_exit, float* dt) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::compute_dt_gpu(R, use_t_exit, F2(t_exit, {R, 1}), maxn, F2(z, {M, 1}), I2(se, {R, 2}), equal,
                                       fixed, F2(dt, {M, 1}));
  }
}
void ref_cumprod(RI_ARGS, float* alpha, float* T, float* bg) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::cumprod_alpha2transmittance_gpu(R, maxn, I2(se, {R, 2}), equal, fixed, F2(alpha, {M, 1}),
                                                        F2(T, {M, 1}), F2(bg, {R, 1}));
  }
}
void ref_cumprod_backward(RI_ARGS, float* g_T, float* g_bg, float* alpha, float* T, float* bg, float* cumsumLV,
                          float* g_alpha) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::cumprod_alpha2transmittance_backward_gpu(R, maxn, I2(se, {R, 2}), equal, fixed, F2(g_T, {M, 1}),
                                                                 F2(g_bg, {R, 1}), F2(alpha, {M, 1}), F2(T, {M, 1}),
                                                                 F2(bg, {R, 1}), F2(cumsumLV, {M, 1}),
                                                                 F2(g_alpha, {M, 1}));
  }
}
void ref_integrate(RI_ARGS, float* rgb, float* w, float* pred) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::integrate_with_weights_gpu(R, maxn, I2(se, {R, 2}), equal, fixed, F2(rgb, {M, 3}),
                                                   F2(w, {M, 1}), F2(pred, {R, 3}));
  }
}
void ref_integrate_backward(RI_ARGS, float* g_pred, float* rgb, float* w, float* pred, float* g_rgb, float* g_w) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::integrate_with_weights_backward_gpu(R, maxn, I2(se, {R, 2}), equal, fixed, F2(g_pred, {R, 3}),
                                                            F2(rgb, {M, 3}), F2(w, {M, 1}), F2(pred, {R, 3}),
                                                            F2(g_rgb, {M, 3}), F2(g_w, {M, 1}));
  }
}
void ref_sdf2alpha(RI_ARGS, float* fdt, float* dt, float* sdf, float inv_s, int dynamic, float mult, float* alpha) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::sdf2alpha_gpu(R, maxn, I2(se, {R, 2}), F2(fdt, {R, 1}), F2(dt, {M, 1}), equal, fixed,
                                      F2(sdf, {M, 1}), inv_s, dynamic, mult, F2(alpha, {M, 1}));
  }
}
void ref_sum_over_each_ray(RI_ARGS, int C, float* v, float* s_ray, float* s_smp) {
  FOR_THREADS(R) {
#define SUMK(c) VolumeRenderingGPU::sum_over_each_ray_gpu<c>(R, maxn, I2(se, {R, 2}), equal, fixed, F2(v, {M, C}), F2(s_ray, {R, C}), F2(s_smp, {M, C}))
    if (C == 1) SUMK(1);
    else if (C == 2) SUMK(2);
    else if (C == 3) SUMK(3);
    else if (C == 32) SUMK(32);
#undef SUMK
  }
}
void ref_sum_over_each_ray_backward(RI_ARGS, int C, float* g_ray, float* g_smp, float* v, float* g) {
  FOR_THREADS(R) {
#define SUMB(c) VolumeRenderingGPU::sum_over_each_ray_backward_gpu<c>(R, maxn, I2(se, {R, 2}), equal, fixed, F2(g_ray, {R, C}), F2(g_smp, {M, C}), F2(v, {M, C}), F2(g, {M, C}))
    if (C == 1) SUMB(1);
    else if (C == 2) SUMB(2);
    else if (C == 3) SUMB(3);
#undef SUMB
  }
}
void ref_cumsum(RI_ARGS, float* v, int inverse, float* out) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::cumsum_over_each_ray_gpu(R, maxn, I2(se, {R, 2}), equal, fixed, F2(v, {M, 1}), inverse,
                                                 F2(out, {M, 1}));
  }
}
void ref_compute_cdf(RI_ARGS, float* w, float* cdf) {
  FOR_THREADS(R) {
    VolumeRenderingGPU::compute_cdf_gpu(R, maxn, I2(se, {R, 2}), equal, fixed, F2(w, {M, 1}), F2(cdf, {M, 1}));
  }
}
void ref_importance_sample(RI_ARGS, float* o, float* d, float* fdt, float* z, float* cdf, int nimp, uint64_t st,
                           uint64_t inc, int jitter, float* o_pos, float* o_dirs, float* o_z, int* o_se) {
  const int Mi = R * nimp;
  FOR_THREADS(R) {
    VolumeRenderingGPU::importance_sample_gpu(R, F2(o, {R, 3}), F2(d, {R, 3}), maxn, I2(se, {R, 2}), F2(fdt, {R, 1}),
                                              equal, fixed, F2(z, {M, 1}), F2(cdf, {M, 1}), nimp, mk_rng(st, inc), jitter,
                                              F2(o_pos, {Mi, 3}), F2(o_dirs, {Mi, 3}), F2(o_z, {Mi, 1}), I2(o_se, {R, 2}));
  }
}
void ref_combine(RI_ARGS, float* o, float* d, float* t_exit, float* fdt, float* z, float* sdf, int has_sdf, int nimp,
                 float* imp_z, float* imp_sdf, int Mc, float* c_pos, float* c_dirs, float* c_z, float* c_dt, float* c_sdf,
                 float* c_fdt, int* c_se, int* c_cur) {
  const int Mi = R * nimp;
  FOR_THREADS(R) {
    VolumeRenderingGPU::combine_uniform_samples_with_imp_gpu(
        R, F2(o, {R, 3}), F2(d, {R, 3}), F2(t_exit, {R, 1}), maxn, I2(se, {R, 2}), F2(fdt, {R, 1}), equal, fixed,
        F2(z, {M, 1}), F2(sdf, {M, 1}), has_sdf, Mi, I2(c_se, {R, 2}), F2(c_fdt, {R, 1}), true, nimp, F2(imp_z, {Mi, 1}),
        F2(imp_sdf, {Mi, 1}), has_sdf, Mc, F2(c_pos, {Mc, 3}), F2(c_dirs, {Mc, 3}), F2(c_z, {Mc, 1}), F2(c_dt, {Mc, 1}),
        F2(c_sdf, {Mc, 1}), F2(c_fdt, {R, 1}), I2(c_se, {R, 2}), I1(c_cur, {1}));
  }
}
}  // extern "C"
