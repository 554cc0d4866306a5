// Per-ray compositing primitives for gfx950: forward + backward.
// Replaces src/VolumeRendering.cu + kernels/permuto_sdf/VolumeRenderingGPU.cuh of the reference
// (volume_render_nerf :68/:158, compute_dt :307, cumprod_alpha2transmittance :371/:1135,
//  integrate_with_weights :425/:1208, sdf2alpha :490, sum_over_each_ray :566/:1271,
//  cumsum_over_each_ray :631, compute_cdf :697, importance_sample :793, combine_uniform_samples_with_imp :950).
//
// The reference runs one THREAD per ray with a serial loop over its samples (strided, uncoalesced).  Here a
// WAVE owns a ray: the 64 lanes sweep the ray's contiguous sample range (one 256-B line per load), and the
// per-ray recurrences (transmittance product, sums, cdf) are wave scans / reductions over lanes, with a carry
// across 64-sample chunks.  Sample data is SoA exactly as in RaySamplesPacked ([M,1] / [M,3] fp32).
// The two kernels whose results are index-exact contracts (importance_sample, combine) keep the reference's
// per-ray serial order of operations.
#include "psdf_common.h"
#include "composite_device.h"

using namespace psdf;

namespace {

// ------------------------------------------------------------------ cumprod alpha -> transmittance
__global__ void __launch_bounds__(PSDF_BLOCK)
    cumprod_fwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ alpha, float* __restrict__ trans,
                       float* __restrict__ bg) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const int n = e - s;
    float carry = 1.f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      // the last sample's factor never enters the product (bg transmittance == T of the last sample)
      const float a = (i < n - 1) ? alpha[s + i] : 1.f;
      const float incl = wave_incl_scan_mul(a);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      if (i < n) trans[s + i] = carry * excl;
      carry = carry * __shfl(incl, 63, 64);
    }
    if (lane == 0) bg[ray] = carry;
  }
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    cumprod_bwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ grad_bg, const float* __restrict__ alpha,
                       const float* __restrict__ bg, const float* __restrict__ cumsumLV, float* __restrict__ grad_alpha) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const int n = e - s;
    const float gb = grad_bg[ray] * bg[ray];
    for (int i = lane; i < n; i += 64) {
      float g = 0.f;
      if (i < n - 1) {
        const float a = fmaxf(alpha[s + i], 1e-6f);
        g = cumsumLV[s + i + 1] / a;
        g += gb / a;
      }
      grad_alpha[s + i] = g;
    }
  }
}

// ------------------------------------------------------------------ integrate with weights
__global__ void __launch_bounds__(PSDF_BLOCK)
    integrate_fwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ rgb, const float* __restrict__ w,
                         float* __restrict__ out) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    float r = 0.f, g = 0.f, b = 0.f;
    for (int i = s + lane; i < e; i += 64) {
      const float wi = w[i];
      r += wi * rgb[3 * (int64_t)i];
      g += wi * rgb[3 * (int64_t)i + 1];
      b += wi * rgb[3 * (int64_t)i + 2];
    }
    r = wave_sum(r);
    g = wave_sum(g);
    b = wave_sum(b);
    if (lane == 0) {
      out[3 * ray] = r;
      out[3 * ray + 1] = g;
      out[3 * ray + 2] = b;
    }
  }
}

// compat != 0 reproduces the reference's grad_weights (VolumeRenderingGPU.cuh:1247 reads channel 1 for z)
__global__ void __launch_bounds__(PSDF_BLOCK)
    integrate_bwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ grad_pred, const float* __restrict__ rgb,
                         const float* __restrict__ w, float* __restrict__ grad_rgb, float* __restrict__ grad_w,
                         int compat) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const float gx = grad_pred[3 * ray], gy = grad_pred[3 * ray + 1], gz = grad_pred[3 * ray + 2];
    for (int i = s + lane; i < e; i += 64) {
      const float wi = w[i];
      const float cx = rgb[3 * (int64_t)i], cy = rgb[3 * (int64_t)i + 1];
      const float cz = compat ? cy : rgb[3 * (int64_t)i + 2];
      grad_rgb[3 * (int64_t)i] = gx * wi;
      grad_rgb[3 * (int64_t)i + 1] = gy * wi;
      grad_rgb[3 * (int64_t)i + 2] = gz * wi;
      grad_w[i] = gx * cx + gy * cy + gz * cz;
    }
  }
}

// ------------------------------------------------------------------ sum over each ray (C channels)
__global__ void __launch_bounds__(PSDF_BLOCK)
    sum_ray_fwd_kernel(int nr_rays, RayIndex ri, int C, const float* __restrict__ vals, float* __restrict__ sum_ray,
                       float* __restrict__ sum_sample) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    for (int c = 0; c < C; c++) {
      float a = 0.f;
      for (int i = s + lane; i < e; i += 64) a += vals[(int64_t)i * C + c];
      a = wave_sum(a);
      if (lane == 0) sum_ray[ray * C + c] = a;
      for (int i = s + lane; i < e; i += 64) sum_sample[(int64_t)i * C + c] = a;
    }
  }
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    sum_ray_bwd_kernel(int nr_rays, RayIndex ri, int C, const float* __restrict__ g_ray, const float* __restrict__ g_sample,
                       float* __restrict__ grad) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const int64_t tot = (int64_t)(e - s) * C;
    for (int64_t j = lane; j < tot; j += 64) {
      const int c = (int)(j % C);
      grad[(int64_t)s * C + j] = g_ray[ray * C + c] + g_sample[(int64_t)s * C + j];
    }
  }
}

// ------------------------------------------------------------------ cumsum (optionally from the ray end) / cdf
__global__ void __launch_bounds__(PSDF_BLOCK)
    cumsum_kernel(int nr_rays, RayIndex ri, const float* __restrict__ vals, int inverse, int exclusive,
                  float* __restrict__ out) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const int n = e - s;
    float carry = 0.f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const int idx = inverse ? (e - 1 - i) : (s + i);
      const float v = (i < n) ? vals[idx] : 0.f;
      const float incl = wave_incl_scan_add(v);
      if (i < n) out[idx] = exclusive ? (carry + (incl - v)) : (carry + incl);
      carry = carry + __shfl(incl, 63, 64);
    }
  }
}

// ------------------------------------------------------------------ sdf -> alpha (NeuS mid-point rule)
__device__ __forceinline__ float map_range(float v, float in0, float in1, float out0, float out1) {
  const float c = fmaxf(in0, fminf(in1, v));
  return out0 + ((out1 - out0) / (in1 - in0)) * (c - in0);
}
__device__ __forceinline__ float sigmoidf(float x) { return (float)(1.0 / (1.0 + (double)expf(-x))); }

__global__ void __launch_bounds__(PSDF_BLOCK)
    sdf2alpha_kernel(int nr_rays, RayIndex ri, const float* __restrict__ ray_fixed_dt, const float* __restrict__ dt,
                     const float* __restrict__ sdf, float inv_s_in, int dynamic_inv_s, float inv_s_mult,
                     float* __restrict__ alpha) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    float inv_s = inv_s_in;
    if (dynamic_inv_s) inv_s = map_range(ray_fixed_dt[ray], 0.0001f, 0.01f, 1024.f, 64.f);
    inv_s = inv_s * inv_s_mult;
    const int n = e - s;
    for (int i = lane; i < n - 1; i += 64) {
      const float d = dt[s + i];
      const float prev = sdf[s + i], next = sdf[s + i + 1];
      const float mid = (float)((double)(prev + next) * 0.5);
      float cosv = (next - prev) / fmaxf(d, 1e-6f);
      cosv = clampf(cosv, -1e3f, 0.0f);
      const float half = (float)((double)(cosv * d) * 0.5);
      const float prev_cdf = sigmoidf((mid - half) * inv_s);
      const float next_cdf = sigmoidf((mid + half) * inv_s);
      alpha[s + i] = (float)(((double)(prev_cdf - next_cdf) + 1e-6) / ((double)prev_cdf + 1e-6));
    }
  }
}

// ------------------------------------------------------------------ sdf -> cdf of the importance sampler, one launch
// importance_sampling_sdf_model (sdf_utils.py:383-423) goes from the SDF of a ray's samples to the cdf it draws from in nine
// launches: sdf2alpha, clip(0, 1), 1 - alpha + 1e-7 (two), cumprod_alpha2transmittance, alpha * T, sum_over_each_ray,
// clamp(min = 1e-6), the division, compute_cdf.  The reference's Python has to (it calls the operators one by one); a trainer
// that owns its sampling loop does not.  Same expressions, the same wave scans / reductions in the same chunk order as the
// kernels above: bit-identical to the chain (tests/test_gpu_volume_rendering.py::test_sdf_importance_cdf_equals_the_operator_
// chain).  `cdf` doubles as the scratch of the weights between the two sweeps (a lane re-reads what it wrote itself).
__global__ void __launch_bounds__(PSDF_BLOCK)
    sdf_importance_cdf_kernel(int nr_rays, RayIndex ri, const float* __restrict__ ray_fixed_dt, const float* __restrict__ dt,
                              const float* __restrict__ sdf, float inv_s_in, int dynamic_inv_s, float inv_s_mult,
                              float* __restrict__ cdf) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    float inv_s = inv_s_in;
    if (dynamic_inv_s) inv_s = map_range(ray_fixed_dt[ray], 0.0001f, 0.01f, 1024.f, 64.f);
    inv_s = inv_s * inv_s_mult;
    const int n = e - s;
    float carry = 1.f, acc = 0.f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      float a = 0.f;                                   // the last sample of a ray keeps alpha 0 (sdf2alpha)
      if (i < n - 1) {
        const float d = dt[s + i];
        const float prev = sdf[s + i], next = sdf[s + i + 1];
        const float mid = (float)((double)(prev + next) * 0.5);
        float cosv = (next - prev) / fmaxf(d, 1e-6f);
        cosv = clampf(cosv, -1e3f, 0.0f);
        const float half = (float)((double)(cosv * d) * 0.5);
        const float prev_cdf = sigmoidf((mid - half) * inv_s);
        const float next_cdf = sigmoidf((mid + half) * inv_s);
        a = (float)(((double)(prev_cdf - next_cdf) + 1e-6) / ((double)prev_cdf + 1e-6));
        a = a < 0.f ? 0.f : (a > 1.f ? 1.f : a);       // torch.clip(0, 1): a NaN stays a NaN
      }
      const float om = (1.f - a) + 1e-7f;              // 1 - alpha + 1e-7
      const float f = (i < n - 1) ? om : 1.f;          // cumprod_fwd_kernel: the last sample's factor never enters
      const float incl = wave_incl_scan_mul(f);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      const float T = carry * excl;
      carry = carry * __shfl(incl, 63, 64);
      if (i < n) {
        const float w = a * T;
        cdf[s + i] = w;
        acc += w;                                      // sum_ray_fwd_kernel: per-lane partials over the chunks, then the wave sum
      }
    }
    const float total = wave_sum(acc);
    const float den = total < 1e-6f ? 1e-6f : total;   // torch.clamp(min = 1e-6) (NaN stays NaN)
    float csum = 0.f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const float v = (i < n) ? cdf[s + i] / den : 0.f;
      const float incl = wave_incl_scan_add(v);
      if (i < n) cdf[s + i] = csum + (incl - v);       // cumsum_kernel, exclusive
      csum = csum + __shfl(incl, 63, 64);
    }
  }
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    compute_dt_kernel(int nr_rays, RayIndex ri, const float* __restrict__ z, const float* __restrict__ t_exit,
                      int use_t_exit, float* __restrict__ dt) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const int n = e - s;
    for (int i = lane; i < n; i += 64) {
      const float next = (i < n - 1) ? z[s + i + 1] : (use_t_exit ? t_exit[ray] : 1e10f);
      dt[s + i] = next - z[s + i];
    }
  }
}

// ------------------------------------------------------------------ fused NeRF compositing
// alpha_i = 1 - exp(-sigma_i dt_i); w_i = alpha_i T_i; stop at the first sample whose incoming T < 1e-4.
__global__ void __launch_bounds__(PSDF_BLOCK)
    render_nerf_fwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ rgb, const float* __restrict__ sigma,
                           const float* __restrict__ z, const float* __restrict__ dt, float* __restrict__ pred_rgb,
                           float* __restrict__ pred_depth, float* __restrict__ bg, float* __restrict__ w_out) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) {
      if (lane == 0) {
        pred_rgb[3 * ray] = pred_rgb[3 * ray + 1] = pred_rgb[3 * ray + 2] = 0.f;
        pred_depth[ray] = 0.f;
        bg[ray] = 1.f;
      }
      continue;
    }
    const int n = e - s;
    float T = 1.f, r = 0.f, g = 0.f, b = 0.f, dep = 0.f;
    bool done = false;
    for (int base = 0; base < n && !done; base += 64) {
      const int i = base + lane;
      const bool act = i < n;
      const float a = act ? (1.f - __expf(-sigma[s + i] * dt[s + i])) : 0.f;
      const float om = 1.f - a;
      const float incl = wave_incl_scan_mul(om);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      const float Ti = T * excl;  // transmittance reaching sample i
      const unsigned long long dead = __ballot(act && Ti < 1e-4f);
      const int first_dead = dead ? (int)__ffsll((long long)dead) - 1 : 64;
      const bool use = act && lane < first_dead;
      const float wi = use ? a * Ti : 0.f;
      if (use) {
        w_out[s + i] = wi;
        r += wi * rgb[3 * (int64_t)(s + i)];
        g += wi * rgb[3 * (int64_t)(s + i) + 1];
        b += wi * rgb[3 * (int64_t)(s + i) + 2];
        dep += wi * z[s + i];
      }
      if (dead) {
        // T after the last processed sample = T reaching the first dead one
        T = __shfl(Ti, first_dead, 64);
        done = true;
      } else {
        T = T * __shfl(incl, 63, 64);
      }
    }
    r = wave_sum(r);
    g = wave_sum(g);
    b = wave_sum(b);
    dep = wave_sum(dep);
    if (lane == 0) {
      pred_rgb[3 * ray] = r;
      pred_rgb[3 * ray + 1] = g;
      pred_rgb[3 * ray + 2] = b;
      pred_depth[ray] = dep;
      bg[ray] = T;
    }
  }
}

// suffix trick of the reference (:262-287): grad_sigma_i = dt_i * ( g . (T_{i+1} c_i - suffix_i) - g_bg * T_final )
__global__ void __launch_bounds__(PSDF_BLOCK)
    render_nerf_bwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ grad_pred, const float* __restrict__ grad_bg,
                           const float* __restrict__ pred_rgb, const float* __restrict__ bg,
                           const float* __restrict__ rgb, const float* __restrict__ sigma, const float* __restrict__ dt,
                           float* __restrict__ grad_rgb, float* __restrict__ grad_sigma) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;
    const int n = e - s;
    const float gx = grad_pred[3 * ray], gy = grad_pred[3 * ray + 1], gz = grad_pred[3 * ray + 2];
    const float fx = pred_rgb[3 * ray], fy = pred_rgb[3 * ray + 1], fz = pred_rgb[3 * ray + 2];
    const float gbg = grad_bg[ray], lastT = bg[ray];
    float T = 1.f, ux = 0.f, uy = 0.f, uz = 0.f;  // colour integrated up to (and including) the previous chunk
    bool done = false;
    for (int base = 0; base < n && !done; base += 64) {
      const int i = base + lane;
      const bool act = i < n;
      const float d = act ? dt[s + i] : 0.f;
      const float a = act ? (1.f - __expf(-sigma[s + i] * d)) : 0.f;
      const float om = 1.f - a;
      const float incl = wave_incl_scan_mul(om);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      const float Ti = T * excl;
      const unsigned long long dead = __ballot(act && Ti < 1e-4f);
      const int first_dead = dead ? (int)__ffsll((long long)dead) - 1 : 64;
      const bool use = act && lane < first_dead;
      const float wi = use ? a * Ti : 0.f;
      float cx = 0.f, cy = 0.f, cz = 0.f;
      if (use) {
        cx = rgb[3 * (int64_t)(s + i)];
        cy = rgb[3 * (int64_t)(s + i) + 1];
        cz = rgb[3 * (int64_t)(s + i) + 2];
      }
      const float px = ux + wave_incl_scan_add(wi * cx);  // colour up to and including sample i
      const float py = uy + wave_incl_scan_add(wi * cy);
      const float pz = uz + wave_incl_scan_add(wi * cz);
      if (use) {
        grad_rgb[3 * (int64_t)(s + i)] = gx * wi;
        grad_rgb[3 * (int64_t)(s + i) + 1] = gy * wi;
        grad_rgb[3 * (int64_t)(s + i) + 2] = gz * wi;
        const float Tn = Ti * om;  // T after this sample
        float gr = gx * d * (Tn * cx - (fx - px));
        gr += gy * d * (Tn * cy - (fy - py));
        gr += gz * d * (Tn * cz - (fz - pz));
        gr += gbg * (-d * lastT);
        grad_sigma[s + i] = gr;
      }
      ux = __shfl(px, 63, 64);
      uy = __shfl(py, 63, 64);
      uz = __shfl(pz, 63, 64);
      if (dead)
        done = true;
      else
        T = T * __shfl(incl, 63, 64);
    }
  }
}

// ------------------------------------------------------------------ importance sampling (index-exact)
// One thread per (ray, importance sample).  The reference thread draws, for sample i of ray `idx`,
// after advancing its by-value generator `idx` steps before EVERY draw (VolumeRenderingGPU.cuh:872-874),
// i.e. draw i sits at stream position i*(idx+1) + idx.
__device__ __forceinline__ int cdf_search(const float* __restrict__ cdf, float val, int imin, int imax) {
  if (imax - imin < 1) return imax;  // single-sample ray: the reference's loop would not terminate
  while (imax >= imin) {
    const int imid = imin + (imax - imin) / 2;
    if (cdf[imid] > val)
      imax = imid;
    else
      imin = imid;
    if ((imax - imin) == 1) return imax;
  }
  return imax;
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    importance_sample_kernel(int nr_rays, RayIndex ri, const float* __restrict__ origins, const float* __restrict__ dirs,
                             const float* __restrict__ ray_fixed_dt, const float* __restrict__ z,
                             const float* __restrict__ cdf, int nr_imp, Pcg rng, int jitter, float* __restrict__ out_pos,
                             float* __restrict__ out_dirs, float* __restrict__ out_z) {
  const int64_t gid = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (gid >= (int64_t)nr_rays * nr_imp) return;
  const int ray = (int)(gid / nr_imp), i = (int)(gid % nr_imp);
  int s, e;
  ri.get(ray, s, e);
  const int64_t o = (int64_t)ray * nr_imp + i;
  if (!ri.valid(s, e)) {
    out_pos[3 * o] = out_pos[3 * o + 1] = out_pos[3 * o + 2] = 0.f;
    out_dirs[3 * o] = out_dirs[3 * o + 1] = out_dirs[3 * o + 2] = 0.f;
    out_z[o] = -1.f;
    return;
  }
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const float fixed_dt = ray_fixed_dt[ray];
  const float step = (float)(1.0 / (nr_imp + 1));
  float u = step + i * step;
  if (jitter) {
    rng.advance((uint64_t)i * ((uint64_t)ray + 1) + (uint64_t)ray);
    const float rnd = rng.next_float();
    const float mov = (float)((double)step / 2.0);
    u += map_range(rnd, 0.0f, 1.0f, -mov, +mov);
  }
  u = clampf(u, (float)(0.0 + 1e-6), (float)(1.0 - 1e-5));
  const int imax = cdf_search(cdf, u, s, e - 1);
  const int imin = (imax - 1 > 0) ? imax - 1 : 0;
  const float cdf_max = cdf[imax], cdf_min = cdf[imin];
  const float z_max = z[imax], z_min = z[imin];
  float z_imp = map_range(u, cdf_min, cdf_max, z_min, z_max);
  float d_min = z_imp - z_min, d_max = z_max - z_imp;
  if (d_min < d_max) {
    d_min = fminf(d_min, fixed_dt);
    z_imp = z_min + d_min;
  } else {
    d_max = fminf(d_max, fixed_dt);
    z_imp = z_max - d_max;
  }
  st3(out_pos + 3 * o, along(org, z_imp, dir));
  st3(out_dirs + 3 * o, dir);
  out_z[o] = z_imp;
}

// ------------------------------------------------------------------ merge uniform + importance samples
// pass 0: per-ray output count (uniform + importance, or 0 when the ray has <= 1 uniform samples)
__global__ void __launch_bounds__(PSDF_BLOCK)
    combine_count_kernel(int nr_rays, RayIndex uni, int nr_imp, int* __restrict__ counts) {
  const int ray = blockIdx.x * PSDF_BLOCK + threadIdx.x;
  if (ray >= nr_rays) return;
  int s, e;
  uni.get(ray, s, e);
  const int n = e - s;
  counts[ray] = (n <= 1) ? 0 : n + nr_imp;
}

// pass 1 (after an exclusive scan of the counts): 2-way merge per ray, ONE WAVE per ray.
// The reference merges serially (thread per ray: take the smaller head, the importance sample on ties); its result for SORTED
// inputs -- which is what the samplers produce: z grows along a ray in both sets -- is the rank merge computed here in
// parallel: uniform sample j lands at j + #{importance z <= z_j}, importance sample i at i + #{uniform z < z_i}; the dt of an
// element needs the z of its successor, which is the smaller of the next element of its own list and the first element of
// the other list that sorts after it.  Same values, same order, bit for bit; every lane places its own elements (binary
// search over <= 16 / <= ~100 cached floats) instead of one lane walking a chain of ~100 dependent loads (76 us per call
// for the ~700 rays of a training step, 2 calls per step: now a few us).  Inputs that are NOT sorted (never produced by this
// library, but the operator accepts any tensors) take the serial walk of the reference on lane 0.
__global__ void __launch_bounds__(PSDF_BLOCK)
    combine_fill_kernel(int nr_rays, RayIndex uni, const float* __restrict__ origins, const float* __restrict__ dirs,
                        const float* __restrict__ t_exit, const float* __restrict__ uni_fixed_dt,
                        const float* __restrict__ uni_z, const float* __restrict__ uni_sdf, int has_sdf, int nr_imp,
                        const float* __restrict__ imp_z, const float* __restrict__ imp_sdf,
                        const int* __restrict__ offsets, int out_max, float* __restrict__ out_pos,
                        float* __restrict__ out_dirs, float* __restrict__ out_z, float* __restrict__ out_dt,
                        float* __restrict__ out_sdf, float* __restrict__ out_fixed_dt, int* __restrict__ out_start_end) {
  const int ray = blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6);
  if (ray >= nr_rays) return;
  const int lane = threadIdx.x & 63;
  int us, ue;
  uni.get(ray, us, ue);
  const int un = ue - us;
  const int base = offsets[ray];
  if (un <= 1) {  // empty range at the running offset (the reference's layout after compaction)
    if (lane == 0) {
      out_fixed_dt[ray] = 0.f;
      out_start_end[2 * ray] = base;
      out_start_end[2 * ray + 1] = base;
    }
    return;
  }
  const int total = un + nr_imp;
  const float fixed_dt = uni_fixed_dt[ray];
  if (lane == 0) {
    out_start_end[2 * ray] = base;
    out_start_end[2 * ray + 1] = base + total;
  }
  if (base + total > out_max) return;
  if (lane == 0) out_fixed_dt[ray] = fixed_dt;
  const v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
  const int is = ray * nr_imp;
  const float* __restrict__ zu_p = uni_z + us;
  const float* __restrict__ zi_p = imp_z + is;
  bool sorted = true;   // (a NaN fails the comparison and sends the ray down the serial walk)
  for (int j = lane; j + 1 < un; j += 64) sorted = sorted && (zu_p[j] <= zu_p[j + 1]);
  for (int i = lane; i + 1 < nr_imp; i += 64) sorted = sorted && (zi_p[i] <= zi_p[i + 1]);
  const float end_z = 1e10f;   // the reference's sentinel of an exhausted list
  if (__ballot(!sorted) != 0ull) {
    if (lane != 0) return;
    int cu = 0, ci = 0;
    float prev_z = 0.f;
    for (int i = 0; i < total; i++) {
      const float zu = (cu < un) ? zu_p[cu] : end_z;
      const float zi = (ci < nr_imp) ? zi_p[ci] : end_z;
      const bool take_u = zu < zi;
      const float zz = take_u ? zu : zi;
      const int64_t o = base + i;
      st3(out_pos + 3 * o, along(org, zz, dir));
      st3(out_dirs + 3 * o, dir);
      out_z[o] = zz;
      if (has_sdf) out_sdf[o] = take_u ? uni_sdf[us + cu] : imp_sdf[is + ci];
      if (i > 0) out_dt[o - 1] = fminf(zz - prev_z, fixed_dt);
      prev_z = zz;
      if (take_u)
        cu++;
      else
        ci++;
    }
    out_dt[base + total - 1] = clampf(t_exit[ray] - prev_z, 0.0f, fixed_dt);
    return;
  }
  const float tx = t_exit[ray];
  auto emit = [&](int r, float zz, float sdf, float next_z) {
    const int64_t o = base + r;
    st3(out_pos + 3 * o, along(org, zz, dir));
    st3(out_dirs + 3 * o, dir);
    out_z[o] = zz;
    if (has_sdf) out_sdf[o] = sdf;
    out_dt[o] = (r == total - 1) ? clampf(tx - zz, 0.0f, fixed_dt) : fminf(next_z - zz, fixed_dt);
  };
  for (int j = lane; j < un; j += 64) {
    const float zz = zu_p[j];
    int lo = 0, hi = nr_imp;          // lo = #{importance z <= zz}: they come first (ties go to the importance sample)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (zi_p[mid] <= zz)
        lo = mid + 1;
      else
        hi = mid;
    }
    const float nu = (j + 1 < un) ? zu_p[j + 1] : end_z, ni = (lo < nr_imp) ? zi_p[lo] : end_z;
    emit(j + lo, zz, has_sdf ? uni_sdf[us + j] : 0.f, nu < ni ? nu : ni);
  }
  for (int i = lane; i < nr_imp; i += 64) {
    const float zz = zi_p[i];
    int lo = 0, hi = un;              // lo = #{uniform z < zz}
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (zu_p[mid] < zz)
        lo = mid + 1;
      else
        hi = mid;
    }
    const float ni = (i + 1 < nr_imp) ? zi_p[i + 1] : end_z, nu = (lo < un) ? zu_p[lo] : end_z;
    emit(i + lo, zz, has_sdf ? imp_sdf[is + i] : 0.f, nu < ni ? nu : ni);
  }
}

// exclusive scan of per-ray counts -> offsets, total in *total_out (single workgroup; R is at most a few 1e5)
__global__ void __launch_bounds__(1024)
    exclusive_scan_kernel(int n, const int* __restrict__ in, int* __restrict__ out, int* __restrict__ total_out) {
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

inline RayIndex mk_ri(const int* se, int equal, int fixed, int maxn) { return RayIndex{se, equal, fixed, maxn}; }

}  // namespace

// ================================================================================== C ABI
// Common ray-index arguments: start_end [R,2] int32 (may be NULL when equal!=0), equal, fixed, max_nr_samples.
extern "C" {

int psdf_exclusive_scan_i32(int n, const int* in, int* out, int* total, void* stream) {
  if (n < 0) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, in, out, total);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_cumprod_alpha2transmittance(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                                     const float* alpha, float* transmittance, float* bg_transmittance, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(cumprod_fwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), alpha, transmittance, bg_transmittance);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_cumprod_alpha2transmittance_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                                              const float* grad_bg, const float* alpha, const float* bg,
                                              const float* cumsumLV, float* grad_alpha, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(cumprod_bwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), grad_bg, alpha, bg, cumsumLV, grad_alpha);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_integrate_with_weights(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                                const float* rgb, const float* weights, float* pred, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(integrate_fwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), rgb, weights, pred);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_integrate_with_weights_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                                         const float* grad_pred, const float* rgb, const float* weights, float* grad_rgb,
                                         float* grad_weights, int reference_compat, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(integrate_bwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), grad_pred, rgb, weights, grad_rgb, grad_weights,
                     reference_compat);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sum_over_each_ray(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int channels,
                           const float* values, float* sum_per_ray, float* sum_per_sample, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (channels <= 0) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(sum_ray_fwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), channels, values, sum_per_ray, sum_per_sample);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sum_over_each_ray_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                                    int channels, const float* grad_per_ray, const float* grad_per_sample,
                                    float* grad_values, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(sum_ray_bwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), channels, grad_per_ray, grad_per_sample, grad_values);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// exclusive == 0: inclusive cumsum (cumsum_over_each_ray); exclusive != 0: compute_cdf
int psdf_cumsum_over_each_ray(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                              const float* values, int inverse, int exclusive, float* out, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(cumsum_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), values, inverse, exclusive, out);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sdf2alpha(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* ray_fixed_dt,
                   const float* samples_dt, const float* sdf, float inv_s, int dynamic_inv_s, float inv_s_multiplier,
                   float* alpha, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(sdf2alpha_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), ray_fixed_dt, samples_dt, sdf, inv_s, dynamic_inv_s,
                     inv_s_multiplier, alpha);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sdf_importance_cdf(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* ray_fixed_dt,
                            const float* samples_dt, const float* sdf, float inv_s, int dynamic_inv_s, float inv_s_multiplier,
                            float* cdf, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(sdf_importance_cdf_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), ray_fixed_dt, samples_dt, sdf, inv_s, dynamic_inv_s,
                     inv_s_multiplier, cdf);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_compute_dt(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* samples_z,
                    const float* ray_t_exit, int use_ray_t_exit, float* dt, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(compute_dt_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), samples_z, ray_t_exit, use_ray_t_exit, dt);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_volume_render_nerf(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                            const float* rgb, const float* density, const float* samples_z, const float* samples_dt,
                            float* pred_rgb, float* pred_depth, float* bg_transmittance, float* weight_per_sample,
                            void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(render_nerf_fwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), rgb, density, samples_z, samples_dt, pred_rgb,
                     pred_depth, bg_transmittance, weight_per_sample);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_volume_render_nerf_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                                     const float* grad_pred_rgb, const float* grad_bg_transmittance,
                                     const float* pred_rgb, const float* bg_transmittance, const float* rgb,
                                     const float* density, const float* samples_dt, float* grad_rgb, float* grad_density,
                                     void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipLaunchKernelGGL(render_nerf_bwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     mk_ri(start_end, equal, fixed, max_nr_samples), grad_pred_rgb, grad_bg_transmittance, pred_rgb,
                     bg_transmittance, rgb, density, samples_dt, grad_rgb, grad_density);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// rng_state / rng_inc: the caller-owned PCG32 generator, passed by value as the reference does.
int psdf_importance_sample(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
                           const float* ray_origins, const float* ray_dirs, const float* ray_fixed_dt,
                           const float* samples_z, const float* cdf, int nr_importance_samples, uint64_t rng_state,
                           uint64_t rng_inc, int jitter, float* out_pos, float* out_dirs, float* out_z, void* stream) {
  if (nr_rays <= 0 || nr_importance_samples <= 0) return PSDF_OK;
  Pcg rng{rng_state, rng_inc};
  const int64_t tot = (int64_t)nr_rays * nr_importance_samples;
  hipLaunchKernelGGL(importance_sample_kernel, dim3(psdf_blocks(tot, PSDF_BLOCK)), dim3(PSDF_BLOCK), 0,
                     (hipStream_t)stream, nr_rays, mk_ri(start_end, equal, fixed, max_nr_samples), ray_origins, ray_dirs,
                     ray_fixed_dt, samples_z, cdf, nr_importance_samples, rng, jitter, out_pos, out_dirs, out_z);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// Deterministic, ray-ordered replacement of the reference's atomicAdd slot reservation:
// counts -> exclusive scan (offsets, total in cur_nr_samples) -> fill.  `scratch` holds 2*nr_rays ints.
int psdf_combine_uniform_samples_with_imp(int nr_rays, const int* uni_start_end, int uni_equal, int uni_fixed,
                                          int uni_max_nr_samples, const float* ray_origins, const float* ray_dirs,
                                          const float* ray_t_exit, const float* uni_fixed_dt, const float* uni_z,
                                          const float* uni_sdf, int has_sdf, int nr_imp, const float* imp_z,
                                          const float* imp_sdf, int out_max_nr_samples, float* out_pos, float* out_dirs,
                                          float* out_z, float* out_dt, float* out_sdf, float* out_fixed_dt,
                                          int* out_start_end, int* out_cur_nr_samples, int* scratch, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  RayIndex uni = mk_ri(uni_start_end, uni_equal, uni_fixed, uni_max_nr_samples);
  int* counts = scratch;
  int* offsets = scratch + nr_rays;
  hipLaunchKernelGGL(combine_count_kernel, dim3(psdf_blocks(nr_rays, PSDF_BLOCK)), dim3(PSDF_BLOCK), 0, st, nr_rays, uni,
                     nr_imp, counts);
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, nr_rays, counts, offsets, out_cur_nr_samples);
  hipLaunchKernelGGL(combine_fill_kernel, dim3(psdf_blocks(nr_rays, PSDF_BLOCK / 64)), dim3(PSDF_BLOCK), 0, st, nr_rays, uni,
                     ray_origins, ray_dirs, ray_t_exit, uni_fixed_dt, uni_z, uni_sdf, has_sdf, nr_imp, imp_z, imp_sdf,
                     offsets, out_max_nr_samples, out_pos, out_dirs, out_z, out_dt, out_sdf, out_fixed_dt, out_start_end);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // extern "C"
