// NeuS rendering of a packed sample container in ONE launch per direction: section-point opacity -> transmittance ->
// weights -> radiance, and its whole backward.  What the reference strings together per ray batch
//   VolumeRenderingNeus.compute_weights + integrate   (permuto_sdf_py/volume_rendering/volume_rendering_modules.py:129-190)
// through four autograd Functions (volume_rendering_funcs.py:55-224: cumprod_alpha2transmittance, integrate_with_weights,
// sum_over_each_ray, and the ~30 torch launches of the opacity) and what hotpath.py / train_step.py used to string together
// from this library's own per-operator kernels: forward 4 launches + 1 torch multiply, backward 5 launches + 4 torch
// elementwise launches + 3 zero fills.  The per-operator entry points stay (they ARE the drop-in API, include/psdf.h); these
// two are the fused form of the same arithmetic for callers that own the whole chain.
//
// A wave owns a ray (as in volume_rendering.hip): lanes sweep its contiguous samples in chunks of 64.
//   forward : any ray length.  Per chunk: section-point opacity (composite_device.h: the expressions of neus.hip) -> 1 - alpha
//             + 1e-7 -> exclusive product scan with a carry (the last sample's factor never enters: cumprod_fwd_kernel) ->
//             w = alpha T -> per-lane partial sums of w rgb, one wave sum per ray at the end (integrate_fwd_kernel's order).
//   backward: rays of at most 64 K samples (K = 2 or 4 chunks held in registers: alpha, T, dL/dw and dL/dT T per sample).
//             sweep 1 recomputes the forward and forms g_w = <g_pred, rgb> (with the reference's channel quirk when asked,
//             VolumeRenderingGPU.cuh:1247) and g_rgb = g_pred w; sweep 2 walks the chunks from the ray's end: suffix sums of
//             g_T T (cumsum_kernel, inverse), the transmittance backward (cumprod_bwd_kernel: (cs[i+1] + g_bg bg) / max(om, 1e-6)),
//             g_alpha = g_w T - g_om, and the opacity backward (neus_alpha_bwd_kernel) -> g_sdf, g_gradients, g_inv_s.
// HBM traffic: 44 B / sample forward, 44 + 4 (+24 with g_gradients / g_rgb) backward -- the separate kernels move ~3x that
// and the launches cost more than the bytes at training batch sizes.
// Summation orders are those of the separate kernels (same scans, same carries) whenever the ray length is a multiple of 64;
// otherwise the suffix sums are chunked from the ray's start instead of its end (last-bit differences).
#include "composite_device.h"

using namespace psdf;

namespace {

// ---------------------------------------------------------------------------------------------------------- forward
__global__ void __launch_bounds__(PSDF_BLOCK)
    neus_composite_fwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ sdf, const float* __restrict__ dirs,
                              const float* __restrict__ gradients, const float* __restrict__ dt, const float* __restrict__ rgb,
                              const float* __restrict__ inv_s_ptr, float cos_anneal_ratio, float* __restrict__ pred,
                              float* __restrict__ bg, float* __restrict__ weights) {
  const int lane = lane_id();
  const float inv_s = inv_s_ptr[0];
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) {                  // an empty / overflowed ray renders nothing: radiance 0, everything transmitted
      if (lane == 0) {                      // (what zero- / one-initialised outputs of the separate kernels hold for it)
        pred[3 * ray] = pred[3 * ray + 1] = pred[3 * ray + 2] = 0.f;
        if (bg) bg[ray] = 1.f;
      }
      continue;
    }
    const int n = e - s;
    float carry = 1.f, r = 0.f, g = 0.f, b = 0.f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool in = i < n;
      const int64_t m = s + (in ? i : n - 1);
      const Section sc = section(sdf[m], ld3(dirs + 3 * m), ld3(gradients + 3 * m), dt[m], inv_s, cos_anneal_ratio);
      const float a = clampf(sc.q, 0.0f, 1.0f);
      const float om = (1.0f - a) + 1e-7f;
      const float fac = (i < n - 1) ? om : 1.f;
      const float incl = wave_incl_scan_mul(fac);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      const float T = carry * excl;
      carry = carry * __shfl(incl, 63, 64);
      if (in) {
        const float w = a * T;
        if (weights) weights[m] = w;
        r += w * rgb[3 * m];
        g += w * rgb[3 * m + 1];
        b += w * rgb[3 * m + 2];
      }
    }
    r = wave_sum(r);
    g = wave_sum(g);
    b = wave_sum(b);
    if (lane == 0) {
      pred[3 * ray] = r;
      pred[3 * ray + 1] = g;
      pred[3 * ray + 2] = b;
      if (bg) bg[ray] = carry;
    }
  }
}

// inclusive SUFFIX sum over the 64 lanes (mirror image of wave_incl_scan_add: the same tree, so the same roundings as the
// separate cumsum kernel's scan over the reversed ray)
__device__ __forceinline__ float wave_incl_suffix_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(v, o, 64);
    if (l + o < 64) v += t;
  }
  return v;
}

// --------------------------------------------------------------------------------------------------------- backward
template <int K>
__global__ void __launch_bounds__(PSDF_BLOCK)
    neus_composite_bwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ g_pred, const float* __restrict__ g_bg,
                              const float* __restrict__ sdf, const float* __restrict__ dirs, const float* __restrict__ gradients,
                              const float* __restrict__ dt, const float* __restrict__ rgb, const float* __restrict__ inv_s_ptr,
                              float cos_anneal_ratio, int compat, float* __restrict__ g_sdf, float* __restrict__ g_gradients,
                              float* __restrict__ g_rgb, float* __restrict__ g_inv_s) {
  const int lane = lane_id();
  const float inv_s = inv_s_ptr[0], rr = cos_anneal_ratio;
  float gs_acc = 0.f;
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    if (!ri.valid(s, e)) continue;          // their per-sample gradients: see the host wrapper (zero-filled for such containers)
    const int n = e - s;                    // <= 64 K is the CALLER'S promise (max_per_ray)
    if (n > 64 * K) {
      // a ray longer than declared: the chunked scans below would silently drop its tail.  Fail loudly instead: every
      // gradient of this ray becomes NaN (the C ABI has no other error channel out of a kernel).
      const float bad = __int_as_float(0x7fc00000);
      for (int i = lane; i < n; i += 64) {
        const int64_t m = (int64_t)s + i;
        g_sdf[m] = bad;
        if (g_rgb) g_rgb[3 * m] = g_rgb[3 * m + 1] = g_rgb[3 * m + 2] = bad;
        if (g_gradients) st3(g_gradients + 3 * m, mk3(bad, bad, bad));
      }
      continue;
    }
    const float gx = g_pred[3 * ray], gy = g_pred[3 * ray + 1], gz = g_pred[3 * ray + 2];
    float a_[K], T_[K], gw_[K], v_[K];
    float carry = 1.f;
    // ---- sweep 1: the forward again, and what the integration hands back
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = 64 * k + lane;
      const bool in = i < n;
      const int64_t m = s + (in ? i : n - 1);
      a_[k] = T_[k] = gw_[k] = v_[k] = 0.f;
      if (64 * k < n) {     // wave-uniform
        const Section sc = section(sdf[m], ld3(dirs + 3 * m), ld3(gradients + 3 * m), dt[m], inv_s, rr);
        const float a = clampf(sc.q, 0.0f, 1.0f);
        const float om = (1.0f - a) + 1e-7f;
        const float fac = (i < n - 1) ? om : 1.f;
        const float incl = wave_incl_scan_mul(fac);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry = carry * __shfl(incl, 63, 64);
        if (in) {
          const float cx = rgb[3 * m], cy = rgb[3 * m + 1], cz = compat ? cy : rgb[3 * m + 2];
          const float w = a * T;
          const float gw = gx * cx + gy * cy + gz * cz;
          if (g_rgb) {
            g_rgb[3 * m] = gx * w;
            g_rgb[3 * m + 1] = gy * w;
            g_rgb[3 * m + 2] = gz * w;
          }
          a_[k] = a;
          T_[k] = T;
          gw_[k] = gw;
          v_[k] = (gw * a) * T;          // g_T * T  (hotpath: cs = cumsum(g_T * T), g_T = g_w * alpha)
        }
      }
    }
    const float gb = (g_bg ? g_bg[ray] : 0.f) * carry;      // carry == bg transmittance
    // ---- sweep 2, from the end of the ray: suffix sums, transmittance backward, opacity backward
    float tail = 0.f;                  // sum of v over the chunks behind the current one
#pragma unroll
    for (int k = K - 1; k >= 0; k--) {
      if (64 * k >= n) continue;       // wave-uniform
      const int i = 64 * k + lane;
      const bool in = i < n;
      const int64_t m = s + (in ? i : n - 1);
      const float suf = wave_incl_suffix_add(v_[k]) + tail;          // cs[i] = sum_{j >= i} v[j]
      float cs_next = __shfl_down(suf, 1, 64);                        // cs[i + 1]
      if (lane == 63) cs_next = tail;
      tail = __shfl(suf, 0, 64);
      if (in) {
        float g_om = 0.f;
        if (i < n - 1) {
          const float om = fmaxf((1.0f - a_[k]) + 1e-7f, 1e-6f);
          g_om = cs_next / om;
          g_om += gb / om;
        }
        const float g_alpha = gw_[k] * T_[k] - g_om;
        // opacity backward (neus_alpha_bwd_kernel)
        const v3 dir = ld3(dirs + 3 * m);
        const float d = dt[m];
        const Section sc = section(sdf[m], dir, ld3(gradients + 3 * m), d, inv_s, rr);
        const float gq = (sc.q >= 0.0f && sc.q <= 1.0f) ? g_alpha : 0.0f;
        const float den = sc.c + 1e-5f;
        const float g_p = gq / den;
        const float g_c = -gq * (sc.p + 1e-5f) / (den * den);
        const float g_up = (g_p + g_c) * (sc.pc * (1.0f - sc.pc));
        const float g_un = -g_p * (sc.nc * (1.0f - sc.nc));
        const float g_ep = g_up * inv_s, g_en = g_un * inv_s;
        gs_acc += g_up * sc.ep + g_un * sc.en;
        g_sdf[m] = g_ep + g_en;
        if (g_gradients) {
          const float g_ic = (g_en - g_ep) * (d * 0.5f);
          const float g_tc = g_ic * ((sc.pre_a > 0.f ? 0.5f * (1.0f - rr) : 0.f) + (sc.pre_b > 0.f ? rr : 0.f));
          st3(g_gradients + 3 * m, g_tc * dir);
        }
      }
    }
  }
  if (g_inv_s) {
    gs_acc = wave_sum(gs_acc);
    if (lane == 0 && gs_acc != 0.f) atomicAdd(g_inv_s, gs_acc);
  }
}

// ------------------------------------------------------------------------------------------------ background NeRF
// NerfHash + VolumeRenderingNerf.compute_weights + integrate (models.py:520, volume_rendering_modules.py:72-86,176-190) and the
// composition with the foreground (train_permuto_sdf.py:160-165, pred = pred_fg + bg_transmittance_fg * pred_bg), one launch
// per direction: density = softplus(raw) -> alpha = 1 - exp(-density dt) -> 1 - alpha + 1e-7 -> exclusive product scan ->
// w = alpha T -> sum w rgb.  The expressions and scan orders of nerf_alpha_kernel (neus.hip), cumprod_fwd_kernel,
// integrate_fwd_kernel; the backward mirrors neus_composite_bwd_kernel with the NeRF opacity in the place of the NeuS one.
__device__ __forceinline__ float softplus20c(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

__global__ void __launch_bounds__(PSDF_BLOCK)
    nerf_composite_fwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ raw, const float* __restrict__ dt,
                              const float* __restrict__ rgb, const float* __restrict__ fg_pred, const float* __restrict__ fg_bg,
                              float* __restrict__ pred_bg, float* __restrict__ pred) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    float r = 0.f, g = 0.f, b = 0.f;
    if (ri.valid(s, e)) {
      const int n = e - s;
      float carry = 1.f;
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool in = i < n;
        const int64_t m = s + (in ? i : n - 1);
        const float a = 1.0f - expf(-softplus20c(raw[m]) * dt[m]);
        const float om = (1.0f - a) + 1e-7f;
        const float fac = (i < n - 1) ? om : 1.f;
        const float incl = wave_incl_scan_mul(fac);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry = carry * __shfl(incl, 63, 64);
        if (in) {
          const float w = a * T;
          r += w * rgb[3 * m];
          g += w * rgb[3 * m + 1];
          b += w * rgb[3 * m + 2];
        }
      }
      r = wave_sum(r);
      g = wave_sum(g);
      b = wave_sum(b);
    }
    if (lane == 0) {
      pred_bg[3 * ray] = r;
      pred_bg[3 * ray + 1] = g;
      pred_bg[3 * ray + 2] = b;
      if (pred) {
        const float t = fg_bg[ray];
        pred[3 * ray] = fg_pred[3 * ray] + t * r;
        pred[3 * ray + 1] = fg_pred[3 * ray + 1] + t * g;
        pred[3 * ray + 2] = fg_pred[3 * ray + 2] + t * b;
      }
    }
  }
}

template <int K>
__global__ void __launch_bounds__(PSDF_BLOCK)
    nerf_composite_bwd_kernel(int nr_rays, RayIndex ri, const float* __restrict__ g_pred, const float* __restrict__ fg_bg,
                              const float* __restrict__ raw, const float* __restrict__ dt, const float* __restrict__ rgb,
                              int compat, float* __restrict__ g_raw, float* __restrict__ g_rgb, float* __restrict__ g_fg_bg) {
  const int lane = lane_id();
  RAY_LOOP(ray, nr_rays) {
    int s, e;
    ri.get(ray, s, e);
    const float t = fg_bg ? fg_bg[ray] : 1.f;
    const float ux = g_pred[3 * ray], uy = g_pred[3 * ray + 1], uz = g_pred[3 * ray + 2];   // dL/d pred
    if (!ri.valid(s, e)) {
      if (g_fg_bg && lane == 0) g_fg_bg[ray] = 0.f;
      continue;
    }
    const int n = e - s;
    if (n > 64 * K) {   // longer than the caller declared: fail loudly (see neus_composite_bwd_kernel)
      const float bad = __int_as_float(0x7fc00000);
      for (int i = lane; i < n; i += 64) {
        const int64_t m = (int64_t)s + i;
        g_raw[m] = bad;
        g_rgb[3 * m] = g_rgb[3 * m + 1] = g_rgb[3 * m + 2] = bad;
      }
      if (g_fg_bg && lane == 0) g_fg_bg[ray] = bad;
      continue;
    }
    const float gx = t * ux, gy = t * uy, gz = t * uz;                                       // dL/d pred_bg
    float a_[K], T_[K], gw_[K], v_[K], e_[K];
    float carry = 1.f, pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = 64 * k + lane;
      const bool in = i < n;
      const int64_t m = s + (in ? i : n - 1);
      a_[k] = T_[k] = gw_[k] = v_[k] = e_[k] = 0.f;
      if (64 * k < n) {     // wave-uniform
        const float ex = expf(-softplus20c(raw[m]) * dt[m]);
        const float a = 1.0f - ex;
        const float om = (1.0f - a) + 1e-7f;
        const float fac = (i < n - 1) ? om : 1.f;
        const float incl = wave_incl_scan_mul(fac);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry = carry * __shfl(incl, 63, 64);
        if (in) {
          const float cx = rgb[3 * m], cy = rgb[3 * m + 1], cz_true = rgb[3 * m + 2];
          const float cz = compat ? cy : cz_true;
          const float w = a * T;
          pr += w * cx;
          pg += w * cy;
          pb += w * cz_true;
          const float gw = gx * cx + gy * cy + gz * cz;
          g_rgb[3 * m] = gx * w;
          g_rgb[3 * m + 1] = gy * w;
          g_rgb[3 * m + 2] = gz * w;
          a_[k] = a;
          T_[k] = T;
          gw_[k] = gw;
          e_[k] = ex;
          v_[k] = (gw * a) * T;
        }
      }
    }
    if (g_fg_bg) {   // dL/d (foreground bg transmittance) = <dL/d pred, pred_bg>
      pr = wave_sum(pr);
      pg = wave_sum(pg);
      pb = wave_sum(pb);
      if (lane == 0) g_fg_bg[ray] = (ux * pr + uy * pg) + uz * pb;
    }
    float tail = 0.f;
#pragma unroll
    for (int k = K - 1; k >= 0; k--) {
      if (64 * k >= n) continue;       // wave-uniform
      const int i = 64 * k + lane;
      const bool in = i < n;
      const int64_t m = s + (in ? i : n - 1);
      const float suf = wave_incl_suffix_add(v_[k]) + tail;
      float cs_next = __shfl_down(suf, 1, 64);
      if (lane == 63) cs_next = tail;
      tail = __shfl(suf, 0, 64);
      if (in) {
        float g_om = 0.f;
        if (i < n - 1) g_om = cs_next / fmaxf((1.0f - a_[k]) + 1e-7f, 1e-6f);
        const float g_alpha = gw_[k] * T_[k] - g_om;                 // alpha enters as w = alpha T and as 1 - alpha + 1e-7
        const float x = raw[m];
        const float g_dens = g_alpha * e_[k] * dt[m];                // alpha = 1 - exp(-dens dt)
        g_raw[m] = g_dens * (x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)));
      }
    }
  }
}

}  // namespace

extern "C" {

// Background NeRF rendering + composition with the foreground in one launch per direction (see the kernels): raw_density [M],
// dt [M], rgb [M,3] of the background container; fg_pred [R,3] / fg_bg [R] (both or neither): pred [R,3] = fg_pred + fg_bg *
// pred_bg; pred_bg [R,3] always written (zeros for empty rays).
int psdf_nerf_composite_forward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* raw_density,
                                const float* dt, const float* rgb, const float* fg_pred, const float* fg_bg, float* pred_bg,
                                float* pred, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (!raw_density || !dt || !rgb || !pred_bg || (!equal && !start_end) || ((fg_pred || fg_bg || pred) && !(fg_pred && fg_bg && pred)))
    return PSDF_ERR_ARG;
  hipLaunchKernelGGL(nerf_composite_fwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     RayIndex{start_end, equal, fixed, max_nr_samples}, raw_density, dt, rgb, fg_pred, fg_bg, pred_bg, pred);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// Backward for dL/d pred [R,3] (of the COMPOSED radiance when fg_bg is given, else of pred_bg): grad_raw_density [M], grad_rgb
// [M,3], and (optional, with fg_bg) grad_fg_bg [R] = <dL/d pred, pred_bg>.  max_per_ray as in psdf_neus_composite_backward.
int psdf_nerf_composite_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int max_per_ray,
                                 const float* grad_pred, const float* fg_bg, const float* raw_density, const float* dt,
                                 const float* rgb, int reference_compat, float* grad_raw_density, float* grad_rgb,
                                 float* grad_fg_bg, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (!grad_pred || !raw_density || !dt || !rgb || !grad_raw_density || !grad_rgb || (!equal && !start_end) || max_per_ray < 1)
    return PSDF_ERR_ARG;
  if (max_per_ray > 256) return PSDF_ERR_UNSUPPORTED;
  const RayIndex ri{start_end, equal, fixed, max_nr_samples};
#define GO(K_)                                                                                                          \
  hipLaunchKernelGGL(nerf_composite_bwd_kernel<K_>, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream,  \
                     nr_rays, ri, grad_pred, fg_bg, raw_density, dt, rgb, reference_compat, grad_raw_density, grad_rgb, \
                     grad_fg_bg)
  if (max_per_ray <= 64)
    GO(1);
  else if (max_per_ray <= 128)
    GO(2);
  else
    GO(4);
#undef GO
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// pred [R,3] (rows of invalid / empty rays: 0), bg [R] optional (bg transmittance; 1 for such rays),
// weights [N] optional (alpha T per sample).  Ray-index arguments as everywhere (include/psdf.h).
int psdf_neus_composite_forward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* sdf,
                                const float* dirs, const float* gradients, const float* dt, const float* rgb, const float* inv_s,
                                float cos_anneal_ratio, float* pred, float* bg, float* weights, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (!sdf || !dirs || !gradients || !dt || !rgb || !inv_s || !pred || (!equal && !start_end)) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(neus_composite_fwd_kernel, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, nr_rays,
                     RayIndex{start_end, equal, fixed, max_nr_samples}, sdf, dirs, gradients, dt, rgb, inv_s, cos_anneal_ratio,
                     pred, bg, weights);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// Backward of the above for upstream g_pred [R,3] and (optional) g_bg [R]: g_sdf [N]; optional g_gradients [N,3], g_rgb [N,3],
// g_inv_s [1] (ACCUMULATED into: zero it first).  max_per_ray: an upper bound of the samples of any ray (the caller knows it:
// a fixed count, or max_nr_samples_per_ray + the importance samples); > 256 -> -2 (use the per-operator kernels).  Samples of
// rays the reference skips (empty, overflowed) are not written: zero-fill the outputs when the container can hold such rays.
int psdf_neus_composite_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int max_per_ray,
                                 const float* grad_pred, const float* grad_bg, const float* sdf, const float* dirs,
                                 const float* gradients, const float* dt, const float* rgb, const float* inv_s,
                                 float cos_anneal_ratio, int reference_compat, float* grad_sdf, float* grad_gradients,
                                 float* grad_rgb, float* grad_inv_s, void* stream) {
  if (nr_rays <= 0) return PSDF_OK;
  if (!grad_pred || !sdf || !dirs || !gradients || !dt || !rgb || !inv_s || !grad_sdf || (!equal && !start_end) || max_per_ray < 1)
    return PSDF_ERR_ARG;
  if (max_per_ray > 256) return PSDF_ERR_UNSUPPORTED;
  const RayIndex ri{start_end, equal, fixed, max_nr_samples};
#define GO(K_)                                                                                                            \
  hipLaunchKernelGGL(neus_composite_bwd_kernel<K_>, dim3(ray_grid(nr_rays)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream,     \
                     nr_rays, ri, grad_pred, grad_bg, sdf, dirs, gradients, dt, rgb, inv_s, cos_anneal_ratio,             \
                     reference_compat, grad_sdf, grad_gradients, grad_rgb, grad_inv_s)
  if (max_per_ray <= 64)
    GO(1);
  else if (max_per_ray <= 128)
    GO(2);
  else
    GO(4);
#undef GO
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // extern "C"
