// NeuS section-point opacity (forward + backward) and the loss tails of the training step, one launch each.
//
// The reference computes these with ~30 torch elementwise launches per direction:
//   permuto_sdf_py/volume_rendering/volume_rendering_modules.py:129-172 (VolumeRenderingNeus.compute_weights: cosine
//   annealing, section-point SDFs, two sigmoids, (p + 1e-5) / (c + 1e-5) clipped to [0, 1], then 1 - alpha + 1e-7 into the
//   transmittance product), permuto_sdf_py/utils/permuto_sdf_utils.py:43-51 (rgb_loss: L1 * hit mask, mean; eikonal_loss).
// Here each is one streaming kernel over the packed samples (HBM bound: 36 B read + 8 B written per sample forward,
// 40 B + 16 B backward), with the scalar reductions (d/d inv_s, the loss values) folded in: wave sum -> one atomic per wave.
// Arithmetic is fp32 in the order the reference's expressions are written (contraction off), the sigmoid is
// 1 / (1 + exp(-x)) as torch evaluates it.
#include "psdf_common.h"
#include "composite_device.h"

using namespace psdf;

namespace {

__global__ void __launch_bounds__(PSDF_BLOCK)
    neus_alpha_fwd_kernel(int64_t N, const float* __restrict__ sdf, const float* __restrict__ dirs,
                          const float* __restrict__ gradients, const float* __restrict__ dt,
                          const float* __restrict__ inv_s_ptr, float cos_anneal_ratio, float* __restrict__ alpha,
                          float* __restrict__ one_minus_alpha) {
  const float inv_s = inv_s_ptr[0];
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const Section s = section(sdf[n], ld3(dirs + 3 * n), ld3(gradients + 3 * n), dt[n], inv_s, cos_anneal_ratio);
    const float a = clampf(s.q, 0.0f, 1.0f);
    alpha[n] = a;
    if (one_minus_alpha) one_minus_alpha[n] = (1.0f - a) + 1e-7f;         // what cumprod_alpha2transmittance is fed
  }
}

// g_alpha [N] -> g_sdf [N], g_gradients [N,3] (optional), g_inv_s [1] (optional, ACCUMULATED: zero it first)
__global__ void __launch_bounds__(PSDF_BLOCK)
    neus_alpha_bwd_kernel(int64_t N, const float* __restrict__ g_alpha, const float* __restrict__ sdf,
                          const float* __restrict__ dirs, const float* __restrict__ gradients,
                          const float* __restrict__ dt, const float* __restrict__ inv_s_ptr, float cos_anneal_ratio,
                          float* __restrict__ g_sdf, float* __restrict__ g_gradients, float* __restrict__ g_inv_s) {
  const float inv_s = inv_s_ptr[0], r = cos_anneal_ratio;
  float gs_acc = 0.f;
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const v3 dir = ld3(dirs + 3 * n);
    const float d = dt[n];
    const Section s = section(sdf[n], dir, ld3(gradients + 3 * n), d, inv_s, r);
    // clip(q, 0, 1) passes the gradient inside the closed interval (torch.clamp)
    const float gq = (s.q >= 0.0f && s.q <= 1.0f) ? g_alpha[n] : 0.0f;
    const float den = s.c + 1e-5f;
    const float g_p = gq / den;
    const float g_c = -gq * (s.p + 1e-5f) / (den * den);
    const float g_up = (g_p + g_c) * (s.pc * (1.0f - s.pc));              // through sigmoid(ep * inv_s)
    const float g_un = -g_p * (s.nc * (1.0f - s.nc));                     // through sigmoid(en * inv_s)
    const float g_ep = g_up * inv_s, g_en = g_un * inv_s;
    gs_acc += g_up * s.ep + g_un * s.en;
    g_sdf[n] = g_ep + g_en;
    if (g_gradients) {
      const float g_ic = (g_en - g_ep) * (d * 0.5f);
      // ic = -(relu(pre_a) (1-r) + relu(pre_b) r);  pre_a = -tc/2 + 1/2;  pre_b = -tc
      const float g_tc = g_ic * ((s.pre_a > 0.f ? 0.5f * (1.0f - r) : 0.f) + (s.pre_b > 0.f ? r : 0.f));
      st3(g_gradients + 3 * n, g_tc * dir);
    }
  }
  if (g_inv_s) {
    gs_acc = wave_sum(gs_acc);
    if (lane_id() == 0) atomicAdd(g_inv_s, gs_acc);
  }
}

// loss += scale * sum |gt - pred| * mask;  g_pred = scale * sign(pred - gt) * mask          (rgb_loss, scale = 1/(R*C))
// workgroup sum of `acc`, then ONE atomicAdd of scale * sum into *dst (dst may be NULL)
__device__ __forceinline__ void block_sum_atomic(float acc, float scale, float* __restrict__ dst) {
  __shared__ float part[PSDF_BLOCK / 64];
  acc = wave_sum(acc);
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && dst) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < PSDF_BLOCK / 64; w++) t += part[w];
    atomicAdd(dst, t * scale);
  }
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    l1_loss_kernel(int64_t R, int C, const float* __restrict__ pred, const float* __restrict__ gt,
                   const unsigned char* __restrict__ mask, float scale, float* __restrict__ loss,
                   float* __restrict__ g_pred) {
  float acc = 0.f;
  const int64_t total = R * C;
  for (int64_t i = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * PSDF_BLOCK) {
    const float m = mask ? (mask[i / C] ? 1.0f : 0.0f) : 1.0f;
    const float d = pred[i] - gt[i];
    acc += fabsf(d) * m;
    if (g_pred) g_pred[i] = (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) * m;
  }
  // one atomic per WORKGROUP (atomics to a single address serialise at ~10 ns each: one per wave cost 8 us of this 12-us launch)
  block_sum_atomic(acc, scale, loss);
}

// loss += scale * sum (|g| - 1)^2;  g_grad = scale * 2 (|g| - 1) g / |g|                   (eikonal_loss, scale = w/N)
__global__ void __launch_bounds__(PSDF_BLOCK)
    eikonal_loss_kernel(int64_t N, const float* __restrict__ grad, float scale, float* __restrict__ loss,
                        float* __restrict__ g_grad) {
  float acc = 0.f;
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const v3 g = ld3(grad + 3 * n);
    const float nrm = sqrtf(dot3(g, g));
    const float e = nrm - 1.0f;
    acc += e * e;
    if (g_grad) st3(g_grad + 3 * n, (nrm > 0.f ? scale * 2.0f * e / nrm : 0.f) * g);
  }
  block_sum_atomic(acc, scale, loss);
}

// ---------------------------------------------------------------------------------------------------------------------
// More elementwise chains of the training step, one launch per direction each (the torch forms are 5-25 launches forward
// and about twice that backward; the step is host bound, csrc/../train_step.py):
//   normalize3      F.normalize(x, dim=-1) (eps 1e-12)                                       models.py:272,367 ...
//   curvature shift points + eps * cross(normalize(g), normalize(rand))                      models.py:266-277
//   curvature loss  scale * sum acos(clamp(n(g) . n(g2), -1+1e-6, 1-1e-6)) / pi              models.py:282-289, train_permuto_sdf.py:363
//   offsurface loss scale * sum exp(-100 |sdf|)                                              train_permuto_sdf.py:372-375
//   nerf alpha      alpha = 1 - exp(-softplus(raw) dt), one_minus = 1 - alpha + 1e-7        models.py:520, volume_rendering_modules.py:72-86
// All fp32, expressions in the order torch evaluates them.
struct Nrm {
  v3 y;
  float norm, denom;
};
__device__ __forceinline__ Nrm normalize_eps(v3 x) {
  Nrm r;
  r.norm = sqrtf(dot3(x, x));
  r.denom = fmaxf(r.norm, 1e-12f);
  r.y = v3{x.x / r.denom, x.y / r.denom, x.z / r.denom};
  return r;
}
// gradient of y = x / max(|x|, eps) for an upstream gy
__device__ __forceinline__ v3 normalize_bwd(const Nrm& n, v3 gy) {
  const float inv = 1.0f / n.denom;
  v3 g = inv * gy;
  if (n.norm > 1e-12f) {   // the clamp passes the gradient of the norm only above eps
    const float s = dot3(gy, n.y) * inv;
    g = g - s * n.y;
  }
  return g;
}
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
  return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    normalize3_kernel(int64_t N, const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ out) {
  // gy == NULL: out = normalize(x); else out = d normalize / dx applied to gy
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const Nrm r = normalize_eps(ld3(x + 3 * n));
    st3(out + 3 * n, gy ? normalize_bwd(r, ld3(gy + 3 * n)) : r.y);
  }
}

// The colour heads end in a sigmoid applied to a 3-row feature-major MLP output that the compositing kernels want as [N, 3]:
// forward  y[n][c] = sigmoid(x_fm[c][n])            (transpose + sigmoid in one pass instead of a copy and an elementwise launch)
// backward out_fm[c][n] = g[n][c] y[n][c] (1 - y[n][c])   (three elementwise launches and a transposing copy otherwise)
__global__ void __launch_bounds__(PSDF_BLOCK)
    sigmoid_rows_kernel(int64_t N, int C, const float* __restrict__ x_fm, const float* __restrict__ g, const float* __restrict__ y,
                        float* __restrict__ out) {
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    for (int c = 0; c < C; c++) {
      if (x_fm) {
        out[n * C + c] = 1.0f / (1.0f + expf(-x_fm[(int64_t)c * N + n]));
      } else {
        const float yv = y[n * C + c];
        out[(int64_t)c * N + n] = g[n * C + c] * yv * (1.0f - yv);
      }
    }
  }
}

// forward (g_shifted == NULL): out = points + eps * cross(normalize(g), normalize(rnd));  backward: out = d / d g applied
// to g_shifted (points and rnd carry no gradient)
__global__ void __launch_bounds__(PSDF_BLOCK)
    curvature_shift_kernel(int64_t N, const float* __restrict__ points, const float* __restrict__ g,
                           const float* __restrict__ rnd, float eps, const float* __restrict__ g_shifted,
                           float* __restrict__ out) {
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const Nrm a = normalize_eps(ld3(g + 3 * n));
    const v3 r = normalize_eps(ld3(rnd + 3 * n)).y;
    if (!g_shifted) {
      const v3 t = cross3(a.y, r);
      st3(out + 3 * n, ld3(points + 3 * n) + eps * t);
    } else {
      const v3 gt = eps * ld3(g_shifted + 3 * n);
      st3(out + 3 * n, normalize_bwd(a, cross3(r, gt)));   // d (a x r) . gt / d a = r x gt
    }
  }
}

// loss += scale * sum acos(clamp(n(a) . n(b))) / pi;  ga, gb (optional) = its gradients
__global__ void __launch_bounds__(PSDF_BLOCK)
    curvature_loss_kernel(int64_t N, const float* __restrict__ a, const float* __restrict__ b, float scale,
                          float* __restrict__ loss, float* __restrict__ ga, float* __restrict__ gb) {
  float acc = 0.f;
  const float lo = -1.0f + 1e-6f, hi = 1.0f - 1e-6f, inv_pi = (float)(1.0 / 3.14159265358979323846);
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const Nrm na = normalize_eps(ld3(a + 3 * n)), nb = normalize_eps(ld3(b + 3 * n));
    const float dot = dot3(na.y, nb.y);
    const float u = fminf(fmaxf(dot, lo), hi);
    acc += acosf(u) * inv_pi;
    if (ga) {
      const float gu = (dot >= lo && dot <= hi) ? -(scale * inv_pi) / sqrtf((1.0f - u) * (1.0f + u)) : 0.0f;
      st3(ga + 3 * n, normalize_bwd(na, gu * nb.y));
      st3(gb + 3 * n, normalize_bwd(nb, gu * na.y));
    }
  }
  block_sum_atomic(acc, scale, loss);
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    offsurface_loss_kernel(int64_t N, const float* __restrict__ sdf, float sharp, float scale, float* __restrict__ loss,
                           float* __restrict__ g_sdf) {
  float acc = 0.f;
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const float s = sdf[n];
    const float e = expf(-sharp * fabsf(s));
    acc += e;
    if (g_sdf) g_sdf[n] = scale * e * (s > 0.f ? -sharp : (s < 0.f ? sharp : 0.f));
  }
  block_sum_atomic(acc, scale, loss);
}

__device__ __forceinline__ float softplus20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// forward (g_alpha == NULL): alpha, one_minus;  backward: g_raw = d / d raw of (alpha . g_alpha + one_minus . g_one_minus)
__global__ void __launch_bounds__(PSDF_BLOCK)
    nerf_alpha_kernel(int64_t N, const float* __restrict__ raw, const float* __restrict__ dt, float* __restrict__ alpha,
                      float* __restrict__ one_minus, const float* __restrict__ g_alpha,
                      const float* __restrict__ g_one_minus, float* __restrict__ g_raw) {
  for (int64_t n = (int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x; n < N; n += (int64_t)gridDim.x * PSDF_BLOCK) {
    const float x = raw[n], d = dt[n];
    const float dens = softplus20(x);
    const float e = expf(-dens * d);
    if (!g_raw) {
      const float a = 1.0f - e;
      alpha[n] = a;
      one_minus[n] = (1.0f - a) + 1e-7f;
    } else {
      const float ga = (g_alpha ? g_alpha[n] : 0.f) - (g_one_minus ? g_one_minus[n] : 0.f);   // one_minus = 1 - alpha + 1e-7
      const float g_dens = ga * e * d;                                                         // alpha = 1 - exp(-dens dt)
      g_raw[n] = g_dens * (x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)));                       // softplus' = sigmoid
    }
  }
}

inline unsigned stream_grid(int64_t n) {
  const unsigned b = psdf_blocks(n, PSDF_BLOCK);
  return b < 4096u ? (b ? b : 1u) : 4096u;     // >= 16 workgroups per CU on 256 CUs, grid-stride beyond
}

}  // namespace

extern "C" {

int psdf_neus_alpha_forward(int64_t N, const float* sdf, const float* dirs, const float* gradients, const float* dt,
                            const float* inv_s, float cos_anneal_ratio, float* alpha, float* one_minus_alpha,
                            void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !sdf || !dirs || !gradients || !dt || !inv_s || !alpha) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(neus_alpha_fwd_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, sdf, dirs,
                     gradients, dt, inv_s, cos_anneal_ratio, alpha, one_minus_alpha);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_neus_alpha_backward(int64_t N, const float* grad_alpha, const float* sdf, const float* dirs,
                             const float* gradients, const float* dt, const float* inv_s, float cos_anneal_ratio,
                             float* grad_sdf, float* grad_gradients, float* grad_inv_s, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !grad_alpha || !sdf || !dirs || !gradients || !dt || !inv_s || !grad_sdf) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(neus_alpha_bwd_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, grad_alpha,
                     sdf, dirs, gradients, dt, inv_s, cos_anneal_ratio, grad_sdf, grad_gradients, grad_inv_s);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_l1_loss(int64_t R, int C, const float* pred, const float* gt, const unsigned char* mask, float scale,
                 float* loss, float* grad_pred, void* stream) {
  if (R == 0) return PSDF_OK;
  if (R < 0 || C <= 0 || !pred || !gt) return PSDF_ERR_ARG;
  unsigned blocks = psdf_blocks(R * C, PSDF_BLOCK * 4);      // four elements per thread: few workgroups, few atomics
  blocks = blocks < 1u ? 1u : (blocks > 256u ? 256u : blocks);
  hipLaunchKernelGGL(l1_loss_kernel, dim3(blocks), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, R, C, pred, gt,
                     mask, scale, loss, grad_pred);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_eikonal_loss(int64_t N, const float* gradients, float scale, float* loss, float* grad_gradients, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !gradients) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(eikonal_loss_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, gradients,
                     scale, loss, grad_gradients);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_normalize3(int64_t N, const float* x, const float* grad_y, float* out, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !x || !out) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(normalize3_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, x, grad_y, out);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sigmoid_rows(int64_t N, int C, const float* x_fm, float* y, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || C < 1 || C > 16 || !x_fm || !y) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(sigmoid_rows_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, C, x_fm,
                     (const float*)nullptr, (const float*)nullptr, y);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_sigmoid_rows_backward(int64_t N, int C, const float* grad_y, const float* y, float* grad_x_fm, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || C < 1 || C > 16 || !grad_y || !y || !grad_x_fm) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(sigmoid_rows_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, C,
                     (const float*)nullptr, grad_y, y, grad_x_fm);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_curvature_shift(int64_t N, const float* points, const float* gradients, const float* rand_directions, float epsilon,
                         const float* grad_shifted, float* out, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !gradients || !rand_directions || !out || (!grad_shifted && !points)) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(curvature_shift_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, points,
                     gradients, rand_directions, epsilon, grad_shifted, out);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_curvature_loss(int64_t N, const float* gradients, const float* gradients_shifted, float scale, float* loss,
                        float* grad_gradients, float* grad_gradients_shifted, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !gradients || !gradients_shifted || (!grad_gradients) != (!grad_gradients_shifted)) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(curvature_loss_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, gradients,
                     gradients_shifted, scale, loss, grad_gradients, grad_gradients_shifted);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_offsurface_loss(int64_t N, const float* sdf, float sharpness, float scale, float* loss, float* grad_sdf,
                         void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !sdf) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(offsurface_loss_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, sdf,
                     sharpness, scale, loss, grad_sdf);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_nerf_alpha_forward(int64_t N, const float* raw_density, const float* dt, float* alpha, float* one_minus_alpha,
                            void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !raw_density || !dt || !alpha || !one_minus_alpha) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(nerf_alpha_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, raw_density, dt,
                     alpha, one_minus_alpha, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

int psdf_nerf_alpha_backward(int64_t N, const float* raw_density, const float* dt, const float* grad_alpha,
                             const float* grad_one_minus_alpha, float* grad_raw_density, void* stream) {
  if (N == 0) return PSDF_OK;
  if (N < 0 || !raw_density || !dt || !grad_raw_density) return PSDF_ERR_ARG;
  hipLaunchKernelGGL(nerf_alpha_kernel, dim3(stream_grid(N)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, N, raw_density, dt,
                     (float*)nullptr, (float*)nullptr, grad_alpha, grad_one_minus_alpha, grad_raw_density);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

}  // extern "C"
