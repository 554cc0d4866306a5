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

using namespace psdf;

namespace {

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
  acc = wave_sum(acc);
  if (lane_id() == 0 && loss) atomicAdd(loss, acc * scale);
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
  acc = wave_sum(acc);
  if (lane_id() == 0 && loss) atomicAdd(loss, acc * scale);
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
  hipLaunchKernelGGL(l1_loss_kernel, dim3(stream_grid(R * C)), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, R, C, pred, gt,
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

}  // extern "C"
