// Fused AdamW step over a flat fp32 parameter buffer (one launch per tensor, 16-B vector accesses).
// The reference optimises 3 x 12.6 M lattice parameters with torch.optim.AdamW(betas=(0.9,0.99), eps=1e-15)
// (permuto_sdf_py/train_permuto_sdf.py:293-304); this is the same update rule as torch's, streaming
// 4 arrays once: 28 B/parameter of HBM traffic, pure bandwidth.
#include "psdf_common.h"

namespace {
__device__ __forceinline__ void adamw_range(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                            float* __restrict__ m, float* __restrict__ v, float lr, float beta1,
                                            float beta2, float eps, float weight_decay, float bias_corr1,
                                            float bias_corr2_sqrt, float grad_scale, int64_t first, int64_t stride) {
  for (int64_t i = first; i < n; i += stride) {
    if (i + 3 < n) {
      float4 P = *reinterpret_cast<float4*>(p + i);
      const float4 G = *reinterpret_cast<const float4*>(g + i);
      float4 M = *reinterpret_cast<float4*>(m + i);
      float4 V = *reinterpret_cast<float4*>(v + i);
      float* pp = &P.x;
      const float* gp = &G.x;
      float* mp = &M.x;
      float* vp = &V.x;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float gk = gp[k] * grad_scale;
        pp[k] = pp[k] * (1.f - lr * weight_decay);
        mp[k] = beta1 * mp[k] + (1.f - beta1) * gk;
        vp[k] = beta2 * vp[k] + (1.f - beta2) * gk * gk;
        const float denom = sqrtf(vp[k]) / bias_corr2_sqrt + eps;
        pp[k] = pp[k] - (lr / bias_corr1) * (mp[k] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = P;
      *reinterpret_cast<float4*>(m + i) = M;
      *reinterpret_cast<float4*>(v + i) = V;
    } else {
      for (int64_t j = i; j < n; j++) {
        const float gk = g[j] * grad_scale;
        float pj = p[j] * (1.f - lr * weight_decay);
        const float mj = beta1 * m[j] + (1.f - beta1) * gk;
        const float vj = beta2 * v[j] + (1.f - beta2) * gk * gk;
        const float denom = sqrtf(vj) / bias_corr2_sqrt + eps;
        pj = pj - (lr / bias_corr1) * (mj / denom);
        p[j] = pj;
        m[j] = mj;
        v[j] = vj;
      }
    }
  }
}

__global__ void __launch_bounds__(PSDF_BLOCK)
    adamw_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, float lr, float beta1, float beta2, float eps, float weight_decay,
                 float bias_corr1, float bias_corr2_sqrt, float grad_scale) {
  adamw_range(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2_sqrt, grad_scale,
              ((int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x) * 4, (int64_t)gridDim.x * PSDF_BLOCK * 4);
}

// Many SMALL tensors in one launch (the MLP weights and biases: ~35 tensors of a few hundred to a few thousand floats
// per step would otherwise be ~35 launches): blockIdx.y selects the tensor.
constexpr int ADAMW_MAX_TENSORS = 64;
struct AdamwMulti {
  int64_t n[ADAMW_MAX_TENSORS];
  float* p[ADAMW_MAX_TENSORS];
  const float* g[ADAMW_MAX_TENSORS];
  float* m[ADAMW_MAX_TENSORS];
  float* v[ADAMW_MAX_TENSORS];
};
__global__ void __launch_bounds__(PSDF_BLOCK)
    adamw_multi_kernel(AdamwMulti a, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                       float bias_corr2_sqrt, float grad_scale) {
  const int t = blockIdx.y;
  adamw_range(a.n[t], a.p[t], a.g[t], a.m[t], a.v[t], lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2_sqrt,
              grad_scale, ((int64_t)blockIdx.x * PSDF_BLOCK + threadIdx.x) * 4, (int64_t)gridDim.x * PSDF_BLOCK * 4);
}
// AdamW over a tensor cut into blocks of `block_elems` floats, skipping the blocks where the dense update is the identity:
// a block whose gradient AND both moments are exactly zero (its rows were never touched by any batch) keeps p, m, v as they
// are under AdamW with weight_decay = 0 -- so with `touched` (set by psdf_encode_forward_mark, a superset of the rows the
// backward writes) and `active` (1 once a block has ever been updated: its moments keep decaying, exactly as torch's dense
// AdamW makes them) the result is bit-for-bit what the dense kernel produces, while never-touched blocks are not even
// read.  The gradient is zeroed in the same pass (the buffer is persistent: no allocation, no separate fill launch).
// One wave per block, float4 per lane.
__device__ __forceinline__ void adamw_blocks_body(int64_t n_blocks, int block_elems, float* __restrict__ p, float* __restrict__ g,
                        float* __restrict__ m, float* __restrict__ v, unsigned char* __restrict__ touched,
                        unsigned char* __restrict__ active, float lr, float beta1, float beta2, float eps, float bias_corr1,
                        float bias_corr2_sqrt, float grad_scale, int zero_grad) {
  const int lane = threadIdx.x & 63;
  for (int64_t b = (int64_t)blockIdx.x * (PSDF_BLOCK / 64) + (threadIdx.x >> 6); b < n_blocks;
       b += (int64_t)gridDim.x * (PSDF_BLOCK / 64)) {
    const unsigned char t = touched[b], a = active[b];
    if (!(t | a)) continue;   // wave-uniform
    if (lane == 0) {
      if (!a) active[b] = 1;
      if (t) touched[b] = 0;
    }
    const int64_t base = b * block_elems;
    for (int i = lane * 4; i < block_elems; i += 256) {
      float4 P = *reinterpret_cast<float4*>(p + base + i);
      float4 G = *reinterpret_cast<float4*>(g + base + i);
      float4 M = *reinterpret_cast<float4*>(m + base + i);
      float4 V = *reinterpret_cast<float4*>(v + base + i);
      float* pp = &P.x;
      const float* gp = &G.x;
      float* mp = &M.x;
      float* vp = &V.x;
#pragma unroll
      for (int k = 0; k < 4; k++) {   // same expression order as adamw_range with weight_decay = 0
        const float gk = gp[k] * grad_scale;
        pp[k] = pp[k] * (1.f - lr * 0.f);
        mp[k] = beta1 * mp[k] + (1.f - beta1) * gk;
        vp[k] = beta2 * vp[k] + (1.f - beta2) * gk * gk;
        const float denom = sqrtf(vp[k]) / bias_corr2_sqrt + eps;
        pp[k] = pp[k] - (lr / bias_corr1) * (mp[k] / denom);
      }
      *reinterpret_cast<float4*>(p + base + i) = P;
      *reinterpret_cast<float4*>(m + base + i) = M;
      *reinterpret_cast<float4*>(v + base + i) = V;
      // every processed block is cleared, touched or not: a gradient that reached an active, unmarked block (a retained graph's
      // backward after the step that consumed its marks) is applied ONCE, not on every later step (G is in registers: free)
      if (zero_grad) *reinterpret_cast<float4*>(g + base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__global__ void __launch_bounds__(PSDF_BLOCK)
    adamw_blocks_kernel(int64_t n_blocks, int block_elems, float* __restrict__ p, float* __restrict__ g,
                        float* __restrict__ m, float* __restrict__ v, unsigned char* __restrict__ touched,
                        unsigned char* __restrict__ active, float lr, float beta1, float beta2, float eps, float bias_corr1,
                        float bias_corr2_sqrt, float grad_scale, int zero_grad) {
  adamw_blocks_body(n_blocks, block_elems, p, g, m, v, touched, active, lr, beta1, beta2, eps, bias_corr1, bias_corr2_sqrt,
                    grad_scale, zero_grad);
}
constexpr int ADAMW_BLOCKS_MAX = 8;
struct AdamwBlocksMulti {
  int64_t n_blocks[ADAMW_BLOCKS_MAX];
  int block_elems[ADAMW_BLOCKS_MAX];
  float *p[ADAMW_BLOCKS_MAX], *g[ADAMW_BLOCKS_MAX], *m[ADAMW_BLOCKS_MAX], *v[ADAMW_BLOCKS_MAX];
  unsigned char *touched[ADAMW_BLOCKS_MAX], *active[ADAMW_BLOCKS_MAX];
  float lr[ADAMW_BLOCKS_MAX], beta1[ADAMW_BLOCKS_MAX], beta2[ADAMW_BLOCKS_MAX], eps[ADAMW_BLOCKS_MAX], bc1[ADAMW_BLOCKS_MAX],
      bc2[ADAMW_BLOCKS_MAX];
};
__global__ void __launch_bounds__(PSDF_BLOCK) adamw_blocks_multi_kernel(AdamwBlocksMulti a, float grad_scale, int zero_grad) {
  const int t = blockIdx.y;
  adamw_blocks_body(a.n_blocks[t], a.block_elems[t], a.p[t], a.g[t], a.m[t], a.v[t], a.touched[t], a.active[t], a.lr[t], a.beta1[t],
                    a.beta2[t], a.eps[t], a.bc1[t], a.bc2[t], grad_scale, zero_grad);
}
}  // namespace

extern "C" {
// AdamW (weight_decay = 0) over n_blocks blocks of block_elems floats (a multiple of 4; tensors 16-byte aligned), skipping
// never-touched blocks; `touched` / `active` are [n_blocks] bytes (touched is consumed: reset to 0), zero_grad != 0 clears
// the gradient of the blocks it processed.  Same result as psdf_adamw_step on the whole tensor.
int psdf_adamw_step_blocks(int64_t n_blocks, int block_elems, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                           unsigned char* touched, unsigned char* active, float lr, float beta1, float beta2, float eps,
                           int step, float grad_scale, int zero_grad, void* stream) {
  if (n_blocks <= 0) return PSDF_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !touched || !active || step < 1 || block_elems <= 0 || (block_elems & 3))
    return PSDF_ERR_ARG;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return PSDF_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  unsigned blocks = psdf_blocks(n_blocks, PSDF_BLOCK / 64);
  if (blocks > 8192u) blocks = 8192u;
  hipLaunchKernelGGL(adamw_blocks_kernel, dim3(blocks), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, n_blocks, block_elems,
                     param, grad, exp_avg, exp_avg_sq, touched, active, lr, beta1, beta2, eps, bc1, bc2, grad_scale, zero_grad);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// The same update for up to ADAMW_BLOCKS_MAX touched-rows tensors in ONE launch (blockIdx.y = tensor; every tensor with its own
// learning rate, betas, eps and step count: the lattices of a training step sit in different parameter groups).  A cfg-4 step
// updates three lattices: three launches of ~42 us each, far from filling the chip (profiles/r04_cfg4_manual_kernel_stats.txt).
int psdf_adamw_step_blocks_multi(int n_tensors, const int64_t* n_blocks, const int* block_elems, float* const* params,
                                 float* const* grads, float* const* exp_avgs, float* const* exp_avg_sqs,
                                 unsigned char* const* touched, unsigned char* const* active, const float* lr,
                                 const float* beta1, const float* beta2, const float* eps, const int* step, float grad_scale,
                                 int zero_grad, void* stream) {
  if (n_tensors <= 0) return PSDF_OK;
  if (n_tensors > ADAMW_BLOCKS_MAX || !n_blocks || !block_elems || !params || !grads || !exp_avgs || !exp_avg_sqs || !touched ||
      !active || !lr || !beta1 || !beta2 || !eps || !step)
    return PSDF_ERR_ARG;
  AdamwBlocksMulti a;
  int64_t most = 0;
  for (int t = 0; t < n_tensors; t++) {
    if (n_blocks[t] < 0 || !params[t] || !grads[t] || !exp_avgs[t] || !exp_avg_sqs[t] || !touched[t] || !active[t] || step[t] < 1 ||
        block_elems[t] <= 0 || (block_elems[t] & 3))
      return PSDF_ERR_ARG;
    if ((((uintptr_t)params[t] | (uintptr_t)grads[t] | (uintptr_t)exp_avgs[t] | (uintptr_t)exp_avg_sqs[t]) & 15) != 0) return PSDF_ERR_ARG;
    a.n_blocks[t] = n_blocks[t];
    a.block_elems[t] = block_elems[t];
    a.p[t] = params[t]; a.g[t] = grads[t]; a.m[t] = exp_avgs[t]; a.v[t] = exp_avg_sqs[t];
    a.touched[t] = touched[t]; a.active[t] = active[t];
    a.lr[t] = lr[t]; a.beta1[t] = beta1[t]; a.beta2[t] = beta2[t]; a.eps[t] = eps[t];
    a.bc1[t] = 1.f - powf(beta1[t], (float)step[t]);
    a.bc2[t] = sqrtf(1.f - powf(beta2[t], (float)step[t]));
    most = n_blocks[t] > most ? n_blocks[t] : most;
  }
  if (most == 0) return PSDF_OK;
  unsigned blocks = psdf_blocks(most, PSDF_BLOCK / 64);
  if (blocks > 8192u) blocks = 8192u;
  hipLaunchKernelGGL(adamw_blocks_multi_kernel, dim3(blocks, n_tensors), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, a, grad_scale,
                     zero_grad);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// step >= 1.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
int psdf_adamw_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  if (n <= 0) return PSDF_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return PSDF_ERR_ARG;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return PSDF_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  unsigned blocks = psdf_blocks((n + 3) / 4, PSDF_BLOCK);
  if (blocks > 4096u) blocks = 4096u;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, n, param, grad, exp_avg,
                     exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}

// The same update for up to 64 tensors that share the step count, in ONE launch (host arrays of sizes and device
// pointers; every pointer 16-byte aligned).
int psdf_adamw_step_multi(int n_tensors, const int64_t* sizes, float* const* params, const float* const* grads,
                          float* const* exp_avgs, float* const* exp_avg_sqs, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float grad_scale, void* stream) {
  if (n_tensors <= 0) return PSDF_OK;
  if (n_tensors > ADAMW_MAX_TENSORS || !sizes || !params || !grads || !exp_avgs || !exp_avg_sqs || step < 1)
    return PSDF_ERR_ARG;
  AdamwMulti a;
  int64_t nmax = 0;
  for (int t = 0; t < n_tensors; t++) {
    if (sizes[t] < 0 || !params[t] || !grads[t] || !exp_avgs[t] || !exp_avg_sqs[t]) return PSDF_ERR_ARG;
    if ((((uintptr_t)params[t] | (uintptr_t)grads[t] | (uintptr_t)exp_avgs[t] | (uintptr_t)exp_avg_sqs[t]) & 15) != 0)
      return PSDF_ERR_ARG;
    a.n[t] = sizes[t];
    a.p[t] = params[t];
    a.g[t] = grads[t];
    a.m[t] = exp_avgs[t];
    a.v[t] = exp_avg_sqs[t];
    if (sizes[t] > nmax) nmax = sizes[t];
  }
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  unsigned blocks = psdf_blocks((nmax + 3) / 4, PSDF_BLOCK);
  if (blocks > 64u) blocks = 64u;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks, (unsigned)n_tensors), dim3(PSDF_BLOCK), 0, (hipStream_t)stream, a, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
  PSDF_LAUNCH_CHECK();
  return PSDF_OK;
}
}
