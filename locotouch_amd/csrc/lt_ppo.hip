// PPO minibatch loss, forward AND gradients, in one launch.
//
// Reference: loco_rl/loco_rl/algorithms/ppo.py:251-311 - Normal log-prob of the stored actions under the current policy, the
// KL to the behaviour policy (for the adaptive learning rate), the clipped surrogate, the clipped value loss and the entropy
// bonus.  In PyTorch ops that is ~45 small launches forward and ~60 backward per minibatch step (232 + ~300 us at 24 576 rows,
// tools/ppo_step_probe.py) on tensors of 12 floats per row; here a thread owns a row, the four means and the gradient of the
// state-independent std are block-reduced and accumulated with one atomic per block, and d loss / d mu, d loss / d value
// are written directly.  `torch.max(a, b)` splits the gradient evenly on ties; inside the clip range both surrogate branches
// (and both value branches) are equal AND have the same derivative, so the closed forms below are exactly autograd's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lt_env.h"
#include "lt_internal.h"

namespace {

constexpr int MAX_A = 16;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// acc layout: [0] sum surrogate, [1] sum value loss, [2] sum kl, [3] unused, [4 .. 4 + A) sum over rows of d surrogate / d sigma_a
__global__ __launch_bounds__(256) void lt_ppo_loss_kernel(const float* __restrict__ mu, const float* __restrict__ stdp, const float* __restrict__ value,
                                                          const float* __restrict__ actions, const float* __restrict__ old_logp,
                                                          const float* __restrict__ adv, const float* __restrict__ returns,
                                                          const float* __restrict__ old_values, const float* __restrict__ old_mu,
                                                          const float* __restrict__ old_sigma, long long M, int A, float clip, float vcoef,
                                                          int clipped_value, float* __restrict__ dmu, float* __restrict__ dvalue,
                                                          float* __restrict__ acc) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = row < M;
  float part[4 + MAX_A];
#pragma unroll
  for (int i = 0; i < 4 + MAX_A; ++i) part[i] = 0.f;
  if (ok) {
    const float inv_m = 1.f / (float)M;
    float logp = 0.f, kl = 0.f;
    float z[MAX_A], isg[MAX_A];
#pragma unroll
    for (int a = 0; a < MAX_A; ++a) {
      if (a < A) {
        const float sg = stdp[a], m = mu[row * A + a], x = actions[row * A + a];
        const float om = old_mu[row * A + a], os = old_sigma[row * A + a];
        isg[a] = 1.f / sg;
        z[a] = (x - m) * isg[a];
        logp += -0.5f * z[a] * z[a] - __logf(sg) - kHalfLog2Pi;
        kl += __logf(sg / os + 1.0e-5f) + (os * os + (om - m) * (om - m)) / (2.f * sg * sg) - 0.5f;
      }
    }
    const float advr = adv[row];
    const float ratio = __expf(logp - old_logp[row]);
    const float s1 = -advr * ratio;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float s2 = -advr * rc;
    const bool inside = ratio >= 1.f - clip && ratio <= 1.f + clip;
    const float dsurr_dratio = (s1 > s2 || inside) ? -advr : 0.f;
    const float dlogp = dsurr_dratio * ratio * inv_m;
#pragma unroll
    for (int a = 0; a < MAX_A; ++a) {
      if (a < A) {
        dmu[row * A + a] = dlogp * z[a] * isg[a];
        part[4 + a] = dlogp * (z[a] * z[a] - 1.f) * isg[a];  // d logp / d sigma_a = ((x - mu)^2 / sigma^3 - 1 / sigma)
      }
    }
    const float v = value[row], R = returns[row];
    float vl, dv;
    if (clipped_value) {
      const float ov = old_values[row];
      const float dcl = fminf(fmaxf(v - ov, -clip), clip);
      const float vc = ov + dcl;
      const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
      const bool in_v = (v - ov) >= -clip && (v - ov) <= clip;
      vl = fmaxf(l1, l2);
      dv = (l1 > l2 || in_v) ? 2.f * (v - R) : 0.f;  // outside the clip range the clipped branch has no derivative
    } else {
      vl = (R - v) * (R - v);
      dv = 2.f * (v - R);
    }
    dvalue[row] = vcoef * dv * inv_m;
    part[0] = fmaxf(s1, s2);
    part[1] = vl;
    part[2] = kl;
  }
  // block reduction (wave butterflies, then 4 partials through LDS), one atomic per block and slot
  __shared__ float red[4][4 + MAX_A];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4 + MAX_A; ++i) {
    float v = part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4 + A && threadIdx.x != 3) {
    const int i = threadIdx.x;
    atomicAdd(acc + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

}  // namespace

extern "C" int lt_ppo_loss(const float* mu, const float* stdp, const float* value, const float* actions, const float* old_logp, const float* adv,
                           const float* returns, const float* old_values, const float* old_mu, const float* old_sigma, int64_t M, int A,
                           float clip, float value_loss_coef, int use_clipped_value_loss, float* dmu, float* dvalue, float* acc, void* stream) {
  if (!mu || !stdp || !value || !actions || !old_logp || !adv || !returns || !old_values || !old_mu || !old_sigma || !dmu || !dvalue || !acc ||
      M < 1 || A < 1 || A > MAX_A) {
    lt_set_error("lt_ppo_loss: invalid argument (1 <= num_actions <= 16)");
    return LT_EINVAL;
  }
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * (4 + MAX_A), (hipStream_t)stream);
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  hipLaunchKernelGGL(lt_ppo_loss_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, stdp, value, actions, old_logp, adv,
                     returns, old_values, old_mu, old_sigma, (long long)M, A, clip, value_loss_coef, use_clipped_value_loss, dmu, dvalue, acc);
  e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}
