// lt_rollout.hip - fused rollout-side kernels (policy sampling + rollout-storage writes).
//
// In the reference the per-step trainer work between the policy GEMMs and env.step() is ~30 tiny eager torch ops
// (loco_rl/loco_rl/algorithms/ppo.py:129-170: Normal.sample, log_prob, clones, time-out bootstrap;
// loco_rl/loco_rl/storage/rollout_storage.py:79-107: nine copy_ calls).  Each is a launch-bound ~5 us kernel on MI355X;
// fused here into two launches that touch every byte once:
//   lt_rollout_act    : a = mu + sigma * N(0,1), log_prob, and the storage-slot writes (obs, critic obs, actions, mu, sigma,
//                       values, log_prob) - HBM-bound on the 2 x N x obs_dim observation copy.
//   lt_rollout_record : rewards (+ gamma * V * time_out bootstrap, ppo.py:162-165) and dones into the storage slot.
// The GEMMs of the actor / critic MLPs stay in PyTorch-ROCm (hipBLASLt, MFMA).
#include <hip/hip_runtime.h>

#include "lt_device_math.h"
#include "lt_internal.h"

using namespace lt;

namespace {

struct ActArgs {
  long long n;
  int obs_dim;
  unsigned long long seed;
  const long long* step_counter;  // device-resident (captured graphs replay with a fresh step every time)
  const float* mu; const float* std12; const float* value;
  const float* obs; const float* critic_obs;
  float* st_obs; float* st_critic_obs; float* st_actions; float* st_mu; float* st_sigma; float* st_values; float* st_logp;
  float* actions_out;
};

constexpr int ENVS_PER_BLOCK = 8;
constexpr unsigned RS_POLICY = 0x400;  // Philox stream id of the policy noise (env streams live below 0x400)

template <int VEC>
__global__ __launch_bounds__(256) void lt_rollout_act_kernel(const ActArgs a) {
  const long long e0 = (long long)blockIdx.x * ENVS_PER_BLOCK;
  const int tid = threadIdx.x;
  // ---- observation rows -> storage slot (coalesced VEC-wide copies; the block's rows are contiguous) ----
  if (a.obs) {
    const long long nrows = (a.n - e0) < ENVS_PER_BLOCK ? (a.n - e0) : ENVS_PER_BLOCK;
    const long long nvec = nrows * a.obs_dim / VEC;
    const long long base = e0 * a.obs_dim;
    if (VEC == 4) {
      const float4* s0 = (const float4*)(a.obs + base); float4* d0 = (float4*)(a.st_obs + base);
      const float4* s1 = (const float4*)(a.critic_obs + base); float4* d1 = (float4*)(a.st_critic_obs + base);
      for (long long i = tid; i < nvec; i += 256) { d0[i] = s0[i]; d1[i] = s1[i]; }
    } else {
      const float2* s0 = (const float2*)(a.obs + base); float2* d0 = (float2*)(a.st_obs + base);
      const float2* s1 = (const float2*)(a.critic_obs + base); float2* d1 = (float2*)(a.st_critic_obs + base);
      for (long long i = tid; i < nvec; i += 256) { d0[i] = s0[i]; d1[i] = s1[i]; }
    }
  }
  // ---- sampling + log-prob: 16 lanes per env (12 active), butterfly sum inside the 16-lane row ----
  if (tid < ENVS_PER_BLOCK * 16) {
    const int el = tid >> 4, k = tid & 15;
    const long long e = e0 + el;
    const bool act = k < 12 && e < a.n;
    float lp = 0.f;
    if (act) {
      const unsigned long long step = (unsigned long long)a.step_counter[0];
      // one Philox block per (env, group of 4 actions): Box-Muller on (u0,u1) and (u2,u3)
      const U4 u = rng4(a.seed, (unsigned)e, step, RS_POLICY + (k >> 2));
      const int j = k & 3;
      const float ua = (j < 2 ? u.a : u.c), ub = (j < 2 ? u.b : u.d);
      const float r = sqrtf(-2.f * __logf(1.f - ua));  // 1-u in (0,1]: never log(0)
      float sn, cs;
      __sincosf(6.28318530717958647692f * ub, &sn, &cs);
      const float z = r * ((j & 1) ? sn : cs);
      const float mu = a.mu[e * 12 + k], sg = a.std12[k];
      const float x = mu + sg * z;
      a.st_actions[e * 12 + k] = x;
      a.actions_out[e * 12 + k] = x;
      a.st_mu[e * 12 + k] = mu;
      a.st_sigma[e * 12 + k] = sg;
      // Normal.log_prob: -(x-mu)^2 / (2 sigma^2) - log sigma - log sqrt(2 pi)
      lp = -(z * z) * 0.5f - __logf(sg) - 0.91893853320467274178f;
    }
    // sum over the 16-lane row (DPP row shifts via shuffles; 4 steps)
    lp += __shfl_xor(lp, 8, 64); lp += __shfl_xor(lp, 4, 64); lp += __shfl_xor(lp, 2, 64); lp += __shfl_xor(lp, 1, 64);
    if (k == 0 && e < a.n) {
      a.st_logp[e] = lp;
      if (a.value) a.st_values[e] = a.value[e];
    }
  }
}

__global__ void lt_rollout_record_kernel(long long n, float gamma, const float* reward, const long long* dones, const unsigned char* time_out,
                                         const float* values, float* st_rewards, unsigned char* st_dones, float* st_values,
                                         long long* bump_counter) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0 && bump_counter) bump_counter[0] += 1;  // the only writer; every reader is a later launch on the same stream
  if (e >= n) return;
  const float v = values[e];
  if (st_values) st_values[e] = v;
  st_rewards[e] = reward[e] + (time_out[e] ? gamma * v : 0.f);
  st_dones[e] = dones[e] != 0 ? 1 : 0;
}

}  // namespace

extern "C" {

int lt_rollout_act(int64_t n, int obs_dim, uint64_t seed, const int64_t* step_counter, const float* mu, const float* std12,
                   const float* value, const float* obs, const float* critic_obs, float* st_obs, float* st_critic_obs, float* st_actions,
                   float* st_mu, float* st_sigma, float* st_values, float* st_logp, float* actions_out, void* stream) {
  const bool rows_ok = !obs || (critic_obs && st_obs && st_critic_obs);
  const bool value_ok = !value || st_values;
  if (n <= 0 || obs_dim <= 0 || (obs_dim & 1) || !step_counter || !mu || !std12 || !rows_ok || !value_ok || !st_actions || !st_mu ||
      !st_sigma || !st_logp || !actions_out) {
    lt_set_error("lt_rollout_act: invalid argument");
    return LT_EINVAL;
  }
  ActArgs a;
  a.n = n; a.obs_dim = obs_dim; a.seed = seed; a.step_counter = (const long long*)step_counter;
  a.mu = mu; a.std12 = std12; a.value = value; a.obs = obs; a.critic_obs = critic_obs;
  a.st_obs = st_obs; a.st_critic_obs = st_critic_obs; a.st_actions = st_actions; a.st_mu = st_mu; a.st_sigma = st_sigma;
  a.st_values = st_values; a.st_logp = st_logp; a.actions_out = actions_out;
  const dim3 grid((unsigned)((n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK)), block(256);
  if (obs_dim % 4 == 0 && (ENVS_PER_BLOCK * obs_dim) % 4 == 0) hipLaunchKernelGGL(lt_rollout_act_kernel<4>, grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(lt_rollout_act_kernel<2>, grid, block, 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

int lt_rollout_record(int64_t n, float gamma, const float* reward, const int64_t* dones, const uint8_t* time_out, const float* values,
                      float* st_rewards, uint8_t* st_dones, float* st_values, int64_t* bump_counter, void* stream) {
  if (n <= 0 || !reward || !dones || !time_out || !values || !st_rewards || !st_dones) {
    lt_set_error("lt_rollout_record: invalid argument");
    return LT_EINVAL;
  }
  hipLaunchKernelGGL(lt_rollout_record_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n, gamma,
                     reward, (const long long*)dones, time_out, values, st_rewards, st_dones, st_values, (long long*)bump_counter);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

}  // extern "C"
