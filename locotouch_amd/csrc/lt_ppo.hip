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
#include <math.h>
#include <stdint.h>

#include "lt_env.h"
#include "lt_internal.h"

namespace {

constexpr int MAX_A = 16;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// acc layout: [0] sum surrogate, [1] sum value loss, [2] sum kl, [3] unused, [4 .. 4 + A) sum over rows of d surrogate / d sigma_a,
// [20] max |dmu|, [21] max |dvalue| (bit patterns of non-negative floats, merged with an integer atomicMax: order-independent) - the
// scales the backward chain brings the two gradients into f16's range with (lt_mlp_backward_pair)
__global__ __launch_bounds__(256) void lt_ppo_loss_kernel(const float* __restrict__ mu, const float* __restrict__ stdp, const float* __restrict__ value,
                                                          const float* __restrict__ actions, const float* __restrict__ old_logp,
                                                          const float* __restrict__ adv, const float* __restrict__ returns,
                                                          const float* __restrict__ old_values, const float* __restrict__ old_mu,
                                                          const float* __restrict__ old_sigma, const long long* __restrict__ idx,
                                                          long long M, int A, float clip, float vcoef, int clipped_value,
                                                          float* __restrict__ dmu, float* __restrict__ dvalue, float* __restrict__ acc) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = row < M;
  const long long src = (ok && idx) ? idx[row] : row;  // row of the rollout storage this minibatch row was drawn from
  float part[4 + MAX_A];
  float amax_mu = 0.f, amax_v = 0.f;
#pragma unroll
  for (int i = 0; i < 4 + MAX_A; ++i) part[i] = 0.f;
  if (ok) {
    const float inv_m = 1.f / (float)M;
    float logp = 0.f, kl = 0.f;
    float z[MAX_A], isg[MAX_A];
#pragma unroll
    for (int a = 0; a < MAX_A; ++a) {
      if (a < A) {
        const float sg = stdp[a], m = mu[row * A + a], x = actions[src * A + a];
        const float om = old_mu[src * A + a], os = old_sigma[src * A + a];
        isg[a] = 1.f / sg;
        z[a] = (x - m) * isg[a];
        logp += -0.5f * z[a] * z[a] - __logf(sg) - kHalfLog2Pi;
        kl += __logf(sg / os + 1.0e-5f) + (os * os + (om - m) * (om - m)) / (2.f * sg * sg) - 0.5f;
      }
    }
    const float advr = adv[src];
    const float ratio = __expf(logp - old_logp[src]);
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
        amax_mu = fmaxf(amax_mu, fabsf(dlogp * z[a] * isg[a]));
        part[4 + a] = dlogp * (z[a] * z[a] - 1.f) * isg[a];  // d logp / d sigma_a = ((x - mu)^2 / sigma^3 - 1 / sigma)
      }
    }
    const float v = value[row], R = returns[src];
    float vl, dv;
    if (clipped_value) {
      const float ov = old_values[src];
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
    amax_v = fabsf(vcoef * dv * inv_m);
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
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    amax_mu = fmaxf(amax_mu, __shfl_xor(amax_mu, off, 64));
    amax_v = fmaxf(amax_v, __shfl_xor(amax_v, off, 64));
  }
  __shared__ float redm[4][2];
  if (lane == 0) { redm[wave][0] = amax_mu; redm[wave][1] = amax_v; }
  __syncthreads();
  if (threadIdx.x < 4 + A && threadIdx.x != 3) {
    const int i = threadIdx.x;
    atomicAdd(acc + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  } else if (threadIdx.x >= 64 && threadIdx.x < 66) {  // (another wave: one atomic per block and maximum)
    const int j = threadIdx.x - 64;
    atomicMax((unsigned*)acc + 20 + j, __float_as_uint(fmaxf(fmaxf(redm[0][j], redm[1][j]), fmaxf(redm[2][j], redm[3][j]))));
  }
}

// The scalars of the minibatch loss from the sums of lt_ppo_loss_kernel (one wave, behind it on the stream): out[0] loss, [1] mean
// surrogate, [2] mean value loss, [3] entropy, [4] mean KL, [8 + a] d loss / d sigma_a.  In PyTorch ops this was a dozen launches on
// 12-float tensors.  entropy = sum_a (0.5 + 0.5 log 2 pi + log sigma_a) (the same in every row: the std is state-independent).
__global__ __launch_bounds__(64) void lt_ppo_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ stdp, int A, float inv_m,
                                                             float vcoef, float ecoef, float* __restrict__ out) {
  const int a = threadIdx.x;
  float e = 0.f;
  if (a < A) {
    const float sg = stdp[a];
    e = 0.5f + kHalfLog2Pi + logf(sg);
    out[8 + a] = acc[4 + a] - ecoef / sg;
  }
  for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
  if (a == 0) {
    const float surr = acc[0] * inv_m, vl = acc[1] * inv_m;
    out[0] = surr + vcoef * vl - ecoef * e;
    out[1] = surr; out[2] = vl; out[3] = e; out[4] = acc[2] * inv_m;
  }
}

// ---- ELU backward fused with the bias gradient ---------------------------------------------------------------------------------
// dz = da * elu'(z) with elu'(z) recovered from the OUTPUT a = elu(z): 1 for a > 0, a + alpha otherwise (PyTorch's own
// `elu_backward(..., is_result=true)`), and db[n] = sum over rows of dz[., n] in the same pass.  A thread owns 4 adjacent columns
// (one 16-byte load per operand and row); the 256 / (N / 4) row lanes of a block meet in LDS; per-block column sums go to `ws`
// and a second, tiny launch adds them in a fixed order (deterministic, no float atomics).
constexpr int EB_ROWS = 48;  // rows per block: 512 blocks at 24 576 rows

__global__ __launch_bounds__(256) void lt_elu_bwd_bias_kernel(const float* da, const float* __restrict__ a, long long M, int N, float alpha,
                                                              float* dz, float* __restrict__ ws, float* __restrict__ amax) {
  const int n4 = N >> 2, lanes = 256 / n4;
  const int c4 = threadIdx.x % n4, rl = threadIdx.x / n4;
  const long long r0 = (long long)blockIdx.x * EB_ROWS;
  float4 sum = {0.f, 0.f, 0.f, 0.f};
  float mx = 0.f;  // max |dz| of this block (lt_wgrad scales the gradient into f16's range by it)
  if (rl < lanes) {
    for (int r = rl; r < EB_ROWS; r += lanes) {
      const long long row = r0 + r;
      if (row >= M) break;
      const float4 g = ((const float4*)(da + row * N))[c4], y = ((const float4*)(a + row * N))[c4];
      float4 o;
      o.x = g.x * (y.x > 0.f ? 1.f : y.x + alpha);
      o.y = g.y * (y.y > 0.f ? 1.f : y.y + alpha);
      o.z = g.z * (y.z > 0.f ? 1.f : y.z + alpha);
      o.w = g.w * (y.w > 0.f ? 1.f : y.w + alpha);
      ((float4*)(dz + row * N))[c4] = o;
      sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  __shared__ float4 red[256];
  __shared__ float redm[4];
  red[threadIdx.x] = sum;
  if (amax) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (amax && threadIdx.x == 0) amax[blockIdx.x] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  if (threadIdx.x < n4) {
    float4 t = red[threadIdx.x];
    for (int l = 1; l < lanes; ++l) {
      const float4 u = red[l * n4 + threadIdx.x];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    ((float4*)(ws + (long long)blockIdx.x * N))[threadIdx.x] = t;
  }
}

// out[e] = sum over b < nblk of ws[b * stride + e], e < count, added in block order by 16 row lanes of 16 adjacent elements each
// (a lane's loads are independent, so a block's latency is ~nblk / 16 loads deep, not nblk); elements >= split go to out1.
__global__ __launch_bounds__(256) void lt_partial_sum_kernel(const float* __restrict__ ws, int nblk, long long stride, int count, int split,
                                                             float* __restrict__ out0, float* __restrict__ out1) {
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < count) {
    int b = lane;
    for (; b + 48 < nblk; b += 64) {
      s0 += ws[(long long)b * stride + e];
      s1 += ws[(long long)(b + 16) * stride + e];
      s2 += ws[(long long)(b + 32) * stride + e];
      s3 += ws[(long long)(b + 48) * stride + e];
    }
    for (; b < nblk; b += 16) s0 += ws[(long long)b * stride + e];
  }
  __shared__ float red[16][16];
  red[lane][threadIdx.x & 15] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (lane == 0 && e < count) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][threadIdx.x];
    if (e < split) out0[e] = t;
    else if (out1) out1[e - split] = t;
  }
}

// Several such sums in ONE launch (the backward pass of both stacks leaves ~14 partial buffers - per-block bias sums of
// lt_elu_bwd_bias_kernel / lt_head_wgrad_kernel and the split-K slabs of the weight-gradient GEMMs - whose sums were 14 launches of
// ~5 us): job j owns blocks [first[j], first[j + 1]); a fixed order of additions per element (not lt_partial_sum_kernel's: see the kernel).
constexpr int SUM_MAX_JOBS = 24;
struct SumJobs {
  const float* ws[SUM_MAX_JOBS];
  float* out0[SUM_MAX_JOBS];
  float* out1[SUM_MAX_JOBS];
  long long stride[SUM_MAX_JOBS];
  int nblk[SUM_MAX_JOBS], count[SUM_MAX_JOBS], split[SUM_MAX_JOBS], first[SUM_MAX_JOBS + 1];
  int njobs;
};
// row lanes of a job: 4 for the usual handful of partials, 16 where there are hundreds (the head kernels leave one per 96 rows: with
// 4 lanes their few blocks walked 64 dependent loads each and ended the launch alone - 41 us for 17 us of traffic)
__host__ __device__ inline int sum_row_lanes(int nblk) { return nblk >= 96 ? 16 : 4; }

__global__ __launch_bounds__(256) void lt_partial_sums_kernel(const SumJobs J) {
  // A block owns (256 / RL) * 4 consecutive elements: thread (c, row lane) adds elements 4 c .. 4 c + 3 of the partials b = lane,
  // lane + RL, ... in four interleaved chains - a wave reads whole KiB stretches of a partial per instruction (the first version's 16
  // lanes x 64 B moved 95 MB of slabs at 1.6 TB/s: 60 us of a PPO step).  Order per element: chain by chain, then the row lanes in
  // order - fixed.
  // which job: ONE load of the block-range table by the lanes of a wave + a ballot.  (A scalar loop over J.first[] is a chain of
  // dependent reads of the kernel-argument segment - host-visible memory, a fabric round trip per miss: ~1.5 us x up to 14 per block,
  // 36 us for a launch that moves 48 MB.)
  const int ln = (int)threadIdx.x & 63;
  const int f = (ln >= 1 && ln < J.njobs) ? J.first[ln] : 0x7fffffff;
  const int j = __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(f <= (int)blockIdx.x)));
  const float* __restrict__ ws = J.ws[j];
  const int nblk = J.nblk[j], count = J.count[j], split = J.split[j];
  const long long stride = J.stride[j];
  const int RL = sum_row_lanes(nblk), cols = 256 / RL;
  const int c = (int)threadIdx.x % cols, lane = (int)threadIdx.x / cols;
  const int e = ((int)blockIdx.x - J.first[j]) * cols * 4 + 4 * c;
  const bool vec = e + 3 < count && ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < count) {
    auto at = [&](int b) __attribute__((always_inline)) {
      const float* p = ws + (long long)b * stride + e;
      if (vec) return *(const float4*)p;
      return make_float4(p[0], e + 1 < count ? p[1] : 0.f, e + 2 < count ? p[2] : 0.f, e + 3 < count ? p[3] : 0.f);
    };
    int b = lane;
    for (; b + 3 * RL < nblk; b += 4 * RL) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = at(b + u * RL);
        s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {  // at most three partials are left to this lane (static chain index: a run-time one puts s[] in scratch)
      if (b < nblk) {
        const float4 v = at(b);
        s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
        b += RL;
      }
    }
  }
  __shared__ float4 red[256];
  red[threadIdx.x] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                                 (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  if (lane == 0 && e < count) {
    float4 t = red[c];
    for (int l = 1; l < RL; ++l) { const float4 v = red[l * cols + c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const float o[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ei = e + i;
      if (ei >= count) break;
      if (ei < split) J.out0[j][ei] = o[i];
      else if (J.out1[j]) J.out1[j][ei - split] = o[i];
    }
  }
}

// ---- weight + bias gradient of a narrow head layer -----------------------------------------------------------------------------------
// dW[n][k] = sum_m dy[m][n] x[m][k], db[n] = sum_m dy[m][n] for n <= 16 outputs (the action-mean and value heads: 12 x 128 and
// 1 x 128 over 24 576 rows).  As GEMMs these are all reduction and no tile: hipBLASLt takes 35-52 us for 75 MFLOP, plus a
// column reduction for the bias.  Here a thread owns 4 adjacent k columns and all n outputs (one 16-byte load of x per row, dy
// broadcast), row lanes meet by shuffle + LDS, a block writes ONE partial [n][k] and a second launch adds the partials in order.
constexpr int HW_ROWS = 96, HW_MAX_N = 16;

template <int NN>
__global__ __launch_bounds__(256) void lt_head_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int x_split, long long M, int n, int k,
                                                            float* __restrict__ ws) {
  const int k4 = k >> 2, lanes = 256 / k4;
  const int c4 = threadIdx.x % k4, rl = threadIdx.x / k4;
  const long long r0 = (long long)blockIdx.x * HW_ROWS;
  float4 acc[NN];
  float accb[NN];
#pragma unroll
  for (int j = 0; j < NN; ++j) { acc[j] = {0.f, 0.f, 0.f, 0.f}; accb[j] = 0.f; }
  if (rl < lanes) {
    for (int r = rl; r < HW_ROWS; r += 4 * lanes) {  // 4 rows in flight: their loads are issued before the first FMA
      float4 xv[4];
      float d[4][NN];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long row = r0 + r + u * lanes;
        const bool ok = (r + u * lanes) < HW_ROWS && row < M;
        xv[u] = ok ? ((const float4*)(x + row * k))[c4] : float4{0.f, 0.f, 0.f, 0.f};
        if (x_split) {  // (uniform) split format of the training forward (lt_mlp.hip): dword = f16 hi | f16 lo << 16, x = hi + lo / 64
          const unsigned w[4] = {__float_as_uint(xv[u].x), __float_as_uint(xv[u].y), __float_as_uint(xv[u].z), __float_as_uint(xv[u].w)};
          float f[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            f[i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] & 0xFFFFu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] >> 16)) * (1.f / 64.f);
          xv[u] = float4{f[0], f[1], f[2], f[3]};
        }
#pragma unroll
        for (int j = 0; j < NN; ++j) d[u][j] = (ok && j < n) ? dy[row * n + j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NN; ++j) {
          acc[j].x += d[u][j] * xv[u].x; acc[j].y += d[u][j] * xv[u].y; acc[j].z += d[u][j] * xv[u].z; acc[j].w += d[u][j] * xv[u].w;
          accb[j] += d[u][j];
        }
    }
  }
  // partial of this block: ws[blk][n][k] then [n] bias sums; row lanes are added through LDS in lane order
  extern __shared__ float4 red[];  // [lanes][NN][k4] + bias [lanes][NN] floats behind it
  float* const redb = (float*)(red + (size_t)lanes * NN * k4);
  if (rl < lanes) {
#pragma unroll
    for (int j = 0; j < NN; ++j) red[((size_t)rl * NN + j) * k4 + c4] = acc[j];
    if (c4 == 0)
#pragma unroll
      for (int j = 0; j < NN; ++j) redb[rl * NN + j] = accb[j];
  }
  __syncthreads();
  float* const out = ws + (size_t)blockIdx.x * ((size_t)n * k + HW_MAX_N);
  for (int e = threadIdx.x; e < n * k4; e += 256) {
    const int j = e / k4, c = e - j * k4;
    float4 t = red[(size_t)j * k4 + c];
    for (int l = 1; l < lanes; ++l) {
      const float4 u = red[((size_t)l * NN + j) * k4 + c];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    ((float4*)(out + (size_t)j * k))[c] = t;
  }
  if (threadIdx.x < n) {
    float t = redb[threadIdx.x];
    for (int l = 1; l < lanes; ++l) t += redb[l * NN + threadIdx.x];
    out[(size_t)n * k + threadIdx.x] = t;
  }
}

// ---- GAE(lambda) over a rollout ----------------------------------------------------------------------------------------------------
// RolloutStorage.compute_returns (loco_rl/loco_rl/storage/rollout_storage.py:170-186) walks the T steps backwards with seven small
// tensor ops per step - ~170 launches per iteration.  One thread per env does the same recursion in registers; the T x N arrays are
// read / written one coalesced row per step.  delta = r + (1 - done) gamma V' - V;  A = delta + (1 - done) gamma lambda A';
// returns = A + V;  advantages = returns - V (as the reference forms them, not A itself).
__global__ __launch_bounds__(256) void lt_gae_kernel(const float* __restrict__ rewards, const unsigned char* __restrict__ dones,
                                                     const float* __restrict__ values, const float* __restrict__ last_values, float gamma,
                                                     float lam, int T, long long N, float* __restrict__ returns, float* __restrict__ advantages) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  float next_v = last_values[e], gae = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    const long long o = (long long)t * N + e;
    const float v = values[o], alive = 1.f - (float)dones[o];
    const float delta = rewards[o] + alive * gamma * next_v - v;
    gae = delta + alive * gamma * lam * gae;
    const float ret = gae + v;
    returns[o] = ret;
    advantages[o] = ret - v;
    next_v = v;
  }
}

// ---- gradient-norm clip + Adam on flat buffers ------------------------------------------------------------------------------------
// torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step() (loco_rl/loco_rl/algorithms/ppo.py:318-319) are ~12 multi-tensor
// launches over 17 small tensors; with parameters, gradients and both moments flat they are two: sum of squares per block, then
// every block re-adds the block sums in order (deterministic), scales its gradients and applies the update.  Arithmetic in
// torch's own order: clip coefficient min(1, max_norm / (norm + 1e-6)); exp_avg.lerp_(g, 1 - b1); exp_avg_sq = b2 * v + (1 - b2) g g;
// denom = sqrt(v) / sqrt(1 - b2^t) + eps; p -= lr / (1 - b1^t) * m / denom.
constexpr int AD_PER_BLOCK = 2048;  // elements per block (256 threads x 2 float4)

__global__ __launch_bounds__(256) void lt_sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ ws) {
  const long long base = (long long)blockIdx.x * AD_PER_BLOCK;
  float s = 0.f;
  for (int k = threadIdx.x; k < AD_PER_BLOCK; k += 256) {
    const long long i = base + k;
    if (i < n) s += g[i] * g[i];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void lt_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                      long long n, const float* __restrict__ ws, int nblk, float max_norm, float b1,
                                                      float b2, float eps, float wd, float step_size, float bc2_sqrt, float* __restrict__ norm_out,
                                                      const float* __restrict__ lr_dev, float inv_bc1) {
  __shared__ float s_coef;
  if (lr_dev) step_size = *lr_dev * inv_bc1;  // the learning rate lives on the device (lt_ppo_lr_rule): no host round trip per step
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += ws[b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) {
      const float norm = sqrtf(s);
      const float c = max_norm / (norm + 1.0e-6f);
      s_coef = max_norm > 0.f ? (c < 1.f ? c : 1.f) : 1.f;
      if (blockIdx.x == 0 && norm_out) *norm_out = norm;
    }
  }
  __syncthreads();
  const float coef = s_coef;
  const long long base = (long long)blockIdx.x * AD_PER_BLOCK;
  for (int k = threadIdx.x; k < AD_PER_BLOCK; k += 256) {
    const long long i = base + k;
    if (i >= n) break;
    float gi = g[i] * coef;
    g[i] = gi;  // clip_grad_norm_ leaves the scaled gradients in .grad
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

// The adaptive learning-rate rule of the reference on the device (loco_rl/loco_rl/algorithms/ppo.py:273-281): one lane.
//   kl > 2 desired -> lr = max(lr_min, lr / factor);   0 < kl < desired / 2 -> lr = min(lr_max, lr * factor)
// `kl_mean`: the minibatch's mean KL (lt_ppo_loss's out[4]; all-reduced by the caller in a multi-rank job).  The host read of the KL
// per minibatch step that the rule cost (20 per PPO iteration) becomes one read of *lr per iteration.
// `stats` (optional): running sums of (value loss, surrogate, entropy) = scalars[2], scalars[1], scalars[3] of lt_ppo_loss's `out`.
__global__ __launch_bounds__(64) void lt_ppo_lr_rule_kernel(const float* __restrict__ kl_mean, float desired, float lr_min, float lr_max, float factor,
                                                            float* __restrict__ lr, float* __restrict__ stats, const float* __restrict__ scalars,
                                                            float* __restrict__ dstd_out, int A) {
  if (dstd_out && scalars && (int)threadIdx.x < A) dstd_out[threadIdx.x] = scalars[8 + threadIdx.x];  // d loss / d sigma_a -> the gradient bucket
  if (threadIdx.x != 0) return;
  if (kl_mean && desired > 0.f) {
    const float kl = *kl_mean;
    float v = *lr;
    if (kl > desired * 2.f) v = fmaxf(lr_min, v / factor);
    else if (kl < desired * 0.5f && kl > 0.f) v = fminf(lr_max, v * factor);
    *lr = v;
  }
  if (stats && scalars) { stats[0] += scalars[2]; stats[1] += scalars[1]; stats[2] += scalars[3]; }
}

}  // namespace

extern "C" int lt_ppo_lr_rule(const float* kl_mean, float desired_kl, float lr_min, float lr_max, float factor, float* lr, float* stats,
                              const float* scalars, float* dstd_out, int num_actions, void* stream) {
  if (!lr || factor <= 1.f || num_actions < 0 || num_actions > MAX_A) { lt_set_error("lt_ppo_lr_rule: invalid argument"); return LT_EINVAL; }
  hipLaunchKernelGGL(lt_ppo_lr_rule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kl_mean, desired_kl, lr_min, lr_max, factor, lr, stats, scalars,
                     dstd_out, num_actions);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

static int elu_backward_bias(const float* da, const float* a, int64_t M, int N, float alpha, float* dz, float* db, float* ws, float* amax, void* stream);
extern "C" int lt_elu_backward_bias(const float* da, const float* a, int64_t M, int N, float alpha, float* dz, float* db, float* ws, void* stream) {
  return elu_backward_bias(da, a, M, N, alpha, dz, db, ws, nullptr, stream);
}
// ... and per-block maxima of |dz| (amax_blocks: lt_elu_backward_bias_nblk(M) floats), what lt_wgrad scales dz by
extern "C" int lt_elu_backward_bias2(const float* da, const float* a, int64_t M, int N, float alpha, float* dz, float* db, float* ws, float* amax_blocks,
                                     void* stream) {
  return elu_backward_bias(da, a, M, N, alpha, dz, db, ws, amax_blocks, stream);
}
static int elu_backward_bias(const float* da, const float* a, int64_t M, int N, float alpha, float* dz, float* db, float* ws, float* amax, void* stream) {
  if (!da || !a || !dz || !ws || M < 1 || N < 4 || (N & 3) || N > 1024) {
    lt_set_error("lt_elu_backward_bias: invalid argument (N a multiple of 4, 4 <= N <= 1024)");
    return LT_EINVAL;
  }
  const int nblk = (int)((M + EB_ROWS - 1) / EB_ROWS);
  hipLaunchKernelGGL(lt_elu_bwd_bias_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, da, a, (long long)M, N, alpha, dz, ws, amax);
  if (db) hipLaunchKernelGGL(lt_partial_sum_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, ws, nblk, (long long)N, N, N, db, (float*)nullptr);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

extern "C" int64_t lt_head_wgrad_ws_floats(int64_t M, int n, int k) { return ((M + HW_ROWS - 1) / HW_ROWS) * ((int64_t)n * k + HW_MAX_N); }

extern "C" int lt_head_wgrad(const float* dy, const float* x, int x_split, int64_t M, int n, int k, float* dw, float* db, float* ws, void* stream) {
  if (!dy || !x || !ws || M < 1 || n < 1 || n > HW_MAX_N || k < 4 || (k & 3) || k > 1024) {
    lt_set_error("lt_head_wgrad: invalid argument (1 <= n <= 16, k a multiple of 4, 4 <= k <= 1024)");
    return LT_EINVAL;
  }
  const int nblk = (int)((M + HW_ROWS - 1) / HW_ROWS), k4 = k / 4, lanes = 256 / k4;
  const int nn = n <= 1 ? 1 : (n <= 4 ? 4 : (n <= 8 ? 8 : (n <= 12 ? 12 : 16)));
  const size_t lds = (size_t)lanes * nn * k4 * 16 + (size_t)lanes * nn * 4;
  // (13 .. 16 outputs need 64 KiB + the bias lanes: more than the default dynamic-LDS limit)
  static bool attr_set = false;
  if (!attr_set) {
    attr_set = true;
    (void)hipFuncSetAttribute((const void*)lt_head_wgrad_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  }
  if (lds > 72 * 1024) {
    lt_set_error("lt_head_wgrad: n * k too large for one block's LDS partials");
    return LT_EINVAL;
  }
  const dim3 g((unsigned)nblk), b(256);
  hipStream_t st = (hipStream_t)stream;
  switch (nn) {
    case 1: hipLaunchKernelGGL(lt_head_wgrad_kernel<1>, g, b, lds, st, dy, x, x_split, (long long)M, n, k, ws); break;
    case 4: hipLaunchKernelGGL(lt_head_wgrad_kernel<4>, g, b, lds, st, dy, x, x_split, (long long)M, n, k, ws); break;
    case 8: hipLaunchKernelGGL(lt_head_wgrad_kernel<8>, g, b, lds, st, dy, x, x_split, (long long)M, n, k, ws); break;
    case 12: hipLaunchKernelGGL(lt_head_wgrad_kernel<12>, g, b, lds, st, dy, x, x_split, (long long)M, n, k, ws); break;
    default: hipLaunchKernelGGL(lt_head_wgrad_kernel<16>, g, b, lds, st, dy, x, x_split, (long long)M, n, k, ws); break;
  }
  if (dw) hipLaunchKernelGGL(lt_partial_sum_kernel, dim3((unsigned)((n * k + n + 15) / 16)), b, 0, st, ws, nblk, (long long)n * k + HW_MAX_N, n * k + n, n * k, dw, db);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

// Partial-sum jobs in one launch: out0[j][e] = sum_b ws[j][b * stride[j] + e] for e < split[j], out1[j][e - split[j]] for split[j] <= e <
// count[j] (out1 optional).  The deferred sums of lt_elu_backward_bias (db == NULL: ws holds lt_elu_backward_bias_nblk(M) blocks of N),
// lt_head_wgrad (dw == NULL: lt_head_wgrad_nblk(M) blocks of n * k + 16, split n * k) and of split-K GEMM slabs.
extern "C" int lt_partial_sums(int njobs, const float* const* ws, const int* nblk, const int64_t* stride, const int* count, const int* split,
                               float* const* out0, float* const* out1, void* stream) {
  if (njobs < 1 || njobs > SUM_MAX_JOBS || !ws || !nblk || !stride || !count || !split || !out0 || !out1) {
    lt_set_error("lt_partial_sums: invalid argument (1 <= njobs <= 24)");
    return LT_EINVAL;
  }
  SumJobs J = {};
  J.njobs = njobs;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!ws[j] || !out0[j] || nblk[j] < 1 || count[j] < 1) { lt_set_error("lt_partial_sums: invalid job"); return LT_EINVAL; }
    J.ws[j] = ws[j]; J.out0[j] = out0[j]; J.out1[j] = out1[j]; J.stride[j] = stride[j];
    J.nblk[j] = nblk[j]; J.count[j] = count[j]; J.split[j] = split[j];
    J.first[j] = blocks;
    const int epb = 256 / sum_row_lanes(nblk[j]) * 4;  // elements per block
    blocks += (count[j] + epb - 1) / epb;
  }
  J.first[njobs] = blocks;
  hipLaunchKernelGGL(lt_partial_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}
extern "C" int lt_elu_backward_bias_nblk(int64_t M) { return (int)((M + EB_ROWS - 1) / EB_ROWS); }
extern "C" int lt_head_wgrad_nblk(int64_t M) { return (int)((M + HW_ROWS - 1) / HW_ROWS); }

extern "C" int lt_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float gamma, float lam, int T,
                      int64_t N, float* returns, float* advantages, void* stream) {
  if (!rewards || !dones || !values || !last_values || !returns || !advantages || T < 1 || N < 1) {
    lt_set_error("lt_gae: invalid argument");
    return LT_EINVAL;
  }
  hipLaunchKernelGGL(lt_gae_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, dones, values, last_values, gamma,
                     lam, T, (long long)N, returns, advantages);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

extern "C" int64_t lt_elu_backward_bias_ws_floats(int64_t M, int N) { return ((M + EB_ROWS - 1) / EB_ROWS) * (int64_t)N; }

static int adam_clip_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float max_norm, float lr, const float* lr_dev,
                          float beta1, float beta2, float eps, float weight_decay, int64_t step, float* ws, float* grad_norm, void* stream);

extern "C" int lt_adam_clip_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float max_norm, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int64_t step, float* ws, float* grad_norm, void* stream) {
  return adam_clip_step(params, grads, exp_avg, exp_avg_sq, n, max_norm, lr, nullptr, beta1, beta2, eps, weight_decay, step, ws, grad_norm, stream);
}

extern "C" int lt_adam_clip_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float max_norm, const float* lr_dev,
                                     float beta1, float beta2, float eps, float weight_decay, int64_t step, float* ws, float* grad_norm, void* stream) {
  if (!lr_dev) { lt_set_error("lt_adam_clip_step_dev: null learning-rate pointer"); return LT_EINVAL; }
  return adam_clip_step(params, grads, exp_avg, exp_avg_sq, n, max_norm, 0.f, lr_dev, beta1, beta2, eps, weight_decay, step, ws, grad_norm, stream);
}

static int adam_clip_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float max_norm, float lr, const float* lr_dev,
                          float beta1, float beta2, float eps, float weight_decay, int64_t step, float* ws, float* grad_norm, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !ws || n < 1 || step < 1) {
    lt_set_error("lt_adam_clip_step: invalid argument");
    return LT_EINVAL;
  }
  const int nblk = (int)((n + AD_PER_BLOCK - 1) / AD_PER_BLOCK);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(lt_sumsq_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, grads, (long long)n, ws);
  hipLaunchKernelGGL(lt_adam_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, (long long)n, ws, nblk,
                     max_norm, beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)sqrt(bc2), grad_norm, lr_dev, (float)(1.0 / bc1));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

extern "C" int64_t lt_adam_clip_step_ws_floats(int64_t n) { return (n + AD_PER_BLOCK - 1) / AD_PER_BLOCK; }

extern "C" int lt_ppo_loss(const float* mu, const float* stdp, const float* value, const float* actions, const float* old_logp, const float* adv,
                           const float* returns, const float* old_values, const float* old_mu, const float* old_sigma, const int64_t* idx,
                           int64_t M, int A, float clip, float value_loss_coef, float entropy_coef, int use_clipped_value_loss, float* dmu, float* dvalue,
                           float* acc, float* out, void* stream) {
  if (!mu || !stdp || !value || !actions || !old_logp || !adv || !returns || !old_values || !old_mu || !old_sigma || !dmu || !dvalue || !acc ||
      M < 1 || A < 1 || A > MAX_A) {
    lt_set_error("lt_ppo_loss: invalid argument (1 <= num_actions <= 16)");
    return LT_EINVAL;
  }
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * (4 + MAX_A + 4), (hipStream_t)stream);
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  hipLaunchKernelGGL(lt_ppo_loss_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, stdp, value, actions, old_logp, adv,
                     returns, old_values, old_mu, old_sigma, (const long long*)idx, (long long)M, A, clip, value_loss_coef, use_clipped_value_loss, dmu, dvalue, acc);
  if (out) hipLaunchKernelGGL(lt_ppo_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, stdp, A, 1.f / (float)M, value_loss_coef, entropy_coef, out);
  e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}
