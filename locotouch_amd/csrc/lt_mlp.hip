// lt_mlp.hip - fused fp32 MLP forward (actor / critic inference of the rollout) on the f32-input MFMA.
//
// Reference: the policy evaluation inside the rollout loop, loco_rl/loco_rl/modules/actor_critic.py:113-131
// (`act` -> update_distribution -> self.actor(obs); `evaluate` -> self.critic(obs)) - four Linear layers with ELU between,
// which PyTorch runs as 4 GEMM launches + 3 activation launches per network and step.  Here the whole network is ONE launch:
//
//   * a workgroup (4 waves) owns 16 batch rows; their activations live in LDS ([16][S] floats, S = widest layer + 4 so that
//     the 16 rows of a ds_read_b128 / ds_write_b128 phase fall on distinct banks) and never visit HBM between layers;
//   * out^T = W . act^T on v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain, no reduced precision): the weight
//     tile is the A operand (16 output features x 4 k), the activations are the B operand (4 k x 16 rows), so a lane ends
//     up with 4 CONSECUTIVE output features of ONE row -> the next layer's input is written back with one ds_write_b128;
//   * each wave owns T = N/64 output tiles (independent accumulators -> the 40-cycle dependent MFMA latency never shows);
//   * weights are pre-packed once per policy update (lt_mlp_pack) into the exact per-lane operand order
//     [tile][k-group of 16][lane][4], so every wave-instruction of the weight stream is one fully coalesced 1-KiB
//     global_load_dwordx4 that feeds 4 MFMAs; PF groups are kept in flight per wave (register ring) to cover L2 latency.
//     All workgroups stream the same 1.4 MB, which stays L2-resident.
//   * k inside a 16-group is permuted (lane quarter q takes k = 16g + 4q + i for MFMA i) - a summation-index relabelling
//     applied to both operands - which is what makes both operand fetches 16-byte vectors.
//
// MODE_POLICY adds the sampling epilogue of lt_rollout_act (a = mu + sigma N(0,1), log-prob, storage-slot writes) to the
// last layer, so the actor side of a rollout step is a single launch.
#include <hip/hip_runtime.h>

#include "lt_device_math.h"
#include "lt_internal.h"

using namespace lt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 16;
constexpr unsigned RS_POLICY = 0x400;  // same Philox stream id as lt_rollout_act
constexpr int MODE_FORWARD = 0, MODE_POLICY = 1;

__host__ __device__ inline int pad16(int x) { return (x + 15) & ~15; }

struct MlpArgs {
  int L;
  int dims[LT_MLP_MAX_LAYERS + 1];
  int activation;
  long long woff[LT_MLP_MAX_LAYERS];  // float offsets into `packed`
  long long boff[LT_MLP_MAX_LAYERS];
  int stride;                         // LDS row stride in floats
  const float* packed;
  const float* x;
  long long m;
  float* y;
  // MODE_POLICY
  unsigned long long seed;
  const long long* step_counter;
  long long step_offset;
  const float* std12;
  float* st_actions; float* st_mu; float* st_sigma; float* st_logp; float* actions_out;
};

__device__ __forceinline__ float activate(float x, int kind) {
  if (kind == LT_ACT_ELU) return x > 0.f ? x : expf(x) - 1.f;
  if (kind == LT_ACT_RELU) return x > 0.f ? x : 0.f;
  if (kind == LT_ACT_TANH) return tanhf(x);
  return x;
}

// Tiles per wave of a layer with `ntiles` 16-feature output tiles: the smallest of {1, 2, 4, 8} that covers ntiles with 4 waves.
__host__ __device__ inline int tiles_per_wave(int ntiles) {
  const int per = (ntiles + 3) >> 2;
  return per > 4 ? 8 : (per > 2 ? 4 : (per > 1 ? 2 : 1));
}
// The packed weights hold whole waves' worth of tiles (zero rows pad the last wave), so the inner loops carry no per-tile guards.
__host__ __device__ inline int padded_tiles(int ntiles) {
  const int T = tiles_per_wave(ntiles);
  return (ntiles + T - 1) / T * T;
}

// One layer for this wave: T output tiles starting at tile0 = wave * T, G k-groups.
template <int T, int MODE>
__device__ __forceinline__ void mlp_layer(const MlpArgs& a, int l, bool last, float* s_act, int wave, int lane) {
#ifdef LT_MLP_PF8
  constexpr int PF = T == 8 ? LT_MLP_PF8 : (T == 4 ? 2 * LT_MLP_PF8 : 8);
#else
  constexpr int PF = T == 8 ? 2 : (T == 4 ? 4 : 8);
#endif  // k-groups of weights in flight per wave (16 x dwordx4 per lane)
  const int r = lane & 15, q = lane >> 4;
  const int S = a.stride;
  const int G = pad16(a.dims[l]) / 16;
  const int N = a.dims[l + 1];
  const int ntiles = padded_tiles(pad16(N) / 16);
  const int tile0 = wave * T;
  const bool active = tile0 < ntiles;
  const float4* __restrict__ wp = (const float4*)(a.packed + a.woff[l]) + (long long)tile0 * G * 64 + lane;
  const float* __restrict__ bias = a.packed + a.boff[l];
  const float* const xrow = s_act + r * S + 4 * q;
  f32x4 acc[T];
  if (active) {
    float4 w[PF][T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float4 b = *(const float4*)(bias + 16 * (tile0 + t) + 4 * q);
      acc[t] = f32x4{b.x, b.y, b.z, b.w};
    }
    const int glast = G - 1;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int g = p < glast ? p : glast;
#pragma unroll
      for (int t = 0; t < T; ++t) w[p][t] = wp[(t * G + g) * 64];
    }
    float4 x = *(const float4*)xrow;
    const int gmain = G - G % PF;
    for (int g0 = 0; g0 < gmain; g0 += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int g = g0 + p;
        const int gx = g + 1 < glast ? g + 1 : glast;
#ifdef LT_MLP_EXP_NOLDS
        const float4 xn = x;
#else
        const float4 xn = *(const float4*)(xrow + 16 * gx);
#endif
#ifdef LT_MLP_EXP_NOMFMA
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t][0] += w[p][t].x * x.x + w[p][t].y * x.y + w[p][t].z * x.z + w[p][t].w * x.w;
#else
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].x, x.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].y, x.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].z, x.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].w, x.w, acc[t], 0, 0, 0);
#endif
#ifdef LT_MLP_EXP_SAMEGROUP
        const int gn = 0;
#else
        const int gn = g + PF < glast ? g + PF : glast;  // past the end: a redundant re-load instead of a branch
#endif
#ifndef LT_MLP_EXP_NOLOAD
#pragma unroll
        for (int t = 0; t < T; ++t) w[p][t] = wp[(t * G + gn) * 64];
#endif
        // keep the refill HERE: left alone, the scheduler sinks it behind the next slot's MFMAs to save registers, which
        // collapses the ring to one group in flight
        __builtin_amdgcn_sched_barrier(0);
        x = xn;
      }
    }
    // tail (G not a multiple of PF): ring slots 0.. hold groups gmain..
#pragma unroll
    for (int p = 0; p < PF - 1; ++p) {
      const int g = gmain + p;
      if (g < G) {
        const int gx = g + 1 < glast ? g + 1 : glast;
        const float4 xn = *(const float4*)(xrow + 16 * gx);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].x, x.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].y, x.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].z, x.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p][t].w, x.w, acc[t], 0, 0, 0);
        x = xn;
      }
    }
  }
  __syncthreads();  // every wave is done reading this layer's input
  const long long e = (long long)blockIdx.x * ROWS + r;
  if (!last) {
    if (active) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float4 o;
        o.x = activate(acc[t][0], a.activation); o.y = activate(acc[t][1], a.activation);
        o.z = activate(acc[t][2], a.activation); o.w = activate(acc[t][3], a.activation);
        if (16 * (tile0 + t) < pad16(N)) *(float4*)(s_act + r * S + 16 * (tile0 + t) + 4 * q) = o;  // zero-pad tiles stay out of LDS
      }
    }
    __syncthreads();
    return;
  }
  if (MODE == MODE_FORWARD) {
    if (active && e < a.m) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = 16 * (tile0 + t) + 4 * q + i;
          if (n < N) a.y[e * N + n] = acc[t][i];
        }
      }
    }
  } else {
    // policy head: N == 12 -> one tile, held by wave 0; lane (r, q) owns actions 4q..4q+3 of env e (q == 3: padding)
    if (wave == 0) {
      float lp = 0.f;
      if (q < 3 && e < a.m) {
        const unsigned long long step = (unsigned long long)(a.step_counter[0] + a.step_offset);
        const U4 u = rng4(a.seed, (unsigned)e, step, RS_POLICY + q);
        const float ra = sqrtf(-2.f * __logf(1.f - u.a)), rb = sqrtf(-2.f * __logf(1.f - u.c));  // 1-u in (0,1]: never log(0)
        float sa, ca, sb, cb;
        __sincosf(6.28318530717958647692f * u.b, &sa, &ca);
        __sincosf(6.28318530717958647692f * u.d, &sb, &cb);
        const float z[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
        const float4 sg = *(const float4*)(a.std12 + 4 * q);
        const float sgv[4] = {sg.x, sg.y, sg.z, sg.w};
        float xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xv[i] = acc[0][i] + sgv[i] * z[i];
          lp += -(z[i] * z[i]) * 0.5f - __logf(sgv[i]) - 0.91893853320467274178f;  // Normal.log_prob
        }
        const long long o = e * 12 + 4 * q;
        const float4 xa = make_float4(xv[0], xv[1], xv[2], xv[3]);
        *(float4*)(a.st_actions + o) = xa;
        *(float4*)(a.actions_out + o) = xa;
        *(float4*)(a.st_mu + o) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
        *(float4*)(a.st_sigma + o) = sg;
      }
      lp += __shfl_xor(lp, 16, 64);
      lp += __shfl_xor(lp, 32, 64);
      if (q == 0 && e < a.m) a.st_logp[e] = lp;
    }
  }
}

#ifdef LT_MLP_STAMPS
__device__ unsigned long long g_mlp_stamps[1024 * 8];
#define MLP_STAMP(i) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_mlp_stamps[blockIdx.x * 8 + (i)] = t_; } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif

template <int MODE>
__global__ __launch_bounds__(256) void lt_mlp_kernel(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_act[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  MLP_STAMP(0);
  const long long row0 = (long long)blockIdx.x * ROWS;
  const int S = a.stride;
  {
    const int K0 = a.dims[0], K0p = pad16(K0);
    if ((K0 & 3) == 0) {
      const int kv = K0p >> 2, k4 = K0 >> 2;
      for (int idx = tid; idx < ROWS * kv; idx += 256) {
        const int rr = idx / kv, cc = idx - rr * kv;
        const long long e = row0 + rr;
        const float4 v = (cc < k4 && e < a.m) ? *(const float4*)(a.x + e * K0 + 4 * cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(s_act + rr * S + 4 * cc) = v;
      }
    } else {
      for (int idx = tid; idx < ROWS * K0p; idx += 256) {
        const int rr = idx / K0p, cc = idx - rr * K0p;
        const long long e = row0 + rr;
        s_act[rr * S + cc] = (cc < K0 && e < a.m) ? a.x[e * K0 + cc] : 0.f;
      }
    }
  }
  __syncthreads();
  MLP_STAMP(1);
  for (int l = 0; l < a.L; ++l) {
    const int T = tiles_per_wave(pad16(a.dims[l + 1]) / 16);
    const bool last = l == a.L - 1;
    if (T == 8) mlp_layer<8, MODE>(a, l, last, s_act, wave, lane);
    else if (T == 4) mlp_layer<4, MODE>(a, l, last, s_act, wave, lane);
    else if (T == 2) mlp_layer<2, MODE>(a, l, last, s_act, wave, lane);
    else mlp_layer<1, MODE>(a, l, last, s_act, wave, lane);
    MLP_STAMP(2 + l);
  }
}

// weights [N][K] (torch.nn.Linear layout) + bias [N] -> packed operand order, zero padded
__global__ void lt_mlp_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, int K, int N, float* __restrict__ wdst,
                                   float* __restrict__ bdst) {
  const int G = pad16(K) / 16, ntiles = padded_tiles(pad16(N) / 16);
  const long long total = (long long)ntiles * G * 64;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) {
    const int lane = (int)(idx & 63);
    const long long tg = idx >> 6;
    const int g = (int)(tg % G), tile = (int)(tg / G);
    const int n = 16 * tile + (lane & 15), k0 = 16 * g + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (n < N && k0 + i < K) ? w[(long long)n * K + k0 + i] : 0.f;
    *(float4*)(wdst + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (idx < ntiles * 16) bdst[idx] = idx < N ? b[idx] : 0.f;
}

bool desc_ok(const lt_mlp_desc* d) {
  if (!d || d->num_layers < 1 || d->num_layers > LT_MLP_MAX_LAYERS) return false;
  if (d->activation < LT_ACT_NONE || d->activation > LT_ACT_TANH) return false;
  for (int l = 0; l <= d->num_layers; ++l)
    if (d->dims[l] < 1 || d->dims[l] > LT_MLP_MAX_WIDTH) return false;
  for (int l = 1; l <= d->num_layers; ++l)
    if (d->dims[l] > 512) return false;  // 8 output tiles per wave at most
  return true;
}

void fill_args(const lt_mlp_desc* d, MlpArgs& a) {
  a.L = d->num_layers;
  a.activation = d->activation;
  long long off = 0;
  int widest = 0;
  for (int l = 0; l <= d->num_layers; ++l) {
    a.dims[l] = d->dims[l];
    widest = pad16(d->dims[l]) > widest ? pad16(d->dims[l]) : widest;
  }
  for (int l = 0; l < d->num_layers; ++l) {
    a.woff[l] = off;
    const int np = 16 * padded_tiles(pad16(d->dims[l + 1]) / 16);
    off += (long long)pad16(d->dims[l]) * np;
    a.boff[l] = off;
    off += np;
  }
  a.stride = widest + 4;
}

int launch(const lt_mlp_desc* d, MlpArgs& a, int mode, hipStream_t s) {
  size_t lds = (size_t)ROWS * a.stride * sizeof(float);
#ifdef LT_MLP_EXP_LDS
  lds = LT_MLP_EXP_LDS;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<MODE_POLICY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<MODE_FORWARD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
#endif
  const dim3 grid((unsigned)((a.m + ROWS - 1) / ROWS)), block(256);
  if (mode == MODE_POLICY) hipLaunchKernelGGL(lt_mlp_kernel<MODE_POLICY>, grid, block, lds, s, a);
  else hipLaunchKernelGGL(lt_mlp_kernel<MODE_FORWARD>, grid, block, lds, s, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

}  // namespace

extern "C" {

#ifdef LT_MLP_STAMPS
int lt_debug_mlp_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mlp_stamps), sizeof(unsigned long long) * 1024 * 8);
}
#endif

int lt_mlp_packed_floats(const lt_mlp_desc* desc, size_t* floats) {
  if (!desc_ok(desc) || !floats) { lt_set_error("lt_mlp_packed_floats: unsupported network shape"); return LT_EINVAL; }
  MlpArgs a;
  fill_args(desc, a);
  const int L = desc->num_layers;
  *floats = (size_t)(a.boff[L - 1] + 16 * padded_tiles(pad16(desc->dims[L]) / 16));
  return LT_OK;
}

int lt_mlp_pack(const lt_mlp_desc* desc, const float* const* weights, const float* const* biases, float* packed, void* stream) {
  if (!desc_ok(desc) || !weights || !biases || !packed) { lt_set_error("lt_mlp_pack: invalid argument"); return LT_EINVAL; }
  MlpArgs a;
  fill_args(desc, a);
  for (int l = 0; l < desc->num_layers; ++l) {
    if (!weights[l] || !biases[l]) { lt_set_error("lt_mlp_pack: null layer pointer"); return LT_EINVAL; }
    const int K = desc->dims[l], N = desc->dims[l + 1];
    const long long total = (long long)pad16(K) / 16 * padded_tiles(pad16(N) / 16) * 64;
    hipLaunchKernelGGL(lt_mlp_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weights[l], biases[l], K, N,
                       packed + a.woff[l], packed + a.boff[l]);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  }
  return LT_OK;
}

int lt_mlp_forward(const lt_mlp_desc* desc, const float* packed, const float* x, int64_t m, float* y, void* stream) {
  if (!desc_ok(desc) || !packed || !x || !y || m <= 0) { lt_set_error("lt_mlp_forward: invalid argument"); return LT_EINVAL; }
  MlpArgs a = {};
  fill_args(desc, a);
  a.packed = packed; a.x = x; a.m = m; a.y = y;
  return launch(desc, a, MODE_FORWARD, (hipStream_t)stream);
}

int lt_rollout_policy(const lt_mlp_desc* actor, const float* packed, const float* obs, int64_t n, uint64_t seed, const int64_t* step_counter,
                      int64_t step_offset, const float* std12, float* st_actions, float* st_mu, float* st_sigma, float* st_logp,
                      float* actions_out, void* stream) {
  if (!desc_ok(actor) || actor->dims[actor->num_layers] != 12 || !packed || !obs || n <= 0 || !step_counter || !std12 || !st_actions ||
      !st_mu || !st_sigma || !st_logp || !actions_out) {
    lt_set_error("lt_rollout_policy: invalid argument (the policy head must have 12 outputs)");
    return LT_EINVAL;
  }
  MlpArgs a = {};
  fill_args(actor, a);
  a.packed = packed; a.x = obs; a.m = n; a.y = nullptr;
  a.seed = seed; a.step_counter = (const long long*)step_counter; a.step_offset = step_offset; a.std12 = std12;
  a.st_actions = st_actions; a.st_mu = st_mu; a.st_sigma = st_sigma; a.st_logp = st_logp; a.actions_out = actions_out;
  return launch(actor, a, MODE_POLICY, (hipStream_t)stream);
}

}  // extern "C"
