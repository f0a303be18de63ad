// lt_mlp.hip - fused fp32 MLP forward (actor / critic inference of the rollout) on the f32-input MFMA.
//
// Reference: the policy evaluation inside the rollout loop, loco_rl/loco_rl/modules/actor_critic.py:113-131
// (`act` -> update_distribution -> self.actor(obs); `evaluate` -> self.critic(obs)) - four Linear layers with ELU between,
// which PyTorch runs as 4 GEMM launches + 3 activation launches per network and step.  Here a whole network - or the actor
// and the critic side by side - is ONE launch:
//
//   * a workgroup (NW = 4 waves, one per SIMD) owns 16 * RT batch rows; their activations live in LDS ([16][S] floats, S = widest layer + 4 so that
//     the 16 rows of a ds_read_b128 / ds_write_b128 phase fall on distinct banks) and never visit HBM between layers;
//   * out^T = W . act^T on v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain, no reduced precision): the weight
//     tile is the A operand (16 output features x 4 k), the activations are the B operand (4 k x 16 rows), so a lane ends
//     up with 4 CONSECUTIVE output features of ONE row -> the next layer's input is written back with one ds_write_b128;
//   * each wave owns T = N/(16 NW) output tiles (independent accumulators -> the 40-cycle dependent MFMA latency never shows);
//   * k inside a 16-group is permuted (lane quarter q takes k = 16g + 4q + i for MFMA i) - a summation-index relabelling
//     applied to both operands - which is what makes both operand fetches 16-byte vectors;
//   * the parameters are pre-packed once per policy update (lt_mlp_pack) into ONE LINEAR STREAM PER WAVE of 1-KiB chunks
//     (64 lanes x float4) in exactly the order the wave consumes them, across layers: [bias chunks of layer 0][weight
//     chunks g-major, tile-minor][pad to 16]...[layer 1]...  The kernel keeps a 16-chunk register ring per wave and refills
//     a slot right after its MFMAs, so 16 KiB per wave are always in flight and the first weights of layer l+1 are already
//     on their way while layer l finishes - no pipeline restart at layer boundaries.  Every wave-instruction of the stream
//     is one fully coalesced global_load_dwordx4; all workgroups stream the same ~1.4 MB, which stays L2-resident.
//
// The policy network's last layer carries the sampling epilogue of lt_rollout_act (a = mu + sigma N(0,1), log-prob,
// storage-slot writes), so the actor side of a rollout step needs no further launch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "lt_device_math.h"
#include "lt_internal.h"

using namespace lt;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RING = 16;               // chunks in flight per wave
#ifndef LT_MLP_WAVES
#define LT_MLP_WAVES 4
#endif
constexpr int NW = LT_MLP_WAVES;       // waves per workgroup (4 or 8).  Measured: 8 (two per SIMD) buys nothing - the waves of a workgroup
                                       // run in lockstep, so their epilogues and barriers coincide instead of hiding behind MFMAs
constexpr unsigned RS_POLICY = 0x400;  // same Philox stream id as lt_rollout_act
constexpr int MODE_FORWARD = 0, MODE_POLICY = 1;

__host__ __device__ inline int pad16(int x) { return (x + 15) & ~15; }
// Tiles per wave of a layer with `ntiles` 16-feature output tiles: the smallest of {1, 2, 4, 8} that covers ntiles with NW waves.
__host__ __device__ inline int tiles_per_wave(int ntiles) {
  const int per = (ntiles + NW - 1) / NW;
  return per > 4 ? 8 : (per > 2 ? 4 : (per > 1 ? 2 : 1));
}
// chunks of one layer in the stream of an ACTIVE wave: T bias chunks + G*T weight chunks, padded to whole rings
__host__ __device__ inline int layer_chunks(int K, int N) {
  const int T = tiles_per_wave(pad16(N) / 16), G = pad16(K) / 16;
  return (T * (G + 1) + RING - 1) / RING * RING;
}
__host__ __device__ inline int active_waves(int N) {
  const int nt = pad16(N) / 16, T = tiles_per_wave(nt);
  return (nt + T - 1) / T;
}

struct MlpArgs {
  int L;
  int dims[LT_MLP_MAX_LAYERS + 1];
  int activation;
  int mode;
  int stride;                 // LDS row stride in floats
  long long wave_base[NW];    // chunk offset of each wave's stream inside `packed`
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
struct DualArgs {
  MlpArgs net[2];
  int split;  // blocks [0, split) run net[0], the rest net[1]
};

template <int KIND>
__device__ __forceinline__ float activate(float x) {
  if (KIND == LT_ACT_ELU) return x > 0.f ? x : __expf(x) - 1.f;  // |abs err| ~1e-7 (v_exp_f32), far inside the parity tolerance
  if (KIND == LT_ACT_RELU) return x > 0.f ? x : 0.f;
  if (KIND == LT_ACT_TANH) return tanhf(x);
  return x;
}
// activation + write-back of a wave's T output tiles as the next layer's input (KIND is a compile-time constant per call
// site: a runtime `kind` inside the loop gets if-converted into computing EVERY activation for every element)
template <int KIND, int T>
__device__ __forceinline__ void write_activated(const f32x4 (&acc)[T], float* dst, int tile0, int npad) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float4 o;
    o.x = activate<KIND>(acc[t][0]); o.y = activate<KIND>(acc[t][1]); o.z = activate<KIND>(acc[t][2]); o.w = activate<KIND>(acc[t][3]);
    if (16 * (tile0 + t) < npad) *(float4*)(dst + 16 * t) = o;  // zero-pad tiles stay out of LDS
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the 16 weight
// chunks every wave keeps in flight - twice per layer - and undo the streaming across layer boundaries.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef LT_MLP_STAMPS
__device__ unsigned long long g_mlp_stamps[1024 * 8 * NW];
#define MLP_STAMP(i) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_mlp_stamps[blockIdx.x * 8 * NW + (threadIdx.x >> 6) * 8 + (i)] = t_; } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif

// One layer for this wave.  `ring` slot s holds chunk c0 + s of the wave's stream; the layer consumes its chunks in order
// (item 0 = the T bias chunks, item i = k-group i-1) and leaves the ring positioned on the next layer's first chunk.
// RT row tiles (16 rows each) share every weight chunk: RT x the MFMA work per byte streamed from L2.
template <int T, int RT>
__device__ __forceinline__ void mlp_layer(const MlpArgs& a, int l, bool last, float* s_act, int wave, int lane, long long row_block,
                                          float4 (&ring)[RING], const float4* __restrict__ stream, long long& c0) {
  constexpr int R = RING / T;  // items per ring round
  const int r = lane & 15, q = lane >> 4;
  const int S = a.stride;
  const int G = pad16(a.dims[l]) / 16;
  const int N = a.dims[l + 1];
  const int tile0 = wave * T;
  const bool active = wave < active_waves(N);
  const float* const xrow = s_act + r * S + 4 * q;  // row tile rt: + 16 * rt * S
  f32x4 acc[RT][T];
  if (active) {
    float4 xa[RT], xb[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { xa[rt] = *(const float4*)(xrow + 16 * rt * S); xb[rt] = xa[rt]; }
    for (int i0 = 0; i0 <= G; i0 += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int i = i0 + j;  // item: 0 = bias, 1..G = k-group i-1
        // the activations of the NEXT item are fetched before this item's MFMAs (two alternating register sets)
        const int gx = i < G ? i : G - 1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if ((j & 1) == 0) xb[rt] = *(const float4*)(xrow + 16 * rt * S + 16 * gx);
          else xa[rt] = *(const float4*)(xrow + 16 * rt * S + 16 * gx);
        }
        if (i == 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[rt][t] = f32x4{ring[j * T + t].x, ring[j * T + t].y, ring[j * T + t].z, ring[j * T + t].w};
        } else if (i <= G) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const float4 x = (j & 1) == 0 ? xa[rt] : xb[rt];
#pragma unroll
            for (int t = 0; t < T; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[j * T + t].x, x.x, acc[rt][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[j * T + t].y, x.y, acc[rt][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[j * T + t].z, x.z, acc[rt][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[j * T + t].w, x.w, acc[rt][t], 0, 0, 0);
          }
        }
        // refill the slots just consumed (pad chunks of a partial last round included: the ring invariant must hold)
#pragma unroll
        for (int t = 0; t < T; ++t) ring[j * T + t] = stream[(c0 + RING + j * T + t) * 64];
        // keep the refill HERE: left alone, the scheduler sinks it behind the next item's MFMAs to save registers, which
        // collapses the ring to one group in flight
        __builtin_amdgcn_sched_barrier(0);
      }
      c0 += RING;
#ifdef LT_MLP_STAMPS
      if (l == 0 && i0 == 0) MLP_STAMP(7);
#endif
    }
  }
#ifdef LT_MLP_STAMPS
  if (l == 0) MLP_STAMP(6);
#endif
  lds_barrier();  // every wave is done reading this layer's input
  if (!last) {
    if (active) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float* const dst = s_act + (r + 16 * rt) * S + 16 * tile0 + 4 * q;
        if (a.activation == LT_ACT_ELU) write_activated<LT_ACT_ELU, T>(acc[rt], dst, tile0, pad16(N));
        else if (a.activation == LT_ACT_RELU) write_activated<LT_ACT_RELU, T>(acc[rt], dst, tile0, pad16(N));
        else if (a.activation == LT_ACT_TANH) write_activated<LT_ACT_TANH, T>(acc[rt], dst, tile0, pad16(N));
        else write_activated<LT_ACT_NONE, T>(acc[rt], dst, tile0, pad16(N));
      }
    }
    lds_barrier();
    return;
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const long long e = row_block * (16 * RT) + 16 * rt + r;
    if (a.mode == MODE_FORWARD) {
      if (active && e < a.m) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = 16 * (tile0 + t) + 4 * q + i;
            if (n < N) a.y[e * N + n] = acc[rt][t][i];
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
            xv[i] = acc[rt][0][i] + sgv[i] * z[i];
            lp += -(z[i] * z[i]) * 0.5f - __logf(sgv[i]) - 0.91893853320467274178f;  // Normal.log_prob
          }
          const long long o = e * 12 + 4 * q;
          const float4 xo = make_float4(xv[0], xv[1], xv[2], xv[3]);
          *(float4*)(a.st_actions + o) = xo;
          *(float4*)(a.actions_out + o) = xo;
          *(float4*)(a.st_mu + o) = make_float4(acc[rt][0][0], acc[rt][0][1], acc[rt][0][2], acc[rt][0][3]);
          *(float4*)(a.st_sigma + o) = sg;
        }
        lp += __shfl_xor(lp, 16, 64);
        lp += __shfl_xor(lp, 32, 64);
        if (q == 0 && e < a.m) a.st_logp[e] = lp;
      }
    }
  }
}

template <int RT>
__global__ __launch_bounds__(64 * NW) void lt_mlp_kernel(const DualArgs d) {
  extern __shared__ __attribute__((aligned(16))) float s_act[];
  constexpr int ROWS = 16 * RT;
  const bool second = (int)blockIdx.x >= d.split;
  const MlpArgs& a = second ? d.net[1] : d.net[0];
  const long long row_block = second ? (long long)blockIdx.x - d.split : (long long)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  MLP_STAMP(0);
  // weight stream: 16 chunks in flight before anything else
  const float4* __restrict__ stream = (const float4*)a.packed + a.wave_base[wave] * 64 + lane;
  float4 ring[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) ring[s] = stream[s * 64];
  long long c0 = 0;
  const long long row0 = row_block * ROWS;
  const int S = a.stride;
  {
    const int K0 = a.dims[0], K0p = pad16(K0);
    if ((K0 & 3) == 0) {
      const int kv = K0p >> 2, k4 = K0 >> 2;
      for (int idx = tid; idx < ROWS * kv; idx += 64 * NW) {
        const int rr = idx / kv, cc = idx - rr * kv;
        const long long e = row0 + rr;
        const float4 v = (cc < k4 && e < a.m) ? *(const float4*)(a.x + e * K0 + 4 * cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(s_act + rr * S + 4 * cc) = v;
      }
    } else {
      for (int idx = tid; idx < ROWS * K0p; idx += 64 * NW) {
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
    if (T == 8) mlp_layer<8, RT>(a, l, last, s_act, wave, lane, row_block, ring, stream, c0);
    else if (T == 4) mlp_layer<4, RT>(a, l, last, s_act, wave, lane, row_block, ring, stream, c0);
    else if (T == 2) mlp_layer<2, RT>(a, l, last, s_act, wave, lane, row_block, ring, stream, c0);
    else mlp_layer<1, RT>(a, l, last, s_act, wave, lane, row_block, ring, stream, c0);
    MLP_STAMP(2 + l);
  }
}

// One layer of one network: weights [N][K] (torch.nn.Linear layout) + bias [N] -> the per-wave chunk streams.
struct PackArgs {
  const float* w; const float* b;
  int K, N;
  long long chunk_off[NW];  // first chunk of this layer in each wave's stream (absolute, in chunks)
  float* packed;
};
__global__ void lt_mlp_pack_kernel(const PackArgs p) {
  const int T = tiles_per_wave(pad16(p.N) / 16), G = pad16(p.K) / 16;
  const int chunks = layer_chunks(p.K, p.N), nact = active_waves(p.N);
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (wave, chunk, lane)
  if (idx >= (long long)nact * chunks * 64) return;
  const int lane = (int)(idx & 63);
  const int c = (int)((idx >> 6) % chunks), wv = (int)((idx >> 6) / chunks);
  const int r = lane & 15, q = lane >> 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < T) {  // bias chunk of tile wv*T + c: lane (r, q) starts its accumulator with features 4q..4q+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * (wv * T + c) + 4 * q + i;
      v[i] = n < p.N ? p.b[n] : 0.f;
    }
  } else if (c < T * (G + 1)) {
    const int cc = c - T, g = cc / T, t = cc - g * T;
    const int n = 16 * (wv * T + t) + r, k0 = 16 * g + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (n < p.N && k0 + i < p.K) ? p.w[(long long)n * p.K + k0 + i] : 0.f;
  }
  *(float4*)(p.packed + ((p.chunk_off[wv] + c) * 64 + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

bool desc_ok(const lt_mlp_desc* d) {
  if (!d || d->num_layers < 1 || d->num_layers > LT_MLP_MAX_LAYERS) return false;
  if (d->activation < LT_ACT_NONE || d->activation > LT_ACT_TANH) return false;
  for (int l = 0; l <= d->num_layers; ++l)
    if (d->dims[l] < 1 || d->dims[l] > LT_MLP_MAX_WIDTH) return false;
  for (int l = 1; l <= d->num_layers; ++l)
    if (d->dims[l] > 512) return false;  // 8 output tiles per wave at most (NW = 4)
  return true;
}

// stream geometry: per-wave totals (with one ring of zero chunks behind each stream: the refill runs one ring ahead)
struct Geometry {
  long long wave_base[NW];
  long long layer_off[LT_MLP_MAX_LAYERS][NW];
  long long total_chunks;
};
Geometry geometry(const lt_mlp_desc* d) {
  Geometry g;
  long long base = 0;
  for (int w = 0; w < NW; ++w) {
    g.wave_base[w] = base;
    long long off = 0;
    for (int l = 0; l < d->num_layers; ++l) {
      g.layer_off[l][w] = base + off;
      if (w < active_waves(d->dims[l + 1])) off += layer_chunks(d->dims[l], d->dims[l + 1]);
    }
    base += off + RING;
  }
  g.total_chunks = base;
  return g;
}

void fill_args(const lt_mlp_desc* d, MlpArgs& a) {
  a.L = d->num_layers;
  a.activation = d->activation;
  int widest = 0;
  for (int l = 0; l <= d->num_layers; ++l) {
    a.dims[l] = d->dims[l];
    widest = pad16(d->dims[l]) > widest ? pad16(d->dims[l]) : widest;
  }
  const Geometry g = geometry(d);
  for (int w = 0; w < NW; ++w) a.wave_base[w] = g.wave_base[w];
  a.stride = widest + 4;
}

// Row tiles per workgroup: the most (of 1, 2, 4) that still leaves every CU a workgroup - each doubling halves the bytes
// streamed from L2 per FLOP, and at one 16-row tile per CU the kernel is bound by L2 -> CU bandwidth, not by the MFMA rate.
int pick_row_tiles(long long rows_total_blocks16) {
  int rt = 1;
  while (rt < 4 && rows_total_blocks16 / (2 * rt) >= 256) rt *= 2;
  return rt;
}

int launch(DualArgs& d, int nets, hipStream_t s) {
  int stride = d.net[0].stride;
  if (nets == 2 && d.net[1].stride > stride) stride = d.net[1].stride;
  const long long t0 = (d.net[0].m + 15) / 16, t1 = nets == 2 ? (d.net[1].m + 15) / 16 : 0;
  int rt = pick_row_tiles(t0 + t1);
  if (const char* o = getenv("LT_MLP_ROW_TILES")) rt = atoi(o) == 4 ? 4 : (atoi(o) == 2 ? 2 : 1);  // diagnostic override
  while (rt > 1 && (size_t)16 * rt * stride * sizeof(float) > 160 * 1024) rt /= 2;  // one workgroup's activations must fit the LDS
  const size_t lds = (size_t)16 * rt * stride * sizeof(float);
  const long long b0 = (t0 + rt - 1) / rt, b1 = (t1 + rt - 1) / rt;
  d.split = (int)b0;
  const dim3 grid((unsigned)(b0 + b1)), block(64 * NW);
  static bool attr_set = false;
  if (!attr_set) {  // more than the default 64 KB of dynamic LDS
    attr_set = true;
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (rt == 4) hipLaunchKernelGGL(lt_mlp_kernel<4>, grid, block, lds, s, d);
  else if (rt == 2) hipLaunchKernelGGL(lt_mlp_kernel<2>, grid, block, lds, s, d);
  else hipLaunchKernelGGL(lt_mlp_kernel<1>, grid, block, lds, s, d);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

bool policy_args_ok(const lt_mlp_desc* actor, const float* packed, const float* obs, int64_t n, const int64_t* step_counter, const float* std12,
                    float* st_actions, float* st_mu, float* st_sigma, float* st_logp, float* actions_out) {
  return desc_ok(actor) && actor->dims[actor->num_layers] == 12 && packed && obs && n > 0 && step_counter && std12 && st_actions && st_mu &&
         st_sigma && st_logp && actions_out;
}

void fill_policy(const lt_mlp_desc* actor, const float* packed, const float* obs, int64_t n, uint64_t seed, const int64_t* step_counter,
                 int64_t step_offset, const float* std12, float* st_actions, float* st_mu, float* st_sigma, float* st_logp, float* actions_out,
                 MlpArgs& a) {
  fill_args(actor, a);
  a.mode = MODE_POLICY;
  a.packed = packed; a.x = obs; a.m = n; a.y = nullptr;
  a.seed = seed; a.step_counter = (const long long*)step_counter; a.step_offset = step_offset; a.std12 = std12;
  a.st_actions = st_actions; a.st_mu = st_mu; a.st_sigma = st_sigma; a.st_logp = st_logp; a.actions_out = actions_out;
}

}  // namespace

extern "C" {

#ifdef LT_MLP_STAMPS
int lt_debug_mlp_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mlp_stamps), sizeof(unsigned long long) * 1024 * 8 * NW);
}
#endif

int lt_mlp_packed_floats(const lt_mlp_desc* desc, size_t* floats) {
  if (!desc_ok(desc) || !floats) { lt_set_error("lt_mlp_packed_floats: unsupported network shape"); return LT_EINVAL; }
  *floats = (size_t)geometry(desc).total_chunks * 256;
  return LT_OK;
}

int lt_mlp_pack(const lt_mlp_desc* desc, const float* const* weights, const float* const* biases, float* packed, void* stream) {
  if (!desc_ok(desc) || !weights || !biases || !packed) { lt_set_error("lt_mlp_pack: invalid argument"); return LT_EINVAL; }
  const Geometry g = geometry(desc);
  for (int l = 0; l < desc->num_layers; ++l) {
    if (!weights[l] || !biases[l]) { lt_set_error("lt_mlp_pack: null layer pointer"); return LT_EINVAL; }
    PackArgs p;
    p.w = weights[l]; p.b = biases[l]; p.K = desc->dims[l]; p.N = desc->dims[l + 1]; p.packed = packed;
    for (int w = 0; w < NW; ++w) p.chunk_off[w] = g.layer_off[l][w];
    const long long total = (long long)active_waves(p.N) * layer_chunks(p.K, p.N) * 64;
    hipLaunchKernelGGL(lt_mlp_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  }
  return LT_OK;
}

int lt_mlp_forward(const lt_mlp_desc* desc, const float* packed, const float* x, int64_t m, float* y, void* stream) {
  if (!desc_ok(desc) || !packed || !x || !y || m <= 0) { lt_set_error("lt_mlp_forward: invalid argument"); return LT_EINVAL; }
  DualArgs d = {};
  fill_args(desc, d.net[0]);
  d.net[0].mode = MODE_FORWARD;
  d.net[0].packed = packed; d.net[0].x = x; d.net[0].m = m; d.net[0].y = y;
  return launch(d, 1, (hipStream_t)stream);
}

int lt_rollout_policy(const lt_mlp_desc* actor, const float* packed, const float* obs, int64_t n, uint64_t seed, const int64_t* step_counter,
                      int64_t step_offset, const float* std12, float* st_actions, float* st_mu, float* st_sigma, float* st_logp,
                      float* actions_out, void* stream) {
  if (!policy_args_ok(actor, packed, obs, n, step_counter, std12, st_actions, st_mu, st_sigma, st_logp, actions_out)) {
    lt_set_error("lt_rollout_policy: invalid argument (the policy head must have 12 outputs)");
    return LT_EINVAL;
  }
  DualArgs d = {};
  fill_policy(actor, packed, obs, n, seed, step_counter, step_offset, std12, st_actions, st_mu, st_sigma, st_logp, actions_out, d.net[0]);
  return launch(d, 1, (hipStream_t)stream);
}

int lt_rollout_policy_value(const lt_mlp_desc* actor, const float* actor_packed, const float* obs, const lt_mlp_desc* critic,
                            const float* critic_packed, const float* critic_obs, float* values, int64_t n, uint64_t seed,
                            const int64_t* step_counter, int64_t step_offset, const float* std12, float* st_actions, float* st_mu,
                            float* st_sigma, float* st_logp, float* actions_out, void* stream) {
  if (!policy_args_ok(actor, actor_packed, obs, n, step_counter, std12, st_actions, st_mu, st_sigma, st_logp, actions_out) || !desc_ok(critic) ||
      !critic_packed || !critic_obs || !values) {
    lt_set_error("lt_rollout_policy_value: invalid argument");
    return LT_EINVAL;
  }
  DualArgs d = {};
  fill_policy(actor, actor_packed, obs, n, seed, step_counter, step_offset, std12, st_actions, st_mu, st_sigma, st_logp, actions_out, d.net[0]);
  fill_args(critic, d.net[1]);
  d.net[1].mode = MODE_FORWARD;
  d.net[1].packed = critic_packed; d.net[1].x = critic_obs; d.net[1].m = n; d.net[1].y = values;
  return launch(d, 2, (hipStream_t)stream);
}

}  // extern "C"
