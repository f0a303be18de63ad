// lt_mlp.hip - fused MLP forward (actor / critic inference of the rollout) on the f16 MFMA with error-compensated
// operand splitting: f32-equivalent results at a multiple of the f32-MFMA rate.
//
// Reference: the policy evaluation inside the rollout loop, loco_rl/loco_rl/modules/actor_critic.py:113-131
// (`act` -> update_distribution -> self.actor(obs); `evaluate` -> self.critic(obs)) - four Linear layers with ELU between,
// which PyTorch runs as 4 GEMM launches + 3 activation launches per network and step.  Here a whole network - or the actor
// and the critic side by side - is ONE launch.
//
// Arithmetic.  The reference rolls out in fp32, and gfx950 has no TF32-like mode: the f32-input MFMA runs at the f32 vector
// rate (157 TF), 1/16 of the f16/bf16 MFMA.  Every f32 operand x is therefore split into two f16 numbers,
//     x = hi + lo * 2^-6,   hi = f16(x),  lo = f16((x - hi) * 2^6)       (x - hi is exact in f32)
// and a product sum  sum_k w_k x_k  is evaluated as three f16 MFMAs into ONE f32 accumulator that holds 2^6 x the result:
//     acc += w_hi (2^6 x_hi)  +  w_hi x_lo  +  w_lo x_hi                  result = acc * 2^-6
// (2^6 x_hi is exact: |x| is clamped to 1000; the dropped w_lo x_lo term is 2^-22 relative).  The scale is 2^6 rather than
// the 2^11 that would normalise lo: with it the main operand stays an f16 number up to |x| = 1000, and lo = 2^6 (x - hi)
// keeps its full 11 bits down to |x| = 2^-8 - below that it goes subnormal with an ABSOLUTE error of 2^-30, nothing against
// the f32 rounding of the sum.  One accumulator set instead of (main, corr) halves the accumulator registers - what makes
// four row tiles per workgroup fit - and the layer epilogues read half as many AGPRs.  f16 x f16 products are exact in
// the f32 accumulator, so the only errors are the 2^-22 operand truncation and the f32 accumulation itself - measured
// against an fp64 reference the result is as close as PyTorch's own fp32 GEMM chain
// (tests/test_hip_parity.py::test_fused_mlp_matches_torch keeps the round-1 tolerance).  3 MFMAs at 16x the f32-MFMA rate:
// 5.3x less matrix-pipe time than the exact-f32 form.
//
// Structure:
//   * a workgroup (NW = 8 waves, two per SIMD) owns 16 * RT batch rows; their activations live in LDS ([rows][S] slots of 4
//     bytes per element, S = widest layer + 4) and never visit HBM between layers.  The image holds them ALREADY SPLIT: a
//     group of 8 consecutive elements is 16 B of hi halves + 16 B of lo halves, made once per element - by the prologue for
//     the input rows, by convert_pass (all threads, in place over the raw f32 sums) behind a hidden layer.  (Round 2 split
//     them in the hot loop, every wave for itself in front of its MFMAs, to keep arithmetic out of the once-per-layer code: the
//     narrow layers then ran at the VALU rate of NW redundant conversions, and the epilogues stayed cold all the same.)
//   * WHAT BOUNDS THE SMALL-BATCH LAUNCH.  A 4096-env rollout step gives every CU 32 rows of one network: 1.37 MB of weights
//     through the CU's L2 port (60-90 GB/s per CU: ~16 us) and, around that, code that runs ONCE per launch - prologue, layer
//     epilogues, the narrow last layers, the policy head - at instruction-fetch speed on cold caches (~14 cycles per
//     instruction; the step kernel between two launches replaces the instruction cache's contents).  Hence: eight waves (the
//     once-through phases are latency-bound, and a second wave per SIMD fills them: 33.0 -> 30.2 us), ONE rolled copy of the
//     conversion loop shared by all layers (30.2 -> 29.8 us) instead of one unrolled epilogue per layer shape;
//   * out^T = W . act^T on v_mfma_f32_16x16x32_f16: the weight tile is the A operand (16 output features x 32 k), the
//     activations are the B operand (32 k x 16 rows), so a lane ends up with 4 CONSECUTIVE output features of ONE row ->
//     the raw sums go back into the image with one ds_write_b128 per tile;
//   * each wave owns T = N/(16 NW) output tiles x RT row tiles (independent accumulators); RT = 2 from 8192 rows per launch
//     (every CU still gets a workgroup), 4 from 16384 (several rounds of workgroups, each streaming the weights again);
//   * the weights are pre-packed once per policy update (lt_mlp_pack) - already split into (hi, lo) f16 - into ONE LINEAR
//     STREAM PER WAVE of 1-KiB chunks (64 lanes x 16 B) in exactly the order the wave consumes them, across layers: per
//     32-wide k-group one item of T x (hi chunk, lo chunk).  The kernel keeps a register ring of 16 chunks per wave (128 KiB
//     in flight per CU; 8 chunks at four row tiles) and refills a slot right after its MFMAs, so the first weights of layer l+1 are
//     already on their way while layer l finishes.  Every wave-instruction of the stream is one fully coalesced global_load_dwordx4; all
//     workgroups stream the same ~1.4 MB per network from L2 (measured with stamps: 89 GB/s per CU in the first layer);
//   * biases live in LDS (staged once) and are added by the conversion pass;
//   * the policy head's noise is drawn in the prologue by the last wave under the latency of the input rows, and a narrow
//     last layer goes through the two-tile loop the layer before it has just run instead of a one-tile instantiation of its
//     own, on waves that had no share in the layer before it; a narrow hidden layer is dealt twice, by halves of k (Plan).
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef LT_MLP_WAVES
#define LT_MLP_WAVES 8
#endif
constexpr int NW = LT_MLP_WAVES;       // waves per workgroup (two per SIMD)
#ifndef LT_MLP_MIN_WAVES_PER_SIMD
#define LT_MLP_MIN_WAVES_PER_SIMD (LT_MLP_WAVES / 4)  // register budget: 512 / this per wave (probe builds: four waves at 2 = two workgroups per CU)
#endif
#ifndef LT_MLP_RING_GRAIN
#define LT_MLP_RING_GRAIN (LT_MLP_WAVES > 4 ? 16 : 32)
#endif
constexpr int RING = LT_MLP_RING_GRAIN;  // chunk granularity of the packed streams (layers are padded to multiples of it) = the largest ring
constexpr unsigned RS_POLICY = 0x400;  // same Philox stream id as lt_rollout_act
constexpr int MODE_FORWARD = 0, MODE_POLICY = 1, MODE_BACKWARD = 2;
constexpr int KIND_GATE = 100;  // the backward chain's "activation": multiply by ELU'(a) of the forward activation a (lt_mlp_backward_pair)
// scale of the low parts - and of the main product's activation operand, so that all three MFMAs of the split sum into ONE
// accumulator (module header, "Arithmetic")
constexpr float LO_SCALE = 64.f, LO_INV = 1.f / LO_SCALE;
// |x| beyond this saturates (64 x must stay an f16 number, and inf - inf must not appear)
constexpr float F16_CLAMP = (float)LT_MLP_INPUT_CLAMP;  // include/lt_env.h states the contract

__host__ __device__ inline int pad16(int x) { return (x + 15) & ~15; }
__host__ __device__ inline int pad32(int x) { return (x + 31) & ~31; }
// Tiles per wave of a layer with `ntiles` 16-feature output tiles: the smallest of {MIN_TILES, .., 8} that covers ntiles with
// NW waves.  Not below MIN_TILES = 2: a narrow last layer (12 actions = one tile) then runs through the loop code the layer
// before it has just pulled into the instruction cache, instead of through a one-tile instantiation fetched cold (the
// second tile is zero weights; 8 KB more in one wave's stream).
#ifndef LT_MLP_MIN_TILES
#define LT_MLP_MIN_TILES 2
#endif
constexpr int MIN_TILES = LT_MLP_MIN_TILES;
__host__ __device__ inline int tiles_per_wave(int ntiles) {
  const int per = (ntiles + NW - 1) / NW;
  const int t = per > 4 ? 8 : (per > 2 ? 4 : (per > 1 ? 2 : 1));
  return t < MIN_TILES ? MIN_TILES : t;
}
__host__ __device__ inline int active_waves(int N) {
  const int nt = pad16(N) / 16, T = tiles_per_wave(nt);
  return (nt + T - 1) / T;
}
// How a layer is dealt to the waves.  `nact` waves share the output tiles (T each).  K-SPLIT: a hidden layer narrow enough to
// leave half of the waves idle is dealt twice - waves [first, first + nact) take the first half of the k-groups, the next nact
// waves the second half of the SAME tiles; both halves put their raw sums into the image side by side (columns n and koff + n)
// and the conversion pass adds them.  A wave's share of such a layer is then one ring of chunks, all of them requested while the
// layer before it was still running: without it the third layer of the LocoTouch networks (256 -> 128: four active waves of
// eight, 32 chunks each behind a 16-slot ring) began with an L2 round trip for its second half (2.3 us for 0.3 us of MFMAs).
// The LAST layer, when it and the layer before it leave waves free, takes the waves BEHIND the previous layer's: such a wave
// has no chunks in the layer before, so its ring turns to the last layer's weights one layer early.
struct Plan {
  int T, nact, ks, G, Gl, first, waves, koff;
};
__host__ __device__ inline Plan plan_of(const int* dims, int L, int l, bool with_first = true) {
  Plan p;
  const int K = dims[l], N = dims[l + 1];
  p.T = tiles_per_wave(pad16(N) / 16);
  p.nact = active_waves(N);
  p.G = pad32(K) / 32;
  int widest = 0;
  for (int i = 0; i <= L; ++i) widest = dims[i] > widest ? dims[i] : widest;
#ifndef LT_MLP_WIDE_KSPLIT
#define LT_MLP_WIDE_KSPLIT 0
#endif
  // A hidden layer of exactly two tiles per wave (256 outputs at eight waves) dealt as FOUR tiles to half of the waves, the k-groups
  // split between the halves: every wave then reads half of the activation image for twice the tiles - the layer's LDS traffic halves
  // (at four row tiles each wave of such a layer reads the whole 64 x 512 image for 2 tiles: 170 B/clk asked of a 128 B/clk LDS).
  // MEASURED (probe build -DLT_MLP_WIDE_KSPLIT=1, r04): policy launch 175.8 against 176.0 us at 32768 envs, 29.8 against 29.3 us at
  // 4096 - the layer is not bound by its LDS reads; left off.
  if (LT_MLP_WIDE_KSPLIT && l < L - 1 && p.T == 2 && p.nact == NW && p.G >= 2 && (p.G & 1) == 0 && 2 * pad32(N) <= pad32(widest)) {
    p.T = 4;
    p.nact = NW / 2;
  }
  p.ks = (l < L - 1 && 2 * p.nact <= NW && p.G >= 2 && (p.G & 1) == 0 && 2 * pad32(N) <= pad32(widest)) ? 2 : 1;
  p.Gl = p.G / p.ks;
  p.waves = p.nact * p.ks;
  p.koff = pad32(N);
  p.first = 0;
  if (with_first && l == L - 1 && l > 0) {
    const Plan q = plan_of(dims, L, l - 1, false);
    if (q.waves + p.waves <= NW) p.first = q.waves;
  }
  return p;
}
// chunks of one layer in the stream of an ACTIVE wave: one item of 2T chunks per 32-wide k-group of its share, padded to whole ring rounds
__host__ __device__ inline int layer_chunks(const Plan& p) {
  const int R = RING / (2 * p.T);
  return (p.Gl + R - 1) / R * RING;
}

struct MlpArgs {
  int L;
  int dims[LT_MLP_MAX_LAYERS + 1];
  int activation;
  int x_bf16;                 // the input rows are bf16 (uint16 [m][dims[0]])
  int mode;
  int stride;                 // LDS row stride in floats (== 4 mod 64: the 16 rows of a ds_read_b128 phase fall on distinct banks)
  int bias_total;             // floats of the bias block (sum of pad16(N_l))
  int noise_off;              // MODE_POLICY: float offset inside the LDS image of the [rows][16] block of N(0,1) draws (launch())
  unsigned in_magic;          // dims[0] % 4 == 0: floor(2^32 / (dims[0] / 4)) + 1, the reciprocal the input staging divides by; else 0
  signed char l_first[LT_MLP_MAX_LAYERS], l_ks[LT_MLP_MAX_LAYERS];  // per layer: first active wave, k-split factor (plan_of, host side)
  signed char l_T[LT_MLP_MAX_LAYERS], l_nact[LT_MLP_MAX_LAYERS];   // ... tiles per wave, waves per k-half
  unsigned in_magic2;         // dims[0] % 4 == 2 (f32 rows): the same for pairs, floor(2^32 / (dims[0] / 2)) + 1; else 0
  long long bias_chunk;       // chunk offset of the bias block inside `packed`
  long long wave_base[NW];    // chunk offset of each wave's stream inside `packed`
  const float* packed;
  const float* x;
  long long m;
  float* y;
  float* act_out[LT_MLP_MAX_LAYERS];  // optional: the activations behind hidden layer l, [m][dims[l + 1]] (training forward)
  // MODE_POLICY
  unsigned long long seed;
  const long long* step_counter;
  long long step_offset;
  const float* std12;
  float* st_actions; float* st_mu; float* st_sigma; float* st_logp; float* actions_out;
  // MODE_BACKWARD (the chain of input gradients, module footer): per chain layer the forward activations that gate its output,
  // the per-workgroup maxima of |output| (for lt_wgrad's scale) and a counter of saturated workgroups
  int acts_split;  // act_out (forward) / gate_in (backward chain) are in the split format: per element one dword, f16 hi | f16 lo << 16 (lt_env.h)
  int dz_split;             // MODE_BACKWARD: act_out (dz) is written in the split format, still SCALED by the launch's scale (scale_out)
  const float* in_amax;     // MODE_BACKWARD: global max |input| (device scalar) - one scale for the whole launch; nullptr: one per workgroup
  float* scale_out;         // MODE_BACKWARD: the scale (workgroup 0's), for whoever reads a split dz
  const float* gate_in[LT_MLP_MAX_LAYERS];
  float* amax_out[LT_MLP_MAX_LAYERS];
  float* sat_count;
};
struct DualArgs {
  MlpArgs net[2];
  int split;     // blocks [0, split) run net[0], the rest net[1] ...
  int xcd_split; // ... unless set: blocks with (blockIdx % 8) < 4 run net[0], the others net[1] (see launch())
  int blocks_per_net;
};

template <int KIND>
__device__ __forceinline__ float activate(float x) {
  if (KIND == LT_ACT_ELU) return x > 0.f ? x : __expf(x) - 1.f;  // |abs err| ~1e-7 (v_exp_f32), far inside the parity tolerance
  if (KIND == LT_ACT_RELU) return x > 0.f ? x : 0.f;
  if (KIND == LT_ACT_TANH) return tanhf(x);
  return x;
}
// The LDS image of the activations: row r at s_act + r * S floats; inside a row, GROUP j (elements 8j .. 8j + 7) occupies
// floats [8j, 8j + 8): first the eight hi halves (16 B), then the eight lo halves (16 B),  x = hi + lo / 64  (module header).
// A lane's B fragment of a k-group - 8 consecutive k of one row - is therefore two ds_read_b128 at consecutive addresses, the
// same bytes an f32 row would take: the split is made ONCE per element, by whoever stages it (input rows: the prologue;
// hidden activations: convert_pass), not by each of the NW waves in front of its MFMAs.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (written on pairs: one v_cvt_pk_f16_f32 rounds two values, and the residual is one v_pk_add_f32 + one v_pk_mul_f32 per pair -
//  value by value the compiler converts every hi half twice, once for the residual and once for the packed store)
__device__ __forceinline__ void split4(const f32x4& v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x2 x;
    x[0] = __builtin_amdgcn_fmed3f(v[2 * p], -F16_CLAMP, F16_CLAMP);
    x[1] = __builtin_amdgcn_fmed3f(v[2 * p + 1], -F16_CLAMP, F16_CLAMP);
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 d = (x - __builtin_convertvector(h, f32x2)) * LO_SCALE;
    const f16x2 l = __builtin_convertvector(d, f16x2);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}
// a pair of consecutive elements (inputs whose width is 2 mod 4, e.g. the locomotion task's 270): hi halves at `at`, lo at + 16
__device__ __forceinline__ void store_split2(float* s_act, unsigned at, f32x2 v) {
  f32x2 x;
  x[0] = __builtin_amdgcn_fmed3f(v[0], -F16_CLAMP, F16_CLAMP);
  x[1] = __builtin_amdgcn_fmed3f(v[1], -F16_CLAMP, F16_CLAMP);
  const f16x2 h = __builtin_convertvector(x, f16x2);
  const f32x2 d = (x - __builtin_convertvector(h, f32x2)) * LO_SCALE;
  *(f16x2*)((char*)s_act + at) = h;
  *(f16x2*)((char*)s_act + at + 16) = __builtin_convertvector(d, f16x2);
}
__device__ __forceinline__ unsigned pair_at(unsigned rr, unsigned c2, int S) { return (rr * (unsigned)S + (c2 >> 2) * 8u) * 4u + (c2 & 3u) * 4u; }
// byte offset inside the image of the hi halves of elements 4 c4 .. 4 c4 + 3 of row rr (their lo halves: + 16)
__device__ __forceinline__ unsigned half_group_at(unsigned rr, unsigned c4, int S) { return (rr * (unsigned)S + (c4 >> 1) * 8u) * 4u + (c4 & 1u) * 8u; }
__device__ __forceinline__ void store_split4(float* s_act, unsigned at, const f32x4& v) {
  f16x4 hi, lo;
  split4(v, hi, lo);
  *(f16x4*)((char*)s_act + at) = hi;
  *(f16x4*)((char*)s_act + at + 16) = lo;
}
// one element (inputs whose width is not a multiple of 4)
__device__ __forceinline__ void store_split1(float* s_act, int rr, int col, int S, float v) {
  const float x = fminf(fmaxf(v, -F16_CLAMP), F16_CLAMP);
  const _Float16 h = (_Float16)x;
  _Float16* const g = (_Float16*)(s_act + rr * S + (col >> 3) * 8);
  g[col & 7] = h;
  g[8 + (col & 7)] = (_Float16)((x - (float)h) * LO_SCALE);
}


// The SPLIT FORMAT of an activation matrix in HBM (training forward -> backward chain, weight gradients): one dword per element,
// low half = f16 hi, high half = f16 lo with x = hi + lo / 64 (|x| <= F16_CLAMP) - the pair this kernel forms anyway for its LDS
// image.  Consumers that multiply on the f16 matrix cores (lt_wgrad) take the halves as they are: no conversion on their side.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 interleave4(const f16x4& hi, const f16x4& lo) {
  const u32x2 h = __builtin_bit_cast(u32x2, hi), l = __builtin_bit_cast(u32x2, lo);
  return u32x4{__builtin_amdgcn_perm(l[0], h[0], 0x05040100u), __builtin_amdgcn_perm(l[0], h[0], 0x07060302u),
               __builtin_amdgcn_perm(l[1], h[1], 0x05040100u), __builtin_amdgcn_perm(l[1], h[1], 0x07060302u)};
}
__device__ __forceinline__ f32x4 unsplit4(const u32x4& w) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] & 0xFFFFu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] >> 16)) * LO_INV;
  return r;
}

// four consecutive input elements starting at element `off` of the input rows: f32, or bf16 widened exactly (read-once: nontemporal)
__device__ __forceinline__ f32x4 load_in4(const MlpArgs& a, long long off) {
  if (a.x_bf16) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = __builtin_nontemporal_load((const u32x2*)((const unsigned short*)a.x + off));
    return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u)};
  }
  return __builtin_nontemporal_load((const f32x4*)(a.x + off));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the weight
// chunks every wave keeps in flight - twice per layer - and undo the streaming across layer boundaries.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef LT_MLP_STAMPS
#ifndef LT_MLP_STAMP_L
#define LT_MLP_STAMP_L 0  // the layer whose loop end (6) and barrier exit (7) are stamped
#endif
__device__ unsigned long long g_mlp_stamps[1024 * 8 * NW];
#define MLP_STAMP(i) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_mlp_stamps[blockIdx.x * 8 * NW + (threadIdx.x >> 6) * 8 + (i)] = t_; } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif

// Epilogue of a hidden layer, first half: the accumulators go to the image RAW (f32, still scaled by 2^6), each lane's 4
// consecutive output features of a row with one ds_write_b128 at its f32 position - T x RT stores and no arithmetic in this
// once-per-layer, instruction-cache-cold code.  Only tiles that start below the next layer's k padding (32) are written.
template <int T, int RT>
__device__ __forceinline__ void raw_store(const f32x4 (&am)[RT][T], float* s_act, int r, int q, int S, int tile0, int nwrite) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int n0 = 16 * (tile0 + t) + 4 * q;  // this lane's 4 consecutive output features = next layer's k
    if (16 * (tile0 + t) >= nwrite) continue;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *(f32x4*)(s_act + (r + 16 * rt) * S + n0) = am[rt][t];
    // one tile at a time: left alone, the scheduler reads ALL accumulators out of the AGPRs up front and spills
    __builtin_amdgcn_sched_barrier(0);
  }
}
// Second half, by ALL threads of the workgroup, as ONE rolled loop with ONE copy in the kernel (the layer loop calls it from a
// single site): everything in this kernel that is not a hot loop runs at instruction-fetch speed (~14 cycles per instruction
// on cold caches, every launch), so the body is fetched once, behind the first layer, and runs hot behind the others.  (Unrolled
// four groups deep it was slower: 2 cold trips of 450 instructions.  Started by a warm-up trip in the prologue, or fed the
// input rows as well, it was slower too: the input rows are there before the trip ends.)  A thread takes whole groups - 8
// consecutive features of a row, 32 bytes - reads the raw sums, applies scale + bias + activation, and writes the (hi, lo)
// halves back over the same 32 bytes (in place: nobody else touches the group).  Features N .. pad32(N) become zeros (the next
// layer multiplies them by zero weights, and 0 x stale bits could be 0 x NaN).  The training forward's copy of the activations
// for the backward pass (act_out) leaves from here as well, from registers.
template <int KIND, int ROWS>
__device__ __forceinline__ void convert_pass(const MlpArgs& a, int l, float* s_act, const float* s_bias, int tid, long long row0) {
  const int S = a.stride;
  const int N = a.dims[l + 1];
  const int total = ROWS * (pad32(N) >> 3);
  float* const dst = a.act_out[l];
  const bool ksplit = a.l_ks[l] > 1;
  const int koff = pad32(N);
  // Four row tiles (several rounds of workgroups per CU: the loop's code stays in the instruction cache from the second round
  // on): TWO groups per trip, phase by phase - one group's chain (LDS read, exp, three conversions) is ~600 ns of latency for
  // the two waves of a SIMD; at one or two row tiles the launch runs the loop once per layer and the longer body costs more
  // in instruction fetch than it hides.
  if (ROWS >= 64 && (N & 31) == 0 && (KIND == LT_ACT_ELU || KIND == LT_ACT_NONE)) {
    constexpr int NT = 64 * NW;
    const unsigned gpr = (unsigned)N >> 3, row_magic = 0xFFFFFFFFu / gpr + 1u;  // groups per row; idx / gpr == umulhi(idx, magic) below 2^16
#pragma unroll 1
    for (int base = tid; base < total; base += 2 * NT) {
      float* g[2];
      int n0[2], row[2];
      bool ok[2];
      f32x4 x[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = base + u * NT;
        ok[u] = idx < total;
        const int id = ok[u] ? idx : base;  // (a surplus slot re-reads the first group and stores nothing)
        if (dst) {  // training forward: lanes walk along a ROW, so that a wave's act_out stores are whole KiB stretches (gate_pass)
          const unsigned r_ = __umulhi((unsigned)id, row_magic);
          row[u] = (int)r_;
          n0[u] = 8 * (int)((unsigned)id - r_ * gpr);
        } else {    // rollout: lanes walk down the rows - conflict-free LDS access
          row[u] = id & (ROWS - 1);
          n0[u] = 8 * (id / ROWS);
        }
        g[u] = s_act + row[u] * S + n0[u];
        x[u][0] = *(const f32x4*)g[u];
        x[u][1] = *(const f32x4*)(g[u] + 4);
      }
      if (ksplit) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { x[u][0] += *(const f32x4*)(g[u] + koff); x[u][1] += *(const f32x4*)(g[u] + koff + 4); }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        x[u][0] = x[u][0] * LO_INV + *(const f32x4*)(s_bias + n0[u]);
        x[u][1] = x[u][1] * LO_INV + *(const f32x4*)(s_bias + n0[u] + 4);
      }
      if (KIND == LT_ACT_ELU) {
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 e[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) e[u][h] = __builtin_elementwise_min(x[u][h], zero) * 1.44269504088896340736f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) e[u][i >> 2][i & 3] = __builtin_amdgcn_exp2f(e[u][i >> 2][i & 3]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int h = 0; h < 2; ++h) x[u][h] = __builtin_elementwise_max(x[u][h], zero) + (e[u][h] - 1.f);
      }
      f16x4 hi[2][2], lo[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { split4(x[u][0], hi[u][0], lo[u][0]); split4(x[u][1], hi[u][1], lo[u][1]); }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!ok[u]) continue;
        *(f16x4*)g[u] = hi[u][0]; *((f16x4*)g[u] + 1) = hi[u][1];
        *((f16x4*)g[u] + 2) = lo[u][0]; *((f16x4*)g[u] + 3) = lo[u][1];
        if (dst) {
          const long long e = row0 + row[u];
          if (e < a.m) {
            if (a.acts_split) {
              *(u32x4*)(dst + e * N + n0[u]) = interleave4(hi[u][0], lo[u][0]);
              *(u32x4*)(dst + e * N + n0[u] + 4) = interleave4(hi[u][1], lo[u][1]);
            } else {
              *(f32x4*)(dst + e * N + n0[u]) = x[u][0];
              *(f32x4*)(dst + e * N + n0[u] + 4) = x[u][1];
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll 1
  for (int idx = tid; idx < total; idx += 64 * NW) {
    const int rr = idx & (ROWS - 1), j = idx / ROWS;  // consecutive lanes = consecutive rows: conflict-free (S == 4 mod 64)
    const int n0 = 8 * j;
    float* const g = s_act + rr * S + n0;
    f32x4 x[2];
    if (n0 + 8 <= N) {
      x[0] = *(const f32x4*)g;
      x[1] = *(const f32x4*)(g + 4);
      if (ksplit) {  // (uniform) the second half of the k-groups left its sums koff columns to the right (Plan)
        x[0] += *(const f32x4*)(g + koff);
        x[1] += *(const f32x4*)(g + koff + 4);
      }
      x[0] = x[0] * LO_INV + *(const f32x4*)(s_bias + n0);
      x[1] = x[1] * LO_INV + *(const f32x4*)(s_bias + n0 + 4);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i >> 2][i & 3] = n0 + i < N ? (g[i] + (ksplit ? g[koff + i] : 0.f)) * LO_INV + s_bias[n0 + i] : 0.f;
    }
    {
      const int kind = KIND >= 0 ? KIND : a.activation;
      if (kind == LT_ACT_ELU) {
        // stage by stage on float4 (-> v_pk_*_f32 pairs).  elu(x) = max(x, 0) + (exp(min(x, 0)) - 1): the same bits as
        // x > 0 ? x : exp(x) - 1  without compare / select; elu(0) = 0 keeps the padding features zero.
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 e[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) e[h] = __builtin_elementwise_min(x[h], zero) * 1.44269504088896340736f;
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i >> 2][i & 3] = __builtin_amdgcn_exp2f(e[i >> 2][i & 3]);
#pragma unroll
        for (int h = 0; h < 2; ++h) x[h] = __builtin_elementwise_max(x[h], zero) + (e[h] - 1.f);
      } else if (kind == LT_ACT_RELU) {
#pragma unroll
        for (int h = 0; h < 2; ++h) x[h] = __builtin_elementwise_max(x[h], f32x4{0.f, 0.f, 0.f, 0.f});
      } else if (kind == LT_ACT_TANH) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i >> 2][i & 3] = tanhf(x[i >> 2][i & 3]);  // (tanh(0) = 0: the padding stays zero)
      }
    }
    f16x4 hi[2], lo[2];
    split4(x[0], hi[0], lo[0]);
    split4(x[1], hi[1], lo[1]);
    *(f16x4*)g = hi[0]; *((f16x4*)g + 1) = hi[1];
    *((f16x4*)g + 2) = lo[0]; *((f16x4*)g + 3) = lo[1];
    if (dst) {  // (hidden widths of such networks are multiples of 4: lt_mlp_forward_pair)
      const long long e = row0 + rr;
      if (e < a.m) {
        if (a.acts_split) {
          if (n0 < N) *(u32x4*)(dst + e * N + n0) = interleave4(hi[0], lo[0]);
          if (n0 + 4 < N) *(u32x4*)(dst + e * N + n0 + 4) = interleave4(hi[1], lo[1]);
        } else {
          if (n0 < N) *(f32x4*)(dst + e * N + n0) = x[0];
          if (n0 + 4 < N) *(f32x4*)(dst + e * N + n0 + 4) = x[1];
        }
      }
    }
  }
}

// The backward chain's layer epilogue (KIND_GATE): raw sums -> x = sum / 64 -> x * ELU'(a) with a = the forward activation of the
// same unit, read from HBM (ELU'(z) = 1 for z > 0, else exp(z) = a + 1) -> the (hi, lo) image for the next chain layer, the f32
// copy dz (unscaled: x / scale) for the weight-gradient kernel, and the workgroup's max |x|.
// Consecutive lanes take consecutive GROUPS OF A ROW (unlike convert_pass, whose lanes walk down the rows for conflict-free LDS
// access): a wave's gate loads and dz stores are then whole 2-KiB stretches of a row - lane-per-row, both moved 32 bytes per
// 128-byte line and request (measured on the 24 576-row update: loads +73 us, stores +43 us on a 75 us launch) - and the LDS side
// pays a 2-way bank conflict on its b128 accesses instead.  ALL gate loads of the layer leave before the first is used (at most 8
// groups per thread at 64 rows x 512 features; the accumulators are dead here, the registers are free): one HBM round trip per
// layer.
template <int ROWS>
__device__ __forceinline__ float gate_pass(const MlpArgs& a, int l, float* s_act, int tid, long long row0, float inv_scale, bool to_lds) {
  constexpr int NT = 64 * NW, U = ROWS >= 64 ? 8 : (ROWS >= 32 ? 4 : 2);  // ROWS * 64 groups (N = 512) / NT threads
  const int S = a.stride;
  const int N = a.dims[l + 1];  // a multiple of 8 (lt_mlp_backward_pair)
  const unsigned gpr = (unsigned)N >> 3;  // groups per row (<= 64)
  const unsigned magic = 0xFFFFFFFFu / gpr + 1u;  // idx / gpr == umulhi(idx, magic) for idx < 2^16
  const int total = ROWS * (int)gpr;
  const bool ksplit = a.l_ks[l] > 1;
  const int koff = pad32(N);
  const float* const gate = a.gate_in[l];
  float* const dst = a.act_out[l];
  const long long left = a.m - row0;
  const int rmax = left < ROWS ? (int)left - 1 : ROWS - 1;
  float mx = 0.f;
  f32x4 gt[U][2];
  int at[U];      // float offset of the group inside the image: row * S + 8 j
  int row[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = tid + u * NT;
    row[u] = -1;
    if (idx < total) {  // (wave-uniform: total and NT are multiples of 64)
      const unsigned r = gpr == 1u ? (unsigned)idx : __umulhi((unsigned)idx, magic), j = (unsigned)idx - r * gpr;  // (gpr == 1: the reciprocal wraps to 0)
      row[u] = (int)r;
      at[u] = (int)r * S + 8 * (int)j;
#ifdef LT_GATE_NO_LOAD  // probe builds (tools/mlp_backward_probe.py)
      gt[u][0] = gt[u][1] = f32x4{1.f, 1.f, 1.f, 1.f};
#else
      const float* const gp = gate + (row0 + min((int)r, rmax)) * N + 8 * j;
      gt[u][0] = __builtin_nontemporal_load((const f32x4*)gp);
      gt[u][1] = __builtin_nontemporal_load((const f32x4*)(gp + 4));
#endif
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (row[u] < 0) continue;
    float* const g = s_act + at[u];
    f32x4 x[2];
    x[0] = *(const f32x4*)g;
    x[1] = *(const f32x4*)(g + 4);
    if (ksplit) { x[0] += *(const f32x4*)(g + koff); x[1] += *(const f32x4*)(g + koff + 4); }
    if (a.acts_split) {  // (uniform) the forward left (hi, lo) pairs: a = hi + lo / 64
      gt[u][0] = unsplit4(__builtin_bit_cast(u32x4, gt[u][0]));
      gt[u][1] = unsplit4(__builtin_bit_cast(u32x4, gt[u][1]));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a_ = gt[u][h][i];
        x[h][i] = x[h][i] * LO_INV * (a_ > 0.f ? 1.f : a_ + 1.f);
      }
    const bool live = row[u] <= rmax;
    if (live) {
#pragma unroll
      for (int h = 0; h < 2; ++h) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x[h][0]), fabsf(x[h][1]))), fmaxf(fabsf(x[h][2]), fabsf(x[h][3])));
    }
    f16x4 hi[2], lo[2];
    if (to_lds || a.dz_split) {
      split4(x[0], hi[0], lo[0]);
      split4(x[1], hi[1], lo[1]);
    }
    if (to_lds) {
      *(f16x4*)g = hi[0]; *((f16x4*)g + 1) = hi[1];
      *((f16x4*)g + 2) = lo[0]; *((f16x4*)g + 3) = lo[1];
    }
#ifdef LT_GATE_NO_STORE
    if (live && x[0][0] == 123.456f) {
#else
    if (live) {
#endif
      float* const o = dst + (row0 + row[u]) * N + (at[u] - row[u] * S);
      if (a.dz_split) {  // (uniform) the halves the next chain layer multiplies, as they are - scaled: lt_wgrad divides its sums
        *(u32x4*)o = interleave4(hi[0], lo[0]);
        *(u32x4*)(o + 4) = interleave4(hi[1], lo[1]);
      } else {
        *(f32x4*)o = x[0] * inv_scale;
        *(f32x4*)(o + 4) = x[1] * inv_scale;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  return mx;
}

// Policy head, first half: the standard-normal draws of this workgroup's rows and the log-density of the sample
// (Normal.log_prob summed over the 12 actions needs z and sigma only, not mu).  The last wave runs it in the prologue, between
// requesting the weight ring and the arrival of the input rows - ~250 instructions that would otherwise sit, fetched cold,
// behind the last layer where nothing overlaps them.  Lane (r, q) owns actions 4q..4q+3 of row r (q == 3: padding).
template <int RT>
__device__ __forceinline__ void policy_noise(const MlpArgs& a, long long row0, int lane, float* s_noise) {
  const int r = lane & 15, q = lane >> 4;
  if (q >= 3) return;
  const unsigned long long step = (unsigned long long)(a.step_counter[0] + a.step_offset);
  const float4 sg = *(const float4*)(a.std12 + 4 * q);
  const float sgv[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const long long e = row0 + 16 * rt + r;
    const U4 u = rng4(a.seed, (unsigned)e, step, RS_POLICY + q);
    const float ra = sqrtf(-2.f * __logf(1.f - u.a)), rb = sqrtf(-2.f * __logf(1.f - u.c));  // 1-u in (0,1]: never log(0)
    float sa, ca, sb, cb;
    __sincosf(6.28318530717958647692f * u.b, &sa, &ca);
    __sincosf(6.28318530717958647692f * u.d, &sb, &cb);
    const float z[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
    float lp = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) lp += -(z[i] * z[i]) * 0.5f - __logf(sgv[i]) - 0.91893853320467274178f;  // Normal.log_prob
    *(float4*)(s_noise + (16 * rt + r) * 16 + 4 * q) = make_float4(z[0], z[1], z[2], z[3]);
    s_noise[(16 * rt + r) * 16 + 12 + q] = lp;
  }
}

// One layer for this wave.  `ring` slot s holds chunk c0 + s of the wave's stream; the layer consumes its items in order
// (item i = k-group i: T x (hi chunk, lo chunk)) and leaves the ring positioned on the next layer's first chunk.
// RT row tiles (16 rows each) share every weight chunk: RT x the MFMA work per byte streamed from L2.
// RG = ring slots (chunks in flight per wave): 32, or 16 where the accumulators need the registers (RG divides RING, so a
// ring round never straddles a layer).
template <int T, int RT, int RG, int KIND>
__device__ __forceinline__ void mlp_layer(const MlpArgs& a, int l, bool last, float* s_act, const float* s_bias, const float* s_noise, int wave, int lane,
                                          long long row_block, float4 (&ring)[RG], const float4* __restrict__ stream, long long& c0) {
  constexpr int C = 2 * T;      // chunks per item
  constexpr int R = RG / C;     // items per ring round
  constexpr int U = (R & 1) ? 2 : 1;  // rounds per loop trip: the activation double buffer alternates by (compile-time) item parity
  static_assert(R >= 1 && RING % RG == 0, "ring geometry");
  const int r = lane & 15, q = lane >> 4;
  const int S = a.stride;
  const int N = a.dims[l + 1];
  // the layer's deal (Plan): `nact` waves share the tiles; k-split layers are dealt twice, the second half of the k-groups to the
  // next nact waves, whose raw sums go `koff` columns to the right
  const int first = a.l_first[l], ks = a.l_ks[l], nact = a.l_nact[l];
  const int rel = wave - first, kh = rel >= nact ? 1 : 0;
  const int G = pad32(a.dims[l]) / 32 / ks;  // this wave's k-groups: [kh G, (kh + 1) G)
  const int tile0 = (rel - kh * nact) * T;
  const bool active = rel >= 0 && rel < nact * ks;
  const float* const xrow = s_act + r * S + 8 * q + 32 * G * kh;  // row tile rt: + 16 * rt * S; k-group g of this wave's share: + 32 g
  f32x4 am[RT][T];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < T; ++t) am[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {
    // software pipeline over items: the (hi, lo) fragments of item i+1 in flight from LDS while item i is in the MFMAs
#ifndef LT_MLP_SINGLE_FRAG
#define LT_MLP_SINGLE_FRAG 0
#endif
    // SF (four row tiles): ONE fragment set, fetched at the item's start instead of one item ahead - the 32 registers of the second
    // set pay for a second ring round (16 slots: two items of the widest layer in flight instead of one; with 8 slots every item of a
    // wide layer waits out an L2 round trip: 11 items x ~1.3 us = the 13.8 us of the first layer).  The LDS latency this exposes (~0.2
    // us per item) hides behind the SIMD's other wave and the MFMAs still executing.  MEASURED (probe build -DLT_MLP_SINGLE_FRAG=1
    // -DLT_MLP_RING4=16: 241 VGPRs, no scratch): policy launch 174 against 172 us at 32768 envs, training forward 154.6 against 156.0 us -
    // the deeper ring buys nothing at four row tiles either; left off.
    constexpr bool SF = RT >= 4 && LT_MLP_SINGLE_FRAG != 0;
    f16x8 xh[SF ? 1 : 2][RT], xl[SF ? 1 : 2][RT];
    if (!SF) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        xh[0][rt] = *(const f16x8*)(xrow + 16 * rt * S);
        xl[0][rt] = *(const f16x8*)(xrow + 16 * rt * S + 4);
      }
    }
    // items incl. the zero-weight pad items of the last RING round: they are walked (refill only) to leave the ring on the
    // next layer's first chunk - except behind the last layer, where nothing follows
    const int Gp = last ? G : (G + RING / C - 1) / (RING / C) * (RING / C);
    for (int i0 = 0; i0 < Gp; i0 += R * U) {
#pragma unroll
      for (int j = 0; j < R * U; ++j) {
        const int i = i0 + j;  // item = k-group
        const int sl = (j % R) * C;  // first ring slot of the item
        if (i < G) {
          // fetch item i+1's fragments (the last item re-reads itself)
          const int gx = SF ? i : (i + 1 < G ? i + 1 : G - 1);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            xh[SF ? 0 : ((j & 1) ^ 1)][rt] = *(const f16x8*)(xrow + 16 * rt * S + 32 * gx);
            xl[SF ? 0 : ((j & 1) ^ 1)][rt] = *(const f16x8*)(xrow + 16 * rt * S + 32 * gx + 4);
          }
          // the three products of the split, all into the one accumulator (product-major order - two MFMAs on the same
          // accumulator T x RT apart instead of T - measured the same within noise)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const f16x8 bh = xh[SF ? 0 : (j & 1)][rt], bl = xl[SF ? 0 : (j & 1)][rt];
            const f16x8 bs1 = bh * (_Float16)LO_SCALE;  // exact: |x| <= F16_CLAMP
#pragma unroll
            for (int t = 0; t < T; ++t) am[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ring[sl + 2 * t]), bs1, am[rt][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) am[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ring[sl + 2 * t]), bl, am[rt][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) am[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ring[sl + 2 * t + 1]), bh, am[rt][t], 0, 0, 0);
          }
        }
        // refill the slots just consumed (pad items included: the ring invariant must hold for the next layer).  UNCONDITIONAL:
        // behind a branch the compiler cannot count these loads, and its s_waitcnt in front of the NEXT item's MFMAs then
        // waits for all of them - vmcnt(3), (1), (0) instead of (12+) - which made the ring one item deep (every item paid an
        // L2 round trip: 350 ns per item in the narrow layers, 730 ns in the first).  The last layer's surplus refills land in
        // the ring of zero chunks behind every stream (geometry()).
#pragma unroll
        for (int s = 0; s < C; ++s) ring[sl + s] = stream[(c0 + RG + (j / R) * RG + sl + s) * 64];
        // keep the refill HERE: left alone, the scheduler sinks it behind the next item's MFMAs to save registers, which
        // collapses the ring to one item in flight
        __builtin_amdgcn_sched_barrier(0);
      }
      c0 += RG * U;
    }
  }
#ifdef LT_MLP_STAMPS
  if (l == LT_MLP_STAMP_L) MLP_STAMP(6);
#endif
  lds_barrier();  // every wave is done reading this layer's input
#ifdef LT_MLP_STAMPS
  if (l == LT_MLP_STAMP_L) MLP_STAMP(7);
#endif
  if (!last) {
    if (active) raw_store<T, RT>(am, s_act + kh * pad32(N), r, q, S, tile0, pad32(N));
    return;
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const long long e = row_block * (16 * RT) + 16 * rt + r;
    float out[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const f32x4 bias = active ? *(const f32x4*)(s_bias + 16 * (tile0 + t) + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) out[t][i] = active ? am[rt][t][i] * LO_INV + bias[i] : 0.f;
    }
    if (a.mode == MODE_FORWARD) {
      if (active && e < a.m) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = 16 * (tile0 + t) + 4 * q + i;
            if (n < N) a.y[e * N + n] = out[t][i];
          }
        }
      }
    } else {
      // policy head: N == 12 -> one tile, held by the layer's first active wave; lane (r, q) owns actions 4q..4q+3 of env e (q == 3: padding)
      if (rel == 0) {
        float lp = 0.f;
        if (q < 3 && e < a.m) {
          // the N(0,1) draws and their log-density were made while the input rows were in flight (policy_noise)
          const float4 zz = *(const float4*)(s_noise + (16 * rt + r) * 16 + 4 * q);
          lp = s_noise[(16 * rt + r) * 16 + 12 + q];
          const float4 sg = *(const float4*)(a.std12 + 4 * q);
          const long long o = e * 12 + 4 * q;
          const float4 xo = make_float4(out[0][0] + sg.x * zz.x, out[0][1] + sg.y * zz.y, out[0][2] + sg.z * zz.z, out[0][3] + sg.w * zz.w);
          *(float4*)(a.st_actions + o) = xo;
          *(float4*)(a.actions_out + o) = xo;
          *(float4*)(a.st_mu + o) = make_float4(out[0][0], out[0][1], out[0][2], out[0][3]);
          *(float4*)(a.st_sigma + o) = sg;
        }
        lp += __shfl_xor(lp, 16, 64);
        lp += __shfl_xor(lp, 32, 64);
        if (q == 0 && e < a.m) a.st_logp[e] = lp;
      }
    }
  }
}

// KIND: the hidden activation as a compile-time constant (the one every LocoTouch network uses gets its own, smaller kernel:
// the epilogues run once per layer, at instruction-fetch speed), or -1 = read it from the arguments.
// IN: how the input rows are staged, as a compile-time constant - f32 rows by float4, bf16 rows by 4 elements, f32 rows by pairs
// (widths 2 mod 4) - or IN_ANY = decided at run time (odd widths, and networks of the generic-activation kernel).  With the three
// forms behind run-time branches the loaded registers met in PHI copies at the join: the copies waited for the input rows, and
// the bias and weight-ring requests behind them left ~3 us late.
constexpr int IN_ANY = 0, IN_F32X4 = 1, IN_BF16X4 = 2, IN_F32X2 = 3;
template <int RT, int KIND, int IN>
__global__ __launch_bounds__(64 * NW, LT_MLP_MIN_WAVES_PER_SIMD) void lt_mlp_kernel(const DualArgs d) {
  extern __shared__ __attribute__((aligned(16))) float s_img[];
  constexpr int ROWS = 16 * RT;
  bool second;
  long long row_block;
  if (d.xcd_split) {  // workgroups are dealt round-robin over the 8 XCDs: net 0 on four of them, net 1 on the other four
    const int x = blockIdx.x & 7;
    second = x >= 4;
    row_block = (long long)(blockIdx.x >> 3) * 4 + (x & 3);
    if (row_block >= d.blocks_per_net) return;
  } else {
    second = (int)blockIdx.x >= d.split;
    row_block = second ? (long long)blockIdx.x - d.split : (long long)blockIdx.x;
  }
  const MlpArgs& a = second ? d.net[1] : d.net[0];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: per-wave decisions become s_cbranch, not exec masks
  MLP_STAMP(0);
  const long long row0 = row_block * ROWS;
  const int S = a.stride;
  float* const s_act = s_img;
  float* const s_bias = s_img + ROWS * S;
  // Input rows first: their loads enter the memory queue AHEAD of the weight ring (vmcnt retires in order - behind 128 KiB
  // of weight chunks the 45 KB of rows would arrive ~2 us later).  Read-once rows: nontemporal, so they do not push the
  // weight stream out of the XCD's L2.
  // This prologue runs once per launch, i.e. at instruction-fetch speed (~14 cycles per instruction on cold caches): it is
  // written for instruction COUNT.  Thread t takes float4 number t, t + 256, ... of the workgroup's ROWS x K0/4 real input
  // (index -> (row, column) by a host-made reciprocal, no predicates: indices past the end re-read the last float4 and store
  // it where it already is; rows beyond m re-read row m - 1 - rows never mix in the MFMA and those results are not stored);
  // the k padding (columns K0 .. pad32(K0), zero weights) is zeroed separately - 0 x stale LDS bits could be 0 x NaN.
  const int K0 = a.dims[0], K0p = pad32(K0);
  const bool vec_in = IN == IN_ANY ? a.in_magic != 0 : (IN == IN_F32X4 || IN == IN_BF16X4);  // K0 % 4 == 0 (fill_args)
  const bool is_bf16 = IN == IN_ANY ? a.x_bf16 != 0 : IN == IN_BF16X4;
  constexpr int B = (6 * RT * 4 + NW - 1) / NW;  // float4 in flight per thread: one batch covers a 348-wide input (5.4 per thread and row tile at four waves, 2.7 at eight)
  constexpr int NT = 64 * NW;
  const int kv = K0p >> 2, k4 = K0 >> 2;
  const unsigned tv = ROWS * k4;
  const long long left = a.m - row0;
  if (left <= 0) return;  // (launch() starts no such workgroup)
  const unsigned rmax = left < ROWS ? (unsigned)left - 1u : ROWS - 1u;
  f32x4 vin[B];
  unsigned lds_at[B];
  if (vec_in) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const char* const xb = (const char*)a.x + row0 * K0 * (is_bf16 ? 2 : 4);
    if (is_bf16) {  // (the format test outside the unrolled batch: one scalar branch, two straight-line load sequences)
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const unsigned idx = min((unsigned)tid + u * NT, tv - 1u);
        const unsigned rr = __umulhi(idx, a.in_magic), cc = idx - rr * k4;
        lds_at[u] = half_group_at(rr, cc, S);
        // raw bits now, widened when they go to LDS: a conversion here would wait for the row load before the weight ring is requested
        const u32x2 w = __builtin_nontemporal_load((const u32x2*)(xb + 2u * (min(rr, rmax) * K0 + 4 * cc)));
        vin[u] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const unsigned idx = min((unsigned)tid + u * NT, tv - 1u);
        const unsigned rr = __umulhi(idx, a.in_magic), cc = idx - rr * k4;
        lds_at[u] = half_group_at(rr, cc, S);
        vin[u] = __builtin_nontemporal_load((const f32x4*)(xb + 4u * (min(rr, rmax) * K0 + 4 * cc)));
      }
    }
  }
  // widths that are 2 mod 4 (f32 rows): the same scheme on pairs - two 8-byte loads per float4 slot
  const bool vec2_in = IN == IN_ANY ? a.in_magic2 != 0 : IN == IN_F32X2;
  const unsigned p2 = K0 >> 1, tv2 = ROWS * p2;
  unsigned lds_at2[B];
  if (vec2_in) {
    const char* const xb = (const char*)a.x + row0 * K0 * 4;
#pragma unroll
    for (int u = 0; u < B; ++u) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned idx = min((unsigned)tid + (2 * u + h) * NT, tv2 - 1u);
        const unsigned rr = __umulhi(idx, a.in_magic2), c2 = idx - rr * p2;
        (h ? lds_at2[u] : lds_at[u]) = pair_at(rr, c2, S);
        const f32x2 w = __builtin_nontemporal_load((const f32x2*)(xb + 4u * (min(rr, rmax) * K0 + 2 * c2)));
        vin[u][2 * h] = w[0]; vin[u][2 * h + 1] = w[1];
      }
    }
  }
  // biases: requested before the weight ring as well (behind it they would wait for every chunk of it)
  constexpr int BB = 4;
  const float* const bsrc = a.packed + a.bias_chunk * 256;
  float bv[BB];
#pragma unroll
  for (int u = 0; u < BB; ++u) bv[u] = bsrc[min(tid + u * NT, a.bias_total - 1)];
  // weight stream: a full ring in flight.  Two waves per SIMD leave a wave 256 registers: accumulators (16 RT at the widest
  // layer), fragments (16 RT) and a 16-slot ring (64) fit without scratch at two row tiles (a scratch reload in a layer
  // epilogue would sit behind the whole ring in the queue); four row tiles spill 88 bytes.
#ifndef LT_MLP_RING2
#define LT_MLP_RING2 16  // ring slots at two row tiles (32 at eight waves: 352 B of scratch)
#endif
#ifndef LT_MLP_RING4
#define LT_MLP_RING4 8   // ring slots at four row tiles: one item of the widest layer - 16 slots spill 72 bytes and measured slower (177 against 171 us at 32768 envs: the launch is not stream-bound there)
#endif
#ifndef LT_MLP_RING1
#define LT_MLP_RING1 (LT_MLP_WAVES > 4 ? 16 : 32)  // ring slots at one row tile
#endif
  constexpr int RG = RT >= 4 ? LT_MLP_RING4 : (RT >= 2 ? LT_MLP_RING2 : LT_MLP_RING1);  // (two waves per SIMD: 256 registers each)
  const float4* __restrict__ stream = (const float4*)a.packed + a.wave_base[wave] * 64 + lane;
  float4 ring[RG];
#pragma unroll
  for (int s = 0; s < RG; ++s) ring[s] = stream[s * 64];
  long long c0 = 0;
  float* const s_noise = s_img + a.noise_off;
  // The policy head's noise: by the last wave (idle in the narrow layers), under the latency of the input rows.
  if (a.mode == MODE_POLICY && wave == NW - 1) policy_noise<RT>(a, row0, lane, s_noise);
  // Backward chain: the workgroup's input rows (gradients: 1e-3 .. 1e-9) are scaled by a power of two that brings their largest
  // magnitude to [1, 2) - the (hi, lo) f16 image resolves 2^-30 absolutely, and saturates at F16_CLAMP: three layers of growth by
  // less than 500 each time are covered; a workgroup that saturates all the same is counted (sat_count).  Rows are independent in
  // the chain, so every workgroup has its own scale and takes it out again where it writes dz.
  float scale = 1.f, inv_scale = 1.f;
  if constexpr (KIND == KIND_GATE) {
    float m_ = 0.f;
    if (a.in_amax) {  // (uniform) ONE scale for the launch, from the maximum the producer of the input left (lt_ppo_loss): a dz
                      // written in the split format must carry the same scale in every row - lt_wgrad adds rows of all workgroups
      m_ = *a.in_amax;
    } else {
      float mx = 0.f;
      if (vec_in || vec2_in) {
#pragma unroll
        for (int u = 0; u < B; ++u) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(vin[u][0]), fabsf(vin[u][1]))), fmaxf(fabsf(vin[u][2]), fabsf(vin[u][3])));
      } else {
        for (int idx = tid; idx < ROWS * K0; idx += NT) {
          const unsigned rr = idx / K0;
          mx = fmaxf(mx, fabsf(a.x[(row0 + min(rr, rmax)) * K0 + (idx - rr * K0)]));
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      if (lane == 0) s_noise[wave] = mx;
      lds_barrier();
#pragma unroll
      for (int w = 0; w < NW; ++w) m_ = fmaxf(m_, s_noise[w]);
    }
    int se = 127 - (int)((__float_as_uint(m_) >> 23) & 0xFF);  // -floor(log2 max); zero / denormal rows: 127 -> clamped
    se = m_ > 0.f ? (se > 100 ? 100 : (se < -100 ? -100 : se)) : 0;
    scale = __uint_as_float((unsigned)(127 + se) << 23);
    inv_scale = __uint_as_float((unsigned)(127 - se) << 23);
    if (a.scale_out && row_block == 0 && tid == 0) *a.scale_out = scale;
  }
  {
#pragma unroll
    for (int u = 0; u < BB; ++u) s_bias[min(tid + u * NT, a.bias_total - 1)] = bv[u];
    for (int i = tid + BB * NT; i < a.bias_total; i += NT) s_bias[i] = bsrc[i];
    if (vec_in) {
      if (is_bf16) {  // (uniform: two straight-line sequences, not a select per value)
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const unsigned w0 = __float_as_uint(vin[u][0]), w1 = __float_as_uint(vin[u][1]);
          store_split4(s_act, lds_at[u], f32x4{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xFFFF0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xFFFF0000u)});
        }
      } else {
#pragma unroll
        for (int u = 0; u < B; ++u) store_split4(s_act, lds_at[u], KIND == KIND_GATE ? vin[u] * scale : vin[u]);
      }
      for (unsigned idx = tid + B * NT; idx < tv; idx += NT) {  // inputs wider than one batch (not in the backward chain: its inputs are <= 64 wide)
        const unsigned rr = idx / k4, cc = idx - rr * k4;
        store_split4(s_act, half_group_at(rr, cc, S), load_in4(a, (row0 + min(rr, rmax)) * K0 + 4 * cc));
      }
      for (int i = tid; i < ROWS * (kv - k4); i += NT) {  // the k padding: zero halves
        const unsigned at = half_group_at(i & (ROWS - 1), k4 + i / ROWS, S);
        *(f16x4*)((char*)s_act + at) = f16x4{0, 0, 0, 0};
        *(f16x4*)((char*)s_act + at + 16) = f16x4{0, 0, 0, 0};
      }
    } else if (vec2_in) {
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const f32x4 v = KIND == KIND_GATE ? vin[u] * scale : vin[u];
        store_split2(s_act, lds_at[u], f32x2{v[0], v[1]});
        store_split2(s_act, lds_at2[u], f32x2{v[2], v[3]});
      }
      for (unsigned idx = tid + 2 * B * NT; idx < tv2; idx += NT) {  // inputs wider than one batch
        const unsigned rr = idx / p2, c2 = idx - rr * p2;
        store_split2(s_act, pair_at(rr, c2, S), __builtin_nontemporal_load((const f32x2*)(a.x + (row0 + min(rr, rmax)) * K0 + 2 * c2)));
      }
      for (int i = tid; i < ROWS * ((K0p >> 1) - (int)p2); i += NT) {  // the k padding: zero halves
        const unsigned at = pair_at(i & (ROWS - 1), p2 + i / ROWS, S);
        *(f16x2*)((char*)s_act + at) = f16x2{0, 0};
        *(f16x2*)((char*)s_act + at + 16) = f16x2{0, 0};
      }
    } else {
      for (int idx = tid; idx < ROWS * K0p; idx += NT) {
        const int rr = idx / K0p, cc = idx - rr * K0p;
        const long long e = row0 + rr;
        const float v = (cc < K0 && e < a.m) ? (a.x_bf16 ? __uint_as_float((unsigned)((const unsigned short*)a.x)[e * K0 + cc] << 16) : a.x[e * K0 + cc]) : 0.f;
        store_split1(s_act, rr, cc, S, KIND == KIND_GATE ? v * scale : v);
      }
    }
  }
  lds_barrier();  // (not __syncthreads: that would drain the weight ring as well)
  MLP_STAMP(1);
  // per layer: hot loop, raw sums -> image, conversion pass (one copy: above)
  int boff = 0;
#pragma unroll 1
  for (int l = 0; l < a.L; ++l) {
    const int T = a.l_T[l];
    const bool last = KIND != KIND_GATE && l == a.L - 1;  // (the chain's last layer is gated and written like the others)
    const float* const bias_l = s_bias + boff;
    if (NW <= 4 && T == 8) mlp_layer<(NW <= 4 ? 8 : 4), RT, RG, KIND>(a, l, last, s_act, bias_l, s_noise, wave, lane, row_block, ring, stream, c0);  // (eight waves: at most 4 tiles each)
    else if (T == 4) mlp_layer<4, RT, RG, KIND>(a, l, last, s_act, bias_l, s_noise, wave, lane, row_block, ring, stream, c0);
    else if (MIN_TILES >= 2 || T == 2) mlp_layer<(MIN_TILES > 2 ? MIN_TILES : 2), RT, RG, KIND>(a, l, last, s_act, bias_l, s_noise, wave, lane, row_block, ring, stream, c0);
    else mlp_layer<1, RT, RG, KIND>(a, l, last, s_act, bias_l, s_noise, wave, lane, row_block, ring, stream, c0);
    MLP_STAMP(2 + l);
    if (last) break;
    lds_barrier();  // the raw sums of the whole layer are in the image
    if constexpr (KIND == KIND_GATE) {
      const float mx = gate_pass<ROWS>(a, l, s_act, tid, row0, inv_scale, l + 1 < a.L);
      if (lane == 0) s_noise[wave] = mx;
      lds_barrier();
      if (tid == 0) {
        float m_ = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m_ = fmaxf(m_, s_noise[w]);
        a.amax_out[l][row_block] = m_ * inv_scale;
        if (m_ >= F16_CLAMP && a.sat_count) atomicAdd(a.sat_count, 1.f);
      }
    } else {
      convert_pass<KIND, ROWS>(a, l, s_act, bias_l, tid, row0);
      boff += pad16(a.dims[l + 1]);
      lds_barrier();
    }
  }
}

// One layer of one network: weights [N][K] (torch.nn.Linear layout) -> the per-wave chunk streams, bias [N] -> the bias block.
struct PackArgs {
  const float* w; const float* b;  // b == nullptr: zero biases
  int K, N;
  int transposed;           // element (n, k) is w[k * N + n] (the backward chain multiplies by W^T of a torch.nn.Linear weight [K][N])
  long long chunk_off[NW];  // first chunk of this layer in each wave's stream (absolute, in chunks)
  Plan plan;                // how the layer is dealt to the waves
  long long bias_float_off; // first float of this layer's (pad16(N)) bias slice
  float* packed;
};
// all layers of a network in one launch: blockIdx.y = layer
struct PackAll { PackArgs layer[LT_MLP_MAX_LAYERS]; };
// forward + backward streams of two networks (lt_mlp_pack_training): 2 x (L + L - 1) layers in one launch
struct PackMany { PackArgs layer[4 * LT_MLP_MAX_LAYERS]; };
template <class ALL>
__global__ void lt_mlp_pack_kernel(const ALL all) {
  const PackArgs& p = all.layer[blockIdx.y];
  const Plan& pl = p.plan;
  const int T = pl.T, G = pl.Gl, C = 2 * T;
  const int chunks = layer_chunks(pl), nact = pl.waves;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (wave, chunk, lane)
  if (idx < pad16(p.N)) p.packed[p.bias_float_off + idx] = (idx < p.N && p.b) ? p.b[idx] : 0.f;
  if (idx >= (long long)nact * chunks * 64) return;
  const int lane = (int)(idx & 63);
  const int c = (int)((idx >> 6) % chunks), wv = (int)((idx >> 6) / chunks);
  const int kh = wv / pl.nact, tw = wv - kh * pl.nact;  // k half, tile wave
  const int r = lane & 15, q = lane >> 4;
  const int g = c / C, slot = c - g * C;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    const int t = slot >> 1, comp = slot & 1;
    const int n = 16 * (tw * T + t) + r, k0 = 32 * (kh * G + g) + 8 * q;
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = (n < p.N && k0 + j < p.K) ? (p.transposed ? p.w[(long long)(k0 + j) * p.N + n] : p.w[(long long)n * p.K + k0 + j]) : 0.f;
      x = fminf(fmaxf(x, -F16_CLAMP), F16_CLAMP);
      const _Float16 hi = (_Float16)x;
      h[j] = comp == 0 ? hi : (_Float16)((x - (float)hi) * LO_SCALE);
    }
    out = __builtin_bit_cast(float4, h);
  }
  *(float4*)(p.packed + ((p.chunk_off[pl.first + wv] + c) * 64 + lane) * 4) = out;
}

bool desc_ok(const lt_mlp_desc* d) {
  if (!d || d->num_layers < 1 || d->num_layers > LT_MLP_MAX_LAYERS) return false;
  if (d->input_format != LT_ROWS_F32 && !(d->input_format == LT_ROWS_BF16 && (d->dims[0] & 3) == 0)) return false;
  if (d->activation < LT_ACT_NONE || d->activation > LT_ACT_TANH) return false;
  for (int l = 0; l <= d->num_layers; ++l)
    if (d->dims[l] < 1 || d->dims[l] > LT_MLP_MAX_WIDTH) return false;
  for (int l = 1; l <= d->num_layers; ++l)
    if (d->dims[l] > 512) return false;  // 4 output tiles per wave at most (NW = 8; 8 at NW = 4)
  return true;
}

// stream geometry: per-wave totals (with one ring of zero chunks behind each stream: the refill runs one ring ahead)
struct Geometry {
  long long wave_base[NW];
  long long layer_off[LT_MLP_MAX_LAYERS][NW];
  long long bias_chunk;   // the bias block (sum of pad16(N_l) floats) sits behind the streams
  int bias_total;
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
      const Plan pl = plan_of(d->dims, d->num_layers, l);
      if (w >= pl.first && w < pl.first + pl.waves) off += layer_chunks(pl);
    }
    base += off + RING;
  }
  g.bias_chunk = base;
  g.bias_total = 0;
  for (int l = 0; l < d->num_layers; ++l) g.bias_total += pad16(d->dims[l + 1]);
  g.total_chunks = base + (g.bias_total + 255) / 256;
  return g;
}

void fill_args(const lt_mlp_desc* d, MlpArgs& a) {
  a.L = d->num_layers;
  a.activation = d->activation;
  a.x_bf16 = d->input_format == LT_ROWS_BF16;
  int widest = 0;
  for (int l = 0; l <= d->num_layers; ++l) {
    a.dims[l] = d->dims[l];
    widest = d->dims[l] > widest ? d->dims[l] : widest;
  }
  const Geometry g = geometry(d);
  for (int w = 0; w < NW; ++w) a.wave_base[w] = g.wave_base[w];
  for (int l = 0; l < d->num_layers; ++l) {
    const Plan pl = plan_of(d->dims, d->num_layers, l);
    a.l_first[l] = (signed char)pl.first; a.l_ks[l] = (signed char)pl.ks; a.l_T[l] = (signed char)pl.T; a.l_nact[l] = (signed char)pl.nact;
  }
  a.stride = pad32(widest) + 4;
  a.bias_total = g.bias_total;
  // exact for every index below ROWS x K0/4 as long as ROWS (K0/4)^2 < 2^32 - LDS holds no such row; K0 = 4 has no 32-bit reciprocal
  a.in_magic = (d->dims[0] % 4 == 0 && d->dims[0] >= 8) ? (unsigned)(0x100000000ull / (unsigned)(d->dims[0] / 4)) + 1u : 0u;
  a.in_magic2 = (d->dims[0] % 4 == 2 && d->dims[0] >= 6 && d->input_format == LT_ROWS_F32) ? (unsigned)(0x100000000ull / (unsigned)(d->dims[0] / 2)) + 1u : 0u;
  a.bias_chunk = g.bias_chunk;
}

// Row tiles per workgroup: the most (of 1, 2) that still leaves every CU a workgroup - doubling halves the bytes streamed
// from L2 per FLOP (every workgroup streams the whole network) - and 4 once the grid has several rounds of workgroups anyway
// (measured, policy + value launch: 8192 envs 52.1 us against 58.5 at two row tiles, 16384: 97.5 / 112.5, 32768: 187.5 / 219.6).
int pick_row_tiles(long long rows_total_blocks16) {
  if (rows_total_blocks16 / 4 >= 256) return 4;  // large batches: several rounds of workgroups - each round streams the weights again
  return rows_total_blocks16 / 2 >= 256 ? 2 : 1;
}

int input_kind(const MlpArgs& a) {
  if (a.in_magic) return a.x_bf16 ? IN_BF16X4 : IN_F32X4;
  return a.in_magic2 ? IN_F32X2 : IN_ANY;
}
template <int RT>
void launch_rt(const DualArgs& d, bool elu, bool gate, int in, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {  // more than the default 64 KB of dynamic LDS
    attr_set = true;
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<RT, -1, IN_ANY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<RT, LT_ACT_ELU, IN_ANY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<RT, LT_ACT_ELU, IN_F32X4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<RT, LT_ACT_ELU, IN_BF16X4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<RT, LT_ACT_ELU, IN_F32X2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)lt_mlp_kernel<RT, KIND_GATE, IN_ANY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (gate) hipLaunchKernelGGL((lt_mlp_kernel<RT, KIND_GATE, IN_ANY>), grid, block, lds, s, d);
  else if (!elu) hipLaunchKernelGGL((lt_mlp_kernel<RT, -1, IN_ANY>), grid, block, lds, s, d);
  else if (in == IN_F32X4) hipLaunchKernelGGL((lt_mlp_kernel<RT, LT_ACT_ELU, IN_F32X4>), grid, block, lds, s, d);
  else if (in == IN_BF16X4) hipLaunchKernelGGL((lt_mlp_kernel<RT, LT_ACT_ELU, IN_BF16X4>), grid, block, lds, s, d);
  else if (in == IN_F32X2) hipLaunchKernelGGL((lt_mlp_kernel<RT, LT_ACT_ELU, IN_F32X2>), grid, block, lds, s, d);
  else hipLaunchKernelGGL((lt_mlp_kernel<RT, LT_ACT_ELU, IN_ANY>), grid, block, lds, s, d);
}

// row tiles per workgroup and the LDS bytes of a launch of `nets` networks (their shared LDS geometry is written back into d)
int launch_shape(DualArgs& d, int nets, size_t* lds_out) {
  int stride = d.net[0].stride, bias = d.net[0].bias_total;
  if (nets == 2 && d.net[1].stride > stride) stride = d.net[1].stride;
  if (nets == 2 && d.net[1].bias_total > bias) bias = d.net[1].bias_total;
  d.net[0].stride = stride;  // both networks of a launch share one LDS geometry
  if (nets == 2) d.net[1].stride = stride;
  const size_t row_bytes = (size_t)stride * sizeof(float);
  const size_t bias_bytes = (size_t)bias * sizeof(float);
  const long long t0 = (d.net[0].m + 15) / 16, t1 = nets == 2 ? (d.net[1].m + 15) / 16 : 0;
  int rt = pick_row_tiles(t0 + t1);
  if (const char* o = getenv("LT_MLP_ROW_TILES")) rt = atoi(o) == 4 ? 4 : (atoi(o) == 2 ? 2 : 1);  // diagnostic override
  // one workgroup's activations (+ the policy head's [rows][12 draws + 3 log-density partials + pad] block) must fit the LDS
  while (rt > 1 && (size_t)16 * rt * row_bytes + bias_bytes + (size_t)16 * rt * 64 > 160 * 1024) rt /= 2;
  *lds_out = (size_t)16 * rt * row_bytes + bias_bytes + (size_t)16 * rt * 64;
  for (int n = 0; n < nets; ++n) d.net[n].noise_off = 16 * rt * stride + bias;
  return rt;
}

int launch(DualArgs& d, int nets, hipStream_t s) {
  size_t lds;
  const int rt = launch_shape(d, nets, &lds);
  const long long t0 = (d.net[0].m + 15) / 16, t1 = nets == 2 ? (d.net[1].m + 15) / 16 : 0;
  const long long b0 = (t0 + rt - 1) / rt, b1 = (t1 + rt - 1) / rt;
  d.split = (int)b0;
  // Two networks of equal row count: split them by XCD instead of by block range.  Each XCD's 4 MiB L2 then holds ONE
  // network's weight stream (~1.5 MB) instead of both - with both resident plus the observation rows streaming through,
  // the L2 thrashes and the launch runs at half the stream rate.  (Round-robin dealing of blocks over XCDs is observed
  // behaviour used for speed only; any placement computes the same result.)
  d.xcd_split = (nets == 2 && b0 == b1) ? 1 : 0;
  d.blocks_per_net = (int)b0;
  const long long nblocks = d.xcd_split ? (b0 + 3) / 4 * 8 : b0 + b1;
  const dim3 grid((unsigned)nblocks), block(64 * NW);
  const bool gate = d.net[0].mode == MODE_BACKWARD;
  const bool elu = d.net[0].activation == LT_ACT_ELU && (nets == 1 || d.net[1].activation == LT_ACT_ELU);
  // the input-staging form as a compile-time constant when both networks of the launch take the same one
  int in = input_kind(d.net[0]);
  if (nets == 2 && input_kind(d.net[1]) != in) in = IN_ANY;
  if (rt == 4) {
    launch_rt<4>(d, elu, gate, in, grid, block, lds, s);
  } else if (rt == 2) launch_rt<2>(d, elu, gate, in, grid, block, lds, s);
  else launch_rt<1>(d, elu, gate, in, grid, block, lds, s);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

// The chain of input gradients of a forward network `fwd` (L layers) as a network of its own: L - 1 layers, widths
// [out, h_{L-2}, ..., h_0], layer j multiplies by W_{L-1-j}^T (no bias) and its output is gated by ELU'(a_{L-2-j}).
bool backward_desc(const lt_mlp_desc* fwd, lt_mlp_desc* bd) {
  if (!desc_ok(fwd) || fwd->num_layers < 2 || fwd->activation != LT_ACT_ELU || fwd->dims[fwd->num_layers] > 64) return false;
  const int L = fwd->num_layers;
  *bd = *fwd;
  bd->num_layers = L - 1;
  bd->input_format = LT_ROWS_F32;
  for (int j = 0; j <= L - 1; ++j) bd->dims[j] = fwd->dims[L - j];
  for (int j = 1; j <= L - 1; ++j)
    if (bd->dims[j] & 7) return false;  // gate_pass works on groups of 8 features
  return desc_ok(bd);
}

void fill_backward(const lt_mlp_desc* bd, int Lf, const float* packed, const float* dy, int64_t m, const float* const* acts, float* const* dz,
                   float* const* amax, float* sat_count, MlpArgs& a) {
  fill_args(bd, a);
  a.mode = MODE_BACKWARD;
  a.packed = packed; a.x = dy; a.m = m; a.y = nullptr; a.sat_count = sat_count;
  for (int j = 0; j < bd->num_layers; ++j) {
    a.gate_in[j] = acts[Lf - 2 - j];
    a.act_out[j] = dz[Lf - 2 - j];
    a.amax_out[j] = amax[Lf - 2 - j];
  }
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


// PackArgs of every layer of a network's forward stream / backward (transposed) stream; returns the largest thread count a layer needs
long long fill_pack_forward(const lt_mlp_desc* desc, const float* const* weights, const float* const* biases, float* packed, PackArgs* out) {
  const Geometry g = geometry(desc);
  long long bias_off = 0, most = 0;
  for (int l = 0; l < desc->num_layers; ++l) {
    PackArgs& p = out[l];
    p = PackArgs{};
    p.w = weights[l]; p.b = biases[l]; p.K = desc->dims[l]; p.N = desc->dims[l + 1]; p.packed = packed;
    for (int w = 0; w < NW; ++w) p.chunk_off[w] = g.layer_off[l][w];
    p.plan = plan_of(desc->dims, desc->num_layers, l);
    p.bias_float_off = g.bias_chunk * 256 + bias_off;
    bias_off += pad16(p.N);
    long long total = (long long)p.plan.waves * layer_chunks(p.plan) * 64;
    total = total < pad16(p.N) ? pad16(p.N) : total;  // (the bias slice is written by the first pad16(N) threads)
    most = total > most ? total : most;
  }
  return most;
}
long long fill_pack_backward(const lt_mlp_desc* fwd, const lt_mlp_desc& bd, const float* const* weights, float* packed, PackArgs* out) {
  const Geometry g = geometry(&bd);
  const int L = fwd->num_layers;
  long long bias_off = 0, most = 0;
  for (int j = 0; j < bd.num_layers; ++j) {
    PackArgs& p = out[j];
    p = PackArgs{};
    p.w = weights[L - 1 - j]; p.b = nullptr; p.K = bd.dims[j]; p.N = bd.dims[j + 1]; p.transposed = 1; p.packed = packed;
    for (int w = 0; w < NW; ++w) p.chunk_off[w] = g.layer_off[j][w];
    p.plan = plan_of(bd.dims, bd.num_layers, j);
    p.bias_float_off = g.bias_chunk * 256 + bias_off;
    bias_off += pad16(p.N);
    long long total = (long long)p.plan.waves * layer_chunks(p.plan) * 64;
    total = total < pad16(p.N) ? pad16(p.N) : total;
    most = total > most ? total : most;
  }
  return most;
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
  long long bias_off = 0, most = 0;
  PackAll all = {};
  for (int l = 0; l < desc->num_layers; ++l) {
    if (!weights[l] || !biases[l]) { lt_set_error("lt_mlp_pack: null layer pointer"); return LT_EINVAL; }
    PackArgs& p = all.layer[l];
    p.w = weights[l]; p.b = biases[l]; p.K = desc->dims[l]; p.N = desc->dims[l + 1]; p.packed = packed;
    for (int w = 0; w < NW; ++w) p.chunk_off[w] = g.layer_off[l][w];
    p.plan = plan_of(desc->dims, desc->num_layers, l);
    p.bias_float_off = g.bias_chunk * 256 + bias_off;
    bias_off += pad16(p.N);
    long long total = (long long)p.plan.waves * layer_chunks(p.plan) * 64;
    total = total < pad16(p.N) ? pad16(p.N) : total;  // (the bias slice is written by the first pad16(N) threads)
    most = total > most ? total : most;
  }
  hipLaunchKernelGGL(lt_mlp_pack_kernel<PackAll>, dim3((unsigned)((most + 255) / 256), (unsigned)desc->num_layers), dim3(256), 0, (hipStream_t)stream, all);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
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

// Training forward of two networks in one launch (PPO update: actor on the observations, critic on the critic observations):
// outputs y0 [m][dims0[L0]], y1 [m][dims1[L1]] and, for the backward pass, the activations behind every hidden layer
// (acts0[l], acts1[l]: [m][dims[l + 1]], l < L - 1; hidden widths must be multiples of 4).
int lt_mlp_forward_pair(const lt_mlp_desc* d0, const float* packed0, const float* x0, const lt_mlp_desc* d1, const float* packed1, const float* x1,
                        int64_t m, float* y0, float* y1, float* const* acts0, float* const* acts1, int acts_split, void* stream) {
  if (!desc_ok(d0) || !desc_ok(d1) || !packed0 || !packed1 || !x0 || !x1 || !y0 || !y1 || !acts0 || !acts1 || m <= 0) {
    lt_set_error("lt_mlp_forward_pair: invalid argument");
    return LT_EINVAL;
  }
  DualArgs d = {};
  const lt_mlp_desc* ds[2] = {d0, d1};
  const float* pk[2] = {packed0, packed1};
  const float* xs[2] = {x0, x1};
  float* ys[2] = {y0, y1};
  float* const* as[2] = {acts0, acts1};
  for (int k = 0; k < 2; ++k) {
    fill_args(ds[k], d.net[k]);
    d.net[k].mode = MODE_FORWARD;
    d.net[k].packed = pk[k]; d.net[k].x = xs[k]; d.net[k].m = m; d.net[k].y = ys[k]; d.net[k].acts_split = acts_split != 0;
    for (int l = 0; l + 1 < ds[k]->num_layers; ++l) {
      if (!as[k][l] || (ds[k]->dims[l + 1] & 3)) { lt_set_error("lt_mlp_forward_pair: hidden widths must be multiples of 4 and every activation buffer given"); return LT_EINVAL; }
      d.net[k].act_out[l] = as[k][l];
    }
  }
  return launch(d, 2, (hipStream_t)stream);
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


// ---- the backward data path of the PPO update: dz_l = (dz_{l+1} W_{l+1}) * ELU'(a_l) for every hidden layer, one launch -------------
// loco_rl/loco_rl/algorithms/ppo.py:283-289 (loss.backward()): autograd runs, per network, three GEMMs dz @ W and three ELU-backward
// kernels over [24576][512 / 256 / 128] - 150 us of f32 library GEMMs and 95 us of HBM-bound elementwise passes per optimizer step.
// The chain is a forward pass of a network with the transposed weights (backward_desc), so it runs through the same kernel:
// gradients stay in LDS between layers, the gate is applied where the sums leave the accumulators, each dz_l is written once (the
// weight-gradient kernel lt_wgrad reads it) together with the per-workgroup max |dz_l| lt_wgrad scales by.
int lt_mlp_backward_packed_floats(const lt_mlp_desc* fwd, size_t* floats) {
  lt_mlp_desc bd;
  if (!fwd || !floats || !backward_desc(fwd, &bd)) { lt_set_error("lt_mlp_backward_packed_floats: needs an ELU network of >= 2 layers, hidden widths multiples of 8, <= 64 outputs"); return LT_EINVAL; }
  *floats = (size_t)geometry(&bd).total_chunks * 256;
  return LT_OK;
}

int lt_mlp_pack_backward(const lt_mlp_desc* fwd, const float* const* weights, float* packed, void* stream) {
  lt_mlp_desc bd;
  if (!fwd || !weights || !packed || !backward_desc(fwd, &bd)) { lt_set_error("lt_mlp_pack_backward: invalid argument"); return LT_EINVAL; }
  const Geometry g = geometry(&bd);
  const int L = fwd->num_layers;
  long long bias_off = 0, most = 0;
  PackAll all = {};
  for (int j = 0; j < bd.num_layers; ++j) {
    if (!weights[L - 1 - j]) { lt_set_error("lt_mlp_pack_backward: null layer pointer"); return LT_EINVAL; }
    PackArgs& p = all.layer[j];
    p.w = weights[L - 1 - j]; p.b = nullptr; p.K = bd.dims[j]; p.N = bd.dims[j + 1]; p.transposed = 1; p.packed = packed;
    for (int w = 0; w < NW; ++w) p.chunk_off[w] = g.layer_off[j][w];
    p.plan = plan_of(bd.dims, bd.num_layers, j);
    p.bias_float_off = g.bias_chunk * 256 + bias_off;
    bias_off += pad16(p.N);
    long long total = (long long)p.plan.waves * layer_chunks(p.plan) * 64;
    total = total < pad16(p.N) ? pad16(p.N) : total;
    most = total > most ? total : most;
  }
  hipLaunchKernelGGL(lt_mlp_pack_kernel<PackAll>, dim3((unsigned)((most + 255) / 256), (unsigned)bd.num_layers), dim3(256), 0, (hipStream_t)stream, all);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

// workgroups per network of lt_mlp_backward_pair over m rows = entries of every amax array it writes
int64_t lt_mlp_backward_blocks(const lt_mlp_desc* fwd0, const lt_mlp_desc* fwd1, int64_t m) {
  lt_mlp_desc b0, b1;
  if (!fwd0 || !fwd1 || m <= 0 || !backward_desc(fwd0, &b0) || !backward_desc(fwd1, &b1)) return 0;
  DualArgs d = {};
  fill_args(&b0, d.net[0]);
  fill_args(&b1, d.net[1]);
  d.net[0].m = d.net[1].m = m;
  size_t lds;
  const int rt = launch_shape(d, 2, &lds);
  return ((m + 15) / 16 + rt - 1) / rt;
}

// dy0 [m][out0], dy1 [m][out1]: gradients w.r.t. the two networks' outputs.  acts*[l]: the forward activations behind hidden layer
// l (lt_mlp_forward_pair); dz*[l]: OUT, the gradient w.r.t. hidden layer l's pre-activation, [m][dims[l + 1]]; amax*[l]: OUT,
// lt_mlp_backward_blocks() floats, the per-workgroup max |dz_l|.  sat_count (optional): += 1 per workgroup and layer whose scaled
// gradients reached the f16 image's bound (LT_MLP_INPUT_CLAMP) - the result is then saturated, not exact.
int lt_mlp_backward_pair(const lt_mlp_desc* fwd0, const float* bpacked0, const float* dy0, const float* const* acts0, float* const* dz0, float* const* amax0,
                         const lt_mlp_desc* fwd1, const float* bpacked1, const float* dy1, const float* const* acts1, float* const* dz1, float* const* amax1,
                         int64_t m, int acts_split, const float* in_amax0, const float* in_amax1, int dz_split, float* scales_out, float* sat_count,
                         void* stream) {
  lt_mlp_desc bd[2];
  const float* ia[2] = {in_amax0, in_amax1};
  if (dz_split && (!in_amax0 || !in_amax1 || !scales_out)) { lt_set_error("lt_mlp_backward_pair: a split dz needs the global input maxima and scales_out"); return LT_EINVAL; }
  const lt_mlp_desc* fw[2] = {fwd0, fwd1};
  const float* pk[2] = {bpacked0, bpacked1};
  const float* dy[2] = {dy0, dy1};
  const float* const* ac[2] = {acts0, acts1};
  float* const* dz[2] = {dz0, dz1};
  float* const* am[2] = {amax0, amax1};
  DualArgs d = {};
  for (int k = 0; k < 2; ++k) {
    if (!fw[k] || !pk[k] || !dy[k] || !ac[k] || !dz[k] || !am[k] || m <= 0 || !backward_desc(fw[k], &bd[k])) {
      lt_set_error("lt_mlp_backward_pair: invalid argument");
      return LT_EINVAL;
    }
    for (int l = 0; l + 1 < fw[k]->num_layers; ++l)
      if (!ac[k][l] || !dz[k][l] || !am[k][l]) { lt_set_error("lt_mlp_backward_pair: null layer buffer"); return LT_EINVAL; }
    fill_backward(&bd[k], fw[k]->num_layers, pk[k], dy[k], m, ac[k], dz[k], am[k], sat_count, d.net[k]);
    d.net[k].acts_split = acts_split != 0;
    d.net[k].in_amax = ia[k]; d.net[k].dz_split = dz_split != 0; d.net[k].scale_out = scales_out ? scales_out + k : nullptr;
  }
  return launch(d, 2, (hipStream_t)stream);
}


// Everything a training step needs packed, in ONE launch: the forward streams of both networks (lt_mlp_pack) and, where the network
// qualifies (lt_mlp_backward_packed_floats), their transposed streams for lt_mlp_backward_pair (bpacked* may be NULL: forward only).
// The optimizer moves the weights at every step, so this runs 20 times per PPO update: four launches of ~5 us were 3 % of a step.
int lt_mlp_pack_training(const lt_mlp_desc* d0, const float* const* weights0, const float* const* biases0, float* packed0, float* bpacked0,
                         const lt_mlp_desc* d1, const float* const* weights1, const float* const* biases1, float* packed1, float* bpacked1, void* stream) {
  const lt_mlp_desc* ds[2] = {d0, d1};
  const float* const* ws[2] = {weights0, weights1};
  const float* const* bs[2] = {biases0, biases1};
  float* pk[2] = {packed0, packed1};
  float* bp[2] = {bpacked0, bpacked1};
  PackMany all = {};
  int n = 0;
  long long most = 0;
  for (int k = 0; k < 2; ++k) {
    if (!desc_ok(ds[k]) || !ws[k] || !bs[k] || !pk[k]) { lt_set_error("lt_mlp_pack_training: invalid argument"); return LT_EINVAL; }
    for (int l = 0; l < ds[k]->num_layers; ++l)
      if (!ws[k][l] || !bs[k][l]) { lt_set_error("lt_mlp_pack_training: null layer pointer"); return LT_EINVAL; }
    const long long m = fill_pack_forward(ds[k], ws[k], bs[k], pk[k], all.layer + n);
    most = m > most ? m : most;
    n += ds[k]->num_layers;
    if (bp[k]) {
      lt_mlp_desc bd;
      if (!backward_desc(ds[k], &bd)) { lt_set_error("lt_mlp_pack_training: the network has no backward stream (lt_mlp_backward_packed_floats)"); return LT_EINVAL; }
      const long long mb = fill_pack_backward(ds[k], bd, ws[k], bp[k], all.layer + n);
      most = mb > most ? mb : most;
      n += bd.num_layers;
    }
  }
  hipLaunchKernelGGL(lt_mlp_pack_kernel<PackMany>, dim3((unsigned)((most + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, all);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

}  // extern "C"
