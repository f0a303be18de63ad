// lt_wgrad.hip - weight gradient of a Linear layer over a PPO minibatch, f32-equivalent, on the f16 matrix cores.
//
// Reference: the backward pass of the reference's ActorCritic MLPs inside PPO.update (loco_rl/loco_rl/algorithms/ppo.py:316,
// `loss.backward()`): for every Linear layer  dW[n][k] = sum_m dz[m][n] x[m][k]  over the minibatch (M = 24 576 rows at 4096
// envs), N <= 512 outputs, K <= 512 inputs.  As library GEMMs (split-K batched, rocBLAS / hipBLASLt f32) these five shapes per
// network take 120 TFLOP/s at best - 142 us per network and minibatch step, the largest block of the PPO update
// (profiles/r04_ppo_step_timeline.txt).  gfx950 has no TF32-like mode: the f32-input MFMA runs at the f32 vector rate.
//
// Arithmetic (the same operand split as lt_mlp.hip): every f32 operand is x = hi + lo / 64 with hi = f16(x), lo = f16(64 (x - hi));
//     sum_m a b  =  sum a_hi b_hi  +  (sum a_hi b_lo + sum a_lo b_hi) / 64          (the lo lo term is 2^-22 relative: dropped)
// three f16 MFMAs (v_mfma_f32_16x16x32_f16, f32 accumulation, exact products) into ONE accumulator that holds 64 x the sum:
//     acc += (64 a_hi) b_hi + a_hi b_lo + a_lo b_hi                                (64 a_hi is exact: see the scale of dz)
// (two accumulator sets - main and correction - were 128 registers: one wave per SIMD and nobody to hide a fetch behind; with one
// set a wave needs < 256 and two share a SIMD).
// dz is a gradient - 1e-4 ... 1e-9 at the PPO loss's 1 / M scale, far below f16's range: it is multiplied by a power of two `s` that
// brings max |dz| to [2^7, 2^8) before the split (s from the per-block maxima the producing kernel leaves - lt_mlp_backward_pair,
// lt_elu_backward_bias2; elements more than 2^21 below the maximum lose relative precision, at an absolute error 2^-37 of the
// maximum) and the sums are divided by 64 s at the end.  x (observations, ELU outputs) is used as it is: |x| <= 65504.
// Against an f64 reference the result is as close as an f32 GEMM's own rounding (tests/test_hip_wgrad.py).
//
// Structure: NO LDS and no barriers.  The MFMA's A operand wants, per lane (i = lane & 15, g = lane >> 4), 8 values of the REDUCTION
// index for one output row - a COLUMN of the row-major dz.  The order of a reduction is free and so is the order of a tile's rows,
// as long as A and B agree: a wave's 32 x 64 block of dz is fetched by 8 fully coalesced 16-byte loads (load t: lane (c, g) takes row
// m0 + 4 t + g, columns 4 c .. 4 c + 3) and lane (c, g) then HOLDS, for each of its four columns u, the eight rows g, g + 4, ..., g + 28
// - the fragment of reduction indices kk = 8 g + t <-> m = m0 + 4 t + g for output row 4 c + u of MFMA tile u (tile u = the rows that
// are u mod 4).  No shuffle, no transpose: component u of the eight loaded float4s IS the fragment.  The same for B = x.  (A first
// version fetched columns by dword loads - 64 load instructions of 4 x 64 B per step instead of 16 of 1 KiB - and ran at 58 TFLOP/s.)
// A wave owns a 64 x 64 output tile (4 x 4 MFMA tiles, 16 accumulators) over a slice of M; the epilogue's lane holds 4 adjacent
// columns of a row (one from each tile along k): float4 stores.  A workgroup is TWO waves on the two halves of a slice (8 waves
// per CU, two per SIMD: one wave's fetch hides behind the other's MFMAs); the second wave hands its tile over through 16 KiB of
// LDS and the first adds and stores - a fixed order.  The slices' partial tiles go to a slab buffer and one ordered-sum launch
// (lt_partial_sums) adds them - deterministic, no float atomics.  Loads of step s + 1 are in flight while step s is split and
// multiplied.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "lt_env.h"
#include "lt_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int TA = 4, TB = 4;  // MFMA tiles per wave along n (dz columns) and k (x columns): a 64 x 64 output tile
constexpr float LO_SCALE = 64.f;

// x = hi + lo / 64 on component U of eight loaded float4s (pairs: v_cvt_pk_f16_f32, v_pk_add_f32, v_pk_mul_f32)
template <int U>
__device__ __forceinline__ void split8(const f32x4 (&v)[8], float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x;
    x[0] = v[2 * p][U] * scale; x[1] = v[2 * p + 1][U] * scale;
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 d = (x - __builtin_convertvector(h, f32x2)) * LO_SCALE;
    const f16x2 l = __builtin_convertvector(d, f16x2);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

// the same fragments from operands stored in the SPLIT FORMAT (dword = f16 hi | f16 lo << 16; written by the training forward and
// by lt_split_rows): two byte permutes per pair of rows instead of ~7 conversion instructions
template <int U>
__device__ __forceinline__ void unpack8(const f32x4 (&v)[8], f16x8& hi, f16x8& lo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned w0 = __float_as_uint(v[2 * p][U]), w1 = __float_as_uint(v[2 * p + 1][U]);
    h[p] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
    l[p] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
  }
  hi = __builtin_bit_cast(f16x8, h);
  lo = __builtin_bit_cast(f16x8, l);
}

struct WgradArgs {
  const float* dz;      // [M][N]
  const float* x;       // [M][K]
  long long M;
  int N, K;
  int tiles_n, tiles_k, splits;
  const float* amax;    // per-block maxima of |dz| (nblk_amax floats) or nullptr (scale 1)
  int nblk_amax;
  float* slabs;         // [splits][N][K]
  int probe;            // measurements only (LT_WGRAD_PROBE): 1 = fetch the first step's panels only, 2 = no MFMAs, 3 = no LDS staging after the first, 4 = 1 + 3
  int dz_split;         // dz is in the split format AND multiplied by *dz_scale (lt_mlp_backward_pair, dz_split = 1): no conversion, no amax
  const float* dz_scale;
  int x_split;          // x is in the split format (one dword per element: f16 hi | f16 lo << 16, lt_mlp.hip): no conversion here
  float* db;            // optional [splits][N]: the slices' column sums of dz (the bias gradient's partials), by the tiles of the first k column
};

constexpr int WG_WAVES = 4;  // waves per workgroup, each on a quarter of the slice: a slab per FOUR waves (two: twice the slab bytes for lt_partial_sums)
// AS / XS: dz / x arrive in the split format (compile-time: with both forms behind run-time branches the loop spilled 68 bytes)
template <bool AS, bool XS>
__global__ __launch_bounds__(64 * WG_WAVES, 2) void lt_wgrad_kernel(const WgradArgs a) {
  __shared__ f32x4 s_tile[WG_WAVES - 1][TA * TB + 1][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  // block -> (split, tile): the tiles of a slice of M on ONE XCD (blocks are dealt round-robin over the 8 XCDs - observed placement,
  // used for speed only): they sweep the same rows of dz / x at about the same time, so a row is fetched into that XCD's L2 once and
  // read from there by the 48 tiles.  Dealt block by block the tiles of a slice landed on all 8 XCDs and every XCD fetched every
  // row: 8 x 84 MB from the Infinity Cache per launch of the 512 x 348 layer - 149 us at 58 TFLOP/s (f32-equivalent).  XCD x takes
  // the work items [x, x + 1) * gridDim / 8 of the (split-major) list: at most one slice per XCD is shared with a neighbour.
  const int tiles = a.tiles_n * a.tiles_k;
  const int per_xcd = (int)gridDim.x >> 3;
  const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (item >= tiles * a.splits) return;
  const int split = item / tiles, tile = item - split * tiles;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * 16 * TA, k0 = tk * 16 * TB;
  // this slice's 32-row steps (the M / 32 steps are dealt as evenly as possible), a quarter to each wave
  const long long steps = (a.M + 31) / 32;
  const long long b0 = steps * split / a.splits, b1 = steps * (split + 1) / a.splits;
  const long long s0 = b0 + (b1 - b0) * wave / WG_WAVES, s1 = b0 + (b1 - b0) * (wave + 1) / WG_WAVES;
  // scale of dz: a power of two that brings max |dz| to [2^7, 2^8) - 64 x its hi half is still an f16 number
  float scale = 1.f;
  if (AS) {
    scale = *a.dz_scale;
  } else if (a.amax) {
    float m = 0.f;
    for (int b = lane; b < a.nblk_amax; b += 64) m = fmaxf(m, a.amax[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const int e = (int)((__float_as_uint(m) >> 23) & 0xFF) - 127;  // floor(log2 m) (m == 0 or denormal: e = -127)
    int se = 7 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    scale = __uint_as_float((unsigned)(127 + se) << 23);
  }
  f32x4 acc[TA][TB];
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int q = 0; q < TB; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 32 rows x 64 columns per operand and step: load t = row m0 + 4 t + g, 16 bytes at column 4 c (clamped to the matrix: a tile may
  // hang over its edge; what an overhanging lane loads is multiplied into output elements that are never stored)
  f32x4 va[8], vb[8];
  const int ca = min(n0 + 4 * i, a.N - 4), cb = min(k0 + 4 * i, a.K - 4);
  // addresses as 32-bit byte offsets from the (uniform) matrix bases: one add per load and one per step (as 64-bit pointers rebuilt
  // from the row index the 16 loads of a step cost 132 VALU instructions - a third of the loop).  Rows beyond M are redirected to the
  // last valid offset: what they load is multiplied by zeroed dz rows (below).  M x N x 4 bytes < 4 GiB (lt_wgrad checks).
  const unsigned row_a = (unsigned)a.N * 4u, row_b = (unsigned)a.K * 4u;
  const unsigned last_a = (unsigned)(a.M - 1) * row_a + (unsigned)ca * 4u, last_b = (unsigned)(a.M - 1) * row_b + (unsigned)cb * 4u;
  unsigned off_a = ((unsigned)s0 * 32u + (unsigned)g) * row_a + (unsigned)ca * 4u, off_b = ((unsigned)s0 * 32u + (unsigned)g) * row_b + (unsigned)cb * 4u;
  auto issue = [&](long long) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      va[t] = *(const f32x4*)((const char*)a.dz + min(off_a + (unsigned)(4 * t) * row_a, last_a));
      vb[t] = *(const f32x4*)((const char*)a.x + min(off_b + (unsigned)(4 * t) * row_b, last_b));
    }
    off_a += 32u * row_a;
    off_b += 32u * row_b;
  };
  const bool colsum = a.db != nullptr && tk == 0;  // (uniform)
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  if (s0 < s1) issue(s0);
  for (long long s = s0; s < s1; ++s) {
    if ((s + 1) * 32 > a.M) {  // rows beyond M (the last step of a ragged M) must not contribute: their clamped loads repeat row M - 1
#pragma unroll
      for (int t = 0; t < 8; ++t) if (s * 32 + 4 * t + g >= a.M) va[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (colsum) {
      if (AS) {  // (the tiles of the first k column only) value = hi + lo / 64, still scaled
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned w = __float_as_uint(va[t][u]);
            cs[u] += (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)) * (1.f / LO_SCALE);
          }
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) cs += va[t];
      }
    }
    f16x8 ah[TA], al[TA], bh[TB], bl[TB];
    if (AS) {
      unpack8<0>(va, ah[0], al[0]); unpack8<1>(va, ah[1], al[1]); unpack8<2>(va, ah[2], al[2]); unpack8<3>(va, ah[3], al[3]);
    } else {
      split8<0>(va, scale, ah[0], al[0]); split8<1>(va, scale, ah[1], al[1]); split8<2>(va, scale, ah[2], al[2]); split8<3>(va, scale, ah[3], al[3]);
    }
    if (XS) {
      unpack8<0>(vb, bh[0], bl[0]); unpack8<1>(vb, bh[1], bl[1]); unpack8<2>(vb, bh[2], bl[2]); unpack8<3>(vb, bh[3], bl[3]);
    } else {
      split8<0>(vb, 1.f, bh[0], bl[0]); split8<1>(vb, 1.f, bh[1], bl[1]); split8<2>(vb, 1.f, bh[2], bl[2]); split8<3>(vb, 1.f, bh[3], bl[3]);
    }
    if (s + 1 < s1) issue(s + 1);  // the next step's 16 loads fly while this step's 48 MFMAs run
#pragma unroll
    for (int p = 0; p < TA; ++p) {
      const f16x8 a64 = ah[p] * (_Float16)LO_SCALE;  // exact: |scaled dz| < 2^8
#pragma unroll
      for (int q = 0; q < TB; ++q) {
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a64, bh[q], acc[p][q], 0, 0, 0);
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], bl[q], acc[p][q], 0, 0, 0);
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p], bh[q], acc[p][q], 0, 0, 0);
      }
    }
  }
  // waves 1 .. hand their tiles and column sums to wave 0, which adds them in wave order
  if (wave > 0) {
#pragma unroll
    for (int p = 0; p < TA; ++p)
#pragma unroll
      for (int q = 0; q < TB; ++q) s_tile[wave - 1][p * TB + q][lane] = acc[p][q];
    s_tile[wave - 1][TA * TB][lane] = cs;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < WG_WAVES - 1; ++w) {
#pragma unroll
    for (int p = 0; p < TA; ++p)
#pragma unroll
      for (int q = 0; q < TB; ++q) acc[p][q] += s_tile[w][p * TB + q][lane];
    cs += s_tile[w][TA * TB][lane];
  }
  if (colsum) {  // lane (c, g) holds the sums of rows g mod 4 of columns n0 + 4 c .. + 3
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cs[u] += __shfl_xor(cs[u], 16, 64);
      cs[u] += __shfl_xor(cs[u], 32, 64);
    }
    if (AS) cs *= 1.f / scale;
    if (g == 0 && n0 + 4 * i < a.N) *(f32x4*)(a.db + (long long)split * a.N + n0 + 4 * i) = cs;
  }
  // the partial tile: MFMA tile (p, q) holds C[row rho = 4 g + r][col kappa = i] = dW[n0 + 4 rho + p][k0 + 4 kappa + q]: a lane's four
  // q tiles are 4 ADJACENT columns of one row
  const float inv = 1.f / (scale * LO_SCALE);
  float* const out = a.slabs + (long long)split * a.N * a.K;
  const int k = k0 + 4 * i;
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * (4 * g + r) + p;
      if (n < a.N && k < a.K) {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < TB; ++q) o[q] = acc[p][q][r] * inv;
        *(f32x4*)(out + (long long)n * a.K + k) = o;
      }
    }
}

// Both operands in the split format (the production path of the PPO update), NST steps of loads in flight per wave, ONE wave per SIMD:
// with nothing left to convert a wave's step is ~0.5 us of byte permutes and MFMAs behind a ~2 us HBM / L2 round trip, and what
// hides the round trip is loads in flight - two waves x one step in the kernel above, one wave x NST steps here (the stages take the
// registers a second wave would; those beyond 256 live in AGPRs and cost a copy each).
template <int NST>
__global__ __launch_bounds__(128, 1) void lt_wgrad_split_kernel(const WgradArgs a) {
  __shared__ f32x4 s_tile[TA * TB + 1][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int tiles = a.tiles_n * a.tiles_k;
  const int per_xcd = (int)gridDim.x >> 3;
  const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);  // (lt_wgrad_kernel: a slice's tiles on one XCD)
  if (item >= tiles * a.splits) return;
  const int split = item / tiles, tile = item - split * tiles;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * 16 * TA, k0 = tk * 16 * TB;
  const long long steps = (a.M + 31) / 32;
  const long long b0 = steps * split / a.splits, b1 = steps * (split + 1) / a.splits, mid = b0 + (b1 - b0 + 1) / 2;
  const long long s0 = wave ? mid : b0, s1 = wave ? b1 : mid;
  const float scale = *a.dz_scale;
  f32x4 acc[TA][TB];
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int q = 0; q < TB; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct Stage { f32x4 a[8], b[8]; };
  Stage st[NST];
  const int ca = min(n0 + 4 * i, a.N - 4), cb = min(k0 + 4 * i, a.K - 4);
  const unsigned row_a = (unsigned)a.N * 4u, row_b = (unsigned)a.K * 4u;
  const unsigned last_a = (unsigned)(a.M - 1) * row_a + (unsigned)ca * 4u, last_b = (unsigned)(a.M - 1) * row_b + (unsigned)cb * 4u;
  unsigned off_a = ((unsigned)s0 * 32u + (unsigned)g) * row_a + (unsigned)ca * 4u, off_b = ((unsigned)s0 * 32u + (unsigned)g) * row_b + (unsigned)cb * 4u;
  long long issued = s0;
  auto issue = [&](Stage& S) __attribute__((always_inline)) {  // the next step not yet requested
    if (issued >= s1) return;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      S.a[t] = *(const f32x4*)((const char*)a.dz + min(off_a + (unsigned)(4 * t) * row_a, last_a));
      S.b[t] = *(const f32x4*)((const char*)a.x + min(off_b + (unsigned)(4 * t) * row_b, last_b));
    }
    off_a += 32u * row_a;
    off_b += 32u * row_b;
    ++issued;
  };
  const bool colsum = a.db != nullptr && tk == 0;  // (uniform)
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  auto consume = [&](Stage& S, long long s) __attribute__((always_inline)) {
    if ((s + 1) * 32 > a.M) {  // rows beyond M (the last step of a ragged M): zero halves contribute nothing
#pragma unroll
      for (int t = 0; t < 8; ++t) if (s * 32 + 4 * t + g >= a.M) S.a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (colsum) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned w = __float_as_uint(S.a[t][u]);
          cs[u] += (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)) * (1.f / LO_SCALE);
        }
    }
    f16x8 ah[TA], al[TA], bh[TB], bl[TB];
    unpack8<0>(S.a, ah[0], al[0]); unpack8<1>(S.a, ah[1], al[1]); unpack8<2>(S.a, ah[2], al[2]); unpack8<3>(S.a, ah[3], al[3]);
    unpack8<0>(S.b, bh[0], bl[0]); unpack8<1>(S.b, bh[1], bl[1]); unpack8<2>(S.b, bh[2], bl[2]); unpack8<3>(S.b, bh[3], bl[3]);
    issue(S);  // this stage's registers are free again
#pragma unroll
    for (int p = 0; p < TA; ++p) {
      const f16x8 a64 = ah[p] * (_Float16)LO_SCALE;  // exact: |scaled dz| <= LT_MLP_INPUT_CLAMP
#pragma unroll
      for (int q = 0; q < TB; ++q) {
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a64, bh[q], acc[p][q], 0, 0, 0);
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], bl[q], acc[p][q], 0, 0, 0);
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p], bh[q], acc[p][q], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int j = 0; j < NST; ++j) issue(st[j]);
  for (long long s = s0; s < s1; s += NST) {
#pragma unroll
    for (int j = 0; j < NST; ++j)
      if (s + j < s1) consume(st[j], s + j);
  }
  if (wave == 1) {
#pragma unroll
    for (int p = 0; p < TA; ++p)
#pragma unroll
      for (int q = 0; q < TB; ++q) s_tile[p * TB + q][lane] = acc[p][q];
    s_tile[TA * TB][lane] = cs;
  }
  __syncthreads();
  if (wave == 1) return;
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int q = 0; q < TB; ++q) acc[p][q] += s_tile[p * TB + q][lane];
  cs += s_tile[TA * TB][lane];
  if (colsum) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cs[u] += __shfl_xor(cs[u], 16, 64);
      cs[u] += __shfl_xor(cs[u], 32, 64);
    }
    cs *= 1.f / scale;
    if (g == 0 && n0 + 4 * i < a.N) *(f32x4*)(a.db + (long long)split * a.N + n0 + 4 * i) = cs;
  }
  const float inv = 1.f / (scale * LO_SCALE);
  float* const out = a.slabs + (long long)split * a.N * a.K;
  const int k = k0 + 4 * i;
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * (4 * g + r) + p;
      if (n < a.N && k < a.K) {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < TB; ++q) o[q] = acc[p][q][r] * inv;
        *(f32x4*)(out + (long long)n * a.K + k) = o;
      }
    }
}

// ---- the same product on 128 x 128 tiles shared through LDS ---------------------------------------------------------------------------
// The one-wave-per-tile kernel above moves 16 KiB from L2 per wave and step for 48 MFMAs - 8 waves of a CU ask the L2 port for
// 128 KiB per 1536 matrix-pipe cycles, 83 B/clk of a port that delivers ~37 (89 GB/s per CU measured in lt_mlp.hip): 54 us for the
// 512 x 348 layer whose MFMA time is 10 us.  Here a workgroup of four waves (2 x 2) owns a 128 x 128 tile: a step's 32 x 128 panels
// of dz and x enter the CU once (32 KiB per 4 x 48 MFMAs: half the traffic), go through LDS as (hi, lo) dwords - dz split by the
// thread that fetched it (a quarter of the conversions per wave), x copied as it is when it arrives in the split format - and every
// wave reads its two 32 x 64 halves as the same 16-byte (row 4 t + g, columns 4 c ..) pieces the register kernel loads from HBM.
// Two panel stages in LDS (64 KiB per workgroup, two workgroups per CU), one barrier per step.
constexpr int PT = 128;  // panel / tile width
struct Panels { unsigned a[2][32][PT]; unsigned b[2][32][PT]; };

__global__ __launch_bounds__(256, 2) void lt_wgrad128_kernel(const WgradArgs a) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  Panels& S = *reinterpret_cast<Panels*>(s_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;  // this wave's 64 x 64 quarter of the tile
  const int i = lane & 15, g = lane >> 4;
  const int tiles = a.tiles_n * a.tiles_k;
  const int per_xcd = (int)gridDim.x >> 3;
  const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);  // (see lt_wgrad_kernel: a slice's tiles on one XCD)
  if (item >= tiles * a.splits) return;
  const int split = item / tiles, tile = item - split * tiles;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * PT, k0 = tk * PT;
  const long long steps = (a.M + 31) / 32;
  const long long s0 = steps * split / a.splits, s1 = steps * (split + 1) / a.splits;
  float scale = 1.f;
  if (a.dz_split) {
    scale = *a.dz_scale;
  } else if (a.amax) {
    float m = 0.f;
    for (int b = lane; b < a.nblk_amax; b += 64) m = fmaxf(m, a.amax[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const int e = (int)((__float_as_uint(m) >> 23) & 0xFF) - 127;
    int se = 7 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    scale = __uint_as_float((unsigned)(127 + se) << 23);
  }
  f32x4 acc[TA][TB];
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int q = 0; q < TB; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // panel fetch: thread -> rows (tid >> 5) + 8 j, j < 4, columns 4 (tid & 31) .. + 3 of both panels (a wave: two whole 512-byte rows)
  const int prow = tid >> 5, pcol = 4 * (tid & 31);
  const int ca = min(n0 + pcol, a.N - 4), cb = min(k0 + pcol, a.K - 4);  // (overhanging columns re-read the edge: their products are never stored)
  f32x4 ga[4], gb[4];
  auto fetch = [&](long long s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long r = s * 32 + prow + 8 * j;
      r = r < a.M ? r : a.M - 1;
      ga[j] = *(const f32x4*)(a.dz + r * a.N + ca);
      gb[j] = *(const f32x4*)(a.x + r * a.K + cb);
    }
  };
  const bool colsum = a.db != nullptr && tk == 0;  // (uniform)
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  auto to_pair = [&](const f32x4& v, float sc) __attribute__((always_inline)) {  // four values -> four (hi | lo << 16) dwords
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x2 x;
      x[0] = v[2 * p] * sc; x[1] = v[2 * p + 1] * sc;
      const f16x2 h = __builtin_convertvector(x, f16x2);
      const f32x2 d = (x - __builtin_convertvector(h, f32x2)) * LO_SCALE;
      const f16x2 l = __builtin_convertvector(d, f16x2);
      const unsigned hw = __builtin_bit_cast(unsigned, h), lw = __builtin_bit_cast(unsigned, l);
      o[2 * p] = __builtin_amdgcn_perm(lw, hw, 0x05040100u);
      o[2 * p + 1] = __builtin_amdgcn_perm(lw, hw, 0x07060302u);
    }
    return o;
  };
  auto stage = [&](int st, long long s) __attribute__((always_inline)) {  // the fetched registers -> panels of stage st
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = s * 32 + prow + 8 * j < a.M;  // rows beyond M contribute nothing
      const f32x4 va = live ? ga[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (colsum) {
        if (a.dz_split) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned w = __float_as_uint(va[u]);
            cs[u] += (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)) * (1.f / LO_SCALE);
          }
        } else {
          cs += va;
        }
      }
      *(u32x4*)&S.a[st][prow + 8 * j][pcol] = a.dz_split ? __builtin_bit_cast(u32x4, va) : to_pair(va, scale);
      *(u32x4*)&S.b[st][prow + 8 * j][pcol] = a.x_split ? __builtin_bit_cast(u32x4, gb[j]) : to_pair(gb[j], 1.f);
    }
  };
  if (s0 < s1) {
    fetch(s0);
    stage(0, s0);
  }
  __syncthreads();
  int cur = 0;
  for (long long s = s0; s < s1; ++s) {
    if (s + 1 < s1 && a.probe != 1 && a.probe != 4) fetch(s + 1);  // in flight under this step's LDS reads and MFMAs
    f32x4 va[8], vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      va[t] = __builtin_bit_cast(f32x4, *(const u32x4*)&S.a[cur][4 * t + g][64 * wi + 4 * i]);
      vb[t] = __builtin_bit_cast(f32x4, *(const u32x4*)&S.b[cur][4 * t + g][64 * wj + 4 * i]);
    }
    f16x8 bh[TB], bl[TB];
    unpack8<0>(vb, bh[0], bl[0]); unpack8<1>(vb, bh[1], bl[1]); unpack8<2>(vb, bh[2], bl[2]); unpack8<3>(vb, bh[3], bl[3]);
    f16x8 ah[TA], al[TA];
    unpack8<0>(va, ah[0], al[0]); unpack8<1>(va, ah[1], al[1]); unpack8<2>(va, ah[2], al[2]); unpack8<3>(va, ah[3], al[3]);
    if (a.probe == 2) {
#pragma unroll
      for (int p = 0; p < TA; ++p)
#pragma unroll
        for (int q = 0; q < TB; ++q) acc[p][q][0] += (float)ah[p][0] * (float)bh[q][0] + (float)al[p][1] * (float)bl[q][1];
    } else {
#pragma unroll
    for (int p = 0; p < TA; ++p) {
      const f16x8 a64 = ah[p] * (_Float16)LO_SCALE;  // exact: |scaled dz| < 2^8
#pragma unroll
      for (int q = 0; q < TB; ++q) {
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a64, bh[q], acc[p][q], 0, 0, 0);
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p], bl[q], acc[p][q], 0, 0, 0);
        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p], bh[q], acc[p][q], 0, 0, 0);
      }
    }
    }
    if (s + 1 < s1 && a.probe != 3 && a.probe != 4) stage(cur ^ 1, s + 1);  // (issued in front of the MFMAs instead: 53.9 -> 57.7 us)
    __syncthreads();  // stage cur^1 is complete; everybody is done reading stage cur
    cur ^= 1;
  }
  // column sums of dz (bias-gradient partials): a thread holds columns pcol .. + 3 over its rows; the 8 row groups meet in LDS
  if (colsum) {
    f32x4* const red = reinterpret_cast<f32x4*>(s_raw);  // [8][32] (the panels are dead: the loop ended with a barrier)
    red[(tid >> 5) * 32 + (tid & 31)] = cs;
    __syncthreads();
    if (tid < 32) {
      f32x4 t = red[tid];
#pragma unroll
      for (int r = 1; r < 8; ++r) t += red[r * 32 + tid];
      if (a.dz_split) t *= 1.f / scale;
      if (n0 + 4 * tid < a.N) *(f32x4*)(a.db + (long long)split * a.N + n0 + 4 * tid) = t;
    }
  }
  const float inv = 1.f / (scale * LO_SCALE);
  float* const out = a.slabs + (long long)split * a.N * a.K;
  const int k = k0 + 64 * wj + 4 * i;
#pragma unroll
  for (int p = 0; p < TA; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 64 * wi + 4 * (4 * g + r) + p;
      if (n < a.N && k < a.K) {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < TB; ++q) o[q] = acc[p][q][r] * inv;
        *(f32x4*)(out + (long long)n * a.K + k) = o;
      }
    }
}

// Which form runs a shape.  MEASURED (tools/wgrad_bench.py, 24 576 rows, x in the split format; us for 512 x 348 / 256 x 512 / 128 x 256):
// one wave per 64 x 64 tile 52.7 / 35.6 / 14.8, tiled 50.9 / 35.2 / 16.6 with twice the slices (42 / 64 / 192: twice the slab bytes
// for lt_partial_sums) - no gain worth the slabs, so the tiled form is OFF unless LT_WGRAD_TILED=1.  Its probes (LT_WGRAD_PROBE) say
// why: without the panel fetches 49.5 us of 53.9, without the MFMAs 44.8, without the LDS staging 40.0, with neither fetches nor staging
// (fragment reads + unpacking + MFMAs + the barrier alone: what a DMA-fed form of this loop could reach at best) 34.0 - no single phase bounds it;
// LDS reads + unpacking, MFMAs and staging of a wave run one after the other between two barriers, and two workgroups per CU do not
// hide that.  The floor of either form is the 84 MB of operands from HBM (~17 us) and 10.5 us of MFMA time.
bool use_tiled(int N, int K) {
  static const int force = [] { const char* e = getenv("LT_WGRAD_TILED"); return e ? atoi(e) : 0; }();
  return force != 0 && N >= 96 && K >= 96;
}
int pick_splits_tiled(long long M, int tiles) {
  static const int target = [] { const char* e = getenv("LT_WGRAD_TILED_BLOCKS"); return e ? atoi(e) : 512; }();
  const long long steps = (M + 31) / 32;
  long long s = target / tiles;
  if (s > steps / 4) s = steps / 4;
  return (int)(s < 1 ? 1 : s);
}

int pick_splits(long long M, int tiles) {
  // at most 512 four-wave workgroups (two per CU, two waves per SIMD: a 513th block would run alone after the others - 24 slices x
  // 48 tiles = 1152 single-wave blocks took 97 us where 21 x 48 = 1008 took 69 us), at least 2 steps of 32 rows per wave
  static const int target = [] { const char* e = getenv("LT_WGRAD_BLOCKS"); if (e) return atoi(e); const char* d = getenv("LT_WGRAD_DEEP"); return (d && atoi(d) >= 3) ? 512 : 512; }();  // (measurements)
  const long long steps = (M + 31) / 32;
  long long s = target / tiles;
  if (s > steps / 8) s = steps / 8;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" int lt_wgrad_splits(int64_t M, int N, int K) {
  if (use_tiled(N, K)) return pick_splits_tiled((long long)M, ((N + PT - 1) / PT) * ((K + PT - 1) / PT));
  const int tiles = ((N + 16 * TA - 1) / (16 * TA)) * ((K + 16 * TB - 1) / (16 * TB));
  return pick_splits((long long)M, tiles);
}

extern "C" int64_t lt_wgrad_ws_floats(int64_t M, int N, int K) { return (int64_t)lt_wgrad_splits(M, N, K) * N * K; }

extern "C" int lt_wgrad(const float* dz, int dz_split, const float* dz_scale, const float* x, int x_split, int64_t M, int N, int K, const float* amax_blocks,
                        int nblk_amax, float* slabs, float* db_slabs, void* stream) {
  if (!dz || !x || !slabs || M < 1 || N < 4 || K < 4 || (N & 3) || (K & 3) || (amax_blocks && nblk_amax < 1) || (long long)M * (N > K ? N : K) * 4 >= (1ll << 32)) {
    lt_set_error("lt_wgrad: invalid argument (N and K multiples of 4, each operand below 4 GiB)");
    return LT_EINVAL;
  }
  WgradArgs a;
  a.dz = dz; a.x = x; a.M = M; a.N = N; a.K = K;
  a.amax = amax_blocks; a.nblk_amax = amax_blocks ? nblk_amax : 0;
  a.slabs = slabs; a.db = db_slabs; a.x_split = x_split != 0; a.dz_split = dz_split != 0; a.dz_scale = dz_scale;
  if (a.dz_split && !dz_scale) { lt_set_error("lt_wgrad: a split dz needs its scale"); return LT_EINVAL; }
  static const int probe = [] { const char* e = getenv("LT_WGRAD_PROBE"); return e ? atoi(e) : 0; }();
  a.probe = probe;
  if (use_tiled(N, K)) {
    a.tiles_n = (N + PT - 1) / PT;
    a.tiles_k = (K + PT - 1) / PT;
    a.splits = pick_splits_tiled((long long)M, a.tiles_n * a.tiles_k);
    static bool attr_set = false;
    if (!attr_set) { attr_set = true; (void)hipFuncSetAttribute((const void*)lt_wgrad128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Panels)); }
    hipLaunchKernelGGL(lt_wgrad128_kernel, dim3((unsigned)((a.tiles_n * a.tiles_k * a.splits + 7) / 8 * 8)), dim3(256), sizeof(Panels), (hipStream_t)stream, a);
  } else {
    a.tiles_n = (N + 16 * TA - 1) / (16 * TA);
    a.tiles_k = (K + 16 * TB - 1) / (16 * TB);
    a.splits = pick_splits((long long)M, a.tiles_n * a.tiles_k);
    const dim3 grid((unsigned)((a.tiles_n * a.tiles_k * a.splits + 7) / 8 * 8));
    static const int deep = [] { const char* e = getenv("LT_WGRAD_DEEP"); return e ? atoi(e) : 0; }();
    const bool both_split = a.dz_split && a.x_split;
    if (both_split && deep == 3) hipLaunchKernelGGL(lt_wgrad_split_kernel<3>, grid, dim3(128), 0, (hipStream_t)stream, a);
    else if (both_split && deep == 4) hipLaunchKernelGGL(lt_wgrad_split_kernel<4>, grid, dim3(128), 0, (hipStream_t)stream, a);
    else if (both_split && deep == 5) hipLaunchKernelGGL(lt_wgrad_split_kernel<5>, grid, dim3(128), 0, (hipStream_t)stream, a);
    else if (a.dz_split && a.x_split) hipLaunchKernelGGL((lt_wgrad_kernel<true, true>), grid, dim3(64 * WG_WAVES), 0, (hipStream_t)stream, a);
    else if (a.x_split) hipLaunchKernelGGL((lt_wgrad_kernel<false, true>), grid, dim3(64 * WG_WAVES), 0, (hipStream_t)stream, a);
    else if (a.dz_split) hipLaunchKernelGGL((lt_wgrad_kernel<true, false>), grid, dim3(64 * WG_WAVES), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((lt_wgrad_kernel<false, false>), grid, dim3(64 * WG_WAVES), 0, (hipStream_t)stream, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

// Rows in the split format (see unpack8): out[i] = f16 hi | f16 lo << 16 of clamp(x[i], +-LT_MLP_INPUT_CLAMP) - the observation rows of a
// PPO update, once per update (the first layer's weight gradient reads them in each of the 20 optimizer steps; the forward pass
// saturates its input rows at the same bound, so these ARE the values the first layer multiplied).
namespace {
__global__ __launch_bounds__(256) void lt_split_rows_kernel(const float* __restrict__ x, unsigned* __restrict__ out, long long n4) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = ((const f32x4*)x)[i];
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float c = __builtin_amdgcn_fmed3f(v[k], -(float)LT_MLP_INPUT_CLAMP, (float)LT_MLP_INPUT_CLAMP);
    const _Float16 h = (_Float16)c;
    const _Float16 l = (_Float16)((c - (float)h) * LO_SCALE);
    o[k] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
  }
  ((u32x4*)out)[i] = o;
}
}  // namespace

extern "C" int lt_split_rows(const float* x, void* out, int64_t count, void* stream) {
  if (!x || !out || count < 4 || (count & 3)) { lt_set_error("lt_split_rows: invalid argument (count a multiple of 4)"); return LT_EINVAL; }
  hipLaunchKernelGGL(lt_split_rows_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned*)out, (long long)(count / 4));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}
