// Tactile observation pass of the student tasks: LT_F_PLATE_SAMPLES -> taxel normal forces -> BinaryTactileSignals.
//
// Reference: locotouch/mdp/observations.py:95-199 (TactileSignals: thresholds, dropout, addition), :281-308 (binary map, two
// identical channels), cfg object_transport_student_env_cfg.py:13-43.  The reference reads one net contact force per taxel
// body from PhysX [DEP]; here the cylinder-on-plate contact is the 4-sample line contact of the step kernel, so the taxel
// forces are restated from it (DESIGN.md "Tactile model"): the samples carry a piecewise-linear line pressure whose cell
// integrals are the sample forces, and a taxel's force is the integral of that pressure over the part of the contact line
// inside its collision box (boxes overlap, as in the URDF: a point can load two neighbours).
//
// One workgroup per env, one thread per taxel: 221 of 256 lanes busy, 12 broadcast loads + 2-3 Philox calls per lane and term,
// 884-byte coalesced row stores.  HBM-bound and tiny (1.8 - 10.6 KB per env per step).  All TactileSignals classes of the
// reference are served (:248-429): `cfg.tactile_format` picks the `tactile` group's, `cfg.tactile_aux_groups` adds the -Play- env's
// `original_tactile` / `processed_tactile`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lt_device_math.h"
#include "lt_env.h"
#include "lt_go1_model.h"
#include "lt_internal.h"
#include "lt_layout.h"

namespace {
using namespace lt;

enum { RS_TACTILE_THR = 0x400, RS_TACTILE = 0x500 };  // stream ids (see lt_env.hip): + 0x40 / 0x200 per term, + taxel / 4, + 2 * taxel + {0, 1}

// cumulative integral of the line pressure from the first sample to arc length s (0 <= s <= 3 * dl)
__device__ __forceinline__ float pressure_integral(const float (&p)[4], float dl, float s) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float t = s - (float)k * dl;
    t = t < 0.f ? 0.f : (t > dl ? dl : t);
    acc += t * (p[k] + (p[k + 1] - p[k]) * t / (2.f * dl));
  }
  return acc;
}

__device__ __forceinline__ float taxel_force(const float (&x)[4], const float (&y)[4], const float (&f)[4], float cx, float cy) {
  const float dx = x[3] - x[0], dy = y[3] - y[0];
  const float len = sqrtf(dx * dx + dy * dy);
  const float ftot = f[0] + f[1] + f[2] + f[3];
  if (!(ftot > 0.f)) return 0.f;
  if (len < 1e-6f)  // the contact line has collapsed to a point (cylinder axis along the plate normal)
    return (fabsf(x[0] - cx) <= LT_TAXEL_HALF_X && fabsf(y[0] - cy) <= LT_TAXEL_HALF_Y) ? ftot : 0.f;
  // clip the segment P(t) = P0 + t * (P3 - P0), t in [0, 1], against the taxel box (Liang-Barsky)
  float t0 = 0.f, t1 = 1.f;
  const float pp[2] = {x[0] - cx, y[0] - cy}, dd[2] = {dx, dy}, hh[2] = {LT_TAXEL_HALF_X, LT_TAXEL_HALF_Y};
#pragma unroll
  for (int ax = 0; ax < 2; ++ax) {
    if (fabsf(dd[ax]) < 1e-9f) {
      if (fabsf(pp[ax]) > hh[ax]) return 0.f;
    } else {
      float ta = (-hh[ax] - pp[ax]) / dd[ax], tb = (hh[ax] - pp[ax]) / dd[ax];
      if (ta > tb) { const float t = ta; ta = tb; tb = t; }
      t0 = ta > t0 ? ta : t0;
      t1 = tb < t1 ? tb : t1;
    }
  }
  if (!(t1 > t0)) return 0.f;
  // node pressures: the end cells are half as long as the inner ones
  const float dl = len / 3.f;
  const float p[4] = {f[0] / (0.5f * dl), f[1] / dl, f[2] / dl, f[3] / (0.5f * dl)};
  // (the whole line integrates to f0 + f1 + f2 + f3)
  return pressure_integral(p, dl, t1 * len) - pressure_integral(p, dl, t0 * len);
}

// One TactileSignals term of the env (observations.py:95-199, 201-246): thresholds -> contact map -> dropout / addition / force
// noise -> normalised force -> per-env min-max normalisation -> discretisation with level noise, written in `format`'s channel
// layout.  Every term instance of the reference draws its own thresholds and noise, hence the per-term stream bases.
struct TermOut { float contact, norm, minmax, disc; };

__device__ __forceinline__ float pick4(const U4& u, int k) { return k == 0 ? u.a : (k == 1 ? u.b : (k == 2 ? u.c : u.d)); }

// block-wide (min, max) over the 221 taxel threads; `s` is [2][4] floats of LDS.  Two barriers; every thread of the block calls.
__device__ __forceinline__ void block_min_max(float v, bool active, float* s, float& mn, float& mx) {
  float lo = active ? v : 2.f, hi = active ? v : -1.f;  // v is in [0, 1] (finite-math build: no infinities)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  __syncthreads();  // (the previous term's readers are done with `s`)
  if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = lo; s[4 + (threadIdx.x >> 6)] = hi; }
  __syncthreads();
  mn = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
  mx = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
}

__device__ __forceinline__ TermOut tactile_term(const lt_cfg& c, int term, bool original, float force, int t, bool active, uint32_t ekey,
                                                uint64_t step, float* s_red) {
  // observations.py:121-126: threshold + U(n_min, n_max) per (env, taxel), drawn once at construction
  const U4 ut = rng4(c.seed, ekey, ~0ull, RS_TACTILE_THR + 0x40u * (uint32_t)term + (uint32_t)(t >> 2));
  const float n_min = -c.tactile_threshold_noise, n_max = c.tactile_threshold_noise;
  const float thr = c.tactile_threshold + (pick4(ut, t & 3) * (n_max - n_min) + n_min);
  bool contact = force > thr;                                                                    // :158
  float f = force;
  float u_level;
  if (original) {  // TactileSignals.__call__ (:248-279): the raw reading; only the level noise is drawn
    u_level = rng4(c.seed, ekey, step, RS_TACTILE + 0x200u * (uint32_t)term + 2u * (uint32_t)t + 1u).c;
  } else {
    const U4 ua = rng4(c.seed, ekey, step, RS_TACTILE + 0x200u * (uint32_t)term + 2u * (uint32_t)t);
    if (c.tactile_dropout_prob > 0.f && contact && ua.a < c.tactile_dropout_prob) { f = ua.c * thr; contact = false; }                // :168-173
    if (c.tactile_addition_prob > 0.f && !contact && ua.b < c.tactile_addition_prob) { f = thr * (1.f + 0.2f * ua.d); contact = true; }  // :177-183
    U4 ub = {0.f, 0.f, 0.f, 0.f};
    if (c.tactile_format != LT_TACTILE_BINARY || term != 0) ub = rng4(c.seed, ekey, step, RS_TACTILE + 0x200u * (uint32_t)term + 2u * (uint32_t)t + 1u);
    if (c.tactile_force_noise > 0.f) {                                                                                                // :186-191
      const float p_min = -c.tactile_force_noise, p_max = c.tactile_force_noise;
      if (contact) f *= 1.f + (ub.a * (p_max - p_min) + p_min);
      f = fmaxf(f, 0.f);
      if (contact && f < thr) f = thr * (1.f + 0.2f * ub.b);
    }
    u_level = ub.c;
  }
  TermOut o;
  o.contact = contact ? 1.f : 0.f;
  o.norm = fminf(fmaxf(__fdiv_rn(f, c.tactile_maximal_force), 0.f), 1.f);                        // :199-203 (not masked by the contact map)
  const float valid = contact ? o.norm : 0.f;                                                    // :205-222
  float mn, mx;
  block_min_max(valid, active, s_red, mn, mx);
  const float range = (mx - mn) > 0.f ? (mx - mn) : 1.f;
  o.minmax = fminf(fmaxf(__fdiv_rn(valid - mn, range), 0.f), 1.f);
  const float bin = __fdiv_rn(1.f, (float)c.tactile_total_levels);                               // :224-235
  float d = rintf(__fdiv_rn(o.minmax, bin));                                                     // torch.round: half to even
  if (c.tactile_level_noise > 0.f) d += u_level * (c.tactile_level_noise - (-c.tactile_level_noise)) + (-c.tactile_level_noise);
  d = fminf(fmaxf(d * bin, 0.f), 1.f);
  o.disc = contact ? d : 0.f;
  return o;
}

__global__ __launch_bounds__(256) void lt_tactile_kernel(const lt_dev_args* __restrict__ d, char* arena) {
  const lt_cfg& c = d->cfg;
  const lt_layout& L = d->layout;
  const long long env = blockIdx.x;
  const int nt = LT_TAXEL_ROWS * LT_TAXEL_COLS;
  const bool active = threadIdx.x < nt;
  const int t = active ? (int)threadIdx.x : 0;
  __shared__ float s_red[8];
  const float* const S = (const float*)(arena + L.quad_off[LT_F_PLATE_SAMPLES]);
  float x[4], y[4], f[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = S[env * 4 + k];
    y[k] = S[L.npad * 4 + env * 4 + k];
    f[k] = S[2 * L.npad * 4 + env * 4 + k];
    f[k] = f[k] > 0.f ? f[k] : 0.f;
  }
  const int row = t / LT_TAXEL_COLS, col = t - row * LT_TAXEL_COLS;
  const float cx = LT_TAXEL_X0 - LT_TAXEL_DX * (float)row, cy = LT_TAXEL_Y0 - LT_TAXEL_DY * (float)col;
  const float force = taxel_force(x, y, f, cx, cy);
  const uint32_t ekey = (uint32_t)env + (uint32_t)c.env_index_offset;
  const uint64_t step = (uint64_t)((const long long*)(arena + L.off_counters))[0];
  float* const base = (float*)(arena + L.off_obs_tactile);
  const long long blk = L.npad * (long long)LT_TACTILE_WIDE_DIM;
  {  // group `tactile`
    const int fmt = c.tactile_format;
    const TermOut o = tactile_term(c, 0, fmt == LT_TACTILE_ORIGINAL, force, t, active, ekey, step, s_red);
    if (active) {
      if (fmt == LT_TACTILE_PROCESSED || fmt == LT_TACTILE_ORIGINAL) {
        float* const out = base + env * LT_TACTILE_WIDE_DIM;
        out[t] = o.contact; out[nt + t] = o.norm; out[2 * nt + t] = o.minmax; out[3 * nt + t] = o.disc;
      } else {
        float* const out = base + env * LT_TACTILE_DIM;
        out[t] = o.contact;                                                                      // :307-308: (contact, second channel)
        out[nt + t] = fmt == LT_TACTILE_BINARY ? o.contact : (fmt == LT_TACTILE_NORMALIZED ? o.minmax : (fmt == LT_TACTILE_DISCRETE ? o.disc : o.norm));
      }
    }
  }
  if (c.tactile_aux_groups & 1) {  // `original_tactile` (student -Play- env)
    const TermOut o = tactile_term(c, 1, true, force, t, active, ekey, step, s_red);
    float* const out = base + blk + env * LT_TACTILE_WIDE_DIM;
    if (active) { out[t] = o.contact; out[nt + t] = o.norm; out[2 * nt + t] = o.minmax; out[3 * nt + t] = o.disc; }
  }
  if (c.tactile_aux_groups & 2) {  // `processed_tactile`
    const TermOut o = tactile_term(c, 2, false, force, t, active, ekey, step, s_red);
    float* const out = base + 2 * blk + env * LT_TACTILE_WIDE_DIM;
    if (active) { out[t] = o.contact; out[nt + t] = o.norm; out[2 * nt + t] = o.minmax; out[3 * nt + t] = o.disc; }
  }
}

}  // namespace

int lt_launch_tactile(const lt_env* env, void* stream) {
  static_assert(LT_TAXEL_ROWS == LT_TACTILE_ROWS && LT_TAXEL_COLS == LT_TACTILE_COLS && LT_TACTILE_DIM == 2 * LT_TAXEL_ROWS * LT_TAXEL_COLS &&
                    LT_TACTILE_WIDE_DIM == 4 * LT_TAXEL_ROWS * LT_TAXEL_COLS,
                "lt_env.h and the URDF-derived taxel grid disagree");
  const lt_dev_args* d = (const lt_dev_args*)((const char*)env->arena + env->layout.off_dev_args);
  hipLaunchKernelGGL(lt_tactile_kernel, dim3((unsigned)env->cfg.num_envs), dim3(256), 0, (hipStream_t)stream, d, (char*)env->arena);
  return (int)hipGetLastError();
}
