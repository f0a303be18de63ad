// Tactile observation pass of the student tasks: LT_F_PLATE_SAMPLES -> taxel normal forces -> BinaryTactileSignals.
//
// Reference: locotouch/mdp/observations.py:95-199 (TactileSignals: thresholds, dropout, addition), :281-308 (binary map, two
// identical channels), cfg object_transport_student_env_cfg.py:13-43.  The reference reads one net contact force per taxel
// body from PhysX [DEP]; here the cylinder-on-plate contact is the 4-sample line contact of the step kernel, so the taxel
// forces are restated from it (DESIGN.md "Tactile model"): the samples carry a piecewise-linear line pressure whose cell
// integrals are the sample forces, and a taxel's force is the integral of that pressure over the part of the contact line
// inside its collision box (boxes overlap, as in the URDF: a point can load two neighbours).
//
// One workgroup per env, one thread per taxel: 221 of 256 lanes busy, 12 broadcast loads + 2 Philox calls per lane, two
// 884-byte coalesced row stores per env.  HBM-bound and tiny (1.8 KB per env per step).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lt_device_math.h"
#include "lt_env.h"
#include "lt_go1_model.h"
#include "lt_internal.h"
#include "lt_layout.h"

namespace {
using namespace lt;

enum { RS_TACTILE_THR = 0x400, RS_TACTILE = 0x500 };  // stream ids: see lt_env.hip

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

__global__ __launch_bounds__(256) void lt_tactile_kernel(const lt_dev_args* __restrict__ d, char* arena) {
  const lt_cfg& c = d->cfg;
  const lt_layout& L = d->layout;
  const long long env = blockIdx.x;
  const int t = threadIdx.x;
  if (t >= LT_TAXEL_ROWS * LT_TAXEL_COLS) return;
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
  // observations.py:121-126: threshold + U(n_min, n_max) per (env, taxel), drawn once at construction
  const U4 ut = rng4(c.seed, ekey, ~0ull, RS_TACTILE_THR + (uint32_t)(t >> 2));
  const float u_thr = (t & 3) == 0 ? ut.a : ((t & 3) == 1 ? ut.b : ((t & 3) == 2 ? ut.c : ut.d));
  const float n_min = -c.tactile_threshold_noise, n_max = c.tactile_threshold_noise;
  const float thr = c.tactile_threshold + (u_thr * (n_max - n_min) + n_min);
  bool contact = force > thr;                                                                    // :158
  const uint64_t step = (uint64_t)((const long long*)(arena + L.off_counters))[0];
  const U4 un = rng4(c.seed, ekey, step, RS_TACTILE + (uint32_t)(t >> 1));
  const float u_drop = (t & 1) ? un.c : un.a, u_add = (t & 1) ? un.d : un.b;
  if (contact && u_drop < c.tactile_dropout_prob) contact = false;                              // :171-175
  if (!contact && u_add < c.tactile_addition_prob) contact = true;                              // :179-184 (after the dropout)
  float* const out = (float*)(arena + L.off_obs_tactile) + env * LT_TACTILE_DIM;
  const float v = contact ? 1.f : 0.f;
  out[t] = v;                                                                                   // :307-308: two channels
  out[LT_TAXEL_ROWS * LT_TAXEL_COLS + t] = v;
}

}  // namespace

int lt_launch_tactile(const lt_env* env, void* stream) {
  static_assert(LT_TAXEL_ROWS == LT_TACTILE_ROWS && LT_TAXEL_COLS == LT_TACTILE_COLS && LT_TACTILE_DIM == 2 * LT_TAXEL_ROWS * LT_TAXEL_COLS,
                "lt_env.h and the URDF-derived taxel grid disagree");
  const lt_dev_args* d = (const lt_dev_args*)((const char*)env->arena + env->layout.off_dev_args);
  hipLaunchKernelGGL(lt_tactile_kernel, dim3((unsigned)env->cfg.num_envs), dim3(256), 0, (hipStream_t)stream, d, (char*)env->arena);
  return (int)hipGetLastError();
}
