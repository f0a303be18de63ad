/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Small vector / quaternion / Philox helpers for oracle/lt_oracle.c.
 * Quaternions are wxyz as in isaaclab.utils.math (reference call sites: locotouch/mdp/rewards.py:7,
 * locotouch/mdp/observations.py:8).  IsaacLab itself is absent from the build image, so these restate its
 * documented formulas: parity is pinned at the locotouch.mdp level (tests/golden), unpinned below it.
 */
#ifndef LT_ORACLE_MATH_H
#define LT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef LT_REAL
#define LT_REAL float
#endif
typedef LT_REAL real;

#define LT_PI 3.14159265358979323846

static inline void v3_set(real* o, real x, real y, real z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void v3_copy(real* o, const real* a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3_add(real* o, const real* a, const real* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3_sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3_scale(real* o, const real* a, real s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
static inline void v3_axpy(real* o, real s, const real* a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static inline real v3_dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline real v3_norm(const real* a) { return (real)sqrt((double)v3_dot(a, a)); }
static inline void v3_cross(real* o, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

/* 3x3 row-major */
static inline void m3_mulv(real* o, const real m[9], const real* v) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_tmulv(real* o, const real m[9], const real* v) { /* m^T v */
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_mul(real o[9], const real a[9], const real b[9]) {
  real t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
  memcpy(o, t, sizeof(t));
}
static inline void m3_transpose(real o[9], const real a[9]) {
  real t[9] = {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]};
  memcpy(o, t, sizeof(t));
}
static inline void m3_skew(real o[9], const real* v) {
  o[0] = 0; o[1] = -v[2]; o[2] = v[1];
  o[3] = v[2]; o[4] = 0; o[5] = -v[0];
  o[6] = -v[1]; o[7] = v[0]; o[8] = 0;
}

/* quaternion wxyz -> rotation matrix (body -> world) */
static inline void quat_to_mat(real R[9], const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
static inline void quat_mul(real* o, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static inline void quat_conj(real* o, const real* a) { o[0] = a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = -a[3]; }
/* v' = q v q^-1 (isaaclab quat_apply) */
static inline void quat_apply(real* o, const real* q, const real* v) {
  real t[3], u[3];
  v3_cross(t, q + 1, v);
  v3_scale(t, t, 2);
  v3_cross(u, q + 1, t);
  o[0] = v[0] + q[0] * t[0] + u[0]; o[1] = v[1] + q[0] * t[1] + u[1]; o[2] = v[2] + q[0] * t[2] + u[2];
}
static inline void quat_apply_inv(real* o, const real* q, const real* v) {
  real t[3], u[3];
  v3_cross(t, q + 1, v);
  v3_scale(t, t, 2);
  v3_cross(u, q + 1, t);
  o[0] = v[0] - q[0] * t[0] + u[0]; o[1] = v[1] - q[0] * t[1] + u[1]; o[2] = v[2] - q[0] * t[2] + u[2];
}
static inline void quat_from_euler(real* o, real roll, real pitch, real yaw) {
  real cy = (real)cos(yaw * 0.5), sy = (real)sin(yaw * 0.5);
  real cr = (real)cos(roll * 0.5), sr = (real)sin(roll * 0.5);
  real cp = (real)cos(pitch * 0.5), sp = (real)sin(pitch * 0.5);
  o[0] = cy * cr * cp + sy * sr * sp;
  o[1] = cy * sr * cp - sy * cr * sp;
  o[2] = cy * cr * sp + sy * sr * cp;
  o[3] = sy * cr * cp - cy * sr * sp;
}
/* yaw of isaaclab euler_xyz_from_quat, wrapped to [0, 2*pi) like `% (2*pi)` */
static inline real quat_yaw_2pi(const real* q) {
  real s = 2 * (q[0] * q[3] + q[1] * q[2]);
  real c = 1 - 2 * (q[2] * q[2] + q[3] * q[3]);
  real yaw = (real)atan2(s, c);
  real two_pi = (real)(2 * LT_PI);
  real m = (real)fmod(yaw, two_pi); /* python % : result has the sign of the divisor */
  if (m < 0) m += two_pi;
  return m;
}
static inline void quat_normalize(real* q) {
  real n = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
  real inv = 1 / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}

/* ---- Philox4x32-10 counter-based RNG (Salmon et al., SC'11); bit-exact twin of the HIP kernel's ---- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* 4 uniforms in [0,1) with 24-bit mantissas for (seed, env, step, stream) */
static inline void lt_rng4(uint64_t seed, uint32_t env, uint64_t step, uint32_t stream, float u[4]) {
  uint32_t o[4];
  philox4x32_10(env, (uint32_t)step, stream, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
  for (int i = 0; i < 4; ++i) u[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
}
static inline real lt_lerp(const float r[2], float u) { return (real)r[0] + (real)u * ((real)r[1] - (real)r[0]); }

#endif
