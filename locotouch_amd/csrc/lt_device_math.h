// Device-side small-vector algebra, quad (4-lane) DPP primitives and the Philox RNG for lt_env.hip.
//
// Execution layout (DESIGN.md "Lane mapping"): 4 lanes per environment - one per leg (FR, FL, RR, RL) - so a
// wave64 carries 16 environments.  Per-env reductions over legs/joints/feet are DPP quad_perm butterflies
// (2 VALU ops, no LDS); per-env vectors are loaded one component per lane and broadcast inside the quad.
#pragma once

#include <cstdint>

#ifndef LT_PRIMS_H
#define LT_PRIMS_H "lt_device_prims.h"
#endif
#include LT_PRIMS_H

namespace lt {

// value held by lane I of the caller's quad
template <int I>
__device__ __forceinline__ float qbcast(float x) { return dpp<I * 0x55>(x); }
template <int I>
__device__ __forceinline__ int qbcasti(int x) { return dppi<I * 0x55>(x); }
// sum over the 4 lanes of a quad; bitwise identical in all 4 lanes (a+b == b+a)
__device__ __forceinline__ float qsum(float x) {
  x += dpp<0xB1>(x);  // quad_perm [1,0,3,2]
  x += dpp<0x4E>(x);  // quad_perm [2,3,0,1]
  return x;
}
__device__ __forceinline__ int qor(int x) {
  x |= dppi<0xB1>(x);
  x |= dppi<0x4E>(x);
  return x;
}
// pick one of four replicated values by leg id (used to scatter a replicated per-env vector, 1 component per lane)
__device__ __forceinline__ float sel4(int leg, float a, float b, float c, float d) {
  return leg == 0 ? a : (leg == 1 ? b : (leg == 2 ? c : d));
}

// ---- vectors / matrices --------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3& operator+=(V3& a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
__device__ __forceinline__ V3& operator-=(V3& a, V3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float norm(V3 a) { return fsqrt(dot(a, a)); }
__device__ __forceinline__ V3 qsum(V3 a) { return v3(qsum(a.x), qsum(a.y), qsum(a.z)); }

struct S3 { float xx, xy, xz, yy, yz, zz; };  // symmetric 3x3
struct M3 { float m[9]; };  // row-major
__device__ __forceinline__ M3 m3_zero() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.f; return r; }
__device__ __forceinline__ M3 m3_diag(float d) { M3 r = m3_zero(); r.m[0] = r.m[4] = r.m[8] = d; return r; }
__device__ __forceinline__ V3 mul(const M3& a, V3 v) {
  return v3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
__device__ __forceinline__ V3 tmul(const M3& a, V3 v) {  // a^T v
  return v3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
__device__ __forceinline__ V3 row(const M3& a, int i) { return v3(a.m[3 * i], a.m[3 * i + 1], a.m[3 * i + 2]); }
__device__ __forceinline__ V3 col(const M3& a, int j) { return v3(a.m[j], a.m[3 + j], a.m[6 + j]); }
__device__ __forceinline__ void set_row(M3& a, int i, V3 v) { a.m[3 * i] = v.x; a.m[3 * i + 1] = v.y; a.m[3 * i + 2] = v.z; }
__device__ __forceinline__ void set_col(M3& a, int j, V3 v) { a.m[j] = v.x; a.m[3 + j] = v.y; a.m[6 + j] = v.z; }
__device__ __forceinline__ M3& operator+=(M3& a, const M3& b) { for (int i = 0; i < 9; ++i) a.m[i] += b.m[i]; return a; }
__device__ __forceinline__ M3& operator-=(M3& a, const M3& b) { for (int i = 0; i < 9; ++i) a.m[i] -= b.m[i]; return a; }
__device__ __forceinline__ M3 transpose(const M3& a) {
  M3 r;
  r.m[0] = a.m[0]; r.m[1] = a.m[3]; r.m[2] = a.m[6];
  r.m[3] = a.m[1]; r.m[4] = a.m[4]; r.m[5] = a.m[7];
  r.m[6] = a.m[2]; r.m[7] = a.m[5]; r.m[8] = a.m[8];
  return r;
}
// r~ M  (rows)
__device__ __forceinline__ M3 skew_mul(V3 r, const M3& a) {
  M3 o;
  V3 r0 = row(a, 0), r1 = row(a, 1), r2 = row(a, 2);
  set_row(o, 0, r.y * r2 - r.z * r1);
  set_row(o, 1, r.z * r0 - r.x * r2);
  set_row(o, 2, r.x * r1 - r.y * r0);
  return o;
}
// M r~  (columns)
__device__ __forceinline__ M3 mul_skew(const M3& a, V3 r) {
  M3 o;
  V3 c0 = col(a, 0), c1 = col(a, 1), c2 = col(a, 2);
  set_col(o, 0, r.z * c1 - r.y * c2);
  set_col(o, 1, r.x * c2 - r.z * c0);
  set_col(o, 2, r.y * c0 - r.x * c1);
  return o;
}
__device__ __forceinline__ M3 outer(V3 a, V3 b) {
  M3 o;
  set_row(o, 0, a.x * b);
  set_row(o, 1, a.y * b);
  set_row(o, 2, a.z * b);
  return o;
}
__device__ __forceinline__ M3 quat_to_mat(float w, float x, float y, float z) {  // body -> world
  M3 R;
  R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - w * z);     R.m[2] = 2 * (x * z + w * y);
  R.m[3] = 2 * (x * y + w * z);     R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - w * x);
  R.m[6] = 2 * (x * z - w * y);     R.m[7] = 2 * (y * z + w * x);     R.m[8] = 1 - 2 * (x * x + y * y);
  return R;
}

// ---- packed pairs: two 3-vectors side by side ---------------------------------------------------------------
// gfx950 issues v_pk_{fma,mul,add}_f32 - two f32 operations on a 64-bit register pair - at the rate of one scalar VALU
// operation, and each source picks its halves freely (op_sel / neg_lo / neg_hi: broadcasts, swaps and sign flips cost
// nothing).  The physics is issue-bound on one wave per SIMD, so wherever two 3-vectors go through the same arithmetic
// (angular | linear parts of a spatial vector, velocity | acceleration of a link, two joints' Jacobian columns) they are held
// as a P3 from the start: the pairs live in adjacent registers by construction (the SLP vectoriser's automatic packing lost
// to its own register moves).
typedef float f2 __attribute__((ext_vector_type(2)));
struct P3 { f2 x, y, z; };
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 sp2(float a) { f2 r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ f2 lolo(f2 a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ f2 hihi(f2 a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ f2 swp(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ P3 pair(V3 a, V3 b) { P3 r; r.x = mk2(a.x, b.x); r.y = mk2(a.y, b.y); r.z = mk2(a.z, b.z); return r; }
__device__ __forceinline__ P3 both(V3 a) { return pair(a, a); }
__device__ __forceinline__ V3 lo(const P3& a) { return v3(a.x.x, a.y.x, a.z.x); }
__device__ __forceinline__ V3 hi(const P3& a) { return v3(a.x.y, a.y.y, a.z.y); }
__device__ __forceinline__ P3 p3(f2 x, f2 y, f2 z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ P3 operator+(const P3& a, const P3& b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ P3 operator-(const P3& a, const P3& b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ P3 operator-(const P3& a) { return p3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ P3 operator*(f2 s, const P3& a) { return p3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ P3 operator*(float s, const P3& a) { return sp2(s) * a; }
__device__ __forceinline__ P3& operator+=(P3& a, const P3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
__device__ __forceinline__ P3& operator-=(P3& a, const P3& b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
__device__ __forceinline__ P3 cross(const P3& a, const P3& b) { return p3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ P3 cross(V3 a, const P3& b) {  // (a x b.lo | a x b.hi)
  return p3(sp2(a.y) * b.z - sp2(a.z) * b.y, sp2(a.z) * b.x - sp2(a.x) * b.z, sp2(a.x) * b.y - sp2(a.y) * b.x);
}
__device__ __forceinline__ P3 cross(const P3& a, V3 b) {  // (a.lo x b | a.hi x b)
  return p3(a.y * sp2(b.z) - a.z * sp2(b.y), a.z * sp2(b.x) - a.x * sp2(b.z), a.x * sp2(b.y) - a.y * sp2(b.x));
}
__device__ __forceinline__ f2 dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f2 dot(V3 a, const P3& b) { return sp2(a.x) * b.x + sp2(a.y) * b.y + sp2(a.z) * b.z; }
__device__ __forceinline__ P3 mul(const M3& a, const P3& v) {  // (a v.lo | a v.hi)
  return p3(sp2(a.m[0]) * v.x + sp2(a.m[1]) * v.y + sp2(a.m[2]) * v.z, sp2(a.m[3]) * v.x + sp2(a.m[4]) * v.y + sp2(a.m[5]) * v.z,
            sp2(a.m[6]) * v.x + sp2(a.m[7]) * v.y + sp2(a.m[8]) * v.z);
}
__device__ __forceinline__ P3 tmul(const M3& a, const P3& v) {  // (a^T v.lo | a^T v.hi)
  return p3(sp2(a.m[0]) * v.x + sp2(a.m[3]) * v.y + sp2(a.m[6]) * v.z, sp2(a.m[1]) * v.x + sp2(a.m[4]) * v.y + sp2(a.m[7]) * v.z,
            sp2(a.m[2]) * v.x + sp2(a.m[5]) * v.y + sp2(a.m[8]) * v.z);
}
__device__ __forceinline__ f2 qsum(f2 a) { return mk2(qsum(a.x), qsum(a.y)); }
__device__ __forceinline__ P3 qsum(const P3& a) { return p3(qsum(a.x), qsum(a.y), qsum(a.z)); }
__device__ __forceinline__ P3 mul(const S3& s, const P3& v) {
  return p3(sp2(s.xx) * v.x + sp2(s.xy) * v.y + sp2(s.xz) * v.z, sp2(s.xy) * v.x + sp2(s.yy) * v.y + sp2(s.yz) * v.z,
            sp2(s.xz) * v.x + sp2(s.yz) * v.y + sp2(s.zz) * v.z);
}

// ---- principal-axis joint rotations: R = rot(AX, q) maps child coords -> parent coords ------------
template <int AX>
__device__ __forceinline__ V3 rot_fwd(float c, float s, V3 v) {  // R v
  if (AX == 0) return v3(v.x, c * v.y - s * v.z, s * v.y + c * v.z);
  return v3(c * v.x + s * v.z, v.y, -s * v.x + c * v.z);
}
template <int AX>
__device__ __forceinline__ V3 rot_inv(float c, float s, V3 v) {  // R^T v
  if (AX == 0) return v3(v.x, c * v.y + s * v.z, -s * v.y + c * v.z);
  return v3(c * v.x - s * v.z, v.y, s * v.x + c * v.z);
}
template <int AX>
__device__ __forceinline__ P3 rot_fwd(float c, float s, const P3& v) {  // (R v.lo | R v.hi)
  const f2 cc = sp2(c), ss = sp2(s);
  if (AX == 0) return p3(v.x, cc * v.y - ss * v.z, ss * v.y + cc * v.z);
  return p3(cc * v.x + ss * v.z, v.y, cc * v.z - ss * v.x);
}
template <int AX>
__device__ __forceinline__ P3 rot_inv(float c, float s, const P3& v) {  // (R^T v.lo | R^T v.hi)
  const f2 cc = sp2(c), ss = sp2(s);
  if (AX == 0) return p3(v.x, cc * v.y + ss * v.z, cc * v.z - ss * v.y);
  return p3(cc * v.x - ss * v.z, v.y, ss * v.x + cc * v.z);
}
// R M R^T
template <int AX>
__device__ __forceinline__ M3 rot_sim(float c, float s, const M3& a) {
  M3 t, o;
  for (int i = 0; i < 3; ++i) set_row(t, i, rot_fwd<AX>(c, s, row(a, i)));   // M R^T
  for (int j = 0; j < 3; ++j) set_col(o, j, rot_fwd<AX>(c, s, col(t, j)));   // R (M R^T)
  return o;
}
// M R : world rotation of the child link, Rw_child = Rw_parent * R
template <int AX>
__device__ __forceinline__ M3 mul_rot(const M3& a, float c, float s) {
  M3 o = a;
  if (AX == 0) {
    V3 c1 = col(a, 1), c2 = col(a, 2);
    set_col(o, 1, c * c1 + s * c2);
    set_col(o, 2, c * c2 - s * c1);
  } else {
    V3 c0 = col(a, 0), c2 = col(a, 2);
    set_col(o, 0, c * c0 - s * c2);
    set_col(o, 2, s * c0 + c * c2);
  }
  return o;
}
template <int AX>
__device__ __forceinline__ float comp(V3 v) { return AX == 0 ? v.x : (AX == 1 ? v.y : v.z); }
template <int AX>
__device__ __forceinline__ V3 axis_scaled(float s) { return AX == 0 ? v3(s, 0, 0) : (AX == 1 ? v3(0, s, 0) : v3(0, 0, s)); }

// ---- quaternions (wxyz, isaaclab.utils.math conventions) -----------------------------------------
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  Q4 o;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return o;
}
__device__ __forceinline__ Q4 qconj(Q4 a) { Q4 o; o.w = a.w; o.x = -a.x; o.y = -a.y; o.z = -a.z; return o; }
__device__ __forceinline__ V3 qapply(Q4 q, V3 v) {
  V3 qv = v3(q.x, q.y, q.z);
  V3 t = 2.f * cross(qv, v);
  return v + q.w * t + cross(qv, t);
}
__device__ __forceinline__ V3 qapply_inv(Q4 q, V3 v) {
  V3 qv = v3(q.x, q.y, q.z);
  V3 t = 2.f * cross(qv, v);
  return v - q.w * t + cross(qv, t);
}
__device__ __forceinline__ Q4 q_from_euler(float roll, float pitch, float yaw) {
  // half angles within [-pi, pi]: v_sin_f32 / v_cos_f32 (lt_device_prims.h; |err| ~1e-6, the parity bands are 5e-5).  sincosf
  // is ~95 instructions a call, 13 calls in the step kernel - on the reset path and in the object-state row, i.e. in once-through
  // code that runs at instruction-fetch speed on the tiles that end the launch.
  const float sy = fsin(yaw * 0.5f), cy = fcos(yaw * 0.5f), sr = fsin(roll * 0.5f), cr = fcos(roll * 0.5f), sp = fsin(pitch * 0.5f), cp = fcos(pitch * 0.5f);
  Q4 o;
  o.w = cy * cr * cp + sy * sr * sp;
  o.x = cy * sr * cp - sy * cr * sp;
  o.y = cy * cr * sp + sy * sr * cp;
  o.z = sy * cr * cp - cy * sr * sp;
  return o;
}
// (cos, sin) of a quaternion's yaw (euler_xyz_from_quat(...)[2] = atan2(2(wz + xy), 1 - 2(y^2 + z^2))) WITHOUT the angle: what the
// reference builds from the angle - yaw_quat rotations, yaw differences - needs only this pair (atan2f + fmodf + a sincos per use
// were ~200 instructions; atan2(0, 0) = 0 keeps the degenerate case)
__device__ __forceinline__ void q_yaw_cs(Q4 q, float& c, float& s) {
  const float sn = 2.f * (q.w * q.z + q.x * q.y), cn = 1.f - 2.f * (q.y * q.y + q.z * q.z);
  const float n2 = sn * sn + cn * cn;
  const bool ok = n2 > 1e-24f;
  const float inv = frsqrt(ok ? n2 : 1.f);
  c = ok ? cn * inv : 1.f;
  s = ok ? sn * inv : 0.f;
}
__device__ __forceinline__ Q4 q_normalize(Q4 q) {
  float inv = frsqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q.w *= inv; q.x *= inv; q.y *= inv; q.z *= inv;
  return q;
}
// first-order integration of orientation with a world-frame angular velocity, then normalise
__device__ __forceinline__ Q4 q_integrate(Q4 q, V3 w, float h) {
  Q4 dq; dq.w = 0.f; dq.x = w.x; dq.y = w.y; dq.z = w.z;
  Q4 t = qmul(dq, q);
  q.w += 0.5f * h * t.w; q.x += 0.5f * h * t.x; q.y += 0.5f * h * t.y; q.z += 0.5f * h * t.z;
  return q_normalize(q);
}

// ---- Philox4x32-10, bit-exact twin of oracle/lt_oracle_math.h ----------------------------------------
struct U4 { float a, b, c, d; };
__device__ __forceinline__ U4 rng4(uint64_t seed, uint32_t env, uint64_t step, uint32_t stream) {
  uint32_t c0 = env, c1 = (uint32_t)step, c2 = stream, c3 = (uint32_t)(step >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float s = 1.0f / 16777216.0f;
  U4 u;
  u.a = (float)(c0 >> 8) * s; u.b = (float)(c1 >> 8) * s; u.c = (float)(c2 >> 8) * s; u.d = (float)(c3 >> 8) * s;
  return u;
}
__device__ __forceinline__ float lerp2(const float r[2], float u) { return r[0] + u * (r[1] - r[0]); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

}  // namespace lt
