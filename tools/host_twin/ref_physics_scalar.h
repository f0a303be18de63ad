// FROZEN COPY (round 3) of the scalar-formulation physics (csrc/lt_physics_crba.h before the packed-pair rewrite): the reference the
// host twin compares the product physics against.  Test tool only; not built into the product.
// Forward dynamics of one env (floating base + 4 x 3-link legs + carried cylinder) for lt_env.hip.
//
// Formulation (kernel side; the oracle uses a generic ABA - the two must agree to fp32 tolerance):
//   (M + h Jc^T B Jc) a = tau - C + Jc^T F0'      with a = (base spatial accel, 12 joint accels)
// * C by RNEA with zero accelerations (velocity products, gravity as explicit link wrenches, explicit contact parts),
// * M by CRBA on rigid 10-parameter composites (mass, first moment, rotational inertia about the frame origin),
// * the implicit penalty contacts (DESIGN.md "Physics model") enter through the contact Jacobian columns,
// * per-leg Schur complement: each lane eliminates its 3 joints (3x3 Cholesky), the 4 lanes of the quad butterfly-sum
//   their 6x6 base shares, every lane solves the same 6x6 and back-substitutes its own joints.
// Lane = leg.  Everything lives in registers; no LDS, no barriers.
#pragma once

// ---- robot model (generated from the reference URDF by tools/compile_robot_model.py), mirror form: every per-leg
// constant is (FL-leg literal) x sign[pattern], sign = {1, sx, sy, sx*sy} of the lane's leg -> no table loads, nothing
// held in registers across the physics loop.
__device__ constexpr float k_m_mass[3] = LT_MIRROR_LINK_MASS_INIT;
__device__ constexpr float k_m_mc[3][3] = LT_MIRROR_LINK_MC_INIT;
__device__ constexpr int k_m_mc_pat[3][3] = LT_MIRROR_LINK_MC_PAT;
__device__ constexpr float k_m_io[3][6] = LT_MIRROR_LINK_IO_INIT;
__device__ constexpr int k_m_io_pat[3][6] = LT_MIRROR_LINK_IO_PAT;
__device__ constexpr float k_m_off[3][3] = LT_MIRROR_JOINT_OFFSET_INIT;
__device__ constexpr int k_m_off_pat[3][3] = LT_MIRROR_JOINT_OFFSET_PAT;
__device__ constexpr float k_m_dq[3] = LT_MIRROR_JOINT_DEFAULT_INIT;
__device__ constexpr int k_m_dq_pat[3] = LT_MIRROR_JOINT_DEFAULT_PAT;
__device__ constexpr float k_joint_lo[3] = LT_JOINT_LOWER_INIT;
__device__ constexpr float k_joint_hi[3] = LT_JOINT_UPPER_INIT;
__device__ constexpr float k_trunk_com[3] = LT_TRUNK_COM_INIT;
__device__ constexpr float k_trunk_icom[6] = LT_TRUNK_ICOM_INIT;
__device__ constexpr float k_trunk_half[3] = LT_TRUNK_BOX_HALF_INIT;

// ---- spatial algebra types ---------------------------------------------------------------------------
struct I6 { M3 A, B, C; };  // [A B; B^T C], angular first
struct S6 { V3 a, l; };
struct LinkC { float m; V3 mc; S3 Io; V3 r; };  // mass, m*com, rotational inertia about the link origin, joint offset
// constants of link K of the lane's leg: literal x sign (zero components stay literal zeros and fold away)
template <int K>
__device__ __forceinline__ LinkC make_link(const float (&sgn)[4]) {
  LinkC L;
  L.m = k_m_mass[K];
  L.mc = v3(k_m_mc[K][0] * sgn[k_m_mc_pat[K][0]], k_m_mc[K][1] * sgn[k_m_mc_pat[K][1]], k_m_mc[K][2] * sgn[k_m_mc_pat[K][2]]);
  L.Io.xx = k_m_io[K][0] * sgn[k_m_io_pat[K][0]]; L.Io.xy = k_m_io[K][1] * sgn[k_m_io_pat[K][1]]; L.Io.xz = k_m_io[K][2] * sgn[k_m_io_pat[K][2]];
  L.Io.yy = k_m_io[K][3] * sgn[k_m_io_pat[K][3]]; L.Io.yz = k_m_io[K][4] * sgn[k_m_io_pat[K][4]]; L.Io.zz = k_m_io[K][5] * sgn[k_m_io_pat[K][5]];
  L.r = v3(k_m_off[K][0] * sgn[k_m_off_pat[K][0]], k_m_off[K][1] * sgn[k_m_off_pat[K][1]], k_m_off[K][2] * sgn[k_m_off_pat[K][2]]);
  return L;
}

__device__ __forceinline__ M3 skew_of(V3 v) {
  M3 o = m3_zero();
  o.m[1] = -v.z; o.m[2] = v.y; o.m[3] = v.z; o.m[5] = -v.x; o.m[6] = -v.y; o.m[7] = v.x;
  return o;
}
__device__ __forceinline__ M3 inertia_about_origin(float m, V3 c, const float ic[6], float scale) {
  // Ic*scale - m c~ c~ = Ic*scale + m (|c|^2 1 - c c^T)
  M3 o;
  const float cc = dot(c, c);
  o.m[0] = ic[0] * scale + m * (cc - c.x * c.x); o.m[1] = ic[1] * scale - m * c.x * c.y; o.m[2] = ic[2] * scale - m * c.x * c.z;
  o.m[3] = o.m[1]; o.m[4] = ic[3] * scale + m * (cc - c.y * c.y); o.m[5] = ic[4] * scale - m * c.y * c.z;
  o.m[6] = o.m[2]; o.m[7] = o.m[5]; o.m[8] = ic[5] * scale + m * (cc - c.z * c.z);
  return o;
}

// ---- per-env (replicated in the quad) and per-leg (one lane) register state ---------------------------
struct Base {
  V3 p; Q4 q; V3 u, w;  // root pose, world linear / angular velocity
};
struct Obj {
  V3 p; Q4 q; V3 u, w;
  float cur_air, cur_con, last_air, last_con;
  float rad, len, mass, mu;
};
struct Leg {
  float q[3], qd[3], qdd[3], tau[3], raw[3], prev[3], prev2[3];
  float fh[3][4];  // |F| history [slot][hip,thigh,calf,foot]
  float cur_air, cur_con, last_air, last_con;
  float mu;
  V3 foot_p, foot_v;
  float g_last_air, g_last_con, g_valid;
  int g_flags;
};
struct Misc {
  float trunk_mass_add, trunk_mu, trunk_rest, obj_rest;
  float trunk_fh[3];
  V3 cmd; float cmd_time_left; V3 cmd_buf; float cmd_standing;
  float push_robot_left, push_obj_left;
  V3 gait_cmd; float gait_step;
  long long ep_len;
};
struct Report {  // contact forces of the last physics substep (world frame)
  V3 body[4];    // hip, thigh, calf, foot of this lane's leg
  V3 trunk_part; // this lane's share of the trunk force (corners + plate reactions)
  V3 obj_part;   // this lane's share of the object force
  V3 plate;      // tactile tasks: this lane's plate sample - contact point (x, y) in the trunk frame, normal force on the plate
};

// ---- contact law (DESIGN.md "contact model"; executable spec: oracle/lt_oracle.c contact_eval) ----------
struct Law { bool active; float fx, fy, fn, cte, Bn; };
__device__ __forceinline__ Law contact_law(float d, V3 vrel, float kn, float cn, float ct, float mu, float ramp_depth, float h) {
  Law c;
  c.active = false; c.fx = c.fy = c.fn = c.cte = c.Bn = 0.f;
  if (!(d > 0.f)) return c;
  float ramp = d / ramp_depth;
  ramp = ramp > 1.f ? 1.f : ramp;
  const float Bn = kn * h + cn * ramp;
  const float f0n = kn * d - Bn * vrel.z;
  if (!(f0n > 0.f)) return c;
  const float vt = fsqrt(vrel.x * vrel.x + vrel.y * vrel.y);
  float cte = mu * f0n / (vt > 1e-6f ? vt : 1e-6f);
  cte = cte > ct ? ct : cte;
  c.active = true;
  c.fx = -cte * vrel.x; c.fy = -cte * vrel.y; c.fn = f0n; c.cte = cte; c.Bn = Bn;
  return c;
}
// add h J^T B J (J = [-r~ 1]) with B = cte 1 + (Bn - cte) n n^T to a 6x6, n in the coords of the 6x6
__device__ __forceinline__ void add_contact_inertia(I6& IA, V3 r, V3 n, float cte, float Bn, float h) {
  M3 hB = m3_diag(h * cte);
  hB += outer(h * (Bn - cte) * n, n);
  const M3 T = skew_mul(r, hB);
  IA.B += T;
  IA.A -= mul_skew(T, r);
  IA.C += hB;
}

// ---- kinematics of one joint: parent (w,v,Rw,pw) -> child, plus the velocity-product term c = v x S qd -----
template <int AX>
__device__ __forceinline__ void joint_fk(V3 wp, V3 vp, const M3& Rwp, V3 pwp, V3 r, float c, float s, float qd,
                                         V3& om, V3& vl, M3& Rw, V3& pw, V3& ca, V3& cl) {
  const V3 t = vp + cross(wp, r);
  const V3 vj = axis_scaled<AX>(qd);
  om = rot_inv<AX>(c, s, wp) + vj;
  vl = rot_inv<AX>(c, s, t);
  ca = cross(om, vj);
  cl = cross(vl, vj);
  Rw = mul_rot<AX>(Rwp, c, s);
  pw = pwp + mul(Rwp, r);
}
struct RC { Law law; V3 f0w; };
// ground contact of a sphere (centre r in the link frame, radius rho); accumulates into (IA, pA)
__device__ __forceinline__ RC ground_contact(const lt_cfg& c, float h, V3 r, float rho, float mu, const M3& Rw, V3 pw, V3 om, V3 vl,
                                             I6& IA, S6& pA) {
  RC out;
  out.law.active = false; out.law.fx = out.law.fy = out.law.fn = out.law.cte = out.law.Bn = 0.f;
  out.f0w = v3(0, 0, 0);
  const V3 zb = row(Rw, 2);
  const V3 rc = r - rho * zb;
  if (!(pw.z + dot(zb, rc) < 0.f)) return out;  // above the ground: nothing else to compute
  const V3 Pc = pw + mul(Rw, rc);
  const V3 vw = mul(Rw, vl + cross(om, rc));
  out.law = contact_law(-Pc.z, vw, c.ground_kn, c.ground_cn, c.ground_ct, mu, c.contact_ramp, h);
  out.f0w = v3(out.law.fx, out.law.fy, out.law.fn);
  if (out.law.active) {
    add_contact_inertia(IA, rc, zb, out.law.cte, out.law.Bn, h);
    const V3 f0b = tmul(Rw, out.f0w);
    pA.a -= cross(rc, f0b);
    pA.l -= f0b;
  }
  return out;
}
// final (post-solve) force of a ground contact on a link with spatial acceleration (aa, al)
__device__ __forceinline__ V3 ground_force(const RC& rc_, float h, V3 r, float rho, const M3& Rw, V3 aa, V3 al) {
  if (!rc_.law.active) return v3(0.f, 0.f, 0.f);
  const V3 rc = r - rho * row(Rw, 2);
  const V3 aw = mul(Rw, al + cross(aa, rc));
  return v3(rc_.f0w.x - h * rc_.law.cte * aw.x, rc_.f0w.y - h * rc_.law.cte * aw.y, rc_.f0w.z - h * rc_.law.Bn * aw.z);
}

// 6x6 SPD solve (Cholesky), fully unrolled into registers.  M = [A B; B^T C] (upper blocks), rhs (a, l)
__device__ __forceinline__ void spd6_solve(const I6& M, V3 ba, V3 bl, V3& xa, V3& xl) {
  float A[6][6];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[i][j] = M.A.m[i * 3 + j];
      A[i][3 + j] = M.B.m[i * 3 + j];
      A[3 + i][j] = M.B.m[j * 3 + i];
      A[3 + i][3 + j] = M.C.m[i * 3 + j];
    }
  float Lm[6][6], inv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float sacc = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) sacc -= Lm[i][k] * Lm[j][k];
      if (i == j) inv[i] = frsqrt(sacc);  // (the diagonal itself is never used)
      else Lm[i][j] = sacc * inv[j];
    }
  }
  float b[6] = {ba.x, ba.y, ba.z, bl.x, bl.y, bl.z}, y[6], x[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float sacc = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) sacc -= Lm[i][k] * y[k];
    y[i] = sacc * inv[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float sacc = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) sacc -= Lm[k][i] * x[k];
    x[i] = sacc * inv[i];
  }
  xa = v3(x[0], x[1], x[2]);
  xl = v3(x[3], x[4], x[5]);
}
__device__ __forceinline__ M3 qsum_sym(const M3& a) {  // quad-sum of a symmetric 3x3 (6 butterflies)
  M3 o;
  o.m[0] = qsum(a.m[0]); o.m[1] = qsum(a.m[1]); o.m[2] = qsum(a.m[2]); o.m[4] = qsum(a.m[4]); o.m[5] = qsum(a.m[5]); o.m[8] = qsum(a.m[8]);
  o.m[3] = o.m[1]; o.m[6] = o.m[2]; o.m[7] = o.m[5];
  return o;
}
__device__ __forceinline__ M3 qsum_full(const M3& a) {
  M3 o;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.m[i] = qsum(a.m[i]);
  return o;
}

// sphere table of this lane: 0 foot (calf), 1 calf mid (calf), 2 knee (thigh), 3 hip (hip), 4/5 trunk corners
__device__ __forceinline__ V3 trunk_corner(int leg, bool hi) {
  const float sx = leg < 2 ? 1.f : -1.f, sy = (leg & 1) ? 1.f : -1.f;
  return hi ? v3(sx * k_trunk_half[0], sy * LT_BACK_HALF_Y, LT_BACK_TOP_Z) : v3(sx * k_trunk_half[0], sy * k_trunk_half[1], -k_trunk_half[2]);
}


__device__ __forceinline__ V3 mul(const S3& s, V3 v) {
  return v3(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
__device__ __forceinline__ S3 s3_from(const M3& a) { S3 s; s.xx = a.m[0]; s.xy = a.m[1]; s.xz = a.m[2]; s.yy = a.m[4]; s.yz = a.m[5]; s.zz = a.m[8]; return s; }
template <int AX>
__device__ __forceinline__ V3 s3_col(const S3& s) { return AX == 0 ? v3(s.xx, s.xy, s.xz) : (AX == 1 ? v3(s.xy, s.yy, s.yz) : v3(s.xz, s.yz, s.zz)); }
// R S R^T for R = rot(AX, q)
template <int AX>
__device__ __forceinline__ S3 rot_sym(float c, float s, const S3& a) {
  S3 o;
  const float cc = c * c, ss = s * s, cs = c * s;
  if (AX == 0) {
    o.xx = a.xx; o.xy = c * a.xy - s * a.xz; o.xz = s * a.xy + c * a.xz;
    o.yy = cc * a.yy - 2.f * cs * a.yz + ss * a.zz;
    o.zz = ss * a.yy + 2.f * cs * a.yz + cc * a.zz;
    o.yz = cs * (a.yy - a.zz) + (cc - ss) * a.yz;
  } else {
    o.yy = a.yy; o.xy = c * a.xy + s * a.yz; o.yz = -s * a.xy + c * a.yz;
    o.xx = cc * a.xx + 2.f * cs * a.xz + ss * a.zz;
    o.zz = ss * a.xx - 2.f * cs * a.xz + cc * a.zz;
    o.xz = cs * (a.zz - a.xx) + (cc - ss) * a.xz;
  }
  return o;
}

struct Rigid { float m; V3 mc; S3 Io; };  // rigid (or composite) inertia about the frame origin
struct F6 { V3 n, f; };                   // spatial force: moment about the frame origin, force

// composite inertia of a child subtree re-expressed in the parent frame (child origin at r, child->parent rotation R)
template <int AX>
__device__ __forceinline__ Rigid rigid_to_parent(const Rigid& ch, float c, float s, V3 r) {
  Rigid o;
  o.m = ch.m;
  const V3 cp = rot_fwd<AX>(c, s, ch.mc);  // first moment in parent axes, still about the child origin
  o.mc = cp + ch.m * r;
  S3 I = rot_sym<AX>(c, s, ch.Io);
  const float rr = dot(r, r), rc2 = 2.f * dot(r, cp);
  // + m(|r|^2 1 - r r^T) + 2 (r.c') 1 - r c'^T - c' r^T
  I.xx += ch.m * (rr - r.x * r.x) + rc2 - 2.f * r.x * cp.x;
  I.yy += ch.m * (rr - r.y * r.y) + rc2 - 2.f * r.y * cp.y;
  I.zz += ch.m * (rr - r.z * r.z) + rc2 - 2.f * r.z * cp.z;
  I.xy += -ch.m * r.x * r.y - r.x * cp.y - cp.x * r.y;
  I.xz += -ch.m * r.x * r.z - r.x * cp.z - cp.x * r.z;
  I.yz += -ch.m * r.y * r.z - r.y * cp.z - cp.y * r.z;
  o.Io = I;
  return o;
}
__device__ __forceinline__ Rigid rigid_add(const Rigid& a, const Rigid& b) {
  Rigid o;
  o.m = a.m + b.m; o.mc = a.mc + b.mc;
  o.Io.xx = a.Io.xx + b.Io.xx; o.Io.xy = a.Io.xy + b.Io.xy; o.Io.xz = a.Io.xz + b.Io.xz;
  o.Io.yy = a.Io.yy + b.Io.yy; o.Io.yz = a.Io.yz + b.Io.yz; o.Io.zz = a.Io.zz + b.Io.zz;
  return o;
}
template <int AX>
__device__ __forceinline__ F6 force_to_parent(const F6& x, float c, float s, V3 r) {
  F6 o;
  o.f = rot_fwd<AX>(c, s, x.f);
  o.n = rot_fwd<AX>(c, s, x.n) + cross(r, o.f);
  return o;
}
// column AX of a rigid 6x6: I [e;0] = (Io e, e x mc)
template <int AX>
__device__ __forceinline__ F6 rigid_col(const Rigid& R) {
  F6 o;
  o.n = s3_col<AX>(R.Io);
  o.f = cross(axis_scaled<AX>(1.f), R.mc);
  return o;
}
// rigid-body wrench for spatial velocity (om, vl) and acceleration (aa, al): I a + v x* I v - gravity
__device__ __forceinline__ F6 rigid_bias(const Rigid& R, V3 om, V3 vl, V3 aa, V3 al, V3 gb) {
  const V3 n = mul(R.Io, om) + cross(R.mc, vl);
  const V3 f = cross(om, R.mc) + R.m * vl;
  F6 o;
  o.n = mul(R.Io, aa) + cross(R.mc, al) + cross(om, n) + cross(vl, f) - cross(R.mc, gb);
  o.f = cross(aa, R.mc) + R.m * al + cross(om, f) - R.m * gb;
  return o;
}

struct LegSys {        // this lane's 3 joints: H (sym 3x3), coupling to the base (3 x 6), right-hand side
  float h00, h01, h02, h11, h12, h22;
  V3 bn[3], bl[3];     // row j of H_lb: (moment part, force part), base body coords
  float rhs[3];
};

// ground contact of a sphere on link K (0 hip, 1 thigh, 2 calf) of this leg.
//   (aa, al) = velocity-product spatial acceleration of the link (zero joint / base accelerations)
template <int K>
__device__ __forceinline__ RC leg_contact(const lt_cfg& c, float h, V3 r, float rho, float mu, const M3& Rw, V3 pwk, V3 om, V3 vl,
                                          V3 aa, V3 al, const M3& R0, V3 p0, const V3 (&axw)[3], const V3 (&pw)[3],
                                          F6& fk, LegSys& S, I6& Mbb) {
  RC out;
  out.law.active = false; out.law.fx = out.law.fy = out.law.fn = out.law.cte = out.law.Bn = 0.f;
  out.f0w = v3(0, 0, 0);
  const V3 zb = row(Rw, 2);
  const V3 rc = r - rho * zb;
  const float pz = pwk.z + dot(zb, rc);  // world height of the contact point
  if (!(pz < 0.f)) return out;          // above the ground: nothing else to compute
  const V3 Pc = pwk + mul(Rw, rc);
  const V3 vw = mul(Rw, vl + cross(om, rc));
  out.law = contact_law(-Pc.z, vw, c.ground_kn, c.ground_cn, c.ground_ct, mu, c.contact_ramp, h);
  out.f0w = v3(out.law.fx, out.law.fy, out.law.fn);
  if (!out.law.active) return out;
  const float hct = h * out.law.cte, hbn = h * out.law.Bn;
  // explicit part with the velocity-product acceleration of the point folded in
  const V3 avp = mul(Rw, al + cross(aa, rc));
  const V3 F0 = v3(out.f0w.x - hct * avp.x, out.f0w.y - hct * avp.y, out.f0w.z - hbn * avp.z);
  const V3 f0b = tmul(Rw, F0);
  fk.n -= cross(rc, f0b);
  fk.f -= f0b;
  // implicit part through the Jacobian columns (world frame; the ground contact frame is world-aligned => B diagonal)
  const V3 rho_b = tmul(R0, Pc - p0);
  V3 g[3];
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    const V3 w = cross(axw[j], Pc - pw[j]);
    g[j] = v3(hct * w.x, hct * w.y, hbn * w.z);
    const V3 gb = tmul(R0, g[j]);
    S.bn[j] += cross(rho_b, gb);
    S.bl[j] += gb;
  }
  {
    const V3 w0 = cross(axw[0], Pc - pw[0]);
    S.h00 += dot(w0, g[0]);
    if (K >= 1) {
      const V3 w1 = cross(axw[1], Pc - pw[1]);
      S.h01 += dot(w0, g[1]);
      S.h11 += dot(w1, g[1]);
      if (K >= 2) {
        const V3 w2 = cross(axw[2], Pc - pw[2]);
        S.h02 += dot(w0, g[2]);
        S.h12 += dot(w1, g[2]);
        S.h22 += dot(w2, g[2]);
      }
    }
  }
  add_contact_inertia(Mbb, rho_b, row(R0, 2), out.law.cte, out.law.Bn, h);
  return out;
}

// ---- CRBA on rigid composites: this leg's joint-space inertia, its coupling to the base and its rigid share of the base
//      block.  A function of the joint angles alone, so a helper wave can run it beside the leg dynamics. ----
struct CrbaOut {
  float h00, h01, h02, h11, h12, h22;
  V3 bn[3], bl[3];
  S3 Io; V3 mc; float m;  // composite of the whole leg about the base origin
};
__device__ __forceinline__ CrbaOut crba_part(const float (&sgn)[4], const float (&cq)[3], const float (&sq)[3]) {
  CrbaOut o;
  const LinkC LC[3] = {make_link<0>(sgn), make_link<1>(sgn), make_link<2>(sgn)};
  Rigid RB[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { RB[k].m = LC[k].m; RB[k].mc = LC[k].mc; RB[k].Io = LC[k].Io; }
  {
    const Rigid C3 = RB[2];
    const Rigid C2 = rigid_add(RB[1], rigid_to_parent<1>(C3, cq[2], sq[2], LC[2].r));
    const Rigid C1 = rigid_add(RB[0], rigid_to_parent<1>(C2, cq[1], sq[1], LC[1].r));
    const Rigid Cb = rigid_to_parent<0>(C1, cq[0], sq[0], LC[0].r);
    // joint 2 (calf, axis y)
    const F6 F3 = rigid_col<1>(C3);
    o.h22 = C3.Io.yy;
    const F6 F32 = force_to_parent<1>(F3, cq[2], sq[2], LC[2].r);
    o.h12 = F32.n.y;
    const F6 F31 = force_to_parent<1>(F32, cq[1], sq[1], LC[1].r);
    o.h02 = F31.n.x;
    const F6 F30 = force_to_parent<0>(F31, cq[0], sq[0], LC[0].r);
    o.bn[2] = F30.n; o.bl[2] = F30.f;
    // joint 1 (thigh, axis y)
    const F6 F2 = rigid_col<1>(C2);
    o.h11 = C2.Io.yy;
    const F6 F21 = force_to_parent<1>(F2, cq[1], sq[1], LC[1].r);
    o.h01 = F21.n.x;
    const F6 F20 = force_to_parent<0>(F21, cq[0], sq[0], LC[0].r);
    o.bn[1] = F20.n; o.bl[1] = F20.f;
    // joint 0 (hip, axis x)
    const F6 F1 = rigid_col<0>(C1);
    o.h00 = C1.Io.xx;
    const F6 F10 = force_to_parent<0>(F1, cq[0], sq[0], LC[0].r);
    o.bn[0] = F10.n; o.bl[0] = F10.f;
    o.Io = Cb.Io; o.mc = Cb.mc; o.m = Cb.m;  // rigid share of the base block
  }
  return o;
}

// ---- the carried cylinder's share of a substep: its contacts with the plate (sample = lane) and the ground, its own 6x6
//      solve, and its integration.  Couples to the robot only through the base STATE at the substep start (the plate
//      reaction is an explicit wrench on the trunk), so a helper wave can run it beside the leg dynamics. ----
struct ObjOut {
  V3 pb_n, pb_f;      // this lane's share of the base bias wrench (reaction of its plate sample)
  V3 obj_part;        // ... of the net contact force on the object (world)
  V3 trunk_part;      // ... of the net contact force on the trunk (world), plate part
  V3 plate;           // tactile tasks: this lane's plate sample (x, y in the trunk frame, normal force)
};
template <bool TAC>
__device__ __forceinline__ ObjOut object_part(const lt_cfg& c, float h, int leg, const Base& B, Obj& O, float trunk_mu) {
  ObjOut out;
  out.pb_n = v3(0, 0, 0); out.pb_f = v3(0, 0, 0); out.obj_part = v3(0, 0, 0); out.trunk_part = v3(0, 0, 0); out.plate = v3(0, 0, 0);
  const float g = c.gravity;
  const M3 R0 = quat_to_mat(B.q.w, B.q.x, B.q.y, B.q.z);
  // ---- carried cylinder: free body, implicit contacts with the plate (sample = lane) and the ground ----
  V3 obj_aa = v3(0, 0, 0), obj_al = v3(0, 0, 0);
  {
    const M3 Ro = quat_to_mat(O.q.w, O.q.x, O.q.y, O.q.z);
    const V3 ay = col(Ro, 1);
    const float rad = O.rad, half = 0.5f * O.len;
    const float mu_plate = 0.5f * (trunk_mu + O.mu);
    const V3 ct = tmul(R0, O.p - B.p), at = tmul(R0, ay);
    const float hx = LT_BACK_HALF_X, hy = LT_RAIL_Y + LT_RAIL_RADIUS, zp = LT_BACK_TOP_Z;
    float s0 = -half, s1 = half;
    bool ok = true;
    {
      const float cc[2] = {ct.x, ct.y}, dd[2] = {at.x, at.y}, lim[2] = {hx, hy};
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        if (ok) {
          if (fabsf(dd[ax]) < 1e-9f) { if (fabsf(cc[ax]) > lim[ax]) ok = false; }
          else {
            float ta = (-lim[ax] - cc[ax]) / dd[ax], tb = (lim[ax] - cc[ax]) / dd[ax];
            if (ta > tb) { const float t = ta; ta = tb; tb = t; }
            s0 = ta > s0 ? ta : s0;
            s1 = tb < s1 ? tb : s1;
            if (s0 > s1) ok = false;
          }
        }
      }
    }
    I6 Mo; Mo.A = m3_zero(); Mo.B = m3_zero(); Mo.C = m3_zero();
    V3 rhs_a = v3(0, 0, 0), rhs_l = v3(0, 0, 0);
    Law lp; lp.active = false; lp.fx = lp.fy = lp.fn = lp.cte = lp.Bn = 0.f;
    V3 Pw_p = v3(0, 0, 0), rho_p = v3(0, 0, 0), F0_p = v3(0, 0, 0);
    const V3 nw = col(R0, 2);
    out.plate = v3(0, 0, 0);
    if (ok) {
      const float nza = at.z;
      const V3 up = v3(-nza * at.x, -nza * at.y, 1.f - nza * at.z);
      const float un = norm(up);
      const float inv = 1.f / (un > 1e-6f ? un : 1e-6f);
      const float sk = s0 + (s1 - s0) * (float)leg / 3.f;
      const V3 Pt = v3(ct.x + sk * at.x - rad * up.x * inv, ct.y + sk * at.y - rad * up.y * inv, ct.z + sk * at.z - rad * up.z * inv);
      const float d = zp - Pt.z;
      if (TAC) { out.plate.x = Pt.x; out.plate.y = Pt.y; }
      if (d > 0.f) {
        Pw_p = B.p + mul(R0, Pt);
        rho_p = Pw_p - O.p;
        const V3 vo = O.u + cross(O.w, rho_p);
        const V3 vt = B.u + cross(B.w, Pw_p - B.p);
        const V3 vrel = tmul(R0, vo - vt);
        lp = contact_law(d, vrel, c.plate_kn / 4, c.plate_cn / 4, c.plate_ct / 4, mu_plate, c.contact_ramp, h);
        if (lp.active) {
          F0_p = mul(R0, v3(lp.fx, lp.fy, lp.fn));
          add_contact_inertia(Mo, rho_p, nw, lp.cte, lp.Bn, h);
          rhs_a += cross(rho_p, F0_p);
          rhs_l += F0_p;
        }
      }
    }
    Law lg; lg.active = false; lg.fx = lg.fy = lg.fn = lg.cte = lg.Bn = 0.f;
    V3 rho_g = v3(0, 0, 0), F0_g = v3(0, 0, 0);
    if (leg < 2) {
      const float nza = ay.z;
      const V3 up = v3(-nza * ay.x, -nza * ay.y, 1.f - nza * ay.z);
      const float un = norm(up);
      const float inv = 1.f / (un > 1e-6f ? un : 1e-6f);
      const float sk = leg == 0 ? -half : half;
      const V3 Pw = v3(O.p.x + sk * ay.x - rad * up.x * inv, O.p.y + sk * ay.y - rad * up.y * inv, O.p.z + sk * ay.z - rad * up.z * inv);
      if (Pw.z < 0.f) {
        rho_g = Pw - O.p;
        const V3 vo = O.u + cross(O.w, rho_g);
        lg = contact_law(-Pw.z, vo, c.ground_kn, c.ground_cn, c.ground_ct, O.mu * c.ground_mu, c.contact_ramp, h);
        if (lg.active) {
          F0_g = v3(lg.fx, lg.fy, lg.fn);
          add_contact_inertia(Mo, rho_g, v3(0, 0, 1), lg.cte, lg.Bn, h);
          rhs_a += cross(rho_g, F0_g);
          rhs_l += F0_g;
        }
      }
    }
    I6 M;
    M.A = qsum_sym(Mo.A); M.B = qsum_full(Mo.B); M.C = qsum_sym(Mo.C);
    rhs_a = qsum(rhs_a); rhs_l = qsum(rhs_l);
    const float m = O.mass;
    const float Iyy = 0.5f * m * rad * rad, Ixx = m * (3.f * rad * rad + O.len * O.len) / 12.f;
    M3 Iw = m3_diag(Ixx);
    Iw += outer((Iyy - Ixx) * ay, ay);
    M.A += Iw;
    M.C += m3_diag(m);
    rhs_a -= cross(O.w, mul(Iw, O.w));
    rhs_l.z -= m * g;
    spd6_solve(M, rhs_a, rhs_l, obj_aa, obj_al);
    if (lp.active) {
      const V3 ap = obj_al + cross(obj_aa, rho_p);
      const float an = dot(nw, ap);
      const V3 F = F0_p - h * (lp.cte * ap + ((lp.Bn - lp.cte) * an) * nw);
      out.obj_part += F;
      if (TAC) out.plate.z = dot(nw, F);  // the cylinder presses the taxels with the plate-normal part of its contact force
      const V3 Fn = -F;
      const V3 rb = tmul(R0, Pw_p - B.p), fb = tmul(R0, Fn);
      out.pb_n -= cross(rb, fb);
      out.pb_f -= fb;
      out.trunk_part += Fn;
    }
    if (lg.active) {
      const V3 ap = obj_al + cross(obj_aa, rho_g);
      out.obj_part += v3(F0_g.x - h * lg.cte * ap.x, F0_g.y - h * lg.cte * ap.y, F0_g.z - h * lg.Bn * ap.z);
    }
  }

  O.w += h * obj_aa;
  O.u += h * obj_al;
  O.p += h * O.u;
  O.q = q_integrate(O.q, O.w, h);
  return out;
}

// =====================================================================================================
// K2 physics: one integrator substep of length h (torques held).  Reference: PhysX (closed source) - this is the
// engine's own model; executable spec: oracle/lt_oracle.c physics_substep; description: DESIGN.md "Physics model".
// =====================================================================================================
// `Ext` decides where the CRBA and object parts run.  InlineParts: here, in this wave.  The step kernel's helper form passes
// a policy whose publish() hands (cos q, sin q, base state) to two helper waves and whose fetch() waits for their results
// (lt_env.hip): the object is then owned by its helper wave and `O` is not touched here.
struct PhysExt { CrbaOut crba; ObjOut obj; };
struct InlineParts {
  static constexpr bool external = false;
  __device__ __forceinline__ void publish(const float (&)[3], const float (&)[3], const Base&) const {}
  __device__ __forceinline__ void fetch(PhysExt&) const {}
};
template <bool HAS_OBJ, bool TAC = false, class Ext = InlineParts>
__device__ __forceinline__ void physics_substep(const lt_cfg& c, float h, int leg, const float (&sgn)[4], Base& B, Leg& G, Obj& O,
                                                const Misc& X, Report& rep, const Ext& ext = Ext()) {
  const float g = c.gravity;
  const LinkC LC[3] = {make_link<0>(sgn), make_link<1>(sgn), make_link<2>(sgn)};
  float cq[3], sq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { sq[k] = fsin(G.q[k]); cq[k] = fcos(G.q[k]); }
  ext.publish(cq, sq, B);
  PhysExt pe;
  if (!Ext::external && HAS_OBJ) pe.obj = object_part<TAC>(c, h, leg, B, O, X.trunk_mu);  // first: nothing else is live yet
  const M3 R0 = quat_to_mat(B.q.w, B.q.x, B.q.y, B.q.z);
  const V3 wb = tmul(R0, B.w), vb = tmul(R0, B.u);
  V3 om[3], vl[3], pw[3], ca[3], cl[3];
  M3 Rw[3];
  joint_fk<0>(wb, vb, R0, B.p, LC[0].r, cq[0], sq[0], G.qd[0], om[0], vl[0], Rw[0], pw[0], ca[0], cl[0]);
  joint_fk<1>(om[0], vl[0], Rw[0], pw[0], LC[1].r, cq[1], sq[1], G.qd[1], om[1], vl[1], Rw[1], pw[1], ca[1], cl[1]);
  joint_fk<1>(om[1], vl[1], Rw[1], pw[1], LC[2].r, cq[2], sq[2], G.qd[2], om[2], vl[2], Rw[2], pw[2], ca[2], cl[2]);
  // velocity-product spatial accelerations (base and joint accelerations zero)
  V3 aa[3], al[3];
  aa[0] = ca[0]; al[0] = cl[0];
  aa[1] = rot_inv<1>(cq[1], sq[1], aa[0]) + ca[1];
  al[1] = rot_inv<1>(cq[1], sq[1], al[0] + cross(aa[0], LC[1].r)) + cl[1];
  aa[2] = rot_inv<1>(cq[2], sq[2], aa[1]) + ca[2];
  al[2] = rot_inv<1>(cq[2], sq[2], al[1] + cross(aa[1], LC[2].r)) + cl[2];
  const V3 axw[3] = {col(Rw[0], 0), col(Rw[1], 1), col(Rw[2], 1)};  // joint axes in world

  // this lane's share of the base block / base bias wrench
  I6 Mbb; Mbb.A = m3_zero(); Mbb.B = m3_zero(); Mbb.C = m3_zero();
  F6 pb; pb.n = v3(0, 0, 0); pb.f = v3(0, 0, 0);
  rep.trunk_part = v3(0, 0, 0);
  rep.obj_part = v3(0, 0, 0);

  // ---- RNEA forces of this leg's links (zero accelerations) + contacts ----
  Rigid RB[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { RB[k].m = LC[k].m; RB[k].mc = LC[k].mc; RB[k].Io = LC[k].Io; }
  F6 f[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) f[k] = rigid_bias(RB[k], om[k], vl[k], aa[k], al[k], (-g) * row(Rw[k], 2));
  LegSys S;
  S.h00 = S.h01 = S.h02 = S.h11 = S.h12 = S.h22 = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) { S.bn[j] = v3(0, 0, 0); S.bl[j] = v3(0, 0, 0); }
  const float mu_foot = G.mu * c.ground_mu, mu_body = c.ground_mu;
  const V3 r_foot = v3(0.f, 0.f, -0.213f), r_calf = v3(0.f, 0.f, -0.1065f), r_knee = v3(0.f, 0.f, -0.213f);
  const V3 r_hip = v3(0.f, LT_MIRROR_HIP_CYL_Y * sgn[LT_MIRROR_HIP_CYL_Y_PAT], 0.f);
  const RC c_foot = leg_contact<2>(c, h, r_foot, LT_FOOT_RADIUS, mu_foot, Rw[2], pw[2], om[2], vl[2], aa[2], al[2], R0, B.p, axw, pw, f[2], S, Mbb);
  const RC c_calf = leg_contact<2>(c, h, r_calf, 0.012f, mu_body, Rw[2], pw[2], om[2], vl[2], aa[2], al[2], R0, B.p, axw, pw, f[2], S, Mbb);
  const RC c_knee = leg_contact<1>(c, h, r_knee, 0.022f, mu_body, Rw[1], pw[1], om[1], vl[1], aa[1], al[1], R0, B.p, axw, pw, f[1], S, Mbb);
  const RC c_hip = leg_contact<0>(c, h, r_hip, LT_HIP_CYL_RADIUS, mu_body, Rw[0], pw[0], om[0], vl[0], aa[0], al[0], R0, B.p, axw, pw, f[0], S, Mbb);
  // trunk corners of this lane (base accelerations are the unknowns: no velocity-product part)
  const V3 r_tlo = trunk_corner(leg, false), r_thi = trunk_corner(leg, true);
  RC c_tlo, c_thi;
  {
    I6 dummyI = Mbb;
    S6 pA; pA.a = pb.n; pA.l = pb.f;
    c_tlo = ground_contact(c, h, r_tlo, 0.f, mu_body, R0, B.p, wb, vb, dummyI, pA);
    c_thi = ground_contact(c, h, r_thi, 0.f, mu_body, R0, B.p, wb, vb, dummyI, pA);
    Mbb = dummyI; pb.n = pA.a; pb.f = pA.l;
  }
  // backward force pass
  {
    const F6 t2 = force_to_parent<1>(f[2], cq[2], sq[2], LC[2].r);
    f[1].n += t2.n; f[1].f += t2.f;
    const F6 t1 = force_to_parent<1>(f[1], cq[1], sq[1], LC[1].r);
    f[0].n += t1.n; f[0].f += t1.f;
    const F6 t0 = force_to_parent<0>(f[0], cq[0], sq[0], LC[0].r);
    pb.n += t0.n; pb.f += t0.f;
  }
  S.rhs[0] = G.tau[0] - f[0].n.x;
  S.rhs[1] = G.tau[1] - f[1].n.y;
  S.rhs[2] = G.tau[2] - f[2].n.y;

  // ---- the two parts that do not depend on this lane's leg dynamics: the carried cylinder (object_part) and the CRBA
  //      (crba_part) - computed here, or fetched from the helper waves that ran them beside the code above ----
  if (Ext::external) {
    ext.fetch(pe);
  } else {
    pe.crba = crba_part(sgn, cq, sq);
  }
  if (HAS_OBJ) {
    pb.n += pe.obj.pb_n; pb.f += pe.obj.pb_f;
    rep.obj_part = pe.obj.obj_part;
    rep.trunk_part += pe.obj.trunk_part;
    if (TAC) rep.plate = pe.obj.plate;
  }
  {
    const CrbaOut& cr = pe.crba;
    S.h00 += cr.h00; S.h01 += cr.h01; S.h02 += cr.h02; S.h11 += cr.h11; S.h12 += cr.h12; S.h22 += cr.h22;
    // (r04: the model's joint limits became implicit spring-dampers inside the solve, lt_cfg.joint_limit_*; this frozen formulation follows the model)
    for (int k = 0; k < 3; ++k) {
      const float Bl = c.joint_limit_kp * h + c.joint_limit_kd;
      const float dlo = k_joint_lo[k] - G.q[k], dhi = G.q[k] - k_joint_hi[k];
      const float sg = dlo > dhi ? 1.f : -1.f, d = dlo > dhi ? dlo : dhi;
      const float f0 = c.joint_limit_kp * d - Bl * sg * G.qd[k];
      if (d - h * sg * G.qd[k] > 0.f && f0 > 0.f) {
        S.rhs[k] += sg * f0;
        if (k == 0) S.h00 += h * Bl; else if (k == 1) S.h11 += h * Bl; else S.h22 += h * Bl;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { S.bn[j] += cr.bn[j]; S.bl[j] += cr.bl[j]; }
    Mbb.A.m[0] += cr.Io.xx; Mbb.A.m[1] += cr.Io.xy; Mbb.A.m[2] += cr.Io.xz;
    Mbb.A.m[3] += cr.Io.xy; Mbb.A.m[4] += cr.Io.yy; Mbb.A.m[5] += cr.Io.yz;
    Mbb.A.m[6] += cr.Io.xz; Mbb.A.m[7] += cr.Io.yz; Mbb.A.m[8] += cr.Io.zz;
    Mbb.B += skew_of(cr.mc);
    Mbb.C.m[0] += cr.m; Mbb.C.m[4] += cr.m; Mbb.C.m[8] += cr.m;
  }

  // ---- eliminate this leg's joints: H = L L^T (3x3), Y = L^-1 H_lb, z = L^-1 rhs ----
  float l10, l20, l21, i0, i1, i2;  // (only the inverse diagonal is ever used: one v_rsq_f32 each)
  {
    i0 = frsqrt(S.h00);
    l10 = S.h01 * i0; l20 = S.h02 * i0;
    i1 = frsqrt(S.h11 - l10 * l10);
    l21 = (S.h12 - l20 * l10) * i1;
    i2 = frsqrt(S.h22 - l20 * l20 - l21 * l21);
  }
  V3 yn[3], yl[3];
  float z[3];
  yn[0] = i0 * S.bn[0]; yl[0] = i0 * S.bl[0]; z[0] = S.rhs[0] * i0;
  yn[1] = i1 * (S.bn[1] - l10 * yn[0]); yl[1] = i1 * (S.bl[1] - l10 * yl[0]); z[1] = (S.rhs[1] - l10 * z[0]) * i1;
  yn[2] = i2 * (S.bn[2] - l20 * yn[0] - l21 * yn[1]); yl[2] = i2 * (S.bl[2] - l20 * yl[0] - l21 * yl[1]);
  z[2] = (S.rhs[2] - l20 * z[0] - l21 * z[1]) * i2;
  V3 rb_n = -pb.n, rb_f = -pb.f;  // base rhs share: -(bias wrench) - Y^T z
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Mbb.A -= outer(yn[j], yn[j]);
    Mbb.B -= outer(yn[j], yl[j]);
    Mbb.C -= outer(yl[j], yl[j]);
    rb_n -= z[j] * yn[j];
    rb_f -= z[j] * yl[j];
  }

  // ---- floating base: quad-sum the four shares, add the trunk's own rigid-body terms, solve 6x6 ----
  I6 M;
  M.A = qsum_sym(Mbb.A); M.B = qsum_full(Mbb.B); M.C = qsum_sym(Mbb.C);
  V3 r0a = qsum(rb_n), r0l = qsum(rb_f);
  {
    const float mt = LT_TRUNK_MASS + X.trunk_mass_add;
    Rigid T;
    T.m = mt;
    const V3 ctr = v3(k_trunk_com[0], k_trunk_com[1], k_trunk_com[2]);
    T.mc = mt * ctr;
    T.Io = s3_from(inertia_about_origin(mt, ctr, k_trunk_icom, mt / LT_TRUNK_MASS));
    const F6 ft = rigid_bias(T, wb, vb, v3(0, 0, 0), v3(0, 0, 0), (-g) * row(R0, 2));
    r0a -= ft.n; r0l -= ft.f;
    M.A.m[0] += T.Io.xx; M.A.m[1] += T.Io.xy; M.A.m[2] += T.Io.xz;
    M.A.m[3] += T.Io.xy; M.A.m[4] += T.Io.yy; M.A.m[5] += T.Io.yz;
    M.A.m[6] += T.Io.xz; M.A.m[7] += T.Io.yz; M.A.m[8] += T.Io.zz;
    M.B += skew_of(T.mc);
    M.C.m[0] += mt; M.C.m[4] += mt; M.C.m[8] += mt;
  }
  V3 a0a, a0l;
  spd6_solve(M, r0a, r0l, a0a, a0l);

  // ---- back-substitute the joints: qdd = L^-T (z - Y a_b) ----
  float qdd[3];
  {
    const float t0 = z[0] - dot(yn[0], a0a) - dot(yl[0], a0l);
    const float t1 = z[1] - dot(yn[1], a0a) - dot(yl[1], a0l);
    const float t2 = z[2] - dot(yn[2], a0a) - dot(yl[2], a0l);
    qdd[2] = t2 * i2;
    qdd[1] = (t1 - l21 * qdd[2]) * i1;
    qdd[0] = (t0 - l10 * qdd[1] - l20 * qdd[2]) * i0;
  }
  // total link accelerations (for the final contact forces)
  V3 ta[3], tl[3];
  ta[0] = rot_inv<0>(cq[0], sq[0], a0a) + ca[0] + axis_scaled<0>(qdd[0]);
  tl[0] = rot_inv<0>(cq[0], sq[0], a0l + cross(a0a, LC[0].r)) + cl[0];
  ta[1] = rot_inv<1>(cq[1], sq[1], ta[0]) + ca[1] + axis_scaled<1>(qdd[1]);
  tl[1] = rot_inv<1>(cq[1], sq[1], tl[0] + cross(ta[0], LC[1].r)) + cl[1];
  ta[2] = rot_inv<1>(cq[2], sq[2], ta[1]) + ca[2] + axis_scaled<1>(qdd[2]);
  tl[2] = rot_inv<1>(cq[2], sq[2], tl[1] + cross(ta[1], LC[2].r)) + cl[2];
  rep.body[3] = ground_force(c_foot, h, r_foot, LT_FOOT_RADIUS, Rw[2], ta[2], tl[2]);
  rep.body[2] = ground_force(c_calf, h, r_calf, 0.012f, Rw[2], ta[2], tl[2]);
  rep.body[1] = ground_force(c_knee, h, r_knee, 0.022f, Rw[1], ta[1], tl[1]);
  rep.body[0] = ground_force(c_hip, h, r_hip, LT_HIP_CYL_RADIUS, Rw[0], ta[0], tl[0]);
  rep.trunk_part += ground_force(c_tlo, h, r_tlo, 0.f, R0, a0a, a0l);
  rep.trunk_part += ground_force(c_thi, h, r_thi, 0.f, R0, a0a, a0l);

  // ---- semi-implicit Euler ----
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float qd = G.qd[k] + h * qdd[k];
    G.q[k] += h * qd; G.qd[k] = qd;
  }
  {
    const V3 acl = a0l + cross(wb, vb);
    B.u += h * mul(R0, acl);
    B.w += h * mul(R0, a0a);
    B.p += h * B.u;
    B.q = q_integrate(B.q, B.w, h);
  }
}
