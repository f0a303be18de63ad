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
  float m_exy, m_eyaw, m_airvar;  // the command term's per-env metrics as of the end of the last step (commands.py:392-396)
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

// =====================================================================================================
// Packed-pair formulation of the leg dynamics (lt_device_math.h "packed pairs").  What is paired:
//   * (omega | velocity-product angular acceleration) and (v | velocity-product linear acceleration) of every link: the
//     kinematics recursion, I v and I a of the bias wrench, the velocity and the velocity-product acceleration of a contact
//     point all run once for both halves.  Gravity rides in the linear half as the fictitious base acceleration +g z, so no
//     link has a gravity term of its own (the contact points subtract it again: one operation);
//   * (moment | force) of every spatial force: the backward pass, the joints' coupling rows to the base, the Schur update
//     of the base block - whose A and C blocks take the same outer products on the two halves, and whose B block takes
//     B_ik and B_ki from one product with the halves of one operand swapped;
//   * the Jacobian columns of the thigh and calf joints (their axes are the same world vector).
// =====================================================================================================
// base block [A B; B^T C] of the quad's 6x6 with the halves that take the same arithmetic side by side:
//   ac[k] = (A_k | C_k), k = xx xy xz yy yz zz;   bo[] = (B01 | B10), (B02 | B20), (B12 | B21);   bd[i] = B_ii
struct I6p { f2 ac[6]; f2 bo[3]; float bd[3]; };
__device__ __forceinline__ I6p i6p_zero() {
  I6p o;
#pragma unroll
  for (int k = 0; k < 6; ++k) o.ac[k] = sp2(0.f);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o.bo[k] = sp2(0.f); o.bd[k] = 0.f; }
  return o;
}
// M += s u u^T for the 6-vector u = (u.lo ; u.hi)   [su = s u]
__device__ __forceinline__ void i6p_rank1(I6p& M, const P3& su, const P3& u) {
  M.ac[0] += su.x * u.x; M.ac[1] += su.x * u.y; M.ac[2] += su.x * u.z; M.ac[3] += su.y * u.y; M.ac[4] += su.y * u.z; M.ac[5] += su.z * u.z;
  M.bo[0] += su.x * swp(u.y); M.bo[1] += su.x * swp(u.z); M.bo[2] += su.y * swp(u.z);
  M.bd[0] += su.x.x * u.x.y; M.bd[1] += su.y.x * u.y.y; M.bd[2] += su.z.x * u.z.y;
}
// M += rigid inertia (Io about the origin, first moment mc, mass m): A += Io, B += mc~, C += m 1
__device__ __forceinline__ void i6p_add_rigid(I6p& M, const S3& Io, V3 mc, float m) {
  M.ac[0] += mk2(Io.xx, m); M.ac[1].x += Io.xy; M.ac[2].x += Io.xz; M.ac[3] += mk2(Io.yy, m); M.ac[4].x += Io.yz; M.ac[5] += mk2(Io.zz, m);
  M.bo[0] += mk2(-mc.z, mc.z); M.bo[1] += mk2(mc.y, -mc.y); M.bo[2] += mk2(-mc.x, mc.x);
}
// M += h J^T B J, J = [-r~ 1], B = cte 1 + (Bn - cte) n n^T, n in the coords of the 6x6:
//   A += a (|r|^2 1 - r r^T) + d m m^T,  B += a r~ + d m n^T,  C += a 1 + d n n^T   with a = h cte, d = h (Bn - cte), m = r x n
__device__ __forceinline__ void i6p_add_contact(I6p& M, V3 r, V3 n, float a, float hbn) {  // a = h cte, hbn = h Bn
  const float d = hbn - a;
  const P3 u = pair(cross(r, n), n);
  i6p_rank1(M, d * u, u);
  const V3 ar = a * r;
  const float rr = a * dot(r, r);
  M.ac[0] += mk2(rr - ar.x * r.x, a); M.ac[3] += mk2(rr - ar.y * r.y, a); M.ac[5] += mk2(rr - ar.z * r.z, a);
  M.ac[1].x -= ar.x * r.y; M.ac[2].x -= ar.x * r.z; M.ac[4].x -= ar.y * r.z;
  M.bo[0] += mk2(-ar.z, ar.z); M.bo[1] += mk2(ar.y, -ar.y); M.bo[2] += mk2(-ar.x, ar.x);
}
__device__ __forceinline__ I6p qsum6(const I6p& a) {
  I6p o;
#pragma unroll
  for (int k = 0; k < 6; ++k) o.ac[k] = qsum(a.ac[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o.bo[k] = qsum(a.bo[k]); o.bd[k] = qsum(a.bd[k]); }
  return o;
}
// 6x6 SPD solve on the packed block layout
__device__ __forceinline__ void spd6_solve(const I6p& M, const P3& b, V3& xa, V3& xl) {
  I6 F;
  F.A.m[0] = M.ac[0].x; F.A.m[1] = M.ac[1].x; F.A.m[2] = M.ac[2].x; F.A.m[4] = M.ac[3].x; F.A.m[5] = M.ac[4].x; F.A.m[8] = M.ac[5].x;
  F.A.m[3] = F.A.m[1]; F.A.m[6] = F.A.m[2]; F.A.m[7] = F.A.m[5];
  F.C.m[0] = M.ac[0].y; F.C.m[1] = M.ac[1].y; F.C.m[2] = M.ac[2].y; F.C.m[4] = M.ac[3].y; F.C.m[5] = M.ac[4].y; F.C.m[8] = M.ac[5].y;
  F.C.m[3] = F.C.m[1]; F.C.m[6] = F.C.m[2]; F.C.m[7] = F.C.m[5];
  F.B.m[0] = M.bd[0]; F.B.m[4] = M.bd[1]; F.B.m[8] = M.bd[2];
  F.B.m[1] = M.bo[0].x; F.B.m[3] = M.bo[0].y; F.B.m[2] = M.bo[1].x; F.B.m[6] = M.bo[1].y; F.B.m[5] = M.bo[2].x; F.B.m[7] = M.bo[2].y;
  spd6_solve(F, lo(b), hi(b), xa, xl);
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
    I6p Mo = i6p_zero();  // (the packed block layout of the leg dynamics: a contact's h J^T B J is ~30 operations instead of ~90)
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
          i6p_add_contact(Mo, rho_p, nw, h * lp.cte, h * lp.Bn);
          rhs_a += cross(rho_p, F0_p);
          rhs_l += F0_p;
        }
      }
    }
    Law lg; lg.active = false; lg.fx = lg.fy = lg.fn = lg.cte = lg.Bn = 0.f;
    V3 rho_g = v3(0, 0, 0), F0_g = v3(0, 0, 0);
    // (the two rim points: only when the cylinder can reach the ground at all - its lowest point lies above p.z - half - rad;
    //  on the robot's back that is ~0.3 m, and the ~50 operations of the points' construction were paid every substep)
    if (leg < 2 && O.p.z - half - rad < 0.f) {
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
          i6p_add_contact(Mo, rho_g, v3(0, 0, 1), h * lg.cte, h * lg.Bn);
          rhs_a += cross(rho_g, F0_g);
          rhs_l += F0_g;
        }
      }
    }
    I6p M = qsum6(Mo);
    rhs_a = qsum(rhs_a); rhs_l = qsum(rhs_l);
    const float m = O.mass;
    const float Iyy = 0.5f * m * rad * rad, Ixx = m * (3.f * rad * rad + O.len * O.len) / 12.f;
    M3 Iw = m3_diag(Ixx);
    Iw += outer((Iyy - Ixx) * ay, ay);
    M.ac[0] += mk2(Iw.m[0], m); M.ac[1].x += Iw.m[1]; M.ac[2].x += Iw.m[2];
    M.ac[3] += mk2(Iw.m[4], m); M.ac[4].x += Iw.m[5]; M.ac[5] += mk2(Iw.m[8], m);
    rhs_a -= cross(O.w, mul(Iw, O.w));
    rhs_l.z -= m * g;
    spd6_solve(M, pair(rhs_a, rhs_l), obj_aa, obj_al);
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

// child link's (omega | aa) and (v | al) from the parent's: rotate both halves, add the joint rate to omega and the
// velocity products c = v x (S qd) to the acceleration halves
template <int AX>
__device__ __forceinline__ void joint_fk2(const P3& Wp, const P3& Vp, V3 r, float c, float s, float qd, P3& W, P3& V) {
  W = rot_inv<AX>(c, s, Wp);
  V = rot_inv<AX>(c, s, Vp + cross(Wp, r));
  if (AX == 0) {
    W.x.x += qd;
    W.y.y += W.z.x * qd; W.z.y -= W.y.x * qd;
    V.y.y += V.z.x * qd; V.z.y -= V.y.x * qd;
  } else {
    W.y.x += qd;
    W.x.y -= W.z.x * qd; W.z.y += W.x.x * qd;
    V.x.y -= V.z.x * qd; V.z.y += V.x.x * qd;
  }
}
// M R for R = rot(AX, q): the two columns that mix come out of one packed product per row
template <int AX>
__device__ __forceinline__ M3 mul_rot2(const M3& a, float c, float s) {
  M3 o = a;
  const f2 cs = mk2(c, s);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (AX == 0) {  // (o1 | o2) = a1 (c | -s) + a2 (s | c)
      const f2 r = sp2(a.m[3 * i + 1]) * mk2(c, -s) + sp2(a.m[3 * i + 2]) * swp(cs);
      o.m[3 * i + 1] = r.x; o.m[3 * i + 2] = r.y;
    } else {        // (o0 | o2) = a0 (c | s) + a2 (-s | c)
      const f2 r = sp2(a.m[3 * i]) * cs + sp2(a.m[3 * i + 2]) * mk2(-s, c);
      o.m[3 * i] = r.x; o.m[3 * i + 2] = r.y;
    }
  }
  return o;
}
// wrench (n | f) a rigid body needs for velocity (W.lo, V.lo) and acceleration (W.hi, V.hi): I a + v x* I v
__device__ __forceinline__ P3 rigid_bias2(const Rigid& R, const P3& W, const P3& V) {
  const P3 N = mul(R.Io, W) + cross(R.mc, V);  // (n_v | n_a)
  const P3 F = cross(W, R.mc) + R.m * V;       // (f_v | f_a)
  const V3 om = lo(W), vl = lo(V), nv = lo(N), fv = lo(F);
  return pair(hi(N) + cross(om, nv) + cross(vl, fv), hi(F) + cross(om, fv));
}
// spatial force (n | f) re-expressed in the parent frame (child origin at r, child -> parent rotation R)
template <int AX>
__device__ __forceinline__ P3 force_to_parent2(const P3& x, float c, float s, V3 r) {
  P3 o = rot_fwd<AX>(c, s, x);
  const V3 t = cross(r, hi(o));
  o.x.x += t.x; o.y.x += t.y; o.z.x += t.z;
  return o;
}

// ---- everything of a substep that depends on the joint ANGLES and the velocities but not on the contacts: CRBA on rigid
//      composites (this leg's joint-space inertia, its coupling to the base, its rigid share of the base block) and the
//      velocity-product / gravity bias (RNEA with zero accelerations: joint torques and the wrench on the base).  Lane 0 of the
//      quad adds the trunk's own inertia and bias to its share.  A helper wave runs it beside the contacts (lt_env.hip). ----
struct CrbaOut {
  float h00, h01, h02, h11, h12, h22;
  V3 bn[3], bl[3];
  S3 Io; V3 mc; float m;  // composite of the whole leg (+ trunk, lane 0) about the base origin
  float tb[3];            // bias torque of the three joints
  V3 pbn, pbf;            // bias wrench on the base (moment, force), this lane's share
};
__device__ __forceinline__ CrbaOut crba_part(const float (&sgn)[4], int leg, const float (&cq)[3], const float (&sq)[3], const float (&qd)[3],
                                             const P3& Wb, const P3& Vb, float trunk_mass_add) {
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
    // joint 2 (calf, axis y): column of the composite, carried down to the base as a packed (moment | force)
    const P3 F3 = pair(s3_col<1>(C3.Io), cross(axis_scaled<1>(1.f), C3.mc));
    o.h22 = C3.Io.yy;
    const P3 F32 = force_to_parent2<1>(F3, cq[2], sq[2], LC[2].r);
    o.h12 = F32.y.x;
    const P3 F31 = force_to_parent2<1>(F32, cq[1], sq[1], LC[1].r);
    o.h02 = F31.x.x;
    const P3 F30 = force_to_parent2<0>(F31, cq[0], sq[0], LC[0].r);
    o.bn[2] = lo(F30); o.bl[2] = hi(F30);
    // joint 1 (thigh, axis y)
    const P3 F2 = pair(s3_col<1>(C2.Io), cross(axis_scaled<1>(1.f), C2.mc));
    o.h11 = C2.Io.yy;
    const P3 F21 = force_to_parent2<1>(F2, cq[1], sq[1], LC[1].r);
    o.h01 = F21.x.x;
    const P3 F20 = force_to_parent2<0>(F21, cq[0], sq[0], LC[0].r);
    o.bn[1] = lo(F20); o.bl[1] = hi(F20);
    // joint 0 (hip, axis x)
    const P3 F1 = pair(s3_col<0>(C1.Io), cross(axis_scaled<0>(1.f), C1.mc));
    o.h00 = C1.Io.xx;
    const P3 F10 = force_to_parent2<0>(F1, cq[0], sq[0], LC[0].r);
    o.bn[0] = lo(F10); o.bl[0] = hi(F10);
    o.Io = Cb.Io; o.mc = Cb.mc; o.m = Cb.m;  // rigid share of the base block
  }
  // velocity-product / gravity bias of the three links (the same kinematics recursion as physics_substep's: inline, one copy)
  P3 W[3], V[3];
  joint_fk2<0>(Wb, Vb, LC[0].r, cq[0], sq[0], qd[0], W[0], V[0]);
  joint_fk2<1>(W[0], V[0], LC[1].r, cq[1], sq[1], qd[1], W[1], V[1]);
  joint_fk2<1>(W[1], V[1], LC[2].r, cq[2], sq[2], qd[2], W[2], V[2]);
  P3 f2 = rigid_bias2(RB[2], W[2], V[2]);
  P3 f1 = rigid_bias2(RB[1], W[1], V[1]) + force_to_parent2<1>(f2, cq[2], sq[2], LC[2].r);
  P3 f0 = rigid_bias2(RB[0], W[0], V[0]) + force_to_parent2<1>(f1, cq[1], sq[1], LC[1].r);
  P3 pb = force_to_parent2<0>(f0, cq[0], sq[0], LC[0].r);
  o.tb[0] = f0.x.x; o.tb[1] = f1.y.x; o.tb[2] = f2.y.x;
  {  // the trunk's own rigid body: once per env (lane 0's share)
    const float on = leg == 0 ? 1.f : 0.f;
    const float mt = (LT_TRUNK_MASS + trunk_mass_add) * on;
    Rigid T;
    T.m = mt;
    const V3 ctr = v3(k_trunk_com[0], k_trunk_com[1], k_trunk_com[2]);
    T.mc = mt * ctr;
    T.Io = s3_from(inertia_about_origin(mt, ctr, k_trunk_icom, mt / LT_TRUNK_MASS));
    pb += rigid_bias2(T, Wb, Vb);
    o.Io.xx += T.Io.xx; o.Io.xy += T.Io.xy; o.Io.xz += T.Io.xz; o.Io.yy += T.Io.yy; o.Io.yz += T.Io.yz; o.Io.zz += T.Io.zz;
    o.mc += T.mc; o.m += mt;
  }
  o.pbn = lo(pb); o.pbf = hi(pb);
  return o;
}

struct LegSys2 {       // this lane's 3 joints: H (sym 3x3), coupling rows to the base (moment | force), right-hand side
  float h00, h01, h02, h11, h12, h22;
  P3 b[3];
  float rhs[3];
};
// what the post-solve force of a contact needs: the law's damping, and the explicit force with the velocity-product
// acceleration of the point already folded in
struct RC2 { bool active; float hct, hbn; V3 F0; };
__device__ __forceinline__ RC2 rc2_none() { RC2 o; o.active = false; o.hct = o.hbn = 0.f; o.F0 = v3(0, 0, 0); return o; }

// the contact law of one ground-contact sphere (centre r in the link frame, radius rho) of a link with packed motion (W, V):
// contact point, its velocity and velocity-product acceleration (world), the law.  BRANCHY: early exits (the spheres that rarely
// touch); otherwise straight-line code whose result is all zeros when the sphere is off the ground (the foot: nearly always on it
// in some lane of the wave, and as straight-line code its contributions INITIALISE the accumulators - no zero fill, no merge copies).
template <bool BRANCHY>
__device__ __forceinline__ RC2 contact_eval(const lt_cfg& c, float h, V3 rc, float mu, const M3& Rw, V3 pwk, const P3& W, const P3& V, V3& Pc) {
  RC2 out = rc2_none();
  Pc = pwk + mul(Rw, rc);
  const float d = -Pc.z;
  if (BRANCHY && !(d > 0.f)) return out;
  const P3 PV = mul(Rw, V + cross(W, rc));  // (velocity | velocity-product acceleration + g z) of the point, world
  // contact_law, inlined for the two forms
  float ramp = d / c.contact_ramp;
  ramp = ramp > 1.f ? 1.f : ramp;
  const float Bn = c.ground_kn * h + c.ground_cn * ramp;
  const float f0n = c.ground_kn * d - Bn * PV.z.x;
  const bool active = (d > 0.f) && (f0n > 0.f);
  if (BRANCHY && !active) return out;
  const float vt = fsqrt(PV.x.x * PV.x.x + PV.y.x * PV.y.x);
  float cte = mu * f0n / (vt > 1e-6f ? vt : 1e-6f);
  cte = cte > c.ground_ct ? c.ground_ct : cte;
  const float hct = active ? h * cte : 0.f, hbn = active ? h * Bn : 0.f;
  cte = active ? cte : 0.f;
  out.active = active; out.hct = hct; out.hbn = hbn;
  // explicit part with the velocity-product acceleration of the point folded in (the fictitious +g z taken out again)
  out.F0 = v3(-cte * PV.x.x - hct * PV.x.y, -cte * PV.y.x - hct * PV.y.y, (active ? f0n : 0.f) - hbn * (PV.z.y - c.gravity));
  return out;
}
// implicit + explicit contributions of an evaluated contact on link K (0 hip, 1 thigh, 2 calf) to the link force, the joint
// system and the base block.  INIT: the accumulators are written, not added to (first contact of the substep).
template <int K, bool INIT>
__device__ __forceinline__ void leg_contact_add(float h, const RC2& k, V3 rc, V3 Pc, const M3& Rw, const M3& R0, V3 p0, V3 ax0, V3 ax12,
                                                const V3 (&pw)[3], P3& Fk, LegSys2& S, I6p& Mbb) {
  const float hct = k.hct, hbn = k.hbn;
  const V3 f0b = tmul(Rw, k.F0);
  Fk -= pair(cross(rc, f0b), f0b);
  // implicit part through the Jacobian columns (world frame; the ground contact frame is world-aligned => B diagonal)
  const V3 rho_b = tmul(R0, Pc - p0);
  const V3 w0 = cross(ax0, Pc - pw[0]);
  const V3 g0 = v3(hct * w0.x, hct * w0.y, hbn * w0.z);
  {
    const V3 gb = tmul(R0, g0);
    const P3 t = pair(cross(rho_b, gb), gb);
    if (INIT) { S.b[0] = t; S.h00 = dot(w0, g0); } else { S.b[0] += t; S.h00 += dot(w0, g0); }
  }
  if (K >= 1) {  // thigh and calf joints: one axis, two lever arms
    const P3 W12 = cross(ax12, pair(Pc - pw[1], Pc - pw[2]));
    const P3 G12 = p3(sp2(hct) * W12.x, sp2(hct) * W12.y, sp2(hbn) * W12.z);
    const P3 GB = tmul(R0, G12);
    const V3 gb1 = lo(GB);
    const P3 t1 = pair(cross(rho_b, gb1), gb1);
    const f2 h0x = dot(w0, G12), hxx = dot(W12, G12);
    if (INIT) { S.b[1] = t1; S.h01 = h0x.x; S.h11 = hxx.x; } else { S.b[1] += t1; S.h01 += h0x.x; S.h11 += hxx.x; }
    if (K >= 2) {
      const V3 gb2 = hi(GB);
      const P3 t2 = pair(cross(rho_b, gb2), gb2);
      const float h12 = W12.x.x * G12.x.y + W12.y.x * G12.y.y + W12.z.x * G12.z.y;
      if (INIT) { S.b[2] = t2; S.h02 = h0x.y; S.h22 = hxx.y; S.h12 = h12; } else { S.b[2] += t2; S.h02 += h0x.y; S.h22 += hxx.y; S.h12 += h12; }
    }
  }
  if (INIT) Mbb = i6p_zero();
  i6p_add_contact(Mbb, rho_b, row(R0, 2), hct, hbn);
}
// ground contact of a trunk corner (centre r in the base frame); (wb, vb) base velocity in base coords
__device__ __forceinline__ RC2 trunk_eval(const lt_cfg& c, float h, V3 r, float mu, const M3& R0, V3 p0, V3 wb, V3 vb) {
  RC2 out = rc2_none();
  const V3 zb = row(R0, 2);
  if (!(p0.z + dot(zb, r) < 0.f)) return out;
  const V3 Pc = p0 + mul(R0, r);
  const V3 vw = mul(R0, vb + cross(wb, r));
  const Law law = contact_law(-Pc.z, vw, c.ground_kn, c.ground_cn, c.ground_ct, mu, c.contact_ramp, h);
  out.F0 = v3(law.fx, law.fy, law.fn);
  if (!law.active) return out;
  out.active = true; out.hct = h * law.cte; out.hbn = h * law.Bn;
  return out;
}
__device__ __forceinline__ void trunk_add(float h, const RC2& k, V3 r, const M3& R0, I6p& Mbb, P3& pb) {
  if (!k.active) return;
  i6p_add_contact(Mbb, r, row(R0, 2), k.hct, k.hbn);
  const V3 f0b = tmul(R0, k.F0);
  pb -= pair(cross(r, f0b), f0b);
}
// final (post-solve) force of a contact at rc on a link whose acceleration BEYOND the velocity products is d = (ang | lin)
__device__ __forceinline__ V3 contact_force2(const RC2& k, V3 rc, const M3& Rw, const P3& d) {
  const V3 aw = mul(Rw, hi(d) + cross(lo(d), rc));
  return v3(k.F0.x - k.hct * aw.x, k.F0.y - k.hct * aw.y, k.F0.z - k.hbn * aw.z);  // (inactive: hct = hbn = 0, F0 = 0)
}
// one of the rarely touching spheres of link K: its contributions before the solve; the record the force after the solve needs
template <int K>
__device__ __forceinline__ RC2 rare_contact(const lt_cfg& c, float h, V3 r, float rho, float mu, const M3& Rw, V3 pwk, const P3& W, const P3& V,
                                            const M3& R0, V3 p0, V3 ax0, V3 ax12, const V3 (&pw)[3], P3& Fk, LegSys2& S, I6p& Mbb) {
  const V3 rc = r - rho * row(Rw, 2);
  V3 Pc;
  const RC2 k = contact_eval<true>(c, h, rc, mu, Rw, pwk, W, V, Pc);
  if (k.active) leg_contact_add<K, false>(h, k, rc, Pc, Rw, R0, p0, ax0, ax12, pw, Fk, S, Mbb);
  return k;
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
  __device__ __forceinline__ void publish(const float (&)[3], const float (&)[3], const float (&)[3], const Base&) const {}
  __device__ __forceinline__ void fetch(PhysExt&) const {}
  __device__ __forceinline__ void mark(int) const {}  // (phase stamps of the LT_STAMPS build, helper form)
};
template <bool HAS_OBJ, bool TAC = false, class Ext = InlineParts>
__device__ __forceinline__ void physics_substep(const lt_cfg& c, float h, int leg, const float (&sgn)[4], Base& B, Leg& G, Obj& O,
                                                const Misc& X, Report& rep, const Ext& ext = Ext()) {
  const float g = c.gravity;
  const LinkC LC[3] = {make_link<0>(sgn), make_link<1>(sgn), make_link<2>(sgn)};
  float cq[3], sq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { sq[k] = fsin(G.q[k]); cq[k] = fcos(G.q[k]); }
  ext.mark(0);
  ext.publish(cq, sq, G.qd, B);
  ext.mark(1);
  PhysExt pe;
  if (!Ext::external && HAS_OBJ) pe.obj = object_part<TAC>(c, h, leg, B, O, X.trunk_mu);  // first: nothing else is live yet
  const M3 R0 = quat_to_mat(B.q.w, B.q.x, B.q.y, B.q.z);
  const P3 WVb = tmul(R0, pair(B.w, B.u));
  const V3 wb = lo(WVb), vb = hi(WVb);
  // the base's packed motion: (omega | 0) and (v | +g z): gravity as a fictitious linear acceleration of the base
  const P3 Wb = pair(wb, v3(0, 0, 0)), Vb = pair(vb, g * row(R0, 2));
  P3 W[3], V[3];
  joint_fk2<0>(Wb, Vb, LC[0].r, cq[0], sq[0], G.qd[0], W[0], V[0]);
  joint_fk2<1>(W[0], V[0], LC[1].r, cq[1], sq[1], G.qd[1], W[1], V[1]);
  joint_fk2<1>(W[1], V[1], LC[2].r, cq[2], sq[2], G.qd[2], W[2], V[2]);
  M3 Rw[3];
  V3 pw[3];
  Rw[0] = mul_rot2<0>(R0, cq[0], sq[0]);    pw[0] = B.p + mul(R0, LC[0].r);
  Rw[1] = mul_rot2<1>(Rw[0], cq[1], sq[1]); pw[1] = pw[0] + mul(Rw[0], LC[1].r);
  Rw[2] = mul_rot2<1>(Rw[1], cq[2], sq[2]); pw[2] = pw[1] + mul(Rw[1], LC[2].r);
  const V3 ax0 = col(Rw[0], 0), ax12 = col(Rw[1], 1);  // joint axes in world (thigh and calf axes are parallel)

  // this lane's share of the base block / base bias wrench (moment | force)
  P3 pb = both(v3(0, 0, 0));
  rep.trunk_part = v3(0, 0, 0);
  rep.obj_part = v3(0, 0, 0);

  // ---- contact forces on this leg's links (the links' own bias forces come with the CRBA part) ----
  P3 f[3];
  f[0] = f[1] = f[2] = both(v3(0, 0, 0));
  LegSys2 S;
  I6p Mbb;
  const float mu_foot = G.mu * c.ground_mu, mu_body = c.ground_mu;
  const V3 r_foot = v3(0.f, 0.f, -0.213f), r_calf = v3(0.f, 0.f, -0.1065f), r_knee = v3(0.f, 0.f, -0.213f);
  const V3 r_hip = v3(0.f, LT_MIRROR_HIP_CYL_Y * sgn[LT_MIRROR_HIP_CYL_Y_PAT], 0.f);
  // the foot: straight-line code whose contributions initialise the joint system and the base block
  const V3 rc_foot = r_foot - LT_FOOT_RADIUS * row(Rw[2], 2);
  V3 Pc_foot;
  const RC2 c_foot = contact_eval<false>(c, h, rc_foot, mu_foot, Rw[2], pw[2], W[2], V[2], Pc_foot);
  leg_contact_add<2, true>(h, c_foot, rc_foot, Pc_foot, Rw[2], R0, B.p, ax0, ax12, pw, f[2], S, Mbb);
  // The five spheres that rarely touch (calf, knee, hip, two trunk corners) sit behind ONE test of their heights: as five
  // separate conditional regions they cost the common path ~440 operations of tests, accumulator copies and branch overhead
  // per substep.  (height of a sphere's lowest point: p.z + z_b . r - rho, z_b = world z in link coordinates)
  const V3 r_tlo = trunk_corner(leg, false), r_thi = trunk_corner(leg, true);
  const float z_calf = pw[2].z + dot(row(Rw[2], 2), r_calf) - 0.012f, z_knee = pw[1].z + dot(row(Rw[1], 2), r_knee) - 0.022f;
  const float z_hip = pw[0].z + dot(row(Rw[0], 2), r_hip) - LT_HIP_CYL_RADIUS;
  const float z_tlo = B.p.z + dot(row(R0, 2), r_tlo), z_thi = B.p.z + dot(row(R0, 2), r_thi);
#ifdef LT_PROBE_NO_RARE  // instruction-count probe of the common path (tools/asm_profile.py); never shipped
  const bool rare = false;
#else
  const bool rare = fminf(fminf(fminf(z_calf, z_knee), fminf(z_hip, z_tlo)), z_thi) < 0.f;
#endif
  RC2 k_calf = rc2_none(), k_knee = k_calf, k_hip = k_calf, k_tlo = k_calf, k_thi = k_calf;
  if (rare) {
    k_calf = rare_contact<2>(c, h, r_calf, 0.012f, mu_body, Rw[2], pw[2], W[2], V[2], R0, B.p, ax0, ax12, pw, f[2], S, Mbb);
    k_knee = rare_contact<1>(c, h, r_knee, 0.022f, mu_body, Rw[1], pw[1], W[1], V[1], R0, B.p, ax0, ax12, pw, f[1], S, Mbb);
    k_hip = rare_contact<0>(c, h, r_hip, LT_HIP_CYL_RADIUS, mu_body, Rw[0], pw[0], W[0], V[0], R0, B.p, ax0, ax12, pw, f[0], S, Mbb);
    // trunk corners of this lane (base accelerations are the unknowns: no velocity-product part)
    k_tlo = trunk_eval(c, h, r_tlo, mu_body, R0, B.p, wb, vb);
    trunk_add(h, k_tlo, r_tlo, R0, Mbb, pb);
    k_thi = trunk_eval(c, h, r_thi, mu_body, R0, B.p, wb, vb);
    trunk_add(h, k_thi, r_thi, R0, Mbb, pb);
  }
  // backward force pass
  f[1] += force_to_parent2<1>(f[2], cq[2], sq[2], LC[2].r);
  f[0] += force_to_parent2<1>(f[1], cq[1], sq[1], LC[1].r);
  pb += force_to_parent2<0>(f[0], cq[0], sq[0], LC[0].r);

  // ---- the two parts that do not depend on this lane's leg dynamics: the carried cylinder (object_part) and the CRBA
  //      (crba_part) - computed here, or fetched from the helper waves that ran them beside the code above ----
  ext.mark(2);
  if (Ext::external) {
    ext.fetch(pe);
  } else {
    pe.crba = crba_part(sgn, leg, cq, sq, G.qd, Wb, Vb, X.trunk_mass_add);
  }
  if (HAS_OBJ) {
    pb += pair(pe.obj.pb_n, pe.obj.pb_f);
    rep.obj_part = pe.obj.obj_part;
    rep.trunk_part += pe.obj.trunk_part;
    if (TAC) rep.plate = pe.obj.plate;
  }
  {
    const CrbaOut& cr = pe.crba;
    S.h00 += cr.h00; S.h01 += cr.h01; S.h02 += cr.h02; S.h11 += cr.h11; S.h12 += cr.h12; S.h22 += cr.h22;
#pragma unroll
    for (int j = 0; j < 3; ++j) S.b[j] += pair(cr.bn[j], cr.bl[j]);
    i6p_add_rigid(Mbb, cr.Io, cr.mc, cr.m);
    pb += pair(cr.pbn, cr.pbf);
    S.rhs[0] = G.tau[0] - f[0].x.x - cr.tb[0];
    S.rhs[1] = G.tau[1] - f[1].y.x - cr.tb[1];
    S.rhs[2] = G.tau[2] - f[2].y.x - cr.tb[2];
  }
  // ---- joint limits: unilateral implicit spring-dampers on the joint coordinates (lt_cfg.joint_limit_*; oracle: joint_limit()).
  //      Beyond a limit by d the torque along the inward direction s is  k d - B s qd_new,  B = k h + c:  s f0 on the right-hand
  //      side, h B on the joint's diagonal - an internal torque between parent and child, solved with everything else.
  {
    const float Bl = c.joint_limit_kp * h + c.joint_limit_kd, hB = h * Bl;
    float add[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float dlo = k_joint_lo[k] - G.q[k], dhi = G.q[k] - k_joint_hi[k];
      const float sg = dlo > dhi ? 1.f : -1.f, d = dlo > dhi ? dlo : dhi;  // the nearer limit: its inward direction, the excursion beyond it (< 0 inside)
      const float f0 = c.joint_limit_kp * d - Bl * sg * G.qd[k];
      // active when the joint WOULD be beyond the limit at the end of this step at its present velocity (d may still be negative:
      // at 30 rad/s a joint travels 0.15 rad per step - a penalty that waits for d > 0 lets it through that far), and only pushes
      const bool on = d - h * sg * G.qd[k] > 0.f && f0 > 0.f;
      S.rhs[k] += on ? sg * f0 : 0.f;
      add[k] = on ? hB : 0.f;
    }
    S.h00 += add[0]; S.h11 += add[1]; S.h22 += add[2];
  }

  ext.mark(3);
  // ---- eliminate this leg's joints: H = L L^T (3x3), Y = L^-1 H_lb (rows (moment | force)), z = L^-1 rhs ----
  float l10, l20, l21, i0, i1, i2;  // (only the inverse diagonal is ever used: one v_rsq_f32 each)
  {
    i0 = frsqrt(S.h00);
    l10 = S.h01 * i0; l20 = S.h02 * i0;
    i1 = frsqrt(S.h11 - l10 * l10);
    l21 = (S.h12 - l20 * l10) * i1;
    i2 = frsqrt(S.h22 - l20 * l20 - l21 * l21);
  }
  P3 y[3];
  float z[3];
  y[0] = i0 * S.b[0]; z[0] = S.rhs[0] * i0;
  y[1] = i1 * (S.b[1] - l10 * y[0]); z[1] = (S.rhs[1] - l10 * z[0]) * i1;
  y[2] = i2 * (S.b[2] - l20 * y[0] - l21 * y[1]); z[2] = (S.rhs[2] - l20 * z[0] - l21 * z[1]) * i2;
  P3 rb = -pb;  // base rhs share: -(bias wrench) - Y^T z
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    i6p_rank1(Mbb, -y[j], y[j]);
    rb -= z[j] * y[j];
  }

  ext.mark(4);
  // ---- floating base: quad-sum the four shares (lane 0's carries the trunk's own rigid body), solve 6x6 ----
  const I6p M = qsum6(Mbb);
  const P3 r0 = qsum(rb);
  ext.mark(5);
  V3 a0a, a0l;
  spd6_solve(M, r0, a0a, a0l);
  ext.mark(6);

  // ---- back-substitute the joints: qdd = L^-T (z - Y a_b) ----
  float qdd[3];
  const P3 A0 = pair(a0a, a0l);
  {
    const f2 d0 = dot(y[0], A0), d1 = dot(y[1], A0), d2 = dot(y[2], A0);
    const float t0 = z[0] - d0.x - d0.y, t1 = z[1] - d1.x - d1.y, t2 = z[2] - d2.x - d2.y;
    qdd[2] = t2 * i2;
    qdd[1] = (t1 - l21 * qdd[2]) * i1;
    qdd[0] = (t0 - l10 * qdd[1] - l20 * qdd[2]) * i0;
  }
  // link accelerations beyond the velocity products (for the final contact forces): base and joint accelerations propagated
  P3 dl[3];
  {
    P3 t = A0;
    V3 cr = cross(a0a, LC[0].r);
    t.x.y += cr.x; t.y.y += cr.y; t.z.y += cr.z;
    dl[0] = rot_inv<0>(cq[0], sq[0], t); dl[0].x.x += qdd[0];
    t = dl[0]; cr = cross(lo(dl[0]), LC[1].r);
    t.x.y += cr.x; t.y.y += cr.y; t.z.y += cr.z;
    dl[1] = rot_inv<1>(cq[1], sq[1], t); dl[1].y.x += qdd[1];
    t = dl[1]; cr = cross(lo(dl[1]), LC[2].r);
    t.x.y += cr.x; t.y.y += cr.y; t.z.y += cr.z;
    dl[2] = rot_inv<1>(cq[2], sq[2], t); dl[2].y.x += qdd[2];
  }
  rep.body[3] = contact_force2(c_foot, rc_foot, Rw[2], dl[2]);
  rep.body[2] = rep.body[1] = rep.body[0] = v3(0, 0, 0);
  if (rare) {
    // (the hip and thigh rotations again, from inputs the optimiser cannot tie to the first evaluation: 18 registers that do
    //  not stay live across the solve)
    const M3 Rh = mul_rot2<0>(R0, opaque(cq[0]), opaque(sq[0]));
    const M3 Rt = mul_rot2<1>(Rh, opaque(cq[1]), opaque(sq[1]));
    rep.body[2] = contact_force2(k_calf, r_calf - 0.012f * row(Rw[2], 2), Rw[2], dl[2]);
    rep.body[1] = contact_force2(k_knee, r_knee - 0.022f * row(Rt, 2), Rt, dl[1]);
    rep.body[0] = contact_force2(k_hip, r_hip - LT_HIP_CYL_RADIUS * row(Rh, 2), Rh, dl[0]);
    rep.trunk_part += contact_force2(k_tlo, r_tlo, R0, A0);
    rep.trunk_part += contact_force2(k_thi, r_thi, R0, A0);
  }

  ext.mark(7);
  // ---- semi-implicit Euler ----
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float qd = G.qd[k] + h * qdd[k];
    G.q[k] += h * qd; G.qd[k] = qd;  // (joint limits act inside the solve above: no clamp)
  }
  {
    const V3 acl = a0l + cross(wb, vb);
    const P3 d = mul(R0, pair(a0a, acl));
    B.u += h * hi(d);
    B.w += h * lo(d);
    B.p += h * B.u;
    B.q = q_integrate(B.q, B.w, h);
  }
}
