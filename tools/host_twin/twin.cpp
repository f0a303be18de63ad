// Host twin of the step kernel's physics: runs csrc/lt_physics_crba.h (the product formulation) and the frozen scalar
// formulation (ref_physics_scalar.h) on the CPU - four lock-stepped threads per env stand for the quad's lanes - on random
// plausible states and reports the largest disagreement.  Test tool (tests/test_host_twin.py); nothing here ships.
//
//   build: clang++ -std=c++20 -O1 -I tools/host_twin -I include -DLT_PRIMS_H='"twin_prims.h"' tools/host_twin/twin.cpp locotouch_amd/csrc/lt_cfg.cpp -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../../locotouch_amd/csrc/lt_device_math.h"
#include "../../include/lt_env.h"
#include "../../include/lt_go1_model.h"

using namespace lt;
namespace ref {
#include "ref_physics_scalar.h"
}
namespace neu {
#include "../../locotouch_amd/csrc/lt_physics_crba.h"
}

struct State {
  float bp[3], bq[4], bu[3], bw[3];
  float q[4][3], qd[4][3], tau[4][3], mu[4];
  float op[3], oq[4], ou[3], ow[3], orad, olen, omass, omu;
  float trunk_mass_add, trunk_mu;
};
struct Out {
  float v[4][64];  // per lane: everything the substeps leave behind
  int n;
};

template <class NS_Base, class NS_Leg, class NS_Obj, class NS_Misc, class NS_Report, class Fn>
static void run_quad(const lt_cfg& c, const State& s, int nsub, bool has_obj, Out& out, Fn substep) {
  TwinQuad quad;
  std::thread th[4];
  for (int leg = 0; leg < 4; ++leg)
    th[leg] = std::thread([&, leg] {
      t_quad = &quad; t_lane = leg;
      const float sx = leg < 2 ? 1.f : -1.f, sy = (leg & 1) ? 1.f : -1.f;
      const float sgn[4] = {1.f, sx, sy, sx * sy};
      NS_Base B; NS_Leg G; NS_Obj O; NS_Misc X; NS_Report rep;
      B.p = v3(s.bp[0], s.bp[1], s.bp[2]); B.q.w = s.bq[0]; B.q.x = s.bq[1]; B.q.y = s.bq[2]; B.q.z = s.bq[3];
      B.u = v3(s.bu[0], s.bu[1], s.bu[2]); B.w = v3(s.bw[0], s.bw[1], s.bw[2]);
      for (int k = 0; k < 3; ++k) { G.q[k] = s.q[leg][k]; G.qd[k] = s.qd[leg][k]; G.tau[k] = s.tau[leg][k]; }
      G.mu = s.mu[leg];
      O.p = v3(s.op[0], s.op[1], s.op[2]); O.q.w = s.oq[0]; O.q.x = s.oq[1]; O.q.y = s.oq[2]; O.q.z = s.oq[3];
      O.u = v3(s.ou[0], s.ou[1], s.ou[2]); O.w = v3(s.ow[0], s.ow[1], s.ow[2]);
      O.rad = s.orad; O.len = s.olen; O.mass = s.omass; O.mu = s.omu;
      X.trunk_mass_add = s.trunk_mass_add; X.trunk_mu = s.trunk_mu;
      rep.plate = v3(0, 0, 0);
      const float h = c.sim_dt / (float)c.phys_substeps;
      for (int i = 0; i < nsub; ++i) substep(c, h, leg, sgn, B, G, O, X, rep, has_obj);
      float* o = out.v[leg];
      int n = 0;
      auto put3 = [&](V3 a) { o[n++] = a.x; o[n++] = a.y; o[n++] = a.z; };
      put3(B.p); o[n++] = B.q.w; o[n++] = B.q.x; o[n++] = B.q.y; o[n++] = B.q.z; put3(B.u); put3(B.w);
      for (int k = 0; k < 3; ++k) { o[n++] = G.q[k]; o[n++] = G.qd[k]; }
      if (has_obj) { put3(O.p); o[n++] = O.q.w; o[n++] = O.q.x; o[n++] = O.q.y; o[n++] = O.q.z; put3(O.u); put3(O.w); put3(rep.obj_part); put3(rep.plate); }
      for (int b = 0; b < 4; ++b) put3(rep.body[b]);
      put3(rep.trunk_part);
      out.n = n;
    });
  for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
  const int nstates = argc > 1 ? atoi(argv[1]) : 400;
  const int nsub = argc > 2 ? atoi(argv[2]) : 4;
  const double tol = argc > 3 ? atof(argv[3]) : 2e-4;
  double worst_all = 0;
  for (int task = 0; task < 2; ++task) {
    lt_cfg c;
    if (lt_cfg_preset(task == 0 ? "Isaac-RandCylinderTransportTeacher-LocoTouch-v1" : "Isaac-Locomotion-LocoTouch-v1", &c) != 0) { fprintf(stderr, "preset failed\n"); return 2; }
    const bool has_obj = task == 0;
    std::mt19937 rng(1234 + task);
    auto U = [&](float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) / 16777216.f; };
    double worst = 0, worst_state = 0; int worst_i = -1, worst_k = -1; int active[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nstates; ++i) {
      State s;
      const float kdef[3] = {0.1f, 0.9f, -1.8f};
      const int mode = i % 4;  // 0 standing, 1 crouched (knees / calves touch), 2 lying / tilted (hips, trunk corners), 3 airborne
      const float zlo[4] = {0.24f, 0.10f, 0.03f, 0.40f}, zhi[4] = {0.31f, 0.20f, 0.10f, 0.60f};
      s.bp[0] = U(-1, 1); s.bp[1] = U(-1, 1); s.bp[2] = U(zlo[mode], zhi[mode]);
      const float tilt = mode == 2 ? 0.6f : 0.15f;
      const Q4 q = q_from_euler(U(-tilt, tilt), U(-tilt, tilt), U(-3.1f, 3.1f));
      s.bq[0] = q.w; s.bq[1] = q.x; s.bq[2] = q.y; s.bq[3] = q.z;
      for (int k = 0; k < 3; ++k) { s.bu[k] = U(-0.6f, 0.6f); s.bw[k] = U(-1.5f, 1.5f); }
      for (int l = 0; l < 4; ++l) {
        const float sy = (l & 1) ? 1.f : -1.f;
        const float spread = i % 7 == 0 ? 1.15f : (mode == 1 ? 0.9f : 0.35f);  // (every seventh state: calves beyond their limits)
        s.q[l][0] = clampf(sy * kdef[0] + U(-0.3f, 0.3f) * (i % 5 == 0 ? 3.2f : 1.f), -0.92f, 0.92f);  // (every fifth state may sit beyond a joint limit: the implicit limit torques)
        s.q[l][1] = clampf(kdef[1] + U(-spread, spread), -0.75f, 4.56f);
        s.q[l][2] = clampf(kdef[2] + U(-spread, spread), -2.87f, -0.84f);
        for (int k = 0; k < 3; ++k) { s.qd[l][k] = U(-4.f, 4.f); s.tau[l][k] = U(-20.f, 20.f); }
        s.mu[l] = U(0.3f, 1.0f);
      }
      if (mode < 2) {  // put the lowest foot slightly into / above the ground
        const M3 R0 = quat_to_mat(q.w, q.x, q.y, q.z);
        float zmin = 1e9f;
        for (int l = 0; l < 4; ++l) {
          const float sx = l < 2 ? 1.f : -1.f, sy = (l & 1) ? 1.f : -1.f;
          V3 p = v3(sx * 0.1881f, sy * 0.04675f, 0.f);
          const float c0 = std::cos(s.q[l][0]), s0 = std::sin(s.q[l][0]);
          M3 R1 = mul_rot<0>(R0, c0, s0);
          V3 pw = mul(R0, p);
          pw += mul(R1, v3(0.f, sy * 0.08f, 0.f));
          M3 R2 = mul_rot<1>(R1, std::cos(s.q[l][1]), std::sin(s.q[l][1]));
          pw += mul(R2, v3(0.f, 0.f, -0.213f));
          M3 R3 = mul_rot<1>(R2, std::cos(s.q[l][2]), std::sin(s.q[l][2]));
          pw += mul(R3, v3(0.f, 0.f, -0.213f));
          zmin = std::min(zmin, pw.z - 0.02f);
        }
        s.bp[2] = -zmin + U(-0.004f, 0.001f);
      }
      s.orad = U(0.03f, 0.07f); s.olen = U(0.1f, 0.4f); s.omass = U(0.8f, 1.5f); s.omu = U(0.3f, 1.f);
      s.trunk_mass_add = U(-1.f, 2.f); s.trunk_mu = U(0.4f, 1.f);
      {  // cylinder resting on (slightly into / above) the carrying plate, or - every 8th state - lying on the ground
        const Q4 bq = q;
        const V3 d = v3(U(-0.06f, 0.06f), U(-0.04f, 0.04f), LT_BACK_TOP_Z + s.orad + U(-0.002f, 0.002f));
        V3 p = v3(s.bp[0], s.bp[1], s.bp[2]) + qapply(bq, d);
        Q4 oq = qmul(bq, q_from_euler(U(-0.1f, 0.1f), 0.f, U(1.3f, 1.8f)));
        if (i % 8 == 7) { p = v3(s.bp[0] + 0.5f, s.bp[1], s.orad - 0.001f); oq = q_from_euler(0.f, 0.f, U(-3.f, 3.f)); }
        s.op[0] = p.x; s.op[1] = p.y; s.op[2] = p.z; s.oq[0] = oq.w; s.oq[1] = oq.x; s.oq[2] = oq.y; s.oq[3] = oq.z;
        for (int k = 0; k < 3; ++k) { s.ou[k] = s.bu[k] + U(-0.2f, 0.2f); s.ow[k] = U(-1.f, 1.f); }
      }
      Out a, b;
      run_quad<ref::Base, ref::Leg, ref::Obj, ref::Misc, ref::Report>(c, s, nsub, has_obj, a,
        [](const lt_cfg& c, float h, int leg, const float (&sgn)[4], ref::Base& B, ref::Leg& G, ref::Obj& O, const ref::Misc& X, ref::Report& rep, bool ho) {
          if (ho) ref::physics_substep<true, true>(c, h, leg, sgn, B, G, O, X, rep); else ref::physics_substep<false, false>(c, h, leg, sgn, B, G, O, X, rep); });
      if (getenv("TWIN_SELF")) {  // noise floor: the reference formulation against itself on a state perturbed in the last bit
        State s2 = s;
        for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) { s2.q[l][k] = std::nextafter(s.q[l][k], 10.f); s2.qd[l][k] = std::nextafter(s.qd[l][k], 10.f); }
        for (int k = 0; k < 3; ++k) { s2.bp[k] = std::nextafter(s.bp[k], 10.f); s2.bu[k] = std::nextafter(s.bu[k], 10.f); }
        run_quad<ref::Base, ref::Leg, ref::Obj, ref::Misc, ref::Report>(c, s2, nsub, has_obj, b,
          [](const lt_cfg& c, float h, int leg, const float (&sgn)[4], ref::Base& B, ref::Leg& G, ref::Obj& O, const ref::Misc& X, ref::Report& rep, bool ho) {
            if (ho) ref::physics_substep<true, true>(c, h, leg, sgn, B, G, O, X, rep); else ref::physics_substep<false, false>(c, h, leg, sgn, B, G, O, X, rep); });
      } else
      run_quad<neu::Base, neu::Leg, neu::Obj, neu::Misc, neu::Report>(c, s, nsub, has_obj, b,
        [](const lt_cfg& c, float h, int leg, const float (&sgn)[4], neu::Base& B, neu::Leg& G, neu::Obj& O, const neu::Misc& X, neu::Report& rep, bool ho) {
          if (ho) neu::physics_substep<true, true>(c, h, leg, sgn, B, G, O, X, rep); else neu::physics_substep<false, false>(c, h, leg, sgn, B, G, O, X, rep); });
      const int nb = a.n - 15;  // first index of the per-body contact forces
      for (int l = 0; l < 4; ++l) {
        for (int b4 = 0; b4 < 4; ++b4) if (a.v[l][nb + 3 * b4 + 2] != 0.f) active[b4]++;
        if (a.v[l][nb + 14] != 0.f) active[4]++;
        if (has_obj && a.v[l][nb - 6 + 2] != 0.f) active[5]++;
        for (int k = 0; k < a.n; ++k) {
          // state slots: |diff| / (1 + |ref|); contact forces (stiff: k_n ~ 1e4..1e5 N/m turns 1e-7 m into 1e-2 N): relative to 20 N + |ref|
          const bool force = k >= nb - (has_obj ? 6 : 0) && !(has_obj && k >= nb - 3 && k < nb);
          const double d = std::fabs((double)a.v[l][k] - (double)b.v[l][k]) / (force ? 20.0 + std::fabs((double)a.v[l][k]) : 1.0 + std::fabs((double)a.v[l][k]));
          if (!(d <= worst)) { worst = d; worst_i = i; worst_k = k + 100 * l; }
          if (!force && !(d <= worst_state)) worst_state = d;
        }
      }
    }
    printf("%s: %d states x %d substeps: worst scaled |ref - new| = %.3g (state %d, slot %d), state slots alone %.3g; active contacts hip %d thigh %d calf %d foot %d trunk %d obj %d\n",
           has_obj ? "teacher" : "locomotion", nstates, nsub, worst, worst_i, worst_k, worst_state, active[0], active[1], active[2], active[3], active[4], active[5]);
    worst_all = std::max(worst_all, worst);
  }
  if (!(worst_all <= tol)) { printf("FAIL (tolerance %.3g)\n", tol); return 1; }
  printf("OK\n");
  return 0;
}
