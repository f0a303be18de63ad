// Curriculum / population-gate pass of one env step: per-wave partial sums at the tail of the step kernel (lt_env.hip), the global
// decision in a one-wave kernel behind it (lt_gate_decide_kernel).
//
// Reference: ModifyVelCommandsRangeBasedonReward (locotouch/mdp/curriculums.py:184-275) + MultiSampling.set_ranges
// (locotouch/mdp/commands.py:471-505), the population gate of rewards.py:190 and `common_step_counter += 1`.
//
// The pass needs sums over ALL envs and then a global decision:
//   1. every wave (16 envs) of the step kernel first applies the tracker operations the PREVIOUS pass decided
//      (LT_F_CMD_PARAMS[27..30]) to its own envs' trackers, then forms its partial sums with this step's records merged in
//      hypothetically (registers only) and stores them - 8 floats - into its slot (curriculum_publish);
//   2. lt_gate_decide_kernel (one wave, next on the stream - or, in the rollout graph, on a side branch beside the next policy
//      launch: nothing but the next STEP kernel reads what it writes) reduces the slots in a fixed order (deterministic, run-to-run
//      identical), replays the reference's decision sequence (lin gate -> maybe widen -> ang gate -> maybe widen), writes the
//      command block + the tracker operations for the next pass and bumps the step counter (curriculum_decide).
// Until round 2 step 2 ran inside the step kernel, in the wave that took the last of an agent-scope ticket: the write-through
// stores, the drain, the ticket round trip and the last arriver's dependent loads put 18 k cycles (7.6 us) behind the physics of
// the last tile (tools/stamp_probe.py) for 1.5 us of boundary saved.  No wave ever touches another wave's envs, so the per-env
// trackers need no cross-workgroup visibility at all; they simply lag the reference's by one pass (the oracle keeps the same
// representation; the decisions are the reference's own).
#pragma once

namespace lt {

__device__ __forceinline__ void set_range(float* P, int d, float lo, float hi) {
  P[6 + 2 * d] = P[2 * d]; P[6 + 2 * d + 1] = P[2 * d + 1];
  P[2 * d] = lo; P[2 * d + 1] = hi;
  P[12 + d] = (P[6 + 2 * d] == P[2 * d] && P[6 + 2 * d + 1] == P[2 * d + 1]) ? 1.f : 0.f;
}
// wave-wide butterflies (fixed order: deterministic); every lane ends with the total
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
  return v;
}

// This step's record of the lane's env (replicated inside the quad) + what the population gate needs.
struct CurIn {
  bool valid;        // env < n (padded tail envs take no part)
  bool reset;        // the env reset this step
  float ep_len;      // episode length / episode sums of the two tracking rewards at the reset
  float sum_lin, sum_ang;
  bool cmd_nonzero;  // the env's command after this step's command update
};

// Step 1.  One block of the launch = one wave64 with lane = env * 4 + leg (the step kernel's mapping); every 16-env tile of the
// grid calls this exactly once per pass.
// `P`: the command block as this tile may read it (global memory, or the workgroup's LDS copy in the step kernel's helper form).
// `set`: which of the two slot sets (the step's parity: the pass of step t may still be reading set t & 1 inside the launch of step
// t + 1 - chained mode - while that launch's tiles publish into the other one).
__device__ __forceinline__ void curriculum_publish(const lt_layout& L, char* const arena, const float* const P, long long gid, int leg, const CurIn& in, int set) {
  const long long q4 = L.npad * 4;
  float* const rec_p = (float*)(arena + L.quad_off[LT_F_CURRICULUM]) + gid;
  float* const t1_p = rec_p + q4;
  float* const t2_p = rec_p + 2 * q4;
  const float recp = *rec_p;
  float t1 = *t1_p, t2 = *t2_p;
  // ---- 1. tracker operations decided by the previous pass, on the record that pass saw ----
  {
    const bool had = qbcast<0>(recp) != 0.f;
    const float rp_len = qbcast<1>(recp), rp_lin = qbcast<2>(recp), rp_ang = qbcast<3>(recp);
    const bool mlin = P[27] != 0.f && had, mang = P[28] != 0.f && had, clin = P[29] != 0.f, cang = P[30] != 0.f;
    if (leg < 3) {  // (reset_lin, len_lin, sum_lin)
      if (mlin) t1 = sel4(leg, 1.f, rp_len, rp_lin, 0.f);
      if (clin) t1 = 0.f;
    } else {        // reset_ang
      if (mang) t1 = 1.f;
      if (cang) t1 = 0.f;
    }
    if (leg < 2) {  // (len_ang, sum_ang)
      if (mang) t2 = leg == 0 ? rp_len : rp_ang;
      if (cang) t2 = 0.f;
    }
    *t1_p = t1; *t2_p = t2;
    *rec_p = in.reset ? sel4(leg, 1.f, in.ep_len, in.sum_lin, in.sum_ang) : 0.f;
  }
  // ---- 2. partial sums with this step's record merged in (curriculums.py:221-224, 241-244) ----
  const float f_lin = in.reset ? 1.f : qbcast<0>(t1), l_lin = in.reset ? in.ep_len : qbcast<1>(t1), s_lin = in.reset ? in.sum_lin : qbcast<2>(t1);
  const float f_ang = in.reset ? 1.f : qbcast<3>(t1), l_ang = in.reset ? in.ep_len : qbcast<0>(t2), s_ang = in.reset ? in.sum_ang : qbcast<1>(t2);
  const bool mine = in.valid && leg == 0;
  int flags = mine ? ((in.cmd_nonzero ? 1 : 0) | (in.reset ? 2 : 0) | (f_lin == 0.f ? 4 : 0) | (f_ang == 0.f ? 8 : 0)) : 0;
  flags = wave_or(flags);
  const float w_llin = wave_sum(mine ? l_lin : 0.f), w_slin = wave_sum(mine ? s_lin : 0.f);
  const float w_lang = wave_sum(mine ? l_ang : 0.f), w_sang = wave_sum(mine ? s_ang : 0.f);
  // ---- 3. publish: this tile's slot (plain stores; the decision kernel starts behind a kernel boundary) ----
  const int lane = threadIdx.x & 63;
  float* const slots = (float*)(arena + L.off_partials) + (long long)set * (L.npad / 16) * LT_PARTIAL_FLOATS;
  {
    const float bit[4] = {(float)(flags & 1), (float)((flags >> 1) & 1), (float)((flags >> 2) & 1), (float)((flags >> 3) & 1)};
    const float v = lane == 0 ? bit[0] : lane == 1 ? bit[1] : lane == 2 ? bit[2] : lane == 3 ? w_llin : lane == 4 ? w_slin
                    : lane == 5 ? bit[3] : lane == 6 ? w_lang : w_sang;
    if (lane < LT_PARTIAL_FLOATS) slots[(long long)blockIdx.x * LT_PARTIAL_FLOATS + lane] = v;
  }
}

// The reference's decision sequence on one pass's population sums `r` (lin gate -> maybe widen -> ang gate -> maybe widen);
// updates the command block copy `Pl` in place.  `allow_*`: whether a success may be declared for that group here.
struct GateOut { bool run, lin_open, lin_pass, ang_open, ang_pass; };
__device__ __forceinline__ GateOut gate_decision(const lt_cfg& c, float (&Pl)[31], const float (&r)[LT_PARTIAL_FLOATS], float inv_n,
                                                 bool allow_lin, bool allow_ang) {
  GateOut g;
  const float* mx = c.cmd_range_max;
  g.run = c.cur_enabled != 0 && r[1] > 0.f;  // _reset_idx (and the curriculum with it) only runs when some env reset this step
  g.lin_open = g.run && (Pl[1] != mx[0] || Pl[12] == 0.f || Pl[3] != mx[1] || Pl[13] == 0.f) &&
               (Pl[17] - Pl[18] <= (float)c.cur_max_distance_bins);                                        // curriculums.py:218-220
  g.lin_pass = false;
  if (g.lin_open && allow_lin) {
    g.lin_pass = r[2] == 0.f && r[3] * inv_n > c.cur_len_threshold && r[4] * inv_n > c.cur_reward_threshold[0];  // :224
    if (g.lin_pass) {
      Pl[19] += 1.f;
      if ((int)Pl[19] == c.cur_repeat_times[0]) {                                                          // :226-235
        const float lx = clampf(Pl[0] - Pl[21], -mx[0], 0.f), ly = clampf(Pl[2] - Pl[22], -mx[1], 0.f);
        set_range(Pl, 0, lx, -lx);
        set_range(Pl, 1, ly, -ly);
        if (Pl[12] != 0.f && Pl[13] != 0.f && Pl[14] != 0.f) { Pl[15] = (float)c.cmd_zero_steps_final; Pl[16] = c.cmd_rel_standing_final; }
        Pl[19] = 0.f; Pl[17] += 1.f;
      }
    }
  }
  // the ang gate is evaluated after the lin update, as in the reference's call order (:239-240)
  g.ang_open = g.run && (Pl[5] != mx[2] || Pl[14] == 0.f) && (Pl[18] - Pl[17] <= (float)c.cur_max_distance_bins);
  g.ang_pass = false;
  if (g.ang_open && allow_ang) {
    g.ang_pass = r[5] == 0.f && r[6] * inv_n > c.cur_len_threshold && r[7] * inv_n > c.cur_reward_threshold[1];
    if (g.ang_pass) {
      Pl[20] += 1.f;
      if ((int)Pl[20] == c.cur_repeat_times[1]) {
        const float lz = clampf(Pl[4] - Pl[23], -mx[2], 0.f);
        set_range(Pl, 2, lz, -lz);
        if (Pl[12] != 0.f && Pl[13] != 0.f && Pl[14] != 0.f) { Pl[15] = (float)c.cmd_zero_steps_final; Pl[16] = c.cmd_rel_standing_final; }
        Pl[20] = 0.f; Pl[18] += 1.f;
      }
    }
  }
  return g;
}

// Multi-rank gate (cfg.cur_gate_external), one wave: the decision sequence replayed over the last `nsteps` passes on sums
// all-reduced over the ranks.  After a group's first success the later rows of this window are stale for it (their sums
// were formed from trackers the success clears), so the group is closed for the rest of the window and the clear is
// scheduled (P[29] / P[30]) for the next step's tracker pass.
__device__ __forceinline__ void curriculum_apply_global(const lt_cfg& c, const lt_layout& L, char* const arena, const float* ring_sums,
                                                        int nsteps, float inv_n_total) {
  float* const P = (float*)(arena + L.off_cmd_params);
  const long long pos = ((const long long*)(arena + L.off_counters))[3];
  float Pl[31];
#pragma unroll
  for (int i = 0; i < 31; ++i) Pl[i] = P[i];
  bool lin_done = false, ang_done = false;
  for (int k = 0; k < nsteps; ++k) {
    const long long pass = pos - nsteps + k;
    if (pass < 0) continue;
    const float* row = ring_sums + (pass % LT_GATE_RING) * LT_PARTIAL_FLOATS;
    float r[LT_PARTIAL_FLOATS];
#pragma unroll
    for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) r[i] = row[i];
    const GateOut g = gate_decision(c, Pl, r, inv_n_total, !lin_done, !ang_done);
    lin_done |= g.lin_pass;
    ang_done |= g.ang_pass;
  }
  if (lin_done) Pl[29] = 1.f;
  if (ang_done) Pl[30] = 1.f;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 31; ++i) P[i] = Pl[i];
  }
}

// Step 2 (one wave): fixed-order reduction of the slots, then the reference's decision sequence.
// `bump_counter`: added to common_step_counter (the steps whose pass this call completes).
// `chain_flag` != 0 - the chained form: the pass of step t runs inside the launch of step t + 1 (lt_step_kernel, workgroup 0's RNG
// wave, beside the first physics substep; nothing of a step reads the command block before its physics is over).  The command block
// is then stored write-through and drained, and counters[2] = chain_flag (the launch's step id) tells the other workgroups of that
// launch that it is final (MI355X_MICROARCH.md, "Valid forms": sc1 payload -> vmcnt(0) -> flag; consumers poll and load with sc1).
// `set` < 0: the slot set of the last step launched, (common_step_counter + bump_counter - 1) & 1.
// Lane-strided partial sums of the tiles' slots (8 floats = two float4 each): thread `t` of `nt` takes slots t, t + nt, ...
// EIGHT slots (16 loads of 16 B) are requested per trip before anything is added - at 32768 envs (2048 slots) one wave walked its
// 32 slots as 32 x 8 dependent dword round trips: 12.8 us per step behind every step launch (profiles/r03_kernel_stats_32768_f32.csv).
// Out-of-range slots re-read the last one and are masked by a select, never by a branch (a branch between a load and its wait makes
// the compiler wait for every outstanding load).  The order of the additions is fixed by (nt, nwaves): run-to-run identical.
// Both forms of the pass add in the SAME order - thread t of 256 walks slots t, t + 256, ... in increasing order, the four partials of a
// lane column are added ((p0 + p1) + p2) + p3, then the 64-lane butterfly - so the chained pass (one wave standing in for the four)
// and the pass behind every step leave bit-identical sums (tests/test_hip_parity.py::test_chained_population_pass_...).
template <int U>
__device__ __forceinline__ void slot_sums(const float* slots, unsigned nwaves, unsigned t, unsigned nt, float (&r)[LT_PARTIAL_FLOATS]) {
#pragma unroll
  for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) r[i] = 0.f;
  const float4* const s4 = (const float4*)slots;
  for (unsigned base = t; base < nwaves; base += U * nt) {
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned w = base + u * nt, wc = w < nwaves ? w : nwaves - 1u;
      a[u] = s4[2ull * wc];
      b[u] = s4[2ull * wc + 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = base + u * nt < nwaves;
      r[0] += ok ? a[u].x : 0.f; r[1] += ok ? a[u].y : 0.f; r[2] += ok ? a[u].z : 0.f; r[3] += ok ? a[u].w : 0.f;
      r[4] += ok ? b[u].x : 0.f; r[5] += ok ? b[u].y : 0.f; r[6] += ok ? b[u].z : 0.f; r[7] += ok ? b[u].w : 0.f;
    }
  }
}

// `pre`: this lane's column sum ((p0 + p1) + p2) + p3 made by the caller's four waves (lt_gate_decide_kernel), or nullptr = this wave
// stands in for the four itself.
__device__ __forceinline__ void curriculum_decide(const lt_cfg& c, const lt_layout& L, char* const arena, int bump_counter, long long chain_flag = 0, int set = -1,
                                                  const float* pre = nullptr) {
  float* const P = (float*)(arena + L.off_cmd_params);
  const int lane = threadIdx.x & 63;
  const unsigned nwaves = (unsigned)(L.npad / 16);
  long long* const cnt = (long long*)(arena + L.off_counters);
  if (set < 0) set = (int)((cnt[0] + (bump_counter > 0 ? bump_counter - 1 : 0)) & 1);
  const float* const slots = (const float*)(arena + L.off_partials) + (long long)set * (L.npad / 16) * LT_PARTIAL_FLOATS;
  // everything this wave reads is requested up front: one memory round trip, not four dependent ones
  float Pl[31];
#pragma unroll
  for (int i = 0; i < 31; ++i) Pl[i] = P[i];
  const long long cnt0 = cnt[0], cnt3 = cnt[3];
  float r[LT_PARTIAL_FLOATS];
  if (pre) {
#pragma unroll
    for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) r[i] = pre[i];
  } else {
    float p[4][LT_PARTIAL_FLOATS];
#pragma unroll
    for (int w = 0; w < 4; ++w) slot_sums<1>(slots, nwaves, (unsigned)lane + 64u * w, 256u, p[w]);
#pragma unroll
    for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) r[i] = ((p[0][i] + p[1][i]) + p[2][i]) + p[3][i];
  }
#pragma unroll
  for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) r[i] = wave_sum(r[i]);
  const bool ext = c.cur_gate_external != 0;  // multi-rank: the success test runs on cross-rank sums (curriculum_apply_global)
  GateOut g = gate_decision(c, Pl, r, 1.f / (float)L.n, !ext, !ext);
  if (g.run) { Pl[24] = g.lin_open ? 1.f : 0.f; Pl[25] = g.ang_open ? 1.f : 0.f; }
  Pl[26] = r[0] > 0.f ? 1.f : 0.f;
  Pl[27] = g.lin_open ? 1.f : 0.f; Pl[28] = g.ang_open ? 1.f : 0.f;  // tracker operations for the next pass
  Pl[29] = g.lin_pass ? 1.f : 0.f; Pl[30] = g.ang_pass ? 1.f : 0.f;
  if (lane == 0) {  // this pass's population sums, for a cross-rank gate
    float* const ring = (float*)(arena + L.off_gate_ring) + (cnt3 % LT_GATE_RING) * LT_PARTIAL_FLOATS;
#pragma unroll
    for (int i = 0; i < LT_PARTIAL_FLOATS; ++i) ring[i] = r[i];
    cnt[3] = cnt3 + 1;
    if (chain_flag) {
#pragma unroll
      for (int i = 0; i < 31; ++i) __hip_atomic_store(P + i, Pl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(cnt + 2, chain_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
      for (int i = 0; i < 31; ++i) P[i] = Pl[i];
    }
    if (bump_counter) cnt[0] = cnt0 + bump_counter;
  }
}

}  // namespace lt
