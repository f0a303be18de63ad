// Curriculum / population-gate pass of one env step (one workgroup): body of lt_post_kernel (lt_env.hip).
#pragma once

namespace lt {

// block-wide sum of K values at once (one LDS round, two barriers); result replicated in every thread
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float* sh) {
  const int tid = threadIdx.x, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k)
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
  __syncthreads();  // protect sh against the previous use
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < K; ++k) sh[(tid >> 6) * K + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += sh[w * K + k];
    v[k] = t;
  }
}
__device__ __forceinline__ void set_range(float* P, int d, float lo, float hi) {
  P[6 + 2 * d] = P[2 * d]; P[6 + 2 * d + 1] = P[2 * d + 1];
  P[2 * d] = lo; P[2 * d + 1] = hi;
  P[12 + d] = (P[6 + 2 * d] == P[2 * d] && P[6 + 2 * d + 1] == P[2 * d + 1]) ? 1.f : 0.f;
}
// Every thread keeps the per-env trackers of its envs in registers for the whole kernel (<= 8 envs per thread at
// N = 8192; larger N strides further) and replays the reference's call order: lin gate -> maybe widen -> ang gate.
// One workgroup (any multiple of 64 threads up to 1024); `sh`: 128 floats of LDS.
// Every thread keeps its envs' records and trackers in registers from ONE batch of loads to the final write-back, so the
// pass costs two memory round trips (command block, env rows) however many reduction phases the curriculum has.  Rounds of E envs per thread
// cover any N; the cross-phase decisions are global, so with more than one round the phases re-read instead (kept simple:
// rounds > 1 only happens for N > E * threads; E = 4 in the 1024-thread kernel).
template <int E>
__device__ __forceinline__ void post_body(const lt_dev_args* __restrict__ d, char* const arena, int bump_counter, int gates_only, float* sh) {
  const lt_cfg& c = d->cfg;
  const lt_layout& L = d->layout;
  float* P = (float*)(arena + L.off_cmd_params);
  const long long n = L.n, q4 = L.npad * 4;
  const float4* rec = (const float4*)(arena + L.quad_off[LT_F_CURRICULUM]);
  float4* trk1 = (float4*)((float*)(arena + L.quad_off[LT_F_CURRICULUM]) + q4);
  float4* trk2 = (float4*)((float*)(arena + L.quad_off[LT_F_CURRICULUM]) + 2 * q4);
  const float4* cmd = (const float4*)(arena + L.quad_off[LT_F_CMD]);
  const int tid = threadIdx.x, nt = blockDim.x;
  const bool one_round = n <= (long long)E * nt;
  // env rows of this thread (round 0), all loads in flight together with the command block
  float4 cc[E], rc[E], t1[E];
  float2 t2[E];  // only .x/.y of the second tracker row are ever touched
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const long long e = tid + (long long)k * nt;
    const long long ec = e < n ? e : n - 1;  // clamped, unconditional: a predicated load would get a wait of its own
    cc[k] = cmd[ec]; rc[k] = rec[ec]; t1[k] = trk1[ec]; t2[k] = *(const float2*)(trk2 + ec);
  }
  // command block snapshot (uniform; thread 0 is the only writer, at the very end)
  float Pl[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) Pl[i] = P[i];
  const float* mx = c.cmd_range_max;
  const bool cur = c.cur_enabled != 0 && !gates_only;
  const bool lin_open = cur && (Pl[1] != mx[0] || Pl[12] == 0.f || Pl[3] != mx[1] || Pl[13] == 0.f) && (Pl[17] - Pl[18] <= (float)c.cur_max_distance_bins);
  // pass 1: population gate, any-reset flag, lin statistics with this step's records merged in
  float r1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // nz, any, not-all-reset(lin), sum len, sum reward
#pragma unroll
  for (int k = 0; k < E; ++k) {
    if (tid + (long long)k * nt >= n) continue;
    if (cc[k].x != 0.f || cc[k].y != 0.f || cc[k].z != 0.f) r1[0] = 1.f;
    if (rc[k].x != 0.f) r1[1] = 1.f;
    if (lin_open) {
      float4 t = t1[k];
      if (rc[k].x != 0.f) { t.x = 1.f; t.y = rc[k].y; t.z = rc[k].z; }
      if (t.x == 0.f) r1[2] = 1.f;
      r1[3] += t.y; r1[4] += t.z;
    }
  }
  if (!one_round)
    for (long long e = tid + (long long)E * nt; e < n; e += nt) {
      const float4 c4 = cmd[e], r4 = rec[e];
      if (c4.x != 0.f || c4.y != 0.f || c4.z != 0.f) r1[0] = 1.f;
      if (r4.x != 0.f) r1[1] = 1.f;
      if (lin_open) {
        float4 t = trk1[e];
        if (r4.x != 0.f) { t.x = 1.f; t.y = r4.y; t.z = r4.z; }
        if (t.x == 0.f) r1[2] = 1.f;
        r1[3] += t.y; r1[4] += t.z;
      }
    }
  block_sum<5>(r1, sh);
  const bool any = r1[1] > 0.f;
  const float inv_n = 1.f / (float)n;
  const bool run = cur && any;  // _reset_idx (and the curriculum with it) only runs when some env reset this step
  bool lin_pass = false;
  if (run && lin_open) {
    lin_pass = r1[2] == 0.f && r1[3] * inv_n > c.cur_len_threshold && r1[4] * inv_n > c.cur_reward_threshold[0];
    if (lin_pass) {
      Pl[19] += 1.f;
      if ((int)Pl[19] == c.cur_repeat_times[0]) {
        const float lx = clampf(Pl[0] - Pl[21], -mx[0], 0.f), ly = clampf(Pl[2] - Pl[22], -mx[1], 0.f);
        set_range(Pl, 0, lx, -lx);
        set_range(Pl, 1, ly, -ly);
        if (Pl[12] != 0.f && Pl[13] != 0.f && Pl[14] != 0.f) { Pl[15] = (float)c.cmd_zero_steps_final; Pl[16] = c.cmd_rel_standing_final; }
        Pl[19] = 0.f; Pl[17] += 1.f;
      }
    }
  }
  const bool ang_open = run && (Pl[5] != mx[2] || Pl[14] == 0.f) && (Pl[18] - Pl[17] <= (float)c.cur_max_distance_bins);
  // pass 2: ang statistics (gate evaluated after the lin update, as in the reference's call order) + tracker update
  float r2[3] = {0.f, 0.f, 0.f};
  if (run) {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (tid + (long long)k * nt >= n) continue;
      if (lin_open) {
        if (rc[k].x != 0.f) { t1[k].x = 1.f; t1[k].y = rc[k].y; t1[k].z = rc[k].z; }
        if (lin_pass) { t1[k].x = 0.f; t1[k].y = 0.f; t1[k].z = 0.f; }
      }
      if (ang_open) {
        if (rc[k].x != 0.f) { t1[k].w = 1.f; t2[k].x = rc[k].y; t2[k].y = rc[k].w; }
        if (t1[k].w == 0.f) r2[0] = 1.f;
        r2[1] += t2[k].x; r2[2] += t2[k].y;
      }
    }
    if (!one_round)
      for (long long e = tid + (long long)E * nt; e < n; e += nt) {
        const float4 r4 = rec[e];
        float4 a1 = trk1[e], a2 = trk2[e];
        if (lin_open) {
          if (r4.x != 0.f) { a1.x = 1.f; a1.y = r4.y; a1.z = r4.z; }
          if (lin_pass) { a1.x = 0.f; a1.y = 0.f; a1.z = 0.f; }
        }
        if (ang_open) {
          if (r4.x != 0.f) { a1.w = 1.f; a2.x = r4.y; a2.y = r4.w; }
          if (a1.w == 0.f) r2[0] = 1.f;
          r2[1] += a2.x; r2[2] += a2.y;
        }
        trk1[e] = a1; trk2[e] = a2;
      }
    bool ang_pass = false;
    if (ang_open) {
      block_sum<3>(r2, sh);
      ang_pass = r2[0] == 0.f && r2[1] * inv_n > c.cur_len_threshold && r2[2] * inv_n > c.cur_reward_threshold[1];
      if (ang_pass) {
        Pl[20] += 1.f;
        if ((int)Pl[20] == c.cur_repeat_times[1]) {
          const float lz = clampf(Pl[4] - Pl[23], -mx[2], 0.f);
          set_range(Pl, 2, lz, -lz);
          if (Pl[12] != 0.f && Pl[13] != 0.f && Pl[14] != 0.f) { Pl[15] = (float)c.cmd_zero_steps_final; Pl[16] = c.cmd_rel_standing_final; }
          Pl[20] = 0.f; Pl[18] += 1.f;
        }
      }
    }
    // write-back (each thread only ever touches its own rows)
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const long long e = tid + (long long)k * nt;
      if (e >= n) continue;
      if (ang_pass) { t1[k].w = 0.f; t2[k].x = 0.f; t2[k].y = 0.f; }
      trk1[e] = t1[k]; *(float2*)(trk2 + e) = t2[k];
    }
    if (!one_round && ang_pass)
      for (long long e = tid + (long long)E * nt; e < n; e += nt) {
        float4 a1 = trk1[e];
        a1.w = 0.f;
        trk1[e] = a1;
        trk2[e] = make_float4(0.f, 0.f, trk2[e].z, trk2[e].w);
      }
    Pl[24] = lin_open ? 1.f : 0.f; Pl[25] = ang_open ? 1.f : 0.f;
  }
  Pl[26] = r1[0] > 0.f ? 1.f : 0.f;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 27; ++i) P[i] = Pl[i];
    if (bump_counter) ((long long*)(arena + L.off_counters))[0] += 1;
  }
}

}  // namespace lt
