// Single-layer GRU over a padded batch of whole trajectories: the recurrence of the student's tactile encoder
// (reference loco_rl/loco_rl/models/memory_module.py:10-14 -> nn.GRU; shapes of BASELINE.json configs[3]: input 64, hidden 512,
// L ~ 500 steps, B ~ 50-100 trajectories).
//
// The recurrence is L dependent steps of one small GEMM ([B, H] x [H, 3H]) and a gate formula.  Issued as library calls a step
// costs ~36 us of GPU time forward and as much backward on this stack (hipBLASLt puts a 101-row GEMM on ~24 workgroups);
// here a step is ONE launch whose grid covers the chip: a workgroup owns a 16 x 16 (hidden unit x batch row) output tile,
// its four waves split the reduction (K = H forward, 3H backward) and meet in LDS, and the gate arithmetic is the epilogue.
// The time loop runs on the host side of the C ABI (lt_gru_forward / lt_gru_backward): no Python between steps.
// Arithmetic: v_mfma_f32_16x16x4_f32, exact f32 products, f32 accumulation (sum order differs from a library GEMM's).
//
// Operand trick: a 16 x 16 x 4 MFMA wants lane (i = l % 16, q = l / 16) to supply A[i][k0 + q] and B[k0 + q][n = i].  Summation
// over k is order-free, so MFMA step s of a 16-wide k block consumes the k-set {kb + 4 q + s}: lane (i, q) then supplies
// component s of ONE float4 load A[i][kb + 4q .. + 3] - 16-byte loads, four MFMAs per load.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lt_env.h"
#include "lt_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { const float e = __expf(-2.f * fabsf(x)); const float t = (1.f - e) / (1.f + e); return x < 0.f ? -t : t; }

// ---- forward step: h' = GRU(ig_t, h) for a 16-unit x 16-row tile; ws = (r, z, n, q = W_hn h + b_hn) ----------------------
// grid (H / 16, ceil(B / 16)), block 256 (4 waves, wave w reduces k in [w * H / 4, (w + 1) * H / 4))
// KB > 0: H = 64 KB known at compile time - the wave's KB k-blocks are fully unrolled so that all 4 KB operand loads (16 bytes each)
// are in flight before the first MFMA; with a runtime trip count every k-block exposed an L2 round trip (9.0 us per step at
// H = 512 against ~2 us of loads + MFMAs).  KB == 0: any H that is a multiple of 64.
template <int KB>
__global__ __launch_bounds__(256) void lt_gru_step_fwd(const float* __restrict__ ig, const float* __restrict__ h, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ h_out,
                                                       float* __restrict__ ws, int B, int H_rt) {
  const int H = KB > 0 ? 64 * KB : H_rt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int row = b0 + i;                       // batch row this lane feeds as the B operand
  const bool row_ok = row < B;
  const float* hrow = h + (long long)(row_ok ? row : 0) * H;
  const float* wr = w_hh + (long long)(j0 + i) * H;  // A operand rows: unit j0 + i of gate r; + H*H, + 2*H*H for z, n
  const long long gate = (long long)H * H;
  f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;
  // epilogue lanes (wave 0): lane (n, g) owns units j0 + 4 g .. + 3 of batch row b0 + n.  Their operands (this step's input-gate
  // pre-activations - first touch, HBM -, biases, h) are requested NOW, so the round trip overlaps the GEMM instead of following it.
  const int be = b0 + (lane & 15), je = j0 + 4 * (lane >> 4);
  const bool ep = wave == 0 && be < B;
  f32x4 e_ig[3], e_bi[3], e_bh[3], e_hp;
  if (ep) {
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) {
      e_ig[gt] = *(const f32x4*)(ig + (long long)be * 3 * H + gt * H + je);
      e_bi[gt] = *(const f32x4*)(b_ih + gt * H + je);
      e_bh[gt] = *(const f32x4*)(b_hh + gt * H + je);
    }
    e_hp = *(const f32x4*)(h + (long long)be * H + je);
  }
  const int kq = H / 4, k_begin = wave * kq, k_end = k_begin + kq;
  if (KB > 0) {
    f32x4 hv[KB > 0 ? KB : 1], wrv[KB > 0 ? KB : 1], wzv[KB > 0 ? KB : 1], wnv[KB > 0 ? KB : 1];
#pragma unroll
    for (int it = 0; it < KB; ++it) {
      const int k = k_begin + 16 * it + 4 * q;
      hv[it] = *(const f32x4*)(hrow + k);
      wrv[it] = *(const f32x4*)(wr + k); wzv[it] = *(const f32x4*)(wr + gate + k); wnv[it] = *(const f32x4*)(wr + 2 * gate + k);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep every load above the first MFMA (the scheduler would re-serialise them to save registers)
#pragma unroll
    for (int it = 0; it < KB; ++it) {
      if (!row_ok) hv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(wrv[it][s], hv[it][s], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(wzv[it][s], hv[it][s], acc_z, 0, 0, 0);
        acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(wnv[it][s], hv[it][s], acc_n, 0, 0, 0);
      }
    }
  } else {
    for (int kb = k_begin; kb < k_end; kb += 16) {
      const int k = kb + 4 * q;
      f32x4 hv = *(const f32x4*)(hrow + k);
      if (!row_ok) hv = (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 wrv = *(const f32x4*)(wr + k), wzv = *(const f32x4*)(wr + gate + k), wnv = *(const f32x4*)(wr + 2 * gate + k);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(wrv[s], hv[s], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(wzv[s], hv[s], acc_z, 0, 0, 0);
        acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(wnv[s], hv[s], acc_n, 0, 0, 0);
      }
    }
  }
  // D layout: acc[v] of lane (n = l % 16, g = l / 16) is D[unit 4 g + v][row n]
  __shared__ float red[3][4][4][64];  // [gate][wave][v][lane]
#pragma unroll
  for (int v = 0; v < 4; ++v) { red[0][wave][v][lane] = acc_r[v]; red[1][wave][v][lane] = acc_z[v]; red[2][wave][v][lane] = acc_n[v]; }
  __syncthreads();
  if (!ep) return;
  f32x4 o_h, o_r, o_z, o_n, o_q;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float sr = red[0][0][v][lane] + red[0][1][v][lane] + red[0][2][v][lane] + red[0][3][v][lane];
    const float sz = red[1][0][v][lane] + red[1][1][v][lane] + red[1][2][v][lane] + red[1][3][v][lane];
    const float sn = red[2][0][v][lane] + red[2][1][v][lane] + red[2][2][v][lane] + red[2][3][v][lane];
    const float r = sigmoidf_(e_ig[0][v] + e_bi[0][v] + sr + e_bh[0][v]);
    const float z = sigmoidf_(e_ig[1][v] + e_bi[1][v] + sz + e_bh[1][v]);
    const float qn = sn + e_bh[2][v];
    const float nn = tanhf_(e_ig[2][v] + e_bi[2][v] + r * qn);
    o_h[v] = nn + z * (e_hp[v] - nn);
    o_r[v] = r; o_z[v] = z; o_n[v] = nn; o_q[v] = qn;
  }
  *(f32x4*)(h_out + (long long)be * H + je) = o_h;
  float* w = ws + (long long)be * 4 * H + je;
  *(f32x4*)w = o_r; *(f32x4*)(w + H) = o_z; *(f32x4*)(w + 2 * H) = o_n; *(f32x4*)(w + 3 * H) = o_q;
}

// ---- backward step, pointwise part: gate gradients from dh' = dout_t + dh_next -------------------------------------------
// grid ceil(B * H / 256); writes dig_t, dhg_t [B, 3H] and dh_direct [B, H] = dh' * z
__global__ __launch_bounds__(256) void lt_gru_step_bwd_gates(const float* __restrict__ dout, const float* __restrict__ dh_next,
                                                             const float* __restrict__ ws, const float* __restrict__ h_prev,
                                                             float* __restrict__ dig, float* __restrict__ dhg, float* __restrict__ dh_direct,
                                                             int B, int H) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * H) return;
  const int b = (int)(idx / H), j = (int)(idx - (long long)b * H);
  const float* w = ws + (long long)b * 4 * H;
  const float r = w[j], z = w[H + j], n = w[2 * H + j], qn = w[3 * H + j];
  const float dh = dout[idx] + (dh_next ? dh_next[idx] : 0.f);
  const float dn = dh * (1.f - z), dz = dh * (h_prev[idx] - n);
  const float dan = dn * (1.f - n * n), daz = dz * z * (1.f - z);
  const float dar = dan * qn * r * (1.f - r);
  float* gi = dig + (long long)b * 3 * H;
  float* gh = dhg + (long long)b * 3 * H;
  gi[j] = dar; gi[H + j] = daz; gi[2 * H + j] = dan;
  gh[j] = dar; gh[H + j] = daz; gh[2 * H + j] = dan * r;
  dh_direct[idx] = dh * z;
}

// ---- backward step, recurrent part + the NEXT step's pointwise part --------------------------------------------------------------
// dh_prev = dh_direct + dhg_t W_hh for a 16-k x 16-row tile, then - same lanes, same (row, unit) elements - the gate gradients of
// step t - 1 from dh' = dout_{t-1} + dh_prev (what lt_gru_step_bwd_gates computes; that kernel now only opens the recursion at
// t = L - 1): one launch per backward step instead of two.  `direct` is read and rewritten element-wise by its owning lane.
// grid (H / 16, ceil(B / 16)), block 256 (wave w reduces j in [w * 3H / 4, (w + 1) * 3H / 4)).  t == 0: writes dh0 instead.
template <int KB>  // as lt_gru_step_fwd: KB > 0 -> H = 64 KB, the wave's 3 KB j-blocks unrolled in groups of 8 with their loads issued first
__global__ __launch_bounds__(256) void lt_gru_step_bwd_fused(const float* __restrict__ dhg_t, const float* __restrict__ w_hh, float* __restrict__ direct,
                                                             const float* __restrict__ dout_prev, const float* __restrict__ ws_prev,
                                                             const float* __restrict__ h_prev2, float* __restrict__ dig_prev,
                                                             float* __restrict__ dhg_prev, float* __restrict__ dh0, int B, int H_rt) {
  const int H = KB > 0 ? 64 * KB : H_rt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int row = b0 + i;
  const bool row_ok = row < B;
  const float* grow = dhg_t + (long long)(row_ok ? row : 0) * 3 * H;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // epilogue operands of wave 0's lanes (row b0 + n, units k0 + 4 g .. + 3), requested before the GEMM (see lt_gru_step_fwd)
  const int be = b0 + (lane & 15), je = k0 + 4 * (lane >> 4);
  const bool ep = wave == 0 && be < B;
  f32x4 e_direct, e_dout, e_hp2, e_w[4];
  if (ep) {
    const long long o = (long long)be * H + je;
    e_direct = *(const f32x4*)(direct + o);
    if (dout_prev) {
      e_dout = *(const f32x4*)(dout_prev + o);
      e_hp2 = *(const f32x4*)(h_prev2 + o);
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) e_w[gt] = *(const f32x4*)(ws_prev + (long long)be * 4 * H + gt * H + je);
    }
  }
  const int jq = 3 * H / 4, j_begin = wave * jq, j_end = j_begin + jq;
  // A operand: A[out = k0 + i][reduction index j + s] = W[j + s][k0 + i]  (16 consecutive k across the lanes of one q: coalesced)
  if (KB > 0) {
    constexpr int G = 8;  // j-blocks per group: 8 x (4 + 4) operand registers in flight
#pragma unroll
    for (int it0 = 0; it0 < 3 * KB; it0 += G) {
      f32x4 gv[G], wv[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (it0 + u < 3 * KB) {
          const int j = j_begin + 16 * (it0 + u) + 4 * q;
          gv[u] = *(const f32x4*)(grow + j);
          const float* wp = w_hh + (long long)j * H + k0 + i;
          wv[u] = (f32x4){wp[0], wp[H], wp[2 * (long long)H], wp[3 * (long long)H]};
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // loads of the group first, then its MFMAs
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (it0 + u < 3 * KB) {
          if (!row_ok) gv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][s], gv[u][s], acc, 0, 0, 0);
        }
      }
    }
  } else {
    for (int jb = j_begin; jb < j_end; jb += 16) {
      const int j = jb + 4 * q;
      f32x4 gv = *(const f32x4*)(grow + j);
      if (!row_ok) gv = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* wp = w_hh + (long long)j * H + k0 + i;
      const float w0 = wp[0], w1 = wp[H], w2 = wp[2 * (long long)H], w3 = wp[3 * (long long)H];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, gv[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, gv[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2, gv[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3, gv[3], acc, 0, 0, 0);
    }
  }
  __shared__ float red[4][4][64];
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][v][lane] = acc[v];
  __syncthreads();
  if (!ep) return;
  f32x4 dhp;
#pragma unroll
  for (int v = 0; v < 4; ++v) dhp[v] = e_direct[v] + red[0][v][lane] + red[1][v][lane] + red[2][v][lane] + red[3][v][lane];
  const long long o = (long long)be * H + je;
  if (!dout_prev) { *(f32x4*)(dh0 + o) = dhp; return; }  // t == 0
  f32x4 g_r, g_z, g_n, g_nr, d_out;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float r = e_w[0][v], z = e_w[1][v], nn = e_w[2][v], qn = e_w[3][v];
    const float dh = e_dout[v] + dhp[v];
    const float dn = dh * (1.f - z), dz = dh * (e_hp2[v] - nn);
    const float dan = dn * (1.f - nn * nn), daz = dz * z * (1.f - z);
    const float dar = dan * qn * r * (1.f - r);
    g_r[v] = dar; g_z[v] = daz; g_n[v] = dan; g_nr[v] = dan * r;
    d_out[v] = dh * z;
  }
  float* gi = dig_prev + (long long)be * 3 * H + je;
  float* gh = dhg_prev + (long long)be * 3 * H + je;
  *(f32x4*)gi = g_r; *(f32x4*)(gi + H) = g_z; *(f32x4*)(gi + 2 * H) = g_n;
  *(f32x4*)gh = g_r; *(f32x4*)(gh + H) = g_z; *(f32x4*)(gh + 2 * H) = g_nr;
  *(f32x4*)(direct + o) = d_out;
}

}  // namespace

extern "C" {

// ig: [L][B][3H] input-gate pre-activations WITHOUT bias (X W_ih^T); h0 [B][H]; out [L][B][H]; ws [L][B][4H].
int lt_gru_forward(const float* ig, const float* h0, const float* w_hh, const float* b_ih, const float* b_hh, int L, int B, int H,
                   float* out, float* ws, void* stream) {
  if (!ig || !h0 || !w_hh || !b_ih || !b_hh || !out || !ws || L < 1 || B < 1 || H < 64 || (H % 64) != 0) {
    lt_set_error("lt_gru_forward: invalid argument (H must be a multiple of 64)");
    return LT_EINVAL;
  }
  const dim3 grid((unsigned)(H / 16), (unsigned)((B + 15) / 16));
  const float* h = h0;
  for (int t = 0; t < L; ++t) {
    float* ht = out + (long long)t * B * H;
    const float* igt = ig + (long long)t * B * 3 * H;
    float* wst = ws + (long long)t * B * 4 * H;
    switch (H) {  // the student's encoder is H = 512; the other compile-time sizes cover the usual powers of two
      case 512: hipLaunchKernelGGL(lt_gru_step_fwd<8>, grid, dim3(256), 0, (hipStream_t)stream, igt, h, w_hh, b_ih, b_hh, ht, wst, B, H); break;
      case 256: hipLaunchKernelGGL(lt_gru_step_fwd<4>, grid, dim3(256), 0, (hipStream_t)stream, igt, h, w_hh, b_ih, b_hh, ht, wst, B, H); break;
      case 128: hipLaunchKernelGGL(lt_gru_step_fwd<2>, grid, dim3(256), 0, (hipStream_t)stream, igt, h, w_hh, b_ih, b_hh, ht, wst, B, H); break;
      default: hipLaunchKernelGGL(lt_gru_step_fwd<0>, grid, dim3(256), 0, (hipStream_t)stream, igt, h, w_hh, b_ih, b_hh, ht, wst, B, H); break;
    }
    h = ht;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

// dout [L][B][H]; dhn [B][H] or NULL; out / ws / h0 as left by lt_gru_forward; dig, dhg [L][B][3H] (outputs: gate gradients, input
// side and hidden side); scratch [3][B][H] (the first [B][H] is used); dh0 [B][H] (output).
int lt_gru_backward(const float* dout, const float* dhn, const float* out, const float* ws, const float* h0, const float* w_hh, int L, int B,
                    int H, float* dig, float* dhg, float* scratch, float* dh0, void* stream) {
  if (!dout || !out || !ws || !h0 || !w_hh || !dig || !dhg || !scratch || !dh0 || L < 1 || B < 1 || H < 64 || (H % 64) != 0) {
    lt_set_error("lt_gru_backward: invalid argument (H must be a multiple of 64)");
    return LT_EINVAL;
  }
  const dim3 grid((unsigned)(H / 16), (unsigned)((B + 15) / 16));
  const unsigned pw = (unsigned)(((long long)B * H + 255) / 256);
  float* direct = scratch;
  // open the recursion: gate gradients of the last step from dh' = dout_{L-1} + dhn; then one fused launch per step
  hipLaunchKernelGGL(lt_gru_step_bwd_gates, dim3(pw), dim3(256), 0, (hipStream_t)stream, dout + (long long)(L - 1) * B * H, dhn,
                     ws + (long long)(L - 1) * B * 4 * H, L > 1 ? out + (long long)(L - 2) * B * H : h0, dig + (long long)(L - 1) * B * 3 * H,
                     dhg + (long long)(L - 1) * B * 3 * H, direct, B, H);
  for (int t = L - 1; t >= 0; --t) {
    const bool last = t == 0;
    const float* hp2 = t > 1 ? out + (long long)(t - 2) * B * H : h0;  // h_{t-2}: the state BEFORE step t - 1
    const float* a0 = dhg + (long long)t * B * 3 * H;
    const float* a3 = last ? (const float*)nullptr : dout + (long long)(t - 1) * B * H;
    const float* a4 = last ? (const float*)nullptr : ws + (long long)(t - 1) * B * 4 * H;
    const float* a5 = last ? (const float*)nullptr : hp2;
    float* a6 = last ? (float*)nullptr : dig + (long long)(t - 1) * B * 3 * H;
    float* a7 = last ? (float*)nullptr : dhg + (long long)(t - 1) * B * 3 * H;
    switch (H) {
      case 512: hipLaunchKernelGGL(lt_gru_step_bwd_fused<8>, grid, dim3(256), 0, (hipStream_t)stream, a0, w_hh, direct, a3, a4, a5, a6, a7, dh0, B, H); break;
      case 256: hipLaunchKernelGGL(lt_gru_step_bwd_fused<4>, grid, dim3(256), 0, (hipStream_t)stream, a0, w_hh, direct, a3, a4, a5, a6, a7, dh0, B, H); break;
      case 128: hipLaunchKernelGGL(lt_gru_step_bwd_fused<2>, grid, dim3(256), 0, (hipStream_t)stream, a0, w_hh, direct, a3, a4, a5, a6, a7, dh0, B, H); break;
      default: hipLaunchKernelGGL(lt_gru_step_bwd_fused<0>, grid, dim3(256), 0, (hipStream_t)stream, a0, w_hh, direct, a3, a4, a5, a6, a7, dh0, B, H); break;
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

}  // extern "C"
