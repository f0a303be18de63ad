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
__global__ __launch_bounds__(256) void lt_gru_step_fwd(const float* __restrict__ ig, const float* __restrict__ h, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ h_out,
                                                       float* __restrict__ ws, int B, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int row = b0 + i;                       // batch row this lane feeds as the B operand
  const bool row_ok = row < B;
  const float* hrow = h + (long long)(row_ok ? row : 0) * H;
  const float* wr = w_hh + (long long)(j0 + i) * H;  // A operand rows: unit j0 + i of gate r; + H*H, + 2*H*H for z, n
  const long long gate = (long long)H * H;
  f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;
  const int kq = H / 4, k_begin = wave * kq, k_end = k_begin + kq;
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
  // D layout: acc[v] of lane (n = l % 16, g = l / 16) is D[unit 4 g + v][row n]
  __shared__ float red[3][4][4][64];  // [gate][wave][v][lane]
#pragma unroll
  for (int v = 0; v < 4; ++v) { red[0][wave][v][lane] = acc_r[v]; red[1][wave][v][lane] = acc_z[v]; red[2][wave][v][lane] = acc_n[v]; }
  __syncthreads();
  if (wave != 0) return;
  const int n = lane & 15, g = lane >> 4;
  const int b = b0 + n;
  if (b >= B) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int j = j0 + 4 * g + v;
    const float sr = red[0][0][v][lane] + red[0][1][v][lane] + red[0][2][v][lane] + red[0][3][v][lane];
    const float sz = red[1][0][v][lane] + red[1][1][v][lane] + red[1][2][v][lane] + red[1][3][v][lane];
    const float sn = red[2][0][v][lane] + red[2][1][v][lane] + red[2][2][v][lane] + red[2][3][v][lane];
    const float* igb = ig + (long long)b * 3 * H;
    const float r = sigmoidf_(igb[j] + b_ih[j] + sr + b_hh[j]);
    const float z = sigmoidf_(igb[H + j] + b_ih[H + j] + sz + b_hh[H + j]);
    const float qn = sn + b_hh[2 * H + j];
    const float nn = tanhf_(igb[2 * H + j] + b_ih[2 * H + j] + r * qn);
    const float hp = h[(long long)b * H + j];
    h_out[(long long)b * H + j] = nn + z * (hp - nn);
    float* w = ws + (long long)b * 4 * H;
    w[j] = r; w[H + j] = z; w[2 * H + j] = nn; w[3 * H + j] = qn;
  }
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

// ---- backward step, recurrent part: dh_prev = dh_direct + dhg_t W_hh for a 16-k x 16-row tile -------------------------------
// grid (H / 16, ceil(B / 16)), block 256 (wave w reduces j in [w * 3H / 4, (w + 1) * 3H / 4))
__global__ __launch_bounds__(256) void lt_gru_step_bwd_dh(const float* __restrict__ dhg, const float* __restrict__ w_hh,
                                                          const float* __restrict__ dh_direct, float* __restrict__ dh_prev, int B, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int row = b0 + i;
  const bool row_ok = row < B;
  const float* grow = dhg + (long long)(row_ok ? row : 0) * 3 * H;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int jq = 3 * H / 4, j_begin = wave * jq, j_end = j_begin + jq;
  for (int jb = j_begin; jb < j_end; jb += 16) {
    const int j = jb + 4 * q;
    f32x4 gv = *(const f32x4*)(grow + j);
    if (!row_ok) gv = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A operand: A[out = k0 + i][reduction index j + s] = W[j + s][k0 + i]  (16 consecutive k across the lanes of one q: coalesced)
    const float* wp = w_hh + (long long)j * H + k0 + i;
    const float w0 = wp[0], w1 = wp[H], w2 = wp[2 * (long long)H], w3 = wp[3 * (long long)H];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, gv[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, gv[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2, gv[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3, gv[3], acc, 0, 0, 0);
  }
  __shared__ float red[4][4][64];
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][v][lane] = acc[v];
  __syncthreads();
  if (wave != 0) return;
  const int n = lane & 15, g = lane >> 4;
  const int b = b0 + n;
  if (b >= B) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const long long o = (long long)b * H + k0 + 4 * g + v;
    dh_prev[o] = dh_direct[o] + red[0][v][lane] + red[1][v][lane] + red[2][v][lane] + red[3][v][lane];
  }
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
    hipLaunchKernelGGL(lt_gru_step_fwd, grid, dim3(256), 0, (hipStream_t)stream, ig + (long long)t * B * 3 * H, h, w_hh, b_ih, b_hh, ht,
                       ws + (long long)t * B * 4 * H, B, H);
    h = ht;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

// dout [L][B][H]; dhn [B][H] or NULL; out / ws / h0 as left by lt_gru_forward; dig, dhg [L][B][3H] (outputs: gate gradients, input
// side and hidden side); scratch [3][B][H]; dh0 [B][H] (output).
int lt_gru_backward(const float* dout, const float* dhn, const float* out, const float* ws, const float* h0, const float* w_hh, int L, int B,
                    int H, float* dig, float* dhg, float* scratch, float* dh0, void* stream) {
  if (!dout || !out || !ws || !h0 || !w_hh || !dig || !dhg || !scratch || !dh0 || L < 1 || B < 1 || H < 64 || (H % 64) != 0) {
    lt_set_error("lt_gru_backward: invalid argument (H must be a multiple of 64)");
    return LT_EINVAL;
  }
  const dim3 grid((unsigned)(H / 16), (unsigned)((B + 15) / 16));
  const unsigned pw = (unsigned)(((long long)B * H + 255) / 256);
  float* direct = scratch;
  float* dh_buf[2] = {scratch + (long long)B * H, scratch + 2 * (long long)B * H};
  const float* dh_next = dhn;
  for (int t = L - 1; t >= 0; --t) {
    const float* hp = t > 0 ? out + (long long)(t - 1) * B * H : h0;
    float* dst = t > 0 ? dh_buf[t & 1] : dh0;
    hipLaunchKernelGGL(lt_gru_step_bwd_gates, dim3(pw), dim3(256), 0, (hipStream_t)stream, dout + (long long)t * B * H, dh_next,
                       ws + (long long)t * B * 4 * H, hp, dig + (long long)t * B * 3 * H, dhg + (long long)t * B * 3 * H, direct, B, H);
    hipLaunchKernelGGL(lt_gru_step_bwd_dh, grid, dim3(256), 0, (hipStream_t)stream, dhg + (long long)t * B * 3 * H, w_hh, direct, dst, B, H);
    dh_next = dst;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { lt_set_error(hipGetErrorString(e)); return LT_EHIP; }
  return LT_OK;
}

}  // extern "C"
