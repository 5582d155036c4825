// decoder_step.hip -- the per-step pieces of the attention-based LSTM decoder (LAS / MoChA), fp32.
//
// The teacher-forced loop `for i in range(ymax)` (neural_sp/models/seq2seq/decoders/las.py:667-776) runs a handful of
// [B, .]-sized operations per output token.  Its GEMMs were on this library's kernels since round 2; what the
// reference (and round 2's restatement) builds from element-wise ATen ops is fused here:
//   * additive attention energy  e[b,t] = sum_a v_a act(K[b,t,a] + Q[b,a] (+ C[b,t,a]))   (modules/attention.py:148-156
//     with tanh: LAS 'add' / 'location', C = the location-convolution term; modules/mocha/monotonic_energy.py:137-146
//     and chunk_energy.py with relu) -- one pass over the [B,T,adim] key projection instead of four
//     (add, tanh / relu, multiply by v, reduce), backward in two kernels (d tmp; d query / d v column sums);
//   * masked, sharpened soft-max over the encoder frames of one row (attention.py:170-176);
//   * the LSTM cell non-linearity (PyTorch gate order i, f, g, o; las.py:860-866 calls nn.LSTMCell).
// All are launch-latency-sized ([B,T] ~ a few thousand elements, [B,T,adim] ~ a few MB at most).
#include "common.h"

namespace {

__device__ __forceinline__ float step_act(float x, int act) { return act == NSP_ACT_RELU ? fmaxf(x, 0.f) : nsp_tanh(x); }
__device__ __forceinline__ float step_dact_from_out(float y, float x, int act) {
  return act == NSP_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : 1.f - y * y;
}

// one wave per (b, t) row
__global__ __launch_bounds__(256) void add_energy_fwd_kernel(const float* __restrict__ K, const float* __restrict__ Q,
                                                             const float* __restrict__ C, const float* __restrict__ v,
                                                             float* __restrict__ e, int rows, int T, int A, int act) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / T;
  const float* k = K + (long long)row * A;
  const float* c = C ? C + (long long)row * A : nullptr;
  const float* q = Q + (long long)b * A;
  float acc = 0.f;
  for (int a = lane; a < A; a += 64) acc += v[a] * step_act(k[a] + q[a] + (c ? c[a] : 0.f), act);
  acc = wave_reduce_sum(acc);
  if (lane == 0) e[row] = acc;
}

// d tmp[b,t,a] = de[b,t] v_a act'(tmp): the gradient of the key projection (and of the location term)
__global__ __launch_bounds__(256) void add_energy_bwd_kernel(const float* __restrict__ de, const float* __restrict__ K,
                                                             const float* __restrict__ Q, const float* __restrict__ C,
                                                             const float* __restrict__ v, float* __restrict__ dtmp,
                                                             int rows, int T, int A, int act) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / T;
  const float* k = K + (long long)row * A;
  const float* c = C ? C + (long long)row * A : nullptr;
  const float* q = Q + (long long)b * A;
  const float g = de[row];
  for (int a = lane; a < A; a += 64) {
    const float x = k[a] + q[a] + (c ? c[a] : 0.f);
    const float y = step_act(x, act);
    dtmp[(long long)row * A + a] = g * v[a] * step_dact_from_out(y, x, act);
  }
}

// dQ[b,a] = sum_t dtmp[b,t,a];  dvp[b,a] = sum_t de[b,t] act(tmp[b,t,a])  (dv = sum_b dvp): grid (B, ceil(A / 64)),
// the four waves of a block take every fourth frame of the block's 64 columns (256-B coalesced reads), LDS combine
__global__ __launch_bounds__(256) void add_energy_bwd_reduce_kernel(const float* __restrict__ de, const float* __restrict__ K,
                                                                    const float* __restrict__ Q, const float* __restrict__ C,
                                                                    const float* __restrict__ dtmp, float* __restrict__ dQ,
                                                                    float* __restrict__ dvp, int T, int A, int act) {
  __shared__ float part[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x, a = blockIdx.y * 64 + lane;
  float sq = 0.f, sv = 0.f;
  if (a < A) {
    const float q = Q[(long long)b * A + a];
    for (int t = w; t < T; t += 4) {
      const long long i = ((long long)b * T + t) * A + a;
      sq += dtmp[i];
      sv += de[(long long)b * T + t] * step_act(K[i] + q + (C ? C[i] : 0.f), act);
    }
  }
  part[0][w][lane] = sq;
  part[1][w][lane] = sv;
  __syncthreads();
  if (w == 0 && a < A) {
    dQ[(long long)b * A + a] = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
    dvp[(long long)b * A + a] = (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
  }
}

// aw = softmax(sharp * e) over the frames with mask != 0 (masked frames: e := -FLT_MAX, as masked_fill(NEG_INF));
// one wave per row
__global__ __launch_bounds__(256) void row_softmax_fwd_kernel(const float* __restrict__ e, const unsigned char* __restrict__ mask,
                                                              float* __restrict__ aw, int rows, int T, float sharp) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* er = e + (long long)row * T;
  const unsigned char* mr = mask ? mask + (long long)row * T : nullptr;
  float m = -FLT_MAX;
  for (int t = lane; t < T; t += 64) m = fmaxf(m, ((mr && !mr[t]) ? -FLT_MAX : er[t]) * sharp);
  m = wave_reduce_max(m);
  float s = 0.f;
  for (int t = lane; t < T; t += 64) s += expf(((mr && !mr[t]) ? -FLT_MAX : er[t]) * sharp - m);
  s = wave_reduce_sum(s);
  const float inv = 1.f / s;
  for (int t = lane; t < T; t += 64) aw[(long long)row * T + t] = expf(((mr && !mr[t]) ? -FLT_MAX : er[t]) * sharp - m) * inv;
}

__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* __restrict__ aw, const float* __restrict__ daw,
                                                              const unsigned char* __restrict__ mask, float* __restrict__ de,
                                                              int rows, int T, float sharp) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* a = aw + (long long)row * T;
  const float* d = daw + (long long)row * T;
  const unsigned char* mr = mask ? mask + (long long)row * T : nullptr;
  float dot = 0.f;
  for (int t = lane; t < T; t += 64) dot += a[t] * d[t];
  dot = wave_reduce_sum(dot);
  for (int t = lane; t < T; t += 64)
    de[(long long)row * T + t] = (mr && !mr[t]) ? 0.f : sharp * a[t] * (d[t] - dot);   // masked_fill passes no gradient
}

// LSTM cell: gates [B, 4H] in PyTorch order (i, f, g, o), pre-activation
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev, float* __restrict__ h,
                                     float* __restrict__ c, int B, int H) {
  const long long n = (long long)B * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / H, j = i % H;
    const float* g = gates + b * 4 * H;
    const float ig = nsp_sigmoid(g[j]), fg = nsp_sigmoid(g[H + j]), gg = nsp_tanh(g[2 * H + j]), og = nsp_sigmoid(g[3 * H + j]);
    const float cn = fg * c_prev[i] + ig * gg;
    c[i] = cn;
    h[i] = og * nsp_tanh(cn);
  }
}

__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc_next,
                                     const float* __restrict__ gates, const float* __restrict__ c_prev,
                                     const float* __restrict__ c, float* __restrict__ dgates, float* __restrict__ dc_prev,
                                     int B, int H) {
  const long long n = (long long)B * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / H, j = i % H;
    const float* g = gates + b * 4 * H;
    float* dg = dgates + b * 4 * H;
    const float ig = nsp_sigmoid(g[j]), fg = nsp_sigmoid(g[H + j]), gg = nsp_tanh(g[2 * H + j]), og = nsp_sigmoid(g[3 * H + j]);
    const float tc = nsp_tanh(c[i]);
    const float dhv = dh ? dh[i] : 0.f;
    const float dcn = (dc_next ? dc_next[i] : 0.f) + dhv * og * (1.f - tc * tc);
    dg[j] = dcn * gg * ig * (1.f - ig);
    dg[H + j] = dcn * c_prev[i] * fg * (1.f - fg);
    dg[2 * H + j] = dcn * ig * (1.f - gg * gg);
    dg[3 * H + j] = dhv * tc * og * (1.f - og);
    dc_prev[i] = dcn * fg;
  }
}

}  // namespace

extern "C" int nsp_add_energy_fwd(const float* K, const float* Q, const float* C, const float* v, float* e, int B, int T,
                                  int A, int act, void* stream) {
  if (B <= 0 || T <= 0 || A <= 0) return NSP_OK;
  if (!K || !Q || !v || !e || (act != NSP_ACT_TANH && act != NSP_ACT_RELU)) return NSP_EINVAL;
  hipLaunchKernelGGL(add_energy_fwd_kernel, dim3(nsp_cdiv(B * T, 4)), dim3(256), 0, (hipStream_t)stream, K, Q, C, v, e,
                     B * T, T, A, act);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_add_energy_bwd(const float* de, const float* K, const float* Q, const float* C, const float* v,
                                  float* dtmp, float* dQ, float* dv_part, int B, int T, int A, int act, void* stream) {
  if (B <= 0 || T <= 0 || A <= 0) return NSP_OK;
  if (!de || !K || !Q || !v || !dtmp || !dQ || !dv_part || (act != NSP_ACT_TANH && act != NSP_ACT_RELU)) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(add_energy_bwd_kernel, dim3(nsp_cdiv(B * T, 4)), dim3(256), 0, st, de, K, Q, C, v, dtmp, B * T, T, A, act);
  hipLaunchKernelGGL(add_energy_bwd_reduce_kernel, dim3(B, nsp_cdiv(A, 64)), dim3(256), 0, st, de, K, Q, C, dtmp, dQ, dv_part,
                     T, A, act);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_row_softmax_fwd(const float* e, const unsigned char* mask, float* aw, int rows, int T, float sharp,
                                   void* stream) {
  if (rows <= 0 || T <= 0) return NSP_OK;
  if (!e || !aw) return NSP_EINVAL;
  hipLaunchKernelGGL(row_softmax_fwd_kernel, dim3(nsp_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, e, mask, aw, rows, T, sharp);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_row_softmax_bwd(const float* aw, const float* daw, const unsigned char* mask, float* de, int rows, int T,
                                   float sharp, void* stream) {
  if (rows <= 0 || T <= 0) return NSP_OK;
  if (!aw || !daw || !de) return NSP_EINVAL;
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3(nsp_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, aw, daw, mask, de, rows, T, sharp);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_cell_fwd(const float* gates, const float* c_prev, float* h, float* c, int B, int H, void* stream) {
  if (B <= 0 || H <= 0) return NSP_OK;
  if (!gates || !c_prev || !h || !c) return NSP_EINVAL;
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(nsp_cdiv(B * H, 256)), dim3(256), 0, (hipStream_t)stream, gates, c_prev, h, c, B, H);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_cell_bwd(const float* dh, const float* dc_next, const float* gates, const float* c_prev, const float* c,
                                 float* dgates, float* dc_prev, int B, int H, void* stream) {
  if (B <= 0 || H <= 0) return NSP_OK;
  if (!gates || !c_prev || !c || !dgates || !dc_prev) return NSP_EINVAL;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(nsp_cdiv(B * H, 256)), dim3(256), 0, (hipStream_t)stream, dh, dc_next, gates,
                     c_prev, c, dgates, dc_prev, B, H);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
