// lstm.hip -- single-layer LSTM recurrence (the RNN-T prediction network,
// reference rnn_transducer.py:101-111,278-311: nn.LSTM(1 layer, batch_first)).
//
// The input projection x W_ih^T + b for all time steps, and all weight / input
// gradients, are plain GEMMs over [B*L, .] done by the MFMA GEMM kernels.  What
// is inherently sequential is h_{t-1} W_hh^T: one small kernel per time step.
// A workgroup owns 4 hidden units (all four gates i,f,g,o of them = one 16-row
// MFMA tile of W_hh) for up to 16 batch rows; its 4 waves split the reduction
// over H (operands straight from L2 into registers -- W_hh is 8 MB in bf16 and
// each XCD's 1/8 of it stays L2 resident across steps), the partial tiles meet
// in LDS and 64 threads apply the cell update.
// Backward walks t = L-1..0 with the transposed recurrence
// dh_{t} += dgates_{t+1} W_hh (reduction over 4H split across 16 waves).
// Gate order is PyTorch's (i, f, g, o).  Latency-bound by construction:
// ~2 x L launches per layer and direction.
#include "common.h"

namespace {

template <int MODE>
__device__ __forceinline__ f32x4 dot_tile(const void* __restrict__ arow, bool a_valid,
                                          const void* __restrict__ brow, int K, int lane) {
  // returns D[i][j] = sum_k A[i][k] B[j][k] for the lane's (A row i = lane&15 | B row j = lane&15)
  // layout: D lane holds j = lane&15, i = (lane>>4)*4 + reg  (A is the SECOND mfma operand)
  const int g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    const __bf16* a = reinterpret_cast<const __bf16*>(arow);
    const __bf16* b = reinterpret_cast<const __bf16*>(brow);
    for (int k0 = 0; k0 < K; k0 += 256) {
      bf16x8 af[8], bf[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = k0 + s * 32 + g * 8;
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
        af[s] = (a_valid && k < K) ? *reinterpret_cast<const bf16x8*>(a + k) : z;
        bf[s] = (k < K) ? *reinterpret_cast<const bf16x8*>(b + k) : z;
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], bf[s], acc, 0, 0, 0);
    }
  } else {
    const float* a = reinterpret_cast<const float*>(arow);
    const float* b = reinterpret_cast<const float*>(brow);
    for (int k0 = 0; k0 < K; k0 += 64) {
      float af[16], bf[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int k = k0 + s * 4 + g;
        af[s] = (a_valid && k < K) ? a[k] : 0.f;
        bf[s] = (k < K) ? b[k] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc, 0, 0, 0);
    }
  }
  return acc;
}

// One step's loads must all be in flight at once (the step is a pure latency chain: L2 load ->
// MFMA -> LDS -> gate math -> store), so the reduction dimension is split across the waves of a
// workgroup and the units across MANY workgroups: the first version (64 workgroups, each wave
// walking K = 1024 in four dependent rounds) took 10 us per step.

// forward step t.  grid: (H/4, ceil(B/16)); block 256.  The workgroup owns 4 hidden units = 16
// gate rows (row j -> gate j>>2, unit u0 + (j&3)); wave w reduces k in [w*Kw, (w+1)*Kw).
template <int MODE>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(
    const float* __restrict__ gi, const void* __restrict__ Whh, float* __restrict__ y,
    void* __restrict__ yshadow, float* __restrict__ c_all, float* __restrict__ gates, int B, int L,
    int H, int t, int Kw) {
  __shared__ float part[4][16][17];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 4, b0 = blockIdx.y * 16;
  const int r = lane & 15;
  const size_t esz = MODE == 0 ? 2 : 4;
  // the cell-update threads fetch their elementwise inputs BEFORE the dot product: these loads do
  // not depend on it, and issuing them after the barrier put a second full memory latency on the
  // step's critical path
  const int bb = threadIdx.x >> 2, uu = threadIdx.x & 3;
  const bool upd = threadIdx.x < 64 && (b0 + bb) < B;
  const long long row = (long long)(b0 + bb) * L + t;
  const int u = u0 + uu;
  float gin[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
  if (upd) {
    const float* g = gi + row * 4 * H;
#pragma unroll
    for (int q = 0; q < 4; ++q) gin[q] = g[q * H + u];
    if (t > 0) cp = c_all[(row - 1) * H + u];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int kbeg = w * Kw;
  const int klen = min(Kw, H - kbeg);
  if (t > 0 && klen > 0) {
    const bool a_valid = (b0 + r) < B;
    const char* arow = reinterpret_cast<const char*>(yshadow) +
                       (((long long)(min(b0 + r, B - 1)) * L + (t - 1)) * H + kbeg) * esz;
    const char* brow = reinterpret_cast<const char*>(Whh) +
                       ((long long)((r >> 2) * H + u0 + (r & 3)) * H + kbeg) * esz;
    // D[i = batch row][j = gate row]; lane: j = lane&15, i = (lane>>4)*4+reg
    acc = dot_tile<MODE>(arow, a_valid, brow, klen, lane);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[w][(lane >> 4) * 4 + e][r] = acc[e];
  __syncthreads();
  if (upd) {
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      pre[q] = part[0][bb][q * 4 + uu] + part[1][bb][q * 4 + uu] + part[2][bb][q * 4 + uu] + part[3][bb][q * 4 + uu];
    const float ig = nsp_sigmoid(gin[0] + pre[0]);
    const float fg = nsp_sigmoid(gin[1] + pre[1]);
    const float gg = nsp_tanh(gin[2] + pre[2]);
    const float og = nsp_sigmoid(gin[3] + pre[3]);
    const float c = fg * cp + ig * gg;
    const float h = og * nsp_tanh(c);
    c_all[row * H + u] = c;
    y[row * H + u] = h;
    if (MODE == 0) reinterpret_cast<__bf16*>(yshadow)[row * H + u] = (__bf16)h;
    float* gs = gates + row * 4 * H;
    gs[u] = ig; gs[H + u] = fg; gs[2 * H + u] = gg; gs[3 * H + u] = og;
  }
}

// backward step t.  dgates (fp32) and its shadow (bf16 in MODE 0, unused in MODE 1) are
// [B, L, 4H]; WhhT is W_hh^T [H, 4H]; dc is [B, H] (carried across steps).
// grid: (H/16, ceil(B/16)); block 1024: the 16 waves split the 4H-long reduction.
template <int MODE>
__global__ __launch_bounds__(1024) void lstm_step_bwd_kernel(
    const float* __restrict__ dy, const void* __restrict__ WhhT, const float* __restrict__ c_all,
    const float* __restrict__ gates, float* __restrict__ dgates, void* __restrict__ dgshadow,
    float* __restrict__ dc, int B, int L, int H, int t, int Kw) {
  __shared__ float part[16][16][17];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int r = lane & 15;
  const size_t esz = MODE == 0 ? 2 : 4;
  const int K4 = 4 * H;
  // elementwise inputs first (independent of the dot product; see the forward kernel)
  const int bb = threadIdx.x >> 4, uu = threadIdx.x & 15;
  const int u = u0 + uu;
  const bool upd = threadIdx.x < 256 && (b0 + bb) < B && u < H;
  const long long row = (long long)(b0 + bb) * L + t;
  float dyv = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cp = 0.f, dcn = 0.f;
  if (upd) {
    const float* gs = gates + row * K4;
    dyv = dy[row * H + u];
    ig = gs[u]; fg = gs[H + u]; gg = gs[2 * H + u]; og = gs[3 * H + u];
    c = c_all[row * H + u];
    if (t > 0) cp = c_all[(row - 1) * H + u];
    if (t + 1 < L) dcn = dc[(long long)(b0 + bb) * H + u];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int kbeg = w * Kw;
  const int klen = min(Kw, K4 - kbeg);
  if (t + 1 < L && klen > 0) {
    // dh_rec[b][u] = sum_k dgates_{t+1}[b][k] * W_hh[k][u]
    const bool a_valid = (b0 + r) < B;
    const void* ag = MODE == 0 ? (const void*)dgshadow : (const void*)dgates;
    const char* arow = reinterpret_cast<const char*>(ag) +
                       (((long long)(min(b0 + r, B - 1)) * L + (t + 1)) * K4 + kbeg) * esz;
    const char* brow = reinterpret_cast<const char*>(WhhT) + ((long long)(u0 + r) * K4 + kbeg) * esz;
    acc = dot_tile<MODE>(arow, a_valid, brow, klen, lane);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[w][(lane >> 4) * 4 + e][r] = acc[e];
  __syncthreads();
  if (upd) {
    float dh = dyv;
#pragma unroll
    for (int q = 0; q < 16; ++q) dh += part[q][bb][uu];
    const float tc = nsp_tanh(c);
    const float dct = dcn + dh * og * (1.f - tc * tc);
    const float d_o = dh * tc * og * (1.f - og);
    const float d_i = dct * gg * ig * (1.f - ig);
    const float d_f = dct * cp * fg * (1.f - fg);
    const float d_g = dct * ig * (1.f - gg * gg);
    dc[(long long)(b0 + bb) * H + u] = dct * fg;
    float* dg = dgates + row * K4;
    dg[u] = d_i; dg[H + u] = d_f; dg[2 * H + u] = d_g; dg[3 * H + u] = d_o;
    if (MODE == 0) {
      __bf16* ds = reinterpret_cast<__bf16*>(dgshadow) + row * K4;
      ds[u] = (__bf16)d_i; ds[H + u] = (__bf16)d_f; ds[2 * H + u] = (__bf16)d_g; ds[3 * H + u] = (__bf16)d_o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wavefront over (layer, time): see nsp_lstm_stack_params in include/nsp_hip.h.
// forward stage s.  grid: (H/4, nl, ceil(B/16)); block 512 = 8 waves splitting the reduction.
// Layer l handles t = s - l.  The workgroup owns 4 hidden units (16 gate rows).
__global__ __launch_bounds__(512) void lstm_stack_fwd_kernel(const nsp_lstm_stack_params p, int s) {
  __shared__ float part[8][16][17];
  const int l = blockIdx.y;
  const int t = s - l;
  if (t < 0 || t >= p.L) return;
  const int H = p.H, L = p.L, B = p.B, top = p.nl - 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 4, b0 = blockIdx.z * 16;
  const int r = lane & 15;
  const int Ktot = l == 0 ? H : 2 * H;
  const int Kw = Ktot >> 3;
  // cell-update inputs first (independent of the dot product)
  const int bb = threadIdx.x >> 2, uu = threadIdx.x & 3;
  const bool upd = threadIdx.x < 64 && (b0 + bb) < B;
  const long long row = (long long)(b0 + bb) * L + t;
  const int u = u0 + uu;
  float gin[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
  if (upd) {
    if (l == 0) {
      const float* g = p.gi0 + row * 4 * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) gin[q] = g[q * H + u];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) gin[q] = p.bias[l][q * H + u];
    }
    if (t > 0) cp = p.c_all[l][(row - 1) * H + u];
  }
  const int kbeg = w * Kw;
  const bool a_valid = (b0 + r) < B;
  const long long arow = (long long)min(b0 + r, B - 1) * L + t;
  const __bf16* abase;
  if (l == 0 || kbeg >= H)   // recurrent half: h_{t-1} (hp16 holds h shifted by one step, 0 at t = 0)
    abase = reinterpret_cast<const __bf16*>(p.hp16[l]) + arow * H + (l == 0 ? kbeg : kbeg - H);
  else                       // input half: dropout(h_t of the layer below)
    abase = reinterpret_cast<const __bf16*>(p.yd16[l - 1]) + arow * H + kbeg;
  const __bf16* brow = reinterpret_cast<const __bf16*>(p.w[l]) +
                       (long long)((r >> 2) * H + u0 + (r & 3)) * Ktot + kbeg;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (t > 0 || (l > 0 && kbeg < H)) acc = dot_tile<0>(abase, a_valid, brow, Kw, lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) part[w][(lane >> 4) * 4 + e][r] = acc[e];
  __syncthreads();
  if (upd) {
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = gin[q];
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) v += part[ww][bb][q * 4 + uu];
      pre[q] = v;
    }
    const float ig = nsp_sigmoid(pre[0]);
    const float fg = nsp_sigmoid(pre[1]);
    const float gg = nsp_tanh(pre[2]);
    const float og = nsp_sigmoid(pre[3]);
    const float c = fg * cp + ig * gg;
    const float h = og * nsp_tanh(c);
    p.c_all[l][row * H + u] = c;
    float* gs = p.gates[l] + row * 4 * H;
    gs[u] = ig; gs[H + u] = fg; gs[2 * H + u] = gg; gs[3 * H + u] = og;
    __bf16* hp = reinterpret_cast<__bf16*>(p.hp16[l]);
    if (t + 1 < L) hp[(row + 1) * H + u] = (__bf16)h;
    if (t == 0) hp[row * H + u] = (__bf16)0.f;
    if (l == top) {
      p.y_top[row * H + u] = h;
    } else {
      float hd = h;
      if (p.dropout_p > 0.f)
        hd *= nsp_keep_scale(p.seed[l], p.offset[l] + (unsigned long long)(row * H + u), p.dropout_p);
      reinterpret_cast<__bf16*>(p.yd16[l])[row * H + u] = (__bf16)hd;
    }
  }
}

// backward stage s.  grid: (H/16, nl, ceil(B/16)); block 1024 = 16 waves.  Layer l handles
// t = L-1 - (s - (top - l)).  dh = [l == top ? dy : mask * (dgates_{l+1}[t] W_ih_{l+1})] +
// dgates_l[t+1] W_hh_l; for l < top the two products are one reduction over the concatenated
// [W_ih_{l+1}^T | W_hh_l^T] rows (waves 0-7: input-gradient half, waves 8-15: recurrent half).
__global__ __launch_bounds__(1024) void lstm_stack_bwd_kernel(const nsp_lstm_stack_params p, int s) {
  __shared__ float part[16][16][17];
  const int top = p.nl - 1;
  const int l = blockIdx.y;
  const int H = p.H, L = p.L, B = p.B;
  const int t = L - 1 - (s - (top - l));
  if (t < 0 || t >= L) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 16, b0 = blockIdx.z * 16;
  const int r = lane & 15;
  const int K4 = 4 * H;
  const bool has_ext = l < top;
  const int bb = threadIdx.x >> 4, uu = threadIdx.x & 15;
  const int u = u0 + uu;
  const bool upd = threadIdx.x < 256 && (b0 + bb) < B;
  const long long row = (long long)(b0 + bb) * L + t;
  float dyv = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cp = 0.f, dcn = 0.f, keep = 1.f;
  if (upd) {
    const float* gs = p.gates[l] + row * K4;
    if (!has_ext) dyv = p.dy_top[row * H + u];
    else if (p.dropout_p > 0.f)
      keep = nsp_keep_scale(p.seed[l], p.offset[l] + (unsigned long long)(row * H + u), p.dropout_p);
    ig = gs[u]; fg = gs[H + u]; gg = gs[2 * H + u]; og = gs[3 * H + u];
    c = p.c_all[l][row * H + u];
    if (t > 0) cp = p.c_all[l][(row - 1) * H + u];
    if (t + 1 < L) dcn = p.dc[l][(long long)(b0 + bb) * H + u];
  }
  const int Ktot = has_ext ? 2 * K4 : K4;
  const int Kw = Ktot >> 4;
  const int kbeg = w * Kw;
  const bool a_valid = (b0 + r) < B;
  const long long arow = (long long)min(b0 + r, B - 1) * L + t;
  const __bf16* abase;
  bool live;
  if (has_ext && kbeg < K4) {   // d(input of layer l+1) at the SAME time step
    abase = reinterpret_cast<const __bf16*>(p.dg16[l + 1]) + arow * K4 + kbeg;
    live = true;
  } else {                      // recurrent: this layer's dgates at t + 1
    abase = reinterpret_cast<const __bf16*>(p.dg16[l]) + (arow + 1) * K4 + (has_ext ? kbeg - K4 : kbeg);
    live = t + 1 < L;
  }
  const __bf16* brow = reinterpret_cast<const __bf16*>(p.w[l]) + (long long)(u0 + r) * Ktot + kbeg;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (live) acc = dot_tile<0>(abase, a_valid, brow, Kw, lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) part[w][(lane >> 4) * 4 + e][r] = acc[e];
  __syncthreads();
  if (upd) {
    float ext = 0.f, rec = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) ext += part[q][bb][uu];
#pragma unroll
    for (int q = 8; q < 16; ++q) rec += part[q][bb][uu];
    const float dh = has_ext ? keep * ext + rec : dyv + ext + rec;
    const float tc = nsp_tanh(c);
    const float dct = dcn + dh * og * (1.f - tc * tc);
    const float d_o = dh * tc * og * (1.f - og);
    const float d_i = dct * gg * ig * (1.f - ig);
    const float d_f = dct * cp * fg * (1.f - fg);
    const float d_g = dct * ig * (1.f - gg * gg);
    p.dc[l][(long long)(b0 + bb) * H + u] = dct * fg;
    __bf16* ds = reinterpret_cast<__bf16*>(p.dg16[l]) + row * K4;
    ds[u] = (__bf16)d_i; ds[H + u] = (__bf16)d_f; ds[2 * H + u] = (__bf16)d_g; ds[3 * H + u] = (__bf16)d_o;
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent variants: weights in registers for the whole sequence, grid barrier per stage.
#ifndef NSP_LSTM_FWD_DEPTH
#define NSP_LSTM_FWD_DEPTH 8
#endif
#ifndef NSP_LSTM_BWD_DEPTH
#define NSP_LSTM_BWD_DEPTH 12
#endif
#ifndef NSP_LSTM_ABL
#define NSP_LSTM_ABL 0   // development ablations (tools/lstm_ablate.sh), 0 in the product: 1 no operand loads, 4 no forward
                         // state stores, 8 no grid barrier, 16 no publishing stores, 32 no backward dgates image, 64 no
                         // prefetch of the saved forward state
#endif
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

// One monotonic counter; every workgroup arrives once per stage.  Payload (h / dgates) is stored
// write-through (agent-scope atomic stores -> sc1) by the publishing lanes, every wave drains
// its stores, then ONE lane arrives.  The consumer side polls relaxed, takes ONE agent-scope
// acquire (invalidates this CU's L1) and the workgroup barrier that follows covers all waves.
__device__ __forceinline__ void grid_arrive(unsigned int* sync) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*)sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns false once the launch has been declared dead (a bounded spin expired somewhere)
__device__ __forceinline__ bool grid_wait(unsigned int* sync, unsigned int target, int* dead_sh) {
  if (threadIdx.x == 0) {
    int dead = 0;
    unsigned int spins = 0;
    while (__hip_atomic_load((gu32*)sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0) {
        if (spins > (1u << 22) ||
            __hip_atomic_load((gu32*)(sync + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store((gu32*)(sync + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dead = 1;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *dead_sh = dead;
  }
  __syncthreads();
  return *dead_sh == 0;
}

// ---- persistent kernels, second version (round 3).  What the first version's stage time was made of, measured at
// B = 64, 2 x 1024, L = 200 by compiling pieces out (profiles/r03t_lstm_*_ablation.log; microseconds per stage):
//   forward  17.1 = operand loads 6.0 + grid barrier 6.0 + state stores 2.3 + MFMA 1.1 + the rest 1.7
//   backward 43   = operand loads 27.7 + grid barrier 5.9 + MFMA 1.5 + the rest ~8
// Now 8.5 and ~21 (profiles/r03ae_lstm_v3_bwd_ablation.log), from four changes:
//   * the exchange buffers below: the loads were never latency-bound -- more of them in flight, another row stride,
//     another unit -> XCD map, a rotated sweep order, no L2 invalidate all measured nothing -- but bound by the vector
//     L1's line lookups;
//   * a layer's operand from the layer BELOW / ABOVE (forward: dropout(h_{l-1}) at the same step; backward: the
//     dgates of layer l+1 at the same step) is consumed one stage late (layer l lags its neighbour by TWO stages
//     instead of one): its loads and MFMAs run between the workgroup's arrival at the barrier and the end of its
//     wait and leave their partial sums in the LDS slots of the reduction; only the recurrent half is in front of
//     the cell update;
//   * no register spills: a rolling ring of 8 (forward) / 12 (backward, now 8 waves x 256 VGPRs instead of 16 x 128)
//     fragment loads with (uniform base + 32-bit lane offset + immediate) addresses.  A spilled build of the same
//     source (deeper ring, 30 - 90 dwords of scratch) is 1.7 - 2.5x slower per stage: scratch reloads are memory round
//     trips on the critical path;
//   * the forward's saved state (c, activated gates, y) and the [B, L, .] images for the GEMMs are stored at the END
//     of the stage, behind the arrival and the early half's loads (vmcnt is in-order).
// Rows beyond B carry garbage through the exchange (their MFMA rows are independent and never stored).

// Exchange buffers.  An MFMA operand fragment read straight from a [B, L, K] image puts 16 DIFFERENT rows (utterances)
// into every quarter-wave, 16 bytes each: the vector L1 looks up 16 lines per quarter-wave and such loads crawl at
// ~16 B/clk per CU (40 GB/s: 6 us for the forward's 256 KB, 27 us for the backward's 1 MB per stage -- whatever the
// number of loads in flight, the row stride, the XCD mapping or the sweep order; profiles/r03*_lstm_*.log).  So the
// recurrent hand-over goes through a FRAGMENT-MAJOR image per layer,
//   lstm_xchg_index(tau, bk, k, b) = (((tau * NB + bk) * K/8 + k/8) * 16 + b) * 8 + k%8      (bf16 elements),
// step tau, batch block bk of 16 utterances b, column k of K (H forward, 4H backward): the 16 utterances of one
// 8-column group are 256 contiguous bytes, a wave-wide fragment load is 1 KB contiguous, and a workgroup publishes
// its 16 units as whole 256-byte pieces.  The [B, L, K] images the weight-gradient GEMMs need are written as well,
// by plain stores after the arrival at the barrier (off the critical path).  p.xchg[l]: forward 2 x L x NB x H x 16
// elements (h, then dropout(h)), backward L x NB x 4H x 16; the caller sizes them for NB = 4.
//
// forward.  grid: (H/16, nl); block 512.  Workgroup (ub, l) owns hidden units 16*ub.. of layer l: wave w holds, for
// each of the 4 gates, the 16 W rows of those units restricted to its 1/8 of the recurrent reduction (brec) and,
// for layers >= 1, of the input reduction (bin) as MFMA B-fragments.  NF = H / 256 fragments per wave and half.
// Stage s runs step t = s - 2 l of layer l.
template <int NB, int NF>
__global__ __launch_bounds__(512) void lstm_stack_fwd_persistent_kernel(const nsp_lstm_stack_params p,
                                                                        unsigned int* sync) {
  extern __shared__ __attribute__((aligned(16))) float fdyn[];
  typedef float part_t[8][4][16][17];
  part_t* part = reinterpret_cast<part_t*>(fdyn);                                   // [NB]
  typedef __bf16 hs_t[2][16][16];
  hs_t* hs = reinterpret_cast<hs_t*>(fdyn + NB * (8 * 4 * 16 * 17));                // [NB]
  __shared__ int dead_sh;
  constexpr int NU = (NB + 1) / 2;   // blocks per cell-update thread (thread group tg: blocks tg, tg+2)
  constexpr int Q = NB * NF;         // operand fragments per wave and half: all in flight at once
  const int l = blockIdx.y;
  const int H = p.H, L = p.L, B = p.B, top = p.nl - 1;
  const int nwg = gridDim.x * gridDim.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * 16;
  const bool has_in = l > 0;
  const int Ktot = has_in ? 2 * H : H;
  const int koff = w * (NF * 32) + g * 8;   // the lane's first column inside a half
  bf16x8 bin[4][NF], brec[4][NF];
  {
    const __bf16* W = reinterpret_cast<const __bf16*>(p.w[l]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __bf16* wr = W + (long long)(j * H + u0 + r) * Ktot + koff;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        brec[j][f] = *reinterpret_cast<const bf16x8*>(wr + (has_in ? H : 0) + f * 32);
        bin[j][f] = *reinterpret_cast<const bf16x8*>(wr + f * 32);   // layer 0: never used
      }
    }
  }
  // exchange buffers (see lstm_xchg_index): slot tau of hx holds h_tau, of ydx dropout(h_tau)
  __bf16* hx = reinterpret_cast<__bf16*>(p.xchg[l]);
  __bf16* ydx = hx + (long long)L * NB * H * 16;
  const __bf16* inx = has_in ? reinterpret_cast<const __bf16*>(p.xchg[l - 1]) + (long long)L * NB * H * 16 : hx;
  const int lane_off = ((koff >> 3) * 16 + r) * 8;

  // acc(block) = init + A(16 utterances x this wave's k-slice) . W^T -> the wave's LDS slot
  // acc(block) = init + A(16 utterances x this wave's k-slice of exchange slot `slot`) . W^T -> the wave's LDS slot.
  // Addresses are (uniform slot / block base) + (32-bit lane offset) + (immediate fragment offset): one VGPR for all.
  auto dot_half = [&](const __bf16* xbuf, int slot, const bf16x8 (&bw)[4][NF], bool init_from_slot, bool live) {
    constexpr int D = Q < NSP_LSTM_FWD_DEPTH ? Q : NSP_LSTM_FWD_DEPTH;   // fragment loads in flight per wave
    bf16x8 ring[D];
    const __bf16* blk0 = xbuf + (long long)slot * NB * H * 16;
    auto frag = [&](int q) {
#if NSP_LSTM_ABL & 1
      return bw[q % 4][q % NF];
#else
      return *reinterpret_cast<const bf16x8*>(blk0 + (long long)(q / NF) * H * 16 + (lane_off + (q % NF) * 512));
#endif
    };
    if (live) {
#pragma unroll
      for (int q = 0; q < D; ++q) ring[q] = frag(q);
    }
#pragma unroll
    for (int bk = 0; bk < NB; ++bk) {
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (init_from_slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] = part[bk][w][j][g * 4 + e][r];
      }
      if (live) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int q = bk * NF + f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q % D], bw[j][f], acc[j], 0, 0, 0);
          if (q + D < Q) ring[q % D] = frag(q + D);
        }
      }
      // D[i = batch][j = unit]: lane holds unit r, batches 4g..4g+3
      if (live || !init_from_slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) part[bk][w][j][g * 4 + e][r] = acc[j][e];
      }
    }
  };

  // cell-update thread (block group tg, b within block, unit)
  const int tg = threadIdx.x >> 8, bb = (threadIdx.x >> 4) & 15, uu = threadIdx.x & 15;
  const int u = u0 + uu;
  float c_reg[NU];
  float gin[NU][4];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    c_reg[i] = 0.f;
    const int bk = tg + 2 * i;
    const bool upd = bk < NB && bk * 16 + bb < B;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gin[i][q] = 0.f;
      if (upd) gin[i][q] = l > 0 ? p.bias[l][q * H + u] : p.gi0[((long long)(bk * 16 + bb) * L) * 4 * H + q * H + u];
    }
  }
  const int nstage = L + 2 * top;
  bool alive = true;
  for (int s = 0; s < nstage; ++s) {
    const int t = s - 2 * l;
    const bool active = t >= 0 && t < L;
    float st_i[NU], st_f[NU], st_g[NU], st_o[NU], st_h[NU];   // this step's state, stored after the arrival
    if (active) {
      dot_half(hx, t - 1, brec, has_in, t > 0);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int bk = tg + 2 * i;
        if (bk < NB && bk * 16 + bb < B) {
          const long long row = (long long)(bk * 16 + bb) * L + t;
          float pre[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = gin[i][q];
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) v += part[bk][ww][q][bb][uu];
            pre[q] = v;
          }
          st_i[i] = nsp_sigmoid(pre[0]);
          st_f[i] = nsp_sigmoid(pre[1]);
          st_g[i] = nsp_tanh(pre[2]);
          st_o[i] = nsp_sigmoid(pre[3]);
          c_reg[i] = st_f[i] * c_reg[i] + st_i[i] * st_g[i];
          const float h = st_o[i] * nsp_tanh(c_reg[i]);
          st_h[i] = h;
          hs[bk][0][bb][uu] = (__bf16)h;
          if (l < top) {
            float hd = h;
            if (p.dropout_p > 0.f)
              hd *= nsp_keep_scale(p.seed[l], p.offset[l] + (unsigned long long)(row * H + u), p.dropout_p);
            hs[bk][1][bb][uu] = (__bf16)hd;
          }
          if (l == 0 && t + 1 < L) {   // next step's input projection (plain data from an earlier kernel)
            const float* gnext = p.gi0 + (row + 1) * 4 * H;
#pragma unroll
            for (int q = 0; q < 4; ++q) gin[i][q] = gnext[q * H + u];
          }
        }
      }
      __syncthreads();
      // publish h and dropout(h) into the exchange buffers with 8-byte write-through stores: a workgroup's 16 units
      // are two whole 256-byte pieces (16 utterances x 8 units) per batch block
      if (threadIdx.x < 64 * NB && !(NSP_LSTM_ABL & 16)) {
        const int bk = threadIdx.x >> 6, pb = (threadIdx.x >> 2) & 15, pq = threadIdx.x & 3;
        const long long xo = (long long)(t * NB + bk) * H * 16 + (((u0 >> 3) + (pq >> 1)) * 16 + pb) * 8 + (pq & 1) * 4;
        if (t + 1 < L)
          __hip_atomic_store((gu64*)(hx + xo), *reinterpret_cast<const unsigned long long*>(&hs[bk][0][pb][pq * 4]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (l < top)
          __hip_atomic_store((gu64*)(ydx + xo), *reinterpret_cast<const unsigned long long*>(&hs[bk][1][pb][pq * 4]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const bool more = s + 1 < nstage;
#if !(NSP_LSTM_ABL & 8)
    if (more) grid_arrive(sync);
#endif
    // the NEXT step's input half: dropout(h_{l-1}) at t + 1 was published one stage ago
    if (has_in && more && t + 1 >= 0 && t + 1 < L) dot_half(inx, t + 1, bin, false, true);
    if (active && !(NSP_LSTM_ABL & 4)) {
      // this step's saved state, after the arrival and after the input half's loads (vmcnt is in-order: in front of the
      // barrier's drain, or of those loads, ~20 KB of scattered 4-byte stores cost 2 - 6 us per stage)
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int bk = tg + 2 * i;
        if (bk < NB && bk * 16 + bb < B) {
          const long long row = (long long)(bk * 16 + bb) * L + t;
          p.c_all[l][row * H + u] = c_reg[i];
          float* gs = p.gates[l] + row * 4 * H;
          gs[u] = st_i[i]; gs[H + u] = st_f[i]; gs[2 * H + u] = st_g[i]; gs[3 * H + u] = st_o[i];
          if (l == top) p.y_top[row * H + u] = st_h[i];
        }
      }
      // the [B, L, H] images the weight-gradient GEMMs read after this kernel (hs is not touched again before the
      // next stage's cell update)
      if (threadIdx.x < 64 * NB) {
        const int bk = threadIdx.x >> 6, pb = (threadIdx.x >> 2) & 15, pq = threadIdx.x & 3;
        if (bk * 16 + pb < B) {
          const long long prow = (long long)(bk * 16 + pb) * L + t;
          if (t + 1 < L)
            *reinterpret_cast<unsigned long long*>(reinterpret_cast<__bf16*>(p.hp16[l]) + (prow + 1) * H + u0 + pq * 4) =
                *reinterpret_cast<const unsigned long long*>(&hs[bk][0][pb][pq * 4]);
          if (t == 0)
            *reinterpret_cast<unsigned long long*>(reinterpret_cast<__bf16*>(p.hp16[l]) + prow * H + u0 + pq * 4) = 0ull;
          if (l < top)
            *reinterpret_cast<unsigned long long*>(reinterpret_cast<__bf16*>(p.yd16[l]) + prow * H + u0 + pq * 4) =
                *reinterpret_cast<const unsigned long long*>(&hs[bk][1][pb][pq * 4]);
        }
      }
    }
#if !(NSP_LSTM_ABL & 8)
    if (more && alive) alive = grid_wait(sync, (unsigned int)(s + 1) * nwg, &dead_sh);
#else
    __syncthreads();
#endif
  }
  if (!alive && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.y_top[0] = __builtin_nanf("");
}

// backward.  grid: (H/16, nl); block 512 = 8 waves, each holding its 1/8 of the reduction over the workgroup's 16
// rows of W_hh^T (brec, K = 4H) and, below the top layer, of W_ih_{l+1}^T (bext, K = 4H): NF = H / 64 fragments per
// wave and half.  Stage s runs step t = L - 1 - (s - 2 (top - l)) of layer l.  LDS slots per batch block: 0..7 the
// waves' "ext" partial sums (written one stage early, see above), 8..15 the recurrent ones.
template <int NB, int NF>
__global__ __launch_bounds__(512) void lstm_stack_bwd_persistent_kernel(const nsp_lstm_stack_params p,
                                                                        unsigned int* sync) {
  extern __shared__ __attribute__((aligned(16))) float bdyn[];
  typedef float part_t[16][16][17];
  part_t* part = reinterpret_cast<part_t*>(bdyn);                                  // [NB]
  typedef __bf16 dss_t[4][16][16];
  dss_t* dss = reinterpret_cast<dss_t*>(bdyn + NB * (16 * 16 * 17));               // [NB]
  __shared__ int dead_sh;
  constexpr int NU = (NB + 1) / 2;
  constexpr int Q = NB * NF;
  constexpr int D = Q < NSP_LSTM_BWD_DEPTH ? Q : NSP_LSTM_BWD_DEPTH;   // operand fragments in flight per wave
  const int top = p.nl - 1;
  const int l = blockIdx.y;
  const int H = p.H, L = p.L, B = p.B;
  const int nwg = gridDim.x * gridDim.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * 16;
  const int K4 = 4 * H;
  const bool has_ext = l < top;
  const int Ktot = has_ext ? 2 * K4 : K4;
  const int koff = w * (NF * 32) + g * 8;
  bf16x8 bext[NF], brec[NF];
  {
    const __bf16* wr = reinterpret_cast<const __bf16*>(p.w[l]) + (long long)(u0 + r) * Ktot + koff;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      brec[f] = *reinterpret_cast<const bf16x8*>(wr + (has_ext ? K4 : 0) + f * 32);
      bext[f] = *reinterpret_cast<const bf16x8*>(wr + f * 32);   // top layer: never used
    }
  }
  // exchange buffer of layer l: slot tau holds its dgates at step tau (see lstm_xchg_index)
  __bf16* dgx = reinterpret_cast<__bf16*>(p.xchg[l]);
  const __bf16* extx = has_ext ? reinterpret_cast<const __bf16*>(p.xchg[l + 1]) : dgx;
  const int lane_off = ((koff >> 3) * 16 + r) * 8;

  // part[block][slot0 + w] = A(16 utterances x this wave's k-slice at time row trow) . W^T, a rolling ring of D loads
  auto dot_half = [&](const __bf16* xbuf, int trow, const bf16x8 (&bw)[NF], int slot0, bool live) {
    bf16x8 ring[D];
    const __bf16* blk0 = xbuf + (long long)trow * NB * K4 * 16;   // uniform; + block * K4 * 16 + 32-bit lane offset + immediate
    auto frag = [&](int q) {
#if NSP_LSTM_ABL & 1
      return bw[q % NF];
#else
      return *reinterpret_cast<const bf16x8*>(blk0 + (long long)(q / NF) * K4 * 16 + (lane_off + (q % NF) * 512));
#endif
    };
    if (live) {
#pragma unroll
      for (int q = 0; q < D; ++q) ring[q] = frag(q);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (live) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q % D], bw[q % NF], acc, 0, 0, 0);
        if (q + D < Q) ring[q % D] = frag(q + D);
      }
      if (q % NF == NF - 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) part[q / NF][slot0 + w][g * 4 + e][r] = acc[e];
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  const int tg = threadIdx.x >> 8, bb = (threadIdx.x >> 4) & 15, uu = threadIdx.x & 15;
  const int u = u0 + uu;
  bool upd[NU];
  long long brow[NU];
  float dc_reg[NU];
  // the cell-update inputs of the NEXT stage (saved by the forward pass: plain data) are fetched a stage ahead
  float n_dy[NU], n_ig[NU], n_fg[NU], n_gg[NU], n_og[NU], n_cp[NU], n_c[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int bk = tg + 2 * i;
    upd[i] = bk < NB && bk * 16 + bb < B;
    brow[i] = (long long)(bk * 16 + bb) * L;
    dc_reg[i] = 0.f;
    n_dy[i] = n_ig[i] = n_fg[i] = n_gg[i] = n_og[i] = n_cp[i] = n_c[i] = 0.f;
    if (upd[i]) n_c[i] = p.c_all[l][(brow[i] + L - 1) * H + u];
  }
  auto fetch = [&](int tn) {
    if (tn < 0 || (NSP_LSTM_ABL & 64)) return;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if (!upd[i]) continue;
      const long long rown = brow[i] + tn;
      const float* gs = p.gates[l] + rown * K4;
      if (!has_ext) n_dy[i] = p.dy_top[rown * H + u];
      n_ig[i] = gs[u]; n_fg[i] = gs[H + u]; n_gg[i] = gs[2 * H + u]; n_og[i] = gs[3 * H + u];
      n_cp[i] = tn > 0 ? p.c_all[l][(rown - 1) * H + u] : 0.f;
    }
  };
  fetch(L - 1);
  const int lag = 2 * (top - l);
  const int nstage = L + 2 * top;
  bool alive = true;
  for (int s = 0; s < nstage; ++s) {
    const int t = L - 1 - (s - lag);
    const bool active = t >= 0 && t < L;
    if (active) {
      float dyv[NU], ig[NU], fg[NU], gg[NU], og[NU], cp[NU], c[NU], keep[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        dyv[i] = n_dy[i]; ig[i] = n_ig[i]; fg[i] = n_fg[i]; gg[i] = n_gg[i]; og[i] = n_og[i]; cp[i] = n_cp[i]; c[i] = n_c[i];
        keep[i] = 1.f;
        if (upd[i] && has_ext && p.dropout_p > 0.f)
          keep[i] = nsp_keep_scale(p.seed[l], p.offset[l] + (unsigned long long)((brow[i] + t) * H + u), p.dropout_p);
        n_c[i] = cp[i];          // c_{t-1} is this step's c_prev
      }
      fetch(t - 1);
      dot_half(dgx, t + 1, brec, 8, t + 1 < L);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (!upd[i]) continue;
        const int bk = tg + 2 * i;
        float ext = 0.f, rec = 0.f;
        if (has_ext) {
#pragma unroll
          for (int q = 0; q < 8; ++q) ext += part[bk][q][bb][uu];
        }
#pragma unroll
        for (int q = 8; q < 16; ++q) rec += part[bk][q][bb][uu];
        const float dh = has_ext ? keep[i] * ext + rec : dyv[i] + rec;
        const float tc = nsp_tanh(c[i]);
        const float dct = dc_reg[i] + dh * og[i] * (1.f - tc * tc);
        dss[bk][3][bb][uu] = (__bf16)(dh * tc * og[i] * (1.f - og[i]));
        dss[bk][0][bb][uu] = (__bf16)(dct * gg[i] * ig[i] * (1.f - ig[i]));
        dss[bk][1][bb][uu] = (__bf16)(dct * cp[i] * fg[i] * (1.f - fg[i]));
        dss[bk][2][bb][uu] = (__bf16)(dct * ig[i] * (1.f - gg[i] * gg[i]));
        dc_reg[i] = dct * fg[i];
      }
      __syncthreads();
      // (block, gate q, batch pb, 4-unit group pq): one 8-byte write-through store
#pragma unroll
      for (int k = 0; k < (NB * 256 + 511) / 512; ++k) {
        const int idx = threadIdx.x + 512 * k;
        const int bk = idx >> 8, q = (idx >> 6) & 3, pb = (idx >> 2) & 15, pq = idx & 3;
        if (bk < NB && !(NSP_LSTM_ABL & 16))
          __hip_atomic_store((gu64*)(dgx + (long long)(t * NB + bk) * K4 * 16 + ((((q * H + u0) >> 3) + (pq >> 1)) * 16 + pb) * 8 + (pq & 1) * 4),
                             *reinterpret_cast<const unsigned long long*>(&dss[bk][q][pb][pq * 4]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const bool more = s + 1 < nstage;
#if !(NSP_LSTM_ABL & 8)
    if (more) grid_arrive(sync);
#endif
    if (active && !(NSP_LSTM_ABL & 32)) {   // the [B, L, 4H] image the weight / input gradient GEMMs read after this kernel
#pragma unroll
      for (int k = 0; k < (NB * 256 + 511) / 512; ++k) {
        const int idx = threadIdx.x + 512 * k;
        const int bk = idx >> 8, q = (idx >> 6) & 3, pb = (idx >> 2) & 15, pq = idx & 3;
        if (bk < NB && bk * 16 + pb < B)
          *reinterpret_cast<unsigned long long*>(reinterpret_cast<__bf16*>(p.dg16[l]) + ((long long)(bk * 16 + pb) * L + t) * K4 + q * H + u0 + pq * 4) =
              *reinterpret_cast<const unsigned long long*>(&dss[bk][q][pb][pq * 4]);
      }
    }
    // the NEXT step's "ext" half: layer l+1's dgates at t - 1 were published one stage ago
    if (has_ext && more && t - 1 >= 0 && t - 1 < L) dot_half(extx, t - 1, bext, 0, true);
#if !(NSP_LSTM_ABL & 8)
    if (more && alive) alive = grid_wait(sync, (unsigned int)(s + 1) * nwg, &dead_sh);
#else
    __syncthreads();
#endif
  }
  if (!alive && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    reinterpret_cast<__bf16*>(p.dg16[0])[0] = (__bf16)__builtin_nanf("");
}

}  // namespace

// Runs all L steps.  Whh: [4H,H] (bf16 shadow in NSP_COMPUTE_BF16, fp32 in NSP_COMPUTE_F32);
// yshadow: bf16 [B,L,H] written alongside y (mode bf16; pass y itself in mode f32).
// Steps t_begin .. t_end-1 only (the same step kernels, nothing else changes): with t_begin = 1 the step reads
// its previous state from row 0 of yshadow / c_all, which the caller has filled -- an LSTM with a given
// INITIAL STATE (encoders/rnn.py:466-475: the forward direction of the latency-controlled BLSTM carries its state
// from chunk to chunk).
extern "C" int nsp_lstm_fwd_range(const float* gi, const void* Whh, float* y, void* yshadow, float* c_all,
                                  float* gates, int B, int L, int H, int mode, int t_begin, int t_end,
                                  void* stream) {
  if (B <= 0 || L <= 0 || H <= 0 || H % 16) return NSP_EUNSUPPORTED;
  if (mode == NSP_COMPUTE_BF16 && H % 8) return NSP_EUNSUPPORTED;
  if (t_begin < 0 || t_end > L || t_begin > t_end) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(H / 4, nsp_cdiv(B, 16)), block(256);
  const int Kw = ((H + 3) / 4 + 7) / 8 * 8;  // per-wave slice of the reduction (whole bf16x8 chunks)
  for (int t = t_begin; t < t_end; ++t) {
    if (mode == NSP_COMPUTE_BF16)
      hipLaunchKernelGGL((lstm_step_fwd_kernel<0>), grid, block, 0, st, gi, Whh, y, yshadow, c_all, gates, B, L, H, t, Kw);
    else
      hipLaunchKernelGGL((lstm_step_fwd_kernel<1>), grid, block, 0, st, gi, Whh, y, (void*)y, c_all, gates, B, L, H, t, Kw);
  }
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_fwd(const float* gi, const void* Whh, float* y, void* yshadow, float* c_all,
                            float* gates, int B, int L, int H, int mode, void* stream) {
  return nsp_lstm_fwd_range(gi, Whh, y, yshadow, c_all, gates, B, L, H, mode, 0, L, stream);
}

// dy: [B,L,H] gradient w.r.t. the outputs; produces dgates [B,L,4H] (+ bf16 shadow) from which
// the caller derives dx, dW_ih, dW_hh, db with GEMMs.  dc: [B,H] scratch.
// Steps t_end-1 down to t_begin.  With t_end < L the first step executed reads the incoming cell-state gradient
// from `dc` and the recurrent term from dgates[:, t_end] (the caller fills both: the gradient w.r.t. the FINAL
// state); after the last step `dc` holds the gradient w.r.t. the cell state BEFORE step t_begin.
extern "C" int nsp_lstm_bwd_range(const float* dy, const void* WhhT, const float* c_all, const float* gates,
                                  float* dgates, void* dgshadow, float* dc, int B, int L, int H, int mode,
                                  int t_begin, int t_end, void* stream) {
  if (B <= 0 || L <= 0 || H <= 0 || H % 16) return NSP_EUNSUPPORTED;
  if (t_begin < 0 || t_end > L || t_begin > t_end) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(H / 16, nsp_cdiv(B, 16)), block(1024);
  const int Kw = ((4 * H + 15) / 16 + 7) / 8 * 8;
  for (int t = t_end - 1; t >= t_begin; --t) {
    if (mode == NSP_COMPUTE_BF16)
      hipLaunchKernelGGL((lstm_step_bwd_kernel<0>), grid, block, 0, st, dy, WhhT, c_all, gates, dgates, dgshadow, dc, B, L, H, t, Kw);
    else
      hipLaunchKernelGGL((lstm_step_bwd_kernel<1>), grid, block, 0, st, dy, WhhT, c_all, gates, dgates, (void*)dgates, dc, B, L, H, t, Kw);
  }
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_bwd(const float* dy, const void* WhhT, const float* c_all, const float* gates,
                            float* dgates, void* dgshadow, float* dc, int B, int L, int H, int mode,
                            void* stream) {
  return nsp_lstm_bwd_range(dy, WhhT, c_all, gates, dgates, dgshadow, dc, B, L, H, mode, 0, L, stream);
}

static int lstm_stack_check(const nsp_lstm_stack_params* p) {
  if (!p || p->nl < 1 || p->nl > NSP_LSTM_MAX_LAYERS) return NSP_EINVAL;
  if (p->B <= 0 || p->L <= 0 || p->H <= 0 || p->H % 64) return NSP_EUNSUPPORTED;
  return NSP_OK;
}

extern "C" int nsp_lstm_stack_fwd(const nsp_lstm_stack_params* p, void* stream) {
  int rc = lstm_stack_check(p);
  if (rc != NSP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(p->H / 4, p->nl, nsp_cdiv(p->B, 16)), block(512);
  for (int s = 0; s < p->L + p->nl - 1; ++s)
    hipLaunchKernelGGL(lstm_stack_fwd_kernel, grid, block, 0, st, *p, s);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_stack_bwd(const nsp_lstm_stack_params* p, void* stream) {
  int rc = lstm_stack_check(p);
  if (rc != NSP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(p->H / 16, p->nl, nsp_cdiv(p->B, 16)), block(1024);
  for (int s = 0; s < p->L + p->nl - 1; ++s)
    hipLaunchKernelGGL(lstm_stack_bwd_kernel, grid, block, 0, st, *p, s);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

static int lstm_persistent_check(const nsp_lstm_stack_params* p) {
  int rc = lstm_stack_check(p);
  if (rc != NSP_OK) return rc;
  if (p->B > 64 || p->H % 256 || p->H > 1024 || p->nl * (p->H / 16) > 256) return NSP_EUNSUPPORTED;
  for (int l = 0; l < p->nl; ++l)
    if (!p->xchg[l]) return NSP_EINVAL;   // the hand-over scratch is the caller's (sizes: include/nsp_hip.h)
  return NSP_OK;
}

// The grid barrier needs every workgroup of the launch co-resident: refuse (NSP_EUNSUPPORTED, the
// caller falls back to one launch per stage) unless the device could hold the whole grid even if
// nothing else were running -- a CU-masked / partitioned device or an oversized grid fails here
// instead of timing out inside the kernel.  (Transient crowding by co-running kernels only delays
// residency; a genuine timeout raises the `dead` word sync[1], which the host checks.)
static bool lstm_grid_fits(const void* kernel, int block, size_t shmem, int nwg) {
  // capacity (workgroups the idle device holds) cached per (kernel, dynamic LDS size)
  static struct { const void* k; size_t sh; long long cap; } cache[16];
  static int ncache = 0;
  for (int i = 0; i < ncache; ++i)
    if (cache[i].k == kernel && cache[i].sh == shmem) return cache[i].cap >= nwg;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, shmem) != hipSuccess) return false;
  const long long cap = (long long)per_cu * cus;
  if (ncache < 16) { cache[ncache].k = kernel; cache[ncache].sh = shmem; cache[ncache].cap = cap; ++ncache; }
  return cap >= nwg;
}

template <typename KernelT>
static int lstm_launch_persistent(KernelT kernel, dim3 grid, int block, size_t shmem, const nsp_lstm_stack_params* p,
                                  unsigned int* sync, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (!lstm_grid_fits((const void*)kernel, block, shmem, grid.x * grid.y)) return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(kernel, grid, dim3(block), shmem, st, *p, sync);
  return NSP_OK;
}

// NB in {1, 2, 4} batch blocks (3 runs as 4: rows beyond B repeat row B - 1 and are never stored), HB = H / 256
#define LSTM_P_CASE(KERNEL, NBT, HB, FMUL)                                                                   \
  case NBT * 8 + HB:                                                                                          \
    rc = lstm_launch_persistent(KERNEL<NBT, HB * FMUL>, grid, 512, (size_t)NBT * shmem_per_block, p, sync, st); \
    break;
#define LSTM_P_DISPATCH(KERNEL, FMUL)                                                                         \
  do {                                                                                                        \
    const int nb_ = nsp_cdiv(p->B, 16), nbt = nb_ <= 1 ? 1 : (nb_ == 2 ? 2 : 4);                              \
    switch (nbt * 8 + p->H / 256) {                                                                           \
      LSTM_P_CASE(KERNEL, 1, 1, FMUL) LSTM_P_CASE(KERNEL, 1, 2, FMUL) LSTM_P_CASE(KERNEL, 1, 3, FMUL)         \
      LSTM_P_CASE(KERNEL, 1, 4, FMUL) LSTM_P_CASE(KERNEL, 2, 1, FMUL) LSTM_P_CASE(KERNEL, 2, 2, FMUL)         \
      LSTM_P_CASE(KERNEL, 2, 3, FMUL) LSTM_P_CASE(KERNEL, 2, 4, FMUL) LSTM_P_CASE(KERNEL, 4, 1, FMUL)         \
      LSTM_P_CASE(KERNEL, 4, 2, FMUL) LSTM_P_CASE(KERNEL, 4, 3, FMUL) LSTM_P_CASE(KERNEL, 4, 4, FMUL)         \
      default: rc = NSP_EUNSUPPORTED;                                                                         \
    }                                                                                                         \
  } while (0)

extern "C" int nsp_lstm_stack_fwd_persistent(const nsp_lstm_stack_params* p, unsigned int* sync, void* stream) {
  int rc = lstm_persistent_check(p);
  if (rc != NSP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(sync, 0, 2 * sizeof(unsigned int), st);
  const dim3 grid(p->H / 16, p->nl);
  const size_t shmem_per_block = sizeof(float) * 8 * 4 * 16 * 17 + 2 * 16 * 16 * 2;
  LSTM_P_DISPATCH(lstm_stack_fwd_persistent_kernel, 1);
  if (rc != NSP_OK) return rc;
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_stack_bwd_persistent(const nsp_lstm_stack_params* p, unsigned int* sync, void* stream) {
  int rc = lstm_persistent_check(p);
  if (rc != NSP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(sync, 0, 2 * sizeof(unsigned int), st);
  const dim3 grid(p->H / 16, p->nl);
  const size_t shmem_per_block = sizeof(float) * 16 * 16 * 17 + 4 * 16 * 16 * 2;
  LSTM_P_DISPATCH(lstm_stack_bwd_persistent_kernel, 4);
  if (rc != NSP_OK) return rc;
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
#undef LSTM_P_DISPATCH
#undef LSTM_P_CASE
