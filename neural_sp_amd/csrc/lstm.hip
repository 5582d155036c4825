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

// forward.  grid: (H/16, nl); block 512.  Workgroup (ub, l) owns hidden units 16*ub.. of layer l:
// wave w holds, for each of the 4 gates, the 16 W rows of those units restricted to its 1/8 of the
// reduction (layer 0: K = H; layers >= 1: K = 2H over [W_ih | W_hh]) as MFMA B-fragments.
// NB = ceil(B/16) batch blocks share a stage: all their partial products go to LDS first (their
// operand loads are independent, so the latencies overlap), ONE workgroup barrier, then the two
// halves of the workgroup run the cell updates of alternating blocks, one barrier, publish.
template <int NB>
__global__ __launch_bounds__(512) void lstm_stack_fwd_persistent_kernel(const nsp_lstm_stack_params p,
                                                                        unsigned int* sync) {
  extern __shared__ __attribute__((aligned(16))) float fdyn[];
  typedef float part_t[8][4][16][17];
  part_t* part = reinterpret_cast<part_t*>(fdyn);                                   // [NB]
  typedef __bf16 hs_t[2][16][16];
  hs_t* hs = reinterpret_cast<hs_t*>(fdyn + NB * (8 * 4 * 16 * 17));                // [NB]
  __shared__ int dead_sh;
  constexpr int NU = (NB + 1) / 2;   // blocks per cell-update thread (thread group tg: blocks tg, tg+2)
  const int l = blockIdx.y;
  const int H = p.H, L = p.L, B = p.B, top = p.nl - 1;
  const int nwg = gridDim.x * gridDim.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int u0 = blockIdx.x * 16;
  const int Ktot = l == 0 ? H : 2 * H;
  const int Kw = Ktot >> 3;
  const int nf = Kw >> 5;  // 32-wide k-fragments per wave: <= 8
  const int kbeg = w * Kw;
  bf16x8 bfrag[4][8];
  {
    const __bf16* W = reinterpret_cast<const __bf16*>(p.w[l]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int f = 0; f < 8; ++f)
        if (f < nf)
          bfrag[j][f] = *reinterpret_cast<const bf16x8*>(W + (long long)(j * H + u0 + r) * Ktot + kbeg + f * 32 + g * 8);
  }
  const __bf16* abase0;
  bool a_rec;  // the wave's k-slice lies in the recurrent half
  if (l == 0 || kbeg >= H) {
    abase0 = reinterpret_cast<const __bf16*>(p.hp16[l]) + (l == 0 ? kbeg : kbeg - H);
    a_rec = true;
  } else {
    abase0 = reinterpret_cast<const __bf16*>(p.yd16[l - 1]) + kbeg;
    a_rec = false;
  }
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
  const int nstage = L + p.nl - 1;
  bool alive = true;
  for (int s = 0; s < nstage; ++s) {
    const int t = s - l;
    if (t >= 0 && t < L) {
#pragma unroll
      for (int bk = 0; bk < NB; ++bk) {
        const int b0 = bk * 16;
        const bool a_valid = b0 + r < B;
        const long long arow0 = (long long)min(b0 + r, B - 1) * L;
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!a_rec || t > 0) {
          const __bf16* ap = abase0 + (arow0 + t) * H + g * 8;
          bf16x8 af[8];
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
            af[f] = (f < nf && a_valid) ? *reinterpret_cast<const bf16x8*>(ap + f * 32) : z;
          }
#pragma unroll
          for (int f = 0; f < 8; ++f)
            if (f < nf) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f], bfrag[j][f], acc[j], 0, 0, 0);
            }
        }
        // D[i = batch][j = unit]: lane holds unit r, batches 4g..4g+3
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) part[bk][w][j][g * 4 + e][r] = acc[j][e];
      }
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
          const float ig = nsp_sigmoid(pre[0]);
          const float fg = nsp_sigmoid(pre[1]);
          const float gg = nsp_tanh(pre[2]);
          const float og = nsp_sigmoid(pre[3]);
          c_reg[i] = fg * c_reg[i] + ig * gg;
          const float h = og * nsp_tanh(c_reg[i]);
          p.c_all[l][row * H + u] = c_reg[i];
          float* gs = p.gates[l] + row * 4 * H;
          gs[u] = ig; gs[H + u] = fg; gs[2 * H + u] = gg; gs[3 * H + u] = og;
          hs[bk][0][bb][uu] = (__bf16)h;
          if (l == top) {
            p.y_top[row * H + u] = h;
          } else {
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
      // publish h (shifted by one step) and dropout(h) with 8-byte write-through stores
      if (threadIdx.x < 64 * NB) {
        const int bk = threadIdx.x >> 6, pb = (threadIdx.x >> 2) & 15, pq = threadIdx.x & 3;
        if (bk * 16 + pb < B) {
          const long long prow = (long long)(bk * 16 + pb) * L + t;
          if (t + 1 < L)
            __hip_atomic_store((gu64*)(reinterpret_cast<__bf16*>(p.hp16[l]) + (prow + 1) * H + u0 + pq * 4),
                               *reinterpret_cast<const unsigned long long*>(&hs[bk][0][pb][pq * 4]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          if (t == 0)
            *reinterpret_cast<unsigned long long*>(reinterpret_cast<__bf16*>(p.hp16[l]) + prow * H + u0 + pq * 4) = 0ull;
          if (l < top)
            __hip_atomic_store((gu64*)(reinterpret_cast<__bf16*>(p.yd16[l]) + prow * H + u0 + pq * 4),
                               *reinterpret_cast<const unsigned long long*>(&hs[bk][1][pb][pq * 4]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (s + 1 < nstage) {
      grid_arrive(sync);
      if (alive) alive = grid_wait(sync, (unsigned int)(s + 1) * nwg, &dead_sh);
    }
  }
  if (!alive && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.y_top[0] = __builtin_nanf("");
}

// backward.  grid: (H/16, nl); block 1024 = 16 waves, each holding its 1/16 of the reduction of the
// workgroup's 16 rows of W_hh^T (top layer, K = 4H) or [W_ih_{l+1}^T | W_hh_l^T] (K = 8H).
// NB <= 4 batch blocks per stage: the dot products of all blocks go to LDS, then thread group
// tg = tid/256 runs the cell update of block tg (so every update thread owns ONE (block, b, unit)
// for the whole sequence: dc and the prefetched inputs of the next stage stay in registers).
template <int NB>
__global__ __launch_bounds__(1024) void lstm_stack_bwd_persistent_kernel(const nsp_lstm_stack_params p,
                                                                         unsigned int* sync) {
  extern __shared__ __attribute__((aligned(16))) float bdyn[];
  typedef float part_t[16][16][17];
  part_t* part = reinterpret_cast<part_t*>(bdyn);                                  // [NB]
  typedef __bf16 dss_t[4][16][16];
  dss_t* dss = reinterpret_cast<dss_t*>(bdyn + NB * (16 * 16 * 17));               // [NB]
  __shared__ int dead_sh;
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
  const int Kw = Ktot >> 4;
  const int nf = Kw >> 5;  // <= 16
  const int kbeg = w * Kw;
  bf16x8 bfrag[16];
  {
    const __bf16* W = reinterpret_cast<const __bf16*>(p.w[l]) + (long long)(u0 + r) * Ktot + kbeg + g * 8;
#pragma unroll
    for (int f = 0; f < 16; ++f)
      if (f < nf) bfrag[f] = *reinterpret_cast<const bf16x8*>(W + f * 32);
  }
  const bool a_ext = has_ext && kbeg < K4;
  const __bf16* abase0 = a_ext ? reinterpret_cast<const __bf16*>(p.dg16[l + 1]) + kbeg
                               : reinterpret_cast<const __bf16*>(p.dg16[l]) + (has_ext ? kbeg - K4 : kbeg);
  const int tg = threadIdx.x >> 8, bb = (threadIdx.x >> 4) & 15, uu = threadIdx.x & 15;
  const int u = u0 + uu;
  const bool upd = tg < NB && tg * 16 + bb < B;
  const long long brow = (long long)(tg * 16 + bb) * L;
  float dc_reg = 0.f;
  const int nstage = L + p.nl - 1;
  bool alive = true;
  // the cell-update inputs of the NEXT stage (saved by the forward pass: plain data) are fetched
  // before the grid barrier, so their latency hides behind it
  float n_dy = 0.f, n_ig = 0.f, n_fg = 0.f, n_gg = 0.f, n_og = 0.f, n_cp = 0.f, n_c = 0.f;
  auto fetch = [&](int tn) {
    if (!upd || tn < 0) return;
    const long long rown = brow + tn;
    const float* gs = p.gates[l] + rown * K4;
    if (!has_ext) n_dy = p.dy_top[rown * H + u];
    n_ig = gs[u]; n_fg = gs[H + u]; n_gg = gs[2 * H + u]; n_og = gs[3 * H + u];
    n_cp = tn > 0 ? p.c_all[l][(rown - 1) * H + u] : 0.f;
  };
  if (upd) n_c = p.c_all[l][(brow + L - 1) * H + u];
  fetch(L - 1);
  for (int s = 0; s < nstage; ++s) {
    const int t = L - 1 - (s - (top - l));
    if (t >= 0 && t < L) {
      const long long row = brow + t;
      const float dyv = n_dy, ig = n_ig, fg = n_fg, gg = n_gg, og = n_og, cp = n_cp, c = n_c;
      float keep = 1.f;
      if (upd && has_ext && p.dropout_p > 0.f)
        keep = nsp_keep_scale(p.seed[l], p.offset[l] + (unsigned long long)(row * H + u), p.dropout_p);
      n_c = cp;          // c_{t-1} is this step's c_prev
      fetch(t - 1);
#pragma unroll
      for (int bk = 0; bk < NB; ++bk) {
        const bool a_valid = bk * 16 + r < B;
        const long long arow0 = (long long)min(bk * 16 + r, B - 1) * L;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (a_ext || t + 1 < L) {
          const __bf16* ap = abase0 + (arow0 + t + (a_ext ? 0 : 1)) * K4 + g * 8;
          // four rounds of 4 fragments: 4 waves per SIMD leave 128 VGPRs per wave, 64 hold weights
#pragma unroll
          for (int h2 = 0; h2 < 4; ++h2) {
            bf16x8 af[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              bf16x8 z;
#pragma unroll
              for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
              af[f] = (h2 * 4 + f < nf && a_valid) ? *reinterpret_cast<const bf16x8*>(ap + (h2 * 4 + f) * 32) : z;
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
              if (h2 * 4 + f < nf) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f], bfrag[h2 * 4 + f], acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) part[bk][w][g * 4 + e][r] = acc[e];
      }
      __syncthreads();
      if (upd) {
        float ext = 0.f, rec = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) ext += part[tg][q][bb][uu];
#pragma unroll
        for (int q = 8; q < 16; ++q) rec += part[tg][q][bb][uu];
        const float dh = has_ext ? keep * ext + rec : dyv + ext + rec;
        const float tc = nsp_tanh(c);
        const float dct = dc_reg + dh * og * (1.f - tc * tc);
        dss[tg][3][bb][uu] = (__bf16)(dh * tc * og * (1.f - og));
        dss[tg][0][bb][uu] = (__bf16)(dct * gg * ig * (1.f - ig));
        dss[tg][1][bb][uu] = (__bf16)(dct * cp * fg * (1.f - fg));
        dss[tg][2][bb][uu] = (__bf16)(dct * ig * (1.f - gg * gg));
        dc_reg = dct * fg;
      }
      __syncthreads();
      {   // (block tg, gate q, batch pb, 4-unit group pq): one 8-byte write-through store
        const int q = (threadIdx.x >> 6) & 3, pb = (threadIdx.x >> 2) & 15, pq = threadIdx.x & 3;
        if (tg < NB && tg * 16 + pb < B)
          __hip_atomic_store((gu64*)(reinterpret_cast<__bf16*>(p.dg16[l]) + ((long long)(tg * 16 + pb) * L + t) * K4 + q * H + u0 + pq * 4),
                             *reinterpret_cast<const unsigned long long*>(&dss[tg][q][pb][pq * 4]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (s + 1 < nstage) {
      grid_arrive(sync);
      if (alive) alive = grid_wait(sync, (unsigned int)(s + 1) * nwg, &dead_sh);
    }
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

extern "C" int nsp_lstm_stack_fwd_persistent(const nsp_lstm_stack_params* p, unsigned int* sync, void* stream) {
  int rc = lstm_persistent_check(p);
  if (rc != NSP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(sync, 0, 2 * sizeof(unsigned int), st);
  const dim3 grid(p->H / 16, p->nl);
  const int nb = nsp_cdiv(p->B, 16);
  const size_t shmem = (size_t)nb * (sizeof(float) * 8 * 4 * 16 * 17 + 2 * 16 * 16 * 2);
#define LSTM_FWD_P(N)                                                                                         \
  do {                                                                                                        \
    (void)hipFuncSetAttribute((const void*)lstm_stack_fwd_persistent_kernel<N>,                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                        \
    if (!lstm_grid_fits((const void*)lstm_stack_fwd_persistent_kernel<N>, 512, shmem, grid.x * grid.y))       \
      return NSP_EUNSUPPORTED;                                                                                \
    hipLaunchKernelGGL((lstm_stack_fwd_persistent_kernel<N>), grid, dim3(512), shmem, st, *p, sync);          \
  } while (0)
  switch (nb) {
    case 1: LSTM_FWD_P(1); break;
    case 2: LSTM_FWD_P(2); break;
    case 3: LSTM_FWD_P(3); break;
    default: LSTM_FWD_P(4); break;
  }
#undef LSTM_FWD_P
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_stack_bwd_persistent(const nsp_lstm_stack_params* p, unsigned int* sync, void* stream) {
  int rc = lstm_persistent_check(p);
  if (rc != NSP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(sync, 0, 2 * sizeof(unsigned int), st);
  const dim3 grid(p->H / 16, p->nl);
  const int nb = nsp_cdiv(p->B, 16);
  const size_t shmem = (size_t)nb * (sizeof(float) * 16 * 16 * 17 + 4 * 16 * 16 * 2);
#define LSTM_BWD_P(N)                                                                                         \
  do {                                                                                                        \
    (void)hipFuncSetAttribute((const void*)lstm_stack_bwd_persistent_kernel<N>,                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                        \
    if (!lstm_grid_fits((const void*)lstm_stack_bwd_persistent_kernel<N>, 1024, shmem, grid.x * grid.y))      \
      return NSP_EUNSUPPORTED;                                                                                \
    hipLaunchKernelGGL((lstm_stack_bwd_persistent_kernel<N>), grid, dim3(1024), shmem, st, *p, sync);         \
  } while (0)
  switch (nb) {
    case 1: LSTM_BWD_P(1); break;
    case 2: LSTM_BWD_P(2); break;
    case 3: LSTM_BWD_P(3); break;
    default: LSTM_BWD_P(4); break;
  }
#undef LSTM_BWD_P
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
